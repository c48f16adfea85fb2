// C ABI of libskeres_amd.so (include/skeres_amd.h).  No C++ types or
// exceptions cross this boundary.
#include <cmath>
#include <cstdlib>
#include <new>

#include "bal_kernels.hpp"
#include "dense_kernels.hpp"
#include "dense_rows_kernels.hpp"
#include "solver.hpp"

using namespace sk;

struct sk_ptrvec { std::vector<double*> v; };
struct sk_loss_function { LossFunction l; };
struct sk_cost_function { CostFunction c; };
struct sk_problem { Problem p; };
struct sk_local_parameterization { LocalParameterization p; };
struct sk_options { Options o; };
struct sk_summary { Summary s; };
struct sk_solver { std::unique_ptr<SolverBase> impl; };

static std::string g_program_name;

#define SK_GUARD_BEGIN try {
#define SK_GUARD_END(ret)                                                   \
  } catch (const std::bad_alloc&) { set_error("out of host memory"); return ret; } \
  catch (const std::exception& e) { set_error("internal error: %s", e.what()); return ret; } \
  catch (...) { set_error("internal error"); return ret; }

namespace sk {
// Every distinct loss expression is stored once per problem; the key is its content, so freed and
// re-allocated caller objects cannot alias.
int Problem::intern_loss(const LossFunction* l) {
  if (!l || l->nodes.empty()) return -1;
  std::string key(reinterpret_cast<const char*>(l->nodes.data()), l->nodes.size() * sizeof(LossNode));
  auto it = loss_root_of.find(key);
  if (it != loss_root_of.end()) return it->second;
  const int base = (int)loss_nodes.size();
  for (LossNode n : l->nodes) {
    if (n.f >= 0) n.f += base;
    if (n.g >= 0) n.g += base;
    loss_nodes.push_back(n);
  }
  const int root = (int)loss_nodes.size() - 1;
  loss_root_of.emplace(std::move(key), root);
  has_loss = true;
  return root;
}
}  // namespace sk

namespace sk { int rotation_apply(int op, int row_major, int jet_dim, const double* in, int n, double* out); }

namespace sk {
const DevKnobs& dev_knobs() {
  static const DevKnobs knobs = [] {
    DevKnobs k;
    auto num = [](const char* name, int def) { const char* e = getenv(name); return e ? atoi(e) : def; };
    k.chain_server = num("SK_CHOL_CHAIN_SERVER", 1);
    k.bs_resident = num("SK_BS_RESIDENT", 1);
    k.chain_stamps = getenv("SK_CHAIN_STAMPS");
    k.bs_stamps = getenv("SK_BS_STAMPS");
    if (const char* e = getenv("SK_DEBUG")) {
      const std::string topics = std::string(",") + e + ",";
      k.debug_queues = topics.find(",queues,") != std::string::npos;
      k.debug_envelope = topics.find(",envelope,") != std::string::npos;
      k.debug_segments = topics.find(",segments,") != std::string::npos;
      k.debug_setup = topics.find(",setup,") != std::string::npos;
      k.debug_chain_abort = topics.find(",chain_abort,") != std::string::npos;
    }
    k.queue_shift = num("SK_QUEUE_SHIFT", 0);
    k.chain_queues = num("SK_CHAIN_QUEUES", -1);
    k.pair_max_trailing = num("SK_CHAIN_PAIR_MAX_TRAILING", 0);
    k.dissect_at = num("SK_DISSECT_AT", -1);
    k.schedule_plain = num("SK_SCHEDULE_PLAIN", 0);
    k.chain_xcd_local = num("SK_CHAIN_XCD_LOCAL", 0);
    if (const char* e = getenv("SK_BS_PAIR")) k.bs_pair = atoi(e);
    if (const char* e = getenv("SK_BS_SPREAD")) k.bs_spread = atoi(e);
    if (const char* e = getenv("SK_CHAIN_EARLY_SERVER")) k.chain_early_server = atoi(e);
    if (const char* e = getenv("SK_BULK_RESERVE")) {
      k.bulk_reserve = atoi(e);
      const char* c = strchr(e, ',');
      k.bulk_reserve_early = c ? atoi(c + 1) : k.bulk_reserve;
    }
    return k;
  }();
  return knobs;
}
}  // namespace sk

extern "C" {

const char* sk_version(void) { return "skeres_amd 0.1 (gfx950)"; }
const char* sk_last_error(void) { return get_error(); }
void sk_init_logging(const char* name) { g_program_name = name ? name : ""; }
int sk_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

// ---- DoubleArray ------------------------------------------------------------
double* sk_array_new(int n) { if (n < 0) return nullptr; return static_cast<double*>(std::calloc((size_t)(n > 0 ? n : 1), sizeof(double))); }
void sk_array_free(double* a) { std::free(a); }
double sk_array_getitem(const double* a, int i) { return a[i]; }
void sk_array_setitem(double* a, int i, double v) { a[i] = v; }
double* sk_array_slice(double* buffer, int start) { return &buffer[start]; }
void sk_array_copy_in(double* dst, const double* src, int n) { if (n > 0) std::memcpy(dst, src, (size_t)n * sizeof(double)); }
void sk_array_copy_out(const double* src, double* dst, int n) { if (n > 0) std::memcpy(dst, src, (size_t)n * sizeof(double)); }

// ---- DoubleMatrix -------------------------------------------------------------
int sk_matrix_is_null(double* const* m) { return m == nullptr; }
double* sk_matrix_row(double* const* m, int i) { return m[i]; }

// ---- StdVectorDoublePointer -----------------------------------------------------
sk_ptrvec* sk_ptrvec_new(void) { return new (std::nothrow) sk_ptrvec(); }
void sk_ptrvec_free(sk_ptrvec* v) { delete v; }
void sk_ptrvec_add(sk_ptrvec* v, double* p) { v->v.push_back(p); }
int sk_ptrvec_size(const sk_ptrvec* v) { return (int)v->v.size(); }
double* sk_ptrvec_get(const sk_ptrvec* v, int i) { return v->v[i]; }
void sk_ptrvec_set(sk_ptrvec* v, int i, double* p) { v->v[i] = p; }
double** sk_ptrvec_to_pointer_pointer(sk_ptrvec* v) { return v->v.empty() ? nullptr : v->v.data(); }

// ---- Rotation ---------------------------------------------------------------------
int sk_rotation_apply(int op, int row_major, int jet_dim, const double* in, int n, double* out) {
  SK_GUARD_BEGIN
  return sk::rotation_apply(op, row_major, jet_dim, in, n, out);
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

// ---- LossFunction -----------------------------------------------------------------
sk_loss_function* sk_loss_trivial(void) { return new (std::nothrow) sk_loss_function(); }
void sk_loss_free(sk_loss_function* l) { delete l; }

static sk_loss_function* loss_leaf_new(int type, double a, double b) {
  if (!(a > 0.0) || (type == kLossTolerant && !(b > 0.0))) { set_error("loss function parameters must be positive"); return nullptr; }
  sk_loss_function* l = new (std::nothrow) sk_loss_function();
  if (!l) { set_error("out of host memory"); return nullptr; }
  LossNode n; n.type = type; n.f = n.g = -1; n.depth = 0; n.a = a; n.b = b;
  l->l.nodes.push_back(n);
  return l;
}
sk_loss_function* sk_loss_huber(double a) { return loss_leaf_new(kLossHuber, a, 0.0); }
sk_loss_function* sk_loss_soft_l_one(double a) { return loss_leaf_new(kLossSoftLOne, a, 0.0); }
sk_loss_function* sk_loss_cauchy(double a) { return loss_leaf_new(kLossCauchy, a, 0.0); }
sk_loss_function* sk_loss_tukey(double a) { return loss_leaf_new(kLossTukey, a, 0.0); }
sk_loss_function* sk_loss_tolerant(double a, double b) { return loss_leaf_new(kLossTolerant, a, b); }

// appends a copy of `child` to `dst`, returns its root index there (-1 for a NULL / trivial child) and its depth
static int loss_append(std::vector<LossNode>* dst, const sk_loss_function* child, int* depth) {
  *depth = -1;
  if (!child || child->l.nodes.empty()) return -1;
  const int base = (int)dst->size();
  for (LossNode n : child->l.nodes) {
    if (n.f >= 0) n.f += base;
    if (n.g >= 0) n.g += base;
    dst->push_back(n);
  }
  *depth = dst->back().depth;
  return (int)dst->size() - 1;
}
sk_loss_function* sk_loss_composed(const sk_loss_function* f, const sk_loss_function* g) {
  SK_GUARD_BEGIN
  sk_loss_function* l = new sk_loss_function();
  int df = -1, dg = -1;
  LossNode n; n.type = kLossComposed; n.a = n.b = 0.0;
  n.g = loss_append(&l->l.nodes, g, &dg);
  n.f = loss_append(&l->l.nodes, f, &df);
  n.depth = 1 + (df > dg ? df : dg);
  if (n.depth < 1) n.depth = 1;
  if (n.depth > kLossMaxDepth) { delete l; set_error("loss functions nested deeper than %d", kLossMaxDepth); return nullptr; }
  l->l.nodes.push_back(n);
  return l;
  SK_GUARD_END(nullptr)
}
sk_loss_function* sk_loss_scaled(const sk_loss_function* rho, double a) {
  SK_GUARD_BEGIN
  sk_loss_function* l = new sk_loss_function();
  int d = -1;
  LossNode n; n.type = kLossScaled; n.a = a; n.b = 0.0; n.g = -1;
  n.f = loss_append(&l->l.nodes, rho, &d);
  n.depth = 1 + (d > 0 ? d : 0);
  if (n.depth > kLossMaxDepth) { delete l; set_error("loss functions nested deeper than %d", kLossMaxDepth); return nullptr; }
  l->l.nodes.push_back(n);
  return l;
  SK_GUARD_END(nullptr)
}

__global__ void loss_evaluate_kernel(const LossNode* nodes, int root, const double* s, int n, double* rho) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r[3];
  loss_evaluate(nodes, root, s[i], r);
  rho[3 * i] = r[0]; rho[3 * i + 1] = r[1]; rho[3 * i + 2] = r[2];
}
// LossFunction::Evaluate for n squared norms at once, ON THE DEVICE (the code the solvers run)
int sk_loss_evaluate(const sk_loss_function* loss, const double* sq_norm, int n, double* rho) {
  SK_GUARD_BEGIN
  if (n < 0 || (n > 0 && (!sq_norm || !rho))) { set_error("sk_loss_evaluate: bad arguments"); return SK_ERR_INVALID_ARGUMENT; }
  if (n == 0) return SK_OK;
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  DevBuf<LossNode> dn; DevBuf<double> ds, dr;
  std::vector<LossNode> nodes;
  if (loss) nodes = loss->l.nodes;
  const int root = nodes.empty() ? -1 : (int)nodes.size() - 1;
  if (nodes.empty()) { LossNode t; t.type = kLossTrivial; t.f = t.g = -1; t.depth = 0; t.a = t.b = 0.0; nodes.push_back(t); }
  SK_HIP_TRY(dn.upload(nodes, nullptr));
  SK_HIP_TRY(ds.upload(std::vector<double>(sq_norm, sq_norm + n), nullptr));
  SK_HIP_TRY(dr.alloc(3 * (size_t)n));
  hipLaunchKernelGGL(loss_evaluate_kernel, dim3((n + 127) / 128), dim3(128), 0, nullptr, dn.p, root, ds.p, n, dr.p);
  SK_HIP_TRY(hipGetLastError());
  SK_HIP_TRY(hipMemcpy(rho, dr.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

// ---- PredefinedLocalParameterizations (ceres.i:186-210) ----------------------------
static sk_local_parameterization* param_new(int type, int global_size, int local_size, unsigned mask) {
  if (global_size < 1 || global_size > kParamMaxSize) { set_error("local parameterization: size %d outside 1..%d", global_size, kParamMaxSize); return nullptr; }
  sk_local_parameterization* h = new (std::nothrow) sk_local_parameterization();
  if (!h) { set_error("out of host memory"); return nullptr; }
  h->p.type = type; h->p.global_size = global_size; h->p.local_size = local_size; h->p.constant_mask = mask;
  return h;
}
sk_local_parameterization* sk_local_parameterization_identity(int size) { return param_new(kParamIdentity, size, size, 0u); }
sk_local_parameterization* sk_local_parameterization_subset(int size, const int* constant_parameters, int num_constant) {
  if (num_constant < 0 || (num_constant > 0 && !constant_parameters)) { set_error("subset parameterization: bad arguments"); return nullptr; }
  unsigned mask = 0;
  for (int i = 0; i < num_constant; ++i) {
    const int c = constant_parameters[i];
    if (c < 0 || c >= size) { set_error("subset parameterization: constant index %d outside [0, %d)", c, size); return nullptr; }  // ceres: "Indices indicating constant parameter must be ... "
    if ((mask >> c) & 1u) { set_error("subset parameterization: the set of constant parameters cannot contain duplicates"); return nullptr; }
    mask |= 1u << c;
  }
  return param_new(kParamSubset, size, size - num_constant, mask);
}
sk_local_parameterization* sk_local_parameterization_quaternion(void) { return param_new(kParamQuaternion, 4, 3, 0u); }
sk_local_parameterization* sk_local_parameterization_homogeneous_vector(int size) {
  if (size < 2) { set_error("homogeneous vector parameterization: the size of the homogeneous vector needs to be greater than 1"); return nullptr; }
  return param_new(kParamHomogeneousVector, size, size - 1, 0u);
}
void sk_local_parameterization_free(sk_local_parameterization* p) { delete p; }
int sk_local_parameterization_global_size(const sk_local_parameterization* p) { return p ? p->p.global_size : -1; }
int sk_local_parameterization_local_size(const sk_local_parameterization* p) { return p ? p->p.local_size : -1; }

__global__ void param_apply_kernel(ParamBlock pb, const double* x, const double* delta, int n, double* x_plus, double* jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (x_plus) param_plus(pb, x + (size_t)i * pb.global_size, delta + (size_t)i * (pb.local_size > 0 ? pb.local_size : 1), x_plus + (size_t)i * pb.global_size);
  if (jac) param_jacobian(pb, x + (size_t)i * pb.global_size, jac + (size_t)i * pb.global_size * pb.local_size);
}
static ParamBlock param_block_of(const LocalParameterization& lp) {
  ParamBlock pb; pb.type = lp.type; pb.global_size = lp.global_size; pb.local_size = lp.local_size; pb.constant_mask = lp.constant_mask;
  pb.global_off = pb.local_off = 0;
  return pb;
}
// n points at once, ON THE DEVICE (the code the solver runs): x [n][global], delta [n][local] -> x_plus [n][global]
int sk_local_parameterization_plus(const sk_local_parameterization* p, const double* x, const double* delta, int n, double* x_plus) {
  SK_GUARD_BEGIN
  if (!p || n < 0 || (n > 0 && (!x || !delta || !x_plus))) { set_error("sk_local_parameterization_plus: bad arguments"); return SK_ERR_INVALID_ARGUMENT; }
  if (n == 0) return SK_OK;
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  const int gs = p->p.global_size, ls = std::max(p->p.local_size, 1);
  DevBuf<double> dx, dd, dout;
  SK_HIP_TRY(dx.upload(std::vector<double>(x, x + (size_t)n * gs), nullptr));
  SK_HIP_TRY(dd.upload(std::vector<double>(delta, delta + (size_t)n * ls), nullptr));
  SK_HIP_TRY(dout.alloc((size_t)n * gs));
  hipLaunchKernelGGL(param_apply_kernel, dim3((n + 127) / 128), dim3(128), 0, nullptr, param_block_of(p->p), (const double*)dx.p, (const double*)dd.p, n, dout.p, (double*)nullptr);
  SK_HIP_TRY(hipGetLastError());
  SK_HIP_TRY(hipMemcpy(x_plus, dout.p, (size_t)n * gs * sizeof(double), hipMemcpyDeviceToHost));
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
// jacobian [n][global x local], row-major
int sk_local_parameterization_compute_jacobian(const sk_local_parameterization* p, const double* x, int n, double* jacobian) {
  SK_GUARD_BEGIN
  if (!p || n < 0 || (n > 0 && (!x || !jacobian))) { set_error("sk_local_parameterization_compute_jacobian: bad arguments"); return SK_ERR_INVALID_ARGUMENT; }
  if (n == 0 || p->p.local_size == 0) return SK_OK;
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  const int gs = p->p.global_size, ls = p->p.local_size;
  DevBuf<double> dx, dj;
  SK_HIP_TRY(dx.upload(std::vector<double>(x, x + (size_t)n * gs), nullptr));
  SK_HIP_TRY(dj.alloc((size_t)n * gs * ls));
  hipLaunchKernelGGL(param_apply_kernel, dim3((n + 127) / 128), dim3(128), 0, nullptr, param_block_of(p->p), (const double*)dx.p, (const double*)nullptr, n, (double*)nullptr, dj.p);
  SK_HIP_TRY(hipGetLastError());
  SK_HIP_TRY(hipMemcpy(jacobian, dj.p, (size_t)n * gs * ls * sizeof(double), hipMemcpyDeviceToHost));
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

// ---- CostFunction -----------------------------------------------------------------
sk_cost_function* sk_cost_function_new_autodiff(int functor_id, const double* consts, int num_consts) {
  FunctorDesc d;
  if (!functor_desc(functor_id, &d)) { set_error("unknown device functor id %d", functor_id); return nullptr; }
  if (num_consts != d.num_consts || (num_consts > 0 && !consts)) { set_error("functor %d captures %d constants, %d given", functor_id, d.num_consts, num_consts); return nullptr; }
  sk_cost_function* cf = new (std::nothrow) sk_cost_function();
  if (!cf) { set_error("out of host memory"); return nullptr; }
  cf->c.functor_id = functor_id;
  cf->c.consts.assign(consts, consts + num_consts);
  cf->c.num_residuals = d.num_residuals;
  cf->c.block_sizes.assign(d.block_sizes, d.block_sizes + d.num_blocks);
  return cf;
}

sk_cost_function* sk_cost_function_new_callback(sk_evaluate_fn fn, void* user, int num_residuals, const int* block_sizes, int num_blocks) {
  // CostFunctor / SizedCostFunction validation (CORE/CostFunctor.scala:31-34, CORE/SizedCostFunction.scala:7-11)
  if (!fn) { set_error("null evaluate callback"); return nullptr; }
  if (num_residuals <= 0) { set_error("Nonpositive number of residuals specified: %d", num_residuals); return nullptr; }
  if (num_blocks <= 0 || !block_sizes) { set_error("a cost function needs at least one parameter block"); return nullptr; }
  for (int i = 0; i < num_blocks; ++i) if (block_sizes[i] <= 0) { set_error("Nonpositive parameter block sizes specified"); return nullptr; }
  sk_cost_function* cf = new (std::nothrow) sk_cost_function();
  if (!cf) { set_error("out of host memory"); return nullptr; }
  cf->c.functor_id = SK_FUNCTOR_HOST_CALLBACK; cf->c.callback = fn; cf->c.user = user; cf->c.num_residuals = num_residuals;
  cf->c.block_sizes.assign(block_sizes, block_sizes + num_blocks);
  return cf;
}
sk_cost_function* sk_cost_function_new_tape(int num_residuals, const int* block_sizes, int num_blocks, const int* instructions, int num_instructions,
                                            const double* tape_constants, int num_tape_constants, int num_registers, const int* output_operands,
                                            const double* captured, int num_captured) {
  SK_GUARD_BEGIN
  // CostFunctor / SizedCostFunction validation (CORE/CostFunctor.scala:31-34, CORE/SizedCostFunction.scala:7-11), then the tape's own
  if (num_residuals <= 0) { set_error("Nonpositive number of residuals specified: %d", num_residuals); return nullptr; }
  if (num_blocks <= 0 || !block_sizes) { set_error("a cost function needs at least one parameter block"); return nullptr; }
  if (num_instructions < 0 || (num_instructions > 0 && !instructions) || num_tape_constants < 0 || (num_tape_constants > 0 && !tape_constants) ||
      !output_operands || num_captured < 0 || (num_captured > 0 && !captured)) { set_error("invalid argument"); return nullptr; }
  auto t = std::make_shared<Tape>();
  t->num_residuals = num_residuals; t->num_registers = num_registers; t->num_obs_consts = num_captured;
  t->block_sizes.assign(block_sizes, block_sizes + num_blocks);
  t->ins.resize((size_t)num_instructions);
  for (int i = 0; i < num_instructions; ++i) t->ins[i] = TapeIns{instructions[5 * i], instructions[5 * i + 1], instructions[5 * i + 2], instructions[5 * i + 3], instructions[5 * i + 4]};
  t->consts.assign(tape_constants, tape_constants + num_tape_constants);
  t->out.assign(output_operands, output_operands + num_residuals);
  const std::string why = tape_validate(*t);
  if (!why.empty()) { set_error("%s", why.c_str()); return nullptr; }
  sk_cost_function* cf = new (std::nothrow) sk_cost_function();
  if (!cf) { set_error("out of host memory"); return nullptr; }
  cf->c.functor_id = SK_FUNCTOR_TAPE; cf->c.num_residuals = num_residuals;
  cf->c.block_sizes = t->block_sizes;
  cf->c.consts.assign(captured, captured + num_captured);
  cf->c.tape = t;
  return cf;
  SK_GUARD_END(nullptr)
}
void sk_cost_function_free(sk_cost_function* cf) { delete cf; }
int sk_cost_function_num_residuals(const sk_cost_function* cf) { return cf->c.num_residuals; }
int sk_cost_function_num_parameter_blocks(const sk_cost_function* cf) { return (int)cf->c.block_sizes.size(); }
int sk_cost_function_parameter_block_size(const sk_cost_function* cf, int i) { return cf->c.block_sizes[i]; }

int sk_cost_function_evaluate(const sk_cost_function* cf, double const* const* parameters, double* residuals, double** jacobians) {
  SK_GUARD_BEGIN
  const CostFunction& c = cf->c;
  if (c.functor_id == SK_FUNCTOR_HOST_CALLBACK) return c.callback(c.user, parameters, residuals, jacobians) ? 1 : 0;
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return -SK_ERR_NO_DEVICE; }
  const int nb = (int)c.block_sizes.size(), nres = c.num_residuals;
  std::vector<int> x_off(nb), j_off(nb);
  int nx = 0, nj = 0; unsigned mask = 0;
  for (int q = 0; q < nb; ++q) { x_off[q] = nx; nx += c.block_sizes[q]; j_off[q] = nj; nj += nres * c.block_sizes[q]; if (jacobians && jacobians[q]) mask |= 1u << q; }
  std::vector<double> x(nx);
  for (int q = 0; q < nb; ++q) std::memcpy(&x[x_off[q]], parameters[q], c.block_sizes[q] * sizeof(double));
  DevBuf<double> dx, dc, dr, dj; DevBuf<int> dxo, djo, dok;
  std::vector<double> consts = c.consts; if (consts.empty()) consts.push_back(0.0);
  hipStream_t s = nullptr;
#define SK_TRYN(e) do { if ((e) != hipSuccess) { set_error("HIP error in sk_cost_function_evaluate"); return -SK_ERR_HIP; } } while (0)
  SK_TRYN(dx.upload(x, s)); SK_TRYN(dc.upload(consts, s)); SK_TRYN(dxo.upload(x_off, s)); SK_TRYN(djo.upload(j_off, s));
  SK_TRYN(dr.alloc(nres)); SK_TRYN(dj.alloc(nj)); SK_TRYN(dok.alloc(1));
  if (c.functor_id == SK_FUNCTOR_TAPE) {
    TapeDevBuffers tb;
    SK_TRYN(tb.upload(*c.tape, s));
    if (!launch_single_eval_tape(tb, dc.p, dx.p, dxo.p, dr.p, dj.p, djo.p, jacobians ? 1 : 0, mask, dok.p, s)) {
      set_error("the recorded functor needs %d registers: more than the device interpreter holds", c.tape->num_registers);
      return -SK_ERR_UNSUPPORTED;
    }
  } else {
    launch_single_eval(c.functor_id, dc.p, dx.p, dxo.p, dr.p, dj.p, djo.p, jacobians ? 1 : 0, mask, dok.p, s);
  }
  int ok = 0;
  std::vector<double> r(nres), j(nj);
  SK_TRYN(hipMemcpy(&ok, dok.p, sizeof(int), hipMemcpyDeviceToHost));
  SK_TRYN(hipMemcpy(r.data(), dr.p, nres * sizeof(double), hipMemcpyDeviceToHost));
  SK_TRYN(hipMemcpy(j.data(), dj.p, nj * sizeof(double), hipMemcpyDeviceToHost));
#undef SK_TRYN
  if (!ok) return 0;
  std::memcpy(residuals, r.data(), nres * sizeof(double));
  for (int q = 0; q < nb; ++q) if ((mask >> q) & 1u) std::memcpy(jacobians[q], &j[j_off[q]], (size_t)nres * c.block_sizes[q] * sizeof(double));
  return 1;
  SK_GUARD_END(-SK_ERR_INVALID_ARGUMENT)
}

// ---- Problem ------------------------------------------------------------------------
sk_problem* sk_problem_new(void) { return new (std::nothrow) sk_problem(); }
void sk_problem_free(sk_problem* p) { delete p; }

static int register_block(Problem& P, double* ptr, int size) {
  if (!ptr) { set_error("null parameter block pointer"); return -1; }
  auto it = P.block_of.find(ptr);
  if (it != P.block_of.end()) {
    if (P.block_size[it->second] != size) { set_error("parameter block %p was registered with size %d, now used with size %d", (void*)ptr, P.block_size[it->second], size); return -1; }
    return it->second;
  }
  const int id = (int)P.block_ptr.size();
  P.block_of.emplace(ptr, id); P.block_ptr.push_back(ptr); P.block_size.push_back(size);
  return id;
}

// ceres::Problem::AddParameterBlock / SetParameterization / SetParameterBlockConstant / SetParameterBlockVariable,
// which the reference's Problem inherits (CORE/Problem.scala:16)
static int set_block_param(Problem& P, int id, const sk_local_parameterization* lp) {
  P.block_param.resize(P.block_ptr.size(), -1);
  if (!lp) { P.block_param[id] = -1; return SK_OK; }
  if (lp->p.global_size != P.block_size[id]) {
    set_error("local parameterization of global size %d set on a parameter block of size %d", lp->p.global_size, P.block_size[id]);
    return SK_ERR_INVALID_ARGUMENT;
  }
  P.params.push_back(lp->p);
  P.block_param[id] = (int)P.params.size() - 1;
  return SK_OK;
}
int sk_problem_add_parameter_block(sk_problem* p, double* values, int size, const sk_local_parameterization* parameterization) {
  SK_GUARD_BEGIN
  if (!p || size < 1) { set_error("sk_problem_add_parameter_block: bad arguments"); return SK_ERR_INVALID_ARGUMENT; }
  const int id = register_block(p->p, values, size);
  if (id < 0) return SK_ERR_INVALID_ARGUMENT;
  return parameterization ? set_block_param(p->p, id, parameterization) : SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
static int find_block(Problem& P, double* values) {
  auto it = P.block_of.find(values);
  if (it == P.block_of.end()) { set_error("parameter block %p is not part of the problem", (void*)values); return -1; }
  return it->second;
}
int sk_problem_set_parameterization(sk_problem* p, double* values, const sk_local_parameterization* parameterization) {
  SK_GUARD_BEGIN
  if (!p) { set_error("null problem"); return SK_ERR_INVALID_ARGUMENT; }
  const int id = find_block(p->p, values);
  return id < 0 ? SK_ERR_INVALID_ARGUMENT : set_block_param(p->p, id, parameterization);
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
static int set_block_constant(sk_problem* p, double* values, char constant) {
  if (!p) { set_error("null problem"); return SK_ERR_INVALID_ARGUMENT; }
  const int id = find_block(p->p, values);
  if (id < 0) return SK_ERR_INVALID_ARGUMENT;
  p->p.block_constant.resize(p->p.block_ptr.size(), 0);
  p->p.block_constant[id] = constant;
  return SK_OK;
}
int sk_problem_set_parameter_block_constant(sk_problem* p, double* values) { return set_block_constant(p, values, 1); }
int sk_problem_set_parameter_block_variable(sk_problem* p, double* values) { return set_block_constant(p, values, 0); }

int sk_problem_add_residual_block(sk_problem* p, const sk_cost_function* cost, const sk_loss_function* loss, double* const* parameter_blocks,
                                  int num_parameter_blocks, sk_residual_block_id* id_out) {
  SK_GUARD_BEGIN
  if (!p || !cost || !parameter_blocks) { set_error("null argument"); return SK_ERR_INVALID_ARGUMENT; }
  const CostFunction& c = cost->c;
  if (num_parameter_blocks != (int)c.block_sizes.size()) { set_error("cost function expects %d parameter blocks, %d given", (int)c.block_sizes.size(), num_parameter_blocks); return SK_ERR_INVALID_ARGUMENT; }
  Problem& P = p->p;
  std::vector<int> ids(num_parameter_blocks);
  for (int q = 0; q < num_parameter_blocks; ++q) {
    ids[q] = register_block(P, parameter_blocks[q], c.block_sizes[q]);
    if (ids[q] < 0) return SK_ERR_INVALID_ARGUMENT;
    for (int t = 0; t < q; ++t) if (ids[t] == ids[q]) { set_error("duplicate parameter blocks in a residual block are not allowed"); return SK_ERR_INVALID_ARGUMENT; }
  }
  P.rb_functor.push_back(c.functor_id == SK_FUNCTOR_TAPE ? kTapeFunctorBase + P.intern_tape(c.tape) : c.functor_id);
  P.rb_num_residuals.push_back(c.num_residuals);
  P.rb_const_off.push_back(P.consts.size()); P.consts.insert(P.consts.end(), c.consts.begin(), c.consts.end());
  P.rb_pidx.insert(P.rb_pidx.end(), ids.begin(), ids.end()); P.rb_pidx_off.push_back(P.rb_pidx.size());
  P.rb_cost.push_back(c.functor_id == SK_FUNCTOR_HOST_CALLBACK ? &c : nullptr);
  P.rb_loss.push_back(P.intern_loss(loss ? &loss->l : nullptr));
  if (c.functor_id == SK_FUNCTOR_HOST_CALLBACK) P.has_callbacks = true;
  P.num_residuals += c.num_residuals;
  if (id_out) *id_out = (int)P.rb_functor.size() - 1;
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_problem_add_residual_blocks(sk_problem* p, int functor_id, int n, const double* consts, const sk_loss_function* loss, double* const* parameter_blocks) {
  SK_GUARD_BEGIN
  FunctorDesc d;
  if (!p || n < 0 || !parameter_blocks) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (!functor_desc(functor_id, &d)) { set_error("unknown device functor id %d", functor_id); return SK_ERR_INVALID_ARGUMENT; }
  if (d.num_consts > 0 && !consts) { set_error("functor %d needs %d constants per block", functor_id, d.num_consts); return SK_ERR_INVALID_ARGUMENT; }
  Problem& P = p->p;
  const int loss_root = P.intern_loss(loss ? &loss->l : nullptr);
  P.rb_functor.reserve(P.rb_functor.size() + n); P.rb_pidx.reserve(P.rb_pidx.size() + (size_t)n * d.num_blocks);
  P.consts.reserve(P.consts.size() + (size_t)n * d.num_consts);
  for (int b = 0; b < n; ++b) {
    int ids[10];
    for (int q = 0; q < d.num_blocks; ++q) {
      ids[q] = register_block(P, parameter_blocks[(size_t)b * d.num_blocks + q], d.block_sizes[q]);
      if (ids[q] < 0) return SK_ERR_INVALID_ARGUMENT;
      for (int t = 0; t < q; ++t) if (ids[t] == ids[q]) { set_error("duplicate parameter blocks in a residual block are not allowed"); return SK_ERR_INVALID_ARGUMENT; }
    }
    P.rb_functor.push_back(functor_id); P.rb_num_residuals.push_back(d.num_residuals);
    P.rb_const_off.push_back(P.consts.size());
    if (d.num_consts) P.consts.insert(P.consts.end(), consts + (size_t)b * d.num_consts, consts + (size_t)(b + 1) * d.num_consts);
    P.rb_pidx.insert(P.rb_pidx.end(), ids, ids + d.num_blocks); P.rb_pidx_off.push_back(P.rb_pidx.size());
    P.rb_cost.push_back(nullptr);
    P.rb_loss.push_back(loss_root);
    P.num_residuals += d.num_residuals;
  }
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
int sk_problem_add_residual_blocks_tape(sk_problem* p, const sk_cost_function* cost, int n, const double* captured, const sk_loss_function* loss,
                                        double* const* parameter_blocks) {
  SK_GUARD_BEGIN
  if (!p || !cost || n < 0 || !parameter_blocks) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (cost->c.functor_id != SK_FUNCTOR_TAPE || !cost->c.tape) { set_error("not a recorded functor (sk_cost_function_new_tape)"); return SK_ERR_INVALID_ARGUMENT; }
  const Tape& t = *cost->c.tape;
  if (t.num_obs_consts > 0 && !captured) { set_error("the recorded functor captures %d doubles per block", t.num_obs_consts); return SK_ERR_INVALID_ARGUMENT; }
  Problem& P = p->p;
  const int loss_root = P.intern_loss(loss ? &loss->l : nullptr);
  const int fid = kTapeFunctorBase + P.intern_tape(cost->c.tape), nbk = (int)t.block_sizes.size();
  std::vector<int> ids((size_t)nbk);
  for (int b = 0; b < n; ++b) {
    for (int q = 0; q < nbk; ++q) {
      ids[q] = register_block(P, parameter_blocks[(size_t)b * nbk + q], t.block_sizes[q]);
      if (ids[q] < 0) return SK_ERR_INVALID_ARGUMENT;
      for (int u = 0; u < q; ++u) if (ids[u] == ids[q]) { set_error("duplicate parameter blocks in a residual block are not allowed"); return SK_ERR_INVALID_ARGUMENT; }
    }
    P.rb_functor.push_back(fid); P.rb_num_residuals.push_back(t.num_residuals);
    P.rb_const_off.push_back(P.consts.size());
    if (t.num_obs_consts) P.consts.insert(P.consts.end(), captured + (size_t)b * t.num_obs_consts, captured + (size_t)(b + 1) * t.num_obs_consts);
    P.rb_pidx.insert(P.rb_pidx.end(), ids.begin(), ids.end()); P.rb_pidx_off.push_back(P.rb_pidx.size());
    P.rb_cost.push_back(nullptr);
    P.rb_loss.push_back(loss_root);
    P.num_residuals += t.num_residuals;
  }
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
int sk_problem_add_dense_rows(sk_problem* p, int functor_id, int num_rows, const double* consts, const sk_loss_function* loss, double* x, int n) {
  SK_GUARD_BEGIN
  if (!p || num_rows < 0 || !consts || !x || n <= 0) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (functor_id != SK_FUNCTOR_SYNTH_TANH_ROW) { set_error("functor %d is not a dense-row functor", functor_id); return SK_ERR_INVALID_ARGUMENT; }
  Problem& P = p->p;
  const int id = register_block(P, x, n);
  if (id < 0) return SK_ERR_INVALID_ARGUMENT;
  const size_t base = P.rb_functor.size();
  P.rb_functor.resize(base + num_rows, functor_id); P.rb_num_residuals.resize(base + num_rows, 1); P.rb_cost.resize(base + num_rows, nullptr);
  P.rb_loss.resize(base + num_rows, P.intern_loss(loss ? &loss->l : nullptr));  // (one loss for the rows of a call; the solver takes one for all rows)
  P.rb_const_off.reserve(base + num_rows); P.rb_pidx.reserve(P.rb_pidx.size() + num_rows); P.rb_pidx_off.reserve(P.rb_pidx_off.size() + num_rows);
  for (int i = 0; i < num_rows; ++i) {
    P.rb_const_off.push_back(P.consts.size() + 3 * (size_t)i);
    P.rb_pidx.push_back(id); P.rb_pidx_off.push_back(P.rb_pidx.size());
  }
  P.consts.insert(P.consts.end(), consts, consts + 3 * (size_t)num_rows);
  P.num_residuals += num_rows;
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
int sk_problem_num_residual_blocks(const sk_problem* p) { return (int)p->p.rb_functor.size(); }
int sk_problem_num_parameter_blocks(const sk_problem* p) { return (int)p->p.block_size.size(); }
int sk_problem_num_parameters(const sk_problem* p) { return p->p.num_parameters(); }
int sk_problem_num_residuals(const sk_problem* p) { return (int)p->p.num_residuals; }

// ---- Options --------------------------------------------------------------------------
sk_options* sk_options_new(void) { return new (std::nothrow) sk_options(); }
void sk_options_free(sk_options* o) { delete o; }
int sk_options_set_linear_solver_type(sk_options* o, int t) {
  if (t != SK_DENSE_NORMAL_CHOLESKY && t != SK_DENSE_QR && t != SK_DENSE_SCHUR) { set_error("linear solver type %s is not implemented (DENSE_QR, DENSE_NORMAL_CHOLESKY, DENSE_SCHUR are)", linear_solver_name(t)); return SK_ERR_UNSUPPORTED; }
  o->o.linear_solver_type = t; return SK_OK;
}
int sk_options_set_minimizer_type(sk_options* o, int t) {
  if (t != SK_TRUST_REGION) { set_error("only TRUST_REGION is implemented"); return SK_ERR_UNSUPPORTED; }
  o->o.minimizer_type = t; return SK_OK;
}
int sk_options_set_max_num_iterations(sk_options* o, int n) { if (n < 0) { set_error("max_num_iterations must be >= 0"); return SK_ERR_INVALID_ARGUMENT; } o->o.max_num_iterations = n; return SK_OK; }
int sk_options_set_minimizer_progress_to_stdout(sk_options* o, int on) { o->o.progress_to_stdout = on != 0; return SK_OK; }
#define SK_SET_D(NAME, FIELD) int sk_options_set_##NAME(sk_options* o, double v) { if (!(v >= 0.0)) { set_error(#NAME " must be >= 0"); return SK_ERR_INVALID_ARGUMENT; } o->o.FIELD = v; return SK_OK; }
SK_SET_D(function_tolerance, function_tolerance)
SK_SET_D(gradient_tolerance, gradient_tolerance)
SK_SET_D(parameter_tolerance, parameter_tolerance)
SK_SET_D(initial_trust_region_radius, initial_trust_region_radius)
SK_SET_D(max_trust_region_radius, max_trust_region_radius)
SK_SET_D(min_trust_region_radius, min_trust_region_radius)
SK_SET_D(min_relative_decrease, min_relative_decrease)
SK_SET_D(min_lm_diagonal, min_lm_diagonal)
SK_SET_D(max_lm_diagonal, max_lm_diagonal)
int sk_options_set_jacobi_scaling(sk_options* o, int on) { o->o.jacobi_scaling = on != 0; return SK_OK; }
int sk_options_set_max_num_consecutive_invalid_steps(sk_options* o, int n) { if (n < 1) { set_error("must be >= 1"); return SK_ERR_INVALID_ARGUMENT; } o->o.max_num_consecutive_invalid_steps = n; return SK_OK; }
int sk_options_set_device(sk_options* o, int dev) { o->o.device = dev; return SK_OK; }
int sk_options_set_cholesky_tuning(sk_options* o, int group, int lookahead) {
  if (group > 64) { set_error("group must be <= 64"); return SK_ERR_INVALID_ARGUMENT; }
  if (group > 0) o->o.cholesky_group = group;
  o->o.lookahead = lookahead != 0;
  return SK_OK;
}
int sk_options_set_stream(sk_options* o, void* stream) { o->o.stream = (hipStream_t)stream; o->o.stream_set = true; return SK_OK; }
int sk_options_set_distributed(sk_options* o, int rank, int world, sk_allreduce_fn fn, void* user) {
  if (world < 1 || rank < 0 || rank >= world) { set_error("invalid rank/world %d/%d", rank, world); return SK_ERR_INVALID_ARGUMENT; }
  if (world > 1 && !fn) { set_error("world > 1 needs an allreduce hook"); return SK_ERR_INVALID_ARGUMENT; }
  o->o.rank = rank; o->o.world = world; o->o.allreduce = fn; o->o.allreduce_user = user; return SK_OK;
}
int sk_options_set_cholesky_envelope(sk_options* o, int on) { o->o.envelope = on != 0; return SK_OK; }
int sk_options_set_cholesky_dissection(sk_options* o, int mode) {
  if (mode != SK_DISSECTION_AUTO && mode != SK_DISSECTION_ON && mode != SK_DISSECTION_OFF) { set_error("invalid dissection mode %d", mode); return SK_ERR_INVALID_ARGUMENT; }
  o->o.dissection = mode; return SK_OK;
}
int sk_options_set_distribution_mode(sk_options* o, int mode) {
  if (mode != SK_DISTRIBUTION_AUTO && mode != SK_DISTRIBUTION_SHARDED && mode != SK_DISTRIBUTION_REPLICATED && mode != SK_DISTRIBUTION_SEGMENTED) { set_error("invalid distribution mode %d", mode); return SK_ERR_INVALID_ARGUMENT; }
  o->o.distribution_mode = mode; return SK_OK;
}
int sk_options_set_resident_kernels(sk_options* o, int on) { o->o.resident_kernels = on != 0; return SK_OK; }
int sk_options_set_graph_replay(sk_options* o, int on) { o->o.graph_replay = on != 0; return SK_OK; }
int sk_options_set_max_segments(sk_options* o, int n) {
  if (n < 0 || n == 1) { set_error("max_segments must be 0 (one per rank) or at least 2"); return SK_ERR_INVALID_ARGUMENT; }
  o->o.max_segments = n; return SK_OK;
}
int sk_options_set_cholesky_border(sk_options* o, int mode) {
  if (mode != SK_BORDER_AUTO && mode != SK_BORDER_ON && mode != SK_BORDER_OFF) { set_error("invalid border mode %d", mode); return SK_ERR_INVALID_ARGUMENT; }
  o->o.border = mode; return SK_OK;
}
int sk_options_set_retained_points(sk_options* o, int mode, int max_points) {
  if (mode != SK_RETAINED_AUTO && mode != SK_RETAINED_ON && mode != SK_RETAINED_OFF) { set_error("invalid retained-points mode %d", mode); return SK_ERR_INVALID_ARGUMENT; }
  if (max_points < 0) { set_error("max_points must not be negative"); return SK_ERR_INVALID_ARGUMENT; }
  o->o.retained = mode; o->o.retained_max = max_points; return SK_OK;
}
int sk_options_set_reduce_buffer(sk_options* o, void* ptr, size_t bytes) { o->o.reduce_buffer = ptr; o->o.reduce_buffer_bytes = bytes; return SK_OK; }
size_t sk_reduce_buffer_bytes(const sk_options* o, const sk_problem* p) {
  // cameras = distinct blocks in parameter slot 0 of the residual blocks
  const Problem& P = p->p;
  std::vector<char> seen(P.block_size.size(), 0); size_t C = 0;
  for (size_t b = 0; b < P.rb_functor.size(); ++b) { const int c = P.rb_pidx[P.rb_pidx_off[b]]; if (!seen[c]) { seen[c] = 1; ++C; } }
  // (+ the pseudo-cameras of retained points, three points each — sk_options_set_retained_points: what the plan retains for this problem,
  // which is a function of host data alone (bal_retained_plan), not the most the options would allow: 512 pseudo-cameras made a 16-camera
  // problem's buffer 97 MB and Ladybug-1723's 1.6 GB instead of 0.98 — ADVICE r04.  A world of ranks may have to drop the memory-order
  // candidate of the camera orders (setup: the ranks compare): the plan without it can retain more, so the larger of the two counts.)
  if (!o || o->o.retained != SK_RETAINED_OFF) {
    const int mode = o ? o->o.retained : SK_RETAINED_AUTO, maxp = o ? o->o.retained_max : 0, border = o ? o->o.border : SK_BORDER_AUTO;
    std::string why;
    size_t kept = 0;
    if (problem_is_bal_shaped(P, &why)) {
      std::vector<int> flags;
      kept = (size_t)std::max(bal_retained_plan(P, mode, maxp, border, &flags, nullptr, nullptr, true), bal_retained_plan(P, mode, maxp, border, &flags, nullptr, nullptr, false));
    }
    C += (kept + 2) / 3;
  }
  const size_t n = 9 * C, npad = ((n + 1 + 127) / 128) * 128;
  return tri_packed_elems((int)(npad / 128)) * sizeof(double);  // lower block triangle, packed
}

// ---- Summary --------------------------------------------------------------------------
sk_summary* sk_summary_new(void) { return new (std::nothrow) sk_summary(); }
void sk_summary_free(sk_summary* s) { delete s; }
double sk_summary_initial_cost(const sk_summary* s) { return s->s.initial_cost; }
double sk_summary_final_cost(const sk_summary* s) { return s->s.final_cost; }
int sk_summary_num_iterations(const sk_summary* s) { return (int)s->s.iterations.size(); }
int sk_summary_num_successful_steps(const sk_summary* s) { return s->s.num_successful_steps; }
int sk_summary_num_unsuccessful_steps(const sk_summary* s) { return s->s.num_unsuccessful_steps; }
int sk_summary_termination_type(const sk_summary* s) { return s->s.termination_type; }
const char* sk_summary_message(const sk_summary* s) { return s->s.message.c_str(); }
const char* sk_summary_brief_report(const sk_summary* s) { return s->s.brief.c_str(); }
const char* sk_summary_full_report(const sk_summary* s) { return s->s.full.c_str(); }
int sk_summary_num_logged_iterations(const sk_summary* s) { return (int)s->s.iterations.size(); }
double sk_summary_iteration_field(const sk_summary* s, int it, int field) {
  if (it < 0 || it >= (int)s->s.iterations.size()) return NAN;
  const IterationLog& L = s->s.iterations[it];
  switch (field) {
    case 0: return L.cost; case 1: return L.cost_change; case 2: return L.gradient_max_norm; case 3: return L.step_norm;
    case 4: return L.relative_decrease; case 5: return L.trust_region_radius; case 6: return L.step_is_valid; case 7: return L.step_is_successful;
  }
  return NAN;
}
int sk_summary_linear_solver_type_used(const sk_summary* s) { return s->s.linear_solver_type; }
int sk_summary_linear_solver_type_given(const sk_summary* s) { return s->s.linear_solver_type_given >= 0 ? s->s.linear_solver_type_given : s->s.linear_solver_type; }
double sk_summary_phase_seconds(const sk_summary* s, int phase) { return (phase >= 0 && phase < 7) ? s->s.phase_seconds[phase] : NAN; }

// ---- solve ------------------------------------------------------------------------------
static std::unique_ptr<SolverBase> make_solver(const Options& o, Problem* p, int* rc) {
  *rc = SK_OK;
  if (p->has_parameterization() && o.linear_solver_type != SK_DENSE_SCHUR && problem_is_dense_rows(*p)) {
    set_error("local parameterizations and constant parameter blocks are implemented for residual-block problems (DENSE_QR / DENSE_NORMAL_CHOLESKY; identity, subset and constant blocks under DENSE_SCHUR), not for dense rows (not supported here)");
    *rc = SK_ERR_UNSUPPORTED; return nullptr;
  }
  if (o.linear_solver_type == SK_DENSE_SCHUR) {
    std::string why;
    if (problem_is_bal_shaped(*p, &why)) return make_bal_solver(o, p);
    // Not the (2; 9, 3) structure the Schur path eliminates.  Ceres, given a Schur-type solver and nothing to eliminate, falls
    // back to its alternate — DENSE_QR for DENSE_SCHUR (trust_region_preprocessor: the LM step is the same whatever solves the
    // linear system) — and reports "Given / Used".  So does this library for residual-block problems (the dense path takes every
    // shape, parameterization and loss; it refuses only what does not fit in device memory); dense rows and local
    // parameterizations on them stay refused, with the reason.
    if (problem_is_dense_rows(*p)) { set_error("%s", why.c_str()); *rc = SK_ERR_UNSUPPORTED; return nullptr; }
    for (int f : p->rb_functor) if (f == SK_FUNCTOR_SYNTH_TANH_ROW) { set_error("%s", why.c_str()); *rc = SK_ERR_UNSUPPORTED; return nullptr; }
    Options alt = o;
    alt.linear_solver_type_given = SK_DENSE_SCHUR;
    alt.linear_solver_type = SK_DENSE_QR;
    return make_dense_solver(alt, p);
  }
  if (problem_is_dense_rows(*p)) {
    if (o.linear_solver_type != SK_DENSE_NORMAL_CHOLESKY) { set_error("dense-row problems are implemented for DENSE_NORMAL_CHOLESKY only (not supported: %s)", linear_solver_name(o.linear_solver_type)); *rc = SK_ERR_UNSUPPORTED; return nullptr; }
    return make_dense_rows_solver(o, p);
  }
  for (int f : p->rb_functor) if (f == SK_FUNCTOR_SYNTH_TANH_ROW) { set_error("dense-row functors cannot be mixed with other residual blocks (not supported)"); *rc = SK_ERR_UNSUPPORTED; return nullptr; }
  return make_dense_solver(o, p);
}

sk_solver* sk_solver_create(const sk_options* options, sk_problem* problem) {
  SK_GUARD_BEGIN
  set_status(SK_OK);
  if (!options || !problem) { set_error("null argument"); set_status(SK_ERR_INVALID_ARGUMENT); return nullptr; }
  int rc;
  std::unique_ptr<SolverBase> impl = make_solver(options->o, &problem->p, &rc);
  if (!impl) { set_status(rc ? rc : SK_ERR_INVALID_ARGUMENT); return nullptr; }
  rc = impl->create();
  if (rc) { set_status(rc); return nullptr; }
  sk_solver* s = new sk_solver();
  s->impl = std::move(impl);
  return s;
  SK_GUARD_END(nullptr)
}
int sk_last_status(void) { return get_status(); }
void sk_solver_free(sk_solver* s) { delete s; }
int sk_solver_step(sk_solver* s, int* done) {
  SK_GUARD_BEGIN
  bool d = false;
  const int rc = s->impl->step(&d);
  if (done) *done = d ? 1 : 0;
  return rc;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
int sk_solver_finish(sk_solver* s, sk_summary* summary) {
  SK_GUARD_BEGIN
  Summary tmp;
  const int rc = s->impl->finish(&tmp);
  if (rc == SK_OK && summary) summary->s = tmp;
  return rc;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}
int sk_solver_set_kernel_timing(sk_solver* s, int on) {
  s->impl->set_kernel_timing(on);
  return SK_OK;
}
double sk_solver_kernel_seconds(const sk_solver* s, const char* name, int* launches) {
  KernelTimer::Stat st = s->impl->kernel_stat(name);
  if (launches) *launches = st.launches;
  return st.seconds;
}
double sk_solver_syrk_flops_per_solve(const sk_solver* s) { return s->impl->syrk_flops_per_solve(); }
double sk_solver_syrk_c_bytes_per_solve(const sk_solver* s) { return s->impl->syrk_c_bytes_per_solve(); }
int sk_solver_distribution(const sk_solver* s, double* allreduce_seconds, double* saved_seconds) { return s->impl->distribution(allreduce_seconds, saved_seconds); }
int sk_solver_stat(const sk_solver* s, const char* name, double* value) {
  SK_GUARD_BEGIN
  if (!s || !name || !value) { set_error("null argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (strncmp(name, "phase_seconds_", 14) == 0 && name[14] >= '0' && name[14] <= '5' && !name[15]) { *value = s->impl->phase_seconds(name[14] - '0'); return SK_OK; }
  if (!s->impl->stat(name, value)) { set_error("this solver reports no figure named '%s'", name); return SK_ERR_INVALID_ARGUMENT; }
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_solve(const sk_options* options, sk_problem* problem, sk_summary* summary) {
  SK_GUARD_BEGIN
  sk_solver* s = sk_solver_create(options, problem);
  if (!s) return get_status() != SK_OK ? get_status() : SK_ERR_INVALID_ARGUMENT;  // the status sk_solver_create recorded (sk_last_status)
  int rc = SK_OK, done = 0;
  while (!done) { rc = sk_solver_step(s, &done); if (rc) break; }
  if (rc == SK_OK) rc = sk_solver_finish(s, summary);
  sk_solver_free(s);
  return rc;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_problem_point_partition(const sk_problem* p, int world, int* cuts, int* num_cameras, int* num_points, int* point_of_block) {
  SK_GUARD_BEGIN
  if (!p || world < 1 || !cuts) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  std::string why;
  if (!problem_is_bal_shaped(p->p, &why)) { set_error("%s", why.c_str()); return SK_ERR_UNSUPPORTED; }
  std::vector<int> cam_block, pt_block, ocam, opt, cut;
  bal_index_problem(p->p, &cam_block, &pt_block, &ocam, &opt);
  bal_partition_points(opt, (int)pt_block.size(), world, &cut);
  for (int r = 0; r <= world; ++r) cuts[r] = cut[r];
  if (num_cameras) *num_cameras = (int)cam_block.size();
  if (num_points) *num_points = (int)pt_block.size();
  if (point_of_block) for (size_t b = 0; b < opt.size(); ++b) point_of_block[b] = opt[b];
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_problem_segment_plan(const sk_problem* p, int max_segments, int forced, int* num_segments, int* camera_part_of_block, int* point_owner_of_block) {
  SK_GUARD_BEGIN
  if (!p || max_segments < 1 || !num_segments) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  std::string why;
  if (!problem_is_bal_shaped(p->p, &why)) { set_error("%s", why.c_str()); return SK_ERR_UNSUPPORTED; }
  std::vector<int> part, owner;
  *num_segments = bal_segment_plan(p->p, max_segments, forced != 0, &part, &owner);
  for (size_t b = 0; b < part.size(); ++b) {
    if (camera_part_of_block) camera_part_of_block[b] = part[b];
    if (point_owner_of_block) point_owner_of_block[b] = owner[b];
  }
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_problem_border_plan(const sk_problem* p, int mode, int* num_border_cameras, int* camera_position_of_block, int* gap, double* model_us, double* model_us_plain,
                           double* envelope_fill) {
  SK_GUARD_BEGIN
  if (!p || !num_border_cameras) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (mode != SK_BORDER_AUTO && mode != SK_BORDER_ON && mode != SK_BORDER_OFF) { set_error("invalid border mode %d", mode); return SK_ERR_INVALID_ARGUMENT; }
  std::string why;
  if (!problem_is_bal_shaped(p->p, &why)) { set_error("%s", why.c_str()); return SK_ERR_UNSUPPORTED; }
  std::vector<int> pos;
  *num_border_cameras = bal_border_plan(p->p, mode, &pos, gap, model_us, model_us_plain, envelope_fill);
  if (camera_position_of_block) for (size_t b = 0; b < pos.size(); ++b) camera_position_of_block[b] = pos[b];
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_problem_retained_plan(const sk_problem* p, int mode, int max_points, int border_mode, int* num_retained, int* retained_of_block, double* model_us,
                             double* model_us_without) {
  SK_GUARD_BEGIN
  if (!p || !num_retained) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (mode != SK_RETAINED_AUTO && mode != SK_RETAINED_ON && mode != SK_RETAINED_OFF) { set_error("invalid retained-points mode %d", mode); return SK_ERR_INVALID_ARGUMENT; }
  if (border_mode != SK_BORDER_AUTO && border_mode != SK_BORDER_ON && border_mode != SK_BORDER_OFF) { set_error("invalid border mode %d", border_mode); return SK_ERR_INVALID_ARGUMENT; }
  std::string why;
  if (!problem_is_bal_shaped(p->p, &why)) { set_error("%s", why.c_str()); return SK_ERR_UNSUPPORTED; }
  std::vector<int> flag;
  *num_retained = bal_retained_plan(p->p, mode, max_points, border_mode, &flag, model_us, model_us_without);
  if (retained_of_block) for (size_t b = 0; b < flag.size(); ++b) retained_of_block[b] = flag[b];
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_synth_dense_targets(double seed, int m, int n, const double* x_star, double* y_out) {
  SK_GUARD_BEGIN
  if (m <= 0 || n <= 0 || !x_star || !y_out) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  std::vector<double> consts(3 * (size_t)m);
  for (int i = 0; i < m; ++i) { consts[3 * (size_t)i] = seed; consts[3 * (size_t)i + 1] = (double)i; consts[3 * (size_t)i + 2] = 0.0; }
  std::vector<double> xs(x_star, x_star + n);
  DevBuf<double> dc, dx, dr;
  hipStream_t s = nullptr;
  SK_HIP_TRY(dc.upload(consts, s)); SK_HIP_TRY(dx.upload(xs, s)); SK_HIP_TRY(dr.alloc(m));
  DenseRowsArgs a{}; a.loss_root = -1; a.m = m; a.n = n; a.m_pad = m; a.consts = dc.p; a.inv_sqrt_n = 1.0 / std::sqrt((double)n);
  launch_rows_residual(a, dx.p, dr.p, nullptr, false, s);
  SK_HIP_TRY(hipMemcpy(y_out, dr.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_cholesky_solve(int n, const double* A, const double* b, double* x, double* L, int group) {
  return sk_cholesky_solve_ex(n, A, b, x, L, group, nullptr, 0);
}

static int cholesky_solve_impl(int n, const double* A, const double* b, double* x, double* L, int group, const int* last, int automatic_plan, int border_begin_row);
int sk_cholesky_solve_ex(int n, const double* A, const double* b, double* x, double* L, int group, const int* last, int automatic_plan) {
  return cholesky_solve_impl(n, A, b, x, L, group, last, automatic_plan, -1);
}
int sk_cholesky_solve_bordered(int n, const double* A, const double* b, double* x, double* L, int group, int border_begin_row, int automatic_plan) {
  if (border_begin_row < 0 || border_begin_row > n) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  return cholesky_solve_impl(n, A, b, x, L, group, nullptr, automatic_plan, border_begin_row);
}
static std::vector<int> host_block_first_cols(const std::vector<double>& M, size_t ld, int nblk);
// border_begin_row >= 0: the bordered envelope of the matrix itself (cholesky_envelope_bordered), rows from there on being the border
static int cholesky_solve_impl(int n, const double* A, const double* b, double* x, double* L, int group, const int* last, int automatic_plan, int border_begin_row) {
  SK_GUARD_BEGIN
  if (n <= 0 || !A || !b || !x) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  if (group <= 0) group = automatic_plan ? 1 : 3;
  const int rhs_row = n, npad = ((n + 1 + 127) / 128) * 128, nblk = npad / 128;
  if (last) {
    for (int c = 0; c < nblk; ++c) {
      const bool ok = last[c] >= std::min(c, nblk - 2) && last[c] <= nblk - 1 && (c == 0 || last[c] >= last[c - 1]);
      if (!ok) { set_error("invalid block envelope at block column %d", c); return SK_ERR_INVALID_ARGUMENT; }
    }
  }
  std::vector<double> S((size_t)npad * npad, 0.0);
  for (int i = 0; i < n; ++i) std::memcpy(&S[(size_t)i * npad], A + (size_t)i * n, (size_t)(i + 1) * sizeof(double));
  std::memcpy(&S[(size_t)rhs_row * npad], b, (size_t)n * sizeof(double));
  S[(size_t)rhs_row * npad + rhs_row] = 1e300;
  for (int j = n + 1; j < npad; ++j) S[(size_t)j * npad + j] = 1.0;
  std::vector<int> b_last, b_tail;
  const int* tail = nullptr;
  if (border_begin_row >= 0) {
    cholesky_envelope_bordered(host_block_first_cols(S, (size_t)npad, nblk), border_begin_row / 128, &b_last, &b_tail);
    last = b_last.data(); tail = b_tail.data();
  }
  DevBuf<double> dS, dLinv, dy; DevBuf<int> dinfo;
  hipStream_t s = nullptr;
  SK_HIP_TRY(dS.upload(S, s)); SK_HIP_TRY(dLinv.alloc((size_t)npad * 128)); SK_HIP_TRY(dLinv.zero(s));
  DevBuf<double> dw;
  SK_HIP_TRY(dy.alloc(npad)); SK_HIP_TRY(dw.alloc(npad)); SK_HIP_TRY(dinfo.alloc(1)); SK_HIP_TRY(dinfo.zero(s));
  SK_HIP_TRY(cholesky_init());
  CholeskyContext ctx;  // exercise the look-ahead path the solver uses
  const bool la = ctx.init() == hipSuccess;
  if (!la) (void)hipGetLastError();
  SK_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  SK_HIP_TRY(hipDeviceSynchronize());  // uploads above ran on the null stream
  const bool chain = automatic_plan != 0 && la && ctx.server != nullptr;  // (the plan; resident or not is the device's state)
  cholesky_factor(dS.p, npad, npad, dLinv.p, dinfo.p, group, s, la ? &ctx : nullptr, nullptr, last, chain, -1, 1, nullptr, tail);
  cholesky_backsolve(dS.p, npad, n, npad, rhs_row, dLinv.p, dw.p, dy.p, s, nullptr, last, dinfo.p, tail);
  SK_HIP_TRY(hipStreamSynchronize(s));
  SK_HIP_TRY(hipStreamDestroy(s));
  s = nullptr;
  SK_HIP_TRY(hipStreamSynchronize(s));
  int info = 0;
  SK_HIP_TRY(hipMemcpy(&info, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
  if (info == 2) { (void)cholesky_note_info(&ctx, info); set_error("the resident panel chain timed out"); return SK_ERR_HIP; }
  if (info) { set_error("matrix is not positive definite"); return SK_ERR_EVALUATION_FAILED; }
  std::vector<double> y(npad);
  SK_HIP_TRY(hipMemcpy(y.data(), dy.p, npad * sizeof(double), hipMemcpyDeviceToHost));
  std::memcpy(x, y.data(), (size_t)n * sizeof(double));
  if (L) {
    SK_HIP_TRY(hipMemcpy(S.data(), dS.p, S.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) L[(size_t)i * n + j] = j <= i ? S[(size_t)i * npad + j] : 0.0;
  }
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

// Block envelope of a front held on the host (lower triangle, row-major, ld): first non-zero block column per block row.
static std::vector<int> host_block_first_cols(const std::vector<double>& M, size_t ld, int nblk) {
  std::vector<int> first_col(nblk);
  for (int i = 0; i < nblk; ++i) {
    first_col[i] = i;
    for (int j = 0; j < i && first_col[i] == i; ++j) {
      bool nz = false;
      for (int r = 0; r < 128 && !nz; ++r) {
        const double* row = &M[((size_t)i * 128 + r) * ld + (size_t)j * 128];
        for (int c = 0; c < 128; ++c) if (row[c] != 0.0) { nz = true; break; }
      }
      if (nz) first_col[i] = j;
    }
  }
  return first_col;
}
static std::vector<int> host_block_envelope(const std::vector<double>& M, size_t ld, int nblk, int tail_rows = 1) {
  return cholesky_envelope_last(host_block_first_cols(M, ld, nblk), tail_rows);
}

int sk_cholesky_solve_dissected(int n, const double* A, const double* b, double* x, int head, int tail_begin, int group, int automatic_plan) {
  SK_GUARD_BEGIN
  if (n <= 0 || !A || !b || !x || head < 0 || tail_begin < head || tail_begin > n) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  const int ra = head, rb = tail_begin, nt = n - rb, msep = rb - ra;
  for (int i = rb; i < n; ++i)
    for (int j = 0; j < ra; ++j)
      if (A[(size_t)i * n + j] != 0.0) { set_error("the tail couples with the head at (%d, %d): not a separator", i, j); return SK_ERR_INVALID_ARGUMENT; }
  if (group <= 0) group = automatic_plan ? 1 : 3;
  const int nA = (ra + 127) / 128, nB = (nt + 127) / 128, E = (msep + 1 + 127) / 128;
  const size_t dA = (size_t)(nA + E) * 128, dB = (size_t)(nB + E) * 128, dR = (size_t)E * 128;
  std::vector<double> FA(dA * dA, 0.0), FB(dB * dB, 0.0), FR(dR * dR, 0.0);
  // head: interior in order, border = separator in order, then the right-hand side
  for (int i = 0; i < ra; ++i) std::memcpy(&FA[(size_t)i * dA], A + (size_t)i * n, (size_t)(i + 1) * sizeof(double));
  for (int i = ra; i < nA * 128; ++i) FA[(size_t)i * dA + i] = 1.0;
  for (int k = 0; k < msep; ++k) std::memcpy(&FA[((size_t)nA * 128 + k) * dA], A + (size_t)(ra + k) * n, (size_t)ra * sizeof(double));
  std::memcpy(&FA[((size_t)nA * 128 + msep) * dA], b, (size_t)ra * sizeof(double));
  // tail: interior in REVERSE order (t <-> global n - 1 - t), border = separator in reverse order, then the right-hand side
  for (int t1 = 0; t1 < nt; ++t1)
    for (int t2 = 0; t2 <= t1; ++t2) FB[(size_t)t1 * dB + t2] = A[(size_t)(n - 1 - t2) * n + (n - 1 - t1)];
  for (int i = nt; i < nB * 128; ++i) FB[(size_t)i * dB + i] = 1.0;
  for (int k = 0; k < msep; ++k)
    for (int t = 0; t < nt; ++t) FB[((size_t)nB * 128 + k) * dB + t] = A[(size_t)(n - 1 - t) * n + (ra + msep - 1 - k)];
  for (int t = 0; t < nt; ++t) FB[((size_t)nB * 128 + msep) * dB + t] = b[n - 1 - t];
  // root: the separator's own block and right-hand side
  for (int i = 0; i < msep; ++i) std::memcpy(&FR[(size_t)i * dR], A + (size_t)(ra + i) * n + ra, (size_t)(i + 1) * sizeof(double));
  std::memcpy(&FR[(size_t)msep * dR], b + ra, (size_t)msep * sizeof(double));
  FR[(size_t)msep * dR + msep] = 1e300;
  for (size_t j = (size_t)msep + 1; j < dR; ++j) FR[j * dR + j] = 1.0;
  std::vector<int> mapB(dR, -1);
  for (int k = 0; k < msep; ++k) mapB[k] = msep - 1 - k;
  mapB[msep] = msep;
  const std::vector<int> lastA = host_block_envelope(FA, dA, nA + E), lastB = host_block_envelope(FB, dB, nB + E);
  DevBuf<double> dFA, dFB, dFR, dLinv, dw, dy; DevBuf<int> dinfo, dmap;
  hipStream_t s = nullptr;
  SK_HIP_TRY(dFA.upload(FA, s)); SK_HIP_TRY(dFB.upload(FB, s)); SK_HIP_TRY(dFR.upload(FR, s)); SK_HIP_TRY(dmap.upload(mapB, s));
  SK_HIP_TRY(dLinv.alloc((size_t)(nA + nB + E) * 128 * 128)); SK_HIP_TRY(dLinv.zero(s));
  SK_HIP_TRY(dw.alloc(dA + dB + 2 * dR)); SK_HIP_TRY(dy.alloc(dA + dB + dR)); SK_HIP_TRY(dy.zero(s)); SK_HIP_TRY(dinfo.alloc(1)); SK_HIP_TRY(dinfo.zero(s));
  SK_HIP_TRY(cholesky_init());
  CholeskyContext ctx, ctxB;
  const bool la = ctx.init() == hipSuccess;
  if (!la) (void)hipGetLastError();
  SK_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  SK_HIP_TRY(hipDeviceSynchronize());
  if (la) cholesky_prepare(&ctx, s);  // the device's queue choice first: the second context takes the queues it leaves over
  const bool side = la && ctxB.init_secondary(ctx) == hipSuccess;
  if (la && !side) (void)hipGetLastError();
  DissectedSystem d;
  d.A.S = dFA.p; d.A.ld = (long)dA; d.A.nblk = nA + E; d.A.ncols = nA; d.A.last = lastA.data(); d.A.Linv = dLinv.p; d.A.rhs_row = nA * 128 + msep;
  d.B.S = dFB.p; d.B.ld = (long)dB; d.B.nblk = nB + E; d.B.ncols = nB; d.B.last = lastB.data(); d.B.Linv = dLinv.p + (size_t)nA * 128 * 128; d.B.rhs_row = nB * 128 + msep;
  d.R.S = dFR.p; d.R.ld = (long)dR; d.R.nblk = E; d.R.ncols = E; d.R.last = nullptr; d.R.Linv = dLinv.p + (size_t)(nA + nB) * 128 * 128; d.R.rhs_row = msep;
  d.border_blocks = E; d.mapB = dmap.p;
  const bool chain = automatic_plan != 0 && la && ctx.server != nullptr;
  cholesky_dissected_factor(d, dinfo.p, group, s, la ? &ctx : nullptr, side ? &ctxB : nullptr, nullptr, nullptr, chain);
  double *wA = dw.p, *wB = dw.p + dA, *wR = dw.p + dA + dB, *ybB = dw.p + dA + dB + dR;
  double *yA = dy.p, *yB = dy.p + dA, *yR = dy.p + dA + dB;
  cholesky_dissected_backsolve(d, msep, wR, yR, wA, yA, wB, yB, ybB, s, side ? &ctxB : nullptr, nullptr, dinfo.p);
  SK_HIP_TRY(hipStreamSynchronize(s));
  SK_HIP_TRY(hipStreamDestroy(s));
  int info = 0;
  SK_HIP_TRY(hipMemcpy(&info, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
  if (info == 2) { (void)cholesky_note_info(&ctx, info); set_error("the resident panel chain timed out"); return SK_ERR_HIP; }
  if (info) { set_error("matrix is not positive definite"); return SK_ERR_EVALUATION_FAILED; }
  std::vector<double> y(dA + dB + dR);
  SK_HIP_TRY(hipMemcpy(y.data(), dy.p, y.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int i = 0; i < ra; ++i) x[i] = y[i];
  for (int t = 0; t < nt; ++t) x[n - 1 - t] = y[dA + t];
  for (int k = 0; k < msep; ++k) x[ra + k] = y[dA + dB + k];
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

int sk_cholesky_solve_segments(int n, const double* A, const double* b, double* x, int num_segments, const int* cuts, int group, int automatic_plan) {
  SK_GUARD_BEGIN
  const int R = num_segments;
  if (n <= 0 || !A || !b || !x || R < 2 || !cuts) { set_error("invalid argument"); return SK_ERR_INVALID_ARGUMENT; }
  if (sk_device_count() <= 0) { set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  // separator k (1 <= k < R) = rows [sb[k], se[k]); segment k = rows [se[k], sb[k + 1]) with se[0] = 0, sb[R] = n
  std::vector<int> sb(R + 1, 0), se(R + 1, 0);
  sb[R] = n;
  for (int k = 1; k < R; ++k) { sb[k] = cuts[2 * (k - 1)]; se[k] = cuts[2 * (k - 1) + 1]; }
  for (int k = 1; k <= R; ++k)
    if (sb[k] < se[k - 1] || (k < R && (se[k] < sb[k] || se[k] > n))) { set_error("invalid cuts"); return SK_ERR_INVALID_ARGUMENT; }
  for (int k = 1; k < R; ++k)
    for (int i = se[k]; i < n; ++i)
      for (int j = 0; j < sb[k]; ++j)
        if (A[(size_t)i * n + j] != 0.0) { set_error("rows behind separator %d couple with rows before it at (%d, %d): not a separator", k, i, j); return SK_ERR_INVALID_ARGUMENT; }
  if (group <= 0) group = automatic_plan ? 1 : 3;
  auto a = [&](int i, int j) { return i >= j ? A[(size_t)i * n + j] : A[(size_t)j * n + i]; };  // (lower triangle given)
  // root: every separator in order, then the right-hand side
  std::vector<int> sep_off(R, 0);
  for (int k = 1; k < R; ++k) sep_off[k] = sep_off[k - 1] + (se[k] - sb[k]);  // sep_off[k - 1]: offset of separator k; sep_off[R - 1]: total
  const int nroot = sep_off[R - 1], E = (nroot + 1 + 127) / 128;
  const size_t dR = (size_t)E * 128;
  auto root_index = [&](int row) {  // global row of a separator -> root index
    for (int k = 1; k < R; ++k) if (row >= sb[k] && row < se[k]) return sep_off[k - 1] + (row - sb[k]);
    return -1;
  };
  std::vector<double> FR(dR * dR, 0.0);
  for (int k = 1; k < R; ++k)
    for (int i = sb[k]; i < se[k]; ++i) {
      const int ri = root_index(i);
      for (int k2 = 1; k2 <= k; ++k2)
        for (int j = sb[k2]; j < se[k2] && j <= i; ++j) FR[(size_t)ri * dR + root_index(j)] = a(i, j);
      FR[(size_t)nroot * dR + ri] = b[i];
    }
  FR[(size_t)nroot * dR + nroot] = 1e300;
  for (size_t j = (size_t)nroot + 1; j < dR; ++j) FR[j * dR + j] = 1.0;
  const std::vector<int> lastR = root_envelope(std::vector<int>(sep_off.begin(), sep_off.end()));
  SK_HIP_TRY(cholesky_init());
  CholeskyContext ctx;
  const bool la = ctx.init() == hipSuccess;
  if (!la) (void)hipGetLastError();
  hipStream_t s = nullptr;
  SK_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (la) cholesky_prepare(&ctx, s);
  const bool chain = automatic_plan != 0 && la && ctx.server != nullptr;
  DevBuf<double> dFR; DevBuf<int> dinfo;
  SK_HIP_TRY(dFR.upload(FR, s)); SK_HIP_TRY(dinfo.alloc(1)); SK_HIP_TRY(dinfo.zero(s));
  struct Leaf { SegmentLayout L; std::vector<int> order, map, last; DevBuf<double> F, Linv, w, y, yb; DevBuf<int> dmap; size_t dim; };
  std::vector<std::unique_ptr<Leaf>> leaves;
  for (int k = 0; k < R; ++k) {
    std::unique_ptr<Leaf> lf(new Leaf());
    const int lo = se[k], hi = sb[k + 1], ni = hi - lo;
    const int left_n = k > 0 ? se[k] - sb[k] : 0, right_n = k + 1 < R ? se[k + 1] - sb[k + 1] : 0;
    lf->L = segment_layout(ni, left_n, right_n);
    const SegmentLayout& L = lf->L;
    lf->dim = (size_t)L.nblk * 128;
    const size_t d = lf->dim;
    // front row -> global row (-1: padding / right-hand side)
    lf->order.assign(d, -1);
    for (int t = 0; t < ni; ++t) lf->order[t] = L.reversed ? hi - 1 - t : lo + t;
    const int bo = L.ncols * 128;
    if (L.right_off >= 0) for (int t = 0; t < right_n; ++t) lf->order[bo + L.right_off + t] = sb[k + 1] + t;
    if (L.left_off >= 0) for (int t = 0; t < left_n; ++t) lf->order[bo + L.left_off + t] = L.reversed ? se[k] - 1 - t : sb[k] + t;
    std::vector<double> F(d * d, 0.0);
    for (size_t i = 0; i < d; ++i) {
      const int gi = lf->order[i];
      if (gi < 0) continue;
      const bool border_i = (int)i >= bo;
      for (size_t j = 0; j <= i && j < (size_t)ni; ++j) F[i * d + j] = a(gi, lf->order[j]);  // interior columns only: the border x border block starts at zero
      (void)border_i;
    }
    for (int t = 0; t < ni; ++t) F[(size_t)L.rhs_row * d + t] = b[lf->order[t]];
    for (int i = ni; i < bo; ++i) F[(size_t)i * d + i] = 1.0;
    lf->last = host_block_envelope(F, d, L.nblk, L.tail_rows);
    lf->map.assign(d - bo, -1);
    for (size_t i = bo; i < d; ++i) if (lf->order[i] >= 0) lf->map[i - bo] = root_index(lf->order[i]);
    lf->map[L.rhs_row - bo] = nroot;
    SK_HIP_TRY(lf->F.upload(F, s)); SK_HIP_TRY(lf->dmap.upload(lf->map, s));
    SK_HIP_TRY(lf->Linv.alloc((size_t)L.ncols * 128 * 128)); SK_HIP_TRY(lf->Linv.zero(s));
    SK_HIP_TRY(lf->w.alloc(d)); SK_HIP_TRY(lf->y.alloc(d)); SK_HIP_TRY(lf->y.zero(s)); SK_HIP_TRY(lf->yb.alloc(d - bo));
    if (ni > 0) {
      cholesky_factor(lf->F.p, (long)d, (int)d, lf->Linv.p, dinfo.p, group, s, la ? &ctx : nullptr, nullptr, lf->last.data(), chain, L.ncols, L.tail_rows);
      cholesky_border_add(dFR.p, (long)dR, lf->F.p, (long)d, L.ncols, L.nblk - L.ncols, lf->dmap.p, s);
    }
    SK_HIP_TRY(hipStreamSynchronize(s));  // (F, the host copy, goes out of scope)
    leaves.push_back(std::move(lf));
  }
  DevBuf<double> dLinvR, dwR, dyR;
  SK_HIP_TRY(dLinvR.alloc(dR * 128)); SK_HIP_TRY(dLinvR.zero(s)); SK_HIP_TRY(dwR.alloc(dR)); SK_HIP_TRY(dyR.alloc(dR));
  cholesky_factor(dFR.p, (long)dR, (int)dR, dLinvR.p, dinfo.p, group, s, la ? &ctx : nullptr, nullptr, lastR.empty() ? nullptr : lastR.data(), chain);
  cholesky_backsolve(dFR.p, (long)dR, nroot, (int)dR, nroot, dLinvR.p, dwR.p, dyR.p, s, nullptr, lastR.empty() ? nullptr : lastR.data(), dinfo.p);
  std::vector<double> y(dR);
  for (auto& lf : leaves) {
    const SegmentLayout& L = lf->L;
    if (L.ncols == 0) continue;
    const int m = (L.nblk - L.ncols) * 128;
    // border unknowns in the leaf's border order (zero in padding rows and in the right-hand-side row)
    std::vector<int> gmap(lf->map);
    gmap[L.rhs_row - L.ncols * 128] = -1;
    DevBuf<int> dg;
    SK_HIP_TRY(dg.upload(gmap, s));
    cholesky_gather_map(dyR.p, dg.p, lf->yb.p, m, s);
    cholesky_backsolve_front(lf->F.p, (long)lf->dim, L.nblk, L.ncols, L.rhs_row, lf->Linv.p, lf->yb.p, lf->w.p, lf->y.p, s, lf->last.data(), L.spike, L.tail_rows, dinfo.p);
    SK_HIP_TRY(hipStreamSynchronize(s));
  }
  SK_HIP_TRY(hipStreamSynchronize(s));
  int info = 0;
  SK_HIP_TRY(hipMemcpy(&info, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
  if (info == 2) { (void)cholesky_note_info(&ctx, info); (void)hipStreamDestroy(s); set_error("the resident panel chain timed out"); return SK_ERR_HIP; }
  if (info) { (void)hipStreamDestroy(s); set_error("matrix is not positive definite"); return SK_ERR_EVALUATION_FAILED; }
  SK_HIP_TRY(hipMemcpy(y.data(), dyR.p, dR * sizeof(double), hipMemcpyDeviceToHost));
  for (int k = 1; k < R; ++k) for (int i = sb[k]; i < se[k]; ++i) x[i] = y[root_index(i)];
  for (auto& lf : leaves) {
    std::vector<double> yl(lf->dim);
    SK_HIP_TRY(hipMemcpy(yl.data(), lf->y.p, lf->dim * sizeof(double), hipMemcpyDeviceToHost));
    for (int t = 0; t < lf->L.ncols * 128; ++t) if (lf->order[t] >= 0) x[lf->order[t]] = yl[t];
  }
  SK_HIP_TRY(hipStreamDestroy(s));
  return SK_OK;
  SK_GUARD_END(SK_ERR_INVALID_ARGUMENT)
}

}  // extern "C"
