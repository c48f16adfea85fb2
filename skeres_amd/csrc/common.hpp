// Shared host-side helpers: error reporting across the C ABI, device buffers,
// and the host mirror of Problem / CostFunction / Options / Summary.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/skeres_amd.h"
#include "dev_knobs.hpp"
#include "functors.hpp"
#include "loss.hpp"
#include "parameterization.hpp"
#include "tape.hpp"

namespace sk {

void set_error(const char* fmt, ...);
const char* get_error();
void set_status(int status);  // status of the last failure that could only be reported as a null handle (sk_last_status)
int get_status();

#define SK_HIP_TRY(expr)                                                                 \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      sk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SK_ERR_HIP;                                                                 \
    }                                                                                    \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owned = true;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr; n = 0; owned = true;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
  }
  void adopt(T* ext, size_t count) { release(); p = ext; n = count; owned = false; }
  hipError_t upload(const std::vector<T>& h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  }
  hipError_t zero(hipStream_t s) { return n ? hipMemsetAsync(p, 0, n * sizeof(T), s) : hipSuccess; }
};

// ---- host mirror of the reference objects ----------------------------------
struct CostFunction {  // com.google.ceres.CostFunction as sized by CORE/SizedCostFunction.scala
  int functor_id = 0;
  std::vector<double> consts;
  sk_evaluate_fn callback = nullptr;
  void* user = nullptr;
  int num_residuals = 0;
  std::vector<int> block_sizes;
  std::shared_ptr<const Tape> tape;  // SK_FUNCTOR_TAPE: the recorded body (consts: the captured doubles of THIS cost function)
};

// PredefinedLossFunctions (ceres.i:159-184): a flattened expression, children before parents; empty == trivial
struct LossFunction {
  std::vector<LossNode> nodes;
  int root() const { return nodes.empty() ? -1 : (int)nodes.size() - 1; }
};

// PredefinedLocalParameterizations (ceres.i:186-210)
struct LocalParameterization {
  int type = kParamIdentity, global_size = 0, local_size = 0;
  unsigned constant_mask = 0;  // subset: bit i = coordinate i held constant
};

struct Problem {  // CeresProblem; parameter blocks identified by pointer value
  std::unordered_map<double*, int> block_of;
  std::vector<double*> block_ptr;
  std::vector<int> block_size;
  // residual blocks (compact form)
  std::vector<int> rb_functor;
  std::vector<int> rb_num_residuals;
  std::vector<size_t> rb_const_off;
  std::vector<size_t> rb_pidx_off;  // size = blocks + 1
  std::vector<int> rb_pidx;
  std::vector<double> consts;
  std::vector<const CostFunction*> rb_cost;  // non-null only for host-callback blocks
  // recorded functor bodies, every distinct one once (by content: the caller may free its cost function); a tape block's
  // rb_functor is kTapeFunctorBase + its index here
  std::vector<std::shared_ptr<const Tape>> tapes;
  std::unordered_map<std::string, int> tape_of;
  int intern_tape(const std::shared_ptr<const Tape>& t) {
    const std::string k = t->key();
    auto it = tape_of.find(k);
    if (it != tape_of.end()) return it->second;
    tapes.push_back(t);
    tape_of.emplace(k, (int)tapes.size() - 1);
    return (int)tapes.size() - 1;
  }
  const Tape* tape_of_block(size_t b) const { const int f = rb_functor[b]; return f >= kTapeFunctorBase ? tapes[f - kTapeFunctorBase].get() : nullptr; }
  // loss functions: every distinct loss expression once in loss_nodes (copied: the caller may free its object);
  // rb_loss[b] = root node of block b's loss, -1 = trivial
  std::vector<LossNode> loss_nodes;
  std::vector<int> rb_loss;
  std::unordered_map<std::string, int> loss_root_of;
  bool has_loss = false;
  int intern_loss(const LossFunction* l);
  // local parameterizations: block_param[b] = index into params (copied: the caller may free its object), -1 = none;
  // block_constant[b]: Problem::SetParameterBlockConstant
  std::vector<LocalParameterization> params;
  std::vector<int> block_param;
  std::vector<char> block_constant;
  bool has_parameterization() const {
    for (size_t b = 0; b < block_ptr.size(); ++b)
      if ((b < block_param.size() && block_param[b] >= 0) || (b < block_constant.size() && block_constant[b])) return true;
    return false;
  }
  long num_residuals = 0;
  bool has_callbacks = false;
  Problem() { rb_pidx_off.push_back(0); }
  int num_parameters() const { long s = 0; for (int b : block_size) s += b; return (int)s; }
};

struct Options {  // Solver.Options; Ceres 1.x defaults
  int linear_solver_type = SK_DENSE_QR;  // ceres default is SPARSE_NORMAL_CHOLESKY when built with a sparse backend, else DENSE_QR
  int linear_solver_type_given = -1;     // set when the solver in use is an alternate for the one asked for (capi.hip: make_solver)
  int minimizer_type = SK_TRUST_REGION;
  int max_num_iterations = 50;
  bool progress_to_stdout = false;
  double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
  double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  bool jacobi_scaling = true;
  int max_num_consecutive_invalid_steps = 5;
  int device = -1;
  hipStream_t stream = nullptr;
  bool stream_set = false;
  int rank = 0, world = 1;
  sk_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int distribution_mode = SK_DISTRIBUTION_AUTO;
  void* reduce_buffer = nullptr;
  size_t reduce_buffer_bytes = 0;
  // SYRK depth in 128-column blocks (K = group * 128).  0 = automatic: 3 when the trailing SYRK is the long pole
  // (full factorisation; 2 and 4 within 1 %), 1 when the block envelope leaves so little of it that the serial
  // panel chain decides (measured on Ladybug-1723: 13.2 / 13.4 / 14.0 / 14.6 ms per iteration for 1 / 2 / 3 / 4).
  int cholesky_group = 0;
  int group_or(int automatic) const { return cholesky_group > 0 ? cholesky_group : automatic; }
  bool lookahead = true;   // potrf128 on a second stream, off the critical path
  int dissection = SK_DISSECTION_AUTO;  // DENSE_SCHUR: eliminate the head and the tail of a camera sequence side by side (sk_options_set_cholesky_dissection)
  bool envelope = true;    // DENSE_SCHUR: skip the blocks of the reduced system outside its block envelope (bit-identical result)
  bool resident_kernels = true;  // sk_options_set_resident_kernels: 0 = no kernel of this solver waits for another one (same plans, launch by launch)
  bool graph_replay = true;      // sk_options_set_graph_replay: launch-bound problems replay their iteration as a hipGraph
  int max_segments = 0;          // sk_options_set_max_segments: SEGMENTED / AUTO cut the camera sequence into at most this many segments (0: one per rank)
  int border = SK_BORDER_AUTO;  // DENSE_SCHUR: order the cameras of loop closures into a trailing border of the reduced system (sk_options_set_cholesky_border)
  int retained = SK_RETAINED_AUTO, retained_max = 0;  // DENSE_SCHUR: the points with the widest tracks stay in the reduced system (sk_options_set_retained_points)
};

struct IterationLog {
  int iteration = 0;
  double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, relative_decrease = 0,
         trust_region_radius = 0, iter_time = 0, total_time = 0;
  int step_is_valid = 1, step_is_successful = 1;
};

struct Summary {
  double initial_cost = 0, final_cost = 0;
  int num_successful_steps = 0, num_unsuccessful_steps = 0;
  int termination_type = SK_NO_CONVERGENCE;
  std::string message;
  std::vector<IterationLog> iterations;
  double phase_seconds[7] = {0, 0, 0, 0, 0, 0, 0};
  // problem / solver description for the reports
  int num_parameter_blocks = 0, num_parameters = 0, num_residual_blocks = 0;
  long num_residuals = 0;
  int linear_solver_type = 0;
  int linear_solver_type_given = -1;  // what Solver.Options asked for when the solver used is its alternate (Ceres: "Given / Used")
  int num_e_blocks = 0, num_f_blocks = 0;
  int world = 1;
  std::string device_name;
  std::string brief, full;
  void build_reports();
};

// device copy of a tape (owned by a solver, or by one sk_cost_function_evaluate)
struct TapeDevBuffers {
  DevBuf<TapeIns> ins; DevBuf<double> consts; DevBuf<int32_t> out; DevBuf<int> param_block, param_index;
  TapeDev view;
  Tape host;
  hipError_t upload(const Tape& t, hipStream_t s);
};

const char* linear_solver_name(int t);
const char* termination_name(int t);

}  // namespace sk
