// Generic dense-Jacobian path: DENSE_QR and DENSE_NORMAL_CHOLESKY over any mix
// of registered device functors and host-callback cost functions
// (EX/CurveFitting.scala:100-133, EX/Powell.scala:55-91).
//
// Host-callback blocks are the reference's director path (ceres.i:48): the
// caller's Evaluate runs on the host at the current x and its rows are
// uploaded into the device Jacobian; everything after evaluation (scaling,
// normal equations / QR, Cholesky, trust-region arithmetic) runs on the GPU.
#include <algorithm>
#include <cmath>
#include <map>

#include "bal_kernels.hpp"
#include "dense_kernels.hpp"
#include "solver.hpp"

namespace sk {
namespace {

class DenseSolver : public SolverBase {
 public:
  DenseSolver(const Options& o, Problem* p) : SolverBase(o, p) {}
  ~DenseSolver() override { if (h_scal_) (void)hipHostFree(h_scal_); }

 protected:
  int setup() override;
  int evaluate_with_jacobian(bool first) override;
  int try_step(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm) override;
  void accept_candidate() override { std::swap(x_, x_new_); }
  int write_back() override;
  void describe(Summary* s) override {
    s->num_parameter_blocks = (int)problem_->block_size.size();
    s->num_parameters = ng_; s->num_residual_blocks = (int)problem_->rb_functor.size(); s->num_residuals = m_;
  }

 private:
  int evaluate(const double* x_dev, bool jac);
  int host_callbacks(const double* x_dev, bool jac);

  // n_: columns of the Jacobian the minimiser works with = size of the tangent space; ng_: size of x.  Equal unless
  // the problem has local parameterizations or constant blocks (tangent_): then the functors fill the global Jacobian
  // b_Jg_ unscaled, and dense_project_kernel forms b_J_ = b_Jg_ blockdiag(dPlus/ddelta) diag(scale).
  int n_ = 0, ng_ = 0, m_ = 0, npad_ = 0, rhs_row_ = 0;
  bool tangent_ = false;
  DevBuf<double> b_Jg_, b_ones_;
  DevBuf<ParamBlock> b_pblocks_;
  int num_pblocks_ = 0;
  std::vector<int> block_off_;
  std::map<int, std::vector<int>> by_functor_;  // functor id -> residual block ids
  std::map<int, DevBuf<int>> by_functor_dev_;
  std::map<int, TapeDevBuffers> tapes_dev_;  // recorded functors (keys >= kTapeFunctorBase of by_functor_)
  std::vector<int> cb_blocks_;
  std::vector<int> res_off_h_;
  DevBuf<double> b_consts_, b_xa_, b_xb_, b_scale_, b_colsq_, b_gs_, b_D_, b_step_, b_y_, b_r_, b_rc_, b_J_, b_H_, b_Linv_, b_A_, b_b_, b_scal_;
  DevBuf<double> b_w_;
  DevBuf<size_t> b_const_off_, b_pidx_off_;
  DevBuf<int> b_pidx_, b_res_off_, b_fail_, b_info_, b_ok_;
  // robust losses (only when the problem has any)
  bool has_loss_ = false;
  DevBuf<int> b_psize_, b_rb_loss_;
  DevBuf<LossNode> b_loss_nodes_;
  DevBuf<double> b_cterm_;
  void apply_loss(double* r, bool jac);
  double* x_ = nullptr; double* x_new_ = nullptr;
  double* h_scal_ = nullptr;
};

int DenseSolver::setup() {
  const Problem& p = *problem_;
  if (opt_.world > 1) { set_error("the dense path does not shard: run replicas (world must be 1)"); return SK_ERR_UNSUPPORTED; }
  const int nb = (int)p.rb_functor.size();
  if (nb == 0) { set_error("problem has no residual blocks"); return SK_ERR_INVALID_ARGUMENT; }
  block_off_.resize(p.block_size.size());
  ng_ = 0;
  for (size_t b = 0; b < p.block_size.size(); ++b) { block_off_[b] = ng_; ng_ += p.block_size[b]; }
  n_ = ng_;
  tangent_ = p.has_parameterization();
  std::vector<ParamBlock> pblocks;
  if (tangent_) {
    n_ = 0;
    for (size_t b = 0; b < p.block_size.size(); ++b) {
      ParamBlock pb; pb.type = kParamIdentity; pb.global_size = pb.local_size = p.block_size[b]; pb.constant_mask = 0u;
      if (b < p.block_param.size() && p.block_param[b] >= 0) {
        const LocalParameterization& lp = p.params[p.block_param[b]];
        pb.type = lp.type; pb.local_size = lp.local_size; pb.constant_mask = lp.constant_mask;
      }
      if (b < p.block_constant.size() && p.block_constant[b]) { pb.type = kParamConstant; pb.local_size = 0; }
      if (pb.type != kParamIdentity && pb.global_size > kParamMaxSize) { set_error("parameterized block of size %d (max %d)", pb.global_size, kParamMaxSize); return SK_ERR_UNSUPPORTED; }
      if (pb.type == kParamIdentity && pb.global_size > kParamMaxSize) {  // a plain large block next to parameterized ones: as pieces
        for (int o = 0; o < p.block_size[b]; o += kParamMaxSize) {
          ParamBlock piece = pb; piece.global_size = piece.local_size = std::min(kParamMaxSize, p.block_size[b] - o);
          piece.global_off = block_off_[b] + o; piece.local_off = n_; n_ += piece.local_size; pblocks.push_back(piece);
        }
        continue;
      }
      pb.global_off = block_off_[b]; pb.local_off = n_; n_ += pb.local_size;
      pblocks.push_back(pb);
    }
    num_pblocks_ = (int)pblocks.size();
    if (n_ == 0) { set_error("every parameter block is constant: nothing to optimise"); return SK_ERR_INVALID_ARGUMENT; }
  }
  m_ = (int)p.num_residuals;
  if ((double)m_ * n_ > 2e9) { set_error("dense Jacobian of %d x %d is too large for this build", m_, n_); return SK_ERR_UNSUPPORTED; }
  res_off_h_.resize(nb + 1);
  std::vector<int> pidx(p.rb_pidx.size());
  int row = 0;
  for (int b = 0; b < nb; ++b) {
    res_off_h_[b] = row; row += p.rb_num_residuals[b];
    if (p.rb_functor[b] == SK_FUNCTOR_HOST_CALLBACK) cb_blocks_.push_back(b); else by_functor_[p.rb_functor[b]].push_back(b);
  }
  res_off_h_[nb] = row;
  for (size_t i = 0; i < pidx.size(); ++i) pidx[i] = block_off_[p.rb_pidx[i]];
  hipStream_t s = stream_;
  has_loss_ = p.has_loss;
  if (has_loss_) {
    std::vector<int> psize(p.rb_pidx.size());
    for (size_t i = 0; i < psize.size(); ++i) psize[i] = p.block_size[p.rb_pidx[i]];
    SK_HIP_TRY(b_psize_.upload(psize, s)); SK_HIP_TRY(b_rb_loss_.upload(p.rb_loss, s)); SK_HIP_TRY(b_loss_nodes_.upload(p.loss_nodes, s));
    SK_HIP_TRY(b_cterm_.alloc(m_));
  }
  std::vector<double> consts = p.consts; if (consts.empty()) consts.push_back(0.0);
  SK_HIP_TRY(b_consts_.upload(consts, s)); SK_HIP_TRY(b_const_off_.upload(p.rb_const_off, s));
  SK_HIP_TRY(b_pidx_off_.upload(p.rb_pidx_off, s)); SK_HIP_TRY(b_pidx_.upload(pidx, s)); SK_HIP_TRY(b_res_off_.upload(res_off_h_, s));
  for (auto& kv : by_functor_) {
    SK_HIP_TRY(by_functor_dev_[kv.first].upload(kv.second, s));
    if (kv.first >= kTapeFunctorBase) {  // a recorded functor: its tape goes to the device once
      const Tape& t = *p.tapes[kv.first - kTapeFunctorBase];
      if (tape_pick_width(t, 128) == 0) { set_error("a recorded functor needs %d registers: more than the device interpreter holds", t.num_registers); return SK_ERR_UNSUPPORTED; }
      SK_HIP_TRY(tapes_dev_[kv.first].upload(t, s));
    }
  }
  std::vector<double> x(ng_);
  for (size_t b = 0; b < p.block_size.size(); ++b) std::memcpy(&x[block_off_[b]], p.block_ptr[b], p.block_size[b] * sizeof(double));
  SK_HIP_TRY(b_xa_.upload(x, s)); SK_HIP_TRY(b_xb_.alloc(ng_));
  if (tangent_) {
    SK_HIP_TRY(b_pblocks_.upload(pblocks, s));
    SK_HIP_TRY(b_ones_.upload(std::vector<double>(ng_, 1.0), s));
    SK_HIP_TRY(b_Jg_.alloc((size_t)m_ * ng_)); SK_HIP_TRY(b_Jg_.zero(s));
  }
  x_ = b_xa_.p; x_new_ = b_xb_.p;
  SK_HIP_TRY(b_scale_.alloc(n_)); SK_HIP_TRY(b_colsq_.alloc(n_)); SK_HIP_TRY(b_gs_.alloc(n_)); SK_HIP_TRY(b_D_.alloc(n_)); SK_HIP_TRY(b_step_.alloc(n_));
  { std::vector<double> ones(n_, 1.0); SK_HIP_TRY(hipMemcpyAsync(b_scale_.p, ones.data(), n_ * sizeof(double), hipMemcpyHostToDevice, s)); SK_HIP_TRY(hipStreamSynchronize(s)); }
  SK_HIP_TRY(b_r_.alloc(m_)); SK_HIP_TRY(b_rc_.alloc(m_)); SK_HIP_TRY(b_J_.alloc((size_t)m_ * n_)); SK_HIP_TRY(b_J_.zero(s));
  rhs_row_ = n_; npad_ = ((n_ + 1 + 127) / 128) * 128;
  if (opt_.linear_solver_type == SK_DENSE_NORMAL_CHOLESKY) {
    SK_HIP_TRY(b_H_.alloc((size_t)npad_ * npad_)); SK_HIP_TRY(b_Linv_.alloc((size_t)npad_ * 128)); SK_HIP_TRY(b_Linv_.zero(s));
    SK_HIP_TRY(cholesky_init());
  } else {
    SK_HIP_TRY(b_A_.alloc((size_t)(m_ + n_) * n_)); SK_HIP_TRY(b_b_.alloc(m_ + n_));
  }
  SK_HIP_TRY(b_y_.alloc(npad_)); SK_HIP_TRY(b_w_.alloc(npad_)); SK_HIP_TRY(b_scal_.alloc(16));
  SK_HIP_TRY(b_fail_.alloc(1)); SK_HIP_TRY(b_fail_.zero(s)); SK_HIP_TRY(b_info_.alloc(1)); SK_HIP_TRY(b_info_.zero(s)); SK_HIP_TRY(b_ok_.alloc(1));
  SK_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_scal_), 64 * sizeof(double), hipHostMallocDefault));
  SK_HIP_TRY(hipStreamSynchronize(s));
  return SK_OK;
}

// Director path: run the caller's Evaluate for every host-callback block at x.
int DenseSolver::host_callbacks(const double* x_dev, bool jac) {
  if (cb_blocks_.empty()) return SK_OK;
  const Problem& p = *problem_;
  std::vector<double> x(ng_), scale(ng_, 1.0);
  SK_HIP_TRY(hipMemcpyAsync(x.data(), x_dev, ng_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
  if (!tangent_) SK_HIP_TRY(hipMemcpyAsync(scale.data(), b_scale_.p, n_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
  double* Jrows = tangent_ ? b_Jg_.p : b_J_.p;
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  double* r_dev = jac ? b_r_.p : b_rc_.p;
  for (int b : cb_blocks_) {
    const CostFunction* cf = p.rb_cost[b];
    const int nblk = (int)cf->block_sizes.size(), nres = cf->num_residuals;
    std::vector<const double*> params(nblk);
    std::vector<std::vector<double>> jbuf(nblk);
    std::vector<double*> jptr(nblk);
    for (int q = 0; q < nblk; ++q) {
      params[q] = &x[block_off_[p.rb_pidx[p.rb_pidx_off[b] + q]]];
      jbuf[q].assign((size_t)nres * cf->block_sizes[q], 0.0);
      jptr[q] = jbuf[q].data();
    }
    std::vector<double> res(nres, 0.0);
    if (!cf->callback(cf->user, params.data(), res.data(), jac ? jptr.data() : nullptr)) return SK_ERR_EVALUATION_FAILED;
    SK_HIP_TRY(hipMemcpyAsync(r_dev + res_off_h_[b], res.data(), nres * sizeof(double), hipMemcpyHostToDevice, stream_));
    if (jac) {
      std::vector<double> rowbuf;
      for (int q = 0; q < nblk; ++q) {
        const int off = block_off_[p.rb_pidx[p.rb_pidx_off[b] + q]], nq = cf->block_sizes[q];
        for (int r = 0; r < nres; ++r) {
          rowbuf.resize(nq);
          for (int j = 0; j < nq; ++j) rowbuf[j] = jbuf[q][(size_t)r * nq + j] * scale[off + j];
          // (on the solver's stream, never the null stream: a null-stream copy would wait for every blocking stream of the device —
          // among them another solver's resident potrf server — and hold back whatever is enqueued behind it)
          SK_HIP_TRY(hipMemcpyAsync(Jrows + (size_t)(res_off_h_[b] + r) * ng_ + off, rowbuf.data(), nq * sizeof(double), hipMemcpyHostToDevice, stream_));
          SK_HIP_TRY(hipStreamSynchronize(stream_));  // (rowbuf is reused)
        }
      }
    }
    SK_HIP_TRY(hipStreamSynchronize(stream_));
  }
  return SK_OK;
}

// loss correction of the freshly evaluated rows (and Jacobian) + the per-row cost terms
void DenseSolver::apply_loss(double* r, bool jac) {
  DenseLossArgs a;
  a.num_blocks = (int)problem_->rb_functor.size(); a.res_off = b_res_off_.p; a.rb_loss = b_rb_loss_.p; a.nodes = b_loss_nodes_.p;
  a.pidx = b_pidx_.p; a.psize = b_psize_.p; a.pidx_off = b_pidx_off_.p; a.r = r; a.J = jac ? (tangent_ ? b_Jg_.p : b_J_.p) : nullptr; a.cterm = b_cterm_.p;
  a.n = ng_;
  launch_dense_loss(a, stream_);
}

int DenseSolver::evaluate(const double* x_dev, bool jac) {
  DenseEvalArgs a;
  a.consts = b_consts_.p; a.const_off = b_const_off_.p; a.pidx = b_pidx_.p; a.pidx_off = b_pidx_off_.p; a.res_off = b_res_off_.p;
  a.x = x_dev; a.scale = tangent_ ? b_ones_.p : b_scale_.p; a.r = jac ? b_r_.p : b_rc_.p; a.J = tangent_ ? b_Jg_.p : b_J_.p; a.n = ng_; a.fail_flag = b_fail_.p;
  for (auto& kv : by_functor_) {
    a.count = (int)kv.second.size(); a.blocks = by_functor_dev_[kv.first].p;
    if (kv.first >= kTapeFunctorBase) launch_dense_eval_tape(tapes_dev_[kv.first], jac, a, stream_);
    else launch_dense_eval(kv.first, jac, a, stream_);
  }
  return host_callbacks(x_dev, jac);
}

int DenseSolver::evaluate_with_jacobian(bool first) {
  hipStream_t s = stream_;
  SK_HIP_TRY(hipEventRecord(ev_[kEvBegin], s));
  SK_HIP_TRY(hipMemsetAsync(b_fail_.p, 0, sizeof(int), s));
  int rc = evaluate(x_, true);
  if (rc) return rc;
  if (has_loss_) apply_loss(b_r_.p, true);  // before the column norms: the Jacobi scaling is that of the corrected Jacobian
  if (tangent_) launch_dense_project(b_Jg_.p, m_, ng_, b_pblocks_.p, num_pblocks_, x_, b_scale_.p, b_J_.p, n_, s);
  launch_dense_col_reduce(b_J_.p, b_r_.p, m_, n_, b_colsq_.p, b_gs_.p, s);
  if (first && opt_.jacobi_scaling) {
    launch_jacobi_scale(b_colsq_.p, b_scale_.p, n_, s);
    launch_dense_scale(b_J_.p, b_scale_.p, m_, n_, s);
    launch_apply_scale_to_reductions(b_colsq_.p, b_gs_.p, b_scale_.p, n_, s);
  }
  if (has_loss_) launch_dense_sum(b_cterm_.p, m_, b_scal_.p, s); else launch_dense_sumsq(b_r_.p, m_, b_scal_.p, s);
  launch_dense_gmax(b_gs_.p, b_scale_.p, x_, n_, b_scal_.p + 1, s);
  if (tangent_) launch_dense_sumsq(x_, ng_, b_scal_.p + 2, s);  // |x|^2 over the ambient vector
  SK_HIP_TRY(hipMemcpyAsync(h_scal_, b_scal_.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_ + 16, b_fail_.p, sizeof(int), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipEventRecord(ev_[kEvJac], s));
  SK_HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ev_[kEvBegin], ev_[kEvJac]) == hipSuccess) phase_[0] += 1e-3 * ms;
  int fail = 0; std::memcpy(&fail, h_scal_ + 16, sizeof(int));
  cost_ = 0.5 * h_scal_[0]; gmax_ = h_scal_[1]; xnorm_ = std::sqrt(h_scal_[2]);
  if (fail || !std::isfinite(cost_)) return SK_ERR_EVALUATION_FAILED;
  return SK_OK;
}

int DenseSolver::try_step(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm) {
  hipStream_t s = stream_;
  *valid = false;
  SK_HIP_TRY(hipEventRecord(ev_[kEvBegin], s));
  launch_lm_diagonal(b_colsq_.p, b_D_.p, n_, opt_.min_lm_diagonal, opt_.max_lm_diagonal, radius, s);
  SK_HIP_TRY(hipMemsetAsync(b_info_.p, 0, sizeof(int), s));
  SK_HIP_TRY(hipMemsetAsync(b_fail_.p, 0, sizeof(int), s));
  { const int one = 1; SK_HIP_TRY(hipMemcpyAsync(b_ok_.p, &one, sizeof(int), hipMemcpyHostToDevice, s)); }
  if (opt_.linear_solver_type == SK_DENSE_NORMAL_CHOLESKY) {
    SK_HIP_TRY(hipMemsetAsync(b_H_.p, 0, (size_t)npad_ * npad_ * sizeof(double), s));
    launch_dense_normal(b_J_.p, b_r_.p, m_, n_, b_H_.p, npad_, rhs_row_, s);
    launch_finish_normal_matrix(b_H_.p, npad_, n_, npad_, rhs_row_, b_D_.p, s);
    SK_HIP_TRY(hipEventRecord(ev_[kEvAssemble], s));
    cholesky_factor(b_H_.p, npad_, npad_, b_Linv_.p, b_info_.p, opt_.group_or(3), s, nullptr, &kt_);
    cholesky_backsolve(b_H_.p, npad_, n_, npad_, rhs_row_, b_Linv_.p, b_w_.p, b_y_.p, s, &kt_);
  } else {
    SK_HIP_TRY(hipEventRecord(ev_[kEvAssemble], s));
    launch_dense_qr(b_J_.p, b_r_.p, b_D_.p, m_, n_, b_A_.p, b_b_.p, b_y_.p, b_ok_.p, s);
  }
  SK_HIP_TRY(hipEventRecord(ev_[kEvChol], s));
  if (tangent_) launch_dense_plus(b_y_.p, b_scale_.p, x_, b_step_.p, x_new_, b_pblocks_.p, num_pblocks_, b_scal_.p, s);
  else launch_dense_step(b_y_.p, b_scale_.p, x_, b_step_.p, x_new_, n_, b_scal_.p, s);
  launch_dense_model(b_J_.p, b_r_.p, b_step_.p, m_, n_, b_scal_.p + 1, s);
  SK_HIP_TRY(hipEventRecord(ev_[kEvBacksub], s));
  int rc = evaluate(x_new_, false);
  const bool eval_failed = rc == SK_ERR_EVALUATION_FAILED;
  if (rc && !eval_failed) return rc;
  if (has_loss_) { apply_loss(b_rc_.p, false); launch_dense_sum(b_cterm_.p, m_, b_scal_.p + 2, s); }
  else launch_dense_sumsq(b_rc_.p, m_, b_scal_.p + 2, s);
  SK_HIP_TRY(hipEventRecord(ev_[kEvCost], s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_, b_scal_.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_ + 16, b_fail_.p, sizeof(int), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_ + 17, b_info_.p, sizeof(int), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_ + 18, b_ok_.p, sizeof(int), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ev_[kEvBegin], ev_[kEvAssemble]) == hipSuccess) phase_[1] += 1e-3 * ms;
  if (hipEventElapsedTime(&ms, ev_[kEvAssemble], ev_[kEvChol]) == hipSuccess) phase_[2] += 1e-3 * ms;
  if (hipEventElapsedTime(&ms, ev_[kEvChol], ev_[kEvBacksub]) == hipSuccess) phase_[3] += 1e-3 * ms;
  if (hipEventElapsedTime(&ms, ev_[kEvBacksub], ev_[kEvCost]) == hipSuccess) phase_[4] += 1e-3 * ms;
  int fail = 0, info = 0, ok = 1;
  std::memcpy(&fail, h_scal_ + 16, sizeof(int)); std::memcpy(&info, h_scal_ + 17, sizeof(int)); std::memcpy(&ok, h_scal_ + 18, sizeof(int));
  if (info || !ok || !std::isfinite(h_scal_[0]) || !std::isfinite(h_scal_[1])) return SK_OK;  // invalid step
  *valid = true;
  *step_norm = std::sqrt(h_scal_[0]);
  *mcc = -h_scal_[1];
  *new_cost = (fail || eval_failed) ? std::numeric_limits<double>::infinity() : 0.5 * h_scal_[2];
  return SK_OK;
}

int DenseSolver::write_back() {
  std::vector<double> x(ng_);
  SK_HIP_TRY(hipMemcpyAsync(x.data(), x_, ng_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  for (size_t b = 0; b < problem_->block_size.size(); ++b) std::memcpy(problem_->block_ptr[b], &x[block_off_[b]], problem_->block_size[b] * sizeof(double));
  return SK_OK;
}

}  // namespace

std::unique_ptr<SolverBase> make_dense_solver(const Options& o, Problem* p) { return std::unique_ptr<SolverBase>(new DenseSolver(o, p)); }

}  // namespace sk
