// DENSE_NORMAL_CHOLESKY on a tall dense problem whose residual blocks are rows of ONE
// parameter block (BASELINE.json config 5: 10k parameters x 1M residuals).  The Jacobian is held
// transposed in HBM (80 GB at full size, of 288 GB), J^T J is one long-K fp64-MFMA SYRK, the
// n x n normal matrix goes through the same blocked Cholesky as the reduced camera system.
#include <cmath>
#include <limits>

#include "bal_kernels.hpp"
#include "dense_kernels.hpp"
#include "dense_rows_kernels.hpp"
#include "solver.hpp"

namespace sk {

bool problem_is_dense_rows(const Problem& p) {
  if (p.rb_functor.empty() || p.block_size.size() != 1) return false;
  for (int f : p.rb_functor) if (f != SK_FUNCTOR_SYNTH_TANH_ROW) return false;
  return true;
}

namespace {

class DenseRowsSolver : public SolverBase {
 public:
  DenseRowsSolver(const Options& o, Problem* p) : SolverBase(o, p) {}
  ~DenseRowsSolver() override { if (h_scal_) (void)hipHostFree(h_scal_); }
  // algorithmic flops of J^T J (lower-triangular 128x128 tiles, 2*128*128*K each)
  double syrk_flops_per_solve() const override { return 0.5 * (npad_ / 128) * (npad_ / 128 + 1.0) * 2.0 * 128.0 * 128.0 * (double)m_pad_; }
  int distribution(double* allreduce_s, double* saved_s) const override {
    if (allreduce_s) *allreduce_s = 0.0;
    if (saved_s) *saved_s = 0.0;
    return opt_.allreduce && opt_.world > 1 ? SK_DISTRIBUTION_SHARDED : SK_DISTRIBUTION_REPLICATED;
  }
  bool stat(const std::string& name, double* value) const override {
    if (name == "allreduce_bytes") { *value = (double)b_pack_.n * sizeof(double); return true; }
    if (name == "jtj_flops_algorithmic") { *value = (double)m_ * (double)n_ * ((double)n_ + 1.0); return true; }
    return false;
  }

 protected:
  int setup() override;
  int evaluate_with_jacobian(bool first) override;
  int try_step(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm) override;
  void accept_candidate() override { std::swap(x_, x_new_); }
  int write_back() override;
  void describe(Summary* s) override {
    s->num_parameter_blocks = 1; s->num_parameters = n_; s->num_residual_blocks = m_all_; s->num_residuals = m_all_;
  }

 private:
  int n_ = 0, m_ = 0, npad_ = 0, rhs_row_ = 0;  // m_: this rank's rows
  int m_all_ = 0, row0_ = 0;
  DevBuf<int> b_pack_col0_;
  DevBuf<long long> b_pack_off_;
  DevBuf<double> b_pack_, b_sum2_;
  size_t m_pad_ = 0;
  int nslabs_ = 1;
  DevBuf<double> b_slabs_;
  DenseRowsArgs a_{};
  DevBuf<double> b_consts_, b_xa_, b_xb_, b_scale_, b_colsq_, b_gs_, b_D_, b_step_, b_y_, b_w_, b_r_, b_rc_, b_sd_, b_Jt_, b_H_, b_Linv_,
      b_partial_, b_small_, b_scal_;
  DevBuf<int> b_info_;
  DevBuf<LossNode> b_loss_nodes_;
  DevBuf<double> b_cterm_;
  CholeskyContext chol_ctx_;
  double* x_ = nullptr; double* x_new_ = nullptr;
  double* h_scal_ = nullptr;
};

int DenseRowsSolver::setup() {
  const Problem& p = *problem_;
  // Several ranks ("C5: shard rows of J"): rank r takes a contiguous run of the rows; J^T J, J^T r and
  // the sums of squares are all-reduced, the n x n Cholesky runs replicated.  The row's index is among its captured doubles
  // (consts), so a shard is a slice of them.
  n_ = p.block_size[0]; m_all_ = (int)p.rb_functor.size();
  row0_ = 0; m_ = m_all_;
  if (opt_.allreduce && opt_.world > 1) {
    row0_ = (int)((long long)m_all_ * opt_.rank / opt_.world);
    m_ = (int)((long long)m_all_ * (opt_.rank + 1) / opt_.world) - row0_;
    if (m_ <= 0) { set_error("fewer rows than ranks"); return SK_ERR_INVALID_ARGUMENT; }
  }
  // J^T J splits K = m_pad into nslabs_ chunks when the tile count alone would leave a ragged last wave of workgroups
  {
    const long tiles = (long)((n_ + 1 + 127) / 128) * ((n_ + 1 + 127) / 128 + 1) / 2;
    nslabs_ = 1;
    while (nslabs_ < 16 && tiles * nslabs_ < 8192 && (size_t)m_ / (nslabs_ * 2) >= 4096) nslabs_ *= 2;
    const size_t q = (size_t)16 * nslabs_;
    m_pad_ = ((size_t)m_ + q - 1) / q * q;
  }
  rhs_row_ = n_; npad_ = ((n_ + 1 + 127) / 128) * 128;
  hipStream_t s = stream_;
  {
    std::vector<double> mine(p.consts.begin() + 3 * (size_t)row0_, p.consts.begin() + 3 * (size_t)(row0_ + m_));
    SK_HIP_TRY(b_consts_.upload(mine, s));
  }
  std::vector<double> x(p.block_ptr[0], p.block_ptr[0] + n_);
  SK_HIP_TRY(b_xa_.upload(x, s)); SK_HIP_TRY(b_xb_.alloc(n_));
  x_ = b_xa_.p; x_new_ = b_xb_.p;
  SK_HIP_TRY(b_scale_.alloc(n_)); SK_HIP_TRY(b_colsq_.alloc(n_)); SK_HIP_TRY(b_gs_.alloc(n_)); SK_HIP_TRY(b_D_.alloc(n_)); SK_HIP_TRY(b_step_.alloc(n_));
  { std::vector<double> ones(n_, 1.0); SK_HIP_TRY(hipMemcpyAsync(b_scale_.p, ones.data(), n_ * sizeof(double), hipMemcpyHostToDevice, s)); SK_HIP_TRY(hipStreamSynchronize(s)); }
  SK_HIP_TRY(b_r_.alloc(m_)); SK_HIP_TRY(b_rc_.alloc(m_)); SK_HIP_TRY(b_sd_.alloc(m_));
  // rows [n, npad) of Jt stay zero: they pad the SYRK tiles; columns [m, m_pad) stay zero: they pad K
  SK_HIP_TRY(b_Jt_.alloc((size_t)npad_ * m_pad_)); SK_HIP_TRY(b_Jt_.zero(s));
  SK_HIP_TRY(b_H_.alloc((size_t)npad_ * npad_)); SK_HIP_TRY(b_H_.zero(s));
  if (nslabs_ > 1) SK_HIP_TRY(b_slabs_.alloc((size_t)nslabs_ * npad_ * npad_));
  SK_HIP_TRY(b_Linv_.alloc((size_t)npad_ * 128)); SK_HIP_TRY(b_Linv_.zero(s));
  SK_HIP_TRY(b_y_.alloc(npad_)); SK_HIP_TRY(b_w_.alloc(npad_)); SK_HIP_TRY(b_scal_.alloc(16));
  SK_HIP_TRY(b_partial_.alloc((size_t)(m_ + 255) / 256 + 16)); SK_HIP_TRY(b_small_.alloc(512)); SK_HIP_TRY(b_info_.alloc(1));
  SK_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_scal_), 64 * sizeof(double), hipHostMallocDefault));
  if (opt_.allreduce) {
    // what travels of J^T J: its lower block triangle, packed (414 MB at n = 10^4 instead of 800 MB)
    const int nblk = npad_ / 128;
    std::vector<int> col0(nblk, 0);
    std::vector<long long> off(nblk + 1, 0);
    for (int kb = 0; kb < nblk; ++kb) off[kb + 1] = off[kb] + (long long)128 * 128 * (kb + 1);
    SK_HIP_TRY(b_pack_col0_.upload(col0, s)); SK_HIP_TRY(b_pack_off_.upload(off, s));
    SK_HIP_TRY(b_pack_.alloc((size_t)off[nblk])); SK_HIP_TRY(b_sum2_.alloc(2 * (size_t)n_));
  }
  SK_HIP_TRY(cholesky_init());
  if (opt_.lookahead && chol_ctx_.init() != hipSuccess) { (void)hipGetLastError(); opt_.lookahead = false; }
  a_.m = m_; a_.n = n_; a_.m_pad = m_pad_; a_.consts = b_consts_.p; a_.inv_sqrt_n = 1.0 / std::sqrt((double)n_);
  // one robust loss for every row (ceres.i:159-184; the residual kernel applies loss and corrector per row)
  a_.loss_nodes = nullptr; a_.loss_root = p.rb_loss.empty() ? -1 : p.rb_loss[0]; a_.cterm = nullptr;
  for (size_t b = 1; b < p.rb_loss.size(); ++b)
    if (p.rb_loss[b] != p.rb_loss[0]) { set_error("dense rows take one loss function for all rows"); return SK_ERR_UNSUPPORTED; }
  if (a_.loss_root >= 0) {
    SK_HIP_TRY(b_loss_nodes_.upload(p.loss_nodes, s)); SK_HIP_TRY(b_cterm_.alloc(m_));
    a_.loss_nodes = b_loss_nodes_.p; a_.cterm = b_cterm_.p;
  }
  SK_HIP_TRY(hipStreamSynchronize(s));
  return SK_OK;
}

int DenseRowsSolver::evaluate_with_jacobian(bool first) {
  hipStream_t s = stream_;
  SK_HIP_TRY(hipEventRecord(ev_[kEvBegin], s));
  launch_rows_residual(a_, x_, b_r_.p, b_sd_.p, true, s);
  kt_.begin("rows_jacobian", s); launch_rows_jacobian(a_, b_sd_.p, b_scale_.p, b_Jt_.p, s); kt_.end("rows_jacobian", s);
  launch_rows_col_reduce(b_Jt_.p, b_r_.p, m_, n_, m_pad_, b_colsq_.p, b_gs_.p, s);
  if (opt_.allreduce) {  // column norms and gradient over every rank's rows
    SK_HIP_TRY(hipMemcpyAsync(b_sum2_.p, b_colsq_.p, n_ * sizeof(double), hipMemcpyDeviceToDevice, s));
    SK_HIP_TRY(hipMemcpyAsync(b_sum2_.p + n_, b_gs_.p, n_ * sizeof(double), hipMemcpyDeviceToDevice, s));
    int rc = allreduce(b_sum2_.p, 2 * (size_t)n_);
    if (rc) return rc;
    SK_HIP_TRY(hipMemcpyAsync(b_colsq_.p, b_sum2_.p, n_ * sizeof(double), hipMemcpyDeviceToDevice, s));
    SK_HIP_TRY(hipMemcpyAsync(b_gs_.p, b_sum2_.p + n_, n_ * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  if (first && opt_.jacobi_scaling) {
    launch_jacobi_scale(b_colsq_.p, b_scale_.p, n_, s);
    launch_rows_scale(b_Jt_.p, b_scale_.p, m_, n_, m_pad_, s);
    launch_apply_scale_to_reductions(b_colsq_.p, b_gs_.p, b_scale_.p, n_, s);
  }
  const int g = launch_rows_sumsq(b_r_.p, a_.cterm, m_, b_partial_.p, s);
  launch_final_reduce(b_partial_.p, g, g, 1, 0, b_scal_.p, s);
  if (opt_.allreduce) { int rc = allreduce(b_scal_.p, 1); if (rc) return rc; }  // sum r^2 over the ranks' rows
  const int gg = launch_grad_max_xnorm(b_gs_.p, b_scale_.p, x_, n_, b_small_.p, 256, s);
  launch_final_reduce(b_small_.p, 256, gg, 2, 1, b_scal_.p + 1, s);
  SK_HIP_TRY(hipMemcpyAsync(h_scal_, b_scal_.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipEventRecord(ev_[kEvJac], s));
  SK_HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ev_[kEvBegin], ev_[kEvJac]) == hipSuccess) phase_[0] += 1e-3 * ms;
  cost_ = 0.5 * h_scal_[0]; gmax_ = h_scal_[1]; xnorm_ = std::sqrt(h_scal_[2]);
  if (!std::isfinite(cost_)) return SK_ERR_EVALUATION_FAILED;
  return SK_OK;
}

int DenseRowsSolver::try_step(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm) {
  hipStream_t s = stream_;
  *valid = false;
  SK_HIP_TRY(hipEventRecord(ev_[kEvBegin], s));
  launch_lm_diagonal(b_colsq_.p, b_D_.p, n_, opt_.min_lm_diagonal, opt_.max_lm_diagonal, radius, s);
  SK_HIP_TRY(hipMemsetAsync(b_info_.p, 0, sizeof(int), s));
  // J^T J: one long-K MFMA SYRK over the transposed Jacobian (every lower tile is overwritten)
  launch_syrk_gram(b_H_.p, npad_, b_Jt_.p, (long)m_pad_, (int)(m_pad_ / nslabs_), nslabs_, b_slabs_.p, npad_ / 128, s, &kt_);
  if (opt_.allreduce) {  // J^T J = the sum of the ranks' partial products
    launch_tri_pack(b_H_.p, npad_, b_pack_.p, npad_ / 128, b_pack_col0_.p, b_pack_off_.p, true, s);
    int rc = allreduce(b_pack_.p, b_pack_.n);
    if (rc) return rc;
    launch_tri_pack(b_H_.p, npad_, b_pack_.p, npad_ / 128, b_pack_col0_.p, b_pack_off_.p, false, s);
  }
  launch_rows_set_rhs(b_H_.p, npad_, rhs_row_, b_gs_.p, n_, s);
  launch_finish_normal_matrix(b_H_.p, npad_, n_, npad_, rhs_row_, b_D_.p, s);
  SK_HIP_TRY(hipEventRecord(ev_[kEvAssemble], s));
  cholesky_factor(b_H_.p, npad_, npad_, b_Linv_.p, b_info_.p, opt_.group_or(3), s, opt_.lookahead ? &chol_ctx_ : nullptr, &kt_);
  cholesky_backsolve(b_H_.p, npad_, n_, npad_, rhs_row_, b_Linv_.p, b_w_.p, b_y_.p, s, &kt_);
  SK_HIP_TRY(hipEventRecord(ev_[kEvChol], s));
  launch_dense_step(b_y_.p, b_scale_.p, x_, b_step_.p, x_new_, n_, b_scal_.p, s);
  const int gm = launch_rows_model(b_Jt_.p, b_r_.p, b_step_.p, m_, n_, m_pad_, b_partial_.p, s);
  launch_final_reduce(b_partial_.p, gm, gm, 1, 0, b_scal_.p + 1, s);
  SK_HIP_TRY(hipEventRecord(ev_[kEvBacksub], s));
  launch_rows_residual(a_, x_new_, b_rc_.p, b_sd_.p, false, s);
  const int g = launch_rows_sumsq(b_rc_.p, a_.cterm, m_, b_partial_.p, s);
  launch_final_reduce(b_partial_.p, g, g, 1, 0, b_scal_.p + 2, s);
  if (opt_.allreduce) { int rc = allreduce(b_scal_.p + 1, 2); if (rc) return rc; }  // model term and candidate sum r^2 over the ranks' rows
  SK_HIP_TRY(hipEventRecord(ev_[kEvCost], s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_, b_scal_.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_ + 16, b_info_.p, sizeof(int), hipMemcpyDeviceToHost, s));
  SK_HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ev_[kEvBegin], ev_[kEvAssemble]) == hipSuccess) phase_[1] += 1e-3 * ms;
  if (hipEventElapsedTime(&ms, ev_[kEvAssemble], ev_[kEvChol]) == hipSuccess) phase_[2] += 1e-3 * ms;
  if (hipEventElapsedTime(&ms, ev_[kEvChol], ev_[kEvBacksub]) == hipSuccess) phase_[3] += 1e-3 * ms;
  if (hipEventElapsedTime(&ms, ev_[kEvBacksub], ev_[kEvCost]) == hipSuccess) phase_[4] += 1e-3 * ms;
  int info = 0; std::memcpy(&info, h_scal_ + 16, sizeof(int));
  if (info || !std::isfinite(h_scal_[0]) || !std::isfinite(h_scal_[1])) return SK_OK;  // invalid step
  *valid = true;
  *step_norm = std::sqrt(h_scal_[0]);
  *mcc = -h_scal_[1];
  *new_cost = std::isfinite(h_scal_[2]) ? 0.5 * h_scal_[2] : std::numeric_limits<double>::infinity();
  return SK_OK;
}

int DenseRowsSolver::write_back() {
  SK_HIP_TRY(hipMemcpyAsync(problem_->block_ptr[0], x_, n_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  return SK_OK;
}

}  // namespace

std::unique_ptr<SolverBase> make_dense_rows_solver(const Options& o, Problem* p) { return std::unique_ptr<SolverBase>(new DenseRowsSolver(o, p)); }

}  // namespace sk
