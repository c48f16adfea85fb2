// DENSE_SCHUR path for bundle-adjustment-shaped problems
// (EX/SimpleBundleAdjuster.scala:126-155): every residual block is
// SnavelyReprojectionError on (camera[9], point[3]).  Points are the e-blocks
// that the Schur complement eliminates; the 9C x 9C reduced camera system is
// factored by the MFMA Cholesky.  Multi-GPU: points (and their observations)
// are partitioned over ranks, cameras are replicated, and the reduced system
// is summed with one all-reduce per linear solve.
#include <algorithm>
#include <atomic>
#include <map>
#include <thread>
#include <cmath>
#include <limits>
#include <numeric>

#include "bal_kernels.hpp"
#include "solver.hpp"

namespace sk {

// The block structure DENSE_SCHUR eliminates: every residual block has r residuals over TWO parameter blocks, a "camera" of c
// coordinates (the f-block that stays in the reduced system) and a "point" of q coordinates (the e-block that is eliminated),
// the same (r; c, q) for every block, with r <= 2, c <= 9, q <= 3.  The kernels are written for the reference's bundle
// adjuster, (2; 9, 3) (EX/SimpleBundleAdjuster.scala:79-119); a smaller shape — a pinhole camera of six coordinates, a planar
// point — runs in the same kernels PADDED: the missing coordinates are inert unknowns (Jacobi scale 0, as a coordinate held
// constant by a SubsetParameterization: a zero Jacobian column, min_lm_diagonal / radius on the diagonal, step exactly 0), a
// missing residual row is zero.  The LM trajectory is that of the unpadded problem; the reduced system is 9 C wide instead of c C.
bool bal_block_shape(const Problem& p, int* r, int* c, int* q) {
  if (p.rb_functor.empty()) return false;
  const size_t b0 = 0;
  if (p.rb_pidx_off[b0 + 1] - p.rb_pidx_off[b0] != 2) return false;
  *r = p.rb_num_residuals[b0];
  *c = p.block_size[p.rb_pidx[p.rb_pidx_off[b0]]];
  *q = p.block_size[p.rb_pidx[p.rb_pidx_off[b0] + 1]];
  return true;
}
bool problem_is_bal_shaped(const Problem& p, std::string* why) {
  const size_t nb = p.rb_functor.size();
  if (nb == 0) { *why = "problem has no residual blocks"; return false; }
  int R = 0, Cs = 0, Qs = 0;
  const char* shape_msg = "DENSE_SCHUR is implemented for residual blocks with at most 2 residuals over a camera block of at most 9 and a point block of at most 3 "
                          "parameters, the same sizes for every block (SnavelyReprojectionError on the device, a recorded functor, or any host-callback cost "
                          "function of such a shape); not supported: another shape";
  if (!bal_block_shape(p, &R, &Cs, &Qs) || R < 1 || R > 2 || Cs < 1 || Cs > 9 || Qs < 1 || Qs > 3) { *why = shape_msg; return false; }
  // the registered device functor (2; 9, 3), a recorded functor, or ANY cost function of the shape through the director path
  // (sk_cost_function_new_callback: the caller's Evaluate, run on the host — CORE/CostFunctor.scala:40-51, ceres.i:48)
  for (size_t b = 0; b < nb; ++b) {
    if (p.rb_pidx_off[b + 1] - p.rb_pidx_off[b] != 2 || p.rb_num_residuals[b] != R || p.block_size[p.rb_pidx[p.rb_pidx_off[b]]] != Cs ||
        p.block_size[p.rb_pidx[p.rb_pidx_off[b] + 1]] != Qs) { *why = shape_msg; return false; }
    const CostFunction* cf = b < p.rb_cost.size() ? p.rb_cost[b] : nullptr;
    const bool host_ok = p.rb_functor[b] == SK_FUNCTOR_HOST_CALLBACK && cf && cf->callback;
    const bool tape_ok = p.tape_of_block(b) != nullptr;
    const bool snavely_ok = p.rb_functor[b] == SK_FUNCTOR_SNAVELY_REPROJECTION && R == 2 && Cs == 9 && Qs == 3;
    if (!snavely_ok && !host_ok && !tape_ok) { *why = shape_msg; return false; }
  }
  // one device functor per problem: the evaluation kernels are launched over all device-evaluated observations at once
  int device_functor = -1;
  for (size_t b = 0; b < nb; ++b) {
    if (p.rb_functor[b] == SK_FUNCTOR_HOST_CALLBACK) continue;
    if (device_functor < 0) device_functor = p.rb_functor[b];
    else if (device_functor != p.rb_functor[b]) { *why = "DENSE_SCHUR takes one device functor for all residual blocks (host-callback cost functions may be mixed in): not supported"; return false; }
  }
  if (device_functor >= kTapeFunctorBase && bal_tape_width(*p.tapes[device_functor - kTapeFunctorBase]) == 0) {
    *why = "the recorded functor needs more registers (or captures more doubles) than the device interpreter holds: not supported";
    return false;
  }
  // the Schur path carries identity and subset parameterizations and constant blocks; a quaternion or homogeneous-vector block
  // (a 4-block cannot be a camera or a point here anyway) sends the problem to the alternate solver like any other shape
  for (size_t b = 0; b < p.block_param.size(); ++b)
    if (p.block_param[b] >= 0) {
      const int t = p.params[p.block_param[b]].type;
      if (t != kParamIdentity && t != kParamSubset) {
        *why = "DENSE_SCHUR takes identity and subset parameterizations and constant parameter blocks (quaternion / homogeneous-vector blocks are implemented for DENSE_QR / DENSE_NORMAL_CHOLESKY; not supported here)";
        return false;
      }
    }
  std::vector<char> role(p.block_size.size(), 0);
  for (size_t b = 0; b < nb; ++b) {
    const int c = p.rb_pidx[p.rb_pidx_off[b]], q = p.rb_pidx[p.rb_pidx_off[b] + 1];
    if ((role[c] | 1) != 1 || (role[q] | 2) != 2) { *why = "a parameter block is used both as camera and as point"; return false; }
    role[c] = 1; role[q] = 2;
  }
  return true;
}

// Cameras (parameter slot 0) and points (slot 1) in first-appearance order, the
// per-observation indices, and the partition of points over `world` ranks:
// contiguous runs with (nearly) equal sum of k_p^2, because the Schur work of a
// point is quadratic in its track length k_p.
void bal_index_problem(const Problem& p, std::vector<int>* cam_block, std::vector<int>* pt_block, std::vector<int>* ocam,
                       std::vector<int>* opt) {
  const int Nall = (int)p.rb_functor.size();
  std::vector<int> cam_of_block(p.block_size.size(), -1), pt_of_block(p.block_size.size(), -1);
  ocam->resize(Nall); opt->resize(Nall);
  for (int b = 0; b < Nall; ++b) {
    const int cb = p.rb_pidx[p.rb_pidx_off[b]], pb = p.rb_pidx[p.rb_pidx_off[b] + 1];
    if (cam_of_block[cb] < 0) { cam_of_block[cb] = (int)cam_block->size(); cam_block->push_back(cb); }
    if (pt_of_block[pb] < 0) { pt_of_block[pb] = (int)pt_block->size(); pt_block->push_back(pb); }
    (*ocam)[b] = cam_of_block[cb]; (*opt)[b] = pt_of_block[pb];
  }
}

void bal_partition_points(const std::vector<int>& opt, int num_points, int world, std::vector<int>* cut) {
  std::vector<int> kp(num_points, 0);
  for (int q : opt) kp[q]++;
  cut->assign(world + 1, num_points);
  (*cut)[0] = 0;
  double total = 0.0;
  for (int q = 0; q < num_points; ++q) total += (double)kp[q] * kp[q];
  double acc = 0.0;
  int r = 1;
  for (int q = 0; q < num_points && r < world; ++q) {
    acc += (double)kp[q] * kp[q];
    while (r < world && acc >= total * r / world) (*cut)[r++] = q + 1;
  }
}

namespace {

// A stream that is capturing when an error makes the enqueueing function return early would fail every later call with
// a capture error instead of the real one: end and discard the capture, and stop replaying graphs for this solver.
struct CaptureGuard {
  hipStream_t s; bool active; bool* graph_mode;
  CaptureGuard(hipStream_t st, bool on, bool* gm) : s(st), active(on), graph_mode(gm) {}
  void release() { active = false; }
  ~CaptureGuard() {
    if (!active) return;
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(s, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    *graph_mode = false;
  }
};

class BalSolver : public SolverBase {
 public:
  BalSolver(const Options& o, Problem* p) : SolverBase(o, p) {}
  // over the fronts of the reduced system (one when it is not dissected); the tail front is factored launch by launch
  double syrk_flops_per_solve() const override {
    double f = 0.0;
    for (int k = 0; k < 3; ++k) if (fr_[k].nblk > 0) f += cholesky_syrk_flops((int)fr_[k].dim, group_, fr_[k].env(), chain_ok() && (k != 1 || tail_chain()), nullptr, fr_[k].ncols, fr_[k].tail_rows, fr_[k].tl());
    return f;
  }
  double syrk_c_bytes_per_solve() const override {
    double tiles = 0.0;
    for (int k = 0; k < 3; ++k) {
      double t = 0.0;
      if (fr_[k].nblk > 0) (void)cholesky_syrk_flops((int)fr_[k].dim, group_, fr_[k].env(), chain_ok() && (k != 1 || tail_chain()), &t, fr_[k].ncols, fr_[k].tail_rows, fr_[k].tl());
      tiles += t;
    }
    return tiles * 2.0 * 128.0 * 128.0 * sizeof(double);
  }
  void set_kernel_timing(int on) override {
    SolverBase::set_kernel_timing(on);
    kt_b_.only(on == 2 ? "gemm_syrk" : ""); kt_b_.enable(on != 0);
  }
  KernelTimer::Stat kernel_stat(const std::string& name) override {
    KernelTimer::Stat a = kt_.get_stat(name), b = kt_b_.get_stat(name);
    a.seconds += b.seconds; a.launches += b.launches;
    return a;
  }
  bool stat(const std::string& name, double* value) const override {
    const int nblk = npad_ / 128;
    if (name == "envelope_fill") {  // 128-blocks that are factored or updated, over the lower triangle of the undissected system
      double in = 0.0;
      for (int k = 0; k < 3; ++k) {
        const FrontHost& F = fr_[k];
        const int* env = F.env();
        for (int c = 0; c < F.ncols; ++c) {
          const int lm = env ? std::min(env[c], F.nblk - 1) : F.nblk - 1;
          const int t0 = F.tl() ? F.tl()[c] : F.nblk - F.tail_rows;
          in += (lm - c + 1) + std::max(0, F.nblk - std::max(t0, lm + 1));  // the run from the diagonal block down, and the tail rows (right-hand side; a border)
        }
      }
      *value = in / (0.5 * nblk * (nblk + 1.0));
      return true;
    }
    if (name == "reduced_system_blocks") { *value = nblk; return true; }  // (block rows of the undissected reduced system: cameras, pseudo-cameras of retained points, the right-hand side)
    if (name == "camera_order") { *value = camera_order_; return true; }
    if (name == "cholesky_flops_full") { const double n = 9.0 * C_; *value = n * n * n / 3.0; return true; }
    if (name == "cholesky_flops_plan") {
      double f = 0.0;
      for (int k = 0; k < 3; ++k) if (fr_[k].nblk > 0) f += cholesky_plan_flops(fr_[k].nblk, fr_[k].env(), fr_[k].ncols, fr_[k].tail_rows, fr_[k].tl());
      *value = f;
      return true;
    }
    if (name == "cholesky_columns_resident") {
      int r = 0;
      for (int k = 0; k < 3; ++k) {
        if (fr_[k].nblk == 0) continue;
        const CholeskyPlan plan = cholesky_plan(fr_[k].nblk, group_, fr_[k].env(), chain_live() && (k != 1 || tail_chain()), fr_[k].ncols, fr_[k].tail_rows, fr_[k].tl());
        for (char c : plan.resident) r += c ? 1 : 0;
      }
      *value = r;
      return true;
    }
    if (name == "chain_steps") {  // serial steps of the factorisation: block columns, the two leaf fronts of a lock-step dissection counted as one sequence
      *value = (dissected_ ? std::max(fr_[0].ncols, fr_[1].ncols) : 0) + fr_[2].ncols;
      return true;
    }
    if (name == "allreduce_bytes") { *value = (double)packed_elems_ * sizeof(double); return true; }
    if (name == "allreduce_bytes_full_triangle") { *value = (double)tri_packed_elems(nblk) * sizeof(double); return true; }
    if (name == "dissected") { *value = dissected_ ? 1.0 : 0.0; return true; }
    if (name == "retained_points") { *value = (double)retained_pts_.size(); return true; }
    if (name == "retained_model_us") { *value = retained_model_us_; return true; }
    if (name == "retained_model_us_without") { *value = retained_without_us_; return true; }
    if (name == "border_cameras") { *value = border_cams_; return true; }
    if (name == "border_gap") { *value = border_gap_; return true; }
    if (name == "border_model_us") { *value = border_model_us_; return true; }
    if (name == "border_model_us_plain") { *value = border_plain_us_; return true; }
    if (name == "segments") { *value = segmented_ ? segments_ : (dissected_ ? 2 : 1); return true; }
    if (name == "segment_cameras") { *value = segmented_ ? my_hi_ - my_lo_ : C_; return true; }
    if (name.rfind("model_us_segments_", 0) == 0) { const int k = atoi(name.c_str() + 18); if (k < 1 || k > 8) return false; *value = model_us_[k]; return true; }
    if (name == "dissection_head_cameras") { *value = cam_a_; return true; }
    if (name == "dissection_tail_cameras") { *value = cam_b_ - cam_a_; return true; }
    if (name == "dissection_separator_cameras") { *value = dissected_ ? C_ - cam_b_ : 0; return true; }
    if (name == "dissection_model_us_plain") { *value = dissect_t_plain_; return true; }
    if (name == "dissection_model_us") { *value = dissect_t_model_; return true; }
    if (name == "model_us_two_segments_with_members") { *value = two_segments_members_us_; return true; }
    return false;
  }
  // the grouping is the library's choice (Options::cholesky_group == 0) and the masked streams of the resident panel chain exist
  // ... (the PLAN is then the one with resident runs; whether they really run resident or, with the same grouping, launch
  // by launch — SK_CHOL_CHAIN_SERVER=0, a time-out — is cholesky_factor's business: chain_live())
  bool chain_ok() const { return opt_.cholesky_group == 0 && opt_.lookahead && chol_ctx_.server != nullptr; }
  bool chain_live() const { return chain_ok() && cholesky_chain_enabled(&chol_ctx_); }
  bool tail_chain() const { return chol_ctx_b_.server != nullptr; }  // one device, dissected: the tail front has a resident chain of its own
  int distribution(double* allreduce_s, double* saved_s) const override {
    if (allreduce_s) *allreduce_s = est_allreduce_s_;
    if (saved_s) *saved_s = est_saved_s_;
    return distribution_;
  }

 protected:
  int setup() override;
  int evaluate_with_jacobian(bool first) override;
  int try_step(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm) override;
  int try_step_once(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm, bool* chain_lost);
  void accept_candidate() override { std::swap(d_.xc, d_.xc_new); std::swap(d_.xp, d_.xp_new); parity_ ^= 1; }
  int write_back() override;
  void describe(Summary* s) override {
    s->num_parameter_blocks = (int)problem_->block_size.size();
    s->num_parameters = problem_->num_parameters();
    s->num_residual_blocks = (int)problem_->rb_functor.size();
    s->num_residuals = problem_->num_residuals;
    s->num_e_blocks = P_total_; s->num_f_blocks = C_ - pseudo_cams_;
  }

 private:
  int gather_rank_scalars(double* vals, int K, const int* ops);
  // director path: the host-evaluated observations at the point held in x_dev ([cameras | points]); jac: with Jacobians
  int host_callbacks(const double* x_dev, bool jac, bool* failed);
  std::vector<int> host_obs_;                 // local observation index of every host-evaluated residual block
  std::vector<const CostFunction*> host_cf_;  // ... and its cost function
  std::vector<double> host_x_, host_rows_h_;
  DevBuf<unsigned char> b_is_host_;
  DevBuf<int> b_host_obs_;
  DevBuf<double> b_host_rows_;
  std::vector<int> h_cam_, h_pt_;             // camera / local point of every local observation (host copy, for the callbacks)
  int gather_rank_scalars_signed(double* vals, int K);
  bool rank_table_on_device(int K) const;
  int enqueue_rank_table(int mode, int K);
  int fold_rank_table(double* vals, int K, const int* ops);

  int C_ = 0, P_total_ = 0, P_ = 0, N_ = 0;   // cameras, all points, local points, local observations
  int res_size_ = 2, cam_size_ = 9, pt_size_ = 3;  // the problem's own (r; c, q) (bal_block_shape): padded to (2; 9, 3) inside
  int n_ = 0, npad_ = 0, rhs_row_ = 0;
  std::vector<int> cam_block_, pt_block_;     // parameter block id of camera i / global point p
  std::vector<int> local_pt_;                 // global point id of local point
  BalDev d_{};
  DevBuf<LossNode> b_loss_nodes_;
  DevBuf<int> b_loss_of_obs_;
  DevBuf<int> b_cam_, b_pt_, b_pt_start_, b_cam_start_, b_cam_obs_, b_obs_slot_, b_seg_start_, b_seg_row_, b_seg_col_, b_pair_row_, b_pair_col_, b_short_segs_, b_long_segs_;
  int *fail_p_ = nullptr, *info_p_ = nullptr;  // device flags: slots 14 and 15 of b_scal_ (one reset, one copy back with the scalars)
  DevBuf<double> b_obs_, b_xc_, b_xp_, b_xc_new_, b_xp_new_, b_scale_, b_colsq_, b_gs_, b_step_, b_y_,
      b_r_, b_F_, b_Fcam_, b_E_, b_W_, b_rt_, b_M_, b_q_, b_S_, b_Linv_, b_partial_, b_scal_, b_small_;
  std::vector<int> env_last_;  // block envelope of S (cholesky_factor); empty = dense
  std::vector<int> env_tail_;  // ... and its tail profile when loop-closure cameras are ordered into a trailing border (choose_border); empty = none
  // Retained points (choose_retained_points): not eliminated, three to a pseudo-camera of the reduced system's layout.  C_ includes
  // the pseudo-cameras (cam_block_[i] == -1; no observations, no parameters: inert coordinates); retained_cam_[k] / k % 3: the
  // pseudo-camera (final numbering) and the slot in it of retained point k.
  std::vector<int> retained_pts_, retained_cam_;
  int pseudo_cams_ = 0;
  int border_members_ = 0;  // cameras and pseudo-cameras ordered behind the band (the bordered envelope's border; one device, dissected: the end of the separator)
  double retained_model_us_ = 0.0, retained_without_us_ = 0.0;
  std::vector<int> struct_ocam_, struct_opt_;   // the structure of the reduced system WITH pseudo-cameras (retained_graphs), final numbering; empty: ocam / opt as they are
  int struct_P_ = 0;
  DevBuf<int> b_kept_pt_, b_kept_cam_, b_kept_obs_, b_kept_obs_slot_;
  int P_own_ = 0;  // local points this rank accounts for in |x|^2, max |g| and the write-back: all of them, but for the copies of retained points whose home is another rank (a segmented world; they come last)
  DevBuf<int> b_kept_home_, b_kept_global_;  // per local retained point: 1 = this rank is its home; its index among ALL retained points (the slot of its sums in the small all-reduce)
  DevBuf<int> b_dup_a_, b_dup_b_, b_dup_cam_;  // two residual blocks on one (camera, point) pair: BalDev::dup_*
  int num_dup_ = 0;
  int num_kept_obs_ = 0;
  DevBuf<unsigned char> b_pseudo_;
  int border_cams_ = 0, border_gap_ = 0;          // cameras in that border; the jump in a point's camera list that made a visit
  double border_model_us_ = 0.0, border_plain_us_ = 0.0;
  // The reduced camera system as fronts (BalDev::front): 0 head, 1 tail, 2 root.  Not dissected: only the root, which is
  // then the whole system.  Each front is a dense dim x dim matrix inside b_S_.
  struct FrontHost {
    int nblk = 0, ncols = 0, cams = 0;   // block rows; block columns factored here; cameras eliminated here
    size_t dim = 0, s_off = 0, linv_off = 0, y_off = 0;
    int rhs_row = 0;
    int tail_rows = 1;                   // block rows at the end that every column reaches (cholesky_plan): > 1 for a segment between two separators
    std::vector<int> last;               // block envelope (empty: dense)
    std::vector<int> tail;               // tail profile of a bordered envelope (cholesky_factor; empty: the uniform tail_rows)
    const int* env() const { return last.empty() ? nullptr : last.data(); }
    const int* tl() const { return tail.empty() ? nullptr : tail.data(); }
  };
  FrontHost fr_[3];
  bool dissected_ = false;
  // Several ranks (SK_DISTRIBUTION_SEGMENTED): the camera sequence is cut into segments_ segments with a separator between
  // neighbours; this rank holds ONE leaf front (fr_[0]: segment role_, cameras [my_lo_, my_hi_) of the final numbering)
  // and the root (every separator).  Ranks beyond segments_ replicate rank (r mod segments_) and add zeros to every sum.
  bool segmented_ = false;
  int segments_ = 0, role_ = 0, my_lo_ = 0, my_hi_ = 0;
  bool replica_ = false;
  int fold_world_ = 0;      // ranks whose contributions count in gather_rank_scalars (0: all; segmented: the first segments_)
  int cam_a_ = 0, cam_b_ = 0, border_blocks_ = 0;
  std::vector<int> seg_off_;   // dissected: first camera of every segment in the final numbering, then cam_b_
  std::vector<int> sep_first_; // ... and of every separator, then C_
  std::vector<int> root_last_, root_tail_; // segmented: block envelope of the root (empty: dense — one separator); its tail profile (members of a border behind several separators)
  FrontView leaf_;             // segmented: this rank's leaf front
  DevBuf<int> b_border_row_[2], b_leaf_map_, b_leaf_gmap_;  // separator camera -> row of a leaf's border; border index -> root index (gmap: rhs row -> -1)
  double model_us_[9] = {0};   // the chain model's prediction per number of segments (index: segments; [1] = undissected)
  double dissect_t_plain_ = 0.0, dissect_t_model_ = 0.0, two_segments_members_us_ = 0.0;
  DissectedSystem ds_;
  CholeskyContext chol_ctx_b_;
  KernelTimer kt_b_;  // launches enqueued by the tail front's own thread
  // Small problems (BASELINE.json configs[1], BAL-49: a reduced system of four 128-blocks) are bound by launch latency:
  // ~60 launches of a few microseconds each per iteration.  Their two launch sequences — the linear solve with the
  // candidate evaluation, and the Jacobian evaluation — are identical from iteration to iteration except for the trust
  // region radius (read from device memory here) and for which of the two parameter buffers is current (`parity_`), so
  // each is captured into a hipGraph once per parity and replayed.
  bool tape_mode_ = false;      // the device functor is a recorded one (tape.hpp), interpreted by the evaluation kernels
  TapeDevBuffers tape_dev_;
  bool graph_mode_ = false;
  int parity_ = 0;
  hipGraphExec_t g_step_[2] = {nullptr, nullptr}, g_eval_[2] = {nullptr, nullptr};
  bool graph_ok() const { return graph_mode_ && !kt_.enabled(); }
  int finish_capture(hipStream_t s, hipGraphExec_t* exec);
  DevBuf<int> b_zero_col0_f_[3], b_mapB_;
  bool mapB_involution_ = false;
  DevBuf<double> b_yf_, b_wf_, b_ybB_;
  double order_hash_ = 0.0;    // of the camera order and the envelope: equal on every rank, or setup() fails
  int camera_order_ = 0;       // which candidate order of the cameras was kept (0 first appearance, 1 memory, 2 RCM)
  int group_ = 3;              // SYRK depth actually used (Options::cholesky_group, or chosen from the envelope)
  DevBuf<double> b_pack_;
  size_t packed_elems_ = 0;
  std::vector<int> pack_col0_h_;            // all-reduce packing: first block column of every block row that travels
  std::vector<long long> pack_off_h_;       // ... and where the row starts in the packed buffer
  DevBuf<int> b_pack_col0_;
  DevBuf<long long> b_pack_off_;
  int distribution_ = SK_DISTRIBUTION_SHARDED;
  double est_allreduce_s_ = 0.0, est_saved_s_ = 0.0;
  int choose_distribution(const std::vector<int>& opt);
  CholeskyContext chol_ctx_;
  double* h_scal_ = nullptr;  // pinned
  int partial_stride_ = 0;
  // The envelope of S has to be zero again before the next assembly (the factor overwrote it).  Round 4: the back-substitution
  // does it — every block of L below the diagonal is read exactly once there, by the owner workgroup of its column, which
  // writes zeros back (cholesky_backsolve(..., zero_after)); what it does not visit, the diagonal 128-blocks and a leaf front's
  // border x border square, is a small pass at the start of the next assembly (b_zero_min_f_: 16-40 MB instead of the envelope's
  // 378 MB on Ladybug-1723).  Rounds 2-3 zeroed the whole envelope on a stream of its own next to the Jacobian evaluation, which
  // that slowed to a third (bal_eval_jac 40 us alone, 116 us beside the zeroing: profiles/r04_point_phases_*).  The full pass
  // remains for a solver without resident kernels, after a factorisation that failed or timed out, and under SK_SCHEDULE_PLAIN.
  bool zero_by_backsolve_ = false;
  bool need_full_zero_ = false;  // the last back-substitution did not (or not surely) zero what it read
  DevBuf<int> b_zero_min_f_[3];
  bool pair_claimed_ = false;  // cholesky_claim_pair_servers: this solver may run a partner front's server beside its own
 public:
  ~BalSolver() override {
    if (pair_claimed_) cholesky_release_pair_servers(&chol_ctx_);
    for (hipGraphExec_t g : {g_step_[0], g_step_[1], g_eval_[0], g_eval_[1]}) if (g) (void)hipGraphExecDestroy(g);
    if (h_scal_) (void)hipHostFree(h_scal_);
  }
};

// ---- camera ordering for the reduced system --------------------------------------------------------------
// The order of the cameras inside S is the solver's to choose (Ceres, too, orders the blocks of the reduced
// system itself).  A block-banded S factors in a fraction of the flops of a full one (cholesky_factor's
// envelope), and whether S is banded depends on that order alone.  Candidates: first appearance in the residual
// blocks (what bal_index_problem yields), memory order of the camera blocks (the BAL file's numbering when the
// caller uses the reference's layout, EX/SimpleBundleAdjuster.scala:18-34), and reverse Cuthill-McKee on the
// co-visibility graph.  The one with the fewest trailing-update flops wins; ties keep the earlier candidate.
static std::vector<int> envelope_of_order(const std::vector<int>& ocam, const std::vector<int>& opt, const std::vector<int>& new_id, int C, int P,
                                          int nblk, std::vector<int>* first_col_out = nullptr) {
  std::vector<int> cmin(P, C);
  for (size_t b = 0; b < ocam.size(); ++b) cmin[opt[b]] = std::min(cmin[opt[b]], new_id[ocam[b]]);
  std::vector<int> first_col(nblk);
  for (int i = 0; i < nblk; ++i) first_col[i] = i;
  for (size_t b = 0; b < ocam.size(); ++b) {  // camera c shares point opt[b] with camera cmin: block (rows of c, columns of cmin)
    const int c = new_id[ocam[b]], col = (9 * cmin[opt[b]]) / 128;
    for (int row = (9 * c) / 128; row <= (9 * c + 8) / 128; ++row) first_col[row] = std::min(first_col[row], col);
  }
  if (first_col_out) *first_col_out = first_col;
  return cholesky_envelope_last(first_col);
}

// ---- two-way dissection of the camera sequence (chol_kernels.hip, "Two-way dissection") ---------------------------------
// Cameras in the chosen (banded) order: head [0, a), separator [a, b), tail [b, C), with no point seen from both the head
// and the tail: b = 1 + the last camera that shares a point with a camera before a.  The head is eliminated front to back
// and the tail back to front, side by side, so the serial panel chain is about half as long.  Where to cut is decided
// by a model of the two chains (microseconds per block column; constants measured on MI355X, profiles/r02_*): a block
// column costs the larger of its panel chain and its trailing update.
struct Dissection { int a = 0, b = 0; double t_plain = 0.0, t_dissected = 0.0; };
// Round 5: recalibrated on the bench line's chain_model records of round 4 (measured / model was 1.22-1.47 on the plans with retained
// points — every column chain-bound, two fronts in lock-step — 0.98 with every point eliminated, 0.83-0.89 on wide envelopes):
//   - a chain-bound column's cycle is the LONGER of the panel chain (40 us; 44.5 us when two fronts share the launches) and the thin
//     trailing SYRK it overlaps with, which the next column launch but one waits for: 12 us + 0.049 us per 32 x 128 tile, + 8 us of
//     hand-over (180 tiles 21 us, 760 tiles 49 us: profiles/r04_factor_timeline*.txt; "40-50 us up to 13 trailing rows, 60 at 20, 100
//     at 30" of round 1 is the same line);
//   - wide updates run at 38 TFLOP/s at a few dozen block rows and at 44-46 towards a full matrix (roofline_full: 44.9);
//   - the back-substitution is part of the phase the model is held against: 3.3 us per block column of a resident launch (+ 20 us).
static double thin_syrk_us(double tiles) { return tiles > 0.0 ? 12.0 + 0.049 * tiles : 0.0; }
static double thin_tiles(int h) { return h > 1 ? 2.0 * (h - 1.0) * h : 0.0; }  // 32 x 128 tiles of the trailing update behind the next block column: h - 1 block rows, lower triangle
static double column_cost_us(int h, bool resident_capable) {
  const double flops = 128.0 * 128.0 * 128.0 * ((double)h * h + h);
  if (resident_capable && h <= 24) return std::max(40.0, thin_syrk_us(thin_tiles(h)) + 8.0);
  if (h <= 24) return std::max(70.0, flops / 14e6);  // four launches per column
  return flops / (std::min(46.0, 34.0 + 0.12 * h) * 1e6) + 14.0;
}
// two chain-bound block columns, one of each leaf front, in ONE column launch and ONE thin SYRK (the lock-step dissection)
static double pair_cost_us(int h1, int h2) { return std::max(44.5, thin_syrk_us(thin_tiles(h1) + thin_tiles(h2)) + 8.0); }
static double backsolve_us(int block_columns) { return block_columns > 0 ? 20.0 + 3.3 * block_columns : 0.0; }
// tail_resident: the tail has a device of its own (segmented world) and runs under a resident panel chain like the head;
// on one device it is factored launch by launch next to the head's chain.
// lockstep: the schedule of ONE device since the end of round 3 — the tail's block columns ride in the launches of the head's
// trailing run of chain-bound columns (CholeskyPartner), so a paired step costs the dearer of its two columns and the head's
// block columns before that run are not shortened at all.
// extra_sep: cameras that join the separator whatever the cut (the pseudo-cameras of retained points, which every camera may couple with)
// extra_fwd (optional, per block column of the band): border block rows that are active in that column on top of its run — the tail
// profile of the bordered envelope — when the head is eliminated front to back; extra_bwd: the same for the tail front, for which the
// profile is not known (its columns reach the border's rows in another order): all of them.
// reach[c]: the last camera that shares a point with any camera <= c (cameras in the chosen order) — what a cut behind camera c - 1 needs
// as its separator's end.  Three passes over the observations: a caller that plans several cuts of one sequence forms it once (reach_in).
static std::vector<int> camera_reach(const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P) {
  std::vector<int> cmin(P, C), cmax(P, -1), reach(C);
  for (size_t b = 0; b < ocam.size(); ++b) { cmin[opt[b]] = std::min(cmin[opt[b]], ocam[b]); cmax[opt[b]] = std::max(cmax[opt[b]], ocam[b]); }
  for (int c = 0; c < C; ++c) reach[c] = c;
  for (int q = 0; q < P; ++q) if (cmax[q] >= 0) reach[cmin[q]] = std::max(reach[cmin[q]], cmax[q]);
  for (int c = 1; c < C; ++c) reach[c] = std::max(reach[c], reach[c - 1]);
  return reach;
}
static Dissection choose_dissection(const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P, int nblk, const std::vector<int>& last,
                                    const std::vector<int>& first_col, bool tail_resident, bool lockstep = false, int extra_sep = 0,
                                    const std::vector<int>* extra_fwd = nullptr, int extra_bwd = 0, const std::vector<int>* extra_bwd_col = nullptr,
                                    const std::vector<int>* reach_in = nullptr) {
  Dissection d;
  if (C < 64 || nblk < 24) return d;
  // reach[c]: the last camera that shares a point with any camera <= c (cameras in the chosen order)
  const std::vector<int> reach_own = reach_in ? std::vector<int>() : camera_reach(ocam, opt, C, P);
  const std::vector<int>& reach = reach_in ? *reach_in : reach_own;
  // per block column: height forward (rows below, as the envelope has it) and backward (rows above: the tail's view)
  std::vector<double> fwd(nblk), bwd(nblk), fwd_sum(nblk + 1, 0.0), bwd_sum(nblk + 1, 0.0);
  std::vector<int> height(nblk), height_b(nblk);
  std::vector<int> fc(first_col);
  for (int i = nblk - 2; i >= 0; --i) fc[i] = std::min(fc[i], fc[i + 1] < i + 1 ? fc[i + 1] : i);  // (monotone, as the backward envelope is)
  for (int c = 0; c < nblk; ++c) {
    const int hf = std::min(last[c], nblk - 1) - c + (last[c] < nblk - 1 ? 1 : 0) + (extra_fwd && c < (int)extra_fwd->size() ? (*extra_fwd)[c] : 0);
    height[c] = hf;
    fwd[c] = column_cost_us(hf, true);
    height_b[c] = c - std::min(fc[c], c) + 1 + (extra_bwd_col && c < (int)extra_bwd_col->size() ? (*extra_bwd_col)[c] : extra_bwd);
    bwd[c] = column_cost_us(height_b[c], tail_resident);
    d.t_plain += fwd[c];
  }
  d.t_plain += backsolve_us(nblk + (extra_sep > 0 ? (9 * extra_sep + 127) / 128 : 0));
  for (int c = 0; c < nblk; ++c) { fwd_sum[c + 1] = fwd_sum[c] + fwd[c]; bwd_sum[c + 1] = bwd_sum[c] + bwd[c]; }
  double best = d.t_plain;
  for (int a = 14; a + 14 < C; a += 7) {
    const int b = reach[a - 1] + 1;
    if (b >= C - 14) break;
    const int ca = (9 * a + 127) / 128, cb = (9 * b) / 128, E = (9 * (b - a + extra_sep) + 1 + 127) / 128;
    if (E > 24) continue;  // a separator that wide is no separator: its dense system is factored after both chains, alone
    double root = 0.0;
    for (int i = 0; i < E; ++i) root += column_cost_us(E - 1 - i, true);
    // (+ the back-substitutions: the root's, then the two interiors side by side)
    const double bs = backsolve_us(E) + backsolve_us(std::max(ca, nblk - cb)) + 20.0;
    double t = std::max(fwd_sum[ca], bwd_sum[nblk] - bwd_sum[cb]) + root + 120.0 + bs;  // + fork, join, border add
    if (lockstep) {
      // the head's chain-bound columns (its first few, and its trailing run) each carry one of the tail's; the others cost what they cost
      const int nb = nblk - cb;
      int k = 0;
      t = root + 100.0 + bs;  // + border add, the root's own start and join (180 us until the chain's starts and joins lost their events: round 5)
      for (int c = 0; c < ca; ++c) {
        if (height[c] <= 24 && k < nb) { t += height_b[nblk - 1 - k] <= 24 ? pair_cost_us(height[c], height_b[nblk - 1 - k]) : std::max(fwd[c], bwd[nblk - 1 - k]); ++k; }
        else t += fwd[c];
      }
      for (; k < nb; ++k) t += bwd[nblk - 1 - k];
    }
    if (t < best) { best = t; d.a = a; d.b = b; }
  }
  d.t_dissected = best;
  // (the lock-step schedule adds no queues and no second chain's interference: it is taken for half the predicted gain the
  // side-by-side one needed — measured on Ladybug-1723: predicted 9.8 %, 2.3 % of the iteration in the bench line)
  if (best > (lockstep ? 0.95 : 0.9) * d.t_plain) { d.a = d.b = 0; }
  return d;
}

// Block envelope of one front: `pos[c]` is camera c's first row in the front (interior or border), -1 when the camera has
// no rows in it; `interior[c]` whether its columns are eliminated in this front.  Only points that touch an interior
// camera shape the envelope (the border x border block is the Schur complement's, covered by the last columns' reach).
// tail_begin_row >= 0: the front's rows from there on (the rows of retained points, at the end of its border) are a border in the sense
// of cholesky_envelope_bordered — active from the first column that reaches them, not part of a column's contiguous run: *tail_out.
static std::vector<int> front_envelope(const std::vector<int>& ocam, const std::vector<int>& opt, const std::vector<int>& pos, const std::vector<char>& interior,
                                       int P, int nblk, int tail_rows = 1, int tail_begin_row = -1, std::vector<int>* tail_out = nullptr) {
  const int kNone = 1 << 30;
  std::vector<int> minpos(P, kNone);
  for (size_t b = 0; b < ocam.size(); ++b) if (interior[ocam[b]]) minpos[opt[b]] = std::min(minpos[opt[b]], pos[ocam[b]]);
  std::vector<int> first_col(nblk);
  for (int i = 0; i < nblk; ++i) first_col[i] = i;
  for (size_t b = 0; b < ocam.size(); ++b) {
    const int c = ocam[b];
    if (pos[c] < 0 || minpos[opt[b]] == kNone) continue;
    const int col = minpos[opt[b]] / 128;
    for (int row = pos[c] / 128; row <= (pos[c] + 8) / 128; ++row) first_col[row] = std::min(first_col[row], std::min(col, row));
  }
  if (tail_begin_row >= 0 && tail_out) {
    std::vector<int> last;
    cholesky_envelope_bordered(first_col, tail_begin_row / 128, &last, tail_out);
    return last;
  }
  return cholesky_envelope_last(first_col, tail_rows);
}

// ---- multi-way dissection over the ranks of a world ---------------------------------------------------
// R segments of the camera sequence with a separator between neighbours: separator k (1 <= k < R) = cameras [a[k-1], b[k-1])
// of the banded numbering, b = 1 + the last camera that shares a point with a camera before a.  Every segment is
// eliminated on a device of its own — the last one back to front, the others front to back — and the separators'
// block-tridiagonal system by every rank.  The cuts balance the segments' chains under the same model of a block
// column's cost as choose_dissection; the number of segments (at most max_segments) is the one with the shortest
// predicted critical path.  forced: cut wherever separators exist (tests, small problems), as evenly as the sequence allows.
struct Segments { std::vector<int> a, b; double t_plain = 0.0, t_model = 0.0; double model_us[9] = {0}; };
// member_cams: cameras that join the root whatever the cuts (the border's members: pseudo-cameras of retained points, loop-closure cameras) —
// rows of EVERY segment's front, active in a block column as extra_fwd / extra_bwd_col have it (as in choose_dissection), and a border of the root.
static Segments choose_segments(const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P, int nblk, const std::vector<int>& last,
                                const std::vector<int>& first_col, int max_segments, bool forced, int world = 0, int member_cams = 0,
                                const std::vector<int>* extra_fwd = nullptr, const std::vector<int>* extra_bwd_col = nullptr,
                                const std::vector<int>* reach_in = nullptr) {
  Segments out;
  const int mb = member_cams > 0 ? (9 * member_cams + 127) / 128 : 0;
  if (world <= 0) world = max_segments;
  max_segments = std::min(max_segments, 8);
  if (max_segments < 2 || C < 6) return out;
  const std::vector<int> reach_own = reach_in ? std::vector<int>() : camera_reach(ocam, opt, C, P);
  const std::vector<int>& reach = reach_in ? *reach_in : reach_own;
  if (C >= 64 && nblk >= 24) {
    std::vector<double> fwd(nblk), bwd(nblk), fwd_sum(nblk + 1, 0.0), bwd_sum(nblk + 1, 0.0);
    std::vector<int> fc(first_col);
    for (int i = nblk - 2; i >= 0; --i) fc[i] = std::min(fc[i], fc[i + 1] < i + 1 ? fc[i + 1] : i);
    for (int c = 0; c < nblk; ++c) {
      const int hf = std::min(last[c], nblk - 1) - c + (last[c] < nblk - 1 ? 1 : 0) + (extra_fwd && c < (int)extra_fwd->size() ? (*extra_fwd)[c] : mb);
      fwd[c] = column_cost_us(hf, true);
      bwd[c] = column_cost_us(c - std::min(fc[c], c) + 1 + (extra_bwd_col && c < (int)extra_bwd_col->size() ? (*extra_bwd_col)[c] : mb), true);
      out.t_plain += fwd[c];
    }
    for (int i = 0; i < mb; ++i) out.t_plain += column_cost_us(mb - 1 - i, true);
    for (int c = 0; c < nblk; ++c) { fwd_sum[c + 1] = fwd_sum[c] + fwd[c]; bwd_sum[c + 1] = bwd_sum[c] + bwd[c]; }
    out.model_us[1] = out.t_plain + backsolve_us(nblk + mb);
    const double plain_with_solve = out.model_us[1];
    auto sep_blocks = [&](int a) { return (9 * (reach[a - 1] + 1 - a) + 1 + 127) / 128; };
    // room[k]: the last camera at which a segment may START so that k more cuts (each a candidate below, each followed by a
    // segment of at least 14 cameras) still fit behind it
    std::vector<int> room(9, -1);
    room[0] = C - 15;
    for (int k = 1; k <= 8; ++k)
      for (int a = 14; a + 14 < C; a += 7) {
        const int b = reach[a - 1] + 1;
        if (b >= C - 14) break;
        if (sep_blocks(a) <= 24 && b <= room[k - 1]) room[k] = std::max(room[k], a - 14);
      }
    // can the sequence be cut into Rn segments none of whose chains is longer than T?  Greedy: every segment as long as T allows.
    auto plan_for = [&](int Rn, double T, std::vector<int>* as) {
      int pos = 0;
      as->clear();
      for (int sgm = 0; sgm + 1 < Rn; ++sgm) {
        int best_a = -1;
        for (int a = ((pos + 14 + 6) / 7) * 7; a + 14 < C; a += 7) {
          const int b = reach[a - 1] + 1;
          if (b >= C - 14) break;
          if (b > room[Rn - 2 - sgm]) continue;  // (room for the cuts still to come: a generous T must not spend the whole sequence on this segment)
          if (fwd_sum[(9 * a + 127) / 128] - fwd_sum[(9 * pos) / 128] > T) break;
          if (sep_blocks(a) <= 24) best_a = a;
        }
        if (best_a < 0) return false;
        as->push_back(best_a);
        pos = reach[best_a - 1] + 1;
      }
      return bwd_sum[nblk] - bwd_sum[(9 * pos) / 128] <= T;
    };
    double best = plain_with_solve;
    std::vector<int> best_as;
    for (int Rn = 2; Rn <= max_segments; ++Rn) {
      std::vector<int> as;
      if (!plan_for(Rn, out.t_plain, &as)) break;
      double lo = 0.0, hi = out.t_plain;
      for (int it = 0; it < 40; ++it) { const double mid = 0.5 * (lo + hi); if (plan_for(Rn, mid, &as)) hi = mid; else lo = mid; }
      (void)plan_for(Rn, hi, &as);
      double root = 0.0;  // the separators' block-tridiagonal system, one resident column after the other, on every rank
      for (size_t k = 0; k < as.size(); ++k) {
        const int E = sep_blocks(as[k]), Enext = k + 1 < as.size() ? sep_blocks(as[k + 1]) : 0;
        for (int i = 0; i < E; ++i) root += column_cost_us(E - 1 - i + Enext + mb, true);
      }
      for (int i = 0; i < mb; ++i) root += column_cost_us(mb - 1 - i, true);  // (the members: a dense border of the root)
      // + the all-reduce of the root over the world's ranks: its lower triangle inside the block-tridiagonal envelope, a ring over
      // one xGMI link per direction (153 GB/s: 2 (W - 1) / W x the bytes) — 39 MB and 0.4 ms for ONE 24-block separator, which is
      // what keeps a second wide separator from paying on the Ladybug-shaped problem
      double blocks = 0.0;
      for (size_t k = 0; k < as.size(); ++k) {
        const double E = sep_blocks(as[k]), Eprev = k > 0 ? sep_blocks(as[k - 1]) : 0.0;
        blocks += 0.5 * E * (E + 1.0) + E * Eprev + E * mb;
      }
      blocks += 0.5 * mb * (mb + 1.0);
      const int W = std::max(2, world);
      const double allreduce_us = 50.0 + 2.0 * (W - 1.0) / W * blocks * 128.0 * 128.0 * 8.0 / 153e3;
      int root_blocks = mb;
      for (size_t k = 0; k < as.size(); ++k) root_blocks += sep_blocks(as[k]);
      const double t = hi + root + 170.0 + allreduce_us + backsolve_us(root_blocks) + backsolve_us((nblk + Rn - 1) / Rn);  // + fork, join, border add; the root's and a segment's back-substitution
      out.model_us[Rn] = t;
      if (dev_knobs().debug_segments) std::fprintf(stderr, "[skeres_amd] %d segments: longest chain %.0f us, root %.0f us, all-reduce %.0f us (%.0f blocks) -> %.0f us (undissected %.0f)\n", Rn, hi, root, allreduce_us, blocks, t, out.t_plain);
      // (forced: a segment per rank, as far as the sequence can be cut; otherwise a further segment has to beat the plan so far by
      // 5 %: the model is no better than that, and every separator is more to all-reduce and to factor on every rank)
      if (forced ? true : t < best * (Rn > 2 ? 0.95 : 1.0)) { best = t; best_as = as; }
    }
    out.t_model = best;
    if (!best_as.empty() && (forced || best <= 0.9 * plain_with_solve)) {
      for (int a : best_as) { out.a.push_back(a); out.b.push_back(reach[a - 1] + 1); }
      return out;
    }
  }
  if (!forced) return out;
  for (int Rn = max_segments; Rn >= 2 && out.a.empty(); --Rn) {
    std::vector<int> as, bs;
    int pos = 0;
    bool ok = true;
    for (int k = 1; k < Rn && ok; ++k) {
      // the cut nearest to the k-th Rn-th of the sequence (from there towards the front) whose separator leaves room behind it
      int a = std::min(C - 1, std::max(pos + 1, (int)((long)k * C / Rn)));
      while (a > pos && reach[a - 1] + 1 >= C - (Rn - 1 - k)) --a;
      if (a <= pos) { ok = false; break; }
      as.push_back(a); bs.push_back(reach[a - 1] + 1);
      pos = bs.back();
    }
    if (ok && pos < C) { out.a = as; out.b = bs; }
  }
  return out;
}

// f(i) for i in [0, n) on up to twelve host threads (the caller's among them).  The planner's candidates — each a handful of passes over every
// observation — are independent of each other; which one is taken is decided afterwards, in the candidates' own order, so the plan does not
// depend on the number of threads (round 5: the plan of the reduced system was 0.7 s of Ladybug-1723's 0.8 s of set-up, 8 s of Venice-1778's 9).
static std::atomic<int> g_plan_helpers{0};  // helper threads of the planner alive in this process (nested calls share one budget)
template <class F>
static void plan_parallel_for(int n, F f) {
  const int budget = std::min(12, std::max(1, (int)std::thread::hardware_concurrency())) - 1;
  int helpers = 0;
  while (helpers < n - 1) {  // (claim helper threads one by one, as far as the budget goes)
    int cur = g_plan_helpers.load();
    if (cur >= budget) break;
    if (g_plan_helpers.compare_exchange_weak(cur, cur + 1)) ++helpers;
  }
  if (helpers == 0) { for (int i = 0; i < n; ++i) f(i); return; }
  std::atomic<int> next{0};
  auto work = [&] { for (int i; (i = next.fetch_add(1)) < n;) f(i); };
  std::vector<std::thread> pool;
  pool.reserve((size_t)helpers);
  for (int t = 0; t < helpers; ++t) {
    try { pool.emplace_back(work); } catch (...) { break; }  // (no thread to be had: the ones there are, and this one, do the work)
  }
  work();
  for (std::thread& t : pool) t.join();
  g_plan_helpers.fetch_sub(helpers);
}
static std::vector<int> rcm_order(const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P) {
  // co-visibility graph, thinned: the cameras of a point are chained in index order and the ends joined
  // (the cameras of every point as flat sorted lists — a counting sort and one small sort per point)
  std::vector<int> pstart((size_t)P + 1, 0), pcam(ocam.size());
  for (int q : opt) pstart[(size_t)q + 1]++;
  for (int q = 0; q < P; ++q) pstart[(size_t)q + 1] += pstart[(size_t)q];
  { std::vector<int> fill(pstart.begin(), pstart.end() - 1); for (size_t b = 0; b < ocam.size(); ++b) pcam[(size_t)fill[(size_t)opt[b]]++] = ocam[b]; }
  for (int q = 0; q < P; ++q) std::sort(pcam.begin() + pstart[(size_t)q], pcam.begin() + pstart[(size_t)q + 1]);
  std::vector<std::vector<int>> adj(C);
  if (C <= 16384) {
    // the edge set as a bit matrix (32 MB at 16384 cameras): no list of a few hundred thousand pairs to sort and to thin out
    const size_t words = ((size_t)C + 63) / 64;
    std::vector<unsigned long long> bits((size_t)C * words, 0ull);
    auto edge = [&](int a, int b) { if (a != b) { bits[(size_t)a * words + (size_t)b / 64] |= 1ull << (b % 64); bits[(size_t)b * words + (size_t)a / 64] |= 1ull << (a % 64); } };
    for (int q = 0; q < P; ++q) {
      const int a = pstart[(size_t)q], e = pstart[(size_t)q + 1];
      for (int k = a; k + 1 < e; ++k) edge(pcam[(size_t)k], pcam[(size_t)k + 1]);
      if (e - a > 2) edge(pcam[(size_t)a], pcam[(size_t)e - 1]);
    }
    for (int u = 0; u < C; ++u)
      for (size_t w = 0; w < words; ++w)
        for (unsigned long long m = bits[(size_t)u * words + w]; m; m &= m - 1) adj[(size_t)u].push_back((int)(w * 64) + __builtin_ctzll(m));
  } else {
    std::vector<std::pair<int, int>> edges;
    for (int q = 0; q < P; ++q) {
      const int a = pstart[(size_t)q], e = pstart[(size_t)q + 1];
      for (int k = a; k + 1 < e; ++k) edges.emplace_back(pcam[(size_t)k], pcam[(size_t)k + 1]);
      if (e - a > 2) edges.emplace_back(pcam[(size_t)a], pcam[(size_t)e - 1]);
    }
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    for (auto& e : edges) if (e.first != e.second) { adj[e.first].push_back(e.second); adj[e.second].push_back(e.first); }
  }
  for (auto& a : adj) std::sort(a.begin(), a.end(), [&](int x, int y) { return adj[x].size() != adj[y].size() ? adj[x].size() < adj[y].size() : x < y; });
  std::vector<int> order, level(C, -1);
  order.reserve(C);
  auto bfs = [&](int root, std::vector<int>* out) {  // Cuthill-McKee sweep of root's component (unvisited part); returns the last vertex
    const size_t begin = out->size();
    out->push_back(root); level[root] = 0;
    for (size_t h = begin; h < out->size(); ++h) {
      const int u = (*out)[h];
      for (int v : adj[u]) if (level[v] < 0) { level[v] = level[u] + 1; out->push_back(v); }
    }
    return out->back();
  };
  std::vector<char> done(C, 0);
  for (int seed = 0; seed < C; ++seed) {
    if (done[seed]) continue;
    // pseudo-peripheral start: two sweeps, each restarting from the far end of the previous one
    int root = seed;
    for (int rep = 0; rep < 2; ++rep) {
      std::vector<int> tmp;
      const int far = bfs(root, &tmp);
      for (int v : tmp) level[v] = -1;
      root = far;
    }
    const size_t begin = order.size();
    bfs(root, &order);
    for (size_t h = begin; h < order.size(); ++h) done[order[h]] = 1;
  }
  std::reverse(order.begin(), order.end());
  std::vector<int> new_id(C);
  for (int k = 0; k < C; ++k) new_id[order[k]] = k;
  return new_id;
}

// The candidate orders of the cameras inside the reduced system and the one with the fewest trailing-update flops (ties keep
// the earlier candidate).  with_memory_order: host addresses are this process's own — with separately allocated camera
// blocks the order could differ from rank to rank, and the ranks must build the same reduced system: one process only (the
// slot then repeats candidate 0, which keeps the numbering of sk_solver_stat("camera_order")).
static std::vector<std::vector<int>> camera_order_candidates(const Problem& p, const std::vector<int>& cam_block, const std::vector<int>& ocam,
                                                             const std::vector<int>& opt, int C, int P, bool with_memory_order) {
  std::vector<std::vector<int>> cand;
  { std::vector<int> id(C); std::iota(id.begin(), id.end(), 0); cand.push_back(id); }  // first appearance
  if (with_memory_order) {
    std::vector<int> by_addr(C); std::iota(by_addr.begin(), by_addr.end(), 0);
    std::sort(by_addr.begin(), by_addr.end(), [&](int a, int b) { return p.block_ptr[cam_block[a]] < p.block_ptr[cam_block[b]]; });
    std::vector<int> id(C); for (int k = 0; k < C; ++k) id[by_addr[k]] = k; cand.push_back(id);
  } else {
    cand.push_back(cand[0]);
  }
  cand.push_back(rcm_order(ocam, opt, C, P));
  return cand;
}
static void choose_camera_order(const std::vector<std::vector<int>>& cand, const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P, int npad,
                                int* best_k, std::vector<int>* best_env, double* best_flops) {
  double best = -1.0;
  std::vector<std::vector<int>> envs(cand.size());
  std::vector<double> flops(cand.size(), 0.0);
  plan_parallel_for((int)cand.size(), [&](int k) {  // (the candidates' envelopes side by side; the choice in their order)
    envs[(size_t)k] = envelope_of_order(ocam, opt, cand[(size_t)k], C, P, npad / 128);
    flops[(size_t)k] = cholesky_syrk_flops(npad, 1, envs[(size_t)k].data());
  });
  for (size_t k = 0; k < cand.size(); ++k) {
    const double f = flops[k];
    if (best < 0.0 || f < best * (1.0 - 1e-9)) { best = f; *best_k = (int)k; best_env->swap(envs[k]); }
  }
  *best_flops = best;
}

// ---- loop closures: the cameras that revisit a place, ordered into a trailing BORDER (round 4) ------------------------------
// A camera sequence that comes back to a street it has seen couples two distant windows of the band: in the band's own
// order every block column between the two windows is dragged into the envelope (a handful of such tracks fill it: 0.37 ->
// 0.99 of the blocks on the Ladybug-shaped problem).  Numbered BEHIND the band instead, the revisiting cameras are border
// rows — "rows active in every column from the column that first reaches them", the tail profile of cholesky_factor — the
// band keeps its width, and the border's own few block columns are factored last.  Which cameras: in a point's ascending
// camera list a jump of more than `gap` cameras separates visits; the cameras of the later visits (or, the other variant,
// of all visits but the last) go to the border.  Which gap and variant, and whether at all: the chain model of
// choose_dissection (microseconds per block column), over a few gaps; the border is taken when it predicts 10 % less than
// the band's own envelope.  A result of the solve does not depend on the order (EX/SimpleBundleAdjuster.scala:147-152: DENSE_SCHUR
// of Ceres orders its reduced system itself); tests hold the bordered order against the plain one and the oracle.
struct BorderChoice {
  std::vector<int> new_id;        // banded numbering -> final numbering (band cameras in order, then the border)
  std::vector<int> last, tail;    // bordered envelope (cholesky_envelope_bordered)
  int border_cams = 0, gap = 0, variant = 0;
  double model_us = 0.0, plain_us = 0.0;
};
static double envelope_model_us(int nblk, const std::vector<int>& last, const int* tail) {
  double t = 0.0;
  for (int c = 0; c < nblk; ++c) {
    const int lm = std::min(last[c], nblk - 1);
    const int main_rows = lm > c ? lm - c : 0;
    const int t0 = tail ? tail[c] : nblk - 1;
    const int h = main_rows + std::max(0, nblk - std::max(t0, c + 1 + main_rows));
    t += column_cost_us(h, true);
  }
  return t + backsolve_us(nblk);
}
// A camera graph: per observation its camera and its point.  Two of them describe a problem with RETAINED points (below): `g`, the
// observations of the points the Schur complement eliminates, over the C real cameras; `x`, the structure of the reduced system —
// g's observations, and for every observation of a retained point a point of its own that couples the observation's camera with
// the retained point's pseudo-camera (index >= g.C).  Without retained points x is g.
struct CamGraph { const std::vector<int>* ocam; const std::vector<int>* opt; int C, P; };
static void point_camera_lists(const CamGraph& g, std::vector<int>* pstart, std::vector<int>* pcam) {
  pstart->assign(g.P + 1, 0); pcam->resize(g.ocam->size());
  for (int q : *g.opt) (*pstart)[q + 1]++;
  for (int q = 0; q < g.P; ++q) (*pstart)[q + 1] += (*pstart)[q];
  { std::vector<int> fill(pstart->begin(), pstart->end() - 1); for (size_t b = 0; b < g.ocam->size(); ++b) (*pcam)[fill[(*g.opt)[b]]++] = (*g.ocam)[b]; }
  for (int q = 0; q < g.P; ++q) std::sort(pcam->begin() + (*pstart)[q], pcam->begin() + (*pstart)[q + 1]);
}
// The cameras of every point as lists (point_camera_lists) — what choose_border works on.  A caller that scores many variants of one
// graph (choose_retained_points: the same observations with a few points taken out) forms them once and derives each variant's
// by copying, instead of a counting sort and 150 000 small sorts per variant.
struct PointLists { std::vector<int> start, cam; };
// first_col of envelope_of_order from the lists: block row of every camera of a point <- the block column of the point's first camera
static void first_cols_from_lists(const std::vector<int>& start, const std::vector<int>& cam, const std::vector<int>& new_id, int nblk, std::vector<int>* first_col) {
  first_col->resize((size_t)nblk);
  for (int i = 0; i < nblk; ++i) (*first_col)[(size_t)i] = i;
  const int P = (int)start.size() - 1;
  for (int q = 0; q < P; ++q) {
    const int a = start[(size_t)q], e = start[(size_t)q + 1];
    if (a == e) continue;
    int mn = new_id[(size_t)cam[(size_t)a]];
    for (int k = a + 1; k < e; ++k) mn = std::min(mn, new_id[(size_t)cam[(size_t)k]]);
    const int col = (9 * mn) / 128;
    for (int k = a; k < e; ++k) {
      const int c = new_id[(size_t)cam[(size_t)k]];
      for (int row = (9 * c) / 128; row <= (9 * c + 8) / 128; ++row) (*first_col)[(size_t)row] = std::min((*first_col)[(size_t)row], col);
    }
  }
}
// g, x: cameras in the banded numbering (pseudo-cameras behind the real ones).  mode: SK_BORDER_AUTO (the model decides) / SK_BORDER_ON
// (the best candidate whatever the model says).  gaps_ok: loop-closure cameras may go to the border; pseudo-cameras always do, and with
// them a border is always returned (plain_us is then the model of the border of pseudo-cameras alone).
// gl, xl (optional): the lists of g and of x, formed by the caller — g.ocam / x.ocam may then be null (the lists are all that is read).
static bool choose_border(const CamGraph& g, const CamGraph& x, int nblk, const std::vector<int>& plain_last, int mode, bool gaps_ok, BorderChoice* out,
                          const PointLists* gl = nullptr, const PointLists* xl = nullptr) {
  const int C = g.C, Cx = x.C;
  const bool forced = Cx > C;
  out->plain_us = forced ? 0.0 : envelope_model_us(nblk, plain_last, nullptr);
  if (!forced && (C < 8 || !gaps_ok)) return false;
  // cameras of every point, ascending
  std::vector<int> pstart_s, pcam_s, xstart_s, xcam_s;
  if (!gl) point_camera_lists(g, &pstart_s, &pcam_s);
  if (forced && !xl) point_camera_lists(x, &xstart_s, &xcam_s);
  const std::vector<int>& pstart = gl ? gl->start : pstart_s;
  const std::vector<int>& pcam = gl ? gl->cam : pcam_s;
  const std::vector<int>& xstart = forced ? (xl ? xl->start : xstart_s) : pstart;
  const std::vector<int>& xcam = forced ? (xl ? xl->cam : xcam_s) : pcam;
  const int xP = (int)xstart.size() - 1;  // (== x.P)
  int max_jump = 0;
  for (int q = 0; q < g.P; ++q)
    for (int k = pstart[q] + 1; k < pstart[q + 1]; ++k) max_jump = std::max(max_jump, pcam[k] - pcam[k - 1]);
  bool found = false;
  double best = mode == SK_BORDER_ON ? std::numeric_limits<double>::max() : 0.9 * out->plain_us;
  // one candidate: the real cameras marked in `mark` (nb of them) and every pseudo-camera behind the band
  auto candidate = [&](std::vector<char>& mark, int nb, int gap, int variant, BorderChoice* cand) {
    mark.resize(Cx, 1);
    // first band camera each border camera couples with (through any of its points): the border is ordered so that the
    // cameras reached first come LAST — a column's tail rows are a suffix of the matrix
    std::vector<int> band_id(Cx, -1);
    int Cb = 0;
    for (int c = 0; c < Cx; ++c) if (!mark[c]) band_id[c] = Cb++;
    std::vector<int> first_band(Cx, Cx);
    for (int q = 0; q < xP; ++q) {
      int mn = Cx;
      for (int k = xstart[q]; k < xstart[q + 1]; ++k) if (!mark[xcam[k]]) { mn = band_id[xcam[k]]; break; }
      for (int k = xstart[q]; k < xstart[q + 1]; ++k) if (mark[xcam[k]]) first_band[xcam[k]] = std::min(first_band[xcam[k]], mn);
    }
    std::vector<int> border;
    for (int c = 0; c < Cx; ++c) if (mark[c]) border.push_back(c);
    std::stable_sort(border.begin(), border.end(), [&](int a, int b) { return first_band[a] > first_band[b]; });
    cand->new_id = band_id;
    for (size_t k = 0; k < border.size(); ++k) cand->new_id[border[k]] = Cb + (int)k;
    std::vector<int> first_col;
    if (xl || (gl && !forced)) first_cols_from_lists(xstart, xcam, cand->new_id, nblk, &first_col);  // (the same minima as envelope_of_order's, point by point)
    else (void)envelope_of_order(*x.ocam, *x.opt, cand->new_id, Cx, x.P, nblk, &first_col);
    cholesky_envelope_bordered(first_col, (9 * Cb) / 128, &cand->last, &cand->tail);
    cand->model_us = envelope_model_us(nblk, cand->last, cand->tail.data());
    cand->border_cams = nb; cand->gap = gap; cand->variant = variant;
    mark.resize(C);
  };
  if (forced) {
    std::vector<char> mark(C, 0);
    BorderChoice cand;
    candidate(mark, 0, 0, 0, &cand);
    out->plain_us = cand.model_us;
    cand.plain_us = cand.model_us;
    *out = cand; found = true;
    best = mode == SK_BORDER_ON ? cand.model_us : 0.9 * cand.model_us;  // (loop-closure cameras on top of the pseudo-cameras: when the model gains another 10 %)
  }
  // the candidates (gap, variant): their marks first — cheap, and a candidate whose marks repeat the one before it is dropped — then their
  // envelopes side by side on host threads (plan_parallel_for), then the choice, in the candidates' order
  struct GapCand { int gap, variant, nb; std::vector<char> mark; BorderChoice bc; };
  std::vector<GapCand> gc, all;
  for (int gap = 4; gaps_ok && C >= 8 && gap < C && gap < max_jump; gap *= 2)
    for (int variant = 0; variant < 2; ++variant) all.push_back(GapCand{gap, variant, 0, std::vector<char>(), BorderChoice()});
  plan_parallel_for((int)all.size(), [&](int ai) {  // (a pass over every point's list per candidate: side by side)
    GapCand& c = all[(size_t)ai];
    const int gap = c.gap;
    std::vector<char>& mark = c.mark;
    mark.assign((size_t)C, 0);
    int nb = 0;
    for (int q = 0; q < g.P; ++q) {
      const int a = pstart[q], e = pstart[q + 1];
      if (c.variant == 0) {  // everything behind the first jump
        int k = a + 1;
        while (k < e && pcam[k] - pcam[k - 1] <= gap) ++k;
        for (; k < e; ++k) if (!mark[pcam[k]]) { mark[pcam[k]] = 1; ++nb; }
      } else {             // everything before the last jump
        int k = e - 1;
        while (k > a && pcam[k] - pcam[k - 1] <= gap) --k;
        for (int i = a; i < k; ++i) if (!mark[pcam[i]]) { mark[pcam[i]] = 1; ++nb; }
      }
    }
    c.nb = nb;
  });
  {
    std::vector<char> prev_mark;
    for (GapCand& c : all) {
      if (c.nb == 0 || c.nb > C / 4 || C - c.nb < 4) continue;   // (a border that wide is no border: its dense system would be the factorisation)
      if (c.mark == prev_mark) continue;
      prev_mark = c.mark;
      gc.push_back(std::move(c));
    }
  }
  plan_parallel_for((int)gc.size(), [&](int i) { candidate(gc[(size_t)i].mark, gc[(size_t)i].nb, gc[(size_t)i].gap, gc[(size_t)i].variant, &gc[(size_t)i].bc); });
  for (GapCand& c : gc) {
    c.bc.plain_us = out->plain_us;
    if (c.bc.model_us < best) { best = c.bc.model_us; *out = c.bc; found = true; }
  }
  return found;
}

// The order of the cameras inside the reduced system as setup() takes it: the candidate with the fewest trailing-update flops,
// or a bordered variant of one of the candidates when the chain model prefers it.  From host data alone.
struct CameraOrderPlan {
  std::vector<int> id;           // first-appearance numbering -> final numbering (pseudo-cameras of retained points: indices >= the real cameras')
  std::vector<int> plain_id;     // ... of the best candidate as it stands (real cameras only; what the retained points are chosen on)
  std::vector<std::vector<int>> candidates;  // the candidate orders this plan was chosen from (camera_order_candidates of g)
  std::vector<int> last, tail;   // the envelope of the reduced system in that numbering (tail: empty unless bordered)
  int candidate = 0;             // 0 first appearance, 1 memory order, 2 RCM
  bool bordered = false;
  BorderChoice border;
  double flops = 0.0;            // trailing-update flops of the envelope
  double model_us = 0.0;         // the chain model of the plan
};
// g: the eliminated points' observations over the real cameras; x: the reduced system's structure (== g without retained points), whose
// pseudo-cameras always go to the border.  npad: padded order of the reduced system (x.C cameras).
static CameraOrderPlan plan_camera_order(const Problem& p, const std::vector<int>& cam_block, const CamGraph& g, const CamGraph& x,
                                         int npad, bool with_memory_order, bool border_ok, int border_mode) {
  CameraOrderPlan out;
  const int nblk = npad / 128, C = g.C, Cx = x.C;
  const bool forced = Cx > C;
  const std::vector<std::vector<int>> cand = camera_order_candidates(p, cam_block, *g.ocam, *g.opt, C, g.P, with_memory_order);
  std::vector<int> best_env;
  int best_k = 0;
  double best = 0.0;
  if (!forced) choose_camera_order(cand, *g.ocam, *g.opt, C, g.P, npad, &best_k, &best_env, &best);
  if (border_ok || forced) {
    // every candidate order may hide a band behind a few revisits: the border is tried on each, the chain model compares
    double best_us = 0.0;
    // (the candidates' borders side by side on host threads; which one is taken: in the candidates' order, below)
    std::vector<BorderChoice> bcs(cand.size());
    std::vector<char> bc_ok(cand.size(), 0);
    // the cameras of every point once, in the numbering the graphs come in: a candidate order's lists are these, renumbered and sorted again
    // point by point (no counting sort over every observation per candidate)
    PointLists gl0, xl0;
    point_camera_lists(g, &gl0.start, &gl0.cam);
    if (forced) point_camera_lists(x, &xl0.start, &xl0.cam);
    auto renumbered = [&](const PointLists& l0, const std::vector<int>& id, PointLists* out) {
      out->start = l0.start;
      out->cam.resize(l0.cam.size());
      for (size_t i = 0; i < l0.cam.size(); ++i) { const int c = l0.cam[i]; out->cam[i] = c < C ? id[(size_t)c] : c; }  // (pseudo-cameras keep their places behind the real ones)
      const int P = (int)l0.start.size() - 1;
      for (int q = 0; q < P; ++q) std::sort(out->cam.begin() + l0.start[(size_t)q], out->cam.begin() + l0.start[(size_t)q + 1]);
    };
    plan_parallel_for((int)cand.size(), [&](int ki) {
      const size_t k = (size_t)ki;
      if (k == 1 && !with_memory_order) return;  // (the slot repeats candidate 0)
      PointLists glk, xlk;
      renumbered(gl0, cand[k], &glk);
      if (forced) renumbered(xl0, cand[k], &xlk);
      const CamGraph gk{nullptr, nullptr, C, g.P}, xk{nullptr, nullptr, Cx, x.P};  // (choose_border reads the lists alone)
      std::vector<int> plain;
      if (!forced) plain = (int)k == best_k ? best_env : envelope_of_order(*g.ocam, *g.opt, cand[k], C, g.P, nblk);
      bc_ok[k] = choose_border(gk, xk, nblk, plain, border_mode, border_ok, &bcs[k], &glk, forced ? &xlk : nullptr) ? 1 : 0;
    });
    for (size_t k = 0; k < cand.size(); ++k) {
      if (k == 1 && !with_memory_order) continue;
      if (!bc_ok[k]) continue;
      const BorderChoice& bc = bcs[k];
      if (!out.bordered || bc.model_us < best_us) {
        best_us = bc.model_us; out.bordered = true; out.candidate = (int)k;
        out.border = bc;
        for (int c = 0; c < Cx; ++c) out.border.new_id[c] = bc.new_id[c < C ? cand[k][c] : c];  // first-appearance numbering -> final numbering
      }
    }
    // (against the envelope of the order that would be used otherwise)
    if (!forced) {
      const double plain_us = envelope_model_us(nblk, best_env, nullptr);
      if (out.bordered && border_mode != SK_BORDER_ON && out.border.model_us >= 0.9 * plain_us) out.bordered = false;
    }
  }
  if (!forced) { out.border.plain_us = envelope_model_us(nblk, best_env, nullptr); out.plain_id = cand[best_k]; }
  out.candidates = cand;
  if (out.bordered) {
    out.id = out.border.new_id; out.last = out.border.last; out.tail = out.border.tail;
    out.flops = cholesky_syrk_flops(npad, 1, out.last.data(), false, nullptr, -1, 1, out.tail.data());
    out.model_us = out.border.model_us;
  } else {
    out.id = cand[best_k]; out.last = best_env; out.candidate = best_k; out.flops = best;
    out.model_us = out.border.plain_us;
  }
  return out;
}

// ---- retained points (round 4): the few points with the longest tracks stay IN the reduced system ---------------------------------
// The Schur complement of a point seen by k cameras is a dense k x k square of camera blocks.  A landmark that stays in view for
// hundreds of frames — five such points among the 156 502 of the Ladybug-shaped problem — sets the height of the block envelope for
// every block column it spans (there: 25-55 block rows where the other points need 8-19; 166 of the 176 GFlop of the
// factorisation).  Such a point is not eliminated: its three coordinates stay in the reduced system as three more rows, behind the
// cameras — [S W; W^T T] (y_c; y_p) = (g_c; g_p) with S, g_c formed from the other points alone, W = F^T E (9 x 3 per observation),
// T = sum E^T E + D_p^2 — which is a BORDER in the sense of the loop-closure cameras above: rows that are active from the first
// camera that sees the point.  Three retained points share a pseudo-camera (nine rows), so that every layout of the reduced
// system — the bordered envelope, the fronts of a dissection — takes them as they take cameras.  The step is the same linear
// system's solution (EX/SimpleBundleAdjuster.scala:147-152: the result of DENSE_SCHUR does not depend on which unknowns were
// eliminated first); tests hold it against the all-eliminated order and the oracle.
// Which points: by the span of their cameras in the banded numbering, widest first, in steps of 3, 6, 12, ... as long as the chain
// model of the bordered envelope improves; taken when it predicts 10 % less than the plan without them (SK_RETAINED_ON: the best
// count whatever the model says).
struct RetainedChoice {
  std::vector<int> points;   // point ids, three to a pseudo-camera, pseudo-cameras in index order
  double model_us = 0.0;
};
// the two graphs of a problem whose points `points` (slot s -> pseudo-camera C + s / 3) are retained
struct RetainedGraphs {
  std::vector<int> ocam_g, opt_g, ocam_x, opt_x;
  int Cx = 0, Px = 0;
  CamGraph g(int C, int P) const { return CamGraph{&ocam_g, &opt_g, C, P}; }
  CamGraph x() const { return CamGraph{&ocam_x, &opt_x, Cx, Px}; }
};
static RetainedGraphs retained_graphs(const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P, const std::vector<int>& points) {
  RetainedGraphs r;
  std::vector<int> slot(P, -1);
  for (size_t s = 0; s < points.size(); ++s) slot[points[s]] = (int)s;
  r.Cx = C + ((int)points.size() + 2) / 3; r.Px = P;
  r.ocam_g.reserve(ocam.size()); r.opt_g.reserve(ocam.size()); r.ocam_x.reserve(ocam.size() + ocam.size() / 8); r.opt_x.reserve(ocam.size() + ocam.size() / 8);
  for (size_t b = 0; b < ocam.size(); ++b) {
    const int s = slot[opt[b]];
    if (s < 0) { r.ocam_g.push_back(ocam[b]); r.opt_g.push_back(opt[b]); r.ocam_x.push_back(ocam[b]); r.opt_x.push_back(opt[b]); continue; }
    r.ocam_x.push_back(ocam[b]); r.opt_x.push_back(r.Px);
    r.ocam_x.push_back(C + s / 3); r.opt_x.push_back(r.Px);
    ++r.Px;
  }
  return r;
}
// ocam: cameras in the banded numbering (the best candidate order, no border).  base_us: the model of the plan without retained points.
// families: bit 0 — the widest tracks by span and by number of observations, in doubling counts; bit 1 — the tracks of loop closures (below)
static RetainedChoice choose_retained_points(const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P, int mode, int max_points, bool gaps_ok,
                                             double base_us, int families = 3) {
  RetainedChoice out;
  if (mode == SK_RETAINED_OFF) return out;
  if (mode == SK_RETAINED_AUTO && (C < 64 || (9 * C + 128) / 128 < 16)) return out;  // (a reduced system of a few blocks: nothing to gain)
  std::vector<int> cmin(P, C), cmax(P, -1), cnt(P, 0);
  for (size_t b = 0; b < ocam.size(); ++b) { cmin[opt[b]] = std::min(cmin[opt[b]], ocam[b]); cmax[opt[b]] = std::max(cmax[opt[b]], ocam[b]); cnt[opt[b]]++; }
  // (a point with two residual blocks on one camera is never retained: the rows of a retained point have one writer per block)
  // (the cameras of every point, ascending — formed once: the candidates' graphs are derived from these lists, below — and a camera that
  // comes twice in a point's list is two residual blocks on one pair)
  PointLists base;
  { const CamGraph g0{&ocam, &opt, C, P}; point_camera_lists(g0, &base.start, &base.cam); }
  std::vector<char> twice(P, 0);
  for (int q = 0; q < P; ++q)
    for (int k = base.start[(size_t)q] + 1; k < base.start[(size_t)q + 1]; ++k) if (base.cam[(size_t)k] == base.cam[(size_t)k - 1]) twice[(size_t)q] = 1;
  std::vector<int> wide;
  for (int q = 0; q < P; ++q) if (cnt[q] >= 2 && 9 * (cmax[q] - cmin[q]) >= 128 && !twice[q]) wide.push_back(q);
  const bool exactly = mode == SK_RETAINED_ON && max_points > 0;  // (ON with a count: that many, as far as there are candidates)
  if (max_points <= 0) max_points = 1536;
  max_points = std::min(max_points - max_points % 3, (int)wide.size() - (int)wide.size() % 3);
  double best = mode == SK_RETAINED_ON ? std::numeric_limits<double>::max() : 0.9 * base_us;
  // Two orders of the candidates: by the span of their cameras (whatever widens the envelope: landmarks AND the tracks of loop
  // closures, which a border of retained points can take as well as a border of cameras can), and by the number of their
  // observations (the landmarks alone — the loop closures are then left to the border of cameras, when the problem has both)
  // One candidate set of points -> the chain model of the plan with them retained (a border of their pseudo-cameras, loop-closure cameras on top
  // where that pays).  Memoised, and evaluated a few candidates AHEAD in parallel: the loops below take them in their own order.
  struct Scored { bool ok = false; double model_us = 0.0; };
  std::map<std::vector<int>, Scored> scored;  // (key: the set, sorted)
  // the cameras of every point, once; a candidate's graphs are these lists with the retained points' taken out (g) and, behind them, a point
  // of two cameras — the observation's and the retained point's pseudo-camera — for every observation of a retained point (x: retained_graphs)
  auto lists_with = [&](const std::vector<int>& pts, PointLists* l, int* Cx, int* Px) {
    std::vector<int> slot((size_t)P, -1);
    for (size_t k = 0; k < pts.size(); ++k) slot[(size_t)pts[k]] = (int)k;
    int extra = 0;
    for (int q : pts) extra += base.start[(size_t)q + 1] - base.start[(size_t)q];
    *Cx = C + ((int)pts.size() + 2) / 3; *Px = P + extra;
    l->start.assign((size_t)*Px + 1, 0);
    l->cam.clear(); l->cam.reserve(base.cam.size() + (size_t)extra);
    for (int q = 0; q < P; ++q) {
      if (slot[(size_t)q] < 0) l->cam.insert(l->cam.end(), base.cam.begin() + base.start[(size_t)q], base.cam.begin() + base.start[(size_t)q + 1]);
      l->start[(size_t)q + 1] = (int)l->cam.size();
    }
    int np = P;
    for (int q : pts)
      for (int k = base.start[(size_t)q]; k < base.start[(size_t)q + 1]; ++k) {
        l->cam.push_back(base.cam[(size_t)k]); l->cam.push_back(C + slot[(size_t)q] / 3);
        l->start[(size_t)++np] = (int)l->cam.size();
      }
  };
  auto key_of = [](const std::vector<int>& pts) { std::vector<int> k(pts); std::sort(k.begin(), k.end()); return k; };
  auto score_ahead = [&](const std::vector<std::vector<int>>& sets) {
    std::vector<const std::vector<int>*> todo;
    std::vector<std::vector<int>> keys;
    for (const std::vector<int>& pts : sets) {
      std::vector<int> k = key_of(pts);
      if (scored.count(k) || std::find(keys.begin(), keys.end(), k) != keys.end()) continue;
      keys.push_back(std::move(k)); todo.push_back(&pts);
    }
    std::vector<Scored> res(todo.size());
    plan_parallel_for((int)todo.size(), [&](int i) {
      PointLists l;
      int Cx = 0, Px = 0;
      lists_with(*todo[i], &l, &Cx, &Px);
      const int nblk = (9 * Cx + 1 + 127) / 128;
      BorderChoice bc;
      // (g's lists are the first P of x's: one object serves as both)
      res[i].ok = choose_border(CamGraph{nullptr, nullptr, C, P}, CamGraph{nullptr, nullptr, Cx, Px}, nblk, {}, SK_BORDER_AUTO, gaps_ok, &bc, &l, &l);
      res[i].model_us = bc.model_us;
    });
    for (size_t i = 0; i < todo.size(); ++i) scored[keys[i]] = res[i];
  };
  auto score = [&](const std::vector<int>& pts) { std::vector<int> k = key_of(pts); if (!scored.count(k)) score_ahead({pts}); return scored[k]; };
  // the candidate sets of one order of the wide tracks, in doubling counts
  auto sets_of_order = [&](std::vector<std::vector<int>>* sets) {
    for (int R = exactly ? std::max(3, max_points) : 3; R <= max_points; R = R < 6 ? 6 : 2 * R) {
      std::vector<int> pts(wide.begin(), wide.begin() + R);
      // pseudo-cameras in the order the border wants them: the points reached first come last
      std::sort(pts.begin(), pts.end(), [&](int a, int b) { return cmin[a] != cmin[b] ? cmin[a] > cmin[b] : a < b; });
      sets->push_back(std::move(pts));
    }
  };
  std::vector<std::vector<int>> seen_sets;
  for (int by_count = 0; (families & 1) && by_count < (exactly ? 1 : 2); ++by_count) {
    std::sort(wide.begin(), wide.end(), [&](int a, int b) {
      const int sa = by_count ? cnt[a] : cmax[a] - cmin[a], sb = by_count ? cnt[b] : cmax[b] - cmin[b];
      return sa != sb ? sa > sb : a < b;
    });
    double best_here = std::numeric_limits<double>::max();
    std::vector<std::vector<int>> sets;
    sets_of_order(&sets);
    for (size_t si = 0; si < sets.size(); ++si) {
      const std::vector<int>& pts = sets[si];
      const int R = (int)pts.size();
      // (the same set of points under the other order — the landmarks are usually the widest tracks by either measure — is not planned twice:
      // a candidate of Venice-1778's size costs most of a second)
      { std::vector<int> key = key_of(pts); if (std::find(seen_sets.begin(), seen_sets.end(), key) != seen_sets.end()) continue; seen_sets.push_back(key); }
      // (this candidate and the next three: the loop usually ends — 25 % past its best — within a few counts of where it is)
      if (!scored.count(key_of(pts))) score_ahead(std::vector<std::vector<int>>(sets.begin() + (long)si, sets.begin() + (long)std::min(sets.size(), si + 4)));
      const Scored sc = score(pts);
      if (!sc.ok) continue;
      struct { double model_us; } bc{sc.model_us};
      if (dev_knobs().debug_envelope) std::fprintf(stderr, "[skeres_amd] retained candidates: the %d widest tracks by %s: chain model %.0f us (best so far %.0f, base %.0f)\n", R, by_count ? "observations" : "span", bc.model_us, best, base_us);
      // (a LARGER set has to beat a smaller one by 1 %: the model is no finer than that, and every retained point is three more rows that
      // every later column carries.  Ladybug-1723, one box, Cholesky phase per iteration with 6 / 12 / 24 points retained: 4.06 / 3.82 /
      // 3.92 ms, where the model says 5362 / 5304 / 5303 us — profiles/r05_retained_count_ab.txt)
      if (bc.model_us < (out.points.empty() || pts.size() <= out.points.size() ? 1.0 : 0.99) * best) { best = bc.model_us; out.points = pts; out.model_us = bc.model_us; }
      if (bc.model_us > 1.25 * best_here) break;  // (well past the best count of this order: more border rows only cost)
      best_here = std::min(best_here, bc.model_us);
    }
  }
  // A third family (round 5): the tracks of LOOP CLOSURES — points whose ascending camera list has a jump of more than `gap` cameras, the
  // test choose_border applies to cameras — ALL of them, at their exact number, then the widest of the other tracks behind them.  The
  // doubling counts above cannot find this set: 0.5 % of the Ladybug-shaped problem's tracks seen from two distant windows are 782 points,
  // whose rows keep the envelope full until the last one of them is retained (768 points: nothing gained) and cost twice their rows at
  // the next count (1536); retained exactly — 19 block rows of border — the band keeps its own width.
  if (!exactly && C >= 64 && (families & 2)) {
    const std::vector<int>& pstart = base.start;  // (the cameras of every point, ascending: formed above)
    const std::vector<int>& pcam = base.cam;
    std::vector<int> jump(P, 0);
    int max_jump = 0;
    for (int q = 0; q < P; ++q) {
      for (int k = pstart[q] + 1; k < pstart[q + 1]; ++k) jump[q] = std::max(jump[q], pcam[k] - pcam[k - 1]);
      max_jump = std::max(max_jump, jump[q]);
    }
    std::vector<int> by_cnt(wide);
    std::sort(by_cnt.begin(), by_cnt.end(), [&](int a, int b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
    size_t prev_size = 0;
    for (int gap = 16; gap < C && gap < max_jump; gap *= 2) {
      std::vector<int> closing;
      for (int q : wide) if (jump[q] > gap) closing.push_back(q);
      if (closing.empty()) break;
      if (dev_knobs().debug_envelope) std::fprintf(stderr, "[skeres_amd] retained candidates: %zu wide tracks with a jump of more than %d cameras (max_points %d)\n", closing.size(), gap, max_points);
      if ((int)closing.size() > max_points || closing.size() == prev_size) continue;
      prev_size = closing.size();
      std::vector<char> in(P, 0);
      for (int q : closing) in[q] = 1;
      std::vector<std::vector<int>> both;
      for (int more : {12, 48}) {  // ... and a few of the landmarks on top (by their number of observations)
        std::vector<int> pts(closing);
        int added = 0;
        for (size_t k = 0; k < by_cnt.size() && (added < more || pts.size() % 3 != 0); ++k)
          if (!in[by_cnt[k]]) { pts.push_back(by_cnt[k]); ++added; }
        if (pts.size() % 3 != 0 || (int)pts.size() > max_points + 2) { both.emplace_back(); continue; }
        std::sort(pts.begin(), pts.end(), [&](int a, int b) { return cmin[a] != cmin[b] ? cmin[a] > cmin[b] : a < b; });
        both.push_back(std::move(pts));
      }
      { std::vector<std::vector<int>> ahead; for (const auto& v : both) if (!v.empty()) ahead.push_back(v); score_ahead(ahead); }
      for (int mi = 0; mi < 2; ++mi) {
        const int more = mi == 0 ? 12 : 48;
        const std::vector<int>& pts = both[mi];
        if (pts.empty()) continue;
        const Scored sc = score(pts);
        if (!sc.ok) continue;
        struct { double model_us; } bc{sc.model_us};
        if (dev_knobs().debug_envelope) std::fprintf(stderr, "[skeres_amd] retained candidates: gap %d + %d landmarks = %zu points: chain model %.0f us (best so far %.0f, base %.0f)\n", gap, more, pts.size(), bc.model_us, best, base_us);
        if (bc.model_us < (out.points.empty() || pts.size() <= out.points.size() ? 1.0 : 0.99) * best) { best = bc.model_us; out.points = pts; out.model_us = bc.model_us; }
      }
    }
  }
  return out;
}

// The layout of the reduced system as setup() takes it: the camera order (with its border of loop-closure cameras), and — when
// retained_mode allows and the chain model agrees — the retained points with their pseudo-cameras.  From host data alone.
struct ReducedSystemPlan {
  CameraOrderPlan order;            // over the real cameras and the pseudo-cameras
  std::vector<int> retained;        // the retained points (three to a pseudo-camera, in pseudo-camera order); empty: every point is eliminated
  RetainedGraphs graphs;            // ... and the structure with them (first-appearance numbering)
  double without_us = 0.0;          // the chain model of the plan with every point eliminated
};
static ReducedSystemPlan plan_reduced_system(const Problem& p, const std::vector<int>& cam_block, const std::vector<int>& ocam, const std::vector<int>& opt, int C, int P,
                                             bool with_memory_order, bool border_ok, int border_mode, int retained_mode, int retained_max) {
  ReducedSystemPlan out;
  const CamGraph g0{&ocam, &opt, C, P};
  const int npad = ((9 * C + 1 + 127) / 128) * 128;
  out.order = plan_camera_order(p, cam_block, g0, g0, npad, with_memory_order, border_ok, border_mode);
  out.without_us = out.order.model_us;
  if (retained_mode == SK_RETAINED_OFF) return out;
  // The retained points are chosen on a banded numbering of the cameras: the best candidate order as it stands — and, round 5, the
  // other candidates too: with loop closures all over the sequence the order with the fewest flops is a reverse Cuthill-McKee one in
  // which every track has jumps, and the tracks that CLOSE loops (choose_retained_points, third family) can only be told from the
  // others in the capture order (memory order / first appearance).  The chain model of the whole plan compares.
  std::vector<std::vector<int>> bases;
  std::vector<int> families;
  const std::vector<std::vector<int>> cands = out.order.candidates;  // (of the same graph: formed once)
  bases.push_back(out.order.plain_id);
  families.push_back(1 | ((out.order.plain_id == cands[0] || out.order.plain_id == cands[1]) ? 2 : 0));
  for (int k = 0; k < 2; ++k)  // (the capture orders: first appearance, memory order)
    if (std::find(bases.begin(), bases.end(), cands[k]) == bases.end()) { bases.push_back(cands[k]); families.push_back(2); }
  std::vector<std::vector<int>> tried;
  const CameraOrderPlan without = out.order;
  for (size_t bi = 0; bi < bases.size(); ++bi) {
    const std::vector<int>& base = bases[bi];
    std::vector<int> oc(ocam.size());
    for (size_t b = 0; b < ocam.size(); ++b) oc[b] = base[ocam[b]];
    const RetainedChoice rc = choose_retained_points(oc, opt, C, P, retained_mode, retained_max, border_ok, without.model_us, families[bi]);
    if (rc.points.empty()) continue;
    std::vector<int> key(rc.points);
    std::sort(key.begin(), key.end());
    if (std::find(tried.begin(), tried.end(), key) != tried.end()) continue;
    tried.push_back(key);
    RetainedGraphs rg = retained_graphs(ocam, opt, C, P, rc.points);
    const int npadx = ((9 * rg.Cx + 1 + 127) / 128) * 128;
    CameraOrderPlan px = plan_camera_order(p, cam_block, rg.g(C, P), rg.x(), npadx, with_memory_order, border_ok, border_mode);
    const bool first = out.retained.empty();
    if (first ? (retained_mode == SK_RETAINED_ON || px.model_us < 0.9 * without.model_us) : px.model_us < out.order.model_us) {
      out.order = std::move(px); out.retained = rc.points; out.graphs = std::move(rg);
    }
    if (retained_mode == SK_RETAINED_ON && retained_max > 0) break;  // (an exact count: the widest tracks of the plan's own order)
  }
  return out;
}

// Sharding the points pays when the per-iteration work it removes from a rank (evaluation, Schur
// assembly, back-substitution: linear in observations and pair entries) exceeds the all-reduce of the
// reduced system it adds.  The all-reduce is MEASURED here (second and third call of the hook on the real
// buffer); the work is estimated from constants measured on MI355X (profiles/r01_c_*).  All ranks
// take the same decision: the measured times are averaged over ranks through the hook itself.
// In replicated mode every rank solves the whole problem with no collective at all (the results are
// bitwise those of one GPU); the speed-up is then 1, which for a small reduced system beats < 1.
int BalSolver::choose_distribution(const std::vector<int>& opt) {
  const int W = opt_.world;
  distribution_ = SK_DISTRIBUTION_SHARDED;
  if (W <= 1 || opt_.distribution_mode == SK_DISTRIBUTION_SHARDED) return SK_OK;
  if (opt_.distribution_mode != SK_DISTRIBUTION_REPLICATED) {
    std::vector<size_t> k(P_total_, 0);
    for (int v : opt) k[v]++;
    double pairs = 0.0;
    for (size_t v : k) pairs += 0.5 * (double)v * (double)(v - 1);
    // seconds on one MI355X of the phases that shard with the points (Jacobians, Schur assembly, back-substitution, candidate cost).  Round 5:
    // recalibrated on the bench records — Ladybug-1723 0.66 ms (679 k observations, 2.4 M pair entries), Venice-1778 3.65 ms (5.0 M, 25 M);
    // round 1's constants (0.15 ns per pair entry + 1.9 ns per observation) were those kernels three rounds ago
    const double per_iter = 0.03e-9 * pairs + 0.6e-9 * (double)opt.size();
    est_saved_s_ = per_iter * (1.0 - 1.0 / W);
    int rc = allreduce(b_pack_.p, packed_elems_);  // first call: connection set-up, not timed
    if (rc) return rc;
    SK_HIP_TRY(hipStreamSynchronize(stream_));
    const auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 2; ++rep) { rc = allreduce(b_pack_.p, packed_elems_); if (rc) return rc; }
    SK_HIP_TRY(hipStreamSynchronize(stream_));
    double mine = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 2.0;
    double vals[1] = {mine};
    const int ops[1] = {0};
    SK_HIP_TRY(b_small_.alloc(2 * 9 * (size_t)C_ + 6 * retained_pts_.size() + 64 + 16 * (size_t)W));
    rc = gather_rank_scalars(vals, 1, ops);
    if (rc) return rc;
    est_allreduce_s_ = vals[0] / W;
    collect_allreduce_time(true);
    phase_[5] = 0.0;  // the probe is set-up, not an iteration phase
    if (est_allreduce_s_ <= est_saved_s_) return SK_OK;  // sharding pays
  }
  distribution_ = SK_DISTRIBUTION_REPLICATED;
  opt_.allreduce = nullptr; opt_.world = 1; opt_.rank = 0;
  b_pack_.release();
  return SK_OK;
}

int BalSolver::setup() {
  // (SK_DEBUG=setup: where the set-up's wall time goes, one line per stage on stderr)
  const auto setup_t0 = std::chrono::steady_clock::now();
  auto stage = [&, last = setup_t0](const char* what) mutable {
    if (!dev_knobs().debug_setup) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[skeres_amd] set-up: %-40s %8.1f ms (at %.1f ms)\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
                 std::chrono::duration<double, std::milli>(now - setup_t0).count());
    last = now;
  };
  std::string why;
  if (!problem_is_bal_shaped(*problem_, &why)) { set_error("%s", why.c_str()); return SK_ERR_UNSUPPORTED; }
  const Problem& p = *problem_;
  for (size_t b = 0; b < p.block_param.size(); ++b)
    if (p.block_param[b] >= 0) {
      const int t = p.params[p.block_param[b]].type;
      if (t != kParamIdentity && t != kParamSubset) {
        set_error("DENSE_SCHUR takes identity and subset parameterizations and constant parameter blocks (quaternion / homogeneous-vector blocks are implemented for DENSE_QR / DENSE_NORMAL_CHOLESKY; not supported here)");
        return SK_ERR_UNSUPPORTED;
      }
    }
  const int Nall = (int)p.rb_functor.size();
  (void)bal_block_shape(p, &res_size_, &cam_size_, &pt_size_);
  std::vector<int> ocam, opt;
  bal_index_problem(p, &cam_block_, &pt_block_, &ocam, &opt);
  C_ = (int)cam_block_.size(); P_total_ = (int)pt_block_.size();
  const int Creal = C_;  // (C_ grows by the pseudo-cameras of retained points, below)
  n_ = 9 * C_; rhs_row_ = n_; npad_ = ((n_ + 1 + 127) / 128) * 128;
  SK_HIP_TRY(cholesky_init());
  {
    // Launch-bound problems replay their iteration as a hipGraph on ONE stream (below): decided before the look-ahead
    // context exists, so that a reduced system of a few blocks pays neither the queue trial nor its 134 MB of scratch.
    bool host_or_tape = false;
    for (size_t b = 0; b < p.rb_functor.size() && !host_or_tape; ++b) host_or_tape = p.rb_functor[b] == SK_FUNCTOR_HOST_CALLBACK || p.tape_of_block(b) != nullptr;
    graph_mode_ = npad_ / 128 <= 8 && !opt_.allreduce && !host_or_tape && opt_.dissection != SK_DISSECTION_ON && dev_knobs().dissect_at < 0 && opt_.graph_replay;
    if (graph_mode_) opt_.lookahead = false;  // one stream: the whole iteration is one in-order launch sequence
  }
  chol_ctx_.resident = chol_ctx_b_.resident = opt_.resident_kernels;
  if (opt_.lookahead && chol_ctx_.init() != hipSuccess) {  // CU-masked streams unavailable: plain in-order factorisation
    (void)hipGetLastError();
    opt_.lookahead = false;
  }
  // The device's queue trial (once per device) BEFORE the plans below are chosen: on a device that cannot run the resident panel chain —
  // shared with another process, its kernels serialised — the trial says so, and the lock-step dissection, which exists for that chain,
  // is then not chosen (until round 4 the trial ran after the layout was fixed: two processes sharing one device took the dissected plan
  // launch by launch, 1.9 s per iteration where the undissected one takes 0.2)
  if (opt_.lookahead) cholesky_prepare(&chol_ctx_, stream_);
  if (opt_.allreduce && opt_.world > 1) {
    // the ranks must factor by ONE plan and take ONE distribution decision: a rank without CU-masked streams (no
    // look-ahead, hence no resident chain and no dissection) takes every rank there
    SK_HIP_TRY(b_small_.alloc(2 * 9 * ((size_t)C_ + (size_t)(std::max(1536, opt_.retained_max) + 12) / 3) + 6 * (size_t)(std::max(1536, opt_.retained_max) + 12) + 64 + 16 * (size_t)opt_.world));  // (room for the pseudo-cameras and the retained points' sums the plan below may add)
    double off[1] = {opt_.lookahead ? 0.0 : 1.0};
    int rc = gather_rank_scalars_signed(off, 1);
    if (rc) return rc;
    if (off[0] > 0.0) opt_.lookahead = false;
  }
  std::vector<int> env_for_model;  // the envelope of the chosen order (whether or not it is then used)
  std::vector<int> band_ocam, band_opt;  // retained points: the eliminated points' observations over the real cameras, final numbering (the band a dissection cuts)
  // ---- camera order + block envelope of the reduced system (all ranks' observations: the all-reduced S has the union structure).
  // The order is chosen the same way whether or not the envelope is then used (opt_.envelope), so that the two
  // settings differ in nothing but the blocks they skip and give bit-identical results. ----
  {
    const int nblk = npad_ / 128;
    std::vector<int> best_env;
    double best = 0.0;
    // The memory order of the camera blocks (the BAL file's numbering under the reference's layout) is usually the best
    // candidate by far — and host addresses are a process's own: with separately allocated camera blocks it could differ from
    // rank to rank, and the ranks must build the SAME reduced system.  Round 3: the ranks try it and compare (a hash of the
    // order and the envelope: one tiny exchange); only if they disagree do they all fall back to the rank-invariant candidates.
    // (Until then a world of ranks never used it: on the Ladybug-shaped problem the chain model of the best remaining order,
    // reverse Cuthill-McKee, is 12.1 ms against 9.1 — every multi-rank run would have factored a third more slowly.)
    // (the border of loop-closure cameras: not with an explicit dissection or segmentation — the fronts of those have borders of
    // their own kind — and only inside the envelope machinery)
    RetainedGraphs rgraphs;
    // (an explicitly SEGMENTED world takes a border of loop-closure cameras — its members join the one separator — when it is cut in TWO:
    // sk_options_set_max_segments(o, 2); retained points it takes with any number of segments: their pseudo-cameras are a border of the root)
    const bool many_segments = opt_.allreduce && opt_.world > 1 && opt_.distribution_mode == SK_DISTRIBUTION_SEGMENTED && opt_.max_segments != 2;
    const bool border_ok = opt_.envelope && opt_.border != SK_BORDER_OFF && opt_.dissection != SK_DISSECTION_ON && dev_knobs().dissect_at < 0 && !many_segments;
    // (retained points: not with an explicit dissection; a launch-bound problem under hipGraph replay has nothing to gain)
    const bool retained_ok = opt_.retained != SK_RETAINED_OFF && opt_.dissection != SK_DISSECTION_ON && dev_knobs().dissect_at < 0 && !graph_mode_;
    CameraOrderPlan plan;
    auto pick = [&](bool with_memory_order) {
      ReducedSystemPlan rp = plan_reduced_system(p, cam_block_, ocam, opt, Creal, P_total_, with_memory_order, border_ok, opt_.border,
                                                 retained_ok ? opt_.retained : SK_RETAINED_OFF, opt_.retained_max);
      plan = std::move(rp.order);
      retained_pts_ = rp.retained; rgraphs = std::move(rp.graphs);
      retained_without_us_ = rp.without_us; retained_model_us_ = retained_pts_.empty() ? 0.0 : plan.model_us;
      unsigned long long h = 1469598103934665603ull;
      for (int v : retained_pts_) { h ^= (unsigned)v; h *= 1099511628211ull; }
      for (int v : plan.id) { h ^= (unsigned)v; h *= 1099511628211ull; }
      for (int v : plan.last) { h ^= (unsigned)v; h *= 1099511628211ull; }
      for (int v : plan.tail) { h ^= (unsigned)v; h *= 1099511628211ull; }
      order_hash_ = (double)(h >> 12);  // 52 bits: exact in a double
    };
    stage("structure, queue trial");
    pick(true);
    stage("plan of the reduced system");
    if (opt_.allreduce && opt_.world > 1) {
      double v[2] = {order_hash_, -order_hash_};
      int rc = gather_rank_scalars_signed(v, 2);
      if (rc) return rc;
      if (v[0] != order_hash_ || v[1] != -order_hash_) pick(false);  // (every rank sees the disagreement: max and min differ)
    }
    const std::vector<int>& id = plan.id;
    camera_order_ = plan.candidate;
    best_env = plan.last; best = plan.flops;
    if (!retained_pts_.empty()) {
      // the reduced system has a pseudo-camera for every three retained points: C_ counts them from here on (cam_block_: -1)
      pseudo_cams_ = rgraphs.Cx - Creal;
      C_ = rgraphs.Cx;
      n_ = 9 * C_; rhs_row_ = n_; npad_ = ((n_ + 1 + 127) / 128) * 128;
      struct_ocam_ = rgraphs.ocam_x; struct_opt_ = rgraphs.opt_x; struct_P_ = rgraphs.Px;
      for (int& c : struct_ocam_) c = id[c];
      band_ocam = rgraphs.ocam_g; band_opt = rgraphs.opt_g;
      for (int& c : band_ocam) c = id[c];
      retained_cam_.resize(retained_pts_.size());
      for (size_t k = 0; k < retained_pts_.size(); ++k) retained_cam_[k] = id[Creal + (int)k / 3];
    }
    std::vector<int> cb(C_, -1);
    for (int c = 0; c < Creal; ++c) cb[id[c]] = cam_block_[c];
    cam_block_.swap(cb);
    for (int& c : ocam) c = id[c];
    const double full = cholesky_syrk_flops(npad_, 1, nullptr);
    border_plain_us_ = plan.border.plain_us;
    if (plan.bordered) {
      env_tail_ = plan.tail;
      border_cams_ = plan.border.border_cams; border_gap_ = plan.border.gap;
      border_model_us_ = plan.border.model_us;
    }
    group_ = opt_.group_or(opt_.envelope && best < 0.5 * full ? 1 : 3);
    env_for_model = best_env;
    if (opt_.envelope) env_last_.swap(best_env);
    else env_tail_.clear();  // (retained points make a border with or without the envelope: without it every block is factored)
    if (dev_knobs().debug_envelope && opt_.envelope) {
      long h = 0;
      for (int c = 0; c < nblk; ++c) h += env_last_[c] - c;
      std::fprintf(stderr, "[skeres_amd] camera order %d (0 first appearance, 1 memory, 2 RCM); envelope: %d block columns, mean height %.1f; "
                   "trailing-update flops %.3e (full %.3e)\n", camera_order_, nblk, (double)h / nblk, best, full);
      if (border_cams_ > 0)
        std::fprintf(stderr, "[skeres_amd] loop closures: %d cameras in a trailing border (visits split at jumps of more than %d cameras): chain model %.0f us against %.0f\n",
                     border_cams_, border_gap_, border_model_us_, border_plain_us_);
    }
  }
  stage("camera order applied");
  // ---- multi-GPU: shard the points, or replicate? ----
  // What travels in the all-reduce of the reduced system is the part of its lower block triangle INSIDE the envelope:
  // block row kb from the first block column that reaches it (the right-hand-side row whole) — 0.36 GB instead of 0.98 GB
  // on the Ladybug-1723-shaped system, exactly the blocks the assembly can write.
  {
    const int nblk = npad_ / 128;
    std::vector<int> pack_col0(nblk, 0);
    std::vector<long long> pack_off(nblk + 1, 0);
    if (!env_last_.empty()) pack_col0 = cholesky_row_first_cols(nblk, env_last_.data(), env_tail_.empty() ? nullptr : env_tail_.data());
    for (int kb = 0; kb < nblk; ++kb) pack_off[kb + 1] = pack_off[kb] + (long long)128 * 128 * (kb + 1 - pack_col0[kb]);
    packed_elems_ = (size_t)pack_off[nblk];
    pack_col0_h_ = pack_col0; pack_off_h_ = pack_off;
  }
  // the buffer the reduced system travels in (the caller's, or our own)
  auto prepare_pack = [&]() -> int {
    if (opt_.reduce_buffer) {
      if (opt_.reduce_buffer_bytes < packed_elems_ * sizeof(double)) { set_error("reduce buffer too small: need %zu bytes", packed_elems_ * sizeof(double)); return SK_ERR_INVALID_ARGUMENT; }
      b_pack_.adopt(static_cast<double*>(opt_.reduce_buffer), packed_elems_);
    } else {
      SK_HIP_TRY(b_pack_.alloc(packed_elems_));
    }
    SK_HIP_TRY(b_pack_.zero(stream_));
    return SK_OK;
  };
  // Retained points rule out the segmented distribution (their rows couple with every segment), so a world of ranks decides HERE
  // between sharding the points and replicating the solve — before the dissection: a rank that replicates is a single device from
  // here on (choose_distribution), and takes the lock-step dissection a single device takes
  // Round 5: ... unless the world can take the sequence as TWO segments — head and tail on two ranks' devices, the retained points'
  // pseudo-cameras (and a border of loop-closure cameras) members of the one separator, exactly the fronts a single device holds side by
  // side: tried first (pass 0 below), against what one device would do with the lock-step plan.
  bool distribution_decided = false;
  auto shard_or_replicate = [&]() -> int {
    int rc = prepare_pack();
    if (rc) return rc;
    rc = choose_distribution(opt);
    if (rc) return rc;
    distribution_decided = true;
    return SK_OK;
  };
  const bool seg_modes = opt_.distribution_mode == SK_DISTRIBUTION_AUTO || opt_.distribution_mode == SK_DISTRIBUTION_SEGMENTED;
  if (opt_.allreduce && opt_.world > 1 && pseudo_cams_ > 0 && !seg_modes) {
    int rc = shard_or_replicate();
    if (rc) return rc;
  }
  for (int pass = 0; pass < 2; ++pass) {
    const int nblk = npad_ / 128;
    // ---- dissect?  One process: only when forced (measured not to pay on one chip).  Several ranks: the SEGMENTED
    // distribution — every rank's device eliminates one segment of the camera sequence — when the model of the chains
    // predicts a gain (or when asked for). ----
    // (one device: a border — the cameras of loop closures, the pseudo-cameras of retained points — joins the ONE separator, which both
    // fronts border on, and the band in front of it is what is cut; several ranks: a bordered system is not dissected)
    const bool multi = opt_.allreduce != nullptr && opt_.world >= 2;
    border_members_ = env_tail_.empty() ? 0 : border_cams_ + pseudo_cams_;
    // (a border of loop-closure cameras alone — every point eliminated — is left undissected, as until round 4: the band then keeps its
    // SYRK-bound block columns, which the lock-step cannot pair and the border's rows make dearer; measured on Ladybug-1723 with three
    // places revisited: 8.1-9.2 ms of Cholesky phase for three cuts against 8.2 undissected)
    // ... unless no block column of the bordered band is SYRK-bound to begin with (a sequence of a few hundred cameras)
    bool band_chain_bound = border_members_ > 0 && !env_for_model.empty() && !env_tail_.empty();
    for (int c = 0; band_chain_bound && c < nblk - 1 && c < (9 * (C_ - border_members_)) / 128; ++c) {
      const int lm = std::min(env_for_model[c], nblk - 1), main_rows = lm > c ? lm - c : 0;
      band_chain_bound = main_rows + std::max(0, nblk - std::max(env_tail_[c], c + 1 + main_rows)) <= 24;
    }
    const bool two_seg_try = multi && pseudo_cams_ > 0 && seg_modes && !distribution_decided;  // (see above)
    const bool pseudo_border = border_members_ > 0 && (!multi || two_seg_try) && (pseudo_cams_ > 0 || band_chain_bound);
    const int Cband = C_ - (pseudo_border ? border_members_ : 0);
    const bool plan_ok = opt_.dissection != SK_DISSECTION_OFF && opt_.envelope && opt_.lookahead && opt_.cholesky_group == 0 && (env_tail_.empty() || pseudo_border);
    bool may_dissect = plan_ok && (multi ? ((pseudo_cams_ == 0 || two_seg_try) && seg_modes)
                                         : (!opt_.allreduce && chol_ctx_b_.init_secondary(chol_ctx_) == hipSuccess));
    if (!may_dissect) (void)hipGetLastError();
    if (!plan_ok && opt_.distribution_mode == SK_DISTRIBUTION_SEGMENTED) {
      set_error("the segmented distribution needs the library's own factorisation plan (envelope, look-ahead, no explicit group); not supported with these options");
      return SK_ERR_UNSUPPORTED;
    }
    std::vector<int> cut_a, cut_b;  // the separators [a, b) in the banded numbering, ascending
    if (may_dissect && multi && !two_seg_try) {
      std::vector<int> first_col;
      (void)envelope_of_order(ocam, opt, [&] { std::vector<int> e(C_); std::iota(e.begin(), e.end(), 0); return e; }(), C_, P_total_, nblk, &first_col);
      int max_seg = opt_.world;
      if (opt_.max_segments >= 2) max_seg = std::min(max_seg, opt_.max_segments);  // (sk_options_set_max_segments)
      const Segments sg = choose_segments(ocam, opt, C_, P_total_, nblk, env_for_model, first_col, max_seg, opt_.distribution_mode == SK_DISTRIBUTION_SEGMENTED, opt_.world);
      cut_a = sg.a; cut_b = sg.b;
      dissect_t_plain_ = sg.t_plain; dissect_t_model_ = sg.t_model;
      for (int k = 0; k < 9; ++k) model_us_[k] = sg.model_us[k];
    } else if (may_dissect) {
      Dissection ds;
      std::vector<int> first_col, many_cut_a, many_cut_b;
      double dissect_t_model_many = 0.0;
      // what is cut: the cameras' band — with a border, the band cameras under the points that are eliminated (the border's cameras
      // and what they see belong to the separator whatever the cut)
      std::vector<int> cut_ocam, cut_opt;
      if (pseudo_border) {
        const std::vector<int>& so = pseudo_cams_ > 0 ? band_ocam : ocam;
        const std::vector<int>& sp = pseudo_cams_ > 0 ? band_opt : opt;
        for (size_t b = 0; b < so.size(); ++b) if (so[b] < Cband) { cut_ocam.push_back(so[b]); cut_opt.push_back(sp[b]); }
      }
      const std::vector<int>& docam = pseudo_border ? cut_ocam : ocam;
      const std::vector<int>& dopt = pseudo_border ? cut_opt : opt;
      const int dnblk = pseudo_border ? (9 * Cband + 1 + 127) / 128 : nblk;
      const std::vector<int> band_env = envelope_of_order(docam, dopt, [&] { std::vector<int> e(Cband); std::iota(e.begin(), e.end(), 0); return e; }(), Cband, P_total_, dnblk, &first_col);
      const std::vector<int>& denv = pseudo_border ? band_env : env_for_model;
      // one device: the lock-step schedule and its own cut
      // (only under the resident chain: the partner front rides in ITS launches — with SK_CHOL_CHAIN_SERVER=0, or on a device that
      // lost its chain, a single device stays undissected)
      const bool lockstep_cut = !multi && opt_.dissection == SK_DISSECTION_AUTO && opt_.resident_kernels && cholesky_chain_enabled(&chol_ctx_);
      // (the border's rows that a band column reaches on top of its run: the tail profile of the bordered envelope)
      std::vector<int> extra_fwd, extra_bwd_col;
      int extra_bwd = 0;
      if (pseudo_border) {
        extra_fwd.assign(dnblk, 0);
        for (int c = 0; c < dnblk && c < (int)env_tail_.size(); ++c) extra_fwd[c] = std::max(0, nblk - 1 - std::max(env_tail_[c], c + 1));
        // (the tail front reaches the border's rows in another order; counting all of them in every one of its columns moved the cut of
        // Venice-1778 to a worse place — Cholesky phase 5.26 against 4.94 ms — and counting none made the model of a border of 800 retained
        // points 40 % too low (round 5: 6.1 against 10.1 ms with the tracks of scattered loop closures retained).  Counted per column: a
        // member of the border is active in the tail's columns from the LAST band camera that sees it back to the cut.)
        const std::vector<int>& xo = struct_ocam_.empty() ? ocam : struct_ocam_;
        const std::vector<int>& xp = struct_ocam_.empty() ? opt : struct_opt_;
        const int xP = struct_ocam_.empty() ? P_total_ : struct_P_;
        std::vector<int> pmax(xP, -1), reach_max(C_ - Cband, -1);
        for (size_t b = 0; b < xo.size(); ++b) if (xo[b] < Cband) pmax[xp[b]] = std::max(pmax[xp[b]], xo[b]);
        for (size_t b = 0; b < xo.size(); ++b) if (xo[b] >= Cband) reach_max[xo[b] - Cband] = std::max(reach_max[xo[b] - Cband], pmax[xp[b]]);
        std::vector<int> active(dnblk + 1, 0);  // members whose last band camera lies in block column c or behind it
        for (int m : reach_max) if (m >= 0) active[std::min(dnblk - 1, (9 * m) / 128)]++;
        for (int c = dnblk - 2; c >= 0; --c) active[c] += active[c + 1];
        extra_bwd_col.assign(dnblk, 0);
        for (int c = 0; c < dnblk; ++c) extra_bwd_col[c] = (9 * active[c] + 127) / 128;
      }
      const std::vector<int> band_reach = camera_reach(docam, dopt, Cband, P_total_);  // (once for every cut planned on this sequence, below)
      ds = choose_dissection(docam, dopt, Cband, P_total_, dnblk, denv, first_col, lockstep_cut || two_seg_try, lockstep_cut, C_ - Cband, pseudo_border ? &extra_fwd : nullptr, extra_bwd,
                             pseudo_border ? &extra_bwd_col : nullptr, &band_reach);
      if (two_seg_try) {
        // two devices, a chain each, against ONE device with the two fronts in lock-step (what a replicating rank would run): + the
        // all-reduce of the separator's system (its lower triangle over one xGMI link per direction: choose_segments)
        const Dissection one = choose_dissection(docam, dopt, Cband, P_total_, dnblk, denv, first_col, true, true, C_ - Cband, pseudo_border ? &extra_fwd : nullptr, extra_bwd,
                                                 pseudo_border ? &extra_bwd_col : nullptr, &band_reach);
        const double one_us = one.a > 0 ? one.t_dissected : one.t_plain;
        const double E = ds.a > 0 ? (9.0 * (ds.b - ds.a + C_ - Cband) + 1.0 + 127.0) / 128.0 : 0.0;
        const int W = std::max(2, opt_.world);
        // (+ the three small collectives of a segmented iteration — column norms and gradient, two tables of scalars — at the ~40 us of a
        // latency-bound all-reduce each)
        const double allreduce_us = 50.0 + 3.0 * 40.0 + 2.0 * (W - 1.0) / W * 0.5 * E * (E + 1.0) * 128.0 * 128.0 * 8.0 / 153e3;
        // (the phases that shard with the points — two ranks take half of them each: choose_distribution's constants)
        double pairs = 0.0;
        { std::vector<int> k(P_total_, 0); for (int v : opt) k[v]++; for (int v : k) pairs += 0.5 * (double)v * (double)(v - 1); }
        const double shard_us = 1e6 * (0.03e-9 * pairs + 0.6e-9 * (double)opt.size());
        model_us_[1] = one_us; model_us_[2] = ds.a > 0 ? ds.t_dissected + allreduce_us : 0.0;
        if (dev_knobs().debug_segments) std::fprintf(stderr, "[skeres_amd] two segments with %d border members in the separator: %.0f us + all-reduce %.0f us against %.0f us on one device\n",
                                                     border_members_, ds.t_dissected, allreduce_us, one_us);
        if (ds.a > 0 && opt_.distribution_mode != SK_DISTRIBUTION_SEGMENTED && ds.t_dissected + allreduce_us + 0.5 * shard_us >= 0.9 * (one_us + shard_us)) ds.a = ds.b = 0;
        // ... and MORE than two segments, a device each: every segment's front has the members' rows in its border, the root is the
        // separators' block-tridiagonal system bordered by the members (choose_segments; every rank factors it)
        int max_seg = opt_.world;
        if (opt_.max_segments >= 2) max_seg = std::min(max_seg, opt_.max_segments);
        if (max_seg > 2) {
          const bool forced = opt_.distribution_mode == SK_DISTRIBUTION_SEGMENTED;
          const Segments sg = choose_segments(docam, dopt, Cband, P_total_, dnblk, denv, first_col, max_seg, forced, opt_.world, C_ - Cband,
                                              pseudo_border ? &extra_fwd : nullptr, pseudo_border ? &extra_bwd_col : nullptr, &band_reach);
          for (int k = 3; k < 9; ++k) model_us_[k] = sg.model_us[k];
          const int Rn = (int)sg.a.size() + 1;
          const double two_us = ds.a > 0 ? ds.t_dissected + allreduce_us + 0.5 * shard_us : 1e300;
          if (Rn > 2 && (forced || (sg.t_model + shard_us / Rn < 0.95 * two_us && sg.t_model + shard_us / Rn < 0.9 * (one_us + shard_us)))) {
            if (dev_knobs().debug_segments) std::fprintf(stderr, "[skeres_amd] %d segments with %d border members in the root: %.0f us against %.0f us in two\n", Rn, border_members_, sg.t_model, two_us);
            many_cut_a = sg.a; many_cut_b = sg.b;
            dissect_t_model_many = sg.t_model;
          }
        }
      }
      if (lockstep_cut && ds.a > 0) {  // (two resident servers per factorisation: the fifth such solver alive on a device stays undissected)
        if (!pair_claimed_) pair_claimed_ = cholesky_claim_pair_servers(&chol_ctx_);
        if (!pair_claimed_) ds.a = ds.b = 0;
      }
      // AUTO does not dissect on ONE device: measured on MI355X (profiles/r02_dissection_*), the two chains side by side
      // on one chip take longer than one after the other — each alone 5.0 and 3.0 ms, together 10-13 ms; 6.6 ms only under
      // rocprofv3's kernel tracing — so the model's prediction (kept in sk_solver_stat) is not acted upon there.
      if (opt_.dissection == SK_DISSECTION_AUTO && dev_knobs().dissect_at < 0 && !lockstep_cut && !two_seg_try) { ds.a = ds.b = 0; }
      if (two_seg_try && opt_.distribution_mode == SK_DISTRIBUTION_SEGMENTED && ds.a == 0 && many_cut_a.empty() && Cband >= 6) {
        // forced (tests, small problems): cut the band at its middle camera wherever that leaves a tail
        std::vector<int> cmin(P_total_, Cband), cmax(P_total_, -1);
        for (size_t b = 0; b < docam.size(); ++b) { cmin[dopt[b]] = std::min(cmin[dopt[b]], docam[b]); cmax[dopt[b]] = std::max(cmax[dopt[b]], docam[b]); }
        for (int a = Cband / 2; a >= 1 && ds.a == 0; --a) {
          int b = a;
          for (int q = 0; q < P_total_; ++q) if (cmin[q] < a) b = std::max(b, cmax[q] + 1);
          if (b < Cband) { ds.a = a; ds.b = b; }
        }
      }
      if (opt_.dissection == SK_DISSECTION_ON && ds.a == 0 && C_ >= 6) {
        // forced (tests, small problems): cut at the middle camera wherever that leaves a tail
        std::vector<int> cmin(P_total_, C_), cmax(P_total_, -1);
        for (size_t b = 0; b < ocam.size(); ++b) { cmin[opt[b]] = std::min(cmin[opt[b]], ocam[b]); cmax[opt[b]] = std::max(cmax[opt[b]], ocam[b]); }
        for (int a = C_ / 2; a >= 1 && ds.a == 0; --a) {
          int b = a;
          for (int q = 0; q < P_total_; ++q) if (cmin[q] < a) b = std::max(b, cmax[q] + 1);
          if (b < C_) { ds.a = a; ds.b = b; }
        }
      }
      if (dev_knobs().dissect_at >= 0) {  // developer variable SK_DISSECT_AT: head size in cameras (0: no dissection)
        ds.a = dev_knobs().dissect_at; ds.b = 0;
        if (ds.a > 0 && ds.a < C_) {
          ds.b = ds.a;
          std::vector<int> cmin(P_total_, C_);
          for (size_t b = 0; b < ocam.size(); ++b) cmin[opt[b]] = std::min(cmin[opt[b]], ocam[b]);
          for (size_t b = 0; b < ocam.size(); ++b) if (cmin[opt[b]] < ds.a) ds.b = std::max(ds.b, ocam[b] + 1);
          if (ds.b >= C_) ds.a = ds.b = 0;
        } else ds.a = 0;
      }
      dissect_t_plain_ = ds.t_plain; dissect_t_model_ = ds.t_dissected;
      if (!two_seg_try) {  // what the chain model predicts for 2 .. 8 devices (sk_solver_stat "model_us_segments_<n>": bench.py prints it beside what it measures)
        const Segments sg = choose_segments(docam, dopt, Cband, P_total_, dnblk, denv, first_col, 8, false, 0, C_ - Cband, pseudo_border ? &extra_fwd : nullptr,
                                            pseudo_border ? &extra_bwd_col : nullptr, &band_reach);
        for (int k = 0; k < 9; ++k) model_us_[k] = sg.model_us[k];
        if (pseudo_border) {
          // ... and with the border's members (retained points) in the one separator of TWO segments, a device each — what a world of ranks
          // takes when it beats this device's plan by 10 % (pass 0 above): "model_us_two_segments_with_members"
          const Dissection two = choose_dissection(docam, dopt, Cband, P_total_, dnblk, denv, first_col, true, false, C_ - Cband, &extra_fwd, extra_bwd, &extra_bwd_col, &band_reach);
          if (two.a > 0) {
            const double E = (9.0 * (two.b - two.a + C_ - Cband) + 1.0 + 127.0) / 128.0;
            two_segments_members_us_ = two.t_dissected + 50.0 + 0.5 * E * (E + 1.0) * 128.0 * 128.0 * 8.0 / 153e3;  // (+ the root's all-reduce over two ranks)
          }
        }
      }
      if (!many_cut_a.empty()) { cut_a = many_cut_a; cut_b = many_cut_b; dissect_t_model_ = dissect_t_model_many; }
      else if (ds.a > 0 && ds.b < Cband) { cut_a.push_back(ds.a); cut_b.push_back(ds.b); }
    }
    if (two_seg_try && cut_a.empty() && opt_.distribution_mode != SK_DISTRIBUTION_SEGMENTED) {
      // no cut that pays: shard the points or replicate (a replicating rank is a single device from here on), then once more
      int rc = shard_or_replicate();
      if (rc) return rc;
      continue;
    }
    if (multi && opt_.distribution_mode == SK_DISTRIBUTION_SEGMENTED && cut_a.empty()) {
      set_error("the segmented distribution needs a separator in the camera sequence (no point seen from both ends); not supported for this problem");
      return SK_ERR_UNSUPPORTED;
    }
    if (!cut_a.empty()) {
      dissected_ = true;
      segmented_ = multi;
      // final numbering: the segments one after the other — the last one REVERSED (it is eliminated back to front) —
      // then the separators, in sequence order
      const int R = (int)cut_a.size() + 1;
      seg_off_.assign(R + 1, 0);
      for (int sg = 0; sg < R; ++sg) {
        const int lo = sg == 0 ? 0 : cut_b[sg - 1], hi = sg + 1 < R ? cut_a[sg] : Cband;
        seg_off_[sg + 1] = seg_off_[sg] + (hi - lo);
      }
      cam_b_ = seg_off_[R];
      std::vector<int> fin(C_);
      int sep_pos = cam_b_;
      for (int sg = 0; sg < R; ++sg) {
        const int lo = sg == 0 ? 0 : cut_b[sg - 1], hi = sg + 1 < R ? cut_a[sg] : Cband;
        for (int c = lo; c < hi; ++c) fin[c] = sg + 1 < R ? seg_off_[sg] + (c - lo) : seg_off_[sg] + (hi - 1 - c);
        if (sg + 1 < R) { sep_first_.push_back(sep_pos); for (int c = cut_a[sg]; c < cut_b[sg]; ++c) fin[c] = sep_pos++; }
      }
      for (int c = Cband; c < C_; ++c) fin[c] = sep_pos++;  // the border's members (loop-closure cameras, pseudo-cameras of retained points): the end of the (one) separator
      sep_first_.push_back(C_);
      std::vector<int> cb2(C_);
      for (int c = 0; c < C_; ++c) cb2[fin[c]] = cam_block_[c];
      cam_block_.swap(cb2);
      for (int& c : ocam) c = fin[c];
      for (int& c : struct_ocam_) c = fin[c];
      for (int& c : retained_cam_) c = fin[c];
      env_tail_.clear();  // (the fronts have envelopes of their own)
      cam_a_ = seg_off_[1];  // (one device: the head [0, cam_a_), the tail [cam_a_, cam_b_))
      segments_ = R;
    }
    break;
  }
  if (segmented_) {
    // Rank r < segments_ owns segment r; further ranks are replicas of rank (r mod segments_): they do the same work and
    // add zeros to every sum.  What is exchanged per iteration: the root front (the separators' block-tridiagonal system
    // with the segments' Schur complements), the cameras' column norms and gradient, and a handful of scalars.
    distribution_ = SK_DISTRIBUTION_SEGMENTED;
    role_ = opt_.rank % segments_;
    replica_ = opt_.rank >= segments_;
    fold_world_ = segments_;
    my_lo_ = seg_off_[role_]; my_hi_ = seg_off_[role_ + 1];
    // (the border's members — behind the last separator — are a border of the root too: active from its first block column)
    std::vector<int> sep_off;  // scalar offsets of the separators in the root, then their total
    for (int f : sep_first_) sep_off.push_back(9 * (std::min(f, C_ - border_members_) - cam_b_));
    root_tail_.clear();
    root_last_ = root_envelope(sep_off, 9 * border_members_, &root_tail_);
    const int E = (9 * (C_ - cam_b_) + 1 + 127) / 128;
    pack_col0_h_.assign(E, 0);
    pack_off_h_.assign(E + 1, 0);
    if (!root_last_.empty()) pack_col0_h_ = cholesky_row_first_cols(E, root_last_.data(), root_tail_.empty() ? nullptr : root_tail_.data());
    for (int kb = 0; kb < E; ++kb) pack_off_h_[kb + 1] = pack_off_h_[kb] + (long long)128 * 128 * (kb + 1 - pack_col0_h_[kb]);
    packed_elems_ = (size_t)pack_off_h_[E];
  }
  if (opt_.allreduce && !distribution_decided) {
    int rc = prepare_pack();
    if (rc) return rc;
    if (!segmented_) {
      rc = choose_distribution(opt);
      if (rc) return rc;
    }
  }
  // ---- this rank's points: a contiguous run of equal sum k^2 (sharded), or (segmented) the points that see a camera of
  // its segment — each such point sees only that segment and the separator — plus every other of the points that see the
  // separator alone ----
  std::vector<int> local_of(P_total_, -1);
  std::vector<int> obs_rank;  // segmented world with retained points: the rank of every observation of a retained point (-1: not one)
  if (segmented_) {
    std::vector<int> seg_of_cam(C_, -1);  // (separator cameras: -1)
    for (int sg = 0; sg < segments_; ++sg) for (int c = seg_off_[sg]; c < seg_off_[sg + 1]; ++c) seg_of_cam[c] = sg;
    std::vector<int> seg_of_pt(P_total_, -1);
    // A RETAINED point is seen from every segment its track crosses: its observations are split by camera — those of a segment's cameras
    // to that segment's rank, those of separator cameras to its HOME rank (q mod segments) — and every rank that has any of them keeps a
    // copy of the point.  What is a sum over the point's observations (its column norms, gradient, T = sum E^T E) is summed over the
    // ranks; what is the point's own (D_p^2, the right-hand side's entry, |x_p|^2, |delta_p|^2, the value written back) is the home rank's.
    std::vector<char> is_kept(P_total_, 0);
    for (int q : retained_pts_) is_kept[q] = 1;
    obs_rank.assign(Nall, -1);  // (kept observations only)
    for (int b = 0; b < Nall; ++b) {
      const int sg = seg_of_cam[ocam[b]];
      if (is_kept[opt[b]]) { obs_rank[b] = sg >= 0 ? sg : opt[b] % segments_; continue; }
      if (sg < 0) continue;
      if (seg_of_pt[opt[b]] >= 0 && seg_of_pt[opt[b]] != sg) { set_error("internal: a point is seen from two segments of the camera sequence"); return SK_ERR_UNSUPPORTED; }
      seg_of_pt[opt[b]] = sg;
    }
    std::vector<char> kept_here(P_total_, 0);
    for (int b = 0; b < Nall; ++b) if (obs_rank[b] == role_) kept_here[opt[b]] = 1;
    std::vector<int> guests;  // copies of retained points whose home is another rank: LAST among the local points (the norms run over the others)
    for (int q = 0; q < P_total_; ++q) {
      if (is_kept[q]) {
        const bool home = (q % segments_) == role_;
        if (home) { local_of[q] = (int)local_pt_.size(); local_pt_.push_back(q); }
        else if (kept_here[q]) guests.push_back(q);
        continue;
      }
      const bool mine = seg_of_pt[q] == role_ || (seg_of_pt[q] < 0 && (q % segments_) == role_);
      if (mine) { local_of[q] = (int)local_pt_.size(); local_pt_.push_back(q); }
    }
    P_own_ = (int)local_pt_.size();
    for (int q : guests) { local_of[q] = (int)local_pt_.size(); local_pt_.push_back(q); }
  } else {
    int p_lo = 0, p_hi = P_total_;
    if (opt_.world > 1) {
      std::vector<int> cut;
      bal_partition_points(opt, P_total_, opt_.world, &cut);
      p_lo = cut[opt_.rank]; p_hi = cut[opt_.rank + 1];
    }
    local_pt_.resize(p_hi - p_lo); std::iota(local_pt_.begin(), local_pt_.end(), p_lo);
    for (int q = p_lo; q < p_hi; ++q) local_of[q] = q - p_lo;
  }
  P_ = (int)local_pt_.size();
  if (!segmented_) P_own_ = P_;
  auto obs_here = [&](int b) { return local_of[opt[b]] >= 0 && (obs_rank.empty() || obs_rank[b] < 0 || obs_rank[b] == role_); };
  // local observations, point-major, ascending camera within a point
  std::vector<int> pt_start(P_ + 1, 0);
  for (int b = 0; b < Nall; ++b) if (obs_here(b)) pt_start[local_of[opt[b]] + 1]++;
  for (int q = 0; q < P_; ++q) pt_start[q + 1] += pt_start[q];
  N_ = pt_start[P_];
  std::vector<int> order(N_);
  { std::vector<int> fill(pt_start.begin(), pt_start.end() - 1);
    for (int b = 0; b < Nall; ++b) if (obs_here(b)) order[fill[local_of[opt[b]]]++] = b; }
  for (int q = 0; q < P_; ++q)
    std::sort(order.begin() + pt_start[q], order.begin() + pt_start[q + 1], [&](int a, int b) { return ocam[a] != ocam[b] ? ocam[a] < ocam[b] : a < b; });
  std::vector<int> cam(N_), pt(N_);
  // captured doubles per observation: (observedX, observedY) of SnavelyReprojectionError, or whatever a recorded functor captures
  const Tape* tape = nullptr;
  for (int b = 0; b < Nall && !tape; ++b) tape = p.tape_of_block(b);
  const int nobs = tape ? tape->num_obs_consts : 2;
  std::vector<double> obs((size_t)std::max(1, nobs) * (size_t)N_);
  for (int o = 0; o < N_; ++o) {
    const int b = order[o];
    cam[o] = ocam[b]; pt[o] = local_of[opt[b]];
    if (p.rb_functor[b] == SK_FUNCTOR_HOST_CALLBACK) {  // no captured doubles on the device: the caller's object holds them
      host_obs_.push_back(o); host_cf_.push_back(p.rb_cost[b]);
    } else {
      for (int k = 0; k < nobs; ++k) obs[(size_t)k * N_ + o] = p.consts[p.rb_const_off[b] + k];
    }
  }
  // Two residual blocks on one (camera, point) pair (the reference's set-up loop adds whatever the file holds: EX/SimpleBundleAdjuster.scala:139-145):
  // both observations enter every sum; their cross term of the Schur complement belongs to the camera's diagonal block (BalDev::dup_*)
  bool has_dup = false;
  for (int o = 1; o < N_ && !has_dup; ++o) has_dup = pt[o] == pt[o - 1] && cam[o] == cam[o - 1];
  // camera CSR (ascending point because observation order is point-major)
  std::vector<int> cam_start(C_ + 1, 0), cam_obs(N_);
  for (int o = 0; o < N_; ++o) cam_start[cam[o] + 1]++;
  for (int i = 0; i < C_; ++i) cam_start[i + 1] += cam_start[i];
  { std::vector<int> fill(cam_start.begin(), cam_start.end() - 1); for (int o = 0; o < N_; ++o) cam_obs[fill[cam[o]]++] = o; }
  // retained points of this rank: local point, pseudo-camera, slot
  std::vector<int> kept_of_local(P_, -1), kept_pt, kept_cam, kept_home, kept_global;
  for (size_t k = 0; k < retained_pts_.size(); ++k) {
    const int q = local_of[retained_pts_[k]];
    if (q < 0) continue;
    kept_of_local[q] = (int)k;
    kept_pt.push_back(q); kept_cam.push_back(3 * retained_cam_[k] + (int)(k % 3));
    kept_home.push_back(q < P_own_ ? 1 : 0); kept_global.push_back((int)k);
  }
  stage("dissection, local observations, camera lists");
  // pair lists: for every point that is eliminated, every (larger camera, smaller camera) pair of its observations
  size_t npairs = 0;
  for (int q = 0; q < P_; ++q) { if (kept_of_local[q] >= 0) continue; const size_t k = pt_start[q + 1] - pt_start[q]; npairs += k * (k - 1) / 2; }
  std::vector<int> dup_a, dup_b, dup_cam;  // (observation indices here; record slots below)
  if (has_dup) {
    npairs = 0;
    for (int q = 0; q < P_; ++q) {
      bool q_dup = false;
      for (int b = pt_start[q] + 1; b < pt_start[q + 1]; ++b)
        for (int a = pt_start[q]; a < b; ++a) {
          if (cam[b] != cam[a]) { if (kept_of_local[q] < 0) ++npairs; continue; }
          q_dup = true;
          dup_a.push_back(a); dup_b.push_back(b); dup_cam.push_back(cam[a]);
        }
      if (q_dup && kept_of_local[q] >= 0) { set_error("internal: a retained point has two residual blocks on one camera"); return SK_ERR_UNSUPPORTED; }
    }
    // camera by camera (bal_dup_diag_kernel: one workgroup per camera's run)
    std::vector<int> ord(dup_cam.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return dup_cam[x] < dup_cam[y]; });
    std::vector<int> a2(ord.size()), b2(ord.size()), c2(ord.size());
    for (size_t k = 0; k < ord.size(); ++k) { a2[k] = dup_a[ord[k]]; b2[k] = dup_b[ord[k]]; c2[k] = dup_cam[ord[k]]; }
    dup_a.swap(a2); dup_b.swap(b2); dup_cam.swap(c2);
  }
  if (npairs > 2000000000ull) { set_error("pair list too large"); return SK_ERR_UNSUPPORTED; }
  std::vector<int> pair_row(npairs), pair_col(npairs), seg_start, seg_row, seg_col;
  {
    const size_t CC = (size_t)C_ * C_;
    std::vector<unsigned> count(CC + 1, 0);  // key = row * C + col
    for (int q = 0; q < P_; ++q) {
      if (kept_of_local[q] >= 0) continue;
      for (int b = pt_start[q] + 1; b < pt_start[q + 1]; ++b)
        for (int a = pt_start[q]; a < b; ++a) if (cam[b] != cam[a]) count[(size_t)cam[b] * C_ + cam[a] + 1]++;
    }
    seg_start.push_back(0);
    std::vector<unsigned> pos(CC, 0);
    unsigned run = 0;
    for (size_t key = 0; key < CC; ++key) {
      pos[key] = run;
      if (count[key + 1]) { seg_row.push_back((int)(key / C_)); seg_col.push_back((int)(key % C_)); run += count[key + 1]; seg_start.push_back((int)run); }
    }
    for (int q = 0; q < P_; ++q) {  // ascending point => entries of a segment are in ascending point order
      if (kept_of_local[q] >= 0) continue;
      for (int b = pt_start[q] + 1; b < pt_start[q + 1]; ++b)
        for (int a = pt_start[q]; a < b; ++a) { if (cam[b] == cam[a]) continue; const unsigned e = pos[(size_t)cam[b] * C_ + cam[a]]++; pair_row[e] = b; pair_col[e] = a; }
    }
  }
  // ---- device buffers ----
  hipStream_t s = stream_;
  SK_HIP_TRY(b_cam_.upload(cam, s)); SK_HIP_TRY(b_pt_.upload(pt, s)); SK_HIP_TRY(b_obs_.upload(obs, s));
  if (tape) { tape_mode_ = true; SK_HIP_TRY(tape_dev_.upload(*tape, s)); }
  SK_HIP_TRY(b_pt_start_.upload(pt_start, s)); SK_HIP_TRY(b_cam_start_.upload(cam_start, s)); SK_HIP_TRY(b_cam_obs_.upload(cam_obs, s));
  {
    std::vector<int> slot(N_);
    for (int e = 0; e < N_; ++e) slot[cam_obs[e]] = e;
    SK_HIP_TRY(b_obs_slot_.upload(slot, s));
    // the pair lists address the What records, which are in camera-major order (bal_kernels.hpp: kWs)
    for (int& v : pair_row) v = slot[v];
    for (int& v : pair_col) v = slot[v];
    for (int& v : dup_a) v = slot[v];
    for (int& v : dup_b) v = slot[v];
  }
  if (!dup_cam.empty()) { SK_HIP_TRY(b_dup_a_.upload(dup_a, s)); SK_HIP_TRY(b_dup_b_.upload(dup_b, s)); SK_HIP_TRY(b_dup_cam_.upload(dup_cam, s)); }
  num_dup_ = (int)dup_cam.size();
  SK_HIP_TRY(b_seg_start_.upload(seg_start, s)); SK_HIP_TRY(b_seg_row_.upload(seg_row, s)); SK_HIP_TRY(b_seg_col_.upload(seg_col, s));
  SK_HIP_TRY(b_pair_row_.upload(pair_row, s)); SK_HIP_TRY(b_pair_col_.upload(pair_col, s));
  std::vector<int> short_segs, long_segs;
  for (int g = 0; g < (int)seg_row.size(); ++g) (seg_start[g + 1] - seg_start[g] >= kLongSegment ? long_segs : short_segs).push_back(g);
  // Both lists stay in (row camera, column camera) order: neighbouring waves then gather the records of the same
  // points.  (Round 1 sorted the long list by length, longest first, against a long tail: 2.14 ms on Venice-1778 where
  // camera order takes 1.65.  Sorting the short list by length, so that the seven lane groups of a wave finish together,
  // changes nothing in time and fetches 588 MB instead of 345 on Ladybug-1723.)
  // (Round 4 also tried sorting the short list by length inside windows of 224 consecutive segments — the 32 waves that run on
  // one XCD together — so that the seven lane groups of a wave carry segments of like length: bal_pair 131 -> 178 us on
  // Ladybug-1723, 250 -> 322 on Venice-1778.  A wave of seven LONG short segments gathers seven times the records at once;
  // mixed lengths spread that load.  And bal_pair_long at six waves per SIMD (79 VGPRs, no scratch) instead of five: 83-88 ->
  // 87 us, Venice 1088 -> 1130.  Neither kept.)
  // (Round 4 tried the lists in Z-order of (row camera, column camera) — runs of consecutive segments inside small squares of
  // camera pairs, so that the rows' AND the columns' records stay in an XCD's L2: no gain, Schur assembly 0.41 -> 0.42-0.44 ms on
  // Ladybug-1723, 2.46 -> 2.40-2.54 on Venice-1778 for cells of 1, 4 and 16 cameras: the gathers are not bound by L2 misses.)
  SK_HIP_TRY(b_short_segs_.upload(short_segs, s)); SK_HIP_TRY(b_long_segs_.upload(long_segs, s));
  d_.num_short_segments = (int)short_segs.size(); d_.num_long_segments = (int)long_segs.size();
  const size_t nc = 9 * (size_t)C_, np = 3 * (size_t)P_, nx = nc + np;
  std::vector<double> x(nx, 0.0);  // (the padding coordinates of a smaller shape: zeros, and inert — see free_mask below)
  for (int i = 0; i < C_; ++i) if (cam_block_[i] >= 0) std::memcpy(&x[9 * (size_t)i], p.block_ptr[cam_block_[i]], cam_size_ * sizeof(double));  // (a pseudo-camera: zeros)
  for (int q = 0; q < P_; ++q) std::memcpy(&x[nc + 3 * (size_t)q], p.block_ptr[pt_block_[local_pt_[q]]], pt_size_ * sizeof(double));
  // x vectors are stored [cameras | points] so whole-vector kernels run once
  SK_HIP_TRY(b_xc_.upload(x, s)); SK_HIP_TRY(b_xc_new_.alloc(nx));
  SK_HIP_TRY(b_scale_.alloc(nx)); SK_HIP_TRY(b_colsq_.alloc(nx)); SK_HIP_TRY(b_gs_.alloc(nx)); SK_HIP_TRY(b_step_.alloc(nx));
  {
    // scale starts as the mask of free coordinates: 1, or 0 for a coordinate that is held constant — a constant parameter
    // block (Problem::SetParameterBlockConstant) or the constant coordinates of a SubsetParameterization (ceres.i:186-210);
    // an IdentityParameterization changes nothing.  See jacobi_scale_kernel.
    std::vector<double> free_mask(nx, 1.0);
    auto mask_block = [&](int block, size_t off, int size, int padded) {
      for (int k = size; k < padded; ++k) free_mask[off + k] = 0.0;  // padding of a shape smaller than (2; 9, 3): inert coordinates
      if ((size_t)block < p.block_constant.size() && p.block_constant[block]) { for (int k = 0; k < size; ++k) free_mask[off + k] = 0.0; return; }
      const int pi = (size_t)block < p.block_param.size() ? p.block_param[block] : -1;
      if (pi < 0) return;
      const LocalParameterization& lp = p.params[pi];
      if (lp.type == kParamSubset) for (int k = 0; k < size; ++k) if ((lp.constant_mask >> k) & 1u) free_mask[off + k] = 0.0;
    };
    for (int i = 0; i < C_; ++i) {
      if (cam_block_[i] >= 0) mask_block(cam_block_[i], 9 * (size_t)i, cam_size_, 9);
      else for (int k = 0; k < 9; ++k) free_mask[9 * (size_t)i + k] = 0.0;  // a pseudo-camera's coordinates are nobody's parameters: inert
    }
    for (int q = 0; q < P_; ++q) mask_block(pt_block_[local_pt_[q]], nc + 3 * (size_t)q, pt_size_, 3);
    SK_HIP_TRY(hipMemcpyAsync(b_scale_.p, free_mask.data(), nx * sizeof(double), hipMemcpyHostToDevice, s));
    SK_HIP_TRY(hipStreamSynchronize(s));
  }
  SK_HIP_TRY(b_y_.alloc(npad_ + 128));
  SK_HIP_TRY(b_r_.alloc(2 * (size_t)N_)); SK_HIP_TRY(b_F_.alloc(18 * (size_t)N_)); SK_HIP_TRY(b_Fcam_.alloc(kFcam * (size_t)N_)); SK_HIP_TRY(b_E_.alloc(6 * (size_t)N_));
  SK_HIP_TRY(b_W_.alloc(kWs * (size_t)N_)); SK_HIP_TRY(b_rt_.alloc(5 * (size_t)N_));
  if (res_size_ < 2 || cam_size_ < 9 || pt_size_ < 3) {  // the planes of the padding coordinates / the missing residual row are never written: zero, once
    SK_HIP_TRY(b_r_.zero(s)); SK_HIP_TRY(b_F_.zero(s)); SK_HIP_TRY(b_E_.zero(s));
  }
  SK_HIP_TRY(b_M_.alloc(6 * (size_t)P_)); SK_HIP_TRY(b_q_.alloc(3 * (size_t)P_));
  if (pseudo_cams_ > 0) {
    std::vector<unsigned char> pseudo(C_, 0);
    for (int i = 0; i < C_; ++i) pseudo[i] = cam_block_[i] < 0 ? 1 : 0;
    SK_HIP_TRY(b_pseudo_.upload(pseudo, s));
    SK_HIP_TRY(b_kept_pt_.upload(kept_pt, s)); SK_HIP_TRY(b_kept_cam_.upload(kept_cam, s));
    if (segmented_) { SK_HIP_TRY(b_kept_home_.upload(kept_home, s)); SK_HIP_TRY(b_kept_global_.upload(kept_global, s)); }
    std::vector<int> kept_obs, kept_obs_slot;
    for (size_t k = 0; k < kept_pt.size(); ++k)
      for (int o = pt_start[kept_pt[k]]; o < pt_start[kept_pt[k] + 1]; ++o) { kept_obs.push_back(o); kept_obs_slot.push_back((int)k); }
    SK_HIP_TRY(b_kept_obs_.upload(kept_obs, s)); SK_HIP_TRY(b_kept_obs_slot_.upload(kept_obs_slot, s));
    num_kept_obs_ = (int)kept_obs.size();
  }
  stage("pair lists, uploads");
  // ---- the fronts of the reduced camera system ----
  std::vector<int> border_row_h[2], leaf_map_h, leaf_gmap_h;
  if (!dissected_) {
    FrontHost& r = fr_[2];
    r.nblk = r.ncols = npad_ / 128; r.cams = C_; r.dim = (size_t)npad_; r.rhs_row = rhs_row_; r.last = env_last_; r.tail = env_tail_;
    border_blocks_ = 0;
  } else {
    const int nsep = C_ - cam_b_;
    // the leaf fronts this device holds: one device — the head (0) and the tail (1); a rank of a segmented world — its segment (0)
    for (int f = 0; f < 2; ++f) {
      if (segmented_ && f != 0) continue;
      const int seg = segmented_ ? role_ : f;
      const int lo = seg_off_[seg], hi = seg_off_[seg + 1];
      // the separators next to the segment (cameras of the final numbering): left [ll, lh), right [rl, rh)
      // (the members of a border — pseudo-cameras of retained points, loop-closure cameras — come behind the last separator and are
      // rows of EVERY leaf front: the end of its border, before the right-hand side)
      const int nbm = border_members_, mf = C_ - nbm;
      const int ll = seg > 0 ? sep_first_[seg - 1] : 0, lh = seg > 0 ? std::min(sep_first_[seg], mf) : 0;
      const int rl = seg + 1 < segments_ ? sep_first_[seg] : 0, rh = seg + 1 < segments_ ? std::min(sep_first_[seg + 1], mf) : 0;
      // first segment: [right | members | rhs]; last: [left reversed | members | rhs]; between two: [right, padded | left | members | rhs] —
      // there the members are tail rows like the left separator's (active in every column)
      const SegmentLayout lay = seg == 0 ? segment_layout(9 * (hi - lo), 0, 9 * (rh - rl + nbm)) : segment_layout(9 * (hi - lo), 9 * (lh - ll + nbm), 9 * (rh - rl));
      FrontHost& L = fr_[f];
      L.cams = hi - lo; L.ncols = lay.ncols; L.nblk = lay.nblk; L.dim = (size_t)L.nblk * 128; L.rhs_row = lay.rhs_row; L.tail_rows = lay.tail_rows;
      // rows of every camera in this front: its own interior, or (a separator next to it) the border
      std::vector<int> pos(C_, -1);
      std::vector<char> interior(C_, 0);
      for (int c = lo; c < hi; ++c) { pos[c] = 9 * (c - lo); interior[c] = 1; }
      const int bo = L.ncols * 128;
      for (int c = rl; c < rh; ++c) pos[c] = bo + lay.right_off + 9 * (c - rl);
      // (the members of a border — one device: the end of the one separator — stay at the END of a reversed border too: their rows are
      // tail rows of the front's envelope, a suffix of the matrix)
      for (int c = ll; c < lh; ++c) pos[c] = bo + lay.left_off + (lay.reversed ? 9 * (lh - 1 - c) : 9 * (c - ll));
      for (int c = mf; c < C_; ++c) pos[c] = (seg == 0 ? bo + lay.right_off + 9 * (rh - rl) : bo + lay.left_off + 9 * (lh - ll)) + 9 * (c - mf);
      const std::vector<int>& fo = struct_ocam_.empty() ? ocam : struct_ocam_;
      const std::vector<int>& fp = struct_ocam_.empty() ? opt : struct_opt_;
      const int fP = struct_ocam_.empty() ? P_total_ : struct_P_;
      if (nbm > 0 && !lay.spike) L.last = front_envelope(fo, fp, pos, interior, fP, L.nblk, L.tail_rows, pos[mf], &L.tail);
      else L.last = front_envelope(fo, fp, pos, interior, fP, L.nblk, L.tail_rows);
      border_row_h[f].assign(std::max(1, nsep), 0);
      for (int c = cam_b_; c < C_; ++c) border_row_h[f][c - cam_b_] = pos[c] >= 0 ? pos[c] : 0;  // (a separator that is not next to the segment: no block of it here)
      if (segmented_) {
        // border index -> root index (cholesky_border_add, cholesky_gather_map): separator cameras, and the right-hand-side row
        leaf_map_h.assign((size_t)(L.nblk - L.ncols) * 128, -1);
        for (int c = cam_b_; c < C_; ++c) if (pos[c] >= 0) for (int k = 0; k < 9; ++k) leaf_map_h[pos[c] - bo + k] = 9 * (c - cam_b_) + k;
        leaf_gmap_h = leaf_map_h;
        leaf_map_h[L.rhs_row - bo] = 9 * nsep;
      }
    }
    border_blocks_ = segmented_ ? fr_[0].nblk - fr_[0].ncols : (9 * nsep + 1 + 127) / 128;
    FrontHost& r = fr_[2];
    r.nblk = r.ncols = (9 * nsep + 1 + 127) / 128; r.cams = nsep; r.dim = (size_t)r.nblk * 128; r.rhs_row = 9 * nsep;
    r.last = root_last_;  // (one separator: dense)
    r.tail = root_tail_;
  }
  {
    size_t s_off = 0, linv_off = 0, y_off = 0;
    for (int f = 0; f < 3; ++f) {
      fr_[f].s_off = s_off; fr_[f].linv_off = linv_off; fr_[f].y_off = y_off;
      s_off += fr_[f].dim * fr_[f].dim; linv_off += (size_t)fr_[f].ncols * 128 * 128; y_off += fr_[f].dim;
    }
    SK_HIP_TRY(b_S_.alloc(s_off));
    SK_HIP_TRY(b_Linv_.alloc(linv_off)); SK_HIP_TRY(b_Linv_.zero(s));
    SK_HIP_TRY(b_yf_.alloc(y_off)); SK_HIP_TRY(b_yf_.zero(s)); SK_HIP_TRY(b_wf_.alloc(y_off)); SK_HIP_TRY(b_ybB_.alloc((size_t)std::max(1, border_blocks_) * 128));
  }
  if (opt_.allreduce) { SK_HIP_TRY(b_pack_col0_.upload(pack_col0_h_, s)); SK_HIP_TRY(b_pack_off_.upload(pack_off_h_, s)); }
  cholesky_prepare(&chol_ctx_, s);  // (once per device: which queues the panel, bulk and server streams sit on; nothing without look-ahead)
  if (dissected_ && !segmented_) SK_HIP_TRY(chol_ctx_b_.init_secondary(chol_ctx_));  // (again, now that the queue choice is made: the queues it left over)
  SK_HIP_TRY(b_S_.zero(s));  // once: the blocks outside the envelopes are never touched again
  for (int f = 0; f < 3; ++f) {
    // first block column each block row is zeroed from: the row envelope, widened by the SYRK depth - 1 (inside a
    // group, the lazy updates read every column of the group down to the LAST column's envelope) — and the whole
    // width for the last block row (right-hand side) and without an envelope
    const FrontHost& F = fr_[f];
    if (F.nblk == 0) continue;
    std::vector<int> col0(F.nblk, 0);
    // the widest group of either way to factor (with / without the resident chain, which a timing mode switches off)
    const bool chain_here = chain_ok() && (f != 1 || tail_chain());
    const int widen = F.last.empty() ? 1 : std::max(group_, cholesky_plan_max_group(cholesky_plan(F.nblk, group_, F.last.data(), chain_here, F.ncols, F.tail_rows, F.tl())));
    if (!F.last.empty()) {
      const std::vector<int> first = cholesky_row_first_cols(F.nblk, F.last.data(), F.tl(), F.tail_rows);  // (the tail rows: from column 0, or as the profile has them)
      for (int i = 0; i < F.nblk; ++i) col0[i] = std::max(0, first[i] - (widen - 1));
    }
    SK_HIP_TRY(b_zero_col0_f_[f].upload(col0, s));
    // ... and what the back-substitution leaves to zero: the diagonal block of every factored block column, and from the first border
    // column on in the border's block rows (the Schur complement a leaf front accumulates there)
    std::vector<int> col_min(F.nblk);
    for (int i = 0; i < F.nblk; ++i) col_min[i] = i < F.ncols ? i : F.ncols;
    SK_HIP_TRY(b_zero_min_f_[f].upload(col_min, s));
  }
  for (int f = 0; f < 2; ++f) if (!border_row_h[f].empty()) SK_HIP_TRY(b_border_row_[f].upload(border_row_h[f], s));
  if (segmented_) { SK_HIP_TRY(b_leaf_map_.upload(leaf_map_h, s)); SK_HIP_TRY(b_leaf_gmap_.upload(leaf_gmap_h, s)); }
  if (dissected_ && !segmented_) {
    const int nsep = C_ - cam_b_;
    std::vector<int> mapB((size_t)border_blocks_ * 128, -1);
    const int nreal = nsep - border_members_;
    for (int k = 0; k < nsep; ++k) for (int c = 0; c < 9; ++c) mapB[9 * k + c] = k < nreal ? 9 * (nreal - 1 - k) + c : 9 * k + c;  // camera order reversed, coordinates in order (pseudo-cameras: in place)
    mapB[9 * nsep] = 9 * nsep;  // right-hand-side row
    SK_HIP_TRY(b_mapB_.upload(mapB, s));
    mapB_involution_ = true;  // (a reversal of the real separator cameras, the rest in place)
    for (size_t i = 0; i < mapB.size() && mapB_involution_; ++i) if (mapB[i] >= 0 && mapB[(size_t)mapB[i]] != (int)i) mapB_involution_ = false;
  }
  partial_stride_ = std::max(std::max(std::max(bal_partial_blocks(N_), bal_point_blocks(P_) + 1), (9 * C_ + 255) / 256), 256) + bal_partial_blocks((int)host_obs_.size());  // (+ 1: the retained points' slot)
  SK_HIP_TRY(b_partial_.alloc(4 * (size_t)partial_stride_));
  SK_HIP_TRY(b_scal_.alloc(16)); SK_HIP_TRY(b_scal_.zero(s)); SK_HIP_TRY(b_small_.alloc(2 * nc + 6 * retained_pts_.size() + 64 + 16 * (size_t)opt_.world));
  fail_p_ = reinterpret_cast<int*>(b_scal_.p + 14); info_p_ = reinterpret_cast<int*>(b_scal_.p + 15);
  SK_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_scal_), 64 * sizeof(double), hipHostMallocDefault));
  stage("fronts, zero pass, tables");
  // ---- device view ----
  d_.C = C_; d_.P = P_; d_.N = N_;
  d_.pseudo = pseudo_cams_ > 0 ? b_pseudo_.p : nullptr; d_.num_kept = pseudo_cams_ > 0 ? (int)kept_pt.size() : 0; d_.kept_pt = b_kept_pt_.p; d_.kept_cam = b_kept_cam_.p;
  d_.num_kept_obs = pseudo_cams_ > 0 ? num_kept_obs_ : 0; d_.kept_obs = b_kept_obs_.p; d_.kept_obs_slot = b_kept_obs_slot_.p;
  d_.kept_home = segmented_ && pseudo_cams_ > 0 ? b_kept_home_.p : nullptr; d_.kept_global = segmented_ && pseudo_cams_ > 0 ? b_kept_global_.p : nullptr;
  d_.num_dup = num_dup_; d_.dup_a = b_dup_a_.p; d_.dup_b = b_dup_b_.p; d_.dup_cam = b_dup_cam_.p;
  d_.res_size = res_size_; d_.cam_size = cam_size_; d_.pt_size = pt_size_;
  d_.cam = b_cam_.p; d_.pt = b_pt_.p; d_.obs = b_obs_.p; d_.pt_start = b_pt_start_.p; d_.cam_start = b_cam_start_.p; d_.cam_obs = b_cam_obs_.p; d_.obs_slot = b_obs_slot_.p;
  d_.num_segments = (int)seg_row.size(); d_.seg_start = b_seg_start_.p; d_.seg_row = b_seg_row_.p; d_.seg_col = b_seg_col_.p;
  d_.short_segments = b_short_segs_.p; d_.long_segments = b_long_segs_.p;
  d_.pair_row_obs = b_pair_row_.p; d_.pair_col_obs = b_pair_col_.p;
  d_.xc = b_xc_.p; d_.xp = b_xc_.p + nc; d_.xc_new = b_xc_new_.p; d_.xp_new = b_xc_new_.p + nc;
  d_.scale_c = b_scale_.p; d_.scale_p = b_scale_.p + nc; d_.colsq_c = b_colsq_.p; d_.colsq_p = b_colsq_.p + nc;
  d_.gs_c = b_gs_.p; d_.gs_p = b_gs_.p + nc; d_.step_c = b_step_.p; d_.step_p = b_step_.p + nc;
  d_.y_c = b_y_.p; d_.r = b_r_.p; d_.F = b_F_.p; d_.Fcam = b_Fcam_.p; d_.E = b_E_.p; d_.What = b_W_.p; d_.u = b_rt_.p; d_.M = b_M_.p; d_.q = b_q_.p;
  for (int f = 0; f < 3; ++f) {
    d_.front[f].S = fr_[f].nblk > 0 ? b_S_.p + fr_[f].s_off : nullptr; d_.front[f].ld = (int)fr_[f].dim; d_.front[f].interior = fr_[f].ncols * 128;
    d_.front[f].rhs_row = fr_[f].rhs_row; d_.front[f].border_row = f < 2 ? b_border_row_[f].p : nullptr;
    d_.y_front[f] = b_yf_.p + fr_[f].y_off;
  }
  d_.seg_lo = segmented_ ? my_lo_ : 0; d_.cam_a = segmented_ ? my_hi_ : cam_a_; d_.cam_b = cam_b_;
  d_.S = d_.front[2].S; d_.ld = d_.front[2].ld; d_.rhs_row = fr_[2].rhs_row;
  if (segmented_) {
    const FrontHost& F = fr_[0];
    leaf_.S = d_.front[0].S; leaf_.ld = (long)F.dim; leaf_.nblk = F.nblk; leaf_.ncols = F.ncols; leaf_.last = F.env();
    leaf_.Linv = b_Linv_.p + F.linv_off; leaf_.rhs_row = F.rhs_row; leaf_.tail_rows = F.tail_rows; leaf_.spike = role_ > 0 && role_ + 1 < segments_;
    leaf_.tail = F.tl();  // (two segments with border members in the separator: their rows are a tail profile of the leaf's envelope)
  }
  if (dissected_ && !segmented_) {
    auto view = [&](int f) {
      FrontView v;
      v.S = d_.front[f].S; v.ld = (long)fr_[f].dim; v.nblk = fr_[f].nblk; v.ncols = fr_[f].ncols; v.last = fr_[f].env();
      v.Linv = b_Linv_.p + fr_[f].linv_off; v.rhs_row = fr_[f].rhs_row; v.tail = fr_[f].tl();
      return v;
    };
    ds_.A = view(0); ds_.B = view(1); ds_.R = view(2); ds_.border_blocks = border_blocks_; ds_.mapB = b_mapB_.p; ds_.mapB_involution = mapB_involution_;
  }
  d_.partial = b_partial_.p; d_.partial_stride = partial_stride_; d_.fail_flag = fail_p_;
  d_.loss_nodes = nullptr; d_.loss_root = p.rb_loss.empty() ? -1 : p.rb_loss[0]; d_.loss_of_obs = nullptr;
  {
    // one loss for every residual block (the usual case: SimpleBundleAdjuster shares one trivialLoss, EX/SimpleBundleAdjuster.scala:135),
    // or a loss per block: then every observation carries its own root
    bool mixed = false;
    int any_root = -1;
    for (size_t b = 0; b < p.rb_loss.size(); ++b) { mixed = mixed || p.rb_loss[b] != p.rb_loss[0]; any_root = std::max(any_root, p.rb_loss[b]); }
    if (mixed) {
      std::vector<int> roots(N_);
      for (int o = 0; o < N_; ++o) roots[o] = p.rb_loss[order[o]];
      SK_HIP_TRY(b_loss_of_obs_.upload(roots, s));
      d_.loss_of_obs = b_loss_of_obs_.p; d_.loss_root = any_root;
    }
  }
  if (d_.loss_root >= 0) { SK_HIP_TRY(b_loss_nodes_.upload(p.loss_nodes, s)); d_.loss_nodes = b_loss_nodes_.p; }
  d_.is_host = nullptr; d_.num_host = (int)host_obs_.size(); d_.host_obs = nullptr; d_.host_rows = nullptr;
  if (!host_obs_.empty()) {
    std::vector<unsigned char> flag(N_, 0);
    for (int o : host_obs_) flag[o] = 1;
    SK_HIP_TRY(b_is_host_.upload(flag, s)); SK_HIP_TRY(b_host_obs_.upload(host_obs_, s));
    SK_HIP_TRY(b_host_rows_.alloc(host_obs_.size() * (size_t)kHostRow));
    d_.is_host = b_is_host_.p; d_.host_obs = b_host_obs_.p; d_.host_rows = b_host_rows_.p;
    h_cam_ = cam; h_pt_ = pt;
    host_x_.resize(nx); host_rows_h_.resize(host_obs_.size() * (size_t)kHostRow);
  }
  graph_mode_ = graph_mode_ && host_obs_.empty() && !dissected_ && !tape_mode_;
  zero_by_backsolve_ = !graph_mode_ && !dev_knobs().schedule_plain && opt_.resident_kernels;
  for (int f = 0; f < 3; ++f) if (fr_[f].nblk > 0 && !cholesky_backsolve_resident(fr_[f].nblk)) zero_by_backsolve_ = false;
  SK_HIP_TRY(hipStreamSynchronize(s));
  if (opt_.allreduce) {
    // every rank derived the camera order and the envelope for itself (from rank-invariant data): they must be the same
    // reduced system, or the all-reduce would sum mismatched matrices
    double v[2] = {order_hash_, -order_hash_};
    const int ops[2] = {1, 1};
    int rc = gather_rank_scalars_signed(v, 2);
    if (rc) return rc;
    if (v[0] != order_hash_ || v[1] != -order_hash_) {
      set_error("the ranks derived different camera orders for the reduced system (are the residual blocks added in the same order on every rank?)");
      return SK_ERR_COMM;
    }
    (void)ops;
    // ... and the same factorisation plan: a rank whose device cannot run the resident panel chain (its queue trial said
    // so: a shared or serialised device) takes every rank to the launch-by-launch plan — replicated factorisations must
    // round alike, or the ranks' parameters drift apart
    double off[1] = {chain_ok() && !chain_live() ? 1.0 : 0.0};
    rc = gather_rank_scalars_signed(off, 1);
    if (rc) return rc;
    if (off[0] > 0.0) cholesky_disable_chain(&chol_ctx_);
  }
  stage("device view, the ranks' agreement");
  return SK_OK;
}

// max over ranks of each value (values of either sign)
// The same table formed on the DEVICE (bal_pack_rank_scalars_kernel), summed and copied to pinned host memory behind whatever the
// stream holds — the caller synchronises once and folds.  For worlds whose table fits the pinned scalars' spare room.
bool BalSolver::rank_table_on_device(int K) const { return opt_.allreduce != nullptr && opt_.world > 1 && opt_.world * K <= 24; }
int BalSolver::enqueue_rank_table(int mode, int K) {
  double* dev = b_small_.p + 2 * 9 * (size_t)C_ + 6 * retained_pts_.size() + 64;
  launch_bal_pack_rank_scalars(b_scal_.p, dev, opt_.rank, opt_.world, mode, segmented_, stream_);
  int rc = allreduce(dev, (size_t)opt_.world * K);
  if (rc) return rc;
  SK_HIP_TRY(hipMemcpyAsync(h_scal_ + 36, dev, (size_t)opt_.world * K * sizeof(double), hipMemcpyDeviceToHost, stream_));
  return SK_OK;
}
int BalSolver::fold_rank_table(double* vals, int K, const int* ops) {
  const int W = opt_.world;
  const double* table = h_scal_ + 36;
  const int fold = fold_world_ > 0 ? std::min(fold_world_, W) : W;
  for (int k = 0; k < K; ++k) {
    double a = 0.0;
    for (int r = 0; r < fold; ++r) a = ops[k] ? std::max(a, table[(size_t)r * K + k]) : a + table[(size_t)r * K + k];
    vals[k] = a;
  }
  return SK_OK;
}
int BalSolver::gather_rank_scalars_signed(double* vals, int K) {
  const int W = opt_.world;
  std::vector<double> table((size_t)W * K, 0.0);
  for (int k = 0; k < K; ++k) table[(size_t)opt_.rank * K + k] = vals[k];
  double* dev = b_small_.p + 2 * 9 * (size_t)C_ + 6 * retained_pts_.size() + 64;
  SK_HIP_TRY(hipMemcpyAsync(dev, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
  int rc = allreduce(dev, table.size());
  if (rc) return rc;
  SK_HIP_TRY(hipMemcpyAsync(table.data(), dev, table.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  for (int k = 0; k < K; ++k) {
    double a = table[k];
    for (int r = 1; r < W; ++r) a = std::max(a, table[(size_t)r * K + k]);
    vals[k] = a;
  }
  return SK_OK;
}

// Combine per-rank scalars: every rank writes its K values into its own slot of
// a world x K table, the table is sum-reduced, then each rank folds the slots
// in rank order (identical result on every rank; ops: 0 sum, 1 max).
int BalSolver::gather_rank_scalars(double* vals, int K, const int* ops) {
  if (!opt_.allreduce) return SK_OK;
  const int W = opt_.world;
  std::vector<double> table((size_t)W * K, 0.0);
  for (int k = 0; k < K; ++k) table[(size_t)opt_.rank * K + k] = vals[k];
  double* dev = b_small_.p + 2 * 9 * (size_t)C_ + 6 * retained_pts_.size() + 64;
  SK_HIP_TRY(hipMemcpyAsync(dev, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
  int rc = allreduce(dev, table.size());
  if (rc) return rc;
  SK_HIP_TRY(hipMemcpyAsync(table.data(), dev, table.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  const int fold = fold_world_ > 0 ? std::min(fold_world_, W) : W;  // (segmented: ranks beyond the first two are replicas)
  for (int k = 0; k < K; ++k) {
    double a = 0.0;
    for (int r = 0; r < fold; ++r) a = ops[k] ? std::max(a, table[(size_t)r * K + k]) : a + table[(size_t)r * K + k];
    vals[k] = a;
  }
  return SK_OK;
}

int BalSolver::evaluate_with_jacobian(bool first) {
  hipStream_t s = stream_;
  const size_t nc = 9 * (size_t)C_, np = 3 * (size_t)P_;
  SK_HIP_TRY(hipEventRecord(ev_[kEvBegin], s));
  const bool graph = graph_ok() && !first;  // (iteration 0 also derives the Jacobi scaling: its own sequence, run once)
  const bool replay = graph && g_eval_[parity_] != nullptr;
  if (graph && !replay && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); graph_mode_ = false; return evaluate_with_jacobian(first); }
  CaptureGuard capture(s, graph && !replay, &graph_mode_);  // an early return below must not leave the stream capturing
  if (!replay) {
  kt_.begin("bal_eval_jac", s);
  if (tape_mode_) launch_bal_eval_jac_tape(d_, tape_dev_, s); else launch_bal_eval_jac(d_, s);
  kt_.end("bal_eval_jac", s);
  int nb = bal_partial_blocks(N_);
  if (d_.num_host > 0) {
    bool failed = false;
    int rc = host_callbacks(d_.xc, true, &failed);
    if (rc) return rc;
    if (failed) return SK_ERR_EVALUATION_FAILED;
    nb += launch_bal_host_jac(d_, nb, s);
  }
  kt_.begin("bal_cam_records", s); launch_bal_cam_records(d_, s); kt_.end("bal_cam_records", s);
  kt_.begin("bal_reduce", s); launch_bal_reduce(d_, s); kt_.end("bal_reduce", s);
  if (opt_.allreduce) {  // camera columns are summed over all ranks' observations
    double* buf = b_small_.p;
    // (a segmented world with retained points: their observations are split over the ranks — their column norms and gradient travel too)
    const size_t nk = segmented_ && pseudo_cams_ > 0 ? 6 * retained_pts_.size() : 0;
    if (nk) SK_HIP_TRY(hipMemsetAsync(buf + 2 * nc, 0, nk * sizeof(double), s));
    if (replica_) {  // its sums are rank (r mod segments_)'s over again
      SK_HIP_TRY(hipMemsetAsync(buf, 0, 2 * nc * sizeof(double), s));
    } else {
      SK_HIP_TRY(hipMemcpyAsync(buf, d_.colsq_c, nc * sizeof(double), hipMemcpyDeviceToDevice, s));
      SK_HIP_TRY(hipMemcpyAsync(buf + nc, d_.gs_c, nc * sizeof(double), hipMemcpyDeviceToDevice, s));
      if (nk) launch_bal_kept_sums(d_, buf + 2 * nc, (int)retained_pts_.size(), true, s);
    }
    int rc = allreduce(buf, 2 * nc + nk);
    if (rc) return rc;
    SK_HIP_TRY(hipMemcpyAsync(d_.colsq_c, buf, nc * sizeof(double), hipMemcpyDeviceToDevice, s));
    SK_HIP_TRY(hipMemcpyAsync(d_.gs_c, buf + nc, nc * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (nk) launch_bal_kept_sums(d_, buf + 2 * nc, (int)retained_pts_.size(), false, s);
  }
  if (first && opt_.jacobi_scaling) {
    launch_jacobi_scale(b_colsq_.p, b_scale_.p, (int)(nc + np), s);
    launch_bal_scale_jac(d_, s);
    launch_bal_cam_records(d_, s);  // F changed under the records
    launch_apply_scale_to_reductions(b_colsq_.p, b_gs_.p, b_scale_.p, (int)(nc + np), s);
  }
  // scalars: sum r^2 (slot 4) ; gradient max-norm and |x|^2 (cameras once, points local)
  if (!opt_.allreduce) {
    // one process: cameras and points as ONE vector ([cameras | points] in every buffer), and the three reductions — sum r^2,
    // max |g|, |x|^2 — in one launch (round 4: five launches of ~6 us each became two)
    const int g = launch_grad_max_xnorm(b_gs_.p, b_scale_.p, d_.xc, (int)(nc + np), b_partial_.p + (size_t)partial_stride_, partial_stride_, s);
    // (slots 2 and 3, the points' share in a world of ranks, stay zero: it is inside the cameras' slots here)
    ReduceRows rows;
    rows.n = 3;
    rows.row[0] = 0; rows.count[0] = nb; rows.out[0] = b_scal_.p + 4;
    rows.row[1] = 1; rows.count[1] = g; rows.is_max[1] = 1; rows.out[1] = b_scal_.p;
    rows.row[2] = 2; rows.count[2] = g; rows.out[2] = b_scal_.p + 1;
    launch_final_reduce_rows(b_partial_.p, partial_stride_, rows, s);
  } else {
  launch_final_reduce(b_partial_.p, partial_stride_, nb, 1, 0, b_scal_.p + 4, s);
  // cameras: every rank holds all of them — but in a segmented world only its own segment's (and the separator's) are
  // current, and the separator's |x|^2 must be counted once: the head's rank takes it
  int c_lo = 0, c_n = (int)nc;
  if (segmented_) { c_lo = 9 * my_lo_; c_n = 9 * (my_hi_ - my_lo_); }
  int gc = launch_grad_max_xnorm(d_.gs_c + c_lo, d_.scale_c + c_lo, d_.xc + c_lo, c_n, b_partial_.p, partial_stride_, s);
  if (segmented_ && role_ == 0) {  // + the separators (slots behind the segment's)
    const int lo2 = 9 * cam_b_, n2 = (int)nc - lo2;
    gc += launch_grad_max_xnorm(d_.gs_c + lo2, d_.scale_c + lo2, d_.xc + lo2, n2, b_partial_.p + gc, partial_stride_, s);
  }
  launch_final_reduce(b_partial_.p, partial_stride_, gc, 2, 1, b_scal_.p, s);
  // (the points this rank accounts for: all its own — not the copies of retained points whose home is another rank, which come last)
  const int np_own = 3 * P_own_;
  const int gp = launch_grad_max_xnorm(d_.gs_p, d_.scale_p, d_.xp, np_own, b_partial_.p + 2 * (size_t)partial_stride_, partial_stride_, s);
  launch_final_reduce(b_partial_.p + 2 * (size_t)partial_stride_, partial_stride_, np_own ? gp : 0, 2, 1, b_scal_.p + 2, s);
  }
  SK_HIP_TRY(hipMemcpyAsync(h_scal_, b_scal_.p, 5 * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (graph) {
    if (!replay) { capture.release(); int rc = finish_capture(s, &g_eval_[parity_]); if (rc) return rc; if (!graph_mode_) return evaluate_with_jacobian(first); }
    SK_HIP_TRY(hipGraphLaunch(g_eval_[parity_], s));
  }
  SK_HIP_TRY(hipEventRecord(ev_[kEvJac], s));
  // a world of ranks: the table of the ranks' scalars is formed on the device and summed BEFORE the one host synchronisation (round 5;
  // until then: synchronise, pack on the host, copy up, all-reduce, copy down, synchronise again)
  const bool dev_gather = rank_table_on_device(3) && !graph;
  if (dev_gather) { int rc = enqueue_rank_table(0, 3); if (rc) return rc; }
  SK_HIP_TRY(hipStreamSynchronize(s));
  const double sumsq = h_scal_[4];
  const double gmax_c = h_scal_[0], x2_c = h_scal_[1];
  // local: sum r^2, max |g_p|, |x_p|^2 — and in a segmented world this rank's cameras' share of max |g_c| and |x_c|^2 too
  double loc[3] = {sumsq, segmented_ ? std::max(h_scal_[2], gmax_c) : h_scal_[2], h_scal_[3] + (segmented_ ? x2_c : 0.0)};
  const int ops3[3] = {0, 1, 0};
  int rc = dev_gather ? fold_rank_table(loc, 3, ops3) : gather_rank_scalars(loc, 3, ops3);
  if (rc) return rc;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ev_[kEvBegin], ev_[kEvJac]) == hipSuccess) phase_[0] += 1e-3 * ms;
  cost_ = 0.5 * loc[0];
  gmax_ = segmented_ ? loc[1] : std::max(gmax_c, loc[1]);
  xnorm_ = std::sqrt((segmented_ ? 0.0 : x2_c) + loc[2]);
  if (!std::isfinite(cost_)) return SK_ERR_EVALUATION_FAILED;
  return SK_OK;
}

// A time-out of the resident panel chain (info == 2: the device is shared, or its kernels are being serialised) loses
// that factorisation, not the step: the chain is switched off for the device and the same linear system is assembled
// and factored again, launch by launch, in the same iteration — the trajectory does not change.
int BalSolver::try_step(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm) {
  bool chain_lost = false;
  int rc = try_step_once(radius, valid, mcc, new_cost, step_norm, &chain_lost);
  if (rc == SK_OK && chain_lost) rc = try_step_once(radius, valid, mcc, new_cost, step_norm, &chain_lost);
  return rc;
}

int BalSolver::try_step_once(double radius, bool* valid, double* mcc, double* new_cost, double* step_norm, bool* chain_lost) {
  hipStream_t s = stream_;
  *chain_lost = false;
  const size_t nc = 9 * (size_t)C_;
  *valid = false;
  SK_HIP_TRY(hipEventRecord(ev_[kEvBegin], s));
  const bool graph = graph_ok();
  const bool replay = graph && g_step_[parity_] != nullptr;
  bool candidate_failed = false;  // a cost function that cannot be evaluated at the candidate: the step is rejected (cost = max)
  if (graph) {
    h_scal_[32] = radius;  // pinned: the captured host-to-device copy reads it when the graph RUNS
    if (!replay && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); graph_mode_ = false; return try_step_once(radius, valid, mcc, new_cost, step_norm, chain_lost); }
  }
  CaptureGuard capture(s, graph && !replay, &graph_mode_);
  if (!replay) {
  // the LM diagonal is formed where it is used (bal_lm_diag): the radius travels in the kernel arguments, or — a replayed graph —
  // through device memory
  d_.lm_lo = opt_.min_lm_diagonal; d_.lm_hi = opt_.max_lm_diagonal; d_.lm_radius = radius; d_.lm_radius_dev = nullptr;
  if (graph) {
    SK_HIP_TRY(hipMemcpyAsync(b_scal_.p + 12, h_scal_ + 32, sizeof(double), hipMemcpyHostToDevice, s));
    d_.lm_radius_dev = b_scal_.p + 12;
  }
  // ---- B. Schur complement assembly ----
  // (this step's back-substitution zeroes what it reads when it is the resident launch: decided here, once, for the whole step)
  // (... and whether the back-substitutions of this step are the resident launch at all: one decision for every front — the process-wide
  // switch can be cleared by another solver's time-out at any moment)
  const int bs_resident = opt_.resident_kernels && cholesky_backsolve_resident(npad_ / 128) ? 1 : 0;
  const bool zero_after = zero_by_backsolve_ && bs_resident;
  kt_.begin("memset_S", s);
  for (int f = 0; f < 3; ++f)
    if (fr_[f].nblk > 0) launch_zero_envelope(d_.front[f].S, (int)fr_[f].dim, (zero_by_backsolve_ && !need_full_zero_) ? b_zero_min_f_[f].p : b_zero_col0_f_[f].p, fr_[f].nblk, s);
  kt_.end("memset_S", s);
  need_full_zero_ = !zero_after;
  SK_HIP_TRY(hipMemsetAsync(b_scal_.p + 14, 0, 2 * sizeof(double), s));  // the failure flag and the factorisation's info
  launch_bal_point_block(d_, s);
  launch_bal_kept_points(d_, s);  // (retained points: their rows of the reduced system; nothing of theirs enters the Schur complement)
  launch_bal_obs_precompute(d_, s);
  kt_.begin("bal_cam_diag", s); launch_bal_cam_diag(d_, s); kt_.end("bal_cam_diag", s);
  kt_.begin("bal_pair", s); launch_bal_pair(d_, s); kt_.end("bal_pair", s);
  if (opt_.allreduce && !segmented_) {  // (segmented: the root front is summed after the leaf has been factored, below)
    // sum S (with the rhs row) over ranks: only its lower block triangle travels (half the bytes)
    launch_tri_pack(d_.S, npad_, b_pack_.p, npad_ / 128, b_pack_col0_.p, b_pack_off_.p, true, s);  // (never dissected here: front[2] is the whole system)
    int rc = allreduce(b_pack_.p, packed_elems_);
    if (rc) return rc;
    launch_tri_pack(d_.S, npad_, b_pack_.p, npad_ / 128, b_pack_col0_.p, b_pack_off_.p, false, s);
  }
  // D_c^2 onto the cameras' diagonal entries; the padded tails of the interiors and of the root are identities, and the
  // augmented right-hand-side row of the root gets a huge diagonal so that its factorisation stays positive definite (the
  // entry itself is unused; in a leaf's border that diagonal stays zero: it is ADDED to the root's)
  auto finish_root = [&]() {
    launch_bal_finish_S(d_, 4, s);
    launch_set_diagonal(d_.front[2].S, (int)fr_[2].dim, fr_[2].rhs_row, fr_[2].rhs_row + 1, 1e300, s);
    launch_set_diagonal(d_.front[2].S, (int)fr_[2].dim, fr_[2].rhs_row + 1, (int)fr_[2].dim, 1.0, s);
  };
  if (!segmented_) {
    // one launch: D^2 of every camera, the identities on the leaf fronts' padding, the root's two diagonal ranges
    BalFinishRanges r;
    int k = 0;
    for (int f = 0; f < 2; ++f)
      if (fr_[f].nblk > 0 && fr_[f].ncols * 128 > 9 * fr_[f].cams) { r.S[k] = d_.front[f].S; r.ld[k] = (int)fr_[f].dim; r.from[k] = 9 * fr_[f].cams; r.to[k] = fr_[f].ncols * 128; r.value[k] = 1.0; ++k; }
    r.S[k] = d_.front[2].S; r.ld[k] = (int)fr_[2].dim; r.from[k] = fr_[2].rhs_row; r.to[k] = fr_[2].rhs_row + 1; r.value[k] = 1e300; ++k;
    if ((int)fr_[2].dim > fr_[2].rhs_row + 1) { r.S[k] = d_.front[2].S; r.ld[k] = (int)fr_[2].dim; r.from[k] = fr_[2].rhs_row + 1; r.to[k] = (int)fr_[2].dim; r.value[k] = 1.0; ++k; }
    launch_bal_finish_all(d_, r, s);
  } else {
    launch_bal_finish_S(d_, 3, s);
    for (int f = 0; f < 2; ++f)
      if (fr_[f].nblk > 0) launch_set_diagonal(d_.front[f].S, (int)fr_[f].dim, 9 * fr_[f].cams, fr_[f].ncols * 128, 1.0, s);
    // (segmented: finish_root() after the ranks' root fronts have been summed — D^2 and the diagonals are added once)
  }
  if (!graph) SK_HIP_TRY(hipEventRecord(ev_[kEvAssemble], s));
  // ---- C. dense Cholesky + solves ----
  CholeskyContext* ctx = opt_.lookahead ? &chol_ctx_ : nullptr;
  int* bs_info = bs_resident ? info_p_ : nullptr;  // (nullptr: the back-substitutions one launch per block step — nothing resident, nothing that waits)
  double* yf[3] = {b_yf_.p + fr_[0].y_off, b_yf_.p + fr_[1].y_off, b_yf_.p + fr_[2].y_off};
  double* wf[3] = {b_wf_.p + fr_[0].y_off, b_wf_.p + fr_[1].y_off, b_wf_.p + fr_[2].y_off};
  // (the resident back-substitutions find their "not there yet" pattern in every front's y already: one fill here, in front of the
  // factorisation, instead of one in front of each of them, between the factorisation's end and the first hop)
  const bool prefilled = bs_resident != 0;
  if (prefilled) SK_HIP_TRY(hipMemsetAsync(b_yf_.p, 0xff, (fr_[0].dim + fr_[1].dim + fr_[2].dim) * sizeof(double), s));
  if (segmented_) {
    // this rank's segment: factor its interior, leave its Schur complement on the separators next to it; sum the root
    // fronts over the ranks (the separators' own blocks come from whichever rank owns the point, the Schur complements
    // from the segments on either side); then every rank factors the same root and solves its own interior
    const FrontView& L = leaf_;
    const FrontHost& R = fr_[2];
    double* Rs = d_.front[2].S;
    double* RLinv = b_Linv_.p + R.linv_off;
    if (L.ncols > 0) {
      cholesky_factor(L.S, L.ld, L.nblk * 128, L.Linv, info_p_, group_, s, ctx, &kt_, L.last, chain_ok(), L.ncols, L.tail_rows, nullptr, L.tail);
      cholesky_border_add(Rs, (long)R.dim, L.S, L.ld, L.ncols, L.nblk - L.ncols, b_leaf_map_.p, s);
    }
    if (replica_) SK_HIP_TRY(hipMemsetAsync(b_pack_.p, 0, packed_elems_ * sizeof(double), s));  // a replica adds nothing
    else launch_tri_pack(Rs, (int)R.dim, b_pack_.p, R.nblk, b_pack_col0_.p, b_pack_off_.p, true, s);
    int rc = allreduce(b_pack_.p, packed_elems_);
    if (rc) return rc;
    launch_tri_pack(Rs, (int)R.dim, b_pack_.p, R.nblk, b_pack_col0_.p, b_pack_off_.p, false, s);
    finish_root();
    cholesky_factor(Rs, (long)R.dim, (int)R.dim, RLinv, info_p_, group_, s, ctx, &kt_, R.env(), chain_ok(), -1, 1, nullptr, R.tl());
    cholesky_backsolve(Rs, (long)R.dim, 9 * R.cams, (int)R.dim, R.rhs_row, RLinv, wf[2], yf[2], s, &kt_, R.env(), bs_info, R.tl(), zero_after, bs_resident, prefilled);
    if (L.ncols > 0) {
      cholesky_gather_map(yf[2], b_leaf_gmap_.p, b_ybB_.p, (L.nblk - L.ncols) * 128, s);
      cholesky_backsolve_front(L.S, L.ld, L.nblk, L.ncols, L.rhs_row, L.Linv, b_ybB_.p, wf[0], yf[0], s, L.last, L.spike, L.tail_rows, bs_info, zero_after, L.tail, nullptr, bs_resident, prefilled);
    }
  } else if (dissected_) {
    cholesky_dissected_factor(ds_, info_p_, group_, s, ctx, &chol_ctx_b_, &kt_, &kt_b_, chain_ok());
    cholesky_dissected_backsolve(ds_, 9 * fr_[2].cams, wf[2], yf[2], wf[0], yf[0], wf[1], yf[1], b_ybB_.p, s, &chol_ctx_b_, &kt_, bs_info, zero_after, bs_resident, prefilled);
  } else {
    const FrontHost& R = fr_[2];
    cholesky_factor(d_.front[2].S, (long)R.dim, (int)R.dim, b_Linv_.p, info_p_, group_, s, ctx, &kt_, R.env(), chain_ok(), -1, 1, nullptr, R.tl());
    cholesky_backsolve(d_.front[2].S, (long)R.dim, n_, (int)R.dim, R.rhs_row, b_Linv_.p, wf[2], yf[2], s, &kt_, R.env(), bs_info, R.tl(), zero_after, bs_resident, prefilled);
  }
  if (!graph) SK_HIP_TRY(hipEventRecord(ev_[kEvChol], s));
  // ---- D. back-substitution, candidate point ----
  {
    // whose cameras' steps this rank accounts for in |step|^2: all, or (segmented) its segment's, + the separator's on the head's rank
    int lo = 0, hi = (int)nc, lo2 = 0, hi2 = 0;
    if (segmented_) {
      lo = 9 * my_lo_; hi = 9 * my_hi_;
      if (role_ == 0) { lo2 = 9 * cam_b_; hi2 = (int)nc; }
    }
    launch_bal_backsub(d_, b_scal_.p + 8, lo, hi, lo2, hi2, s);  // (|delta_c|^2 to slot 8, |delta_p|^2 to slot 9)
  }
  if (!graph) SK_HIP_TRY(hipEventRecord(ev_[kEvBacksub], s));
  kt_.begin("bal_eval_cost", s);
  if (tape_mode_) launch_bal_eval_cost_tape(d_, tape_dev_, s); else launch_bal_eval_cost(d_, s);
  kt_.end("bal_eval_cost", s);
  int nb_cost = bal_partial_blocks(N_);
  if (d_.num_host > 0) {
    int rc = host_callbacks(d_.xc_new, false, &candidate_failed);
    if (rc) return rc;
    nb_cost += launch_bal_host_cost(d_, nb_cost, s);
  }
  launch_final_reduce(b_partial_.p, partial_stride_, nb_cost, 2, 0, b_scal_.p, s);
  if (!graph) SK_HIP_TRY(hipEventRecord(ev_[kEvCost], s));
  SK_HIP_TRY(hipMemcpyAsync(h_scal_, b_scal_.p, 16 * sizeof(double), hipMemcpyDeviceToHost, s));  // scalars 0-9, the two flags in 14 and 15
  }
  const bool dev_gather = rank_table_on_device(4) && !graph;  // (see evaluate_with_jacobian)
  if (dev_gather) { int rc = enqueue_rank_table(1, 4); if (rc) return rc; }
  if (graph) {
    if (!replay) { capture.release(); int rc = finish_capture(s, &g_step_[parity_]); if (rc) return rc; if (!graph_mode_) return try_step_once(radius, valid, mcc, new_cost, step_norm, chain_lost); }
    SK_HIP_TRY(hipGraphLaunch(g_step_[parity_], s));
    SK_HIP_TRY(hipEventRecord(ev_[kEvCost], s));
  }
  SK_HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  if (graph) {  // one replayed graph: no events inside it — the whole linear solve + candidate evaluation is reported as "factor"
    if (hipEventElapsedTime(&ms, ev_[kEvBegin], ev_[kEvCost]) == hipSuccess) phase_[2] += 1e-3 * ms;
  } else {
    if (hipEventElapsedTime(&ms, ev_[kEvBegin], ev_[kEvAssemble]) == hipSuccess) phase_[1] += 1e-3 * ms;
    if (hipEventElapsedTime(&ms, ev_[kEvAssemble], ev_[kEvChol]) == hipSuccess) phase_[2] += 1e-3 * ms;
    if (hipEventElapsedTime(&ms, ev_[kEvChol], ev_[kEvBacksub]) == hipSuccess) phase_[3] += 1e-3 * ms;
    if (hipEventElapsedTime(&ms, ev_[kEvBacksub], ev_[kEvCost]) == hipSuccess) phase_[4] += 1e-3 * ms;
  }
  int fail = 0, info = 0;
  std::memcpy(&fail, h_scal_ + 14, sizeof(int)); std::memcpy(&info, h_scal_ + 15, sizeof(int));
  if (info != 0) need_full_zero_ = true;  // (a pivot that was not positive, a wait that gave up: whatever the back-substitution did, start from a clean envelope)
  if (cholesky_note_info(opt_.lookahead ? &chol_ctx_ : nullptr, info) && !opt_.allreduce) {  // factor again, launch by launch
    if (graph_mode_) {
      // a replayed graph would launch the resident kernel that has just timed out again and again (the choice is made at capture): the
      // iteration is enqueued launch by launch from here on
      for (hipGraphExec_t* g : {&g_step_[0], &g_step_[1], &g_eval_[0], &g_eval_[1]}) if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
      graph_mode_ = false;
    }
    *chain_lost = true;
    return SK_OK;
  }
  // sum r_new^2, model term, |delta_p|^2 (segmented: + this rank's cameras' |delta_c|^2, which no other rank has), failure
  double loc[4] = {h_scal_[0], h_scal_[1], h_scal_[9] + (segmented_ ? h_scal_[8] : 0.0), (double)(fail | info)};
  const int ops4[4] = {0, 0, 0, 1};
  int rc = dev_gather ? fold_rank_table(loc, 4, ops4) : gather_rank_scalars(loc, 4, ops4);
  if (rc) return rc;
  if (opt_.allreduce && loc[3] >= 2.0) cholesky_disable_chain(&chol_ctx_);  // a rank's chain timed out (info == 2): launch by launch on every rank from here on
  const double step_sq = (segmented_ ? 0.0 : h_scal_[8]) + loc[2];
  if (loc[3] != 0.0 || !std::isfinite(step_sq) || !std::isfinite(loc[1])) return SK_OK;  // invalid step
  *valid = true;
  *mcc = -loc[1];
  *new_cost = candidate_failed ? std::numeric_limits<double>::max() : 0.5 * loc[0];
  *step_norm = std::sqrt(step_sq);
  return SK_OK;
}

// End the capture on `s` and instantiate what was captured.  A runtime that cannot capture this sequence switches the
// replay off for good (graph_mode_ = false; the caller then enqueues the launches directly), with one line on stderr.
int BalSolver::finish_capture(hipStream_t s, hipGraphExec_t* exec) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s, &g);
  if (e == hipSuccess && g) e = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
  if (g) (void)hipGraphDestroy(g);
  if (e != hipSuccess || !*exec) {
    (void)hipGetLastError();
    *exec = nullptr;
    graph_mode_ = false;
    std::fprintf(stderr, "[skeres_amd] hipGraph capture of the iteration failed (%s): launches are enqueued one by one\n", hipGetErrorString(e));
  }
  return SK_OK;
}

// The reference's director upcall (ceres.i:48; CORE/AutodiffCostFunction.scala:74-78) for the residual blocks whose cost
// function has no device body: parameters down to the host, the caller's Evaluate once per block with the exact native
// signature (jacobians == nullptr on the cost-only branch), the rows back up.  Slow by construction — one PCIe round
// trip per evaluation and host arithmetic — but any generic (9, 3) -> 2 functor can enter DENSE_SCHUR this way.
int BalSolver::host_callbacks(const double* x_dev, bool jac, bool* failed) {
  *failed = false;
  const size_t nc = 9 * (size_t)C_, np = 3 * (size_t)P_;
  SK_HIP_TRY(hipMemcpyAsync(host_x_.data(), x_dev, (nc + np) * sizeof(double), hipMemcpyDeviceToHost, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  for (size_t h = 0; h < host_obs_.size(); ++h) {
    const int o = host_obs_[h];
    const CostFunction* cf = host_cf_[h];
    const double* params[2] = {&host_x_[9 * (size_t)h_cam_[o]], &host_x_[nc + 3 * (size_t)h_pt_[o]]};
    double* row = &host_rows_h_[h * (size_t)kHostRow];
    for (int k = 0; k < kHostRow; ++k) row[k] = 0.0;
    if (res_size_ == 2 && cam_size_ == 9 && pt_size_ == 3) {
      double* jptr[2] = {row + 2, row + 20};
      if (!cf->callback(cf->user, params, row, jac ? jptr : nullptr)) { *failed = true; return SK_OK; }
    } else {
      // a smaller shape: the caller's Evaluate writes r residuals and row-major r x c / r x q blocks (CORE/AutodiffCostFunction.scala:115-130);
      // they go into the (2; 9, 3) row the kernels read, the rest of it zero
      double res[2] = {0.0, 0.0}, jc[2 * 9], jq[2 * 3];
      double* jptr[2] = {jc, jq};
      if (!cf->callback(cf->user, params, res, jac ? jptr : nullptr)) { *failed = true; return SK_OK; }
      for (int r = 0; r < res_size_; ++r) {
        row[r] = res[r];
        if (jac) {
          for (int k = 0; k < cam_size_; ++k) row[2 + 9 * r + k] = jc[r * cam_size_ + k];
          for (int k = 0; k < pt_size_; ++k) row[20 + 3 * r + k] = jq[r * pt_size_ + k];
        }
      }
    }
  }
  SK_HIP_TRY(hipMemcpyAsync(b_host_rows_.p, host_rows_h_.data(), host_rows_h_.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));  // (host_rows_h_ is pageable and reused by the next evaluation)
  return SK_OK;
}

int BalSolver::write_back() {
  const size_t nc = 9 * (size_t)C_, np = 3 * (size_t)P_;
  std::vector<double> x(nc + np);
  SK_HIP_TRY(hipMemcpyAsync(x.data(), d_.xc, (nc + np) * sizeof(double), hipMemcpyDeviceToHost, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  if (segmented_) {
    // a rank's cameras of the OTHER segment were never updated: every camera from the rank that owns it (the separator's
    // from rank 0; replicas add zeros), summed into a zero-filled table
    std::vector<double> cams(nc, 0.0);
    if (!replica_) {
      const size_t lo = 9 * (size_t)my_lo_, hi = 9 * (size_t)my_hi_;
      std::memcpy(&cams[lo], &x[lo], (hi - lo) * sizeof(double));
      if (role_ == 0) std::memcpy(&cams[9 * (size_t)cam_b_], &x[9 * (size_t)cam_b_], (nc - 9 * (size_t)cam_b_) * sizeof(double));
    }
    DevBuf<double> tmpc;
    SK_HIP_TRY(tmpc.upload(cams, stream_));
    int rc = allreduce(tmpc.p, cams.size());
    if (rc) return rc;
    SK_HIP_TRY(hipMemcpyAsync(x.data(), tmpc.p, nc * sizeof(double), hipMemcpyDeviceToHost, stream_));
    SK_HIP_TRY(hipStreamSynchronize(stream_));
  }
  for (int i = 0; i < C_; ++i) if (cam_block_[i] >= 0) std::memcpy(problem_->block_ptr[cam_block_[i]], &x[9 * (size_t)i], cam_size_ * sizeof(double));
  if (!opt_.allreduce) {
    for (int q = 0; q < P_; ++q) std::memcpy(problem_->block_ptr[pt_block_[local_pt_[q]]], &x[nc + 3 * (size_t)q], pt_size_ * sizeof(double));
    return SK_OK;
  }
  // every rank returns ALL points: zero-filled table, own slice filled, sum-reduced
  std::vector<double> all(3 * (size_t)P_total_, 0.0);
  if (!replica_)  // (a replica adds zeros; so does a copy of a retained point whose home is another rank: the copies come last)
    for (int q = 0; q < P_own_; ++q) std::memcpy(&all[3 * (size_t)local_pt_[q]], &x[nc + 3 * (size_t)q], 3 * sizeof(double));
  DevBuf<double> tmp;
  SK_HIP_TRY(tmp.upload(all, stream_));
  int rc = allreduce(tmp.p, all.size());
  if (rc) return rc;
  SK_HIP_TRY(hipMemcpyAsync(all.data(), tmp.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
  SK_HIP_TRY(hipStreamSynchronize(stream_));
  for (int q = 0; q < P_total_; ++q) std::memcpy(problem_->block_ptr[pt_block_[q]], &all[3 * (size_t)q], pt_size_ * sizeof(double));
  return SK_OK;
}

}  // namespace

// The segmented distribution's plan as the solver derives it (BalSolver::setup), from host data alone: for every residual
// block the segment its camera belongs to (0 .. segments - 1; -k for a camera of separator k, 1 <= k < segments) and the rank
// that owns its point.  Returns the number of segments (1: the sequence was not cut).  Rank-invariant by construction: the
// same code every rank runs.
int bal_segment_plan(const Problem& p, int max_segments, bool forced, std::vector<int>* block_camera_part, std::vector<int>* block_point_owner) {
  std::vector<int> cam_block, pt_block, ocam, opt;
  bal_index_problem(p, &cam_block, &pt_block, &ocam, &opt);
  const int C = (int)cam_block.size(), P = (int)pt_block.size();
  const int npad = ((9 * C + 1 + 127) / 128) * 128, nblk = npad / 128;
  std::vector<int> env;
  double flops = 0.0;
  int best_k = 0;
  const std::vector<std::vector<int>> cand = camera_order_candidates(p, cam_block, ocam, opt, C, P, true);  // (as the ranks do when their address orders agree)
  choose_camera_order(cand, ocam, opt, C, P, npad, &best_k, &env, &flops);
  for (int& c : ocam) c = cand[best_k][c];
  std::vector<int> first_col;
  (void)envelope_of_order(ocam, opt, [&] { std::vector<int> e(C); std::iota(e.begin(), e.end(), 0); return e; }(), C, P, nblk, &first_col);
  const Segments sg = choose_segments(ocam, opt, C, P, nblk, env, first_col, max_segments, forced);
  const int R = (int)sg.a.size() + 1;
  std::vector<int> part(C, 0);  // per camera of the banded numbering
  for (int c = 0; c < C; ++c) {
    int seg = 0;
    for (int k = 0; k < R - 1; ++k) {
      if (c >= sg.a[k] && c < sg.b[k]) { seg = -(k + 1); break; }
      if (c >= sg.b[k]) seg = k + 1;
    }
    part[c] = seg;
  }
  std::vector<int> seg_of_pt(P, -1);
  for (size_t b = 0; b < ocam.size(); ++b) if (part[ocam[b]] >= 0) seg_of_pt[opt[b]] = part[ocam[b]];
  block_camera_part->resize(ocam.size()); block_point_owner->resize(ocam.size());
  for (size_t b = 0; b < ocam.size(); ++b) {
    (*block_camera_part)[b] = part[ocam[b]];
    (*block_point_owner)[b] = seg_of_pt[opt[b]] >= 0 ? seg_of_pt[opt[b]] : opt[b] % R;
  }
  return R;
}

// The camera order and the border of loop-closure cameras as BalSolver::setup derives them (one process), from host data alone.
// final_index_of_block[b]: the position of residual block b's camera inside the reduced system; returns the number of border cameras.
int bal_border_plan(const Problem& p, int mode, std::vector<int>* final_index_of_block, int* gap, double* model_us, double* plain_us, double* fill) {
  std::vector<int> cam_block, pt_block, ocam, opt;
  bal_index_problem(p, &cam_block, &pt_block, &ocam, &opt);
  const int C = (int)cam_block.size(), P = (int)pt_block.size();
  const int npad = ((9 * C + 1 + 127) / 128) * 128, nblk = npad / 128;
  const CamGraph g0{&ocam, &opt, C, P};
  const CameraOrderPlan plan = plan_camera_order(p, cam_block, g0, g0, npad, true, mode != SK_BORDER_OFF, mode);
  final_index_of_block->resize(ocam.size());
  for (size_t b = 0; b < ocam.size(); ++b) (*final_index_of_block)[b] = plan.id[ocam[b]];
  if (gap) *gap = plan.bordered ? plan.border.gap : 0;
  if (model_us) *model_us = plan.bordered ? plan.border.model_us : plan.border.plain_us;
  if (plain_us) *plain_us = plan.border.plain_us;
  if (fill) {
    double in = 0.0;
    for (int c = 0; c < nblk; ++c) {
      const int lm = std::min(plan.last[c], nblk - 1), t0 = plan.tail.empty() ? nblk - 1 : plan.tail[c];
      in += (lm - c + 1) + std::max(0, nblk - std::max(t0, lm + 1));
    }
    *fill = in / (0.5 * nblk * (nblk + 1.0));
  }
  return plan.bordered ? plan.border.border_cams : 0;
}

int bal_retained_plan(const Problem& p, int mode, int max_points, int border_mode, std::vector<int>* retained_of_block, double* model_us, double* model_us_without, bool with_memory_order) {
  std::vector<int> cam_block, pt_block, ocam, opt;
  bal_index_problem(p, &cam_block, &pt_block, &ocam, &opt);
  const int C = (int)cam_block.size(), P = (int)pt_block.size();
  const ReducedSystemPlan rp = plan_reduced_system(p, cam_block, ocam, opt, C, P, with_memory_order, border_mode != SK_BORDER_OFF, border_mode, mode, max_points);
  std::vector<char> kept(P, 0);
  for (int q : rp.retained) kept[q] = 1;
  retained_of_block->resize(ocam.size());
  for (size_t b = 0; b < ocam.size(); ++b) (*retained_of_block)[b] = kept[opt[b]];
  if (model_us) *model_us = rp.order.model_us;
  if (model_us_without) *model_us_without = rp.without_us;
  return (int)rp.retained.size();
}

std::unique_ptr<SolverBase> make_bal_solver(const Options& o, Problem* p) { return std::unique_ptr<SolverBase>(new BalSolver(o, p)); }

}  // namespace sk
