// Device-side view of one rank's shard of a bundle-adjustment-shaped problem
// and the launchers of bal_kernels.hip.  See bal_kernels.hip for the layout.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "loss.hpp"

namespace sk {

constexpr int kLongSegment = 32;  // entries from which a camera pair gets a wave of its own (bal_pair_long_kernel)
// doubles per observation record of What: 27 of F^T (E M^T), then rt (2), then padding — 256 bytes, two whole 128-byte lines, so
// that a gathered record costs exactly its own bytes (at 224 bytes a record straddled 2.75 lines on average).  Records are in
// CAMERA-major order (record e belongs to observation cam_obs[e]): the diagonal-block kernel streams them, and a pair
// segment's two gathers each walk one camera's records in ascending order.
constexpr int kWs = 32;
constexpr int kWu = 28;  // ... of which the pair kernels read the first 28 (27 used)
struct BalDev {
  int C, P, N;            // cameras (all, replicated), LOCAL points, LOCAL observations
  int res_size, cam_size, pt_size;  // the problem's own block shape (r; c, q) <= (2; 9, 3) (bal_solver.hip: bal_block_shape); smaller shapes are padded
  // structure (built once on the host, point-major observation order)
  const int* cam;         // [N] camera of observation o
  const int* pt;          // [N] local point of observation o
  const double* obs;      // [2][N] observed x, y (a recorded functor: its captured doubles, [num captured][N])
  const int* pt_start;    // [P+1] observations of point p are [pt_start[p], pt_start[p+1])
  const int* cam_start;   // [C+1] CSR into cam_obs
  const int* cam_obs;     // [N] observation indices of each camera, ascending point
  // pair lists for the off-diagonal blocks of S
  int num_segments;
  const int* seg_start;   // [num_segments+1]
  const int* seg_row;     // [num_segments] row camera i
  const int* seg_col;     // [num_segments] col camera j (< i)
  int num_short_segments, num_long_segments;  // segments with fewer / at least kLongSegment entries
  const int* short_segments;  // [num_short_segments] segment ids
  const int* long_segments;   // [num_long_segments]
  const int* pair_row_obs;  // [num_pairs] observation of camera i
  const int* pair_col_obs;  // [num_pairs] observation of camera j
  // Retained points (bal_solver.hip, choose_retained_points): local points that are NOT eliminated — their three coordinates are rows
  // of the reduced system, three points to a PSEUDO-camera (a camera index of the layout with no observations and no parameters:
  // pseudo[i] != 0).  kept_pt[k]: the local point; kept_cam[k] = 3 * pseudo-camera + slot.  num_kept == 0 / pseudo == nullptr: none.
  int num_kept;
  const int* kept_pt;  const int* kept_cam;
  int num_kept_obs;  const int* kept_obs;  const int* kept_obs_slot;  // the retained points' observations, and which retained point (index into kept_pt) each belongs to
  const unsigned char* pseudo;  // [C]
  // a SEGMENTED world of ranks: a retained point's observations are split over the ranks by camera, every rank with some of them holds a copy;
  // kept_home[k] = 1 on the ONE rank that adds the point's own terms (D_p^2, the right-hand side, its norms); kept_global[k]: its index among all
  // retained points (where its column norms and gradient travel in the small all-reduce).  nullptr: every local retained point is at home.
  const int* kept_home;  const int* kept_global;
  // Two residual blocks on the SAME (camera, point) pair (the reference's set-up loop adds whatever the file holds,
  // EX/SimpleBundleAdjuster.scala:139-145, and Ceres accepts it): every sum over observations takes both as they stand; what differs is
  // the Schur complement's cross term between the two, which belongs to the camera's DIAGONAL block — such pairs are not in the pair
  // lists (one writer per block), bal_dup_diag_kernel adds  -(What_a What_b^T + What_b What_a^T)  behind bal_cam_diag.  Slots of the
  // camera-major records.  Usually none.
  int num_dup;  const int* dup_a;  const int* dup_b;  const int* dup_cam;
  // state
  double* xc;  double* xp;          // current parameters [9C], [3P]
  double* xc_new;  double* xp_new;  // candidate
  double* scale_c;  double* scale_p;  // Jacobi scaling
  double* colsq_c;  double* colsq_p;  // squared column norms of the scaled Jacobian
  double* gs_c;  double* gs_p;        // scaled gradient J_s^T r
  // LM diagonal D_j = sqrt(clamp(colsq_j, lm_lo, lm_hi) / radius) (LevenbergMarquardtStrategy), formed where it is used — the points'
  // Schur blocks, the cameras' diagonal entries — since round 4 (a launch of its own over every coordinate until then).  The radius is
  // lm_radius, or *lm_radius_dev when that is set (hipGraph replay: the launch must not change from one iteration to the next)
  double lm_lo, lm_hi, lm_radius;  const double* lm_radius_dev;
  double* step_c;  double* step_p;    // step in scaled space
  double* y_c;                        // reduced-system solution
  // per-observation planes
  double* r;  double* F;  double* E;  double* What;  double* u;  // u [5][N]: E^T F y_c per observation (back-substitution), then F y_c itself (2)
  // CAMERA-major copy of what the per-camera kernels read of an observation: record e (the e-th entry of the camera CSR,
  // i.e. observation cam_obs[e]) holds F row 0 (9), F row 1 (9), r (2).  The planes above are point-major — lane o streams —
  // and a wave that walks a camera's observations through them fetches 20 scattered 64-byte sectors per observation for
  // 160 bytes used; through the records it streams.  Written by bal_cam_records_kernel after every Jacobian evaluation.
  double* Fcam;  const int* obs_slot;  // [N][kFcam]; slot of observation o in the camera CSR (the inverse of cam_obs)
  // per-point
  double* M;  double* q;
  // Reduced camera system: up to three dense row-major matrices ("fronts", chol_kernels.hip "Two-way dissection"), lower
  // triangles.  Part of camera i: 2 (separators, the root front) when i >= cam_b; 0 when seg_lo <= i < cam_a (one device:
  // the head, seg_lo == 0; a rank of a segmented world: ITS segment); otherwise 1 — one device: the tail [cam_a, cam_b), in
  // its elimination order, back to front along the sequence; a rank of a segmented world: the other ranks' segments, which
  // have no front here (front[1].S == nullptr) and no observation among this rank's.  Block (i, j), i >= j, lives in the
  // front of camera j's part at column 9 (j - first camera of the part); its rows are camera i's interior rows when i is in
  // the same part, else (i is a separator camera) row border_row[i - cam_b] of that front — the leaf's border, laid out as
  // chol_kernels.hpp's SegmentLayout says.  The right-hand side of a part's cameras is the front's row `rhs_row`.
  // Without dissection seg_lo == cam_a == cam_b == 0: every camera is "separator" and front[2] is the whole system.
  struct Front { double* S; int ld; int interior; int rhs_row; const int* border_row; };
  Front front[3];
  int seg_lo, cam_a, cam_b;
  const double* y_front[3];  // the fronts' solutions (interior order) -> y_c (bal_gather_y)
  double* S;  int ld;  int rhs_row;  // == front[2] when the system is not dissected (the all-reduce packs this one)
  // reductions
  double* partial;  int partial_stride;
  int* fail_flag;
  // robust loss (loss.hpp): loss_root < 0 = every residual block has the trivial loss; otherwise the root of the one loss all
  // blocks share, or — loss_of_obs != nullptr: the blocks have different losses (CORE/Problem.scala:20 takes one per block) —
  // any valid root (it only selects the kernels with a loss) and per observation its own root, -1 = trivial
  const LossNode* loss_nodes;  int loss_root;  const int* loss_of_obs;
  // residual blocks whose cost function is the caller's host code (the reference's director path, ceres.i:48:
  // sk_cost_function_new_callback with 2 residuals over a 9- and a 3-block): evaluated on the host, their rows uploaded
  const unsigned char* is_host;  // [N] 1 = host-evaluated observation; nullptr when there is none
  int num_host;
  const int* host_obs;           // [num_host] observation index
  const double* host_rows;       // [num_host][kHostRow]: r (2), d r / d camera (2 x 9 row-major), d r / d point (2 x 3 row-major)
};
constexpr int kHostRow = 26;
constexpr int kFcam = 20;

int bal_partial_blocks(int N);
int bal_point_blocks(int P);  // workgroups of the per-point kernels (bal_point_backsub writes one partial sum each)
// the part of the lower block triangle of S inside its block envelope as one contiguous run: block row kb holds its
// 128 rows x (kb + 1 - col0[kb]) * 128 columns row-major; this is what travels in the all-reduce
void launch_zero_envelope(double* S, int ld, const int* col0, int nblk, hipStream_t s);
size_t tri_packed_elems(int nblk);
// col0[kb]: first block column of block row kb that travels; off[kb]: where the row starts in the packed buffer (elements)
void launch_tri_pack(double* S, int ld, double* packed, int nblk, const int* col0, const long long* off, bool to_packed, hipStream_t s);
void launch_bal_eval_jac(const BalDev& d, hipStream_t s);
void launch_bal_eval_cost(const BalDev& d, hipStream_t s);
// the same two for a recorded functor (tape.hpp); bal_tape_width == 0: its register file does not fit the LDS
int bal_tape_width(const Tape& t);
void launch_bal_eval_jac_tape(const BalDev& d, const TapeDevBuffers& tb, hipStream_t s);
void launch_bal_eval_cost_tape(const BalDev& d, const TapeDevBuffers& tb, hipStream_t s);
// the uploaded rows of the host-evaluated observations -> r / F / E planes (loss correction and column scaling as the
// device functors' kernel applies them); cost partial sums from slot `partial_off` on.  Returns the number of slots.
int launch_bal_host_jac(const BalDev& d, int partial_off, hipStream_t s);
// candidate cost and model term of the host-evaluated observations from their uploaded residuals at the candidate point
int launch_bal_host_cost(const BalDev& d, int partial_off, hipStream_t s);
void launch_bal_scale_jac(const BalDev& d, hipStream_t s);
void launch_bal_cam_records(const BalDev& d, hipStream_t s);  // F, r planes -> camera-major records (Fcam)
void launch_bal_reduce(const BalDev& d, hipStream_t s);  // colsq and gs of cameras (from the camera-major records) and points, one launch
void launch_jacobi_scale(const double* colsq, double* scale, int n, hipStream_t s);
void launch_apply_scale_to_reductions(double* colsq, double* gs, const double* scale, int n, hipStream_t s);
void launch_lm_diagonal(const double* colsq, double* D, int n, double lo, double hi, double radius, hipStream_t s);
int launch_grad_max_xnorm(const double* gs, const double* scale, const double* x, int n, double* partial, int stride, hipStream_t s);
void launch_final_reduce(const double* partial, int stride, int count, int K, int maxmask, double* out, hipStream_t s);
void launch_bal_pack_rank_scalars(const double* scal, double* table, int rank, int world, int mode, bool segmented, hipStream_t s);  // (bal_pack_rank_scalars_kernel)
struct ReduceRows { int n = 0; int row[4] = {0, 0, 0, 0}, count[4] = {0, 0, 0, 0}, is_max[4] = {0, 0, 0, 0}; double* out[4] = {nullptr, nullptr, nullptr, nullptr}; };
void launch_final_reduce_rows(const double* partial, int stride, const ReduceRows& rows, hipStream_t s);  // up to four reductions of different lengths, one launch
void launch_bal_point_block(const BalDev& d, hipStream_t s);
void launch_bal_kept_points(const BalDev& d, hipStream_t s);
// the retained points' column norms and gradient (3 + 3 doubles each) to / from their slots of a buffer [colsq (3 K) | gs (3 K)], K = all retained points
void launch_bal_kept_sums(const BalDev& d, double* buf, int K, bool gather, hipStream_t s);  // after bal_point_block: the retained points' rows of the reduced system; M = 0, q = 0 for them
void launch_bal_obs_precompute(const BalDev& d, hipStream_t s);
void launch_bal_cam_diag(const BalDev& d, hipStream_t s);  // (+ the cross terms of duplicate (camera, point) pairs, when there are any)
void launch_bal_pair(const BalDev& d, hipStream_t s);
void launch_finish_normal_matrix(double* S, int ld, int n, int npad, int rhs_row, const double* D, hipStream_t s);
void launch_bal_finish_S(const BalDev& d, int parts, hipStream_t s);  // D_c^2 onto the diagonal entries of the cameras of `parts` (bit mask)
struct BalFinishRanges { double* S[4] = {nullptr, nullptr, nullptr, nullptr}; int ld[4] = {0, 0, 0, 0}, from[4] = {0, 0, 0, 0}, to[4] = {0, 0, 0, 0}; double value[4] = {0, 0, 0, 0}; };
void launch_bal_finish_all(const BalDev& d, const BalFinishRanges& r, hipStream_t s);  // ... of every camera, + S[k][j][j] = value[k] for from[k] <= j < to[k]
void launch_set_diagonal(double* S, int ld, int from, int to, double value, hipStream_t s);  // S[j][j] = value, from <= j < to
// y_c from the fronts' solutions, the cameras' step (out[0] = |delta_c|^2 over [norm_lo, norm_hi) and [norm_lo2, norm_hi2)), the points'
// back-substitution and step (out[1] = |delta_p|^2)
void launch_bal_backsub(const BalDev& d, double* out, int norm_lo, int norm_hi, int norm_lo2, int norm_hi2, hipStream_t s);

}  // namespace sk
