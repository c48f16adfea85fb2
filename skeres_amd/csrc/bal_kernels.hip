// HIP kernels (gfx950) for the bundle-adjustment-shaped hot path:
//   A. per-residual-block Jet autodiff  (replaces the director upcall into
//      CORE/AutodiffCostFunction.scala:74-134 + EX/SimpleBundleAdjuster.scala:79-119)
//   B. Schur-complement assembly        (replaces [ext] ceres SchurEliminator<2,3,9>)
//   D. back-substitution + candidate cost (replaces [ext] ceres BackSubstitute / Evaluate)
//
// Data layout in HBM (all fp64, structure-of-arrays so that lane o touches
// element o of each plane: every wave access is a contiguous 512-B run):
//   observations are stored POINT-MAJOR (all observations of a local point are
//   contiguous, ascending camera) — obs index `o` below is that order.
//   r   [2][N]   residuals                      F  [18][N]  d r / d camera  (row-major 2x9 per obs)
//   E   [6][N]   d r / d point (2x3)            What [N][32] F^T (E M^T) (9x3) + rt = r - E T^-1 g (2): one 256-byte record per
//                                                            observation, CAMERA-major (bal_kernels.hpp: kWs)
//   u   [3][N]   E^T F y_c (back-substitution)
// Jacobi column scaling is folded into F / E as they are written.
#include <hip/hip_runtime.h>
#include "functors.hpp"
#include "bal_kernels.hpp"

namespace sk {

static constexpr int kBlock = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Where the reduced camera system keeps block (camera i rows, camera j columns), i >= j, and the right-hand side of
// camera i (BalDev::front).
__device__ __forceinline__ int bal_part(const BalDev& d, int i) { return i >= d.cam_b ? 2 : ((i >= d.seg_lo && i < d.cam_a) ? 0 : 1); }
__device__ __forceinline__ int bal_pos(const BalDev& d, int i, int part) { return 9 * (i - (part == 0 ? d.seg_lo : (part == 1 ? d.cam_a : d.cam_b))); }
__device__ __forceinline__ double* bal_block(const BalDev& d, int i, int j, int* ld) {
  const int pj = bal_part(d, j), pi = bal_part(d, i);
  const BalDev::Front& f = d.front[pj];
  const int row = pi == pj ? bal_pos(d, i, pi) : f.border_row[i - d.cam_b];
  *ld = f.ld;
  return f.S + (size_t)row * f.ld + bal_pos(d, j, pj);
}
__device__ __forceinline__ double* bal_rhs(const BalDev& d, int i) {
  const int pi = bal_part(d, i);
  const BalDev::Front& f = d.front[pi];
  return f.S + (size_t)f.rhs_row * f.ld + bal_pos(d, i, pi);
}

// LM diagonal of a coordinate from its squared column norm (BalDev::lm_*): the same operations, in the same order, as lm_diagonal_kernel
__device__ __forceinline__ double bal_lm_diag(const BalDev& d, double colsq) {
  const double radius = d.lm_radius_dev ? d.lm_radius_dev[0] : d.lm_radius;
  return sqrt(fmin(fmax(colsq, d.lm_lo), d.lm_hi) / radius);
}

// rho(s) of observation o's loss: the one loss of the problem, or (mixed losses: BalDev::loss_of_obs) the observation's own —
// a trivial one among robust ones is rho(s) = s, whose corrector is the identity.
__device__ __forceinline__ void bal_loss_eval(const BalDev& d, size_t o, double s, double rho[3]) {
  const int root = d.loss_of_obs ? d.loss_of_obs[o] : d.loss_root;
  if (root < 0) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; }
  loss_evaluate(d.loss_nodes, root, s, rho);
}

// Block-level sum of up to 4 values; result valid in thread 0.
// The per-point kernels give a point kPointLanes adjacent lanes, which take its observations in turn and are summed
// in a fixed tree: with a lane per point, the points seen by a hundred cameras set the length of the whole launch
// (a wave runs its longest lane's loop; Ladybug-shaped: 141 / 120 / 177 us for the three kernels, most of it that tail).
constexpr int kPointLanes = 8;
__device__ __forceinline__ double point_lanes_sum(double v) {
#pragma unroll
  for (int off = 1; off < kPointLanes; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* out_partial, int nblocks_stride) {
  __shared__ double sh[K][kBlock / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = wave_sum(v[k]);
    if (lane == 0) sh[k][w] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int i = 0; i < kBlock / 64; ++i) s += sh[k][i];
      out_partial[(size_t)k * nblocks_stride + blockIdx.x] = s;
    }
  }
}

// ---------------------------------------------------------------------------
// A. residuals + Jacobians, T = Jet<12>, seeding order camera[0..9) then
//    point[0..3)  (AutodiffCostFunction.scala:96-106); Jacobian blocks row-major
//    kNumResiduals x N(i) (:115-130), stored as SoA planes.
// ---------------------------------------------------------------------------
// kLoss: robust loss (loss.hpp) — the block's residuals and both Jacobian blocks are corrected before they
// are stored, the cost term is rho(|r|^2).  A separate instantiation: the trivial-loss kernel stays as it was.
// (launch bounds: three waves per SIMD.  Left to itself the kernel takes 170 VGPRs — two waves; asked for three it fits 168 without a
// byte of scratch, and the kernel is bound by how many waves are in flight, not by its arithmetic — round 4)
template <bool kLoss>
__global__ __launch_bounds__(kBlock, kLoss ? 2 : 3) void bal_eval_jac_kernel(BalDev d) {
  double acc[1] = {0.0};
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < d.N; o += gridDim.x * kBlock) {
    if (d.is_host && d.is_host[o]) continue;  // the caller's host code evaluates this one (bal_host_jac_kernel)
    const int ci = d.cam[o], pi = d.pt[o];
    typedef Jet<12> J;
    J cam[9], X[3], out[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) cam[k] = J(d.xc[9 * (size_t)ci + k], k);
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = J(d.xp[3 * (size_t)pi + k], 9 + k);
    const J* params[2] = {cam, X};
    const double c[2] = {d.obs[o], d.obs[(size_t)d.N + o]};
    SnavelyReprojectionError::apply<J>(c, params, out);
    const double r0 = out[0].a, r1 = out[1].a;
    if (kLoss) {
      const double sq = r0 * r0 + r1 * r1;
      double rho[3];
      bal_loss_eval(d, o, sq, rho);
      const LossCorrector lc(sq, rho);
      acc[0] += rho[0];
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const double j0 = out[0].v[k], j1 = out[1].v[k];
        const double rtj = lc.alpha_sq_norm * (r0 * j0 + r1 * j1);
        out[0].v[k] = lc.sqrt_rho1 * (j0 - r0 * rtj);
        out[1].v[k] = lc.sqrt_rho1 * (j1 - r1 * rtj);
      }
      d.r[o] = r0 * lc.residual_scaling;
      d.r[(size_t)d.N + o] = r1 * lc.residual_scaling;
    } else {
      d.r[o] = r0;
      d.r[(size_t)d.N + o] = r1;
      acc[0] += r0 * r0 + r1 * r1;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double s = d.scale_c[9 * (size_t)ci + k];
      d.F[(size_t)k * d.N + o] = out[0].v[k] * s;
      d.F[(size_t)(9 + k) * d.N + o] = out[1].v[k] * s;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double s = d.scale_p[3 * (size_t)pi + k];
      d.E[(size_t)k * d.N + o] = out[0].v[9 + k] * s;
      d.E[(size_t)(3 + k) * d.N + o] = out[1].v[9 + k] * s;
    }
  }
  block_sum<1>(acc, d.partial, d.partial_stride);
}
// (Round 4 tried two things on this kernel, neither kept.  (i) The camera-major records of (F, r) written here, staged in LDS and
// stored by the wave, instead of by bal_cam_records_kernel in a launch of its own: 148-165 us against 99 + 50 — the kernel runs two
// waves per SIMD on its 172 registers, and the staging and the store loop are serial work on each of them, where the separate
// transposition has the whole chip's occupancy.  (ii) Forward mode in 12 / W passes of Jet<W>, W = 6 or 4, as the tape
// interpreter does — a Jet component depends on the real parts and on that component of the operands only, so the passes give the
// same Jacobian — to fit more than two waves per SIMD: the compiler keeps the body at 250 VGPRs whatever W, and capped at 128
// it spills 400-800 bytes per lane.)

// Candidate cost at (xc_new, xp_new) with T = double (cost-only branch,
// AutodiffCostFunction.scala:80-93) fused with the model residual J*step that
// the trust-region ratio needs:  m = F s_c + E s_p ; term = m . (r + m/2).
template <bool kLoss>
__global__ __launch_bounds__(kBlock) void bal_eval_cost_kernel(BalDev d) {
  double acc[2] = {0.0, 0.0};
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < d.N; o += gridDim.x * kBlock) {
    if (d.is_host && d.is_host[o]) continue;  // (bal_host_cost_kernel)
    const int ci = d.cam[o], pi = d.pt[o];
    double cam[9], X[3], out[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) cam[k] = d.xc_new[9 * (size_t)ci + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = d.xp_new[3 * (size_t)pi + k];
    const double* params[2] = {cam, X};
    const double c[2] = {d.obs[o], d.obs[(size_t)d.N + o]};
    SnavelyReprojectionError::apply<double>(c, params, out);
    if (kLoss) {
      double rho[3];
      bal_loss_eval(d, o, out[0] * out[0] + out[1] * out[1], rho);
      acc[0] += rho[0];
    } else {
      acc[0] += out[0] * out[0] + out[1] * out[1];
    }
    // F s_c = -(F y_c): the back-substitution formed F y_c per observation (bal_obs_backsub_kernel: d.u planes 3 and 4), in the
    // order the sum over the camera's nine coordinates was formed here until round 4 — the same bits, 128 bytes less to read
    double m0 = -d.u[3 * (size_t)d.N + o], m1 = -d.u[4 * (size_t)d.N + o];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double s = d.step_p[3 * (size_t)pi + k];
      m0 += d.E[(size_t)k * d.N + o] * s;
      m1 += d.E[(size_t)(3 + k) * d.N + o] * s;
    }
    acc[1] += m0 * (d.r[o] + m0 / 2.0) + m1 * (d.r[(size_t)d.N + o] + m1 / 2.0);
  }
  block_sum<2>(acc, d.partial, d.partial_stride);
}

// ---- recorded functors (tape.hpp): the same two kernels with the functor body interpreted ------------------------
// `d.obs` holds the tape's captured doubles as planes [num_obs][N].  Forward mode in 12 / W passes of Jet<W> (W divides
// 12); the registers of the workgroup's threads are the dynamic LDS.  Everything after the functor — loss correction,
// column scaling, the planes written — is the code of bal_eval_jac_kernel, applied column by column.
constexpr int kTapeMaxObs = 8;
template <bool kLoss, int W>
__global__ __launch_bounds__(kBlock) void bal_eval_jac_tape_kernel(BalDev d, TapeDev t, int num_obs) {
  extern __shared__ __attribute__((aligned(16))) double tape_lds[];
  typedef Jet<W> J;
  const TapeRegs<J> regs{tape_lds, kBlock, (int)threadIdx.x};
  double acc[1] = {0.0};
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < d.N; o += gridDim.x * kBlock) {
    if (d.is_host && d.is_host[o]) continue;
    const int ci = d.cam[o], pi = d.pt[o];
    double c[kTapeMaxObs];
    for (int k = 0; k < num_obs; ++k) c[k] = d.obs[(size_t)k * d.N + o];
    // (the functor's own block sizes — at most (2; 9, 3): flattened parameter k is camera coordinate k, or point coordinate k - cam_size)
    const int cs = d.cam_size, dim = d.cam_size + d.pt_size;
    auto param = [&](int k) { return k < cs ? d.xc[9 * (size_t)ci + k] : d.xp[3 * (size_t)pi + (k - cs)]; };
    double r0 = 0.0, r1 = 0.0, sqrt_rho1 = 1.0, alpha_sq_norm = 0.0;
    for (int first = 0; first < dim; first += W) {
      J out[2];
      out[1] = J(0.0);  // (a functor with one residual: the second row stays zero)
      tape_run<J>(t, c, param, first, regs, out);
      if (first == 0) {
        r0 = out[0].a; r1 = out[1].a;
        if (kLoss) {
          const double sq = r0 * r0 + r1 * r1;
          double rho[3];
          bal_loss_eval(d, o, sq, rho);
          const LossCorrector lc(sq, rho);
          acc[0] += rho[0];
          sqrt_rho1 = lc.sqrt_rho1; alpha_sq_norm = lc.alpha_sq_norm;
          d.r[o] = r0 * lc.residual_scaling;
          d.r[(size_t)d.N + o] = r1 * lc.residual_scaling;
        } else {
          d.r[o] = r0;
          d.r[(size_t)d.N + o] = r1;
          acc[0] += r0 * r0 + r1 * r1;
        }
      }
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const int k = first + w;
        if (k >= dim) continue;  // (the last pass of a shape whose dimension W does not divide)
        double j0 = out[0].v[w], j1 = out[1].v[w];
        if (kLoss) {
          const double rtj = alpha_sq_norm * (r0 * j0 + r1 * j1);
          const double a0 = sqrt_rho1 * (j0 - r0 * rtj), a1 = sqrt_rho1 * (j1 - r1 * rtj);
          j0 = a0; j1 = a1;
        }
        if (k < cs) {
          const double s = d.scale_c[9 * (size_t)ci + k];
          d.F[(size_t)k * d.N + o] = j0 * s;
          d.F[(size_t)(9 + k) * d.N + o] = j1 * s;
        } else {
          const double s = d.scale_p[3 * (size_t)pi + (k - cs)];
          d.E[(size_t)(k - cs) * d.N + o] = j0 * s;
          d.E[(size_t)(3 + k - cs) * d.N + o] = j1 * s;
        }
      }
    }
  }
  block_sum<1>(acc, d.partial, d.partial_stride);
}
template <bool kLoss>
__global__ __launch_bounds__(kBlock) void bal_eval_cost_tape_kernel(BalDev d, TapeDev t, int num_obs) {
  extern __shared__ __attribute__((aligned(16))) double tape_lds[];
  const TapeRegs<double> regs{tape_lds, kBlock, (int)threadIdx.x};
  double acc[2] = {0.0, 0.0};
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < d.N; o += gridDim.x * kBlock) {
    if (d.is_host && d.is_host[o]) continue;
    const int ci = d.cam[o], pi = d.pt[o];
    double c[kTapeMaxObs], out[2];
    for (int k = 0; k < num_obs; ++k) c[k] = d.obs[(size_t)k * d.N + o];
    const int cs = d.cam_size;
    auto param = [&](int k) { return k < cs ? d.xc_new[9 * (size_t)ci + k] : d.xp_new[3 * (size_t)pi + (k - cs)]; };
    out[1] = 0.0;
    tape_run<double>(t, c, param, 0, regs, out);
    if (kLoss) {
      double rho[3];
      bal_loss_eval(d, o, out[0] * out[0] + out[1] * out[1], rho);
      acc[0] += rho[0];
    } else {
      acc[0] += out[0] * out[0] + out[1] * out[1];
    }
    // F s_c = -(F y_c): the back-substitution formed F y_c per observation (bal_obs_backsub_kernel: d.u planes 3 and 4), in the
    // order the sum over the camera's nine coordinates was formed here until round 4 — the same bits, 128 bytes less to read
    double m0 = -d.u[3 * (size_t)d.N + o], m1 = -d.u[4 * (size_t)d.N + o];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double s = d.step_p[3 * (size_t)pi + k];
      m0 += d.E[(size_t)k * d.N + o] * s;
      m1 += d.E[(size_t)(3 + k) * d.N + o] * s;
    }
    acc[1] += m0 * (d.r[o] + m0 / 2.0) + m1 * (d.r[(size_t)d.N + o] + m1 / 2.0);
  }
  block_sum<2>(acc, d.partial, d.partial_stride);
}

// Host-evaluated observations (director path): what the caller's Evaluate wrote — residuals and the two row-major
// Jacobian blocks, CORE/AutodiffCostFunction.scala:113-130 — goes through the same loss correction and column scaling
// as the device functors' output and into the same planes; everything downstream cannot tell the difference.
template <bool kLoss>
__global__ __launch_bounds__(kBlock) void bal_host_jac_kernel(BalDev d, int partial_off) {
  double acc[1] = {0.0};
  for (int h = blockIdx.x * kBlock + threadIdx.x; h < d.num_host; h += gridDim.x * kBlock) {
    const int o = d.host_obs[h];
    const int ci = d.cam[o], pi = d.pt[o];
    const double* row = d.host_rows + (size_t)h * kHostRow;
    const double r0 = row[0], r1 = row[1];
    double j0[12], j1[12];
#pragma unroll
    for (int k = 0; k < 9; ++k) { j0[k] = row[2 + k]; j1[k] = row[11 + k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { j0[9 + k] = row[20 + k]; j1[9 + k] = row[23 + k]; }
    if (kLoss) {
      const double sq = r0 * r0 + r1 * r1;
      double rho[3];
      bal_loss_eval(d, o, sq, rho);
      const LossCorrector lc(sq, rho);
      acc[0] += rho[0];
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const double a = j0[k], b = j1[k];
        const double rtj = lc.alpha_sq_norm * (r0 * a + r1 * b);
        j0[k] = lc.sqrt_rho1 * (a - r0 * rtj);
        j1[k] = lc.sqrt_rho1 * (b - r1 * rtj);
      }
      d.r[o] = r0 * lc.residual_scaling;
      d.r[(size_t)d.N + o] = r1 * lc.residual_scaling;
    } else {
      d.r[o] = r0;
      d.r[(size_t)d.N + o] = r1;
      acc[0] += r0 * r0 + r1 * r1;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double s = d.scale_c[9 * (size_t)ci + k];
      d.F[(size_t)k * d.N + o] = j0[k] * s;
      d.F[(size_t)(9 + k) * d.N + o] = j1[k] * s;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double s = d.scale_p[3 * (size_t)pi + k];
      d.E[(size_t)k * d.N + o] = j0[9 + k] * s;
      d.E[(size_t)(3 + k) * d.N + o] = j1[9 + k] * s;
    }
  }
  block_sum<1>(acc, d.partial + partial_off, d.partial_stride);
}

template <bool kLoss>
__global__ __launch_bounds__(kBlock) void bal_host_cost_kernel(BalDev d, int partial_off) {
  double acc[2] = {0.0, 0.0};
  for (int h = blockIdx.x * kBlock + threadIdx.x; h < d.num_host; h += gridDim.x * kBlock) {
    const int o = d.host_obs[h];
    const int ci = d.cam[o], pi = d.pt[o];
    const double n0 = d.host_rows[(size_t)h * kHostRow], n1 = d.host_rows[(size_t)h * kHostRow + 1];  // residuals at the candidate point
    if (kLoss) {
      double rho[3];
      bal_loss_eval(d, o, n0 * n0 + n1 * n1, rho);
      acc[0] += rho[0];
    } else {
      acc[0] += n0 * n0 + n1 * n1;
    }
    // F s_c = -(F y_c): the back-substitution formed F y_c per observation (bal_obs_backsub_kernel: d.u planes 3 and 4), in the
    // order the sum over the camera's nine coordinates was formed here until round 4 — the same bits, 128 bytes less to read
    double m0 = -d.u[3 * (size_t)d.N + o], m1 = -d.u[4 * (size_t)d.N + o];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double s = d.step_p[3 * (size_t)pi + k];
      m0 += d.E[(size_t)k * d.N + o] * s;
      m1 += d.E[(size_t)(3 + k) * d.N + o] * s;
    }
    acc[1] += m0 * (d.r[o] + m0 / 2.0) + m1 * (d.r[(size_t)d.N + o] + m1 / 2.0);
  }
  block_sum<2>(acc, d.partial + partial_off, d.partial_stride);
}

// In-place column scaling of F / E (only at iteration 0, when the Jacobi
// scale is first known).
__global__ __launch_bounds__(kBlock) void bal_scale_jac_kernel(BalDev d) {
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < d.N; o += gridDim.x * kBlock) {
    const int ci = d.cam[o], pi = d.pt[o];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double s = d.scale_c[9 * (size_t)ci + k];
      d.F[(size_t)k * d.N + o] *= s;
      d.F[(size_t)(9 + k) * d.N + o] *= s;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double s = d.scale_p[3 * (size_t)pi + k];
      d.E[(size_t)k * d.N + o] *= s;
      d.E[(size_t)(3 + k) * d.N + o] *= s;
    }
  }
}

// ---------------------------------------------------------------------------
// Column reductions.  Camera columns: one wave per camera over its observation
// list (fixed lane assignment + fixed butterfly => run-to-run reproducible).
//   colsq_c = sum F^2 (squared column norm), gs_c = sum F^T r (scaled gradient)
// ---------------------------------------------------------------------------
// F, r planes (point-major) -> camera-major records.  A wave takes 64 consecutive observations: coalesced plane reads,
// the 64 records staged in LDS, then written 3.2 records per store instruction (each record's 160 bytes contiguous).
__global__ __launch_bounds__(kBlock) void bal_cam_records_kernel(BalDev d) {
  __shared__ __attribute__((aligned(16))) double stage[kBlock / 64][64 * kFcam];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double* st = stage[w];
  const size_t N = d.N;
  for (long o0 = ((long)blockIdx.x * (kBlock / 64) + w) * 64; o0 < d.N; o0 += (long)gridDim.x * kBlock) {
    const long o = o0 + lane;
    const int my_slot = o < d.N ? d.obs_slot[o] : 0;
    if (o < d.N) {
#pragma unroll
      for (int k = 0; k < 18; ++k) st[lane * kFcam + k] = d.F[(size_t)k * N + o];
      st[lane * kFcam + 18] = d.r[o];
      st[lane * kFcam + 19] = d.r[N + o];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    const int nrec = (int)((d.N - o0) < 64 ? (d.N - o0) : 64);
    // ten lanes to a record, 16 bytes each: 6.4 records per store instruction, the slots passed from lane to lane
#pragma unroll
    for (int i = 0; i < kFcam / 2; ++i) {
      const int e2 = i * 64 + lane, rr = e2 / (kFcam / 2), f2 = e2 - rr * (kFcam / 2);
      const int slot = __shfl(my_slot, rr, 64);
      if (rr < nrec) reinterpret_cast<double2*>(d.Fcam + (size_t)slot * kFcam)[f2] = reinterpret_cast<const double2*>(st)[e2];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Round 4: the cameras' and the points' column reductions are ONE launch — workgroups [0, cam_blocks) take the cameras, the rest the
// points (bal_reduce_kernel below): the two do not depend on each other, and side by side they take as long as the longer one.
__device__ __forceinline__ void bal_cam_reduce_body(const BalDev& d, int block) {
  const int wave = (block * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= d.C) return;
  double sq[9], g[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { sq[k] = 0.0; g[k] = 0.0; }
  for (int e = d.cam_start[wave] + lane; e < d.cam_start[wave + 1]; e += 64) {
    // (the observations of a camera in CSR order, as before: the sums round as before)
    const double2* rec = reinterpret_cast<const double2*>(d.Fcam + (size_t)e * kFcam);
    double v[kFcam];
#pragma unroll
    for (int k = 0; k < kFcam / 2; ++k) { const double2 t = rec[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
    const double r0 = v[18], r1 = v[19];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double f0 = v[k], f1 = v[9 + k];
      sq[k] += f0 * f0 + f1 * f1;
      g[k] += f0 * r0 + f1 * r1;
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) { sq[k] = wave_sum(sq[k]); g[k] = wave_sum(g[k]); }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { d.colsq_c[9 * (size_t)wave + k] = sq[k]; d.gs_c[9 * (size_t)wave + k] = g[k]; }
  }
}

__device__ __forceinline__ void bal_pt_reduce_body(const BalDev& d, int block) {
  const int gid = block * kBlock + threadIdx.x, sub = gid % kPointLanes;
  const int p = gid / kPointLanes;
  if (p >= d.P) return;  // (whole lane groups: kBlock is a multiple of kPointLanes)
  double sq[3] = {0, 0, 0}, g[3] = {0, 0, 0};
  for (int o = d.pt_start[p] + sub; o < d.pt_start[p + 1]; o += kPointLanes) {
    const double r0 = d.r[o], r1 = d.r[(size_t)d.N + o];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double e0 = d.E[(size_t)k * d.N + o], e1 = d.E[(size_t)(3 + k) * d.N + o];
      sq[k] += e0 * e0 + e1 * e1;
      g[k] += e0 * r0 + e1 * r1;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { sq[k] = point_lanes_sum(sq[k]); g[k] = point_lanes_sum(g[k]); }
  if (sub == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { d.colsq_p[3 * (size_t)p + k] = sq[k]; d.gs_p[3 * (size_t)p + k] = g[k]; }
  }
}

__global__ __launch_bounds__(kBlock) void bal_reduce_kernel(BalDev d, int cam_blocks) {
  if ((int)blockIdx.x < cam_blocks) bal_cam_reduce_body(d, (int)blockIdx.x);
  else bal_pt_reduce_body(d, (int)blockIdx.x - cam_blocks);
}

// Generic small vector kernels -------------------------------------------------
// scale_j = 1 / (1 + sqrt(colsq_j))   (Jacobi scaling, fixed at iteration 0)
// A coordinate that is held constant (SetParameterBlockConstant, SubsetParameterization) carries scale 0 from set-up on:
// its Jacobian column is written as zeros and the step applied to it is step * 0 — the parameter keeps its bits.  Its
// row of the normal equations is min_lm_diagonal / radius on the diagonal and zero elsewhere: the reduced program's
// system with an inert extra unknown (oracle/lm.cpp, Bal::cam_mask).
__global__ void jacobi_scale_kernel(const double* colsq, double* scale, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) scale[j] = scale[j] == 0.0 ? 0.0 : 1.0 / (1.0 + sqrt(colsq[j]));
}
// After scaling J in place, colsq and the scaled gradient follow algebraically.
__global__ void apply_scale_to_reductions_kernel(double* colsq, double* gs, const double* scale, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) { const double s = scale[j]; colsq[j] = colsq[j] * s * s; gs[j] = gs[j] * s; }
}
// D_j = sqrt(clamp(colsq_j, lo, hi) / radius)   (LevenbergMarquardtStrategy)
__global__ void lm_diagonal_kernel(const double* colsq, double* D, int n, double lo, double hi, double radius) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) D[j] = sqrt(fmin(fmax(colsq[j], lo), hi) / radius);
}
// partial[0] = max |gs_j / scale_j| , partial[1] = sum x_j^2 over this block
__global__ __launch_bounds__(kBlock) void grad_max_xnorm_kernel(const double* gs, const double* scale, const double* x,
                                                                int n, double* partial, int stride) {
  double m = 0.0, s = 0.0;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    if (scale[j] != 0.0) m = fmax(m, fabs(gs[j] / scale[j]));  // (a constant coordinate has no gradient entry)
    s += x[j] * x[j];
  }
  __shared__ double shm[kBlock / 64], shs[kBlock / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { m = fmax(m, __shfl_xor(m, off, 64)); s += __shfl_xor(s, off, 64); }
  if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = m; shs[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < kBlock / 64; ++i) { m = fmax(m, shm[i]); s += shs[i]; }
    partial[blockIdx.x] = m;
    partial[stride + blockIdx.x] = s;
  }
}

// out[k] = reduce(partial[k*stride .. +count)) in fixed order; bit k of maxmask: 0 sum, 1 max
// One workgroup per value; fixed lane assignment + fixed tree => reproducible.
__global__ __launch_bounds__(kBlock) void final_reduce_kernel(const double* partial, int stride, int count, int K, int maxmask, double* out) {
  __shared__ double sh[kBlock];
  const int k = blockIdx.x;
  const bool is_max = ((maxmask >> k) & 1) != 0;
  double a = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) {
    const double v = partial[(size_t)k * stride + i];
    a = is_max ? fmax(a, v) : a + v;
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] = is_max ? fmax(sh[threadIdx.x], sh[threadIdx.x + w]) : sh[threadIdx.x] + sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = sh[0];
}

// ---------------------------------------------------------------------------
// B. Schur elimination of the point blocks.
//   T_p = sum E^T E + D_p^2 ; M = chol(T)^-1 (lower) so T^-1 = M^T M ; q = T^-1 g
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bal_point_block_kernel(BalDev d) {
  const int gid = blockIdx.x * kBlock + threadIdx.x, sub = gid % kPointLanes;
  const int p = gid / kPointLanes;
  if (p >= d.P) return;
  double t00 = 0, t10 = 0, t11 = 0, t20 = 0, t21 = 0, t22 = 0;
  for (int o = d.pt_start[p] + sub; o < d.pt_start[p + 1]; o += kPointLanes) {
    const double a0 = d.E[o], a1 = d.E[(size_t)d.N + o], a2 = d.E[2 * (size_t)d.N + o];
    const double b0 = d.E[3 * (size_t)d.N + o], b1 = d.E[4 * (size_t)d.N + o], b2 = d.E[5 * (size_t)d.N + o];
    t00 += a0 * a0 + b0 * b0; t10 += a1 * a0 + b1 * b0; t11 += a1 * a1 + b1 * b1;
    t20 += a2 * a0 + b2 * b0; t21 += a2 * a1 + b2 * b1; t22 += a2 * a2 + b2 * b2;
  }
  t00 = point_lanes_sum(t00); t10 = point_lanes_sum(t10); t11 = point_lanes_sum(t11);
  t20 = point_lanes_sum(t20); t21 = point_lanes_sum(t21); t22 = point_lanes_sum(t22);
  if (sub != 0) return;
  const double d0 = bal_lm_diag(d, d.colsq_p[3 * (size_t)p]), d1 = bal_lm_diag(d, d.colsq_p[3 * (size_t)p + 1]), d2 = bal_lm_diag(d, d.colsq_p[3 * (size_t)p + 2]);
  t00 += d0 * d0; t11 += d1 * d1; t22 += d2 * d2;
  // Cholesky T = L L^T
  const double l00 = sqrt(t00);
  const double l10 = t10 / l00, l20 = t20 / l00;
  const double l11 = sqrt(t11 - l10 * l10);
  const double l21 = (t21 - l20 * l10) / l11;
  const double l22 = sqrt(t22 - l20 * l20 - l21 * l21);
  // M = L^-1
  const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
  const double m10 = -(l10 * m00) / l11;
  const double m21 = -(l21 * m11) / l22;
  const double m20 = -(l20 * m00 + l21 * m10) / l22;
  const size_t P = d.P;
  d.M[p] = m00; d.M[P + p] = m10; d.M[2 * P + p] = m11; d.M[3 * P + p] = m20; d.M[4 * P + p] = m21; d.M[5 * P + p] = m22;
  // q = M^T (M g)
  const double g0 = d.gs_p[3 * (size_t)p], g1 = d.gs_p[3 * (size_t)p + 1], g2 = d.gs_p[3 * (size_t)p + 2];
  const double u0 = m00 * g0, u1 = m10 * g0 + m11 * g1, u2 = m20 * g0 + m21 * g1 + m22 * g2;
  d.q[p] = m00 * u0 + m10 * u1 + m20 * u2;
  d.q[P + p] = m11 * u1 + m21 * u2;
  d.q[2 * P + p] = m22 * u2;
  if (!(l00 > 0.0) || !(l11 > 0.0) || !(l22 > 0.0)) *d.fail_flag = 1;  // not positive definite
}

// Retained points (BalDev::kept_pt).  Their rows of the reduced system —
//   block (pseudo-camera, camera c of observation o) rows 3 t + a:  (E_o^T F_o)[a][.]      (9 x 3 per observation, transposed into place)
//   diagonal block, rows / columns 3 t ..:                          T = sum E^T E + D_p^2  (lower triangle)
//   right-hand side, entries 3 t ..:                                g_p
// — and M = 0, q = 0, so that what bal_obs_precompute forms for their observations is What = 0, rt = r: nothing of such a point enters
// the Schur complement, and its cameras' own blocks and right-hand sides take F^T F and F^T r as they stand (bal_cam_diag_kernel).
// One launch, two kinds of workgroup: the first `obs_blocks` take a lane per observation of a retained point (kept_obs: a landmark
// seen from 392 cameras is seven waves' worth, not one wave's seven turns) and write the off-diagonal blocks; the others a wave per
// point for T, g, M and q.  Every block has one writer (a camera sees a point once: setup() refuses two residual blocks on one pair).
__global__ __launch_bounds__(kBlock) void bal_kept_points_kernel(BalDev d, int obs_blocks) {
  const size_t N = d.N, P = d.P;
  if ((int)blockIdx.x < obs_blocks) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= d.num_kept_obs) return;
    const int o = d.kept_obs[e], k = d.kept_obs_slot[e];
    const int ic = d.kept_cam[k] / 3, t3 = 3 * (d.kept_cam[k] % 3);
    const int c = d.cam[o];
    const bool below = ic > c;  // the pseudo-camera's rows are below camera c's (else: a border camera that is numbered behind it)
    if (!d.front[bal_part(d, below ? c : ic)].S) return;  // (another rank's front)
    double ea[3], eb[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ea[a] = d.E[(size_t)a * N + o]; eb[a] = d.E[(size_t)(3 + a) * N + o]; }
    int ld;
    double* blk = below ? bal_block(d, ic, c, &ld) : bal_block(d, c, ic, &ld);
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      const double f0 = d.F[(size_t)b * N + o], f1 = d.F[(size_t)(9 + b) * N + o];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double v = ea[a] * f0 + eb[a] * f1;
        if (below) blk[(size_t)(t3 + a) * ld + b] = v; else blk[(size_t)b * ld + t3 + a] = v;
      }
    }
    return;
  }
  const int k = (((int)blockIdx.x - obs_blocks) * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (k >= d.num_kept) return;  // wave-uniform
  const int p = d.kept_pt[k], ic = d.kept_cam[k] / 3, t3 = 3 * (d.kept_cam[k] % 3);
  double t00 = 0, t10 = 0, t11 = 0, t20 = 0, t21 = 0, t22 = 0;
  for (int o = d.pt_start[p] + lane; o < d.pt_start[p + 1]; o += 64) {
    const double a0 = d.E[o], a1 = d.E[N + o], a2 = d.E[2 * N + o];
    const double b0 = d.E[3 * N + o], b1 = d.E[4 * N + o], b2 = d.E[5 * N + o];
    t00 += a0 * a0 + b0 * b0; t10 += a1 * a0 + b1 * b0; t11 += a1 * a1 + b1 * b1;
    t20 += a2 * a0 + b2 * b0; t21 += a2 * a1 + b2 * b1; t22 += a2 * a2 + b2 * b2;
  }
  t00 = wave_sum(t00); t10 = wave_sum(t10); t11 = wave_sum(t11); t20 = wave_sum(t20); t21 = wave_sum(t21); t22 = wave_sum(t22);
  if (lane != 0) return;
  d.M[p] = 0.0; d.M[P + p] = 0.0; d.M[2 * P + p] = 0.0; d.M[3 * P + p] = 0.0; d.M[4 * P + p] = 0.0; d.M[5 * P + p] = 0.0;
  d.q[p] = 0.0; d.q[P + p] = 0.0; d.q[2 * P + p] = 0.0;
  if (!d.front[bal_part(d, ic)].S) return;
  // (a segmented world: T is this rank's observations' share — the root front is summed over the ranks — and D_p^2 and the right-hand side,
  // which are the point's own and whose inputs are global after the small all-reduce, come from its home rank alone)
  const bool home = !d.kept_home || d.kept_home[k] != 0;
  const double d0 = home ? bal_lm_diag(d, d.colsq_p[3 * (size_t)p]) : 0.0, d1 = home ? bal_lm_diag(d, d.colsq_p[3 * (size_t)p + 1]) : 0.0,
               d2 = home ? bal_lm_diag(d, d.colsq_p[3 * (size_t)p + 2]) : 0.0;
  int ld;
  double* blk = bal_block(d, ic, ic, &ld) + (size_t)t3 * ld + t3;
  blk[0] = t00 + d0 * d0;
  blk[ld] = t10; blk[ld + 1] = t11 + d1 * d1;
  blk[2 * (size_t)ld] = t20; blk[2 * (size_t)ld + 1] = t21; blk[2 * (size_t)ld + 2] = t22 + d2 * d2;
  double* rhs = bal_rhs(d, ic) + t3;
  rhs[0] = home ? d.gs_p[3 * (size_t)p] : 0.0; rhs[1] = home ? d.gs_p[3 * (size_t)p + 1] : 0.0; rhs[2] = home ? d.gs_p[3 * (size_t)p + 2] : 0.0;
}

// The retained points' column norms and gradient to (gather) / from their slots of buf = [colsq (3 K) | gs (3 K)]: what a segmented world
// sums over its ranks (every rank holds the observations of its own segment's cameras).  A thread per local retained point and coordinate.
__global__ __launch_bounds__(kBlock) void bal_kept_sums_kernel(BalDev d, double* buf, int K, int gather) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 3 * d.num_kept) return;
  const int k = t / 3, a = t - 3 * k;
  const size_t p = (size_t)d.kept_pt[k], g = (size_t)(d.kept_global ? d.kept_global[k] : k);
  if (gather) { buf[3 * g + a] = d.colsq_p[3 * p + a]; buf[3 * (size_t)K + 3 * g + a] = d.gs_p[3 * p + a]; }
  else { d.colsq_p[3 * p + a] = buf[3 * g + a]; d.gs_p[3 * p + a] = buf[3 * (size_t)K + 3 * g + a]; }
}

// Per observation: Ehat = E M^T (2x3), What = F^T Ehat (9x3), rt = r - E q: the observation's 256-byte record, written at
// its slot of the CAMERA-major order (obs_slot).  A lane storing its own record would write 29 doubles 256 bytes apart from
// every other lane's — 64 cache lines per store instruction; the wave's 64 records go through LDS and out sixteen lanes to
// a record, 16 bytes each: four whole records per store instruction.
__global__ __launch_bounds__(kBlock, 4) void bal_obs_precompute_kernel(BalDev d) {
  // Round 4: the records of HALF a wave are staged at a time (the values wait in registers): 8.7 KB of LDS per wave instead of
  // 17.4, four workgroups per CU instead of two — the kernel streams, and what bounds it is the loads in flight.
  constexpr int kLs = kWs + 2;  // record stride in LDS (even: 16-byte reads)
  __shared__ __attribute__((aligned(16))) double stage[kBlock / 64][32 * kLs];
  const size_t N = d.N, P = d.P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* mine = &stage[wave][0];
  // whole waves walk the observations (the last wave may be partly out of range)
  for (long base = (long)(blockIdx.x * kBlock + wave * 64); base < (long)d.N; base += (long)gridDim.x * kBlock) {
    const long o = base + lane;
    const int my_slot = o < (long)d.N ? d.obs_slot[o] : 0;
    double w[27], rt[2];
    if (o < (long)d.N) {
      const int p = d.pt[o];
      const double m00 = d.M[p], m10 = d.M[P + p], m11 = d.M[2 * P + p], m20 = d.M[3 * P + p], m21 = d.M[4 * P + p], m22 = d.M[5 * P + p];
      const double q0 = d.q[p], q1 = d.q[P + p], q2 = d.q[2 * P + p];
      double eh[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double e0 = d.E[(3 * r) * N + o], e1 = d.E[(3 * r + 1) * N + o], e2 = d.E[(3 * r + 2) * N + o];
        eh[r][0] = e0 * m00;
        eh[r][1] = e0 * m10 + e1 * m11;
        eh[r][2] = e0 * m20 + e1 * m21 + e2 * m22;
        rt[r] = d.r[r * N + o] - (e0 * q0 + e1 * q1 + e2 * q2);
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const double f0 = d.F[c * N + o], f1 = d.F[(9 + c) * N + o];
#pragma unroll
        for (int a = 0; a < 3; ++a) w[3 * c + a] = f0 * eh[0][a] + f1 * eh[1][a];
      }
    }
    const int nrec = (int)(((long)d.N - base) < 64 ? ((long)d.N - base) : 64);
    const int sub = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((lane >> 5) == half && o < (long)d.N) {
        double* rec = mine + (lane & 31) * kLs;
#pragma unroll
        for (int k = 0; k < 27; ++k) rec[k] = w[k];
        rec[27] = rt[0]; rec[28] = rt[1];
        rec[29] = 0.0; rec[30] = 0.0; rec[31] = 0.0;  // (the record's padding)
      }
      __builtin_amdgcn_wave_barrier();  // LDS writes of a wave are in order with its reads; keep the compiler from mixing them
#pragma unroll 4
      for (int it = 0; it < 8; ++it) {
        const int rl = 4 * it + sub, rec = 32 * half + rl;  // sixteen lanes to a record, 16 bytes each: four whole records per store instruction
        const int slot = __shfl(my_slot, rec, 64);
        if (rec < nrec) {
          const double2 v = *reinterpret_cast<const double2*>(&mine[rl * kLs + 2 * chunk]);
          *reinterpret_cast<double2*>(d.What + (size_t)slot * kWs + 2 * chunk) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Diagonal blocks + right-hand side, one wave per camera:
//   S_ii = sum_o (F^T F - What What^T)      (D_c^2 is added after the all-reduce)
//   rhs_i = sum_o F^T rt
// rhs goes to row `rhs_row` of S (the augmented row the factorisation carries).
// (Round 4 tried a WORKGROUP per camera — four waves, a quarter of the camera's observations each, summed in wave order through
// LDS — against the cameras with thousands of observations that set the length of the launch: 91-96 us against 66-68; the 54
// wave-level sums at the end of every wave cost more than the long cameras' turns.)
__global__ __launch_bounds__(kBlock) void bal_cam_diag_kernel(BalDev d) {
  const int i = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (i >= d.C) return;
  if (d.pseudo && d.pseudo[i]) return;  // (a pseudo-camera of retained points: its block and right-hand side are bal_kept_points_kernel's)
  double acc[45], rh[9];
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) rh[k] = 0.0;
  for (int e = d.cam_start[i] + lane; e < d.cam_start[i + 1]; e += 64) {
    double f0[9], f1[9], w[28];
    {
      const double2* rec = reinterpret_cast<const double2*>(d.Fcam + (size_t)e * kFcam);  // camera-major: streamed
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const double2 t = rec[k];
        if (2 * k < 9) f0[2 * k] = t.x; else f1[2 * k - 9] = t.x;
        if (2 * k + 1 < 9) f0[2 * k + 1] = t.y; else f1[2 * k + 1 - 9] = t.y;
      }
    }
    const double2* wrec = reinterpret_cast<const double2*>(d.What + (size_t)e * kWs);  // camera-major: streamed
#pragma unroll
    for (int k = 0; k < 14; ++k) { const double2 t = wrec[k]; w[2 * k] = t.x; w[2 * k + 1] = t.y; }
    const double r0 = w[27], r1 = wrec[14].x;
    int k = 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      rh[c] += f0[c] * r0 + f1[c] * r1;
#pragma unroll
      for (int e2 = 0; e2 <= c; ++e2, ++k)
        acc[k] += (f0[c] * f0[e2] + f1[c] * f1[e2]) - (w[3 * c] * w[3 * e2] + w[3 * c + 1] * w[3 * e2 + 1] + w[3 * c + 2] * w[3 * e2 + 2]);
    }
  }
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = wave_sum(acc[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) rh[k] = wave_sum(rh[k]);
  if (lane == 0 && d.front[bal_part(d, i)].S) {  // (a rank of a segmented world holds one leaf front and the root)
    int k = 0, ld;
    double* blk = bal_block(d, i, i, &ld);
    double* rhs = bal_rhs(d, i);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      rhs[c] = rh[c];
#pragma unroll
      for (int e2 = 0; e2 <= c; ++e2, ++k) blk[(size_t)c * ld + e2] = acc[k];
    }
  }
}

// Two observations a, b of ONE point by ONE camera (BalDev::dup_*): the cross term of the point's Schur complement between them,
//   S_ii -= What_a What_b^T + What_b What_a^T   (lower triangle),
// which the pair lists leave out (block (i, i) is bal_cam_diag_kernel's).  A thread per (pair, entry of the lower triangle); pairs
// of one camera are added one after the other (one thread per entry walks them: the same sums in the same order, run after run).
__global__ __launch_bounds__(64) void bal_dup_diag_kernel(BalDev d) {
  const int first = blockIdx.x;  // index of the first pair of a camera's run (the host lists the pairs camera by camera; one workgroup per run)
  const int i = d.dup_cam[first];
  if (first > 0 && d.dup_cam[first - 1] == i) return;
  if (!d.front[bal_part(d, i)].S) return;
  const int t = threadIdx.x;
  if (t >= 45) return;
  int c = 0, rem = t;
  while (rem > c) { rem -= c + 1; ++c; }  // t = c (c + 1) / 2 + e2
  const int e2 = rem;
  double acc = 0.0;
  for (int k = first; k < d.num_dup && d.dup_cam[k] == i; ++k) {
    const double* wa = d.What + (size_t)d.dup_a[k] * kWs;
    const double* wb = d.What + (size_t)d.dup_b[k] * kWs;
    acc += (wa[3 * c] * wb[3 * e2] + wa[3 * c + 1] * wb[3 * e2 + 1] + wa[3 * c + 2] * wb[3 * e2 + 2]) +
           (wb[3 * c] * wa[3 * e2] + wb[3 * c + 1] * wa[3 * e2 + 1] + wb[3 * c + 2] * wa[3 * e2 + 2]);
  }
  int ld;
  double* blk = bal_block(d, i, i, &ld);
  blk[(size_t)c * ld + e2] -= acc;
}

// acc[k] += sum over entries e = e_begin, e_begin + stride, ... < e_end of  What[row obs][3c..3c+2] . What[col obs][3k..3k+2].
// The nine lanes of a block need the same 27 values of the column observation's record: each lane fetches three of them
// (as it does of the row observation's) and the group shares them through `slot` (its 27 doubles of LDS) — fourteen
// 16-byte LDS broadcasts per entry where there were fourteen 16-byte global loads per lane (the texture path was the
// limit: 17 load instructions per wave and entry group; 2.79 -> 2.12 ms on Venice-1778, 361 -> 264 us on Ladybug-1723).  The data of the next entry and the indices
// of the one after it are in flight while the current one is multiplied.
// (Round 4 tried fetching a record by LINE — lanes 0..7 of a group the eight 16-byte chunks of its first 128-byte line, lanes 0..5 the
// six of its second, one line per group and load instruction instead of two, the row record through LDS as well: half the line
// accesses of the texture path, the same bits — and the Schur assembly got SLOWER, 0.38 -> 0.43 ms on Ladybug-1723, 2.43 -> 2.70 on
// Venice-1778: two more LDS writes and three reads per entry cost more than the line accesses saved.)
constexpr int kPairSlot = 34;  // doubles per nine-lane group: 272 bytes, so that the seven groups' 16-byte reads fall on disjoint banks
__device__ __forceinline__ void pair_accumulate(const BalDev& d, int e_begin, int e_end, int stride, int c, double* slot, double (&acc)[9]) {
  if (e_begin >= e_end) return;
  const int last = e_begin + (e_end - 1 - e_begin) / stride * stride;  // the last entry of this lane group
  const double* wr = d.What + (size_t)d.pair_row_obs[e_begin] * kWs + 3 * c;
  const double* wc = d.What + (size_t)d.pair_col_obs[e_begin] * kWs + 3 * c;
  double y0 = wr[0], y1 = wr[1], y2 = wr[2], v0 = wc[0], v1 = wc[1], v2 = wc[2];
  int e1 = e_begin + stride <= last ? e_begin + stride : last;
  int ob1 = d.pair_row_obs[e1], oa1 = d.pair_col_obs[e1];
  for (int e = e_begin; e < e_end; e += stride) {
    slot[3 * c] = v0; slot[3 * c + 1] = v1; slot[3 * c + 2] = v2;
    __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave execute in order: no wait, only no reordering by the compiler)
    double w[kWu];
#pragma unroll
    for (int k = 0; k < kWu / 2; ++k) { const double2 t = reinterpret_cast<const double2*>(slot)[k]; w[2 * k] = t.x; w[2 * k + 1] = t.y; }
    __builtin_amdgcn_wave_barrier();
    const double* wrn = d.What + (size_t)ob1 * kWs + 3 * c;
    const double* wcn = d.What + (size_t)oa1 * kWs + 3 * c;
    const double yn0 = wrn[0], yn1 = wrn[1], yn2 = wrn[2], vn0 = wcn[0], vn1 = wcn[1], vn2 = wcn[2];
    const int e2 = e + 2 * stride <= last ? e + 2 * stride : last;
    const int ob2 = d.pair_row_obs[e2], oa2 = d.pair_col_obs[e2];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] += y0 * w[3 * k] + y1 * w[3 * k + 1] + y2 * w[3 * k + 2];
    y0 = yn0; y1 = yn1; y2 = yn2; v0 = vn0; v1 = vn1; v2 = vn2;
    ob1 = ob2; oa1 = oa2;
  }
}

// Workgroup w of a launch goes to XCD w mod 8, each with an L2 of its own.  The segment lists are in camera order — the
// segments of one row camera follow each other and gather that camera's records over and over — so the launch is cut into
// runs of `group` consecutive logical blocks per XCD: what a row's segments re-read is then in ONE L2 instead of being
// fetched into all eight.  (grid: a multiple of 8 * group; logical blocks beyond the work return.)
__device__ __forceinline__ int xcd_grouped_block(int w, int group) {
  if (group <= 0) return w;
  const int x = w & 7, q = w >> 3;
  return ((q / group) * 8 + x) * group + q % group;
}

// Short segments: seven per wave, nine lanes each, entries in list order.
__global__ __launch_bounds__(kBlock) void bal_pair_kernel(BalDev d, int xcd_group) {
  const int lane = threadIdx.x & 63;
  const int wave = (xcd_grouped_block((int)blockIdx.x, xcd_group) * kBlock + threadIdx.x) >> 6;
  __shared__ __attribute__((aligned(16))) double share[kBlock / 64][7][kPairSlot];
  const int sub = lane / 9, c = lane - 9 * sub;
  const int slot = wave * 7 + sub;
  if (sub >= 7 || slot >= d.num_short_segments) return;
  const int seg = d.short_segments[slot];
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  pair_accumulate(d, d.seg_start[seg], d.seg_start[seg + 1], 1, c, share[threadIdx.x >> 6][sub], acc);
  const int i = d.seg_row[seg], j = d.seg_col[seg];
  int ld;
  double* out = bal_block(d, i, j, &ld) + (size_t)c * ld;
#pragma unroll
  for (int k = 0; k < 9; ++k) out[k] = -acc[k];
}

// Long segments (camera pairs that share many points; a few per cent of the segments, most of the entries):
// one wave per segment.  The seven nine-lane groups take every seventh entry; their partial blocks are summed
// in group order through LDS, so the result does not depend on timing (but the summation order differs from
// the short-segment kernel's list order: which kernel a segment goes to is fixed at set-up).
__global__ __launch_bounds__(kBlock) void bal_pair_long_kernel(BalDev d, int xcd_group) {
  __shared__ double red[kBlock / 64][7][81];
  __shared__ __attribute__((aligned(16))) double share[kBlock / 64][7][kPairSlot];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int slot = (xcd_grouped_block((int)blockIdx.x, xcd_group) * kBlock + threadIdx.x) >> 6;
  if (slot >= d.num_long_segments) return;  // wave-uniform
  const int seg = d.long_segments[slot];
  const int sub = lane / 9, c = lane - 9 * sub;
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  if (sub < 7) {
    pair_accumulate(d, d.seg_start[seg] + sub, d.seg_start[seg + 1], 7, c, share[w][sub], acc);
#pragma unroll
    for (int k = 0; k < 9; ++k) red[w][sub][9 * c + k] = acc[k];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's LDS writes have landed
  __builtin_amdgcn_wave_barrier();
  const int i = d.seg_row[seg], j = d.seg_col[seg];
  int ld;
  double* blk = bal_block(d, i, j, &ld);
  for (int e = lane; e < 81; e += 64) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 7; ++g) s += red[w][g][e];
    blk[(size_t)(e / 9) * ld + e % 9] = -s;
  }
}

// After the (optional) all-reduce: add D_c^2 on the diagonal, make the padded
// tail of S an identity and give the augmented rhs row a huge diagonal so the
// factorisation stays positive definite (its own diagonal entry is unused).
// the same for a single dense normal matrix (dense paths): D^2 on the diagonal, identity on the padded tail, a huge
// diagonal for the augmented right-hand-side row
__global__ void finish_normal_matrix_kernel(double* S, int ld, int n, int npad, int rhs_row, const double* D) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npad) return;
  if (j < n) S[(size_t)j * ld + j] += D[j] * D[j];
  else if (j == rhs_row) S[(size_t)j * ld + j] = 1e300;
  else S[(size_t)j * ld + j] = 1.0;
}
// parts: bit k set = the cameras of part k (0 head, 1 tail, 2 separator / everything when undissected)
__global__ void bal_finish_S_kernel(BalDev d, int parts) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 9 * d.C) return;
  const int i = j / 9, c = j - 9 * i, part = bal_part(d, i);
  if (!((parts >> part) & 1) || !d.front[part].S) return;
  if (d.pseudo && d.pseudo[i]) return;  // (retained points carry their own D_p^2: bal_kept_points_kernel)
  int ld;
  double* blk = bal_block(d, i, i, &ld);
  const double D = bal_lm_diag(d, d.colsq_c[j]);
  blk[(size_t)c * ld + c] += D * D;
}
__global__ void set_diagonal_kernel(double* S, int ld, int from, int to, double value) {
  const int j = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (j < to) S[(size_t)j * ld + j] = value;
}

// ---------------------------------------------------------------------------
// D. back-substitution.  y_c is the reduced-system solution (length 9C).
//   y_p = T^-1 (g_p - sum_o E_o^T (F_o y_c[cam o]))
//   step = -y ; x_new = x + step * scale
// partial[0] += |delta|^2 (points),  one thread per point.
// ---------------------------------------------------------------------------
// Per observation u_o = E_o^T (F_o y_c[cam o]), three point-major planes.  Keeps the per-point pass light, so a
// 400-observation track does not stall its wave.
__global__ __launch_bounds__(kBlock) void bal_obs_backsub_kernel(BalDev d) {
  const size_t N = d.N;
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < d.N; o += gridDim.x * kBlock) {
    const double* yc = d.y_c + 9 * (size_t)d.cam[o];
    double f0 = 0.0, f1 = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) { f0 += d.F[c * N + o] * yc[c]; f1 += d.F[(9 + c) * N + o] * yc[c]; }
    d.u[o] = d.E[o] * f0 + d.E[3 * N + o] * f1;
    d.u[N + o] = d.E[N + o] * f0 + d.E[4 * N + o] * f1;
    d.u[2 * N + o] = d.E[2 * N + o] * f0 + d.E[5 * N + o] * f1;
    d.u[3 * N + o] = f0;  // F y_c itself: the candidate evaluation's model residual starts from it (bal_eval_cost_kernel)
    d.u[4 * N + o] = f1;
  }
}

__global__ __launch_bounds__(kBlock) void bal_point_backsub_kernel(BalDev d) {
  const int gid = blockIdx.x * kBlock + threadIdx.x, sub = gid % kPointLanes;
  const int p = gid / kPointLanes;
  double acc[1] = {0.0};
  double w0 = 0.0, w1 = 0.0, w2 = 0.0;
  if (p < d.P) {
    for (int o = d.pt_start[p] + sub; o < d.pt_start[p + 1]; o += kPointLanes) {
      w0 += d.u[o];
      w1 += d.u[(size_t)d.N + o];
      w2 += d.u[2 * (size_t)d.N + o];
    }
  }
  w0 = point_lanes_sum(w0); w1 = point_lanes_sum(w1); w2 = point_lanes_sum(w2);
  if (p < d.P && sub == 0) {
    const size_t P = d.P;
    const double t0 = d.gs_p[3 * (size_t)p] - w0, t1 = d.gs_p[3 * (size_t)p + 1] - w1, t2 = d.gs_p[3 * (size_t)p + 2] - w2;
    const double m00 = d.M[p], m10 = d.M[P + p], m11 = d.M[2 * P + p], m20 = d.M[3 * P + p], m21 = d.M[4 * P + p], m22 = d.M[5 * P + p];
    const double u0 = m00 * t0, u1 = m10 * t0 + m11 * t1, u2 = m20 * t0 + m21 * t1 + m22 * t2;
    const double y[3] = {m00 * u0 + m10 * u1 + m20 * u2, m11 * u1 + m21 * u2, m22 * u2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double st = -y[k];
      const double dl = st * d.scale_p[3 * (size_t)p + k];
      d.step_p[3 * (size_t)p + k] = st;
      const double xo = d.xp[3 * (size_t)p + k];
      const double xn = xo + dl;
      d.xp_new[3 * (size_t)p + k] = xn;
      const double df = xo - xn;
      acc[0] += df * df;
    }
  }
  block_sum<1>(acc, d.partial + d.partial_stride, d.partial_stride);  // (row 1: row 0 holds the cameras' partial sums, bal_cam_step_kernel)
}

// Retained points: their part of the reduced system's solution is their y_p (bal_point_backsub_kernel left them a zero step: M = 0).
// One workgroup; its |delta_p|^2 goes to slot `slot` of the points' row of partial sums.
__global__ __launch_bounds__(kBlock) void bal_kept_step_kernel(BalDev d, int slot) {
  double acc[1] = {0.0};
  for (int k = threadIdx.x; k < d.num_kept; k += kBlock) {
    const int p = d.kept_pt[k], ic = d.kept_cam[k] / 3, t3 = 3 * (d.kept_cam[k] % 3);
    const bool home = !d.kept_home || d.kept_home[k] != 0;  // (a copy of another rank's point takes the same step; its length is counted once)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double st = -d.y_c[9 * (size_t)ic + t3 + a];
      d.step_p[3 * (size_t)p + a] = st;
      const double xo = d.xp[3 * (size_t)p + a];
      const double xn = xo + st * d.scale_p[3 * (size_t)p + a];
      d.xp_new[3 * (size_t)p + a] = xn;
      const double df = xo - xn;
      if (home) acc[0] += df * df;
    }
  }
  block_sum<1>(acc, d.partial + d.partial_stride + slot, d.partial_stride);
}

// cameras: step_c = -y_c ; xc_new = xc + step_c * scale_c ; out[0] = |delta_c|^2
// (norm_lo, norm_hi: the coordinates whose step this rank accounts for — all of them, or in a segmented world its own
// segment's, the separator's being counted by the head's rank alone)
// Round 4: one thread per coordinate over many workgroups (one workgroup walked the 9 C values in sixteen dependent rounds:
// 17 us), reading the reduced system's solution straight from the fronts (what bal_gather_y_kernel did in a launch of its
// own) and leaving y_c for the back-substitution; the workgroups' partial sums go to row 0 of `partial`, summed in a fixed
// order by the final reduction that follows the points' (launch_bal_point_backsub).
__global__ __launch_bounds__(kBlock) void bal_cam_step_kernel(BalDev d, int norm_lo, int norm_hi, int norm_lo2, int norm_hi2) {
  double acc[1] = {0.0};
  const int n = 9 * d.C;
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j < n) {
    const int i = j / 9, c = j - 9 * i, part = bal_part(d, i);
    const double y = d.front[part].S ? d.y_front[part][bal_pos(d, i, part) + c] : 0.0;  // (another rank's segment: no step here)
    d.y_c[j] = y;
    const double st = -y;
    d.step_c[j] = st;
    const double xo = d.xc[j];
    const double xn = xo + st * d.scale_c[j];
    d.xc_new[j] = xn;
    const double df = xo - xn;
    if ((j >= norm_lo && j < norm_hi) || (j >= norm_lo2 && j < norm_hi2)) acc[0] += df * df;
  }
  block_sum<1>(acc, d.partial, d.partial_stride);
}
// Up to four reductions of different lengths in one launch: workgroup k reduces partial[row[k] * stride .. + count[k]) — a sum or a
// maximum, in a fixed order — into *out[k].
__global__ __launch_bounds__(1024) void final_reduce_rows_kernel(const double* partial, int stride, ReduceRows rows) {
  __shared__ double sh[1024];
  const int k = blockIdx.x, count = rows.count[k];
  const bool is_max = rows.is_max[k] != 0;
  const double* src = partial + (size_t)rows.row[k] * stride;
  double a = 0.0;
  for (int i = threadIdx.x; i < count; i += 1024) a = is_max ? fmax(a, src[i]) : a + src[i];  // (Venice-1778: 31 000 partial sums of the points)
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] = is_max ? fmax(sh[threadIdx.x], sh[threadIdx.x + w]) : sh[threadIdx.x] + sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *rows.out[k] = sh[0];
}
void launch_final_reduce_rows(const double* partial, int stride, const ReduceRows& rows, hipStream_t s) {
  if (rows.n > 0) hipLaunchKernelGGL(final_reduce_rows_kernel, dim3(rows.n), dim3(1024), 0, s, partial, stride, rows);
}

// Lower block triangle of S <-> packed buffer (see bal_kernels.hpp), 16 bytes per lane.
// Block row kb travels from block column col0[kb] to its diagonal block; off[kb] = where it starts in the packed buffer.
__global__ __launch_bounds__(256) void tri_pack_kernel(double* S, int ld, double* packed, const int* __restrict__ col0, const long long* __restrict__ off,
                                                       int to_packed) {
  const int kb = blockIdx.y;
  const size_t c0 = (size_t)col0[kb] * 128, width = (size_t)(kb + 1) * 128 - c0;
  double* pk = packed + off[kb];
  const size_t n2 = 128 * width / 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const size_t e = 2 * i, r = e / width, c = e % width;
    double2* a = reinterpret_cast<double2*>(S + ((size_t)kb * 128 + r) * ld + c0 + c);
    double2* b = reinterpret_cast<double2*>(pk + e);
    if (to_packed) *b = *a; else *a = *b;
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline int grid_for(int n, int cap = 2048) {
  int g = (n + kBlock - 1) / kBlock;
  return g < 1 ? 1 : (g > cap ? cap : g);
}

// S <- 0 on block row kb, columns [col0[kb] * 128, (kb + 1) * 128): the part of the lower block triangle the
// factorisation can read (everything left of it is outside the block envelope, everything right of it is the
// upper triangle)
__global__ __launch_bounds__(256) void zero_envelope_kernel(double* S, int ld, const int* col0) {
  const int kb = blockIdx.y;
  const size_t c0 = (size_t)col0[kb] * 128, width = (size_t)(kb + 1) * 128 - c0;
  const size_t n2 = 128 * width / 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const size_t e = 2 * i, r = e / width, c = e % width;
    *reinterpret_cast<double2*>(S + ((size_t)kb * 128 + r) * ld + c0 + c) = make_double2(0.0, 0.0);
  }
}
void launch_zero_envelope(double* S, int ld, const int* col0, int nblk, hipStream_t s) {
  if (nblk > 0) hipLaunchKernelGGL(zero_envelope_kernel, dim3(32, nblk), dim3(256), 0, s, S, ld, col0);
}

int bal_partial_blocks(int N) { return grid_for(N); }
size_t tri_packed_elems(int nblk) { return (size_t)128 * 128 * ((size_t)nblk * (nblk + 1) / 2); }
void launch_tri_pack(double* S, int ld, double* packed, int nblk, const int* col0, const long long* off, bool to_packed, hipStream_t s) {
  if (nblk > 0) hipLaunchKernelGGL(tri_pack_kernel, dim3(64, nblk), dim3(256), 0, s, S, ld, packed, col0, off, to_packed ? 1 : 0);
}

void launch_bal_eval_jac(const BalDev& d, hipStream_t s) {
  if (d.loss_root >= 0) hipLaunchKernelGGL(bal_eval_jac_kernel<true>, dim3(grid_for(d.N)), dim3(kBlock), 0, s, d);
  else hipLaunchKernelGGL(bal_eval_jac_kernel<false>, dim3(grid_for(d.N)), dim3(kBlock), 0, s, d);
}
void launch_bal_eval_cost(const BalDev& d, hipStream_t s) {
  if (d.loss_root >= 0) hipLaunchKernelGGL(bal_eval_cost_kernel<true>, dim3(grid_for(d.N)), dim3(kBlock), 0, s, d);
  else hipLaunchKernelGGL(bal_eval_cost_kernel<false>, dim3(grid_for(d.N)), dim3(kBlock), 0, s, d);
}
// recorded functors.  W: derivative slots per pass (1, 2 or 3: bal_tape_width)
int bal_tape_width(const Tape& t) { return t.num_obs_consts > kTapeMaxObs ? 0 : tape_pick_width(t, kBlock); }
template <class K> static void bal_tape_allow_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
void launch_bal_eval_jac_tape(const BalDev& d, const TapeDevBuffers& tb, hipStream_t s) {
  const int W = bal_tape_width(tb.host), nobs = tb.host.num_obs_consts;
  const size_t lds = tape_lds_bytes(tb.host, W, kBlock);
  const dim3 g(grid_for(d.N)), b(kBlock);
#define SK_TAPE_JAC(LOSS, WW) { bal_tape_allow_lds(bal_eval_jac_tape_kernel<LOSS, WW>, lds); hipLaunchKernelGGL((bal_eval_jac_tape_kernel<LOSS, WW>), g, b, lds, s, d, tb.view, nobs); }
  if (d.loss_root >= 0) { if (W == 3) SK_TAPE_JAC(true, 3) else if (W == 2) SK_TAPE_JAC(true, 2) else SK_TAPE_JAC(true, 1) }
  else { if (W == 3) SK_TAPE_JAC(false, 3) else if (W == 2) SK_TAPE_JAC(false, 2) else SK_TAPE_JAC(false, 1) }
#undef SK_TAPE_JAC
}
void launch_bal_eval_cost_tape(const BalDev& d, const TapeDevBuffers& tb, hipStream_t s) {
  const size_t lds = tape_lds_bytes(tb.host, 0, kBlock);
  const int nobs = tb.host.num_obs_consts;
  if (d.loss_root >= 0) { bal_tape_allow_lds(bal_eval_cost_tape_kernel<true>, lds); hipLaunchKernelGGL(bal_eval_cost_tape_kernel<true>, dim3(grid_for(d.N)), dim3(kBlock), lds, s, d, tb.view, nobs); }
  else { bal_tape_allow_lds(bal_eval_cost_tape_kernel<false>, lds); hipLaunchKernelGGL(bal_eval_cost_tape_kernel<false>, dim3(grid_for(d.N)), dim3(kBlock), lds, s, d, tb.view, nobs); }
}
int launch_bal_host_jac(const BalDev& d, int partial_off, hipStream_t s) {
  if (d.num_host <= 0) return 0;
  const int g = grid_for(d.num_host);
  if (d.loss_root >= 0) hipLaunchKernelGGL(bal_host_jac_kernel<true>, dim3(g), dim3(kBlock), 0, s, d, partial_off);
  else hipLaunchKernelGGL(bal_host_jac_kernel<false>, dim3(g), dim3(kBlock), 0, s, d, partial_off);
  return g;
}
int launch_bal_host_cost(const BalDev& d, int partial_off, hipStream_t s) {
  if (d.num_host <= 0) return 0;
  const int g = grid_for(d.num_host);
  if (d.loss_root >= 0) hipLaunchKernelGGL(bal_host_cost_kernel<true>, dim3(g), dim3(kBlock), 0, s, d, partial_off);
  else hipLaunchKernelGGL(bal_host_cost_kernel<false>, dim3(g), dim3(kBlock), 0, s, d, partial_off);
  return g;
}
void launch_bal_scale_jac(const BalDev& d, hipStream_t s) { hipLaunchKernelGGL(bal_scale_jac_kernel, dim3(grid_for(d.N)), dim3(kBlock), 0, s, d); }
void launch_bal_cam_records(const BalDev& d, hipStream_t s) { hipLaunchKernelGGL(bal_cam_records_kernel, dim3(grid_for((d.N + 63) / 64 * 64)), dim3(kBlock), 0, s, d); }
// the cameras' and the points' column norms and gradient entries (colsq, gs), one launch
void launch_bal_reduce(const BalDev& d, hipStream_t s) {
  const int cam_blocks = (d.C * 64 + kBlock - 1) / kBlock, pt_blocks = d.P > 0 ? (int)(((long)d.P * kPointLanes + kBlock - 1) / kBlock) : 0;
  hipLaunchKernelGGL(bal_reduce_kernel, dim3(cam_blocks + pt_blocks), dim3(kBlock), 0, s, d, cam_blocks);
}
static int point_grid(int P) { return (int)(((long)P * kPointLanes + kBlock - 1) / kBlock); }
int bal_point_blocks(int P) { return P > 0 ? point_grid(P) : 1; }
void launch_jacobi_scale(const double* colsq, double* scale, int n, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(jacobi_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, colsq, scale, n); }
void launch_apply_scale_to_reductions(double* colsq, double* gs, const double* scale, int n, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(apply_scale_to_reductions_kernel, dim3((n + 255) / 256), dim3(256), 0, s, colsq, gs, scale, n); }
void launch_lm_diagonal(const double* colsq, double* D, int n, double lo, double hi, double radius, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(lm_diagonal_kernel, dim3((n + 255) / 256), dim3(256), 0, s, colsq, D, n, lo, hi, radius); }
int launch_grad_max_xnorm(const double* gs, const double* scale, const double* x, int n, double* partial, int stride, hipStream_t s) {
  const int g = grid_for(n, 256);
  hipLaunchKernelGGL(grad_max_xnorm_kernel, dim3(g), dim3(kBlock), 0, s, gs, scale, x, n, partial, stride);
  return g;
}
// This rank's row of the table of scalars a world of ranks sums (BalSolver::gather_rank_scalars), formed on the device from the
// reductions' results — zeros in the other ranks' rows — so that the all-reduce follows without a round trip through the host.
//   mode 0 (after an evaluation): sum r^2, max |g| of what this rank accounts for, |x|^2 of it
//   mode 1 (after a step):        sum r_new^2, model term, |delta|^2 of what this rank accounts for, failure flags
__global__ void bal_pack_rank_scalars_kernel(const double* __restrict__ scal, double* __restrict__ table, int rank, int world, int mode, int segmented) {
  const int K = mode == 0 ? 3 : 4;
  for (int i = threadIdx.x; i < world * K; i += blockDim.x) table[i] = 0.0;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double* t = table + (size_t)rank * K;
  if (mode == 0) {
    t[0] = scal[4];
    t[1] = segmented ? fmax(scal[2], scal[0]) : scal[2];
    t[2] = scal[3] + (segmented ? scal[1] : 0.0);
  } else {
    const int fail = *reinterpret_cast<const int*>(scal + 14), info = *reinterpret_cast<const int*>(scal + 15);
    t[0] = scal[0]; t[1] = scal[1];
    t[2] = scal[9] + (segmented ? scal[8] : 0.0);
    t[3] = (double)(fail | info);
  }
}
void launch_bal_pack_rank_scalars(const double* scal, double* table, int rank, int world, int mode, bool segmented, hipStream_t s) {
  hipLaunchKernelGGL(bal_pack_rank_scalars_kernel, dim3(1), dim3(64), 0, s, scal, table, rank, world, mode, segmented ? 1 : 0);
}
void launch_final_reduce(const double* partial, int stride, int count, int K, int maxmask, double* out, hipStream_t s) { hipLaunchKernelGGL(final_reduce_kernel, dim3(K), dim3(kBlock), 0, s, partial, stride, count, K, maxmask, out); }
void launch_bal_point_block(const BalDev& d, hipStream_t s) { if (d.P > 0) hipLaunchKernelGGL(bal_point_block_kernel, dim3(point_grid(d.P)), dim3(kBlock), 0, s, d); }
void launch_bal_kept_points(const BalDev& d, hipStream_t s) {
  if (d.num_kept <= 0) return;
  const int obs_blocks = (d.num_kept_obs + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(bal_kept_points_kernel, dim3(obs_blocks + (d.num_kept * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d, obs_blocks);
}
void launch_bal_obs_precompute(const BalDev& d, hipStream_t s) { hipLaunchKernelGGL(bal_obs_precompute_kernel, dim3(grid_for(d.N)), dim3(kBlock), 0, s, d); }
void launch_bal_kept_sums(const BalDev& d, double* buf, int K, bool gather, hipStream_t s) {
  if (d.num_kept > 0) hipLaunchKernelGGL(bal_kept_sums_kernel, dim3((3 * d.num_kept + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d, buf, K, gather ? 1 : 0);
}
void launch_bal_cam_diag(const BalDev& d, hipStream_t s) {
  hipLaunchKernelGGL(bal_cam_diag_kernel, dim3((d.C * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d);
  if (d.num_dup > 0) hipLaunchKernelGGL(bal_dup_diag_kernel, dim3(d.num_dup), dim3(64), 0, s, d);  // (stream order: behind the diagonal blocks it adds to)
}
void launch_bal_pair(const BalDev& d, hipStream_t s) {
  // runs of eight logical blocks per XCD (developer variable SK_SCHEDULE_PLAIN=1: plain order.  Measured, Schur-assembly phase per iteration, plain -> 8:
  // Ladybug-1723 0.520 -> 0.499 ms, Venice-1778 3.02 -> 2.79 ms; 32 is worse on Ladybug — profiles/r03_pair_xcd_sweep.txt)
  const int group_long = dev_knobs().schedule_plain ? 0 : 8, group_short = group_long;
  auto grid = [](int blocks, int group) { return group > 0 ? (blocks + 8 * group - 1) / (8 * group) * (8 * group) : blocks; };
  if (d.num_long_segments > 0)  // first: the long ones take longest
    hipLaunchKernelGGL(bal_pair_long_kernel, dim3(grid((d.num_long_segments * 64 + kBlock - 1) / kBlock, group_long)), dim3(kBlock), 0, s, d, group_long);
  if (d.num_short_segments > 0) {
    const int waves = (d.num_short_segments + 6) / 7;
    hipLaunchKernelGGL(bal_pair_kernel, dim3(grid((waves * 64 + kBlock - 1) / kBlock, group_short)), dim3(kBlock), 0, s, d, group_short);
  }
}
void launch_finish_normal_matrix(double* S, int ld, int n, int npad, int rhs_row, const double* D, hipStream_t s) { hipLaunchKernelGGL(finish_normal_matrix_kernel, dim3((npad + 255) / 256), dim3(256), 0, s, S, ld, n, npad, rhs_row, D); }
// bal_finish_S for every front + up to four set_diagonal ranges, in one launch (a single device: six launches of ~6 us each
// in front of the factorisation when the system is dissected, four when it is not)
__global__ void bal_finish_all_kernel(BalDev d, BalFinishRanges r, int jmax) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= jmax) return;
  if (j < 9 * d.C) {
    const int i = j / 9, c = j - 9 * i, part = bal_part(d, i);
    if (d.front[part].S && !(d.pseudo && d.pseudo[i])) {
      int ld;
      double* blk = bal_block(d, i, i, &ld);
      const double D = bal_lm_diag(d, d.colsq_c[j]);
      blk[(size_t)c * ld + c] += D * D;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (r.S[k] && j >= r.from[k] && j < r.to[k]) r.S[k][(size_t)j * r.ld[k] + j] = r.value[k];
}
void launch_bal_finish_all(const BalDev& d, const BalFinishRanges& r, hipStream_t s) {
  int jmax = 9 * d.C;
  for (int k = 0; k < 4; ++k) if (r.S[k] && r.to[k] > jmax) jmax = r.to[k];
  hipLaunchKernelGGL(bal_finish_all_kernel, dim3((jmax + 255) / 256), dim3(256), 0, s, d, r, jmax);
}
void launch_bal_finish_S(const BalDev& d, int parts, hipStream_t s) { hipLaunchKernelGGL(bal_finish_S_kernel, dim3((9 * d.C + 255) / 256), dim3(256), 0, s, d, parts); }
void launch_set_diagonal(double* S, int ld, int from, int to, double value, hipStream_t s) {
  if (to > from) hipLaunchKernelGGL(set_diagonal_kernel, dim3((to - from + 255) / 256), dim3(256), 0, s, S, ld, from, to, value);
}
// Phase D up to the candidate point: the cameras' step from the fronts' solutions (out[0] = |delta_c|^2 of the coordinates in the two
// ranges), the points' back-substitution (out[1] = |delta_p|^2) — three launches and one final reduction
void launch_bal_backsub(const BalDev& d, double* out, int norm_lo, int norm_hi, int norm_lo2, int norm_hi2, hipStream_t s) {
  const int gc = (9 * d.C + kBlock - 1) / kBlock, gp = bal_point_blocks(d.P);
  hipLaunchKernelGGL(bal_cam_step_kernel, dim3(gc), dim3(kBlock), 0, s, d, norm_lo, norm_hi, norm_lo2, norm_hi2);
  hipLaunchKernelGGL(bal_obs_backsub_kernel, dim3(grid_for(d.N)), dim3(kBlock), 0, s, d);
  hipLaunchKernelGGL(bal_point_backsub_kernel, dim3(gp), dim3(kBlock), 0, s, d);
  if (d.num_kept > 0) hipLaunchKernelGGL(bal_kept_step_kernel, dim3(1), dim3(kBlock), 0, s, d, gp);
  ReduceRows rows;
  rows.n = 2;
  rows.row[0] = 0; rows.count[0] = gc; rows.out[0] = out;
  rows.row[1] = 1; rows.count[1] = d.P > 0 ? gp + (d.num_kept > 0 ? 1 : 0) : 0; rows.out[1] = out + 1;
  launch_final_reduce_rows(d.partial, d.partial_stride, rows, s);
}

}  // namespace sk
