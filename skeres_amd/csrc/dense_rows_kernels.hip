// HIP kernels of the dense-rows path (BASELINE.json config 5): m residuals r_i = tanh(a_i . x) - y_i
// over ONE parameter block of size n, a_i regenerated from a counter-based generator (synth.hpp).
// The Jacobian is stored TRANSPOSED, Jt[n_pad][m_pad] (residual index contiguous), so that
//   * every pass over it is coalesced along the residual index, and
//   * J^T J = Jt Jt^T is the NT MFMA SYRK of chol_kernels.hip with K = m_pad.
// Forward-mode autodiff of the row functor: the Jet of u = a . x has infinitesimal part a, and
// tanh propagates (tanh u, (1 - tanh^2 u) a)  — the spire / Ceres Jet rule for tanh.
#include <hip/hip_runtime.h>
#include "dense_rows_kernels.hpp"
#include "loss.hpp"
#include "synth.hpp"

namespace sk {

// u_i = a_i . x for 256 rows per workgroup; x staged through LDS in chunks.
// mode 0: r_i = tanh(u_i) - y_i and sd_i = 1 - tanh^2(u_i)   (Jacobian pass follows)
// mode 1: r only (cost-only evaluation at the candidate point)
__global__ __launch_bounds__(256) void rows_residual_kernel(DenseRowsArgs a, const double* __restrict__ x, double* __restrict__ r,
                                                            double* __restrict__ sd, int want_sd) {
  __shared__ double xs[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < a.m;
  const uint64_t seed = live ? (uint64_t)a.consts[3 * (size_t)i] : 0, row = live ? (uint64_t)a.consts[3 * (size_t)i + 1] : 0;
  double u = 0.0;
  for (int j0 = 0; j0 < a.n; j0 += 1024) {
    const int jn = a.n - j0 < 1024 ? a.n - j0 : 1024;
    __syncthreads();
    for (int j = threadIdx.x; j < jn; j += 256) xs[j] = x[j0 + j];
    __syncthreads();
    if (live)
      for (int j = 0; j < jn; ++j) u += synth_unit(seed, row, (uint64_t)a.n, (uint64_t)(j0 + j)) * xs[j];
  }
  if (!live) return;
  const double t = tanh(u * a.inv_sqrt_n);
  double res = t - a.consts[3 * (size_t)i + 2], d = (1.0 - t * t) * a.inv_sqrt_n;
  if (a.loss_root >= 0) {
    const double sq = res * res;
    double rho[3];
    loss_evaluate(a.loss_nodes, a.loss_root, sq, rho);
    a.cterm[i] = rho[0];
    if (want_sd) {  // (the candidate evaluation needs the cost term only)
      const LossCorrector c(sq, rho);
      d *= c.sqrt_rho1 * (1.0 - c.alpha_sq_norm * sq);
      res *= c.residual_scaling;
    }
  }
  r[i] = res;
  if (want_sd) sd[i] = d;
}

// Jt[j][i] = sd_i * unit(i, j) * scale_j  — one workgroup = 256 residuals x 16 parameters
__global__ __launch_bounds__(256) void rows_jacobian_kernel(DenseRowsArgs a, const double* __restrict__ sd, const double* __restrict__ scale,
                                                            double* __restrict__ Jt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.m) return;
  const uint64_t seed = (uint64_t)a.consts[3 * (size_t)i], row = (uint64_t)a.consts[3 * (size_t)i + 1];
  const double s = sd[i];
  const int j0 = blockIdx.y * 16;
#pragma unroll 4
  for (int jj = 0; jj < 16; ++jj) {
    const int j = j0 + jj;
    if (j < a.n) Jt[(size_t)j * a.m_pad + i] = s * synth_unit(seed, row, (uint64_t)a.n, (uint64_t)j) * scale[j];
  }
}

// per parameter j (one workgroup): colsq_j = sum_i Jt[j][i]^2, gs_j = sum_i Jt[j][i] r_i
__global__ __launch_bounds__(256) void rows_col_reduce_kernel(const double* __restrict__ Jt, const double* __restrict__ r, int m, size_t m_pad,
                                                              double* __restrict__ colsq, double* __restrict__ gs) {
  __shared__ double sa[256], sb[256];
  const double* row = Jt + (size_t)blockIdx.x * m_pad;
  double q = 0.0, g = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) { const double v = row[i]; q += v * v; g += v * r[i]; }
  sa[threadIdx.x] = q; sb[threadIdx.x] = g;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { sa[threadIdx.x] += sa[threadIdx.x + w]; sb[threadIdx.x] += sb[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { colsq[blockIdx.x] = sa[0]; gs[blockIdx.x] = sb[0]; }
}

__global__ __launch_bounds__(256) void rows_scale_kernel(double* __restrict__ Jt, const double* __restrict__ scale, int m, size_t m_pad) {
  double* row = Jt + (size_t)blockIdx.y * m_pad;
  const double s = scale[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) row[i] *= s;
}

// partial[b] = sum over this workgroup's residuals of m_i (r_i + m_i / 2), m_i = sum_j Jt[j][i] step_j
__global__ __launch_bounds__(256) void rows_model_kernel(const double* __restrict__ Jt, const double* __restrict__ r, const double* __restrict__ step,
                                                         int m, int n, size_t m_pad, double* __restrict__ partial) {
  __shared__ double sh[256];
  __shared__ double ss[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double mr = 0.0;
  for (int j0 = 0; j0 < n; j0 += 1024) {
    const int jn = n - j0 < 1024 ? n - j0 : 1024;
    __syncthreads();
    for (int j = threadIdx.x; j < jn; j += 256) ss[j] = step[j0 + j];
    __syncthreads();
    if (i < m)
      for (int j = 0; j < jn; ++j) mr += Jt[(size_t)(j0 + j) * m_pad + i] * ss[j];
  }
  sh[threadIdx.x] = i < m ? mr * (r[i] + mr / 2.0) : 0.0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// partial[b] = sum r_i^2 over the workgroup (cterm != nullptr: sum of the rows' cost terms rho(r_i^2) instead)
__global__ __launch_bounds__(256) void rows_sumsq_kernel(const double* __restrict__ r, const double* __restrict__ cterm, int m, double* __restrict__ partial) {
  __shared__ double sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  sh[threadIdx.x] = i < m ? (cterm ? cterm[i] : r[i] * r[i]) : 0.0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void rows_set_rhs_kernel(double* H, long ld, int rhs_row, const double* gs, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) H[(size_t)rhs_row * ld + j] = gs[j];
}

void launch_rows_residual(const DenseRowsArgs& a, const double* x, double* r, double* sd, bool want_sd, hipStream_t s) {
  hipLaunchKernelGGL(rows_residual_kernel, dim3((a.m + 255) / 256), dim3(256), 0, s, a, x, r, sd, want_sd ? 1 : 0);
}
void launch_rows_jacobian(const DenseRowsArgs& a, const double* sd, const double* scale, double* Jt, hipStream_t s) {
  hipLaunchKernelGGL(rows_jacobian_kernel, dim3((a.m + 255) / 256, (a.n + 15) / 16), dim3(256), 0, s, a, sd, scale, Jt);
}
void launch_rows_col_reduce(const double* Jt, const double* r, int m, int n, size_t m_pad, double* colsq, double* gs, hipStream_t s) {
  hipLaunchKernelGGL(rows_col_reduce_kernel, dim3(n), dim3(256), 0, s, Jt, r, m, m_pad, colsq, gs);
}
void launch_rows_scale(double* Jt, const double* scale, int m, int n, size_t m_pad, hipStream_t s) {
  hipLaunchKernelGGL(rows_scale_kernel, dim3((m + 255) / 256, n), dim3(256), 0, s, Jt, scale, m, m_pad);
}
int launch_rows_model(const double* Jt, const double* r, const double* step, int m, int n, size_t m_pad, double* partial, hipStream_t s) {
  const int g = (m + 255) / 256;
  hipLaunchKernelGGL(rows_model_kernel, dim3(g), dim3(256), 0, s, Jt, r, step, m, n, m_pad, partial);
  return g;
}
int launch_rows_sumsq(const double* r, const double* cterm, int m, double* partial, hipStream_t s) {
  const int g = (m + 255) / 256;
  hipLaunchKernelGGL(rows_sumsq_kernel, dim3(g), dim3(256), 0, s, r, cterm, m, partial);
  return g;
}
void launch_rows_set_rhs(double* H, long ld, int rhs_row, const double* gs, int n, hipStream_t s) {
  hipLaunchKernelGGL(rows_set_rhs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, H, ld, rhs_row, gs, n);
}

}  // namespace sk
