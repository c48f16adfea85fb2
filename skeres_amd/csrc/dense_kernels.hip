// HIP kernels for the generic (dense Jacobian) path: any mix of registered
// device functors over arbitrary parameter blocks, DENSE_QR or
// DENSE_NORMAL_CHOLESKY  (EX/CurveFitting.scala:100-133, EX/Powell.scala:55-91).
// J is row-major m x n in HBM with the Jacobi column scaling folded in.
#include <hip/hip_runtime.h>
#include "dense_kernels.hpp"
#include "parameterization.hpp"
#include "functors.hpp"

namespace sk {

// One lane per residual block of functor F.  Mirrors
// AutoDiffCostFunction.evaluate (CORE/AutodiffCostFunction.scala:74-134):
// cost-only with T = double, else T = Jet<sum N(i)> seeded in block order.
template <class F, bool kJac>
__global__ void dense_eval_kernel(DenseEvalArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.count) return;
  const int b = a.blocks[i];
  const double* c = a.consts + a.const_off[b];
  const int* pidx = a.pidx + a.pidx_off[b];
  const int row0 = a.res_off[b];
  if (!kJac) {
    double store[F::kDim];
    const double* params[F::kBlocks];
    int k = 0;
#pragma unroll
    for (int q = 0; q < F::kBlocks; ++q) {
      params[q] = &store[k];
      const double* src = a.x + pidx[q];
      for (int j = 0; j < F::N(q); ++j) store[k++] = src[j];
    }
    double out[F::kRes];
    if (!F::template apply<double>(c, params, out)) { *a.fail_flag = 1; return; }
#pragma unroll
    for (int r = 0; r < F::kRes; ++r) a.r[row0 + r] = out[r];
  } else {
    typedef Jet<F::kDim> J;
    J store[F::kDim];
    const J* params[F::kBlocks];
    int k = 0;
#pragma unroll
    for (int q = 0; q < F::kBlocks; ++q) {
      params[q] = &store[k];
      const double* src = a.x + pidx[q];
      for (int j = 0; j < F::N(q); ++j) { store[k] = J(src[j], k); ++k; }
    }
    J out[F::kRes];
    if (!F::template apply<J>(c, params, out)) { *a.fail_flag = 1; return; }
#pragma unroll
    for (int r = 0; r < F::kRes; ++r) a.r[row0 + r] = out[r].a;
    k = 0;
#pragma unroll
    for (int q = 0; q < F::kBlocks; ++q) {
      const int off = pidx[q];
      for (int j = 0; j < F::N(q); ++j, ++k)
#pragma unroll
        for (int r = 0; r < F::kRes; ++r) a.J[(size_t)(row0 + r) * a.n + off + j] = out[r].v[k] * a.scale[off + j];
    }
  }
}

void launch_dense_eval(int functor_id, bool jac, const DenseEvalArgs& a, hipStream_t s) {
  if (a.count <= 0) return;
  const dim3 g((a.count + 127) / 128), b(128);
#define SK_LAUNCH(F)                                                             \
  do {                                                                           \
    if (jac) hipLaunchKernelGGL((dense_eval_kernel<F, true>), g, b, 0, s, a);    \
    else hipLaunchKernelGGL((dense_eval_kernel<F, false>), g, b, 0, s, a);       \
  } while (0)
  SK_DISPATCH_FUNCTOR(functor_id, SK_LAUNCH)
#undef SK_LAUNCH
}

// Robust losses (loss.hpp): one lane per residual block, after the evaluation (and after the upload of
// host-callback rows).  s = |r|^2; the block's rows of r and J are corrected in place (J has non-zeros
// only in the columns of the block's parameter blocks, and the correction maps that pattern to itself),
// and cterm receives the block's cost term: rho(s) in its first row, zero in the others, so that
// cost = 1/2 sum(cterm).  Blocks with the trivial loss get cterm = r^2 row by row.
__global__ void dense_loss_kernel(DenseLossArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.num_blocks) return;
  const int row0 = a.res_off[b], nres = a.res_off[b + 1] - row0;
  const int root = a.rb_loss[b];
  double s = 0.0;
  for (int r = 0; r < nres; ++r) { const double v = a.r[row0 + r]; s += v * v; if (root < 0) a.cterm[row0 + r] = v * v; }
  if (root < 0) return;
  double rho[3];
  loss_evaluate(a.nodes, root, s, rho);
  a.cterm[row0] = rho[0];
  for (int r = 1; r < nres; ++r) a.cterm[row0 + r] = 0.0;
  const LossCorrector c(s, rho);
  if (a.J) {
    for (size_t q = a.pidx_off[b]; q < a.pidx_off[b + 1]; ++q) {
      const int off = a.pidx[q], nq = a.psize[q];
      for (int j = 0; j < nq; ++j) {
        double* col = a.J + (size_t)row0 * a.n + off + j;
        double rtj = 0.0;
        if (c.alpha_sq_norm != 0.0)
          for (int r = 0; r < nres; ++r) rtj += a.r[row0 + r] * col[(size_t)r * a.n];
        for (int r = 0; r < nres; ++r) col[(size_t)r * a.n] = c.sqrt_rho1 * (col[(size_t)r * a.n] - c.alpha_sq_norm * a.r[row0 + r] * rtj);
      }
    }
  }
  for (int r = 0; r < nres; ++r) a.r[row0 + r] *= c.residual_scaling;
}
void launch_dense_loss(const DenseLossArgs& a, hipStream_t s) {
  if (a.num_blocks > 0) hipLaunchKernelGGL(dense_loss_kernel, dim3((a.num_blocks + 127) / 128), dim3(128), 0, s, a);
}
__global__ __launch_bounds__(256) void dense_sum_kernel(const double* v, int m, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) s += v[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) *out = sh[0];
}
void launch_dense_sum(const double* v, int m, double* out, hipStream_t s) { hipLaunchKernelGGL(dense_sum_kernel, dim3(1), dim3(256), 0, s, v, m, out); }

// Single residual block (sk_cost_function_evaluate): parameters / outputs are
// small flat device buffers.  jac_mask bit q set => write block q's Jacobian
// (row-major kRes x N(q)) at jac + jac_off[q].
template <class F>
__global__ void single_eval_kernel(const double* consts, const double* x, const int* x_off, double* residuals, double* jac,
                                   const int* jac_off, int want_jac, unsigned jac_mask, int* ok) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!want_jac) {
    const double* params[F::kBlocks];
    for (int q = 0; q < F::kBlocks; ++q) params[q] = x + x_off[q];
    double out[F::kRes];
    const bool good = F::template apply<double>(consts, params, out);
    *ok = good ? 1 : 0;
    if (good) for (int r = 0; r < F::kRes; ++r) residuals[r] = out[r];
    return;
  }
  typedef Jet<F::kDim> J;
  J store[F::kDim];
  const J* params[F::kBlocks];
  int k = 0;
  for (int q = 0; q < F::kBlocks; ++q) {
    params[q] = &store[k];
    for (int j = 0; j < F::N(q); ++j) { store[k] = J(x[x_off[q] + j], k); ++k; }
  }
  J out[F::kRes];
  const bool good = F::template apply<J>(consts, params, out);
  *ok = good ? 1 : 0;
  if (!good) return;
  for (int r = 0; r < F::kRes; ++r) residuals[r] = out[r].a;
  int off = 0;
  for (int q = 0; q < F::kBlocks; ++q) {
    const int nq = F::N(q);
    if ((jac_mask >> q) & 1u) {
      int col = 0;
      for (int r = 0; r < F::kRes; ++r)
        for (int p = 0; p < nq; ++p) jac[jac_off[q] + col++] = out[r].v[off + p];
    }
    off += nq;
  }
}

// ---- recorded functors (tape.hpp) --------------------------------------------------------------------------------
// One lane per residual block of the tape; W = 0: residuals only (T = double), else ceil(dim / W) passes of Jet<W>.
// Dynamic LDS: the register files of the workgroup's threads.
template <int W>
__global__ void dense_eval_tape_kernel(DenseEvalArgs a, TapeDev t) {
  extern __shared__ __attribute__((aligned(16))) double tape_lds[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.count) return;
  const int b = a.blocks[i];
  const double* c = a.consts + a.const_off[b];
  const int* pidx = a.pidx + a.pidx_off[b];
  const int row0 = a.res_off[b];
  auto param = [&](int k) { return a.x[pidx[t.param_block[k]] + t.param_index[k]]; };
  if (W == 0) {
    const TapeRegs<double> regs{tape_lds, (int)blockDim.x, (int)threadIdx.x};
    double out[kTapeMaxResiduals];
    tape_run<double>(t, c, param, 0, regs, out);
    for (int r = 0; r < t.num_residuals; ++r) a.r[row0 + r] = out[r];
  } else {
    typedef Jet<(W > 0 ? W : 1)> J;
    const TapeRegs<J> regs{tape_lds, (int)blockDim.x, (int)threadIdx.x};
    J out[kTapeMaxResiduals];
    for (int first = 0; first < t.dim; first += W) {
      tape_run<J>(t, c, param, first, regs, out);
      if (first == 0) for (int r = 0; r < t.num_residuals; ++r) a.r[row0 + r] = out[r].a;
      for (int w = 0; w < W && first + w < t.dim; ++w) {
        const int col = pidx[t.param_block[first + w]] + t.param_index[first + w];
        for (int r = 0; r < t.num_residuals; ++r) a.J[(size_t)(row0 + r) * a.n + col] = out[r].v[w] * a.scale[col];
      }
    }
  }
}
// one residual block (sk_cost_function_evaluate); thread 0 of one wave
template <int W>
__global__ void single_eval_tape_kernel(TapeDev t, const double* consts, const double* x, const int* x_off, double* residuals, double* jac,
                                        const int* jac_off, int want_jac, unsigned jac_mask, int* ok) {
  extern __shared__ __attribute__((aligned(16))) double tape_lds[];
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  auto param = [&](int k) { return x[x_off[t.param_block[k]] + t.param_index[k]]; };
  *ok = 1;
  if (!want_jac) {
    const TapeRegs<double> regs{tape_lds, (int)blockDim.x, 0};
    double out[kTapeMaxResiduals];
    tape_run<double>(t, consts, param, 0, regs, out);
    for (int r = 0; r < t.num_residuals; ++r) residuals[r] = out[r];
    return;
  }
  typedef Jet<W> J;
  const TapeRegs<J> regs{tape_lds, (int)blockDim.x, 0};
  J out[kTapeMaxResiduals];
  // size of parameter block q: the number of parameters numbered into it
  for (int first = 0; first < t.dim; first += W) {
    tape_run<J>(t, consts, param, first, regs, out);
    if (first == 0) for (int r = 0; r < t.num_residuals; ++r) residuals[r] = out[r].a;
    for (int w = 0; w < W && first + w < t.dim; ++w) {
      const int q = t.param_block[first + w], j = t.param_index[first + w];
      if (!((jac_mask >> q) & 1u)) continue;
      int nq = 0;
      for (int k = 0; k < t.dim; ++k) nq += t.param_block[k] == q ? 1 : 0;
      for (int r = 0; r < t.num_residuals; ++r) jac[jac_off[q] + r * nq + j] = out[r].v[w];  // row-major num_residuals x N(q)
    }
  }
}

template <class K>
static void tape_allow_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
bool launch_dense_eval_tape(const TapeDevBuffers& tb, bool jac, const DenseEvalArgs& a, hipStream_t s) {
  if (a.count <= 0) return true;
  const int threads = 128;
  const dim3 g((a.count + threads - 1) / threads), b(threads);
  if (!jac) {
    const size_t lds = tape_lds_bytes(tb.host, 0, threads);
    if (lds > kTapeLdsBudget) return false;
    tape_allow_lds(dense_eval_tape_kernel<0>, lds);
    hipLaunchKernelGGL(dense_eval_tape_kernel<0>, g, b, lds, s, a, tb.view);
    return true;
  }
  const int W = tape_pick_width(tb.host, threads);
  const size_t lds = tape_lds_bytes(tb.host, W, threads);
  switch (W) {
    case 3: tape_allow_lds(dense_eval_tape_kernel<3>, lds); hipLaunchKernelGGL(dense_eval_tape_kernel<3>, g, b, lds, s, a, tb.view); return true;
    case 2: tape_allow_lds(dense_eval_tape_kernel<2>, lds); hipLaunchKernelGGL(dense_eval_tape_kernel<2>, g, b, lds, s, a, tb.view); return true;
    case 1: tape_allow_lds(dense_eval_tape_kernel<1>, lds); hipLaunchKernelGGL(dense_eval_tape_kernel<1>, g, b, lds, s, a, tb.view); return true;
    default: return false;
  }
}
bool launch_single_eval_tape(const TapeDevBuffers& tb, const double* consts, const double* x, const int* x_off, double* residuals, double* jac,
                             const int* jac_off, int want_jac, unsigned jac_mask, int* ok, hipStream_t s) {
  const int threads = 64;
  const int W = tape_pick_width(tb.host, threads);
  if (W == 0) return false;
  const size_t lds = tape_lds_bytes(tb.host, W, threads);
  switch (W) {
    case 3: tape_allow_lds(single_eval_tape_kernel<3>, lds); hipLaunchKernelGGL(single_eval_tape_kernel<3>, dim3(1), dim3(threads), lds, s, tb.view, consts, x, x_off, residuals, jac, jac_off, want_jac, jac_mask, ok); break;
    case 2: tape_allow_lds(single_eval_tape_kernel<2>, lds); hipLaunchKernelGGL(single_eval_tape_kernel<2>, dim3(1), dim3(threads), lds, s, tb.view, consts, x, x_off, residuals, jac, jac_off, want_jac, jac_mask, ok); break;
    default: tape_allow_lds(single_eval_tape_kernel<1>, lds); hipLaunchKernelGGL(single_eval_tape_kernel<1>, dim3(1), dim3(threads), lds, s, tb.view, consts, x, x_off, residuals, jac, jac_off, want_jac, jac_mask, ok); break;
  }
  return true;
}

void launch_single_eval(int functor_id, const double* consts, const double* x, const int* x_off, double* residuals, double* jac,
                        const int* jac_off, int want_jac, unsigned jac_mask, int* ok, hipStream_t s) {
#define SK_LAUNCH(F) hipLaunchKernelGGL((single_eval_kernel<F>), dim3(1), dim3(64), 0, s, consts, x, x_off, residuals, jac, jac_off, want_jac, jac_mask, ok)
  SK_DISPATCH_FUNCTOR(functor_id, SK_LAUNCH)
#undef SK_LAUNCH
}

// colsq_j = sum_i J_ij^2 ; gs_j = sum_i J_ij r_i   (lane j: coalesced rows)
__global__ void dense_col_reduce_kernel(const double* J, const double* r, int m, int n, double* colsq, double* gs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double sq = 0.0, g = 0.0;
  for (int i = 0; i < m; ++i) { const double v = J[(size_t)i * n + j]; sq += v * v; g += v * r[i]; }
  colsq[j] = sq; gs[j] = g;
}
__global__ void dense_scale_kernel(double* J, const double* scale, int m, int n) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (size_t)m * n) J[e] *= scale[e % n];
}
// 0.5 * sum r^2 partials and friends -----------------------------------------------------------
__global__ __launch_bounds__(256) void dense_sumsq_kernel(const double* r, int m, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) s += r[i] * r[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}
// H (n_pad x ld, lower) = J^T J ; rhs row = J^T r.  Small-n form: one lane per (a, b <= a).
__global__ void dense_normal_kernel(const double* J, const double* r, int m, int n, double* H, int ld, int rhs_row) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * (n + 1)) return;
  const int a = e / (n + 1), b = e % (n + 1);
  if (b == n) {  // rhs
    double s = 0.0;
    for (int i = 0; i < m; ++i) s += J[(size_t)i * n + a] * r[i];
    H[(size_t)rhs_row * ld + a] = s;
  } else if (b <= a) {
    double s = 0.0;
    for (int i = 0; i < m; ++i) s += J[(size_t)i * n + a] * J[(size_t)i * n + b];
    H[(size_t)a * ld + b] = s;
  }
}
// step = -y ; x_new = x + step*scale ; out[0] = |x - x_new|^2
__global__ __launch_bounds__(256) void dense_step_kernel(const double* y, const double* scale, const double* x, double* step, double* x_new,
                                                         int n, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double st = -y[j];
    step[j] = st;
    const double xn = x[j] + st * scale[j];
    x_new[j] = xn;
    const double d = x[j] - xn;
    s += d * d;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}
// out[0] = sum_i m_i (r_i + m_i / 2),  m = J step
__global__ __launch_bounds__(256) void dense_model_kernel(const double* J, const double* r, const double* step, int m, int n, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) {
    double mr = 0.0;
    for (int j = 0; j < n; ++j) mr += J[(size_t)i * n + j] * step[j];
    s += mr * (r[i] + mr / 2.0);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}
// out[0] = max_j |gs_j / scale_j| ; out[1] = |x|^2
__global__ __launch_bounds__(256) void dense_gmax_kernel(const double* gs, const double* scale, const double* x, int n, double* out) {
  __shared__ double shm[256], shs[256];
  double mx = 0.0, s = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) { mx = fmax(mx, fabs(gs[j] / scale[j])); s += x[j] * x[j]; }
  shm[threadIdx.x] = mx; shs[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) { shm[threadIdx.x] = fmax(shm[threadIdx.x], shm[threadIdx.x + w]); shs[threadIdx.x] += shs[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = shm[0]; out[1] = shs[0]; }
}

// ---------------------------------------------------------------------------
// DENSE_QR: Householder QR of the augmented system [J ; diag(D)] y = [r ; 0]
// (what Ceres' DenseQRSolver hands to Eigen), one workgroup, A column-major
// rows x n in HBM, b = rhs (length rows).  y[0..n) on exit; *ok = 0 on a zero pivot.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_reduce_sum(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int w = blockDim.x / 2; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  const double out = sh[0];
  __syncthreads();
  return out;
}
__global__ __launch_bounds__(256) void dense_qr_build_kernel(const double* J, const double* r, const double* D, int m, int n, double* A, double* b) {
  const int rows = m + n;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)rows * n; e += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(e / rows), i = (int)(e % rows);
    A[e] = i < m ? J[(size_t)i * n + j] : (i - m == j ? D[j] : 0.0);
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) b[i] = i < m ? r[i] : 0.0;
}
__global__ __launch_bounds__(256) void dense_qr_solve_kernel(double* A, double* b, int rows, int n, double* y, int* ok) {
  __shared__ double sh[256];
  const int t = threadIdx.x;
  for (int k = 0; k < n; ++k) {
    double* a = A + (size_t)k * rows;
    double s = 0.0;
    for (int i = k + t; i < rows; i += 256) s += a[i] * a[i];
    const double norm = sqrt(block_reduce_sum(s, sh));
    if (norm == 0.0) { if (t == 0) *ok = 0; return; }
    const double akk = a[k];
    const double alpha = akk > 0.0 ? -norm : norm;
    const double v0 = akk - alpha;
    // |v|^2 = norm^2 - akk^2 + v0^2
    const double vnorm2 = (norm * norm - akk * akk) + v0 * v0;
    __syncthreads();
    if (vnorm2 != 0.0) {
      // apply H = I - 2 v v^T / |v|^2 to the remaining columns and to b; v = (v0, a[k+1..])
      for (int j = k + 1; j <= n; ++j) {
        double* c = j < n ? A + (size_t)j * rows : b;
        double dot = 0.0;
        for (int i = k + t; i < rows; i += 256) dot += (i == k ? v0 : a[i]) * c[i];
        const double f = 2.0 * block_reduce_sum(dot, sh) / vnorm2;
        for (int i = k + t; i < rows; i += 256) c[i] -= f * (i == k ? v0 : a[i]);
        __syncthreads();
      }
    }
    if (t == 0) a[k] = alpha;  // R_kk (the rest of the column is the reflector and is not needed again)
    __syncthreads();
  }
  if (t == 0) {
    for (int k = n - 1; k >= 0; --k) {
      double s = b[k];
      for (int j = k + 1; j < n; ++j) s -= A[(size_t)j * rows + k] * y[j];
      const double d = A[(size_t)k * rows + k];
      if (d == 0.0) { *ok = 0; return; }
      y[k] = s / d;
    }
  }
}

void launch_dense_col_reduce(const double* J, const double* r, int m, int n, double* colsq, double* gs, hipStream_t s) { hipLaunchKernelGGL(dense_col_reduce_kernel, dim3((n + 127) / 128), dim3(128), 0, s, J, r, m, n, colsq, gs); }
// ---------------------------------------------------------------------------
// Local parameterizations (parameterization.hpp): the minimiser works in the tangent space.
// ---------------------------------------------------------------------------
// Jl (m x nl) = Jg (m x ng) blockdiag(dPlus/ddelta at x) diag(scale): one lane per (row, parameter block)
__global__ __launch_bounds__(256) void dense_project_kernel(const double* Jg, int m, int ng, const ParamBlock* blocks, int nblocks, const double* x,
                                                           const double* scale, double* Jl, int nl) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)m * nblocks) return;
  const int row = (int)(e / nblocks);
  const ParamBlock pb = blocks[e % nblocks];
  if (pb.local_size == 0) return;
  const double* g = Jg + (size_t)row * ng + pb.global_off;
  bool any = false;
  for (int i = 0; i < pb.global_size; ++i) any = any || g[i] != 0.0;
  double* out = Jl + (size_t)row * nl + pb.local_off;
  if (!any) { for (int c = 0; c < pb.local_size; ++c) out[c] = 0.0; return; }
  double P[kParamMaxSize * kParamMaxSize];
  param_jacobian(pb, x + pb.global_off, P);
  for (int c = 0; c < pb.local_size; ++c) {
    double acc = 0.0;
    for (int i = 0; i < pb.global_size; ++i) acc += g[i] * P[i * pb.local_size + c];
    out[c] = acc * scale[pb.local_off + c];
  }
}
// step = -y (tangent, scaled); x_new = Plus(x, step * scale) block by block; out[0] = |x - x_new|^2
__global__ __launch_bounds__(256) void dense_plus_kernel(const double* y, const double* scale, const double* x, double* step, double* x_new,
                                                        const ParamBlock* blocks, int nblocks, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const ParamBlock pb = blocks[b];
    double delta[kParamMaxSize], xp[kParamMaxSize];
    for (int c = 0; c < pb.local_size; ++c) {
      const double st = -y[pb.local_off + c];
      step[pb.local_off + c] = st;
      delta[c] = st * scale[pb.local_off + c];
    }
    param_plus(pb, x + pb.global_off, delta, xp);
    for (int i = 0; i < pb.global_size; ++i) {
      x_new[pb.global_off + i] = xp[i];
      const double d = x[pb.global_off + i] - xp[i];
      s += d * d;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}
void launch_dense_project(const double* Jg, int m, int ng, const ParamBlock* blocks, int nblocks, const double* x, const double* scale, double* Jl, int nl,
                          hipStream_t s) {
  const long e = (long)m * nblocks;
  hipLaunchKernelGGL(dense_project_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, s, Jg, m, ng, blocks, nblocks, x, scale, Jl, nl);
}
void launch_dense_plus(const double* y, const double* scale, const double* x, double* step, double* x_new, const ParamBlock* blocks, int nblocks, double* out,
                       hipStream_t s) {
  hipLaunchKernelGGL(dense_plus_kernel, dim3(1), dim3(256), 0, s, y, scale, x, step, x_new, blocks, nblocks, out);
}

void launch_dense_scale(double* J, const double* scale, int m, int n, hipStream_t s) { const size_t e = (size_t)m * n; hipLaunchKernelGGL(dense_scale_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, s, J, scale, m, n); }
void launch_dense_sumsq(const double* r, int m, double* out, hipStream_t s) { hipLaunchKernelGGL(dense_sumsq_kernel, dim3(1), dim3(256), 0, s, r, m, out); }
void launch_dense_normal(const double* J, const double* r, int m, int n, double* H, int ld, int rhs_row, hipStream_t s) { const int e = n * (n + 1); hipLaunchKernelGGL(dense_normal_kernel, dim3((e + 255) / 256), dim3(256), 0, s, J, r, m, n, H, ld, rhs_row); }
void launch_dense_step(const double* y, const double* scale, const double* x, double* step, double* x_new, int n, double* out, hipStream_t s) { hipLaunchKernelGGL(dense_step_kernel, dim3(1), dim3(256), 0, s, y, scale, x, step, x_new, n, out); }
void launch_dense_model(const double* J, const double* r, const double* step, int m, int n, double* out, hipStream_t s) { hipLaunchKernelGGL(dense_model_kernel, dim3(1), dim3(256), 0, s, J, r, step, m, n, out); }
void launch_dense_gmax(const double* gs, const double* scale, const double* x, int n, double* out, hipStream_t s) { hipLaunchKernelGGL(dense_gmax_kernel, dim3(1), dim3(256), 0, s, gs, scale, x, n, out); }
void launch_dense_qr(const double* J, const double* r, const double* D, int m, int n, double* A, double* b, double* y, int* ok, hipStream_t s) {
  hipLaunchKernelGGL(dense_qr_build_kernel, dim3(64), dim3(256), 0, s, J, r, D, m, n, A, b);
  hipLaunchKernelGGL(dense_qr_solve_kernel, dim3(1), dim3(256), 0, s, A, b, m + n, n, y, ok);
}

}  // namespace sk
