// Dense fp64 Cholesky of the reduced camera system on gfx950 matrix cores.
// Replaces the Eigen LLT that native Ceres [ext] runs inside DENSE_SCHUR /
// DENSE_NORMAL_CHOLESKY (call sites: EX/SimpleBundleAdjuster.scala:148-152).
//
// Storage: S row-major, npad x ld, npad a multiple of 128, only the lower
// triangle referenced.  The padded tail is an identity and one padded row
// carries the right-hand side, so the forward substitution L z = rhs falls out
// of the factorisation itself (row `rhs_row` of L is z^T).
//
// Blocking: 128-wide block columns.  For block column kb
//   (1) lazy left-looking update from the block columns of the current group
//   (2) potrf128: factor the 128x128 diagonal block in LDS and invert it
//   (3) TRSM as a GEMM with the inverse
// and every `group` block columns one right-looking SYRK with K = group*128 on
// the trailing matrix.  (1), (3) and the SYRK are one kernel:
//   gemm_nt_f64_kernel : C (-)= A B^T, 128x128 tile, v_mfma_f64_16x16x4_f64.
#include <hip/hip_runtime.h>
#include <math.h>
#include "chol_kernels.hpp"

namespace sk {

typedef double d4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// C[128x128 tile] = (mode 0) C - A B^T   |  (mode 1) A B^T
// A: rows of the tile's row block, K contiguous (row-major, lda)
// B: rows of the tile's column block, K contiguous (row-major, ldb)
// 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 MFMA tiles.
// K-loop in steps of 16 through double-buffered LDS (row stride 18 doubles:
// conflict-free ds_read_b64 for the MFMA operand pattern), global loads of the
// next step in flight during the MFMAs of the current one.
// ---------------------------------------------------------------------------
static constexpr int kBK = 16;
static constexpr int kLd = 18;  // LDS row stride in doubles

__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(double* C, long ldc, const double* A, long lda,
                                                               const double* B, long ldb,
                                                               int K, int tiles_m, int tri, int mode) {
  __shared__ __attribute__((aligned(16))) double sh[2][2][128 * kLd];
  int ti, tj;
  if (tri) {
    const int b = blockIdx.x;
    int r = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= b) ++r;
    while (r * (r + 1) / 2 > b) --r;
    ti = r; tj = b - r * (r + 1) / 2;
  } else {
    ti = blockIdx.x % tiles_m; tj = blockIdx.x / tiles_m;
  }
  const double* Ag = A + (long)ti * 128 * lda;
  const double* Bg = B + (long)tj * 128 * ldb;
  double* Cg = C + (long)ti * 128 * ldc + (long)tj * 128;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  d4 acc[4][4];
  if (mode == 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[mt][nt][i] = -Cg[(long)(wr * 64 + mt * 16 + l4 + 4 * i) * ldc + wc * 64 + nt * 16 + l15];
  } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};
  }

  // staging: chunk q = i*256 + t -> row q>>3, 16-byte column chunk q&7
  double2 ra[4], rb[4];
  const int srow = t >> 3, sc = (t & 7) * 2;
#define SK_LOAD_STAGE(kbase)                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
    ra[i] = *reinterpret_cast<const double2*>(Ag + (long)(i * 32 + srow) * lda + (kbase) + sc); \
    rb[i] = *reinterpret_cast<const double2*>(Bg + (long)(i * 32 + srow) * ldb + (kbase) + sc); \
  }
#define SK_STORE_STAGE(buf)                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
    *reinterpret_cast<double2*>(&sh[buf][0][(i * 32 + srow) * kLd + sc]) = ra[i];        \
    *reinterpret_cast<double2*>(&sh[buf][1][(i * 32 + srow) * kLd + sc]) = rb[i];        \
  }
  const int nk = K / kBK;
  SK_LOAD_STAGE(0)
  SK_STORE_STAGE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // prefetch the next K-step (the last iteration re-reads its own step: no branch, no effect)
    const int knext = (kt + 1 < nk ? kt + 1 : kt) * kBK;
    SK_LOAD_STAGE(knext)
    const double* sa = &sh[buf][0][(wr * 64 + l15) * kLd + l4];
    const double* sb = &sh[buf][1][(wc * 64 + l15) * kLd + l4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { a[m] = sa[m * 16 * kLd + kk * 4]; b[m] = sb[m * 16 * kLd + kk * 4]; }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    SK_STORE_STAGE(buf ^ 1)
    __syncthreads();
  }
#undef SK_LOAD_STAGE
#undef SK_STORE_STAGE
  const double sgn = mode == 0 ? -1.0 : 1.0;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        Cg[(long)(wr * 64 + mt * 16 + l4 + 4 * i) * ldc + wc * 64 + nt * 16 + l15] = sgn * acc[mt][nt][i];
}

// ---------------------------------------------------------------------------
// potrf128: Cholesky of one 128x128 diagonal block and its inverse, one
// workgroup, everything resident in LDS as ten packed 32x32 blocks.
//   in : A (row-major, ld) lower triangle
//   out: A <- L (lower);  Linv (row-major 128x128, dense lower, upper stays 0)
//        info flag set when a pivot is not positive
// ---------------------------------------------------------------------------
static constexpr int kB = 32, kBs = 33;  // sub-block size and its LDS row stride
__device__ __forceinline__ int blk_off(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * (kB * kBs); }

// One wave: factor the 32x32 block D (LDS, row stride kBs) in place and
// replace it by its INVERSE (zeros above the diagonal).  The row of L each
// lane computed is returned in `a` for the write-back to global memory.
// Lane l works on row l & 31 (lanes 32..63 mirror 0..31: same values to the
// same addresses).  Cross-lane values travel through LDS broadcast reads
// (uniform address), which a single wave sees in program order.
__device__ __forceinline__ void wave_potrf32_inv(double* D, double* colbuf, double (&a)[32], bool* ok) {
  const int row = threadIdx.x & 31;
  double rinv[32];
  bool good = true;
#pragma unroll
  for (int c = 0; c < 32; ++c) a[c] = D[row * kBs + c];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    colbuf[row] = a[j];
    __builtin_amdgcn_wave_barrier();
    const double djj = colbuf[j];
    good = good && (djj > 0.0);
    const double ri = 1.0 / sqrt(djj);
    rinv[j] = ri;
    a[j] = a[j] * ri;  // column j of L (row j: sqrt(djj))
#pragma unroll
    for (int c = j + 1; c < 32; ++c) a[c] -= a[j] * (colbuf[c] * ri);
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int c = 0; c < 32; ++c) D[row * kBs + c] = a[c];
  __builtin_amdgcn_wave_barrier();
  // inverse: lane c solves L x = e_c ; L(i,t) is a broadcast read
  double x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    double s = (i == row) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < i; ++t) s -= D[i * kBs + t] * x[t];
    x[i] = (i >= row) ? s * rinv[i] : 0.0;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 32; ++i) D[i * kBs + row] = x[i];
  *ok = good;
}

// acc(2x2 MFMA tiles of a 32x32 block) += P Q^T, P/Q 32x32 blocks in LDS (stride kBs)
// transQ == false: Q used as rows (C = P Q^T);  true: Q used as is (C = P Q)
template <bool kPlainB>
__device__ __forceinline__ void block_mma32(d4 (&acc)[2][2], const double* P, const double* Q, int lane) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int k = kk * 4 + l4;
    double a[2], b[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      a[m] = P[(m * 16 + l15) * kBs + k];
      b[m] = kPlainB ? Q[k * kBs + m * 16 + l15] : Q[(m * 16 + l15) * kBs + k];
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
  }
}
__device__ __forceinline__ void block_store32(double* dst, const d4 (&acc)[2][2], int lane, double sgn) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(mt * 16 + l4 + 4 * i) * kBs + nt * 16 + l15] = sgn * acc[mt][nt][i];
}
__device__ __forceinline__ void block_load32(d4 (&acc)[2][2], const double* src, int lane, double sgn) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[mt][nt][i] = sgn * src[(mt * 16 + l4 + 4 * i) * kBs + nt * 16 + l15];
}

__global__ __launch_bounds__(256, 1) void potrf128_kernel(double* __restrict__ A, long ld, double* __restrict__ Linv, int* info) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* T = lds;                        // 10 packed blocks
  double* tmp = lds + 10 * kB * kBs;      // 4 per-wave 32x32 scratch blocks
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // load lower blocks
  for (int bi = 0; bi < 4; ++bi)
    for (int bj = 0; bj <= bi; ++bj) {
      double* dst = T + blk_off(bi, bj);
      for (int e = t; e < kB * kB; e += 256) {
        const int r = e >> 5, c = e & 31;
        dst[r * kBs + c] = A[(long)(bi * kB + r) * ld + bj * kB + c];
      }
    }
  __syncthreads();
  for (int jb = 0; jb < 4; ++jb) {
    // (a) diagonal sub-block: factor + invert in registers, wave 0
    if (wave == 0) {
      double a[32];
      double* D = T + blk_off(jb, jb);
      const int row = lane & 31;
      bool ok;
      wave_potrf32_inv(D, tmp, a, &ok);
      if (!ok && lane == 0) *info = 1;
      if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 32; ++c)
          if (c <= row) A[(long)(jb * kB + row) * ld + jb * kB + c] = a[c];
      }
    }
    __syncthreads();
    // (b) panel: X = T(bi, jb) * InvD^T  for bi > jb  (waves 1..3 -> bi = jb+wave, wave 0 helps when needed)
    {
      const double* InvD = T + blk_off(jb, jb);
      for (int bi = jb + 1 + wave; bi < 4; bi += 4) {
        d4 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};
        double* Xb = T + blk_off(bi, jb);
        block_mma32<false>(acc, Xb, InvD, lane);
        // all operand reads of this wave are done before it overwrites its own block
        __builtin_amdgcn_s_waitcnt(0xC07F);
        block_store32(Xb, acc, lane, 1.0);
        const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              A[(long)(bi * kB + mt * 16 + l4 + 4 * i) * ld + jb * kB + nt * 16 + l15] = acc[mt][nt][i];
      }
    }
    __syncthreads();
    // (c) trailing update inside the tile: T(bi,bj) -= X_bi X_bj^T, jb < bj <= bi
    {
      int idx = 0;
      for (int bi = jb + 1; bi < 4; ++bi)
        for (int bj = jb + 1; bj <= bi; ++bj, ++idx) {
          if ((idx & 3) != wave) continue;
          d4 acc[2][2];
          double* Cb = T + blk_off(bi, bj);
          block_load32(acc, Cb, lane, -1.0);
          block_mma32<false>(acc, T + blk_off(bi, jb), T + blk_off(bj, jb), lane);
          block_store32(Cb, acc, lane, -1.0);
        }
    }
    __syncthreads();
  }
  // LDS now: diagonal blocks = inverses of the diagonal blocks of L,
  // off-diagonal blocks = L.  Blocked in-place triangular inverse, block
  // columns right to left:  Inv(bi,bj) = -(sum_{t=bj+1..bi} Inv(bi,t) L(t,bj)) InvD_bj
  for (int bj = 2; bj >= 0; --bj) {
    const int bi = bj + 1 + wave;  // one wave per block of this block column
    d4 acc[2][2];
    const bool active = bi < 4;
    if (active) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};
      for (int tt = bj + 1; tt <= bi; ++tt) block_mma32<true>(acc, T + blk_off(bi, tt), T + blk_off(tt, bj), lane);
      block_store32(tmp + wave * kB * kBs, acc, lane, 1.0);
    }
    __syncthreads();  // every L(t,bj) of this column has been read
    if (active) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};
      block_mma32<true>(acc, tmp + wave * kB * kBs, T + blk_off(bj, bj), lane);
      block_store32(T + blk_off(bi, bj), acc, lane, -1.0);
    }
    __syncthreads();
  }
  // write the inverse (lower blocks; diagonal blocks carry explicit zeros above the diagonal)
  for (int bi = 0; bi < 4; ++bi)
    for (int bj = 0; bj <= bi; ++bj) {
      const double* src = T + blk_off(bi, bj);
      for (int e = t; e < kB * kB; e += 256) {
        const int r = e >> 5, c = e & 31;
        Linv[(long)(bi * kB + r) * 128 + bj * kB + c] = src[r * kBs + c];
      }
    }
}

// ---------------------------------------------------------------------------
// Backward substitution L^T y = z over 128-blocks, last block first.
//   bs_diag  : y_kb = Linv_kb^T w_kb                      (one workgroup)
//   bs_update: w[c] -= sum_r L[kb*128+r][c] y_kb[r], c < kb*128
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(128) void bs_diag_kernel(const double* __restrict__ Linv, double* __restrict__ y) {
  __shared__ double w[128];
  const int c = threadIdx.x;
  w[c] = y[c];
  __syncthreads();
  double s = 0.0;
  for (int r = c; r < 128; ++r) s += Linv[(long)r * 128 + c] * w[r];
  y[c] = s;
}
__global__ __launch_bounds__(256) void bs_update_kernel(const double* __restrict__ Lrow, long ld, const double* __restrict__ ykb,
                                                         double* __restrict__ w, int ncols) {
  __shared__ double ys[128];
  if (threadIdx.x < 128) ys[threadIdx.x] = ykb[threadIdx.x];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  double s = 0.0;
#pragma unroll 8
  for (int r = 0; r < 128; ++r) s += Lrow[(long)r * ld + c] * ys[r];
  w[c] -= s;
}

__global__ void copy_row_kernel(const double* __restrict__ src, double* __restrict__ dst, int n, int npad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < npad) dst[j] = j < n ? src[j] : 0.0;
}

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------
static void launch_gemm(double* C, long ldc, const double* A, long lda, const double* B, long ldb, int K,
                        int tiles_m, int tiles_n, bool tri, int mode, hipStream_t s, KernelTimer* kt, const char* name) {
  const int nblocks = tri ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
  if (nblocks <= 0) return;
  if (kt) kt->begin(name, s);
  hipLaunchKernelGGL(gemm_nt_f64_kernel, dim3(nblocks), dim3(256), 0, s, C, ldc, A, lda, B, ldb, K, tiles_m, tri ? 1 : 0, mode);
  if (kt) kt->end(name, s);
}

size_t potrf128_lds_bytes() { return (size_t)(10 + 4) * kB * kBs * sizeof(double); }

hipError_t cholesky_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(potrf128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)potrf128_lds_bytes());
}

// Factor S (npad x ld, lower) in place.  Linv: nblk x 128 x 128 (pre-zeroed once).
void cholesky_factor(double* S, long ld, int npad, double* Linv, int* info, int group, hipStream_t s, KernelTimer* kt) {
  const int nblk = npad / 128;
  int k0 = 0;
  for (int kb = 0; kb < nblk; ++kb) {
    double* Akk = S + (long)kb * 128 * ld + (long)kb * 128;
    const int rows_below = nblk - kb - 1;
    if (kb > k0)  // (1) lazy update of block column kb (diagonal block included) from columns [k0, kb)
      launch_gemm(Akk, ld, S + (long)kb * 128 * ld + (long)k0 * 128, ld, S + (long)kb * 128 * ld + (long)k0 * 128, ld,
                  (kb - k0) * 128, rows_below + 1, 1, false, 0, s, kt, "gemm_panel_update");
    if (kt) kt->begin("potrf128", s);
    hipLaunchKernelGGL(potrf128_kernel, dim3(1), dim3(256), potrf128_lds_bytes(), s, Akk, ld, Linv + (long)kb * 128 * 128, info);
    if (kt) kt->end("potrf128", s);
    if (rows_below > 0) {
      double* A21 = Akk + 128 * ld;
      launch_gemm(A21, ld, A21, ld, Linv + (long)kb * 128 * 128, 128, 128, rows_below, 1, false, 1, s, kt, "gemm_trsm");
      if (kb + 1 - k0 == group || kb + 1 == nblk) {
        // right-looking SYRK of the trailing matrix with the whole group, K = (kb+1-k0)*128
        double* A22 = S + (long)(kb + 1) * 128 * ld + (long)(kb + 1) * 128;
        const double* P = S + (long)(kb + 1) * 128 * ld + (long)k0 * 128;
        launch_gemm(A22, ld, P, ld, P, ld, (kb + 1 - k0) * 128, rows_below, rows_below, true, 0, s, kt, "gemm_syrk");
        k0 = kb + 1;
      }
    }
  }
}

// y (npad) <- solution of L^T y = z, with z^T = row rhs_row of L (first n entries).
void cholesky_backsolve(const double* S, long ld, int n, int npad, int rhs_row, const double* Linv, double* y, hipStream_t s,
                        KernelTimer* kt) {
  const int nblk = npad / 128;
  hipLaunchKernelGGL(copy_row_kernel, dim3((npad + 255) / 256), dim3(256), 0, s, S + (long)rhs_row * ld, y, n, npad);
  if (kt) kt->begin("backsolve", s);
  for (int kb = nblk - 1; kb >= 0; --kb) {
    hipLaunchKernelGGL(bs_diag_kernel, dim3(1), dim3(128), 0, s, Linv + (long)kb * 128 * 128, y + kb * 128);
    if (kb > 0) {
      const int ncols = kb * 128;
      hipLaunchKernelGGL(bs_update_kernel, dim3((ncols + 255) / 256), dim3(256), 0, s, S + (long)kb * 128 * ld, ld, y + kb * 128, y, ncols);
    }
  }
  if (kt) kt->end("backsolve", s);
}

double cholesky_syrk_flops(int npad, int group) {
  // algorithmic flops of the trailing SYRK launches: lower-triangular tiles incl. the diagonal tiles
  const int nblk = npad / 128;
  double f = 0.0;
  int k0 = 0;
  for (int kb = 0; kb < nblk; ++kb) {
    const int rows_below = nblk - kb - 1;
    if (rows_below > 0 && (kb + 1 - k0 == group || kb + 1 == nblk)) {
      const double tiles = 0.5 * rows_below * (rows_below + 1.0);
      f += tiles * 2.0 * 128.0 * 128.0 * (double)((kb + 1 - k0) * 128);
      k0 = kb + 1;
    }
  }
  return f;
}

}  // namespace sk
