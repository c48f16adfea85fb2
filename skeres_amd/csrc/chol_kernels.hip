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
#include <algorithm>
#include <atomic>
#include <chrono>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include "chol_kernels.hpp"
#include "device_table.hpp"

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
// kShape: 0 rectangular grid (tiles_m x tiles_n; skip_upper drops tiles with ti < tj),
//         1 lower-triangular tile enumeration.
// kBKT  : K-step width (16 or 32 doubles); LDS row stride kBKT + 2 keeps ds_read_b64 conflict-free.
// kPF   : prefetch distance in K-steps (register sets in flight).  The throughput kernel (SYRK,
//         thousands of workgroups, 2 per CU) uses <16, 1>; the latency-bound panel kernels (a few
//         dozen workgroups, nothing else on the CU to hide an HBM round trip) use <32, 2>.
// kTM x kTN: output tile of one workgroup (2 x 2 waves, each (kTM/2) x (kTN/2)).  128 x 128 for the
//         throughput kernels; 64 x 128 / 64 x 64 for the panel kernels, whose grids are otherwise
//         too small to occupy 256 CUs (their cost is latency, not flops).
// skip  : kShape 0: bit 0 drops tiles whose 128-block row is above their 128-block column, bit 1
//         drops the tiles of 128-block (0, 0), bit 2 drops tiles entirely above the diagonal.
//         kShape 1: first tile of the slice.
// main_t, jump_t: the tile rows (kShape 1: and tile columns) of the launch are the first main_t consecutive
//         ones, then rows jump_t tiles further down (units of kTM rows): the active rows of an envelope
//         factorisation are a contiguous run plus the last block row, which carries the right-hand side.
constexpr int gemm_lds_doubles(int kBKT, int kTM, int kTN) { return 2 * (kTM + kTN) * (kBKT + 2); }
#define SK_GEMM_LDS(kBKT, kTM, kTN) __shared__ __attribute__((aligned(16))) double sh[gemm_lds_doubles(kBKT, kTM, kTN)];
template <int kMode, int kShape, int kBKT, int kPF, int kTM, int kTN>
__device__ __forceinline__ void gemm_nt_f64_body(double* shp, double* C, long ldc, const double* A, long lda, const double* B, long ldb, int K,
                                                 int tiles_m, int skip, int main_t = 0x7fffffff, int jump_t = 0, int main_n = 0x7fffffff,
                                                 int jump_n = 0, int block_id = -1, int* first_column_done = nullptr, int count_cols = 1,
                                                 int* diag_done = nullptr) {
  constexpr int mode = kMode;
  constexpr int kLdT = kBKT + 2;            // LDS row stride in doubles
  constexpr int kChA = kTM * kBKT / 512;    // 16-byte chunks per thread and stage, A operand
  constexpr int kChB = kTN * kBKT / 512;    // ... B operand
  constexpr int kRowStep = 512 / kBKT;      // rows covered by the 256 threads per chunk index
  constexpr int kMT = kTM / 32, kNT = kTN / 32;  // 16x16 MFMA tiles per wave
  static_assert(kChA >= 1 && kChB >= 1, "tile too small for 256 staging threads");
  constexpr int kShBuf = (kTM + kTN) * kLdT;  // shp: 2 * kShBuf doubles of LDS (gemm_lds_doubles), 16-byte aligned
  int ti, tj;
  bool first_column = false;  // kShape 2: a tile of the first block column(s) of the enumeration (see the epilogue)
  bool diag_tile = false;     // ... of the first DIAGONAL block: what the resident potrf server waits for after a paired SYRK
  if (kShape == 1) {
    const int b = blockIdx.x + skip;  // lower-triangular enumeration
    int r = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= b) ++r;
    while (r * (r + 1) / 2 > b) --r;
    ti = r; tj = b - r * (r + 1) / 2;
    if (ti >= main_t) ti += jump_t;
    if (tj >= main_t) tj += jump_t;
  } else if (kShape == 2) {
    // lower-triangular enumeration of 128-blocks, 128 / kTM row tiles each (kTN == 128), COLUMN by column (the tiles of a
    // block column follow each other, as in the rectangle this replaces: they share the 128 x K operand of their
    // column — enumerated row by row the launch fetched 127 MB instead of 82): the small SYRK without the workgroups
    // above the diagonal, which were half of the rectangle.  (An XCD-aware map on top — workgroup w, which goes to XCD
    // w mod 8, taking the tiles of one contiguous eighth of either triangular enumeration — was measured and dropped:
    // 102.1 MB per SYRK launch against 98.0, no difference in time; profiles/r02_d_pmc_traffic.json.)
    constexpr int kSub = 128 / kTM;
    const int T = tiles_m / kSub, nb = T * (T + 1) / 2;
    const int w = (int)blockIdx.x;
    const int k = nb - 1 - w / kSub;  // from the last (shortest) column backwards: the row-by-row enumeration of a triangle
    int r = (int)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= k) ++r;
    while (r * (r + 1) / 2 > k) --r;
    tj = T - 1 - r;
    first_column = tj < count_cols;
    diag_tile = tj == 0 && (k - r * (r + 1) / 2) == T - 1;
    ti = (T - 1 - (k - r * (r + 1) / 2)) * kSub + w % kSub;
    if (ti >= main_t) ti += jump_t;
    if (tj >= main_n) tj += jump_n;
  } else {
    const int bx = block_id >= 0 ? block_id : (int)blockIdx.x;  // (chain_column_kernel numbers its tiles itself)
    ti = bx % tiles_m; tj = bx / tiles_m;
    if (ti >= main_t) ti += jump_t;
    if (tj >= main_n) tj += jump_n;  // kShape 0 with mapped tile columns (units of kTN): a small SYRK run as a rectangle
    const int bi = ti * kTM / 128, bj = tj * kTN / 128;
    if ((skip & 1) && bi < bj) return;
    if ((skip & 2) && bi == 0 && bj == 0) return;
    if ((skip & 4) && (ti + 1) * kTM <= tj * kTN) return;
  }
  const double* Ag = A + (long)ti * kTM * lda;
  const double* Bg = B + (long)tj * kTN * ldb;
  double* Cg = C + (long)ti * kTM * ldc + (long)tj * kTN;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int crow = wr * (kTM / 2) + l4, ccol = wc * (kTN / 2) + l15;

  d4 acc[kMT][kNT];
  if (mode == 0) {
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[mt][nt][i] = -Cg[(long)(crow + mt * 16 + 4 * i) * ldc + ccol + nt * 16];
  } else {
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};
  }

  // staging: chunk q = i*256 + t -> row q / (kBKT/2), 16-byte column chunk q % (kBKT/2): every
  // wave-level load covers whole 128-B (kBKT = 16) or 256-B (kBKT = 32) row segments
  typedef double dstage_a __attribute__((ext_vector_type(2 * kChA)));  // one operand stage of a lane: SSA vector, never an alloca
  typedef double dstage_b __attribute__((ext_vector_type(2 * kChB)));
  dstage_a ra0, ra1;
  dstage_b rb0, rb1;
  const int srow = t / (kBKT / 2), sc = (t % (kBKT / 2)) * 2;
#define SK_LOAD_STAGE(set, kbase)                                                                         \
  _Pragma("unroll") for (int i = 0; i < kChA; ++i) {                                                       \
    const double2 v_ = *reinterpret_cast<const double2*>(Ag + (long)(i * kRowStep + srow) * lda + (kbase) + sc); ra##set[2 * i] = v_.x; ra##set[2 * i + 1] = v_.y; \
  }                                                                                                        \
  _Pragma("unroll") for (int i = 0; i < kChB; ++i) {                                                       \
    const double2 v_ = *reinterpret_cast<const double2*>(Bg + (long)(i * kRowStep + srow) * ldb + (kbase) + sc); rb##set[2 * i] = v_.x; rb##set[2 * i + 1] = v_.y; \
  }
#define SK_STORE_STAGE(set, buf)                                                                   \
  _Pragma("unroll") for (int i = 0; i < kChA; ++i)                                                  \
    *reinterpret_cast<double2*>(&shp[(buf) * kShBuf + (i * kRowStep + srow) * kLdT + sc]) = make_double2(ra##set[2 * i], ra##set[2 * i + 1]); \
  _Pragma("unroll") for (int i = 0; i < kChB; ++i)                                                  \
    *reinterpret_cast<double2*>(&shp[(buf) * kShBuf + (kTM + i * kRowStep + srow) * kLdT + sc]) = make_double2(rb##set[2 * i], rb##set[2 * i + 1]);
  const int nk = K / kBKT;  // a multiple of kPF (K is a multiple of 128)
  SK_LOAD_STAGE(0, 0)
  if (kPF == 2) { SK_LOAD_STAGE(1, (nk > 1 ? 1 : 0) * kBKT) }
  SK_STORE_STAGE(0, 0)
  __syncthreads();
  // one K-step: register set SETL held K-step kt, which is already in LDS: refill it with K-step
  // kt + kPF (clamped at the end: re-reads the last step, no branch, no effect); compute K-step kt;
  // then K-step kt + 1 (set SETS, loaded kPF iterations ago) goes to the other LDS buffer.
#define SK_LOAD_STAGE_X(set, kbase) SK_LOAD_STAGE(set, kbase)
#define SK_STORE_STAGE_X(set, buf) SK_STORE_STAGE(set, buf)
#define SK_KSTEP(kt, SETL, SETS)                                                                   \
  {                                                                                                \
    const int buf = (kt) & 1;                                                                      \
    const int knext = ((kt) + kPF < nk ? (kt) + kPF : nk - 1) * kBKT;                              \
    SK_LOAD_STAGE_X(SETL, knext)                                                                   \
    const double* sa = &shp[buf * kShBuf + (wr * (kTM / 2) + l15) * kLdT + l4];                               \
    const double* sb = &shp[buf * kShBuf + (kTM + wc * (kTN / 2) + l15) * kLdT + l4];                         \
    _Pragma("unroll") for (int kk = 0; kk < kBKT / 4; ++kk) {                                      \
      double a[kMT], b[kNT];                                                                       \
      _Pragma("unroll") for (int m = 0; m < kMT; ++m) a[m] = sa[m * 16 * kLdT + kk * 4];           \
      _Pragma("unroll") for (int m = 0; m < kNT; ++m) b[m] = sb[m * 16 * kLdT + kk * 4];           \
      _Pragma("unroll") for (int mt = 0; mt < kMT; ++mt)                                           \
        _Pragma("unroll") for (int nt = 0; nt < kNT; ++nt)                                         \
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt][nt], 0, 0, 0);  \
    }                                                                                              \
    SK_STORE_STAGE_X(SETS, buf ^ 1)                                                                \
    __syncthreads();                                                                               \
  }
  if (kPF == 1) {
    for (int kt = 0; kt < nk; ++kt) SK_KSTEP(kt, 0, 0)
  } else {
    for (int kt = 0; kt < nk; kt += 2) {
      SK_KSTEP(kt, 0, 1)
      SK_KSTEP(kt + 1, 1, 0)
    }
  }
#undef SK_KSTEP
#undef SK_LOAD_STAGE_X
#undef SK_STORE_STAGE_X
  (void)ra1; (void)rb1;
#undef SK_LOAD_STAGE
#undef SK_STORE_STAGE
  const double sgn = mode == 0 ? -1.0 : 1.0;
  if (kShape == 2 && first_column_done && first_column) {
    // The first block column of a trailing SYRK under the resident chain is all the NEXT column launch needs of it
    // (next(j+1) updates block column j+2, the first one syrk(j) touches): its tiles are written through to memory and
    // counted, and that launch waits for the count instead of the whole SYRK's completion (marker kernel and all).
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __hip_atomic_store(&Cg[(long)(crow + mt * 16 + 4 * i) * ldc + ccol + nt * 16], sgn * acc[mt][nt][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(first_column_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (diag_done && diag_tile) __hip_atomic_fetch_add(diag_done, 16 / (128 / kTM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // 16 when the block's tiles are in
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        Cg[(long)(crow + mt * 16 + 4 * i) * ldc + ccol + nt * 16] = sgn * acc[mt][nt][i];
}

// The three uses get their own kernel symbols so profiles separate them.
// Trailing SYRK of the blocked Cholesky (the dominant kernel): lower-triangular tiles, C -= A A^T.
// Launched in slices of the tile enumeration [tile_off, tile_off + gridDim.x).
__global__ __launch_bounds__(256, 2) void syrk_trailing_f64_kernel(double* C, long ldc, const double* A, long lda, int K, int tile_off, int main_t,
                                                                   int jump_t) {
  SK_GEMM_LDS(16, 128, 128)
  gemm_nt_f64_body<0, 1, 16, 1, 128, 128>(sh, C, ldc, A, lda, A, lda, K, 0, tile_off, main_t, jump_t);
}
// The same update for a SMALL trailing matrix (block envelope: a few dozen block rows): in 32 x 128 tiles, four to a
// 128-block of the lower triangle (kShape 2).  A 128 x 128 tile with K = 128 is one workgroup-latency of
// 30-50 us however few tiles there are (load 128 KB of C, eight K-steps, store 128 KB); 32-row tiles are four
// times as many workgroups, each a quarter as long, and the whole launch fits in two rounds.
__global__ __launch_bounds__(256, 2) void syrk_trailing_thin_f64_kernel(double* C, long ldc, const double* A, long lda, int K, int tiles_m, int main_t,
                                                                        int jump_t, int main_n, int jump_n, int* first_column_done, int count_cols, int* diag_done) {
  SK_GEMM_LDS(16, 32, 128)
  gemm_nt_f64_body<0, 2, 16, 2, 32, 128>(sh, C, ldc, A, lda, A, lda, K, tiles_m, 0, main_t, jump_t, main_n, jump_n, -1, first_column_done, count_cols, diag_done);
}
// ... and of two fronts in one launch (blockIdx.y picks the front)
struct ThinSyrkArgs { double* C; long ldc; const double* A; long lda; int K, tiles_m, main_t, jump_t, main_n, jump_n; int* first_column_done; int grid; };
struct ThinSyrkPair { ThinSyrkArgs f[2]; };
__global__ __launch_bounds__(256, 2) void syrk_trailing_thin_pair_f64_kernel(ThinSyrkPair pair) {
  const ThinSyrkArgs& p = pair.f[blockIdx.y];
  if ((int)blockIdx.x >= p.grid) return;
  SK_GEMM_LDS(16, 32, 128)
  gemm_nt_f64_body<0, 2, 16, 2, 32, 128>(sh, p.C, p.ldc, p.A, p.lda, p.A, p.lda, p.K, p.tiles_m, 0, p.main_t, p.jump_t, p.main_n, p.jump_n, -1, p.first_column_done, 1, nullptr);
}
// Gram matrix H = A A^T (lower-triangular tiles): J^T J of the dense path with the Jacobian stored
// transposed (A = J^T, K = number of residuals) — BASELINE.json config 5.
// The K range is split over blockIdx.y (chunk c covers columns [c*K, (c+1)*K) of A and writes slab c of C):
// a few thousand tiles x one huge K leave a ragged last wave of workgroups; tiles x chunks do not.
__global__ __launch_bounds__(256, 2) void syrk_gram_f64_kernel(double* C, long ldc, size_t slab_stride, const double* A, long lda, int K) {
  SK_GEMM_LDS(16, 128, 128)
  gemm_nt_f64_body<1, 1, 16, 1, 128, 128>(sh, C + (size_t)blockIdx.y * slab_stride, ldc, A + (size_t)blockIdx.y * K, lda, A + (size_t)blockIdx.y * K, lda, K, 0, 0);
}
// H = sum of the slabs, in slab order (deterministic); lower-triangular 128x128 tiles only
__global__ __launch_bounds__(256) void gram_reduce_kernel(double* __restrict__ H, const double* __restrict__ slabs, size_t slab_stride, long ld,
                                                          int nslabs) {
  const int b = blockIdx.x;
  int r = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= b) ++r;
  while (r * (r + 1) / 2 > b) --r;
  const int ti = r, tj = b - r * (r + 1) / 2;
  for (int e = threadIdx.x; e < 128 * 128; e += 256) {
    const size_t off = (size_t)(ti * 128 + (e >> 7)) * ld + tj * 128 + (e & 127);
    double s = 0.0;
    for (int c = 0; c < nslabs; ++c) s += slabs[(size_t)c * slab_stride + off];
    H[off] = s;
  }
}
// Panel updates (lazy left-looking update of a block column; look-ahead part of the SYRK): C -= A B^T,
// 64 x 128 tiles (tiles_m counts 64-row tiles), two workgroups per CU.
__global__ __launch_bounds__(256, 2) void gemm_update_f64_kernel(double* C, long ldc, const double* A, long lda, const double* B,
                                                                 long ldb, int K, int tiles_m, int skip, int main_t, int jump_t) {
  __builtin_amdgcn_s_setprio(2);  // on the critical path of the factorisation
  SK_GEMM_LDS(16, 64, 128)
  gemm_nt_f64_body<0, 0, 16, 2, 64, 128>(sh, C, ldc, A, lda, B, ldb, K, tiles_m, skip, main_t, jump_t);
}
// 32 x 128 tiles for short panels (a sparse envelope leaves a few dozen block rows per column): twice the
// workgroups again, each with half the rows to fetch; these kernels are bound by the latency of one workgroup.
__global__ __launch_bounds__(256, 2) void gemm_update_thin_f64_kernel(double* C, long ldc, const double* A, long lda, const double* B,
                                                                      long ldb, int K, int tiles_m, int skip, int main_t, int jump_t) {
  __builtin_amdgcn_s_setprio(2);
  SK_GEMM_LDS(16, 32, 128)
  gemm_nt_f64_body<0, 0, 16, 2, 32, 128>(sh, C, ldc, A, lda, B, ldb, K, tiles_m, skip, main_t, jump_t);
}
__global__ __launch_bounds__(256, 2) void trsm_gemm_thin_f64_kernel(double* C, long ldc, const double* A, long lda, const double* Linv,
                                                                    int tiles_m, int main_t, int jump_t) {
  __builtin_amdgcn_s_setprio(2);
  SK_GEMM_LDS(16, 32, 128)
  gemm_nt_f64_body<1, 0, 16, 2, 32, 128>(sh, C, ldc, A, lda, Linv, 128, 128, tiles_m, 0, main_t, jump_t);
}
// One 128 x 128 diagonal block, C -= A A^T, as 64 x 64 tiles (the upper one skipped): potrf128 waits on it.
__global__ __launch_bounds__(256, 1) void gemm_diag_f64_kernel(double* C, long ldc, const double* A, long lda, int K) {
  __builtin_amdgcn_s_setprio(3);
  SK_GEMM_LDS(32, 64, 64)
  gemm_nt_f64_body<0, 0, 32, 2, 64, 64>(sh, C, ldc, A, lda, A, lda, K, 2, 4);
}
// TRSM as a GEMM with the inverted diagonal block: C = A Linv^T (in place, C == A); 64 x 128 tiles.
__global__ __launch_bounds__(256, 2) void trsm_gemm_f64_kernel(double* C, long ldc, const double* A, long lda, const double* Linv,
                                                               int tiles_m, int main_t, int jump_t) {
  __builtin_amdgcn_s_setprio(2);
  SK_GEMM_LDS(16, 64, 128)
  gemm_nt_f64_body<1, 0, 16, 2, 64, 128>(sh, C, ldc, A, lda, Linv, 128, 128, tiles_m, 0, main_t, jump_t);
}

// ---------------------------------------------------------------------------
// potrf128: Cholesky of one 128x128 diagonal block and its inverse, one
// workgroup (4 waves), everything resident in LDS as packed 32x32 blocks.
//   in : A (row-major, ld) lower triangle
//   out: A <- L (lower);  Linv (row-major 128x128, dense lower, upper stays 0)
//        info flag set when a pivot is not positive
//
// The kernel is the serial bottleneck of the blocked factorisation, so it is
// organised around its critical path, which wave 0 walks alone:
//   P(jb)  the 32x32 diagonal factorisation (32 dependent column steps), L_jj -> global,
//          W_jj = L_jj^-1 -> LDS (free: lanes 32..63, see wave_potrf32)
//   B(jb)  X(jb+1,jb) = T(jb+1,jb) W_jj^T, T(jb+1,jb+1) -= X X^T
// Waves 1..3 do everything else between the same barriers: the other panel blocks and their
// diagonal updates, the off-diagonal updates, the blocked inverse (row by row, partial sums
// S(bi,bj) = sum_t L(bi,t) Inv(t,bj) accumulated as soon as their operands are final) and the
// write-backs, balanced so that no phase is longer than wave 0's.
// ---------------------------------------------------------------------------
static constexpr int kB = 32, kBs = 33;  // sub-block size and its LDS row stride
static constexpr int kBlk = kB * kBs;
__device__ __forceinline__ int blk_off(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * kBlk; }       // bj <= bi: L / W blocks
__device__ __forceinline__ int inv_off(int bi, int bj) { return (10 + bi * (bi - 1) / 2 + bj) * kBlk; }  // bj <  bi: inverse blocks

// 1/sqrt(x): hardware estimate + two Newton steps (~1 ulp)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
// 1/x: hardware estimate (~2^-23) + one cubic step y (1 + e + e^2), e = 1 - x y: three dependent fmas
__device__ __forceinline__ double fast_rcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y, 1.0);
  const double q = __builtin_fma(e, e, e);
  return __builtin_fma(y, q, y);
}
// value of x in lane `src` (compile-time constant) as a wave-uniform scalar
__device__ __forceinline__ double lane_bcast(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

// One wave: factor the 32x32 block D (LDS, row stride kBs).  Lanes 0..31 hold row `lane` of D and
// end with row `lane` of L in `a`.  Lanes 32..63 start from the unit vector e_k (k = lane - 32) and
// execute the same column operations, which for them is the forward substitution L w = e_k: they
// end with column k of W = L^-1, a[c] = W(c,k).
//
// Column j, with d = A'(j,j) and every lane's a[j] = A'(row,j) (updated by the columns before):
//   m = a[j] / d;  a[c] -= m A'(c,j) for c > j;  a[j] /= sqrt(d).
// The dependent chain from one column to the next is
//   d (v_readlane from lane j) -> y = rcp(d) -> e = 1 - d y -> q = e + e^2 -> m = a[j] y (1 + q) -> a[j+1] -= m A'(j+1,j)
// (A'(j+1,j) by v_readlane from lane j+1): five dependent VALU instructions and no LDS round trip.
// Everything else is filler that is issued BETWEEN the chain instructions, one column late:
//   - the other updates of column j-1 (c >= j+1), whose A'(c,j-1) every lane published to LDS as soon
//     as its a[j-1] was final (colbuf: 3 x 64 doubles, column j in buffer j % 3) and which were read back
//     into registers one column ago,
//   - the 1/sqrt(d) of column j-1 (two Newton steps) and the scaling of a[j-1].
// The in-order issue of one wave makes the placement matter: left to itself the scheduler puts the
// fillers first and the chain last, and a column costs their sum (measured 390 clocks per column).
// Each slot is pinned with sched_barrier.
__device__ __forceinline__ bool wave_potrf32(const double* D, double* colbuf, double (&a)[32], int lane) {
  const int row = lane & 31;
  const bool unit = lane >= 32;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const double d = D[row * kBs + c];
    a[c] = unit ? (c == row ? 1.0 : 0.0) : d;
  }
  colbuf[lane] = a[0];  // lanes 32..63 write slots nobody reads (no branch in the chain)
  // The broadcast reads of the column buffer below are at wave-uniform constant addresses, which the compiler moves from
  // an SGPR into a VGPR one by one (ds_read needs a VGPR address): 8 v_mov per column step on average, in a loop whose
  // length is its instruction count.  One opaque VGPR base and immediate offsets instead.
  typedef __attribute__((address_space(3))) const double* lds_cdp;
  unsigned cb_base = (unsigned)(unsigned long)(lds_cdp)colbuf;
  asm volatile("" : "+v"(cb_base));
  const lds_cdp cb3 = (lds_cdp)(unsigned long)cb_base;
  double cbp[32], cbn[32];  // column j-1 (in use) and column j (in flight) of A', c-indexed; SSA after unrolling
  double m_prev = 0.0, d_prev = 1.0;
  int dv_lo = 0, dv_hi = 0x3ff00000;  // 1.0 in every lane until its pivot arrives (lanes 32..63 keep it)
  int sgn = 0;  // sign bits of the pivots
  // keeps a filler value where it is written: pure arithmetic otherwise sinks to its use after the
  // loop, where the 32 refinements of 1/sqrt(d) ran back to back (measured 1.3 us per block)
#define SK_PIN(x) asm volatile("" : "+v"(x));
#ifndef SK_PROBE_VARIANT
#define SK_PROBE_VARIANT 0
#endif
#define SK_FILL(k)                                                                                  \
  if (j >= 1 && SK_PROBE_VARIANT != 1) {                                                                                     \
    _Pragma("unroll") for (int c = j + 1 + ((k) * (31 - j) + 4) / 5; c < j + 1 + (((k) + 1) * (31 - j) + 4) / 5; ++c) /* c = j+1 always in slot 0 */ \
      a[c] = __builtin_fma(-m_prev, cbp[c], a[c]);                                                  \
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    // prefetch: column j of A' for the late updates of the NEXT step
    {
      const lds_cdp cbj = cb3 + (j % 3) * 64;
#pragma unroll
      for (int c = j + 2; c < 32; ++c) cbn[c] = SK_PROBE_VARIANT == 2 ? a[c] * 0.5 : cbj[c];
    }
    // ---- chain 0
    const int d_lo = __builtin_amdgcn_readlane(__double2loint(a[j]), j), d_hi = __builtin_amdgcn_readlane(__double2hiint(a[j]), j);
    const double d = __hiloint2double(d_hi, d_lo);
    const double y = __builtin_amdgcn_rcp(d);
    const double l1 = lane_bcast(a[j], j + 1 < 32 ? j + 1 : j);
    const double m0 = a[j] * y;
    __builtin_amdgcn_sched_barrier(0);
    sgn |= d_hi;  // scalar: d is wave-uniform
    asm volatile("" : "+s"(sgn));
    // the pivot goes to lane j of dv: the 32 factors 1 / sqrt(d_j) are formed once, after the loop, in one vector pass
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dv_lo) : "s"(d_lo), "n"(j));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dv_hi) : "s"(d_hi), "n"(j));
    SK_FILL(0)
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain 1
    const double e = __builtin_fma(-d, y, 1.0);
    __builtin_amdgcn_sched_barrier(0);
    SK_FILL(1)
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain 2
    const double q = __builtin_fma(e, e, e);
    __builtin_amdgcn_sched_barrier(0);
    SK_FILL(2)
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain 3
    // (Round 5 tried the next column's value without waiting for the refined multiplier — (a[j+1] - m0 l1) - q (m0 l1), the dependent chain
    // rcp, e, q, a[j+1] instead of rcp, e, q, m, a[j+1]: 8936 clocks per 32 columns against 8732.  The loop is bound by the ISSUE of its
    // ~38 instructions per column — a lone wave issues an fp64 instruction every ~6 clocks, two waves on one SIMD every 3.9:
    // tools/valu_f64_probe.hip — not by the chain's latency: without the late updates a column takes 158 clocks, with them 272.)
    const double m = __builtin_fma(m0, q, m0);
    __builtin_amdgcn_sched_barrier(0);
    SK_FILL(3)
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain 4
    if (j + 1 < 32) {
      a[j + 1] = __builtin_fma(-m, l1, a[j + 1]);
      colbuf[((j + 1) % 3) * 64 + lane] = a[j + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    SK_FILL(4)
    __builtin_amdgcn_sched_barrier(0);
    m_prev = m;
    d_prev = d;
#pragma unroll
    for (int c = j + 2; c < 32; ++c) cbp[c] = cbn[c];
  }
#undef SK_FILL
#undef SK_PIN
  // column c of L (and of W) is what the elimination left, times 1 / sqrt(d_c): lane c of r holds that factor.  (Inside
  // the loop the same scaling cost eight fp64 instructions per column on wave-uniform values, in a loop whose length is
  // its fp64 instruction count: 0.6 us per 32 columns.)
  {
    const double r = fast_rsqrt(__hiloint2double(dv_hi, dv_lo));
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = a[c] * lane_bcast(r, c);
  }
  // a zero or NaN pivot poisons every later one, so the last pivot speaks for them
  return sgn >= 0 && d_prev > 0.0;
}

// acc(2x2 MFMA tiles of a 32x32 block) += P Q^T (kPlainB false) or P Q (true); P, Q 32x32 blocks in LDS (stride kBs)
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
// one 16x16 quadrant (qi, qj) of a 32x32 block product P Q^T: 8 MFMAs over two accumulators (fresh == true: acc starts at zero)
__device__ __forceinline__ void quad_mma16(d4& acc, const double* P, const double* Q, int qi, int qj, int lane, bool fresh) {
  const int l15 = lane & 15, l4 = lane >> 4;
  d4 a0 = fresh ? (d4){0.0, 0.0, 0.0, 0.0} : acc, a1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 8; kk += 2) {
    const int k = kk * 4 + l4;
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(qi * 16 + l15) * kBs + k], Q[(qj * 16 + l15) * kBs + k], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(qi * 16 + l15) * kBs + k + 4], Q[(qj * 16 + l15) * kBs + k + 4], a1, 0, 0, 0);
  }
  acc = a0 + a1;
}
__device__ __forceinline__ void quad_store16(double* dst, const d4& acc, int qi, int qj, int lane, double sgn) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[(qi * 16 + l4 + 4 * i) * kBs + qj * 16 + l15] = sgn * acc[i];
}
__device__ __forceinline__ void quad_load16(d4& acc, const double* src, int qi, int qj, int lane, double sgn) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = sgn * src[(qi * 16 + l4 + 4 * i) * kBs + qj * 16 + l15];
}
__device__ __forceinline__ void block_zero32(d4 (&acc)[2][2]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (d4){0.0, 0.0, 0.0, 0.0};
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
// C -= P Q^T on one wave (all three 32x32 LDS blocks)
__device__ __forceinline__ void block_update32(double* Cb, const double* P, const double* Q, int lane) {
  d4 acc[2][2];
  block_load32(acc, Cb, lane, -1.0);
  block_mma32<false>(acc, P, Q, lane);
  block_store32(Cb, acc, lane, -1.0);
}
// one wave: 32x32 LDS block -> global, 256-byte row segments.  dst is wave-uniform; goff = (lane >> 5) * ldd + (lane & 31)
// and loff = (lane >> 5) * kBs + (lane & 31) are the lane's offsets, step2 = 2 * ldd (32-bit: scalar base + vector offset addressing)
__device__ __forceinline__ void block_to_global(double* dst, unsigned step2, unsigned goff, const double* src, int loff) {
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[goff + i * step2] = src[loff + i * 2 * kBs];
}
// one wave: dst (LDS block, may be a scratch partial sum S) <- -W S, S read completely before the overwrite
__device__ __forceinline__ void block_neg_left_mul32(double* dst, const double* W, int lane) {
  d4 acc[2][2];
  block_zero32(acc);
  block_mma32<true>(acc, W, dst, lane);
  // the MFMA results depend on every read of dst, so the overwrite cannot pass them
  block_store32(dst, acc, lane, -1.0);
}

#ifdef SK_POTRF_STAMPS
__device__ long long g_potrf_stamps[4][16], g_potrf_clk[4][16];
#define SK_STAMP(i) if (lane == 0) { g_potrf_stamps[wave][i] = wall_clock64(); g_potrf_clk[wave][i] = clock64(); }
#else
#define SK_STAMP(i)
#endif

// `info` only ever rises: 0 ok, 1 a pivot was not positive, 2 a wait of the resident panel chain gave up.  After a
// time-out the hand-back kernels still run, on a half-updated matrix that is easily indefinite; their 1 must not replace
// the 2, or the host would count an invalid LM step instead of factoring the same system again (cholesky_note_info).
__device__ __forceinline__ void info_raise(int* info, int v) { (void)__hip_atomic_fetch_max(info, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void potrf128_body(double* lds_in, double* __restrict__ A, long ld, double* __restrict__ Linv, int* info) {
  // Inlined into the server's loop, everything below that does not depend on the column — LDS addresses, lane-derived
  // offsets — is loop-invariant, gets hoisted, and the 256-register body spills (384-716 bytes of scratch).  Opaque
  // copies of the three roots keep the arithmetic where it is used.
  double* lds = lds_in;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid), "+s"(ld));
  double* T = lds;                      // blocks 0..9: lower triangle of the tile (diagonal slots end as W_jj); 10..15: inverse, below the diagonal
  double* E = lds + 16 * kBlk;          // blocks 16, 17: staging of L_jj on its way to global memory (even / odd jb)
  double* colbuf = lds + 18 * kBlk;     // 3 x 64 doubles
  volatile int* flags = reinterpret_cast<volatile int*>(colbuf + 192);  // 4 hand-over flags of waves 1..3
  const int t = tid, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  __builtin_amdgcn_s_setprio(3);
  SK_STAMP(0)
  // Load the lower blocks.  Block (0,0) by all four waves, so that wave 0 starts the first diagonal factorisation after one
  // short round trip; the other nine blocks by waves 1..3 (three blocks each, 48 loads per lane, issued before anything
  // is waited for): they are first read after P(0), behind the barrier that ends it.
  {
    double v0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t + 256 * i;
      v0[i] = A[(long)(e >> 5) * ld + (e & 31)];
    }
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int e = t + 256 * i; T[blk_off(0, 0) + (e >> 5) * kBs + (e & 31)] = v0[i]; }
      if (t < 4) flags[t] = 0;
      __syncthreads();
    } else {
      double v[3][16];
      const int first = 3 * (wave - 1) + 1;  // index in the lower-triangular enumeration (bi (bi + 1) / 2 + bj)
      const long lo = (long)(lane >> 5) * ld + (lane & 31);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = first + k, bi = idx >= 6 ? 3 : (idx >= 3 ? 2 : 1), bj = idx - bi * (bi + 1) / 2;
        const double* Ab = A + ((long)bi * kB * ld + bj * kB) + lo;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[k][i] = Ab[(long)(2 * i) * ld];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int e = t + 256 * i; T[blk_off(0, 0) + (e >> 5) * kBs + (e & 31)] = v0[i]; }
      __syncthreads();
      const int ll = (lane >> 5) * kBs + (lane & 31);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = first + k, bi = idx >= 6 ? 3 : (idx >= 3 ? 2 : 1), bj = idx - bi * (bi + 1) / 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) T[blk_off(bi, bj) + ll + 2 * i * kBs] = v[k][i];
      }
    }
  }
  SK_STAMP(1)
  const unsigned uld = (unsigned)ld;
  const unsigned go_a = (unsigned)(lane >> 5) * uld + (lane & 31), go_i = (unsigned)(lane >> 5) * 128u + (lane & 31);
  const int lo_w = (lane >> 5) * kBs + (lane & 31);
#define SK_L(bi, bj) (T + blk_off(bi, bj))
#define SK_V(bi, bj) (T + inv_off(bi, bj))
#define SK_GA(bi, bj) (A + ((long)(bi) * kB * ld + (bj) * kB))
#define SK_GI(bi, bj) (Linv + ((bi) * kB * 128 + (bj) * kB))
  // one wave: X(bi,jb) = T(bi,jb) W_jj^T in place, then its own diagonal update (same wave: no barrier in between)
#define SK_PANEL_ROW(jb, bi)                                                                    \
  {                                                                                             \
    d4 acc[2][2];                                                                               \
    block_zero32(acc);                                                                          \
    block_mma32<false>(acc, SK_L(bi, jb), SK_L(jb, jb), lane);                                  \
    block_store32(SK_L(bi, jb), acc, lane, 1.0); /* every read of the block feeds the MFMAs */  \
  }
  // the step between two diagonal factorisations, on the critical path: X(jb+1,jb) = T(jb+1,jb) W_jj^T and
  // T(jb+1,jb+1) -= X X^T, each split into four 16x16 quadrants, one per wave (8 MFMAs instead of 32)
#define SK_CRITICAL_B(jb)                                                                        \
  {                                                                                              \
    d4 xq;                                                                                       \
    quad_mma16(xq, SK_L((jb) + 1, jb), SK_L(jb, jb), qi, qj, lane, true);                        \
    __syncthreads(); /* every wave has read T(jb+1,jb) */                                        \
    quad_store16(SK_L((jb) + 1, jb), xq, qi, qj, lane, 1.0);                                     \
    __syncthreads(); /* X(jb+1,jb) is complete */                                                \
    d4 uq;                                                                                       \
    quad_load16(uq, SK_L((jb) + 1, (jb) + 1), qi, qj, lane, -1.0);                               \
    quad_mma16(uq, SK_L((jb) + 1, jb), SK_L((jb) + 1, jb), qi, qj, lane, false);                 \
    quad_store16(SK_L((jb) + 1, (jb) + 1), uq, qi, qj, lane, -1.0);                              \
    __syncthreads();                                                                             \
  }
#define SK_WRITE_L(bi, bj) block_to_global(SK_GA(bi, bj), 2 * uld, go_a, SK_L(bi, bj), lo_w);
#define SK_WRITE_W(bi) block_to_global(SK_GI(bi, bi), 256u, go_i, SK_L(bi, bi), lo_w);
#define SK_WRITE_V(bi, bj) block_to_global(SK_GI(bi, bj), 256u, go_i, SK_V(bi, bj), lo_w);
#define SK_WRITE_E(bi) block_to_global(SK_GA(bi, bi), 2 * uld, go_a, E + ((bi) & 1) * kBlk, lo_w);
  // flags in LDS by which waves 1..3 pass blocks to each other while wave 0 is inside a diagonal factorisation
  // (a workgroup barrier would need wave 0): the producer's LDS writes are complete (lgkmcnt) before the flag
#define SK_FLAG_SET(i) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); if (lane == 0) flags[i] = 1; }
#define SK_FLAG_WAIT(i) { while (__builtin_amdgcn_readfirstlane(flags[i]) == 0) __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
  const int qi = wave >> 1, qj = wave & 1;
  // Wave 0 runs a loop (one copy of the 32-column chain in the instruction cache); waves 1..3 run
  // straight-line code with the same barriers.  `wave` is scalar: the branches are uniform.
  if (wave == 0) {
#pragma unroll 1
    for (int jb = 0; jb < 4; ++jb) {
      double* D = SK_L(jb, jb);
      double a[32];
      const bool ok = wave_potrf32(D, colbuf, a, lane);
      SK_STAMP(11 + jb)
      if (!ok && lane == 0) info_raise(info, 1);
      {
        // lanes 0..31: row of L_jj -> staging block E[jb & 1] (written to global by wave 3 during the next diagonal
        // factorisation); lanes 32..63: column k of W_jj -> the slot of D.  W has its zeros above the diagonal (they
        // are computed); the row of L is stored as it stands, with whatever the elimination left right of the diagonal:
        // nothing reads the upper triangle of a diagonal block of the factor (sk_cholesky_solve returns tril).
        const int row = lane & 31;
        double* dst = lane < 32 ? E + (jb & 1) * kBlk + row * kBs : D + row;
        const int step = lane < 32 ? 1 : kBs;
#pragma unroll
        for (int c = 0; c < 32; ++c) dst[c * step] = a[c];
      }
      SK_STAMP(2 + 2 * jb)
      __syncthreads();
      if (jb == 0) SK_CRITICAL_B(0)
      if (jb == 1) SK_CRITICAL_B(1)
      if (jb == 2) SK_CRITICAL_B(2)
      SK_STAMP(3 + 2 * jb)
    }
  } else {
    d4 accs[2][2];  // a partial sum S carried in registers
    SK_STAMP(2)
    __syncthreads();
    SK_CRITICAL_B(0)
    SK_STAMP(3)
    // ---- next to P(1): the rest of block column 0 and everything it updates
    if (wave == 1) {
      SK_PANEL_ROW(0, 2)
      SK_FLAG_SET(0)
      block_update32(SK_L(2, 2), SK_L(2, 0), SK_L(2, 0), lane);
      block_update32(SK_L(2, 1), SK_L(2, 0), SK_L(1, 0), lane);
      SK_WRITE_L(2, 0)
    }
    if (wave == 2) {
      SK_PANEL_ROW(0, 3)
      SK_FLAG_SET(1)
      block_update32(SK_L(3, 3), SK_L(3, 0), SK_L(3, 0), lane);
      block_update32(SK_L(3, 1), SK_L(3, 0), SK_L(1, 0), lane);
      SK_WRITE_L(3, 0)
    }
    if (wave == 3) {
      SK_WRITE_E(0) SK_WRITE_L(1, 0) SK_WRITE_W(0)
      SK_FLAG_WAIT(0) SK_FLAG_WAIT(1)
      block_update32(SK_L(3, 2), SK_L(3, 0), SK_L(2, 0), lane);
    }
    SK_STAMP(4)
    __syncthreads();
    SK_CRITICAL_B(1)
    SK_STAMP(5)
    // ---- next to P(2): the rest of block column 1; Inv(1,0) = -W_1 (L_10 W_0); S(2,0) = L_20 W_0 + L_21 Inv(1,0)
    if (wave == 1) {
      SK_PANEL_ROW(1, 3)
      block_update32(SK_L(3, 3), SK_L(3, 1), SK_L(3, 1), lane);
      block_update32(SK_L(3, 2), SK_L(3, 1), SK_L(2, 1), lane);
      SK_WRITE_L(3, 1) SK_WRITE_L(2, 1)
    }
    if (wave == 2) {
      d4 acc[2][2];
      block_zero32(acc);
      block_mma32<true>(acc, SK_L(1, 0), SK_L(0, 0), lane);
      block_store32(SK_V(1, 0), acc, lane, 1.0);
      block_neg_left_mul32(SK_V(1, 0), SK_L(1, 1), lane);
      SK_FLAG_SET(2)
      block_zero32(acc);  // S(2,1) = L_21 W_1
      block_mma32<true>(acc, SK_L(2, 1), SK_L(1, 1), lane);
      block_store32(SK_V(2, 1), acc, lane, 1.0);
      SK_WRITE_V(1, 0)
    }
    if (wave == 3) {
      SK_WRITE_E(1) SK_WRITE_W(1)
      block_zero32(accs);
      block_mma32<true>(accs, SK_L(2, 0), SK_L(0, 0), lane);
      SK_FLAG_WAIT(2)
      block_mma32<true>(accs, SK_L(2, 1), SK_V(1, 0), lane);
      block_store32(SK_V(2, 0), accs, lane, 1.0);
    }
    SK_STAMP(6)
    __syncthreads();
    SK_CRITICAL_B(2)
    SK_STAMP(7)
    // ---- next to P(3): row 2 of the inverse, and the partial sums S(3,bj) = sum_{t<3} L(3,t) Inv(t,bj) of row 3
    if (wave == 1) {  // S(3,0) = L_30 W_0 + L_31 Inv(1,0) + L_32 Inv(2,0)
      block_zero32(accs);
      block_mma32<true>(accs, SK_L(3, 0), SK_L(0, 0), lane);
      block_mma32<true>(accs, SK_L(3, 1), SK_V(1, 0), lane);
      SK_WRITE_L(3, 2)
      SK_FLAG_WAIT(3)
      block_mma32<true>(accs, SK_L(3, 2), SK_V(2, 0), lane);
      block_store32(SK_V(3, 0), accs, lane, 1.0);
      SK_WRITE_V(2, 0)
    }
    if (wave == 2) {  // Inv(2,1) = -W_2 S(2,1);  S(3,1) = L_31 W_1 + L_32 Inv(2,1)
      block_neg_left_mul32(SK_V(2, 1), SK_L(2, 2), lane);
      block_zero32(accs);
      block_mma32<true>(accs, SK_L(3, 1), SK_L(1, 1), lane);
      block_mma32<true>(accs, SK_L(3, 2), SK_V(2, 1), lane);
      block_store32(SK_V(3, 1), accs, lane, 1.0);
      SK_WRITE_V(2, 1)
    }
    if (wave == 3) {  // Inv(2,0) = -W_2 S(2,0);  S(3,2) = L_32 W_2
      block_neg_left_mul32(SK_V(2, 0), SK_L(2, 2), lane);
      SK_FLAG_SET(3)
      block_zero32(accs);
      block_mma32<true>(accs, SK_L(3, 2), SK_L(2, 2), lane);
      block_store32(SK_V(3, 2), accs, lane, 1.0);
      SK_WRITE_E(2) SK_WRITE_W(2)
    }
    SK_STAMP(8)
    __syncthreads();
  }
  // row 3 of the inverse: Inv(3,bj) = -W_3 S(3,bj)
  if (wave < 3) {
    block_neg_left_mul32(SK_V(3, wave), SK_L(3, 3), lane);
    SK_WRITE_V(3, wave)
  } else {
    SK_WRITE_E(3) SK_WRITE_W(3)
  }
  SK_STAMP(10)
#undef SK_PANEL_ROW
#undef SK_CRITICAL_B
#undef SK_FLAG_SET
#undef SK_FLAG_WAIT
#undef SK_WRITE_L
#undef SK_WRITE_W
#undef SK_WRITE_V
#undef SK_WRITE_E
#undef SK_L
#undef SK_V
#undef SK_GA
#undef SK_GI
}

__global__ __launch_bounds__(256, 1) void potrf128_kernel(double* __restrict__ A, long ld, double* __restrict__ Linv, int* info) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  potrf128_body(lds, A, ld, Linv, info);
}
// ---------------------------------------------------------------------------
// The panel chain of an envelope factorisation (groups of one block column) without a kernel launch on
// it.  Per block column j the serial dependence is
//   potrf(j) -> X(j+1,j) = S(j+1,j) W_jj^T -> S(j+1,j+1) -= X X^T -> potrf(j+1)
// and as four launches on one stream each arrow costs 8-10 us of dispatch and completion-signal latency
// (tools/queue_overheads.hip), twice the arithmetic.  Here
//   potrf_server_kernel   ONE workgroup, launched once per factorisation, factors the diagonal blocks one
//                         after the other as their last update lands;
//   chain_column_kernel   one launch per block column j, enqueued ahead of time: 32 x 128 tiles of the active
//                         rows below the diagonal.  Its workgroups wait for potrf(j), form X(.,j) (the TRSM as a
//                         GEMM with W_jj), wait for X(j+1,j) and for syrk(j-1) on the bulk stream, and apply
//                         panel j to block column j+1 ("next(j)" of cholesky_factor);
// hand over through counters in device memory (ChainSync): a producer's data is written back to memory
// (device-scope release) before its counter moves, a consumer invalidates its caches (acquire) after it
// saw the counter — the L2 caches of the 8 XCDs are not coherent with each other inside a kernel.
// Launches of one stream still follow each other in order, so column j+1's workgroups start (and wait)
// while potrf(j+1) runs, and their dispatch latency is off the chain.
// No workgroup waits for a workgroup with a higher block index of its own launch, the server workgroup is
// resident before the first column launch, and every wait gives up after kChainTimeoutTicks (then sets
// the abort flag, which ends every other wait, and *info): the grid always drains.
// ---------------------------------------------------------------------------
// kSyncServerXcc .. kSyncDSlot: the XCD-local hand-over of round 5 (below): the server's XCD (+ 1), the column it has factored last (a flag that
// lives in that XCD's L2), and sixteen slots each for the tiles of block row j + 1 — X(j+1,j) formed / S(j+1,j+1) updated — 64-byte aligned
// so that one scalar load reads all sixteen.
enum : int { kSyncPotrfDone = 0, kSyncAbort = 1, kSyncSyrkSeq = 2, kSyncSyrkColumn = 3, kSyncServerXcc = 4, kSyncPotrfFast = 5, kSyncStart = 6, kSyncEnd = 7,
             kSyncXSlot = 16, kSyncDSlot = 32, kSyncHeader = 64 };  // then diag_ready[maxblk], x_ready[maxblk], ticket[maxblk]
constexpr int kSyncArrays = 3;
// Workgroups at the head of a column launch that may become one of the 16 tiles of block row j + 1: those that find themselves on the
// potrf server's XCD take a ticket, the first sixteen get a tile, the others leave (workgroups go to the eight XCDs round-robin: 20 each)
constexpr int kCritCandidates = 160;
constexpr long long kChainTimeoutTicks = 100000000;  // 1 s of the 100 MHz wall clock (a block column takes 40 us: 25 000 times that)

// developer timeline (SK_CHAIN_STAMPS=<file>): wall-clock stamps of the server and of tile 0 of every column launch
__device__ long long g_chain_stamps[1024][8];
__device__ int g_chain_stamps_on;
// (two fronts in lock-step: the partner front's columns at 512 + j; stamp_front is a local of the two bodies)
#define SK_CHAIN_STAMP(col, i) if (g_chain_stamps_on && threadIdx.x == 0 && (col) < 512) g_chain_stamps[(col) + 512 * stamp_front][i] = wall_clock64();
__device__ __forceinline__ int sync_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: wait until *p >= target; false = aborted / timed out
__device__ __forceinline__ bool chain_wait(const int* p, int target, int* abort_flag) {
  if (sync_load(p) < target) {
    const long long t0 = wall_clock64();
    do {
      __builtin_amdgcn_s_sleep(4);
      if (sync_load(abort_flag) != 0) return false;
      if (wall_clock64() - t0 > kChainTimeoutTicks) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    } while (sync_load(p) < target);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidation has completed before the barrier that releases the other waves
  return true;
}
// one lane: wait until *p == value (the start signal of a factorisation: chain_start_kernel); false = aborted / timed out
__device__ __forceinline__ bool chain_wait_start(const int* p, int value) {
  if (sync_load(p) != value) {
    const long long t0 = wall_clock64();
    int spins = 0;
    do {
      __builtin_amdgcn_s_sleep(8);
      // (the counters — the abort flag among them — are this factorisation's only after the signal: nothing else is looked at before it.
      // The signal is a kernel on the caller's stream, enqueued before this launch: it cannot be lost, only late — behind whatever else that
      // stream holds, a caller's own kernels included — so the wait is a long one: thirty time-outs)
      if ((++spins & 63) == 0 && wall_clock64() - t0 > 30 * kChainTimeoutTicks) return false;
    } while (sync_load(p) != value);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return true;
}
// every wave of a consumer invalidates after the workgroup barrier that follows the wait
#ifdef SK_CHAIN_STRONG_FENCES
#define SK_CHAIN_ACQUIRE_ALL __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
#define SK_CHAIN_ACQUIRE_ALL
#endif
// whole workgroup: every wave's stores have reached the L2, then one lane writes the L2 back and moves the counter
__device__ __forceinline__ void chain_publish(int* p, int add) {
#ifdef SK_CHAIN_STRONG_FENCES
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's stores are in the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");           // one write-back of the L2 covers every wave's
    __hip_atomic_fetch_add(p, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
}

// For a payload of a few values per lane: the values themselves written through to memory (device-scope stores), so
// that the hand-over needs no write-back of the whole L2 — s_waitcnt vmcnt(0) per wave, then the counter.
__device__ __forceinline__ void store_through(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_publish_through(int* p, int add) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- XCD-local hand-over (round 5) ----------------------------------------------------------------------------------------------
// The L2 of an XCD is coherent for its own CUs.  When the sixteen tiles of block row j + 1 run on the potrf server's XCD, the chain's
// hand-overs — potrf(j) done, X(j+1,j) formed, S(j+1,j+1) updated — need neither a write-back of that L2 nor an invalidation, and the
// blocks handed over (W_jj, X, the diagonal block) are read from the L2 instead of memory: a flag is a plain store behind the data
// (s_waitcnt vmcnt(0) + barrier first: every wave's stores are in the L2), a poll is a SCALAR load behind s_dcache_inv (the scalar cache
// has just been emptied: the load comes from the L2; one request per wave — sixteen workgroups polling with device-scope vector loads
// slowed every kernel on the chip), and what is read has not been read before by any workgroup of the launch (a scratch block per column),
// so no stale line can sit in a CU's vector L1.  Everything that leaves the XCD — the thin tiles of the other rows, the trailing SYRK, the
// launches that follow — still goes through the device-scope counters, behind a release, off the critical path.
// (Round 1 measured the idea in a side experiment — a hop of 0.3-0.5 us instead of 1-2.5, a block from the L2 in 0.7 us instead of
// 2 — and did not finish it: the diagonal tile then waited for a marker kernel behind the whole SYRK; since round 2 the SYRK's first
// block column counts itself.)
typedef int int16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int load_l2(const int* p) {
  int v;
  asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
__device__ __forceinline__ int min16_l2(const int* p) {  // p: 64-byte aligned
  int16v v;
  asm volatile("s_dcache_inv\n\ts_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  int m = v[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = v[i] < m ? v[i] : m;
  return m;
}
__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ void store_l2(int* p, int v) { *(volatile int*)p = v; }  // (write-through L1: the store lands in the L2)
// one lane, the writer on the same XCD: wait until *p >= target; false = aborted / timed out.  No fence: see above.
__device__ __forceinline__ bool chain_wait_l2(const int* p, int target, int* abort_flag) {
  if (load_l2(p) < target) {
    const long long t0 = wall_clock64();
    int spins = 0;
    do {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 63) == 0) {
        if (sync_load(abort_flag) != 0) return false;
        if (wall_clock64() - t0 > kChainTimeoutTicks) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
      }
    } while (load_l2(p) < target);
  }
  return true;
}
// ... until all sixteen slots are >= target
__device__ __forceinline__ bool chain_wait_slots(const int* slots, int target, int* abort_flag) {
  if (min16_l2(slots) < target) {
    const long long t0 = wall_clock64();
    int spins = 0;
    do {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 63) == 0) {
        if (sync_load(abort_flag) != 0) return false;
        if (wall_clock64() - t0 > kChainTimeoutTicks) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
      }
    } while (min16_l2(slots) < target);
  }
  return true;
}
// The server's wait for diagonal block j: the sixteen slots of the tiles on its own XCD (the usual case) — or the device-scope counter
// (the first column of a resident run, set by the launch-by-launch part before it; a column whose block row lies outside the envelope;
// a factorisation without the XCD-local hand-over), looked at every sixteenth poll, with the acquire that path needs.
__device__ __forceinline__ bool server_wait_diag(int* sync, int j, int local) {
  const int* slots = sync + kSyncDSlot;
  const int* counter = sync + kSyncHeader + j;
  int* abort_flag = sync + kSyncAbort;
  if (!local) return chain_wait(counter, 16, abort_flag);
  const long long t0 = wall_clock64();
  for (int spins = 0;; ++spins) {
    if (min16_l2(slots) >= j) return true;
    if ((spins & 15) == 0) {
      if (sync_load(counter) >= 16) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return true;
      }
      if (sync_load(abort_flag) != 0) return false;
      if (wall_clock64() - t0 > kChainTimeoutTicks) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

struct ChainRanges { int n; int begin[8], end[8]; };  // the resident runs of block columns, [begin, end)

// The JOIN of a factorisation under the resident chain (round 5): instead of the caller's stream waiting for an event of each of the
// chain's streams — it gets there first and sits in a blocked wait that wakes up 25-35 us after the last signal — every stream ends with
// a one-thread marker (the server: its own exit) that counts into sync[kSyncEnd], and the caller's stream runs chain_gate_kernel, one
// wave that polls the counter: what follows on that stream starts a few microseconds after the last marker.  The gate does not give
// up on the chain's abort flag (kernels that abort still end, and their markers still run: what follows must not start next to them); a
// stream that never ends costs it four time-outs, then info = 2.
__device__ __forceinline__ void chain_end_signal(int* counter) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void chain_end_marker_kernel(int* counter) { chain_end_signal(counter); }
__global__ void chain_gate_kernel(const int* counter, int expected, int* info) {
  if (threadIdx.x != 0) return;
  if (sync_load(counter) < expected) {
    const long long t0 = wall_clock64();
    int spins = 0;
    do {
      __builtin_amdgcn_s_sleep(4);
      if ((++spins & 63) == 0 && wall_clock64() - t0 > 4 * kChainTimeoutTicks) { info_raise(info, 2); break; }
    } while (sync_load(counter) < expected);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__global__ void chain_start_gate_kernel(const int* p, int value, int* info) {
  if (threadIdx.x != 0) return;
  if (!chain_wait_start(p, value)) info_raise(info, 2);
}
// start != 0 (round 5): the launch was enqueued without an event in front of it — a blocked cross-queue wait wakes up 25-60 us after its
// signal (profiles/r05_iteration_trace.txt) — and waits here for chain_start_kernel, which the caller's stream runs once the system is
// assembled and the counters are reset: sync[kSyncStart] == start (a number that no earlier factorisation of this context used).
__device__ __forceinline__ void potrf_server_body(double* lds, double* S, long ld, const ChainRanges& ranges, double* Linv, int* info, int* sync, int maxblk, int local, int start) {
  __shared__ int ok_s;
  const int stamp_front = gridDim.x == 2 ? (int)blockIdx.x : 0;
  int done = 0;  // value of the potrf counter
  if (start != 0) {
    if (threadIdx.x == 0) ok_s = chain_wait_start(sync + kSyncStart, start) ? 1 : 0;
    __syncthreads();
    if (!ok_s) {
      if (threadIdx.x == 0) info_raise(info, 2);
      return;
    }
    SK_CHAIN_ACQUIRE_ALL
    __syncthreads();
  }
  if (local && threadIdx.x == 0) __hip_atomic_store(sync + kSyncServerXcc, 1 + xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // which XCD the tiles of block row j + 1 have to be on
  for (int r = 0; r < ranges.n; ++r)
    for (int j = ranges.begin[r]; j < ranges.end[r]; ++j) {
      if (j > 0) {  // the first column of a run: set by the launch-by-launch part before it; the others: by column launch j-1
        if (threadIdx.x == 0) ok_s = server_wait_diag(sync, j, local) ? 1 : 0;
        __syncthreads();
        if (!ok_s) {
          if (threadIdx.x == 0) info_raise(info, 2);
          return;
        }
        SK_CHAIN_ACQUIRE_ALL
      }
      SK_CHAIN_STAMP(j, 0)
      potrf128_body(lds, S + (long)j * 128 * ld + (long)j * 128, ld, Linv + (long)j * 128 * 128, info);
      if (local) {
        // L_jj and W_jj are in this XCD's L2: the tiles on this XCD may go ahead; then the write-back and the counter for everyone else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
          store_l2(sync + kSyncPotrfFast, j + 1);
          SK_CHAIN_STAMP(j, 1)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_fetch_add(sync + kSyncPotrfDone, j + 1 - done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        chain_publish(sync + kSyncPotrfDone, j + 1 - done);  // the counter reads j + 1 after column j
        SK_CHAIN_STAMP(j, 1)
      }
      done = j + 1;
    }
}
__global__ __launch_bounds__(256, 1) void potrf_server_kernel(double* S, long ld, ChainRanges ranges, double* Linv, int* info, int* sync, int maxblk, int local, int start) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  potrf_server_body(lds, S, ld, ranges, Linv, info, sync, maxblk, local, start);
  if (start != 0) chain_end_signal(sync + kSyncEnd);  // (the join of the factorisation: chain_gate_kernel)
}
// the start signal of a factorisation whose server waits for it in the kernel (one thread; a second front's counters too)
// (... and the reset of the counters themselves: n_a, n_b ints from sync_a, sync_b — one launch where there were two fills and a signal)
__global__ __launch_bounds__(256) void chain_start_kernel(int* sync_a, int n_a, int* sync_b, int n_b, int value) {
  for (int i = threadIdx.x; i < n_a; i += 256) sync_a[i] = 0;
  if (sync_b) for (int i = threadIdx.x; i < n_b; i += 256) sync_b[i] = 0;
  __syncthreads();
  if (threadIdx.x != 0) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __hip_atomic_store(sync_a + kSyncStart, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (sync_b) __hip_atomic_store(sync_b + kSyncStart, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Two fronts eliminated in ONE sequence of launches (cholesky_factor with a partner front): a
// server workgroup each, on a CU 0 each.
struct ServerArgs { double* S; long ld; ChainRanges ranges; double* Linv; int* sync; int maxblk; };
struct ServerPair { ServerArgs f[2]; };
__global__ __launch_bounds__(256, 1) void potrf_server_pair_kernel(ServerPair pair, int* info, int local, int start) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const ServerArgs& p = pair.f[blockIdx.x];  // (indexed in the kernel-argument segment: scalar loads)
  potrf_server_body(lds, p.S, p.ld, p.ranges, p.Linv, info, p.sync, p.maxblk, local, start);
  if (start != 0) chain_end_signal(pair.f[0].sync + kSyncEnd);  // (both servers count into the FIRST front's counter)
}

// One 32 x 32 tile with K = 128 in a single memory round trip: C = A B^T (kMode 1) or C -= A B^T (kMode 0), A and B
// 32 x 128 row slices.  Every load of the workgroup is in flight at once, then 32 MFMAs per wave (one 16 x 16
// quadrant each): about 5 us with the hand-over against 9 us for the K-stepped 32 x 128 tile, whose eight steps each wait
// for a round trip that nothing hides on the serial chain.  The time is the round trip and the publication, not the
// MFMAs (two accumulators instead of one changed nothing).  shp: 2 * 32 * kCritLd doubles.
constexpr int kCritLd = 130;
__device__ __forceinline__ void crit_tile_load(double* shp, const double* A, long lda, const double* B, long ldb) {
  const int t = threadIdx.x, row = t >> 3, c0 = (t & 7) * 16;
  double2 va[8], vb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) va[i] = *reinterpret_cast<const double2*>(A + (long)row * lda + c0 + 2 * i);
#pragma unroll
  for (int i = 0; i < 8; ++i) vb[i] = *reinterpret_cast<const double2*>(B + (long)row * ldb + c0 + 2 * i);
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<double2*>(shp + row * kCritLd + c0 + 2 * i) = va[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<double2*>(shp + (32 + row) * kCritLd + c0 + 2 * i) = vb[i];
}
__device__ __forceinline__ d4 crit_tile_mma(const double* shp, d4 acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const double* sa = shp + ((wave >> 1) * 16 + l15) * kCritLd + l4;
  const double* sb = shp + (32 + (wave & 1) * 16 + l15) * kCritLd + l4;
  // two accumulators (even / odd K-steps): a wave alone on its SIMD issues dependent fp64 MFMAs at half rate
  d4 acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 32; kk += 2) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[kk * 4], sb[kk * 4], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[kk * 4 + 4], sb[kk * 4 + 4], acc1, 0, 0, 0);
  }
  return acc + acc1;
}
// element i of a wave's accumulator: row (wave >> 1) * 16 + (lane >> 4) + 4 i, column (wave & 1) * 16 + (lane & 15) of the tile
__device__ __forceinline__ long crit_tile_off(long ldc, int i) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (long)((wave >> 1) * 16 + (lane >> 4) + 4 * i) * ldc + (wave & 1) * 16 + (lane & 15);
}

// Workgroups [0, ncrit): the 16 tiles (32 x 32) of block row j+1, which the server waits for; the rest: 32 x 128 tiles
// of the other active rows (thin tile 4 + (blockIdx.x - ncrit)).  ncrit == 0: block row j+1 is outside the envelope.
// Xs: 128 x 128 scratch.  X(j+1,j) cannot be formed in place tile by tile (every column tile reads whole rows of
// S(j+1,j)): the tiles go to Xs, which next(j) reads, and into S(j+1,j) once all 16 are known to have loaded theirs.
// ncand > 16: the XCD-local hand-over (above).  The launch's first ncand workgroups are CANDIDATES for the 16 tiles: a candidate on the
// potrf server's XCD takes a ticket — the first sixteen are the tiles, in ticket order — every other candidate leaves at once.
__device__ __forceinline__ void chain_column_body(int b, double* S, long ld, int j, const double* Linv_j, int tiles_m, int main_t, int jump_t,
                                                  int ncrit, double* Xs, int* sync, int maxblk, int syrk_need, int column_need, int* info, int do_next, int wait_first,
                                                  int ncand) {
  __shared__ __attribute__((aligned(16))) double sh[2 * 32 * kCritLd];
  static_assert(2 * 32 * kCritLd >= gemm_lds_doubles(16, 32, 128), "LDS of the thin tiles");
  __shared__ int ok_s;
  const int stamp_front = gridDim.y == 2 ? (int)blockIdx.y : 0;
  __builtin_amdgcn_s_setprio(2);
  double* A21 = S + (long)(j + 1) * 128 * ld + (long)j * 128;
  int* x_ready = sync + kSyncHeader + maxblk + j;
  int* diag_ready = sync + kSyncHeader + j + 1;
  int* abort_flag = sync + kSyncAbort;
  const bool local = ncand > ncrit;
  bool crit = b < ncand;
  if (crit && local) {
    // a tile of block row j + 1 on the server's XCD, or nothing
    if (threadIdx.x == 0) {
      int t = -1;
      if (chain_wait(sync + kSyncServerXcc, 1, abort_flag) && sync_load(sync + kSyncServerXcc) == 1 + xcc_id())
        t = __hip_atomic_fetch_add(sync + kSyncHeader + 2 * maxblk + j, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok_s = t >= 0 && t < 16 ? t : -1;
    }
    __syncthreads();
    const int tile = ok_s;
    __syncthreads();
    if (tile < 0) return;
    b = tile;
  }
  const int ri = b >> 2, q = b & 3;        // critical tile: rows ri * 32, columns q * 32 of the block
  const int thin_id = 4 + (b - ncand);     // other workgroups (ncrit == 16), or every workgroup from 0 (ncrit == 0)
  const int thin_bid = ncrit ? thin_id : b;
  const bool stamp = b == 0 && crit == (ncrit > 0);
  if (stamp) SK_CHAIN_STAMP(j, 2)
  if (crit && local) {
    // ---- the XCD-local path of a tile (ri, q) of block row j + 1
    double* Xj = Xs + (size_t)j * 128 * 128;  // this column's own scratch block: nothing of it has been read before by anybody
    const bool lower = q <= ri;               // the server reads the lower 32-blocks of the diagonal block only
    double* Ct = A21 + 128 + (long)(ri * 32) * ld + q * 32;
    {
      // rows ri of S(j+1,j) -> LDS (first half of sh): final since the launch before this one ended (stream order), which is also
      // when this CU's caches were last invalidated — no wait, no fence
      const int t = threadIdx.x, row = t >> 3, c0 = (t & 7) * 16;
      double2 va[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) va[i] = *reinterpret_cast<const double2*>(A21 + (long)(ri * 32 + row) * ld + c0 + 2 * i);
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<double2*>(sh + row * kCritLd + c0 + 2 * i) = va[i];
    }
    // ---- potrf(j), through the L2
    if (threadIdx.x == 0) ok_s = chain_wait_l2(sync + kSyncPotrfFast, j + 1, abort_flag) ? 1 : 0;
    __syncthreads();
    if (!ok_s) { if (threadIdx.x == 0) info_raise(info, 2); return; }
    if (stamp) SK_CHAIN_STAMP(j, 3)
    {
      // W_jj rows q -> LDS (second half of sh)
      const int t = threadIdx.x, row = t >> 3, c0 = (t & 7) * 16;
      double2 vb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) vb[i] = *reinterpret_cast<const double2*>(Linv_j + (long)(q * 32 + row) * 128 + c0 + 2 * i);
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<double2*>(sh + (32 + row) * kCritLd + c0 + 2 * i) = vb[i];
    }
    __syncthreads();
    d4 xacc = {0.0, 0.0, 0.0, 0.0};
    xacc = crit_tile_mma(sh, xacc);
#pragma unroll
    for (int i = 0; i < 4; ++i) Xj[(long)(ri * 32) * 128 + q * 32 + crit_tile_off(128, i)] = xacc[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's part of the tile is in the L2
    __syncthreads();
    if (threadIdx.x == 0) store_l2(sync + kSyncXSlot + b, j + 1);
    if (stamp) SK_CHAIN_STAMP(j, 4)
    // the other fifteen tiles of X (this XCD's L2), and — from the other XCDs — syrk(j-1)'s update of S(j+1,j+1): its first block column,
    // written through to memory and counted by its own workgroups.  Two lanes wait side by side.
    __shared__ int ok2_s;
    if (threadIdx.x == 0) ok_s = chain_wait_slots(sync + kSyncXSlot, j + 1, abort_flag) ? 1 : 0;
    if (threadIdx.x == 64) {
      // (no acquire fence: what depends on these counters is read past the caches, below)
      bool ok = true;
      const int* cnt[2] = {sync + kSyncSyrkSeq, sync + kSyncSyrkColumn};
      const int need[2] = {syrk_need, column_need};
      for (int w = 0; w < 2 && ok; ++w)
        if (sync_load(cnt[w]) < need[w]) {
          const long long t0 = wall_clock64();
          do {
            __builtin_amdgcn_s_sleep(2);
            if (sync_load(abort_flag) != 0 || wall_clock64() - t0 > kChainTimeoutTicks) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
          } while (sync_load(cnt[w]) < need[w]);
        }
      ok2_s = ok ? 1 : 0;
    }
    __syncthreads();
    if (!ok_s || !ok2_s) { if (threadIdx.x == 0) info_raise(info, 2); return; }
    if (stamp) SK_CHAIN_STAMP(j, 5)
    d4 cacc = {0.0, 0.0, 0.0, 0.0};
    double2 va[8], vb[8];
    {
      const int t = threadIdx.x, row = t >> 3, c0 = (t & 7) * 16;
      if (lower) {
        // the C tile past the caches (device-scope loads: syrk(j-1)'s tiles wrote it through from other XCDs; a stale line may sit in this
        // XCD's L2), and rows ri and q of X from the L2 — all in flight together
#pragma unroll
        for (int i = 0; i < 4; ++i) cacc[i] = -__hip_atomic_load(Ct + crit_tile_off(ld, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = *reinterpret_cast<const double2*>(Xj + (long)(ri * 32 + row) * 128 + c0 + 2 * i);
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = *reinterpret_cast<const double2*>(Xj + (long)(q * 32 + row) * 128 + c0 + 2 * i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) A21[(long)(ri * 32) * ld + q * 32 + crit_tile_off(ld, i)] = xacc[i];  // every tile has read S(j+1,j) by now
      if (b == 1 && threadIdx.x == 0) {
        // (tile (0, 1) has no update to do) X(j+1,j) to memory for the thin tiles of this launch, which may be anywhere on the chip: one
        // write-back of this XCD's L2 covers all sixteen tiles' stores.  (Writing the tiles through beside the plain stores, so that every
        // tile could count itself without a write-back, put the acknowledgement of those stores on the critical path: 8-9 us per tile step
        // instead of 5 — measured, not kept.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(x_ready, 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (lower) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<double2*>(sh + row * kCritLd + c0 + 2 * i) = va[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<double2*>(sh + (32 + row) * kCritLd + c0 + 2 * i) = vb[i];
      }
    }
    __syncthreads();
    if (lower) {
      // S(j+1,j+1) tile -= X[rows ri] X[rows q]^T
      cacc = crit_tile_mma(sh, cacc);
#pragma unroll
      for (int i = 0; i < 4; ++i) Ct[crit_tile_off(ld, i)] = -cacc[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) store_l2(sync + kSyncDSlot + b, j + 1);
    if (stamp) SK_CHAIN_STAMP(j, 6)
    return;
  }
  // A wait that gave up (its own time-out, or the abort flag another wait raised): this column was NOT computed.  The
  // host must hear of it from whichever kernel noticed — the server may have nothing left to wait for (the last
  // column of a resident run hands back to launch-by-launch kernels, which would factor stale data): *info = 2.
  // wait_first: block column j received its last update from the K = 256 SYRK of the resident pair before it (not from the
  // column launch before this one, which stream order would cover): the server went ahead on the diagonal block alone, the
  // TRSM needs every row — the SYRK's first block columns, counted by its workgroups
  if (threadIdx.x == 0)
    ok_s = (chain_wait(sync + kSyncPotrfDone, j + 1, abort_flag) &&
            (!wait_first || (chain_wait(sync + kSyncSyrkSeq, syrk_need, abort_flag) && chain_wait(sync + kSyncSyrkColumn, column_need, abort_flag)))) ? 1 : 0;
  __syncthreads();
  if (!ok_s) { if (threadIdx.x == 0) info_raise(info, 2); return; }
  SK_CHAIN_ACQUIRE_ALL
  if (stamp) SK_CHAIN_STAMP(j, 3)
  // ---- X(r,j) = S(r,j) W_jj^T
  d4 xacc = {0.0, 0.0, 0.0, 0.0};
  if (crit) {
    crit_tile_load(sh, A21 + (long)ri * 32 * ld, ld, Linv_j + (long)q * 32 * 128, 128);
    __syncthreads();
    xacc = crit_tile_mma(sh, xacc);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_through(Xs + (long)(ri * 32) * 128 + q * 32 + crit_tile_off(128, i), xacc[i]);
    chain_publish_through(x_ready, 1);
  } else {
    gemm_nt_f64_body<1, 0, 16, 2, 32, 128>(sh, A21, ld, A21, ld, Linv_j, 128, 128, tiles_m, 0, main_t, jump_t, 0x7fffffff, 0, thin_bid);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (stamp) SK_CHAIN_STAMP(j, 4)
  // do_next == 0: the SECOND column of a resident pair — its panel is applied by the pair's K = 256 SYRK, which also
  // hands block column j + 1 to the server; only the 16 tiles of block row j + 1 have something left to do (X into place)
  if (!do_next && !crit) return;
  if (threadIdx.x == 0)
    ok_s = (chain_wait(x_ready, ncrit, abort_flag) && (!do_next || (chain_wait(sync + kSyncSyrkSeq, syrk_need, abort_flag) &&
            chain_wait(sync + kSyncSyrkColumn, column_need, abort_flag)))) ? 1 : 0;
  __syncthreads();
  if (!ok_s) { if (threadIdx.x == 0) info_raise(info, 2); return; }
  SK_CHAIN_ACQUIRE_ALL
  if (stamp) SK_CHAIN_STAMP(j, 5)
  // ---- next(j): S(r,j+1) -= X(r,j) X(j+1,j)^T
  const double* Xsrc = local ? (const double*)(Xs + (size_t)j * 128 * 128) : (const double*)Xs;  // (the XCD-local tiles wrote this column's own scratch block)
  if (crit) {
#pragma unroll
    for (int i = 0; i < 4; ++i) A21[(long)(ri * 32) * ld + q * 32 + crit_tile_off(ld, i)] = xacc[i];  // every tile has read S(j+1,j) by now
    if (!do_next) return;
    if (q <= ri) {  // the server reads the lower 32-blocks only
      double* Ct = A21 + 128 + (long)(ri * 32) * ld + q * 32;
      d4 acc;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = -Ct[crit_tile_off(ld, i)];
      crit_tile_load(sh, Xs + (long)ri * 32 * 128, 128, Xs + (long)q * 32 * 128, 128);
      __syncthreads();
      acc = crit_tile_mma(sh, acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) store_through(Ct + crit_tile_off(ld, i), -acc[i]);
    }
    chain_publish_through(diag_ready, 1);  // (the copy of X into S(j+1,j) above is for after the factorisation: the end of the launch writes it back)
  } else {
    const double* Bx = ncrit ? Xsrc : (const double*)A21;
    gemm_nt_f64_body<0, 0, 16, 2, 32, 128>(sh, A21 + 128, ld, A21, ld, Bx, ncrit ? 128 : ld, 128, tiles_m, 1, main_t, jump_t, 0x7fffffff, 0, thin_bid);
    if (ncrit == 0 && b == 0) chain_publish(diag_ready, 16);
  }
  if (stamp) SK_CHAIN_STAMP(j, 6)
}
__global__ __launch_bounds__(256, 2) void chain_column_kernel(double* S, long ld, int j, const double* Linv_j, int tiles_m, int main_t, int jump_t,
                                                              int ncrit, double* Xs, int* sync, int maxblk, int syrk_need, int column_need, int* info, int do_next, int wait_first,
                                                              int ncand) {
  chain_column_body((int)blockIdx.x, S, ld, j, Linv_j, tiles_m, main_t, jump_t, ncrit, Xs, sync, maxblk, syrk_need, column_need, info, do_next, wait_first, ncand);
}
// ... the column launches of two fronts as one launch: blockIdx.y picks the front (grid.x: the larger of the two grids)
struct ColumnArgs { double* S; long ld; int j; const double* Linv_j; int tiles_m, main_t, jump_t, ncrit; double* Xs; int* sync; int maxblk, syrk_need, column_need, grid, ncand; };
struct ColumnPair { ColumnArgs f[2]; };
__global__ __launch_bounds__(256, 2) void chain_column_pair_kernel(ColumnPair pair, int* info) {
  const ColumnArgs& p = pair.f[blockIdx.y];
  if ((int)blockIdx.x >= p.grid) return;
  chain_column_body((int)blockIdx.x, p.S, p.ld, p.j, p.Linv_j, p.tiles_m, p.main_t, p.jump_t, p.ncrit, p.Xs, p.sync, p.maxblk, p.syrk_need, p.column_need, info, 1, 0, p.ncand);
}

// after a SYRK on its stream: the SYRK's completion (and its end-of-kernel write-back) as a counter the chain can poll
__global__ void chain_marker_kernel(int* flag, int seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// Backward substitution L^T y = z over 128-blocks, last block first; one
// launch per block step kb:
//   y_kb = Linv_kb^T w_kb                      (every workgroup, redundantly: L2-resident 128 KB)
//   w[c] -= sum_r L[kb*128+r][c] y_kb[r]       (c < kb*128; 64 columns per workgroup)
// 1024 threads so that each lane has only a few dependent HBM/L2 round trips.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bs_step_kernel(const double* __restrict__ Linv, const double* __restrict__ Lrow, long ld,
                                                        double* __restrict__ w, double* __restrict__ yout, int kb, int col0, int ncols) {
  __shared__ double wk[128], ykb[128], part[8][128];
  const int t = threadIdx.x;
  // every global load of the step is issued before anything is waited for (w_kb, this lane's 16 entries of Linv and its 8
  // entries of the block row of L): one memory round trip per step instead of three dependent ones
  const int c = t & 127, g = t >> 7;
  const int cl = t & 63, rg = t >> 6;
  const int col = col0 + blockIdx.x * 64 + cl;  // columns left of col0 are structurally zero in this block row
  double wv = 0.0, li[16], lr[8];
  if (t < 128) wv = w[kb * 128 + t];
#pragma unroll
  for (int i = 0; i < 16; ++i) li[i] = Linv[(long)(g * 16 + i) * 128 + c];
  if (col < ncols) {
    const double* p = Lrow + (long)(rg * 8) * ld + col;
#pragma unroll
    for (int i = 0; i < 8; ++i) lr[i] = p[(long)i * ld];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) lr[i] = 0.0;
  }
  if (t < 128) wk[t] = wv;
  __syncthreads();
  {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += li[i] * wk[g * 16 + i];
    part[g][c] = s;
  }
  __syncthreads();
  if (t < 128) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += part[q][t];
    ykb[t] = s;
    if (blockIdx.x == 0) yout[kb * 128 + t] = s;
  }
  __syncthreads();
  // 64 columns per workgroup (512-B row segments), 16 row groups of 8 rows: four times the workgroups of
  // a 256-column split — the grid is what limits this HBM-bound phase (<= 244 workgroups on 256 CUs)
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += lr[i] * ykb[rg * 8 + i];
  double* red = &part[0][0];  // part[][] is free again: its readers finished before the barrier above
  red[rg * 64 + cl] = s;
  __syncthreads();
  if (t < 64 && col < ncols) {
    double u = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) u += red[g * 64 + t];
    w[col] -= u;
  }
}

// ---------------------------------------------------------------------------
// The same backward substitution as ONE resident launch (round 3).  The 122 dependent launches of bs_step_kernel cost
// 5.7 us each on Ladybug-1723 — dispatch, a memory round trip for w, one for the block row — 0.65 ms of an 8.9 ms
// iteration for 0.3 ms of memory traffic.  Here block column kb has an OWNER workgroup (blockIdx nblk - 1 - kb: the
// last block column, which starts the chain, is dispatched first, and no workgroup waits for one dispatched after it,
// so the grid drains however few of its workgroups are resident at once).  The owner keeps w_kb in LDS and Linv_kb in
// registers, applies the block rows kb' > kb of its envelope in descending order as their y_kb' appear —
//   w_kb -= L[kb'][kb]^T y_kb'      (the 128 x 128 block prefetched into registers while it waits)
// — then forms y_kb = Linv_kb^T w_kb and publishes it.  There is no flag: y itself is the signal.  The y buffer is
// filled with a pattern no arithmetic produces (all ones) before the launch, a producer writes its 128 values through
// to memory, consumers poll the values they need with device-scope loads: ONE memory round trip per hop of the chain.
// The sums are grouped exactly as in bs_step_kernel (16 groups of 8 rows per update, 8 groups of 16 rows per y): the
// solution is the same, bit for bit (tests/test_gpu_parity.py::test_resident_backsolve_is_bitwise_the_launch_by_launch_one).
// A poll gives up after kChainTimeoutTicks: info = 2 (the caller factors and solves again, launch by launch).
// ---------------------------------------------------------------------------
#ifdef SK_TESTING
__device__ int g_bs_test_janitor_giveup;  // fault injection (SK_BS_TEST_JANITOR_GIVEUP=1, testing build only): janitor 0 of the next resident back-substitution gives up
#endif
__device__ long long g_bs_stamps[1024][4];  // developer timeline SK_BS_STAMPS=<file>: per block column, wall clock when its owner started waiting for
__device__ int g_bs_stamps_on;              // the block row next to the diagonal, when that y had arrived, and when its own y was stored
constexpr int kBsMaxBlocks = 960;  // (two tables of that many entries + the other arguments: inside the 4 KB kernel-argument segment)
struct BsTop { unsigned short top[kBsMaxBlocks]; unsigned short tail[kBsMaxBlocks]; };  // per block column: the last block row of its contiguous run (<= nblk - 1), and the first of its tail rows
constexpr unsigned long long kBsSentinel = ~0ull;
// (the arguments of one front's back-substitution; the kernels below hand them to the body with the block's index in that front's launch)
struct BsArgs {
  const double* Linv; double* S; long ld; const double* rhs; int n; double* y; int nblk; int* info; int nown; const double* yb;
  int janitors;       // number of janitor workgroups behind the owners
  const int* yb_map;  // border index -> index into yb (< 0: zero); nullptr: the border's own order
  int jan_parts;      // > 0 (the XCD-spread launch, below): a janitor takes ONE block column — janitor jn: column nown - 1 - jn / jan_parts — and every jan_parts-th block row of it
  int l2_poll;        // the owners sit on ONE XCD: a solution block is stored plainly first (into that XCD's L2) and polled with L1-bypassing loads
};
// 8-byte load that bypasses the CU's vector L1 (served by the XCD's L2)
__device__ __forceinline__ unsigned long long load_sc0_u64(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <typename Env>
__device__ __forceinline__ void bs_resident_body(const BsArgs& a, const Env& env, const int block) {
  const double* __restrict__ Linv = a.Linv; double* S = a.S; const long ld = a.ld; const double* rhs = a.rhs; const int n = a.n; double* y = a.y;
  const int nblk = a.nblk; int* info = a.info; const int nown = a.nown; const double* __restrict__ yb = a.yb; const int zero_after = a.janitors;
  const int* __restrict__ yb_map = a.yb_map;
  // nown < nblk: the interior of a LEAF FRONT (cholesky_backsolve_front): block columns [0, nown) have owners, the unknowns of
  // the block rows below them — the front's border — are known (yb, in the border's order: the separators' solution); the block
  // rows from env.tail[kb] on are active in block column kb whatever its run (the right-hand-side row; a spike: SegmentLayout; the
  // border of a bordered envelope, from the column that first reaches it), the others up to the column's envelope.  nown == nblk: a whole system.
  // 1024 threads: column c of the block and one of eight groups of sixteen rows each — sixteen products per thread and matrix
  // on the chain's critical path where 256 threads had sixty-four (0.41 -> 0.39 ms on Ladybug-1723: the hop is mostly hand-over and barriers); the partial sums are formed over the
  // same rows and added in the same order as before (and as in bs_step_kernel)
  // janitors > 0 (round 4): the launch has `janitors` more workgroups behind the owners.  Every block of L below the diagonal is read
  // here exactly once, by the owner of its column; when y_kb is out, the owner is done with column kb, and janitor kb mod
  // janitors writes ZEROS over that column's blocks: the assembly of the next linear system needs the envelope zero again, and
  // a separate pass over it (378 MB on Ladybug-1723) either sits on the critical path or — on a stream of its own next to the
  // Jacobian evaluation, as in rounds 2-3 — slows that evaluation to a third.  (The OWNERS storing the zeros behind their loads
  // was tried first: +0.25 ms of Cholesky phase — a workgroup's store path is one CU's, and an owner that has to write back every
  // block it reads no longer keeps up with the chain's 3.3 us per hop.)  The launch has idle CUs to spare (an owner per block
  // column on 256 CUs); a janitor waits for an owner, which was dispatched before it.  What the owners do not visit — the diagonal
  // blocks, a leaf front's border x border square — stays with zero_envelope_kernel (BalSolver: b_zero_min_f_).
  if (block >= nown) {
    const int jn = block - nown, t = threadIdx.x, c = t & 127, rg = t >> 7;
    __shared__ int ok_s;
    const int parts = a.jan_parts > 0 ? a.jan_parts : 1, part = a.jan_parts > 0 ? jn % a.jan_parts : 0;
    for (int kb = nown - 1 - (a.jan_parts > 0 ? jn / a.jan_parts : jn); kb >= 0; kb -= (a.jan_parts > 0 ? nown : zero_after)) {
      if (t == 0) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(y + (long)kb * 128);
        const long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kBsSentinel) {
          __builtin_amdgcn_s_sleep(8);
          if (wall_clock64() - t0 > kChainTimeoutTicks) { ok = 0; break; }
        }
#ifdef SK_TESTING
        if (jn == 0 && g_bs_test_janitor_giveup && atomicExch(&g_bs_test_janitor_giveup, 0) != 0) ok = 0;  // fault injection: janitor 0 gives up once
#endif
        // A janitor that gives up leaves blocks of the envelope unzeroed — and may be the ONLY workgroup that timed out (an owner's clock
        // restarts at every hop, a janitor's first wait spans up to 95 of them): it reports the time-out itself (info = 2: the host
        // assembles and factors the same system again, after a full zero pass — BalSolver::try_step_once, need_full_zero_)
        if (!ok && info) info_raise(info, 2);
        ok_s = ok;
      }
      __syncthreads();
      if (!ok_s) return;
      const int top = min((int)env.top[kb], nblk - 1);
      const int tlo = max((int)env.tail[kb], top + 1);
      double* col = S + (long)(rg * 16) * ld + (long)kb * 128 + c;
      int visited = 0;
      for (int cur = kb + 1; cur < nblk; ++cur) {
        if (cur > top && cur < tlo) { cur = tlo - 1; continue; }
        if ((visited++) % parts != part) continue;
        double* pz = col + (long)cur * 128 * ld;
#pragma unroll
        for (int i = 0; i < 16; ++i) pz[(long)i * ld] = 0.0;
      }
      __syncthreads();
    }
    return;
  }
  __shared__ double ysh[128], wsh[128], part[16][128];
  __shared__ int abort_s;
  const int kb = nown - 1 - block;
  const int t = threadIdx.x, c = t & 127, rg = t >> 7;
  if (t == 0) abort_s = 0;
  if (t < 128) { const int j = kb * 128 + t; wsh[t] = j < n ? rhs[j] : 0.0; }
  double li[16], lr[16];
  {
    const double* p = Linv + (long)kb * 128 * 128 + (long)(rg * 16) * 128 + c;
#pragma unroll
    for (int i = 0; i < 16; ++i) li[i] = p[(long)i * 128];
  }
  const int top = min((int)env.top[kb], nblk - 1);
  const int tlo = max((int)env.tail[kb], top + 1);                 // the tail rows that the envelope's run does not reach: [tlo, nblk)
  int cur = (kb < nblk - 1 && tlo <= nblk - 1) ? nblk - 1 : top;  // block rows to apply: nblk - 1 .. tlo, then top .. kb + 1
  const double* Lcol = S + (long)(rg * 16) * ld + (long)kb * 128 + c;
  if (cur > kb) {
    const double* p = Lcol + (long)cur * 128 * ld;
#pragma unroll
    for (int i = 0; i < 16; ++i) lr[i] = p[(long)i * ld];
  }
  __syncthreads();
  while (cur > kb) {
    if (g_bs_stamps_on && t == 0 && cur == kb + 1) g_bs_stamps[kb][0] = wall_clock64();
    if (t < 128 && cur >= nown) {
      const int bi = (cur - nown) * 128 + t;      // a border row: known
      const int src = yb_map ? yb_map[bi] : bi;
      ysh[t] = src >= 0 ? yb[src] : 0.0;
    } else if (t < 128) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(y + (long)cur * 128 + t);
      // (l2_poll: the producer is on this XCD — its plain store is in the L2 both share a microsecond and a half before the device-scope
      // one is visible; every poll looks there first, then at the device scope, which is what counts wherever the producer really is)
      unsigned long long v = a.l2_poll ? load_sc0_u64(src) : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v == kBsSentinel) {
        // (four polls in flight instead of one at a time: measured, no difference — a hop waits for one coherent read, 1.9 us
        // from the producer's completed store to the consumer's barrier, not for the poll that happens to see it)
        const long long t0 = wall_clock64();
        int spins = 0;
        do {
          __builtin_amdgcn_s_sleep(1);
          if (a.l2_poll) {
            v = load_sc0_u64(src);
            if (v == kBsSentinel && (++spins & 3) == 0) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (v == kBsSentinel && wall_clock64() - t0 > kChainTimeoutTicks) { abort_s = 1; v = 0ull; }
        } while (v == kBsSentinel);
      }
      ysh[t] = __longlong_as_double((long long)v);
    }
    __syncthreads();
    if (g_bs_stamps_on && t == 0 && cur == kb + 1) g_bs_stamps[kb][1] = wall_clock64();
#pragma unroll
    for (int g = 0; g < 2; ++g) {  // groups of eight rows: 2 rg and 2 rg + 1
      double sacc = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) sacc += lr[g * 8 + i] * ysh[rg * 16 + g * 8 + i];
      part[rg * 2 + g][c] = sacc;
    }
    const int nxt = (cur > top && cur - 1 < tlo) ? top : cur - 1;  // from the always-active rows down into the envelope's run
    if (nxt > kb) {  // the next block: its loads are in flight across the barrier and the next poll
      const double* p = Lcol + (long)nxt * 128 * ld;
#pragma unroll
      for (int i = 0; i < 16; ++i) lr[i] = p[(long)i * ld];
    }

    __syncthreads();
    if (t < 128) {
      double u = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) u += part[g][t];
      wsh[t] -= u;
    }
    cur = nxt;
  }
  __syncthreads();
  {
    double sacc = 0.0;  // the group of sixteen rows rg
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc += li[i] * wsh[rg * 16 + i];
    part[rg][c] = sacc;
  }
  __syncthreads();
  if (t < 128) {
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sacc += part[q][t];
    unsigned long long bits = (unsigned long long)__double_as_longlong(sacc);
    if (bits == kBsSentinel) bits ^= 1ull;  // (a NaN out of a failed factorisation must not look like "not there yet")
    if (a.l2_poll) *reinterpret_cast<volatile unsigned long long*>(y + (long)kb * 128 + t) = bits;  // (into this XCD's L2: the next owner's first look)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(y + (long)kb * 128 + t), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g_bs_stamps_on && t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_bs_stamps[kb][2] = wall_clock64(); }
  }
  if (t == 0 && abort_s && info) info_raise(info, 2);
}

__global__ __launch_bounds__(1024, 1) void bs_resident_kernel(const double* __restrict__ Linv, double* S, long ld, const double* rhs,
                                                               int n, double* y, int nblk, BsTop env, int* info, int nown, const double* __restrict__ yb, int zero_after /* number of janitor workgroups */,
                                                               const int* __restrict__ yb_map) {
  // spread != 0 (round 5): workgroup b of a launch goes to XCD b mod 8 — the owners are the workgroups 0, 8, 16, ... (all on ONE XCD: a hop's
  // hand-over goes through that XCD's L2), the seven between two owners are the janitors of the earlier one's block column, a seventh of its block rows each
  const int spread = zero_after < 0 ? 1 : 0;
  const BsArgs a{Linv, S, ld, rhs, n, y, nblk, info, nown, yb, spread ? 7 * nown : zero_after, yb_map, spread ? 7 : 0, spread};
  const int b = (int)blockIdx.x;
  bs_resident_body(a, env, spread ? ((b & 7) == 0 ? b >> 3 : nown + (b >> 3) * 7 + (b & 7) - 1) : b);
}
// Two leaf fronts' back-substitutions in ONE launch (round 5): side by side on two streams they cost a fork in front (~20 us) and, behind,
// a join that the main stream reaches first and sits in blocked (~65 us until the queue wakes up: profiles/r05_iteration_trace.txt).
// Owners of the two fronts alternate in dispatch order (each front's chain of hops starts at once), the janitors come behind all owners.
constexpr int kBsPairMaxBlocks = 448;  // (two pairs of tables of that many entries: inside the 4 KB kernel-argument segment)
struct BsTopHalf { unsigned short top[kBsPairMaxBlocks]; unsigned short tail[kBsPairMaxBlocks]; };
struct BsPair { BsArgs f[2]; BsTopHalf env[2]; };
__global__ __launch_bounds__(1024, 1) void bs_resident_pair_kernel(BsPair p) {
  const int b = (int)blockIdx.x, nA = p.f[0].nown, nB = p.f[1].nown, own = nA + nB;
  int front, blk;
  if (p.f[0].jan_parts > 0) {
    // the XCD-spread launch: slot 0 of every eight workgroups an owner of the first front (XCD 0), slot 4 one of the second (XCD 4), slots
    // 1-3 and 5-7 the janitors of those two block columns, a third of the block rows each
    const int g = b >> 3, slot = b & 7;
    front = slot >> 2;
    const int nown = front ? nB : nA;
    if (g >= nown) return;
    blk = (slot & 3) == 0 ? g : nown + g * 3 + (slot & 3) - 1;
    bs_resident_body(p.f[front], p.env[front], blk);
    return;
  }
  if (b < own) {
    const int m = nA < nB ? nA : nB;
    if (b < 2 * m) { front = b & 1; blk = b >> 1; } else { front = nA > nB ? 0 : 1; blk = b - m; }
  } else {
    const int j = b - own;
    if (j < p.f[0].janitors) { front = 0; blk = nA + j; } else { front = 1; blk = nB + (j - p.f[0].janitors); }
  }
  bs_resident_body(p.f[front], p.env[front], blk);  // (indexed in the kernel-argument segment)
}

__global__ void copy_row_kernel(const double* __restrict__ src, double* __restrict__ dst, int n, int npad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < npad) dst[j] = j < n ? src[j] : 0.0;
}

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------
// 18 packed 32x32 blocks + the column buffer + flags: 150 KB of dynamic LDS
static const size_t g_potrf_lds = (size_t)(18 * kBlk + 192 + 4) * sizeof(double);
size_t potrf128_lds_bytes() { return g_potrf_lds; }

static std::vector<hipStream_t> g_orphan_streams;  // streams of a device whose queue set could not be completed: kept, never destroyed
// Settled values of the tuning sweeps of rounds 1-3 (they were environment variables until round 4):
constexpr int g_early_column = 1;  // a resident column launch waits for the FIRST block column of the SYRK before it, not for a marker kernel behind all of it
constexpr int g_thin_grid = 512;   // panel launches of at most this many 32-row workgroups use the 32 x 128 kernels
constexpr int g_ext_events = 1;    // events ride on the producing kernel's own dispatch (hipExtLaunchKernelGGL) instead of separate records
constexpr unsigned g_event_flags = hipEventDisableTiming | hipEventDisableSystemFence;
static int g_chain_server = 1;  // developer variable SK_CHOL_CHAIN_SERVER=0 (dev_knobs): the launch-per-step panel chain even where the resident one applies (initial value of every device's DeviceQueues::chain_server)
constexpr int g_chain_max_trailing = 24, g_chain_prefix_group = 2;  // see cholesky_plan (profiles/r03_prefix_group_sweep.txt)
// Most trailing tile rows of a resident PAIR of block columns (cholesky_plan; developer knob SK_CHAIN_PAIR_MAX_TRAILING).  0 = no
// pairs, the DEFAULT: measured on Ladybug-1723 with 56 (block columns 8-44 as 18 pairs + 1: every column under the server) the
// factorisation takes 7.14 ms against 7.13 with those columns launch by launch, and the bench line 8.86 ms against 8.65 —
// that region is bound by the in-situ rate of its K = 256 SYRKs either way (34 TFLOP/s in the 32-row tiling the pairs need for
// their first-column signal, 38 in the 128-row tiling the launch-by-launch plan uses above 48 tile rows), not by its panel
// chain (profiles/r03_pair_chain_timeline.txt).  Kept behind the knob, covered by tests/pair_plan_worker.py.
static int g_pair_max_trailing = 0;
constexpr int g_thin_syrk_tiles = 48;  // trailing matrices of at most this many block rows use the 32 x 128-tile SYRK
constexpr int g_tail_tiles = 48, g_tail_group = 1;  // see cholesky_group_bounds (measured: 40-54 within 0.3 %)
static int g_chain_local = 0;  // developer variable SK_CHAIN_XCD_LOCAL=1: the XCD-local hand-over below (measured in round 5: 39.5 -> 37.3 us per column of ONE front, but 44.5 -> ~48 us per
                               // step of two fronts in lock-step, which is what the default plan runs: off by default; profiles/r05_xcd_local_chain.txt)
static std::atomic<int> g_bs_resident{1};  // developer knob SK_BS_RESIDENT=0: the back-substitution as one launch per block step (bs_step_kernel)
// Fault injection, compiled in only with -DSK_TESTING (libskeres_amd_testing.so, `make testing`; never in the product library):
// SK_CHAIN_TEST_WITHHOLD_MARKER=<block column> withholds, once per process, what that column's launch waits for — the wait
// times out (tests/chain_abort_worker.py).
#ifdef SK_TESTING
static std::atomic<int> g_test_withhold{-1};
static std::atomic<int> g_test_janitor_giveup{0};  // SK_BS_TEST_JANITOR_GIVEUP=<n>: from the n-th resident back-substitution with janitors of a whole system on, once
#endif
hipError_t cholesky_init() {
  static std::once_flag once;
  std::call_once(once, [] {
#ifdef SK_TESTING
    if (const char* e = getenv("SK_CHAIN_TEST_WITHHOLD_MARKER")) g_test_withhold.store(atoi(e));
    if (const char* e = getenv("SK_BS_TEST_JANITOR_GIVEUP")) g_test_janitor_giveup.store(atoi(e) > 0 ? 1 : 0);
#endif
    const DevKnobs& k = dev_knobs();
    g_bs_resident.store(k.bs_resident);
    g_chain_server = k.chain_server;
    g_pair_max_trailing = k.pair_max_trailing;
    g_chain_local = k.chain_xcd_local;
  });
  hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_server_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_potrf_lds);
  if (rc != hipSuccess) return rc;
  rc = hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_server_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_potrf_lds);
  if (rc != hipSuccess) return rc;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(potrf128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

CholeskyContext::~CholeskyContext() {
  for (hipEvent_t e : events) (void)hipEventDestroy(e);
  if (fork_ev) (void)hipEventDestroy(fork_ev);
  if (join_ev) (void)hipEventDestroy(join_ev);
  if (sync) (void)hipFree(sync);
  if (xs) (void)hipFree(xs);
}

int* CholeskyContext::sync_for(int nblk) {
  if (nblk > sync_blk) {
    if (sync) (void)hipFree(sync);
    if (xs) (void)hipFree(xs);
    sync = nullptr; xs = nullptr;
    sync_blk = 0;
    const int cap = (nblk + 63) / 64 * 64;
    // (a 128 x 128 scratch block PER block column: the XCD-local hand-over must not read an address twice within a factorisation)
    if (hipMalloc(reinterpret_cast<void**>(&xs), sizeof(double) * 128 * 128 * (size_t)cap) != hipSuccess) { (void)hipGetLastError(); xs = nullptr; return nullptr; }
    if (hipMalloc(reinterpret_cast<void**>(&sync), sizeof(int) * (size_t)(kSyncHeader + kSyncArrays * cap)) != hipSuccess) { (void)hipGetLastError(); sync = nullptr; return nullptr; }
    sync_blk = cap;
  }
  return sync;
}

// The queues of the look-ahead factorisation exist once per DEVICE (device_table.hpp) and are shared by every
// context on that device; they live until process exit (creating a second CU-masked stream after destroying the
// first one hangs on this ROCm build — see DESIGN.md section 4 for what is known about that).  A process may drive
// several devices, one solver handle each: a context's queues are those of the device that was current when it was
// initialised, created under that device.
//
// The queues of the panel stream and of the resident potrf workgroup are chosen by measurement.  With some
// assignments of the process's queues to the command processor's pipes every launch on the panel and bulk streams
// takes 30-50 us longer for as long as the server is resident — the whole factorisation twice as long.  Which
// assignment a process gets follows from the order in which it (this library, PyTorch, RCCL) created its queues: one
// more queue created first flips it (measured: 10.4 / 16.0 / 10.4 / 16.1 ms per iteration with 0 / 1 / 2 / 3 extra
// queues; over the runs, fast exactly when the panel queue had one parity of creation index and the server queue
// the other).  Probes with a spinning kernel and event-ordered empty launches do not show the effect, so two
// queues are created for the bulk streams, four for the panel stream and three for the server, and the first
// factorisation with a resident chain first runs a small synthetic one (identity matrix, banded envelope, 1 ms) with
// each of the 24 combinations (kBulkCand x kPanelCand x kServerCand) and keeps the fastest (tune_chain_queues: 0.1 s, once per device; developer knob
// SK_CHAIN_QUEUES=<n> fixes the combination).
constexpr int kBulkCand = 2, kPanelCand = 4, kServerCand = 3;
struct DeviceQueues {
  int device = -1;
  hipStream_t panel = nullptr, bulk = nullptr, bulk_early = nullptr, server = nullptr;  // the combination in use
  int reserved_cus = 0, early_tiles = 0;
  hipStream_t bulk_candidates[kBulkCand] = {}, bulk_early_candidates[kBulkCand] = {}, panel_candidates[kPanelCand] = {},
              server_candidates[kServerCand] = {};
  std::vector<hipStream_t> all_streams;  // destroyed at exit
  int queue_choice = -1;  // -1: not measured yet; else (bulk * kPanelCand + panel) * kServerCand + server
  std::atomic<bool> tuning{false};  // the queue trial of this device is running (under the table's operation mutex)
  std::atomic<int> chain_server{1};   // 0: block columns are factored launch by launch on this device (knob, or a time-out happened)
  // Solvers on this device that run TWO resident servers per factorisation (a partner front: CholeskyPartner).  There are eight
  // CU 0s; a pair launch whose first server has one and whose second waits for one holds it while it waits, so more than four
  // such solvers factoring at once could wait for each other until the chain's time-out: at most four claim the right.
  std::atomic<int> pair_users{0};
  hipStream_t fork = nullptr;  // stands in for the caller's stream in a secondary context (CholeskyContext::init_secondary)
  hipStream_t plain[2] = {nullptr, nullptr};
  // The panel / bulk / server streams are the DEVICE's: two solvers on one device, driven from two threads, enqueue onto the
  // same streams.  Interleaved, their cross-stream event waits can close a cycle (A's wait on the panel stream in front of
  // B's panel kernel, B's wait on the bulk stream in front of A's SYRK: each waits for the other's queue to move).  A
  // factorisation's launches are therefore enqueued as ONE unit per stream set: [0] the primary set, [1] the secondary one
  // (the tail front of a dissected system, enqueued by its own thread NEXT TO the head's on purpose).
  std::mutex enqueue_mutex[2];
};
static PerDeviceTable<DeviceQueues> g_device_queues;

// stream whose kernels stay off the first `per_xcd` CUs of every XCD (plain stream when per_xcd == 0 or masks are unavailable)
static hipError_t create_bulk_stream(hipStream_t* out, int per_xcd, int ncu, int* reserved) {
  *reserved = 0;
  if (per_xcd > 0 && ncu >= 64 && ncu % 8 == 0 && per_xcd < ncu / 8) {
    // mask bit i <-> XCD i % 8, CU i / 8 of that XCD (measured with tools/cumask_probe.hip)
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int i = per_xcd * 8; i < ncu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()) == hipSuccess) { *reserved = per_xcd * 8; return hipSuccess; }
    (void)hipGetLastError();
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// Creates the queues of device `dev`, which the caller has made current.  nullptr: the streams could not be created.
static DeviceQueues* create_device_queues(int dev) {
  std::unique_ptr<DeviceQueues> q(new DeviceQueues());
  q->device = dev;
  q->chain_server.store(g_chain_server);
  int ncu = 0, reserved = 0, reserved_early = 0;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  // CUs per XCD kept free of the SYRK; developer knobs (0 = no mask)
  // (settled by the sweeps of rounds 1 and 2: profiles/r02_cu_reservation_sweep.txt, tools/sweep_early_tiles.sh)
  const int per_xcd = dev_knobs().bulk_reserve >= 0 ? dev_knobs().bulk_reserve : 4, per_xcd_early = dev_knobs().bulk_reserve_early >= 0 ? dev_knobs().bulk_reserve_early : 2, early_tiles = 72;
  if (dev_knobs().queue_shift > 0) {  // developer variable SK_QUEUE_SHIFT: extra queues first, as another library in the process would create them
    std::vector<uint32_t> all((size_t)(ncu + 31) / 32, 0xffffffffu);
    for (int k = 0; k < dev_knobs().queue_shift; ++k) {
      hipStream_t d = nullptr;
      if (hipExtStreamCreateWithCUMask(&d, (uint32_t)all.size(), all.data()) != hipSuccess) (void)hipGetLastError();
      else q->all_streams.push_back(d);
    }
  }
  auto fail = [&]() -> DeviceQueues* {
    // (streams that were created stay allocated until exit, like every other: see above)
    for (hipStream_t st : q->all_streams) g_orphan_streams.push_back(st);
    return nullptr;
  };
  hipStream_t b = nullptr, be = nullptr;
  if (create_bulk_stream(&b, per_xcd, ncu, &reserved) != hipSuccess) { (void)hipGetLastError(); return fail(); }
  q->all_streams.push_back(b);
  if (create_bulk_stream(&be, per_xcd_early, ncu, &reserved_early) != hipSuccess) { (void)hipGetLastError(); return fail(); }
  q->all_streams.push_back(be);
  q->bulk_candidates[0] = b; q->bulk_early_candidates[0] = be;
  // The resident potrf workgroup gets a CU to itself — CU 0 of whichever XCD its launch lands on: the bulk masks
  // exclude the first CUs of every XCD, the panel mask below CU 0 of every XCD (mask bit i <-> XCD i % 8, CU i / 8; an
  // XCD without any bit would be unrestricted, so a mask cannot choose the XCD) — and a hardware queue to itself: a
  // CU-masked stream owns its queue, a plain one is mapped onto a small shared pool and could end up behind a stream
  // it waits for.  The workgroups that wait for the server can then never occupy the CU it needs (150 KB of LDS).
  hipStream_t p = nullptr, sv = nullptr;
  if (reserved > 0) {
    const size_t words = (size_t)(ncu + 31) / 32;
    std::vector<uint32_t> cu0(words, 0u), rest(words, 0u);
    cu0[0] = 0xffu;
    for (int i = 8; i < ncu; ++i) rest[(size_t)i / 32] |= 1u << (i % 32);
    bool ok = true;
    auto masked = [&](hipStream_t* out, const std::vector<uint32_t>& m) {
      if (!ok) return;
      if (hipExtStreamCreateWithCUMask(out, (uint32_t)words, m.data()) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; ok = false; return; }
      q->all_streams.push_back(*out);
    };
    for (int k = 1; k < kBulkCand && ok; ++k) {
      int r = 0;
      if (create_bulk_stream(&q->bulk_candidates[k], per_xcd, ncu, &r) != hipSuccess || create_bulk_stream(&q->bulk_early_candidates[k], per_xcd_early, ncu, &r) != hipSuccess) {
        (void)hipGetLastError();
        ok = false;
      }
      if (q->bulk_candidates[k]) q->all_streams.push_back(q->bulk_candidates[k]);
      if (q->bulk_early_candidates[k]) q->all_streams.push_back(q->bulk_early_candidates[k]);
    }
    for (int k = 0; k < kPanelCand; ++k) masked(&q->panel_candidates[k], rest);
    for (int k = 0; k < kServerCand; ++k) masked(&q->server_candidates[k], cu0);
    p = q->panel_candidates[0];
    sv = p ? q->server_candidates[0] : nullptr;
    if (!ok) q->queue_choice = 0;  // no choice to make
    if (dev_knobs().chain_queues >= 0) q->queue_choice = dev_knobs().chain_queues % (kBulkCand * kPanelCand * kServerCand);
    if (ok && q->queue_choice >= 0) {
      b = q->bulk_candidates[q->queue_choice / (kPanelCand * kServerCand)];
      be = q->bulk_early_candidates[q->queue_choice / (kPanelCand * kServerCand)];
      p = q->panel_candidates[q->queue_choice / kServerCand % kPanelCand];
      sv = q->server_candidates[q->queue_choice % kServerCand];
    }
  }
  if (!p) {
    if (hipStreamCreateWithFlags(&p, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return fail(); }
    q->all_streams.push_back(p);
  }
  q->server = sv; q->panel = p; q->bulk = b; q->bulk_early = be; q->reserved_cus = reserved; q->early_tiles = early_tiles;
  // destroyed once, at process exit, before the HIP runtime's own teardown (handlers run in reverse order of registration)
  static bool registered = false;
  if (!registered) {
    registered = true;
    atexit([] {
      int cur = 0;
      (void)hipGetDevice(&cur);
      g_device_queues.for_each([](int dev, DeviceQueues& dq) {
        (void)hipSetDevice(dev);
        for (hipStream_t st : dq.all_streams) (void)hipStreamDestroy(st);
        dq.all_streams.clear();
        dq.panel = dq.bulk = dq.bulk_early = dq.server = nullptr;
      });
      (void)hipSetDevice(cur);
    });
  }
  return q.release();
}

void CholeskyContext::use(DeviceQueues* q) {
  dq = q;
  panel = q->panel; bulk = q->bulk; bulk_early = q->bulk_early; server = q->server;
  reserved_cus = q->reserved_cus; early_tiles = q->early_tiles;
}

hipError_t CholeskyContext::init() {
  if (panel) return hipSuccess;
  int dev = 0;
  hipError_t rc = hipGetDevice(&dev);
  if (rc != hipSuccess) return rc;
  DeviceQueues* q = g_device_queues.get_or_create(dev, create_device_queues);
  if (!q) return hipErrorOutOfMemory;
  device = dev;
  use(q);
  return hipSuccess;
}

hipError_t CholeskyContext::init_secondary(const CholeskyContext& primary) {
  panel = bulk = bulk_early = nullptr;  // (called again after the queue choice: the queues the primary context does NOT use now)
  if (!primary.dq) return hipErrorInvalidValue;
  DeviceQueues* q = primary.dq;
  std::lock_guard<std::mutex> lock(g_device_queues.operation_mutex());  // (the device's entry may gain streams below)
  dq = q; device = primary.device;
  // A resident potrf server of its own (round 3): one of the candidates the primary context does not use, on the same
  // CU mask (CU 0 of every XCD: eight CUs, a server each).
  server = nullptr;
  if (primary.server)
    for (int k = kServerCand - 1; k >= 0 && !server; --k) if (q->server_candidates[k] && q->server_candidates[k] != primary.server) server = q->server_candidates[k];
  prepared = true;  // (the queues are chosen here, from what the device's trial left over: cholesky_prepare must not adopt the primary's)
  reserved_cus = q->reserved_cus; early_tiles = q->early_tiles;
  {
    // The queues the primary context does not use — of the same creation parity as the ones it does: which queues suffer
    // from a resident kernel on the server's queue goes by that parity (tune_chain_queues), and the primary's were
    // measured to be on the good side.
    int kp = -1, kb = -1;
    for (int k = 0; k < kPanelCand; ++k) if (q->panel_candidates[k] == primary.panel) kp = k;
    for (int k = 0; k < kBulkCand; ++k) if (q->bulk_candidates[k] == primary.bulk) kb = k;
    int bp = -1, bb = -1;
    for (int k = 0; k < kPanelCand; ++k) if (k != kp && q->panel_candidates[k] && (kp < 0 || (k - kp) % 2 == 0)) { bp = k; break; }
    if (bp < 0) for (int k = 0; k < kPanelCand; ++k) if (k != kp && q->panel_candidates[k]) { bp = k; break; }
    for (int k = 0; k < kBulkCand; ++k) if (k != kb && q->bulk_candidates[k]) { bb = k; break; }
    if (bp >= 0) panel = q->panel_candidates[bp];
    if (bb >= 0) { bulk = q->bulk_candidates[bb]; bulk_early = q->bulk_early_candidates[bb]; }
    if (dev_knobs().debug_queues) std::fprintf(stderr, "[skeres_amd] secondary context: panel candidate %d (primary %d), bulk candidate %d (primary %d)\n", bp, kp, bb, kb);
  }
  if (!panel || !bulk) { panel = bulk = bulk_early = nullptr; server = nullptr; return hipErrorNotSupported; }  // (no CU-masked candidates on this device)
  if (!q->fork) {
    if (hipStreamCreateWithFlags(&q->fork, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); panel = bulk = bulk_early = nullptr; return hipErrorOutOfMemory; }
    q->all_streams.push_back(q->fork);
  }
  fork = q->fork;
  return hipSuccess;
}

hipEvent_t CholeskyContext::event(size_t i) {
  while (events.size() <= i) {
    hipEvent_t e = nullptr;
    // ordering between streams of ONE device: no system-scope fence (cache write-back and invalidation towards the host)
    // when the event fires — the kernels' own device-scope release at their end is what the consumers need
    (void)hipEventCreateWithFlags(&e, g_event_flags);
    events.push_back(e);
  }
  return events[i];
}

// Block-column groups of the factorisation: `group` columns while the trailing matrix is large (the
// SYRK is the long pole and wants a deep K), then groups of kTailGroup once fewer than kTailTiles tile
// rows remain: there the serial panel chain is the long pole and a narrow group keeps it short (no lazy
// updates inside the group, a K = 128 update for the next panel).  Returns the group start indices + nblk.
std::vector<int> cholesky_group_bounds(int nblk, int group) {
  std::vector<int> b;
  int k = 0;
  while (k < nblk) {
    b.push_back(k);
    const int remaining = nblk - k;
    int g = remaining <= g_tail_tiles ? g_tail_group : group;
    if (g < 1) g = 1;
    k += g < remaining ? g : remaining;
  }
  b.push_back(nblk);
  return b;
}

// The groups actually used, and the block columns under the resident panel chain (potrf_server_kernel /
// chain_column_kernel): the runs of at least kMinChainRun block columns whose trailing update is short (at most
// g_chain_max_trailing tile rows: the serial chain, not the SYRK, decides there — measured on Ladybug-1723, a
// column cycle is 40-50 us up to 13 trailing tile rows, 60 us at 20, 100 us at 30) and whose panels fit the 32-row
// tiles — on Ladybug-1723 the first 8 block columns and the last 77.  Those columns are groups of one; the columns
// between the runs are factored launch by launch in groups of `group` (of g_chain_prefix_group when the caller
// asked for 1: their SYRK is the long pole, and K = 128 leaves it bound by the traffic of its C tiles).
// chain == false: cholesky_group_bounds, no resident chain.
constexpr int kMinChainRun = 4;
// ncols < nblk (a PARTIAL factorisation: a leaf front of the dissected reduced system, cholesky_dissected_*): only the
// first ncols block columns are factored; the trailing block rows — the front's border — receive their updates and are
// left holding the Schur complement.  The groups then cover [0, ncols), followed by [ncols, ncols + 1) and
// [ncols + 1, nblk), which cholesky_factor never factors: they only tell the last real group where next(g) (one block
// column, as after a resident column) ends and syrk(g) begins.
static void plan_close_partial(CholeskyPlan* plan, int nblk, int ncols) {
  std::vector<int>& b = plan->bounds;
  while (!b.empty() && b.back() >= ncols) b.pop_back();
  b.push_back(ncols);
  if (ncols + 1 < nblk) b.push_back(ncols + 1);
  if (ncols < nblk) b.push_back(nblk);
}
// tail_rows: how many block rows at the end of the matrix are active in EVERY block column (1: the block row that carries
// the right-hand side; more for a leaf front whose border has rows that the first columns of the interior couple with —
// the "spike" of a segment between two separators); only a partial factorisation has more than one.
CholeskyPlan cholesky_plan(int nblk, int group, const int* last, bool chain, int ncols, int tail_rows, const int* tail) {
  CholeskyPlan plan;
  plan.resident.assign(nblk, 0);
  plan.paired.assign(nblk, 0);
  if (group < 1) group = 1;
  if (ncols < 0 || ncols > nblk) ncols = nblk;
  if (tail_rows < 1 || ncols > nblk - tail_rows) tail_rows = 1;
  const int tail_uniform = nblk - tail_rows;  // first block row of the tail
  auto tail0_of = [&](int c) { return tail ? tail[c < nblk ? c : nblk - 1] : tail_uniform; };  // ... as block column c sees it (a profile: cholesky_factor)
  const bool partial = ncols < nblk;
  if (!chain || nblk < 3) {
    plan.bounds = cholesky_group_bounds(nblk, group);
    if (partial) {
      // group boundaries of [0, ncols) (a group never straddles the border), then the border's pseudo-groups
      std::vector<int> b;
      for (int k : plan.bounds) if (k < ncols) b.push_back(k);
      plan.bounds.swap(b);
      plan_close_partial(&plan, nblk, ncols);
    }
    return plan;
  }
  auto last_main = [&](int c) { return last ? (last[c] < nblk - 1 ? last[c] : nblk - 1) : nblk - 1; };
  auto count = [&](int first_row, int col) {  // active block rows of block column col from first_row: its run, and the tail rows
    const int last_row = last_main(col);
    const int main_rows = last_row >= first_row ? last_row - first_row + 1 : 0;
    return main_rows + std::max(0, nblk - std::max(tail0_of(col), first_row + main_rows));
  };
  const int jend = partial ? ncols : nblk - 1;  // block columns that have a column launch of their own
  for (int j = 0; j < jend; ++j)
    plan.resident[j] = count(j + 2, j) <= g_chain_max_trailing && 4 * count(j + 1, j) <= g_thin_grid;
  for (int j = 0; j < jend;) {  // drop the short runs: a hand-over costs more than a few columns gain
    if (!plan.resident[j]) { ++j; continue; }
    int e = j;
    while (e < jend && plan.resident[e]) ++e;
    if (e - j < kMinChainRun) for (int i = j; i < e; ++i) plan.resident[i] = 0;
    j = e;
  }
  // Resident PAIRS (round 3): the block columns between two resident runs whose trailing update is too long for a K = 128
  // SYRK per column (it would be bound by the traffic of its C tiles) but short enough for the 32-row tiling
  // (g_pair_max_trailing) are taken two at a time UNDER THE SERVER as well: the first column's launch applies its panel to the
  // second column only, the second column's launch is a TRSM, and ONE K = 256 SYRK applies both panels to everything from the
  // next pair on — its first diagonal block, written through and counted by its own workgroups, is what the server waits
  // for.  The serial chain of a pair is then two resident column cycles plus the first block column of that SYRK instead of
  // eight launches with their dispatch and completion latencies; the SYRKs follow each other on the bulk stream as before.
  if (g_pair_max_trailing > 0) {
    bool any_resident = false;
    for (int j = 0; j < jend; ++j) any_resident = any_resident || plan.resident[j];
    auto pairable = [&](int c) { return !plan.resident[c] && count(c + 2, c) <= g_pair_max_trailing && count(c + 2, c) >= 1 && 4 * count(c + 1, c) <= g_thin_grid; };
    for (int j = 0; any_resident && j < jend;) {
      if (!pairable(j)) { ++j; continue; }
      int e = j;
      while (e < jend && pairable(e)) ++e;
      for (int c = j; c + 1 < e; c += 2) { plan.paired[c] = 1; plan.paired[c + 1] = 2; }
      for (int c = j; c < e; ++c) plan.resident[c] = 1;  // (an odd column left over at the end of the run: resident on its own, one K = 128 SYRK)
      j = e;
    }
  }
  if (!partial) plan.resident[nblk - 1] = plan.resident[nblk - 2];  // the server factors the last diagonal block too when it has the column before it
  bool any = false;
  for (int j = 0; j < nblk; ++j) any = any || plan.resident[j];
  if (!any) return cholesky_plan(nblk, group, last, false, ncols, tail_rows, tail);
  const int pg = group == 1 ? g_chain_prefix_group : group;
  for (int k = 0; k < ncols;) {
    plan.bounds.push_back(k);
    if (plan.paired[k] == 1) { k += 2; continue; }
    if (plan.resident[k]) { ++k; continue; }
    int stop = k;
    while (stop < ncols && !plan.resident[stop]) ++stop;
    int g = nblk - k <= g_tail_tiles ? g_tail_group : pg;
    if (g < 1) g = 1;
    k += std::min(g, stop - k);
  }
  if (partial) plan_close_partial(&plan, nblk, ncols);
  else plan.bounds.push_back(nblk);
  return plan;
}
int cholesky_plan_max_group(const CholeskyPlan& plan) {
  int m = 1;
  for (size_t g = 0; g + 1 < plan.bounds.size(); ++g) m = std::max(m, plan.bounds[g + 1] - plan.bounds[g]);
  return m;
}

// Factor the lower triangle of S (npad x ld) in place.  Linv: nblk blocks of 128x128
// (zero-initialised once).  Right-looking over groups of `group` block columns, lazy
// left-looking inside a group.
//
// Per group g (block columns [k0, k1)):
//   panel(g)      potrf128 / TRSM of each block column, lazy updates from the columns of the group before it
//   next(g)       the part of the trailing update that the NEXT panel needs: block columns [k1, k1 + group)
//   syrk(g)       the rest of the trailing update (the dominant launch)
// panel(g+1) depends on next(g) only and next(g) on panel(g) and syrk(g-1), so with a context
// next(g), panel(g+1) run on the panel stream next to syrk(g) on the bulk stream and the SYRKs
// follow each other without a gap: the serial chain (potrf128 is one workgroup; the TRSM and update
// grids are a few hundred workgroups) is hidden behind the SYRK as long as the SYRK is the longer
// of the two, which it is for the first half of the groups (87 % of the flops).  The bulk stream's CU
// mask keeps a few CUs per XCD free: without them the panel kernels sit behind the SYRK's
// workgroups in the dispatcher and nothing overlaps (measured).
// `last` (optional, nblk entries): the block envelope of the matrix.  last[c] >= c is the last block row, among
// rows 0 .. nblk-2, in which block column c of the FACTOR can be non-zero (non-decreasing in c); the final block
// row nblk-1, which carries the right-hand side, is always active.  Every launch then covers the active rows only:
// a contiguous run plus that last row (main_t / jump_t of the GEMM body).  Blocks outside are exact zeros in the
// dense algorithm too (0 - 0 * x), so the result is bit-identical to last == nullptr; only the work differs
// (Ladybug-1723-shaped S: 0.17 of 1.27 TFlop).
__global__ void set_identity_kernel(double* A, long ld, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[(long)i * ld + i] = 1.0;
}

// See DeviceQueues::queue_choice: a small synthetic factorisation (identity matrix — the launches do not depend on the values —, an
// envelope four blocks high, every column under the resident chain) with each pair of candidate queues, on the
// caller's stream: its queue is part of the assignment being measured.
static void tune_chain_queues(CholeskyContext* ctx, hipStream_t s) {
  DeviceQueues& q = *ctx->dq;
  q.queue_choice = 0;
  const int nblk = 26, nblk2 = 32, n = nblk2 * 128;  // (stage 1 factors the leading 26 blocks of A, stage 2 all 32)
  double *A = nullptr, *Linv = nullptr;
  int* info = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&A), sizeof(double) * (size_t)n * n) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&Linv), sizeof(double) * (size_t)n * 128) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&info), sizeof(int)) != hipSuccess) {
    (void)hipGetLastError();
    if (A) (void)hipFree(A);
    if (Linv) (void)hipFree(Linv);
    return;
  }
  (void)hipMemsetAsync(A, 0, sizeof(double) * (size_t)n * n, s);
  (void)hipMemsetAsync(Linv, 0, sizeof(double) * (size_t)n * 128, s);
  (void)hipMemsetAsync(info, 0, sizeof(int), s);
  hipLaunchKernelGGL(set_identity_kernel, dim3((n + 255) / 256), dim3(256), 0, s, A, (long)n, n);
  std::vector<int> last(nblk);
  for (int c = 0; c < nblk; ++c) last[c] = std::min(c + 4, nblk - 2);
  last[nblk - 1] = nblk - 1;
  q.tuning = true;
  ctx->in_trial = true;
  const int ncomb = kBulkCand * kPanelCand * kServerCand;
  std::vector<double> ms((size_t)ncomb, 1e30);
  auto select = [&](int c) {
    q.bulk = q.bulk_candidates[c / (kPanelCand * kServerCand)];
    q.bulk_early = q.bulk_early_candidates[c / (kPanelCand * kServerCand)];
    q.panel = q.panel_candidates[c / kServerCand % kPanelCand];
    q.server = q.server_candidates[c % kServerCand];
    ctx->use(&q);
  };
  bool serialised = false;
  // Stage 1, every combination: the banded system above, all of it under the resident chain (the hand-overs between the
  // server, the column launches and the SYRKs are what a bad combination delays: 1.0 ms against 1.5-4).
  for (int c = 0; c < ncomb && !serialised; ++c) {
    select(c);
    for (int rep = 0; rep < 3; ++rep) {  // (the first one also pages the kernels in)
      (void)hipStreamSynchronize(s);
      const auto t0 = std::chrono::steady_clock::now();
      cholesky_factor(A, n, nblk * 128, Linv, info, 1, s, ctx, nullptr, last.data(), true);
      (void)hipStreamSynchronize(s);
      const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep > 0 && t < ms[c]) ms[c] = t;
      if (c == 0 && rep == 0) {
        // A tool that serialises the kernels of the process (counter collection does) lets the server run alone: it
        // waits for a column launch that cannot start, gives up (kChainTimeoutTicks) and sets info.  Then every
        // block column is factored launch by launch for the rest of the process — the same plan, the same launches.
        int h = 0;
        (void)hipMemcpyAsync(&h, info, sizeof(int), hipMemcpyDeviceToHost, s);  // (not the null stream: it would wait for other solvers' resident servers)
        (void)hipStreamSynchronize(s);
        if (h != 0) { serialised = true; break; }
      }
    }
  }
  if (serialised) {
    q.tuning = false;
    ctx->in_trial = false;
    q.chain_server = 0;
    std::fprintf(stderr, "[skeres_amd] the resident panel chain timed out in its trial run (are kernels being serialised, e.g. by counter "
                         "collection?): block columns are factored launch by launch in this process\n");
    select(0);
    q.queue_choice = 0;
    (void)hipFree(A); (void)hipFree(Linv); (void)hipFree(info);
    return;
  }
  double best = ms[0];
  for (int c = 1; c < ncomb; ++c) best = std::min(best, ms[c]);
  if (best > 8.0) {
    // 26 block columns take 1.1 ms.  Several times that with every combination: the device is being time-sliced with
    // another process, and kernels that wait for each other lose whole slices: launch by launch for this process.
    q.chain_server = 0;
    std::fprintf(stderr, "[skeres_amd] the resident panel chain ran %.0f times slower than it should in its trial run (is the device shared with "
                         "another process?): block columns are factored launch by launch in this process\n", best / 1.1);
  }
  // Stage 2, the combinations within noise of the best: a dense system factored launch by launch in groups of two with
  // every SYRK on the queue of the EARLY groups, which a banded system never uses and a dense one uses for most of its
  // flops — a combination can be right for one and wrong for the other (Ladybug-1723: four combinations at 9.1 ms per
  // iteration, of which one factors the dense system of the same size in 34.7 ms and the others in 29.2-29.6).
  std::vector<double> ms2((size_t)ncomb, 1e30);
  double best2 = 1e30;
  for (int c = 0; c < ncomb; ++c) {
    if (ms[c] > 1.08 * best) continue;
    select(c);
    ctx->early_tiles = 0;
    double sum = 0.0;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipStreamSynchronize(s);
      const auto t0 = std::chrono::steady_clock::now();
      cholesky_factor(A, n, n, Linv, info, 2, s, ctx, nullptr, nullptr, false);
      (void)hipStreamSynchronize(s);
      if (rep > 0) sum += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    ms2[c] = sum / 3.0;  // (the mean: a conflict between two queues shows in some runs and not in others)
    best2 = std::min(best2, ms2[c]);
  }
  for (int c = ncomb - 1; c >= 0; --c) if (ms2[c] <= 1.04 * best2) q.queue_choice = c;  // the first of those within noise of the best: the same choice run after run
  if (dev_knobs().debug_queues) {
    std::fprintf(stderr, "[skeres_amd] synthetic factorisations (ms: banded under the resident chain / dense on the early-group queue) per (bulk, panel, server) queue candidates:");
    for (int c = 0; c < ncomb; ++c) {
      std::fprintf(stderr, "%s%.2f", c % kServerCand == 0 ? "  " : " ", ms[c]);
      if (ms2[c] < 1e29) std::fprintf(stderr, "/%.2f", ms2[c]);
    }
    std::fprintf(stderr, ": combination %d (device %d)\n", q.queue_choice, q.device);
  }
  q.tuning = false;
  ctx->in_trial = false;
  select(q.queue_choice);
  (void)hipFree(A);
  (void)hipFree(Linv);
  (void)hipFree(info);
}

// What a caller read back from `info` after a factorisation.  2 is the resident panel chain giving up on a wait
// (kChainTimeoutTicks): the device is shared or its kernels are being serialised.  That factorisation is lost; the
// device factors launch by launch from here on, and the caller factors the same matrix again that way (BalSolver::try_step).
// Returns true when the caller should do that.
bool cholesky_note_info(CholeskyContext* ctx, int info) {
  if (info == 2) g_bs_resident.store(0);  // (whichever resident kernel it was: the back-substitution is launch by launch from here on, too)
  if (info == 2 && ctx && ctx->sync && dev_knobs().debug_chain_abort) {  // developer knob: where the chain stood when a wait gave up
    std::vector<int> h((size_t)kSyncHeader + kSyncArrays * ctx->sync_blk);
    (void)hipMemcpy(h.data(), ctx->sync, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[skeres_amd] chain abort: potrf_done %d abort %d syrk_seq %d syrk_column %d | diag_ready:", h[kSyncPotrfDone], h[kSyncAbort], h[kSyncSyrkSeq], h[kSyncSyrkColumn]);
    for (int j = 0; j < 40 && j < ctx->sync_blk; ++j) std::fprintf(stderr, " %d", h[kSyncHeader + j]);
    std::fprintf(stderr, " | x_ready:");
    for (int j = 0; j < 40 && j < ctx->sync_blk; ++j) std::fprintf(stderr, " %d", h[kSyncHeader + ctx->sync_blk + j]);
    std::fprintf(stderr, "\n");
  }
  if (info != 2) return false;
  // (no look-ahead queues — a solver with the look-ahead off, or one that replays its iteration as a graph: the wait that gave up
  // was the resident back-substitution's.  It is switched off above; the same system is to be solved again — ADVICE r03)
  if (!ctx || !ctx->dq) {
    std::fprintf(stderr, "[skeres_amd] the resident back-substitution timed out: one launch per block step from now on\n");
    return true;
  }
  std::lock_guard<std::mutex> lock(g_device_queues.operation_mutex());
  if (ctx->dq->chain_server) {
    ctx->dq->chain_server = 0;
    std::fprintf(stderr, "[skeres_amd] the resident panel chain timed out on device %d (is the device shared with another process, or are "
                         "kernels being serialised?): block columns are factored launch by launch on it from now on\n", ctx->dq->device);
  }
  return true;
}
// Another rank of the world lost its resident chain (or never had one): every rank must factor by the same plan, or the
// replicated factorisations differ in rounding and the ranks' parameters drift apart bit by bit.
void cholesky_disable_chain(CholeskyContext* ctx) {
  if (!ctx || !ctx->dq) return;
  std::lock_guard<std::mutex> lock(g_device_queues.operation_mutex());
  ctx->dq->chain_server = 0;
}
bool cholesky_chain_enabled(const CholeskyContext* ctx) { return ctx && ctx->dq && ctx->server && ctx->resident && ctx->dq->chain_server; }
bool cholesky_claim_pair_servers(CholeskyContext* ctx) {
  if (!ctx || !ctx->dq) return false;
  if (ctx->dq->pair_users.fetch_add(1) < 4) return true;
  ctx->dq->pair_users.fetch_sub(1);
  return false;
}
void cholesky_release_pair_servers(CholeskyContext* ctx) { if (ctx && ctx->dq) ctx->dq->pair_users.fetch_sub(1); }

// Before the first factorisation with allow_chain on stream s (cholesky_factor does it otherwise): choose the queues.
// One trial per device, one at a time (two solvers on two threads would otherwise measure each other).
void cholesky_prepare(CholeskyContext* ctx, hipStream_t s) {
  if (!ctx || !ctx->dq || !ctx->server || ctx->prepared) return;
  // (a factorisation that finds the trial of its device running waits here for it to end, instead of running alongside
  // it and skewing what it measures; the trial's own factorisations never get here: cholesky_factor checks `tuning`)
  std::lock_guard<std::mutex> lock(g_device_queues.operation_mutex());
  DeviceQueues& q = *ctx->dq;
  // (a context that may not run resident kernels — sk_options_set_resident_kernels(o, 0) — does not run the trial either: its
  // first stage IS a resident chain; such a solver takes the device's choice if another solver has made one, else combination 0)
  if (!q.chain_server || !ctx->resident) { if (q.queue_choice >= 0) { ctx->use(&q); ctx->prepared = true; } return; }
  if (q.queue_choice < 0) tune_chain_queues(ctx, s);
  ctx->use(&q);  // (every context of the device: the choice is the device's)
  ctx->prepared = q.queue_choice >= 0;  // the choice is final: later factorisations of this context take no lock
}

void cholesky_factor(double* S, long ld, int npad, double* Linv, int* info, int group, hipStream_t s, CholeskyContext* ctx,
                     KernelTimer* kt, const int* last, bool allow_chain, int ncols, int tail_rows, const CholeskyPartner* partner, const int* tail) {
  const int nblk = npad / 128;
  if (ncols < 0 || ncols > nblk) ncols = nblk;
  if (tail_rows < 1 || ncols > nblk - tail_rows) tail_rows = 1;
  const int tail_uniform = nblk - tail_rows;
  // first block row of the tail as block column c sees it (a profile: the border of a bordered envelope; else uniform)
  auto tail0_of = [&](int c) { return tail ? tail[c < nblk ? c : nblk - 1] : tail_uniform; };
  // Which hardware queues the panel and bulk streams sit on decides how well their kernels overlap — with or without the
  // resident chain (Venice-1778 in explicit groups of two: 30.6 ms per iteration on the first combination, 19.1 on the
  // one the trial picks): every look-ahead factorisation asks for the trial, which runs once per device.
  if (ctx && ctx->dq && !ctx->prepared && !(ctx->dq->tuning && ctx->in_trial)) cholesky_prepare(ctx, s);
  const bool la = ctx != nullptr && ctx->panel != nullptr && ctx->bulk != nullptr;
  std::unique_lock<std::mutex> enqueue_lock;
  const auto t_wait0 = std::chrono::steady_clock::now();
  if (la && ctx->dq) enqueue_lock = std::unique_lock<std::mutex>(ctx->dq->enqueue_mutex[ctx->fork ? 1 : 0]);
  struct EnqueueClock {  // developer knob SK_DEBUG_CHAIN_ABORT: a factorisation whose launches took the host unusually long to enqueue
    std::chrono::steady_clock::time_point t0, t1 = std::chrono::steady_clock::now();
    ~EnqueueClock() {
      const bool on = dev_knobs().debug_chain_abort;
      const double held = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
      const double waited = std::chrono::duration<double, std::milli>(t1 - t0).count();
      if (on && (held > 50.0 || waited > 500.0)) std::fprintf(stderr, "[skeres_amd] cholesky_factor: waited %.1f ms for the device's enqueue lock, enqueued for %.1f ms\n", waited, held);
    }
  } enqueue_clock;
  enqueue_clock.t0 = t_wait0;
  hipStream_t sp = la ? ctx->panel : s;
  hipStream_t sb = la ? ctx->bulk : s;  // of the current group (chosen below)
  hipStream_t sb_prev = sb;
  size_t ev = 0;
  auto order = [&](hipStream_t from, hipStream_t to) {  // work enqueued on `to` from here on runs after everything enqueued on `from` so far
    if (from == to) return;
    hipEvent_t e = ctx->event(ev++);
    (void)hipEventRecord(e, from);
    (void)hipStreamWaitEvent(to, e, 0);
  };
  // last active main row of block column c, and whether block row nblk-1 comes on top of the run that ends there
  auto last_main = [&](int c) { return last ? (last[c] < nblk - 1 ? last[c] : nblk - 1) : nblk - 1; };
  struct Rows { int main, extra, jump; };  // `main` consecutive block rows from `first_row`, then `extra` rows of the tail (what is left of it), `jump` blocks further
  // the active block rows of block column `col` from first_row on (a group's update takes those of the group's LAST column:
  // envelope and tail profile both widen from column to column, and what the earlier columns lack of them is exact zeros)
  auto rows_from = [&](int first_row, int col) {
    Rows r;
    const int last_row = last_main(col);
    r.main = last_row >= first_row ? last_row - first_row + 1 : 0;
    const int t0 = std::max(tail0_of(col), first_row + r.main);
    r.extra = std::max(0, nblk - t0);
    r.jump = r.extra ? t0 - (first_row + r.main) : 0;
    return r;
  };
  // C[rows x tiles_n tiles of 128] -= A B^T; rows as above, 64-row tiles
  auto update = [&](hipStream_t st, const char* name, double* C, const double* A, const double* B, int K, Rows r, int tiles_n, int flags) {
    const int tiles_m = r.main + r.extra;
    if (tiles_m <= 0 || tiles_n <= 0) return;
    if (kt) kt->begin(name, st);
    if (4 * tiles_m * tiles_n <= g_thin_grid)
      hipLaunchKernelGGL(gemm_update_thin_f64_kernel, dim3(4 * tiles_m * tiles_n), dim3(256), 0, st, C, ld, A, ld, B, ld, K, 4 * tiles_m, flags, 4 * r.main,
                         4 * r.jump);
    else
      hipLaunchKernelGGL(gemm_update_f64_kernel, dim3(2 * tiles_m * tiles_n), dim3(256), 0, st, C, ld, A, ld, B, ld, K, 2 * tiles_m, flags, 2 * r.main,
                         2 * r.jump);
    if (kt) kt->end(name, st);
  };
  // one diagonal 128-block: C -= A A^T
  auto update_diag = [&](hipStream_t st, const char* name, double* C, const double* A, int K) {
    if (kt) kt->begin(name, st);
    hipLaunchKernelGGL(gemm_diag_f64_kernel, dim3(4), dim3(256), 0, st, C, ld, A, ld, K);
    if (kt) kt->end(name, st);
  };
  // the completion of the group's last TRSM is what syrk(g) waits for: the event rides on that kernel's own dispatch
  // (hipExtLaunchKernelGGL stop event) instead of a separate record, which would sit in the panel queue as a packet
  // of its own in front of the next chain kernel (about 10 us)
  hipEvent_t panel_done = nullptr;
  auto panel = [&](int k0, int k1) {
    panel_done = nullptr;
    for (int kb = k0; kb < k1; ++kb) {
      double* Akk = S + (long)kb * 128 * ld + (long)kb * 128;
      const Rows below = rows_from(kb + 1, kb);
      const double* P = S + (long)kb * 128 * ld + (long)k0 * 128;
      const int K = (kb - k0) * 128;
      if (kb > k0) update_diag(sp, "gemm_diag_update", Akk, P, K);  // lazy left-looking update from columns [k0, kb): diagonal tile ...
      if (kt) kt->begin("potrf128", sp);
      hipLaunchKernelGGL(potrf128_kernel, dim3(1), dim3(256), potrf128_lds_bytes(), sp, Akk, ld, Linv + (long)kb * 128 * 128, info);
      if (kt) kt->end("potrf128", sp);
      if (below.main + below.extra > 0) {
        double* A21 = Akk + 128 * ld;
        if (kb > k0) update(sp, "gemm_panel_update", A21, P + 128 * ld, P, K, below, 1, 0);  // ... and the rows below it
        hipEvent_t stop = nullptr;
        if (la && g_ext_events && kb == k1 - 1) stop = panel_done = ctx->event(ev++);
        if (kt) kt->begin("gemm_trsm", sp);
        if (4 * (below.main + below.extra) <= g_thin_grid)
          hipExtLaunchKernelGGL(trsm_gemm_thin_f64_kernel, dim3(4 * (below.main + below.extra)), dim3(256), 0, sp, nullptr, stop, 0, A21, ld,
                                (const double*)A21, ld, (const double*)(Linv + (long)kb * 128 * 128), 4 * (below.main + below.extra), 4 * below.main,
                                4 * below.jump);
        else
          hipExtLaunchKernelGGL(trsm_gemm_f64_kernel, dim3(2 * (below.main + below.extra)), dim3(256), 0, sp, nullptr, stop, 0, A21, ld,
                                (const double*)A21, ld, (const double*)(Linv + (long)kb * 128 * 128), 2 * (below.main + below.extra), 2 * below.main,
                                2 * below.jump);
        if (kt) kt->end("gemm_trsm", sp);
      }
    }
  };
  // the block columns under the resident panel chain (cholesky_plan)
  const CholeskyPlan plan = cholesky_plan(nblk, group, last, la && allow_chain && ctx->server != nullptr, ncols, tail_rows, tail);
  const std::vector<int>& gb = plan.bounds;
  const int ngroups = (int)gb.size() - 1;
  ChainRanges ranges;
  ranges.n = 0;
  for (int j = 0; j < nblk; ++j)
    if (plan.resident[j] && (j == 0 || !plan.resident[j - 1])) {
      if (ranges.n == 8) { ranges.n = 0; break; }  // (more runs than the server takes: launch by launch)
      int e = j;
      while (e < nblk && plan.resident[e]) ++e;
      ranges.begin[ranges.n] = j; ranges.end[ranges.n] = e; ++ranges.n;
    }
  // the same groups launch by launch: while every chain kernel is being timed, and with SK_CHOL_CHAIN_SERVER=0 (PMC
  // passes serialise the kernels of a process: a resident kernel that waits for another one would time out)
  bool chain = ranges.n > 0 && ctx->dq && ctx->dq->chain_server && ctx->resident && !(kt && kt->times_all());
  int* sync = chain ? ctx->sync_for(nblk) : nullptr;
  if (!sync) chain = false;  // (the same groups, launch by launch)
  auto is_resident = [&](int k) { return chain && k < nblk && plan.resident[k]; };
  // the XCD-local hand-over between the server and the tiles of block row j + 1 (above): every resident single column's launch
  // (SK_CHAIN_XCD_LOCAL=2: only for a factorisation of ONE front — it pays there and not where a partner front rides in the launches)
  const int local = chain && (g_chain_local == 1 || (g_chain_local == 2 && !(partner && partner->ncols > 0))) ? 1 : 0;
  auto column_grid = [&](int ncrit, int T, int ncand) { return ncrit ? ncand + 4 * T - 4 : 4 * T; };
  const int maxblk = chain ? ctx->sync_blk : 0;
  const char* stamps_file = chain ? dev_knobs().chain_stamps : nullptr;
  hipStream_t srv = nullptr;
  hipEvent_t start_ev = nullptr;
  int start_value = 0;  // != 0: the server waits in the kernel for chain_start_kernel, and the join is chain_gate_kernel (no events: below)
  struct PartnerState {
    bool on = false; CholeskyPlan plan; int nblk = 0, ncols = 0, tail0 = 0, start_at = 0, maxblk = 0; const int* last = nullptr; const int* tail = nullptr;
    double* S = nullptr; long ld = 0; double* Linv = nullptr; int* sync = nullptr; double* xs = nullptr;
    int next = 0, seq = 0, col_seq = 0;
  } pb;
  // the partner's block column kB: its column launch and its thin SYRK (Tb == 0: none), as the resident branch below forms them
  struct PartnerStep { ColumnArgs col; ThinSyrkArgs syrk; int Tb; bool next_resident, by_column; };
  auto partner_step = [&](int kB) {
    PartnerStep st;
    auto lm = [&](int c) { return pb.last ? (pb.last[c] < pb.nblk - 1 ? pb.last[c] : pb.nblk - 1) : pb.nblk - 1; };
    auto rf = [&](int first_row, int last_row) {  // (of block column kB: its run and its tail rows)
      Rows r;
      r.main = last_row >= first_row ? last_row - first_row + 1 : 0;
      const int t0 = std::max(pb.tail ? pb.tail[kB] : pb.tail0, first_row + r.main);
      r.extra = std::max(0, pb.nblk - t0);
      r.jump = r.extra ? t0 - (first_row + r.main) : 0;
      return r;
    };
    const int k1 = kB + 1, Lg = lm(kB);
    const Rows rn = rf(k1, Lg), rs = rf(k1 + 1, Lg);
    const int T = rn.main + rn.extra, ncrit = (rn.main > 0 || rn.jump == 0) ? 16 : 0;
    st.Tb = rs.main + rs.extra;
    st.next_resident = k1 < pb.ncols;
    st.by_column = st.next_resident && g_early_column;
    const int ncand = ncrit && local ? kCritCandidates : ncrit;
    st.col = ColumnArgs{pb.S, pb.ld, kB, pb.Linv + (long)kB * 128 * 128, 4 * T, 4 * rn.main, 4 * rn.jump, ncrit, pb.xs, pb.sync, pb.maxblk, pb.seq, pb.col_seq,
                        column_grid(ncrit, T, ncand), ncand};
    st.syrk = ThinSyrkArgs{pb.S + (long)(k1 + 1) * 128 * pb.ld + (long)(k1 + 1) * 128, pb.ld, pb.S + (long)(k1 + 1) * 128 * pb.ld + (long)kB * 128, pb.ld, 128, 4 * st.Tb,
                           4 * rs.main, 4 * rs.jump, rs.main, rs.jump, st.by_column ? pb.sync + kSyncSyrkColumn : (int*)nullptr, st.Tb > 0 ? 2 * st.Tb * (st.Tb + 1) : 0};
    return st;
  };
  // ... and what its SYRK leaves for its next column launch to wait for (after the launch that carried it)
  auto partner_advance = [&](const PartnerStep& st, hipStream_t bulk_stream) {
    if (st.Tb > 0) {
      if (st.by_column) pb.col_seq += 4 * st.Tb;
      else if (st.next_resident) { ++pb.seq; hipLaunchKernelGGL(chain_marker_kernel, dim3(1), dim3(1), 0, bulk_stream, pb.sync + kSyncSyrkSeq, pb.seq); }
    }
    ++pb.next;
  };
  if (chain) {
    const int stamps_on = stamps_file ? 1 : 0;
    static int stamps_state = 0;
    if (stamps_on != stamps_state) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stamps_on), &stamps_on, sizeof(int)); stamps_state = stamps_on; }
    srv = ctx->server;
    // (whether the server waits in the kernel for its start signal — which then resets the counters too — or behind an event)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusActive; }
    const bool early = dev_knobs().chain_early_server != 0 && capturing == hipStreamCaptureStatusNone;
    if (!early) (void)hipMemsetAsync(sync, 0, sizeof(int) * (size_t)(kSyncHeader + kSyncArrays * maxblk), s);
    // ---- the partner front (see CholeskyPartner): taken along if every one of its block columns is a resident single column,
    // in the launches of this front's resident single columns
    if (partner && partner->ncols > 0 && partner->ctx && !(kt && kt->times_all())) {
      pb.plan = cholesky_plan(partner->nblk, group, partner->last, true, partner->ncols, partner->tail_rows, partner->tail);
      bool all = (int)pb.plan.resident.size() >= partner->ncols;
      for (int j = 0; all && j < partner->ncols; ++j) all = pb.plan.resident[j] && !pb.plan.paired[j];
      // (from the first resident single column of this front on — its first few block columns are chain-bound too, before the
      // wide ones: the partner's chain pauses while those are factored launch by launch, and goes on in the trailing run)
      int start = 0;
      while (start < ncols && !(plan.resident[start] && !plan.paired[start])) ++start;
      int* sb_sync = all && start < ncols ? partner->ctx->sync_for(partner->nblk) : nullptr;
      if (sb_sync && partner->ctx->xs) {
        pb.on = true; pb.start_at = start; pb.sync = sb_sync; pb.maxblk = partner->ctx->sync_blk; pb.xs = partner->ctx->xs;
        pb.nblk = partner->nblk; pb.ncols = partner->ncols; pb.tail0 = partner->nblk - std::max(1, std::min(partner->tail_rows, partner->nblk - partner->ncols));
        pb.last = partner->last; pb.tail = partner->tail; pb.S = partner->S; pb.ld = partner->ld; pb.Linv = partner->Linv;
        if (!early) (void)hipMemsetAsync(pb.sync, 0, sizeof(int) * (size_t)(kSyncHeader + kSyncArrays * pb.maxblk), s);
        partner->taken = true;
      }
    }
    // ONE event on the caller's stream for all four queues (the server's here, the others below): each record is a packet of
    // its own in that queue, in front of everything that follows
    start_ev = ctx->event(ev++);
    (void)hipEventRecord(start_ev, s);
    // the server: no event in front of it — it is resident as soon as its queue is free and waits for chain_start_kernel, which runs on
    // the caller's stream behind the reset of the counters above (developer variable SK_CHAIN_EARLY_SERVER=0: the event, as until round 5;
    // under stream capture too)
    if (early) {
      static std::atomic<int> epoch{0};
      start_value = 1 + (epoch.fetch_add(1) & 0x3fffffff);
      hipLaunchKernelGGL(chain_start_kernel, dim3(1), dim3(256), 0, s, sync, kSyncHeader + kSyncArrays * maxblk, pb.on ? pb.sync : (int*)nullptr,
                         pb.on ? kSyncHeader + kSyncArrays * pb.maxblk : 0, start_value);
    } else {
      (void)hipStreamWaitEvent(srv, start_ev, 0);
    }
    if (pb.on) {
      ServerPair sp2;
      sp2.f[0] = ServerArgs{S, ld, ranges, Linv, sync, maxblk};
      ChainRanges rb; rb.n = 1; rb.begin[0] = 0; rb.end[0] = pb.ncols;
      sp2.f[1] = ServerArgs{pb.S, pb.ld, rb, pb.Linv, pb.sync, pb.maxblk};
      hipLaunchKernelGGL(potrf_server_pair_kernel, dim3(2), dim3(256), potrf128_lds_bytes(), srv, sp2, info, local, start_value);
    } else {
      hipLaunchKernelGGL(potrf_server_kernel, dim3(1), dim3(256), potrf128_lds_bytes(), srv, S, ld, ranges, Linv, info, sync, maxblk, local, start_value);
    }
  }
  if (la && chain && start_value != 0 && sp != s) {
    // the panel stream starts behind a polling wave as well (chain_start_gate_kernel), not behind a blocked wait; the bulk streams need
    // nothing in front: every kernel they get waits for an event of the panel stream, which lies behind that wave
    hipLaunchKernelGGL(chain_start_gate_kernel, dim3(1), dim3(64), 0, sp, (const int*)(sync + kSyncStart), start_value, info);
  } else if (la) {
    if (!start_ev) { start_ev = ctx->event(ev++); (void)hipEventRecord(start_ev, s); }
    for (hipStream_t q : {sp, ctx->bulk, ctx->bulk_early}) if (q != s) (void)hipStreamWaitEvent(q, start_ev, 0);
  }
  if (!is_resident(0) && ncols > 0) panel(gb[0], gb[1]);
  bool used_bulk_early = false;
  hipEvent_t syrk_done = nullptr;  // syrk(g-1), which writes the tiles next(g) updates
  hipEvent_t syrk_any = nullptr, syrk_done2 = nullptr;  // the completion of the last SYRK launched (whenever it carries an event), and of the one before it
  int seq = 0;                     // SYRK completions announced to the chain so far (chain_marker_kernel)
  int col_seq = 0;                 // ... and first-column workgroups of SYRKs that announce themselves (syrk_trailing_thin_f64_kernel)
  for (int g = 0; g + 1 < ngroups; ++g) {
    const int k0 = gb[g], k1 = gb[g + 1];
    if (k0 >= ncols) break;  // (a partial factorisation: the border's pseudo-groups)
    const int K = (k1 - k0) * 128;
    bool with_partner = false;  // this group's column launch and SYRK carry a block column of the partner front
    PartnerStep pst{};
    const bool resident = is_resident(k0);        // potrf(k0) by the server; TRSM and next(g) by one column launch
    const bool next_resident = is_resident(k1);   // ... and the same for the next group
    // tile columns that next(g) updates: those of the next group — but a resident column's launch applies its panel to
    // block column k1 alone (chain_column_kernel), whatever the width of the group that follows: where a resident run
    // hands back to launch-by-launch groups of two, the SYRK takes everything from k1 + 1.  (Round 1 split at the next
    // group's width there: block column k1 + 1 never received this panel's update — a wrong factor that the
    // full-size property tests could not see; tests/test_gpu_parity.py::test_default_plan_matches_explicit_grouping_at_full_size
    // and test_factorisation_plans_vs_numpy now do.)
    // a resident PAIR (cholesky_plan): one K = 256 SYRK applies both panels to everything from k1 on — there is no next(g);
    // launch by launch (no resident chain on this device) the same launches, the next panel behind the whole SYRK
    const bool pair = plan.paired[k0] == 1 && k1 - k0 == 2;
    const bool after_pair = k0 > 0 && plan.paired[k0 - 1] == 2 && is_resident(k0 - 1);  // block column k0 was last updated by a resident pair's SYRK
    const int na = pair ? 0 : (resident ? 1 : gb[g + 2] - k1);
    // (rows below the run of the group's last column, other than its tail rows, are zero in every column of this group)
    const Rows rn = rows_from(k1, k1 - 1);       // rows that next(g) updates
    const Rows rs = rows_from(k1 + na, k1 - 1);  // rows (and columns) that syrk(g) updates
    double* A22 = S + (long)k1 * 128 * ld + (long)k1 * 128;
    const double* P = S + (long)k1 * 128 * ld + (long)k0 * 128;
    const int Tb = rs.main + rs.extra;
    if (la) sb = Tb >= ctx->early_tiles ? ctx->bulk_early : ctx->bulk;
    if (la && Tb > 0 && sb == ctx->bulk_early) used_bulk_early = true;
    if (sb != sb_prev && syrk_done) (void)hipStreamWaitEvent(sb, syrk_done, 0);  // syrk(g) after syrk(g-1) across the two bulk streams
    sb_prev = sb;
    if (resident && pair) {
      // first column: TRSM + its panel applied to the second column (the server waits for S(k0+1,k0+1)); second column: TRSM.
      // The first column's workgroups wait (resident, 66 KB of LDS each, a few hundred of them) for the first block columns of
      // the SYRK before this pair, which starts when the SYRK before THAT has finished: launched any earlier they would sit on
      // the CUs for the whole length of that SYRK and take its occupancy.  So the launch follows the completion of the SYRK
      // two back (an event that rode on its dispatch).
      if (syrk_done2) (void)hipStreamWaitEvent(sp, syrk_done2, 0);
      hipEvent_t col_done = ctx->event(ev++);
      for (int c = 0; c < 2; ++c) {
        const Rows rc = rows_from(k0 + c + 1, k0 + c);
        const int T = rc.main + rc.extra;
        const int ncrit = (rc.main > 0 || rc.jump == 0) ? 16 : 0;
        hipExtLaunchKernelGGL(chain_column_kernel, dim3(ncrit ? 16 + 4 * T - 4 : 4 * T), dim3(256), 0, sp, nullptr, c == 1 ? col_done : (hipEvent_t) nullptr, 0, S, ld,
                              k0 + c, (const double*)(Linv + (long)(k0 + c) * 128 * 128), 4 * T, 4 * rc.main, 4 * rc.jump, ncrit, ctx->xs, sync, maxblk, seq, col_seq, info,
                              c == 0 ? 1 : 0, c == 0 && after_pair ? 1 : 0, ncrit);  // (the columns of a resident PAIR: the device-scope hand-over)
      }
      (void)hipStreamWaitEvent(sb, col_done, 0);
    } else if (resident) {
      const int T = rn.main + rn.extra;  // rows of X(.,k0) == rows that next(g) updates (K = 128, na == 1)
      // block row k0+1 is active as the start of the main run, or as the last block row itself
      const int ncrit = (rn.main > 0 || rn.jump == 0) ? 16 : 0;
      const int ncand = ncrit && local && !after_pair ? kCritCandidates : ncrit;
      // (a partner rides only with SYRKs in the thin tiling, which all go to ctx->bulk: early_tiles > g_thin_syrk_tiles, or consecutive
      // SYRKs of a front could land on two streams with only their first block column awaited — ADVICE r03)
      with_partner = pb.on && k0 >= pb.start_at && pb.next < pb.ncols && !after_pair && Tb <= g_thin_syrk_tiles && ctx->early_tiles > g_thin_syrk_tiles;
      if (with_partner) pst = partner_step(pb.next);
      hipEvent_t col_done = (Tb > 0 || (with_partner && pst.Tb > 0)) ? ctx->event(ev++) : nullptr;
      if (with_partner) {
        ColumnPair cp;
        cp.f[0] = ColumnArgs{S, ld, k0, Linv + (long)k0 * 128 * 128, 4 * T, 4 * rn.main, 4 * rn.jump, ncrit, ctx->xs, sync, maxblk, seq, col_seq, column_grid(ncrit, T, ncand), ncand};
        cp.f[1] = pst.col;
        hipExtLaunchKernelGGL(chain_column_pair_kernel, dim3(std::max(cp.f[0].grid, cp.f[1].grid), 2), dim3(256), 0, sp, nullptr, col_done, 0, cp, info);
      } else {
        hipExtLaunchKernelGGL(chain_column_kernel, dim3(column_grid(ncrit, T, ncand)), dim3(256), 0, sp, nullptr, col_done, 0, S, ld, k0,
                              (const double*)(Linv + (long)k0 * 128 * 128), 4 * T, 4 * rn.main, 4 * rn.jump, ncrit, ctx->xs, sync, maxblk, seq, col_seq, info, 1, after_pair ? 1 : 0, ncand);
      }
      if (col_done) (void)hipStreamWaitEvent(sb, col_done, 0);
    } else {
      // panel(g) is final: syrk(g) may start (after syrk(g-1))
      if (panel_done) (void)hipStreamWaitEvent(sb, panel_done, 0);
      else order(sp, sb);
      // next(g): active rows x na tiles, above-diagonal tiles skipped; on the panel stream, after syrk(g-1)
      if (la && syrk_done) (void)hipStreamWaitEvent(sp, syrk_done, 0);
      update(sp, "gemm_syrk_next", A22, P, P, K, rn, na, 1);
    }
    // syrk(g): everything right of them, lower triangle over the active rows
    syrk_done = nullptr;
    if (with_partner) {
      // both fronts' thin SYRKs as one launch; each front's first block column counts into its own counter
      const bool by_col = next_resident && g_early_column;
      bool withheld = false;  // (fault injection of the testing build, as in the unpaired branch below)
#ifdef SK_TESTING
      if (next_resident && Tb > 0 && !ctx->dq->tuning) { int want = k1; withheld = want >= 0 && g_test_withhold.compare_exchange_strong(want, -1); }
#endif
      ThinSyrkPair tp;
      tp.f[0] = ThinSyrkArgs{S + (long)(k1 + 1) * 128 * ld + (long)(k1 + 1) * 128, ld, S + (long)(k1 + 1) * 128 * ld + (long)k0 * 128, ld, 128, 4 * Tb, 4 * rs.main, 4 * rs.jump,
                             rs.main, rs.jump, by_col && !withheld ? sync + kSyncSyrkColumn : (int*)nullptr, Tb > 0 ? 2 * Tb * (Tb + 1) : 0};
      tp.f[1] = pst.syrk;
      const int gx = std::max(tp.f[0].grid, tp.f[1].grid);
      hipEvent_t t_start = nullptr, t_stop = nullptr;
      if (kt && gx > 0) kt->pair("gemm_syrk", &t_start, &t_stop);  // (the SYRK timer of the bench line covers both fronts' tiles: so does its flop count)
      hipEvent_t stop_ev = t_stop ? t_stop : ctx->event(ev++);
      if (gx > 0) hipExtLaunchKernelGGL(syrk_trailing_thin_pair_f64_kernel, dim3(gx, 2), dim3(256), 0, sb, t_start, stop_ev, 0, tp);
      if (Tb > 0) {
        if (by_col) col_seq += 4 * Tb;
        else if (next_resident) { ++seq; if (!withheld) hipLaunchKernelGGL(chain_marker_kernel, dim3(1), dim3(1), 0, sb, sync + kSyncSyrkSeq, seq); }
      }
      partner_advance(pst, sb);
      if (gx > 0 && !next_resident) syrk_done = stop_ev;  // (the launch-by-launch panel that follows waits for it)
      syrk_done2 = syrk_any;
      syrk_any = gx > 0 ? stop_ev : nullptr;
    } else if (Tb > 0) {
      double* Cb = S + (long)(k1 + na) * 128 * ld + (long)(k1 + na) * 128;
      const double* Pb = S + (long)(k1 + na) * 128 * ld + (long)k0 * 128;
      // the launch-by-launch next(g+1) waits for this SYRK as an event; so does a SYRK that follows on the other bulk stream
      bool record = la && (!next_resident || (pair && !resident));  // (a pair launch by launch: the next panel waits for the whole SYRK)
      if (la && !record && g + 3 < (int)gb.size()) {
        // (where the NEXT group's SYRK starts: behind its next(g + 1) — one block column after a resident column, none after a pair)
        const int k2 = gb[g + 2], na1 = (plan.paired[k1] == 1 && k2 - k1 == 2) ? 0 : (is_resident(k1) ? 1 : gb[g + 3] - k2);
        const Rows rs1 = rows_from(k2 + na1, k2 - 1);
        record = (rs1.main + rs1.extra >= ctx->early_tiles) != (sb == ctx->bulk_early);
      }
      // ... an event that rides on the SYRK's own dispatch (as do the two of the kernel timer): a separate record is a
      // packet of its own behind the SYRK, in front of the next one
      hipEvent_t t_start = nullptr, t_stop = nullptr;
      if (kt) kt->pair("gemm_syrk", &t_start, &t_stop);
      hipEvent_t stop_ev = t_stop ? t_stop : ((record || (la && chain)) ? ctx->event(ev++) : nullptr);  // (under the chain every SYRK: a pair's column launch is gated on one, below)
      // What the next column launch waits for when it is resident: the FIRST block column of this SYRK, counted by its
      // own workgroups as they finish (the 32-row tiling: 4 Tb of them, enumerated first) — or, with the 128-row tiling,
      // the whole SYRK, announced by a marker kernel behind it.
      // (fault injection for tests: SK_CHAIN_TEST_WITHHOLD_MARKER=<block column> withholds what that column's launch
      // waits for, once per process — the wait times out, and the factorisation must be reported as lost: info = 2)
      bool withheld = false;
#ifdef SK_TESTING
      if (next_resident && !ctx->dq->tuning) { int want = k1; withheld = want >= 0 && g_test_withhold.compare_exchange_strong(want, -1); }
#endif
      const bool pair_chain = pair && resident;  // (always the 32-row tiling: its first block columns count themselves)
      const bool thin = Tb <= g_thin_syrk_tiles || pair_chain, by_column = next_resident && thin && (g_early_column || pair_chain);
      // after a resident pair the next column launch updates the SECOND block column this SYRK touches (its own next(j) applies
      // panel j to block column j + 1): both are counted; and the first diagonal block is what the server waits for
      const int count_cols = pair_chain ? 2 : 1;
      if (thin)
        hipExtLaunchKernelGGL(syrk_trailing_thin_f64_kernel, dim3(2 * Tb * (Tb + 1)), dim3(256), 0, sb, t_start, stop_ev, 0, Cb, ld, Pb, ld, K, 4 * Tb, 4 * rs.main,
                              4 * rs.jump, rs.main, rs.jump, by_column && !withheld ? sync + kSyncSyrkColumn : (int*)nullptr, count_cols,
                              pair_chain && next_resident && !withheld ? sync + kSyncHeader + k1 : (int*)nullptr);
      else
        hipExtLaunchKernelGGL(syrk_trailing_f64_kernel, dim3(Tb * (Tb + 1) / 2), dim3(256), 0, sb, t_start, stop_ev, 0, Cb, ld, Pb, ld, K, 0, rs.main, rs.jump);
      if (by_column) {
        col_seq += 4 * Tb + (count_cols > 1 && Tb > 1 ? 4 * (Tb - 1) : 0);
      } else if (next_resident) {
        ++seq;
        if (!withheld) hipLaunchKernelGGL(chain_marker_kernel, dim3(1), dim3(1), 0, sb, sync + kSyncSyrkSeq, seq);
      }
      if (record) syrk_done = stop_ev;
      syrk_done2 = syrk_any;
      syrk_any = stop_ev;
    } else {
      syrk_done2 = syrk_any;
      syrk_any = nullptr;
    }
    if (next_resident) {
      // hand-over: block column k1 has its last launch-by-launch update; the server takes it from here (a column launch
      // has told it already)
      if (!resident) hipLaunchKernelGGL(chain_marker_kernel, dim3(1), dim3(1), 0, sp, sync + kSyncHeader + k1, 16);
    } else {
      // (after a resident column: hand-back, in stream order behind its column launch; the columns after k1 of a wider
      // group take this panel's update from syrk(g), on the bulk stream)
      const int k2 = gb[g + 2];
      if (k1 < ncols) {
        if ((pair || (resident && k2 - k1 > 1)) && la && syrk_done) (void)hipStreamWaitEvent(sp, syrk_done, 0);
        panel(k1, k2);
      }
    }
  }
  // the partner's block columns that found no column of this front to ride with: launches of their own, on the same streams
  while (pb.on && pb.next < pb.ncols) {
    const PartnerStep st = partner_step(pb.next);
    hipEvent_t col_done = st.Tb > 0 ? ctx->event(ev++) : nullptr;
    ColumnPair cp;
    cp.f[0] = st.col; cp.f[1] = st.col; cp.f[1].grid = 0;
    hipExtLaunchKernelGGL(chain_column_pair_kernel, dim3(st.col.grid, 1), dim3(256), 0, sp, nullptr, col_done, 0, cp, info);
    if (st.Tb > 0) {
      (void)hipStreamWaitEvent(ctx->bulk, col_done, 0);
      ThinSyrkPair tp;
      tp.f[0] = st.syrk; tp.f[1] = st.syrk; tp.f[1].grid = 0;
      hipEvent_t t_start = nullptr, t_stop = nullptr;
      if (kt) kt->pair("gemm_syrk", &t_start, &t_stop);
      hipExtLaunchKernelGGL(syrk_trailing_thin_pair_f64_kernel, dim3(st.syrk.grid, 1), dim3(256), 0, ctx->bulk, t_start, t_stop, 0, tp);
    }
    partner_advance(st, ctx->bulk);
  }
  if (chain && start_value != 0) {
    // the join without events (chain_gate_kernel): a marker at the end of every stream the factorisation used, the server's own exit
    int expected = pb.on ? 2 : 1;
    for (hipStream_t q : {sp, ctx->bulk, used_bulk_early ? ctx->bulk_early : (hipStream_t) nullptr})
      if (q && q != s) { hipLaunchKernelGGL(chain_end_marker_kernel, dim3(1), dim3(64), 0, q, sync + kSyncEnd); ++expected; }
    hipLaunchKernelGGL(chain_gate_kernel, dim3(1), dim3(64), 0, s, (const int*)(sync + kSyncEnd), expected, info);
  } else {
  order(sp, s);
  // (a factorisation of chain-bound block columns never touches the stream of the wide SYRKs: one cross-queue dependency less in
  // front of whatever follows — each is a packet of its own, ~10 us)
  if (la) { order(ctx->bulk, s); if (used_bulk_early) order(ctx->bulk_early, s); }
  if (chain) order(srv, s);
  }
  if (stamps_file) {
    (void)hipStreamSynchronize(s);
    std::vector<long long> st((size_t)1024 * 8);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_chain_stamps), st.size() * sizeof(long long));
    // (two fronts in lock-step: <file>.pair — the root's factorisation, which follows, writes <file>)
    const std::string stamps_name = std::string(stamps_file) + (partner && partner->ncols > 0 ? ".pair" : "");
    if (FILE* f = fopen(stamps_name.c_str(), "w")) {
      for (int j = 0; j < nblk && j < 512; ++j) {
        if (!plan.resident[j]) continue;
        fprintf(f, "%d ", j);
        for (int i = 0; i < 7; ++i) fprintf(f, "%lld ", st[(size_t)j * 8 + i]);
        const Rows rs = rows_from(j + 2, j);
        fprintf(f, "%d\n", rs.main + rs.extra);
      }
      // (the partner front's block columns, two fronts in lock-step: 512 + j)
      for (int j = 0; partner && j < partner->ncols && j < 512; ++j) {
        fprintf(f, "%d ", 512 + j);
        for (int i = 0; i < 7; ++i) fprintf(f, "%lld ", st[(size_t)(512 + j) * 8 + i]);
        fprintf(f, "0\n");
      }
      fclose(f);
    }
  }
}

// y (npad) <- solution of L^T y = z, with z^T = row rhs_row of L (first n entries).  w: scratch (npad).
// With an envelope, block row kb of L is zero left of the first block column c with last[c] >= kb.
bool cholesky_backsolve_resident(int nblk) { return g_bs_resident.load() != 0 && nblk <= kBsMaxBlocks; }
// janitor workgroups of a resident back-substitution with `owners` owner workgroups (bs_resident_kernel): as many as there are owners, up
// to 96 — with the owners they stay inside the chip's 256 CUs (a workgroup of 1024 threads at 256 VGPRs fills one)
static int bs_janitors(int owners) { return std::max(1, std::min(owners, 96)); }
void cholesky_backsolve(double* S, long ld, int n, int npad, int rhs_row, const double* Linv, double* w, double* y,
                        hipStream_t s, KernelTimer* kt, const int* last, int* info, const int* tail, bool zero_after, int resident, bool prefilled) {
  const int nblk = npad / 128;
  if (info && (resident < 0 ? g_bs_resident.load() != 0 : resident != 0) && nblk <= kBsMaxBlocks) {
    BsTop env;
    for (int c = 0; c < nblk; ++c) {
      env.top[c] = (unsigned short)(last ? std::min(std::max(last[c], c), nblk - 1) : nblk - 1);
      env.tail[c] = (unsigned short)(tail ? std::min(std::max(tail[c], 0), nblk - 1) : nblk - 1);
    }
    if (kt) kt->begin("backsolve", s);
    if (!prefilled) (void)hipMemsetAsync(y, 0xff, sizeof(double) * (size_t)npad, s);  // (prefilled: the caller has set the sentinels, off the critical path)
    const char* bs_stamps = dev_knobs().bs_stamps;
    if (bs_stamps) { static bool on = false; if (!on) { const int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bs_stamps_on), &one, sizeof(int)); on = true; } }
    const bool spread = zero_after && dev_knobs().bs_spread != 0;  // (the XCD-spread launch: bs_resident_kernel)
    const int janitors = zero_after ? bs_janitors(nblk) : 0;
#ifdef SK_TESTING
    if (janitors > 0) { int want = 1; if (g_test_janitor_giveup.compare_exchange_strong(want, 0)) { const int one = 1; (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_bs_test_janitor_giveup), &one, sizeof(int), 0, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); } }
#endif
    hipLaunchKernelGGL(bs_resident_kernel, dim3(spread ? 8 * nblk : nblk + janitors), dim3(1024), 0, s, Linv, S, ld, (const double*)(S + (long)rhs_row * ld), n, y, nblk, env, info, nblk,
                       (const double*)nullptr, spread ? -1 : janitors, (const int*)nullptr);
    if (kt) kt->end("backsolve", s);
    if (bs_stamps) {
      (void)hipStreamSynchronize(s);
      std::vector<long long> st((size_t)1024 * 4);
      (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_bs_stamps), st.size() * sizeof(long long));
      if (FILE* f = fopen(bs_stamps, "w")) {
        for (int kb = nblk - 1; kb >= 0; --kb) fprintf(f, "%d %lld %lld %lld\n", kb, st[(size_t)kb * 4], st[(size_t)kb * 4 + 1], st[(size_t)kb * 4 + 2]);
        fclose(f);
      }
    }
    return;
  }
  hipLaunchKernelGGL(copy_row_kernel, dim3((npad + 255) / 256), dim3(256), 0, s, S + (long)rhs_row * ld, w, n, npad);
  if (kt) kt->begin("backsolve", s);
  const std::vector<int> first = cholesky_row_first_cols(nblk, last, tail);
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int ncols = kb * 128, col0 = first[kb] * 128;
    const int grid = ncols > col0 ? (ncols - col0 + 63) / 64 : 1;
    hipLaunchKernelGGL(bs_step_kernel, dim3(grid), dim3(1024), 0, s, Linv + (long)kb * 128 * 128, S + (long)kb * 128 * ld, ld, w, y, kb, col0, ncols);
  }
  if (kt) kt->end("backsolve", s);
}

// ---------------------------------------------------------------------------
// Two-way dissection of a block-banded system ("Dissection").
//
// The reduced camera system of a camera sequence is block-banded, and a banded Cholesky is a serial chain of one
// block column after the other: 122 of them at C = 1723, each 40-60 us of latency however idle the chip is.  The
// cameras are therefore split into a head A, a separator and a tail B such that no point is seen from both A and B;
// A is eliminated front to back and B BACK TO FRONT, concurrently — two chains of half the length — each leaving its
// Schur complement on the separator, which is then factored: the chain is ~(n + separator) / 2 block columns long
// instead of n.  The same arithmetic as the banded factorisation in another elimination order (a "twisted" / "burn at
// both ends" factorisation); no flops are added, B's are the mirror image of what they were.
//
// Storage: three dense matrices ("fronts").  A leaf front holds its interior block columns [0, ncols) followed by a
// BORDER of E block rows: the separator's unknowns and, in the last block, the right-hand-side row, exactly as a whole
// system holds it.  Ordered so that the separator unknowns a leaf's last columns couple with come FIRST in its border
// (A: separator in camera order; B: in reverse camera order), a leaf front is an ordinary envelope matrix — a contiguous
// run of active rows plus the last block row — of which cholesky_factor factors the first ncols block columns
// (cholesky_plan, partial).  The root front is the separator's own system (E blocks, right-hand side in its last row),
// to which the leaves' border blocks are added before it is factored.
// ---------------------------------------------------------------------------
// root (lower triangle, row-major, ld_r) += the lower triangle of a leaf's border x border block, border index i going to
// root index map[i] (< 0: an unused padding row).  map == nullptr: identity.  One thread per element of the border block.
__global__ __launch_bounds__(256) void border_add_kernel(double* __restrict__ root, long ld_r, const double* __restrict__ border, long ld_f, int m,
                                                         const int* __restrict__ map) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)m * m) return;
  const int i = (int)(e / m), j = (int)(e % m);
  if (j > i) return;
  int ri = i, rj = j;
  if (map) { ri = map[i]; rj = map[j]; }
  if (ri < 0 || rj < 0) return;
  const double v = border[(long)i * ld_f + j];
  if (v == 0.0) return;
  if (ri >= rj) root[(long)ri * ld_r + rj] += v;
  else root[(long)rj * ld_r + ri] += v;
}

// w[c] -= sum_r L[border row r][c] * yb[r] for the interior columns c in [col0, ncols) of a leaf front: the border
// unknowns (the separator's solution, in the leaf's border order) are known; this is their part of L^T y = z.
// Lb: first border row of the front.  64 columns per workgroup, 16 row groups.
__global__ __launch_bounds__(1024) void bs_border_kernel(const double* __restrict__ Lb, long ld, int m, const double* __restrict__ yb, double* __restrict__ w,
                                                          int col0, int ncols) {
  __shared__ double red[16 * 64];
  const int t = threadIdx.x, cl = t & 63, rg = t >> 6;
  const int col = col0 + blockIdx.x * 64 + cl;
  double acc = 0.0;
  if (col < ncols)
    for (int r = rg; r < m; r += 16) acc += Lb[(long)r * ld + col] * yb[r];
  red[rg * 64 + cl] = acc;
  __syncthreads();
  if (t < 64 && col < ncols) {
    double u = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) u += red[g * 64 + t];
    w[col] -= u;
  }
}

// dst[i] = map[i] >= 0 ? src[map[i]] : 0   (a leaf's border unknowns from the root's solution)
__global__ void gather_map_kernel(const double* __restrict__ src, const int* __restrict__ map, double* __restrict__ dst, int m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) dst[i] = map ? (map[i] >= 0 ? src[map[i]] : 0.0) : src[i];
}

// root += the head's border (in the root's order), then += the tail's (through the map, which is its own inverse: the root entry (i, j)
// takes the tail's entry (map[i], map[j])) — the same two additions in the same order as two border_add_kernel launches, in one
__global__ __launch_bounds__(256) void border_add2_kernel(double* __restrict__ root, long ld_r, const double* __restrict__ a, long ld_a, const double* __restrict__ b, long ld_b,
                                                          int m, const int* __restrict__ map) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)m * m) return;
  const int i = (int)(e / m), j = (int)(e % m);
  if (j > i) return;
  double v = root[(long)i * ld_r + j];
  bool touched = false;
  const double va = a[(long)i * ld_a + j];
  if (va != 0.0) { v += va; touched = true; }
  const int bi = map[i], bj = map[j];
  if (bi >= 0 && bj >= 0) {
    const double vb = bi >= bj ? b[(long)bi * ld_b + bj] : b[(long)bj * ld_b + bi];
    if (vb != 0.0) { v += vb; touched = true; }
  }
  if (touched) root[(long)i * ld_r + j] = v;
}
void cholesky_border_add(double* root, long ld_r, const double* front, long ld_f, int ncols, int border_blocks, const int* map, hipStream_t s) {
  const int m = border_blocks * 128;
  const long total = (long)m * m;
  hipLaunchKernelGGL(border_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, root, ld_r, front + (long)ncols * 128 * ld_f + (long)ncols * 128, ld_f, m, map);
}

// Interior part of L^T y = z for a leaf front whose border unknowns are known.  yb: border unknowns (border_blocks * 128
// values, zero where the border has padding or its right-hand-side row).  w: scratch (ncols * 128); y: interior solution.
void cholesky_backsolve_front(double* S, long ld, int nblk, int ncols, int rhs_row, const double* Linv, const double* yb, double* w, double* y,
                              hipStream_t s, const int* last, bool spike, int tail_rows, int* info, bool zero_after, const int* tail, const int* yb_map, int resident,
                              bool prefilled) {
  const int ni = ncols * 128, m = (nblk - ncols) * 128;
  if (ncols <= 0) return;
  const bool res = info && (resident < 0 ? g_bs_resident.load() != 0 : resident != 0) && nblk <= kBsMaxBlocks;
  if (yb_map && !res) {  // (a map is the resident launch's: the launch-by-launch form below reads yb in the border's own order — a caller's bug, loud)
    std::fprintf(stderr, "[skeres_amd] internal: cholesky_backsolve_front got a border map without the resident launch\n");
    if (info) (void)hipMemsetAsync(info, 0x7f, sizeof(int), s);
    return;
  }
  if (res) {
    // one resident launch (bs_resident_kernel): an owner per interior block column, the border's unknowns read from yb.  Not the
    // same grouping of the border's terms as bs_border_kernel below (block row by block row there, sixteen interleaved row
    // groups here): equal to rounding, not to the bit — a front's interior solution belongs to one rank.
    if (tail_rows < 1 || tail_rows > nblk - ncols) tail_rows = 1;
    BsTop env;
    for (int c = 0; c < nblk; ++c) {
      env.top[c] = (unsigned short)(last ? std::min(std::max(last[c], c), nblk - 1) : nblk - 1);
      env.tail[c] = (unsigned short)(tail ? std::min(std::max(tail[c], 0), nblk - 1) : nblk - tail_rows);
    }
    if (!prefilled) (void)hipMemsetAsync(y, 0xff, sizeof(double) * (size_t)ni, s);
    const bool spread = zero_after && dev_knobs().bs_spread != 0;
    const int janitors = zero_after ? bs_janitors(ncols) : 0;
    hipLaunchKernelGGL(bs_resident_kernel, dim3(spread ? 8 * ncols : ncols + janitors), dim3(1024), 0, s, Linv, S, ld, (const double*)(S + (long)rhs_row * ld), ni, y, nblk, env, info, ncols, yb,
                       spread ? -1 : janitors, yb_map);
    return;
  }
  // (yb_map is the resident launch's: a caller that passes one has checked cholesky_backsolve_resident and info)
  hipLaunchKernelGGL(copy_row_kernel, dim3((ni + 255) / 256), dim3(256), 0, s, S + (long)rhs_row * ld, w, ni, ni);
  // block rows' first non-zero block columns (the border rows: from the first interior column that reaches the border)
  std::vector<int> first(nblk, 0);
  int c0 = 0;
  if (last)
    for (int kb = 0; kb < nblk - 1; ++kb) {
      while (c0 < kb && last[c0] < kb) ++c0;
      first[kb] = c0;
    }
  // (spike: border rows that couple with the FIRST interior columns — the left separator of a segment between two — and so
  // reach every column; otherwise the border starts where the first interior column reaches it.  A tail profile: its rows may share
  // the LAST block row with the right-hand side, which every column has — from the first column on, then; what a column does not
  // couple with is zeros)
  const int bfirst = first[ncols < nblk - 1 ? ncols : nblk - 1];
  const int bcol0 = (spike || tail) ? 0 : std::min(bfirst, ncols) * 128;
  if (ni > bcol0)
    hipLaunchKernelGGL(bs_border_kernel, dim3((ni - bcol0 + 63) / 64), dim3(1024), 0, s, S + (long)ncols * 128 * ld, ld, m, yb, w, bcol0, ni);
  for (int kb = ncols - 1; kb >= 0; --kb) {
    const int nc = kb * 128, col0 = first[kb] * 128;
    const int grid = nc > col0 ? (nc - col0 + 63) / 64 : 1;
    hipLaunchKernelGGL(bs_step_kernel, dim3(grid), dim3(1024), 0, s, Linv + (long)kb * 128 * 128, S + (long)kb * 128 * ld, ld, w, y, kb, col0, nc);
  }
}

static void fork_join_events(CholeskyContext* c) {
  if (!c->fork_ev) (void)hipEventCreateWithFlags(&c->fork_ev, g_event_flags);
  if (!c->join_ev) (void)hipEventCreateWithFlags(&c->join_ev, g_event_flags);
}

void cholesky_dissected_factor(const DissectedSystem& d, int* info, int group, hipStream_t s, CholeskyContext* ctxA, CholeskyContext* ctxB,
                               KernelTimer* kt, KernelTimer* ktB, bool allow_chain) {
  // Under the resident chain the tail front rides in the launches of the head's chain-bound block columns (CholeskyPartner) instead
  // of having queues of its own: the only form that pays on one device (the side-by-side form below
  // was slower on every cut measured — profiles/r02_dissection_*.txt, r03_dissection_probe.txt, r03_lockstep_cut_sweep.txt)
  if (d.A.ncols > 0 && d.B.ncols > 0 && ctxA && ctxB && allow_chain) {
    CholeskyPartner pt{d.B.S, d.B.ld, d.B.nblk, d.B.ncols, 1, d.B.last, d.B.Linv, ctxB, d.B.tail};
    cholesky_factor(d.A.S, d.A.ld, d.A.nblk * 128, d.A.Linv, info, group, s, ctxA, kt, d.A.last, allow_chain, d.A.ncols, 1, &pt, d.A.tail);
    if (!pt.taken)  // (not every block column of the tail is a resident single column: factored afterwards, on the primary context)
      cholesky_factor(d.B.S, d.B.ld, d.B.nblk * 128, d.B.Linv, info, group, s, ctxA, kt, d.B.last, allow_chain, d.B.ncols, 1, nullptr, d.B.tail);
    cholesky_border_add(d.R.S, d.R.ld, d.A.S, d.A.ld, d.A.ncols, d.border_blocks, nullptr, s);
    cholesky_border_add(d.R.S, d.R.ld, d.B.S, d.B.ld, d.B.ncols, d.border_blocks, d.mapB, s);
    cholesky_factor(d.R.S, d.R.ld, d.R.nblk * 128, d.R.Linv, info, group, s, ctxA, kt, d.R.last, allow_chain);
    return;
  }
  // Without the resident chain (a device without CU-masked streams, a forced dissection with an explicit grouping): the tail
  // side by side with the head on a second set of queues, its launches enqueued by a thread of its own — or, without such
  // queues, one front after the other.
  const bool side = d.B.ncols > 0 && ctxB && ctxB->fork;
  if (d.B.ncols > 0 && !side) {
    cholesky_factor(d.B.S, d.B.ld, d.B.nblk * 128, d.B.Linv, info, group, s, ctxA, kt, d.B.last, allow_chain && ctxB && ctxB->server, d.B.ncols, 1, nullptr, d.B.tail);
  } else if (side) {
    fork_join_events(ctxB);
    hipStream_t sB = ctxB->fork;
    (void)hipEventRecord(ctxB->fork_ev, s);
    (void)hipStreamWaitEvent(sB, ctxB->fork_ev, 0);
    // the tail's launches (and the event that says they are all enqueued behind `fork`) from ctxB's own thread: the head's are
    // enqueued by this one meanwhile
    const bool chainB = allow_chain && ctxB->server != nullptr;  // the tail under a resident chain of its own
    if (!ctxB->runner) ctxB->runner.reset(new AsyncRunner(ctxB->device));
    ctxB->runner->run([&d, info, group, ctxB, ktB, sB, chainB] {
      cholesky_factor(d.B.S, d.B.ld, d.B.nblk * 128, d.B.Linv, info, group, sB, ctxB, ktB, d.B.last, chainB, d.B.ncols, 1, nullptr, d.B.tail);
      (void)hipEventRecord(ctxB->join_ev, sB);
    });
  }
  if (d.A.ncols > 0) cholesky_factor(d.A.S, d.A.ld, d.A.nblk * 128, d.A.Linv, info, group, s, ctxA, kt, d.A.last, allow_chain, d.A.ncols, 1, nullptr, d.A.tail);
  if (side) {
    ctxB->runner->wait();  // (the join event has been recorded)
    (void)hipStreamWaitEvent(s, ctxB->join_ev, 0);
  }
  if (d.A.ncols > 0 && d.B.ncols > 0 && d.mapB && d.mapB_involution) {
    const int m = d.border_blocks * 128;
    hipLaunchKernelGGL(border_add2_kernel, dim3((unsigned)(((long)m * m + 255) / 256)), dim3(256), 0, s, d.R.S, d.R.ld,
                       (const double*)(d.A.S + (long)d.A.ncols * 128 * d.A.ld + (long)d.A.ncols * 128), d.A.ld,
                       (const double*)(d.B.S + (long)d.B.ncols * 128 * d.B.ld + (long)d.B.ncols * 128), d.B.ld, m, d.mapB);
  } else {
    if (d.A.ncols > 0) cholesky_border_add(d.R.S, d.R.ld, d.A.S, d.A.ld, d.A.ncols, d.border_blocks, nullptr, s);
    if (d.B.ncols > 0) cholesky_border_add(d.R.S, d.R.ld, d.B.S, d.B.ld, d.B.ncols, d.border_blocks, d.mapB, s);
  }
  cholesky_factor(d.R.S, d.R.ld, d.R.nblk * 128, d.R.Linv, info, group, s, ctxA, kt, d.R.last, allow_chain);
}

void cholesky_dissected_backsolve(const DissectedSystem& d, int n_root, double* wR, double* yR, double* wA, double* yA, double* wB, double* yB, double* ybB,
                                  hipStream_t s, CholeskyContext* ctxB, KernelTimer* kt, int* info, bool zero_after, int resident, bool prefilled) {
  if (resident < 0) resident = g_bs_resident.load() != 0 ? 1 : 0;  // (once for the three fronts)
  cholesky_backsolve(d.R.S, d.R.ld, n_root, d.R.nblk * 128, d.R.rhs_row, d.R.Linv, wR, yR, s, nullptr, d.R.last, info, nullptr, zero_after, resident, prefilled);
  if (kt) kt->begin("backsolve", s);
  const int m = d.border_blocks * 128;
  // both leaf fronts under the resident launch: ONE launch for the two (bs_resident_pair_kernel) — no fork, no join
  const bool paired = dev_knobs().bs_pair != 0 && d.A.ncols > 0 && d.B.ncols > 0 && info && resident != 0 && d.A.nblk <= kBsPairMaxBlocks && d.B.nblk <= kBsPairMaxBlocks;
  const bool side = !paired && d.B.ncols > 0 && d.A.ncols > 0 && ctxB && ctxB->fork;
  hipStream_t sB = s;
  if (paired) {
    BsPair p;
    const bool spread = zero_after && dev_knobs().bs_spread != 0;  // (the XCD-spread launch: bs_resident_pair_kernel)
    const FrontView* fv[2] = {&d.A, &d.B};
    double* ys[2] = {yA, yB};
    int grid = 0, jan = 0;
    for (int f = 0; f < 2; ++f) {
      const FrontView& F = *fv[f];
      for (int c = 0; c < F.nblk; ++c) {
        p.env[f].top[c] = (unsigned short)(F.last ? std::min(std::max(F.last[c], c), F.nblk - 1) : F.nblk - 1);
        p.env[f].tail[c] = (unsigned short)(F.tail ? std::min(std::max(F.tail[c], 0), F.nblk - 1) : F.nblk - 1);
      }
      const int janitors = spread ? 3 * F.ncols : (zero_after ? bs_janitors(F.ncols) : 0);
      // (the tail front reads the root's solution through the map, the head as it stands: cholesky_backsolve_front below)
      p.f[f] = BsArgs{F.Linv, F.S, F.ld, (const double*)(F.S + (long)F.rhs_row * F.ld), F.ncols * 128, ys[f], F.nblk, info, F.ncols, yR, janitors, f == 1 ? d.mapB : nullptr,
                      spread ? 3 : 0, spread ? 1 : 0};
      grid += F.ncols; jan += janitors;
    }
    if (spread) { grid = 8 * std::max(d.A.ncols, d.B.ncols); jan = 0; }
    // (one fill for the two solutions where they lie one behind the other)
    const size_t nA = (size_t)d.A.ncols * 128, nB = (size_t)d.B.ncols * 128;
    if (prefilled) {}
    else if (yB >= yA + nA && (size_t)(yB - yA) <= nA + 4096) (void)hipMemsetAsync(yA, 0xff, sizeof(double) * ((size_t)(yB - yA) + nB), s);
    else { (void)hipMemsetAsync(yA, 0xff, sizeof(double) * nA, s); (void)hipMemsetAsync(yB, 0xff, sizeof(double) * nB, s); }
    hipLaunchKernelGGL(bs_resident_pair_kernel, dim3(grid + jan), dim3(1024), 0, s, p);
  } else {
  if (d.B.ncols > 0) {
    if (side) {
      fork_join_events(ctxB);
      (void)hipEventRecord(ctxB->fork_ev, s);
      (void)hipStreamWaitEvent(ctxB->fork, ctxB->fork_ev, 0);
      sB = ctxB->fork;
    }
    // (the resident launch reads the root's solution through the map itself: no gather launch in front of the tail's chain of hops)
    const bool mapped = info && resident != 0 && d.B.nblk <= kBsMaxBlocks;
    if (!mapped) cholesky_gather_map(yR, d.mapB, ybB, m, sB);
    cholesky_backsolve_front(d.B.S, d.B.ld, d.B.nblk, d.B.ncols, d.B.rhs_row, d.B.Linv, mapped ? yR : ybB, wB, yB, sB, d.B.last, false, 1, info, zero_after, d.B.tail,
                             mapped ? d.mapB : nullptr, resident, prefilled);
  }
  }
  // (yR is zero in the root's padding rows and in its right-hand-side row: it serves as A's border unknowns as it stands)
  if (!paired && d.A.ncols > 0) cholesky_backsolve_front(d.A.S, d.A.ld, d.A.nblk, d.A.ncols, d.A.rhs_row, d.A.Linv, yR, wA, yA, s, d.A.last, false, 1, info, zero_after, d.A.tail, nullptr, resident, prefilled);
  if (side) {
    (void)hipEventRecord(ctxB->join_ev, ctxB->fork);
    (void)hipStreamWaitEvent(s, ctxB->join_ev, 0);
  }
  if (kt) kt->end("backsolve", s);
}

void cholesky_gather_map(const double* src, const int* map, double* dst, int m, hipStream_t s) {
  hipLaunchKernelGGL(gather_map_kernel, dim3((m + 255) / 256), dim3(256), 0, s, src, map, dst, m);
}

// H (tiles x tiles blocks of 128, lower) = A A^T, A row-major (tiles*128) x K, K a multiple of 16.
// K = nslabs * Kc; slabs: nslabs x (tiles*128 x ldh) scratch (may be null when nslabs == 1: H is written directly).
void launch_syrk_gram(double* H, long ldh, const double* A, long lda, int Kc, int nslabs, double* slabs, int tiles, hipStream_t s,
                      KernelTimer* kt) {
  if (tiles <= 0) return;
  const size_t slab_stride = (size_t)tiles * 128 * ldh;
  if (kt) kt->begin("syrk_gram", s);
  hipLaunchKernelGGL(syrk_gram_f64_kernel, dim3(tiles * (tiles + 1) / 2, nslabs), dim3(256), 0, s, nslabs > 1 ? slabs : H, ldh, slab_stride, A, lda, Kc);
  if (kt) kt->end("syrk_gram", s);
  if (nslabs > 1) hipLaunchKernelGGL(gram_reduce_kernel, dim3(tiles * (tiles + 1) / 2), dim3(256), 0, s, H, slabs, slab_stride, ldh, nslabs);
}

// Algorithmic flops of the dominant kernel's launches (part (b) of each trailing SYRK:
// lower-triangular 128x128 tiles incl. the diagonal tiles, 2*128*128*K each).
double cholesky_syrk_flops(int npad, int group, const int* last, bool chain, double* c_tiles, int ncols, int tail_rows, const int* tail) {
  const int nblk = npad / 128;
  if (ncols < 0 || ncols > nblk) ncols = nblk;
  if (tail_rows < 1 || ncols > nblk - tail_rows) tail_rows = 1;
  const CholeskyPlan plan = cholesky_plan(nblk, group, last, chain, ncols, tail_rows, tail);
  const std::vector<int>& gb = plan.bounds;
  double f = 0.0, tiles = 0.0;
  for (size_t g = 0; g + 2 < gb.size(); ++g) {
    if (gb[g] >= ncols) break;
    const int k0 = gb[g], k1 = gb[g + 1];
    const int na = (plan.paired[k0] == 1 && k1 - k0 == 2) ? 0 : (plan.resident[k0] ? 1 : gb[g + 2] - k1);  // as cholesky_factor splits next(g) / syrk(g)
    const int Lg = last ? (last[k1 - 1] < nblk - 1 ? last[k1 - 1] : nblk - 1) : nblk - 1;
    const int first_row = k1 + na;
    const int main_rows = Lg >= first_row ? Lg - first_row + 1 : 0;
    const int tail0 = tail ? tail[k1 - 1] : nblk - tail_rows;  // (of the group's last column, as cholesky_factor takes them)
    const int Tb = main_rows + std::max(0, nblk - std::max(tail0, first_row + main_rows));
    f += 0.5 * Tb * (Tb + 1.0) * 2.0 * 128.0 * 128.0 * (double)((k1 - k0) * 128);
    tiles += 0.5 * Tb * (Tb + 1.0);
  }
  if (c_tiles) *c_tiles = tiles;  // 128 x 128 tiles of C read and written once per launch, summed over the launches
  return f;
}

// Algorithmic flops of factoring the blocks inside the envelope (last == nullptr: every block): per block column with
// h active block rows below it, 128^3 (1/3 + h + h^2) — diagonal factorisation, triangular solve of h blocks, symmetric
// update of h (h + 1) / 2 blocks with its diagonal blocks counted once.  Sums to n^3 / 3 for a full matrix.
double cholesky_plan_flops(int nblk, const int* last, int ncols, int tail_rows, const int* tail) {
  double f = 0.0;
  if (ncols < 0 || ncols > nblk) ncols = nblk;
  if (tail_rows < 1 || ncols > nblk - tail_rows) tail_rows = 1;
  for (int c = 0; c < ncols; ++c) {
    const int tail0 = tail ? tail[c] : nblk - tail_rows;
    const int lm = last ? (last[c] < nblk - 1 ? last[c] : nblk - 1) : nblk - 1;
    const int main_rows = lm >= c + 1 ? lm - c : 0;
    const double h = main_rows + std::max(0, nblk - std::max(tail0, c + 1 + main_rows));
    f += 128.0 * 128.0 * 128.0 * (1.0 / 3.0 + h + h * h);
  }
  return f;
}

std::vector<int> root_envelope(const std::vector<int>& sep_off, int members_n, std::vector<int>* tail_out) {
  const int nsep = (int)sep_off.size() - 1;
  if (nsep <= 1) return {};
  const int total = sep_off[nsep] + (members_n > 0 ? members_n : 0), nblk = (total + 1 + 127) / 128;
  std::vector<int> first_col(nblk);
  for (int i = 0; i < nblk; ++i) first_col[i] = i;
  for (int k = 0; k < nsep; ++k) {
    const int col = sep_off[k > 0 ? k - 1 : 0] / 128;
    for (int r = sep_off[k] / 128; r <= (sep_off[k + 1] - 1) / 128 && r < nblk; ++r) first_col[r] = std::min(first_col[r], col);
  }
  if (members_n > 0 && tail_out) {
    const int border_begin = sep_off[nsep] / 128;
    for (int r = border_begin; r < nblk; ++r) first_col[r] = 0;
    std::vector<int> last;
    cholesky_envelope_bordered(first_col, border_begin, &last, tail_out);
    return last;
  }
  return cholesky_envelope_last(first_col);
}

// Envelope from the block rows' first non-zero block columns (first_col[i] <= i for i < nblk-1; the entry of the
// last block row is ignored: that row is always active): last[c] = max{ i <= nblk-2 : first_col[i] <= c }.
std::vector<int> cholesky_envelope_last(const std::vector<int>& first_col, int tail_rows) {
  const int nblk = (int)first_col.size();
  if (tail_rows < 1) tail_rows = 1;
  std::vector<int> last(nblk);
  for (int c = 0; c < nblk; ++c) last[c] = c < nblk - 1 ? c : nblk - 1;
  for (int i = 0; i + tail_rows < nblk; ++i) { const int c = first_col[i] < i ? first_col[i] : i; if (c >= 0 && last[c] < i) last[c] = i; }  // (the tail rows are active in every column anyway)
  for (int c = 1; c < nblk; ++c) if (last[c] < last[c - 1]) last[c] = last[c - 1];
  if (nblk >= 2 && last[nblk - 2] > nblk - 2) last[nblk - 2] = nblk - 2;
  return last;
}


void cholesky_envelope_bordered(const std::vector<int>& first_col, int border_begin, std::vector<int>* last_out, std::vector<int>* tail_out) {
  const int nblk = (int)first_col.size();
  const int bb = std::max(0, std::min(border_begin, nblk - 1));
  std::vector<int>& last = *last_out;
  std::vector<int>& tail = *tail_out;
  last.assign(nblk, 0); tail.assign(nblk, nblk - 1);
  // the band: rows before the border
  for (int c = 0; c < nblk; ++c) last[c] = c < bb ? c : nblk - 1;
  for (int i = 0; i < bb; ++i) { const int c = first_col[i] < i ? first_col[i] : i; if (c >= 0 && last[c] < i) last[c] = i; }
  for (int c = 1; c < bb; ++c) if (last[c] < last[c - 1]) last[c] = last[c - 1];
  // the border: row i is active from reach[i] on — its own first column, or that of any border row before it (a column's tail rows
  // are the LAST rows of the matrix: once row i is in, so is everything behind it); the right-hand-side row from column 0
  std::vector<int> reach(nblk, 0);
  int r = nblk;
  for (int i = bb; i < nblk - 1; ++i) { r = std::min(r, std::max(0, std::min(first_col[i], i))); reach[i] = r; }
  reach[nblk - 1] = 0;
  for (int c = 0; c < nblk; ++c) {  // first active border row of column c: reach is non-increasing in i, so the rows active in column c are a suffix
    int t = nblk - 1;
    while (t - 1 >= bb && reach[t - 1] <= c) --t;
    tail[c] = t;
  }
}

std::vector<int> cholesky_row_first_cols(int nblk, const int* last, const int* tail, int tail_rows) {
  std::vector<int> first(nblk, 0);
  if (!last) return first;
  int c0 = 0;
  for (int kb = 0; kb < nblk; ++kb) {  // last is non-decreasing: one sweep
    while (c0 < kb && last[c0] < kb) ++c0;
    first[kb] = c0;
  }
  if (tail) {
    for (int kb = 0; kb < nblk; ++kb) {  // ... or earlier, as a tail row (tail is non-increasing)
      int c = 0;
      while (c < first[kb] && tail[c] > kb) ++c;
      first[kb] = c;
    }
  } else {
    if (tail_rows < 1) tail_rows = 1;
    for (int kb = std::max(0, nblk - tail_rows); kb < nblk; ++kb) first[kb] = 0;
  }
  return first;
}

}  // namespace sk
