// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.hpp header).
//
// Dense Cholesky A = L L^T on a row-major, lower-stored matrix, plus the
// triangular solves.  Stands in for the Eigen LLT that native Ceres [ext] uses
// for DENSE_NORMAL_CHOLESKY and for the reduced camera system of DENSE_SCHUR.
// Blocked right-looking, OpenMP over trailing tiles, AVX2 register-tiled
// micro-kernel: it doubles as the timed CPU baseline, so it is written to be
// a fair multi-core CPU implementation rather than a naive triple loop.
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace oracle {

typedef double v4d __attribute__((vector_size(32), aligned(8)));

static inline v4d loadu(const double* p) { v4d v; std::memcpy(&v, p, 32); return v; }
static inline double hsum(v4d v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// C[i][j] -= sum_t A[i][t] * B[j][t]   (i < mi, j < nj, t < k); if lower, only j <= i + joff
static void gemm_nt_sub(double* C, int ldc, const double* A, int lda, const double* B, int ldb,
                        int mi, int nj, int k, bool diag_tile) {
  const int k4 = k & ~3;
  for (int i0 = 0; i0 < mi; i0 += 3) {
    const int ih = std::min(3, mi - i0);
    const int jmax = diag_tile ? std::min(nj, i0 + ih) : nj;
    for (int j0 = 0; j0 < jmax; j0 += 4) {
      const int jh = std::min(4, jmax - j0);
      if (ih == 3 && jh == 4) {
        v4d acc[3][4];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) acc[r][c] = (v4d){0, 0, 0, 0};
        const double* a0 = A + (size_t)(i0 + 0) * lda; const double* a1 = A + (size_t)(i0 + 1) * lda; const double* a2 = A + (size_t)(i0 + 2) * lda;
        const double* b0 = B + (size_t)(j0 + 0) * ldb; const double* b1 = B + (size_t)(j0 + 1) * ldb;
        const double* b2 = B + (size_t)(j0 + 2) * ldb; const double* b3 = B + (size_t)(j0 + 3) * ldb;
        for (int t = 0; t < k4; t += 4) {
          const v4d x0 = loadu(a0 + t), x1 = loadu(a1 + t), x2 = loadu(a2 + t);
          v4d y = loadu(b0 + t); acc[0][0] += x0 * y; acc[1][0] += x1 * y; acc[2][0] += x2 * y;
          y = loadu(b1 + t); acc[0][1] += x0 * y; acc[1][1] += x1 * y; acc[2][1] += x2 * y;
          y = loadu(b2 + t); acc[0][2] += x0 * y; acc[1][2] += x1 * y; acc[2][2] += x2 * y;
          y = loadu(b3 + t); acc[0][3] += x0 * y; acc[1][3] += x1 * y; acc[2][3] += x2 * y;
        }
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 4; ++c) {
            double s = hsum(acc[r][c]);
            for (int t = k4; t < k; ++t) s += A[(size_t)(i0 + r) * lda + t] * B[(size_t)(j0 + c) * ldb + t];
            if (!diag_tile || j0 + c <= i0 + r) C[(size_t)(i0 + r) * ldc + j0 + c] -= s;
          }
      } else {
        for (int r = 0; r < ih; ++r)
          for (int c = 0; c < jh; ++c) {
            if (diag_tile && j0 + c > i0 + r) continue;
            double s = 0.0;
            for (int t = 0; t < k; ++t) s += A[(size_t)(i0 + r) * lda + t] * B[(size_t)(j0 + c) * ldb + t];
            C[(size_t)(i0 + r) * ldc + j0 + c] -= s;
          }
      }
    }
  }
}

static int potrf_unblocked(double* A, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * ld + j];
    for (int t = 0; t < j; ++t) d -= A[(size_t)j * ld + t] * A[(size_t)j * ld + t];
    if (!(d > 0.0)) return j + 1;
    d = std::sqrt(d); A[(size_t)j * ld + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * ld + j];
      for (int t = 0; t < j; ++t) s -= A[(size_t)i * ld + t] * A[(size_t)j * ld + t];
      A[(size_t)i * ld + j] = s / d;
    }
  }
  return 0;
}

// In place; only the lower triangle (incl. diagonal) is referenced / written.
// last_row (optional, n entries, non-decreasing, last_row[j] >= j): the column envelope of the matrix — rows below
// last_row[j] are structurally zero in column j, and a Cholesky factor keeps the envelope of the matrix it factors.
// The panel and trailing updates then stop at the panel's last envelope row.  What is left out is 0 - 0 * x in the
// full algorithm, so the factor is bit-identical to last_row == nullptr (tests/test_oracle_kat.py); this is the CPU
// counterpart of the block envelope the GPU factorisation uses, so that the timed CPU baseline does the same work.
int cholesky_lower_inplace(double* A, int n, int ld, int num_threads, const int* last_row) {
  const int NB = 96, TB = 192;
  if (num_threads <= 0) num_threads = 1;
  for (int k = 0; k < n; k += NB) {
    const int nb = std::min(NB, n - k);
    double* Akk = A + (size_t)k * ld + k;
    const int info = potrf_unblocked(Akk, nb, ld);
    if (info) return k + info;
    const int rem = (last_row ? std::min(n, last_row[k + nb - 1] + 1) : n) - k - nb;
    if (rem <= 0) continue;
    double* A21 = A + (size_t)(k + nb) * ld + k;
    // TRSM: X L11^T = A21, row by row
#pragma omp parallel for num_threads(num_threads) schedule(static)
    for (int i = 0; i < rem; ++i) {
      double* x = A21 + (size_t)i * ld;
      for (int j = 0; j < nb; ++j) {
        double s = x[j]; const double* l = Akk + (size_t)j * ld;
        for (int t = 0; t < j; ++t) s -= x[t] * l[t];
        x[j] = s / l[j];
      }
    }
    // SYRK on the trailing lower triangle, tile by tile
    const int nt = (rem + TB - 1) / TB;
    const int ntiles = nt * (nt + 1) / 2;
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; ++tile) {
      // largest-first: map tile -> (bi, bj), bj <= bi
      int bi = (int)((std::sqrt(8.0 * tile + 1.0) - 1.0) / 2.0);
      while ((bi + 1) * (bi + 2) / 2 <= tile) ++bi;
      while (bi * (bi + 1) / 2 > tile) --bi;
      const int bj = tile - bi * (bi + 1) / 2;
      const int i0 = bi * TB, j0 = bj * TB;
      const int mi = std::min(TB, rem - i0), nj = std::min(TB, rem - j0);
      gemm_nt_sub(A + (size_t)(k + nb + i0) * ld + (k + nb + j0), ld, A21 + (size_t)i0 * ld, ld,
                  A21 + (size_t)j0 * ld, ld, mi, nj, nb, bi == bj);
    }
  }
  return 0;
}

// Solve L L^T y = b in place (L row-major lower).  last_row as above: row i of L starts at the first column j with
// last_row[j] >= i (the entries left of it are structural zeros; skipping them changes no sum).
void cholesky_solve_lower(const double* L, int n, int ld, double* b, const int* last_row) {
  std::vector<int> first(n, 0);
  if (last_row) { int c = 0; for (int i = 0; i < n; ++i) { while (c < i && last_row[c] < i) ++c; first[i] = c; } }
  for (int i = 0; i < n; ++i) {
    double s = b[i]; const double* l = L + (size_t)i * ld;
    for (int t = first[i]; t < i; ++t) s = __builtin_fma(-l[t], b[t], s);  // explicit: one rounding per term wherever the loop starts
    b[i] = s / l[i];
  }
  // back substitution with L^T: column-oriented sweep keeps row-major access
  for (int i = n - 1; i >= 0; --i) {
    const double* l = L + (size_t)i * ld;
    const double yi = b[i] / l[i];
    b[i] = yi;
    for (int t = first[i]; t < i; ++t) b[t] = __builtin_fma(-l[t], yi, b[t]);
  }
}

}  // namespace oracle
