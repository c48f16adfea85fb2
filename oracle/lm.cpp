// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.hpp header).
//
// CPU restatement of the solve the reference delegates to native Ceres:
// `ceres.solve(options, problem, summary)` at
//   examples/.../SimpleBundleAdjuster.scala:147-152  (DENSE_SCHUR)
//   examples/.../CurveFitting.scala:119-127          (DENSE_QR, 25 iterations)
//   examples/.../Powell.scala:76-87                  (DENSE_QR, 100 iterations)
//
// *** PARITY UNPINNED for the LM trajectory. ***  ceres-solver is an
// un-vendored, un-pinned dependency of the reference (configuration.sh:8-11;
// version window 1.12 <= v < 2.0).  It is absent from /root/reference, there
// is no test in the reference that calls solve(), and no JVM/Ceres exists in
// this pipeline to generate vectors.  The algorithm below restates the
// PUBLISHED Ceres trust-region Levenberg-Marquardt algorithm (Solver::Options
// documentation + trust_region_minimizer / levenberg_marquardt_strategy /
// schur_eliminator as published) with its documented default constants.  What
// IS pinned: the per-residual-block evaluate step (functors.hpp) by the
// reference's own KATs, and the converged optimum by SciPy cross-checks
// (tests/golden/).
//
// Linear algebra conventions (published Ceres):
//   cost = 1/2 sum r^2 ; Jacobi scaling s_j = 1/(1+||J_j||) fixed at iteration 0
//   D = sqrt(clamp(diag(J^T J), min_lm_diagonal, max_lm_diagonal) / radius)
//   solve (J^T J + D^2) y = J^T r ; step = -y
//   model_cost_change = -(J step) . (r + (J step)/2)
//   rho = (cost - new_cost) / model_cost_change ; accept iff rho > min_relative_decrease
//   accept: radius /= max(1/3, 1 - (2 rho - 1)^3), decrease_factor = 2
//   reject: radius /= decrease_factor, decrease_factor *= 2
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "functors.hpp"
#include "loss.hpp"
#include "parameterization.hpp"
#include "rotation.hpp"
#include "oracle.h"

namespace oracle {

// dense Cholesky (chol.cpp; compiled with vectorisation flags)
int cholesky_lower_inplace(double* A, int n, int ld, int num_threads, const int* last_row = nullptr);
void cholesky_solve_lower(const double* L, int n, int ld, double* b, const int* last_row = nullptr);

static double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// --------------------------------------------------------------------------
// functor dispatch
// --------------------------------------------------------------------------
struct FunctorInfo { int num_residuals, num_blocks, num_consts; const int* N; };

template <class F> static FunctorInfo info_of() {
  FunctorInfo fi; fi.num_residuals = F::kNumResiduals; fi.num_blocks = F::kNumBlocks;
  fi.num_consts = F::kNumConsts; fi.N = F::N; return fi;
}

static bool functor_info(int id, FunctorInfo* fi) {
  switch (id) {
    case kSnavelyReprojectionError: *fi = info_of<SnavelyReprojectionError>(); return true;
    case kExponentialResidual: *fi = info_of<ExponentialResidual>(); return true;
    case kPowellF1: *fi = info_of<PowellF1>(); return true;
    case kPowellF2: *fi = info_of<PowellF2>(); return true;
    case kPowellF3: *fi = info_of<PowellF3>(); return true;
    case kPowellF4: *fi = info_of<PowellF4>(); return true;
    case kBinaryScalarCost: *fi = info_of<BinaryScalarCost>(); return true;
    case kBinaryVector3Cost: *fi = info_of<BinaryVector3Cost>(); return true;
    case kTenParameterCost: *fi = info_of<TenParameterCost>(); return true;
    case kHelloCostFunctor: *fi = info_of<HelloCostFunctor>(); return true;
    case kQuaternionRotationError: *fi = info_of<QuaternionRotationError>(); return true;
  }
  return false;
}

static bool evaluate_block(int id, const double* consts, double const* const* params,
                           double* residuals, double** jacobians) {
  switch (id) {
    case kSnavelyReprojectionError: return AutoDiff<SnavelyReprojectionError>::evaluate(consts, params, residuals, jacobians);
    case kExponentialResidual: return AutoDiff<ExponentialResidual>::evaluate(consts, params, residuals, jacobians);
    case kPowellF1: return AutoDiff<PowellF1>::evaluate(consts, params, residuals, jacobians);
    case kPowellF2: return AutoDiff<PowellF2>::evaluate(consts, params, residuals, jacobians);
    case kPowellF3: return AutoDiff<PowellF3>::evaluate(consts, params, residuals, jacobians);
    case kPowellF4: return AutoDiff<PowellF4>::evaluate(consts, params, residuals, jacobians);
    case kBinaryScalarCost: return AutoDiff<BinaryScalarCost>::evaluate(consts, params, residuals, jacobians);
    case kBinaryVector3Cost: return AutoDiff<BinaryVector3Cost>::evaluate(consts, params, residuals, jacobians);
    case kTenParameterCost: return AutoDiff<TenParameterCost>::evaluate(consts, params, residuals, jacobians);
    case kHelloCostFunctor: return AutoDiff<HelloCostFunctor>::evaluate(consts, params, residuals, jacobians);
    case kQuaternionRotationError: return AutoDiff<QuaternionRotationError>::evaluate(consts, params, residuals, jacobians);
  }
  return false;
}

// --------------------------------------------------------------------------
// small dense helpers
// --------------------------------------------------------------------------
static inline bool all_finite(const double* v, size_t n) {
  for (size_t i = 0; i < n; ++i) if (!std::isfinite(v[i])) return false;
  return true;
}

// 3x3 SPD inverse through its Cholesky factor (Ceres: InvertPSDMatrix -> LLT).
static bool invert_spd3(const double* T, double* Tinv) {
  double l00 = T[0]; if (!(l00 > 0.0)) return false; l00 = std::sqrt(l00);
  const double l10 = T[3] / l00, l20 = T[6] / l00;
  double l11 = T[4] - l10 * l10; if (!(l11 > 0.0)) return false; l11 = std::sqrt(l11);
  const double l21 = (T[7] - l20 * l10) / l11;
  double l22 = T[8] - l20 * l20 - l21 * l21; if (!(l22 > 0.0)) return false; l22 = std::sqrt(l22);
  // M = L^-1 (lower)
  const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
  const double m10 = -(l10 * m00) / l11;
  const double m21 = -(l21 * m11) / l22;
  const double m20 = -(l20 * m00 + l21 * m10) / l22;
  // Tinv = M^T M
  Tinv[0] = m00 * m00 + m10 * m10 + m20 * m20;
  Tinv[1] = Tinv[3] = m10 * m11 + m20 * m21;
  Tinv[2] = Tinv[6] = m20 * m22;
  Tinv[4] = m11 * m11 + m21 * m21;
  Tinv[5] = Tinv[7] = m21 * m22;
  Tinv[8] = m22 * m22;
  return true;
}

// --------------------------------------------------------------------------
// Problem in oracle form
// --------------------------------------------------------------------------
struct Problem {
  int num_blocks = 0;
  std::vector<int> block_size, block_off;  // parameter blocks in x
  int num_params = 0;
  int num_res_blocks = 0;
  std::vector<int> functor, res_off, const_off, pidx_off, pidx;
  const double* consts = nullptr;
  int num_residuals = 0;
  const double* loss_nodes = nullptr;  // oracle/loss.hpp; block_loss[b] = root node or -1
  std::vector<int> block_loss;
  // oracle/parameterization.hpp: one per parameter block when any block has one (identity otherwise)
  std::vector<Parameterization> params;
  std::vector<int> local_off;
  int num_local = 0;
};

struct DenseEval {  // dense Jacobian, row-major m x n
  std::vector<double> r, J;
};

static bool evaluate_dense(const Problem& P, const double* x, bool want_jac, DenseEval* out,
                           double* cost) {
  const int m = P.num_residuals, n = P.num_params;
  out->r.assign(m, 0.0);
  if (want_jac) out->J.assign((size_t)m * n, 0.0);
  bool ok = true;
  double csum = 0.0;  // sum of the blocks' cost terms rho(|r_b|^2)
  for (int b = 0; b < P.num_res_blocks; ++b) {
    if (P.functor[b] == kSynthTanhRow) {  // dense row over one block of any size
      const int blk = P.pidx[P.pidx_off[b]], off = P.block_off[blk], nb = P.block_size[blk];
      if (!synth_tanh_row_evaluate(P.consts + P.const_off[b], x + off, nb, &out->r[P.res_off[b]],
                                   want_jac ? &out->J[(size_t)P.res_off[b] * n + off] : nullptr)) { ok = false; break; }
      {  // the block's loss and corrector, as for every other residual block below (the Jacobian row is corrected in place)
        int Nb[1] = {nb};
        double* jrow[1] = {want_jac ? &out->J[(size_t)P.res_off[b] * n + off] : nullptr};
        csum += oracle::loss_correct(P.loss_nodes, P.block_loss.empty() ? -1 : P.block_loss[b], 1, &out->r[P.res_off[b]], 1, Nb, want_jac ? jrow : nullptr);
      }
      continue;
    }
    FunctorInfo fi; functor_info(P.functor[b], &fi);
    const double* params[16]; double jbuf[16][64]; double* jac[16];
    for (int i = 0; i < fi.num_blocks; ++i) {
      params[i] = x + P.block_off[P.pidx[P.pidx_off[b] + i]];
      jac[i] = jbuf[i];
    }
    double res[8];
    if (!evaluate_block(P.functor[b], P.consts + P.const_off[b], params, res,
                        want_jac ? jac : nullptr)) { ok = false; break; }
    csum += oracle::loss_correct(P.loss_nodes, P.block_loss.empty() ? -1 : P.block_loss[b], fi.num_residuals, res, fi.num_blocks, fi.N,
                                 want_jac ? jac : nullptr);
    for (int r = 0; r < fi.num_residuals; ++r) out->r[P.res_off[b] + r] = res[r];
    if (want_jac) {
      for (int i = 0; i < fi.num_blocks; ++i) {
        const int off = P.block_off[P.pidx[P.pidx_off[b] + i]];
        for (int r = 0; r < fi.num_residuals; ++r)
          for (int p = 0; p < fi.N[i]; ++p)
            out->J[(size_t)(P.res_off[b] + r) * n + off + p] += jbuf[i][r * fi.N[i] + p];
      }
    }
  }
  if (!ok) return false;
  *cost = 0.5 * csum;
  return true;
}

// Householder QR least squares on the (rows x n) column-major matrix A,
// rhs b (length rows).  Returns y with R y = (Q^T b)[0:n].
static bool householder_qr_solve(std::vector<double>& A, std::vector<double>& b, int rows, int n,
                                 double* y) {
  for (int k = 0; k < n; ++k) {
    double* a = &A[(size_t)k * rows];
    double norm = 0.0;
    for (int i = k; i < rows; ++i) norm += a[i] * a[i];
    norm = std::sqrt(norm);
    if (norm == 0.0) return false;
    const double alpha = a[k] > 0 ? -norm : norm;
    // v = a[k:] - alpha e_k ; normalised implicitly
    const double v0 = a[k] - alpha;
    double vnorm2 = v0 * v0;
    for (int i = k + 1; i < rows; ++i) vnorm2 += a[i] * a[i];
    if (vnorm2 == 0.0) { a[k] = alpha; continue; }
    std::vector<double> v(rows - k);
    v[0] = v0; for (int i = k + 1; i < rows; ++i) v[i - k] = a[i];
    a[k] = alpha; for (int i = k + 1; i < rows; ++i) a[i] = 0.0;
    for (int j = k + 1; j < n; ++j) {
      double* c = &A[(size_t)j * rows];
      double dot = 0.0; for (int i = k; i < rows; ++i) dot += v[i - k] * c[i];
      const double f = 2.0 * dot / vnorm2;
      for (int i = k; i < rows; ++i) c[i] -= f * v[i - k];
    }
    double dot = 0.0; for (int i = k; i < rows; ++i) dot += v[i - k] * b[i];
    const double f = 2.0 * dot / vnorm2;
    for (int i = k; i < rows; ++i) b[i] -= f * v[i - k];
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < n; ++j) s -= A[(size_t)j * rows + k] * y[j];
    const double d = A[(size_t)k * rows + k];
    if (d == 0.0) return false;
    y[k] = s / d;
  }
  return true;
}

// --------------------------------------------------------------------------
// BAL-shaped (kRes=2, f-block 9, e-block 3) Schur path
// --------------------------------------------------------------------------
struct Bal {
  int C = 0, P = 0, N = 0;
  std::vector<int> cam, pt;                  // per observation
  std::vector<int> cam_start, cam_obs;       // CSR: observations of a camera (sorted by point)
  std::vector<int> pt_start, pt_obs;         // CSR: observations of a point (sorted by camera)
  const double* consts = nullptr;            // 2 per observation
  std::vector<double> r, F, E;               // 2N, 18N, 6N
  const double* loss_nodes = nullptr; int loss_root = -1;  // one loss for all blocks (oracle/loss.hpp)
  // or_options::cholesky_envelope: column envelope of the reduced camera system in the caller's camera numbering
  // (last_row[j] >= j, non-decreasing; empty = factor every entry).  Camera c shares a point with no camera beyond
  // the last camera of any of its points, so rows below 9 * that + 8 are structural zeros in S and in its factor.
  // Parameter-block state on the Schur path: bit k of cam_mask[i] / pt_mask[p] = coordinate k of camera i / point p is
  // held constant — SetParameterBlockConstant (all bits) or a SubsetParameterization (ceres.i:186-210; oracle/parameterization.hpp
  // P_SUBSET / P_CONSTANT).  For both the tangent-space Jacobian is the ambient one without the constant columns and Plus
  // adds the step to the free coordinates; here the constant columns are ZEROED instead of removed (the blocks keep their
  // 9 / 3 columns): a zero column gets Jacobi scale 1, LM diagonal min_lm_diagonal / radius, no coupling and a zero
  // right-hand side, so its step is exactly 0 and every other entry of the system is the reduced one's.  The dense path
  // (or_solve_param) removes the columns for real: tests/test_oracle_kat.py holds the two against each other.
  std::vector<int> cam_mask, pt_mask;  // empty = everything free
  std::vector<int> last_row;
  void build_envelope() {
    std::vector<int> last_cam(C);
    for (int c = 0; c < C; ++c) last_cam[c] = c;
    for (int p = 0; p < P; ++p) {
      if (pt_start[p] == pt_start[p + 1]) continue;
      const int cmax = cam[pt_obs[pt_start[p + 1] - 1]];  // pt_obs is sorted by camera
      for (int k = pt_start[p]; k < pt_start[p + 1]; ++k) { int& l = last_cam[cam[pt_obs[k]]]; if (cmax > l) l = cmax; }
    }
    for (int c = 1; c < C; ++c) if (last_cam[c] < last_cam[c - 1]) last_cam[c] = last_cam[c - 1];
    last_row.resize(9 * (size_t)C);
    for (int j = 0; j < 9 * C; ++j) last_row[j] = 9 * last_cam[j / 9] + 8;
  }
};

static bool bal_evaluate(Bal& B, const double* x, bool want_jac, double* cost, int nthreads) {
  const int N = B.N; const double* cams = x; const double* pts = x + 9 * (size_t)B.C;
  B.r.resize(2 * (size_t)N);
  if (want_jac) { B.F.resize(18 * (size_t)N); B.E.resize(6 * (size_t)N); }
  int bad = 0;
  std::vector<double> term(B.loss_root >= 0 ? N : 0);
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : bad)
  for (int o = 0; o < N; ++o) {
    const double* params[2] = {cams + 9 * (size_t)B.cam[o], pts + 3 * (size_t)B.pt[o]};
    double* jac[2] = {want_jac ? &B.F[18 * (size_t)o] : nullptr, want_jac ? &B.E[6 * (size_t)o] : nullptr};
    if (!AutoDiff<SnavelyReprojectionError>::evaluate(B.consts + 2 * (size_t)o, params,
                                                      &B.r[2 * (size_t)o], want_jac ? jac : nullptr)) { bad++; continue; }
    if (B.loss_root >= 0) {
      static const int kN[2] = {9, 3};
      term[o] = oracle::loss_correct(B.loss_nodes, B.loss_root, 2, &B.r[2 * (size_t)o], 2, kN, want_jac ? jac : nullptr);
    }
    if (want_jac && !B.cam_mask.empty()) {
      const int mc = B.cam_mask[B.cam[o]], mp = B.pt_mask[B.pt[o]];
      for (int c = 0; c < 9; ++c) if ((mc >> c) & 1) { jac[0][c] = 0.0; jac[0][9 + c] = 0.0; }
      for (int a = 0; a < 3; ++a) if ((mp >> a) & 1) { jac[1][a] = 0.0; jac[1][3 + a] = 0.0; }
    }
  }
  if (bad) return false;
  double c = 0.0;
  if (B.loss_root >= 0) for (int o = 0; o < N; ++o) c += term[o];
  else for (size_t i = 0; i < 2 * (size_t)N; ++i) c += B.r[i] * B.r[i];
  *cost = 0.5 * c;
  return std::isfinite(*cost);
}

struct SchurWork {
  std::vector<double> S, rhs, Tinv, g, W, Y;
  double t_assemble = 0, t_chol = 0, t_backsub = 0;
};

// Reduced camera system S (lower), rhs for the Jacobian held in B.F / B.E:
// eliminate the point blocks of (J^T J + D^2) y = J^T r.  add_Dc == false leaves
// D_c^2 off the diagonal (used to check that per-shard systems add up).
static bool bal_schur_assemble(const Bal& B, const double* D, SchurWork& w, int nthreads, bool add_Dc) {
  const int C = B.C, P = B.P, N = B.N; const int n = 9 * C;
  const double* Dc = D; const double* Dp = D + 9 * (size_t)C;
  double t0 = now_s();
  w.S.assign((size_t)n * n, 0.0); w.rhs.assign(n, 0.0);
  w.Tinv.resize(9 * (size_t)P); w.g.resize(3 * (size_t)P);
  w.W.resize(27 * (size_t)N); w.Y.resize(27 * (size_t)N);
  int bad = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256) reduction(+ : bad)
  for (int p = 0; p < P; ++p) {
    double T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int k = B.pt_start[p]; k < B.pt_start[p + 1]; ++k) {
      const int o = B.pt_obs[k]; const double* E = &B.E[6 * (size_t)o]; const double* r = &B.r[2 * (size_t)o];
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[3 * a + b] += E[a] * E[b] + E[3 + a] * E[3 + b];
        g[a] += E[a] * r[0] + E[3 + a] * r[1];
      }
    }
    for (int a = 0; a < 3; ++a) T[4 * a] += Dp[3 * (size_t)p + a] * Dp[3 * (size_t)p + a];
    double* Ti = &w.Tinv[9 * (size_t)p];
    if (!invert_spd3(T, Ti)) { bad++; continue; }
    for (int a = 0; a < 3; ++a) w.g[3 * (size_t)p + a] = g[a];
    for (int k = B.pt_start[p]; k < B.pt_start[p + 1]; ++k) {
      const int o = B.pt_obs[k]; const double* F = &B.F[18 * (size_t)o]; const double* E = &B.E[6 * (size_t)o];
      double* W = &w.W[27 * (size_t)o]; double* Y = &w.Y[27 * (size_t)o];
      for (int c = 0; c < 9; ++c)
        for (int a = 0; a < 3; ++a) W[3 * c + a] = F[c] * E[a] + F[9 + c] * E[3 + a];
      for (int c = 0; c < 9; ++c)
        for (int a = 0; a < 3; ++a)
          Y[3 * c + a] = W[3 * c] * Ti[a] + W[3 * c + 1] * Ti[3 + a] + W[3 * c + 2] * Ti[6 + a];
    }
  }
  if (bad) return false;
  // Row-block i of S is owned by one thread: deterministic accumulation order
  // (observations of camera i ascending by point; partners ascending by camera).
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4)
  for (int i = 0; i < C; ++i) {
    double* Srow = &w.S[(size_t)(9 * i) * n]; double* rhs = &w.rhs[9 * (size_t)i];
    if (add_Dc) for (int c = 0; c < 9; ++c) Srow[(size_t)c * n + 9 * i + c] += Dc[9 * (size_t)i + c] * Dc[9 * (size_t)i + c];
    for (int k = B.cam_start[i]; k < B.cam_start[i + 1]; ++k) {
      const int a = B.cam_obs[k]; const int p = B.pt[a];
      const double* F = &B.F[18 * (size_t)a]; const double* r = &B.r[2 * (size_t)a];
      const double* Ya = &w.Y[27 * (size_t)a]; const double* g = &w.g[3 * (size_t)p];
      for (int c = 0; c < 9; ++c) {
        for (int d = 0; d < 9; ++d) Srow[(size_t)c * n + 9 * i + d] += F[c] * F[d] + F[9 + c] * F[9 + d];
        rhs[c] += F[c] * r[0] + F[9 + c] * r[1];
        rhs[c] -= Ya[3 * c] * g[0] + Ya[3 * c + 1] * g[1] + Ya[3 * c + 2] * g[2];
      }
      for (int kk = B.pt_start[p]; kk < B.pt_start[p + 1]; ++kk) {
        const int b = B.pt_obs[kk]; const int j = B.cam[b];
        if (j > i) break;  // lower triangle only (pt_obs sorted by camera)
        const double* Wb = &w.W[27 * (size_t)b];
        for (int c = 0; c < 9; ++c)
          for (int d = 0; d < 9; ++d)
            Srow[(size_t)c * n + 9 * j + d] -= Ya[3 * c] * Wb[3 * d] + Ya[3 * c + 1] * Wb[3 * d + 1] + Ya[3 * c + 2] * Wb[3 * d + 2];
      }
    }
  }
  w.t_assemble += now_s() - t0;
  return true;
}

// Solve (J^T J + D^2) y = J^T r for the scaled Jacobian held in B.F / B.E.
static bool bal_schur_solve(const Bal& B, const double* D, double* y, SchurWork& w, int nthreads) {
  const int C = B.C, P = B.P; const int n = 9 * C;
  if (!bal_schur_assemble(B, D, w, nthreads, true)) return false;
  double t1 = now_s();
  const int* env = B.last_row.empty() ? nullptr : B.last_row.data();
  if (cholesky_lower_inplace(w.S.data(), n, n, nthreads, env) != 0) return false;
  std::vector<double> yc(w.rhs);
  cholesky_solve_lower(w.S.data(), n, n, yc.data(), env);
  double t2 = now_s(); w.t_chol += t2 - t1;
  for (int i = 0; i < n; ++i) y[i] = yc[i];
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
  for (int p = 0; p < P; ++p) {
    double t[3] = {w.g[3 * (size_t)p], w.g[3 * (size_t)p + 1], w.g[3 * (size_t)p + 2]};
    for (int k = B.pt_start[p]; k < B.pt_start[p + 1]; ++k) {
      const int o = B.pt_obs[k]; const double* W = &w.W[27 * (size_t)o]; const double* yi = &yc[9 * (size_t)B.cam[o]];
      for (int a = 0; a < 3; ++a) {
        double s = 0.0; for (int c = 0; c < 9; ++c) s += W[3 * c + a] * yi[c];
        t[a] -= s;
      }
    }
    const double* Ti = &w.Tinv[9 * (size_t)p];
    for (int a = 0; a < 3; ++a) y[n + 3 * (size_t)p + a] = Ti[3 * a] * t[0] + Ti[3 * a + 1] * t[1] + Ti[3 * a + 2] * t[2];
  }
  w.t_backsub += now_s() - t2;
  return true;
}

// --------------------------------------------------------------------------
// The trust-region loop, shared by the dense and the BAL/Schur representations
// --------------------------------------------------------------------------
struct Model {
  // evaluate at x: cost (+ residuals/Jacobian kept inside when want_jac)
  virtual bool evaluate(const double* x, bool want_jac, double* cost) = 0;
  virtual int num_params() const = 0;
  virtual int num_local() const { return num_params(); }  // size of the tangent space (local parameterizations)
  virtual void plus(const double* x, const double* delta, double* out) const { for (int j = 0; j < num_params(); ++j) out[j] = x[j] + delta[j]; }
  virtual void column_sq_norms(double* out) = 0;      // of the CURRENT (possibly scaled) Jacobian
  virtual void scale_columns(const double* s) = 0;    // J <- J diag(s)
  virtual void gradient(double* g) = 0;               // J^T r of the CURRENT Jacobian
  virtual bool solve(const double* D, double* y) = 0;  // (J^T J + D^2) y = J^T r
  virtual double model_cost_change(const double* step) = 0;
  virtual ~Model() {}
};

struct DenseModel : Model {
  const Problem& P; int solver; DenseEval cur, cand;
  DenseModel(const Problem& p, int s) : P(p), solver(s) {}
  int num_params() const override { return P.num_params; }
  int num_local() const override { return P.params.empty() ? P.num_params : P.num_local; }
  void plus(const double* x, const double* delta, double* out) const override {
    if (P.params.empty()) { Model::plus(x, delta, out); return; }
    for (int b = 0; b < P.num_blocks; ++b) parameterization_plus(P.params[b], x + P.block_off[b], delta + P.local_off[b], out + P.block_off[b]);
  }
  bool evaluate(const double* x, bool want_jac, double* cost) override {
    if (!(evaluate_dense(P, x, want_jac, want_jac ? &cur : &cand, cost) && std::isfinite(*cost))) return false;
    if (want_jac && !P.params.empty()) {
      // the Jacobian the minimiser sees is w.r.t. the tangent vector: J_local = J_global * blockdiag(d Plus / d delta)
      const int m = P.num_residuals, ng = P.num_params, nl = P.num_local;
      std::vector<double> Jl((size_t)m * nl, 0.0), pj;
      for (int b = 0; b < P.num_blocks; ++b) {
        const int gs = P.block_size[b], ls = P.params[b].local_size();
        pj.assign((size_t)gs * std::max(ls, 1), 0.0);
        parameterization_jacobian(P.params[b], x + P.block_off[b], pj.data());
        for (int i = 0; i < m; ++i)
          for (int c = 0; c < ls; ++c) {
            double acc = 0.0;
            for (int g = 0; g < gs; ++g) acc += cur.J[(size_t)i * ng + P.block_off[b] + g] * pj[(size_t)g * ls + c];
            Jl[(size_t)i * nl + P.local_off[b] + c] = acc;
          }
      }
      cur.J.swap(Jl);
    }
    return true;
  }
  void column_sq_norms(double* out) override {
    const int m = P.num_residuals, n = num_local();
    for (int j = 0; j < n; ++j) out[j] = 0.0;
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) out[j] += cur.J[(size_t)i * n + j] * cur.J[(size_t)i * n + j];
  }
  void scale_columns(const double* s) override {
    const int m = P.num_residuals, n = num_local();
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) cur.J[(size_t)i * n + j] *= s[j];
  }
  void gradient(double* g) override {
    const int m = P.num_residuals, n = num_local();
    for (int j = 0; j < n; ++j) g[j] = 0.0;
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) g[j] += cur.J[(size_t)i * n + j] * cur.r[i];
  }
  bool solve(const double* D, double* y) override {
    const int m = P.num_residuals, n = num_local();
    if (solver == OR_DENSE_QR) {
      const int rows = m + n;
      std::vector<double> A((size_t)rows * n, 0.0), b(rows, 0.0);
      for (int i = 0; i < m; ++i) { b[i] = cur.r[i]; for (int j = 0; j < n; ++j) A[(size_t)j * rows + i] = cur.J[(size_t)i * n + j]; }
      for (int j = 0; j < n; ++j) A[(size_t)j * rows + m + j] = D[j];
      return householder_qr_solve(A, b, rows, n, y);
    }
    std::vector<double> H((size_t)n * n, 0.0);
    for (int i = 0; i < m; ++i) {
      const double* Ji = &cur.J[(size_t)i * n];
      for (int a = 0; a < n; ++a) { const double ja = Ji[a]; if (ja == 0.0) continue; for (int b = 0; b <= a; ++b) H[(size_t)a * n + b] += ja * Ji[b]; }
    }
    for (int a = 0; a < n; ++a) H[(size_t)a * n + a] += D[a] * D[a];
    gradient(y);
    if (cholesky_lower_inplace(H.data(), n, n, 1) != 0) return false;
    cholesky_solve_lower(H.data(), n, n, y);
    return true;
  }
  double model_cost_change(const double* step) override {
    const int m = P.num_residuals, n = num_local(); double acc = 0.0;
    for (int i = 0; i < m; ++i) {
      double mr = 0.0; for (int j = 0; j < n; ++j) mr += cur.J[(size_t)i * n + j] * step[j];
      acc += mr * (cur.r[i] + mr / 2.0);
    }
    return -acc;
  }
};

struct BalModel : Model {
  Bal& B; SchurWork w; int nthreads; std::vector<double> r_cur;
  BalModel(Bal& b, int nt) : B(b), nthreads(nt) {}
  int num_params() const override { return 9 * B.C + 3 * B.P; }
  bool evaluate(const double* x, bool want_jac, double* cost) override {
    if (!want_jac) { std::vector<double> keep; keep.swap(B.r); bool ok = bal_evaluate(B, x, false, cost, nthreads); B.r.swap(keep); return ok; }
    return bal_evaluate(B, x, true, cost, nthreads);
  }
  void column_sq_norms(double* out) override {
    const int C = B.C, P = B.P;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
    for (int i = 0; i < C; ++i) {
      double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = B.cam_start[i]; k < B.cam_start[i + 1]; ++k) { const double* F = &B.F[18 * (size_t)B.cam_obs[k]]; for (int c = 0; c < 9; ++c) s[c] += F[c] * F[c] + F[9 + c] * F[9 + c]; }
      for (int c = 0; c < 9; ++c) out[9 * (size_t)i + c] = s[c];
    }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
    for (int p = 0; p < P; ++p) {
      double s[3] = {0, 0, 0};
      for (int k = B.pt_start[p]; k < B.pt_start[p + 1]; ++k) { const double* E = &B.E[6 * (size_t)B.pt_obs[k]]; for (int a = 0; a < 3; ++a) s[a] += E[a] * E[a] + E[3 + a] * E[3 + a]; }
      for (int a = 0; a < 3; ++a) out[9 * (size_t)C + 3 * (size_t)p + a] = s[a];
    }
  }
  void scale_columns(const double* s) override {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int o = 0; o < B.N; ++o) {
      const double* sc = s + 9 * (size_t)B.cam[o]; const double* sp = s + 9 * (size_t)B.C + 3 * (size_t)B.pt[o];
      double* F = &B.F[18 * (size_t)o]; double* E = &B.E[6 * (size_t)o];
      for (int c = 0; c < 9; ++c) { F[c] *= sc[c]; F[9 + c] *= sc[c]; }
      for (int a = 0; a < 3; ++a) { E[a] *= sp[a]; E[3 + a] *= sp[a]; }
    }
  }
  void gradient(double* g) override {
    const int C = B.C, P = B.P;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
    for (int i = 0; i < C; ++i) {
      double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = B.cam_start[i]; k < B.cam_start[i + 1]; ++k) { const int o = B.cam_obs[k]; const double* F = &B.F[18 * (size_t)o]; for (int c = 0; c < 9; ++c) s[c] += F[c] * B.r[2 * (size_t)o] + F[9 + c] * B.r[2 * (size_t)o + 1]; }
      for (int c = 0; c < 9; ++c) g[9 * (size_t)i + c] = s[c];
    }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
    for (int p = 0; p < P; ++p) {
      double s[3] = {0, 0, 0};
      for (int k = B.pt_start[p]; k < B.pt_start[p + 1]; ++k) { const int o = B.pt_obs[k]; const double* E = &B.E[6 * (size_t)o]; for (int a = 0; a < 3; ++a) s[a] += E[a] * B.r[2 * (size_t)o] + E[3 + a] * B.r[2 * (size_t)o + 1]; }
      for (int a = 0; a < 3; ++a) g[9 * (size_t)C + 3 * (size_t)p + a] = s[a];
    }
  }
  bool solve(const double* D, double* y) override { return bal_schur_solve(B, D, y, w, nthreads); }
  double model_cost_change(const double* step) override {
    double acc = 0.0;
    for (int o = 0; o < B.N; ++o) {
      const double* sc = step + 9 * (size_t)B.cam[o]; const double* sp = step + 9 * (size_t)B.C + 3 * (size_t)B.pt[o];
      const double* F = &B.F[18 * (size_t)o]; const double* E = &B.E[6 * (size_t)o];
      for (int r = 0; r < 2; ++r) {
        double mr = 0.0;
        for (int c = 0; c < 9; ++c) mr += F[9 * r + c] * sc[c];
        for (int a = 0; a < 3; ++a) mr += E[3 * r + a] * sp[a];
        acc += mr * (B.r[2 * (size_t)o + r] + mr / 2.0);
      }
    }
    return -acc;
  }
};

static double max_abs(const double* v, int n) { double m = 0.0; for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(v[i])); return m; }
static double norm2(const double* v, int n) { double s = 0.0; for (int i = 0; i < n; ++i) s += v[i] * v[i]; return std::sqrt(s); }

static void log_iter(or_summary* S, int it, double cost, double cost_change, double gmax, double step_norm,
                     double rho, double radius, int valid, int success) {
  if (it < OR_MAX_LOG) {
    or_iteration& L = S->iterations[it];
    L.iteration = it; L.cost = cost; L.cost_change = cost_change; L.gradient_max_norm = gmax;
    L.step_norm = step_norm; L.relative_decrease = rho; L.trust_region_radius = radius;
    L.step_is_valid = valid; L.step_is_successful = success;
    L.time_s = now_s();  // wall clock at the end of the iteration (differences = iteration times; bench.py's CPU baseline)
    S->num_logged = it + 1;
  }
}

static void minimize(Model& M, const or_options& O, double* x, or_summary* S) {
  const int n = M.num_local(), ng = M.num_params();  // tangent / ambient sizes (equal without local parameterizations)
  std::vector<double> scale(n, 1.0), diag(n), D(n), y(n), step(n), delta(n), xc(ng), g(n), gs(n);
  double cost = 0.0;
  std::memset(S, 0, sizeof(*S));
  double t_begin = now_s();
  if (!M.evaluate(x, true, &cost)) { S->termination_type = OR_FAILURE; std::snprintf(S->message, sizeof(S->message), "Initial residual and Jacobian evaluation failed."); return; }
  S->initial_cost = cost;
  M.gradient(g.data());  // unscaled Jacobian
  double gmax = max_abs(g.data(), n);
  if (O.jacobi_scaling) {
    M.column_sq_norms(diag.data());
    for (int j = 0; j < n; ++j) scale[j] = 1.0 / (1.0 + std::sqrt(diag[j]));
    M.scale_columns(scale.data());
  }
  double radius = O.initial_trust_region_radius, decrease_factor = 2.0;
  double x_norm = norm2(x, ng);
  int iteration = 0, invalid = 0, n_success = 0, n_unsuccess = 0;
  log_iter(S, 0, cost, 0.0, gmax, 0.0, 0.0, radius, 1, 1);
  S->termination_type = OR_NO_CONVERGENCE;
  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iteration >= O.max_num_iterations) { S->termination_type = OR_NO_CONVERGENCE; std::snprintf(S->message, sizeof(S->message), "Maximum number of iterations reached. Number of iterations: %d.", iteration); break; }
    if (gmax <= O.gradient_tolerance) { S->termination_type = OR_CONVERGENCE; std::snprintf(S->message, sizeof(S->message), "Gradient tolerance reached. Gradient max norm: %e <= %e", gmax, O.gradient_tolerance); break; }
    if (radius < O.min_trust_region_radius) { S->termination_type = OR_CONVERGENCE; std::snprintf(S->message, sizeof(S->message), "Minimum trust region radius reached. Trust region radius: %e <= %e", radius, O.min_trust_region_radius); break; }
    ++iteration;
    // LevenbergMarquardtStrategy::ComputeStep
    M.column_sq_norms(diag.data());
    for (int j = 0; j < n; ++j) D[j] = std::sqrt(std::min(std::max(diag[j], O.min_lm_diagonal), O.max_lm_diagonal) / radius);
    bool valid = M.solve(D.data(), y.data()) && all_finite(y.data(), n);
    double mcc = 0.0;
    if (valid) {
      for (int j = 0; j < n; ++j) step[j] = -y[j];
      mcc = M.model_cost_change(step.data());
      if (!(mcc > 0.0)) valid = false;
    }
    if (!valid) {
      ++invalid;
      if (invalid >= O.max_num_consecutive_invalid_steps) { S->termination_type = OR_FAILURE; std::snprintf(S->message, sizeof(S->message), "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d", O.max_num_consecutive_invalid_steps); log_iter(S, iteration, cost, 0.0, gmax, 0.0, 0.0, radius, 0, 0); break; }
      radius /= decrease_factor; decrease_factor *= 2.0;  // StepIsInvalid == StepRejected
      log_iter(S, iteration, cost, 0.0, gmax, 0.0, 0.0, radius, 0, 0);
      continue;
    }
    invalid = 0;
    for (int j = 0; j < n; ++j) delta[j] = step[j] * scale[j];
    M.plus(x, delta.data(), xc.data());
    double new_cost = std::numeric_limits<double>::max();
    if (!M.evaluate(xc.data(), false, &new_cost)) new_cost = std::numeric_limits<double>::max();
    double sn = 0.0; for (int j = 0; j < ng; ++j) { const double d = x[j] - xc[j]; sn += d * d; } sn = std::sqrt(sn);
    const double cost_change = cost - new_cost;
    if (sn <= O.parameter_tolerance * (x_norm + O.parameter_tolerance)) {
      S->termination_type = OR_CONVERGENCE; std::snprintf(S->message, sizeof(S->message), "Parameter tolerance reached. Relative step_norm: %e <= %e.", sn / (x_norm + O.parameter_tolerance), O.parameter_tolerance);
      log_iter(S, iteration, cost, cost_change, gmax, sn, 0.0, radius, 1, 0); break;
    }
    if (std::fabs(cost_change) <= O.function_tolerance * cost) {
      S->termination_type = OR_CONVERGENCE; std::snprintf(S->message, sizeof(S->message), "Function tolerance reached. |cost_change|/cost: %e <= %e", std::fabs(cost_change) / cost, O.function_tolerance);
      log_iter(S, iteration, cost, cost_change, gmax, sn, 0.0, radius, 1, 0); break;
    }
    const double rho = cost_change / mcc;
    if (rho > O.min_relative_decrease) {
      for (int j = 0; j < ng; ++j) x[j] = xc[j];
      x_norm = norm2(x, ng);
      if (!M.evaluate(x, true, &cost)) { S->termination_type = OR_FAILURE; std::snprintf(S->message, sizeof(S->message), "Residual and Jacobian evaluation failed."); break; }
      M.gradient(g.data()); gmax = max_abs(g.data(), n);
      if (O.jacobi_scaling) M.scale_columns(scale.data());
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
      radius = std::min(O.max_trust_region_radius, radius);
      decrease_factor = 2.0; ++n_success;
      log_iter(S, iteration, cost, cost_change, gmax, sn, rho, radius, 1, 1);
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; ++n_unsuccess;
      log_iter(S, iteration, cost, cost_change, gmax, sn, rho, radius, 1, 0);
    }
  }
  S->final_cost = cost; S->num_iterations = iteration;
  S->num_successful_steps = n_success; S->num_unsuccessful_steps = n_unsuccess;
  S->total_time_s = now_s() - t_begin;
}

}  // namespace oracle

// ============================================================================
// C ABI (ctypes from tests/, bench.py cpu_baseline leg)
// ============================================================================
using namespace oracle;

extern "C" {

void or_options_default(or_options* o) {
  o->linear_solver_type = OR_DENSE_QR;
  o->max_num_iterations = 50;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->jacobi_scaling = 1; o->max_num_consecutive_invalid_steps = 5; o->num_threads = 1;
  o->cholesky_envelope = 0;
}

int or_functor_info(int id, int* num_residuals, int* num_blocks, int* num_consts, int* block_sizes) {
  FunctorInfo fi; if (!functor_info(id, &fi)) return 0;
  *num_residuals = fi.num_residuals; *num_blocks = fi.num_blocks; *num_consts = fi.num_consts;
  for (int i = 0; i < fi.num_blocks; ++i) block_sizes[i] = fi.N[i];
  return 1;
}

// AutoDiffCostFunction.evaluate for one residual block (CORE/AutodiffCostFunction.scala:74-134)
int or_evaluate(int functor_id, const double* consts, double const* const* parameters,
                double* residuals, double** jacobians) {
  return evaluate_block(functor_id, consts, parameters, residuals, jacobians) ? 1 : 0;
}

void or_angle_axis_rotate_point(const double* aa, const double* pt, double* out) {
  angleAxisRotatePoint<double>(aa, pt, out);
}
void or_angle_axis_to_rotation_matrix(const double* aa, double* R_rowmajor) {
  angleAxisToRotationMatrix(aa, R_rowmajor);
}

// Generic problem: parameter blocks live in x (block b at x[block_off[b]]).
int or_solve(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
             const int* functor_ids, const double* consts, const int* const_off,
             const int* pidx, const int* pidx_off, const or_options* opt, or_summary* summary) {
  return or_solve_loss(num_blocks, block_sizes, x, num_res_blocks, functor_ids, consts, const_off, pidx, pidx_off, nullptr, nullptr, opt, summary);
}

// The same with robust losses: loss_nodes as in oracle/loss.hpp, block_loss[b] = root node of block b's loss or -1.
static int solve_dense(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
                       const int* functor_ids, const double* consts, const int* const_off,
                       const int* pidx, const int* pidx_off, const double* loss_nodes, const int* block_loss,
                       const int* param_type, const int* param_const_off, const int* param_const,
                       const or_options* opt, or_summary* summary);

int or_solve_loss(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
                  const int* functor_ids, const double* consts, const int* const_off,
                  const int* pidx, const int* pidx_off, const double* loss_nodes, const int* block_loss,
                  const or_options* opt, or_summary* summary) {
  return solve_dense(num_blocks, block_sizes, x, num_res_blocks, functor_ids, consts, const_off, pidx, pidx_off, loss_nodes, block_loss,
                     nullptr, nullptr, nullptr, opt, summary);
}

// ... and with local parameterizations (oracle/parameterization.hpp): param_type[b] = P_* or -1 (none) per parameter
// block; the indices a subset parameterization holds constant are param_const[param_const_off[b] .. param_const_off[b+1]).
int or_solve_param(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
                   const int* functor_ids, const double* consts, const int* const_off,
                   const int* pidx, const int* pidx_off, const double* loss_nodes, const int* block_loss,
                   const int* param_type, const int* param_const_off, const int* param_const,
                   const or_options* opt, or_summary* summary) {
  return solve_dense(num_blocks, block_sizes, x, num_res_blocks, functor_ids, consts, const_off, pidx, pidx_off, loss_nodes, block_loss,
                     param_type, param_const_off, param_const, opt, summary);
}

static Parameterization make_parameterization(int type, int size, const int* constant, int nconst) {
  Parameterization p; p.type = type < 0 ? P_IDENTITY : type; p.global_size = size;
  if (p.type == P_SUBSET) p.constant.assign(constant, constant + nconst);
  return p;
}
int or_parameterization_local_size(int type, int size, int nconst) { return make_parameterization(type, size, nullptr, 0).type == P_SUBSET ? size - nconst : make_parameterization(type, size, nullptr, 0).local_size(); }
void or_parameterization_plus(int type, int size, const int* constant, int nconst, const double* x, const double* delta, double* x_plus) {
  parameterization_plus(make_parameterization(type, size, constant, nconst), x, delta, x_plus);
}
void or_parameterization_jacobian(int type, int size, const int* constant, int nconst, const double* x, double* J) {
  parameterization_jacobian(make_parameterization(type, size, constant, nconst), x, J);
}

static int solve_dense(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
                       const int* functor_ids, const double* consts, const int* const_off,
                       const int* pidx, const int* pidx_off, const double* loss_nodes, const int* block_loss,
                       const int* param_type, const int* param_const_off, const int* param_const,
                       const or_options* opt, or_summary* summary) {
  Problem P; P.num_blocks = num_blocks;
  if (loss_nodes && block_loss) { P.loss_nodes = loss_nodes; P.block_loss.assign(block_loss, block_loss + num_res_blocks); } P.block_size.assign(block_sizes, block_sizes + num_blocks);
  P.block_off.resize(num_blocks); int off = 0;
  for (int b = 0; b < num_blocks; ++b) { P.block_off[b] = off; off += block_sizes[b]; }
  P.num_params = off; P.num_res_blocks = num_res_blocks;
  if (param_type) {
    P.params.resize(num_blocks); P.local_off.resize(num_blocks); int loff = 0;
    for (int b = 0; b < num_blocks; ++b) {
      const int c0 = param_const_off ? param_const_off[b] : 0, c1 = param_const_off ? param_const_off[b + 1] : 0;
      P.params[b] = make_parameterization(param_type[b], block_sizes[b], param_const ? param_const + c0 : nullptr, c1 - c0);
      P.local_off[b] = loff; loff += P.params[b].local_size();
    }
    P.num_local = loff;
    if (loff == 0) return -5;
  }
  P.functor.assign(functor_ids, functor_ids + num_res_blocks);
  P.const_off.assign(const_off, const_off + num_res_blocks);
  P.pidx_off.assign(pidx_off, pidx_off + num_res_blocks + 1);
  P.pidx.assign(pidx, pidx + pidx_off[num_res_blocks]);
  P.consts = consts; P.res_off.resize(num_res_blocks); int m = 0;
  for (int b = 0; b < num_res_blocks; ++b) {
    if (functor_ids[b] == kSynthTanhRow) {
      if (P.pidx_off[b + 1] - P.pidx_off[b] != 1) return -2;
      P.res_off[b] = m; m += 1;
      continue;
    }
    FunctorInfo fi; if (!functor_info(functor_ids[b], &fi)) return -1;
    if (P.pidx_off[b + 1] - P.pidx_off[b] != fi.num_blocks) return -2;
    for (int i = 0; i < fi.num_blocks; ++i) if (block_sizes[P.pidx[P.pidx_off[b] + i]] != fi.N[i]) return -3;
    P.res_off[b] = m; m += fi.num_residuals;
  }
  P.num_residuals = m;
  if (opt->linear_solver_type != OR_DENSE_QR && opt->linear_solver_type != OR_DENSE_NORMAL_CHOLESKY) return -4;
  DenseModel M(P, opt->linear_solver_type);
  minimize(M, *opt, x, summary);
  return 0;
}

// BAL-shaped problem, SnavelyReprojectionError blocks only, x = [9C cameras | 3P points]
// (memory layout of EX/SimpleBundleAdjuster.scala:18-34).  DENSE_SCHUR.
static int bal_build(Bal& B, int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs) {
  B.C = C; B.P = P; B.N = N; B.cam.assign(cam_idx, cam_idx + N); B.pt.assign(pt_idx, pt_idx + N);
  B.consts = obs;
  B.cam_start.assign(C + 1, 0); B.pt_start.assign(P + 1, 0);
  for (int o = 0; o < N; ++o) { if (cam_idx[o] < 0 || cam_idx[o] >= C || pt_idx[o] < 0 || pt_idx[o] >= P) return -1; B.cam_start[cam_idx[o] + 1]++; B.pt_start[pt_idx[o] + 1]++; }
  for (int i = 0; i < C; ++i) B.cam_start[i + 1] += B.cam_start[i];
  for (int p = 0; p < P; ++p) B.pt_start[p + 1] += B.pt_start[p];
  B.cam_obs.resize(N); B.pt_obs.resize(N);
  { std::vector<int> fill(B.cam_start.begin(), B.cam_start.end() - 1); for (int o = 0; o < N; ++o) B.cam_obs[fill[cam_idx[o]]++] = o; }
  { std::vector<int> fill(B.pt_start.begin(), B.pt_start.end() - 1); for (int o = 0; o < N; ++o) B.pt_obs[fill[pt_idx[o]]++] = o; }
  for (int i = 0; i < C; ++i) std::sort(B.cam_obs.begin() + B.cam_start[i], B.cam_obs.begin() + B.cam_start[i + 1], [&](int a, int b) { return pt_idx[a] != pt_idx[b] ? pt_idx[a] < pt_idx[b] : a < b; });
  for (int p = 0; p < P; ++p) std::sort(B.pt_obs.begin() + B.pt_start[p], B.pt_obs.begin() + B.pt_start[p + 1], [&](int a, int b) { return cam_idx[a] != cam_idx[b] ? cam_idx[a] < cam_idx[b] : a < b; });
  return 0;
}

int or_solve_bal(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                 double* x, const or_options* opt, or_summary* summary) {
  return or_solve_bal_loss(C, P, N, cam_idx, pt_idx, obs, nullptr, -1, x, opt, summary);
}

// Rotation functions (oracle/rotation.hpp).  jet_dim 0: in [n][in_len] doubles; jet_dim K in 1..4: every value is
// (real, K infinitesimals).  Returns 0, or -1 for a bad op / jet_dim / a quaternionToRotation of the zero quaternion.
int or_rotation_apply(int op, int row_major, int jet_dim, const double* in, int n, double* out) {
  if (op < 0 || op > 13) return -1;
  if (jet_dim == 0) {
    for (int i = 0; i < n; ++i)
      if (!oracle::rotationApply<double>(op, row_major, in + (size_t)i * oracle::kRotationIn[op], out + (size_t)i * oracle::kRotationOut[op])) return -1;
    return 0;
  }
  switch (jet_dim) {
    case 1: return oracle::rotationApplyJets<1>(op, row_major, in, n, out) ? 0 : -1;
    case 2: return oracle::rotationApplyJets<2>(op, row_major, in, n, out) ? 0 : -1;
    case 3: return oracle::rotationApplyJets<3>(op, row_major, in, n, out) ? 0 : -1;
    case 4: return oracle::rotationApplyJets<4>(op, row_major, in, n, out) ? 0 : -1;
  }
  return -1;
}

void or_loss_evaluate(const double* loss_nodes, int root, double s, double* rho) { oracle::loss_evaluate(loss_nodes, root, s, rho); }

int or_solve_bal_loss(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                      const double* loss_nodes, int loss_root, double* x, const or_options* opt, or_summary* summary) {
  return or_solve_bal_masks(C, P, N, cam_idx, pt_idx, obs, loss_nodes, loss_root, nullptr, nullptr, x, opt, summary);
}

int or_solve_bal_masks(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                       const double* loss_nodes, int loss_root, const int* cam_mask, const int* pt_mask, double* x,
                       const or_options* opt, or_summary* summary) {
  Bal B;
  if (int rc = bal_build(B, C, P, N, cam_idx, pt_idx, obs)) return rc;
  B.loss_nodes = loss_nodes; B.loss_root = loss_nodes ? loss_root : -1;
  if (cam_mask || pt_mask) {
    B.cam_mask.assign(C, 0); B.pt_mask.assign(P, 0);
    if (cam_mask) for (int i = 0; i < C; ++i) B.cam_mask[i] = cam_mask[i] & 0x1ff;
    if (pt_mask) for (int p = 0; p < P; ++p) B.pt_mask[p] = pt_mask[p] & 0x7;
  }
  if (opt->cholesky_envelope) B.build_envelope();
  int nt = opt->num_threads;
#ifdef _OPENMP
  if (nt <= 0) nt = omp_get_max_threads();
#else
  nt = 1;
#endif
  BalModel M(B, nt);
  minimize(M, *opt, x, summary);
  summary->t_linear_assemble_s = M.w.t_assemble; summary->t_linear_cholesky_s = M.w.t_chol; summary->t_linear_backsub_s = M.w.t_backsub;
  summary->num_threads_used = nt;
  return 0;
}

// Stand-alone pieces for kernel-level parity tests -------------------------------
// Evaluate all Snavely blocks: r (2N), F (18N row-major 2x9), E (6N row-major 2x3).
int or_bal_evaluate(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                    const double* x, double* r, double* F, double* E, double* cost) {
  Bal B; B.C = C; B.P = P; B.N = N; B.cam.assign(cam_idx, cam_idx + N); B.pt.assign(pt_idx, pt_idx + N); B.consts = obs;
  const bool jac = F != nullptr;
  if (!bal_evaluate(B, x, jac, cost, 1)) return 0;
  std::memcpy(r, B.r.data(), sizeof(double) * 2 * (size_t)N);
  if (jac) { std::memcpy(F, B.F.data(), sizeof(double) * 18 * (size_t)N); std::memcpy(E, B.E.data(), sizeof(double) * 6 * (size_t)N); }
  return 1;
}

// Reduced camera system of the observations given (any subset of a problem's
// observations; P is the full point count): S (n x n row-major, lower triangle
// filled) and rhs (n), n = 9C, unscaled Jacobian at x, LM diagonal D (9C + 3P).
// add_Dc != 0 adds D_c^2 to the diagonal.
int or_bal_reduced_system(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                          const double* x, const double* D, int add_Dc, double* S, double* rhs) {
  Bal B;
  if (int rc = bal_build(B, C, P, N, cam_idx, pt_idx, obs)) return rc;
  double cost;
  if (!bal_evaluate(B, x, true, &cost, 1)) return -2;
  SchurWork w;
  if (!bal_schur_assemble(B, D, w, 1, add_Dc != 0)) return -3;
  const size_t n = 9 * (size_t)C;
  std::memcpy(S, w.S.data(), n * n * sizeof(double));
  std::memcpy(rhs, w.rhs.data(), n * sizeof(double));
  return 0;
}

// cost 1/2 sum r_i^2 of m dense rows (kSynthTanhRow; consts: m x 3) at x — the cost-only branch of
// CORE/AutodiffCostFunction.scala:80-93 over every residual block, without the m x n Jacobian a solve would hold
// (BASELINE.json config 5 at full size: 80 GB).  Rows in fixed chunks of 1024, chunk sums added in chunk order.
int or_dense_rows_cost(const double* consts, int m, const double* x, int n, int num_threads, double* cost) {
  int nt = num_threads;
  if (nt <= 0) nt = omp_get_max_threads();
  const int chunks = (m + 1023) / 1024;
  std::vector<double> part((size_t)chunks, 0.0);
  int bad = 0;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 4) reduction(+ : bad)
  for (int c = 0; c < chunks; ++c) {
    double s = 0.0;
    for (int i = c * 1024; i < m && i < (c + 1) * 1024; ++i) {
      double r = 0.0;
      if (!oracle::synth_tanh_row_evaluate(consts + 3 * (size_t)i, x, n, &r, nullptr)) ++bad;
      s += r * r;
    }
    part[c] = s;
  }
  double total = 0.0;
  for (int c = 0; c < chunks; ++c) total += part[c];
  *cost = 0.5 * total;
  return bad ? 1 : 0;
}

int or_cholesky_lower(double* A, int n, int num_threads) { return cholesky_lower_inplace(A, n, n, num_threads); }
int or_cholesky_lower_envelope(double* A, int n, int num_threads, const int* last_row) { return cholesky_lower_inplace(A, n, n, num_threads, last_row); }
void or_cholesky_solve_envelope(const double* L, int n, double* b, const int* last_row) { cholesky_solve_lower(L, n, n, b, last_row); }
void or_cholesky_solve(const double* L, int n, double* b) { cholesky_solve_lower(L, n, n, b); }

}  // extern "C"
