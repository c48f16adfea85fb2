// ORACLE — TEST INFRASTRUCTURE ONLY.
// Robust loss functions and the residual/Jacobian correction, CPU restatement.
//
// The reference only *constructs* native Ceres losses (ceres.i:159-184,
// PredefinedLossFunctions: trivialLoss, huberLoss, softLOneLoss, cauchyLoss,
// tukeyLoss, tolerantLoss, composedLoss, scaledLoss) and hands them to
// Problem.addResidualBlock (CORE/Problem.scala:20-27); what they compute is
// inside Ceres [ext, not vendored].  This file restates the published
// ceres::LossFunction::Evaluate contract of Ceres 1.x (rho[0] = rho(s),
// rho[1] = rho'(s), rho[2] = rho''(s), s = squared residual norm) and
// ceres::internal::Corrector.  PARITY UNPINNED against a Ceres build; pinned
// instead by closed forms, finite differences and SciPy's least_squares
// (loss = 'huber' / 'soft_l1' / 'cauchy' are the same functions), see
// tests/test_oracle_loss.py.
#pragma once
#include <cmath>
#include <limits>

namespace oracle {

// A loss expression is an array of nodes, 5 doubles each: {type, a, b, f, g};
// f, g index child nodes (-1: none).  Types as in PredefinedLossFunctions order.
enum { kTrivialLoss = 0, kHuberLoss = 1, kSoftLOneLoss = 2, kCauchyLoss = 3, kTukeyLoss = 4, kTolerantLoss = 5, kComposedLoss = 6, kScaledLoss = 7 };

inline void loss_evaluate(const double* nodes, int id, double s, double rho[3]) {
  if (id < 0) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; }
  const double* nd = nodes + 5 * (size_t)id;
  const int type = (int)nd[0];
  const double a = nd[1], b = nd[2];
  const int f = (int)nd[3], g = (int)nd[4];
  const double tiny = std::numeric_limits<double>::min();
  switch (type) {
    case kHuberLoss: {  // rho(s) = s for s <= a^2, 2 a sqrt(s) - a^2 beyond
      const double a2 = a * a;
      if (s > a2) {
        const double r = std::sqrt(s);
        rho[0] = 2.0 * a * r - a2; rho[1] = std::max(tiny, a / r); rho[2] = -rho[1] / (2.0 * s);
      } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
      return;
    }
    case kSoftLOneLoss: {  // rho(s) = 2 a^2 (sqrt(1 + s / a^2) - 1)
      const double a2 = a * a, sum = 1.0 + s / a2, root = std::sqrt(sum);
      rho[0] = 2.0 * a2 * (root - 1.0); rho[1] = std::max(tiny, 1.0 / root); rho[2] = -((1.0 / a2) * rho[1]) / (2.0 * sum);
      return;
    }
    case kCauchyLoss: {  // rho(s) = a^2 log(1 + s / a^2)
      const double a2 = a * a, c = 1.0 / a2, sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = a2 * std::log(sum); rho[1] = std::max(tiny, inv); rho[2] = -c * (inv * inv);
      return;
    }
    case kTukeyLoss: {  // rho(s) = a^2/6 (1 - (1 - s/a^2)^3) for s <= a^2, a^2/6 beyond  (Ceres 1.x scaling)
      const double a2 = a * a;
      if (s <= a2) {
        const double v = 1.0 - s / a2, v2 = v * v;
        rho[0] = a2 / 6.0 * (1.0 - v2 * v); rho[1] = 0.5 * v2; rho[2] = -1.0 / a2 * v;
      } else { rho[0] = a2 / 6.0; rho[1] = 0.0; rho[2] = 0.0; }
      return;
    }
    case kTolerantLoss: {  // rho(s) = b log(1 + exp((s - a) / b)) - b log(1 + exp(-a / b))
      const double c = b * std::log(1.0 + std::exp(-a / b));
      const double x = (s - a) / b;
      if (x > 36.7) { rho[0] = s - a - c; rho[1] = 1.0; rho[2] = 0.0; }
      else {
        const double ex = std::exp(x);
        rho[0] = b * std::log(1.0 + ex) - c; rho[1] = std::max(tiny, ex / (1.0 + ex)); rho[2] = 0.5 / (b * (1.0 + std::cosh(x)));
      }
      return;
    }
    case kComposedLoss: {  // rho(s) = f(g(s))
      double rg[3], rf[3];
      loss_evaluate(nodes, g, s, rg);
      loss_evaluate(nodes, f, rg[0], rf);
      rho[0] = rf[0]; rho[1] = rf[1] * rg[1]; rho[2] = rf[2] * rg[1] * rg[1] + rf[1] * rg[2];
      return;
    }
    case kScaledLoss: {  // rho(s) = a f(s); f absent: a s
      if (f < 0) { rho[0] = a * s; rho[1] = a; rho[2] = 0.0; return; }
      loss_evaluate(nodes, f, s, rho);
      rho[0] *= a; rho[1] *= a; rho[2] *= a;
      return;
    }
    default: rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return;
  }
}

// Corrector: in place on one residual block (nres residuals, Jacobian blocks jac[i] row-major nres x N[i], may be null).
// Returns the block's cost term rho(s).
inline double loss_correct(const double* nodes, int id, int nres, double* res, int nblocks, const int* N, double** jac) {
  double s = 0.0;
  for (int r = 0; r < nres; ++r) s += res[r] * res[r];
  if (id < 0) return s;
  double rho[3];
  loss_evaluate(nodes, id, s, rho);
  const double sqrt_rho1 = std::sqrt(rho[1]);
  double residual_scaling, alpha_sq_norm;
  if (s == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
  else {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - std::sqrt(D);
    residual_scaling = sqrt_rho1 / (1.0 - alpha);
    alpha_sq_norm = alpha / s;
  }
  if (jac) {
    for (int i = 0; i < nblocks; ++i) {
      if (!jac[i]) continue;
      for (int p = 0; p < N[i]; ++p) {
        double rtj = 0.0;
        for (int r = 0; r < nres; ++r) rtj += res[r] * jac[i][r * N[i] + p];
        for (int r = 0; r < nres; ++r) jac[i][r * N[i] + p] = sqrt_rho1 * (jac[i][r * N[i] + p] - alpha_sq_norm * res[r] * rtj);
      }
    }
  }
  for (int r = 0; r < nres; ++r) res[r] *= residual_scaling;
  return rho[0];
}

}  // namespace oracle
