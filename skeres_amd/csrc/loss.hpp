// Robust loss functions rho(s), s = |r|^2 of one residual block, and the
// residual / Jacobian correction that folds them into a least-squares problem.
//
// The reference exposes native Ceres' predefined losses through
// PredefinedLossFunctions (ceres.i:159-184: trivial, huber, softLOne, cauchy,
// tukey, tolerant, composed, scaled) and passes them to
// Problem.addResidualBlock (CORE/Problem.scala:20-27; EX/RobustCurveFitting.scala:107).
// Ceres itself is not vendored in the reference, so the formulas below restate
// the published ceres::LossFunction::Evaluate contracts (rho, rho', rho'') of
// Ceres 1.x and its Corrector (Triggs et al.): recalled, unverified against a
// Ceres build — see DESIGN.md §2.
#pragma once
#include <hip/hip_runtime.h>

namespace sk {

enum LossType : int {
  kLossTrivial = 0, kLossHuber = 1, kLossSoftLOne = 2, kLossCauchy = 3, kLossTukey = 4, kLossTolerant = 5, kLossComposed = 6, kLossScaled = 7
};

// One node of a flattened loss expression: children come before their parent.
struct LossNode {
  int type;
  int f, g;     // children (composed: rho = f(g(s)); scaled: f), -1 = none (scaled: the NULL loss, rho = a s)
  int depth;    // nesting depth below this node (leaves: 0)
  double a, b;
};

constexpr int kLossMaxDepth = 4;                 // composed / scaled nesting accepted by sk_loss_composed / sk_loss_scaled
constexpr double kLossMinPositive = 2.2250738585072014e-308;  // std::numeric_limits<double>::min()

__host__ __device__ inline double loss_max(double x, double y) { return x > y ? x : y; }

// rho[0..2] = rho(s), rho'(s), rho''(s) of a leaf
__host__ __device__ inline void loss_leaf(const LossNode& n, double s, double rho[3]) {
  switch (n.type) {
    case kLossHuber: {
      const double b = n.a * n.a;
      if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * n.a * r - b;
        rho[1] = loss_max(kLossMinPositive, n.a / r);
        rho[2] = -rho[1] / (2.0 * s);
      } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
      return;
    }
    case kLossSoftLOne: {
      const double b = n.a * n.a, c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0);
      rho[1] = loss_max(kLossMinPositive, 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      return;
    }
    case kLossCauchy: {
      const double b = n.a * n.a, c = 1.0 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * log(sum);
      rho[1] = loss_max(kLossMinPositive, inv);
      rho[2] = -c * (inv * inv);
      return;
    }
    case kLossTukey: {
      const double a2 = n.a * n.a;
      if (s <= a2) {
        const double v = 1.0 - s / a2, v2 = v * v;
        rho[0] = a2 / 6.0 * (1.0 - v2 * v);
        rho[1] = 0.5 * v2;
        rho[2] = -1.0 / a2 * v;
      } else { rho[0] = a2 / 6.0; rho[1] = 0.0; rho[2] = 0.0; }
      return;
    }
    case kLossTolerant: {
      const double a = n.a, b = n.b;
      const double c = b * log(1.0 + exp(-a / b));
      const double x = (s - a) / b;
      const double kLog2Pow53 = 36.7;  // ln(2^53): beyond it exp(x) swamps the 1
      if (x > kLog2Pow53) { rho[0] = s - a - c; rho[1] = 1.0; rho[2] = 0.0; }
      else {
        const double ex = exp(x);
        rho[0] = b * log(1.0 + ex) - c;
        rho[1] = loss_max(kLossMinPositive, ex / (1.0 + ex));
        rho[2] = 0.5 / (b * (1.0 + cosh(x)));
      }
      return;
    }
    default: rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return;
  }
}

// Evaluate node `id` of `nodes` at s.  kDepth bounds the recursion at compile time.
template <int kDepth>
__host__ __device__ inline void loss_evaluate_d(const LossNode* nodes, int id, double s, double rho[3]) {
  if (id < 0) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; }
  const LossNode n = nodes[id];
  if (n.type == kLossComposed) {
    double rg[3], rf[3];
    loss_evaluate_d<kDepth - 1>(nodes, n.g, s, rg);
    loss_evaluate_d<kDepth - 1>(nodes, n.f, rg[0], rf);
    rho[0] = rf[0];
    rho[1] = rf[1] * rg[1];
    rho[2] = rf[2] * rg[1] * rg[1] + rf[1] * rg[2];
  } else if (n.type == kLossScaled) {
    if (n.f < 0) { rho[0] = n.a * s; rho[1] = n.a; rho[2] = 0.0; return; }
    loss_evaluate_d<kDepth - 1>(nodes, n.f, s, rho);
    rho[0] *= n.a; rho[1] *= n.a; rho[2] *= n.a;
  } else {
    loss_leaf(n, s, rho);
  }
}
template <>
__host__ __device__ inline void loss_evaluate_d<0>(const LossNode* nodes, int id, double s, double rho[3]) {
  if (id < 0) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; }
  loss_leaf(nodes[id], s, rho);  // creation rejects deeper nesting, so a node reached here is a leaf
}
__host__ __device__ inline void loss_evaluate(const LossNode* nodes, int id, double s, double rho[3]) {
  loss_evaluate_d<kLossMaxDepth>(nodes, id, s, rho);
}

// Triggs correction (ceres Corrector): with the corrected residuals/Jacobian the Gauss-Newton model of
// 1/2 |r~|^2 matches the second-order model of 1/2 rho(|r|^2).
//   r~ = residual_scaling r ;  J~ = sqrt(rho') (J - alpha_sq_norm r (r^T J))
struct LossCorrector {
  double sqrt_rho1, residual_scaling, alpha_sq_norm;
  __host__ __device__ LossCorrector(double sq_norm, const double rho[3]) {
    sqrt_rho1 = sqrt(rho[1]);
    if (sq_norm == 0.0 || rho[2] <= 0.0) {
      residual_scaling = sqrt_rho1;
      alpha_sq_norm = 0.0;
    } else {
      const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
      const double alpha = 1.0 - sqrt(D);
      residual_scaling = sqrt_rho1 / (1.0 - alpha);
      alpha_sq_norm = alpha / sq_norm;
    }
  }
};

}  // namespace sk
