// TEST INFRASTRUCTURE (oracle): CPU restatement of the Ceres 1.x local parameterizations the reference exposes through
// PredefinedLocalParameterizations (ceres.i:186-210: identity, subset, quaternion, homogeneousVector), plus "constant"
// for Problem::SetParameterBlockConstant.  Ceres is an un-vendored dependency of the reference: restated from the
// published contracts (local_parameterization.h/.cc, householder_vector.h of Ceres 1.12-1.14); PARITY UNPINNED — the
// reference holds no test or fixture for them.  tests/ additionally check every Jacobian against central differences
// of Plus, which is independent of both this file and the device code.
//
// Written matrix-first (explicit Householder matrix, explicit quaternion product), unlike the device code's
// in-register loops: two routes to the same numbers.
#pragma once
#include <cmath>
#include <vector>

namespace oracle {

enum { P_IDENTITY = 0, P_SUBSET = 1, P_QUATERNION = 2, P_HOMOGENEOUS = 3, P_CONSTANT = 4 };

struct Parameterization {
  int type = P_IDENTITY, global_size = 0;
  std::vector<int> constant;  // subset: indices held constant
  int local_size() const {
    switch (type) {
      case P_SUBSET: return global_size - (int)constant.size();
      case P_QUATERNION: return 3;
      case P_HOMOGENEOUS: return global_size - 1;
      case P_CONSTANT: return 0;
      default: return global_size;
    }
  }
};

inline void quaternion_product(const double* z, const double* w, double* zw) {  // ceres/rotation.h QuaternionProduct
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

// H = I - beta v v^T with H x = |x| e_n, as a dense n x n matrix (row-major)
inline std::vector<double> householder_matrix(const double* x, int n) {
  std::vector<double> v(x, x + n);
  double sigma = 0.0, beta = 0.0;
  for (int i = 0; i + 1 < n; ++i) sigma += x[i] * x[i];
  v[n - 1] = 1.0;
  const double pivot = x[n - 1];
  if (sigma <= 2.220446049250313e-16) {
    if (pivot < 0.0) beta = 2.0;
  } else {
    const double mu = std::sqrt(pivot * pivot + sigma);
    const double vp = pivot <= 0.0 ? pivot - mu : -sigma / (pivot + mu);
    beta = 2.0 * vp * vp / (sigma + vp * vp);
    for (int i = 0; i + 1 < n; ++i) v[i] /= vp;
  }
  std::vector<double> H((size_t)n * n);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) H[(size_t)r * n + c] = (r == c ? 1.0 : 0.0) - beta * v[r] * v[c];
  return H;
}

inline void parameterization_plus(const Parameterization& p, const double* x, const double* delta, double* out) {
  const int n = p.global_size;
  if (p.type == P_CONSTANT) { for (int i = 0; i < n; ++i) out[i] = x[i]; return; }
  if (p.type == P_IDENTITY) { for (int i = 0; i < n; ++i) out[i] = x[i] + delta[i]; return; }
  if (p.type == P_SUBSET) {
    std::vector<char> fixed(n, 0);
    for (int c : p.constant) fixed[c] = 1;
    int l = 0;
    for (int i = 0; i < n; ++i) out[i] = fixed[i] ? x[i] : x[i] + delta[l++];
    return;
  }
  if (p.type == P_QUATERNION) {
    const double nd = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (nd > 0.0) {
      const double k = std::sin(nd) / nd;
      const double q[4] = {std::cos(nd), k * delta[0], k * delta[1], k * delta[2]};
      quaternion_product(q, x, out);
    } else {
      for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
    return;
  }
  // homogeneous vector: x_plus = |x| H y, y = [sin(|d|/2) d/|d| ; cos(|d|/2)]
  double sq = 0.0;
  for (int i = 0; i + 1 < n; ++i) sq += delta[i] * delta[i];
  if (sq == 0.0) { for (int i = 0; i < n; ++i) out[i] = x[i]; return; }
  const double nd = std::sqrt(sq);
  std::vector<double> y(n);
  for (int i = 0; i + 1 < n; ++i) y[i] = std::sin(0.5 * nd) / nd * delta[i];
  y[n - 1] = std::cos(0.5 * nd);
  const std::vector<double> H = householder_matrix(x, n);
  double xn = 0.0;
  for (int i = 0; i < n; ++i) xn += x[i] * x[i];
  xn = std::sqrt(xn);
  for (int r = 0; r < n; ++r) {
    double s = 0.0;
    for (int c = 0; c < n; ++c) s += H[(size_t)r * n + c] * y[c];
    out[r] = xn * s;
  }
}

// J (global x local, row-major) = d Plus(x, delta) / d delta at delta = 0
inline void parameterization_jacobian(const Parameterization& p, const double* x, double* J) {
  const int n = p.global_size, l = p.local_size();
  for (int i = 0; i < n * l; ++i) J[i] = 0.0;
  if (p.type == P_CONSTANT) return;
  if (p.type == P_IDENTITY) { for (int i = 0; i < n; ++i) J[i * l + i] = 1.0; return; }
  if (p.type == P_SUBSET) {
    std::vector<char> fixed(n, 0);
    for (int c : p.constant) fixed[c] = 1;
    int col = 0;
    for (int i = 0; i < n; ++i) if (!fixed[i]) J[i * l + col++] = 1.0;
    return;
  }
  if (p.type == P_QUATERNION) {
    const double rows[12] = {-x[1], -x[2], -x[3], x[0], x[3], -x[2], -x[3], x[0], x[1], x[2], -x[1], x[0]};
    for (int i = 0; i < 12; ++i) J[i] = rows[i];
    return;
  }
  // homogeneous vector: the first n-1 columns of 0.5 |x| H  (dy/d delta at 0 is 0.5 [I; 0])
  const std::vector<double> H = householder_matrix(x, n);
  double xn = 0.0;
  for (int i = 0; i < n; ++i) xn += x[i] * x[i];
  xn = std::sqrt(xn);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < l; ++c) J[r * l + c] = 0.5 * xn * H[(size_t)r * n + c];
}

}  // namespace oracle
