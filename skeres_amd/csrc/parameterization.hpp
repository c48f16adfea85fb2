// Local parameterizations: x (+) delta on a manifold, and d(x (+) delta)/d(delta) at delta = 0.
//
// The reference exposes native Ceres' predefined ones through PredefinedLocalParameterizations (ceres.i:186-210:
// identity, subset, quaternion, homogeneousVector) for ceres::Problem::AddParameterBlock / SetParameterization, which
// its Problem inherits (CORE/Problem.scala:16).  Ceres itself is not vendored in the reference, so the formulas below
// restate the published ceres::LocalParameterization contracts of Ceres 1.x (local_parameterization.cc,
// internal/ceres/householder_vector.h): recalled, unverified against a Ceres build — DESIGN.md section 2.
// A constant parameter block (Problem::SetParameterBlockConstant) is kParamConstant: local size 0.
#pragma once
#include <hip/hip_runtime.h>

namespace sk {

enum ParameterizationType : int { kParamIdentity = 0, kParamSubset = 1, kParamQuaternion = 2, kParamHomogeneousVector = 3, kParamConstant = 4 };
constexpr int kParamMaxSize = 16;  // global size of a parameterized block (the Jacobian lives in registers / on the stack)

// One parameter block's parameterization, flattened for the device: `constant_mask` bit i set = coordinate i is held
// constant (subset).
struct ParamBlock {
  int type, global_size, local_size;
  unsigned constant_mask;
  int global_off, local_off;  // of the block in x and in the tangent vector
};

// v, beta with (I - beta v v^T) x = |x| e_n (internal/ceres/householder_vector.h)
__host__ __device__ inline void householder_vector(const double* x, int n, double* v, double* beta) {
  double sigma = 0.0;
  for (int i = 0; i < n - 1; ++i) sigma += x[i] * x[i];
  for (int i = 0; i < n; ++i) v[i] = x[i];
  v[n - 1] = 1.0;
  *beta = 0.0;
  const double x_pivot = x[n - 1];
  if (sigma <= 2.220446049250313e-16) {
    if (x_pivot < 0.0) *beta = 2.0;
    return;
  }
  const double mu = sqrt(x_pivot * x_pivot + sigma);
  double v_pivot = 1.0;
  if (x_pivot <= 0.0) v_pivot = x_pivot - mu; else v_pivot = -sigma / (x_pivot + mu);
  *beta = 2.0 * v_pivot * v_pivot / (sigma + v_pivot * v_pivot);
  for (int i = 0; i < n - 1; ++i) v[i] /= v_pivot;
}

// x_plus = x (+) delta; x, x_plus: global_size values, delta: local_size values
__host__ __device__ inline void param_plus(const ParamBlock& p, const double* x, const double* delta, double* x_plus) {
  const int n = p.global_size;
  switch (p.type) {
    case kParamSubset: {
      int l = 0;
      for (int i = 0; i < n; ++i) x_plus[i] = ((p.constant_mask >> i) & 1u) ? x[i] : x[i] + delta[l++];
      return;
    }
    case kParamQuaternion: {
      const double norm_delta = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
      if (norm_delta > 0.0) {
        const double s = sin(norm_delta) / norm_delta;
        const double q[4] = {cos(norm_delta), s * delta[0], s * delta[1], s * delta[2]};
        // QuaternionProduct(q_delta, x)
        x_plus[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
        x_plus[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
        x_plus[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
        x_plus[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
      } else {
        for (int i = 0; i < 4; ++i) x_plus[i] = x[i];
      }
      return;
    }
    case kParamHomogeneousVector: {
      double sq = 0.0;
      for (int i = 0; i < n - 1; ++i) sq += delta[i] * delta[i];
      if (sq == 0.0) { for (int i = 0; i < n; ++i) x_plus[i] = x[i]; return; }
      const double norm_delta = sqrt(sq), half = 0.5 * norm_delta;
      const double sin_by = sin(half) / half;  // y = [0.5 sin(|d|/2)/(|d|/2) d ; cos(|d|/2)]
      double y[kParamMaxSize], v[kParamMaxSize], beta, xn = 0.0, vy = 0.0;
      for (int i = 0; i < n - 1; ++i) y[i] = 0.5 * sin_by * delta[i];
      y[n - 1] = cos(half);
      householder_vector(x, n, v, &beta);
      for (int i = 0; i < n; ++i) { xn += x[i] * x[i]; vy += v[i] * y[i]; }
      xn = sqrt(xn);
      for (int i = 0; i < n; ++i) x_plus[i] = xn * (y[i] - v[i] * (beta * vy));
      return;
    }
    case kParamConstant:
      for (int i = 0; i < n; ++i) x_plus[i] = x[i];
      return;
    default:
      for (int i = 0; i < n; ++i) x_plus[i] = x[i] + delta[i];
      return;
  }
}

// J: global_size x local_size, row-major
__host__ __device__ inline void param_jacobian(const ParamBlock& p, const double* x, double* J) {
  const int n = p.global_size, l = p.local_size;
  for (int i = 0; i < n * l; ++i) J[i] = 0.0;
  switch (p.type) {
    case kParamSubset: {
      int c = 0;
      for (int i = 0; i < n; ++i) if (!((p.constant_mask >> i) & 1u)) J[i * l + c++] = 1.0;
      return;
    }
    case kParamQuaternion:
      J[0] = -x[1]; J[1] = -x[2]; J[2] = -x[3];
      J[3] = x[0];  J[4] = x[3];  J[5] = -x[2];
      J[6] = -x[3]; J[7] = x[0];  J[8] = x[1];
      J[9] = x[2];  J[10] = -x[1]; J[11] = x[0];
      return;
    case kParamHomogeneousVector: {
      double v[kParamMaxSize], beta, xn = 0.0;
      householder_vector(x, n, v, &beta);
      for (int i = 0; i < n; ++i) xn += x[i] * x[i];
      xn = sqrt(xn);
      for (int c = 0; c < n - 1; ++c) {
        for (int r = 0; r < n; ++r) J[r * l + c] = -0.5 * beta * v[c] * v[r];
        J[c * l + c] += 0.5;
        for (int r = 0; r < n; ++r) J[r * l + c] *= xn;
      }
      return;
    }
    case kParamConstant:
      return;
    default:
      for (int i = 0; i < n; ++i) J[i * l + i] = 1.0;
      return;
  }
}

}  // namespace sk
