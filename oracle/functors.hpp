// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.hpp header).
//
// CPU restatement of the generic-T functor bodies on the reference's hot path
// and of AutoDiffCostFunction.evaluate.  Paths are relative to /root/reference:
//   CORE = core/src/main/scala/org/somelightprojections/skeres
//   EX   = examples/src/main/scala/org/somelightprojections/skeres/examples
//   TEST = core/src/test/scala/org/somelightprojections/skeres
// Operation ORDER follows the reference expressions exactly; the oracle is
// compiled with -ffp-contract=off so no FMA contraction changes rounding.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include "jet.hpp"

namespace oracle {

// Functor ids — same numbering as include/skeres_amd.h (SK_FUNCTOR_*).
enum FunctorId {
  kSnavelyReprojectionError = 1,  // EX/SimpleBundleAdjuster.scala:79-119
  kExponentialResidual = 2,       // EX/CurveFitting.scala:92-98
  kPowellF1 = 3,                  // EX/Powell.scala:14-21
  kPowellF2 = 4,                  // EX/Powell.scala:24-31 (sqrt(5)*x3 - x4, as written)
  kPowellF3 = 5,                  // EX/Powell.scala:34-42
  kPowellF4 = 6,                  // EX/Powell.scala:45-53
  kBinaryScalarCost = 7,          // TEST/AutodiffCostFuntionSpec.scala:14-26
  kBinaryVector3Cost = 8,         // TEST/AutodiffCostFuntionSpec.scala:55-69
  kTenParameterCost = 9,          // TEST/AutodiffCostFuntionSpec.scala:111-119
  kHelloCostFunctor = 11,         // EX/HelloWorld.scala:11-14
  kQuaternionRotationError = 12,  // no reference counterpart: a 4-parameter block for the local-parameterization tests
};

// ulp(1.0) == 2^-52  (CORE/package.scala:15, CORE/Rotation.scala:457)
static const double kUlpOne = std::numeric_limits<double>::epsilon();

// CORE/Rotation.scala:445-446 — spire ArrayInnerProductSpace.dot: z = 0; z += x(i)*y(i).
template <class T>
inline T dotProduct3(const T* x, const T* y) {
  T z(0.0);
  for (int i = 0; i < 3; ++i) z = z + x[i] * y[i];
  return z;
}

// CORE/Rotation.scala:449-522
template <class T>
inline void angleAxisRotatePoint(const T* angleAxis, const T* pt, T* result) {
  const T theta2 = dotProduct3(angleAxis, angleAxis);
  const T eps(kUlpOne);
  if (theta2 > eps) {  // compares real parts only (package.scala:27)
    const T theta = sqrt(theta2);
    const T cosTheta = cos(theta);
    const T sinTheta = sin(theta);
    const T thetaInverse = 1.0 / theta;
    const T w[3] = {angleAxis[0] * thetaInverse, angleAxis[1] * thetaInverse,
                    angleAxis[2] * thetaInverse};
    const T wCrossPt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                           w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - cosTheta);
    result[0] = pt[0] * cosTheta + wCrossPt[0] * sinTheta + w[0] * tmp;
    result[1] = pt[1] * cosTheta + wCrossPt[1] * sinTheta + w[1] * tmp;
    result[2] = pt[2] * cosTheta + wCrossPt[2] * sinTheta + w[2] * tmp;
  } else {
    const T wCrossPt[3] = {angleAxis[1] * pt[2] - angleAxis[2] * pt[1],
                           angleAxis[2] * pt[0] - angleAxis[0] * pt[2],
                           angleAxis[0] * pt[1] - angleAxis[1] * pt[0]};
    result[0] = pt[0] + wCrossPt[0];
    result[1] = pt[1] + wCrossPt[1];
    result[2] = pt[2] + wCrossPt[2];
  }
}

// CORE/Rotation.scala:206-255 (double only; used by the rotation KAT that the
// reference's RotationSpec.scala:616-655 runs against angleAxisRotatePoint).
// R is row-major 3x3.
inline void angleAxisToRotationMatrix(const double* aa, double* R) {
  static const double kOne = 1.0;
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > kUlpOne) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = std::cos(theta), s = std::sin(theta);
    R[0] = c + wx * wx * (kOne - c);
    R[3] = wz * s + wx * wy * (kOne - c);
    R[6] = -wy * s + wx * wz * (kOne - c);
    R[1] = wx * wy * (kOne - c) - wz * s;
    R[4] = c + wy * wy * (kOne - c);
    R[7] = wx * s + wy * wz * (kOne - c);
    R[2] = wy * s + wx * wz * (kOne - c);
    R[5] = -wx * s + wy * wz * (kOne - c);
    R[8] = c + wz * wz * (kOne - c);
  } else {
    R[0] = kOne;  R[3] = aa[2];  R[6] = -aa[1];
    R[1] = -aa[2]; R[4] = kOne;  R[7] = aa[0];
    R[2] = aa[1]; R[5] = -aa[0]; R[8] = kOne;
  }
}

// ---------------------------------------------------------------------------
// Functor bodies.  Each is  template<class T> bool apply(consts, params, out)
// with `params[i]` the i-th parameter block.  Return false == the reference's
// "empty array" failure convention (CORE/CostFunctor.scala:15-26).
// ---------------------------------------------------------------------------

struct SnavelyReprojectionError {  // EX/SimpleBundleAdjuster.scala:79-119
  static constexpr int kNumResiduals = 2, kNumBlocks = 2, kNumConsts = 2;
  static constexpr int N[2] = {9, 3};
  template <class T>
  static bool apply(const double* c, const T* const* params, T* out) {
    const T* camera = params[0];
    const T* point = params[1];
    T p[3];
    angleAxisRotatePoint(camera, point, p);       // :91-92
    p[0] = p[0] + camera[3];                      // :95-97
    p[1] = p[1] + camera[4];
    p[2] = p[2] + camera[5];
    const T xp = (-p[0]) / p[2];                  // :102-103
    const T yp = (-p[1]) / p[2];
    const T l1 = camera[7];                       // :106-109
    const T l2 = camera[8];
    const T r2 = xp * xp + yp * yp;
    const T distortion = 1.0 + r2 * (l1 + l2 * r2);
    const T focal = camera[6];                    // :112-114
    const T predictedX = focal * distortion * xp;
    const T predictedY = focal * distortion * yp;
    out[0] = predictedX - c[0];                   // :117
    out[1] = predictedY - c[1];
    return true;
  }
};

struct ExponentialResidual {  // EX/CurveFitting.scala:92-98; consts = (x, y)
  static constexpr int kNumResiduals = 1, kNumBlocks = 2, kNumConsts = 2;
  static constexpr int N[2] = {1, 1};
  template <class T>
  static bool apply(const double* c, const T* const* p, T* out) {
    out[0] = c[1] - exp(p[0][0] * c[0] + p[1][0]);
    return true;
  }
};

struct PowellF1 {  // EX/Powell.scala:14-21
  static constexpr int kNumResiduals = 1, kNumBlocks = 2, kNumConsts = 0;
  static constexpr int N[2] = {1, 1};
  template <class T>
  static bool apply(const double*, const T* const* x, T* out) {
    out[0] = x[0][0] + 10.0 * x[1][0];
    return true;
  }
};
struct PowellF2 {  // EX/Powell.scala:24-31: sqrt(5.0) * x3(0) - x4(0)
  static constexpr int kNumResiduals = 1, kNumBlocks = 2, kNumConsts = 0;
  static constexpr int N[2] = {1, 1};
  template <class T>
  static bool apply(const double*, const T* const* x, T* out) {
    out[0] = std::sqrt(5.0) * x[0][0] - x[1][0];
    return true;
  }
};
struct PowellF3 {  // EX/Powell.scala:34-42
  static constexpr int kNumResiduals = 1, kNumBlocks = 2, kNumConsts = 0;
  static constexpr int N[2] = {1, 1};
  template <class T>
  static bool apply(const double*, const T* const* x, T* out) {
    const T d = x[0][0] - 2.0 * x[1][0];
    out[0] = d * d;
    return true;
  }
};
struct PowellF4 {  // EX/Powell.scala:45-53: sqrt(10) * d * d
  static constexpr int kNumResiduals = 1, kNumBlocks = 2, kNumConsts = 0;
  static constexpr int N[2] = {1, 1};
  template <class T>
  static bool apply(const double*, const T* const* x, T* out) {
    const T d = x[0][0] - x[1][0];
    out[0] = std::sqrt(10.0) * d * d;
    return true;
  }
};

struct BinaryScalarCost {  // TEST/AutodiffCostFuntionSpec.scala:14-26; consts = (a)
  static constexpr int kNumResiduals = 1, kNumBlocks = 2, kNumConsts = 1;
  static constexpr int N[2] = {2, 2};
  template <class T>
  static bool apply(const double* c, const T* const* p, T* out) {
    const T* x = p[0]; const T* y = p[1];
    out[0] = x[0] * y[0] + x[1] * y[1] - c[0];
    return true;
  }
};
struct BinaryVector3Cost {  // TEST/AutodiffCostFuntionSpec.scala:55-69; consts = (a)
  static constexpr int kNumResiduals = 3, kNumBlocks = 2, kNumConsts = 1;
  static constexpr int N[2] = {2, 2};
  template <class T>
  static bool apply(const double* c, const T* const* p, T* out) {
    const T* x = p[0]; const T* y = p[1];
    out[0] = x[0] * y[0] + x[1] * y[1] - c[0];
    out[1] = x[0] * y[0] - x[1] * y[1] + c[0];
    out[2] = x[0] * x[1] + y[0] * y[1] + 10.0 * c[0];
    return true;
  }
};
struct TenParameterCost {  // TEST/AutodiffCostFuntionSpec.scala:111-119
  static constexpr int kNumResiduals = 1, kNumBlocks = 10, kNumConsts = 0;
  static constexpr int N[10] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  template <class T>
  static bool apply(const double*, const T* const* p, T* out) {
    T s = p[0][0];  // p.map(_(0)).reduce(_ + _): left fold
    for (int i = 1; i < 10; ++i) s = s + p[i][0];
    out[0] = s;
    return true;
  }
};

struct HelloCostFunctor {  // EX/HelloWorld.scala:11-14
  static constexpr int kNumResiduals = 1, kNumBlocks = 1, kNumConsts = 0;
  static constexpr int N[1] = {1};
  template <class T>
  static bool apply(const double*, const T* const* p, T* out) {
    out[0] = 10.0 - p[0][0];
    return true;
  }
};

// r = R(q) p - t for a quaternion block q = (w, x, y, z) (normalised first: Rotation.quaternionRotatePoint,
// CORE/Rotation.scala:393-430, spelled out); consts = (p[3], t[3]).  Exists so that the quaternion / homogeneous-vector
// parameterizations have a registered functor with a block of size 4 to be tested on.
struct QuaternionRotationError {
  static constexpr int kNumResiduals = 3, kNumBlocks = 1, kNumConsts = 6;
  static constexpr int N[1] = {4};
  template <class T>
  static bool apply(const double* c, const T* const* p, T* out) {
    const T* q = p[0];
    const T scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const T a = q[0] * scale, b = q[1] * scale, cc = q[2] * scale, d = q[3] * scale;
    const T t2 = a * b, t3 = a * cc, t4 = a * d, t5 = -(b * b), t6 = b * cc, t7 = b * d, t8 = -(cc * cc), t9 = cc * d, t1 = -(d * d);
    out[0] = 2.0 * ((t8 + t1) * c[0] + (t6 - t4) * c[1] + (t3 + t7) * c[2]) + c[0] - c[3];
    out[1] = 2.0 * ((t4 + t6) * c[0] + (t5 + t1) * c[1] + (t9 - t2) * c[2]) + c[1] - c[4];
    out[2] = 2.0 * ((t7 - t3) * c[0] + (t2 + t9) * c[1] + (t5 + t8) * c[2]) + c[2] - c[5];
    return true;
  }
};

// ---------------------------------------------------------------------------
// BASELINE.json config 5 (synthetic dense problem, no reference counterpart):
//   r(x) = tanh(a . x) - y over ONE block of any size n, a_j = unit(seed, row, j) / sqrt(n) with
//   unit() the counter-based draw documented in skeres_amd/csrc/synth.hpp (restated here).
// Evaluated with Jet semantics: the Jet of u = a . x has infinitesimal part a, and
// tanh(u, v) = (tanh u, (1 - tanh^2 u) v)  [spire / Ceres jet rule].
// ---------------------------------------------------------------------------
static const int kSynthTanhRow = 10;
inline uint64_t synth_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double synth_unit(uint64_t seed, uint64_t i, uint64_t n, uint64_t j) {
  const uint64_t h = synth_mix64(seed ^ ((i * n + j) * 0xD6E8FEB86659FD93ull));
  const double s = (double)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + ((h >> 48) & 0xffff)) + 2.0;
  return (s * (1.0 / 65536.0) - 2.0) * 1.7320508075688772;
}
// jac_row may be null (cost-only branch)
inline bool synth_tanh_row_evaluate(const double* c, const double* x, int n, double* residual, double* jac_row) {
  const uint64_t seed = (uint64_t)c[0], row = (uint64_t)c[1];
  const double inv_sqrt_n = 1.0 / std::sqrt((double)n);
  double u = 0.0;
  for (int j = 0; j < n; ++j) u += synth_unit(seed, row, (uint64_t)n, (uint64_t)j) * x[j];
  const double t = std::tanh(u * inv_sqrt_n);
  residual[0] = t - c[2];
  if (jac_row) {
    const double sd = (1.0 - t * t) * inv_sqrt_n;
    for (int j = 0; j < n; ++j) jac_row[j] = sd * synth_unit(seed, row, (uint64_t)n, (uint64_t)j);
  }
  return true;
}

// ---------------------------------------------------------------------------
// AutoDiffCostFunction.evaluate  (CORE/AutodiffCostFunction.scala:74-134)
//   parameters : double const* const*   (one pointer per block)
//   residuals  : double[kNumResiduals]
//   jacobians  : double** or nullptr; jacobians[i] may itself be nullptr
//                (:80, :118).  Block i is row-major kNumResiduals x N(i).
// ---------------------------------------------------------------------------
template <class F>
struct AutoDiff {
  static constexpr int kBlocks = F::kNumBlocks;
  static constexpr int kRes = F::kNumResiduals;
  static constexpr int jetDim() { int s = 0; for (int i = 0; i < kBlocks; ++i) s += F::N[i]; return s; }
  static constexpr int kJetDim = jetDim();  // JetDim(costFunctor.N.sum)  (:72)

  static bool evaluate(const double* consts, double const* const* parameters,
                       double* residuals, double** jacobians) {
    if (jacobians == nullptr) {  // cost-only branch (:80-93), T = Double
      const double* x[kBlocks];
      for (int i = 0; i < kBlocks; ++i) x[i] = parameters[i];
      double y[kRes];
      if (!F::template apply<double>(consts, x, y)) return false;
      for (int r = 0; r < kRes; ++r) residuals[r] = y[r];
      return true;
    }
    typedef Jet<kJetDim> J;
    J store[kJetDim];
    const J* jx[kBlocks];
    int k = 0;
    for (int i = 0; i < kBlocks; ++i) {  // seeding order (:96-106)
      jx[i] = &store[k];
      for (int j = 0; j < F::N[i]; ++j) { store[k] = J(parameters[i][j], k); ++k; }
    }
    J jy[kRes];
    if (!F::template apply<J>(consts, jx, jy)) return false;  // (:108-111)
    for (int r = 0; r < kRes; ++r) residuals[r] = jy[r].a;    // (:113)
    int parBlockOffset = 0;
    for (int i = 0; i < kBlocks; ++i) {  // (:115-130)
      const int ni = F::N[i];
      if (jacobians[i] != nullptr) {
        int col = 0;
        for (int r = 0; r < kRes; ++r)
          for (int p = 0; p < ni; ++p) jacobians[i][col++] = jy[r].v[parBlockOffset + p];
      }
      parBlockOffset += ni;
    }
    return true;
  }
};

}  // namespace oracle
