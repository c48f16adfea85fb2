// Device functor registry: the generic-T functor bodies of the reference's hot
// path, written once and instantiated for T = double (cost-only) and
// T = Jet<N> (Jacobian).  In the reference these are JVM closures reached by a
// SWIG director upcall per residual block (ceres.i:48); here they are device
// code addressed by sk_functor_id (include/skeres_amd.h), with the closure's
// captured doubles passed as `c`.
//   EX   = examples/src/main/scala/org/somelightprojections/skeres/examples
//   CORE = core/src/main/scala/org/somelightprojections/skeres
//   TEST = core/src/test/scala/org/somelightprojections/skeres
#pragma once
#include "jet.hpp"

namespace sk {

// Math.ulp(1.0)  (CORE/Rotation.scala:457)
#define SK_ULP_ONE 2.220446049250313e-16

// Rotation.angleAxisRotatePoint  (CORE/Rotation.scala:449-522); dotProduct :445-446
template <class T>
SK_HD void angle_axis_rotate_point(const T* aa, const T* pt, T* out) {
  const T theta2 = (aa[0] * aa[0] + aa[1] * aa[1]) + aa[2] * aa[2];
  if (jgt(theta2, SK_ULP_ONE)) {  // Order.by(_.real): real parts only
    const T theta = jsqrt(theta2);
    T s, c;
    jsincos(theta, &s, &c);
    const T ti = 1.0 / theta;
    const T w0 = aa[0] * ti, w1 = aa[1] * ti, w2 = aa[2] * ti;
    const T x0 = w1 * pt[2] - w2 * pt[1];
    const T x1 = w2 * pt[0] - w0 * pt[2];
    const T x2 = w0 * pt[1] - w1 * pt[0];
    const T tmp = ((w0 * pt[0] + w1 * pt[1]) + w2 * pt[2]) * (1.0 - c);
    out[0] = (pt[0] * c + x0 * s) + w0 * tmp;
    out[1] = (pt[1] * c + x1 * s) + w1 * tmp;
    out[2] = (pt[2] * c + x2 * s) + w2 * tmp;
  } else {
    out[0] = pt[0] + (aa[1] * pt[2] - aa[2] * pt[1]);
    out[1] = pt[1] + (aa[2] * pt[0] - aa[0] * pt[2]);
    out[2] = pt[2] + (aa[0] * pt[1] - aa[1] * pt[0]);
  }
}

// Every functor: static sizes + template<class T> bool apply(c, params, out).
// `params` is an array of pointers, one per parameter block.  false == the
// reference's "empty result" failure (CORE/CostFunctor.scala:15-26).

struct SnavelyReprojectionError {  // EX/SimpleBundleAdjuster.scala:79-119
  static constexpr int kRes = 2, kBlocks = 2, kConsts = 2, kDim = 12;
  static SK_HD int N(int i) { return i == 0 ? 9 : 3; }
  template <class T>
  static SK_HD bool apply(const double* c, const T* const* params, T* out) {
    const T* cam = params[0];
    const T* X = params[1];
    T p[3];
    angle_axis_rotate_point(cam, X, p);
    p[0] = p[0] + cam[3];
    p[1] = p[1] + cam[4];
    p[2] = p[2] + cam[5];
    const T xp = (-p[0]) / p[2];
    const T yp = (-p[1]) / p[2];
    const T r2 = xp * xp + yp * yp;
    const T distortion = 1.0 + r2 * (cam[7] + cam[8] * r2);
    const T fd = cam[6] * distortion;
    out[0] = fd * xp - c[0];
    out[1] = fd * yp - c[1];
    return true;
  }
};

struct ExponentialResidual {  // EX/CurveFitting.scala:92-98; c = (x, y)
  static constexpr int kRes = 1, kBlocks = 2, kConsts = 2, kDim = 2;
  static SK_HD int N(int) { return 1; }
  template <class T>
  static SK_HD bool apply(const double* c, const T* const* p, T* out) {
    out[0] = c[1] - jexp(p[0][0] * c[0] + p[1][0]);
    return true;
  }
};

struct PowellF1 {  // EX/Powell.scala:14-21
  static constexpr int kRes = 1, kBlocks = 2, kConsts = 0, kDim = 2;
  static SK_HD int N(int) { return 1; }
  template <class T>
  static SK_HD bool apply(const double*, const T* const* x, T* out) {
    out[0] = x[0][0] + 10.0 * x[1][0];
    return true;
  }
};
struct PowellF2 {  // EX/Powell.scala:24-31 (as written: sqrt(5)*x3 - x4)
  static constexpr int kRes = 1, kBlocks = 2, kConsts = 0, kDim = 2;
  static SK_HD int N(int) { return 1; }
  template <class T>
  static SK_HD bool apply(const double*, const T* const* x, T* out) {
    out[0] = 2.23606797749979 * x[0][0] - x[1][0];  // sqrt(5.0)
    return true;
  }
};
struct PowellF3 {  // EX/Powell.scala:34-42
  static constexpr int kRes = 1, kBlocks = 2, kConsts = 0, kDim = 2;
  static SK_HD int N(int) { return 1; }
  template <class T>
  static SK_HD bool apply(const double*, const T* const* x, T* out) {
    const T d = x[0][0] - 2.0 * x[1][0];
    out[0] = d * d;
    return true;
  }
};
struct PowellF4 {  // EX/Powell.scala:45-53
  static constexpr int kRes = 1, kBlocks = 2, kConsts = 0, kDim = 2;
  static SK_HD int N(int) { return 1; }
  template <class T>
  static SK_HD bool apply(const double*, const T* const* x, T* out) {
    const T d = x[0][0] - x[1][0];
    out[0] = (3.1622776601683795 * d) * d;  // sqrt(10) * d * d
    return true;
  }
};

struct BinaryScalarCost {  // TEST/AutodiffCostFuntionSpec.scala:14-26; c = (a)
  static constexpr int kRes = 1, kBlocks = 2, kConsts = 1, kDim = 4;
  static SK_HD int N(int) { return 2; }
  template <class T>
  static SK_HD bool apply(const double* c, const T* const* p, T* out) {
    out[0] = (p[0][0] * p[1][0] + p[0][1] * p[1][1]) - c[0];
    return true;
  }
};
struct BinaryVector3Cost {  // TEST/AutodiffCostFuntionSpec.scala:55-69; c = (a)
  static constexpr int kRes = 3, kBlocks = 2, kConsts = 1, kDim = 4;
  static SK_HD int N(int) { return 2; }
  template <class T>
  static SK_HD bool apply(const double* c, const T* const* p, T* out) {
    const T* x = p[0];
    const T* y = p[1];
    out[0] = (x[0] * y[0] + x[1] * y[1]) - c[0];
    out[1] = (x[0] * y[0] - x[1] * y[1]) + c[0];
    out[2] = (x[0] * x[1] + y[0] * y[1]) + 10.0 * c[0];
    return true;
  }
};
struct TenParameterCost {  // TEST/AutodiffCostFuntionSpec.scala:111-119
  static constexpr int kRes = 1, kBlocks = 10, kConsts = 0, kDim = 10;
  static SK_HD int N(int) { return 1; }
  template <class T>
  static SK_HD bool apply(const double*, const T* const* p, T* out) {
    T s = p[0][0];
#pragma unroll
    for (int i = 1; i < 10; ++i) s = s + p[i][0];
    out[0] = s;
    return true;
  }
};

struct HelloCostFunctor {  // EX/HelloWorld.scala:11-14: 10 - x
  static constexpr int kRes = 1, kBlocks = 1, kConsts = 0, kDim = 1;
  static SK_HD int N(int) { return 1; }
  template <class T>
  static SK_HD bool apply(const double*, const T* const* p, T* out) {
    out[0] = 10.0 - p[0][0];
    return true;
  }
};

// r = R(q) p - t for a quaternion block q = (w, x, y, z), normalised first (Rotation.quaternionRotatePoint,
// CORE/Rotation.scala:393-430, spelled out); c = (p[3], t[3]).  No reference counterpart: a registered functor
// with a block of size 4, for the local parameterizations (quaternion, homogeneous vector) to act on.
struct QuaternionRotationError {
  static constexpr int kRes = 3, kBlocks = 1, kConsts = 6, kDim = 4;
  static SK_HD int N(int) { return 4; }
  template <class T>
  static SK_HD bool apply(const double* c, const T* const* p, T* out) {
    const T* q = p[0];
    const T scale = 1.0 / jsqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    const T a = q[0] * scale, b = q[1] * scale, cc = q[2] * scale, d = q[3] * scale;
    const T t2 = a * b, t3 = a * cc, t4 = a * d, t5 = -(b * b), t6 = b * cc, t7 = b * d, t8 = -(cc * cc), t9 = cc * d, t1 = -(d * d);
    out[0] = (2.0 * (((t8 + t1) * c[0] + (t6 - t4) * c[1]) + (t3 + t7) * c[2]) + c[0]) - c[3];
    out[1] = (2.0 * (((t4 + t6) * c[0] + (t5 + t1) * c[1]) + (t9 - t2) * c[2]) + c[1]) - c[4];
    out[2] = (2.0 * (((t7 - t3) * c[0] + (t2 + t9) * c[1]) + (t5 + t8) * c[2]) + c[2]) - c[5];
    return true;
  }
};

// Static description usable on the host (sizes for validation).
struct FunctorDesc {
  int id, num_residuals, num_blocks, num_consts;
  int block_sizes[10];
};
inline bool functor_desc(int id, FunctorDesc* d) {
#define SK_DESC(ID, F)                                                   \
  case ID:                                                               \
    d->id = ID; d->num_residuals = F::kRes; d->num_blocks = F::kBlocks;  \
    d->num_consts = F::kConsts;                                          \
    for (int i = 0; i < F::kBlocks; ++i) d->block_sizes[i] = F::N(i);    \
    return true;
  switch (id) {
    SK_DESC(1, SnavelyReprojectionError)
    SK_DESC(2, ExponentialResidual)
    SK_DESC(3, PowellF1)
    SK_DESC(4, PowellF2)
    SK_DESC(5, PowellF3)
    SK_DESC(6, PowellF4)
    SK_DESC(7, BinaryScalarCost)
    SK_DESC(8, BinaryVector3Cost)
    SK_DESC(9, TenParameterCost)
    SK_DESC(11, HelloCostFunctor)
    SK_DESC(12, QuaternionRotationError)
  }
#undef SK_DESC
  return false;
}

// Dispatch a generic lambda-like functor `OP` templated on the functor type.
#define SK_DISPATCH_FUNCTOR(id, MACRO)            \
  switch (id) {                                   \
    case 1: MACRO(sk::SnavelyReprojectionError); break; \
    case 2: MACRO(sk::ExponentialResidual); break;      \
    case 3: MACRO(sk::PowellF1); break;                 \
    case 4: MACRO(sk::PowellF2); break;                 \
    case 5: MACRO(sk::PowellF3); break;                 \
    case 6: MACRO(sk::PowellF4); break;                 \
    case 7: MACRO(sk::BinaryScalarCost); break;         \
    case 8: MACRO(sk::BinaryVector3Cost); break;        \
    case 9: MACRO(sk::TenParameterCost); break;         \
    case 11: MACRO(sk::HelloCostFunctor); break;        \
    case 12: MACRO(sk::QuaternionRotationError); break; \
    default: break;                               \
  }

}  // namespace sk
