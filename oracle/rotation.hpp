// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of CORE/Rotation.scala:63-522 (itself a port of ceres/rotation.h), generic in
// T = double or oracle::Jet<N>.  Pinned by the reference's own RotationSpec.scala:248-655 through
// tests/test_rotation.py.  Quaternions are (w, x, y, z) (spire's Quaternion(r, i, j, k)); matrices are
// addressed as M(i, j) = data[i * rowStride + j * colStride] like the reference's MatrixAdapter (:14-57).
#pragma once
#include "functors.hpp"  // angleAxisRotatePoint, kUlpOne
#include "jet.hpp"

namespace oracle {

template <class T> struct Mat3 {  // MatrixAdapter over 9 values
  T* data; int rowStride, colStride;
  T& operator()(int i, int j) const { return data[i * rowStride + j * colStride]; }
};
template <class T> inline double realPart(const T& x) { return Scalar<T>::real(x); }

template <class T>  // :72-92
inline void angleAxisToQuaternion(const T* angleAxis, T* quaternion) {
  const T a0 = angleAxis[0], a1 = angleAxis[1], a2 = angleAxis[2];
  const T thetaSquared = a0 * a0 + a1 * a1 + a2 * a2;
  if (realPart(thetaSquared) > 0.0) {
    const T theta = sqrt(thetaSquared);
    const T halfTheta = theta * 0.5;
    const T k = sin(halfTheta) / theta;
    quaternion[0] = cos(halfTheta); quaternion[1] = a0 * k; quaternion[2] = a1 * k; quaternion[3] = a2 * k;
  } else {
    quaternion[0] = T(1.0); quaternion[1] = a0 * 0.5; quaternion[2] = a1 * 0.5; quaternion[3] = a2 * 0.5;
  }
}

template <class T>  // :104-131
inline void quaternionToAngleAxis(const T* quaternion, T* angleAxis) {
  const T q1 = quaternion[1], q2 = quaternion[2], q3 = quaternion[3];
  const T sinSquaredTheta = q1 * q1 + q2 * q2 + q3 * q3;
  if (realPart(sinSquaredTheta) != 0.0) {
    const T sinTheta = sqrt(sinSquaredTheta);
    const T cosTheta = quaternion[0];
    const T twoTheta = realPart(cosTheta) < 0.0 ? 2.0 * atan2(-sinTheta, -cosTheta) : 2.0 * atan2(sinTheta, cosTheta);
    const T k = twoTheta / sinTheta;
    angleAxis[0] = q1 * k; angleAxis[1] = q2 * k; angleAxis[2] = q3 * k;
  } else {
    angleAxis[0] = q1 * 2.0; angleAxis[1] = q2 * 2.0; angleAxis[2] = q3 * 2.0;
  }
}

template <class T>  // :162-189
inline void rotationMatrixToQuaternion(const Mat3<const T>& R, T* q) {
  const T trace = R(0, 0) + R(1, 1) + R(2, 2);
  if (realPart(trace) >= 0.0) {
    T t = sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R(2, 1) - R(1, 2)) * t; q[2] = (R(0, 2) - R(2, 0)) * t; q[3] = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (realPart(R(1, 1)) > realPart(R(0, 0))) i = 1;
    if (realPart(R(2, 2)) > realPart(R(i, i))) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    T t = sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R(k, j) - R(j, k)) * t; q[j + 1] = (R(j, i) + R(i, j)) * t; q[k + 1] = (R(k, i) + R(i, k)) * t;
  }
}

template <class T>  // :211-250
inline void angleAxisToRotationMatrixT(const T* angleAxis, const Mat3<T>& R) {
  const T theta2 = angleAxis[0] * angleAxis[0] + angleAxis[1] * angleAxis[1] + angleAxis[2] * angleAxis[2];
  if (realPart(theta2) > kUlpOne) {
    const T theta = sqrt(theta2);
    const T wx = angleAxis[0] / theta, wy = angleAxis[1] / theta, wz = angleAxis[2] / theta;
    const T costheta = cos(theta), sintheta = sin(theta);
    const T kOne(1.0);
    R(0, 0) = costheta + wx * wx * (kOne - costheta);
    R(1, 0) = wz * sintheta + wx * wy * (kOne - costheta);
    R(2, 0) = -wy * sintheta + wx * wz * (kOne - costheta);
    R(0, 1) = wx * wy * (kOne - costheta) - wz * sintheta;
    R(1, 1) = costheta + wy * wy * (kOne - costheta);
    R(2, 1) = wx * sintheta + wy * wz * (kOne - costheta);
    R(0, 2) = wy * sintheta + wx * wz * (kOne - costheta);
    R(1, 2) = -wx * sintheta + wy * wz * (kOne - costheta);
    R(2, 2) = costheta + wz * wz * (kOne - costheta);
  } else {
    R(0, 0) = T(1.0); R(1, 0) = angleAxis[2]; R(2, 0) = -angleAxis[1];
    R(0, 1) = -angleAxis[2]; R(1, 1) = T(1.0); R(2, 1) = angleAxis[0];
    R(0, 2) = angleAxis[1]; R(1, 2) = -angleAxis[0]; R(2, 2) = T(1.0);
  }
}

template <class T>  // :269-298
inline void eulerAnglesToRotationMatrix(const T* euler, const Mat3<T>& R) {
  const double degreesToRadians = M_PI / 180.0;
  const T pitch = euler[0] * degreesToRadians, roll = euler[1] * degreesToRadians, yaw = euler[2] * degreesToRadians;
  const T c1 = cos(yaw), s1 = sin(yaw), c2 = cos(roll), s2 = sin(roll), c3 = cos(pitch), s3 = sin(pitch);
  R(0, 0) = c1 * c2; R(0, 1) = -s1 * c3 + c1 * s2 * s3; R(0, 2) = s1 * s3 + c1 * s2 * c3;
  R(1, 0) = s1 * c2; R(1, 1) = c1 * c3 + s1 * s2 * s3; R(1, 2) = -c1 * s3 + s1 * s2 * c3;
  R(2, 0) = -s2; R(2, 1) = c2 * s3; R(2, 2) = c2 * c3;
}

template <class T>  // :326-353
inline void quaternionToScaledRotation(const T* q, const Mat3<T>& R) {
  const T aa = q[0] * q[0], ab = q[0] * q[1], ac = q[0] * q[2], ad = q[0] * q[3];
  const T bb = q[1] * q[1], bc = q[1] * q[2], bd = q[1] * q[3], cc = q[2] * q[2], cd = q[2] * q[3], dd = q[3] * q[3];
  R(0, 0) = aa + bb - cc - dd; R(0, 1) = 2.0 * (bc - ad); R(0, 2) = 2.0 * (ac + bd);
  R(1, 0) = 2.0 * (ad + bc); R(1, 1) = aa - bb + cc - dd; R(1, 2) = 2.0 * (cd - ab);
  R(2, 0) = 2.0 * (bd - ac); R(2, 1) = 2.0 * (ab + cd); R(2, 2) = aa - bb - cc + dd;
}

template <class T>  // :364-381
inline bool quaternionToRotation(const T* q, const Mat3<T>& R) {
  quaternionToScaledRotation(q, R);
  const T norm = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (realPart(norm) == 0.0) return false;
  const T invNorm = 1.0 / norm;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = R(i, j) * invNorm;
  return true;
}

template <class T>  // :393-415
inline void unitQuaternionRotatePoint(const T* q, const T* pt, T* out) {
  const T t2 = q[0] * q[1], t3 = q[0] * q[2], t4 = q[0] * q[3], t5 = -q[1] * q[1], t6 = q[1] * q[2], t7 = q[1] * q[3];
  const T t8 = -q[2] * q[2], t9 = q[2] * q[3], t1 = -q[3] * q[3];
  out[0] = 2.0 * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
  out[1] = 2.0 * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
  out[2] = 2.0 * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}

template <class T>  // :422-430
inline void quaternionRotatePoint(const T* q, const T* pt, T* out) {
  const T scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T unit[4] = {q[0] * scale, q[1] * scale, q[2] * scale, q[3] * scale};
  unitQuaternionRotatePoint(unit, pt, out);
}

template <class T>  // :435-438, spire Quaternion *
inline void quaternionProduct(const T* z, const T* w, T* zw) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

// :441-442 — the mathematical cross product; the reference's first component (x1*y2 - y2*x1 == 0) is a typo
template <class T>
inline void crossProduct(const T* x, const T* y, T* out) {
  out[0] = x[1] * y[2] - x[2] * y[1]; out[1] = x[2] * y[0] - x[0] * y[2]; out[2] = x[0] * y[1] - x[1] * y[0];
}

// One item of op `op` (ids of include/skeres_amd.h: sk_rotation_op) on values of type T.
template <class T>
inline bool rotationApply(int op, int rowMajor, const T* in, T* out) {
  const int rs = rowMajor ? 3 : 1, cs = rowMajor ? 1 : 3;
  switch (op) {
    case 0: angleAxisToQuaternion(in, out); return true;
    case 1: quaternionToAngleAxis(in, out); return true;
    case 2: rotationMatrixToQuaternion(Mat3<const T>{in, rs, cs}, out); return true;
    case 3: { T q[4]; rotationMatrixToQuaternion(Mat3<const T>{in, rs, cs}, q); quaternionToAngleAxis(q, out); return true; }
    case 4: angleAxisToRotationMatrixT(in, Mat3<T>{out, rs, cs}); return true;
    case 5: eulerAnglesToRotationMatrix(in, Mat3<T>{out, rs, cs}); return true;
    case 6: quaternionToScaledRotation(in, Mat3<T>{out, rs, cs}); return true;
    case 7: return quaternionToRotation(in, Mat3<T>{out, rs, cs});
    case 8: unitQuaternionRotatePoint(in, in + 4, out); return true;
    case 9: quaternionRotatePoint(in, in + 4, out); return true;
    case 10: quaternionProduct(in, in + 4, out); return true;
    case 11: crossProduct(in, in + 3, out); return true;
    case 12: out[0] = in[0] * in[3] + in[1] * in[4] + in[2] * in[5]; return true;
    case 13: angleAxisRotatePoint(in, in + 3, out); return true;
  }
  return false;
}

static const int kRotationIn[14] = {3, 4, 9, 9, 3, 3, 4, 4, 7, 7, 8, 6, 6, 6};
static const int kRotationOut[14] = {4, 3, 4, 3, 9, 9, 9, 9, 3, 3, 4, 3, 1, 3};

template <int K>
inline bool rotationApplyJets(int op, int rowMajor, const double* in, int n, double* out) {
  const int ni = kRotationIn[op], no = kRotationOut[op];
  for (int it = 0; it < n; ++it) {
    Jet<K> x[9], y[9];
    for (int e = 0; e < ni; ++e) {
      const double* p = in + ((size_t)it * ni + e) * (1 + K);
      x[e].a = p[0];
      for (int k = 0; k < K; ++k) x[e].v[k] = p[1 + k];
    }
    if (!rotationApply<Jet<K>>(op, rowMajor, x, y)) return false;
    for (int e = 0; e < no; ++e) {
      double* p = out + ((size_t)it * no + e) * (1 + K);
      p[0] = y[e].a;
      for (int k = 0; k < K; ++k) p[1 + k] = y[e].v[k];
    }
  }
  return true;
}

}  // namespace oracle
