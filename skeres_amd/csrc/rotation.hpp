// Rotation (CORE/Rotation.scala:63-522): conversions between angle-axis, quaternion, rotation-matrix and
// Euler-angle representations and the point rotations, generic in T (double or Jet<N>) so that device
// functors can call them under automatic differentiation.  The reference file is itself a port of
// ceres/rotation.h; conventions kept: quaternions are (w, x, y, z); a matrix is addressed through
// (row stride, column stride) as the reference's MatrixAdapter does (column-major 3x3: (1, 3), the
// default of the angle-axis / quaternion conversions; row-major: (3, 1), the default of the Euler and
// quaternion-to-rotation functions); Euler angles are (pitch, roll, yaw) in DEGREES.
#pragma once
#include "jet.hpp"

namespace sk {

template <class T> SK_HD bool jnonzero(const T& x) { return JetTraits<T>::real(x) != 0.0; }
template <class T> SK_HD bool jneg(const T& x) { return JetTraits<T>::real(x) < 0.0; }

// Rotation.scala:72-92
template <class T>
SK_HD void angle_axis_to_quaternion(const T* aa, T* q) {
  const T a0 = aa[0], a1 = aa[1], a2 = aa[2];
  const T theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (jgt(theta2, 0.0)) {
    const T theta = jsqrt(theta2);
    const T half = theta * 0.5;
    T s, c;
    jsincos(half, &s, &c);
    const T k = s / theta;
    q[0] = c; q[1] = a0 * k; q[2] = a1 * k; q[3] = a2 * k;
  } else {  // first-order expansion at the origin, keeps the derivatives right
    q[0] = T(1.0); q[1] = a0 * 0.5; q[2] = a1 * 0.5; q[3] = a2 * 0.5;
  }
}

// Rotation.scala:104-131: unit quaternion -> angle-axis with an angle in [0, pi]
template <class T>
SK_HD void quaternion_to_angle_axis(const T* q, T* aa) {
  const T q1 = q[1], q2 = q[2], q3 = q[3];
  const T sin2 = q1 * q1 + q2 * q2 + q3 * q3;
  if (jnonzero(sin2)) {
    const T sin_theta = jsqrt(sin2);
    const T cos_theta = q[0];
    // |angle| <= pi: for cos < 0 use atan2(-sin, -cos), the same rotation the short way round
    const T two_theta = jneg(cos_theta) ? 2.0 * jatan2(-sin_theta, -cos_theta) : 2.0 * jatan2(sin_theta, cos_theta);
    const T k = two_theta / sin_theta;
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  } else {
    aa[0] = q1 * 2.0; aa[1] = q2 * 2.0; aa[2] = q3 * 2.0;
  }
}

// Rotation.scala:162-189
template <class T>
SK_HD void rotation_matrix_to_quaternion(const T* R, int rs, int cs, T* q) {
#define SK_R(i, j) R[(i) * rs + (j) * cs]
  const T trace = SK_R(0, 0) + SK_R(1, 1) + SK_R(2, 2);
  if (!jneg(trace)) {
    T t = jsqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (SK_R(2, 1) - SK_R(1, 2)) * t;
    q[2] = (SK_R(0, 2) - SK_R(2, 0)) * t;
    q[3] = (SK_R(1, 0) - SK_R(0, 1)) * t;
  } else {
    int i = 0;
    if (SK_R(1, 1) > SK_R(0, 0)) i = 1;
    if (SK_R(2, 2) > SK_R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    T t = jsqrt(SK_R(i, i) - SK_R(j, j) - SK_R(k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (SK_R(k, j) - SK_R(j, k)) * t;
    q[j + 1] = (SK_R(j, i) + SK_R(i, j)) * t;
    q[k + 1] = (SK_R(k, i) + SK_R(i, k)) * t;
  }
#undef SK_R
}

// Rotation.scala:203-204
template <class T>
SK_HD void rotation_matrix_to_angle_axis(const T* R, int rs, int cs, T* aa) {
  T q[4];
  rotation_matrix_to_quaternion(R, rs, cs, q);
  quaternion_to_angle_axis(q, aa);
}

// Rotation.scala:211-250
template <class T>
SK_HD void angle_axis_to_rotation_matrix(const T* aa, T* R, int rs, int cs) {
#define SK_R(i, j) R[(i) * rs + (j) * cs]
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (jgt(theta2, 2.220446049250313e-16)) {  // ulp(1.0)
    const T theta = jsqrt(theta2);
    const T wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    T s, c;
    jsincos(theta, &s, &c);
    const T omc = 1.0 - c;
    SK_R(0, 0) = c + wx * wx * omc;
    SK_R(1, 0) = wz * s + wx * wy * omc;
    SK_R(2, 0) = -wy * s + wx * wz * omc;
    SK_R(0, 1) = wx * wy * omc - wz * s;
    SK_R(1, 1) = c + wy * wy * omc;
    SK_R(2, 1) = wx * s + wy * wz * omc;
    SK_R(0, 2) = wy * s + wx * wz * omc;
    SK_R(1, 2) = -wx * s + wy * wz * omc;
    SK_R(2, 2) = c + wz * wz * omc;
  } else {  // near zero: first-order expansion
    SK_R(0, 0) = T(1.0); SK_R(1, 0) = aa[2]; SK_R(2, 0) = -aa[1];
    SK_R(0, 1) = -aa[2]; SK_R(1, 1) = T(1.0); SK_R(2, 1) = aa[0];
    SK_R(0, 2) = aa[1]; SK_R(1, 2) = -aa[0]; SK_R(2, 2) = T(1.0);
  }
#undef SK_R
}

// Rotation.scala:269-298: (pitch, roll, yaw) in degrees
template <class T>
SK_HD void euler_angles_to_rotation_matrix(const T* euler, T* R, int rs, int cs) {
#define SK_R(i, j) R[(i) * rs + (j) * cs]
  const double d2r = 3.141592653589793 / 180.0;
  const T pitch = euler[0] * d2r, roll = euler[1] * d2r, yaw = euler[2] * d2r;
  T c1, s1, c2, s2, c3, s3;
  jsincos(yaw, &s1, &c1);
  jsincos(roll, &s2, &c2);
  jsincos(pitch, &s3, &c3);
  SK_R(0, 0) = c1 * c2;
  SK_R(0, 1) = -s1 * c3 + c1 * s2 * s3;
  SK_R(0, 2) = s1 * s3 + c1 * s2 * c3;
  SK_R(1, 0) = s1 * c2;
  SK_R(1, 1) = c1 * c3 + s1 * s2 * s3;
  SK_R(1, 2) = -c1 * s3 + s1 * s2 * c3;
  SK_R(2, 0) = -s2;
  SK_R(2, 1) = c2 * s3;
  SK_R(2, 2) = c2 * c3;
#undef SK_R
}

// Rotation.scala:326-353: R = |q|^2 * rotation(q)
template <class T>
SK_HD void quaternion_to_scaled_rotation(const T* q, T* R, int rs, int cs) {
#define SK_R(i, j) R[(i) * rs + (j) * cs]
  const T aa = q[0] * q[0], ab = q[0] * q[1], ac = q[0] * q[2], ad = q[0] * q[3];
  const T bb = q[1] * q[1], bc = q[1] * q[2], bd = q[1] * q[3];
  const T cc = q[2] * q[2], cd = q[2] * q[3], dd = q[3] * q[3];
  SK_R(0, 0) = aa + bb - cc - dd; SK_R(0, 1) = 2.0 * (bc - ad); SK_R(0, 2) = 2.0 * (ac + bd);
  SK_R(1, 0) = 2.0 * (ad + bc); SK_R(1, 1) = aa - bb + cc - dd; SK_R(1, 2) = 2.0 * (cd - ab);
  SK_R(2, 0) = 2.0 * (bd - ac); SK_R(2, 1) = 2.0 * (ab + cd); SK_R(2, 2) = aa - bb - cc + dd;
#undef SK_R
}

// Rotation.scala:364-381; false for the zero quaternion (the reference `require`s a non-zero norm)
template <class T>
SK_HD bool quaternion_to_rotation(const T* q, T* R, int rs, int cs) {
  quaternion_to_scaled_rotation(q, R, rs, cs);
  const T norm = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (!jnonzero(norm)) return false;
  const T inv = 1.0 / norm;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * rs + j * cs] = R[i * rs + j * cs] * inv;
  return true;
}

// Rotation.scala:393-415
template <class T>
SK_HD void unit_quaternion_rotate_point(const T* q, const T* pt, T* out) {
  const T t2 = q[0] * q[1], t3 = q[0] * q[2], t4 = q[0] * q[3];
  const T t5 = -q[1] * q[1], t6 = q[1] * q[2], t7 = q[1] * q[3];
  const T t8 = -q[2] * q[2], t9 = q[2] * q[3], t1 = -q[3] * q[3];
  out[0] = 2.0 * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
  out[1] = 2.0 * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
  out[2] = 2.0 * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}

// Rotation.scala:422-430
template <class T>
SK_HD void quaternion_rotate_point(const T* q, const T* pt, T* out) {
  const T scale = 1.0 / jsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T unit[4] = {q[0] * scale, q[1] * scale, q[2] * scale, q[3] * scale};
  unit_quaternion_rotate_point(unit, pt, out);
}

// Rotation.scala:435-438 (spire's Hamilton product z * w)
template <class T>
SK_HD void quaternion_product(const T* z, const T* w, T* zw) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

// Rotation.scala:441-442.  The reference's first component reads x(1)*y(2) - y(2)*x(1), which is identically
// zero: a typo for x(1)*y(2) - x(2)*y(1) (its own angleAxisRotatePoint, :478-480, spells the product out
// correctly).  The mathematical cross product is implemented here; DESIGN.md §6 records the deviation.
template <class T>
SK_HD void cross_product(const T* x, const T* y, T* out) {
  out[0] = x[1] * y[2] - x[2] * y[1];
  out[1] = x[2] * y[0] - x[0] * y[2];
  out[2] = x[0] * y[1] - x[1] * y[0];
}

// Rotation.scala:445-446
template <class T>
SK_HD T dot_product(const T* x, const T* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; }

}  // namespace sk
