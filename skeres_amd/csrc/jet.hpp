// Device dual numbers for per-residual-block forward-mode autodiff.
//
// Replaces spire.math.Jet[Double] as used by the reference at
// core/src/main/scala/org/somelightprojections/skeres/AutodiffCostFunction.scala:95-122
// (seeded as Jet(x, k), read back through .real / .infinitesimal).  Semantics
// follow spire 0.11 / Ceres jet.h: first-order propagation, comparisons by
// real part only (core/.../package.scala:27).
#pragma once
#include <hip/hip_runtime.h>

#define SK_HD __host__ __device__ __forceinline__

namespace sk {

template <int N>
struct Jet {
  double a;
  double v[N];
  SK_HD Jet() {}
  SK_HD Jet(double x) : a(x) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  SK_HD Jet(double x, int k) : a(x) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (i == k) ? 1.0 : 0.0;
  }
};

template <int N> SK_HD Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a + g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i];
  return h;
}
template <int N> SK_HD Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a - g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i];
  return h;
}
template <int N> SK_HD Jet<N> operator-(const Jet<N>& f) {
  Jet<N> h; h.a = -f.a;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = -f.v[i];
  return h;
}
template <int N> SK_HD Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a * g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = g.a * f.v[i] + f.a * g.v[i];
  return h;
}
template <int N> SK_HD Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  const double gi = 1.0 / g.a;
  const double q = f.a * gi;
  Jet<N> h; h.a = q;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = gi * (f.v[i] - q * g.v[i]);
  return h;
}
// double (op) Jet: the double is lifted to a constant Jet, as spire's literal syntax does.
template <int N> SK_HD Jet<N> operator+(double s, const Jet<N>& g) { Jet<N> h = g; h.a = s + g.a; return h; }
template <int N> SK_HD Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a = f.a + s; return h; }
template <int N> SK_HD Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a = f.a - s; return h; }
template <int N> SK_HD Jet<N> operator-(double s, const Jet<N>& g) {
  Jet<N> h; h.a = s - g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = -g.v[i];
  return h;
}
template <int N> SK_HD Jet<N> operator*(double s, const Jet<N>& g) {
  Jet<N> h; h.a = s * g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = s * g.v[i];
  return h;
}
template <int N> SK_HD Jet<N> operator*(const Jet<N>& f, double s) { return s * f; }
template <int N> SK_HD Jet<N> operator/(double s, const Jet<N>& g) {
  const double gi = 1.0 / g.a;
  const double q = s * gi;
  Jet<N> h; h.a = q;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = gi * (-(q * g.v[i]));
  return h;
}
template <int N> SK_HD bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> SK_HD bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> SK_HD bool operator>(const Jet<N>& f, double g) { return f.a > g; }

template <int N> SK_HD Jet<N> jsqrt(const Jet<N>& f) {
  const double s = ::sqrt(f.a);
  const double d = 0.5 / s;
  Jet<N> h; h.a = s;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i];
  return h;
}
template <int N> SK_HD void jsincos(const Jet<N>& f, Jet<N>* s, Jet<N>* c) {
  double sn, cs;
  ::sincos(f.a, &sn, &cs);
  s->a = sn; c->a = cs;
#pragma unroll
  for (int i = 0; i < N; ++i) { s->v[i] = cs * f.v[i]; c->v[i] = (-sn) * f.v[i]; }
}
template <int N> SK_HD Jet<N> jexp(const Jet<N>& f) {
  const double e = ::exp(f.a);
  Jet<N> h; h.a = e;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = e * f.v[i];
  return h;
}
// atan2(y, x): d = (x dy - y dx) / (x^2 + y^2)
template <int N> SK_HD Jet<N> jatan2(const Jet<N>& y, const Jet<N>& x) {
  const double t = 1.0 / (x.a * x.a + y.a * y.a);
  Jet<N> h; h.a = ::atan2(y.a, x.a);
  const double cy = x.a * t, cx = -(y.a * t);
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = cx * x.v[i] + cy * y.v[i];
  return h;
}
SK_HD double jatan2(double y, double x) { return ::atan2(y, x); }
SK_HD double jsqrt(double x) { return ::sqrt(x); }
SK_HD void jsincos(double x, double* s, double* c) { ::sincos(x, s, c); }
SK_HD double jexp(double x) { return ::exp(x); }
SK_HD bool jgt(double a, double b) { return a > b; }
template <int N> SK_HD bool jgt(const Jet<N>& a, double b) { return a.a > b; }

template <class T> struct JetTraits;
template <> struct JetTraits<double> {
  static SK_HD double real(double x) { return x; }
};
template <int N> struct JetTraits<Jet<N>> {
  static SK_HD double real(const Jet<N>& x) { return x.a; }
};

}  // namespace sk
