// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product (skeres_amd/csrc) never includes this file.
//
// CPU restatement of the dual-number arithmetic the reference uses for
// autodiff: spire 0.11.0 `spire.math.Jet` [ext, build.sbt:3], itself a port
// of Ceres jet.h.  spire is NOT vendored under /root/reference, so the
// operation formulas below are restated from its published algorithm:
//   (a, v) + (b, w) = (a+b, v+w)
//   (a, v) * (b, w) = (a*b, b*v + a*w)
//   (a, v) / (b, w) = (a*(1/b), (1/b) * (v - (a*(1/b)) * w))
//   sqrt(a, v)      = (sqrt a, (0.5/sqrt a) * v)
//   sin / cos / exp : first-order chain rule
// Pinned by the reference's own known-answer tests
// (core/src/test/scala/.../AutodiffCostFuntionSpec.scala:13-139, exact doubles)
// through tests/test_oracle_kat.py.
//
// Ordering: Jets compare by REAL PART ONLY
// (core/src/main/scala/org/somelightprojections/skeres/package.scala:27).
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  // Field.fromDouble: a constant has zero infinitesimal part.
  Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  // Jet[Double](x, k): real part x, infinitesimal part = k-th unit vector
  // (AutodiffCostFunction.scala:102).
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h;
}
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h;
}
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h;
}
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a * g.a;
  for (int i = 0; i < N; ++i) h.v[i] = g.a * f.v[i] + f.a * g.v[i];
  return h;
}
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  const double g_inv = 1.0 / g.a;
  const double f_by_g = f.a * g_inv;
  Jet<N> h; h.a = f_by_g;
  for (int i = 0; i < N; ++i) h.v[i] = g_inv * (f.v[i] - f_by_g * g.v[i]);
  return h;
}
// Mixed double/Jet operators: spire's literal-double syntax lifts the double
// with Field.fromDouble and then applies the Jet operator, so these forward.
template <int N> inline Jet<N> operator+(double s, const Jet<N>& g) { return Jet<N>(s) + g; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { return f + Jet<N>(s); }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& g) { return Jet<N>(s) - g; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { return f - Jet<N>(s); }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& g) { return Jet<N>(s) * g; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) { return f * Jet<N>(s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f / Jet<N>(s); }

// Order.by(_.real)  (package.scala:27)
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }

template <int N> inline Jet<N> sqrt(const Jet<N>& f) {
  const double sa = std::sqrt(f.a);
  const double two_sa_inv = 0.5 / sa;
  Jet<N> h; h.a = sa; for (int i = 0; i < N; ++i) h.v[i] = two_sa_inv * f.v[i]; return h;
}
template <int N> inline Jet<N> cos(const Jet<N>& f) {
  const double c = std::cos(f.a), ms = -std::sin(f.a);
  Jet<N> h; h.a = c; for (int i = 0; i < N; ++i) h.v[i] = ms * f.v[i]; return h;
}
template <int N> inline Jet<N> sin(const Jet<N>& f) {
  const double s = std::sin(f.a), c = std::cos(f.a);
  Jet<N> h; h.a = s; for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h;
}
template <int N> inline Jet<N> exp(const Jet<N>& f) {
  const double e = std::exp(f.a);
  Jet<N> h; h.a = e; for (int i = 0; i < N; ++i) h.v[i] = e * f.v[i]; return h;
}

// atan2(y, x) with d = (x dy - y dx) / (x^2 + y^2)
template <int N> inline Jet<N> atan2(const Jet<N>& y, const Jet<N>& x) {
  const double t = 1.0 / (x.a * x.a + y.a * y.a);
  Jet<N> h; h.a = std::atan2(y.a, x.a);
  for (int i = 0; i < N; ++i) h.v[i] = (-(y.a * t)) * x.v[i] + (x.a * t) * y.v[i];
  return h;
}
inline double atan2(double y, double x) { return std::atan2(y, x); }

// Plain-double overloads so functor bodies are generic in T.
inline double sqrt(double x) { return std::sqrt(x); }
inline double cos(double x) { return std::cos(x); }
inline double sin(double x) { return std::sin(x); }
inline double exp(double x) { return std::exp(x); }

template <class T> struct Scalar;
template <> struct Scalar<double> {
  static double real(double x) { return x; }
  static double inf(double, int) { return 0.0; }
};
template <int N> struct Scalar<Jet<N>> {
  static double real(const Jet<N>& x) { return x.a; }
  static double inf(const Jet<N>& x, int k) { return x.v[k]; }
};

}  // namespace oracle
