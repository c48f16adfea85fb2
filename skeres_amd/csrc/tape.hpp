// Recorded functor bodies ("tapes") and their device interpreter.
//
// The reference's residual is a JVM closure, generic in T over spire's Field / Trig / NRoot / Order
// (CORE/CostFunctor.scala:40-51); a GPU cannot call it back.  The functors of the reference's own examples have
// device bodies in functors.hpp; any OTHER functor reaches the device as a tape: its apply[T] is run ONCE on the host
// with a recording T (skeres_amd/tape.py is the Python mirror of that recording T) and what
// it did to its arguments — a straight-line list of arithmetic instructions over virtual registers — is what
// sk_cost_function_new_tape takes.  Comparisons (spire's Order on the real part, CORE/Rotation.scala:458) cannot
// branch in a recording, so both arms are recorded and a SELECT picks one per evaluation; the arm not taken may well
// produce NaN (1 / theta at theta = 0) — a select copies, it never blends.
//
// Evaluation is forward-mode autodiff as in CORE/AutodiffCostFunction.scala:96-130, W derivative slots per pass:
// pass p seeds parameters [p W, (p + 1) W) and yields those columns of the Jacobian (the real parts are recomputed in
// every pass).  The registers of a thread live in LDS, component-major (reg, component, thread): a register index is a
// run-time value, so they cannot live in VGPRs, and 13 doubles per register in scratch memory would be HBM traffic.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>
#include "jet.hpp"

namespace sk {

// operand = kind << 28 | index
enum : int { kTapeReg = 0, kTapeParam = 1, kTapeObs = 2, kTapeConst = 3 };
enum : int {
  kTapeMov = 0, kTapeAdd, kTapeSub, kTapeMul, kTapeDiv, kTapeNeg, kTapeSqrt, kTapeExp, kTapeLog, kTapeSin, kTapeCos, kTapeTan,
  kTapeAsin, kTapeAcos, kTapeAtan, kTapeAtan2, kTapeAbs, kTapeLt, kTapeLe, kTapeSelect, kTapeNumOps
};
struct TapeIns { int32_t op, dst, a, b, c; };  // dst: register; a, b, c: operands (SELECT: a = condition, b = then, c = else)

// host copy (owned by the cost function, interned per problem by content)
struct Tape {
  int num_residuals = 0, num_registers = 0, num_obs_consts = 0;
  std::vector<int> block_sizes;
  std::vector<TapeIns> ins;
  std::vector<double> consts;
  std::vector<int32_t> out;  // operand per residual
  int dim() const { int d = 0; for (int b : block_sizes) d += b; return d; }
  std::string key() const;   // content key
};
// "" when well-formed, else what is wrong
std::string tape_validate(const Tape& t);

// device view
struct TapeDev {
  const TapeIns* ins; int num_ins;
  const double* consts;
  const int32_t* out; int num_residuals;
  int num_registers, dim, num_blocks;
  const int* param_block;  // [dim] parameter block of flattened parameter k
  const int* param_index;  // [dim] its index inside the block
};

SK_HD int tape_kind(int32_t operand) { return (operand >> 28) & 7; }
SK_HD int tape_index(int32_t operand) { return operand & 0x0fffffff; }

#ifdef __HIPCC__
template <class T> struct TapeWidth;
template <> struct TapeWidth<double> { static constexpr int W = 0; };
template <int N> struct TapeWidth<Jet<N>> { static constexpr int W = N; };

// a thread's registers in LDS: component c of register r at base[(r (W + 1) + c) * nthreads + tid]
template <class T>
struct TapeRegs {
  double* base; int nthreads, tid;
  static constexpr int W = TapeWidth<T>::W;
  __device__ __forceinline__ T load(int r) const;
  __device__ __forceinline__ void store(int r, const T& v) const;
};
template <> __device__ __forceinline__ double TapeRegs<double>::load(int r) const { return base[r * nthreads + tid]; }
template <> __device__ __forceinline__ void TapeRegs<double>::store(int r, const double& v) const { base[r * nthreads + tid] = v; }
template <class T> __device__ __forceinline__ T TapeRegs<T>::load(int r) const {
  T v;
  const double* p = base + (size_t)r * (W + 1) * nthreads + tid;
  v.a = p[0];
#pragma unroll
  for (int i = 0; i < W; ++i) v.v[i] = p[(i + 1) * nthreads];
  return v;
}
template <class T> __device__ __forceinline__ void TapeRegs<T>::store(int r, const T& v) const {
  double* p = base + (size_t)r * (W + 1) * nthreads + tid;
  p[0] = v.a;
#pragma unroll
  for (int i = 0; i < W; ++i) p[(i + 1) * nthreads] = v.v[i];
}

__device__ __forceinline__ double tape_make(double x, int, double*) { return x; }
template <int N> __device__ __forceinline__ Jet<N> tape_make(double x, int slot, Jet<N>*) { return Jet<N>(x, slot); }
__device__ __forceinline__ double tape_real(double x) { return x; }
template <int N> __device__ __forceinline__ double tape_real(const Jet<N>& x) { return x.a; }
// f(x) with derivative df: the chain rule on a Jet, the value on a double
__device__ __forceinline__ double tape_chain(double, double f, double) { return f; }
template <int N> __device__ __forceinline__ Jet<N> tape_chain(const Jet<N>& x, double f, double df) {
  Jet<N> h; h.a = f;
#pragma unroll
  for (int i = 0; i < N; ++i) h.v[i] = df * x.v[i];
  return h;
}

// Runs the tape for one residual block.  param(k) = value of flattened parameter k; `first` = first parameter seeded
// in this pass (slot = k - first; outside [0, W) the derivative is zero).  Results: out[r], r < num_residuals.
template <class T, class ParamFn>
__device__ __forceinline__ void tape_run(const TapeDev& t, const double* obs, const ParamFn& param, int first, const TapeRegs<T>& regs, T* out) {
  auto fetch = [&](int32_t code) -> T {
    const int idx = tape_index(code);
    switch (tape_kind(code)) {
      case kTapeReg: return regs.load(idx);
      case kTapeParam: return tape_make(param(idx), idx - first, (T*)nullptr);
      case kTapeObs: return T(obs[idx]);
      default: return T(t.consts[idx]);
    }
  };
  for (int i = 0; i < t.num_ins; ++i) {
    const TapeIns in = t.ins[i];
    T r;
    switch (in.op) {
      case kTapeMov: r = fetch(in.a); break;
      case kTapeAdd: r = fetch(in.a) + fetch(in.b); break;
      case kTapeSub: r = fetch(in.a) - fetch(in.b); break;
      case kTapeMul: r = fetch(in.a) * fetch(in.b); break;
      case kTapeDiv: r = fetch(in.a) / fetch(in.b); break;
      case kTapeNeg: r = -fetch(in.a); break;
      case kTapeSqrt: r = jsqrt(fetch(in.a)); break;
      case kTapeExp: r = jexp(fetch(in.a)); break;
      case kTapeLog: { const T x = fetch(in.a); const double xr = tape_real(x); r = tape_chain(x, ::log(xr), 1.0 / xr); break; }
      case kTapeSin: { const T x = fetch(in.a); double s, c; ::sincos(tape_real(x), &s, &c); r = tape_chain(x, s, c); break; }
      case kTapeCos: { const T x = fetch(in.a); double s, c; ::sincos(tape_real(x), &s, &c); r = tape_chain(x, c, -s); break; }
      case kTapeTan: { const T x = fetch(in.a); const double tn = ::tan(tape_real(x)); r = tape_chain(x, tn, 1.0 + tn * tn); break; }
      case kTapeAsin: { const T x = fetch(in.a); const double xr = tape_real(x); r = tape_chain(x, ::asin(xr), 1.0 / ::sqrt(1.0 - xr * xr)); break; }
      case kTapeAcos: { const T x = fetch(in.a); const double xr = tape_real(x); r = tape_chain(x, ::acos(xr), -1.0 / ::sqrt(1.0 - xr * xr)); break; }
      case kTapeAtan: { const T x = fetch(in.a); const double xr = tape_real(x); r = tape_chain(x, ::atan(xr), 1.0 / (1.0 + xr * xr)); break; }
      case kTapeAtan2: r = jatan2(fetch(in.a), fetch(in.b)); break;
      case kTapeAbs: { const T x = fetch(in.a); r = tape_real(x) < 0.0 ? -x : x; break; }
      case kTapeLt: r = T(tape_real(fetch(in.a)) < tape_real(fetch(in.b)) ? 1.0 : 0.0); break;
      case kTapeLe: r = T(tape_real(fetch(in.a)) <= tape_real(fetch(in.b)) ? 1.0 : 0.0); break;
      default: r = tape_real(fetch(in.a)) != 0.0 ? fetch(in.b) : fetch(in.c); break;  // kTapeSelect
    }
    regs.store(in.dst, r);
  }
  for (int r = 0; r < t.num_residuals; ++r) out[r] = fetch(t.out[r]);
}
#endif  // __HIPCC__

constexpr int kTapeFunctorBase = 1000;  // Problem::rb_functor of a tape block = kTapeFunctorBase + index into Problem::tapes
constexpr int kTapeMaxResiduals = 16, kTapeMaxDim = 64;
constexpr size_t kTapeLdsBudget = 144 * 1024;
// LDS bytes of the register file of `threads` threads at W derivative slots (W = 0: cost only)
inline size_t tape_lds_bytes(const Tape& t, int W, int threads) { return (size_t)std::max(1, t.num_registers) * (W + 1) * threads * sizeof(double); }
int tape_pick_width(const Tape& t, int threads);

}  // namespace sk
