// Developer tool (round 5): issue rate and dependent latency of fp64 VALU instructions on one SIMD of gfx950, with one wave and
// with two waves on the SAME SIMD (waves of a workgroup go to the SIMDs round-robin: wave w -> SIMD w % 4) — what a second wave
// next to potrf128's chain wave could take off it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_f64_probe.hip -o build/probes/valu_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ long long g_out[16][8];
__device__ double g_sink[1024];

template <int kIndep>
__device__ __forceinline__ long long fma_loop(double seed, int iters, double* sink) {
  double a[kIndep];
#pragma unroll
  for (int i = 0; i < kIndep; ++i) a[i] = seed + i;
  const double m = seed * 1e-9, b = 1.0 + seed * 1e-12;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < kIndep; ++i) a[i] = __builtin_fma(-m, b, a[i]);
#pragma unroll
    for (int i = 0; i < kIndep; ++i) asm volatile("" : "+v"(a[i]));
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kIndep; ++i) s += a[i];
  *sink = s;
  return t1 - t0;
}

// mode 0: wave 0 alone, 16 independent accumulators; 1: waves 0 and 4 (same SIMD) both; 2: dependent chain (1 accumulator);
// 3: wave 0 dependent chain, wave 4 independent stream on the same SIMD; 4: rcp chain; 5: readlane -> fma chain
__global__ void probe(int mode, int iters, double seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long dt = 0;
  double sink = 0.0;
  if (mode == 0) { if (wave == 0) dt = fma_loop<16>(seed + lane, iters, &sink); }
  else if (mode == 1) { if (wave == 0 || wave == 4) dt = fma_loop<16>(seed + lane, iters, &sink); }
  else if (mode == 2) { if (wave == 0) dt = fma_loop<1>(seed + lane, iters * 16, &sink); }
  else if (mode == 3) { if (wave == 0) dt = fma_loop<1>(seed + lane, iters * 16, &sink); else if (wave == 4) dt = fma_loop<16>(seed + lane, iters * 3, &sink); }
  else if (mode == 4) {
    if (wave == 0) {
      double y = seed + lane + 1.5;
      const long long t0 = clock64();
      for (int it = 0; it < iters * 16; ++it) { y = __builtin_amdgcn_rcp(y); asm volatile("" : "+v"(y)); }
      dt = clock64() - t0; sink = y;
    }
  } else if (mode == 5) {
    if (wave == 0) {
      double a = seed + lane + 1.5;
      const long long t0 = clock64();
      for (int it = 0; it < iters * 16; ++it) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3);
        const double d = __hiloint2double(hi, lo);
        a = __builtin_fma(d, 1e-9, a);
        asm volatile("" : "+v"(a));
      }
      dt = clock64() - t0; sink = a;
    }
  } else if (mode == 6) {  // 16 independent accumulators, operands chosen by the compiler: same as mode 0 but 2 waves on DIFFERENT SIMDs (0 and 1)
    if (wave == 0 || wave == 1) dt = fma_loop<16>(seed + lane, iters, &sink);
  }
  if (lane == 0 && wave < 8) g_out[mode][wave] = dt;
  g_sink[threadIdx.x] = sink;
}

int main() {
  const int iters = 4000;
  const char* names[7] = {"1 wave, 16 independent fma", "2 waves same SIMD, 16 independent fma each", "1 wave, dependent fma chain", "dependent chain (wave 0) + independent stream (wave 4, same SIMD)",
                          "rcp_f64 dependent chain", "readlane x2 -> fma chain", "2 waves on different SIMDs"};
  for (int mode = 0; mode < 7; ++mode) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, mode, iters, 1.0);
    hipDeviceSynchronize();
    long long out[16][8];
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_out), sizeof(out));
    const double n0 = (mode == 2 || mode == 3 || mode >= 4) && mode != 6 ? iters * 16.0 : iters * 16.0;
    printf("%-70s wave0 %.2f clk/instr", names[mode], out[mode][0] / n0);
    if (mode == 1) printf("  wave4 %.2f", out[mode][4] / (iters * 16.0));
    if (mode == 3) printf("  wave4 %.2f (per fma of its stream)", out[mode][4] / (iters * 3 * 16.0));
    if (mode == 6) printf("  wave1 %.2f", out[mode][1] / (iters * 16.0));
    printf("\n");
  }
  return 0;
}
