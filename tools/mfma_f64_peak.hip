// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on the whole chip
// (back-to-back issue, independent accumulators, operands in registers), and
// a streaming HBM copy, to replace datasheet peaks with on-box measurements.
//   build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0, long long* clk) {
  const long long w0 = wall_clock64(), c0 = clock64();
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = wall_clock64() - w0; clk[1] = clock64() - c0; }
}

__global__ void copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <int NACC>
static void run(int waves_per_simd, int iters, int cus = 256) {
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
  double* out;
  long long* clk;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0, 2.0, clk);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0 + rep, 2.0, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * (double)iters * NACC * 2.0 * 16 * 16 * 4;
  long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mhz = 100.0 * (double)h[1] / (double)h[0];  // s_memtime ticks per 100 MHz wall-clock tick
  printf("mfma_f64_16x16x4 acc=%d waves/SIMD=%d on %3d CUs: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz; shader clock counter %.0f MHz)\n", NACC,
         waves_per_simd, cus, best, flops / best * 1e-9, 2.4e9 * best * 1e-3 / ((double)iters * NACC * waves_per_simd), mhz);
  hipFree(clk);
  hipFree(out);
}

int main() {
  run<4>(1, 20000); run<8>(1, 10000); run<16>(1, 5000); run<8>(2, 10000); run<8>(4, 5000); run<4>(8, 5000);
  // fewer busy CUs: is the per-SIMD rate a property of the pipe, or of the whole chip under load?
  run<8>(2, 10000, 8); run<8>(2, 10000, 64); run<8>(2, 10000, 128); run<8>(1, 10000, 8);
  const size_t n = (size_t)1 << 28;  // 4 GiB in + 4 GiB out
  double2 *a, *b;
  hipMalloc(&a, n * sizeof(double2)); hipMalloc(&b, n * sizeof(double2));
  hipMemset(a, 1, n * sizeof(double2));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(copy_kernel, dim3(2048 * 4), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("HBM copy 16B/lane: %.3f ms  %.2f TB/s (read+write)\n", best, 2.0 * n * sizeof(double2) / best * 1e-9);
  return 0;
}
