// Developer tool: the trailing SYRK of a SMALL trailing matrix (T block rows, K = 128 or 256) in several tilings and
// prefetch depths of gemm_nt_f64_body, alone on the device: time per launch, TFLOP/s, and whether the result equals the
// shipped thin kernel's bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I skeres_amd/csrc tools/thin_syrk_probe.hip -o gpurun_out/thin_syrk_probe
#include "../skeres_amd/csrc/chol_kernels.hip"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

namespace sk {
const DevKnobs& dev_knobs() { static DevKnobs k; return k; }
}

template <int kBKT, int kPF, int kTM, int kWM, int kOcc>
__global__ __launch_bounds__(256, kOcc) void thin_variant(double* C, long ldc, const double* A, long lda, int K, int tiles_m) {
  __shared__ __attribute__((aligned(16))) double sh[sk::gemm_lds_doubles(kBKT, kTM, 128)];
  sk::gemm_nt_f64_body<0, 2, kBKT, kPF, kTM, 128, kWM>(sh, C, ldc, A, lda, A, lda, K, tiles_m, 0);
}

// A stand-in for the column launch of the resident chain beside the SYRK: workgroups that hold 66.5 KB of LDS and wait
// (one lane polls a clock, the others sit at the barrier), as chain_column_kernel's do until potrf(j) is published.
__global__ __launch_bounds__(256, 2) void squat_kernel(long long ticks, int* sink) {
  __shared__ double hold[2 * 32 * 130];
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    hold[0] = (double)ticks;
  }
  __syncthreads();
  if (hold[0] < 0.0) *sink = 1;
}

struct Variant {
  const char* name;
  int tm;
  void (*fn)(double*, long, const double*, long, int, int);
};

int main(int argc, char** argv) {
  const bool sweep = argc < 2 || std::strcmp(argv[1], "conditions") != 0;
  const long ld = 15616;
  const int rows = 48 * 128;
  const size_t n = (size_t)rows * ld;
  std::vector<double> h(n);
  std::mt19937_64 rng(3);
  std::uniform_real_distribution<double> ud(-1.0, 1.0);
  for (size_t i = 0; i < n; ++i) h[i] = ud(rng);
  double *S, *S0, *Ref;
  if (hipMalloc(&S, n * 8) != hipSuccess || hipMalloc(&S0, n * 8) != hipSuccess || hipMalloc(&Ref, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemcpy(S0, h.data(), n * 8, hipMemcpyHostToDevice);
  const Variant variants[] = {
      {"32x128 kPF2 2x2 (shipped)", 32, thin_variant<16, 2, 32, 2, 2>},
      {"32x128 kPF4 2x2", 32, thin_variant<16, 4, 32, 2, 2>},
      {"32x128 kPF8 2x2", 32, thin_variant<16, 8, 32, 2, 2>},
      {"32x128 kPF2 1x4", 32, thin_variant<16, 2, 32, 1, 2>},
      {"32x128 kPF4 1x4", 32, thin_variant<16, 4, 32, 1, 2>},
      {"32x128 kPF8 1x4", 32, thin_variant<16, 8, 32, 1, 2>},
      {"64x128 kPF2 2x2", 64, thin_variant<16, 2, 64, 2, 2>},
      {"64x128 kPF4 2x2", 64, thin_variant<16, 4, 64, 2, 2>},
      {"64x128 kPF4 1x4", 64, thin_variant<16, 4, 64, 1, 2>},
      {"128x128 kPF1 2x2 (shipped 128-row body)", 128, thin_variant<16, 1, 128, 2, 2>},
      {"128x128 kPF2 2x2", 128, thin_variant<16, 2, 128, 2, 2>},
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::vector<double> a, b;
  for (int K : {128, 256}) {
    if (!sweep) break;
    for (int T : {14, 20, 28, 40}) {
      const int c0 = 2;  // the panel is block columns [c0 - K/128, c0); the trailing matrix starts at block (c0, c0)
      double* C = S + (size_t)c0 * 128 * ld + (size_t)c0 * 128;
      const double* A = S + (size_t)c0 * 128 * ld + (size_t)(c0 * 128 - K);
      const int nb = T * (T + 1) / 2;
      const double flops = (double)nb * 2.0 * 128 * 128 * K;
      printf("K = %d, T = %d block rows (%d blocks, %.2f GFlop, C read + written %.1f MB)\n", K, T, nb, flops * 1e-9, nb * 131072.0 * 2e-6);
      bool have_ref = false;
      for (const Variant& v : variants) {
        const int sub = 128 / v.tm, grid = nb * sub;
        (void)hipMemcpy(S, S0, n * 8, hipMemcpyDeviceToDevice);
        hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, C, ld, A, ld, K, T * sub);
        if (hipDeviceSynchronize() != hipSuccess) { printf("  %s: launch failed\n", v.name); return 1; }
        bool same = true;
        if (!have_ref) {
          (void)hipMemcpy(Ref, S, n * 8, hipMemcpyDeviceToDevice);
          have_ref = true;
        } else {
          a.resize((size_t)(T + c0) * 128 * ld);
          b.resize(a.size());
          (void)hipMemcpy(a.data(), S, a.size() * 8, hipMemcpyDeviceToHost);
          (void)hipMemcpy(b.data(), Ref, b.size() * 8, hipMemcpyDeviceToHost);
          same = std::memcmp(a.data(), b.data(), a.size() * 8) == 0;
        }
        float best = 1e9f, sum = 0.f;
        const int reps = 5, per = 20;
        for (int r = 0; r < reps; ++r) {
          (void)hipEventRecord(e0, 0);
          for (int i = 0; i < per; ++i) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, C, ld, A, ld, K, T * sub);
          (void)hipEventRecord(e1, 0);
          (void)hipDeviceSynchronize();
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          ms /= per;
          sum += ms;
          if (ms < best) best = ms;
        }
        printf("  %-42s grid %5d  %7.1f us (mean %7.1f)  %5.1f TFLOP/s  %s\n", v.name, grid, best * 1e3, sum / reps * 1e3, flops / (best * 1e-3) * 1e-12,
               same ? "bitwise equal" : "DIFFERENT");
      }
    }
  }
  // ---- the shipped thin kernel under the conditions of the resident chain: on a stream that keeps off 4 (or 2) CUs per XCD,
  // and beside `nsquat` waiting workgroups of a column launch on a stream that may use every CU but CU 0 of each XCD
  {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t words = (size_t)(ncu + 31) / 32;
    auto masked_stream = [&](int first_cu_per_xcd, int end_cu_per_xcd) {
      std::vector<uint32_t> m(words, 0u);
      for (int i = first_cu_per_xcd * 8; i < std::min(ncu, end_cu_per_xcd * 8); ++i) m[(size_t)i / 32] |= 1u << (i % 32);
      hipStream_t st = nullptr;
      if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, m.data()) != hipSuccess) { printf("no CU masks\n"); exit(1); }
      return st;
    };
    hipStream_t bulk4 = masked_stream(4, ncu / 8), bulk2 = masked_stream(2, ncu / 8), bulk8 = masked_stream(8, ncu / 8), panel_rest = masked_stream(1, ncu / 8), panel4 = masked_stream(1, 4), panel8 = masked_stream(1, 8);
    hipStream_t plain = nullptr, plain2 = nullptr;
    (void)hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking);
    int* sink = nullptr;
    (void)hipMalloc(&sink, 4);
    // ... and beside REAL panel work: the thin-tile update a column launch does (gemm_update_thin_f64_kernel over `nwork` 32 x 128 tiles,
    // K = 128, s_setprio 2) on the panel stream, twice per SYRK as TRSM + next(j) are
    struct Cond { const char* name; hipStream_t syrk; hipStream_t squat; int nsquat; int nwork = 0; };
    const Cond conds[] = {
        {"plain stream, alone", plain, nullptr, 0},
        {"off 4 CUs per XCD, alone", bulk4, nullptr, 0},
        {"off 2 CUs per XCD, alone", bulk2, nullptr, 0},
        {"off 4 CUs per XCD, 112 waiting workgroups anywhere but CU 0", bulk4, panel_rest, 112},
        {"off 4 CUs per XCD, 56 waiting workgroups anywhere but CU 0", bulk4, panel_rest, 56},
        {"off 4 CUs per XCD, 112 waiting workgroups on CUs 1-3 only", bulk4, panel4, 112},
        {"off 8 CUs per XCD, 112 waiting workgroups on CUs 1-7 only", bulk8, panel8, 112},
        {"off 4 CUs per XCD, 2 x 160 thin update tiles anywhere but CU 0", bulk4, panel_rest, 0, 160},
        {"off 4 CUs per XCD, 2 x 160 thin update tiles on CUs 1-3 only", bulk4, panel4, 0, 160},
        {"plain stream, 2 x 160 thin update tiles on another plain stream", plain, plain2, 0, 160},
    };
    const Variant& v = variants[0];
    for (int T : {14, 19, 20, 28}) {
      const int K = 128, c0 = 2;
      double* C = S + (size_t)c0 * 128 * ld + (size_t)c0 * 128;
      const double* A = S + (size_t)c0 * 128 * ld + (size_t)(c0 * 128 - K);
      const int nb = T * (T + 1) / 2, grid = nb * 4;
      const double flops = (double)nb * 2.0 * 128 * 128 * K;
      printf("shipped thin kernel, K = 128, T = %d, in the chain's conditions\n", T);
      for (const Cond& c : conds) {
        float best = 1e9f, sum = 0.f;
        const int reps = 5, per = 20;
        for (int r = 0; r < reps; ++r) {
          (void)hipDeviceSynchronize();
          (void)hipEventRecord(e0, c.syrk);
          for (int i = 0; i < per; ++i) {
            if (c.nsquat) hipLaunchKernelGGL(squat_kernel, dim3(c.nsquat), dim3(256), 0, c.squat, (long long)4000, sink);  // 40 us of the 100 MHz clock
            if (c.nwork) {  // (rows far below the SYRK's: no overlap of the data)
              // 40 tile rows (block rows 36..45 of the 48) x nwork / 40 tile columns: inside the allocation
              double* Cw = S + (size_t)36 * 128 * ld + (size_t)1 * 128;
              const double* Aw = S + (size_t)36 * 128 * ld;
              for (int rep = 0; rep < 2; ++rep)
                hipLaunchKernelGGL(sk::gemm_update_thin_f64_kernel, dim3(c.nwork), dim3(256), 0, c.squat, Cw, ld, Aw, ld, Aw, ld, 128, 40, 0, 0x7fffffff, 0);
            }
            hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, c.syrk, C, ld, A, ld, K, T * 4);
          }
          (void)hipEventRecord(e1, c.syrk);
          (void)hipDeviceSynchronize();
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          ms /= per;
          sum += ms;
          if (ms < best) best = ms;
        }
        printf("  %-62s %7.1f us (mean %7.1f)  %5.1f TFLOP/s\n", c.name, best * 1e3, sum / reps * 1e3, flops / (best * 1e-3) * 1e-12);
      }
    }
  }
  // ---- cold caches: in the factorisation a SYRK's C tiles were last written by the SYRK of the step before — other workgroups, other
  // XCDs (the grid shrinks from step to step) — and its panel by a column launch; here every launch is preceded by a pass over 1 GB
  // that evicts the L2s and the Infinity Cache, and timed by its own start / stop events
  {
    hipEvent_t es[40];
    for (auto& e : es) (void)hipEventCreate(&e);
    double* trash = nullptr;
    const size_t trash_n = (size_t)1 << 27;  // 1 GB
    if (hipMalloc(&trash, trash_n * 8) == hipSuccess) {
      const Variant& v = variants[0];
      for (int T : {14, 19, 28}) {
        const int K = 128, c0 = 2;
        double* C = S + (size_t)c0 * 128 * ld + (size_t)c0 * 128;
        const double* A = S + (size_t)c0 * 128 * ld + (size_t)(c0 * 128 - K);
        const int nb = T * (T + 1) / 2, grid = nb * 4;
        const double flops = (double)nb * 2.0 * 128 * 128 * K;
        for (int cold = 0; cold < 2; ++cold) {
          float sum = 0.f;
          for (int i = 0; i < 20; ++i) {
            if (cold) (void)hipMemsetAsync(trash, i, trash_n * 8, 0);
            hipExtLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, es[2 * i], es[2 * i + 1], 0, C, ld, A, ld, K, T * 4);
          }
          (void)hipDeviceSynchronize();
          for (int i = 0; i < 20; ++i) { float ms; (void)hipEventElapsedTime(&ms, es[2 * i], es[2 * i + 1]); sum += ms; }
          printf("T = %2d, %s: %7.1f us per launch (own events), %5.1f TFLOP/s\n", T, cold ? "after a 1 GB memset (cold L2 and Infinity Cache)" : "back to back (warm)                            ",
                 sum / 20 * 1e3, flops / (sum / 20 * 1e-3) * 1e-12);
        }
      }
    }
  }
  return 0;
}
