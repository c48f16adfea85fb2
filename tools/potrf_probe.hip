// Developer tool: run potrf128 alone on one SPD block, check L and L^-1 against a host
// factorisation, and print the per-phase wall-clock stamps of the four waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I skeres_amd/csrc tools/potrf_probe.hip -o gpurun_out/potrf_probe
#define SK_POTRF_STAMPS 1
#include "../skeres_amd/csrc/chol_kernels.hip"
namespace sk { const DevKnobs& dev_knobs() { static DevKnobs k; return k; } }

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

int main() {
  const int n = 128, ld = 15616;
  std::vector<double> G(n * 160), A((size_t)n * ld, 0.0), L(n * n, 0.0);
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  for (auto& g : G) g = nd(rng);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = i == j ? 1.0 : 0.0;
      for (int k = 0; k < 160; ++k) s += G[i * 160 + k] * G[j * 160 + k];
      A[(size_t)i * ld + j] = s;
      L[i * n + j] = s;
    }
  for (int j = 0; j < n; ++j) {  // host Cholesky (lower, in place)
    for (int k = 0; k < j; ++k)
      for (int i = j; i < n; ++i) L[i * n + j] -= L[i * n + k] * L[j * n + k];
    const double d = std::sqrt(L[j * n + j]);
    for (int i = j; i < n; ++i) L[i * n + j] = i == j ? d : L[i * n + j] / d;
  }
  double *dA, *dLinv;
  int* dinfo;
  hipMalloc(&dA, A.size() * 8);
  hipMalloc(&dLinv, n * n * 8);
  hipMalloc(&dinfo, 4);
  hipMemset(dLinv, 0, n * n * 8);
  hipMemset(dinfo, 0, 4);
  if (sk::cholesky_init() != hipSuccess) { printf("init failed\n"); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 20; ++rep) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(sk::potrf128_kernel, dim3(1), dim3(256), sk::potrf128_lds_bytes(), 0, dA, (long)ld, dLinv, dinfo);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<double> A2(A.size()), Li(n * n);
  int info = -1;
  hipMemcpy(A2.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(Li.data(), dLinv, n * n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
  double eL = 0, eI = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) eL = std::fmax(eL, std::fabs(A2[(size_t)i * ld + j] - L[i * n + j]));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += Li[i * n + k] * (k >= j ? L[k * n + j] : 0.0);
      eI = std::fmax(eI, std::fabs(s - (i == j ? 1.0 : 0.0)));
    }
  printf("info %d  max|L - L_host| %.3e  max|Linv L - I| %.3e  best event time %.1f us\n", info, eL, eI, best * 1e3);
  long long st[4][16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(sk::g_potrf_stamps), sizeof(st));
  long long ck[4][16];
  hipMemcpyFromSymbol(ck, HIP_SYMBOL(sk::g_potrf_clk), sizeof(ck));
  printf("shader clock / 100 MHz wall clock over the kernel: %.2f  (=> %.0f MHz)\n", (double)(ck[0][10] - ck[0][0]) / (st[0][10] - st[0][0]),
         100.0 * (ck[0][10] - ck[0][0]) / (st[0][10] - st[0][0]));
  for (int jb = 0; jb < 4; ++jb)
    printf("P(%d): chain %.2f us (%lld shader clocks), write-back %.2f us\n", jb, (st[0][11 + jb] - st[0][jb == 0 ? 1 : 1 + 2 * jb]) * 0.01,
           ck[0][11 + jb] - ck[0][jb == 0 ? 1 : 1 + 2 * jb], (st[0][2 + 2 * jb] - st[0][11 + jb]) * 0.01);
  const char* names[11] = {"start", "loaded", "A0", "B0", "A1", "B1", "A2", "B2", "A3", "B3", "end"};
  for (int w = 0; w < 4; ++w) {
    printf("wave %d:", w);
    for (int i = 1; i <= 10; ++i) if (i != 9) printf("  %s %+.2f", names[i], (st[w][i] - st[0][0]) * 0.01);  // 100 MHz counter -> us
    printf("\n");
  }
  return (info == 0 && eL < 1e-9 && eI < 1e-9) ? 0 : 2;
}
