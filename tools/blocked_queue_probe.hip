// Developer tool (round 3): what OTHER hardware queues cost a chain of dependent launches on one queue.
// NQ CU-masked streams (each owns a hardware queue) are created in order; stream A runs N tiny kernels back to back and is timed
// while one other queue j is (i) idle, (ii) RUNNING a one-wave kernel that spins for a few ms, (iii) BLOCKED on a barrier packet
// (hipStreamWaitEvent on the completion of a spinning kernel on the last queue).  Printed: microseconds per kernel of A.
//   hipcc --offload-arch=gfx950 -O3 tools/blocked_queue_probe.hip -o gpurun_out/blocked_queue_probe && gpurun_out/blocked_queue_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void tiny(double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }
__global__ void spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void noop() {}

int main(int argc, char** argv) {
  const int NQ = argc > 1 ? atoi(argv[1]) : 13;
  const int N = 300;
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  std::vector<uint32_t> all((size_t)(ncu + 31) / 32, 0xffffffffu);
  double* d;
  hipMalloc(&d, 8 * 64);
  hipMemset(d, 0, 8 * 64);
  std::vector<hipStream_t> q(NQ);
  for (auto& s : q)
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)all.size(), all.data()) != hipSuccess) { printf("no CU-masked streams\n"); return 1; }
  hipEvent_t e;
  hipEventCreateWithFlags(&e, hipEventDisableTiming);
  const int spinner = NQ - 1;  // the queue whose spinning kernel the blocked queues wait for
  const long long ticks = 400000;  // 4 ms of the 100 MHz wall clock
  auto run_a = [&](int a) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, q[a], d);
    hipStreamSynchronize(q[a]);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / N;
  };
  for (int a : {0, 1, 2, 5}) {
    if (a >= NQ - 1) continue;
    hipDeviceSynchronize();
    (void)run_a(a);
    hipDeviceSynchronize();
    const double idle = run_a(a);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, q[spinner], ticks);
    const double with_spinner = run_a(a);
    hipDeviceSynchronize();
    printf("A = queue %d: %.2f us per kernel alone; %.2f with a spinning wave on queue %d\n", a, idle, with_spinner, spinner);
    printf("   other queue j:   ");
    for (int j = 0; j < NQ - 1; ++j) if (j != a) printf("%6d", j);
    printf("\n   j RUNNING a spin:");
    for (int j = 0; j < NQ - 1; ++j) {
      if (j == a) continue;
      hipDeviceSynchronize();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, q[j], ticks);
      printf("%6.2f", run_a(a));
    }
    printf("\n   j BLOCKED:       ");
    for (int j = 0; j < NQ - 1; ++j) {
      if (j == a) continue;
      hipDeviceSynchronize();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, q[spinner], ticks);
      hipEventRecord(e, q[spinner]);
      hipStreamWaitEvent(q[j], e, 0);
      hipLaunchKernelGGL(noop, dim3(1), dim3(1), 0, q[j]);
      printf("%6.2f", run_a(a));
    }
    printf("\n   ALL others blocked: ");
    hipDeviceSynchronize();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, q[spinner], ticks);
    hipEventRecord(e, q[spinner]);
    for (int j = 0; j < NQ - 1; ++j) {
      if (j == a) continue;
      hipStreamWaitEvent(q[j], e, 0);
      hipLaunchKernelGGL(noop, dim3(1), dim3(1), 0, q[j]);
    }
    printf("%.2f\n", run_a(a));
    hipDeviceSynchronize();
  }
  // the same with PLAIN streams as the bystanders (they share the runtime's small pool of hardware queues)
  {
    std::vector<hipStream_t> p(4);
    for (auto& s : p) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int a = 0;
    printf("A = masked queue 0, bystanders = plain streams:\n   j BLOCKED:       ");
    for (int j = 0; j < 4; ++j) {
      hipDeviceSynchronize();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, q[spinner], ticks);
      hipEventRecord(e, q[spinner]);
      hipStreamWaitEvent(p[j], e, 0);
      hipLaunchKernelGGL(noop, dim3(1), dim3(1), 0, p[j]);
      printf("%6.2f", run_a(a));
    }
    printf("\n   j RUNNING a spin:");
    for (int j = 0; j < 4; ++j) {
      hipDeviceSynchronize();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, p[j], ticks);
      printf("%6.2f", run_a(a));
    }
    printf("\n");
    hipDeviceSynchronize();
  }
  return 0;
}
