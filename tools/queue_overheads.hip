// Developer tool: what a dependency packet costs the next kernel of a HIP stream on this machine.
// N tiny kernels back to back on one stream, (a) plain, (b) an event record after each, (c) a wait on an event of
// another stream that completed long ago before each, (d) the event attached to the kernel's own dispatch
// (hipExtLaunchKernelGGL stop event), (e) record + satisfied wait (what a look-ahead hand-shake does per block column).
//   hipcc --offload-arch=gfx950 -O3 tools/queue_overheads.hip -o gpurun_out/queue_overheads
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void tiny(double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }

int main() {
  const int N = 400;
  double* d;
  hipMalloc(&d, 8);
  hipMemset(d, 0, 8);
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t old;
  hipEventCreateWithFlags(&old, hipEventDisableTiming);
  hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
  hipEventRecord(old, b);
  hipDeviceSynchronize();
  const char* names[5] = {"plain", "event record after each kernel", "satisfied cross-stream wait before each kernel",
                          "stop event on the kernel's own dispatch", "record + satisfied wait (per-column hand-shake)"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (mode == 2 || mode == 4) hipStreamWaitEvent(a, old, 0);
        if (mode == 3) hipExtLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, nullptr, ev[i], 0, d);
        else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
        if (mode == 1 || mode == 4) hipEventRecord(ev[i], a);
      }
      hipStreamSynchronize(a);
      const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / N;
      if (rep == 1) printf("%-52s %6.2f us per kernel\n", names[mode], us);
    }
  }
  // (f) a wait on an event of the OTHER stream that is not complete at enqueue time but long complete when the
  // queue reaches it: stream b runs N tiny kernels, each followed by a record; stream a first spins on a slow kernel,
  // then waits for b's event i before its own kernel i.
  std::vector<hipEvent_t> evb(N);
  for (unsigned fl : {(unsigned)hipEventDisableTiming, (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence)}) {
    for (auto& e : evb) hipEventCreateWithFlags(&e, fl);
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d + 0); hipEventRecord(evb[i], b); }
      hipStreamSynchronize(b);  // b is done: every wait below is satisfied when a's queue reaches it (but a real packet)
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) { hipStreamWaitEvent(a, evb[i], 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d); }
      hipStreamSynchronize(a);
      const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / N;
      if (rep == 1) printf("%-52s %6.2f us per kernel (event flags 0x%x)\n", "wait on a completed event of another stream", us, fl);
    }
    // (g) lock-step: b's kernel i may only start after a's kernel i-1 and vice versa (a ping-pong of real dependencies)
    std::vector<hipEvent_t> eva(N);
    for (auto& e : eva) hipEventCreateWithFlags(&e, fl);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      if (i > 0) hipStreamWaitEvent(b, eva[i - 1], 0);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
      hipEventRecord(evb[i], b);
      hipStreamWaitEvent(a, evb[i], 0);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
      hipEventRecord(eva[i], a);
    }
    hipDeviceSynchronize();
    const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / N;
    printf("%-52s %6.2f us per round trip (event flags 0x%x)\n", "ping-pong between two streams", us, fl);
  }
  return 0;
}
