// Developer probe: which compute units does a CU-masked stream really use on gfx950?
// Every workgroup records (XCC id, HW_ID) ; the host prints the distinct (xcc, se, cu) sets
// for a one-bit mask and for its complement.
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <string>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // burn a little time so the grid spreads over every CU it is allowed to use
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  if (x == 12345.678) out[0] = 0;
}

static void run(const char* label, const std::vector<uint32_t>& mask, int blocks) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", label); return; }
  unsigned* d; hipMalloc(&d, blocks * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 64 * 1024, s, d, 200000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(blocks * 2);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::set<unsigned> cus;
  for (int b = 0; b < blocks; ++b) {
    const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
  }
  printf("%s: %zu distinct CUs:", label, cus.size());
  int n = 0;
  for (unsigned c : cus) { if (n++ < 12) printf(" x%u.se%u.sh%u.cu%u", c >> 12, (c >> 8) & 0xf, (c >> 4) & 0xf, c & 0xf); }
  printf("%s\n", cus.size() > 12 ? " ..." : "");
  hipFree(d); hipStreamDestroy(s);
}

// VERDICT r02 item 5a / ADVICE r01: "creating a masked stream after destroying one hangs on this ROCm build" — observed in
// round 1 when the factorisation's streams belonged to a solver and were destroyed with it.  The four run() calls below
// already are create -> launch -> synchronise -> destroy cycles of ONE idle masked stream at a time, and they complete
// (profiles/r01_cumask_probe.txt).  cycle() is the situation of the library: three masked streams alive at once (the panel,
// bulk and server masks), a RESIDENT kernel on one of them that waits for a flag another stream's kernel sets, a
// cross-stream event wait — then everything synchronised, every stream destroyed, and the same three created again.
__global__ void wait_flag(volatile int* flag, unsigned* out) {
  long long t0 = wall_clock64();
  while (*flag == 0 && wall_clock64() - t0 < 200000000) __builtin_amdgcn_s_sleep(8);  // (gives up after 2 s)
  if (threadIdx.x == 0) out[0] = (unsigned)*flag;
}
__global__ void set_flag(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

static bool cycle(int ncu, int round) {
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> cu0(words, 0u), rest(words, 0u), bulk(words, 0u);
  cu0[0] = 0xffu;
  for (int i = 8; i < ncu; ++i) rest[i / 32] |= 1u << (i % 32);
  for (int i = 32; i < ncu; ++i) bulk[i / 32] |= 1u << (i % 32);
  hipStream_t server, panel, bk;
  if (hipExtStreamCreateWithCUMask(&server, words, cu0.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&panel, words, rest.data()) != hipSuccess ||
      hipExtStreamCreateWithCUMask(&bk, words, bulk.data()) != hipSuccess) { printf("cycle %d: stream creation failed\n", round); return false; }
  int* flag; unsigned* out;
  hipMalloc(&flag, sizeof(int)); hipMalloc(&out, 4096 * 2 * sizeof(unsigned));
  hipMemset(flag, 0, sizeof(int));
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipLaunchKernelGGL(wait_flag, dim3(1), dim3(256), 150 * 1024, server, flag, out);          // resident: waits for the panel stream's kernel
  hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, bk, out, 20000);                        // a bulk grid next to it
  hipEventRecord(ev, bk);
  hipStreamWaitEvent(panel, ev, 0);                                                           // cross-stream dependency
  hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, panel, flag);
  hipStreamSynchronize(server); hipStreamSynchronize(panel); hipStreamSynchronize(bk);
  unsigned seen = 0;
  hipMemcpy(&seen, out, sizeof(unsigned), hipMemcpyDeviceToHost);
  hipEventDestroy(ev);
  hipStreamDestroy(server); hipStreamDestroy(panel); hipStreamDestroy(bk);
  hipFree(flag); hipFree(out);
  printf("cycle %d: three masked streams created, resident kernel released by another stream, synchronised, destroyed\n", round);
  fflush(stdout);
  return true;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "cycle") {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    hipFuncSetAttribute(reinterpret_cast<const void*>(wait_flag), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 3; ++r) if (!cycle(p.multiProcessorCount, r)) return 1;
    printf("create / destroy / create of CU-masked streams: ok (3 cycles)\n");
    return 0;
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  printf("multiProcessorCount %d\n", ncu);
  std::vector<uint32_t> all(words, 0xffffffffu), one(words, 0), rest(words, 0xffffffffu), first8(words, 0);
  one[(ncu - 1) / 32] = 1u << ((ncu - 1) % 32);
  rest[(ncu - 1) / 32] &= ~(1u << ((ncu - 1) % 32));
  first8[0] = 0xff;
  run("all bits      ", all, 2048);
  run("bit ncu-1 only", one, 256);
  run("all but ncu-1 ", rest, 2048);
  run("bits 0..7     ", first8, 512);
  return 0;
}
