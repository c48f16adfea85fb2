// Developer probe: which compute units does a CU-masked stream really use on gfx950?
// Every workgroup records (XCC id, HW_ID) ; the host prints the distinct (xcc, se, cu) sets
// for a one-bit mask and for its complement.
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
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

int main() {
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
