// Micro-probe (round 6): which compute units does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X (8 XCDs x 32 CUs)?
// For a few masks: launch many one-wave workgroups that each record HW_REG_XCC_ID and the CU / SE fields of HW_REG_HW_ID, print how many
// distinct (XCD, SE, CU) places ran workgroups and the per-XCD histogram.
// build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip ; run: ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const unsigned long long t0 = clock64();
  while (clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
  const int n = 4096;
  unsigned* d;
  hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, s, d, 20000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  int per_xcd[8] = {0};
  std::set<unsigned> places;
  for (int i = 0; i < n; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_xcd[xcc & 7]++;
    places.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
  }
  printf("%-28s places %3zu   workgroups per XCD:", name, places.size());
  for (int x = 0; x < 8; ++x) printf(" %4d", per_xcd[x]);
  printf("\n");
  hipFree(d);
  hipStreamDestroy(s);
}
int main() {
  std::vector<uint32_t> all(8, 0xffffffffu), lo(8, 0), hi(8, 0), even(8, 0), first4of8(8, 0), m16(8, 0);
  for (int i = 0; i < 256; ++i) {
    if (i < 128) lo[i / 32] |= 1u << (i % 32); else hi[i / 32] |= 1u << (i % 32);
    if ((i & 1) == 0) even[i / 32] |= 1u << (i % 32);
    if ((i & 7) < 4) first4of8[i / 32] |= 1u << (i % 32);
    if (i < 16) m16[i / 32] |= 1u << (i % 32);
  }
  run("all 256 bits", all);
  run("bits 0..127", lo);
  run("bits 128..255", hi);
  run("even bits", even);
  run("bits with (i & 7) < 4", first4of8);
  run("bits 0..15", m16);
  return 0;
}
