// Micro-benchmark (round 6): how many bytes per clock can ONE CU take through LDS-DMA (global_load_lds_dwordx4) from L2-resident data,
// by access pattern and by the number of issuing waves?  One workgroup per CU; each wave loops over 1-KiB pieces into a 64-KiB LDS ring
// with `keep` pieces in flight.  Patterns: 0 linear 1 KiB; 1 = 8 rows x 128 B (row stride LD bytes); 2 = the same with the GEMM's XOR
// chunk swizzle; 3 = 16 rows x 64 B; 4 = pattern 2 through buffer_load_dwordx4 ... lds (SGPR resource + 32-bit offsets).
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate ldsdma_rate.hip ; run: ./ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
constexpr int LD = 16384;            // bytes per matrix row (K = 8192 halfs)
template <int PAT, int KEEP>
__global__ __launch_bounds__(512) void k(const unsigned char* src, long region, int iters, int nwaves, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves) return;
  // per-lane offset inside a piece
  long loff;
  if (PAT == 0) loff = lane * 16;
  else if (PAT == 1) loff = (long)(lane >> 3) * LD + (lane & 7) * 16;
  else if (PAT == 2 || PAT == 4) { const int r = lane >> 3; loff = (long)r * LD + (((lane & 7) ^ ((r >> 1) & 7)) * 16); }
  else loff = (long)(lane >> 2) * LD + (lane & 3) * 16;
  const unsigned char* base = src;
  const unsigned long long t0 = clock64();
  const long pos0 = ((long)(blockIdx.x * 8 + wave) * 8 * LD) % (region / 2);   // each wave walks its own rows
  long pos = pos0;
  unsigned char* dst = smem + wave * 8192;
#if 1
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#endif
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long o = pos + loff + j * 128;         // next 128-B column block of the same 8 rows (a k-tile step)
      if (PAT == 0) o = pos + loff + j * 1024;
      if (PAT == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (AS3 void*)(dst + j * 1024), 16, (int)o, 0, 0, 0);
      else __builtin_amdgcn_global_load_lds((const AS1 void*)(base + o), (AS3 void*)(dst + j * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
    }
    pos += 8 * 128;
    if (PAT == 0) pos += 8 * 1024 - 8 * 128;
    if (pos + 8 * LD + 16384 > region) pos = pos0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}
template <int PAT, int KEEP>
void run(const unsigned char* d, long region, int nwaves, unsigned long long* dout, const char* name) {
  const int iters = 2000, cus = 256;
  hipFuncSetAttribute((const void*)k<PAT, KEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<PAT, KEEP><<<cus, 512, 65536>>>(d, region, 200, nwaves, dout);
  hipEventRecord(e0);
  k<PAT, KEEP><<<cus, 512, 65536>>>(d, region, iters, nwaves, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(cus); hipMemcpy(h.data(), dout, cus * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += v; cyc /= cus;
  const double bytes = (double)iters * 8 * 1024 * nwaves;
  printf("%-34s waves %d keep %2d: %6.1f B/clk/CU (s_memtime), %6.1f GB/s/CU, chip %5.2f TB/s, %5.0f cycles per piece per wave\n", name, nwaves, KEEP,
         bytes / cyc, bytes / (ms * 1e6), bytes * cus / (ms * 1e9), cyc / (iters * 8.0));
}
int main() {
  const long region = 24l << 20;      // 24 MiB: L2 / MALL resident across the chip
  unsigned char* d; hipMalloc(&d, region + (1 << 20)); hipMemset(d, 1, region + (1 << 20));
  unsigned long long* dout; hipMalloc(&dout, 256 * 8);
  for (int nw : {8, 4, 2, 1}) {
    run<0, 6>(d, region, nw, dout, "linear 1 KiB");
    run<1, 6>(d, region, nw, dout, "8 rows x 128 B");
    run<2, 6>(d, region, nw, dout, "8 rows x 128 B, xor swizzle");
    run<3, 6>(d, region, nw, dout, "16 rows x 64 B");
    run<4, 6>(d, region, nw, dout, "8x128 swizzle, buffer_load lds");
  }
  run<2, 0>(d, region, 8, dout, "8x128 swizzle keep 0");
  run<2, 2>(d, region, 8, dout, "8x128 swizzle keep 2");
  run<2, 12>(d, region, 8, dout, "8x128 swizzle keep 12");
  return 0;
}
