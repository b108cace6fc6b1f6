// Micro-benchmark (round 6): does the access pattern of the GEMM epilogue — a wave reads / writes a 32-row x 32-channel fp32 block as four
// instructions of 32 rows x 32 contiguous bytes (row stride = the tensor's 1280-byte rows) — reach the HBM bandwidth a fully coalesced
// copy does?  [M, 320] fp32, M = 49152 (the 32x32 level of the 24-frame step: 63 MB), read + add + write, and write-only.
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ; run: ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 320;
// coalesced: consecutive lanes take consecutive 16-byte units
template <bool RD>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ a, float* __restrict__ o, long units) {
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    if (RD) v = *reinterpret_cast<const f32x4*>(a + u * 4) + v;
    *reinterpret_cast<f32x4*>(o + u * 4) = v;
  }
}
// epilogue pattern: workgroup = 12 waves on a 192-row x 320-column tile (6 x 2 waves, a wave owns 32 rows x 160 columns = 5 blocks of 32);
// lane l: row l % 32, columns 8 q + 4 (l / 32) .. + 3 of each 32-column block, q = 0 .. 3
template <bool RD>
__global__ __launch_bounds__(768) void k_epi(const float* __restrict__ a, float* __restrict__ o, int M) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / 2, wn = wave % 2;
  const long row = (long)blockIdx.x * 192 + wm * 32 + (lane & 31);
  if (row >= M) return;
  f32x4 v[5][4];
#pragma unroll
  for (int b = 0; b < 5; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = wn * 160 + b * 32 + 8 * q + 4 * (lane >> 5);
      v[b][q] = (f32x4){1.f, 2.f, 3.f, 4.f};
      if (RD) v[b][q] += *reinterpret_cast<const f32x4*>(a + row * N + col);
    }
#pragma unroll
  for (int b = 0; b < 5; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = wn * 160 + b * 32 + 8 * q + 4 * (lane >> 5);
      *reinterpret_cast<f32x4*>(o + row * N + col) = v[b][q];
    }
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}
int main() {
  for (int M : {49152, 12288 * 2, 442368}) {
    const size_t bytes = (size_t)M * N * 4;
    float *a, *o;
    hipMalloc(&a, bytes); hipMalloc(&o, bytes);
    hipMemset(a, 0, bytes);
    const long units = (long)M * N / 4;
    const int g_lin = 256 * 8, g_epi = (M + 191) / 192;
    const float t1 = timeit([&] { hipLaunchKernelGGL(k_linear<true>, dim3(g_lin), dim3(256), 0, 0, a, o, units); });
    const float t2 = timeit([&] { hipLaunchKernelGGL(k_epi<true>, dim3(g_epi), dim3(768), 0, 0, a, o, M); });
    const float t3 = timeit([&] { hipLaunchKernelGGL(k_linear<false>, dim3(g_lin), dim3(256), 0, 0, a, o, units); });
    const float t4 = timeit([&] { hipLaunchKernelGGL(k_epi<false>, dim3(g_epi), dim3(768), 0, 0, a, o, M); });
    printf("M %6d x 320 fp32 (%5.1f MB): read+write coalesced %6.1f us %5.2f TB/s | epilogue pattern %6.1f us %5.2f TB/s || write-only coalesced %6.1f us %5.2f TB/s | epilogue pattern %6.1f us %5.2f TB/s\n",
           M, bytes / 1e6, t1 * 1e3, 2 * bytes / t1 / 1e9, t2 * 1e3, 2 * bytes / t2 / 1e9, t3 * 1e3, bytes / t3 / 1e9, t4 * 1e3, bytes / t4 / 1e9);
    hipFree(a); hipFree(o);
  }
  return 0;
}
