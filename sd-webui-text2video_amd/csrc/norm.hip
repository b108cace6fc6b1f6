// GroupNorm(32)(+SiLU) and LayerNorm on channels-last tokens, gfx950 (wave64).
//
// Replaces nn.GroupNorm(32, C) [+ nn.SiLU] at every site of the reference UNet / VAE
// (SURVEY.md §2.3 K9): ResBlock in/out layers (4-D input => per-frame statistics, eps 1e-5,
// t2v_model.py:927,952), TemporalConvBlock_v2 and TemporalTransformer.norm (5-D input =>
// statistics over ALL frames of a sample, t2v_model.py:1202-1211, :690), SpatialTransformer
// .norm (eps 1e-6, :612), VAE Normalize (eps 1e-6, autoencoder_modules.py:34-35); and
// nn.LayerNorm(C) of BasicTransformerBlock (t2v_model.py:798-800, K10).
//
// A statistics "instance" is a contiguous run of `rows` token rows (one frame, or all F frames
// of one sample) x one of 32 channel groups.  Input is the fp32 (or fp16) residual stream
// [rows, C]; output is fp16 — the operand format of the following MFMA GEMM.
//   pass 1: per-thread fp32 partial sums over a row chunk -> LDS per-channel -> per-group
//           fp64 atomics into stats[inst][group] = {sum, sumsq}
//   pass 2: normalise + affine (+SiLU), 4 channels per thread, 8-byte stores
// HBM-bound: 4 B + 4 B read, 2 B written per element.
#include "t2v_kernels.h"

namespace {

template <typename T> struct Load4;
template <> struct Load4<float> {
  static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
};
template <> struct Load4<f16> {
  static __device__ __forceinline__ f32x4 ld(const f16* p) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return r;
  }
};

constexpr int GN_ROWS_PER_BLOCK = 64;  // rows of one instance reduced by one workgroup

// grid: (ceil(rows / GN_ROWS_PER_BLOCK), n_inst)
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* x, double* stats, int rows, int C, int ld,
                                                       int groups) {
  extern __shared__ float sh[];  // [2][C] per-channel sum / sumsq
  float* csum = sh;
  float* csq = sh + C;
  const int tid = threadIdx.x;
  for (int c = tid; c < 2 * C; c += 256) sh[c] = 0.f;
  __syncthreads();
  const int inst = blockIdx.y;
  const int r0 = blockIdx.x * GN_ROWS_PER_BLOCK;
  const int r1 = min(rows, r0 + GN_ROWS_PER_BLOCK);
  const T* base = x + ((size_t)inst * rows) * ld;
  const int cv = C >> 2;  // float4 units per row
  if (cv <= 256) {
    // R row-replicas, each thread owns one channel quad
    const int R = 256 / cv;
    if (tid < R * cv) {
      const int c4 = (tid % cv) * 4, rr = tid / cv;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
      for (int r = r0 + rr; r < r1; r += R) {
        const f32x4 v = Load4<T>::ld(base + (size_t)r * ld + c4);
        s += v;
        q += v * v;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&csum[c4 + e], s[e]);
        atomicAdd(&csq[c4 + e], q[e]);
      }
    }
  } else {
    for (int u = tid; u < cv; u += 256) {
      const int c4 = u * 4;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
      for (int r = r0; r < r1; ++r) {
        const f32x4 v = Load4<T>::ld(base + (size_t)r * ld + c4);
        s += v;
        q += v * v;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { csum[c4 + e] = s[e]; csq[c4 + e] = q[e]; }
    }
  }
  __syncthreads();
  if (tid < groups) {
    const int cpg = C / groups;
    double s = 0.0, q = 0.0;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { s += (double)csum[c]; q += (double)csq[c]; }
    double* st = stats + ((size_t)inst * groups + tid) * 2;
    atomicAdd(st, s);
    atomicAdd(st + 1, q);
  }
}

// grid-stride over float4 units of the whole [n_inst*rows, C] tensor
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* x, const double* stats, const float* gamma,
                                                       const float* beta, f16* out, int n_inst, int rows, int C,
                                                       int ld_in, int ld_out, int groups, float eps, int silu,
                                                       double* stats_next, int stats_next_len) {
  // housekeeping for the ping-pong statistics scratch: zero the *other* buffer for the next op
  for (int i = blockIdx.x * 256 + threadIdx.x; i < stats_next_len; i += gridDim.x * 256) stats_next[i] = 0.0;
  const int cv = C >> 2;
  const int cpg = C / groups;
  const double inv_n = 1.0 / ((double)rows * cpg);
  const long total = (long)n_inst * rows * cv;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
    const long row = u / cv;
    const int c4 = (int)(u - row * cv) * 4;
    const int inst = (int)(row / rows);
    const f32x4 v = Load4<T>::ld(x + (size_t)row * ld_in + c4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c4);
    f16x4 o;
    int gprev = -1;
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int grp = (c4 + e) / cpg;
      if (grp != gprev) {
        const double* st = stats + ((size_t)inst * groups + grp) * 2;
        const double m = st[0] * inv_n;
        double var = st[1] * inv_n - m * m;
        var = var < 0.0 ? 0.0 : var;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)eps));
        gprev = grp;
      }
      float y = (v[e] - mean) * rstd * g[e] + b[e];
      if (silu) y = t2v_silu(y);
      o[e] = (f16)y;
    }
    *reinterpret_cast<f16x4*>(out + (size_t)row * ld_out + c4) = o;
  }
}

// LayerNorm: one wave per row, row held in registers (C <= 64*4*MAXV)
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* gamma, const float* beta,
                                                        f16* out, int M, int C, int ld_in, int ld_out, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * ld_in;
  const int cv = C >> 2;
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int u = lane + i * 64;
    if (u < cv) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + u * 4);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int u = lane + i * 64;
    if (u < cv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
  f16* yr = out + (size_t)row * ld_out;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int u = lane + i * 64;
    if (u < cv) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + u * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + u * 4);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)((v[i][e] - mean) * rstd * g[e] + b[e]);
      *reinterpret_cast<f16x4*>(yr + u * 4) = o;
    }
  }
}

}  // namespace

// stats scratch layout (fp64): two ping-pong buffers of `T2V_GN_STATS_LEN` doubles each.
// op.i[8] selects the buffer used by THIS op (0/1); its apply pass zeroes the first op.i[9]
// doubles of the other one (the host passes the largest n_inst*groups*2 of the program and
// zeroes both buffers with a MEMSET op at program start).
static constexpr int T2V_GN_STATS_LEN = 4096 * 32 * 2;  // up to 4096 instances x 32 groups

hipError_t t2v_launch_groupnorm(const t2v_op& op, hipStream_t s) {
  const int n_inst = op.i[0], rows = op.i[1], C = op.i[2], ld_in = op.i[3], groups = op.i[4];
  const int in_dt = op.i[5], silu = op.i[6], ld_out = op.i[7], which = op.i[8] & 1;
  if (C % 4 != 0 || C % groups != 0 || groups > 256 || n_inst * groups * 2 > T2V_GN_STATS_LEN)
    return hipErrorInvalidValue;
  double* stats = reinterpret_cast<double*>(op.p[4]) + (size_t)which * T2V_GN_STATS_LEN;
  double* other = reinterpret_cast<double*>(op.p[4]) + (size_t)(which ^ 1) * T2V_GN_STATS_LEN;
  const dim3 g1((rows + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK, n_inst);
  const size_t lds = 2 * (size_t)C * sizeof(float);
  const long units = (long)n_inst * rows * (C / 4);
  const int g2 = (int)((units + 255) / 256 < 4096 ? (units + 255) / 256 : 4096);
  const float* gamma = reinterpret_cast<const float*>(op.p[1]);
  const float* beta = reinterpret_cast<const float*>(op.p[2]);
  f16* out = reinterpret_cast<f16*>(op.p[3]);
  const int zero_len = op.i[9] < T2V_GN_STATS_LEN ? op.i[9] : T2V_GN_STATS_LEN;
  if (in_dt == T2V_F32) {
    const float* x = reinterpret_cast<const float*>(op.p[0]);
    hipLaunchKernelGGL(gn_stats_kernel<float>, g1, dim3(256), lds, s, x, stats, rows, C, ld_in, groups);
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(g2), dim3(256), 0, s, x, stats, gamma, beta, out, n_inst,
                       rows, C, ld_in, ld_out, groups, op.f[0], silu, other, zero_len);
  } else {
    const f16* x = reinterpret_cast<const f16*>(op.p[0]);
    hipLaunchKernelGGL(gn_stats_kernel<f16>, g1, dim3(256), lds, s, x, stats, rows, C, ld_in, groups);
    hipLaunchKernelGGL(gn_apply_kernel<f16>, dim3(g2), dim3(256), 0, s, x, stats, gamma, beta, out, n_inst,
                       rows, C, ld_in, ld_out, groups, op.f[0], silu, other, zero_len);
  }
  return hipGetLastError();
}

hipError_t t2v_launch_layernorm(const t2v_op& op, hipStream_t s) {
  const int M = op.i[0], C = op.i[1], ld_in = op.i[2], ld_out = op.i[3];
  if (C % 4 != 0 || C > 64 * 4 * 8) return hipErrorInvalidValue;
  const float* x = reinterpret_cast<const float*>(op.p[0]);
  const float* gamma = reinterpret_cast<const float*>(op.p[1]);
  const float* beta = reinterpret_cast<const float*>(op.p[2]);
  f16* out = reinterpret_cast<f16*>(op.p[3]);
  const dim3 grid((M + 3) / 4);
  if (C <= 64 * 4 * 2)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, x, gamma, beta, out, M, C, ld_in, ld_out, op.f[0]);
  else if (C <= 64 * 4 * 5)
    hipLaunchKernelGGL(layernorm_kernel<5>, grid, dim3(256), 0, s, x, gamma, beta, out, M, C, ld_in, ld_out, op.f[0]);
  else
    hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, s, x, gamma, beta, out, M, C, ld_in, ld_out, op.f[0]);
  return hipGetLastError();
}
