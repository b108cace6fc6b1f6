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
//   pass 1: per-thread fp32 partial sums over a 64-row chunk -> LDS -> ordered fp64 fold per
//           group -> one partial per (instance, block, group); no atomics (bitwise reproducible)
//   pass 2: one wave per (instance, group): ordered fold of the partials -> {mean, rstd}
//   pass 3: normalise + affine (+SiLU), 4 channels per thread, 8-byte stores
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

constexpr int GN_ROWS_PER_BLOCK = T2V_GN_ROWS_PER_BLOCK;  // rows of one instance reduced by one workgroup

// Pass 1 — grid (nblk, n_inst).  Deterministic: per-thread fp32 partials over <= 64 rows are
// parked in LDS [R][C]; 32 threads then fold replicas + the channels of their group in a fixed
// order in fp64 and store ONE partial per (instance, block, group).  No atomics anywhere, so
// results are bitwise reproducible run to run.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* x, double* partials, int rows, int C, int ld,
                                                       int groups) {
  extern __shared__ float sh[];  // [2][R][C]
  const int tid = threadIdx.x;
  const int inst = blockIdx.y;
  const int r0 = blockIdx.x * GN_ROWS_PER_BLOCK;
  const int r1 = min(rows, r0 + GN_ROWS_PER_BLOCK);
  const T* base = x + ((size_t)inst * rows) * ld;
  const int cv = C >> 2;  // float4 units per row
  const int R = cv <= 256 ? 256 / cv : 1;
  float* psum = sh;
  float* psq = sh + R * C;
  if (cv <= 256) {
    if (tid < R * cv) {
      const int c4 = (tid % cv) * 4, rr = tid / cv;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
      for (int r = r0 + rr; r < r1; r += R) {
        const f32x4 v = Load4<T>::ld(base + (size_t)r * ld + c4);
        s += v;
        q += v * v;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { psum[rr * C + c4 + e] = s[e]; psq[rr * C + c4 + e] = q[e]; }
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
      for (int e = 0; e < 4; ++e) { psum[c4 + e] = s[e]; psq[c4 + e] = q[e]; }
    }
  }
  __syncthreads();
  if (tid < groups) {
    const int cpg = C / groups;
    double s = 0.0, q = 0.0;
    for (int rr = 0; rr < R; ++rr)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { s += (double)psum[rr * C + c]; q += (double)psq[rr * C + c]; }
    double* st = partials + (((size_t)inst * gridDim.x + blockIdx.x) * groups + tid) * 2;
    st[0] = s;
    st[1] = q;
  }
}

// Pass 2 — one wave per (instance, group): ordered fold of the block partials -> {mean, rstd}.
// `nparts` > 1: the partials of all T-shard ranks, gathered as [part][inst][blk][group][2]; the fold
// order (part-major, then block) is the same on every rank, so all ranks get identical statistics.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* partials, float* finals, int n_inst, int nblk,
                                                          int groups, double inv_n, float eps, int nparts) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);  // (inst, group) pair
  if (idx >= n_inst * groups) return;
  const int inst = idx / groups, g = idx - inst * groups;
  double s = 0.0, q = 0.0;
  for (int u = lane; u < nblk * nparts; u += 64) {
    const int part = u / nblk, b = u - part * nblk;
    const double* st = partials + ((((size_t)part * n_inst + inst) * nblk + b) * groups + g) * 2;
    s += st[0];
    q += st[1];
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if (lane == 0) {
    const double m = s * inv_n;
    double var = q * inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    finals[2 * idx] = (float)m;
    finals[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// Pass 3 — grid-stride over float4 units of the whole [n_inst*rows, C] tensor
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* x, const float* finals, const float* gamma,
                                                       const float* beta, f16* out, int n_inst, int rows, int C,
                                                       int ld_in, int ld_out, int groups, int silu) {
  const int cv = C >> 2;
  const int cpg = C / groups;
  const long total = (long)n_inst * rows * cv;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
    const long row = u / cv;
    const int c4 = (int)(u - row * cv) * 4;
    const int inst = (int)(row / rows);
    const f32x4 v = Load4<T>::ld(x + (size_t)row * ld_in + c4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c4);
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int grp = (c4 + e) / cpg;
      const float2 mr = *reinterpret_cast<const float2*>(finals + 2 * ((size_t)inst * groups + grp));
      float y = (v[e] - mr.x) * mr.y * g[e] + b[e];
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

// Scratch (op.p[4], owned by the caller): fp64 partials [nparts][n_inst][nblk][groups][2] followed by
// fp32 finals [n_inst][groups][2], nblk = ceil(rows / T2V_GN_ROWS_PER_BLOCK).
// op.i[8] = phase: 0 = whole op; 1 = statistics only (writes this rank's partials into part op.i[10]);
//                  2 = fold the op.i[9] gathered parts + normalise (after the partials all-gather).
hipError_t t2v_launch_groupnorm(const t2v_op& op, hipStream_t s) {
  const int n_inst = op.i[0], rows = op.i[1], C = op.i[2], ld_in = op.i[3], groups = op.i[4];
  const int in_dt = op.i[5], silu = op.i[6], ld_out = op.i[7];
  const int phase = op.i[8], nparts = op.i[9] > 0 ? op.i[9] : 1, part = op.i[10];
  if (C % 4 != 0 || C % groups != 0 || groups > 256 || n_inst <= 0 || rows <= 0 || op.p[4] == 0 || part < 0 ||
      part >= nparts || phase < 0 || phase > 2)
    return hipErrorInvalidValue;
  const int nblk = (rows + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK;
  const size_t part_len = (size_t)n_inst * nblk * groups * 2;
  double* partials = reinterpret_cast<double*>(op.p[4]);
  float* finals = reinterpret_cast<float*>(partials + part_len * nparts);
  const dim3 g1(nblk, n_inst);
  const int cv = C / 4;
  const int R = cv <= 256 ? 256 / cv : 1;
  const size_t lds = 2 * (size_t)R * C * sizeof(float);
  const long units = (long)n_inst * rows * cv;
  const int g3 = (int)((units + 255) / 256 < 4096 ? (units + 255) / 256 : 4096);
  const int g2 = (n_inst * groups + 3) / 4;
  const double inv_n = 1.0 / ((double)rows * nparts * (C / groups));
  const float* gamma = reinterpret_cast<const float*>(op.p[1]);
  const float* beta = reinterpret_cast<const float*>(op.p[2]);
  f16* out = reinterpret_cast<f16*>(op.p[3]);
  if (in_dt == T2V_F32) {
    const float* x = reinterpret_cast<const float*>(op.p[0]);
    if (phase != 2)
      hipLaunchKernelGGL(gn_stats_kernel<float>, g1, dim3(256), lds, s, x, partials + part_len * part, rows, C, ld_in, groups);
    if (phase != 1) {
      hipLaunchKernelGGL(gn_finalize_kernel, dim3(g2), dim3(256), 0, s, partials, finals, n_inst, nblk, groups, inv_n, op.f[0], nparts);
      hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(g3), dim3(256), 0, s, x, finals, gamma, beta, out, n_inst, rows, C,
                         ld_in, ld_out, groups, silu);
    }
  } else {
    const f16* x = reinterpret_cast<const f16*>(op.p[0]);
    if (phase != 2)
      hipLaunchKernelGGL(gn_stats_kernel<f16>, g1, dim3(256), lds, s, x, partials + part_len * part, rows, C, ld_in, groups);
    if (phase != 1) {
      hipLaunchKernelGGL(gn_finalize_kernel, dim3(g2), dim3(256), 0, s, partials, finals, n_inst, nblk, groups, inv_n, op.f[0], nparts);
      hipLaunchKernelGGL(gn_apply_kernel<f16>, dim3(g3), dim3(256), 0, s, x, finals, gamma, beta, out, n_inst, rows, C,
                         ld_in, ld_out, groups, silu);
    }
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
