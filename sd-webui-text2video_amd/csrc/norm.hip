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
//   launch 1: per-thread fp32 partial sums over a chunk of rows (8 rows in flight per thread) -> LDS ->
//             ordered fp64 fold per group -> one partial per (instance, block, group); no atomics.
//   launch 2: one wave per (instance, group): ordered fold of the block partials -> {mean, rstd}
//             (a ticket-elected in-kernel fold was measured: same-line L2 atomics serialise at ~25 ns
//             each and the fence + fold tail costs as much as this 4-5 us launch — not kept).
//   launch 3: normalise + affine (+SiLU): scale/shift per channel built once per workgroup in LDS,
//             then 8 rows x 8 channels per thread, 16-byte loads issued together, 16-byte stores.
// Bitwise reproducible run to run.
// (T-sharded clips: statistics / all-gather of the partials / fold + normalise, see phase below.)
// HBM-bound: 4 B + 4 B read, 2 B written per element.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "t2v_kernels.h"

namespace {

// 8 consecutive channels of one token row as fp32 (one 16-byte load for fp16, two for fp32)
template <typename T> struct Load8;
template <> struct Load8<float> {
  static __device__ __forceinline__ f32x8 ld(const float* __restrict__ p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
  }
};
template <> struct Load8<f16> {
  static __device__ __forceinline__ f32x8 ld(const f16* __restrict__ p) {
    const f16x8 h = *reinterpret_cast<const f16x8*>(p);
    f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (float)h[e];
    return r;
  }
};

// Optional second output of a GroupNorm op (round 5): the RAW input cast to fp16 (+ its low-order image at column `lo`) — the operand of the
// 1x1 skip convolution of a ResBlock whose input the norm reads anyway (t2v_model.py:965: skip_connection(x) beside in_layers(x)); the
// separate cast pass (read 4 B + write 2-4 B per element of the concat tensor) disappears.
struct GnCast {
  f16* out;      // null: none
  int ld, lo;    // leading dimension; column offset of the low-order image (0: none)
};
template <typename V>
__device__ __forceinline__ void gn_cast_store8(const GnCast& c, size_t row, int col, const V& v) {
  f16x8 o, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] = (f16)v[e]; l[e] = (f16)(v[e] - (float)o[e]); }
  f16* dst = c.out + row * c.ld + col;
  *reinterpret_cast<f16x8*>(dst) = o;
  if (c.lo) *reinterpret_cast<f16x8*>(dst + c.lo) = l;
}

__device__ __forceinline__ void gn_store_pair(double* st, double s, double q) {
  st[0] = s;
  st[1] = q;
}

constexpr int GN_UNROLL = 8;   // token rows a thread keeps in flight (HBM-bound: ~48 KiB per CU must be outstanding)

// Launch 1 — grid (nblk, n_inst), nblk = ceil(rows / rpb).  Threads form R row-replicas x TPR column slots of
// 8 channels.  Deterministic: per-thread fp32 partials are parked in LDS [R][C]; `groups` threads then fold
// replicas + the channels of their group in a fixed order in fp64 and store ONE partial per (instance,
// block, group).  No atomics: bitwise reproducible run to run.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* partials, int rows, int C,
                                                       int ld, int groups, int rpb) {
  extern __shared__ float sh[];  // [2][R][C] floats
  const int tid = threadIdx.x;
  const int inst = blockIdx.y;
  const int r0 = blockIdx.x * rpb;
  const int r1 = min(rows, r0 + rpb);
  const T* base = x + ((size_t)inst * rows) * ld;
  const int cv = C >> 3;                       // 8-channel units per row
  const int TPR = cv < 256 ? cv : 256;
  const int R = 256 / TPR;
  const int cs = tid % TPR, rr = tid / TPR;
  float* psum = sh;
  float* psq = sh + R * C;
  if (rr < R) {
    for (int u = cs; u < cv; u += TPR) {
      const int c8 = u * 8;
      f32x8 s, q;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
      for (int r = r0 + rr; r < r1; r += R * GN_UNROLL) {
        f32x8 v[GN_UNROLL];
#pragma unroll
        for (int k = 0; k < GN_UNROLL; ++k) {
          const int rk = r + k * R;
          if (rk < r1) v[k] = Load8<T>::ld(base + (size_t)rk * ld + c8);
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < GN_UNROLL; ++k) { s += v[k]; q += v[k] * v[k]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { psum[rr * C + c8 + e] = s[e]; psq[rr * C + c8 + e] = q[e]; }
    }
  }
  __syncthreads();
  const int cpg = C / groups;
  if (groups * 8 <= 256) {
    // 8 adjacent lanes per group: each folds a fixed strided subset of the R x cpg parked sums, then a fixed xor-tree
    const int g = tid >> 3, sub = tid & 7;
    double s = 0.0, q = 0.0;
    if (g < groups) {
      const int n = R * cpg;
      for (int i = sub; i < n; i += 8) {
        const int k = i / cpg, c = g * cpg + (i - k * cpg);
        s += (double)psum[k * C + c];
        q += (double)psq[k * C + c];
      }
    }
    for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (g < groups && sub == 0) gn_store_pair(partials + (((size_t)inst * gridDim.x + blockIdx.x) * groups + g) * 2, s, q);
  } else if (tid < groups) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < R; ++k)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { s += (double)psum[k * C + c]; q += (double)psq[k * C + c]; }
    gn_store_pair(partials + (((size_t)inst * gridDim.x + blockIdx.x) * groups + tid) * 2, s, q);
  }
}

// Pass 2 — one wave per (instance, group): ordered fold of the block partials -> {mean, rstd}.
// `nparts` > 1: the partials of all T-shard ranks, gathered as [part][inst][blk][group][2]; the fold
// order (part-major, then block) is the same on every rank, so all ranks get identical statistics.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* partials, float* finals, int n_inst, int nblk,
                                                          int groups, double inv_n, float eps, int nparts, double* raw) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);  // (inst, group) pair
  if (idx >= n_inst * groups) return;
  const int inst = idx / groups, g = idx - inst * groups;
  double s = 0.0, q = 0.0;
  const int total = nblk * nparts;
  auto at = [&](int u) {
    const int part = u / nblk, b = u - part * nblk;
    return partials + ((((size_t)part * n_inst + inst) * nblk + b) * groups + g) * 2;
  };
  int u = lane;
  for (; u + 192 < total; u += 256) {   // four strided (one cache line each) loads in flight; same summation order
    const double *p0 = at(u), *p1 = at(u + 64), *p2 = at(u + 128), *p3 = at(u + 192);
    const double a0 = p0[0], b0 = p0[1], a1 = p1[0], b1 = p1[1], a2 = p2[0], b2 = p2[1], a3 = p3[0], b3 = p3[1];
    s += a0; q += b0; s += a1; q += b1; s += a2; q += b2; s += a3; q += b3;
  }
  for (; u < total; u += 64) {
    const double* st = at(u);
    s += st[0];
    q += st[1];
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if (lane == 0 && raw != nullptr) {        // T-shard, local fold: this rank's {sum, sum of squares} for the all-gather
    raw[2 * idx] = s;
    raw[2 * idx + 1] = q;
  } else if (lane == 0) {
    const double m = s * inv_n;
    double var = q * inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    finals[2 * idx] = (float)m;
    finals[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// Phase 3 — statistics from the producing GEMM's epilogue (T2V_EPI_STATS): strips[(inst * nstrips + s)][0 | 1][c] = sum / sum of squares
// of one 32-row strip of column c.  One wave per (instance, group) folds nstrips x cpg pairs in a fixed order in fp64 -> {mean, rstd}.
// raw != nullptr (T-sharded clip, round 6): this rank's {sum, sum of squares} per (instance, group) for the exchange instead of {mean, rstd}.
__global__ __launch_bounds__(256) void gn_finalize_strips_kernel(const float* __restrict__ strips, float* finals, int n_inst, int nstrips,
                                                                 int groups, int cpg, int ldn, double inv_n, float eps, double* raw) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= n_inst * groups) return;
  const int inst = idx / groups, g = idx - inst * groups;
  const float* base = strips + (size_t)inst * nstrips * 2 * ldn + g * cpg;
  const int total = nstrips * cpg;
  double s = 0.0, q = 0.0;
  for (int u = lane; u < total; u += 64) {
    const int st = u / cpg, c = u - st * cpg;
    const float* p = base + (size_t)st * 2 * ldn + c;
    s += (double)p[0];
    q += (double)p[ldn];
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if (lane == 0 && raw != nullptr) {
    raw[2 * idx] = s;
    raw[2 * idx + 1] = q;
  } else if (lane == 0) {
    const double m = s * inv_n;
    double var = q * inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    finals[2 * idx] = (float)m;
    finals[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// The same fold with a whole workgroup per (instance, group), for instances of many strips (cross-frame statistics: 768 strips x 10
// channels at the 32x32 level — one wave per pair walked them in 120 dependent-latency rounds, 46 us; 256 threads keep ~60 independent
// loads each in flight).  Thread t takes strips t, t + 256, ...; fixed reduction order (wave xor-tree, then the 4 waves in order).
__global__ __launch_bounds__(256) void gn_finalize_strips_wg_kernel(const float* __restrict__ strips, float* finals, int nstrips, int groups,
                                                                    int cpg, int ldn, double inv_n, float eps, double* raw) {
  __shared__ double red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int idx = blockIdx.x, inst = idx / groups, g = idx - inst * groups;
  const float* base = strips + (size_t)inst * nstrips * 2 * ldn + g * cpg;
  double s = 0.0, q = 0.0;
  for (int st = tid; st < nstrips; st += 256) {
    const float* p = base + (size_t)st * 2 * ldn;
    float fs = 0.f, fq = 0.f;
    for (int c = 0; c < cpg; ++c) { fs += p[c]; fq += p[ldn + c]; }
    s += (double)fs;
    q += (double)fq;
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if (lane == 0) { red[2 * wave] = s; red[2 * wave + 1] = q; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 4; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
    if (raw != nullptr) { raw[2 * idx] = a; raw[2 * idx + 1] = b; return; }
    const double m = a * inv_n;
    double var = b * inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    finals[2 * idx] = (float)m;
    finals[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// Launch 2 — grid (ceil(rows / (R*GN_UNROLL)), n_inst), same thread layout.  The workgroup first builds
// scale[c] = rstd*gamma[c] and shift[c] = beta[c] - mean*scale[c] for the instance in LDS; every thread then
// normalises GN_UNROLL rows of its 8 channels per column unit (16-byte loads issued together, 16-byte stores).
// parts != nullptr (phase 2 of a T-sharded norm, round 6): the gathered {sum, sum of squares} parts of all ranks ([part][inst][group][2]
// doubles, part stride part_len) are folded HERE, in rank order, by `groups` threads of every workgroup — no finalize launch in front of
// the pass; every rank and every workgroup folds the same values in the same order: identical statistics everywhere.
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ finals,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       f16* __restrict__ out, int rows, int C, int ld_in, int ld_out,
                                                       int groups, int lo_off, GnCast cast, const double* parts, int nparts, long part_len,
                                                       double inv_n, float eps) {
  extern __shared__ float sh[];   // scale[C], shift[C] (+ {mean, rstd}[groups] with parts)
  float* sc = sh;
  float* sf = sh + C;
  const int tid = threadIdx.x;
  const int inst = blockIdx.y;
  const int cv = C >> 3;
  const int cpg = C / groups;
  if (parts != nullptr) {
    float* fin = sh + 2 * C;
    for (int g = tid; g < groups; g += 256) {
      double s = 0.0, q = 0.0;
      for (int pz = 0; pz < nparts; ++pz) {
        const double* st = parts + (size_t)pz * part_len + ((size_t)inst * groups + g) * 2;
        s += st[0];
        q += st[1];
      }
      const double m = s * inv_n;
      double var = q * inv_n - m * m;
      var = var < 0.0 ? 0.0 : var;
      fin[2 * g] = (float)m;
      fin[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      const int grp = c / cpg;
      const float a = fin[2 * grp + 1] * gamma[c];
      sc[c] = a;
      sf[c] = beta[c] - fin[2 * grp] * a;
    }
  } else {
    const float* fin = finals + (size_t)inst * groups * 2;
    for (int c = tid; c < C; c += 256) {
      const int grp = c / cpg;
      const float a = fin[2 * grp + 1] * gamma[c];
      sc[c] = a;
      sf[c] = beta[c] - fin[2 * grp] * a;
    }
  }
  __syncthreads();
  const int TPR = cv < 256 ? cv : 256;
  const int R = 256 / TPR;
  const int cs = tid % TPR, rr = tid / TPR;
  if (rr >= R) return;
  const int r0 = blockIdx.x * (R * GN_UNROLL) + rr;
  const T* xb = x + (size_t)inst * rows * ld_in;
  f16* ob = out + (size_t)inst * rows * ld_out;
  for (int u = cs; u < cv; u += TPR) {
    const int c8 = u * 8;
    f32x8 v[GN_UNROLL];
#pragma unroll
    for (int k = 0; k < GN_UNROLL; ++k) {
      const int rk = r0 + k * R;
      if (rk < rows) v[k] = Load8<T>::ld(xb + (size_t)rk * ld_in + c8);
    }
    const f32x8 a = Load8<float>::ld(sc + c8), b = Load8<float>::ld(sf + c8);
#pragma unroll
    for (int k = 0; k < GN_UNROLL; ++k) {
      const int rk = r0 + k * R;
      if (rk < rows) {
        f16x8 o, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float y = v[k][e] * a[e] + b[e];
          if (SILU) y = t2v_silu(y);
          o[e] = (f16)y;
          l[e] = (f16)(y - (float)o[e]);
        }
        *reinterpret_cast<f16x8*>(ob + (size_t)rk * ld_out + c8) = o;
        if (lo_off) *reinterpret_cast<f16x8*>(ob + (size_t)rk * ld_out + lo_off + c8) = l;     // hi + lo operand split (op.i[16])
        if (cast.out) gn_cast_store8(cast, (size_t)inst * rows + rk, c8, v[k]);
      }
    }
  }
}

// Single-launch GroupNorm for small statistics slices (op.i[12] = 1; the 16x16 / 8x8 / 4x4 levels, where the three
// launches above cost more than the data movement): one workgroup per (instance, group), XCD-contiguous, streams its
// [rows x C/groups] slice twice — sums, then normalise; the second pass hits L2.  4-channel units (C/groups % 4 == 0),
// 4 rows in flight per thread, fp32 per-thread sums folded in fp64 in a fixed order (wave xor-tree, then waves in
// order): bitwise reproducible.
constexpr int GNF_THREADS = 512;
constexpr int GNF_UNROLL = 4;

template <typename T> struct Load4;
template <> struct Load4<float> {
  static __device__ __forceinline__ f32x4 ld(const float* __restrict__ p) { return *reinterpret_cast<const f32x4*>(p); }
};
template <> struct Load4<f16> {
  static __device__ __forceinline__ f32x4 ld(const f16* __restrict__ p) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return r;
  }
};

template <typename T, bool SILU>
__global__ __launch_bounds__(GNF_THREADS) void gn_fused_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, f16* __restrict__ out, int rows,
                                                               int C, int ld_in, int ld_out, int groups, float eps, int lo_off, GnCast cast) {
  __shared__ double red[2 * (GNF_THREADS / 64)];
  __shared__ float stat[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup i runs on XCD i % 8 and neighbouring groups share 128-byte lines of every row: give each XCD (= each
  // private L2) a contiguous range of (instance, group) pairs.
  const int nwg = gridDim.x;
  const int pair = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int inst = pair / groups, g = pair - inst * groups;
  const int cpg = C / groups;
  const int nu = cpg >> 2;                    // 4-channel units per row of this group
  const int R = GNF_THREADS / nu;             // row replicas
  const int cs = tid % nu, rr = tid / nu;
  const bool live = rr < R;
  const int c0 = g * cpg + cs * 4;
  const T* xb = x + (size_t)inst * rows * ld_in + c0;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    for (int r = rr; r < rows; r += R * GNF_UNROLL) {
      f32x4 v[GNF_UNROLL];
#pragma unroll
      for (int k = 0; k < GNF_UNROLL; ++k) {
        const int rk = r + k * R;
        if (rk < rows) v[k] = Load4<T>::ld(xb + (size_t)rk * ld_in);
        else v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < GNF_UNROLL; ++k) { s += v[k]; q += v[k] * v[k]; }
    }
  }
  double ds = (double)s[0] + (double)s[1] + (double)s[2] + (double)s[3];
  double dq = (double)q[0] + (double)q[1] + (double)q[2] + (double)q[3];
  for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o); dq += __shfl_xor(dq, o); }
  if (lane == 0) { red[2 * wave] = ds; red[2 * wave + 1] = dq; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < GNF_THREADS / 64; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
    const double inv_n = 1.0 / ((double)rows * cpg);
    const double m = a * inv_n;
    double var = b * inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    stat[0] = (float)m;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (!live) return;
  const float mean = stat[0], rstd = stat[1];
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c0), bt = *reinterpret_cast<const f32x4*>(beta + c0);
  f32x4 sc, sf;
#pragma unroll
  for (int e = 0; e < 4; ++e) { sc[e] = rstd * gm[e]; sf[e] = bt[e] - mean * sc[e]; }
  f16* ob = out + (size_t)inst * rows * ld_out + c0;
  for (int r = rr; r < rows; r += R * GNF_UNROLL) {
    f32x4 v[GNF_UNROLL];
#pragma unroll
    for (int k = 0; k < GNF_UNROLL; ++k) {
      const int rk = r + k * R;
      if (rk < rows) v[k] = Load4<T>::ld(xb + (size_t)rk * ld_in);
    }
#pragma unroll
    for (int k = 0; k < GNF_UNROLL; ++k) {
      const int rk = r + k * R;
      if (rk < rows) {
        f16x4 o, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y = v[k][e] * sc[e] + sf[e];
          if (SILU) y = t2v_silu(y);
          o[e] = (f16)y;
          l[e] = (f16)(y - (float)o[e]);
        }
        *reinterpret_cast<f16x4*>(ob + (size_t)rk * ld_out) = o;
        if (lo_off) *reinterpret_cast<f16x4*>(ob + (size_t)rk * ld_out + lo_off) = l;
        if (cast.out) {
          f16x4 co, cl;
#pragma unroll
          for (int e = 0; e < 4; ++e) { co[e] = (f16)v[k][e]; cl[e] = (f16)(v[k][e] - (float)co[e]); }
          f16* cd = cast.out + ((size_t)inst * rows + rk) * cast.ld + c0;
          *reinterpret_cast<f16x4*>(cd) = co;
          if (cast.lo) *reinterpret_cast<f16x4*>(cd + cast.lo) = cl;
        }
      }
    }
  }
}


// ---- single-pass cooperative GroupNorm (op.i[15] = 1) ---------------------------------------------------------------------
// The three launches above read the tensor twice (4 + 4 B read, 2 B written per element at 2.6 TB/s aggregate) and pay two
// launch boundaries per site.  Here the grid is at most ONE workgroup per CU, every workgroup loads its rows x C chunk ONCE into
// registers (KR rows x 8 channels per thread: the chip's register files hold the whole tensor, 63 MB at the 32x32 level against
// 128 MB of VGPRs), publishes its {sum, sum of squares} partial per group, meets the others at a grid-wide barrier, folds the
// partials of its instance (every workgroup the same values in the same order: bit-identical statistics everywhere, no atomics
// on data), and normalises from registers: 4 B read + 2 B written per element, one launch.
// Co-residency: grid <= number of CUs and <= half a CU's threads / registers per workgroup, so all workgroups of the launch
// are resident once earlier kernels on the GPU drain (a spinning workgroup never waits for one that cannot be scheduled).
// With MORE than two processes sharing one GPU that guarantee is gone — such set-ups (the one-GPU multi-process rehearsals)
// select the three-launch path with T2V_GN_COOP=0.
constexpr int GNC_THREADS = 512;

// The grid barrier (sense-reversing, two levels, bounded wait) is t2v_grid_barrier of t2v_kernels.h, shared with the fused-norm GEMM epilogues.
template <typename T, bool SILU, int KR>
__global__ __launch_bounds__(GNC_THREADS) void gn_coop_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, f16* __restrict__ out, double* partials,
                                                              unsigned* bar, unsigned* fault, int rows, int C, int ld_in, int ld_out,
                                                              int groups, int nchunk, int rc, double inv_n, float eps, int lo_off,
                                                              unsigned seq, unsigned want, GnCast cast) {
  extern __shared__ float sh[];            // phase 1: parked sums [2][R][C]; phase 2: scale[C] | shift[C]
  __shared__ float stat[2 * 32];           // {mean, rstd} per group (groups <= 32)
  const int tid = threadIdx.x;
  const int inst = blockIdx.x / nchunk, chunk = blockIdx.x - inst * nchunk;
  const int cv = C >> 3;
  const int R = GNC_THREADS / cv;
  const int cs = tid % cv, rr = tid / cv;
  const bool live = rr < R;
  const int r0 = chunk * rc, r1 = min(rows, r0 + rc);
  const int c8 = cs * 8;
  const T* xb = x + (size_t)inst * rows * ld_in + c8;
  const bool tags = seq != 0u;            // tagged records instead of the grid barrier (t2v_kernels.h)
  unsigned gen0 = 0;
  if (!tags && tid == 0) gen0 = t2v_grid_epoch(bar);
  f32x8 v[KR];
  f32x8 s, q;
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  if (live) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int r = r0 + rr + k * R;
      if (r < r1) v[k] = Load8<T>::ld(xb + (size_t)r * ld_in);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) { s += v[k]; q += v[k] * v[k]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[rr * C + c8 + e] = s[e]; sh[(R + rr) * C + c8 + e] = q[e]; }
  }
  __syncthreads();
  const int cpg = C / groups;
  const int g = tid >> 4, sub = tid & 15;          // 16 lanes per group (groups <= 32)
  {
    double ds = 0.0, dq = 0.0;
    if (g < groups) {
      const int n = R * cpg;
      for (int i = sub; i < n; i += 16) {
        const int k = i / cpg, c = g * cpg + (i - k * cpg);
        ds += (double)sh[k * C + c];
        dq += (double)sh[(R + k) * C + c];
      }
    }
    for (int o = 1; o < 16; o <<= 1) { ds += __shfl_xor(ds, o); dq += __shfl_xor(dq, o); }
    if (g < groups && sub == 0) {
      double* st = partials + (((size_t)inst * nchunk + chunk) * groups + g) * 2;
      t2v_st_dev(reinterpret_cast<float*>(st), t2v_rec_pack(ds, dq, seq));           // one 16-byte device-scope (write-through) store,
      if (!tags) t2v_wait_vm0();                                                     // (barrier mode) complete before this workgroup arrives
    }
  }
  if (!tags) t2v_grid_barrier(bar, gridDim.x, gen0, fault);
  {
    double ds = 0.0, dq = 0.0;
    if (g < groups) {
      // fixed order (the same on every workgroup of the instance); EIGHT 16-byte device-scope loads in flight per round —
      // one load-use round trip at a time made this fold the longest phase of the kernel (8 dependent ~2 us latencies)
      const double* base = partials + ((size_t)inst * nchunk * groups + g) * 2;
      for (int c = sub; c < nchunk; c += 128) {
        f32x4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nrec = min(8, (nchunk - c + 15) / 16);
        t2v_rec_fetch8([&](int j) { return reinterpret_cast<const float*>(base + (size_t)(c + 16 * j) * groups * 2); }, nrec, want, fault, t);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nrec) t2v_rec_add(t[j], ds, dq);
      }
    }
    for (int o = 1; o < 16; o <<= 1) { ds += __shfl_xor(ds, o); dq += __shfl_xor(dq, o); }
    if (g < groups && sub == 0) {
      const double m = ds * inv_n;
      double var = dq * inv_n - m * m;
      var = var < 0.0 ? 0.0 : var;
      stat[2 * g] = (float)m;
      stat[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();                                            // (also: everyone is done with the parked sums)
  for (int c = tid; c < C; c += GNC_THREADS) {
    const int grp = c / cpg;
    const float a = stat[2 * grp + 1] * gamma[c];
    sh[c] = a;
    sh[C + c] = beta[c] - stat[2 * grp] * a;
  }
  __syncthreads();
  if (!live) return;
  const f32x8 a = Load8<float>::ld(sh + c8), b = Load8<float>::ld(sh + C + c8);
  f16* ob = out + (size_t)inst * rows * ld_out + c8;
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int r = r0 + rr + k * R;
    if (r < r1) {
      f16x8 o, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = v[k][e] * a[e] + b[e];
        if (SILU) y = t2v_silu(y);
        o[e] = (f16)y;
        l[e] = (f16)(y - (float)o[e]);
      }
      *reinterpret_cast<f16x8*>(ob + (size_t)r * ld_out) = o;
      if (lo_off) *reinterpret_cast<f16x8*>(ob + (size_t)r * ld_out + lo_off) = l;
      if (cast.out) gn_cast_store8(cast, (size_t)inst * rows + r, c8, v[k]);
    }
  }
}

// Host-side state of the cooperative path: the fault word (host-mapped, one per process), and per (instantiation, device) the
// number of workgroups of that kernel the occupancy API allows per CU.
struct CoopState {
  unsigned* fault = nullptr;     // hipHostMalloc'ed, mapped: written by a timed-out waiter, read by the host
  bool tried = false, disabled = false, reported = false, peer_reported = false;
};
CoopState g_coop;

unsigned* coop_fault_word() {
  if (!g_coop.tried) {
    g_coop.tried = true;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p != nullptr) {
      g_coop.fault = static_cast<unsigned*>(p);
      g_coop.fault[0] = 0u;      // word 0: a fused-norm wait gave up;  word 1: a peer-window wait gave up (comm.hip)
      g_coop.fault[1] = 0u;
    } else {
      (void)hipGetLastError();
      g_coop.disabled = true;      // no way to report a timeout -> never take the barrier path
    }
  }
  return g_coop.fault;
}

// true when the launch may rely on co-residency: the flag word exists, no earlier fault, and `nwg` workgroups of this
// instantiation fit the stream's device at the occupancy the runtime computes for it (registers, LDS, waves)
template <typename K>
bool coop_fits(K kernel, int nwg, size_t lds, hipStream_t s, int* cache) {
  if (g_coop.disabled || coop_fault_word() == nullptr) return false;
  if (__atomic_load_n(g_coop.fault, __ATOMIC_RELAXED) != 0u) { g_coop.disabled = true; return false; }
  return t2v_grid_fits(reinterpret_cast<const void*>(kernel), GNC_THREADS, lds, nwg, s, cache);
}

template <typename T, bool SILU>
bool gn_coop_launch(int kr, dim3 grid, size_t lds, hipStream_t s, const T* x, const float* gamma, const float* beta, f16* out,
                    double* partials, unsigned* bar, int rows, int C, int ld_in, int ld_out, int groups, int nchunk, int rc, double inv_n,
                    float eps, int lo_off, double* records, GnCast cast) {
  // tagged records need a region that only ever holds records (a recycled arena block could hold a matching bit pattern by chance):
  // op.p[7], the program's exchange scratch; without it the grid barrier synchronises and the per-op scratch carries the partials
  unsigned seq = 0, want = 0;
#define GNC_CASE(K)                                                                                                              \
  case K: {                                                                                                                      \
    static int occ[T2V_MAX_DEVICES] = {};                                                                                        \
    auto kern = gn_coop_kernel<T, SILU, K>;                                                                                      \
    if (!coop_fits(kern, (int)grid.x, lds, s, occ)) return false;                                                                \
    if (records != nullptr) t2v_exchange_ids(&seq, &want);                                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(GNC_THREADS), lds, s, x, gamma, beta, out, seq != 0u ? records : partials, bar,           \
                       g_coop.fault, rows, C, ld_in, ld_out, groups, nchunk, rc, inv_n, eps, lo_off, seq, want, cast);           \
    return true;                                                                                                                 \
  }
  switch (kr) {
    GNC_CASE(4) GNC_CASE(8) GNC_CASE(12) GNC_CASE(16) GNC_CASE(20)
    default: break;
  }
#undef GNC_CASE
  return false;
}

// ---- split-K reduction + GroupNorm (+SiLU) in ONE cooperative launch (T2V_EPI_GN on a split-K GEMM, round 5) ------------------------
// The long-K convolutions of the 8x8 / 4x4 levels run split-K: fp32 slabs, then splitk_reduce_kernel (sum + bias + row bias + residual
// -> the result), then the GroupNorm that consumes it (one more launch, one more read).  Here the reduction IS the loader of the
// single-pass GroupNorm above: every thread sums its rows x 8 channels over the slabs into registers (+ bias / row bias / residual),
// stores the fp32 (fp16) result only if someone else reads it, and the statistics / barrier / normalise phases are gn_coop_kernel's.
template <bool SILU, int KR>
__global__ __launch_bounds__(GNC_THREADS) void splitk_gn_kernel(const GemmParams p, int nchunk, int rc) {
  extern __shared__ float sh[];            // phase 1: parked sums [2][R][C]; phase 2: scale[C] | shift[C]
  __shared__ float stat[2 * 32];
  const int tid = threadIdx.x;
  const int C = p.N, rows = p.gn_rows, groups = C / p.gn_cpg;
  const int inst = blockIdx.x / nchunk, chunk = blockIdx.x - inst * nchunk;
  const int cv = C >> 3;
  const int R = GNC_THREADS / cv;
  const int cs = tid % cv, rr = tid / cv;
  const bool live = rr < R;
  const int r0 = chunk * rc, r1 = min(rows, r0 + rc);
  const int c8 = cs * 8;
  const bool tags = p.gn_seq != 0u;
  unsigned gen0 = 0;
  if (!tags && tid == 0) gen0 = t2v_grid_epoch(p.gn_bar);
  f32x8 v[KR];
  f32x8 s, q;
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  if (live) {
    f32x8 cb;
#pragma unroll
    for (int e = 0; e < 8; ++e) cb[e] = 0.f;
    if (p.bias) cb = Load8<float>::ld(p.bias + c8);
    const size_t zs = (size_t)p.M * p.N;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int r = r0 + rr + k * R;
      if (r < r1) {
        const size_t m = (size_t)inst * rows + r;
        const float* col = p.ws + m * p.N + c8;
        f32x8 a = Load8<float>::ld(col);
        for (int z = 1; z < p.splitk; ++z) a += Load8<float>::ld(col + (size_t)z * zs);      // split order, as the reduction kernel
        a += cb;
        if (p.rowbias) a += Load8<float>::ld(p.rowbias + (m / p.rows_per_batch) * p.ldrb + c8);
        if (p.res) a += Load8<float>::ld(p.res + m * p.ldr + c8);
        if (p.gn_store_out) {
          if (p.out_f32) {
            float* dst = reinterpret_cast<float*>(p.out) + m * p.ldc + c8;
            *reinterpret_cast<f32x4*>(dst) = f32x4{a[0], a[1], a[2], a[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{a[4], a[5], a[6], a[7]};
          } else {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)a[e];
            *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + m * p.ldc + c8) = o;
          }
        }
        v[k] = a;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) { s += v[k]; q += v[k] * v[k]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[rr * C + c8 + e] = s[e]; sh[(R + rr) * C + c8 + e] = q[e]; }
  }
  __syncthreads();
  const int cpg = p.gn_cpg;
  const int g = tid >> 4, sub = tid & 15;          // 16 lanes per group (groups <= 32)
  double* partials = p.gn_part;
  {
    double ds = 0.0, dq = 0.0;
    if (g < groups) {
      const int n = R * cpg;
      for (int i = sub; i < n; i += 16) {
        const int k = i / cpg, c = g * cpg + (i - k * cpg);
        ds += (double)sh[k * C + c];
        dq += (double)sh[(R + k) * C + c];
      }
    }
    for (int o = 1; o < 16; o <<= 1) { ds += __shfl_xor(ds, o); dq += __shfl_xor(dq, o); }
    if (g < groups && sub == 0) {
      double* st = partials + (((size_t)inst * nchunk + chunk) * groups + g) * 2;
      t2v_st_dev(reinterpret_cast<float*>(st), t2v_rec_pack(ds, dq, p.gn_seq));
      if (!tags) t2v_wait_vm0();
    }
  }
  if (!tags) t2v_grid_barrier(p.gn_bar, gridDim.x, gen0, p.gn_fault);
  {
    double ds = 0.0, dq = 0.0;
    if (g < groups) {
      const double* base = partials + ((size_t)inst * nchunk * groups + g) * 2;
      for (int c = sub; c < nchunk; c += 128) {
        f32x4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nrec = min(8, (nchunk - c + 15) / 16);
        t2v_rec_fetch8([&](int j) { return reinterpret_cast<const float*>(base + (size_t)(c + 16 * j) * groups * 2); }, nrec, p.gn_want, p.gn_fault, t);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nrec) t2v_rec_add(t[j], ds, dq);
      }
    }
    for (int o = 1; o < 16; o <<= 1) { ds += __shfl_xor(ds, o); dq += __shfl_xor(dq, o); }
    if (g < groups && sub == 0) {
      const double inv_n = 1.0 / ((double)rows * cpg);
      const double m = ds * inv_n;
      double var = dq * inv_n - m * m;
      var = var < 0.0 ? 0.0 : var;
      stat[2 * g] = (float)m;
      stat[2 * g + 1] = (float)(1.0 / sqrt(var + (double)p.gn_eps));
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += GNC_THREADS) {
    const int grp = c / cpg;
    const float a = stat[2 * grp + 1] * p.gn_gb[c];
    sh[c] = a;
    sh[C + c] = p.gn_gb[C + c] - stat[2 * grp] * a;
  }
  __syncthreads();
  if (!live) return;
  const f32x8 a = Load8<float>::ld(sh + c8), b = Load8<float>::ld(sh + C + c8);
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int r = r0 + rr + k * R;
    if (r < r1) {
      f16x8 o, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = v[k][e] * a[e] + b[e];
        if (SILU) y = t2v_silu(y);
        o[e] = (f16)y;
        l[e] = (f16)(y - (float)o[e]);
      }
      f16* dst = p.gn_out + ((size_t)inst * rows + r) * p.ld_gn + c8;
      *reinterpret_cast<f16x8*>(dst) = o;
      if (p.gn_lo) *reinterpret_cast<f16x8*>(dst + p.gn_lo) = l;
    }
  }
}

// LayerNorm: one wave per row, row held in registers (C <= 64*4*MAXV).  A wave walks several rows (grid-stride): gamma / beta
// stay in registers and the NEXT row's loads are issued before the current row's two reductions, so the wave always has a
// row in flight (12 k single-row workgroups per launch were latency-bound: 3.8 TB/s at the 32x32 level).
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* x, const float* gamma, const float* beta,
                                                             f16* out, int M, int C, int ld_in, int ld_out, float eps) {
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cv = C >> 2;
  f32x4 g[MAXV], b[MAXV], v[MAXV], nx[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int u = lane + i * 64;
    if (u < cv) {
      g[i] = *reinterpret_cast<const f32x4*>(gamma + u * 4);
      b[i] = *reinterpret_cast<const f32x4*>(beta + u * 4);
      nx[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * ld_in + u * 4);
    }
  }
  const float inv_c = 1.0f / (float)C;
  for (; row < M; row += nwaves) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) v[i] = nx[i];
    const int nrow = row + nwaves;
    if (nrow < M) {                                   // wave-uniform
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int u = lane + i * 64;
        if (u < cv) nx[i] = *reinterpret_cast<const f32x4*>(x + (size_t)nrow * ld_in + u * 4);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + i * 64 < cv) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + i * 64 < cv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
      }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * inv_c + eps);
    f16* yr = out + (size_t)row * ld_out;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int u = lane + i * 64;
      if (u < cv) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((v[i][e] - mean) * rstd * g[i][e] + b[i][e]);
        *reinterpret_cast<f16x4*>(yr + u * 4) = o;
      }
    }
  }
}

// Wide rows (C > 768: the 8x8 / 4x4 levels, few rows): one row per wave, no register double-buffering.
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

int t2v_num_cus(hipStream_t s) {
  static int ncu[T2V_MAX_DEVICES] = {};
  const int d = t2v_device_of(s);
  if (ncu[d] == 0) {
    hipDeviceProp_t prop;
    ncu[d] = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
  }
  return ncu[d];
}

// rows per thread of the split-K + GroupNorm launch for this geometry on the stream's device (0: no co-resident grid exists)
static int splitk_gn_kr(const GemmParams& p, hipStream_t s, int* rc_out, int* nchunk_out) {
  const int cv = p.N / 8;
  if (p.N % 8 != 0 || cv > GNC_THREADS || p.gn_cpg <= 0 || p.N / p.gn_cpg > 32 || p.gn_rows <= 0 || p.M % p.gn_rows != 0) return 0;
  const int Rc = GNC_THREADS / cv, ncu = t2v_num_cus(s), n_inst = p.M / p.gn_rows;
  // the FEWEST rows per thread whose grid is still resident at once: the loader sums `splitk` slabs per element, and at the 4x4 level
  // (768 rows) 4 rows per thread left 64 workgroups to read 8 slabs — measured slower than the reduction kernel + the norm it replaces
  for (int kr : {1, 2, 4, 8, 12, 16, 20}) {
    const int rc = Rc * kr, nchunk = (p.gn_rows + rc - 1) / rc;
    if ((long)n_inst * nchunk <= ncu) { *rc_out = rc; *nchunk_out = nchunk; return kr; }
  }
  return 0;
}

hipError_t t2v_launch_splitk_reduce_gn(const GemmParams& p, hipStream_t s) {
  int rc = 0, nchunk = 0;
  const int kr = splitk_gn_kr(p, s, &rc, &nchunk);
  if (kr == 0 || !t2v_coop_allowed() || p.epi != T2V_EPI_NONE || p.act != 0 || p.bias_m) return hipErrorCooperativeLaunchTooLarge;
  const int cv = p.N / 8, n_inst = p.M / p.gn_rows;
  const size_t lds = (size_t)2 * (GNC_THREADS / cv) * p.N * sizeof(float);
  const dim3 grid(n_inst * nchunk);
#define SKG_CASE(K)                                                                                                         \
  case K: {                                                                                                                 \
    static int occ_s[T2V_MAX_DEVICES] = {}, occ_n[T2V_MAX_DEVICES] = {};                                                    \
    if (p.gn_silu) {                                                                                                        \
      auto kern = splitk_gn_kernel<true, K>;                                                                                \
      if (!t2v_grid_fits(reinterpret_cast<const void*>(kern), GNC_THREADS, lds, grid.x, s, occ_s)) return hipErrorCooperativeLaunchTooLarge; \
      hipLaunchKernelGGL(kern, grid, dim3(GNC_THREADS), lds, s, p, nchunk, rc);                                             \
    } else {                                                                                                                \
      auto kern = splitk_gn_kernel<false, K>;                                                                               \
      if (!t2v_grid_fits(reinterpret_cast<const void*>(kern), GNC_THREADS, lds, grid.x, s, occ_n)) return hipErrorCooperativeLaunchTooLarge; \
      hipLaunchKernelGGL(kern, grid, dim3(GNC_THREADS), lds, s, p, nchunk, rc);                                             \
    }                                                                                                                       \
    return hipGetLastError();                                                                                               \
  }
  switch (kr) {
    SKG_CASE(1) SKG_CASE(2) SKG_CASE(4) SKG_CASE(8) SKG_CASE(12) SKG_CASE(16) SKG_CASE(20)
    default: break;
  }
#undef SKG_CASE
  return hipErrorInvalidValue;
}

// Sequence numbers of the tagged-record exchange: process-wide, never 0, never repeated within 2^32 fused-norm launches (a record slot
// is rewritten by every launch that uses it, long before the numbers wrap).  T2V_EXCHANGE=barrier selects the grid barrier instead
// (seq = want = 0).  t2v_debug_poison_exchange(n): the next n launches expect a number nobody publishes — the bounded wait's test.
static unsigned g_seq = 0;
static int g_poison = 0;
static int g_exchange_mode = -1;      // 1 = tagged records, 0 = grid barrier
void t2v_exchange_ids(unsigned* seq, unsigned* want) {
  if (g_exchange_mode < 0) {
    const char* e = getenv("T2V_EXCHANGE");
    g_exchange_mode = (e != nullptr && strcmp(e, "barrier") == 0) ? 0 : 1;
  }
  if (g_exchange_mode == 0) { *seq = *want = 0u; return; }
  unsigned v = __atomic_add_fetch(&g_seq, 1u, __ATOMIC_RELAXED);
  if (v == 0u) v = __atomic_add_fetch(&g_seq, 1u, __ATOMIC_RELAXED);
  *seq = v;
  *want = v;
  if (__atomic_load_n(&g_poison, __ATOMIC_RELAXED) > 0) {
    __atomic_sub_fetch(&g_poison, 1, __ATOMIC_RELAXED);
    *want = v ^ 0x80008000u;
  }
}
extern "C" void t2v_debug_poison_exchange(int n) { __atomic_store_n(&g_poison, n, __ATOMIC_RELAXED); }

unsigned* t2v_coop_fault_word() { return coop_fault_word(); }

bool t2v_coop_allowed() {
  if (g_coop.disabled || coop_fault_word() == nullptr) return false;
  if (__atomic_load_n(g_coop.fault, __ATOMIC_RELAXED) != 0u) { g_coop.disabled = true; return false; }
  return true;
}

unsigned* t2v_peer_fault_word() {
  unsigned* f = coop_fault_word();
  return f ? f + 1 : nullptr;
}

int t2v_async_fault_pending() {
  if (g_coop.fault == nullptr) return 0;
  if (__atomic_load_n(g_coop.fault + 1, __ATOMIC_RELAXED) != 0u && !g_coop.peer_reported) return 1;
  return __atomic_load_n(g_coop.fault, __ATOMIC_RELAXED) != 0u && !g_coop.reported;
}

int t2v_async_fault_consume(std::string* msg) {
  if (!t2v_async_fault_pending()) return 0;
  if (__atomic_load_n(g_coop.fault + 1, __ATOMIC_RELAXED) != 0u && !g_coop.peer_reported) {
    g_coop.peer_reported = true;
    if (msg) *msg = "peer exchange: a workgroup gave up waiting for a peer rank's message in its window (T2V_PEER_TIMEOUT_MS; the peer died, "
                    "ran a different program, or its stores are not visible here); the results of the run that was in flight are invalid.  "
                    "The exchanges of this process go through RCCL from now on (T2V_PEER_WINDOW=0 selects that from the start)";
    return 1;
  }
  g_coop.reported = true;
  g_coop.disabled = true;
  if (msg) *msg = "fused normalisation: a workgroup gave up waiting for the statistics of the other workgroups of its launch (the device is "
                  "shared with another client that holds compute units); the results of the run that was in flight are invalid.  The "
                  "three-launch GroupNorm is used from now on; programs with norms fused into GEMM epilogues must be lowered again with "
                  "T2V_GN_COOP=0 (which selects the unfused forms from the start)";
  return 1;
}

// Scratch (op.p[4], owned by the caller): fp64 partials [nparts][n_inst][nblk][groups][2] followed by
// fp32 finals [n_inst][groups][2], nblk = ceil(rows / rpb), rpb = op.i[11] (0: T2V_GN_ROWS_PER_BLOCK).
// op.i[8] = phase: 0 = whole op; 1 = statistics only (writes this rank's partials into part op.i[10]);
//                  2 = fold the op.i[9] gathered parts + normalise (after the partials all-gather).
hipError_t t2v_launch_groupnorm(const t2v_op& op, hipStream_t s) {
  const int n_inst = op.i[0], rows = op.i[1], C = op.i[2], ld_in = op.i[3], groups = op.i[4];
  const int in_dt = op.i[5], silu = op.i[6], ld_out = op.i[7];
  const int phase = op.i[8], nparts = op.i[9] > 0 ? op.i[9] : 1, part = op.i[10];
  const int rpb = op.i[11] > 0 ? op.i[11] : T2V_GN_ROWS_PER_BLOCK;
  if (C % 8 != 0 || ld_in % 8 != 0 || ld_out % 8 != 0 || C % groups != 0 || groups > 256 || n_inst <= 0 || rows <= 0 ||
      op.p[4] == 0 || part < 0 || part >= nparts || phase < 0 || phase > 3)
    return hipErrorInvalidValue;
  if (phase == 3 && (op.p[6] == 0 || rows % 32 != 0 || op.i[17] < C || nparts != 1)) return hipErrorInvalidValue;
  // phase 1 with producer strips (p[6] != 0, round 6): this rank's part of a T-sharded cross-frame norm comes from the T2V_EPI_STATS strips of
  // the GEMM that produced x — ONE small fold launch instead of a statistics pass over the tensor + a fold
  const bool strips1 = phase == 1 && op.p[6] != 0;
  if (strips1 && (rows % 32 != 0 || op.i[17] < C)) return hipErrorInvalidValue;
  if ((op.i[21] != 0 || op.i[22] != 0) && (phase != 2 || n_inst != 1 || op.i[21] < 0 || op.i[22] < 0 || op.p[8] != 0)) return hipErrorInvalidValue;
  // T-sharded clips (phase 1 / 2): every rank folds ITS OWN block partials to one {sum, sum of squares} pair per
  // (instance, group) — 512 bytes per instance — and only those are all-gathered; phase 2 sums the parts in rank order
  // (every rank: same values, same order => bit-identical statistics) over the rows of the whole clip (i[14]).
  const long rows_total = op.i[14] > 0 ? (long)op.i[14] : (long)rows * nparts;
  const int nblk = (rows + rpb - 1) / rpb;
  const bool whole = phase == 0 || phase == 3;         // scratch layout of an unsharded op: [block partials][finals]
  const size_t part_len = whole ? (size_t)n_inst * nblk * groups * 2 : (size_t)n_inst * groups * 2;        // doubles per gathered part
  double* partials = reinterpret_cast<double*>(op.p[4]);
  double* local = whole ? partials : partials + part_len * nparts;                                         // block partials of this rank
  float* finals = reinterpret_cast<float*>(local + (size_t)n_inst * nblk * groups * 2);
  const dim3 g1(nblk, n_inst);
  const int cv = C / 8;
  const int R = cv < 256 ? 256 / cv : 1;
  const size_t lds = 2 * (size_t)R * C * sizeof(float);
  const int g2 = (n_inst * groups + 3) / 4;
  const double inv_n = 1.0 / ((double)rows_total * (C / groups));
  const float* gamma = reinterpret_cast<const float*>(op.p[1]);
  const float* beta = reinterpret_cast<const float*>(op.p[2]);
  f16* out = reinterpret_cast<f16*>(op.p[3]);
  const bool fused = op.i[12] != 0;
  const int lo_off = op.i[16] != 0 ? C : 0;        // hi + lo operand split: low-order images at columns C .. 2C-1 of the output rows
  // second output: the raw input as fp16 (+ low-order image), p[8] [n_inst * rows, i[19]], i[20] = column of the low-order image (0: none)
  const GnCast cast = {reinterpret_cast<f16*>(op.p[8]), op.i[19], op.i[20]};
  if (cast.out != nullptr && (phase == 1 || cast.ld < C + (cast.lo ? C : 0) || cast.ld % 8 != 0 || cast.lo % 8 != 0)) return hipErrorInvalidValue;
  if (lo_off && (phase == 1 || ld_out < 2 * C)) return hipErrorInvalidValue;
  if (fused && (phase != 0 || (C / groups) % 4 != 0 || (C / groups) / 4 > GNF_THREADS)) return hipErrorInvalidValue;
  // single-pass cooperative variant: the smallest rows-per-thread count whose grid still fits one workgroup per CU
  int coop_kr = 0, coop_rc = 0, coop_nchunk = 0;
  if (op.i[15] != 0 && op.p[5] != 0 && phase == 0 && !fused && groups <= 32 && cv <= GNC_THREADS) {
    const int Rc = GNC_THREADS / cv, ncu = t2v_num_cus(s);
    for (int kr : {4, 8, 12, 16, 20}) {
      const int rc = Rc * kr, nchunk = (rows + rc - 1) / rc;
      if ((long)n_inst * nchunk <= ncu && nchunk <= nblk) { coop_kr = kr; coop_rc = rc; coop_nchunk = nchunk; break; }
    }
  }
  auto run = [&](auto* x) {
    using T = typename std::remove_cv<typename std::remove_pointer<decltype(x)>::type>::type;
    if (coop_kr) {
      const size_t ldsc = (size_t)2 * (GNC_THREADS / cv) * C * sizeof(float);
      unsigned* bar = reinterpret_cast<unsigned*>(op.p[5]);
      const dim3 grid(n_inst * coop_nchunk);
      double* records = ((size_t)n_inst * coop_nchunk * groups * 16 <= (size_t)op.i[18]) ? reinterpret_cast<double*>(op.p[7]) : nullptr;
      const bool done = silu ? gn_coop_launch<T, true>(coop_kr, grid, ldsc, s, x, gamma, beta, out, partials, bar, rows, C, ld_in, ld_out, groups,
                                                       coop_nchunk, coop_rc, inv_n, op.f[0], lo_off, records, cast)
                             : gn_coop_launch<T, false>(coop_kr, grid, ldsc, s, x, gamma, beta, out, partials, bar, rows, C, ld_in, ld_out, groups,
                                                        coop_nchunk, coop_rc, inv_n, op.f[0], lo_off, records, cast);
      if (done) return;             // else: not provably co-resident (or a fault was raised earlier) -> the three launches below
    }
    if (fused) {
      if (silu) hipLaunchKernelGGL((gn_fused_kernel<T, true>), dim3(groups * n_inst), dim3(GNF_THREADS), 0, s, x, gamma, beta, out, rows,
                                   C, ld_in, ld_out, groups, op.f[0], lo_off, cast);
      else hipLaunchKernelGGL((gn_fused_kernel<T, false>), dim3(groups * n_inst), dim3(GNF_THREADS), 0, s, x, gamma, beta, out, rows, C,
                              ld_in, ld_out, groups, op.f[0], lo_off, cast);
      return;
    }
    double* raw_part = strips1 ? partials + part_len * part : nullptr;
    if ((phase == 3 || strips1) && (long)(rows / 32) * (C / groups) > 1024)
      hipLaunchKernelGGL(gn_finalize_strips_wg_kernel, dim3(n_inst * groups), dim3(256), 0, s, reinterpret_cast<const float*>(op.p[6]), finals,
                         rows / 32, groups, C / groups, op.i[17], inv_n, op.f[0], raw_part);
    else if (phase == 3 || strips1)
      hipLaunchKernelGGL(gn_finalize_strips_kernel, dim3(g2), dim3(256), 0, s, reinterpret_cast<const float*>(op.p[6]), finals, n_inst, rows / 32,
                         groups, C / groups, op.i[17], inv_n, op.f[0], raw_part);
    else if (phase != 2)
      hipLaunchKernelGGL(gn_stats_kernel<T>, g1, dim3(256), lds, s, x, local, rows, C, ld_in, groups, rpb);
    if (phase == 1 && !strips1)
      hipLaunchKernelGGL(gn_finalize_kernel, dim3(g2), dim3(256), 0, s, local, finals, n_inst, nblk, groups, 0.0, 0.f, 1,
                         partials + part_len * part);
    if (phase != 1) {
      // phase 2 (round 6): the gathered parts are folded inside the apply pass (no finalize launch)
      const double* parts2 = phase == 2 ? partials : nullptr;
      if (phase == 0)
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(g2), dim3(256), 0, s, partials, finals, n_inst, nblk, groups,
                           inv_n, op.f[0], nparts, static_cast<double*>(nullptr));
      const int rpa = R * GN_UNROLL;    // rows per normalise workgroup
      // phase 2 of a T-sharded clip (T2V_OP_STATS_HALO): i[21] / i[22] rows in front of / behind x hold the neighbours' RAW boundary
      // frames; they take the same statistics and land in front of / behind `out` (what the neighbour wrote for itself, bit for bit)
      const int before = phase == 2 ? op.i[21] : 0, after = phase == 2 ? op.i[22] : 0;
      const int arows = rows + before + after;
      const T* ax = x - (size_t)before * ld_in;
      f16* aout = out - (size_t)before * ld_out;
      const dim3 g3((arows + rpa - 1) / rpa, n_inst);
      const size_t lds3 = (2 * (size_t)C + (parts2 ? 2 * (size_t)groups : 0)) * sizeof(float);
      if (silu) hipLaunchKernelGGL((gn_apply_kernel<T, true>), g3, dim3(256), lds3, s, ax, finals, gamma, beta, aout, arows, C, ld_in, ld_out, groups, lo_off, cast,
                                   parts2, nparts, (long)part_len, inv_n, op.f[0]);
      else hipLaunchKernelGGL((gn_apply_kernel<T, false>), g3, dim3(256), lds3, s, ax, finals, gamma, beta, aout, arows, C, ld_in, ld_out, groups, lo_off, cast,
                              parts2, nparts, (long)part_len, inv_n, op.f[0]);
    }
  };
  if (in_dt == T2V_F32) run(reinterpret_cast<const float*>(op.p[0]));
  else run(reinterpret_cast<const f16*>(op.p[0]));
  return hipGetLastError();
}

hipError_t t2v_launch_layernorm(const t2v_op& op, hipStream_t s) {
  const int M = op.i[0], C = op.i[1], ld_in = op.i[2], ld_out = op.i[3];
  if (C % 4 != 0 || C > 64 * 4 * 8) return hipErrorInvalidValue;
  const float* x = reinterpret_cast<const float*>(op.p[0]);
  const float* gamma = reinterpret_cast<const float*>(op.p[1]);
  const float* beta = reinterpret_cast<const float*>(op.p[2]);
  f16* out = reinterpret_cast<f16*>(op.p[3]);
  // C <= 768 (the 32x32 / 16x16 levels: many rows): at most 8 workgroups per CU, longer tensors are walked grid-stride
  // (~6 rows per wave at the 32x32 level); wider rows: one row per wave
  const int wgs = (M + 3) / 4;
  const int cap = op.i[4] > 0 ? op.i[4] : 8 * 256;
  const dim3 grid_rows(wgs < cap ? wgs : cap), grid(wgs);
  if (C <= 64 * 4 * 2)
    hipLaunchKernelGGL(layernorm_rows_kernel<2>, grid_rows, dim3(256), 0, s, x, gamma, beta, out, M, C, ld_in, ld_out, op.f[0]);
  else if (C <= 64 * 4 * 3)
    hipLaunchKernelGGL(layernorm_rows_kernel<3>, grid_rows, dim3(256), 0, s, x, gamma, beta, out, M, C, ld_in, ld_out, op.f[0]);
  else if (C <= 64 * 4 * 5)
    hipLaunchKernelGGL(layernorm_kernel<5>, grid, dim3(256), 0, s, x, gamma, beta, out, M, C, ld_in, ld_out, op.f[0]);
  else
    hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, s, x, gamma, beta, out, M, C, ld_in, ld_out, op.f[0]);
  return hipGetLastError();
}
