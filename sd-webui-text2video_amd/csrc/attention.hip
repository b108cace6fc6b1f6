// Fused softmax(Q K^T * scale) V for head_dim 40 / 64 / 80 / 160 on gfx950 MFMA (wave64), and the
// relative-position temporal attention of the VideoCrafter LVDM UNet.
//
// Replaces CrossAttention.forward's SDPA / einsum-softmax-einsum dispatch
// (reference scripts/modelscope/t2v_model.py:540-584) for the three call shapes of the UNet
// (SURVEY.md §2.3 K6-K8):
//   spatial self-attention   n_q = n_k = h*w,  batch = (b f) x heads
//   text cross-attention     n_q = h*w, n_k = 77 (K/V broadcast over frames via stride 0)
//   temporal self-attention  n_q = n_k = F,    batch = (b h w) x heads, sequence stride = h*w*ld
// Q/K/V/O are addressed by (sequence, batch_outer, batch_inner) element strides + 64*head, so
// the fused QKV GEMM output [tokens, 3C] is consumed in place — none of the reference's
// 'b n (h d) -> (b h) n d' / '(b f) c h w -> (b h w) f c' rearrange copies exist.
//
// Structure: one wave owns 32 query rows; a workgroup (NW waves) shares 64-key K/V tiles in
// LDS.  S^T = K Q^T is computed with the *key* index on the MFMA row axis, so every lane
// holds 2x16 scores of ONE query: the online-softmax max/sum are in-lane reductions plus one
// lane^32 exchange.  The P fragment for P·V is taken from the accumulator registers directly
// (the key order inside a 16-slot MFMA step is a free permutation as long as V^T uses the
// same one), so P never goes through LDS.  V is transposed while being written to LDS
// ([d][key], 136-B rows: conflict-free 8-byte fragment reads).
// head_dim: ModelScope uses 64 everywhere; the LVDM UNet has 8 heads of C/8 = 40 / 80 / 160 channels
// (openaimodel3d.py:459-466).  The reduction of Q K^T is zero-padded to a multiple of 16 and the rows of
// O^T to a multiple of 32 inside LDS / registers only; nothing padded is read from or written to HBM.
#include <cstdlib>

#include "t2v_kernels.h"

namespace {

struct AttnParams {
  const f16* q; const f16* k; const f16* v; f16* o;
  int nq, nk, heads, b_outer, b_inner;
  long sq_seq, sq_out, sq_in;
  long sk_seq, sk_out, sk_in;
  long so_seq, so_out, so_in;
  float scale_log2;
  int causal;   // 1: key s is visible to query t only if s <= t (CLIP text towers)
  int lo_off;   // != 0: also store fp16(o - float(fp16(o))) at o + lo_off (elements): rows [hi | lo] for a K-doubled to_out (precise_operands)
};

// KT = keys per LDS tile: 64, or 32 for sequences of <= 32 keys (temporal attention over the frames of one pixel:
// half the LDS per workgroup, twice the workgroups per CU for a kernel that is bound by memory latency).
constexpr int KT_MAIN = 64;
template <int NW, int D, int KT = 64, bool PREFETCH = false>
__global__ __launch_bounds__(NW * 64) void attn_kernel(const AttnParams p) {
  constexpr int NT = NW * 64;
  constexpr int NKT = KT / 32;             // 32-key MFMA tiles per LDS tile
  constexpr int VT_ROW = KT * 2 + 8;       // bytes per V^T row (KT keys * 2 B + 8 B pad)
  constexpr int DK = (D + 15) / 16 * 16;   // Q K^T reduction length (zero-padded in LDS / registers)
  constexpr int NKK = DK / 16;
  constexpr int DV = (D + 31) / 32 * 32;   // rows of O^T (zero-padded)
  constexpr int NDT = DV / 32;
  constexpr int KCH = DK / 8;              // 16-byte chunks per K row
  constexpr int K_ROW = DK * 2 + 16;       // padded LDS row: 32 consecutive rows at one chunk hit disjoint banks
  static_assert(D % 8 == 0, "head_dim must be a multiple of 8");
  __shared__ __attribute__((aligned(16))) unsigned char k_lds[KT * K_ROW];    // [key][DK]
  __shared__ __attribute__((aligned(16))) unsigned char vt_lds[DV * VT_ROW];  // [d][KT keys]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int head = blockIdx.y;
  const int bo = blockIdx.z / p.b_inner, bi = blockIdx.z % p.b_inner;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const int frow = lane & 31, fhalf = lane >> 5;

  const f16* qb = p.q + bo * p.sq_out + bi * p.sq_in + head * D;
  const f16* kb = p.k + bo * p.sk_out + bi * p.sk_in + head * D;
  const f16* vb = p.v + bo * p.sk_out + bi * p.sk_in + head * D;
  f16* ob = p.o + bo * p.so_out + bi * p.so_in + head * D;

  // Q fragments: query (lane&31), d = kk*16 + 8*fhalf + 0..7
  const int qrow = q0 + frow;
  f16x8 qf[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    const int d0 = kk * 16 + fhalf * 8;
    if (qrow < p.nq && d0 < D)
      qf[kk] = *reinterpret_cast<const f16x8*>(qb + (long)qrow * p.sq_seq + d0);
    else
      for (int e = 0; e < 8; ++e) qf[kk][e] = (f16)0.f;
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // last visible key of this lane's query; key 0 is visible to every query, so m_run is finite after the first tile
  const int key_end = p.causal ? min(p.nk, qrow + 1) : p.nk;

  // PREFETCH (experiment, env T2V_ATTN_PREFETCH=1, 4-wave variants with head_dim <= 80): the K / V tile of the next
  // iteration is fetched into registers while the current one is processed.  Measured equal-to-slower (66.5 vs 66.0 ms
  // on the 9216-token level, 24 more VGPRs): the kernel is bound by the softmax VALU work, not by load latency.
  constexpr bool PF = PREFETCH;
  constexpr int KI = (KT * KCH + NT - 1) / NT;                 // 16-byte K chunks per thread
  constexpr int VI = ((KT / 2) * (DV / 4) + NT - 1) / NT;      // (key pair, 4 d) V units per thread
  f16x8 kreg[KI];
  f16x4 vra[VI], vrb[VI];
  auto gload = [&](int kt0) {
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int u = tid + i * NT;
      const int row = u / KCH, c = u - row * KCH;
      const int key = kt0 + row;
      if (u < KT * KCH && key < p.nk && c * 8 < D)
        kreg[i] = *reinterpret_cast<const f16x8*>(kb + (long)key * p.sk_seq + c * 8);
      else
        for (int e = 0; e < 8; ++e) kreg[i][e] = (f16)0.f;
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int u = tid + i * NT;
      const int kp = u / (DV / 4), dq = u - kp * (DV / 4);
      const int key = kt0 + 2 * kp;
      const bool dok = u < (KT / 2) * (DV / 4) && dq * 4 < D;
      if (dok && key < p.nk) vra[i] = *reinterpret_cast<const f16x4*>(vb + (long)key * p.sk_seq + dq * 4);
      else for (int e = 0; e < 4; ++e) vra[i][e] = (f16)0.f;
      if (dok && key + 1 < p.nk) vrb[i] = *reinterpret_cast<const f16x4*>(vb + (long)(key + 1) * p.sk_seq + dq * 4);
      else for (int e = 0; e < 4; ++e) vrb[i][e] = (f16)0.f;
    }
  };
  auto lstore = [&]() {
    // K tile: KT rows x KCH chunks of 16 B;  V tile, transposed: unit = (key pair, 4 d) -> 4 x 32-bit {V[2kp][d], V[2kp+1][d]}
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int u = tid + i * NT;
      const int row = u / KCH, c = u - row * KCH;
      if (u < KT * KCH) *reinterpret_cast<f16x8*>(k_lds + row * K_ROW + (c << 4)) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int u = tid + i * NT;
      const int kp = u / (DV / 4), dq = u - kp * (DV / 4);
      if (u < (KT / 2) * (DV / 4)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          typedef f16 f16x2 __attribute__((ext_vector_type(2)));
          f16x2 w = {vra[i][e], vrb[i][e]};
          *reinterpret_cast<f16x2*>(vt_lds + (dq * 4 + e) * VT_ROW + kp * 4) = w;
        }
      }
    }
  };
  if (PF) gload(0);

  for (int kt0 = 0; kt0 < p.nk; kt0 += KT) {
    __syncthreads();  // previous tile fully consumed
#ifdef T2V_ATTN_NOSTAGE   // timing experiment only (wrong results): the K / V tile staged once
    if (kt0 == 0)
#endif
    if constexpr (PF) {
      lstore();
    } else {
      // straight global -> LDS (few registers: the single-wave variants live on occupancy)
      for (int u = tid; u < KT * KCH; u += NT) {
        const int row = u / KCH, c = u - row * KCH;
        const int key = kt0 + row;
        f16x8 val;
        if (key < p.nk && c * 8 < D)
          val = *reinterpret_cast<const f16x8*>(kb + (long)key * p.sk_seq + c * 8);
        else
          for (int e = 0; e < 8; ++e) val[e] = (f16)0.f;
        *reinterpret_cast<f16x8*>(k_lds + row * K_ROW + (c << 4)) = val;
      }
      for (int u = tid; u < (KT / 2) * (DV / 4); u += NT) {
        const int kp = u / (DV / 4), dq = u - kp * (DV / 4);
        const int key = kt0 + 2 * kp;
        const bool dok = dq * 4 < D;
        f16x4 a, b;
        if (dok && key < p.nk) a = *reinterpret_cast<const f16x4*>(vb + (long)key * p.sk_seq + dq * 4);
        else for (int e = 0; e < 4; ++e) a[e] = (f16)0.f;
        if (dok && key + 1 < p.nk) b = *reinterpret_cast<const f16x4*>(vb + (long)(key + 1) * p.sk_seq + dq * 4);
        else for (int e = 0; e < 4; ++e) b[e] = (f16)0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          typedef f16 f16x2 __attribute__((ext_vector_type(2)));
          f16x2 w = {a[e], b[e]};
          *reinterpret_cast<f16x2*>(vt_lds + (dq * 4 + e) * VT_ROW + kp * 4) = w;
        }
      }
    }
    __syncthreads();
    if (PF && kt0 + KT < p.nk) gload(kt0 + KT);

    // ---- S^T = K Q^T : NKT 32-key tiles --------------------------------------------
    const bool t1_live = (kt0 + 32) < p.nk;   // wave-uniform
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NKT];
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      s[T] = zero16;
      if (T == 1 && !t1_live) continue;
      const int row = T * 32 + frow;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(k_lds + row * K_ROW + ((kk * 2 + fhalf) << 4));
        s[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], kk == 0 ? zero16 : s[T], 0, 0, 0);
      }
    }
    // ---- online softmax for this lane's query -------------------------------------
    // The softmax is the VALU-bound part of this kernel (32 scores per lane and tile against 16 MFMAs), so it is kept
    // to max / fma / exp2 / add per score: keys are masked only in tiles that need it (ragged end, causal diagonal),
    // the scale is folded into the exponent's fma, exp2 is the bare v_exp_f32 (arguments <= 8, flushed denormals are
    // zeros of the sum anyway), and the running maximum is only advanced when it grows by more than 2^8 — O and l
    // carry the same stale factor, so the result is exact and the accumulator rescale (AGPR round trips) is rare.
    if (kt0 + KT > p.nk || (p.causal && kt0 + KT - 1 > q0)) {     // wave-uniform
#pragma unroll
      for (int T = 0; T < NKT; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt0 + T * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          if (key >= key_end) s[T][r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int T = 0; T < NKT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[T][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_tile = mx * p.scale_log2;                 // scale > 0 (checked at launch)
    const bool grow = m_tile - m_run > 8.0f;                // first tile: m_run = -inf (key kt0 = 0 is always visible)
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {           // wave-uniform
      const float alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_tile) : 1.0f;   // exp2(-inf) = 0 on the first tile
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      m_run = grow ? m_tile : m_run;
    }
    float psum = 0.f;
    const float neg_m = -m_run;
#pragma unroll
    for (int T = 0; T < NKT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef T2V_ATTN_NOEXP     // timing experiment only (wrong results): the softmax without its transcendental
        const float pv = __builtin_fmaf(s[T][r], p.scale_log2, neg_m);
#else
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[T][r], p.scale_log2, neg_m));
#endif
        s[T][r] = pv;
        psum += pv;
      }
    psum += __shfl_xor(psum, 32);
    l_run += psum;

    // ---- O^T += V^T P^T -----------------------------------------------------------
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      if (T == 1 && !t1_live) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (f16)s[T][8 * t + e];
        const int kofs = (T * 32 + t * 16 + 4 * fhalf) * 2;  // byte offset of key slot e=0
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const unsigned char* vrow = vt_lds + (d * 32 + frow) * VT_ROW + kofs;
          const f16x4 lo = *reinterpret_cast<const f16x4*>(vrow);
          const f16x4 hi = *reinterpret_cast<const f16x4*>(vrow + 16);
          const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[d], 0, 0, 0);
        }
      }
    }
  }

  if (qrow < p.nq) {
    const float inv = 1.0f / l_run;
    f16* orow = ob + (long)qrow * p.so_seq;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int col = d * 32 + 8 * qd + 4 * fhalf;
        if (col < D) {
          f16x4 o, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float val = oacc[d][4 * qd + r] * inv;
            o[r] = (f16)val;
            lo[r] = (f16)(val - (float)o[r]);
          }
          *reinterpret_cast<f16x4*>(orow + col) = o;
          if (p.lo_off) *reinterpret_cast<f16x4*>(orow + p.lo_off + col) = lo;
        }
      }
  }
}

// ---- spatial self-attention at head_dim 64 with the K / V^T tiles staged by LDS-DMA (round 6) ---------------------------------------
// Ablation of attn_kernel on the 9216-token level (profiles/r06_attention.txt): the exp2 costs 3 %, the K / V staging 32 % — every
// workgroup (128 queries) re-stages every 64-key tile through registers, synchronously, and TRANSPOSES V with 4-byte scattered LDS writes;
// at 9216 tokens 72 workgroups per (frame, head) repeat that transposition.  Here
//  * V is transposed ONCE per launch in HBM by vt_transpose_kernel: vt[(batch, head)][d][key] with n_pad (multiple of 64) keys per row,
//    zero beyond nk, the keys of every 16 stored in the order the MFMA's K-slots take them from the score registers (see attn_kernel:
//    lane-half f, element e <-> key 4 f + (e & 3) + 8 (e >> 2)), so that one 16-byte LDS read is a whole A fragment of V^T;
//  * a 64-key tile is then two [64 rows][128 B] images — K rows = keys, V^T rows = d — staged by 16 `global_load_lds` pieces (8 rows x
//    128 B each, XOR chunk swizzle on the per-lane source address and on the fragment read, exactly gemm2.hip's operand tiles), no
//    registers, no ds_write, double-buffered: the pieces of tile t + 1 are issued after the one barrier of tile t and land under its
//    MFMAs / softmax;
//  * NW = 8 waves (256 queries) share a tile.
// Scores, online softmax, P fragments and the epilogue are attn_kernel's, unchanged (same values, same order: bit-identical results).
__device__ __attribute__((aligned(256))) unsigned char attn_zero_page[256];

__global__ __launch_bounds__(256) void vt_transpose_kernel(const AttnParams p, f16* vt, int n_pad) {
  __shared__ f16 tile[64][66];                                // [key][d], 132-byte rows: the 2-byte column reads below spread over the banks
  const int tid = threadIdx.x;
  const int head = blockIdx.y, bo = blockIdx.z / p.b_inner, bi = blockIdx.z % p.b_inner;
  const int k0 = blockIdx.x * 64;
  const f16* vb = p.v + bo * p.sk_out + bi * p.sk_in + head * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * 256, key = u >> 3, c = u & 7;
    f16x8 val;
    if (k0 + key < p.nk) val = *reinterpret_cast<const f16x8*>(vb + (long)(k0 + key) * p.sk_seq + c * 8);
    else for (int e = 0; e < 8; ++e) val[e] = (f16)0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[key][c * 8 + e] = val[e];
  }
  __syncthreads();
  f16* out = vt + ((long)blockIdx.z * p.heads + head) * 64 * n_pad + k0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * 256, d = u >> 3, c = u & 7;      // 8 key positions c * 8 .. c * 8 + 7 of row d
    f16x8 val;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = c >> 1, f = c & 1;                         // 16-key group, lane-half slot
      val[e] = tile[16 * g + 4 * f + (e & 3) + 8 * (e >> 2)][d];
    }
    *reinterpret_cast<f16x8*>(out + (long)d * n_pad + c * 8) = val;
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void attn2_kernel(const AttnParams p, const f16* __restrict__ vt, int n_pad) {
  constexpr int D = 64, KT = 64, NKK = 4, NDT = 2, NKT = 2;
  constexpr int PIECES = 16, PPW = PIECES / NW;               // 1-KiB DMA pieces per tile (8 K + 8 V^T), per wave
  static_assert(PIECES % NW == 0, "pieces must divide over the waves");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];      // [buffer][K image 8 KiB | V^T image 8 KiB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int head = blockIdx.y;
  const int bo = blockIdx.z / p.b_inner, bi = blockIdx.z % p.b_inner;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const int frow = lane & 31, fhalf = lane >> 5;

  const f16* qb = p.q + bo * p.sq_out + bi * p.sq_in + head * D;
  const f16* kb = p.k + bo * p.sk_out + bi * p.sk_in + head * D;
  const f16* vtb = vt + ((long)blockIdx.z * p.heads + head) * 64 * n_pad;
  f16* ob = p.o + bo * p.so_out + bi * p.so_in + head * D;

  const int qrow = q0 + frow;
  f16x8 qf[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    if (qrow < p.nq) qf[kk] = *reinterpret_cast<const f16x8*>(qb + (long)qrow * p.sq_seq + kk * 16 + fhalf * 8);
    else for (int e = 0; e < 8; ++e) qf[kk][e] = (f16)0.f;
  }
  f32x16 oacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // per-lane DMA sources of this wave's pieces: piece j < 8 = keys 8 j .. 8 j + 7 of the tile, j >= 8 = rows d = 8 (j - 8) .. of V^T
  const int lrow = lane >> 3, pch = lane & 7;
  const f16* src0[PPW];        // source of tile 0 (K: advanced by 64 keys per tile; V^T: by 64 columns)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int j = wave + NW * i, row = 8 * (j & 7) + lrow, lc = pch ^ ((row >> 1) & 7);
    src0[i] = j < 8 ? kb + (long)row * p.sk_seq + lc * 8 : vtb + (long)row * n_pad + lc * 8;
  }
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int j = wave + NW * i;                            // wave-uniform
      const f16* src = src0[i] + (long)t * (j < 8 ? (long)KT * p.sk_seq : (long)KT);
      if (j < 8 && t * KT + 8 * (j & 7) + lrow >= p.nk) src = reinterpret_cast<const f16*>(attn_zero_page);   // K rows past the sequence
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + buf * 16384 + (j >> 3) * 8192 + (j & 7) * 1024), 16, 0, 0);
    }
  };
  const int nt = (p.nk + KT - 1) / KT;
  const int sw = ((frow >> 1) & 7);                           // swizzle term of every fragment row (sub-tile rows are multiples of 32)
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int kt0 = t * KT;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of tile t have landed (and, at t = 0, its Q fragments)
    __builtin_amdgcn_s_barrier();                             // ... everyone's have; everyone is done reading tile t - 1
    asm volatile("" ::: "memory");
    if (t + 1 < nt) stage(t + 1, (t + 1) & 1);                // wave-uniform
    const unsigned char* kl = smem + (t & 1) * 16384;
    const unsigned char* vl = kl + 8192;
    const bool t1_live = (kt0 + 32) < p.nk;                   // wave-uniform
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NKT];
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      s[T] = zero16;
      if (T == 1 && !t1_live) continue;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(kl + (T * 32 + frow) * 128 + (((kk * 2 + fhalf) ^ sw) << 4));
        s[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], kk == 0 ? zero16 : s[T], 0, 0, 0);
      }
    }
    if (kt0 + KT > p.nk) {                                    // ragged last tile (wave-uniform)
#pragma unroll
      for (int T = 0; T < NKT; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt0 + T * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf >= p.nk) s[T][r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int T = 0; T < NKT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[T][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_tile = mx * p.scale_log2;
    const bool grow = m_tile - m_run > 8.0f;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_tile) : 1.0f;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      m_run = grow ? m_tile : m_run;
    }
    float psum = 0.f;
    const float neg_m = -m_run;
#pragma unroll
    for (int T = 0; T < NKT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[T][r], p.scale_log2, neg_m));
        s[T][r] = pv;
        psum += pv;
      }
    psum += __shfl_xor(psum, 32);
    l_run += psum;
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      if (T == 1 && !t1_live) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (f16)s[T][8 * u + e];
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const f16x8 vf = *reinterpret_cast<const f16x8*>(vl + (d * 32 + frow) * 128 + (((4 * T + 2 * u + fhalf) ^ sw) << 4));
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[d], 0, 0, 0);
        }
      }
    }
  }
  if (qrow < p.nq) {
    const float inv = 1.0f / l_run;
    f16* orow = ob + (long)qrow * p.so_seq;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int col = d * 32 + 8 * qd + 4 * fhalf;
        f16x4 o, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float val = oacc[d][4 * qd + r] * inv;
          o[r] = (f16)val;
          lo[r] = (f16)(val - (float)o[r]);
        }
        *reinterpret_cast<f16x4*>(orow + col) = o;
        if (p.lo_off) *reinterpret_cast<f16x4*>(orow + p.lo_off + col) = lo;
      }
  }
}

// The same kernel with the scores of tile t + 1 issued BEFORE the softmax of tile t (experiment, round 6): the 8 Q K^T MFMAs run in the matrix
// pipe while this wave's own VALU works on the previous tile's scores.  K tiles in a 3-deep ring (K(t + 1) is read one tile early), V^T in 2.
template <int NW, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void attn2p_kernel(const AttnParams p, const f16* __restrict__ vt, int n_pad) {
  constexpr int D = 64, KT = 64, NKK = 4, NDT = 2, NKT = 2;
  constexpr int PIECES = 16, PPW = PIECES / NW;               // 1-KiB DMA pieces per tile (8 K + 8 V^T), per wave
  static_assert(PIECES % NW == 0, "pieces must divide over the waves");
  __shared__ __attribute__((aligned(16))) unsigned char smem[5 * 8192];       // K ring [3][8 KiB] | V^T ring [2][8 KiB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int head = blockIdx.y;
  const int bo = blockIdx.z / p.b_inner, bi = blockIdx.z % p.b_inner;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const int frow = lane & 31, fhalf = lane >> 5;

  const f16* qb = p.q + bo * p.sq_out + bi * p.sq_in + head * D;
  const f16* kb = p.k + bo * p.sk_out + bi * p.sk_in + head * D;
  const f16* vtb = vt + ((long)blockIdx.z * p.heads + head) * 64 * n_pad;
  f16* ob = p.o + bo * p.so_out + bi * p.so_in + head * D;

  const int qrow = q0 + frow;
  f16x8 qf[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    if (qrow < p.nq) qf[kk] = *reinterpret_cast<const f16x8*>(qb + (long)qrow * p.sq_seq + kk * 16 + fhalf * 8);
    else for (int e = 0; e < 8; ++e) qf[kk][e] = (f16)0.f;
  }
  f32x16 oacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // per-lane DMA sources of this wave's pieces: piece j < 8 = keys 8 j .. 8 j + 7 of the tile, j >= 8 = rows d = 8 (j - 8) .. of V^T
  const int lrow = lane >> 3, pch = lane & 7;
  const f16* src0[PPW];        // source of tile 0 (K: advanced by 64 keys per tile; V^T: by 64 columns)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int j = wave + NW * i, row = 8 * (j & 7) + lrow, lc = pch ^ ((row >> 1) & 7);
    src0[i] = j < 8 ? kb + (long)row * p.sk_seq + lc * 8 : vtb + (long)row * n_pad + lc * 8;
  }
  auto stage = [&](int t, bool want_k, bool want_v) {          // K(t) -> K ring slot t % 3, V^T(t) -> V ring slot t & 1
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int j = wave + NW * i;                            // wave-uniform
      if (j < 8 ? !want_k : !want_v) continue;
      const f16* src = src0[i] + (long)t * (j < 8 ? (long)KT * p.sk_seq : (long)KT);
      if (j < 8 && t * KT + 8 * (j & 7) + lrow >= p.nk) src = reinterpret_cast<const f16*>(attn_zero_page);
      unsigned char* dst = j < 8 ? smem + (t % 3) * 8192 + (j & 7) * 1024 : smem + 3 * 8192 + (t & 1) * 8192 + (j & 7) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  const int nt = (p.nk + KT - 1) / KT;
  const int sw = ((frow >> 1) & 7);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto scores = [&](int t, f32x16 (&s)[NKT]) {                // S^T of tile t from its K ring slot
    const unsigned char* kl = smem + (t % 3) * 8192;
    const bool t1_live = (t * KT + 32) < p.nk;
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      s[T] = zero16;
      if (T == 1 && !t1_live) continue;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(kl + (T * 32 + frow) * 128 + (((kk * 2 + fhalf) ^ sw) << 4));
        s[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], kk == 0 ? zero16 : s[T], 0, 0, 0);
      }
    }
  };
  stage(0, true, true);
  if (nt > 1) stage(1, true, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  f32x16 s[NKT], sn[NKT];
  scores(0, s);
  for (int t = 0; t < nt; ++t) {
    const int kt0 = t * KT;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // K(t + 1), V^T(t): this wave's pieces have landed
    __builtin_amdgcn_s_barrier();                             // ... everyone's; everyone is done with K(t) [scores issued last iteration], V^T(t - 1)
    asm volatile("" ::: "memory");
    if (t + 2 < nt) stage(t + 2, true, false);
    if (t + 1 < nt) stage(t + 1, false, true);
    if (t + 1 < nt) scores(t + 1, sn);                        // in the matrix pipe while the VALU below works on tile t
    const unsigned char* vl = smem + 3 * 8192 + (t & 1) * 8192;
    const bool t1_live = (kt0 + 32) < p.nk;                   // wave-uniform
    if (kt0 + KT > p.nk) {                                    // ragged last tile (wave-uniform)
#pragma unroll
      for (int T = 0; T < NKT; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt0 + T * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf >= p.nk) s[T][r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int T = 0; T < NKT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[T][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_tile = mx * p.scale_log2;
    const bool grow = m_tile - m_run > 8.0f;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_tile) : 1.0f;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      m_run = grow ? m_tile : m_run;
    }
    float psum = 0.f;
    const float neg_m = -m_run;
#pragma unroll
    for (int T = 0; T < NKT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[T][r], p.scale_log2, neg_m));
        s[T][r] = pv;
        psum += pv;
      }
    psum += __shfl_xor(psum, 32);
    l_run += psum;
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      if (T == 1 && !t1_live) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (f16)s[T][8 * u + e];
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const f16x8 vf = *reinterpret_cast<const f16x8*>(vl + (d * 32 + frow) * 128 + (((4 * T + 2 * u + fhalf) ^ sw) << 4));
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[d], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int T = 0; T < NKT; ++T) s[T] = sn[T];
  }
  if (qrow < p.nq) {
    const float inv = 1.0f / l_run;
    f16* orow = ob + (long)qrow * p.so_seq;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int col = d * 32 + 8 * qd + 4 * fhalf;
        f16x4 o, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float val = oacc[d][4 * qd + r] * inv;
          o[r] = (f16)val;
          lo[r] = (f16)(val - (float)o[r]);
        }
        *reinterpret_cast<f16x4*>(orow + col) = o;
        if (p.lo_off) *reinterpret_cast<f16x4*>(orow + p.lo_off + col) = lo;
      }
  }
}

// ---- attention over the <= 32 frames of one pixel (ModelScope temporal self-attention, t2v_model.py:716-767, and
//      LVDM TemporalCrossAttention with relative-position terms, attention_temporal.py:107-144 / RelativePosition :46-65)
//   sim[t,s] = scale * q[t] . (k[s] + Ek[clip(s-t)])       p = softmax_s(sim)
//   out[t]   = sum_s p[t,s] * (v[s] + Ev[clip(s-t)])       clip(x) = clamp(x, -R, R) + R; tables [2R+1, D] fp32 (REL only)
// The relative-position terms make the score of every (t, s) pair a different dot product, so this is a VALU
// kernel (for plain temporal attention the MFMA kernel above measured faster, 24 frames x d=64): one wave per (pixel, head), four per workgroup.  q/k/v rows are staged once in LDS (fp16, padded pitch);
// each lane accumulates a TB x TB register block of scores (rows ti + 8a, columns sj + 8b) from 16-byte LDS reads,
// the row softmax is three xor-shuffles over the 8 lanes that share a row, P goes through LDS and the output is
// produced 8 channels per lane.  Addressing = the (sequence, outer, inner) strides of the MFMA kernel.
struct SeqAttnParams {
  const f16* q; const f16* k; const f16* v; f16* o;
  const float* ek; const float* ev;
  int T, D, heads, b_inner, R, lds_per_wave;
  int Tq, q_off;                                           // queries = frames [q_off, q_off + Tq) of the T key frames (T-sharded clip)
  long n_items;
  long sq_seq, sq_out, sq_in;
  long sk_seq, sk_out, sk_in;
  long so_seq, so_out, so_in;
  float scale;
  int lo_off;                                              // as AttnParams::lo_off
};

__device__ __forceinline__ f32x8 ld_f32x8(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return r;
}

template <int TB, bool REL>
__global__ __launch_bounds__(256) void seqattn_kernel(const SeqAttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long item = (long)blockIdx.x * 4 + wave;
  const bool live = item < p.n_items;                      // wave-uniform
  const int T = p.T, D = p.D, d8 = D >> 3;
  const int DP = D * 2 + 16;                               // LDS row pitch (bytes): 16-B reads of 8 rows hit disjoint banks
  unsigned char* qs = ssm + (size_t)wave * p.lds_per_wave;
  unsigned char* ks = qs + T * DP;
  unsigned char* vs = ks + T * DP;
  float* ps = reinterpret_cast<float*>(vs + T * DP);       // [T][T + 1]
  const int head = live ? (int)(item % p.heads) : 0;
  const long pos = live ? item / p.heads : 0;
  const long bo = pos / p.b_inner, bi = pos % p.b_inner;
  const f16* qb = p.q + bo * p.sq_out + bi * p.sq_in + head * D;
  const f16* kb = p.k + bo * p.sk_out + bi * p.sk_in + head * D;
  const f16* vb = p.v + bo * p.sk_out + bi * p.sk_in + head * D;
  f16* ob = p.o + bo * p.so_out + bi * p.so_in + head * D;
  const int Tq = p.Tq;
  if (live) {
    for (int u = lane; u < T * d8; u += 64) {
      const int t = u / d8, c = u - t * d8;
      if (t < Tq) *reinterpret_cast<f16x8*>(qs + t * DP + c * 16) = *reinterpret_cast<const f16x8*>(qb + (long)t * p.sq_seq + c * 8);
      *reinterpret_cast<f16x8*>(ks + t * DP + c * 16) = *reinterpret_cast<const f16x8*>(kb + (long)t * p.sk_seq + c * 8);
      *reinterpret_cast<f16x8*>(vs + t * DP + c * 16) = *reinterpret_cast<const f16x8*>(vb + (long)t * p.sk_seq + c * 8);
    }
  }
  __syncthreads();
  if (live) {
    const int ti = lane >> 3, sj = lane & 7;
    float acc[TB][TB];
#pragma unroll
    for (int a = 0; a < TB; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) acc[a][b] = 0.f;
    int rrow[2 * TB - 1];
    if (REL) {
#pragma unroll
      for (int e = 0; e < 2 * TB - 1; ++e) {
        int dlt = 8 * (e - (TB - 1)) + sj - ti - p.q_off;
        dlt = dlt < -p.R ? -p.R : (dlt > p.R ? p.R : dlt);
        rrow[e] = (dlt + p.R) * D;
      }
    }
    for (int c = 0; c < d8; ++c) {
      f16x8 qv[TB], kv[TB];
#pragma unroll
      for (int a = 0; a < TB; ++a) {
        const int t = ti + 8 * a;
        if (t < Tq) qv[a] = *reinterpret_cast<const f16x8*>(qs + t * DP + c * 16);
        else for (int j = 0; j < 8; ++j) qv[a][j] = (f16)0.f;
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const int sx = sj + 8 * b;
        if (sx < T) kv[b] = *reinterpret_cast<const f16x8*>(ks + sx * DP + c * 16);
        else for (int j = 0; j < 8; ++j) kv[b][j] = (f16)0.f;
      }
      if (REL) {
        f32x8 er[2 * TB - 1];
#pragma unroll
        for (int e = 0; e < 2 * TB - 1; ++e) er[e] = ld_f32x8(p.ek + rrow[e] + c * 8);
#pragma unroll
        for (int a = 0; a < TB; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[a][b] += (float)qv[a][j] * ((float)kv[b][j] + er[b - a + TB - 1][j]);
      } else {
#pragma unroll
        for (int a = 0; a < TB; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[a][b] += (float)qv[a][j] * (float)kv[b][j];
      }
    }
    // row softmax: a row's T scores live in TB registers of the 8 lanes with the same ti
#pragma unroll
    for (int a = 0; a < TB; ++a) {
      const int t = ti + 8 * a;
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        acc[a][b] = (sj + 8 * b < T) ? acc[a][b] * p.scale : -INFINITY;
        mx = fmaxf(mx, acc[a][b]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
      float sum = 0.f;
#pragma unroll
      for (int b = 0; b < TB; ++b) { acc[a][b] = __expf(acc[a][b] - mx); sum += acc[a][b]; }
      sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
      const float inv = 1.0f / sum;
      if (t < Tq) {
#pragma unroll
        for (int b = 0; b < TB; ++b)
          if (sj + 8 * b < T) ps[t * (T + 1) + sj + 8 * b] = acc[a][b] * inv;
      }
    }
  }
  __syncthreads();
  if (live) {
    for (int u = lane; u < Tq * d8; u += 64) {
      const int t = u / d8, c = u - t * d8;
      f32x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
      for (int sx = 0; sx < T; ++sx) {
        const float pv = ps[t * (T + 1) + sx];
        const f16x8 vv = *reinterpret_cast<const f16x8*>(vs + sx * DP + c * 16);
        if (REL) {
          int dlt = sx - t - p.q_off;
          dlt = dlt < -p.R ? -p.R : (dlt > p.R ? p.R : dlt);
          const f32x8 e = ld_f32x8(p.ev + (long)(dlt + p.R) * D + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += pv * ((float)vv[j] + e[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += pv * (float)vv[j];
        }
      }
      f16x8 oh, ol;
#pragma unroll
      for (int j = 0; j < 8; ++j) { oh[j] = (f16)o[j]; ol[j] = (f16)(o[j] - (float)oh[j]); }
      *reinterpret_cast<f16x8*>(ob + (long)t * p.so_seq + c * 8) = oh;
      if (p.lo_off) *reinterpret_cast<f16x8*>(ob + p.lo_off + (long)t * p.so_seq + c * 8) = ol;
    }
  }
}


// ---- relative-position temporal attention on MFMA (round 3) -------------------------------------------------------------
//   sim[t,s] = scale * (q[t].k[s] + q[t].Ek[s-t+R]),   out[t] = sum_s p[t,s] v[s] + sum_s p[t,s] Ev[s-t+R]
// for the unclipped case R >= T-1 (the released model: 16 frames, R = 16), all T queries of a pixel.  The two relative terms
// are GEMMs against the tables — QE^T = Ek . Q^T ([2T-1 relevant rows] x T) and O^T += Ev^T . Pskew^T — with a SKEW between
// them and the score / probability matrices: entry (t, s) pairs with table row j = s - t + R.  The skew goes through LDS: QE^T
// is written as [t][j] and read back at j = s - t + (T-1) by the lane that owns (t, s); the probabilities are written as rows
// [zeros | p[t, 0..T) | zeros] and the B operand of the Ev product reads them at column s = j + t - (T-1), zeros outside.
// One wave per (pixel, head), four per workgroup; the tables are staged once per workgroup as fp16 (exact for a .half()
// model).  Q and K fragments come straight from global memory (16 B per lane), V is transposed through LDS.
struct RelMfmaParams {
  SeqAttnParams a;
  int jbase;                  // first relevant table row: R - (T - 1)
  int nblk;                   // 32-row blocks of relevant table rows: 1 (T <= 16) or 2
};

template <int D>
__global__ __launch_bounds__(256) void relpos_mfma_kernel(const RelMfmaParams pp) {
  const SeqAttnParams& p = pp.a;
  constexpr int DK = (D + 15) / 16 * 16, NKK = DK / 16;
  constexpr int DV = (D + 31) / 32 * 32, NDT = DV / 32;
  constexpr int EK_PITCH = DK * 2 + 16;          // bytes per Ek row (fp16)
  constexpr int EV_PITCH = 64 * 2 + 8;           // bytes per Ev^T row: 64 table-row slots
  constexpr int VT_PITCH = 32 * 2 + 8;           // bytes per V^T row: 32 key slots
  constexpr int QE_PITCH = 65;                   // floats per QE row (t): 64 slots + 1
  constexpr int P_PITCH = 128;                   // halves per probability row: [32 zeros | 32 p | 64 zeros]
  constexpr int WAVE_BYTES = DV * VT_PITCH + 32 * QE_PITCH * 4 + 32 * P_PITCH * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
  unsigned char* ekl = ssm;                                      // [64][EK_PITCH]
  unsigned char* evt = ssm + 64 * EK_PITCH;                      // [DV][EV_PITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* wbase = ssm + 64 * EK_PITCH + DV * EV_PITCH + wave * WAVE_BYTES;
  unsigned char* vt = wbase;                                     // [DV][VT_PITCH]
  float* qeb = reinterpret_cast<float*>(wbase + DV * VT_PITCH);  // [32][QE_PITCH]
  f16* pb = reinterpret_cast<f16*>(wbase + DV * VT_PITCH + 32 * QE_PITCH * 4);   // [32][P_PITCH]
  const int T = p.T, R = p.R;
  const int nrows = 2 * R + 1;
  // ---- tables (whole workgroup): Ek rows jbase .. jbase+63 as fp16 [slot][d]; Ev transposed [d][slot]
  for (int u = tid; u < 64 * (DK / 8); u += 256) {
    const int jl = u / (DK / 8), c = u - jl * (DK / 8);
    const int j = jl + pp.jbase;
    f16x8 val;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = c * 8 + e;
      val[e] = (j < nrows && d < D) ? (f16)p.ek[(long)j * D + d] : (f16)0.f;
    }
    *reinterpret_cast<f16x8*>(ekl + jl * EK_PITCH + c * 16) = val;
  }
  for (int u = tid; u < DV * 64; u += 256) {
    const int d = u >> 6, jl = u & 63;
    const int j = jl + pp.jbase;
    *reinterpret_cast<f16*>(evt + d * EV_PITCH + jl * 2) = (j < nrows && d < D) ? (f16)p.ev[(long)j * D + d] : (f16)0.f;
  }
  // zero this wave's probability rows once: the pads stay zero, the 32 middle slots are rewritten per item
  for (int u = lane; u < 32 * P_PITCH / 8; u += 64) {
    const f16x8 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    reinterpret_cast<f16x8*>(pb)[u] = z;
  }
  __syncthreads();
  const long item = (long)blockIdx.x * 4 + wave;
  if (item >= p.n_items) return;                                 // wave-uniform; no workgroup barrier below
  const int head = (int)(item % p.heads);
  const long pos = item / p.heads;
  const long bo = pos / p.b_inner, bi = pos % p.b_inner;
  const f16* qb = p.q + bo * p.sq_out + bi * p.sq_in + head * D;
  const f16* kb = p.k + bo * p.sk_out + bi * p.sk_in + head * D;
  const f16* vb = p.v + bo * p.sk_out + bi * p.sk_in + head * D;
  f16* ob = p.o + bo * p.so_out + bi * p.so_in + head * D;
  const int frow = lane & 31, fhalf = lane >> 5;
  // ---- V^T into LDS: unit = (key pair, 4 d)
  for (int u = lane; u < 16 * (DV / 4); u += 64) {
    const int kp = u / (DV / 4), dq = u - kp * (DV / 4);
    const int key = 2 * kp;
    const bool dok = dq * 4 < D;
    f16x4 a, b;
    if (dok && key < T) a = *reinterpret_cast<const f16x4*>(vb + (long)key * p.sk_seq + dq * 4);
    else for (int e = 0; e < 4; ++e) a[e] = (f16)0.f;
    if (dok && key + 1 < T) b = *reinterpret_cast<const f16x4*>(vb + (long)(key + 1) * p.sk_seq + dq * 4);
    else for (int e = 0; e < 4; ++e) b[e] = (f16)0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      typedef f16 f16x2 __attribute__((ext_vector_type(2)));
      f16x2 w = {a[e], b[e]};
      *reinterpret_cast<f16x2*>(vt + (dq * 4 + e) * VT_PITCH + kp * 4) = w;
    }
  }
  // ---- Q (B operand) and K (A operand) fragments from global
  f16x8 qf[NKK], kf[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    const int d0 = kk * 16 + fhalf * 8;
    if (frow < T && d0 < D) {
      qf[kk] = *reinterpret_cast<const f16x8*>(qb + (long)frow * p.sq_seq + d0);
      kf[kk] = *reinterpret_cast<const f16x8*>(kb + (long)frow * p.sk_seq + d0);
    } else {
      for (int e = 0; e < 8; ++e) { qf[kk][e] = (f16)0.f; kf[kk][e] = (f16)0.f; }
    }
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // ---- S^T = K Q^T
  f32x16 sc = zero16;
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kk], qf[kk], sc, 0, 0, 0);
  // ---- QE^T = Ek Q^T for the relevant table rows, written skew-ready as [t][slot]
  for (int blk = 0; blk < pp.nblk; ++blk) {
    f32x16 qe = zero16;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const f16x8 ef = *reinterpret_cast<const f16x8*>(ekl + (blk * 32 + frow) * EK_PITCH + ((kk * 2 + fhalf) << 4));
      qe = __builtin_amdgcn_mfma_f32_32x32x16_f16(ef, qf[kk], qe, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)                     // accumulator rows 8 g + 4 fhalf + {0..3} of this block, column t = frow
#pragma unroll
      for (int e = 0; e < 4; ++e) qeb[frow * QE_PITCH + blk * 32 + 8 * g + 4 * fhalf + e] = qe[4 * g + e];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // the same wave reads back below
  // ---- scores + softmax for this lane's query t = frow
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
    float v = -INFINITY;
    if (key < T && frow < T) v = sc[r] + qeb[frow * QE_PITCH + key - frow + (T - 1)];
    sc[r] = v;
    mx = fmaxf(mx, v);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (frow >= T) mx = 0.f;
  const float sl2 = p.scale * 1.44269504088896340736f;
  const float neg_m = -mx * sl2;
  float psum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], sl2, neg_m));
    sc[r] = pv;
    psum += pv;
  }
  psum += __shfl_xor(psum, 32);
  // probabilities (unnormalised) into this query's LDS row: slots 32 + key
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f16x4 pw = {(f16)sc[4 * g], (f16)sc[4 * g + 1], (f16)sc[4 * g + 2], (f16)sc[4 * g + 3]};
    *reinterpret_cast<f16x4*>(pb + frow * P_PITCH + 32 + 8 * g + 4 * fhalf) = pw;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  // ---- O^T = V^T P^T + Ev^T Pskew^T
  f32x16 oacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) oacc[d] = zero16;
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    f16x8 pf;
#pragma unroll
    for (int e = 0; e < 8; ++e) pf[e] = (f16)sc[8 * t2 + e];
    const int kofs = (t2 * 16 + 4 * fhalf) * 2;
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      const unsigned char* vrow = vt + (d * 32 + frow) * VT_PITCH + kofs;
      const f16x4 lo = *reinterpret_cast<const f16x4*>(vrow);
      const f16x4 hi = *reinterpret_cast<const f16x4*>(vrow + 16);
      const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[d], 0, 0, 0);
    }
  }
  for (int u = 0; u < 2 * pp.nblk; ++u) {            // 16 table-row slots per step: slot = 16 u + 4 fhalf + (e & 3) + 8 (e >> 2)
    f16x8 pf;
    const f16* prow = pb + frow * P_PITCH + 32 + 16 * u + 4 * fhalf - (T - 1) + frow;     // column of slot e = 0 (s = slot + t - (T-1))
#pragma unroll
    for (int e = 0; e < 8; ++e) pf[e] = prow[(e & 3) + 8 * (e >> 2)];
    const int jofs = (16 * u + 4 * fhalf) * 2;
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      const unsigned char* erow = evt + (d * 32 + frow) * EV_PITCH + jofs;
      const f16x4 lo = *reinterpret_cast<const f16x4*>(erow);
      const f16x4 hi = *reinterpret_cast<const f16x4*>(erow + 16);
      const f16x8 ef = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ef, pf, oacc[d], 0, 0, 0);
    }
  }
  if (frow < T) {
    const float inv = 1.0f / psum;
    f16* orow = ob + (long)frow * p.so_seq;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int col = d * 32 + 8 * qd + 4 * fhalf;
        if (col < D) {
          f16x4 o, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float val = oacc[d][4 * qd + r] * inv;
            o[r] = (f16)val;
            lo[r] = (f16)(val - (float)o[r]);
          }
          *reinterpret_cast<f16x4*>(orow + col) = o;
          if (p.lo_off) *reinterpret_cast<f16x4*>(orow + p.lo_off + col) = lo;
        }
      }
  }
}

template <int D>
hipError_t launch_relpos_mfma(const SeqAttnParams& p, hipStream_t s) {
  constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32;
  constexpr int lds = 64 * (DK * 2 + 16) + DV * (64 * 2 + 8) + 4 * (DV * (32 * 2 + 8) + 32 * 65 * 4 + 32 * 128 * 2);
  static_assert(lds <= 160 * 1024, "LDS budget");
  RelMfmaParams pp;
  pp.a = p;
  pp.jbase = p.R - (p.T - 1);
  pp.nblk = (2 * p.T - 1 > 32) ? 2 : 1;
  auto kern = relpos_mfma_kernel<D>;
  static t2v_device_flags attr_set;       // per (instantiation, device)
  {
    const hipError_t e = t2v_set_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_set, s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((p.n_items + 3) / 4)), dim3(256), lds, s, pp);
  return hipGetLastError();
}

// ---- relative-position temporal attention on MFMA, clips of <= 16 frames (round 5) ---------------------------------------
// The same products as relpos_mfma_kernel, with the three things that made that kernel lose to the VALU kernel removed:
//  * the tables arrive READY from the host — Ek as fp16 rows [32 slots][DK] and Ev TRANSPOSED as fp16 [DV][32 slots], slot jl = table
//    row jl + R - (T-1), zero-padded (packing: videocrafter.py) — so staging them is a straight copy, not 10 k scalar conversions with a
//    transposing gather per workgroup;
//  * waves are PERSISTENT: the grid is what the chip holds at once and every wave walks its items, so the tables are staged once per
//    workgroup for all of them (the 32x32 level: 8 items per wave);
//  * T <= 16 means ONE 16-key step and ONE 32-slot table block: V^T, the skew buffer and the probability rows shrink to 6.7-10.5 KB per
//    wave (was 21), the workgroup to 35 / 45 / 65 KB at head_dim 40 / 80 / 160 -> 4 / 3 / 2 workgroups per CU instead of one.
struct Rel16Params {
  SeqAttnParams a;
  const f16* ek16;            // [32][DK]
  const f16* evt16;           // [DV][32]
};

template <int D>
__global__ __launch_bounds__(256) void relpos16_kernel(const Rel16Params pp) {
  const SeqAttnParams& p = pp.a;
  constexpr int DK = (D + 15) / 16 * 16, NKK = DK / 16;
  constexpr int DV = (D + 31) / 32 * 32, NDT = DV / 32;
  constexpr int EK_PITCH = DK * 2 + 16;          // bytes per Ek slot row
  constexpr int EV_PITCH = 32 * 2 + 8;           // bytes per Ev^T row: 32 slots
  constexpr int VT_PITCH = 16 * 2 + 8;           // bytes per V^T row: 16 keys
  constexpr int QE_PITCH = 33;                   // floats per QE row (query t): 32 slots + 1
  constexpr int P_PITCH = 64;                    // halves per probability row: [16 zeros | p[t, 0..16) | 32 zeros]
  constexpr int WAVE_BYTES = DV * VT_PITCH + 16 * QE_PITCH * 4 + 16 * P_PITCH * 2;
  static_assert(WAVE_BYTES % 16 == 0 && (DV * VT_PITCH) % 16 == 0 && (32 * EK_PITCH + DV * EV_PITCH) % 16 == 0, "LDS carve-up alignment");
  extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
  unsigned char* ekl = ssm;                                      // [32][EK_PITCH]
  unsigned char* evt = ssm + 32 * EK_PITCH;                      // [DV][EV_PITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* wbase = ssm + 32 * EK_PITCH + DV * EV_PITCH + wave * WAVE_BYTES;
  unsigned char* vt = wbase;                                     // [DV][VT_PITCH]
  float* qeb = reinterpret_cast<float*>(wbase + DV * VT_PITCH);  // [16][QE_PITCH]
  f16* pb = reinterpret_cast<f16*>(wbase + DV * VT_PITCH + 16 * QE_PITCH * 4);   // [16][P_PITCH]
  const int T = p.T;
  for (int u = tid; u < 32 * (DK / 8); u += 256) {
    const int jl = u / (DK / 8), c = u - jl * (DK / 8);
    *reinterpret_cast<f16x8*>(ekl + jl * EK_PITCH + c * 16) = *reinterpret_cast<const f16x8*>(pp.ek16 + jl * DK + c * 8);
  }
  for (int u = tid; u < DV * 8; u += 256) {
    const int d = u >> 3, c = u & 7;
    *reinterpret_cast<f16x4*>(evt + d * EV_PITCH + c * 8) = *reinterpret_cast<const f16x4*>(pp.evt16 + d * 32 + c * 4);
  }
  // zero this wave's probability rows once: the pads stay zero, the 16 middle slots are rewritten per item
  for (int u = lane; u < 16 * P_PITCH / 8; u += 64) {
    const f16x8 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    reinterpret_cast<f16x8*>(pb)[u] = z;
  }
  __syncthreads();                                               // the only workgroup barrier: the item loop below is per wave
  const int frow = lane & 31, fhalf = lane >> 5;
  const int trow = frow < 16 ? frow : 15;                        // LDS row of lanes that own no query (their results are discarded)
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float sl2 = p.scale * 1.44269504088896340736f;
  for (long item = (long)blockIdx.x * 4 + wave; item < p.n_items; item += (long)gridDim.x * 4) {
    const int head = (int)(item % p.heads);
    const long pos = item / p.heads;
    const long bo = pos / p.b_inner, bi = pos % p.b_inner;
    const f16* qb = p.q + bo * p.sq_out + bi * p.sq_in + head * D;
    const f16* kb = p.k + bo * p.sk_out + bi * p.sk_in + head * D;
    const f16* vb = p.v + bo * p.sk_out + bi * p.sk_in + head * D;
    f16* ob = p.o + bo * p.so_out + bi * p.so_in + head * D;
    // ---- V^T into LDS: unit = (key pair, 4 d)
    for (int u = lane; u < 8 * (DV / 4); u += 64) {
      const int kp = u / (DV / 4), dq = u - kp * (DV / 4);
      const int key = 2 * kp;
      const bool dok = dq * 4 < D;
      f16x4 a, b;
      if (dok && key < T) a = *reinterpret_cast<const f16x4*>(vb + (long)key * p.sk_seq + dq * 4);
      else for (int e = 0; e < 4; ++e) a[e] = (f16)0.f;
      if (dok && key + 1 < T) b = *reinterpret_cast<const f16x4*>(vb + (long)(key + 1) * p.sk_seq + dq * 4);
      else for (int e = 0; e < 4; ++e) b[e] = (f16)0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        typedef f16 f16x2 __attribute__((ext_vector_type(2)));
        f16x2 w = {a[e], b[e]};
        *reinterpret_cast<f16x2*>(vt + (dq * 4 + e) * VT_PITCH + kp * 4) = w;
      }
    }
    // ---- S^T = K Q^T (K fragments are used once, Q fragments again for the table product)
    f16x8 qf[NKK];
    f32x16 sc = zero16;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int d0 = kk * 16 + fhalf * 8;
      f16x8 kf;
      if (frow < T && d0 < D) {
        qf[kk] = *reinterpret_cast<const f16x8*>(qb + (long)frow * p.sq_seq + d0);
        kf = *reinterpret_cast<const f16x8*>(kb + (long)frow * p.sk_seq + d0);
      } else {
        for (int e = 0; e < 8; ++e) { qf[kk][e] = (f16)0.f; kf[e] = (f16)0.f; }
      }
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], sc, 0, 0, 0);
    }
    // ---- QE^T = Ek Q^T over the 32 table slots, written skew-ready as [t][slot]
    {
      f32x16 qe = zero16;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const f16x8 ef = *reinterpret_cast<const f16x8*>(ekl + frow * EK_PITCH + ((kk * 2 + fhalf) << 4));
        qe = __builtin_amdgcn_mfma_f32_32x32x16_f16(ef, qf[kk], qe, 0, 0, 0);
      }
      if (frow < 16) {
#pragma unroll
        for (int g = 0; g < 4; ++g)                   // accumulator rows (slots) 8 g + 4 fhalf + {0..3}, column t = frow
#pragma unroll
          for (int e = 0; e < 4; ++e) qeb[frow * QE_PITCH + 8 * g + 4 * fhalf + e] = qe[4 * g + e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // the same wave reads back below
    // ---- scores + softmax of this lane's query t = frow: keys (r & 3) + 8 (r >> 2) + 4 fhalf, r < 8 (keys 16 .. 31 do not exist)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      const float rel = qeb[trow * QE_PITCH + key - trow + 15 - (16 - T)];     // slot = key - t + (T - 1)
      const float v = (key < T && frow < T) ? sc[r] + rel : -INFINITY;
      sc[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (frow >= T) mx = 0.f;
    const float neg_m = -mx * sl2;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], sl2, neg_m));
      sc[r] = pv;
      psum += pv;
    }
    psum += __shfl_xor(psum, 32);
    // probabilities (unnormalised) into this query's LDS row: slots 16 + key
    if (frow < 16) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f16x4 pw = {(f16)sc[4 * g], (f16)sc[4 * g + 1], (f16)sc[4 * g + 2], (f16)sc[4 * g + 3]};
        *reinterpret_cast<f16x4*>(pb + frow * P_PITCH + 16 + 8 * g + 4 * fhalf) = pw;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // ---- O^T = V^T P^T + Ev^T Pskew^T
    f32x16 oacc[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d) oacc[d] = zero16;
    {
      f16x8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[e] = (f16)sc[e];
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const unsigned char* vrow = vt + (d * 32 + frow) * VT_PITCH + 8 * fhalf;
        const f16x4 lo = *reinterpret_cast<const f16x4*>(vrow);
        const f16x4 hi = *reinterpret_cast<const f16x4*>(vrow + 16);
        const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[d], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {                       // 16 table slots per step: slot = 16 u + 4 fhalf + (e & 3) + 8 (e >> 2)
      f16x8 pf;
      const f16* prow = pb + trow * P_PITCH + 16 + 16 * u + 4 * fhalf - (T - 1) + trow;       // column of slot e = 0 (s = slot + t - (T-1))
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[e] = prow[(e & 3) + 8 * (e >> 2)];
      const int jofs = (16 * u + 4 * fhalf) * 2;
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const unsigned char* erow = evt + (d * 32 + frow) * EV_PITCH + jofs;
        const f16x4 lo = *reinterpret_cast<const f16x4*>(erow);
        const f16x4 hi = *reinterpret_cast<const f16x4*>(erow + 16);
        const f16x8 ef = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ef, pf, oacc[d], 0, 0, 0);
      }
    }
    if (frow < T) {
      const float inv = 1.0f / psum;
      f16* orow = ob + (long)frow * p.so_seq;
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int col = d * 32 + 8 * qd + 4 * fhalf;
          if (col < D) {
            f16x4 o, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float val = oacc[d][4 * qd + r] * inv;
              o[r] = (f16)val;
              lo[r] = (f16)(val - (float)o[r]);
            }
            *reinterpret_cast<f16x4*>(orow + col) = o;
            if (p.lo_off) *reinterpret_cast<f16x4*>(orow + p.lo_off + col) = lo;
          }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // this item's LDS reads before the next item's writes
  }
}

template <int D>
hipError_t launch_relpos16(const SeqAttnParams& p, const f16* ek16, const f16* evt16, hipStream_t s) {
  constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32;
  constexpr int lds = 32 * (DK * 2 + 16) + DV * (32 * 2 + 8) + 4 * (DV * (16 * 2 + 8) + 16 * 33 * 4 + 16 * 64 * 2);
  static_assert(lds <= 160 * 1024, "LDS budget");
  Rel16Params pp;
  pp.a = p;
  pp.ek16 = ek16;
  pp.evt16 = evt16;
  auto kern = relpos16_kernel<D>;
  static t2v_device_flags attr_set;       // per (instantiation, device)
  {
    const hipError_t e = t2v_set_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_set, s);
    if (e != hipSuccess) return e;
  }
  const int per_cu = (160 * 1024 / lds) < 4 ? (160 * 1024 / lds) : 4;
  const long want = (p.n_items + 3) / 4, hold = (long)t2v_num_cus(s) * per_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)(want < hold ? want : hold)), dim3(256), lds, s, pp);
  return hipGetLastError();
}

template <bool REL>
hipError_t launch_seqattn(const t2v_op& op, hipStream_t s) {
  SeqAttnParams p;
  p.q = reinterpret_cast<const f16*>(op.p[0]);
  p.k = reinterpret_cast<const f16*>(op.p[1]);
  p.v = reinterpret_cast<const f16*>(op.p[2]);
  p.o = reinterpret_cast<f16*>(op.p[3]);
  p.ek = reinterpret_cast<const float*>(op.p[4]);
  p.ev = reinterpret_cast<const float*>(op.p[5]);
  p.Tq = op.i[0]; p.T = op.i[1]; p.heads = op.i[2]; p.b_inner = op.i[4]; p.D = op.i[14] > 0 ? op.i[14] : 64; p.R = op.i[15];
  p.q_off = op.i[16];
  p.n_items = (long)op.i[3] * op.i[4] * p.heads;
  p.sq_seq = op.i[5]; p.sq_out = op.i[6]; p.sq_in = op.i[7];
  p.sk_seq = op.i[8]; p.sk_out = op.i[9]; p.sk_in = op.i[10];
  p.so_seq = op.i[11]; p.so_out = op.i[12]; p.so_in = op.i[13];
  p.scale = op.f[0];
  p.lo_off = REL ? op.i[18] : 0;
  if (p.T <= 0 || p.T > 32 || p.Tq <= 0 || p.q_off < 0 || p.q_off + p.Tq > p.T || p.D <= 0 || p.D % 8 != 0 || p.D > 160 || p.n_items <= 0)
    return hipErrorInvalidValue;
  if (REL && (p.R < 0 || p.ek == nullptr || p.ev == nullptr)) return hipErrorInvalidValue;
  if (REL && op.i[17] == 2 && op.p[6] != 0 && op.p[7] != 0 && p.Tq == p.T && p.q_off == 0 && p.T <= 16 && p.R >= p.T - 1 && p.scale > 0.f) {
    // i[17] = 2 (round 5): the persistent MFMA kernel for whole clips of <= 16 frames, tables pre-packed by the host (p[6], p[7])
    const f16* ek16 = reinterpret_cast<const f16*>(op.p[6]);
    const f16* evt16 = reinterpret_cast<const f16*>(op.p[7]);
    switch (p.D) {
      case 40: return launch_relpos16<40>(p, ek16, evt16, s);
      case 64: return launch_relpos16<64>(p, ek16, evt16, s);
      case 80: return launch_relpos16<80>(p, ek16, evt16, s);
      case 160: return launch_relpos16<160>(p, ek16, evt16, s);
      default: break;
    }
  }
  if (REL && op.i[17] == 1 && p.Tq == p.T && p.q_off == 0 && p.R >= p.T - 1 && p.R <= 31 && p.scale > 0.f) {
    // i[17]: the MFMA kernel for the unclipped, unsharded form (the released model: 16 frames, R = 16).  Correct (op tests against
    // the interpreter and the explicit formula), but MEASURED SLOWER than the VALU kernel below on the VideoCrafter step (2.25 vs
    // 1.65 ms for the 32 launches): its LDS footprint (tables + skew buffers, ~100 KB per 4-wave workgroup) leaves 4 waves per CU
    // against 24, and every workgroup re-stages the tables for four 16x16 problems.  Kept as an opt-in (T2V_RELPOS_MFMA=1); round 5's
    // relpos16_kernel (i[17] = 2, above) removes both and is the default for clips of <= 16 frames (0.80 ms for the same 32 launches).
    {
      switch (p.D) {
        case 40: return launch_relpos_mfma<40>(p, s);
        case 64: return launch_relpos_mfma<64>(p, s);
        case 80: return launch_relpos_mfma<80>(p, s);
        case 160: return launch_relpos_mfma<160>(p, s);
        default: break;
      }
    }
  }
  p.lds_per_wave = (3 * p.T * (p.D * 2 + 16) + p.T * (p.T + 1) * 4 + 15) / 16 * 16;
  const int lds = 4 * p.lds_per_wave;
  const dim3 grid((unsigned)((p.n_items + 3) / 4));
  const int tb = (p.T + 7) / 8;
  auto go = [&](auto kern) -> hipError_t {
    static t2v_device_flags attr_set;      // per (instantiation, device); the lambda is instantiated per kernel type
    {
      const hipError_t e = t2v_set_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set, s);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    return hipGetLastError();
  };
  switch (tb) {
    case 1: return go(seqattn_kernel<1, REL>);
    case 2: return go(seqattn_kernel<2, REL>);
    case 3: return go(seqattn_kernel<3, REL>);
    default: return go(seqattn_kernel<4, REL>);
  }
}

// ---- row softmax fp32 -> fp16 (VAE AttnBlock, autoencoder_modules.py:104-106) -----------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* in, f16* out, int rows, int cols,
                                                           int ld_in, int ld_out, float scale) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  if (row >= rows) return;
  const float* x = in + (size_t)row * ld_in;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, x[c] * scale);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) sum += __expf(x[c] * scale - mx);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / sum;
  f16* y = out + (size_t)row * ld_out;
  for (int c = tid; c < cols; c += 256) y[c] = (f16)(__expf(x[c] * scale - mx) * inv);
}

}  // namespace

hipError_t t2v_launch_attention(const t2v_op& op, hipStream_t s) {
  AttnParams p;
  p.q = reinterpret_cast<const f16*>(op.p[0]);
  p.k = reinterpret_cast<const f16*>(op.p[1]);
  p.v = reinterpret_cast<const f16*>(op.p[2]);
  p.o = reinterpret_cast<f16*>(op.p[3]);
  p.nq = op.i[0]; p.nk = op.i[1]; p.heads = op.i[2]; p.b_outer = op.i[3]; p.b_inner = op.i[4];
  p.sq_seq = op.i[5]; p.sq_out = op.i[6]; p.sq_in = op.i[7];
  p.sk_seq = op.i[8]; p.sk_out = op.i[9]; p.sk_in = op.i[10];
  p.so_seq = op.i[11]; p.so_out = op.i[12]; p.so_in = op.i[13];
  p.scale_log2 = op.f[0] * 1.44269504088896340736f;
  p.causal = op.i[15] != 0;
  p.lo_off = op.i[16];
  if (p.lo_off < 0) return hipErrorInvalidValue;
  if (p.nq <= 0 || p.nk <= 0 || !(op.f[0] > 0.f)) return hipErrorInvalidValue;
  const int nbatch = p.b_outer * p.b_inner;
  const int hd = op.i[14] > 0 ? op.i[14] : 64;
  if (op.p[6] != 0) {
    // round 6: V transposed once in HBM into the scratch p[6] (fp16 [batch * heads * 64, i[17]]), K / V^T tiles by LDS-DMA (attn2_kernel);
    // validated by the executor: head_dim 64, not causal, i[17] = keys padded to a multiple of 64
    f16* vt = reinterpret_cast<f16*>(op.p[6]);
    const int n_pad = op.i[17];
    hipLaunchKernelGGL(vt_transpose_kernel, dim3(n_pad / 64, p.heads, nbatch), dim3(256), 0, s, p, vt, n_pad);
    if (op.i[18] == 4) hipLaunchKernelGGL((attn2_kernel<4>), dim3((p.nq + 127) / 128, p.heads, nbatch), dim3(256), 0, s, p, vt, n_pad);
#ifdef T2V_ATTN2_EXPERIMENTS   // the score-pipelined variant: i[18] = 12 (4 waves, 3 per SIMD) / 16 (8 waves, 2 per SIMD) / 20 (8 waves, 3 per SIMD)
    else if (op.i[18] == 12) hipLaunchKernelGGL((attn2p_kernel<4, 3>), dim3((p.nq + 127) / 128, p.heads, nbatch), dim3(256), 0, s, p, vt, n_pad);
    else if (op.i[18] == 16) hipLaunchKernelGGL((attn2p_kernel<8, 2>), dim3((p.nq + 255) / 256, p.heads, nbatch), dim3(512), 0, s, p, vt, n_pad);
    else if (op.i[18] == 20) hipLaunchKernelGGL((attn2p_kernel<8, 3>), dim3((p.nq + 255) / 256, p.heads, nbatch), dim3(512), 0, s, p, vt, n_pad);
#endif
    else hipLaunchKernelGGL((attn2_kernel<8>), dim3((p.nq + 255) / 256, p.heads, nbatch), dim3(512), 0, s, p, vt, n_pad);
    return hipGetLastError();
  }
  const bool small = p.nq <= 32;
  static const bool prefetch_env = [] { const char* e = getenv("T2V_ATTN_PREFETCH"); return e ? atoi(e) != 0 : false; }();
  const bool prefetch = prefetch_env && p.nk > KT_MAIN;
  const dim3 g1(1, p.heads, nbatch), g4((p.nq + 127) / 128, p.heads, nbatch);
  switch (hd) {
    case 40:
      if (small && p.nk <= 32) hipLaunchKernelGGL((attn_kernel<1, 40, 32>), g1, dim3(64), 0, s, p);
      else if (small) hipLaunchKernelGGL((attn_kernel<1, 40>), g1, dim3(64), 0, s, p);
      else if (prefetch) hipLaunchKernelGGL((attn_kernel<4, 40, 64, true>), g4, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<4, 40>), g4, dim3(256), 0, s, p);
      break;
    case 64:
      if (small && p.nk <= 32) hipLaunchKernelGGL((attn_kernel<1, 64, 32>), g1, dim3(64), 0, s, p);
      else if (small) hipLaunchKernelGGL((attn_kernel<1, 64>), g1, dim3(64), 0, s, p);
      else if (prefetch) hipLaunchKernelGGL((attn_kernel<4, 64, 64, true>), g4, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<4, 64>), g4, dim3(256), 0, s, p);
      break;
    case 80:
      if (small && p.nk <= 32) hipLaunchKernelGGL((attn_kernel<1, 80, 32>), g1, dim3(64), 0, s, p);
      else if (small) hipLaunchKernelGGL((attn_kernel<1, 80>), g1, dim3(64), 0, s, p);
      else if (prefetch) hipLaunchKernelGGL((attn_kernel<4, 80, 64, true>), g4, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<4, 80>), g4, dim3(256), 0, s, p);
      break;
    case 160:
      if (small && p.nk <= 32) hipLaunchKernelGGL((attn_kernel<1, 160, 32>), g1, dim3(64), 0, s, p);
      else if (small) hipLaunchKernelGGL((attn_kernel<1, 160>), g1, dim3(64), 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<4, 160>), g4, dim3(256), 0, s, p);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t t2v_launch_relpos_attention(const t2v_op& op, hipStream_t s) { return launch_seqattn<true>(op, s); }

hipError_t t2v_launch_softmax(const t2v_op& op, hipStream_t s) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(op.i[0]), dim3(256), 0, s,
                     reinterpret_cast<const float*>(op.p[0]), reinterpret_cast<f16*>(op.p[1]), op.i[0],
                     op.i[1], op.i[2], op.i[3], op.f[0]);
  return hipGetLastError();
}
