// Implicit-GEMM convolution / linear kernel for gfx950 (MI355X, CDNA4).
//
//   out[M,N] = epilogue( gather(A)[M,K] · W[N,K]^T )      fp16 operands, fp32 accumulate
//
// Replaces, on the reference hot path:
//   nn.Conv2d 3x3 (ResBlock in/out, Up/Downsample, stem, head)   t2v_model.py:929,956,870,1034,167,323
//   nn.Conv3d (3,1,1) (TemporalConvBlock_v2)                      t2v_model.py:1201-1212
//   nn.Conv2d 1x1 / nn.Conv1d k=1 / nn.Linear (+GEGLU)            t2v_model.py:965,693,709,533-538,817
//
// Data layout: activations are channels-last tokens [(b f) h w, C] (row m = one pixel of one
// frame, C contiguous) so every reduction index k is memory-contiguous; weights are packed
// [N, K] with K ordered (64-channel chunk, tap, channel) for the convolutions.  No im2col buffer exists: the A tile of a
// convolution is gathered straight from the activation tensor by per-lane source addresses of
// the LDS-DMA loads, and zero padding / M,N,K tails read a 256-byte zero page.
//
// Tiling (wave64): block tile BM x BN x 64, WM x WN waves, each wave owns (BM/WM) x (BN/WN)
// as 32x32 MFMA tiles.  Operands are swapped into the MFMA (A-operand = weight rows,
// B-operand = activation rows) so that each lane ends up holding 4 *consecutive output
// channels* of one token per accumulator quad: stores are 8/16-byte vectors in the
// channels-last output and bias / residual / rowbias are vector loads.
//
// Staging: `global_load_lds` 16-byte LDS-DMA, double-buffered.  The LDS image of a tile is
// [rows][64] fp16 (128 B per row, 8 chunks of 16 B); chunk c of row r is stored at physical
// chunk c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads conflict-free.  The
// DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address
// (guide rule 21) and again on the read.
#include "t2v_kernels.h"

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

__device__ __attribute__((aligned(256))) unsigned char t2v_zero_page[256];

namespace {

constexpr int BK = 64;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_wave_base, 16, 0, 0);
}

// ---- fused epilogue on 4 consecutive output channels of one token ---------------------
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, float v0, float v1,
                                               float v2, float v3) {
  if (p.bias) {
    if (p.bias_m) {
      const float b = p.bias[m];
      v0 += b; v1 += b; v2 += b; v3 += b;
    } else {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
      v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
    }
  }
  if (p.rowbias) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)(m / p.rows_per_batch) * p.ldrb + n);
    v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
  }
  if (p.act == 1) {
    v0 = t2v_silu(v0); v1 = t2v_silu(v1); v2 = t2v_silu(v2); v3 = t2v_silu(v3);
  }
  if (p.res) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldr + n);
    v0 += r[0]; v1 += r[1]; v2 += r[2]; v3 += r[3];
  }
  if (p.out_f32) {
    f32x4 o = {v0, v1, v2, v3};
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = o;
  } else {
    f16x4 o = {(f16)v0, (f16)v1, (f16)v2, (f16)v3};
    f16* dst = reinterpret_cast<f16*>(p.out) + (size_t)m * p.ldc + n;
    *reinterpret_cast<f16x4*>(dst) = o;
    if (p.out_lo) {       // low-order image of the rounding, beside the row's N values (GemmParams::out_lo)
      const f16x4 l = {(f16)(v0 - (float)o[0]), (f16)(v1 - (float)o[1]), (f16)(v2 - (float)o[2]), (f16)(v3 - (float)o[3])};
      *reinterpret_cast<f16x4*>(dst + p.N) = l;
    }
  }
}

// GEGLU: packed columns n (value quad) and n+8 (gate quad) -> output column n_out
__device__ __forceinline__ void epilogue_store_geglu(const GemmParams& p, int m, int n_val, int n_out,
                                                     const float* v, const float* g) {
  float bv[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0};
  if (p.bias) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.bias + n_val);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n_val + 8);
    for (int r = 0; r < 4; ++r) { bv[r] = a[r]; bg[r] = b[r]; }
  }
  f16x4 o;
  for (int r = 0; r < 4; ++r) o[r] = (f16)((v[r] + bv[r]) * t2v_gelu_erf(g[r] + bg[r]));
  *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.out) + (size_t)m * p.ldc + n_out) = o;
}

// XE = 2: the plain epilogue is t2v_epilogue_rows_gn — GroupNorm (+SiLU) of the result inside the epilogue, statistics across the launch's
// workgroups at a grid barrier (T2V_EPI_GN; its own instantiations, launched only on a co-resident grid); XE = 3: t2v_epilogue_rows_lnx, the
// LayerNorm second output across the launch's column tiles (same exchange along the rows)
template <int BM, int BN, int WM, int WN, int GATHER, int XE = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 32;       // 32-row token tiles per wave
  constexpr int TN = BN / WN / 32;       // 32-row channel tiles per wave
  constexpr int XS = BM / 8 / NW;        // 1-KiB DMA slabs (8 rows) of the activation tile per wave
  constexpr int WS = BN / 8 / NW;        // ... of the weight tile per wave
  constexpr int X_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = X_BYTES + W_BYTES;
  static_assert(TM >= 1 && TN >= 1 && XS >= 1 && WS >= 1, "tile too small for wave grid");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M - p.m_begin + BM - 1) / BM;     // (m_begin: t2v_launch_coresident; else 0)
  int tile_m, tile_n;
  t2v_tile_of_block(blockIdx.x, tiles_m, tiles_n, p.panel, tile_m, tile_n);   // XCD-aware order
  const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;

  const int KT = (p.K + BK - 1) / BK;
  const int kt_begin = blockIdx.y * p.kt_per_split;
  const int kt_end = min(KT, kt_begin + p.kt_per_split);

  const unsigned char* zero = t2v_zero_page;

  // ---- per-lane DMA descriptors ------------------------------------------------------
  // lane -> (row within slab, physical 16-B chunk); logical chunk differs per row (swizzle)
  const int lrow = lane >> 3;  // 0..7
  const int pchunk = lane & 7;

  // activation rows handled by this lane
  long xoff[XS];                    // PLAIN: element offset of row start; conv: image base row
  int xy[XS], xx[XS];               // conv: (yo*stride, xo*stride); tconv: frame index in xy
  bool xvalid[XS];
  int xlchunk[XS];
#pragma unroll
  for (int j = 0; j < XS; ++j) {
    const int r = (wave * XS + j) * 8 + lrow;
    const int m = m0 + r;
    xlchunk[j] = pchunk ^ ((r >> 1) & 7);
    xvalid[j] = m < p.M;
    xoff[j] = 0; xy[j] = 0; xx[j] = 0;
    if (GATHER == T2V_GATHER_PLAIN) {
      xoff[j] = (long)((p.a_wrap && m >= p.a_wrap) ? m - p.a_wrap : m) * p.lda;
    } else if (GATHER == T2V_GATHER_CONV3X3 || GATHER == T2V_GATHER_CONV3X3_C8) {
      const int hw = p.Hout * p.Wout;
      const int img = m / hw, rem = m - img * hw;
      const int yo = rem / p.Wout, xo = rem - yo * p.Wout;
      xoff[j] = (long)img * p.Hin * p.Win;
      xy[j] = yo * p.stride;
      xx[j] = xo * p.stride;
    } else {  // TCONV3 (halo layout: input rows [clip][F+2][HW], every tap in range)
      const int clip = m / (p.HW * p.F);
      xoff[j] = p.halo ? (long)m + (long)(2 * clip + 1) * p.HW : (long)m;
      xy[j] = p.halo ? 1 : (m / p.HW) - clip * p.F;
    }
  }
  // weight rows handled by this lane
  const f16* wrow[WS];
  int wlchunk[WS];
#pragma unroll
  for (int j = 0; j < WS; ++j) {
    const int r = (wave * WS + j) * 8 + lrow;
    const int n = n0 + r;
    wlchunk[j] = pchunk ^ ((r >> 1) & 7);
    wrow[j] = (n < p.N) ? (p.W + (size_t)n * p.ldw) : nullptr;
  }

  auto stage = [&](int buf, int kt) {
    unsigned char* xt = smem + buf * STAGE;
    unsigned char* wt = xt + X_BYTES;
    const int k0 = kt * BK;
    // reduction order is (64-channel chunk, tap, channel): k-tile kt = chunk * TAPS + tap
    int tap = 0, ci0 = k0;
    if (GATHER == T2V_GATHER_CONV3X3 || GATHER == T2V_GATHER_TCONV3) {
      const int taps = GATHER == T2V_GATHER_CONV3X3 ? 9 : 3;
      const int chunk = kt / taps;
      tap = kt - chunk * taps;
      ci0 = chunk * BK;
    }
#pragma unroll
    for (int j = 0; j < XS; ++j) {
      const int kc = k0 + xlchunk[j] * 8;  // first k of this lane's chunk
      const void* src = zero;
      if (GATHER == T2V_GATHER_PLAIN) {
        if (xvalid[j] && kc < p.K) src = p.A + xoff[j] + kc;
      } else if (GATHER == T2V_GATHER_CONV3X3) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int pl = p.halo ? 0 : 1;       // conv3x3: halo = 1 selects the (0,1,0,1) padding of the LDM encoder Downsample
        const int yv = xy[j] + ky - pl, xv = xx[j] + kx - pl;
        const int hv = p.Hin << p.up, wv = p.Win << p.up;
        if (xvalid[j] && yv >= 0 && yv < hv && xv >= 0 && xv < wv) {
          const long row = xoff[j] + (long)(yv >> p.up) * p.Win + (xv >> p.up);
          src = p.A + row * p.lda + ci0 + xlchunk[j] * 8;
        }
      } else if (GATHER == T2V_GATHER_CONV3X3_C8) {
        const int t = kc >> 3;  // one 8-channel chunk == one tap
        const int ky = t / 3, kx = t - ky * 3;
        const int yv = xy[j] + ky - 1, xv = xx[j] + kx - 1;
        const int hv = p.Hin << p.up, wv = p.Win << p.up;
        if (xvalid[j] && t < 9 && yv >= 0 && yv < hv && xv >= 0 && xv < wv) {
          const long row = xoff[j] + (long)(yv >> p.up) * p.Win + (xv >> p.up);
          src = p.A + row * p.lda;
        }
      } else {  // TCONV3
        const int fi = xy[j] + tap - 1;
        if (xvalid[j] && (p.halo || (fi >= 0 && fi < p.F))) {
          const long row = xoff[j] + (long)(tap - 1) * p.HW;
          src = p.A + row * p.lda + ci0 + xlchunk[j] * 8;
        }
      }
      glds16(src, xt + (wave * XS + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < WS; ++j) {
      const int kc = k0 + wlchunk[j] * 8;
      const void* src = (wrow[j] != nullptr && kc < p.K) ? (const void*)(wrow[j] + kc) : (const void*)zero;
      glds16(src, wt + (wave * WS + j) * 1024);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (kt_begin < kt_end) {
    stage(0, kt_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    const int frow = lane & 31;   // fragment row within a 32-row MFMA tile
    const int fhalf = lane >> 5;  // which 8-wide k group of the 16-deep MFMA step
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      if (kt + 1 < kt_end) stage(buf ^ 1, kt + 1);
      const unsigned char* xt = smem + buf * STAGE;
      const unsigned char* wt = xt + X_BYTES;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        const int lchunk = kk * 2 + fhalf;
        f16x8 xf[TM], wf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          const int r = (wm * TM + a) * 32 + frow;
          xf[a] = *reinterpret_cast<const f16x8*>(xt + r * 128 + ((lchunk ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int r = (wn * TN + b) * 32 + frow;
          wf[b] = *reinterpret_cast<const f16x8*>(wt + r * 128 + ((lchunk ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            // D[i = channel][j = token] += W[i][k] * X[j][k]
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue: lane holds token (lane & 31), channels 8q + 4*(lane>>5) + {0..3} -------
  if (p.epi == T2V_EPI_NONE) {      // row-coalesced through a per-wave LDS buffer (t2v_kernels.h); also the split-K slabs
    __syncthreads();                // every wave is done reading the operand stages
    if constexpr (XE == 2) {
      t2v_epilogue_rows_gn<WM, WN, TM, TN>(p, acc, smem, lane, wave, m0, n0, tile_m, tile_n, tiles_m, tiles_n);
      return;
    }
    if constexpr (XE == 3) {
      t2v_epilogue_rows_lnx<WM, WN, TM, TN>(p, acc, smem, lane, wave, m0, n0, tile_m, tile_n, tiles_n);
      return;
    }
    if constexpr (XE == 4) {
      t2v_epilogue_xattn<WM, WN, TM, TN>(p, acc, smem, lane, wave, m0, n0);
      return;
    }
    t2v_epilogue_rows<TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wave * (32 * T2V_EPI_SP), lane, m0 + wm * TM * 32,
                              n0 + wn * TN * 32, blockIdx.y, tile_m * tiles_n + tile_n);
    return;
  }
  const int mlane = lane & 31, nhalf = (lane >> 5) * 4;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int m = m0 + (wm * TM + a) * 32 + mlane;
    if (m >= p.M) continue;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int nt = n0 + (wn * TN + b) * 32;  // first packed channel of this MFMA tile
      if (p.splitk > 1) {
        float* ws = p.ws + ((size_t)blockIdx.y * p.M + m) * p.N;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nt + 8 * q + nhalf;
          if (n < p.N) {
            f32x4 o = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
            *reinterpret_cast<f32x4*>(ws + n) = o;
          }
        }
      } else if (p.epi == T2V_EPI_GEGLU) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int n_val = nt + 16 * qq + nhalf;
          if (n_val < p.N) {
            float v[4], g[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[a][b][8 * qq + r]; g[r] = acc[a][b][8 * qq + 4 + r]; }
            epilogue_store_geglu(p, m, n_val, (nt >> 1) + 8 * qq + nhalf, v, g);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nt + 8 * q + nhalf;
          if (n < p.N)
            epilogue_store(p, m, n, acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2],
                           acc[a][b][4 * q + 3]);
        }
      }
    }
  }
}

// split-K reduction + fused epilogue: one thread per 4 output channels
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
  const int geglu = p.epi == T2V_EPI_GEGLU;
  const int ncols = geglu ? p.N / 2 : p.N;   // output columns
  const int quads = ncols / 4;
  const long total = (long)p.M * quads;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / quads);
    const int co = (int)(idx - (long)m * quads) * 4;
    if (!geglu) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < p.splitk; ++z)
        s += *reinterpret_cast<const f32x4*>(p.ws + ((size_t)z * p.M + m) * p.N + co);
      epilogue_store(p, m, co, s[0], s[1], s[2], s[3]);
    } else {
      const int n_val = (co >> 3) * 16 + (co & 4);
      f32x4 sv = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < p.splitk; ++z) {
        const float* base = p.ws + ((size_t)z * p.M + m) * p.N;
        sv += *reinterpret_cast<const f32x4*>(base + n_val);
        sg += *reinterpret_cast<const f32x4*>(base + n_val + 8);
      }
      float v[4] = {sv[0], sv[1], sv[2], sv[3]}, g[4] = {sg[0], sg[1], sg[2], sg[3]};
      epilogue_store_geglu(p, m, n_val, co, v, g);
    }
  }
}

template <int BM, int BN, int WM, int WN>
hipError_t launch_tile(const GemmParams& pin, hipStream_t s) {
  GemmParams p = pin;
  p.panel = t2v_choose_panel(p, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const dim3 grid(tiles, p.splitk > 1 ? p.splitk : 1);
  const dim3 block(WM * WN * 64);
  constexpr int lds = 2 * (BM + BN) * BK * 2;
  if (p.xa_k != nullptr) {         // fused to_q + text cross-attention (T2V_EPI_XATTN): the 128x128 tile = two heads per column tile, plain gather
    if constexpr (BM == 128 && BN == 128) {
      static_assert(t2v_xattn_epilogue_lds(BM, BN) <= lds, "Q [BM][BN] fp16 re-uses the operand stages");
      if (p.splitk != 1 || p.gather != T2V_GATHER_PLAIN) return hipErrorInvalidValue;
      auto k = gemm_kernel<BM, BN, WM, WN, T2V_GATHER_PLAIN, 4>;
      static t2v_device_flags once_xa;
      (void)t2v_set_dynamic_lds(reinterpret_cast<const void*>(k), lds, once_xa, s);
      hipLaunchKernelGGL(k, grid, block, lds, s, p);
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  if ((p.gn_out != nullptr && p.splitk == 1) || p.ln_x) {
    // GroupNorm inside the epilogue: the 128x128 tile only; the grid barrier needs the whole launch resident (no split-K, the grid
    // within what the occupancy API grants this instantiation on the stream's device)
    if constexpr (BM == 128 && BN == 128) {
      static_assert(t2v_gn_epilogue_lds(WM * WN, BM / 32, BN) <= lds && t2v_lnx_epilogue_lds(WM * WN, WN, BM) <= lds, "the norm epilogues re-use the operand stages");
      if (p.splitk != 1 || !t2v_coop_allowed() || p.gather == T2V_GATHER_CONV3X3_C8) return hipErrorInvalidValue;
      auto launch = [&](auto k, int* occ, t2v_device_flags& once) {
        (void)t2v_set_dynamic_lds(reinterpret_cast<const void*>(k), lds, once, s);
        // (round 6: a grid the device does not hold co-resident is cut into row chunks of whole tiles and whole instances)
        const long cap = t2v_grid_capacity(reinterpret_cast<const void*>(k), WM * WN * 64, lds, s, occ);
        return t2v_launch_coresident(p, BM, (p.N + BN - 1) / BN, cap, p.ln_x ? BM : t2v_lcm(BM, p.gn_rows), p.ln_x ? BM * 16 : 2 * T2V_GN_PIECES * 16,
                                     [&](const GemmParams& q, int nwg) {
          hipLaunchKernelGGL(k, dim3(nwg, 1), block, lds, s, q);
          return hipGetLastError();
        });
      };
      static int occ0[T2V_MAX_DEVICES] = {}, occ1[T2V_MAX_DEVICES] = {}, occ2[T2V_MAX_DEVICES] = {};
      static t2v_device_flags g0, g1, g2;
      static int occ3[T2V_MAX_DEVICES] = {};
      static t2v_device_flags g3;
      if (p.ln_x) return p.gather == T2V_GATHER_PLAIN ? launch(gemm_kernel<BM, BN, WM, WN, T2V_GATHER_PLAIN, 3>, occ3, g3) : hipErrorInvalidValue;
      if (p.gather == T2V_GATHER_PLAIN) return launch(gemm_kernel<BM, BN, WM, WN, T2V_GATHER_PLAIN, 2>, occ0, g0);
      if (p.gather == T2V_GATHER_CONV3X3) return launch(gemm_kernel<BM, BN, WM, WN, T2V_GATHER_CONV3X3, 2>, occ1, g1);
      if (p.gather == T2V_GATHER_TCONV3) return launch(gemm_kernel<BM, BN, WM, WN, T2V_GATHER_TCONV3, 2>, occ2, g2);
    }
    return hipErrorInvalidValue;
  }
  switch (p.gather) {
    case T2V_GATHER_PLAIN: {
      auto k = gemm_kernel<BM, BN, WM, WN, T2V_GATHER_PLAIN>;
      static t2v_device_flags once0;
      (void)t2v_set_dynamic_lds(reinterpret_cast<const void*>(k), lds, once0, s);
      hipLaunchKernelGGL(k, grid, block, lds, s, p);
      break;
    }
    case T2V_GATHER_CONV3X3: {
      auto k = gemm_kernel<BM, BN, WM, WN, T2V_GATHER_CONV3X3>;
      static t2v_device_flags once1;
      (void)t2v_set_dynamic_lds(reinterpret_cast<const void*>(k), lds, once1, s);
      hipLaunchKernelGGL(k, grid, block, lds, s, p);
      break;
    }
    case T2V_GATHER_TCONV3: {
      auto k = gemm_kernel<BM, BN, WM, WN, T2V_GATHER_TCONV3>;
      static t2v_device_flags once2;
      (void)t2v_set_dynamic_lds(reinterpret_cast<const void*>(k), lds, once2, s);
      hipLaunchKernelGGL(k, grid, block, lds, s, p);
      break;
    }
    case T2V_GATHER_CONV3X3_C8: {
      auto k = gemm_kernel<BM, BN, WM, WN, T2V_GATHER_CONV3X3_C8>;
      static t2v_device_flags once3;
      (void)t2v_set_dynamic_lds(reinterpret_cast<const void*>(k), lds, once3, s);
      hipLaunchKernelGGL(k, grid, block, lds, s, p);
      break;
    }
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

hipError_t t2v_launch_splitk_reduce(const GemmParams& p, hipStream_t s) {
  const long quads = (long)p.M * ((p.epi == T2V_EPI_GEGLU ? p.N / 2 : p.N) / 4);
  const int blocks = (int)min((long)2048, (quads + 255) / 256);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t t2v_launch_gemm(const GemmParams& pin, hipStream_t s) {
  GemmParams p = pin;
  {
    const int KT = (p.K + BK - 1) / BK;
    if (p.splitk > KT) p.splitk = KT;
    if (p.splitk < 1) p.splitk = 1;
    p.kt_per_split = (KT + p.splitk - 1) / p.splitk;
    p.splitk = (KT + p.kt_per_split - 1) / p.kt_per_split;  // no empty splits
  }
  hipError_t e;
  // Tile choice: 128x128 by default; 128x64 when the last 128-wide column tile would be at
  // most half full (N = 320, 960, 4, 8 ...), to avoid 25-97 % padded columns.
  const bool narrow = (p.N % 128 != 0) && (p.N % 128 <= 64);
  {
    const int bn = narrow ? 64 : 128;
    const long tiles = (long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
    // in-kernel fold by the last-arriving workgroup of a tile (t2v_epilogue_rows) where a ticket buffer is given; else the reduction kernel
    if (p.splitk <= 1 || p.epi != T2V_EPI_NONE || tiles > T2V_SYNC_INTS || p.gn_out != nullptr) p.tickets = nullptr;
  }
  if (narrow)
    e = launch_tile<128, 64, 4, 1>(p, s);
  else
    e = launch_tile<128, 128, 2, 2>(p, s);
  if (e != hipSuccess) return e;
  // (split-K whose result feeds a fused GroupNorm: the reduction is the loader of a cooperative GroupNorm launch, norm.hip)
  if (p.splitk > 1 && p.tickets == nullptr) e = p.gn_out != nullptr ? t2v_launch_splitk_reduce_gn(p, s) : t2v_launch_splitk_reduce(p, s);
  return e;
}
