// Second-generation implicit-GEMM kernel (large tiles, deep LDS-DMA ring) for gfx950.
//
// Same contract as gemm.hip (out[M,N] = epilogue(gather(A)[M,K] · W[N,K]^T), fp16 operands,
// fp32 accumulate, swapped MFMA operands so a lane owns 4 consecutive output channels), but
// built for the regime the UNet lives in: a 128x128 tile needs ~64 B/clk/CU of operand
// traffic at MFMA peak — above what one CU gets from L2 — so tiles here are 256x256,
// 256x320, 128x256 and 128x320 (29-45 B/clk/CU).  N = 320*k (every C-output GEMM of the
// ModelScope UNet: 320, 640, 960, 1280, 1920, 2560 ...) uses the 320-wide tile with no padded
// columns.
//
// Pipeline: k-tiles of 32 (64-byte LDS rows), STAGES-deep ring filled by `global_load_lds`
// 16-byte LDS-DMA; STAGES-1 k-tiles are always in flight.  Per k-tile each wave executes
//     s_waitcnt vmcnt(LPS*(STAGES-2))   -> its own DMA pieces of the tile to compute have landed
//     s_barrier                          -> everyone's have; everyone finished the previous tile
//     issue the DMA of tile t+STAGES-1 into the slot just freed
//     ds_read_b128 fragments + MFMA 32x32x16
// i.e. ONE barrier per k-tile and no vmcnt(0) drain anywhere in the main loop (the loads of
// k-tiles past the end are redirected to a zero page so the outstanding-load count is constant).
// LDS image: row r, 16-byte chunk c at r*64 + ((c ^ ((r>>2)&3))<<4): conflict-free for the
// fragment reads; the XOR is applied to the per-lane DMA *source* (the destination is
// lane-linear) and again on the read.
#include "t2v_kernels.h"

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))


namespace {

__device__ __attribute__((aligned(256))) unsigned char g2_zero_page[256];

// internal gather id: 3x3 conv with the nearest-2x upsample folded in (no affine tap offset)
constexpr int G_CONV_UP = 100;
constexpr int BM_OF(int wm, int tm) { return wm * tm * 32; }


__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the same with a count that is a constant only after loop unrolling (the switch folds away); n < 0: no wait
__device__ __forceinline__ void wait_vmcnt_n(int n) {
#define T2V_W(N) case N: wait_vmcnt<N>(); break;
  switch (n) {
    T2V_W(0) T2V_W(1) T2V_W(2) T2V_W(3) T2V_W(4) T2V_W(5) T2V_W(6) T2V_W(7) T2V_W(8) T2V_W(9) T2V_W(10) T2V_W(11) T2V_W(12) T2V_W(13) T2V_W(14) T2V_W(15)
    T2V_W(16) T2V_W(17) T2V_W(18) T2V_W(19) T2V_W(20) T2V_W(21) T2V_W(22) T2V_W(23) T2V_W(24) T2V_W(25) T2V_W(26) T2V_W(27) T2V_W(28) T2V_W(29) T2V_W(30)
    T2V_W(31) T2V_W(32) T2V_W(33) T2V_W(34) T2V_W(35) T2V_W(36) T2V_W(37) T2V_W(38) T2V_W(39) T2V_W(40)
    default: break;
  }
#undef T2V_W
}

// ---- epilogue helpers (identical semantics to gemm.hip) ----------------------------------
__device__ __forceinline__ void epi_store(const GemmParams& p, int m, int n, float v0, float v1, float v2, float v3) {
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

__device__ __forceinline__ void epi_store_geglu(const GemmParams& p, int m, int n_val, int n_out, const float* v,
                                                const float* g) {
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

#ifndef T2V_G2_PRIO
#define T2V_G2_PRIO 1          // experiment switch: 0 no priorities, 1 the MFMA segment at priority 1, 2 the load segment at priority 1
#endif
#ifndef T2V_G2_DMAFIRST
#define T2V_G2_DMAFIRST 0      // experiment switch: 1 = the phase's DMA pieces before its fragment reads
#endif
// ---- region schedule of the staggered two-group main loop (PP == 4, round 6) -----------------------------------------------------------
// A k-tile is cut into P = (TM / 2) * TN phases of 8 MFMAs (one PAIR of token sub-tiles x one weight sub-tile x BK); the operands of a
// k-tile are staged as REGIONS — X pair a (read by every wave in phase a * TN only) and W sub-tile b (read in phase b only: the
// fragments stay in registers for the second pair) — so that a region's LDS can be refilled for k-tile t + 2 two phases after its
// only read in k-tile t, long before the rest of the buffer is free.  The table below places every DMA piece of a wave in the
// earliest phase its region allows (at most CAP pieces per phase) and derives, per phase, the `s_waitcnt vmcnt(N)` that retires
// exactly the regions first read in the NEXT phase: nothing is ever drained, every piece has about one whole k-tile to land.
template <int WM, int WN, int TM, int TN>
struct G2Sched {
  static constexpr int NAP = TM / 2, P = NAP * TN;
  static constexpr int XR = WM, WR = WN / 2;                 // DMA pieces per wave per X-pair region / per W region
  static constexpr int NREG = NAP + TN, LPS = NAP * XR + TN * WR;
  static constexpr int CAP = (LPS + P - 1) / P;
  int u[LPS] = {};           // piece j is staged in phase u % P of k-tile T for k-tile T + 2 - u / P
  int pos[LPS] = {};         // position of piece j in the issue stream of one period
  int vm[P] = {};            // vmcnt after the staging of phase q (-1: no region becomes due)
  static constexpr int first_of(int r) { return r < NAP ? r * TN : r - NAP; }
  static constexpr int region_of(int j) { return j < NAP * XR ? j / XR : NAP + (j - NAP * XR) / WR; }
  constexpr G2Sched() {
    int order[NREG] = {};
    bool used[NREG] = {};
    for (int i = 0; i < NREG; ++i) {             // regions by the phase they become free (= read phase + 2), X before W
      int best = -1;
      for (int r = 0; r < NREG; ++r)
        if (!used[r] && (best < 0 || first_of(r) < first_of(best))) best = r;
      used[best] = true;
      order[i] = best;
    }
    int cur = 0, cnt = 0, n = 0;
    for (int i = 0; i < NREG; ++i) {
      const int r = order[i], f = first_of(r) + 2;
      for (int j = 0; j < LPS; ++j) {
        if (region_of(j) != r) continue;
        if (f > cur) { cur = f; cnt = 0; }
        u[j] = cur;
        pos[j] = n++;
        if (++cnt == CAP) { ++cur; cnt = 0; }
      }
    }
    for (int q = 0; q < P; ++q) {
      const int due = (q + 1) % P, t_req = (q + 1) / P;       // regions first read in the next phase (of k-tile t_req relative to this one)
      int best = -1;
      for (int j = 0; j < LPS; ++j) {
        if (first_of(region_of(j)) != due) continue;
        const int gj = P * (t_req - 2) + u[j];              // when piece j of that k-tile was issued (this k-tile's phase 0 = 0)
        int c = 0;
        for (int k = 0; k < LPS; ++k)
          for (int m = -4; m <= 2; ++m) {
            const int g = u[k] + P * m;
            if ((g > gj && g <= q) || (g == gj && pos[k] > pos[j])) ++c;
          }
        if (best < 0 || c < best) best = c;
      }
      vm[q] = best;
    }
  }
  constexpr bool feasible() const {
    for (int j = 0; j < LPS; ++j)
      if (u[j] > 2 * P + first_of(region_of(j)) - 1 || u[j] >= 2 * P) return false;
    return true;
  }
};

// WM x WN waves; each wave owns TM x TN MFMA tiles (32 tokens x 32 channels each).
// XE: extra epilogue of the plain (T2V_EPI_NONE) path — 0 none, 1 fused LayerNorm second output (whole-row tiles), 2 fused GroupNorm
// (+SiLU) of the result with a grid barrier (T2V_EPI_GN); separate instantiations, so the plain kernels keep their register budgets.
template <int WM, int WN, int TM, int TN, int BK, int STAGES, int MINW, int GATHER, int PP, int XE = 0, bool TAT = false>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm2_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int ROW_BYTES = BK * 2;                          // 64 (BK=32) or 128 (BK=64) bytes per LDS row
  constexpr int RPS = 1024 / ROW_BYTES;                      // rows per 1-KiB DMA piece: 16 or 8
  constexpr int CPR = BK / 8;                                // 16-byte chunks per row: 4 or 8
  constexpr int XSLABS = BM / RPS, WSLABS = BN / RPS;        // 1-KiB DMA pieces per tile
  static_assert(XSLABS % NW == 0, "token slabs must divide evenly over the waves");
  constexpr int XPW = XSLABS / NW;
  constexpr int WPW = (WSLABS + NW - 1) / NW;                // the last may be a dummy piece
  constexpr int LPS = XPW + WPW;                             // DMA instructions per wave per stage
  constexpr bool REGION = PP == 4;                           // pieces grouped by region (G2Sched) instead of interleaved over the tile
  static_assert(!REGION || (NW == 8 && TM % 2 == 0 && (WN == 2 || WN == 4) && WSLABS % NW == 0), "region schedule: 8 waves, token sub-tiles in pairs");
  constexpr int STAGE_BYTES = (XSLABS + WSLABS) * 1024;
  constexpr int DUMMY_OFF = STAGES * STAGE_BYTES;            // 1 KiB scratch for dummy pieces
  static_assert(LPS * (STAGES - 1) < 64, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware tile order (t2v_kernels.h): each XCD walks one contiguous run of the panel numbering
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M - p.m_begin + BM - 1) / BM;        // (m_begin: row chunk of a fused-norm launch, t2v_launch_coresident; else 0)
  int tile_m, tile_n;
  t2v_tile_of_block(blockIdx.x, tiles_m, tiles_n, p.panel, tile_m, tile_n);
  const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;

  const int KT = (p.K + BK - 1) / BK;
  const int kt_begin = blockIdx.y * p.kt_per_split;          // in units of BK-wide k-tiles
  const int kt_end = min(KT, kt_begin + p.kt_per_split);
  const int nkt = kt_end - kt_begin;

  const unsigned char* zero = g2_zero_page;

  // ---- per-lane DMA state -------------------------------------------------------------------
  // Every lane keeps ONE running source pointer per DMA piece; a k-tile costs one 64-bit add per
  // piece.  Conv gathers recompute the pointers only when the filter tap changes (9 / 3 times per
  // workgroup); rows that fall into padding, past M / N, or belong to a dummy piece park on the
  // zero page with step 0.  (K % BK == 0 is guaranteed by the dispatcher.)
  const int lrow = lane / CPR;     // row inside a 1-KiB piece
  const int pchunk = lane % CPR;   // physical 16-B chunk inside the row
  auto swz = [](int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
  constexpr int STEP = BK * 2;     // bytes per k-tile along a row

  constexpr int TAPS = (GATHER == T2V_GATHER_CONV3X3 || GATHER == G_CONV_UP) ? 9 : (GATHER == T2V_GATHER_TCONV3 ? 3 : 1);
  constexpr bool general = GATHER == G_CONV_UP;
  const unsigned char* xptr[XPW];  // PLAIN: running pointer; conv: pointer of the CENTRE tap, chunk 0
  int xstep[XPW];                  // PLAIN only
  unsigned xmask[XPW];             // conv: bit t set <=> tap t of this row is inside the image / clip
  long xoff[general ? XPW : 1];    // only the upsample path keeps per-row coordinates
  int xy[general ? XPW : 1], xx[general ? XPW : 1];
  // first tile row of DMA piece j of this wave.  Interleaved: piece s = wave + j * NW of the tile.  Region order (PP == 4): X piece
  // j = a * WM + i is piece idx = wave + 8 i of the WM * 8 pieces of pair a (rows wm' * TM * 32 + a * 64 + ...), W piece j = b * (WN / 2) + i
  // piece idx = wave + 8 i of the WN * 4 pieces of weight sub-tile b (rows wn' * TN * 32 + b * 32 + ...)
  auto xrow0 = [&](int j) {
    if constexpr (REGION) {
      const int a = j / WM, idx = wave + 8 * (j % WM);
      return (idx >> 3) * (TM * 32) + a * 64 + (idx & 7) * 8;
    } else {
      return (wave + j * NW) * RPS;
    }
  };
  auto wrow0 = [&](int j) {
    if constexpr (REGION) {
      const int b = j / (WN / 2), idx = wave + 8 * (j % (WN / 2));
      return (idx >> 2) * (TN * 32) + b * 32 + (idx & 3) * 8;
    } else {
      return (wave + j * NW) * RPS;
    }
  };
#pragma unroll
  for (int j = 0; j < XPW; ++j) {
    const int r = xrow0(j) + lrow;                // row inside the token tile
    const int m = m0 + r;
    const int lc = pchunk ^ swz(r);
    const bool valid = m < p.M;
    xmask[j] = 0; xstep[j] = 0;
    xptr[j] = zero;
    if (GATHER == T2V_GATHER_PLAIN) {
      if constexpr (TAT) {
        // fused QKV + temporal attention: the tile's rows are p.tpix pixels x p.F frames of ONE sample (row r = pixel r / F,
        // frame r % F), so that every sequence (the frames of a pixel) is complete inside the workgroup
        const int pl = r / p.F, f = r - pl * p.F;
        const int smp = tile_m / p.tiles_ps, pix = (tile_m - smp * p.tiles_ps) * p.tpix + pl;
        if (pl < p.tpix && pix < p.HW) {
          const long mm = ((long)smp * p.F + f) * p.HW + pix;
          xptr[j] = reinterpret_cast<const unsigned char*>(p.A + mm * p.lda + (long)kt_begin * BK + lc * 8);
          xstep[j] = STEP;
        }
      } else if (valid || REGION) {
        // (region schedule: rows past M read the last row instead of the zero page — their results are never stored — so that every
        //  pointer advances by the same constant and no per-piece step register is needed)
        const int mc = REGION ? min(m, p.M - 1) : m;
        const int ma = (p.a_wrap && mc >= p.a_wrap) ? mc - p.a_wrap : mc;     // shared (one-sample) operand: rows wrap once
        xptr[j] = reinterpret_cast<const unsigned char*>(p.A + (long)ma * p.lda + (long)kt_begin * BK + lc * 8);
        xstep[j] = STEP;
      }
    } else if (GATHER == T2V_GATHER_CONV3X3 || GATHER == G_CONV_UP) {
      const int hw = p.Hout * p.Wout;
      const int img = m / hw, rem = m - img * hw;
      const int yo = rem / p.Wout, xo = rem - yo * p.Wout;
      const long ibase = (long)img * p.Hin * p.Win;
      const int ys = yo * p.stride, xs = xo * p.stride;
      xptr[j] = reinterpret_cast<const unsigned char*>(p.A + (ibase + (long)ys * p.Win + xs) * p.lda + lc * 8);
      const int pl = p.halo ? 0 : 1;     // conv3x3: halo = 1 -> taps at +0..+2 (LDM encoder Downsample pads only bottom / right)
      for (int t = 0; t < 9; ++t) {
        const int yv = ys + t / 3 - pl, xv = xs + t % 3 - pl;
        if (valid && yv >= 0 && yv < (p.Hin << p.up) && xv >= 0 && xv < (p.Win << p.up)) xmask[j] |= 1u << t;
      }
      if (general) {
        xoff[j] = ibase;
        xy[j] = ys;
        xx[j] = xs | (lc << 24);     // keep the chunk too (coordinates are < 2^24)
      }
    } else {  // TCONV3
      const int clip = m / (p.HW * p.F);
      const int f = (m / p.HW) - clip * p.F;
      // halo layout: input rows are [clip][F+2][HW]; output row m reads input frames f, f+1, f+2
      const long in_row = p.halo ? (long)m + (long)(2 * clip + 1) * p.HW : (long)m;
      xptr[j] = reinterpret_cast<const unsigned char*>(p.A + in_row * p.lda + lc * 8);
      for (int t = 0; t < 3; ++t)
        if (valid && (p.halo || (f + t - 1 >= 0 && f + t - 1 < p.F))) xmask[j] |= 1u << t;
    }
  }
  int tap = 0, chunk = 0;          // wave-uniform position of the NEXT k-tile to be staged
  if (GATHER != T2V_GATHER_PLAIN) {
    chunk = kt_begin / TAPS;
    tap = kt_begin - chunk * TAPS;
  }
  const unsigned char* wptr[WPW];
  int wstep[WPW];
  bool wdummy[WPW];
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int r = wrow0(j) + lrow;               // row inside the weight tile
    const int n = n0 + r;
    wdummy[j] = wrow0(j) >= BN;                  // wave-uniform
    const bool ok = REGION || (!wdummy[j] && n < p.N);
    const int nc = REGION ? min(n, p.N - 1) : n;     // (region schedule: channels past N read the last row, see the token rows above)
    wptr[j] = ok ? reinterpret_cast<const unsigned char*>(p.W + (size_t)nc * p.ldw + kt_begin * BK + (pchunk ^ swz(r)) * 8) : zero;
    wstep[j] = ok ? STEP : 0;
  }

  // Staging one k-tile = LPS DMA instructions per wave.  They are issued as `pieces` so that the
  // main loop can spread them between the MFMA k-steps: an LDS-DMA instruction holds the wave's
  // issue port for ~60+ cycles while the TA walks its 64 addresses, so issuing all of them
  // back-to-back before the MFMAs (as a monolithic stage() would) serialises DMA issue and MFMA
  // execution — measured: 27 GB/s/CU of operand delivery, 50 % of wave cycles parked.
  int staged = 0;                  // k-tiles whose staging has been started (wave-uniform)
  bool live = true;                // staging position still inside [kt_begin, kt_end)
  long boff = 0;                   // conv: wave-uniform byte offset of the k-tile being staged
  int ktap = 0;
  auto stage_begin = [&]() {
    live = staged < nkt;
    if (GATHER != T2V_GATHER_PLAIN) {
      ktap = tap;
      long delta = 0;
      if (GATHER == T2V_GATHER_CONV3X3) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int pl = p.halo ? 0 : 1;
        delta = ((long)(ky - pl) * p.Win + (kx - pl)) * p.lda;
      } else if (GATHER == T2V_GATHER_TCONV3) {
        delta = (long)(tap - 1) * p.HW * p.lda;
      }
      boff = (delta + (long)chunk * BK) * 2;
    }
  };
  // source of piece j of the k-tile being staged (per-lane pointer; advances the running pointers) and its 1-KiB LDS destination
  auto piece_src = [&](int j) -> const void* {     // j in [0, LPS): token pieces first, then weights
    const void* src = zero;
    if (j < XPW) {
      if (GATHER == T2V_GATHER_PLAIN) {
        if (live) { src = xptr[j]; xptr[j] += xstep[j]; }
      } else if (!general) {
        if (live && ((xmask[j] >> ktap) & 1u)) src = xptr[j] + boff;
      } else if (general) {
        // nearest-2x upsample folded into the gather (3 convs per forward): per-lane recompute
        const int ky = ktap / 3, kx = ktap - ky * 3;
        const int lc = xx[j] >> 24, x0 = xx[j] & 0xFFFFFF;
        const int yv = xy[j] + ky - 1, xv = x0 + kx - 1;      // (upsample path: symmetric padding only)
        const long row = xoff[j] + (long)(yv >> 1) * p.Win + (xv >> 1);
        if (live && ((xmask[j] >> ktap) & 1u)) src = p.A + row * p.lda + (long)chunk * BK + lc * 8;
      }
    } else {
      const int jw = j - XPW;
      if (live) { src = wptr[jw]; wptr[jw] += wstep[jw]; }
    }
    return src;
  };
  auto piece_dst = [&](int slot, int j) -> unsigned char* {
    unsigned char* base = smem + slot * STAGE_BYTES;
    if (j < XPW) return base + (xrow0(j) / RPS) * 1024;
    const int jw = j - XPW;
    return wdummy[jw] ? (smem + DUMMY_OFF) : (base + (XSLABS + wrow0(jw) / RPS) * 1024);
  };
  auto stage_piece = [&](int slot, int j) {
#ifdef T2V_G2_NODMA      // timing experiment only (wrong results): the main loop WITHOUT its operand DMA — what do the LDS-DMA instructions cost?
    if (staged >= STAGES - 1) return;
#endif
    const void* src = piece_src(j);
    glds16(src, piece_dst(slot, j));
  };
  auto stage_end = [&]() {
    if (GATHER != T2V_GATHER_PLAIN && live) {
      if (++tap == TAPS) { tap = 0; ++chunk; }
    }
    ++staged;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // prologue: STAGES-1 k-tiles in flight (the register-staged schedule, PP == 3, has its own)
  if constexpr (PP != 3 && PP != 4) {
#pragma unroll
    for (int g = 0; g < STAGES - 1; ++g) {
      stage_begin();
#pragma unroll
      for (int j = 0; j < LPS; ++j) stage_piece(g, j);
      stage_end();
    }
  }

  // fragment read addressing: lane reads row (tile_row0 + lane&31), logical chunk kk*2 + (lane>>5);
  // per tile-row keep the byte base and the swizzle term, one xor-add per read
  const int frow = lane & 31, fhalf = lane >> 5;
  int xbase[TM], xsw[TM], wbase[TN], wsw[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int r = (wm * TM + a) * 32 + frow;
    xbase[a] = r * ROW_BYTES;
    xsw[a] = swz(r) << 4;
  }
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int r = (wn * TN + b) * 32 + frow;
    wbase[b] = XSLABS * 1024 + r * ROW_BYTES;
    wsw[b] = swz(r) << 4;
  }

  constexpr int KSTEPS = BK / 16;
  if constexpr (PP == 0) {
    int slot = 0;
#ifdef T2V_G2_TIMING     // experiment build (tools/build_variant.py): where does a wave park in a k-tile?  cycles -> p.ws[wave slot][4]
    unsigned long long tm_vm = 0, tm_bar = 0, tm_lds = 0;
    const unsigned long long tm_begin = clock64();
#endif
    for (int t = 0; t < nkt; ++t) {
#ifdef T2V_G2_TIMING
      const unsigned long long tm0 = clock64();
#endif
      wait_vmcnt<LPS*(STAGES - 2)>();
#ifdef T2V_G2_TIMING
      const unsigned long long tm1 = clock64();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef T2V_G2_TIMING
      const unsigned long long tm2 = clock64();
      tm_vm += tm1 - tm0; tm_bar += tm2 - tm1;
#endif
      int fs = slot + STAGES - 1;
      if (fs >= STAGES) fs -= STAGES;
      stage_begin();
      const unsigned char* st = smem + slot * STAGE_BYTES;
  #pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const int lc4 = (kk * 2 + fhalf) << 4;
        f16x8 xf[TM], wf[TN];
  #pragma unroll
        for (int a = 0; a < TM; ++a) xf[a] = *reinterpret_cast<const f16x8*>(st + xbase[a] + (lc4 ^ xsw[a]));
  #pragma unroll
        for (int b = 0; b < TN; ++b) wf[b] = *reinterpret_cast<const f16x8*>(st + wbase[b] + (lc4 ^ wsw[b]));
#ifdef T2V_G2_TIMING
        if (kk == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tm_lds += clock64() - tm2; }
#endif
        // this k-step's share of the next tile's DMA, issued between the fragment reads and the MFMAs.
        // With a 2-deep ring the pieces must land before the next barrier, so they go out in the first
        // half of the k-tile; with 3 stages they have a whole extra k-tile and are spread over all steps.
        constexpr int SPREAD = STAGES == 2 ? KSTEPS / 2 : KSTEPS;
        if (kk < SPREAD) {
  #pragma unroll
          for (int j = (LPS * kk) / SPREAD; j < (LPS * (kk + 1)) / SPREAD; ++j) stage_piece(fs, j);
        }
  #pragma unroll
        for (int a = 0; a < TM; ++a)
  #pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
      stage_end();
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
#ifdef T2V_G2_TIMING
    if (lane == 0 && p.ws != nullptr && p.splitk == 1) {
      float* dst = p.ws + ((size_t)blockIdx.x * NW + wave) * 4;
      dst[0] = (float)tm_vm; dst[1] = (float)tm_bar; dst[2] = (float)tm_lds; dst[3] = (float)(clock64() - tm_begin);
    }
#endif
  }
#ifdef T2V_G2_EXPERIMENTS
#include "gemm2_experiments.inc"
#else
  static_assert(PP == 0, "experimental schedules: build with -DT2V_G2_EXPERIMENTS");
#endif
  wait_vmcnt<0>();   // drain the zero-page loads of the dead stages before the LDS goes away

  // ---- epilogue ---------------------------------------------------------------------------------
  if constexpr (TAT) {
    // Fused temporal self-attention (t2v_model.py:716-767 with CrossAttention :540-584): this tile holds q | k | v (64 channels
    // each: ONE head, the weight rows are packed head-major) of all F frames of p.tpix pixels.  The accumulators go to LDS as
    // fp16 (q | k row-major, V transposed: [pixel][d][frame slot]), then one wave per pixel runs the 32-key attention tile of
    // attention.hip (S^T = K Q^T with the key on the MFMA row axis, in-lane softmax, P from the accumulator registers) and
    // writes O for its F frames.  Q, K, V never reach HBM: 3 x the tensor written + read per attention before.
    static_assert(TM == 1 && TN == 3 && WM == 6 && WN == 2, "the fused attention epilogue is written for the 192x192 tile");
    constexpr int QK_PITCH = 272;                          // bytes per row: q (128 B) | k (128 B) + 16
    constexpr int VT_PITCH = 72;                           // bytes per V^T row: 32 frame slots x 2 B + 8
    __syncthreads();                                       // every wave is done reading the operand stages
    unsigned char* qk = smem;
    unsigned char* vt = smem + BM * QK_PITCH;
    const int F = p.F;
    {
      const int wrow = wm * 32 + (lane & 31);              // tile row of this lane's accumulator column
      const int pl = wrow / F, f = wrow - pl * F;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wn * (TN * 32) + b * 32 + 8 * q + 4 * (lane >> 5);
          const f16x4 v = {(f16)acc[0][b][4 * q], (f16)acc[0][b][4 * q + 1], (f16)acc[0][b][4 * q + 2], (f16)acc[0][b][4 * q + 3]};
          if (col < 128) {
            *reinterpret_cast<f16x4*>(qk + wrow * QK_PITCH + col * 2) = v;
          } else if (pl < p.tpix) {
            const int d = col - 128;
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<f16*>(vt + (pl * 64 + d + e) * VT_PITCH + f * 2) = v[e];
          }
        }
      // frame slots F .. 31 of V^T: zero (they meet probabilities that are exactly 0, but must not hold NaN patterns)
      for (int u = tid; u < p.tpix * 64; u += NW * 64)
        for (int k = F; k < 32; ++k) *reinterpret_cast<f16*>(vt + u * VT_PITCH + k * 2) = (f16)0.f;
    }
    __syncthreads();
    const int frow = lane & 31, fhalf = lane >> 5;
    const int smp = tile_m / p.tiles_ps;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int pl = wave; pl < p.tpix; pl += NW) {
      const int pix = (tile_m - smp * p.tiles_ps) * p.tpix + pl;
      if (pix >= p.HW) continue;                           // wave-uniform
      const unsigned char* rowp = qk + (pl * F + frow) * QK_PITCH;    // rows >= F of the pixel: next pixel / V^T bytes, finite, masked below
      f32x16 sc = zero16;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f16x8 qf = *reinterpret_cast<const f16x8*>(rowp + ((kk * 2 + fhalf) << 4));
        const f16x8 kf = *reinterpret_cast<const f16x8*>(rowp + 128 + ((kk * 2 + fhalf) << 4));
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, sc, 0, 0, 0);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        if (key >= F) sc[r] = -INFINITY;
        mx = fmaxf(mx, sc[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (frow >= F) mx = 0.f;                             // padded queries: any finite reference, the result is dropped
      const float neg_m = -mx * p.attn_scale_log2;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], p.attn_scale_log2, neg_m));
        sc[r] = pv;
        psum += pv;
      }
      psum += __shfl_xor(psum, 32);
      f32x16 oacc[2] = {zero16, zero16};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (f16)sc[8 * t + e];
        const int kofs = (t * 16 + 4 * fhalf) * 2;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const unsigned char* vrow = vt + (pl * 64 + d * 32 + frow) * VT_PITCH + kofs;
          const f16x4 lo = *reinterpret_cast<const f16x4*>(vrow);
          const f16x4 hi = *reinterpret_cast<const f16x4*>(vrow + 16);
          const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[d], 0, 0, 0);
        }
      }
      if (frow < F) {
        const float inv = 1.0f / psum;
        f16* orow = reinterpret_cast<f16*>(p.out) + (((long)smp * F + frow) * p.HW + pix) * p.ldc + tile_n * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (f16)(oacc[d][4 * qd + r] * inv);
            *reinterpret_cast<f16x4*>(orow + d * 32 + 8 * qd + 4 * fhalf) = o;
          }
      }
    }
    return;
  }
  if (p.epi == T2V_EPI_NONE) {      // row-coalesced through a per-wave LDS buffer (t2v_kernels.h); also the split-K slabs
    __builtin_amdgcn_s_barrier();   // every wave is done reading the operand stages
    if constexpr (XE == 2) {        // GroupNorm (+SiLU) of the result inside the epilogue: statistics meet at a grid barrier (t2v_kernels.h)
      t2v_epilogue_rows_gn<WM, WN, TM, TN>(p, acc, smem, lane, wave, m0, n0, tile_m, tile_n, tiles_m, tiles_n);
      return;
    }
    if constexpr (XE == 4) {        // to_q projection + text cross-attention: the accumulators are Q (T2V_EPI_XATTN, t2v_kernels.h)
      t2v_epilogue_xattn<WM, WN, TM, TN>(p, acc, smem, lane, wave, m0, n0);
      return;
    }
    if constexpr (XE == 3) {        // LayerNorm second output across the column tiles of the launch (partial row sums meet at the grid barrier)
      t2v_epilogue_rows_lnx<WM, WN, TM, TN>(p, acc, smem, lane, wave, m0, n0, tile_m, tile_n, tiles_n);
      return;
    }
    if constexpr (XE == 1) {        // whole rows in this tile (192x320, N == 320, validated by the executor): fused LayerNorm output
      float* fs = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {            // (TM == 2: the 256x320 tile, round 6 — a wave's two 32-row blocks one after the other)
        if (tm > 0) __syncthreads();                // the partner wave has read this wave's row sums of the previous block
        t2v_epilogue_rows_ln<TN>(p, reinterpret_cast<f32x16 (&)[1][TN]>(acc[tm]), fs + wave * (32 * T2V_EPI_SP), fs + NW * (32 * T2V_EPI_SP), lane, wave,
                                 m0 + (wm * TM + tm) * 32, n0 + wn * TN * 32);
      }
      return;
    }
    t2v_epilogue_rows<TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wave * (32 * T2V_EPI_SP), lane, m0 + wm * TM * 32,
                              n0 + wn * TN * 32, blockIdx.y, tile_m * tiles_n + tile_n);
    return;
  }
  const int mlane = lane & 31, nhalf = (lane >> 5) * 4;
  if (p.epi == T2V_EPI_GEGLU && p.splitk == 1) {
    // GEGLU through a per-wave LDS strip: a lane's (value + bias) * gelu(gate + bias) results are 4 channels of ONE row
    // (8-byte stores scattered over 32 rows); staged as [32 rows][TN * 16 channels] they leave as whole 16-byte chunks of
    // TN * 32-byte row segments (6 rows per store instruction at TN = 5).  Same-box A/B: the 32x32-level GEGLU GEMM
    // 153 -> 142 us, the 16x16-level one 116 -> 111 us.
    constexpr int PITCH = TN * 32 + 16;                   // bytes per staged row (+16: rows start on different banks)
    constexpr int CPRW = TN * 2;                          // 16-byte chunks per row
    __builtin_amdgcn_s_barrier();                         // every wave is done reading the operand stages
    unsigned char* strip = smem + wave * (32 * PITCH);
    const int nw = n0 + wn * TN * 32;                     // first packed column of this wave
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int mt = m0 + (wm * TM + a) * 32;
      if (mt >= p.M) continue;                            // wave-uniform
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int nt = nw + b * 32;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int n_val = nt + 16 * qq + nhalf;
          f16x4 o = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
          if (n_val < p.N) {
            float bv[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0};
            if (p.bias) {
              const f32x4 ba = *reinterpret_cast<const f32x4*>(p.bias + n_val);
              const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n_val + 8);
#pragma unroll
              for (int r = 0; r < 4; ++r) { bv[r] = ba[r]; bg[r] = bb[r]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (f16)((acc[a][b][8 * qq + r] + bv[r]) * t2v_gelu_erf(acc[a][b][8 * qq + 4 + r] + bg[r]));
          }
          *reinterpret_cast<f16x4*>(strip + mlane * PITCH + (b * 16 + 8 * qq + nhalf) * 2) = o;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      const int c_out0 = nw >> 1;                         // first output channel of the strip
#pragma unroll
      for (int u = lane; u < 32 * CPRW; u += 64) {
        const int row = u / CPRW, ch = u - row * CPRW;
        const int m = mt + row, c = c_out0 + ch * 8;
        if (m < p.M && c < (p.N >> 1)) {
          const f16x8 v = *reinterpret_cast<const f16x8*>(strip + row * PITCH + ch * 16);
          *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + (size_t)m * p.ldc + c) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int m = m0 + (wm * TM + a) * 32 + mlane;
    if (m >= p.M) continue;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int nt = n0 + (wn * TN + b) * 32;
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
            epi_store_geglu(p, m, n_val, (nt >> 1) + 8 * qq + nhalf, v, g);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nt + 8 * q + nhalf;
          if (n < p.N)
            epi_store(p, m, n, acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
      }
    }
  }
}

template <int WM, int WN, int TM, int TN, int BK, int STAGES, int MINW, int GATHER, int PP, int XE = 0, bool TAT = false>
hipError_t launch_cfg_gather(const GemmParams& p, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  // TAT: the attention epilogue re-uses the operand ring for q | k (BM x 272 B) and V^T (<= 12 pixels x 64 x 72 B)
  constexpr int ring = STAGES * (BM + BN) * BK * 2 + 1024;
  constexpr int gn_lds = t2v_gn_epilogue_lds(WM * WN, WM * TM, BN);
  constexpr int lnx_lds = t2v_lnx_epilogue_lds(WM * WN, WN, BM);
  constexpr int xa_lds = t2v_xattn_epilogue_lds(BM, BN);
  constexpr int lds = TAT ? (BM * 272 + 12 * 64 * 72 > ring ? BM * 272 + 12 * 64 * 72 : ring)
                          : (XE == 2 && gn_lds > ring ? gn_lds : (XE == 3 && lnx_lds > ring ? lnx_lds : (XE == 4 && xa_lds > ring ? xa_lds : ring)));
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles = ((p.M - p.m_begin + BM - 1) / BM) * tiles_n;
  auto k = gemm2_kernel<WM, WN, TM, TN, BK, STAGES, MINW, GATHER, PP, XE, TAT>;
  static t2v_device_flags attr_set;     // once per (instantiation, device): the call costs microseconds on the host
  {
    const hipError_t e = t2v_set_dynamic_lds(reinterpret_cast<const void*>(k), lds, attr_set, s);
    if (e != hipSuccess) return e;
  }
  if constexpr (XE == 2 || XE == 3) {
    // the epilogue's grid barrier needs every workgroup of the launch resident: no split-K, and the grid within what the occupancy
    // API grants this instantiation on the stream's device (cached); a process in which a barrier already timed out stays off it
    // (round 6: a grid larger than that is cut into row chunks of whole tiles AND whole statistics instances, one launch each)
    static int occ[T2V_MAX_DEVICES] = {};
    if (p.splitk != 1 || !t2v_coop_allowed()) return hipErrorCooperativeLaunchTooLarge;
    const long cap = t2v_grid_capacity(reinterpret_cast<const void*>(k), WM * WN * 64, lds, s, occ);
    return t2v_launch_coresident(p, BM, tiles_n, cap, XE == 2 ? t2v_lcm(BM, p.gn_rows) : BM, XE == 2 ? 2 * T2V_GN_PIECES * 16 : BM * 16,
                                 [&](const GemmParams& q, int nwg) {
      hipLaunchKernelGGL(k, dim3(nwg, 1), dim3(WM * WN * 64), lds, s, q);
      return hipGetLastError();
    });
  }
  hipLaunchKernelGGL(k, dim3(tiles, p.splitk > 1 ? p.splitk : 1), dim3(WM * WN * 64), lds, s, p);
  return hipGetLastError();
}

template <int WM, int WN, int TM, int TN, int BK, int STAGES, int MINW, int PP = 0>
hipError_t launch_cfg(const GemmParams& pin, hipStream_t s) {
  GemmParams p = pin;
  const int KT = (p.K + BK - 1) / BK;
  if (p.splitk > KT) p.splitk = KT;
  if (p.splitk < 1) p.splitk = 1;
  p.kt_per_split = (KT + p.splitk - 1) / p.splitk;
  p.splitk = (KT + p.kt_per_split - 1) / p.kt_per_split;  // no empty splits
  {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    p.panel = t2v_choose_panel(p, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    // in-kernel fold by the last-arriving workgroup of a tile (t2v_epilogue_rows) where a ticket buffer is given; else the reduction kernel
    if (p.splitk <= 1 || p.epi != T2V_EPI_NONE || tiles > T2V_SYNC_INTS || p.gn_out != nullptr) p.tickets = nullptr;
  }
  // tiles with a T2V_EPI_GN instantiation (validated by the executor): the whole-row tiles 8 / 11, and the 128-row tiles 3 / 5
  constexpr bool GN_TILE = PP != 1 && PP != 4 && (((WM == 6 || WM == 4) && WN == 2 && TM == 1 && TN == 5) || (WM == 2 && WN == 4 && TM == 2 && (TN == 1 || TN == 2)));
  // (with split-K the norm runs in the reduction's launch instead: any tile, t2v_launch_splitk_reduce_gn below)
  const bool gn_here = p.gn_out != nullptr && p.splitk == 1;
  if (gn_here && (!GN_TILE || (p.gather == T2V_GATHER_CONV3X3 && p.up))) return hipErrorInvalidValue;
  // ... and with a cross-tile LayerNorm instantiation: 5 (128x128), 12 (64x64), 9 (192x256), 3 (128x256); plain gather only
  constexpr bool LNX_TILE = PP != 1 && PP != 4 && ((WM == 2 && WN == 4 && TM == 2 && (TN == 1 || TN == 2)) || (WM == 2 && WN == 2 && TM == 1 && TN == 1) ||
                                    (WM == 6 && WN == 2 && TM == 1 && TN == 4));
  if (p.ln_x && (!LNX_TILE || p.gather != T2V_GATHER_PLAIN)) return hipErrorInvalidValue;
  hipError_t e;
#ifdef T2V_G2_DEV        // development build (seconds instead of minutes): the plain gather and the 3x3 convolution, no fused-norm instantiations
  if (p.gather == T2V_GATHER_PLAIN && p.epi != T2V_EPI_TATTN && p.xa_k == nullptr && p.ln_out == nullptr && !gn_here && !p.ln_x)
    e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP>(p, s);
  else if (p.gather == T2V_GATHER_CONV3X3 && !p.up && !gn_here)
    e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_CONV3X3, PP>(p, s);
  else
    return hipErrorInvalidValue;
  if (e != hipSuccess) return e;
  if (p.splitk > 1 && p.tickets == nullptr) e = p.gn_out != nullptr ? t2v_launch_splitk_reduce_gn(p, s) : t2v_launch_splitk_reduce(p, s);
  return e;
#else
  switch (p.gather) {
    case T2V_GATHER_PLAIN:
      if constexpr (WM == 6 && WN == 2 && TM == 1 && TN == 3 && PP == 0) {
        if (p.epi != T2V_EPI_TATTN || p.splitk != 1 || p.tpix < 1 || p.tpix > 12 || p.tpix * p.F > BM_OF(WM, TM) || p.F > 32) return hipErrorInvalidValue;
        e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP, 0, true>(p, s);
        break;
      }
      if (p.epi == T2V_EPI_TATTN) return hipErrorInvalidValue;
      if constexpr ((WM == 6 || WM == 4) && WN == 2 && TM == 1 && TN == 5 && PP == 0) {      // whole-row tiles: 192x320 / 128x320
        if (p.xa_k != nullptr) {      // fused to_q + text cross-attention (validated: N == 320 = 5 heads, fp16 out, no split-K)
          e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP, 4>(p, s);
          break;
        }
      }
      if constexpr (WM == 2 && WN == 4 && TM == 2 && TN == 1 && PP == 0) {                  // 128x128 on 8 waves (tile 5): two heads per column tile
        if (p.xa_k != nullptr) {
          e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP, 4>(p, s);
          break;
        }
      }
      if (p.xa_k != nullptr) return hipErrorInvalidValue;
      if constexpr ((WM == 6 || WM == 4) && WN == 2 && (TM == 1 || (WM == 4 && TM == 2)) && TN == 5 && PP != 1) {      // whole-row tiles: 192x320 / 128x320 / 256x320
        if (p.ln_out != nullptr) {
          e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP, 1>(p, s);
          break;
        }
      }
      if constexpr (GN_TILE) {
        if (gn_here) { e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP, 2>(p, s); break; }
      }
      if constexpr (LNX_TILE) {
        if (p.ln_x) { e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP, 3>(p, s); break; }
      }
      e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_PLAIN, PP>(p, s);
      break;
    case T2V_GATHER_CONV3X3:
      if constexpr (GN_TILE) {
        if (gn_here && !p.up) { e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_CONV3X3, PP, 2>(p, s); break; }
      }
      if (p.up) e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, G_CONV_UP, PP>(p, s);
      else e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_CONV3X3, PP>(p, s);
      break;
    case T2V_GATHER_TCONV3:
      if constexpr (GN_TILE) {
        if (gn_here) { e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_TCONV3, PP, 2>(p, s); break; }
      }
      e = launch_cfg_gather<WM, WN, TM, TN, BK, STAGES, MINW, T2V_GATHER_TCONV3, PP>(p, s);
      break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  // (split-K whose result feeds a fused GroupNorm: the reduction is the loader of a cooperative GroupNorm launch, norm.hip)
  if (p.splitk > 1 && p.tickets == nullptr) e = p.gn_out != nullptr ? t2v_launch_splitk_reduce_gn(p, s) : t2v_launch_splitk_reduce(p, s);
  return e;
#endif
}

}  // namespace

// tile ids (t2v_op.i[22]):  1 = 256x256, 2 = 256x320, 8 = 192x320 / 9 = 192x256 (12 waves), 11 = 128x320 (8 waves), 3 = 128x256 (8 waves), 4 / 5 = 128x128 with a 4-deep ring
// (few-row levels: latency-bound, keep 96 KiB per CU in flight) — 64-wide k-tiles
// (full 128-byte lines per row = one conv reduction chunk), 2-3 stage ring, one workgroup per CU.
hipError_t t2v_launch_gemm2(const GemmParams& p, int tile, hipStream_t s) {
#ifdef T2V_G2_DEVMIN
  switch (tile) {
    case 1: return launch_cfg<2, 4, 4, 2, 64, 2, 2>(p, s);
    case 22: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 4>(p, s);
    default: return hipErrorInvalidValue;
  }
#endif
#ifdef T2V_G2_DEV
  switch (tile) {
    case 1: return launch_cfg<2, 4, 4, 2, 64, 2, 2>(p, s);
    case 2: return launch_cfg<4, 2, 2, 5, 64, 2, 2>(p, s);
    case 3: return launch_cfg<2, 4, 2, 2, 64, 3, 2>(p, s);
    case 22: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 4>(p, s);
    case 23: return launch_cfg<4, 2, 2, 5, 64, 2, 2, 4>(p, s);
    case 24: return launch_cfg<2, 4, 2, 2, 64, 2, 2, 4>(p, s);
    default: return hipErrorInvalidValue;
  }
#endif
#ifdef T2V_G2_FEW        // experiment build (with -DT2V_G2_EXPERIMENTS): only the tiles under study (compile time)
  switch (tile) {
    case 1: return launch_cfg<2, 4, 4, 2, 64, 2, 2>(p, s);
    case 2: return launch_cfg<4, 2, 2, 5, 64, 2, 2>(p, s);
    case 3: return launch_cfg<2, 4, 2, 2, 64, 3, 2>(p, s);
    case 8: return launch_cfg<6, 2, 1, 5, 64, 2, 3>(p, s);
    case 13: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 2>(p, s);
    case 14: return launch_cfg<4, 2, 2, 5, 64, 2, 2, 2>(p, s);
    case 15: return launch_cfg<2, 4, 2, 2, 64, 3, 2, 2>(p, s);
    case 17: return launch_cfg<6, 2, 1, 5, 64, 2, 3, 2>(p, s);
    case 18: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 3>(p, s);
    case 19: return launch_cfg<4, 2, 2, 5, 64, 2, 2, 3>(p, s);
    case 20: return launch_cfg<6, 2, 1, 5, 64, 2, 3, 3>(p, s);
    case 21: return launch_cfg<2, 4, 2, 2, 64, 2, 2, 3>(p, s);
    default: return hipErrorInvalidValue;
  }
#endif
  switch (tile) {
    case 1: return launch_cfg<2, 4, 4, 2, 64, 2, 2>(p, s);   // 2 x 64 KiB
    case 2: return launch_cfg<4, 2, 2, 5, 64, 2, 2>(p, s);   // 2 x 72 KiB
    case 3: return launch_cfg<2, 4, 2, 2, 64, 3, 2>(p, s);   // 3 x 48 KiB
    case 4: return launch_cfg<2, 2, 2, 2, 64, 4, 1>(p, s);   // 128x128, 4 waves, 4 x 32 KiB: 3 k-tiles in flight
    case 5: return launch_cfg<2, 4, 2, 1, 64, 4, 2>(p, s);   // 128x128, 8 waves, 4 x 32 KiB
    case 8: return launch_cfg<6, 2, 1, 5, 64, 2, 3>(p, s);   // 192x320, 12 waves (3 per SIMD), 2 x 64 KiB: M = 49152 -> exactly 256 workgroups
    case 9: return launch_cfg<6, 2, 1, 4, 64, 2, 3>(p, s);   // 192x256, 12 waves, 2 x 56 KiB (N = 256 * j where 256-row grids fill badly)
    case 10: return launch_cfg<6, 2, 1, 3, 64, 2, 3>(p, s);  // 192x192, 12 waves: fused QKV projection + temporal attention (T2V_EPI_TATTN only)
    case 12: return launch_cfg<2, 2, 1, 1, 64, 4, 2>(p, s);  // 64x64, 4 waves, 4 x 16 KiB (2 workgroups per CU): the 4x4 level (M = 768) as 240 tiles with the
                                                             // FULL reduction each — no split-K slabs, no reduction launch (experiment, round 4)
    case 11: return launch_cfg<4, 2, 1, 5, 64, 2, 2>(p, s);  // 128x320, 8 waves (2 per SIMD), 2 x 56 KiB: M = 32768 (VideoCrafter, 16 frames) -> exactly
                                                             // 256 workgroups where 192-row tiles make 171; also the b = 1 per-GPU shapes (M = 24576 -> 192)
#ifdef T2V_G2_EXPERIMENTS   // round-5 schedule experiments, measured and NOT selected (tools/gemm_pf_probe.py): instantiated on request only
    // 13 .. 17 = 1, 2, 3, 5, 8 with the next k-tile's first fragments prefetched across the barrier: +-0 (-1.5 % .. +3.7 %)
    case 13: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 2>(p, s);
    case 14: return launch_cfg<4, 2, 2, 5, 64, 2, 2, 2>(p, s);
    case 15: return launch_cfg<2, 4, 2, 2, 64, 3, 2, 2>(p, s);
    case 16: return launch_cfg<2, 4, 2, 1, 64, 4, 2, 2>(p, s);
    case 17: return launch_cfg<6, 2, 1, 5, 64, 2, 3, 2>(p, s);
    // 18 .. 21 = 1, 2, 8 and the 128x256 tile with the operands staged through registers instead of LDS-DMA: +4 % on 8192^3, -1 .. -8 % on the UNet's shapes
    case 18: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 3>(p, s);
    case 19: return launch_cfg<4, 2, 2, 5, 64, 2, 2, 3>(p, s);
    case 20: return launch_cfg<6, 2, 1, 5, 64, 2, 3, 3>(p, s);
    case 21: return launch_cfg<2, 4, 2, 2, 64, 2, 2, 3>(p, s);
#endif
#ifdef T2V_G2_EXPERIMENTS
    // 22 / 23 / 24 = 256x256 / 256x320 / 128x256 on the round-6 schedule: two wave groups one barrier apart, region staging two k-tiles
    // ahead, counted vmcnt (PP == 4).  Correct, and measured equal to lock-step (+-3 % long K, -7 .. -20 % short K): the calibration GEMM
    // spends the same cycles at 62 % MFMA busy under either schedule and the chip clocks 1.55-1.68 GHz on random operands (2.3 GHz on
    // zeros) — power, not the issue schedule, is the ceiling (profiles/r06_gemm_mainloop_findings.txt)
    case 22: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 4>(p, s);
    case 23:
      if (p.gather == T2V_GATHER_CONV3X3 && p.up) return launch_cfg<4, 2, 2, 5, 64, 2, 2>(p, s);   // (upsample gather: register budget)
      return launch_cfg<4, 2, 2, 5, 64, 2, 2, 4>(p, s);
    case 24: return launch_cfg<2, 4, 2, 2, 64, 2, 2, 4>(p, s);
    // 6 / 7 = 256x256 / 256x320 on the round-1 ping-pong (the next k-tile's DMA drained with vmcnt(0)): equal to lock-step
    case 6: return launch_cfg<2, 4, 4, 2, 64, 2, 2, 1>(p, s);
    case 7:
      if (p.gather == T2V_GATHER_CONV3X3 && p.up) return launch_cfg<4, 2, 2, 5, 64, 2, 2>(p, s);   // (upsample gather: register budget)
      return launch_cfg<4, 2, 2, 5, 64, 2, 2, 1>(p, s);
#endif
    default: return hipErrorInvalidValue;
  }
}
