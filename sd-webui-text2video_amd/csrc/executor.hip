// C-ABI of libt2v_hip.so (see include/t2v_hip.h): validates a denoise program and executes it
// by launching the HIP kernels of this directory on the caller's stream.  No torch types, no
// allocation on the data path, no CPU fallback: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "t2v_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

struct Resolved {
  t2v_op op;
};

inline bool resolve_ptrs(t2v_op& op, const uint64_t* ext, int n_ext) {
  for (int k = 0; k < T2V_OP_NP; ++k) {
    const uint64_t v = op.p[k];
    if (v != 0 && v < T2V_EXT_SLOTS) {
      if ((int)v >= n_ext || ext == nullptr) return false;
      op.p[k] = ext[v];
    }
  }
  return true;
}

int validate_op(const t2v_op& op, int idx) {
  char buf[160];
  auto bad = [&](const char* why) {
    snprintf(buf, sizeof buf, "op %d (kind %d, tag %d): %s", idx, op.kind, op.tag, why);
    return fail(T2V_ERR_BAD_ARG, buf);
  };
  switch (op.kind) {
    case T2V_OP_GEMM: {
      const int M = op.i[0], N = op.i[1], K = op.i[2], g = op.i[7];
      if (M <= 0 || N <= 0 || K <= 0) return bad("empty GEMM");
      if (N % 4 != 0) return bad("N must be a multiple of 4");
      if (K % 8 != 0 || op.i[3] % 8 != 0 || op.i[4] % 8 != 0) return bad("K, lda, ldw must be multiples of 8");
      if (op.i[5] % 4 != 0) return bad("ldc must be a multiple of 4");
      if (g == T2V_GATHER_CONV3X3 || g == T2V_GATHER_TCONV3) {
        if (op.i[10] % 64 != 0) return bad("conv Cin must be a multiple of 64");
        if (K != (g == T2V_GATHER_CONV3X3 ? 9 : 3) * op.i[10]) return bad("K != taps*Cin");
      } else if (g == T2V_GATHER_CONV3X3_C8) {
        if (op.i[10] != 8 || K != 72 || op.i[3] != 8) return bad("C8 conv needs Cin == lda == 8, K == 72");
      } else if (g != T2V_GATHER_PLAIN) {
        return bad("unknown gather mode");
      }
      if (g == T2V_GATHER_CONV3X3 || g == T2V_GATHER_CONV3X3_C8) {
        if (op.i[11] != 1 && op.i[11] != 2) return bad("stride must be 1 or 2");
        if (op.i[12] != 0 && op.i[12] != 1) return bad("upsample must be 0 or 1");
        if (op.i[13] <= 0 || op.i[14] <= 0 || M % (op.i[13] * op.i[14]) != 0) return bad("M not a multiple of Hout*Wout");
      }
      if (g == T2V_GATHER_TCONV3 && (op.i[8] <= 0 || op.i[9] <= 0 || M % (op.i[8] * op.i[9]) != 0))
        return bad("M not a multiple of F*HW");
      if (op.i[16] == T2V_EPI_GEGLU && (N % 32 != 0 || op.i[17] != T2V_F16)) return bad("GEGLU needs N % 32 == 0, fp16 out");
      if (op.i[16] == T2V_EPI_STATS && (op.p[7] == 0 || op.i[19] > 1 || (g == T2V_GATHER_PLAIN && op.i[8] == 1)))
        return bad("column statistics (T2V_EPI_STATS): strips pointer p[7], no split-K, no fused LayerNorm");
      if (op.i[16] < 0 || op.i[16] > T2V_EPI_XATTN) return bad("unknown epilogue");
      if (op.i[16] == T2V_EPI_XATTN) {
        const int tile = op.i[22];
        if (g != T2V_GATHER_PLAIN || N % 64 != 0 || K % 64 != 0 || op.i[17] != T2V_F16 || op.i[19] > 1 || op.i[18] != 0 || op.i[8] != 0 || op.i[11] == 1)
          return bad("fused cross-attention: plain gather, N = 64 * heads, K % 64 == 0, fp16 out, no split-K / activation / LayerNorm / hi + lo output");
        if (!(((tile == 8 || tile == 11) && N == 320) || ((tile == 0 || tile == 5) && N % 128 == 0))) return bad("fused cross-attention: tile 8 / 11 with N == 320, or tile 0 / 5 with N % 128 == 0");
        if (op.p[2] != 0 || op.p[3] != 0 || op.p[4] != 0 || op.p[8] == 0 || op.p[9] == 0 || !(op.f[1] > 0.f)) return bad("fused cross-attention: no bias / row bias / residual; K, V^T and a positive scale are required");
        if (op.i[25] < 1 || op.i[25] > 96 || op.i[24] < N || op.i[24] % 8 != 0 || op.i[26] < ((op.i[25] + 31) / 32) * 32 || op.i[26] % 8 != 0 || op.i[15] <= 0 || op.i[15] % 32 != 0 || op.i[27] < 0 || op.i[28] < 0)
          return bad("fused cross-attention: 1 .. 96 keys, ld(K) >= N, V^T rows of >= ceil32(keys) halfs, rows per sample a multiple of 32");
        if (op.i[5] < N || op.i[5] % 4 != 0) return bad("fused cross-attention: ldc");
        if (op.i[12] != 0) return bad("fused cross-attention: no residual wrap");
        return 0;
      }
      if (op.i[16] == T2V_EPI_GN) {
        // GroupNorm (+SiLU) of the result inside the epilogue: the tiles with an instantiation, whole 32-row strips per statistics
        // instance, at most two instances per row tile, whole groups, every pointer of the exchange
        const int tile = op.i[22], rows = op.i[24], groups = op.i[28];
        const int bm = tile == 8 ? 192 : 128, bn = (tile == 8 || tile == 11) ? 320 : (tile == 3 ? 256 : 128);
        if (g == T2V_GATHER_CONV3X3_C8 || (g == T2V_GATHER_CONV3X3 && op.i[12] != 0)) return bad("fused GroupNorm: not for the C8 stem / the upsampling gather");
        if (op.i[20] != 0 || K % 64 != 0 || op.i[18] != 0) return bad("fused GroupNorm: no bias along M, no activation, K % 64 == 0");
        if (g == T2V_GATHER_PLAIN && (op.i[8] != 0 || op.i[11] == 1)) return bad("fused GroupNorm: no fused LayerNorm / hi + lo output on the same op");
        if (groups <= 0 || N % groups != 0 || rows <= 0 || M % rows != 0) return bad("fused GroupNorm: N % groups == 0, rows per instance must divide M");
        if (op.i[19] > 1) {
          // split-K: the norm runs in the reduction's launch (one thread per row x 8 channels, any tile)
          if (groups > 32 || N % 8 != 0 || N / 8 > 512 || op.p[6] == 0) return bad("fused GroupNorm on a split-K GEMM: groups <= 32, N % 8 == 0, N <= 4096, a slab workspace");
        } else {
          if (tile != 8 && tile != 11 && tile != 3 && tile != 5 && tile != 0) return bad("fused GroupNorm: tile must be 8, 11, 3, 5 or 0");
          if (tile == 0 && N % 128 != 0) return bad("fused GroupNorm on the 128x128-class kernel: N % 128 == 0");
          if (N / groups > bn || (bn + N / groups - 1) / (N / groups) + 1 > T2V_GN_PIECES) return bad("fused GroupNorm: a group no wider than the tile");
          if (rows % 32 != 0 || !(rows >= bm || 2 * rows == bm))
            return bad("fused GroupNorm: rows per instance must be a multiple of 32 and >= the tile's rows (or exactly half of them): at most two instances per row tile");
        }
        if (op.p[8] == 0 || op.p[9] == 0 || op.p[10] == 0 || op.p[11] == 0) return bad("fused GroupNorm: gamma|beta, output, scratch and barrier words are required");
        if (op.i[25] < N * (op.i[27] ? 2 : 1) || op.i[25] % 4 != 0) return bad("fused GroupNorm: leading dimension of the normalised output");
        if (op.i[29] == 0 && op.p[5] == 0) return bad("fused GroupNorm: `out` is null but i[29] asks for it");
        if (op.i[5] < N || (op.p[4] != 0 && (op.i[6] < N || op.i[6] % 4 != 0))) return bad("fused GroupNorm: ldc / ldr must be >= N and multiples of 4");
      }
      if (op.i[19] > 1 && op.p[6] == 0) return bad("split-K without workspace");
      if (op.p[3] != 0 && op.i[15] <= 0 && !(g == T2V_GATHER_PLAIN && (op.i[8] == 1 || op.i[8] == 2))) return bad("rowbias without rows_per_batch");
      if (op.i[22] < 0 || op.i[22] > 24) return bad("unknown tile id");
      if ((op.i[16] == T2V_EPI_TATTN) != (op.i[22] == 10)) return bad("tile 10 is the fused QKV + temporal attention tile (T2V_EPI_TATTN), and only that");
      if (op.i[16] == T2V_EPI_TATTN) {
        const int F = op.i[8], HW = op.i[9], tpix = op.i[10];
        if (g != T2V_GATHER_PLAIN || N % 192 != 0 || K % 64 != 0 || op.i[17] != T2V_F16 || op.i[19] > 1 || op.i[18] != 0)
          return bad("fused temporal attention: plain gather, N = 192 * heads, K % 64 == 0, fp16 out, no split-K / activation");
        if (F < 2 || F > 32 || HW < 1 || tpix < 1 || tpix > 12 || tpix * F > 192) return bad("fused temporal attention: 2 <= F <= 32, 1 <= pixels per tile <= 12, pixels * F <= 192");
        const int tiles_ps = (HW + tpix - 1) / tpix;
        if (M % 192 != 0 || (M / 192) % tiles_ps != 0) return bad("fused temporal attention: M must be samples * ceil(HW / pixels per tile) * 192");
        if (op.p[2] != 0 || op.p[3] != 0 || op.p[4] != 0 || op.p[0] == 0 || op.p[1] == 0 || op.p[5] == 0 || !(op.f[1] > 0.f))
          return bad("fused temporal attention: no bias / row bias / residual; A, W, out and a positive scale are required");
        if (op.i[5] < (N / 192) * 64 || op.i[5] % 4 != 0) return bad("fused temporal attention: ldc < heads * 64");
        return 0;
      }
      if (g == T2V_GATHER_PLAIN && (op.i[12] != 0 || op.i[13] != 0)) {       // row wrap of the residual / the A operand
        if (op.i[12] < 0 || op.i[13] < 0 || op.i[19] > 1 || op.i[16] == T2V_EPI_GEGLU) return bad("row wrap: non-negative, no split-K, plain epilogue");
        if ((op.i[12] != 0 && (M > 2 * op.i[12] || op.p[4] == 0)) || (op.i[13] != 0 && M > 2 * op.i[13])) return bad("row wrap: M <= 2 * wrap (and a residual for the residual wrap)");
      }
      if (g == T2V_GATHER_PLAIN && op.i[11] == 1) {       // hi + lo fp16 output
        if (op.i[17] != T2V_F16 || op.i[16] != T2V_EPI_NONE || op.i[8] == 1 || op.i[5] < 2 * N)
          return bad("hi + lo output: fp16 out, plain epilogue, no fused LayerNorm, ldc >= 2 N");
      }
      if (g == T2V_GATHER_PLAIN && op.i[8] == 2) {         // LayerNorm second output across the column tiles (grid barrier)
        const int tile = op.i[22];
        if ((tile != 0 && tile != 5 && tile != 12 && tile != 9 && tile != 3) || (tile == 0 && N % 128 != 0) || op.i[19] > 1 || op.i[16] != T2V_EPI_NONE ||
            op.i[17] != T2V_F32 || op.i[18] != 0 || op.i[20] != 0 || K % 64 != 0 || op.i[11] == 1)
          return bad("cross-tile LayerNorm output: tile 0 (N % 128 == 0), 3, 5, 9 or 12, fp32 out, plain epilogue, no split-K / activation");
        if (op.p[3] == 0 || op.p[7] == 0 || op.p[10] == 0 || op.p[11] == 0 || op.i[9] < N || op.i[9] % 4 != 0)
          return bad("cross-tile LayerNorm output: gamma|beta, output, scratch, barrier words or leading dimension");
        if (op.i[5] < N || (op.p[4] != 0 && (op.i[6] < N || op.i[6] % 4 != 0))) return bad("cross-tile LayerNorm output: ldc / ldr must be >= N and multiples of 4");
      }
      if (g == T2V_GATHER_PLAIN && op.i[8] == 1) {
        if ((op.i[22] != 8 && op.i[22] != 11 && op.i[22] != 2) || N != 320 || op.i[19] > 1 || op.i[16] != T2V_EPI_NONE || op.i[17] != T2V_F32 || op.i[18] != 0 || op.i[20] != 0 ||
            K % 64 != 0)
          return bad("fused LayerNorm output needs a whole-row tile (192x320, 128x320 or 256x320), N == 320, fp32 out, no split-K / activation");
        if (op.p[3] == 0 || op.p[7] == 0 || op.i[9] < N || op.i[9] % 4 != 0) return bad("fused LayerNorm output: gamma|beta, output pointer or leading dimension");
        // the epilogue moves whole f32x4 / f16x4 groups without tail guards
        if (op.i[5] < N || (op.p[4] != 0 && (op.i[6] < N || op.i[6] % 4 != 0))) return bad("fused LayerNorm output: ldc / ldr must be >= N and multiples of 4");
      }
      return 0;
    }
    case T2V_OP_GROUPNORM: {
      const int C = op.i[2], groups = op.i[4], phase = op.i[8], nparts = op.i[9] > 0 ? op.i[9] : 1;
      if (op.i[0] <= 0 || op.i[1] <= 0 || C <= 0 || groups <= 0) return bad("empty GroupNorm");
      if (C % groups != 0 || C % 8 != 0 || op.i[3] % 8 != 0 || op.i[7] % 8 != 0 || groups > 256) return bad("GroupNorm needs C % groups == 0, C / ld % 8 == 0, groups <= 256");
      if (phase < 0 || phase > 3 || op.i[10] < 0 || op.i[10] >= nparts) return bad("bad GroupNorm phase / part");
      if (phase == 3 && (op.p[6] == 0 || op.i[1] % 32 != 0 || op.i[17] < C || nparts != 1)) return bad("GroupNorm phase 3: producer strips p[6], rows % 32 == 0, strip row length i[17] >= C");
      if (op.i[13] != 0 && op.i[13] < op.i[1]) return bad("rows of the largest part < rows");
      if (op.i[12] != 0 && (phase != 0 || (C / groups) % 4 != 0)) return bad("single-launch GroupNorm: phase 0, (C/groups) % 4 == 0");
      if (op.i[16] != 0 && (phase == 1 || op.i[7] < 2 * C)) return bad("GroupNorm low-order output: not for the statistics-only phase, ld_out >= 2 C");
      if (op.p[8] != 0 && (phase == 1 || op.i[19] % 8 != 0 || op.i[20] % 8 != 0 || op.i[19] < C + (op.i[20] ? C : 0) || (op.i[20] != 0 && op.i[20] < C)))
        return bad("GroupNorm cast output: not for the statistics-only phase; leading dimension >= C (+ C for the low-order image), multiples of 8");
      if (op.p[0] == 0 || op.p[1] == 0 || op.p[2] == 0 || op.p[4] == 0 || (phase != 1 && op.p[3] == 0)) return bad("null GroupNorm pointer");
      return 0;
    }
    case T2V_OP_LAYERNORM:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[1] % 4 != 0 || op.i[1] > 64 * 4 * 8) return bad("LayerNorm needs 0 < C <= 2048, C % 4 == 0");
      if (op.i[2] < op.i[1] || op.i[3] < op.i[1]) return bad("LayerNorm leading dimension < C");
      if (op.p[0] == 0 || op.p[1] == 0 || op.p[2] == 0 || op.p[3] == 0) return bad("null LayerNorm pointer");
      return 0;
    case T2V_OP_ATTENTION: {
      const int d = op.i[14] > 0 ? op.i[14] : 64;
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[2] <= 0 || op.i[3] <= 0 || op.i[4] <= 0) return bad("empty attention");
      if (d != 40 && d != 64 && d != 80 && d != 160) return bad("attention head_dim must be 40, 64, 80 or 160");
      for (int k = 5; k <= 13; ++k)
        if (op.i[k] < 0) return bad("negative attention stride");
      if (op.i[15] != 0 && op.i[0] != op.i[1]) return bad("causal attention needs nq == nk");
      if (op.i[16] < 0) return bad("negative low-order output offset");
      if (!(op.f[0] > 0.f)) return bad("attention scale must be > 0");
      if (op.p[0] == 0 || op.p[1] == 0 || op.p[2] == 0 || op.p[3] == 0) return bad("null attention pointer");
      if (op.p[6] != 0 && (d != 64 || op.i[15] != 0 || op.i[17] < op.i[1] || op.i[17] % 64 != 0 || (op.i[18] != 0 && op.i[18] != 4 && op.i[18] != 8)))
        return bad("attention with a V^T scratch (p[6]): head_dim 64, not causal, i[17] = keys padded to a multiple of 64, i[18] = 0 | 4 | 8 waves");
      return 0;
    }
    case T2V_OP_RELPOS_ATTN:
      if (op.i[1] <= 0 || op.i[1] > 32 || op.i[0] <= 0 || op.i[16] < 0 || op.i[16] + op.i[0] > op.i[1])
        return bad("relative-position attention needs nk <= 32 and queries [q_off, q_off + nq) inside the keys");
      if (op.i[14] <= 0 || op.i[14] % 8 != 0 || op.i[14] > 160) return bad("relative-position attention head_dim: multiple of 8, <= 160");
      if (op.i[2] <= 0 || op.i[3] <= 0 || op.i[4] <= 0 || op.i[15] < 0 || op.i[18] < 0) return bad("empty relative-position attention");
      for (int k = 5; k <= 13; ++k)
        if (op.i[k] < 0) return bad("negative attention stride");
      for (int k = 0; k < 6; ++k)
        if (op.p[k] == 0) return bad("null relative-position attention pointer");
      if (op.i[17] < 0 || op.i[17] > 2) return bad("relative-position attention kernel selector i[17]: 0 | 1 | 2");
      if (op.i[17] == 2 && (op.p[6] == 0 || op.p[7] == 0)) return bad("relative-position attention i[17] = 2 needs the packed fp16 tables p[6], p[7]");
      return 0;
    case T2V_OP_SOFTMAX:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[2] < op.i[1] || op.i[3] < op.i[1] || op.p[0] == 0 || op.p[1] == 0) return bad("bad softmax shape / pointer");
      return 0;
    case T2V_OP_NCTHW_TO_CL:
    case T2V_OP_CL_TO_NCTHW:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[2] <= 0 || op.i[3] <= 0 || op.i[4] < op.i[1]) return bad("bad layout-conversion shape");
      if (op.p[0] == 0 || op.p[1] == 0) return bad("null layout-conversion pointer");
      return 0;
    case T2V_OP_TIME_EMBED:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[1] % 2 != 0 || op.p[0] == 0 || op.p[1] == 0 || op.p[2] == 0) return bad("bad time-embedding op");
      return 0;
    case T2V_OP_COPY2D:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[1] % 4 != 0) return bad("copy2d needs cols % 4 == 0");
      if (op.i[2] < op.i[1] || op.i[3] < op.i[1] || op.i[6] < 0 || op.i[6] > 3) return bad("copy2d leading dimension / activation");
      if (op.p[0] == 0 || op.p[1] == 0) return bad("null copy2d pointer");
      if (op.p[2] != 0 && !(op.i[4] == T2V_F32 && op.i[5] == T2V_F16)) return bad("copy2d: the low-order output exists for fp32 -> fp16 casts only");
      return 0;
    case T2V_OP_DDIM_STEP: {
      const int C = op.i[0], cps = op.i[6] > 0 ? op.i[6] : C;
      if (C <= 0 || op.i[1] <= 0 || cps <= 0 || C % cps != 0) return bad("DDIM step needs C > 0, inner > 0, C % channels-per-sample == 0");
      if (op.i[2] < 0 || op.i[2] > cps || (op.i[5] != 0 && op.i[5] != 1)) return bad("DDIM step: guided channels / mode");
      if (op.p[0] == 0 || op.p[1] == 0 || op.p[3] == 0) return bad("null DDIM step pointer");
      return 0;
    }
    case T2V_OP_MEMSET:
      if (op.p[0] == 0) return bad("null memset pointer");
      return 0;
    case T2V_OP_LINCOMB:
      if (op.i[0] <= 0 || op.i[1] < 1 || op.i[1] > 6 || op.p[6] == 0) return bad("lincomb needs n > 0, 1..6 terms, an output");
      for (int k = 0; k < op.i[1]; ++k)
        if (op.p[k] == 0) return bad("null lincomb term");
      return 0;
    case T2V_OP_EMBED_ROWS:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[2] <= 0 || op.i[3] <= 0) return bad("empty embedding lookup");
      for (int k = 0; k < 4; ++k)
        if (op.p[k] == 0) return bad("null embedding pointer");
      return 0;
    case T2V_OP_TO_UINT8:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[1] > 4 || op.i[2] <= 0 || op.i[3] <= 0 || op.i[4] <= 0) return bad("empty uint8 conversion");
      if (op.p[0] == 0 || op.p[1] == 0) return bad("null uint8-conversion pointer");
      return 0;
    case T2V_OP_ALLGATHER:
      if (op.i[2] < 1 || op.i[3] < 0 || op.i[3] >= op.i[2] || op.p[0] == 0) return bad("bad all-gather record");
      return 0;
    case T2V_OP_HALO_EXCHANGE:
      if (op.i[2] < 1 || op.p[0] == 0 || op.i[3] < -1 || op.i[4] < -1) return bad("bad halo-exchange record");
      return 0;
    case T2V_OP_STATS_HALO:
      if (op.i[2] < 1 || op.i[3] < 0 || op.i[3] >= op.i[2] || op.p[0] == 0 || op.p[1] == 0) return bad("bad statistics + halo record");
      if (op.i[6] < 1 || op.i[7] < -1 || op.i[8] < -1 || op.i[7] >= op.i[2] || op.i[8] >= op.i[2]) return bad("bad statistics + halo neighbours");
      return 0;
    case T2V_OP_RESHARD_ROWS:
      if (op.i[0] <= 0 || op.i[1] <= 0 || op.i[2] <= 0 || op.i[3] < 0 || op.i[4] < 0) return bad("bad row-resharding shape");
      if (op.i[5] < op.i[1] || op.i[6] < op.i[1] || op.p[0] == 0 || op.p[1] == 0) return bad("row resharding: leading dimension / pointer");
      if (op.p[2] != 0 && op.i[7] != T2V_F32) return bad("row resharding: the residual form is fp32");
      if (op.i[9] > 1 && (op.i[9] > 64 || op.i[10] < 0 || op.i[11] < 0 || op.i[12] < 0 || op.i[13] < -1 || op.i[13] >= op.i[9] ||
                          (op.i[14] != 0 && op.i[14] != 1) || (op.i[13] >= 0 && op.p[3] == 0)))
        return bad("row resharding: bad multi-part record (parts, part strides, own part)");
      return 0;
    case T2V_OP_ALLTOALL:
      if (op.i[2] < 1 || op.i[3] < 0 || op.i[3] >= op.i[2] || op.i[4] < 1 || op.i[5] < 1 || op.i[5] > op.i[4]) return bad("bad all-to-all record");
      if ((op.i[6] != 0 && op.i[6] != 1) || op.p[0] == 0 || op.p[1] == 0) return bad("bad all-to-all direction / pointer");
      return 0;
    default:
      return bad("unknown op kind");
  }
}

hipError_t launch_op(const t2v_op& op, hipStream_t s) {
  switch (op.kind) {
    case T2V_OP_GEMM: {
      GemmParams p;
      memset(&p, 0, sizeof p);
      p.M = op.i[0]; p.N = op.i[1]; p.K = op.i[2];
      p.lda = op.i[3]; p.ldw = op.i[4]; p.ldc = op.i[5]; p.ldr = op.i[6];
      p.gather = op.i[7];
      if (p.gather == T2V_GATHER_TCONV3) { p.F = op.i[8]; p.HW = op.i[9]; }
      else { p.Hin = op.i[8]; p.Win = op.i[9]; }
      p.Cin = op.i[10]; p.stride = op.i[11]; p.up = op.i[12]; p.Hout = op.i[13]; p.Wout = op.i[14];
      p.rows_per_batch = op.i[15] > 0 ? op.i[15] : 1;
      p.epi = op.i[16]; p.out_f32 = op.i[17] == T2V_F32; p.act = op.i[18];
      if (p.epi == T2V_EPI_STATS) { p.epi = T2V_EPI_NONE; p.stats = reinterpret_cast<float*>(op.p[7]); }   // (validated: no split-K -> p[7] is not a ticket buffer)
      if (p.epi == T2V_EPI_GN) {                                  // GroupNorm (+SiLU) of the result inside the epilogue (validated above)
        p.epi = T2V_EPI_NONE;
        p.gn_gb = reinterpret_cast<const float*>(op.p[8]);
        p.gn_out = reinterpret_cast<f16*>(op.p[9]);
        p.gn_part = reinterpret_cast<double*>(op.p[10]);
        p.gn_bar = reinterpret_cast<unsigned*>(op.p[11]);
        p.gn_fault = t2v_coop_fault_word();
        p.gn_rows = op.i[24]; p.ld_gn = op.i[25]; p.gn_silu = op.i[26]; p.gn_lo = op.i[27] ? op.i[1] : 0;
        p.gn_cpg = op.i[1] / op.i[28]; p.gn_store_out = op.i[29] ? 0 : 1;
        p.gn_eps = op.f[2];
        t2v_exchange_ids(&p.gn_seq, &p.gn_want);
      }
      p.splitk = op.i[19] > 1 ? op.i[19] : 1;
      p.bias_m = op.i[20]; p.ldrb = op.i[21];
      p.A = reinterpret_cast<const f16*>(op.p[0]);
      p.W = reinterpret_cast<const f16*>(op.p[1]);
      p.bias = reinterpret_cast<const float*>(op.p[2]);
      p.rowbias = reinterpret_cast<const float*>(op.p[3]);
      p.res = reinterpret_cast<const float*>(op.p[4]);
      p.out = reinterpret_cast<void*>(op.p[5]);
      p.ws = reinterpret_cast<float*>(op.p[6]);
      p.halo = op.i[23];
      p.out_lo = (p.gather == T2V_GATHER_PLAIN && op.i[11] == 1) ? 1 : 0;
      if (p.gather == T2V_GATHER_PLAIN && op.i[16] != T2V_EPI_TATTN) { p.res_wrap = op.i[12]; p.a_wrap = op.i[13]; }
      const int tile = op.i[22];
      if (p.epi == T2V_EPI_TATTN) {                              // fused QKV projection + temporal attention (tile 10)
        p.F = op.i[8]; p.HW = op.i[9]; p.tpix = op.i[10];
        p.tiles_ps = (p.HW + p.tpix - 1) / p.tpix;
        p.attn_scale_log2 = op.f[1] * 1.44269504088896340736f;
        p.Hin = p.Win = 0; p.Cin = 0;
        return t2v_launch_gemm2(p, tile, s);
      }
      // split-K: p[7] = T2V_SYNC_INTS zeroed ints -> the fold runs in the GEMM's last-arriving workgroups; 0 -> reduction kernel
      if (p.epi == T2V_EPI_XATTN) {                              // fused to_q + text cross-attention (validated above)
        p.epi = T2V_EPI_NONE;
        p.xa_k = reinterpret_cast<const f16*>(op.p[8]);
        p.xa_vt = reinterpret_cast<const f16*>(op.p[9]);
        p.xa_ldk = op.i[24]; p.xa_lc = op.i[25]; p.xa_lcp = op.i[26]; p.xa_k_sample = op.i[27]; p.xa_vt_sample = op.i[28];
        p.attn_scale_log2 = op.f[1] * 1.44269504088896340736f;
      }
      const bool ln_any = p.gather == T2V_GATHER_PLAIN && (op.i[8] == 1 || op.i[8] == 2);
      if (p.splitk > 1 && !ln_any) p.tickets = reinterpret_cast<int*>(op.p[7]);
      if (ln_any) {       // fused LayerNorm second output (validated: tile 8 / 11, N == 320 — or, i[8] == 2, across column tiles; TATTN returned above)
        p.ln_gb = reinterpret_cast<const float*>(op.p[3]);
        p.rowbias = nullptr;
        p.ln_out = reinterpret_cast<f16*>(op.p[7]);
        p.ld_ln = op.i[9];
        p.ln_eps = op.f[0];
        if (op.i[8] == 2) {
          p.ln_x = 1;
          p.ln_out = reinterpret_cast<f16*>(op.p[7]);
          p.gn_part = reinterpret_cast<double*>(op.p[10]);
          p.gn_bar = reinterpret_cast<unsigned*>(op.p[11]);
          p.gn_fault = t2v_coop_fault_word();
          t2v_exchange_ids(&p.gn_seq, &p.gn_want);
        }
      }
      // the large-tile kernel advances its source pointers by whole k-tiles: needs K % BK == 0
      if (tile >= 1 && tile <= 24 && tile != 10 && p.gather != T2V_GATHER_CONV3X3_C8 && p.K % 64 == 0) return t2v_launch_gemm2(p, tile, s);
      return t2v_launch_gemm(p, s);
    }
    case T2V_OP_GROUPNORM: return t2v_launch_groupnorm(op, s);
    case T2V_OP_LAYERNORM: return t2v_launch_layernorm(op, s);
    case T2V_OP_ATTENTION: return t2v_launch_attention(op, s);
    case T2V_OP_SOFTMAX: return t2v_launch_softmax(op, s);
    case T2V_OP_NCTHW_TO_CL: return t2v_launch_ncthw_to_cl(op, s);
    case T2V_OP_CL_TO_NCTHW: return t2v_launch_cl_to_ncthw(op, s);
    case T2V_OP_TIME_EMBED: return t2v_launch_time_embed(op, s);
    case T2V_OP_COPY2D: return t2v_launch_copy2d(op, s);
    case T2V_OP_DDIM_STEP: return t2v_launch_ddim_step(op, s);
    case T2V_OP_LINCOMB: return t2v_launch_lincomb(op, s);
    case T2V_OP_RELPOS_ATTN: return t2v_launch_relpos_attention(op, s);
    case T2V_OP_EMBED_ROWS: return t2v_launch_embed_rows(op, s);
    case T2V_OP_TO_UINT8: return t2v_launch_to_uint8(op, s);
    case T2V_OP_RESHARD_ROWS: return t2v_launch_reshard_rows(op, s);
    case T2V_OP_MEMSET: {
      const size_t bytes = (size_t)(uint32_t)op.i[0] | ((size_t)(uint32_t)op.i[1] << 32);
      return hipMemsetAsync(reinterpret_cast<void*>(op.p[0]), 0, bytes, s);
    }
    default: return hipErrorInvalidValue;
  }
}

int run_resolved(const t2v_op* ops, int n, const uint64_t* ext, int n_ext, hipStream_t s, float* ms, t2v_comm* comm = nullptr) {
  {
    // a kernel of an EARLIER run gave up at a grid barrier (norm.hip): that run's results are invalid — say so now, once
    std::string why;
    if (t2v_async_fault_consume(&why)) return fail(T2V_ERR_ASYNC, why);
  }
  std::vector<hipEvent_t> ev;
  if (ms) {
    ev.resize(n + 1);
    for (auto& e : ev)
      if (hipEventCreate(&e) != hipSuccess) return fail(T2V_ERR_LAUNCH, "hipEventCreate failed");
    (void)hipEventRecord(ev[0], s);
  }
  for (int k = 0; k < n; ++k) {
    t2v_op op = ops[k];
    if (!resolve_ptrs(op, ext, n_ext)) {
      char buf[96];
      snprintf(buf, sizeof buf, "op %d (tag %d): unresolved external pointer slot", k, op.tag);
      return fail(T2V_ERR_BAD_ARG, buf);
    }
    if (op.kind == T2V_OP_ALLGATHER || op.kind == T2V_OP_HALO_EXCHANGE || op.kind == T2V_OP_ALLTOALL || op.kind == T2V_OP_STATS_HALO) {
      const size_t bytes = (size_t)(uint32_t)op.i[0] | ((size_t)(uint32_t)op.i[1] << 32);
      std::string err;
      const int rc = op.kind == T2V_OP_STATS_HALO
                         ? t2v_comm_stats_halo(comm, reinterpret_cast<void*>(op.p[0]), bytes, op.i[2], op.i[3], reinterpret_cast<void*>(op.p[1]),
                                               (size_t)(uint32_t)op.i[4] | ((size_t)(uint32_t)op.i[5] << 32), op.i[6], op.i[7], op.i[8], s, err)
                     : op.kind == T2V_OP_ALLGATHER
                         ? t2v_comm_allgather(comm, reinterpret_cast<void*>(op.p[0]), bytes, op.i[2], op.i[3], s, err)
                         : op.kind == T2V_OP_HALO_EXCHANGE
                               ? t2v_comm_halo(comm, reinterpret_cast<void*>(op.p[0]), bytes, op.i[2], op.i[3], op.i[4], s, err)
                               : t2v_comm_alltoall(comm, reinterpret_cast<void*>(op.p[0]), reinterpret_cast<void*>(op.p[1]), bytes, op.i[2],
                                                   op.i[3], op.i[4], op.i[5], op.i[6], s, err);
      if (rc != T2V_OK) {
        char buf[300];
        snprintf(buf, sizeof buf, "op %d (kind %d, tag %d): %s", k, op.kind, op.tag, err.c_str());
        return fail(rc, buf);
      }
      if (ms) (void)hipEventRecord(ev[k + 1], s);
      continue;
    }
    const hipError_t e = launch_op(op, s);
    if (e == hipErrorCooperativeLaunchTooLarge) {
      // A launch that needs its whole grid co-resident was refused.  Either a workgroup of an EARLIER launch of this run gave up waiting
      // and raised the fault word (the co-resident path switches off mid-run): that is the asynchronous fault, report it as such now —
      // or the lowering's occupancy table disagrees with this device (other architecture / LDS size / T2V_DEVICE_CUS): a distinct code,
      // so that the caller lowers again without fused norms instead of failing the same way on every call.
      std::string why;
      if (t2v_async_fault_consume(&why)) return fail(T2V_ERR_ASYNC, why);
      char buf[260];
      snprintf(buf, sizeof buf, "op %d (kind %d, tag %d): the launch needs its whole grid co-resident and the occupancy check refused it", k, op.kind, op.tag);
      return fail(T2V_ERR_RESIDENCY, buf);
    }
    if (e != hipSuccess) {
      char buf[200];
      snprintf(buf, sizeof buf, "op %d (kind %d, tag %d) launch failed: %s", k, op.kind, op.tag, hipGetErrorString(e));
      return fail(T2V_ERR_LAUNCH, buf);
    }
    if (ms) (void)hipEventRecord(ev[k + 1], s);
  }
  if (ms) {
    if (hipStreamSynchronize(s) != hipSuccess) return fail(T2V_ERR_LAUNCH, "stream sync failed");
    for (int k = 0; k < n; ++k) (void)hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]);
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  return T2V_OK;
}

}  // namespace

struct t2v_plan {
  std::vector<t2v_op> ops;
  t2v_comm* comm = nullptr;   // borrowed (t2v_plan_set_comm)
};

extern "C" {

int t2v_abi_version(void) { return T2V_ABI_VERSION; }

const char* t2v_last_error(void) { return g_err.c_str(); }

int t2v_device_info(char* name, int len, int* compute_units, uint64_t* hbm_bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(T2V_ERR_NO_DEVICE, "no HIP device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(T2V_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (name && len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)len - 1);
    name[len - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
  return T2V_OK;
}

int t2v_async_status(void) {
  std::string why;
  if (t2v_async_fault_consume(&why)) return fail(T2V_ERR_ASYNC, why);
  return T2V_OK;
}

int t2v_sync_reset(void* sync_words, void* stream) {
  if (!sync_words) return fail(T2V_ERR_BAD_ARG, "null sync words");
  const hipError_t e = hipMemsetAsync(sync_words, 0, sizeof(int32_t) * (T2V_SYNC_INTS + T2V_SYNC_BARRIER_INTS), reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(T2V_ERR_LAUNCH, std::string("hipMemsetAsync of the sync words failed: ") + hipGetErrorString(e));
  return T2V_OK;
}

int t2v_run_ops(const t2v_op* ops, int n, const uint64_t* ext, int n_ext, void* stream) {
  if (ops == nullptr || n < 0) return fail(T2V_ERR_BAD_ARG, "null program");
  for (int k = 0; k < n; ++k) {
    const int rc = validate_op(ops[k], k);
    if (rc != 0) return rc;
  }
  return run_resolved(ops, n, ext, n_ext, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int t2v_plan_create(const t2v_op* ops, int n, t2v_plan** out) {
  if (ops == nullptr || n <= 0 || out == nullptr) return fail(T2V_ERR_BAD_ARG, "null program");
  for (int k = 0; k < n; ++k) {
    const int rc = validate_op(ops[k], k);
    if (rc != 0) return rc;
  }
  t2v_plan* p = new t2v_plan();
  p->ops.assign(ops, ops + n);
  *out = p;
  return T2V_OK;
}

int t2v_plan_num_ops(const t2v_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

int t2v_plan_run(t2v_plan* plan, const uint64_t* ext, int n_ext, void* stream) {
  if (!plan) return fail(T2V_ERR_BAD_ARG, "null plan");
  return run_resolved(plan->ops.data(), (int)plan->ops.size(), ext, n_ext, reinterpret_cast<hipStream_t>(stream), nullptr, plan->comm);
}

int t2v_plan_run_timed(t2v_plan* plan, const uint64_t* ext, int n_ext, void* stream, float* ms) {
  if (!plan || !ms) return fail(T2V_ERR_BAD_ARG, "null plan / ms");
  return run_resolved(plan->ops.data(), (int)plan->ops.size(), ext, n_ext, reinterpret_cast<hipStream_t>(stream), ms, plan->comm);
}

void t2v_plan_destroy(t2v_plan* plan) { delete plan; }

int t2v_plan_set_comm(t2v_plan* plan, t2v_comm* comm) {
  if (!plan) return fail(T2V_ERR_BAD_ARG, "null plan");
  plan->comm = comm;
  return T2V_OK;
}

int t2v_comm_unique_id(unsigned char id[128]) {
  if (!id) return fail(T2V_ERR_BAD_ARG, "null id");
  std::string err;
  const int rc = t2v_comm_impl_unique_id(id, err);
  return rc == T2V_OK ? rc : fail(rc, err);
}

int t2v_comm_create(const unsigned char id[128], int nranks, int rank, t2v_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(T2V_ERR_BAD_ARG, "bad communicator arguments");
  std::string err;
  const int rc = t2v_comm_impl_create(id, nranks, rank, out, err);
  return rc == T2V_OK ? rc : fail(rc, err);
}

int t2v_comm_size(const t2v_comm* comm) { return t2v_comm_impl_size(comm); }

int t2v_comm_window_create(t2v_comm* comm, uint64_t slot_bytes, unsigned char handle_out[64]) {
  if (!comm || !handle_out || slot_bytes == 0) return fail(T2V_ERR_BAD_ARG, "window_create needs a communicator, a slot size and room for the handle");
  std::string err;
  const int rc = t2v_comm_impl_window_create(comm, (size_t)slot_bytes, handle_out, err);
  return rc == T2V_OK ? rc : fail(rc, err);
}

int t2v_comm_window_open(t2v_comm* comm, const unsigned char* handles) {
  if (!comm) return fail(T2V_ERR_BAD_ARG, "window_open needs a communicator");
  std::string err;
  const int rc = t2v_comm_impl_window_open(comm, handles, err);
  return rc == T2V_OK ? rc : fail(rc, err);
}

const char* t2v_comm_window_kind(const t2v_comm* comm) { return t2v_comm_impl_window_kind(comm); }

void t2v_comm_counters(const t2v_comm* comm, uint64_t out[2]) {
  unsigned long long v[2];
  t2v_comm_impl_counters(comm, v);
  out[0] = v[0];
  out[1] = v[1];
}

void t2v_comm_destroy(t2v_comm* comm) { t2v_comm_impl_destroy(comm); }

int t2v_comm_all_gather(t2v_comm* comm, void* base, uint64_t bytes, void* stream) {
  if (!comm || !base || bytes == 0) return fail(T2V_ERR_BAD_ARG, "all-gather needs a communicator, a base pointer and a part size");
  std::string err;
  const int rc = t2v_comm_impl_all_gather(comm, base, (size_t)bytes, reinterpret_cast<hipStream_t>(stream), err);
  return rc == T2V_OK ? rc : fail(rc, err);
}

int t2v_unet_forward(t2v_plan* plan, const void* x, const float* t, const void* ctx, void* eps_out, void* stream) {
  uint64_t ext[T2V_EXT_SLOTS] = {0};
  ext[T2V_EXT_X] = (uint64_t)x;
  ext[T2V_EXT_T] = (uint64_t)t;
  ext[T2V_EXT_CTX] = (uint64_t)ctx;
  ext[T2V_EXT_OUT] = (uint64_t)eps_out;
  return t2v_plan_run(plan, ext, T2V_EXT_SLOTS, stream);
}

int t2v_vae_decode(t2v_plan* plan, const void* z, void* img_out, void* stream) {
  uint64_t ext[T2V_EXT_SLOTS] = {0};
  ext[T2V_EXT_X] = (uint64_t)z;
  ext[T2V_EXT_OUT] = (uint64_t)img_out;
  return t2v_plan_run(plan, ext, T2V_EXT_SLOTS, stream);
}

int t2v_ddim_step(t2v_plan* plan, const void* xt, const void* eps_pair, const void* noise, void* xt_out,
                  const float coef[6], void* stream) {
  if (!plan || plan->ops.size() != 1 || plan->ops[0].kind != T2V_OP_DDIM_STEP)
    return fail(T2V_ERR_BAD_ARG, "t2v_ddim_step needs a single DDIM_STEP plan");
  t2v_op op = plan->ops[0];
  for (int k = 0; k < 6; ++k) op.f[k] = coef[k];
  uint64_t ext[T2V_EXT_SLOTS] = {0};
  ext[T2V_EXT_XT] = (uint64_t)xt;
  ext[T2V_EXT_EPS] = (uint64_t)eps_pair;
  ext[T2V_EXT_NOISE] = (uint64_t)noise;
  ext[T2V_EXT_XT_OUT] = (uint64_t)xt_out;
  return run_resolved(&op, 1, ext, T2V_EXT_SLOTS, reinterpret_cast<hipStream_t>(stream), nullptr);
}

}  // extern "C"
