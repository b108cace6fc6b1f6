/*
 * t2v_hip.h — C ABI of libt2v_hip.so, the MI355X (gfx950) implementation of the
 * ModelScope / ZeroScope text-to-video denoising hot path.
 *
 * The reference (kabachuha/sd-webui-text2video) has no FFI: its seam is Python duck-typing
 * (SURVEY.md §8b).  This header is the boundary a maintainer binds with ctypes
 *  to replace the torch.nn delegation layer under
 *   - UNetSD.forward / _forward_single      scripts/modelscope/t2v_model.py:386-501   (B4)
 *   - AutoencoderKL.decode                  scripts/modelscope/t2v_model.py:1646-1649 (B5)
 *   - GaussianDiffusion.sample step update  scripts/samplers/ddim/gaussian_sampler.py:257-283
 *
 * Design: the host (Python) lowers a model + shape into a flat *denoise program* — an array
 * of t2v_op records over one activation arena and one packed-weight arena — and this library
 * executes the program by launching hand-written HIP kernels on the caller's HIP stream.
 * Plain pointers and sizes only; no torch types.  All device memory is owned by the caller;
 * pointers are borrowed for the duration of a call.  No internal threads; one in-flight call
 * per plan.  Every function returns 0 on success or a negative T2V_ERR_* code; the message of
 * the last failure is available from t2v_last_error().
 */
#ifndef T2V_HIP_H
#define T2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2V_ABI_VERSION 8   /* 8 (round 6): peer windows (t2v_comm_window_create / _open / t2v_comm_counters): exchanges as device-initiated stores into IPC-mapped mailboxes;
                               7 (round 6): ATTENTION p[6] / i[17] / i[18] (V^T scratch: spatial self-attention with LDS-DMA tiles); T2V_ERR_RESIDENCY — a launch that needs its whole grid co-resident was refused by the occupancy check (the caller
                               lowers again without norms fused into GEMM epilogues); a launch refused because a fault was raised mid-run reports T2V_ERR_ASYNC;
                               6 (round 5, second half): T2V_OP_STATS_HALO — the statistics parts of a T-sharded cross-frame GroupNorm and the RAW boundary
                               frames of the temporal convolution behind it in ONE grouped exchange; GROUPNORM i[21] / i[22] (phase 2 also normalises the
                               received boundary frames); t2v_comm_all_gather (eps pair / frame gathers over the library's communicators);
                               5 (round 5): T2V_OP_NI 32 / T2V_OP_NP 12 (record grew), T2V_EPI_GN — GroupNorm (+SiLU) of a GEMM's result fused into its epilogue;
                               4 (round 4): t2v_async_status, t2v_sync_reset, T2V_ERR_ASYNC (bounded grid barrier of the single-pass GroupNorm);
                               3 (round 3): T2V_EPI_TATTN + tile 10, GROUPNORM i[15] / p[5], GEMM p[7] tickets, RELPOS i[17], low-order outputs of the cast ops, T2V_SYNC_* */

/* error codes */
#define T2V_OK 0
#define T2V_ERR_BAD_ARG (-1)
#define T2V_ERR_UNSUPPORTED (-2)
#define T2V_ERR_LAUNCH (-3)
#define T2V_ERR_NO_DEVICE (-4)
#define T2V_ERR_COMM (-5)      /* RCCL not loadable / communicator call failed */
#define T2V_ERR_ASYNC (-6)     /* a kernel of an EARLIER run raised a fault (bounded grid barrier timed out): that run's results are invalid */
#define T2V_ERR_RESIDENCY (-7) /* a fused-norm launch (T2V_EPI_GN, cross-tile LayerNorm, cooperative GroupNorm) does not fit co-resident on this device */

/* ---- op kinds ------------------------------------------------------------------------- */
enum t2v_op_kind {
  T2V_OP_GEMM = 1,        /* implicit-GEMM conv / linear on MFMA, fused epilogue            */
  T2V_OP_GROUPNORM = 2,   /* GroupNorm(32) (+SiLU), per-frame or cross-frame statistics     */
  T2V_OP_LAYERNORM = 3,   /* row LayerNorm fp32 -> fp16                                     */
  T2V_OP_ATTENTION = 4,   /* softmax(QK^T * scale) V, head_dim 40|64|80|160, strided batch  */
  T2V_OP_SOFTMAX = 5,     /* row softmax fp32 -> fp16 (VAE single-head attention)           */
  T2V_OP_NCTHW_TO_CL = 6, /* [B,C,F,H,W] -> channels-last fp16 tokens [B*F*H*W, ld]         */
  T2V_OP_CL_TO_NCTHW = 7, /* channels-last fp32 tokens -> [B,C,F,H,W]                       */
  T2V_OP_TIME_EMBED = 8,  /* sinusoidal timestep embedding (cos | sin) -> fp16              */
  T2V_OP_COPY2D = 9,      /* strided 2-D copy / cast / SiLU (concat, fp32->fp16 staging)    */
  T2V_OP_DDIM_STEP = 10,  /* DDIM_Gaussian update with half-channel CFG                     */
  T2V_OP_MEMSET = 11,     /* zero a byte range                                              */
  T2V_OP_LINCOMB = 12,    /* out = sum_i c_i * T_i (<= 6 latent-sized tensors): UniPC / DDIM updates */
  T2V_OP_RELPOS_ATTN = 13, /* LVDM temporal attention with relative-position K / V terms (frames <= 32) */
  T2V_OP_EMBED_ROWS = 14,  /* token + positional embedding lookup (CLIP text towers) */
  T2V_OP_TO_UINT8 = 15,    /* tensor2vid: float video -> uint8 frames [F,H,(i W),3], truncating (t2v_pipeline.py:447-460) */
  T2V_OP_ALLGATHER = 16,   /* in-place all-gather of equal byte parts over the plan's communicator (RCCL, launch stream) */
  T2V_OP_HALO_EXCHANGE = 17, /* +-1 frame neighbour exchange of a [F+2]-frame token buffer over the plan's communicator */
  T2V_OP_RESHARD_ROWS = 18,  /* chunked row regrouping (pack / unpack of a frame <-> pixel resharding), optional fp32 residual */
  T2V_OP_ALLTOALL = 19,      /* frame <-> pixel resharding of a T-sharded clip over the plan's communicator */
  T2V_OP_STATS_HALO = 20,    /* GroupNorm statistics parts to every rank + raw boundary frames to the two neighbours: one grouped exchange */
  T2V_OP_KIND_MAX = 21
};

/* GEMM gather modes: how row m / reduction index k of the A operand are addressed          */
enum t2v_gather {
  T2V_GATHER_PLAIN = 0,   /* A[m*lda + k]                                   (nn.Linear, 1x1) */
  T2V_GATHER_CONV3X3 = 1, /* 3x3 spatial conv, pad 1, stride 1|2, optional nearest-2x
                             upsample folded in; K = 9*Cin, Cin % 64 == 0, reduction order
                             (64-channel chunk, tap, channel)                                */
  T2V_GATHER_TCONV3 = 2,  /* (3,1,1) temporal conv over frames, pad (1,0,0); K = 3*Cin       */
  T2V_GATHER_CONV3X3_C8 = 3 /* 3x3 spatial conv with Cin == 8 (4 real + 4 zero channels)     */
};

/* GEMM epilogues */
#define T2V_EPI_NONE 0
#define T2V_EPI_GEGLU 1   /* out[m, j] = (a_j + b_j) * gelu(g_j + c_j); weight rows interleaved
                             in blocks of 16 = 8 value rows | 8 gate rows                    */
#define T2V_EPI_TATTN 2   /* fused QKV projection + temporal self-attention (tile 10): W = [heads][q 64 | k 64 | v 64][K]
                             (head-major), N = 192 * heads; the tile's rows are i[10] pixels x F (= i[8] <= 32) frames of
                             one sample (input / output row of (sample s, frame f, pixel x) = (s*F + f)*HW + x, HW = i[9]);
                             M = samples * ceil(HW / i[10]) * 192 (tile rows, not tokens); out fp16 [samples*F*HW, ldc]:
                             softmax(q k^T f[1]) v of every pixel's frame sequence, head h at columns 64 h .. 64 h + 63 */

#define T2V_EPI_STATS 3   /* = T2V_EPI_NONE, and p[7] (fp32 [ceil(M/32)][2][N]) receives per 32-row strip the column sums and sums of squares of the
                             STORED result (after bias / row bias / activation / residual; fp16 outputs: of the rounded values) — the input of a
                             GROUPNORM phase 3, which then needs no pass over the tensor for its statistics.  No split-K, no fused LayerNorm. */

#define T2V_EPI_GN 4      /* = T2V_EPI_NONE, and the GroupNorm (+SiLU) that consumes the result runs INSIDE the epilogue (round 5): every workgroup keeps
                             its bias / row-bias / residual-added fp32 tile in registers, publishes {sum, sum of squares} per (statistics instance, group
                             piece), all workgroups of the launch meet at a bounded grid barrier (p[11]: T2V_SYNC_BARRIER_INTS words, as GROUPNORM p[5]),
                             fold the partials of their instances in a fixed order (bit-identical statistics everywhere) and write
                             p[9] = fp16 [M, i[25]] = (silu)((v - mean) * rstd * gamma + beta) — no statistics pass, no second launch, and for a
                             result that only the norm consumes (i[29] = 1) no fp16 / fp32 round trip of the tensor at all.
                             i[24] rows per statistics instance (a frame: H*W; all frames of a sample: F*H*W; multiple of 32, >= the tile's rows),
                             i[25] ld of the normalised output, i[26] SiLU, i[27] = 1: low-order fp16 image at column N + n (ld >= 2N, as GROUPNORM i[16]),
                             i[28] groups (N % groups == 0), i[29] = 1: `out` (p[5]) is NOT written; f[2] eps;
                             p[8] gamma | beta fp32 [2N], p[9] normalised output fp16, p[10] scratch fp64 [tiles_m][2][tiles_n][T2V_GN_PIECES][2],
                             p[11] grid-barrier words.  No split-K / GEGLU / fused LayerNorm; tiles 8 / 11 (whole rows, N == 320), 0, 3, 5 — the
                             grid must be co-resident on the device (the library checks with the occupancy API and refuses otherwise). */
#define T2V_EPI_XATTN 5   /* fused to_q projection + text cross-attention (round 5): out fp16 [M, ldc] = softmax(q k^T f[1]) v per (sample, head of 64
                             channels), q = A W^T (no bias), keys = the i[25] (<= 96) text tokens.  p[8] = K fp16 [samples][i[25]][i[24]] (this
                             site's N columns; i[27] elements between samples), p[9] = V^T fp16 [samples][N][i[26]] (keys contiguous, finite
                             beyond i[25]; i[28] elements between samples), both written by step-invariant ops; the sample of row m is
                             m / i[15].  N % 64 == 0; tiles 8 / 11 (N == 320) and 0 / 5 (N % 128 == 0); plain gather, no split-K, fp16 out. */
#define T2V_GN_PIECES 36  /* group pieces (group x column tile intersections) per column tile in the T2V_EPI_GN scratch */

/* dtype tags */
#define T2V_F16 0
#define T2V_F32 1

/* external pointer slots: a pointer field whose value is < T2V_EXT_SLOTS is replaced at run
 * time by ext[value] (value 0 = null). */
#define T2V_EXT_SLOTS 16
#define T2V_EXT_X 1      /* UNet: x [B,4,F,h,w]   | VAE: z [n,4,h,w]   | text tower: token ids int32 [B,L] */
#define T2V_EXT_T 2      /* UNet: timesteps, float32 [B]                                     */
#define T2V_EXT_CTX 3    /* UNet: context [B,L,ctx_dim]                                      */
#define T2V_EXT_OUT 4    /* UNet: eps [B,4,F,h,w] | VAE: image [n,3,8h,8w] | text tower: z fp32 [B,L,width] */
#define T2V_EXT_XT 5     /* DDIM: x_t in                                                     */
#define T2V_EXT_XT_OUT 6 /* DDIM: x_{t-1} out                                                */
#define T2V_EXT_NOISE 7  /* DDIM: eta-noise (may be null when sigma == 0)                    */
#define T2V_EXT_EPS 8    /* DDIM: stacked UNet outputs [2,C,F,h,w] (cond, uncond)            */

/* GroupNorm pass-1 granularity: rows of one statistics instance reduced per workgroup.  The
 * caller sizes the scratch as nparts*n_inst*ceil(rows/16)*groups*16 + n_inst*groups*8 bytes. */
#define T2V_GN_ROWS_PER_BLOCK 64

/* Device-side synchronisation words a program may hand to its ops: T2V_SYNC_INTS int32 arrival counters for the split-K fold
 * (GEMM p[7]) followed by T2V_SYNC_BARRIER_INTS more for the two-level grid barrier of the single-pass GroupNorm (GROUPNORM
 * p[5] points at the first of them).  All zero before the first launch; the kernels leave them zero (counters) or monotonic
 * (barrier generation).  They must not alias any buffer an op of the program writes.  t2v_sync_reset() zeroes the region on a
 * stream (call it when the program is bound to an arena, before the first run — never while a run is in flight). */
#define T2V_SYNC_INTS 4096
#define T2V_SYNC_BARRIER_INTS 512
#define T2V_GN_PART_BYTES 2097152 /* bytes of the statistics-exchange scratch (GEMM p[10]) every fused-norm launch may use: the launcher sizes its row chunks by it */

#define T2V_OP_NI 32
#define T2V_OP_NF 8
#define T2V_OP_NP 12

/* One program record.  Field meaning per kind:
 *
 * GEMM:  out[M,N] = epi( gather(A)[M,K] * W[N,K]^T )       fp16 operands, fp32 accumulate
 *   i: 0 M, 1 N, 2 K, 3 lda, 4 ldw, 5 ldc, 6 ldr, 7 gather, 8 Hin|F, 9 Win|HW, 10 Cin,
 *      11 stride, 12 upsample, 13 Hout, 14 Wout, 15 rows_per_batch, 16 epilogue,
 *      17 out dtype, 18 act (0 none, 1 SiLU), 19 split_k, 20 bias_along_m, 21 ldrb,
 *      22 tile (0 = 128x128-class kernel; 1 256x256, 2 256x320, 3 128x256 (8 waves, 3-stage ring), 4 / 5 128x128 with a 4-deep
 *         ring on 4 / 8 waves, 6 / 7 = 1 / 2 with the two-group ping-pong schedule, 8 / 9 192x320 / 192x256 on 12 waves,
 *         10 = 192x192 on 12 waves, T2V_EPI_TATTN only; 11 = 128x320 on 8 waves, 12 = 64x64 on 4 waves with a 4-deep ring),
 *      23 tconv halo (input rows are [clip][F+2][HW]: one halo frame either side, T-sharding);
 *         for CONV3X3: 1 = zero padding (0,1,0,1) instead of (1,1,1,1) (LDM encoder Downsample, taps at +0..+2)
 *      PLAIN gather only: 12 = R > 0 -> the residual has R rows and row m >= R reads residual row m - R; 13 = R > 0 -> likewise for the A
 *         operand (M <= 2R, no split-K): tensors shared by the cond | uncond pair of a guided step are computed once and wrapped;
 *      PLAIN gather only: 11 = 1 -> hi + lo fp16 output (fp16 out, plain epilogue, ldc >= 2N): out[m, N + n] = fp16(v - float(fp16(v)))
 *         beside out[m, n] = fp16(v) — a consumer GEMM over rows [hi | lo] with weights [W | W] (K = 2N) sees v with ~22 bits;
 *      PLAIN gather only: 8 = 1 -> fused LayerNorm second output (tile 8 or 11, N == 320, fp32 out, no split-K): p[7] fp16 [M, i[9]] =
 *         LayerNorm(out row, eps f[0]) * gamma + beta with p[3] = fp32 [2N] gamma | beta (instead of a row bias)
 *   p: 0 A fp16, 1 W fp16 [N,K], 2 bias fp32 [N] (or [M] if bias_along_m), 3 rowbias fp32
 *      [M/rows_per_batch, ldrb], 4 residual fp32 [M,ldr], 5 out, 6 split-K workspace fp32 [split_k, M, N],
 *      7 split-K (epilogue NONE): T2V_SYNC_INTS zeroed int32 tile tickets -> the last workgroup of a tile to arrive folds the
 *        slabs in split order and applies the epilogue (no reduction launch); 0 -> a reduction kernel follows;
 *      8 .. 11 and i[24 .. 29], f[2]: T2V_EPI_GN (above)
 * GROUPNORM: i: 0 n_inst, 1 rows_per_inst, 2 C, 3 ld_in, 4 groups, 5 in dtype, 6 silu,
 *      7 ld_out, 8 phase (0 whole op | 1 statistics only | 2 fold gathered parts + normalise | 3 statistics from the producing GEMM:
 *         p[6] = its T2V_EPI_STATS strips fp32 [n_inst * rows / 32][2][i[17]], rows % 32 == 0 — a fold of the strips + ONE apply pass),
 *      9 nparts, 10 this rank's part, 11 rows per workgroup (0: T2V_GN_ROWS_PER_BLOCK; sizes the scratch),
 *      12 single-launch variant (phase 0 only, (C/groups) % 4 == 0): one workgroup per (instance, group);
 *      14 rows of the whole instance over all parts (0 = rows * nparts; T-sharded clips with uneven slices);
 *      15 single-PASS cooperative variant allowed (phase 0, groups <= 32, p[5] given): grid <= one workgroup per CU, the tensor
 *         is read once into registers, statistics meet at a grid barrier (p[5] = T2V_SYNC_BARRIER_INTS zero-initialised uint32 words); the library
 *         uses it when the instance chunks fit (else the launches above on the same scratch);
 *      16 = 1 (phases 0 / 2, ld_out >= 2C): hi + lo operand split — the low-order fp16 image fp16(y - float(fp16(y))) of every output
 *         value goes to column C + c of the same output row (consumer: a GEMM with K = 2C against weights [W | W]);
 *      21 / 22 (phase 2, n_inst == 1): rows in front of / behind x that are normalised with the same statistics and written in front of /
 *         behind `out` — the neighbours' RAW boundary frames of a halo-padded buffer (T2V_OP_STATS_HALO); i[1] stays this rank's own rows
 *         (it sizes the scratch), i[14] the rows the statistics cover;
 *      scratch, phase 0: block partials [n_inst][nblk][groups][2] fp64, then {mean, rstd} fp32;  phases 1 / 2: gathered parts
 *      [nparts][n_inst][groups][2] fp64 (phase 1 folds this rank's block partials into its part; ALLGATHER of
 *      n_inst*groups*16 bytes per part), then this rank's block partials, then {mean, rstd};
 *      f: 0 eps;  p: 0 x, 1 gamma, 2 beta, 3 out fp16, 4 scratch, 5 grid-barrier words (i[15]), 6 producer strips (phase 3),
 *      8 (optional; phases 0 / 2 / 3) second output: the RAW input as fp16 [n_inst * rows, i[19]] (+ its low-order image at column i[20] > 0) — the
 *        operand of a 1x1 convolution that reads the same tensor (ResBlock skip_connection beside in_layers, t2v_model.py:965)
 *      7 (i[15], optional) exchange-record region of i[18] bytes that nothing but fused-norm launches ever writes (zero at bind time): the
 *        single-pass kernel then exchanges TAGGED records (no grid barrier) — the same region GEMM p[10] names
 * LAYERNORM: i: 0 M, 1 C, 2 ld_in, 3 ld_out, 4 workgroup cap (0 = 2048; rows beyond 4 x cap are walked grid-stride); f: 0 eps;
 *      p: 0 x fp32, 1 gamma, 2 beta, 3 out fp16
 * ATTENTION: i: 0 nq, 1 nk, 2 heads, 3 batch_outer, 4 batch_inner, 5..7 q strides (seq,
 *      outer, inner), 8..10 k/v strides, 11..13 out strides (elements; head h at +head_dim*h),
 *      14 head_dim (0 = 64), 15 causal (1: key s visible to query t iff s <= t), 16 low-order output offset (elements, 0 = none): also
 *      store fp16(o - float(fp16(o))) at out + i[16] — rows [hi | lo] for a K-doubled output projection;  f: 0 scale; p: 0 q, 1 k, 2 v, 3 out (all fp16)
 *      ABI 7, optional: p 6 = scratch fp16 [batch_outer * batch_inner * heads * 64, i[17]] with i[17] = nk rounded up to a multiple of 64
 *      (head_dim 64, not causal): V is transposed into it once per launch and the K / V^T tiles are staged by LDS-DMA (same bits as without
 *      the scratch; pays from ~512 keys); i[18] = waves per 64-key tile (0 = 8 | 4 | 8)
 * RELPOS_ATTN: i: as ATTENTION with nk == frames of the clip (<= 32), 14 head_dim (multiple of 8, <= 160),
 *      15 max relative position R, 16 q_off: the nq queries are frames [q_off, q_off + nq) of the clip (nq == nk, q_off == 0
 *      unless the clip is T-sharded: then s - t below is s - (t + q_off)); 17 = 1: use the MFMA kernel where it applies (nq == nk,
 *      q_off == 0, nk - 1 <= R <= 31, head_dim 40 | 64 | 80 | 160; the relative tables are staged as fp16) — else the VALU kernel;
 *      17 = 2 (round 5): the persistent MFMA kernel for whole clips of nk <= 16 frames (nq == nk, q_off == 0, R >= nk - 1, same head
 *      dims), with the tables ALSO given packed for it: p[6] = fp16 [32][DK] rows jl = Ek[jl + R - (nk-1)] (DK = head_dim rounded up
 *      to 16; zero beyond the table / head_dim), p[7] = fp16 [DV][32] = Ev transposed, column jl = Ev[jl + R - (nk-1)] (DV = head_dim
 *      rounded up to 32) — where it does not apply the VALU kernel runs on p[4], p[5];
 *      18 low-order output offset (as ATTENTION i[16]);  f: 0 scale;  p: 0 q, 1 k, 2 v, 3 out (fp16), 4 Ek fp32 [2R+1, head_dim],
 *      5 Ev fp32 [2R+1, head_dim]:  sim[t,s] = scale * q[t].(k[s] + Ek[clip(s-t)]),
 *      out[t] = sum_s softmax_s(sim)[t,s] * (v[s] + Ev[clip(s-t)])   (attention_temporal.py:107-144)
 * SOFTMAX: i: 0 rows, 1 cols, 2 ld_in, 3 ld_out; f: 0 scale; p: 0 in fp32, 1 out fp16
 * NCTHW_TO_CL: i: 0 B, 1 C, 2 F, 3 HW, 4 ld_out, 5 in dtype, 6 samples in the source (0 = B; fewer: output sample b reads
 *      source sample b % i[6] — the cond | uncond pair of a guided step shares x_t); f: 0 scale; p: 0 in, 1 out fp16,
 *      2 (optional) low-order fp16 image, same layout: out_lo = fp16(v - float(fp16(v))) — hi + lo operand split of a consumer GEMM;
 *      i[7] = 1 (ld_out >= 2C): the low-order images are written into channels C .. 2C-1 of the same row instead (a consumer whose
 *      weights repeat W for those channels computes (hi + lo) . W in one pass)
 * CL_TO_NCTHW: i: 0 B, 1 C, 2 F, 3 HW, 4 ld_in, 5 out dtype; p: 0 in fp32, 1 out
 * TIME_EMBED: i: 0 B, 1 dim; p: 0 t fp32 [B], 1 freqs fp32 [dim/2], 2 out fp16 [B,dim]
 * COPY2D: i: 0 rows, 1 cols, 2 ld_src, 3 ld_dst, 4 src dtype, 5 dst dtype, 6 act (0 none, 1 SiLU, 2 GELU(erf),
 *      3 quick-GELU x*sigmoid(1.702x)); p: 0 src, 1 dst, 2 (optional, fp32 -> fp16 only) low-order fp16 image of the cast, ld_dst
 * DDIM_STEP: i: 0 C (= samples * channels), 1 inner (F*h*w), 2 guided channels (per sample), 3 eps dtype, 4 x dtype, 5 mode,
 *      6 channels per sample (0 = C: one video per batch); x [samples, channels, inner], eps [2, samples, channels, inner];
 *      mode 0 (DDIM_Gaussian, gaussian_sampler.py:103-108,199-211,269-283):
 *        f: 0 sqrt_recip_ac, 1 sqrt_recipm1_ac, 2 sqrt(a_prev), 3 dir coef, 4 sigma (masked), 5 guidance scale
 *      mode 1 (LDM DDIM, samplers/ddim/sampler.py:197-219):  x0 = (x - f0*e)/f1;  out = f2*x0 + f3*e + f4*noise
 *        f: 0 sqrt(1-a_t), 1 sqrt(a_t), 2 sqrt(a_prev), 3 sqrt(1-a_prev-sigma^2), 4 sigma, 5 guidance scale
 *      p: 0 xt, 1 eps pair, 2 noise, 3 out
 * LINCOMB: i: 0 n elements, 1 n terms (<= 6), 2 out dtype, 3..8 term dtypes; f: 0..5 coefficients;
 *      p: 0..5 terms, 6 out
 * MEMSET: i: 0 bytes (lo), 1 bytes (hi); p: 0 dst
 * EMBED_ROWS: out[r,:] = table[ids[r],:] + pos[r % L,:]   i: 0 rows, 1 width, 2 L, 3 vocab, 4 table dtype;
 *      p: 0 ids int32 [rows], 1 table [vocab,width], 2 pos fp32 [L,width], 3 out fp32 [rows,width]
 * TO_UINT8: out[f, y, i*W + x, c] = trunc(clamp(v*0.5 + 0.5, 0, 1) * 255), v = in[i*si + c*sc + f*sf + y*sy + x*sx]
 *      i: 0 NI (videos side by side), 1 C, 2 F, 3 H, 4 W, 5 in dtype, 6 arithmetic (0 fp32 | 1 every intermediate rounded
 *      to fp16, the reference's half-precision VAE path), 7 channel order reversed (RGB -> BGR), 8/9 si (lo, hi), 10 sc,
 *      11/12 sf (lo, hi), 13 sy, 14 sx (element strides);  p: 0 in, 1 out uint8
 * ALLGATHER: part q of `nparts` equal parts lives at base + q*bytes; this rank's part is already in place.
 *      i: 0/1 bytes per part (lo, hi), 2 nparts, 3 this rank's part;  p: 0 base.   nparts == 1: no-op.
 * HALO_EXCHANGE: token buffer of F+2 frames (frame = `bytes`): frame 1 -> previous rank's frame F+1 slot ... i.e. this rank
 *      sends its first real frame to `prev` and its last to `next`, and receives their boundary frames into frame 0 / F+1.
 *      i: 0/1 bytes per frame (lo, hi), 2 F (local frames), 3 prev rank in the communicator (-1 none), 4 next rank (-1 none);
 *      p: 0 base.   The collectives need t2v_plan_set_comm unless they are no-ops.
 * STATS_HALO: ALLGATHER of the statistics parts and HALO_EXCHANGE of the RAW (not yet normalised) boundary frames as ONE group of
 *      point-to-point transfers: every rank sends its part (p0 + part*bytes) to every other rank; frame 1 of the halo-padded raw buffer
 *      p1 goes to `prev`, frame F to `next`, theirs arrive in frame 0 / F + 1.  The consumer (GROUPNORM phase 2 with i[21] / i[22]) then
 *      normalises its own frames AND the received ones with the same gathered statistics — bit-identical to what the neighbour wrote for
 *      itself — so a T-sharded temporal convolution costs one exchange instead of two (t2v_model.py:1202-1229).
 *      i: 0/1 bytes per statistics part (lo, hi), 2 nparts, 3 this rank's part, 4/5 bytes per frame (lo, hi), 6 F (local frames),
 *      7 prev rank (-1 none), 8 next rank (-1 none);  p: 0 gathered parts, 1 raw buffer of F + 2 frames
 * RESHARD_ROWS: row r reads src row (r / P) * S_src + r % P and writes dst row (r / P) * S_dst + r % P (+ residual at the dst row)
 *      i: 0 rows, 1 cols, 2 P (rows per chunk), 3 S_src, 4 S_dst (chunk strides in rows), 5 ld_src, 6 ld_dst, 7 dtype (src == dst;
 *      fp16: cols % 8 == 0, fp32: cols % 4 == 0), 8 ld_res;  p: 0 src, 1 dst, 2 residual fp32 (optional, fp32 only)
 *      ABI 8: i[9] = nparts > 1: that many regroupings of this shape in ONE launch — part q reads p0 + q*i[10] elements, writes p1 + q*i[11]
 *      elements (residual p2 + q*i[12]); part i[13] (-1: none) is the rank's own, which does not travel: its source (i[14] = 1) or its
 *      destination (i[14] = 0) is p3 instead (the R packs in front of a frames -> pixels ALLTOALL, the R unpacks behind the way back)
 * ALLTOALL: slice q of the clip holds cnt(q) frames (i[4] each, i[5] on the LAST slice); chunk = bytes of one frame's share for one
 *      rank.  direction 0 (frames -> pixels): send cnt(me) chunks to every peer q from p0 + q*cnt(me)*chunk, receive cnt(q) chunks
 *      from q at p1 + q*i[4]*chunk;  direction 1 (pixels -> frames): send cnt(q) chunks to q from p0 + q*i[4]*chunk, receive cnt(me)
 *      chunks from q at p1 + q*cnt(me)*chunk.  The rank's own part is moved by RESHARD_ROWS ops.
 *      i: 0/1 chunk bytes (lo, hi), 2 nparts, 3 this rank's part, 4 frames per slice, 5 frames of the last slice, 6 direction;
 *      p: 0 send base, 1 receive base
 */
typedef struct t2v_op {
  int32_t kind;
  int32_t tag;                 /* free for the host (debug id) */
  int32_t i[T2V_OP_NI];
  float f[T2V_OP_NF];
  uint64_t p[T2V_OP_NP];
} t2v_op;

typedef struct t2v_plan t2v_plan; /* opaque: a validated copy of a program */

/* library / device */
int t2v_abi_version(void);
const char* t2v_last_error(void);
/* fills name (<= len bytes), number of CUs and total HBM bytes of the current HIP device */
int t2v_device_info(char* name, int len, int* compute_units, uint64_t* hbm_bytes);

/* Asynchronous faults.  The single-pass GroupNorm synchronises its workgroups with a grid barrier; the library only launches it when
 * the occupancy API says the whole grid is co-resident, and the wait is bounded: a workgroup that sees no release within 0.25 s
 * (another client of the device holds compute units) raises a flag in host-mapped memory and falls through instead of hanging the
 * device.  The run in flight then produced invalid results; the NEXT t2v_run_ops / t2v_plan_run* call returns T2V_ERR_ASYNC (once)
 * and the three-launch GroupNorm is used for the rest of the process.  t2v_async_status() reports the same condition without running
 * anything (call it after synchronising the stream at the end of a job): T2V_OK or T2V_ERR_ASYNC. */
int t2v_async_status(void);
/* test hook: the next `n` fused-norm launches (single-pass GroupNorm, T2V_EPI_GN, cross-tile LayerNorm) wait for exchange records that
 * nobody publishes — every waiter gives up after 0.25 s and the asynchronous fault is reported as above.  Never call it in production. */
void t2v_debug_poison_exchange(int n);
/* zero the T2V_SYNC_INTS + T2V_SYNC_BARRIER_INTS int32 words at `sync_words` on `stream` */
int t2v_sync_reset(void* sync_words, void* stream);

/* Execute `n` ops in order on `stream` (a hipStream_t; null = default stream). */
int t2v_run_ops(const t2v_op* ops, int n, const uint64_t* ext, int n_ext, void* stream);

/* Plans: validate + copy a program once, run it many times. */
int t2v_plan_create(const t2v_op* ops, int n, t2v_plan** out);
int t2v_plan_num_ops(const t2v_plan* plan);
int t2v_plan_run(t2v_plan* plan, const uint64_t* ext, int n_ext, void* stream);
/* Same, bracketing every op with HIP events on `stream`; ms[i] = duration of op i.
 * Synchronises the stream.  Used by bench.py for the live roofline measurement. */
int t2v_plan_run_timed(t2v_plan* plan, const uint64_t* ext, int n_ext, void* stream, float* ms);
void t2v_plan_destroy(t2v_plan* plan);

/* Communicators (T-axis sharding, one process per GPU): RCCL is dlopen'ed on first use, the library itself does not link
 * it.  Rank 0 of a group calls t2v_comm_unique_id and hands the 128 bytes to the others out of band (the host uses
 * torch.distributed's store); every rank then calls t2v_comm_create on its own device.  Collective ops of a plan run on
 * the launch stream, in program order with the kernels: a sharded UNet forward is ONE host call. */
typedef struct t2v_comm t2v_comm;
int t2v_comm_unique_id(unsigned char id[128]);
int t2v_comm_create(const unsigned char id[128], int nranks, int rank, t2v_comm** out);
int t2v_comm_size(const t2v_comm* comm);
void t2v_comm_destroy(t2v_comm* comm);
int t2v_plan_set_comm(t2v_plan* plan, t2v_comm* comm);   /* borrowed; must outlive the plan's runs */
/* Peer windows (ABI 8, csrc/comm.hip): device-initiated exchanges instead of RCCL calls.  Every rank creates a window on its
 * communicator (memory of the library on the current device; each message gets one of two slots of `slot_bytes` per peer) and
 * receives its 64-byte hipIpcMemHandle; the host gathers the handles of all ranks ([nranks][64], communicator order — the reference's
 * launcher has the same out-of-band channel, ddp_wrapper.py:9-13) and every rank opens them.  From then on a collective op of a plan
 * (and t2v_comm_all_gather) whose largest message fits a slot is ONE kernel of this library: workgroups store their share of the
 * message into the peer's window over xGMI, raise a sequence flag there, and wait (bounded: T2V_PEER_TIMEOUT_MS, default 20 s ->
 * T2V_ERR_ASYNC at the next call) for the peer's flag in their own window.  Larger ops keep going through RCCL.
 * t2v_comm_window_open(comm, NULL) releases the window again (the group found a rank that could not map a peer: nobody uses them).
 * t2v_comm_counters: out[0] = exchanges that went over the window, out[1] = exchanges that went through RCCL since creation. */
int t2v_comm_window_create(t2v_comm* comm, uint64_t slot_bytes, unsigned char handle_out[64]);
int t2v_comm_window_open(t2v_comm* comm, const unsigned char* handles);
void t2v_comm_counters(const t2v_comm* comm, uint64_t out[2]);
/* "uncached" | "finegrained" | "default": the kind of device memory the window got (uncached first — peers write it through the fabric and
 * this device polls it, so its L2 must not keep lines of it; T2V_PEER_WINDOW_MEM forces one); "" without a window */
const char* t2v_comm_window_kind(const t2v_comm* comm);
/* In-place all-gather outside a plan, on `stream`: part q of t2v_comm_size(comm) equal parts of `bytes` bytes lives at base + q*bytes,
 * the caller's own part is in place.  The per-step exchanges around the UNet — the eps of a classifier-free-guidance pair
 * (gaussian_sampler.py:161-163 evaluates the two forwards one after the other), the uint8 frames of the decoded clip
 * (lvdm/utils/dist_utils.py:13-19) — use it, so a sharded run's data path has ONE collective stack. */
int t2v_comm_all_gather(t2v_comm* comm, void* base, uint64_t bytes, void* stream);

/* Drop-in entry points for the reference's call sites (thin wrappers over t2v_plan_run that
 * fix the external-slot convention):
 *   t2v_unet_forward  <->  eps = UNetSD.forward(x, t, y)      t2v_model.py:386
 *   t2v_vae_decode    <->  img = AutoencoderKL.decode(z)      t2v_model.py:1646
 *   t2v_ddim_step     <->  loop body of GaussianDiffusion.sample  gaussian_sampler.py:257-283 */
int t2v_unet_forward(t2v_plan* plan, const void* x, const float* t, const void* ctx, void* eps_out,
                     void* stream);
int t2v_vae_decode(t2v_plan* plan, const void* z, void* img_out, void* stream);
int t2v_ddim_step(t2v_plan* plan, const void* xt, const void* eps_pair, const void* noise,
                  void* xt_out, const float coef[6], void* stream);

#ifdef __cplusplus
}
#endif
#endif /* T2V_HIP_H */
