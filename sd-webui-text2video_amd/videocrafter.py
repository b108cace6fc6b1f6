"""VideoCrafter (LVDM) hot path — drop-in for the denoiser, sampler and latent-diffusion wrapper of
reference scripts/videocrafter (BASELINE.json configs[4]):

  UNetModel                  lvdm/models/modules/openaimodel3d.py:310-670   (same ctor keywords, same
                             622-tensor state-dict key set incl. `attn1_tmp.relative_position_k.embeddings_table`)
  DDIMSampler                lvdm/samplers/ddim.py:9-279
  LatentDiffusion (subset)   lvdm/models/ddpm3d.py: apply_model :849-865, decode_first_stage(_2DAE) :776-793,
                             register_schedule buffers
  sample_text2video          videocrafter/sample_text2video.py:92-152

As for ModelScope (unet.py) the nn.Module tree only holds parameters under the reference's names;
`forward` lowers the network to a denoise program executed by the HIP kernels of libt2v_hip.so:
Conv3d (1,3,3) -> implicit-GEMM conv3x3 per frame, GroupNorm32 / Normalize over (c/g, t, h, w) ->
cross-frame GroupNorm, spatial self / text cross attention with 8 heads of C/8 = 40 / 80 / 160 channels ->
the MFMA flash-attention kernel, TemporalCrossAttention with relative-position terms -> T2V_OP_RELPOS_ATTN.
"""
from __future__ import annotations

import math
from functools import partial
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import packing as pk
from .program import Buf, Program, Ref
from .unet import UNetSD, _Compiled, _Lowering, _attn_params


# ------------------------------------------------------------------------------------------
# topology (openaimodel3d.py:411-589)
# ------------------------------------------------------------------------------------------
def lvdm_layout(model_channels, channel_mult, num_res_blocks, attention_resolutions):
    """-> (inputs, middle, outputs, last_ch): lists of (prefix, [(kind, cin, cout), ...]); part j of a block
    lives under '<prefix>.<j>'.  kinds: stem | res | st | down | up."""
    mc = model_channels
    inputs = [("input_blocks.0", [("stem", None, mc)])]
    chans = [mc]
    ch, ds, idx = mc, 1, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            parts = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in attention_resolutions:
                parts.append(("st", ch, ch))
            inputs.append((f"input_blocks.{idx}", parts))
            idx += 1
            chans.append(ch)
        if level != len(channel_mult) - 1:
            inputs.append((f"input_blocks.{idx}", [("down", ch, ch)]))
            idx += 1
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch), ("st", ch, ch), ("res", ch, ch)]
    outputs = []
    oidx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            parts = [("res", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in attention_resolutions:
                parts.append(("st", ch, ch))
            if level and i == num_res_blocks:
                parts.append(("up", ch, ch))
                ds //= 2
            outputs.append((f"output_blocks.{oidx}", parts))
            oidx += 1
    return inputs, middle, outputs, ch


# ------------------------------------------------------------------------------------------
# parameter containers (reference key names)
# ------------------------------------------------------------------------------------------
def _conv133(cin, cout):
    return nn.Conv3d(cin, cout, (1, 3, 3), padding=(0, 1, 1))


def _lvdm_res_params(cin, emb, cout):
    m = nn.Module()
    m.in_layers = nn.Sequential(nn.GroupNorm(32, cin), nn.SiLU(), _conv133(cin, cout))
    m.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb, cout))
    m.out_layers = nn.Sequential(nn.GroupNorm(32, cout), nn.SiLU(), nn.Dropout(0.0), _conv133(cout, cout))
    m.skip_connection = nn.Identity() if cin == cout else nn.Conv3d(cin, cout, 1)
    return m


def _rel_pos_params(d_head, max_rel):
    m = nn.Module()
    m.embeddings_table = nn.Parameter(torch.empty(2 * max_rel + 1, d_head))
    nn.init.xavier_uniform_(m.embeddings_table)
    return m


def _temporal_attn_params(dim, heads, d_head, temporal_length):
    m = _attn_params(dim, None, heads, d_head)
    m.relative_position_k = _rel_pos_params(d_head, temporal_length)
    m.relative_position_v = _rel_pos_params(d_head, temporal_length)
    return m


def _st_block_params(dim, heads, d_head, context_dim, temporal_length):
    m = nn.Module()
    m.attn1 = _attn_params(dim, None, heads, d_head)
    m.attn2 = _attn_params(dim, context_dim, heads, d_head)
    ff = nn.Module()
    geglu = nn.Module()
    geglu.proj = nn.Linear(dim, dim * 4 * 2)
    ff.net = nn.Sequential(geglu, nn.Dropout(0.0), nn.Linear(dim * 4, dim))
    m.ff = ff
    m.norm1, m.norm2, m.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
    m.attn1_tmp = _temporal_attn_params(dim, heads, d_head, temporal_length)
    m.attn2_tmp = _temporal_attn_params(dim, heads, d_head, temporal_length)
    m.norm4, m.norm5 = nn.LayerNorm(dim), nn.LayerNorm(dim)
    return m


def _st_transformer_params(channels, heads, d_head, context_dim, temporal_length):
    inner = heads * d_head
    m = nn.Module()
    m.norm = nn.GroupNorm(32, channels, eps=1e-6, affine=True)
    m.proj_in = nn.Conv3d(channels, inner, 1)
    m.transformer_blocks = nn.ModuleList([_st_block_params(inner, heads, d_head, context_dim, temporal_length)])
    m.proj_out = nn.Conv3d(inner, channels, 1)
    return m


def _holder(attr, mod):
    m = nn.Module()
    setattr(m, attr, mod)
    return m


class UNetModel(UNetSD):
    """LVDM 3-D UNet (openaimodel3d.py:310).  Supported configuration = the released VideoCrafter base model
    (base_t2v/model_config.yaml): dims=3, kernel_size_t=1, transformer_depth=1, SpatialTemporalTransformer with
    relative positions, no class / fps conditioning, no scale-shift norm, conv resampling."""

    supports_cfg_batch = True

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=3, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, transformer_depth=1, context_dim=None, legacy=True, kernel_size_t=1, padding_t=1,
                 use_temporal_transformer=True, temporal_length=None, use_relative_position=False,
                 cross_attn_on_tempoal=False, temporal_crossattn_type="crossattn", order="stst", nonlinearity_type="silu",
                 temporalcrossfirst=False, split_stcontext=False, temporal_context_dim=None, use_tempoal_causal_attn=False,
                 ST_transformer_module="attention_temporal", ST_transformer_class="SpatialTemporalTransformer",
                 init_weights=True, **kwargs):
        nn.Module.__init__(self)
        unsupported = [("num_classes", num_classes is not None), ("use_scale_shift_norm", use_scale_shift_norm),
                       ("resblock_updown", resblock_updown), ("kernel_size_t != 1", kernel_size_t != 1),
                       ("transformer_depth != 1", transformer_depth != 1), ("dims != 3", dims != 3),
                       ("conv_resample=False", not conv_resample), ("num_head_channels", num_head_channels != -1),
                       ("use_relative_position=False", not use_relative_position),
                       ("cross_attn_on_tempoal", cross_attn_on_tempoal), ("use_tempoal_causal_attn", use_tempoal_causal_attn),
                       ("nonlinearity_type", nonlinearity_type != "silu"),
                       ("ST_transformer_class", ST_transformer_class != "SpatialTemporalTransformer")]
        bad = [n for n, b in unsupported if b]
        if bad or num_heads <= 0 or temporal_length is None or context_dim is None:
            raise NotImplementedError(f"UNetModel: configuration outside the built hot path: {bad or 'num_heads/temporal_length/context_dim'}")
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, list(attention_resolutions)
        self.channel_mult, self.num_heads, self.temporal_length = list(channel_mult), num_heads, temporal_length
        self.use_relative_position, self.dtype = True, torch.float32
        self.time_embed_dim = model_channels * 4
        # names the shared runtime (UNetSD.forward) reads
        self.in_dim, self.out_dim, self.dim, self.embed_dim = in_channels, out_channels, model_channels, model_channels * 4
        self.context_dim = context_dim[0] if isinstance(context_dim, (list, tuple)) else context_dim
        self.parameterization, self.v_posterior = "eps", 0
        for ch in {model_channels * m for m in channel_mult}:
            if ch // num_heads not in (40, 64, 80, 160):
                raise NotImplementedError(f"attention head_dim {ch // num_heads} (kernels: 40, 64, 80, 160)")
        self._layout = lvdm_layout(model_channels, self.channel_mult, num_res_blocks, self.attention_resolutions)
        inputs, middle, outputs, last = self._layout
        emb = self.time_embed_dim

        def make(kind, cin, cout):
            if kind == "stem":
                return _conv133(in_channels, cout)
            if kind == "res":
                return _lvdm_res_params(cin, emb, cout)
            if kind == "st":
                return _st_transformer_params(cout, num_heads, cout // num_heads, self.context_dim, temporal_length)
            if kind == "down":
                return _holder("op", nn.Conv3d(cin, cout, (1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
            if kind == "up":
                return _holder("conv", _conv133(cin, cout))
            raise ValueError(kind)

        ctx = torch.device("meta") if not init_weights else torch.device("cpu")
        with ctx:
            self.time_embed = nn.Sequential(nn.Linear(model_channels, emb), nn.SiLU(), nn.Linear(emb, emb))
            self.input_blocks = nn.ModuleList([nn.ModuleList([make(*p) for p in parts]) for _, parts in inputs])
            self.middle_block = nn.ModuleList([make(*p) for p in middle])
            self.output_blocks = nn.ModuleList([nn.ModuleList([make(*p) for p in parts]) for _, parts in outputs])
            self.out = nn.Sequential(nn.GroupNorm(32, last), nn.SiLU(), _conv133(model_channels, out_channels))
        if not init_weights:
            self.to_empty(device="cpu")
        else:
            self._zero_init()
        self._init_runtime()
        # Measured on this model (round 4, 16 f @256x256, deployed-weights golden): the attention-output and resample-cast splits move the
        # 10-step output 1.077e-3 -> 0.92e-3 but the 50-step output only 1.069e-3 -> 1.035e-3, for +0.3 / +0.6 ms on a 17.9 ms step — the
        # 50-step error of this model is not an operand-rounding term (DESIGN.md "Precision") — so they stay opt-in here too
        # (T2V_PRECISE_ATTN=1 / T2V_PRECISE_RESAMPLE=1, inherited defaults of UNetSD).

    def _zero_init(self):
        """zero_module(...) sites of the reference: ResBlock out conv (:209-213), transformer proj_out
        (attention_temporal.py:375-379), the q/k/v/out projections of the temporal attentions (:96-100), head conv."""
        with torch.no_grad():
            for n, m in self.named_modules():
                if n.endswith("out_layers.3") or n.endswith(".proj_out") or n == "out.2":
                    for p in m.parameters():
                        p.zero_()
                if n.endswith("attn1_tmp") or n.endswith("attn2_tmp"):
                    for q in (m.to_q, m.to_k, m.to_v, m.to_out[0]):
                        for p in q.parameters():
                            p.zero_()

    def register_schedule(self, *a, **k):
        raise AttributeError("the LVDM schedule lives on LatentDiffusion, not on the UNet")

    # ---- forward (openaimodel3d.py:632-670) ------------------------------------------------------
    def forward(self, x, timesteps=None, time_emb_replace=None, context=None, features_adapter=None, y=None, **kwargs):
        if time_emb_replace is not None or features_adapter is not None or y is not None:
            raise NotImplementedError("time_emb_replace / adapter features / class labels are not on the hot path")
        return UNetSD.forward(self, x, timesteps, context)

    def _compile(self, B, F, H, W, Lctx, x_dt, out_dt, ctx_dt="f32", shard=None, x_batch=0):
        """shard (program.TShardSpec): the clip's frames are split over a T group.  Every GroupNorm32 of this UNet spans
        all frames (all-gather of statistics partials) and the temporal attentions gather K/V along T; the (1,3,3)
        convolutions, spatial / text attention and feed-forward are frame-local (no halo exchange: kernel_size_t = 1)."""
        low = _LvdmLowering(self, B, F, H, W, Lctx, x_dt, out_dt, ctx_dt, keep_taps=self.debug_taps, shard=shard, x_batch=x_batch)
        return _Compiled(low.build(), low.packer)


# ------------------------------------------------------------------------------------------
# lowering
# ------------------------------------------------------------------------------------------
class _LvdmLowering(_Lowering):
    def w_conv133(self, key, cin_pad=0) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":c133", "f16", lambda sd, k=key, c=cin_pad:
                                                pk.pad_rows(pk.conv3x3(sd[k + ".weight"][:, :, 0], c))))

    def w_conv133_dup(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":c133d", "f16", lambda sd, k=key: pk.pad_rows(pk.conv3x3_c8_dup(sd[k + ".weight"][:, :, 0]))))

    def table(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":tab", "f32", lambda sd, k=key: sd[k]))

    def table16(self, key, frames: int, transposed: bool) -> Optional[Ref]:
        """The same table packed for the persistent MFMA kernel (whole clips of <= 16 frames, no clipping of s - t): packing.relpos_table16."""
        if frames > 16 or self.net.temporal_length < frames - 1:
            return None
        return Ref("weight", 0, self.packer.add(f"{key}:tab16{'T' if transposed else ''}.{frames}", "f16",
                                                lambda sd, k=key, f=frames, t=transposed: pk.relpos_table16(sd[k].float(), f, t)))

    def w_conv133_hilo(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":c133hl", "f16", lambda sd, k=key: pk.pad_rows(pk.conv3x3(torch.cat([sd[k + ".weight"][:, :, 0]] * 2, dim=1)))))

    def conv133(self, name, a: Buf, key, cout, h, w, *, stride=1, up=0, out_dtype="f32", rowbias=None, residual=None, cin=None,
                dest: Optional[Buf] = None, dup_c8: bool = False, stats: Optional[Buf] = None, hilo: bool = False) -> Buf:
        cin = a.cols if cin is None else cin
        ho, wo = (h * 2, w * 2) if up else ((h + 1) // 2 if stride == 2 else h, (w + 1) // 2 if stride == 2 else w)
        n = (cout + 3) // 4 * 4
        out = self._dest(dest, self.Bc * self.F * ho * wo, n, out_dtype)
        gather = L.GATHER_CONV3X3_C8 if cin == 8 else L.GATHER_CONV3X3
        wref = self.w_conv133_hilo(key) if hilo else (self.w_conv133_dup(key) if dup_c8 else self.w_conv133(key, 8 if cin == 8 else 0))
        op = self.P.gemm(name, a, wref, n, 9 * cin, out, bias=self.vec(key + ".bias"),
                         gather=gather, conv=dict(Hin=h, Win=w, Cin=cin, stride=stride, up=up, Hout=ho, Wout=wo),
                         rowbias=rowbias, rows_per_batch=self.F * ho * wo if rowbias is not None else 0, residual=residual, stats=stats,
                         k_alg=9 * cin // 2 if hilo else None)
        self.last_stats = stats if (stats is not None and op.meta.get("stats")) else None
        return out

    def res_block(self, prefix, x: Buf, cin, cout, h, w, dest: Optional[Buf] = None) -> Buf:
        """ResBlock._forward (openaimodel3d.py:244-271): GroupNorm32 statistics span all frames of a sample."""
        P = self.P
        a = self.gn(prefix + ".in_layers.0", x, prefix + ".in_layers.0", per_frame=False, eps=1e-5, silu=True)
        e0, e1 = self.emb_slices[prefix]
        st = self.strips_for(x.rows, cout, self.F * h * w)       # column statistics from the conv's epilogue for the norm that follows (unet.py)
        h1 = self.conv133(prefix + ".in_layers.2", a, prefix + ".in_layers.2", cout, h, w,
                          rowbias=self.emb_out.col_slice(e0, e1), out_dtype=self.net.norm_input_dtype, stats=st)
        P.free(a)
        b = self.gn(prefix + ".out_layers.0", h1, prefix + ".out_layers.0", per_frame=False, eps=1e-5, silu=True, stats=self.last_stats, x_dead=True)
        P.free(h1, st)
        if cin != cout:
            skip = P.alloc(x.rows, cout, "f32")
            if self.precise:         # hi + lo operand split in one pass (UNetSD.precise_operands): rows [hi | lo], weights [W | W]
                x16 = P.alloc(x.rows, 2 * cin, "f16")
                P.copy2d(prefix + ".skip.cast", x, x16.col_slice(0, cin), lo=x16.col_slice(cin, 2 * cin))
                P.gemm(prefix + ".skip_connection", x16, self.w_linear_dup(prefix + ".skip_connection"), cout, 2 * cin, skip,
                       bias=self.vec(prefix + ".skip_connection.bias"), k_alg=cin)
            else:
                x16 = P.alloc(x.rows, cin, "f16")
                P.copy2d(prefix + ".skip.cast", x, x16)
                P.gemm(prefix + ".skip_connection", x16, self.w_linear(prefix + ".skip_connection"), cout, cin, skip,
                       bias=self.vec(prefix + ".skip_connection.bias"))
            P.free(x16)
        else:
            skip = x
        out = self.conv133(prefix + ".out_layers.3", b, prefix + ".out_layers.3", cout, h, w, residual=skip, dest=dest)
        P.free(b)
        if skip is not x:
            P.free(skip)
        return out

    def st_transformer(self, prefix, x: Buf, c, h, w, dest: Optional[Buf] = None) -> Buf:
        """SpatialTemporalTransformer.forward + BasicTransformerBlockST._forward (attention_temporal.py:301-335,
        386-399): s-self, t-self (rel-pos), s-cross (text), t-self (rel-pos), GEGLU feed-forward."""
        P, net, B, F, hw = self.P, self.net, self.Bc, self.F, h * w
        heads, d = net.num_heads, c // net.num_heads
        scale = d ** -0.5
        M = x.rows
        n = self.gn(prefix + ".norm", x, prefix + ".norm", per_frame=False, eps=1e-6, silu=False, lo=self.precise_at(self.precise_gn, h, w))
        tb = prefix + ".transformer_blocks.0"
        # LayerNorms as a second output of the GEMM that produces their input (fused into the epilogue of the 192x320 tile where
        # the tile holds whole rows, C = 320: the 32x32 level; a separate LayerNorm op elsewhere — Program.gemm decides).  The
        # T-sharded lowering keeps explicit LayerNorm ops (its temporal attentions normalise inside their K/V-gather form).
        fuse = self.shard is None

        def ln_of(tag) -> Optional[tuple]:
            return self.ln_arg(f"{tb}.{tag}", P.alloc(M, c, "f16")) if fuse else None

        def layer_norm(tag, src: Buf) -> Buf:
            o = P.alloc(M, c, "f16")
            P.layernorm(f"{tb}.{tag}", src, self.vec(f"{tb}.{tag}.weight"), self.vec(f"{tb}.{tag}.bias"), o)
            return o

        cur = P.alloc(M, c, "f32")
        ln = ln_of("norm1")
        P.gemm(prefix + ".proj_in", n, self.w_proj(prefix + ".proj_in", n.cols // c), c, n.cols, cur, bias=self.vec(prefix + ".proj_in.bias"), ln=ln,
               k_alg=c)
        P.free(n)
        nxt = ln[3] if ln is not None else None          # LayerNorm(cur) for the next consumer, when already produced

        attn_lo = self.precise_at(self.precise_attn, h, w)     # attention outputs as rows [hi | lo], to_out on K = 2c against [W | W]

        def attn_out() -> Buf:
            return P.alloc(M, 2 * c if attn_lo else c, "f16")

        def out_proj(attn, a: Buf, res: Buf, next_norm: str, res_wrap: int = 0):
            o = P.alloc(M, c, "f32")
            ln = ln_of(next_norm)
            P.gemm(f"{tb}.{attn}.to_out", a, self.w_proj(f"{tb}.{attn}.to_out.0", a.cols // c), c, a.cols, o,
                   bias=self.vec(f"{tb}.{attn}.to_out.0.bias"), residual=res, ln=ln, k_alg=c, res_wrap=res_wrap)
            P.free(a, res)
            return o, (ln[3] if ln is not None else None)

        def temporal_attn_sharded(attn, norm, src: Buf, next_norm: str):
            """Queries = this rank's frames, keys / values = all frames of the clip (K/V projections all-gathered along T);
            the relative position of key s to local query t is s - (t + first frame of this slice)."""
            sh = self.shard
            Mmax, Ftot = sh.max_frames * hw, sh.total
            nrm = layer_norm(norm, src)
            q = P.alloc(M, c, "f16")
            P.gemm(f"{tb}.{attn}.to_q", nrm, self.w_linear(f"{tb}.{attn}.to_q"), c, c, q)
            kv_all = P.alloc(sh.size * Mmax, 2 * c, "f16")
            mine = kv_all.row_slice(sh.index * Mmax, sh.index * Mmax + M)
            P.gemm(f"{tb}.{attn}.kv", nrm, self.w_kv(f"{tb}.{attn}"), 2 * c, c, mine)
            P.free(nrm)
            full = Buf(kv_all.ref, sh.size * Mmax * 2 * c * 2, 1, 1, "u8", kv_all.alloc_off)
            P.allgather(f"{tb}.{attn}.kv.allgather", full, Mmax * 2 * c * 2, sh)
            a = attn_out()
            ldk, lo = 2 * c, a.ld
            P.attention(f"{tb}.{attn}", q.ref, kv_all.col_slice(0, c).ref, kv_all.col_slice(c, 2 * c).ref, a.ref, out_buf=a,
                        nq=F, nk=Ftot, heads=heads, b_outer=1, b_inner=hw, q_strides=(hw * c, 0, c), kv_strides=(hw * ldk, 0, ldk),
                        o_strides=(hw * lo, 0, lo), scale=scale, head_dim=d, lo_off=c if attn_lo else 0,
                        rel_k=self.table(f"{tb}.{attn}.relative_position_k.embeddings_table"),
                        rel_v=self.table(f"{tb}.{attn}.relative_position_v.embeddings_table"),
                        max_rel=net.temporal_length, q_offset=sh.offset)
            P.free(q, kv_all)
            return out_proj(attn, a, src, next_norm)

        def self_attn(attn, norm, src: Buf, nrm: Optional[Buf], temporal: bool, next_norm: str):
            if temporal and self.shard is not None:
                return temporal_attn_sharded(attn, norm, src, next_norm)
            nrm = layer_norm(norm, src) if nrm is None else nrm
            qkv = P.alloc(M, 3 * c, "f16")
            P.gemm(f"{tb}.{attn}.qkv", nrm, self.w_qkv(f"{tb}.{attn}"), 3 * c, c, qkv)
            P.free(nrm)
            a = attn_out()
            ld, lo = 3 * c, a.ld
            q, k, v = qkv.col_slice(0, c), qkv.col_slice(c, 2 * c), qkv.col_slice(2 * c, 3 * c)
            if not temporal:
                P.attention(f"{tb}.{attn}", q.ref, k.ref, v.ref, a.ref, out_buf=a, nq=hw, nk=hw, heads=heads, b_outer=B * F,
                            b_inner=1, q_strides=(ld, hw * ld, 0), kv_strides=(ld, hw * ld, 0), o_strides=(lo, hw * lo, 0),
                            scale=scale, head_dim=d, lo_off=c if attn_lo else 0)
            else:
                P.attention(f"{tb}.{attn}", q.ref, k.ref, v.ref, a.ref, out_buf=a, nq=F, nk=F, heads=heads, b_outer=B,
                            b_inner=hw, q_strides=(hw * ld, F * hw * ld, ld), kv_strides=(hw * ld, F * hw * ld, ld),
                            o_strides=(hw * lo, F * hw * lo, lo), scale=scale, head_dim=d, lo_off=c if attn_lo else 0,
                            rel_k=self.table(f"{tb}.{attn}.relative_position_k.embeddings_table"),
                            rel_v=self.table(f"{tb}.{attn}.relative_position_v.embeddings_table"),
                            rel_k16=self.table16(f"{tb}.{attn}.relative_position_k.embeddings_table", F, False),
                            rel_vT16=self.table16(f"{tb}.{attn}.relative_position_v.embeddings_table", F, True),
                            max_rel=net.temporal_length)
            P.free(qkv)
            return out_proj(attn, a, src, next_norm)

        cur, nxt = self_attn("attn1", "norm1", cur, nxt, temporal=False, next_norm="norm4")
        cur, nxt = self_attn("attn1_tmp", "norm4", cur, nxt, temporal=True, next_norm="norm2")
        # text cross-attention: K/V of all transformers come from ONE projection GEMM of the context
        nrm = layer_norm("norm2", cur) if nxt is None else nxt
        q = P.alloc(M, c, "f16")
        P.gemm(f"{tb}.attn2.to_q", nrm, self.w_linear(f"{tb}.attn2.to_q"), c, c, q)
        P.free(nrm)
        k0, k1 = self.kv_slices[tb + ".attn2"]
        kv = self.kv_all
        # cond and uncond part at the text cross-attention (unet.py transformer_block): q / cur hold ONE sample's rows while the prefix
        # is shared — q is read with a zero sample stride, to_out adds `cur` through the residual row wrap, M becomes B samples' rows
        parting = self.sharing
        q_b_stride, shared_rows = (0, M) if parting else (F * hw * c, 0)
        if parting:
            self.sharing, self.Bc = False, self.B
            B, M = self.B, self.B * M
        a = attn_out()
        Lc, lo = self.Lctx, a.ld
        P.attention(f"{tb}.attn2", q.ref, kv.col_slice(k0, k0 + c).ref, kv.col_slice(k0 + c, k1).ref, a.ref, out_buf=a, nq=hw,
                    nk=Lc, heads=heads, b_outer=B, b_inner=F, q_strides=(c, q_b_stride, hw * c), kv_strides=(kv.ld, Lc * kv.ld, 0),
                    o_strides=(lo, F * hw * lo, hw * lo), scale=scale, head_dim=d, lo_off=c if attn_lo else 0)
        P.free(q)
        cur, nxt = out_proj("attn2", a, cur, "norm5", res_wrap=shared_rows)
        cur, nxt = self_attn("attn2_tmp", "norm5", cur, nxt, temporal=True, next_norm="norm3")
        nrm = layer_norm("norm3", cur) if nxt is None else nxt
        wg, bg = self.w_geglu(f"{tb}.ff.net.0.proj")
        g = P.alloc(M, 4 * c, "f16")
        P.gemm(f"{tb}.ff.geglu", nrm, wg, 8 * c, c, g, bias=bg, epi=L.EPI_GEGLU)
        P.free(nrm)
        if self.precise_at(self.precise_ff, h, w):          # x4 as rows [hi | lo], proj_out against [W | W] (unet.py transformer_block)
            x4 = P.alloc(M, 2 * c, "f16")
            P.gemm(f"{tb}.ff.net.2", g, self.w_linear(f"{tb}.ff.net.2"), c, 4 * c, x4.col_slice(0, c), bias=self.vec(f"{tb}.ff.net.2.bias"),
                   residual=cur, out_lo=True)
        else:
            x4 = P.alloc(M, c, "f16")
            P.gemm(f"{tb}.ff.net.2", g, self.w_linear(f"{tb}.ff.net.2"), c, 4 * c, x4, bias=self.vec(f"{tb}.ff.net.2.bias"), residual=cur)
        P.free(g, cur)
        out = self._dest(dest, M, c, "f32")
        P.gemm(prefix + ".proj_out", x4, self.w_proj(prefix + ".proj_out", x4.cols // c), c, x4.cols, out, bias=self.vec(prefix + ".proj_out.bias"),
               residual=x, k_alg=c, res_wrap=x.rows if x.rows != M else 0)
        P.free(x4)
        return out

    def build(self) -> Program:
        net, P, B, F = self.net, self.P, self.B, self.F
        inputs, middle, outputs, last = net._layout
        mc, emb = net.model_channels, net.time_embed_dim
        h, w = self.H, self.W
        P.begin()
        if self.sharing and not any(kind == "st" for _, parts in inputs for kind, _, _ in parts):
            self.sharing, self.Bc = False, B
        blocks = [(f"{pre}.{j}", part) for pre, parts in inputs + outputs for j, part in enumerate(parts)] + \
                 [(f"middle_block.{j}", part) for j, part in enumerate(middle)]
        res_prefixes = [(p, part[2]) for p, part in blocks if part[0] == "res"]
        st_prefixes = [(p, part[2]) for p, part in blocks if part[0] == "st"]
        off = 0
        for p, cout in res_prefixes:
            self.emb_slices[p] = (off, off + cout)
            off += cout
        n_emb = off
        off = 0
        for p, c in st_prefixes:
            self.kv_slices[p + ".transformer_blocks.0.attn2"] = (off, off + 2 * c)
            off += 2 * c
        n_kv = off

        self.kv_all = P.alloc(B * self.Lctx, n_kv, "f16")     # first allocation, freed last: survives between runs (see unet.py)
        # ---- timestep embedding (util.py:142-162 cos|sin, base 10000) -> MLP; every ResBlock's emb projection in one GEMM
        freqs = Ref("weight", 0, self.packer.add("time_embed.freqs", "f32", lambda sd, half=mc // 2: torch.exp(
            -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)))
        te = P.alloc(B, mc, "f16")
        P.time_embed("time_embed.sincos", Ref("ext", L.EXT_T), freqs, te)
        e1 = P.alloc(B, emb, "f16")
        P.gemm("time_embed.0", te, self.w_linear("time_embed.0"), emb, mc, e1, bias=self.vec("time_embed.0.bias"), act=1)
        P.free(te)
        e_silu = P.alloc(B, emb, "f16")      # emb is consumed only through emb_layers = SiLU -> Linear
        P.gemm("time_embed.2", e1, self.w_linear("time_embed.2"), emb, emb, e_silu, bias=self.vec("time_embed.2.bias"), act=1)
        P.free(e1)
        ps = tuple(p for p, _ in res_prefixes)
        w_emb = Ref("weight", 0, self.packer.add("emb_all:lin", "f16", lambda sd, ps=ps: torch.cat([sd[p + ".emb_layers.1.weight"] for p in ps], dim=0)))
        b_emb = Ref("weight", 0, self.packer.add("emb_all:v", "f32", lambda sd, ps=ps: torch.cat([sd[p + ".emb_layers.1.bias"] for p in ps], dim=0)))
        self.emb_out = P.alloc(B, n_emb, "f32")
        P.gemm("emb_layers.all", e_silu, w_emb, n_emb, emb, self.emb_out, bias=b_emb)
        P.free(e_silu)

        ctx16 = P.alloc(B * self.Lctx, net.context_dim, "f16")
        P.copy2d("context.cast", Buf(Ref("ext", L.EXT_CTX), B * self.Lctx, net.context_dim, net.context_dim, self.ctx_dt),
                 ctx16).meta["step_invariant"] = True
        sts = tuple(p for p, _ in st_prefixes)
        w_kv = Ref("weight", 0, self.packer.add("kv_all:lin", "f16", lambda sd, sts=sts: torch.cat(
            [torch.cat([sd[p + ".transformer_blocks.0.attn2.to_k.weight"], sd[p + ".transformer_blocks.0.attn2.to_v.weight"]], dim=0)
             for p in sts], dim=0)))
        P.gemm("attn2.kv.all", ctx16, w_kv, n_kv, net.context_dim, self.kv_all).meta["step_invariant"] = True
        P.free(ctx16)

        xin = P.alloc(self.M(h, w), 8, "f16")
        self.stem_dup = self.precise and self.x_dt == "f32" and net.in_dim == 4
        P.ncthw_to_cl("x.to_tokens", Ref("ext", L.EXT_X), self.x_dt, xin, B=self.Bc, C=net.in_dim, F=F, HW=h * w,
                      src_batch=self.x_batch if self.x_batch != self.Bc else 0, lo_in_pad=self.stem_dup)

        def run_parts(prefix, parts, x, h, w, dest=None):
            for j, (kind, cin, cout) in enumerate(parts):
                p = f"{prefix}.{j}"
                d = dest if j == len(parts) - 1 else None
                spread = d if (self.sharing and d is not None and kind != "st") else None      # shared cond | uncond prefix: unet.py run_parts
                if spread is not None:
                    d = None
                if kind == "stem":
                    y = self.conv133(p, x, p, cout, h, w, cin=8, dest=d, dup_c8=self.stem_dup)
                elif kind == "res":
                    y = self.res_block(p, x, cin, cout, h, w, dest=d)
                elif kind == "st":
                    y = self.st_transformer(p, x, cout, h, w, dest=d)
                elif kind in ("down", "up"):
                    attr = "op" if kind == "down" else "conv"
                    if self.precise_rs and cin % 64 == 0:        # the cast as rows [hi | lo], the convolution against [W | W] (unet.py resample)
                        x16 = P.alloc(x.rows, 2 * cin, "f16")
                        P.copy2d(p + ".cast", x, x16.col_slice(0, cin), lo=x16.col_slice(cin, 2 * cin))
                        y = self.conv133(f"{p}.{attr}", x16, f"{p}.{attr}", cout, h, w, stride=2 if kind == "down" else 1,
                                         up=1 if kind == "up" else 0, dest=d, hilo=True)
                    else:
                        x16 = P.alloc(x.rows, cin, "f16")
                        P.copy2d(p + ".cast", x, x16)
                        y = self.conv133(f"{p}.{attr}", x16, f"{p}.{attr}", cout, h, w, stride=2 if kind == "down" else 1,
                                         up=1 if kind == "up" else 0, dest=d)
                    P.free(x16)
                    h, w = ((h + 1) // 2, (w + 1) // 2) if kind == "down" else (h * 2, w * 2)
                else:
                    raise ValueError(kind)
                if spread is not None:
                    assert spread.rows == self.B * y.rows and spread.cols == y.cols
                    for b in range(self.B):
                        P.copy2d(f"{p}.to_skip.{b}", y, spread.row_slice(b * y.rows, (b + 1) * y.rows))
                P.tap(p, y)
                P.free(x)              # borrowed windows of a concat buffer are ignored by free()
                x = y
            return x, h, w

        def out_hw(parts, h, w):
            for kind, _, _ in parts:
                if kind == "down":
                    h, w = (h + 1) // 2, (w + 1) // 2
                elif kind == "up":
                    h, w = h * 2, w * 2
            return h, w

        # `th.cat([h, hs.pop()], dim=1)` (openaimodel3d.py:665): both halves are written in place by their producers
        n_skip = len(inputs)
        cats: List[Buf] = []
        x = xin
        for k, (prefix, parts) in enumerate(inputs):
            sc = parts[-1][2]
            cin_total = outputs[n_skip - 1 - k][1][0][1]
            ho, wo = out_hw(parts, h, w)
            cat = P.alloc(self.B * self.F * ho * wo, cin_total, "f32")
            cats.append(cat)
            x, h, w = run_parts(prefix, parts, x, h, w, dest=cat.borrow_cols(cin_total - sc, cin_total))
        cat = cats.pop()
        x, h, w = run_parts("middle_block", middle, x, h, w, dest=cat.borrow_cols(0, cat.cols - inputs[-1][1][-1][2]))
        for j, (prefix, parts) in enumerate(outputs):
            nxt = cats.pop() if cats else None
            dest = nxt.borrow_cols(0, nxt.cols - inputs[n_skip - 2 - j][1][-1][2]) if nxt is not None else None
            x, h, w = run_parts(prefix, parts, cat, h, w, dest=dest)
            cat = nxt

        assert not self.sharing and self.Bc == B, "the shared cond | uncond prefix never reached a text cross-attention"
        a = self.gn("out.0", x, "out.0", per_frame=False, eps=1e-5, silu=True)
        P.free(x)
        y = self.conv133("out.2", a, "out.2", net.out_dim, h, w)
        P.free(a)
        P.cl_to_ncthw("eps.from_tokens", y, Ref("ext", L.EXT_OUT), self.out_dt, B=B, C=net.out_dim, F=F, HW=h * w)
        P.free(y, self.emb_out, self.kv_all)
        P.finish()
        return P


# ------------------------------------------------------------------------------------------
# latent-diffusion wrapper (the subset of lvdm/models/ddpm3d.py the sampling path touches)
# ------------------------------------------------------------------------------------------
class DiffusionWrapper(nn.Module):
    """ddpm3d.py:1362-1380 — only the 'crossattn' (released model) and unconditional keys are on the path."""

    def __init__(self, diffusion_model: UNetModel, conditioning_key="crossattn"):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, **kwargs):
        if self.conditioning_key != "crossattn":
            raise NotImplementedError(f"conditioning_key {self.conditioning_key!r}")
        return self.diffusion_model(x, t, context=torch.cat(c_crossattn, 1), **kwargs)


class LatentDiffusion(nn.Module):
    """State-dict layout of the released checkpoint for the parts on the path: 'model.diffusion_model.*'
    (UNetModel), 'first_stage_model.*' (AutoencoderKL) and the schedule buffers.  The text encoder
    ('cond_stage_model.*', FrozenCLIPEmbedder) is outside the hot path: pass any object with `.encode(prompts)`."""

    def __init__(self, unet_config: dict, first_stage_config: Optional[dict] = None, cond_stage_model=None, timesteps=1000,
                 linear_start=0.00085, linear_end=0.012, image_size=(32, 32), video_length=16, channels=4,
                 scale_factor=0.18215, shift_factor=0.0, conditioning_key="crossattn", init_weights=True, **ignored):
        super().__init__()
        from .vae import AutoencoderKL
        self.model = DiffusionWrapper(UNetModel(**unet_config, init_weights=init_weights), conditioning_key)
        self.first_stage_model = None
        if first_stage_config is not None:
            self.first_stage_model = AutoencoderKL(first_stage_config["ddconfig"], first_stage_config.get("embed_dim", 4),
                                                   init_weights=init_weights)
        self.cond_stage_model = cond_stage_model
        self.image_size, self.video_length, self.channels = image_size, video_length, channels
        self.scale_factor, self.shift_factor = scale_factor, shift_factor
        self.conditioning_key = conditioning_key
        self.parameterization, self.v_posterior, self.encoder_type = "eps", 0.0, "2d"
        self.cond_stage2_config = None
        self.register_schedule(timesteps=timesteps, linear_start=linear_start, linear_end=linear_end)

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2,
                          cosine_s=8e-3):
        """DDPM.register_schedule (ddpm3d.py:125-165): 'linear' = linspace(sqrt(s), sqrt(e), T, f64)^2, fp32 buffers."""
        if given_betas is not None:
            betas = np.asarray(given_betas, dtype=np.float64)
        elif beta_schedule == "linear":
            betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
        else:
            raise NotImplementedError(beta_schedule)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        t32 = partial(torch.tensor, dtype=torch.float32)
        for name, val in (("betas", betas), ("alphas_cumprod", ac), ("alphas_cumprod_prev", ac_prev),
                          ("sqrt_alphas_cumprod", np.sqrt(ac)), ("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - ac)),
                          ("log_one_minus_alphas_cumprod", np.log(1.0 - ac)), ("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / ac)),
                          ("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / ac - 1))):
            self.register_buffer(name, t32(val))
        self._schedule_names = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                                "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")

    def _apply(self, fn, *args, **kwargs):
        """`.half()` puts the NETWORKS in fp16; the schedule stays fp32 (the reference registers it as fp32 and never halves this
        module: ddpm3d.py:125-165, sample_text2video.py).  A schedule rounded to fp16 is a 3e-4 error of every DDIM coefficient that
        is the same for all pixels and does not average out over the steps — measured on configs[4]: 50-step output 1.06e-3 with the
        rounded schedule, 4.96e-4 with this one (10 steps: 1.08e-3 -> 9.4e-4)."""
        keep = {n: self._buffers[n] for n in getattr(self, "_schedule_names", ()) if self._buffers.get(n) is not None}
        super()._apply(fn, *args, **kwargs)
        for n, v in keep.items():
            cur = self._buffers[n]
            # only a cast to a LOWER-precision float is undone (.half() / .bfloat16()); .double() / .to(torch.float64) take effect
            if cur.dtype != v.dtype and cur.dtype in (torch.float16, torch.bfloat16):
                self._buffers[n] = v.to(cur.device)
        return self

    @property
    def device(self):
        return self.betas.device

    def get_learned_conditioning(self, c):
        """ddpm3d.py:647-658."""
        m = self.cond_stage_model
        if m is None:
            raise RuntimeError("no cond_stage_model (text encoder) attached")
        if hasattr(m, "encode") and callable(m.encode):
            return m.encode(c)
        return m(c)

    def apply_model(self, x_noisy, t, cond, return_ids=False, **kwargs):
        """ddpm3d.py:849-865."""
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        out = self.model(x_noisy, t, **cond, **kwargs)
        return out[0] if isinstance(out, tuple) and not return_ids else out

    @torch.no_grad()
    def decode(self, z, **kwargs):
        z = 1.0 / self.scale_factor * z - self.shift_factor
        return self.first_stage_model.decode(z)

    @torch.no_grad()
    def decode_first_stage_2DAE(self, z, decode_bs=16, return_cpu=True, **kwargs):
        """ddpm3d.py:776-788: frames decoded in chunks of decode_bs (None = all at once)."""
        b, _, t, _, _ = z.shape
        zf = z.permute(0, 2, 1, 3, 4).reshape(b * t, z.shape[1], z.shape[3], z.shape[4])
        chunks = [zf] if decode_bs is None else torch.split(zf, decode_bs, dim=0)
        outs = [self.decode(c) for c in chunks]
        res = torch.cat([o.cpu() for o in outs] if return_cpu else outs, dim=0)
        return res.reshape(b, t, *res.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()

    @torch.no_grad()
    def decode_first_stage(self, z, decode_bs=16, return_cpu=True, **kwargs):
        assert self.encoder_type == "2d" and z.dim() == 5
        return self.decode_first_stage_2DAE(z, decode_bs=decode_bs, return_cpu=return_cpu, **kwargs)


# ------------------------------------------------------------------------------------------
# DDIM sampler (lvdm/samplers/ddim.py)
# ------------------------------------------------------------------------------------------
class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0
        self.noise_gen = torch.Generator(device="cpu")      # seeded by process_videocrafter.py:70

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        """ddim.py:24-53 with make_ddim_timesteps / make_ddim_sampling_parameters (util.py:36-63) on fp32 cumprods."""
        if ddim_discretize != "uniform":
            raise NotImplementedError(ddim_discretize)
        T = self.ddpm_num_timesteps
        self.ddim_timesteps = np.asarray(list(range(0, T, T // ddim_num_steps))) + 1
        ac = self.model.alphas_cumprod.detach().cpu()
        assert ac.shape[0] == T, "alphas have to be defined for each timestep"
        alphas = ac[self.ddim_timesteps]
        alphas_prev = torch.cat([ac[0:1], ac[self.ddim_timesteps[:-1]]])
        a64, p64 = alphas.double(), alphas_prev.double()        # (util.py:52-63 does this in numpy float64 on the fp32 values)
        self.ddim_sigmas = ddim_eta * torch.sqrt((1 - p64) / (1 - a64) * (1 - a64 / p64))
        self.ddim_alphas, self.ddim_alphas_prev = alphas, alphas_prev
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - alphas)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, img_callback=None, quantize_x0=False, eta=0.0,
               mask=None, x0=None, temperature=1.0, noise_dropout=0.0, score_corrector=None, corrector_kwargs=None,
               verbose=True, schedule_verbose=False, x_T=None, log_every_t=100, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, postprocess_fn=None, sample_noise=None, cond_fn=None, **kwargs):
        """ddim.py:56-132 -> (samples, intermediates)."""
        if mask is not None or quantize_x0 or noise_dropout or score_corrector is not None or postprocess_fn is not None \
                or cond_fn is not None or kwargs.get("uc_type") is not None:
            raise NotImplementedError("mask blending / quantisation / noise dropout / score correctors are not on the hot path")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=schedule_verbose)
        size = (batch_size, *shape)
        assert len(size) == 5
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback, temperature=temperature,
                                  x_T=x_T, log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, sample_noise=sample_noise, verbose=verbose)

    @staticmethod
    def _ctx(c):
        if isinstance(c, dict):
            return torch.cat(c["c_crossattn"], 1)
        if isinstance(c, list):
            return torch.cat(c, 1)
        return c

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100, temperature=1.0,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, sample_noise=None, verbose=True, **kw):
        """ddim.py:135-206, p_sample_ddim :209-279 fused into one kernel launch per step (T2V_OP_DDIM_STEP mode 1)."""
        from . import samplers as S
        device = self.model.betas.device
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device)
        img = img.float().contiguous().clone()
        timesteps = self.ddim_timesteps
        total_steps = timesteps.shape[0]
        intermediates = {"x_inter": [img.clone()], "pred_x0": [img.clone()]}
        c, uc = self._ctx(cond), self._ctx(unconditional_conditioning) if unconditional_conditioning is not None else None
        guide = unconditional_guidance_scale
        guided = uc is not None and guide != 1.0
        unet = self.model.model.diffusion_model
        unet.refresh_weights(device)
        prev_auto, unet.auto_refresh = unet.auto_refresh, False
        prev_eps = S._want_fp32_eps(unet)            # eps in fp32 inside the loop: the CFG combination amplifies fp16 output roundings
        nxt = torch.empty_like(img)
        nb, C = img.shape[0], img.shape[1]          # nb videos per batch (sample_text2video's batch_size)
        f32 = torch.float32
        S.state.sampling_steps = total_steps
        iterator = np.flip(timesteps)
        if verbose and S.tqdm is not None:
            iterator = S.tqdm(iterator, desc="DDIM Sampler", total=total_steps)
        try:
            for i, step in enumerate(iterator):
                S.state.sampling_step = i
                if S.state.interrupted:
                    raise S.InterruptedException
                index = total_steps - i - 1
                ts = torch.full((nb,), int(step), device=device, dtype=torch.long)
                if guided:
                    # ONE x_t for the [cond | uncond] pair (the reference builds torch.cat([x] * 2), lvdm/samplers/ddim.py:206-209): the
                    # entry op reads it for both samples, and with UNetSD.share_cfg_prefix every op up to the first text
                    # cross-attention is computed once
                    xin = img if hasattr(unet, "share_cfg_prefix") else torch.cat([img, img])      # (a foreign model gets the reference's batch)
                    if hasattr(unet, "single_timestep"):
                        unet.single_timestep = True        # [ts | ts]: one timestep for the pair -> the prefix may be shared
                    eps = self.model.apply_model(xin, torch.cat([ts, ts]), torch.cat([c, uc])).contiguous()
                else:
                    eps = self.model.apply_model(img, ts, c).contiguous()
                a_t, a_prev = self.ddim_alphas[index].to(f32), self.ddim_alphas_prev[index].to(f32)
                sigma_t, s1m = self.ddim_sigmas[index].to(f32), self.ddim_sqrt_one_minus_alphas[index].to(f32)
                coef = [float(s1m), float(a_t.sqrt()), float(a_prev.sqrt()), float((1.0 - a_prev - sigma_t ** 2).sqrt()),
                        float(sigma_t) * float(temperature), float(guide) if guided else 1.0]
                if sample_noise is None:      # noise_like(..., noise_gen): CPU generator -> same stream on any device
                    noise = torch.randn(tuple(img.shape), generator=self.noise_gen).to(device)
                else:
                    noise = sample_noise.to(device=device, dtype=f32).contiguous()
                want_x0 = img_callback is not None or index % log_every_t == 0 or index == total_steps - 1
                if want_x0:                   # pred_x0 = (x - sqrt(1-a_t) e) / sqrt(a_t), e = u + g (c - u)
                    px0 = torch.empty_like(img)
                    ia = 1.0 / coef[1]
                    if guided:
                        g = float(guide)
                        S._lincomb(px0, [(ia, img), (-coef[0] * g * ia, eps[0:nb]), (-coef[0] * (1.0 - g) * ia, eps[nb:2 * nb])])
                    else:
                        S._lincomb(px0, [(ia, img), (-coef[0] * ia, eps[0:nb])])
                S._ddim_update(nxt, img, eps, noise, coef, C if guided else 0, mode=1)
                img, nxt = nxt, img
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(px0, i)
                if index % log_every_t == 0 or index == total_steps - 1:
                    intermediates["x_inter"].append(img.clone())
                    intermediates["pred_x0"].append(px0)
                if S.state.skipped:
                    break
        finally:
            unet.auto_refresh = prev_auto
            S._restore_eps(unet, prev_eps)
        return img, intermediates


# ------------------------------------------------------------------------------------------
# entry point (videocrafter/sample_text2video.py:92-152, sample_utils.py)
# ------------------------------------------------------------------------------------------
def get_conditions(prompts, model, batch_size):
    if isinstance(prompts, (str, int)):
        prompts = [prompts]
    if len(prompts) == 1:
        prompts = prompts * batch_size
    assert len(prompts) == batch_size, f"invalid prompts length: {len(prompts)}"
    return {"c_crossattn": [model.get_learned_conditioning(prompts)]}


def make_model_input_shape(model, batch_size, T=None):
    image_size = [model.image_size, model.image_size] if isinstance(model.image_size, int) else list(model.image_size)
    unet = model.model.diffusion_model
    return [batch_size, unet.in_channels, unet.temporal_length if T is None else T, *image_size]


def torch_to_np(x):
    """sample_utils.py:104-114: ((x + 1) * 127.5).clamp(0, 255) -> uint8, channels last."""
    sample = ((x.detach().float().cpu() + 1) * 127.5).clamp(0, 255).to(torch.uint8)
    return (sample.permute(0, 2, 3, 4, 1) if sample.dim() == 5 else sample.permute(0, 2, 3, 1)).contiguous()


@torch.no_grad()
def sample_text2video(model, prompt, n_prompt, n_samples, batch_size, sample_type="ddim", sampler=None, ddim_steps=50, eta=1.0,
                      cfg_scale=7.5, decode_frame_bs=1, ddp=False, all_gather=True, batch_progress=True,
                      show_denoising_progress=False, num_frames=None):
    """-> np.uint8 [n, T, H, W, 3].  `decode_frame_bs=None` decodes all frames in one VAE launch sequence."""
    if sample_type != "ddim" or sampler is None:
        raise NotImplementedError("only the DDIM path of the webui (process_videocrafter.py:57-79) is built")
    cond = get_conditions(prompt, model, batch_size)
    uncond = get_conditions(n_prompt, model, batch_size) if cfg_scale != 1.0 else None
    videos = []
    for _ in range(math.ceil(n_samples / batch_size)):
        noise_shape = make_model_input_shape(model, batch_size, T=num_frames)
        latent, _ = sampler.sample(S=ddim_steps, conditioning=cond, batch_size=noise_shape[0], shape=noise_shape[1:],
                                   verbose=show_denoising_progress, unconditional_guidance_scale=cfg_scale,
                                   unconditional_conditioning=uncond, eta=eta)
        samples = model.decode_first_stage(latent, decode_bs=decode_frame_bs, return_cpu=False)
        videos.append(torch_to_np(samples).numpy())
    out = np.concatenate(videos, axis=0)
    assert out.shape[0] >= n_samples
    return out
