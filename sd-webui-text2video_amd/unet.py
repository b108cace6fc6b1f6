"""UNetSD — drop-in for the reference's 3-D UNet denoiser (boundary B4).

Same constructor keywords, same `state_dict()` key set / shapes (1480 tensors for the
ModelScope config, including the reference's `temopral_conv` spelling), same
`forward(x, t, y)` contract and `register_schedule` buffers as
reference scripts/modelscope/t2v_model.py:98-501 — but `forward` does no torch arithmetic:
it lowers the network + input geometry once into a denoise program (program.py) and executes
it with the hand-written HIP kernels of libt2v_hip.so.

The nn.Module tree below exists only to hold parameters under the reference's names (so that
`load_state_dict(strict=True)`, `.half()`, `.to()` and the LoRA hook keep working); the
sub-modules' own `forward`s are never called.

Internal data layout (resident in HBM between kernels):
  * activations: channels-last tokens, row m = ((b*F + f)*H + y)*W + x, C contiguous.
    Residual-stream tensors are fp32, MFMA operands (outputs of norms / activations /
    projections feeding a GEMM or attention) are fp16.
  * weights: packed once to [N, K] fp16, reduction ordered (64-channel chunk, tap, channel) (packing.py).
Every `rearrange(...).contiguous()` of the reference (t2v_model.py:429,458,648,655,727-761,
1006-1008) is folded into GEMM addressing or attention strides.
"""
from __future__ import annotations

import os
from functools import partial
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import packing as pk
from .program import COLLECTIVE_KINDS, BoundProgram, Buf, Program, Ref, TShardSpec


# ------------------------------------------------------------------------------------------
# topology (shared by the parameter tree and the lowering)
# ------------------------------------------------------------------------------------------
def unet_layout(dim, dim_mult, num_res_blocks, attn_scales, temporal_attention=True):
    """Block list in execution order.  Each entry: (prefix, [(kind, cin, cout), ...]).
    Mirrors the constructor loop of the reference (t2v_model.py:148-318)."""
    enc = [dim * u for u in [1] + list(dim_mult)]
    dec = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    shortcut = [dim]
    scale = 1.0
    inputs, middle, outputs = [], [], []
    stem = [("stem", None, dim)] + ([("tt", dim, dim)] if temporal_attention else [])
    inputs.append(("input_blocks.0", stem, False))
    idx = 1
    cout = dim
    for i, (cin, cout) in enumerate(zip(enc[:-1], enc[1:])):
        for j in range(num_res_blocks):
            parts = [("res", cin, cout)]
            if scale in attn_scales:
                parts.append(("st", cout, cout))
                if temporal_attention:
                    parts.append(("tt", cout, cout))
            cin = cout
            inputs.append((f"input_blocks.{idx}", parts, False))
            idx += 1
            shortcut.append(cout)
            if i != len(dim_mult) - 1 and j == num_res_blocks - 1:
                inputs.append((f"input_blocks.{idx}", [("down", cout, cout)], True))
                idx += 1
                shortcut.append(cout)
                scale /= 2.0
    middle = [("res", cout, cout), ("st", cout, cout)] + ([("tt", cout, cout)] if temporal_attention else []) + \
             [("res", cout, cout)]
    oidx = 0
    for i, (cin, cout) in enumerate(zip(dec[:-1], dec[1:])):
        for j in range(num_res_blocks + 1):
            parts = [("res", cin + shortcut.pop(), cout)]
            if scale in attn_scales:
                parts.append(("st", cout, cout))
                if temporal_attention:
                    parts.append(("tt", cout, cout))
            cin = cout
            if i != len(dim_mult) - 1 and j == num_res_blocks:
                parts.append(("up", cout, cout))
                scale *= 2.0
            outputs.append((f"output_blocks.{oidx}", parts, False))
            oidx += 1
    return inputs, middle, outputs, cout


# ------------------------------------------------------------------------------------------
# parameter containers (reference key names)
# ------------------------------------------------------------------------------------------
def _attn_params(query_dim, context_dim, heads, dim_head):
    inner = heads * dim_head
    m = nn.Module()
    m.to_q = nn.Linear(query_dim, inner, bias=False)
    m.to_k = nn.Linear(context_dim or query_dim, inner, bias=False)
    m.to_v = nn.Linear(context_dim or query_dim, inner, bias=False)
    m.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))
    return m


def _transformer_block_params(dim, heads, d_head, context_dim):
    m = nn.Module()
    m.attn1 = _attn_params(dim, None, heads, d_head)
    ff = nn.Module()
    geglu = nn.Module()
    geglu.proj = nn.Linear(dim, dim * 4 * 2)
    ff.net = nn.Sequential(geglu, nn.Dropout(0.0), nn.Linear(dim * 4, dim))
    m.ff = ff
    m.attn2 = _attn_params(dim, context_dim, heads, d_head)
    m.norm1, m.norm2, m.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
    return m


def _spatial_transformer_params(channels, heads, d_head, context_dim):
    inner = heads * d_head
    m = nn.Module()
    m.norm = nn.GroupNorm(32, channels, eps=1e-6, affine=True)
    m.proj_in = nn.Linear(channels, inner)
    m.transformer_blocks = nn.ModuleList([_transformer_block_params(inner, heads, d_head, context_dim)])
    m.proj_out = nn.Linear(channels, inner)     # (in, out) order as in the reference, t2v_model.py:636
    return m


def _temporal_transformer_params(channels, heads, d_head):
    inner = heads * d_head
    m = nn.Module()
    m.norm = nn.GroupNorm(32, channels, eps=1e-6, affine=True)
    m.proj_in = nn.Conv1d(channels, inner, kernel_size=1)
    m.transformer_blocks = nn.ModuleList([_transformer_block_params(inner, heads, d_head, None)])
    m.proj_out = nn.Conv1d(inner, channels, kernel_size=1)
    return m


def _temporal_conv_params(c, dropout):
    m = nn.Module()
    m.conv1 = nn.Sequential(nn.GroupNorm(32, c), nn.SiLU(), nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0)))
    for name in ("conv2", "conv3", "conv4"):
        setattr(m, name, nn.Sequential(nn.GroupNorm(32, c), nn.SiLU(), nn.Dropout(dropout),
                                       nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))))
    return m


def _res_block_params(cin, emb, cout, dropout):
    m = nn.Module()
    m.in_layers = nn.Sequential(nn.GroupNorm(32, cin), nn.SiLU(), nn.Conv2d(cin, cout, 3, padding=1))
    m.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb, cout))
    m.out_layers = nn.Sequential(nn.GroupNorm(32, cout), nn.SiLU(), nn.Dropout(dropout),
                                 nn.Conv2d(cout, cout, 3, padding=1))
    m.skip_connection = nn.Identity() if cin == cout else nn.Conv2d(cin, cout, 1)
    m.temopral_conv = _temporal_conv_params(cout, 0.1)
    return m


def _conv_holder(attr, cin, cout, **kw):
    m = nn.Module()
    setattr(m, attr, nn.Conv2d(cin, cout, 3, **kw))
    return m


class UNetSD(nn.Module):
    """See module docstring.  Extra keyword (not in the reference): `init_weights=False` skips
    the (slow, 1.4 G parameter) default initialisation when a state dict is loaded right after."""

    supports_cfg_batch = True     # the sampler may stack cond/uncond into one b=2 call

    def __init__(self, in_dim=7, dim=512, y_dim=512, context_dim=512, out_dim=6, dim_mult=[1, 2, 3, 4],
                 num_heads=None, head_dim=64, num_res_blocks=3, attn_scales=[1 / 2, 1 / 4, 1 / 8],
                 use_scale_shift_norm=True, dropout=0.1, temporal_attn_times=2, temporal_attention=True,
                 use_checkpoint=False, use_image_dataset=False, use_fps_condition=False, use_sim_mask=False,
                 parameterization="eps", init_weights=True):
        super().__init__()
        if use_fps_condition or use_image_dataset:
            raise NotImplementedError("fps conditioning / image-dataset mode are not on the hot path")
        embed_dim = dim * 4
        num_heads = num_heads if num_heads else dim // 32
        self.in_dim, self.dim, self.y_dim, self.context_dim = in_dim, dim, y_dim, context_dim
        self.embed_dim, self.out_dim, self.dim_mult = embed_dim, out_dim, list(dim_mult)
        self.num_heads, self.head_dim, self.num_res_blocks = num_heads, head_dim, num_res_blocks
        self.attn_scales = list(attn_scales)
        self.temporal_attention = temporal_attention
        self.parameterization = parameterization
        self.v_posterior = 0
        if head_dim != 64:
            raise NotImplementedError("attention kernel is specialised for head_dim 64")

        self._layout = unet_layout(dim, dim_mult, num_res_blocks, attn_scales, temporal_attention)
        inputs, middle, outputs, last = self._layout

        ctx = torch.device("meta") if not init_weights else torch.device("cpu")
        with ctx:
            self.time_embed = nn.Sequential(nn.Linear(dim, embed_dim), nn.SiLU(), nn.Linear(embed_dim, embed_dim))
            self.input_blocks = nn.ModuleList()
            self.middle_block = nn.ModuleList()
            self.output_blocks = nn.ModuleList()
            for prefix, parts, bare in inputs:
                mods = [self._make(kind, cin, cout, dropout, decoder=False, stem=(prefix == "input_blocks.0"))
                        for kind, cin, cout in parts]
                self.input_blocks.append(mods[0] if bare else nn.ModuleList(mods))
            for kind, cin, cout in middle:
                self.middle_block.append(self._make(kind, cin, cout, dropout, decoder=False))
            for prefix, parts, bare in outputs:
                self.output_blocks.append(nn.ModuleList(
                    [self._make(kind, cin, cout, dropout, decoder=True) for kind, cin, cout in parts]))
            self.out = nn.Sequential(nn.GroupNorm(32, last), nn.SiLU(), nn.Conv2d(last, out_dim, 3, padding=1))
        if not init_weights:
            self.to_empty(device="cpu")
        else:
            self._zero_init()

        self._init_runtime()

    def _init_runtime(self):
        """Runtime state (not part of the state dict); shared with videocrafter.UNetModel."""
        self._programs: Dict[tuple, "_Compiled"] = {}
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._packed_sig = None
        self._packed_device = None
        self._packed_deps = None
        self._param_slots = None
        self.last_repack = None       # images rewritten by the last refresh_weights (-1 = full pack)
        self.debug_taps = False
        # Storage type of tensors that are consumed ONLY by a GroupNorm (ResBlock's first conv output
        # and the three inner temporal-conv outputs): "f16" halves their HBM traffic (what the
        # reference's .half() path stores everywhere), "f32" keeps them in the fp32 stream.
        self.norm_input_dtype = L.knob("T2V_NORM_INPUT", "f16")
        # GroupNorm statistics as a by-product of the GEMM that produces the normalised tensor (T2V_EPI_STATS strips, round 4): the
        # ResBlock's conv -> norm pairs and the temporal-conv chain then run "fold strips + apply" instead of the statistics pass /
        # the single-pass kernel with its grid barrier.  Part of the program cache key.
        # MEASURED SLOWER than the single-pass kernel it replaces (same box, round 4: 31.8 vs 25.2 us at the 32x32 level, 22.9 vs 18.8 us at
        # 8x8, GroupNorm 3.40 vs 3.21 ms per step) — two launches cost what one grid barrier costs — so it is OFF by default.
        self.gn_producer_stats = L.knob("T2V_GN_STRIPS", "0") != "0"
        self.context_token = None     # one-shot hint consumed by the next forward (see forward_cfg_pair)
        # one-shot hint consumed by the next forward: every sample of the batch has the SAME timestep (forward_cfg_pair and the samplers
        # of this package set it).  Only then may the cond | uncond pair share its prefix — the shared ops use sample 0's time embedding,
        # and forward(x[1], t = [t0, t1], ctx[2]) with t0 != t1 must keep each sample's own t
        self.single_timestep = False
        self._share_now = False
        # Precision option (off by default): weights whose packed-image name starts with one of these prefixes are applied as
        # hi + lo fp16 images in two MFMA passes (fp32 weights only; e.g. ("input_blocks.0", "input_blocks.1") — the blocks
        # that produce 46 % of the weight-rounding error).  Set before the first forward.
        self.split_weight_prefixes = ()
        # Precision (on by default): the fp32 -> fp16 operand casts whose rounding error reaches the output un-normalised — the
        # latent at the entry and the residual stream in front of the 1x1 skip convolutions — are emitted as hi + lo fp16
        # images and their (small) GEMMs run twice: 23 % of the activation-rounding error variance of a forward for ~15 extra
        # launches and +1.5 % FLOPs (tests/precision_probe.py; DESIGN.md "Precision").  Part of the program cache key.
        # Round 4 adds the next two classes of that ranking: the GroupNorm output in front of every transformer's proj_in and the
        # feed-forward output x4 in front of proj_out (rows [hi | lo] written by the producing kernel, K doubled in the C -> C
        # linear that consumes them) — at the input-resolution level, where the probe puts their whole gain; "all" = at every level
        # (+0.7 ms per step for ~1 % less error), "r3" = the round-3 subset only (A/B).
        self.precise_operands = {"0": False, "r3": "r3", "all": "all"}.get(os.environ.get("T2V_PRECISE", "1"), True)
        # Two more classes of the same ranking, built and parity-tested but OFF by default (ModelScope's outputs are inside north_star's
        # 1e-3 without them at +0.4 / +0.6 ms per step; on VideoCrafter they move the 10-step output 1.08e-3 -> 0.92e-3 but the 50-step
        # output only 1.07e-3 -> 1.04e-3): the attention output in front of to_out (input-resolution level) and the fp32 -> fp16 cast in
        # front of the Down / Upsample convolutions, both as rows [hi | lo] against [W | W].  Part of the program cache key.
        # (`precise_attn_out` is read by the VideoCrafter lowering only — videocrafter._LvdmLowering; the ModelScope lowering has no
        #  [hi | lo] attention output, so the flag does not enter ITS cache key: `_lowering_options`.)
        self.precise_attn_out = L.knob("T2V_PRECISE_ATTN", "0") != "0"
        self.precise_resample = L.knob("T2V_PRECISE_RESAMPLE", "0") != "0"
        # TemporalTransformer self-attention as ONE launch per attention (QKV projection + attention of every pixel's frame
        # sequence in the GEMM epilogue, T2V_EPI_TATTN): Q / K / V never reach HBM.  Clips of 2..32 frames; longer clips (and the
        # K/V-gather form of a T-sharded clip) keep the projection GEMM + attention kernel pair, and so do clips whose sequences fill
        # less than 144 of the tile's 192 rows (fewer than 12 frames) unless the option is "force".  Part of the program cache key.
        self.fused_temporal_attention = {"0": False, "force": "force"}.get(L.knob("T2V_FUSED_TATTN", "1"), True)
        self.t_shard = None           # parallel.TShard: this rank holds a contiguous slice of the clip's frames
        # Output dtype override for the package's own samplers.  `forward` returns what the reference's autocast path returns (fp16
        # for a `.half()` model) — but a guided step combines the two predictions as u + s (c - u): the INDEPENDENT fp16 roundings of
        # c and u (2.8e-4 each) come out multiplied by sqrt(s^2 + (s - 1)^2) ~ 12 at s = 9, which made the eps rounding the largest
        # single term of every sampled OUTPUT's error (round 4).  The samplers therefore set this to torch.float32 inside their loops
        # (the last convolution's result is fp32 in the arena anyway; the exit op writes 0.4 MB instead of 0.2 MB) and restore it.
        self.eps_out_dtype = None
        # Guided steps (forward_cfg_pair: one x_t, one t, [cond | uncond] contexts): everything up to the FIRST text cross-attention —
        # the stem, input_blocks.0's TemporalTransformer, the first ResBlock with its temporal convolutions, the first
        # SpatialTransformer's GroupNorm / proj_in / self-attention / to_q — is identical for the two samples.  With this option
        # (default on) the lowering computes that prefix ONCE (one sample's rows) and the first per-sample GEMMs read it through a row
        # wrap (T2V_OP_GEMM i[12]); the reference runs the two forwards separately and computes it twice (gaussian_sampler.py:161-162).
        # Same arithmetic on the same values; 31 of 725 ops run at half their rows.  Part of the program cache key.
        self.share_cfg_prefix = L.knob("T2V_SHARE_PREFIX", "1") != "0"
        # to_q projection + text cross-attention as ONE launch (T2V_EPI_XATTN, round 5): Q never reaches HBM; part of the program cache key.
        # Built, parity-tested and MEASURED NEUTRAL: per-op events say -6 us (32x32 level) / -12 us (16x16) per site, but the back-to-back step
        # is 26.09 vs 26.03 ms and 26.51 vs 26.37 ms on two boxes (the epilogue's K / V^T fragment loads from L2 are latency-bound: the fused
        # launch takes 53 us where projection + attention take 28 + 32) -> opt-in (T2V_XATTN=1)
        self.fused_cross_attention = L.knob("T2V_XATTN", "0") != "0"
        self.auto_refresh = True      # re-check parameter versions on every forward (~1 ms); the sampler
                                      # turns this off inside its loop after one explicit refresh
        self.device = torch.device("cpu")   # SamplerBase.register_buffers_to_model overwrites it (samplers_common.py:82)

    # ---- parameter tree -------------------------------------------------------------------
    def _make(self, kind, cin, cout, dropout, decoder, stem=False):
        if kind == "stem":
            return nn.Conv2d(self.in_dim, cout, 3, padding=1)
        if kind == "res":
            return _res_block_params(cin, self.embed_dim, cout, dropout)
        if kind == "st":
            # decoder SpatialTransformers hard-code context_dim=1024 in the reference (:293)
            return _spatial_transformer_params(cout, cout // self.head_dim, self.head_dim,
                                               1024 if decoder else self.context_dim)
        if kind == "tt":
            # the stem TemporalTransformer gets (num_heads, head_dim) from the config (:171-179),
            # all others cout // head_dim heads
            heads = self.num_heads if stem else cout // self.head_dim
            return _temporal_transformer_params(cout, heads, self.head_dim)
        if kind == "down":
            return _conv_holder("op", cin, cout, stride=2, padding=1)
        if kind == "up":
            return _conv_holder("conv", cin, cout, padding=1)
        raise ValueError(kind)

    def _zero_init(self):
        """The reference zero-initialises the last layer of every residual branch
        (t2v_model.py:326,631-636,708-713,955-956,1215-1216)."""
        with torch.no_grad():
            for n, m in self.named_modules():
                if n.endswith("proj_out") or n.endswith("out_layers.3"):
                    for p in m.parameters():
                        p.zero_()
                if n.endswith("temopral_conv.conv4.3"):
                    m.weight.zero_(); m.bias.zero_()
            self.out[-1].weight.zero_()

    # ---- schedule buffers (t2v_model.py:329-384) --------------------------------------------
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        if given_betas is None:
            raise NotImplementedError("pass given_betas (the pipeline always does, t2v_pipeline.py:113)")
        betas = np.asarray(given_betas, dtype=np.float64)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        t32 = partial(torch.tensor, dtype=torch.float32)
        reg = self.register_buffer
        reg("betas", t32(betas)); reg("alphas_cumprod", t32(ac)); reg("alphas_cumprod_prev", t32(ac_prev))
        reg("sqrt_alphas_cumprod", t32(np.sqrt(ac)))
        reg("sqrt_one_minus_alphas_cumprod", t32(np.sqrt(1.0 - ac)))
        reg("log_one_minus_alphas_cumprod", t32(np.log(1.0 - ac)))
        reg("sqrt_recip_alphas_cumprod", t32(np.sqrt(1.0 / ac)))
        reg("sqrt_recipm1_alphas_cumprod", t32(np.sqrt(1.0 / ac - 1)))
        pv = (1 - self.v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + self.v_posterior * betas
        reg("posterior_variance", t32(pv))
        reg("posterior_log_variance_clipped", t32(np.log(np.maximum(pv, 1e-20))))
        reg("posterior_mean_coef1", t32(betas * np.sqrt(ac_prev) / (1.0 - ac)))
        reg("posterior_mean_coef2", t32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)))

    # ---- weights --------------------------------------------------------------------------
    def _param_signature(self):
        """name -> (identity, version, device, dtype, shape) of every parameter.  Walks the cached per-module
        `_parameters` dicts (the module tree is fixed; re-assigned or added Parameters are still seen) instead of
        `named_parameters()`: this runs before every forward when `auto_refresh` is on."""
        if self._param_slots is None:
            self._param_slots = [(n + "." if n else "", m._parameters) for n, m in self.named_modules() if m._parameters]
        sig = {}
        for prefix, pd in self._param_slots:
            for k, p in pd.items():
                if p is not None:
                    sig[prefix + k] = (id(p), p._version, p.device.type, p.dtype, p.shape)
        return sig

    def invalidate(self):
        """Drop the packed weight images (call after mutating parameters in place)."""
        self._packed = None
        self._packed_sig = None

    def refresh_weights(self, device=None):
        """(Re)pack weights if any parameter object / version changed since the last pack.  When only some
        parameters changed (the LoRA merge / un-merge of lora_processor.py:202-246 replaces `.weight` of the
        matched Linear / Conv modules), only the packed images that read them are rewritten, in place
        (`last_repack` = number of images, -1 for a full pack).  The packed set is the UNION of the images of every
        compiled program (a T-sharded lowering declares `:kv` / `.to_q:lin` images the unsharded one does not, and
        vice versa)."""
        device = torch.device(device) if device is not None else self._packed_device
        if device is None:
            return
        sig = self._param_signature()
        same_dev = self._packed is not None and device == self._packed_device
        if same_dev and sig == self._packed_sig:
            return
        packer = self._union_packer()
        sd = {k: v for k, v in self.state_dict().items()}
        if same_dev and sig.keys() == self._packed_sig.keys():
            changed = [n for n, v in sig.items() if self._packed_sig[n] != v]
            n = packer.update(self._packed, sd, device, changed, deps=self._packed_deps)
            if n >= 0:
                # images that only an EVICTED program declared are not refreshed by `update`: drop them, a re-compiled
                # geometry packs them again (`_pack_missing`) instead of finding a stale copy
                for k in [k for k in self._packed if k not in packer._names]:
                    del self._packed[k]
                    self._packed_deps.pop(k, None)
                self._packed_sig, self.last_repack = sig, n
                for c in self._programs.values():
                    c.ctx_token = None          # cached context K/V were made with the old projection weights
                return
        elif same_dev:
            extra = sorted(set(sig) - set(self._packed_sig))
            if extra:
                # e.g. a LoRA file with bias terms for bias-free projections (lora_processor.py:219-222): the
                # compiled programs have no slot for them; refuse rather than silently ignore
                raise L.T2VError("parameters were added after the denoise programs were built: "
                                 f"{extra[:4]}{' ...' if len(extra) > 4 else ''} — not supported")
        self._packed = packer.materialise(sd, device)
        self._packed_deps = dict(packer.deps)
        self._packed_sig, self._packed_device, self.last_repack = sig, device, -1
        for c in self._programs.values():
            c.bound = None
            c.ctx_token = None

    def _union_packer(self) -> pk.WeightPacker:
        if not self._programs:
            self._get_compiled_any()
        packer = pk.WeightPacker()
        for comp in self._programs.values():
            for name, dtype, fn in comp.packer.recipes:
                packer.add(name, dtype, fn)
        return packer

    def _pack_missing(self, comp: "_Compiled", device):
        """Images a newly compiled program needs that no earlier program declared: packed and merged into the live
        set (existing device tensors — and the programs bound to them — are untouched)."""
        missing = [(n, d, f) for n, d, f in comp.packer.recipes if n not in self._packed]
        if not missing:
            return
        tmp = pk.WeightPacker()
        for n, d, f in missing:
            tmp.add(n, d, f)
        self._packed.update(tmp.materialise(self.state_dict(), device))
        self._packed_deps.update(tmp.deps)

    def _get_compiled_any(self):
        if self._programs:
            return next(iter(self._programs.values()))
        key = (1, 1, 8, 8, 77, "f32", "f32", "f32")          # same layout as forward()'s keys
        self._programs[key] = self._compile(1, 1, 8, 8, 77, "f32", "f32")
        return self._programs[key]

    # ---- forward --------------------------------------------------------------------------
    def forward(self, x, t, y, fps=None, video_mask=None, focus_present_mask=None, prob_focus_present=0.0,
                mask_last_frame_num=0):
        """eps = model(x[b,4,F,h,w], t[b], y[b,L,ctx])  — reference t2v_model.py:386-459.
        Output dtype follows the reference under autocast: fp16 for fp16 weights, else x.dtype."""
        # one-shot hint of the samplers ("every sample of this call has the same timestep"): consumed before anything can raise, so that
        # a failed call never leaves it set for a later call with genuinely different timesteps
        hint_single, self.single_timestep = bool(self.single_timestep), False
        if not x.is_cuda:
            raise L.T2VError("UNetSD.forward needs device tensors on an AMD GPU (no CPU fallback); "
                             "use oracle/torch_port.py for a CPU reference")
        Bx, C, F, H, W = x.shape
        B = y.shape[0]
        # x with fewer samples than the context: sample b of the batch reads x[b % Bx] (`forward_cfg_pair`)
        assert C == self.in_dim and B % Bx == 0 and y.shape[2] == self.context_dim
        x = x.contiguous()
        y = y.contiguous()
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        if y.dtype not in (torch.float16, torch.float32):
            y = y.float()
        p0 = next(self.parameters())
        out_dtype = torch.float16 if p0.dtype == torch.float16 else torch.float32
        if getattr(self, "eps_out_dtype", None) is not None:
            out_dtype = self.eps_out_dtype              # the samplers of this package ask for fp32 (see `eps_out_dtype`)
        tf = t.to(device=x.device, dtype=torch.float32).contiguous()
        if tf.ndim == 0:
            tf = tf.expand(B).contiguous()
        shard = None
        if self.t_shard is not None and self.t_shard.size > 1:
            shard = self.t_shard.spec                             # x holds only this rank's frames
            if shard.frames != F:
                raise L.T2VError(f"T-sharded forward: this rank holds {shard.frames} of {shard.total} frames, got {F}")
        single = hint_single or tf.numel() == 1 or t.ndim == 0
        if not single and not tf.is_cuda:
            single = bool((tf == tf.reshape(-1)[0]).all())
        self._share_now = bool(getattr(self, "share_cfg_prefix", False)) and single
        key = self._program_key(B, F, H, W, y.shape[1], x.dtype, y.dtype, out_dtype, shard, Bx)
        comp = self._programs.get(key)
        if comp is not None and comp.prog.gn_epilogue and L.exchange_disabled():
            comp = None        # an asynchronous fault was reported: lower again, without the norms fused into GEMM epilogues (_lib.exchange_disabled)
        if comp is None:
            comp = self._compile(B, F, H, W, y.shape[1], _dt(x.dtype), _dt(out_dtype), _dt(y.dtype), shard=shard,
                                 x_batch=Bx if Bx != B else 0)
            self._programs[key] = comp
            self._evict_programs(keep=key)
        if self._packed is None or self._packed_device != x.device or self.auto_refresh:
            self.refresh_weights(x.device)
        self._pack_missing(comp, x.device)
        comp.ensure_bound(self._packed, x.device, t_shard=self.t_shard if shard is not None else None)
        out = torch.empty((B, self.out_dim, F, H, W), device=x.device, dtype=out_dtype)
        ext = {L.EXT_X: x.data_ptr(), L.EXT_T: tf.data_ptr(), L.EXT_CTX: y.data_ptr(), L.EXT_OUT: out.data_ptr()}
        # step-invariant prologue: skipped only when the caller passed the token of the previous run of THIS binding
        token, self.context_token = self.context_token, None
        reuse = token is not None and comp.ctx_token == token and comp.ctx_bound is comp.bound
        if isinstance(comp.bound, BoundProgram):
            comp.bound.run(ext, torch.cuda.current_stream(x.device).cuda_stream, skip_invariant=reuse)   # one host call
        else:
            comp.bound.run(ext, torch.cuda.current_stream(x.device).cuda_stream)
        comp.ctx_token, comp.ctx_bound = token, comp.bound
        return out

    def _program_key(self, B, F, H, W, Lctx, x_dtype, y_dtype, out_dtype, shard=None, Bx=None):
        """Cache key of a compiled geometry (one helper for forward and forward_timed)."""
        return (B, F, H, W, Lctx, _dt(x_dtype), _dt(y_dtype), _dt(out_dtype)) + ((shard,) if shard else ()) + \
            ((("xb", Bx),) if (Bx is not None and Bx != B) else ()) + \
            ((("split",) + tuple(self.split_weight_prefixes),) if self.split_weight_prefixes else ()) + self._lowering_options()

    def _lowering_options(self) -> tuple:
        """Lowering switches that change the program (part of the cache key)."""
        return ((("strips",),) if getattr(self, "gn_producer_stats", False) else ()) + \
            ((("share",),) if getattr(self, "_share_now", False) else ()) + \
            ((("xattn",),) if getattr(self, "fused_cross_attention", False) else ()) + \
            ((("pattn",),) if (getattr(self, "precise_attn_out", False) and type(self) is not UNetSD) else ()) + ((("presample",),) if getattr(self, "precise_resample", False) else ()) + \
            ((("precise", str(self.precise_operands)),) if getattr(self, "precise_operands", False) else ()) + \
            ((("tattn", str(self.fused_temporal_attention)),) if getattr(self, "fused_temporal_attention", False) else ())

    def forward_cfg_pair(self, x, t, ctx_pair, context_token=None, single_t=None):
        """One guided step's two evaluations (gaussian_sampler.py:161-162) as ONE forward: x [V,4,F,h,w] is read twice by
        the entry op (no torch.cat([x, x])), ctx_pair = [cond (V) | uncond (V)]; -> eps [2V,...].  `context_token`: any
        hashable the caller changes whenever ctx_pair's CONTENT changes — equal to the previous call's token, the
        text-context K/V projections of that call are reused (they do not depend on x or t)."""
        tt = t.to(device=x.device, dtype=torch.float32).reshape(-1)
        if single_t is None:          # one t for every sample?  (a [1] tensor: yes; per-sample values: compared — the samplers say it outright)
            single_t = tt.shape[0] == 1 or bool((tt == tt[0]).all())
        tt = tt.repeat(ctx_pair.shape[0] // tt.shape[0]) if tt.shape[0] != ctx_pair.shape[0] else tt
        self.context_token = context_token
        self.single_timestep = bool(single_t)
        return UNetSD.forward(self, x, tt, ctx_pair)

    max_programs = 4      # compiled geometries kept (each owns a device arena: 0.4 GiB per 24-frame sample, GiBs for long clips)

    def _evict_programs(self, keep):
        """Least-recently-compiled eviction: a webui session that varies frames / resolution / batch_count would otherwise
        accumulate one arena per geometry (the reference frees its activations after every call)."""
        while len(self._programs) > self.max_programs:
            victim = next(k for k in self._programs if k != keep)
            del self._programs[victim]

    def forward_timed(self, x, t, y):
        """Like forward, but returns (eps, per-op milliseconds) using HIP events around every op
        on the launch stream (bench.py roofline measurement)."""
        single = bool((t.reshape(-1) == t.reshape(-1)[0]).all())      # (a profiling entry point: the comparison may synchronise)
        self.single_timestep = single
        out = UNetSD.forward(self, x, t, y)
        comp = self._programs[self._program_key(y.shape[0], x.shape[2], x.shape[3], x.shape[4], y.shape[1], x.dtype, y.dtype, out.dtype,
                                                Bx=x.shape[0])]
        xs, ys = x.contiguous(), y.contiguous()
        tf = t.to(device=x.device, dtype=torch.float32).contiguous()
        ext = {L.EXT_X: xs.data_ptr(), L.EXT_T: tf.data_ptr(), L.EXT_CTX: ys.data_ptr(), L.EXT_OUT: out.data_ptr()}
        ms = comp.bound.run_timed(ext, torch.cuda.current_stream(x.device).cuda_stream)
        return out, ms, comp.prog

    # ---- lowering -------------------------------------------------------------------------
    def _compile(self, B, F, H, W, Lctx, x_dt, out_dt, ctx_dt="f32", shard=None, x_batch=0):
        low = _Lowering(self, B, F, H, W, Lctx, x_dt, out_dt, ctx_dt, keep_taps=self.debug_taps, shard=shard, x_batch=x_batch)
        prog = low.build()
        return _Compiled(prog, low.packer)


def _dt(dtype) -> str:
    return "f16" if dtype == torch.float16 else "f32"


class _Compiled:
    def __init__(self, prog: Program, packer: pk.WeightPacker):
        self.prog = prog
        self.packer = packer
        self.bound: Optional[BoundProgram] = None
        self.arena: Optional[torch.Tensor] = None
        self.ctx_token = None          # context token of the last run (step-invariant prologue reuse)
        self.ctx_bound = None

    def ensure_bound(self, packed: Dict[str, torch.Tensor], device, t_shard=None):
        """Bind the program to a device arena and the packed weights.  A program with collective ops (T-sharded
        forward) is bound together with the T group's communicator: its exchanges then run inside the library (RCCL on
        the launch stream) — or, with T2V_COLLECTIVES=host / a non-GPU group, through parallel.ShardedExecutor."""
        if self.bound is not None and self.arena is not None and self.arena.device == device:
            return
        # zero-filled once at bind time: padding lanes that an op reads before any op wrote them (they only ever meet
        # zero weights) must not hold NaN bit patterns of recycled allocator blocks
        self.arena = torch.zeros(self.prog.arena.high + 256, dtype=torch.uint8, device=device)
        wptr = {k: v.data_ptr() for k, v in packed.items()}
        st = torch.cuda.current_stream(device).cuda_stream       # the bind-time reset of the sync words is ordered with the first run
        if any(op.kind in COLLECTIVE_KINDS for op in self.prog.ops):
            if t_shard is None:
                raise L.T2VError("a T-sharded program needs the T group (UNetSD.t_shard)")
            comm = t_shard.communicator(device)
            if comm is not None:
                self.bound = BoundProgram(self.prog, self.arena.data_ptr(), wptr, comm=comm, stream=st)
            else:
                from .parallel import ShardedExecutor
                self.bound = ShardedExecutor(self.prog, self.arena, t_shard,
                                             lambda ops: BoundProgram(self.prog, self.arena.data_ptr(), wptr, ops=ops))
        else:
            self.bound = BoundProgram(self.prog, self.arena.data_ptr(), wptr, stream=st)


# ------------------------------------------------------------------------------------------
# lowering: network + geometry -> denoise program
# ------------------------------------------------------------------------------------------
class _Lowering:
    def __init__(self, net: UNetSD, B, F, H, W, Lctx, x_dt, out_dt, ctx_dt, keep_taps=False, shard=None, x_batch=0):
        """F = frames held by THIS rank.  shard = TShardSpec: the clip's frames are split contiguously over the ranks
        of a T group (slices of ceil(F_total / R) frames, a shorter last one); temporal ops then exchange data
        (SURVEY §5.7): cross-frame GroupNorm -> all-gather of statistics partials, temporal conv -> +-1 frame halo,
        temporal attention -> all-gather of K/V.  Everything else is frame-local."""
        self.shard: Optional[TShardSpec] = shard if (shard is not None and shard.size > 1) else None
        if self.shard is not None:
            assert B == 1, "T-sharded forwards run one sample per rank (the CFG pair is split over ranks)"
            assert F == self.shard.frames
        self.net, self.B, self.F, self.H, self.W, self.Lctx = net, B, F, H, W, Lctx
        # cond | uncond prefix sharing (UNetSD.share_cfg_prefix): while `sharing`, activations hold Bc = 1 sample; the first spatial
        # transformer's text cross-attention is where the two samples part (transformer_block) and Bc becomes B
        # (`_share_now`: forward() sets it from `share_cfg_prefix` AND the single-timestep hint; a direct _compile keeps the attribute's value)
        self.sharing = bool(getattr(net, "_share_now", False)) and B == 2 and x_batch == 1 and self.shard is None
        self.Bc = 1 if self.sharing else B
        self.x_dt, self.out_dt, self.ctx_dt = x_dt, out_dt, ctx_dt
        self.x_batch = x_batch if 0 < x_batch < B else 0      # x holds fewer samples than the batch: sample b reads x[b % x_batch]
        self.precise = bool(getattr(net, "precise_operands", False))
        # round 4: the two next classes of the operand ranking (DESIGN.md "Precision") — GroupNorm -> proj_in and x4 -> proj_out.
        # Their consumers are the C -> C linears (HBM-bound at the 32x32 level: a doubled K costs bytes, not MFMA time)
        # Measured on MI355X (same box): every doubled C -> C linear costs ~12 us at every level (+1.04 ms per step for all 66), and the
        # probe by level (tests/precision_probe.py ... levels) puts 9.7 of the 10 % / 10.7 of the 13 % of these two classes' gain at
        # the INPUT resolution (its 11 + 11 sites) -> by default only the transformers of the top level are split; "all" = every level
        level = getattr(net, "precise_operands", False)
        self.precise_gn = self.precise and level != "r3"
        self.precise_ff = self.precise and level != "r3"
        self.precise_all_levels = level == "all"
        self.precise_attn = self.precise and bool(getattr(net, "precise_attn_out", False))
        self.precise_rs = self.precise and bool(getattr(net, "precise_resample", False))
        # GroupNorm statistics from the producing GEMM's epilogue (ResBlock conv -> norm, the temporal-conv chain): opt-in (measured slower)
        self.gn_strips = bool(getattr(net, "gn_producer_stats", False))
        self.last_stats: Optional[Buf] = None
        self.fused_tattn = bool(getattr(net, "fused_temporal_attention", False))
        self.stem_dup = False
        self.P = Program(f"unet b{B} f{F} {H}x{W}")
        self.P.keep_taps = keep_taps
        self.P.small_rank_tiles = self.shard is not None       # (Program.choose_tile: no temporal convolution of a rank program carries a norm epilogue)
        self.packer = pk.WeightPacker()
        prefixes = tuple(getattr(net, "split_weight_prefixes", ()) or ())
        if prefixes:
            self.P.weight_lo = lambda ref: (Ref("weight", ref.off, self.packer.add_lo(ref.name))
                                            if ref.name.startswith(prefixes) and ref.name.rsplit(":", 1)[-1] in ("lin", "lin2", "c3", "c3d", "t3", "qkv", "kv")
                                            else None)
        self.emb_slices: Dict[str, Tuple[int, int]] = {}
        self.kv_slices: Dict[str, Tuple[int, int]] = {}
        # fused to_q + text cross-attention (T2V_EPI_XATTN, round 5): V^T of every site, [B][sum of inner][keys padded], step-invariant
        self.vt_all: Optional[Buf] = None
        self.vt_slices: Dict[str, int] = {}

    # -- packed-weight declarations ------------------------------------------------------------
    def w_linear(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":lin", "f16", lambda sd, k=key: pk.pad_rows(pk.linear(sd[k + ".weight"]))))

    def precise_at(self, flag: bool, h: int, w: int) -> bool:
        """Is a round-4 operand split (flag) applied at the level whose frames are h x w?"""
        return bool(flag) and (self.precise_all_levels or (h == self.H and w == self.W))

    def w_proj(self, key, copies: int) -> Ref:
        """Linear weights for an operand stored as `copies` column blocks (1: plain; 2: rows [hi | lo] -> [W | W])."""
        assert copies in (1, 2)
        return self.w_linear(key) if copies == 1 else self.w_linear_dup(key)

    def w_linear_dup(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":lin2", "f16", lambda sd, k=key: pk.pad_rows(pk.linear_dup(sd[k + ".weight"]))))

    def w_conv3_c8dup(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":c3d", "f16", lambda sd, k=key: pk.pad_rows(pk.conv3x3_c8_dup(sd[k + ".weight"]))))

    def w_conv3(self, key, cin_pad=0) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":c3", "f16", lambda sd, k=key, c=cin_pad: pk.pad_rows(pk.conv3x3(sd[k + ".weight"], c))))

    def w_conv3_hilo(self, key) -> Ref:
        """3x3 conv weights for an operand of 2 Cin channels = [hi (Cin) | lo (Cin)]: [W | W] along the input channels."""
        return Ref("weight", 0, self.packer.add(key + ":c3hl", "f16", lambda sd, k=key: pk.pad_rows(pk.conv3x3(torch.cat([sd[k + ".weight"]] * 2, dim=1)))))

    def w_tconv(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":t3", "f16", lambda sd, k=key: pk.tconv3(sd[k + ".weight"])))

    def w_qkv(self, prefix) -> Ref:
        def fn(sd, p=prefix):
            return torch.cat([sd[p + ".to_q.weight"], sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]], dim=0)
        return Ref("weight", 0, self.packer.add(prefix + ":qkv", "f16", fn))

    def w_qkv_heads(self, prefix) -> Ref:
        def fn(sd, p=prefix):
            return pk.qkv_head_major(sd[p + ".to_q.weight"], sd[p + ".to_k.weight"], sd[p + ".to_v.weight"])
        return Ref("weight", 0, self.packer.add(prefix + ":qkvh", "f16", fn))

    def w_kv(self, prefix) -> Ref:
        def fn(sd, p=prefix):
            return torch.cat([sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]], dim=0)
        return Ref("weight", 0, self.packer.add(prefix + ":kv", "f16", fn))

    def w_geglu(self, key) -> Tuple[Ref, Ref]:
        def wfn(sd, k=key):
            w = sd[k + ".weight"]
            return w[pk.geglu_perm(w.shape[0] // 2, w.device)]
        def bfn(sd, k=key):
            b = sd[k + ".bias"]
            return b[pk.geglu_perm(b.shape[0] // 2, b.device)]
        return (Ref("weight", 0, self.packer.add(key + ":geglu", "f16", wfn)),
                Ref("weight", 0, self.packer.add(key + ":geglu_b", "f32", bfn)))

    def ln_arg(self, key, out: Buf, eps: float = 1e-5) -> tuple:
        """`ln=` argument of Program.gemm for the LayerNorm `key` applied to a GEMM's fp32 result: fused into the epilogue of the
        192x320 tile (one fp32 [2C] vector gamma | beta), a separate LayerNorm op otherwise."""
        gb = Ref("weight", 0, self.packer.add(key + ":ln_gb", "f32", lambda sd, k=key: torch.cat([sd[k + ".weight"], sd[k + ".bias"]])))
        return (gb, self.vec(key + ".weight"), self.vec(key + ".bias"), out, eps)

    def vec(self, key) -> Ref:
        """fp32 vector (bias / norm affine), zero-padded to a multiple of 4."""
        return Ref("weight", 0, self.packer.add(key + ":v", "f32", lambda sd, k=key: pk.pad_rows(sd[k])))

    # -- geometry helpers ----------------------------------------------------------------------
    def M(self, h, w):
        """Token rows of an activation at h x w in the CURRENT batch (one sample while the cond | uncond prefix is shared)."""
        return self.Bc * self.F * h * w

    # -- building blocks ------------------------------------------------------------------------
    def gn_gb(self, key) -> Ref:
        """gamma | beta of the GroupNorm `key` as ONE fp32 [2C] vector (the T2V_EPI_GN epilogue reads both from one pointer)."""
        return Ref("weight", 0, self.packer.add(key + ":gn_gb", "f32", lambda sd, k=key: torch.cat([sd[k + ".weight"], sd[k + ".bias"]])))

    def gn(self, name, x: Buf, key, *, per_frame: bool, eps, silu, out: Optional[Buf] = None, lo: bool = False,
           stats: Optional[Buf] = None, x_dead: bool = False, cast: Optional[Buf] = None, cast_lo: bool = False,
           halo_raw: Optional[Buf] = None) -> Buf:
        """lo (precise_operands): the result is a [rows, 2C] buffer of rows [hi | lo] — fp16(y) and the low-order image of that
        rounding — for a consumer GEMM with weights [W | W] (K = 2C)."""
        if lo:
            assert out is None
            full = self.P.alloc(x.rows, 2 * x.cols, "f16")
            out = full.col_slice(0, x.cols)
        else:
            full = out = self.P.alloc(x.rows, x.cols, "f16") if out is None else out
        n_inst = self.Bc * self.F if per_frame else self.Bc
        shard = None if per_frame else self.shard
        if stats is not None and (x.rows // n_inst) % 32 != 0:
            stats = None                  # (strips are 32 rows; a T-sharded cross-frame norm folds ITS part from them, round 6)
        # (x produced by the op emitted last, on a tile with the instantiation: the norm becomes that GEMM's epilogue — Program._fuse_groupnorm;
        #  x_dead: only this norm reads x)
        self.P.groupnorm(name, x, self.vec(key + ".weight"), self.vec(key + ".bias"), out, n_inst=n_inst, eps=eps, silu=silu,
                         shard=shard, lo=lo, stats=stats, gb=self.gn_gb(key) if shard is None and x.cols % 4 == 0 else None, x_dead=x_dead,
                         cast=cast, cast_lo=cast_lo, halo_raw=halo_raw if shard is not None else None)
        return full

    def strips_for(self, rows: int, n: int, inst_rows: int, k: Optional[int] = None, gather: int = L.GATHER_PLAIN) -> Optional[Buf]:
        """Buffer for the column statistics a GEMM's epilogue leaves for the GroupNorm that consumes its output (T2V_EPI_STATS:
        fp32 [rows / 32][2][n]) — round 4, VERDICT r03 next #4: that GroupNorm then needs neither a statistics pass over the
        tensor nor a grid barrier.  None when the option is off or the consumer's statistics instances (`inst_rows` rows each) are
        not whole 32-row strips."""
        if inst_rows % 32 != 0:
            return None
        if self.shard is not None and inst_rows == self.F * (rows // (self.Bc * self.F)):
            # T-sharded cross-frame norm (never an epilogue: its statistics cross ranks): this rank's part folded from the producer's
            # strips instead of a statistics pass over the tensor (round 6) — one small launch instead of two, per norm
            if L.knob("T2V_TSHARD_STRIPS", "1") == "0":
                return None
        elif not self.gn_strips:
            # round 6: automatically where the consuming norm can NOT run in its producer's epilogue because ONE statistics instance
            # has more row tiles than the device holds co-resident (cross-frame norms of 125-frame / 1024x576 clips): strips + fold +
            # one apply pass instead of the three-launch GroupNorm (two passes over the tensor + one)
            if k is None or not self.P.gn_instance_too_large(rows, n, k, gather, inst_rows) or L.knob("T2V_GN_STRIPS_AUTO", "1") == "0":
                return None
        return self.P.alloc(-(-rows // 32), 2 * n, "f32")

    def conv3(self, name, a: Buf, key, cout, h, w, *, stride=1, up=0, out_dtype="f32", rowbias=None,
              residual=None, cin=None, dest: Optional[Buf] = None, dup_c8: bool = False, stats: Optional[Buf] = None,
              hilo: bool = False) -> Buf:
        """`dest`: write the result into this (sub-)buffer instead of a fresh allocation — the producers of the two
        halves of a skip-connection concat write straight into the concat buffer (no copy ops)."""
        cin = a.cols if cin is None else cin
        ho, wo = (h * 2, w * 2) if up else ((h + 1) // 2 if stride == 2 else h, (w + 1) // 2 if stride == 2 else w)
        Mo = self.Bc * self.F * ho * wo
        n = (cout + 3) // 4 * 4
        out = self._dest(dest, Mo, n, out_dtype)
        gather = L.GATHER_CONV3X3_C8 if cin == 8 else L.GATHER_CONV3X3
        wref = self.w_conv3_hilo(key) if hilo else (self.w_conv3_c8dup(key) if dup_c8 else self.w_conv3(key, 8 if cin == 8 else 0))
        op = self.P.gemm(name, a, wref, n, 9 * cin, out, bias=self.vec(key + ".bias"),
                         gather=gather, conv=dict(Hin=h, Win=w, Cin=cin, stride=stride, up=up, Hout=ho, Wout=wo),
                         rowbias=rowbias, rows_per_batch=self.F * ho * wo if rowbias is not None else 0,
                         residual=residual, stats=stats, k_alg=9 * cin // 2 if hilo else None)
        self.last_stats = stats if (stats is not None and op.meta.get("stats")) else None     # None: the op ran with split-K
        return out

    def _dest(self, dest: Optional[Buf], rows, cols, dtype) -> Buf:
        if dest is None:
            return self.P.alloc(rows, cols, dtype)
        assert (dest.rows, dest.cols, dest.dtype) == (rows, cols, dtype), (dest, rows, cols, dtype)
        return dest

    def res_block(self, prefix, x: Buf, cin, cout, h, w, dest: Optional[Buf] = None) -> Buf:
        P = self.P
        # The 1x1 skip convolution (cin != cout) reads the block input as an fp16 operand.  Where the in_layers norm is its own op anyway
        # (its input is a skip-connection concat: two producers), that norm writes the cast as a second output — no separate pass over the
        # concat tensor (round 5).  Elsewhere (the norm is fused into its producer's epilogue) the cast op stays.
        x16 = None
        cast_in_norm = (cin != cout and self.shard is None and x.cols % 8 == 0 and L.knob("T2V_GN_CAST", "1") != "0"
                        and not (P.ops and P.ops[-1].kind == L.OP_GEMM and P.ops[-1].out is x))
        if cast_in_norm:
            x16 = P.alloc(x.rows, 2 * cin if self.precise else cin, "f16")
        a = self.gn(prefix + ".in_layers.0", x, prefix + ".in_layers.0", per_frame=True, eps=1e-5, silu=True, cast=x16, cast_lo=self.precise and cast_in_norm)
        e0, e1 = self.emb_slices[prefix]
        st = self.strips_for(x.rows, cout, h * w, 9 * cin, L.GATHER_CONV3X3)
        h1 = self.conv3(prefix + ".in_layers.2", a, prefix + ".in_layers.2", cout, h, w,
                        rowbias=self.emb_out.col_slice(e0, e1), out_dtype=self.net.norm_input_dtype, stats=st)
        P.free(a)
        b = self.gn(prefix + ".out_layers.0", h1, prefix + ".out_layers.0", per_frame=True, eps=1e-5, silu=True, stats=self.last_stats, x_dead=True)
        P.free(h1, st)
        if cin != cout:
            skip = P.alloc(x.rows, cout, "f32")
            if self.precise:
                # hi + lo operand split in ONE pass: rows [hi (cin) | lo (cin)], weights [W | W], K = 2 cin
                if x16 is None:
                    x16 = P.alloc(x.rows, 2 * cin, "f16")
                    P.copy2d(prefix + ".skip.cast", x, x16.col_slice(0, cin), lo=x16.col_slice(cin, 2 * cin))
                P.gemm(prefix + ".skip_connection", x16, self.w_linear_dup(prefix + ".skip_connection"), cout, 2 * cin, skip,
                       bias=self.vec(prefix + ".skip_connection.bias"), k_alg=cin)
            else:
                if x16 is None:
                    x16 = P.alloc(x.rows, cin, "f16")
                    P.copy2d(prefix + ".skip.cast", x, x16)
                P.gemm(prefix + ".skip_connection", x16, self.w_linear(prefix + ".skip_connection"), cout, cin, skip,
                       bias=self.vec(prefix + ".skip_connection.bias"))
            P.free(x16)
        else:
            skip = x
        st = self.strips_for(x.rows, cout, self.F * h * w, 9 * cout, L.GATHER_CONV3X3)
        # T-sharded: every tensor a cross-frame GroupNorm + (3,1,1) convolution pair reads lives in a buffer with one halo frame either
        # side, RAW: the statistics exchange of that norm carries the boundary frames along (T2V_OP_STATS_HALO) and the norm's apply pass
        # normalises them on arrival — one exchange per temporal convolution instead of two (T2V_STATS_HALO=0: the two-exchange form)
        merged = self.shard is not None and L.knob("T2V_STATS_HALO", "1") != "0"
        hwp = h * w
        raw = P.alloc((self.F + 2) * hwp, cout, "f32") if merged else None
        h2 = self.conv3(prefix + ".out_layers.3", b, prefix + ".out_layers.3", cout, h, w, residual=skip, stats=st,
                        dest=raw.row_slice(hwp, (self.F + 1) * hwp) if merged else None)
        st_live = self.last_stats
        P.free(b)
        if skip is not x:
            P.free(skip)
        # temporal convolution block: 4 x (GN over all frames + SiLU + (3,1,1) conv), identity residual
        t = h2
        tp = prefix + ".temopral_conv"
        for name, idx in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
            if self.shard is None:
                nrm = self.gn(f"{tp}.{name}.0", t, f"{tp}.{name}.0", per_frame=False, eps=1e-5, silu=True, stats=st_live, x_dead=t is not h2)
                P.free(st)
                st = st_live = None
            else:
                # T-sharded: normalised activations go into a buffer with one halo frame either side;
                # neighbours fill the halos (zeros at the two ends of the clip = the conv's zero padding)
                R, r = self.shard.size, self.shard.index
                nrm = P.alloc((self.F + 2) * hwp, cout, "f16")
                if r == 0:
                    P.memset(f"{tp}.{name}.halo0", nrm.row_slice(0, hwp))
                if r == R - 1:
                    P.memset(f"{tp}.{name}.halo1", nrm.row_slice((self.F + 1) * hwp, (self.F + 2) * hwp))
                self.gn(f"{tp}.{name}.0", t, f"{tp}.{name}.0", per_frame=False, eps=1e-5, silu=True,
                        out=nrm.row_slice(hwp, (self.F + 1) * hwp), halo_raw=raw, stats=st_live)
                P.free(st)
                st = st_live = None
                if not merged:
                    P.halo_exchange(f"{tp}.{name}.halo", nrm, hwp, self.F, self.shard)
            if t is not h2:
                P.free(t)
            if name == "conv4":
                t = self._dest(dest, h2.rows, cout, "f32")
            elif merged:
                raw = P.alloc((self.F + 2) * hwp, cout, self.net.norm_input_dtype)
                t = raw.row_slice(hwp, (self.F + 1) * hwp)
            else:
                t = P.alloc(h2.rows, cout, self.net.norm_input_dtype)
            key = f"{tp}.{name}.{idx}"
            if name != "conv4":
                st = self.strips_for(h2.rows, cout, self.F * h * w, 3 * cout, L.GATHER_TCONV3)    # conv1 .. conv3 feed the next cross-frame GroupNorm
            op = P.gemm(key, nrm, self.w_tconv(key), cout, 3 * cout, t, bias=self.vec(key + ".bias"),
                        gather=L.GATHER_TCONV3, conv=dict(F=self.F, HW=h * w, Cin=cout),
                        residual=h2 if name == "conv4" else None, halo=self.shard is not None, stats=st if name != "conv4" else None)
            st_live = st if (st is not None and name != "conv4" and op.meta.get("stats")) else None
            P.free(nrm)
        P.free(h2)
        return t

    def transformer_block(self, prefix, x1: Buf, n1: Optional[Buf], inner, heads, kind, h, w, geom=None) -> Buf:
        """x1: fp32 [M, inner] residual stream (consumed); n1 = LayerNorm1(x1) fp16, written by the proj_in GEMM (None in the
        K/V-gather lowering of a T-sharded temporal block, which normalises on its own).  Returns fp16 [M, inner] (feeds proj_out).
        geom = (frames, pixels per frame) of x1's rows when they are NOT this rank's frame slice: the pixel-sharded
        layout of a T-sharded TemporalTransformer — all frames of the clip, hw / R pixels — where the block is local."""
        P, Mrows, hw, F, B = self.P, x1.rows, h * w, self.F, self.Bc
        local = geom is not None
        if local:
            F, hw = geom
            assert B == 1 and Mrows == F * hw
        scale = 64 ** -0.5

        def self_attention_sharded(tag, xin: Buf) -> Buf:
            """Temporal self-attention over a T-sharded clip: queries = local frames, keys/values = ALL
            frames (K/V projections of the local tokens are all-gathered along T)."""
            R, r = self.shard.size, self.shard.index
            Mmax, Ftot = self.shard.max_frames * hw, self.shard.total     # rows of the largest slice; frames of the clip
            n = P.alloc(Mrows, inner, "f16")
            P.layernorm(f"{prefix}.norm{tag}", xin, self.vec(f"{prefix}.norm{tag}.weight"), self.vec(f"{prefix}.norm{tag}.bias"), n)
            q = P.alloc(Mrows, inner, "f16")
            P.gemm(f"{prefix}.attn{tag}.to_q", n, self.w_linear(f"{prefix}.attn{tag}.to_q"), inner, inner, q)
            # slice q of the gathered buffer starts at frame q * max_frames: only the LAST slice can be short, so the
            # clip's frames are contiguous from row 0 and the padding rows at the very end are never addressed
            kv_all = P.alloc(R * Mmax, 2 * inner, "f16")
            mine = kv_all.row_slice(r * Mmax, r * Mmax + Mrows)
            P.gemm(f"{prefix}.attn{tag}.kv", n, self.w_kv(f"{prefix}.attn{tag}"), 2 * inner, inner, mine)
            P.free(n)
            full = Buf(kv_all.ref, R * Mmax * 2 * inner * 2, 1, 1, "u8", kv_all.alloc_off)
            P.allgather(f"{prefix}.attn{tag}.kv.allgather", full, Mmax * 2 * inner * 2, self.shard)
            a = P.alloc(Mrows, inner, "f16")
            ldk = 2 * inner
            k, v = kv_all.col_slice(0, inner), kv_all.col_slice(inner, 2 * inner)
            P.attention(f"{prefix}.attn{tag}", q.ref, k.ref, v.ref, a.ref, out_buf=a, nq=F, nk=Ftot, heads=heads,
                        b_outer=1, b_inner=hw, q_strides=(hw * inner, 0, inner), kv_strides=(hw * ldk, 0, ldk),
                        o_strides=(hw * inner, 0, inner), scale=scale)
            P.free(q, kv_all)
            xo = P.alloc(Mrows, inner, "f32")
            P.gemm(f"{prefix}.attn{tag}.to_out", a, self.w_linear(f"{prefix}.attn{tag}.to_out.0"), inner, inner, xo,
                   bias=self.vec(f"{prefix}.attn{tag}.to_out.0.bias"), residual=xin)
            P.free(a, xin)
            return xo

        def self_attention(tag, xin: Buf, n: Buf, next_norm: str) -> Tuple[Buf, Buf]:
            """n = LayerNorm{tag}(xin) (made by the op that produced xin).  Returns (x + attention, LayerNorm{next_norm} of it):
            the to_out GEMM writes the fp32 stream AND — fused into its epilogue where the tile holds whole rows — the next
            LayerNorm's fp16 output."""
            # (measured, b=2 x 24 frames: 63 vs 61 + 38 us at the 32x32 level, 55 vs 49 + 21 us at 16x16; at K = 1280 the 192x192
            #  tile's main loop loses more than the fusion saves — 60 vs 43 + 13 us — so the 8x8 / 4x4 levels keep the pair)
            #  The tile holds min(12, 192 // F) pixels x F frames: below 144 of its 192 rows (clips shorter than 12 frames: 50 % at 8 frames,
            #  25 % at 4) the padding rows cost more MFMA time than the fusion saves, and the projection + attention pair is kept; so it is
            #  when this attention's weights are hi + lo split (`split_weight_prefixes`: the fused op has no second weight pass).
            split_here = bool(self.net.split_weight_prefixes) and f"{prefix}.attn{tag}".startswith(tuple(self.net.split_weight_prefixes))
            min_rows = 1 if getattr(self.net, "fused_temporal_attention", False) == "force" else 144      # "force": tests of short clips
            if kind != "spatial" and self.fused_tattn and P.tattn_pixels_per_tile(F) * F >= min_rows and inner % 64 == 0 and inner <= 640 \
                    and not split_here:
                # temporal self-attention: QKV projection + attention in ONE launch (q / k / v never reach HBM)
                a = P.alloc(Mrows, inner, "f16")
                P.qkv_temporal_attention(f"{prefix}.attn{tag}.qkv_attn", n, self.w_qkv_heads(f"{prefix}.attn{tag}"), a, samples=B, frames=F,
                                         hw=hw, heads=heads, k=inner, scale=scale)
                P.free(n)
                xo = P.alloc(Mrows, inner, "f32")
                nn_ = P.alloc(Mrows, inner, "f16")
                P.gemm(f"{prefix}.attn{tag}.to_out", a, self.w_linear(f"{prefix}.attn{tag}.to_out.0"), inner, inner, xo,
                       bias=self.vec(f"{prefix}.attn{tag}.to_out.0.bias"), residual=xin, ln=self.ln_arg(f"{prefix}.{next_norm}", nn_))
                P.free(a, xin)
                return xo, nn_
            qkv = P.alloc(Mrows, 3 * inner, "f16")
            P.gemm(f"{prefix}.attn{tag}.qkv", n, self.w_qkv(f"{prefix}.attn{tag}"), 3 * inner, inner, qkv)
            P.free(n)
            a = P.alloc(Mrows, inner, "f16")
            ld = 3 * inner
            q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
            if kind == "spatial":
                # >= 1024 tokens (the 32x32 level; the two upper levels of a 1024x576 clip): V transposed once per launch into a scratch,
                # K / V^T tiles staged by LDS-DMA, 8 waves per tile (csrc/attention.hip attn2_kernel, round 6: 656 -> 877 TF/s at 9216
                # tokens, 454 -> 560 at 1024 incl. the transposition; bit-identical results).  Fewer tokens: the extra launch costs more
                # (576 tokens: 337 vs 366 TF/s, 256: 260 vs 292 — tools/attn_probe.py).
                vt = None
                if hw >= 1024 and L.knob("T2V_ATTN2", "1") != "0":
                    vt = P.alloc(B * F * heads * 64, -(-hw // 64) * 64, "f16")
                P.attention(f"{prefix}.attn{tag}", q.ref, k.ref, v.ref, a.ref, out_buf=a, nq=hw, nk=hw, heads=heads,
                            b_outer=B * F, b_inner=1, q_strides=(ld, hw * ld, 0), kv_strides=(ld, hw * ld, 0),
                            o_strides=(inner, hw * inner, 0), scale=scale, vt_scratch=vt)
                P.free(vt)
            else:   # temporal: sequence = frames of one pixel
                P.attention(f"{prefix}.attn{tag}", q.ref, k.ref, v.ref, a.ref, out_buf=a, nq=F, nk=F, heads=heads,
                            b_outer=B, b_inner=hw, q_strides=(hw * ld, F * hw * ld, ld),
                            kv_strides=(hw * ld, F * hw * ld, ld),
                            o_strides=(hw * inner, F * hw * inner, inner), scale=scale)
            P.free(qkv)
            xo = P.alloc(Mrows, inner, "f32")
            nn_ = P.alloc(Mrows, inner, "f16")
            P.gemm(f"{prefix}.attn{tag}.to_out", a, self.w_linear(f"{prefix}.attn{tag}.to_out.0"), inner, inner, xo,
                   bias=self.vec(f"{prefix}.attn{tag}.to_out.0.bias"), residual=xin, ln=self.ln_arg(f"{prefix}.{next_norm}", nn_))
            P.free(a, xin)
            return xo, nn_

        sharded_temporal = kind == "temporal" and self.shard is not None and not local
        if sharded_temporal:
            # K/V-gather lowering (pixel count not divisible by the T group): LayerNorms as separate ops
            x2 = self_attention_sharded(1, x1)
            x3 = self_attention_sharded(2, x2)
            n3 = P.alloc(Mrows, inner, "f16")
            P.layernorm(f"{prefix}.norm3", x3, self.vec(f"{prefix}.norm3.weight"), self.vec(f"{prefix}.norm3.bias"), n3)
        else:
            x2, n2 = self_attention(1, x1, n1, "norm2")
            if kind == "spatial":
                k0, k1 = self.kv_slices[prefix + ".attn2"]
                kv = self.kv_all
                kbuf, vbuf = kv.col_slice(k0, k0 + inner), kv.col_slice(k0 + inner, k1)
                Lc = self.Lctx
                # The text cross-attention is where cond and uncond part.  While the prefix is shared, q / x2 hold ONE sample's rows:
                # the attention reads q with a zero sample stride against each sample's own K / V, and to_out adds x2 through the
                # residual row wrap — from here on every tensor has B samples.
                parting = self.sharing
                Bq, q_b_stride, Mshared = (self.B, 0, Mrows) if parting else (B, F * hw * inner, 0)
                Mq = Mrows
                if parting:
                    self.sharing, self.Bc = False, self.B
                    B, Mrows = self.B, self.B * Mrows
                a = P.alloc(Mrows, inner, "f16")
                fused_op = None
                if self.vt_all is not None and (F * hw) % 32 == 0 and Lc <= 96:
                    # to_q + the 77-key attention as ONE launch (round 5): q never reaches HBM.  (At the parting site the projection runs
                    # for both samples from the one sample's rows: the A-operand row wrap.)
                    v0 = self.vt_slices[prefix + ".attn2"]
                    vt = self.vt_all.row_slice(v0, self.vt_all.rows)           # this site's rows first; sample stride = all rows of a sample
                    vt_site = Buf(vt.ref, self.vt_all.rows, self.vt_all.cols, self.vt_all.ld, "f16", owns=False)
                    fused_op = P.to_q_cross_attention(f"{prefix}.attn2.q_attn", n2, self.w_linear(f"{prefix}.attn2.to_q"), a, k=inner, heads=heads,
                                                      kbuf=kbuf, vt=vt_site, n_keys=Lc, rows_per_sample=F * hw, samples=self.B, scale=scale,
                                                      a_wrap=Mq if parting else 0)
                if fused_op is None:
                    q = P.alloc(Mq, inner, "f16")
                    P.gemm(f"{prefix}.attn2.to_q", n2, self.w_linear(f"{prefix}.attn2.to_q"), inner, inner, q)
                    P.attention(f"{prefix}.attn2", q.ref, kbuf.ref, vbuf.ref, a.ref, out_buf=a, nq=hw, nk=Lc, heads=heads,
                                b_outer=Bq, b_inner=F, q_strides=(inner, q_b_stride, hw * inner),
                                kv_strides=(kv.ld, Lc * kv.ld, 0), o_strides=(inner, F * hw * inner, hw * inner), scale=scale)
                    P.free(q)
                P.free(n2)
                x3 = P.alloc(Mrows, inner, "f32")
                n3 = P.alloc(Mrows, inner, "f16")
                P.gemm(f"{prefix}.attn2.to_out", a, self.w_linear(f"{prefix}.attn2.to_out.0"), inner, inner, x3,
                       bias=self.vec(f"{prefix}.attn2.to_out.0.bias"), residual=x2, ln=self.ln_arg(f"{prefix}.norm3", n3), res_wrap=Mshared)
                P.free(a, x2)
            else:
                x3, n3 = self_attention(2, x2, n2, "norm3")
        # feed-forward: GEGLU fused in the first GEMM's epilogue
        n = n3
        wg, bg = self.w_geglu(f"{prefix}.ff.net.0.proj")
        g = P.alloc(Mrows, 4 * inner, "f16")
        P.gemm(f"{prefix}.ff.geglu", n, wg, 8 * inner, inner, g, bias=bg, epi=L.EPI_GEGLU)
        P.free(n)
        # x4 = x3 + FF feeds proj_out as an fp16 operand: with precise_operands it is stored as rows [hi | lo] (the low-order image
        # of the fp16 rounding beside the value) and proj_out runs with K doubled against [W | W]
        if self.precise_at(self.precise_ff, h, w):
            x4 = P.alloc(Mrows, 2 * inner, "f16")
            P.gemm(f"{prefix}.ff.net.2", g, self.w_linear(f"{prefix}.ff.net.2"), inner, 4 * inner, x4.col_slice(0, inner),
                   bias=self.vec(f"{prefix}.ff.net.2.bias"), residual=x3, out_lo=True)
        else:
            x4 = P.alloc(Mrows, inner, "f16")
            P.gemm(f"{prefix}.ff.net.2", g, self.w_linear(f"{prefix}.ff.net.2"), inner, 4 * inner, x4,
                   bias=self.vec(f"{prefix}.ff.net.2.bias"), residual=x3)
        P.free(g, x3)
        return x4

    def spatial_transformer(self, prefix, x: Buf, c, h, w, dest: Optional[Buf] = None) -> Buf:
        P = self.P
        heads = c // 64
        n = self.gn(prefix + ".norm", x, prefix + ".norm", per_frame=True, eps=1e-6, silu=False, lo=self.precise_at(self.precise_gn, h, w))
        x1 = P.alloc(x.rows, c, "f32")
        n1 = P.alloc(x.rows, c, "f16")
        tb = prefix + ".transformer_blocks.0"
        P.gemm(prefix + ".proj_in", n, self.w_proj(prefix + ".proj_in", n.cols // c), c, n.cols, x1, bias=self.vec(prefix + ".proj_in.bias"),
               ln=self.ln_arg(tb + ".norm1", n1), k_alg=c)
        P.free(n)
        x4 = self.transformer_block(tb, x1, n1, c, heads, "spatial", h, w)
        out = self._dest(dest, x4.rows, c, "f32")          # (x4 has B samples even when the block's input x was the shared single sample)
        P.gemm(prefix + ".proj_out", x4, self.w_proj(prefix + ".proj_out", x4.cols // c), c, x4.cols, out,
               bias=self.vec(prefix + ".proj_out.bias"), residual=x, k_alg=c, res_wrap=x.rows if x.rows != x4.rows else 0)
        P.free(x4)
        return out

    def temporal_transformer(self, prefix, x: Buf, c, heads, h, w, dest: Optional[Buf] = None) -> Buf:
        P = self.P
        inner = heads * 64
        if self.shard is not None and (h * w) % self.shard.size == 0:
            return self.temporal_transformer_resharded(prefix, x, c, heads, h, w, dest)
        n = self.gn(prefix + ".norm", x, prefix + ".norm", per_frame=False, eps=1e-6, silu=False, lo=self.precise_at(self.precise_gn, h, w))
        x1 = P.alloc(x.rows, inner, "f32")
        tb = prefix + ".transformer_blocks.0"
        n1 = None if self.shard is not None else P.alloc(x.rows, inner, "f16")
        P.gemm(prefix + ".proj_in", n, self.w_proj(prefix + ".proj_in", n.cols // c), inner, n.cols, x1, bias=self.vec(prefix + ".proj_in.bias"),
               ln=None if n1 is None else self.ln_arg(tb + ".norm1", n1), k_alg=c)
        P.free(n)
        x4 = self.transformer_block(tb, x1, n1, inner, heads, "temporal", h, w)
        out = self._dest(dest, x.rows, c, "f32")
        P.gemm(prefix + ".proj_out", x4, self.w_proj(prefix + ".proj_out", x4.cols // inner), c, x4.cols, out,
               bias=self.vec(prefix + ".proj_out.bias"), residual=x, k_alg=inner)
        P.free(x4)
        return out

    def temporal_transformer_resharded(self, prefix, x: Buf, c, heads, h, w, dest: Optional[Buf] = None) -> Buf:
        """T-sharded clip: everything between the block's GroupNorm and its `+ x` is local to a PIXEL (its tokens are the
        frames of one pixel, t2v_model.py:716-767), so instead of gathering K/V of all frames twice (6 slice-units received
        per rank) the block is resharded once: frame split -> pixel split (all frames of hw / R pixels) of the normalised
        fp16 tokens, the unsharded block lowering on that geometry, and the fp32 result back to the frame split where the
        residual is added — 0.75 + 1.5 slice-units moved per rank, and the rows are balanced even for uneven frame slices.
        Layouts: frame-sharded row = f_local * hw + pixel; pixel-sharded row = f_clip * (hw / R) + pixel_local."""
        P, sh = self.P, self.shard
        R, r, hw = sh.size, sh.index, h * w
        hwr, Ft, Fl, cb = hw // R, sh.total, self.F, sh.counts[0]          # pixels per rank, clip frames, my frames, frames per slice
        inner = heads * 64
        n = self.gn(prefix + ".norm", x, prefix + ".norm", per_frame=False, eps=1e-6, silu=False)     # [Fl*hw, c] fp16, frame split
        # ---- frames -> pixels: part q of `n` (pixels [q*hwr, (q+1)*hwr) of my frames) goes to rank q
        xp = P.alloc(Ft * hwr, c, "f16")
        stage = P.alloc(R * Fl * hwr, c, "f16")
        # (round 6: the R packs are ONE launch — part q = pixels [q*hwr, (q+1)*hwr) of my frames -> stage slot q; my own part straight
        #  into its place in xp)
        P.reshard_parts(prefix + ".f2p.pack", n, stage, parts=R, rows=Fl * hwr, chunk=hwr, s_src=hw, s_dst=hwr, part_rows_src=hwr,
                        part_rows_dst=Fl * hwr, own=r, own_other=xp.row_slice(sh.offset * hwr, (sh.offset + Fl) * hwr), own_is_src=False)
        P.alltoall(prefix + ".f2p", stage, xp, hwr * c * 2, sh, 0)
        P.free(n, stage)
        x1 = P.alloc(Ft * hwr, inner, "f32")
        n1 = P.alloc(Ft * hwr, inner, "f16")
        tb = prefix + ".transformer_blocks.0"
        P.gemm(prefix + ".proj_in", xp, self.w_linear(prefix + ".proj_in"), inner, c, x1, bias=self.vec(prefix + ".proj_in.bias"),
               ln=self.ln_arg(tb + ".norm1", n1))
        P.free(xp)
        x4 = self.transformer_block(tb, x1, n1, inner, heads, "temporal", h, w, geom=(Ft, hwr))
        yp = P.alloc(Ft * hwr, c, "f32")
        P.gemm(prefix + ".proj_out", x4, self.w_proj(prefix + ".proj_out", x4.cols // inner), c, x4.cols, yp, bias=self.vec(prefix + ".proj_out.bias"),
               k_alg=inner)
        P.free(x4)
        # ---- pixels -> frames: frames of slice q (rows [q*cb*hwr, ...) of yp) go to rank q; then unpack + residual
        back = P.alloc(R * Fl * hwr, c, "f32")
        P.alltoall(prefix + ".p2f", yp, back, hwr * c * 4, sh, 1)
        out = self._dest(dest, x.rows, c, "f32")
        P.reshard_parts(prefix + ".p2f.unpack", back, out, parts=R, rows=Fl * hwr, chunk=hwr, s_src=hwr, s_dst=hw, part_rows_src=Fl * hwr,
                        part_rows_dst=hwr, own=r, own_other=yp.row_slice(sh.offset * hwr, (sh.offset + Fl) * hwr), own_is_src=True, residual=x)
        P.free(yp, back)
        return out

    def resample(self, prefix, attr, x: Buf, c, h, w, *, up, dest: Optional[Buf] = None) -> Buf:
        P = self.P
        if self.precise_rs and c % 64 == 0:
            # the stream's fp16 cast as rows [hi | lo] (2c channels), the convolution on K = 9 * 2c against [W | W]
            x16 = P.alloc(x.rows, 2 * c, "f16")
            P.copy2d(prefix + ".cast", x, x16.col_slice(0, c), lo=x16.col_slice(c, 2 * c))
            out = self.conv3(f"{prefix}.{attr}", x16, f"{prefix}.{attr}", c, h, w, stride=1 if up else 2, up=1 if up else 0, dest=dest, hilo=True)
        else:
            x16 = P.alloc(x.rows, c, "f16")
            P.copy2d(prefix + ".cast", x, x16)
            out = self.conv3(f"{prefix}.{attr}", x16, f"{prefix}.{attr}", c, h, w, stride=1 if up else 2, up=1 if up else 0, dest=dest)
        P.free(x16)
        return out

    # -- whole network ------------------------------------------------------------------------------
    def build(self) -> Program:
        net, P, B, F = self.net, self.P, self.B, self.F
        inputs, middle, outputs, last = net._layout
        dim, emb = net.dim, net.embed_dim
        h, w = self.H, self.W
        P.begin()
        if self.sharing and not any(kind == "st" for _, parts, _ in inputs for kind, _, _ in parts):
            self.sharing, self.Bc = False, B           # no text cross-attention on the way down: nothing to share up to

        # ---- step-level prologue: time embedding, all ResBlock emb projections, all cross-attn K/V
        res_prefixes, st_prefixes = [], []
        for prefix, parts, bare in inputs + [("middle_block", middle, False)] + outputs:
            for i, (kind, cin, cout) in enumerate(parts):
                p = prefix if bare else f"{prefix}.{i}"
                if kind == "res":
                    res_prefixes.append((p, cout))
                elif kind == "st":
                    st_prefixes.append((p, cout))
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

        # The text-context K/V projections do not depend on x or t: they are the program's step-invariant prologue
        # (Op.meta['step_invariant']), skipped when the caller vouches that the context is the one of the previous run
        # (UNetSD.context_token).  Their buffer is allocated FIRST and freed last, so no other
        # buffer of the program can alias it between two runs.
        if n_kv:
            self.kv_all = P.alloc(B * self.Lctx, n_kv, "f16")
        # V^T of the text context for the fused to_q + cross-attention launches: rows = the value channels of every site, keys contiguous
        # (padded to a multiple of 32, zero beyond the context length: allocated here, never freed before the end, zero from the bind)
        n_vt = n_kv // 2
        xattn = (n_kv and self.shard is None and self.Lctx <= 96 and bool(getattr(net, "fused_cross_attention", True))
                 and all(c % 64 == 0 for _, c in st_prefixes))
        if xattn:
            off = 0
            for p, c in st_prefixes:
                self.vt_slices[p + ".transformer_blocks.0.attn2"] = off
                off += c
            self.vt_all = P.alloc(B * n_vt, -(-self.Lctx // 32) * 32, "f16")
        freqs = Ref("weight", 0, self.packer.add("time_freqs", "f32", lambda sd, d=dim: torch.pow(
            10000, -torch.arange(d // 2).to(torch.float32).div(d // 2))))
        te = P.alloc(B, dim, "f16")
        P.time_embed("time_embed.sincos", Ref("ext", L.EXT_T), freqs, te)
        e1 = P.alloc(B, emb, "f16")
        P.gemm("time_embed.0", te, self.w_linear("time_embed.0"), emb, dim, e1, bias=self.vec("time_embed.0.bias"), act=1)
        P.free(te)
        e_silu = P.alloc(B, emb, "f16")       # SiLU(e): the only form in which e is consumed (emb_layers = SiLU -> Linear)
        P.gemm("time_embed.2", e1, self.w_linear("time_embed.2"), emb, emb, e_silu, bias=self.vec("time_embed.2.bias"), act=1)
        P.free(e1)
        w_emb = Ref("weight", 0, self.packer.add("emb_all:lin", "f16", lambda sd, ps=tuple(p for p, _ in res_prefixes):
                                                 torch.cat([sd[p + ".emb_layers.1.weight"] for p in ps], dim=0)))
        b_emb = Ref("weight", 0, self.packer.add("emb_all:v", "f32", lambda sd, ps=tuple(p for p, _ in res_prefixes):
                                                 torch.cat([sd[p + ".emb_layers.1.bias"] for p in ps], dim=0)))
        self.emb_out = P.alloc(B, n_emb, "f32")
        P.gemm("emb_layers.all", e_silu, w_emb, n_emb, emb, self.emb_out, bias=b_emb)
        P.free(e_silu)

        n4 = -(-self.Lctx // 4) * 4                 # keys of the V^T GEMMs (N % 4 == 0)
        ctx_all = P.alloc(B * self.Lctx, net.context_dim, "f16")
        ctx16 = ctx_all
        ctx_src = Buf(Ref("ext", L.EXT_CTX), B * self.Lctx, net.context_dim, net.context_dim, self.ctx_dt)
        P.copy2d("context.cast", ctx_src, ctx16).meta["step_invariant"] = True
        if n_kv:
            w_kv = Ref("weight", 0, self.packer.add("kv_all:lin", "f16", lambda sd, ps=tuple(p for p, _ in st_prefixes): torch.cat(
                [torch.cat([sd[p + ".transformer_blocks.0.attn2.to_k.weight"], sd[p + ".transformer_blocks.0.attn2.to_v.weight"]], dim=0)
                 for p in ps], dim=0)))
            P.gemm("attn2.kv.all", ctx16, w_kv, n_kv, net.context_dim, self.kv_all, step_invariant=True)
        if self.vt_all is not None:
            # V^T [value channels of all sites, keys] = W_v x context^T per sample: swapped operands (the weights are the "token" operand).
            # The GEMM wants N % 4 == 0 keys: every sample's tokens are copied into its own zero-padded [n4, ctx] block, and the whole V^T
            # buffer is zeroed first, so that key columns Lctx .. lcp-1 are exactly 0 whatever the arena held and whatever the NEXT
            # sample's context contains (0 x Inf = NaN: a non-finite context must not leak into its neighbour's output).
            P.memset("attn2.vT.zero", self.vt_all).meta["step_invariant"] = True
            ctx_pad = ctx16
            if n4 > self.Lctx:
                ctx_pad = P.alloc(B * n4, net.context_dim, "f16")
                P.memset("context.pad", ctx_pad).meta["step_invariant"] = True
                for b in range(B):
                    P.copy2d(f"context.pad.{b}", ctx16.row_slice(b * self.Lctx, (b + 1) * self.Lctx),
                             ctx_pad.row_slice(b * n4, b * n4 + self.Lctx)).meta["step_invariant"] = True
            w_v = Ref("weight", 0, self.packer.add("v_all:lin", "f16", lambda sd, ps=tuple(p for p, _ in st_prefixes): torch.cat(
                [sd[p + ".transformer_blocks.0.attn2.to_v.weight"] for p in ps], dim=0)))
            wv_as_a = Buf(w_v, n_vt, net.context_dim, net.context_dim, "f16")
            for b in range(B):
                P.gemm(f"attn2.vT.all.{b}", wv_as_a, ctx_pad.row_slice(b * n4, (b + 1) * n4).ref, n4, net.context_dim,
                       self.vt_all.row_slice(b * n_vt, (b + 1) * n_vt).col_slice(0, n4), ldw=net.context_dim, allow_splitk=False,
                       step_invariant=True)
            if ctx_pad is not ctx16:
                P.free(ctx_pad)
        P.free(ctx_all)

        # ---- entry layout conversion: b c f h w -> tokens x 8 channels (4 real + 4 zero)
        xin = P.alloc(self.M(h, w), 8, "f16")
        # precise operands: the low-order fp16 images of the 4 latent channels ride in the 4 padding channels of the 8-channel
        # token rows, the stem's weights repeat W there — (x_hi + x_lo) . W in the one GEMM pass, no extra launch
        self.stem_dup = self.precise and self.x_dt == "f32" and net.in_dim == 4
        P.ncthw_to_cl("x.to_tokens", Ref("ext", L.EXT_X), self.x_dt, xin, B=self.Bc, C=net.in_dim, F=F, HW=h * w,
                      src_batch=self.x_batch if self.x_batch != self.Bc else 0, lo_in_pad=self.stem_dup)

        def run_parts(prefix, parts, bare, x, h, w, dest=None):
            for i, (kind, cin, cout) in enumerate(parts):
                p = prefix if bare else f"{prefix}.{i}"
                d = dest if i == len(parts) - 1 else None         # the block's result goes straight into a concat buffer
                # shared cond | uncond prefix: this part's result has ONE sample's rows; a skip-connection window wants it for every
                # sample -> the part writes its own buffer and two row-block copies fill the window (one site: input_blocks.0)
                spread = d if (self.sharing and d is not None and kind != "st") else None
                if spread is not None:
                    d = None
                if kind == "stem":
                    y = self.conv3(p, x, p, cout, h, w, cin=8, dest=d, dup_c8=self.stem_dup)
                elif kind == "res":
                    y = self.res_block(p, x, cin, cout, h, w, dest=d)
                elif kind == "st":
                    y = self.spatial_transformer(p, x, cout, h, w, dest=d)
                elif kind == "tt":
                    heads = net.num_heads if p == "input_blocks.0.1" else cout // 64
                    y = self.temporal_transformer(p, x, cout, heads, h, w, dest=d)
                elif kind == "down":
                    y = self.resample(p, "op", x, cout, h, w, up=False, dest=d)
                    h, w = (h + 1) // 2, (w + 1) // 2
                elif kind == "up":
                    y = self.resample(p, "conv", x, cout, h, w, up=True, dest=d)
                    h, w = h * 2, w * 2
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

        # Skip connections (t2v_model.py:447-452 `torch.cat([x, xs.pop()], dim=1)`): the concat buffer of decoder block j
        # is allocated when its encoder half is produced; encoder block k writes the right window, the op that produces
        # the decoder stream (middle block / previous decoder block) writes the left one — no copy kernels.
        n_skip = len(inputs)
        cats: List[Buf] = []
        x = xin
        for k, (prefix, parts, bare) in enumerate(inputs):
            sc = parts[-1][2]
            cin_total = outputs[n_skip - 1 - k][1][0][1]            # input channels of the consuming decoder ResBlock
            ho, wo = out_hw(parts, h, w)
            cat = P.alloc(self.B * self.F * ho * wo, cin_total, "f32")        # (every sample, whatever the current batch of the prefix)
            cats.append(cat)
            x, h, w = run_parts(prefix, parts, bare, x, h, w, dest=cat.borrow_cols(cin_total - sc, cin_total))
        cat = cats.pop()
        x, h, w = run_parts("middle_block", middle, False, x, h, w, dest=cat.borrow_cols(0, cat.cols - inputs[-1][1][-1][2]))
        for j, (prefix, parts, bare) in enumerate(outputs):
            nxt = cats.pop() if cats else None
            dest = nxt.borrow_cols(0, nxt.cols - inputs[n_skip - 2 - j][1][-1][2]) if nxt is not None else None
            x, h, w = run_parts(prefix, parts, bare, cat, h, w, dest=dest)
            cat = nxt

        assert not self.sharing and self.Bc == B, "the shared cond | uncond prefix never reached a text cross-attention"
        # ---- head: GN + SiLU + conv 3x3 -> out_dim, then tokens -> b c f h w
        a = self.gn("out.0", x, "out.0", per_frame=True, eps=1e-5, silu=True)
        P.free(x)
        y = self.conv3("out.2", a, "out.2", net.out_dim, h, w)
        P.free(a)
        P.cl_to_ncthw("eps.from_tokens", y, Ref("ext", L.EXT_OUT), self.out_dt, B=B, C=net.out_dim, F=F, HW=h * w)
        P.free(y, self.emb_out)
        if n_kv:
            P.free(self.kv_all)
        if self.vt_all is not None:
            P.free(self.vt_all)
        P.finish()
        return P
