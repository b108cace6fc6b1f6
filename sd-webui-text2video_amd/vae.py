"""AutoencoderKL — drop-in for the reference's VAE (boundary B5): decode (every video) and encode (vid2vid).

Same constructor (`ddconfig`, `embed_dim`, `ckpt_path`), same state-dict keys
(`post_quant_conv.*`, `decoder.*`, and — as parameter holders only — `encoder.*`,
`quant_conv.*`) and `decode(z[n,4,h,w]) -> [n,3,8h,8w]` contract as
reference scripts/modelscope/t2v_model.py:1585-1649, whose Decoder comes from Stability's
`ldm.modules.diffusionmodules.model` (not vendored; in-tree twin followed here:
scripts/videocrafter/lvdm/models/modules/autoencoder_modules.py:484-596).

`decode` lowers the decoder into a denoise program executed by libt2v_hip.so: conv 3x3 as
implicit GEMM (nearest-2x upsample folded into the gather), GroupNorm(eps 1e-6)+swish fused,
the single-head d=C mid attention as  S = Q K^T (GEMM) -> row softmax -> O = P V (GEMM with
V^T produced directly by a swapped-operand GEMM).  `encode(x[n,3,H,W])` (vid2vid input side,
t2v_model.py:1640-1644, Encoder autoencoder_modules.py:387-481) lowers the encoder the same way — its
Downsample pads (0,1,0,1) and convolves with stride 2 — and returns the DiagonalGaussianDistribution of
`quant_conv`'s moments (distributions.py:5-46).
"""
from __future__ import annotations

from typing import Dict

import os

import torch
import torch.nn as nn

from . import _lib as L
from . import packing as pk
from .program import Buf, Program, Ref
from .unet import _Compiled, _dt


def _resnet_params(cin, cout):
    m = nn.Module()
    m.norm1 = nn.GroupNorm(32, cin, eps=1e-6, affine=True)
    m.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
    m.norm2 = nn.GroupNorm(32, cout, eps=1e-6, affine=True)
    m.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
    if cin != cout:
        m.nin_shortcut = nn.Conv2d(cin, cout, 1)
    return m


def _attn_params(c):
    m = nn.Module()
    m.norm = nn.GroupNorm(32, c, eps=1e-6, affine=True)
    m.q, m.k, m.v, m.proj_out = nn.Conv2d(c, c, 1), nn.Conv2d(c, c, 1), nn.Conv2d(c, c, 1), nn.Conv2d(c, c, 1)
    return m


def _decoder_params(ch, out_ch, ch_mult, num_res_blocks, z_channels, **_):
    d = nn.Module()
    nres = len(ch_mult)
    block_in = ch * ch_mult[nres - 1]
    d.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
    d.mid = nn.Module()
    d.mid.block_1 = _resnet_params(block_in, block_in)
    d.mid.attn_1 = _attn_params(block_in)
    d.mid.block_2 = _resnet_params(block_in, block_in)
    ups = []
    for lvl in reversed(range(nres)):
        up = nn.Module()
        block_out = ch * ch_mult[lvl]
        blocks = []
        for _j in range(num_res_blocks + 1):
            blocks.append(_resnet_params(block_in, block_out))
            block_in = block_out
        up.block = nn.ModuleList(blocks)
        up.attn = nn.ModuleList()
        if lvl != 0:
            us = nn.Module()
            us.conv = nn.Conv2d(block_in, block_in, 3, padding=1)
            up.upsample = us
        ups.insert(0, up)
    d.up = nn.ModuleList(ups)
    d.norm_out = nn.GroupNorm(32, block_in, eps=1e-6, affine=True)
    d.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)
    return d


def _encoder_params(ch, ch_mult, num_res_blocks, in_channels, z_channels, double_z=True, **_):
    """Parameter holder only (keys/shapes of ldm Encoder) so that strict checkpoint loading works."""
    e = nn.Module()
    nres = len(ch_mult)
    e.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
    in_mult = (1,) + tuple(ch_mult)
    downs = []
    block_in = ch
    for lvl in range(nres):
        dn = nn.Module()
        block_in = ch * in_mult[lvl]
        block_out = ch * ch_mult[lvl]
        blocks = []
        for _j in range(num_res_blocks):
            blocks.append(_resnet_params(block_in, block_out))
            block_in = block_out
        dn.block = nn.ModuleList(blocks)
        dn.attn = nn.ModuleList()
        if lvl != nres - 1:
            ds = nn.Module()
            ds.conv = nn.Conv2d(block_in, block_in, 3, stride=2, padding=0)
            dn.downsample = ds
        downs.append(dn)
    e.down = nn.ModuleList(downs)
    e.mid = nn.Module()
    e.mid.block_1 = _resnet_params(block_in, block_in)
    e.mid.attn_1 = _attn_params(block_in)
    e.mid.block_2 = _resnet_params(block_in, block_in)
    e.norm_out = nn.GroupNorm(32, block_in, eps=1e-6, affine=True)
    e.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)
    return e


class DiagonalGaussianDistribution(object):
    """videocrafter/lvdm/models/modules/distributions.py:5-46 (the ldm class of the same name): `parameters` =
    [mean | logvar] along channels; latent-sized elementwise plumbing on the moments the encoder program produced."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, device=self.parameters.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, embed_dim, ckpt_path=None, image_key="image", colorize_nlabels=None,
                 monitor=None, ema_decay=None, learn_logvar=False, init_weights=True):
        super().__init__()
        assert ddconfig["double_z"]
        if ddconfig.get("attn_resolutions"):
            raise NotImplementedError("attention at up/down levels is not used by the reference config")
        self.ddconfig = dict(ddconfig)
        self.embed_dim = embed_dim
        ctx = torch.device("meta") if not init_weights else torch.device("cpu")
        with ctx:
            self.encoder = _encoder_params(**ddconfig)
            self.decoder = _decoder_params(**ddconfig)
            self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
            self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        if not init_weights:
            self.to_empty(device="cpu")
        self._programs: Dict[tuple, _Compiled] = {}
        self._packed = None
        self._packed_sig = None
        self._packed_device = None
        self.debug_taps = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    def init_from_ckpt(self, path):
        """Keys filtered by the 'first_stage_model.' prefix, as reference t2v_model.py:1619-1631."""
        sd = torch.load(path, map_location="cpu")["state_dict"]
        new = {k.split("first_stage_model.")[-1]: v for k, v in sd.items() if "first_stage_model" in k}
        self.load_state_dict(new, strict=True)

    def encode(self, x):
        """x [n, 3, H, W] in [-1, 1] -> DiagonalGaussianDistribution over [n, 4, H/8, W/8] (t2v_model.py:1640-1644)."""
        if not x.is_cuda:
            raise L.T2VError("AutoencoderKL.encode needs device tensors on an AMD GPU (no CPU fallback)")
        n, c, h, w = x.shape
        nlev = len(self.ddconfig["ch_mult"]) - 1
        assert c == self.ddconfig["in_channels"] and h % (1 << nlev) == 0 and w % (1 << nlev) == 0
        x = x.contiguous()
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        key = ("enc", n, h, w, _dt(x.dtype))
        comp = self._programs.get(key)
        if comp is None:
            low = _VaeLowering(self, n, h, w, _dt(x.dtype), "f32", self.debug_taps)
            comp = _Compiled(low.build_encoder(), low.packer)
            self._programs[key] = comp
        self._refresh(comp, x.device)
        comp.ensure_bound(self._packed, x.device)
        zc2 = 2 * self.embed_dim
        moments = torch.empty((n, zc2, h >> nlev, w >> nlev), device=x.device, dtype=torch.float32)
        comp.bound.run({L.EXT_X: x.data_ptr(), L.EXT_OUT: moments.data_ptr()}, torch.cuda.current_stream(x.device).cuda_stream)
        p0 = next(self.parameters())
        return DiagonalGaussianDistribution(moments.to(torch.float16 if p0.dtype == torch.float16 else torch.float32))

    # ---- weights ------------------------------------------------------------------------------
    def _param_signature(self):
        return tuple((id(p), p._version, p.device.type, p.dtype) for p in self.parameters())

    def invalidate(self):
        self._packed, self._packed_sig = None, None

    def _refresh(self, comp, device):
        sig = self._param_signature()
        have = self._packed is not None and all(name in self._packed for name, _, _ in comp.packer.recipes)
        if have and sig == self._packed_sig and device == self._packed_device:
            return
        packed = comp.packer.materialise(self.state_dict(), device)
        if self._packed is not None and sig == self._packed_sig and device == self._packed_device:
            packed = {**self._packed, **packed}          # decoder and encoder programs pack disjoint weight sets
        self._packed = packed
        self._packed_sig, self._packed_device = sig, device
        for c in self._programs.values():
            c.bound = None

    # ---- decode -------------------------------------------------------------------------------
    def decode(self, z):
        """z [n, 4, h, w] (the pipeline passes x0/0.18215, t2v_pipeline.py:348) -> [n, 3, 8h, 8w]."""
        return self._decode(z, None)

    def decode_to_uint8(self, z, videos: int = 1, bgr: bool = False):
        """decode + tensor2vid in ONE program: z [(videos f), 4, h, w] -> uint8 [f, 8h, videos * 8w, 3] (RGB, or BGR as
        postprocess_video returns them, t2v_pipeline.py:412-435).  The uint8 conversion is the decoder program's last
        op (T2V_OP_TO_UINT8 on the conv_out tokens); with fp16 weights it uses the fp16 arithmetic the reference's
        half-precision VAE path hands tensor2vid (t2v_pipeline.py:337-351,447-460)."""
        assert z.shape[0] % videos == 0
        return self._decode(z, (videos, bool(bgr)))

    def _decode(self, z, u8):
        if not z.is_cuda:
            raise L.T2VError("AutoencoderKL.decode needs device tensors on an AMD GPU (no CPU fallback)")
        n, c, h, w = z.shape
        z = z.contiguous()
        if z.dtype not in (torch.float16, torch.float32):
            z = z.float()
        p0 = next(self.parameters())
        out_dtype = torch.float16 if p0.dtype == torch.float16 else torch.float32
        key = (n, h, w, _dt(z.dtype), _dt(out_dtype), u8)
        comp = self._programs.get(key)
        if comp is None:
            low = _VaeLowering(self, n, h, w, _dt(z.dtype), _dt(out_dtype), self.debug_taps)
            comp = _Compiled(low.build(u8), low.packer)
            self._programs[key] = comp
            while len(self._programs) > self.max_programs:          # each program owns a device arena
                del self._programs[next(k for k in self._programs if k != key)]
        self._refresh(comp, z.device)
        comp.ensure_bound(self._packed, z.device)
        if u8 is None:
            out = torch.empty((n, self.ddconfig["out_ch"], 8 * h, 8 * w), device=z.device, dtype=out_dtype)
        else:
            out = torch.empty((n // u8[0], 8 * h, u8[0] * 8 * w, self.ddconfig["out_ch"]), device=z.device, dtype=torch.uint8)
        comp.bound.run({L.EXT_X: z.data_ptr(), L.EXT_OUT: out.data_ptr()},
                       torch.cuda.current_stream(z.device).cuda_stream)
        return out

    max_programs = 4

    def forward(self, input, sample_posterior=True):
        raise NotImplementedError("only decode() is on the hot path")


class _VaeLowering:
    def __init__(self, vae: AutoencoderKL, n, h, w, z_dt, out_dt, keep_taps=False):
        self.vae, self.n, self.h, self.w, self.z_dt, self.out_dt = vae, n, h, w, z_dt, out_dt
        self.P = Program(f"vae n{n} {h}x{w}")
        self.P.keep_taps = keep_taps
        self.packer = pk.WeightPacker()

    def w_linear(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":lin", "f16", lambda sd, k=key: pk.pad_rows(pk.linear(sd[k + ".weight"]))))

    def w_conv3(self, key, cin_pad=0) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":c3", "f16", lambda sd, k=key, c=cin_pad: pk.pad_rows(pk.conv3x3(sd[k + ".weight"], c))))

    def vec(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":v", "f32", lambda sd, k=key: pk.pad_rows(sd[k])))

    def gn(self, key, x: Buf, silu: bool) -> Buf:
        out = self.P.alloc(x.rows, x.cols, "f16")
        self.P.groupnorm(key, x, self.vec(key + ".weight"), self.vec(key + ".bias"), out, n_inst=self.n, eps=1e-6, silu=silu)
        return out

    def conv3(self, key, a: Buf, cout, h, w, *, up=0, down=False, residual=None, cin=None) -> Buf:
        """down=True: the encoder's Downsample — F.pad(x, (0,1,0,1)) then 3x3 stride 2 without padding
        (autoencoder_modules.py:150-166)."""
        cin = a.cols if cin is None else cin
        ho, wo = (2 * h, 2 * w) if up else ((h // 2, w // 2) if down else (h, w))
        nn_ = (cout + 3) // 4 * 4
        out = self.P.alloc(self.n * ho * wo, nn_, "f32")
        gather = L.GATHER_CONV3X3_C8 if cin == 8 else L.GATHER_CONV3X3
        conv = dict(Hin=h, Win=w, Cin=cin, stride=2 if down else 1, up=up, Hout=ho, Wout=wo)
        if down:
            conv["pad_after_only"] = 1
        self.P.gemm(key, a, self.w_conv3(key, 8 if cin == 8 else 0), nn_, 9 * cin, out, bias=self.vec(key + ".bias"),
                    gather=gather, conv=conv, residual=residual)
        return out

    def resnet(self, p, x: Buf, cin, cout, h, w) -> Buf:
        P = self.P
        a = self.gn(p + ".norm1", x, True)
        h1 = self.conv3(p + ".conv1", a, cout, h, w)
        P.free(a)
        b = self.gn(p + ".norm2", h1, True)
        P.free(h1)
        if cin != cout:
            x16 = P.alloc(x.rows, cin, "f16")
            P.copy2d(p + ".skip.cast", x, x16)
            skip = P.alloc(x.rows, cout, "f32")
            P.gemm(p + ".nin_shortcut", x16, self.w_linear(p + ".nin_shortcut"), cout, cin, skip,
                   bias=self.vec(p + ".nin_shortcut.bias"))
            P.free(x16)
        else:
            skip = x
        out = self.conv3(p + ".conv2", b, cout, h, w, residual=skip)
        P.free(b)
        if skip is not x:
            P.free(skip)
        return out

    def attn(self, p, x: Buf, c, h, w) -> Buf:
        """Single-head attention with d = C over the h*w tokens of each image (AttnBlock)."""
        P, hw = self.P, h * w
        nrm = self.gn(p + ".norm", x, False)
        wqk = Ref("weight", 0, self.packer.add(p + ":qk", "f16", lambda sd, k=p: torch.cat(
            [pk.linear(sd[k + ".q.weight"]), pk.linear(sd[k + ".k.weight"])], dim=0)))
        bqk = Ref("weight", 0, self.packer.add(p + ":qk_b", "f32", lambda sd, k=p: torch.cat([sd[k + ".q.bias"], sd[k + ".k.bias"]], dim=0)))
        qk = P.alloc(x.rows, 2 * c, "f16")
        P.gemm(p + ".qk", nrm, wqk, 2 * c, c, qk, bias=bqk)
        out_attn = P.alloc(x.rows, c, "f16")
        for img in range(self.n):
            rows = slice(img * hw, (img + 1) * hw)
            nrm_i = nrm.row_slice(rows.start, rows.stop)
            # V^T [c, hw] = Wv [c, c] x nrm_i^T : swapped operands, bias along rows
            vt = P.alloc(c, hw, "f16")
            wv_as_a = Buf(self.w_linear(p + ".v"), c, c, c, "f16")
            P.gemm(f"{p}.vT.{img}", wv_as_a, nrm_i.ref, hw, c, vt, bias=self.vec(p + ".v.bias"), ldw=nrm_i.ld,
                   bias_along_m=True, allow_splitk=False)
            q_i = qk.row_slice(rows.start, rows.stop).col_slice(0, c)
            k_i = qk.row_slice(rows.start, rows.stop).col_slice(c, 2 * c)
            # Scores are materialised per QUERY BLOCK, never for the whole image (round 4): at 1024x576 (9216
            # tokens) the [hw, hw] fp32 matrix is 340 MB + 170 MB of fp16 probabilities per frame.  Softmax rows are independent, so
            # blocking does not change a bit.  Round 5: the block is as LARGE as 256 MB of scores + probabilities allow (4608 rows at
            # 9216 tokens, two blocks per image) and the P V GEMM may split its reduction: 1152-row blocks made 36-workgroup P V
            # launches (M = 1152, N = 512 on 128 x 128 tiles, K = 9216 deep) on a 256-CU chip — 154 TF/s; measured on a 1024x576
            # frame (tools/gpu_pass.sh vaeattn, profiles/r05_vae_mid_attention_blocks.txt): score / softmax / P V core 1.113 ms
            # (1152 rows) -> 0.837 (1152 + split-K) -> 0.630 (2304 + split-K) -> 0.580 ms (4608), frame 11.48 -> 11.02 ms.  Memory is
            # not the constraint on a 288 GB part; launch fill is.  (A fused flash kernel at d = 512 was sized again: DESIGN.md §5 —
            # single pass needs the 128-query x 512 O tile spread over 8 waves with K / V ping-ponged through one LDS buffer each;
            # the two-pass form recomputes Q K^T per 128-wide O slice = 3x the FLOPs of this GEMM form.)
            cap = (256 << 20) // (6 * hw)                    # rows whose fp32 scores + fp16 probabilities fit 256 MB
            bq = hw if hw <= 4096 else next(b for b in (4608, 4096, 3072, 2304, 2048, 1152, 1024, 768, 512, 256, hw) if hw % b == 0 and (b <= cap or b == hw))
            if L.knob("T2V_VAE_BQ", None):           # experiment switch (tools/profile_vae.py): query rows per block
                bq = int(L.knob("T2V_VAE_BQ", None))
                assert hw % bq == 0
            pv_split = L.knob("T2V_VAE_PV_SPLITK", "1") == "1"
            for q0 in range(0, hw, bq):
                s = P.alloc(bq, hw, "f32")
                P.gemm(f"{p}.qk^T.{img}.{q0}", q_i.row_slice(q0, q0 + bq), k_i.ref, hw, c, s, ldw=k_i.ld, allow_splitk=False)
                pm = P.alloc(bq, hw, "f16")
                P.softmax(f"{p}.softmax.{img}.{q0}", s, pm, float(int(c) ** (-0.5)))
                P.free(s)
                P.gemm(f"{p}.pv.{img}.{q0}", pm, vt.ref, c, hw, out_attn.row_slice(rows.start + q0, rows.start + q0 + bq), ldw=vt.ld,
                       allow_splitk=pv_split)
                P.free(pm)
            P.free(vt)
        P.free(nrm, qk)
        out = P.alloc(x.rows, c, "f32")
        P.gemm(p + ".proj_out", out_attn, self.w_linear(p + ".proj_out"), c, c, out, bias=self.vec(p + ".proj_out.bias"),
               residual=x)
        P.free(out_attn)
        return out

    def build_encoder(self) -> Program:
        """Encoder.forward (autoencoder_modules.py:447-481) + quant_conv: image -> moments [n, 2*z, h/8, w/8]."""
        P, n, h, w = self.P, self.n, self.h, self.w
        dd = self.vae.ddconfig
        ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
        nres = len(ch_mult)
        zc2 = 2 * dd["z_channels"]
        assert dd["in_channels"] <= 8 and zc2 == 2 * self.vae.embed_dim == 8
        P.begin()
        xin = P.alloc(n * h * w, 8, "f16")
        P.ncthw_to_cl("x.to_tokens", Ref("ext", L.EXT_X), self.z_dt, xin, B=n, C=dd["in_channels"], F=1, HW=h * w)
        x = self.conv3("encoder.conv_in", xin, ch, h, w, cin=8)
        P.free(xin)
        P.tap("encoder.conv_in", x)

        def step(fn, name, *a):
            nonlocal x
            y = fn(name, x, *a)
            P.tap(name, y)
            P.free(x)
            x = y

        block_in = ch
        for lvl in range(nres):
            block_out = ch * ch_mult[lvl]
            for j in range(nrb):
                step(self.resnet, f"encoder.down.{lvl}.block.{j}", block_in, block_out, h, w)
                block_in = block_out
            if lvl != nres - 1:
                x16 = P.alloc(x.rows, block_in, "f16")
                P.copy2d(f"encoder.down.{lvl}.downsample.cast", x, x16)
                y = self.conv3(f"encoder.down.{lvl}.downsample.conv", x16, block_in, h, w, down=True)
                P.free(x16, x)
                x = y
                h, w = h // 2, w // 2
                P.tap(f"encoder.down.{lvl}.downsample", x)
        step(self.resnet, "encoder.mid.block_1", block_in, block_in, h, w)
        step(self.attn, "encoder.mid.attn_1", block_in, h, w)
        step(self.resnet, "encoder.mid.block_2", block_in, block_in, h, w)
        a = self.gn("encoder.norm_out", x, True)
        P.free(x)
        y = self.conv3("encoder.conv_out", a, zc2, h, w)
        P.free(a)
        y16 = P.alloc(y.rows, zc2, "f16")
        P.copy2d("encoder.conv_out.cast", y, y16)
        P.free(y)
        m = P.alloc(y16.rows, zc2, "f32")
        P.gemm("quant_conv", y16, self.w_linear("quant_conv"), zc2, zc2, m, bias=self.vec("quant_conv.bias"))
        P.free(y16)
        P.cl_to_ncthw("moments.from_tokens", m, Ref("ext", L.EXT_OUT), "f32", B=n, C=zc2, F=1, HW=h * w)
        P.free(m)
        P.finish()
        return P

    def build(self, u8=None) -> Program:
        P, n, h, w = self.P, self.n, self.h, self.w
        dd = self.vae.ddconfig
        ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
        nres = len(ch_mult)
        zc = dd["z_channels"]
        assert zc == 4 and self.vae.embed_dim == 4
        P.begin()
        zin = P.alloc(n * h * w, 8, "f16")
        P.ncthw_to_cl("z.to_tokens", Ref("ext", L.EXT_X), self.z_dt, zin, B=n, C=zc, F=1, HW=h * w)
        # post_quant_conv 1x1 (4 -> 4), written as an 8-channel fp16 token tensor (channels 4..7 = 0)
        wpq = Ref("weight", 0, self.packer.add("post_quant_conv:lin8", "f16", lambda sd: _pad2(pk.linear(sd["post_quant_conv.weight"]), 8, 8)))
        bpq = Ref("weight", 0, self.packer.add("post_quant_conv:b8", "f32", lambda sd: pk.pad_rows(sd["post_quant_conv.bias"], 8)))
        z2 = P.alloc(n * h * w, 8, "f16")
        P.gemm("post_quant_conv", zin, wpq, 8, 8, z2, bias=bpq)
        P.free(zin)
        block_in = ch * ch_mult[-1]
        x = self.conv3("decoder.conv_in", z2, block_in, h, w, cin=8)
        P.free(z2)
        P.tap("decoder.conv_in", x)

        def step(fn, name, *a):
            nonlocal x
            y = fn(name, x, *a)
            P.tap(name, y)
            P.free(x)
            x = y

        step(self.resnet, "decoder.mid.block_1", block_in, block_in, h, w)
        step(self.attn, "decoder.mid.attn_1", block_in, h, w)
        step(self.resnet, "decoder.mid.block_2", block_in, block_in, h, w)
        for lvl in reversed(range(nres)):
            block_out = ch * ch_mult[lvl]
            for j in range(nrb + 1):
                step(self.resnet, f"decoder.up.{lvl}.block.{j}", block_in, block_out, h, w)
                block_in = block_out
            if lvl != 0:
                x16 = P.alloc(x.rows, block_in, "f16")
                P.copy2d(f"decoder.up.{lvl}.upsample.cast", x, x16)
                y = self.conv3(f"decoder.up.{lvl}.upsample.conv", x16, block_in, h, w, up=1)
                P.free(x16, x)
                x = y
                h, w = 2 * h, 2 * w
                P.tap(f"decoder.up.{lvl}.upsample", x)
        a = self.gn("decoder.norm_out", x, True)
        P.free(x)
        y = self.conv3("decoder.conv_out", a, dd["out_ch"], h, w)
        P.free(a)
        if u8 is None:
            P.cl_to_ncthw("img.from_tokens", y, Ref("ext", L.EXT_OUT), self.out_dt, B=n, C=dd["out_ch"], F=1, HW=h * w)
        else:
            videos, bgr = u8
            Fr = n // videos                    # token row = ((video * Fr + f) * h + y) * w + x
            P.to_uint8("img.to_uint8", y.ref, y.dtype, Ref("ext", L.EXT_OUT), NI=videos, C=dd["out_ch"], F=Fr, H=h, W=w,
                       strides=(Fr * h * w * y.ld, 1, h * w * y.ld, w * y.ld, y.ld), half=self.out_dt == "f16", bgr=bgr)
        P.free(y)
        P.finish()
        return P


def _pad2(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    out = w.new_zeros(rows, cols)
    out[: w.shape[0], : w.shape[1]] = w
    return out
