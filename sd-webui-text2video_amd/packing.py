"""Weight plumbing: turns the reference-layout state dict (torch [Cout, Cin, k...] tensors)
into the GEMM-ready packed images the HIP kernels read ([N, K] fp16, reduction ordered (64-channel
chunk, tap, channel); fp32 bias / affine vectors).  Pure layout work with torch ops, run once per
weight version (and again after `invalidate()`, e.g. when the LoRA hook of the reference,
scripts/stable_lora/stable_utils/lora_processor.py:215-246, has mutated `.weight` in place).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import torch

Recipe = Callable[[Dict[str, torch.Tensor]], torch.Tensor]


def _f16(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float16).contiguous()


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def pad_rows(w: torch.Tensor, mult: int = 4) -> torch.Tensor:
    n = w.shape[0]
    if n % mult == 0:
        return w
    pad = mult - n % mult
    return torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], dim=0)


def linear(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear / Conv1d k=1 / Conv2d 1x1 weight -> [N, K]."""
    return w.reshape(w.shape[0], -1)


KCHUNK = 64   # channels per reduction chunk (= the GEMM k-tile)


def conv3x3(w: torch.Tensor, cin_pad: int = 0) -> torch.Tensor:
    """[Co, Ci, 3, 3] -> [Co, 9*Ci'].  Reduction order for Ci % 64 == 0 (every conv of the UNet/VAE
    body): k = (ci // 64) * 9*64 + (ky*3+kx) * 64 + ci % 64 — "64-channel chunk, tap, channel" — so
    that the 9 taps that re-read the same activation cache lines are CONSECUTIVE k-tiles (they hit
    L1/L2 instead of re-streaming the tensor 9x).  The 4(+4 zero)-channel stem uses k = tap*8 + ci."""
    co, ci = w.shape[0], w.shape[1]
    w = w.permute(0, 2, 3, 1)                      # Co, ky, kx, Ci
    if cin_pad and cin_pad > ci:
        w = torch.cat([w, w.new_zeros(co, 3, 3, cin_pad - ci)], dim=3)
        ci = cin_pad
    if ci % KCHUNK == 0:
        w = w.reshape(co, 9, ci // KCHUNK, KCHUNK).permute(0, 2, 1, 3)   # Co, chunk, tap, 64
    return w.reshape(co, -1)


def conv3x3_c8_dup(w: torch.Tensor) -> torch.Tensor:
    """Stem conv [Co, 4, 3, 3] -> [Co, 9 * 8] with the 4 real input channels REPEATED in the 4 padding channels (k = tap * 8 + ci):
    the entry op puts the low-order fp16 images of the latent there, so the one GEMM pass computes (x_hi + x_lo) . W."""
    assert w.shape[1] == 4
    return conv3x3(torch.cat([w, w], dim=1))


def relpos_table16(tab: torch.Tensor, frames: int, transposed: bool) -> torch.Tensor:
    """A relative-position table [2R+1, d] (lvdm RelativePosition.embeddings_table, attention_temporal.py:21-41) as the persistent MFMA
    kernel of clips of <= 16 frames reads it (csrc/attention.hip relpos16_kernel): only rows R-(F-1) .. R+(F-1) are ever addressed by
    a clip of F frames (no clipping when R >= F-1), so slot jl = row jl + R-(F-1), 32 slots, fp16, zero-padded.
    transposed=False: [32, DK] (K-side table, DK = d rounded up to 16);  True: [DV, 32] (V-side table, DV = d rounded up to 32)."""
    n_rows, d = tab.shape
    R = (n_rows - 1) // 2
    assert frames <= 16 and R >= frames - 1
    jbase = R - (frames - 1)
    n = min(32, n_rows - jbase)
    if transposed:
        out = torch.zeros((d + 31) // 32 * 32, 32, dtype=torch.float16, device=tab.device)
        out[:d, :n] = tab[jbase:jbase + n].t().to(torch.float16)
    else:
        out = torch.zeros(32, (d + 15) // 16 * 16, dtype=torch.float16, device=tab.device)
        out[:n, :d] = tab[jbase:jbase + n].to(torch.float16)
    return out.contiguous()


def linear_dup(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> [N, 2K] = [W | W]: consumer of an operand laid out [hi (K) | lo (K)] per row."""
    w = linear(w)
    return torch.cat([w, w], dim=1)


def tconv3(w: torch.Tensor) -> torch.Tensor:
    """[Co, Ci, 3, 1, 1] -> [Co, 3*Ci] with k = (ci // 64) * 3*64 + kt * 64 + ci % 64 (Ci % 64 == 0)."""
    co, ci = w.shape[0], w.shape[1]
    w = w[:, :, :, 0, 0].permute(0, 2, 1)          # Co, kt, Ci
    assert ci % KCHUNK == 0
    return w.reshape(co, 3, ci // KCHUNK, KCHUNK).permute(0, 2, 1, 3).reshape(co, 3 * ci)


def qkv_head_major(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, head_dim: int = 64) -> torch.Tensor:
    """to_q / to_k / to_v weights [heads * d, K] -> [heads][q (d) | k (d) | v (d)][K]: one head's three projections are 3 * d
    consecutive output columns = one tile of the fused QKV + temporal attention GEMM (T2V_EPI_TATTN)."""
    heads = wq.shape[0] // head_dim
    parts = [w.reshape(heads, head_dim, -1) for w in (wq, wk, wv)]
    return torch.stack(parts, dim=1).reshape(3 * heads * head_dim, -1)


def geglu_perm(n_half: int, device=None) -> torch.Tensor:
    """Row permutation for the fused GEGLU epilogue: packed row 16u + 8g + j  <-  source row
    (g ? n_half : 0) + 8u + j   (value rows first, gate rows second in nn.Linear(dim, 2*inner),
    reference t2v_model.py:817-821)."""
    assert n_half % 8 == 0
    u = torch.arange(n_half // 8, device=device).view(-1, 1, 1)
    g = torch.arange(2, device=device).view(1, -1, 1)
    j = torch.arange(8, device=device).view(1, 1, -1)
    return (g * n_half + 8 * u + j).reshape(-1)


class _ReadLog:
    """Read-only view of a state dict that notes which keys a recipe touches."""

    def __init__(self, sd: Dict[str, torch.Tensor]):
        self.sd, self.read = sd, set()

    def __getitem__(self, k):
        self.read.add(k)
        return self.sd[k]

    def __contains__(self, k):
        self.read.add(k)
        return k in self.sd

    def get(self, k, default=None):
        self.read.add(k)
        return self.sd.get(k, default)


class WeightPacker:
    """Collects (name -> recipe) pairs during lowering and materialises them on a device.
    `materialise` packs everything and learns which state-dict keys each recipe reads; `update` re-packs, in
    place, only the images that depend on changed keys — the LoRA merge of the reference
    (scripts/stable_lora/stable_utils/lora_processor.py:202-246) rewrites some hundred attention / conv weights
    out of ~2600 tensors, and the device addresses bound into the denoise programs stay valid."""

    def __init__(self):
        self.recipes: List[Tuple[str, str, Recipe]] = []   # (name, 'f16'|'f32', fn)
        self._names = set()
        self.deps: Dict[str, frozenset] = {}                # packed name -> state-dict keys its recipe read

    def add(self, name: str, dtype: str, fn: Recipe) -> str:
        if name not in self._names:
            self._names.add(name)
            self.recipes.append((name, dtype, fn))
        return name

    def add_lo(self, name: str) -> str:
        """Second-order image of an fp16 weight image: lo = fp16(W - fp16(W)), so that A.W_hi + A.W_lo carries more of
        the fp32 weight through two fp16 MFMA passes (packing is layout-only, hence linear: pack(W) splits like W).
        All zeros for a model that is already `.half()` (the deployed form, t2v_pipeline.py:103-104).
        Precision actually carried: the residual is ~2^-12 |W|; for |W| below 2^-3 it lies in the fp16 SUBNORMAL
        range, whose spacing is 2^-24 absolute — torch's conversion and the gfx950 MFMA keep subnormals, so nothing is flushed,
        but the pair then resolves W to ~2^-25 absolute = 19-21 significant bits for the 1e-2..1e-1 weights of these networks
        (the full 22 only for |W| >= 2^-3).  A 2^11 pre-scale of `lo` with the factor folded into the second pass would restore
        the rest; not done — with fp16 deployment the option only matters for fp32-weight experiments."""
        lo = name + ":lo"
        if lo not in self._names:
            base = next(fn for n, d, fn in self.recipes if n == name)

            def fn(sd, base=base):
                t = base(sd).float()
                return t - t.half().float()
            self.add(lo, "f16", fn)
        return lo

    @staticmethod
    def _cast(t: torch.Tensor, dtype: str, device) -> torch.Tensor:
        return _f16(t, device) if dtype == "f16" else _f32(t, device)

    def materialise(self, sd: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
        out = {}
        for name, dtype, fn in self.recipes:
            log = _ReadLog(sd)
            out[name] = self._cast(fn(log), dtype, device)
            self.deps[name] = frozenset(log.read)
        return out

    def update(self, packed: Dict[str, torch.Tensor], sd: Dict[str, torch.Tensor], device, changed, deps=None) -> int:
        """Re-pack the images whose recipes read any key in `changed`, writing through the existing device
        tensors.  `deps` = the key sets learnt by the `materialise` that produced `packed` (default: this
        packer's own).  Returns the number of images rewritten, or -1 if an image or its key set is missing, or
        its shape changed (the caller then falls back to `materialise` and re-binds its programs)."""
        changed = set(changed)
        deps = self.deps if deps is None else deps
        todo = [(n, d, f) for n, d, f in self.recipes if n not in deps or (deps[n] & changed)]
        for name, dtype, fn in todo:
            if name not in packed or name not in deps:
                return -1
        fresh = []
        for name, dtype, fn in todo:
            t = self._cast(fn(sd), dtype, device)
            if t.shape != packed[name].shape:
                return -1
            fresh.append((name, t))
        for name, t in fresh:
            packed[name].copy_(t)
        return len(fresh)
