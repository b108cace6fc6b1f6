"""Diagnostic (GPU): run a denoise program op by op over an arena pre-filled with NaN bit patterns and report the first op
whose output contains a NaN — i.e. an op that reads arena bytes no earlier op wrote (harmless with zero-filled fresh
memory, fatal with recycled memory).   python tests/poison_probe.py [vae|unet|lvdm]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import configs, synth  # noqa: E402
from sd_webui_text2video_amd import _lib as L, unet as U, vae as V  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram  # noqa: E402

DEV = torch.device("cuda:0")


def probe(comp, packed, ext_tensors, label):
    prog = comp.prog
    arena = torch.full((prog.arena.high + 256,), 0xFF, dtype=torch.uint8, device=DEV)
    wptr = {k: v.data_ptr() for k, v in packed.items()}
    stream = torch.cuda.current_stream(DEV).cuda_stream
    bad = 0
    for idx, op in enumerate(prog.ops):
        bp = BoundProgram(prog, arena.data_ptr(), wptr, ops=[op])
        bp.run({k: v.data_ptr() for k, v in ext_tensors.items()}, stream)
        torch.cuda.synchronize()
        b = op.out
        if b is None or b.ref.space != "arena":
            continue
        dt = torch.float32 if b.dtype == "f32" else torch.float16
        item = 4 if b.dtype == "f32" else 2
        flat = arena.view(dt)
        view = torch.as_strided(flat, (b.rows, b.cols), (b.ld, 1), b.ref.off // item)
        n = int(torch.isnan(view.float()).sum())
        if n:
            bad += 1
            print(f"[{label}] op {idx} {op.name} (kind {op.kind}) wrote {n} NaN of {b.rows * b.cols}; i={list(op.i)[:24]} meta={ {k: v for k, v in op.meta.items() if k != 'conv'} }")
            if bad >= 4:
                break
    for k, t in ext_tensors.items():
        if t.is_floating_point() and torch.isnan(t.float()).any():
            print(f"[{label}] ext slot {k} holds NaN")
    print(f"[{label}] {len(prog.ops)} ops, {bad} poisoned outputs")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vae"
    g = torch.Generator().manual_seed(5)
    if which == "vae":
        ae = V.AutoencoderKL(configs.TINY_VAE_DDCONFIG, 4)
        synth.load_synth(ae, seed=3)
        z = torch.randn(2, 4, 8, 8, generator=g).to(DEV)
        ae.decode(z)
        comp = next(iter(ae._programs.values()))
        out = torch.zeros(2, 3, 64, 64, device=DEV)
        probe(comp, ae._packed, {L.EXT_X: z, L.EXT_OUT: out}, "vae tiny 2x8x8")
    else:
        net = U.UNetSD(**configs.TINY_UNET)
        synth.load_synth(net, seed=0)
        x = torch.randn(2, 4, 3, 16, 16, generator=g).to(DEV)
        y = torch.randn(2, 7, 1024, generator=g).to(DEV)
        t = torch.tensor([801.0, 401.0], device=DEV)
        o = net(x, t, y)
        comp = next(iter(net._programs.values()))
        probe(comp, net._packed, {L.EXT_X: x, L.EXT_T: t, L.EXT_CTX: y, L.EXT_OUT: torch.zeros_like(o)}, "unet tiny")


if __name__ == "__main__":
    main()
