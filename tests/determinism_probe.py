"""Diagnostic (not collected by pytest; run by hand on the GPU box: python tests/determinism_probe.py [full]).
Run a denoise program op by op twice from identical state and report every op whose output
differs between the two runs (in-kernel races / uninitialised reads show up here)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import configs, synth  # noqa: E402
from sd_webui_text2video_amd import _lib as L, unet as U  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, Program  # noqa: E402

ITEM = {"f16": 2, "f32": 4, "f64": 8}


def single_op_programs(prog):
    out = []
    for op in prog.ops:
        p = Program()
        p.ops = [op]
        p.arena = prog.arena
        out.append(p)
    return out


def main():
    dev = torch.device("cuda:0")
    big = len(sys.argv) > 1 and sys.argv[1] == "full"
    cfg = configs.MODELSCOPE_UNET if big else configs.TINY_UNET
    net = U.UNetSD(**cfg, init_weights=False)
    synth.load_synth(net, seed=0)
    B, F, H, W, Lc = (2, 8, 32, 32, 77) if big else (2, 3, 16, 16, 7)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 4, F, H, W, generator=g).to(dev)
    y = torch.randn(B, Lc, 1024, generator=g).to(dev)
    t = torch.tensor([801.0, 401.0]).to(dev)
    out = torch.empty(B, 4, F, H, W, device=dev)
    comp = net._compile(B, F, H, W, Lc, "f32", "f32", "f32")
    packed = comp.packer.materialise(net.state_dict(), dev)
    wptr = {k: v.data_ptr() for k, v in packed.items()}
    prog = comp.prog
    arena = torch.zeros(prog.arena.high + 256, dtype=torch.uint8, device=dev)
    ext = {L.EXT_X: x.data_ptr(), L.EXT_T: t.data_ptr(), L.EXT_CTX: y.data_ptr(), L.EXT_OUT: out.data_ptr()}
    singles = [BoundProgram(p, arena.data_ptr(), wptr) for p in single_op_programs(prog)]
    stream = torch.cuda.current_stream(dev).cuda_stream
    reps = 3
    bad = 0
    for idx, (op, bp) in enumerate(zip(prog.ops, singles)):
        if op.out is None or op.out.ref.space != "arena":
            bp.run(ext, stream)
            torch.cuda.synchronize()
            continue
        o = op.out
        lo, n = o.ref.off, ((o.rows - 1) * o.ld + o.cols) * ITEM[o.dtype]
        snaps = []
        for r in range(reps):
            arena[lo:lo + n].fill_(0x7F)         # poison the output region
            bp.run(ext, stream)
            torch.cuda.synchronize()
            snaps.append(arena[lo:lo + n].clone())
        diff = [int((snaps[0] != s).sum()) for s in snaps[1:]]
        if any(diff):
            bad += 1
            dt = torch.float16 if o.dtype == "f16" else torch.float32
            a, b = snaps[0].view(dt).float(), snaps[1 + diff.index(max(diff))].view(dt).float()
            m = torch.isfinite(a) & torch.isfinite(b)
            print(f"op {idx:4d} kind {op.kind} {op.name:60s} differing bytes {diff} max|d| {float((a[m] - b[m]).abs().max()):.3e} "
                  f"meta {op.meta if op.kind == 1 else list(op.i[:8])}")
    print(f"{bad} nondeterministic ops of {len(prog.ops)}")


if __name__ == "__main__":
    main()
