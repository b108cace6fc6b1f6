"""GEMM micro-benchmark: every tile configuration x split-K over the UNet's dominant shapes
(SURVEY App. D at b=2, 24 frames).  Prints TF/s per (shape, tile, split)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
# (label, gather, M, N, K, conv)
F = int(os.environ.get("SWEEP_FRAMES", "24"))      # 16 + SWEEP_BATCH=2: VideoCrafter's M = 32768
B = int(os.environ.get("SWEEP_BATCH", "2"))      # 2 = cond+uncond batched (1 GPU); 1 = one CFG role per GPU (N >= 2)
SHAPES = []
for (C, hw, lvl) in [(320, 32, "L0"), (640, 16, "L1"), (1280, 8, "L2"), (1280, 4, "L3")]:
    M = B * F * hw * hw
    SHAPES.append((f"{lvl} lin C->C", L.GATHER_PLAIN, M, C, C, None, 0))
    SHAPES.append((f"{lvl} qkv", L.GATHER_PLAIN, M, 3 * C, C, None, 0))
    SHAPES.append((f"{lvl} geglu", L.GATHER_PLAIN, M, 8 * C, C, None, 1))
    SHAPES.append((f"{lvl} ff2", L.GATHER_PLAIN, M, C, 4 * C, None, 0))
    SHAPES.append((f"{lvl} conv3x3", L.GATHER_CONV3X3, M, C, 9 * C, dict(Hin=hw, Win=hw, Cin=C, stride=1, up=0, Hout=hw, Wout=hw), 0))
    SHAPES.append((f"{lvl} tconv", L.GATHER_TCONV3, M, C, 3 * C, dict(F=F, HW=hw * hw, Cin=C), 0))
SHAPES.append(("L1 conv cat", L.GATHER_CONV3X3, B * F * 256, 640, 9 * 1920, dict(Hin=16, Win=16, Cin=1920, stride=1, up=0, Hout=16, Wout=16), 0))
SHAPES.append(("L2 conv cat", L.GATHER_CONV3X3, B * F * 64, 1280, 9 * 2560, dict(Hin=8, Win=8, Cin=2560, stride=1, up=0, Hout=8, Wout=8), 0))

SHAPES.append(("L0 lin hi|lo", L.GATHER_PLAIN, B * F * 1024, 320, 640, None, 0))
TILES = tuple(int(t) for t in os.environ["SWEEP_TILES"].split(",")) if os.environ.get("SWEEP_TILES") else (0, 1, 2, 3, 4, 5, 8, 9, 11, 12)
only = sys.argv[1] if len(sys.argv) > 1 else None
print(f"{'shape':14s} {'M':>6s} {'N':>6s} {'K':>6s} | tile:split -> TF/s")
for label, gather, M, N, K, conv, epi in SHAPES:
    if only and only not in label:
        continue
    res = []
    for tile in TILES:
        if tile in (2, 7, 8, 11) and N % 320 != 0:
            continue
        splits = [1]
        P0 = Program(); P0.force_tile = tile
        _, auto = P0.choose_tile(M, N, K, gather)
        for s in sorted({1, auto, 2, 4}):
            if s > 1 and K < 1024:
                continue
            P = Program()
            P.force_tile = tile
            a_rows = M if conv is None or gather != L.GATHER_CONV3X3 else B * F * conv["Hin"] * conv["Win"]
            kin = K if gather == L.GATHER_PLAIN else conv["Cin"]
            a = P.alloc(a_rows, kin, "f16")
            n_out = N // 2 if epi else N
            out = P.alloc(M, n_out, "f16" if epi else "f32")
            res_buf = None if epi else P.alloc(M, N, "f32")
            orig = P.choose_tile
            P.choose_tile = lambda *aa, _t=tile, _s=s, **kw: (_t, _s)
            P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), gather=gather, conv=conv,
                   residual=res_buf, epi=epi)
            arena = torch.empty(P.arena.high + 256, dtype=torch.uint8, device=dev)
            arena.view(torch.float16).normal_(0, 1)
            w = (torch.randn(N, K, device=dev) / math.sqrt(K)).half()
            bvec = torch.randn(N, device=dev)
            bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr(), "b": bvec.data_ptr()})
            st = torch.cuda.current_stream(dev).cuda_stream
            P.ops = P.ops * 12            # the same op 12x in ONE plan: per-op HIP events, no host launch gaps
            bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr(), "b": bvec.data_ptr()})
            bp.run({}, st)
            mss = sorted(bp.run_timed({}, st))
            ms = mss[len(mss) // 2]
            res.append(f"{tile}:{s} {2.0 * M * N * K / ms / 1e9:6.0f}")
    print(f"{label:14s} {M:6d} {N:6d} {K:6d} | " + " | ".join(res), flush=True)
