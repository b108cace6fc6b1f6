"""Experiment: the prefetch-across-the-barrier schedule (tiles 13 / 14 / 15 / 17) against its lock-step base (1 / 2 / 3 / 8) on the
shapes that matter: TF/s, median of 12 back-to-back launches (per-op HIP events)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
c0 = dict(Hin=32, Win=32, Cin=320, stride=1, up=0, Hout=32, Wout=32)
c1 = dict(Hin=16, Win=16, Cin=640, stride=1, up=0, Hout=16, Wout=16)
c2 = dict(Hin=8, Win=8, Cin=1280, stride=1, up=0, Hout=8, Wout=8)
SHAPES = [("square 8192", 8192, 8192, 8192, None, 0, (1, 18, 3, 21)),
          ("8192x8320x8192", 8192, 8320, 8192, None, 0, (2, 19, 8, 20)),
          ("L0 conv3x3", 49152, 320, 2880, c0, 0, (2, 19, 8, 20)),
          ("L0 conv cat", 49152, 320, 5760, dict(c0, Cin=640), 0, (2, 19, 8, 20)),
          ("L0 geglu", 49152, 2560, 320, None, 1, (2, 19, 8, 20)),
          ("L0 ff2", 49152, 320, 1280, None, 0, (8, 20)),
          ("L0 qkv", 49152, 960, 320, None, 0, (8, 20)),
          ("L1 geglu", 12288, 5120, 640, None, 1, (2, 19)),
          ("L1 conv3x3 s2", 12288, 640, 5760, c1, 0, (8, 20)),
          ("L2 geglu", 3072, 10240, 1280, None, 1, (1, 18)),
          ("L2 conv3x3 s2", 3072, 1280, 11520, c2, 0, (3, 21)),
          ("L2 qkv", 3072, 3840, 1280, None, 0, (1, 18, 3, 21))]
for label, M, N, K, conv, epi, tiles in SHAPES:
    res = []
    for tile in tiles:
        P = Program()
        P.force_tile = tile
        split = 2 if "s2" in label else 1
        P.choose_tile = lambda *a, _t=tile, _s=split, **kw: (_t, _s)
        a = P.alloc(M, K if conv is None else conv["Cin"], "f16")
        out = P.alloc(M, N // 2 if epi else N, "f16")
        P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_PLAIN if conv is None else L.GATHER_CONV3X3,
               conv=conv, epi=epi)
        P.ops = P.ops * 12
        arena = torch.empty(P.arena.high + 256, dtype=torch.uint8, device=dev)
        arena.view(torch.float16).normal_(0, 1)
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).half()
        b = torch.randn(N, device=dev)
        bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr(), "b": b.data_ptr()})
        st = torch.cuda.current_stream(dev).cuda_stream
        bp.run({}, st)
        ms = sorted(bp.run_timed({}, st))[6]
        res.append(f"{tile}: {2.0 * M * N * K / ms / 1e9:6.0f}")
    print(f"{label:16s} {M:6d} {N:6d} {K:6d} | " + " | ".join(res), flush=True)
