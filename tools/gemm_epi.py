"""Epilogue cost probe: the same GEMM with (a) fp16 out, (b) fp32 out, (c) fp32 out + fp32 residual, (d) + bias.
Prints per-launch time (HIP events inside one plan) and the implied HBM rate."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(49152, 320, 320), (49152, 320, 1280), (12288, 640, 640), (3072, 1280, 1280), (49152, 960, 320)]
for M, N, K in SHAPES:
    for tile in (None, 0, 2, 5):
        row = []
        for name, odt, res, bias in (("f16", "f16", False, False), ("f32", "f32", False, False), ("f32+res", "f32", True, False),
                                     ("f32+res+bias", "f32", True, True)):
            P = Program()
            if tile is not None:
                P.force_tile = tile
            a = P.alloc(M, K, "f16")
            out = P.alloc(M, N, odt)
            r = P.alloc(M, N, "f32") if res else None
            op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b") if bias else Ref("null"),
                        residual=r, allow_splitk=False)
            P.ops = P.ops * 12
            arena = torch.empty(P.arena.high + 256, dtype=torch.uint8, device=dev)
            arena.view(torch.float16).normal_(0, 1)
            w = (torch.randn(N, K, device=dev) / math.sqrt(K)).half()
            b = torch.randn(N, device=dev)
            bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr(), "b": b.data_ptr()})
            st = torch.cuda.current_stream(dev).cuda_stream
            bp.run({}, st)
            ms = sorted(bp.run_timed({}, st))[6]
            by = M * K * 2 + N * K * 2 + M * N * (2 if odt == "f16" else 4) + (M * N * 4 if res else 0)
            row.append(f"{name} {ms * 1e3:6.1f}us {by / ms / 1e9:5.2f}TB/s")
        print(f"{M:6d}x{N:4d}x{K:4d} tile {str(op.i[22]) if tile is None else tile}: " + " | ".join(row), flush=True)
