"""Square plain GEMMs (fp16 in, fp16 out, no epilogue extras) for comparison with published gfx950 numbers:
4096^3 and 8192^3, uniform [-1, 1) and zero operands, tiles 1 / 6 (256x256 lock-step / ping-pong) and 2 / 7."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
for n in (4096, 8192):
    for data in ("uniform", "normal", "zeros"):
        row = []
        for tile in (1, 6, 3):
            P = Program()
            P.force_tile = tile
            a, out = P.alloc(n, n, "f16"), P.alloc(n, n, "f16")
            P.gemm("g", a, Ref("weight", 0, "w"), n, n, out, allow_splitk=False)
            P.ops = P.ops * 8
            arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
            w = torch.zeros(n, n, device=dev, dtype=torch.float16)
            if data == "uniform":
                arena.view(torch.float16).uniform_(-1, 1)
                w.uniform_(-1, 1)
            elif data == "normal":
                arena.view(torch.float16).normal_(0, 1)
                w.normal_(0, 1)
            bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr()})
            st = torch.cuda.current_stream(dev).cuda_stream
            bp.run({}, st)
            ms = sorted(bp.run_timed({}, st))[4]
            row.append(f"tile {tile}: {2.0 * n ** 3 / ms / 1e9:6.0f} TF/s")
        print(f"{n}^3 {data:8s} | " + " | ".join(row), flush=True)
