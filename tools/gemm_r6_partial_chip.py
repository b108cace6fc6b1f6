"""Round 6 experiment: is the calibration GEMM power-limited?  The same 256x256 tiles (lock-step tile 1, staggered tile 22) on grids that
occupy 16 / 64 / 128 / 256 CUs (one tile per CU, K = 8192) and on the full 8192^2 output: TF/s per busy CU against the per-CU peak at
2.4 GHz (2.5 PF / 256 = 9.77).  A schedule-bound loop gives the same per-CU rate at every grid size; a power-bound chip gives more per CU
when fewer CUs are busy."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
K = 8192
for data in ("normal", "zeros"):
    for (M, N) in [(1024, 1024), (2048, 2048), (2048, 4096), (4096, 4096), (8192, 8192)]:
        row = []
        for tile in (1, 22):
            P = Program()
            P.force_tile = tile
            a, out = P.alloc(M, K, "f16"), P.alloc(M, N, "f16")
            P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, allow_splitk=False)
            P.ops = P.ops * 12
            arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
            w = torch.zeros(N, K, device=dev, dtype=torch.float16)
            if data == "normal":
                arena.view(torch.float16).normal_(0, 1)
                w.normal_(0, 1)
            bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr()})
            st = torch.cuda.current_stream(dev).cuda_stream
            bp.run({}, st)
            torch.cuda.synchronize()
            ms = sorted(bp.run_timed({}, st))[6]
            tiles = (M // 256) * (N // 256)
            busy = min(tiles, 256)
            tf = 2.0 * M * N * K / ms / 1e9
            rounds = -(-tiles // 256)
            row.append(f"tile {tile}: {tf:6.0f} TF/s = {tf / busy * (rounds * busy / tiles):5.2f} per busy CU ({tf / busy * (rounds * busy / tiles) / 9.77 * 100:4.1f} % of 9.77)")
        print(f"{data:6s} {M}x{N}x{K} ({(M // 256) * (N // 256):4d} tiles) | " + " | ".join(row), flush=True)
