"""Target for rocprofv3 --pmc passes on the GEMM kernel alone: 8192^3 (tile 1, 256x256), the 32x32-level 3x3 convolution
shape as a plain GEMM (M49152 N320 K2880, tile 2, 256x320) and 8192^3 on the 3-stage 128x256 tile (3); normal data."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
# round 5: tiles 1 / 2 / 8 on the square shape (N = 8320 for the 320-wide tiles) and on the 32x32-level convolution shape
SHAPES = [(8192, 8192, 8192, 1), (8192, 8320, 8192, 2), (8192, 8320, 8192, 8), (49152, 320, 2880, 2), (49152, 320, 2880, 8), (8192, 8192, 8192, 3)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[1].split(",")]
for (M, N, K, tile) in SHAPES:
    P = Program()
    P.force_tile = tile
    a, out = P.alloc(M, K, "f16"), P.alloc(M, N, "f16")
    P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, allow_splitk=False)
    P.ops = P.ops * 3
    arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
    arena.view(torch.float16).normal_(0, 1)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr()})
    bp.run({}, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
print("ok")
