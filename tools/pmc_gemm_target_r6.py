"""Target for the round-6 rocprofv3 --pmc passes: the calibration GEMM on the lock-step tile (1) and the staggered tile (22), on the whole
chip (8192^2 outputs) and on 16 CUs (1024^2 outputs, K = 8192), N(0,1) and zero operands.  Needs a build with tile 22
(tools/build_variant.py dev -DT2V_G2_DEV)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
for data in ("normal", "zeros"):
    for (M, N, K, tile) in [(8192, 8192, 8192, 1), (8192, 8192, 8192, 22), (1024, 1024, 8192, 1), (1024, 1024, 8192, 22)]:
        P = Program()
        P.force_tile = tile
        a, out = P.alloc(M, K, "f16"), P.alloc(M, N, "f16")
        P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, allow_splitk=False)
        P.ops = P.ops * 4
        arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
        w = torch.zeros(N, K, device=dev, dtype=torch.float16)
        if data == "normal":
            arena.view(torch.float16).normal_(0, 1)
            w.normal_(0, 1)
        bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr()})
        bp.run({}, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
print("ok")
