"""Is the GEMM main loop waiting for its operand loads?  The same GEMM with (a) normal operands, (b) lda = 0 (every A row
is row 0: the activation loads hit the cache), (c) lda = ldw = 0.  Zero data throughout, so the clock is not power-limited
and the comparison isolates the memory system.  If (b)/(c) are much faster than (a), the 2-stage LDS ring does not
cover the load latency."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd.program import BoundProgram, Buf, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
for (M, N, K, tile) in [(49152, 320, 2880, 2), (49152, 320, 320, 2), (8192, 8192, 8192, 1), (12288, 640, 5760, 2), (8192, 8192, 8192, 3)]:
    row = []
    for variant in ("normal", "lda=0", "lda=ldw=0"):
        P = Program()
        P.force_tile = tile
        a, out = P.alloc(M, K, "f16"), P.alloc(M, N, "f16")
        av = a if variant == "normal" else Buf(a.ref, M, K, 0, "f16")
        P.gemm("g", av, Ref("weight", 0, "w"), N, K, out, allow_splitk=False, ldw=(0 if variant == "lda=ldw=0" else None))
        P.ops = P.ops * 8
        arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
        w = torch.zeros(N, K, device=dev, dtype=torch.float16)
        bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr()})
        st = torch.cuda.current_stream(dev).cuda_stream
        bp.run({}, st)
        ms = sorted(bp.run_timed({}, st))[4]
        row.append(f"{variant}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.0f} TF/s")
    print(f"M{M} N{N} K{K} tile {tile} | " + " | ".join(row), flush=True)
