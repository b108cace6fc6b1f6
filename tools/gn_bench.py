"""GroupNorm / LayerNorm micro-benchmark over the UNet's shapes (b=2, 24 frames): per-op HIP-event time and
the achieved HBM rate (read x twice + write fp16 once).  `stats` = the statistics launch alone (phase 1)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(2, 24576, 320, "f16"), (2, 24576, 320, "f32"), (48, 1024, 320, "f32"), (2, 6144, 640, "f16"), (2, 6144, 640, "f32"),
          (48, 256, 640, "f32"), (2, 1536, 1280, "f16"), (48, 64, 1280, "f32"), (2, 384, 1280, "f16"), (48, 16, 1280, "f16"),
          (48, 1024, 960, "f32"), (48, 16, 2560, "f32"), (48, 64, 2560, "f32"), (48, 256, 1280, "f32"), (48, 256, 1920, "f32"),
          (2, 6144, 1280, "f16"), (2, 1536, 2560, "f32")]
print(f"{'n_inst':>6s} {'rows':>6s} {'C':>5s} {'dt':>4s} | {'op us':>7s} {'GB/s':>6s} | {'stats us':>8s} | {'1-launch us':>11s}")
for n_inst, rows, C, dt in SHAPES:
    res = []
    for shard in (None, (1, 0), "fused"):
        if shard == "fused" and (C // 32) % 4 != 0:
            res.append(float("nan"))
            continue
        P = Program()
        P.gn_fused_slice_bytes = (1 << 30) if shard == "fused" else 0
        P.gn_fused_total_bytes = 1 << 30
        x, out = P.alloc(n_inst * rows, C, dt), P.alloc(n_inst * rows, C, "f16")
        if shard is None or shard == "fused":
            P.groupnorm("gn", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), out, n_inst=n_inst, eps=1e-5, silu=True)
            gn = [op for op in P.ops if op.kind == 2]
            P.ops = gn * 12
        else:
            P.groupnorm("gn", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), out, n_inst=n_inst, eps=1e-5, silu=True, shard=(2, 0))
            P.ops = [op for op in P.ops if op.kind == 2 and op.i[8] == 1] * 12
        arena = torch.empty(P.arena.high + 256, dtype=torch.uint8, device=dev)
        arena.view(torch.float16).normal_(0, 1)
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        bp = BoundProgram(P, arena.data_ptr(), {"g": g.data_ptr(), "b": b.data_ptr()})
        st = torch.cuda.current_stream(dev).cuda_stream
        bp.run({}, st)
        ms = sorted(m for m, op in zip(bp.run_timed({}, st), P.ops) if op.kind == 2)
        res.append(ms[len(ms) // 2] * 1e3)
    item = 2 if dt == "f16" else 4
    by = n_inst * rows * C * (2 * item + 2)
    print(f"{n_inst:6d} {rows:6d} {C:5d} {dt:>4s} | {res[0]:7.1f} {by / res[0] / 1e3:6.0f} | {res[1]:8.1f} | {res[2]:11.1f}", flush=True)
