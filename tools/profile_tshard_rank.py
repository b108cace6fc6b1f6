"""Compute time of ONE rank's T-sharded UNet forward on one GPU: the real sharded program (the ops a rank of the frame-parallel layout
executes: statistics / apply passes, reshard packs, halo-padded buffers, ...) with its collective ops LEFT OUT — the exchanged regions
hold whatever the arena held, so the output is meaningless; kernel timings do not depend on the data.  This is the compute term of
DESIGN.md §6's projection (the b = 1 unsharded forward of tools/profile_unet.py under-counts it: a sharded forward has ~2x the ops).
Usage: python tools/profile_tshard_rank.py <total frames> <R slices> <slice index> [latent_h latent_w]"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sd_webui_text2video_amd import configs  # noqa: E402
from sd_webui_text2video_amd import _lib as L, unet as U  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, COLLECTIVE_KINDS, TShardSpec  # noqa: E402
from profile_unet import random_weights_  # noqa: E402


def main():
    total, R, idx = (int(a) for a in sys.argv[1:4])
    H, W = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (32, 32)
    dev = torch.device("cuda:0")
    net = U.UNetSD(**configs.MODELSCOPE_UNET, init_weights=False).half().to(dev)
    random_weights_(net)
    spec = TShardSpec.make(total, R, idx)
    comp = net._compile(1, spec.frames, H, W, 77, "f32", "f32", "f16", shard=spec)
    net._programs[("profile_tshard_rank", spec)] = comp      # the packed weight set is the union over the registered programs
    net.refresh_weights(dev)
    prog = comp.prog
    ops = [op for op in prog.ops if op.kind not in COLLECTIVE_KINDS]
    n_coll = len(prog.ops) - len(ops)
    arena = torch.zeros(prog.arena.high + 256, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    bound = BoundProgram(prog, arena.data_ptr(), {k: v.data_ptr() for k, v in net._packed.items()}, ops=ops, reset_sync=True, stream=st)
    x = torch.randn(1, 4, spec.frames, H, W, device=dev)
    y = torch.randn(1, 77, net.context_dim, device=dev, dtype=torch.float16)
    t = torch.full((1,), 500.0, device=dev)
    out = torch.empty(1, 4, spec.frames, H, W, device=dev)
    ext = {L.EXT_X: x.data_ptr(), L.EXT_T: t.data_ptr(), L.EXT_CTX: y.data_ptr(), L.EXT_OUT: out.data_ptr()}
    for _ in range(3):
        bound.run(ext, st)
    torch.cuda.synchronize()
    n = 10
    t0 = time.time()
    for _ in range(n):
        bound.run(ext, st)
    torch.cuda.synchronize()
    wall = (time.time() - t0) / n * 1e3
    ms = bound.run_timed(ext, st)
    names = {1: "gemm", 2: "groupnorm", 3: "layernorm", 4: "attention", 6: "to_cl", 7: "from_cl", 8: "time_embed", 9: "copy2d", 11: "memset",
             18: "reshard_rows"}
    kinds = collections.defaultdict(lambda: [0.0, 0])
    for op, m in zip(ops, ms):
        k = names.get(op.kind, str(op.kind))
        if op.kind == 1 and op.i[16] == L.EPI_GN:
            k = "gemm+norm"
        kinds[k][0] += m
        kinds[k][1] += 1
    flops = sum(op.flops for op in ops)
    print(f"T-shard rank program: {total} frames over {R} slices, slice {idx} = {spec.frames} frames, {H}x{W} latent, b = 1: "
          f"{len(ops)} compute ops + {n_coll} collective ops (left out)  wall/forward {wall:.2f} ms  sum(op events) {sum(ms):.2f} ms  "
          f"{flops / 1e12:.3f} TFLOP -> {flops / (wall * 1e-3) / 1e12:.1f} TF/s")
    for k, (m, c) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:14s} {m:8.3f} ms {c:5d} ops")


if __name__ == "__main__":
    main()
