"""Round 6 experiment (build: python tools/build_variant.py timing -DT2V_G2_TIMING -DT2V_G2_DEV -DT2V_G2_DEVMIN; run with
T2V_LIB_PATH=tools/variants/libt2v_hip_timing.so): cycles (s_memtime) per phase of the staggered main loop (tile 22, PP == 4), split
into [load segment | wait at the first barrier | MFMA segment | wait at the second barrier], mean over waves, per k-tile."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
GEO = {22: (8, 256, 256, 4), 23: (8, 256, 320, 5)}
for (M, N, K, tile) in [(8192, 8192, 8192, 22), (4096, 4096, 4096, 22), (3072, 10240, 1280, 22)]:
    NW, BM, BN, PH = GEO[tile]
    P = Program()
    P.force_tile = tile
    a = P.alloc(M, K, "f16")
    out = P.alloc(M, N, "f16")
    op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, allow_splitk=False)
    tiles = -(-M // BM) * -(-N // BN)
    W = 4 * PH + 1
    ws = P.alloc(tiles * NW, W, "f32")
    op.p[6] = ws.ref
    arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
    arena[a.ref.off: a.ref.off + a.rows * a.ld * 2].view(torch.float16).normal_(0, 1)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr()})
    st = torch.cuda.current_stream(dev).cuda_stream
    bp.run({}, st); bp.run({}, st)
    torch.cuda.synchronize()
    ms = sorted(bp.run_timed({}, st))[0]
    t = arena[ws.ref.off: ws.ref.off + tiles * NW * W * 4].view(torch.float32).view(tiles, NW, W).float()
    kt = K // 64
    for grp in (0, 1):
        m = t[:, 4 * grp:4 * grp + 4].mean(dim=(0, 1)) / kt
        print(f"tile {tile} M{M} N{N} K{K} group {grp}: " + " | ".join(
            f"ph{q}: load {m[4*q]:4.0f} bar {m[4*q+1]:4.0f} mfma {m[4*q+2]:4.0f} bar {m[4*q+3]:4.0f}" for q in range(PH)) +
            f" | k-tile {m[4*PH]:5.0f} cycles; {2.0*M*N*K/ms/1e9:5.0f} TF/s (instrumented)", flush=True)
