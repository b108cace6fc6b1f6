"""Experiment (build: python tools/build_variant.py timing -DT2V_G2_TIMING -DT2V_G2_FEW; run with T2V_LIB_PATH=.../libt2v_hip_timing.so):
per-wave cycles parked in `s_waitcnt vmcnt` (operand DMA), `s_barrier` and the first fragment `ds_read`s of every k-tile of gemm2_kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
NW = {1: 8, 2: 8, 3: 8, 8: 12}
BM = {1: 256, 2: 256, 3: 128, 8: 192}
BN = {1: 256, 2: 320, 3: 256, 8: 320}
for (M, N, K, tile, conv) in [(8192, 8192, 8192, 1, None), (8192, 8320, 8192, 2, None), (8192, 8320, 8192, 8, None), (8192, 8192, 8192, 3, None),
                              (49152, 320, 2880, 2, dict(Hin=32, Win=32, Cin=320, stride=1, up=0, Hout=32, Wout=32)),
                              (49152, 320, 2880, 8, dict(Hin=32, Win=32, Cin=320, stride=1, up=0, Hout=32, Wout=32)),
                              (49152, 320, 320, 8, None), (49152, 2560, 320, 2, None)]:
    P = Program()
    P.force_tile = tile
    a = P.alloc(M, K if conv is None else conv["Cin"], "f16")
    out = P.alloc(M, N, "f16")
    op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, allow_splitk=False, gather=L.GATHER_PLAIN if conv is None else L.GATHER_CONV3X3, conv=conv)
    tiles = -(-M // BM[tile]) * -(-N // BN[tile])
    ws = P.alloc(tiles * NW[tile], 4, "f32")
    op.p[6] = ws.ref                      # the timing build writes its per-wave counters to the (unused) split-K workspace pointer
    arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
    arena[a.ref.off: a.ref.off + a.rows * a.ld * 2].view(torch.float16).normal_(0, 1)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr()})
    st = torch.cuda.current_stream(dev).cuda_stream
    bp.run({}, st); bp.run({}, st)
    torch.cuda.synchronize()
    t = arena[ws.ref.off: ws.ref.off + tiles * NW[tile] * 16].view(torch.float32).view(-1, 4).float().mean(dim=0).tolist()
    kt = K // 64
    print(f"tile {tile} {'conv' if conv else 'plain'} M{M} N{N} K{K}: per k-tile per wave: vmcnt wait {t[0] / kt:7.0f}  barrier {t[1] / kt:7.0f}  first ds_reads {t[2] / kt:6.0f}  "
          f"| whole main loop {t[3] / kt:7.0f} cycles per k-tile ({kt} k-tiles)", flush=True)
