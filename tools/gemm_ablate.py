"""Ablation of the large-tile GEMM main loop on one shape: time with DMA / MFMA / ds_read removed."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import _lib as L
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref
dev = torch.device("cuda:0")
B, F, hw, C = 2, 24, 32, 320
M, N, K = B * F * hw * hw, C, 9 * C
conv = dict(Hin=hw, Win=hw, Cin=C, stride=1, up=0, Hout=hw, Wout=hw)
for tile in (2, 1, 3):
    for gather, KK, cv in ((L.GATHER_CONV3X3, K, conv), (L.GATHER_PLAIN, K, None)):
        row = []
        for dbg in (0,):
            P = Program(); P.force_tile = tile
            a = P.alloc(M, KK if gather == L.GATHER_PLAIN else C, "f16")
            out = P.alloc(M, N, "f32")
            P.choose_tile = lambda *aa, _t=tile, **kw: (_t, 1)
            op = P.gemm("g", a, Ref("weight", 0, "w"), N, KK, out, bias=Ref("weight", 0, "b"), gather=gather, conv=cv)
            op.i[23] = dbg
            arena = torch.empty(P.arena.high + 256, dtype=torch.uint8, device=dev)
            arena.view(torch.float16).normal_(0, 1)
            w = (torch.randn(N, KK, device=dev) / math.sqrt(KK)).half(); bv = torch.randn(N, device=dev)
            bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr(), "b": bv.data_ptr()})
            st = torch.cuda.current_stream(dev).cuda_stream
            P.ops = P.ops * 12            # the same op 12x in ONE plan: per-op HIP events, no host launch gaps
            bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr(), "b": bv.data_ptr()})
            bp.run({}, st)
            ms = sorted(bp.run_timed({}, st))
            row.append(f"dbg{dbg}: {ms[len(ms) // 2] * 1e3:7.1f}us")
        print(f"tile {tile} gather {gather} M{M} N{N} K{KK} | " + " | ".join(row), flush=True)
print("dbg bits: 1 = no DMA, 2 = no MFMA, 4 = no ds_read (all fragment reads hit address 0)")
