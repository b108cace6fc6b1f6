"""Round 6: the staggered two-group / region-staged schedule (tiles 22 / 23 / 24) against its lock-step bases (1 / 2 / 3) and the 12-wave
tiles (8 / 9) on the shapes that matter: TF/s, median of 12 back-to-back launches (per-op HIP events), plus a correctness check of every
(tile, shape) pair against torch (fp32 matmul / conv of the fp16 operands).  GEMM_R6_TILES="1,22" restricts the tiles, GEMM_R6_CHECK=0
skips the check."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, Program, Ref  # noqa: E402

dev = torch.device("cuda:0")
c0 = dict(Hin=32, Win=32, Cin=320, stride=1, up=0, Hout=32, Wout=32)
c1 = dict(Hin=16, Win=16, Cin=640, stride=1, up=0, Hout=16, Wout=16)
c2 = dict(Hin=8, Win=8, Cin=1280, stride=1, up=0, Hout=8, Wout=8)
only = os.environ.get("GEMM_R6_TILES")
only = None if not only else {int(t) for t in only.split(",")}
check = os.environ.get("GEMM_R6_CHECK", "1") != "0"
SHAPES = [("square 4096", 4096, 4096, 4096, None, 0, (1, 22, 3, 24)),
          ("square 8192", 8192, 8192, 8192, None, 0, (1, 22, 3, 24)),
          ("8192x8320x8192", 8192, 8320, 8192, None, 0, (2, 23, 8)),
          ("L0 conv3x3", 49152, 320, 2880, c0, 0, (2, 23, 8)),
          ("L0 conv cat", 49152, 320, 5760, dict(c0, Cin=640), 0, (2, 23, 8)),
          ("L0 geglu", 49152, 2560, 320, None, 1, (2, 23, 8)),
          ("L0 ff2", 49152, 320, 1280, None, 0, (2, 23, 8)),
          ("L0 qkv", 49152, 960, 320, None, 0, (2, 23, 8)),
          ("L1 geglu", 12288, 5120, 640, None, 1, (2, 23, 1, 22)),
          ("L1 conv3x3 s2", 12288, 640, 5760, c1, 0, (2, 23, 8)),
          ("L1 qkv", 12288, 1920, 640, None, 0, (9, 3, 24, 23)),
          ("L1 ff2", 12288, 640, 2560, None, 0, (0, 3, 24, 23)),
          ("L2 geglu", 3072, 10240, 1280, None, 1, (1, 22, 2, 23)),
          ("L2 conv3x3 s2", 3072, 1280, 11520, c2, 0, (3, 24)),
          ("L2 qkv", 3072, 3840, 1280, None, 0, (1, 22, 3, 24, 9)),
          ("L2 ff2", 3072, 1280, 5120, None, 0, (5, 3, 24))]
for label, M, N, K, conv, epi, tiles in SHAPES:
    res = []
    a16 = w = b = None
    for tile in tiles:
        if only is not None and tile not in only:
            continue
        P = Program()
        P.force_tile = tile
        split = 2 if "s2" in label else 1
        P.choose_tile = lambda *a, _t=tile, _s=split, **kw: (_t, _s)
        cin = K if conv is None else conv["Cin"]
        a = P.alloc(M, cin, "f16")
        out = P.alloc(M, N // 2 if epi else N, "f16")
        P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_PLAIN if conv is None else L.GATHER_CONV3X3,
               conv=conv, epi=epi)
        nops = len(P.ops)
        P.ops = P.ops * 12
        arena = torch.empty(P.arena.high + 256, dtype=torch.uint8, device=dev)
        g = torch.Generator(device=dev).manual_seed(7)
        arena.view(torch.float16).normal_(0, 1, generator=g)
        if w is None:
            w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).half()
            b = torch.randn(N, device=dev, generator=g)
        bp = BoundProgram(P, arena.data_ptr(), {"w": w.data_ptr(), "b": b.data_ptr()})
        st = torch.cuda.current_stream(dev).cuda_stream
        try:
            bp.run({}, st)
        except L.T2VError:                     # a development build without this tile
            res.append(f"{tile}:    n/a")
            continue
        torch.cuda.synchronize()
        tm = bp.run_timed({}, st)
        ms = sorted(sum(tm[i * nops:(i + 1) * nops]) for i in range(12))[6]       # (split-K: the GEMM and its reduction)
        err = ""
        if check:
            av = arena[a.ref.off:a.ref.off + M * cin * 2].view(torch.float16).view(M, cin).float()
            got = arena[out.ref.off:out.ref.off + out.rows * out.ld * 2].view(torch.float16).view(out.rows, out.ld)[:, :out.cols].float()
            rows = torch.arange(0, M, max(1, M // 1024), device=dev)          # a sample of rows (every tile row is hit over the stride)
            if conv is None:
                ref = av[rows] @ w.float().t() + b
            else:
                Bn = M // (conv["Hout"] * conv["Wout"])
                x = av.view(Bn, conv["Hin"], conv["Win"], cin).permute(0, 3, 1, 2)
                wk = w.float().view(N, cin // 64, 9, 64).permute(0, 1, 3, 2).reshape(N, cin, 3, 3)     # packing.conv3x3: [Co][chunk][tap][64]
                ref = torch.nn.functional.conv2d(x, wk, b, stride=conv["stride"], padding=1).permute(0, 2, 3, 1).reshape(M, N)[rows]
            if epi:
                # packed GEGLU columns: blocks of 16 = 8 value | 8 gate channels
                r = ref.view(len(rows), N // 16, 2, 8)
                ref = (r[:, :, 0] * torch.nn.functional.gelu(r[:, :, 1])).reshape(len(rows), N // 2)
            e = (got[rows] - ref).norm() / ref.norm()
            err = f" ({e:.1e})"
            if not e < 2e-3:
                err += " FAIL"
        res.append(f"{tile}: {2.0 * M * N * K / ms / 1e9:6.0f}{err}")
    if res:
        print(f"{label:16s} {M:6d} {N:6d} {K:6d} | " + " | ".join(res), flush=True)
