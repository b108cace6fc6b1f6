"""Compare tools/gemm_sweep.py output files with Program.choose_tile: for every swept shape print the policy's (tile, split), its TF/s
and the best measured configuration; flag shapes where the best is > 7 % ahead.  Usage: python tools/sweep_vs_policy.py <sweep files...>"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("T2V_DEVICE_CUS", "256")
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd.program import Program  # noqa: E402

for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r"^(L\d [^|]*?)\s+(\d+)\s+(\d+)\s+(\d+) \| (.*)$", line)
        if not m:
            continue
        label, M, N, K = m.group(1).strip(), int(m.group(2)), int(m.group(3)), int(m.group(4))
        res = {}
        for item in m.group(5).split("|"):
            mm = re.match(r"\s*(\d+):(\d+)\s+(\d+)", item)
            if mm:
                res[(int(mm.group(1)), int(mm.group(2)))] = int(mm.group(3))
        gather = L.GATHER_CONV3X3 if "conv" in label and "tconv" not in label else (L.GATHER_TCONV3 if "tconv" in label else L.GATHER_PLAIN)
        P = Program()
        P.small_rank_tiles = os.environ.get("SWEEP_RANK") == "1"
        pol = P.choose_tile(M, N, K, gather)
        best = max(res.items(), key=lambda kv: kv[1])
        got = res.get(pol)
        flag = "  <-- best is +%d %%" % round(100 * (best[1] / got - 1)) if got and best[1] > 1.07 * got else ("" if got else "  (policy configuration not in the sweep)")
        print(f"{os.path.basename(path):28s} {label:14s} {M:6d} {N:6d} {K:6d}  policy {pol[0]}:{pol[1]} {got if got else '?':>5}  best {best[0][0]}:{best[0][1]} {best[1]:5d}{flag}")
