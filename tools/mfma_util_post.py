"""MFMA utilisation per kernel from a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass, summarised by
tools/pmc_generic_post.py (input: its text output).  SQ_VALU_MFMA_BUSY_CYCLES = MFMA instructions x passes x 4 cycles summed
over the chip's 1024 SIMDs (8192^3 on the 32x32x16 instruction: exactly 2^30); GRBM_GUI_ACTIVE = kernel duration in cycles
summed over the 8 XCDs.  utilisation = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8)."""
import re
import sys
from collections import defaultdict

rows = defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(.{60}) grid\s+(\d+) (\S+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
    if m:
        rows[(m.group(1).strip(), m.group(2))][m.group(3)] = (float(m.group(4)), int(m.group(5)))
OURS = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"gemm|attn_kernel|gn_|layernorm|copy2d|splitk_reduce|softmax_rows|ncthw|time_embed|ddim_step|lincomb|embed_rows")
per = defaultdict(lambda: [0.0, 0.0, 0])
for (name, grid), c in rows.items():
    if not OURS.search(name):      # leave out the torch kernels of the weight initialisation in the profiling target
        continue
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    gui, n = c["GRBM_GUI_ACTIVE"]
    mf, _ = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    per[name][0] += mf * n
    per[name][1] += gui * n
    per[name][2] += n
tm = sum(v[0] for v in per.values())
tg = sum(v[1] for v in per.values())
print(f"{'kernel':62s} {'launches':>8s} {'share of GPU time':>18s} {'MFMA utilisation':>17s}")
for name, (mf, gui, n) in sorted(per.items(), key=lambda x: -x[1][1]):
    if gui / tg < 0.002:
        continue
    print(f"{name:62s} {n:8d} {100 * gui / tg:17.1f}% {100 * mf / (128 * gui):16.1f}%")
print(f"{'all library kernels of the run':62s} {sum(v[2] for v in per.values()):8d} {100.0:17.1f}% {100 * tm / (128 * tg):16.1f}%")
