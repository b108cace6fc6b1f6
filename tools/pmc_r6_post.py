"""Round 6: per (kernel, grid, launch group) means of the collected counters plus the kernel duration from the same pass and the
effective shader clock GRBM_GUI_ACTIVE / duration.  Launches are grouped in order of appearance (4 launches per case of
tools/pmc_gemm_target_r6.py).  Usage: python tools/pmc_r6_post.py <dir> <dir> ..."""
import csv
import glob
import re
import sys
from collections import OrderedDict, defaultdict

CASES = ["normal 8192^3 tile 1", "normal 8192^3 tile 22", "normal 1024^2x8192 tile 1 (16 CUs)", "normal 1024^2x8192 tile 22 (16 CUs)",
         "zeros 8192^3 tile 1", "zeros 8192^3 tile 22", "zeros 1024^2x8192 tile 1 (16 CUs)", "zeros 1024^2x8192 tile 22 (16 CUs)"]
table = defaultdict(dict)
for d in sys.argv[1:]:
    dur = {}
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "gemm2_kernel" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    per = OrderedDict()
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "gemm2_kernel" not in r["Kernel_Name"]:
                continue
            per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per, key=int)
    for i, did in enumerate(ids):
        case = CASES[(i // 4) % len(CASES)]
        for k, v in per[did].items():
            table[case].setdefault(k, []).append(v)
        if did in dur:
            table[case].setdefault("duration_us(" + d.split("/")[-1] + ")", []).append(dur[did])
for case in CASES:
    if case not in table:
        continue
    t = {k: sum(v) / len(v) for k, v in table[case].items()}
    print("## " + case)
    for k in sorted(t):
        print(f"   {k:34s} {t[k]:18.1f}")
    durs = [v for k, v in t.items() if k.startswith("duration_us")]
    if "GRBM_GUI_ACTIVE" in t and durs:
        print(f"   effective clock = GRBM_GUI_ACTIVE / duration = {t['GRBM_GUI_ACTIVE'] / durs[-1] / 1e3:6.3f} GHz")
    if "SQ_WAVE_CYCLES" in t:
        wc = t["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if k in t:
                print(f"   {k} / SQ_WAVE_CYCLES = {100 * t[k] / wc:5.1f} %")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in t and "SQ_BUSY_CYCLES" in t:
            print(f"   MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 32) = {100 * t['SQ_VALU_MFMA_BUSY_CYCLES'] / (t['SQ_BUSY_CYCLES'] * 32):5.1f} %   (fraction of all 1024 SIMDs x busy time)")
