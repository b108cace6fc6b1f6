"""Mean of every collected counter per kernel from rocprofv3 --pmc CSV output.  Usage: python tools/pmc_generic_post.py <dir> [filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", r["Kernel_Name"])
        if flt and flt not in name:
            continue
        key = (name, r["Grid_Size"], r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1] += 1
for (name, grid, ctr), (tot, n) in sorted(acc.items()):
    print(f"{name[:60]:60s} grid {grid:>8s} {ctr:28s} {tot / n:16.1f}  (n={n})")
