"""Aggregate rocprofv3 --pmc CSV output per kernel: HBM bytes per launch (gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request => x2; units KiB)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0][-70:]
            c = row["Counter_Name"]
            agg[name][c] += float(row["Counter_Value"])
            cnt[name][c] += 1
print(f"{'kernel':72s} {'launches':>8s} {'fetch MB/launch (x2 corr.)':>26s} {'write MB/launch':>16s}")
for name in sorted(agg, key=lambda n: -(agg[n].get("FETCH_SIZE", 0) + agg[n].get("WRITE_SIZE", 0))):
    f = agg[name].get("FETCH_SIZE", 0.0)
    w = agg[name].get("WRITE_SIZE", 0.0)
    nf = max(cnt[name].get("FETCH_SIZE", 0), 1)
    nw = max(cnt[name].get("WRITE_SIZE", 0), 1)
    print(f"{name:72s} {max(nf, nw):8d} {2 * f * 1024 / nf / 1e6:26.2f} {w * 1024 / nw / 1e6:16.2f}")
