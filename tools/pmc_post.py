"""Aggregate rocprofv3 --pmc CSV output per kernel: HBM bytes per launch.
gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads => x2 (WRITE_SIZE is taken as reported).
Usage: python tools/pmc_post.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/ subdirs>"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)_kernel", name)
    if m:
        return m.group(1) + "_kernel" + ("<f16>" if "IDF16_" in name else "<f32>" if "IfE" in name else "")
    return name.split("(")[0][:78]


agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(float)
files = glob.glob(root + "/pmc_FETCH_SIZE/**/*counter_collection.csv", recursive=True) + \
    glob.glob(root + "/pmc_WRITE_SIZE/**/*counter_collection.csv", recursive=True)
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = short(row["Kernel_Name"])
            c = row["Counter_Name"]
            agg[name][c] += float(row["Counter_Value"])
            cnt[name][c] += 1
            if c == "FETCH_SIZE":
                dur[name] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
tot_f = sum(2 * a.get("FETCH_SIZE", 0.0) * 1024 for a in agg.values())
tot_w = sum(a.get("WRITE_SIZE", 0.0) * 1024 for a in agg.values())
print(f"# total over the run: fetch {tot_f / 1e9:.2f} GB (x2-corrected), write {tot_w / 1e9:.2f} GB")
print(f"{'kernel':80s} {'launches':>8s} {'fetch MB/launch':>16s} {'write MB/launch':>16s} {'avg us (pmc run)':>17s}")
for name in sorted(agg, key=lambda n: -(2 * agg[n].get("FETCH_SIZE", 0) + agg[n].get("WRITE_SIZE", 0))):
    f = agg[name].get("FETCH_SIZE", 0.0)
    w = agg[name].get("WRITE_SIZE", 0.0)
    nf = max(cnt[name].get("FETCH_SIZE", 0), 1)
    nw = max(cnt[name].get("WRITE_SIZE", 0), 1)
    print(f"{name:80s} {max(nf, nw):8d} {2 * f * 1024 / nf / 1e6:16.2f} {w * 1024 / nw / 1e6:16.2f} {dur[name] / nf / 1e3:17.1f}")

# ---- the GEMM family (bench.py's dominant kernel): average HBM bytes per launch -> JSON for bench.py
import json  # noqa: E402
fam = [n for n in agg if n.startswith("gemm_kernel") or n.startswith("gemm2_kernel")]
nl = sum(max(cnt[n].get("FETCH_SIZE", 0), cnt[n].get("WRITE_SIZE", 0)) for n in fam)
fb = sum(2 * agg[n].get("FETCH_SIZE", 0.0) * 1024 for n in fam)
wb = sum(agg[n].get("WRITE_SIZE", 0.0) * 1024 for n in fam)
summary = {"kernel_family": "gemm_kernel / gemm2_kernel (all instantiations)", "launches": nl,
           "fetch_bytes_per_launch": fb / max(nl, 1), "write_bytes_per_launch": wb / max(nl, 1),
           "hbm_bytes_per_launch": (fb + wb) / max(nl, 1),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/pmc_target.py; "
                     "KiB units, FETCH_SIZE x2 (MI355X_MICROARCH.md HBM section)"}
print("# " + json.dumps(summary))
if len(sys.argv) > 2:
    json.dump(summary, open(sys.argv[2], "w"), indent=1)
