"""Idle time between consecutive kernels of the UNet forward, from a rocprofv3 --kernel-trace CSV (no counters):
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/pmc_target.py 4
  python tools/gap_post.py <dir>
Takes the longest run of kernels without a host-side pause > 2 ms (the back-to-back forwards) and prints the busy time (sum of kernel
durations), the span and the distribution of the start-after-previous-end gaps."""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
segs, cut = [], 0
for i in range(1, len(rows) + 1):
    if i == len(rows) or rows[i][0] - rows[i - 1][1] > 2_000_000:
        segs.append(rows[cut:i])
        cut = i
fw = max(segs, key=len)              # the back-to-back forwards (host-side pauses > 2 ms split the trace)
busy = sum(e - s for s, e, _ in fw)
span = fw[-1][1] - fw[0][0]
gaps = sorted(max(0, fw[i][0] - fw[i - 1][1]) for i in range(1, len(fw)))
n = len(gaps)
print(f"{path}\nlongest back-to-back run: {len(fw)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms "
      f"({100 * (span - busy) / span:.1f}%)")
print(f"gap us: median {gaps[n // 2] / 1e3:.2f}, p10 {gaps[n // 10] / 1e3:.2f}, p90 {gaps[9 * n // 10] / 1e3:.2f}, max {gaps[-1] / 1e3:.1f}, "
      f"mean {sum(gaps) / n / 1e3:.2f}")
short = sorted((e - s) for s, e, _ in fw)
print(f"kernel duration us: median {short[len(short) // 2] / 1e3:.1f}, p10 {short[len(short) // 10] / 1e3:.1f}, "
      f"<5us: {sum(1 for d in short if d < 5000)}, <10us: {sum(1 for d in short if d < 10000)}")
