#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "(TCC|TCP|TA|TD)_[A-Za-z_0-9]*sum" | sort -u | tr '\n' ' ' | head -c 1500; echo
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c_$tag -o g -- python $R/tools/gemm_sweep.py "L0 conv3x3" > $R/gpurun_out/pmc_c_$tag.log 2>&1; echo "pmc $tag exit $?"; tail -2 $R/gpurun_out/pmc_c_$tag.log | cut -c1-300
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_c_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "gemm" not in n: continue
        if "gemm2_kernel<4, 2, 2, 5" in n and row["Grid_Size"] == "98304": key = "cfg2 split1"
        elif "gemm_kernel<128, 64" in n and row["Grid_Size"] in ("491520",): key = "old 128x64 split1"
        else: continue
        agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"    {name:36s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
