#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|LDSBankConflict" | sort -u > $R/gpurun_out/counters.txt
wc -l $R/gpurun_out/counters.txt
grep -E "MFMA|WAIT|LDS|BUSY_CYCLES|WAVE_CYCLES|ACTIVE_INST" $R/gpurun_out/counters.txt | tr '\n' ' '
echo
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm -o g -- python $R/tools/gemm_sweep.py "L0 conv3x3" > $R/gpurun_out/pmc_gemm.log 2>&1; echo "pmc exit $?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm2 -o g -- python $R/tools/gemm_sweep.py "L0 conv3x3" > $R/gpurun_out/pmc_gemm2.log 2>&1; echo "pmc2 exit $?"
cd $R
tail -3 gpurun_out/pmc_gemm.log
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_gemm", "gpurun_out/pmc_gemm2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "gemm" not in n: continue
            key = n[n.find("gemm"):][:60] + " grid=" + row["Grid_Size"]
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
            agg[key]["_dur"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, c in agg.items():
        print(k)
        for name, v in sorted(c.items()):
            print(f"    {name:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
