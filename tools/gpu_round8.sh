#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_ops.log 2>&1
echo "pytest exit $?"; tail -n 6 gpurun_out/pytest_ops.log
timeout 900 python tools/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep exit $?"; cat gpurun_out/gemm_sweep.log | tail -30
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm2 -o g -- python $R/tools/gemm_sweep.py "L0 conv3x3" > $R/gpurun_out/pmc_gemm2.log 2>&1; echo "pmc2 exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm -o g -- python $R/tools/gemm_sweep.py "L0 conv3x3" > $R/gpurun_out/pmc_gemm.log 2>&1; echo "pmc exit $?"
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_gemm", "gpurun_out/pmc_gemm2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "gemm2_kernel<4, 2, 2, 5" not in n or row["Grid_Size"] != "98304": continue
            agg["cfg2 split1"][row["Counter_Name"]].append(float(row["Counter_Value"]))
            agg["cfg2 split1"]["_dur"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, c in agg.items():
        print(k)
        for name, v in sorted(c.items()):
            print(f"    {name:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
