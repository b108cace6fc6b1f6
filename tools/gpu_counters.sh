#!/bin/bash
# SQ counter passes on the GEMM kernel alone: where do the main loop's cycles go?  Separate --pmc runs with
# --kernel-trace only (MI355X_MICROARCH.md: 8 SQ counters per pass).   gpurun --timeout 900 -- 'bash tools/gpu_counters.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rm -rf $R/gpurun_out/pmc_sq$i
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq$i -- python $R/tools/pmc_gemm_target.py > $R/gpurun_out/pmc_sq$i.log 2>&1
  echo "pass $i exit $?"; tail -n 2 $R/gpurun_out/pmc_sq$i.log | cut -c1-200
done
cd $R
(python tools/pmc_generic_post.py gpurun_out/pmc_sq1 gemm2; python tools/pmc_generic_post.py gpurun_out/pmc_sq2 gemm2) > gpurun_out/pmc_gemm_counters.txt 2>&1
find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*.csv" -size +8M -delete
cat gpurun_out/pmc_gemm_counters.txt | cut -c1-170
