#!/bin/bash
# Round 6: SQ / GRBM counter passes on the calibration GEMM (lock-step tile 1 vs staggered tile 22; whole chip vs 16 CUs; random vs zero
# operands).  Separate --pmc runs with --kernel-trace only.   gpurun --timeout 900 -- 'bash tools/gpu_counters_r6.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export T2V_LIB_PATH=$R/tools/variants/libt2v_hip_dev.so
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i + 1))
  rm -rf $R/gpurun_out/pmc6_$i
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc6_$i -- python $R/tools/pmc_gemm_target_r6.py > $R/gpurun_out/pmc6_$i.log 2>&1
  echo "pass $i exit $?"; tail -n 2 $R/gpurun_out/pmc6_$i.log | cut -c1-200
done
cd $R
python tools/pmc_r6_post.py gpurun_out/pmc6_1 gpurun_out/pmc6_2 > gpurun_out/r6_pmc_gemm_counters.txt 2>&1
find gpurun_out/pmc6_1 gpurun_out/pmc6_2 -name "*.csv" -size +8M -delete
cat gpurun_out/r6_pmc_gemm_counters.txt | cut -c1-170
