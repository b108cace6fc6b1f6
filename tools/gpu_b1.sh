#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
SWEEP_BATCH=1 timeout 300 python tools/gemm_sweep.py L0 > gpurun_out/gemm_sweep_L0_b1.log 2>&1; cut -c1-260 gpurun_out/gemm_sweep_L0_b1.log; grep -o "8:[0-9]  *[0-9]*" gpurun_out/gemm_sweep_L0_b1.log | tr '\n' ' '; echo
for v in 0 1; do
  T2V_TILE8=$v timeout 300 python tools/profile_unet.py 24 32 32 1 > gpurun_out/b1_ab_$v.log 2>&1; echo "== b=1 T2V_TILE8=$v"; sed -n 4p gpurun_out/b1_ab_$v.log
  T2V_TILE8=$v timeout 300 python tools/profile_unet.py > gpurun_out/b2_ab_$v.log 2>&1; echo "== b=2 T2V_TILE8=$v"; sed -n 4p gpurun_out/b2_ab_$v.log
done
