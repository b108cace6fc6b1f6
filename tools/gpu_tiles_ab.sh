#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1 0 1; do
  T2V_TILE8=$v timeout 300 python tools/profile_unet.py > gpurun_out/tile8_ab_$v.log 2>&1
  echo "== T2V_TILE8=$v"; sed -n 4p gpurun_out/tile8_ab_$v.log; grep -E "^gemm" gpurun_out/tile8_ab_$v.log | awk '{printf "   %-14s %8s ms\n", $1, $2}'
done
