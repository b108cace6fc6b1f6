#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for l in L1 L2; do timeout 400 python tools/gemm_sweep.py $l > gpurun_out/gemm_sweep_$l.log 2>&1; cut -c1-1200 gpurun_out/gemm_sweep_$l.log; done
