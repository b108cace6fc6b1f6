#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm2" > gpurun_out/pytest_gemm2.log 2>&1
echo "pytest exit $?"; tail -n 8 gpurun_out/pytest_gemm2.log
timeout 900 python tools/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep exit $?"; cat gpurun_out/gemm_sweep.log | tail -40
