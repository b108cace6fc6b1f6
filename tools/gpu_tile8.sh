#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -k "gemm2" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_tile8.log 2>&1; echo "pytest exit $?"; tail -n 5 gpurun_out/pytest_tile8.log
timeout 600 python tools/gemm_sweep.py L0 > gpurun_out/gemm_sweep_L0.log 2>&1; cat gpurun_out/gemm_sweep_L0.log | cut -c1-330
