#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -k "gemm2" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_tile9.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/pytest_tile9.log
bash tools/gpu_tile8_ab.sh
