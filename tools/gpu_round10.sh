#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_ops.log 2>&1
echo "pytest exit $?"; tail -n 6 gpurun_out/pytest_ops.log
timeout 900 python tools/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep exit $?"; cat gpurun_out/gemm_sweep.log | tail -30
