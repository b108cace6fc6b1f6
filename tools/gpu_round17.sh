#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -n 6 gpurun_out/pytest_gpu.log
timeout 600 python tools/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep exit $?"; cat gpurun_out/gemm_sweep.log
timeout 300 python tools/profile_unet.py 24 32 32 2 > gpurun_out/profile_unet.log 2>&1; echo "profile exit $?"; head -n 45 gpurun_out/profile_unet.log
