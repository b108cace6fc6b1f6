#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x -k "groupnorm or vae or layout" > gpurun_out/pytest_ops.log 2>&1
echo "pytest exit $?"; tail -n 6 gpurun_out/pytest_ops.log
timeout 300 python tools/gn_bench.py > gpurun_out/gn_bench.log 2>&1; echo "gn exit $?"; cat gpurun_out/gn_bench.log
timeout 300 python tools/profile_unet.py 24 32 32 2 > gpurun_out/profile_unet.log 2>&1; echo "profile exit $?"; head -n 20 gpurun_out/profile_unet.log
