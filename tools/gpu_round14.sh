#!/bin/bash
# GPU parity suite (ops + e2e) then per-op UNet profile
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -n 12 gpurun_out/pytest_gpu.log
timeout 300 python tools/profile_unet.py 24 32 32 2 > gpurun_out/profile_unet.log 2>&1; echo "profile exit $?"; head -n 40 gpurun_out/profile_unet.log
