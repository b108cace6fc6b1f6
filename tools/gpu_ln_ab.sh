#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -k "layernorm" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_ln.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/pytest_ln.log
for cap in 1000000 0 1000000 0 1024 4096; do
  T2V_LN_CAP=$cap timeout 300 python tools/profile_unet.py > gpurun_out/ln_ab_$cap.log 2>&1
  echo "== T2V_LN_CAP=$cap"; sed -n 4p gpurun_out/ln_ab_$cap.log; grep -E "^layernorm" gpurun_out/ln_ab_$cap.log
done
