#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -k "fused_layernorm or gemm2_plain" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_lnfuse.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/pytest_lnfuse.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_fullsize.py -k "c1_24f_forward" -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/pytest_lnfuse2.log 2>&1; echo "pytest exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/pytest_lnfuse2.log | tail -n 3
for v in 0 1 0 1; do
  T2V_LN_FUSE=$v timeout 300 python tools/profile_unet.py > gpurun_out/lnfuse_ab_$v.log 2>&1
  echo "== T2V_LN_FUSE=$v"; sed -n 4p gpurun_out/lnfuse_ab_$v.log; grep -E "^(gemm/plain|layernorm)" gpurun_out/lnfuse_ab_$v.log
done
