#!/bin/bash
# Round 3, GPU pass K: csrc/comm.hip under real multi-process runs on one GPU (tests/fake_rccl), rel-pos op test variants.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fake_rccl.py -q -rP --tb=short -p no:cacheprovider > gpurun_out/k_fake_rccl.log 2>&1; echo "fake rccl exit $?"; grep -E "library collectives|passed|failed|Error|error" gpurun_out/k_fake_rccl.log | tail -n 12
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rccl.py -q --tb=short -p no:cacheprovider -k "relpos or rccl" > gpurun_out/k_ops.log 2>&1; echo "ops exit $?"; tail -n 3 gpurun_out/k_ops.log
