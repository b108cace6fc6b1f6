#!/bin/bash
# Targeted GPU check of the round-2 work: poison probe, new / changed test files with their printed measurements.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/poison_probe.py vae > gpurun_out/poison_vae.log 2>&1; tail -n 6 gpurun_out/poison_vae.log
timeout 300 python tools/poison_probe.py unet > gpurun_out/poison_unet.log 2>&1; tail -n 6 gpurun_out/poison_unet.log
timeout 1500 python -m pytest ${T2V_TESTS:-tests/test_gpu_boundary.py tests/test_gpu_fullsize.py tests/test_gpu_videocrafter.py tests/test_gpu_multiproc.py} "tests/test_gpu_ops.py::test_groupnorm_split_phases_two_parts" "tests/test_gpu_e2e.py::test_tsharded_forward_two_shards_emulated_on_one_gpu" -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/pytest_round2.log 2>&1
echo "pytest exit $?"; grep -E "rel-L2|identical|passed|failed|FAILED|ERROR" gpurun_out/pytest_round2.log | tail -n 40
