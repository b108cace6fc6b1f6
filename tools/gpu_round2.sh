#!/bin/bash
# Targeted GPU check of the round-2 work: poison probe, new / changed test files with their printed measurements.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/poison_probe.py vae > gpurun_out/poison_vae.log 2>&1; tail -n 6 gpurun_out/poison_vae.log
timeout 300 python tools/poison_probe.py unet > gpurun_out/poison_unet.log 2>&1; tail -n 6 gpurun_out/poison_unet.log
timeout 1500 python -m pytest ${T2V_TESTS:-tests/test_gpu_boundary.py tests/test_gpu_fullsize.py tests/test_gpu_videocrafter.py tests/test_gpu_multiproc.py} "tests/test_gpu_ops.py::test_groupnorm_split_phases_two_parts" "tests/test_gpu_e2e.py::test_tsharded_forward_two_shards_emulated_on_one_gpu" -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/pytest_round2.log 2>&1
echo "pytest exit $?"; grep -E "rel-L2|identical|passed|failed|FAILED|ERROR" gpurun_out/pytest_round2.log | tail -n 40
timeout 300 python tools/gemm_epi.py > gpurun_out/gemm_epi.log 2>&1; grep -E "tile (2|5|0):|tile 3:" gpurun_out/gemm_epi.log | head -n 24
timeout 600 python bench.py --no-cpu-baseline --also-batched 0 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('gpurun_out/bench_quick.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['unet_step_ms_events'], r['whole_video'])"
timeout 300 python tools/profile_unet.py > gpurun_out/profile_unet.log 2>&1; sed -n 4,17p gpurun_out/profile_unet.log
