#!/bin/bash
# Round 3, GPU pass J: relative-position temporal attention on MFMA (VideoCrafter) — op tests, network tests, A/B.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -k "relpos" > gpurun_out/j_ops.log 2>&1; echo "ops exit $?"; tail -n 12 gpurun_out/j_ops.log
timeout -k 10 600 python -m pytest tests/test_gpu_videocrafter.py tests/test_gpu_fullsize.py -q -rP --tb=short -p no:cacheprovider -k "videocrafter or lvdm or c4" > gpurun_out/j_tests.log 2>&1; echo "tests exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/j_tests.log | tail -n 12
timeout 300 python tools/profile_unet.py 16 32 32 2 lvdm > gpurun_out/j_prof_lvdm.log 2>&1; sed -n 4,12p gpurun_out/j_prof_lvdm.log
T2V_RELPOS_MFMA=0 timeout 300 python tools/profile_unet.py 16 32 32 2 lvdm > gpurun_out/j_prof_lvdm_valu.log 2>&1; sed -n 4,12p gpurun_out/j_prof_lvdm_valu.log
timeout -k 10 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/j_bench_lvdm.json 2> gpurun_out/j_bench_lvdm.err; echo "lvdm exit $?"; cut -c1-200 gpurun_out/j_bench_lvdm.json
