#!/bin/bash
# Round 3, GPU pass I: VideoCrafter lowering with the LayerNorms fused into the producing GEMMs' epilogues (32x32 level).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_videocrafter.py tests/test_gpu_fullsize.py -q -rP --tb=short -p no:cacheprovider -k "videocrafter or lvdm or c4" > gpurun_out/i_tests.log 2>&1; echo "tests exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/i_tests.log | tail -n 12
timeout 300 python tools/profile_unet.py 16 32 32 2 lvdm > gpurun_out/i_prof_lvdm.log 2>&1; sed -n 4,16p gpurun_out/i_prof_lvdm.log
timeout -k 10 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/i_bench_lvdm.json 2> gpurun_out/i_bench_lvdm.err; echo "lvdm exit $?"; cut -c1-200 gpurun_out/i_bench_lvdm.json
