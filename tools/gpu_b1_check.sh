#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -k "c1 or 24f or c2" -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/pytest_b1.log 2>&1; echo "pytest exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/pytest_b1.log | tail -n 8
for v in 0 1; do T2V_TILE8=$v timeout 300 python tools/profile_unet.py 24 32 32 1 > gpurun_out/b1_ab_$v.log 2>&1; echo "== b=1 T2V_TILE8=$v"; sed -n 4p gpurun_out/b1_ab_$v.log; done
T2V_TILE8=1 timeout 300 python tools/profile_unet.py 32 32 32 1 > gpurun_out/b1_32f.log 2>&1; echo "== b=1 32 frames (one T slice of the 125-frame clip)"; sed -n 4,5p gpurun_out/b1_32f.log
