#!/bin/bash
# Round 3, GPU pass F: one-pass hi+lo casts (K doubled / padding channels), fusion rule by level, smoke on deployed weights.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1; echo "smoke exit $?"; tail -n 5 gpurun_out/f_smoke.log
timeout 300 python tools/profile_unet.py > gpurun_out/f_prof.log 2>&1; echo "prof $?"; sed -n 4,14p gpurun_out/f_prof.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/f_ops.json
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_videocrafter.py tests/test_gpu_e2e.py -q -rP --tb=short -p no:cacheprovider > gpurun_out/f_tests.log 2>&1; echo "tests exit $?"; grep -E "DEPLOYED|passed|failed" gpurun_out/f_tests.log | tail -n 16
timeout -k 10 600 python -m pytest tests/test_gpu_multiproc.py -q -x --tb=short -p no:cacheprovider > gpurun_out/f_multiproc.log 2>&1; echo "multiproc exit $?"; tail -n 3 gpurun_out/f_multiproc.log
