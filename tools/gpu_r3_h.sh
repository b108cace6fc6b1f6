#!/bin/bash
# Round 3, GPU pass H: GroupNorm barrier generation read at kernel start (one round trip off the chain) — op tests + step profile.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x --tb=short -p no:cacheprovider -k "groupnorm" > gpurun_out/h_ops.log 2>&1; echo "ops exit $?"; tail -n 3 gpurun_out/h_ops.log
timeout 300 python tools/profile_unet.py > gpurun_out/h_prof.log 2>&1; sed -n 4,10p gpurun_out/h_prof.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/h_ops.json
T2V_LIB_PATH=$GRAFT_REPO_ROOT/sd-webui-text2video_amd/libt2v_hip_prev.so timeout 300 python tools/profile_unet.py > gpurun_out/h_prof_prev.log 2>&1; sed -n 4,10p gpurun_out/h_prof_prev.log
timeout 300 python tools/profile_unet.py > gpurun_out/h_prof2.log 2>&1; sed -n 4,10p gpurun_out/h_prof2.log
timeout -k 10 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -x --tb=short -p no:cacheprovider -k "tiny_unet_forward or c1_24f_forward" > gpurun_out/h_tests.log 2>&1; echo "tests exit $?"; tail -n 3 gpurun_out/h_tests.log
