#!/bin/bash
# Round 3, GPU pass D: fused QKV + temporal attention (T2V_EPI_TATTN) — op test, network tests, same-box A/B of the step.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x --tb=short -p no:cacheprovider -k "temporal_attention or groupnorm" > gpurun_out/d_ops.log 2>&1; echo "ops exit $?"; tail -n 6 gpurun_out/d_ops.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x --tb=short -p no:cacheprovider > gpurun_out/d_e2e.log 2>&1; echo "e2e exit $?"; tail -n 4 gpurun_out/d_e2e.log
timeout 300 python tools/profile_unet.py > gpurun_out/d_prof_fused.log 2>&1; echo "prof fused $?"; sed -n 4,12p gpurun_out/d_prof_fused.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/d_ops_fused.json
T2V_FUSED_TATTN=0 timeout 300 python tools/profile_unet.py > gpurun_out/d_prof_unfused.log 2>&1; sed -n 4,12p gpurun_out/d_prof_unfused.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/d_ops_unfused.json
timeout 300 python tools/profile_unet.py > gpurun_out/d_prof_fused2.log 2>&1; sed -n 4,5p gpurun_out/d_prof_fused2.log
timeout -k 10 700 python -m pytest tests/test_gpu_fullsize.py -q -rP --tb=short -p no:cacheprovider -k "c1 or c2" > gpurun_out/d_full.log 2>&1; echo "fullsize exit $?"; grep -E "rel-L2|identical|passed|failed" gpurun_out/d_full.log | tail -n 14
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err; echo "bench exit $?"; cut -c1-200 gpurun_out/d_bench.json
