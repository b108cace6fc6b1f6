#!/bin/bash
# Round 3, GPU pass B: device-scope accesses instead of agent fences; locate the hang of pass A's full-size sampling test.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/profile_unet.py > gpurun_out/b_prof_new.log 2>&1; echo "prof new $?"; sed -n 4,14p gpurun_out/b_prof_new.log
T2V_GN_COOP=0 timeout 300 python tools/profile_unet.py > gpurun_out/b_prof_nocoop.log 2>&1; sed -n 4,10p gpurun_out/b_prof_nocoop.log
T2V_GN_COOP=0 T2V_SPLITK_TICKETS=0 timeout 300 python tools/profile_unet.py > gpurun_out/b_prof_old.log 2>&1; sed -n 4,10p gpurun_out/b_prof_old.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/b_ops_old.json
timeout 300 python tools/profile_unet.py > gpurun_out/b_prof_new2.log 2>&1; sed -n 4,5p gpurun_out/b_prof_new2.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/b_ops_new.json
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x --tb=short -p no:cacheprovider -k "groupnorm or split_k or ticket or layout_time" > gpurun_out/b_ops.log 2>&1; echo "ops exit $?"; tail -n 3 gpurun_out/b_ops.log
T2V_GN_COOP=0 timeout -k 10 700 python -m pytest tests/test_gpu_fullsize.py -q -rP --tb=short -p no:cacheprovider -k "c1" > gpurun_out/b_full_c1_nocoop.log 2>&1; echo "fullsize c1 (no coop) exit $?"; grep -E "rel-L2|identical|passed|failed" gpurun_out/b_full_c1_nocoop.log | tail -n 12
timeout -k 10 420 python -X faulthandler -m pytest tests/test_gpu_fullsize.py -q -s --tb=short -p no:cacheprovider -o faulthandler_timeout=240 -k "sampling" > gpurun_out/b_full_sampling_coop.log 2>&1; echo "fullsize sampling (coop) exit $?"; grep -E "rel-L2|identical|passed|failed|File|Thread" gpurun_out/b_full_sampling_coop.log | tail -n 25
timeout -k 10 500 python bench.py > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; echo "bench exit $?"; cut -c1-300 gpurun_out/b_bench.json
