#!/bin/bash
# Round 3, GPU pass A: new kernels (cooperative GroupNorm, split-K ticket fold, hi+lo casts) — op tests first under short
# timeouts, then same-box A/B of the UNet step, the configs[1] parity against both goldens, the default bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_ops.py -q -x --tb=short -p no:cacheprovider -k "groupnorm" > gpurun_out/a_ops_gn.log 2>&1; echo "ops gn exit $?"; tail -n 4 gpurun_out/a_ops_gn.log
timeout 240 python -m pytest tests/test_gpu_ops.py -q -x --tb=short -p no:cacheprovider -k "split_k or ticket" > gpurun_out/a_ops_sk.log 2>&1; echo "ops splitk exit $?"; tail -n 4 gpurun_out/a_ops_sk.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -k "not groupnorm and not split_k and not ticket" > gpurun_out/a_ops_rest.log 2>&1; echo "ops rest exit $?"; tail -n 4 gpurun_out/a_ops_rest.log
timeout 300 python tools/profile_unet.py > gpurun_out/a_prof_new.log 2>&1; echo "prof new $?"; sed -n 4,17p gpurun_out/a_prof_new.log
T2V_GN_COOP=0 timeout 300 python tools/profile_unet.py > gpurun_out/a_prof_nocoop.log 2>&1; sed -n 4,9p gpurun_out/a_prof_nocoop.log
T2V_GN_COOP=0 T2V_SPLITK_TICKETS=0 timeout 300 python tools/profile_unet.py > gpurun_out/a_prof_old.log 2>&1; sed -n 4,9p gpurun_out/a_prof_old.log
timeout 300 python tools/profile_unet.py > gpurun_out/a_prof_new2.log 2>&1; sed -n 4,9p gpurun_out/a_prof_new2.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -rP --tb=short -p no:cacheprovider -k "c1" > gpurun_out/a_full_c1.log 2>&1; echo "fullsize c1 exit $?"; grep -E "rel-L2|identical|passed|failed" gpurun_out/a_full_c1.log | tail -n 12
timeout 600 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench exit $?"; cut -c1-400 gpurun_out/a_bench.json; tail -n 3 gpurun_out/a_bench.err
