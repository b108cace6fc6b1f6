#!/bin/bash
# Round 3, GPU pass C: cooperative GroupNorm with the two-level barrier (A/B against the three launches), the whole -m gpu
# suite, smoke, the default / VideoCrafter / 125-frame bench lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/profile_unet.py > gpurun_out/c_prof_coop.log 2>&1; echo "prof coop $?"; sed -n 4,12p gpurun_out/c_prof_coop.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/c_ops_coop.json
T2V_GN_COOP=0 timeout 300 python tools/profile_unet.py > gpurun_out/c_prof_nocoop.log 2>&1; sed -n 4,12p gpurun_out/c_prof_nocoop.log
cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/c_ops_nocoop.json
timeout 300 python tools/profile_unet.py > gpurun_out/c_prof_coop2.log 2>&1; sed -n 4,5p gpurun_out/c_prof_coop2.log
timeout -k 10 1500 python -m pytest tests -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/c_pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/c_pytest_gpu.log | tail -n 12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c_smoke.log 2>&1; echo "smoke exit $?"; tail -n 4 gpurun_out/c_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err; echo "bench exit $?"; cut -c1-200 gpurun_out/c_bench.json
timeout -k 10 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/c_bench_lvdm.json 2> gpurun_out/c_bench_lvdm.err; echo "bench lvdm exit $?"; cut -c1-300 gpurun_out/c_bench_lvdm.json; tail -n 2 gpurun_out/c_bench_lvdm.err
