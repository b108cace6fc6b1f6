#!/bin/bash
# Round-2 measurement pass: changed tests, smoke, the default bench line, rocprofv3 kernel stats + PMC traffic of the same
# workload, MFMA utilisation over a forward, the 125-frame clip on one GPU.
#   gpurun --timeout 2400 -- 'bash tools/gpu_final.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest ${T2V_TESTS:-tests/test_gpu_ops.py tests/test_gpu_multiproc.py tests/test_gpu_boundary.py tests/test_gpu_e2e.py} -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/pytest_final.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_final.log | tail -n 12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 4 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench exit $?"; cat gpurun_out/bench_n1.json; tail -n 3 gpurun_out/bench_n1.err
bash tools/gpu_profile.sh
cd /tmp
rm -rf $R/gpurun_out/pmc_mfma
timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python $R/tools/pmc_target.py 2 > $R/gpurun_out/pmc_mfma.log 2>&1; echo "mfma pmc exit $?"
cd $R
python tools/pmc_generic_post.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma_counters.txt 2>&1
python tools/mfma_util_post.py gpurun_out/pmc_mfma_counters.txt > gpurun_out/mfma_utilisation_unet.txt 2>&1; tail -n 14 gpurun_out/mfma_utilisation_unet.txt
find gpurun_out/pmc_mfma -name "*.csv" -size +20M -delete
timeout 600 python bench.py --frames 125 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/bench_n1_125f.json 2> gpurun_out/bench_n1_125f.err; echo "bench125 exit $?"; cat gpurun_out/bench_n1_125f.json | cut -c1-400
timeout 300 python tools/profile_unet.py > gpurun_out/profile_unet.log 2>&1; sed -n 4,17p gpurun_out/profile_unet.log
