#!/bin/bash
# Round 3, final verification of HEAD: the whole -m gpu suite, smoke, the default bench line, kernel stats of the bench command.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/final_pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/final_pytest_gpu.log | tail -n 8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke exit $?"; tail -n 4 gpurun_out/final_smoke.log
timeout -k 10 900 python bench.py > gpurun_out/final_bench_n1.json 2> gpurun_out/final_bench_n1.err; echo "bench exit $?"; cut -c1-220 gpurun_out/final_bench_n1.json
cd /tmp
rm -rf $R/gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --also-batched 0 > $R/gpurun_out/final_rocprof_bench.log 2>&1; echo "rocprof exit $?"
find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/final_bench_kernel_stats.csv
find $R/gpurun_out/prof_bench -name "*.csv" -size +8M -delete
head -n 8 $R/gpurun_out/final_bench_kernel_stats.csv | cut -c1-150
