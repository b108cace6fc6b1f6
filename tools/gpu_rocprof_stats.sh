#!/bin/bash
# rocprofv3 kernel-trace statistics of the bench command (10 DDIM steps) only.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --also-batched 0 > $R/gpurun_out/rocprof_bench.log 2>&1; echo "rocprof exit $?"
find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/bench_kernel_stats.csv
find $R/gpurun_out/prof_bench -name "*.csv" -size +8M -delete
head -n 8 $R/gpurun_out/bench_kernel_stats.csv | cut -c1-160
