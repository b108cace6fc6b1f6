#!/bin/bash
# rocprofv3 kernel-trace statistics of the bench command (10 DDIM steps are enough for per-kernel averages), then the
# HBM counter passes (separate --pmc runs, MI355X_MICROARCH.md §HBM).  Summaries are copied to profiles/ by hand.
#   gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --also-batched 0 > $R/gpurun_out/rocprof_bench.log 2>&1; echo "rocprof exit $?"
tail -n 2 $R/gpurun_out/rocprof_bench.log
find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/bench_kernel_stats.csv
find $R/gpurun_out/prof_bench -name "*.csv" -size +8M -delete
head -n 14 $R/gpurun_out/bench_kernel_stats.csv | cut -c1-150
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python $R/tools/pmc_target.py 2 > $R/gpurun_out/pmc_$C.log 2>&1
  echo "$C exit $?"
done
cd $R
python tools/pmc_post.py gpurun_out gpurun_out/pmc_traffic.json > gpurun_out/pmc_traffic.txt 2>&1; head -n 12 gpurun_out/pmc_traffic.txt
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +20M -delete
