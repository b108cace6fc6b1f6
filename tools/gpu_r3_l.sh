#!/bin/bash
# Round 3, GPU pass L: final-build lines of the other BASELINE configurations + rocprofv3 kernel stats of the VideoCrafter line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python bench.py --frames 125 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/l_bench_125f.json 2> gpurun_out/l_bench_125f.err; echo "bench125 exit $?"; cut -c1-160 gpurun_out/l_bench_125f.json
timeout 400 python bench.py --height 576 --width 1024 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/l_bench_xl.json 2> gpurun_out/l_bench_xl.err; echo "bench XL exit $?"; cut -c1-160 gpurun_out/l_bench_xl.json
cd /tmp
rm -rf $R/gpurun_out/prof_lvdm
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lvdm -o lvdm -- python $R/bench.py --model lvdm --steps 1 --warmup 1 --ddim-steps 10 > $R/gpurun_out/l_rocprof_lvdm.log 2>&1; echo "rocprof lvdm exit $?"
find $R/gpurun_out/prof_lvdm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/l_lvdm_kernel_stats.csv
find $R/gpurun_out/prof_lvdm -name "*.csv" -size +8M -delete
head -n 6 $R/gpurun_out/l_lvdm_kernel_stats.csv | cut -c1-140
