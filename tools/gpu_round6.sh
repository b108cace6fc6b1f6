#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench exit $?"; cat gpurun_out/bench_n1.json; tail -n 3 gpurun_out/bench_n1.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1b -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline > $R/gpurun_out/rocprof_bench.log 2>&1; echo "rocprof exit $?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/tools/profile_unet.py 24 32 32 2 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/tools/profile_unet.py 24 32 32 2 > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc write exit $?"
cd $R
find gpurun_out/prof_r1b gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace*" -size +4M -delete
mkdir -p gpurun_out/pmc_all; cp $(find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv") gpurun_out/pmc_all/ 2>/dev/null
python tools/pmc_post.py gpurun_out/pmc_all > gpurun_out/pmc_traffic.txt 2>&1; head -n 20 gpurun_out/pmc_traffic.txt
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv" -size +6M -delete
f=$(find gpurun_out/prof_r1b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 16 "$f"
du -sh gpurun_out
