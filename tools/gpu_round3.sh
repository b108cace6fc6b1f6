#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -n 25 gpurun_out/pytest_gpu.log
timeout 300 python tools/determinism.py > gpurun_out/determinism_tiny.log 2>&1; echo "determinism exit $?"; tail -n 3 gpurun_out/determinism_tiny.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench exit $?"; cat gpurun_out/bench_n1.json; tail -n 5 gpurun_out/bench_n1.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r1 -name "*kernel_trace*" -size +8M -delete
find gpurun_out/prof_r1 -type f | head; 
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 25 "$f"
tail -n 3 gpurun_out/rocprof_bench.log
