#!/bin/bash
# Round 3, GPU pass G: the bench lines of HEAD (stdout carries only the JSON line; roofline.traffic from the round-3 PMC file),
# twice for the box-to-box / run-to-run spread, plus the launcher tests.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/g_bench1.json 2> gpurun_out/g_bench1.err; echo "bench exit $?"; wc -l gpurun_out/g_bench1.json; cut -c1-200 gpurun_out/g_bench1.json
timeout -k 10 600 python bench.py --no-cpu-baseline --also-batched 0 > gpurun_out/g_bench2.json 2> gpurun_out/g_bench2.err; echo "bench2 exit $?"; cut -c1-200 gpurun_out/g_bench2.json
timeout -k 10 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/g_bench_lvdm.json 2> gpurun_out/g_bench_lvdm.err; echo "lvdm exit $?"; cut -c1-200 gpurun_out/g_bench_lvdm.json
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_boundary.py tests/test_gpu_rccl.py -q -rs --tb=short -p no:cacheprovider > gpurun_out/g_tests.log 2>&1; echo "tests exit $?"; tail -n 6 gpurun_out/g_tests.log
