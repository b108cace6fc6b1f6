#!/bin/bash
# Round-end check on the GPU box: full parity suite, smoke(), the default bench line.
#   gpurun --timeout 1800 -- 'bash tools/gpu_check.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -n 8 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 3 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench exit $?"; cat gpurun_out/bench_n1.json; tail -n 3 gpurun_out/bench_n1.err
timeout 300 python tools/profile_unet.py > gpurun_out/profile_unet.log 2>&1; echo "profile exit $?"; grep -E "geometry|algorithmic" gpurun_out/profile_unet.log
