#!/bin/bash
# Last pass of the round: the full -m gpu suite, smoke, the default bench line of HEAD, the ZeroScope-XL line, the per-kind
# step profile of the five BASELINE geometries, the GroupNorm threshold A/B and the symmetric-failure rehearsal.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.log | tail -n 12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 4 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench exit $?"; cut -c1-700 gpurun_out/bench_n1.json
timeout 600 python bench.py --height 576 --width 1024 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/bench_n1_zeroscope_xl.json 2> gpurun_out/bench_n1_zeroscope_xl.err; echo "bench XL exit $?"; cut -c1-500 gpurun_out/bench_n1_zeroscope_xl.json
bash tools/gpu_gn_ab.sh
for g in "125 32 32 2 modelscope" "24 72 128 2 modelscope" "24 32 32 1 modelscope" "16 32 32 2 lvdm"; do
  timeout 300 python tools/profile_unet.py $g > "gpurun_out/profile_$(echo $g | tr ' ' '_').log" 2>&1; sed -n 4,5p "gpurun_out/profile_$(echo $g | tr ' ' '_').log"
done
export T2V_BENCH_ONE_DEVICE=1 T2V_BENCH_INJECT_FAILURE=all
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 1 --warmup 0 --ddim-steps 2 --frames 9 --no-cpu-baseline > gpurun_out/rehearsal_n4_fallback.json 2> gpurun_out/rehearsal_n4_fallback.err
echo "fallback rehearsal exit $?"; cut -c1-900 gpurun_out/rehearsal_n4_fallback.json; tail -n 2 gpurun_out/rehearsal_n4_fallback.err | cut -c1-300
