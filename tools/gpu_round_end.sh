#!/bin/bash
# Round-end pass on HEAD (round 3): full -m gpu suite, smoke, the default bench line, rocprofv3 kernel stats + PMC traffic + MFMA
# utilisation on the SAME build, the 125-frame / ZeroScope-XL / VideoCrafter lines, per-kind step profiles of the BASELINE
# geometries, the stages either side of the loop, and the self-launched N > 1 rehearsals on this box's one GPU.
#   gpurun --timeout 3000 -- 'bash tools/gpu_round_end.sh'      then copy gpurun_out/* summaries to profiles/r03_*
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests -m gpu -q -rP --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.log | tail -n 12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 4 gpurun_out/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench exit $?"; cut -c1-260 gpurun_out/bench_n1.json
timeout 300 python tools/profile_unet.py > gpurun_out/profile_unet.log 2>&1; sed -n 4,17p gpurun_out/profile_unet.log
bash tools/gpu_profile.sh > gpurun_out/gpu_profile.out 2>&1; tail -n 14 gpurun_out/gpu_profile.out | cut -c1-200
cd /tmp
rm -rf $R/gpurun_out/pmc_mfma
timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python $R/tools/pmc_target.py 2 > $R/gpurun_out/pmc_mfma.log 2>&1; echo "mfma pmc exit $?"
cd $R
python tools/pmc_generic_post.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma_counters.txt 2>&1
python tools/mfma_util_post.py gpurun_out/pmc_mfma_counters.txt > gpurun_out/mfma_utilisation_unet.txt 2>&1; tail -n 3 gpurun_out/mfma_utilisation_unet.txt
find gpurun_out/pmc_mfma -name "*.csv" -size +20M -delete
timeout 600 python bench.py --frames 125 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/bench_n1_125f.json 2> gpurun_out/bench_n1_125f.err; echo "bench125 exit $?"; cut -c1-200 gpurun_out/bench_n1_125f.json
timeout 600 python bench.py --height 576 --width 1024 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/bench_n1_zeroscope_xl.json 2> gpurun_out/bench_n1_zeroscope_xl.err; echo "bench XL exit $?"; cut -c1-200 gpurun_out/bench_n1_zeroscope_xl.json
timeout 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/bench_n1_lvdm.json 2> gpurun_out/bench_n1_lvdm.err; echo "bench lvdm exit $?"; cut -c1-200 gpurun_out/bench_n1_lvdm.json
for g in "125 32 32 2 modelscope" "24 72 128 2 modelscope" "24 32 32 1 modelscope" "16 32 32 2 lvdm"; do
  timeout 300 python tools/profile_unet.py $g > "gpurun_out/profile_$(echo $g | tr ' ' '_').log" 2>&1; sed -n 4,5p "gpurun_out/profile_$(echo $g | tr ' ' '_').log"
done
timeout 300 python tools/profile_aux.py > gpurun_out/aux_stages.txt 2>&1; tail -n 8 gpurun_out/aux_stages.txt
# self-launched N > 1 rehearsals (all ranks on this one GPU over gloo: code-path checks, never measurements)
export T2V_BENCH_ONE_DEVICE=1
timeout -k 10 900 python bench.py --gpus 4 --steps 1 --warmup 0 --ddim-steps 2 --frames 6 --no-cpu-baseline --collective-timeout 500 > gpurun_out/rehearsal_n4.json 2> gpurun_out/rehearsal_n4.err; echo "rehearsal n4 exit $?"; cut -c1-300 gpurun_out/rehearsal_n4.json
timeout -k 10 600 python bench.py --gpus 2 --steps 1 --warmup 0 --ddim-steps 2 --frames 6 --no-cpu-baseline --collective-timeout 400 > gpurun_out/rehearsal_n2.json 2> gpurun_out/rehearsal_n2.err; echo "rehearsal n2 exit $?"; cut -c1-300 gpurun_out/rehearsal_n2.json
unset T2V_BENCH_ONE_DEVICE
