#!/bin/bash
# One GPU-box pass made of named stages (round 4; replaces the per-session gpu_r3_*.sh scripts):
#   gpurun --timeout 1500 -- 'bash tools/gpu_pass.sh ops fault parity ab smoke rehearsal bench'
# Every stage writes its logs under gpurun_out/<tag>_* (tag = $T2V_PASS_TAG, default "p") and prints a short digest.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export T2V_EXPERIMENTAL=1     # the A/B switches used below are experiment knobs (sd_webui_text2video_amd._lib.knob)
TAG=${T2V_PASS_TAG:-p}
PYT="python -m pytest -q --tb=short -p no:cacheprovider"
digest() { grep -E "passed|failed|FAILED|ERROR|error" "$1" | tail -n "${2:-6}" | cut -c1-300; }

stage_ops() {       # the op-level tests that changed or are new this round
  timeout 900 $PYT tests/test_gpu_ops.py -x -k "torch or lo_output or hi_lo or groupnorm or epilogue or split_k or tattn or temporal_attention or strips or layernorm" > gpurun_out/${TAG}_ops.log 2>&1
  echo "ops exit $?"; digest gpurun_out/${TAG}_ops.log
}
stage_opsall() {
  timeout 1500 $PYT tests/test_gpu_ops.py > gpurun_out/${TAG}_opsall.log 2>&1; echo "ops(all) exit $?"; digest gpurun_out/${TAG}_opsall.log
}
stage_fault() {
  timeout 400 $PYT tests/test_gpu_gn_fault.py -rP > gpurun_out/${TAG}_fault.log 2>&1; echo "gn fault exit $?"; grep -E "FAULT_OK|passed|failed" gpurun_out/${TAG}_fault.log | tail -n 3
}
stage_parity() {    # the full-size goldens that the round-4 operand splits are meant to move
  timeout 1500 $PYT tests/test_gpu_fullsize.py -rP -k "${T2V_PARITY_K:-c1_24f_forward or c3_zeroscope_xl_forward_72x128 or c4 or c1_other or c3_zeroscope_xl_sampled or c2_125f_forward}" > gpurun_out/${TAG}_parity.log 2>&1
  echo "parity exit $?"; grep -E "rel-L2|identical" gpurun_out/${TAG}_parity.log | cut -c1-220; digest gpurun_out/${TAG}_parity.log 3
}
stage_parityall() {
  timeout 2400 $PYT tests/test_gpu_fullsize.py tests/test_gpu_videocrafter.py -rP > gpurun_out/${TAG}_parityall.log 2>&1
  echo "parity(all) exit $?"; grep -E "rel-L2|identical" gpurun_out/${TAG}_parityall.log | cut -c1-220; digest gpurun_out/${TAG}_parityall.log 3
}
prof() {            # name, then env assignments
  name=$1; shift
  env "$@" timeout 300 python tools/profile_unet.py > gpurun_out/${TAG}_prof_$name.log 2>&1
  echo "== profile $name"; sed -n 4,5p gpurun_out/${TAG}_prof_$name.log; grep -E "^(gemm|groupnorm|attention|layernorm|copy2d)" gpurun_out/${TAG}_prof_$name.log | awk '{printf "   %-14s %8s ms %5s\n", $1, $2, $4}'
  cp gpurun_out/unet_ops_b2_f24_32x32.json gpurun_out/${TAG}_ops_$name.json 2>/dev/null
}
stage_ab() {        # same-box A/B of the per-op UNet step: default | round-3 operand splits only | round-4 splits at every level | default again
  prof default T2V_X=0
  prof precise_r3 T2V_PRECISE=r3
  prof precise_all T2V_PRECISE=all
  prof default2 T2V_X=0
}
stage_abgn() {      # same-box A/B: GroupNorm statistics from the producing GEMM's epilogue (default) vs the statistics pass / cooperative kernel
  prof strips T2V_X=0
  prof nostrips T2V_GN_STRIPS=0
  prof strips2 T2V_X=0
  prof nostrips2 T2V_GN_STRIPS=0
}
stage_gnepi() {     # round 5: GroupNorm inside the producing GEMM's epilogue (T2V_EPI_GN) — op tests, the bounded-barrier fault test
  timeout 900 $PYT tests/test_gpu_ops.py -x -k "producer_epilogue or groupnorm or layernorm or cross_attention or splitk" > gpurun_out/${TAG}_gnepi.log 2>&1
  echo "gnepi exit $?"; digest gpurun_out/${TAG}_gnepi.log
}
stage_abgnepi() {   # same-box A/B of the per-op UNet step: fused GroupNorm epilogues (default) vs T2V_GN_EPI=0, twice; then per level
  prof epi T2V_X=0
  prof noepi T2V_GN_EPI=0
  prof epi2 T2V_X=0
  prof noepi2 T2V_GN_EPI=0
  prof epi_only32 T2V_GN_EPI_TILE0=0 T2V_GN_EPI_TILE5=0 T2V_GN_EPI_TILE3=0
  prof nolnx T2V_LN_X=0
  prof barrier T2V_EXCHANGE=barrier
  prof epi3 T2V_X=0
  prof barrier2 T2V_EXCHANGE=barrier
}
stage_ab2() {       # short same-box A/B: fused norms on / off, ModelScope and VideoCrafter steps
  prof epi T2V_X=0
  prof noepi T2V_GN_EPI=0
  prof noxattn T2V_XATTN=0
  for v in "T2V_X=0" "T2V_GN_EPI=0"; do
    env $v timeout 300 python tools/profile_unet.py 16 32 32 2 lvdm > gpurun_out/${TAG}_prof_lvdm_$(echo $v | tr '=' '_').log 2>&1
    echo "== lvdm $v"; sed -n 4,5p gpurun_out/${TAG}_prof_lvdm_$(echo $v | tr '=' '_').log
  done
}
stage_lvdm() {      # configs[4]: bench line + step profile with the 128x320 tile (default) and without
  timeout 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/${TAG}_bench_lvdm.json 2> gpurun_out/${TAG}_bench_lvdm.err; echo "bench lvdm exit $?"; cut -c1-200 gpurun_out/${TAG}_bench_lvdm.json
  T2V_TILE11=0 timeout 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/${TAG}_bench_lvdm_notile11.json 2> gpurun_out/${TAG}_bench_lvdm_notile11.err; echo "bench lvdm (no tile 11) exit $?"; cut -c1-200 gpurun_out/${TAG}_bench_lvdm_notile11.json
  python - <<PY
import json
for n in ("bench_lvdm", "bench_lvdm_notile11"):
    try:
        d = json.loads([l for l in open("gpurun_out/${TAG}_%s.json" % n) if l.startswith("{")][-1]); r = d["roofline"]
        print(n, d["value"], {k: r[k] for k in ("achieved", "frac", "unet_step_ms_events", "unet_step_frac_of_peak")}, r["whole_video"]["frac"])
    except Exception as e:
        print(n, "no line:", e)
PY
}
stage_lvdmp() {     # configs[4] parity only (10 / 50-step x0 against the reference on the deployed weights)
  timeout 900 $PYT tests/test_gpu_fullsize.py -rP -k "c4" > gpurun_out/${TAG}_c4.log 2>&1; echo "c4 exit $?"
  grep -E "rel-L2|passed|failed" gpurun_out/${TAG}_c4.log | cut -c1-200
}
stage_lvdmx() {     # configs[4] parity + bench under the heavier precision settings (every level split, fp32 GroupNorm-only tensors)
  for cfg in "T2V_PRECISE=all" "T2V_PRECISE=all T2V_NORM_INPUT=f32" "T2V_NORM_INPUT=f32"; do
    tag=$(echo $cfg | tr ' =' '__')
    env $cfg timeout 900 $PYT tests/test_gpu_fullsize.py -rP -k "c4" > gpurun_out/${TAG}_c4_$tag.log 2>&1
    echo "== $cfg"; grep -E "DEPLOYED" gpurun_out/${TAG}_c4_$tag.log | cut -c1-200
    env $cfg timeout 400 python bench.py --model lvdm --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   bench', d['value'], d['roofline']['unet_step_ms_events'])"
  done
}
stage_sweep12() {   # does the 128x320 tile win anywhere at the 16x16 / 8x8 levels (b = 2)?
  timeout 600 python tools/gemm_sweep.py L1 > gpurun_out/${TAG}_sweep_L1.txt 2>&1; cut -c1-400 gpurun_out/${TAG}_sweep_L1.txt
  timeout 600 python tools/gemm_sweep.py L2 > gpurun_out/${TAG}_sweep_L2.txt 2>&1; cut -c1-400 gpurun_out/${TAG}_sweep_L2.txt
}
stage_sweep3() {    # the 4x4 level (M = 768): 64x64 tiles with the full reduction (tile 12) against the split-K configurations
  timeout 600 python tools/gemm_sweep.py L3 > gpurun_out/${TAG}_sweep_L3.txt 2>&1; cut -c1-700 gpurun_out/${TAG}_sweep_L3.txt
}
stage_sweep() {     # tile sweeps of the shapes the 128x320 tile is meant for: VideoCrafter (16 frames, b = 2) and one CFG role per GPU (b = 1)
  SWEEP_FRAMES=16 timeout 500 python tools/gemm_sweep.py L0 > gpurun_out/${TAG}_sweep_L0_f16.txt 2>&1; cat gpurun_out/${TAG}_sweep_L0_f16.txt | cut -c1-330
  SWEEP_BATCH=1 timeout 500 python tools/gemm_sweep.py L0 > gpurun_out/${TAG}_sweep_L0_b1.txt 2>&1; cat gpurun_out/${TAG}_sweep_L0_b1.txt | cut -c1-330
}
stage_smoke() {
  timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?"; grep smoke gpurun_out/${TAG}_smoke.log | tail -n 5
}
stage_rehearsal() { bash tools/gpu_rehearsal.sh; }
stage_bench() {
  timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; echo "bench exit $?"; cut -c1-330 gpurun_out/${TAG}_bench_n1.json
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench_n1.json") if l.startswith("{")][-1]); r = d["roofline"]
    print({k: r[k] for k in ("achieved", "frac", "unet_step_ms_events", "unet_step_frac_of_peak")}, r["whole_video"], r["calibration"].get("gemm_8192_tflops_before"), d.get("batched", {}).get("value"), d["cpu_baseline"]["value"])
except Exception as e:
    print("bench line:", e)
PY
}
stage_bench20() {   # the driver's K: 20 timed videos after 2 warm-ups (sustained clocks), no CPU leg
  timeout -k 10 900 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --also-batched 0 > gpurun_out/${TAG}_bench_n1_steps20.json 2> gpurun_out/${TAG}_bench_n1_steps20.err; echo "bench20 exit $?"; cut -c1-330 gpurun_out/${TAG}_bench_n1_steps20.json
}
stage_multiproc() {
  timeout 1200 $PYT tests/test_gpu_multiproc.py tests/test_gpu_fake_rccl.py tests/test_gpu_boundary.py -rP > gpurun_out/${TAG}_multiproc.log 2>&1; echo "multiproc exit $?"; grep -E "identical|passed|failed" gpurun_out/${TAG}_multiproc.log | tail -n 8 | cut -c1-250
}
stage_coll() {      # round 5: one exchange per temporal convolution (T2V_OP_STATS_HALO), eps / frame gathers through t2v_comm_all_gather
  timeout 1200 $PYT tests/test_gpu_fake_rccl.py tests/test_gpu_boundary.py tests/test_gpu_multiproc.py tests/test_gpu_e2e.py -rP \
    -k "library_collectives or single_rank_communicator or runner_layouts or tsharded" > gpurun_out/${TAG}_coll.log 2>&1; echo "coll exit $?"
  grep -E "library collectives|identical|passed|failed|Error" gpurun_out/${TAG}_coll.log | tail -n 10 | cut -c1-600
}
stage_tshardrank() {   # compute time of ONE rank's real T-sharded program (collectives left out), fused per-frame norms on / off
  for g in "24 4 1" "24 2 0" "125 4 1" "125 2 0"; do
    for v in "T2V_X=0" "T2V_GN_EPI=0"; do
      env $v timeout 300 python tools/profile_tshard_rank.py $g > "gpurun_out/${TAG}_tshard_rank_$(echo $g | tr ' ' '_')_$(echo $v | tr '=' '_').log" 2>&1
      echo "== $g $v"; grep -E "T-shard rank|Error|error" "gpurun_out/${TAG}_tshard_rank_$(echo $g | tr ' ' '_')_$(echo $v | tr '=' '_').log" | cut -c1-400
    done
  done
  timeout 600 $PYT tests/test_gpu_e2e.py -k "tsharded" > gpurun_out/${TAG}_tshard_e2e.log 2>&1; echo "tsharded e2e exit $?"; digest gpurun_out/${TAG}_tshard_e2e.log 3
}
stage_abnt() {      # round 5: non-temporal hints on the fp32 residual stream (reads / writes / both) — variant libraries of tools/build_variant.py
  prof base T2V_X=0
  for v in ntboth ntres ntout; do
    [ -f tools/variants/libt2v_hip_$v.so ] && prof $v T2V_LIB_PATH=$PWD/tools/variants/libt2v_hip_$v.so
  done
  prof base2 T2V_X=0
}
stage_vaepar() {    # VAE parity after the mid-attention blocking change + the new default's profile + the two other end-to-end bench lines
  timeout 900 $PYT tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -rP -k "vae" > gpurun_out/${TAG}_vaepar.log 2>&1; echo "vae parity exit $?"
  grep -E "rel-L2|passed|failed" gpurun_out/${TAG}_vaepar.log | cut -c1-220 | tail -n 12
  timeout 300 python tools/profile_vae.py 1 72 128 > gpurun_out/${TAG}_profile_vae_xl.txt 2>&1; grep -E "^VAE decode|mid attention|qk\^T|softmax\.|\.pv\." gpurun_out/${TAG}_profile_vae_xl.txt | head -6 | cut -c1-260
  timeout 600 python bench.py --frames 125 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/${TAG}_bench_n1_125f.json 2> gpurun_out/${TAG}_bench_n1_125f.err; echo "bench125 exit $?"; cut -c1-200 gpurun_out/${TAG}_bench_n1_125f.json
  timeout 600 python bench.py --height 576 --width 1024 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/${TAG}_bench_n1_zeroscope_xl.json 2> gpurun_out/${TAG}_bench_n1_zeroscope_xl.err; echo "bench XL exit $?"; cut -c1-200 gpurun_out/${TAG}_bench_n1_zeroscope_xl.json
}
stage_vae24() {     # the headline's own VAE stage: 24 frames at 256x256 in one decode
  timeout 300 python tools/profile_vae.py 24 32 32 > gpurun_out/${TAG}_vae_24f_256.txt 2>&1; head -n 40 gpurun_out/${TAG}_vae_24f_256.txt | cut -c1-200
}
stage_vaeattn() {   # ZeroScope-XL VAE mid-attention (9216 tokens, d = 512): query rows per block x split-K of the P V GEMM
  for cfg in "T2V_X=0" "T2V_VAE_PV_SPLITK=1" "T2V_VAE_BQ=2304" "T2V_VAE_BQ=2304 T2V_VAE_PV_SPLITK=1" "T2V_VAE_BQ=4608" "T2V_VAE_BQ=4608 T2V_VAE_PV_SPLITK=1" "T2V_VAE_BQ=9216"; do
    tag=$(echo $cfg | tr ' =' '__')
    env $cfg timeout 300 python tools/profile_vae.py 1 72 128 > gpurun_out/${TAG}_vae_$tag.txt 2>&1
    echo "== $cfg"; grep -E "^VAE decode|mid attention|qk\^T|softmax\.|\.pv\." gpurun_out/${TAG}_vae_$tag.txt | head -6 | cut -c1-260
  done
}
stage_e2e() {
  timeout 1500 $PYT tests/test_gpu_e2e.py tests/test_gpu_text_encoder.py tests/test_gpu_videocrafter.py > gpurun_out/${TAG}_e2e.log 2>&1; echo "e2e exit $?"; digest gpurun_out/${TAG}_e2e.log
}
stage_suite() {     # what the driver runs at round end
  timeout -k 10 2400 python -m pytest tests -m gpu -q -rP --tb=short --durations=12 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu exit $?"; digest gpurun_out/${TAG}_pytest_gpu.log 12
}
stage_roundend() {  # the measurements that go to profiles/r0N_*: rocprofv3 kernel stats + PMC traffic + MFMA utilisation on this build, the other
                    # BASELINE geometries, per-kind step profiles, the stages either side of the loop
  R=$GRAFT_REPO_ROOT
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
  for g in "24 32 32 2 modelscope" "125 32 32 2 modelscope" "24 72 128 2 modelscope" "32 32 32 1 modelscope" "24 32 32 1 modelscope" "12 32 32 1 modelscope" "6 32 32 1 modelscope" "16 32 32 2 lvdm"; do
    timeout 300 python tools/profile_unet.py $g > "gpurun_out/profile_$(echo $g | tr ' ' '_').log" 2>&1; sed -n 4,5p "gpurun_out/profile_$(echo $g | tr ' ' '_').log"
  done
  timeout 300 python tools/profile_aux.py > gpurun_out/aux_stages.txt 2>&1; tail -n 8 gpurun_out/aux_stages.txt
  timeout 300 python tools/profile_vae.py 1 72 128 > gpurun_out/profile_vae_xl.txt 2>&1; tail -n 6 gpurun_out/profile_vae_xl.txt
}
stage_r5a() {       # round 5, re-entry: the evidence the DESIGN tables cite, without the rocprofv3 / PMC passes (those are in profiles/ already)
  stage_smoke
  stage_bench
  prof epi T2V_X=0
  prof noepi T2V_GN_EPI=0
  for v in "T2V_RELPOS_MFMA=2" "T2V_RELPOS_MFMA=0"; do
    env $v timeout 300 python tools/profile_unet.py 16 32 32 2 lvdm > gpurun_out/${TAG}_prof_lvdm_$(echo $v | tr '=' '_').log 2>&1
    echo "== lvdm $v"; sed -n 4,5p gpurun_out/${TAG}_prof_lvdm_$(echo $v | tr '=' '_').log
  done
  timeout 400 python bench.py --model lvdm --steps 2 --warmup 1 > gpurun_out/${TAG}_bench_n1_lvdm.json 2> gpurun_out/${TAG}_bench_n1_lvdm.err; echo "bench lvdm exit $?"; cut -c1-200 gpurun_out/${TAG}_bench_n1_lvdm.json
  for g in "125 32 32 2 modelscope" "24 72 128 2 modelscope" "32 32 32 1 modelscope" "24 32 32 1 modelscope" "12 32 32 1 modelscope" "6 32 32 1 modelscope"; do
    timeout 300 python tools/profile_unet.py $g > "gpurun_out/${TAG}_profile_$(echo $g | tr ' ' '_').log" 2>&1; sed -n 4,5p "gpurun_out/${TAG}_profile_$(echo $g | tr ' ' '_').log"
  done
  timeout 300 python tools/profile_vae.py 1 72 128 > gpurun_out/${TAG}_profile_vae_xl.txt 2>&1; tail -n 6 gpurun_out/${TAG}_profile_vae_xl.txt
}
for st in "$@"; do
  t0=$(date +%s); echo "######## stage $st"; stage_$st; echo "######## $st took $(( $(date +%s) - t0 )) s"
done
