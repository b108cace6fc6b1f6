#!/bin/bash
# same-box A/B of the shared cond | uncond prefix: whole-video bench (the sampler's forward_cfg_pair path) with and without
export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 1 0 1 0; do
  T2V_SHARE_PREFIX=$v timeout 600 python bench.py --no-cpu-baseline --also-batched 0 --steps 3 > gpurun_out/share_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/share_$v.json") if l.startswith("{")][-1]); r=d["roofline"]
print("share=$v", d["value"], "frames/s", d["ms_per_step"], "ms/video", "step(events)", r["unet_step_ms_events"], "flops/step T", r["flops_per_unet_step_T"], r["calibration"]["gemm_8192_tflops_before"])
PY
done
