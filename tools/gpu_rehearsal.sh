#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a 1-GPU box: every rank on cuda:0 over gloo (T2V_BENCH_ONE_DEVICE=1), two DDIM
# steps.  Not a measurement — it checks group set-up, layouts, the sharded forward through the production runner, the
# `replicas` side pass, the JSON line, and the collective fallback.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp T2V_BENCH_ONE_DEVICE=1
run() {  # name nproc args...
  name=$1; n=$2; shift 2
  timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 1 --warmup 0 --ddim-steps 2 "$@" > gpurun_out/rehearsal_$name.json 2> gpurun_out/rehearsal_$name.err
  echo "== $name exit $?"; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/rehearsal_$name.json") if l.startswith("{")][-1])
    print({k: d[k] for k in ("metric", "value", "n_gpus", "scaling")}, d["config"]["layout"], d["config"].get("layout_fallback"), d.get("replicas"), d["roofline"]["whole_video"])
except Exception as e:
    print("no JSON line:", e)
PY
  tail -n 3 gpurun_out/rehearsal_$name.err | cut -c1-300
}
run n4_tshard 4 --frames 9
run n2_pairs 2 --frames 6
T2V_BENCH_INJECT_FAILURE=all run n4_fallback 4 --frames 9      # every rank fails at the same point -> collective switch to replicas
