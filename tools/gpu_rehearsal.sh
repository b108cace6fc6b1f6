#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a 1-GPU box: every rank on cuda:0 over gloo (T2V_BENCH_ONE_DEVICE=1), two DDIM
# steps, short clips.  Not a measurement — it checks the round-4 flow: rank 0 launches the frame-parallel layout FIRST as its own
# bounded N-rank job (pairs at N = 2, T-shard x CFG pair at N = 4: self-check, >= 1 warm-up + K timed clips, the second clip),
# its line becomes the headline (scaling strong), the outer ranks add `replicas`; and with an injected failure the headline falls
# back to replicas with `config.layout_fallback.reason`.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp T2V_BENCH_ONE_DEVICE=1
run() {  # name nproc args...
  name=$1; n=$2; shift 2
  timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 3 --warmup 1 --ddim-steps 2 --no-cpu-baseline --collective-timeout 600 "$@" > gpurun_out/rehearsal_$name.json 2> gpurun_out/rehearsal_$name.err
  echo "== $name exit $?"; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/rehearsal_$name.json") if l.startswith("{")][-1])
    c = d["config"]
    print({k: d[k] for k in ("metric", "value", "n_gpus", "scaling", "steps", "warmup")}, "layout", c["layout"], "| self_check", c.get("self_check"),
          "| fallback", c.get("layout_fallback"), "| replicas", (d.get("replicas") or {}).get("value"), "| clip_other", (d.get("clip_other") or d.get("clip_125f") or {}).get("value"),
          "| job", d.get("collective_job", {}).get("job_s"))
except Exception as e:
    print("no JSON line:", e)
PY
  grep "\[bench\]" gpurun_out/rehearsal_$name.err | tail -n 3 | cut -c1-400
}
if [ "${T2V_REHEARSAL_ONLY:-}" != "fake_rccl" ]; then      # T2V_REHEARSAL_ONLY=fake_rccl: only the last run (library collectives)
run n4 4 --frames 6 --also-frames 10
run n2 2 --frames 6 --also-frames 10
T2V_BENCH_INJECT_FAILURE=all run n4_fallback 4 --frames 6 --also-frames 10      # the frame-parallel job fails -> replicas headline + reason
fi
# the same N = 4 run with the exchanges INSIDE the library (csrc/comm.hip) over the shared-memory RCCL stand-in of tests/fake_rccl: the
# self-check then really compares the library's collectives with the host executor (config.self_check.in_library == true)
FAKE=tests/fake_rccl/libfakerccl.so
[ -f $FAKE ] || /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared tests/fake_rccl/fake_rccl.cpp -o $FAKE -lrt
T2V_COLLECTIVES=library T2V_RCCL_SONAME=$PWD/$FAKE run n4_fake_rccl 4 --frames 6 --also-frames 0
# N = 2 the same way: the CFG pair's eps exchange and frame gather through t2v_comm_all_gather on the pair's library communicator
T2V_COLLECTIVES=library T2V_RCCL_SONAME=$PWD/$FAKE run n2_fake_rccl 2 --frames 6 --also-frames 0
