"""GPU (-m gpu), >= 2 devices: the library's OWN RCCL communicator under a real multi-rank run (VERDICT r02 #2 / ADVICE r02 #2).

One process per GPU.  Each rank lowers the T-sharded tiny UNet (uneven frame slices), binds it with a t2v_comm created from
a unique id that travels over torch.distributed, and runs the forward as ONE t2v_plan_run: T2V_OP_ALLGATHER (GroupNorm
statistics), T2V_OP_STATS_HALO (statistics + raw temporal-conv boundary frames in one group) and T2V_OP_ALLTOALL (frame <-> pixel resharding around
the TemporalTransformers) all execute through csrc/comm.hip on the launch stream.  The same records are then executed by
parallel.ShardedExecutor (collectives through torch.distributed — the path the gloo CPU tests pin) and the two results must
be bit-identical; the concatenated frames are checked against the unsharded forward of rank 0.

Skips only when the box has fewer than 2 GPUs (RCCL refuses two ranks on one device).  Every rank runs under a hard
timeout, so a hang in a collective fails the test instead of the suite."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker.py")      # shared with test_gpu_fake_rccl.py


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,frames", [(2, 5), (4, 10)])        # 5 = 3 + 2, 10 = 3 + 3 + 3 + 1
def test_library_rccl_collectives_multi_rank(world, frames):
    import json
    n_dev = torch.cuda.device_count()
    if n_dev < world:
        pytest.skip(f"needs {world} GPUs, this box has {n_dev} (RCCL refuses several ranks on one device)")
    port = _free_port()
    procs = []
    for r in range(world):
        env = {**os.environ, "RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1",
               "MASTER_PORT": str(port), "T2V_TEST_FRAMES": str(frames), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
        env.pop("T2V_COLLECTIVES", None)
        env.update(T2V_TEST_BACKEND="nccl", T2V_TEST_ONE_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=420)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    results = []
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        line = next(ln for ln in out.splitlines() if ln.startswith("RESULT "))
        results.append(json.loads(line[7:]))
    print(f"RCCL x{world}: {results}")
    for res in results:
        assert res["lib_vs_host_equal"] and res["rerun_equal"] and res["n_collectives"] > 100
        # round 5: one grouped statistics + boundary-frame exchange per temporal convolution, bit-equal to the two-exchange lowering;
        # the gathers around the forward through t2v_comm_all_gather on a second communicator
        assert res["stats_halo"] == 88 and res["two_exchange_form_equal"] and res["group_comm_gather_equal"]
    assert results[0]["rel_l2_vs_unsharded"] < 4e-3
