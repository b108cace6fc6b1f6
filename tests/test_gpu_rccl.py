"""GPU (-m gpu), >= 2 devices: the library's OWN RCCL communicator under a real multi-rank run (VERDICT r02 #2 / ADVICE r02 #2).

One process per GPU.  Each rank lowers the T-sharded tiny UNet (uneven frame slices), binds it with a t2v_comm created from
a unique id that travels over torch.distributed, and runs the forward as ONE t2v_plan_run: T2V_OP_ALLGATHER (GroupNorm
statistics), T2V_OP_HALO_EXCHANGE (temporal-conv boundary frames) and T2V_OP_ALLTOALL (frame <-> pixel resharding around
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

WORKER = r'''
import os, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rank, world, F = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["T2V_TEST_FRAMES"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", device_id=dev)
from oracle import configs, synth
from sd_webui_text2video_amd import _lib as L, parallel, unet as U
from sd_webui_text2video_amd.program import BoundProgram, COLLECTIVE_KINDS, TShardSpec
from harness import rel_l2
net = U.UNetSD(**configs.TINY_UNET, init_weights=False)
synth.load_synth(net, seed=0)
net = net.to(dev)
g = torch.Generator().manual_seed(21)
x = torch.randn(1, 4, F, 8, 8, generator=g).to(dev)
y = torch.randn(1, 5, 1024, generator=g).to(dev)
t = torch.tensor([613.0], device=dev)
spec = TShardSpec.make(F, world, rank)
ts = parallel.TShard(dist.group.WORLD, list(range(world)), spec)
xl = x[:, :, spec.offset:spec.offset + spec.frames].contiguous()
# (1) production path: collectives inside the library over its own RCCL communicator
net.t_shard = ts
out_lib = net(xl, t, y).clone()
comp = next(c for k, c in net._programs.items() if spec in k)
assert isinstance(comp.bound, BoundProgram) and comp.bound.comm is not None and comp.bound.comm.size == world
kinds = [op.kind for op in comp.prog.ops if op.kind in COLLECTIVE_KINDS]
assert L.OP_ALLGATHER in kinds and L.OP_HALO_EXCHANGE in kinds and (L.OP_ALLTOALL in kinds or 64 % world)
out_lib2 = net(xl, t, y).clone()
torch.cuda.synchronize()
# (2) the same op records through the host executor (torch.distributed collectives on views of the arena)
os.environ["T2V_COLLECTIVES"] = "host"
comp.bound = None
out_host = net(xl, t, y).clone()
assert isinstance(comp.bound, parallel.ShardedExecutor)
torch.cuda.synchronize()
# gather the slices (padded to the largest) and compare with the unsharded forward
pad = torch.zeros(1, 4, spec.max_frames, 8, 8, device=dev, dtype=out_lib.dtype)
pad[:, :, :spec.frames] = out_lib
allp = [torch.empty_like(pad) for _ in range(world)]
dist.all_gather(allp, pad)
res = {"rank": rank, "lib_vs_host_equal": bool(torch.equal(out_lib, out_host)), "rerun_equal": bool(torch.equal(out_lib, out_lib2)),
       "n_collectives": len(kinds)}
if rank == 0:
    net.t_shard = None
    os.environ.pop("T2V_COLLECTIVES")
    whole = net(x, t, y)
    sharded = torch.cat([allp[q][:, :, :spec.counts[q]] for q in range(world)], dim=2)
    res["rel_l2_vs_unsharded"] = rel_l2(sharded.float().cpu(), whole.float().cpu())
print("RESULT " + json.dumps(res), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


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
        procs.append(subprocess.Popen([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
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
    assert results[0]["rel_l2_vs_unsharded"] < 4e-3
