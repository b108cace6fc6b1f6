"""TEST INFRASTRUCTURE — one rank of the library-communicator tests (tests/test_gpu_rccl.py: real RCCL, one GPU per rank;
tests/test_gpu_fake_rccl.py: the shared-memory stand-in of tests/fake_rccl, all ranks on ONE GPU).  Run as a script with
RANK / WORLD_SIZE / MASTER_* in the environment; T2V_TEST_BACKEND = nccl | gloo, T2V_TEST_ONE_DEVICE = 0 | 1, T2V_TEST_FRAMES.

Lowers the T-sharded tiny UNet (uneven frame slices), runs the forward with the exchanges INSIDE the library (t2v_comm: statistics
all-gathers, halo send / recv, frame <-> pixel all-to-alls as program ops on the launch stream), then the same op records through
parallel.ShardedExecutor (collectives via torch.distributed, the path the gloo CPU tests pin); prints one RESULT line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
rank, world, F = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["T2V_TEST_FRAMES"])
backend = os.environ.get("T2V_TEST_BACKEND", "nccl")
one_device = os.environ.get("T2V_TEST_ONE_DEVICE", "0") == "1"
dev = torch.device("cuda", 0 if one_device else rank)
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group("gloo")
from harness import rel_l2                                             # noqa: E402
from oracle import configs, synth                                      # noqa: E402
from sd_webui_text2video_amd import _lib as L, parallel, unet as U    # noqa: E402
from sd_webui_text2video_amd.program import BoundProgram, COLLECTIVE_KINDS, TShardSpec   # noqa: E402

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
# (1) production path: collectives inside the library over its own communicator
os.environ["T2V_COLLECTIVES"] = "library"
net.t_shard = ts
out_lib = net(xl, t, y).clone()
comp = next(c for k, c in net._programs.items() if spec in k)
assert isinstance(comp.bound, BoundProgram) and comp.bound.comm is not None and comp.bound.comm.size == world
kinds = [op.kind for op in comp.prog.ops if op.kind in COLLECTIVE_KINDS]
assert L.OP_ALLGATHER in kinds and L.OP_STATS_HALO in kinds and (L.OP_ALLTOALL in kinds or 64 % world)
out_lib2 = net(xl, t, y).clone()
torch.cuda.synchronize()
via_window, via_rccl = comp.bound.comm.counters()          # two forwards: how many of their exchanges were kernels of the library (peer window)
window_desc = comp.bound.comm.window
# (2) the same op records through the host executor (torch.distributed collectives on views of the arena)
os.environ["T2V_COLLECTIVES"] = "host"
comp.bound = None
out_host = net(xl, t, y).clone()
assert isinstance(comp.bound, parallel.ShardedExecutor)
torch.cuda.synchronize()
# (3) the two-exchange lowering of rounds 1-4 (statistics all-gather, normalise, halo exchange of the NORMALISED frames), in the library:
# a neighbour that normalises my raw boundary frame with the gathered statistics must produce the bits I produced for it
os.environ["T2V_COLLECTIVES"], os.environ["T2V_STATS_HALO"] = "library", "0"
net._programs.clear()
out_two = net(xl, t, y).clone()
comp2 = next(c for k, c in net._programs.items() if spec in k)
kinds2 = [op.kind for op in comp2.prog.ops if op.kind in COLLECTIVE_KINDS]
assert L.OP_STATS_HALO not in kinds2 and kinds2.count(L.OP_HALO_EXCHANGE) == kinds.count(L.OP_STATS_HALO)
torch.cuda.synchronize()
os.environ.pop("T2V_STATS_HALO")
# (4) the gathers AROUND the forward (eps of a CFG pair per step, uint8 frames per video) through t2v_comm_all_gather on a second
# library communicator, against torch.distributed
os.environ["T2V_COLLECTIVES"] = "library"
gc = parallel.GroupComm(dist.group.WORLD, list(range(world)), rank)
mine = (torch.arange(3001, device=dev, dtype=torch.float32) * (rank + 1)).half()
via_lib, via_torch = torch.empty(world * 3001, device=dev, dtype=torch.float16), torch.empty(world * 3001, device=dev, dtype=torch.float16)
gc.all_gather_into(via_lib, mine)
assert gc._comm is not None and gc._comm.size == world
parallel.all_gather_into(via_torch, mine, group=dist.group.WORLD)
torch.cuda.synchronize()
gathers_equal = bool(torch.equal(via_lib, via_torch))
# (the 6002-byte parts above are not made of 16-byte units: RCCL carries them even with a peer window; an eps-sized part goes over the window)
mine2 = torch.randn(4 * F * 64, device=dev, generator=torch.Generator(device=dev).manual_seed(rank)).float()
via_lib2, via_torch2 = torch.empty(world * mine2.numel(), device=dev), torch.empty(world * mine2.numel(), device=dev)
gc.all_gather_into(via_lib2, mine2)
parallel.all_gather_into(via_torch2, mine2, group=dist.group.WORLD)
torch.cuda.synchronize()
gathers_equal = gathers_equal and bool(torch.equal(via_lib2, via_torch2))
pad = torch.zeros(1, 4, spec.max_frames, 8, 8, device=dev, dtype=out_lib.dtype)
pad[:, :, :spec.frames] = out_lib
if backend == "nccl":
    allp = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allp, pad)
else:
    host = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(world)]
    dist.all_gather(host, pad.cpu())
    allp = [h.to(dev) for h in host]
res = {"rank": rank, "lib_vs_host_equal": bool(torch.equal(out_lib, out_host)), "rerun_equal": bool(torch.equal(out_lib, out_lib2)),
       "n_collectives": len(kinds), "alltoall": kinds.count(L.OP_ALLTOALL), "halo": kinds.count(L.OP_HALO_EXCHANGE),
       "stats_halo": kinds.count(L.OP_STATS_HALO), "allgather": kinds.count(L.OP_ALLGATHER),
       "two_exchange_form_equal": bool(torch.equal(out_lib, out_two)), "group_comm_gather_equal": gathers_equal, "n_collectives_two_exchange_form": len(kinds2),
       "via_window": via_window, "via_rccl": via_rccl, "window": window_desc, "gather_via_window": gc._comm.counters()[0]}
if rank == 0:
    net.t_shard = None
    os.environ.pop("T2V_COLLECTIVES")
    whole = net(x, t, y)
    sharded = torch.cat([allp[q][:, :, :spec.counts[q]] for q in range(world)], dim=2)
    res["rel_l2_vs_unsharded"] = rel_l2(sharded.float().cpu(), whole.float().cpu())
print("RESULT " + json.dumps(res), flush=True)
dist.barrier()
dist.destroy_process_group()
