"""TEST INFRASTRUCTURE — one rank of tests/test_gpu_fake_rccl.py::test_peer_window_wait_is_bounded (2 processes on ONE GPU).

Both ranks attach a peer window to a library communicator.  Rank 1 then skips one gather: rank 0's exchange kernel finds no flag, gives up
after T2V_PEER_TIMEOUT_MS and raises the fault word — the host sees T2V_ERR_ASYNC ("peer exchange ...") at its next status call instead
of a hung device.  Both ranks then release their windows (what bench.py's self-check does after an all-reduced verdict) and the same
gather goes through the RCCL entry points (tests/fake_rccl) and equals torch.distributed's."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
from sd_webui_text2video_amd import _lib as L, parallel   # noqa: E402

os.environ["T2V_COLLECTIVES"] = "library"
gc = parallel.GroupComm(dist.group.WORLD, list(range(world)), rank)
n = 4096
mine = torch.arange(n, device=dev, dtype=torch.float32) + 1000 * rank
out = torch.empty(world * n, device=dev)
gc.all_gather_into(out, mine)                       # creates the communicator + window; a healthy exchange
torch.cuda.synchronize()
L.async_status()
assert gc._comm.window, "no peer window attached"
healthy = gc._comm.counters() == (1, 0)
fault_msg, waited = None, 0.0
if rank == 0:
    t0 = time.time()
    gc.all_gather_into(out, mine)                   # rank 1 never sends its part of this one
    torch.cuda.synchronize()
    waited = time.time() - t0
    try:
        L.async_status()
    except L.T2VError as e:
        fault_msg = str(e)
dist.barrier()
gc._comm._lib.t2v_comm_window_open(gc._comm.handle, None)
gc._comm.window = None
out.zero_()
gc.all_gather_into(out, mine)                       # RCCL entry points now
ref = torch.empty_like(out)
parallel.all_gather_into(ref, mine, group=dist.group.WORLD)
torch.cuda.synchronize()
L.async_status()
print("RESULT " + json.dumps({"rank": rank, "healthy_first": bool(healthy), "fault": fault_msg, "waited_s": round(waited, 2),
                              "after_release_equal": bool(torch.equal(out, ref)), "counters": list(gc._comm.counters())}), flush=True)
dist.barrier()
dist.destroy_process_group()
