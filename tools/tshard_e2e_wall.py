"""End-to-end wall time of a T-sharded UNet forward with its exchanges, R ranks as R processes on ONE GPU (the pool has 1-GPU boxes).

The R processes share the device, so the wall time of one forward is ~ the SUM of the ranks' compute plus what the exchange path adds
(launches, flag waits, the copies through the windows); tools/profile_tshard_rank.py gives one rank's compute alone.  The difference
per forward / per exchange is the software overhead of the exchange path — a number that does not need a second GPU (xGMI latency and
bandwidth do).  Transports: peer windows (device-initiated stores into IPC-mapped mailboxes, csrc/comm.hip) and the RCCL entry points
of tests/fake_rccl (a host-side shared-memory stand-in: its cost says nothing about RCCL's).

  python tools/tshard_e2e_wall.py <total frames> <R>            launcher: runs both transports, prints one line each
Fused norms are off in every process (several processes on one device cannot rely on co-resident grids)."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    rank, world, total = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["E2E_FRAMES"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    from sd_webui_text2video_amd import configs, parallel, unet as U
    from sd_webui_text2video_amd.program import COLLECTIVE_KINDS, TShardSpec
    from profile_unet import random_weights_
    net = U.UNetSD(**configs.MODELSCOPE_UNET, init_weights=False).half().to(dev)
    random_weights_(net)
    spec = TShardSpec.make(total, world, rank)
    ts = parallel.TShard(dist.group.WORLD, list(range(world)), spec)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, total, 32, 32, generator=g)[:, :, spec.offset:spec.offset + spec.frames].contiguous().to(dev)
    y = torch.randn(1, 77, net.context_dim, generator=g).half().to(dev)
    t = torch.tensor([500.0], device=dev)
    os.environ["T2V_COLLECTIVES"] = "library"
    net.t_shard = ts
    net.auto_refresh = False
    for _ in range(2):
        out = net(x, t, y)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    comp = next(c for k, c in net._programs.items() if spec in k)
    n_coll = sum(1 for op in comp.prog.ops if op.kind in COLLECTIVE_KINDS)
    n_ops = len(comp.prog.ops) - n_coll
    res = []
    for _ in range(3):
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            net(x, t, y)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 5 * 1e3)
    via = comp.bound.comm.counters()
    print("RESULT " + json.dumps({"rank": rank, "frames": spec.frames, "ms_per_forward": round(sorted(res)[1], 3), "compute_ops": n_ops,
                                  "collective_ops": n_coll, "via_window": via[0], "via_rccl_entry_points": via[1]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def launch(total, R, window):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
    if not os.path.exists(fake):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-o", fake, "-lrt"], check=True)
    procs = []
    for r in range(R):
        env = {**os.environ, "RANK": str(r), "WORLD_SIZE": str(R), "LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
               "E2E_FRAMES": str(total), "E2E_WORKER": "1", "T2V_RCCL_SONAME": fake, "T2V_GN_COOP": "0", "T2V_GN_EPI": "0", "T2V_PEER_WINDOW": str(window)}
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    rs = []
    for p, o in zip(procs, outs):
        if p.returncode != 0:
            print(o[-2000:])
            raise SystemExit(1)
        rs.append(json.loads(next(ln for ln in o.splitlines() if ln.startswith("RESULT "))[7:]))
    ms = max(r["ms_per_forward"] for r in rs)
    print(f"{total} frames over {R} ranks (processes on ONE GPU), {'peer windows' if window else 'RCCL entry points of tests/fake_rccl (host stand-in)'}: "
          f"{ms:.2f} ms per sharded forward of all {R} ranks together; per rank {rs[0]['compute_ops']} compute ops + {rs[0]['collective_ops']} exchanges "
          f"({rs[0]['via_window']} kernels over the window, {rs[0]['via_rccl_entry_points']} through the RCCL entry points, warm-ups included)", flush=True)


if __name__ == "__main__":
    if os.environ.get("E2E_WORKER") == "1":
        worker()
    else:
        total, R = int(sys.argv[1]), int(sys.argv[2])
        for window in (1, 0):
            launch(total, R, window)
