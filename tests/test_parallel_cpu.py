"""CPU, world_size 2 over gloo: the N>1 communication logic of parallel.py (pair layout, eps
exchange ordering, uneven frame partition + gather)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sd_webui_text2video_amd import parallel


def test_partition_frames_uneven():
    parts = parallel.partition_frames(125, 8)
    assert parts[0] == (0, 16) and parts[-1][1] == 125
    assert [b - a for a, b in parts] == [16] * 5 + [15] * 3
    assert parallel.partition_frames(24, 2) == [(0, 12), (12, 24)]
    assert parallel.partition_frames(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]


def test_pair_layout():
    assert [parallel.pair_layout(8, r) for r in range(8)] == [(r // 2, r % 2, 2) for r in range(8)]
    assert parallel.pair_layout(3, 2) == (1, 0, 1)
    assert parallel.pair_layout(1, 0) == (0, 0, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pair = parallel.CfgPair(world, rank)
        assert (pair.index, pair.role, pair.size) == (0, rank, 2)
        eps_local = torch.full((1, 4, 5, 2, 2), float(rank + 1))
        both = pair.exchange_eps(eps_local)
        assert both.shape == (2, 4, 5, 2, 2)
        assert torch.all(both[0] == 1.0) and torch.all(both[1] == 2.0)      # index 0 = conditional rank
        # uneven frame split: 5 frames over 2 ranks -> 3 + 2
        f0, f1 = pair.my_frames(5)
        assert (f0, f1) == ((0, 3) if rank == 0 else (3, 5))
        local = torch.arange(f0, f1, dtype=torch.uint8).view(-1, 1, 1, 1).expand(-1, 2, 2, 3).contiguous()
        full = pair.gather_frames(local, 5)
        assert full.shape == (5, 2, 2, 3)
        assert full[:, 0, 0, 0].tolist() == [0, 1, 2, 3, 4]
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_cfg_pair_collectives_gloo_world2():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def _worker4(rank, world, port, ret):
    """Two independent CFG pairs (bench.py --gpus 4 layout): every rank creates every group, exchanges stay inside the pair."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pair = parallel.CfgPair(world, rank)
        assert (pair.index, pair.role, pair.size) == (rank // 2, rank % 2, 2) and pair.members == [2 * (rank // 2), 2 * (rank // 2) + 1]
        both = pair.exchange_eps(torch.full((1, 4, 3, 2, 2), float(10 * pair.index + pair.role)))
        assert both[0].unique().tolist() == [10.0 * pair.index] and both[1].unique().tolist() == [10.0 * pair.index + 1]
        f0, f1 = pair.my_frames(4)
        local = (torch.arange(f0, f1, dtype=torch.uint8) + 100 * pair.index).view(-1, 1, 1, 1).expand(-1, 1, 1, 3).contiguous()
        full = pair.gather_frames(local, 4)
        assert full[:, 0, 0, 0].tolist() == [100 * pair.index + k for k in range(4)]
        dist.barrier()
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_cfg_pairs_gloo_world4():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker4, args=(4, port, ret), nprocs=4, join=True)
    assert dict(ret) == {0: 1, 1: 1, 2: 1, 3: 1}


def test_tshard_frame_bookkeeping_uneven():
    """north_star's 8-GPU layout: 125 frames over 4 slices (32+32+32+29) x 2 CFG roles; every frame is decoded by
    exactly one rank and the gathered order is the clip order."""
    from sd_webui_text2video_amd.program import TShardSpec
    for F, R in ((125, 4), (24, 2), (10, 4), (125, 8), (7, 3)):
        spec = TShardSpec.make(F, R, 0)
        assert sum(spec.counts) == F and all(c == spec.counts[0] for c in spec.counts[:-1]) and 1 <= spec.counts[-1] <= spec.counts[0]
        assert [TShardSpec.make(F, R, r).offset for r in range(R)] == [r * spec.counts[0] for r in range(R)]
        order = parallel.tshard_frame_order(spec.counts)
        covered = []
        for rank, f0, n in order:
            role, t = rank // R, rank % R
            a, b = parallel.tshard_decode_share(spec.counts, role, t)
            assert b - a == n and sum(spec.counts[:t]) + a == f0
            covered += list(range(f0, f0 + n))
        assert covered == list(range(F))
    assert TShardSpec.make(125, 4, 3).counts == (32, 32, 32, 29)
    import pytest
    with pytest.raises(ValueError):
        TShardSpec.make(4, 4, 0).counts and TShardSpec.make(5, 4, 0)      # 5 frames / 4 slices of 2 leaves an empty last slice


def test_make_runner_default_layouts():
    """Layout bookkeeping of the runners that need no process group (replicas / single GPU)."""
    class Pipe:          # the runner only stores it
        pass
    kw = dict(frames=24, height=256, width=256, ddim_steps=50, guidance=9.0)
    r = parallel.make_runner(Pipe(), 8, 3, mode="replicas", **kw)
    assert isinstance(r, parallel._ReplicaRunner) and r.frames_per_video_all_ranks == 24 * 8 and r.unet_batch == 2
    r1 = parallel.make_runner(Pipe(), 1, 0, **kw)
    assert type(r1) is parallel._Runner and r1.frames_per_video_all_ranks == 24 and r1.unet_batch == 2
    # several videos per batch and GPU (bench.py --videos V): frame accounting and the UNet batch follow
    r4 = parallel.make_runner(Pipe(), 1, 0, videos=4, **kw)
    assert r4.frames_per_video_all_ranks == 96 and r4.unet_batch == 8 and "4 videos per batch" in r4.describe
    r8 = parallel.make_runner(Pipe(), 8, 5, videos=2, mode="replicas", **kw)
    assert r8.frames_per_video_all_ranks == 24 * 8 * 2 and r8.unet_batch == 4 and "16 independent videos" in r8.describe
    calls = []

    class P2:
        def infer_conditioned(self, *a, **k):
            calls.append((a[4], k.get("videos")))
            return "rgb", None
    r8.pipe = P2()
    assert r8(None, None, 100) == "rgb" and calls == [(100 + 1000 * 5, 2)]      # every rank draws its own seeds


def test_collective_stack_selection(monkeypatch):
    """Which stack carries a group's data-path collectives (parallel._library_collectives): the library's RCCL communicator for CUDA
    tensors unless T2V_COLLECTIVES=host; over a gloo group only when forced (the one-GPU tests over tests/fake_rccl); never for CPU
    tensors.  A GroupComm of one rank, or on the host path, creates no communicator."""
    from sd_webui_text2video_amd import parallel as P
    monkeypatch.setattr(P, "_host_staged", lambda group: group == "gloo")
    monkeypatch.delenv("T2V_COLLECTIVES", raising=False)
    assert P._library_collectives("nccl", "cuda:0") and not P._library_collectives("gloo", "cuda:0")
    assert not P._library_collectives("nccl", "cpu")
    monkeypatch.setenv("T2V_COLLECTIVES", "host")
    assert not P._library_collectives("nccl", "cuda:0")
    monkeypatch.setenv("T2V_COLLECTIVES", "library")
    assert P._library_collectives("gloo", "cuda:0") and not P._library_collectives("gloo", "cpu")
    assert P.GroupComm("nccl", [3], 0).communicator("cuda:0") is None          # a group of one: nothing to exchange
    monkeypatch.setenv("T2V_COLLECTIVES", "host")
    assert P.GroupComm("nccl", [0, 1], 0).communicator("cuda:0") is None
