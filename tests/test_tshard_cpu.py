"""CPU, world_size 2 over gloo: the frame-axis (T) sharded UNet forward.  Each rank lowers its own
sharded denoise program and runs it with parallel.ShardedExecutor — the SAME orchestration the multi-GPU
path uses — with the CPU interpreter standing in for the HIP segments.  The concatenated result must
match the unsharded program (cross-frame GroupNorm statistics, +-1-frame conv halos and gathered
temporal-attention K/V all have to be exact for that)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from harness import rel_l2
from interp import Interp
from oracle import configs, synth, torch_port as tp
from sd_webui_text2video_amd import _lib as L
from sd_webui_text2video_amd import parallel
from sd_webui_text2video_amd import unet as U
from sd_webui_text2video_amd.program import TShardSpec


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Seg:
    def __init__(self, it, ops):
        self.it, self.ops = it, ops

    def run(self, ext, stream):
        self.it.run(ext, ops=self.ops)


def _inputs(F):
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 4, F, 8, 8, generator=g)
    y = torch.randn(1, 5, 1024, generator=g)
    return x, torch.tensor([613.0]), y


def _worker(rank, world, port, F, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cfg = configs.TINY_UNET
        net = U.UNetSD(**cfg, init_weights=False)
        synth.load_synth(net, seed=0)
        x, t, y = _inputs(F)
        spec = TShardSpec.make(F, world, rank)
        shard = parallel.TShard(dist.group.WORLD, list(range(world)), spec)
        comp = net._compile(1, spec.frames, 8, 8, 5, "f32", "f32", "f32", shard=spec)
        it = Interp(comp.prog, comp.packer.materialise(net.state_dict(), "cpu"))
        ex = parallel.ShardedExecutor(comp.prog, it.arena, shard, lambda ops: _Seg(it, ops))
        kinds = [op.kind for k, op in ex.steps if k == "coll"]
        # 39 temporal sites — 22 ResBlocks x 4 (cross-frame GroupNorm + (3,1,1) convolution) pairs and 17 TemporalTransformers.
        # TemporalTransformers per level (8x8 / 4x4 / 2x2 / 1x1 pixels): 6 / 5 / 5 / 1.  Where the pixel count divides by the
        # ranks the block is resharded frames <-> pixels (2 all-to-alls), elsewhere (1x1; everything at 3 ranks) it gathers
        # K/V (2 all-gathers); its GroupNorm gathers statistics (17).
        n_resharded = sum(n for px, n in ((64, 6), (16, 5), (4, 5), (1, 1)) if px % world == 0)
        assert kinds.count(L.OP_ALLTOALL) == 2 * n_resharded
        if os.environ.get("T2V_STATS_HALO", "1") != "0":
            # round 5: ONE exchange per temporal convolution — the statistics parts and the raw boundary frames travel together
            assert ex.n_collectives == 22 * 4 + 17 * 3 == 139
            assert kinds.count(L.OP_STATS_HALO) == 88 and kinds.count(L.OP_HALO_EXCHANGE) == 0
            assert kinds.count(L.OP_ALLGATHER) == 17 + 2 * (17 - n_resharded)
        else:
            # the two-exchange form (rounds 1-4): statistics all-gather, normalise, halo exchange of the normalised frames
            assert ex.n_collectives == 22 * 8 + 17 * 3
            assert kinds.count(L.OP_STATS_HALO) == 0 and kinds.count(L.OP_HALO_EXCHANGE) == 88
            assert kinds.count(L.OP_ALLGATHER) == 105 + 2 * (17 - n_resharded)
        out = torch.empty(1, 4, spec.frames, 8, 8)
        xl = x[:, :, spec.offset:spec.offset + spec.frames].contiguous()
        ex.run({L.EXT_X: xl, L.EXT_T: t, L.EXT_CTX: y, L.EXT_OUT: out}, None)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(2, 4), (3, 7), (4, 10)])      # 7 = 3+3+1, 10 = 3+3+3+1: uneven last slices
def test_tsharded_unet_matches_unsharded_gloo(world, F, monkeypatch):
    mgr = mp.Manager()

    def run(merged):
        monkeypatch.setenv("T2V_STATS_HALO", str(merged))         # inherited by the spawned ranks
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), F, ret), nprocs=world, join=True)
        return torch.cat([ret[r] for r in range(world)], dim=2)

    sharded = run(1)
    if world == 3:
        # the two-exchange form of rounds 1-4 (normalise, then exchange the normalised frames) gives the SAME bits: a neighbour that
        # normalises my raw boundary frame with the gathered statistics computes what I computed for it
        assert torch.equal(sharded, run(0))
    # unsharded reference: same program family, one rank
    cfg = configs.TINY_UNET
    net = U.UNetSD(**cfg, init_weights=False)
    sd = synth.load_synth(net, seed=0)
    x, t, y = _inputs(F)
    comp = net._compile(1, F, 8, 8, 5, "f32", "f32", "f32")
    it = Interp(comp.prog, comp.packer.materialise(net.state_dict(), "cpu"))
    whole = torch.empty(1, 4, F, 8, 8)
    it.run({L.EXT_X: x, L.EXT_T: t, L.EXT_CTX: y, L.EXT_OUT: whole})
    assert not torch.isnan(sharded).any()
    # Same arithmetic, but the sharded GEMMs have other M (other fp32 summation order in the CPU matmul):
    # 1e-7 differences flip fp16 roundings, which this random-weight network amplifies to ~1e-3 (the same
    # run-to-run level seen with atomics on the GPU).  A wrong halo / statistic / K-V slice is an O(0.1) error,
    # concentrated on the frames next to the shard boundary - so bound every frame separately.
    ref = tp.unet_forward(sd, cfg, x, t.long(), y)
    assert rel_l2(sharded, whole) < 4e-3
    assert rel_l2(sharded, ref) < 5e-3
    for f in range(F):
        assert rel_l2(sharded[:, :, f], ref[:, :, f]) < 6e-3, f


# ---- VideoCrafter (LVDM) UNet: every GroupNorm32 spans all frames, temporal attention has relative positions ------------
def _lvdm_inputs(F):
    g = torch.Generator().manual_seed(31)
    return torch.randn(1, 4, F, 8, 8, generator=g), torch.tensor([431.0]), torch.randn(1, 9, 768, generator=g)


def _lvdm_worker(rank, world, port, F, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from sd_webui_text2video_amd import videocrafter as VC
        net = VC.UNetModel(**configs.TINY_LVDM_UNET, init_weights=False)
        net.load_state_dict(synth.synth_state_dict(synth.param_spec(net), seed=0), strict=True)
        x, t, ctx = _lvdm_inputs(F)
        spec = TShardSpec.make(F, world, rank)
        shard = parallel.TShard(dist.group.WORLD, list(range(world)), spec)
        comp = net._compile(1, spec.frames, 8, 8, 9, "f32", "f32", "f32", shard=spec)
        it = Interp(comp.prog, comp.packer.materialise(net.state_dict(), "cpu"))
        ex = parallel.ShardedExecutor(comp.prog, it.arena, shard, lambda ops: _Seg(it, ops))
        assert ex.n_collectives > 0 and all(op.kind == L.OP_ALLGATHER for k, op in ex.steps if k == "coll")   # no halo: kernel_size_t = 1
        out = torch.empty(1, 4, spec.frames, 8, 8)
        ex.run({L.EXT_X: x[:, :, spec.offset:spec.offset + spec.frames].contiguous(), L.EXT_T: t, L.EXT_CTX: ctx, L.EXT_OUT: out}, None)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(2, 5), (3, 7)])       # 5 = 3 + 2, 7 = 3 + 3 + 1
def test_tsharded_lvdm_unet_matches_unsharded_gloo(world, F):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_lvdm_worker, args=(world, port, F, ret), nprocs=world, join=True)
    sharded = torch.cat([ret[r] for r in range(world)], dim=2)
    from sd_webui_text2video_amd import videocrafter as VC
    net = VC.UNetModel(**configs.TINY_LVDM_UNET, init_weights=False)
    sd = synth.synth_state_dict(synth.param_spec(net), seed=0)
    net.load_state_dict(sd, strict=True)
    x, t, ctx = _lvdm_inputs(F)
    ref = tp.lvdm_unet_forward(sd, configs.TINY_LVDM_UNET, x, t, ctx)
    assert not torch.isnan(sharded).any()
    assert rel_l2(sharded, ref) < 5e-3
    for f in range(F):
        assert rel_l2(sharded[:, :, f], ref[:, :, f]) < 7e-3, f
