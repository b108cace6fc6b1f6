"""Multi-GPU decomposition of the hot path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) / xGMI.

The reference has exactly one collective — sample-level data parallelism whose decoded videos are
all-gathered (scripts/videocrafter/lvdm/utils/dist_utils.py:13-19, sample_text2video.py:123-125)
— and two sequential UNet calls per guided step (gaussian_sampler.py:161-162).  Round-1 layout:

  * classifier-free-guidance pair (2 ranks): rank role 0 evaluates the conditional UNet forward,
    role 1 the unconditional one (b=1 each, zero communication inside the UNet); ONE exchange per
    DDIM step — an all-gather of eps [1,4,F,h,w] inside the pair — after which both ranks apply
    the same fused update kernel (bitwise-identical, so x_t never diverges).
  * VAE decode: the frames of the finished latent are split contiguously over the ranks of the
    pair, decoded independently (frames are independent in the VAE) and gathered as uint8.
  * more than 2 GPUs: independent CFG pairs, one video each (data parallel over videos, the
    reference's own multi-GPU mode); per-GPU work is fixed => weak scaling.

Frame-axis (T) sharding inside one UNet forward — north_star's 8-GPU layout — needs an exchange
before every temporal op (39 sites per forward) and is the next step.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .program import OP_COLLECTIVE, Op, Program


def partition_frames(n_frames: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal frame ranges [f0, f1) (uneven tail allowed, e.g. 125 = 7x16 + 13)."""
    base, rem = divmod(n_frames, parts)
    out, f0 = [], 0
    for p in range(parts):
        n = base + (1 if p < rem else 0)
        out.append((f0, f0 + n))
        f0 += n
    return out


class TShard:
    """Membership of one T (frame-axis) shard group: `size` ranks hold contiguous, equal frame slices of
    one clip; this is slice `index`.  `group` is the torch.distributed process group (RCCL on GPUs,
    gloo in the CPU tests); `ranks[i]` = global rank of slice i."""

    def __init__(self, group, ranks: List[int], index: int):
        self.group, self.ranks, self.index, self.size = group, list(ranks), index, len(ranks)

    @property
    def prev(self) -> Optional[int]:
        return self.ranks[self.index - 1] if self.index > 0 else None

    @property
    def next(self) -> Optional[int]:
        return self.ranks[self.index + 1] if self.index + 1 < self.size else None


class ShardedExecutor:
    """Runs a denoise program that contains collective pseudo-ops: the compute ops between two
    collectives form a segment (one t2v_plan each); collectives run through torch.distributed on typed
    views of the SAME arena, stream-ordered with the kernels.  `make_segment(ops)` returns an object with
    .run(ext, stream) — a BoundProgram on the GPU, the CPU interpreter in the gloo tests — so the
    orchestration below is exactly what the multi-GPU path executes."""

    def __init__(self, prog: Program, arena: torch.Tensor, make_segment: Callable[[List[Op]], object]):
        self.prog, self.arena = prog, arena
        self.steps: List[Tuple[str, object]] = []
        cur: List[Op] = []
        for op in prog.ops:
            if op.kind == OP_COLLECTIVE:
                if cur:
                    self.steps.append(("seg", make_segment(cur)))
                    cur = []
                self.steps.append(("coll", op))
            else:
                cur.append(op)
        if cur:
            self.steps.append(("seg", make_segment(cur)))
        self.n_collectives = sum(1 for k, _ in self.steps if k == "coll")

    def _bytes(self, off: int, n: int) -> torch.Tensor:
        return self.arena[off: off + n]

    def run(self, ext: Dict[int, int], stream, shard: TShard):
        for kind, item in self.steps:
            if kind == "seg":
                item.run(ext, stream)
                continue
            meta = item.meta
            if meta["type"] == "allgather":
                full, nb = meta["full"], meta["part_bytes"]
                out = self._bytes(full.ref.off, nb * shard.size)
                mine = out[shard.index * nb: (shard.index + 1) * nb]
                dist.all_gather_into_tensor(out, mine, group=shard.group)
            elif meta["type"] == "halo":
                buf, fr, nf = meta["buf"], meta["frame_rows"], meta["frames"]
                row_b = buf.ld * buf.item
                fb = fr * row_b                                    # bytes of one frame
                base = buf.ref.off
                first, last = self._bytes(base + fb, fb), self._bytes(base + nf * fb, fb)
                halo0, halo1 = self._bytes(base, fb), self._bytes(base + (nf + 1) * fb, fb)
                ops = []
                if shard.prev is not None:
                    ops += [dist.P2POp(dist.isend, first, shard.prev, group=shard.group),
                            dist.P2POp(dist.irecv, halo0, shard.prev, group=shard.group)]
                if shard.next is not None:
                    ops += [dist.P2POp(dist.isend, last, shard.next, group=shard.group),
                            dist.P2POp(dist.irecv, halo1, shard.next, group=shard.group)]
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
            else:
                raise ValueError(meta["type"])

    # BoundProgram-compatible timing hook (per-op times are not defined across collectives)
    def run_timed(self, ext, stream):
        raise NotImplementedError("per-op timing is only available for unsharded programs")


def pair_layout(world: int, rank: int) -> Tuple[int, int, int]:
    """-> (pair index, role, pair size).  Ranks (2p, 2p+1) form CFG pair p; an odd last rank runs solo."""
    if world % 2 == 1 and rank == world - 1:
        return rank // 2, 0, 1
    return rank // 2, rank % 2, 2


class CfgPair:
    """Communication of one classifier-free-guidance pair (or a solo rank when size == 1)."""

    def __init__(self, world: int, rank: int):
        self.world, self.rank = world, rank
        self.index, self.role, self.size = pair_layout(world, rank)
        self.group = None
        if world > 1:
            # every rank must create every group, in the same order
            for p in range((world + 1) // 2):
                members = [r for r in (2 * p, 2 * p + 1) if r < world]
                g = dist.new_group(ranks=members)
                if rank in members:
                    self.group = g
                    self.members = members
        else:
            self.members = [0]

    def exchange_eps(self, eps_local: torch.Tensor) -> torch.Tensor:
        """[1,C,F,h,w] on each rank -> [2,C,F,h,w] (index 0 = conditional, 1 = unconditional)."""
        if self.size == 1:
            return eps_local
        out = torch.empty((2,) + tuple(eps_local.shape[1:]), dtype=eps_local.dtype, device=eps_local.device)
        dist.all_gather_into_tensor(out, eps_local.contiguous(), group=self.group)
        return out

    def my_frames(self, n_frames: int) -> Tuple[int, int]:
        return partition_frames(n_frames, self.size)[self.role]

    def gather_frames(self, local: torch.Tensor, n_frames: int) -> torch.Tensor:
        """uint8 [F_local,H,W,3] per rank -> [F,H,W,3] on every rank of the pair (variable counts)."""
        if self.size == 1:
            return local
        parts = partition_frames(n_frames, self.size)
        nmax = max(b - a for a, b in parts)          # pad to equal counts: one fixed-size all-gather
        send = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
        recv = torch.empty((self.size * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return torch.cat([recv[i * nmax: i * nmax + (b - a)] for i, (a, b) in enumerate(parts)], dim=0)


class _Runner:
    def __init__(self, pipe, pair: CfgPair, frames, height, width, ddim_steps, guidance, videos: int = 1):
        self.pipe, self.pair = pipe, pair
        self.frames, self.height, self.width, self.ddim_steps, self.guidance = frames, height, width, ddim_steps, guidance
        assert videos == 1 or pair.size == 1, "several videos per batch only in the one-GPU-per-video layouts"
        self.videos = videos
        n_videos = (pair.world + 1) // 2 if pair.world > 1 else 1
        self.frames_per_video_all_ranks = frames * n_videos * videos
        self.unet_batch = 2 * videos if pair.size == 1 else 1
        self.unet_frames = frames
        self.describe = ((f"1 GPU: cond+uncond batched as b=2" if videos == 1 else
                          f"1 GPU: {videos} videos per batch, cond+uncond batched as b={2 * videos}") if pair.world == 1 else
                         f"{n_videos} video(s) in parallel, each on a CFG pair (cond | uncond UNet forwards on 2 GPUs, "
                         f"eps all-gather per step over RCCL), VAE frames split over the pair")

    @torch.no_grad()
    def __call__(self, cond, uncond, seed):
        pipe, pair = self.pipe, self.pair
        seed = seed + 1000 * pair.index            # every pair makes its own video
        if pair.size == 1:
            rgb, _ = pipe.infer_conditioned(cond, uncond, self.ddim_steps, self.frames, seed, self.guidance,
                                            self.width, self.height, 0.0, to_host=False, videos=self.videos)
            return rgb
        pipe.diffusion.get_sampler("DDIM_Gaussian", return_sampler=False)
        pipe.diffusion.sampler.cfg_parallel = pair
        _, x0 = pipe.infer_conditioned(cond, uncond, self.ddim_steps, self.frames, seed, self.guidance,
                                       self.width, self.height, 0.0, decode=False, _keep_sampler=True)
        f0, f1 = pair.my_frames(self.frames)
        rgb_local = pipe.decode_frames(x0[:, :, f0:f1])
        return pair.gather_frames(rgb_local, self.frames)


class TShardTopology:
    """world = 2 (CFG roles) x R (frame shards).  rank = role * R + t.
    T group of a role = its R ranks (exchanges inside a UNet forward); pair group of a shard index t =
    {t, R + t} (one eps exchange per DDIM step).  Every rank creates every group, in the same order."""

    def __init__(self, world: int, rank: int):
        assert world >= 4 and world % 2 == 0
        self.world, self.rank = world, rank
        self.R = world // 2
        self.role, self.t = rank // self.R, rank % self.R
        self.size = 2                                   # CfgPair-compatible interface for the sampler
        self.tshard = None
        for role in range(2):
            ranks = list(range(role * self.R, (role + 1) * self.R))
            g = dist.new_group(ranks=ranks)
            if role == self.role:
                self.tshard = TShard(g, ranks, self.t)
        self.pair_group = None
        for t in range(self.R):
            g = dist.new_group(ranks=[t, self.R + t])
            if t == self.t:
                self.pair_group = g

    def exchange_eps(self, eps_local: torch.Tensor) -> torch.Tensor:
        out = torch.empty((2,) + tuple(eps_local.shape[1:]), dtype=eps_local.dtype, device=eps_local.device)
        dist.all_gather_into_tensor(out, eps_local.contiguous(), group=self.pair_group)
        return out                                       # index 0 = role 0 = conditional


class _TShardRunner:
    """One video of R*frames frames on 2R GPUs: frame slices of `frames` per rank along T, the CFG pair
    across the two roles.  Weak scaling: per-GPU work (frames per rank, b=1) is fixed as N grows."""

    def __init__(self, pipe, topo: TShardTopology, frames, height, width, ddim_steps, guidance):
        self.pipe, self.topo = pipe, topo
        self.frames_local, self.height, self.width, self.ddim_steps, self.guidance = frames, height, width, ddim_steps, guidance
        self.frames_total = frames * topo.R
        self.frames_per_video_all_ranks = self.frames_total
        self.unet_batch, self.unet_frames = 1, frames
        assert frames % 2 == 0, "the two CFG roles split the VAE decode of their shared frame slice"
        self.describe = (f"one {self.frames_total}-frame video on {topo.world} GPUs: T-axis sharding x{topo.R} inside the UNet "
                         f"({frames} frames per rank; statistics / halo / K-V exchanges over RCCL before temporal ops) "
                         f"x CFG pair (eps all-gather per step); VAE frames split over all ranks")

    @torch.no_grad()
    def __call__(self, cond, uncond, seed):
        pipe, topo = self.pipe, self.topo
        dev = pipe.device
        pipe.sd_model.t_shard = topo.tshard
        pipe.diffusion.get_sampler("DDIM_Gaussian", return_sampler=False)
        sampler = pipe.diffusion.sampler
        sampler.cfg_parallel = topo
        _, noise, _ = pipe.diffusion.get_noise(1, 4, self.frames_total, self.height, self.width, seed=seed)
        Fl = self.frames_local
        x_T = noise[:, :, topo.t * Fl:(topo.t + 1) * Fl].contiguous()
        from .samplers import SamplerStepCallback
        x0 = sampler.sample(S=self.ddim_steps, conditioning=cond.to(dev), unconditional_conditioning=uncond.to(dev),
                            x_T=x_T, shape=tuple(x_T.shape), unconditional_guidance_scale=self.guidance, eta=0.0,
                            callback=SamplerStepCallback("DDIM_Gaussian", self.ddim_steps, progress=False))
        pipe.sd_model.t_shard = None
        half = Fl // 2                                   # both roles hold the same slice: decode one half each
        rgb = pipe.decode_frames(x0[:, :, topo.role * half:(topo.role + 1) * half])
        out = torch.empty((topo.world * half,) + tuple(rgb.shape[1:]), dtype=rgb.dtype, device=rgb.device)
        dist.all_gather_into_tensor(out, rgb.contiguous())
        # rank (role, t) decoded frames [t*Fl + role*half, +half): put them in frame order
        order = [(f // Fl) + ((f % Fl) // half) * topo.R for f in range(0, self.frames_total, half)]
        return torch.cat([out[r * half:(r + 1) * half] for r in order], dim=0)


class _ReplicaRunner(_Runner):
    """Throughput layout: every GPU generates its own videos (cond + uncond batched as b=2, exactly the 1-GPU
    workload) — videos are independent objects, so there is no data-path collective at all."""

    def __init__(self, pipe, world, rank, frames, height, width, ddim_steps, guidance, videos: int = 1):
        super().__init__(pipe, CfgPair(1, 0), frames, height, width, ddim_steps, guidance, videos=videos)
        self.rank = rank
        self.frames_per_video_all_ranks = frames * world * videos
        self.describe = (f"{world * videos} independent videos in flight, {videos} per GPU (cond+uncond batched as "
                         f"b={2 * videos}, no data-path collective)")

    def __call__(self, cond, uncond, seed):
        return super().__call__(cond, uncond, seed + 1000 * self.rank)


def make_runner(pipe, world: int, rank: int, *, frames, height, width, ddim_steps, guidance, mode: str = "auto",
                videos: int = 1):
    """'replicas': one video per GPU (throughput; no collective).  'pairs': one video per CFG pair — cond | uncond UNet
    forwards on 2 GPUs, one eps all-gather per step (latency: 1.65x faster per video).  'tshard': one long video on
    2 x R GPUs, T-sharded inside the UNet (world >= 4, even; statistics / halo / K-V exchanges before the temporal ops).
    'auto' = replicas for world > 1: the metric is whole-job frames/s and videos are independent; the exchange-carrying
    layouts (validated with gloo and in lock-step on one GPU) stay opt-in until their RCCL cost has been measured on a
    multi-GPU node."""
    if mode == "auto":
        mode = "replicas" if world > 1 else "pairs"
    if mode == "replicas":
        return _ReplicaRunner(pipe, world, rank, frames, height, width, ddim_steps, guidance, videos=videos)
    if mode == "tshard":
        assert videos == 1
        return _TShardRunner(pipe, TShardTopology(world, rank), frames, height, width, ddim_steps, guidance)
    return _Runner(pipe, CfgPair(world, rank), frames, height, width, ddim_steps, guidance, videos=videos)
