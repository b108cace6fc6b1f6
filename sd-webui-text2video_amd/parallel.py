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

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def partition_frames(n_frames: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal frame ranges [f0, f1) (uneven tail allowed, e.g. 125 = 7x16 + 13)."""
    base, rem = divmod(n_frames, parts)
    out, f0 = [], 0
    for p in range(parts):
        n = base + (1 if p < rem else 0)
        out.append((f0, f0 + n))
        f0 += n
    return out


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
    def __init__(self, pipe, pair: CfgPair, frames, height, width, ddim_steps, guidance):
        self.pipe, self.pair = pipe, pair
        self.frames, self.height, self.width, self.ddim_steps, self.guidance = frames, height, width, ddim_steps, guidance
        n_videos = (pair.world + 1) // 2 if pair.world > 1 else 1
        self.frames_per_video_all_ranks = frames * n_videos
        self.unet_batch = 2 if pair.size == 1 else 1
        self.unet_frames = frames
        self.describe = ("1 GPU: cond+uncond batched as b=2" if pair.world == 1 else
                         f"{n_videos} video(s) in parallel, each on a CFG pair (cond | uncond UNet forwards on 2 GPUs, "
                         f"eps all-gather per step over RCCL), VAE frames split over the pair")

    @torch.no_grad()
    def __call__(self, cond, uncond, seed):
        pipe, pair = self.pipe, self.pair
        seed = seed + 1000 * pair.index            # every pair makes its own video
        if pair.size == 1:
            rgb, _ = pipe.infer_conditioned(cond, uncond, self.ddim_steps, self.frames, seed, self.guidance,
                                            self.width, self.height, 0.0, to_host=False)
            return rgb
        pipe.diffusion.get_sampler("DDIM_Gaussian", return_sampler=False)
        pipe.diffusion.sampler.cfg_parallel = pair
        _, x0 = pipe.infer_conditioned(cond, uncond, self.ddim_steps, self.frames, seed, self.guidance,
                                       self.width, self.height, 0.0, decode=False, _keep_sampler=True)
        f0, f1 = pair.my_frames(self.frames)
        rgb_local = pipe.decode_frames(x0[:, :, f0:f1])
        return pair.gather_frames(rgb_local, self.frames)


def make_runner(pipe, world: int, rank: int, *, frames, height, width, ddim_steps, guidance):
    return _Runner(pipe, CfgPair(world, rank), frames, height, width, ddim_steps, guidance)
