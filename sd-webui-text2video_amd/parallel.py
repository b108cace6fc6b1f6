"""Multi-GPU decomposition of the hot path: one process per GPU, RCCL over xGMI.

The reference has exactly one collective — sample-level data parallelism whose decoded videos are
all-gathered (scripts/videocrafter/lvdm/utils/dist_utils.py:13-19, sample_text2video.py:123-125)
— and two sequential UNet calls per guided step (gaussian_sampler.py:161-162).  Layouts here:

  * T-axis (frame) sharding x CFG pair (north_star's layout; even N >= 4, mode "tshard"): world = 2 roles x R frame slices.
    The clip's frames are split contiguously over the R ranks of a role (slices of ceil(F / R) frames, a shorter last
    one: 125 = 32 + 32 + 32 + 29); all spatial work is frame-local, and before each temporal op the lowering inserts
    an exchange op into the denoise program (program.py: T2V_OP_STATS_HALO — the GroupNorm statistics parts AND the raw boundary
    frames of the (3,1,1) convolution behind that norm in one grouped exchange, T2V_OP_ALLGATHER of the other cross-frame
    GroupNorm statistics, T2V_OP_ALLTOALL resharding frames <-> pixels around each
    TemporalTransformer — or an all-gather of its K/V where the pixel count does not divide).  The library
    executes them with RCCL on the launch stream (csrc/comm.hip), so a sharded forward is ONE host call.  The two
    roles evaluate the conditional / unconditional forward of the same frames; one eps all-gather per DDIM step
    inside each {cond, uncond} pair, then both apply the same bitwise-deterministic update kernel.
  * classifier-free-guidance pair alone (N = 2): role 0 / 1 evaluate the conditional / unconditional forward (b = 1
    each, no communication inside the UNet), one eps all-gather per step.
  * replicas: every GPU makes its own videos, no data-path collective (the reference's own data-parallel mode).
  * VAE decode: the frames of the finished latent are split over all ranks that hold them, decoded independently
    (frames are independent in the VAE) and gathered as uint8 — the reference's gather of decoded samples.
"""
from __future__ import annotations

import ctypes
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib as L
from .program import COLLECTIVE_KINDS, Op, Program, TShardSpec


def _host_staged(group) -> bool:
    """gloo moves host memory only: device tensors are staged through the host (multi-process tests that share one GPU)."""
    return dist.get_backend(group) == "gloo"


def all_gather_into(out: torch.Tensor, mine: torch.Tensor, group=None):
    """dist.all_gather_into_tensor, also for device tensors over a gloo group."""
    if out.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, mine.cpu().contiguous(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, mine, group=group)


def _library_collectives(group, device) -> bool:
    """Do this group's collectives on `device` run inside libt2v_hip.so (RCCL communicator owned by the library)?  CUDA tensors over an
    RCCL-capable group by default; T2V_COLLECTIVES=host keeps torch.distributed, =library forces the library communicator even over a
    gloo group (tests: several ranks on ONE GPU with T2V_RCCL_SONAME pointing at tests/fake_rccl)."""
    mode = os.environ.get("T2V_COLLECTIVES", "auto")
    return torch.device(device).type == "cuda" and mode != "host" and (not _host_staged(group) or mode == "library")


class GroupComm:
    """The library communicator of one torch.distributed group, created on first use, and the gathers that run over it.  The
    per-step eps exchange of a CFG pair and the gather of the decoded uint8 frames go through `t2v_comm_all_gather` on the current
    stream — the same RCCL stack the exchanges INSIDE a sharded forward use (csrc/comm.hip) — instead of torch.distributed's."""

    def __init__(self, group, ranks: List[int], index: int):
        self.group, self.ranks, self.index = group, list(ranks), index
        self._comm: Optional["Communicator"] = None

    def communicator(self, device) -> Optional["Communicator"]:
        if len(self.ranks) < 2 or not _library_collectives(self.group, device):
            return None
        if self._comm is None:
            self._comm = Communicator(self.group, self.ranks, self.index, device)
        return self._comm

    def all_gather_into(self, out: torch.Tensor, mine: torch.Tensor):
        """out = concatenation over the group's ranks of `mine` (equal sizes)."""
        comm = self.communicator(out.device)
        if comm is None:
            return all_gather_into(out, mine, group=self.group)
        assert out.is_contiguous() and out.numel() * out.element_size() == len(self.ranks) * mine.numel() * mine.element_size()
        nbytes = mine.numel() * mine.element_size()
        flat = out.view(-1).view(torch.uint8)
        flat[self.index * nbytes: (self.index + 1) * nbytes].copy_(mine.contiguous().view(-1).view(torch.uint8))
        L.check(comm._lib.t2v_comm_all_gather(comm.handle, ctypes.c_void_p(out.data_ptr()), nbytes,
                                              ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)))


def exchange_pairs(sends, recvs, group):
    """sends / recvs: [(tensor, global rank)] — one batched neighbour exchange."""
    staged = bool(sends or recvs) and (sends + recvs)[0][0].is_cuda and _host_staged(group)
    if staged:
        hs = [(t.cpu(), r) for t, r in sends]
        hr = [(torch.empty(t.shape, dtype=t.dtype), r) for t, r in recvs]
    else:
        hs, hr = sends, recvs
    ops = [dist.P2POp(dist.isend, t, r, group=group) for t, r in hs] + [dist.P2POp(dist.irecv, t, r, group=group) for t, r in hr]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged:
        for (dst, _), (src, _) in zip(recvs, hr):
            dst.copy_(src)


def partition_frames(n_frames: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal frame ranges [f0, f1) (uneven tail allowed, e.g. 125 = 7x16 + 13)."""
    base, rem = divmod(n_frames, parts)
    out, f0 = [], 0
    for p in range(parts):
        n = base + (1 if p < rem else 0)
        out.append((f0, f0 + n))
        f0 += n
    return out


class Communicator:
    """t2v_comm handle (include/t2v_hip.h): an RCCL communicator owned by libt2v_hip.so for the collective ops of a
    denoise program.  The 128-byte unique id comes from `index` 0 of the group and travels over torch.distributed."""

    def __init__(self, group, ranks: List[int], index: int, device):
        lib = L.load()
        buf = ctypes.create_string_buffer(128)
        if index == 0:
            L.check(lib.t2v_comm_unique_id(buf))
        box = [bytes(buf.raw)]
        if len(ranks) > 1:      # the id travels over the group itself (host tensors for a gloo group)
            dist.broadcast_object_list(box, src=ranks[0], group=group,
                                       device=torch.device("cpu") if _host_staged(group) else torch.device(device))
        handle = ctypes.c_void_p()
        L.check(lib.t2v_comm_create(box[0], len(ranks), index, ctypes.byref(handle)))
        self.handle, self._lib, self.size = handle, lib, len(ranks)
        self.window = self._attach_window(group, ranks, device)

    def _attach_window(self, group, ranks, device) -> Optional[str]:
        """Peer window (include/t2v_hip.h ABI 8, csrc/comm.hip): every rank allocates a mailbox, the 64-byte IPC handles travel over
        torch.distributed, every rank maps the others' — from then on an exchange whose messages fit a slot is one kernel of the library
        (stores into the peer's window over xGMI + a sequence flag) instead of an RCCL group call.  Collective decision: the window is
        used only if EVERY rank of the group could create and open it (otherwise all stay on RCCL).  T2V_PEER_WINDOW=0 switches it off;
        T2V_PEER_SLOT_MB sizes a slot (default 16: the largest reshard chunk of a 32-frame slice at the 32x32 level is 8.4 MB).
        -> a description for bench.py's `rccl_communicators`, None when not attached."""
        n = len(ranks)
        if n < 2 or n > 8 or os.environ.get("T2V_PEER_WINDOW", "1") == "0":
            return None
        slot = int(float(os.environ.get("T2V_PEER_SLOT_MB", "16")) * (1 << 20))
        cpu = torch.device("cpu") if _host_staged(group) else torch.device(device)
        buf = ctypes.create_string_buffer(64)
        ok = self._lib.t2v_comm_window_create(self.handle, slot, buf) == 0
        mine = torch.tensor([1 if ok else 0] + list(buf.raw), dtype=torch.uint8, device=cpu)
        allh = [torch.empty_like(mine) for _ in range(n)]
        dist.all_gather(allh, mine, group=group)
        if not all(int(h[0]) == 1 for h in allh):
            self._lib.t2v_comm_window_open(self.handle, None)
            return None
        blob = b"".join(bytes(h[1:].cpu().tolist()) for h in allh)
        ok = self._lib.t2v_comm_window_open(self.handle, blob) == 0
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cpu)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            # some rank could not map a peer: nobody may use the windows (a rank that opened them would push into mailboxes nobody reads)
            self._lib.t2v_comm_window_open(self.handle, None)
            return None
        kind = (self._lib.t2v_comm_window_kind(self.handle) or b"").decode()
        return f"peer window: {n} ranks x 2 slots x {slot >> 20} MiB per rank ({kind} device memory), device-initiated stores + sequence flags (csrc/comm.hip)"

    def counters(self) -> Tuple[int, int]:
        """(exchanges that went over the peer window, exchanges that went through RCCL) since this communicator was created."""
        out = (ctypes.c_uint64 * 2)()
        self._lib.t2v_comm_counters(self.handle, out)
        return int(out[0]), int(out[1])

    def __del__(self):
        try:
            if self.handle:
                self._lib.t2v_comm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class TShard:
    """Membership of one T (frame-axis) shard group: `spec.size` ranks hold contiguous frame slices of one clip
    (spec.counts frames each; this is slice spec.index).  `group` is the torch.distributed process group of the slice
    holders (RCCL on GPUs, gloo in the CPU tests); `ranks[i]` = global rank of slice i."""

    def __init__(self, group, ranks: List[int], spec: TShardSpec):
        assert len(ranks) == spec.size
        self.group, self.ranks, self.spec = group, list(ranks), spec
        self.index, self.size = spec.index, spec.size
        self._comm: Optional[Communicator] = None

    @property
    def prev(self) -> Optional[int]:
        return self.ranks[self.index - 1] if self.index > 0 else None

    @property
    def next(self) -> Optional[int]:
        return self.ranks[self.index + 1] if self.index + 1 < self.size else None

    def communicator(self, device) -> Optional[Communicator]:
        """The in-library communicator for programs on `device`; None = run the exchanges from the host through
        torch.distributed (CPU / gloo groups, or T2V_COLLECTIVES=host).  T2V_COLLECTIVES=library forces the library communicator
        even over a gloo group (tests: several ranks on ONE GPU with T2V_RCCL_SONAME pointing at tests/fake_rccl)."""
        if not _library_collectives(self.group, device):
            return None
        if self._comm is None:
            self._comm = Communicator(self.group, self.ranks, self.index, device)
        return self._comm


class ShardedExecutor:
    """Host-side executor of a denoise program that contains collective ops: the compute ops between two collectives
    form a segment, collectives run through torch.distributed on typed views of the SAME arena.  `make_segment(ops)`
    returns an object with .run(ext, stream) — the CPU interpreter in the gloo tests, a BoundProgram when the exchanges
    are kept on the host (T2V_COLLECTIVES=host) or emulated in lock-step on one GPU (tests/harness.py).  It reads the
    very op records the library executes (byte counts, part index, neighbour ranks), so the gloo tests cover what the
    RCCL path is handed."""

    def __init__(self, prog: Program, arena: torch.Tensor, shard: Optional[TShard], make_segment: Callable[[List[Op]], object]):
        self.prog, self.arena, self.shard = prog, arena, shard
        self.steps: List[Tuple[str, object]] = []
        cur: List[Op] = []
        for op in prog.ops:
            if op.kind in COLLECTIVE_KINDS:
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

    def run(self, ext: Dict[int, int], stream):
        shard = self.shard
        for kind, item in self.steps:
            if kind == "seg":
                item.run(ext, stream)
                continue
            i = item.i
            nb = (i[0] & 0xFFFFFFFF) | (i[1] << 32)
            base = item.p[0].off
            if item.kind == L.OP_ALLGATHER:
                assert (i[2], i[3]) == (shard.size, shard.index)
                out = self._bytes(base, nb * shard.size)
                mine = out[shard.index * nb: (shard.index + 1) * nb]
                all_gather_into(out, mine, group=shard.group)
            elif item.kind == L.OP_ALLTOALL:
                # frame <-> pixel resharding (layouts: include/t2v_hip.h, csrc/comm.hip)
                n, me, base_cnt, last_cnt, direction = i[2], i[3], i[4], i[5], i[6]
                assert (n, me) == (shard.size, shard.index)
                cnt = lambda q: last_cnt if q == n - 1 else base_cnt
                sbase, rbase = item.p[0].off, item.p[1].off
                sends, recvs = [], []
                for q in range(n):
                    if q == me:
                        continue
                    if direction == 0:
                        sends.append((self._bytes(sbase + q * cnt(me) * nb, cnt(me) * nb), shard.ranks[q]))
                        recvs.append((self._bytes(rbase + q * base_cnt * nb, cnt(q) * nb), shard.ranks[q]))
                    else:
                        sends.append((self._bytes(sbase + q * base_cnt * nb, cnt(q) * nb), shard.ranks[q]))
                        recvs.append((self._bytes(rbase + q * cnt(me) * nb, cnt(me) * nb), shard.ranks[q]))
                exchange_pairs(sends, recvs, shard.group)
            else:   # OP_HALO_EXCHANGE: frame 1 -> prev, frame F -> next; their boundary frames into frame 0 / F + 1
                if item.kind == L.OP_STATS_HALO:
                    # statistics parts to every rank, then the RAW boundary frames of the halo-padded buffer p[1] to the neighbours
                    # (the library issues both as one group of transfers; the bytes and their places are the same)
                    assert (i[2], i[3]) == (shard.size, shard.index)
                    out = self._bytes(base, nb * shard.size)
                    all_gather_into(out, out[shard.index * nb: (shard.index + 1) * nb], group=shard.group)
                    base, nb = item.p[1].off, (i[4] & 0xFFFFFFFF) | (i[5] << 32)
                    nf, prev, nxt = i[6], i[7], i[8]
                else:
                    nf, prev, nxt = i[2], i[3], i[4]
                first, last = self._bytes(base + nb, nb), self._bytes(base + nf * nb, nb)
                halo0, halo1 = self._bytes(base, nb), self._bytes(base + (nf + 1) * nb, nb)
                sends, recvs = [], []
                if prev >= 0:
                    sends.append((first, shard.ranks[prev]))
                    recvs.append((halo0, shard.ranks[prev]))
                if nxt >= 0:
                    sends.append((last, shard.ranks[nxt]))
                    recvs.append((halo1, shard.ranks[nxt]))
                exchange_pairs(sends, recvs, shard.group)

    # BoundProgram-compatible timing hook (per-op times are not defined across host-side collectives)
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
        self.gcomm = GroupComm(self.group, self.members, self.role) if self.size == 2 else None

    def exchange_eps(self, eps_local: torch.Tensor) -> torch.Tensor:
        """[1,C,F,h,w] on each rank -> [2,C,F,h,w] (index 0 = conditional, 1 = unconditional)."""
        if self.size == 1:
            return eps_local
        out = torch.empty((2,) + tuple(eps_local.shape[1:]), dtype=eps_local.dtype, device=eps_local.device)
        self.gcomm.all_gather_into(out, eps_local.contiguous())
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
        self.gcomm.all_gather_into(recv, send)
        return torch.cat([recv[i * nmax: i * nmax + (b - a)] for i, (a, b) in enumerate(parts)], dim=0)


class _Runner:
    def __init__(self, pipe, pair: CfgPair, frames, height, width, ddim_steps, guidance, videos: int = 1, eta: float = 0.0):
        self.pipe, self.pair, self.eta = pipe, pair, float(eta)
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

    def communicators(self) -> List[dict]:
        """What this layout created for its data path (bench.py reports it: `config.rccl_communicators`)."""
        if self.pair.size == 1:
            return []
        lib = self.pair.gcomm._comm is not None
        return [{"kind": ("libt2v_hip t2v_comm: RCCL communicator owned by the library" if lib else
                          f"torch.distributed {dist.get_backend(self.pair.group)} group") + " (eps all-gather per step, uint8 frame gather)",
                 "size": self.pair.size}]

    @torch.no_grad()
    def self_check(self, cond) -> Optional[dict]:
        """CFG pair: the eps exchange / frame gather through the pair's library communicator (`t2v_comm_all_gather`, RCCL) against
        torch.distributed's all-gather on the same seeded data — equal on both ranks, or bench.py times nothing.  None for a layout
        without data-path collectives.  Collective: every rank of the job calls it."""
        if self.pair.size == 1:
            return None
        gc, dev = self.pair.gcomm, self.pipe.device
        in_library = gc.communicator(dev) is not None
        released = None

        def once() -> bool:
            equal = True
            for n in (4099, 4100) if in_library else ():      # 16 396 bytes: RCCL carries it; 16 400 bytes = 16-byte units: eligible for the peer window
                mine = torch.full((n,), float(self.pair.rank + 1), device=dev) + torch.arange(n, device=dev)
                a, b = torch.empty(2 * n, device=dev), torch.empty(2 * n, device=dev)
                try:
                    gc.all_gather_into(a, mine)
                    torch.cuda.synchronize(dev)
                    L.async_status()                            # a peer-window wait that gave up shows here
                except L.T2VError:
                    equal = False
                    a.zero_()
                all_gather_into(b, mine, group=gc.group)
                torch.cuda.synchronize(dev)
                equal = equal and bool(torch.equal(a, b))
            flag = torch.tensor([0 if equal else 1], dtype=torch.int32)
            if not _host_staged(None):
                flag = flag.to(dev)
            dist.all_reduce(flag)
            return int(flag.item()) == 0

        ok = once()
        if not ok and in_library and gc._comm.window:
            # round 6: the gathers went over the peer window and differ (or a wait gave up): release it on both ranks (same verdict
            # everywhere: all-reduced) and check the RCCL transport before giving the layout up
            gc._comm._lib.t2v_comm_window_open(gc._comm.handle, None)
            gc._comm.window, released = None, "released after a failed check over the peer window; this result is the RCCL transport's"
            ok = once()
        return {"ok": ok, "eps_and_frame_gathers_in_library": bool(in_library),
                "peer_windows": released or (gc._comm.window if in_library else None),
                "compared": ("t2v_comm_all_gather on the pair's library communicator vs torch.distributed: equal on every rank" if in_library
                             else "the pair's gathers run through torch.distributed in this set-up: nothing to compare")}

    @torch.no_grad()
    def __call__(self, cond, uncond, seed):
        pipe, pair = self.pipe, self.pair
        seed = seed + 1000 * pair.index            # every pair makes its own video
        if pair.size == 1:
            rgb, _ = pipe.infer_conditioned(cond, uncond, self.ddim_steps, self.frames, seed, self.guidance,
                                            self.width, self.height, self.eta, to_host=False, videos=self.videos)
            return rgb
        from .samplers import SharedNoise
        pipe.diffusion.get_sampler("DDIM_Gaussian", return_sampler=False)
        pipe.diffusion.sampler.cfg_parallel = pair
        # eta > 0: both ranks of the pair draw the SAME per-step noise (one seeded generator each), so x_t stays bit-identical
        pipe.diffusion.sampler.shared_noise = SharedNoise(seed, self.frames, 0, pipe.device) if self.eta != 0.0 else None
        _, x0 = pipe.infer_conditioned(cond, uncond, self.ddim_steps, self.frames, seed, self.guidance,
                                       self.width, self.height, self.eta, decode=False, _keep_sampler=True)
        f0, f1 = pair.my_frames(self.frames)
        rgb_local = pipe.decode_frames(x0[:, :, f0:f1])
        return pair.gather_frames(rgb_local, self.frames)


class TShardTopology:
    """world = 2 (CFG roles) x R (frame slices).  rank = role * R + t.
    T group of a role = its R ranks (exchanges inside a UNet forward); pair group of a slice index t =
    {t, R + t} (one eps exchange per DDIM step).  Every rank creates every group, in the same order."""

    def __init__(self, world: int, rank: int, total_frames: int):
        assert world >= 4 and world % 2 == 0
        self.world, self.rank = world, rank
        self.R = world // 2
        self.role, self.t = rank // self.R, rank % self.R
        self.size = 2                                   # CfgPair-compatible interface for the sampler
        self.spec = TShardSpec.make(total_frames, self.R, self.t)
        self.tshard = None
        for role in range(2):
            ranks = list(range(role * self.R, (role + 1) * self.R))
            g = dist.new_group(ranks=ranks)
            if role == self.role:
                self.tshard = TShard(g, ranks, self.spec)
        self.pair_group = None
        for t in range(self.R):
            g = dist.new_group(ranks=[t, self.R + t])
            if t == self.t:
                self.pair_group = g
        self.pair_comm = GroupComm(self.pair_group, [self.t, self.R + self.t], self.role)     # eps exchange per step
        self.world_comm = GroupComm(None, list(range(world)), rank)                            # uint8 frame gather per video

    def exchange_eps(self, eps_local: torch.Tensor) -> torch.Tensor:
        out = torch.empty((2,) + tuple(eps_local.shape[1:]), dtype=eps_local.dtype, device=eps_local.device)
        self.pair_comm.all_gather_into(out, eps_local.contiguous())
        return out                                       # index 0 = role 0 = conditional

    def decode_share(self) -> Tuple[int, int]:
        return tshard_decode_share(self.spec.counts, self.role, self.t)

    def frame_order(self) -> List[Tuple[int, int, int]]:
        return tshard_frame_order(self.spec.counts)


def tshard_decode_share(counts, role: int, t: int) -> Tuple[int, int]:
    """Frames [a, b) of slice t's latent that the rank of `role` decodes: the two roles hold the same slice and split
    it (the conditional role takes the larger half of an odd count)."""
    n = counts[t]
    half = (n + 1) // 2
    return (0, half) if role == 0 else (half, n)


def tshard_frame_order(counts) -> List[Tuple[int, int, int]]:
    """(global rank, first clip frame, count) of every rank's decoded share, in clip order (rank = role * R + t)."""
    R, out, off = len(counts), [], 0
    for t, n in enumerate(counts):
        half = (n + 1) // 2
        out.append((t, off, half))
        if n - half:
            out.append((R + t, off + half, n - half))
        off += n
    return out


class _TShardRunner:
    """ONE video of `frames` frames on 2R GPUs: contiguous frame slices along T (uneven tail allowed), the CFG pair
    across the two roles.  Strong scaling of a fixed clip: frames/s = frames / (time of the whole video)."""

    def __init__(self, pipe, topo: TShardTopology, frames, height, width, ddim_steps, guidance, eta: float = 0.0):
        self.pipe, self.topo, self.eta = pipe, topo, float(eta)
        self.frames_total, self.height, self.width, self.ddim_steps, self.guidance = frames, height, width, ddim_steps, guidance
        self.frames_per_video_all_ranks = frames
        self.unet_batch, self.unet_frames = 1, topo.spec.frames
        counts = "+".join(str(c) for c in topo.spec.counts)
        self.describe = (f"one {frames}-frame video on {topo.world} GPUs: T-axis sharding x{topo.R} inside the UNet "
                         f"({counts} frames per slice; statistics / halo / K-V exchanges as program ops over RCCL on the launch "
                         f"stream) x CFG pair (eps all-gather per step); VAE frames split over all ranks")

    def communicators(self) -> List[dict]:
        """The communicators this rank's data path runs over (bench.py: `config.rccl_communicators`): the library's own RCCL
        communicator of the T group (created lazily by the first sharded forward; None = the exchanges went through the host
        executor), the CFG pair group and the world group of the frame gather."""
        ts = self.topo.tshard
        lib_comm = ts._comm.size if ts._comm is not None else None

        def stack(gc: GroupComm, backend: str) -> str:
            return "libt2v_hip t2v_comm (RCCL, t2v_comm_all_gather on the launch stream)" if gc._comm is not None else f"torch.distributed {backend} group"

        return [{"kind": "libt2v_hip t2v_comm: RCCL communicator owned by the library (statistics + raw boundary frames as one grouped "
                         "exchange per temporal convolution, statistics / K-V all-gathers, frame<->pixel all-to-all as program ops on the "
                         "launch stream)", "size": lib_comm, "t_group_ranks": ts.size,
                 "peer_window": (ts._comm.window if ts._comm is not None else None),
                 "exchanges_via_window_and_via_rccl": (list(ts._comm.counters()) if ts._comm is not None else None)},
                {"kind": stack(self.topo.pair_comm, dist.get_backend(self.topo.pair_group)) + ": eps all-gather per step", "size": 2},
                {"kind": stack(self.topo.world_comm, dist.get_backend()) + ": uint8 frame gather per video", "size": self.topo.world}]

    @torch.no_grad()
    def self_check(self, cond) -> dict:
        """First forward of the T-sharded UNet, twice, on the same seeded inputs: (1) the production path — the exchanges are ops of
        the denoise program executed by the library over its own RCCL communicator (csrc/comm.hip) — and (2) the same op records
        through `ShardedExecutor`, whose collectives go through torch.distributed (the path the gloo CPU tests pin against the
        unsharded forward).  Both must be BIT-equal on every rank; bench.py times nothing if they are not (a wrong
        answer must not be timed).  Collective: every rank of the job calls it."""
        first = self._self_check_once(cond)
        if first["ok"] or not first.pop("_windows"):
            first.pop("_windows", None)
            return first
        # Round 6: the exchanges went over peer windows (device-initiated stores into IPC-mapped mailboxes) and the result differs from the
        # host executor's, or a wait gave up.  Every rank has the same verdict (all-reduced): release the windows everywhere and check the
        # RCCL transport before anything is given up.
        for comm in self._library_comms():
            comm._lib.t2v_comm_window_open(comm.handle, None)
            comm.window = None
        second = self._self_check_once(cond)
        second.pop("_windows", None)
        second["peer_windows"] = "released: the first check over peer windows failed (" + first["compared"] + "); this result is the RCCL transport's"
        return second

    def _library_comms(self) -> List[Communicator]:
        topo = self.topo
        return [c for c in (topo.tshard._comm, topo.pair_comm._comm, topo.world_comm._comm) if c is not None]

    def _all_ok(self, ok: bool) -> bool:
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32)
        if not _host_staged(None):
            flag = flag.to(self.pipe.device)
        dist.all_reduce(flag)
        return int(flag.item()) == 0

    def _self_check_once(self, cond) -> dict:
        from .program import BoundProgram
        net, topo, dev = self.pipe.sd_model, self.topo, self.pipe.device
        spec = topo.spec
        g = torch.Generator().manual_seed(4242)                    # the same clip on every rank; each takes its slice
        x = torch.randn(1, 4, self.frames_total, self.height // 8, self.width // 8, generator=g)
        x = x[:, :, spec.offset:spec.offset + spec.frames].contiguous().to(dev)
        t = torch.tensor([500.0], device=dev)
        c = cond.to(dev)
        saved_env = os.environ.get("T2V_COLLECTIVES")
        net.t_shard = topo.tshard
        n_coll, in_library, equal, why = 0, False, True, ""
        try:
            try:
                out_lib = net(x, t, c).clone()
                torch.cuda.synchronize(dev)
                L.async_status()                                    # a peer-window wait that gave up shows here, not in the forward's status
            except L.T2VError as e:
                equal, why = False, f"the library forward failed: {e}"
            # every rank learns whether every rank's library forward went through BEFORE the host-executor forward (a collective) starts
            lib_ok = self._all_ok(equal)
            comp = next((cmp for key, cmp in net._programs.items() if spec in key), None)
            if comp is not None:
                n_coll = sum(1 for op in comp.prog.ops if op.kind in COLLECTIVE_KINDS)
                in_library = isinstance(comp.bound, BoundProgram) and comp.bound.comm is not None
            if not lib_ok:
                equal = False
                why = why or "the library forward failed on another rank"
            elif in_library:
                keep_bound, keep_arena = comp.bound, comp.arena
                os.environ["T2V_COLLECTIVES"] = "host"
                comp.bound = None                                   # re-bind: fresh arena, exchanges from the host
                out_host = net(x, t, c).clone()
                assert isinstance(comp.bound, ShardedExecutor)
                torch.cuda.synchronize(dev)
                equal = bool(torch.equal(out_lib, out_host)) and bool(torch.isfinite(out_lib.float()).all())
                comp.bound, comp.arena = keep_bound, keep_arena     # the timed runs use the library path again
                comp.ctx_token = comp.ctx_bound = None
        finally:
            net.t_shard = None
            if saved_env is None:
                os.environ.pop("T2V_COLLECTIVES", None)
            else:
                os.environ["T2V_COLLECTIVES"] = saved_env
        # the two gathers AROUND the forward (eps pair per step, uint8 frames per video): library communicators against torch.distributed
        gathers_in_library = False
        for gc, n in ((topo.pair_comm, 2), (topo.world_comm, topo.world)):
            if equal and gc.communicator(dev) is not None:
                gathers_in_library = True
                mine = torch.full((4100,), float(topo.rank + 1), device=dev) + torch.arange(4100, device=dev)     # 16-byte units: window-eligible
                a, b = torch.empty(n * 4100, device=dev), torch.empty(n * 4100, device=dev)
                gc.all_gather_into(a, mine)
                all_gather_into(b, mine, group=gc.group)
                torch.cuda.synchronize(dev)
                equal = equal and bool(torch.equal(a, b))
        windows = [c.window for c in self._library_comms() if c.window]
        via = [c.counters() for c in self._library_comms()]
        ok = self._all_ok(equal)
        transport = (f"peer windows ({sum(v[0] for v in via)} exchanges as device-initiated stores, {sum(v[1] for v in via)} through RCCL)" if windows
                     else "RCCL on the launch stream")
        return {"ok": ok, "collective_ops_per_forward": n_coll,
                "compared": (why or (f"library communicator ({transport}) vs host executor (torch.distributed): bit-equal on every rank"
                                     if in_library else "exchanges already run through the host executor in this set-up (gloo group): nothing to compare")),
                "in_library": bool(in_library), "eps_and_frame_gathers_in_library": bool(gathers_in_library),
                "peer_windows": windows or None, "_windows": bool(windows)}

    @torch.no_grad()
    def __call__(self, cond, uncond, seed):
        pipe, topo = self.pipe, self.topo
        dev = pipe.device
        pipe.sd_model.t_shard = topo.tshard
        try:
            pipe.diffusion.get_sampler("DDIM_Gaussian", return_sampler=False)
            sampler = pipe.diffusion.sampler
            sampler.cfg_parallel = topo
            _, noise, _ = pipe.diffusion.get_noise(1, 4, self.frames_total, self.height, self.width, seed=seed)
            f0 = topo.spec.offset
            x_T = noise[:, :, f0:f0 + topo.spec.frames].contiguous()
            from .samplers import SamplerStepCallback, SharedNoise
            # eta > 0: every rank draws the per-step noise of the WHOLE clip from an identically seeded generator and keeps its frames
            sampler.shared_noise = SharedNoise(seed, self.frames_total, f0, dev) if self.eta != 0.0 else None
            x0 = sampler.sample(S=self.ddim_steps, conditioning=cond.to(dev), unconditional_conditioning=uncond.to(dev),
                                x_T=x_T, shape=tuple(x_T.shape), unconditional_guidance_scale=self.guidance, eta=self.eta,
                                callback=SamplerStepCallback("DDIM_Gaussian", self.ddim_steps, progress=False))
        finally:
            pipe.sd_model.t_shard = None
        a, b = topo.decode_share()
        order = topo.frame_order()
        nmax = max(n for _, _, n in order)
        H, W = 8 * (self.height // 8), 8 * (self.width // 8)        # what the decoder produces for a latent of H//8 x W//8
        send = torch.zeros((nmax, H, W, 3), dtype=torch.uint8, device=dev)
        if b > a:
            send[: b - a] = pipe.decode_frames(x0[:, :, a:b])
        out = torch.empty((topo.world * nmax, H, W, 3), dtype=torch.uint8, device=dev)
        topo.world_comm.all_gather_into(out, send)
        return torch.cat([out[r * nmax: r * nmax + n] for r, _, n in order], dim=0)


class _ReplicaRunner(_Runner):
    """Throughput layout: every GPU generates its own videos (cond + uncond batched as b=2, exactly the 1-GPU
    workload) — videos are independent objects, so there is no data-path collective at all."""

    def __init__(self, pipe, world, rank, frames, height, width, ddim_steps, guidance, videos: int = 1, eta: float = 0.0):
        super().__init__(pipe, CfgPair(1, 0), frames, height, width, ddim_steps, guidance, videos=videos, eta=eta)
        self.rank = rank
        self.frames_per_video_all_ranks = frames * world * videos
        self.describe = (f"{world * videos} independent videos in flight, {videos} per GPU (cond+uncond batched as "
                         f"b={2 * videos}, no data-path collective)")

    def __call__(self, cond, uncond, seed):
        return super().__call__(cond, uncond, seed + 1000 * self.rank)


def make_runner(pipe, world: int, rank: int, *, frames, height, width, ddim_steps, guidance, mode: str = "auto",
                videos: int = 1, eta: float = 0.0):
    """'replicas' (auto for world > 1): one video per GPU (throughput; no data-path collective — the reference's own
    data-parallel mode).  'tshard' (even world >= 4): ONE `frames`-frame video on 2 x R GPUs, T-sharded inside the UNet
    (statistics / halo / frame<->pixel exchanges before the temporal ops, executed by the library over RCCL) x the CFG pair.
    'pairs': one video per CFG pair — cond | uncond UNet forwards on 2 GPUs, one eps all-gather per step (world == 1: the
    single-GPU runner)."""
    if mode == "auto":
        mode = "pairs" if world == 1 else "replicas"
    if mode == "replicas":
        return _ReplicaRunner(pipe, world, rank, frames, height, width, ddim_steps, guidance, videos=videos, eta=eta)
    if mode == "tshard":
        assert videos == 1
        return _TShardRunner(pipe, TShardTopology(world, rank, frames), frames, height, width, ddim_steps, guidance, eta=eta)
    return _Runner(pipe, CfgPair(world, rank), frames, height, width, ddim_steps, guidance, videos=videos, eta=eta)
