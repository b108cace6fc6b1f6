"""Shared helpers for the parity tests (test infrastructure)."""
from __future__ import annotations

import torch

from interp import Interp
from sd_webui_text2video_amd import _lib as L
from sd_webui_text2video_amd.program import BoundProgram, Buf, Program

TD = {"f16": torch.float16, "f32": torch.float32}


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def fill(it: Interp, buf: Buf, gen: torch.Generator, scale: float = 1.0, total_cols=None):
    """Random-fill a buffer (all `ld` columns of its rows unless total_cols given) in the CPU arena."""
    v = it.mat(buf.ref, buf.rows, buf.cols, buf.ld, TD[buf.dtype], {})
    v.copy_((torch.randn(buf.rows, buf.cols, generator=gen) * scale).to(v.dtype))
    return v


def read(arena_or_interp, buf: Buf) -> torch.Tensor:
    it = arena_or_interp
    return it.mat(buf.ref, buf.rows, buf.cols, buf.ld, TD[buf.dtype], {}).clone()


def run_both(prog: Program, weights_cpu: dict, ext_cpu: dict, init):
    """Run `prog` in the CPU interpreter and on the GPU from identical initial arena contents.
    Returns (interp_after, gpu_arena_as_interp_view, ext_gpu_back_on_cpu)."""
    it = Interp(prog, weights_cpu, poison=False)
    init(it)
    arena0 = it.arena.clone()
    ext_ref = {k: v.clone() for k, v in ext_cpu.items()}
    it.run(ext_ref)

    dev = torch.device("cuda:0")
    arena_gpu = arena0.to(dev)
    w_gpu = {k: v.to(dev).contiguous() for k, v in weights_cpu.items()}
    ext_gpu = {k: v.to(dev).contiguous() for k, v in ext_cpu.items()}
    bound = BoundProgram(prog, arena_gpu.data_ptr(), {k: v.data_ptr() for k, v in w_gpu.items()})
    bound.run({k: v.data_ptr() for k, v in ext_gpu.items()}, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    got = Interp(prog, weights_cpu, poison=False)
    got.arena = arena_gpu.cpu()
    return it, got, ext_ref, {k: v.cpu() for k, v in ext_gpu.items()}


def run_lockstep(executors, exts, stream):
    """Single-process emulation of a T-shard group (tests on ONE GPU): the executors of all 'ranks' are
    stepped together; collectives are performed by direct copies between their arenas."""
    n = len(executors)
    steps = [ex.steps for ex in executors]
    assert len({len(s) for s in steps}) == 1
    for k in range(len(steps[0])):
        kind = steps[0][k][0]
        if kind == "seg":
            for r in range(n):
                steps[r][k][1].run(exts[r], stream)
            continue
        ops = [steps[r][k][1] for r in range(n)]
        nb = (ops[0].i[0] & 0xFFFFFFFF) | (ops[0].i[1] << 32)
        if ops[0].kind in (L.OP_ALLGATHER, L.OP_STATS_HALO):
            parts = []
            for r in range(n):
                assert (ops[r].i[2], ops[r].i[3]) == (n, r)
                off = ops[r].p[0].off
                parts.append(executors[r].arena[off + r * nb: off + (r + 1) * nb].clone())
            for r in range(n):
                off = ops[r].p[0].off
                for q in range(n):
                    executors[r].arena[off + q * nb: off + (q + 1) * nb] = parts[q]
        elif ops[0].kind == L.OP_ALLTOALL:
            base_cnt, last_cnt, direction = ops[0].i[4], ops[0].i[5], ops[0].i[6]
            cnt = lambda q: last_cnt if q == n - 1 else base_cnt
            moves = []
            for me in range(n):                                  # gather every (src -> dst) byte range first, then copy
                for q in range(n):
                    if q == me:
                        continue
                    if direction == 0:      # me sends cnt(me) chunks to q; q receives them at q.recv + me * base_cnt * nb
                        src = (me, ops[me].p[0].off + q * cnt(me) * nb, cnt(me) * nb)
                        dst = (q, ops[q].p[1].off + me * base_cnt * nb)
                    else:                   # me sends cnt(q) chunks to q; q receives them at q.recv + me * cnt(q) * nb
                        src = (me, ops[me].p[0].off + q * base_cnt * nb, cnt(q) * nb)
                        dst = (q, ops[q].p[1].off + me * cnt(q) * nb)
                    moves.append((executors[src[0]].arena[src[1]: src[1] + src[2]].clone(), dst))
            for data, (q, off) in moves:
                executors[q].arena[off: off + data.numel()] = data
        if ops[0].kind in (L.OP_HALO_EXCHANGE, L.OP_STATS_HALO):
            # frame 1 -> prev's frame F+1 ... (byte counts / neighbours read from the op records); STATS_HALO: the halo-padded RAW buffer p[1]
            merged = ops[0].kind == L.OP_STATS_HALO
            if merged:
                nb = (ops[0].i[4] & 0xFFFFFFFF) | (ops[0].i[5] << 32)
            geo = lambda op: (op.p[1].off, op.i[6], op.i[7], op.i[8]) if merged else (op.p[0].off, op.i[2], op.i[3], op.i[4])
            firsts, lasts = [], []
            for r in range(n):
                base, nf, _, _ = geo(ops[r])
                firsts.append(executors[r].arena[base + nb: base + 2 * nb].clone())
                lasts.append(executors[r].arena[base + nf * nb: base + (nf + 1) * nb].clone())
            for r in range(n):
                base, nf, prev, nxt = geo(ops[r])
                assert prev == (r - 1 if r > 0 else -1) and nxt == (r + 1 if r + 1 < n else -1)
                if r > 0:
                    executors[r].arena[base: base + nb] = lasts[r - 1]
                if r + 1 < n:
                    executors[r].arena[base + (nf + 1) * nb: base + (nf + 2) * nb] = firsts[r + 1]
