"""Denoise-program IR: the host lowers a network + input geometry into a flat list of op
records over one activation arena and a packed-weight store; libt2v_hip.so executes it.

Everything here is shape/pointer bookkeeping (no arithmetic on activations).  The record
layout mirrors `t2v_op` in include/t2v_hip.h field by field.
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from . import _lib as L

_ITEM = {"f16": 2, "f32": 4, "f64": 8, "u8": 1}
COLLECTIVE_KINDS = (L.OP_ALLGATHER, L.OP_HALO_EXCHANGE, L.OP_ALLTOALL, L.OP_STATS_HALO)   # executed over the plan's communicator (RCCL) or by a host executor
_DT = {"f16": L.F16, "f32": L.F32}


@dataclass(frozen=True)
class TShardSpec:
    """Frame-axis (T) split of one clip over `size` ranks: contiguous slices, `counts[i]` frames on slice i — all equal
    to ceil(F / size) except a shorter LAST one (125 frames / 4 -> 32, 32, 32, 29), so that gathered per-slice buffers
    padded to counts[0] frames keep the real frames contiguous from 0."""
    size: int
    index: int
    counts: Tuple[int, ...]

    @staticmethod
    def make(total_frames: int, size: int, index: int) -> "TShardSpec":
        base = -(-total_frames // size)
        last = total_frames - base * (size - 1)
        if last < 1:
            raise ValueError(f"{total_frames} frames cannot be split over {size} ranks in slices of {base}")
        return TShardSpec(size, index, tuple([base] * (size - 1) + [last]))

    @property
    def frames(self) -> int:
        return self.counts[self.index]

    @property
    def max_frames(self) -> int:
        return self.counts[0]

    @property
    def total(self) -> int:
        return sum(self.counts)

    @property
    def offset(self) -> int:
        return sum(self.counts[: self.index])


# ------------------------------------------------------------------------------------------
# symbolic pointers
# ------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Ref:
    space: str            # 'arena' | 'weight' | 'ext' | 'null'
    off: int = 0          # byte offset (arena / weight) or slot index (ext)
    name: str = ""        # weight tensor name

    def shifted(self, nbytes: int) -> "Ref":
        assert self.space in ("arena", "weight")
        return Ref(self.space, self.off + nbytes, self.name)


NULL = Ref("null")


@dataclass
class Buf:
    """A row-major 2-D view [rows, cols] with leading dimension `ld` (elements)."""
    ref: Ref
    rows: int
    cols: int
    ld: int
    dtype: str
    alloc_off: int = -1   # arena allocation this view belongs to (for free())
    owns: bool = True     # False: a borrowed window of another buffer — free() ignores it

    @property
    def item(self) -> int:
        return _ITEM[self.dtype]

    def col_slice(self, c0: int, c1: int) -> "Buf":
        return Buf(self.ref.shifted(c0 * self.item), self.rows, c1 - c0, self.ld, self.dtype, self.alloc_off)

    def borrow_cols(self, c0: int, c1: int) -> "Buf":
        """Non-owning column window (a producer writes half of a concat buffer in place)."""
        b = self.col_slice(c0, c1)
        b.owns = False
        return b

    def row_slice(self, r0: int, r1: int) -> "Buf":
        return Buf(self.ref.shifted(r0 * self.ld * self.item), r1 - r0, self.cols, self.ld, self.dtype, self.alloc_off)


class Arena:
    """First-fit allocator over one device buffer; 256-byte granularity; tracks the high-water mark."""

    ALIGN = 256

    def __init__(self):
        self.free_list: List[Tuple[int, int]] = []   # (offset, size), sorted, coalesced
        self.top = 0
        self.live: Dict[int, int] = {}
        self.high = 0

    def alloc(self, nbytes: int) -> int:
        n = (max(nbytes, 1) + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        for idx, (off, size) in enumerate(self.free_list):
            if size >= n:
                if size == n:
                    self.free_list.pop(idx)
                else:
                    self.free_list[idx] = (off + n, size - n)
                self.live[off] = n
                return off
        # grow: merge with a trailing free block if it touches the top
        if self.free_list and self.free_list[-1][0] + self.free_list[-1][1] == self.top:
            off, size = self.free_list.pop()
            self.top = off
        off = self.top
        self.top += n
        self.high = max(self.high, self.top)
        self.live[off] = n
        return off

    def free(self, off: int):
        n = self.live.pop(off)
        self.free_list.append((off, n))
        self.free_list.sort()
        merged: List[Tuple[int, int]] = []
        for o, s in self.free_list:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        self.free_list = merged


@dataclass
class Op:
    kind: int
    name: str
    i: List[int] = field(default_factory=lambda: [0] * L.OP_NI)
    f: List[float] = field(default_factory=lambda: [0.0] * L.OP_NF)
    p: List[Ref] = field(default_factory=lambda: [NULL] * L.OP_NP)
    flops: float = 0.0            # algorithmic 2*MAC (matmul/conv only), for the roofline
    meta: dict = field(default_factory=dict)
    out: Optional["Buf"] = None   # arena buffer this op writes (debug / determinism tooling)
    out2: Optional["Buf"] = None  # second output (the normalised tensor of a T2V_EPI_GN op)


class Program:
    def __init__(self, label: str = ""):
        self.label = label
        self.ops: List[Op] = []
        self.arena = Arena()
        self.taps: Dict[str, Buf] = {}          # debug: named live buffers (never freed when tapping)
        self.keep_taps = False
        self.target_cus = 256                   # MI355X compute units (tile / split-K policy)
        self.force_tile = None                  # tests: pin a tile id
        self.small_rank_tiles = False           # set by the lowering of a T-sharded program (choose_tile: 64x64 tiles for temporal convolutions)
        # GroupNorm as ONE launch when a statistics slice (rows x C/groups) is small (bytes; tools/gn_bench.py: the
        # per-frame instances of the 16x16 and lower levels and the cross-frame ones of the 4x4 level gain 8-19 us each,
        # larger slices on only 64 workgroups do not)
        self.gn_fused_slice_bytes = int(L.knob("T2V_GN_FUSED_SLICE", 64 * 1024))
        self.gn_fused_total_bytes = int(L.knob("T2V_GN_FUSED_TOTAL", 128 * 1024 * 1024))
        # precision option: weight Ref -> Ref of its low-order image (packing.WeightPacker.add_lo) or None; set by a lowering
        self.weight_lo = None
        # Device-side synchronisation words (never freed, zero from the bind-time fill, self-resetting): L.SYNC_INTS tickets of
        # the split-K fold (one per output tile) followed by the {arrivals, generation} pair of the cooperative GroupNorm's grid barrier
        # Allocated HERE, before any other buffer: an allocation made later could land on memory that an EARLIER op of the
        # program (whose buffer was already freed at lowering time) rewrites on every run — the words must only ever be
        # touched by the kernels that own them.
        self._sync: Buf = self.alloc(L.SYNC_INTS + L.SYNC_BARRIER_INTS, 1, "f32")
        # split-K fold in the last-arriving workgroup of a tile instead of a reduction launch: implemented and bit-identical, but
        # MEASURED SLOWER (same box, 83 split-K ops of a step: 5.24 vs 4.73 ms): only `tiles` workgroups fold, each behind a chain
        # of device-scope load latencies, where the reduction kernel uses the whole chip -> off by default
        self.splitk_tickets = L.knob("T2V_SPLITK_TICKETS", "0") != "0"
        self.gn_coop = os.environ.get("T2V_GN_COOP", "1") != "0"
        # GroupNorm inside the epilogue of the GEMM that produces its input (T2V_EPI_GN, round 5): a peephole of groupnorm() below.
        # It relies on the same co-residency as the single-pass kernel (one grid barrier per launch), so T2V_GN_COOP=0 turns it off too.
        self.gn_epilogue = self.gn_coop and os.environ.get("T2V_GN_EPI", "1") != "0" and not L.exchange_disabled()
        # its exchange scratch ([tiles_m][2][tiles_n][GN_PIECES] fp64 pairs, <= 512 tiles): ONE region for every fused op of the program
        # (launches are stream-ordered and each rewrites every word it reads), allocated here for the reason given above for the sync words
        self._gn_part: Optional[Buf] = self.alloc(L.GN_PART_BYTES, 1, "u8") if self.gn_epilogue else None
        # Buffers whose free() is postponed: operands of the GEMM emitted last.  If the next groupnorm() makes itself that GEMM's
        # epilogue, its output is written by the SAME launch that still reads these — it must not be allocated over them.
        self._deferred: List[Buf] = []

    # ---- memory ---------------------------------------------------------------------------
    def alloc(self, rows: int, cols: int, dtype: str, ld: Optional[int] = None) -> Buf:
        ld = cols if ld is None else ld
        off = self.arena.alloc(rows * ld * _ITEM[dtype])
        return Buf(Ref("arena", off), rows, cols, ld, dtype, off)

    def free(self, *bufs: Buf):
        for b in bufs:
            if b is None or b.alloc_off < 0 or not b.owns:
                continue
            if self.keep_taps and any(t.alloc_off == b.alloc_off for t in self.taps.values()):
                continue
            if self._is_operand_of_last_gemm(b):
                self._deferred.append(b)          # until the next op is emitted (or groupnorm() has decided)
                continue
            self.arena.free(b.alloc_off)

    def _is_operand_of_last_gemm(self, b: Buf) -> bool:
        if not getattr(self, "gn_epilogue", False) or not self.ops:
            return False
        op = self.ops[-1]
        if op.kind != L.OP_GEMM or op.i[16] != L.EPI_NONE or (op.meta.get("tile") not in self._GN_EPI_TILES and op.meta.get("tile") != 12) \
                or op.meta.get("split", 1) != 1:
            return False                                # (tile 12: _fuse_groupnorm may move the GEMM to the 128x128 tile)
        lo, hi = b.alloc_off, b.alloc_off + self.arena.live.get(b.alloc_off, 0)
        # every arena-space input of the launch: A, an arena-resident weight operand (V^T / q k^T GEMMs), bias, row bias, residual
        return any(r is not None and r.space == "arena" and lo <= r.off < hi for r in (op.p[0], op.p[1], op.p[2], op.p[3], op.p[4]))

    def _flush_deferred(self):
        pending, self._deferred = self._deferred, []
        for b in pending:
            if b.alloc_off in self.arena.live:
                self.arena.free(b.alloc_off)

    def sync_ref(self, which: str) -> Ref:
        return self._sync.ref if which == "tickets" else self._sync.ref.shifted(4 * L.SYNC_INTS)

    def tap(self, name: str, buf: Buf):
        if self.keep_taps:
            self.taps[name] = buf

    def _emit(self, op: Op) -> Op:
        self.ops.append(op)
        self._flush_deferred()       # (operands of the PREVIOUS op: it can no longer grow an epilogue)
        return op

    # ---- GEMM tiling policy ----------------------------------------------------------------
    def choose_tile(self, M: int, n: int, k: int, gather: int, allow_splitk: bool = True):
        """-> (tile id, split_k).  Tile ids as in t2v_op.i[22]: 0 = 128x128-class kernel (any N, the C8
        stem); 1 256x256, 2 256x320, 3 128x256, 4/5 128x128 with a 4-deep ring, 8 / 9 / 11 / 12 192x320 / 192x256 /
        128x320 / 64x64 (csrc/gemm2.hip); 6 / 7 / 13-24 exist only in -DT2V_G2_EXPERIMENTS builds (other main-loop
        schedules, csrc/gemm2_experiments.inc).
        Policy for a 256-CU chip: the widest tile whose grid still gives >= ~0.75 wave of
        workgroups; otherwise the 128x256 tile, then split-K over the (long) reduction."""
        cus = self.target_cus
        forced = self.force_tile is not None
        conv_like = gather in (L.GATHER_CONV3X3, L.GATHER_TCONV3)
        r6 = L.knob("T2V_TILE_R6", "1") != "0"          # round-6 additions for the rows of T-shard ranks / VideoCrafter (A/B switch)
        if forced:
            tile = self.force_tile
        elif gather == L.GATHER_CONV3X3_C8 or n < 256 or k < 64 or k % 64 != 0:
            tile = 0
        elif M >= 32768:                               # 32x32 level: big tiles, 320-wide when it divides
            tile = 2 if n % 320 == 0 and n != 960 else 1
            if r6 and M > 65536 and n == 2560 and k <= 320:
                tile = 1               # the GEGLU projection over many rounds, (256000 / 442368, 2560, 320): 256x256 608 / 608 vs 563 / 579 TF/s
            # 192x320 on 12 waves: M = 49152 gives exactly 256 / 768 workgroups for N = 320 / 960 instead of 192 / 768 tiles of
            # 256 rows on 256 CUs; measured (tools/gemm_sweep.py L0) +13 % / +6 % on the C -> C and QKV linears, +3-6 % on the
            # K = 960 .. 2880 convolutions, -7 % on the 8-wave-deep GEGLU GEMM (LDS traffic per MFMA is higher) -> only where
            # the 256-row grid is a few, badly filled waves
            if n % 320 == 0 and math.ceil(M / 256) * (n // 320) <= int(L.knob("T2V_TILE8_LIMIT", 640)) and L.knob("T2V_TILE8", "1") != "0" \
                    and not (r6 and n != 320 and k > 320 and M > 49152):
                # (round 6, SWEEP_FRAMES=125: at (64000, 640, 640 .. 17280) — several column tiles, several rounds of workgroups — 256x320 is
                #  ahead of the 12-wave / 8-wave tiles by 6-19 %: 411 / 778 / 991 / 720 / 1110 vs 389 / 661 / 848 / 649 / 934 TF/s)
                tile = 8
                # 128x320 on 8 waves (tile 11, round 4): VideoCrafter's M = 32768 makes 171 / 513 workgroups of 192 rows for N = 320 / 960
                # (0.67 of the last wave of CUs) and exactly 256 / 768 of 128 rows; M = 49152 never gets here (192 rows fill it)
                w8, w11 = math.ceil(M / 192) * (n // 320), math.ceil(M / 128) * (n // 320)
                fill = lambda wgs: wgs / (math.ceil(wgs / cus) * cus)
                if fill(w11) > fill(w8) + 0.05 and L.knob("T2V_TILE11", "1") != "0":
                    tile = 11
            elif r6 and n == 960 and k <= 320 and M > 65536 and L.knob("T2V_TILE8", "1") != "0":
                tile = 8               # (256000, 960, 320), the QKV projection of a 125-frame clip: 308 vs 275 TF/s on 256x256
            elif tile == 1 and n % 256 == 0 and L.knob("T2V_TILE8", "1") != "0":
                # the stem TemporalTransformer (inner = 512): 192 x 2 tiles of 256x256 = 384 workgroups, 192x256 gives 512
                w1, w9 = math.ceil(M / 256) * (n // 256), math.ceil(M / 192) * (n // 256)
                fill = lambda wgs: wgs / (math.ceil(wgs / cus) * cus)
                if w1 <= 1280 and fill(w9) > fill(w1) + 0.05:
                    tile = 9
        elif M >= 8192:                                # 16x16 level (b=2) / 32x32 level of a single CFG role (b=1)
            if M >= 16384 and n <= 1024 and L.knob("T2V_TILE8", "1") != "0":
                # one CFG role per GPU (pairs / T-shard layouts: M = 24576): 192x256 on 12 waves gives 128 x ceil(N / 256)
                # workgroups; measured (SWEEP_BATCH=1 tools/gemm_sweep.py L0) +7 % QKV, +9 % feed-forward, +13 % temporal conv
                tile = 9
                if r6 and n == 640 and M >= 24576:
                    tile = 8           # (24576, 640, 640 / 2560 / 5760 conv / 1920 tconv), 48 frames b = 2: 192x320 445 / 776 / 878 / 718 vs 354 / 601 / 682 / 579
                if n == 320 and L.knob("T2V_TILE11", "1") != "0":
                    # 128x320 on 8 waves: 192 workgroups, no padded columns (N = 320 on 256-wide tiles wastes 37 %); measured
                    # (SWEEP_BATCH=1 tools/gemm_sweep.py L0, round 4): ff2 497 vs 461 TF/s, conv3x3 681 vs 598, tconv 442 vs 419, C -> C
                    # 216 vs 213; N = 960 stays on 192x256 (324 vs 263)
                    tile = 11
            elif gather == L.GATHER_CONV3X3 and k >= 2560 and n % 320 == 0:
                tile = self._fill_choice(M, n, k) if allow_splitk else 2
            elif n >= 2560:
                tile = 2 if n % 320 == 0 else 1
                if r6 and tile == 2 and k <= 640 and L.knob("T2V_TILE8", "1") != "0":
                    # (12288, 2560, 320), the GEGLU projection of a 12-frame rank: 48 x 8 = 384 workgroups of 256x320 (1.5 rounds) against
                    # 512 of 192x320 (two full rounds): 518 vs 418 TF/s
                    fill = lambda wgs: wgs / (math.ceil(wgs / cus) * cus)
                    if fill(math.ceil(M / 192) * (n // 320)) > fill(math.ceil(M / 256) * (n // 320)) + 0.05:
                        tile = 8
            elif n >= 1536:
                tile = 9 if L.knob("T2V_TILE8", "1") != "0" else 3      # 16x16-level QKV (12288, 1920, 640): 492 vs 452 TF/s
                if r6 and M < 12288:
                    tile = 1                                            # (8192, 1920, 640): 256x256 463 vs 373 TF/s
            else:
                tile = 0               # (the 4-deep-ring 128x128 tile measured 640 vs 668 TF/s on the K = 2560 feed-forward GEMM)
                if r6 and n % 320 == 0 and math.ceil(M / 256) * (n // 320) >= 200:
                    # several rounds of 256x320 tiles (125 frames: M = 16000, 1024x576: M = 27648; N = 1280): C -> C 686 vs 491, ff2 917 vs
                    # 687, temporal conv 945 vs 726 TF/s
                    tile = 2
                elif r6 and M < 12288 and n == 640:
                    # M = 8192 (VideoCrafter's 16x16 level, a 32-frame T-shard rank): 128x256 tiles, 64 x 3 workgroups — C -> C 281 vs 269,
                    # temporal conv 508 vs 438, ff2 557 vs 516 TF/s; at M = 12288 (24 frames, b = 2) the 128x128 kernel stays ahead
                    tile = 3
        else:                                          # 8x8 / 4x4 levels: few rows, latency-bound
            if n >= 8192:
                tile = 1 if M >= 1024 else (0 if M >= 512 else 5)
                if r6 and M >= 1024 and n % 320 == 0:
                    # 256x320 where its grid fills the last round of workgroups better than 256x256's: (2048, 10240, 1280) 256 against 320
                    # workgroups, 862 vs 652 TF/s; (4000, ...) 884 vs 797; (1536, ...) 192 against 240: 256x256 825 vs 711; (3072, ...) 853 on 256x256
                    fill = lambda wgs: wgs / (math.ceil(wgs / cus) * cus)
                    w1, w2 = math.ceil(M / 256) * math.ceil(n / 256), math.ceil(M / 256) * (n // 320)
                    tile = 2 if fill(w2) > fill(w1) else 1
                elif r6 and 512 <= M < 1024:
                    tile = 3           # (512, 10240, 1280): 508 vs 465; (768, 10240, 1280): 667 vs 613
                elif r6 and 64 <= M <= 256:
                    tile = 12          # (96 / 192, 10240, 1280): 182 / 331 vs 159 / 289
            elif n >= 2560:
                tile = 1 if M >= 2048 else (3 if M >= 1024 else 5)
                if r6 and M == 2048:
                    tile = 3           # (2048, 3840, 1280): 571 vs 419 (256x256) / 469 (192x256)
                elif r6 and M <= 512:
                    tile = 12          # (512 / 384 / 96, 3840, 1280): 303 / 234 / 76 vs 257 / 204 / 59
                # 256-row tiles of M = 3072 x N = 3840 are 180 workgroups for 256 CUs; 192-row tiles give 240
                if tile == 1 and n % 256 == 0 and L.knob("T2V_TILE8", "1") != "0":
                    w1, w9 = math.ceil(M / 256) * (n // 256), math.ceil(M / 192) * (n // 256)
                    fill = lambda wgs: wgs / (math.ceil(wgs / cus) * cus)
                    if fill(w9) > fill(w1) + 0.05:
                        tile = 9
            elif gather == L.GATHER_CONV3X3 and k >= 8192:
                tile = 2 if (M >= 4096 and n % 320 == 0) else 3
                if r6 and ((3584 <= M < 4096) or (2048 <= M < 4096 and k >= 16384)):
                    tile = 1           # (4000, 1280, 11520): 256x256 x 3 splits 886 vs 729; (3072, 1280, 23040): x 4 splits 931 vs 865
            else:
                tile = 5                               # 128x128, 4-deep ring: 96 KiB per CU in flight
                if r6 and 6144 < M < 8192 and n == 1280 and gather in (L.GATHER_PLAIN, L.GATHER_TCONV3):
                    # 1024x576 clips, 8x8 level (M = 6912): ONE round of 128x320 (216 workgroups) / 192x256 (180) tiles — C -> C 516, ff2 775 on
                    # 128x320, temporal conv 718 on 192x256 vs 392 / 556 / 520 on 128x256 (270 workgroups: a second, nearly empty round)
                    tile = 11 if gather == L.GATHER_PLAIN else 9
                elif r6 and math.ceil(M / 128) * math.ceil(n / 128) >= 256 and n % 256 == 0:
                    tile = 3                           # (4000, 1280, 1280 / 3840 tconv / 5120): 128x256 411 / 595 / 647 vs 362 / 467 / 537 (M = 3072: 240 tiles, stays)
                if M <= 1024 and n <= 1280 and k <= 1280 and L.knob("T2V_TILE12", "1") != "0":
                    # the 4x4 level's C -> C linears (768, 1280, 1280): 64x64 tiles with the FULL reduction = 240 workgroups, no
                    # split-K slabs and no reduction launch: 201 vs 146 TF/s (tools/gemm_sweep.py L3, round 4); longer K / wider N
                    # stay on the split-K configurations (ff2 381 vs 341, conv3x3 525 vs 410, qkv 365 vs 331)
                    tile = 12
                elif r6 and gather == L.GATHER_PLAIN and k <= 320 and M >= 4096 and math.ceil(M / 128) * math.ceil(n / 128) > 150:
                    tile = 0           # (6144, 960, 320), the QKV projection of a 6-frame rank: 219 vs 173 TF/s
                elif r6 and gather == L.GATHER_PLAIN and n == 1920 and 4096 <= M < 8192:
                    tile = 9           # (6144, 1920, 640), one CFG role's 16x16-level QKV: 192x256 432 vs 335 TF/s
                elif r6 and k <= 2880 and math.ceil(M / 128) * math.ceil(n / 128) <= 150 and L.knob("T2V_TILE12", "1") != "0" and \
                        (gather in (L.GATHER_PLAIN, L.GATHER_CONV3X3) or (gather == L.GATHER_TCONV3 and self.small_rank_tiles)) and \
                        (gather != L.GATHER_CONV3X3 or k > 2560):
                    # round 6 (SWEEP_BATCH=1 SWEEP_FRAMES=6 tools/gemm_sweep.py: the rows of a 6-frame T-shard rank): where 128x128 tiles
                    # are <= 150 workgroups and the reduction is short, 64x64 tiles with the full reduction: (6144, 320, 320 .. 1280) 123 /
                    # 206 / 302 vs 108 / 179 / 266 TF/s, (1536, 640, 640 / 2560) 125 / 259 vs 94 / 237, temporal conv (6144, 320, 960) 250
                    # vs 211 and (1536, 640, 1920) 219 vs 167 — the latter only in a T-sharded program, where the cross-frame norm behind
                    # a temporal convolution is never that GEMM's epilogue (64x64 tiles have no GroupNorm epilogue); the 3x3 convolution
                    # (6144, 320, 2880) 378 vs 333
                    tile = 12
        if tile == 0:
            bm, bn, bk = 128, (64 if (n % 128 != 0 and n % 128 <= 64) else 128), 64
        else:
            bm, bn, bk = {1: (256, 256, 64), 2: (256, 320, 64), 3: (128, 256, 64), 4: (128, 128, 64), 5: (128, 128, 64),
                          6: (256, 256, 64), 7: (256, 320, 64), 8: (192, 320, 64), 9: (192, 256, 64), 11: (128, 320, 64), 12: (64, 64, 64),
                          13: (256, 256, 64), 14: (256, 320, 64), 15: (128, 256, 64), 16: (128, 128, 64), 17: (192, 320, 64),
                          18: (256, 256, 64), 19: (256, 320, 64), 20: (192, 320, 64), 21: (128, 256, 64),
                          22: (256, 256, 64), 23: (256, 320, 64), 24: (128, 256, 64)}[tile]
        tiles = math.ceil(M / bm) * math.ceil(n / bn)
        kt = math.ceil(k / bk)
        split = 1
        if allow_splitk and kt >= 16:
            # measured with tools/gemm_sweep.py on MI355X (256 CUs)
            if tile == 0:
                if kt >= 32 and tiles < cus:
                    split = max(1, min(round(2 * cus / tiles), kt // 16, 32))
                elif tiles < 0.5 * cus:
                    split = max(1, min(round(2 * cus / tiles), kt // 8, 32))
            elif tile == 12 and not forced:
                split = 1
            elif tile in (3, 4, 5) and not forced:
                split = max(1, min(cus // tiles, kt // 8, 8))
            elif tiles < 0.6 * cus:
                # one workgroup per CU: keep tiles*split within ONE wave of workgroups
                split = max(1, min(cus // tiles, kt // 8, 32))
        return tile, split

    def _fill_choice(self, M: int, n: int, k: int) -> int:
        """256x320 (tile 2) or 192x320 on 12 waves (tile 8) for a long-K convolution with N = 320 * j: whichever grid,
        after its split-K, fills the last wave of workgroups better (M = 12288, N = 640: 96 x 2 splits = 192 of 256 CUs
        against 128 x 2 = 256)."""
        if L.knob("T2V_TILE8", "1") == "0":
            return 2
        cus, kt = self.target_cus, math.ceil(k / 64)
        best, best_fill = 2, -1.0
        for tile, bm in ((2, 256), (8, 192)):
            tiles = math.ceil(M / bm) * (n // 320)
            split = max(1, min(cus // tiles, kt // 8, 32)) if (kt >= 16 and tiles < 0.6 * cus) else 1
            wgs = tiles * split
            fill = wgs / (math.ceil(wgs / cus) * cus)
            if fill > best_fill + 0.05:
                best, best_fill = tile, fill
        return best

    # ---- ops ------------------------------------------------------------------------------
    def begin(self):
        """Program prologue (kept for symmetry; nothing to initialise)."""

    def finish(self):
        self._flush_deferred()

    def gemm(self, name: str, a: Buf, w: Ref, n: int, k: int, out: Buf, *, bias: Ref = NULL,
             ldw: Optional[int] = None, gather: int = L.GATHER_PLAIN, conv: Optional[dict] = None,
             rowbias: Optional[Buf] = None, rows_per_batch: int = 0, residual: Optional[Buf] = None,
             epi: int = L.EPI_NONE, act: int = 0, bias_along_m: bool = False, m: Optional[int] = None,
             allow_splitk: bool = True, halo: bool = False, ln: Optional[tuple] = None, step_invariant: bool = False,
             a_lo: Optional[Buf] = None, out_lo: bool = False, k_alg: Optional[int] = None, stats: Optional[Buf] = None,
             a_wrap: int = 0, res_wrap: int = 0) -> Op:
        """out[M, n_out] = epi(gather(a)[M, k] @ w[n, k]^T).  conv['pad_after_only'] (3x3, stride 2): zero padding
        (0,1,0,1) instead of 1 on every side.
        a_wrap / res_wrap (plain gather, M <= 2 * wrap, no split-K): `a` / `residual` hold only `wrap` rows and row m >= wrap reads row
        m - wrap — the operands the cond | uncond pair of a guided step shares are computed once (unet.py `share_cfg_prefix`).
        stats (fp32 [ceil(M / 32), 2 n]): the epilogue also writes per 32-row strip the column sums / sums of squares of the stored
        result (T2V_EPI_STATS) for a GroupNorm that consumes it (`groupnorm(..., stats=)`); honoured when the op runs without split-K
        (op.meta["stats"] says whether it was).
        k_alg: the reduction length that counts as ALGORITHMIC work (Op.flops = 2 M n k_alg) when `k` repeats operand columns — the
        [hi | lo] x [W | W] forms of precise_operands run 2 k_alg deep but compute the reference's k_alg-deep product.
        out_lo (fp16 `out` that is the left half of a [M, 2n] buffer): also write the low-order image fp16(v - fp16(v)) at columns
        n .. 2n-1 of the same rows — the consumer GEMM runs on the [hi | lo] rows against [W | W] (precise_operands).
        ln = (gamma|beta Ref (fp32 [2n]), gamma Ref, beta Ref, ln_out Buf fp16, eps): LayerNorm of the fp32 result rows as a second
        output.  Fused into the GEMM epilogue when the op runs on the 192x320 tile with whole rows (n == 320, no split-K:
        the 32x32-level C -> C linears); otherwise a separate LayerNorm op follows."""
        conv = conv or {}
        M = out.rows if m is None else m
        if a_lo is not None:
            # hi + lo ACTIVATION split as TWO passes (the weights are exact fp16): t = A_lo.W (+ residual) in fp32, then the normal
            # GEMM on A_hi with t as its residual — the operand reaches the MFMA with ~22 bits instead of 11.  General form (any
            # gather); the lowerings use the ONE-pass forms instead where the layout allows it (rows [hi | lo] against [W | W]
            # for the 1x1 skip convolutions, lo in the padding channels for the stem: unet.py `precise_operands`)
            assert a_lo.dtype == "f16" and (a_lo.rows, a_lo.cols, a_lo.ld) == (a.rows, a.cols, a.ld) and epi == L.EPI_NONE and ln is None
            t_lo = self.alloc(M, n, "f32")
            self.gemm(name + ".a_lo", a_lo, w, n, k, t_lo, ldw=ldw, gather=gather, conv=conv, residual=residual, m=m,
                      allow_splitk=allow_splitk, halo=halo, step_invariant=step_invariant)
            op = self.gemm(name, a, w, n, k, out, bias=bias, ldw=ldw, gather=gather, conv=conv, rowbias=rowbias, rows_per_batch=rows_per_batch,
                           residual=t_lo, act=0, bias_along_m=bias_along_m, m=m, allow_splitk=allow_splitk, halo=halo,
                           step_invariant=step_invariant)
            assert act == 0
            self.free(t_lo)
            return op
        lo = self.weight_lo(w) if (self.weight_lo is not None and epi == L.EPI_NONE and not bias_along_m and w.space == "weight") else None
        lo_tmp = None
        if lo is not None:
            # hi + lo weight split: t = A.W_lo (+ residual) in fp32, then the normal GEMM on W_hi with t as its residual
            lo_tmp = self.alloc(M, n, "f32")
            self.gemm(name + ".w_lo", a, lo, n, k, lo_tmp, ldw=ldw, gather=gather, conv=conv, residual=residual, m=m,
                      allow_splitk=allow_splitk, halo=halo, step_invariant=step_invariant)
            residual = lo_tmp
        n_out = n // 2 if epi == L.EPI_GEGLU else n
        assert out.cols == n_out, (name, out.cols, n_out)
        assert a.dtype == "f16" and n % 4 == 0 and k % 8 == 0
        op = Op(L.OP_GEMM, name)
        I = op.i
        I[0], I[1], I[2] = M, n, k
        I[3], I[4], I[5] = a.ld, (k if ldw is None else ldw), out.ld
        I[6] = residual.ld if residual is not None else 0
        I[7] = gather
        if gather == L.GATHER_TCONV3:
            I[8], I[9] = conv["F"], conv["HW"]
        elif gather in (L.GATHER_CONV3X3, L.GATHER_CONV3X3_C8):
            I[8], I[9] = conv["Hin"], conv["Win"]
            I[11], I[12], I[13], I[14] = conv.get("stride", 1), conv.get("up", 0), conv["Hout"], conv["Wout"]
        I[10] = conv.get("Cin", 0)
        I[15] = rows_per_batch
        I[16], I[17], I[18] = epi, _DT[out.dtype], act
        I[20] = 1 if bias_along_m else 0
        I[21] = rowbias.ld if rowbias is not None else 0
        I[23] = 1 if (halo or conv.get("pad_after_only")) else 0
        if out_lo:
            assert gather == L.GATHER_PLAIN and out.dtype == "f16" and epi == L.EPI_NONE and ln is None and out.ld >= 2 * n
            I[11] = 1
        assert not (conv.get("pad_after_only") and (gather != L.GATHER_CONV3X3 or conv.get("up")))
        op.p[0], op.p[1], op.p[2] = a.ref, w, bias
        op.p[3] = rowbias.ref if rowbias is not None else NULL
        op.p[4] = residual.ref if residual is not None else NULL
        op.p[5] = out.ref
        if a_wrap or res_wrap:
            assert gather == L.GATHER_PLAIN and epi == L.EPI_NONE and a_lo is None
            assert (not a_wrap or (M <= 2 * a_wrap and a.rows >= a_wrap)) and (not res_wrap or (M <= 2 * res_wrap and residual is not None))
            I[12], I[13] = res_wrap, a_wrap
            allow_splitk = False
        if residual is not None:
            assert residual.dtype == "f32" and residual.rows >= (res_wrap or M) and residual.cols == n_out
        tile, split = self.choose_tile(M, n, k, gather, allow_splitk)
        # (one CFG role per GPU, M = 24576: forcing the 320-wide tile on the LayerNorm-producing Linears so that the norm fuses was
        #  measured: LayerNorm -0.36 ms, GEMMs +0.34 ms per step — not taken; the 192x256 tile stays there)
        I[19], I[22] = split, tile
        ln_fused = False
        if ln is not None:
            gb, gamma, beta, ln_out, ln_eps = ln
            assert out.dtype == "f32" and ln_out.dtype == "f16" and ln_out.cols == n and ln_out.rows >= M
            # (tile 2 = 256x320, round 6: the several-round grids of 125-frame / 1024x576 clips — a wave's two 32-row blocks one after the other)
            ln_fused = (tile in ((8, 11, 2) if L.knob("T2V_LN_TILE2", "1") != "0" else (8, 11)) and split == 1 and n == 320 and gather == L.GATHER_PLAIN and epi == L.EPI_NONE and act == 0
                        and rowbias is None and not bias_along_m and k % 64 == 0 and L.knob("T2V_LN_FUSE", "1") != "0")
            # ... or ACROSS the column tiles of the launch (round 5, t2v_epilogue_rows_lnx): the partial row sums meet at a grid barrier, so
            # the whole grid must be resident at once (the 16x16 / 8x8 / 4x4-level C -> C linears: 480 / 240 / 240 workgroups)
            ln_x = False
            if not ln_fused and self.gn_epilogue and tile in self._LNX_TILES and L.knob("T2V_LN_X", "1") != "0":
                bm, bn, per_cu = self._LNX_TILES[tile]
                tiles_m, tiles_n = -(-M // bm), -(-n // bn)
                ln_x = (split == 1 and gather == L.GATHER_PLAIN and epi == L.EPI_NONE and act == 0 and rowbias is None and not bias_along_m
                        and k % 64 == 0 and not out_lo and (tile != 0 or n % 128 == 0) and ln_out.ld % 4 == 0)
                if ln_x:
                    # rows are independent: a grid larger than the device holds (or the scratch has records for) runs as row chunks (round 6)
                    cap = min(per_cu * self.device_cus(), self._gn_part.rows // (bm * 16))
                    ln_x = tiles_n <= cap and (tiles_m * tiles_n <= cap or L.knob("T2V_GN_EPI_CHUNKS", "1") != "0")
            if ln_fused or ln_x:
                I[8], I[9] = (2 if ln_x else 1), ln_out.ld
                op.f[0] = ln_eps
                op.p[3], op.p[7] = gb, ln_out.ref
                if ln_x:
                    op.p[10], op.p[11] = self._gn_part.ref, self.sync_ref("barrier")
                    ln_fused = True
        with_stats = stats is not None and split == 1 and epi == L.EPI_NONE and not ln_fused and ln is None
        if with_stats:
            assert stats.dtype == "f32" and stats.rows >= -(-M // 32) and stats.ld == 2 * n
            I[16] = L.EPI_STATS
            op.p[7] = stats.ref
        ws = None
        if split > 1:
            ws = self.alloc(split * M, n, "f32")
            op.p[6] = ws.ref
            if self.splitk_tickets and epi == L.EPI_NONE and not ln_fused:
                op.p[7] = self.sync_ref("tickets")     # the last-arriving workgroup of a tile folds the slabs: no reduction launch
        op.flops = 2.0 * M * n * (k if k_alg is None else k_alg)
        op.out = out
        op.meta = dict(M=M, N=n, K=k, gather=gather, conv=dict(conv), epi=epi, split=split, tile=tile, halo=halo, ln=int(ln_fused),
                       stats=int(with_stats))
        if ln_fused:
            op.out2 = ln[3]
        if step_invariant:
            op.meta["step_invariant"] = True      # (also set on the .w_lo pass above: both halves of a split weight are skipped together)
        self._emit(op)
        if ln is not None and not ln_fused:
            self.layernorm(name + ".ln", out if m is None else out.row_slice(0, M), gamma, beta,
                           ln_out if m is None else ln_out.row_slice(0, M), ln_eps)
        if ws is not None:
            self.free(ws)     # stream order makes immediate reuse safe
        if lo_tmp is not None:
            self.free(lo_tmp)
        return op

    def to_q_cross_attention(self, name: str, a: Buf, w: Ref, out: Buf, *, k: int, heads: int, kbuf: Buf, vt: Buf, n_keys: int,
                             rows_per_sample: int, samples: int, scale: float, a_wrap: int = 0) -> Optional[Op]:
        """out[M, heads*64] = text cross-attention of q = a @ w^T — ONE launch (T2V_EPI_XATTN): the to_q accumulators become Q in LDS and
        the attention against the step-invariant K (`kbuf`: [samples * n_keys, heads*64] window, row stride kbuf.ld) / V^T (`vt`:
        [samples * N_total_rows..., lcp] with this site's heads*64 rows first, keys contiguous) runs in the epilogue.  Returns None where
        the tile policy picks a tile without the instantiation (the caller then emits the projection + attention pair)."""
        n = heads * 64
        M = out.rows
        assert a.dtype == "f16" and out.dtype == "f16" and out.cols == n and k % 64 == 0 and n_keys <= 96 and rows_per_sample % 32 == 0
        tile, split = self.choose_tile(M, n, k, L.GATHER_PLAIN, allow_splitk=False)
        if not (((tile in (8, 11)) and n == 320) or (tile in (0, 5) and n % 128 == 0)) or split != 1:
            return None
        op = Op(L.OP_GEMM, name)
        I = op.i
        I[0], I[1], I[2] = M, n, k
        I[3], I[4], I[5] = a.ld, k, out.ld
        I[7], I[13], I[15] = L.GATHER_PLAIN, a_wrap, rows_per_sample
        I[16], I[17], I[19], I[22] = L.EPI_XATTN, L.F16, 1, tile
        I[24], I[25], I[26], I[27], I[28] = kbuf.ld, n_keys, vt.ld, n_keys * kbuf.ld, vt.rows // samples * vt.ld
        assert I[27] < 2 ** 31 and I[28] < 2 ** 31 and vt.ld >= -(-n_keys // 32) * 32 and vt.rows % samples == 0
        op.f[1] = scale
        op.p[0], op.p[1], op.p[5], op.p[8], op.p[9] = a.ref, w, out.ref, kbuf.ref, vt.ref
        op.flops = 2.0 * M * n * k + 4.0 * M * n_keys * n
        op.out = out
        op.meta = dict(M=M, N=n, K=k, gather=L.GATHER_PLAIN, conv={}, epi=L.EPI_XATTN, split=1, tile=tile, halo=False, ln=0, heads=heads, n_keys=n_keys)
        return self._emit(op)

    @staticmethod
    def tattn_pixels_per_tile(frames: int) -> int:
        """Pixels whose `frames`-long sequences share one 192-row tile of the fused QKV + temporal attention GEMM."""
        return min(12, 192 // frames) if 2 <= frames <= 32 else 0

    def qkv_temporal_attention(self, name: str, a: Buf, w: Ref, out: Buf, *, samples: int, frames: int, hw: int, heads: int,
                               k: int, scale: float) -> Op:
        """out[tokens, heads*64] = temporal self-attention of the QKV projection of `a` — ONE launch (T2V_EPI_TATTN, tile 10):
        every 192-row tile is `pixels` pixels x `frames` frames of one sample, its 192 columns q | k | v of one head (`w` =
        [heads][q | k | v][k] head-major, packing.qkv_head_major), and the attention of those sequences runs in the epilogue.
        Token row of (sample s, frame f, pixel x) = (s * frames + f) * hw + x in `a` and `out`."""
        pix = self.tattn_pixels_per_tile(frames)
        assert pix >= 1 and a.dtype == "f16" and out.dtype == "f16" and out.cols == heads * 64 and k % 64 == 0 and a.cols == k
        assert a.rows == samples * frames * hw == out.rows
        tiles_ps = -(-hw // pix)
        op = Op(L.OP_GEMM, name)
        I = op.i
        I[0], I[1], I[2] = samples * tiles_ps * 192, 192 * heads, k
        I[3], I[4], I[5] = a.ld, k, out.ld
        I[7], I[8], I[9], I[10] = L.GATHER_PLAIN, frames, hw, pix
        I[16], I[17], I[19], I[22] = L.EPI_TATTN, L.F16, 1, 10
        op.f[1] = scale
        op.p[0], op.p[1], op.p[5] = a.ref, w, out.ref
        op.flops = 2.0 * a.rows * (192 * heads) * k + 4.0 * frames * frames * 64 * heads * samples * hw
        op.out = out
        op.meta = dict(M=a.rows, N=192 * heads, K=k, gather=L.GATHER_PLAIN, conv={}, epi=L.EPI_TATTN, split=1, tile=10, halo=False, ln=0,
                       frames=frames, hw=hw, heads=heads)
        return self._emit(op)

    # (rows, columns) of the tiles that have a T2V_EPI_GN instantiation, workgroups per CU they are resident with
    _GN_EPI_TILES = {8: (192, 320, 1), 11: (128, 320, 1), 3: (128, 256, 1), 5: (128, 128, 1), 0: (128, 128, 2)}
    # ... with a cross-tile LayerNorm instantiation (t2v_epilogue_rows_lnx)
    _LNX_TILES = {0: (128, 128, 2), 5: (128, 128, 1), 12: (64, 64, 2), 9: (192, 256, 1), 3: (128, 256, 1)}

    def gn_instance_too_large(self, M: int, n: int, k: int, gather: int, inst_rows: int) -> bool:
        """Would the GroupNorm that consumes this GEMM's [M, n] result (statistics instances of `inst_rows` rows) be refused as the GEMM's
        epilogue ONLY because one instance spans more row tiles than a co-resident launch may have?  (Then the producer-statistics route
        pays: unet.strips_for.)"""
        if not self.gn_epilogue:
            return False
        tile, split = self.choose_tile(M, n, k, gather, True)
        if split != 1 or tile not in self._GN_EPI_TILES:
            return False
        bm, bn, per_cu = self._GN_EPI_TILES[tile]
        tiles_n = -(-n // bn)
        cap = min(per_cu * self.device_cus(), self._gn_part.rows // (2 * L.GN_PIECES * 16))
        return -(-M // bm) * tiles_n > cap and math.lcm(bm, inst_rows) // bm * tiles_n > cap

    _DEVICE_CUS = None

    @classmethod
    def device_cus(cls) -> int:
        """Compute units of the current device (the co-residency bound of the T2V_EPI_GN launches); 256 (MI355X) when there is no
        device to ask (lowering tests on the CPU).  T2V_DEVICE_CUS overrides."""
        if cls._DEVICE_CUS is None:
            n = int(os.environ.get("T2V_DEVICE_CUS", "0"))
            if n <= 0:
                try:
                    import torch
                    n = L.device_info()[1] if torch.cuda.is_available() else 256
                except Exception:
                    n = 256
            Program._DEVICE_CUS = n
        return cls._DEVICE_CUS

    def _fuse_groupnorm(self, name: str, x: Buf, gb: Optional[Ref], out: Buf, *, n_inst: int, eps: float, silu: bool, groups: int, lo: bool,
                        x_dead: bool) -> Optional[Op]:
        """The peephole: if the LAST emitted op is the GEMM that produced `x` and it runs on a tile with a T2V_EPI_GN instantiation,
        without split-K, on a grid that is resident at once, the norm becomes that GEMM's epilogue (the op record is patched) and no
        GROUPNORM op is emitted.  x_dead: nothing but this norm reads `x` — the GEMM then does not store it at all."""
        if not self.gn_epilogue or gb is None or not self.ops:
            return None
        op = self.ops[-1]
        if op.kind != L.OP_GEMM or op.out is not x or op.i[16] != L.EPI_NONE or op.meta.get("ln"):
            return None
        tile, I, split = op.meta.get("tile"), op.i, op.meta.get("split", 1)
        M, N, K, gather = I[0], I[1], I[2], I[7]
        rows = x.rows // n_inst
        if gather == L.GATHER_CONV3X3_C8 or (gather == L.GATHER_CONV3X3 and I[12]) or K % 64 or I[20] or I[18] or (gather == L.GATHER_PLAIN and (I[8] or I[11] == 1)):
            return None
        if x.rows != M or x.cols != N or N % groups or M % rows:
            return None
        if x.dtype == "f16" and not x_dead:
            return None                                   # (an fp16 stream that someone else reads: keep the rounding where it was)
        if split > 1:
            # split-K: the norm runs in the REDUCTION's launch (norm.hip splitk_gn_kernel: 512 threads, a thread = rows x 8 channels in
            # registers) — a co-resident grid of n_inst x chunks workgroups must exist for some rows-per-thread count
            if L.knob("T2V_GN_EPI_SPLITK", "1") == "0" or op.p[7].space != "null" or N % 8 or N // 8 > 512 or groups > 32:
                return None
            rpass = 512 // (N // 8)
            chunks = [n_inst * -(-rows // (rpass * kr)) for kr in (1, 2, 4, 8, 12, 16, 20)]
            fit = [c for c in chunks if c <= self.device_cus()]
            if not fit or fit[0] * groups * 16 > self._gn_part.rows:
                return None
        else:
            retile = None
            if tile == 12 and self.force_tile is None:
                # 64x64 tiles (chosen for a small grid, choose_tile) have no GroupNorm epilogue; the 128x128 tile with the 4-deep ring —
                # what such a GEMM ran on before round 6 — has: the fused norm is worth more than the tile (a launch and two passes)
                tile = retile = 5
            if tile not in self._GN_EPI_TILES or L.knob(f"T2V_GN_EPI_TILE{tile}", "1") == "0":
                return None
            bm, bn, per_cu = self._GN_EPI_TILES[tile]
            if tile == 0 and N % 128 != 0:
                return None                               # (the narrow 128x64 form of the 128x128-class kernel has no instantiation)
            if N // groups > bn or rows % 32 or not (rows >= bm or 2 * rows == bm):
                return None                               # (a row tile may touch at most two statistics instances)
            tiles_m, tiles_n = -(-M // bm), -(-N // bn)
            # workgroups ONE launch may have: what the device holds at once (the exchange needs the grid resident) and what the exchange
            # scratch has records for
            cap = min(per_cu * self.device_cus(), self._gn_part.rows // (2 * L.GN_PIECES * 16))
            if tiles_m * tiles_n > cap:
                # round 6: the launcher cuts such a grid into row chunks of whole tiles AND whole instances (t2v_launch_coresident),
                # one co-resident launch each — possible when one such unit fits (per-frame norms of long / large clips; a cross-frame
                # instance of a 125-frame clip does not: that norm takes the producer-statistics route, `gn_instance_too_large`)
                if math.lcm(bm, rows) // bm * tiles_n > cap or L.knob("T2V_GN_EPI_CHUNKS", "1") == "0":
                    return None
            # the normalised tensor is written by the launch that reads the GEMM's operands: never over them (free() defers those)
            o_lo = out.ref.off
            o_hi = o_lo + ((out.rows - 1) * out.ld + out.cols * (2 if lo else 1)) * 2
            for r in (op.p[0], op.p[1], op.p[2], op.p[3], op.p[4]):
                if r is not None and r.space == "arena":
                    a_lo = max((off for off in self.arena.live if off <= r.off), default=None)
                    a_hi = a_lo + self.arena.live[a_lo] if a_lo is not None else None
                    if a_lo is None or not (o_hi <= a_lo or o_lo >= a_hi):
                        return None
        part = self._gn_part
        if split == 1 and retile is not None:
            I[22] = op.meta["tile"] = retile
        I[16] = L.EPI_GN
        I[24], I[25], I[26], I[27], I[28], I[29] = rows, out.ld, int(silu), int(lo), groups, int(x_dead)
        op.f[2] = eps
        op.p[8], op.p[9], op.p[10], op.p[11] = gb, out.ref, part.ref, self.sync_ref("barrier")
        op.meta["gn"] = dict(name=name, n_inst=n_inst, rows=rows, silu=int(silu), lo=int(lo), dead=int(x_dead))
        op.meta["epi"] = L.EPI_GN
        op.out2 = out
        return op

    def groupnorm(self, name: str, x: Buf, gamma: Ref, beta: Ref, out: Buf, *, n_inst: int, eps: float,
                  silu: bool, groups: int = 32, shard: Optional[TShardSpec] = None, lo: bool = False, stats: Optional[Buf] = None,
                  gb: Optional[Ref] = None, x_dead: bool = False, cast: Optional[Buf] = None, cast_lo: bool = False,
                  halo_raw: Optional[Buf] = None) -> Op:
        """GroupNorm(+SiLU).  gb (fp32 [2C] gamma | beta) + x produced by the op emitted last: the norm may become that GEMM's
        epilogue (`_fuse_groupnorm`; x_dead = nothing else reads x).
        cast (fp16 [x.rows, >= C (2C with cast_lo)]): second output — the RAW input as fp16 (+ its low-order image at column C): the operand
        of a 1x1 skip convolution that reads the same tensor (no separate cast pass); the norm then stays its own op.  With `shard` (cross-frame statistics of a T-sharded clip; n_inst = 1) the op is split
        into: statistics (this rank's partials) -> all-gather of the fp64 partials over the T group ->
        ordered fold of all parts + normalise; every rank ends up with bit-identical statistics.  Each rank folds its own
        block partials first, so a part is one {sum, sum of squares} pair per group: 512 bytes per instance, whatever the
        slice lengths (uneven slices need no special care).
        halo_raw (with `shard`; the norm in front of a T-sharded temporal convolution): `x` is the interior of this halo-padded RAW buffer
        [(frames + 2) * frame_rows, C] and `out` the interior of a halo-padded fp16 buffer of the same shape.  The statistics exchange then
        also carries the raw boundary frames to the two neighbours (ONE T2V_OP_STATS_HALO instead of an all-gather and, after the
        normalise, a halo exchange), and the apply pass normalises the received frames with the same statistics into out's halo slots.
        lo: `out` is the left half of a [rows, 2C] buffer; the low-order fp16 image of every output value goes to columns C .. 2C-1
        (hi + lo operand split of the consuming GEMM, precise_operands).
        stats: the T2V_EPI_STATS strips written by the GEMM that produced `x` (fp32 [x.rows / 32, 2 C]): the op folds them (phase 3)
        instead of reading the tensor for its statistics — two small launches, no grid barrier."""
        rows = x.rows // n_inst
        assert not lo or out.ld >= 2 * x.cols
        assert rows * n_inst == x.rows and out.dtype == "f16" and x.cols % 4 == 0
        if shard is None and stats is None and cast is None:
            fused = self._fuse_groupnorm(name, x, gb, out, n_inst=n_inst, eps=eps, silu=silu, groups=groups, lo=lo, x_dead=x_dead)
            self._flush_deferred()
            if fused is not None:
                return fused
        nparts, part = (shard.size, shard.index) if shard is not None else (1, 0)
        rows_total = rows * nparts
        if shard is not None:
            assert n_inst == 1 and rows % shard.frames == 0
            rows_total = rows // shard.frames * shard.total
        # rows reduced by one statistics workgroup: the smallest power of two >= 4 that keeps the grid within
        # ~4 workgroups per CU (each thread then has several rows in flight)
        rpb = 4
        while n_inst * ((rows + rpb - 1) // rpb) > 4 * self.target_cus:
            rpb *= 2
        nblk = (rows + rpb - 1) // rpb
        if shard is None:
            part_bytes = n_inst * nblk * groups * 16
            scratch = self.alloc(part_bytes + n_inst * groups * 8, 1, "u8")
        else:
            # T-sharded: [nparts gathered {sum, sumsq} per (instance, group)] [this rank's block partials] [finals]
            part_bytes = n_inst * groups * 16
            scratch = self.alloc(nparts * part_bytes + n_inst * nblk * groups * 16 + n_inst * groups * 8, 1, "u8")

        def make(phase, suffix):
            op = Op(L.OP_GROUPNORM, name + suffix)
            op.i[0:12] = [n_inst, rows, x.cols, x.ld, groups, _DT[x.dtype], int(silu), out.ld, phase, nparts, part, rpb]
            if shard is not None:
                op.i[14] = rows_total
            if lo and phase != 1:
                op.i[16] = 1
            if cast is not None and phase != 1:
                assert cast.dtype == "f16" and cast.rows == x.rows and cast.ld % 8 == 0 and cast.ld >= x.cols * (2 if cast_lo else 1) and x.cols % 8 == 0
                op.p[8], op.i[19], op.i[20] = cast.ref, cast.ld, (x.cols if cast_lo else 0)
            op.f[0] = eps
            op.p[0:5] = [x.ref, gamma, beta, out.ref, scratch.ref]
            return op

        if nparts == 1 and stats is not None:
            assert rows % 32 == 0 and stats.ld == 2 * x.cols and stats.rows * 32 >= x.rows
            op = make(3, "")
            op.i[17] = x.cols
            op.p[6] = stats.ref
            op.meta = dict(n_inst=n_inst, rows=rows, C=x.cols, dt=x.dtype, fused=0, coop=0, strips=1)
            op.out = out
            self._emit(op)
        elif nparts == 1:
            op = make(0, "")
            # small statistics slices (16x16 / 8x8 / 4x4 levels): one launch, one workgroup per (instance, group)
            cpg = x.cols // groups
            item = 2 if x.dtype == "f16" else 4
            if self.gn_fused_slice_bytes and x.cols % groups == 0 and cpg % 4 == 0 and rows * cpg * item <= self.gn_fused_slice_bytes \
                    and x.rows * x.cols * item <= self.gn_fused_total_bytes:
                op.i[12] = 1
            if self.gn_coop and not op.i[12] and groups <= 32:
                # single-pass cooperative kernel where the tensor fits the chip's registers at one workgroup per CU (decided
                # by the library from the device's CU count; otherwise it runs the three launches on the same scratch)
                op.i[15] = 1
                op.p[5] = self.sync_ref("barrier")
                if self._gn_part is not None:        # tagged-record exchange instead of the barrier: the program's record-only region
                    op.p[7], op.i[18] = self._gn_part.ref, self._gn_part.rows
            op.meta = dict(n_inst=n_inst, rows=rows, C=x.cols, dt=x.dtype, fused=int(op.i[12]), coop=int(op.i[15]))
            op.out = out
            self._emit(op)
        else:
            op1 = make(1, ".stats")
            if stats is not None:
                # round 6: this rank's part folded from the producing GEMM's T2V_EPI_STATS strips — no statistics pass over the tensor
                assert rows % 32 == 0 and stats.ld == 2 * x.cols and stats.rows * 32 >= x.rows
                op1.p[6], op1.i[17] = stats.ref, x.cols
            self._emit(op1)
            full = Buf(scratch.ref, nparts * part_bytes, 1, 1, "u8", scratch.alloc_off)
            if halo_raw is None:
                self.allgather(name + ".stats.allgather", full, part_bytes, shard)
                op = make(2, ".apply")
            else:
                frame_rows = rows // shard.frames
                assert halo_raw.rows == rows + 2 * frame_rows and halo_raw.dtype == x.dtype and halo_raw.ld == x.ld and not lo and cast is None
                assert x.ref.off == halo_raw.ref.off + frame_rows * x.ld * x.item and x.ref.space == halo_raw.ref.space
                self.stats_halo(name + ".stats_halo", full, part_bytes, halo_raw, frame_rows, shard.frames, shard)
                op = make(2, ".apply")
                op.i[21] = frame_rows if shard.index > 0 else 0                       # the clip's two ends keep the convolution's zero padding
                op.i[22] = frame_rows if shard.index + 1 < shard.size else 0
            op.out = out
            self._emit(op)
        self.free(scratch)      # stream order makes immediate reuse safe
        return op

    def allgather(self, name: str, full: Buf, part_bytes: int, shard: TShardSpec) -> Op:
        """In-place all-gather over the T group: `full` = shard.size parts of part_bytes bytes, this rank's part
        (index shard.index) already written by the preceding ops.  One T2V_OP_ALLGATHER record: RCCL on the launch
        stream inside the library, or torch.distributed in a host executor (gloo CPU tests)."""
        op = Op(L.OP_ALLGATHER, name)
        op.i[0], op.i[1], op.i[2], op.i[3] = part_bytes & 0xFFFFFFFF, part_bytes >> 32, shard.size, shard.index
        op.p[0] = full.ref
        op.meta = dict(type="allgather", full=full, part_bytes=part_bytes)
        return self._emit(op)

    def stats_halo(self, name: str, full: Buf, part_bytes: int, raw: Buf, frame_rows: int, frames: int, shard: TShardSpec) -> Op:
        """T2V_OP_STATS_HALO: the statistics parts of `full` to every rank of the T group and the first / last frame of the halo-padded
        RAW buffer `raw` to the previous / next slice's halo slot, in one grouped exchange (include/t2v_hip.h)."""
        op = Op(L.OP_STATS_HALO, name)
        fb = frame_rows * raw.ld * raw.item
        op.i[0], op.i[1], op.i[2], op.i[3] = part_bytes & 0xFFFFFFFF, part_bytes >> 32, shard.size, shard.index
        op.i[4], op.i[5], op.i[6] = fb & 0xFFFFFFFF, fb >> 32, frames
        op.i[7] = shard.index - 1 if shard.index > 0 else -1
        op.i[8] = shard.index + 1 if shard.index + 1 < shard.size else -1
        op.p[0], op.p[1] = full.ref, raw.ref
        op.meta = dict(type="stats_halo", full=full, part_bytes=part_bytes, buf=raw, frame_rows=frame_rows, frames=frames)
        return self._emit(op)

    def halo_exchange(self, name: str, buf: Buf, frame_rows: int, frames: int, shard: TShardSpec) -> Op:
        """+-1 frame neighbour exchange: `buf` = halo-padded token buffer [(frames + 2) * frame_rows, C] whose interior
        is filled; the first / last real frame goes to the previous / next slice's halo slot (T2V_OP_HALO_EXCHANGE)."""
        op = Op(L.OP_HALO_EXCHANGE, name)
        fb = frame_rows * buf.ld * buf.item
        op.i[0], op.i[1], op.i[2] = fb & 0xFFFFFFFF, fb >> 32, frames
        op.i[3] = shard.index - 1 if shard.index > 0 else -1
        op.i[4] = shard.index + 1 if shard.index + 1 < shard.size else -1
        op.p[0] = buf.ref
        op.meta = dict(type="halo", buf=buf, frame_rows=frame_rows, frames=frames)
        return self._emit(op)

    def reshard_rows(self, name: str, src: Buf, dst: Buf, *, rows: int, chunk: int, s_src: int, s_dst: int,
                     residual: Optional[Buf] = None) -> Op:
        """Row r: src row (r // chunk) * s_src + r % chunk -> dst row (r // chunk) * s_dst + r % chunk (+ residual at the dst
        row).  `src` / `dst` are views whose first row is row 0 of the mapping (pass row_slice()s for offsets)."""
        assert src.dtype == dst.dtype and src.cols == dst.cols and src.cols % (8 if src.dtype == "f16" else 4) == 0
        op = Op(L.OP_RESHARD_ROWS, name)
        op.i[0:8] = [rows, src.cols, chunk, s_src, s_dst, src.ld, dst.ld, _DT[src.dtype]]
        op.p[0], op.p[1] = src.ref, dst.ref
        if residual is not None:
            assert residual.dtype == "f32" and src.dtype == "f32"
            op.i[8] = residual.ld
            op.p[2] = residual.ref
        op.out = dst
        return self._emit(op)

    def reshard_parts(self, name: str, src: Buf, dst: Buf, *, parts: int, rows: int, chunk: int, s_src: int, s_dst: int,
                      part_rows_src: int, part_rows_dst: int, own: int, own_other: Buf, own_is_src: bool,
                      residual: Optional[Buf] = None) -> Op:
        """`parts` regroupings of reshard_rows' shape in ONE launch (RESHARD_ROWS, ABI 8 multi-part form): part q reads the view `src`
        shifted by q * part_rows_src rows and writes `dst` shifted by q * part_rows_dst rows (the residual moves with the destination's
        rows when its part stride equals the destination's: see below), except part `own`, whose source (own_is_src) or destination is
        `own_other` — the rank's own share of an all-to-all never goes through the staging buffer."""
        assert src.dtype == dst.dtype == own_other.dtype and src.cols == dst.cols == own_other.cols
        assert src.cols % (8 if src.dtype == "f16" else 4) == 0 and 0 <= own < parts
        assert own_other.ld == (src.ld if own_is_src else dst.ld)
        op = Op(L.OP_RESHARD_ROWS, name)
        op.i[0:8] = [rows, src.cols, chunk, s_src, s_dst, src.ld, dst.ld, _DT[src.dtype]]
        op.i[9:15] = [parts, part_rows_src * src.ld, part_rows_dst * dst.ld, 0, own, 1 if own_is_src else 0]
        op.p[0], op.p[1], op.p[3] = src.ref, dst.ref, own_other.ref
        if residual is not None:
            assert residual.dtype == "f32" and src.dtype == "f32"
            op.i[8], op.i[12] = residual.ld, part_rows_dst * residual.ld
            op.p[2] = residual.ref
        op.out = dst
        return self._emit(op)

    def alltoall(self, name: str, send: Buf, recv: Buf, chunk_bytes: int, shard: TShardSpec, direction: int) -> Op:
        """Frame <-> pixel resharding over the T group (T2V_OP_ALLTOALL; layouts in include/t2v_hip.h).  direction 0:
        frames -> pixels, 1: pixels -> frames.  The rank's own part is moved by reshard_rows ops."""
        op = Op(L.OP_ALLTOALL, name)
        op.i[0], op.i[1] = chunk_bytes & 0xFFFFFFFF, chunk_bytes >> 32
        op.i[2], op.i[3], op.i[4], op.i[5], op.i[6] = shard.size, shard.index, shard.counts[0], shard.counts[-1], direction
        op.p[0], op.p[1] = send.ref, recv.ref
        op.meta = dict(type="alltoall")
        return self._emit(op)

    def memset(self, name: str, buf: Buf) -> Op:
        op = Op(L.OP_MEMSET, name)
        nbytes = ((buf.rows - 1) * buf.ld + buf.cols) * buf.item
        op.i[0], op.i[1] = nbytes & 0xFFFFFFFF, nbytes >> 32
        op.p[0] = buf.ref
        return self._emit(op)

    def layernorm(self, name: str, x: Buf, gamma: Ref, beta: Ref, out: Buf, eps: float = 1e-5) -> Op:
        assert x.dtype == "f32" and out.dtype == "f16"
        op = Op(L.OP_LAYERNORM, name)
        op.i[0:4] = [x.rows, x.cols, x.ld, out.ld]
        op.i[4] = int(L.knob("T2V_LN_CAP", 0))      # workgroup cap of the grid-stride kernel (0 = library default; A/B knob)
        op.f[0] = eps
        op.p[0:4] = [x.ref, gamma, beta, out.ref]
        op.out = out
        return self._emit(op)

    def attention(self, name: str, q: Ref, k: Ref, v: Ref, o: Ref, *, out_buf: Optional[Buf] = None, nq: int, nk: int, heads: int,
                  b_outer: int, b_inner: int, q_strides, kv_strides, o_strides, scale: float, head_dim: int = 64,
                  rel_k: Optional[Ref] = None, rel_v: Optional[Ref] = None, max_rel: int = 0, causal: bool = False,
                  q_offset: int = 0, relpos_mfma: Optional[int] = None, lo_off: int = 0,
                  rel_k16: Optional[Ref] = None, rel_vT16: Optional[Ref] = None, vt_scratch: Optional[Buf] = None, waves: int = 0) -> Op:
        """softmax(q k^T scale) v over strided (sequence, outer, inner) batches.  With rel_k / rel_v (fp32
        [2*max_rel+1, head_dim] tables) the LVDM relative-position temporal attention op is emitted instead.
        lo_off (elements): also store the low-order fp16 image of every output value at o + lo_off (rows [hi | lo] for a K-doubled
        output projection, precise_operands)."""
        assert head_dim in (40, 64, 80, 160) or rel_k is not None
        op = Op(L.OP_ATTENTION if rel_k is None else L.OP_RELPOS_ATTN, name)
        op.i[0:5] = [nq, nk, heads, b_outer, b_inner]
        op.i[14] = head_dim
        if rel_k is not None:
            assert nk <= 32 and 0 <= q_offset and q_offset + nq <= nk and head_dim % 8 == 0
            op.i[15], op.i[16] = max_rel, q_offset
            # i[17]: 0 the VALU kernel; 1 the round-3 MFMA kernel (measured slower, opt-in); 2 the persistent MFMA kernel for whole
            # clips of <= 16 frames on tables packed for it (rel_k16 / rel_vT16: packing.relpos_tables16) — the default where it applies
            sel = int(L.knob("T2V_RELPOS_MFMA", "2")) if relpos_mfma is None else int(relpos_mfma)
            fits16 = (rel_k16 is not None and rel_vT16 is not None and nq == nk and q_offset == 0 and nk <= 16 and max_rel >= nk - 1
                      and head_dim in (40, 64, 80, 160))
            if sel == 2 and not fits16:
                sel = 0
            op.i[17] = sel
            op.p[4], op.p[5] = rel_k, rel_v
            if sel == 2:
                op.p[6], op.p[7] = rel_k16, rel_vT16
            op.i[18] = lo_off
            assert not causal
        elif causal:
            assert nq == nk
            op.i[15] = 1
        if rel_k is None:
            op.i[16] = lo_off
            if vt_scratch is not None:
                # round 6 (csrc/attention.hip attn2_kernel): V transposed once per launch into this scratch — fp16 [b_outer * b_inner * heads * 64,
                # keys padded to a multiple of 64] — and the K / V^T tiles staged by LDS-DMA; head_dim 64, not causal
                n_pad = -(-nk // 64) * 64
                assert head_dim == 64 and not causal and vt_scratch.dtype == "f16" and vt_scratch.ld == n_pad
                assert vt_scratch.rows >= b_outer * b_inner * heads * 64 and waves in (0, 4, 8)
                op.p[6], op.i[17], op.i[18] = vt_scratch.ref, n_pad, waves
        op.i[5:8] = list(q_strides)
        op.i[8:11] = list(kv_strides)
        op.i[11:14] = list(o_strides)
        for s in list(q_strides) + list(kv_strides) + list(o_strides):
            assert 0 <= s < 2 ** 31
        op.f[0] = scale
        op.p[0:4] = [q, k, v, o]
        op.flops = 4.0 * nq * nk * head_dim * heads * b_outer * b_inner
        op.out = out_buf
        return self._emit(op)

    def softmax(self, name: str, x: Buf, out: Buf, scale: float) -> Op:
        op = Op(L.OP_SOFTMAX, name)
        op.i[0:4] = [x.rows, x.cols, x.ld, out.ld]
        op.f[0] = scale
        op.p[0:2] = [x.ref, out.ref]
        op.out = out
        return self._emit(op)

    def ncthw_to_cl(self, name: str, src: Ref, src_dtype: str, out: Buf, *, B, C, F, HW, scale=1.0, src_batch: int = 0,
                    lo: Optional[Buf] = None, lo_in_pad: bool = False) -> Op:
        """src_batch (< B): the source holds that many samples, output sample b reads sample b % src_batch.
        lo: second fp16 output of the same layout holding fp16(v - fp16(v)) (hi + lo operand split of the consumer)."""
        assert src_batch == 0 or B % src_batch == 0
        op = Op(L.OP_NCTHW_TO_CL, name)
        op.i[0:7] = [B, C, F, HW, out.ld, _DT[src_dtype], src_batch]
        op.f[0] = scale
        op.p[0:2] = [src, out.ref]
        if lo is not None:
            assert (lo.rows, lo.cols, lo.ld, lo.dtype) == (out.rows, out.cols, out.ld, "f16")
            op.p[2] = lo.ref
        if lo_in_pad:             # low-order images in the padding channels C .. 2C-1 of the same rows
            assert out.ld >= 2 * C
            op.i[7] = 1
        op.out = out
        return self._emit(op)

    def cl_to_ncthw(self, name: str, x: Buf, dst: Ref, dst_dtype: str, *, B, C, F, HW) -> Op:
        op = Op(L.OP_CL_TO_NCTHW, name)
        op.i[0:6] = [B, C, F, HW, x.ld, _DT[dst_dtype]]
        op.p[0:2] = [x.ref, dst]
        return self._emit(op)

    def to_uint8(self, name: str, src: Ref, src_dtype: str, dst: Ref, *, NI: int, C: int, F: int, H: int, W: int, strides,
                 half: bool, bgr: bool = False) -> Op:
        """tensor2vid on device: element (i, c, f, y, x) of the float video at i*si + c*sc + f*sf + y*sy + x*sx ->
        uint8 out[f, y, i*W + x, c]; `half` = the reference's fp16 arithmetic (half-precision VAE)."""
        si, sc, sf, sy, sx = strides
        op = Op(L.OP_TO_UINT8, name)
        op.i[0:8] = [NI, C, F, H, W, _DT[src_dtype], int(half), int(bgr)]
        op.i[8], op.i[9], op.i[10], op.i[11], op.i[12], op.i[13], op.i[14] = si & 0xFFFFFFFF, si >> 32, sc, sf & 0xFFFFFFFF, sf >> 32, sy, sx
        for v in (sc, sy, sx):
            assert 0 <= v < 2 ** 31
        op.p[0:2] = [src, dst]
        return self._emit(op)

    def time_embed(self, name: str, t: Ref, freqs: Ref, out: Buf) -> Op:
        op = Op(L.OP_TIME_EMBED, name)
        op.i[0:2] = [out.rows, out.cols]
        op.p[0:3] = [t, freqs, out.ref]
        op.out = out
        return self._emit(op)

    def copy2d(self, name: str, src: Buf, dst: Buf, act: int = 0, lo: Optional[Buf] = None) -> Op:
        """lo (fp32 -> fp16 casts): second fp16 output, same shape / leading dimension, = fp16(v - fp16(v))."""
        assert src.rows == dst.rows and src.cols == dst.cols and src.cols % 4 == 0
        op = Op(L.OP_COPY2D, name)
        op.i[0:7] = [src.rows, src.cols, src.ld, dst.ld, _DT[src.dtype], _DT[dst.dtype], act]
        op.p[0:2] = [src.ref, dst.ref]
        if lo is not None:
            assert src.dtype == "f32" and dst.dtype == "f16" and (lo.rows, lo.cols, lo.ld, lo.dtype) == (dst.rows, dst.cols, dst.ld, "f16")
            op.p[2] = lo.ref
        op.out = dst
        return self._emit(op)

    def embed_rows(self, name: str, ids: Ref, table: Ref, table_dtype: str, pos: Ref, out: Buf, *, L_pos: int, vocab: int) -> Op:
        """out[r,:] = table[ids[r],:] + pos[r % L_pos,:]  (fp32 out; ids int32)."""
        assert out.dtype == "f32" and out.ld == out.cols
        op = Op(L.OP_EMBED_ROWS, name)
        op.i[0:5] = [out.rows, out.cols, L_pos, vocab, _DT[table_dtype]]
        op.p[0:4] = [ids, table, pos, out.ref]
        op.out = out
        return self._emit(op)

    def ddim_step(self, name: str, *, C: int, inner: int, guided: int, eps_dtype: str, x_dtype: str, mode: int = 0,
                  samples: int = 1) -> Op:
        """C channels per video, `samples` videos per batch (x [samples, C, inner], eps [2, samples, C, inner])."""
        op = Op(L.OP_DDIM_STEP, name)
        op.i[0:7] = [C * samples, inner, guided, _DT[eps_dtype], _DT[x_dtype], mode, C]
        op.p[0:4] = [Ref("ext", L.EXT_XT), Ref("ext", L.EXT_EPS), Ref("ext", L.EXT_NOISE), Ref("ext", L.EXT_XT_OUT)]
        return self._emit(op)

    # ---- stats ----------------------------------------------------------------------------
    def total_flops(self) -> float:
        return sum(op.flops for op in self.ops)


# ------------------------------------------------------------------------------------------
# binding to device memory + execution through the C ABI
# ------------------------------------------------------------------------------------------
class BoundProgram:
    """A Program whose symbolic pointers are resolved against a device arena and weight
    tensors, compiled into a `t2v_plan`."""

    def __init__(self, prog: Program, arena_ptr: int, weight_ptrs: Dict[str, int], ops: Optional[List[Op]] = None, comm=None,
                 reset_sync: Optional[bool] = None, stream: int = 0):
        """`comm`: a parallel.Communicator (t2v_comm over RCCL) — required to RUN a program that holds collective ops.
        `reset_sync` (default: when the WHOLE program is bound, i.e. `ops` is None): zero the program's device-side sync words
        (split-K tickets, GroupNorm grid-barrier counters) in this arena on `stream` — the kernels' contract is "all zero before
        the first launch", and the arena may be recycled or randomly filled memory (tools/, benchmarks).  Never done for a
        sub-list of ops: those are bound while other plans of the same program may be in flight."""
        self.prog = prog
        lib = L.load()
        if arena_ptr and ((ops is None) if reset_sync is None else reset_sync):
            L.check(lib.t2v_sync_reset(ctypes.c_void_p(arena_ptr + prog._sync.ref.off), ctypes.c_void_p(stream)))
        ops = prog.ops if ops is None else ops
        self.ops = ops
        self.comm = comm
        n = len(ops)
        arr = (L.T2VOp * n)()
        for idx, op in enumerate(ops):
            r = arr[idx]
            r.kind, r.tag = op.kind, idx
            for j, v in enumerate(op.i):
                v = int(v)
                r.i[j] = v - (1 << 32) if v >= (1 << 31) else v      # low words of 64-bit sizes are stored as raw bits
            for j, v in enumerate(op.f):
                r.f[j] = float(v)
            for j, ref in enumerate(op.p):
                if ref.space == "null":
                    r.p[j] = 0
                elif ref.space == "arena":
                    r.p[j] = arena_ptr + ref.off
                elif ref.space == "weight":
                    r.p[j] = weight_ptrs[ref.name] + ref.off
                elif ref.space == "ext":
                    r.p[j] = ref.off
                else:
                    raise ValueError(ref.space)
        self._arr = arr
        handle = ctypes.c_void_p()
        L.check(lib.t2v_plan_create(arr, n, ctypes.byref(handle)))
        self.handle = handle
        self._skip_handle = None
        self._lib = lib
        if comm is not None:
            L.check(lib.t2v_plan_set_comm(handle, comm.handle))

    def run(self, ext: Dict[int, int], stream: int, skip_invariant: bool = False):
        """skip_invariant: leave out the program's step-invariant ops (Op.meta['step_invariant'] — the text-context K/V
        projection): their results are still in the arena from the previous run of this plan with the same context."""
        e = (ctypes.c_uint64 * L.EXT_SLOTS)()
        for k, v in ext.items():
            e[k] = v
        handle = self.handle
        if skip_invariant:
            if self._skip_handle is None:
                keep = [k for k, op in enumerate(self.ops) if not op.meta.get("step_invariant")]
                arr = (L.T2VOp * len(keep))(*[self._arr[k] for k in keep])
                h = ctypes.c_void_p()
                L.check(self._lib.t2v_plan_create(arr, len(keep), ctypes.byref(h)))
                if self.comm is not None:
                    L.check(self._lib.t2v_plan_set_comm(h, self.comm.handle))
                self._skip_handle, self._skip_arr = h, arr
            handle = self._skip_handle
        L.check(self._lib.t2v_plan_run(handle, e, L.EXT_SLOTS, ctypes.c_void_p(stream)))

    def run_timed(self, ext: Dict[int, int], stream: int) -> List[float]:
        e = (ctypes.c_uint64 * L.EXT_SLOTS)()
        for k, v in ext.items():
            e[k] = v
        ms = (ctypes.c_float * len(self.ops))()
        L.check(self._lib.t2v_plan_run_timed(self.handle, e, L.EXT_SLOTS, ctypes.c_void_p(stream), ms))
        return list(ms)

    def __del__(self):
        try:
            if self.handle:
                self._lib.t2v_plan_destroy(self.handle)
                self.handle = None
            if self._skip_handle:
                self._lib.t2v_plan_destroy(self._skip_handle)
                self._skip_handle = None
        except Exception:
            pass
