"""TEST INFRASTRUCTURE — a CPU interpreter for denoise programs (sd_webui_text2video_amd.program).

It executes the SAME op list, arena offsets and packed weights that libt2v_hip.so executes
on the GPU, with plain torch on the CPU, so the host logic — lowering, arena liveness,
weight packing, stride bookkeeping — can be validated against the oracle without a GPU
(`pytest -m "not gpu"`).  It is NOT a fallback: nothing under sd-webui-text2video_amd/
imports it, and the product path fails loudly without the HIP library.

Storage precision is emulated (fp16 buffers hold fp16-rounded values, fp32 accumulate), so
the interpreter also predicts the quantisation error of the device path.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from sd_webui_text2video_amd import _lib as L
from sd_webui_text2video_amd.program import Program, Ref

_TD = {L.F16: torch.float16, L.F32: torch.float32}


class Interp:
    def __init__(self, prog: Program, weights: Dict[str, torch.Tensor], poison: bool = True):
        self.prog = prog
        self.weights = {k: v.cpu() for k, v in weights.items()}
        self.arena = torch.zeros(prog.arena.high + 256, dtype=torch.uint8)
        if poison:      # NaN-poison so that reads of never-written memory are caught
            self.arena.view(torch.float16)[:] = float("nan")

    # ---- memory views -------------------------------------------------------------------------
    def _flat(self, ref: Ref, dtype, ext):
        item = torch.empty((), dtype=dtype).element_size()
        if ref.space == "arena":
            assert ref.off % item == 0
            return self.arena.view(dtype), ref.off // item
        if ref.space == "weight":
            w = self.weights[ref.name]
            assert w.dtype == dtype, (ref.name, w.dtype, dtype)
            assert ref.off % item == 0
            return w.reshape(-1), ref.off // item
        if ref.space == "ext":
            t = ext[ref.off]
            assert t.dtype == dtype, (t.dtype, dtype)
            return t.reshape(-1), 0
        raise ValueError(ref.space)

    def view(self, ref: Ref, shape, strides, dtype, ext):
        flat, off = self._flat(ref, dtype, ext)
        return torch.as_strided(flat, tuple(shape), tuple(strides), flat.storage_offset() + off)   # (ext tensors may be views)

    def mat(self, ref: Ref, rows, cols, ld, dtype, ext):
        return self.view(ref, (rows, cols), (ld, 1), dtype, ext)

    def _st(self, view, value, dtype):
        """Store `value` into an output view of logical type `dtype` (fp16 buffers hold fp16-rounded values).  One hook for
        every fp16 write, so that tests/precision_probe.py can switch the rounding of chosen tensors off."""
        view.copy_(value.to(dtype))

    # ---- execution -----------------------------------------------------------------------------
    def run(self, ext: Dict[int, torch.Tensor], ops=None):
        for op in (self.prog.ops if ops is None else ops):
            getattr(self, f"_op{op.kind}")(op, ext)

    # GEMM ------------------------------------------------------------------------------------------
    def _op1_tattn(self, op, ext):
        """T2V_EPI_TATTN: QKV projection (fp32 accumulate, fp16 q / k / v) + softmax(q k^T scale) v per (sample, pixel, head)."""
        I = op.i
        N, K, lda, ldc, Fr, HW, pix = I[1], I[2], I[3], I[5], I[8], I[9], I[10]
        heads = N // 192
        samples = I[0] // 192 // -(-HW // pix)
        T = samples * Fr * HW
        A = self.mat(op.p[0], T, K, lda, torch.float16, ext).float()
        W = self.mat(op.p[1], N, K, K, torch.float16, ext).float()
        qkv = (A @ W.t()).half().float().view(samples, Fr, HW, heads, 3, 64)
        q, k, v = qkv[..., 0, :], qkv[..., 1, :], qkv[..., 2, :]
        s = torch.einsum("bfxhd,bgxhd->bxhfg", q, k) * op.f[1]
        o = torch.einsum("bxhfg,bgxhd->bfxhd", torch.softmax(s, dim=-1), v).reshape(T, heads * 64)
        self._st(self.mat(op.p[5], T, heads * 64, ldc, torch.float16, ext), o, torch.float16)

    def _op1_xattn(self, op, ext):
        """T2V_EPI_XATTN: q = A W^T (fp32 accumulate, fp16 q), softmax(q k^T scale) v against the step-invariant text K / V^T, per (sample, head)."""
        I = op.i
        M, N, K, lda, ldc, aw, rps = I[0], I[1], I[2], I[3], I[5], I[13], I[15]
        ldk, Lc, lcp, ks, vs = I[24], I[25], I[26], I[27], I[28]
        A = self.mat(op.p[0], aw if aw else M, K, lda, torch.float16, ext).float()
        if aw:
            A = torch.cat([A, A[: M - aw]], dim=0)
        W = self.mat(op.p[1], N, K, I[4], torch.float16, ext).float()
        q = (A @ W.t()).half().float()
        heads, samples = N // 64, M // rps
        out = torch.empty(M, N)
        for b in range(samples):
            kb = self.mat(op.p[8].shifted(2 * b * ks), Lc, N, ldk, torch.float16, ext).float()            # [Lc, N]
            vt = self.mat(op.p[9].shifted(2 * b * vs), N, Lc, lcp, torch.float16, ext).float()            # [N, Lc]
            qb = q[b * rps:(b + 1) * rps].view(rps, heads, 64)
            s_ = torch.einsum("mhd,khd->hmk", qb, kb.view(Lc, heads, 64)) * op.f[1]
            pm = torch.exp(s_ - s_.max(dim=-1, keepdim=True).values)
            o = torch.einsum("hmk,hdk->mhd", pm.half().float(), vt.view(heads, 64, Lc)) / pm.sum(dim=-1).permute(1, 0)[:, :, None]
            out[b * rps:(b + 1) * rps] = o.reshape(rps, N)
        self._st(self.mat(op.p[5], M, N, ldc, torch.float16, ext), out, torch.float16)

    def _op1(self, op, ext):
        I = op.i
        if I[16] == L.EPI_TATTN:
            return self._op1_tattn(op, ext)
        if I[16] == L.EPI_XATTN:
            return self._op1_xattn(op, ext)
        M, N, K, lda, ldw, ldc, ldr, gather = I[0:8]
        A16 = None
        if gather == L.GATHER_PLAIN:
            aw = I[13]                                             # row wrap: only `aw` operand rows exist, row m >= aw reads row m - aw
            A = self.mat(op.p[0], aw if aw else M, K, lda, torch.float16, ext).float()
            if aw:
                A = torch.cat([A, A[: M - aw]], dim=0)
        elif gather in (L.GATHER_CONV3X3, L.GATHER_CONV3X3_C8):
            Hin, Win, Cin, stride, up, Hout, Wout = I[8], I[9], I[10], I[11], I[12], I[13], I[14]
            nimg = M // (Hout * Wout)
            X = self.view(op.p[0], (nimg, Hin, Win, Cin), (Hin * Win * lda, Win * lda, lda, 1), torch.float16, ext).float()
            if up:
                X = X.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
            Xp = F.pad(X, (0, 0, 0, 2, 0, 2)) if (gather == L.GATHER_CONV3X3 and I[23]) else F.pad(X, (0, 0, 1, 1, 1, 1))
            cols = []
            for ky in range(3):
                for kx in range(3):
                    cols.append(Xp[:, ky:ky + stride * Hout:stride, kx:kx + stride * Wout:stride, :])
            if gather == L.GATHER_CONV3X3:      # k = (chunk, tap, ci): see packing.conv3x3
                A = torch.stack(cols, dim=3).reshape(nimg, Hout, Wout, 9, Cin // 64, 64).permute(0, 1, 2, 4, 3, 5).reshape(M, 9 * Cin)
            else:                               # C8 stem: k = tap*8 + ci
                A = torch.cat(cols, dim=3).reshape(M, 9 * Cin)
            assert K == 9 * Cin
        elif gather == L.GATHER_TCONV3:
            Fr, HW, Cin = I[8], I[9], I[10]
            nb = M // (Fr * HW)
            if I[23]:     # halo layout: the input already holds one frame before/after each clip
                Xp = self.view(op.p[0], (nb, Fr + 2, HW, Cin), ((Fr + 2) * HW * lda, HW * lda, lda, 1), torch.float16, ext).float()
            else:
                X = self.view(op.p[0], (nb, Fr, HW, Cin), (Fr * HW * lda, HW * lda, lda, 1), torch.float16, ext).float()
                Xp = F.pad(X, (0, 0, 0, 0, 1, 1))
            A = torch.stack([Xp[:, kt:kt + Fr] for kt in range(3)], dim=3)          # nb, F, HW, 3, Cin
            A = A.reshape(nb, Fr, HW, 3, Cin // 64, 64).permute(0, 1, 2, 4, 3, 5).reshape(M, 3 * Cin)
        else:
            raise ValueError(gather)
        W = self.mat(op.p[1], N, K, ldw, torch.float16, ext).float()
        acc = A @ W.t()
        if op.p[2].space != "null":
            if I[20]:
                acc = acc + self.view(op.p[2], (M,), (1,), torch.float32, ext)[:, None]
            else:
                acc = acc + self.view(op.p[2], (N,), (1,), torch.float32, ext)[None, :]
        epi = I[16]
        want_stats = epi == L.EPI_STATS
        want_gn = epi == L.EPI_GN
        if want_stats or want_gn:
            epi = L.EPI_NONE
        if epi == L.EPI_GEGLU:
            a = acc.view(M, N // 16, 2, 8)
            val, gate = a[:, :, 0, :].reshape(M, N // 2), a[:, :, 1, :].reshape(M, N // 2)
            res = val * F.gelu(gate)
            n_out = N // 2
        else:
            res, n_out = acc, N
            ln_fused = gather == L.GATHER_PLAIN and I[8] in (1, 2)     # fused LayerNorm second output (2: across column tiles): p[3] = gamma | beta, p[7] = fp16 out
            if op.p[3].space != "null" and not ln_fused:
                rpb, ldrb = I[15], I[21]
                rb = self.mat(op.p[3], M // rpb, N, ldrb, torch.float32, ext)
                res = res + rb.repeat_interleave(rpb, dim=0)
            if I[18] == 1:
                res = F.silu(res)
            if op.p[4].space != "null":
                rw = I[12] if gather == L.GATHER_PLAIN else 0
                R = self.mat(op.p[4], rw if rw else M, N, ldr, torch.float32, ext)
                res = res + (torch.cat([R, R[: M - rw]], dim=0) if rw else R)
        if want_gn:
            # T2V_EPI_GN: GroupNorm (+SiLU) of the fp32 result inside the epilogue — statistics and normalisation on the UNROUNDED values;
            # `out` itself is stored only if someone else reads it (i[29] == 0)
            rows, ld_gn, silu, lo, groups, dead = I[24], I[25], I[26], I[27], I[28], I[29]
            if not dead:
                self._st(self.mat(op.p[5], M, N, ldc, _TD[I[17]], ext), res, _TD[I[17]])
            cpg = N // groups
            x = res.double().view(M // rows, rows, groups, cpg)
            mean = x.mean(dim=(1, 3), keepdim=True)
            var = (x * x).mean(dim=(1, 3), keepdim=True) - mean * mean
            y = ((x - mean) / torch.sqrt(var.clamp_min(0) + op.f[2])).view(M, N).float()
            gb = self.view(op.p[8], (2 * N,), (1,), torch.float32, ext)
            y = y * gb[:N] + gb[N:]
            if silu:
                y = F.silu(y)
            self._st(self.mat(op.p[9], M, N, ld_gn, torch.float16, ext), y, torch.float16)
            if lo:
                self._st(self.view(op.p[9].shifted(2 * N), (M, N), (ld_gn, 1), torch.float16, ext), y.float() - y.half().float(), torch.float16)
            return
        out = self.mat(op.p[5], M, n_out, ldc, _TD[I[17]], ext)
        self._st(out, res, _TD[I[17]])
        if want_stats:       # per 32-row strip: column sums / sums of squares of the STORED values -> fp32 [ceil(M / 32)][2][N]
            stored = self.mat(op.p[5], M, n_out, ldc, _TD[I[17]], ext).float()
            ns = -(-M // 32)
            pad = torch.zeros(ns * 32, N)
            pad[:M] = stored
            strips = pad.view(ns, 32, N)
            sv = self.view(op.p[7], (ns, 2, N), (2 * N, N, 1), torch.float32, ext)
            sv[:, 0] = strips.sum(dim=1)
            sv[:, 1] = (strips * strips).sum(dim=1)
        if gather == L.GATHER_PLAIN and I[11] == 1:          # hi + lo fp16 output: the rounding's low-order image beside the row
            lo_view = self.view(op.p[5].shifted(2 * N), (M, N), (ldc, 1), torch.float16, ext)
            self._st(lo_view, res.float() - res.half().float(), torch.float16)
        if epi != L.EPI_GEGLU and gather == L.GATHER_PLAIN and I[8] in (1, 2):
            gb = self.view(op.p[3], (2 * N,), (1,), torch.float32, ext)
            y = F.layer_norm(res.float(), (N,), gb[:N], gb[N:], op.f[0])
            self._st(self.mat(op.p[7], M, N, I[9], torch.float16, ext), y, torch.float16)

    # GROUPNORM ----------------------------------------------------------------------------------------
    def _op2(self, op, ext):
        n_inst, rows, C, ld_in, groups, in_dt, silu, ld_out, phase, nparts, part = op.i[0:11]
        nparts = max(nparts, 1)
        cpg = C // groups
        x = self.mat(op.p[0], n_inst * rows, C, ld_in, _TD[in_dt], ext).double().view(n_inst, rows, groups, cpg)
        rows_total = op.i[14] if op.i[14] > 0 else rows * nparts
        if phase == 3:       # statistics from the producing GEMM's strips (T2V_EPI_STATS)
            ldn = op.i[17]
            sv = self.view(op.p[6], (n_inst, rows // 32, 2, groups, cpg), (rows // 32 * 2 * ldn, 2 * ldn, ldn, cpg, 1), torch.float32, ext).double()
            s1, s2, n = sv[:, :, 0].sum(dim=(1, 3)), sv[:, :, 1].sum(dim=(1, 3)), rows * cpg
        elif phase == 0:
            s1, s2, n = x.sum(dim=(1, 3)), (x * x).sum(dim=(1, 3)), rows * cpg
        else:
            # scratch head: gathered fp64 {sum, sum of squares} [nparts][n_inst][groups][2] (each rank folds its own block
            # partials before the all-gather)
            pv = self.view(op.p[4], (nparts, n_inst, groups, 2), (n_inst * groups * 2, groups * 2, 2, 1), torch.float64, ext)
            if phase == 1 and op.p[6].space != "null":        # this rank's part from the producing GEMM's strips (round 6)
                ldn = op.i[17]
                sv = self.view(op.p[6], (n_inst, rows // 32, 2, groups, cpg), (rows // 32 * 2 * ldn, 2 * ldn, ldn, cpg, 1), torch.float32, ext).double()
                pv[part, :, :, 0] = sv[:, :, 0].sum(dim=(1, 3))
                pv[part, :, :, 1] = sv[:, :, 1].sum(dim=(1, 3))
                return
            if phase == 1:
                pv[part, :, :, 0] = x.sum(dim=(1, 3))
                pv[part, :, :, 1] = (x * x).sum(dim=(1, 3))
                return
            s1, s2, n = pv[..., 0].sum(dim=0), pv[..., 1].sum(dim=0), rows_total * cpg
        if len(op.p) > 8 and op.p[8].space != "null" and phase != 1:      # second output: the raw input as fp16 (+ low-order image at column i[20])
            raw = self.mat(op.p[0], n_inst * rows, C, ld_in, _TD[in_dt], ext).float()
            self._st(self.mat(op.p[8], n_inst * rows, C, op.i[19], torch.float16, ext), raw, torch.float16)
            if op.i[20]:
                self._st(self.view(op.p[8].shifted(2 * op.i[20]), (n_inst * rows, C), (op.i[19], 1), torch.float16, ext), raw - raw.half().float(), torch.float16)
        mean = (s1 / n).view(n_inst, 1, groups, 1)
        var = (s2 / n).view(n_inst, 1, groups, 1) - mean * mean
        before, after = (op.i[21], op.i[22]) if phase == 2 else (0, 0)
        out_ref = op.p[3]
        if before or after:      # the neighbours' RAW boundary frames in front of / behind x take the same statistics (T2V_OP_STATS_HALO)
            assert n_inst == 1
            item = 4 if _TD[in_dt] == torch.float32 else 2
            rows = rows + before + after
            x = self.mat(op.p[0].shifted(-before * ld_in * item), rows, C, ld_in, _TD[in_dt], ext).double().view(1, rows, groups, cpg)
            out_ref = op.p[3].shifted(-before * ld_out * 2)
        y = ((x - mean) / torch.sqrt(var.clamp_min(0) + op.f[0])).view(n_inst * rows, C).float()
        g = self.view(op.p[1], (C,), (1,), torch.float32, ext)
        b = self.view(op.p[2], (C,), (1,), torch.float32, ext)
        y = y * g + b
        if silu:
            y = F.silu(y)
        self._st(self.mat(out_ref, n_inst * rows, C, ld_out, torch.float16, ext), y, torch.float16)
        if op.i[16]:                                           # low-order image of the fp16 rounding at columns C .. 2C-1
            lo_view = self.view(op.p[3].shifted(2 * C), (n_inst * rows, C), (ld_out, 1), torch.float16, ext)
            self._st(lo_view, y.float() - y.half().float(), torch.float16)

    # LAYERNORM ----------------------------------------------------------------------------------------
    def _op3(self, op, ext):
        M, C, ld_in, ld_out = op.i[0:4]
        x = self.mat(op.p[0], M, C, ld_in, torch.float32, ext)
        g = self.view(op.p[1], (C,), (1,), torch.float32, ext)
        b = self.view(op.p[2], (C,), (1,), torch.float32, ext)
        y = F.layer_norm(x, (C,), g, b, op.f[0])
        self._st(self.mat(op.p[3], M, C, ld_out, torch.float16, ext), y, torch.float16)

    # ATTENTION ----------------------------------------------------------------------------------------
    def _op4(self, op, ext, rel=False):
        nq, nk, heads, bo, bi = op.i[0:5]
        sq, sk, so = op.i[5:8], op.i[8:11], op.i[11:14]
        D = op.i[14] if op.i[14] > 0 else 64

        def v(ref, n, s):
            return self.view(ref, (bo, bi, heads, n, D), (s[1], s[2], D, s[0], 1), torch.float16, ext)

        q, k, vv = v(op.p[0], nq, sq).float(), v(op.p[1], nk, sk).float(), v(op.p[2], nk, sk).float()
        s = torch.einsum("abhid,abhjd->abhij", q, k)
        if rel:
            # LVDM relative-position terms (attention_temporal.py:46-65,120-140)
            R = op.i[15]
            ek = self.view(op.p[4], (2 * R + 1, D), (D, 1), torch.float32, ext)
            ev = self.view(op.p[5], (2 * R + 1, D), (D, 1), torch.float32, ext)
            idx = (torch.arange(nk)[None, :] - (torch.arange(nq)[:, None] + op.i[16])).clamp(-R, R) + R
            s = s + torch.einsum("abhtd,tsd->abhts", q, ek[idx])
        if not rel and op.i[15]:
            s = s.masked_fill(torch.arange(nk)[None, :] > torch.arange(nq)[:, None], float("-inf"))
        p = torch.softmax(s * op.f[0], dim=-1)
        o = torch.einsum("abhij,abhjd->abhid", p, vv)
        if rel:
            o = o + torch.einsum("abhts,tsd->abhtd", p, ev[idx])
        self._st(v(op.p[3], nq, so), o, torch.float16)
        lo_off = op.i[18] if rel else op.i[16]
        if lo_off:                                             # low-order image of the fp16 rounding at out + lo_off elements
            self._st(v(op.p[3].shifted(2 * lo_off), nq, so), o.float() - o.half().float(), torch.float16)

    def _op13(self, op, ext):
        self._op4(op, ext, rel=True)

    # SOFTMAX ------------------------------------------------------------------------------------------
    def _op5(self, op, ext):
        rows, cols, ld_in, ld_out = op.i[0:4]
        x = self.mat(op.p[0], rows, cols, ld_in, torch.float32, ext)
        self._st(self.mat(op.p[1], rows, cols, ld_out, torch.float16, ext), torch.softmax(x * op.f[0], dim=1), torch.float16)

    # NCTHW_TO_CL --------------------------------------------------------------------------------------
    def _op6(self, op, ext):
        B, C, Fr, HW, ld, in_dt = op.i[0:6]
        Bsrc = op.i[6] if 0 < op.i[6] < B else B
        x = self.view(op.p[0], (Bsrc, C, Fr, HW), (C * Fr * HW, Fr * HW, HW, 1), _TD[in_dt], ext).float() * op.f[0]
        x = x.repeat(B // Bsrc, 1, 1, 1)
        out = self.mat(op.p[1], B * Fr * HW, ld, ld, torch.float16, ext)
        out.zero_()
        v = x.permute(0, 2, 3, 1).reshape(B * Fr * HW, C)
        self._st(out[:, :C], v, torch.float16)
        if op.p[2].space != "null":                      # low-order image of the cast
            lo = self.mat(op.p[2], B * Fr * HW, ld, ld, torch.float16, ext)
            lo.zero_()
            self._st(lo[:, :C], v - v.half().float(), torch.float16)
        if op.i[7] and ld >= 2 * C:                      # ... or in the padding channels of the same rows
            self._st(out[:, C:2 * C], v - v.half().float(), torch.float16)

    # CL_TO_NCTHW --------------------------------------------------------------------------------------
    def _op7(self, op, ext):
        B, C, Fr, HW, ld, out_dt = op.i[0:6]
        x = self.mat(op.p[0], B * Fr * HW, C, ld, torch.float32, ext)
        out = self.view(op.p[1], (B, C, Fr, HW), (C * Fr * HW, Fr * HW, HW, 1), _TD[out_dt], ext)
        out.copy_(x.view(B, Fr, HW, C).permute(0, 3, 1, 2).to(out.dtype))

    # TIME_EMBED ---------------------------------------------------------------------------------------
    def _op8(self, op, ext):
        B, dim = op.i[0:2]
        t = self.view(op.p[0], (B,), (1,), torch.float32, ext)
        fr = self.view(op.p[1], (dim // 2,), (1,), torch.float32, ext)
        a = torch.outer(t, fr)
        self._st(self.mat(op.p[2], B, dim, dim, torch.float16, ext), torch.cat([torch.cos(a), torch.sin(a)], dim=1), torch.float16)

    # COPY2D -------------------------------------------------------------------------------------------
    def _op9(self, op, ext):
        rows, cols, lds, ldd, sdt, ddt, act = op.i[0:7]
        x = self.mat(op.p[0], rows, cols, lds, _TD[sdt], ext).float()
        if act == 1:
            x = F.silu(x)
        elif act == 2:
            x = F.gelu(x)
        elif act == 3:
            x = x * torch.sigmoid(1.702 * x)
        out = self.mat(op.p[1], rows, cols, ldd, _TD[ddt], ext)
        self._st(out, x, _TD[ddt])
        if op.p[2].space != "null":                      # low-order image of an fp32 -> fp16 cast
            self._st(self.mat(op.p[2], rows, cols, ldd, torch.float16, ext), x - x.half().float(), torch.float16)

    # EMBED_ROWS ---------------------------------------------------------------------------------------
    def _op14(self, op, ext):
        rows, W, Lp, vocab, tdt = op.i[0:5]
        ids = self.view(op.p[0], (rows,), (1,), torch.int32, ext).long()
        table = self.mat(op.p[1], vocab, W, W, _TD[tdt], ext).float()
        pos = self.mat(op.p[2], Lp, W, W, torch.float32, ext)
        ok = (ids >= 0) & (ids < vocab)
        val = table[ids.clamp(0, vocab - 1)] * ok[:, None]
        self.mat(op.p[3], rows, W, W, torch.float32, ext).copy_(val + pos[torch.arange(rows) % Lp])

    # DDIM_STEP ----------------------------------------------------------------------------------------
    def _op10(self, op, ext):
        C, inner, guided, edt, xdt = op.i[0:5]
        cps = op.i[6] if op.i[6] > 0 else C
        a_recip, a_recipm1, sqrt_aprev, dir_coef, sigma, gscale = op.f[0:6]
        xt = self.view(op.p[0], (C, inner), (inner, 1), _TD[xdt], ext).float()
        e = self.view(op.p[1], (2, C, inner), (C * inner, inner, 1), _TD[edt], ext).float()
        y, u = e[0], e[1]
        o = y.clone()
        g = (torch.arange(C) % cps) < guided
        o[g] = u[g] + gscale * (y[g] - u[g])
        if op.i[5] == 0:
            x0 = a_recip * xt - a_recipm1 * o
            eps = (a_recip * xt - x0) / a_recipm1
            xn = sqrt_aprev * x0 + dir_coef * eps
        else:
            x0 = (xt - a_recip * o) / a_recipm1
            xn = sqrt_aprev * x0 + dir_coef * o
        if op.p[2].space != "null" and sigma != 0.0 and ext.get(L.EXT_NOISE) is not None:
            xn = xn + sigma * self.view(op.p[2], (C, inner), (inner, 1), torch.float32, ext)
        out = self.view(op.p[3], (C, inner), (inner, 1), _TD[xdt], ext)
        out.copy_(xn.to(out.dtype))

    # MEMSET -------------------------------------------------------------------------------------------
    def _op12(self, op, ext):
        n, k = op.i[0], op.i[1]
        acc = None
        for j in range(k):
            t = self.view(op.p[j], (n,), (1,), _TD[op.i[3 + j]], ext).float()
            acc = op.f[j] * t if acc is None else acc + op.f[j] * t
        out = self.view(op.p[6], (n,), (1,), _TD[op.i[2]], ext)
        out.copy_(acc.to(out.dtype))

    def _op16(self, op, ext):
        raise RuntimeError("collectives are executed by parallel.ShardedExecutor, not the interpreter")

    _op17 = _op16
    _op19 = _op16

    # RESHARD_ROWS ---------------------------------------------------------------------------------------
    def _op18(self, op, ext):
        rows, cols, P, s_src, s_dst, ld_src, ld_dst, dt, ld_res = op.i[0:9]
        nparts = op.i[9] if op.i[9] > 1 else 1
        ps_src, ps_dst, ps_res, own, own_is_src = op.i[10:15] if nparts > 1 else (0, 0, 0, -1, 0)
        item = 2 if _TD[dt] == torch.float16 else 4
        r = torch.arange(rows)
        rs, rd = (r // P) * s_src + r % P, (r // P) * s_dst + r % P
        for q in range(nparts):      # (multi-part form, ABI 8: part q at + q * part stride elements; the own part's other side is p[3])
            ps = op.p[3] if (q == own and own_is_src) else op.p[0].shifted(q * ps_src * item)
            pd = op.p[3] if (q == own and not own_is_src) else op.p[1].shifted(q * ps_dst * item)
            src = self.mat(ps, int(rs.max()) + 1, cols, ld_src, _TD[dt], ext)
            dst = self.mat(pd, int(rd.max()) + 1, cols, ld_dst, _TD[dt], ext)
            v = src[rs].clone()
            if op.p[2].space != "null":
                v = v + self.mat(op.p[2].shifted(q * ps_res * 4), int(rd.max()) + 1, cols, ld_res, torch.float32, ext)[rd]
            dst[rd] = v

    # TO_UINT8 (tensor2vid) ------------------------------------------------------------------------------
    def _op15(self, op, ext):
        NI, C, Fr, H, W, in_dt, half, bgr = op.i[0:8]
        si, sc = (op.i[8] & 0xFFFFFFFF) | (op.i[9] << 32), op.i[10]
        sf, sy, sx = (op.i[11] & 0xFFFFFFFF) | (op.i[12] << 32), op.i[13], op.i[14]
        v = self.view(op.p[0], (NI, C, Fr, H, W), (si, sc, sf, sy, sx), _TD[in_dt], ext)
        if half:
            v = v.half()
            v = (v * 0.5)
            v = (v + 0.5).clamp(0, 1)
            v = (v * 255).float()
        else:
            v = v.float()
            v = ((v * 0.5) + 0.5).clamp(0, 1) * 255
        u8 = v.to(torch.uint8).permute(2, 3, 0, 4, 1).reshape(Fr, H, NI * W, C)
        if bgr:
            u8 = u8.flip(-1)
        out = ext[op.p[1].off] if op.p[1].space == "ext" else None
        assert out is not None and out.dtype == torch.uint8
        out.view(Fr, H, NI * W, C).copy_(u8)

    def _op11(self, op, ext):
        nbytes = (op.i[0] & 0xFFFFFFFF) | (op.i[1] << 32)
        assert op.p[0].space == "arena"
        self.arena[op.p[0].off: op.p[0].off + nbytes] = 0
