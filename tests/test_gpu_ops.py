"""GPU (-m gpu): per-kernel parity through the C ABI.  Every op kind / template variant is run
on the MI355X and in the CPU interpreter from identical arena contents; outputs must agree to
rounding (fp32 accumulate order differs; fp16 outputs differ by <= 1-2 ulp)."""
import math

import pytest
import torch

from harness import TD, fill, read, rel_l2, run_both
from sd_webui_text2video_amd import _lib as L
from sd_webui_text2video_amd import packing as pk
from sd_webui_text2video_amd.program import BoundProgram, Buf, Program, Ref

pytestmark = pytest.mark.gpu

GEMM2_TILES = [1, 2, 3, 4, 5, 8, 9, 11, 12]     # csrc/gemm2.hip configurations (t2v_op.i[22]); 6 / 7 / 13-24: experiment builds only


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


def _check(it, got, buf, tol, what=""):
    a, b = read(got, buf).float(), read(it, buf).float()
    assert not torch.isnan(a).any(), f"NaN in GPU output {what}"
    r = rel_l2(a, b)
    assert r < tol, f"{what}: rel-L2 {r:.3e} (max abs {float((a - b).abs().max()):.3e})"


def test_mfma_layout_asymmetric_gemm():
    """A = identity-like / asymmetric operands: catches a transposed or mis-mapped MFMA fragment."""
    M, N, K = 128, 128, 64
    P = Program()
    a, out = P.alloc(M, K, "f16"), P.alloc(M, N, "f32")
    w = {"w": torch.zeros(N, K, dtype=torch.float16)}
    for n in range(N):
        w["w"][n, n % K] = 1.0 + n / 256.0            # asymmetric: W[n, n%K]
    P.gemm("g", a, Ref("weight", 0, "w"), N, K, out)

    def init(it):
        v = it.mat(a.ref, M, K, K, torch.float16, {})
        v.copy_((torch.arange(M).view(-1, 1) * 0.01 + torch.arange(K).view(1, -1) * 1.0).half())
    it, got, _, _ = run_both(P, w, {}, init)
    assert torch.allclose(read(got, out), read(it, out), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 320, 320), (1000, 640, 1280), (77, 4, 2880 // 9 * 9 // 64 * 64),
                                   (2, 1280, 320), (256, 960, 320), (130, 8, 8), (4096, 2560, 320)])
def test_gemm_plain_shapes(M, N, K):
    P = Program()
    a, out = P.alloc(M, K, "f16"), P.alloc(M, N, "f32")
    g = _g(1)
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g)}
    P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"))
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, a, g))
    _check(it, got, out, 2e-5, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("out_dtype,act,with_res,with_rowbias", [("f16", 0, False, False), ("f32", 1, True, False),
                                                                 ("f16", 0, True, True), ("f32", 0, False, True)])
def test_gemm_epilogues(out_dtype, act, with_res, with_rowbias):
    M, N, K, rpb = 384, 320, 640, 96
    P = Program()
    g = _g(2)
    a, out = P.alloc(M, K, "f16"), P.alloc(M, N, out_dtype)
    res = P.alloc(M, N, "f32") if with_res else None
    rb = P.alloc(M // rpb, 2 * N, "f32") if with_rowbias else None
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g)}
    P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), act=act, residual=res,
           rowbias=rb.col_slice(N, 2 * N) if rb is not None else None, rows_per_batch=rpb if rb is not None else 0)

    def init(it):
        fill(it, a, g)
        if res is not None: fill(it, res, g)
        if rb is not None: fill(it, rb, g)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out, 1e-3 if out_dtype == "f16" else 2e-5, "epilogue")


def test_gemm_geglu_epilogue():
    M, C = 200, 320
    P = Program()
    g = _g(3)
    a, out = P.alloc(M, C, "f16"), P.alloc(M, 4 * C, "f16")
    wsrc, bsrc = torch.randn(8 * C, C, generator=g) / math.sqrt(C), torch.randn(8 * C, generator=g) * 0.1
    perm = pk.geglu_perm(4 * C)
    w = {"w": wsrc[perm].half(), "b": bsrc[perm].contiguous()}
    P.gemm("g", a, Ref("weight", 0, "w"), 8 * C, C, out, bias=Ref("weight", 0, "b"), epi=L.EPI_GEGLU)
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, a, g))
    _check(it, got, out, 1e-3, "geglu")
    # and against the unpermuted definition: x * gelu(gate)
    x = read(it, a).float()
    hg = x @ wsrc.half().float().t() + bsrc
    ref = hg[:, :4 * C] * torch.nn.functional.gelu(hg[:, 4 * C:])
    assert rel_l2(read(got, out).float(), ref) < 2e-3


@pytest.mark.parametrize("split", [2, 5, 16])
def test_gemm_split_k(split):
    M, N, K = 96, 1280, 5760
    P = Program()
    P.force_tile = 0
    P.target_cus = 10 * split                     # steer the heuristic: 10 output tiles
    g = _g(4)
    a, out, res = P.alloc(M, K, "f16"), P.alloc(M, N, "f32"), P.alloc(M, N, "f32")
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g)}
    op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), residual=res)
    assert op.i[19] > 1

    def init(it):
        fill(it, a, g); fill(it, res, g)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out, 2e-5, f"split-K {op.i[19]}")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(3, 16, 16, 64, 128, 1, 0), (2, 8, 8, 128, 64, 2, 0),
                                                      (2, 6, 10, 64, 64, 1, 1), (5, 4, 4, 320, 320, 1, 0),
                                                      (2, 9, 7, 64, 4, 1, 0)])
def test_conv3x3_gather(B, H, W, Cin, Cout, stride, up):
    Ho, Wo = (2 * H, 2 * W) if up else ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    P = Program()
    g = _g(5)
    a, out = P.alloc(B * H * W, Cin, "f16"), P.alloc(B * Ho * Wo, Cout, "f32")
    wt = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    w = {"w": pk.conv3x3(wt).half(), "b": torch.randn(Cout, generator=g)}
    P.gemm("c", a, Ref("weight", 0, "w"), Cout, 9 * Cin, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
           conv=dict(Hin=H, Win=W, Cin=Cin, stride=stride, up=up, Hout=Ho, Wout=Wo))
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, a, g))
    _check(it, got, out, 2e-5, "conv3x3")
    # independent check against torch's own convolution
    x = read(it, a).float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    if up:
        x = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(x, wt.half().float(), w["b"], stride=stride, padding=1)
    mine = read(got, out).view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert rel_l2(mine, ref) < 1e-4


def test_conv3x3_c8_stem():
    B, H, W, Cout = 3, 16, 16, 320
    P = Program()
    g = _g(6)
    a, out = P.alloc(B * H * W, 8, "f16"), P.alloc(B * H * W, Cout, "f32")
    wt = torch.randn(Cout, 4, 3, 3, generator=g) / 6.0
    w = {"w": pk.conv3x3(wt, 8).half(), "b": torch.randn(Cout, generator=g)}
    P.gemm("c", a, Ref("weight", 0, "w"), Cout, 72, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3_C8,
           conv=dict(Hin=H, Win=W, Cin=8, stride=1, up=0, Hout=H, Wout=W))

    def init(it):
        v = fill(it, a, g)
        v[:, 4:] = 0
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out, 2e-5, "stem conv")
    x = read(it, a).float()[:, :4].reshape(B, H, W, 4).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x, wt.half().float(), w["b"], padding=1)
    assert rel_l2(read(got, out).view(B, H, W, Cout).permute(0, 3, 1, 2), ref) < 1e-4


@pytest.mark.parametrize("B,F,HW,C", [(2, 5, 16, 64), (1, 24, 4, 128), (2, 3, 64, 320)])
def test_temporal_conv_gather(B, F, HW, C):
    P = Program()
    g = _g(7)
    M = B * F * HW
    a, out, res = P.alloc(M, C, "f16"), P.alloc(M, C, "f32"), P.alloc(M, C, "f32")
    wt = torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)
    w = {"w": pk.tconv3(wt).half(), "b": torch.randn(C, generator=g)}
    P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * C, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3,
           conv=dict(F=F, HW=HW, Cin=C), residual=res)

    def init(it):
        fill(it, a, g); fill(it, res, g)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out, 2e-5, "tconv")
    x = read(it, a).float().view(B, F, HW, 1, C).permute(0, 4, 1, 2, 3)
    ref = torch.nn.functional.conv3d(x, wt.half().float(), w["b"], padding=(1, 0, 0))
    ref = ref.permute(0, 2, 3, 4, 1).reshape(M, C) + read(it, res)
    assert rel_l2(read(got, out), ref) < 1e-4


@pytest.mark.parametrize("n_inst,rows,C,dt,silu", [(6, 256, 320, "f32", True), (2, 24 * 64, 640, "f32", True),
                                                   (4, 16, 1280, "f32", False), (3, 64, 64, "f16", True),
                                                   (2, 100, 2560, "f32", True), (2, 4096, 128, "f32", True)])
@pytest.mark.parametrize("variant", ["three_launch", "cooperative", "single_launch"])
def test_groupnorm(n_inst, rows, C, dt, silu, variant):
    if variant == "single_launch" and (C // 32) % 4 != 0:
        pytest.skip("single-launch GroupNorm needs (C/groups) % 4 == 0")
    P = Program()
    P.gn_coop = variant == "cooperative"          # single-pass kernel with a grid barrier (csrc/norm.hip) vs statistics / fold / apply
    P.gn_fused_slice_bytes = 1 << 30 if variant == "single_launch" else 0
    P.gn_fused_total_bytes = 1 << 30
    P.begin()
    g = _g(8)
    x, out = P.alloc(n_inst * rows, C, dt), P.alloc(n_inst * rows, C, "f16")
    out2 = P.alloc(n_inst * rows, C, "f16")
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "b": 0.1 * torch.randn(C, generator=g)}
    P.groupnorm("gn", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), out, n_inst=n_inst, eps=1e-5, silu=silu)
    P.groupnorm("gn2", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), out2, n_inst=n_inst, eps=1e-6, silu=silu)   # ping-pong buffer
    P.finish()
    assert all(op.i[12] == (variant == "single_launch") and op.i[15] == (variant == "cooperative") for op in P.ops if op.kind == L.OP_GROUPNORM)

    def init(it):
        v = fill(it, x, g, scale=2.0)
        v += 0.7
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out, 1e-3, "groupnorm")
    _check(it, got, out2, 1e-3, "groupnorm (second, other stats buffer)")


@pytest.mark.parametrize("M,C", [(1000, 320), (77, 512), (513, 640), (64, 1280), (10, 64)])
def test_layernorm(M, C):
    P = Program()
    g = _g(9)
    x, out = P.alloc(M, C, "f32"), P.alloc(M, C, "f16")
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "b": 0.1 * torch.randn(C, generator=g)}
    P.layernorm("ln", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), out)
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, x, g, 3.0))
    _check(it, got, out, 1e-3, "layernorm")


@pytest.mark.parametrize("kind,B,F,hw,heads,Lc", [("spatial", 1, 2, 1024, 5, 0), ("spatial", 2, 3, 256, 2, 0),
                                                  ("spatial", 1, 2, 16, 4, 0), ("spatial", 1, 1, 144, 3, 0),
                                                  ("cross", 2, 3, 64, 2, 77), ("cross", 1, 2, 256, 5, 7),
                                                  ("temporal", 2, 24, 16, 3, 0), ("temporal", 1, 125, 4, 2, 0),
                                                  ("temporal", 1, 3, 64, 8, 0)])
def test_attention(kind, B, F, hw, heads, Lc):
    inner = heads * 64
    M = B * F * hw
    P = Program()
    g = _g(10)
    scale = 64 ** -0.5
    if kind == "cross":
        q, kv, o = P.alloc(M, inner, "f16"), P.alloc(B * Lc, 2 * inner + 64, "f16"), P.alloc(M, inner, "f16")
        k, v = kv.col_slice(64, 64 + inner), kv.col_slice(64 + inner, 64 + 2 * inner)
        P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=Lc, heads=heads, b_outer=B, b_inner=F,
                    q_strides=(inner, F * hw * inner, hw * inner), kv_strides=(kv.ld, Lc * kv.ld, 0),
                    o_strides=(inner, F * hw * inner, hw * inner), scale=scale)
        bufs = [q, kv]
    else:
        qkv, o = P.alloc(M, 3 * inner, "f16"), P.alloc(M, inner, "f16")
        ld = 3 * inner
        q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
        if kind == "spatial":
            P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=hw, heads=heads, b_outer=B * F, b_inner=1,
                        q_strides=(ld, hw * ld, 0), kv_strides=(ld, hw * ld, 0), o_strides=(inner, hw * inner, 0), scale=scale)
        else:
            P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=F, nk=F, heads=heads, b_outer=B, b_inner=hw,
                        q_strides=(hw * ld, F * hw * ld, ld), kv_strides=(hw * ld, F * hw * ld, ld),
                        o_strides=(hw * inner, F * hw * inner, inner), scale=scale)
        bufs = [qkv]

    def init(it):
        for b in bufs:
            fill(it, b, g, 1.5)
    it, got, _, _ = run_both(P, {}, {}, init)
    _check(it, got, o, 3e-3, f"attention {kind}")


@pytest.mark.parametrize("D", [40, 80, 160])
@pytest.mark.parametrize("kind,B,F,hw,Lc", [("spatial", 1, 2, 256, 0), ("spatial", 1, 3, 20, 0), ("cross", 2, 2, 64, 77),
                                             ("cross", 1, 1, 1024, 9)])
def test_attention_lvdm_head_dims(D, kind, B, F, hw, Lc):
    """8 heads of C/8 channels (VideoCrafter LVDM: head_dim 40 / 80 / 160); the zero-padding to the MFMA
    granularity must stay inside the kernel (the neighbouring head's columns are live data)."""
    heads = 8 if D < 160 else 3
    inner = heads * D
    M = B * F * hw
    P = Program()
    g = _g(40 + D)
    scale = D ** -0.5
    if kind == "cross":
        q, kv, o = P.alloc(M, inner, "f16"), P.alloc(B * Lc, 2 * inner + 8, "f16"), P.alloc(M, inner, "f16")
        k, v = kv.col_slice(8, 8 + inner), kv.col_slice(8 + inner, 8 + 2 * inner)
        P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=Lc, heads=heads, b_outer=B, b_inner=F,
                    q_strides=(inner, F * hw * inner, hw * inner), kv_strides=(kv.ld, Lc * kv.ld, 0),
                    o_strides=(inner, F * hw * inner, hw * inner), scale=scale, head_dim=D)
        bufs = [q, kv]
    else:
        qkv, o = P.alloc(M, 3 * inner, "f16"), P.alloc(M, inner, "f16")
        ld = 3 * inner
        q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
        P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=hw, heads=heads, b_outer=B * F, b_inner=1,
                    q_strides=(ld, hw * ld, 0), kv_strides=(ld, hw * ld, 0), o_strides=(inner, hw * inner, 0), scale=scale,
                    head_dim=D)
        bufs = [qkv]

    def init(it):
        for b in bufs:
            fill(it, b, g, 1.5)
        fill(it, o, g, 3.0)          # poison: every output column must be overwritten exactly once
    it, got, _, _ = run_both(P, {}, {}, init)
    _check(it, got, o, 3e-3, f"attention {kind} d={D}")


@pytest.mark.parametrize("mfma", [False, True])
@pytest.mark.parametrize("D,T,R", [(40, 16, 16), (80, 16, 16), (160, 16, 16), (40, 24, 16), (64, 5, 2), (160, 32, 16),
                                   (40, 9, 16), (80, 16, 31), (64, 16, 20), (40, 17, 16), (40, 32, 31), (160, 20, 19)])
def test_relpos_temporal_attention(D, T, R, mfma):
    """LVDM TemporalCrossAttention with relative-position K / V terms (attention_temporal.py:107-144): the VALU kernel (default)
    and, with `relpos_mfma=True` (t2v_op.i[17]) where R >= T - 1 (no clipping: the released model's 16 frames / R = 16 and the
    added shapes), the MFMA kernel of round 3 (Q.Ek^T and P_skew.Ev as GEMMs against the tables with an LDS skew; the clipped
    shapes fall through to the VALU kernel).  The tables are fp16-representable, as in a `.half()` model (the MFMA kernel stages
    them as fp16)."""
    heads, B, hw = 8 if D < 160 else 2, 2, 12
    inner = heads * D
    M = B * T * hw
    P = Program()
    g = _g(50 + D + T)
    qkv, o = P.alloc(M, 3 * inner, "f16"), P.alloc(M, inner, "f16")
    ld = 3 * inner
    q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
    w = {"ek": (torch.randn(2 * R + 1, D, generator=g) * 0.5).half().float(), "ev": (torch.randn(2 * R + 1, D, generator=g) * 0.5).half().float()}
    P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=T, nk=T, heads=heads, b_outer=B, b_inner=hw,
                q_strides=(hw * ld, T * hw * ld, ld), kv_strides=(hw * ld, T * hw * ld, ld),
                o_strides=(hw * inner, T * hw * inner, inner), scale=D ** -0.5, head_dim=D,
                rel_k=Ref("weight", 0, "ek"), rel_v=Ref("weight", 0, "ev"), max_rel=R, relpos_mfma=mfma)
    assert P.ops[0].kind == L.OP_RELPOS_ATTN and P.ops[0].i[17] == int(mfma)

    def init(it):
        fill(it, qkv, g, 1.2)
        fill(it, o, g, 3.0)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, o, 2e-3, f"relpos attention d={D} T={T} mfma={mfma}")


@pytest.mark.parametrize("D,T,R,hw,lo", [(40, 16, 16, 12, False), (80, 16, 16, 12, True), (160, 16, 16, 12, False), (64, 5, 8, 12, False),
                                         (40, 9, 16, 7, True), (80, 16, 31, 5, False), (64, 16, 20, 12, False), (40, 1, 4, 3, False),
                                         (160, 12, 11, 9, True), (40, 16, 16, 700, False), (160, 16, 16, 300, False)])
def test_relpos_temporal_attention_persistent_mfma(D, T, R, hw, lo):
    """Round 5: t2v_op.i[17] = 2 — whole clips of <= 16 frames on the persistent MFMA kernel with the tables packed for it
    (packing.relpos_table16: 32 slots from row R-(T-1), fp16, the V-side table transposed) against the interpreter's explicit formula
    (attention_temporal.py:107-144).  hw = 700 / 300: more items than the resident grid holds, so every wave walks several."""
    from sd_webui_text2video_amd import packing as pk
    heads, B = 8 if D < 160 else 2, 2
    inner = heads * D
    M = B * T * hw
    P = Program()
    g = _g(150 + D + T + hw)
    qkv, o = P.alloc(M, 3 * inner, "f16"), P.alloc(M, 2 * inner if lo else inner, "f16")
    ld, lo_ld = 3 * inner, o.ld
    q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
    ek, ev = (torch.randn(2 * R + 1, D, generator=g) * 0.5).half().float(), (torch.randn(2 * R + 1, D, generator=g) * 0.5).half().float()
    w = {"ek": ek, "ev": ev, "ek16": pk.relpos_table16(ek, T, False), "ev16": pk.relpos_table16(ev, T, True)}
    P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=T, nk=T, heads=heads, b_outer=B, b_inner=hw,
                q_strides=(hw * ld, T * hw * ld, ld), kv_strides=(hw * ld, T * hw * ld, ld),
                o_strides=(hw * lo_ld, T * hw * lo_ld, lo_ld), scale=D ** -0.5, head_dim=D, lo_off=inner if lo else 0,
                rel_k=Ref("weight", 0, "ek"), rel_v=Ref("weight", 0, "ev"), rel_k16=Ref("weight", 0, "ek16"), rel_vT16=Ref("weight", 0, "ev16"),
                max_rel=R, relpos_mfma=2)
    assert P.ops[0].kind == L.OP_RELPOS_ATTN and P.ops[0].i[17] == 2

    def init(it):
        fill(it, qkv, g, 1.2)
        fill(it, o, g, 3.0)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, o.col_slice(0, inner), 2e-3, f"relpos attention (persistent MFMA) d={D} T={T} hw={hw}")
    if lo:          # hi + lo is the fp32 result to ~2^-22: compare the sum with the interpreter's sum
        want, have = read(it, o).float(), read(got, o).float()
        s_w, s_h = want[:, :inner] + want[:, inner:], have[:, :inner] + have[:, inner:]
        assert float((s_h - s_w).norm() / s_w.norm()) < 2e-3


def test_relpos_persistent_mfma_falls_back_where_it_does_not_apply():
    """A clip of more than 16 frames, a clipped table (R < T - 1) or missing packed tables: the op keeps i[17] = 0 (VALU kernel)."""
    for T, R, packed in ((24, 24, True), (16, 8, True), (16, 16, False)):
        P = Program()
        qkv, o = P.alloc(T * 4, 3 * 40, "f16"), P.alloc(T * 4, 40, "f16")
        extra = dict(rel_k16=Ref("weight", 0, "a"), rel_vT16=Ref("weight", 0, "b")) if packed else {}
        P.attention("a", qkv.ref, qkv.ref, qkv.ref, o.ref, nq=T, nk=T, heads=1, b_outer=1, b_inner=4, q_strides=(480, 0, 120),
                    kv_strides=(480, 0, 120), o_strides=(160, 0, 40), scale=0.1, head_dim=40, rel_k=Ref("weight", 0, "ek"),
                    rel_v=Ref("weight", 0, "ev"), max_rel=R, relpos_mfma=2, **extra)
        assert P.ops[0].i[17] == 0


def test_attention_peaked_softmax():
    """Large logits: exercises the running-max rescale path of the online softmax."""
    hw, heads = 300, 1
    P = Program()
    g = _g(11)
    qkv, o = P.alloc(hw, 192, "f16"), P.alloc(hw, 64, "f16")
    q, k, v = qkv.col_slice(0, 64), qkv.col_slice(64, 128), qkv.col_slice(128, 192)
    P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=hw, heads=1, b_outer=1, b_inner=1, q_strides=(192, 0, 0),
                kv_strides=(192, 0, 0), o_strides=(64, 0, 0), scale=0.125)

    def init(it):
        t = fill(it, qkv, g, 1.0)
        t[:, :128] *= 6.0           # |logit| up to ~ 6*6*8 -> strongly peaked rows, late maxima
        t[250:, 64:128] *= 2.0
    it, got, _, _ = run_both(P, {}, {}, init)
    _check(it, got, o, 5e-3, "peaked attention")


def test_softmax_rows():
    P = Program()
    g = _g(12)
    x, out = P.alloc(300, 1024, "f32"), P.alloc(300, 1024, "f16")
    P.softmax("s", x, out, 0.044)
    it, got, _, _ = run_both(P, {}, {}, lambda it: fill(it, x, g, 30.0))
    _check(it, got, out, 1e-3, "softmax")


def test_layout_time_embed_copy_ddim():
    B, C, F, HW = 2, 4, 3, 64
    P = Program()
    g = _g(13)
    tok = P.alloc(B * F * HW, 8, "f16")
    P.ncthw_to_cl("in", Ref("ext", L.EXT_X), "f32", tok, B=B, C=C, F=F, HW=HW, scale=0.5)
    y32 = P.alloc(B * F * HW, 8, "f32")
    P.copy2d("cast", tok, y32, act=1)
    P.cl_to_ncthw("out", y32, Ref("ext", L.EXT_OUT), "f16", B=B, C=C, F=F, HW=HW)
    te = P.alloc(B, 320, "f16")
    w = {"fr": torch.pow(10000, -torch.arange(160).float().div(160))}
    P.time_embed("te", Ref("ext", L.EXT_T), Ref("weight", 0, "fr"), te)
    P.ddim_step("s", C=C, inner=F * HW, guided=2, eps_dtype="f16", x_dtype="f32")
    P.ops[-1].f[0:6] = [1.3, 0.83, 0.9, 0.43, 0.2, 9.0]
    ext = {L.EXT_X: torch.randn(B, C, F, HW, generator=g), L.EXT_OUT: torch.zeros(B, C, F, HW, dtype=torch.float16),
           L.EXT_T: torch.tensor([981.0, 1.0]), L.EXT_XT: torch.randn(1, C, F * HW, generator=g),
           L.EXT_EPS: torch.randn(2, C, F * HW, generator=g).half(), L.EXT_NOISE: torch.randn(1, C, F * HW, generator=g),
           L.EXT_XT_OUT: torch.zeros(1, C, F * HW)}
    it, got, ext_ref, ext_got = run_both(P, w, ext, lambda it: None)
    assert rel_l2(ext_got[L.EXT_OUT].float(), ext_ref[L.EXT_OUT].float()) < 1e-3
    assert rel_l2(read(got, te).float(), read(it, te).float()) < 2e-3
    assert rel_l2(ext_got[L.EXT_XT_OUT], ext_ref[L.EXT_XT_OUT]) < 1e-6


# ---- second-generation GEMM (csrc/gemm2.hip): 256/128 x 256/320 tiles, 4-/3-stage DMA ring --------
@pytest.mark.parametrize("tile", GEMM2_TILES)
@pytest.mark.parametrize("M,N,K", [(512, 640, 320), (300, 320, 192), (1000, 960, 1280), (256, 512, 64), (77, 1280, 128),
                                   (2304, 320, 2880)])
def test_gemm2_plain_tiles(tile, M, N, K):
    P = Program()
    P.force_tile = tile
    a, out = P.alloc(M, K, "f16"), P.alloc(M, N, "f32")
    g = _g(21)
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g)}
    op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), allow_splitk=False)
    assert op.i[22] == tile
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, a, g))
    _check(it, got, out, 2e-5, f"gemm2 tile {tile} {M}x{N}x{K}")


@pytest.mark.parametrize("tile", GEMM2_TILES)
def test_gemm2_epilogues_and_geglu(tile):
    M, C, rpb = 384, 320, 96
    P = Program()
    P.force_tile = tile
    g = _g(22)
    a = P.alloc(M, C, "f16")
    out16, res, rb = P.alloc(M, C, "f16"), P.alloc(M, C, "f32"), P.alloc(M // rpb, C, "f32")
    gout = P.alloc(M, 4 * C, "f16")
    wsrc, bsrc = torch.randn(8 * C, C, generator=g) / math.sqrt(C), torch.randn(8 * C, generator=g) * 0.1
    perm = pk.geglu_perm(4 * C)
    w = {"w": (torch.randn(C, C, generator=g) / math.sqrt(C)).half(), "b": torch.randn(C, generator=g),
         "wg": wsrc[perm].half(), "bg": bsrc[perm].contiguous()}
    P.gemm("e", a, Ref("weight", 0, "w"), C, C, out16, bias=Ref("weight", 0, "b"), act=1, residual=res, rowbias=rb,
           rows_per_batch=rpb, allow_splitk=False)
    P.gemm("g", a, Ref("weight", 0, "wg"), 8 * C, C, gout, bias=Ref("weight", 0, "bg"), epi=L.EPI_GEGLU, allow_splitk=False)

    def init(it):
        fill(it, a, g); fill(it, res, g); fill(it, rb, g)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out16, 1e-3, f"gemm2 epilogue tile {tile}")
    _check(it, got, gout, 1e-3, f"gemm2 geglu tile {tile}")


@pytest.mark.parametrize("tile", GEMM2_TILES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(3, 16, 16, 64, 320, 1, 0), (2, 8, 8, 128, 256, 2, 0),
                                                      (2, 6, 10, 64, 640, 1, 1)])
def test_gemm2_conv3x3(tile, B, H, W, Cin, Cout, stride, up):
    Ho, Wo = (2 * H, 2 * W) if up else ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    P = Program()
    P.force_tile = tile
    g = _g(23)
    a, out = P.alloc(B * H * W, Cin, "f16"), P.alloc(B * Ho * Wo, Cout, "f32")
    wt = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    w = {"w": pk.conv3x3(wt).half(), "b": torch.randn(Cout, generator=g)}
    P.gemm("c", a, Ref("weight", 0, "w"), Cout, 9 * Cin, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
           conv=dict(Hin=H, Win=W, Cin=Cin, stride=stride, up=up, Hout=Ho, Wout=Wo), allow_splitk=False)
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, a, g))
    _check(it, got, out, 2e-5, f"gemm2 conv3x3 tile {tile}")


@pytest.mark.parametrize("tile", GEMM2_TILES)
def test_gemm2_temporal_conv_and_split_k(tile):
    B, F, HW, C = 2, 5, 16, 640
    P = Program()
    P.force_tile = tile
    P.target_cus = 16 if tile <= 3 else 64   # 1-10 output tiles -> split-K 2..3
    g = _g(24)
    M = B * F * HW
    a, out, res = P.alloc(M, C, "f16"), P.alloc(M, C, "f32"), P.alloc(M, C, "f32")
    wt = torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)
    w = {"w": pk.tconv3(wt).half(), "b": torch.randn(C, generator=g)}
    op = P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * C, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3,
                conv=dict(F=F, HW=HW, Cin=C), residual=res)
    assert op.i[19] > 1

    def init(it):
        fill(it, a, g); fill(it, res, g)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out, 2e-5, f"gemm2 tconv split-K tile {tile}")


def test_groupnorm_split_phases_two_parts():
    """T-sharding: statistics partials of two 'ranks' folded in the apply phase (here both parts on one GPU); UNEVEN slices
    (2 frames + 1 frame of 24 rows): every part folds its own block partials to one {sum, sum of squares} pair per group, the
    mean is over the rows of all parts."""
    from sd_webui_text2video_amd.program import TShardSpec
    C, fr = 320, 24
    rows0, rows1 = 2 * fr, fr
    P = Program()
    P.target_cus = 1          # several statistics workgroups per part: rows per workgroup 4 -> 12 / 6 partial slots
    g = _g(31)
    x0, x1 = P.alloc(rows0, C, "f32"), P.alloc(rows1, C, "f32")
    o0, o1 = P.alloc(rows0, C, "f16"), P.alloc(rows1, C, "f16")
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "b": 0.1 * torch.randn(C, generator=g)}
    # two programs' worth of ops sharing ONE scratch: emit by hand through the same emitter, then alias the scratch
    P.groupnorm("a", x0, Ref("weight", 0, "g"), Ref("weight", 0, "b"), o0, n_inst=1, eps=1e-5, silu=True, shard=TShardSpec.make(3, 2, 0))
    P.groupnorm("b", x1, Ref("weight", 0, "g"), Ref("weight", 0, "b"), o1, n_inst=1, eps=1e-5, silu=True, shard=TShardSpec.make(3, 2, 1))
    ops = [op for op in P.ops if op.kind == L.OP_GROUPNORM]          # a.stats, a.apply, b.stats, b.apply
    scratch = ops[0].p[4]
    for op in ops:
        op.p[4] = scratch
    assert ops[0].i[14] == ops[2].i[14] == rows0 + rows1
    P.ops = [ops[0], ops[2], ops[1], ops[3]]                         # both statistics first, then both applies
    rows = None

    def init(it):
        fill(it, x0, g, 2.0); fill(it, x1, g, 0.5)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, o0, 1e-3, "split GN part 0")
    _check(it, got, o1, 1e-3, "split GN part 1")
    # and against one GroupNorm over the concatenation
    x = torch.cat([read(it, x0), read(it, x1)]).double().view(1, rows0 + rows1, 32, C // 32)
    m, v = x.mean(dim=(1, 3), keepdim=True), x.var(dim=(1, 3), unbiased=False, keepdim=True)
    ref = torch.nn.functional.silu((((x - m) / torch.sqrt(v + 1e-5)).view(rows0 + rows1, C).float() * w["g"] + w["b"]))
    assert rel_l2(torch.cat([read(got, o0), read(got, o1)]).float(), ref) < 1e-3


@pytest.mark.parametrize("tile", [0] + GEMM2_TILES)
def test_temporal_conv_halo_layout(tile):
    B, F, HW, C = 1, 3, 64, 320
    P = Program()
    P.force_tile = tile
    g = _g(32)
    a = P.alloc((F + 2) * HW, C, "f16")
    out = P.alloc(F * HW, C, "f32")
    wt = torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)
    w = {"w": pk.tconv3(wt).half(), "b": torch.randn(C, generator=g)}
    P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * C, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3,
           conv=dict(F=F, HW=HW, Cin=C), halo=True, allow_splitk=False)
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, a, g))
    _check(it, got, out, 2e-5, f"tconv halo tile {tile}")
    x = read(it, a).float().view(1, F + 2, HW, 1, C).permute(0, 4, 1, 2, 3)
    ref = torch.nn.functional.conv3d(x, wt.half().float(), w["b"], padding=(0, 0, 0)).permute(0, 2, 3, 4, 1).reshape(F * HW, C)
    assert rel_l2(read(got, out), ref) < 1e-4


def test_lincomb_and_ldm_ddim_update():
    """T2V_OP_LINCOMB (mixed fp16/fp32 terms, ragged length) and T2V_OP_DDIM_STEP mode 1 (LDM form with
    eta-noise) against the torch restatement of their documented semantics (include/t2v_hip.h)."""
    from sd_webui_text2video_amd import samplers as S
    from test_samplers_cpu import _ddim_update_cpu, _lincomb_cpu
    g = _g(3)
    n = 4 * 3 * 16 * 16 + 0
    for shape, dts in (((1, 4, 3, 16, 16), ["f32", "f16", "f32"]), ((1, 4, 5, 7, 9), ["f32"] * 6), ((1, 4, 1, 3, 3), ["f16", "f16"])):
        terms = [(float(torch.randn(1, generator=g)), torch.randn(shape, generator=g).to(torch.float16 if d == "f16" else torch.float32))
                 for d in dts]
        for odt in (torch.float32, torch.float16):
            want = _lincomb_cpu(torch.empty(shape, dtype=odt), terms)
            got = S._lincomb(torch.empty(shape, dtype=odt, device="cuda"), [(c, t.cuda()) for c, t in terms])
            torch.cuda.synchronize()
            tol = 2e-3 if odt == torch.float16 else 1e-6
            assert rel_l2(got.float().cpu(), want.float()) < tol
    shape = (1, 4, 3, 16, 16)
    xt, noise = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    eps = torch.randn((2,) + shape[1:], generator=g).half()
    coef = [0.6, 0.8, 0.9, 0.3, 0.25, 7.5]
    for guided, e in ((4, eps), (0, eps[0:1].contiguous())):
        want = _ddim_update_cpu(torch.empty(shape), xt, e, noise, coef, guided, 1)
        got = S._ddim_update(torch.empty(shape, device="cuda"), xt.cuda(), e.cuda(), noise.cuda(), coef, guided, 1)
        torch.cuda.synchronize()
        assert rel_l2(got.cpu(), want) < 1e-6


# ---- CLIP text tower ops ---------------------------------------------------------------------
@pytest.mark.parametrize("B,Lseq,heads", [(2, 77, 16), (3, 20, 2), (1, 200, 3), (2, 64, 1), (1, 129, 2)])
def test_attention_causal(B, Lseq, heads):
    """i[15] = 1: key s contributes to query t only if s <= t (one- and four-wave kernels, several key tiles)."""
    inner = heads * 64
    M = B * Lseq
    P = Program()
    g = _g(40)
    qkv, o = P.alloc(M, 3 * inner, "f16"), P.alloc(M, inner, "f16")
    ld = 3 * inner
    q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
    P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=Lseq, nk=Lseq, heads=heads, b_outer=B, b_inner=1,
                q_strides=(ld, Lseq * ld, 0), kv_strides=(ld, Lseq * ld, 0), o_strides=(inner, Lseq * inner, 0),
                scale=64 ** -0.5, causal=True)
    it, got, _, _ = run_both(P, {}, {}, lambda it: fill(it, qkv, g, 1.5))
    _check(it, got, o, 3e-3, "causal attention")
    # row 0 attends to key 0 only: out[0] == v[0] exactly
    a = read(got, o)
    v0 = read(got, qkv)[:, 2 * inner:]
    for b in range(B):
        assert torch.equal(a[b * Lseq], v0[b * Lseq])


def test_copy2d_gelu_variants_and_embed_rows():
    P = Program()
    g = _g(41)
    rows, cols = 154, 512
    src = P.alloc(rows, cols, "f16")
    d2, d3 = P.alloc(rows, cols, "f16"), P.alloc(rows, cols, "f32")
    P.copy2d("gelu", src, d2, act=2)
    P.copy2d("quick", src, d3, act=3)
    inplace = P.alloc(rows, cols, "f16")
    P.copy2d("cp", src, inplace)
    P.copy2d("gelu.inplace", inplace, inplace, act=2)
    vocab, W, Lp = 300, 128, 77
    emb = P.alloc(2 * Lp, W, "f32")
    P.embed_rows("emb", Ref("ext", L.EXT_X), Ref("weight", 0, "tab"), "f32", Ref("weight", 0, "pos"), emb, L_pos=Lp, vocab=vocab)
    emb16 = P.alloc(2 * Lp, W, "f32")
    P.embed_rows("emb16", Ref("ext", L.EXT_X), Ref("weight", 0, "tab16"), "f16", Ref("weight", 0, "pos"), emb16, L_pos=Lp, vocab=vocab)
    tab = torch.randn(vocab, W, generator=g)
    w = {"tab": tab, "tab16": tab.half(), "pos": torch.randn(Lp, W, generator=g)}
    ids = torch.randint(0, vocab, (2 * Lp,), generator=g).to(torch.int32)
    ids[5], ids[9] = vocab + 3, -1                      # out of range: zeros + positional, never an OOB read
    it, got, _, _ = run_both(P, w, {L.EXT_X: ids}, lambda it: fill(it, src, g, 3.0))
    _check(it, got, d2, 1e-3, "gelu")
    _check(it, got, d3, 1e-5, "quick gelu")
    assert torch.equal(read(got, inplace), read(got, d2))
    assert torch.equal(read(got, emb), read(it, emb))
    assert torch.equal(read(got, emb16), read(it, emb16))
    assert torch.equal(read(got, emb)[5], w["pos"][5])


def test_reshard_rows_pack_and_unpack_with_residual():
    """T2V_OP_RESHARD_ROWS: chunked row regrouping between the frame-sharded and the pixel-sharded layout of a clip (fp16 pack)
    and back (fp32 unpack + residual at the destination row), strided source / destination / residual."""
    frames, hw, hwr, C = 3, 16, 4, 64
    for dt, with_res in (("f16", False), ("f32", True)):
        P = Program()
        g = _g(41)
        src = P.alloc(frames * hw, C, dt, ld=C + 8)
        packed = P.alloc(frames * hwr, C, dt)
        back = P.alloc(frames * hw, C, dt, ld=C + 16)
        res = P.alloc(frames * hw, C, "f32", ld=C + 4) if with_res else None
        q = 2                                                            # the pixel range of "rank" 2
        P.reshard_rows("pack", src.row_slice(q * hwr, src.rows), packed, rows=frames * hwr, chunk=hwr, s_src=hw, s_dst=hwr)
        P.reshard_rows("unpack", packed, back.row_slice(q * hwr, back.rows), rows=frames * hwr, chunk=hwr, s_src=hwr, s_dst=hw,
                       residual=res.row_slice(q * hwr, res.rows) if with_res else None)

        def init(it):
            fill(it, src, g)
            it.mat(back.ref, back.rows, back.cols, back.ld, TD[dt], {}).zero_()
            if with_res:
                fill(it, res, g)
        it, got, _, _ = run_both(P, {}, {}, init)
        a, b = read(it, back), read(got, back)
        assert torch.equal(a, b)
        s_, r_ = read(it, src).float(), (read(it, res) if with_res else None)
        want = torch.zeros(frames * hw, C)
        for f in range(frames):
            rows = slice(f * hw + q * hwr, f * hw + (q + 1) * hwr)
            want[rows] = s_[rows] + (r_[rows] if with_res else 0)
        assert torch.equal(b.float(), want.to(TD[dt]).float())
        assert torch.equal(read(got, packed).float(), torch.cat([s_[f * hw + q * hwr: f * hw + (q + 1) * hwr] for f in range(frames)]))


def test_reshard_parts_one_launch_equals_the_per_part_launches():
    """RESHARD_ROWS multi-part form (ABI 8, round 6): the R packs in front of a frames -> pixels all-to-all and the R unpacks (+ residual)
    behind the way back as ONE launch each, the rank's own part going straight to / coming straight from its place in the pixel-sharded
    tensor — bit-equal to the R single-part launches they replace, on the GPU and in the CPU interpreter."""
    R, own, Fl, hw, C, offset, Ft = 4, 1, 3, 16, 64, 3, 11
    hwr = hw // R
    outs = []
    for multi in (True, False):
        P = Program()
        g = _g(43)
        n = P.alloc(Fl * hw, C, "f16")
        stage, xp = P.alloc(R * Fl * hwr, C, "f16"), P.alloc(Ft * hwr, C, "f16")
        yp, back = P.alloc(Ft * hwr, C, "f32"), P.alloc(R * Fl * hwr, C, "f32")
        x, out = P.alloc(Fl * hw, C, "f32", ld=C + 4), P.alloc(Fl * hw, C, "f32")
        xp_own, yp_own = xp.row_slice(offset * hwr, (offset + Fl) * hwr), yp.row_slice(offset * hwr, (offset + Fl) * hwr)
        if multi:
            P.reshard_parts("pack", n, stage, parts=R, rows=Fl * hwr, chunk=hwr, s_src=hw, s_dst=hwr, part_rows_src=hwr, part_rows_dst=Fl * hwr,
                            own=own, own_other=xp_own, own_is_src=False)
            P.reshard_parts("unpack", back, out, parts=R, rows=Fl * hwr, chunk=hwr, s_src=hwr, s_dst=hw, part_rows_src=Fl * hwr, part_rows_dst=hwr,
                            own=own, own_other=yp_own, own_is_src=True, residual=x)
            assert len(P.ops) == 2
        else:
            for q in range(R):
                P.reshard_rows(f"pack{q}", n.row_slice(q * hwr, n.rows), xp_own if q == own else stage.row_slice(q * Fl * hwr, (q + 1) * Fl * hwr),
                               rows=Fl * hwr, chunk=hwr, s_src=hw, s_dst=hwr)
            for q in range(R):
                P.reshard_rows(f"unpack{q}", yp_own if q == own else back.row_slice(q * Fl * hwr, (q + 1) * Fl * hwr), out.row_slice(q * hwr, out.rows),
                               rows=Fl * hwr, chunk=hwr, s_src=hwr, s_dst=hw, residual=x.row_slice(q * hwr, x.rows))

        def init(it):
            for b in (n, yp, back, x):
                fill(it, b, g)
            for b in (stage, xp, out):
                it.mat(b.ref, b.rows, b.cols, b.ld, TD[b.dtype], {}).zero_()
        it, got, _, _ = run_both(P, {}, {}, init)
        for b in (stage, xp, out):
            assert torch.equal(read(it, b), read(got, b))
        outs.append([read(got, b).clone() for b in (stage, xp, out)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][0][own * Fl * hwr:(own + 1) * Fl * hwr].abs().sum() == 0          # the own part never touched the staging buffer


@pytest.mark.parametrize("n_inst,rows,C,dt,silu", [(2, 24 * 1024, 320, "f32", True),      # 32x32 level, cross-frame: 256 workgroups, 16 rows / thread
                                                   (48, 1024, 320, "f16", True),        # 32x32 level, per frame: 5 chunks per instance, 20 rows / thread
                                                   (2, 24 * 256, 640, "f16", False),    # 16x16 level, cross-frame
                                                   (2, 24 * 64, 1280, "f32", True),     # 8x8 level
                                                   (1, 24 * 1024, 320, "f32", True),    # one CFG role (b = 1)
                                                   (5, 77, 64, "f32", False)])          # ragged chunks, tiny C
def test_groupnorm_cooperative_full_size(n_inst, rows, C, dt, silu):
    """The single-pass GroupNorm at the sizes of the UNet's 32x32 .. 8x8 levels (grids up to one workgroup per CU): against
    torch.nn.functional.group_norm, against the three-launch kernels (same statistics to ~1e-7), and run back to back on the same
    barrier words (the generation counter keeps growing, the arrival counter returns to zero) with bit-identical results."""
    import torch.nn.functional as F
    g = _g(31)
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "b": 0.1 * torch.randn(C, generator=g)}
    xs = torch.randn(n_inst * rows, C, generator=g) * 1.7 + 0.4
    outs = {}
    for coop in (True, False):
        P = Program()
        P.gn_coop = coop
        P.gn_fused_slice_bytes = 0
        x = P.alloc(n_inst * rows, C, dt)
        o = [P.alloc(n_inst * rows, C, "f16") for _ in range(3)]
        for k in range(3):
            P.groupnorm(f"gn{k}", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), o[k], n_inst=n_inst, eps=1e-5, silu=silu)
        dev = torch.device("cuda:0")
        arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
        td = torch.float16 if dt == "f16" else torch.float32
        arena[x.ref.off: x.ref.off + xs.numel() * (2 if dt == "f16" else 4)].view(td).copy_(xs.to(td).reshape(-1))
        wg = {k: v.to(dev) for k, v in w.items()}
        from sd_webui_text2video_amd.program import BoundProgram
        bp = BoundProgram(P, arena.data_ptr(), {k: v.data_ptr() for k, v in wg.items()})
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(2):
            bp.run({}, st)
        torch.cuda.synchronize()
        got = [arena[b.ref.off: b.ref.off + b.rows * C * 2].view(torch.float16).view(b.rows, C).float().cpu() for b in o]
        assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])
        outs[coop] = got[0]
    xin = xs.to(torch.float16 if dt == "f16" else torch.float32).float().view(n_inst, rows, C).permute(0, 2, 1)
    want = F.group_norm(xin, 32, w["g"], w["b"], 1e-5)
    want = (F.silu(want) if silu else want).permute(0, 2, 1).reshape(n_inst * rows, C)
    assert rel_l2(outs[True], want) < 6e-4 and rel_l2(outs[False], want) < 6e-4
    assert rel_l2(outs[True], outs[False]) < 2e-4        # both round the same fp32 values to fp16: differences are 1-ulp flips


@pytest.mark.parametrize("samples,F,hw,heads,K", [(2, 24, 64, 5, 320), (1, 16, 20, 2, 128), (1, 32, 9, 1, 64), (2, 5, 7, 3, 192),
                                                  (1, 24, 1024, 5, 320), (2, 2, 16, 2, 128)])
def test_fused_qkv_temporal_attention(samples, F, hw, heads, K):
    """T2V_EPI_TATTN (tile 10): QKV projection + temporal self-attention of every pixel's frame sequence in one launch, against
    the interpreter (fp32-accumulated projection rounded to fp16, then softmax(q k^T scale) v) and against the unfused pair of
    ops (QKV GEMM + attention kernel) this replaces.  Ragged pixel counts (hw % pixels-per-tile != 0), F = 2 .. 32."""
    g = _g(41)
    C = heads * 64
    T = samples * F * hw
    wq, wk, wv = [(torch.randn(C, K, generator=g) / math.sqrt(K)) for _ in range(3)]
    w = {"wh": pk.qkv_head_major(wq, wk, wv).half(), "wf": torch.cat([wq, wk, wv], 0).half()}
    P = Program()
    a = P.alloc(T, K, "f16")
    o_fused, o_ref = P.alloc(T, C, "f16"), P.alloc(T, C, "f16")
    qkv = P.alloc(T, 3 * C, "f16")
    scale = 64 ** -0.5
    op = P.qkv_temporal_attention("tattn", a, Ref("weight", 0, "wh"), o_fused, samples=samples, frames=F, hw=hw, heads=heads, k=K, scale=scale)
    assert op.i[22] == 10 and op.i[16] == L.EPI_TATTN and op.i[10] == min(12, 192 // F)
    P.gemm("qkv", a, Ref("weight", 0, "wf"), 3 * C, K, qkv)
    q, k, v = qkv.col_slice(0, C), qkv.col_slice(C, 2 * C), qkv.col_slice(2 * C, 3 * C)
    ld = 3 * C
    P.attention("attn", q.ref, k.ref, v.ref, o_ref.ref, out_buf=o_ref, nq=F, nk=F, heads=heads, b_outer=samples, b_inner=hw,
                q_strides=(hw * ld, F * hw * ld, ld), kv_strides=(hw * ld, F * hw * ld, ld), o_strides=(hw * C, F * hw * C, C), scale=scale)
    it, got, _, _ = run_both(P, w, {}, lambda it: fill(it, a, g, scale=1.5))
    _check(it, got, o_fused, 1.5e-3, "fused QKV + temporal attention vs interpreter")
    r = rel_l2(read(got, o_fused).float(), read(got, o_ref).float())
    assert r < 1.5e-3, f"fused vs the GEMM + attention kernel pair: {r:.3e}"


def test_split_k_ticket_fold_is_bitwise_the_reduction_kernel():
    """Split-K with the fold in the last-arriving workgroup of each tile (T2V_SPLITK_TICKETS=1 / Program.splitk_tickets; off by
    default: measured slower than the reduction launch) against the stand-alone reduction kernel: same slabs summed in the same
    order -> identical bits, for every
    kernel family and epilogue variant; repeated runs re-arm the tickets."""
    from sd_webui_text2video_amd.program import BoundProgram
    dev = torch.device("cuda:0")
    g = _g(77)
    cases = [dict(M=768, N=1280, K=3840, gather=L.GATHER_TCONV3, tile=5, cus=256, out="f16", res=False),
             dict(M=768, N=1280, K=11520, gather=L.GATHER_CONV3X3, tile=3, cus=256, out="f32", res=True),
             dict(M=768, N=1280, K=1280, gather=L.GATHER_PLAIN, tile=0, cus=256, out="f32", res=True),
             dict(M=3072, N=1280, K=11520, gather=L.GATHER_CONV3X3, tile=2, cus=256, out="f16", res=False),
             dict(M=200, N=324, K=2560, gather=L.GATHER_PLAIN, tile=0, cus=64, out="f16", res=True)]      # ragged M / N tails
    for c in cases:
        M, N, K = c["M"], c["N"], c["K"]
        results = []
        for tickets in (True, False):
            P = Program()
            P.force_tile, P.target_cus, P.splitk_tickets = c["tile"], c["cus"], tickets
            cin = K // (9 if c["gather"] == L.GATHER_CONV3X3 else 3 if c["gather"] == L.GATHER_TCONV3 else 1)
            a = P.alloc(M, cin, "f16")
            out = P.alloc(M, N, c["out"])
            res = P.alloc(M, N, "f32") if c["res"] else None
            conv = {}
            if c["gather"] == L.GATHER_CONV3X3:
                hh = 16 if M % 256 == 0 else 8
                conv = dict(Hin=hh, Win=hh, Cin=cin, stride=1, up=0, Hout=hh, Wout=hh)
                assert M % (hh * hh) == 0
            elif c["gather"] == L.GATHER_TCONV3:
                conv = dict(F=6, HW=M // 12, Cin=cin)
            op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), gather=c["gather"], conv=conv, residual=res)
            assert op.i[19] > 1 and (op.p[7].space != "null") == tickets
            arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
            gg = torch.Generator().manual_seed(5)
            arena[a.ref.off: a.ref.off + M * cin * 2].view(torch.float16).copy_(torch.randn(M * cin, generator=gg).half())
            if res is not None:
                arena[res.ref.off: res.ref.off + M * N * 4].view(torch.float32).copy_(torch.randn(M * N, generator=gg))
            w = {"w": (torch.randn(N, K, generator=gg) / math.sqrt(K)).half().to(dev), "b": torch.randn(N, generator=gg).to(dev)}
            bp = BoundProgram(P, arena.data_ptr(), {k: v.data_ptr() for k, v in w.items()})
            st = torch.cuda.current_stream(dev).cuda_stream
            runs = []
            for _ in range(3):
                bp.run({}, st)
                torch.cuda.synchronize()
                nb = M * N * (2 if c["out"] == "f16" else 4)
                runs.append(arena[out.ref.off: out.ref.off + nb].clone())
            assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), c
            if tickets:
                sync = P._sync
                assert int(arena[sync.ref.off: sync.ref.off + 4 * L.SYNC_INTS].view(torch.int32).abs().sum()) == 0      # all tickets re-armed
            results.append(runs[0])
        assert torch.equal(results[0], results[1]), c
        assert not torch.isnan(results[0].view(torch.float16 if c["out"] == "f16" else torch.float32)).any()


# ------------------------------------------------------------------------------------------------------------------
# Independent op-level checks: the HIP kernels against torch.nn.functional / explicit torch formulas (NOT the interpreter
# of tests/interp.py, which is this repository's own restatement of each op)
# ------------------------------------------------------------------------------------------------------------------
def _gpu_run(P, w, init):
    it, got, _, _ = run_both(P, w, {}, init)
    return it, got


@pytest.mark.parametrize("kind,B,F,hw,heads,D,Lc", [("spatial", 1, 2, 1024, 5, 64, 0), ("spatial", 1, 2, 80, 8, 40, 0),
                                                    ("cross", 2, 3, 64, 2, 64, 77), ("temporal", 2, 24, 16, 3, 64, 0),
                                                    ("temporal", 1, 125, 4, 2, 64, 0), ("spatial", 1, 1, 144, 3, 160, 0)])
def test_attention_against_torch_sdpa(kind, B, F, hw, heads, D, Lc):
    inner = heads * D
    M = B * F * hw
    P = Program()
    g = _g(110)
    scale = D ** -0.5
    if kind == "cross":
        q, kv, o = P.alloc(M, inner, "f16"), P.alloc(B * Lc, 2 * inner, "f16"), P.alloc(M, inner, "f16")
        k, v = kv.col_slice(0, inner), kv.col_slice(inner, 2 * inner)
        P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=Lc, heads=heads, b_outer=B, b_inner=F,
                    q_strides=(inner, F * hw * inner, hw * inner), kv_strides=(kv.ld, Lc * kv.ld, 0),
                    o_strides=(inner, F * hw * inner, hw * inner), scale=scale, head_dim=D)
        bufs = [q, kv]
    else:
        qkv, o = P.alloc(M, 3 * inner, "f16"), P.alloc(M, inner, "f16")
        ld = 3 * inner
        q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
        if kind == "spatial":
            P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=hw, heads=heads, b_outer=B * F, b_inner=1,
                        q_strides=(ld, hw * ld, 0), kv_strides=(ld, hw * ld, 0), o_strides=(inner, hw * inner, 0), scale=scale, head_dim=D)
        else:
            P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=F, nk=F, heads=heads, b_outer=B, b_inner=hw,
                        q_strides=(hw * ld, F * hw * ld, ld), kv_strides=(hw * ld, F * hw * ld, ld),
                        o_strides=(hw * inner, F * hw * inner, inner), scale=scale, head_dim=D)
        bufs = [qkv]
    it, got = _gpu_run(P, {}, lambda it: [fill(it, b, g, 1.2) for b in bufs])
    qf, kf, vf = read(it, q).float(), read(it, k).float(), read(it, v).float()
    if kind == "spatial":        # '(b f) n (h d)'
        sh = lambda t: t.view(B * F, hw, heads, D).transpose(1, 2)
        ref = torch.nn.functional.scaled_dot_product_attention(sh(qf), sh(kf), sh(vf)).transpose(1, 2).reshape(M, inner)
    elif kind == "temporal":     # rows (b f hw): sequences over f for every (b, pixel)
        sh = lambda t: t.view(B, F, hw, heads, D).permute(0, 2, 3, 1, 4)
        ref = torch.nn.functional.scaled_dot_product_attention(sh(qf), sh(kf), sh(vf)).permute(0, 3, 1, 2, 4).reshape(M, inner)
    else:                        # text keys of sample b shared by its F frames
        qs = qf.view(B, F, hw, heads, D).permute(0, 1, 3, 2, 4)
        ks = kf.view(B, 1, Lc, heads, D).permute(0, 1, 3, 2, 4).expand(B, F, heads, Lc, D)
        vs = vf.view(B, 1, Lc, heads, D).permute(0, 1, 3, 2, 4).expand(B, F, heads, Lc, D)
        ref = torch.nn.functional.scaled_dot_product_attention(qs, ks, vs).permute(0, 1, 3, 2, 4).reshape(M, inner)
    assert rel_l2(read(got, o).float(), ref) < 2e-3, kind


@pytest.mark.parametrize("hw,waves,lo", [(1024, 8, False), (1024, 4, False), (1000, 8, True), (520, 0, False), (2304, 8, False)])
def test_spatial_attention_with_lds_dma_tiles_against_torch_sdpa(hw, waves, lo):
    """Round 6 (csrc/attention.hip attn2_kernel): V transposed once per launch into a scratch, K / V^T tiles staged by LDS-DMA, 8 (4) waves
    share a 64-key tile.  Against torch SDPA and — same scores, same order — BITWISE against attn_kernel; ragged sequences (1000, 520 keys:
    partial last tile, K rows past the sequence read the zero page, V^T zero-padded), the hi + lo output form."""
    B, heads, D = 3, 5, 64
    inner, M = heads * D, B * hw
    g = _g(130)
    outs = []
    for new in (True, False):
        P = Program()
        qkv = P.alloc(M, 3 * inner, "f16")
        o = P.alloc(M, 2 * inner if lo else inner, "f16")
        ld = 3 * inner
        q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
        vt = P.alloc(B * heads * 64, -(-hw // 64) * 64, "f16") if new else None
        P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=hw, nk=hw, heads=heads, b_outer=B, b_inner=1, q_strides=(ld, hw * ld, 0),
                    kv_strides=(ld, hw * ld, 0), o_strides=(o.ld, hw * o.ld, 0), scale=D ** -0.5, head_dim=D, lo_off=inner if lo else 0,
                    vt_scratch=vt, waves=waves)
        assert (P.ops[0].p[6].space != "null") == new
        gg = _g(131)
        it, got = _gpu_run(P, {}, lambda it: fill(it, qkv, gg, 1.2))
        outs.append(read(got, o).clone())
        if new:
            sh = lambda t: read(it, t).float().view(B, hw, heads, D).transpose(1, 2)
            ref = torch.nn.functional.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(M, inner)
            hi = outs[0][:, :inner].float()
            assert rel_l2(hi, ref) < 2e-3
            if lo:
                assert rel_l2(hi + outs[0][:, inner:].float(), ref) < 3e-4      # (what is left is the fp16 P / V arithmetic: 2.0e-4; hi alone 2.9e-4)
    assert torch.equal(outs[0], outs[1]), "attn2_kernel and attn_kernel must give the same bits"


@pytest.mark.parametrize("n_inst,rows,C,dt,silu", [(6, 1024, 320, "f32", True), (2, 3 * 256, 640, "f16", True), (4, 64, 1280, "f32", False),
                                                  (2, 24 * 16, 1280, "f16", False), (3, 256, 128, "f32", True)])
def test_groupnorm_against_torch_functional(n_inst, rows, C, dt, silu):
    P = Program()
    g = _g(120)
    x, out = P.alloc(n_inst * rows, C, dt), P.alloc(n_inst * rows, C, "f16")
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "b": 0.1 * torch.randn(C, generator=g)}
    P.groupnorm("gn", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), out, n_inst=n_inst, eps=1e-5, silu=silu)

    def init(it):
        v = fill(it, x, g, 1.7)
        v += 0.4
    it, got = _gpu_run(P, w, init)
    xf = read(it, x).float().view(n_inst, rows, C).permute(0, 2, 1)                 # [N, C, L]
    ref = torch.nn.functional.group_norm(xf, 32, w["g"], w["b"], 1e-5)
    if silu:
        ref = torch.nn.functional.silu(ref)
    assert rel_l2(read(got, out).float(), ref.permute(0, 2, 1).reshape(n_inst * rows, C)) < 1e-3


@pytest.mark.parametrize("M,C", [(1000, 320), (77, 1024), (513, 640), (64, 1280)])
def test_layernorm_and_softmax_against_torch_functional(M, C):
    P = Program()
    g = _g(130)
    x, out = P.alloc(M, C, "f32"), P.alloc(M, C, "f16")
    s_in, s_out = P.alloc(M, C, "f32"), P.alloc(M, C, "f16")
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "b": 0.1 * torch.randn(C, generator=g)}
    P.layernorm("ln", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), out)
    P.softmax("sm", s_in, s_out, 0.37)
    it, got = _gpu_run(P, w, lambda it: (fill(it, x, g, 3.0), fill(it, s_in, g, 4.0)))
    ref = torch.nn.functional.layer_norm(read(it, x), (C,), w["g"], w["b"], 1e-5)
    assert rel_l2(read(got, out).float(), ref) < 1e-3
    ref = torch.softmax(read(it, s_in) * 0.37, dim=1)
    assert rel_l2(read(got, s_out).float(), ref) < 1e-3


@pytest.mark.parametrize("D,T,Tq,off,R", [(40, 16, 16, 0, 16), (80, 16, 6, 5, 16), (160, 16, 4, 12, 16), (64, 24, 24, 0, 8), (40, 7, 3, 3, 2)])
def test_relpos_attention_against_explicit_formula(D, T, Tq, off, R):
    """attention_temporal.py:107-144 written out with torch (the reference's RelativePosition index: clamp(s - t, -R, R) + R),
    including the T-sharded form: Tq local queries = frames [off, off + Tq) of the T key frames."""
    heads, hw = 2, 6
    inner = heads * D
    P = Program()
    g = _g(140 + D)
    q, kv, o = P.alloc(Tq * hw, inner, "f16"), P.alloc(T * hw, 2 * inner, "f16"), P.alloc(Tq * hw, inner, "f16")
    w = {"ek": 0.3 * torch.randn(2 * R + 1, D, generator=g), "ev": 0.3 * torch.randn(2 * R + 1, D, generator=g)}
    scale = D ** -0.5
    P.attention("a", q.ref, kv.col_slice(0, inner).ref, kv.col_slice(inner, 2 * inner).ref, o.ref, nq=Tq, nk=T, heads=heads,
                b_outer=1, b_inner=hw, q_strides=(hw * inner, 0, inner), kv_strides=(hw * kv.ld, 0, kv.ld),
                o_strides=(hw * inner, 0, inner), scale=scale, head_dim=D, rel_k=Ref("weight", 0, "ek"), rel_v=Ref("weight", 0, "ev"),
                max_rel=R, q_offset=off)
    it, got = _gpu_run(P, w, lambda it: (fill(it, q, g, 1.0), fill(it, kv, g, 1.0)))
    qf = read(it, q).float().view(Tq, hw, heads, D).permute(1, 2, 0, 3)                     # [hw, h, Tq, D]
    kvf = read(it, kv).float().view(T, hw, 2, heads, D)
    kf, vf = kvf[:, :, 0].permute(1, 2, 0, 3), kvf[:, :, 1].permute(1, 2, 0, 3)             # [hw, h, T, D]
    idx = (torch.arange(T)[None, :] - (torch.arange(Tq)[:, None] + off)).clamp(-R, R) + R   # [Tq, T]
    sim = (torch.einsum("phtd,phsd->phts", qf, kf) + torch.einsum("phtd,tsd->phts", qf, w["ek"][idx])) * scale
    p = sim.softmax(dim=-1)
    ref = torch.einsum("phts,phsd->phtd", p, vf) + torch.einsum("phts,tsd->phtd", p, w["ev"][idx])
    assert rel_l2(read(got, o).float(), ref.permute(2, 0, 1, 3).reshape(Tq * hw, inner)) < 2e-3


@pytest.mark.parametrize("tile", [8, 11, 2])
@pytest.mark.parametrize("M,K,with_res", [(400, 320, True), (192 * 3 + 5, 1280, True), (77, 64, False), (4096, 320, True)])
def test_gemm_with_fused_layernorm_output(M, K, with_res, tile):
    """192x320 / 128x320 / 256x320 (round 6: a wave's two 32-row blocks one after the other) tile with whole rows (N == 320): the epilogue writes the fp32 stream AND LayerNorm(row) * gamma + beta (fp16) —
    checked against the interpreter and against torch.nn.functional.layer_norm of the device's own fp32 output."""
    N = 320
    P = Program()
    P.force_tile = tile
    g = _g(150 + M)
    a, out, n_out = P.alloc(M, K, "f16"), P.alloc(M, N, "f32"), P.alloc(M, N, "f16", ld=N + 8)
    res = P.alloc(M, N, "f32") if with_res else None
    gamma, beta = 1 + 0.2 * torch.randn(N, generator=g), 0.2 * torch.randn(N, generator=g)
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g), "g": gamma, "be": beta,
         "gb": torch.cat([gamma, beta])}
    op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), residual=res, allow_splitk=False,
                ln=(Ref("weight", 0, "gb"), Ref("weight", 0, "g"), Ref("weight", 0, "be"), n_out, 1e-5))
    assert op.i[22] == tile and op.i[8] == 1 and len(P.ops) == 1          # fused: no separate LayerNorm op

    def init(it):
        fill(it, a, g)
        if with_res:
            fill(it, res, g, 2.0)
    it, got, _, _ = run_both(P, w, {}, init)
    _check(it, got, out, 2e-5, "gemm + fused LN: fp32 stream")
    _check(it, got, n_out, 1e-3, "gemm + fused LN: fp16 LayerNorm output")
    ref = torch.nn.functional.layer_norm(read(got, out), (N,), gamma, beta, 1e-5)
    assert rel_l2(read(got, n_out).float(), ref) < 1e-3


# ------------------------------------------------------------------------------------------------------------------
# Round 4: every GEMM epilogue and every gather mode against torch.nn.functional — not the interpreter —
# on BASELINE shapes, plus the hi + lo operand forms of `precise_operands`.
# ------------------------------------------------------------------------------------------------------------------
def _torch_linear(it, a, w, bias=None):
    return torch.nn.functional.linear(read(it, a).float(), w.float(), bias)


@pytest.mark.parametrize("M,C", [(49152 // 8, 320), (3072, 1280)])
def test_geglu_epilogue_against_torch_unpermuted_definition(M, C):
    """(49152, 2560, 320) / (3072, 10240, 1280) of SURVEY App. D (the 32x32 shape at 1/8 of its rows): GEGLU.forward
    (t2v_model.py:817-821) `x, gate = proj(x).chunk(2); x * gelu(gate)` on the UNPERMUTED nn.Linear weights."""
    P = Program()
    g = _g(201)
    a, out = P.alloc(M, C, "f16"), P.alloc(M, 4 * C, "f16")
    wsrc, bsrc = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).half(), torch.randn(8 * C, generator=g) * 0.2
    perm = pk.geglu_perm(4 * C)
    P.gemm("g", a, Ref("weight", 0, "w"), 8 * C, C, out, bias=Ref("weight", 0, "b"), epi=L.EPI_GEGLU)
    it, got = _gpu_run(P, {"w": wsrc[perm].contiguous(), "b": bsrc[perm].contiguous()}, lambda it: fill(it, a, g))
    hg = _torch_linear(it, a, wsrc, bsrc)
    ref = hg[:, :4 * C] * torch.nn.functional.gelu(hg[:, 4 * C:])
    assert rel_l2(read(got, out).float(), ref) < 1e-3


@pytest.mark.parametrize("tile", [0, 2, 8, 5])
def test_rowbias_silu_residual_epilogue_against_torch(tile):
    """bias + per-sample row bias (the time-embedding projection of a ResBlock, t2v_model.py:941-947) + SiLU + fp32 residual."""
    M, N, K, rpb = 1536, 320, 640, 384
    P = Program()
    P.force_tile = tile
    g = _g(202)
    a, out, res, rb = P.alloc(M, K, "f16"), P.alloc(M, N, "f32"), P.alloc(M, N, "f32"), P.alloc(M // rpb, N, "f32")
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g)}
    P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), act=1, residual=res, rowbias=rb, rows_per_batch=rpb,
           allow_splitk=False)
    it, got = _gpu_run(P, w, lambda it: (fill(it, a, g), fill(it, res, g, 2.0), fill(it, rb, g)))
    ref = torch.nn.functional.silu(_torch_linear(it, a, w["w"], w["b"]) + read(it, rb).repeat_interleave(rpb, dim=0)) + read(it, res)
    assert rel_l2(read(got, out), ref) < 2e-5


@pytest.mark.parametrize("tickets", [False, True])
def test_split_k_against_torch(tickets):
    """(768, 1280, 11520, split 8): the 4x4-level 3x3 convolution's GEMM shape, as a plain GEMM with bias + fp32 residual."""
    M, N, K = 768, 1280, 11520
    P = Program()
    P.splitk_tickets = tickets
    g = _g(203)
    a, out, res = P.alloc(M, K, "f16"), P.alloc(M, N, "f32"), P.alloc(M, N, "f32")
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g)}
    op = P.gemm("g", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), residual=res)
    assert op.i[19] > 1
    it, got = _gpu_run(P, w, lambda it: (fill(it, a, g), fill(it, res, g)))
    ref = _torch_linear(it, a, w["w"], w["b"]) + read(it, res)
    assert rel_l2(read(got, out), ref) < 2e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up,pad_after", [(4, 32, 32, 320, 320, 1, 0, False), (2, 16, 16, 640, 640, 2, 0, False),
                                                                (2, 8, 8, 1280, 1280, 1, 1, False), (2, 16, 16, 128, 128, 2, 0, True)])
def test_conv3x3_gather_against_torch_conv2d(B, H, W, Cin, Cout, stride, up, pad_after):
    """Every 3x3 gather mode against F.conv2d on the NCHW view of the same tokens: stride 1 / 2, the nearest-2x upsample folded into
    the gather (t2v_model.py:1044-1050), and the encoder's (0,1,0,1) padding (autoencoder_modules.py Downsample)."""
    Ho, Wo = (2 * H, 2 * W) if up else (((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W))
    if pad_after:
        Ho, Wo = H // 2, W // 2
    P = Program()
    g = _g(204)
    a, out = P.alloc(B * H * W, Cin, "f16"), P.alloc(B * Ho * Wo, Cout, "f32")
    w4 = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).half()
    w = {"w": pk.pad_rows(pk.conv3x3(w4.float())).half(), "b": torch.randn(Cout, generator=g)}
    P.gemm("c", a, Ref("weight", 0, "w"), Cout, 9 * Cin, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
           conv=dict(Hin=H, Win=W, Cin=Cin, stride=stride, up=up, Hout=Ho, Wout=Wo, pad_after_only=pad_after))
    it, got = _gpu_run(P, w, lambda it: fill(it, a, g))
    x = read(it, a).float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    if up:
        x = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
    if pad_after:
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (0, 1, 0, 1)), w4.float(), w["b"], stride=2)
    else:
        ref = torch.nn.functional.conv2d(x, w4.float(), w["b"], stride=stride, padding=1)
    assert rel_l2(read(got, out), ref.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)) < 2e-5


@pytest.mark.parametrize("B,F,HW,C,halo", [(2, 24, 64, 640, False), (1, 6, 256, 320, True)])
def test_temporal_conv_gather_against_torch_conv3d(B, F, HW, C, halo):
    """(3,1,1) temporal convolution (t2v_model.py:1202-1211) against F.conv3d; `halo`: the T-sharded input layout [F + 2] frames."""
    P = Program()
    g = _g(205)
    rows_in = B * (F + 2) * HW if halo else B * F * HW
    a, out = P.alloc(rows_in, C, "f16"), P.alloc(B * F * HW, C, "f32")
    w5 = (torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)).half()
    w = {"w": pk.tconv3(w5.float()).half(), "b": torch.randn(C, generator=g)}
    P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * C, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3, conv=dict(F=F, HW=HW, Cin=C), halo=halo)
    it, got = _gpu_run(P, w, lambda it: fill(it, a, g))
    x = read(it, a).float().view(B, F + 2 if halo else F, HW, C).permute(0, 3, 1, 2).unsqueeze(-1)      # [B, C, F(+2), HW, 1]
    ref = torch.nn.functional.conv3d(x, w5.float(), w["b"], padding=(0, 0, 0) if halo else (1, 0, 0))
    assert rel_l2(read(got, out), ref.squeeze(-1).permute(0, 2, 3, 1).reshape(B * F * HW, C)) < 2e-5


def test_c8_stem_gather_with_lo_channels_against_torch_conv2d():
    """The stem (Cin = 8 = 4 latent channels + their 4 low-order images, weights repeated): (hi + lo) . W in one pass equals the
    fp32 convolution of the UNROUNDED latent to ~1e-6 — what `precise_operands` buys at the entry."""
    B, F, H, W, Cout = 1, 3, 16, 16, 320
    P = Program()
    g = _g(206)
    xin, out = P.alloc(B * F * H * W, 8, "f16"), P.alloc(B * F * H * W, Cout, "f32")
    x5 = torch.randn(B, 4, F, H, W, generator=g)
    w4 = (torch.randn(Cout, 4, 3, 3, generator=g) / 6.0).half()
    w = {"w": pk.pad_rows(pk.conv3x3_c8_dup(w4.float())).half(), "b": torch.randn(Cout, generator=g)}
    P.ncthw_to_cl("in", Ref("ext", L.EXT_X), "f32", xin, B=B, C=4, F=F, HW=H * W, lo_in_pad=True)
    P.gemm("stem", xin, Ref("weight", 0, "w"), Cout, 72, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3_C8,
           conv=dict(Hin=H, Win=W, Cin=8, stride=1, up=0, Hout=H, Wout=W))
    it, got, _, _ = run_both(P, w, {L.EXT_X: x5}, lambda it: None)
    ref = torch.nn.functional.conv2d(x5.permute(0, 2, 1, 3, 4).reshape(B * F, 4, H, W), w4.float(), w["b"], padding=1)
    r = rel_l2(read(got, out), ref.permute(0, 2, 3, 1).reshape(-1, Cout))
    assert r < 2e-5, r          # (an fp16-rounded latent alone would give ~2e-4)


@pytest.mark.parametrize("variant", ["three_launch", "cooperative", "single_launch"])
def test_groupnorm_lo_output_and_dup_linear_against_torch(variant):
    """precise_operands, round 4: GroupNorm writes rows [fp16(y) | fp16(y - fp16(y))], the consumer linear runs on K = 2C against
    [W | W] — together they must reproduce linear(group_norm(x)) of torch in fp32 to ~1e-6 relative (fp16 operand alone: ~3e-4)."""
    n_inst, rows, C, N = 4, 256, 320, 320
    P = Program()
    P.gn_coop = variant == "cooperative"
    P.gn_fused_slice_bytes = 1 << 30 if variant == "single_launch" else 0
    P.gn_fused_total_bytes = 1 << 30
    if variant == "single_launch":
        C = N = 640                                  # (C / groups) % 4 == 0
    g = _g(207)
    x, nrm, out = P.alloc(n_inst * rows, C, "f32"), P.alloc(n_inst * rows, 2 * C, "f16"), P.alloc(n_inst * rows, N, "f32")
    wl = (torch.randn(N, C, generator=g) / math.sqrt(C)).half()
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "b": 0.1 * torch.randn(C, generator=g), "w2": pk.linear_dup(wl.float()).half()}
    P.groupnorm("gn", x, Ref("weight", 0, "g"), Ref("weight", 0, "b"), nrm.col_slice(0, C), n_inst=n_inst, eps=1e-6, silu=False, lo=True)
    P.gemm("lin", nrm, Ref("weight", 0, "w2"), N, 2 * C, out)
    gn = [op for op in P.ops if op.kind == L.OP_GROUPNORM][0]
    assert gn.i[16] == 1 and gn.i[12] == (variant == "single_launch") and gn.i[15] == (variant == "cooperative")

    def init(it):
        v = fill(it, x, g, 1.5)
        v += 0.3
    it, got = _gpu_run(P, w, init)
    _check(it, got, nrm, 2e-3, "GroupNorm hi | lo images vs the interpreter")
    xf = read(it, x).view(n_inst, rows, C).permute(0, 2, 1)
    y = torch.nn.functional.group_norm(xf, 32, w["g"], w["b"], 1e-6).permute(0, 2, 1).reshape(n_inst * rows, C)
    hi_lo = read(got, nrm).float()
    assert rel_l2(hi_lo[:, :C] + hi_lo[:, C:], y) < 1e-5          # hi + lo carries the value to ~22 bits (fp16 alone: ~3e-4)
    r = rel_l2(read(got, out), torch.nn.functional.linear(y, wl.float()))
    assert r < 2e-5, r


@pytest.mark.parametrize("M,N,K,tile", [(1536, 320, 1280, None), (768, 1280, 5120, None), (400, 640, 2560, 0)])
def test_gemm_hi_lo_fp16_output_against_torch(M, N, K, tile):
    """precise_operands, round 4: the feed-forward output x4 = x3 + FF as rows [fp16(v) | fp16(v - fp16(v))] (also through the
    split-K reduction kernel at the 4x4-level shape), then proj_out on K = 2N against [W | W]: equals torch's fp32 chain."""
    P = Program()
    P.force_tile = tile
    g = _g(208)
    a, x4, res, out = P.alloc(M, K, "f16"), P.alloc(M, 2 * N, "f16"), P.alloc(M, N, "f32"), P.alloc(M, N, "f32")
    w1, wp = (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), (torch.randn(N, N, generator=g) / math.sqrt(N)).half()
    w = {"w1": w1, "b1": torch.randn(N, generator=g), "wp": pk.linear_dup(wp.float()).half()}
    P.gemm("ff2", a, Ref("weight", 0, "w1"), N, K, x4.col_slice(0, N), bias=Ref("weight", 0, "b1"), residual=res, out_lo=True)
    P.gemm("proj_out", x4, Ref("weight", 0, "wp"), N, 2 * N, out)
    it, got = _gpu_run(P, w, lambda it: (fill(it, a, g), fill(it, res, g, 2.0)))
    v = _torch_linear(it, a, w1, w["b1"]) + read(it, res)
    hi_lo = read(got, x4).float()
    assert rel_l2(hi_lo[:, :N] + hi_lo[:, N:], v) < 1e-5
    r = rel_l2(read(got, out), torch.nn.functional.linear(v, wp.float()))
    assert r < 2e-5, r


@pytest.mark.parametrize("kind,tile", [("conv", None), ("tconv", None), ("plain", 0), ("plain", 8), ("plain", 2)])
@pytest.mark.parametrize("per_frame,out_dt", [(True, "f16"), (False, "f32")])
def test_groupnorm_from_producer_strips_against_torch(kind, tile, per_frame, out_dt):
    """Round 4: the GEMM that produces a GroupNorm's input leaves per-32-row-strip column sums / sums of squares of
    its STORED result (T2V_EPI_STATS), the GroupNorm folds them (phase 3) and normalises in one pass — no statistics pass, no grid
    barrier.  Checked against torch: conv / linear -> (fp16 rounding) -> group_norm -> SiLU."""
    B, F, H, W, C = 2, 3, 8, 8, 128
    M = B * F * H * W
    P = Program()
    P.force_tile = tile
    g = _g(300)
    res = P.alloc(M, C, "f32") if out_dt == "f32" else None
    y, st, out = P.alloc(M, C, out_dt), P.alloc(M // 32, 2 * C, "f32"), P.alloc(M, C, "f16")
    w = {"b": torch.randn(C, generator=g), "g": 1 + 0.1 * torch.randn(C, generator=g), "be": 0.1 * torch.randn(C, generator=g)}
    if kind == "conv":
        a = P.alloc(M, C, "f16")
        w4 = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).half()
        w["w"] = pk.conv3x3(w4.float()).half()
        op = P.gemm("c", a, Ref("weight", 0, "w"), C, 9 * C, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
                    conv=dict(Hin=H, Win=W, Cin=C, stride=1, up=0, Hout=H, Wout=W), residual=res, stats=st, allow_splitk=False)
    elif kind == "tconv":
        a = P.alloc(M, C, "f16")
        w5 = (torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)).half()
        w["w"] = pk.tconv3(w5.float()).half()
        op = P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * C, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3, conv=dict(F=F, HW=H * W, Cin=C),
                    residual=res, stats=st, allow_splitk=False)
    else:
        a = P.alloc(M, 2 * C, "f16")
        w["w"] = (torch.randn(C, 2 * C, generator=g) / math.sqrt(2 * C)).half()
        op = P.gemm("l", a, Ref("weight", 0, "w"), C, 2 * C, y, bias=Ref("weight", 0, "b"), residual=res, stats=st, allow_splitk=False)
    assert op.i[16] == L.EPI_STATS and op.meta["stats"] == 1
    gn = P.groupnorm("gn", y, Ref("weight", 0, "g"), Ref("weight", 0, "be"), out, n_inst=B * F if per_frame else B, eps=1e-5, silu=True, stats=st)
    assert gn.i[8] == 3 and len(P.ops) == 2
    it, got = _gpu_run(P, w, lambda it: (fill(it, a, g), res is not None and fill(it, res, g, 2.0)))
    _check(it, got, st, 2e-5, "strips vs the interpreter")
    _check(it, got, out, 1e-3, "GroupNorm from strips vs the interpreter")
    stored = read(got, y).float()                                      # the device's own GEMM result (fp16-rounded or fp32)
    s = read(got, st).view(M // 32, 2, C)
    assert rel_l2(s[:, 0], stored.view(M // 32, 32, C).sum(dim=1)) < 1e-5
    assert rel_l2(s[:, 1], (stored * stored).view(M // 32, 32, C).sum(dim=1)) < 1e-5
    n_inst = B * F if per_frame else B
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(stored.view(n_inst, M // n_inst, C).permute(0, 2, 1), 32, w["g"], w["be"], 1e-5))
    assert rel_l2(read(got, out).float(), ref.permute(0, 2, 1).reshape(M, C)) < 1e-3


def test_groupnorm_from_strips_many_strips_per_instance():
    """The workgroup-per-(instance, group) fold of phase 3 (instances of > 1024 strip x channel pairs: the cross-frame norms)."""
    n_inst, rows, C, K = 2, 1536, 1280, 256
    M = n_inst * rows
    P = Program()
    g = _g(301)
    a, y, st, out = P.alloc(M, K, "f16"), P.alloc(M, C, "f16"), P.alloc(M // 32, 2 * C, "f32"), P.alloc(M, C, "f16")
    w = {"w": (torch.randn(C, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(C, generator=g),
         "g": 1 + 0.1 * torch.randn(C, generator=g), "be": 0.1 * torch.randn(C, generator=g)}
    op = P.gemm("l", a, Ref("weight", 0, "w"), C, K, y, bias=Ref("weight", 0, "b"), stats=st, allow_splitk=False)
    assert op.meta["stats"] == 1
    P.groupnorm("gn", y, Ref("weight", 0, "g"), Ref("weight", 0, "be"), out, n_inst=n_inst, eps=1e-5, silu=True, stats=st)
    it, got = _gpu_run(P, w, lambda it: fill(it, a, g))
    stored = read(got, y).float()
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(stored.view(n_inst, rows, C).permute(0, 2, 1), 32, w["g"], w["be"], 1e-5))
    assert rel_l2(read(got, out).float(), ref.permute(0, 2, 1).reshape(M, C)) < 1e-3


@pytest.mark.parametrize("kind,D", [("spatial", 64), ("spatial", 40), ("cross", 80), ("relpos", 40), ("relpos", 160)])
def test_attention_lo_output_and_dup_to_out_against_torch(kind, D):
    """precise_operands (VideoCrafter default): the attention kernels also store the low-order fp16 image of every output value
    (rows [hi | lo]), to_out runs on K = 2C against [W | W]: hi + lo reproduces the fp32 attention output to ~22 bits."""
    heads, B, F, hw, Lc, R = 2, 1, 4, 48, 11, 16
    inner = heads * D
    M = B * F * hw
    P = Program()
    g = _g(400 + D)
    scale = D ** -0.5
    qkv, a, out = P.alloc(M, 3 * inner, "f16"), P.alloc(M, 2 * inner, "f16"), P.alloc(M, inner, "f32")
    ld, lo = 3 * inner, 2 * inner
    q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
    wl = (torch.randn(inner, inner, generator=g) / math.sqrt(inner)).half()
    w = {"w2": pk.linear_dup(wl.float()).half(), "ek": 0.3 * torch.randn(2 * R + 1, D, generator=g), "ev": 0.3 * torch.randn(2 * R + 1, D, generator=g)}
    bufs = [qkv]
    if kind == "spatial":
        P.attention("a", q.ref, k.ref, v.ref, a.ref, nq=hw, nk=hw, heads=heads, b_outer=B * F, b_inner=1, q_strides=(ld, hw * ld, 0),
                    kv_strides=(ld, hw * ld, 0), o_strides=(lo, hw * lo, 0), scale=scale, head_dim=D, lo_off=inner)
    elif kind == "cross":
        kv = P.alloc(B * Lc, 2 * inner, "f16")
        bufs.append(kv)
        P.attention("a", q.ref, kv.col_slice(0, inner).ref, kv.col_slice(inner, 2 * inner).ref, a.ref, nq=hw, nk=Lc, heads=heads, b_outer=B,
                    b_inner=F, q_strides=(ld, F * hw * ld, hw * ld), kv_strides=(kv.ld, Lc * kv.ld, 0), o_strides=(lo, F * hw * lo, hw * lo),
                    scale=scale, head_dim=D, lo_off=inner)
    else:
        P.attention("a", q.ref, k.ref, v.ref, a.ref, nq=F, nk=F, heads=heads, b_outer=B, b_inner=hw, q_strides=(hw * ld, F * hw * ld, ld),
                    kv_strides=(hw * ld, F * hw * ld, ld), o_strides=(hw * lo, F * hw * lo, lo), scale=scale, head_dim=D,
                    rel_k=Ref("weight", 0, "ek"), rel_v=Ref("weight", 0, "ev"), max_rel=R, lo_off=inner)
    P.gemm("to_out", a, Ref("weight", 0, "w2"), inner, 2 * inner, out)
    it, got = _gpu_run(P, w, lambda it: [fill(it, b, g, 1.0) for b in bufs])
    _check(it, got, a, 3e-3, "attention hi | lo vs the interpreter")
    qf, kf, vf = read(it, q).float(), read(it, k).float(), read(it, v).float()
    if kind == "spatial":
        sh = lambda t: t.view(B * F, hw, heads, D).transpose(1, 2)
        ref = torch.nn.functional.scaled_dot_product_attention(sh(qf), sh(kf), sh(vf)).transpose(1, 2).reshape(M, inner)
    elif kind == "cross":
        kvf = read(it, bufs[1]).float()
        qs = qf.view(B, F, hw, heads, D).permute(0, 1, 3, 2, 4)
        ks = kvf[:, :inner].view(B, 1, Lc, heads, D).permute(0, 1, 3, 2, 4).expand(B, F, heads, Lc, D)
        vs = kvf[:, inner:].view(B, 1, Lc, heads, D).permute(0, 1, 3, 2, 4).expand(B, F, heads, Lc, D)
        ref = torch.nn.functional.scaled_dot_product_attention(qs, ks, vs).permute(0, 1, 3, 2, 4).reshape(M, inner)
    else:
        sh = lambda t: t.view(B, F, hw, heads, D).permute(0, 2, 3, 1, 4)                       # [B, hw, h, F, D]
        idx = (torch.arange(F)[None, :] - torch.arange(F)[:, None]).clamp(-R, R) + R
        sim = (torch.einsum("bphtd,bphsd->bphts", sh(qf), sh(kf)) + torch.einsum("bphtd,tsd->bphts", sh(qf), w["ek"][idx])) * scale
        pr = sim.softmax(dim=-1)
        o = torch.einsum("bphts,bphsd->bphtd", pr, sh(vf)) + torch.einsum("bphts,tsd->bphtd", pr, w["ev"][idx])
        ref = o.permute(0, 3, 1, 2, 4).reshape(M, inner)
    hi_lo = read(got, a).float()
    r_hi, r_sum = rel_l2(hi_lo[:, :inner], ref), rel_l2(hi_lo[:, :inner] + hi_lo[:, inner:], ref)
    assert r_hi < 2e-3 and r_sum < 0.5 * r_hi + 2e-4, (r_hi, r_sum)        # the sum is closer than the fp16 value (P itself is fp16 inside the kernels)
    want = torch.nn.functional.linear(hi_lo[:, :inner] + hi_lo[:, inner:], wl.float())
    assert rel_l2(read(got, out), want) < 2e-5


def test_conv3x3_on_hi_lo_channel_blocks_against_torch():
    """precise_operands (VideoCrafter default): the fp32 stream in front of a Down / Upsample convolution is cast to rows [hi | lo]
    (2 Cin channels) and the 3x3 convolution runs against [W | W]: equals F.conv2d on the UNROUNDED fp32 input to ~1e-5."""
    B, H, W, Cin, Cout = 2, 8, 8, 64, 128
    M = B * H * W
    P = Program()
    g = _g(410)
    x, x16, out = P.alloc(M, Cin, "f32"), P.alloc(M, 2 * Cin, "f16"), P.alloc(M // 4, Cout, "f32")
    w4 = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).half()
    w = {"w": pk.conv3x3(torch.cat([w4.float()] * 2, dim=1)).half(), "b": torch.randn(Cout, generator=g)}
    P.copy2d("cast", x, x16.col_slice(0, Cin), lo=x16.col_slice(Cin, 2 * Cin))
    P.gemm("down", x16, Ref("weight", 0, "w"), Cout, 9 * 2 * Cin, out, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
           conv=dict(Hin=H, Win=W, Cin=2 * Cin, stride=2, up=0, Hout=H // 2, Wout=W // 2), k_alg=9 * Cin)
    it, got = _gpu_run(P, w, lambda it: fill(it, x, g, 2.0))
    xin = read(it, x).view(B, H, W, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xin, w4.float(), w["b"], stride=2, padding=1).permute(0, 2, 3, 1).reshape(M // 4, Cout)
    r = rel_l2(read(got, out), ref)
    assert r < 2e-5, r


# ---- round 5: GroupNorm (+SiLU) inside the epilogue of the GEMM that produces its input (T2V_EPI_GN) -----------------------------------
@pytest.mark.parametrize("kind,tile,C", [("conv", 8, 320), ("tconv", 8, 320), ("plain", 8, 320), ("tconv", 11, 320), ("plain", 11, 320),
                                         ("tconv", 0, 640), ("conv", 0, 640), ("plain", 0, 640), ("tconv", 5, 640), ("plain", 5, 1280),
                                         ("conv", 3, 640), ("plain", 3, 640)])
@pytest.mark.parametrize("per_frame,dead,lo", [(True, True, False), (False, False, True), (False, True, False), (True, False, False)])
def test_groupnorm_in_producer_epilogue_against_torch(kind, tile, C, per_frame, dead, lo):
    """Round 5: the norm that consumes a convolution / linear result runs in that GEMM's epilogue — the tile stays
    in registers, {sum, sum of squares} per (instance, group piece) meet at a grid barrier, no GROUPNORM op, and a result only the norm
    reads (`dead`) is never stored.  Tiles of 192 / 128 rows against instances of 256 rows (a tile straddles two instances), groups cut
    by the 128- / 256-wide column tiles (C = 640: 20 channels per group), hi + lo output, residual + row bias.  Checked against the CPU
    interpreter and against torch: conv / linear -> group_norm -> SiLU on the fp32 result."""
    B, F, H, W = 2, 3, 16, 16
    M = B * F * H * W                                       # 1536 rows: 8 x 192, 12 x 128; a frame = 256 rows
    P = Program()
    P.force_tile = tile
    g = _g(500 + tile)
    y = P.alloc(M, C, "f16" if dead else "f32")
    res = P.alloc(M, C, "f32") if not dead else None
    rb = P.alloc(B, C, "f32") if kind == "conv" else None
    w = {"b": torch.randn(C, generator=g), "g": 1 + 0.1 * torch.randn(C, generator=g), "be": 0.1 * torch.randn(C, generator=g)}
    w["gb"] = torch.cat([w["g"], w["be"]])
    if kind == "conv":
        cin = 64
        a = P.alloc(M, cin, "f16")
        w4 = (torch.randn(C, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half()
        w["w"] = pk.conv3x3(w4.float()).half()
        op = P.gemm("c", a, Ref("weight", 0, "w"), C, 9 * cin, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
                    conv=dict(Hin=H, Win=W, Cin=cin, stride=1, up=0, Hout=H, Wout=W), residual=res, rowbias=rb, rows_per_batch=F * H * W,
                    allow_splitk=False)
    elif kind == "tconv":
        cin = 64
        a = P.alloc(M, cin, "f16")
        w5 = (torch.randn(C, cin, 3, 1, 1, generator=g) / math.sqrt(3 * cin)).half()
        w["w"] = pk.tconv3(w5.float()).half()
        op = P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * cin, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3, conv=dict(F=F, HW=H * W, Cin=cin),
                    residual=res, allow_splitk=False)
    else:
        a = P.alloc(M, 128, "f16")
        w["w"] = (torch.randn(C, 128, generator=g) / math.sqrt(128)).half()
        op = P.gemm("l", a, Ref("weight", 0, "w"), C, 128, y, bias=Ref("weight", 0, "b"), residual=res, allow_splitk=False)
    assert op.meta["tile"] == tile and op.meta["split"] == 1
    full = P.alloc(M, 2 * C if lo else C, "f16")
    out = full.col_slice(0, C)
    n_inst = B * F if per_frame else B
    fused = P.groupnorm("gn", y, Ref("weight", 0, "g"), Ref("weight", 0, "be"), out, n_inst=n_inst, eps=1e-5, silu=True, lo=lo,
                        gb=Ref("weight", 0, "gb"), x_dead=dead)
    assert fused is op and op.i[16] == L.EPI_GN and len(P.ops) == 1, "the norm did not become the GEMM's epilogue"

    def init(it):
        fill(it, a, g)
        if res is not None:
            fill(it, res, g, 2.0)
        if rb is not None:
            fill(it, rb, g)
        if dead:                                            # the GEMM must not touch a result nobody reads
            it.mat(y.ref, M, C, C, torch.float16, {}).fill_(7.0)
    it, got = _gpu_run(P, w, init)
    _check(it, got, full, 1e-3, "GroupNorm in the producer's epilogue vs the interpreter")
    if dead:
        assert bool((read(got, y) == 7.0).all()), "the dead result was stored"
    else:
        _check(it, got, y, 2e-5, "the fp32 stream beside the fused norm")
        v = read(got, y).float()
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(v.view(n_inst, M // n_inst, C).permute(0, 2, 1), 32, w["g"], w["be"], 1e-5))
        ref = ref.permute(0, 2, 1).reshape(M, C)
        hi = read(got, out).float()
        assert rel_l2(hi, ref) < 1e-3
        if lo:                                              # hi + lo reproduces the normalised value far below fp16 resolution
            assert rel_l2(hi + read(got, full.col_slice(C, 2 * C)).float(), ref) < 2e-5


def test_groupnorm_in_producer_epilogue_bench_shape_is_deterministic():
    """The 32x32-level shape of the bench workload (M = 49152 rows = 256 workgroups of 192 x 320, one per CU; cross-frame statistics over
    24576 rows = 128 tiles): two runs are bitwise equal (fixed fold order, no atomics on data) and match torch."""
    B, F, HW, C = 2, 24, 1024, 320
    M = B * F * HW
    P = Program()
    g = _g(77)
    a, y, out = P.alloc(M, C, "f16"), P.alloc(M, C, "f16"), P.alloc(M, C, "f16")
    w5 = (torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)).half()
    w = {"w": pk.tconv3(w5.float()).half(), "b": torch.randn(C, generator=g), "g": 1 + 0.1 * torch.randn(C, generator=g), "be": 0.1 * torch.randn(C, generator=g)}
    w["gb"] = torch.cat([w["g"], w["be"]])
    op = P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * C, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3, conv=dict(F=F, HW=HW, Cin=C))
    fused = P.groupnorm("gn", y, Ref("weight", 0, "g"), Ref("weight", 0, "be"), out, n_inst=B, eps=1e-5, silu=True, gb=Ref("weight", 0, "gb"), x_dead=True)
    assert fused is op and op.meta["tile"] == 8
    dev = torch.device("cuda:0")
    arena = torch.zeros(P.arena.high + 256, dtype=torch.uint8, device=dev)
    av = arena[a.ref.off: a.ref.off + M * C * 2].view(torch.float16).view(M, C)
    av.copy_(torch.randn(M, C, generator=g).half())
    wg = {k: v.to(dev).contiguous() for k, v in w.items()}
    bp = BoundProgram(P, arena.data_ptr(), {k: v.data_ptr() for k, v in wg.items()})
    st = torch.cuda.current_stream(dev).cuda_stream
    ov = arena[out.ref.off: out.ref.off + M * C * 2].view(torch.float16).view(M, C)
    bp.run({}, st); torch.cuda.synchronize()
    first = ov.clone()
    bp.run({}, st); torch.cuda.synchronize()
    assert torch.equal(first, ov)
    L.async_status()
    x = av.float().view(B, F, HW, C)
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 1, 1))
    conv = sum(xp[:, kt:kt + F] @ w5[:, :, kt, 0, 0].to(dev).float().t() for kt in range(3)) + wg["b"]
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(conv.view(B, F * HW, C).permute(0, 2, 1), 32, wg["g"], wg["be"], 1e-5))
    assert rel_l2(first.float().cpu(), ref.permute(0, 2, 1).reshape(M, C).cpu()) < 1e-3


@pytest.mark.parametrize("tile,N,K,M", [(0, 640, 640, 1536), (5, 1280, 1280, 1152), (12, 1280, 1280, 768), (9, 512, 512, 1152), (3, 768, 256, 1000)])
def test_layernorm_across_column_tiles_against_torch(tile, N, K, M):
    """Round 5: the LayerNorm that follows a C -> C linear whose rows are cut into several column tiles (C = 640 / 1280: the 16x16 / 8x8 /
    4x4 levels) runs in that GEMM's epilogue — per-row {sum, sum of squares} of every column tile meet at the grid barrier — instead of
    as its own launch.  Checked against the interpreter and torch.nn.functional.layer_norm; ragged M (1000 rows: a partial last tile)."""
    P = Program()
    P.force_tile = tile
    g = _g(600 + tile)
    a, res, out, ln_out = P.alloc(M, K, "f16"), P.alloc(M, N, "f32"), P.alloc(M, N, "f32"), P.alloc(M, N, "f16")
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g),
         "g": 1 + 0.1 * torch.randn(N, generator=g), "be": 0.1 * torch.randn(N, generator=g)}
    w["gb"] = torch.cat([w["g"], w["be"]])
    op = P.gemm("l", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), residual=res, allow_splitk=False,
                ln=(Ref("weight", 0, "gb"), Ref("weight", 0, "g"), Ref("weight", 0, "be"), ln_out, 1e-5))
    assert op.meta["tile"] == tile and op.i[8] == 2 and len(P.ops) == 1, "the LayerNorm did not become the GEMM's epilogue"
    it, got = _gpu_run(P, w, lambda it: (fill(it, a, g), fill(it, res, g, 3.0)))
    _check(it, got, out, 2e-5, "fp32 stream")
    _check(it, got, ln_out, 1e-3, "cross-tile LayerNorm vs the interpreter")
    ref = torch.nn.functional.layer_norm(read(got, out).float() + 0.0, (N,), w["g"], w["be"], 1e-5)
    assert rel_l2(read(got, ln_out).float(), ref) < 1e-3
    # rows with a large common offset: the single-pass variance (fp32 sums, fp64 combination) must survive mean^2 >> var
    it2, got2 = _gpu_run(P, w, lambda it: (fill(it, a, g), it.mat(res.ref, M, N, N, torch.float32, {}).copy_(torch.randn(M, N, generator=g) + 40.0)))
    ref2 = torch.nn.functional.layer_norm(read(got2, out).double(), (N,), w["g"].double(), w["be"].double(), 1e-5).float()
    assert rel_l2(read(got2, ln_out).float(), ref2) < 2e-3


# ---- round 6: fused norms on grids LARGER than the device holds co-resident: row chunks of whole tiles and whole instances ---------------
@pytest.mark.parametrize("kind,tile,C,HW,frames", [("conv", 8, 320, 1024, 66), ("tconv", 11, 320, 256, 160), ("plain", 0, 640, 256, 300),
                                                    ("conv", 3, 640, 256, 200), ("plain", 5, 1280, 64, 900)])
def test_fused_groupnorm_on_a_grid_larger_than_the_device_runs_in_row_chunks(kind, tile, C, HW, frames):
    """at 125 frames / 1024x576 a fused-norm GEMM has more tiles than compute units, and the statistics exchange
    needs the grid of ONE launch co-resident.  The launcher (t2v_launch_coresident) cuts the rows into chunks of whole tiles AND whole
    statistics instances (per-frame norms: lcm(tile rows, frame rows)), one launch each, every row index staying global (row bias per
    sample, conv image mapping, frame index of the temporal taps).  352 (192-row) ... 1350 (128 x 128) workgroups here.  Checked
    against the interpreter and torch; the same op on ONE chunk-sized problem is bitwise equal to its rows of the chunked run."""
    H = W = int(math.isqrt(HW))
    M = frames * HW
    P = Program()
    P.force_tile = tile
    g = _g(1100 + tile + C)
    y, res, out = P.alloc(M, C, "f32"), P.alloc(M, C, "f32"), P.alloc(M, C, "f16")
    rb = P.alloc(2, C, "f32") if kind == "conv" else None
    w = {"b": torch.randn(C, generator=g), "g": 1 + 0.1 * torch.randn(C, generator=g), "be": 0.1 * torch.randn(C, generator=g)}
    w["gb"] = torch.cat([w["g"], w["be"]])
    cin = 64
    a = P.alloc(M, cin, "f16")
    if kind == "conv":
        w["w"] = pk.conv3x3((torch.randn(C, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().float()).half()
        op = P.gemm("c", a, Ref("weight", 0, "w"), C, 9 * cin, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
                    conv=dict(Hin=H, Win=W, Cin=cin, stride=1, up=0, Hout=H, Wout=W), residual=res, rowbias=rb, rows_per_batch=M // 2,
                    allow_splitk=False)
    elif kind == "tconv":
        w["w"] = pk.tconv3((torch.randn(C, cin, 3, 1, 1, generator=g) / math.sqrt(3 * cin)).half().float()).half()
        op = P.gemm("t", a, Ref("weight", 0, "w"), C, 3 * cin, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_TCONV3,
                    conv=dict(F=frames // 2, HW=HW, Cin=cin), residual=res, allow_splitk=False)
    else:
        w["w"] = (torch.randn(C, cin, generator=g) / math.sqrt(cin)).half()
        op = P.gemm("l", a, Ref("weight", 0, "w"), C, cin, y, bias=Ref("weight", 0, "b"), residual=res, allow_splitk=False)
    assert op.meta["tile"] == tile and op.meta["split"] == 1
    fused = P.groupnorm("gn", y, Ref("weight", 0, "g"), Ref("weight", 0, "be"), out, n_inst=frames, eps=1e-5, silu=True, gb=Ref("weight", 0, "gb"))
    assert fused is op and op.i[16] == L.EPI_GN and len(P.ops) == 1, "the norm did not become the GEMM's epilogue"
    bm, bn, per_cu = Program._GN_EPI_TILES[tile]
    assert -(-M // bm) * -(-C // bn) > 256 * per_cu, "the grid must exceed what the device holds"

    def init(it):
        fill(it, a, g); fill(it, res, g, 2.0)
        if rb is not None:
            fill(it, rb, g)
    it, got = _gpu_run(P, w, init)
    _check(it, got, y, 2e-5, "the fp32 stream of the chunked launch")
    _check(it, got, out, 1e-3, "chunked fused GroupNorm vs the interpreter")
    v = read(got, y).float()
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(v.view(frames, HW, C).permute(0, 2, 1), 32, w["g"], w["be"], 1e-5))
    assert rel_l2(read(got, out).float(), ref.permute(0, 2, 1).reshape(M, C)) < 1e-3


@pytest.mark.parametrize("tile,N,K,M", [(0, 640, 128, 20000), (12, 1280, 128, 2500), (9, 512, 128, 30000)])
def test_layernorm_across_column_tiles_on_a_grid_larger_than_the_device(tile, N, K, M):
    """The cross-tile LayerNorm epilogue (round 5) on more row tiles than the device holds: rows are independent, so the launcher runs
    row chunks of whole tiles (round 6); ragged M."""
    P = Program()
    P.force_tile = tile
    g = _g(1200 + tile)
    a, res, out, ln_out = P.alloc(M, K, "f16"), P.alloc(M, N, "f32"), P.alloc(M, N, "f32"), P.alloc(M, N, "f16")
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half(), "b": torch.randn(N, generator=g),
         "g": 1 + 0.1 * torch.randn(N, generator=g), "be": 0.1 * torch.randn(N, generator=g)}
    w["gb"] = torch.cat([w["g"], w["be"]])
    op = P.gemm("l", a, Ref("weight", 0, "w"), N, K, out, bias=Ref("weight", 0, "b"), residual=res, allow_splitk=False,
                ln=(Ref("weight", 0, "gb"), Ref("weight", 0, "g"), Ref("weight", 0, "be"), ln_out, 1e-5))
    assert op.meta["tile"] == tile and op.i[8] == 2 and len(P.ops) == 1, "the LayerNorm did not become the GEMM's epilogue"
    bm, bn, per_cu = Program._LNX_TILES[tile]
    assert -(-M // bm) * -(-N // bn) > per_cu * 256
    it, got = _gpu_run(P, w, lambda it: (fill(it, a, g), fill(it, res, g, 3.0)))
    _check(it, got, out, 2e-5, "fp32 stream")
    _check(it, got, ln_out, 1e-3, "chunked cross-tile LayerNorm vs the interpreter")
    ref = torch.nn.functional.layer_norm(read(got, out).float() + 0.0, (N,), w["g"], w["be"], 1e-5)
    assert rel_l2(read(got, ln_out).float(), ref) < 1e-3


@pytest.mark.parametrize("kind,C,per_frame,dead,lo", [("conv", 640, True, True, False), ("conv", 1280, False, False, True), ("plain", 640, False, True, False),
                                                       ("plain", 320, True, False, False)])
def test_groupnorm_in_splitk_reduction_against_torch(kind, C, per_frame, dead, lo):
    """Round 5: a split-K GEMM whose result feeds a GroupNorm — the reduction over the slabs IS the loader of a single-pass cooperative
    GroupNorm launch (norm.hip splitk_gn_kernel): no splitk_reduce_kernel launch, no GROUPNORM op, the result stored only if someone else
    reads it.  The long-K convolutions of the 8x8 / 4x4 levels."""
    B, F, H, W = 2, 3, 8, 8
    M = B * F * H * W                                       # 384 rows: 3 row tiles of 128
    P = Program()
    P.force_tile = 5
    g = _g(700 + C)
    y = P.alloc(M, C, "f16" if dead else "f32")
    res = P.alloc(M, C, "f32") if not dead else None
    rb = P.alloc(B, C, "f32") if kind == "conv" else None
    w = {"b": torch.randn(C, generator=g), "g": 1 + 0.1 * torch.randn(C, generator=g), "be": 0.1 * torch.randn(C, generator=g)}
    w["gb"] = torch.cat([w["g"], w["be"]])
    if kind == "conv":
        cin = 256
        a = P.alloc(M, cin, "f16")
        w4 = (torch.randn(C, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half()
        w["w"] = pk.conv3x3(w4.float()).half()
        op = P.gemm("c", a, Ref("weight", 0, "w"), C, 9 * cin, y, bias=Ref("weight", 0, "b"), gather=L.GATHER_CONV3X3,
                    conv=dict(Hin=H, Win=W, Cin=cin, stride=1, up=0, Hout=H, Wout=W), residual=res, rowbias=rb, rows_per_batch=F * H * W)
    else:
        a = P.alloc(M, 2048, "f16")
        w["w"] = (torch.randn(C, 2048, generator=g) / math.sqrt(2048)).half()
        op = P.gemm("l", a, Ref("weight", 0, "w"), C, 2048, y, bias=Ref("weight", 0, "b"), residual=res)
    assert op.meta["split"] > 1, op.meta
    full = P.alloc(M, 2 * C if lo else C, "f16")
    out = full.col_slice(0, C)
    n_inst = B * F if per_frame else B
    fused = P.groupnorm("gn", y, Ref("weight", 0, "g"), Ref("weight", 0, "be"), out, n_inst=n_inst, eps=1e-5, silu=True, lo=lo,
                        gb=Ref("weight", 0, "gb"), x_dead=dead)
    assert fused is op and op.i[16] == L.EPI_GN and len(P.ops) == 1

    def init(it):
        fill(it, a, g)
        if res is not None:
            fill(it, res, g, 2.0)
        if rb is not None:
            fill(it, rb, g)
        if dead:
            it.mat(y.ref, M, C, C, torch.float16, {}).fill_(7.0)
    it, got = _gpu_run(P, w, init)
    _check(it, got, full, 1e-3, "GroupNorm in the split-K reduction vs the interpreter")
    if dead:
        assert bool((read(got, y) == 7.0).all()), "the dead result was stored"
    else:
        _check(it, got, y, 2e-5, "the fp32 stream beside the fused norm")
        v = read(got, y).float()
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(v.view(n_inst, M // n_inst, C).permute(0, 2, 1), 32, w["g"], w["be"], 1e-5))
        assert rel_l2(read(got, out).float(), ref.permute(0, 2, 1).reshape(M, C)) < 1e-3


@pytest.mark.parametrize("n_inst,rows,C,dt,lo", [(2, 1536, 640, "f32", True), (6, 64, 1280, "f32", True), (2, 24576, 640, "f32", False), (4, 256, 320, "f32", True)])
def test_groupnorm_cast_second_output(n_inst, rows, C, dt, lo):
    """Round 5: a GroupNorm whose input also feeds a 1x1 skip convolution writes the raw input's fp16 cast (+ low-order image) as a second
    output (single-pass cooperative kernel / one-workgroup-per-group kernel / the three-launch path)."""
    M = n_inst * rows
    P = Program()
    g = _g(800 + C)
    x, out, cast = P.alloc(M, C, dt), P.alloc(M, C, "f16"), P.alloc(M, 2 * C if lo else C, "f16")
    w = {"g": 1 + 0.1 * torch.randn(C, generator=g), "be": 0.1 * torch.randn(C, generator=g)}
    op = P.groupnorm("gn", x, Ref("weight", 0, "g"), Ref("weight", 0, "be"), out, n_inst=n_inst, eps=1e-5, silu=True, cast=cast, cast_lo=lo)
    assert op.kind == L.OP_GROUPNORM and len(P.ops) == 1
    it, got = _gpu_run(P, w, lambda it: fill(it, x, g, 3.0))
    _check(it, got, out, 1e-3, "normalised output")
    xv = read(got, x).float()
    hi = read(got, cast.col_slice(0, C)).float()
    assert torch.equal(hi, xv.half().float()), "the cast output is not the rounded input"
    if lo:
        assert rel_l2(hi + read(got, cast.col_slice(C, 2 * C)).float(), xv) < 1e-6


@pytest.mark.parametrize("tile,N,K,B,rows,Lc,wrap", [(8, 320, 320, 2, 768, 77, False), (11, 320, 320, 2, 512, 77, True), (0, 640, 640, 2, 256, 77, False),
                                                      (0, 1280, 1280, 2, 128, 7, False), (8, 320, 320, 1, 384, 96, False), (5, 1280, 1280, 2, 192, 77, False)])
def test_to_q_cross_attention_against_torch(tile, N, K, B, rows, Lc, wrap):
    """Round 5 (VERDICT r04 next #1b, the projection + attention half): q = to_q(LayerNorm(x)) and the 77-key text cross-attention as ONE
    launch — the accumulators become Q in LDS, K fragments come straight from the step-invariant K buffer, V^T from its transposed copy.
    Against the interpreter and torch's scaled_dot_product_attention on the fp16-rounded q; `wrap`: the A rows of one sample serve both
    samples (the cond | uncond parting site)."""
    M = B * rows
    heads, lcp = N // 64, -(-Lc // 32) * 32
    P = Program()
    P.force_tile = tile
    g = _g(900 + tile + N)
    a = P.alloc(rows if wrap else M, K, "f16")
    kv = P.alloc(B * Lc, N + 64, "f16")                    # K at columns 64 .. 64 + N (a window of a wider buffer, as in the UNet)
    vt = P.alloc(B * (N + 128), lcp, "f16")                # this site's rows after 64 rows of another site
    out = P.alloc(M, N, "f16")
    w = {"w": (torch.randn(N, K, generator=g) / math.sqrt(K)).half()}
    kbuf = kv.col_slice(64, 64 + N)
    vt_site = Buf(vt.row_slice(64, vt.rows).ref, vt.rows, lcp, lcp, "f16", owns=False)
    op = P.to_q_cross_attention("xa", a, Ref("weight", 0, "w"), out, k=K, heads=heads, kbuf=kbuf, vt=vt_site, n_keys=Lc, rows_per_sample=rows,
                                samples=B, scale=64 ** -0.5, a_wrap=rows if wrap else 0)
    assert op is not None and op.meta["tile"] == tile
    kmat = torch.randn(B, Lc, N, generator=g)
    vmat = torch.randn(B, Lc, N, generator=g)

    def init(it):
        fill(it, a, g)
        it.mat(kv.ref, B * Lc, N + 64, N + 64, torch.float16, {})[:, 64:] = kmat.reshape(B * Lc, N).half()
        vv = it.mat(vt.ref, B * (N + 128), lcp, lcp, torch.float16, {})
        vv.zero_()
        for b in range(B):
            vv[b * (N + 128) + 64: b * (N + 128) + 64 + N, :Lc] = vmat[b].t().half()
    it, got = _gpu_run(P, w, init)
    _check(it, got, out, 1e-3, "fused to_q + cross-attention vs the interpreter")
    av = read(got, a).float()
    q = ((torch.cat([av, av]) if wrap else av) @ w["w"].float().t()).half().float().view(B, rows, heads, 64).permute(0, 2, 1, 3)
    kk = kmat.half().float().view(B, Lc, heads, 64).permute(0, 2, 1, 3)
    vv = vmat.half().float().view(B, Lc, heads, 64).permute(0, 2, 1, 3)
    ref = torch.nn.functional.scaled_dot_product_attention(q, kk, vv).permute(0, 2, 1, 3).reshape(M, N)
    assert rel_l2(read(got, out).float(), ref) < 2e-3
