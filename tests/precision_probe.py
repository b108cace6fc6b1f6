"""Which fp16-stored ACTIVATION tensors cost how much parity?  (VERDICT r02 "what's weak" #1: weights were ranked in round 2,
activation operands never were.)

Runs the SAME denoise program the GPU executes through the CPU interpreter (tests/interp.py: same op records, arena, packed
weights; it predicted the device's error to 2 % in round 2) with a float32 SHADOW of every fp16 arena buffer: for a chosen
class of tensors the fp16 rounding at the store is switched off (the consumer then sees fp32 values = what an exact hi + lo
operand split delivers), everything else stays fp16.  Error is rel-L2 against the fp32 oracle port on the DEPLOYED weights
(`w.half().float()`, t2v_pipeline.py:103-104), i.e. the comparison of the `*_w16.npz` goldens.

    python tests/precision_probe.py [tiny|small] [frames]

Output: baseline, every class switched off alone (gain), everything but one class (what that class alone costs), cumulative
order.  CPU only; TEST INFRASTRUCTURE (a script, not collected by pytest; lives in tests/ because it imports oracle/ and interp.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from interp import Interp  # noqa: E402
from oracle import configs, synth, torch_port as tp  # noqa: E402
from sd_webui_text2video_amd import _lib as L  # noqa: E402
from sd_webui_text2video_amd import unet as U  # noqa: E402


SPLIT_TEXT_KV = False  # "sample" mode: the text K / V are their own class
BY_LEVEL = False       # "levels" on the command line: the round-4 classes are split by resolution level (rows of the tensor)


def classify(op) -> str:
    c = _classify(op)
    if BY_LEVEL and op.out is not None:
        return f"{c}@{op.out.rows}"
    return c


def _classify(op) -> str:
    """Class of the fp16 tensor an op writes (by op kind / name)."""
    n = op.name
    if op.kind == L.OP_GROUPNORM:
        if ".temopral_conv." in n:
            return "gn.tconv"                 # GN+SiLU -> temporal conv operand
        if n.endswith(".norm") or ".norm." in n:
            return "gn.transformer"           # GN -> proj_in operand
        return "gn.resblock"                  # GN+SiLU -> 3x3 conv operand (incl. the head)
    if op.kind == L.OP_LAYERNORM:
        return "ln"
    if op.kind in (L.OP_ATTENTION, L.OP_RELPOS_ATTN):
        return "attn.out"
    if op.kind == L.OP_NCTHW_TO_CL:
        return "x.latent"
    if op.kind == L.OP_TIME_EMBED:
        return "time"
    if op.kind == L.OP_COPY2D:                # fp32 stream -> fp16 operand casts
        return "cast.ctx" if n.startswith("context") else ("cast.skip" if n.endswith(".skip.cast") else "cast.resample")
    if op.kind == L.OP_GEMM:
        if op.i[16] == L.EPI_GEGLU:
            return "geglu.out"
        if n.endswith(".kv") or n == "attn2.kv.all":
            return "kv.text" if SPLIT_TEXT_KV else "qkv"     # K / V of the text: the SAME rounding at every step of a sampling run
        if n.endswith(".qkv") or n.endswith(".to_q"):
            return "qkv"
        if n.startswith("time_embed") or n == "emb_layers.all":
            return "time"
        if ".temopral_conv." in n or n.endswith(".in_layers.2"):
            return "norm_input"               # conv outputs consumed only by a GroupNorm (norm_input_dtype = f16)
        if n.endswith(".ff.net.2"):
            return "ff.out"                   # x4: fp16 operand of proj_out
        return "gemm.other"
    return "other"


class Probe(Interp):
    def __init__(self, prog, weights, exact=()):
        super().__init__(prog, weights, poison=False)
        self.shadow = torch.zeros(self.arena.numel() // 2, dtype=torch.float32)
        self.exact = set(exact)
        self.cur = None
        self.seen = {}

    def view(self, ref, shape, strides, dtype, ext):
        if dtype == torch.float16 and ref.space == "arena":
            assert ref.off % 2 == 0
            return torch.as_strided(self.shadow, tuple(shape), tuple(strides), ref.off // 2)
        return super().view(ref, shape, strides, dtype, ext)

    round_inner = False      # experiment: the fp32 INNER residual stream of the transformer blocks (x1 / x2 / x3) stored as fp16

    def _st(self, view, value, dtype):
        if self.round_inner and dtype == torch.float32 and self.cur.kind == L.OP_GEMM and self.nst == 0 and \
                (self.cur.name.endswith(".to_out") or (self.cur.name.endswith(".proj_in"))):
            self.nst += 1
            view.copy_(value.half().float())
            return
        if dtype == torch.float16 and view.dtype == torch.float32:          # a shadowed fp16 buffer
            cls = classify(self.cur)
            self.nst += 1
            is_lo = self.cur.kind in (L.OP_COPY2D, L.OP_NCTHW_TO_CL) or (self.cur.kind == L.OP_GROUPNORM and self.cur.i[16]) or \
                (self.cur.kind == L.OP_ATTENTION and self.cur.i[16]) or (self.cur.kind == L.OP_RELPOS_ATTN and self.cur.i[18]) or \
                (self.cur.kind == L.OP_GEMM and self.cur.i[7] == L.GATHER_PLAIN and self.cur.i[11] == 1 and self.cur.i[16] == L.EPI_NONE)
            if self.nst == 2 and is_lo:
                # the low-order image of a hi + lo cast (precise_operands): carries nothing when the hi store is already exact
                view.copy_(value.float() * 0 if cls in self.exact else value.half().float())
                return
            self.seen[cls] = self.seen.get(cls, 0) + 1
            view.copy_(value.float() if cls in self.exact else value.half().float())
        else:
            view.copy_(value.to(dtype))

    def run(self, ext, ops=None):
        for op in (self.prog.ops if ops is None else ops):
            self.cur, self.nst = op, 0
            getattr(self, f"_op{op.kind}")(op, ext)


def main():
    global BY_LEVEL
    if "levels" in sys.argv:
        BY_LEVEL = True
        sys.argv.remove("levels")
    fast = "fast" in sys.argv           # only the "this class exact alone" column
    if fast:
        sys.argv.remove("fast")
    steps, only = 0, None
    for a in list(sys.argv):
        if a.startswith("sample="):       # sample=N[:class,class]: N guided DDIM steps (eta 0, CFG 9) instead of one forward
            global SPLIT_TEXT_KV
            SPLIT_TEXT_KV = True
            body = a.split("=", 1)[1]
            steps = int(body.split(":")[0])
            only = body.split(":")[1].split(",") if ":" in body else None
            sys.argv.remove(a)
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cfg = dict(configs.TINY_UNET)
    hw = 16
    if which == "small":           # same topology, wider: closer to the full model's K (error averages down with K)
        cfg.update(dim=128)
    torch.manual_seed(0)
    lvdm = which.startswith("lvdm")     # VideoCrafter topology (SpatialTemporalTransformer: 4 attentions + 5 LayerNorms per block, rel-pos)
    if which == "full":                 # the RELEASED ModelScope width / depth (1.41 B) on a small clip: the per-config table VERDICT r03 asks for
        cfg = dict(configs.MODELSCOPE_UNET)
    if lvdm:
        from sd_webui_text2video_amd import videocrafter as VC
        cfg, hw = (dict(configs.LVDM_UNET), 16) if which == "lvdmfull" else (dict(configs.TINY_LVDM_UNET), 8)      # lvdmfull: released 0.96 B config
        net = VC.UNetModel(**cfg, init_weights=False)
    else:
        net = U.UNetSD(**cfg, init_weights=False)
    synth.load_synth(net, seed=0)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p.half().float())
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, F, hw, hw, generator=g)
    y = torch.randn(1, 7, cfg["context_dim"], generator=g).half().float()
    t = torch.tensor([801.0])
    ref = tp.lvdm_unet_forward(sd, cfg, x, t.long(), y) if lvdm else tp.unet_forward(sd, cfg, x, t.long(), y)
    net16 = net.half()
    comp = net16._compile(1, F, hw, hw, 7, "f32", "f32" if steps else "f16", "f16")
    weights = comp.packer.materialise(net16.state_dict(), "cpu")

    def run(exact=()):
        it = Probe(comp.prog, weights, exact)
        out = torch.empty(1, 4, F, hw, hw, dtype=torch.float16)
        it.run({L.EXT_X: x, L.EXT_T: t, L.EXT_CTX: y.half(), L.EXT_OUT: out})
        e = out.float() - ref
        return float(e.norm() / ref.norm()), it.seen

    if steps:
        return sample_mode(which, cfg, sd, comp, weights, x, y, F, hw, steps, lvdm, only)
    base, seen = run()
    if "inner16" in sys.argv:
        Probe.round_inner = True
        r16, _ = run()
        Probe.round_inner = False
        print(f"inner residual stream of the transformer blocks (proj_in / to_out outputs) stored as fp16: {base:.3e} -> {r16:.3e}", flush=True)
        return
    classes = sorted(seen)
    if len(sys.argv) > 3:
        classes = [c for c in classes if c.startswith(sys.argv[3])]
    print(f"{which} UNet, {F} frames @{hw}x{hw}, deployed (fp16-representable) weights and context; rel-L2 vs the fp32 oracle")
    print(f"baseline (all fp16 operands, as on the device): {base:.3e}; fp16 output rounding alone ~2.8e-4", flush=True)
    print(f"{'class':16s} {'stores':>6s} {'exact alone':>12s} {'gain':>8s} {'all-but-this exact':>20s}")
    alone = {}
    for c in classes:
        r1, _ = run({c})
        r2 = float("nan") if fast else run(set(sorted(seen)) - {c})[0]
        alone[c] = r1
        print(f"{c:16s} {seen[c]:6d} {r1:12.3e} {100 * (1 - r1 / base):7.1f}% {r2:20.3e}", flush=True)
    if fast:
        return
    rall, _ = run(set(sorted(seen)))
    print(f"every class exact: {rall:.3e} (floor: fp16 eps output + fp32 accumulation order)")
    print(f"every class exact: {rall:.3e}", flush=True)
    # cumulative, in the order of the single-class gains (one run per class)
    order, chosen = [], set()
    for c in sorted(classes, key=lambda c: alone[c]):
        chosen.add(c)
        order.append((c, run(chosen)[0]))
    print("cumulative, best single gains first: " + " -> ".join(f"{c} {r:.2e}" for c, r in order), flush=True)


def sample_mode(which, cfg, sd, comp, weights, x_T, y, F, hw, steps, lvdm, only):
    """Which classes carry the error of a SAMPLED latent?  Roundings that differ from step to step average out over the run, an error
    that is the same at every step (the text K / V, anything derived from constants) adds up: a class whose share GROWS from the single
    forward to the N-step output is of the second kind.  Deterministic DDIM (eta 0), CFG 9, linear-sd schedule, fp32 state; the
    reference is the fp32 oracle port stepping the same recurrence."""
    import numpy as np
    from sd_webui_text2video_amd.pipeline import beta_schedule
    betas = beta_schedule("linear_sd", 1000, init_beta=0.00085, last_beta=0.012).double()
    ac = torch.cumprod(1.0 - betas, 0)
    ts = list(range(999, -1, -(1000 // steps)))[:steps]
    g = torch.Generator().manual_seed(7)
    yu = torch.randn(y.shape, generator=g).half().float()
    scale = 9.0

    def loop(eps_fn):
        x = x_T.clone().float()
        for i, t in enumerate(ts):
            a_t = float(ac[t]); a_p = float(ac[ts[i + 1]]) if i + 1 < len(ts) else 1.0
            ec, eu = eps_fn(x, t, y), eps_fn(x, t, yu)
            eps = eu + scale * (ec - eu)
            x0 = (x - np.sqrt(1 - a_t) * eps) / np.sqrt(a_t)
            x = np.sqrt(a_p) * x0 + np.sqrt(1 - a_p) * eps
        return x

    fwd = tp.lvdm_unet_forward if lvdm else tp.unet_forward
    ref = loop(lambda x, t, c: fwd(sd, cfg, x, torch.tensor([t]), c))

    def run(exact=()):
        seen = {}

        def eps_fn(x, t, c):
            it = Probe(comp.prog, weights, exact)
            out = torch.empty(1, 4, F, hw, hw, dtype=torch.float32)
            it.run({L.EXT_X: x, L.EXT_T: torch.tensor([float(t)]), L.EXT_CTX: c.half(), L.EXT_OUT: out})
            seen.update(it.seen)
            return out.float()
        got = loop(eps_fn)
        return float((got - ref).norm() / ref.norm()), seen

    base, seen = run()
    classes = [c for c in sorted(seen) if only is None or any(c.startswith(o) for o in only)]
    print(f"{which} UNet, {F} frames @{hw}x{hw}: {steps}-step guided DDIM (CFG 9, eta 0), fp32 eps; rel-L2 of the sampled latent vs the fp32 oracle")
    print(f"baseline (all fp16 operands, as on the device): {base:.3e}", flush=True)
    print(f"{'class':16s} {'stores':>6s} {'exact alone':>12s} {'gain':>8s}")
    for c in classes:
        r1, _ = run({c})
        print(f"{c:16s} {seen[c]:6d} {r1:12.3e} {100 * (1 - r1 / base):7.1f}%", flush=True)
    rall, _ = run(set(seen))
    print(f"every class exact: {rall:.3e}", flush=True)


if __name__ == "__main__":
    main()
