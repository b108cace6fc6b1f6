"""Per-op timing of one UNet step on the GPU (HIP events around every op of the denoise program).
Usage: python tools/profile_unet.py [frames] [latent_h] [latent_w] [batch] [modelscope|lvdm]"""
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import configs  # noqa: E402
from sd_webui_text2video_amd import _lib as L, unet as U  # noqa: E402


def random_weights_(module, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                p.normal_(0, 1.0 / fan_in ** 0.5, generator=g)
            elif n.endswith("weight"):
                p.normal_(1.0, 0.1, generator=g)
            else:
                p.normal_(0.0, 0.05, generator=g)


HBM_TBS, MFMA_TFS = 8.0, 2500.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s, dense fp16 MFMA ~2.5 PFLOP/s


def op_bytes(op):
    """Algorithmic HBM bytes of one op in THIS design (what bench.py's `algorithmic_bytes_per_launch` sums for the GEMM family):
    every operand once, every result once."""
    import bench
    i = op.i
    if op.kind == 1:
        return bench.gemm_algorithmic_bytes(op)
    if op.kind == 2:                                   # GroupNorm: input (fp32 / fp16) once, fp16 output (+ its lo image)
        m = op.meta
        return m["n_inst"] * m["rows"] * m["C"] * ((4 if m["dt"] == "f32" else 2) + 2 + (2 if i[16] else 0))
    if op.kind == 3:                                   # LayerNorm: fp32 rows in, fp16 out
        return i[0] * i[1] * 6
    if op.kind in (4, 13):                             # attention: q and o once, k and v once per (batch, head) — fp16
        nq, nk, heads, bo, bi, d = i[0], i[1], i[2], i[3], i[4], i[14]
        kv_shared = (i[8] == 0)                        # text K / V: broadcast over the batch by a zero stride
        return heads * d * 2 * (2 * nq * bo * bi + 2 * nk * (1 if kv_shared else bo * bi))
    if op.kind == 9:                                   # casts / copies
        return i[0] * i[1] * ((4 if i[4] == 1 else 2) + (4 if i[5] == 1 else 2))
    return 0


def roofline_by_class(prog, ms, tot):
    """Every op against ITS roof: t_roof = max(bytes / 8 TB/s, flops / 2.5 PF/s); an op class is 'hbm' or 'mfma' by which term is the
    larger one.  sum(t_roof) is the step's own speed of light in this design (fp32 residual stream, fp16 operands)."""
    names = {1: "gemm", 2: "groupnorm", 3: "layernorm", 4: "attention", 9: "copy2d", 13: "relpos_attn"}
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0, 0.0, 0.0])      # ms, count, roof ms, bytes, flops
    for op, m in zip(prog.ops, ms):
        if op.kind not in names:
            continue
        b, fl = op_bytes(op), op.flops
        t_h, t_m = b / (HBM_TBS * 1e12) * 1e3, fl / (MFMA_TFS * 1e12) * 1e3
        k = names[op.kind]
        if op.kind == 1:
            k = "gemm/" + {0: "plain", 1: "conv3x3", 2: "tconv", 3: "conv_c8"}[op.meta["gather"]]
            lvl = op.meta["M"]
        elif op.kind == 2:
            lvl = op.meta["n_inst"] * op.meta["rows"]
        else:
            lvl = op.i[0] if op.kind in (3, 9) else op.i[0] * op.i[3] * op.i[4]
        key = (k, "hbm" if t_h >= t_m else "mfma", lvl)
        a = agg[key]
        a[0] += m; a[1] += 1; a[2] += max(t_h, t_m); a[3] += b; a[4] += fl
    print("every op against its own roof (bytes / 8 TB/s vs flops / 2.5 PF/s; rows = token rows of the op's level):")
    print(f"{'class':16s} {'bound':5s} {'rows':>7s} {'count':>5s} {'ms':>8s} {'roof ms':>8s} {'of roof':>8s} {'TB/s':>6s} {'TF/s':>7s}")
    sums = collections.defaultdict(lambda: [0.0, 0.0])
    for (k, bound, lvl), (m, c, r, b, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        sums[bound][0] += m; sums[bound][1] += r
        if m >= 0.15:
            print(f"{k:16s} {bound:5s} {lvl:7d} {c:5d} {m:8.3f} {r:8.3f} {100 * r / m:7.1f}% {b / m / 1e9:6.2f} {fl / m / 1e9:7.1f}")
    for bound, (m, r) in sums.items():
        print(f"all {bound}-bound ops: {m:7.3f} ms measured, {r:7.3f} ms at the roof = {100 * r / m:.1f} % of it")
    m_all, r_all = sum(v[0] for v in sums.values()), sum(v[1] for v in sums.values())
    print(f"step: {m_all:.3f} of {tot:.3f} ms classified; its own roofline {r_all:.3f} ms = {100 * r_all / m_all:.1f} % achieved "
          f"(HIP-event timings: +2-3 us per op over back-to-back execution)")


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    model = sys.argv[5] if len(sys.argv) > 5 else "modelscope"
    dev = torch.device("cuda:0")
    print(L.device_info())
    t0 = time.time()
    if model == "lvdm":
        from sd_webui_text2video_amd import videocrafter as VC
        net = VC.UNetModel(**configs.LVDM_UNET, init_weights=False).half().to(dev)
    else:
        net = U.UNetSD(**configs.MODELSCOPE_UNET, init_weights=False).half().to(dev)
    random_weights_(net)
    print(f"{model} model on device in {time.time() - t0:.1f}s")
    # B = 2 is the guided step of the samplers: ONE x_t for the cond | uncond pair (forward_cfg_pair; the ops in front of the first text
    # cross-attention run once, UNetSD.share_cfg_prefix); PROFILE_EXPLICIT_BATCH=1 profiles the explicit two-sample batch instead
    Bx = 1 if (B == 2 and os.environ.get("PROFILE_EXPLICIT_BATCH") != "1") else B
    x = torch.randn(Bx, 4, F, H, W, device=dev)
    y = torch.randn(B, 77, net.context_dim, device=dev, dtype=torch.float16)
    t = torch.full((B,), 500, device=dev)
    def fwd():
        net.single_timestep = True           # one t for the cond | uncond pair (what the samplers tell the UNet): the prefix is shared
        return net(x, t, context=y) if model == "lvdm" else net(x, t, y)

    for _ in range(2):
        out = fwd()
    torch.cuda.synchronize()
    assert os.environ.get("T2V_PROFILE_NOCHECK") == "1" or torch.isfinite(out.float()).all()
    n = 5
    t0 = time.time()
    for _ in range(n):
        out = fwd()
    torch.cuda.synchronize()
    wall = (time.time() - t0) / n * 1e3
    net.auto_refresh = False
    t0 = time.time()
    for _ in range(n):
        out = fwd()
    torch.cuda.synchronize()
    wall2 = (time.time() - t0) / n * 1e3
    _, ms, prog = net.forward_timed(x, t, y)
    _, ms, prog = net.forward_timed(x, t, y)
    tot = sum(ms)
    flops = prog.total_flops()
    print(f"geometry b{B} (x batch {Bx}) f{F} {H}x{W}: ops {len(ms)}  wall/forward {wall:.2f} ms (no auto-refresh {wall2:.2f} ms)  "
          f"sum(op events) {tot:.2f} ms")
    print(f"algorithmic {flops / 1e12:.3f} TFLOP -> {flops / (wall2 * 1e-3) / 1e12:.1f} TF/s wall, "
          f"{flops / (tot * 1e-3) / 1e12:.1f} TF/s by events; arena {prog.arena.high / 2**30:.2f} GiB")
    kinds = collections.defaultdict(lambda: [0.0, 0, 0.0])
    names = {1: "gemm", 2: "groupnorm", 3: "layernorm", 4: "attention", 5: "softmax", 6: "to_cl", 7: "from_cl",
             8: "time_embed", 9: "copy2d", 10: "ddim", 11: "memset", 12: "lincomb", 13: "relpos_attn"}
    for op, m in zip(prog.ops, ms):
        k = names[op.kind]
        if op.kind == 1:
            k = "gemm/" + {0: "plain", 1: "conv3x3", 2: "tconv", 3: "conv_c8"}[op.meta["gather"]]
        kinds[k][0] += m; kinds[k][1] += 1; kinds[k][2] += op.flops
    print(f"{'kind':16s} {'ms':>9s} {'%':>6s} {'count':>6s} {'TF/s':>8s}")
    for k, (m, c, fl) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:16s} {m:9.3f} {100 * m / tot:6.1f} {c:6d} {fl / max(m, 1e-9) / 1e9:8.1f}")
    shapes = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for op, m in zip(prog.ops, ms):
        if op.kind == 1:
            key = (op.meta["gather"], op.meta["M"], op.meta["N"], op.meta["K"], op.meta["split"], op.meta["epi"])
            shapes[key][0] += m; shapes[key][1] += 1; shapes[key][2] += op.flops
    print("top GEMM shapes (gather, M, N, K, split, epi): ms total, count, TF/s")
    for key, (m, c, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"  {str(key):44s} {m:8.3f} {c:4d} {fl / max(m, 1e-9) / 1e9:8.1f}")
    att = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for op, m in zip(prog.ops, ms):
        if op.kind == 4:
            key = (op.i[0], op.i[1], op.i[2], op.i[3] * op.i[4])
            att[key][0] += m; att[key][1] += 1; att[key][2] += op.flops
    print("attention (nq, nk, heads, batch): ms total, count, TF/s")
    for key, (m, c, fl) in sorted(att.items(), key=lambda kv: -kv[1][0]):
        print(f"  {str(key):32s} {m:8.3f} {c:4d} {fl / max(m, 1e-9) / 1e9:8.1f}")
    roofline_by_class(prog, ms, tot)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"unet_ops_b{B}_f{F}_{H}x{W}.json" if model == "modelscope" else f"lvdm_ops_b{B}_f{F}_{H}x{W}.json"), "w") as f:
        json.dump([dict(name=op.name, kind=op.kind, ms=m, flops=op.flops, meta={k: v for k, v in op.meta.items() if k != "conv"})
                   for op, m in zip(prog.ops, ms)], f)


if __name__ == "__main__":
    main()
