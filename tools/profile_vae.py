"""Per-op HIP-event timings of the VAE decode of one clip (24 frames, 32x32 latents -> 256x256), the second stage of the
bench workload.  Usage: python tools/profile_vae.py [frames] [latent_h] [latent_w]   (1 72 128 = one 1024x576 ZeroScope-XL frame)"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import configs  # noqa: E402
from sd_webui_text2video_amd import vae as V  # noqa: E402
from tools.profile_unet import random_weights_  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    ae = V.AutoencoderKL(configs.VAE_DDCONFIG, 4, init_weights=False).half().to(dev).eval()
    random_weights_(ae, 3)
    lh = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    lw = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    z = torch.randn(n, 4, lh, lw, device=dev)
    for _ in range(2):
        out = ae.decode(z)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        out = ae.decode(z)
    torch.cuda.synchronize()
    wall = (time.time() - t0) / 3 * 1e3
    comp = next(c for k, c in ae._programs.items() if k[0] == n)
    st = torch.cuda.current_stream(dev).cuda_stream
    ext = {1: z.data_ptr(), 4: out.data_ptr()}
    comp.bound.run_timed(ext, st)
    ms = comp.bound.run_timed(ext, st)
    prog = comp.prog
    tot, fl = sum(ms), prog.total_flops()
    print(f"VAE decode {n} frames {lh}x{lw} -> {8 * lh}x{8 * lw}: {len(ms)} ops, wall {wall:.2f} ms, sum(op events) {tot:.2f} ms, "
          f"{fl / 1e12:.2f} TFLOP -> {fl / wall / 1e9:.1f} TF/s wall")
    names = {1: "gemm", 2: "groupnorm", 3: "layernorm", 4: "attention", 5: "softmax", 6: "to_cl", 7: "from_cl", 9: "copy2d"}
    rows = sorted(zip(ms, prog.ops), key=lambda r: -r[0])
    kinds = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for m, op in zip(ms, prog.ops):
        k = names.get(op.kind, str(op.kind))
        kinds[k][0] += m; kinds[k][1] += 1; kinds[k][2] += op.flops
    for k, (m, c, f) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:12s} {m:8.3f} ms {100 * m / tot:5.1f}% {c:4d} ops {f / max(m, 1e-9) / 1e9:8.1f} TF/s")
    att = [(m, op) for m, op in zip(ms, prog.ops) if ".attn_1." in op.name]
    core = [m for m, op in att if any(t in op.name for t in (".qk^T.", ".softmax.", ".pv."))]
    print(f"mid attention block: {sum(m for m, _ in att):.3f} ms in {len(att)} ops = {100 * sum(m for m, _ in att) / tot:.1f}% of the frame; "
          f"its score / softmax / PV core (what a fused d = 512 kernel would replace): {sum(core):.3f} ms = {100 * sum(core) / tot:.1f}%")
    for tkn in (".qk^T.", ".softmax.", ".pv."):
        sel = [(m, op) for m, op in att if tkn in op.name]
        if sel:
            meta = {k: v for k, v in sel[0][1].meta.items() if k in ("M", "N", "K", "tile", "split")}
            print(f"  {tkn:10s} {sum(m for m, _ in sel):7.3f} ms in {len(sel):3d} ops  {sum(op.flops for _, op in sel) / max(sum(m for m, _ in sel), 1e-9) / 1e9:7.1f} TF/s  {meta}")
    print("slowest ops:")
    for m, op in rows[:25]:
        meta = {k: v for k, v in op.meta.items() if k in ("M", "N", "K", "tile", "split", "n_inst", "rows", "C", "dt", "fused")}
        print(f"  {op.name:44s} {m * 1e3:8.1f} us {op.flops / max(m, 1e-9) / 1e9:7.1f} TF/s {meta}")


if __name__ == "__main__":
    main()
