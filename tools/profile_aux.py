"""Timings of the stages either side of the UNet loop on one MI355X, seeded random
weights of the released shapes:  VAE decode of 24 frames (+ uint8 conversion as the program's last op), VAE encode of 24
frames (vid2vid input side), tensor2vid alone, the OpenCLIP ViT-H text tower (2 x 77 tokens, 23 blocks), a LoRA-style
merge (re-pack of the images that read the touched attention weights) and the full weight pack.
    python tools/profile_aux.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sd_webui_text2video_amd import configs, pipeline  # noqa: E402
from sd_webui_text2video_amd import text_encoder as TE, unet as U, vae as V  # noqa: E402
from tools.profile_unet import random_weights_  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    F = 24
    ae = V.AutoencoderKL(configs.VAE_DDCONFIG, 4, init_weights=False).half().to(dev).eval()
    random_weights_(ae, 3)
    z = torch.randn(F, 4, 32, 32, device=dev)
    ms = timed(lambda: ae.decode_to_uint8(z, videos=1))
    print(f"VAE decode + tensor2vid, {F} frames 32x32 -> uint8 256x256 (one program): {ms:.2f} ms  ({F * 0.622 / ms * 1e3:.0f} TF/s of the 0.622 TFLOP/frame)")
    ms2 = timed(lambda: ae.decode(z))
    print(f"VAE decode alone (fp16 image out): {ms2:.2f} ms")
    img = torch.rand(F, 3, 256, 256, device=dev, dtype=torch.float16) * 2 - 1
    ms = timed(lambda: ae.encode(img))
    print(f"VAE encode, {F} frames 256x256 -> moments (vid2vid input side): {ms:.2f} ms")
    vid = torch.randn(1, 3, F, 256, 256, device=dev, dtype=torch.float16)
    ms = timed(lambda: pipeline.tensor2vid_device(vid), n=20)
    print(f"tensor2vid alone ({F} x 256x256, fp16 -> uint8, one launch): {ms * 1e3:.1f} us")
    m = TE.OpenClipTextModel(**TE.OPEN_CLIP_TEXT["ViT-H-14"]).half().to(dev)
    random_weights_(m, 9)
    emb = TE.FrozenOpenCLIPEmbedder(model=m, layer="penultimate", device=dev)
    tok = torch.randint(0, 49408, (2, 77), device=dev)
    ms = timed(lambda: emb.encode_with_transformers(tok))
    print(f"OpenCLIP ViT-H-14 text tower, 2 x 77 tokens, 23 blocks (cond + uncond in one batch): {ms:.2f} ms")
    net = U.UNetSD(**configs.MODELSCOPE_UNET, init_weights=False).half().to(dev).eval()
    random_weights_(net, 0)
    x = torch.randn(2, 4, 24, 32, 32, device=dev)
    y = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
    t = torch.full((2,), 500, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net(x, t, y)
    torch.cuda.synchronize()
    print(f"first UNet forward (lowering + full weight pack of 1.41 G parameters + bind): {(time.perf_counter() - t0) * 1e3:.0f} ms")
    mods = [mod for name, mod in net.named_modules() if isinstance(mod, torch.nn.Linear) and name.endswith(("attn1.to_q", "attn2.to_k", "attn2.to_v", "attn1.to_out.0"))]
    saved = [mod.weight for mod in mods]
    for mod in mods:                                 # what lora_processor.py:202-246 does: module.weight = Parameter(W + alpha * B @ A)
        mod.weight = torch.nn.Parameter(mod.weight.detach() * 1.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.refresh_weights(dev)
    torch.cuda.synchronize()
    print(f"LoRA-style merge of {len(mods)} attention projections: {net.last_repack} packed images rewritten in place in {(time.perf_counter() - t0) * 1e3:.1f} ms")
    for mod, w in zip(mods, saved):
        mod.weight = w
    net.refresh_weights(dev)


if __name__ == "__main__":
    main()
