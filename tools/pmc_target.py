"""Minimal target for rocprofv3 --pmc passes: one warm-up and two timed UNet forwards (b=2, 24f @ 32x32 latent,
the bench workload's UNet step).  Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/pmc_target.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd import configs  # noqa: E402
from sd_webui_text2video_amd import unet as U  # noqa: E402
from tools.profile_unet import random_weights_  # noqa: E402

dev = torch.device("cuda:0")
net = U.UNetSD(**configs.MODELSCOPE_UNET, init_weights=False).half().to(dev)
random_weights_(net)
x = torch.randn(1, 4, 24, 32, 32, device=dev)      # ONE x_t for the cond | uncond pair: the samplers' guided step (shared prefix)
y = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
t = torch.full((2,), 500, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    net.single_timestep = True                       # one t for the pair, as the samplers say: the prefix up to the first text cross-attention is shared
    out = net(x, t, y)
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
