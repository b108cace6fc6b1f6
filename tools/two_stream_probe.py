"""Round 6 probe: the guided UNet step as TWO b = 1 forwards (cond, uncond) on two HIP streams instead of one b = 2 forward.

Why: every GEMM of the step is a one-round grid whose phases (operand load -> MFMA main loop -> fp32 epilogue stores, + the statistics
exchange of a fused norm) run in lock-step on all 256 CUs, and the main loop is power-bound (1.55-1.68 GHz with the whole chip
in MFMAs).  Two independent half-size programs drift out of phase: one's HBM-bound epilogue runs under the other's main loop.

  python tools/two_stream_probe.py b2                 the product's step (one b = 2 program, shared cond | uncond prefix)
  python tools/two_stream_probe.py b1                 ONE b = 1 forward alone (what a CFG-pair rank runs)
  python tools/two_stream_probe.py streams            two b = 1 programs on two plain streams
  python tools/two_stream_probe.py mask <kind>        two b = 1 programs on two CU-masked streams (kind: halves | interleaved | xcdsplit),
                                                      lowered for 128 CUs (T2V_DEVICE_CUS=128)
Prints ms per guided step (both forwards), median of 5 batches of 10."""
import copy
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "b2"
kind = sys.argv[2] if len(sys.argv) > 2 else "halves"
F = int(os.environ.get("PROBE_FRAMES", "24"))
if mode == "mask":
    os.environ["T2V_DEVICE_CUS"] = "128"
from sd_webui_text2video_amd import configs, unet as U  # noqa: E402
from tools.profile_unet import random_weights_  # noqa: E402

dev = torch.device("cuda:0")
net = U.UNetSD(**configs.MODELSCOPE_UNET, init_weights=False).half().to(dev)
random_weights_(net)
x = torch.randn(1, 4, F, 32, 32, device=dev)
y2 = torch.randn(2, 77, net.context_dim, device=dev, dtype=torch.float16)
t1, t2 = torch.full((1,), 500, device=dev), torch.full((2,), 500, device=dev)


def masked_stream(words):
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)


def masks(kind):
    a, b = [0] * 8, [0] * 8
    for i in range(256):
        first = {"halves": i < 128, "interleaved": (i & 1) == 0, "xcdsplit": (i & 7) < 4}[kind]
        (a if first else b)[i // 32] |= 1 << (i % 32)
    return a, b


if mode in ("b2", "b1"):
    def step():
        net.single_timestep = True
        return net(x, t2, y2) if mode == "b2" else net(x, t1, y2[:1])
    sync = torch.cuda.synchronize
else:
    net_b = copy.copy(net)              # same parameters and packed weights, its own programs / arena
    net_b._programs = {}
    if mode == "streams":
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    else:
        ma, mb = masks(kind)
        sa, sb = masked_stream(ma), masked_stream(mb)
    ya, yb = y2[:1].contiguous(), y2[1:].contiguous()

    def step():
        with torch.cuda.stream(sa):
            oa = net(x, t1, ya)
        with torch.cuda.stream(sb):
            ob = net_b(x, t1, yb)
        return oa, ob

    def sync():
        sa.synchronize()
        sb.synchronize()

for _ in range(3):
    out = step()
sync()
o = out if torch.is_tensor(out) else torch.cat(out)
assert torch.isfinite(o.float()).all()
res = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    sync()
    res.append((time.perf_counter() - t0) / 10 * 1e3)
res.sort()
print(f"{mode + (' ' + kind if mode == 'mask' else ''):22s} {F} frames: {res[2]:7.2f} ms per guided step (min {res[0]:.2f}, max {res[-1]:.2f})", flush=True)
