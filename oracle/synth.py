"""TEST INFRASTRUCTURE — deterministic synthetic weights and inputs.

There are no checkpoints and no network, so parity and perf run on seeded synthetic
weights.  EVERY parameter is re-drawn, because the reference zero-initialises the last
layer of every residual branch (t2v_model.py:631-636, :708-713, :955-956, :1215-1216, :326)
and a parity test on default init would be vacuous.

The draw depends only on the ORDERED (name, shape) list, so the reference nn.Module, the
torch port and the product module (identical `named_parameters()` order) get bit-identical
weights from the same seed — here and on the GPU box (same torch build, CPU mt19937).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Iterable, Tuple

import torch


def synth_tensor(name: str, shape, gen: torch.Generator) -> torch.Tensor:
    shape = tuple(shape)
    if len(shape) >= 2:                       # conv / linear weight: N(0, 1/fan_in)
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=gen) * (1.0 / math.sqrt(fan_in))
    if name.endswith("weight"):               # norm gamma: 1 + 0.1 N
        return 1.0 + 0.1 * torch.randn(shape, generator=gen)
    return 0.05 * torch.randn(shape, generator=gen)   # biases (conv/linear/norm beta)


def synth_state_dict(spec: Iterable[Tuple[str, tuple]], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    out = OrderedDict()
    for name, shape in spec:
        out[name] = synth_tensor(name, shape, gen)
    return out


def param_spec(module: torch.nn.Module):
    return [(n, tuple(p.shape)) for n, p in module.named_parameters()]


def load_synth(module: torch.nn.Module, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Overwrite every parameter of `module` in place with the seeded draw; returns the dict."""
    sd = synth_state_dict(param_spec(module), seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(sd[n].to(p.dtype))
    return sd


def synth_inputs(frames: int, height: int, width: int, ctx_len: int = 77, ctx_dim: int = 1024,
                 seed: int = 1234):
    """noise exactly as Txt2VideoSampler.get_noise (samplers_common.py:104-121): CPU generator,
    shape [1,4,F,H/8,W/8]; cond/uncond ~ N(0,1) from seeds 1/2."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    noise = torch.randn((1, 4, frames, height // 8, width // 8), generator=g)
    g.manual_seed(1)
    cond = torch.randn((1, ctx_len, ctx_dim), generator=g)
    g.manual_seed(2)
    uncond = torch.randn((1, ctx_len, ctx_dim), generator=g)
    return noise, cond, uncond
