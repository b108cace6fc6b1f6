"""Sampler facade + DDIM_Gaussian solver — drop-in for the reference's boundary B3
(scripts/samplers/samplers_common.py:95-207, scripts/samplers/ddim/gaussian_sampler.py).

Kept: `Txt2VideoSampler(sd_model, device, betas, sampler_name)`, `.get_noise`, `.get_sampler`,
`.encode_latent`, `.sample_loop`, the `available_samplers` registry of `SamplerBase`, the
per-solver `.sample(S, conditioning, ..., callback, ...)` keyword contract, the step callback
(tqdm + webui `state.sampling_step` + Interrupt/Skip -> InterruptedException).

MI355X-native parts: the two UNet evaluations of a guided step run as ONE batched b=2
forward (weights stream from HBM once per step), and the whole pointwise update
(half-channel CFG combine, x0, eps, x_{t-1}) is one HIP kernel (T2V_OP_DDIM_STEP) on the
same stream — the host only computes five fp32 scalars per step and never synchronises.

Only "DDIM_Gaussian" (the UI default, t2v_helpers/args.py:234) is built in this round; "DDIM"
and "UniPC" are SURVEY §8(f)-1 and raise a clear error.
"""
from __future__ import annotations

import ctypes
import types
from typing import Optional

import torch

from . import _lib as L
from .program import BoundProgram, Program

try:  # inside the webui these exist; outside we provide inert stand-ins
    from modules.shared import state  # type: ignore
    from modules.sd_samplers_common import InterruptedException  # type: ignore
except Exception:  # pragma: no cover - exercised outside webui
    state = types.SimpleNamespace(interrupted=False, skipped=False, sampling_step=0, sampling_steps=0)

    class InterruptedException(BaseException):
        pass

try:
    from modules import prompt_parser as _prompt_parser  # type: ignore
except Exception:  # pragma: no cover
    _prompt_parser = None

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None


def reconstruct_conds(cond, uncond, step):
    """t2v_helpers/general_utils.py:27-30 — webui prompt schedules; identity for plain tensors."""
    if _prompt_parser is not None and not torch.is_tensor(cond):
        return (_prompt_parser.reconstruct_cond_batch(cond, step),
                _prompt_parser.reconstruct_cond_batch(uncond, step))
    return cond, uncond


def get_height_width(h, w, divisor):
    return h // divisor, w // divisor


def get_tensor_shape(batch_size, channels, frames, h, w, latents=None):
    if latents is None:
        return (batch_size, channels, frames, h, w)
    return latents.shape


def inpaint_masking(xt, step, steps, mask, add_noise_cb, noise_cb_args):
    """Frame in-painting hook.  In the reference this is a no-op for DDIM_Gaussian (the result is
    neither returned nor assigned, samplers_common.py:17-26) — kept inert."""
    return None


class SamplerStepCallback(object):
    """samplers_common.py:28-69."""

    def __init__(self, sampler_name: str, total_steps: int, progress: bool = True):
        self.sampler_name = sampler_name
        self.total_steps = total_steps
        self.current_step = 0
        state.sampling_steps = total_steps
        self.progress_bar = tqdm(desc=f"Sampling using {sampler_name} for {total_steps} steps.",
                                 total=total_steps) if (tqdm is not None and progress) else None

    def interrupt(self):
        return state.interrupted or state.skipped

    def cancel(self):
        raise InterruptedException

    def update(self, step):
        state.sampling_step = step
        if self.interrupt():
            self.cancel()
        if self.progress_bar is not None:
            self.progress_bar.update(1)
            if step >= self.total_steps:
                self.progress_bar.close()
        if step >= self.total_steps:
            self.current_step = 0

    def __call__(self, *args, **kwargs):
        self.current_step += 1
        self.update(self.current_step)


class GaussianDiffusion(object):
    """DDIM sampler "DDIM_Gaussian" (gaussian_sampler.py:5-296), inference side."""

    def __init__(self, model, betas, mean_type="eps", var_type="learned_range", loss_type="mse",
                 epsilon=1e-12, rescale_timesteps=False, **kwargs):
        if not isinstance(betas, torch.Tensor):
            betas = torch.tensor(betas, dtype=torch.float64)
        betas = betas.detach().to("cpu", torch.float64)
        assert float(betas.min()) > 0 and float(betas.max()) <= 1
        assert mean_type == "eps", "only eps-prediction is used by the pipeline"
        self.model = model
        self.betas = betas
        self.num_timesteps = len(betas)
        self.mean_type, self.var_type, self.loss_type = mean_type, var_type, loss_type
        self.epsilon, self.rescale_timesteps = epsilon, rescale_timesteps
        alphas = 1 - betas
        self.alphas_cumprod = torch.cumprod(alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod - 1)
        self._step_plans = {}
        self.cfg_parallel = None      # parallel.CfgPair: cond / uncond forwards on two GPUs (parallel.py)

    # -- schedule helpers (gaussian_sampler.py:73-91) --------------------------------------------
    def get_time_steps(self, ddim_timesteps, batch_size=1, step=None):
        steps = (1 + torch.arange(0, self.num_timesteps, ddim_timesteps)).clamp(0, self.num_timesteps - 1)
        timesteps = steps.flip(0).to(self.model.device)
        if step is not None:
            timesteps = torch.full((batch_size,), int(timesteps[step]), dtype=torch.long, device=self.model.device)
        return timesteps

    def add_noise(self, xt, noise, t):
        t = t.cpu()
        dev = self.model.device
        return self.sqrt_alphas_cumprod[t].to(dev) * xt + noise * self.sqrt_one_minus_alphas_cumprod[t].to(dev)

    def get_dim(self, y_out):
        return y_out.size(1) if self.var_type.startswith("fixed") else y_out.size(1) // 2

    def is_unconditional(self, guide_scale):
        return guide_scale is None or guide_scale == 1

    # -- fused update kernel -------------------------------------------------------------------
    def _step_plan(self, C, inner, guided, eps_dtype, x_dtype):
        key = (C, inner, guided, eps_dtype, x_dtype)
        plan = self._step_plans.get(key)
        if plan is None:
            prog = Program("ddim_step")
            prog.ddim_step("ddim.step", C=C, inner=inner, guided=guided, eps_dtype=eps_dtype, x_dtype=x_dtype)
            plan = BoundProgram(prog, 0, {})
            self._step_plans[key] = plan
        return plan

    @torch.no_grad()
    def sample(self, x_T=None, S=5, shape=None, conditioning=None, unconditional_conditioning=None,
               model_kwargs={}, clamp=None, percentile=None, condition_fn=None,
               unconditional_guidance_scale=None, eta=0.0, callback=None, mask=None, **kwargs):
        """gaussian_sampler.py:214-296.  Returns x_0-ish latent after S DDIM steps."""
        if clamp is not None or percentile is not None or condition_fn is not None or model_kwargs:
            raise NotImplementedError("clamp / percentile / classifier guidance are not used by the pipeline")
        lib = L.load()
        model = self.model
        dev = torch.device(model.device)
        steps, stride = S, self.num_timesteps // S
        guide = unconditional_guidance_scale
        if x_T is None:
            xt = torch.randn(shape, device=dev)
        else:
            xt = x_T.to(dev).clone()
        if xt.dtype not in (torch.float16, torch.float32):
            xt = xt.float()
        xt = xt.contiguous()
        assert xt.shape[0] == 1, "the reference pipeline samples one video at a time (num_sample = 1)"
        x_next = torch.empty_like(xt)
        Bx, C, Fr, Hh, Ww = xt.shape
        inner = Fr * Hh * Ww
        x_dt = "f16" if xt.dtype == torch.float16 else "f32"

        ac = self.alphas_cumprod
        f32 = torch.float32
        if hasattr(model, "refresh_weights"):
            model.refresh_weights(dev)
        prev_auto = getattr(model, "auto_refresh", None)
        if prev_auto is not None:
            model.auto_refresh = False
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        try:
            all_t = self.get_time_steps(stride, 1).cpu()
            for step in range(steps):
                c, uc = reconstruct_conds(conditioning, unconditional_conditioning, step)
                t = int(all_t[step])
                tt = torch.full((1,), t, dtype=torch.long, device=dev)
                if self.is_unconditional(guide):
                    eps = model(xt, tt, c)
                    guided, gscale = 0, 1.0
                elif self.cfg_parallel is not None and self.cfg_parallel.size == 2:
                    # CFG pair: this rank evaluates ONE of the two forwards; one eps all-gather per step
                    mine = c if self.cfg_parallel.role == 0 else uc
                    eps = self.cfg_parallel.exchange_eps(model(xt, tt, mine))
                    guided, gscale = C // 2 if not self.var_type.startswith("fixed") else C, float(guide)
                else:
                    if getattr(model, "supports_cfg_batch", False):
                        eps = model(torch.cat([xt, xt], dim=0), torch.cat([tt, tt]), torch.cat([c, uc], dim=0))
                    else:
                        eps = torch.cat([model(xt, tt, c), model(xt, tt, uc)], dim=0)
                    guided, gscale = C // 2 if not self.var_type.startswith("fixed") else C, float(guide)
                eps = eps.contiguous()
                # schedule scalars: float64 tables cast to fp32 per lookup, then fp32 arithmetic —
                # the exact operation order of gaussian_sampler.py:269-283 / `_i` (t2v_model.py:1232)
                a_t, a_prev = ac[t].to(f32), ac[max(t - stride, 0)].to(f32)
                sigma = eta * torch.sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev))
                coef = (ctypes.c_float * 6)(
                    float(self.sqrt_recip_alphas_cumprod[t].to(f32)), float(self.sqrt_recipm1_alphas_cumprod[t].to(f32)),
                    float(torch.sqrt(a_prev)), float(torch.sqrt(1 - a_prev - sigma ** 2)),
                    float(sigma) if t != 0 else 0.0, gscale)
                noise = torch.randn_like(xt, dtype=f32)     # drawn every step, like the reference (global RNG)
                _ = torch.randn_like(xt, dtype=f32)         # the (inert) inpaint hook's draw, gaussian_sampler.py:288
                plan = self._step_plan(C, inner, guided, "f16" if eps.dtype == torch.float16 else "f32", x_dt)
                L.check(lib.t2v_ddim_step(plan.handle, xt.data_ptr(), eps.data_ptr(),
                                          noise.data_ptr() if float(sigma) != 0.0 else None,
                                          x_next.data_ptr(), coef, ctypes.c_void_p(stream)))
                xt, x_next = x_next, xt
                if callback is not None:
                    callback(step)
        finally:
            if prev_auto is not None:
                model.auto_refresh = prev_auto
        return xt


class SamplerBase(object):
    """samplers_common.py:71-87."""

    def __init__(self, name: str, Sampler, frame_inpaint_support=False):
        self.name = name
        self.Sampler = Sampler
        self.frame_inpaint_support = frame_inpaint_support

    def register_buffers_to_model(self, sd_model, betas, device):
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        setattr(sd_model, "device", device)
        setattr(sd_model, "betas", betas)
        setattr(sd_model, "alphas_cumprod", self.alphas_cumprod)

    def init_sampler(self, sd_model, betas, device, **kwargs):
        self.register_buffers_to_model(sd_model, betas, device)
        return self.Sampler(sd_model, betas=betas, **kwargs)


def _not_built(name):
    class _Missing(object):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"sampler '{name}' is SURVEY §8(f)-1 (next round); use 'DDIM_Gaussian'")
    _Missing.__name__ = name
    return _Missing


available_samplers = [
    SamplerBase("DDIM_Gaussian", GaussianDiffusion, True),
    SamplerBase("DDIM", _not_built("DDIM")),
    SamplerBase("UniPC", _not_built("UniPC")),
]


class Txt2VideoSampler(object):
    """samplers_common.py:95-207.  (The reference's default sampler_name is "UniPC"; this build's
    default is the UI default "DDIM_Gaussian".)"""

    def __init__(self, sd_model, device, betas=None, sampler_name="DDIM_Gaussian"):
        self.sd_model = sd_model
        self.device = device
        self.noise_gen = torch.Generator(device="cpu")
        self.sampler_name = sampler_name
        self.betas = betas
        self.progress = True
        self.sampler = self.get_sampler(sampler_name, betas=self.betas)

    def get_noise(self, num_sample, channels, frames, height, width, latents=None, seed=1):
        """Seeded CPU (mt19937) randn so the initial noise is identical on every device (:104-121)."""
        if latents is not None:
            latents = latents.to(self.device)
        num_sample = 1
        latent_h, latent_w = get_height_width(height, width, 8)
        shape = get_tensor_shape(num_sample, channels, frames, latent_h, latent_w, latents)
        self.noise_gen.manual_seed(seed)
        noise = torch.randn(shape, generator=self.noise_gen).to(self.device)
        return latents, noise, shape

    def encode_latent(self, latent, noise, strength, steps):
        """vid2vid noising (:123-145) — DDIM_Gaussian branch (`add_noise`)."""
        denoise_steps = int(strength * steps)
        timestep = self.sampler.get_time_steps(denoise_steps, latent.shape[0])
        encoded_latent = self.sampler.add_noise(latent, noise, timestep[0].cpu())
        return encoded_latent, denoise_steps

    def get_sampler(self, sampler_name: str, betas=None, return_sampler=True):
        betas = betas if betas is not None else self.betas
        for Sampler in available_samplers:
            if sampler_name == Sampler.name:
                sampler = Sampler.init_sampler(self.sd_model, betas=betas, device=self.device)
                if Sampler.frame_inpaint_support:
                    setattr(sampler, "inpaint_masking", inpaint_masking)
                if return_sampler:
                    return sampler
                self.sampler = sampler
                return
        raise ValueError(f"Sample {sampler_name} does not exist.")

    def sample_loop(self, steps, strength, conditioning, unconditional_conditioning, batch_size, latents=None,
                    shape=None, noise=None, is_vid2vid=False, guidance_scale=1, eta=0, mask=None,
                    sampler_name="DDIM"):
        denoise_steps = None
        if latents is not None and is_vid2vid:
            latents, denoise_steps = self.encode_latent(latents, noise, strength, steps)
        sampler_callback = SamplerStepCallback(sampler_name, steps, progress=self.progress)
        x0 = self.sampler.sample(
            S=steps, conditioning=conditioning, strength=strength,
            unconditional_conditioning=unconditional_conditioning, batch_size=batch_size,
            x_T=latents if latents is not None else noise, x_latent=latents, t_start=denoise_steps,
            unconditional_guidance_scale=guidance_scale, shape=shape, callback=sampler_callback,
            cond=conditioning, eta=eta, mask=mask)
        return x0
