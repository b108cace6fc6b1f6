"""TextToVideoSynthesis / process_modelscope — the reference's outer entry points (boundaries
B1/B2) over the MI355X hot path.

`TextToVideoSynthesis.infer` keeps the signature and return triple of
reference scripts/modelscope/t2v_pipeline.py:197-216,385:
    (frames_bgr_list, last_tensor, infotext)
and the same stages: conditioning -> seeded CPU noise -> sample_loop -> VAE decode of
x0/0.18215 -> tensor2vid (x*0.5+0.5, clamp, *255 truncated to uint8, RGB->BGR).

What is NOT rebuilt here (out of scope per SURVEY §2.1): the OpenCLIP text encoder
(clip_hardcode.py) — pass a `clip_encoder` callable/object, or call `infer_conditioned` with
pre-computed conditioning tensors; file / ffmpeg / Gradio plumbing of process_modelscope.py.

MI355X differences inside the stages: cond+uncond UNet evaluations are one batched forward,
and all frames are decoded by ONE batched VAE program (the reference decodes frame by frame and
copies each to the host, t2v_pipeline.py:329-355).
"""
from __future__ import annotations

import json
import os
import random
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L
from .samplers import Txt2VideoSampler, available_samplers
from .unet import UNetSD
from .vae import AutoencoderKL

SCALE_FACTOR = 0.18215          # t2v_pipeline.py:297

VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def beta_schedule(schedule, num_timesteps=1000, init_beta=None, last_beta=None):
    """t2v_model.py:1240-1249."""
    if schedule == "linear_sd":
        return torch.linspace(init_beta ** 0.5, last_beta ** 0.5, num_timesteps, dtype=torch.float64) ** 2
    raise ValueError(f"Unsupported schedule: {schedule}")


def tensor2vid_device(video: torch.Tensor, bgr: bool = False) -> torch.Tensor:
    """[i,3,F,H,W] float video -> uint8 [F,H,(i W),3] on the same device with ONE T2V_OP_TO_UINT8 launch: x*0.5+0.5, clamp,
    *255 TRUNCATED like `(image.numpy()*255).astype('uint8')` (t2v_pipeline.py:447-460) — in fp32 for an fp32 video, in
    fp16 for an fp16 one (what `mul_` / `add_` / numpy do on the reference's half-precision VAE output).  Bit-exact
    with the reference for identical float input (tests/test_gpu_boundary.py)."""
    import ctypes
    from . import _lib as L
    if not video.is_cuda:
        raise L.T2VError("tensor2vid_device needs a device tensor on an AMD GPU (no CPU fallback)")
    if video.dtype not in (torch.float16, torch.float32):
        video = video.float()
    video = video.contiguous()
    NI, C, Fr, H, W = video.shape
    out = torch.empty((Fr, H, NI * W, C), dtype=torch.uint8, device=video.device)
    op = L.T2VOp()
    op.kind = L.OP_TO_UINT8
    half = video.dtype == torch.float16
    vals = [NI, C, Fr, H, W, L.F16 if half else L.F32, int(half), int(bgr)]
    si, sc, sf, sy, sx = C * Fr * H * W, Fr * H * W, H * W, W, 1
    vals += [si & 0xFFFFFFFF, si >> 32, sc, sf & 0xFFFFFFFF, sf >> 32, sy, sx]
    if sc >= 2 ** 31:
        raise L.T2VError("video too large for one uint8 conversion launch")
    for k, v in enumerate(vals):
        op.i[k] = v - (1 << 32) if v >= (1 << 31) else v       # low words as raw bits
    op.p[0], op.p[1] = video.data_ptr(), out.data_ptr()
    L.check(L.load().t2v_run_ops(ctypes.byref(op), 1, None, 0, ctypes.c_void_p(torch.cuda.current_stream(video.device).cuda_stream)))
    return out


def tensor2vid(video: torch.Tensor) -> List[np.ndarray]:
    frames = tensor2vid_device(video).cpu().numpy()
    return [frames[i] for i in range(frames.shape[0])]


def create_infotext(vars_: dict) -> str:
    vars_ = dict(vars_)
    prompt = vars_.pop("prompt", "")
    n_prompt = vars_.pop("n_prompt", "")
    params = ", ".join(f"{k}: {v}" for k, v in vars_.items() if v is not None)
    neg = "\nNegative prompt: " + n_prompt if len(n_prompt) > 0 else ""
    return f"{prompt}{neg}\n{params}".strip()


class TextToVideoSynthesis(object):
    def __init__(self, model_dir: Optional[str] = None, *, sd_model: Optional[UNetSD] = None,
                 autoencoder: Optional[AutoencoderKL] = None, clip_encoder=None, betas=None,
                 device=None, tokenizer=None):
        """Either `model_dir` (configuration.json + checkpoints, as t2v_pipeline.py:45-146) or
        ready-made `sd_model` / `autoencoder` modules.  With a `model_dir` and no `clip_encoder`, the OpenCLIP text
        tower is built from `ckpt_clip` (t2v_pipeline.py:137-141) and runs on the GPU (text_encoder.py); `tokenizer` =
        open_clip's `_tokenizer` (BPE vocabulary; not part of this package)."""
        self.model_dir = model_dir
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.keep_in_vram = "All"
        self.clip_encoder = clip_encoder
        if model_dir is not None and sd_model is None:
            with open(os.path.join(model_dir, "configuration.json"), "r") as f:
                self.config = SimpleNamespace(**json.load(f))
            cfg = self.config.model["model_cfg"]
            cfg["temporal_attention"] = True if cfg["temporal_attention"] == "True" else False
            sd_model = UNetSD(in_dim=cfg["unet_in_dim"], dim=cfg["unet_dim"], y_dim=cfg["unet_y_dim"],
                              context_dim=cfg["unet_context_dim"], out_dim=cfg["unet_out_dim"],
                              dim_mult=cfg["unet_dim_mult"], num_heads=cfg["unet_num_heads"],
                              head_dim=cfg["unet_head_dim"], num_res_blocks=cfg["unet_res_blocks"],
                              attn_scales=cfg["unet_attn_scales"], dropout=cfg["unet_dropout"],
                              parameterization=cfg["mean_type"], temporal_attention=cfg["temporal_attention"],
                              init_weights=False)
            args = self.config.model["model_args"]
            sd_model.load_state_dict(torch.load(os.path.join(model_dir, args["ckpt_unet"]), map_location="cpu"), strict=True)
            sd_model.eval().half()
            betas = beta_schedule("linear_sd", cfg["num_timesteps"], init_beta=0.00085, last_beta=0.0120)
            autoencoder = AutoencoderKL(VAE_DDCONFIG, 4, os.path.join(model_dir, args["ckpt_autoencoder"]), init_weights=False)
            clip_path = os.path.join(model_dir, args.get("ckpt_clip", ""))
            if clip_encoder is None and os.path.isfile(clip_path):
                from .text_encoder import FrozenOpenCLIPEmbedder
                self.clip_encoder = FrozenOpenCLIPEmbedder(version=clip_path, device=self.device, layer="penultimate",
                                                           tokenizer=tokenizer)
        if betas is None:
            betas = beta_schedule("linear_sd", 1000, init_beta=0.00085, last_beta=0.0120)
        self.sd_model = sd_model
        self.autoencoder = autoencoder.eval() if autoencoder is not None else None
        self.betas = betas
        self.sd_model.register_schedule(given_betas=betas.numpy())
        self.diffusion = Txt2VideoSampler(self.sd_model, self.device, betas=betas)
        self.noise_gen = torch.Generator(device="cpu")
        self.last_tensor = None

    # ---- conditioning (out of scope: delegated) --------------------------------------------------
    def preprocess(self, prompt, n_prompt, steps):
        if self.clip_encoder is None:
            raise RuntimeError("no text encoder attached: pass clip_encoder=... (the reference's "
                               "FrozenOpenCLIPEmbedder) or call infer_conditioned(c, uc, ...)")
        enc = self.clip_encoder
        c = enc([prompt]) if callable(enc) else enc.encode([prompt])
        uc = enc([n_prompt]) if callable(enc) else enc.encode([n_prompt])
        return c, uc

    # ---- the hot path ---------------------------------------------------------------------------
    @torch.no_grad()
    def infer_conditioned(self, c, uc, steps, frames, seed, scale, width=256, height=256, eta=0.0,
                          device=None, latents=None, strength=None, mask=None, is_vid2vid=False,
                          sampler=available_samplers[0].name, decode=True, to_host=True, _keep_sampler=False,
                          videos: int = 1):
        """Stages 2-4 of `infer` for given conditioning tensors c, uc [1, 77k, 1024].
        Returns (frames, last_tensor): frames = list of HxWx3 uint8 BGR arrays (to_host) or a
        uint8 device tensor [F,H,W,3] RGB (to_host=False), or None when decode=False.
        `videos` > 1 (DDIM_Gaussian, txt2vid): that many independent videos of the same prompt in ONE batch — every
        UNet step is a single 2*videos forward; the frames come back side by side ([F, H, videos*W, 3], the layout
        `tensor2vid` gives a batch, t2v_pipeline.py:447-460).  Video v starts from the noise of seed + v, i.e. the batch is
        the reference's `batch_count` loop (process_modelscope.py:152-221: one video at a time, seed + batch) in one pass."""
        dev = torch.device(device) if device is not None else self.device
        self.device = dev
        self.diffusion.device = dev
        self.sd_model.to(dev)
        if not _keep_sampler:
            self.diffusion.get_sampler(sampler, return_sampler=False)
        latents, noise, shape = self.diffusion.get_noise(1, 4, frames, height, width, seed=seed, latents=latents)
        if videos > 1:
            if sampler != "DDIM_Gaussian" or latents is not None or is_vid2vid:
                raise NotImplementedError("several videos per batch: DDIM_Gaussian text-to-video only")
            one = (1,) + tuple(shape[1:])
            draws = []
            for v in range(videos):                      # each video from its own seed, as the batch_count loop draws them
                self.diffusion.noise_gen.manual_seed(seed + v)
                draws.append(torch.randn(one, generator=self.diffusion.noise_gen))
            noise = torch.cat(draws, dim=0).to(dev)
            shape = (videos,) + tuple(shape[1:])
        x0 = self.diffusion.sample_loop(
            steps=steps, strength=strength, eta=eta, conditioning=c.to(dev), unconditional_conditioning=uc.to(dev),
            batch_size=videos, guidance_scale=scale, latents=latents, shape=shape, noise=noise, is_vid2vid=is_vid2vid,
            sampler_name=sampler, mask=mask)
        self.last_tensor = x0
        if not decode:
            return None, x0
        if not to_host:
            return self.decode_frames(x0), x0
        arr = self.decode_frames(x0, bgr=True).cpu().numpy()          # BGR like postprocess_video (t2v_pipeline.py:430-433)
        L.async_status()               # the copy above synchronised: a fault raised by any kernel of this video surfaces HERE
        return [arr[i] for i in range(arr.shape[0])], x0

    @torch.no_grad()
    def compute_latents(self, vd_out, cpu_vae="GPU (half precision)", device=torch.device("cuda")):
        """vid2vid input side (t2v_pipeline.py:148-194): frames [b, 3, F, H, W] in [-1, 1] -> posterior mean x 0.18215,
        [b, 4, F, H/8, W/8] fp32 on the host.  All frames go through ONE batched encoder program (the reference
        encodes them one at a time)."""
        _note_cpu_vae(cpu_vae)                         # "CPU ..." modes: full-precision VAE, still on the GPU (see _note_cpu_vae)
        self.device = torch.device(device)
        self.autoencoder.to(self.device)
        bs, c, F, h, w = vd_out.shape
        x = vd_out.to(self.device).permute(0, 2, 1, 3, 4).reshape(bs * F, c, h, w).contiguous()
        if "half precision" in str(cpu_vae):
            x = x.half()
        mean = self.autoencoder.encode(x).mean.float() * SCALE_FACTOR
        return mean.view(bs, F, mean.shape[1], mean.shape[2], mean.shape[3]).permute(0, 2, 1, 3, 4).contiguous().cpu()

    @torch.no_grad()
    def decode_frames(self, x0: torch.Tensor, bgr: bool = False) -> torch.Tensor:
        """latent [b,4,F,h,w] -> uint8 [F,H,(b W),3] RGB (or BGR) on device: ONE batched VAE program over all
        frames of x0/0.18215 (t2v_pipeline.py:329-355 decodes them one at a time) + tensor2vid."""
        self.autoencoder.to(x0.device)
        bs, _, F, h, w = x0.shape
        z = (x0 * (1.0 / SCALE_FACTOR)).permute(0, 2, 1, 3, 4).reshape(bs * F, 4, h, w)   # '(b f) c h w'
        return self.autoencoder.decode_to_uint8(z, videos=bs, bgr=bgr)     # uint8 conversion = last op of the decoder program

    def infer(self, prompt, n_prompt, steps, frames, seed, scale, width=256, height=256, eta=0.0,
              cpu_vae="GPU (half precision)", device=torch.device("cuda"), latents=None, skip_steps=0,
              strength=0, mask=None, is_vid2vid=False, sampler=available_samplers[0].name):
        vars_ = dict(prompt=prompt, n_prompt=n_prompt, steps=steps, frames=frames, seed=seed, scale=scale,
                     width=width, height=height, eta=eta, cpu_vae=cpu_vae, device=str(device),
                     skip_steps=skip_steps, strength=strength, is_vid2vid=is_vid2vid, sampler=sampler)
        seed = seed if seed != -1 else random.randint(0, 2 ** 32 - 1)
        vars_["seed"] = seed
        _note_cpu_vae(cpu_vae)
        if "half precision" in str(cpu_vae):
            self.autoencoder.half()                      # t2v_pipeline.py:337-339
        steps = steps - skip_steps
        c, uc = self.preprocess(prompt, n_prompt, steps)
        strength = None if (strength == 0.0 and not is_vid2vid) else strength
        frames_bgr, x0 = self.infer_conditioned(c, uc, steps, frames, seed, scale, width, height, eta, device,
                                                latents, strength, mask, is_vid2vid, sampler)
        return frames_bgr, self.last_tensor, create_infotext(vars_)


_CPU_VAE_NOTED = False


def _note_cpu_vae(cpu_vae) -> None:
    """The reference's "CPU (Low VRAM)" VAE modes (t2v_pipeline.py:154-158,302-327) move the autoencoder to the host and run it in
    fp32 there to save VRAM.  This build has no host arithmetic path (and 288 GB of HBM make the saving moot): the option is
    ACCEPTED — a saved webui setting keeps working — and means what it means numerically in the reference, a VAE that is not
    halved, executed by the same HIP decoder / encoder programs on the GPU.  Said once on stderr, never silently."""
    global _CPU_VAE_NOTED
    if "CPU" in str(cpu_vae) and not _CPU_VAE_NOTED:
        _CPU_VAE_NOTED = True
        import sys
        print(f"[sd-webui-text2video_amd] cpu_vae={cpu_vae!r}: the VAE stays on the GPU (full-precision mode); "
              "this build has no host arithmetic path", file=sys.stderr)


pipe: Optional[TextToVideoSynthesis] = None     # module-global model cache, as process_modelscope.py:29


def frames_to_video_tensor(frames) -> torch.Tensor:
    """[F, H, W, 3] uint8 RGB frames -> [1, 3, F, H, W] float32 in [-1, 1]: the array arithmetic of process_modelscope.py:117-131
    (`/ 255`, `2 * x - 1`, one sample).  Reading / resizing the frames (ffmpeg, PIL) is host plumbing left to the caller."""
    arr = np.asarray(frames)
    if arr.ndim != 4 or arr.shape[-1] != 3:
        raise ValueError(f"expected frames [F, H, W, 3], got {arr.shape}")
    bcfhw = arr[np.newaxis].transpose(0, 4, 1, 2, 3).astype(np.float32) / 255
    return 2 * torch.from_numpy(np.ascontiguousarray(bcfhw)) - 1


def process_modelscope(args_dict: dict, extra_args=None):
    """Entry point B1 (process_modelscope.py:34-266).  The reference body is webui file / ffmpeg / Gradio plumbing around
    the `batch_count` loop of `pipe.infer(...)` calls (:152-221) and returns `list[str]`: one `data:video/mp4;base64,`
    URL per video, made from the mp4 that ffmpeg stitched (:248-266).  Here:
      args_dict = {model_dir | pipe, prompt, n_prompt, steps, frames, seed, cfg_scale, width, height, eta, sampler,
                   batch_count, clip_encoder | (cond, uncond), stitch,
                   do_vid2vid, vid2vid_frames, strength,                      # :80-147
                   inpainting_frames, inpainting_image, inpainting_weights}   # :170-217
    * `stitch(frames_bgr, infotext) -> bytes` (the ffmpeg stage, out of scope — e.g. the reference's own
      `ffmpeg_stitch_video` behind a temp directory) given: returns the reference's list of data-URLs, video b from
      seed + b (seed -1 stays random), exactly the reference loop;
    * no `stitch`: returns the BGR uint8 frames of the (last) video — what the reference writes as PNGs (:225-229);
      with (cond, uncond) tensors and batch_count > 1 the videos of seeds seed .. seed + batch_count - 1 are made in ONE
      batched pass, frames side by side.
    * vid2vid (`do_vid2vid`, :80-142): `vid2vid_frames` = the input clip as [F, H, W, 3] uint8 RGB frames already at
      (height, width) — what the reference has after vid2frames + PIL resize — or a [1, 3, F, H, W] float video in [-1, 1], or
      ready latents [1, 4, F, h, w]; encoded by ONE batched VAE-encoder program (`compute_latents`), then
      `skip_steps = floor(steps * clamp(1 - strength, 0, 1))` and `infer(..., latents, strength, skip_steps, is_vid2vid=True)`.
    * img2vid inpainting (`inpainting_frames` > 0 with an `inpainting_image` [H, W, 3] uint8 RGB, :170-217): the image is encoded
      for every frame, `inpainting_weights` = the per-frame mask weights (a sequence of `frames` floats — the reference reads
      them from its deforum-style key string through T2VAnimKeys, host plumbing), latents = image * (1 - mask) + N(0,1) * mask
      with the noise from numpy's GLOBAL generator exactly like the reference (:205), `strength = 1`, and `mask` goes to the
      sampler (where, for DDIM_Gaussian, the reference's hook is inert: SURVEY App. C #5)."""
    import math
    global pipe
    a = SimpleNamespace(**args_dict)
    if getattr(a, "pipe", None) is not None:
        pipe = a.pipe
    elif pipe is None:
        pipe = TextToVideoSynthesis(a.model_dir, clip_encoder=getattr(a, "clip_encoder", None))
    width, height = getattr(a, "width", 256), getattr(a, "height", 256)
    common = dict(frames=a.frames, scale=a.cfg_scale, width=width, height=height, eta=getattr(a, "eta", 0.0),
                  sampler=getattr(a, "sampler", available_samplers[0].name))
    batch_count = int(getattr(a, "batch_count", 1))
    stitch = getattr(a, "stitch", None)
    cpu_vae = getattr(a, "cpu_vae", "GPU (half precision)")
    device = pipe.device
    do_vid2vid = bool(getattr(a, "do_vid2vid", False))
    inpainting = int(getattr(a, "inpainting_frames", 0) or 0) > 0 and getattr(a, "inpainting_image", None) is not None
    have_cond = getattr(a, "cond", None) is not None

    def to_latents(video):
        """frames / float video / latents -> [1, 4, F, h, w] on the host (compute_latents, t2v_pipeline.py:148-194)."""
        if isinstance(video, torch.Tensor) and video.ndim == 5 and video.shape[1] == 4:
            return video.float().cpu()
        vd = video if (isinstance(video, torch.Tensor) and video.ndim == 5) else frames_to_video_tensor(video)
        if vd.shape[-2:] != (height, width):
            raise ValueError(f"input frames are {tuple(vd.shape[-2:])}, expected (height, width) = {(height, width)}: resize on the host")
        return pipe.compute_latents(vd, cpu_vae, device)

    mask = None
    if do_vid2vid:
        src = getattr(a, "vid2vid_frames", None)
        if src is None:
            raise FileNotFoundError("Please upload a video :()")          # process_modelscope.py:82
        latents = to_latents(src).to(device)
        strength = float(a.strength)
        skip_steps = int(math.floor(a.steps * max(0, min(1 - strength, 1))))
    else:
        latents, strength, skip_steps = None, 1, 0            # `args.strength = 1` (:145): UniPC starts at t_start = 1.0, like ns.T

    if have_cond and stitch is None and not do_vid2vid and not inpainting:
        frames, _ = pipe.infer_conditioned(a.cond, a.uncond, steps=a.steps, seed=a.seed, videos=batch_count, **common)
        return frames
    urls, frames = [], None
    for batch in range(batch_count):
        seed = a.seed + batch if a.seed != -1 else -1
        if inpainting:
            img = np.asarray(a.inpainting_image)
            if img.shape != (height, width, 3):
                raise ValueError(f"inpainting image is {img.shape}, expected {(height, width, 3)}: resize on the host")
            image_latents = to_latents(np.repeat(img[np.newaxis], a.frames, axis=0)).numpy()
            lh, lw = height // 8, width // 8
            latent_noise = np.random.normal(size=(1, 4, a.frames, lh, lw))            # the reference's unseeded draw (:205)
            weights = getattr(a, "inpainting_weights", None)
            if weights is None or isinstance(weights, str):
                raise ValueError("inpainting_weights: pass the per-frame mask weights as a sequence of floats (the key-string "
                                 "parser T2VAnimKeys is webui plumbing)")
            wts = [float(weights(i)) if callable(weights) else float(weights[i]) for i in range(a.frames)]
            m = np.ones(shape=(1, 4, a.frames, lh, lw))
            for i in range(a.frames):
                m[:, :, i, :, :] = wts[i]
            latents = torch.tensor(image_latents * (1 - m) + latent_noise * m).to(device)    # float64, like the reference
            mask = torch.tensor(m).to(device)
            strength = 1
        if have_cond:
            st = None if (strength == 0.0 and not do_vid2vid) else strength
            if "half precision" in str(cpu_vae) and getattr(pipe, "autoencoder", None) is not None:
                pipe.autoencoder.half()                      # what `infer` does before decoding (t2v_pipeline.py:337-339)
            frames, _ = pipe.infer_conditioned(a.cond, a.uncond, a.steps - skip_steps, seed=seed if seed != -1 else random.randint(0, 2 ** 32 - 1),
                                               latents=latents, strength=st, mask=mask, is_vid2vid=do_vid2vid, device=device, **common)
            info = ""
        else:
            frames, _, info = pipe.infer(a.prompt, getattr(a, "n_prompt", ""), a.steps, seed=seed, cpu_vae=cpu_vae, device=device,
                                         latents=latents, skip_steps=skip_steps, strength=strength, mask=mask, is_vid2vid=do_vid2vid,
                                         **common)
        if stitch is not None:
            import base64
            urls.append("data:video/mp4;base64," + base64.b64encode(stitch(frames, info)).decode())
    return urls if stitch is not None else frames
