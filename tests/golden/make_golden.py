"""Generate the committed golden fixtures FROM THE REAL REFERENCE (run in the build container).

    python tests/golden/make_golden.py            # tiny + full ModelScope config (~2 min CPU)

The reference (kabachuha/sd-webui-text2video @ /root/reference) ships no tests / golden
vectors, so these fixtures are outputs of the reference's own classes
(UNetSD, AutoencoderKL, Txt2VideoSampler+GaussianDiffusion), imported read-only through
oracle/ref_bootstrap.py, on the seeded synthetic weights/inputs of oracle/synth.py.
They pin oracle/torch_port.py (tests/test_oracle_pin.py, CPU) and are the end-to-end
targets of the GPU parity tests.  Only outputs are stored; inputs/weights are re-derived
from seeds.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import configs, ref_bootstrap as rb, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def tiny():
    cfg = configs.TINY_UNET
    unet, betas = rb.build_reference_unet(cfg)
    synth.load_synth(unet, seed=0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 3, 16, 16, generator=g)
    y = torch.randn(2, 7, cfg["context_dim"], generator=g)
    t = torch.tensor([801, 401])
    with torch.no_grad():
        eps = unet(x, t, y)
    vae = rb.build_reference_vae(configs.TINY_VAE_DDCONFIG)
    synth.load_synth(vae, seed=3)
    z = torch.randn(2, 4, 8, 8, generator=g)
    with torch.no_grad():
        img = vae.decode(z)
    ref = rb.bootstrap()
    s = ref.samplers.Txt2VideoSampler(unet, torch.device("cpu"), betas=betas, sampler_name="DDIM_Gaussian")
    lat, noise, shape = s.get_noise(1, 4, 3, 128, 128, seed=1234)
    c = torch.randn(1, 7, cfg["context_dim"], generator=g)
    uc = torch.randn(1, 7, cfg["context_dim"], generator=g)
    with torch.no_grad():
        x0 = s.sample_loop(steps=4, strength=None, conditioning=c, unconditional_conditioning=uc,
                           batch_size=1, latents=lat, shape=shape, noise=noise, guidance_scale=9.0,
                           eta=0.0, sampler_name="DDIM_Gaussian")
    extra = other_samplers(ref, unet, betas, lat, noise, shape, c, uc)
    # vid2vid input side: VAE encode of synthetic frames (t2v_model.py:1640-1644) and the DDIM_Gaussian vid2vid loop
    # (samplers_common.py:165-207 with is_vid2vid=True -> encode_latent -> add_noise, :123-145)
    frames = torch.rand(3, 3, 64, 48, generator=torch.Generator().manual_seed(9)) * 2 - 1
    with torch.no_grad():
        extra["vae_moments"] = vae.encode(frames).parameters.numpy()
        z0 = torch.randn(shape, generator=torch.Generator().manual_seed(11))
        s2 = ref.samplers.Txt2VideoSampler(unet, torch.device("cpu"), betas=betas, sampler_name="DDIM_Gaussian")
        extra["vid2vid_x0"] = s2.sample_loop(steps=4, strength=0.5, conditioning=c, unconditional_conditioning=uc, batch_size=1,
                                             latents=z0, shape=shape, noise=noise, is_vid2vid=True, guidance_scale=9.0, eta=0.0,
                                             sampler_name="DDIM_Gaussian").numpy()
    np.savez_compressed(os.path.join(OUT, "tiny.npz"), unet_eps=eps.numpy(), vae_img=img.numpy(),
                        sampler_x0=x0.numpy(), **extra)
    print("tiny done", eps.std().item(), img.std().item(), x0.std().item(), {k: float(v.std()) for k, v in extra.items()})


def other_samplers(ref, unet, betas, lat, noise, shape, c, uc):
    """"DDIM" and "UniPC" through the reference's own Txt2VideoSampler.  Both reference classes pin
    their buffers to torch.device("cuda") (ddim/sampler.py:11,18-22; uni_pc/sampler.py:13-17); the only
    change made here is to point that device at the CPU — the arithmetic is the reference's, untouched."""
    import samplers.uni_pc.sampler as ups
    ups.UniPCSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    out = {}
    cpu = torch.device("cpu")
    with torch.no_grad():
        s = ref.samplers.Txt2VideoSampler(unet, cpu, betas=betas, sampler_name="DDIM")
        s.sampler.device = cpu
        out["ddim_x0"] = s.sample_loop(steps=4, strength=None, conditioning=c, unconditional_conditioning=uc,
                                       batch_size=1, latents=lat, shape=shape, noise=noise, guidance_scale=9.0,
                                       eta=0.0, sampler_name="DDIM").numpy()
        s = ref.samplers.Txt2VideoSampler(unet, cpu, betas=betas, sampler_name="UniPC")
        out["unipc_x0"] = s.sample_loop(steps=6, strength=None, conditioning=c, unconditional_conditioning=uc,
                                        batch_size=1, latents=lat, shape=shape, noise=noise, guidance_scale=9.0,
                                        eta=0.0, sampler_name="UniPC").numpy()
        out["unipc_x0_s07"] = s.sample_loop(steps=4, strength=0.7, conditioning=c, unconditional_conditioning=uc,
                                            batch_size=1, latents=lat, shape=shape, noise=noise, guidance_scale=7.0,
                                            eta=0.0, sampler_name="UniPC").numpy()
        # vid2vid noising helpers (samplers_common.py:123-145)
        g = torch.Generator().manual_seed(11)
        z0 = torch.randn(shape, generator=g)
        out["unipc_encode"] = s.sampler.unipc_encode(z0, cpu, 0.7, 4, noise=noise).numpy()
        s = ref.samplers.Txt2VideoSampler(unet, cpu, betas=betas, sampler_name="DDIM")
        s.sampler.device = cpu
        enc, dsteps = s.encode_latent(z0, noise, 0.75, 4)
        out["ddim_encode"] = enc.numpy()
        # after encode_latent `sampler.sample` is `sampler.decode` (samplers_common.py:137); called directly here
        out["ddim_vid2vid_x0"] = s.sampler.decode(enc, c, dsteps, unconditional_guidance_scale=9.0,
                                                  unconditional_conditioning=uc).numpy()
    return out


def modelscope(frames=8, steps=5):
    """BASELINE.json configs[0]: ModelScope fp32, 8 frames @256x256, 5 DDIM steps, CPU reference."""
    cfg = configs.MODELSCOPE_UNET
    t0 = time.time()
    unet, betas = rb.build_reference_unet(cfg)
    synth.load_synth(unet, seed=0)
    print("weights", time.time() - t0)
    noise, cond, uncond = synth.synth_inputs(frames, 256, 256)
    t = torch.tensor([801])
    with torch.no_grad():
        t0 = time.time()
        eps = unet(noise, t, cond)
        t_fwd = time.time() - t0
    ref = rb.bootstrap()
    s = ref.samplers.Txt2VideoSampler(unet, torch.device("cpu"), betas=betas, sampler_name="DDIM_Gaussian")
    lat, nz, shape = s.get_noise(1, 4, frames, 256, 256, seed=1234)
    assert torch.equal(nz, noise)
    with torch.no_grad():
        t0 = time.time()
        x0 = s.sample_loop(steps=steps, strength=None, conditioning=cond, unconditional_conditioning=uncond,
                           batch_size=1, latents=lat, shape=shape, noise=nz, guidance_scale=9.0, eta=0.0,
                           sampler_name="DDIM_Gaussian")
        t_loop = time.time() - t0
    del unet
    vae = rb.build_reference_vae(configs.VAE_DDCONFIG)
    synth.load_synth(vae, seed=3)
    with torch.no_grad():
        t0 = time.time()
        img = vae.decode(x0[:, :, 0] / configs.SCALE_FACTOR)
        t_vae = time.time() - t0
    np.savez_compressed(os.path.join(OUT, "modelscope_8f.npz"), unet_eps=eps.numpy(), sampler_x0=x0.numpy(),
                        vae_img_frame0=img.numpy().astype(np.float32),
                        timing=np.array([t_fwd, t_loop, t_vae, torch.get_num_threads()], dtype=np.float64))
    print(f"modelscope done fwd {t_fwd:.2f}s loop {t_loop:.2f}s vae {t_vae:.2f}s",
          eps.std().item(), x0.std().item(), img.std().item())


def lvdm_inputs_tiny():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 5, 8, 8, generator=g)
    ctx = torch.randn(2, 9, 768, generator=g)
    x_T = torch.randn(1, 4, 5, 8, 8, generator=g)
    return x, torch.tensor([801, 401]), ctx, x_T


def lvdm(full=True):
    """VideoCrafter: the reference's own UNetModel (openaimodel3d.py) and DDIMSampler (lvdm/samplers/ddim.py).
    LatentDiffusion itself needs pytorch_lightning (absent), so the sampler is given a stand-in exposing exactly what
    it reads from the model: num_timesteps, the DDPM.register_schedule buffers (computed with the reference's
    make_beta_schedule) and apply_model -> UNetModel.forward.  Its cuda-pinned register_buffer is pointed at the CPU."""
    import importlib
    import types
    rb.bootstrap()
    om = importlib.import_module("videocrafter.lvdm.models.modules.openaimodel3d")
    vu = importlib.import_module("videocrafter.lvdm.models.modules.util")
    dd = importlib.import_module("videocrafter.lvdm.samplers.ddim")
    dd.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    cfg = configs.TINY_LVDM_UNET
    net = om.UNetModel(**cfg).eval()
    synth.load_synth(net, seed=0)
    x, t, ctx, x_T = lvdm_inputs_tiny()
    with torch.no_grad():
        eps = net(x, t, context=ctx)
    betas = vu.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas, axis=0)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    model = types.SimpleNamespace(num_timesteps=1000, betas=f32(betas), alphas_cumprod=f32(ac),
                                  alphas_cumprod_prev=f32(np.append(1.0, ac[:-1])), device=torch.device("cpu"),
                                  apply_model=lambda xx, tt, c, **kw: net(xx, tt, context=c))
    smp = dd.DDIMSampler(model)
    smp.noise_gen.manual_seed(123)
    with torch.no_grad():
        x0, _ = smp.sample(S=4, conditioning=ctx[0:1], batch_size=1, shape=list(x_T.shape[1:]), verbose=False,
                           unconditional_guidance_scale=7.5, unconditional_conditioning=ctx[1:2], eta=0.3, x_T=x_T)
    np.savez_compressed(os.path.join(OUT, "lvdm_tiny.npz"), unet_eps=eps.numpy(), ddim_x0=x0.numpy())
    print("lvdm tiny done", eps.std().item(), x0.std().item())
    if not full:
        return
    cfg = configs.LVDM_UNET
    net = om.UNetModel(**cfg).eval()
    synth.load_synth(net, seed=0)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, 4, 16, 32, 32, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g)
    with torch.no_grad():
        t0 = time.time()
        eps = net(x, torch.tensor([500]), context=ctx)
        t_fwd = time.time() - t0
    np.savez_compressed(os.path.join(OUT, "lvdm_16f.npz"), unet_eps=eps.numpy(),
                        timing=np.array([t_fwd, torch.get_num_threads()], dtype=np.float64))
    print(f"lvdm 16f done fwd {t_fwd:.2f}s", eps.std().item())


def infer_inputs_tiny():
    g = torch.Generator().manual_seed(17)
    c = torch.randn(1, 7, configs.TINY_UNET["context_dim"], generator=g)
    uc = torch.randn(1, 7, configs.TINY_UNET["context_dim"], generator=g)
    return c, uc


def infer_tiny():
    """Boundary B2 end to end: the reference's OWN `TextToVideoSynthesis.infer` (t2v_pipeline.py:197-385 — get_noise ->
    sample_loop -> per-frame VAE decode of x0/0.18215 -> tensor2vid -> RGB2BGR) on the tiny UNet / VAE, CPU fp32 ('CPU (full
    precision)' VAE branch).  The object is assembled without `__init__` (which needs configuration.json, checkpoints and
    open_clip); the text-encoding stage (`preprocess`, outside the hot path) is replaced by fixed conditioning tensors."""
    import types
    ref = rb.bootstrap()
    pl = rb.bootstrap_pipeline()
    unet, betas = rb.build_reference_unet(configs.TINY_UNET)
    synth.load_synth(unet, seed=0)
    vae = rb.build_reference_vae(configs.TINY_VAE_DDCONFIG)
    synth.load_synth(vae, seed=3)
    pipe = object.__new__(pl.TextToVideoSynthesis)
    pipe.device = torch.device("cpu")
    pipe.sd_model, pipe.autoencoder = unet, vae
    pipe.keep_in_vram = "All"
    pipe.clip_encoder = types.SimpleNamespace(to=lambda d: None, device=None)
    pipe.diffusion = ref.samplers.Txt2VideoSampler(unet, torch.device("cpu"), betas=betas, sampler_name="DDIM_Gaussian")
    c, uc = infer_inputs_tiny()
    pipe.preprocess = lambda prompt, n_prompt, steps, offload=True: (c, uc)
    out = {}
    for tag, kw in (("", dict(steps=4, frames=3, seed=1234, scale=9.0, width=128, height=128)),
                    ("_wide", dict(steps=3, frames=2, seed=77, scale=7.5, width=192, height=64))):
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            frames, last, info = pipe.infer("a prompt", "a negative prompt", kw["steps"], kw["frames"], kw["seed"], kw["scale"],
                                            kw["width"], kw["height"], 0.0, "CPU (full precision)", torch.device("cpu"),
                                            sampler="DDIM_Gaussian")
        out["frames_bgr" + tag] = np.stack(frames)
        out["last_tensor" + tag] = last.numpy()
        out["infotext" + tag] = np.array(info)
        print("infer", tag, out["frames_bgr" + tag].shape, out["frames_bgr" + tag].mean(), info.replace("\n", " | "))
    # tensor2vid itself on a fixed float video (bit-exact target of the device kernel), fp32 and fp16 inputs
    g = torch.Generator().manual_seed(23)
    vid = torch.randn(2, 3, 3, 16, 24, generator=g) * 0.8
    vid[0, :, 0, 0, :6] = torch.tensor([-1.0, 1.0, 0.999, 0.0, -0.00392, 0.99609])
    out["t2v_in"] = vid.numpy()
    out["t2v_u8_f32"] = np.stack(pl.tensor2vid(vid.clone()))
    out["t2v_u8_f16"] = np.stack(pl.tensor2vid(vid.clone().half()))
    np.savez_compressed(os.path.join(OUT, "infer_tiny.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--infer-only" in sys.argv:
        infer_tiny()
        sys.exit(0)
    if "--lvdm-only" not in sys.argv:
        tiny()
        if "--tiny-only" not in sys.argv:
            modelscope()
    lvdm(full="--tiny-only" not in sys.argv)
