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
    np.savez_compressed(os.path.join(OUT, "tiny.npz"), unet_eps=eps.numpy(), vae_img=img.numpy(),
                        sampler_x0=x0.numpy())
    print("tiny done", eps.std().item(), img.std().item(), x0.std().item())


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


if __name__ == "__main__":
    torch.manual_seed(0)
    tiny()
    if "--tiny-only" not in sys.argv:
        modelscope()
