"""CPU: pin the travelling oracle (oracle/torch_port.py) against the golden fixtures generated
from the REAL reference (tests/golden/make_golden.py) and, when /root/reference is mounted,
against the reference's own classes run live."""
import os

import numpy as np
import pytest
import torch

from oracle import configs, ref_bootstrap as rb, synth, torch_port as tp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _tiny_inputs():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 3, 16, 16, generator=g)
    y = torch.randn(2, 7, configs.TINY_UNET["context_dim"], generator=g)
    z = torch.randn(2, 4, 8, 8, generator=g)
    c = torch.randn(1, 7, configs.TINY_UNET["context_dim"], generator=g)
    uc = torch.randn(1, 7, configs.TINY_UNET["context_dim"], generator=g)
    return x, torch.tensor([801, 401]), y, z, c, uc


def _spec_unet(cfg):
    from sd_webui_text2video_amd import unet
    return synth.param_spec(unet.UNetSD(**cfg, init_weights=False))


def _spec_vae(dd):
    from sd_webui_text2video_amd import vae
    return synth.param_spec(vae.AutoencoderKL(dd, 4, init_weights=False))


def test_port_matches_reference_golden_tiny():
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    x, t, y, z, c, uc = _tiny_inputs()
    sd = synth.synth_state_dict(_spec_unet(configs.TINY_UNET), seed=0)
    eps = tp.unet_forward(sd, configs.TINY_UNET, x, t, y)
    assert np.abs(eps.numpy() - gold["unet_eps"]).max() < 2e-5
    vsd = synth.synth_state_dict(_spec_vae(configs.TINY_VAE_DDCONFIG), seed=3)
    img = tp.vae_decode(vsd, configs.TINY_VAE_DDCONFIG, z)
    assert np.abs(img.numpy() - gold["vae_img"]).max() < 2e-5
    betas = tp.beta_schedule_linear_sd()
    noise, _, _ = synth.synth_inputs(3, 128, 128)
    x0 = tp.ddim_gaussian_sample(lambda a, b, cc: tp.unet_forward(sd, configs.TINY_UNET, a, b, cc), betas, noise, 4, c, uc, 9.0, 0.0)
    assert np.abs(x0.numpy() - gold["sampler_x0"]).max() < 2e-4 * np.abs(gold["sampler_x0"]).max()


def test_port_ddim_and_unipc_match_reference_golden():
    """oracle restatements of "DDIM" (ddim/sampler.py) and "UniPC" (uni_pc/*) vs outputs of the reference's
    own sampler classes (tests/golden/make_golden.py:other_samplers)."""
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    _, _, _, _, c, uc = _tiny_inputs()
    sd = synth.synth_state_dict(_spec_unet(configs.TINY_UNET), seed=0)
    betas = tp.beta_schedule_linear_sd()
    noise, _, _ = synth.synth_inputs(3, 128, 128)
    z0 = torch.randn(noise.shape, generator=torch.Generator().manual_seed(11))

    def f(a, b, cc):
        return tp.unet_forward(sd, configs.TINY_UNET, a, b, cc)

    def rel(a, k):
        return np.abs(a.numpy() - gold[k]).max() / np.abs(gold[k]).max()

    assert rel(tp.ddim_ldm_sample(f, betas, noise, 4, c, uc, 9.0), "ddim_x0") < 2e-5
    assert rel(tp.unipc_sample(f, betas, noise, 6, c, uc, 9.0), "unipc_x0") < 2e-5
    assert rel(tp.unipc_sample(f, betas, noise, 4, c, uc, 7.0, t_start=0.7), "unipc_x0_s07") < 2e-5
    assert rel(tp.unipc_encode(betas, z0, 0.7, noise), "unipc_encode") < 1e-6
    enc = tp.ddim_ldm_encode(betas, 4, z0, 3, noise)
    assert rel(enc, "ddim_encode") < 1e-6
    assert rel(tp.ddim_ldm_sample(f, betas, enc, 4, c, uc, 9.0, t_start=3), "ddim_vid2vid_x0") < 2e-5


def test_port_vae_encode_and_vid2vid_match_reference_golden():
    """VAE encode (posterior moments) and the DDIM_Gaussian vid2vid flow: encode_latent noises the input latents at
    get_time_steps(int(strength*steps))[0] — the step COUNT is passed where a stride is expected, so t = 999 for any
    strength (samplers_common.py:139-143, gaussian_sampler.py:73-85) — then all S steps run."""
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    _, _, _, _, c, uc = _tiny_inputs()
    vsd = synth.synth_state_dict(_spec_vae(configs.TINY_VAE_DDCONFIG), seed=3)
    frames = torch.rand(3, 3, 64, 48, generator=torch.Generator().manual_seed(9)) * 2 - 1
    m = tp.vae_encode(vsd, configs.TINY_VAE_DDCONFIG, frames)
    assert np.abs(m.numpy() - gold["vae_moments"]).max() < 2e-5
    sd = synth.synth_state_dict(_spec_unet(configs.TINY_UNET), seed=0)
    betas = tp.beta_schedule_linear_sd()
    noise, _, _ = synth.synth_inputs(3, 128, 128)
    z0 = torch.randn(noise.shape, generator=torch.Generator().manual_seed(11))
    ac = torch.cumprod(1 - betas, 0)
    x_T = torch.sqrt(ac)[999].float() * z0 + noise * torch.sqrt(1 - ac)[999].float()
    x0 = tp.ddim_gaussian_sample(lambda a, b, cc: tp.unet_forward(sd, configs.TINY_UNET, a, b, cc), betas, x_T, 4, c, uc, 9.0, 0.0)
    assert np.abs(x0.numpy() - gold["vid2vid_x0"]).max() < 2e-4 * np.abs(gold["vid2vid_x0"]).max()


def test_timestep_grid_matches_reference_quirk():
    # S=5 -> [801,601,401,201,1]; S=50 -> [981,...,1]
    assert tp.ddim_gaussian_timesteps(1000, 5).tolist() == [801, 601, 401, 201, 1]
    ts = tp.ddim_gaussian_timesteps(1000, 50)
    assert ts[0] == 981 and ts[-1] == 1 and len(ts) == 50


def test_tensor2vid_truncates():
    v = torch.tensor([0.999, -1.0, 1.0, 0.0]).view(1, 1, 1, 1, 4).repeat(1, 3, 1, 1, 1)
    out = tp.tensor2vid_uint8(v)[0]
    assert out[0, :, 0].tolist() == [254, 0, 255, 127]     # (x*0.5+0.5)*255 truncated


@pytest.mark.skipif(not rb.reference_available(), reason="/root/reference not mounted (GPU box)")
def test_port_matches_live_reference_blocks():
    """Per-sub-module agreement with the reference's own forward (hooks), tiny config."""
    cfg = configs.TINY_UNET
    unet, _ = rb.build_reference_unet(cfg)
    sd = synth.load_synth(unet, seed=0)
    x, t, y, *_ = _tiny_inputs()
    got = {}
    hooks = []
    for name, mod in unet.named_modules():
        if name.count(".") == 2 and name.split(".")[0] in ("input_blocks", "output_blocks") or \
                (name.startswith("middle_block.") and name.count(".") == 1):
            hooks.append(mod.register_forward_hook(lambda m, i, o, n=name: got.__setitem__(n, o)))
    with torch.no_grad():
        ref = unet(x, t, y)
    for h in hooks:
        h.remove()
    taps = {}
    mine = tp.unet_forward(sd, cfg, x, t, y, taps=taps)
    assert (mine - ref).abs().max() < 2e-5
    checked = 0
    for name, r in got.items():
        if name in taps and r.ndim == 4:
            assert (taps[name] - r).abs().max() < 5e-5 * max(1.0, float(r.abs().max())), name
            checked += 1
    assert checked > 40


def test_oracle_chain_reproduces_the_references_own_infer():
    """Row a1: UNet port -> DDIM_Gaussian port -> VAE port -> tensor2vid port -> BGR == the frames the reference's
    `TextToVideoSynthesis.infer` produced (golden infer_tiny.npz, tests/golden/make_golden.py::infer_tiny)."""
    gold = np.load(os.path.join(GOLD, "infer_tiny.npz"))
    cfg = configs.TINY_UNET
    sd = synth.synth_state_dict(_spec_unet(cfg), seed=0)
    vsd = synth.synth_state_dict(_spec_vae(configs.TINY_VAE_DDCONFIG), seed=3)
    g = torch.Generator().manual_seed(17)
    c = torch.randn(1, 7, cfg["context_dim"], generator=g)
    uc = torch.randn(1, 7, cfg["context_dim"], generator=g)
    betas = tp.beta_schedule_linear_sd()
    for tag, (steps, frames, seed, scale, w, h) in (("", (4, 3, 1234, 9.0, 128, 128)), ("_wide", (3, 2, 77, 7.5, 192, 64))):
        noise, _, _ = synth.synth_inputs(frames, h, w, seed=seed)
        x0 = tp.ddim_gaussian_sample(lambda a, b, cc: tp.unet_forward(sd, cfg, a, b, cc), betas, noise, steps, c, uc, scale, 0.0)
        assert np.abs(x0.numpy() - gold["last_tensor" + tag]).max() < 2e-4 * np.abs(gold["last_tensor" + tag]).max()
        img = torch.cat([tp.vae_decode(vsd, configs.TINY_VAE_DDCONFIG, x0[:, :, f] / configs.SCALE_FACTOR) for f in range(frames)], 0)
        vid = img.permute(1, 0, 2, 3).unsqueeze(0)
        bgr = np.stack([np.asarray(f)[:, :, ::-1] for f in tp.tensor2vid_uint8(vid)])
        d = np.abs(bgr.astype(int) - gold["frames_bgr" + tag].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (tag, d.max(), (d > 0).mean())


@pytest.mark.skipif(not rb.reference_available(), reason="/root/reference not mounted (GPU box)")
def test_tensor2vid_port_matches_live_reference_function():
    """fp32 and fp16 videos through the reference's own tensor2vid (t2v_pipeline.py:447-460, imported with stubs)."""
    pl = rb.bootstrap_pipeline()
    vid = torch.randn(2, 3, 3, 16, 24, generator=torch.Generator().manual_seed(23)) * 0.8
    for v in (vid, vid.half()):
        want = np.stack(pl.tensor2vid(v.clone()))
        got = np.stack([np.asarray(f) for f in tp.tensor2vid_uint8(v.clone())])
        assert np.array_equal(got, want)


@pytest.mark.skipif(not rb.reference_available(), reason="/root/reference not mounted (GPU box)")
def test_lvdm_schedule_of_a_halved_model_matches_the_live_reference():
    """The DDIM coefficients of the VideoCrafter path against the reference's own helpers (lvdm/models/modules/util.py:13-63), with the
    product's LatentDiffusion put in fp16 the way bench.py / the tests do it: the schedule must be the reference's fp32 one, bit for bit
    (round 4: `.half()` used to round it to fp16 — 5e-4 of the 50-step output's parity)."""
    import importlib
    from sd_webui_text2video_amd import videocrafter as VC
    rb.bootstrap()
    vu = importlib.import_module("videocrafter.lvdm.models.modules.util")
    betas = vu.make_beta_schedule("linear", 1000, linear_start=configs.LVDM_SCHEDULE["linear_start"],
                                  linear_end=configs.LVDM_SCHEDULE["linear_end"])
    ac = torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)          # ddpm3d.py:141-150 (to_torch = float32)
    ld = VC.LatentDiffusion(configs.TINY_LVDM_UNET, None, init_weights=False, **configs.LVDM_SCHEDULE).half()
    assert ld.alphas_cumprod.dtype == torch.float32 and torch.equal(ld.alphas_cumprod, ac)
    for steps, eta in ((50, 0.0), (10, 1.0)):
        ts = vu.make_ddim_timesteps("uniform", steps, 1000, verbose=False)
        sig, a, a_prev = vu.make_ddim_sampling_parameters(ac.cpu(), ts, eta, verbose=False)
        smp = VC.DDIMSampler(ld)
        smp.make_schedule(steps, ddim_eta=eta, verbose=False)
        assert np.array_equal(smp.ddim_timesteps, ts)
        assert torch.equal(smp.ddim_alphas, torch.as_tensor(a)) and torch.equal(smp.ddim_alphas_prev, torch.as_tensor(a_prev, dtype=torch.float32))
        assert torch.allclose(smp.ddim_sigmas.double(), torch.as_tensor(sig, dtype=torch.float64), rtol=1e-7, atol=0)   # (util.py:58 mixes fp32 tensors and float64 arrays)
