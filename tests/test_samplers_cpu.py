"""CPU: host logic of the "DDIM" and "UniPC" solvers (schedules, multistep coefficients, history
handling, vid2vid noising) against golden outputs of the REAL reference (tests/golden/make_golden.py).

The two device helpers the solvers launch (`_lincomb` = T2V_OP_LINCOMB, `_ddim_update` =
T2V_OP_DDIM_STEP) are replaced by torch restatements of the kernels' documented semantics
(include/t2v_hip.h) and the UNet by the oracle port, so only the product's own host arithmetic is
under test here; the kernels themselves are checked on the GPU (tests/test_gpu_ops.py, test_gpu_e2e.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import configs, synth, torch_port as tp
from sd_webui_text2video_amd import samplers

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _lincomb_cpu(out, terms):
    acc = torch.zeros(out.shape, dtype=torch.float32)
    for c, t in terms:
        acc = acc + np.float32(c) * t.reshape(out.shape).float()
    out.copy_(acc.to(out.dtype))
    return out


def _ddim_update_cpu(out, xt, eps_pair, noise, coef, guided, mode):
    f = [torch.tensor(c, dtype=torch.float32) for c in coef]
    nb = xt.shape[0]                                     # videos per batch: eps_pair = [cond (nb), uncond (nb)]
    e = eps_pair[0:nb].float()
    if guided:
        u = eps_pair[nb:2 * nb].float()
        e = torch.cat([u[:, :guided] + f[5] * (e[:, :guided] - u[:, :guided]), e[:, guided:]], dim=1)
    assert mode == 1
    x0 = (xt.float() - f[0] * e) / f[1]
    r = f[2] * x0 + f[3] * e
    if noise is not None and float(f[4]) != 0.0:
        r = r + f[4] * noise
    out.copy_(r.to(out.dtype))
    return out


class _PortModel:
    """Stands in for UNetSD: oracle forward, batched cond/uncond like the product."""
    supports_cfg_batch = True

    def __init__(self, batched=True):
        from sd_webui_text2video_amd import unet
        spec = synth.param_spec(unet.UNetSD(**configs.TINY_UNET, init_weights=False))
        self.sd = synth.synth_state_dict(spec, seed=0)
        self.num_timesteps = 1000
        self.supports_cfg_batch = batched
        self.calls = []

    def __call__(self, x, t, c):
        self.calls.append((x.shape[0], t.tolist()))
        return tp.unet_forward(self.sd, configs.TINY_UNET, x.float(), t, c)


@pytest.fixture()
def cpu_kernels(monkeypatch):
    monkeypatch.setattr(samplers, "_lincomb", _lincomb_cpu)
    monkeypatch.setattr(samplers, "_ddim_update", _ddim_update_cpu)
    samplers.state.sampling_step = 0


def _inputs():
    g = torch.Generator().manual_seed(5)
    torch.randn(2, 4, 3, 16, 16, generator=g); torch.randn(2, 7, 1024, generator=g); torch.randn(2, 4, 8, 8, generator=g)
    c = torch.randn(1, 7, 1024, generator=g)
    uc = torch.randn(1, 7, 1024, generator=g)
    noise, _, _ = synth.synth_inputs(3, 128, 128)
    return noise, c, uc


def _rel(a, g):
    return float(np.abs(a.numpy() - g).max() / np.abs(g).max())


@pytest.mark.parametrize("batched", [True, False])
def test_ddim_host_logic_matches_reference(cpu_kernels, batched):
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    noise, c, uc = _inputs()
    model = _PortModel(batched)
    betas = tp.beta_schedule_linear_sd()
    smp = samplers.Txt2VideoSampler(model, torch.device("cpu"), betas=betas, sampler_name="DDIM")
    x0 = smp.sample_loop(steps=4, strength=None, conditioning=c, unconditional_conditioning=uc, batch_size=1,
                         latents=None, shape=tuple(noise.shape), noise=noise, guidance_scale=9.0, eta=0.0,
                         sampler_name="DDIM")
    assert _rel(x0, gold["ddim_x0"]) < 2e-5
    assert samplers.state.sampling_step == 4
    # time grid 1 + arange(0, 1000, 250), walked backwards; one (batched) model call per step
    assert [cl[1][0] for cl in model.calls][:: (1 if batched else 2)] == [751, 501, 251, 1]
    assert len(model.calls) == (4 if batched else 8)


def test_ddim_vid2vid_encode_and_decode(cpu_kernels):
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    noise, c, uc = _inputs()
    z0 = torch.randn(noise.shape, generator=torch.Generator().manual_seed(11))
    model = _PortModel()
    smp = samplers.Txt2VideoSampler(model, torch.device("cpu"), betas=tp.beta_schedule_linear_sd(), sampler_name="DDIM")
    enc, dsteps = smp.encode_latent(z0, noise, 0.75, 4)
    assert dsteps == 3 and _rel(enc, gold["ddim_encode"]) < 1e-6
    # after encode_latent, `sample` IS `decode` (samplers_common.py:137)
    x0 = smp.sample_loop(steps=4, strength=0.75, conditioning=c, unconditional_conditioning=uc, batch_size=1,
                         latents=z0, shape=tuple(noise.shape), noise=noise, is_vid2vid=True, guidance_scale=9.0,
                         eta=0.0, sampler_name="DDIM")
    assert _rel(x0, gold["ddim_vid2vid_x0"]) < 2e-5


def test_unipc_host_logic_matches_reference(cpu_kernels):
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    noise, c, uc = _inputs()
    model = _PortModel()
    betas = tp.beta_schedule_linear_sd()
    smp = samplers.Txt2VideoSampler(model, torch.device("cpu"), betas=betas, sampler_name="UniPC")
    x0 = smp.sample_loop(steps=6, strength=None, conditioning=c, unconditional_conditioning=uc, batch_size=1,
                         latents=None, shape=tuple(noise.shape), noise=noise, guidance_scale=9.0, eta=0.0,
                         sampler_name="UniPC")
    assert _rel(x0, gold["unipc_x0"]) < 2e-5
    assert len(model.calls) == 6 and samplers.state.sampling_step == 6     # S model evaluations, S callbacks
    assert abs(model.calls[0][1][0] - 999.0) < 1e-3 and abs(model.calls[-1][1][0] - 1000.0 * (1 - 5 / 6 * 0.999) + 1) < 0.5
    samplers.state.sampling_step = 0
    x0 = smp.sample_loop(steps=4, strength=0.7, conditioning=c, unconditional_conditioning=uc, batch_size=1,
                         latents=None, shape=tuple(noise.shape), noise=noise, guidance_scale=7.0, eta=0.0,
                         sampler_name="UniPC")
    assert _rel(x0, gold["unipc_x0_s07"]) < 2e-5


def test_unipc_unconditional_and_encode(cpu_kernels):
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    noise, c, uc = _inputs()
    model = _PortModel()
    betas = tp.beta_schedule_linear_sd()
    smp = samplers.Txt2VideoSampler(model, torch.device("cpu"), betas=betas, sampler_name="UniPC")
    z0 = torch.randn(noise.shape, generator=torch.Generator().manual_seed(11))
    enc = smp.sampler.unipc_encode(z0, torch.device("cpu"), 0.7, 4, noise=noise)
    assert _rel(enc, gold["unipc_encode"]) < 1e-6
    # guidance 1.0 -> a single conditional evaluation per step; oracle restatement as the target
    x0 = smp.sampler.sample(S=3, batch_size=1, shape=tuple(noise.shape), conditioning=c, x_T=noise,
                            unconditional_guidance_scale=1.0, unconditional_conditioning=uc)
    ref = tp.unipc_sample(lambda a, b, cc: tp.unet_forward(model.sd, configs.TINY_UNET, a, b, cc), betas, noise, 3, c, uc, 1.0)
    assert _rel(x0, ref.numpy()) < 2e-5
    assert all(cl[0] == 1 for cl in model.calls)


def test_vp_schedule_matches_oracle():
    betas = tp.beta_schedule_linear_sd()
    ac = torch.cumprod(1 - betas, 0)
    mine, ref = samplers._VPSchedule(ac), tp.VPDiscrete(ac)
    for t in (1.0, 0.7, 0.5005, 0.25, 0.0015, 0.001):
        assert abs(mine.lam(t) - float(ref.lam(t))) < 2e-5 * max(1.0, abs(mine.lam(t)))
        assert abs(mine.alpha(t) - float(ref.alpha(t))) < 1e-6 and abs(mine.std(t) - float(ref.std(t))) < 1e-6


def test_shared_noise_slices_of_one_stream():
    """eta > 0 with one clip split over ranks (VERDICT r03 missing #3): every rank draws the WHOLE clip's per-step noise from an
    identically seeded generator and keeps its frames — the slices of all ranks tile the single-device draw, step after step."""
    from sd_webui_text2video_amd.samplers import SharedNoise
    F, counts = 10, (4, 4, 2)
    whole = SharedNoise(77, F, 0, "cpu")
    parts, off = [], 0
    for c in counts:
        parts.append((SharedNoise(77, F, off, "cpu"), c))
        off += c
    for _ in range(3):                                   # three sampling steps
        full = whole.draw(torch.empty(1, 4, F, 2, 2))
        got = torch.cat([sn.draw(torch.empty(1, 4, c, 2, 2)) for sn, c in parts], dim=2)
        assert torch.equal(full, got)
    other = SharedNoise(78, F, 0, "cpu").draw(torch.empty(1, 4, F, 2, 2))
    assert not torch.equal(other, whole.draw(torch.empty(1, 4, F, 2, 2)))
