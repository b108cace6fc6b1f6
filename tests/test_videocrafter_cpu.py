"""CPU: VideoCrafter (LVDM) path — oracle restatement pinned against golden outputs of the REAL reference
(tests/golden/make_golden.py:lvdm), product module topology / state-dict keys, the lowering executed in the
CPU interpreter, and the LVDM DDIM sampler's host logic."""
import importlib
import os
import types

import numpy as np
import pytest
import torch

from harness import rel_l2
from interp import Interp
from oracle import configs, ref_bootstrap as rb, synth, torch_port as tp
from sd_webui_text2video_amd import _lib as L
from sd_webui_text2video_amd import samplers, videocrafter as VC
from test_samplers_cpu import _ddim_update_cpu, _lincomb_cpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _inputs_tiny():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 5, 8, 8, generator=g)
    ctx = torch.randn(2, 9, 768, generator=g)
    x_T = torch.randn(1, 4, 5, 8, 8, generator=g)
    return x, torch.tensor([801, 401]), ctx, x_T


@pytest.fixture(scope="module")
def tiny():
    net = VC.UNetModel(**configs.TINY_LVDM_UNET, init_weights=False)
    sd = synth.synth_state_dict(synth.param_spec(net), seed=0)
    net.load_state_dict(sd, strict=True)
    return net, sd


def test_port_matches_reference_golden(tiny):
    net, sd = tiny
    gold = np.load(os.path.join(GOLD, "lvdm_tiny.npz"))
    x, t, ctx, x_T = _inputs_tiny()
    eps = tp.lvdm_unet_forward(sd, configs.TINY_LVDM_UNET, x, t, ctx)
    assert np.abs(eps.numpy() - gold["unet_eps"]).max() < 2e-5
    gen = torch.Generator().manual_seed(123)
    x0 = tp.lvdm_ddim_sample(lambda a, b, c: tp.lvdm_unet_forward(sd, configs.TINY_LVDM_UNET, a, b, c), x_T, 4, ctx[0:1],
                             ctx[1:2], 7.5, eta=0.3, noise_gen=gen)
    assert np.abs(x0.numpy() - gold["ddim_x0"]).max() < 2e-4 * np.abs(gold["ddim_x0"]).max()


@pytest.mark.skipif(not rb.reference_available(), reason="reference not mounted")
def test_state_dict_keys_match_reference():
    rb.bootstrap()
    om = importlib.import_module("videocrafter.lvdm.models.modules.openaimodel3d")
    for cfg in (configs.TINY_LVDM_UNET, dict(configs.LVDM_UNET, model_channels=320, num_res_blocks=1)):
        with torch.device("meta"):          # keys and shapes only: skips the reference's (slow) default initialisation
            ref = om.UNetModel(**cfg)
        mine = VC.UNetModel(**cfg, init_weights=False)
        rs, ms = ref.state_dict(), mine.state_dict()
        assert list(rs.keys()) == list(ms.keys())
        assert all(tuple(rs[k].shape) == tuple(ms[k].shape) for k in rs)


def test_layout_of_released_config():
    net = VC.UNetModel(**configs.LVDM_UNET, init_weights=False)
    sd = net.state_dict()
    assert len(sd) > 600                                # every block present
    assert sd["input_blocks.1.1.transformer_blocks.0.attn1_tmp.relative_position_k.embeddings_table"].shape == (33, 40)
    assert sd["middle_block.1.transformer_blocks.0.attn2.to_k.weight"].shape == (1280, 768)
    assert sd["output_blocks.2.1.conv.weight"].shape == (1280, 1280, 1, 3, 3)       # Upsample after the 4x4 level
    assert sd["input_blocks.3.0.op.weight"].shape == (320, 320, 1, 3, 3)
    with pytest.raises(NotImplementedError):
        VC.UNetModel(**dict(configs.LVDM_UNET, kernel_size_t=3, padding_t=1), init_weights=False)
    with pytest.raises(NotImplementedError):
        VC.UNetModel(**dict(configs.LVDM_UNET, model_channels=64), init_weights=False)  # head_dim 8: no kernel


def test_program_matches_golden_in_interpreter(tiny):
    net, sd = tiny
    x, t, ctx, _ = _inputs_tiny()
    comp = net._compile(2, 5, 8, 8, 9, "f32", "f32", "f32")
    packed = comp.packer.materialise(net.state_dict(), "cpu")
    out = torch.empty(2, 4, 5, 8, 8)
    Interp(comp.prog, packed).run({L.EXT_X: x, L.EXT_T: t.float(), L.EXT_CTX: ctx, L.EXT_OUT: out})
    gold = torch.from_numpy(np.load(os.path.join(GOLD, "lvdm_tiny.npz"))["unet_eps"])
    assert rel_l2(out, gold) < 4e-3
    kinds = [op.kind for op in comp.prog.ops]
    n_st = sum(1 for p in net.state_dict() if p.endswith("attn1_tmp.to_q.weight"))      # transformer sites
    assert kinds.count(L.OP_RELPOS_ATTN) == 2 * n_st and kinds.count(L.OP_ATTENTION) == 2 * n_st


def test_ddim_sampler_host_logic(tiny, monkeypatch):
    net, sd = tiny
    monkeypatch.setattr(samplers, "_lincomb", _lincomb_cpu)
    monkeypatch.setattr(samplers, "_ddim_update", _ddim_update_cpu)
    gold = np.load(os.path.join(GOLD, "lvdm_tiny.npz"))
    _, _, ctx, x_T = _inputs_tiny()
    ld = VC.LatentDiffusion.__new__(VC.LatentDiffusion)
    torch.nn.Module.__init__(ld)
    VC.LatentDiffusion.register_schedule(ld, **configs.LVDM_SCHEDULE)
    calls = []

    def apply_model(x, t, c, **kw):
        calls.append((x.shape[0], t.tolist()))
        return tp.lvdm_unet_forward(sd, configs.TINY_LVDM_UNET, x, t, c)
    ld.apply_model = apply_model
    ld.model = types.SimpleNamespace(diffusion_model=types.SimpleNamespace(refresh_weights=lambda d: None, auto_refresh=True))
    smp = VC.DDIMSampler(ld)
    smp.noise_gen.manual_seed(123)
    seen = []
    x0, inter = smp.sample(S=4, conditioning={"c_crossattn": [ctx[0:1]]}, batch_size=1, shape=list(x_T.shape[1:]), verbose=False,
                           unconditional_guidance_scale=7.5, unconditional_conditioning={"c_crossattn": [ctx[1:2]]}, eta=0.3,
                           x_T=x_T, callback=lambda i: seen.append(i))
    assert np.abs(x0.numpy() - gold["ddim_x0"]).max() < 2e-4 * np.abs(gold["ddim_x0"]).max()
    assert seen == [0, 1, 2, 3] and [c[1][0] for c in calls] == [751, 501, 251, 1] and all(c[0] == 2 for c in calls)
    assert len(inter["x_inter"]) == len(inter["pred_x0"]) == 3 and samplers.state.sampling_step == 3
    with pytest.raises(NotImplementedError):
        smp.sample(S=4, conditioning=ctx[0:1], batch_size=1, shape=list(x_T.shape[1:]), mask=torch.ones(1), x0=x_T)
    # batch_size = 2 (sample_text2video's batch): video 0 of the batch = the single-video result (eta = 0: no noise draw)
    x1, _ = smp.sample(S=4, conditioning={"c_crossattn": [ctx[0:1]]}, batch_size=1, shape=list(x_T.shape[1:]), verbose=False,
                       unconditional_guidance_scale=7.5, unconditional_conditioning={"c_crossattn": [ctx[1:2]]}, eta=0.0, x_T=x_T)
    g = torch.Generator().manual_seed(9)
    xT2 = torch.cat([x_T, torch.randn(x_T.shape, generator=g)], dim=0)
    calls.clear()
    x2, _ = smp.sample(S=4, conditioning={"c_crossattn": [ctx[0:1].repeat(2, 1, 1)]}, batch_size=2, shape=list(x_T.shape[1:]),
                       verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning={"c_crossattn": [ctx[1:2].repeat(2, 1, 1)]}, eta=0.0, x_T=xT2)
    assert x2.shape[0] == 2 and all(c[0] == 4 and len(c[1]) == 4 for c in calls)
    assert (x2[0:1] - x1).abs().max() < 1e-5 * x1.abs().max()
    assert (x2[1] - x2[0]).abs().max() > 0.1


def test_entry_point_helpers():
    x = torch.tensor([-1.0, -0.5, 0.0, 0.999, 1.0, 2.0]).view(1, 1, 1, 1, 6).repeat(1, 3, 2, 1, 1)
    out = VC.torch_to_np(x)
    assert out.shape == (1, 2, 1, 6, 3) and out.dtype == torch.uint8
    assert out[0, 0, 0, :, 0].tolist() == [0, 63, 127, 254, 255, 255]       # ((x+1)*127.5) clamped, truncated
    m = types.SimpleNamespace(image_size=[32, 32], model=types.SimpleNamespace(diffusion_model=types.SimpleNamespace(in_channels=4, temporal_length=16)))
    assert VC.make_model_input_shape(m, 1) == [1, 4, 16, 32, 32] and VC.make_model_input_shape(m, 2, T=24) == [2, 4, 24, 32, 32]


def test_half_keeps_the_schedule_in_fp32():
    """`.half()` is how the networks are put in fp16; the DDIM coefficients must keep coming from the reference's fp32 schedule
    (ddpm3d.py:125-165 registers fp32 buffers and VideoCrafter never halves the module).  An fp16 alphas_cumprod is a per-step
    coefficient error shared by every pixel: it was the part of configs[4]'s 50-step error that did not average out."""
    ld = VC.LatentDiffusion(configs.TINY_LVDM_UNET, None, init_weights=False, **configs.LVDM_SCHEDULE)
    ref = {n: getattr(ld, n).clone() for n in ld._schedule_names}
    ld = ld.half()
    assert next(ld.parameters()).dtype == torch.float16
    for n, v in ref.items():
        assert getattr(ld, n).dtype == torch.float32 and torch.equal(getattr(ld, n), v), n
    smp = VC.DDIMSampler(ld)
    smp.make_schedule(50, verbose=False)
    assert smp.ddim_alphas.dtype == torch.float32
    assert torch.equal(smp.ddim_alphas, ref["alphas_cumprod"][smp.ddim_timesteps])
