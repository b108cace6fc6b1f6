"""GPU (-m gpu): VideoCrafter (LVDM) path through the C ABI against golden outputs of the REAL reference
(UNetModel forward, DDIM sampling loop with eta noise) and against the travelling oracle."""
import os

import numpy as np
import pytest
import torch

from harness import rel_l2
from oracle import configs, synth, torch_port as tp
from sd_webui_text2video_amd import samplers, vae as V, videocrafter as VC

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _inputs_tiny():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 5, 8, 8, generator=g)
    ctx = torch.randn(2, 9, 768, generator=g)
    x_T = torch.randn(1, 4, 5, 8, 8, generator=g)
    return x, torch.tensor([801, 401]), ctx, x_T


@pytest.fixture(scope="module")
def tiny_ld():
    ld = VC.LatentDiffusion(configs.TINY_LVDM_UNET, dict(ddconfig=configs.TINY_VAE_DDCONFIG, embed_dim=4), image_size=[8, 8],
                            video_length=5, init_weights=False, **configs.LVDM_SCHEDULE)
    net = ld.model.diffusion_model
    sd = synth.synth_state_dict(synth.param_spec(net), seed=0)
    net.load_state_dict(sd, strict=True)
    vsd = synth.synth_state_dict(synth.param_spec(ld.first_stage_model), seed=3)
    ld.first_stage_model.load_state_dict(vsd, strict=True)
    return ld.to(DEV), sd, vsd


def test_tiny_unet_matches_reference_golden(tiny_ld):
    ld, sd, _ = tiny_ld
    net = ld.model.diffusion_model
    x, t, ctx, _ = _inputs_tiny()
    gold = torch.from_numpy(np.load(os.path.join(GOLD, "lvdm_tiny.npz"))["unet_eps"])
    out = net(x.to(DEV), t.to(DEV), context=ctx.to(DEV))
    assert out.dtype == torch.float32 and rel_l2(out.cpu(), gold) < 4e-3
    # one sample at a time == the batched call (batch entries are independent)
    one = net(x[1:2].to(DEV), t[1:2].to(DEV), context=ctx[1:2].to(DEV))
    assert torch.equal(one[0], out[1])
    # fractional / float timesteps and other frame counts, against the travelling oracle
    x3 = torch.randn(1, 4, 16, 8, 8, generator=torch.Generator().manual_seed(3))
    tf = torch.tensor([333.25])
    want = tp.lvdm_unet_forward(sd, configs.TINY_LVDM_UNET, x3, tf, ctx[0:1])
    got = net(x3.to(DEV), tf.to(DEV), context=ctx[0:1].to(DEV))
    assert rel_l2(got.cpu(), want) < 4e-3


def test_tiny_ddim_sampling_matches_reference_golden(tiny_ld):
    """lvdm/samplers/ddim.py loop, 4 steps, CFG 7.5, eta 0.3 (noise from the sampler's seeded CPU generator)."""
    ld, sd, _ = tiny_ld
    _, _, ctx, x_T = _inputs_tiny()
    gold = torch.from_numpy(np.load(os.path.join(GOLD, "lvdm_tiny.npz"))["ddim_x0"])
    smp = VC.DDIMSampler(ld)
    smp.noise_gen.manual_seed(123)
    seen = []
    x0, inter = smp.sample(S=4, conditioning={"c_crossattn": [ctx[0:1].to(DEV)]}, batch_size=1, shape=list(x_T.shape[1:]),
                           verbose=False, unconditional_guidance_scale=7.5,
                           unconditional_conditioning={"c_crossattn": [ctx[1:2].to(DEV)]}, eta=0.3, x_T=x_T.to(DEV),
                           callback=lambda i: seen.append(i))
    assert seen == [0, 1, 2, 3]
    r = rel_l2(x0.cpu(), gold)
    assert r < 2e-2, r
    assert len(inter["pred_x0"]) == 3 and torch.isfinite(inter["pred_x0"][-1]).all()
    samplers.state.interrupted = True
    try:
        with pytest.raises(samplers.InterruptedException):
            smp.sample(S=2, conditioning=ctx[0:1].to(DEV), batch_size=1, shape=list(x_T.shape[1:]), verbose=False, x_T=x_T.to(DEV))
    finally:
        samplers.state.interrupted = False


def test_sample_text2video_entry_point(tiny_ld):
    """sample_text2video.py:92-152: conditions -> DDIM -> decode_first_stage -> uint8 [n, T, H, W, 3]."""
    ld, sd, vsd = tiny_ld
    _, _, ctx, _ = _inputs_tiny()

    class Enc:            # stands in for FrozenCLIPEmbedder (outside the hot path)
        def encode(self, prompts):
            return (ctx[0:1] if prompts[0] == "a cat" else ctx[1:2]).to(DEV).repeat(len(prompts), 1, 1)
    ld.cond_stage_model = Enc()
    smp = VC.DDIMSampler(ld)
    smp.noise_gen.manual_seed(5)
    torch.manual_seed(0)
    vids = VC.sample_text2video(ld, "a cat", "", 1, 1, sampler=smp, ddim_steps=4, eta=0.0, cfg_scale=7.5, decode_frame_bs=2,
                                num_frames=5)
    assert vids.shape == (1, 5, 64, 64, 3) and vids.dtype == np.uint8
    # decode path == the oracle's VAE decode of the same latent
    torch.manual_seed(0)
    smp.noise_gen.manual_seed(5)
    lat, _ = smp.sample(S=4, conditioning={"c_crossattn": [ctx[0:1].to(DEV)]}, batch_size=1, shape=[4, 5, 8, 8], verbose=False,
                        unconditional_guidance_scale=7.5, unconditional_conditioning={"c_crossattn": [ctx[1:2].to(DEV)]}, eta=0.0)
    z = (lat / ld.scale_factor)[0].permute(1, 0, 2, 3).float().cpu()
    want = tp.vae_decode(vsd, configs.TINY_VAE_DDCONFIG, z)
    want_u8 = ((want + 1) * 127.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    diff = np.abs(vids[0].astype(np.int32) - want_u8.astype(np.int32))
    assert diff.mean() < 1.0 and np.percentile(diff, 99) <= 3
    # n_samples = batch_size = 2: one 4-row UNet batch and one fused update per step; video 0 starts from the same noise draw
    # order as the single-video run only in its first element, so compare against a batch-2 oracle-free invariant instead:
    # both videos valid, different from each other, and the latent of a batch equals two single runs from the same x_T
    g = torch.Generator().manual_seed(21)
    xT = torch.randn(2, 4, 5, 8, 8, generator=g).to(DEV)
    cond = {"c_crossattn": [ctx[0:1].to(DEV).repeat(2, 1, 1)]}
    unc = {"c_crossattn": [ctx[1:2].to(DEV).repeat(2, 1, 1)]}
    lat2, _ = smp.sample(S=4, conditioning=cond, batch_size=2, shape=[4, 5, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                         unconditional_conditioning=unc, eta=0.0, x_T=xT)
    for v in range(2):
        lat1, _ = smp.sample(S=4, conditioning={"c_crossattn": [ctx[0:1].to(DEV)]}, batch_size=1, shape=[4, 5, 8, 8], verbose=False,
                             unconditional_guidance_scale=7.5, unconditional_conditioning={"c_crossattn": [ctx[1:2].to(DEV)]},
                             eta=0.0, x_T=xT[v:v + 1])
        assert rel_l2(lat2[v:v + 1].float().cpu(), lat1.float().cpu()) < 3e-3, v
    vids2 = VC.sample_text2video(ld, "a cat", "", 2, 2, sampler=smp, ddim_steps=4, eta=0.0, cfg_scale=7.5, decode_frame_bs=2, num_frames=5)
    assert vids2.shape == (2, 5, 64, 64, 3) and np.abs(vids2[0].astype(np.int32) - vids2[1].astype(np.int32)).mean() > 1.0


def test_released_config_forward_matches_reference_golden():
    """BASELINE.json configs[4]: VideoCrafter base UNet (head_dim 40 / 80 / 160), 16 frames @ 32x32 latent, 77 tokens."""
    gold = np.load(os.path.join(GOLD, "lvdm_16f.npz"))["unet_eps"]
    net = VC.UNetModel(**configs.LVDM_UNET, init_weights=False)
    sd = synth.synth_state_dict(synth.param_spec(net), seed=0)
    net.load_state_dict(sd, strict=True)
    del sd
    net = net.to(DEV)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, 4, 16, 32, 32, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g)
    out = net(x.to(DEV), torch.tensor([500], device=DEV), context=ctx.to(DEV))
    r = rel_l2(out.float().cpu(), torch.from_numpy(gold))
    assert r < 4e-3, r
    # fp16 weights + fp16 latent (the deployment precision): eps comes back fp16
    net = net.half()
    out16 = net(x.half().to(DEV), torch.tensor([500], device=DEV), context=ctx.half().to(DEV))
    assert out16.dtype == torch.float16
    r16 = rel_l2(out16.float().cpu(), torch.from_numpy(gold))
    assert r16 < 8e-3, r16
