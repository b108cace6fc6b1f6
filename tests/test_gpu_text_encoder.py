"""GPU (-m gpu): CLIP text towers through the C ABI vs the fp32 oracle."""
import pytest
import torch

from harness import rel_l2
from oracle import torch_port as tp
from sd_webui_text2video_amd import text_encoder as TE
from test_text_encoder_cpu import TINY, _seed_params, _tokens

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_tiny_open_clip_tower_taps():
    m = _seed_params(TE.OpenClipTextModel(**TINY), 4)
    tower = TE.ClipTextTower(m, heads=2, act="gelu", skip_last=1)
    tok = _tokens(2, 77, TINY["vocab_size"], seed=5)
    z = tower(tok.to(DEV)).cpu()
    ref = tp.clip_text_forward(m.state_dict(), tok, heads=2, layers=2)
    assert rel_l2(z, ref) < 3e-3
    tok2 = tok.clone()
    tok2[:, 40:] = (tok2[:, 40:] + 7) % TINY["vocab_size"]
    z2 = tower(tok2.to(DEV)).cpu()
    assert torch.equal(z[:, :40], z2[:, :40]) and not torch.equal(z[:, 40:], z2[:, 40:])


def test_vit_h_text_tower_penultimate_and_process_tokens():
    """The ModelScope configuration: OpenCLIP ViT-H-14 text tower (354 M parameters, seeded synthetic weights),
    layer='penultimate', cond + uncond chunk in one batch; then the emphasis path of process_tokens."""
    m = _seed_params(TE.OpenClipTextModel(**TE.OPEN_CLIP_TEXT["ViT-H-14"]), 9)
    emb = TE.FrozenOpenCLIPEmbedder(model=m, layer="penultimate", device=DEV)
    tok = _tokens(2, 77, 49408, seed=10)
    tok[:, 0], tok[0, 12:], tok[1, 1:] = 49406, 49407, 49407
    z = emb.encode_with_transformers(tok.to(DEV)).cpu()
    ref = tp.clip_text_forward(m.state_dict(), tok, heads=16, layers=23)
    assert z.shape == (2, 77, 1024) and z.dtype == torch.float32
    assert rel_l2(z, ref) < 4e-3
    mult = torch.ones(2, 77)
    mult[0, 3:6] = 1.21
    zp = emb.process_tokens(tok.tolist(), mult.tolist()).cpu()
    tokp = tok.clone()
    tokp[0, 13:], tokp[1, 2:] = 0, 0
    refp = tp.clip_process_tokens(tp.clip_text_forward(m.state_dict(), tokp, heads=16, layers=23), mult)
    assert rel_l2(zp, refp) < 4e-3
    # weights mutated in place are picked up
    with torch.no_grad():
        m.ln_final.weight.mul_(2.0)
    z2 = emb.encode_with_transformers(tok.to(DEV)).cpu()
    b = m.ln_final.bias.detach()
    assert rel_l2(z2 - b, 2 * (z - b)) < 2e-3


def test_hf_clip_l_tower():
    """The VideoCrafter configuration: transformers CLIPTextModel with the openai/clip-vit-large-patch14 text config
    (12 layers, width 768, quick-GELU), seeded synthetic weights; reference = transformers' own forward (fp32, CPU)."""
    transformers = pytest.importorskip("transformers")
    cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                      num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu")
    m = _seed_params(transformers.CLIPTextModel(cfg).eval(), 11)
    fe = TE.FrozenCLIPEmbedder(transformer=m, tokenizer=object(), device=DEV)
    tok = _tokens(3, 77, 49408, seed=12)
    z = fe.encode_tokens(tok).cpu()
    with torch.no_grad():
        ref = m(input_ids=tok).last_hidden_state
    assert rel_l2(z, ref) < 4e-3
    assert rel_l2(tp.clip_text_forward(m.state_dict(), tok, heads=12, layers=12, act="quick_gelu", naming="hf",
                                       prefix=fe._tower.names.p), ref) < 2e-6
