"""CPU: CLIP text towers.  (1) the oracle restatement is pinned against torch.nn.MultiheadAttention
(open_clip's block, the published architecture) and against transformers' own CLIPTextModel (what the VideoCrafter
embedder calls); (2) the product lowering, executed by the CPU interpreter, matches the oracle; (3) host logic of the
embedder mirror (chunking, padding after <end>, emphasis multipliers)."""
import pytest
import torch

from harness import rel_l2
from interp import Interp
from oracle import torch_port as tp
from sd_webui_text2video_amd import _lib as L
from sd_webui_text2video_amd import text_encoder as TE

TINY = dict(width=128, heads=2, layers=3, vocab_size=500, context_length=77)


def _seed_params(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.6 / p.shape[-1] ** 0.5 if "embedding" not in n else 0.5))
            elif n.endswith("weight") and ("ln" in n or "norm" in n):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return m


def _tokens(B, Lseq, vocab, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(1, vocab, (B, Lseq), generator=g)


def test_oracle_matches_multihead_attention_blocks():
    """open_clip's ResidualAttentionBlock.forward: x + attn(ln_1(x), need_weights=False, attn_mask=mask)[0]; x + mlp(ln_2(x)),
    sequence-first, as clip_hardcode.py:110-117 drives it."""
    m = _seed_params(TE.OpenClipTextModel(**TINY), 1)
    tok = _tokens(2, 77, TINY["vocab_size"])
    with torch.no_grad():
        x = m.token_embedding(tok) + m.positional_embedding
        x = x.permute(1, 0, 2)
        for i, r in enumerate(m.transformer.resblocks):
            if i == len(m.transformer.resblocks) - 1:          # layer='penultimate'
                break
            h = r.ln_1(x)
            x = x + r.attn(h, h, h, need_weights=False, attn_mask=m.attn_mask)[0]
            x = x + r.mlp(r.ln_2(x))
        ref = m.ln_final(x.permute(1, 0, 2))
    got = tp.clip_text_forward(m.state_dict(), tok, heads=TINY["heads"], layers=TINY["layers"] - 1)
    assert rel_l2(got, ref) < 2e-6


def test_oracle_matches_transformers_clip_text_model():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.CLIPTextConfig(vocab_size=500, hidden_size=128, intermediate_size=512, num_hidden_layers=3,
                                      num_attention_heads=2, max_position_embeddings=77, hidden_act="quick_gelu")
    m = _seed_params(transformers.CLIPTextModel(cfg).eval(), 2)
    tok = _tokens(2, 77, 500, seed=3)
    with torch.no_grad():
        ref = m(input_ids=tok).last_hidden_state
    sd = m.state_dict()
    prefix = next(k for k in sd if k.endswith("embeddings.token_embedding.weight"))[: -len("embeddings.token_embedding.weight")]
    got = tp.clip_text_forward(sd, tok, heads=2, layers=3, act="quick_gelu", naming="hf", prefix=prefix)
    assert rel_l2(got, ref) < 2e-6


def _run_interp(tower, tok):
    comp = tower._compile(*tok.shape)
    packed = comp.packer.materialise(tower.holder.state_dict(), "cpu")
    z = torch.empty(tok.shape + (tower.width,), dtype=torch.float32)
    Interp(comp.prog, packed).run({L.EXT_X: tok.to(torch.int32).contiguous(), L.EXT_OUT: z})
    return z, comp


def test_open_clip_lowering_matches_oracle():
    m = _seed_params(TE.OpenClipTextModel(**TINY), 4)
    tower = TE.ClipTextTower(m, heads=2, act="gelu", skip_last=1)
    assert tower.names.kind == "open_clip" and tower.n_layers == 3
    tok = _tokens(2, 77, TINY["vocab_size"], seed=5)
    z, comp = _run_interp(tower, tok)
    ref = tp.clip_text_forward(m.state_dict(), tok, heads=2, layers=2)
    assert rel_l2(z, ref) < 3e-3
    kinds = [op.kind for op in comp.prog.ops]
    assert kinds.count(L.OP_ATTENTION) == 2 and kinds[0] == L.OP_EMBED_ROWS
    assert all(op.i[15] == 1 for op in comp.prog.ops if op.kind == L.OP_ATTENTION)
    # causality end to end: changing a later token must not move earlier positions
    tok2 = tok.clone()
    tok2[:, 40:] = (tok2[:, 40:] + 7) % TINY["vocab_size"]
    z2, _ = _run_interp(tower, tok2)
    assert torch.equal(z[:, :40], z2[:, :40]) and not torch.equal(z[:, 40:], z2[:, 40:])


def test_hf_lowering_matches_oracle_short_sequence():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.CLIPTextConfig(vocab_size=500, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                      num_attention_heads=2, max_position_embeddings=77, hidden_act="quick_gelu")
    m = _seed_params(transformers.CLIPTextModel(cfg).eval(), 6)
    tower = TE.ClipTextTower(m)
    assert tower.names.kind == "hf" and tower.heads == 2 and tower.act == TE.ACT_QUICK_GELU
    tok = _tokens(3, 20, 500, seed=7)                     # ragged: shorter than the context length
    z, _ = _run_interp(tower, tok)
    with torch.no_grad():
        ref = m(input_ids=tok).last_hidden_state
    assert rel_l2(z, ref) < 3e-3


class _Tok:
    encoder = {",</w>": 267, "<start_of_text>": 498, "<end_of_text>": 499}

    def encode(self, text):
        return [1 + (ord(c) % 400) for c in text]


def test_embedder_host_logic(monkeypatch):
    m = _seed_params(TE.OpenClipTextModel(**TINY), 8)
    emb = TE.FrozenOpenCLIPEmbedder(model=m, layer="penultimate", tokenizer=_Tok(), device="cpu")
    assert emb.layer_idx == 1 and emb.id_start == 498 and emb.id_end == 499
    chunks, count = emb.tokenize_line("x" * 80)
    assert len(chunks) == 2 and count == 80
    assert all(len(t) == 77 and t[0] == 498 and t[-1] == 499 and len(w) == 77 for t, w in chunks)
    assert chunks[1][0][6:] == [499] * 71
    assert emb.tokenize_line("")[0][0][0] == ([498] + [499] * 76)
    with pytest.raises(L.T2VError):
        TE.FrozenOpenCLIPEmbedder(model=m, device="cpu").tokenize(["a"])
    # process_tokens: pad after the first <end>, multipliers + mean restoration (transformer stubbed by the oracle)
    seen = {}

    def fake(tokens):
        seen["tokens"] = tokens.clone()
        return tp.clip_text_forward(m.state_dict(), tokens.long(), heads=2, layers=2)
    monkeypatch.setattr(emb, "encode_with_transformers", fake)
    toks, mult = chunks[1]
    mult = list(mult)
    mult[2] = 1.3
    z = emb.process_tokens([toks], [mult])
    assert seen["tokens"][0, :7].tolist() == toks[:7] and seen["tokens"][0, 7:].tolist() == [0] * 70
    zr = tp.clip_text_forward(m.state_dict(), seen["tokens"].long(), heads=2, layers=2)
    assert rel_l2(z, tp.clip_process_tokens(zr, torch.tensor([mult]))) < 1e-6
    # forward: two chunks side by side
    monkeypatch.setattr(emb, "process_tokens", lambda t, w: torch.zeros(len(t), 77, 128))
    assert emb(["x" * 80, "y"]).shape == (2, 154, 128)


def test_device_only():
    m = TE.OpenClipTextModel(**TINY)
    with pytest.raises(L.T2VError):
        TE.ClipTextTower(m, heads=2)(torch.zeros(1, 77, dtype=torch.long))
