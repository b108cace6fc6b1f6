"""Text-encoder conditioning on the GPU: the CLIP text towers that produce the `cond` tensor the
denoising loop consumes, lowered to the same denoise-program IR as the UNet and executed by libt2v_hip.so.

Reference call sites:
* ModelScope — `FrozenOpenCLIPEmbedder` (scripts/modelscope/clip_hardcode.py:59-422): OpenCLIP ViT-H-14 text tower,
  `layer='penultimate'` (t2v_pipeline.py:137-141): `encode_with_transformer` (:110-117) = token + positional embedding →
  the first `len(resblocks) - layer_idx` residual attention blocks under the causal mask (:269-274) → `ln_final`;
  `process_tokens` (:392-421) applies the emphasis multipliers and restores the mean.
* VideoCrafter — `FrozenCLIPEmbedder` (lvdm/models/modules/condition_modules.py:15-42): HF `CLIPTextModel`
  (openai/clip-vit-large-patch14, quick-GELU), `last_hidden_state`.
`open_clip` itself is a third-party dependency that the reference does not vendor and this image does not carry; its
text tower is the published pre-LN transformer (`ResidualAttentionBlock`: x += MHA(ln_1 x, causal); x += c_proj(GELU(c_fc
(ln_2 x)))) with `nn.MultiheadAttention` parameter names, which `OpenClipTextModel` below reproduces as a parameter
holder so that `open_clip_pytorch_model.bin` state dicts load unchanged.  Tokenisation, the prompt-emphasis parser and
textual-inversion fixes are webui plumbing (`modules.prompt_parser`, `modules.textual_inversion`) and stay outside: the
boundary is a batch of token ids.

One block of the tower per 77-token chunk: LayerNorm → one QKV GEMM (bias) → causal attention (`T2V_OP_ATTENTION`,
i[15] = 1, 64-wide heads) → out-projection GEMM with bias + fp32 residual → LayerNorm → fc GEMM → GELU / quick-GELU
(`T2V_OP_COPY2D` act 2 / 3) → projection GEMM with bias + residual.  The token lookup is `T2V_OP_EMBED_ROWS`.
No CPU fallback: device tensors only.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import torch
from torch import nn

from . import _lib as L
from . import packing as pk
from .program import Buf, Program, Ref
from .unet import _Compiled

# open_clip model configs of the text towers the reference names (open_clip/model_configs/*.json, text_cfg)
OPEN_CLIP_TEXT = {
    "ViT-H-14": dict(width=1024, heads=16, layers=24, vocab_size=49408, context_length=77),
    "ViT-L-14": dict(width=768, heads=12, layers=12, vocab_size=49408, context_length=77),
}


# ------------------------------------------------------------------------------------------
# parameter holder with open_clip's names
# ------------------------------------------------------------------------------------------
class _ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, 4 * width)), ("gelu", nn.GELU()),
                                              ("c_proj", nn.Linear(4 * width, width))]))


class _Transformer(nn.Module):
    def __init__(self, width, heads, layers):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResidualAttentionBlock(width, heads) for _ in range(layers)])


class OpenClipTextModel(nn.Module):
    """Parameters of an open_clip `CLIP` model without its vision tower (what clip_hardcode.py:75-77 keeps after
    `del model.visual`), under the same state-dict names."""

    def __init__(self, width=1024, heads=16, layers=24, vocab_size=49408, context_length=77, embed_dim=None):
        super().__init__()
        self.width, self.heads, self.context_length, self.vocab_size = width, heads, context_length, vocab_size
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width).normal_(std=0.01))
        self.transformer = _Transformer(width, heads, layers)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim or width).normal_(std=width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592)
        mask = torch.full((context_length, context_length), float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)


# ------------------------------------------------------------------------------------------
# state-dict naming of the two towers
# ------------------------------------------------------------------------------------------
class _Names:
    """Where each tensor of block i lives in the holder's state dict."""

    def __init__(self, kind: str, prefix: str = ""):
        self.kind, self.p = kind, prefix

    def tok(self):
        return self.p + ("token_embedding.weight" if self.kind == "open_clip" else "embeddings.token_embedding.weight")

    def pos(self):
        return self.p + ("positional_embedding" if self.kind == "open_clip" else "embeddings.position_embedding.weight")

    def final(self):
        return self.p + ("ln_final" if self.kind == "open_clip" else "final_layer_norm")

    def block(self, i):
        return self.p + (f"transformer.resblocks.{i}" if self.kind == "open_clip" else f"encoder.layers.{i}")

    def ln(self, i, which):
        return f"{self.block(i)}.{'ln_' + str(which) if self.kind == 'open_clip' else 'layer_norm' + str(which)}"

    def qkv(self, i, part):          # part: 'weight' | 'bias' -> recipe over the state dict
        b = self.block(i)
        if self.kind == "open_clip":
            return lambda sd, k=f"{b}.attn.in_proj_{part}": sd[k]
        return lambda sd, b=b, part=part: torch.cat([sd[f"{b}.self_attn.{n}_proj.{part}"] for n in "qkv"], dim=0)

    def out(self, i):
        return f"{self.block(i)}.{'attn.out_proj' if self.kind == 'open_clip' else 'self_attn.out_proj'}"

    def fc(self, i):
        return f"{self.block(i)}.{'mlp.c_fc' if self.kind == 'open_clip' else 'mlp.fc1'}"

    def proj(self, i):
        return f"{self.block(i)}.{'mlp.c_proj' if self.kind == 'open_clip' else 'mlp.fc2'}"


def _detect(sd_keys) -> _Names:
    for k in sd_keys:
        if k.endswith("transformer.resblocks.0.attn.in_proj_weight"):
            return _Names("open_clip", k[: -len("transformer.resblocks.0.attn.in_proj_weight")])
        if k.endswith("encoder.layers.0.self_attn.q_proj.weight"):
            return _Names("hf", k[: -len("encoder.layers.0.self_attn.q_proj.weight")])
    raise L.T2VError("not a CLIP text tower: neither open_clip (transformer.resblocks.*) nor transformers "
                     "(encoder.layers.*) parameter names found")


# ------------------------------------------------------------------------------------------
# lowering
# ------------------------------------------------------------------------------------------
class _TextLowering:
    def __init__(self, names: _Names, *, B, Lseq, width, heads, layers, vocab, act, eps, keep_taps=False):
        assert width % heads == 0 and width // heads == 64, "CLIP text towers use 64-wide heads"
        self.nm, self.B, self.Lseq, self.W, self.H, self.layers = names, B, Lseq, width, heads, layers
        self.vocab, self.act, self.eps = vocab, act, eps
        self.P = Program(f"clip-text b{B} l{Lseq} w{width} x{layers}")
        self.P.keep_taps = keep_taps
        self.packer = pk.WeightPacker()

    def w_lin(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":lin", "f16", lambda sd, k=key: sd[k + ".weight"]))

    def vec(self, key) -> Ref:
        return Ref("weight", 0, self.packer.add(key + ":v", "f32", lambda sd, k=key: sd[k]))

    def build(self) -> Program:
        P, nm, W, H = self.P, self.nm, self.W, self.H
        M = self.B * self.Lseq
        P.begin()
        ids = Ref("ext", L.EXT_X)
        table = Ref("weight", 0, self.packer.add("token_embedding", "f32", lambda sd, k=nm.tok(): sd[k]))
        pos = Ref("weight", 0, self.packer.add("positional_embedding", "f32", lambda sd, k=nm.pos(), n=self.Lseq: sd[k][:n]))
        x = P.alloc(M, W, "f32")
        P.embed_rows("embed", ids, table, "f32", pos, x, L_pos=self.Lseq, vocab=self.vocab)
        ld = 3 * W
        for i in range(self.layers):
            b = nm.block(i)
            n = P.alloc(M, W, "f16")
            P.layernorm(f"{b}.ln_1", x, self.vec(nm.ln(i, 1) + ".weight"), self.vec(nm.ln(i, 1) + ".bias"), n, eps=self.eps)
            qkv = P.alloc(M, 3 * W, "f16")
            wq = Ref("weight", 0, self.packer.add(f"{b}:qkv", "f16", nm.qkv(i, "weight")))
            bq = Ref("weight", 0, self.packer.add(f"{b}:qkv_b", "f32", nm.qkv(i, "bias")))
            P.gemm(f"{b}.qkv", n, wq, 3 * W, W, qkv, bias=bq)
            P.free(n)
            a = P.alloc(M, W, "f16")
            q, k, v = qkv.col_slice(0, W), qkv.col_slice(W, 2 * W), qkv.col_slice(2 * W, 3 * W)
            P.attention(f"{b}.attn", q.ref, k.ref, v.ref, a.ref, out_buf=a, nq=self.Lseq, nk=self.Lseq, heads=H,
                        b_outer=self.B, b_inner=1, q_strides=(ld, self.Lseq * ld, 0), kv_strides=(ld, self.Lseq * ld, 0),
                        o_strides=(W, self.Lseq * W, 0), scale=64 ** -0.5, causal=True)
            P.free(qkv)
            x1 = P.alloc(M, W, "f32")
            P.gemm(f"{b}.out_proj", a, self.w_lin(nm.out(i)), W, W, x1, bias=self.vec(nm.out(i) + ".bias"), residual=x)
            P.free(a, x)
            n = P.alloc(M, W, "f16")
            P.layernorm(f"{b}.ln_2", x1, self.vec(nm.ln(i, 2) + ".weight"), self.vec(nm.ln(i, 2) + ".bias"), n, eps=self.eps)
            f = P.alloc(M, 4 * W, "f16")
            P.gemm(f"{b}.fc", n, self.w_lin(nm.fc(i)), 4 * W, W, f, bias=self.vec(nm.fc(i) + ".bias"))
            P.free(n)
            P.copy2d(f"{b}.act", f, f, act=self.act)
            x = P.alloc(M, W, "f32")
            P.gemm(f"{b}.proj", f, self.w_lin(nm.proj(i)), W, 4 * W, x, bias=self.vec(nm.proj(i) + ".bias"), residual=x1)
            P.free(f, x1)
            P.tap(b, x)
        n = P.alloc(M, W, "f16")
        P.layernorm("ln_final", x, self.vec(nm.final() + ".weight"), self.vec(nm.final() + ".bias"), n, eps=self.eps)
        P.free(x)
        P.copy2d("z", n, Buf(Ref("ext", L.EXT_OUT), M, W, W, "f32"))
        P.free(n)
        P.finish()
        return P


ACT_GELU, ACT_QUICK_GELU = 2, 3


class ClipTextTower:
    """tokens int [B, L] (device) -> z fp32 [B, L, width]: the transformer part of either embedder, run from the
    parameters of `holder` (an open_clip CLIP / `OpenClipTextModel`, or a transformers `CLIPTextModel` /
    `CLIPTextTransformer`).  `skip_last` = number of trailing blocks left out (`layer_idx` of the reference: 1 for
    'penultimate')."""

    def __init__(self, holder: nn.Module, *, heads: Optional[int] = None, act: Optional[str] = None, skip_last: int = 0,
                 eps: float = 1e-5):
        self.holder = holder
        sd = holder.state_dict()
        self.names = _detect(sd.keys())
        nm = self.names
        self.vocab, self.width = sd[nm.tok()].shape
        self.max_len = sd[nm.pos()].shape[0]
        n_layers = 0
        while f"{nm.ln(n_layers, 1)}.weight" in sd:
            n_layers += 1
        self.n_layers, self.skip_last = n_layers, skip_last
        assert 0 <= skip_last < n_layers
        cfg = getattr(holder, "config", None)
        if heads is None:
            heads = getattr(cfg, "num_attention_heads", None) or getattr(holder, "heads", None) or self.width // 64
        if act is None:
            act = getattr(cfg, "hidden_act", None) or ("gelu" if nm.kind == "open_clip" else "quick_gelu")
        if act not in ("gelu", "quick_gelu"):
            raise L.T2VError(f"unsupported CLIP activation {act!r}")
        self.heads, self.act = heads, (ACT_GELU if act == "gelu" else ACT_QUICK_GELU)
        self.eps = float(getattr(cfg, "layer_norm_eps", eps))
        self._programs: Dict[tuple, _Compiled] = {}
        self._packed, self._packed_sig, self._packed_device = None, None, None
        self.debug_taps = False

    def _signature(self):
        return tuple((id(p), p._version, p.device.type, p.dtype) for p in self.holder.parameters())

    def _compile(self, B, Lseq) -> _Compiled:
        low = _TextLowering(self.names, B=B, Lseq=Lseq, width=self.width, heads=self.heads,
                            layers=self.n_layers - self.skip_last, vocab=self.vocab, act=self.act, eps=self.eps,
                            keep_taps=self.debug_taps)
        return _Compiled(low.build(), low.packer)

    def refresh_weights(self, comp: _Compiled, device):
        sig = self._signature()
        if self._packed is not None and sig == self._packed_sig and device == self._packed_device:
            return
        self._packed = comp.packer.materialise(self.holder.state_dict(), device)
        self._packed_sig, self._packed_device = sig, device
        for c in self._programs.values():
            c.bound = None

    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        if not tokens.is_cuda:
            raise L.T2VError("ClipTextTower needs device tensors on an AMD GPU (no CPU fallback)")
        assert tokens.ndim == 2 and tokens.shape[1] <= self.max_len
        B, Lseq = tokens.shape
        ids = tokens.to(torch.int32).contiguous()
        comp = self._programs.get((B, Lseq))
        if comp is None:
            comp = self._programs[(B, Lseq)] = self._compile(B, Lseq)
        self.refresh_weights(comp, tokens.device)
        comp.ensure_bound(self._packed, tokens.device)
        z = torch.empty((B, Lseq, self.width), device=tokens.device, dtype=torch.float32)
        comp.bound.run({L.EXT_X: ids.data_ptr(), L.EXT_OUT: z.data_ptr()}, torch.cuda.current_stream(tokens.device).cuda_stream)
        return z


# ------------------------------------------------------------------------------------------
# the reference's two embedder classes
# ------------------------------------------------------------------------------------------
class FrozenOpenCLIPEmbedder(nn.Module):
    """Mirror of clip_hardcode.py:59-422 for everything from token ids onwards.  `version` = path of an
    `open_clip_pytorch_model.bin` (text keys are loaded, `visual.*` ignored); `tokenizer` = an object with
    `.encode(text) -> List[int]` and an `.encoder` vocabulary (open_clip's `_tokenizer`), needed only by `forward(texts)`.
    `device` is where the tower runs (the reference keeps its encoder on the CPU; this one has no CPU path)."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version=None, device="cuda", max_length=77, freeze=True, layer="last",
                 tokenizer=None, model: Optional[nn.Module] = None):
        super().__init__()
        assert layer in self.LAYERS
        if model is None:
            model = OpenClipTextModel(**OPEN_CLIP_TEXT[arch])
            if version is not None and os.path.exists(version):
                sd = torch.load(version, map_location="cpu")
                sd = {k: v for k, v in sd.items() if not k.startswith("visual.")}
                model.load_state_dict(sd, strict=True)
        self.model = model
        self.device = device
        self.max_length = max_length
        if freeze:
            self.freeze()
        self.layer = layer
        self.layer_idx = {"last": 0, "penultimate": 1}[layer]
        self.tokenizer = tokenizer
        enc = getattr(tokenizer, "encoder", None) or {}
        self.comma_token = enc.get(",</w>")
        self.id_start = enc.get("<start_of_text>", 49406)
        self.id_end = enc.get("<end_of_text>", 49407)
        self.id_pad = 0
        self.chunk_length = 75
        self._tower: Optional[ClipTextTower] = None

    def freeze(self):
        self.model = self.model.eval()
        for param in self.parameters():
            param.requires_grad = False

    @property
    def tower(self) -> ClipTextTower:
        if self._tower is None or self._tower.skip_last != self.layer_idx:
            heads = getattr(self.model, "heads", None) or self.model.transformer.resblocks[0].attn.num_heads
            self._tower = ClipTextTower(self.model, heads=heads, act="gelu", skip_last=self.layer_idx)
        return self._tower

    # :110-117 / :269-274
    def encode_with_transformer(self, text: torch.Tensor) -> torch.Tensor:
        return self.tower(text.to(self.device))

    def encode_with_transformers(self, tokens):       # :119-123
        return self.encode_with_transformer(tokens)

    def tokenize(self, texts):                        # :101-108
        if self.tokenizer is None:
            raise L.T2VError("FrozenOpenCLIPEmbedder.forward(texts) needs a tokenizer (open_clip's `_tokenizer`); "
                             "pass token ids to process_tokens / encode_with_transformers instead")
        return [self.tokenizer.encode(t) for t in texts]

    def empty_chunk(self):                            # :132-138
        return [self.id_start] + [self.id_end] * (self.chunk_length + 1), [1.0] * (self.chunk_length + 2)

    def tokenize_line(self, line):
        """:147-242 with emphasis off, no textual-inversion embeddings and comma_padding_backtrack = 0 (those are
        webui options served by webui modules): 75-token chunks, start / end framed, end-padded."""
        tokens = self.tokenize([line])[0]
        chunks, count = [], 0
        for c0 in range(0, max(len(tokens), 1), self.chunk_length):
            body = tokens[c0:c0 + self.chunk_length]
            last = c0 + self.chunk_length >= len(tokens)
            count += len(body) if last else self.chunk_length
            body = body + [self.id_end] * (self.chunk_length - len(body))
            chunks.append(([self.id_start] + body + [self.id_end], [1.0] * (self.chunk_length + 2)))
        return chunks, count

    def process_tokens(self, remade_batch_tokens, batch_multipliers):
        """:392-421 — one 77-token chunk per row; tokens after the first <end> become pad (SD2 convention), the
        multipliers scale z and the original mean is restored."""
        dev = torch.device(self.device)
        tokens = torch.as_tensor(remade_batch_tokens).to(dev)
        if self.id_end != self.id_pad:
            for row, toks in enumerate(remade_batch_tokens):
                index = list(toks).index(self.id_end)
                tokens[row, index + 1:tokens.shape[1]] = self.id_pad
        z = self.encode_with_transformers(tokens)
        mult = torch.as_tensor(batch_multipliers, dtype=z.dtype).to(dev)
        original_mean = z.mean()
        z = z * mult.reshape(mult.shape + (1,)).expand(z.shape)
        new_mean = z.mean()
        return z * (original_mean / new_mean)

    def forward(self, texts: Sequence[str]) -> torch.Tensor:          # :364-390
        batch_chunks = [self.tokenize_line(t)[0] for t in texts]
        chunk_count = max(len(c) for c in batch_chunks)
        zs = []
        for i in range(chunk_count):
            chunk = [c[i] if i < len(c) else self.empty_chunk() for c in batch_chunks]
            zs.append(self.process_tokens([t for t, _ in chunk], [m for _, m in chunk]))
        return torch.hstack(zs)

    def encode(self, text):
        return self(text)

    def get_learned_conditioning(self, text):
        return self.encode(text)


class FrozenCLIPEmbedder(nn.Module):
    """Mirror of lvdm/models/modules/condition_modules.py:15-42.  `transformer` = a transformers `CLIPTextModel`
    (the parameter holder; `CLIPTextModel.from_pretrained(version)` when not given), `tokenizer` likewise."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, transformer=None, tokenizer=None):
        super().__init__()
        if transformer is None or tokenizer is None:
            from transformers import CLIPTextModel, CLIPTokenizer
            tokenizer = tokenizer or CLIPTokenizer.from_pretrained(version)
            transformer = transformer or CLIPTextModel.from_pretrained(version)
        self.tokenizer, self.transformer = tokenizer, transformer
        self.device, self.max_length = device, max_length
        self.freeze()
        self._tower: Optional[ClipTextTower] = None

    def freeze(self):
        self.transformer = self.transformer.eval()
        for param in self.parameters():
            param.requires_grad = False

    def encode_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        if self._tower is None:
            self._tower = ClipTextTower(self.transformer)
        return self._tower(tokens.to(self.device))

    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return self.encode_tokens(enc["input_ids"])

    def encode(self, text):
        return self(text)
