"""Spatial self-attention shapes of the UNet (head_dim 64) on the attention kernel alone: TF/s, median of 12 launches, and a check against
torch SDPA on a sample of (batch, head) problems.  ATTN_SHAPES="nq,heads,batch;..." overrides the list."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_webui_text2video_amd.program import BoundProgram, Program  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(1024, 5, 48), (9216, 5, 48), (256, 10, 48), (9216, 5, 8)]
if os.environ.get("ATTN_SHAPES"):
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["ATTN_SHAPES"].split(";")]
for (n, heads, B) in shapes:
    inner, D = heads * 64, 64
    M = B * n
    P = Program()
    qkv, o = P.alloc(M, 3 * inner, "f16"), P.alloc(M, inner, "f16")
    ld = 3 * inner
    q, k, v = qkv.col_slice(0, inner), qkv.col_slice(inner, 2 * inner), qkv.col_slice(2 * inner, 3 * inner)
    waves = int(os.environ.get("ATTN_VT", "0"))          # 0: attn_kernel; 4 / 8: attn2_kernel (V^T scratch, LDS-DMA tiles) with that many waves
    vt = P.alloc(B * heads * 64, -(-n // 64) * 64, "f16") if waves else None
    P.attention("a", q.ref, k.ref, v.ref, o.ref, nq=n, nk=n, heads=heads, b_outer=B, b_inner=1, q_strides=(ld, n * ld, 0), kv_strides=(ld, n * ld, 0),
                o_strides=(inner, n * inner, 0), scale=D ** -0.5, head_dim=D, vt_scratch=vt, waves=waves)
    P.ops = P.ops * 12
    arena = torch.empty(P.arena.high + 256, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    arena.view(torch.float16).normal_(0, 1, generator=g)
    bp = BoundProgram(P, arena.data_ptr(), {})
    st = torch.cuda.current_stream(dev).cuda_stream
    bp.run({}, st)
    torch.cuda.synchronize()
    ms = sorted(bp.run_timed({}, st))[6]
    x = arena[qkv.ref.off:qkv.ref.off + M * ld * 2].view(torch.float16).view(B, n, 3, heads, D)
    got = arena[o.ref.off:o.ref.off + M * inner * 2].view(torch.float16).view(B, n, heads, D)
    bs = [0, B - 1]
    ref = torch.nn.functional.scaled_dot_product_attention(x[bs, :, 0].permute(0, 2, 1, 3).float(), x[bs, :, 1].permute(0, 2, 1, 3).float(),
                                                           x[bs, :, 2].permute(0, 2, 1, 3).float()).permute(0, 2, 1, 3)
    e = float((got[bs].float() - ref).norm() / ref.norm())
    print(f"attention n {n:5d} heads {heads:2d} batch {B:3d}: {ms * 1e3:8.1f} us  {4.0 * B * heads * n * n * D / ms / 1e9:6.0f} TF/s  (rel-L2 vs SDPA {e:.1e})", flush=True)
