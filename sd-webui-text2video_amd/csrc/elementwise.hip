// Layout, pointwise and sampler-update kernels (HBM-bound, vectorised), gfx950.
//
//   ncthw_to_cl   'b c f h w -> (b f h w) c' entry conversion of the 4-channel latent
//                 (reference does this as rearrange + .contiguous(), t2v_model.py:429)
//   cl_to_ncthw   exit conversion of eps / decoded RGB            (t2v_model.py:456-458)
//   time_embed    sinusoidal_embedding (cos | sin), t2v_model.py:504-515
//   copy2d        strided 2-D copy / fp32->fp16 cast / SiLU: torch.cat of skip connections
//                 (t2v_model.py:444), operand staging, SiLU(e) of emb_layers (:936)
//   ddim_step     DDIM_Gaussian update incl. half-channel classifier-free guidance
//                 (samplers/ddim/gaussian_sampler.py:125-136, 103-108, 199-211, 269-283)
#include "t2v_kernels.h"

namespace {

template <typename TIN>
__global__ __launch_bounds__(256) void ncthw_to_cl_kernel(const TIN* in, f16* out, int B, int C, int F, int HW,
                                                          int ld, float scale, int Bsrc, f16* out_lo, int lo_in_pad) {
  // one thread per output token; writes ld (>= C, multiple of 4) channels, zero padded.  Bsrc < B: the source holds
  // Bsrc samples and output sample b reads source sample b % Bsrc (the cond | uncond pair of a guided step shares x_t,
  // gaussian_sampler.py:161-162 — no torch.cat([x, x]) on the host)
  const long total = (long)B * F * HW;
  for (long tkn = (long)blockIdx.x * 256 + threadIdx.x; tkn < total; tkn += (long)gridDim.x * 256) {
    const long bf = tkn / HW;
    const int pix = (int)(tkn - bf * HW);
    const int b = (int)(bf / F), f = (int)(bf - (long)b * F);
    const int bs = b % Bsrc;
    f16* o = out + tkn * ld;
    for (int c = 0; c < ld; ++c) {
      float v = 0.f;
      if (c < C) v = (float)in[(((size_t)bs * C + c) * F + f) * HW + pix] * scale;
      const f16 hi = (f16)v;
      o[c] = hi;
      if (out_lo) out_lo[tkn * ld + c] = (f16)(v - (float)hi);      // low-order image: hi + lo carries the fp32 value (p[2])
    }
    if (lo_in_pad) {     // i[7]: the low-order images go into the padding channels C .. 2C-1 of the SAME row (ld >= 2C): a consumer
                         // whose weights repeat W for those channels computes (hi + lo) . W in one pass
      for (int c = 0; c < C; ++c) {
        const float v = (float)in[(((size_t)bs * C + c) * F + f) * HW + pix] * scale;
        o[C + c] = (f16)(v - (float)(f16)v);
      }
    }
  }
}

template <typename TOUT>
__global__ __launch_bounds__(256) void cl_to_ncthw_kernel(const float* in, TOUT* out, int B, int C, int F, int HW,
                                                          int ld) {
  // one thread per output element (coalesced along pixels)
  const long total = (long)B * C * F * HW;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int pix = (int)(idx % HW);
    long r = idx / HW;
    const int f = (int)(r % F); r /= F;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    out[idx] = (TOUT)in[(((size_t)b * F + f) * HW + pix) * ld + c];
  }
}

__global__ __launch_bounds__(256) void time_embed_kernel(const float* t, const float* freqs, f16* out, int B,
                                                         int dim) {
  const int half = dim / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx - b * half;
  const float a = t[b] * freqs[i];
  out[(size_t)b * dim + i] = (f16)cosf(a);
  out[(size_t)b * dim + half + i] = (f16)sinf(a);
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy2d_kernel(const TS* src, TD* dst, int rows, int cols, int lds_,
                                                     int ldd, int act, f16* dst_lo) {
  const int cv = cols >> 2;
  const long total = (long)rows * cv;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
    const long r = u / cv;
    const int c = (int)(u - r * cv) * 4;
    const TS* s = src + r * lds_ + c;
    TD* d = dst + r * ldd + c;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)s[e];
    if (act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t2v_silu(v[e]);
    } else if (act == 2) {            // nn.GELU (erf): OpenCLIP ViT-H text MLP
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t2v_gelu_erf(v[e]);
    } else if (act == 3) {            // quick GELU x * sigmoid(1.702 x): OpenAI CLIP-L text MLP
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = (TD)v[e];
    if (dst_lo) {                     // low-order fp16 image of an fp32 -> fp16 cast (same leading dimension): hi + lo = v to ~2^-22
#pragma unroll
      for (int e = 0; e < 4; ++e) dst_lo[r * ldd + c + e] = (f16)(v[e] - (float)(f16)v[e]);
    }
  }
}

// out[r, :] = table[ids[r], :] + pos[r % L, :]   (token + positional embedding of the CLIP text towers);
// ids outside [0, vocab) write zeros + pos so that a bad token cannot read out of bounds
template <typename TT>
__global__ __launch_bounds__(256) void embed_rows_kernel(const int* ids, const TT* table, const float* pos, float* out,
                                                         int rows, int W, int L, int vocab) {
  const int r = blockIdx.x;
  const int id = ids[r];
  const bool ok = id >= 0 && id < vocab;
  const TT* t = table + (size_t)(ok ? id : 0) * W;
  const float* pr = pos + (size_t)(r % L) * W;
  for (int c = threadIdx.x; c < W; c += 256) out[(size_t)r * W + c] = (ok ? (float)t[c] : 0.f) + pr[c];
}

struct DdimParams {
  const void* xt; const void* eps; const float* noise; void* out;
  int C, inner, guided, eps_f32, x_f32, mode, cps;
  float a_recip, a_recipm1, sqrt_aprev, dir_coef, sigma, gscale;
};

template <typename TX, typename TE>
__global__ __launch_bounds__(256) void ddim_step_kernel(const DdimParams p) {
  // x_t [S,Cs,inner] with C = S*Cs rows; eps [2,S,Cs,inner] (0 = conditional, 1 = unconditional); S videos per batch
  const TX* xt = reinterpret_cast<const TX*>(p.xt);
  const TE* ec = reinterpret_cast<const TE*>(p.eps);
  const TE* eu = ec + (size_t)p.C * p.inner;
  TX* out = reinterpret_cast<TX*>(p.out);
  const long total = (long)p.C * p.inner;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx / p.inner);
    const float x = (float)xt[idx];
    const float y = (float)ec[idx];
    float o = y;
    if (c % p.cps < p.guided) {
      const float u = (float)eu[idx];
      o = u + p.gscale * (y - u);
    }
    // same operation order as the reference (all fp32)
    float xn;
    if (p.mode == 0) {                       // DDIM_Gaussian
      const float x0 = p.a_recip * x - p.a_recipm1 * o;
      const float eps = (p.a_recip * x - x0) / p.a_recipm1;
      xn = p.sqrt_aprev * x0 + p.dir_coef * eps;
    } else {                                 // LDM DDIM: a_recip = sqrt(1-a_t), a_recipm1 = sqrt(a_t)
      const float x0 = (x - p.a_recip * o) / p.a_recipm1;
      xn = p.sqrt_aprev * x0 + p.dir_coef * o;
    }
    if (p.noise != nullptr && p.sigma != 0.f) xn += p.sigma * p.noise[idx];
    out[idx] = (TX)xn;
  }
}

struct LinParams {
  const void* t[6];
  void* out;
  float c[6];
  int f32[6];
  int n_terms, out_f32;
  long n;
};

__global__ __launch_bounds__(256) void lincomb_kernel(const LinParams p) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < p.n; idx += (long)gridDim.x * 256) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k < p.n_terms) {
        const float v = p.f32[k] ? reinterpret_cast<const float*>(p.t[k])[idx] : (float)reinterpret_cast<const f16*>(p.t[k])[idx];
        acc = k == 0 ? p.c[0] * v : acc + p.c[k] * v;
      }
    }
    if (p.out_f32) reinterpret_cast<float*>(p.out)[idx] = acc;
    else reinterpret_cast<f16*>(p.out)[idx] = (f16)acc;
  }
}

// Row regrouping between the two layouts of a T-sharded clip (frame-sharded [frames][hw] rows <-> pixel-sharded
// [frames][hw / R] rows): row r of the op reads source row (r / P) * S_src + r % P and writes destination row
// (r / P) * S_dst + r % P (chunks of P rows, chunk strides in rows); optional fp32 residual, indexed like the
// destination, added on the way (the TemporalTransformer's `+ x` after its output is resharded back to frames).
// Round 6: blockIdx.y = part q of `nparts` regroupings of the same shape in ONE launch (the R packs in front of a frames -> pixels
// all-to-all, the R unpacks behind the way back): part q reads src + q * ps_src and writes dst + q * ps_dst (residual + q * ps_res),
// except the rank's own part `own`, whose source (own_is_src) or destination lives elsewhere (`alt`: it does not travel).
template <typename T>
__global__ __launch_bounds__(256) void reshard_rows_kernel(const T* src, T* dst, const float* res, int rows, int cols, int P,
                                                           long s_src, long s_dst, int ld_src, int ld_dst, int ld_res,
                                                           long ps_src, long ps_dst, long ps_res, int own, int own_is_src, T* alt) {
  constexpr int V = 16 / sizeof(T);               // elements per 16-byte unit
  const int q = blockIdx.y;
  src = (q == own && own_is_src) ? alt : src + q * ps_src;
  dst = (q == own && !own_is_src) ? alt : dst + q * ps_dst;
  if (res) res += q * ps_res;
  const int cv = cols / V;
  const long total = (long)rows * cv;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
    const long r = u / cv;
    const int c = (int)(u - r * cv) * V;
    const long chunk = r / P, within = r - chunk * P;
    const long rs = chunk * s_src + within, rd = chunk * s_dst + within;
    if constexpr (sizeof(T) == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + rs * ld_src + c);
      if (res) v += *reinterpret_cast<const f32x4*>(res + rd * ld_res + c);
      *reinterpret_cast<f32x4*>(dst + rd * ld_dst + c) = v;
    } else {
      *reinterpret_cast<f16x8*>(dst + rd * ld_dst + c) = *reinterpret_cast<const f16x8*>(src + rs * ld_src + c);
    }
  }
}

// tensor2vid (t2v_pipeline.py:447-460): video[i,c,f,y,x] -> uint8 out[f, y, i*W + x, c]: v*0.5 + 0.5 (two roundings, as
// mul_ / add_), clamp to [0,1], *255, TRUNCATED like `(image.numpy()*255).astype('uint8')`.  HALF: the reference's
// 'GPU (half precision)' VAE hands tensor2vid an fp16 video, so every intermediate is rounded to fp16 (the input too
// when the tokens are fp32).  Input addressed by element strides: channels-last decoder tokens or a plain NCFHW tensor.
template <typename TIN, bool HALF>
__global__ __launch_bounds__(256) void to_uint8_kernel(const TIN* in, unsigned char* out, int NI, int C, int F, int H, int W,
                                                       long si, long sc, long sf, long sy, long sx, int bgr) {
  const long total = (long)F * H * NI * W;
  for (long px = (long)blockIdx.x * 256 + threadIdx.x; px < total; px += (long)gridDim.x * 256) {
    const int xw = (int)(px % ((long)NI * W));
    long r = px / ((long)NI * W);
    const int y = (int)(r % H);
    const int f = (int)(r / H);
    const int i = xw / W, x = xw - i * W;
    const TIN* src = in + i * si + f * sf + y * sy + x * sx;
    unsigned char* o = out + px * C;
    for (int c = 0; c < C; ++c) {
      float v = (float)src[c * sc];
      if (HALF) {
        f16 h = (f16)v;
        h = (f16)__fmul_rn((float)h, 0.5f);
        h = (f16)__fadd_rn((float)h, 0.5f);
        h = h < (f16)0.f ? (f16)0.f : (h > (f16)1.f ? (f16)1.f : h);
        v = (float)(f16)__fmul_rn((float)h, 255.f);
      } else {
        v = __fadd_rn(__fmul_rn(v, 0.5f), 0.5f);
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        v = __fmul_rn(v, 255.f);
      }
      o[bgr ? C - 1 - c : c] = (unsigned char)(int)v;
    }
  }
}

inline int grid_for(long n) {
  const long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

hipError_t t2v_launch_ncthw_to_cl(const t2v_op& op, hipStream_t s) {
  const int B = op.i[0], C = op.i[1], F = op.i[2], HW = op.i[3], ld = op.i[4];
  const long n = (long)B * F * HW;
  const int Bsrc = (op.i[6] > 0 && op.i[6] < B) ? op.i[6] : B;
  f16* out = reinterpret_cast<f16*>(op.p[1]);
  if (op.i[5] == T2V_F32)
    hipLaunchKernelGGL(ncthw_to_cl_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(op.p[0]), out, B, C, F, HW, ld, op.f[0], Bsrc, reinterpret_cast<f16*>(op.p[2]),
                       (op.i[7] != 0 && ld >= 2 * C) ? 1 : 0);
  else
    hipLaunchKernelGGL(ncthw_to_cl_kernel<f16>, dim3(grid_for(n)), dim3(256), 0, s,
                       reinterpret_cast<const f16*>(op.p[0]), out, B, C, F, HW, ld, op.f[0], Bsrc, reinterpret_cast<f16*>(op.p[2]),
                       (op.i[7] != 0 && ld >= 2 * C) ? 1 : 0);
  return hipGetLastError();
}

hipError_t t2v_launch_cl_to_ncthw(const t2v_op& op, hipStream_t s) {
  const int B = op.i[0], C = op.i[1], F = op.i[2], HW = op.i[3], ld = op.i[4];
  const long n = (long)B * C * F * HW;
  const float* in = reinterpret_cast<const float*>(op.p[0]);
  if (op.i[5] == T2V_F32)
    hipLaunchKernelGGL(cl_to_ncthw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, in,
                       reinterpret_cast<float*>(op.p[1]), B, C, F, HW, ld);
  else
    hipLaunchKernelGGL(cl_to_ncthw_kernel<f16>, dim3(grid_for(n)), dim3(256), 0, s, in,
                       reinterpret_cast<f16*>(op.p[1]), B, C, F, HW, ld);
  return hipGetLastError();
}

hipError_t t2v_launch_to_uint8(const t2v_op& op, hipStream_t s) {
  const int NI = op.i[0], C = op.i[1], F = op.i[2], H = op.i[3], W = op.i[4];
  const bool half = op.i[6] != 0;
  const int bgr = op.i[7];
  const long si = (long)(uint32_t)op.i[8] | ((long)op.i[9] << 32), sc = op.i[10], sf = (long)(uint32_t)op.i[11] | ((long)op.i[12] << 32);
  const long sy = op.i[13], sx = op.i[14];
  const int g = grid_for((long)F * H * NI * W);
  unsigned char* out = reinterpret_cast<unsigned char*>(op.p[1]);
  if (op.i[5] == T2V_F32) {
    const float* in = reinterpret_cast<const float*>(op.p[0]);
    if (half) hipLaunchKernelGGL((to_uint8_kernel<float, true>), dim3(g), dim3(256), 0, s, in, out, NI, C, F, H, W, si, sc, sf, sy, sx, bgr);
    else hipLaunchKernelGGL((to_uint8_kernel<float, false>), dim3(g), dim3(256), 0, s, in, out, NI, C, F, H, W, si, sc, sf, sy, sx, bgr);
  } else {
    const f16* in = reinterpret_cast<const f16*>(op.p[0]);
    if (half) hipLaunchKernelGGL((to_uint8_kernel<f16, true>), dim3(g), dim3(256), 0, s, in, out, NI, C, F, H, W, si, sc, sf, sy, sx, bgr);
    else hipLaunchKernelGGL((to_uint8_kernel<f16, false>), dim3(g), dim3(256), 0, s, in, out, NI, C, F, H, W, si, sc, sf, sy, sx, bgr);
  }
  return hipGetLastError();
}

hipError_t t2v_launch_reshard_rows(const t2v_op& op, hipStream_t s) {
  const int rows = op.i[0], cols = op.i[1], P = op.i[2], ld_src = op.i[5], ld_dst = op.i[6], ld_res = op.i[8];
  const long s_src = op.i[3], s_dst = op.i[4];
  const bool f32 = op.i[7] == T2V_F32;
  if (rows <= 0 || cols <= 0 || P <= 0 || cols % (f32 ? 4 : 8) != 0 || ld_src % (f32 ? 4 : 8) != 0 || ld_dst % (f32 ? 4 : 8) != 0)
    return hipErrorInvalidValue;
  if (op.p[2] != 0 && (!f32 || ld_res % 4 != 0)) return hipErrorInvalidValue;
  const int nparts = op.i[9] > 1 ? op.i[9] : 1;
  const int own = nparts > 1 ? op.i[13] : -1, own_is_src = op.i[14];
  const int v = f32 ? 4 : 8;
  if (nparts > 1 && (op.i[10] % v != 0 || op.i[11] % v != 0 || op.i[12] % 4 != 0 || own >= nparts || (own >= 0 && op.p[3] == 0) || nparts > 64))
    return hipErrorInvalidValue;
  const int g = grid_for((long)rows * (cols / v));
  if (f32)
    hipLaunchKernelGGL(reshard_rows_kernel<float>, dim3(g, nparts), dim3(256), 0, s, reinterpret_cast<const float*>(op.p[0]),
                       reinterpret_cast<float*>(op.p[1]), reinterpret_cast<const float*>(op.p[2]), rows, cols, P, s_src, s_dst,
                       ld_src, ld_dst, ld_res, (long)op.i[10], (long)op.i[11], (long)op.i[12], own, own_is_src, reinterpret_cast<float*>(op.p[3]));
  else
    hipLaunchKernelGGL(reshard_rows_kernel<f16>, dim3(g, nparts), dim3(256), 0, s, reinterpret_cast<const f16*>(op.p[0]),
                       reinterpret_cast<f16*>(op.p[1]), static_cast<const float*>(nullptr), rows, cols, P, s_src, s_dst, ld_src,
                       ld_dst, 0, (long)op.i[10], (long)op.i[11], 0L, own, own_is_src, reinterpret_cast<f16*>(op.p[3]));
  return hipGetLastError();
}

hipError_t t2v_launch_time_embed(const t2v_op& op, hipStream_t s) {
  const int B = op.i[0], dim = op.i[1];
  hipLaunchKernelGGL(time_embed_kernel, dim3((B * (dim / 2) + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<const float*>(op.p[0]), reinterpret_cast<const float*>(op.p[1]),
                     reinterpret_cast<f16*>(op.p[2]), B, dim);
  return hipGetLastError();
}

hipError_t t2v_launch_copy2d(const t2v_op& op, hipStream_t s) {
  const int rows = op.i[0], cols = op.i[1], lds_ = op.i[2], ldd = op.i[3], sdt = op.i[4], ddt = op.i[5];
  const int act = op.i[6];
  if (cols % 4 != 0) return hipErrorInvalidValue;
  const int g = grid_for((long)rows * (cols / 4));
  if (sdt == T2V_F32 && ddt == T2V_F32)
    hipLaunchKernelGGL((copy2d_kernel<float, float>), dim3(g), dim3(256), 0, s, reinterpret_cast<const float*>(op.p[0]),
                       reinterpret_cast<float*>(op.p[1]), rows, cols, lds_, ldd, act, static_cast<f16*>(nullptr));
  else if (sdt == T2V_F32 && ddt == T2V_F16)
    hipLaunchKernelGGL((copy2d_kernel<float, f16>), dim3(g), dim3(256), 0, s, reinterpret_cast<const float*>(op.p[0]),
                       reinterpret_cast<f16*>(op.p[1]), rows, cols, lds_, ldd, act, reinterpret_cast<f16*>(op.p[2]));
  else if (sdt == T2V_F16 && ddt == T2V_F16)
    hipLaunchKernelGGL((copy2d_kernel<f16, f16>), dim3(g), dim3(256), 0, s, reinterpret_cast<const f16*>(op.p[0]),
                       reinterpret_cast<f16*>(op.p[1]), rows, cols, lds_, ldd, act, static_cast<f16*>(nullptr));
  else
    hipLaunchKernelGGL((copy2d_kernel<f16, float>), dim3(g), dim3(256), 0, s, reinterpret_cast<const f16*>(op.p[0]),
                       reinterpret_cast<float*>(op.p[1]), rows, cols, lds_, ldd, act, static_cast<f16*>(nullptr));
  return hipGetLastError();
}

hipError_t t2v_launch_embed_rows(const t2v_op& op, hipStream_t s) {
  const int rows = op.i[0], W = op.i[1], L = op.i[2], vocab = op.i[3];
  if (rows <= 0 || W <= 0 || L <= 0 || vocab <= 0) return hipErrorInvalidValue;
  const int* ids = reinterpret_cast<const int*>(op.p[0]);
  const float* pos = reinterpret_cast<const float*>(op.p[2]);
  float* out = reinterpret_cast<float*>(op.p[3]);
  if (op.i[4] == T2V_F32)
    hipLaunchKernelGGL((embed_rows_kernel<float>), dim3(rows), dim3(256), 0, s, ids, reinterpret_cast<const float*>(op.p[1]), pos,
                       out, rows, W, L, vocab);
  else
    hipLaunchKernelGGL((embed_rows_kernel<f16>), dim3(rows), dim3(256), 0, s, ids, reinterpret_cast<const f16*>(op.p[1]), pos,
                       out, rows, W, L, vocab);
  return hipGetLastError();
}

hipError_t t2v_launch_ddim_step(const t2v_op& op, hipStream_t s) {
  DdimParams p;
  p.xt = reinterpret_cast<const void*>(op.p[0]);
  p.eps = reinterpret_cast<const void*>(op.p[1]);
  p.noise = reinterpret_cast<const float*>(op.p[2]);
  p.out = reinterpret_cast<void*>(op.p[3]);
  p.C = op.i[0]; p.inner = op.i[1]; p.guided = op.i[2]; p.eps_f32 = op.i[3] == T2V_F32; p.x_f32 = op.i[4] == T2V_F32;
  p.mode = op.i[5];
  p.cps = op.i[6] > 0 ? op.i[6] : p.C;       // channels per sample (several videos per batch: C = samples * cps)
  if (p.C % p.cps != 0) return hipErrorInvalidValue;
  p.a_recip = op.f[0]; p.a_recipm1 = op.f[1]; p.sqrt_aprev = op.f[2]; p.dir_coef = op.f[3]; p.sigma = op.f[4];
  p.gscale = op.f[5];
  const int g = grid_for((long)p.C * p.inner);
  if (p.x_f32 && p.eps_f32) hipLaunchKernelGGL((ddim_step_kernel<float, float>), dim3(g), dim3(256), 0, s, p);
  else if (p.x_f32) hipLaunchKernelGGL((ddim_step_kernel<float, f16>), dim3(g), dim3(256), 0, s, p);
  else if (p.eps_f32) hipLaunchKernelGGL((ddim_step_kernel<f16, float>), dim3(g), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((ddim_step_kernel<f16, f16>), dim3(g), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t t2v_launch_lincomb(const t2v_op& op, hipStream_t s) {
  LinParams p;
  p.n = op.i[0]; p.n_terms = op.i[1]; p.out_f32 = op.i[2] == T2V_F32;
  if (p.n_terms < 1 || p.n_terms > 6 || p.n <= 0) return hipErrorInvalidValue;
  for (int k = 0; k < 6; ++k) {
    p.t[k] = reinterpret_cast<const void*>(op.p[k]);
    p.c[k] = op.f[k];
    p.f32[k] = op.i[3 + k] == T2V_F32;
  }
  p.out = reinterpret_cast<void*>(op.p[6]);
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(p.n)), dim3(256), 0, s, p);
  return hipGetLastError();
}
