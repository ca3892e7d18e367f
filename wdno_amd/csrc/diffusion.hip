// diffusion.hip -- fused elementwise kernels of the two GaussianDiffusion operators (HBM-bound, 3-5 streams).
// Tensors are in the reference's API layout: smoke [B,F,C,H,W], Burgers [B,C,H,W] (F = 1).
#include "common.h"
#include <algorithm>

// 0 = free (diffused), 1 = forced zero, 2 = clean value.
// smoke  : smoke/ddpm/diffusion_2d.py:1008-1033 -- order init, control, pad, low (later statements win)
// Burgers: burgers/ddpm_burgers/diffusion_1d.py:276-288, call order pad, u0, uT, f, low
__device__ __forceinline__ int cond_code(const wdno_cond_desc& d, int f, int c, int h, int w) {
  if (d.tree == 0) {
    if (d.cond_low && c >= 40 && c < 80) return 2;
    if (d.cond_pad) {
      if ((f >= d.cT && c != d.C - 2) || (c != d.C - 1 && (h >= d.cH || w >= d.cW))) return 1;
    }
    if (c == d.C - 2) return 2;
    if (d.cond_a && c >= 24 && c < 40) return 2;
    return 0;
  } else {
    bool inside = (h < d.cH) && (w < d.cW);
    if (d.cond_low && c >= 8 && c < 16 && inside) return 2;
    if (d.cond_c && c >= 4 && c < 8 && inside) return 2;
    if (d.cond_b && c == d.C - 1 && h >= d.H - d.uT_rows && w < d.cW) return 2;
    if (d.cond_a && c == d.C - 1 && h < d.u_rows && w < d.cW) return 2;
    if (d.cond_pad && ((c != d.C - 1 && h >= d.cH) || w >= d.cW)) return 1;
    return 0;
  }
}

__device__ __forceinline__ void decode(const wdno_cond_desc& d, int64_t i, int& b, int& f, int& c, int& h, int& w) {
  w = (int)(i % d.W); i /= d.W;
  h = (int)(i % d.H); i /= d.H;
  c = (int)(i % d.C); i /= d.C;
  f = (int)(i % d.F);
  b = (int)(i / d.F);
}

__global__ __launch_bounds__(256) void q_sample_cond_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                             const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                             const float* __restrict__ sb, float* __restrict__ xo,
                                                             float* __restrict__ to, wdno_cond_desc d, int64_t total) {
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int b, f, c, h, w;
    decode(d, i, b, f, c, h, w);
    int code = cond_code(d, f, c, h, w);
    float v0 = x0[i], nz = noise[i];
    int64_t tb = t[b];
    float xv = sa[tb] * v0 + sb[tb] * nz;
    xo[i] = code == 0 ? xv : (code == 1 ? 0.0f : v0);
    to[i] = code == 0 ? nz : 0.0f;
  }
}
// float4 forms (W % 4 == 0, total < 2^31): one index decode in 32-bit arithmetic per four elements -- the scalar kernels spend ~5 64-bit
// divisions per element and ran at 1/5 of the memory rate (62 us for the 13 MB smoke state). Per-element arithmetic is unchanged.
__device__ __forceinline__ void decode4(const wdno_cond_desc& d, unsigned i4, int& b, int& f, int& c, int& h, int& w) {
  const unsigned W4 = (unsigned)d.W >> 2;
  unsigned q = i4 / W4;
  w = (int)(i4 - q * W4) * 4;
  unsigned q2 = q / (unsigned)d.H; h = (int)(q - q2 * (unsigned)d.H); q = q2;
  q2 = q / (unsigned)d.C; c = (int)(q - q2 * (unsigned)d.C); q = q2;
  q2 = q / (unsigned)d.F; f = (int)(q - q2 * (unsigned)d.F);
  b = (int)q2;
}
__global__ __launch_bounds__(256) void q_sample_cond4_kernel(const float4* __restrict__ x0, const float4* __restrict__ noise,
                                                              const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                              const float* __restrict__ sb, float4* __restrict__ xo,
                                                              float4* __restrict__ to, wdno_cond_desc d, unsigned total4) {
  const unsigned stride = gridDim.x * 256u;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += stride) {
    int b, f, c, h, w;
    decode4(d, i, b, f, c, h, w);
    const float4 v0 = x0[i], nz = noise[i];
    const int64_t tb = t[b];
    const float ka = sa[tb], kb = sb[tb];
    const float v[4] = {v0.x, v0.y, v0.z, v0.w}, z[4] = {nz.x, nz.y, nz.z, nz.w};
    float xr[4], tr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int code = cond_code(d, f, c, h, w + e);
      const float xv = ka * v[e] + kb * z[e];
      xr[e] = code == 0 ? xv : (code == 1 ? 0.0f : v[e]);
      tr[e] = code == 0 ? z[e] : 0.0f;
    }
    xo[i] = make_float4(xr[0], xr[1], xr[2], xr[3]);
    to[i] = make_float4(tr[0], tr[1], tr[2], tr[3]);
  }
}
__global__ __launch_bounds__(256) void apply_cond4_kernel(float4* __restrict__ x, const float4* __restrict__ src, wdno_cond_desc d, unsigned total4) {
  const unsigned stride = gridDim.x * 256u;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += stride) {
    int b, f, c, h, w;
    decode4(d, i, b, f, c, h, w);
    int code[4];
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) { code[e] = cond_code(d, f, c, h, w + e); any |= code[e] != 0; }
    if (!any) continue;
    const float4 xv = x[i], sv = src[i];
    float o[4] = {xv.x, xv.y, xv.z, xv.w};
    const float q[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = code[e] == 1 ? 0.0f : (code[e] == 2 ? q[e] : o[e]);
    x[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}
static inline bool vec4_ok(const wdno_cond_desc* c, int64_t total, const void* a, const void* b, const void* e, const void* f) {
  return (c->W & 3) == 0 && total < (1ll << 31) && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)e) | ((uintptr_t)f)) & 15) == 0;
}
static int check_cond(const wdno_cond_desc* c) {
  if (!c || (c->tree != 0 && c->tree != 1)) return WDNO_EINVAL;
  if (c->B <= 0 || c->F <= 0 || c->C <= 0 || c->H <= 0 || c->W <= 0) return WDNO_EINVAL;
  if (c->tree == 1 && c->F != 1) return WDNO_EINVAL;
  return WDNO_OK;
}
extern "C" int wdno_q_sample_cond(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                                  float* x_out, float* target_out, const wdno_cond_desc* c, wdno_stream_t s) {
  int rc = check_cond(c);
  if (rc) return rc;
  int64_t total = (int64_t)c->B * c->F * c->C * c->H * c->W;
  if (vec4_ok(c, total, x0, noise, x_out, target_out)) {
    q_sample_cond4_kernel<<<stream_grid(total / 4, 256), 256, 0, as_stream(s)>>>((const float4*)x0, (const float4*)noise, t, sqrt_ac, sqrt_1mac,
                                                                             (float4*)x_out, (float4*)target_out, *c, (unsigned)(total / 4));
    return wdno_check_launch();
  }
  q_sample_cond_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x0, noise, t, sqrt_ac, sqrt_1mac, x_out, target_out, *c, total);
  return wdno_check_launch();
}

__global__ __launch_bounds__(256) void apply_cond_kernel(float* __restrict__ x, const float* __restrict__ src, wdno_cond_desc d, int64_t total) {
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int b, f, c, h, w;
    decode(d, i, b, f, c, h, w);
    int code = cond_code(d, f, c, h, w);
    if (code == 1) x[i] = 0.0f;
    else if (code == 2) x[i] = src[i];
  }
}
extern "C" int wdno_apply_cond(float* x, const float* src, const wdno_cond_desc* c, wdno_stream_t s) {
  int rc = check_cond(c);
  if (rc) return rc;
  int64_t total = (int64_t)c->B * c->F * c->C * c->H * c->W;
  if (vec4_ok(c, total, x, src, nullptr, nullptr)) {
    apply_cond4_kernel<<<stream_grid(total / 4, 256), 256, 0, as_stream(s)>>>((float4*)x, (const float4*)src, *c, (unsigned)(total / 4));
    return wdno_check_launch();
  }
  apply_cond_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x, src, *c, total);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- weighted MSE
__global__ __launch_bounds__(256) void wmse_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                    const float* __restrict__ wc, const float* __restrict__ wb, float inv_count,
                                                    float* __restrict__ grad, double* __restrict__ ws, int64_t total,
                                                    int64_t per_sample, int C, int64_t inner) {
  __shared__ double red[4];
  int64_t stride = (int64_t)gridDim.x * 256;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int64_t b = i / per_sample;
    int c = (int)((i / inner) % C);
    float wgt = wc[c] * wb[b];
    float df = out[i] - tgt[i];
    acc += (double)(df * df) * (double)wgt;
    if (grad) grad[i] = 2.0f * df * wgt * inv_count;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void wmse_final_kernel(const double* __restrict__ ws, int nb, float* __restrict__ out, float scale) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += ws[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * (double)scale);
}
// float4 forms, the sample on blockIdx.y (no 64-bit divisions; the weight of a float4 is one product: inner % 4 == 0)
template <bool BWD>
__global__ __launch_bounds__(256) void wmse4_kernel(const float4* __restrict__ out, const float4* __restrict__ tgt,
                                                     const float* __restrict__ wc, const float* __restrict__ wb, float inv_count,
                                                     const float* __restrict__ gscale, float4* __restrict__ grad, double* __restrict__ ws,
                                                     unsigned per4, int C, unsigned inner4) {
  __shared__ double red[4];
  const float wbb = wb[blockIdx.y];
  const float gs = BWD ? 2.0f * inv_count * (gscale ? gscale[0] : 1.0f) : 0.f;
  const size_t base = (size_t)blockIdx.y * per4;
  double acc = 0.0;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < per4; i += gridDim.x * 256u) {
    const unsigned c = (i / inner4) % (unsigned)C;
    const float wgt = wc[c] * wbb;
    const float4 o = out[base + i], tg = tgt[base + i];
    const float df[4] = {o.x - tg.x, o.y - tg.y, o.z - tg.z, o.w - tg.w};
    if (BWD) {
      grad[base + i] = make_float4(df[0] * wgt * gs, df[1] * wgt * gs, df[2] * wgt * gs, df[3] * wgt * gs);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += (double)(df[e] * df[e]) * (double)wgt;
      if (grad) grad[base + i] = make_float4(2.0f * df[0] * wgt * inv_count, 2.0f * df[1] * wgt * inv_count, 2.0f * df[2] * wgt * inv_count,
                                             2.0f * df[3] * wgt * inv_count);
    }
  }
  if (!BWD) {
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}
static inline bool wmse4_ok(int64_t B, int64_t per_sample, int64_t inner, const void* a, const void* b, const void* c) {
  return (inner & 3) == 0 && (per_sample & 3) == 0 && per_sample / 4 < (1ll << 31) && B <= 2048 &&
         ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}
extern "C" size_t wdno_weighted_mse_ws_bytes(int64_t n) { return (size_t)stream_grid(n, 256) * sizeof(double); }
extern "C" int wdno_weighted_mse(const float* out, const float* target, const float* wc, const float* wb, float inv_count,
                                 float* loss, float* grad, int64_t B, int64_t per_sample, int C, int64_t inner,
                                 void* ws, size_t ws_bytes, wdno_stream_t s) {
  WDNO_REQUIRE(B > 0 && per_sample > 0 && C > 0 && inner > 0);
  int64_t total = B * per_sample;
  if (ws_bytes < wdno_weighted_mse_ws_bytes(total)) return WDNO_EWORKSPACE;
  int nb = stream_grid(total, 256);
  if (wmse4_ok(B, per_sample, inner, out, target, grad) && nb >= B) {
    const int gx = (int)std::min<int64_t>(cdiv64(per_sample / 4, 256), nb / B);         // gx * B partial sums <= the workspace's nb
    wmse4_kernel<false><<<dim3((unsigned)gx, (unsigned)B), 256, 0, as_stream(s)>>>((const float4*)out, (const float4*)target, wc, wb, inv_count, nullptr,
                                                                                 (float4*)grad, (double*)ws, (unsigned)(per_sample / 4), C, (unsigned)(inner / 4));
    wmse_final_kernel<<<1, 256, 0, as_stream(s)>>>((const double*)ws, gx * (int)B, loss, inv_count);
    return wdno_check_launch();
  }
  wmse_kernel<<<nb, 256, 0, as_stream(s)>>>(out, target, wc, wb, inv_count, grad, (double*)ws, total, per_sample, C, inner);
  wmse_final_kernel<<<1, 256, 0, as_stream(s)>>>((const double*)ws, nb, loss, inv_count);
  return wdno_check_launch();
}

// backward of the weighted MSE: grad = 2 (out - target) * wc[c] * wb[b] * inv_count * gscale[0]
__global__ __launch_bounds__(256) void wmse_bwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                        const float* __restrict__ wc, const float* __restrict__ wb, float inv_count,
                                                        const float* __restrict__ gscale, float* __restrict__ grad, int64_t total,
                                                        int64_t per_sample, int C, int64_t inner) {
  int64_t stride = (int64_t)gridDim.x * 256;
  const float gs = 2.0f * inv_count * (gscale ? gscale[0] : 1.0f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int64_t b = i / per_sample;
    int c = (int)((i / inner) % C);
    grad[i] = (out[i] - tgt[i]) * (wc[c] * wb[b]) * gs;
  }
}
extern "C" int wdno_weighted_mse_bwd(const float* out, const float* target, const float* wc, const float* wb, float inv_count,
                                     const float* gscale, float* grad, int64_t B, int64_t per_sample, int C, int64_t inner,
                                     wdno_stream_t s) {
  WDNO_REQUIRE(B > 0 && per_sample > 0 && C > 0 && inner > 0 && grad != nullptr);
  int64_t total = B * per_sample;
  if (wmse4_ok(B, per_sample, inner, out, target, grad)) {
    const int gx = (int)std::min<int64_t>(cdiv64(per_sample / 4, 256), std::max<int64_t>(1, 2048 / B));
    wmse4_kernel<true><<<dim3((unsigned)gx, (unsigned)B), 256, 0, as_stream(s)>>>((const float4*)out, (const float4*)target, wc, wb, inv_count, gscale,
                                                                                (float4*)grad, nullptr, (unsigned)(per_sample / 4), C, (unsigned)(inner / 4));
    return wdno_check_launch();
  }
  wmse_bwd_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(out, target, wc, wb, inv_count, gscale, grad, total, per_sample, C, inner);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- sampler updates
__global__ __launch_bounds__(256) void p_sample_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                        const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                        const float* __restrict__ c1, const float* __restrict__ c2,
                                                        const float* __restrict__ m1, const float* __restrict__ m2,
                                                        const float* __restrict__ lv, float* __restrict__ xn, float* __restrict__ xs,
                                                        int64_t total, int64_t per_sample, int clamp) {
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int64_t tb = t[i / per_sample];
    float xv = x[i];
    float st = c1[tb] * xv - c2[tb] * eps[i];
    if (clamp) st = fminf(fmaxf(st, -1.0f), 1.0f);
    float mean = m1[tb] * st + m2[tb] * xv;
    float o = mean;
    if (noise) o = mean + expf(0.5f * lv[tb]) * noise[i];
    xn[i] = o;
    if (xs) xs[i] = st;
  }
}
// float4 forms with the sample on blockIdx.y: the step coefficients are block-uniform (one gather per block instead of a 64-bit division
// and five dependent gathers per element); per-element arithmetic unchanged
__global__ __launch_bounds__(256) void p_sample4_kernel(const float4* __restrict__ x, const float4* __restrict__ eps,
                                                         const float4* __restrict__ noise, const int64_t* __restrict__ t,
                                                         const float* __restrict__ c1, const float* __restrict__ c2,
                                                         const float* __restrict__ m1, const float* __restrict__ m2,
                                                         const float* __restrict__ lv, float4* __restrict__ xn, float4* __restrict__ xs,
                                                         unsigned per4, int clamp) {
  const int64_t tb = t[blockIdx.y];
  const float k1 = c1[tb], k2 = c2[tb], q1 = m1[tb], q2 = m2[tb];
  const float sg = noise ? expf(0.5f * lv[tb]) : 0.f;
  const size_t base = (size_t)blockIdx.y * per4;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < per4; i += gridDim.x * 256u) {
    const float4 xv4 = x[base + i], ev4 = eps[base + i];
    const float4 nz4 = noise ? noise[base + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, ev[4] = {ev4.x, ev4.y, ev4.z, ev4.w}, nz[4] = {nz4.x, nz4.y, nz4.z, nz4.w};
    float o[4], st4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float st = k1 * xv[e] - k2 * ev[e];
      if (clamp) st = fminf(fmaxf(st, -1.0f), 1.0f);
      const float mean = q1 * st + q2 * xv[e];
      o[e] = noise ? mean + sg * nz[e] : mean;
      st4[e] = st;
    }
    xn[base + i] = make_float4(o[0], o[1], o[2], o[3]);
    if (xs) xs[base + i] = make_float4(st4[0], st4[1], st4[2], st4[3]);
  }
}
__global__ __launch_bounds__(256) void ddim4_kernel(const float4* __restrict__ x, const float4* __restrict__ eps,
                                                     const float4* __restrict__ noise, const int64_t* __restrict__ t,
                                                     const float* __restrict__ c1, const float* __restrict__ c2, float sqrt_an, float cc,
                                                     float sigma, const float* __restrict__ coef_dev, float4* __restrict__ xn,
                                                     float4* __restrict__ xs, unsigned per4) {
  if (coef_dev) { sqrt_an = coef_dev[0]; cc = coef_dev[1]; sigma = coef_dev[2]; }
  const int64_t tb = t[blockIdx.y];
  const float k1 = c1[tb], k2 = c2[tb];
  const size_t base = (size_t)blockIdx.y * per4;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < per4; i += gridDim.x * 256u) {
    const float4 xv4 = x[base + i], ev4 = eps[base + i];
    const float4 nz4 = noise ? noise[base + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, ev[4] = {ev4.x, ev4.y, ev4.z, ev4.w}, nz[4] = {nz4.x, nz4.y, nz4.z, nz4.w};
    float o[4], st4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = k1 * xv[e];
      const float st = fminf(fmaxf(a - k2 * ev[e], -1.0f), 1.0f);
      const float e2 = (a - st) / k2;
      o[e] = noise ? st * sqrt_an + cc * e2 + sigma * nz[e] : st;
      st4[e] = st;
    }
    xn[base + i] = make_float4(o[0], o[1], o[2], o[3]);
    if (xs) xs[base + i] = make_float4(st4[0], st4[1], st4[2], st4[3]);
  }
}
static inline bool upd4_ok(int64_t B, int64_t per_sample, const void* a, const void* b, const void* c, const void* d, const void* e) {
  return (per_sample & 3) == 0 && per_sample / 4 < (1ll << 31) && B <= 65535 &&
         ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) | ((uintptr_t)e)) & 15) == 0;
}
static inline dim3 upd4_grid(int64_t B, int64_t per4) {
  int64_t gx = cdiv64(per4, 256);
  const int64_t cap = cdiv64(2048, B);              // ~2048 blocks in all
  if (gx > cap) gx = cap;
  return dim3((unsigned)gx, (unsigned)B);
}
extern "C" int wdno_p_sample_update(const float* x, const float* eps, const float* noise, const int64_t* t,
                                    const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, const float* pm1, const float* pm2,
                                    const float* plogvar, float* x_next, float* x_start, int64_t B, int64_t per_sample, int clamp,
                                    wdno_stream_t s) {
  WDNO_REQUIRE(B > 0 && per_sample > 0);
  int64_t total = B * per_sample;
  if (upd4_ok(B, per_sample, x, eps, noise, x_next, x_start)) {
    p_sample4_kernel<<<upd4_grid(B, per_sample / 4), 256, 0, as_stream(s)>>>((const float4*)x, (const float4*)eps, (const float4*)noise, t, sqrt_recip_ac,
                                                                           sqrt_recipm1_ac, pm1, pm2, plogvar, (float4*)x_next, (float4*)x_start,
                                                                           (unsigned)(per_sample / 4), clamp);
    return wdno_check_launch();
  }
  p_sample_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x, eps, noise, t, sqrt_recip_ac, sqrt_recipm1_ac, pm1, pm2, plogvar,
                                                                   x_next, x_start, total, per_sample, clamp);
  return wdno_check_launch();
}

__global__ __launch_bounds__(256) void ddim_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                    const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                    const float* __restrict__ c1, const float* __restrict__ c2, float sqrt_an, float cc,
                                                    float sigma, const float* __restrict__ coef_dev, float* __restrict__ xn,
                                                    float* __restrict__ xs, int64_t total, int64_t per_sample) {
  int64_t stride = (int64_t)gridDim.x * 256;
  if (coef_dev) {       // step coefficients in device memory: the launch is identical for every step (HIP-graph replay)
    sqrt_an = coef_dev[0];
    cc = coef_dev[1];
    sigma = coef_dev[2];
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int64_t tb = t[i / per_sample];
    float a = c1[tb] * x[i];
    float st = fminf(fmaxf(a - c2[tb] * eps[i], -1.0f), 1.0f);
    float e2 = (a - st) / c2[tb];
    float o = st;
    if (noise) o = st * sqrt_an + cc * e2 + sigma * noise[i];
    xn[i] = o;
    if (xs) xs[i] = st;
  }
}
extern "C" int wdno_ddim_update(const float* x, const float* eps, const float* noise, const int64_t* t,
                                const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, float sqrt_an, float c, float sigma,
                                float* x_next, float* x_start, int64_t B, int64_t per_sample, wdno_stream_t s) {
  WDNO_REQUIRE(B > 0 && per_sample > 0);
  int64_t total = B * per_sample;
  if (upd4_ok(B, per_sample, x, eps, noise, x_next, x_start)) {
    ddim4_kernel<<<upd4_grid(B, per_sample / 4), 256, 0, as_stream(s)>>>((const float4*)x, (const float4*)eps, (const float4*)noise, t, sqrt_recip_ac,
                                                                       sqrt_recipm1_ac, sqrt_an, c, sigma, nullptr, (float4*)x_next, (float4*)x_start,
                                                                       (unsigned)(per_sample / 4));
    return wdno_check_launch();
  }
  ddim_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x, eps, noise, t, sqrt_recip_ac, sqrt_recipm1_ac, sqrt_an, c, sigma,
                                                               nullptr, x_next, x_start, total, per_sample);
  return wdno_check_launch();
}
extern "C" int wdno_ddim_update_dev(const float* x, const float* eps, const float* noise, const int64_t* t,
                                    const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, const float* coef_dev,
                                    float* x_next, float* x_start, int64_t B, int64_t per_sample, wdno_stream_t s) {
  WDNO_REQUIRE(B > 0 && per_sample > 0 && coef_dev != nullptr && noise != nullptr);
  int64_t total = B * per_sample;
  if (upd4_ok(B, per_sample, x, eps, noise, x_next, x_start)) {
    ddim4_kernel<<<upd4_grid(B, per_sample / 4), 256, 0, as_stream(s)>>>((const float4*)x, (const float4*)eps, (const float4*)noise, t, sqrt_recip_ac,
                                                                       sqrt_recipm1_ac, 0.f, 0.f, 0.f, coef_dev, (float4*)x_next, (float4*)x_start,
                                                                       (unsigned)(per_sample / 4));
    return wdno_check_launch();
  }
  ddim_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x, eps, noise, t, sqrt_recip_ac, sqrt_recipm1_ac, 0.f, 0.f, 0.f,
                                                               coef_dev, x_next, x_start, total, per_sample);
  return wdno_check_launch();
}
