// pointwise.hip -- HBM-bound elementwise, layout and reduction kernels (gfx950).
// All kernels are grid-stride with 16-byte accesses where the layout allows it.
#include "common.h"
extern int wdno_debug_mode;      // api.cpp (WDNO_DEBUG)

// ---------------------------------------------------------------------------------------------- activations
template <int ACT>
__device__ __forceinline__ float act_apply(float x) {
  if (ACT == 0) return silu_f(x);
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
template <int ACT>
__device__ __forceinline__ float act_grad(float x) {
  if (ACT == 0) return silu_grad_f(x);
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

template <int ACT>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * 256;
  int64_t n4 = n >> 2;
  for (int64_t k = i; k < n4; k += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[k];
    v.x = act_apply<ACT>(v.x); v.y = act_apply<ACT>(v.y); v.z = act_apply<ACT>(v.z); v.w = act_apply<ACT>(v.w);
    reinterpret_cast<float4*>(y)[k] = v;
  }
  for (int64_t k = (n4 << 2) + i; k < n; k += stride) y[k] = act_apply<ACT>(x[k]);
}
template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * 256;
  int64_t n4 = n >> 2;
  for (int64_t k = i; k < n4; k += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[k];
    float4 g = reinterpret_cast<const float4*>(dy)[k];
    g.x *= act_grad<ACT>(v.x); g.y *= act_grad<ACT>(v.y); g.z *= act_grad<ACT>(v.z); g.w *= act_grad<ACT>(v.w);
    reinterpret_cast<float4*>(dx)[k] = g;
  }
  for (int64_t k = (n4 << 2) + i; k < n; k += stride) dx[k] = dy[k] * act_grad<ACT>(x[k]);
}

extern "C" int wdno_act_fwd(const float* x, float* y, int64_t n, int act, wdno_stream_t s) {
  WDNO_REQUIRE(n >= 0 && (act == 0 || act == 1));
  if (n == 0) return WDNO_OK;
  int grid = stream_grid(n / 4 + 1, 256);
  if (act == 0) act_fwd_kernel<0><<<grid, 256, 0, as_stream(s)>>>(x, y, n);
  else act_fwd_kernel<1><<<grid, 256, 0, as_stream(s)>>>(x, y, n);
  return wdno_check_launch();
}
extern "C" int wdno_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int act, wdno_stream_t s) {
  WDNO_REQUIRE(n >= 0 && (act == 0 || act == 1));
  if (n == 0) return WDNO_OK;
  int grid = stream_grid(n / 4 + 1, 256);
  if (act == 0) act_bwd_kernel<0><<<grid, 256, 0, as_stream(s)>>>(x, dy, dx, n);
  else act_bwd_kernel<1><<<grid, 256, 0, as_stream(s)>>>(x, dy, dx, n);
  return wdno_check_launch();
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ o, int64_t n, float* __restrict__ amax_rec) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * 256;
  int64_t n4 = n >> 2;
  float am = 0.f;
  for (int64_t k = i; k < n4; k += stride) {
    float4 u = reinterpret_cast<const float4*>(a)[k];
    float4 v = reinterpret_cast<const float4*>(b)[k];
    u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
    reinterpret_cast<float4*>(o)[k] = u;
    am = amax4(am, u);
  }
  for (int64_t k = (n4 << 2) + i; k < n; k += stride) {
    float r = a[k] + b[k];
    o[k] = r;
    am = fmaxf(am, fabsf(r));
  }
  if (amax_rec) amax_record_emit(am, amax_rec, blockIdx.x);
}
extern "C" int wdno_add_amax(const float* a, const float* b, float* out, float* amax_rec, int64_t n, wdno_stream_t s) {
  WDNO_REQUIRE(n >= 0);
  if (n == 0) return WDNO_OK;
  add_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, as_stream(s)>>>(a, b, out, n, amax_rec);
  return wdno_check_launch();
}
extern "C" int wdno_add(const float* a, const float* b, float* out, int64_t n, wdno_stream_t s) {
  return wdno_add_amax(a, b, out, nullptr, n, s);
}

// ---------------------------------------------------------------------------------------------- time embedding
__global__ void sinusoidal_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int B, int dim) {
  int half = dim >> 1;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  int b = i / half, k = i - b * half;
  float a = (float)t[b] * freqs[k];
  out[(int64_t)b * dim + k] = sinf(a);
  out[(int64_t)b * dim + half + k] = cosf(a);
}
extern "C" int wdno_sinusoidal_emb(const int64_t* t, const float* freqs, float* out, int B, int dim, wdno_stream_t s) {
  WDNO_REQUIRE(B > 0 && dim >= 4 && (dim % 2) == 0 && freqs != nullptr);
  int n = B * (dim / 2);
  sinusoidal_kernel<<<cdiv(n, 128), 128, 0, as_stream(s)>>>(t, freqs, out, B, dim);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- layout transforms
// src [N][C][S] -> dst [N][S][Cp]: 32x32 tile transpose through LDS; reads coalesced along S, writes along C.
__global__ __launch_bounds__(256) void nc_to_cl_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        int C, int64_t S, int Cp) {
  __shared__ float tile[32][33];
  int64_t n = blockIdx.z;
  int64_t s0 = (int64_t)blockIdx.x * 32;
  int c0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* sp = src + n * (int64_t)C * S;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int c = c0 + ty + r * 8;
    int64_t sidx = s0 + tx;
    tile[ty + r * 8][tx] = (c < C && sidx < S) ? sp[(int64_t)c * S + sidx] : 0.0f;
  }
  __syncthreads();
  float* dp = dst + n * S * (int64_t)Cp;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t sidx = s0 + ty + r * 8;
    int c = c0 + tx;
    if (sidx < S && c < Cp) dp[sidx * Cp + c] = tile[tx][ty + r * 8];
  }
}
__global__ __launch_bounds__(256) void cl_to_nc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        int C, int64_t S, int Cp) {
  __shared__ float tile[32][33];
  int64_t n = blockIdx.z;
  int64_t s0 = (int64_t)blockIdx.x * 32;
  int c0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* sp = src + n * S * (int64_t)Cp;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t sidx = s0 + ty + r * 8;
    int c = c0 + tx;
    tile[ty + r * 8][tx] = (sidx < S && c < C) ? sp[sidx * Cp + c] : 0.0f;
  }
  __syncthreads();
  float* dp = dst + n * (int64_t)C * S;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int c = c0 + ty + r * 8;
    int64_t sidx = s0 + tx;
    if (c < C && sidx < S) dp[(int64_t)c * S + sidx] = tile[tx][ty + r * 8];
  }
}
extern "C" int wdno_nc_to_cl(const float* src, float* dst, int64_t N, int C, int64_t S, int Cp, wdno_stream_t s) {
  WDNO_REQUIRE(N > 0 && C > 0 && S > 0 && Cp >= C && N < 65536);
  dim3 grid((unsigned)cdiv64(S, 32), (unsigned)cdiv(Cp, 32), (unsigned)N);
  nc_to_cl_kernel<<<grid, 256, 0, as_stream(s)>>>(src, dst, C, S, Cp);
  return wdno_check_launch();
}
extern "C" int wdno_cl_to_nc(const float* src, float* dst, int64_t N, int C, int64_t S, int Cp, wdno_stream_t s) {
  WDNO_REQUIRE(N > 0 && C > 0 && S > 0 && Cp >= C && N < 65536);
  dim3 grid((unsigned)cdiv64(S, 32), (unsigned)cdiv(C, 32), (unsigned)N);
  cl_to_nc_kernel<<<grid, 256, 0, as_stream(s)>>>(src, dst, C, S, Cp);
  return wdno_check_launch();
}

// out[p] = (a[p] | b[p]) ; Ca, Cb multiples of 4
__global__ __launch_bounds__(256) void concat2_kernel(const float4* __restrict__ a, int Ca4, const float4* __restrict__ b, int Cb4,
                                                       float4* __restrict__ out, int64_t total4, float* __restrict__ amax_rec) {
  int Ct4 = Ca4 + Cb4;
  int64_t stride = (int64_t)gridDim.x * 256;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    int64_t p = i / Ct4;
    int c = (int)(i - p * Ct4);
    float4 v = (c < Ca4) ? a[p * Ca4 + c] : b[p * Cb4 + (c - Ca4)];
    out[i] = v;
    am = amax4(am, v);
  }
  if (amax_rec) amax_record_emit(am, amax_rec, blockIdx.x);
}
__global__ __launch_bounds__(256) void split2_kernel(const float4* __restrict__ in, float4* __restrict__ a, int Ca4,
                                                      float4* __restrict__ b, int Cb4, int64_t total4) {
  int Ct4 = Ca4 + Cb4;
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    int64_t p = i / Ct4;
    int c = (int)(i - p * Ct4);
    float4 v = in[i];
    if (c < Ca4) a[p * Ca4 + c] = v; else b[p * Cb4 + (c - Ca4)] = v;
  }
}
extern "C" int wdno_concat2_cl_amax(const float* a, int Ca, const float* b, int Cb, float* out, float* amax_rec, int64_t P,
                                    wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0);
  int64_t total4 = P * ((Ca + Cb) / 4);
  concat2_kernel<<<stream_grid(total4, 256), 256, 0, as_stream(s)>>>((const float4*)a, Ca / 4, (const float4*)b, Cb / 4, (float4*)out,
                                                                     total4, amax_rec);
  return wdno_check_launch();
}
// (a | b) delivered ONLY as the fp16 planes of the convolutions that read it (the up-path concat of a U-Net level feeds block1's
// convolution and the 1 x 1 skip projection of the next ResnetBlock, nothing else): scale from max(max|a|, max|b|) = the exact maximum
// of the result, known from the two amax records before the pass. A thread owns one 8-channel group (Ca, Cb multiples of 8).
typedef _Float16 cc_half8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void concat2_planes_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                                              const float* __restrict__ rec_a, const float* __restrict__ rec_b,
                                                              _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                              float* __restrict__ scale_out, int64_t P) {
  const bool single = lo == nullptr;                 // one bf16 plane, no scale
  const float s = single ? 1.0f : scale_from_amax(fmaxf(amax_record_read(rec_a), amax_record_read(rec_b)));
  if (!single && blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
  const int Ct = Ca + Cb, g8 = Ct >> 3;
  const int64_t total = P * g8, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t p = i / g8;
    const int c0 = (int)(i - p * g8) * 8;
    const float* src = c0 < Ca ? a + p * Ca + c0 : b + p * Cb + (c0 - Ca);
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    cc_half8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) { _Float16 th, tl; plane_pack(v[e], s, single, th, tl); h[e] = th; l[e] = tl; }
    *reinterpret_cast<cc_half8*>(hi + p * Ct + c0) = h;
    if (!single) *reinterpret_cast<cc_half8*>(lo + p * Ct + c0) = l;
  }
}
extern "C" int wdno_concat2_cl_planes(const float* a, int Ca, const float* b, int Cb, const float* rec_a, const float* rec_b, void* hi,
                                      void* lo, float* scale_out, int64_t P, wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0 && hi != nullptr);
  WDNO_REQUIRE(lo == nullptr || (rec_a && rec_b && scale_out));
  const int64_t total = P * ((Ca + Cb) / 8);
  concat2_planes_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(a, Ca, b, Cb, rec_a, rec_b, (_Float16*)hi, (_Float16*)lo, scale_out, P);
  return wdno_check_launch();
}
extern "C" int wdno_concat2_cl(const float* a, int Ca, const float* b, int Cb, float* out, int64_t P, wdno_stream_t s) {
  return wdno_concat2_cl_amax(a, Ca, b, Cb, out, nullptr, P, s);
}
extern "C" int wdno_split2_cl(const float* in, float* a, int Ca, float* b, int Cb, int64_t P, wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0);
  int64_t total4 = P * ((Ca + Cb) / 4);
  split2_kernel<<<stream_grid(total4, 256), 256, 0, as_stream(s)>>>((const float4*)in, (float4*)a, Ca / 4, (float4*)b, Cb / 4, total4);
  return wdno_check_launch();
}

// nearest x2 on CL [N,H,W,C] -> [N,2H,2W,C]
__global__ __launch_bounds__(256) void up2x_fwd_kernel(const float4* __restrict__ in, float4* __restrict__ out,
                                                        int H, int W, int C4, int64_t total4) {
  int64_t stride = (int64_t)gridDim.x * 256;
  int W2 = 2 * W, H2 = 2 * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    int c = (int)(i % C4);
    int64_t p = i / C4;
    int w = (int)(p % W2);
    int64_t q = p / W2;
    int h = (int)(q % H2);
    int64_t n = q / H2;
    out[i] = in[((n * H + (h >> 1)) * W + (w >> 1)) * C4 + c];
  }
}
__global__ __launch_bounds__(256) void up2x_bwd_kernel(const float4* __restrict__ dout, float4* __restrict__ din,
                                                        int H, int W, int C4, int64_t total4) {
  int64_t stride = (int64_t)gridDim.x * 256;
  int W2 = 2 * W, H2 = 2 * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    int c = (int)(i % C4);
    int64_t p = i / C4;
    int w = (int)(p % W);
    int64_t q = p / W;
    int h = (int)(q % H);
    int64_t n = q / H;
    int64_t base = ((n * H2 + 2 * h) * W2 + 2 * w) * C4 + c;
    float4 a = dout[base], b = dout[base + C4], d = dout[base + (int64_t)W2 * C4], e = dout[base + (int64_t)W2 * C4 + C4];
    float4 r;
    r.x = (a.x + b.x) + (d.x + e.x); r.y = (a.y + b.y) + (d.y + e.y);
    r.z = (a.z + b.z) + (d.z + e.z); r.w = (a.w + b.w) + (d.w + e.w);
    din[i] = r;
  }
}
extern "C" int wdno_upsample2x_cl_fwd(const float* in, float* out, int64_t N, int H, int W, int C, wdno_stream_t s) {
  WDNO_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  int64_t total4 = N * 4 * H * W * (C / 4);
  up2x_fwd_kernel<<<stream_grid(total4, 256), 256, 0, as_stream(s)>>>((const float4*)in, (float4*)out, H, W, C / 4, total4);
  return wdno_check_launch();
}
extern "C" int wdno_upsample2x_cl_bwd(const float* dout, float* din, int64_t N, int H, int W, int C, wdno_stream_t s) {
  WDNO_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  int64_t total4 = N * H * W * (C / 4);
  up2x_bwd_kernel<<<stream_grid(total4, 256), 256, 0, as_stream(s)>>>((const float4*)dout, (float4*)din, H, W, C / 4, total4);
  return wdno_check_launch();
}

// coefficient nearest up-sampling: in [outer,a,mid,b,c] -> out [outer,a*fa,mid,b*fb,c*fc]
__global__ __launch_bounds__(256) void upsample_coef_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int a, int mid, int b, int c, int fa, int fb, int fc, int64_t total) {
  int64_t stride = (int64_t)gridDim.x * 256;
  int oc = c * fc, ob = b * fb, oa = a * fa;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int ic = (int)(i % oc);
    int64_t r = i / oc;
    int ib = (int)(r % ob); r /= ob;
    int im = (int)(r % mid); r /= mid;
    int ia = (int)(r % oa);
    int64_t o = r / oa;
    out[i] = in[(((o * a + ia / fa) * mid + im) * b + ib / fb) * c + ic / fc];
  }
}
extern "C" int wdno_upsample_coef(const float* in, float* out, int64_t outer, int a, int mid, int b, int c,
                                  int fa, int fb, int fc, wdno_stream_t s) {
  WDNO_REQUIRE(outer > 0 && a > 0 && mid > 0 && b > 0 && c > 0 && fa > 0 && fb > 0 && fc > 0);
  int64_t total = outer * a * fa * mid * b * fb * c * fc;
  upsample_coef_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(in, out, a, mid, b, c, fa, fb, fc, total);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- column sums
// out[c] = sum_p in[p][c]. Stage 1: block b sums rows [b*rpb, (b+1)*rpb) in double -> ws[b][C]; stage 2: sum over blocks.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ in, double* __restrict__ ws,
                                                              int64_t P, int C, int64_t rows_per_block) {
  __shared__ double red[4][64];
  int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  for (int c0 = blockIdx.y * 64; c0 < C; c0 += 64 * gridDim.y) {      // column tiles over blockIdx.y: few rows x many columns
    int c = c0 + tx;                                                    // (a [64, 2048] matrix was ONE block walking 32 tiles: 51 us)
    double acc = 0.0;
    if (c < C)
      for (int64_t r = r0 + ty; r < r1; r += 4) acc += (double)in[r * C + c];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < C) ws[(int64_t)blockIdx.x * C + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    __syncthreads();
  }
}
static inline int colsum_blocks(int64_t P) {
  int64_t nb = cdiv64(P, 64);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}
// few rows (the per-sample parameter-gradient pieces of a GroupNorm: [N, 2 C] with N = batch): one thread per column, fp64 sum over the
// rows in order -- one launch instead of the partial + finish pair (30 such sums per smoke train step were 13 us each)
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int P, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0;
  for (int p = 0; p < P; ++p) a += (double)in[(int64_t)p * C + c];
  out[c] = (float)a;
}
extern "C" size_t wdno_colsum_ws_bytes(int64_t P, int C) { return (size_t)colsum_blocks(P) * (size_t)C * sizeof(double); }
extern "C" int wdno_colsum(const float* in, float* out, int64_t P, int C, void* ws, size_t ws_bytes, wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && C > 0);
  if (P <= 64 && wdno_debug_mode != 37) {          // debug 37: the two-launch path (A/B)
    colsum_rows_kernel<<<cdiv(C, 256), 256, 0, as_stream(s)>>>(in, out, (int)P, C);
    return wdno_check_launch();
  }
  if (ws_bytes < wdno_colsum_ws_bytes(P, C)) return WDNO_EWORKSPACE;
  int nb = colsum_blocks(P);
  int64_t rpb = cdiv64(P, nb);
  int gy = cdiv(C, 64);
  if ((int64_t)gy * nb > 2048) gy = 2048 / nb > 1 ? 2048 / nb : 1;
  colsum_partial_kernel<<<dim3(nb, gy), 256, 0, as_stream(s)>>>(in, (double*)ws, P, C, rpb);
  partial_rows_sum_kernel<double><<<cdiv(C, 32), PRS_THREADS, 0, as_stream(s)>>>((const double*)ws, out, nb, C);
  return wdno_check_launch();
}

// Many row sums in one launch (include/wdno_hip.h: wdno_rows_sum_multi): items by value in the kernel arguments, a block = 32 columns of one item
struct RowsSumArgs { wdno_rows_sum_item it[WDNO_ROWS_SUM_MAX]; int n; };
__global__ __launch_bounds__(PRS_THREADS) void rows_sum_multi_kernel(RowsSumArgs a) {
  int b = (int)blockIdx.x, i = 0;
  for (; i < a.n; ++i) {                       // (uniform: scalar registers)
    const int nb = (a.it[i].ncols + 31) >> 5;
    if (b < nb) break;
    b -= nb;
  }
  if (i >= a.n) return;
  const wdno_rows_sum_item& t = a.it[i];
  if (t.is_double) partial_rows_sum_body<double>((const double*)t.part + t.col0, t.out, t.rows, t.ncols, b, t.stride);
  else partial_rows_sum_body<float>((const float*)t.part + t.col0, t.out, t.rows, t.ncols, b, t.stride);
}
extern "C" int wdno_rows_sum_multi(const wdno_rows_sum_item* items, int n_items, wdno_stream_t s) {
  WDNO_REQUIRE(n_items >= 0 && (items || n_items == 0));
  for (int base = 0; base < n_items; base += WDNO_ROWS_SUM_MAX) {
    RowsSumArgs a;
    a.n = n_items - base < WDNO_ROWS_SUM_MAX ? n_items - base : WDNO_ROWS_SUM_MAX;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      a.it[i] = items[base + i];
      WDNO_REQUIRE(a.it[i].part && a.it[i].out && a.it[i].rows > 0 && a.it[i].ncols > 0 && a.it[i].col0 >= 0 && a.it[i].stride >= a.it[i].col0 + a.it[i].ncols);
      blocks += cdiv(a.it[i].ncols, 32);
    }
    rows_sum_multi_kernel<<<blocks, PRS_THREADS, 0, as_stream(s)>>>(a);
  }
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- trainer step
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ ws) {
  __shared__ double red[4];
  int64_t stride = (int64_t)gridDim.x * 256;
  int64_t n4 = n >> 2;
  double acc = 0.0;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += stride) {
    float4 v = reinterpret_cast<const float4*>(g)[k];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  for (int64_t k = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) acc += (double)g[k] * g[k];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sum_final_kernel(const double* __restrict__ ws, int nb, float* __restrict__ out, float scale) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += ws[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * (double)scale);
}
extern "C" size_t wdno_sumsq_ws_bytes(int64_t n) { return (size_t)stream_grid(n / 4 + 1, 256) * sizeof(double); }
extern "C" int wdno_sumsq(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, wdno_stream_t s) {
  WDNO_REQUIRE(n > 0);
  if (ws_bytes < wdno_sumsq_ws_bytes(n)) return WDNO_EWORKSPACE;
  int nb = stream_grid(n / 4 + 1, 256);
  sumsq_partial_kernel<<<nb, 256, 0, as_stream(s)>>>(g, n, (double*)ws);
  sum_final_kernel<<<1, 256, 0, as_stream(s)>>>((const double*)ws, nb, out, 1.0f);
  return wdno_check_launch();
}

// torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (single-tensor formulation) on a flat buffer
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const float* __restrict__ sumsq,
                                                    float max_norm, float grad_scale, float step_size, float beta1, float beta2,
                                                    float eps, float bc2_sqrt) {
  float coef = grad_scale;
  if (sumsq != nullptr && max_norm > 0.0f) {
    float total = sqrtf(sumsq[0]) * grad_scale;
    float c = max_norm / (total + 1e-6f);
    coef *= fminf(c, 1.0f);
  }
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
    float gr = g[k] * coef;
    float mk = m[k] + (gr - m[k]) * (1.0f - beta1);   // exp_avg.lerp_(grad, 1 - beta1)
    float vk = v[k] * beta2 + (1.0f - beta2) * gr * gr;
    m[k] = mk;
    v[k] = vk;
    float denom = sqrtf(vk) / bc2_sqrt + eps;
    p[k] = p[k] - step_size * (mk / denom);
  }
}
extern "C" int wdno_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                                   float grad_scale, float lr, float beta1, float beta2, float eps, int step, wdno_stream_t s) {
  WDNO_REQUIRE(n > 0 && step >= 1);
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float bc2_sqrt = (float)sqrt(bc2);
  adam_kernel<<<stream_grid(n, 256), 256, 0, as_stream(s)>>>(p, g, m, v, n, sumsq, max_norm, grad_scale, step_size, beta1, beta2, eps, bc2_sqrt);
  return wdno_check_launch();
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ p, int64_t n, float w) {
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) e[k] = e[k] + (p[k] - e[k]) * w;  // lerp_(p, 1-beta)
}
extern "C" int wdno_ema_update(float* ema, const float* p, int64_t n, float beta, wdno_stream_t s) {
  WDNO_REQUIRE(n > 0);
  ema_kernel<<<stream_grid(n, 256), 256, 0, as_stream(s)>>>(ema, p, n, 1.0f - beta);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- gradient gather
// dst[i] = src[i] (or 0 when src is NULL) for a device-resident table of (src, dst, n) items: ONE launch moves every parameter
// gradient autograd produced into its span of the flat gradient buffer. It replaces ~230 torch `add` launches per step that
// AccumulateGrad issues when .grad already exists (3.9 % of the smoke training step), and the memset of the flat buffer.
__global__ __launch_bounds__(256) void gather_items_kernel(const wdno_copy_item* __restrict__ tab) {
  const wdno_copy_item it = tab[blockIdx.y];
  const float* __restrict__ src = (const float*)it.src;
  float* __restrict__ dst = (float*)it.dst;
  const int64_t n = it.n;
  if (src == dst || (int64_t)blockIdx.x * 256 >= n) return;       // already in place / a small item needs only its first blocks
  const int64_t stride = (int64_t)gridDim.x * 256;
  // spans start at arbitrary element offsets of the flat buffer: 4-byte accesses (fully coalesced), four in flight per thread
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * stride;
      v[u] = (src && i < n) ? src[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n) dst[i] = v[u];
    }
  }
}
extern "C" int wdno_gather_items(const void* table, int n_items, int blocks_per_item, wdno_stream_t s) {
  WDNO_REQUIRE(table && n_items > 0 && n_items <= 65535 && blocks_per_item > 0);
  gather_items_kernel<<<dim3((unsigned)blocks_per_item, (unsigned)n_items), 256, 0, as_stream(s)>>>((const wdno_copy_item*)table);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- relative-position bias
// bias[h][i][j] = W[bucket[i][j]][h] (T5-style table lookup, conv3d.py:106-112) and its gradient dW[b][h] = sum_{bucket[i][j] = b}
// dbias[h][i][j]. 32 x 4 table, 24 x 24 positions: one block each way. torch's embedding backward sorts the 576 indices with a
// radix sort and launches four kernels for this.
__global__ __launch_bounds__(256) void relpos_fwd_kernel(const float* __restrict__ w, const int64_t* __restrict__ bucket, float* __restrict__ out,
                                                          int nn, int heads) {
  for (int e = threadIdx.x; e < heads * nn; e += 256) {
    const int h = e / nn, ij = e - h * nn;
    out[e] = w[bucket[ij] * heads + h];
  }
}
// Deterministic: output (bucket, head) belongs to ONE thread that scans the nn positions in index order (an LDS float atomicAdd per
// position summed in arrival order, so the relative_attention_bias gradient differed from run to run -- and with it every replica
// consistency / replay check of a training run). <= 32 x heads outputs of nn terms: the operands are staged in LDS first.
template <bool STAGED>
__global__ __launch_bounds__(256) void relpos_bwd_kernel(const float* __restrict__ dbias, const int64_t* __restrict__ bucket, float* __restrict__ dw,
                                                          int nn, int heads, int nb) {
  extern __shared__ float sm[];            // STAGED: [heads * nn] dbias, then [nn] bucket indices
  int* sb = (int*)(sm + (size_t)heads * nn);
  if (STAGED) {
    for (int e = threadIdx.x; e < heads * nn; e += 256) sm[e] = dbias[e];
    for (int e = threadIdx.x; e < nn; e += 256) sb[e] = (int)bucket[e];
    __syncthreads();
  }
  for (int o = threadIdx.x; o < nb * heads; o += 256) {
    const int bk = o / heads, h = o - bk * heads;
    float acc = 0.f;
    for (int ij = 0; ij < nn; ++ij) {
      const int bb = STAGED ? sb[ij] : (int)bucket[ij];
      const float v = STAGED ? sm[h * nn + ij] : dbias[(size_t)h * nn + ij];
      if (bb == bk) acc += v;
    }
    dw[o] = acc;
  }
}
extern "C" int wdno_relpos_bias_fwd(const float* w, const int64_t* bucket, float* out, int n, int heads, wdno_stream_t s) {
  WDNO_REQUIRE(n > 0 && heads > 0);
  relpos_fwd_kernel<<<1, 256, 0, as_stream(s)>>>(w, bucket, out, n * n, heads);
  return wdno_check_launch();
}
extern "C" int wdno_relpos_bias_bwd(const float* dbias, const int64_t* bucket, float* dw, int n, int heads, int num_buckets, wdno_stream_t s) {
  WDNO_REQUIRE(n > 0 && heads > 0 && num_buckets > 0 && num_buckets * heads <= 8192);
  const size_t lds = (size_t)n * n * (heads + 1) * sizeof(float);
  if (lds <= 48 * 1024) relpos_bwd_kernel<true><<<1, 256, lds, as_stream(s)>>>(dbias, bucket, dw, n * n, heads, num_buckets);
  else relpos_bwd_kernel<false><<<1, 256, 0, as_stream(s)>>>(dbias, bucket, dw, n * n, heads, num_buckets);
  return wdno_check_launch();
}
