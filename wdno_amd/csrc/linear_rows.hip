// linear_rows.hip -- nn.Linear on a handful of rows (the time-embedding MLPs: 8 samples x 64..512 features).
//
// (any row count up to LR_MAXROWS works -- rows go in groups of 16 -- but the point is the small batches.)
// A train step of the smoke U-Net runs ~50 of these forward and as many backward (time_mlp and the scale/shift projection
// of every ResnetBlock, video_diffusion_pytorch_conv3d.py:118-133, 286-296; burgers unet.py:151-165). As 128 x 128 implicit-GEMM
// tiles each is ONE block walking the whole reduction: ~34 us for 1 MFLOP. Here the weight stays in the reference layout
// [K][C] (no packing), one wave owns one output feature and its lanes split the reduction (coalesced 16-byte loads of the
// weight row, the few x rows come from L1/L2), so a launch is K / 4 short blocks: a couple of microseconds.
//   forward   y[p][k]  = sum_c x[p][c] w[k][c] (+ bias[k]);  y padded to Kp columns with zeros
//   data grad the same kernel on the transposed weight (dx[p][c] = sum_k dy[p][k] wT[c][k]); a kernel that walks w[k][c] itself
//             (one thread per c, four waves over k) was 4x slower per launch: +0.9 ms per smoke step
//   wgrad     dw[k][c] = sum_p dy[p][k] x[p][c], db[k] = sum_p dy[p][k]
#include "common.h"

#define LR_MAXP 16          // rows handled per pass (registers)
#define LR_MAXROWS 1024     // beyond this the implicit-GEMM kernels take over anyway

__global__ __launch_bounds__(256) void linear_rows_fwd_kernel(const float* __restrict__ x, int xs, const float* __restrict__ w, int ws,
                                                               const float* __restrict__ bias, float* __restrict__ y, int P, int C, int K,
                                                               int Kp) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + wave;
  if (k >= Kp) return;
  x += (int64_t)blockIdx.y * LR_MAXP * xs;        // rows in groups of LR_MAXP over blockIdx.y
  y += (int64_t)blockIdx.y * LR_MAXP * Kp;
  P = min(LR_MAXP, P - (int)blockIdx.y * LR_MAXP);
  float acc[LR_MAXP];
#pragma unroll
  for (int p = 0; p < LR_MAXP; ++p) acc[p] = 0.f;
  if (k < K) {
    const float4* wr = reinterpret_cast<const float4*>(w + (int64_t)k * ws);
    // four weight loads in flight per lane: with one, a 2048-wide reduction (the data gradient of the Burgers time projections) was eight
    // dependent memory round trips per wave, ~20 us per launch
    const int C4 = C >> 2;
    for (int c0 = lane; c0 < C4; c0 += 256) {
      float4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wv[u] = c0 + 64 * u < C4 ? wr[c0 + 64 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c4 = c0 + 64 * u;
        if (c4 < C4) {
#pragma unroll
          for (int p = 0; p < LR_MAXP; ++p)
            if (p < P) {
              const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)p * xs)[c4];
              acc[p] += (xv.x * wv[u].x + xv.y * wv[u].y) + (xv.z * wv[u].z + xv.w * wv[u].w);
            }
        }
      }
    }
  }
  const float b = (bias && k < K) ? bias[k] : 0.f;
#pragma unroll
  for (int p = 0; p < LR_MAXP; ++p)
    if (p < P) {
      const float s = wave_sum(acc[p]);
      if (lane == 0) y[(int64_t)p * Kp + k] = s + b;
    }
}

// one wave per output feature k, lanes over the input features (float4): dw row k, and db[k] from lane 0
__global__ __launch_bounds__(256) void linear_rows_wgrad_kernel(const float* __restrict__ x, int xs, const float* __restrict__ dy, int dys,
                                                                 float* __restrict__ dw, float* __restrict__ db, int P, int C, int K) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + wave;
  if (k >= K) return;
  float gs = 0.f;
  float4* orow = reinterpret_cast<float4*>(dw + (int64_t)k * C);
  for (int c4 = lane; c4 < (C >> 2) || c4 == lane; c4 += 64) {       // (every lane makes at least one pass: db)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float gsum = 0.f;
    for (int p0 = 0; p0 < P; p0 += LR_MAXP) {                         // rows in groups of LR_MAXP
      float g[LR_MAXP];
#pragma unroll
      for (int p = 0; p < LR_MAXP; ++p) {
        g[p] = p0 + p < P ? dy[(int64_t)(p0 + p) * dys + k] : 0.f;
        gsum += g[p];
      }
      if (c4 < (C >> 2)) {
#pragma unroll
        for (int p = 0; p < LR_MAXP; ++p)
          if (p0 + p < P) {
            const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)(p0 + p) * xs)[c4];
            a.x += g[p] * xv.x; a.y += g[p] * xv.y; a.z += g[p] * xv.z; a.w += g[p] * xv.w;
          }
      }
    }
    if (c4 < (C >> 2)) orow[c4] = a;
    gs = gsum;
  }
  if (db && lane == 0) db[k] = gs;
}

extern "C" int wdno_linear_rows_fwd(const float* x, int x_stride, const float* w, int w_stride, const float* bias, float* y, int P, int C,
                                    int K, int Kp, wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && C > 0 && K > 0 && Kp >= K && x_stride >= C && w_stride >= C);
  if (P > LR_MAXROWS || (C & 3) || (x_stride & 3) || (w_stride & 3)) return WDNO_EUNSUPPORTED;
  linear_rows_fwd_kernel<<<dim3(cdiv(Kp, 4), cdiv(P, LR_MAXP)), 256, 0, as_stream(s)>>>(x, x_stride, w, w_stride, bias, y, P, C, K, Kp);
  return wdno_check_launch();
}
extern "C" int wdno_linear_rows_wgrad(const float* x, int x_stride, const float* dy, int dy_stride, float* dw, float* db, int P, int C, int K,
                                      wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && C > 0 && K > 0 && x_stride >= C && dy_stride >= K);
  if (P > LR_MAXROWS || (C & 3) || (x_stride & 3)) return WDNO_EUNSUPPORTED;
  linear_rows_wgrad_kernel<<<cdiv(K, 4), 256, 0, as_stream(s)>>>(x, x_stride, dy, dy_stride, dw, db, P, C, K);
  return wdno_check_launch();
}
