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

// ------------------------------------------------------------------------------------------------ all projections of one input in one launch
// Every ResnetBlock projects the SAME activated time embedding x [P][C] with its own nn.Linear (conv3d.py:118-133, unet.py:151-165): 16 of
// them in the smoke U-Net, ~20 in the Burgers one, forward and backward ~80 launches of 5..20 us per training step (plus one transposed
// weight copy and one gradient add per layer). Here the layers form one table (wdno_linear_item: weight, bias, K, first feature index); a
// "feature" f is a row of the concatenation of all weights:
//   forward  y_i[p][k] = sum_c x[p][c] w_i[k][c] + b_i[k]        one wave per feature; y = [P][K_i] blocks, block i at P * k_start_i
//   wgrad    dw[f][c]  = sum_p dy[p][f] x[p][c], db[f] = sum_p dy[p][f]     dw / db = the concatenations over the layers (views per layer)
//   dgrad    dx[p][c]  = sum_f dy[p][f] w[f][c]      blocks of LRM_CHUNK features leave partial sums; a second launch adds them in order
// (no transposed weight copies: the rows of w are read as they lie).
#define LRM_CHUNK 32
__device__ __forceinline__ int lrm_find(const wdno_linear_item* __restrict__ tab, int n_items, int f) {
  int i = 0;
  while (i + 1 < n_items && tab[i + 1].k_start <= f) ++i;     // (<= 64 layers; the table is wave-uniform: scalar loads)
  return i;
}
__global__ __launch_bounds__(256) void linear_multi_fwd_kernel(const wdno_linear_item* __restrict__ tab, int n_items, int F,
                                                                const float* __restrict__ x, float* __restrict__ y, int P, int C) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int f = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (f >= F) return;
  const int i = lrm_find(tab, n_items, f);
  const int ks = tab[i].k_start, K = tab[i].K, k = f - ks;
  const float* __restrict__ w = (const float*)tab[i].w + (int64_t)k * C;
  const float* __restrict__ bias = (const float*)tab[i].bias;
  const int p0 = blockIdx.y * LR_MAXP, np = min(LR_MAXP, P - p0);
  float acc[LR_MAXP];
#pragma unroll
  for (int p = 0; p < LR_MAXP; ++p) acc[p] = 0.f;
  const float4* wr = reinterpret_cast<const float4*>(w);
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
          if (p < np) {
            const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)(p0 + p) * C)[c4];
            acc[p] += (xv.x * wv[u].x + xv.y * wv[u].y) + (xv.z * wv[u].z + xv.w * wv[u].w);       // the sums of linear_rows_fwd_kernel
          }
      }
    }
  }
  const float b = bias ? bias[k] : 0.f;
  float* yo = y + (int64_t)P * ks + k;
#pragma unroll
  for (int p = 0; p < LR_MAXP; ++p)
    if (p < np) {
      const float s = wave_sum(acc[p]);
      if (lane == 0) yo[(int64_t)(p0 + p) * K] = s + b;
    }
}
__global__ __launch_bounds__(256) void linear_multi_wgrad_kernel(const wdno_linear_item* __restrict__ tab, int n_items, int F,
                                                                  const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dw, float* __restrict__ db, int P, int C) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int f = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (f >= F) return;
  const int i = lrm_find(tab, n_items, f);
  const int ks = tab[i].k_start, K = tab[i].K, k = f - ks;
  const float* __restrict__ g0 = dy + (int64_t)P * ks + k;     // dy_i[p][k] = g0[p * K]
  float gs = 0.f;
  float4* orow = reinterpret_cast<float4*>(dw + (int64_t)f * C);
  for (int c4 = lane; c4 < (C >> 2) || c4 == lane; c4 += 64) {       // (every lane makes at least one pass: db)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float gsum = 0.f;
    for (int p0 = 0; p0 < P; p0 += LR_MAXP) {
      float g[LR_MAXP];
#pragma unroll
      for (int p = 0; p < LR_MAXP; ++p) {
        g[p] = p0 + p < P ? g0[(int64_t)(p0 + p) * K] : 0.f;
        gsum += g[p];
      }
      if (c4 < (C >> 2)) {
#pragma unroll
        for (int p = 0; p < LR_MAXP; ++p)
          if (p0 + p < P) {
            const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)(p0 + p) * C)[c4];
            a.x += g[p] * xv.x; a.y += g[p] * xv.y; a.z += g[p] * xv.z; a.w += g[p] * xv.w;
          }
      }
    }
    if (c4 < (C >> 2)) orow[c4] = a;
    gs = gsum;
  }
  if (lane == 0) db[f] = gs;
}
// partial[chunk][p][c] = sum over the chunk's LRM_CHUNK features of dy[p][f] w[f][c]. A wave takes every fourth feature of the chunk, its
// lanes the columns c as float4 (coalesced 16-byte loads of the weight row, all eight of a wave in flight; dy[p][f] is wave-uniform); the
// four waves meet in LDS and are added in wave order.
template <int CV>      // float4 columns per lane: C <= 256 CV
__global__ __launch_bounds__(256) void linear_multi_dgrad_kernel(const wdno_linear_item* __restrict__ tab, int n_items, int F,
                                                                  const float* __restrict__ dy, float* __restrict__ partial, int P, int C) {
  __shared__ float4 red[3][64 * CV];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x;
  const int p0 = blockIdx.y * LR_MAXP, np = min(LR_MAXP, P - p0);
  const int f0 = chunk * LRM_CHUNK, f1 = min(F, f0 + LRM_CHUNK);
  const int C4 = C >> 2;
  constexpr int NF = LRM_CHUNK / 4;                  // features per wave
  float4 wv[NF][CV];
  int goff[NF], Kq[NF];                             // dy_i[p][k] = dy[goff + p * K] (host: P * F < 2^31)
  int i = lrm_find(tab, n_items, f0);
#pragma unroll
  for (int u = 0; u < NF; ++u) {
    const int f = __builtin_amdgcn_readfirstlane(min(f0 + wave + 4 * u, f1 - 1));
    while (i + 1 < n_items && tab[i + 1].k_start <= f) ++i;
    const int ks = tab[i].k_start;
    const float4* wr = reinterpret_cast<const float4*>((const float*)tab[i].w + (int64_t)(f - ks) * C);
#pragma unroll
    for (int v = 0; v < CV; ++v) wv[u][v] = lane + 64 * v < C4 ? wr[lane + 64 * v] : make_float4(0.f, 0.f, 0.f, 0.f);
    goff[u] = P * ks + (f - ks);
    Kq[u] = tab[i].K;
  }
  float4 acc[LR_MAXP][CV];
#pragma unroll
  for (int p = 0; p < LR_MAXP; ++p)
#pragma unroll
    for (int v = 0; v < CV; ++v) acc[p][v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < NF; ++u) {
    const bool live = f0 + wave + 4 * u < f1;        // wave-uniform
#pragma unroll
    for (int p = 0; p < LR_MAXP; ++p)
      if (live && p < np) {
        const float g = dy[goff[u] + (p0 + p) * Kq[u]];
#pragma unroll
        for (int v = 0; v < CV; ++v) {
          acc[p][v].x = fmaf(g, wv[u][v].x, acc[p][v].x); acc[p][v].y = fmaf(g, wv[u][v].y, acc[p][v].y);
          acc[p][v].z = fmaf(g, wv[u][v].z, acc[p][v].z); acc[p][v].w = fmaf(g, wv[u][v].w, acc[p][v].w);
        }
      }
  }
  // rows one at a time through LDS: waves 1..3 park their sums, wave 0 adds them in order and stores
#pragma unroll
  for (int p = 0; p < LR_MAXP; ++p) {
    if (p < np) {                                     // block-uniform
      __syncthreads();
      if (wave > 0) {
#pragma unroll
        for (int v = 0; v < CV; ++v) red[wave - 1][lane + 64 * v] = acc[p][v];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int v = 0; v < CV; ++v)
          if (lane + 64 * v < C4) {
            float4 a = acc[p][v];
#pragma unroll
            for (int q = 0; q < 3; ++q) { const float4 t = red[q][lane + 64 * v]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
            reinterpret_cast<float4*>(partial + ((int64_t)chunk * P + p0 + p) * C)[lane + 64 * v] = a;
          }
      }
    }
  }
}
// dx[e] = sum over the chunks of partial[chunk][e]: 32 elements x 8 slices of the chunk list per block, four sums per slice, fixed order
__global__ __launch_bounds__(256) void linear_multi_dgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dx, int n_chunks,
                                                                         int64_t n) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t e = (int64_t)blockIdx.x * 32 + el;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < n) {
    int q = sl;
    for (; q + 24 < n_chunks; q += 32) {
      a0 += partial[(int64_t)q * n + e]; a1 += partial[(int64_t)(q + 8) * n + e];
      a2 += partial[(int64_t)(q + 16) * n + e]; a3 += partial[(int64_t)(q + 24) * n + e];
    }
    for (; q < n_chunks; q += 8) a0 += partial[(int64_t)q * n + e];
  }
  red[sl][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][el];
    dx[e] = t;
  }
}

static int lrm_check(const void* table, int n_items, int F, int P, int C) {
  WDNO_REQUIRE(table && n_items > 0 && n_items <= 1024 && F > 0 && P > 0 && C > 0);
  if (P > LR_MAXROWS || (C & 3)) return WDNO_EUNSUPPORTED;
  return WDNO_OK;
}
extern "C" int wdno_linear_multi_fwd(const void* table, int n_items, int F, const float* x, float* y, int P, int C, wdno_stream_t s) {
  int rc = lrm_check(table, n_items, F, P, C);
  if (rc) return rc;
  linear_multi_fwd_kernel<<<dim3(cdiv(F, 4), cdiv(P, LR_MAXP)), 256, 0, as_stream(s)>>>((const wdno_linear_item*)table, n_items, F, x, y, P, C);
  return wdno_check_launch();
}
extern "C" int wdno_linear_multi_wgrad(const void* table, int n_items, int F, const float* x, const float* dy, float* dw, float* db, int P,
                                       int C, wdno_stream_t s) {
  int rc = lrm_check(table, n_items, F, P, C);
  if (rc) return rc;
  WDNO_REQUIRE(dw && db);
  linear_multi_wgrad_kernel<<<cdiv(F, 4), 256, 0, as_stream(s)>>>((const wdno_linear_item*)table, n_items, F, x, dy, dw, db, P, C);
  return wdno_check_launch();
}
extern "C" size_t wdno_linear_multi_dgrad_ws_bytes(int F, int P, int C) {
  if (F <= 0 || P <= 0 || C <= 0) return 0;
  return (size_t)cdiv(F, LRM_CHUNK) * P * C * sizeof(float);
}
extern "C" int wdno_linear_multi_dgrad(const void* table, int n_items, int F, const float* dy, float* dx, int P, int C, void* ws, size_t ws_bytes,
                                       wdno_stream_t s) {
  int rc = lrm_check(table, n_items, F, P, C);
  if (rc) return rc;
  WDNO_REQUIRE(ws && ws_bytes >= wdno_linear_multi_dgrad_ws_bytes(F, P, C));
  const int nch = cdiv(F, LRM_CHUNK);
  const dim3 grid(nch, cdiv(P, LR_MAXP));
  if (C <= 256) linear_multi_dgrad_kernel<1><<<grid, 256, 0, as_stream(s)>>>((const wdno_linear_item*)table, n_items, F, dy, (float*)ws, P, C);
  else if (C <= 512) linear_multi_dgrad_kernel<2><<<grid, 256, 0, as_stream(s)>>>((const wdno_linear_item*)table, n_items, F, dy, (float*)ws, P, C);
  else return WDNO_EUNSUPPORTED;
  const int64_t n = (int64_t)P * C;
  linear_multi_dgrad_reduce_kernel<<<(unsigned)cdiv64(n, 32), 256, 0, as_stream(s)>>>((const float*)ws, dx, nch, n);
  return wdno_check_launch();
}
