// pack.hip -- dataset packing on the GPU: the U-Net input of the smoke task from the raw coefficient arrays the offline transform
// stores per simulation, for a whole batch in ONE launch.
//
// Replaces, per sample, Smoke_wave.__getitem__ of smoke/ddpm/data_2d.py:156-221 (base-resolution models): cat of the five fields' eight sub-bands,
// zero padding from [nt, nx, nx] to [pad_t, pad_x, pad_x], the initial-density condition channel (the four sub-bands of the 2-D DWT of rho(t = 0),
// each shown for pad_t / 4 consecutive frames), the smoke-out condition channel (approximation band over the upper half of the image, detail band
// over the lower half), the permute to frames-before-channels and the division by the per-channel RESCALER -- ten torch launches and five passes
// over the 51.6 MB state in round 5 (0.18 ms for a 17 us transform). HBM-bound: B * (8 F nt nx^2 + ...) * 4 bytes in, B pad_t (8 F + 2) pad_x^2 * 4
// out (26.6 + 51.6 MB at the bench batch). The division is IEEE (what torch's `state / RESCALER` computes): results are bit-identical to the
// torch formulation (tests/test_gpu_data.py).
#include "common.h"

#define PK_MAXL 16
#define PK_LB 6          // filters up to this length take the all-rows-in-flight form of the initial-density transform
// n / d as one multiply-high for 0 <= n, n * d < 2^32 (csrc/dwt.hip: FastDiv): a hardware integer division is ~30 VALU instructions on gfx950, and
// four of them per float4 made this kernel ALU-bound (33 us for 78 MB, profiles/r06_pack_kernel_stats.md)
struct PkDiv { unsigned d, m; };
static inline PkDiv pk_make_div(int d) { PkDiv f; f.d = (unsigned)d; f.m = d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1ull) : 0u; return f; }
__device__ __forceinline__ unsigned pk_div(unsigned n, PkDiv f) { return f.d == 1 ? n : __umulhi(n, f.m); }

struct PackSmokeP {
  const float* coef; const float* init; const float* so; const int64_t* idx; const float* resc; float* out;
  int64_t coef_sim, init_sim, so_sim;      // elements between consecutive simulations in the three stores
  int F, nt, nx, pad_t, pad_x, C;
  PkDiv dW4, dPX;
  // FIELDS form: the two condition channels transformed in place from the physical fields (zero-mode analysis):
  // init = rho(t = 0) [.][H0][W0], so = the smoke-out curve [.][T0]; flo / fhi = the FLIPPED decomposition filters (flo[m] = dec_lo[L - 1 - m])
  int L, H0, W0, T0;
  float flo[PK_MAXL], fhi[PK_MAXL];
};

// zero-mode analysis sample k of a line of n values at stride `st`: y[k] = sum_m f[m] x[2 k + m - p], p = (2 L - 3) / 2, x = 0 outside the line
__device__ __forceinline__ float pk_analysis(const float* __restrict__ x, int n, int64_t st, int k, const float* __restrict__ f, int L) {
  const int j0 = 2 * k - (2 * L - 3) / 2;
  float a = 0.f;
  for (int m = 0; m < L; ++m) {
    const int j = j0 + m;
    if (j >= 0 && j < n) a = fmaf(f[m], x[j * st], a);
  }
  return a;
}

// MODE 0: coefficient stores (the three arrays the offline transform wrote); 1: FIELDS (coefficient tensor + physical condition inputs);
// 2: FILL -- as 1, but the rows [f < nt][h < nx] of the 8 F channels are somebody else's (wdno_dwt_fwd_packed stored them pad_x wide, already
// divided): only the zero rows / frames around them and the two condition channels are written.
// grid = (B * pad_t, blocks per frame): a block works inside ONE frame of one sample, the linear index it decodes is < C pad_x^2 / 4.
template <int MODE>
__global__ __launch_bounds__(256) void pack_smoke_state_kernel(PackSmokeP p) {
  constexpr bool FIELDS = MODE != 0;
  const unsigned W4 = (unsigned)p.pad_x >> 2, stride = gridDim.y * 256u;
  const int rep = p.pad_t >> 2, half = p.pad_x >> 1;
  __shared__ float tp[2][PK_MAXL];                     // FIELDS: the flipped filters (indexed by a loop counter: not from the kernel arguments)
  if (FIELDS) {
    if (threadIdx.x < 2 * PK_MAXL) tp[threadIdx.x / PK_MAXL][threadIdx.x % PK_MAXL] = threadIdx.x < PK_MAXL ? p.flo[threadIdx.x] : p.fhi[threadIdx.x - PK_MAXL];
    __syncthreads();
  }
  const unsigned fstride4 = (unsigned)p.C * (unsigned)p.pad_x * W4;      // float4s of a frame = between consecutive frames of a sample
  const int b = (int)(blockIdx.x / (unsigned)p.pad_t), f = (int)(blockIdx.x - (unsigned)b * (unsigned)p.pad_t);
  const int64_t sim = p.idx ? p.idx[b] : b;
  float4* __restrict__ out = reinterpret_cast<float4*>(p.out) + (size_t)blockIdx.x * fstride4;
  for (unsigned i = blockIdx.y * 256u + threadIdx.x; i < fstride4; i += stride) {
    unsigned q = pk_div(i, p.dW4);
    const int w0 = (int)(i - q * W4) * 4;
    const unsigned cc = pk_div(q, p.dPX);
    const int h = (int)(q - cc * (unsigned)p.pad_x), c = (int)cc;
    const float r = p.resc[c];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < 8 * p.F) {
      if (f < p.nt && h < p.nx) {
        if (MODE == 2) continue;                                        // a row of the box: written pad_x wide by the transform's store
        const float* src = p.coef + sim * p.coef_sim + (((int64_t)c * p.nt + f) * p.nx + h) * p.nx + w0;       // [field][band] = [c / 8][c % 8]: consecutive
        if (w0 + 3 < p.nx && !(p.nx & 1) && !(p.coef_sim & 1)) {                // rows of an even number of floats: 8-byte loads
          const float2 a = reinterpret_cast<const float2*>(src)[0], bb = reinterpret_cast<const float2*>(src)[1];
          v[0] = a.x; v[1] = a.y; v[2] = bb.x; v[3] = bb.y;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (w0 + e < p.nx) v[e] = src[e];
        }
      }
    } else if (c == 8 * p.F) {
      if (FIELDS) {
        // sub-band q4 = f / rep of the 2-D transform of rho(t = 0), the same for the rep frames that show it: the thread of the band's FIRST frame computes
        // it and stores all rep frames, the threads of the other frames have nothing to do. H filter = q4 & 1, W filter = q4 >> 1 (csrc/dwt.hip's band
        // order), W pass first: per source row the 4 outputs share a window of 2 * 3 + L samples.
        const int q4 = f / rep;
        if (f != q4 * rep) continue;
        if (h < p.nx) {
          const float* fh = tp[q4 & 1];
          const float* fw = tp[q4 >> 1];
          const float* img = p.init + sim * p.init_sim;
          const int pd = (2 * p.L - 3) / 2, r0 = 2 * h - pd, j0 = 2 * w0 - pd;
          if (p.L <= PK_LB) {
            // short filters (every wavelet of the smoke task): ALL source rows requested before the first one is used -- the few threads of this channel are
            // the launch's critical path (a chain of L dependent row loads was ~20 of the fill launch's 22.7 us). Same sums in the same order as below.
            float xs[PK_LB][6 + PK_LB];
#pragma unroll
            for (int m = 0; m < PK_LB; ++m) {
              const int rr = r0 + m;
              const bool ok = m < p.L && rr >= 0 && rr < p.H0;
              const float* row = img + (int64_t)(ok ? rr : 0) * p.W0;
#pragma unroll
              for (int t = 0; t < 6 + PK_LB; ++t) { const int jj = j0 + t; xs[m][t] = (ok && t < 6 + p.L && jj >= 0 && jj < p.W0) ? row[jj] : 0.f; }
            }
#pragma unroll
            for (int m = 0; m < PK_LB; ++m) {
              const int rr = r0 + m;
              if (m < p.L && rr >= 0 && rr < p.H0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float a = 0.f;
#pragma unroll
                  for (int t = 0; t < PK_LB; ++t) if (t < p.L) a = fmaf(fw[t], xs[m][2 * e + t], a);
                  v[e] = fmaf(fh[m], a, v[e]);
                }
              }
            }
          } else
          for (int m = 0; m < p.L; ++m) {
            const int rr = r0 + m;
            if (rr < 0 || rr >= p.H0) continue;
            const float* row = img + (int64_t)rr * p.W0;
            float xs[6 + PK_MAXL];
#pragma unroll
            for (int t = 0; t < 6 + PK_MAXL; ++t) { const int jj = j0 + t; xs[t] = (t < 6 + p.L && jj >= 0 && jj < p.W0) ? row[jj] : 0.f; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float a = 0.f;
#pragma unroll
              for (int t = 0; t < PK_MAXL; ++t) if (t < p.L) a = fmaf(fw[t], xs[2 * e + t], a);
              v[e] = fmaf(fh[m], a, v[e]);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) if (w0 + e >= p.nx) v[e] = 0.f;
        }
        const float4 o = make_float4(v[0] / r, v[1] / r, v[2] / r, v[3] / r);
        for (int k = 0; k < rep; ++k) out[i + (unsigned)k * fstride4] = o;
        continue;
      }
      if (h < p.nx) {
        const float* src = p.init + sim * p.init_sim + ((int64_t)(f / rep) * p.nx + h) * p.nx;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (w0 + e < p.nx) v[e] = src[w0 + e];
      }
    } else if (f < p.nt) {
      const float s = FIELDS ? pk_analysis(p.so + sim * p.so_sim, p.T0, 1, f, tp[h >= half ? 1 : 0], p.L)
                             : p.so[sim * p.so_sim + (int64_t)(h >= half ? 1 : 0) * p.nt + f];
      v[0] = v[1] = v[2] = v[3] = s;
    }
    out[i] = make_float4(v[0] / r, v[1] / r, v[2] / r, v[3] / r);
  }
}

static inline dim3 pk_grid(int64_t B, int pad_t, int C, int pad_x) {
  const int64_t per_frame4 = (int64_t)C * pad_x * (pad_x >> 2);
  int gx = (int)((per_frame4 + 256 * 8 - 1) / (256 * 8));              // ~8 float4 per thread
  if (gx < 1) gx = 1;
  return dim3((unsigned)(B * pad_t), (unsigned)gx);
}
static inline int pk_check(int64_t B, int pad_t, int pad_x, int C) {
  if ((pad_x & 3) || (pad_t & 3)) return WDNO_EUNSUPPORTED;            // four sub-bands over pad_t frames; 16-byte rows
  if (B * pad_t >= (1ll << 31) || (int64_t)C * pad_x * (pad_x >> 2) * pad_x >= (1ll << 31) || (int64_t)C * pad_x * (pad_x >> 2) > 65535ll * 256 * 8)
    return WDNO_EUNSUPPORTED;           // grid; pk_div range
  return WDNO_OK;
}

static void pk_common(PackSmokeP& p, int64_t B, int F, int nt, int nx, int pad_t, int pad_x) {
  p.F = F; p.nt = nt; p.nx = nx; p.pad_t = pad_t; p.pad_x = pad_x; p.C = 8 * F + 2;
  p.dW4 = pk_make_div(pad_x >> 2); p.dPX = pk_make_div(pad_x);
  p.L = 0; p.H0 = p.W0 = p.T0 = 0;
}

extern "C" int wdno_pack_smoke_state(const float* coef, int64_t coef_sim_stride, const float* init_coef, int64_t init_sim_stride, const float* smokeout,
                                     int64_t so_sim_stride, const int64_t* idx, const float* rescaler, float* state, int64_t B, int F, int nt, int nx,
                                     int pad_t, int pad_x, wdno_stream_t s) {
  WDNO_REQUIRE(coef && init_coef && smokeout && rescaler && state && B > 0 && F > 0 && nt > 0 && nx > 0);
  WDNO_REQUIRE(nt <= pad_t && nx <= pad_x && coef_sim_stride >= (int64_t)F * 8 * nt * nx * nx && init_sim_stride >= 4ll * nx * nx && so_sim_stride >= 2ll * nt);
  const int C = 8 * F + 2;
  int rc = pk_check(B, pad_t, pad_x, C);
  if (rc) return rc;
  PackSmokeP p;
  p.coef = coef; p.init = init_coef; p.so = smokeout; p.idx = idx; p.resc = rescaler; p.out = state;
  p.coef_sim = coef_sim_stride; p.init_sim = init_sim_stride; p.so_sim = so_sim_stride;
  pk_common(p, B, F, nt, nx, pad_t, pad_x);
  pack_smoke_state_kernel<0><<<pk_grid(B, pad_t, C, pad_x), 256, 0, as_stream(s)>>>(p);
  return wdno_check_launch();
}

// The same packing with the two condition channels computed from the PHYSICAL inputs inside the launch (the online pipeline fields -> state:
// smoke/wave_trans_2d.py:150-170 + data_2d.py:156-221): rho0 [.][H0][W0] (the density at t = 0, e.g. a view into the fields tensor: its
// per-simulation stride is given), curve [.][T0] (the smoke-out fraction per frame); zero-mode analysis with the wavelet's decomposition
// filters dec_lo / dec_hi (L taps, host pointers). nx == (W0 + L - 1) / 2 == (H0 + L - 1) / 2, nt == (T0 + L - 1) / 2.
// coef == NULL: FILL -- the rows [f < nt][h < nx] of the 8 F channels are left alone (wdno_dwt_fwd_packed wrote them), everything else is written.
extern "C" int wdno_pack_smoke_fields(const float* coef, int64_t coef_sim_stride, const float* rho0, int64_t rho0_sim_stride, const float* curve,
                                      int64_t curve_sim_stride, const float* dec_lo_host, const float* dec_hi_host, int L, const float* rescaler,
                                      float* state, int64_t B, int F, int nt, int nx, int pad_t, int pad_x, int H0, int W0, int T0, wdno_stream_t s) {
  WDNO_REQUIRE(rho0 && curve && dec_lo_host && dec_hi_host && rescaler && state && B > 0 && F > 0 && nt > 0 && nx > 0);
  WDNO_REQUIRE(L >= 2 && L <= PK_MAXL && nt <= pad_t && nx <= pad_x && (!coef || coef_sim_stride >= (int64_t)F * 8 * nt * nx * nx));
  const int C = 8 * F + 2;
  int rc = pk_check(B, pad_t, pad_x, C);
  if (rc) return rc;
  if (nx != (W0 + L - 1) / 2 || nx != (H0 + L - 1) / 2 || nt != (T0 + L - 1) / 2) return WDNO_EUNSUPPORTED;
  PackSmokeP p;
  p.coef = coef; p.init = rho0; p.so = curve; p.idx = nullptr; p.resc = rescaler; p.out = state;
  p.coef_sim = coef_sim_stride; p.init_sim = rho0_sim_stride; p.so_sim = curve_sim_stride;
  pk_common(p, B, F, nt, nx, pad_t, pad_x);
  p.L = L; p.H0 = H0; p.W0 = W0; p.T0 = T0;
  for (int m = 0; m < L; ++m) { p.flo[m] = dec_lo_host[L - 1 - m]; p.fhi[m] = dec_hi_host[L - 1 - m]; }
  if (coef) pack_smoke_state_kernel<1><<<pk_grid(B, pad_t, C, pad_x), 256, 0, as_stream(s)>>>(p);
  else pack_smoke_state_kernel<2><<<pk_grid(B, pad_t, C, pad_x), 256, 0, as_stream(s)>>>(p);
  return wdno_check_launch();
}
