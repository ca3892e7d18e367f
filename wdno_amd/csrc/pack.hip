// pack.hip -- dataset packing on the GPU: the U-Net input of the smoke task from the raw coefficient arrays the offline transform
// stores per simulation, for a whole batch in ONE launch.
//
// Replaces, per sample, Smoke_wave.__getitem__ of smoke/ddpm/data_2d.py:156-221 (base-resolution models): cat of the five fields' eight sub-bands,
// zero padding from [nt, nx, nx] to [pad_t, pad_x, pad_x], the initial-density condition channel (the four sub-bands of the 2-D DWT of rho(t = 0),
// each shown for pad_t / 4 consecutive frames), the smoke-out condition channel (approximation band over the upper half of the image, detail band
// over the lower half), the permute to frames-before-channels and the division by the per-channel RESCALER -- ten torch launches and five passes
// over the 12.9 MB state in round 5 (0.18 ms for a 17 us transform). HBM-bound: B * (8 F nt nx^2 + ...) * 4 bytes in, B pad_t (8 F + 2) pad_x^2 * 4
// out (6.7 + 12.9 MB at the bench batch). The division is IEEE (what torch's `state / RESCALER` computes): results are bit-identical to the
// torch formulation (tests/test_gpu_data.py).
#include "common.h"

#define PK_MAXL 16
struct PackSmokeP {
  const float* coef; const float* init; const float* so; const int64_t* idx; const float* resc; float* out;
  int64_t coef_sim, init_sim, so_sim;      // elements between consecutive simulations in the three stores
  int F, nt, nx, pad_t, pad_x, C;
  unsigned total4;
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

template <bool FIELDS>
__global__ __launch_bounds__(256) void pack_smoke_state_kernel(PackSmokeP p) {
  const unsigned W4 = (unsigned)p.pad_x >> 2, stride = gridDim.x * 256u;
  const int rep = p.pad_t >> 2, half = p.pad_x >> 1;
  __shared__ float tp[2][PK_MAXL];                     // FIELDS: the flipped filters (indexed by a loop counter: not from the kernel arguments)
  if (FIELDS) {
    if (threadIdx.x < 2 * PK_MAXL) tp[threadIdx.x / PK_MAXL][threadIdx.x % PK_MAXL] = threadIdx.x < PK_MAXL ? p.flo[threadIdx.x] : p.fhi[threadIdx.x - PK_MAXL];
    __syncthreads();
  }
  const unsigned fstride4 = (unsigned)p.C * (unsigned)p.pad_x * W4;      // float4s between consecutive frames of a sample
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < p.total4; i += stride) {
    unsigned q = i / W4;
    const int w0 = (int)(i - q * W4) * 4;
    unsigned q2 = q / (unsigned)p.pad_x; const int h = (int)(q - q2 * (unsigned)p.pad_x); q = q2;
    q2 = q / (unsigned)p.C; const int c = (int)(q - q2 * (unsigned)p.C); q = q2;
    q2 = q / (unsigned)p.pad_t; const int f = (int)(q - q2 * (unsigned)p.pad_t);
    const int b = (int)q2;
    const int64_t sim = p.idx ? p.idx[b] : b;
    const float r = p.resc[c];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < 8 * p.F) {
      if (f < p.nt && h < p.nx) {
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
        for (int k = 0; k < rep; ++k) reinterpret_cast<float4*>(p.out)[i + (unsigned)k * fstride4] = o;
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
    reinterpret_cast<float4*>(p.out)[i] = make_float4(v[0] / r, v[1] / r, v[2] / r, v[3] / r);
  }
}

extern "C" int wdno_pack_smoke_state(const float* coef, int64_t coef_sim_stride, const float* init_coef, int64_t init_sim_stride, const float* smokeout,
                                     int64_t so_sim_stride, const int64_t* idx, const float* rescaler, float* state, int64_t B, int F, int nt, int nx,
                                     int pad_t, int pad_x, wdno_stream_t s) {
  WDNO_REQUIRE(coef && init_coef && smokeout && rescaler && state && B > 0 && F > 0 && nt > 0 && nx > 0);
  WDNO_REQUIRE(nt <= pad_t && nx <= pad_x && coef_sim_stride >= (int64_t)F * 8 * nt * nx * nx && init_sim_stride >= 4ll * nx * nx && so_sim_stride >= 2ll * nt);
  if ((pad_x & 3) || (pad_t & 3)) return WDNO_EUNSUPPORTED;            // four sub-bands over pad_t frames; 16-byte rows
  const int C = 8 * F + 2;
  const int64_t total4 = B * pad_t * C * pad_x * (pad_x >> 2);
  if (total4 >= (1ll << 31)) return WDNO_EUNSUPPORTED;
  PackSmokeP p;
  p.coef = coef; p.init = init_coef; p.so = smokeout; p.idx = idx; p.resc = rescaler; p.out = state;
  p.coef_sim = coef_sim_stride; p.init_sim = init_sim_stride; p.so_sim = so_sim_stride;
  p.F = F; p.nt = nt; p.nx = nx; p.pad_t = pad_t; p.pad_x = pad_x; p.C = C; p.total4 = (unsigned)total4;
  p.L = 0; p.H0 = p.W0 = p.T0 = 0;
  pack_smoke_state_kernel<false><<<stream_grid(total4, 256), 256, 0, as_stream(s)>>>(p);
  return wdno_check_launch();
}

// The same packing with the two condition channels computed from the PHYSICAL inputs inside the launch (the online pipeline fields -> state:
// smoke/wave_trans_2d.py:150-170 + data_2d.py:156-221): rho0 [.][H0][W0] (the density at t = 0, e.g. a view into the fields tensor: its
// per-simulation stride is given), curve [.][T0] (the smoke-out fraction per frame); zero-mode analysis with the wavelet's decomposition
// filters dec_lo / dec_hi (L taps, host pointers). nx == (W0 + L - 1) / 2 == (H0 + L - 1) / 2, nt == (T0 + L - 1) / 2.
extern "C" int wdno_pack_smoke_fields(const float* coef, int64_t coef_sim_stride, const float* rho0, int64_t rho0_sim_stride, const float* curve,
                                      int64_t curve_sim_stride, const float* dec_lo_host, const float* dec_hi_host, int L, const float* rescaler,
                                      float* state, int64_t B, int F, int nt, int nx, int pad_t, int pad_x, int H0, int W0, int T0, wdno_stream_t s) {
  WDNO_REQUIRE(coef && rho0 && curve && dec_lo_host && dec_hi_host && rescaler && state && B > 0 && F > 0 && nt > 0 && nx > 0);
  WDNO_REQUIRE(L >= 2 && L <= PK_MAXL && nt <= pad_t && nx <= pad_x && coef_sim_stride >= (int64_t)F * 8 * nt * nx * nx);
  if ((pad_x & 3) || (pad_t & 3) || nx != (W0 + L - 1) / 2 || nx != (H0 + L - 1) / 2 || nt != (T0 + L - 1) / 2) return WDNO_EUNSUPPORTED;
  const int C = 8 * F + 2;
  const int64_t total4 = B * pad_t * C * pad_x * (pad_x >> 2);
  if (total4 >= (1ll << 31)) return WDNO_EUNSUPPORTED;
  PackSmokeP p;
  p.coef = coef; p.init = rho0; p.so = curve; p.idx = nullptr; p.resc = rescaler; p.out = state;
  p.coef_sim = coef_sim_stride; p.init_sim = rho0_sim_stride; p.so_sim = curve_sim_stride;
  p.F = F; p.nt = nt; p.nx = nx; p.pad_t = pad_t; p.pad_x = pad_x; p.C = C; p.total4 = (unsigned)total4;
  p.L = L; p.H0 = H0; p.W0 = W0; p.T0 = T0;
  for (int m = 0; m < L; ++m) { p.flo[m] = dec_lo_host[L - 1 - m]; p.fhi[m] = dec_hi_host[L - 1 - m]; }
  pack_smoke_state_kernel<true><<<stream_grid(total4, 256), 256, 0, as_stream(s)>>>(p);
  return wdno_check_launch();
}
