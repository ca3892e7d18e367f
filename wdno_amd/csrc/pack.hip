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

struct PackSmokeP {
  const float* coef; const float* init; const float* so; const int64_t* idx; const float* resc; float* out;
  int64_t coef_sim, init_sim, so_sim;      // elements between consecutive simulations in the three stores
  int F, nt, nx, pad_t, pad_x, C;
  unsigned total4;
};

__global__ __launch_bounds__(256) void pack_smoke_state_kernel(PackSmokeP p) {
  const unsigned W4 = (unsigned)p.pad_x >> 2, stride = gridDim.x * 256u;
  const int rep = p.pad_t >> 2, half = p.pad_x >> 1;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < p.total4; i += stride) {
    unsigned q = i / W4;
    const int w0 = (int)(i - q * W4) * 4;
    unsigned q2 = q / (unsigned)p.pad_x; const int h = (int)(q - q2 * (unsigned)p.pad_x); q = q2;
    q2 = q / (unsigned)p.C; const int c = (int)(q - q2 * (unsigned)p.C); q = q2;
    q2 = q / (unsigned)p.pad_t; const int f = (int)(q - q2 * (unsigned)p.pad_t);
    const int b = (int)q2;
    const int64_t sim = p.idx ? p.idx[b] : b;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < 8 * p.F) {
      if (f < p.nt && h < p.nx) {
        const float* src = p.coef + sim * p.coef_sim + (((int64_t)c * p.nt + f) * p.nx + h) * p.nx;       // [field][band] = [c / 8][c % 8]: consecutive
#pragma unroll
        for (int e = 0; e < 4; ++e) if (w0 + e < p.nx) v[e] = src[w0 + e];
      }
    } else if (c == 8 * p.F) {
      if (h < p.nx) {
        const float* src = p.init + sim * p.init_sim + ((int64_t)(f / rep) * p.nx + h) * p.nx;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (w0 + e < p.nx) v[e] = src[w0 + e];
      }
    } else if (f < p.nt) {
      const float s = p.so[sim * p.so_sim + (int64_t)(h >= half ? 1 : 0) * p.nt + f];
      v[0] = v[1] = v[2] = v[3] = s;
    }
    const float r = p.resc[c];
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
  pack_smoke_state_kernel<<<stream_grid(total4, 256), 256, 0, as_stream(s)>>>(p);
  return wdno_check_launch();
}
