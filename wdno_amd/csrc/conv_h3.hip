// conv_h3.hip -- fp32-equivalent implicit-GEMM convolution on the fp16 matrix cores ("3 x fp16 split").
//
// An fp32 value v (scaled by a per-tensor power of two s so that max|v*s| < 2^15) is represented as hi + lo with
//   hi = fp16(v*s),  lo = fp16(v*s - hi)            (hi carries 11 significant bits, lo the next 11)
// and a product a*b is evaluated as  ah*bh + ah*bl + al*bh  on v_mfma_f32_32x32x16_f16 (fp16 products are exact in the
// fp32 accumulator; the dropped al*bl term and the residual of lo are <= 2^-22 relative). The result is fp32-class
// (tests: <= 2e-6 rel-L2 against fp64), while the matrix pipe runs at 1/3 of the 2.5 PFLOP/s fp16 rate instead of the
// 157 TFLOP/s of the exact-fp32 MFMA. Sub-normal flushing of lo is harmless because of the per-tensor scale: it
// perturbs only elements below 2^-18 of the tensor maximum, by <= 2^-40 of that maximum.
//
// Pre-pass (HBM-bound): wdno_amax -> wdno_split_f16 produce the two fp16 planes [rows][C8] (C padded to 8) and the
// scale. The convolution kernel is the fp32 kernel's structure (conv.hip) with 16-byte loads of 8 halves, LDS planes
// [rows][32+8] halves (80-byte rows: conflict-free ds_read_b128 / ds_write_b128) and 3 MFMAs per fragment pair.
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define HBK 32   // reduction elements per step
#define HST 40   // LDS row stride in halves

// ---------------------------------------------------------------------------------------------- amax / split
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
  __shared__ float red[4];
  int64_t stride = (int64_t)gridDim.x * 256;
  int64_t n4 = n >> 2;
  float m = 0.f;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[k];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t k = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) m = fmaxf(m, fabsf(x[k]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
  }
}
extern "C" int wdno_amax(const float* x, int64_t n, float* amax_zeroed, wdno_stream_t s) {
  WDNO_REQUIRE(n > 0);
  amax_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, as_stream(s)>>>(x, n, reinterpret_cast<unsigned*>(amax_zeroed));
  return wdno_check_launch();
}

__device__ __forceinline__ float scale_from_amax(float amax) {
  if (!(amax > 0.f) || !isfinite(amax)) return 1.0f;
  int e = ilogbf(amax);                       // amax = m * 2^e, 1 <= m < 2
  return ldexpf(1.0f, 14 - e);                // amax * s in [2^14, 2^15)
}
// x [rows][C] fp32 -> hi, lo [rows][C8] fp16 (zero padded), scale_out[0] = s
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, const float* __restrict__ amax,
                                                     _Float16* __restrict__ hi, _Float16* __restrict__ lo, float* __restrict__ scale_out,
                                                     int64_t rows, int C, int C8) {
  const float s = scale_from_amax(amax[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
  const int g8 = C8 >> 3;
  const int64_t total = rows * g8;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int64_t r = i / g8;
    int c0 = (int)(i - r * g8) * 8;
    float v[8];
    const float* xp = x + r * C + c0;
    if (c0 + 8 <= C) {
      float4 a = *reinterpret_cast<const float4*>(xp), b = *reinterpret_cast<const float4*>(xp + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? xp[e] : 0.f;
    }
    half8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[e] * s;
      _Float16 th = (_Float16)t;
      h[e] = th;
      l[e] = (_Float16)(t - (float)th);
    }
    *reinterpret_cast<half8*>(hi + r * C8 + c0) = h;
    *reinterpret_cast<half8*>(lo + r * C8 + c0) = l;
  }
}
extern "C" int wdno_split_f16(const float* x, const float* amax, void* hi, void* lo, float* scale_out, int64_t rows, int C, int C8,
                              wdno_stream_t s) {
  WDNO_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (C8 & 7) == 0 && C8 >= C && C8 < C + 8);
  int64_t total = rows * (C8 / 8);
  split_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x, amax, (_Float16*)hi, (_Float16*)lo, scale_out, rows, C, C8);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- convolution
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_fwd_h3_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                           const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                           const float* __restrict__ sx, const float* __restrict__ sw,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           float* __restrict__ y, ConvP p) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AROWS = BM / 64;   // row passes of the 256 loader threads (64 rows x 4 column groups per pass)
  constexpr int BROWS = BN / 64;
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
  // stage layout: [Ah | Al | Bh | Bl], two stages
  constexpr int STAGE = 2 * (BM + BN) * HST;
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int lrow = tid >> 2;          // 0..63
  const int c8 = (tid & 3) * 8;       // element offset inside the 32-wide chunk
  int a_d0[AROWS], a_h0[AROWS], a_w0[AROWS];
  int64_t a_nbase[AROWS];
  bool a_ok[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int64_t pm = m0 + lrow + 64 * i;
    a_ok[i] = pm < p.P;
    int64_t q = a_ok[i] ? pm : 0;
    int ow = (int)(q % g.OW); q /= g.OW;
    int oh = (int)(q % g.OH); q /= g.OH;
    int od = (int)(q % g.OD);
    int64_t n = q / g.OD;
    a_d0[i] = od * g.sd - g.pd;
    a_h0[i] = oh * g.sh - g.ph;
    a_w0[i] = ow * g.sw - g.pw;
    a_nbase[i] = n * g.D;
  }
  int64_t row_off[AROWS];
  bool row_ok[AROWS];
  int64_t b_off[BROWS];
  bool b_ok[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_ok[i] = (n0 + lrow + 64 * i) < g.K;

  int l_tap = 0, l_chunk = 0, l_dz = 0, l_dy = 0, l_r = c8, l_dx = c8 / g.C, l_cc = c8 % g.C;
  const int dx0 = l_dx, cc0 = l_cc;
  auto refresh_tap = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int d = a_d0[i] + l_dz, h = a_h0[i] + l_dy;
      row_ok[i] = a_ok[i] && d >= 0 && d < g.D && h >= 0 && h < g.H;
      row_off[i] = (((a_nbase[i] + d) * g.H + h) * (int64_t)g.W + a_w0[i]) * g.C;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) b_off[i] = ((int64_t)l_tap * g.K + (n0 + lrow + 64 * i)) * p.R;
  };
  refresh_tap();

  uint4 ah[AROWS], al[AROWS], bh[BROWS], bl[BROWS];
  auto load_tile = [&]() {
    const bool r_ok = l_r < p.R;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int w = a_w0[i] + l_dx;
      uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
      if (row_ok[i] && r_ok && w >= 0 && w < g.W) {
        vh = *reinterpret_cast<const uint4*>(xh + row_off[i] + l_r);
        vl = *reinterpret_cast<const uint4*>(xl + row_off[i] + l_r);
      }
      ah[i] = vh; al[i] = vl;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
      if (b_ok[i] && r_ok) {
        vh = *reinterpret_cast<const uint4*>(wh + b_off[i] + l_r);
        vl = *reinterpret_cast<const uint4*>(wl + b_off[i] + l_r);
      }
      bh[i] = vh; bl[i] = vl;
    }
    ++l_chunk;
    l_r += HBK;
    l_cc += HBK;
    while (l_cc >= g.C) { l_cc -= g.C; ++l_dx; }
    if (l_chunk == p.nchunk) {
      l_chunk = 0; l_r = c8; l_dx = dx0; l_cc = cc0;
      ++l_tap;
      if (++l_dy == g.kh) { l_dy = 0; ++l_dz; }
      refresh_tap();
    }
  };
  auto store_tile = [&](int buf) {
    _Float16* Ah = hsm + buf * STAGE;
    _Float16* Al = Ah + BM * HST;
    _Float16* Bh = Al + BM * HST;
    _Float16* Bl = Bh + BN * HST;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      *reinterpret_cast<uint4*>(&Ah[(lrow + 64 * i) * HST + c8]) = ah[i];
      *reinterpret_cast<uint4*>(&Al[(lrow + 64 * i) * HST + c8]) = al[i];
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      *reinterpret_cast<uint4*>(&Bh[(lrow + 64 * i) * HST + c8]) = bh[i];
      *reinterpret_cast<uint4*>(&Bl[(lrow + 64 * i) * HST + c8]) = bl[i];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, hh = lane >> 5;
  const int m_base = wm * (TM * 32), n_base = wn * (TN * 32);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  load_tile();
  store_tile(0);
  __syncthreads();
  if (p.nsteps > 1) load_tile();
  for (int step = 0; step < p.nsteps; ++step) {
    const _Float16* Ah = hsm + (step & 1) * STAGE;
    const _Float16* Al = Ah + BM * HST;
    const _Float16* Bh = Al + BM * HST;
    const _Float16* Bl = Bh + BN * HST;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 fah[TM], fal[TM], fbh[TN], fbl[TN];
      const int col = ks * 16 + hh * 8;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        fah[a] = *reinterpret_cast<const half8*>(&Ah[(m_base + a * 32 + li) * HST + col]);
        fal[a] = *reinterpret_cast<const half8*>(&Al[(m_base + a * 32 + li) * HST + col]);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        fbh[b] = *reinterpret_cast<const half8*>(&Bh[(n_base + b * 32 + li) * HST + col]);
        fbl[b] = *reinterpret_cast<const half8*>(&Bl[(n_base + b * 32 + li) * HST + col]);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[a], fbh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[a], fbl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[a], fbh[b], acc[a][b], 0, 0, 0);
        }
    }
    if (step + 1 < p.nsteps) store_tile((step + 1) & 1);
    __syncthreads();
    if (step + 2 < p.nsteps) load_tile();
  }

  const float inv = 1.0f / (sx[0] * sw[0]);
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
      int64_t pm = m0 + m_base + a * 32 + row;
      if (pm >= p.P) continue;
      int64_t yr = p.identity_out ? pm : out_row(g, pm);
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        int kc = n0 + n_base + b * 32 + li;
        if (kc < g.K) {
          float v = acc[a][b][e] * inv;
          if (bias) v += bias[kc];
          if (res) v += res[yr * g.K + kc];
          y[yr * g.K + kc] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
static int launch_h3(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                     const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  const wdno_conv_geom& g = p.g;
  int64_t tiles_m = cdiv64(p.P, BM);
  p.tiles_n = cdiv(g.K, BN);
  int64_t nt = tiles_m * p.tiles_n;
  if (nt > 0x7fffffff) return WDNO_EUNSUPPORTED;
  p.ntiles = (int)nt;
  size_t lds = (size_t)2 * 2 * (BM + BN) * HST * sizeof(_Float16);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_h3_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  conv_fwd_h3_kernel<BM, BN, WM, WN><<<p.ntiles, 256, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)wh,
                                                              (const _Float16*)wl, sx, sw, bias, residual, y, p);
  return WDNO_OK;
}

extern "C" int wdno_conv_fwd_f16x3(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                                   const float* bias, const float* residual, float* y, const wdno_conv_geom* g, wdno_stream_t s) {
  int rc = check_geom(g);
  if (rc) return rc;
  if (g->C & 7) return WDNO_EUNSUPPORTED;       // fp16 rows must be 16-byte multiples
  ConvP p;
  fill_params(p, g);
  p.nchunk = cdiv(p.R, HBK);
  p.nsteps = g->kd * g->kh * p.nchunk;
  hipStream_t st = as_stream(s);
  const int64_t P = p.P;
  const int K = g->K;
  auto blocks = [&](int bm, int bn) { return cdiv64(P, bm) * cdiv(K, bn); };
  if (K > 64) {
    if (blocks(128, 128) >= 512 || blocks(64, 128) < 2 * blocks(128, 128)) rc = launch_h3<128, 128, 2, 2>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
    else rc = launch_h3<64, 128, 1, 4>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
  } else {
    if (blocks(128, 64) >= 512 || P <= 128) rc = launch_h3<128, 64, 4, 1>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
    else rc = launch_h3<64, 64, 2, 2>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
  }
  if (rc) return rc;
  return wdno_check_launch();
}
