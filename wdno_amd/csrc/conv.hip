// conv.hip -- implicit-GEMM convolution (forward / data-gradient / weight-gradient) on gfx950 matrix cores.
//
// Precision: v_mfma_f32_32x32x2_f32 -- exact fp32 products with fp32 accumulation (bitwise an fmaf chain), so the
// fp32 parity bar of the reference (< 1e-5 rel-L2) holds. Peak for this instruction is 157 TFLOP/s.
//
// Layout: activations channels-last [N, D, H, W, C]; weights packed wp[kd][kh][K][kw*C]. For one output pixel and one
// (dz, dy) tap row, the kw*C reduction run is contiguous in BOTH operands, so global loads are 16-byte and the LDS
// tiles are [rows][32 + 4] with the reduction index fastest (conflict-free ds_read_b128 for 4 MFMAs each).
//
//   forward  : M = output pixels (128 / block), N = output channels (64 or 128 / block), reduction = kd*kh*(kw*C)
//   wgrad    : M = output channels, N = (kw*C) run of one (dz,dy) tap row, reduction = output pixels (split-K over
//              blocks, deterministic two-stage reduction through a workspace)
//   dgrad    : the forward kernel applied to dy with flipped / transposed weights (packed on the host side);
//              stride-2 transposed convolutions are run as 4 parity classes through the output placement terms.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define LDS_STRIDE 36   // BK + 4 floats: 144-byte rows -> 16 consecutive rows hit 16 distinct 16-B slots

struct ConvP {
  wdno_conv_geom g;
  int R;          // kw * C
  int nchunk;     // ceil(R / BK)
  int nsteps;     // kd*kh*nchunk
  int64_t P;      // N*OD*OH*OW
  int identity_out;
  int tiles_n;
  int ntiles;
};

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // bijective remap so that each of the 8 XCDs (block b runs on XCD b % 8) gets a contiguous range of tiles
  int q = nwg >> 3, r = nwg & 7;
  int xcd = bid & 7, idx = bid >> 3;
  int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

__device__ __forceinline__ int64_t out_row(const wdno_conv_geom& g, int64_t p) {
  int ow = (int)(p % g.OW); int64_t t = p / g.OW;
  int oh = (int)(t % g.OH); t /= g.OH;
  int od = (int)(t % g.OD);
  int64_t n = t / g.OD;
  return ((n * g.YD + (od * g.osd + g.ood)) * g.YH + (oh * g.osh + g.ooh)) * g.YW + (ow * g.osw + g.oow);
}

// ------------------------------------------------------------------------------------------------ forward
template <int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, const float* __restrict__ res,
                                                       float* __restrict__ y, ConvP p) {
  constexpr int BM = 128;
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int BROWS = BN / 32;   // B rows loaded per thread
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_STRIDE];

  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-thread load assignment: 4 pixel rows (A) and BROWS weight rows (B), one float4 column each
  const int lrow = tid >> 3;          // 0..31
  const int c4 = (tid & 7) * 4;       // reduction offset inside the BK chunk
  int a_d0[4], a_h0[4], a_w0[4];
  int64_t a_nbase[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t pm = m0 + lrow + 32 * i;
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

  float4 areg[4], breg[BROWS];
  auto load_tile = [&](int step) {
    int tap = step / p.nchunk;
    int chunk = step - tap * p.nchunk;
    int dz = tap / g.kh, dy = tap - dz * g.kh;
    int r = chunk * BK + c4;
    bool r_ok = r < p.R;
    int dx = r / g.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int d = a_d0[i] + dz, h = a_h0[i] + dy, w = a_w0[i] + dx;
      bool ok = a_ok[i] && r_ok && d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        int64_t pix = ((a_nbase[i] + d) * g.H + h) * (int64_t)g.W + a_w0[i];
        v = *reinterpret_cast<const float4*>(x + pix * g.C + r);
      }
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      int k = n0 + lrow + 32 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < g.K && r_ok) v = *reinterpret_cast<const float4*>(wp + ((int64_t)tap * g.K + k) * p.R + r);
      breg[i] = v;
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

  load_tile(0);
  for (int step = 0; step < p.nsteps; ++step) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[(lrow + 32 * i) * LDS_STRIDE + c4]) = areg[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4*>(&Bs[(lrow + 32 * i) * LDS_STRIDE + c4]) = breg[i];
    __syncthreads();
    if (step + 1 < p.nsteps) load_tile(step + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const float4*>(&As[(m_base + a * 32 + li) * LDS_STRIDE + kk * 8 + hh * 4]);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const float4*>(&Bs[(n_base + b * 32 + li) * LDS_STRIDE + kk * 8 + hh * 4]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // ---- epilogue: acc[reg] <-> (row = (reg&3) + 8*(reg>>2) + 4*hh, col = li)
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
          float v = acc[a][b][e];
          if (bias) v += bias[kc];
          if (res) v += res[yr * g.K + kc];
          y[yr * g.K + kc] = v;
        }
      }
    }
  }
}

static int check_geom(const wdno_conv_geom* g) {
  if (!g) return WDNO_EINVAL;
  if (g->N <= 0 || g->D <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->K <= 0) return WDNO_EINVAL;
  if (g->OD <= 0 || g->OH <= 0 || g->OW <= 0 || g->kd <= 0 || g->kh <= 0 || g->kw <= 0) return WDNO_EINVAL;
  if (g->sd <= 0 || g->sh <= 0 || g->sw <= 0 || g->osd <= 0 || g->osh <= 0 || g->osw <= 0) return WDNO_EINVAL;
  if ((g->C & 3) || (g->K & 3)) return WDNO_EUNSUPPORTED;   // rows must be 16-byte multiples (callers pad)
  if ((g->OD - 1) * g->osd + g->ood >= g->YD || (g->OH - 1) * g->osh + g->ooh >= g->YH || (g->OW - 1) * g->osw + g->oow >= g->YW)
    return WDNO_EINVAL;
  return WDNO_OK;
}
static void fill_params(ConvP& p, const wdno_conv_geom* g) {
  p.g = *g;
  p.R = g->kw * g->C;
  p.nchunk = cdiv(p.R, BK);
  p.nsteps = g->kd * g->kh * p.nchunk;
  p.P = (int64_t)g->N * g->OD * g->OH * g->OW;
  p.identity_out = (g->YD == g->OD && g->YH == g->OH && g->YW == g->OW && g->osd == 1 && g->osh == 1 && g->osw == 1 &&
                    g->ood == 0 && g->ooh == 0 && g->oow == 0) ? 1 : 0;
}

extern "C" int wdno_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, float* y,
                             const wdno_conv_geom* g, wdno_stream_t s) {
  int rc = check_geom(g);
  if (rc) return rc;
  ConvP p;
  fill_params(p, g);
  int64_t tiles_m = cdiv64(p.P, 128);
  if (g->K > 64) {
    p.tiles_n = cdiv(g->K, 128);
    int64_t nt = tiles_m * p.tiles_n;
    if (nt > 0x7fffffff) return WDNO_EUNSUPPORTED;
    p.ntiles = (int)nt;
    conv_fwd_kernel<128, 2, 2><<<p.ntiles, 256, 0, as_stream(s)>>>(x, wp, bias, residual, y, p);
  } else {
    p.tiles_n = 1;
    if (tiles_m > 0x7fffffff) return WDNO_EUNSUPPORTED;
    p.ntiles = (int)tiles_m;
    conv_fwd_kernel<64, 4, 1><<<p.ntiles, 256, 0, as_stream(s)>>>(x, wp, bias, residual, y, p);
  }
  return wdno_check_launch();
}

// ------------------------------------------------------------------------------------------------ weight gradient
#define WG_BKP 16          // output pixels per step
#define WG_BN 128          // reduction-run columns per block

struct PixInfo { int64_t yrow; int64_t xbase; int w0; int ok; };

struct WgradP {
  ConvP c;
  int tiles_k, tiles_r;
  int splits;
  int64_t pix_per_split;   // multiple of WG_BKP
};

template <int BM, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ ws, WgradP wpz) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = WG_BN / (WN * 32);
  constexpr int AST = BM + 4, BST = WG_BN + 4;
  constexpr int A_F4_PER_ROW = BM / 4;
  constexpr int A_ROWS_PER_PASS = 256 / A_F4_PER_ROW;     // 8 (BM=128) or 16 (BM=64)
  constexpr int A_PASSES = WG_BKP / A_ROWS_PER_PASS;      // 2 or 1
  __shared__ __attribute__((aligned(16))) float As[WG_BKP * AST];
  __shared__ __attribute__((aligned(16))) float Bs[WG_BKP * BST];
  __shared__ PixInfo pinfo[2][WG_BKP];

  const ConvP& p = wpz.c;
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  // blockIdx.x -> (tile_k, tap, tile_r) ; blockIdx.y -> split
  int b = blockIdx.x;
  const int tile_r = b % wpz.tiles_r; b /= wpz.tiles_r;
  const int tap = b % (g.kd * g.kh);
  const int tile_k = b / (g.kd * g.kh);
  const int dz = tap / g.kh, dyy = tap - dz * g.kh;
  const int k0 = tile_k * BM, r0 = tile_r * WG_BN;
  const int64_t pbeg = (int64_t)blockIdx.y * wpz.pix_per_split;
  int64_t pend = pbeg + wpz.pix_per_split;
  if (pend > p.P) pend = p.P;
  const int nsteps = pbeg < pend ? (int)((pend - pbeg + WG_BKP - 1) / WG_BKP) : 0;

  auto decode = [&](int step, int slot) {
    if (tid < WG_BKP) {
      int64_t pm = pbeg + (int64_t)step * WG_BKP + tid;
      PixInfo pi;
      pi.ok = 0; pi.yrow = 0; pi.xbase = 0; pi.w0 = 0;
      if (pm < pend) {
        int64_t q = pm;
        int ow = (int)(q % g.OW); q /= g.OW;
        int oh = (int)(q % g.OH); q /= g.OH;
        int od = (int)(q % g.OD);
        int64_t n = q / g.OD;
        pi.yrow = ((n * g.YD + (od * g.osd + g.ood)) * g.YH + (oh * g.osh + g.ooh)) * g.YW + (ow * g.osw + g.oow);
        int d = od * g.sd - g.pd + dz, h = oh * g.sh - g.ph + dyy;
        pi.w0 = ow * g.sw - g.pw;
        pi.xbase = (((n * g.D + d) * g.H + h) * (int64_t)g.W + pi.w0) * g.C;
        pi.ok = (d >= 0 && d < g.D && h >= 0 && h < g.H) ? 3 : 1;   // bit0: pixel in range, bit1: input row in range
      }
      pinfo[slot][tid] = pi;
    }
  };

  // B (shifted x) assignment: 2 rows per thread, fixed float4 column
  const int b_row = tid >> 5, b_c4 = (tid & 31) * 4;
  const int r = r0 + b_c4;
  const bool r_ok = r < p.R;
  const int dx = r / g.C;
  // A (dy) assignment
  const int a_row = tid / A_F4_PER_ROW, a_c4 = (tid % A_F4_PER_ROW) * 4;
  const int ka = k0 + a_c4;
  const bool ka_ok = ka < g.K;

  float4 areg[A_PASSES], breg[2];
  auto load_tile = [&](int slot) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const PixInfo pi = pinfo[slot][a_row + i * A_ROWS_PER_PASS];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((pi.ok & 1) && ka_ok) v = *reinterpret_cast<const float4*>(dy + pi.yrow * g.K + ka);
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const PixInfo pi = pinfo[slot][b_row + i * 8];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      int w = pi.w0 + dx;
      if (pi.ok == 3 && r_ok && w >= 0 && w < g.W) v = *reinterpret_cast<const float4*>(x + pi.xbase + r);
      breg[i] = v;
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
    for (int bb = 0; bb < TN; ++bb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][bb][e] = 0.f;

  if (nsteps > 0) {
    decode(0, 0);
    __syncthreads();
    load_tile(0);
    if (nsteps > 1) decode(1, 1);
    for (int step = 0; step < nsteps; ++step) {
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) *reinterpret_cast<float4*>(&As[(a_row + i * A_ROWS_PER_PASS) * AST + a_c4]) = areg[i];
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&Bs[(b_row + i * 8) * BST + b_c4]) = breg[i];
      __syncthreads();
      if (step + 1 < nsteps) load_tile((step + 1) & 1);
      if (step + 2 < nsteps) decode(step + 2, step & 1);
#pragma unroll
      for (int pp = 0; pp < WG_BKP / 2; ++pp) {
        float af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = As[(pp * 2 + hh) * AST + m_base + a * 32 + li];
#pragma unroll
        for (int bb = 0; bb < TN; ++bb) bf[bb] = Bs[(pp * 2 + hh) * BST + n_base + bb * 32 + li];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int bb = 0; bb < TN; ++bb) acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[bb], acc[a][bb], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // partial tile -> ws[split][tap][K][R]
  float* out = ws + ((int64_t)blockIdx.y * (g.kd * g.kh) + tap) * (int64_t)g.K * p.R;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int kk = k0 + m_base + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (kk >= g.K) continue;
#pragma unroll
      for (int bb = 0; bb < TN; ++bb) {
        int rr = r0 + n_base + bb * 32 + li;
        if (rr < p.R) out[(int64_t)kk * p.R + rr] = acc[a][bb][e];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int splits) {
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += ws[(int64_t)s * n + i];
    out[i] = acc;
  }
}

static void wgrad_plan(WgradP& w, const wdno_conv_geom* g) {
  fill_params(w.c, g);
  const int BM = g->K > 64 ? 128 : 64;
  w.tiles_k = cdiv(g->K, BM);
  w.tiles_r = cdiv(w.c.R, WG_BN);
  int64_t tiles = (int64_t)w.tiles_k * w.tiles_r * g->kd * g->kh;
  int64_t want = cdiv64(1024, tiles);              // ~4 blocks per CU in total
  int64_t max_splits = cdiv64(w.c.P, 8 * WG_BKP);  // at least 8 steps per block
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  if (want > 4096) want = 4096;
  int64_t pps = cdiv64(cdiv64(w.c.P, want), WG_BKP) * WG_BKP;
  w.pix_per_split = pps;
  w.splits = (int)cdiv64(w.c.P, pps);
}
extern "C" size_t wdno_conv_wgrad_ws_bytes(const wdno_conv_geom* g) {
  if (check_geom(g) != WDNO_OK) return 0;
  WgradP w;
  wgrad_plan(w, g);
  return (size_t)w.splits * g->kd * g->kh * (size_t)g->K * w.c.R * sizeof(float);
}
extern "C" int wdno_conv_wgrad(const float* x, const float* dy, float* dwp, void* ws, size_t ws_bytes,
                               const wdno_conv_geom* g, wdno_stream_t s) {
  int rc = check_geom(g);
  if (rc) return rc;
  WgradP w;
  wgrad_plan(w, g);
  size_t need = (size_t)w.splits * g->kd * g->kh * (size_t)g->K * w.c.R * sizeof(float);
  if (ws_bytes < need) return WDNO_EWORKSPACE;
  dim3 grid((unsigned)(w.tiles_k * g->kd * g->kh * w.tiles_r), (unsigned)w.splits);
  if (w.splits > 65535) return WDNO_EUNSUPPORTED;
  float* wsf = w.splits == 1 ? dwp : (float*)ws;
  if (g->K > 64) conv_wgrad_kernel<128, 2, 2><<<grid, 256, 0, as_stream(s)>>>(x, dy, wsf, w);
  else conv_wgrad_kernel<64, 1, 4><<<grid, 256, 0, as_stream(s)>>>(x, dy, wsf, w);
  if (w.splits > 1) {
    int64_t n = (int64_t)g->kd * g->kh * g->K * w.c.R;
    wgrad_reduce_kernel<<<stream_grid(n, 256), 256, 0, as_stream(s)>>>((const float*)ws, dwp, n, w.splits);
  }
  return wdno_check_launch();
}
