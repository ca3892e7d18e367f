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
#include "conv_common.h"

// ------------------------------------------------------------------------------------------------ forward
// Double-buffered LDS (one barrier per BK step); global loads for step s+2 are issued right after the barrier of step s
// and land while the 32..64 MFMAs (64 cycles each) of step s+1 run. Address arithmetic is incremental: no divisions in
// the steady state (tap row state is refreshed once per (dz,dy), the (dx, c) position of this thread's float4 column
// advances by BK per step).
template <int BM, int BN, int WM, int WN, bool TWO_LEVEL = false>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, const float* __restrict__ res,
                                                       float* __restrict__ y, ConvP p) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AROWS = BM / 32;   // A rows loaded per thread
  constexpr int BROWS = BN / 32;   // B rows loaded per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                 // [2][BM * LDS_STRIDE]
  float* Bs = smem + 2 * BM * LDS_STRIDE;           // [2][BN * LDS_STRIDE]

  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  const int lrow = tid >> 3;          // 0..31
  const int c4 = (tid & 7) * 4;       // reduction offset inside the BK chunk
  int a_d0[AROWS], a_h0[AROWS], a_w0[AROWS];
  int64_t a_nbase[AROWS];
  bool a_ok[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
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
  // per-tap row state
  int64_t row_off[AROWS];
  bool row_ok[AROWS];
  const float* b_ptr[BROWS];
  bool b_ok[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_ok[i] = (n0 + lrow + 32 * i) < g.K;

  // load cursor
  int l_tap = 0, l_chunk = 0, l_dz = 0, l_dy = 0, l_r = c4, l_dx = c4 / g.C, l_cc = c4 % g.C;
  const int dx0 = l_dx, cc0 = l_cc;
  auto refresh_tap = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int d = a_d0[i] + l_dz, h = a_h0[i] + l_dy;
      row_ok[i] = a_ok[i] && d >= 0 && d < g.D && h >= 0 && h < g.H;
      row_off[i] = (((a_nbase[i] + d) * g.H + h) * (int64_t)g.W + a_w0[i]) * g.C;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) b_ptr[i] = wp + ((int64_t)l_tap * g.K + (n0 + lrow + 32 * i)) * p.R;
  };
  refresh_tap();

  float4 areg[AROWS], breg[BROWS];
  auto load_tile = [&]() {   // loads the tile at the cursor, then advances the cursor
    const bool r_ok = l_r < p.R;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int w = a_w0[i] + l_dx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row_ok[i] && r_ok && w >= 0 && w < g.W) v = *reinterpret_cast<const float4*>(x + row_off[i] + l_r);
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_ok[i] && r_ok) v = *reinterpret_cast<const float4*>(b_ptr[i] + l_r);
      breg[i] = v;
    }
    // advance
    ++l_chunk;
    l_r += BK;
    l_cc += BK;
    while (l_cc >= g.C) { l_cc -= g.C; ++l_dx; }
    if (l_chunk == p.nchunk) {
      l_chunk = 0; l_r = c4; l_dx = dx0; l_cc = cc0;
      ++l_tap;
      if (++l_dy == g.kh) { l_dy = 0; ++l_dz; }
      refresh_tap();
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * (BM * LDS_STRIDE);
    float* Bb = Bs + buf * (BN * LDS_STRIDE);
#pragma unroll
    for (int i = 0; i < AROWS; ++i) *reinterpret_cast<float4*>(&Ab[(lrow + 32 * i) * LDS_STRIDE + c4]) = areg[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4*>(&Bb[(lrow + 32 * i) * LDS_STRIDE + c4]) = breg[i];
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

  // Two-level summation: `acc` collects FLUSH_STEPS steps (16 sequential MFMA accumulations each), then is added into `tot`.
  // A 7x7x7 stem is 490 steps = 7840 sequential roundings on one chain otherwise, and its error grows like the square root of
  // the chain length: with the flush the longest chains are 64 and ~120 long (measured: tools/error_trace.py).
  // Instantiated for long reductions only (TWO_LEVEL; dispatch in wdno_conv_fwd): the second accumulator set costs registers.
  constexpr int FLUSH_STEPS = 4;
  f32x16 tot[TWO_LEVEL ? TM : 1][TWO_LEVEL ? TN : 1];
  if (TWO_LEVEL) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[TWO_LEVEL ? a : 0][TWO_LEVEL ? b : 0][e] = 0.f;
  }

  load_tile();
  store_tile(0);
  __syncthreads();
  if (p.nsteps > 1) load_tile();
  for (int step = 0; step < p.nsteps; ++step) {
    const float* Ab = As + (step & 1) * (BM * LDS_STRIDE);
    const float* Bb = Bs + (step & 1) * (BN * LDS_STRIDE);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const float4*>(&Ab[(m_base + a * 32 + li) * LDS_STRIDE + kk * 8 + hh * 4]);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const float4*>(&Bb[(n_base + b * 32 + li) * LDS_STRIDE + kk * 8 + hh * 4]);
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
    if (TWO_LEVEL && (step & (FLUSH_STEPS - 1)) == FLUSH_STEPS - 1) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            tot[TWO_LEVEL ? a : 0][TWO_LEVEL ? b : 0][e] += acc[a][b][e];
            acc[a][b][e] = 0.f;
          }
    }
    if (step + 1 < p.nsteps) store_tile((step + 1) & 1);
    __syncthreads();
    if (step + 2 < p.nsteps) load_tile();
  }
  if (TWO_LEVEL) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] += tot[TWO_LEVEL ? a : 0][TWO_LEVEL ? b : 0][e];
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

template <int BM, int BN, int WM, int WN, bool TWO_LEVEL = false>
static int launch_fwd(const float* x, const float* wp, const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  const wdno_conv_geom& g = p.g;
  int64_t tiles_m = cdiv64(p.P, BM);
  p.tiles_n = cdiv(g.K, BN);
  int64_t nt = tiles_m * p.tiles_n;
  if (nt > 0x7fffffff) return WDNO_EUNSUPPORTED;
  p.ntiles = (int)nt;
  size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<BM, BN, WM, WN, TWO_LEVEL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  conv_fwd_kernel<BM, BN, WM, WN, TWO_LEVEL><<<p.ntiles, 256, lds, st>>>(x, wp, bias, residual, y, p);
  return WDNO_OK;
}

extern "C" int wdno_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, float* y,
                             const wdno_conv_geom* g, wdno_stream_t s) {
  int rc = check_geom(g);
  if (rc) return rc;
  ConvP p;
  fill_params(p, g);
  hipStream_t st = as_stream(s);
  // tile choice: the largest tile that still yields >= 2 workgroups per CU (512); small problems take smaller tiles
  const int64_t P = p.P;
  const int K = g->K;
  auto blocks = [&](int bm, int bn) { return cdiv64(P, bm) * cdiv(K, bn); };
  if (p.nsteps >= 64) {       // reductions of >= 2048 terms (the 7x7x7 stem: 15 092): two-level summation, see conv_fwd_kernel
    if (K > 64 && blocks(64, 128) >= 256) rc = launch_fwd<64, 128, 1, 4, true>(x, wp, bias, residual, y, p, st);
    else rc = launch_fwd<64, 64, 2, 2, true>(x, wp, bias, residual, y, p, st);
  } else if (K > 64) {
    if (blocks(128, 128) >= 512 || blocks(64, 128) < 2 * blocks(128, 128)) rc = launch_fwd<128, 128, 2, 2>(x, wp, bias, residual, y, p, st);
    else rc = launch_fwd<64, 128, 1, 4>(x, wp, bias, residual, y, p, st);
  } else {
    if (blocks(128, 64) >= 512 || P <= 128) rc = launch_fwd<128, 64, 4, 1>(x, wp, bias, residual, y, p, st);
    else rc = launch_fwd<64, 64, 2, 2>(x, wp, bias, residual, y, p, st);
  }
  if (rc) return rc;
  return wdno_check_launch();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dwp[tap][k][r] = sum over output pixels p of dy[p][k] * x[p shifted by tap][r]. M = output channels, N = a BN-wide
// slice of the contiguous (kw*C) run of one (dz,dy) tap row, reduction = pixels (32 per step, 2 per MFMA), split over
// blockIdx.y with a deterministic second-stage reduction. Pixel coordinates are decoded by 32 threads per step into a
// 3-deep LDS ring so that the steady state has no divisions in the load path; tiles are double-buffered (one barrier
// per step).
#define WG_BKP_MAX 32

struct PixInfo { int64_t yrow; int64_t xbase; int w0; int ok; };

struct WgradP {
  ConvP c;
  int tiles_k, tiles_r;
  int splits;
  int64_t pix_per_split;   // multiple of 32
};

template <int BM, int BN, int WM, int WN, int WG_BKP>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ ws, WgradP wpz) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AST = BM + 4, BST = BN + 4;
  constexpr int A_F4 = BM / 4, B_F4 = BN / 4;              // float4 per tile row
  constexpr int A_RPP = 256 / A_F4, B_RPP = 256 / B_F4;    // rows covered per pass of the 256 threads
  constexpr int A_PASSES = WG_BKP / A_RPP, B_PASSES = WG_BKP / B_RPP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                               // [2][WG_BKP * AST]
  float* Bs = smem + 2 * WG_BKP * AST;            // [2][WG_BKP * BST]
  PixInfo* pinfo = reinterpret_cast<PixInfo*>(smem + 2 * WG_BKP * (AST + BST));   // [3][WG_BKP]

  const ConvP& p = wpz.c;
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tile_r = b % wpz.tiles_r; b /= wpz.tiles_r;
  const int tap = b % (g.kd * g.kh);
  const int tile_k = b / (g.kd * g.kh);
  const int dz = tap / g.kh, dyy = tap - dz * g.kh;
  const int k0 = tile_k * BM, r0 = tile_r * BN;
  const int64_t pbeg = (int64_t)blockIdx.y * wpz.pix_per_split;
  int64_t pend = pbeg + wpz.pix_per_split;
  if (pend > p.P) pend = p.P;
  const int nsteps = pbeg < pend ? (int)((pend - pbeg + WG_BKP - 1) / WG_BKP) : 0;

  auto decode = [&](int step) {
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
      pinfo[(step % 3) * WG_BKP + tid] = pi;
    }
  };

  const int b_row = tid / B_F4, b_c4 = (tid % B_F4) * 4;
  const int r = r0 + b_c4;
  const bool r_ok = r < p.R;
  const int dx = r / g.C;
  const int a_row = tid / A_F4, a_c4 = (tid % A_F4) * 4;
  const int ka = k0 + a_c4;
  const bool ka_ok = ka < g.K;

  float4 areg[A_PASSES], breg[B_PASSES];
  auto load_tile = [&](int step) {
    const PixInfo* ps = pinfo + (step % 3) * WG_BKP;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const PixInfo pi = ps[a_row + i * A_RPP];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((pi.ok & 1) && ka_ok) v = *reinterpret_cast<const float4*>(dy + pi.yrow * g.K + ka);
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const PixInfo pi = ps[b_row + i * B_RPP];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      int w = pi.w0 + dx;
      if (pi.ok == 3 && r_ok && w >= 0 && w < g.W) v = *reinterpret_cast<const float4*>(x + pi.xbase + r);
      breg[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * (WG_BKP * AST);
    float* Bb = Bs + buf * (WG_BKP * BST);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) *reinterpret_cast<float4*>(&Ab[(a_row + i * A_RPP) * AST + a_c4]) = areg[i];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) *reinterpret_cast<float4*>(&Bb[(b_row + i * B_RPP) * BST + b_c4]) = breg[i];
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
    decode(0);
    if (nsteps > 1) decode(1);
    __syncthreads();
    load_tile(0);
    store_tile(0);
    if (nsteps > 2) decode(2);
    __syncthreads();
    if (nsteps > 1) load_tile(1);
    for (int step = 0; step < nsteps; ++step) {
      const float* Ab = As + (step & 1) * (WG_BKP * AST);
      const float* Bb = Bs + (step & 1) * (WG_BKP * BST);
#pragma unroll
      for (int pp = 0; pp < WG_BKP / 2; ++pp) {
        float af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = Ab[(pp * 2 + hh) * AST + m_base + a * 32 + li];
#pragma unroll
        for (int bb = 0; bb < TN; ++bb) bf[bb] = Bb[(pp * 2 + hh) * BST + n_base + bb * 32 + li];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int bb = 0; bb < TN; ++bb) acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[bb], acc[a][bb], 0, 0, 0);
      }
      if (step + 1 < nsteps) store_tile((step + 1) & 1);
      __syncthreads();
      if (step + 2 < nsteps) load_tile(step + 2);
      if (step + 3 < nsteps) decode(step + 3);
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

static int wgrad_bn(const wdno_conv_geom* g) {   // width of the reduction-run slice per block
  int R = g->kw * g->C;
  if (g->K > 64) return 128;
  int waste128 = cdiv(R, 128) * 128 - R, waste256 = cdiv(R, 256) * 256 - R;
  return waste256 <= waste128 ? 256 : 128;
}
static void wgrad_plan(WgradP& w, const wdno_conv_geom* g) {
  fill_params(w.c, g);
  const int BM = g->K > 64 ? 128 : 64;
  const int BN = wgrad_bn(g);
  w.tiles_k = cdiv(g->K, BM);
  w.tiles_r = cdiv(w.c.R, BN);
  int64_t tiles = (int64_t)w.tiles_k * w.tiles_r * g->kd * g->kh;
  int64_t want = cdiv64(768, tiles);               // ~3 workgroups per CU in total
  int64_t max_splits = cdiv64(w.c.P, 8 * WG_BKP_MAX);  // at least 8 steps per block
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  if (want > 4096) want = 4096;
  int64_t pps = cdiv64(cdiv64(w.c.P, want), WG_BKP_MAX) * WG_BKP_MAX;
  w.pix_per_split = pps;
  w.splits = (int)cdiv64(w.c.P, pps);
}
extern "C" size_t wdno_conv_wgrad_ws_bytes(const wdno_conv_geom* g) {
  if (check_geom(g) != WDNO_OK) return 0;
  WgradP w;
  wgrad_plan(w, g);
  return (size_t)w.splits * g->kd * g->kh * (size_t)g->K * w.c.R * sizeof(float);
}
template <int BM, int BN, int WM, int WN, int WG_BKP>
static void launch_wgrad(const float* x, const float* dy, float* wsf, const WgradP& w, dim3 grid, hipStream_t st) {
  size_t lds = (size_t)2 * WG_BKP * (BM + 4 + BN + 4) * sizeof(float) + 3 * WG_BKP * sizeof(PixInfo);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<BM, BN, WM, WN, WG_BKP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  conv_wgrad_kernel<BM, BN, WM, WN, WG_BKP><<<grid, 256, lds, st>>>(x, dy, wsf, w);
}
extern "C" int wdno_conv_wgrad(const float* x, const float* dy, float* dwp, void* ws, size_t ws_bytes,
                               const wdno_conv_geom* g, wdno_stream_t s) {
  int rc = check_geom(g);
  if (rc) return rc;
  WgradP w;
  wgrad_plan(w, g);
  size_t need = (size_t)w.splits * g->kd * g->kh * (size_t)g->K * w.c.R * sizeof(float);
  if (ws_bytes < need) return WDNO_EWORKSPACE;
  if (w.splits > 65535) return WDNO_EUNSUPPORTED;
  dim3 grid((unsigned)(w.tiles_k * g->kd * g->kh * w.tiles_r), (unsigned)w.splits);
  float* wsf = w.splits == 1 ? dwp : (float*)ws;
  hipStream_t st = as_stream(s);
  if (g->K > 64) launch_wgrad<128, 128, 2, 2, 32>(x, dy, wsf, w, grid, st);
  else if (wgrad_bn(g) == 256) launch_wgrad<64, 256, 1, 4, 16>(x, dy, wsf, w, grid, st);
  else launch_wgrad<64, 128, 1, 4, 32>(x, dy, wsf, w, grid, st);
  if (w.splits > 1) {
    int64_t n = (int64_t)g->kd * g->kh * g->K * w.c.R;
    wgrad_reduce_kernel<<<stream_grid(n, 256), 256, 0, st>>>((const float*)ws, dwp, n, w.splits);
  }
  return wdno_check_launch();
}
