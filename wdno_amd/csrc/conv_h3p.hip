// conv_h3p.hip -- "plane-resident" variant of the tap-resident convolution (conv_h3t.hip) for the 3 x 3 (x kd) ResnetBlock convolutions and
// their data gradients: stride 1, output grid == input grid, kh == kw == 3, image rows of at most 64 pixels.
//
// conv_h3t.hip keeps one (dz, dy) tap row of a 32-channel block in LDS and uses it for the three dx: the x rows of a tile still travel
// L2 -> LDS once per dy, although the BM consecutive pixels of a tile and their dy = -1 / +1 neighbours are the SAME pixels but for 2 W of them.
// Here a stage holds, for one dz and one block of 16 input channels,
//     A: the BM + 2 W + 2 consecutive source pixels the tile's BM output pixels touch over all nine (dy, dx)   (rows of 32 B, hi / lo planes)
//     B: the 9 x BN weight rows of that (dz, channel block)
// and is used for nine reduction sub-steps (one 16-deep MFMA step each): sub-step (dy, dx) reads the A rows shifted by dy W + dx.
// Per 256 x 64 tile of a 64 -> 64 layer on a 40-pixel row: 12 stages of 58 KB = 0.70 MB of DMA instead of 18 stages of 57.5 KB = 1.04 MB, with
// the same matrix instructions and the same fragment reads -- the stem kernel (7-wide taps on 16-channel blocks, 0.54 KB per MFMA) is the
// measured precedent: 1.42 PFLOP/s of matrix work against 0.99 for the 3-wide layers (0.80 KB per MFMA) at the same tile shape.
//   * Validity. dz: by the producers, on the pixel a row is the centre (dy = dx = 1) tap of; an invalid row is fetched out of range = zeros.
//     (dy, dx): by the compute waves once per tile -- a lane whose pixel has no neighbour in that direction reads the stage's zero row
//     (rows ZB .. ZB + 15, rewritten with zeros by every DMA pass) instead. A row that serves pixels of two planes / samples is dz-tested for
//     one of them only; for the other it is a (dy, dx)-invalid neighbour and never read.
//   * LDS: the A plane is sized for 64-pixel rows (BM + 130 rows -> 13 KB at BM = 256), 18 KB of B per plane at BN = 64: two stages of 62 KB.
//     The producers fetch only the pieces the actual row length needs, plus the zero piece.
//   * Nine sub-steps per stage: the fragment set a stage starts on alternates, stages run in pairs (an odd stage count gets one all-zero
//     stage), as for the 7-wide stem in conv_h3t.hip. Everything else -- operand formats, weight pack, scheduling of fragment reads behind
//     the matrix instructions, epilogue -- is conv_h3t.hip's; the fp32 accumulation order is (dz, channel block, dy, dx).
#include "conv_common.h"
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
#define P_OOB 0x7ffffff0
#define P_WMAX 64

template <bool LP>
__device__ __forceinline__ f32x16 p_mfma(half8 a, half8 b, f32x16 c) {
  if constexpr (LP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int4v p_rsrc(const void* ptr, unsigned bytes) {
  uint64_t a = reinterpret_cast<uint64_t>(ptr);
  int4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void p_piece(int4v rsrc, int off, unsigned lds_dst) {
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(off), "s"(lds_dst), "s"(rsrc) : "memory");
}

template <int BM, int BN>
struct PlaneShape {
  static constexpr int CB = 16, ROWB = 32, RPP = 32, ZR = 16, NT = 9;
  static constexpr int AROWS_MAX = BM + 2 * P_WMAX + 2;
  static constexpr int ZB = (AROWS_MAX + ZR - 1) / ZR * ZR;        // first zero row
  static constexpr int ZPIECE = ZB / RPP;                          // the piece that holds the zero rows
  static constexpr int APIECES = (ZB + ZR + RPP - 1) / RPP;
  static constexpr int A_PLANE = APIECES * 1024;
  static constexpr int BPIECES = (NT * BN + RPP - 1) / RPP;
  static constexpr int B_PLANE = BPIECES * 1024;
};

template <int BM, int BN, int WM, int WN, bool LP>
__global__ __launch_bounds__(512) void conv_fwd_h3p_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                            const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                            const float* __restrict__ sx, const float* __restrict__ sw,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ y, ConvP p, unsigned x_bytes, unsigned w_bytes) {
  using S = PlaneShape<BM, BN>;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int NPL = LP ? 1 : 2;
  constexpr int CB = S::CB, ROWB = S::ROWB, RPP = S::RPP, NT = S::NT;
  constexpr int A_PLANE = S::A_PLANE, B_PLANE = S::B_PLANE;
  constexpr int A_LO = A_PLANE, B_HI = NPL * A_PLANE, B_LO = B_HI + B_PLANE;
  constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  constexpr int PA = (S::APIECES + 3) / 4, PB = (S::BPIECES + 3) / 4;      // pieces per producer wave and plane
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ncb = g.C / CB;                                        // whole 16-channel blocks (wdno_conv_h3p_takes)
  const bool padded = (g.kd * ncb) & 1;                            // stages run in pairs: an odd count gets one all-zero stage at the end
  const int nstages = g.kd * ncb + (padded ? 1 : 0);
  const int halo = g.W + 1;                                        // A row r = flat pixel tile_p0 + r - halo (of the dz = pd plane)
  const int arows = BM + 2 * halo;

  if (wave >= WM * WN) {
    // ================================================================== producer waves
    const int pq = wave - WM * WN;
    int4v rxh = p_rsrc(xh, x_bytes), rxl = p_rsrc(xl, x_bytes), rwh = p_rsrc(wh, w_bytes), rwl = p_rsrc(wl, w_bytes);
    asm volatile("s_nop 4" : "+s"(rxh), "+s"(rxl), "+s"(rwh), "+s"(rwl));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    // lane -> (row of the piece, logical 16-byte chunk): the XOR swizzle of the fragment reads applied on the source side
    const int prow = lane >> 1;
    const int c8 = ((lane & 1) ^ ((lane >> 4) & 1)) * 8;
    const int na = (arows + RPP - 1) / RPP;                        // data pieces of an A plane; piece S::ZPIECE is fetched too (zero rows)
    int a_off[PA], b_off[PB];
    unsigned a_mask[PA];
    bool b_ok[PB];
    auto tap_bits = [](int c0, int k, int n) -> unsigned {
      int lo = c0 < 0 ? -c0 : 0, hi = n - c0 < k ? n - c0 : k;
      return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    auto setup_tile = [&](int t) {
      const int tile = xcd_swizzle((int)blockIdx.x + t * (int)gridDim.x, p.ntiles);
      const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
      const int64_t p0 = (int64_t)tile_m * BM;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int r = RPP * (pq + 4 * i) + prow;
        const int64_t pr = p0 + r - halo;                          // output pixel this row is the centre tap of
        const bool live = r < arows && pr >= 0 && pr < p.P;
        int q = live ? (int)pr : 0;
        q /= g.OW * g.OH;
        const int od = q % g.OD;
        a_mask[i] = live ? tap_bits(od - g.pd, g.kd, g.D) : 0u;
        a_off[i] = (((int)pr - g.pd * g.H * g.W) * g.C + c8) * 2;  // dz = 0, channel block 0; used only when valid
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = RPP * (pq + 4 * i) + prow;                 // (tap, k) = (row / BN, row % BN)
        const int tap = row / BN, k = tile_n * BN + (row - tap * BN);
        const int dy = tap / 3, dx = tap - dy * 3;
        b_ok[i] = k < g.K && tap < NT;
        b_off[i] = ((dy * g.K + k) * p.R + dx * g.C + c8) * 2;
      }
    };
    int c_tile = 0, s_dz = 0, s_cb = 0;                            // issue cursor
    int x_uni = 0, w_uni = 0;
    if (my_tiles > 0) setup_tile(0);
    auto issue_stage = [&](int buf) {
      const bool pad = s_dz == g.kd;                               // the all-zero stage
      const unsigned need = pad ? 0x10000u : (1u << s_dz);
      const unsigned dst = lds0 + buf * STAGE + pq * 1024;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int piece = pq + 4 * i;
        if (piece >= S::APIECES) continue;                         // compile-time for all but the last i
        if (piece >= na && piece != S::ZPIECE) continue;           // rows this row length does not have (wave-uniform)
        const int off = ((a_mask[i] & need) == need) ? a_off[i] + x_uni : P_OOB;
        p_piece(rxh, off, dst + i * 4096);
        if (!LP) p_piece(rxl, off, dst + A_LO + i * 4096);
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        if (pq + 4 * i >= S::BPIECES) continue;
        const int off = b_ok[i] && !pad ? b_off[i] + w_uni : P_OOB;
        p_piece(rwh, off, dst + B_HI + i * 4096);
        if (!LP) p_piece(rwl, off, dst + B_LO + i * 4096);
      }
      x_uni += ROWB; w_uni += ROWB;
      if (pad || ++s_cb == ncb) {
        s_cb = 0;
        if (!pad) ++s_dz;
        if (s_dz == g.kd && (pad || !padded)) {          // tile finished: move the cursor to the next one
          s_dz = 0;
          if (++c_tile < my_tiles) setup_tile(c_tile);
        }
        x_uni = s_dz * g.H * g.W * g.C * 2;
        w_uni = s_dz * g.kh * g.K * p.R * 2;
      }
    };
    const int total = my_tiles * nstages;
    if (total > 0) issue_stage(0);
    int buf = 0;
    for (int gs = 0; gs < total; ++gs) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : : : "memory");
      buf ^= 1;
      if (gs + 1 < total) issue_stage(buf);
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    return;
  }

  // ================================================================== compute waves
  float am = 0.f;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, hh = lane >> 5;
  const int m_base = wm * (TM * 32), n_base = wn * (TN * 32);
  const int b_row = n_base + li;
  auto swz = [](int row) { return (row >> 3) & 1; };                            // XOR applied to the chunk index of a row
  const int b_rd = B_HI + b_row * ROWB + ((hh ^ swz(b_row)) * 16);
  int a_rd[TM][NT];                                                             // per tile: A row address of (a, tap), or the zero row
  half8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];                         // [set][tile]
  auto read_frags = [&](int set, int boff, int tap) {
    const char* st = smem + boff;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      fah[set][a] = *reinterpret_cast<const half8*>(st + a_rd[a][tap]);
      if (!LP) fal[set][a] = *reinterpret_cast<const half8*>(st + A_LO + a_rd[a][tap]);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      fbh[set][b] = *reinterpret_cast<const half8*>(st + tap * (BN * ROWB) + b * (32 * ROWB) + b_rd);
      if (!LP) fbl[set][b] = *reinterpret_cast<const half8*>(st + B_PLANE + tap * (BN * ROWB) + b * (32 * ROWB) + b_rd);
    }
  };
  f32x16 acc[TM][TN];
  auto mfma_set = [&](int set) {
    if constexpr (!LP) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = p_mfma<false>(fbh[set][b], fal[set][a], acc[a][b]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = p_mfma<false>(fbl[set][b], fah[set][a], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = p_mfma<LP>(fbh[set][b], fah[set][a], acc[a][b]);
  };
  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sw[0]);
  int boff = 0;                                                   // byte offset of the stage being read
  for (int t = 0; t < my_tiles; ++t) {
    const int tile = xcd_swizzle((int)blockIdx.x + t * (int)gridDim.x, p.ntiles);
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    int row0 = m_base + li;
    asm volatile("" : "+v"(row0));                                // (recomputed per tile rather than 2 x 18 values held across the stage loop)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      int q = (int)m0 + row0 + a * 32;                            // P < 2^31 (launch_h3p)
      const int ow = q % g.OW; q /= g.OW;
      const int oh = q % g.OH;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int row = row0 + a * 32 + dy * g.W + dx;
          const bool ok = (unsigned)(ow - 1 + dx) < (unsigned)g.W && (unsigned)(oh - 1 + dy) < (unsigned)g.H;
          // the stand-in zero row keeps the bank slot of the real one (same row % 16, same swizzled chunk), see conv_h3t.hip
          const int zrow = S::ZB + (row & (S::ZR - 1));
          a_rd[a][dy * 3 + dx] = (ok ? row : zrow) * ROWB + ((hh ^ swz(row)) * 16);
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) asm volatile("v_mov_b32 %0, 0" : "=v"(acc[a][b][e]));
    lgkm0_barrier();
    read_frags(0, boff, 0);
    constexpr int NRD = NPL * (TM + TN), NMF = (LP ? 1 : 3) * TM * TN;
    constexpr int RPM = (NRD + NMF - 1) / NMF;                     // reads behind each MFMA until they are used up
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i * RPM < NRD) __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
      }
    };
    // a stage: sub-step `sub` = tap (dy, dx) works on fragment set (PAR + sub) & 1 while the reads of sub + 1 -- at the last sub-step, behind
    // the barrier, the first reads of the next stage -- go to the other set; nine sub-steps: the starting set alternates from stage to stage
    auto stage_body = [&](auto PARC, auto LASTC) {
      constexpr int PAR = decltype(PARC)::value;
      constexpr bool LAST = decltype(LASTC)::value;
#pragma unroll
      for (int sub = 0; sub < NT; ++sub) {
        if (sub + 1 < NT) {
          read_frags((PAR + sub + 1) & 1, boff, sub + 1);
        } else if (!LAST) {
          boff = STAGE - boff;
          lgkm0_barrier();
          read_frags((PAR + NT) & 1, boff, 0);
        }
        mfma_set((PAR + sub) & 1);
        if (sub + 1 < NT || !LAST) interleave();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    for (int s = 0; s + 2 < nstages; s += 2) { stage_body(P0{}, std::false_type{}); stage_body(P1{}, std::false_type{}); }
    stage_body(P0{}, std::false_type{});                           // nstages is even
    stage_body(P1{}, std::true_type{});
    boff = STAGE - boff;
    // epilogue: as conv_h3t.hip (accumulator tile is [channel][pixel]: a lane owns one pixel and runs of four channels)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int64_t pm = m0 + m_base + a * 32 + li;
      if (pm >= p.P) continue;
      float* yrow = y + pm * g.K;
      const float* rrow = res ? res + pm * g.K : nullptr;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int kc = n0 + n_base + b * 32 + 8 * e4 + 4 * hh;
          if (kc < g.K) {
            float4 v = make_float4(acc[a][b][4 * e4] * inv, acc[a][b][4 * e4 + 1] * inv, acc[a][b][4 * e4 + 2] * inv, acc[a][b][4 * e4 + 3] * inv);
            if (bias) { const float4 tb = *reinterpret_cast<const float4*>(bias + kc); v.x += tb.x; v.y += tb.y; v.z += tb.z; v.w += tb.w; }
            if (rrow) { const float4 tr = *reinterpret_cast<const float4*>(rrow + kc); v.x += tr.x; v.y += tr.y; v.z += tr.z; v.w += tr.w; }
            *reinterpret_cast<float4*>(yrow + kc) = v;
            am = amax4(am, v);
          }
        }
      }
    }
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * (WM * WN) + wave);
}

static int p_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

template <int BM, int BN, int WM, int WN, bool LP>
static int launch_h3p(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                      const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  using S = PlaneShape<BM, BN>;
  const wdno_conv_geom& g = p.g;
  int64_t tiles_m = cdiv64(p.P, BM);
  p.tiles_n = cdiv(g.K, BN);
  int64_t nt = tiles_m * p.tiles_n;
  if (nt > 0x7fffffff) return WDNO_EUNSUPPORTED;
  p.ntiles = (int)nt;
  p.tsplit = 1;
  constexpr size_t lds = (size_t)2 * (LP ? 1 : 2) * (S::A_PLANE + S::B_PLANE);
  static_assert(lds <= 160 * 1024, "two stages must fit the LDS");
  const int64_t x_elems = (int64_t)g.N * g.D * g.H * g.W * g.C;
  const int64_t w_elems = (int64_t)g.kd * g.kh * g.K * p.R;
  if (x_elems * 2 >= P_OOB || w_elems * 2 >= P_OOB || p.P >= 0x7fffffff - BM - 2 * P_WMAX - 2) return WDNO_EUNSUPPORTED;
  int grid = p_num_cus() & ~7;
  if (grid < 8) grid = 8;
  if (p.ntiles < grid) grid = p.ntiles;
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute((const void*)conv_fwd_h3p_kernel<BM, BN, WM, WN, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  conv_fwd_h3p_kernel<BM, BN, WM, WN, LP><<<grid, 512, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)wh, (const _Float16*)wl,
                                                               sx, sw, bias, residual, y, p, (unsigned)(x_elems * 2), (unsigned)(w_elems * 2));
  return WDNO_OK;
}

// Geometries the plane-resident kernel takes: stride 1, output grid == input grid, 3 x 3 taps in (H, W) with padding 1 (any kd <= 8), whole
// 16-channel blocks, image rows of at most 64 pixels (the A plane is sized for that).
bool wdno_conv_h3p_takes(const wdno_conv_geom& g) {
  return g.sd == 1 && g.sh == 1 && g.sw == 1 && g.OD == g.D && g.OH == g.H && g.OW == g.W && g.kw == 3 && g.kh == 3 && g.pw == 1 && g.ph == 1 &&
         (g.C % 16) == 0 && g.W <= P_WMAX && g.kd <= 8 && g.kd * g.H * g.W > 0;
}
int wdno_conv_fwd_h3_plane(int shape, const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                           const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  (void)shape;
  if (xl == nullptr) return launch_h3p<256, 64, 4, 1, true>(xh, xh, wh, wh, sx, sw, bias, residual, y, p, st);
  return launch_h3p<256, 64, 4, 1, false>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
}
