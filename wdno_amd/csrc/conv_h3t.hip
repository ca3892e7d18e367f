// conv_h3t.hip -- "tap-resident" variant of the LDS-DMA convolution (conv_h3d.hip) for stride-1 convolutions whose output
// grid equals the input grid (every 3 x 3 x 3 ResnetBlock convolution of the U-Nets and its data gradient).
//
// conv_h3d.hip walks the reduction as (dz, dy, 32-wide chunk of the kw*C run): the run of a pixel overlaps the run of its
// W-neighbour in all but C of its kw*C values, so every x value travels L2 -> LDS kw times per (dz, dy) -- and on gfx950 that
// traffic is what the kernel pays for: with the DMA issue switched off the same MFMA stream needs 23 % fewer shader cycles AND
// the chip clocks 1.54 -> 2.00 GHz (tools/bench_conv.py --stamps, level-0 64 -> 64 layer).
// Here one LDS stage holds, for one (dz, dy) and one block of 32 input channels,
//     A: the BM + kw - 1 consecutive source pixels the tile's BM output pixels touch  (rows of 64 B, fp16 hi / lo planes)
//     B: the kw x BN weight rows of that (dz, dy, channel block)
// and is used for kw reduction sub-steps: sub-step dx reads the A rows shifted by dx. x travels once per (dz, dy):
// 1.04 MB instead of 2.16 MB of DMA per 256 x 64 tile of a 64 -> 64 layer, with 58 instead of 120 pieces.
//   * Flat pixel indices: with equal grids the source pixel of output pixel p at tap (dz, dy, dx) is
//     p + ((dz - pd) H + (dy - ph)) W + dx - pw, so LDS row r of the stage is flat pixel  tile_p0 + r - pw + (tap-row shift).
//   * Validity: the (dz, dy) test is made by the producers on the output pixel a row is the centre tap of (rows shared
//     by two output pixels of different image rows are w-invalid for one of them); an invalid row is fetched out of range =
//     zeros. The dx test is made by the compute waves once per tile: a lane whose pixel has no W-neighbour on that side
//     reads the stage's zero row instead (the row after the halo, which every DMA pass rewrites with zeros).
//   * Two stages (58 .. 74 KB each), ONE barrier per stage, placed before the MFMAs of the stage's last sub-step: by then
//     all fragments of the stage are in registers, so the same barrier hands the buffer back to the producers and takes
//     the next stage from them; the first fragments of the next stage are in flight during those MFMAs.
//     (Tried and dropped: warming L2 for the next tile's leading source plane with plain loads from the producer waves --
//     no change on cold inputs, 0.212 vs 0.193 ms with x resident in the memory-side cache, and none in the training step.)
//     (Tried and dropped, round 4, commit 66bf75e: a "plane-resident" stage -- one dz and a 16-channel block, the BM + 2 W + 2 source pixels of
//     all nine (dy, dx), 0.70 instead of 1.04 MB of DMA per 256 x 64 tile -- 185 -> 190 us on the level-0 64 -> 64 layer. With the tap-resident
//     stage the DMA issue is worth 5 % of this kernel (0.181 vs 0.171 ms with it off); the compute waves' matrix pipe is busy for 82 % of their
//     252 K cycles at a 1.41 GHz clock: what is left is the package power limit, see profiles/r04_conv_limiter.md.)
// Arithmetic, operand formats, weight pack and epilogue are those of conv_h3d.hip; the fp32 accumulation order differs
// (dz, dy, channel block, dx instead of dz, dy, dx, channel block), so the two agree to rounding, not bit for bit.
#include "conv_common.h"
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
#define T_OOB 0x7ffffff0

template <bool LP>
__device__ __forceinline__ f32x16 t_mfma(half8 a, half8 b, f32x16 c) {
  if constexpr (LP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int4v t_rsrc(const void* ptr, unsigned bytes) {
  uint64_t a = reinterpret_cast<uint64_t>(ptr);
  int4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void t_piece(int4v rsrc, int off, unsigned lds_dst) {
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);      // (wave-uniform by construction; with nine pieces per wave the compiler kept it in a vector register)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(off), "s"(lds_dst), "s"(rsrc) : "memory");
}

// Which tap rows (dz, dy) of a tile can contribute at all (ConvP::zb_*): bit dz of z[c] / bit dy of y[c] = some output pixel of the tile has its
// source plane / source row inside the grid AND, for class c = 0 (the channel blocks that obey the zero box), in front of the box. A tile
// of BM consecutive flat pixels lies in one or two planes of an image; (dz, dy) is treated as live when dz is live for one of its planes
// and dy for one of its rows (a superset of the truly live pairs: skipping is exact, never the other way round).
struct TapLive { unsigned z[2], y[2]; };
__device__ __forceinline__ unsigned t_bits(int lo, int hi, int k) {      // bits lo .. hi of a k-bit mask (empty when hi < lo)
  lo = max(lo, 0); hi = min(hi, k - 1);
  return hi >= lo ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}
__device__ __forceinline__ TapLive t_tap_live(const wdno_conv_geom& g, int p0, int p1, int zb_d, int zb_h) {      // (pixel indices fit 32 bits: launch_h3t)
  const int hw = g.OH * g.OW;
  const int f0 = p0 / hw, f1 = p1 / hw;
  const int od0 = f0 % g.OD, od1 = f1 % g.OD;
  const int oh0 = (p0 - f0 * hw) / g.OW, oh1 = (p1 - f1 * hw) / g.OW;
  TapLive t;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int dl = c ? g.D : min(g.D, zb_d), hl = c ? g.H : min(g.H, zb_h);
    // plane od: taps dz with 0 <= od + dz - pd < dl; rows [a, b] of a plane: taps dy with b + dy - ph >= 0 and a + dy - ph < hl
    t.z[c] = f1 - f0 > 1 ? t_bits(0, g.kd - 1, g.kd) : t_bits(g.pd - od0, dl - 1 + g.pd - od0, g.kd) | t_bits(g.pd - od1, dl - 1 + g.pd - od1, g.kd);
    t.y[c] = f0 == f1 ? t_bits(g.ph - oh1, hl - 1 + g.ph - oh0, g.kh)
           : f1 - f0 == 1 ? t_bits(g.ph - (g.OH - 1), hl - 1 + g.ph - oh0, g.kh) | t_bits(g.ph - oh1, hl - 1 + g.ph, g.kh)
           : t_bits(g.ph - (g.OH - 1), hl - 1 + g.ph, g.kh);
  }
  return t;
}
// stages of a tile under the skip rule, made even (the 7-wide kernel runs stages in pairs: one all-zero stage closes an odd count)
__device__ __forceinline__ int t_live_stages(const TapLive& t, int ncb, int nzb) {
  const int n = __builtin_popcount(t.z[0]) * __builtin_popcount(t.y[0]) * nzb + __builtin_popcount(t.z[1]) * __builtin_popcount(t.y[1]) * (ncb - nzb);
  return (n + 1) & ~1;
}

// CB = channels per stage block: 32 (LDS rows of 64 B, two 16-deep sub-steps per dx) or 16 (rows of 32 B, one sub-step per dx: the 7-wide
// stem convolution, whose kw x BN weight rows of a 32-channel block would not leave room for two stages)
template <int BM, int BN, int KW, int CB>
struct TapShape {
  static constexpr int ROWB = CB * 2;                             // bytes of an LDS row (one plane)
  static constexpr int RPP = 1024 / ROWB;                         // rows per 1-KiB piece
  static constexpr int ZR = CB == 32 ? 4 : 16;                    // zero rows kept behind the window (one per bank-slot class, see a_rd)
  static constexpr int AROWS = BM + KW - 1;                       // source pixels of a stage
  static constexpr int ZB = (AROWS + ZR - 1) / ZR * ZR;           // first zero row
  static constexpr int APIECES = (ZB + ZR + RPP - 1) / RPP;       // 1-KiB pieces per A plane
  static constexpr int A_PLANE = APIECES * 1024;
  static constexpr int BPIECES = (KW * BN + RPP - 1) / RPP;
  static constexpr int B_PLANE = BPIECES * 1024;
};

template <int BM, int BN, int WM, int WN, int KW, int CB, bool LP, bool SP = false>
__global__ __launch_bounds__(512) void conv_fwd_h3t_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                            const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                            const float* __restrict__ sx, const float* __restrict__ sw,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ y, ConvP p, unsigned x_bytes, unsigned w_bytes) {
  using S = TapShape<BM, BN, KW, CB>;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int NPL = LP ? 1 : 2;
  constexpr int ROWB = S::ROWB, RPP = S::RPP, CPR = ROWB / 16;     // row bytes, rows per piece, 16-byte chunks per row
  constexpr int A_PLANE = S::A_PLANE, B_PLANE = S::B_PLANE;
  constexpr int A_LO = A_PLANE, B_HI = NPL * A_PLANE, B_LO = B_HI + B_PLANE;
  constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  constexpr int PA = (S::APIECES + 3) / 4, PB = (S::BPIECES + 3) / 4;      // pieces per producer wave and plane
  constexpr int KSN = CB / 16;                                     // 16-deep sub-steps per dx
  constexpr int NSUB = KSN * KW;                                   // sub-steps of a stage: (dx, ks)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // Split reduction (p.tsplit > 1: few pixels x many channels with hundreds of stages per tile -- the 8 x 8 and 16 x 16 levels of the Burgers U-Net
  // at batch 16): a block's unit of work is (run of stages, tile), run-major -- the blocks of an XCD share one run's slice of the weights --
  // and its epilogue writes raw partial sums to p.split_ws[run]; conv_split_reduce_kernel adds the runs in order.
  // (SP: its own instantiation -- the cursor of the unsplit kernels stays as it was)
  const int nruns = SP ? p.tsplit : 1;
  const int nvt = p.ntiles * nruns;
  const int my_tiles = (nvt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ncb = (g.C + CB - 1) / CB;                             // channel blocks (the last one may be ragged: CB == 16 only)
  // stages per tile. With an odd number of sub-steps per stage (7-wide taps on 16-channel blocks) the fragment set a stage starts on
  // alternates, so stages run in pairs: an odd count gets one all-zero stage at the end (every piece out of bounds; 1 / 148 more MFMA work
  // on the smoke stem) rather than a second copy of the loop tail for the other parity (which cost 21 spilled registers)
  const bool padded = (NSUB & 1) && ((g.kd * g.kh * ncb) & 1);   // (launch_h3t: no split reduction there)
  const int nstages = g.kd * g.kh * ncb / nruns + (padded ? 1 : 0);
  // SKIP (the 7-wide stem only): stages that cannot contribute are not run -- source plane / row outside the grid for the whole tile, or
  // inside the caller's zero box for the channel blocks that obey it (ConvP::zb_*; launch_h3t leaves zb_blocks < ncb). Producers and
  // compute waves derive the same per-tile stage count from t_tap_live; WHICH stages run only the producers need to know.
  constexpr bool SKIP = KW == 7 && !SP;
  const int nzb = SKIP ? p.zb_blocks : 0;
  const bool skipping = SKIP && p.zb_blocks >= 0 && p.debug != 57;          // debug 57: every stage (A/B, bit-identity test)
  auto tile_live = [&](int t) {
    const int vt = xcd_swizzle((int)blockIdx.x + t * (int)gridDim.x, nvt);
    const int m0 = (vt / p.tiles_n) * BM;
    return t_tap_live(g, m0, min(m0 + BM, (int)p.P) - 1, nzb > 0 ? p.zb_d : g.D, nzb > 0 ? p.zb_h : g.H);
  };

  if (wave >= WM * WN) {
    // ================================================================== producer waves
    const int pq = wave - WM * WN;
    int4v rxh = t_rsrc(xh, x_bytes), rxl = t_rsrc(xl, x_bytes), rwh = t_rsrc(wh, w_bytes), rwl = t_rsrc(wl, w_bytes);
    asm volatile("s_nop 4" : "+s"(rxh), "+s"(rxl), "+s"(rwh), "+s"(rwl));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    // lane -> (row of the piece, logical 16-byte chunk): the XOR swizzle of the fragment reads applied on the source side. 64-byte rows:
    // chunk (L & 3) ^ ((row >> 2) & 3); 32-byte rows: chunk (L & 1) ^ ((row >> 3) & 1) (the row bits used come from the lane index alone)
    const int prow = lane / CPR;
    const int c8 = (CB == 32 ? ((lane & 3) ^ ((lane >> 4) & 3)) : ((lane & 1) ^ ((lane >> 4) & 1))) * 8;
    int a_off[PA], b_off[PB];
    unsigned a_mask[PA];
    bool b_ok[PB];
    auto tap_bits = [](int c0, int k, int n) -> unsigned {
      int lo = c0 < 0 ? -c0 : 0, hi = n - c0 < k ? n - c0 : k;
      return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    int c_tile = 0, s_dz = 0, s_dy = 0, s_cb = 0, s_rem = 0;      // issue cursor: (tap row, channel block) of the next stage, stages left in the unit
    int x_uni = 0, w_uni = 0;
    TapLive lv = {{0u, 0u}, {0u, 0u}};                             // SKIP: live tap rows of the cursor's tile, stages issued for it so far
    int s_cnt = 0;
    auto next_live = [&]() {                                       // SKIP: the cursor moves to the next stage that runs (s_dz == kd: none left)
      for (;;) {
        if (++s_cb == ncb) { s_cb = 0; if (++s_dy == g.kh) { s_dy = 0; ++s_dz; } }
        if (s_dz >= g.kd) return;
        const int c = s_cb < nzb ? 0 : 1;
        if (((lv.z[c] >> s_dz) & 1u) && ((lv.y[c] >> s_dy) & 1u)) return;
      }
    };
    // piece i of this wave is piece pq + 4 i of the plane; its lane covers LDS row 16 (pq + 4 i) + prow
    auto setup_tile = [&](int t) {
      const int vt = xcd_swizzle((int)blockIdx.x + t * (int)gridDim.x, nvt);
      const int run = SP ? vt / p.ntiles : 0, tile = vt - run * p.ntiles;
      if constexpr (SP) {
        const int g0 = run * nstages;                                // first stage of the run: (dz, dy, cb) in that order
        s_cb = g0 % ncb;
        s_dy = (g0 / ncb) % g.kh;
        s_dz = g0 / (ncb * g.kh);
        s_rem = nstages;
        x_uni = ((s_dz * g.H + s_dy) * g.W * g.C + s_cb * CB) * 2;
        w_uni = ((s_dz * g.kh + s_dy) * g.K * p.R + s_cb * CB) * 2;
      }
      const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
      const int64_t p0 = (int64_t)tile_m * BM;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int r = RPP * (pq + 4 * i) + prow;
        const int64_t pr = p0 + r - g.pw;                          // output pixel this row is the centre tap of
        const bool live = r < S::AROWS && pr >= 0 && pr < p.P;
        int q = live ? (int)pr : 0;
        q /= g.OW;
        const int oh = q % g.OH; q /= g.OH;
        const int od = q % g.OD;
        a_mask[i] = live ? (tap_bits(od - g.pd, g.kd, g.D) | (tap_bits(oh - g.ph, g.kh, g.H) << 8)) : 0u;
        a_off[i] = (((int)pr - (g.pd * g.H + g.ph) * g.W) * g.C + c8) * 2;      // tap row (0, 0), channel block 0; used only when valid
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = RPP * (pq + 4 * i) + prow;                 // (dx, k) = (row / BN, row % BN)
        const int dx = row / BN, k = tile_n * BN + (row - dx * BN);
        b_ok[i] = k < g.K && dx < KW;
        b_off[i] = (k * p.R + dx * g.C + c8) * 2;
      }
    };
    bool fresh = true;                                             // SKIP: the cursor's tile has not issued a stage yet
    if (my_tiles > 0) setup_tile(0);
    const bool no_dma = p.debug == 21 || p.debug >= 100;                             // ablation (tools/bench_conv.py): compute waves alone
    auto issue_stage = [&](int buf) {
      if (skipping) {
        if (fresh) {                                               // cursor on the first stage of the tile that runs
          lv = tile_live(c_tile);
          s_dz = 0; s_dy = 0; s_cb = -1; s_cnt = 0; fresh = false;
          next_live();
        }
        x_uni = ((s_dz * g.H + s_dy) * g.W * g.C + s_cb * CB) * 2;
        w_uni = ((s_dz * g.kh + s_dy) * g.K * p.R + s_cb * CB) * 2;
      }
      const bool pad = s_dz == g.kd;                               // the all-zero stage (a_mask has 16 bits)
      const unsigned need = pad ? 0x10000u : (1u << s_dz) | (1u << (8 + s_dy));
      const unsigned dst = lds0 + buf * STAGE + pq * 1024;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        if (pq + 4 * i >= S::APIECES) continue;                    // compile-time for all but the last i
        const int off = ((a_mask[i] & need) == need) ? a_off[i] + x_uni : T_OOB;
        if (no_dma) continue;
        t_piece(rxh, off, dst + i * 4096);
        if (!LP) t_piece(rxl, off, dst + A_LO + i * 4096);
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        if (pq + 4 * i >= S::BPIECES) continue;                    // (the 14 weight pieces of the 7-wide stem are not a multiple of four)
        const int off = b_ok[i] && !pad ? b_off[i] + w_uni : T_OOB;
        if (no_dma) continue;
        t_piece(rwh, off, dst + B_HI + i * 4096);
        if (!LP) t_piece(rwl, off, dst + B_LO + i * 4096);
      }
      if (skipping) {
        ++s_cnt;
        if (!pad) next_live();
        if (s_dz >= g.kd && (pad || !(s_cnt & 1))) {              // tile finished (an odd count gets the all-zero stage first): next tile
          fresh = true;
          if (++c_tile < my_tiles) setup_tile(c_tile);
        }
        return;
      }
      x_uni += ROWB; w_uni += ROWB;
      if constexpr (SP) {
        if (++s_cb == ncb) {
          s_cb = 0;
          if (++s_dy == g.kh) { s_dy = 0; ++s_dz; }
          x_uni = (s_dz * g.H + s_dy) * g.W * g.C * 2;
          w_uni = (s_dz * g.kh + s_dy) * g.K * p.R * 2;
        }
        if (--s_rem == 0 && ++c_tile < my_tiles) setup_tile(c_tile);   // unit finished: move the cursor to the next one
      } else if (pad || ++s_cb == ncb) {
        s_cb = 0;
        if (!pad && ++s_dy == g.kh) { s_dy = 0; ++s_dz; }
        if (s_dz == g.kd && (pad || !padded)) {          // tile finished: move the cursor to the next one
          s_dz = 0;
          if (++c_tile < my_tiles) setup_tile(c_tile);
        }
        x_uni = (s_dz * g.H + s_dy) * g.W * g.C * 2;
        w_uni = (s_dz * g.kh + s_dy) * g.K * p.R * 2;
      }
    };
    const int total = my_tiles * nstages;
    int buf = 0;
    if (skipping) {                                                // the stage count follows from the cursor: one barrier per issued stage
      if (my_tiles > 0) issue_stage(0);
      for (int left = my_tiles > 0 ? 1 : 0; left > 0; --left) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : : : "memory");
        buf ^= 1;
        if (c_tile < my_tiles) { issue_stage(buf); ++left; }
      }
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
      return;
    }
    if (total > 0) issue_stage(0);
    for (int gs = 0; gs < total; ++gs) {
      // stage gs has landed -> meet the compute waves (which have all fragments of stage gs - 1 in registers by now), then refill
      // the buffer of stage gs - 1 with stage gs + 1
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
  auto swz = [](int row) { return CB == 32 ? ((row >> 2) & 3) : ((row >> 3) & 1); };      // XOR applied to the chunk index of a row
  const int b_rd = B_HI + b_row * ROWB + ((hh ^ swz(b_row)) * 16);             // dx = 0, b = 0, ks = 0 (ks = 1 is ^ 32, 64-byte rows only)
  int a_rd[TM][KW];                                                             // per tile: A row address of (a, dx), or the zero row
  half8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];                         // [set][tile]
  auto read_frags = [&](int set, int boff, int dx, int ks) {
    const char* st = smem + boff;
    const int x = ks * 32;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      fah[set][a] = *reinterpret_cast<const half8*>(st + (a_rd[a][dx] ^ x));
      if (!LP) fal[set][a] = *reinterpret_cast<const half8*>(st + A_LO + (a_rd[a][dx] ^ x));
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      fbh[set][b] = *reinterpret_cast<const half8*>(st + dx * (BN * ROWB) + b * (32 * ROWB) + (b_rd ^ x));
      if (!LP) fbl[set][b] = *reinterpret_cast<const half8*>(st + B_PLANE + dx * (BN * ROWB) + b * (32 * ROWB) + (b_rd ^ x));
    }
  };
  f32x16 acc[TM][TN];
  auto mfma_set = [&](int set) {
    if constexpr (!LP) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = t_mfma<false>(fbh[set][b], fal[set][a], acc[a][b]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = t_mfma<false>(fbl[set][b], fah[set][a], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = t_mfma<LP>(fbh[set][b], fah[set][a], acc[a][b]);
  };
  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sw[0]);
  const int sdbg = p.debug >= 100 ? p.debug - 100 : p.debug;
  const bool stamps = sdbg == 23 || sdbg == 24 || sdbg == 26;     // see conv_h3d.hip
  const uint64_t c_begin = stamps ? __builtin_amdgcn_s_memtime() : 0, r_begin = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t c_epi = 0;
  int boff = 0;                                                   // byte offset of the stage being read
  for (int t = 0; t < my_tiles; ++t) {
    const int vt = xcd_swizzle((int)blockIdx.x + t * (int)gridDim.x, nvt);
    const int run = SP ? vt / p.ntiles : 0, tile = vt - run * p.ntiles;
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    // the tile-invariant halves of the addresses below (two candidates per (a, dx): 28 values with 7-wide taps) would otherwise be hoisted
    // out of the tile loop and held across the stage loop -- the 7-wide kernel spilled 62 registers over that; recomputing them is
    // ~100 VALU instructions per tile
    int row0 = m_base + li;
    asm volatile("" : "+v"(row0));
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int ow = ((int)m0 + row0 + a * 32) % g.OW;                  // P < 2^31 (launch_h3t)
#pragma unroll
      for (int dx = 0; dx < KW; ++dx) {
        const int row = row0 + a * 32 + (p.debug == 15 ? 0 : dx);      // debug 15 / 16: bank-conflict probes (wrong results)
        const bool ok = p.debug == 16 || (unsigned)(ow - g.pw + dx) < (unsigned)g.W;
        // the stand-in zero row keeps the bank slot of the real one (same row % 4, same swizzled chunk): a conflict-free 16-lane group
        // uses all 16 slots once, so a lane redirected anywhere else collides with a neighbour (measured: 13 % more LDS cycles)
        const int zrow = S::ZB + (row & (S::ZR - 1));
        a_rd[a][dx] = (ok ? row : zrow) * ROWB + ((hh ^ swz(row)) * 16);
      }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) asm volatile("v_mov_b32 %0, 0" : "=v"(acc[a][b][e]));   // not `= 0.f`: the 7-wide kernel kept a second,
    lgkm0_barrier();                                                                          // loop-invariant set of 64 zero registers to start each tile from
    read_frags(0, boff, 0, 0);
    // One scheduling region per sub-step: the fragment reads of the NEXT sub-step are dealt one per MFMA of this one (not all reads
    // first, the order the kernel was first written in) -- a wave issues in order, so a block of 8 ds_read_b128 ahead
    // of the MFMAs adds its issue time (the LDS is also serving the other three compute waves) to every sub-step.
    constexpr int NRD = NPL * (TM + TN), NMF = (LP ? 1 : 3) * TM * TN;
    constexpr int RPM = (NRD + NMF - 1) / NMF;                     // reads behind each MFMA until they are used up
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i * RPM < NRD) __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
      }
    };
    // a stage: sub-step `sub` (dx = sub / KSN, ks = sub % KSN) works on fragment set (PAR + sub) & 1 while the reads of sub + 1 -- at the
    // last sub-step, behind the barrier, the first reads of the next stage -- go to the other set. With an odd number of sub-steps
    // (7-wide taps on 16-channel blocks) the starting set alternates from stage to stage.
    auto stage_body = [&](auto PARC, auto LASTC) {
      constexpr int PAR = decltype(PARC)::value;
      constexpr bool LAST = decltype(LASTC)::value;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        if (sub + 1 < NSUB) {
          read_frags((PAR + sub + 1) & 1, boff, (sub + 1) / KSN, (sub + 1) % KSN);
        } else if (!LAST) {
          boff = STAGE - boff;
          lgkm0_barrier();
          read_frags((PAR + NSUB) & 1, boff, 0, 0);
        }
        mfma_set((PAR + sub) & 1);
        if (sub + 1 < NSUB || !LAST) interleave();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if constexpr (NSUB % 2 == 0) {
      for (int s = 0; s + 1 < nstages; ++s) stage_body(P0{}, std::false_type{});
      stage_body(P0{}, std::true_type{});
    } else if constexpr (KW != 7) {                                  // (the 16-channel-block experiment of fwd_h3t, shape 8)
      for (int s = 0; s + 2 < nstages; s += 2) { stage_body(P0{}, std::false_type{}); stage_body(P1{}, std::false_type{}); }
      stage_body(P0{}, std::false_type{});                         // nstages is even
      stage_body(P1{}, std::true_type{});
    } else {
      // The 7-wide stem: 147 stages x 7 sub-steps x 3 products = 3 087 sequential fp32 accumulations per output -- the one place where the split
      // path was measurably further from exact than torch's fp32 (tools/diagnostics/error_trace.py, round 5: 1.39e-6 at the stem's output against
      // 2.7e-7 for the exact-fp32 kernel with its two-level sums, and that offset rode through the whole network). Two-level sums here too: the
      // accumulators of one tap row (dz, dy) -- at most ncb stages, 63 accumulations -- are added into a second set when the row is done.
      // The rows are counted from the same liveness masks as the producers' cursor, so a skipped all-zero stage changes nothing: the
      // skipping and the non-skipping launch (debug 57) agree bit for bit.
      const unsigned allz = (1u << g.kd) - 1u, ally = (1u << g.kh) - 1u;
      const TapLive lvc = skipping ? tile_live(t) : TapLive{{allz, allz}, {ally, ally}};
      const int nst = t_live_stages(lvc, ncb, nzb);                // (not skipping: kd kh ncb made even = nstages)
      f32x16 acc2[TM][TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) asm volatile("v_mov_b32 %0, 0" : "=v"(acc2[a][b][e]));
      int g_dz = 0, g_dy = -1, flush_at = 0;
      auto next_row = [&]() {                                      // flush_at = stages run when the next tap row that runs at all is done
        for (;;) {
          if (++g_dy == g.kh) { g_dy = 0; ++g_dz; }
          if (g_dz >= g.kd) { flush_at = 0x7fffffff; return; }
          const int n = (int)((lvc.z[0] >> g_dz) & (lvc.y[0] >> g_dy) & 1u) * nzb + (int)((lvc.z[1] >> g_dz) & (lvc.y[1] >> g_dy) & 1u) * (ncb - nzb);
          if (n) { flush_at += n; return; }
        }
      };
      next_row();
      auto row_done = [&](int done) {
        if (done != flush_at) return;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              acc2[a][b][e] += acc[a][b][e];
              asm volatile("v_mov_b32 %0, 0" : "=v"(acc[a][b][e]));
            }
        next_row();
      };
      for (int s = 0; s + 2 < nst; s += 2) {
        stage_body(P0{}, std::false_type{}); row_done(s + 1);
        stage_body(P1{}, std::false_type{}); row_done(s + 2);
      }
      stage_body(P0{}, std::false_type{}); row_done(nst - 1);      // nst is even
      stage_body(P1{}, std::true_type{});
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] += acc2[a][b][e];
    }
    boff = STAGE - boff;
    const uint64_t e_begin = stamps ? __builtin_amdgcn_s_memtime() : 0;
    // epilogue: as conv_h3d.hip (accumulator tile is [channel][pixel]: a lane owns one pixel and runs of four channels)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int64_t pm = m0 + m_base + a * 32 + li;
      if (pm >= p.P) continue;
      const int64_t yr = pm;                                      // identity output placement only (conv_h3d.hip checks it before coming here)
      if constexpr (SP) {                                         // raw partial sums of this run; scale, bias, residual and amax in the reduce pass
        float* wrow = p.split_ws + ((int64_t)run * p.P + pm) * g.K;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const int kc = n0 + n_base + b * 32 + 8 * e4 + 4 * hh;
            if (kc < g.K) *reinterpret_cast<float4*>(wrow + kc) = make_float4(acc[a][b][4 * e4], acc[a][b][4 * e4 + 1], acc[a][b][4 * e4 + 2], acc[a][b][4 * e4 + 3]);
          }
        }
        continue;
      }
      float* yrow = y + yr * g.K;
      const float* rrow = res ? res + yr * g.K : nullptr;
      if (LP && p.out_bf16) {                     // bf16 storage (no residual; K % 8 == 0): 16-byte stores of eight channels per lane
        unsigned short* y16 = reinterpret_cast<unsigned short*>(y) + yr * g.K;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
          for (int e4 = 0; e4 < 4; e4 += 2) {
            const int k0 = n0 + n_base + b * 32 + 8 * e4;          // first channel of group e4; group e4 + 1 starts at k0 + 8
            if (k0 >= g.K) continue;
            const int ka = k0 + 4 * hh, kb = ka + 8;
            float4 va = make_float4(acc[a][b][4 * e4] * inv, acc[a][b][4 * e4 + 1] * inv, acc[a][b][4 * e4 + 2] * inv, acc[a][b][4 * e4 + 3] * inv);
            float4 vb = make_float4(acc[a][b][4 * e4 + 4] * inv, acc[a][b][4 * e4 + 5] * inv, acc[a][b][4 * e4 + 6] * inv, acc[a][b][4 * e4 + 7] * inv);
            if (bias) {
              const float4 ta = *reinterpret_cast<const float4*>(bias + ka);
              va.x += ta.x; va.y += ta.y; va.z += ta.z; va.w += ta.w;
              if (k0 + 8 < g.K) { const float4 tb = *reinterpret_cast<const float4*>(bias + kb); vb.x += tb.x; vb.y += tb.y; vb.z += tb.z; vb.w += tb.w; }
            }
            const uint4 o = bf16x8_from_runs(va, vb);
            if (k0 + 8 * hh < g.K) *reinterpret_cast<uint4*>(y16 + k0 + 8 * hh) = o;
            am = amax4(amax4(am, va), vb);
          }
        }
        continue;
      }
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
    if (stamps) c_epi += __builtin_amdgcn_s_memtime() - e_begin;
  }
  if (stamps) am = sdbg == 23 ? (float)(__builtin_amdgcn_s_memtime() - c_begin) : sdbg == 24 ? (float)c_epi : (float)(__builtin_amdgcn_s_memrealtime() - r_begin);
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * (WM * WN) + wave);
}

// y = (run 0 + run 1 + ... in this order) / (sx sw) + bias + residual of a split reduction, and the amax record of y
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(const float4* __restrict__ ws, int nruns, int64_t n4, int K4, const float* __restrict__ sx,
                                                                 const float* __restrict__ sw, const float4* __restrict__ bias, const float4* res,
                                                                 float4* y, float* __restrict__ amax_rec) {
  const float inv = sx ? 1.0f / (sx[0] * sw[0]) : 1.0f;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v = ws[i];
    for (int r = 1; r < nruns; ++r) { const float4 u = ws[(int64_t)r * n4 + i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    if (bias) { const float4 tb = bias[(int)(i % K4)]; v.x += tb.x; v.y += tb.y; v.z += tb.z; v.w += tb.w; }
    if (res) { const float4 tr = res[i]; v.x += tr.x; v.y += tr.y; v.z += tr.z; v.w += tr.w; }
    y[i] = v;
    am = amax4(am, v);
  }
  if (amax_rec) wave_amax_emit(am, amax_rec, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
}

static int t_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

template <int BM, int BN, int WM, int WN, int KW, int CB, bool LP, bool SP = false>
static int launch_h3t(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                      const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  using S = TapShape<BM, BN, KW, CB>;
  const wdno_conv_geom& g = p.g;
  int64_t tiles_m = cdiv64(p.P, BM);
  p.tiles_n = cdiv(g.K, BN);
  int64_t nt = tiles_m * p.tiles_n;
  if (nt > 0x7fffffff) return WDNO_EUNSUPPORTED;
  p.ntiles = (int)nt;
  constexpr size_t lds = (size_t)2 * (LP ? 1 : 2) * (S::A_PLANE + S::B_PLANE);
  static_assert(lds <= 160 * 1024, "two stages must fit the LDS");
  const int64_t x_elems = (int64_t)g.N * g.D * g.H * g.W * g.C;
  const int64_t w_elems = (int64_t)g.kd * g.kh * g.K * p.R;
  if (x_elems * 2 >= T_OOB || w_elems * 2 >= T_OOB || p.P >= 0x7fffffff - BM) return WDNO_EUNSUPPORTED;
  if (SP) {                                       // (conv_h3d.hip asks for it on whole 32-channel blocks and 3-wide taps only)
    const int nst = g.kd * g.kh * ((g.C + CB - 1) / CB);
    if (p.tsplit < 2 || ((CB / 16) * KW & 1) || nst % p.tsplit || !p.split_ws || (size_t)p.tsplit * p.P * g.K * sizeof(float) > p.split_ws_bytes || (g.K & 3))
      return WDNO_EINVAL;
  } else p.tsplit = 1;
  int grid = t_num_cus() & ~7;
  if (grid < 8) grid = 8;
  if ((int64_t)p.ntiles * p.tsplit < grid) grid = p.ntiles * p.tsplit;
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute((const void*)conv_fwd_h3t_kernel<BM, BN, WM, WN, KW, CB, LP, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  conv_fwd_h3t_kernel<BM, BN, WM, WN, KW, CB, LP, SP><<<grid, 512, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)wh, (const _Float16*)wl,
                                                                    sx, sw, bias, residual, y, p, (unsigned)(x_elems * 2), (unsigned)(w_elems * 2));
  if (p.tsplit > 1) {
    const int64_t n4 = p.P * g.K / 4;
    conv_split_reduce_kernel<<<(unsigned)std::min<int64_t>((n4 + 255) / 256, 2048), 256, 0, st>>>(
        (const float4*)p.split_ws, p.tsplit, n4, g.K / 4, LP ? nullptr : sx, sw, (const float4*)bias, (const float4*)residual, (float4*)y, p.amax_rec);
  }
  return WDNO_OK;
}

// Geometries the tap-resident kernel takes: stride 1, output grid == input grid (so flat pixel indices shift by constants),
// kw == 3, whole 32-channel blocks. `shape` = tile shape chosen by the caller (conv_h3d.hip: 0 = 128 x 128, 1 = 192 x 128,
// 2 = 256 x 64, 3 = 192 x 64, 4 = 160 x 128 (one wave row, four wave columns), 5 = 320 x 64).
// ... or kw == 7 on whole 16-channel blocks with at most 64 output channels (the stem of the smoke U-Net: 42 -> 48 channels in the planes):
// there the weight rows of a 32-channel block (7 x 64 x 64 B per plane) would not leave room for two stages, so a stage is a 16-channel
// block with LDS rows of 32 B and one 16-deep sub-step per dx (debug 20: the chunked kernel instead).
static bool t_stem(const wdno_conv_geom& g) { return g.kw == 7 && (g.C % 16) == 0 && g.K <= 64 && wdno_debug_mode != 20; }
bool wdno_conv_h3t_takes(const wdno_conv_geom& g) {
  return g.sd == 1 && g.sh == 1 && g.sw == 1 && g.OD == g.D && g.OH == g.H && g.OW == g.W && ((g.kw == 3 && (g.C % 32) == 0) || t_stem(g)) &&
         g.kd <= 8 && g.kh <= 8 && wdno_debug_mode != 8;
}
// Runs the reduction of a 128 x 128-tiled layer is cut into (conv_h3d.hip; 1 = no split): four where that still is one round of the persistent
// grid, of whole stages, at least 8 per run. (Two runs were measured too: 256 -> 256 channels at 16 x 16 x 16 samples 26 -> 32 us, the
// 1024 -> 1536 data gradient at 8 x 8 116 -> 118 us -- no gain; four: 1024 -> 1024 93 -> 64, 512 -> 512 40 -> 32, 1536 -> 1024 132 -> 84 us.)
int wdno_conv_h3t_split(const wdno_conv_geom& g, int64_t P, int cus) {
  if (wdno_debug_mode == 56 || g.kw != 3 || (g.C % 32) || g.K < 128 || (g.K & 3)) return 1;
  const int64_t t128 = cdiv64(P, 128) * cdiv(g.K, 128);
  const int nst = g.kd * g.kh * (g.C / 32);
  return t128 * 4 <= cus && nst % 4 == 0 && nst / 4 >= 8 ? 4 : 1;
}
template <bool LP>
static int fwd_h3t(int shape, const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                   const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  if (shape == 0 && p.tsplit > 1 && !LP) return launch_h3t<128, 128, 2, 2, 3, 32, false, true>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (shape == 0) return launch_h3t<128, 128, 2, 2, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (shape == 1) return launch_h3t<192, 128, 2, 2, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (shape == 3) return launch_h3t<192, 64, 2, 2, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (shape == 4) return launch_h3t<160, 128, 1, 4, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (shape == 5) return launch_h3t<320, 64, 2, 2, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (shape == 6) return launch_h3t<64, 64, 2, 2, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);      // few pixels x many channels (Burgers 8 x 8 level)
  if (shape == 7) return launch_h3t<128, 64, 2, 2, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  // experiment (debug 45, tools/bench_conv.py): 128 x 64 accumulator tile per compute wave -- 512 x 64 block tiles on 16-channel stages (rows
  // of 32 B: two stages of 32-channel blocks would need 180 KB), 12 fragment reads per 24 matrix instructions instead of 8 per 12
  if (shape == 8) return launch_h3t<512, 64, 4, 1, 3, 16, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  return launch_h3t<256, 64, 4, 1, 3, 32, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
}
int wdno_conv_fwd_h3_tap(int shape, const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                         const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  // single-plane (bf16) mode has a third of the MFMA work per operand byte: with >= 128 output channels and enough tiles, 256 x 128
  // tiles (64 x 128 per wave: 6 fragment reads per 8 MFMAs, 41 KB per stage) -- debug 17: the shapes of the split mode
  if (p.g.kw == 7) {
    const int ncb = (p.g.C + 15) / 16;
    if (p.zb_blocks >= ncb) p.zb_blocks = ncb - 1;          // (a tile must keep at least its centre stage: t_live_stages)
    // less than one round of 256-pixel tiles (the stem at batch 1: 150 tiles on 256 CUs): 192-pixel tiles fill the chip and end a quarter
    // earlier (128-pixel tiles would need a second round) -- debug 72: the 256-pixel tiles always
    const int cus = t_num_cus();
    if (xl != nullptr && cdiv64(p.P, 256) < cus && cdiv64(p.P, 192) <= cus && cdiv64(p.P, 192) > cdiv64(p.P, 256) && wdno_debug_mode != 72)
      return launch_h3t<192, 64, 2, 2, 7, 16, false>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
    if (xl == nullptr) return launch_h3t<256, 64, 4, 1, 7, 16, true>(xh, xh, wh, wh, sx, sw, bias, residual, y, p, st);
    return launch_h3t<256, 64, 4, 1, 7, 16, false>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  }
  if (xl == nullptr && p.g.K > 64 && cdiv64(p.P, 256) * cdiv(p.g.K, 128) >= 2 * t_num_cus() && wdno_debug_mode != 17)
    return launch_h3t<256, 128, 4, 1, 3, 32, true>(xh, xh, wh, wh, sx, sw, bias, residual, y, p, st);
  if (xl == nullptr) return fwd_h3t<true>(shape, xh, xh, wh, wh, sx, sw, bias, residual, y, p, st);
  return fwd_h3t<false>(shape, xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
}
