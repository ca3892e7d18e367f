// conv_wgrad_h3d.hip -- weight gradient, 3 x fp16 split, operands streamed by LDS-DMA (gfx950).
//
// dwp[tap][k][r] = sum_p dy[p][k] * x[p shifted by tap][r]  (r = dx*C + c), same arithmetic and workspace contract as
// conv_wgrad_h3_kernel (conv_h3.hip). Structure = the forward DMA kernel's (conv_h3d.hip):
//   * persistent 512-thread blocks, one per CU: waves 0-3 compute (2 x 2 over the BM x BN tile), waves 4-7 produce;
//   * NS-stage LDS ring, one barrier per 32-pixel step, hand-counted vmcnt so the pieces of later steps stay in flight;
//   * work items (k tile, tap row, r tile, pixel split) are walked in a fixed order, every item has the same number of
//     steps (pixels past the end contribute zeros), so producers and consumers count steps without talking.
// What is specific to the weight gradient:
//   * both operands are pixel-major in memory while the MFMA wants 8 consecutive pixels per lane -> fragments are read
//     with ds_read_b64_tr_b16 (see conv_h3.hip). The LDS plane layout is [4-pixel group][64-byte unit = 32 channels]
//     [pixel in group][64 B]. A DMA piece is lane-linear (lane L lands at +16 L), so the LDS layout IS the lane -> address
//     map of the fetch: here four consecutive lanes fetch 64 contiguous bytes of one pixel row and a piece is 4 pixels x
//     256 contiguous bytes. (The first layout, [group][16-byte chunk][pixel][16 B], put four different pixels into every
//     lane quad: 64 separate 16-byte requests per instruction, producers alone 0.353 of the kernel's 0.395 ms on the
//     64 -> 64 level-0 layer at ~19 B/clk per CU; now 0.269 ms for the kernel, 253 TFLOP/s.) On the read side the 32 lanes
//     that one transpose read serves together (two 16-lane groups = the two 16-channel halves of the units of 4 pixels)
//     still touch 256 contiguous bytes: conflict-free without padding (which lane-linear DMA could not produce).
//   * the per-pixel geometry comes from the host-cached pixel table (wdno_conv_pixel_table). Producer lanes fetch the
//     records of the pixels they serve two steps before they need them, with loads that are part of the same counted
//     queue as the pieces.
//   * the accumulator tile is kept as [r][k] (x fragment as the MFMA row operand), so a lane owns one k and runs of four
//     consecutive r: 16-byte stores into dwp / the split workspace.
#include "conv_common.h"
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
#define WD_OOB 0x7ffffff0
// LP = single bf16 plane per operand, one product (see conv_h3d.hip); !LP = (hi, lo) fp16 planes, three products
template <bool LP>
__device__ __forceinline__ f32x16 mfma_16(half8 a, half8 b, f32x16 c) {
  if constexpr (LP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct WgradDP {
  ConvP c;
  int tiles_k, tiles_r, splits, nsteps;     // nsteps per item (same for all)
  int items;                                // tiles_k * ntap * tiles_r * splits
  int pix_per_split;
  int xcd_chunk;                            // items per XCD (grid is a multiple of 8), or 0: items dealt round-robin over the blocks
  int splitpair, n_a;                       // window kernel: the odd tap row paired across two pixel splits (items >= n_a), see conv_wgrad_h3w_kernel
};

__device__ __forceinline__ int4v wd_rsrc(const void* ptr, unsigned bytes) {
  uint64_t a = reinterpret_cast<uint64_t>(ptr);
  int4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void wd_piece(int4v rsrc, int off, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(off), "s"(lds_dst), "s"(rsrc) : "memory");
}
__device__ __forceinline__ int4v wd_load_rec(int4v rsrc, int off) {
  int4v v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(rsrc) : "memory");
  return v;
}
// transpose-read fragment: channels ch0..ch0+31 (MFMA row/column index = lane & 31) x pixels pix0..pix0+15 (k index) of a
// plane laid out [pixel / 4][chunk16][pixel % 4][16 B] with NCH chunks per pixel
template <int NCH>
__device__ __forceinline__ half8 wd_frag(const char* plane, int lane_off, int pix0, int ch0) {
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef short4v __attribute__((address_space(3))) * lds_s4;
  const char* p0 = plane + lane_off + (pix0 / 4) * (NCH * 64) + (ch0 / 32) * 256;
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + NCH * 64));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8, c);
}

template <int BM, int BN, int NS, bool LP>
__global__ __launch_bounds__(512) void conv_wgrad_h3d_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                              const _Float16* __restrict__ dyh, const _Float16* __restrict__ dyl,
                                                              const float* __restrict__ sx, const float* __restrict__ sdy,
                                                              const int4v* __restrict__ table, float* __restrict__ ws, WgradDP wp,
                                                              unsigned x_bytes, unsigned dy_bytes, unsigned tbl_bytes) {
  constexpr int TM = BM / 64, TN = BN / 64;              // 32-wide MFMA tiles per compute wave (2 x 2 waves)
  constexpr int ACH = BM / 8, BCH = BN / 8;              // 16-byte chunks per pixel row
  constexpr int A_PLANE = 32 * ACH * 16, B_PLANE = 32 * BCH * 16;
  constexpr int NPL = LP ? 1 : 2;                        // planes per operand
  constexpr int A_LO = A_PLANE, B_HI = NPL * A_PLANE, B_LO = B_HI + B_PLANE;
  constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  constexpr int APC = A_PLANE / 1024, BPC = B_PLANE / 1024;      // pieces per plane
  constexpr int NPAIR = APC + BPC;                                // (hi, lo) piece pairs per step
  static_assert(NPAIR % 4 == 0, "pieces are dealt to four producer waves");
  constexpr int R = NPAIR / 4;                                    // pairs (= pixel records) per producer lane and step
  constexpr int PW = NPL * R;
  constexpr int LA = 2;                                           // records are fetched LA steps ahead of their pieces
  static_assert(R + (NS - 2) * (R + PW) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const ConvP& p = wp.c;
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // Items are numbered (split, k tile, tap row, r tile) with the r tile running fastest, and every XCD (block b runs on XCD
  // b % 8) owns one CONTIGUOUS range of them: all tap rows of a pixel split then stream the same dy rows and (shifted) x rows
  // through the same L2 at the same time. Dealt round-robin, the nine taps of a split land on all eight XCDs and each L2
  // fetches the whole of x and dy: 1.34 GB of fabric traffic per launch on the 64 -> 64 level-0 layer for 157 MB of operands.
  int it_first, it_stride, my_items;
  if (wp.xcd_chunk > 0) {
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int beg = xcd * wp.xcd_chunk, end = min(beg + wp.xcd_chunk, wp.items);
    it_stride = (int)gridDim.x >> 3;
    it_first = beg + idx;
    my_items = it_first < end ? (end - it_first + it_stride - 1) / it_stride : 0;
  } else {
    it_first = (int)blockIdx.x; it_stride = (int)gridDim.x;
    my_items = (wp.items - it_first + it_stride - 1) / it_stride;
  }
  const int ntap = g.kd * g.kh;
  auto decode_item = [&](int t, int& tile_k, int& tap, int& tile_r, int& split) {
    int id = it_first + t * it_stride;
    tile_r = id % wp.tiles_r; id /= wp.tiles_r;
    tap = id % ntap; id /= ntap;
    tile_k = id % wp.tiles_k;
    split = id / wp.tiles_k;
  };

  if (wave >= 4) {
    // ================================================================== producer waves
    const int pq = wave - 4;
    int4v rxh = wd_rsrc(xh, x_bytes), rxl = wd_rsrc(xl, x_bytes), rdh = wd_rsrc(dyh, dy_bytes), rdl = wd_rsrc(dyl, dy_bytes);
    int4v rtb = wd_rsrc(table, tbl_bytes);
    asm volatile("s_nop 4" : "+s"(rxh), "+s"(rxl), "+s"(rdh), "+s"(rdl), "+s"(rtb));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    // pair j of this producer = global pair pq + 4*j: pairs [0, APC) are dy pieces, [APC, NPAIR) are x pieces.
    // lane L of piece q covers the 16-byte slot f = 64*q + L of the plane: pixel = 4*(f / (16*NU)) + (f / 4) % 4, unit = (f / 16) % NU, chunk f % 4 of the unit
    int row[R], chk[R];            // pixel inside the step, channel offset (elements)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int gp = pq + 4 * j;
      const bool isA = gp < APC;
      const int f = 64 * (isA ? gp : gp - APC) + lane;
      const int nch = isA ? ACH : BCH;
      const int nu = nch / 4;                         // 64-byte units (32 channels) per pixel
      row[j] = 4 * (f / (16 * nu)) + ((f >> 2) & 3);
      chk[j] = (((f >> 4) % nu) * 4 + (f & 3)) * 8;
    }
    // cursors over the global step stream of this block: (item slot, step)
    struct Cur { int t, s; };
    Cur rc{0, 0}, pc{0, 0};          // record cursor, piece cursor
    int r_pbeg = 0;                  // first pixel of the record cursor's item
    int c_dz = 0, c_dy = 0, c_tapoff = 0;                     // piece cursor's item: tap row and its element offset in x
    int i_off[R], i_dx[R];                                     // per pair: channel byte offset (dy) / run element offset and dx (x)
    bool i_ok[R];
    auto load_item_r = [&]() {
      int tk, tap, tr, sp;
      decode_item(rc.t, tk, tap, tr, sp);
      r_pbeg = sp * wp.pix_per_split;
    };
    auto load_item_p = [&]() {
      int tk, tap, tr, sp;
      decode_item(pc.t, tk, tap, tr, sp);
      c_dz = tap / g.kh; c_dy = tap - c_dz * g.kh;
      c_tapoff = (c_dz * g.H + c_dy) * g.W * g.C;
#pragma unroll
      for (int j = 0; j < R; ++j) {                            // everything that only depends on the item, once per item
        if (pq + 4 * j < APC) {
          const int k = tk * BM + chk[j];
          i_ok[j] = k < g.K;
          i_off[j] = k * 2;
          i_dx[j] = 0;
        } else {
          const int r = tr * BN + chk[j];
          i_ok[j] = r < p.R;
          i_off[j] = r;
          i_dx[j] = r / g.C;
        }
      }
    };
    if (my_items > 0) { load_item_r(); load_item_p(); }
    int4v rec[LA + 1][R];
    bool rok[LA + 1][R];
    auto fetch = [&](auto SLOT) {
      constexpr int S = decltype(SLOT)::value;
      const bool live = rc.t < my_items;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int pm = r_pbeg + rc.s * 32 + row[j];
        rok[S][j] = live && pm < (int)p.P;
        rec[S][j] = wd_load_rec(rtb, rok[S][j] ? pm * 16 : WD_OOB);
      }
      if (live && ++rc.s == wp.nsteps) { rc.s = 0; if (++rc.t < my_items) load_item_r(); }
    };
    // ablations (tools/bench_conv.py; records are still fetched): 21 = every piece out of range (the instructions issue, nothing
    // travels), 22 = no piece instructions at all
    const bool no_dma = p.debug == 21 || p.debug == 22, no_piece = p.debug == 22;
    auto issue = [&](auto SLOT, int stage) {
      constexpr int S = decltype(SLOT)::value;
      const bool live = pc.t < my_items && !no_dma;
      const unsigned sb = lds0 + stage * STAGE;
      const int k2 = g.K * 2;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        int4v e = rec[S][j];
        asm volatile("" : "+v"(e));                                   // consumers of the record stay below the counted wait
        const int gp = pq + 4 * j;                                    // pq is wave-uniform: a scalar branch
        const bool ok0 = live && rok[S][j] && i_ok[j];
        if (no_piece) continue;
        if (gp < APC) {
          const int off = ok0 ? e.x * k2 + i_off[j] : WD_OOB;
          wd_piece(rdh, off, sb + gp * 1024);
          if (!LP) wd_piece(rdl, off, sb + A_LO + gp * 1024);
        } else {
          const int d = (e.z >> 16) + c_dz, h = (int)(short)(e.z & 0xffff) + c_dy, w = e.w + i_dx[j];
          const bool ok = ok0 && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
          const int off = ok ? (e.y + c_tapoff + i_off[j]) * 2 : WD_OOB;
          wd_piece(rxh, off, sb + B_HI + (gp - APC) * 1024);
          if (!LP) wd_piece(rxl, off, sb + B_LO + (gp - APC) * 1024);
        }
      }
      if (live && ++pc.s == wp.nsteps) { pc.s = 0; if (++pc.t < my_items) load_item_p(); }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // counted wait that names the record registers of slot S, so that no reader of them can be scheduled above it
    auto wait_pin = [&](auto SLOT, auto COUNT, auto BAR) {
      constexpr int S = decltype(SLOT)::value;
      constexpr int N = decltype(COUNT)::value;
      constexpr bool WITH_BARRIER = decltype(BAR)::value;
      static_assert(R >= 3 && R <= 5, "record count");
      int4v &r0 = rec[S][0], &r1 = rec[S][1], &r2 = rec[S][2], &r3 = rec[S][R > 3 ? 3 : 0], &r4 = rec[S][R > 4 ? 4 : 0];
      if (WITH_BARRIER) {
        if (R == 3) asm volatile("s_waitcnt vmcnt(%3)\n\ts_barrier" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
        else if (R == 4) asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%5)\n\ts_barrier" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : "n"(N) : "memory");
      } else {
        if (R == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
        else if (R == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : "n"(N) : "memory");
      }
    };
    using YES = std::true_type;
    using NO = std::false_type;
    using ZERO = std::integral_constant<int, 0>;
    using WSTEADY = std::integral_constant<int, R + (NS - 2) * (R + PW)>;
    // prologue: records of steps 0..LA-1+NS-1 ..., pieces of steps 0..NS-2. Slot of step q = q % 3.
    static_assert(NS == 3 && LA == 2, "slot schedule below is written for a 3-stage ring and 2 steps of record look-ahead");
    fetch(S0{}); fetch(S1{}); fetch(S2{});          // records of steps 0, 1, 2
    wait_pin(S0{}, ZERO{}, NO{});
    wait_pin(S1{}, ZERO{}, NO{});
    wait_pin(S2{}, ZERO{}, NO{});
    issue(S0{}, 0);                                 // pieces of step 0
    fetch(S0{});                                    // records of step 3 (slot 0 is free again)
    issue(S1{}, 1);                                 // pieces of step 1
    // steady state, iteration gs: fetch records of step gs+4 | wait pieces(gs) | barrier | issue pieces of step gs+2
    const int total = my_items * wp.nsteps;
    int stage = 0;
    auto iter = [&](auto FS, auto IS) {
      fetch(FS);
      wait_pin(IS, WSTEADY{}, YES{});               // pieces of step gs and the records in slot IS have landed
      int nstage = stage + NS - 1;
      if (nstage >= NS) nstage -= NS;
      issue(IS, nstage);
      stage = stage + 1 == NS ? 0 : stage + 1;
    };
    for (int gs = 0; gs < total; gs += 3) {
      iter(S1{}, S2{});                              // gs % 3 == 0: fetch step gs+4 -> slot 1, issue step gs+2 from slot 2
      if (gs + 1 < total) iter(S2{}, S0{});
      if (gs + 2 < total) iter(S0{}, S1{});
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    return;
  }

  // ================================================================== compute waves
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31;
  const int gq = lane >> 4, xq = lane & 15;
  const int m_base = wm * (TM * 32), n_base = wn * (TN * 32);
  // per-lane part of the transpose-read address (see wd_frag): pixel group 2*(gq>>1), pixel xq>>2, 16-channel half gq&1 of the unit, 8-byte piece xq&3 (xq = lane & 15)
  const int offA = (2 * (gq >> 1)) * (ACH * 64) + (xq >> 2) * 64 + (gq & 1) * 32 + (xq & 3) * 8;
  const int offB = (2 * (gq >> 1)) * (BCH * 64) + (xq >> 2) * 64 + (gq & 1) * 32 + (xq & 3) * 8;
  half8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
  auto read_frags = [&](auto SET, int stage, int ks) {
    constexpr int B = decltype(SET)::value;
    const char* st = smem + stage * STAGE;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      fah[B][a] = wd_frag<ACH>(st, offA, ks * 16, m_base + a * 32);
      if (!LP) fal[B][a] = wd_frag<ACH>(st + A_LO, offA, ks * 16, m_base + a * 32);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      fbh[B][b] = wd_frag<BCH>(st + B_HI, offB, ks * 16, n_base + b * 32);
      if (!LP) fbl[B][b] = wd_frag<BCH>(st + B_LO, offB, ks * 16, n_base + b * 32);
    }
  };
  f32x16 acc[TM][TN];
  auto mfma_set = [&](auto SET) {          // row operand = x fragment (r), column operand = dy fragment (k): acc is [r][k]
    constexpr int B = decltype(SET)::value;
    if constexpr (!LP) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fbh[B][b], fal[B][a], acc[a][b]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fbl[B][b], fah[B][a], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<LP>(fbh[B][b], fah[B][a], acc[a][b]);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  // one scheduling region per half-step: the transpose reads of the next fragments are dealt out behind the MFMAs of the
  // current ones (conv_h3t.hip has the measurement), RPM reads per MFMA until they are used up
  constexpr int NRD = NPL * 2 * (TM + TN), NMF = (LP ? 1 : 3) * TM * TN, RPM = (NRD + NMF - 1) / NMF;
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i * RPM < NRD) __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
    }
  };
  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sdy[0]);
  const int hh = lane >> 5;
  int stage = 0;
  for (int t = 0; t < my_items; ++t) {
    int tile_k, tap, tile_r, split;
    decode_item(t, tile_k, tap, tile_r, split);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    lgkm0_barrier();
    read_frags(B0{}, stage, 0);
    for (int step = 0; step + 1 < wp.nsteps; ++step) {      // last step peeled, see conv_h3d.hip
      read_frags(B1{}, stage, 1);
      mfma_set(B0{});
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      stage = stage + 1 == NS ? 0 : stage + 1;
      lgkm0_barrier();
      read_frags(B0{}, stage, 0);
      mfma_set(B1{});
      interleave();
      __builtin_amdgcn_sched_barrier(0);
    }
    read_frags(B1{}, stage, 1);
    mfma_set(B0{});
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    stage = stage + 1 == NS ? 0 : stage + 1;
    mfma_set(B1{});
    __builtin_amdgcn_sched_barrier(0);
    // acc[a][b]: rows = r (n_base + b*32 + 8*(e>>2) + 4*hh + (e&3)), column = k (m_base + a*32 + li)
    float* out = ws + ((int64_t)split * ntap + tap) * (int64_t)g.K * p.R;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int kk = tile_k * BM + m_base + a * 32 + li;
      if (kk >= g.K) continue;
      float* orow = out + (int64_t)kk * p.R;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int rr = tile_r * BN + n_base + b * 32 + 8 * e4 + 4 * hh;
          if (rr < p.R)                           // R = kw*C is a multiple of 8
            *reinterpret_cast<float4*>(orow + rr) = make_float4(acc[a][b][4 * e4] * inv, acc[a][b][4 * e4 + 1] * inv, acc[a][b][4 * e4 + 2] * inv, acc[a][b][4 * e4 + 3] * inv);
        }
      }
    }
  }
}

// ====================================================================================================================
// Window variant (stride 1, output grid == input grid, kw == 3, C a multiple of 64) -- the weight-gradient counterpart of
// conv_h3t.hip. tools/bench_conv.py --ablate on the kernel above, level-0 64 -> 64 layer: 0.257 ms; with every piece out of
// range (the DMA instructions issue, nothing travels) 0.182 ms; without piece instructions 0.123 ms: it is the operand
// delivery -- 32 pieces and 32 KB of LDS writes per 72 MFMAs -- that costs half the time, not the MFMAs or the transpose reads.
// Here an item is (k tile of 64) x (TWO tap rows) x (3 dx x 64 channels): per 32-pixel step the CU receives the 32 x 64 dy
// tile and, per tap row, ONE window of 32 + 2 consecutive source pixels x 64 channels (the window of the centre tap; the
// fragment of column tile (dx, channel half) is a transpose read of window rows shifted by dx) -- 28 pieces / 28 KB per 144
// MFMAs; and a compute wave owns 64 k x 96 columns (TM = 2, TN = 3), 20 transpose reads per 18 MFMAs instead of 16 per 9.
//   * dx validity (the W-neighbour of a pixel at the end of an image row): the compute waves track ow of the four pixels a
//     lane addresses per step (two transpose reads x two 16-deep halves) and point an invalid (pixel, dx) at a window row that
//     is always zero (rows 34 .. 39 are fetched out of range).
//   * (dz, dy) validity: from the pixel-table record of the pixel the window row is the centre tap of.
//   * 14 (hi, lo) piece pairs per step: producer waves 0 / 1 carry four, 2 / 3 three (the vmcnt immediates differ, so the
//     producer body is instantiated for both counts).
//   * an odd number of tap rows (3 x 3 x 3: nine; 2-D 3 x 3: three) used to leave the second window of the last pair empty: 1 / 10 (1 / 4 in 2-D)
//     of the matrix instructions multiplied zeros (SQ_INSTS_MFMA 1.11 x the work of the layer, profiles/r05_smoke_pmc.md). Round 6: in SPLIT-PAIR
//     mode (wp.splitpair, the plan's choice when it shortens the round) the items with id >= wp.n_a pair the last tap row ACROSS TWO PIXEL SPLITS:
//     window s of such an item is that tap row over split 2 j + s, and because the two windows then belong to different pixels each gets its own dy
//     tile (the second dy plane of a stage; ordinary items leave it unused and both wave pairs read the first). Every item still has wp.nsteps
//     steps and writes whole [split][tap] slots of the workspace, so the counting of producers and consumers and the ordered reduction are
//     untouched. A block walks items of ONE kind only (the plan keeps items <= blocks in this mode), so the kind is a block-uniform branch around
//     the producer body (18 (hi, lo) piece pairs per step instead of 14: five / four per producer wave) and one scalar in the compute waves.
//   * without split-pair mode an odd number of tap rows still leaves the second window of the last pair empty (zeros; its result is not stored).
//   * dx-invalid lanes read a zero row IN THE BANK SLOT of the row they replace (rows 36 .. 39 of a window are never fetched): the constant
//     stand-in address of round 2 collided with the lanes of window row 3 mod 4 -- the 1.03e6 SQ_LDS_BANK_CONFLICT per dispatch of round 5.
template <int NS, bool LP>
__global__ __launch_bounds__(512) void conv_wgrad_h3w_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                              const _Float16* __restrict__ dyh, const _Float16* __restrict__ dyl,
                                                              const float* __restrict__ sx, const float* __restrict__ sdy,
                                                              const int4v* __restrict__ table, float* __restrict__ ws, WgradDP wp,
                                                              unsigned x_bytes, unsigned dy_bytes, unsigned tbl_bytes) {
  constexpr int KW = 3, BM = 64, CT = 64;
  constexpr int TM = 2, TN = 3;
  constexpr int ACH = BM / 8, XCH = CT / 8;               // 16-byte chunks per pixel row
  constexpr int XROWS = 32 + KW - 1;                      // window rows that carry data
  constexpr int A_PLANE = 32 * ACH * 16, W_PLANE = 5 * 1024, B_PLANE = 2 * W_PLANE;      // a window: 40 rows x 128 B, rows >= XROWS stay zero
  constexpr int ZROWS = 9 * (XCH * 64);                   // window rows 36 .. 39: never fetched, i.e. zero
  constexpr int NDY = 2;                                  // dy tiles per stage (the second one is filled by split-pair items only)
  constexpr int NPL = LP ? 1 : 2;
  constexpr int A_LO = NDY * A_PLANE, B_HI = NPL * NDY * A_PLANE, B_LO = B_HI + B_PLANE;
  constexpr int STAGE = NPL * (NDY * A_PLANE + B_PLANE);
  constexpr int APC = A_PLANE / 1024, WPC = W_PLANE / 1024;
  static_assert(APC == 4 && WPC == 5, "ordinary items: pairs 0-3 dy, 4-8 window 0, 9-13 window 1; split-pair items: 0-3 dy 0, 4-7 dy 1, 8-12 window 0, 13-17 window 1");
  constexpr int LA = 2;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const ConvP& p = wp.c;
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  int it_first, it_stride, my_items;                      // item order and XCD ownership: as conv_wgrad_h3d_kernel
  if (wp.xcd_chunk > 0) {
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int beg = xcd * wp.xcd_chunk, end = min(beg + wp.xcd_chunk, wp.items);
    it_stride = (int)gridDim.x >> 3;
    it_first = beg + idx;
    my_items = it_first < end ? (end - it_first + it_stride - 1) / it_stride : 0;
  } else {
    it_first = (int)blockIdx.x; it_stride = (int)gridDim.x;
    my_items = (wp.items - it_first + it_stride - 1) / it_stride;
  }
  const int ntap = g.kd * g.kh, npair = wp.splitpair ? ntap >> 1 : (ntap + 1) >> 1;
  // item -> (k tile, channel tile, and per window s: tap row tp[s], pixel split sp[s]). Ordinary items: tap rows 2 q, 2 q + 1 of one split;
  // split-pair items (id >= n_a): the last tap row over splits 2 j, 2 j + 1. A window whose tap row or split does not exist is empty.
  auto decode_item = [&](int t, int& tile_k, int& tile_c, int (&tp)[2], int (&sp)[2]) {
    int id = it_first + t * it_stride;
    if (id < wp.n_a) {
      tile_c = id % wp.tiles_r; id /= wp.tiles_r;
      const int q = id % npair; id /= npair;
      tile_k = id % wp.tiles_k;
      sp[0] = sp[1] = id / wp.tiles_k;
      tp[0] = 2 * q; tp[1] = 2 * q + 1;
    } else {
      id -= wp.n_a;
      tile_c = id % wp.tiles_r; id /= wp.tiles_r;
      tile_k = id % wp.tiles_k;
      const int j = id / wp.tiles_k;
      sp[0] = 2 * j; sp[1] = 2 * j + 1;
      tp[0] = tp[1] = ntap - 1;
    }
  };
  const bool kind_b = wp.splitpair && it_first >= wp.n_a;      // block-uniform: this block's items are split-pair items (one kind per block)

  if (wave >= 4) {
    // ================================================================== producer waves
    const int pq = wave - 4;
    int4v rxh = wd_rsrc(xh, x_bytes), rxl = wd_rsrc(xl, x_bytes), rdh = wd_rsrc(dyh, dy_bytes), rdl = wd_rsrc(dyl, dy_bytes);
    int4v rtb = wd_rsrc(table, tbl_bytes);
    asm volatile("s_nop 4" : "+s"(rxh), "+s"(rxl), "+s"(rdh), "+s"(rdl), "+s"(rtb));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    auto producer = [&](auto RC, auto KBC) {
      constexpr int R = decltype(RC)::value;          // pairs of this wave: global pairs pq + 4 j
      constexpr bool KB = decltype(KBC)::value;       // split-pair items: two dy tiles
      constexpr int APCT = KB ? 2 * APC : APC;        // dy pairs per step
      constexpr int PW = NPL * R;
      static_assert(R + (NS - 2) * (R + PW) <= 63, "vmcnt is a 6-bit counter");
      // Lane L of piece q of a plane covers its 16-byte slot f = 64 q + L: row 4 (f / 32) + (f / 4) % 4 (pixel of the step / window row),
      // 64-byte unit (f / 16) % 2, chunk f % 4 of the unit.
      int row[R], chk[R], slot[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int gp = pq + 4 * j;
        const bool isdy = gp < APCT;
        const int piece = isdy ? gp % APC : (gp - APCT) % WPC;
        slot[j] = isdy ? gp / APC : (gp - APCT) / WPC;                 // window (and, in split-pair items, dy tile) the pair belongs to
        const int f = 64 * piece + lane;
        row[j] = 4 * (f >> 5) + ((f >> 2) & 3);
        chk[j] = (((f >> 4) & 1) * 4 + (f & 3)) * 8;
      }
      struct Cur { int t, s; };
      Cur rc{0, 0}, pc{0, 0};          // record cursor, piece cursor
      int r_pbeg[2] = {0, 0};
      int c_dz[2], c_dy[2], c_tapoff[2];
      bool c_on[2];
      int i_off[R];
      bool i_ok[R];
      auto load_item_r = [&]() {
        int tk, tc, tp[2], sp[2];
        decode_item(rc.t, tk, tc, tp, sp);
        r_pbeg[0] = sp[0] * wp.pix_per_split; r_pbeg[1] = sp[1] * wp.pix_per_split;
      };
      auto load_item_p = [&]() {
        int tk, tc, tp[2], sp[2];
        decode_item(pc.t, tk, tc, tp, sp);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int tap = tp[s2];
          c_on[s2] = tap < ntap && sp[s2] < wp.splits;
          c_dz[s2] = tap / g.kh; c_dy[s2] = tap - c_dz[s2] * g.kh;
          c_tapoff[s2] = (c_dz[s2] * g.H + c_dy[s2]) * g.W * g.C;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (pq + 4 * j < APCT) {
            const int k = tk * BM + chk[j];
            i_ok[j] = k < g.K;
            i_off[j] = k * 2;
          } else {
            i_ok[j] = row[j] < XROWS;
            i_off[j] = g.pw * g.C + tc * CT + chk[j];          // centre tap of the window row's pixel, channel tile
          }
        }
      };
      if (my_items > 0) { load_item_r(); load_item_p(); }
      int4v rec[LA + 1][R];
      bool rok[LA + 1][R];
      auto fetch = [&](auto SLOT) {
        constexpr int S = decltype(SLOT)::value;
        const bool live = rc.t < my_items;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int pm = r_pbeg[slot[j]] + rc.s * 32 + row[j] - (pq + 4 * j < APCT ? 0 : g.pw);
          rok[S][j] = live && pm >= 0 && pm < (int)p.P;
          rec[S][j] = wd_load_rec(rtb, rok[S][j] ? pm * 16 : WD_OOB);
        }
        if (live && ++rc.s == wp.nsteps) { rc.s = 0; if (++rc.t < my_items) load_item_r(); }
      };
      auto issue = [&](auto SLOT, int stage) {
        constexpr int S = decltype(SLOT)::value;
        const bool live = pc.t < my_items;
        const unsigned sb = lds0 + stage * STAGE;
        const int k2 = g.K * 2;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          int4v e = rec[S][j];
          asm volatile("" : "+v"(e));                                   // consumers of the record stay below the counted wait
          const int gp = pq + 4 * j;                                    // pq is wave-uniform: scalar branches
          const bool ok0 = live && rok[S][j] && i_ok[j];
          if (gp < APCT) {
            const int off = ok0 ? e.x * k2 + i_off[j] : WD_OOB;
            const unsigned dst = sb + (gp / APC) * A_PLANE + (gp % APC) * 1024;
            wd_piece(rdh, off, dst);
            if (!LP) wd_piece(rdl, off, dst + A_LO);
          } else {
            const int s2 = gp - APCT < WPC ? 0 : 1;
            const int d = (e.z >> 16) + c_dz[s2], h = (int)(short)(e.z & 0xffff) + c_dy[s2];
            const bool ok = ok0 && c_on[s2] && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H;
            const int off = ok ? (e.y + c_tapoff[s2] + i_off[j]) * 2 : WD_OOB;
            wd_piece(rxh, off, sb + B_HI + (gp - APCT) * 1024);
            if (!LP) wd_piece(rxl, off, sb + B_LO + (gp - APCT) * 1024);
          }
        }
        if (live && ++pc.s == wp.nsteps) { pc.s = 0; if (++pc.t < my_items) load_item_p(); }
      };
      using S0 = std::integral_constant<int, 0>;
      using S1 = std::integral_constant<int, 1>;
      using S2 = std::integral_constant<int, 2>;
      auto wait_pin = [&](auto SLOT, auto COUNT, auto BAR) {
        constexpr int S = decltype(SLOT)::value;
        constexpr int N = decltype(COUNT)::value;
        constexpr bool WITH_BARRIER = decltype(BAR)::value;
        static_assert(R >= 3 && R <= 5, "record count");
        int4v &r0 = rec[S][0], &r1 = rec[S][1], &r2 = rec[S][2], &r3 = rec[S][R > 3 ? 3 : 0], &r4 = rec[S][R > 4 ? 4 : 0];
        if (WITH_BARRIER) {
          if (R == 3) asm volatile("s_waitcnt vmcnt(%3)\n\ts_barrier" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
          else if (R == 4) asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(N) : "memory");
          else asm volatile("s_waitcnt vmcnt(%5)\n\ts_barrier" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : "n"(N) : "memory");
        } else {
          if (R == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
          else if (R == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(N) : "memory");
          else asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : "n"(N) : "memory");
        }
      };
      using YES = std::true_type;
      using NO = std::false_type;
      using ZERO = std::integral_constant<int, 0>;
      using WSTEADY = std::integral_constant<int, R + (NS - 2) * (R + PW)>;
      static_assert(NS == 3 && LA == 2, "slot schedule below is written for a 3-stage ring and 2 steps of record look-ahead");
      fetch(S0{}); fetch(S1{}); fetch(S2{});
      wait_pin(S0{}, ZERO{}, NO{});
      wait_pin(S1{}, ZERO{}, NO{});
      wait_pin(S2{}, ZERO{}, NO{});
      issue(S0{}, 0);
      fetch(S0{});
      issue(S1{}, 1);
      const int total = my_items * wp.nsteps;
      int stage = 0;
      auto iter = [&](auto FS, auto IS) {
        fetch(FS);
        wait_pin(IS, WSTEADY{}, YES{});
        int nstage = stage + NS - 1;
        if (nstage >= NS) nstage -= NS;
        issue(IS, nstage);
        stage = stage + 1 == NS ? 0 : stage + 1;
      };
      for (int gs = 0; gs < total; gs += 3) {
        iter(S1{}, S2{});
        if (gs + 1 < total) iter(S2{}, S0{});
        if (gs + 2 < total) iter(S0{}, S1{});
      }
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    };
    // 14 pairs per step (ordinary items): producer waves 0 / 1 carry four, 2 / 3 three; 18 (split-pair items): five / four
    if (kind_b) {
      if (pq < 2) producer(std::integral_constant<int, 5>{}, std::true_type{});
      else producer(std::integral_constant<int, 4>{}, std::true_type{});
    } else {
      if (pq < 2) producer(std::integral_constant<int, 4>{}, std::false_type{});
      else producer(std::integral_constant<int, 3>{}, std::false_type{});
    }
    return;
  }

  // ================================================================== compute waves: wave w owns columns [96 w, 96 w + 96) of the 384
  const int wslot = wave >> 1, whalf = wave & 1;            // tap row of the pair, half of its six (dx, channel half) tiles
  const int li = lane & 31;
  const int gq = lane >> 4, xq = lane & 15;
  // (split-pair items: the wave pair of window 1 reads ITS dy tile, the second dy plane of the stage)
  const int offA = (2 * (gq >> 1)) * (ACH * 64) + (xq >> 2) * 64 + (gq & 1) * 32 + (xq & 3) * 8 + (kind_b ? wslot * A_PLANE : 0);
  const int chan = (gq & 1) * 32 + (xq & 3) * 8;                // byte position of this lane's 8 bytes inside a 64-byte unit
  const int pix_l = 8 * (gq >> 1) + (xq >> 2);                 // pixel (of 16) this lane addresses in the first transpose read; second: + 4
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef short short8v __attribute__((ext_vector_type(8)));
  typedef short4v __attribute__((address_space(3))) * lds_s4;
  auto x_addr = [&](int wrow, int unit) { return (wrow >> 2) * (XCH * 64) + unit * 256 + (wrow & 3) * 64 + chan; };
  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sdy[0]);
  const int hh = lane >> 5;
  const int step_ow = 32 % g.W;
  const int wbase = B_HI + wslot * W_PLANE;

  auto run = [&](auto HC) {
    constexpr int HV = decltype(HC)::value;
    // column tile b of this wave = tile 3 HV + b of the tap row's six (dx, channel half) tiles
    constexpr int DX[3] = {(3 * HV) >> 1, (3 * HV + 1) >> 1, (3 * HV + 2) >> 1};
    constexpr int UH[3] = {(3 * HV) & 1, (3 * HV + 1) & 1, (3 * HV + 2) & 1};
    // class c = 2 ks + (second read): base[b][c] = address (inside a window plane) of tile b's read of that class
    int base[TN][4], rd[TN][4], ow[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int b = 0; b < TN; ++b) base[b][c] = x_addr(16 * (c >> 1) + 4 * (c & 1) + pix_l + DX[b], UH[b]);
    auto select = [&](int c) {
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        if (DX[b] == 1) rd[b][c] = base[b][c];                         // pw == 1: the centre tap always exists
        else rd[b][c] = (unsigned)(ow[c] - g.pw + DX[b]) < (unsigned)g.W ? base[b][c] : (base[b][c] & (XCH * 64 - 1)) + ZROWS;      // same (unit, row & 3, chan): same banks
      }
    };
    auto advance = [&](int c) {
      ow[c] += step_ow;
      if (ow[c] >= g.W) ow[c] -= g.W;
    };
    half8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
    auto read_frags = [&](auto SET, int stage, int ks) {
      constexpr int B = decltype(SET)::value;
      const char* st = smem + stage * STAGE;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        fah[B][a] = wd_frag<ACH>(st, offA, ks * 16, a * 32);
        if (!LP) fal[B][a] = wd_frag<ACH>(st + A_LO, offA, ks * 16, a * 32);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const short4v h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(st + wbase + rd[b][2 * ks]));
        const short4v h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(st + wbase + rd[b][2 * ks + 1]));
        fbh[B][b] = __builtin_bit_cast(half8, (short8v)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
        if (!LP) {
          const short4v l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(st + wbase + B_PLANE + rd[b][2 * ks]));
          const short4v l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(st + wbase + B_PLANE + rd[b][2 * ks + 1]));
          fbl[B][b] = __builtin_bit_cast(half8, (short8v)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      }
    };
    f32x16 acc[TM][TN];
    auto mfma_set = [&](auto SET) {          // row operand = x fragment (column tile), column operand = dy fragment (k): acc is [r][k]
      constexpr int B = decltype(SET)::value;
      if constexpr (!LP) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fbh[B][b], fal[B][a], acc[a][b]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fbl[B][b], fah[B][a], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<LP>(fbh[B][b], fah[B][a], acc[a][b]);
    };
    constexpr int NRD = NPL * 2 * (TM + TN), NMF = (LP ? 1 : 3) * TM * TN, RPM = (NRD + NMF - 1) / NMF;
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i * RPM < NRD) __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
      }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int stage = 0;
    for (int t = 0; t < my_items; ++t) {
      int tile_k, tile_c, tps[2], sps[2];
      decode_item(t, tile_k, tile_c, tps, sps);
      const int tap = wslot ? tps[1] : tps[0], split = wslot ? sps[1] : sps[0];      // this wave pair's window
      const int pbeg = split * wp.pix_per_split;
#pragma unroll
      for (int c = 0; c < 4; ++c) ow[c] = (pbeg + 16 * (c >> 1) + 4 * (c & 1) + pix_l) % g.OW;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
      select(0); select(1);
      lgkm0_barrier();
      read_frags(B0{}, stage, 0);
      advance(0); advance(1);
      for (int step = 0; step + 1 < wp.nsteps; ++step) {
        select(2); select(3);
        read_frags(B1{}, stage, 1);
        advance(2); advance(3);
        mfma_set(B0{});
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        stage = stage + 1 == NS ? 0 : stage + 1;
        select(0); select(1);
        lgkm0_barrier();
        read_frags(B0{}, stage, 0);
        advance(0); advance(1);
        mfma_set(B1{});
        interleave();
        __builtin_amdgcn_sched_barrier(0);
      }
      select(2); select(3);
      read_frags(B1{}, stage, 1);
      mfma_set(B0{});
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      stage = stage + 1 == NS ? 0 : stage + 1;
      mfma_set(B1{});
      __builtin_amdgcn_sched_barrier(0);
      // acc[a][b]: rows = run index dx C + channel (tile b -> DX[b], UH[b]; 8 (e >> 2) + 4 hh + (e & 3) inside the tile), column = k
      if (tap < ntap && split < wp.splits) {
        float* out = ws + ((int64_t)split * ntap + tap) * (int64_t)g.K * p.R;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          const int kk = tile_k * BM + a * 32 + li;
          if (kk >= g.K) continue;
          float* orow = out + (int64_t)kk * p.R;
#pragma unroll
          for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const int rr = DX[b] * g.C + tile_c * CT + UH[b] * 32 + 8 * e4 + 4 * hh;
              *reinterpret_cast<float4*>(orow + rr) = make_float4(acc[a][b][4 * e4] * inv, acc[a][b][4 * e4 + 1] * inv, acc[a][b][4 * e4 + 2] * inv, acc[a][b][4 * e4 + 3] * inv);
            }
          }
        }
      }
    }
  };
  if (whalf == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// ====================================================================================================================
// Seven-wide window variant: the 7x7x7 stem of the smoke U-Net (stride 1, output grid == input grid, kw == 7, pw == 3, 48 plane
// channels, K <= 64). Ablation of conv_wgrad_h3d_kernel<64,192> on it (tools/bench_conv.py --ablate): 2.24 ms full, 1.64 ms with
// every piece out of range, 1.04 ms without piece instructions -- the run of a tap row is 7 x 48 channels = 672 contiguous bytes
// of x per pixel, so the chunked kernel moves every source pixel seven times (24 GB of LDS-DMA per launch). Here an item is
// (tap row, pixel split) and a 32-pixel step receives
//   * the 32 x 64 dy tile ([4-pixel group][32-channel unit][pixel][64 B], as above), and
//   * ONE window of 32 + 6 consecutive source pixels x 48 channels, kept exactly as it lies in memory (38 rows of 96 B: the DMA
//     lanes fetch 3648 contiguous bytes). The run of output pixel q is then the 336 halves starting at window row q: column tile
//     t (32 run entries, 11 tiles; the last half tile is padding) is a transpose read at byte 96 q + 64 t, whatever dx it starts in.
// A transpose read takes 4 pixels x 32 B per 16 lanes. With rows of 96 B (24 banks) four CONSECUTIVE pixels collide with the other
// 16-channel half of the tile; pixels two apart (192 B = 48 banks) do not, and the MFMA does not care which pixel is which k index as
// long as both operands agree: k index (8 hh + 4 second + j) of a 16-deep half-step is pixel 8 hh + second + 2 j. The dy tile is
// fetched in that order (its groups of four are those pixel sets).
//   * dx validity: per lane and column tile dx = run entry / 48; (pixel, dx) pairs off the image row read a zero slot (the
//     window plane is 4096 B, bytes past 3648 are fetched out of range). (dz, dy) validity: per window row, from the coordinates
//     the producer lane tracks (a row whose pixel is in another image row than the output pixel is masked by the dx test anyway).
//   * compute wave w owns column tiles 3 w .. 3 w + 2 (wave 3: two) x 64 k; 4 DMA instructions per producer wave and step.
#ifndef WDNO_H3S_STAGES
#define WDNO_H3S_STAGES 4          // ring stages (16 KB each in split mode): 4 x 16 KB = the 64 KB a launch gets without a function attribute
#endif
template <bool LP>
__global__ __launch_bounds__(512) void conv_wgrad_h3s_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                              const _Float16* __restrict__ dyh, const _Float16* __restrict__ dyl,
                                                              const float* __restrict__ sx, const float* __restrict__ sdy,
                                                              float* __restrict__ ws, WgradDP wp, unsigned x_bytes, unsigned dy_bytes) {
  constexpr int KW = 7, CX = 48, NS = WDNO_H3S_STAGES, TM = 2;
  constexpr int ROWB = CX * 2, XROWS = 32 + KW - 1, X_DATA = XROWS * ROWB;       // 96, 38, 3648
  constexpr int A_PLANE = 4096, X_PLANE = 4096, ZADDR = X_DATA;
  constexpr int NPL = LP ? 1 : 2;
  constexpr int A_LO = A_PLANE, B_HI = NPL * A_PLANE, B_LO = B_HI + X_PLANE;
  constexpr int STAGE = NPL * (A_PLANE + X_PLANE);
  constexpr int PER = 2 * NPL;                            // DMA instructions per producer wave and step
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const ConvP& p = wp.c;
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  int it_first, it_stride, my_items;                      // item order and XCD ownership: as conv_wgrad_h3d_kernel
  if (wp.xcd_chunk > 0) {
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int beg = xcd * wp.xcd_chunk, end = min(beg + wp.xcd_chunk, wp.items);
    it_stride = (int)gridDim.x >> 3;
    it_first = beg + idx;
    my_items = it_first < end ? (end - it_first + it_stride - 1) / it_stride : 0;
  } else {
    it_first = (int)blockIdx.x; it_stride = (int)gridDim.x;
    my_items = (wp.items - it_first + it_stride - 1) / it_stride;
  }
  const int ntap = g.kd * g.kh;
  const int total = my_items * wp.nsteps;

  if (wave >= 4) {
    // ================================================================== producer waves: lane = 16-byte slot f of every plane
    const int pq = wave - 4;
    int4v rxh = wd_rsrc(xh, x_bytes), rxl = wd_rsrc(xl, x_bytes), rdh = wd_rsrc(dyh, dy_bytes), rdl = wd_rsrc(dyl, dy_bytes);
    asm volatile("s_nop 4" : "+s"(rxh), "+s"(rxl), "+s"(rdh), "+s"(rdl));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const int f = 64 * pq + lane;
    const int xr = f / (ROWB / 16), xc = f - xr * (ROWB / 16);            // window row, chunk of the row
    const bool x_slot = xr < XROWS;
    const int gi = f >> 5, du = (f >> 4) & 1, dj = (f >> 2) & 3, dc = f & 3;
    const int dpix = 16 * (gi >> 2) + 8 * ((gi >> 1) & 1) + (gi & 1) + 2 * dj;   // pixel of the step this dy slot holds
    const bool d_slot = du * 32 + dc * 8 < g.K;
    int it = 0, st = 0;                                   // cursor: item, step
    int q = 0, qw = 0, qh = 0, qd = 0, x_off = 0, dp = 0, d_off = 0, vdz = 0, vdy = 0;
    auto setup_item = [&]() {
      const int id = it_first + it * it_stride;
      const int tap = id % ntap, split = id / ntap;
      const int dz = tap / g.kh, dy = tap - dz * g.kh;
      vdz = dz - g.pd; vdy = dy - g.ph;
      const int pbeg = split * wp.pix_per_split;
      q = pbeg + xr - g.pw;                               // output-grid pixel whose centre tap this window row is
      const int whd = g.W * g.H * g.D;
      int qq = q + whd;                                   // >= 0 (q >= -pw)
      qw = qq % g.W; qq /= g.W;
      qh = qq % g.H; qq /= g.H;
      qd = qq % g.D;
      x_off = ((q + (vdz * g.H + vdy) * g.W) * g.C + xc * 8) * 2;
      dp = pbeg + dpix;
      d_off = (dp * g.K + du * 32 + dc * 8) * 2;
    };
    const bool no_dma = p.debug == 21;                    // ablation (tools/bench_conv.py): every piece out of range
    auto issue_next = [&](int buf) {
      const unsigned sb = lds0 + buf * STAGE + pq * 1024;
      const bool xv = x_slot && q >= 0 && q < (int)p.P && (unsigned)(qd + vdz) < (unsigned)g.D && (unsigned)(qh + vdy) < (unsigned)g.H;
      const int xo = xv && !no_dma ? x_off : WD_OOB;
      const int d_o = d_slot && dp < (int)p.P && !no_dma ? d_off : WD_OOB;
      wd_piece(rdh, d_o, sb);
      if (!LP) wd_piece(rdl, d_o, sb + A_LO);
      wd_piece(rxh, xo, sb + B_HI);
      if (!LP) wd_piece(rxl, xo, sb + B_LO);
      q += 32; x_off += 32 * g.C * 2; dp += 32; d_off += 32 * g.K * 2;
      qw += 32;
      while (qw >= g.W) { qw -= g.W; if (++qh == g.H) { qh = 0; if (++qd == g.D) qd = 0; } }
      if (++st == wp.nsteps) { st = 0; if (++it < my_items) setup_item(); }
    };
    if (total > 0) setup_item();
    int nbuf = 0;
    for (int a = 0; a < NS - 1 && a < total; ++a) { issue_next(nbuf); nbuf = nbuf + 1 == NS ? 0 : nbuf + 1; }
    for (int gs = 0; gs < total; ++gs) {
      // step gs has landed (the NS - 2 steps after it may still be in flight) -> meet the compute waves, which have every fragment of
      // step gs - 1 in registers by now, then refill that step's buffer with step gs + NS - 1
      if (gs + NS - 2 < total) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"((NS - 2) * PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : : : "memory");
      if (gs + NS - 1 < total) { issue_next(nbuf); nbuf = nbuf + 1 == NS ? 0 : nbuf + 1; }
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    return;
  }

  // ================================================================== compute waves
  const int li = lane & 31, hh = lane >> 5, g1 = (lane >> 4) & 1, xq = lane & 15;
  const int fj = xq >> 2, c4 = xq & 3;
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef short short8v __attribute__((ext_vector_type(8)));
  typedef short4v __attribute__((address_space(3))) * lds_s4;
  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sdy[0]);
  const int step_ow = 32 % g.W;
  // dy fragment of (ks, second, a): group 4 ks + 2 hh + second, unit a, pixel slot fj, bytes 32 g1 + 8 c4 of the unit
  const int a_base = hh * 1024 + fj * 64 + g1 * 32 + c4 * 8;

  auto run = [&](auto TNC) {
    constexpr int TN = decltype(TNC)::value;
    const int tb0 = 3 * wave;
    int xbase[TN][4], rd[TN][4], dlo[TN], ow[4];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int r = 32 * (tb0 + b) + 16 * g1 + 4 * c4;            // run entry of this lane's four source halves
      const int dx = r / CX;
      dlo[b] = dx < KW ? g.pw - dx : (1 << 20);                   // valid iff (unsigned)(ow - dlo) < W
#pragma unroll
      for (int c = 0; c < 4; ++c) xbase[b][c] = (16 * (c >> 1) + 8 * hh + (c & 1) + 2 * fj) * ROWB + 2 * r;
    }
    auto select = [&](int c) {
#pragma unroll
      // (round 6: an invalid (pixel, dx) reads zeros in the BANK SLOT of the address it replaces -- bytes 3840 + (address mod 256) of the window plane
      // lie in the 448 out-of-range bytes behind the 38 rows -- instead of the one stand-in address ZADDR, which collided with every lane whose own
      // address fell on its banks: 1.2e7 SQ_LDS_BANK_CONFLICT per dispatch, 8 % of the kernel's LDS cycles, profiles/r06_smoke_pmc.md)
      for (int b = 0; b < TN; ++b) rd[b][c] = (unsigned)(ow[c] - dlo[b]) < (unsigned)g.W ? xbase[b][c] : 3840 + (xbase[b][c] & 255);
    };
    auto advance = [&](int c) {
      ow[c] += step_ow;
      if (ow[c] >= g.W) ow[c] -= g.W;
    };
    half8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
    auto tr2 = [&](const char* a0, const char* a1) {
      const short4v h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a0);
      const short4v h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a1);
      return __builtin_bit_cast(half8, (short8v)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto read_frags = [&](auto SET, int stage, int ks) {
      constexpr int B = decltype(SET)::value;
      const char* st = smem + stage * STAGE;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const char* pa = st + a_base + ks * 2048 + a * 256;
        fah[B][a] = tr2(pa, pa + 512);
        if (!LP) fal[B][a] = tr2(pa + A_LO, pa + A_LO + 512);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        fbh[B][b] = tr2(st + B_HI + rd[b][2 * ks], st + B_HI + rd[b][2 * ks + 1]);
        if (!LP) fbl[B][b] = tr2(st + B_LO + rd[b][2 * ks], st + B_LO + rd[b][2 * ks + 1]);
      }
    };
    f32x16 acc[TM][TN];
    auto mfma_set = [&](auto SET) {          // row operand = x fragment (column tile), column operand = dy fragment (k): acc is [r][k]
      constexpr int B = decltype(SET)::value;
      if constexpr (!LP) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fbh[B][b], fal[B][a], acc[a][b]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fbl[B][b], fah[B][a], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<LP>(fbh[B][b], fah[B][a], acc[a][b]);
    };
    constexpr int NRD = NPL * 2 * (TM + TN), NMF = (LP ? 1 : 3) * TM * TN, RPM = (NRD + NMF - 1) / NMF;
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i * RPM < NRD) __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
      }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int stage = 0;
    for (int t = 0; t < my_items; ++t) {
      const int id = it_first + t * it_stride;
      const int tap = id % ntap, split = id / ntap;
      const int pbeg = split * wp.pix_per_split;
#pragma unroll
      for (int c = 0; c < 4; ++c) ow[c] = (pbeg + 16 * (c >> 1) + 8 * hh + (c & 1) + 2 * fj) % g.OW;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
      select(0); select(1);
      lgkm0_barrier();
      read_frags(B0{}, stage, 0);
      advance(0); advance(1);
      for (int step = 0; step + 1 < wp.nsteps; ++step) {
        select(2); select(3);
        read_frags(B1{}, stage, 1);
        advance(2); advance(3);
        mfma_set(B0{});
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        stage = stage + 1 == NS ? 0 : stage + 1;
        select(0); select(1);
        lgkm0_barrier();
        read_frags(B0{}, stage, 0);
        advance(0); advance(1);
        mfma_set(B1{});
        interleave();
        __builtin_amdgcn_sched_barrier(0);
      }
      select(2); select(3);
      read_frags(B1{}, stage, 1);
      mfma_set(B0{});
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      stage = stage + 1 == NS ? 0 : stage + 1;
      mfma_set(B1{});
      __builtin_amdgcn_sched_barrier(0);
      // acc[a][b]: rows = run entries 32 (tb0 + b) + 8 (e >> 2) + 4 hh + (e & 3), column = k
      float* out = ws + ((int64_t)split * ntap + tap) * (int64_t)g.K * p.R;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int kk = a * 32 + li;
        if (kk >= g.K) continue;
        float* orow = out + (int64_t)kk * p.R;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const int rr = 32 * (tb0 + b) + 8 * e4 + 4 * hh;
            if (rr < p.R)
              *reinterpret_cast<float4*>(orow + rr) = make_float4(acc[a][b][4 * e4] * inv, acc[a][b][4 * e4 + 1] * inv, acc[a][b][4 * e4 + 2] * inv, acc[a][b][4 * e4 + 3] * inv);
          }
        }
      }
    }
  };
  if (wave < 3) run(std::integral_constant<int, 3>{});
  else run(std::integral_constant<int, 2>{});
}

static int wd_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

// geometries of the window kernel: stride 1, equal grids, kw == 3, whole 64-channel tiles (debug 8: never; debug 11: only for
// K <= 64 -- with more output channels the windows are fetched once per 64-wide k tile)
static bool wd_window_takes(const wdno_conv_geom* g) {
  if (wdno_debug_mode == 8 || (wdno_debug_mode == 11 && g->K > 64)) return false;
  return g->sd == 1 && g->sh == 1 && g->sw == 1 && g->OD == g->D && g->OH == g->H && g->OW == g->W && g->kw == 3 && g->pw == 1 &&
         (g->C % 64) == 0 && (g->K % 8) == 0;
}
// geometries of the seven-wide window kernel (debug 8 / 28: the chunked kernel instead)
static bool wd_stem_takes(const wdno_conv_geom* g) {
  if (wdno_debug_mode == 8 || wdno_debug_mode == 28) return false;
  return g->sd == 1 && g->sh == 1 && g->sw == 1 && g->OD == g->D && g->OH == g->H && g->OW == g->W && g->kw == 7 && g->pw == 3 &&
         g->C == 48 && (g->K % 8) == 0 && g->K <= 64 && g->kd <= 8 && g->kh <= 8;
}
// plan shared by the workspace query and the launch (conv_h3.hip calls both)
void wdno_wgrad_h3d_plan(const wdno_conv_geom* g, int* bm, int* bn, int* splits, int* pix_per_split, int* splitpair) {
  ConvP c;
  fill_params(c, g);
  if (splitpair) *splitpair = 0;
  if (wd_stem_takes(g)) {                                   // items = tap rows x splits, one round of the CUs
    *bm = 64; *bn = 352;
    int64_t want = 256 / (g->kd * g->kh);
    int64_t max_splits = cdiv64(c.P, 16 * 32);
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int64_t pps = cdiv64(cdiv64(c.P, want), 32) * 32;
    *pix_per_split = (int)pps;
    *splits = (int)cdiv64(c.P, pps);
    return;
  }
  if (wd_window_takes(g)) {                                 // items = k tiles x tap-row pairs x channel tiles x splits, one round of the CUs
    *bm = 64; *bn = 192;
    const int ntap = g->kd * g->kh, tkc = cdiv(g->K, 64) * (g->C / 64);
    const int tiles = tkc * ((ntap + 1) / 2);
    int64_t want = 256 / tiles;
    int64_t max_splits = cdiv64(c.P, 16 * 32);
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int64_t pps = cdiv64(cdiv64(c.P, want), 32) * 32;
    *pix_per_split = (int)pps;
    *splits = (int)cdiv64(c.P, pps);
    if (splitpair) *splitpair = 0;
    // Split-pair mode (conv_wgrad_h3w_kernel): with an odd number of tap rows the last one is paired across two pixel splits instead of with an
    // empty window: tkc * (ntap / 2) * S + tkc * ceil(S / 2) items of equal length for S splits. Taken when one round of the CUs still holds
    // them (a block then sees items of one kind only) and the items get SHORTER than in the plan above (debug 70: never -- the A/B).
    if (splitpair && (ntap & 1) && wdno_debug_mode != 70) {
      const int blocks = wd_num_cus() >= 256 ? 256 : (wd_num_cus() & ~7);      // what launch_ww starts at most (it walks items <= blocks one per block)
      int64_t S = max_splits < 256 ? max_splits : 256;
      while (S >= 2 && (int64_t)tkc * (ntap / 2) * S + (int64_t)tkc * ((S + 1) / 2) > blocks) --S;
      if (S >= 2) {
        const int64_t pps2 = cdiv64(cdiv64(c.P, S), 32) * 32;
        const int64_t s2 = cdiv64(c.P, pps2);
        if (s2 >= 2 && pps2 < pps) { *pix_per_split = (int)pps2; *splits = (int)s2; *splitpair = 1; }
      }
    }
    return;
  }
  *bm = g->K > 64 ? 128 : 64;
  // column tile of the kw*C run: the width that pads the run least; 192 on a tie (more MFMAs per step and barrier)
  *bn = cdiv(c.R, 192) * 192 <= cdiv(c.R, 128) * 128 ? 192 : 128;
  const int tiles = cdiv(g->K, *bm) * cdiv(c.R, *bn) * g->kd * g->kh;
  int64_t want = 512 / tiles;                              // two items per persistent block (one item of twice the length measures the same)
  int64_t max_splits = cdiv64(c.P, 16 * 32);               // at least 16 steps per item
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  int64_t pps = cdiv64(cdiv64(c.P, want), 32) * 32;
  *pix_per_split = (int)pps;
  *splits = (int)cdiv64(c.P, pps);
}

template <int BM, int BN, bool LP>
static void launch_wd(const void* xh, const void* xl, const void* dyh, const void* dyl, const float* sx, const float* sdy,
                      const void* table, float* wsf, const WgradDP& w, hipStream_t st) {
  const wdno_conv_geom& g = w.c.g;
  const unsigned x_bytes = (unsigned)((int64_t)g.N * g.D * g.H * g.W * g.C * 2);
  const unsigned dy_bytes = (unsigned)((int64_t)g.N * g.YD * g.YH * g.YW * g.K * 2);
  const unsigned tbl_bytes = (unsigned)(w.c.P * 16);
  constexpr int NS = 3;
  const size_t lds = (size_t)NS * (LP ? 1 : 2) * (32 * (BM / 8) * 16 + 32 * (BN / 8) * 16);
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute((const void*)conv_wgrad_h3d_kernel<BM, BN, NS, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  int grid = wd_num_cus();
  if (w.items < grid) grid = w.items;
  WgradDP wl = w;
  wl.xcd_chunk = 0;
  if (grid >= 64 && wdno_debug_mode != 6) {          // debug 6: round-robin items (the A/B for the XCD grouping)
    wl.xcd_chunk = cdiv(w.items, 8);
    grid = wd_num_cus() & ~7;
    if (8 * wl.xcd_chunk < grid) grid = 8 * wl.xcd_chunk;      // every XCD's blocks walk its chunk in whole rounds; spare blocks idle
  }
  conv_wgrad_h3d_kernel<BM, BN, NS, LP><<<grid, 512, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)dyh, (const _Float16*)dyl,
                                                         sx, sdy, (const int4v*)table, wsf, wl, x_bytes, dy_bytes, tbl_bytes);
}

template <bool LP>
static void launch_ww(const void* xh, const void* xl, const void* dyh, const void* dyl, const float* sx, const float* sdy,
                      const void* table, float* wsf, const WgradDP& w, hipStream_t st) {
  const wdno_conv_geom& g = w.c.g;
  const unsigned x_bytes = (unsigned)((int64_t)g.N * g.D * g.H * g.W * g.C * 2);
  const unsigned dy_bytes = (unsigned)((int64_t)g.N * g.YD * g.YH * g.YW * g.K * 2);
  const unsigned tbl_bytes = (unsigned)(w.c.P * 16);
  constexpr int NS = 3;
  const size_t lds = (size_t)NS * (LP ? 1 : 2) * (2 * 4096 + 2 * 5120);      // two dy tiles + two windows per plane set and stage
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute((const void*)conv_wgrad_h3w_kernel<NS, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  int grid = wd_num_cus();
  if (w.items < grid) grid = w.items;
  WgradDP wl = w;
  wl.xcd_chunk = 0;
  if (grid >= 64 && wdno_debug_mode != 6) {
    wl.xcd_chunk = cdiv(w.items, 8);
    grid = wd_num_cus() & ~7;
    if (8 * wl.xcd_chunk < grid) grid = 8 * wl.xcd_chunk;
  }
  conv_wgrad_h3w_kernel<NS, LP><<<grid, 512, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)dyh, (const _Float16*)dyl,
                                                       sx, sdy, (const int4v*)table, wsf, wl, x_bytes, dy_bytes, tbl_bytes);
}

template <bool LP>
static void launch_ws(const void* xh, const void* xl, const void* dyh, const void* dyl, const float* sx, const float* sdy,
                      float* wsf, const WgradDP& w, hipStream_t st) {
  const wdno_conv_geom& g = w.c.g;
  const unsigned x_bytes = (unsigned)((int64_t)g.N * g.D * g.H * g.W * g.C * 2);
  const unsigned dy_bytes = (unsigned)((int64_t)g.N * g.YD * g.YH * g.YW * g.K * 2);
  const size_t lds = (size_t)WDNO_H3S_STAGES * (LP ? 1 : 2) * (4096 + 4096);
  int grid = wd_num_cus();
  if (w.items < grid) grid = w.items;
  WgradDP wl = w;
  wl.xcd_chunk = 0;
  if (grid >= 64 && wdno_debug_mode != 6) {
    wl.xcd_chunk = cdiv(w.items, 8);
    grid = wd_num_cus() & ~7;
    if (8 * wl.xcd_chunk < grid) grid = 8 * wl.xcd_chunk;
  }
  conv_wgrad_h3s_kernel<LP><<<grid, 512, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)dyh, (const _Float16*)dyl,
                                                   sx, sdy, wsf, wl, x_bytes, dy_bytes);
}

// wsf: split workspace [splits][ntap][K][R] (or dwp itself when splits == 1). Returns WDNO_EUNSUPPORTED for geometries the
// DMA kernel does not take.
int wdno_conv_wgrad_h3_dma(const void* xh, const void* xl, const void* dyh, const void* dyl, const float* sx, const float* sdy,
                           const void* table, float* wsf, const wdno_conv_geom* g, hipStream_t st) {
  WgradDP w;
  fill_params(w.c, g);
  int bm, bn;
  wdno_wgrad_h3d_plan(g, &bm, &bn, &w.splits, &w.pix_per_split, &w.splitpair);
  w.n_a = 0x7fffffff;
  if (w.c.P * 16 >= WD_OOB) return WDNO_EUNSUPPORTED;
  w.tiles_k = cdiv(g->K, bm);
  w.tiles_r = cdiv(w.c.R, bn);
  w.nsteps = w.pix_per_split / 32;
  w.items = w.tiles_k * w.tiles_r * g->kd * g->kh * w.splits;
  if (wd_stem_takes(g)) {
    if ((int64_t)g->N * g->D * g->H * g->W * g->C * 2 >= WD_OOB || w.c.P * g->K * 2 >= WD_OOB) return WDNO_EUNSUPPORTED;
    w.tiles_k = w.tiles_r = 1;
    w.items = g->kd * g->kh * w.splits;
    if (xl == nullptr) launch_ws<true>(xh, xh, dyh, dyh, sx, sdy, wsf, w, st);
    else launch_ws<false>(xh, xl, dyh, dyl, sx, sdy, wsf, w, st);
    return WDNO_OK;
  }
  if (wd_window_takes(g)) {      // tiles_r = C / 64 channel tiles, tap rows in pairs
    w.items = w.tiles_k * w.tiles_r * ((g->kd * g->kh + 1) / 2) * w.splits;
    if (w.splitpair) {           // ordinary items of the even tap rows, then the last tap row over pairs of splits
      w.n_a = w.tiles_k * w.tiles_r * ((g->kd * g->kh) / 2) * w.splits;
      w.items = w.n_a + w.tiles_k * w.tiles_r * ((w.splits + 1) / 2);
    }
    if (xl == nullptr) launch_ww<true>(xh, xh, dyh, dyh, sx, sdy, table, wsf, w, st);
    else launch_ww<false>(xh, xl, dyh, dyl, sx, sdy, table, wsf, w, st);
    return WDNO_OK;
  }
  if (xl == nullptr) {           // single bf16 plane per operand
    if (bm == 128 && bn == 192) launch_wd<128, 192, true>(xh, xh, dyh, dyh, sx, sdy, table, wsf, w, st);
    else if (bm == 128) launch_wd<128, 128, true>(xh, xh, dyh, dyh, sx, sdy, table, wsf, w, st);
    else if (bn == 192) launch_wd<64, 192, true>(xh, xh, dyh, dyh, sx, sdy, table, wsf, w, st);
    else launch_wd<64, 128, true>(xh, xh, dyh, dyh, sx, sdy, table, wsf, w, st);
    return WDNO_OK;
  }
  if (bm == 128 && bn == 192) launch_wd<128, 192, false>(xh, xl, dyh, dyl, sx, sdy, table, wsf, w, st);
  else if (bm == 128) launch_wd<128, 128, false>(xh, xl, dyh, dyl, sx, sdy, table, wsf, w, st);
  else if (bn == 192) launch_wd<64, 192, false>(xh, xl, dyh, dyl, sx, sdy, table, wsf, w, st);
  else launch_wd<64, 128, false>(xh, xl, dyh, dyl, sx, sdy, table, wsf, w, st);
  return WDNO_OK;
}
