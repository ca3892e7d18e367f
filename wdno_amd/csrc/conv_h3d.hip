// conv_h3d.hip -- implicit-GEMM convolution, 3 x fp16 split, operands streamed by LDS-DMA (gfx950).
//
// Same arithmetic and interface as conv_fwd_h3_kernel (conv_h3.hip); what differs is how the tiles travel:
//   * `buffer_load_dwordx4 ... lds`: global -> LDS without staging registers or ds_write. One wave instruction lands
//     64 x 16 B = 1 KiB contiguously at M0 + lane*16 (tools/probes/dma_probe.hip: out-of-range lanes write zeros, which
//     is how zero padding, ragged tile tails and the pipeline tail are expressed -- no branches).
//   * LDS rows are 64 B (one 32-deep reduction step of one fp16 plane), un-padded because the DMA image is lane-linear;
//     bank conflicts are avoided by an XOR swizzle applied on the *source* side: lane L of a piece (16 rows x 4 chunks)
//     fetches logical chunk (L & 3) ^ ((L >> 4) & 3) of row L >> 2, and the MFMA fragment reads apply the same XOR.
//   * persistent blocks: one 512-thread block per CU walks over its tiles; the producers run up to NS-1 steps ahead, so
//     the first stages of the next tile are fetched while the compute waves store the previous tile (per-block set-up,
//     pipeline fill and the store burst used to cost ~30 % of a 54-step tile).
//   * NS-stage ring with ONE barrier per step; the loads of the following NS-1 steps stay in flight across the barrier
//     (hand-counted s_waitcnt vmcnt(N): hipcc does not see these loads). Every step issues the same number of pieces
//     (all-invalid past the end), so the count is a compile-time constant.
//   * wave specialisation: waves 4-7 of the 512-thread block are producers (they only compute addresses and issue the
//     pieces: ~6 instructions and ~30 issue cycles per piece, and a VMEM issue blocks the in-order issue port while the
//     texture path is busy -- either would stall the MFMA stream of a compute wave); waves 0-3 only read fragments and
//     issue MFMAs. One producer and one compute wave share each SIMD.
//   * wave tile 64 x 64 (12 MFMAs per 8 ds_read_b128) instead of 32 x 64 / 64 x 64 mixed: half the LDS read traffic per
//     MFMA of the register-staged kernel on the K <= 64 layers.
// Per-row validity is a bit mask (bit dz | bit 8+dy | bit 16+dx set when that tap coordinate is inside the image),
// tested against a per-step `need` word: 3 VALU per piece instead of a compare chain.
#include "conv_common.h"
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
#define DMA_OOB 0x7ffffff0
// LP ("low precision", BASELINE.json configs[1]): ONE bf16 plane per operand and one v_mfma_f32_32x32x16_bf16 per fragment pair
// instead of the (hi, lo) fp16 planes and three products: a third of the matrix instructions, half of the DMA pieces and of
// the LDS traffic; fp32 accumulators, bias / residual / output stay fp32; no scale (bf16 has the fp32 exponent range).
template <bool LP>
__device__ __forceinline__ f32x16 mfma_16(half8 a, half8 b, f32x16 c) {
  if constexpr (LP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int4v rsrc_words(const void* ptr, unsigned bytes) {
  uint64_t a = reinterpret_cast<uint64_t>(ptr);
  int4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
// one 1-KiB piece: lane l's 16 bytes at buffer offset `off` land at LDS byte address lds_dst + 16*l
__device__ __forceinline__ void dma_piece(int4v rsrc, int off, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(off), "s"(lds_dst), "s"(rsrc) : "memory");
}

template <int BM, int BN, int WM, int WN, int NS, bool UNIFORM_DX, bool LP>
__global__ __launch_bounds__(512) void conv_fwd_h3d_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                            const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                            const float* __restrict__ sx, const float* __restrict__ sw,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ y, ConvP p, unsigned x_bytes, unsigned w_bytes) {
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int AP = BM / 64, BP = BN / 64;              // pieces per producer wave and plane (a piece = 16 rows x 64 B)
  constexpr int NPL = LP ? 1 : 2;                        // planes per operand
  constexpr int PW = NPL * (AP + BP);                    // pieces per producer wave and step
  static_assert((NS - 2) * PW <= 63, "vmcnt is a 6-bit counter");
  constexpr int A_LO = BM * 64, B_HI = NPL * BM * 64;
  constexpr int STAGE = NPL * (BM + BN) * 64;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // tiles of this block: blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8 or equals ntiles: the XCD a tile runs on
  // is the one xcd_swizzle assumes)
  const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  if (wave >= WM * WN) {
    // ================================================================== producer waves
    // producer q owns tile rows [q*BM/4, +BM/4) of A and [q*BN/4, +BN/4) of B; lane l of piece i covers row 16*i + (l >> 2)
    const int pq = wave - WM * WN;
    int4v rxh = rsrc_words(xh, x_bytes), rxl = rsrc_words(xl, x_bytes), rwh = rsrc_words(wh, w_bytes), rwl = rsrc_words(wl, w_bytes);
    asm volatile("s_nop 4" : "+s"(rxh), "+s"(rxl), "+s"(rwh), "+s"(rwl));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const int prow = lane >> 2;
    const int c8 = (((lane & 3) ^ ((lane >> 4) & 3))) * 8;      // logical 8-half chunk this lane fetches (source-side swizzle)
    int a_off[AP], b_off[BP];
    unsigned a_mask[AP];
    bool b_ok[BP];
    // inside-the-image bits of the taps t in [0, k) for a coordinate c0 + t in [0, n)
    auto tap_bits = [](int c0, int k, int n) -> unsigned {
      int lo = c0 < 0 ? -c0 : 0, hi = n - c0 < k ? n - c0 : k;
      return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    auto setup_tile = [&](int t) {
      const int tile = xcd_swizzle((int)blockIdx.x + t * (int)gridDim.x, p.ntiles);
      const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
      const int64_t pm0 = (int64_t)tile_m * BM + pq * (BM / 4) + prow;
      int q = pm0 < p.P ? (int)pm0 : 0;
      int ow = q % g.OW; q /= g.OW;
      int oh = q % g.OH; q /= g.OH;
      int od = q % g.OD;
      int n = q / g.OD;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int d0 = od * g.sd - g.pd, h0 = oh * g.sh - g.ph, w0 = ow * g.sw - g.pw;
        a_off[i] = ((((n * g.D + d0) * g.H + h0) * g.W + w0) * g.C + c8) * 2;      // byte offset of (tap 0, dx 0); used only when valid
        const unsigned m = tap_bits(d0, g.kd, g.D) | (tap_bits(h0, g.kh, g.H) << 8) | (tap_bits(w0, g.kw, g.W) << 16);
        a_mask[i] = (pm0 + 16 * i < p.P) ? m : 0u;
        ow += 16;                                                                    // next piece: 16 output pixels further
        while (ow >= g.OW) { ow -= g.OW; ++oh; }
        while (oh >= g.OH) { oh -= g.OH; ++od; }
        while (od >= g.OD) { od -= g.OD; ++n; }
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const int k = tile_n * BN + pq * (BN / 4) + 16 * i + prow;
        b_ok[i] = k < g.K;
        b_off[i] = (k * p.R + c8) * 2;
      }
    };
    // issue cursor: tile, (dz, dy) tap row, 32-wide chunk inside the kw*C run; runs NS-1 steps ahead of the barrier index
    int c_tile = 0, s_chunk = 0, s_dz = 0, s_dy = 0, s_dx = 0, s_cc = 0;
    int x_uni = 0, w_uni = 0;                                      // uniform byte offsets of (tap, chunk) in x and in the packed weights
    int l_dx = c8 / g.C, l_cc = c8 % g.C;                           // per-lane position inside the run (general case)
    const int l_dx0 = l_dx, l_cc0 = l_cc;
    const int step_dx = 32 / g.C, step_cc = 32 % g.C;
    if (my_tiles > 0) setup_tile(0);
    const bool no_dma = p.debug == 21 || p.debug >= 100;             // ablation (tools/bench_conv.py): compute waves alone, on stale LDS contents
    auto issue_step = [&](int stage) {
      const bool live = c_tile < my_tiles;
      const unsigned need_t = live ? ((1u << s_dz) | (1u << (8 + s_dy))) : 0x80000000u;      // bit 31 is never set in a mask
      const unsigned need = UNIFORM_DX ? (need_t | (1u << (16 + s_dx))) : (need_t | (1u << (16 + l_dx)));
      const unsigned pa_dst = lds0 + stage * STAGE + pq * (BM / 4) * 64, pb_dst = lds0 + stage * STAGE + B_HI + pq * (BN / 4) * 64;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int off = ((a_mask[i] & need) == need) ? a_off[i] + x_uni : DMA_OOB;
        if (no_dma) continue;
        dma_piece(rxh, off, pa_dst + i * 1024);
        if (!LP) dma_piece(rxl, off, pa_dst + A_LO + i * 1024);
      }
      const bool r_ok = live && (s_chunk * 32 + c8 < p.R);
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const int off = (b_ok[i] && r_ok) ? b_off[i] + w_uni : DMA_OOB;
        if (no_dma) continue;
        dma_piece(rwh, off, pb_dst + i * 1024);
        if (!LP) dma_piece(rwl, off, pb_dst + BN * 64 + i * 1024);
      }
      if (!live) return;
      ++s_chunk;
      x_uni += 64; w_uni += 64;
      if (UNIFORM_DX) {
        s_cc += 32;
        if (s_cc >= g.C) { s_cc = 0; ++s_dx; }
      } else {
        l_dx += step_dx; l_cc += step_cc;
        if (l_cc >= g.C) { l_cc -= g.C; ++l_dx; }
      }
      if (s_chunk == p.nchunk) {
        s_chunk = 0; s_dx = 0; s_cc = 0; l_dx = l_dx0; l_cc = l_cc0;
        if (++s_dy == g.kh) { s_dy = 0; ++s_dz; }
        if (s_dz == g.kd) {                                      // tile finished: move the cursor to the next one
          s_dz = 0;
          if (++c_tile < my_tiles) setup_tile(c_tile);
        }
        x_uni = (s_dz * g.H + s_dy) * g.W * g.C * 2;
        w_uni = (s_dz * g.kh + s_dy) * g.K * p.R * 2;
      }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_step(s);
    int stage = 0;
    const int total = my_tiles * p.nsteps;
    for (int gs = 0; gs < total; ++gs) {
      // the pieces of this step are older than the (NS-2)*PW issued after them; once they have landed, meet the compute waves
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"((NS - 2) * PW) : "memory");
      int nstage = stage + NS - 1;
      if (nstage >= NS) nstage -= NS;
      issue_step(nstage);                        // the stage everybody finished reading before this barrier
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");      // drain the all-invalid tail pieces before the LDS is released
    return;
  }

  // ================================================================== compute waves
  float am = 0.f;                                                // max |y| this lane has stored (amax record of the output)
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, hh = lane >> 5;
  const int m_base = wm * (TM * 32), n_base = wn * (TN * 32);
  const int sw0 = (hh ^ ((li >> 2) & 3)) * 16;                   // swizzled byte position of logical chunk hh (ks = 0); ks = 1 is sw0 ^ 32
  const int a_rd = (m_base + li) * 64 + sw0;
  const int b_rd = B_HI + (n_base + li) * 64 + sw0;
  // Software pipeline at 16-deep granularity with two fragment sets. While the MFMAs of (step, ks0) run, the fragments of
  // (step, ks1) are in flight; while those of (step, ks1) run, the barrier of step+1 is passed and the fragments of
  // (step+1, ks0) are in flight. lgkmcnt(0) ahead of the barrier guarantees this wave has finished reading the stage the
  // producers overwrite next.
  half8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];      // [set][tile]
  auto read_frags = [&](auto SET, int stage, int ks) {
    constexpr int B = decltype(SET)::value;
    const char* st = smem + stage * STAGE;
    const int x = ks * 32;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      fah[B][a] = *reinterpret_cast<const half8*>(st + ((a_rd + a * 2048) ^ x));
      if (!LP) fal[B][a] = *reinterpret_cast<const half8*>(st + A_LO + ((a_rd + a * 2048) ^ x));
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      fbh[B][b] = *reinterpret_cast<const half8*>(st + ((b_rd + b * 2048) ^ x));
      if (!LP) fbl[B][b] = *reinterpret_cast<const half8*>(st + BN * 64 + ((b_rd + b * 2048) ^ x));
    }
  };
  f32x16 acc[TM][TN];
  auto mfma_set = [&](auto SET) {
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
  // the fragment reads of the next half-step are dealt out behind the MFMAs of this one (conv_h3t.hip has the measurement)
  constexpr int NRD = NPL * (TM + TN), NMF = (LP ? 1 : 3) * TM * TN, RPM = (NRD + NMF - 1) / NMF;
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i * RPM < NRD) __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
    }
  };
  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sw[0]);
  int stage = 0;
  // debug 23 / 24 / 26 (tools/bench_conv.py --stamps): the amax record carries this wave's shader cycles in the kernel / in
  // the tile epilogues / its 100 MHz wall ticks instead of max|y|
  const int sdbg = p.debug >= 100 ? p.debug - 100 : p.debug;      // 1xx = the same stamp with the DMA issue switched off (21)
  const bool stamps = sdbg == 23 || sdbg == 24 || sdbg == 26;
  const uint64_t c_begin = stamps ? __builtin_amdgcn_s_memtime() : 0, r_begin = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t c_epi = 0;
  for (int t = 0; t < my_tiles; ++t) {
    const int tile = xcd_swizzle((int)blockIdx.x + t * (int)gridDim.x, p.ntiles);
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    lgkm0_barrier();
    read_frags(B0{}, stage, 0);
    // The last step is peeled: with the barrier + next-stage reads under a condition, the two paths into the second MFMA
    // group carry different numbers of outstanding LDS reads and the compiler's s_waitcnt has to assume the smaller one --
    // lgkmcnt(3..0), i.e. every step waited for the fragments it had just requested (1398 -> ~1000 shader cycles per step).
    for (int step = 0; step + 1 < p.nsteps; ++step) {
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
    const uint64_t e_begin = stamps ? __builtin_amdgcn_s_memtime() : 0;
    // epilogue of this tile (the producers are already fetching the next one). The MFMAs are issued with the weight
    // fragment as the row operand, so the accumulator tile is [channel][pixel]: lane li owns ONE pixel and holds runs of
    // four consecutive channels -> 16-byte stores and one output-row computation per lane and tile row block.
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int64_t pm = m0 + m_base + a * 32 + li;
      if (pm >= p.P) continue;
      const int64_t yr = p.identity_out ? pm : out_row(g, pm);
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
          if (kc < g.K) {                          // K is a multiple of 4
            float4 v = make_float4(acc[a][b][4 * e4] * inv, acc[a][b][4 * e4 + 1] * inv, acc[a][b][4 * e4 + 2] * inv, acc[a][b][4 * e4 + 3] * inv);
            if (bias) { const float4 t = *reinterpret_cast<const float4*>(bias + kc); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (rrow) { const float4 t = *reinterpret_cast<const float4*>(rrow + kc); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
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

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

template <int BM, int BN, int WM, int WN, int NS, bool LP>
static int launch_h3d(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                      const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  const wdno_conv_geom& g = p.g;
  int64_t tiles_m = cdiv64(p.P, BM);
  p.tiles_n = cdiv(g.K, BN);
  int64_t nt = tiles_m * p.tiles_n;
  if (nt > 0x7fffffff) return WDNO_EUNSUPPORTED;
  p.ntiles = (int)nt;
  const size_t lds = (size_t)NS * (LP ? 1 : 2) * (BM + BN) * 64;
  const int64_t x_elems = (int64_t)g.N * g.D * g.H * g.W * g.C;
  const int64_t w_elems = (int64_t)g.kd * g.kh * g.K * p.R;
  if (x_elems * 2 >= DMA_OOB || w_elems * 2 >= DMA_OOB || p.P >= 0x7fffffff) return WDNO_EUNSUPPORTED;
  int grid = num_cus() & ~7;                      // one persistent block per CU (the LDS ring allows no more), multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  if (p.ntiles < grid) grid = p.ntiles;
  const bool uni = (g.C % 32) == 0;
  if (uni) {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)conv_fwd_h3d_kernel<BM, BN, WM, WN, NS, true, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    conv_fwd_h3d_kernel<BM, BN, WM, WN, NS, true, LP><<<grid, 512, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)wh, (const _Float16*)wl,
                                                                        sx, sw, bias, residual, y, p, (unsigned)(x_elems * 2), (unsigned)(w_elems * 2));
  } else {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)conv_fwd_h3d_kernel<BM, BN, WM, WN, NS, false, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    conv_fwd_h3d_kernel<BM, BN, WM, WN, NS, false, LP><<<grid, 512, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)wh, (const _Float16*)wl,
                                                                         sx, sw, bias, residual, y, p, (unsigned)(x_elems * 2), (unsigned)(w_elems * 2));
  }
  return WDNO_OK;
}

// Returns WDNO_EUNSUPPORTED when the geometry is outside what the DMA kernels handle (the caller then uses the
// register-staged kernels of conv_h3.hip).
template <bool LP>
static int fwd_h3_dma(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                      const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  const wdno_conv_geom& g = p.g;
  if (g.kd > 8 || g.kh > 8 || g.kw > 8 || g.C < 8) return WDNO_EUNSUPPORTED;
  // Tile shape: the one with the least (rounds of the persistent grid) x (tile area), lightly weighted by how much operand
  // traffic a shape needs per MFMA. Examples on 256 CUs: a 256-channel layer at the 10 x 10 level is 300 tiles of 128 x 128 --
  // two rounds, the second with 44 tiles -- but 200 tiles of 192 x 128 (0.258 -> 0.196 ms); a 64-channel layer at the 20 x 20
  // level is 300 tiles of 256 x 64 but 400 shorter ones of 192 x 64 (0.329 -> 0.267 ms). debug 9: the two original shapes only.
  const int cus = num_cus() & ~7;
  auto cost = [&](int bm, int bn, double weight) {
    return (double)(cdiv64(cdiv64(p.P, bm) * cdiv(g.K, bn), cus) * bm * bn) * weight;
  };
  const bool narrow_only = g.K <= 64, all = wdno_debug_mode != 9;
  int best = narrow_only ? 2 : 0;
  double c = narrow_only ? cost(256, 64, 1.04) : cost(128, 128, 1.0);
  if (!narrow_only && all && cost(192, 128, 1.0) < c) { best = 1; c = cost(192, 128, 1.0); }
  if (!narrow_only && all && cost(256, 64, 1.04) < c) { best = 2; c = cost(256, 64, 1.04); }
  if (all && cost(192, 64, 1.08) < c) { best = 3; c = cost(192, 64, 1.08); }
  if (p.identity_out && wdno_conv_h3t_takes(g)) {
    // tap-resident kernel only: two more shapes of 20480 outputs, five MFMA row tiles per wave. On 256 CUs the 20 x 20-level layers with 128
    // output channels are 600 tiles of 128 x 128 (three rounds, the last with 88 tiles) but 480 of 160 x 128 (two rounds, 94 % full), the
    // 10 x 10-level ones with 256 are 200 of 192 x 128 (78 % of one round) but 240 of 160 x 128; 320 x 64 does the same for 64 output
    // channels at the 20 x 20 level (240 tiles instead of 400 of 192 x 64). debug 29: without them.
    if (all && wdno_debug_mode != 29 && g.kw == 3) {
      if (!narrow_only && cost(160, 128, 1.02) < c) { best = 4; c = cost(160, 128, 1.02); }
      if (cost(320, 64, 1.05) < c) { best = 5; c = cost(320, 64, 1.05); }
      // few pixels x many channels (the 16 x 16 and 8 x 8 levels of the Burgers U-Net at batch 16: 176 tiles of 192 x 64, 64 of 128 x 128):
      // 128 x 64 and 64 x 64 tiles fill the chip; their operand traffic per MFMA is higher (weights 1.15 / 1.35). debug 53: without them.
      if (wdno_debug_mode != 53) {
        if (cost(128, 64, 1.15) < c) { best = 7; c = cost(128, 64, 1.15); }
        if (cost(64, 64, 1.35) < c) { best = 6; c = cost(64, 64, 1.35); }
      }
      if (wdno_debug_mode == 54) best = 6;
      if (wdno_debug_mode == 55) best = 7;
      if (wdno_debug_mode == 30) best = narrow_only ? 5 : 4;     // tests: the new shapes on small cases
      if (wdno_debug_mode == 31) best = 5;
      if (wdno_debug_mode == 45 && best == 2 && (g.C % 16) == 0) best = 8;      // experiment: 128 x 64 accumulator tile per wave (conv_h3t.hip)
      // ... but where those small tiles were chosen for a LONG reduction (8 x 8 x 16 samples x 1024 channels: 256 tiles of 64 x 64 with 96 stages
      // each, every stage a 33 KB delivery for 18 matrix instructions per wave -- delivery-bound, 89 vs 46 us with the DMA issue off), 128 x 128
      // tiles whose reduction is cut into four runs fill the chip too, with 66 KB per 72 matrix instructions: 89 -> 61 us (debug 56: not).
      // (Not where 128 x 64 tiles were chosen -- 16 x 16 x 16 samples x 512 channels: 61 us unsplit, 63 us as two runs.) Needs the caller's
      // workspace for the partial sums (wdno_conv_fwd_split_ws_bytes).
      const int split = wdno_conv_h3t_split(g, p.P, cus);
      if (best == 6 && split > 1 && p.split_ws && (size_t)split * p.P * g.K * sizeof(float) <= p.split_ws_bytes) { best = 0; p.tsplit = split; }
    }
    return wdno_conv_fwd_h3_tap(best, xh, LP ? nullptr : xl, wh, wl, sx, sw, bias, residual, y, p, st);
  }
  // (deeper rings -- NS = 4 / 5 fill the 160 KB -- measured no faster: the 7 x 7 x 7 init convolution keeps its 3.1 M shader cycles)
  if (best == 0) return launch_h3d<128, 128, 2, 2, 3, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (best == 1) return launch_h3d<192, 128, 2, 2, 3, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  if (best == 3) return launch_h3d<192, 64, 2, 2, 3, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
  return launch_h3d<256, 64, 4, 1, 3, LP>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
}
// xl / wl / sx / sw == nullptr selects the single-plane bf16 kernels (LP)
int wdno_conv_fwd_h3_dma(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                         const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st, int stages) {
  (void)stages;
  if (xl == nullptr) return fwd_h3_dma<true>(xh, xh, wh, wh, sx, sw, bias, residual, y, p, st);
  return fwd_h3_dma<false>(xh, xl, wh, wl, sx, sw, bias, residual, y, p, st);
}
