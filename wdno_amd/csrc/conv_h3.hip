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
#include <type_traits>
#include <algorithm>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// LP = single bf16 plane per operand, one product (see conv_h3d.hip); !LP = (hi, lo) fp16 planes, three products
template <bool LP>
__device__ __forceinline__ f32x16 mfma_16(half8 a, half8 b, f32x16 c) {
  if constexpr (LP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#define HBK 32   // reduction elements per step
#define HST 40   // LDS row stride in halves

// ---------------------------------------------------------------------------------------------- amax / split
template <bool RECORD>
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
  __shared__ float red[4];
  const int64_t n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float m = 0.f;
  // four independent 16-byte loads in flight per thread: a bandwidth-bound sweep needs ~8 MB outstanding chip-wide
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t k0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; k0 < n4; k0 += stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t k = k0 + u * 256;
      v[u] = k < n4 ? x4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
  }
  if (blockIdx.x == 0)
    for (int64_t k = (n4 << 2) + threadIdx.x; k < n; k += 256) m = fmaxf(m, fabsf(x[k]));
  if (RECORD) {
    amax_record_emit(m, reinterpret_cast<float*>(out), blockIdx.x);
    return;
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
  }
}
extern "C" int wdno_amax_record(const float* x, int64_t n, float* rec_zeroed, wdno_stream_t s) {
  WDNO_REQUIRE(n >= 0);
  if (n == 0) return WDNO_OK;
  amax_kernel<true><<<stream_grid(n / 16 + 1, 256), 256, 0, as_stream(s)>>>(x, n, reinterpret_cast<unsigned*>(rec_zeroed));
  return wdno_check_launch();
}
extern "C" int wdno_amax(const float* x, int64_t n, float* amax_zeroed, wdno_stream_t s) {
  WDNO_REQUIRE(n > 0);
  amax_kernel<false><<<stream_grid(n / 16 + 1, 256), 256, 0, as_stream(s)>>>(x, n, reinterpret_cast<unsigned*>(amax_zeroed));
  return wdno_check_launch();
}

// x [rows][C] fp32 -> hi, lo [rows][C8] fp16 (zero padded), scale_out[0] = s
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, const float* __restrict__ amax,
                                                     _Float16* __restrict__ hi, _Float16* __restrict__ lo, float* __restrict__ scale_out,
                                                     int64_t rows, int C, int C8) {
  const bool lp = lo == nullptr;                             // single bf16 plane, no scale
  const float s = lp ? 1.0f : scale_from_amax(amax_record_read(amax));
  if (!lp && blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
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
    if (lp) {
      typedef unsigned short us8 __attribute__((ext_vector_type(8)));
      us8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = bf16_rne(v[e]);
      *reinterpret_cast<us8*>(hi + r * C8 + c0) = b;
      continue;
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
extern "C" int wdno_cast_bf16(const float* x, void* out, int64_t rows, int C, int C8, wdno_stream_t s) {
  WDNO_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (C8 & 7) == 0 && C8 >= C && C8 < C + 16);      // (up to a whole 16-channel block: 7-wide stems)
  int64_t total = rows * (C8 / 8);
  split_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x, nullptr, (_Float16*)out, nullptr, nullptr, rows, C, C8);
  return wdno_check_launch();
}
extern "C" int wdno_split_f16(const float* x, const float* amax, void* hi, void* lo, float* scale_out, int64_t rows, int C, int C8,
                              wdno_stream_t s) {
  WDNO_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (C8 & 7) == 0 && C8 >= C && C8 < C + 16);      // (up to a whole 16-channel block: 7-wide stems)
  int64_t total = rows * (C8 / 8);
  split_kernel<<<stream_grid(total, 256), 256, 0, as_stream(s)>>>(x, amax, (_Float16*)hi, (_Float16*)lo, scale_out, rows, C, C8);
  return wdno_check_launch();
}

// split + per-channel column sums in one pass (the bias gradient of a convolution is the column sum of the same dy that is
// being split for the data / weight gradient kernels). Every thread always works on the same 8-channel group (the host
// only calls this for power-of-two group counts <= 256), accumulates its rows in fp32 and the block folds them to one
// double row of partials; partial_rows_sum_kernel<double> finishes, as for wdno_colsum.
__global__ __launch_bounds__(256) void split_colsum_kernel(const float* __restrict__ x, const float* __restrict__ amax,
                                                            _Float16* __restrict__ hi, _Float16* __restrict__ lo, float* __restrict__ scale_out,
                                                            double* __restrict__ part, int64_t rows, int C, int C8) {
  __shared__ float red[256][9];
  const bool lp = lo == nullptr;
  const float s = lp ? 1.0f : scale_from_amax(amax_record_read(amax));
  if (!lp && blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
  const int g8 = C8 >> 3;
  const int64_t total = rows * g8;
  const int64_t stride = (int64_t)gridDim.x * 256;            // a multiple of g8: the channel group of a thread never changes
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
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
    if (lp) {
      typedef unsigned short us8 __attribute__((ext_vector_type(8)));
      us8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs[e] += v[e]; b[e] = bf16_rne(v[e]); }
      *reinterpret_cast<us8*>(hi + r * C8 + c0) = b;
      continue;
    }
    half8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cs[e] += v[e];
      float t = v[e] * s;
      _Float16 th = (_Float16)t;
      h[e] = th;
      l[e] = (_Float16)(t - (float)th);
    }
    *reinterpret_cast<half8*>(hi + r * C8 + c0) = h;
    *reinterpret_cast<half8*>(lo + r * C8 + c0) = l;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = cs[e];
  __syncthreads();
  // threads t, t + g8, t + 2 g8, ... share the group t % g8
  for (int idx = threadIdx.x; idx < C8; idx += 256) {
    const int grp = idx >> 3, e = idx & 7;
    double acc = 0.0;
    for (int t = grp; t < 256; t += g8) acc += (double)red[t][e];
    part[(int64_t)blockIdx.x * C8 + idx] = acc;
  }
}
// ws: doubles [wdno_split_colsum_ws_bytes / 8]; colsum_out[C8] fp32. WDNO_EUNSUPPORTED unless C8/8 is a power of two <= 256.
extern "C" size_t wdno_split_colsum_ws_bytes(int64_t rows, int C8) {
  return (size_t)stream_grid(rows * (C8 / 8), 256) * (size_t)C8 * sizeof(double);
}
extern "C" int wdno_split_f16_colsum(const float* x, const float* amax, void* hi, void* lo, float* scale_out, float* colsum_out,
                                     void* ws, size_t ws_bytes, int64_t rows, int C, int C8, wdno_stream_t s) {
  WDNO_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (C8 & 7) == 0 && C8 >= C && C8 < C + 16);      // (up to a whole 16-channel block: 7-wide stems)
  const int g8 = C8 / 8;
  if (g8 > 256 || (g8 & (g8 - 1))) return WDNO_EUNSUPPORTED;
  if (ws_bytes < wdno_split_colsum_ws_bytes(rows, C8)) return WDNO_EWORKSPACE;
  const int grid = stream_grid(rows * g8, 256);
  split_colsum_kernel<<<grid, 256, 0, as_stream(s)>>>(x, amax, (_Float16*)hi, (_Float16*)lo, scale_out, (double*)ws, rows, C, C8);
  if (colsum_out)      // NULL: the caller sums the ws_bytes / (8 C8) rows of partials later (wdno_rows_sum_multi)
    partial_rows_sum_kernel<double><<<cdiv(C8, 32), PRS_THREADS, 0, as_stream(s)>>>((const double*)ws, colsum_out, grid, C8);
  return wdno_check_launch();
}

extern "C" int wdno_cast_bf16_colsum(const float* x, void* out, float* colsum_out, void* ws, size_t ws_bytes, int64_t rows, int C, int C8,
                                     wdno_stream_t s) {
  WDNO_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (C8 & 7) == 0 && C8 >= C && C8 < C + 16);      // (up to a whole 16-channel block: 7-wide stems)
  const int g8 = C8 / 8;
  if (g8 > 256 || (g8 & (g8 - 1))) return WDNO_EUNSUPPORTED;
  if (ws_bytes < wdno_split_colsum_ws_bytes(rows, C8)) return WDNO_EWORKSPACE;
  const int grid = stream_grid(rows * g8, 256);
  split_colsum_kernel<<<grid, 256, 0, as_stream(s)>>>(x, nullptr, (_Float16*)out, nullptr, nullptr, (double*)ws, rows, C, C8);
  if (colsum_out)
    partial_rows_sum_kernel<double><<<cdiv(C8, 32), PRS_THREADS, 0, as_stream(s)>>>((const double*)ws, colsum_out, grid, C8);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- convolution
// The fp16 MFMAs of one 32-deep step take only ~400 cycles per wave, so the per-step integer work matters as much as the
// math. All operand fetches are raw buffer loads with 32-bit byte offsets: out-of-range pieces (zero padding, tile tails)
// are given an offset beyond the descriptor's range and come back as zeros from the hardware bounds check, so there is
// no branch and no select. Offsets are (per-thread constant) + (uniform per tap) + (uniform per chunk).
typedef int int4v __attribute__((ext_vector_type(4)));
#define OOB_OFFSET 0x7ffffff0      // > any plane size accepted by the host wrapper

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), (short)0, (int)bytes, 0x00020000);
}
// The same descriptor as four SGPR words for the hand-counted loads below (raw buffer, stride 0, bounds = bytes).
__device__ __forceinline__ int4v make_rsrc_words(const void* ptr, unsigned bytes) {
  uint64_t a = reinterpret_cast<uint64_t>(ptr);
  int4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
// hipcc turns every counted prefetch of compiler-visible loads into s_waitcnt vmcnt(0) at the next use, which drains the
// queue once per step and leaves the kernel latency bound. These loads are therefore invisible to its bookkeeping: the
// destination counts as written at issue, and the kernel waits for them itself with vm_wait<N>() naming the registers.
__device__ __forceinline__ int4v asm_buffer_load_b128(int4v rsrc, int byte_off) {
  int4v v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(byte_off), "s"(rsrc) : "memory");
  return v;
}

template <int BM, int BN, int WM, int WN, bool LP>
__global__ __launch_bounds__(256) void conv_fwd_h3_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                           const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                           const float* __restrict__ sx, const float* __restrict__ sw,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           float* __restrict__ y, ConvP p, unsigned x_bytes, unsigned w_bytes) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AROWS = BM / 64;   // row passes of the 256 loader threads (64 rows x 4 column groups per pass)
  constexpr int BROWS = BN / 64;
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
  constexpr int NPL = LP ? 1 : 2;                // planes per operand
  constexpr int STAGE = NPL * (BM + BN) * HST;   // [Ah | Al | Bh | Bl]  (LP: [A | B])
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;
  int4v rxh = make_rsrc_words(xh, x_bytes), rxl = make_rsrc_words(xl, x_bytes);
  int4v rwh = make_rsrc_words(wh, w_bytes), rwl = make_rsrc_words(wl, w_bytes);
  asm volatile("s_nop 4" : "+s"(rxh), "+s"(rxl), "+s"(rwh), "+s"(rwl));      // SGPR write -> VMEM descriptor read wait states

  // 4 lanes cover one 64-byte row piece; consecutive lane quads take rows r and r+4 so that the 8 lanes of a
  // ds_write_b128 group hit 32 distinct banks with the 80-byte row stride (r and r+1 would overlap by 16 bytes)
  const int lq = tid >> 2;
  const int lrow = (lq & ~7) | ((lq & 1) << 2) | ((lq >> 1) & 3);          // 0..63, bijective
  const int c8 = (tid & 3) * 8;       // element offset inside the 32-wide chunk
  int a_d0[AROWS], a_h0[AROWS], a_w0[AROWS], a_base[AROWS];
  bool a_ok[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int64_t pm = m0 + lrow + 64 * i;
    a_ok[i] = pm < p.P;
    int q = a_ok[i] ? (int)pm : 0;                 // the host wrapper guarantees P < 2^31
    int ow = q % g.OW; q /= g.OW;
    int oh = q % g.OH; q /= g.OH;
    int od = q % g.OD;
    int n = q / g.OD;
    a_d0[i] = od * g.sd - g.pd;
    a_h0[i] = oh * g.sh - g.ph;
    a_w0[i] = ow * g.sw - g.pw;
    a_base[i] = (((n * g.D + a_d0[i]) * g.H + a_h0[i]) * g.W + a_w0[i]) * g.C + c8;   // element offset of (tap 0, chunk 0)
  }
  int b_base[BROWS];
  bool b_ok[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    int k = n0 + lrow + 64 * i;
    b_ok[i] = k < g.K;
    b_base[i] = k * p.R + c8;
  }

  // Split over the tap rows (p.ksplit = 2, blockIdx.y): a wide layer with few pixels -- 8 x 8 x 16 samples x 1024 channels in the Burgers
  // U-Net -- has fewer tiles than CUs and 288 steps per tile; two blocks per tile each take half of the (dz, dy) rows and add their
  // partial sums with atomics onto a zeroed output (two addends: the order cannot change the result).
  const int ntap = g.kd * g.kh;
  const int t_first = p.ksplit > 1 ? ntap * (int)blockIdx.y / p.ksplit : 0;
  const int t_last = p.ksplit > 1 ? ntap * ((int)blockIdx.y + 1) / p.ksplit : ntap;
  const int nst = (t_last - t_first) * p.nchunk;          // steps of this block
  // load cursor: (tap, chunk) uniform; (dx, cc) per thread because a chunk may straddle pixels when C % 32 != 0
  int l_chunk = 0, l_dz = t_first / g.kh, l_dy = t_first % g.kh, l_r = c8, l_dx = c8 / g.C, l_cc = c8 % g.C;
  const int dx0 = l_dx, cc0 = l_cc;
  int tap_off = (l_dz * g.H + l_dy) * g.W * g.C, wtap_off = t_first * g.K * p.R;      // uniform element offsets of the current tap row in x and in the packed weights
  const int step_dx = HBK / g.C, step_cc = HBK % g.C;
  bool row_ok[AROWS];
  auto refresh_tap = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int d = a_d0[i] + l_dz, h = a_h0[i] + l_dy;
      row_ok[i] = a_ok[i] && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H;
    }
  };
  refresh_tap();

  int4v ah[2][AROWS], al[2][AROWS], bh[2][BROWS], bl[2][BROWS];     // two register sets: loads run 2 steps ahead
  auto load_tile = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    const bool r_ok = l_r < p.R && p.debug != 1;
    const int chunk_off = l_chunk * HBK;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const bool ok = row_ok[i] && r_ok && (unsigned)(a_w0[i] + l_dx) < (unsigned)g.W;
      const int off = ok ? (a_base[i] + tap_off + chunk_off) * 2 : OOB_OFFSET;
      ah[S][i] = asm_buffer_load_b128(rxh, off);
      if (!LP) al[S][i] = asm_buffer_load_b128(rxl, off);
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const bool ok = b_ok[i] && r_ok;
      const int off = ok ? (b_base[i] + wtap_off + chunk_off) * 2 : OOB_OFFSET;
      bh[S][i] = asm_buffer_load_b128(rwh, off);
      if (!LP) bl[S][i] = asm_buffer_load_b128(rwl, off);
    }
    ++l_chunk;
    l_r += HBK;
    l_dx += step_dx;
    l_cc += step_cc;
    if (l_cc >= g.C) { l_cc -= g.C; ++l_dx; }
    if (l_chunk == p.nchunk) {
      l_chunk = 0; l_r = c8; l_dx = dx0; l_cc = cc0;
      wtap_off += g.K * p.R;
      if (++l_dy == g.kh) { l_dy = 0; ++l_dz; }
      tap_off = (l_dz * g.H + l_dy) * g.W * g.C;
      refresh_tap();
    }
  };
  // wait until at most `newer` younger loads are outstanding; names every register of the set so that no consumer of
  // them can be scheduled above the wait
  constexpr int NLOADS = NPL * (AROWS + BROWS);
  auto wait_set = [&](auto SET, bool newer_in_flight) {
    constexpr int S = decltype(SET)::value;
    if (newer_in_flight) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NLOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int4v &r0 = ah[S][i], &r1 = al[S][LP ? 0 : i];
      if (LP) asm volatile("" : "+v"(r0));
      else asm volatile("" : "+v"(r0), "+v"(r1));
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      int4v &r0 = bh[S][i], &r1 = bl[S][LP ? 0 : i];
      if (LP) asm volatile("" : "+v"(r0));
      else asm volatile("" : "+v"(r0), "+v"(r1));
    }
  };
  auto store_tile = [&](auto SET, int buf) {
    constexpr int S = decltype(SET)::value;
    _Float16* Ah = hsm + buf * STAGE;
    _Float16* Al = Ah + BM * HST;
    _Float16* Bh = Ah + NPL * BM * HST;
    _Float16* Bl = Bh + BN * HST;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      *reinterpret_cast<int4v*>(&Ah[(lrow + 64 * i) * HST + c8]) = ah[S][i];
      if (!LP) *reinterpret_cast<int4v*>(&Al[(lrow + 64 * i) * HST + c8]) = al[S][i];
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      *reinterpret_cast<int4v*>(&Bh[(lrow + 64 * i) * HST + c8]) = bh[S][i];
      if (!LP) *reinterpret_cast<int4v*>(&Bl[(lrow + 64 * i) * HST + c8]) = bl[S][i];
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

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // step t lives in register set t & 1 and LDS stage t & 1
  auto iter = [&](auto PAR, int step) {
    constexpr int B = decltype(PAR)::value;
    const _Float16* Ah = hsm + B * STAGE;
    const _Float16* Al = Ah + BM * HST;
    const _Float16* Bh = Ah + NPL * BM * HST;
    const _Float16* Bl = Bh + BN * HST;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 fah[TM], fal[TM], fbh[TN], fbl[TN];
      const int col = ks * 16 + hh * 8;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        fah[a] = *reinterpret_cast<const half8*>(&Ah[(m_base + a * 32 + li) * HST + col]);
        if (!LP) fal[a] = *reinterpret_cast<const half8*>(&Al[(m_base + a * 32 + li) * HST + col]);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        fbh[b] = *reinterpret_cast<const half8*>(&Bh[(n_base + b * 32 + li) * HST + col]);
        if (!LP) fbl[b] = *reinterpret_cast<const half8*>(&Bl[(n_base + b * 32 + li) * HST + col]);
      }
      // the three partial products go to the same accumulator: issue them tile-interleaved so that consecutive MFMAs
      // never depend on each other
      if constexpr (!LP) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fal[a], fbh[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<false>(fah[a], fbl[b], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = mfma_16<LP>(fah[a], fbh[b], acc[a][b]);
    }
    using NXT = std::integral_constant<int, 1 - B>;
    if (step + 1 < nst) {
      wait_set(NXT{}, step + 2 < nst);      // the set of step+2 (issued one iteration later) may stay in flight
      store_tile(NXT{}, 1 - B);
    }
    __syncthreads();
    if (step + 3 < nst) load_tile(NXT{});
  };
  load_tile(S0{});
  wait_set(S0{}, false);
  store_tile(S0{}, 0);
  __syncthreads();
  if (nst > 1) load_tile(S1{});
  if (nst > 2) load_tile(S0{});
  for (int step = 0; step < nst; step += 2) {
    iter(S0{}, step);
    if (step + 1 < nst) iter(S1{}, step + 1);
  }

  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sw[0]);
  float am = 0.f;
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
          if (p.ksplit > 1) {                    // partial sum: bias / residual ride on the first one
            if (blockIdx.y == 0) {
              if (bias) v += bias[kc];
              if (res) v += res[yr * g.K + kc];
            }
            unsafeAtomicAdd(&y[yr * g.K + kc], v);
            continue;
          }
          if (bias) v += bias[kc];
          if (res) v += res[yr * g.K + kc];
          if (LP && p.out_bf16) reinterpret_cast<unsigned short*>(y)[yr * g.K + kc] = bf16_rne(v);
          else y[yr * g.K + kc] = v;
          am = fmaxf(am, fabsf(v));
        }
      }
    }
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * 4 + wave);
}

template <int BM, int BN, int WM, int WN, bool LP>
static int launch_h3(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                     const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st) {
  const wdno_conv_geom& g = p.g;
  int64_t tiles_m = cdiv64(p.P, BM);
  p.tiles_n = cdiv(g.K, BN);
  int64_t nt = tiles_m * p.tiles_n;
  if (nt > 0x7fffffff) return WDNO_EUNSUPPORTED;
  p.ntiles = (int)nt;
  size_t lds = (size_t)2 * (LP ? 1 : 2) * (BM + BN) * HST * sizeof(_Float16);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_h3_kernel<BM, BN, WM, WN, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int64_t x_elems = (int64_t)g.N * g.D * g.H * g.W * g.C;
  const int64_t w_elems = (int64_t)g.kd * g.kh * g.K * p.R;
  if (x_elems * 2 >= OOB_OFFSET || w_elems * 2 >= OOB_OFFSET || p.P >= 0x7fffffff) return WDNO_EUNSUPPORTED;   // 32-bit buffer offsets
  float* rec = p.amax_rec;
  if (p.ksplit > 1) {                     // partial sums are added onto zeros; the amax record is filled by a sweep afterwards
    if (hipMemsetAsync(y, 0, (size_t)p.P * g.K * sizeof(float), st) != hipSuccess) return WDNO_ELAUNCH;
    p.amax_rec = nullptr;
  }
  conv_fwd_h3_kernel<BM, BN, WM, WN, LP><<<dim3((unsigned)p.ntiles, (unsigned)p.ksplit), 256, lds, st>>>(
      (const _Float16*)xh, (const _Float16*)xl, (const _Float16*)wh, (const _Float16*)wl, sx, sw, bias, residual, y, p,
      (unsigned)(x_elems * 2), (unsigned)(w_elems * 2));
  if (p.ksplit > 1 && rec) {
    amax_kernel<true><<<stream_grid(p.P * g.K / 16 + 1, 256), 256, 0, st>>>(y, p.P * g.K, reinterpret_cast<unsigned*>(rec));
    p.amax_rec = rec;
  }
  return WDNO_OK;
}

extern "C" int wdno_conv_fwd_f16x3(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                                   const float* bias, const float* residual, float* y, const wdno_conv_geom* g, wdno_stream_t s) {
  return wdno_conv_fwd_f16x3_amax(xh, xl, sx, wph, wpl, sw, bias, residual, y, nullptr, g, s);
}
template <bool LP>
static int conv_fwd_16(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                       const float* bias, const float* residual, float* y, float* amax_rec, const wdno_conv_geom* g, wdno_stream_t s,
                       float* split_ws = nullptr, size_t split_ws_bytes = 0, const wdno_zero_box* zb = nullptr, int y_bf16 = 0) {
  int rc = check_geom(g);
  if (rc) return rc;
  if (g->C & 7) return WDNO_EUNSUPPORTED;       // 16-bit rows must be 16-byte multiples
  if (y_bf16 && (!LP || residual)) return WDNO_EUNSUPPORTED;      // bf16 output: single-product mode, no residual (its reader is a GroupNorm)
  ConvP p;
  fill_params(p, g);
  p.out_bf16 = y_bf16 ? 1 : 0;
  if (zb) {                                     // a statement about x, used where a kernel can (conv_h3t.hip, 7-wide taps): whole 16-channel blocks
    if (zb->channels < 0 || zb->d0 < 0 || zb->h0 < 0 || zb->w0 < 0) return WDNO_EINVAL;
    p.zb_blocks = zb->channels / 16; p.zb_d = zb->d0; p.zb_h = zb->h0;
  }
  p.amax_rec = amax_rec;
  p.split_ws = split_ws;
  p.split_ws_bytes = split_ws_bytes;
  p.nchunk = cdiv(p.R, HBK);
  p.nsteps = g->kd * g->kh * p.nchunk;
  hipStream_t st = as_stream(s);
  const int64_t P = p.P;
  const int K = g->K;
  auto blocks = [&](int bm, int bn) { return cdiv64(P, bm) * cdiv(K, bn); };
  // problems with at least 100 tiles: persistent LDS-DMA kernels (256x64 / 128x128 tiles, 64x64 per compute wave); even with
  // fewer tiles than CUs they beat the register-staged kernels (l2 512->128: 0.36 -> 0.24 ms). debug: 5 = never, 7 = always, 8 = always and not the tap-resident variant, 30 / 31 = always with the five-row tile shapes (conv_h3d.hip)
  const int dbg = wdno_debug_mode;
  // ... and the tap-resident kernel (stride 1, equal grids, 3-wide) has 64 x 64 / 128 x 64 tiles for problems of fewer tiles (debug 53: not)
  const bool small_tap = p.identity_out && wdno_conv_h3t_takes(*g) && g->kw == 3 && dbg != 53 && blocks(64, 64) >= 64;
  if (dbg != 5 && dbg != 3 && dbg != 1 && (dbg == 7 || dbg == 8 || dbg == 30 || dbg == 31 || dbg == 54 || dbg == 55 || small_tap || (K > 64 ? blocks(128, 128) : blocks(256, 64)) >= 100)) {
    rc = wdno_conv_fwd_h3_dma(xh, LP ? nullptr : xl, wph, LP ? nullptr : wpl, sx, sw, bias, residual, y, p, st, 3);
    if (rc == WDNO_OK) return wdno_check_launch();
    if (rc != WDNO_EUNSUPPORTED) return rc;
  }
  if (K > 64) {
    if ((blocks(128, 128) >= 512 || blocks(64, 128) < 2 * blocks(128, 128)) && wdno_debug_mode != 3) rc = launch_h3<128, 128, 2, 2, LP>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
    // fewer 64 x 128 tiles than CUs (the 8 x 8 level of the Burgers U-Net at batch 16: 128 tiles of 288 steps each): 64 x 64 tiles
    else if (blocks(64, 128) < 200 && wdno_debug_mode != 18) {
      // ... and when even those leave CUs with a single block of several hundred steps, two blocks per tile (split over the tap rows)
      if (blocks(64, 64) <= 384 && g->kd * g->kh >= 2 && p.nsteps >= 64 && p.identity_out && residual != y && !LP && wdno_debug_mode != 19) p.ksplit = 2;
      rc = launch_h3<64, 64, 2, 2, LP>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
    }
    else rc = launch_h3<64, 128, 1, 4, LP>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
  } else {
    if ((blocks(128, 64) >= 512 || P <= 128) && wdno_debug_mode != 3) rc = launch_h3<128, 64, 4, 1, LP>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
    else rc = launch_h3<64, 64, 2, 2, LP>(xh, xl, wph, wpl, sx, sw, bias, residual, y, p, st);
  }
  if (rc) return rc;
  return wdno_check_launch();
}
extern "C" int wdno_conv_fwd_f16x3_amax(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                                        const float* bias, const float* residual, float* y, float* amax_rec, const wdno_conv_geom* g,
                                        wdno_stream_t s) {
  return conv_fwd_16<false>(xh, xl, sx, wph, wpl, sw, bias, residual, y, amax_rec, g, s);
}
// ... with a workspace for the partial sums of a split reduction (conv_h3t.hip): wdno_conv_fwd_split_ws_bytes() bytes, 0 = this geometry
// never splits. Same results contract; the partial sums are added in a fixed order.
extern "C" int wdno_conv_fwd_f16x3_ws(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                                      const float* bias, const float* residual, float* y, float* amax_rec, const wdno_conv_geom* g,
                                      void* ws, size_t ws_bytes, wdno_stream_t s) {
  return conv_fwd_16<false>(xh, xl, sx, wph, wpl, sw, bias, residual, y, amax_rec, g, s, (float*)ws, ws ? ws_bytes : 0);
}
// ... with a statement about structural zeros of x (wdno_zero_box, include/wdno_hip.h): same results bit for bit (up to the sign of a zero),
// the 7 x 7 x 7 stem skips the reduction stages that only see such zeros
extern "C" int wdno_conv_fwd_f16x3_zbox(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                                        const float* bias, const float* residual, float* y, float* amax_rec, const wdno_conv_geom* g,
                                        const wdno_zero_box* zb, wdno_stream_t s) {
  return conv_fwd_16<false>(xh, xl, sx, wph, wpl, sw, bias, residual, y, amax_rec, g, s, nullptr, 0, zb);
}
// single-product form with both options: zb (may be NULL) as above; y_bf16 != 0: y is bf16 storage [rows][K] -- the output of a convolution whose
// only reader is a GroupNorm (and the data gradient whose only reader is a GroupNorm's backward): wdno_groupnorm_*_t take it as it is
extern "C" int wdno_conv_fwd_bf16_ex(const void* x16, const void* wp16, const float* bias, const float* residual, void* y, int y_bf16, float* amax_rec,
                                     const wdno_conv_geom* g, const wdno_zero_box* zb, wdno_stream_t s) {
  return conv_fwd_16<true>(x16, x16, nullptr, wp16, wp16, nullptr, bias, residual, (float*)y, amax_rec, g, s, nullptr, 0, zb, y_bf16);
}
extern "C" size_t wdno_conv_fwd_split_ws_bytes(const wdno_conv_geom* g) {
  if (!g || check_geom(g) || (g->C & 7)) return 0;
  ConvP p;
  fill_params(p, g);
  if (!p.identity_out || !wdno_conv_h3t_takes(*g)) return 0;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const int runs = wdno_conv_h3t_split(*g, p.P, cus & ~7);
  return runs > 1 ? (size_t)runs * p.P * g->K * sizeof(float) : 0;
}
extern "C" int wdno_conv_fwd_bf16(const void* x16, const void* wp16, const float* bias, const float* residual, float* y, float* amax_rec,
                                  const wdno_conv_geom* g, wdno_stream_t s) {
  return conv_fwd_16<true>(x16, x16, nullptr, wp16, wp16, nullptr, bias, residual, y, amax_rec, g, s);
}

// ---------------------------------------------------------------------------------------------- weight gradient (3 x fp16 split)
// dwp[tap][k][r] = sum_p dy[p][k] * x[p shifted by tap][r]. The reduction index (pixels) is the SLOW index of both
// operands in memory, while v_mfma_f32_32x32x16_f16 wants 8 consecutive reduction elements per lane. The LDS tiles keep
// the natural [pixel][channel] layout (16-byte coalesced loads / stores) and the fragments are read with the gfx950
// transpose read ds_read_b64_tr_b16: inside a 16-lane group, lanes 4j..4j+3 point at the four 8-byte pieces of pixel
// row j (16 channels) and lane c receives channel c of rows 0..3 (verified by tools/probes/tr_probe.hip). Two reads
// give the 8 pixels of one MFMA operand. Rows are padded by 32 halves so the 4 pixel rows of a group fall on distinct
// 64-byte bank quarters.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#define WH_BKP 32
#define WH_PAD 32

// pixel table: one 16-byte record per output pixel, built once per geometry (host-cached):
//   x = physical output row (dy / y row index), y = element offset of input pixel (d0, h0, w0) (tap dz = dy = 0, may be
//   negative), z = d0 << 16 | (h0 & 0xffff), w = w0          with d0 = od*sd - pd etc.
__global__ __launch_bounds__(256) void pixel_table_kernel(int4v* __restrict__ tbl, wdno_conv_geom g, int P) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  int q = p;
  int ow = q % g.OW; q /= g.OW;
  int oh = q % g.OH; q /= g.OH;
  int od = q % g.OD;
  int n = q / g.OD;
  int d0 = od * g.sd - g.pd, h0 = oh * g.sh - g.ph, w0 = ow * g.sw - g.pw;
  int4v e;
  e.x = ((n * g.YD + (od * g.osd + g.ood)) * g.YH + (oh * g.osh + g.ooh)) * g.YW + (ow * g.osw + g.oow);
  e.y = (((n * g.D + d0) * g.H + h0) * g.W + w0) * g.C;
  e.z = (d0 << 16) | (h0 & 0xffff);
  e.w = w0;
  tbl[p] = e;
}
extern "C" int wdno_conv_pixel_table(void* table, const wdno_conv_geom* g, wdno_stream_t s) {
  int rc = check_geom(g);
  if (rc) return rc;
  int64_t P = (int64_t)g->N * g->OD * g->OH * g->OW;
  if (P >= 0x7fffffff || (int64_t)g->N * g->D * g->H * g->W * g->C >= 0x3fffffff) return WDNO_EUNSUPPORTED;
  pixel_table_kernel<<<(unsigned)cdiv64(P, 256), 256, 0, as_stream(s)>>>((int4v*)table, *g, (int)P);
  return wdno_check_launch();
}

struct WgradHP {
  ConvP c;
  int tiles_k, tiles_r, splits, bn;
  int64_t pix_per_split;
  int xcd_group;      // blocks renumbered so that each XCD owns a contiguous range of (split, tile) pairs
};

__device__ __forceinline__ half8 tr_frag(const _Float16* tile, int stride, int pix0, int ch0, int lane) {
  // operand fragment for channels ch0..ch0+31 (lane&31 mapping of the MFMA) and pixels pix0..pix0+15
  const int g = lane >> 4, xl = lane & 15;
  const _Float16* p0 = tile + (pix0 + 8 * (g >> 1) + (xl >> 2)) * stride + ch0 + 16 * (g & 1) + 4 * (xl & 3);
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef short4v __attribute__((address_space(3))) * lds_s4;
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * stride));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8, c);
}

#define WH_RING 4
template <int BM, int BN, int WM, int WN, bool LP>
__global__ __launch_bounds__(256) void conv_wgrad_h3_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                             const _Float16* __restrict__ dyh, const _Float16* __restrict__ dyl,
                                                             const float* __restrict__ sx, const float* __restrict__ sdy,
                                                             const int4v* __restrict__ table, float* __restrict__ ws, WgradHP wpz,
                                                             unsigned x_bytes, unsigned dy_bytes) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int SA = BM + WH_PAD, SB = BN + WH_PAD;          // LDS row strides (halves)
  constexpr int A_C8 = BM / 8, B_C8 = BN / 8;                 // 16-byte pieces per pixel row
  constexpr int A_RPP = 256 / A_C8;                           // pixel rows per pass
  constexpr int A_PASSES = WH_BKP / A_RPP, B_PASSES = WH_BKP * B_C8 / 256;
  static_assert(256 % A_C8 == 0 && (WH_BKP * B_C8) % 256 == 0, "loader mapping");
  constexpr int NPL = LP ? 1 : 2;
  constexpr int STAGE = NPL * WH_BKP * (SA + SB);             // halves per stage: Ah, Al, Bh, Bl (LP: A, B)
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
  int4v* pinfo = reinterpret_cast<int4v*>(hsm + 2 * STAGE);   // [WH_RING][WH_BKP]

  const ConvP& p = wpz.c;
  const wdno_conv_geom& g = p.g;
  const int tid = threadIdx.x;
  // (pixel split, k tile, tap row, r tile) from the block number, r tile fastest -- after xcd_swizzle, so that every XCD
  // (block b runs on XCD b % 8) works on one contiguous range of them: the tiles of a pixel split then pull that split's dy
  // and x rows through ONE L2. Unswizzled, the 27 tiles of a split of the 128 -> 64 level-0 layer sit on all eight XCDs and the
  // launch moves 4.6 GB over the fabric for 236 MB of operands.
  int b = wpz.xcd_group ? xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y))
                        : (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const int split = b / (int)gridDim.x;
  b -= split * (int)gridDim.x;
  const int tile_r = b % wpz.tiles_r; b /= wpz.tiles_r;
  const int tap = b % (g.kd * g.kh);
  const int tile_k = b / (g.kd * g.kh);
  const int dz = tap / g.kh, dyy = tap - dz * g.kh;
  const int k0 = tile_k * BM, r0 = tile_r * BN;
  const int pbeg = (int)((int64_t)split * wpz.pix_per_split);
  int pend = pbeg + (int)wpz.pix_per_split;
  if (pend > (int)p.P) pend = (int)p.P;
  const int nsteps = pbeg < pend ? (pend - pbeg + WH_BKP - 1) / WH_BKP : 0;
  const __amdgpu_buffer_rsrc_t rxh = make_rsrc(xh, x_bytes), rxl = make_rsrc(xl, x_bytes);
  const __amdgpu_buffer_rsrc_t rdh = make_rsrc(dyh, dy_bytes), rdl = make_rsrc(dyl, dy_bytes);
  const int tap_off = (dz * g.H + dyy) * g.W * g.C;          // uniform element offset of this block's tap row

  // pixel records travel table -> register -> LDS ring, several steps ahead of their use
  auto fetch_rec = [&](int step) -> int4v {
    int4v e;
    e.x = 0; e.y = 0; e.z = (int)0x80008000; e.w = 0;         // d0 = h0 = -32768: never valid
    int pm = pbeg + step * WH_BKP + tid;
    if (tid < WH_BKP && step < nsteps && pm < pend) e = table[pm];
    if (tid < WH_BKP && !(step < nsteps && pm < pend)) e.x = -1;
    return e;
  };
  auto put_rec = [&](int step, int4v e) {
    if (tid < WH_BKP) pinfo[(step & (WH_RING - 1)) * WH_BKP + tid] = e;
  };

  // B pieces: piece index tid + 256*i -> (pixel row, 8-channel group); BN need not divide 256 (192-wide tiles)
  int b_row[B_PASSES], b_c8[B_PASSES], r[B_PASSES], dx[B_PASSES];
  bool r_ok[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int idx = tid + 256 * i;
    b_row[i] = idx / B_C8;
    b_c8[i] = (idx - b_row[i] * B_C8) * 8;
    r[i] = r0 + b_c8[i];
    r_ok[i] = r[i] < p.R;
    dx[i] = r[i] / g.C;
  }
  const int a_row = tid / A_C8, a_c8 = (tid % A_C8) * 8;
  const int ka = k0 + a_c8;
  const bool ka_ok = ka < g.K;

  int4v ah[A_PASSES], al[A_PASSES], bh[B_PASSES], bl[B_PASSES];
  auto load_tile = [&](int step) {
    const int4v* ps = pinfo + (step & (WH_RING - 1)) * WH_BKP;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int4v e = ps[a_row + i * A_RPP];
      const int off = (e.x >= 0 && ka_ok) ? (e.x * g.K + ka) * 2 : OOB_OFFSET;
      ah[i] = __builtin_amdgcn_raw_buffer_load_b128(rdh, off, 0, 0);
      if (!LP) al[i] = __builtin_amdgcn_raw_buffer_load_b128(rdl, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int4v e = ps[b_row[i]];
      const int d = (e.z >> 16) + dz, h = (int)(short)(e.z & 0xffff) + dyy, w = e.w + dx[i];
      const bool ok = e.x >= 0 && r_ok[i] && (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
      const int off = ok ? (e.y + tap_off + r[i]) * 2 : OOB_OFFSET;
      bh[i] = __builtin_amdgcn_raw_buffer_load_b128(rxh, off, 0, 0);
      if (!LP) bl[i] = __builtin_amdgcn_raw_buffer_load_b128(rxl, off, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
    _Float16* Ah = hsm + buf * STAGE;
    _Float16* Al = Ah + WH_BKP * SA;
    _Float16* Bh = Ah + NPL * WH_BKP * SA;
    _Float16* Bl = Bh + WH_BKP * SB;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      *reinterpret_cast<int4v*>(&Ah[(a_row + i * A_RPP) * SA + a_c8]) = ah[i];
      if (!LP) *reinterpret_cast<int4v*>(&Al[(a_row + i * A_RPP) * SA + a_c8]) = al[i];
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      *reinterpret_cast<int4v*>(&Bh[b_row[i] * SB + b_c8[i]]) = bh[i];
      if (!LP) *reinterpret_cast<int4v*>(&Bl[b_row[i] * SB + b_c8[i]]) = bl[i];
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
    put_rec(0, fetch_rec(0));
    put_rec(1, fetch_rec(1));
    put_rec(2, fetch_rec(2));
    int4v rec = fetch_rec(3);            // record of step s+3 is written to LDS during iteration s, loaded one iteration earlier
    __syncthreads();
    load_tile(0);
    store_tile(0);
    __syncthreads();
    if (nsteps > 1) load_tile(1);
    for (int step = 0; step < nsteps; ++step) {
      const _Float16* Ah = hsm + (step & 1) * STAGE;
      const _Float16* Al = Ah + WH_BKP * SA;
      const _Float16* Bh = Ah + NPL * WH_BKP * SA;
      const _Float16* Bl = Bh + WH_BKP * SB;
      put_rec(step + 3, rec);            // slot (step+3)&3 was last read by load_tile(step-1): two barriers ago
      rec = fetch_rec(step + 4);
#pragma unroll
      for (int ks = 0; ks < WH_BKP / 16; ++ks) {
        half8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          fah[a] = tr_frag(Ah, SA, ks * 16, m_base + a * 32, lane);
          if (!LP) fal[a] = tr_frag(Al, SA, ks * 16, m_base + a * 32, lane);
        }
#pragma unroll
        for (int bb = 0; bb < TN; ++bb) {
          fbh[bb] = tr_frag(Bh, SB, ks * 16, n_base + bb * 32, lane);
          if (!LP) fbl[bb] = tr_frag(Bl, SB, ks * 16, n_base + bb * 32, lane);
        }
        if constexpr (!LP) {
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int bb = 0; bb < TN; ++bb) acc[a][bb] = mfma_16<false>(fal[a], fbh[bb], acc[a][bb]);
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int bb = 0; bb < TN; ++bb) acc[a][bb] = mfma_16<false>(fah[a], fbl[bb], acc[a][bb]);
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int bb = 0; bb < TN; ++bb) acc[a][bb] = mfma_16<LP>(fah[a], fbh[bb], acc[a][bb]);
      }
      if (step + 1 < nsteps) store_tile((step + 1) & 1);
      __syncthreads();
      if (step + 2 < nsteps) load_tile(step + 2);
    }
  }

  const float inv = LP ? 1.0f : 1.0f / (sx[0] * sdy[0]);
  float* out = ws + ((int64_t)split * (g.kd * g.kh) + tap) * (int64_t)g.K * p.R;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int kk = k0 + m_base + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (kk >= g.K) continue;
#pragma unroll
      for (int bb = 0; bb < TN; ++bb) {
        int rr = r0 + n_base + bb * 32 + li;
        if (rr < p.R) out[(int64_t)kk * p.R + rr] = acc[a][bb][e] * inv;
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_h3_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int splits) {
  // one float4 column per thread, four independent chains over the splits (a fixed order: the result is deterministic)
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 3 < splits; s += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)(s + u) * n + 4 * i);
        a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
      }
    }
    for (; s < splits; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)s * n + 4 * i);
      a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    }
    *reinterpret_cast<float4*>(out + 4 * i) = make_float4((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y),
                                                          (a[0].z + a[1].z) + (a[2].z + a[3].z), (a[0].w + a[1].w) + (a[2].w + a[3].w));
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      float acc = 0.f;
      for (int s = 0; s < splits; ++s) acc += ws[(int64_t)s * n + i];
      out[i] = acc;
    }
}

static void wgrad_h3_plan(WgradHP& w, const wdno_conv_geom* g) {
  fill_params(w.c, g);
  w.xcd_group = wdno_debug_mode != 6;      // debug 6: plain block order (the A/B)
  const int BM = g->K > 64 ? 128 : 64;
  w.tiles_k = cdiv(g->K, BM);
  // column tile of the kw*C run: 192 for the wide-K kernel (64 x 96 per wave beats 64 x 64), and for K <= 64 only when it
  // removes a half-empty tile (kw*C = 192: the 3-wide taps of the 64-channel layers)
  w.bn = (g->K > 64 || (w.c.R % 128 != 0 && w.c.R % 192 == 0)) ? 192 : 128;
  w.tiles_r = cdiv(w.c.R, w.bn);
  int64_t tiles = (int64_t)w.tiles_k * w.tiles_r * g->kd * g->kh;
  int64_t want = 512 / tiles;       // at most 512 blocks (floor): whole rounds of the 256 CUs, never a short extra round
  if (want < 1) want = 1;
  int64_t max_splits = cdiv64(w.c.P, 16 * WH_BKP);   // at least 16 steps per block
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  if (want > 4096) want = 4096;
  int64_t pps = cdiv64(cdiv64(w.c.P, want), WH_BKP) * WH_BKP;
  w.pix_per_split = pps;
  w.splits = (int)cdiv64(w.c.P, pps);
}
extern "C" size_t wdno_conv_wgrad_f16x3_ws_bytes(const wdno_conv_geom* g) {
  if (check_geom(g) != WDNO_OK) return 0;
  WgradHP w;
  wgrad_h3_plan(w, g);
  int bm, bn, splits, pps, sp_mode;
  wdno_wgrad_h3d_plan(g, &bm, &bn, &splits, &pps, &sp_mode);       // the DMA kernel's plan may use a different split count
  if (splits > w.splits) w.splits = splits;
  return (size_t)w.splits * g->kd * g->kh * (size_t)g->K * w.c.R * sizeof(float);
}
template <int BM, int BN, int WM, int WN, bool LP>
static void launch_wgrad_h3(const void* xh, const void* xl, const void* dyh, const void* dyl, const float* sx, const float* sdy,
                            const void* table, float* wsf, const WgradHP& w, dim3 grid, hipStream_t st) {
  const wdno_conv_geom& g = w.c.g;
  const unsigned x_bytes = (unsigned)((int64_t)g.N * g.D * g.H * g.W * g.C * 2);
  const unsigned dy_bytes = (unsigned)((int64_t)g.N * g.YD * g.YH * g.YW * g.K * 2);
  size_t lds = (size_t)2 * (LP ? 1 : 2) * WH_BKP * (BM + WH_PAD + BN + WH_PAD) * sizeof(_Float16) + WH_RING * WH_BKP * sizeof(int4v);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_h3_kernel<BM, BN, WM, WN, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  conv_wgrad_h3_kernel<BM, BN, WM, WN, LP><<<grid, 256, lds, st>>>((const _Float16*)xh, (const _Float16*)xl, (const _Float16*)dyh,
                                                            (const _Float16*)dyl, sx, sdy, (const int4v*)table, wsf, w, x_bytes, dy_bytes);
}
// Runs the weight-gradient kernels; the per-split partial results go to `ws` ([splits][ntap][K][R]; with one split `single`
// may name the final packed destination instead). Returns the split count through *splits_out.
static int wgrad_h3_partials(const void* xh, const void* xl, const float* sx, const void* dyh, const void* dyl, const float* sdy,
                             const void* pixel_table, float* single, void* ws, size_t ws_bytes, const wdno_conv_geom* g, hipStream_t st,
                             int* splits_out) {
  int rc = check_geom(g);
  if (rc) return rc;
  if ((g->C & 7) || (g->K & 7) || !pixel_table) return WDNO_EUNSUPPORTED;
  if ((int64_t)g->N * g->D * g->H * g->W * g->C * 2 >= OOB_OFFSET || (int64_t)g->N * g->YD * g->YH * g->YW * g->K * 2 >= OOB_OFFSET)
    return WDNO_EUNSUPPORTED;        // 32-bit buffer offsets
  WgradHP w;
  wgrad_h3_plan(w, g);
  // persistent LDS-DMA kernel for everything it takes (debug 5 = never); the register-staged kernels below are the fallback
  if (wdno_debug_mode != 5) {
    int bm, bn, splits, pps, sp_mode;
    wdno_wgrad_h3d_plan(g, &bm, &bn, &splits, &pps, &sp_mode);
    size_t need_d = (size_t)splits * g->kd * g->kh * (size_t)g->K * w.c.R * sizeof(float);
    float* dst = (splits == 1 && single) ? single : (float*)ws;
    if (dst == (float*)ws && ws_bytes < need_d) return WDNO_EWORKSPACE;
    rc = wdno_conv_wgrad_h3_dma(xh, xl, dyh, dyl, sx, sdy, pixel_table, dst, g, st);
    if (rc == WDNO_OK) { *splits_out = splits; return WDNO_OK; }
    if (rc != WDNO_EUNSUPPORTED) return rc;
  }
  size_t need = (size_t)w.splits * g->kd * g->kh * (size_t)g->K * w.c.R * sizeof(float);
  float* wsf = (w.splits == 1 && single) ? single : (float*)ws;
  if (wsf == (float*)ws && ws_bytes < need) return WDNO_EWORKSPACE;
  if (w.splits > 65535) return WDNO_EUNSUPPORTED;
  dim3 grid((unsigned)(w.tiles_k * g->kd * g->kh * w.tiles_r), (unsigned)w.splits);
  if (xl == nullptr) {          // single bf16 plane per operand
    if (g->K > 64) launch_wgrad_h3<128, 192, 2, 2, true>(xh, xh, dyh, dyh, sx, sdy, pixel_table, wsf, w, grid, st);
    else if (w.bn == 192) launch_wgrad_h3<64, 192, 2, 2, true>(xh, xh, dyh, dyh, sx, sdy, pixel_table, wsf, w, grid, st);
    else launch_wgrad_h3<64, 128, 1, 4, true>(xh, xh, dyh, dyh, sx, sdy, pixel_table, wsf, w, grid, st);
  } else if (g->K > 64) launch_wgrad_h3<128, 192, 2, 2, false>(xh, xl, dyh, dyl, sx, sdy, pixel_table, wsf, w, grid, st);
  else if (w.bn == 192) launch_wgrad_h3<64, 192, 2, 2, false>(xh, xl, dyh, dyl, sx, sdy, pixel_table, wsf, w, grid, st);
  else launch_wgrad_h3<64, 128, 1, 4, false>(xh, xl, dyh, dyl, sx, sdy, pixel_table, wsf, w, grid, st);
  *splits_out = w.splits;
  return WDNO_OK;
}
extern "C" int wdno_conv_wgrad_f16x3(const void* xh, const void* xl, const float* sx, const void* dyh, const void* dyl, const float* sdy,
                                     const void* pixel_table, float* dwp, void* ws, size_t ws_bytes, const wdno_conv_geom* g, wdno_stream_t s) {
  hipStream_t st = as_stream(s);
  int splits = 0;
  int rc = wgrad_h3_partials(xh, xl, sx, dyh, dyl, sdy, pixel_table, dwp, ws, ws_bytes, g, st, &splits);
  if (rc) return rc;
  if (splits > 1) {
    int64_t n = (int64_t)g->kd * g->kh * g->K * (int64_t)g->kw * g->C;
    wgrad_h3_reduce_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, st>>>((const float*)ws, dwp, n, splits);
  }
  return wdno_check_launch();
}
// Same, but the gradient is delivered in the layout of the parameter, dw[Kn][Cn][kd][kh][kw] (Kn <= g->K, Cn <= g->C: the
// channel padding is dropped), by the split reduction itself -- no packed intermediate, no permute / slice copy afterwards.
// A block = 32 groups of four consecutive elements x 8 slices of the split list: with ~56 splits of a few hundred KB each, one
// thread per element walking all splits is a chain of 14 dependent loads on a few hundred blocks (14.5 us per launch, 72
// launches per train step); here a thread adds 7 float4 and the eight slices meet in LDS.
// (bid of nb blocks: the kernel's own grid, or the blocks one item owns inside a multi-tensor launch)
__device__ __forceinline__ void wgrad_h3_reduce_nat_body(const float* __restrict__ ws, float* __restrict__ dw, int64_t n, int splits,
                                                         int K8, int C8, int kw, int ntap, int Kn, int Cn, float4 (*part)[33], int bid, int nb) {
  const int R = kw * C8;
  const int j = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int64_t n4 = n >> 2;
  for (int64_t i4 = (int64_t)bid * 32 + j; i4 - j < n4; i4 += (int64_t)nb * 32) {      // block-uniform trip count
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (i4 < n4) {
      const float4* src = reinterpret_cast<const float4*>(ws) + i4;
      int sidx = q;
      for (; sidx + 8 < splits; sidx += 16) {
        const float4 u = src[(int64_t)sidx * n4], v = src[(int64_t)(sidx + 8) * n4];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
      }
      if (sidx < splits) {
        const float4 u = src[(int64_t)sidx * n4];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      }
    }
    part[q][j] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    __syncthreads();
    // thread (q, j), q < 4: component q of group j
    if (q < 4 && i4 < n4) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float4 v = part[g][j];
        t += q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
      }
      const int64_t i = i4 * 4 + q;
      const int r = (int)(i % R);
      int64_t tt = i / R;
      const int k = (int)(tt % K8);
      const int tap = (int)(tt / K8);
      const int dx = r / C8, c = r - dx * C8;
      if (k < Kn && c < Cn) dw[(((int64_t)k * Cn + c) * ntap + tap) * kw + dx] = t;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void wgrad_h3_reduce_nat_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t n, int splits,
                                                                   int K8, int C8, int kw, int ntap, int Kn, int Cn) {
  __shared__ float4 part[8][33];
  wgrad_h3_reduce_nat_body(ws, dw, n, splits, K8, C8, kw, ntap, Kn, Cn, part, (int)blockIdx.x, (int)gridDim.x);
}
// The same reduction with coalesced stores. Above, consecutive threads hold consecutive (dx, c) of one (tap row, k) and scatter them
// ntap * kw floats apart in the parameter layout [K][C][taps]. Here a block owns one k and 64 input channels: it reads its T = ntap * kw
// segments of 64 contiguous floats per split (one wave per segment, lane = channel), sums the splits in order, turns the [T][64] tile
// through LDS and writes 64 * T contiguous floats of dw. T <= 64 (everything but the 7 x 7 x 7 stem).
__device__ __forceinline__ void wgrad_h3_reduce_tile_body(const float* __restrict__ ws, float* __restrict__ dw, int64_t n, int splits,
                                                          int K8, int C8, int kw, int ntap, int Kn, int Cn, float (*tile)[65], int bx, int by) {
  const int T = ntap * kw, R = kw * C8;
  const int k = by, c0 = bx * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int sgm = wave; sgm < T; sgm += 4) {
    const int t = sgm / kw, dx = sgm - t * kw;
    float a = 0.f, b = 0.f;
    if (c0 + lane < C8) {
      const float* src = ws + ((int64_t)t * K8 + k) * R + dx * C8 + c0 + lane;
      int sp = 0;
      for (; sp + 1 < splits; sp += 2) { a += src[(int64_t)sp * n]; b += src[(int64_t)(sp + 1) * n]; }
      if (sp < splits) a += src[(int64_t)sp * n];
    }
    tile[lane][sgm] = a + b;
  }
  __syncthreads();
  const int cn = min(64, Cn - c0);                    // channels of this tile that exist in the parameter
  float* out = dw + ((int64_t)k * Cn + c0) * T;
  for (int e = threadIdx.x; e < cn * T; e += 256) {
    const int c = e / T, sgm = e - c * T;
    out[e] = tile[c][sgm];
  }
}
__global__ __launch_bounds__(256) void wgrad_h3_reduce_tile_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t n, int splits,
                                                                    int K8, int C8, int kw, int ntap, int Kn, int Cn) {
  __shared__ float tile[64][65];
  wgrad_h3_reduce_tile_body(ws, dw, n, splits, K8, C8, kw, ntap, Kn, Cn, tile, (int)blockIdx.x, (int)blockIdx.y);
}
// which of the two reductions a layer takes, and with how many blocks (shared by the per-layer launch and the multi-tensor one)
static bool wgrad_h3_reduce_tiled(int splits, const wdno_conv_geom* g, int Kn, int Cn) {
  const int T = g->kd * g->kh * g->kw;
  const bool big = (int64_t)Kn * cdiv(Cn, 64) >= 512 && splits <= 8 && wdno_debug_mode != 39;
  return T <= 64 && (splits <= 2 || big) && wdno_debug_mode != 38;             // debug 38: the scatter kernel (A/B)
}

// ---- every pending split reduction of a backward pass in ONE launch (round 6). A training step ran 58 of these reductions, 5-10 us each, one behind
// every weight-gradient kernel (0.43 ms per smoke step, profiles/r05_smoke_kernel_stats.md); only the optimiser reads their results. The callers
// now run the partial-sum kernels alone (wdno_conv_wgrad_*_partials: the workspace stays alive) and hand the list of reductions over at the end of
// the backward: the items travel BY VALUE in the kernel arguments (<= WDNO_WGRAD_REDUCE_MAX per launch: 2.4 KB of arguments, well inside the 4 KB a launch takes together with the implicit ones), so the launch needs no table upload and
// replays unchanged from a captured graph. A block finds its item by its first-block number and runs the same body as the per-layer kernels in
// the same order of additions: results are bit-identical to them.
struct WgradReduceArgs {
  wdno_wgrad_reduce_item it[WDNO_WGRAD_REDUCE_MAX];
  int first[WDNO_WGRAD_REDUCE_MAX + 1];          // first block of item i; first[n] = grid
  int n;
};
__global__ __launch_bounds__(256) void wgrad_h3_reduce_multi_kernel(const WgradReduceArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[64 * 65];                // tile[64][65] of the tiled body / part[8][33] float4 of the scatter body
  int lo = 0, hi = a.n - 1;                      // last item whose first block is <= blockIdx.x (block-uniform: scalar loads of the arguments)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const wdno_wgrad_reduce_item& w = a.it[lo];
  const int bid = (int)blockIdx.x - a.first[lo], nb = a.first[lo + 1] - a.first[lo];
  if (w.tiled) {
    const int bxn = (w.Cn + 63) >> 6;
    wgrad_h3_reduce_tile_body(w.ws, w.dw, w.n, w.splits, w.K8, w.C8, w.kw, w.ntap, w.Kn, w.Cn, reinterpret_cast<float (*)[65]>(smem), bid % bxn, bid / bxn);
  } else {
    wgrad_h3_reduce_nat_body(w.ws, w.dw, w.n, w.splits, w.K8, w.C8, w.kw, w.ntap, w.Kn, w.Cn, reinterpret_cast<float4 (*)[33]>(smem), bid, nb);
  }
}
extern "C" int wdno_wgrad_reduce_multi(const wdno_wgrad_reduce_item* items, int n_items, wdno_stream_t s) {
  WDNO_REQUIRE(items && n_items >= 0);
  hipStream_t st = as_stream(s);
  for (int base = 0; base < n_items; base += WDNO_WGRAD_REDUCE_MAX) {
    WgradReduceArgs a;
    a.n = n_items - base < WDNO_WGRAD_REDUCE_MAX ? n_items - base : WDNO_WGRAD_REDUCE_MAX;
    int grid = 0;
    for (int i = 0; i < a.n; ++i) {
      a.it[i] = items[base + i];
      const wdno_wgrad_reduce_item& w = a.it[i];
      WDNO_REQUIRE(w.ws && w.dw && w.n > 0 && w.splits >= 1 && w.Kn > 0 && w.Cn > 0 && w.Kn <= w.K8 && w.Cn <= w.C8);
      a.first[i] = grid;
      grid += w.tiled ? cdiv(w.Cn, 64) * w.Kn : stream_grid(w.n / 4, 32);      // the per-layer launches' grids
    }
    a.first[a.n] = grid;
    if (grid > 0) wgrad_h3_reduce_multi_kernel<<<grid, 256, 0, st>>>(a);
  }
  return wdno_check_launch();
}

static void wgrad_h3_reduce_launch(const float* ws, float* dw, int64_t n, int splits, const wdno_conv_geom* g, int Kn, int Cn, hipStream_t st) {
  // (layers with many splits are the small ones: K * C / 64 blocks that each walk T x splits segments are too few and too serial there --
  // 64 -> 64 channels with 18 splits took 2x the scatter kernel's time, +1.1 ms per smoke step when used everywhere)
  // ... but a layer with >= 512 (k, 64-channel) blocks and a handful of splits (256 -> 256 at level 2: 1024 blocks, 3 splits) has enough of them, and
  // the scatter kernel's 4-byte stores 27 floats apart are what it pays for (31-33 us per launch, 7 launches per smoke step)
  if (wgrad_h3_reduce_tiled(splits, g, Kn, Cn)) {
    wgrad_h3_reduce_tile_kernel<<<dim3(cdiv(Cn, 64), Kn), 256, 0, st>>>(ws, dw, n, splits, g->K, g->C, g->kw, g->kd * g->kh, Kn, Cn);
    return;
  }
  wgrad_h3_reduce_nat_kernel<<<stream_grid(n / 4, 32), 256, 0, st>>>(ws, dw, n, splits, g->K, g->C, g->kw, g->kd * g->kh, Kn, Cn);
}
extern "C" int wdno_conv_wgrad_f16x3_param(const void* xh, const void* xl, const float* sx, const void* dyh, const void* dyl, const float* sdy,
                                           const void* pixel_table, float* dw, int Kn, int Cn, void* ws, size_t ws_bytes,
                                           const wdno_conv_geom* g, wdno_stream_t s) {
  WDNO_REQUIRE(g && Kn > 0 && Cn > 0 && Kn <= g->K && Cn <= g->C);
  hipStream_t st = as_stream(s);
  int splits = 0;
  int rc = wgrad_h3_partials(xh, xl, sx, dyh, dyl, sdy, pixel_table, nullptr, ws, ws_bytes, g, st, &splits);
  if (rc) return rc;
  const int64_t n = (int64_t)g->kd * g->kh * g->K * (int64_t)g->kw * g->C;
  wgrad_h3_reduce_launch((const float*)ws, dw, n, splits, g, Kn, Cn, st);
  return wdno_check_launch();
}

// The partial-sum kernels alone: `item` comes back filled for wdno_wgrad_reduce_multi (ws must stay alive and unmodified until that launch).
extern "C" int wdno_conv_wgrad_partials(const void* xh, const void* xl, const float* sx, const void* dyh, const void* dyl, const float* sdy,
                                        const void* pixel_table, float* dw, int Kn, int Cn, void* ws, size_t ws_bytes,
                                        const wdno_conv_geom* g, wdno_wgrad_reduce_item* item, wdno_stream_t s) {
  WDNO_REQUIRE(g && item && dw && Kn > 0 && Cn > 0 && Kn <= g->K && Cn <= g->C);
  int splits = 0;
  int rc = wgrad_h3_partials(xh, xl, sx, dyh, dyl, sdy, pixel_table, nullptr, ws, ws_bytes, g, as_stream(s), &splits);      // xl == NULL: one bf16 plane per operand
  if (rc) return rc;
  const int64_t n = (int64_t)g->kd * g->kh * g->K * (int64_t)g->kw * g->C;
  if (n >= 0x7fffffff) return WDNO_EUNSUPPORTED;
  item->ws = (const float*)ws; item->dw = dw; item->n = (int)n; item->splits = splits; item->K8 = g->K; item->C8 = g->C; item->kw = g->kw;
  item->ntap = g->kd * g->kh; item->Kn = Kn; item->Cn = Cn; item->tiled = wgrad_h3_reduce_tiled(splits, g, Kn, Cn) ? 1 : 0;
  return wdno_check_launch();
}

extern "C" int wdno_conv_wgrad_bf16_param(const void* x16, const void* dy16, const void* pixel_table, float* dw, int Kn, int Cn, void* ws,
                                          size_t ws_bytes, const wdno_conv_geom* g, wdno_stream_t s) {
  WDNO_REQUIRE(g && Kn > 0 && Cn > 0 && Kn <= g->K && Cn <= g->C);
  hipStream_t st = as_stream(s);
  int splits = 0;
  int rc = wgrad_h3_partials(x16, nullptr, nullptr, dy16, nullptr, nullptr, pixel_table, nullptr, ws, ws_bytes, g, st, &splits);
  if (rc) return rc;
  const int64_t n = (int64_t)g->kd * g->kh * g->K * (int64_t)g->kw * g->C;
  wgrad_h3_reduce_launch((const float*)ws, dw, n, splits, g, Kn, Cn, st);
  return wdno_check_launch();
}

// ---------------------------------------------------------------------------------------------- weight pack + split
// raw weight w[K][C][kd][kh][kw] fp32  ->  fp16 planes of the packed operand, in one launch:
//   mode 0 (forward)  : out[dz][dy][a=k < A][dx][b=c < B]  = w[k][c][dz][dy][dx]
//   mode 1 (data grad): out[dz][dy][a=c < A][dx][b=k < B]  = w[k][c][kd-1-dz][kh-1-dy][kw-1-dx]
// A / B are the padded extents of the two channel roles (B % 8 == 0); entries outside K / C are zero.
// Tiled through LDS (round 2). The source element of output (tap, a, b) is w[(k*C + c)*T + tap] with (k, c) = (a, b) for the forward
// operand and (b, a), flipped taps, for the data-gradient operand: in w a run of c at fixed k is contiguous ([c][tap]), in the output a
// run of b is. A block moves a tile of KT k's x CT c's x all T taps: it reads KT contiguous runs of CT*T floats (coalesced 16-byte
// loads) into LDS and writes, per (tap, a), the tile's run of b as 8-element groups (16 B per plane per thread, contiguous across
// threads). The tile is long in the direction the output wants contiguous: forward 8 k x 128 c, data gradient 128 k x 8 c (scaled
// down so that the tile stays under PS_LDS_FLOATS for many taps). The first version ran one grid-stride loop over the output index
// with 4-byte reads T floats apart: every line of w was fetched once per tap and 9x over-fetched per access -- 3.1 ms per step for
// the 563 MB of Burgers weights (8.9 % of its batch-16 training step).
#define PS_LDS_FLOATS 10240
struct PackTile { int KT, CT, CTp, tiles_k, tiles_c; };
__host__ __device__ static inline PackTile pack_tile(int T, int A, int B, int mode) {
  PackTile t;
  // floats of the LDS tile [tap][KT][CT + 1]
  auto need = [&](int longd, int shortd) { return (int64_t)T * (mode ? (int64_t)longd * (shortd + 1) : (int64_t)shortd * (longd + 1)); };
  int longd = 128, shortd = 8;
  while (longd > 8 && need(longd, shortd) > PS_LDS_FLOATS) longd >>= 1;
  while (shortd > 1 && need(longd, shortd) > PS_LDS_FLOATS) shortd >>= 1;
  t.KT = mode ? longd : shortd;          // data gradient: b = k is the contiguous output index
  t.CT = mode ? shortd : longd;
  t.CTp = t.CT + 1;                      // LDS tile [tap][k][c] with an odd c pitch
  const int kext = mode ? B : A, cext = mode ? A : B;      // padded extents of k and c in the output
  t.tiles_k = (kext + t.KT - 1) / t.KT;
  t.tiles_c = (cext + t.CT - 1) / t.CT;
  return t;
}
// n / d for n * d < 2^32 with m = 2^32 / d + 1 (d > 1)
__device__ __forceinline__ unsigned ps_magic(unsigned d) { return d > 1 ? (unsigned)((1ull << 32) / d) + 1u : 0u; }
__device__ __forceinline__ int ps_div(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ void pack_split_body(const float* __restrict__ w, const float* __restrict__ amax,
                                                _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                float* __restrict__ scale_out, int K, int C, int kd, int kh, int kw,
                                                int A, int B, int mode_in, int block, int nblocks) {
  __shared__ float tile[PS_LDS_FLOATS + 16];
  // modes 2 .. 5: forward operand of parity class (py, px) = ((mode - 2) >> 1, (mode - 2) & 1) of the (1,4,4) / stride (1,2,2) transposed
  // convolution, read straight from its weight w[in = C][out = K][1][4][4]:  W[k][c][0][dy][dx] = w[c][k][0][3 - py - 2 dy][3 - px - 2 dx]
  // (kd, kh, kw = 1, 2, 2). Only the gather below differs; everything after it is the forward layout.
  // modes 6 .. 9: data-gradient operand of tap (py, px) = ((mode - 6) >> 1, (mode - 6) & 1) of a (1,2,2) / stride (1,2,2) convolution (the folded
  // Downsample of the Burgers U-Net, unet.py:64-68), whose data gradient is four 1 x 1 convolutions of dy scattered to the four pixel parities:
  // out[a = c][b = k] = w[k][c][0][py][px]  (told kd, kh, kw = 1, 1, 1; data-gradient layout, only the gather differs)
  // modes 10 / 11: the forward operand of a 1 x 1 projection in the FRAGMENT order of csrc/attn_fused_wide.hip (to_qkv [384][C] / to_out [C][128] of
  // a temporal attention block): the 16-byte group (row a, channels 8 bg ..) moves so that the 64 lanes of one matrix-operand load read 1 KB
  //   10: a = ti 128 + h 32 + li, bg = 4 t + 2 hh + s  ->  ((((h 3 + ti) B/32 + t) 2 + s) 64 + hh 32 + li
  //   11: a = wv A/4 + mt 32 + li, bg = 4 t + 2 hh + s ->  ((((wv A/128 + mt) 4 + t) 2 + s) 64 + hh 32 + li
  const int frag = mode_in >= 10 ? mode_in - 9 : 0;
  const int parity = mode_in >= 2 && mode_in < 6 ? mode_in - 2 : -1;
  const int patch = mode_in >= 6 && mode_in < 10 ? mode_in - 6 : -1;
  const int mode = mode_in == 1 || patch >= 0 ? 1 : 0;
  const bool lp = lo == nullptr;                     // one bf16 plane, no scale (amax / scale_out may be NULL)
  const float s = lp ? 1.0f : scale_from_amax(amax[0]);
  if (!lp && block == 0 && threadIdx.x == 0) scale_out[0] = s;
  const int b8 = B >> 3;
  const int T = kd * kh * kw, khw = kh * kw;
  const PackTile pt = pack_tile(T, A, B, mode);
  if ((int64_t)T * pt.KT * pt.CTp > PS_LDS_FLOATS + 16) return;         // > ~600 taps: not a convolution of this code base
  const int ntiles = pt.tiles_k * pt.tiles_c;
  const int run = pt.CT * T;                         // floats per k row of the tile (contiguous in w)
  const int bt8 = (mode ? pt.KT : pt.CT) >> 3;       // 8-element output groups per (tap, a) inside the tile (>= 1: long dims are multiples of 8)
  const int at = mode ? pt.CT : pt.KT;               // values of a inside the tile
  const unsigned m_run = ps_magic(run), m_T = ps_magic(T), m_bt8 = ps_magic(bt8), m_at = ps_magic(at), m_khw = ps_magic(khw), m_kw = ps_magic(kw);
  for (int tl = block; tl < ntiles; tl += nblocks) {
    const int tk = ps_div(tl, pt.tiles_c, ps_magic(pt.tiles_c)), tc = tl - tk * pt.tiles_c;
    const int k0 = tk * pt.KT, c0 = tc * pt.CT;
    __syncthreads();
    // load: row kk of the tile = w[(k0 + kk)*C*T + c0*T ...], `run` contiguous floats ([c][tap]); zero where k >= K or c >= C
    const int cvalid = min(pt.CT, C - c0);           // may be <= 0 for padding tiles
    // eight loads per thread in flight (one at a time, behind a branch, the 36 loads of a thread per 8 x 128 x 9 tile were a chain of
    // round trips: 1.17 ms per step for the 563 MB of Burgers weights = 1.45 TB/s over read + write)
    const int nload = pt.KT * run;
    for (int idx0 = threadIdx.x; idx0 < nload; idx0 += 256 * 8) {
      float v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = min(idx0 + 256 * u, nload - 1);
        const int kk = ps_div(idx, run, m_run), r = idx - kk * run;
        const int cl = ps_div(r, T, m_T), tap = r - cl * T;
        const bool ok = k0 + kk < K && cl < cvalid;
        const int64_t off = !ok ? 0 : patch >= 0 ? (((int64_t)(k0 + kk) * C + c0 + cl) * 4 + patch) : parity < 0 ? ((int64_t)(k0 + kk) * C + c0) * T + r
                            : ((int64_t)(c0 + cl) * K + (k0 + kk)) * 16 + (3 - (parity >> 1) - 2 * (tap >> 1)) * 4 + (3 - (parity & 1) - 2 * (tap & 1));
        v[u] = w[off];
        if (!ok) v[u] = 0.f;
        dst[u] = idx0 + 256 * u < nload ? (tap * pt.KT + kk) * pt.CTp + cl : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) tile[dst[u]] = v[u];
    }
    __syncthreads();
    // store: items (tap, a_local, group of 8 b's), the group index fastest: 16 contiguous bytes per plane per thread
    const int nitems = T * at * bt8;
    for (int idx = threadIdx.x; idx < nitems; idx += 256) {
      int q = ps_div(idx, bt8, m_bt8);
      const int g = idx - q * bt8;
      const int tap = ps_div(q, at, m_at), al = q - tap * at;
      const int dz = ps_div(tap, khw, m_khw), rr = tap - dz * khw;
      const int dy = ps_div(rr, kw, m_kw), dx = rr - dy * kw;
      const int ts = mode ? T - 1 - tap : tap;       // all three axes flipped = the reversed linear tap index
      const int a = (mode ? c0 : k0) + al;
      const int bg = ((mode ? k0 : c0) >> 3) + g;    // 8-element group index along b
      if (a >= A || bg >= b8) continue;
      const float* src = mode ? tile + (ts * pt.KT + g * 8) * pt.CTp + al : tile + (ts * pt.KT + al) * pt.CTp + g * 8;
      const int step = mode ? pt.CTp : 1;
      half8 h, l;
      typedef unsigned short us8 __attribute__((ext_vector_type(8)));
      us8 bq;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = src[e * step];
        const float tv = v * s;
        const _Float16 th = (_Float16)tv;
        h[e] = th;
        l[e] = (_Float16)(tv - (float)th);
        bq[e] = bf16_rne(v);
      }
      int64_t i = ((((int64_t)dz * kh + dy) * A + a) * kw + dx) * b8 + bg;
      if (frag) {
        const int li = a & 31, t = bg >> 2, hh2 = (bg >> 1) & 1, s2 = bg & 1;
        if (frag == 1) { const int ti = a >> 7, hd = (a >> 5) & 3; i = ((((hd * 3 + ti) * (B >> 5) + t) * 2 + s2) * 64 + hh2 * 32 + li); }
        else { const int q4 = A >> 2, wv = a / q4, mt = (a - wv * q4) >> 5; i = ((((wv * (A >> 7) + mt) * 4 + t) * 2 + s2) * 64 + hh2 * 32 + li); }
      }
      if (lp) { *reinterpret_cast<us8*>(hi + i * 8) = bq; continue; }
      *reinterpret_cast<half8*>(hi + i * 8) = h;
      *reinterpret_cast<half8*>(lo + i * 8) = l;
    }
  }
}
__global__ __launch_bounds__(256) void pack_split_weight_kernel(const float* __restrict__ w, const float* __restrict__ amax,
                                                                 _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                                 float* __restrict__ scale_out, int K, int C, int kd, int kh, int kw,
                                                                 int A, int B, int mode) {
  pack_split_body(w, amax, hi, lo, scale_out, K, C, kd, kh, kw, A, B, mode, (int)blockIdx.x, (int)gridDim.x);
}
// Multi-tensor forms: after an optimiser step EVERY weight needs a fresh amax and fresh packed planes; with ~100 (smoke) to
// ~300 (Burgers) operands that is hundreds of 5-microsecond launches. One launch each, grid = (blocks per item, items), the
// per-item arguments come from a device-resident table (wdno_amax_item / wdno_wsplit_item in wdno_hip.h).
__global__ __launch_bounds__(256) void amax_multi_kernel(const wdno_amax_item* __restrict__ tab) {
  __shared__ float red[4];
  const wdno_amax_item it = tab[blockIdx.y];
  const float* x = (const float*)it.x;
  const int64_t n = it.n;
  float m = 0.f;
  // weights sit at 4-byte-aligned offsets of the flat parameter buffer: up to three leading and trailing elements by block 0, the 16-byte-aligned
  // body as float4 with four loads in flight (4-byte loads: 180 us for the 563 MB of Burgers weights = 3.1 TB/s)
  int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) >> 2);
  if (head > n) head = n;
  const int64_t n4 = (n - head) >> 2, tail0 = head + 4 * n4;
  if (blockIdx.x == 0) {
    if ((int64_t)threadIdx.x < head) m = fabsf(x[threadIdx.x]);
    if (tail0 + (int64_t)threadIdx.x < n) m = fmaxf(m, fabsf(x[tail0 + threadIdx.x]));
  }
  const float4* x4 = reinterpret_cast<const float4*>(x + head);
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; k + 3 * stride < n4; k += 4 * stride) {
    const float4 a = x4[k], b = x4[k + stride], c = x4[k + 2 * stride], d = x4[k + 3 * stride];
    m = amax4(amax4(m, a), b);
    m = amax4(amax4(m, c), d);
  }
  for (; k < n4; k += stride) m = amax4(m, x4[k]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax((unsigned*)it.out, __float_as_uint(m));
  }
}
extern "C" int wdno_amax_multi(const void* table, int n_items, int blocks_per_item, wdno_stream_t s) {
  WDNO_REQUIRE(table && n_items > 0 && n_items <= 65535 && blocks_per_item > 0);
  amax_multi_kernel<<<dim3((unsigned)blocks_per_item, (unsigned)n_items), 256, 0, as_stream(s)>>>((const wdno_amax_item*)table);
  return wdno_check_launch();
}
__global__ __launch_bounds__(256) void pack_split_weight_multi_kernel(const wdno_wsplit_item* __restrict__ tab) {
  const wdno_wsplit_item it = tab[blockIdx.y];
  pack_split_body((const float*)it.w, (const float*)it.amax, (_Float16*)it.hi, (_Float16*)it.lo, (float*)it.scale_out, it.K, it.C, it.kd, it.kh,
                  it.kw, it.A, it.B, it.mode, (int)blockIdx.x, (int)gridDim.x);
}
extern "C" int wdno_pack_split_weight_multi(const void* table, int n_items, int blocks_per_item, wdno_stream_t s) {
  WDNO_REQUIRE(table && n_items > 0 && n_items <= 65535 && blocks_per_item > 0);
  pack_split_weight_multi_kernel<<<dim3((unsigned)blocks_per_item, (unsigned)n_items), 256, 0, as_stream(s)>>>((const wdno_wsplit_item*)table);
  return wdno_check_launch();
}
extern "C" int wdno_pack_split_weight(const float* w, const float* amax, void* hi, void* lo, float* scale_out, int K, int C, int kd, int kh,
                                      int kw, int A, int B, int mode, wdno_stream_t s) {
  WDNO_REQUIRE(K > 0 && C > 0 && kd > 0 && kh > 0 && kw > 0 && A > 0 && B > 0 && (B & 7) == 0 && mode >= 0 && mode <= 11);
  WDNO_REQUIRE(mode < 2 || mode > 5 || (kd == 1 && kh == 2 && kw == 2));      // parity classes of the (1,4,4) transposed convolution
  WDNO_REQUIRE(mode < 6 || (kd == 1 && kh == 1 && kw == 1));                  // taps of the (1,2,2) / stride 2 convolution
  WDNO_REQUIRE(mode != 1 && (mode < 6 || mode >= 10) ? (A >= K && B >= C) : (A >= C && B >= K));
  WDNO_REQUIRE(mode != 10 || (A == 384 && (B & 31) == 0 && lo != nullptr));      // fragment order: to_qkv of 4 heads of 32, to_out onto C = 128 m channels
  WDNO_REQUIRE(mode != 11 || (B == 128 && (A & 127) == 0 && lo != nullptr));
  WDNO_REQUIRE(lo == nullptr || (amax != nullptr && scale_out != nullptr));      // lo == NULL: one bf16 plane in `hi`
  int64_t total = (int64_t)kd * kh * A * kw * (B / 8);
  const PackTile ptile = pack_tile(kd * kh * kw, A, B, mode == 1 || (mode >= 6 && mode < 10) ? 1 : 0);
  if ((int64_t)kd * kh * kw * ptile.KT * ptile.CTp > PS_LDS_FLOATS + 16) return WDNO_EUNSUPPORTED;      // > 640 taps (every operand passes here first)
  (void)total;
  pack_split_weight_kernel<<<std::min(2048, ptile.tiles_k * ptile.tiles_c), 256, 0, as_stream(s)>>>(w, amax, (_Float16*)hi, (_Float16*)lo, scale_out, K, C, kd, kh, kw,
                                                                            A, B, mode);
  return wdno_check_launch();
}
