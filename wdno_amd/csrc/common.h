// common.h -- shared host/device helpers for libwdno_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/wdno_hip.h"

extern thread_local hipError_t wdno_tls_last_hip_error;

static inline int wdno_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    wdno_tls_last_hip_error = e;
    return WDNO_ELAUNCH;
  }
  return WDNO_OK;
}

#define WDNO_REQUIRE(cond)            \
  do {                                \
    if (!(cond)) return WDNO_EINVAL;  \
  } while (0)

static inline hipStream_t as_stream(wdno_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// memory-bound kernels: cap the grid and grid-stride the rest (256 CUs x 8 blocks)
static inline int stream_grid(int64_t work_items, int block) {
  int64_t g = cdiv64(work_items, block);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// "amax record": WDNO_AMAX_SLOTS slots of WDNO_AMAX_STRIDE floats (one used float per 64-byte line), zeroed by the host,
// whose maximum is max|x| of a tensor. The kernel that writes a tensor can leave the record behind (one atomicMax per
// block), which saves the separate sweep the fp16 split would otherwise need to find its scale. The spread matters:
// atomics serialise per cache line at ~10 ns each -- 2048 blocks on ONE word cost a 78 MB sweep +19 us (12 -> 31 us), on
// 64 words of four lines +6 us, on 64 separate lines nothing measurable (tools/probes/atomic_probe.hip).
// Non-negative floats order like their bit patterns. Every thread of a 256-thread block calls amax_record_emit.
__device__ __forceinline__ void amax_record_emit(float m, float* __restrict__ rec, unsigned block_linear) {
  __shared__ float amax_red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) amax_red[(threadIdx.x >> 6) & 3] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
    atomicMax(reinterpret_cast<unsigned*>(rec) + (block_linear & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE, __float_as_uint(m));
  }
}
// the same for kernels whose waves work independently (convolution epilogues, one-wave-per-item attention): one atomic per
// wave; all 64 lanes must be active
__device__ __forceinline__ void wave_amax_emit(float am, float* rec, int wave_linear) {
  am = wave_max(am);
  if ((threadIdx.x & 63) == 0)
    atomicMax(reinterpret_cast<unsigned*>(rec) + (wave_linear & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE, __float_as_uint(am));
}
__device__ __forceinline__ float amax_record_read(const float* __restrict__ rec) {      // whole wave
  return wave_max(rec[(threadIdx.x & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE]);
}
__device__ __forceinline__ unsigned short bf16_rne(float v) {      // round-to-nearest-even, NaN kept quiet
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// four fp32 values -> four bf16 (8 bytes)
__device__ __forceinline__ uint2 bf16x4_pack(float4 v) {
  return make_uint2((unsigned)bf16_rne(v.x) | ((unsigned)bf16_rne(v.y) << 16), (unsigned)bf16_rne(v.z) | ((unsigned)bf16_rne(v.w) << 16));
}
// one value of a plane pair: (hi, lo) fp16 of v * s, or -- lo == nullptr, the single-product mode -- one bf16 in the hi plane's storage
__device__ __forceinline__ void plane_pack(float v, float s, bool lp, _Float16& h, _Float16& l) {
  if (lp) { const unsigned short b = bf16_rne(v); h = __builtin_bit_cast(_Float16, b); l = (_Float16)0.f; return; }
  const float t = v * s;
  h = (_Float16)t;
  l = (_Float16)(t - (float)h);
}
// scale of the (hi, lo) fp16 planes of a tensor with max|x| <= amax: the power of two that puts amax into [2^14, 2^15)
__device__ __forceinline__ float scale_from_amax(float amax) {
  if (!(amax > 0.f) || !isfinite(amax)) return 1.0f;
  int e = ilogbf(amax);                       // amax = m * 2^e, 1 <= m < 2
  return ldexpf(1.0f, 14 - e);
}
__device__ __forceinline__ float amax4(float m, float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// reductions inside aligned lane groups of width W (power of two <= 64)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// out[c] = sum_b part[b][c] for a [nb][C] partial-sum matrix: 32 columns x 8 row groups per block.
// Launch with PRS_THREADS threads: the partial matrices have up to 2048 rows (one per block of the producing sweep) but only a
// few dozen columns, so the few blocks of this kernel are latency chains -- 32 row groups x 4 independent sums keep a
// chain at nb/128 loads (256 threads x 1 sum: 12.9 us per launch at nb = 2048, ~90 launches per train step).
#define PRS_GROUPS 32
#define PRS_THREADS (32 * PRS_GROUPS)
// (stride: elements between consecutive rows of `part`, >= C: a window of C columns of a wider matrix -- wdno_rows_sum_multi)
template <typename T>
__device__ __forceinline__ void partial_rows_sum_body(const T* __restrict__ part, float* __restrict__ out, int nb, int C, int bx, int stride = 0) {
  __shared__ double red[PRS_GROUPS][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = bx * 32 + tx;
  if (stride == 0) stride = C;
  double acc = 0.0;
  if (c < C) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = ty;
    for (; b + 3 * PRS_GROUPS < nb; b += 4 * PRS_GROUPS) {
      a0 += (double)part[(int64_t)b * stride + c];
      a1 += (double)part[(int64_t)(b + PRS_GROUPS) * stride + c];
      a2 += (double)part[(int64_t)(b + 2 * PRS_GROUPS) * stride + c];
      a3 += (double)part[(int64_t)(b + 3 * PRS_GROUPS) * stride + c];
    }
    for (; b < nb; b += PRS_GROUPS) a0 += (double)part[(int64_t)b * stride + c];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < C) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < PRS_GROUPS; ++g) a[g & 3] += red[g][tx];
    out[c] = (float)((a[0] + a[1]) + (a[2] + a[3]));
  }
}
template <typename T>
__global__ __launch_bounds__(PRS_THREADS) void partial_rows_sum_kernel(const T* __restrict__ part, float* __restrict__ out, int nb, int C) {
  partial_rows_sum_body<T>(part, out, nb, C, (int)blockIdx.x);
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
  float sg = 1.0f / (1.0f + expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}
#endif
