// attn_fused.h -- helpers shared by the fused temporal-attention kernels (attn_fused.hip forward, attn_fused_bwd.hip backward).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// Streamed weight fragments (attn_fused_wide.hip, linattn_fused_wide.hip): raw buffer loads, address = descriptor (SGPRs) + one loop-invariant
// 32-bit lane offset (VGPR) + a uniform offset (SGPR) -- no address arithmetic in the vector registers.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tf_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ half8 tf_frag(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
  return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uniform_bytes, 0));
}
// Hand-over of streamed fragments: a trip of the streaming loops requests two sets, waits for them IN FULL (s_waitcnt vmcnt(0)), then runs their
// matrix instructions -- no fragment load is in flight while the matrix pipe reads fragments. The overlapped form (the loads of set B in
// flight while the matrix instructions consume set A, handed over by the counted waits vmcnt(5), vmcnt(1), ... the compiler derives -- or
// by full waits) gave run-to-run differences of ~1e-6 in the first pass of the 256-channel linear-attention block whenever two blocks shared
// a CU: not with one block per CU, not with constant fragments or constant token planes, not with this serial form; unchanged by 64-bit lane
// vs buffer addressing, by extra barriers, by idle cycles behind the matrix instructions (tools/probes/lattn_wide_repro.py; the cause was not
// found -- the counted waits are correct for in-order returns). The other waves of the SIMD fill the wait: no time lost (measured), so every
// streaming loop of the wide kernels uses the serial form; tests/test_gpu_wide_repro.py repeats each kernel 40 times under two blocks per CU.
// The fragments are tied to the asm statements so that no consumer is scheduled above the wait and no later load above the consumers.
#define TF_WAIT_SET4(a, b, c, d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")
#define TF_WAIT_SET8(a, b, c, d, e, f, g, h) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "memory")

#define TF_C 64
#define TF_NT 24
#define TF_HEADS 4
#define TF_HD 128
#define TF_AST 72     /* halves per token row of an A plane (144 B) */
#define TF_VST 36     /* floats per row of the V tile */
#define TF_YST 68     /* floats per row of a partial output tile */
#define TF_RST 18     /* float2 per row of the rotary table (16 pairs + pad: the two lane halves start 2 pairs apart) */
#define TF_BST 28     /* floats per query row of the bias table (24 keys + pad: conflict-free 16-byte reads down a column of rows) */

__device__ __forceinline__ int tf_key(int m, int hh) { return 8 * (m >> 2) + 4 * hh + (m & 3); }

// Reductions without the LDS crossbar (a __shfl_xor is a ds_bpermute: ~100 cycles of latency each, and the LayerNorm of a row is a chain of
// eight of them): DPP operands inside a row of 16 lanes -- quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror -- leave the
// sum / maximum of the row in all of its lanes; v_permlane32_swap joins the two halves of the wave.
template <int CTRL>
__device__ __forceinline__ float tf_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float tf_row16_sum(float v) {
  v += tf_dpp<0xB1>(v); v += tf_dpp<0x4E>(v); v += tf_dpp<0x141>(v); v += tf_dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float tf_row16_max(float v) {
  v = fmaxf(v, tf_dpp<0xB1>(v)); v = fmaxf(v, tf_dpp<0x4E>(v)); v = fmaxf(v, tf_dpp<0x141>(v)); v = fmaxf(v, tf_dpp<0x140>(v));
  return v;
}
__device__ __forceinline__ float tf_wave_max(float v) {      // uniform result
  v = tf_row16_max(v);
  const unsigned u = __float_as_uint(v);
  const float a = __uint_as_float(__builtin_amdgcn_readlane(u, 0)), b = __uint_as_float(__builtin_amdgcn_readlane(u, 16));
  const float c = __uint_as_float(__builtin_amdgcn_readlane(u, 32)), d = __uint_as_float(__builtin_amdgcn_readlane(u, 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ void tf_halves(float v, float& lo, float& hi) {      // the value of lane (l & 31) and of lane (l & 31) + 32, in every lane
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}

// one row of 64 channels, 16 lanes x float4: LayerNorm (gain g) -> fp16 (hi, lo) planes of an LDS token tile (pitch TF_AST halves)
__device__ __forceinline__ void tf_ln_row(float4 xv, float4 g, float eps, float ps, _Float16* __restrict__ Ah, _Float16* __restrict__ Al, int row, int c4) {
  // norm.hip's layernorm_kernel (two-pass mean / variance over the 16 lanes of the row), the lane sums taken in DPP order
  const float mean = tf_row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / TF_C);
  xv.x -= mean; xv.y -= mean; xv.z -= mean; xv.w -= mean;
  const float var = tf_row16_sum((xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w)) * (1.0f / TF_C);
  const float rstd = 1.0f / sqrtf(var + eps);
  const float o[4] = {xv.x * rstd * g.x, xv.y * rstd * g.y, xv.z * rstd * g.z, xv.w * rstd * g.w};
  half4v h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = o[j] * ps;
    h[j] = (_Float16)t;
    l[j] = (_Float16)(t - (float)h[j]);
  }
  *reinterpret_cast<half4v*>(Ah + row * TF_AST + 4 * c4) = h;
  *reinterpret_cast<half4v*>(Al + row * TF_AST + 4 * c4) = l;
}

// launch parameters of the forward kernels (attn_fused.hip: 24 frames; attn_fused48.hip: 48)
struct TFusedP {
  const float* x; const float* gamma; float eps;
  const _Float16* wq_hi; const _Float16* wq_lo; const float* wq_scale;      // packed forward operand of to_qkv: [384][64]
  const _Float16* wo_hi; const _Float16* wo_lo; const float* wo_scale;      // ... of to_out: [64][128]
  const float* rcos; const float* rsin; const float* bias;                  // [24][32], [24][32], [4][24][24] (any may be null)
  float* y; float* amax_rec;
  float* qkv_out;                                                           // optional: raw projections [rows][384] (the un-fused backward reads them)
  float* rec_v;                                                             // optional amax record of v (attn_fused_bwd.hip: the plane scale of the attention output)
  int HW; float scale; int64_t nseq;
};
