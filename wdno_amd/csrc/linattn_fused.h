// linattn_fused.h -- shared by the fused SpatialLinearAttention forward kernels (linattn_fused.hip: 64 channels; linattn_fused_wide.hip: 128 / 256).
#pragma once
#include "attn_fused.h"

#define LF_PART (32 + 32 + 32 * 32)          /* floats per (block, head) of the first pass: m[32], Z[32], ctx_raw[32][32] */

struct LFusedP {
  const float* x; const float* gamma; float eps;
  const _Float16* wq_hi; const _Float16* wq_lo; const float* wq_scale;      // packed forward operand of to_qkv: [384][64]
  const _Float16* wo_hi; const _Float16* wo_lo; const float* wo_scale;      // ... of to_out: [64][128]
  const float* bias_out;
  float* part;               // first pass: [units][chunks][heads][LF_PART]
  const float* ctx;          // second pass: [units][heads][32][32] (ctx[d][e])
  float* y; float* amax_rec;
  int n_tok, chunks, tiles_per_chunk; float scale;
};

__device__ __forceinline__ f32x16 lf_zero() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}
__device__ __forceinline__ f32x16 lf_mfma3(half8 ah, half8 al, half8 bh, half8 bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
}
