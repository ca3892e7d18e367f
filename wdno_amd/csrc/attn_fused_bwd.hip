// attn_fused_bwd.hip -- backward of the fused temporal attention block of attn_fused.hip (conv3d.py:165-174 LayerNorm, :277-353 Attention,
// wrapped as Residual(PreNorm(...))), ONE launch:
//
//     y = x + W_out . softmax(rot(scale q) rot(k)^T + bias) v ,      (q | k | v) = W_qkv . LayerNorm(x)
//     given dy:  dx, dgamma, dW_qkv, dW_out, dbias
//
// Nothing but x was saved by the forward: LayerNorm, the projections, the scores and the attention output are recomputed per sequence (the
// 24 frames of one pixel), so the [pixels x 384] qkv / dqkv tensors (472 MB each at the bench size, moved eight times by the layer-by-layer
// backward) never exist. Arithmetic as in the forward and the convolutions: projections and weight gradients on v_mfma_f32_32x32x16_f16 with
// the three-product (hi, lo) split, the score-sized products on the exact-fp32 v_mfma_f32_32x32x2_f32.
//
// One block per CU, four waves = the four heads of one sequence at a time, sequences walked with a grid stride.
//   * W_qkv's planes live in LDS for the whole kernel (96 KB, 16-byte chunks XOR-swizzled by the row pair): read row-wise they are the A
//     fragments of the projection (q^T = W x^T), read with the transpose read ds_read_b64_tr_b16 they are the A fragments of
//     dxn^T = W^T dqkv^T -- one image, both products. W_out^T's fragments (32 registers) stay in registers.
//   * Every product that contracts over features has its operands in accumulator layout already (a lane owns a token and 16 features):
//     S^T = K Q^T and dP^T = V dO^T take both operands in place. Products that contract over tokens take one operand (v, k, q, dO) as a
//     column of a 32 x 32 fp32 LDS tile and, for dK / dV, the other (dS, P) as a row of a transposed tile -- two tiles per head, reused.
//   * Weight gradients contract over the tokens of BOTH operands: the gradient tile (dq, dk, dv, O) is written as fp16 planes [token][32]
//     into the head's second tile and both operands come back through transpose reads ([token][channel] images, no transposed copies).
//     dW_qkv (96 registers per lane) and dW_out (32) are MFMA accumulators over all sequences of the block, never touched by the VALU (they
//     live in the AGPR half of the register file): the planes of dy, O and dq / dk / dv are written at a RUNNING power-of-two scale -- the
//     scale of the largest tile seen so far, i.e. what one scale per tensor (the layer-by-layer path) gives -- and when a larger tile
//     arrives the accumulators are multiplied by the ratio of the scales (exact, a power of two; a handful of times per launch).
//   * Every block writes ONE partial [dW_qkv | dW_out | dgamma | dbias]; tattn_fused_reduce_kernel adds the partials in block order
//     (bit-reproducible: no atomics anywhere).
#include "attn_fused.h"

#define TB_TS 36                      /* floats per row of an fp32 tile */
#define TB_PS 36                      /* halves per row of a per-head plane tile (same bytes as half an fp32 tile) */
#define TB_N_WQ (3 * TF_HD * TF_C)    /* 24576 */
#define TB_N_WO (TF_C * TF_HD)        /* 8192 */
#define TB_OFF_WO TB_N_WQ
#define TB_OFF_DG (TB_N_WQ + TB_N_WO)
#define TB_OFF_DB (TB_OFF_DG + TF_C)
#define TB_E (TB_OFF_DB + TF_HEADS * TF_NT * TF_NT)      /* 35136 floats per partial */

// LDS map (bytes)
#define TB_L_WH 0
#define TB_L_WL 49152
#define TB_L_XH 98304
#define TB_L_XL (TB_L_XH + 4608)
#define TB_L_GH (TB_L_XL + 4608)
#define TB_L_GL (TB_L_GH + 4608)
#define TB_L_T (TB_L_GL + 4608)              /* 116736: per head 2 x 4608 */
#define TB_L_RT (TB_L_T + TF_HEADS * 9216)   /* 153600 */
#define TB_L_WM (TB_L_RT + 4608)             /* 158208 */
#define TB_LDS_BYTES (TB_L_WM + 16)

#define TB_FENCE() asm volatile("" ::: "memory")

struct TFusedBwdP {
  const float* x; const float* dy; const float* gamma; float eps;
  const _Float16* wq_hi; const _Float16* wq_lo; const float* wq_scale;      // packed forward operand of to_qkv: [384][64]
  const _Float16* wo_hi; const _Float16* wo_lo; const float* wo_scale;      // ... of to_out: [64][128]
  const float* rcos; const float* rsin; const float* bias;                  // [24][32], [24][32], [4][24][24] (any may be null)
  float* dx; float* amax_rec; float* part;
  int HW; float scale; int64_t nseq;
};

typedef short tb_short4 __attribute__((ext_vector_type(4)));
typedef short tb_short8 __attribute__((ext_vector_type(8)));
typedef tb_short4 __attribute__((address_space(3))) * tb_lds_s4;

__device__ __forceinline__ half8 tb_tr2(const _Float16* p0, const _Float16* p1) {
  const tb_short4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s4)(p0));
  const tb_short4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s4)(p1));
  const tb_short8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8, c);
}
// operand fragment from a [token][channel] image: lane (li, hh) receives channel ch0 + li of tokens tok0 + 8 hh + (0..7)
// (conv_h3.hip: tr_frag; inside a 16-lane group lanes 4 j .. 4 j + 3 point at the four 8-byte pieces of row j)
__device__ __forceinline__ half8 tb_trf(const _Float16* tile, int stride, int tok0, int ch0, int lane) {
  const int g = lane >> 4, xl = lane & 15;
  const _Float16* p0 = tile + (tok0 + 8 * (g >> 1) + (xl >> 2)) * stride + ch0 + 16 * (g & 1) + 4 * (xl & 3);
  return tb_tr2(p0, p0 + 4 * stride);
}
// halves offset of the 16-byte chunk `chunk` of row f in a swizzled W plane
__device__ __forceinline__ int tb_woff(int f, int chunk) { return f * TF_C + ((chunk ^ ((f >> 1) & 7)) << 3); }
// W^T fragment for dxn^T[c][tok] = sum_f W[f][c] d[tok][f]: lane (li, hh) receives channel 32 ct + li of the rows f0 + 4 hh + (0..3) and
// f0 + 8 + 4 hh + (0..3) -- the features a lane half holds in accumulator registers 8 s .. 8 s + 7 when f0 = base + 16 s
__device__ __forceinline__ half8 tb_wtr(const _Float16* W, int f0, int ct, int lane) {
  const int g = lane >> 4, xl = lane & 15;
  const int ra = f0 + 4 * (g >> 1) + (xl >> 2), rb = ra + 8;
  const int col = 32 * ct + 16 * (g & 1) + 4 * (xl & 3);
  return tb_tr2(W + tb_woff(ra, col >> 3) + (col & 7), W + tb_woff(rb, col >> 3) + (col & 7));
}
__device__ __forceinline__ f32x16 tb_zero() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}
__device__ __forceinline__ f32x16 tb_mfma3(half8 ah, half8 al, half8 bh, half8 bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
}
__device__ __forceinline__ float tb_absmax16(const f32x16& v) {
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(v[e]));
  return m;
}
// accumulator tile X^T[feature e][token li] -> fp32 tile [token][32]
__device__ __forceinline__ void tb_acc_to_tile(float* __restrict__ T, const f32x16& v, int li, int hh) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
    *reinterpret_cast<float4*>(T + li * TB_TS + 8 * c + 4 * hh) = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}
// accumulator tile S^T[key e][query li] -> fp32 tile [key][query] (the lane roles swap when a lane reads ITS row)
__device__ __forceinline__ void tb_acc_to_tile_t(float* __restrict__ T, const f32x16& v, int li, int hh) {
#pragma unroll
  for (int e = 0; e < 16; ++e) T[tf_key(e, hh) * TB_TS + li] = v[e];
}
// D^T[d][j] = sum_{token t < 24} T[t][d] b[t][j]: T columns (lane = d), b = an accumulator tile in place (register m <-> token tf_key(m, hh))
__device__ __forceinline__ f32x16 tb_col_product(const float* __restrict__ T, const f32x16& b, int li, int hh) {
  f32x16 acc = tb_zero();
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) {
    float a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = T[(8 * g4 + 4 * hh + q) * TB_TS + li];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[4 * g4 + q], acc, 0, 0, 0);
  }
  return acc;
}
// D^T[d][j] = sum_{token t < 24} Ta[t][d] Tb[j][t]: Ta columns (lane = d), Tb rows (lane = j)
__device__ __forceinline__ f32x16 tb_row_product(const float* __restrict__ Ta, const float* __restrict__ Tb, int li, int hh) {
  f32x16 acc = tb_zero();
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) {
    float a[4];
    const float4 b4 = *reinterpret_cast<const float4*>(Tb + li * TB_TS + 8 * g4 + 4 * hh);
    const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = Ta[(8 * g4 + 4 * hh + q) * TB_TS + li];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
  }
  return acc;
}
// the 16 accumulator values of a lane as (hi, lo) halves at scale s: k-step s' of a product that contracts over the features takes
// elements 8 s' .. 8 s' + 7; also written as planes [token li][32 features] for the transpose reads of the weight-gradient products
__device__ __forceinline__ void tb_split16(const f32x16& v, float s, half8 (&h)[2], half8 (&l)[2], _Float16* __restrict__ Ph, _Float16* __restrict__ Pl,
                                           int li, int hh) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float t = v[e] * s;
    const _Float16 th = (_Float16)t;
    h[e >> 3][e & 7] = th;
    l[e >> 3][e & 7] = (_Float16)(t - (float)th);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    half4v a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = h[c >> 1][4 * (c & 1) + j]; b[j] = l[c >> 1][4 * (c & 1) + j]; }
    *reinterpret_cast<half4v*>(Ph + li * TB_PS + 8 * c + 4 * hh) = a;
    *reinterpret_cast<half4v*>(Pl + li * TB_PS + 8 * c + 4 * hh) = b;
  }
}
// two weight-gradient tiles  acc[ct][feature][c] += inv * sum_tok P[tok][feature] X[tok][32 ct + c]   (P: the head's plane tile, X: a shared image)
__device__ __forceinline__ void tb_dw_pair(f32x16& acc0, f32x16& acc1, const _Float16* Ph, const _Float16* Pl, const _Float16* Xh, const _Float16* Xl,
                                           int lane) {
  half8 ah[2], al[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) { ah[s] = tb_trf(Ph, TB_PS, 16 * s, 0, lane); al[s] = tb_trf(Pl, TB_PS, 16 * s, 0, lane); }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    f32x16& acc = ct ? acc1 : acc0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const half8 bh = tb_trf(Xh, TF_AST, 16 * s, 32 * ct, lane), bl = tb_trf(Xl, TF_AST, 16 * s, 32 * ct, lane);
      acc = tb_mfma3(ah[s], al[s], bh, bl, acc);
    }
  }
}
// dxn^T[c][tok] += inv * sum_f W[fbase + f][c] d[tok][f] over the head's 32 features of one of q / k / v
__device__ __forceinline__ void tb_dxn(f32x16& d0, f32x16& d1, const _Float16* WH, const _Float16* WL, int fbase, const half8 (&h)[2], const half8 (&l)[2],
                                       int lane) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    f32x16& acc = ct ? d1 : d0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const half8 wh = tb_wtr(WH, fbase + 16 * s, ct, lane), wl = tb_wtr(WL, fbase + 16 * s, ct, lane);
      acc = tb_mfma3(wh, wl, h[s], l[s], acc);
    }
  }
}
__device__ __forceinline__ void tb_rescale(f32x16& a, float r) {
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] *= r;
}
// power-of-two plane scale for a tile of maximum `amax`, never above 2^100 (a denormal maximum must not turn into an infinite scale)
__device__ __forceinline__ float tb_scale(float amax) { return fminf(scale_from_amax(amax), 0x1p100f); }
// LayerNorm of one row by its 16 lanes (norm.hip: layernorm_kernel), planes written, mean and 1/std returned
__device__ __forceinline__ void tb_ln_row(float4 xv, float4 g, float eps, float ps, _Float16* __restrict__ Ah, _Float16* __restrict__ Al, int row, int c4,
                                          float& mean, float& rstd) {
  mean = tf_row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / TF_C);
  xv.x -= mean; xv.y -= mean; xv.z -= mean; xv.w -= mean;
  const float var = tf_row16_sum((xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w)) * (1.0f / TF_C);
  rstd = 1.0f / sqrtf(var + eps);
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
__device__ __forceinline__ void tb_plane_row(float4 v, float s, _Float16* __restrict__ Ah, _Float16* __restrict__ Al, int row, int c4) {
  const float o[4] = {v.x, v.y, v.z, v.w};
  half4v h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = o[j] * s;
    h[j] = (_Float16)t;
    l[j] = (_Float16)(t - (float)h[j]);
  }
  *reinterpret_cast<half4v*>(Ah + row * TF_AST + 4 * c4) = h;
  *reinterpret_cast<half4v*>(Al + row * TF_AST + 4 * c4) = l;
}
// gradient of the rotation: accumulator pairs (2 j, 2 j + 1) of token li; table rows as in the forward
__device__ __forceinline__ void tb_unrotate(f32x16& v, const float2* __restrict__ Rt, int li, int hh, float mul) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 r4 = *reinterpret_cast<const float4*>(Rt + li * TF_RST + 4 * c + 2 * hh);
    const float cs2[2] = {r4.x, r4.z}, sn2[2] = {r4.y, r4.w};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = 2 * c + q;
      const float gx = v[2 * j], gy = v[2 * j + 1];
      v[2 * j] = (gx * cs2[q] + gy * sn2[q]) * mul;
      v[2 * j + 1] = (gy * cs2[q] - gx * sn2[q]) * mul;
    }
  }
}

__global__ __launch_bounds__(256, 1) void tattn_fused_bwd_kernel(TFusedBwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tb_smem[];
  _Float16* WH = reinterpret_cast<_Float16*>(tb_smem + TB_L_WH);
  _Float16* WL = reinterpret_cast<_Float16*>(tb_smem + TB_L_WL);
  _Float16* XH = reinterpret_cast<_Float16*>(tb_smem + TB_L_XH);
  _Float16* XL = reinterpret_cast<_Float16*>(tb_smem + TB_L_XL);
  _Float16* GH = reinterpret_cast<_Float16*>(tb_smem + TB_L_GH);
  _Float16* GL = reinterpret_cast<_Float16*>(tb_smem + TB_L_GL);
  float2* Rt = reinterpret_cast<float2*>(tb_smem + TB_L_RT);
  float* WM = reinterpret_cast<float*>(tb_smem + TB_L_WM);
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  float* T0 = reinterpret_cast<float*>(tb_smem + TB_L_T + h * 9216);
  float* T1 = T0 + 32 * TB_TS;
  _Float16* PH = reinterpret_cast<_Float16*>(T1);
  _Float16* PL = PH + 32 * TB_PS;
  float* Yp = T0;                                          // [24][TF_YST] partial dxn of this head (6528 B of the head's 9216)

  // ---- once per kernel: W_qkv planes -> LDS, zeroed images, rotary table, W_out^T fragments, bias rows, LayerNorm gain
  for (int q = tid; q < 3 * TF_HD * 8; q += 256) {
    const int f = q >> 3, ch = q & 7;
    const int dst = tb_woff(f, ch);
    *reinterpret_cast<uint4*>(WH + dst) = *reinterpret_cast<const uint4*>(p.wq_hi + f * TF_C + ch * 8);
    *reinterpret_cast<uint4*>(WL + dst) = *reinterpret_cast<const uint4*>(p.wq_lo + f * TF_C + ch * 8);
  }
  for (int i = tid; i < (TB_L_RT - TB_L_XH) / 16; i += 256) reinterpret_cast<uint4*>(tb_smem + TB_L_XH)[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < 32 * 16; i += 256) {
    const int t = i >> 4, j = i & 15;
    float2 v = make_float2(1.f, 0.f);
    if (p.rcos && t < TF_NT) v = make_float2(p.rcos[t * 32 + 2 * j], p.rsin[t * 32 + 2 * j]);
    Rt[t * TF_RST + j] = v;
  }
  // dO^T[d][tok] = sum_c W_out[c][32 h + d] dy[tok][c]: A fragment of k-step s = channels 16 s + 8 hh + (0..7) of column 32 h + li
  half8 woth[4], wotl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int off = (16 * s + 8 * hh + t) * TF_HD + 32 * h + li;
      woth[s][t] = p.wo_hi[off];
      wotl[s][t] = p.wo_lo[off];
    }
  float bs[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) bs[e] = (p.bias && li < TF_NT) ? p.bias[(h * TF_NT + li) * TF_NT + tf_key(e, hh)] : 0.f;
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));        // |LayerNorm(x)| <= sqrt(64) max|g|
  const float wq_s = p.wq_scale[0], wo_s = p.wo_scale[0];
  const float inv_qkv = 1.0f / (ps * wq_s);
  const int64_t fstride = (int64_t)p.HW * TF_C;

  f32x16 dwq[3][2], dwo[2];
#pragma unroll
  for (int ti = 0; ti < 3; ++ti) { dwq[ti][0] = tb_zero(); dwq[ti][1] = tb_zero(); }
  dwo[0] = tb_zero(); dwo[1] = tb_zero();
  float dbacc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) dbacc[e] = 0.f;
  float4 dgacc = make_float4(0.f, 0.f, 0.f, 0.f);
  float am = 0.f;
  // running plane scales (powers of two, only ever decreasing): dy (block-uniform), O and dq / dk / dv (per wave)
  float sc_g = 0x1p100f, sc_o = 0x1p100f, sc_d = 0x1p100f;
  __syncthreads();

  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0, ng0 = nx0, ng1 = nx0;
  auto fetch = [&](int64_t r0, float4& x0, float4& x1, float4& g0, float4& g1) {
    const float* xr = p.x + r0 * TF_C;
    const float* gr = p.dy + r0 * TF_C;
    x0 = *reinterpret_cast<const float4*>(xr + lrow * fstride + 4 * lc4);
    g0 = *reinterpret_cast<const float4*>(gr + lrow * fstride + 4 * lc4);
    if (lrow < 8) {
      x1 = *reinterpret_cast<const float4*>(xr + (16 + lrow) * fstride + 4 * lc4);
      g1 = *reinterpret_cast<const float4*>(gr + (16 + lrow) * fstride + 4 * lc4);
    }
  };
  int nb = (int)(blockIdx.x / (unsigned)p.HW), npix = (int)(blockIdx.x - (unsigned)nb * (unsigned)p.HW);
  const int gstep_b = (int)(gridDim.x / (unsigned)p.HW), gstep_p = (int)(gridDim.x - (unsigned)gstep_b * (unsigned)p.HW);
  if ((int64_t)blockIdx.x < p.nseq) fetch((int64_t)nb * TF_NT * p.HW + npix, nx0, nx1, ng0, ng1);
  for (int64_t seq = blockIdx.x; seq < p.nseq; seq += gridDim.x) {
    const int64_t row0 = (int64_t)nb * TF_NT * p.HW + npix;
    nb += gstep_b; npix += gstep_p;
    if (npix >= p.HW) { npix -= p.HW; ++nb; }
    // ---- rows -> LayerNorm -> planes of xn; max|dy| of the sequence. Only (mean, 1/std) of the two rows stay in registers: x and dy are
    // read again (L2) for the last phase, together with the rows of the next sequence.
    float mean0, mean1, rs0, rs1;
    tb_ln_row(nx0, g4, p.eps, ps, XH, XL, lrow, lc4, mean0, rs0);
    tb_ln_row(nx1, g4, p.eps, ps, XH, XL, 16 + lrow, lc4, mean1, rs1);
    {
      const float gm = tf_wave_max(amax4(amax4(0.f, ng0), ng1));
      if (lane == 0) WM[h] = gm;
    }
    __syncthreads();                                                          // B1: xn planes, wave maxima
    {
      const float need = tb_scale(fmaxf(fmaxf(WM[0], WM[1]), fmaxf(WM[2], WM[3])));
      if (need < sc_g) {
        const float r = need / sc_g;
        tb_rescale(dwo[0], r); tb_rescale(dwo[1], r);
        sc_g = need;
      }
    }
    tb_plane_row(ng0, sc_g, GH, GL, lrow, lc4);
    tb_plane_row(ng1, sc_g, GH, GL, 16 + lrow, lc4);
    // ---- (q | k | v)^T of this head
    f32x16 aq = tb_zero(), ak = tb_zero(), av = tb_zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 bh = *reinterpret_cast<const half8*>(XH + li * TF_AST + 16 * s + 8 * hh);
      const half8 bl = *reinterpret_cast<const half8*>(XL + li * TF_AST + 16 * s + 8 * hh);
      const int o0 = tb_woff(h * 32 + li, 2 * s + hh);                       // rows + 128, + 256: same swizzle term ((f >> 1) & 7 has period 16)
      aq = tb_mfma3(*reinterpret_cast<const half8*>(WH + o0), *reinterpret_cast<const half8*>(WL + o0), bh, bl, aq);
      ak = tb_mfma3(*reinterpret_cast<const half8*>(WH + o0 + TF_HD * TF_C), *reinterpret_cast<const half8*>(WL + o0 + TF_HD * TF_C), bh, bl, ak);
      av = tb_mfma3(*reinterpret_cast<const half8*>(WH + o0 + 2 * TF_HD * TF_C), *reinterpret_cast<const half8*>(WL + o0 + 2 * TF_HD * TF_C), bh, bl, av);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { aq[e] *= inv_qkv; ak[e] *= inv_qkv; av[e] *= inv_qkv; }
    {
      const float need = tb_scale(tf_wave_max(tb_absmax16(av)));              // rows of P sum to 1: |O| <= max|v|
      if (need < sc_o) {
        const float r = need / sc_o;
        tb_rescale(dwo[0], r); tb_rescale(dwo[1], r);
        sc_o = need;
      }
    }
    // q * scale, rotary on q and k
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 r4 = *reinterpret_cast<const float4*>(Rt + li * TF_RST + 4 * c + 2 * hh);
      const float cs2[2] = {r4.x, r4.z}, sn2[2] = {r4.y, r4.w};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = 2 * c + q;
        const float qx = aq[2 * j] * p.scale, qy = aq[2 * j + 1] * p.scale;
        aq[2 * j] = qx * cs2[q] - qy * sn2[q];
        aq[2 * j + 1] = qy * cs2[q] + qx * sn2[q];
        const float kx = ak[2 * j], ky = ak[2 * j + 1];
        ak[2 * j] = kx * cs2[q] - ky * sn2[q];
        ak[2 * j + 1] = ky * cs2[q] + kx * sn2[q];
      }
    }
    __syncthreads();                                                          // B2: dy planes
    // ---- dO^T = W_out^T dy^T (this head's 32 columns)
    f32x16 dOT = tb_zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 bh = *reinterpret_cast<const half8*>(GH + li * TF_AST + 16 * s + 8 * hh);
      const half8 bl = *reinterpret_cast<const half8*>(GL + li * TF_AST + 16 * s + 8 * hh);
      dOT = tb_mfma3(woth[s], wotl[s], bh, bl, dOT);
    }
    {
      const float inv_do = 1.0f / (sc_g * wo_s);
#pragma unroll
      for (int e = 0; e < 16; ++e) dOT[e] *= inv_do;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- P^T = softmax(S^T), S^T = K Q^T (exact fp32)
    f32x16 sT = tb_zero();
#pragma unroll
    for (int e = 0; e < 16; ++e) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[e], aq[e], sT, 0, 0, 0);
    {
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 12; ++e) { sT[e] += bs[e]; mx = fmaxf(mx, sT[e]); }
      float m0, m1;
      tf_halves(mx, m0, m1);
      mx = fmaxf(m0, m1);
      float l = 0.f;
#pragma unroll
      for (int e = 0; e < 12; ++e) { sT[e] = expf(sT[e] - mx); l += sT[e]; }
#pragma unroll
      for (int e = 12; e < 16; ++e) sT[e] = 0.f;
      float l0, l1;
      tf_halves(l, l0, l1);
      const float il = 1.0f / (l0 + l1);
#pragma unroll
      for (int e = 0; e < 12; ++e) sT[e] *= il;
    }
    // ---- dP^T = V dO^T, dS^T = P^T (dP^T - delta), delta_i = sum_j P_ij dP_ij
    f32x16 dsT = tb_zero();
#pragma unroll
    for (int e = 0; e < 16; ++e) dsT = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], dOT[e], dsT, 0, 0, 0);
    {
      float delta = 0.f;
#pragma unroll
      for (int e = 0; e < 12; ++e) delta = fmaf(sT[e], dsT[e], delta);
      float d0, d1;
      tf_halves(delta, d0, d1);
      delta = d0 + d1;
#pragma unroll
      for (int e = 0; e < 12; ++e) { dsT[e] = sT[e] * (dsT[e] - delta); dbacc[e] += dsT[e]; }
#pragma unroll
      for (int e = 12; e < 16; ++e) dsT[e] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 dx0 = tb_zero(), dx1 = tb_zero();                                  // dxn^T of this head at scale sc_d * wq_s: channels 0..31, 32..63
    half8 dh[2], dl[2];
    // a gradient tile larger than every one before: the accumulators that carry sc_d move to the new scale
    auto fit_d = [&](float amax) {
      const float need = tb_scale(amax);
      if (need < sc_d) {
        const float r = need / sc_d;
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) { tb_rescale(dwq[ti][0], r); tb_rescale(dwq[ti][1], r); }
        tb_rescale(dx0, r); tb_rescale(dx1, r);
        sc_d = need;
      }
    };
    // ---- O^T = V^T P^T -> dW_out
    tb_acc_to_tile(T0, av, li, hh);
    TB_FENCE();
    {
      const f32x16 oT = tb_col_product(T0, sT, li, hh);
      tb_split16(oT, sc_o, dh, dl, PH, PL, li, hh);
      TB_FENCE();
      // dW_out[c][32 h + d] += sum_tok dy[tok][c] O[tok][d]: rows = channels (A = dy image), columns = d (B = the O planes)
      half8 bh[2], bl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) { bh[s] = tb_trf(PH, TB_PS, 16 * s, 0, lane); bl[s] = tb_trf(PL, TB_PS, 16 * s, 0, lane); }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8 ah = tb_trf(GH, TF_AST, 16 * s, 32 * ct, lane), al = tb_trf(GL, TF_AST, 16 * s, 32 * ct, lane);
          dwo[ct] = tb_mfma3(ah, al, bh[s], bl[s], dwo[ct]);
        }
    }
    TB_FENCE();
    __builtin_amdgcn_sched_barrier(0);
    // ---- dQ'^T = K'^T dS^T -> dq -> dW_q, dxn
    tb_acc_to_tile(T0, ak, li, hh);
    TB_FENCE();
    {
      f32x16 dq = tb_col_product(T0, dsT, li, hh);
      tb_unrotate(dq, Rt, li, hh, p.scale);
      fit_d(tf_wave_max(tb_absmax16(dq)));
      tb_split16(dq, sc_d, dh, dl, PH, PL, li, hh);
      TB_FENCE();
      tb_dw_pair(dwq[0][0], dwq[0][1], PH, PL, XH, XL, lane);
      tb_dxn(dx0, dx1, WH, WL, h * 32, dh, dl, lane);
    }
    TB_FENCE();
    __builtin_amdgcn_sched_barrier(0);
    // ---- dK'^T = Q'^T dS -> dk
    tb_acc_to_tile(T0, aq, li, hh);
    tb_acc_to_tile_t(T1, dsT, li, hh);
    TB_FENCE();
    {
      f32x16 dk = tb_row_product(T0, T1, li, hh);
      tb_unrotate(dk, Rt, li, hh, 1.0f);
      fit_d(tf_wave_max(tb_absmax16(dk)));
      TB_FENCE();
      tb_split16(dk, sc_d, dh, dl, PH, PL, li, hh);
      TB_FENCE();
      tb_dw_pair(dwq[1][0], dwq[1][1], PH, PL, XH, XL, lane);
      tb_dxn(dx0, dx1, WH, WL, TF_HD + h * 32, dh, dl, lane);
    }
    TB_FENCE();
    __builtin_amdgcn_sched_barrier(0);
    // the rows of this sequence again (for the LayerNorm backward and the residual gradient) and those of the next one: in flight under dV
    float4 cx0 = make_float4(0.f, 0.f, 0.f, 0.f), cx1 = cx0, cg0 = cx0, cg1 = cx0;
    fetch(row0, cx0, cx1, cg0, cg1);
    if (seq + gridDim.x < p.nseq) fetch((int64_t)nb * TF_NT * p.HW + npix, nx0, nx1, ng0, ng1);
    // ---- dV^T = dO^T P -> dv
    tb_acc_to_tile(T0, dOT, li, hh);
    tb_acc_to_tile_t(T1, sT, li, hh);
    TB_FENCE();
    {
      const f32x16 dv = tb_row_product(T0, T1, li, hh);
      fit_d(tf_wave_max(tb_absmax16(dv)));
      TB_FENCE();
      tb_split16(dv, sc_d, dh, dl, PH, PL, li, hh);
      TB_FENCE();
      tb_dw_pair(dwq[2][0], dwq[2][1], PH, PL, XH, XL, lane);
      tb_dxn(dx0, dx1, WH, WL, 2 * TF_HD + h * 32, dh, dl, lane);
    }
    TB_FENCE();
    // ---- the head's part of dxn as [token][channel]
    if (li < TF_NT) {
      const float inv = 1.0f / (sc_d * wq_s);
      float* yp = Yp + li * TF_YST + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(yp + 8 * c) = make_float4(dx0[4 * c] * inv, dx0[4 * c + 1] * inv, dx0[4 * c + 2] * inv, dx0[4 * c + 3] * inv);
        *reinterpret_cast<float4*>(yp + 32 + 8 * c) = make_float4(dx1[4 * c] * inv, dx1[4 * c + 1] * inv, dx1[4 * c + 2] * inv, dx1[4 * c + 3] * inv);
      }
    }
    __syncthreads();                                                          // B3
    // ---- heads summed, LayerNorm backward, residual gradient added, rows stored (the lanes that loaded a row finish it)
    float* db = p.dx + row0 * TF_C;
    const float* Y0 = reinterpret_cast<const float*>(tb_smem + TB_L_T);
    auto finish = [&](int row, const float4& xr, float mean, float rstd, const float4& gy) {
      const int o = row * TF_YST + 4 * lc4;
      const float4 a = *reinterpret_cast<const float4*>(Y0 + o), b2 = *reinterpret_cast<const float4*>(Y0 + 2304 + o);
      const float4 c = *reinterpret_cast<const float4*>(Y0 + 4608 + o), d = *reinterpret_cast<const float4*>(Y0 + 6912 + o);
      const float4 xh = make_float4((xr.x - mean) * rstd, (xr.y - mean) * rstd, (xr.z - mean) * rstd, (xr.w - mean) * rstd);
      float4 dn;                                                              // dxn of this row
      dn.x = (a.x + b2.x) + (c.x + d.x); dn.y = (a.y + b2.y) + (c.y + d.y);
      dn.z = (a.z + b2.z) + (c.z + d.z); dn.w = (a.w + b2.w) + (c.w + d.w);
      dgacc.x += dn.x * xh.x; dgacc.y += dn.y * xh.y; dgacc.z += dn.z * xh.z; dgacc.w += dn.w * xh.w;
      dn.x *= g4.x; dn.y *= g4.y; dn.z *= g4.z; dn.w *= g4.w;
      const float m1 = tf_row16_sum((dn.x + dn.y) + (dn.z + dn.w)) * (1.0f / TF_C);
      const float m2 = tf_row16_sum((dn.x * xh.x + dn.y * xh.y) + (dn.z * xh.z + dn.w * xh.w)) * (1.0f / TF_C);
      float4 r;
      r.x = rstd * (dn.x - m1 - xh.x * m2) + gy.x; r.y = rstd * (dn.y - m1 - xh.y * m2) + gy.y;
      r.z = rstd * (dn.z - m1 - xh.z * m2) + gy.z; r.w = rstd * (dn.w - m1 - xh.w * m2) + gy.w;
      *reinterpret_cast<float4*>(db + row * fstride + 4 * lc4) = r;
      am = amax4(am, r);
    };
    finish(lrow, cx0, mean0, rs0, cg0);
    if (lrow < 8) finish(16 + lrow, cx1, mean1, rs1, cg1);
  }
  // ---- this block's partial sums
  float* part = p.part + (size_t)blockIdx.x * TB_E;
  const float inv_wq = 1.0f / (sc_d * ps), inv_wo = 1.0f / (sc_g * sc_o);
#pragma unroll
  for (int ti = 0; ti < 3; ++ti)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) part[(ti * TF_HD + h * 32 + tf_key(e, hh)) * TF_C + 32 * ct + li] = dwq[ti][ct][e] * inv_wq;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int e = 0; e < 16; ++e) part[TB_OFF_WO + (32 * ct + tf_key(e, hh)) * TF_HD + h * 32 + li] = dwo[ct][e] * inv_wo;
  if (li < TF_NT) {
#pragma unroll
    for (int e = 0; e < 12; ++e) part[TB_OFF_DB + (h * TF_NT + li) * TF_NT + tf_key(e, hh)] = dbacc[e];
  }
  __syncthreads();                                     // the tiles are free: dgamma over the 16 row groups of the block, in row-group order
  float* red = reinterpret_cast<float*>(tb_smem + TB_L_T);
  *reinterpret_cast<float4*>(red + lrow * TF_C + 4 * lc4) = dgacc;
  __syncthreads();
  if (tid < TF_C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r * TF_C + tid];
    part[TB_OFF_DG + tid] = t;
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * TF_HEADS + h);
}

// out[e] = sum over the blocks' partials in block order: 32 outputs x 8 chains per block, four independent sums per chain
__global__ __launch_bounds__(256) void tattn_fused_reduce_kernel(const float* __restrict__ part, int nb, float* __restrict__ out, int E) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < E) {
    int b = q;
    for (; b + 24 < nb; b += 32) {
      a0 += part[(size_t)b * E + e]; a1 += part[(size_t)(b + 8) * E + e];
      a2 += part[(size_t)(b + 16) * E + e]; a3 += part[(size_t)(b + 24) * E + e];
    }
    for (; b < nb; b += 8) a0 += part[(size_t)b * E + e];
  }
  red[q][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0 && e < E) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][el];
    out[e] = t;
  }
}

static int tb_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

extern "C" size_t wdno_tattn_fused_bwd_ws_bytes(void) { return (size_t)tb_num_cus() * TB_E * sizeof(float); }
extern "C" int wdno_tattn_fused_bwd_grads(void) { return TB_E; }

extern "C" int wdno_tattn_fused_bwd(const float* x, const float* dy, const float* gamma, float eps, const void* wq_hi, const void* wq_lo,
                                    const float* wq_scale, const void* wo_hi, const void* wo_lo, const float* wo_scale, const float* rot_cos,
                                    const float* rot_sin, const float* bias, float* dx, float* amax_rec, float* grads, void* ws, size_t ws_bytes,
                                    int64_t n_batch, int n_tok, int64_t hw, int C, int heads, float scale, wdno_stream_t s) {
  WDNO_REQUIRE(x && dy && gamma && wq_hi && wq_lo && wq_scale && wo_hi && wo_lo && wo_scale && dx && grads && ws && n_batch > 0 && hw > 0);
  WDNO_REQUIRE((rot_cos == nullptr) == (rot_sin == nullptr));
  if (!wdno_tattn_fused_takes(C, n_tok, heads) || hw > 0x7fffffff / (TF_C * TF_NT)) return WDNO_EUNSUPPORTED;
  if (ws_bytes < wdno_tattn_fused_bwd_ws_bytes()) return WDNO_EWORKSPACE;
  TFusedBwdP p;
  p.x = x; p.dy = dy; p.gamma = gamma; p.eps = eps;
  p.wq_hi = (const _Float16*)wq_hi; p.wq_lo = (const _Float16*)wq_lo; p.wq_scale = wq_scale;
  p.wo_hi = (const _Float16*)wo_hi; p.wo_lo = (const _Float16*)wo_lo; p.wo_scale = wo_scale;
  p.rcos = rot_cos; p.rsin = rot_sin; p.bias = bias;
  p.dx = dx; p.amax_rec = amax_rec; p.part = (float*)ws;
  p.HW = (int)hw; p.scale = scale; p.nseq = n_batch * hw;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)tattn_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES) != hipSuccess) return WDNO_ELAUNCH;
    attr_done = true;
  }
  int64_t grid = tb_num_cus();
  if (grid > p.nseq) grid = p.nseq;
  tattn_fused_bwd_kernel<<<(int)grid, 256, TB_LDS_BYTES, as_stream(s)>>>(p);
  int rc = wdno_check_launch();
  if (rc) return rc;
  tattn_fused_reduce_kernel<<<(TB_E + 31) / 32, 256, 0, as_stream(s)>>>((const float*)ws, (int)grid, grads, TB_E);
  return wdno_check_launch();
}
