// EXPERIMENT (round 5, NOT part of the library): attn_fused_bwd.hip with conflict-free LDS layouts for its transpose reads -- W planes and the xn / dy images
// as 128-byte rows whose 16-byte chunks are XOR-swizzled so that rows R .. R + 3 of a ds_read_b64_tr_b16 cover the 64 banks once, plane tiles as 64-byte rows with
// swizzled 8-byte pieces (see the comment block above tb_sw). Result: the fragment addresses' variants stop being constant offsets from one lane-dependent base, and
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I ../../wdno_amd/csrc -Rpass-analysis=kernel-resource-usage -x hip -c attn_fused_bwd_swizzled.hip
// reports VGPRs 256, AGPRs 256, ScratchSize 100 B / lane, VGPRs Spill 30 (the shipped kernel: 256 / 254 / 0 / 0). See DESIGN.md section 6, round 5. (bash tools/probes/attn_fused_bwd_swizzled.sh prints the report.)
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
//     live in the AGPR half of the register file), so the planes of a tensor carry ONE power-of-two scale at a time: dy and O the scale
//     of their exact maxima (the amax record of dy; max|v| recorded by the forward launch: rows of P sum to 1), dq / dk / dv a RUNNING
//     scale per wave -- the scale of the largest tile seen so far, i.e. what one scale per tensor (the layer-by-layer path) gives; when a
//     larger tile arrives the two accumulator tiles of that tensor are multiplied by the ratio of the scales (exact, a power of two; a
//     handful of times per launch). (Scales from upper BOUNDS of dq / dk / dv -- 2^14 .. 2^18 above the true maxima -- cost 2e-6 of
//     accuracy: the lo plane falls into fp16's subnormal range. Measured, dropped.)
//   * Order inside a head: the operands of a product are requested (LDS) before the matrix instructions of the PREVIOUS product are issued,
//     so one wave per SIMD keeps its matrix pipe busy across the LDS round trips (three tiles per head: column operand, transposed
//     operand, planes).
//   * Every block writes ONE partial [dW_qkv | dW_out | dgamma | dbias]; tattn_fused_reduce_kernel adds the partials in block order
//     (bit-reproducible: no atomics anywhere).
#include <stdlib.h>
#include "attn_fused.h"

#define TB_TS 36                      /* floats per row of an fp32 tile */
#define TB_PS 32                      /* halves per row of a per-head plane tile (8-byte pieces XOR-swizzled by the row pair) */
#define TB_N_WQ (3 * TF_HD * TF_C)    /* 24576 */
#define TB_N_WO (TF_C * TF_HD)        /* 8192 */
#define TB_OFF_WO TB_N_WQ
#define TB_OFF_DG (TB_N_WQ + TB_N_WO)
#define TB_OFF_DB (TB_OFF_DG + TF_C)
#define TB_E (TB_OFF_DB + TF_HEADS * TF_NT * TF_NT)      /* 35136 floats per partial */

// LDS map (bytes). Images hold the 24 token rows only: a lane whose token / reduction rows would be 24 .. 31 reads the zero block instead.
#define TB_IMG (TF_NT * TF_C * 2)            /* 3072: one plane of xn / dy, [24][64] halves, 16-byte chunks XOR-swizzled like the W planes */
#define TB_TILE (TF_NT * TB_TS * 4)          /* 3456: an fp32 tile; also the two planes [24][36] of a head's plane tile */
#define TB_L_WH 0
#define TB_L_WL 49152
#define TB_L_XH 98304
#define TB_L_XL (TB_L_XH + TB_IMG)
#define TB_L_GH (TB_L_XL + TB_IMG)
#define TB_L_GL (TB_L_GH + TB_IMG)
#define TB_L_T (TB_L_GL + TB_IMG)                    /* per head: T0, T1, PL */
#define TB_HEAD_LDS (3 * TB_TILE)
#define TB_L_RT (TB_L_T + TF_HEADS * TB_HEAD_LDS)
#define TB_L_ZB (TB_L_RT + 4608)                     /* 256 bytes of zeros */
#define TB_LDS_BYTES (TB_L_ZB + 256)

#define TB_FENCE() asm volatile("" ::: "memory")

struct TFusedBwdP {
  const float* x; const float* dy; const float* gamma; float eps;
  const _Float16* wq_hi; const _Float16* wq_lo; const float* wq_scale;      // packed forward operand of to_qkv: [384][64]
  const _Float16* wo_hi; const _Float16* wo_lo; const float* wo_scale;      // packed DATA-GRADIENT operand of to_out: W_out^T [128][64]
  const float* rcos; const float* rsin; const float* bias;                  // [24][32], [24][32], [4][24][24] (any may be null)
  const float* rec_dy; const float* rec_v;                                  // amax records: dy; v of the forward launch
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
// LDS layouts (all conflict-free for BOTH access kinds they see -- profiles/r04_smoke_pmc.md had 1.4e7 SQ_LDS_BANK_CONFLICT per dispatch, all of
// it from the transpose reads of the padded round-4 layouts):
//   * W planes [384][64 halves] and the xn / dy images [24][64 halves]: rows of 128 B, the 16-byte chunk c of row r stored at chunk c ^ tb_sw(r).
//     Row-wise ds_read_b128 (16-lane groups, all lanes the same chunk): the 8 row pairs of a group have 8 different tb_sw -> 16 slots.
//     Transpose reads (32 lanes = rows R .. R + 3, R % 4 == 0, 64 B of each): rows R, R + 1 keep the 64-byte half of the chunk index, rows
//     R + 2, R + 3 take the other one (bit 2 of tb_sw is bit 1 of the row) -> the four rows cover the 64 banks exactly once.
//   * per-head plane tiles [24][32 halves]: rows of 64 B, the 8-byte piece q of row r at piece q ^ ((r >> 1) & 7). ds_write_b64 (16 contiguous
//     lanes = 16 rows, same piece): 16 different (row parity, piece) slots; transpose reads: four whole rows = 256 contiguous bytes.
__device__ __forceinline__ int tb_sw(int r) { const int x = (r >> 1) & 7; return ((x & 1) << 2) | (x >> 1); }
// halves offset of the 16-byte chunk `chunk` of row f in a swizzled W plane / image
__device__ __forceinline__ int tb_woff(int f, int chunk) { return f * TF_C + ((chunk ^ tb_sw(f)) << 3); }
// halves offset of the 8-byte piece `piece` of row r in a plane tile
__device__ __forceinline__ int tb_poff(int r, int piece) { return r * TB_PS + ((piece ^ ((r >> 1) & 7)) << 2); }
// operand fragment from a [token][channel] image of 24 rows: lane (li, hh) receives channel ch0 + li of tokens 16 s + 8 hh + (0..7)
// (conv_h3.hip: tr_frag; inside a 16-lane group lanes 4 j .. 4 j + 3 point at the four 8-byte pieces of row j). Tokens 24 .. 31 -- the
// upper lane half of step s = 1 -- come from the zero block.
template <int S>
__device__ __forceinline__ half8 tb_trf(const _Float16* img, int ch0, int lane, const _Float16* zb) {
  const int g = lane >> 4, xl = lane & 15;
  const int row = 16 * S + 8 * (g >> 1) + (xl >> 2), col = ch0 + 16 * (g & 1) + 4 * (xl & 3);
  const _Float16* p0 = img + tb_woff(row, col >> 3) + (col & 7);
  const _Float16* p1 = img + tb_woff(row + 4, col >> 3) + (col & 7);
  if (S == 1 && (g >> 1)) { p0 = zb; p1 = zb; }
  return tb_tr2(p0, p1);
}
// ... from a plane tile [24][32]: lane (li, hh) receives feature li of tokens 16 s + 8 hh + (0..7)
template <int S>
__device__ __forceinline__ half8 tb_trp(const _Float16* tile, int lane, const _Float16* zb) {
  const int g = lane >> 4, xl = lane & 15;
  const int row = 16 * S + 8 * (g >> 1) + (xl >> 2), piece = 4 * (g & 1) + (xl & 3);
  const _Float16* p0 = tile + tb_poff(row, piece);
  const _Float16* p1 = tile + tb_poff(row + 4, piece);
  if (S == 1 && (g >> 1)) { p0 = zb; p1 = zb; }
  return tb_tr2(p0, p1);
}
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
// accumulator tile X^T[feature e][token li] -> fp32 tile [token][32] (tokens < 24)
__device__ __forceinline__ void tb_acc_to_tile(float* __restrict__ T, const f32x16& v, int li, int hh) {
  if (li < TF_NT) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<float4*>(T + li * TB_TS + 8 * c + 4 * hh) = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  }
}
// accumulator tile S^T[key e][query li] -> fp32 tile [key < 24][query] (the lane roles swap when a lane reads ITS row)
__device__ __forceinline__ void tb_acc_to_tile_t(float* __restrict__ T, const f32x16& v, int li, int hh) {
#pragma unroll
  for (int e = 0; e < 12; ++e) T[tf_key(e, hh) * TB_TS + li] = v[e];
}
// the 12 values of column li a lane feeds to a product over the tokens (step m <-> token tf_key(m, hh))
__device__ __forceinline__ void tb_cols(const float* __restrict__ T, int li, int hh, float (&c)[12]) {
#pragma unroll
  for (int m = 0; m < 12; ++m) c[m] = T[tf_key(m, hh) * TB_TS + li];
}
// ... and of row `row` of a transposed tile
__device__ __forceinline__ void tb_rows(const float* __restrict__ T, int row, int hh, float (&r)[12]) {
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) {
    const float4 b4 = *reinterpret_cast<const float4*>(T + row * TB_TS + 8 * g4 + 4 * hh);
    r[4 * g4] = b4.x; r[4 * g4 + 1] = b4.y; r[4 * g4 + 2] = b4.z; r[4 * g4 + 3] = b4.w;
  }
}
// D^T[d][j] = sum_{token t < 24} a[t][d] b[t][j] on the exact-fp32 matrix instruction
template <typename B>
__device__ __forceinline__ f32x16 tb_product12(const float (&a)[12], const B& b) {
  f32x16 acc = tb_zero();
#pragma unroll
  for (int m = 0; m < 12; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[m], acc, 0, 0, 0);
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
  if (li < TF_NT) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      half4v a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = h[c >> 1][4 * (c & 1) + j]; b[j] = l[c >> 1][4 * (c & 1) + j]; }
      *reinterpret_cast<half4v*>(Ph + tb_poff(li, 2 * c + hh)) = a;
      *reinterpret_cast<half4v*>(Pl + tb_poff(li, 2 * c + hh)) = b;
    }
  }
}
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
  *reinterpret_cast<half4v*>(Ah + tb_woff(row, c4 >> 1) + 4 * (c4 & 1)) = h;
  *reinterpret_cast<half4v*>(Al + tb_woff(row, c4 >> 1) + 4 * (c4 & 1)) = l;
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
  *reinterpret_cast<half4v*>(Ah + tb_woff(row, c4 >> 1) + 4 * (c4 & 1)) = h;
  *reinterpret_cast<half4v*>(Al + tb_woff(row, c4 >> 1) + 4 * (c4 & 1)) = l;
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
// power-of-two plane scale for a tensor bounded by `bound`, kept inside [2^-100, 2^100]
__device__ __forceinline__ float tb_scale(float bound) { return fminf(fmaxf(scale_from_amax(bound), 0x1p-100f), 0x1p100f); }

// w *= r without a VALU instruction touching the accumulator (a value the VALU multiplies has to live in the architectural half of the
// register file for its whole life -- 128 such registers spill) and IN PLACE (a fresh result tile merged back at the end of a rare branch
// costs the allocator ~100 registers): sixteen accumulating steps of the exact-fp32 matrix instruction, step e adding (r - 1) * (the two
// rows accumulator register e holds) -- row operand (r - 1) * unit vector, column operand the accumulator register itself. One rounding
// per entry (r is a power of two, (r - 1) w is not exactly representable): 2^-24 relative, a handful of times per launch.
__device__ __forceinline__ void tb_rescale(f32x16& w, float r, int li, int hh) {
  const float r1 = r - 1.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) w = __builtin_amdgcn_mfma_f32_32x32x2f32(li == tf_key(e, hh) ? r1 : 0.f, w[e], w, 0, 0, 0);
}
// a gradient tile larger than every one before it: the tensor's two weight-gradient tiles move to the new scale (exact: a power of two)
__device__ __forceinline__ void tb_fit(float& sc, float amax, f32x16& w0, f32x16& w1, int li, int hh) {
  const float need = tb_scale(amax);
  if (need < sc) {
    const float r = need / sc;
    tb_rescale(w0, r, li, hh);
    tb_rescale(w1, r, li, hh);
    sc = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(need)));
  }
}
__device__ __forceinline__ float tb_absmax16(const f32x16& v) {
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(v[e]));
  return m;
}

// ABL: timing ablations (tools/tattn_ablate.sh, WDNO_TB_ABLATE, built with -DWDNO_TB_ABLATIONS): 1 = no weight-gradient / dxn products, 2 = no
// score-sized fp32 products, 3 = no block barriers, 4 = 1 + 2, 5 = 4 without the projections (results are wrong in all of them)
template <int ABL>
__global__ __launch_bounds__(256, 1) void tattn_fused_bwd_kernel(TFusedBwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tb_smem[];
  _Float16* WH = reinterpret_cast<_Float16*>(tb_smem + TB_L_WH);
  _Float16* WL = reinterpret_cast<_Float16*>(tb_smem + TB_L_WL);
  _Float16* XH = reinterpret_cast<_Float16*>(tb_smem + TB_L_XH);
  _Float16* XL = reinterpret_cast<_Float16*>(tb_smem + TB_L_XL);
  _Float16* GH = reinterpret_cast<_Float16*>(tb_smem + TB_L_GH);
  _Float16* GL = reinterpret_cast<_Float16*>(tb_smem + TB_L_GL);
  float2* Rt = reinterpret_cast<float2*>(tb_smem + TB_L_RT);
  const _Float16* ZB = reinterpret_cast<const _Float16*>(tb_smem + TB_L_ZB);
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  float* T0 = reinterpret_cast<float*>(tb_smem + TB_L_T + h * TB_HEAD_LDS);
  float* T1 = T0 + TF_NT * TB_TS;
  _Float16* PH = reinterpret_cast<_Float16*>(T1 + TF_NT * TB_TS);
  _Float16* PL = PH + TF_NT * TB_PS;
  float* Yp = T0;                                          // [24][TF_YST] partial dxn of this head (6528 B of T0 + T1's 6912)

  // ---- once per kernel: W_qkv planes -> LDS, zeroed images, rotary table, W_out^T fragments, LayerNorm gain, plane scales
  for (int q = tid; q < 3 * TF_HD * 8; q += 256) {
    const int f = q >> 3, ch = q & 7;
    const int dst = tb_woff(f, ch);
    *reinterpret_cast<uint4*>(WH + dst) = *reinterpret_cast<const uint4*>(p.wq_hi + f * TF_C + ch * 8);
    *reinterpret_cast<uint4*>(WL + dst) = *reinterpret_cast<const uint4*>(p.wq_lo + f * TF_C + ch * 8);
  }
  for (int i = tid; i < (TB_LDS_BYTES - TB_L_XH) / 16; i += 256) reinterpret_cast<uint4*>(tb_smem + TB_L_XH)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  for (int i = tid; i < 32 * 16; i += 256) {
    const int t = i >> 4, j = i & 15;
    float2 v = make_float2(1.f, 0.f);
    if (p.rcos && t < TF_NT) v = make_float2(p.rcos[t * 32 + 2 * j], p.rsin[t * 32 + 2 * j]);
    Rt[t * TF_RST + j] = v;
  }
  // dO^T[d][tok] = sum_c W_out[c][32 h + d] dy[tok][c]: the A fragment of k-step s = channels 16 s + 8 hh + (0..7) of row 32 h + li of
  // W_out^T, 16 bytes of the packed data-gradient operand; fetched per sequence (L1 / L2) -- 32 registers held for the whole kernel spill
  const _Float16* wot_h = p.wo_hi + (32 * h + li) * TF_C + 8 * hh;
  const _Float16* wot_l = p.wo_lo + (32 * h + li) * TF_C + 8 * hh;
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));        // |LayerNorm(x)| <= sqrt(64) max|g|
  const float wq_s = p.wq_scale[0], wo_s = p.wo_scale[0];
  const float inv_qkv = 1.0f / (ps * wq_s);
  // plane scales: dy and O fixed for the launch (exact maxima), dq / dk / dv running (powers of two, only ever decreasing)
  const float sc_g = tb_scale(amax_record_read(p.rec_dy)), sc_o = tb_scale(amax_record_read(p.rec_v));
  float sc_q = 0x1p100f, sc_k = 0x1p100f, sc_v = 0x1p100f;
  const float inv_do = 1.0f / (sc_g * wo_s);
  const int64_t fstride = (int64_t)p.HW * TF_C;
  // LDS addresses of this lane's operand rows (lanes of tokens 24 .. 31 read zeros)
  // chunk 2 s + hh of this lane's token row, s = 0 .. 3: the XOR with tb_sw(li) permutes the four s among themselves and hh with it
  int xoff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) xoff[s] = li < TF_NT ? tb_woff(li, 2 * s + hh) : (int)(TB_L_ZB - TB_L_XH) / 2 + 8 * hh;      // halves from XH; XL = + TB_IMG bytes
  const int goff = (int)(TB_L_GH - TB_L_XH) / 2;                                          // the dy image from the xn image (zero-block lanes: 0)
  const int xl_off = li < TF_NT ? TB_IMG / 2 : 0;                                          // halves from the hi to the lo plane
  const int g_off = li < TF_NT ? goff : 0;
  const int trow = li < TF_NT ? li : li - 8;                                               // row of a transposed tile this lane reads

  f32x16 dwq[3][2], dwo[2];
#pragma unroll
  for (int ti = 0; ti < 3; ++ti) { dwq[ti][0] = tb_zero(); dwq[ti][1] = tb_zero(); }
  dwo[0] = tb_zero(); dwo[1] = tb_zero();
  float dbacc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) dbacc[e] = 0.f;
  float4 dgacc = make_float4(0.f, 0.f, 0.f, 0.f);
  float am = 0.f;
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
    // ---- rows -> LayerNorm -> planes of xn; planes of dy. Only (mean, 1/std) of the two rows stay in registers: x and dy are read again
    // (L2) for the last phase, together with the rows of the next sequence.
    float mean0, mean1 = 0.f, rs0, rs1 = 0.f;
    tb_ln_row(nx0, g4, p.eps, ps, XH, XL, lrow, lc4, mean0, rs0);
    tb_plane_row(ng0, sc_g, GH, GL, lrow, lc4);
    if (lrow < 8) {
      tb_ln_row(nx1, g4, p.eps, ps, XH, XL, 16 + lrow, lc4, mean1, rs1);
      tb_plane_row(ng1, sc_g, GH, GL, 16 + lrow, lc4);
    }
    // the bias rows of this lane's query (L1 / L2: 96 bytes per lane), wanted after the first score product
    float bs[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && li < TF_NT) b4 = *reinterpret_cast<const float4*>(p.bias + (h * TF_NT + li) * TF_NT + 8 * c + 4 * hh);
      bs[4 * c] = b4.x; bs[4 * c + 1] = b4.y; bs[4 * c + 2] = b4.z; bs[4 * c + 3] = b4.w;
    }
    half8 woth[4], wotl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      woth[s] = *reinterpret_cast<const half8*>(wot_h + 16 * s);
      wotl[s] = *reinterpret_cast<const half8*>(wot_l + 16 * s);
    }
    if (ABL != 3) __syncthreads();                                            // B1: planes of xn and dy
    // ---- (q | k | v)^T of this head, dO^T = W_out^T dy^T (this head's 32 columns)
    f32x16 aq = tb_zero(), ak = tb_zero(), av = tb_zero(), dOT = tb_zero();
#pragma unroll
    for (int s = 0; s < (ABL == 5 ? 0 : 4); ++s) {
      const half8 bh = *reinterpret_cast<const half8*>(XH + xoff[s]);
      const half8 bl = *reinterpret_cast<const half8*>(XH + xoff[s] + xl_off);
      const int o0 = tb_woff(h * 32 + li, 2 * s + hh);                       // rows + 128, + 256: same swizzle term ((f >> 1) & 7 has period 16)
      aq = tb_mfma3(*reinterpret_cast<const half8*>(WH + o0), *reinterpret_cast<const half8*>(WL + o0), bh, bl, aq);
      ak = tb_mfma3(*reinterpret_cast<const half8*>(WH + o0 + TF_HD * TF_C), *reinterpret_cast<const half8*>(WL + o0 + TF_HD * TF_C), bh, bl, ak);
      av = tb_mfma3(*reinterpret_cast<const half8*>(WH + o0 + 2 * TF_HD * TF_C), *reinterpret_cast<const half8*>(WL + o0 + 2 * TF_HD * TF_C), bh, bl, av);
    }
#pragma unroll
    for (int s = 0; s < (ABL == 5 ? 0 : 4); ++s) {
      const half8 bh = *reinterpret_cast<const half8*>(XH + xoff[s] + g_off);
      const half8 bl = *reinterpret_cast<const half8*>(XH + xoff[s] + g_off + xl_off);
      dOT = tb_mfma3(woth[s], wotl[s], bh, bl, dOT);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { aq[e] *= inv_qkv; ak[e] *= inv_qkv; av[e] *= inv_qkv; dOT[e] *= inv_do; }
    tb_acc_to_tile(T0, av, li, hh);                                           // T0 = v
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
    TB_FENCE();
    // ---- S^T = K Q^T and dP^T = V dO^T (exact fp32, operands in place); under them: the columns of v, then T0 = k
    f32x16 sT = tb_zero(), dsT = tb_zero();
    float cv[12];
    tb_cols(T0, li, hh, cv);
    TB_FENCE();
    tb_acc_to_tile(T0, ak, li, hh);                                           // T0 = k'
#pragma unroll
    for (int e = 0; e < 16; ++e) if (!(ABL == 2 || ABL >= 4)) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[e], aq[e], sT, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) if (!(ABL == 2 || ABL >= 4)) dsT = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], dOT[e], dsT, 0, 0, 0);
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
    TB_FENCE();
    float ck[12];
    tb_cols(T0, li, hh, ck);
    TB_FENCE();
    tb_acc_to_tile(T0, aq, li, hh);                                           // T0 = q'
    // ---- O^T = V^T P^T; under it: dS^T = P^T (dP^T - delta), delta_i = sum_j P_ij dP_ij
    f32x16 oT = (ABL == 2 || ABL >= 4) ? sT : tb_product12(cv, sT);
    {
      float delta = 0.f;
#pragma unroll
      for (int e = 0; e < 12; ++e) delta = fmaf(sT[e], dsT[e], delta);
      float d0, d1;
      tf_halves(delta, d0, d1);
      delta = d0 + d1;
#pragma unroll
      for (int e = 0; e < 12; ++e) { dsT[e] = sT[e] * (dsT[e] - delta); dbacc[e] += dsT[e]; }
    }
    tb_acc_to_tile_t(T1, dsT, li, hh);                                        // T1 = dS (lane roles swapped)
    TB_FENCE();
    float cq[12], rS[12];
    tb_cols(T0, li, hh, cq);
    tb_rows(T1, trow, hh, rS);
    TB_FENCE();
    tb_acc_to_tile(T0, dOT, li, hh);                                          // T0 = dO
    tb_acc_to_tile_t(T1, sT, li, hh);                                         // T1 = P (lane roles swapped)
    // ---- dQ'^T = K'^T dS^T; under it: the planes of O
    f32x16 dq = (ABL == 2 || ABL >= 4) ? dsT : tb_product12(ck, dsT);
    half8 oh[2], ol[2];
    tb_split16(oT, sc_o, oh, ol, PH, PL, li, hh);
    TB_FENCE();
    half8 bo_h[2], bo_l[2];
    bo_h[0] = tb_trp<0>(PH, lane, ZB); bo_l[0] = tb_trp<0>(PL, lane, ZB);
    bo_h[1] = tb_trp<1>(PH, lane, ZB); bo_l[1] = tb_trp<1>(PL, lane, ZB);
    float cdo[12], rP[12];
    tb_cols(T0, li, hh, cdo);
    tb_rows(T1, trow, hh, rP);
    TB_FENCE();
    // ---- dK'^T = Q'^T dS; under it: dq un-rotated, its planes
    f32x16 dk = (ABL == 2 || ABL >= 4) ? dsT : tb_product12(cq, rS);
    tb_unrotate(dq, Rt, li, hh, p.scale);
    tb_fit(sc_q, tf_wave_max(tb_absmax16(dq)), dwq[0][0], dwq[0][1], li, hh);
    half8 qh[2], ql[2];
    tb_split16(dq, sc_q, qh, ql, PH, PL, li, hh);                              // (after the reads of O's planes: LDS keeps a wave's order)
    TB_FENCE();
    // ---- dW_out[c][32 h + d] += sum_tok dy[tok][c] O[tok][d]: rows = channels (A = dy image), columns = d (B = the O planes)
    if (ABL != 1 && ABL < 4) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        dwo[ct] = tb_mfma3(tb_trf<0>(GH, 32 * ct, lane, ZB), tb_trf<0>(GL, 32 * ct, lane, ZB), bo_h[0], bo_l[0], dwo[ct]);
        dwo[ct] = tb_mfma3(tb_trf<1>(GH, 32 * ct, lane, ZB), tb_trf<1>(GL, 32 * ct, lane, ZB), bo_h[1], bo_l[1], dwo[ct]);
      }
    }
    // the rows of this sequence again (for the LayerNorm backward and the residual gradient) and those of the next one: in flight from here
    float4 cx0 = make_float4(0.f, 0.f, 0.f, 0.f), cx1 = cx0, cg0 = cx0, cg1 = cx0;
    fetch(row0, cx0, cx1, cg0, cg1);
    if (seq + gridDim.x < p.nseq) fetch((int64_t)nb * TF_NT * p.HW + npix, nx0, nx1, ng0, ng1);
    // ---- dV^T = dO^T P; under it: the operands of dW_q / dxn_q
    f32x16 dv = (ABL == 2 || ABL >= 4) ? sT : tb_product12(cdo, rP);
    f32x16 dxs0 = tb_zero(), dxs1 = tb_zero();                                // dxn^T of this head, channels 0..31 / 32..63 (fp32 sum of the three tensors' parts)
    // weight-gradient tiles  dW[feature][32 ct + c] += sum_tok P[tok][feature] xn[tok][c]  and  dxn^T[c][tok] += sum_f W[f][c] d[tok][f]
    auto grad_products = [&](f32x16& w0, f32x16& w1, float sc, int fbase, const half8 (&dh)[2], const half8 (&dl)[2]) {
      if (ABL == 1 || ABL >= 4) return;
      f32x16 d0 = tb_zero(), d1 = tb_zero();
      const half8 a0h = tb_trp<0>(PH, lane, ZB), a0l = tb_trp<0>(PL, lane, ZB);
      const half8 a1h = tb_trp<1>(PH, lane, ZB), a1l = tb_trp<1>(PL, lane, ZB);
      w0 = tb_mfma3(a0h, a0l, tb_trf<0>(XH, 0, lane, ZB), tb_trf<0>(XL, 0, lane, ZB), w0);
      w0 = tb_mfma3(a1h, a1l, tb_trf<1>(XH, 0, lane, ZB), tb_trf<1>(XL, 0, lane, ZB), w0);
      w1 = tb_mfma3(a0h, a0l, tb_trf<0>(XH, 32, lane, ZB), tb_trf<0>(XL, 32, lane, ZB), w1);
      w1 = tb_mfma3(a1h, a1l, tb_trf<1>(XH, 32, lane, ZB), tb_trf<1>(XL, 32, lane, ZB), w1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        d0 = tb_mfma3(tb_wtr(WH, fbase + 16 * s, 0, lane), tb_wtr(WL, fbase + 16 * s, 0, lane), dh[s], dl[s], d0);
        d1 = tb_mfma3(tb_wtr(WH, fbase + 16 * s, 1, lane), tb_wtr(WL, fbase + 16 * s, 1, lane), dh[s], dl[s], d1);
      }
      const float inv = 1.0f / (sc * wq_s);
#pragma unroll
      for (int e = 0; e < 16; ++e) { dxs0[e] = fmaf(d0[e], inv, dxs0[e]); dxs1[e] = fmaf(d1[e], inv, dxs1[e]); }
    };
    grad_products(dwq[0][0], dwq[0][1], sc_q, h * 32, qh, ql);
    TB_FENCE();
    tb_unrotate(dk, Rt, li, hh, 1.0f);
    tb_fit(sc_k, tf_wave_max(tb_absmax16(dk)), dwq[1][0], dwq[1][1], li, hh);
    tb_split16(dk, sc_k, qh, ql, PH, PL, li, hh);
    TB_FENCE();
    grad_products(dwq[1][0], dwq[1][1], sc_k, TF_HD + h * 32, qh, ql);
    TB_FENCE();
    tb_fit(sc_v, tf_wave_max(tb_absmax16(dv)), dwq[2][0], dwq[2][1], li, hh);
    tb_split16(dv, sc_v, qh, ql, PH, PL, li, hh);
    TB_FENCE();
    grad_products(dwq[2][0], dwq[2][1], sc_v, 2 * TF_HD + h * 32, qh, ql);
    TB_FENCE();
    // ---- the head's part of dxn as [token][channel]
    if (li < TF_NT) {
      float* yp = Yp + li * TF_YST + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(yp + 8 * c) = make_float4(dxs0[4 * c], dxs0[4 * c + 1], dxs0[4 * c + 2], dxs0[4 * c + 3]);
        *reinterpret_cast<float4*>(yp + 32 + 8 * c) = make_float4(dxs1[4 * c], dxs1[4 * c + 1], dxs1[4 * c + 2], dxs1[4 * c + 3]);
      }
    }
    if (ABL != 3) __syncthreads();                                            // B2
    // ---- heads summed, LayerNorm backward, residual gradient added, rows stored (the lanes that loaded a row finish it)
    float* db = p.dx + row0 * TF_C;
    const float* Y0 = reinterpret_cast<const float*>(tb_smem + TB_L_T);
    auto finish = [&](int row, const float4& xr, float mean, float rstd, const float4& gy) {
      const int o = row * TF_YST + 4 * lc4;
      const float4 a = *reinterpret_cast<const float4*>(Y0 + o), b2 = *reinterpret_cast<const float4*>(Y0 + TB_HEAD_LDS / 4 + o);
      const float4 c = *reinterpret_cast<const float4*>(Y0 + 2 * (TB_HEAD_LDS / 4) + o), d = *reinterpret_cast<const float4*>(Y0 + 3 * (TB_HEAD_LDS / 4) + o);
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
    if (ABL != 3) __syncthreads();                                            // B3: the partial tiles (= T0, T1 of the next sequence) are free
  }
  // ---- this block's partial sums
  float* part = p.part + (size_t)blockIdx.x * TB_E;
  {
    const float inv_w[3] = {1.0f / (sc_q * ps), 1.0f / (sc_k * ps), 1.0f / (sc_v * ps)};
    const float inv_wo = 1.0f / (sc_g * sc_o);
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) part[(ti * TF_HD + h * 32 + tf_key(e, hh)) * TF_C + 32 * ct + li] = dwq[ti][ct][e] * inv_w[ti];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) part[TB_OFF_WO + (32 * ct + tf_key(e, hh)) * TF_HD + h * 32 + li] = dwo[ct][e] * inv_wo;
  }
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
                                    const float* rot_sin, const float* bias, const float* rec_dy, const float* rec_v, float* dx, float* amax_rec,
                                    float* grads, void* ws, size_t ws_bytes,
                                    int64_t n_batch, int n_tok, int64_t hw, int C, int heads, float scale, wdno_stream_t s) {
  WDNO_REQUIRE(x && dy && gamma && wq_hi && wq_lo && wq_scale && wo_hi && wo_lo && wo_scale && dx && grads && ws && n_batch > 0 && hw > 0);
  WDNO_REQUIRE(rec_dy && rec_v);
  WDNO_REQUIRE((rot_cos == nullptr) == (rot_sin == nullptr));
  if (!wdno_tattn_fused_takes(C, n_tok, heads) || n_tok != TF_NT || hw > 0x7fffffff / (TF_C * TF_NT)) return WDNO_EUNSUPPORTED;      // (48 frames: forward only)
  if (ws_bytes < wdno_tattn_fused_bwd_ws_bytes()) return WDNO_EWORKSPACE;
  TFusedBwdP p;
  p.x = x; p.dy = dy; p.gamma = gamma; p.eps = eps;
  p.wq_hi = (const _Float16*)wq_hi; p.wq_lo = (const _Float16*)wq_lo; p.wq_scale = wq_scale;
  p.wo_hi = (const _Float16*)wo_hi; p.wo_lo = (const _Float16*)wo_lo; p.wo_scale = wo_scale;
  p.rcos = rot_cos; p.rsin = rot_sin; p.bias = bias;
  p.rec_dy = rec_dy; p.rec_v = rec_v;
  p.dx = dx; p.amax_rec = amax_rec; p.part = (float*)ws;
  p.HW = (int)hw; p.scale = scale; p.nseq = n_batch * hw;
  static int abl = -1;
  typedef void (*kern_t)(TFusedBwdP);
  static kern_t kern = nullptr;
  if (abl < 0) {
    const char* e = getenv("WDNO_TB_ABLATE");
    abl = e ? atoi(e) : 0;
#ifdef WDNO_TB_ABLATIONS        /* hipcc -DWDNO_TB_ABLATIONS: tools/tattn_ablate.sh */
    kern = abl == 1 ? tattn_fused_bwd_kernel<1> : abl == 2 ? tattn_fused_bwd_kernel<2> : abl == 3 ? tattn_fused_bwd_kernel<3> : abl == 4 ? tattn_fused_bwd_kernel<4> :
           abl == 5 ? tattn_fused_bwd_kernel<5> : tattn_fused_bwd_kernel<0>;
#else
    kern = tattn_fused_bwd_kernel<0>;
#endif
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES) != hipSuccess) return WDNO_ELAUNCH;
  }
  int64_t grid = tb_num_cus();
  if (grid > p.nseq) grid = p.nseq;
  kern<<<(int)grid, 256, TB_LDS_BYTES, as_stream(s)>>>(p);
  int rc = wdno_check_launch();
  if (rc) return rc;
  tattn_fused_reduce_kernel<<<(TB_E + 31) / 32, 256, 0, as_stream(s)>>>((const float*)ws, (int)grid, grads, TB_E);
  return wdno_check_launch();
}
