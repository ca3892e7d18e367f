// linattn_fused_bwd.hip -- backward of the fused SpatialLinearAttention block of linattn_fused.hip (conv3d.py:165-174 LayerNorm, :232-258):
//
//     y = x + b_out + W_out . out ,   out[n][e] = sum_d ctx[d][e] qs[n][d] ,   ctx[d][e] = sum_n ks[n][d] v[n][e] ,
//     qs = scale softmax_d(q) ,  ks = softmax_n(k) ,  (q | k | v) = W_qkv . LayerNorm(x)          per frame (unit) and head
//     given dy:  dx, dgamma, dW_qkv, dW_out, db_out
//
// Kept from the forward: x, ctx [unit][head][32][32] and (max_n k, 1 / Z) per feature -- a few KB per frame. Everything per token is recomputed.
//   dout = W_out^T dy                               dctx[d][e] = sum_n qs[n][d] dout[n][e]            <- the ONE reduction over the tokens
//   dqs[n][d] = sum_e ctx[d][e] dout[n][e]          dq = qs (dqs - sum_d qs dqs / scale)
//   dks[n][d] = sum_e dctx[d][e] v[n][e]            dk = ks (dks - T[d]),   T[d] = sum_n ks dks = sum_e dctx[d][e] ctx[d][e]   (no second reduction)
//   dv[n][e]  = sum_d ks[n][d] dctx[d][e]
//   lattn_fused_dctx_kernel    per (frame, token chunk): q in the usual layout (softmax inside a lane pair), moved to [token][feature] through one
//                              LDS tile; dout with the operands swapped ([token][feature] at once); dctx += on the exact-fp32 MFMA, operands in place
//   lattn_fused_dctx_merge     chunks summed in chunk order, T[d]
//   lattn_fused_bwd_kernel     per token tile: projections recomputed, the four feature-contracting products on the exact-fp32 MFMA with the token
//                              operand in place and the 32 x 32 table (ctx / dctx, either orientation) as the lane = feature operand, then -- as
//                              attn_fused_bwd.hip -- dq / dk / dv / out as fp16 planes for the weight-gradient products (transpose reads, running
//                              power-of-two scales, MFMA accumulators over all tiles of the block, in-place MFMA rescale), dxn = W^T dqkv^T from the
//                              swizzled LDS image of W_qkv, LayerNorm backward, residual gradient. One partial [dW_qkv | dW_out | dgamma | db_out] per
//                              block, summed in block order by lattn_fused_reduce_kernel: no atomics, bit-reproducible.
#include "attn_fused.h"

#define LB_PS 36                      /* halves per row of a per-head plane tile */
#define LB_N_WQ (3 * TF_HD * TF_C)
#define LB_N_WO (TF_C * TF_HD)
#define LB_OFF_WO LB_N_WQ
#define LB_OFF_DG (LB_N_WQ + LB_N_WO)
#define LB_OFF_DB (LB_OFF_DG + TF_C)
#define LB_E (LB_OFF_DB + TF_C)       /* 32896 floats per partial */

// LDS map of lattn_fused_bwd_kernel (bytes)
#define LB_IMG (32 * TF_AST * 2)                     /* 4608: one plane of xn / dy, 32 token rows */
#define LB_L_WH 0
#define LB_L_WL 49152
#define LB_L_XH 98304
#define LB_L_XL (LB_L_XH + LB_IMG)
#define LB_L_GH (LB_L_XL + LB_IMG)
#define LB_L_GL (LB_L_GH + LB_IMG)
#define LB_L_Y (LB_L_GL + LB_IMG)                    /* per head: [32][TF_YST] fp32 partial dxn; its first 4608 bytes double as the plane tile */
#define LB_HEAD_LDS (32 * TF_YST * 4)                /* 8704 */
#define LB_L_TAB (LB_L_Y + TF_HEADS * LB_HEAD_LDS)   /* per head: max_n k, 1 / Z, T: 3 x 32 floats */
#define LB_LDS_BYTES (LB_L_TAB + TF_HEADS * 3 * 32 * 4)

#define LB_FENCE() asm volatile("" ::: "memory")

struct LBwdP {
  const float* x; const float* dy; const float* gamma; float eps;
  const _Float16* wq_hi; const _Float16* wq_lo; const float* wq_scale;      // packed forward operand of to_qkv: [384][64]
  const _Float16* wo_hi; const _Float16* wo_lo; const float* wo_scale;      // packed data-gradient operand of to_out: W_out^T [128][64]
  const float* ctx; const float* kstat; const float* rec_dy;
  float* part1;              // dctx partials [units][chunks][heads][1024]
  float* dctx; float* tvec;  // [units][heads][1024], [units][heads][32]
  float* dx; float* amax_rec; float* part2;
  int n_tok, chunks, tiles_per_chunk; int64_t units; float scale;
  int chunks2, tiles_per_chunk2;          // pass 2 (persistent grid): its own cut of the frames, chosen for equal tiles per block
};

typedef short lb_short4 __attribute__((ext_vector_type(4)));
typedef short lb_short8 __attribute__((ext_vector_type(8)));
typedef lb_short4 __attribute__((address_space(3))) * lb_lds_s4;

__device__ __forceinline__ half8 lb_tr2(const _Float16* p0, const _Float16* p1) {
  const lb_short4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lb_lds_s4)(p0));
  const lb_short4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lb_lds_s4)(p1));
  const lb_short8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8, c);
}
// operand fragment from a [token][channel] image of 32 rows: lane (li, hh) receives channel ch0 + li of tokens tok0 + 8 hh + (0..7)
__device__ __forceinline__ half8 lb_trf(const _Float16* tile, int stride, int tok0, int ch0, int lane) {
  const int g = lane >> 4, xl = lane & 15;
  const _Float16* p0 = tile + (tok0 + 8 * (g >> 1) + (xl >> 2)) * stride + ch0 + 16 * (g & 1) + 4 * (xl & 3);
  return lb_tr2(p0, p0 + 4 * stride);
}
__device__ __forceinline__ int lb_woff(int f, int chunk) { return f * TF_C + ((chunk ^ ((f >> 1) & 7)) << 3); }
// W^T fragment for dxn^T[c][tok] = sum_f W[f][c] d[tok][f] (attn_fused_bwd.hip: tb_wtr)
__device__ __forceinline__ half8 lb_wtr(const _Float16* W, int f0, int ct, int lane) {
  const int g = lane >> 4, xl = lane & 15;
  const int ra = f0 + 4 * (g >> 1) + (xl >> 2), rb = ra + 8;
  const int col = 32 * ct + 16 * (g & 1) + 4 * (xl & 3);
  return lb_tr2(W + lb_woff(ra, col >> 3) + (col & 7), W + lb_woff(rb, col >> 3) + (col & 7));
}
__device__ __forceinline__ f32x16 lb_zero() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}
__device__ __forceinline__ f32x16 lb_mfma3(half8 ah, half8 al, half8 bh, half8 bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
}
// D^T[i][tok] = sum over the 32 features of a 32 x 32 table fragment (lane = i) and an accumulator tile in place
__device__ __forceinline__ f32x16 lb_product16(const float (&a)[16], const f32x16& b) {
  f32x16 acc = lb_zero();
#pragma unroll
  for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void lb_split16(const f32x16& v, float s, half8 (&h)[2], half8 (&l)[2], _Float16* __restrict__ Ph, _Float16* __restrict__ Pl,
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
    *reinterpret_cast<half4v*>(Ph + li * LB_PS + 8 * c + 4 * hh) = a;
    *reinterpret_cast<half4v*>(Pl + li * LB_PS + 8 * c + 4 * hh) = b;
  }
}
__device__ __forceinline__ void lb_ln_row(float4 xv, float4 g, float eps, float ps, _Float16* __restrict__ Ah, _Float16* __restrict__ Al, int row, int c4,
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
__device__ __forceinline__ void lb_plane_row(float4 v, float s, _Float16* __restrict__ Ah, _Float16* __restrict__ Al, int row, int c4) {
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
__device__ __forceinline__ float lb_scale(float bound) { return fminf(fmaxf(scale_from_amax(bound), 0x1p-100f), 0x1p100f); }
// w *= r in place on the matrix pipe (attn_fused_bwd.hip: tb_rescale)
__device__ __forceinline__ void lb_rescale(f32x16& w, float r, int li, int hh) {
  const float r1 = r - 1.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) w = __builtin_amdgcn_mfma_f32_32x32x2f32(li == tf_key(e, hh) ? r1 : 0.f, w[e], w, 0, 0, 0);
}
__device__ __forceinline__ void lb_fit(float& sc, float amax, f32x16& w0, f32x16& w1, int li, int hh) {
  const float need = lb_scale(amax);
  if (need < sc) {
    const float r = need / sc;
    lb_rescale(w0, r, li, hh);
    lb_rescale(w1, r, li, hh);
    sc = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(need)));
  }
}
__device__ __forceinline__ float lb_absmax16(const f32x16& v) {
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(v[e]));
  return m;
}
// qs = scale softmax over the head's 32 features of the token (16 in this lane, 16 in lane ^ 32), in place
__device__ __forceinline__ void lb_softmax_d(f32x16& aq, float scale) {
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) mx = fmaxf(mx, aq[e]);
  float m0, m1;
  tf_halves(mx, m0, m1);
  mx = fmaxf(m0, m1);
  float l = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { aq[e] = expf(aq[e] - mx); l += aq[e]; }
  float l0, l1;
  tf_halves(l, l0, l1);
  const float il = scale / (l0 + l1);
#pragma unroll
  for (int e = 0; e < 16; ++e) aq[e] *= il;
}

// ------------------------------------------------------------------------------------------------ pass 1: dctx per chunk
__global__ __launch_bounds__(256, 2) void lattn_fused_dctx_kernel(LBwdP p) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) _Float16 Gh[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) _Float16 Gl[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) float Tq[TF_HEADS][32 * TF_VST];
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  const int unit = (int)(blockIdx.x / (unsigned)p.chunks), chunk = (int)(blockIdx.x - (unsigned)unit * (unsigned)p.chunks);
  half8 wqh[4], wql[4], woth[4], wotl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int oq = (h * 32 + li) * TF_C + 16 * s + 8 * hh;
    wqh[s] = *reinterpret_cast<const half8*>(p.wq_hi + oq); wql[s] = *reinterpret_cast<const half8*>(p.wq_lo + oq);
    woth[s] = *reinterpret_cast<const half8*>(p.wo_hi + oq); wotl[s] = *reinterpret_cast<const half8*>(p.wo_lo + oq);      // W_out^T rows 32 h + e
  }
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float sc_g = lb_scale(amax_record_read(p.rec_dy));
  const float* xu = p.x + (int64_t)unit * p.n_tok * TF_C;
  const float* gu = p.dy + (int64_t)unit * p.n_tok * TF_C;
  const int tile0 = chunk * p.tiles_per_chunk;
  const int ntiles = (p.n_tok + 31) >> 5;
  const int tile1 = min(ntiles, tile0 + p.tiles_per_chunk);
  f32x16 acc = lb_zero();                               // dctx[d][e] of this chunk: lane (e, hh), register r <-> d = tf_key(r, hh)
  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0, ng0 = nx0, ng1 = nx0;
  auto fetch = [&](int tile) {
    const int r0 = tile * 32 + lrow, r1 = r0 + 16;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    nx0 = r0 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r0 * TF_C + 4 * lc4) : z;
    ng0 = r0 < p.n_tok ? *reinterpret_cast<const float4*>(gu + (int64_t)r0 * TF_C + 4 * lc4) : z;
    nx1 = r1 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r1 * TF_C + 4 * lc4) : z;
    ng1 = r1 < p.n_tok ? *reinterpret_cast<const float4*>(gu + (int64_t)r1 * TF_C + 4 * lc4) : z;
  };
  if (tile0 < tile1) fetch(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    float mean, rstd;
    lb_ln_row(nx0, g4, p.eps, ps, Ah, Al, lrow, lc4, mean, rstd);
    lb_ln_row(nx1, g4, p.eps, ps, Ah, Al, 16 + lrow, lc4, mean, rstd);
    lb_plane_row(ng0, sc_g, Gh, Gl, lrow, lc4);
    lb_plane_row(ng1, sc_g, Gh, Gl, 16 + lrow, lc4);
    __syncthreads();
    if (tile + 1 < tile1) fetch(tile + 1);
    f32x16 aq = lb_zero(), dT = lb_zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 xh = *reinterpret_cast<const half8*>(Ah + li * TF_AST + 16 * s + 8 * hh);
      const half8 xl = *reinterpret_cast<const half8*>(Al + li * TF_AST + 16 * s + 8 * hh);
      const half8 gh = *reinterpret_cast<const half8*>(Gh + li * TF_AST + 16 * s + 8 * hh);
      const half8 gl = *reinterpret_cast<const half8*>(Gl + li * TF_AST + 16 * s + 8 * hh);
      aq = lb_mfma3(wqh[s], wql[s], xh, xl, aq);                 // q^T[d][tok]: a lane owns a token
      dT = lb_mfma3(gh, gl, woth[s], wotl[s], dT);               // dout[tok][e] (swapped operands): a lane owns the feature e, 16 tokens
    }
    __syncthreads();                                             // the planes may be rewritten
#pragma unroll
    for (int e = 0; e < 16; ++e) aq[e] *= inv_qkv;
    lb_softmax_d(aq, p.scale);
    // qs -> [token][feature] through the head's tile: a lane then owns the feature d and the tokens tf_key(m, hh)
    float* tq = Tq[h];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<float4*>(tq + li * TF_VST + 8 * c + 4 * hh) = make_float4(aq[4 * c], aq[4 * c + 1], aq[4 * c + 2], aq[4 * c + 3]);
    __builtin_amdgcn_wave_barrier();
    LB_FENCE();
    float qT[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) qT[m] = tq[tf_key(m, hh) * TF_VST + li];
    LB_FENCE();
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qT[m], dT[m], acc, 0, 0, 0);
  }
  const float inv_do = 1.0f / (sc_g * p.wo_scale[0]);
  float* po = p.part1 + ((int64_t)blockIdx.x * TF_HEADS + h) * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) po[tf_key(r, hh) * 32 + li] = acc[r] * inv_do;
}

__global__ __launch_bounds__(256) void lattn_fused_dctx_merge_kernel(const float* __restrict__ part, const float* __restrict__ ctx, float* __restrict__ dctx,
                                                                       float* __restrict__ tvec, int chunks) {
  __shared__ float red[32][33];
  const int64_t unit = blockIdx.x / TF_HEADS;
  const int h = (int)(blockIdx.x - unit * TF_HEADS);
  const float* p0 = part + ((unit * chunks) * TF_HEADS + h) * (int64_t)1024;
  const int64_t cstride = (int64_t)TF_HEADS * 1024;
  for (int o = threadIdx.x; o < 1024; o += 256) {
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += p0[c * cstride + o];
    dctx[(int64_t)blockIdx.x * 1024 + o] = v;
    red[o >> 5][o & 31] = v * ctx[(int64_t)blockIdx.x * 1024 + o];
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float t = 0.f;
    for (int e = 0; e < 32; ++e) t += red[threadIdx.x][e];
    tvec[(int64_t)blockIdx.x * 32 + threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------------------------ pass 2: everything per token
__global__ __launch_bounds__(256, 1) void lattn_fused_bwd_kernel(LBwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
  _Float16* WH = reinterpret_cast<_Float16*>(lb_smem + LB_L_WH);
  _Float16* WL = reinterpret_cast<_Float16*>(lb_smem + LB_L_WL);
  _Float16* XH = reinterpret_cast<_Float16*>(lb_smem + LB_L_XH);
  _Float16* XL = reinterpret_cast<_Float16*>(lb_smem + LB_L_XL);
  _Float16* GH = reinterpret_cast<_Float16*>(lb_smem + LB_L_GH);
  _Float16* GL = reinterpret_cast<_Float16*>(lb_smem + LB_L_GL);
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  float* Yp = reinterpret_cast<float*>(lb_smem + LB_L_Y + h * LB_HEAD_LDS);
  _Float16* PH = reinterpret_cast<_Float16*>(Yp);
  _Float16* PL = PH + 32 * LB_PS;
  float* KM = reinterpret_cast<float*>(lb_smem + LB_L_TAB) + h * 96;
  float* KZ = KM + 32;
  float* KT = KM + 64;

  for (int q = tid; q < 3 * TF_HD * 8; q += 256) {                // W_qkv planes -> LDS (swizzled: attn_fused_bwd.hip)
    const int f = q >> 3, ch = q & 7;
    const int dst = lb_woff(f, ch);
    *reinterpret_cast<uint4*>(WH + dst) = *reinterpret_cast<const uint4*>(p.wq_hi + f * TF_C + ch * 8);
    *reinterpret_cast<uint4*>(WL + dst) = *reinterpret_cast<const uint4*>(p.wq_lo + f * TF_C + ch * 8);
  }
  const _Float16* wot_h = p.wo_hi + (32 * h + li) * TF_C + 8 * hh;
  const _Float16* wot_l = p.wo_lo + (32 * h + li) * TF_C + 8 * hh;
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));
  const float wq_s = p.wq_scale[0], wo_s = p.wo_scale[0];
  const float inv_qkv = 1.0f / (ps * wq_s);
  const float sc_g = lb_scale(amax_record_read(p.rec_dy));
  const float inv_do = 1.0f / (sc_g * wo_s);
  const float inv_scale = 1.0f / p.scale;
  float sc_o = 0x1p100f, sc_q = 0x1p100f, sc_k = 0x1p100f, sc_v = 0x1p100f;      // running plane scales (attn_fused_bwd.hip)

  f32x16 dwq[3][2], dwo[2];
#pragma unroll
  for (int ti = 0; ti < 3; ++ti) { dwq[ti][0] = lb_zero(); dwq[ti][1] = lb_zero(); }
  dwo[0] = lb_zero(); dwo[1] = lb_zero();
  float4 dgacc = make_float4(0.f, 0.f, 0.f, 0.f), dbacc = dgacc;
  float am = 0.f;
  __syncthreads();

  const int ntiles = (p.n_tok + 31) >> 5;
  const int64_t nitems = p.units * p.chunks2;
  for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
    // chunk-major item order: the items a block takes (item, item + grid, ...) come from different chunk indices, so the short last chunk of a
    // frame is spread over the blocks instead of landing on the same ones
    const int chunk = (int)(item / p.units);
    const int64_t unit = item - (int64_t)chunk * p.units;
    const int tile0 = chunk * p.tiles_per_chunk2;
    const int tile1 = min(ntiles, tile0 + p.tiles_per_chunk2);
    if (tile0 >= tile1) continue;                                  // (block-uniform)
    // ---- tables of this (frame, head): ctx and dctx in both orientations (row operands of the feature-contracting products), k statistics, T
    // (the column orientation stays in registers for the item; the row orientation -- this lane's row of the table, 64 bytes per lane half --
    // is fetched per tile, L1 / L2: 64 more registers held for the whole item spilled)
    float ctxf[16], dcf[16];
    const float* cu = p.ctx + (unit * TF_HEADS + h) * 1024;
    const float* du = p.dctx + (unit * TF_HEADS + h) * 1024;
    auto table_row = [&](const float* tab, float (&t)[16]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(tab + li * 32 + 8 * c + 4 * hh);
        t[4 * c] = a.x; t[4 * c + 1] = a.y; t[4 * c + 2] = a.z; t[4 * c + 3] = a.w;
      }
    };
    {
      float amc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ctxf[r] = cu[tf_key(r, hh) * 32 + li];
        dcf[r] = du[tf_key(r, hh) * 32 + li];
        amc = fmaxf(amc, fabsf(ctxf[r]));
      }
      if (hh == 0) {
        const float* ks = p.kstat + (unit * TF_HEADS + h) * 64;
        KM[li] = ks[li]; KZ[li] = ks[32 + li];
        KT[li] = p.tvec[(unit * TF_HEADS + h) * 32 + li];
      }
      lb_fit(sc_o, p.scale * tf_wave_max(amc), dwo[0], dwo[1], li, hh);        // |out[e]| <= max|ctx| sum_d qs[d] = scale max|ctx|
    }
    const float* xu = p.x + unit * p.n_tok * TF_C;
    const float* gu = p.dy + unit * p.n_tok * TF_C;
    float* du_out = p.dx + unit * p.n_tok * TF_C;
    float4 nx0, nx1, ng0, ng1;
    auto fetch = [&](int tile, float4& x0, float4& x1, float4& g0, float4& g1) {
      const int r0 = tile * 32 + lrow, r1 = r0 + 16;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      x0 = r0 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r0 * TF_C + 4 * lc4) : z;
      g0 = r0 < p.n_tok ? *reinterpret_cast<const float4*>(gu + (int64_t)r0 * TF_C + 4 * lc4) : z;
      x1 = r1 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r1 * TF_C + 4 * lc4) : z;
      g1 = r1 < p.n_tok ? *reinterpret_cast<const float4*>(gu + (int64_t)r1 * TF_C + 4 * lc4) : z;
    };
    fetch(tile0, nx0, nx1, ng0, ng1);
    for (int tile = tile0; tile < tile1; ++tile) {
      // ---- rows -> LayerNorm -> planes of xn; planes of dy; db_out
      float mean0, mean1, rs0, rs1;
      lb_ln_row(nx0, g4, p.eps, ps, XH, XL, lrow, lc4, mean0, rs0);
      lb_ln_row(nx1, g4, p.eps, ps, XH, XL, 16 + lrow, lc4, mean1, rs1);
      lb_plane_row(ng0, sc_g, GH, GL, lrow, lc4);
      lb_plane_row(ng1, sc_g, GH, GL, 16 + lrow, lc4);
      dbacc.x += ng0.x + ng1.x; dbacc.y += ng0.y + ng1.y; dbacc.z += ng0.z + ng1.z; dbacc.w += ng0.w + ng1.w;
      half8 woth[4], wotl[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        woth[s] = *reinterpret_cast<const half8*>(wot_h + 16 * s);
        wotl[s] = *reinterpret_cast<const half8*>(wot_l + 16 * s);
      }
      __syncthreads();                                                          // B1: planes (and, at the first tile of an item, the tables)
      // ---- (q | k | v)^T of this head, dout^T
      f32x16 aq = lb_zero(), ak = lb_zero(), av = lb_zero(), dOT = lb_zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 bh = *reinterpret_cast<const half8*>(XH + li * TF_AST + 16 * s + 8 * hh);
        const half8 bl = *reinterpret_cast<const half8*>(XL + li * TF_AST + 16 * s + 8 * hh);
        const int o0 = lb_woff(h * 32 + li, 2 * s + hh);
        aq = lb_mfma3(*reinterpret_cast<const half8*>(WH + o0), *reinterpret_cast<const half8*>(WL + o0), bh, bl, aq);
        ak = lb_mfma3(*reinterpret_cast<const half8*>(WH + o0 + TF_HD * TF_C), *reinterpret_cast<const half8*>(WL + o0 + TF_HD * TF_C), bh, bl, ak);
        av = lb_mfma3(*reinterpret_cast<const half8*>(WH + o0 + 2 * TF_HD * TF_C), *reinterpret_cast<const half8*>(WL + o0 + 2 * TF_HD * TF_C), bh, bl, av);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 bh = *reinterpret_cast<const half8*>(GH + li * TF_AST + 16 * s + 8 * hh);
        const half8 bl = *reinterpret_cast<const half8*>(GL + li * TF_AST + 16 * s + 8 * hh);
        dOT = lb_mfma3(woth[s], wotl[s], bh, bl, dOT);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) { aq[e] *= inv_qkv; ak[e] *= inv_qkv; av[e] *= inv_qkv; dOT[e] *= inv_do; }
      lb_softmax_d(aq, p.scale);                                                // aq = qs
#pragma unroll
      for (int c = 0; c < 4; ++c) {                                             // ak = ks = exp(k - max_n k) / Z
        const float4 m4 = *reinterpret_cast<const float4*>(KM + 8 * c + 4 * hh), z4 = *reinterpret_cast<const float4*>(KZ + 8 * c + 4 * hh);
        ak[4 * c] = expf(ak[4 * c] - m4.x) * z4.x; ak[4 * c + 1] = expf(ak[4 * c + 1] - m4.y) * z4.y;
        ak[4 * c + 2] = expf(ak[4 * c + 2] - m4.z) * z4.z; ak[4 * c + 3] = expf(ak[4 * c + 3] - m4.w) * z4.w;
      }
      half8 dh[2], dl[2];
      f32x16 dxs0 = lb_zero(), dxs1 = lb_zero();
      auto grad_products = [&](f32x16& w0, f32x16& w1, float sc, int fbase) {
        const half8 a0h = lb_trf(PH, LB_PS, 0, 0, lane), a0l = lb_trf(PL, LB_PS, 0, 0, lane);
        const half8 a1h = lb_trf(PH, LB_PS, 16, 0, lane), a1l = lb_trf(PL, LB_PS, 16, 0, lane);
        w0 = lb_mfma3(a0h, a0l, lb_trf(XH, TF_AST, 0, 0, lane), lb_trf(XL, TF_AST, 0, 0, lane), w0);
        w0 = lb_mfma3(a1h, a1l, lb_trf(XH, TF_AST, 16, 0, lane), lb_trf(XL, TF_AST, 16, 0, lane), w0);
        w1 = lb_mfma3(a0h, a0l, lb_trf(XH, TF_AST, 0, 32, lane), lb_trf(XL, TF_AST, 0, 32, lane), w1);
        w1 = lb_mfma3(a1h, a1l, lb_trf(XH, TF_AST, 16, 32, lane), lb_trf(XL, TF_AST, 16, 32, lane), w1);
        f32x16 d0 = lb_zero(), d1 = lb_zero();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          d0 = lb_mfma3(lb_wtr(WH, fbase + 16 * s, 0, lane), lb_wtr(WL, fbase + 16 * s, 0, lane), dh[s], dl[s], d0);
          d1 = lb_mfma3(lb_wtr(WH, fbase + 16 * s, 1, lane), lb_wtr(WL, fbase + 16 * s, 1, lane), dh[s], dl[s], d1);
        }
        const float inv = 1.0f / (sc * wq_s);
#pragma unroll
        for (int e = 0; e < 16; ++e) { dxs0[e] = fmaf(d0[e], inv, dxs0[e]); dxs1[e] = fmaf(d1[e], inv, dxs1[e]); }
      };
      // ---- out^T = ctx^T qs^T -> dW_out[c][32 h + e] += sum_tok dy[tok][c] out[tok][e]
      {
        const f32x16 oT = lb_product16(ctxf, aq);
        lb_split16(oT, sc_o, dh, dl, PH, PL, li, hh);
        LB_FENCE();
        const half8 b0h = lb_trf(PH, LB_PS, 0, 0, lane), b0l = lb_trf(PL, LB_PS, 0, 0, lane);
        const half8 b1h = lb_trf(PH, LB_PS, 16, 0, lane), b1l = lb_trf(PL, LB_PS, 16, 0, lane);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          dwo[ct] = lb_mfma3(lb_trf(GH, TF_AST, 0, 32 * ct, lane), lb_trf(GL, TF_AST, 0, 32 * ct, lane), b0h, b0l, dwo[ct]);
          dwo[ct] = lb_mfma3(lb_trf(GH, TF_AST, 16, 32 * ct, lane), lb_trf(GL, TF_AST, 16, 32 * ct, lane), b1h, b1l, dwo[ct]);
        }
      }
      LB_FENCE();
      // ---- dq = qs (dqs - sum_d qs dqs / scale),  dqs^T = ctx dout^T
      {
        float ctxT[16];
        table_row(cu, ctxT);
        f32x16 dq = lb_product16(ctxT, dOT);
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sm = fmaf(aq[e], dq[e], sm);
        float s0, s1;
        tf_halves(sm, s0, s1);
        sm = (s0 + s1) * inv_scale;
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[e] = aq[e] * (dq[e] - sm);
        lb_fit(sc_q, tf_wave_max(lb_absmax16(dq)), dwq[0][0], dwq[0][1], li, hh);
        lb_split16(dq, sc_q, dh, dl, PH, PL, li, hh);
        LB_FENCE();
        grad_products(dwq[0][0], dwq[0][1], sc_q, h * 32);
      }
      LB_FENCE();
      // ---- dk = ks (dks - T),  dks^T = dctx v^T
      {
        float dcT[16];
        table_row(du, dcT);
        f32x16 dk = lb_product16(dcT, av);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 t4 = *reinterpret_cast<const float4*>(KT + 8 * c + 4 * hh);
          dk[4 * c] = ak[4 * c] * (dk[4 * c] - t4.x); dk[4 * c + 1] = ak[4 * c + 1] * (dk[4 * c + 1] - t4.y);
          dk[4 * c + 2] = ak[4 * c + 2] * (dk[4 * c + 2] - t4.z); dk[4 * c + 3] = ak[4 * c + 3] * (dk[4 * c + 3] - t4.w);
        }
        lb_fit(sc_k, tf_wave_max(lb_absmax16(dk)), dwq[1][0], dwq[1][1], li, hh);
        lb_split16(dk, sc_k, dh, dl, PH, PL, li, hh);
        LB_FENCE();
        grad_products(dwq[1][0], dwq[1][1], sc_k, TF_HD + h * 32);
      }
      LB_FENCE();
      // the rows of this tile again (LayerNorm backward, residual gradient) and those of the next one: in flight under the last products
      float4 cx0, cx1, cg0, cg1;
      fetch(tile, cx0, cx1, cg0, cg1);
      if (tile + 1 < tile1) fetch(tile + 1, nx0, nx1, ng0, ng1);
      // ---- dv^T = dctx^T ks^T
      {
        const f32x16 dv = lb_product16(dcf, ak);
        lb_fit(sc_v, tf_wave_max(lb_absmax16(dv)), dwq[2][0], dwq[2][1], li, hh);
        lb_split16(dv, sc_v, dh, dl, PH, PL, li, hh);
        LB_FENCE();
        grad_products(dwq[2][0], dwq[2][1], sc_v, 2 * TF_HD + h * 32);
      }
      LB_FENCE();
      // ---- the head's part of dxn as [token][channel] (over the plane tile: every lane has taken its fragments)
      {
        float* yp = Yp + li * TF_YST + 4 * hh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          *reinterpret_cast<float4*>(yp + 8 * c) = make_float4(dxs0[4 * c], dxs0[4 * c + 1], dxs0[4 * c + 2], dxs0[4 * c + 3]);
          *reinterpret_cast<float4*>(yp + 32 + 8 * c) = make_float4(dxs1[4 * c], dxs1[4 * c + 1], dxs1[4 * c + 2], dxs1[4 * c + 3]);
        }
      }
      __syncthreads();                                                          // B2
      const float* Y0 = reinterpret_cast<const float*>(lb_smem + LB_L_Y);
      auto finish = [&](int row, const float4& xr, float mean, float rstd, const float4& gy) {
        const int tok = tile * 32 + row;
        if (tok >= p.n_tok) return;
        const int o = row * TF_YST + 4 * lc4;
        const float4 a = *reinterpret_cast<const float4*>(Y0 + o), b2 = *reinterpret_cast<const float4*>(Y0 + LB_HEAD_LDS / 4 + o);
        const float4 c = *reinterpret_cast<const float4*>(Y0 + 2 * (LB_HEAD_LDS / 4) + o), d = *reinterpret_cast<const float4*>(Y0 + 3 * (LB_HEAD_LDS / 4) + o);
        const float4 xh = make_float4((xr.x - mean) * rstd, (xr.y - mean) * rstd, (xr.z - mean) * rstd, (xr.w - mean) * rstd);
        float4 dn;
        dn.x = (a.x + b2.x) + (c.x + d.x); dn.y = (a.y + b2.y) + (c.y + d.y);
        dn.z = (a.z + b2.z) + (c.z + d.z); dn.w = (a.w + b2.w) + (c.w + d.w);
        dgacc.x += dn.x * xh.x; dgacc.y += dn.y * xh.y; dgacc.z += dn.z * xh.z; dgacc.w += dn.w * xh.w;
        dn.x *= g4.x; dn.y *= g4.y; dn.z *= g4.z; dn.w *= g4.w;
        // (16 lanes of a row take part together: rows are valid or invalid as a whole)
        const float m1 = tf_row16_sum((dn.x + dn.y) + (dn.z + dn.w)) * (1.0f / TF_C);
        const float m2 = tf_row16_sum((dn.x * xh.x + dn.y * xh.y) + (dn.z * xh.z + dn.w * xh.w)) * (1.0f / TF_C);
        float4 r;
        r.x = rstd * (dn.x - m1 - xh.x * m2) + gy.x; r.y = rstd * (dn.y - m1 - xh.y * m2) + gy.y;
        r.z = rstd * (dn.z - m1 - xh.z * m2) + gy.z; r.w = rstd * (dn.w - m1 - xh.w * m2) + gy.w;
        *reinterpret_cast<float4*>(du_out + (int64_t)tok * TF_C + 4 * lc4) = r;
        am = amax4(am, r);
      };
      finish(lrow, cx0, mean0, rs0, cg0);
      finish(16 + lrow, cx1, mean1, rs1, cg1);
      __syncthreads();                                                          // B3: the partial tiles (= the plane tiles of the next tile) are free
    }
  }
  // ---- this block's partial sums
  float* part = p.part2 + (size_t)blockIdx.x * LB_E;
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
      for (int e = 0; e < 16; ++e) part[LB_OFF_WO + (32 * ct + tf_key(e, hh)) * TF_HD + h * 32 + li] = dwo[ct][e] * inv_wo;
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(lb_smem + LB_L_Y);         // dgamma, db_out over the 16 row groups of the block, in row-group order
  *reinterpret_cast<float4*>(red + lrow * TF_C + 4 * lc4) = dgacc;
  *reinterpret_cast<float4*>(red + 1024 + lrow * TF_C + 4 * lc4) = dbacc;
  __syncthreads();
  if (tid < 2 * TF_C) {
    const float* r0 = red + (tid >= TF_C ? 1024 : 0) + (tid & (TF_C - 1));
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += r0[r * TF_C];
    part[LB_OFF_DG + tid] = t;
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * TF_HEADS + h);
}

// out[e] = sum over the blocks' partials in block order (attn_fused_bwd.hip: tattn_fused_reduce_kernel)
__global__ __launch_bounds__(256) void lattn_fused_reduce_kernel(const float* __restrict__ part, int nb, float* __restrict__ out, int E) {
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

static int lb_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}
static int lb_chunks(int64_t units, int n_tok) {          // as linattn_fused.hip: lf_chunks
  const int ntiles = (n_tok + 31) / 32;
  int64_t c = (3 * (int64_t)lb_num_cus() + units - 1) / units;
  if (c > ntiles / 2) c = ntiles / 2;
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  return (int)c;
}

// pass 2 runs one block per CU over (frame, chunk) items: the cut that minimises the tiles of the busiest block -- 192 frames of 50 tiles on
// 256 CUs: 3 chunks = 576 items = 2.25 per block (a quarter of the blocks take three: 51 tiles against 37.5 on average), 4 chunks = exactly three
// items of 11 .. 13 tiles per block (39)
static int lb_chunks2(int64_t units, int n_tok, int64_t grid) {
  const int ntiles = (n_tok + 31) / 32;
  int best = 1;
  int64_t best_cost = -1;
  for (int c = 1; c <= 16 && c <= ntiles; ++c) {
    const int64_t cost = ((units * c + grid - 1) / grid) * ((ntiles + c - 1) / c);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}
extern "C" int wdno_lattn_fused_bwd_grads(void) { return LB_E; }
extern "C" size_t wdno_lattn_fused_bwd_ws_bytes(int64_t units, int n_tok) {
  return ((size_t)units * lb_chunks(units, n_tok) * TF_HEADS * 1024 + (size_t)units * TF_HEADS * (1024 + 32) + (size_t)lb_num_cus() * LB_E) * sizeof(float);
}
extern "C" int wdno_lattn_fused_bwd(const float* x, const float* dy, const float* gamma, float eps, const void* wq_hi, const void* wq_lo,
                                    const float* wq_scale, const void* wot_hi, const void* wot_lo, const float* wot_scale, const float* ctx,
                                    const float* kstat, const float* rec_dy, float* dx, float* amax_rec, float* grads, void* ws, size_t ws_bytes,
                                    int64_t units, int n_tok, int C, int heads, float scale, wdno_stream_t s) {
  WDNO_REQUIRE(x && dy && gamma && wq_hi && wq_lo && wq_scale && wot_hi && wot_lo && wot_scale && ctx && kstat && rec_dy && dx && grads && ws);
  WDNO_REQUIRE(units > 0 && n_tok > 0);
  if (C != TF_C || heads != TF_HEADS || n_tok < 32 || units * (int64_t)n_tok * TF_C > 0x7fffffff0ll) return WDNO_EUNSUPPORTED;
  if (ws_bytes < wdno_lattn_fused_bwd_ws_bytes(units, n_tok)) return WDNO_EWORKSPACE;
  LBwdP p;
  p.x = x; p.dy = dy; p.gamma = gamma; p.eps = eps;
  p.wq_hi = (const _Float16*)wq_hi; p.wq_lo = (const _Float16*)wq_lo; p.wq_scale = wq_scale;
  p.wo_hi = (const _Float16*)wot_hi; p.wo_lo = (const _Float16*)wot_lo; p.wo_scale = wot_scale;
  p.ctx = ctx; p.kstat = kstat; p.rec_dy = rec_dy;
  p.n_tok = n_tok; p.chunks = lb_chunks(units, n_tok); p.units = units; p.scale = scale;
  const int ntiles = (n_tok + 31) / 32;
  p.tiles_per_chunk = (ntiles + p.chunks - 1) / p.chunks;
  float* w = (float*)ws;
  p.part1 = w; w += (size_t)units * p.chunks * TF_HEADS * 1024;
  p.dctx = w; w += (size_t)units * TF_HEADS * 1024;
  p.tvec = w; w += (size_t)units * TF_HEADS * 32;
  p.part2 = w;
  p.dx = dx; p.amax_rec = amax_rec;
  const int64_t nitems = units * p.chunks;
  if (nitems > 0x7fffffff) return WDNO_EUNSUPPORTED;
  p.chunks2 = lb_chunks2(units, n_tok, lb_num_cus());
  p.tiles_per_chunk2 = (ntiles + p.chunks2 - 1) / p.chunks2;
  const int64_t nitems2 = units * p.chunks2;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)lattn_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS_BYTES) != hipSuccess) return WDNO_ELAUNCH;
    attr_done = true;
  }
  hipStream_t st = as_stream(s);
  lattn_fused_dctx_kernel<<<(unsigned)nitems, 256, 0, st>>>(p);
  lattn_fused_dctx_merge_kernel<<<(unsigned)(units * TF_HEADS), 256, 0, st>>>(p.part1, ctx, p.dctx, p.tvec, p.chunks);
  int64_t grid = lb_num_cus();
  if (grid > nitems2) grid = nitems2;
  lattn_fused_bwd_kernel<<<(unsigned)grid, 256, LB_LDS_BYTES, st>>>(p);
  lattn_fused_reduce_kernel<<<(LB_E + 31) / 32, 256, 0, st>>>(p.part2, (int)grid, grads, LB_E);
  return wdno_check_launch();
}
