// linattn_fused_wide.hip -- the SpatialLinearAttention block (linattn_fused.hip: Residual(PreNorm(dim, SpatialLinearAttention(dim, heads=4))),
// conv3d.py:165-174, 232-258) of the 128- and 256-channel levels of the smoke U-Net, FORWARD, for passes that need no gradient (sampling):
// the same two passes + merge as at 64 channels,
//
//     lattn_wide_ctx_kernel    k, v of a 32-token tile -> per-chunk softmax statistics and context (online softmax over the tokens)
//     lattn_fused_merge_kernel (linattn_fused.hip) chunks of a frame merged in chunk order
//     lattn_wide_out_kernel    q of a tile -> softmax over the features -> out^T = ctx^T qs^T -> to_out + bias + residual
//
// with what attn_fused_wide.hip changes for the wider levels: the head's weight rows no longer fit the register file, so the fragments
// of to_qkv / to_out are STREAMED per tile from the fragment-ordered operands (pack modes 10 / 11 of csrc/conv_h3.hip: one load instruction
// = 1 KB of contiguous memory), two fragment sets per trip (requested, waited for in full, consumed: attn_fused.h), k-step pair t of lane half
// hh = channels 32 t + 16 hh .. + 15; and to_out
// contracts over all 128 features per wave (wave w = output channels w C/4 ..) from (hi, lo) planes of the four heads' out tiles in LDS
// under ONE scale (scale x max|ctx| over the heads of the frame) instead of summing per-head partial tiles.
#include "linattn_fused.h"

#define LW_OST 136    /* halves per token row of an out plane (128 features + 8) */

// one row of C channels by its 16 lanes: LayerNorm (gain re-read per row: L1 hits) -> (hi, lo) planes
template <int C>
__device__ __forceinline__ void lw_ln_row(const float4 (&xin)[C / 64], const float* __restrict__ gamma, float eps, float ps, _Float16* __restrict__ Ah,
                                          _Float16* __restrict__ Al, int row, int lc4) {
  constexpr int NJ = C / 64, AST = C + 8;
  float4 xv[NJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) { xv[j] = xin[j]; s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w); }
  const float mean = tf_row16_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    xv[j].x -= mean; xv[j].y -= mean; xv[j].z -= mean; xv[j].w -= mean;
    q += (xv[j].x * xv[j].x + xv[j].y * xv[j].y) + (xv[j].z * xv[j].z + xv[j].w * xv[j].w);
  }
  const float rstd = 1.0f / sqrtf(tf_row16_sum(q) * (1.0f / C) + eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float4 g = reinterpret_cast<const float4*>(gamma)[16 * j + lc4];
    const float o[4] = {xv[j].x * rstd * g.x, xv[j].y * rstd * g.y, xv[j].z * rstd * g.z, xv[j].w * rstd * g.w};
    half4v hv, lv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = o[e] * ps;
      hv[e] = (_Float16)t;
      lv[e] = (_Float16)(t - (float)hv[e]);
    }
    *reinterpret_cast<half4v*>(Ah + row * AST + 64 * j + 4 * lc4) = hv;
    *reinterpret_cast<half4v*>(Al + row * AST + 64 * j + 4 * lc4) = lv;
  }
}
template <int C>
__device__ __forceinline__ float lw_plane_scale(const float* __restrict__ gamma, int lc4) {      // |LayerNorm(x)| <= sqrt(C) max|g|
  float gm = 0.f;
#pragma unroll
  for (int j = 0; j < C / 64; ++j) gm = amax4(gm, reinterpret_cast<const float4*>(gamma)[16 * j + lc4]);
  return scale_from_amax(sqrtf((float)C) * group_max<16>(gm));
}

// ------------------------------------------------------------------------------------------------ pass 1: per-chunk context
template <int C>
__global__ __launch_bounds__(256, 2) void lattn_wide_ctx_kernel(LFusedP p) {
  constexpr int AST = C + 8, NP = C / 32, NJ = C / 64;
  __shared__ __attribute__((aligned(16))) _Float16 Ah[32 * AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[32 * AST];
  __shared__ __attribute__((aligned(16))) float Fs[TF_HEADS][32];

  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  const int unit = (int)(blockIdx.x / (unsigned)p.chunks), chunk = (int)(blockIdx.x - (unsigned)unit * (unsigned)p.chunks);
  // k rows (ti = 1) and v rows (ti = 2) of this head, as the COLUMN operand of the swapped product: + ((ti NP + t) 2 + s) 512
  const unsigned wq_off = (unsigned)(h * 3 * NP * 2 * 64 + lane) * 16u;      // bytes
  const __amdgpu_buffer_rsrc_t rqh = tf_rsrc(p.wq_hi, 3 * TF_HD * C * 2), rql = tf_rsrc(p.wq_lo, 3 * TF_HD * C * 2);
  const float ps = lw_plane_scale<C>(p.gamma, lc4);
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float* xu = p.x + (int64_t)unit * p.n_tok * C;
  const int tile0 = chunk * p.tiles_per_chunk;
  const int ntiles = (p.n_tok + 31) >> 5;
  const int tile1 = min(ntiles, tile0 + p.tiles_per_chunk);

  float m_run = -INFINITY, z_run = 0.f;                 // feature d = li: running maximum over the tokens so far, partial sum of this lane half
  f32x16 ctx = lf_zero();                               // ctx_raw[d][e]: lane (e, hh), register r <-> d = tf_key(r, hh)
  float4 nx0[NJ], nx1[NJ];
  auto fetch = [&](int tile) {
    const int r0 = tile * 32 + lrow, r1 = r0 + 16;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      nx0[j] = r0 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r0 * C + 64 * j + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
      nx1[j] = r1 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r1 * C + 64 * j + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (tile0 < tile1) fetch(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    lw_ln_row<C>(nx0, p.gamma, p.eps, ps, Ah, Al, lrow, lc4);
    lw_ln_row<C>(nx1, p.gamma, p.eps, ps, Ah, Al, 16 + lrow, lc4);
    __syncthreads();
    // k[tok][d], v[tok][e] of the tile: rows = tokens (A = the token planes), columns = features (B = the streamed weight rows)
    f32x16 ak = lf_zero(), av = lf_zero();
    {
      half8 w0[2][2][2], w1[2][2][2];                   // [k | v][hi | lo][step]
      auto wload = [&](half8 (&w)[2][2][2], int t) {
#pragma unroll
        for (int kv = 0; kv < 2; ++kv) {
          const unsigned o = (unsigned)((kv + 1) * NP + t) * 2048u;
          w[kv][0][0] = tf_frag(rqh, wq_off, o); w[kv][0][1] = tf_frag(rqh, wq_off, o + 1024);
          w[kv][1][0] = tf_frag(rql, wq_off, o); w[kv][1][1] = tf_frag(rql, wq_off, o + 1024);
        }
      };
      auto wmma = [&](const half8 (&w)[2][2][2], int t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8 ah = *reinterpret_cast<const half8*>(Ah + li * AST + 32 * t + 16 * hh + 8 * s);
          const half8 al = *reinterpret_cast<const half8*>(Al + li * AST + 32 * t + 16 * hh + 8 * s);
          ak = lf_mfma3(ah, al, w[0][0][s], w[0][1][s], ak);
          av = lf_mfma3(ah, al, w[1][0][s], w[1][1][s], av);
        }
      };
      // No fragment load is in flight while the matrix instructions of a pair run (attn_fused.h: the hand-over note): both sets are
      // requested, waited for in full, consumed. (The other three waves of the SIMD's two blocks fill the wait: no time lost, measured.)
#pragma unroll 1
      for (int t = 0; t < NP; t += 2) {
        wload(w0, t);
        wload(w1, t + 1);
        TF_WAIT_SET8(w0[0][0][0], w0[0][0][1], w0[0][1][0], w0[0][1][1], w0[1][0][0], w0[1][0][1], w0[1][1][0], w0[1][1][1]);
        TF_WAIT_SET8(w1[0][0][0], w1[0][0][1], w1[0][1][0], w1[0][1][1], w1[1][0][0], w1[1][0][1], w1[1][1][0], w1[1][1][1]);
        wmma(w0, t);
        wmma(w1, t + 1);
        asm volatile("" : "+v"(ak), "+v"(av) :: "memory");          // (the next loads stay behind these matrix instructions)
      }
    }
    if (tile + 1 < tile1) fetch(tile + 1);
    __syncthreads();                                     // the planes may be rewritten
    const int tok0 = tile * 32;
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      ak[e] *= inv_qkv; av[e] *= inv_qkv;
      if (tok0 + tf_key(e, hh) < p.n_tok) mt = fmaxf(mt, ak[e]);
    }
    float m0, m1;
    tf_halves(mt, m0, m1);
    const float m_new = fmaxf(m_run, fmaxf(m0, m1));
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {       // a larger maximum somewhere in the head: earlier sums move to it
      const float f = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
      z_run *= f;
      if (hh == 0) Fs[h][li] = f;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 f4 = *reinterpret_cast<const float4*>(&Fs[h][8 * c + 4 * hh]);
        ctx[4 * c] *= f4.x; ctx[4 * c + 1] *= f4.y; ctx[4 * c + 2] *= f4.z; ctx[4 * c + 3] *= f4.w;
      }
      __builtin_amdgcn_wave_barrier();
      m_run = m_new;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float ks = tok0 + tf_key(e, hh) < p.n_tok ? expf(ak[e] - m_run) : 0.f;
      z_run += ks;
      ak[e] = ks;
    }
    // ctx_raw[d][e] += sum over the tile's tokens: step r takes the two tokens register r holds (one per lane half)
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[r], av[r], ctx, 0, 0, 0);
  }
  float z0, z1;
  tf_halves(z_run, z0, z1);
  float* po = p.part + ((int64_t)blockIdx.x * TF_HEADS + h) * LF_PART;
  if (hh == 0) { po[li] = m_run; po[32 + li] = z0 + z1; }
#pragma unroll
  for (int r = 0; r < 16; ++r) po[64 + tf_key(r, hh) * 32 + li] = ctx[r];
}

// ------------------------------------------------------------------------------------------------ pass 2: tokens -> output rows
template <int C>
__global__ __launch_bounds__(256, 2) void lattn_wide_out_kernel(LFusedP p) {
  constexpr int AST = C + 8, NP = C / 32, NJ = C / 64, MT = C / 128;
  __shared__ __attribute__((aligned(16))) _Float16 Ah[32 * AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[32 * AST];
  __shared__ __attribute__((aligned(16))) _Float16 Oh[32 * LW_OST];
  __shared__ __attribute__((aligned(16))) _Float16 Ol[32 * LW_OST];
  __shared__ float Cmax[TF_HEADS];
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  const int unit = (int)(blockIdx.x / (unsigned)p.chunks), chunk = (int)(blockIdx.x - (unsigned)unit * (unsigned)p.chunks);
  const unsigned wq_off = (unsigned)(h * 3 * NP * 2 * 64 + lane) * 16u;     // bytes; q rows (ti = 0): + (t 2 + s) 1024
  const unsigned wo_off = (unsigned)(h * MT * 4 * 2 * 64 + lane) * 16u;     // + ((mt 4 + t) 2 + s) 1024
  const __amdgpu_buffer_rsrc_t rqh = tf_rsrc(p.wq_hi, 3 * TF_HD * C * 2), rql = tf_rsrc(p.wq_lo, 3 * TF_HD * C * 2);
  const __amdgpu_buffer_rsrc_t roh = tf_rsrc(p.wo_hi, C * TF_HD * 2), rol = tf_rsrc(p.wo_lo, C * TF_HD * 2);
  // ctx^T fragments of this (frame, head): step r contracts the features d = tf_key(r, hh); lane li = output feature e
  float ctxf[16];
  float amc = 0.f;
  {
    const float* cu = p.ctx + ((int64_t)unit * TF_HEADS + h) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ctxf[r] = cu[tf_key(r, hh) * 32 + li]; amc = fmaxf(amc, fabsf(ctxf[r])); }
  }
  amc = tf_wave_max(amc);
  if (lane == 0) Cmax[h] = amc;
  __syncthreads();
  // |out[e]| <= max_d |ctx[d][e]| sum_d qs[d] = scale max|ctx|; one scale for the four heads' tiles (they share the reduction of to_out)
  const float so = scale_from_amax(p.scale * fmaxf(fmaxf(Cmax[0], Cmax[1]), fmaxf(Cmax[2], Cmax[3])));
  const float inv_o = 1.0f / (so * p.wo_scale[0]);
  const float ps = lw_plane_scale<C>(p.gamma, lc4);
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float* xu = p.x + (int64_t)unit * p.n_tok * C;
  float* yu = p.y + (int64_t)unit * p.n_tok * C;
  const int tile0 = chunk * p.tiles_per_chunk;
  const int ntiles = (p.n_tok + 31) >> 5;
  const int tile1 = min(ntiles, tile0 + p.tiles_per_chunk);
  float am = 0.f;
  float4 nx0[NJ], nx1[NJ];
  auto fetch = [&](int tile) {
    const int r0 = tile * 32 + lrow, r1 = r0 + 16;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      nx0[j] = r0 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r0 * C + 64 * j + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
      nx1[j] = r1 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r1 * C + 64 * j + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (tile0 < tile1) fetch(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    lw_ln_row<C>(nx0, p.gamma, p.eps, ps, Ah, Al, lrow, lc4);
    lw_ln_row<C>(nx1, p.gamma, p.eps, ps, Ah, Al, 16 + lrow, lc4);
    __syncthreads();                                                                  // S1: planes of this tile complete
    // q^T of the head: [feature][token], a lane owns one token and the features 8 c + 4 hh + (0..3)
    f32x16 aq = lf_zero();
    {
      half8 w0[2][2], w1[2][2];                          // [hi | lo][step]
      auto wload = [&](half8 (&w)[2][2], int t) {
        const unsigned o = (unsigned)t * 2048u;
        w[0][0] = tf_frag(rqh, wq_off, o); w[0][1] = tf_frag(rqh, wq_off, o + 1024);
        w[1][0] = tf_frag(rql, wq_off, o); w[1][1] = tf_frag(rql, wq_off, o + 1024);
      };
      auto wmma = [&](const half8 (&w)[2][2], int t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8 bh = *reinterpret_cast<const half8*>(Ah + li * AST + 32 * t + 16 * hh + 8 * s);
          const half8 bl = *reinterpret_cast<const half8*>(Al + li * AST + 32 * t + 16 * hh + 8 * s);
          aq = lf_mfma3(w[0][s], w[1][s], bh, bl, aq);
        }
      };
#pragma unroll 1
      for (int t = 0; t < NP; t += 2) {        // (requested, waited for in full, consumed: see the first pass)
        wload(w0, t);
        wload(w1, t + 1);
        TF_WAIT_SET4(w0[0][0], w0[0][1], w0[1][0], w0[1][1]);
        TF_WAIT_SET4(w1[0][0], w1[0][1], w1[1][0], w1[1][1]);
        wmma(w0, t);
        wmma(w1, t + 1);
        asm volatile("" : "+v"(aq) :: "memory");
      }
    }
    if (tile + 1 < tile1) fetch(tile + 1);
    // qs = scale softmax over the head's 32 features of the token (16 here, 16 in lane ^ 32)
    {
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) { aq[e] *= inv_qkv; mx = fmaxf(mx, aq[e]); }
      float m0, m1;
      tf_halves(mx, m0, m1);
      mx = fmaxf(m0, m1);
      float l = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { aq[e] = expf(aq[e] - mx); l += aq[e]; }
      float l0, l1;
      tf_halves(l, l0, l1);
      const float il = p.scale / (l0 + l1);
#pragma unroll
      for (int e = 0; e < 16; ++e) aq[e] *= il;
    }
    // out^T[e][tok] = sum_d ctx[d][e] qs[tok][d] (exact fp32, qs in place)
    f32x16 oT = lf_zero();
#pragma unroll
    for (int r = 0; r < 16; ++r) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(ctxf[r], aq[r], oT, 0, 0, 0);
    // (hi, lo) planes of the head's out tile: lane (token li, hh) owns features 8 c + 4 hh + (0..3)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      half4v hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = oT[4 * c + e] * so;
        hv[e] = (_Float16)t;
        lv[e] = (_Float16)(t - (float)hv[e]);
      }
      *reinterpret_cast<half4v*>(Oh + li * LW_OST + 32 * h + 8 * c + 4 * hh) = hv;
      *reinterpret_cast<half4v*>(Ol + li * LW_OST + 32 * h + 8 * c + 4 * hh) = lv;
    }
    __syncthreads();                                                                  // S2: out planes of the four heads complete
    // ---- to_out: channels h C/4 + 32 mt + (0..31) over all 128 features (k-step pair t = head t), + bias + residual
    const int tk = tile * 32 + li;
    const bool tok = tk < p.n_tok;
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
      f32x16 y = lf_zero();
      half8 u0[2][2], u1[2][2];
      auto uload = [&](half8 (&u)[2][2], int t) {
        const unsigned o = (unsigned)(mt * 4 + t) * 2048u;
        u[0][0] = tf_frag(roh, wo_off, o); u[0][1] = tf_frag(roh, wo_off, o + 1024);
        u[1][0] = tf_frag(rol, wo_off, o); u[1][1] = tf_frag(rol, wo_off, o + 1024);
      };
      auto umma = [&](const half8 (&u)[2][2], int t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8 oh = *reinterpret_cast<const half8*>(Oh + li * LW_OST + 32 * t + 16 * hh + 8 * s);
          const half8 ol = *reinterpret_cast<const half8*>(Ol + li * LW_OST + 32 * t + 16 * hh + 8 * s);
          y = lf_mfma3(u[0][s], u[1][s], oh, ol, y);
        }
      };
#pragma unroll 1
      for (int t = 0; t < TF_HEADS; t += 2) {
        uload(u0, t);
        uload(u1, t + 1);
        TF_WAIT_SET4(u0[0][0], u0[0][1], u0[1][0], u0[1][1]);
        TF_WAIT_SET4(u1[0][0], u1[0][1], u1[1][0], u1[1][1]);
        umma(u0, t);
        umma(u1, t + 1);
        asm volatile("" : "+v"(y) :: "memory");
      }
      if (tok) {
        const int ch0 = h * (C / 4) + 32 * mt + 4 * hh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 xr = *reinterpret_cast<const float4*>(xu + (int64_t)tk * C + ch0 + 8 * c);      // (L2 hits: the block read these rows for the LayerNorm)
          const float4 b4 = p.bias_out ? *reinterpret_cast<const float4*>(p.bias_out + ch0 + 8 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
          float4 r;
          r.x = (y[4 * c] * inv_o + b4.x) + xr.x; r.y = (y[4 * c + 1] * inv_o + b4.y) + xr.y;
          r.z = (y[4 * c + 2] * inv_o + b4.z) + xr.z; r.w = (y[4 * c + 3] * inv_o + b4.w) + xr.w;
          *reinterpret_cast<float4*>(yu + (int64_t)tk * C + ch0 + 8 * c) = r;
          am = amax4(am, r);
        }
      }
    }
  }
  if (p.amax_rec) {        // wave_amax_emit with the lane id from mbcnt (see attn_fused_wide.hip)
    am = wave_max(am);
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
      atomicMax(reinterpret_cast<unsigned*>(p.amax_rec) + (((int)blockIdx.x * TF_HEADS + h) & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE, __float_as_uint(am));
  }
}

// the two passes of the wide block (linattn_fused.hip launches the merge between them)
int wdno_lattn_wide_ctx_launch(const LFusedP& p, int C, unsigned grid, hipStream_t st) {
  if (C == 128) lattn_wide_ctx_kernel<128><<<grid, 256, 0, st>>>(p);
  else if (C == 256) lattn_wide_ctx_kernel<256><<<grid, 256, 0, st>>>(p);
  else return WDNO_EUNSUPPORTED;
  return WDNO_OK;
}
int wdno_lattn_wide_out_launch(const LFusedP& p, int C, unsigned grid, hipStream_t st) {
  if (C == 128) lattn_wide_out_kernel<128><<<grid, 256, 0, st>>>(p);
  else if (C == 256) lattn_wide_out_kernel<256><<<grid, 256, 0, st>>>(p);
  else return WDNO_EUNSUPPORTED;
  return WDNO_OK;
}
