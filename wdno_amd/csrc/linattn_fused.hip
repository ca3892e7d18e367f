// linattn_fused.hip -- the SpatialLinearAttention block of the smoke U-Net's 64-channel levels (conv3d.py:165-174 LayerNorm, :232-258
// SpatialLinearAttention, wrapped as Residual(PreNorm(...))) FORWARD as two launches + a merge, for passes that need no gradient (sampling):
//
//     y = x + b_out + W_out . out ,   out[n][e] = sum_d ctx[d][e] qs[n][d] ,   ctx[d][e] = sum_n softmax_n(k)[n][d] v[n][e] ,
//     qs = scale softmax_d(q) ,       (q | k | v) = W_qkv . LayerNorm(x)                      per frame (unit) and head; n = the H W tokens
//
// Layer by layer the block writes the [pixels x 384] projections (472 MB at the bench size) and reads them three times; here a token tile
// (32 consecutive pixels of a frame: 8 KB of x) is normalised and projected in LDS / registers by each pass that needs it:
//   lattn_fused_ctx_kernel   k, v of a tile on the split fp16 MFMA with the operands SWAPPED (A = token planes, B = weight rows), so the
//                            accumulators come out as [token][feature] -- a lane owns one feature d and 16 tokens: the softmax over the
//                            tokens is lane-local (+ one exchange of the lane halves), and the context product contracts over the tokens
//                            with both operands in place (exact-fp32 MFMA, step m = the two tokens accumulator register m holds).
//                            Running maximum per feature (online softmax); a block = (frame, token chunk), four waves = four heads.
//   lattn_fused_merge_kernel the chunks of a frame merged in chunk order (softmax merge), ctx = ctx_raw / Z
//   lattn_fused_out_kernel   q of a tile in the usual [feature][token] layout (softmax over d inside a lane pair), out^T = ctx^T qs^T on
//                            the exact-fp32 MFMA with qs in place, to_out on the split MFMA straight from the accumulators, heads summed
//                            through LDS with the bias and the residual (as attn_fused.hip).
#include "linattn_fused.h"

// LayerNorm of one row by its 16 lanes (norm.hip: layernorm_kernel) -> (hi, lo) planes
__device__ __forceinline__ void lf_ln_row(float4 xv, float4 g, float eps, float ps, _Float16* __restrict__ Ah, _Float16* __restrict__ Al, int row, int c4) {
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

// ------------------------------------------------------------------------------------------------ pass 1: per-chunk context
__global__ __launch_bounds__(256, 2) void lattn_fused_ctx_kernel(LFusedP p) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) float Fs[TF_HEADS][32];
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  const int unit = (int)(blockIdx.x / (unsigned)p.chunks), chunk = (int)(blockIdx.x - (unsigned)unit * (unsigned)p.chunks);
  // weight rows of k and v of this head: as the COLUMN operand of the swapped product, lane li = feature d, 8 channels per k-step
  half8 wkh[4], wkl[4], wvh[4], wvl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ok = (TF_HD + h * 32 + li) * TF_C + 16 * s + 8 * hh, ov = ok + TF_HD * TF_C;
    wkh[s] = *reinterpret_cast<const half8*>(p.wq_hi + ok); wkl[s] = *reinterpret_cast<const half8*>(p.wq_lo + ok);
    wvh[s] = *reinterpret_cast<const half8*>(p.wq_hi + ov); wvl[s] = *reinterpret_cast<const half8*>(p.wq_lo + ov);
  }
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));        // |LayerNorm(x)| <= sqrt(64) max|g|
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float* xu = p.x + (int64_t)unit * p.n_tok * TF_C;
  const int tile0 = chunk * p.tiles_per_chunk;
  const int ntiles = (p.n_tok + 31) >> 5;
  const int tile1 = min(ntiles, tile0 + p.tiles_per_chunk);

  float m_run = -INFINITY, z_run = 0.f;                 // feature d = li: running maximum over the tokens so far (same in both lane halves), partial sum of this half
  f32x16 ctx = lf_zero();                               // ctx_raw[d][e]: lane (e, hh), register r <-> d = tf_key(r, hh)
  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0;
  auto fetch = [&](int tile) {
    const int r0 = tile * 32 + lrow, r1 = r0 + 16;
    nx0 = r0 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r0 * TF_C + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
    nx1 = r1 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r1 * TF_C + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  if (tile0 < tile1) fetch(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    lf_ln_row(nx0, g4, p.eps, ps, Ah, Al, lrow, lc4);
    lf_ln_row(nx1, g4, p.eps, ps, Ah, Al, 16 + lrow, lc4);
    __syncthreads();
    if (tile + 1 < tile1) fetch(tile + 1);
    // k[tok][d], v[tok][e] of the tile: rows = tokens (A = the token planes), columns = features (B = the weight rows)
    f32x16 ak = lf_zero(), av = lf_zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 ah = *reinterpret_cast<const half8*>(Ah + li * TF_AST + 16 * s + 8 * hh);
      const half8 al = *reinterpret_cast<const half8*>(Al + li * TF_AST + 16 * s + 8 * hh);
      ak = lf_mfma3(ah, al, wkh[s], wkl[s], ak);
      av = lf_mfma3(ah, al, wvh[s], wvl[s], av);
    }
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

// chunks of a frame merged in chunk order: ctx[d][e] = sum_c ctx_c[d][e] exp(m_c - M) / sum_c Z_c exp(m_c - M)
// (1024 threads, one per context entry: with 256 a thread walked the chunks -- two dependent strided loads per chunk -- for four entries in turn,
// 17 us per launch at batch 1, profiles/r06_sampling_b1_kernel_stats.md; the sums per entry are unchanged)
__global__ __launch_bounds__(1024) void lattn_fused_merge_kernel(const float* __restrict__ part, float* __restrict__ ctx, int chunks,
                                                                  float* __restrict__ kstat /* optional [units][heads][2][32]: max_n k, 1 / Z */) {
  const int64_t unit = blockIdx.x / TF_HEADS;
  const int h = (int)(blockIdx.x - unit * TF_HEADS);
  const float* p0 = part + ((unit * chunks) * TF_HEADS + h) * (int64_t)LF_PART;
  const int64_t cstride = (int64_t)TF_HEADS * LF_PART;
  float* co = ctx + (int64_t)blockIdx.x * 1024;
  for (int o = threadIdx.x; o < 1024; o += blockDim.x) {
    const int d = o >> 5;
    float M = -INFINITY;
    for (int c = 0; c < chunks; ++c) M = fmaxf(M, p0[c * cstride + d]);
    float Z = 0.f, v = 0.f;
    for (int c = 0; c < chunks; ++c) {
      const float mc = p0[c * cstride + d];
      const float wgt = mc == -INFINITY ? 0.f : expf(mc - M);
      Z += p0[c * cstride + 32 + d] * wgt;
      v += p0[c * cstride + 64 + o] * wgt;
    }
    co[o] = v / Z;
    if (kstat && (o & 31) == 0) {
      kstat[(int64_t)blockIdx.x * 64 + d] = M;
      kstat[(int64_t)blockIdx.x * 64 + 32 + d] = 1.0f / Z;
    }
  }
}

// ------------------------------------------------------------------------------------------------ pass 2: tokens -> output rows
__global__ __launch_bounds__(256, 2) void lattn_fused_out_kernel(LFusedP p) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) float Yp[TF_HEADS][32 * TF_YST];
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 4, lc4 = tid & 15;
  const int unit = (int)(blockIdx.x / (unsigned)p.chunks), chunk = (int)(blockIdx.x - (unsigned)unit * (unsigned)p.chunks);
  half8 wqh[4], wql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int off = (h * 32 + li) * TF_C + 16 * s + 8 * hh;
    wqh[s] = *reinterpret_cast<const half8*>(p.wq_hi + off);
    wql[s] = *reinterpret_cast<const half8*>(p.wq_lo + off);
  }
  // to_out as in attn_fused.hip: output channel 32 ct + li; reduction slot t of k-step s <-> feature 16 s + 8 (t >> 2) + 4 hh + (t & 3)
  half8 woh[2][2], wol[2][2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int off = (32 * ct + li) * TF_HD + 32 * h + 16 * s + 4 * hh;
      const half4v a = *reinterpret_cast<const half4v*>(p.wo_hi + off), b = *reinterpret_cast<const half4v*>(p.wo_hi + off + 8);
      const half4v c = *reinterpret_cast<const half4v*>(p.wo_lo + off), d = *reinterpret_cast<const half4v*>(p.wo_lo + off + 8);
      woh[ct][s] = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
      wol[ct][s] = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  // ctx^T fragments of this (frame, head): step r contracts the features d = tf_key(r, hh); lane li = output feature e
  float ctxf[16];
  float amc = 0.f;
  {
    const float* cu = p.ctx + ((int64_t)unit * TF_HEADS + h) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ctxf[r] = cu[tf_key(r, hh) * 32 + li]; amc = fmaxf(amc, fabsf(ctxf[r])); }
  }
  const float so = scale_from_amax(p.scale * tf_wave_max(amc));               // |out[e]| <= max_d |ctx[d][e]| sum_d qs[d] = scale max|ctx|
  const float inv_o = 1.0f / (so * p.wo_scale[0]);
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float4 b4 = p.bias_out ? reinterpret_cast<const float4*>(p.bias_out)[lc4] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float* xu = p.x + (int64_t)unit * p.n_tok * TF_C;
  float* yu = p.y + (int64_t)unit * p.n_tok * TF_C;
  const int tile0 = chunk * p.tiles_per_chunk;
  const int ntiles = (p.n_tok + 31) >> 5;
  const int tile1 = min(ntiles, tile0 + p.tiles_per_chunk);
  float am = 0.f;
  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0;
  auto fetch = [&](int tile) {
    const int r0 = tile * 32 + lrow, r1 = r0 + 16;
    nx0 = r0 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r0 * TF_C + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
    nx1 = r1 < p.n_tok ? *reinterpret_cast<const float4*>(xu + (int64_t)r1 * TF_C + 4 * lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  if (tile0 < tile1) fetch(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    const float4 x0 = nx0, x1 = nx1;                     // the residual rows of this tile
    lf_ln_row(x0, g4, p.eps, ps, Ah, Al, lrow, lc4);
    lf_ln_row(x1, g4, p.eps, ps, Ah, Al, 16 + lrow, lc4);
    __syncthreads();
    if (tile + 1 < tile1) fetch(tile + 1);
    // q^T of the head: [feature][token], a lane owns one token and the features 8 c + 4 hh + (0..3)
    f32x16 aq = lf_zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 bh = *reinterpret_cast<const half8*>(Ah + li * TF_AST + 16 * s + 8 * hh);
      const half8 bl = *reinterpret_cast<const half8*>(Al + li * TF_AST + 16 * s + 8 * hh);
      aq = lf_mfma3(wqh[s], wql[s], bh, bl, aq);
    }
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
    // to_out, this head's 32 of the 128 reduction values: y_part[c][token]
    half8 oh[2], ol[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float t = oT[e] * so;
      const _Float16 th = (_Float16)t;
      oh[e >> 3][e & 7] = th;
      ol[e >> 3][e & 7] = (_Float16)(t - (float)th);
    }
    f32x16 y0 = lf_zero(), y1 = lf_zero();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      y0 = lf_mfma3(woh[0][s], wol[0][s], oh[s], ol[s], y0);
      y1 = lf_mfma3(woh[1][s], wol[1][s], oh[s], ol[s], y1);
    }
    {
      float* yp = Yp[h] + li * TF_YST + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(yp + 8 * c) = make_float4(y0[4 * c] * inv_o, y0[4 * c + 1] * inv_o, y0[4 * c + 2] * inv_o, y0[4 * c + 3] * inv_o);
        *reinterpret_cast<float4*>(yp + 32 + 8 * c) = make_float4(y1[4 * c] * inv_o, y1[4 * c + 1] * inv_o, y1[4 * c + 2] * inv_o, y1[4 * c + 3] * inv_o);
      }
    }
    __syncthreads();
    // heads summed, bias and residual added, rows stored (the lanes that loaded a row finish it)
    auto finish = [&](int row, const float4& xr) {
      const int tok = tile * 32 + row;
      if (tok >= p.n_tok) return;
      const int o = row * TF_YST + 4 * lc4;
      const float4 a = *reinterpret_cast<const float4*>(Yp[0] + o), b2 = *reinterpret_cast<const float4*>(Yp[1] + o);
      const float4 c = *reinterpret_cast<const float4*>(Yp[2] + o), d = *reinterpret_cast<const float4*>(Yp[3] + o);
      float4 r;
      r.x = (((a.x + b2.x) + (c.x + d.x)) + b4.x) + xr.x; r.y = (((a.y + b2.y) + (c.y + d.y)) + b4.y) + xr.y;
      r.z = (((a.z + b2.z) + (c.z + d.z)) + b4.z) + xr.z; r.w = (((a.w + b2.w) + (c.w + d.w)) + b4.w) + xr.w;
      *reinterpret_cast<float4*>(yu + (int64_t)tok * TF_C + 4 * lc4) = r;
      am = amax4(am, r);
    };
    finish(lrow, x0);
    finish(16 + lrow, x1);
    // (the next tile's planes are written before its barrier, the partial tiles after it: no third barrier)
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * TF_HEADS + h);
}

static int lf_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}
// token chunks per frame: ~3 blocks per CU in flight, at least two tiles each
static int lf_chunks(int64_t units, int n_tok) {
  const int ntiles = (n_tok + 31) / 32;
  int64_t c = (3 * (int64_t)lf_num_cus() + units - 1) / units;
  if (c > ntiles / 2) c = ntiles / 2;
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  return (int)c;
}

// 128 / 256 channels: linattn_fused_wide.hip (forward only; operands of pack modes 10 / 11)
int wdno_lattn_wide_ctx_launch(const LFusedP& p, int C, unsigned grid, hipStream_t st);
int wdno_lattn_wide_out_launch(const LFusedP& p, int C, unsigned grid, hipStream_t st);
extern int wdno_debug_mode;
extern "C" int wdno_lattn_fused_takes(int C, int heads, int n_tok) {
  return (C == TF_C || ((C == 128 || C == 256) && wdno_debug_mode != 62)) && heads == TF_HEADS && n_tok >= 32;
}
extern "C" size_t wdno_lattn_fused_ws_bytes(int64_t units, int n_tok) {
  return ((size_t)units * lf_chunks(units, n_tok) * TF_HEADS * LF_PART + (size_t)units * TF_HEADS * 1024) * sizeof(float);
}
extern "C" int wdno_lattn_fused_fwd(const float* x, const float* gamma, float eps, const void* wq_hi, const void* wq_lo, const float* wq_scale,
                                    const void* wo_hi, const void* wo_lo, const float* wo_scale, const float* bias_out, float* y, float* amax_rec,
                                    float* ctx_out, float* kstat_out, void* ws, size_t ws_bytes, int64_t units, int n_tok, int C, int heads,
                                    float scale, wdno_stream_t s) {
  WDNO_REQUIRE(x && gamma && wq_hi && wq_lo && wq_scale && wo_hi && wo_lo && wo_scale && y && ws && units > 0 && n_tok > 0);
  if (!wdno_lattn_fused_takes(C, heads, n_tok) || units * (int64_t)n_tok * C > 0x7fffffff0ll) return WDNO_EUNSUPPORTED;
  if (ws_bytes < wdno_lattn_fused_ws_bytes(units, n_tok)) return WDNO_EWORKSPACE;
  LFusedP p;
  p.x = x; p.gamma = gamma; p.eps = eps;
  p.wq_hi = (const _Float16*)wq_hi; p.wq_lo = (const _Float16*)wq_lo; p.wq_scale = wq_scale;
  p.wo_hi = (const _Float16*)wo_hi; p.wo_lo = (const _Float16*)wo_lo; p.wo_scale = wo_scale;
  p.bias_out = bias_out;
  p.n_tok = n_tok; p.chunks = lf_chunks(units, n_tok);
  const int ntiles = (n_tok + 31) / 32;
  p.tiles_per_chunk = (ntiles + p.chunks - 1) / p.chunks;
  p.scale = scale;
  p.part = (float*)ws;
  float* ctx = ctx_out ? ctx_out : (float*)ws + (size_t)units * p.chunks * TF_HEADS * LF_PART;
  p.ctx = ctx; p.y = y; p.amax_rec = amax_rec;
  if (units * p.chunks > 0x7fffffff) return WDNO_EUNSUPPORTED;
  hipStream_t st = as_stream(s);
  const unsigned grid = (unsigned)(units * p.chunks);
  if (C != TF_C) {
    int rc = wdno_lattn_wide_ctx_launch(p, C, grid, st);
    if (rc != WDNO_OK) return rc;
    lattn_fused_merge_kernel<<<(unsigned)(units * TF_HEADS), 1024, 0, st>>>(p.part, ctx, p.chunks, kstat_out);
    rc = wdno_lattn_wide_out_launch(p, C, grid, st);
    return rc != WDNO_OK ? rc : wdno_check_launch();
  }
  lattn_fused_ctx_kernel<<<grid, 256, 0, st>>>(p);
  lattn_fused_merge_kernel<<<(unsigned)(units * TF_HEADS), 1024, 0, st>>>(p.part, ctx, p.chunks, kstat_out);
  lattn_fused_out_kernel<<<grid, 256, 0, st>>>(p);
  return wdno_check_launch();
}
