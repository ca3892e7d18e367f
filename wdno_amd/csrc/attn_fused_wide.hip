// attn_fused_wide.hip -- the temporal attention block (attn_fused.hip: Residual(PreNorm(dim, temporal Attention)), conv3d.py:165-174,
// 277-353) of the 128- and 256-channel levels of the smoke U-Net as ONE forward launch:
//
//     y = x + W_out . softmax(rot(scale q) rot(k)^T + bias) v ,      (q | k | v) = W_qkv . LayerNorm(x) ,     24 frames, 4 heads of 32.
//
// These levels are small (3 200 sequences at 20 x 20, 800 at 10 x 10) and ran layer by layer as LayerNorm -> to_qkv -> attention -> to_out,
// four launches of 10-50 us that each wait for the one before. What differs from the 64-channel kernel: the head's 96 rows of W_qkv no longer
// fit the register file (96 x C halves x 2 planes per wave), so the weight fragments are STREAMED from L2 per sequence, two k-step pairs per
// trip: requested, waited for in full, consumed (attn_fused.h: why not overlapped). They are packed in FRAGMENT ORDER (pack modes 10 / 11 of
// csrc/conv_h3.hip): the 64 lanes of one operand load read 1 KB of contiguous memory. Read from the row-major [384][C] operand the same
// instruction touches 32 rows = 64 half-used cache lines, and the kernel is bound by the address / tag rate of the CU's one vector cache:
// measured 132 -> 85 us at [8,24,20,20,128], 77 -> 46 us at [8,24,10,10,256], 42 -> 28 us at [8,24,10,10,128] (layer by layer: 153 / 74 /
// 54 us). The contraction index may be permuted freely as long as both operands agree: k-step pair t of lane half hh covers channels
// 32 t + 16 hh .. + 15.
//   * LayerNorm by the block (16 lanes per row, DPP sums) -> fp16 (hi, lo) planes [24 tokens + one zero row][C] in LDS.
//   * Projection, rotary, exact-fp32 score / value products and softmax exactly as in attn_fused.hip (one wave = one head).
//   * to_out: the four heads' O tiles go to LDS as (hi, lo) planes [token][128] under ONE scale (max|v| over the heads: rows of P sum to
//     1); wave w then owns output channels w C/4 .. and contracts over all 128 features (k-step pair t = head t), W_out streamed like
//     W_qkv; residual added and rows stored from the accumulator layout (a lane owns a token and 4-channel runs).
#include "attn_fused.h"

#define TW_OST 136    /* halves per token row of an O plane (128 features + 8: 272 B = conflict-free ds_read_b128 fragments) */

template <int C>
__global__ __launch_bounds__(256, 2) void tattn_wide_fwd_kernel(TFusedP p) {
  constexpr int AST = C + 8;       // halves per token row of an A plane (pitch = 4 banks mod 64)
  constexpr int NP = C / 32;       // k-step pairs of the projection
  constexpr int NJ = C / 64;       // float4 chunks of a row per lane (16 lanes per row)
  constexpr int MT = C / 128;      // 32-channel output tiles per wave of to_out
  __shared__ __attribute__((aligned(16))) _Float16 Ah[25 * AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[25 * AST];
  __shared__ __attribute__((aligned(16))) _Float16 Oh[25 * TW_OST];
  __shared__ __attribute__((aligned(16))) _Float16 Ol[25 * TW_OST];
  __shared__ __attribute__((aligned(16))) float Vt[TF_HEADS][32 * TF_VST];
  __shared__ __attribute__((aligned(16))) float2 Rt[32 * TF_RST];
  __shared__ __attribute__((aligned(16))) float Bs[TF_HEADS][TF_NT * TF_BST];
  __shared__ float Vmax[TF_HEADS];
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const bool tok = li < TF_NT;
  const int rb = tok ? li : TF_NT;               // plane row of this lane's token (nonexistent tokens: the zero row)
  const int lrow = tid >> 4, lc4 = tid & 15;

  // ---- per-kernel tables: zero rows, rotary (cos, sin) pairs, relative-position bias
  for (int i = tid; i < AST; i += 256) { Ah[TF_NT * AST + i] = (_Float16)0.f; Al[TF_NT * AST + i] = (_Float16)0.f; }
  for (int i = tid; i < TW_OST; i += 256) { Oh[TF_NT * TW_OST + i] = (_Float16)0.f; Ol[TF_NT * TW_OST + i] = (_Float16)0.f; }
  for (int i = tid; i < 32 * 16; i += 256) {
    const int t = i >> 4, j = i & 15;
    float2 v = make_float2(1.f, 0.f);
    if (p.rcos && t < TF_NT) v = make_float2(p.rcos[t * 32 + 2 * j], p.rsin[t * 32 + 2 * j]);
    Rt[t * TF_RST + j] = v;
  }
  for (int i = tid; i < TF_HEADS * TF_NT * TF_BST; i += 256) {
    const int hd = i / (TF_NT * TF_BST), r = i - hd * (TF_NT * TF_BST), q = r / TF_BST, k = r - q * TF_BST;
    Bs[hd][r] = (p.bias && k < TF_NT) ? p.bias[(hd * TF_NT + q) * TF_NT + k] : 0.f;
  }
  float gm = 0.f;        // (the gain is re-read per row -- L1 hits -- rather than held across the matrix phases: registers)
#pragma unroll
  for (int j = 0; j < NJ; ++j) gm = amax4(gm, reinterpret_cast<const float4*>(p.gamma)[16 * j + lc4]);
  const float ps = scale_from_amax(sqrtf((float)C) * group_max<16>(gm));        // |LayerNorm(x)| <= sqrt(C) max|g|
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float sw_o = p.wo_scale[0];
  const int64_t fstride = (int64_t)p.HW * C;
  // operands in fragment order (pack modes 10 / 11 of csrc/conv_h3.hip): [head][q|k|v][pair][step][lane][8] and [wave][tile][head][step][lane][8]
  // (raw buffer loads: descriptor + one 32-bit lane offset + a uniform offset, see attn_fused.h)
  const unsigned wq_off = (unsigned)(h * 3 * NP * 2 * 64 + lane) * 16u;     // bytes; + ((ti NP + t) 2 + s) 1024
  const unsigned wo_off = (unsigned)(h * MT * 4 * 2 * 64 + lane) * 16u;     // + ((mt 4 + t) 2 + s) 1024
  const __amdgpu_buffer_rsrc_t rqh = tf_rsrc(p.wq_hi, 3 * TF_HD * C * 2), rql = tf_rsrc(p.wq_lo, 3 * TF_HD * C * 2);
  const __amdgpu_buffer_rsrc_t roh = tf_rsrc(p.wo_hi, C * TF_HD * 2), rol = tf_rsrc(p.wo_lo, C * TF_HD * 2);
  float am = 0.f;

  float4 nx0[NJ], nx1[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { nx0[j] = make_float4(0.f, 0.f, 0.f, 0.f); nx1[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
  auto fetch = [&](int64_t r0) {
    const float* xr = p.x + r0 * C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      nx0[j] = *reinterpret_cast<const float4*>(xr + lrow * fstride + 64 * j + 4 * lc4);
      if (lrow < 8) nx1[j] = *reinterpret_cast<const float4*>(xr + (16 + lrow) * fstride + 64 * j + 4 * lc4);
    }
  };
  auto ln_row = [&](const float4 (&xin)[NJ], int row) {
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
    const float rstd = 1.0f / sqrtf(tf_row16_sum(q) * (1.0f / C) + p.eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 g = reinterpret_cast<const float4*>(p.gamma)[16 * j + lc4];
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
  };

  int nb = (int)(blockIdx.x / (unsigned)p.HW), npix = (int)(blockIdx.x - (unsigned)nb * (unsigned)p.HW);
  const int gstep_b = (int)(gridDim.x / (unsigned)p.HW), gstep_p = (int)(gridDim.x - (unsigned)gstep_b * (unsigned)p.HW);
  if ((int64_t)blockIdx.x < p.nseq) fetch((int64_t)nb * TF_NT * p.HW + npix);
  for (int64_t seq = blockIdx.x; seq < p.nseq; seq += gridDim.x) {
    const int64_t row0 = (int64_t)nb * TF_NT * p.HW + npix;      // row of frame 0; frame f at + f * HW
    nb += gstep_b; npix += gstep_p;
    if (npix >= p.HW) { npix -= p.HW; ++nb; }
    ln_row(nx0, lrow);
    if (lrow < 8) ln_row(nx1, 16 + lrow);
    __syncthreads();                                                                  // S1: planes of this sequence complete
    // ---- (q | k | v)^T of this head: [feature][token], weight fragments streamed (double-buffered by pair)
    f32x16 aq, ak, av;
#pragma unroll
    for (int e = 0; e < 16; ++e) { aq[e] = 0.f; ak[e] = 0.f; av[e] = 0.f; }
    {
      // two fragment sets per trip; the loop over pairs is NOT unrolled (unrolled, the scheduler hoists every load to the top: 162 spilled
      // registers at C = 256)
      half8 w0h[3][2], w0l[3][2], w1h[3][2], w1l[3][2];
      auto wload = [&](half8 (&wh)[3][2], half8 (&wl)[3][2], int t) {
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {
          const unsigned o = (unsigned)(ti * NP + t) * 2048u;
          wh[ti][0] = tf_frag(rqh, wq_off, o); wh[ti][1] = tf_frag(rqh, wq_off, o + 1024);
          wl[ti][0] = tf_frag(rql, wq_off, o); wl[ti][1] = tf_frag(rql, wq_off, o + 1024);
        }
      };
      auto wmma = [&](const half8 (&wh)[3][2], const half8 (&wl)[3][2], int t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8 ah = *reinterpret_cast<const half8*>(Ah + rb * AST + 32 * t + 16 * hh + 8 * s);
          const half8 al = *reinterpret_cast<const half8*>(Al + rb * AST + 32 * t + 16 * hh + 8 * s);
          aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0][s], al, aq, 0, 0, 0);
          ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1][s], al, ak, 0, 0, 0);
          av = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[2][s], al, av, 0, 0, 0);
          aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0][s], ah, aq, 0, 0, 0);
          ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[1][s], ah, ak, 0, 0, 0);
          av = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[2][s], ah, av, 0, 0, 0);
          aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0][s], ah, aq, 0, 0, 0);
          ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1][s], ah, ak, 0, 0, 0);
          av = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[2][s], ah, av, 0, 0, 0);
        }
      };
#pragma unroll 1
      for (int t = 0; t < NP; t += 2) {        // both sets requested, waited for in full, consumed (attn_fused.h: the hand-over note)
        wload(w0h, w0l, t);
        wload(w1h, w1l, t + 1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0h[0][0]), "+v"(w0h[0][1]), "+v"(w0h[1][0]), "+v"(w0h[1][1]), "+v"(w0h[2][0]), "+v"(w0h[2][1]),
                     "+v"(w0l[0][0]), "+v"(w0l[0][1]), "+v"(w0l[1][0]), "+v"(w0l[1][1]), "+v"(w0l[2][0]), "+v"(w0l[2][1]) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w1h[0][0]), "+v"(w1h[0][1]), "+v"(w1h[1][0]), "+v"(w1h[1][1]), "+v"(w1h[2][0]), "+v"(w1h[2][1]),
                     "+v"(w1l[0][0]), "+v"(w1l[0][1]), "+v"(w1l[1][0]), "+v"(w1l[1][1]), "+v"(w1l[2][0]), "+v"(w1l[2][1]) :: "memory");
        wmma(w0h, w0l, t);
        wmma(w1h, w1l, t + 1);
        asm volatile("" : "+v"(aq), "+v"(ak), "+v"(av) :: "memory");
      }
    }
    // rows of the next sequence: requested now, needed after this sequence's attention and to_out (not before the projection: its two
    // fragment sets own the register file)
    if (seq + gridDim.x < p.nseq) fetch((int64_t)nb * TF_NT * p.HW + npix);
#pragma unroll
    for (int e = 0; e < 16; ++e) { aq[e] *= inv_qkv; ak[e] *= inv_qkv; av[e] *= inv_qkv; }
    // V tile for the transposed read; max|v| over the heads bounds |out| (rows of P sum to 1)
    float amv = 0.f;
    float* vt = Vt[h];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 v4 = make_float4(av[4 * c], av[4 * c + 1], av[4 * c + 2], av[4 * c + 3]);
      amv = amax4(amv, v4);
      *reinterpret_cast<float4*>(vt + li * TF_VST + 8 * c + 4 * hh) = v4;
    }
    amv = tf_wave_max(amv);
    if (lane == 0) Vmax[h] = amv;
    // q * scale, rotary on q and k (pairs (2i, 2i + 1) = accumulator registers (2 j, 2 j + 1))
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
    // ---- S^T = K Q^T (exact fp32), softmax over the keys of this lane's query
    f32x16 sT;
#pragma unroll
    for (int e = 0; e < 16; ++e) sT[e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[e], aq[e], sT, 0, 0, 0);
    {
      const int qb = tok ? li : TF_NT - 1;              // (columns of nonexistent queries are never stored)
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < 3; ++c) {                     // keys 8 c + 4 hh + (0..3) < 24
        const float4 b4 = *reinterpret_cast<const float4*>(Bs[h] + qb * TF_BST + 8 * c + 4 * hh);
        sT[4 * c] += b4.x; sT[4 * c + 1] += b4.y; sT[4 * c + 2] += b4.z; sT[4 * c + 3] += b4.w;
        mx = fmaxf(fmaxf(mx, fmaxf(sT[4 * c], sT[4 * c + 1])), fmaxf(sT[4 * c + 2], sT[4 * c + 3]));
      }
      float m0, m1;
      tf_halves(mx, m0, m1);
      mx = fmaxf(m0, m1);
      float l = 0.f;
#pragma unroll
      for (int e = 0; e < 12; ++e) { sT[e] = expf(sT[e] - mx); l += sT[e]; }
#pragma unroll
      for (int e = 12; e < 16; ++e) sT[e] = 0.f;            // keys 24 .. 31 do not exist
      float l0, l1;
      tf_halves(l, l0, l1);
      const float il = 1.0f / (l0 + l1);
#pragma unroll
      for (int e = 0; e < 12; ++e) sT[e] *= il;
    }
    // ---- O^T = V^T P^T
    __builtin_amdgcn_wave_barrier();
    float va[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) va[m] = vt[tf_key(m, hh) * TF_VST + li];
    f32x16 oT;
#pragma unroll
    for (int e = 0; e < 16; ++e) oT[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(va[m], sT[m], oT, 0, 0, 0);
    __syncthreads();                                                                  // S2: max|v| of the four heads; planes A free
    const float so = scale_from_amax(fmaxf(fmaxf(Vmax[0], Vmax[1]), fmaxf(Vmax[2], Vmax[3])));
    if (tok) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        half4v hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = oT[4 * c + e] * so;
          hv[e] = (_Float16)t;
          lv[e] = (_Float16)(t - (float)hv[e]);
        }
        *reinterpret_cast<half4v*>(Oh + li * TW_OST + 32 * h + 8 * c + 4 * hh) = hv;
        *reinterpret_cast<half4v*>(Ol + li * TW_OST + 32 * h + 8 * c + 4 * hh) = lv;
      }
    }
    __syncthreads();                                                                  // S3: O planes of the four heads complete
    // ---- to_out: channels h C/4 + 32 mt + (0..31) over all 128 features
    const float inv_o = 1.0f / (so * sw_o);
    const float* xb = p.x + row0 * C;
    float* yb = p.y + row0 * C;
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
      f32x16 y;
#pragma unroll
      for (int e = 0; e < 16; ++e) y[e] = 0.f;
      half8 u0h[2], u0l[2], u1h[2], u1l[2];                 // two fragment sets, as in the projection
      auto uload = [&](half8 (&uh)[2], half8 (&ul)[2], int t) {
        const unsigned o = (unsigned)(mt * 4 + t) * 2048u;
        uh[0] = tf_frag(roh, wo_off, o); uh[1] = tf_frag(roh, wo_off, o + 1024);
        ul[0] = tf_frag(rol, wo_off, o); ul[1] = tf_frag(rol, wo_off, o + 1024);
      };
      auto umma = [&](const half8 (&uh)[2], const half8 (&ul)[2], int t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8 oh = *reinterpret_cast<const half8*>(Oh + rb * TW_OST + 32 * t + 16 * hh + 8 * s);
          const half8 ol = *reinterpret_cast<const half8*>(Ol + rb * TW_OST + 32 * t + 16 * hh + 8 * s);
          y = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh[s], ol, y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul[s], oh, y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh[s], oh, y, 0, 0, 0);
        }
      };
#pragma unroll 1
      for (int t = 0; t < TF_HEADS; t += 2) {
        uload(u0h, u0l, t);
        uload(u1h, u1l, t + 1);
        TF_WAIT_SET4(u0h[0], u0h[1], u0l[0], u0l[1]);
        TF_WAIT_SET4(u1h[0], u1h[1], u1l[0], u1l[1]);
        umma(u0h, u0l, t);
        umma(u1h, u1l, t + 1);
        asm volatile("" : "+v"(y) :: "memory");
      }
      // the residual values this lane adds (L2 hits: the block read these rows for the LayerNorm)
      float4 xr[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        xr[c] = tok ? *reinterpret_cast<const float4*>(xb + li * fstride + h * (C / 4) + 32 * mt + 8 * c + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (tok) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float4 r;
          r.x = y[4 * c] * inv_o + xr[c].x; r.y = y[4 * c + 1] * inv_o + xr[c].y;
          r.z = y[4 * c + 2] * inv_o + xr[c].z; r.w = y[4 * c + 3] * inv_o + xr[c].w;
          *reinterpret_cast<float4*>(yb + li * fstride + h * (C / 4) + 32 * mt + 8 * c + 4 * hh) = r;
          am = amax4(am, r);
        }
      }
    }
  }
  if (p.amax_rec) {        // wave_amax_emit with the lane id from mbcnt (threadIdx.x kept alive across the loop costs two spilled registers at C = 256)
    am = wave_max(am);
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
      atomicMax(reinterpret_cast<unsigned*>(p.amax_rec) + (((int)blockIdx.x * TF_HEADS + h) & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE, __float_as_uint(am));
  }
}

static int tw_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

// C = 128 / 256, 24 frames, forward only (attn_fused.hip routes here; qkv_out / rec_v are not produced)
int wdno_tattn_fused_fwd_wide_launch(const TFusedP& p, int C, hipStream_t st) {
  int64_t grid = 2 * (int64_t)tw_num_cus();
  if (grid > p.nseq) grid = p.nseq;
  if (C == 128) tattn_wide_fwd_kernel<128><<<(int)grid, 256, 0, st>>>(p);
  else if (C == 256) tattn_wide_fwd_kernel<256><<<(int)grid, 256, 0, st>>>(p);
  else return WDNO_EUNSUPPORTED;
  return wdno_check_launch();
}
