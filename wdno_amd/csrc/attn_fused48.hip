// attn_fused48.hip -- the fused temporal attention block (attn_fused.hip: LayerNorm -> to_qkv -> rotary / attention over frames -> to_out +
// residual of Residual(PreNorm(EinopsToAndFrom(Attention))), conv3d.py:165-174, :277-353) for sequences of 48 frames: the super-resolution
// U-Net of the smoke tree works on 48 wavelet frames (inference_2d.py, BASELINE configs[4]), where the 24-frame kernel does not apply and a
// level-0 site costs LayerNorm + a 64 -> 384 projection (944 MB of qkv at [2,48,80,80]) + attention + to_out + two split passes.
// Forward only (sampling); a training step of that model runs the block layer by layer.
//
// One block = EIGHT waves = (head h, token tile tt) for ONE sequence at a time: tokens 0..31 and 32..47 (+ 16 rows of padding) are two MFMA
// row tiles. Per sequence:
//   * LayerNorm by the whole block into the fp16 (hi, lo) token planes [64][72] (rows 48..63 zero), as attn_fused.hip.
//   * wave (h, tt): (q | k | v)^T of its head and token tile on the split MFMA, accumulators [feature][token]; scale and rotary in place.
//   * k (rotated) and v go to LDS tiles [token][feature] of the head; after one block barrier every wave reads the OTHER token tile's k in
//     accumulator layout, so both score products S^T[kt] = K[kt] Q^T run on v_mfma_f32_32x32x2_f32 with operands in place; softmax over
//     the 48 keys of the lane's query is lane-pair-local (two accumulator tiles); O^T = sum_kt V[kt]^T P[kt]^T reads V transposed.
//   * to_out as attn_fused.hip; the partial [token][64] tiles of the four heads go through LDS (aliased onto the k / v tiles, which are
//     dead by then), summed with the residual x by the threads that loaded the rows.
// LDS 141 KB (planes 18, k + v tiles 74, rotary 9, bias 40): one block per CU, eight waves -- the occupancy of the 24-frame kernel.
#include "attn_fused.h"

#define T48_NT 48
#define T48_BST 52     /* floats per query row of the bias table (48 keys + pad: 13 sixteen-byte slots, coprime with the 8 slot classes) */
#define T48_A_BYTES (64 * TF_AST * 2)
#define T48_KV_BYTES (TF_HEADS * 64 * TF_VST * 4)
#define T48_RT_BYTES (64 * TF_RST * 8)
#define T48_BS_BYTES (TF_HEADS * T48_NT * T48_BST * 4)
#define T48_LDS_BYTES (2 * T48_A_BYTES + 2 * T48_KV_BYTES + T48_RT_BYTES + T48_BS_BYTES + 64)
static_assert(TF_HEADS * T48_NT * TF_YST * 4 <= 2 * T48_KV_BYTES, "the partial output tiles alias the k / v tiles");

__global__ __launch_bounds__(512, 1) void tattn_fused_fwd48_kernel(TFusedP p) {
  extern __shared__ __attribute__((aligned(16))) char smem48[];
  _Float16* Ah = reinterpret_cast<_Float16*>(smem48);
  _Float16* Al = reinterpret_cast<_Float16*>(smem48 + T48_A_BYTES);
  float* Kt = reinterpret_cast<float*>(smem48 + 2 * T48_A_BYTES);                  // [head][64 tokens][TF_VST]
  float* Vt = reinterpret_cast<float*>(smem48 + 2 * T48_A_BYTES + T48_KV_BYTES);
  float* Yp = Kt;                                                                  // [head][48 tokens][TF_YST], once k / v are dead
  float2* Rt = reinterpret_cast<float2*>(smem48 + 2 * T48_A_BYTES + 2 * T48_KV_BYTES);
  float* Bs = reinterpret_cast<float*>(smem48 + 2 * T48_A_BYTES + 2 * T48_KV_BYTES + T48_RT_BYTES);      // [head][48 queries][T48_BST]
  float* Av = reinterpret_cast<float*>(smem48 + 2 * T48_A_BYTES + 2 * T48_KV_BYTES + T48_RT_BYTES + T48_BS_BYTES);      // max|v| per (head, tile)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave & 3, tt = wave >> 2;
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const int trow = 32 * tt + li;                          // this lane's token
  const bool tok = trow < T48_NT;
  const int lrow = tid >> 4, lc4 = tid & 15;              // loader role: rows lrow (0..31) and 32 + lrow (lrow < 16)

  // ---- per-kernel operands
  half8 wqh[3][4], wql[3][4];
#pragma unroll
  for (int ti = 0; ti < 3; ++ti)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int off = (ti * TF_HD + h * 32 + li) * TF_C + 16 * s + 8 * hh;
      wqh[ti][s] = *reinterpret_cast<const half8*>(p.wq_hi + off);
      wql[ti][s] = *reinterpret_cast<const half8*>(p.wq_lo + off);
    }
  for (int i = tid; i < 64 * 16; i += 512) {
    const int t = i >> 4, j = i & 15;
    float2 v = make_float2(1.f, 0.f);
    if (p.rcos && t < T48_NT) v = make_float2(p.rcos[t * 32 + 2 * j], p.rsin[t * 32 + 2 * j]);
    Rt[t * TF_RST + j] = v;
  }
  for (int i = tid; i < TF_HEADS * T48_NT * T48_BST; i += 512) {
    const int hd = i / (T48_NT * T48_BST), r = i - hd * (T48_NT * T48_BST), q = r / T48_BST, k = r - q * T48_BST;
    Bs[i] = (p.bias && k < T48_NT) ? p.bias[(hd * T48_NT + q) * T48_NT + k] : 0.f;
  }
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));        // |LayerNorm(x)| <= sqrt(64) max|g|
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float sw_o = p.wo_scale[0];
  const int64_t fstride = (int64_t)p.HW * TF_C;
  const int qrow = trow < T48_NT ? trow : T48_NT - 1;     // bias row of this lane's query (lanes beyond the sequence read a real row; unused)
  float am = 0.f, stv = 0.f;

  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int64_t r0) {
    const float* xr = p.x + r0 * TF_C;
    nx0 = *reinterpret_cast<const float4*>(xr + lrow * fstride + 4 * lc4);
    if (lrow < 16) nx1 = *reinterpret_cast<const float4*>(xr + (32 + lrow) * fstride + 4 * lc4);
  };
  int nb = (int)(blockIdx.x / (unsigned)p.HW), npix = (int)(blockIdx.x - (unsigned)nb * (unsigned)p.HW);
  const int gstep_b = (int)(gridDim.x / (unsigned)p.HW), gstep_p = (int)(gridDim.x - (unsigned)gstep_b * (unsigned)p.HW);
  if ((int64_t)blockIdx.x < p.nseq) fetch((int64_t)nb * T48_NT * p.HW + npix);
  for (int64_t seq = blockIdx.x; seq < p.nseq; seq += gridDim.x) {
    const int64_t row0 = (int64_t)nb * T48_NT * p.HW + npix;
    const float* xb = p.x + row0 * TF_C;
    nb += gstep_b; npix += gstep_p;
    if (npix >= p.HW) { npix -= p.HW; ++nb; }
    // ---- rows -> LayerNorm -> planes (rows 48..63: zeros in, zeros out)
    tf_ln_row(nx0, g4, p.eps, ps, Ah, Al, lrow, lc4);
    tf_ln_row(nx1, g4, p.eps, ps, Ah, Al, 32 + lrow, lc4);
    __syncthreads();                                                            // (1) planes; also: the previous sequence's partial tiles are read
    if (seq + gridDim.x < p.nseq) fetch((int64_t)nb * T48_NT * p.HW + npix);
    // ---- (q | k | v)^T of this head and token tile: [feature][token]
    f32x16 aq, ak, av;
#pragma unroll
    for (int e = 0; e < 16; ++e) { aq[e] = 0.f; ak[e] = 0.f; av[e] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 ah = *reinterpret_cast<const half8*>(Ah + trow * TF_AST + 16 * s + 8 * hh);
      const half8 al = *reinterpret_cast<const half8*>(Al + trow * TF_AST + 16 * s + 8 * hh);
      aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(wqh[0][s], al, aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(wqh[1][s], al, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x16_f16(wqh[2][s], al, av, 0, 0, 0);
      aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(wql[0][s], ah, aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(wql[1][s], ah, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x16_f16(wql[2][s], ah, av, 0, 0, 0);
      aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(wqh[0][s], ah, aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(wqh[1][s], ah, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x16_f16(wqh[2][s], ah, av, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { aq[e] *= inv_qkv; ak[e] *= inv_qkv; av[e] *= inv_qkv; }
    // q * scale, rotary on q and k (pairs (2i, 2i + 1) = accumulator registers (2 j, 2 j + 1))
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 r4 = *reinterpret_cast<const float4*>(Rt + trow * TF_RST + 4 * c + 2 * hh);
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
    // k and v tiles of the head: [token][feature]; max|v| bounds |out| (rows of P sum to 1)
    float amv = 0.f;
    {
      float* kt = Kt + (h * 64 + trow) * TF_VST + 4 * hh;
      float* vt = Vt + (h * 64 + trow) * TF_VST + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 v4 = make_float4(av[4 * c], av[4 * c + 1], av[4 * c + 2], av[4 * c + 3]);
        amv = amax4(amv, v4);
        *reinterpret_cast<float4*>(vt + 8 * c) = v4;
        *reinterpret_cast<float4*>(kt + 8 * c) = make_float4(ak[4 * c], ak[4 * c + 1], ak[4 * c + 2], ak[4 * c + 3]);
      }
    }
    stv = fmaxf(stv, amv);
    amv = tf_wave_max(amv);
    if (lane == 0) Av[h * 2 + tt] = amv;
    __syncthreads();                                                            // (2) k / v tiles, max|v|
    amv = fmaxf(Av[h * 2], Av[h * 2 + 1]);
    // ---- S^T[kt] = K[kt] Q^T (exact fp32); the other token tile's k in accumulator layout from its LDS tile
    f32x16 ako;
    {
      const float* ko = Kt + (h * 64 + 32 * (1 - tt) + li) * TF_VST + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 k4 = *reinterpret_cast<const float4*>(ko + 8 * c);
        ako[4 * c] = k4.x; ako[4 * c + 1] = k4.y; ako[4 * c + 2] = k4.z; ako[4 * c + 3] = k4.w;
      }
    }
    f32x16 s0, s1;                                        // keys 0..31, keys 32..63
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    if (tt == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[e], aq[e], s0, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ako[e], aq[e], s1, 0, 0, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ako[e], aq[e], s0, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[e], aq[e], s1, 0, 0, 0);
    }
    {
      // softmax over the 48 keys of this lane's query: keys 8 c + 4 hh + (0..3) of tile 0 (c < 4) and 32 + the same of tile 1 (c < 2)
      const float* bq = Bs + (h * T48_NT + qrow) * T48_BST + 4 * hh;
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 b4 = *reinterpret_cast<const float4*>(bq + 8 * c);
        s0[4 * c] += b4.x; s0[4 * c + 1] += b4.y; s0[4 * c + 2] += b4.z; s0[4 * c + 3] += b4.w;
        mx = fmaxf(fmaxf(mx, fmaxf(s0[4 * c], s0[4 * c + 1])), fmaxf(s0[4 * c + 2], s0[4 * c + 3]));
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float4 b4 = *reinterpret_cast<const float4*>(bq + 32 + 8 * c);
        s1[4 * c] += b4.x; s1[4 * c + 1] += b4.y; s1[4 * c + 2] += b4.z; s1[4 * c + 3] += b4.w;
        mx = fmaxf(fmaxf(mx, fmaxf(s1[4 * c], s1[4 * c + 1])), fmaxf(s1[4 * c + 2], s1[4 * c + 3]));
      }
      float m0, m1;
      tf_halves(mx, m0, m1);
      mx = fmaxf(m0, m1);
      float l = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s0[e] = expf(s0[e] - mx); l += s0[e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = expf(s1[e] - mx); l += s1[e]; }
      float l0, l1;
      tf_halves(l, l0, l1);
      const float il = 1.0f / (l0 + l1);
#pragma unroll
      for (int e = 0; e < 16; ++e) s0[e] *= il;
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] *= il;
    }
    // ---- O^T = V[0]^T P[0]^T + V[1]^T P[1]^T (keys 48..63 do not exist: eight steps of the second tile)
    f32x16 oT;
#pragma unroll
    for (int e = 0; e < 16; ++e) oT[e] = 0.f;
    {
      const float* v0 = Vt + h * 64 * TF_VST + li;
#pragma unroll
      for (int m = 0; m < 16; ++m) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[tf_key(m, hh) * TF_VST], s0[m], oT, 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 8; ++m) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[(32 + tf_key(m, hh)) * TF_VST], s1[m], oT, 0, 0, 0);
    }
    // the residual rows again (L2 hits)
    const float4 x0 = *reinterpret_cast<const float4*>(xb + lrow * fstride + 4 * lc4);
    float4 x1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lrow < 16) x1 = *reinterpret_cast<const float4*>(xb + (32 + lrow) * fstride + 4 * lc4);
    // ---- to_out, this head's 32 of the 128 reduction values (the fragments of W_out are fetched per sequence: L1 / L2 hits; held across
    // the products they would not leave room for the second score tile)
    const float so = scale_from_amax(amv);
    half8 oh[2], ol[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float t = oT[e] * so;
      const _Float16 th = (_Float16)t;
      oh[e >> 3][e & 7] = th;
      ol[e >> 3][e & 7] = (_Float16)(t - (float)th);
    }
    f32x16 y0, y1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { y0[e] = 0.f; y1[e] = 0.f; }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 woh[2], wol[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        // output channel 32 ct + li; reduction slot t of k-step s <-> feature 16 s + 8 (t >> 2) + 4 hh + (t & 3) of the head
        const int off = (32 * ct + li) * TF_HD + 32 * h + 16 * s + 4 * hh;
        const half4v a = *reinterpret_cast<const half4v*>(p.wo_hi + off), b = *reinterpret_cast<const half4v*>(p.wo_hi + off + 8);
        const half4v c = *reinterpret_cast<const half4v*>(p.wo_lo + off), d = *reinterpret_cast<const half4v*>(p.wo_lo + off + 8);
        woh[ct] = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        wol[ct] = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      y0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[0], ol[s], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[1], ol[s], y1, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol[0], oh[s], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol[1], oh[s], y1, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[0], oh[s], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[1], oh[s], y1, 0, 0, 0);
    }
    const float inv_o = 1.0f / (so * sw_o);
    __syncthreads();                                                            // (3) every wave is done with the k / v tiles
    if (tok) {
      float* yp = Yp + (h * T48_NT + trow) * TF_YST + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(yp + 8 * c) = make_float4(y0[4 * c] * inv_o, y0[4 * c + 1] * inv_o, y0[4 * c + 2] * inv_o, y0[4 * c + 3] * inv_o);
        *reinterpret_cast<float4*>(yp + 32 + 8 * c) = make_float4(y1[4 * c] * inv_o, y1[4 * c + 1] * inv_o, y1[4 * c + 2] * inv_o, y1[4 * c + 3] * inv_o);
      }
    }
    __syncthreads();                                                            // (4) partial tiles
    // ---- heads summed, residual added, rows stored (the threads that loaded a row finish it)
    float* yb = p.y + row0 * TF_C;
    {
      const int o = lrow * TF_YST + 4 * lc4;
      const float4 a = *reinterpret_cast<const float4*>(Yp + o), b2 = *reinterpret_cast<const float4*>(Yp + T48_NT * TF_YST + o);
      const float4 c = *reinterpret_cast<const float4*>(Yp + 2 * T48_NT * TF_YST + o), d = *reinterpret_cast<const float4*>(Yp + 3 * T48_NT * TF_YST + o);
      float4 r;
      r.x = ((a.x + b2.x) + (c.x + d.x)) + x0.x; r.y = ((a.y + b2.y) + (c.y + d.y)) + x0.y;
      r.z = ((a.z + b2.z) + (c.z + d.z)) + x0.z; r.w = ((a.w + b2.w) + (c.w + d.w)) + x0.w;
      *reinterpret_cast<float4*>(yb + lrow * fstride + 4 * lc4) = r;
      am = amax4(am, r);
    }
    if (lrow < 16) {
      const int o = (32 + lrow) * TF_YST + 4 * lc4;
      const float4 a = *reinterpret_cast<const float4*>(Yp + o), b2 = *reinterpret_cast<const float4*>(Yp + T48_NT * TF_YST + o);
      const float4 c = *reinterpret_cast<const float4*>(Yp + 2 * T48_NT * TF_YST + o), d = *reinterpret_cast<const float4*>(Yp + 3 * T48_NT * TF_YST + o);
      float4 r;
      r.x = ((a.x + b2.x) + (c.x + d.x)) + x1.x; r.y = ((a.y + b2.y) + (c.y + d.y)) + x1.y;
      r.z = ((a.z + b2.z) + (c.z + d.z)) + x1.z; r.w = ((a.w + b2.w) + (c.w + d.w)) + x1.w;
      *reinterpret_cast<float4*>(yb + (32 + lrow) * fstride + 4 * lc4) = r;
      am = amax4(am, r);
    }
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * 8 + wave);
  if (p.rec_v) wave_amax_emit(stv, p.rec_v, (int)blockIdx.x * 8 + wave);
}

static int t48_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

// called by wdno_tattn_fused_fwd (attn_fused.hip) for n_tok == 48
int wdno_tattn_fused_fwd48_launch(const TFusedP& p, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)tattn_fused_fwd48_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T48_LDS_BYTES) != hipSuccess) return WDNO_ELAUNCH;
    attr_done = true;
  }
  int64_t grid = t48_num_cus();
  if (grid > p.nseq) grid = p.nseq;
  tattn_fused_fwd48_kernel<<<(int)grid, 512, T48_LDS_BYTES, st>>>(p);
  return wdno_check_launch();
}
