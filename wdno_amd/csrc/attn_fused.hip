// attn_fused.hip -- the temporal attention block of the smoke U-Net (conv3d.py:165-174 LayerNorm, :277-353 Attention, wrapped as
// Residual(PreNorm(dim, EinopsToAndFrom('b c f h w', 'b (h w) f c', Attention)))) as ONE kernel at the network's first level:
//
//     y = x + W_out . softmax(rot(scale q) rot(k)^T + bias) v ,      (q | k | v) = W_qkv . LayerNorm(x)
//
// for C = 64 channels, 24 frames, 4 heads of 32. Un-fused, the block moves the [pixels x 384] qkv tensor (472 MB at the bench size)
// HBM -> HBM twice in the forward pass alone, plus the LayerNorm planes and the attention output planes; fused, a sequence (the 24
// frames of one pixel: 24 rows of 256 B, 409 600 B apart) is read once and written once.
//
// One block = four waves = the four heads of ONE sequence at a time; blocks walk sequences with a grid stride, two blocks per CU.
//   * LayerNorm by the whole block (16 lanes per row), result split into fp16 (hi, lo) planes in LDS [32 tokens][64] (rows 24..31 zero;
//     row pitch 144 B = conflict-free ds_read_b128 fragments). Plane scale from the bound |LN| <= sqrt(C) max|g| (as norm.hip does).
//   * Projection on v_mfma_f32_32x32x16_f16 with the three products hi*lo + lo*hi + hi*hi (the fp32-equivalent arithmetic of the
//     convolutions, conv_h3t.hip): A = the head's 96 rows of the packed W_qkv planes, held in registers for the whole kernel
//     (24 fragments), B = the token planes. Accumulators come out as [feature][token]: a lane owns one token and the features
//     8 c + 4 hh + (0..3).
//   * That layout IS the operand layout of the exact-fp32 score product (attention.hip: S^T = K Q^T on v_mfma_f32_32x32x2_f32, step e
//     contracts the features the two lane halves hold in accumulator register e), so q and k never move; rotary pairs (2i, 2i+1) are
//     lane-local. Softmax over the keys inside a lane pair; P^T feeds the second product in place; V goes through a 32 x 32 LDS tile
//     to be read transposed.
//   * to_out: the head's slice of the reduction (32 of 128) on the split MFMA again -- the accumulator registers of O^T are the
//     B fragments once W_out's fragments are gathered in the matching feature order -- partial [64][token] tiles of the four heads
//     summed through LDS together with the residual x (kept in registers by the lanes that loaded it), one coalesced store per row.
#include "attn_fused.h"

extern int wdno_debug_mode;


__global__ __launch_bounds__(256, 2) void tattn_fused_fwd_kernel(TFusedP p) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[32 * TF_AST];
  __shared__ __attribute__((aligned(16))) float Vt[TF_HEADS][32 * TF_VST];
  __shared__ __attribute__((aligned(16))) float Yp[TF_HEADS][TF_NT * TF_YST];
  __shared__ __attribute__((aligned(16))) float2 Rt[32 * TF_RST];
  __shared__ __attribute__((aligned(16))) float Bs[TF_HEADS][32 * TF_BST];
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const bool tok = li < TF_NT;
  const int lrow = tid >> 4, lc4 = tid & 15;

  // ---- per-kernel operands: the head's weight fragments, the lane's rotary entries, the LayerNorm gain
  half8 wqh[3][4], wql[3][4];
#pragma unroll
  for (int ti = 0; ti < 3; ++ti)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int off = (ti * TF_HD + h * 32 + li) * TF_C + 16 * s + 8 * hh;
      wqh[ti][s] = *reinterpret_cast<const half8*>(p.wq_hi + off);
      wql[ti][s] = *reinterpret_cast<const half8*>(p.wq_lo + off);
    }
  // to_out: output channel 32 ct + li; reduction slot t of k-step s <-> feature d = 16 s + 8 (t >> 2) + 4 hh + (t & 3) of the head (the
  // order in which accumulator registers 8 s .. 8 s + 7 of O^T hold them)
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
  // rotary table in LDS as (cos, sin) pairs: entry [token][pair i] -> (cos, sin) of features (2 i, 2 i + 1); rows 24..31 = identity
  for (int i = tid; i < 32 * 16; i += 256) {
    const int t = i >> 4, j = i & 15;
    float2 v = make_float2(1.f, 0.f);
    if (p.rcos && t < TF_NT) v = make_float2(p.rcos[t * 32 + 2 * j], p.rsin[t * 32 + 2 * j]);
    Rt[t * TF_RST + j] = v;
  }
  // relative-position bias [head][query][key] (zeros where absent / beyond the 24 tokens)
  for (int i = tid; i < TF_HEADS * 32 * TF_BST; i += 256) {
    const int hd = i / (32 * TF_BST), r = i - hd * (32 * TF_BST), q = r / TF_BST, k = r - q * TF_BST;
    Bs[hd][r] = (p.bias && q < TF_NT && k < TF_NT) ? p.bias[(hd * TF_NT + q) * TF_NT + k] : 0.f;
  }
  const float4 g4 = reinterpret_cast<const float4*>(p.gamma)[lc4];
  const float ps = scale_from_amax(8.0f * group_max<16>(amax4(0.f, g4)));        // |LayerNorm(x)| <= sqrt(64) max|g|
  const float inv_qkv = 1.0f / (ps * p.wq_scale[0]);
  const float sw_o = p.wo_scale[0];
  const int64_t fstride = (int64_t)p.HW * TF_C;
  float am = 0.f, stv = 0.f;

  // rows of the first sequence; afterwards the rows of sequence n + 1 are requested while sequence n is in the matrix pipes
  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int64_t r0) {
    const float* xr = p.x + r0 * TF_C;
    nx0 = *reinterpret_cast<const float4*>(xr + lrow * fstride + 4 * lc4);
    if (lrow < 8) nx1 = *reinterpret_cast<const float4*>(xr + (16 + lrow) * fstride + 4 * lc4);
  };
  // sequence -> (sample b, pixel): advanced incrementally (a 64-bit division per sequence is ~200 instructions of every wave)
  int nb = (int)(blockIdx.x / (unsigned)p.HW), npix = (int)(blockIdx.x - (unsigned)nb * (unsigned)p.HW);
  const int gstep_b = (int)(gridDim.x / (unsigned)p.HW), gstep_p = (int)(gridDim.x - (unsigned)gstep_b * (unsigned)p.HW);
  if ((int64_t)blockIdx.x < p.nseq) fetch((int64_t)nb * TF_NT * p.HW + npix);
  for (int64_t seq = blockIdx.x; seq < p.nseq; seq += gridDim.x) {
    const int64_t row0 = (int64_t)nb * TF_NT * p.HW + npix;      // row of frame 0; frame f at + f * HW
    const float* xb = p.x + row0 * TF_C;
    nb += gstep_b; npix += gstep_p;
    if (npix >= p.HW) { npix -= p.HW; ++nb; }
    // ---- rows -> LayerNorm -> planes (rows 24..31: zeros in, zeros out)
    tf_ln_row(nx0, g4, p.eps, ps, Ah, Al, lrow, lc4);
    tf_ln_row(nx1, g4, p.eps, ps, Ah, Al, 16 + lrow, lc4);
    __syncthreads();
    if (seq + gridDim.x < p.nseq) fetch((int64_t)nb * TF_NT * p.HW + npix);
    // ---- (q | k | v)^T of this head: [feature][token]
    f32x16 aq, ak, av;
#pragma unroll
    for (int e = 0; e < 16; ++e) { aq[e] = 0.f; ak[e] = 0.f; av[e] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 ah = *reinterpret_cast<const half8*>(Ah + li * TF_AST + 16 * s + 8 * hh);
      const half8 al = *reinterpret_cast<const half8*>(Al + li * TF_AST + 16 * s + 8 * hh);
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
    if (p.qkv_out && tok) {
      float* qr = p.qkv_out + (row0 + (int64_t)li * p.HW) * (3 * TF_HD) + h * 32 + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(qr + 8 * c) = make_float4(aq[4 * c], aq[4 * c + 1], aq[4 * c + 2], aq[4 * c + 3]);
        *reinterpret_cast<float4*>(qr + TF_HD + 8 * c) = make_float4(ak[4 * c], ak[4 * c + 1], ak[4 * c + 2], ak[4 * c + 3]);
        *reinterpret_cast<float4*>(qr + 2 * TF_HD + 8 * c) = make_float4(av[4 * c], av[4 * c + 1], av[4 * c + 2], av[4 * c + 3]);
      }
    }
    // V tile for the transposed read; max|v| bounds |out| (rows of P sum to 1)
    float amv = 0.f;
    float* vt = Vt[h];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 v4 = make_float4(av[4 * c], av[4 * c + 1], av[4 * c + 2], av[4 * c + 3]);
      amv = amax4(amv, v4);
      *reinterpret_cast<float4*>(vt + li * TF_VST + 8 * c + 4 * hh) = v4;
    }
    stv = fmaxf(stv, amv);
    amv = tf_wave_max(amv);
    // q * scale, rotary on q and k (pairs (2i, 2i + 1) = accumulator registers (2 j, 2 j + 1))
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // features 8 c + 4 hh + (0..3) = pairs 4 c + 2 hh, 4 c + 2 hh + 1
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
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < 3; ++c) {                     // keys 8 c + 4 hh + (0..3) < 24
        const float4 b4 = *reinterpret_cast<const float4*>(Bs[h] + li * TF_BST + 8 * c + 4 * hh);
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
    // the residual rows again (L2 hits; not held in registers across the products: the weight fragments own the register file)
    const float4 x0 = *reinterpret_cast<const float4*>(xb + lrow * fstride + 4 * lc4);
    float4 x1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lrow < 8) x1 = *reinterpret_cast<const float4*>(xb + (16 + lrow) * fstride + 4 * lc4);
    // ---- to_out, this head's 32 of the 128 reduction values: y_part[c][token]
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
      y0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[0][s], ol[s], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[1][s], ol[s], y1, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol[0][s], oh[s], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol[1][s], oh[s], y1, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[0][s], oh[s], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[1][s], oh[s], y1, 0, 0, 0);
    }
    const float inv_o = 1.0f / (so * sw_o);
    if (tok) {
      float* yp = Yp[h] + li * TF_YST + 4 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(yp + 8 * c) = make_float4(y0[4 * c] * inv_o, y0[4 * c + 1] * inv_o, y0[4 * c + 2] * inv_o, y0[4 * c + 3] * inv_o);
        *reinterpret_cast<float4*>(yp + 32 + 8 * c) = make_float4(y1[4 * c] * inv_o, y1[4 * c + 1] * inv_o, y1[4 * c + 2] * inv_o, y1[4 * c + 3] * inv_o);
      }
    }
    __syncthreads();
    // ---- heads summed, residual added, rows stored (the lanes that loaded a row finish it)
    float* yb = p.y + row0 * TF_C;
    {
      const int o = lrow * TF_YST + 4 * lc4;
      const float4 a = *reinterpret_cast<const float4*>(Yp[0] + o), b2 = *reinterpret_cast<const float4*>(Yp[1] + o);
      const float4 c = *reinterpret_cast<const float4*>(Yp[2] + o), d = *reinterpret_cast<const float4*>(Yp[3] + o);
      float4 r;
      r.x = ((a.x + b2.x) + (c.x + d.x)) + x0.x; r.y = ((a.y + b2.y) + (c.y + d.y)) + x0.y;
      r.z = ((a.z + b2.z) + (c.z + d.z)) + x0.z; r.w = ((a.w + b2.w) + (c.w + d.w)) + x0.w;
      *reinterpret_cast<float4*>(yb + lrow * fstride + 4 * lc4) = r;
      am = amax4(am, r);
    }
    if (lrow < 8) {
      const int o = (16 + lrow) * TF_YST + 4 * lc4;
      const float4 a = *reinterpret_cast<const float4*>(Yp[0] + o), b2 = *reinterpret_cast<const float4*>(Yp[1] + o);
      const float4 c = *reinterpret_cast<const float4*>(Yp[2] + o), d = *reinterpret_cast<const float4*>(Yp[3] + o);
      float4 r;
      r.x = ((a.x + b2.x) + (c.x + d.x)) + x1.x; r.y = ((a.y + b2.y) + (c.y + d.y)) + x1.y;
      r.z = ((a.z + b2.z) + (c.z + d.z)) + x1.z; r.w = ((a.w + b2.w) + (c.w + d.w)) + x1.w;
      *reinterpret_cast<float4*>(yb + (16 + lrow) * fstride + 4 * lc4) = r;
      am = amax4(am, r);
    }
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * TF_HEADS + h);
  if (p.rec_v) wave_amax_emit(stv, p.rec_v, (int)blockIdx.x * TF_HEADS + h);
}

static int tf_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

// 24 frames: this file (and attn_fused_bwd.hip for the gradients); 48 frames: attn_fused48.hip, forward only
int wdno_tattn_fused_fwd48_launch(const TFusedP& p, hipStream_t st);
// 128 / 256 channels, 24 frames: attn_fused_wide.hip, forward only
int wdno_tattn_fused_fwd_wide_launch(const TFusedP& p, int C, hipStream_t st);
extern "C" int wdno_tattn_fused_takes(int C, int n_tok, int heads) {
  if (heads != TF_HEADS) return 0;
  if (C == TF_C) return n_tok == TF_NT || n_tok == 48;
  return (C == 128 || C == 256) && n_tok == TF_NT && wdno_debug_mode != 62;
}

extern "C" int wdno_tattn_fused_fwd(const float* x, const float* gamma, float eps, const void* wq_hi, const void* wq_lo, const float* wq_scale,
                                    const void* wo_hi, const void* wo_lo, const float* wo_scale, const float* rot_cos, const float* rot_sin,
                                    const float* bias, float* y, float* amax_rec, float* qkv_out, float* rec_v,
                                    int64_t n_batch, int n_tok, int64_t hw, int C, int heads, float scale, wdno_stream_t s) {
  WDNO_REQUIRE(x && gamma && wq_hi && wq_lo && wq_scale && wo_hi && wo_lo && wo_scale && y && n_batch > 0 && hw > 0);
  WDNO_REQUIRE((rot_cos == nullptr) == (rot_sin == nullptr));
  if (!wdno_tattn_fused_takes(C, n_tok, heads) || hw > 0x7fffffff / (C * 48)) return WDNO_EUNSUPPORTED;
  TFusedP p;
  p.x = x; p.gamma = gamma; p.eps = eps;
  p.wq_hi = (const _Float16*)wq_hi; p.wq_lo = (const _Float16*)wq_lo; p.wq_scale = wq_scale;
  p.wo_hi = (const _Float16*)wo_hi; p.wo_lo = (const _Float16*)wo_lo; p.wo_scale = wo_scale;
  p.rcos = rot_cos; p.rsin = rot_sin; p.bias = bias;
  p.y = y; p.amax_rec = amax_rec; p.qkv_out = qkv_out;
  p.rec_v = rec_v;
  p.HW = (int)hw; p.scale = scale; p.nseq = n_batch * hw;
  if (C != TF_C) {
    if (qkv_out) return WDNO_EUNSUPPORTED;            // (nothing is kept for a backward: these levels train layer by layer; rec_v stays untouched)
    return wdno_tattn_fused_fwd_wide_launch(p, C, as_stream(s));
  }
  if (n_tok == 48) {
    if (qkv_out) return WDNO_EUNSUPPORTED;          // (only the un-fused backward of the 24-frame block asks for the projections)
    return wdno_tattn_fused_fwd48_launch(p, as_stream(s));
  }
  int64_t grid = 2 * (int64_t)tf_num_cus();
  if (grid > p.nseq) grid = p.nseq;
  tattn_fused_fwd_kernel<<<(int)grid, 256, 0, as_stream(s)>>>(p);
  return wdno_check_launch();
}
