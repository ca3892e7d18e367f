// attention.hip -- attention kernels of the two U-Nets (head dim fixed at 32, as in the reference).
//
//  * softmax attention over short token axes (temporal: 24/48 frames with rotary + relative-position bias;
//    spatial mid-block: 64/100/400 tokens): one wavefront per (unit, head), K/V staged in LDS, rows streamed.
//    HBM-bound (no reuse at n = 24); QK^T / PV are far too small to feed MFMA tiles.
//  * linear attention (softmax over the 32 head channels for q, over all tokens for k): the 32x32 context
//    k^T v is a genuine dense contraction over thousands of tokens -> v_mfma_f32_32x32x2_f32, streamed from
//    global memory (each lane supplies one k and one v element per MFMA); the per-token 32x32 mat-vecs run
//    on the vector ALU with the context broadcast from LDS.
#include "common.h"
extern int wdno_debug_mode;

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DH 32
#define KST 33

// ================================================================================================ softmax attention
// One THREAD per (unit, head, query row). The rotated K rows of an item live in LDS (every thread of the item reads the
// same row -> broadcast ds_read_b128, no conflicts); V rows and dO rows are read straight from global memory as
// wave-broadcast float4 loads (one cache line per item per load). Scores are recomputed in a second pass instead of
// being stored (n is 24..400), so the forward needs no per-thread arrays beyond q[32] / out[32].
struct AttnP {
  wdno_attn_desc d;
  float scale;
  int RW;   // 3*heads*32
  int HD;   // heads*32
  int ipb;  // items (unit, head) per block
  int kst;  // LDS floats per item for one [n][32] matrix (padded)
  int64_t n_items;
  float* amax_rec;   // optional amax record (common.h) of the tensor the kernel writes: out (forward), dqkv (backward)
  // backward with dqkv delivered as fp16 (hi, lo) planes (wdno_attn_bwd_planes): the plane pointers, the amax records of qkv and
  // dout the scale bound is derived from, and where the scale is left
  _Float16* pl_hi; _Float16* pl_lo; const float* rec_qkv; const float* rec_dout; float* pl_scale;
};
// dqkv as planes: store 4 consecutive values of row `row` (element offset inside the [rows][RW] tensor)
__device__ __forceinline__ void am_store4_planes(_Float16* hi, _Float16* lo, int64_t off, float4 v, float s) {
  typedef _Float16 half4v __attribute__((ext_vector_type(4)));
  const float t[4] = {v.x, v.y, v.z, v.w};
  half4v h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) { _Float16 th, tl; plane_pack(t[j], s, lo == nullptr, th, tl); h[j] = th; l[j] = tl; }
  *reinterpret_cast<half4v*>(hi + off) = h;
  if (lo) *reinterpret_cast<half4v*>(lo + off) = l;
}
// The MFMA accumulator layout gives lane (li, hh) the channel runs 8 e4 + 4 hh + (0..3) of its row: 8-byte plane stores. One
// v_permlane32_swap per component trades the odd run of the low half-wave for the even run of the high one (tools/probes/swap_probe.hip),
// after which a lane owns 8 CONSECUTIVE channels (16 j + 8 hh + 0..7) of each pair j: 16-byte stores. v[e4] = the four runs; off0 = element
// offset of channel 0 of this head in the lane's row. Both lanes of a pair (l, l + 32) must be active.
__device__ __forceinline__ void am_store_row_planes(_Float16* hi, _Float16* lo, int64_t off0, int hh, const float4* v, float s) {
  typedef _Float16 half8v __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float a[4] = {v[2 * j].x, v[2 * j].y, v[2 * j].z, v[2 * j].w}, b[4] = {v[2 * j + 1].x, v[2 * j + 1].y, v[2 * j + 1].z, v[2 * j + 1].w};
    float t[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c]), __float_as_uint(b[c]), false, false);
      t[c] = __uint_as_float(r[0]); t[4 + c] = __uint_as_float(r[1]);
    }
    half8v h, l;
#pragma unroll
    for (int c = 0; c < 8; ++c) { _Float16 th, tl; plane_pack(t[c], s, lo == nullptr, th, tl); h[c] = th; l[c] = tl; }
    const int64_t off = off0 + 16 * j + 8 * hh;
    *reinterpret_cast<half8v*>(hi + off) = h;
    if (lo) *reinterpret_cast<half8v*>(lo + off) = l;
  }
}

#define ATT_THREADS 256
#define ATT_BWD_THREADS 128

__device__ __forceinline__ void load_row32(const float* __restrict__ p, float* v) {
  const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) { float4 t = q[e]; v[4 * e] = t.x; v[4 * e + 1] = t.y; v[4 * e + 2] = t.z; v[4 * e + 3] = t.w; }
}
__device__ __forceinline__ void store_row32(float* __restrict__ p, const float* v) {
  float4* q = reinterpret_cast<float4*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
}
// interleaved-pair rotary: (x0, x1) -> (x0 c - x1 s, x1 c + x0 s); inverse = transpose
__device__ __forceinline__ void rotate32(float* v, const float* __restrict__ cs, const float* __restrict__ sn, bool inverse) {
  float c[DH], s[DH];
  load_row32(cs, c);
  load_row32(sn, s);
#pragma unroll
  for (int e = 0; e < DH; e += 2) {
    float x0 = v[e], x1 = v[e + 1];
    float s0 = inverse ? -s[e] : s[e], s1 = inverse ? -s[e + 1] : s[e + 1];
    v[e] = x0 * c[e] - x1 * s0;
    v[e + 1] = x1 * c[e + 1] + x0 * s1;
  }
}
__device__ __forceinline__ float dot32_lds(const float* q, const float* __restrict__ krow) {
  const float4* k4 = reinterpret_cast<const float4*>(krow);
  float a = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float4 t = k4[e];
    a = fmaf(q[4 * e], t.x, a); a = fmaf(q[4 * e + 1], t.y, a); a = fmaf(q[4 * e + 2], t.z, a); a = fmaf(q[4 * e + 3], t.w, a);
  }
  return a;
}

__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                                const float* __restrict__ rsin, const float* __restrict__ bias,
                                                                float* __restrict__ out, AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = p.d.n_tok;
  const int il = threadIdx.x / n;                 // item slot inside the block (valid when < ipb)
  const int i0 = threadIdx.x - il * n;            // first row of this thread
  const int64_t item = (int64_t)blockIdx.x * p.ipb + il;
  const bool active = il < p.ipb && item < p.n_items;
  const int h = active ? (int)(item % p.d.heads) : 0;
  const int64_t unit = active ? item / p.d.heads : 0;
  const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
  const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
  float* Ks = smem + (active ? il : 0) * p.kst;
  const int rstep = p.ipb > 1 ? n : ATT_THREADS;  // ipb == 1: threads stride over rows
  if (active) {
    for (int j = i0; j < n; j += rstep) {
      float k[DH];
      load_row32(qkv + (row0 + (int64_t)j * p.d.st) * p.RW + p.HD + h * DH, k);
      if (rcos) rotate32(k, rcos + j * DH, rsin + j * DH, false);
      store_row32(Ks + j * DH, k);
    }
  }
  __syncthreads();
  if (!active) return;
  for (int i = i0; i < n; i += rstep) {
    float q[DH];
    load_row32(qkv + (row0 + (int64_t)i * p.d.st) * p.RW + h * DH, q);
#pragma unroll
    for (int e = 0; e < DH; ++e) q[e] *= p.scale;
    if (rcos) rotate32(q, rcos + i * DH, rsin + i * DH, false);
    const float* brow = bias ? bias + ((int64_t)h * n + i) * n : nullptr;
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < n; ++j) {
      float sv = dot32_lds(q, Ks + j * DH);
      if (brow) sv += brow[j];
      float mn = fmaxf(m, sv);
      l = l * expf(m - mn) + expf(sv - mn);
      m = mn;
    }
    const float inv = 1.0f / l;
    float o[DH];
#pragma unroll
    for (int e = 0; e < DH; ++e) o[e] = 0.f;
    for (int j = 0; j < n; ++j) {
      float sv = dot32_lds(q, Ks + j * DH);
      if (brow) sv += brow[j];
      float pj = expf(sv - m) * inv;
      const float4* v4 = reinterpret_cast<const float4*>(qkv + (row0 + (int64_t)j * p.d.st) * p.RW + 2 * p.HD + h * DH);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float4 t = v4[e];
        o[4 * e] = fmaf(pj, t.x, o[4 * e]); o[4 * e + 1] = fmaf(pj, t.y, o[4 * e + 1]);
        o[4 * e + 2] = fmaf(pj, t.z, o[4 * e + 2]); o[4 * e + 3] = fmaf(pj, t.w, o[4 * e + 3]);
      }
    }
    store_row32(out + (row0 + (int64_t)i * p.d.st) * p.HD + h * DH, o);
  }
}

// backward. Phase A (thread = query row i): p_ij, dS_ij, dQ_i ; phase B (thread = key row j): dK_j, dV_j from the P / dS
// matrices and the rotated Q rows left in LDS. delta_i = <dO_i, O_i> uses the saved forward output.
__global__ __launch_bounds__(ATT_BWD_THREADS) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                                    const float* __restrict__ rsin, const float* __restrict__ bias,
                                                                    const float* __restrict__ fout, const float* __restrict__ dout,
                                                                    float* __restrict__ dqkv, float* __restrict__ dbias, AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = p.d.n_tok, n1 = n + 1;
  const int item_lds = 2 * p.kst + 2 * n * n1;       // K, Q, P, dS per item
  float* dBs = smem + p.ipb * item_lds;              // [heads][n][n] when dbias
  const int il = threadIdx.x / n;
  const int i = threadIdx.x - il * n;
  const bool slot_ok = il < p.ipb;
  float* Ks = smem + (slot_ok ? il : 0) * item_lds;
  float* Qs = Ks + p.kst;
  float* Ps = Qs + p.kst;
  float* Ss = Ps + n * n1;
  if (dbias)
    for (int e = threadIdx.x; e < p.d.heads * n * n; e += ATT_BWD_THREADS) dBs[e] = 0.f;
  const int64_t ngroups = (p.n_items + p.ipb - 1) / p.ipb;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t item = grp * p.ipb + il;
    const bool active = slot_ok && item < p.n_items;
    const int h = active ? (int)(item % p.d.heads) : 0;
    const int64_t unit = active ? item / p.d.heads : 0;
    const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
    const int64_t row = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si + (int64_t)i * p.d.st;
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    float q[DH];
    __syncthreads();                                  // previous group's phase B is done with the LDS matrices
    if (active) {
      float k[DH];
      load_row32(qkv + row * p.RW + p.HD + h * DH, k);
      load_row32(qkv + row * p.RW + h * DH, q);
#pragma unroll
      for (int e = 0; e < DH; ++e) q[e] *= p.scale;
      if (rcos) {
        rotate32(k, rcos + i * DH, rsin + i * DH, false);
        rotate32(q, rcos + i * DH, rsin + i * DH, false);
      }
      store_row32(Ks + i * DH, k);
      store_row32(Qs + i * DH, q);
    }
    __syncthreads();
    if (active) {
      float go[DH], dq[DH];
      load_row32(dout + row * p.HD + h * DH, go);
      float delta = 0.f;
      {
        const float4* o4 = reinterpret_cast<const float4*>(fout + row * p.HD + h * DH);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float4 t = o4[e];
          delta = fmaf(go[4 * e], t.x, delta); delta = fmaf(go[4 * e + 1], t.y, delta);
          delta = fmaf(go[4 * e + 2], t.z, delta); delta = fmaf(go[4 * e + 3], t.w, delta);
        }
      }
      const float* brow = bias ? bias + ((int64_t)h * n + i) * n : nullptr;
      float m = -INFINITY, l = 0.f;
      for (int j = 0; j < n; ++j) {
        float sv = dot32_lds(q, Ks + j * DH);
        if (brow) sv += brow[j];
        float mn = fmaxf(m, sv);
        l = l * expf(m - mn) + expf(sv - mn);
        m = mn;
      }
      const float inv = 1.0f / l;
#pragma unroll
      for (int e = 0; e < DH; ++e) dq[e] = 0.f;
      for (int j = 0; j < n; ++j) {
        const float4* k4 = reinterpret_cast<const float4*>(Ks + j * DH);
        const float4* v4 = reinterpret_cast<const float4*>(qkv + (row0 + (int64_t)j * p.d.st) * p.RW + 2 * p.HD + h * DH);
        float sv = 0.f, dp = 0.f;
        float4 kk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          kk[e] = k4[e];
          float4 t = v4[e];
          sv = fmaf(q[4 * e], kk[e].x, sv); sv = fmaf(q[4 * e + 1], kk[e].y, sv); sv = fmaf(q[4 * e + 2], kk[e].z, sv); sv = fmaf(q[4 * e + 3], kk[e].w, sv);
          dp = fmaf(go[4 * e], t.x, dp); dp = fmaf(go[4 * e + 1], t.y, dp); dp = fmaf(go[4 * e + 2], t.z, dp); dp = fmaf(go[4 * e + 3], t.w, dp);
        }
        if (brow) sv += brow[j];
        float pj = expf(sv - m) * inv;
        float ds = pj * (dp - delta);
        Ps[i * n1 + j] = pj;
        Ss[i * n1 + j] = ds;
        if (dbias) atomicAdd(&dBs[(h * n + i) * n + j], ds);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dq[4 * e] = fmaf(ds, kk[e].x, dq[4 * e]); dq[4 * e + 1] = fmaf(ds, kk[e].y, dq[4 * e + 1]);
          dq[4 * e + 2] = fmaf(ds, kk[e].z, dq[4 * e + 2]); dq[4 * e + 3] = fmaf(ds, kk[e].w, dq[4 * e + 3]);
        }
      }
      if (rcos) rotate32(dq, rcos + i * DH, rsin + i * DH, true);
#pragma unroll
      for (int e = 0; e < DH; ++e) dq[e] *= p.scale;
      store_row32(dqkv + row * p.RW + h * DH, dq);
    }
    __syncthreads();
    if (active) {   // phase B: this thread owns key/value row j = i
      float dk[DH], dv[DH];
#pragma unroll
      for (int e = 0; e < DH; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
      for (int r = 0; r < n; ++r) {
        float ds = Ss[r * n1 + i], pj = Ps[r * n1 + i];
        const float4* q4 = reinterpret_cast<const float4*>(Qs + r * DH);
        const float4* g4 = reinterpret_cast<const float4*>(dout + (row0 + (int64_t)r * p.d.st) * p.HD + h * DH);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float4 a = q4[e], b = g4[e];
          dk[4 * e] = fmaf(ds, a.x, dk[4 * e]); dk[4 * e + 1] = fmaf(ds, a.y, dk[4 * e + 1]);
          dk[4 * e + 2] = fmaf(ds, a.z, dk[4 * e + 2]); dk[4 * e + 3] = fmaf(ds, a.w, dk[4 * e + 3]);
          dv[4 * e] = fmaf(pj, b.x, dv[4 * e]); dv[4 * e + 1] = fmaf(pj, b.y, dv[4 * e + 1]);
          dv[4 * e + 2] = fmaf(pj, b.z, dv[4 * e + 2]); dv[4 * e + 3] = fmaf(pj, b.w, dv[4 * e + 3]);
        }
      }
      if (rcos) rotate32(dk, rcos + i * DH, rsin + i * DH, true);
      store_row32(dqkv + row * p.RW + p.HD + h * DH, dk);
      store_row32(dqkv + row * p.RW + 2 * p.HD + h * DH, dv);
    }
  }
  if (dbias) {         // dbias: this block's partial [heads][n][n] (plain stores; attn_dbias_reduce_kernel sums the blocks in index order)
    __syncthreads();
    float* part = dbias + (size_t)blockIdx.x * p.d.heads * n * n;
    for (int e = threadIdx.x; e < p.d.heads * n * n; e += ATT_BWD_THREADS) part[e] = dBs[e];
  }
}

// Backward over MANY tokens (129 .. 576; no rotation, no bias): the mid spatial attention of a model applied to a finer grid than it
// was built for (the space super-resolution model on 80 x 80 tensors attends over 20 x 20 = 400 tokens; the reference trains it on
// <= 40 x 40, data_2d.py:182-183, but nothing in its code stops a fine-tune at the sampling size). The n x n matrices P and dS do not
// fit LDS any more, so nothing is stored: one block per (unit, head) item;
//   phase A, thread = query row i: K and V rows in LDS; softmax statistics (m_i, 1 / l_i), delta_i = <dO_i, O_i>, dQ_i;
//   phase B, thread = key row j (k_j, v_j in registers): the rows q_i, dO_i stream from global memory as wave-broadcast loads (every
//   thread reads the same row: one line per load) and P_ij / dS_ij are recomputed from the statistics phase A left in LDS.
#define ATT_BIG_THREADS 256
#define ATT_BIG_MAXTOK 576
__global__ __launch_bounds__(ATT_BIG_THREADS) void attn_bwd_big_kernel(const float* __restrict__ qkv, const float* __restrict__ fout,
                                                                        const float* __restrict__ dout, float* __restrict__ dqkv, AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = p.d.n_tok;
  float* Ks = smem;                                   // [n][32]
  float* Vs = Ks + n * DH;                            // [n][32]
  float* Ms = Vs + n * DH;                            // [n] row maximum
  float* Ls = Ms + n;                                 // [n] 1 / row sum
  float* Dl = Ls + n;                                 // [n] delta
  for (int64_t item = blockIdx.x; item < p.n_items; item += gridDim.x) {
    const int h = (int)(item % p.d.heads);
    const int64_t unit = item / p.d.heads;
    const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    __syncthreads();                                  // the previous item's phase B has finished with the statistics
    for (int j = threadIdx.x; j < n; j += ATT_BIG_THREADS) {
      float k[DH], v[DH];
      const int64_t row = row0 + (int64_t)j * p.d.st;
      load_row32(qkv + row * p.RW + p.HD + h * DH, k);
      load_row32(qkv + row * p.RW + 2 * p.HD + h * DH, v);
      store_row32(Ks + j * DH, k);
      store_row32(Vs + j * DH, v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += ATT_BIG_THREADS) {
      const int64_t row = row0 + (int64_t)i * p.d.st;
      float q[DH], go[DH], dq[DH];
      load_row32(qkv + row * p.RW + h * DH, q);
#pragma unroll
      for (int e = 0; e < DH; ++e) q[e] *= p.scale;
      load_row32(dout + row * p.HD + h * DH, go);
      float delta = 0.f;
      {
        const float4* o4 = reinterpret_cast<const float4*>(fout + row * p.HD + h * DH);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float4 t = o4[e];
          delta = fmaf(go[4 * e], t.x, delta); delta = fmaf(go[4 * e + 1], t.y, delta);
          delta = fmaf(go[4 * e + 2], t.z, delta); delta = fmaf(go[4 * e + 3], t.w, delta);
        }
      }
      float m = -INFINITY, l = 0.f;
      for (int j = 0; j < n; ++j) {
        const float sv = dot32_lds(q, Ks + j * DH);
        const float mn = fmaxf(m, sv);
        l = l * expf(m - mn) + expf(sv - mn);
        m = mn;
      }
      const float inv = 1.0f / l;
#pragma unroll
      for (int e = 0; e < DH; ++e) dq[e] = 0.f;
      for (int j = 0; j < n; ++j) {
        const float4* k4 = reinterpret_cast<const float4*>(Ks + j * DH);
        const float sv = dot32_lds(q, Ks + j * DH);
        const float dp = dot32_lds(go, Vs + j * DH);
        const float ds = expf(sv - m) * inv * (dp - delta);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float4 kk = k4[e];
          dq[4 * e] = fmaf(ds, kk.x, dq[4 * e]); dq[4 * e + 1] = fmaf(ds, kk.y, dq[4 * e + 1]);
          dq[4 * e + 2] = fmaf(ds, kk.z, dq[4 * e + 2]); dq[4 * e + 3] = fmaf(ds, kk.w, dq[4 * e + 3]);
        }
      }
#pragma unroll
      for (int e = 0; e < DH; ++e) dq[e] *= p.scale;
      store_row32(dqkv + row * p.RW + h * DH, dq);
      Ms[i] = m; Ls[i] = inv; Dl[i] = delta;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += ATT_BIG_THREADS) {
      float k[DH], v[DH], dk[DH], dv[DH];
      load_row32(Ks + j * DH, k);
      load_row32(Vs + j * DH, v);
#pragma unroll
      for (int e = 0; e < DH; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
      for (int i = 0; i < n; ++i) {
        const int64_t row = row0 + (int64_t)i * p.d.st;
        const float4* q4 = reinterpret_cast<const float4*>(qkv + row * p.RW + h * DH);
        const float4* g4 = reinterpret_cast<const float4*>(dout + row * p.HD + h * DH);
        float4 qq[8], gg[8];
        float sv = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float4 a = q4[e];
          a.x *= p.scale; a.y *= p.scale; a.z *= p.scale; a.w *= p.scale;      // the same scaled q row phase A and the forward use
          qq[e] = a; gg[e] = g4[e];
          sv = fmaf(a.x, k[4 * e], sv); sv = fmaf(a.y, k[4 * e + 1], sv); sv = fmaf(a.z, k[4 * e + 2], sv); sv = fmaf(a.w, k[4 * e + 3], sv);
          dp = fmaf(gg[e].x, v[4 * e], dp); dp = fmaf(gg[e].y, v[4 * e + 1], dp); dp = fmaf(gg[e].z, v[4 * e + 2], dp); dp = fmaf(gg[e].w, v[4 * e + 3], dp);
        }
        const float pj = expf(sv - Ms[i]) * Ls[i];
        const float ds = pj * (dp - Dl[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dk[4 * e] = fmaf(ds, qq[e].x, dk[4 * e]); dk[4 * e + 1] = fmaf(ds, qq[e].y, dk[4 * e + 1]);
          dk[4 * e + 2] = fmaf(ds, qq[e].z, dk[4 * e + 2]); dk[4 * e + 3] = fmaf(ds, qq[e].w, dk[4 * e + 3]);
          dv[4 * e] = fmaf(pj, gg[e].x, dv[4 * e]); dv[4 * e + 1] = fmaf(pj, gg[e].y, dv[4 * e + 1]);
          dv[4 * e + 2] = fmaf(pj, gg[e].z, dv[4 * e + 2]); dv[4 * e + 3] = fmaf(pj, gg[e].w, dv[4 * e + 3]);
        }
      }
      const int64_t row = row0 + (int64_t)j * p.d.st;
      store_row32(dqkv + row * p.RW + p.HD + h * DH, dk);
      store_row32(dqkv + row * p.RW + 2 * p.HD + h * DH, dv);
    }
  }
}

// Sum of the per-block relative-position-bias gradient partials in block order: E = heads * n * n outputs, each the sum of nb terms.
// (Global float atomics summed them in arrival order: the gradient of relative_attention_bias differed from run to run in its last
// bits, and with it every bit-reproducibility check of a training step.) 64 outputs per block, four partial sums per output.
#define DBR_EL 8                     // outputs per block
#define DBR_CH (256 / DBR_EL)        // partial chains per output
__global__ __launch_bounds__(256) void attn_dbias_reduce_kernel(const float* __restrict__ part, int nb, float* __restrict__ dbias, int E) {
  // 8 outputs per block (E / 8 = 288 blocks for 4 heads x 24 x 24), 32 chains per output with four independent sums each: the 768
  // partials of an output are 6 rounds of loads per thread (64 outputs per block and 4 chains were 192 dependent rounds on 36 blocks:
  // 50 us per launch). The order of every sum is fixed by the indices alone.
  __shared__ float red[DBR_CH][DBR_EL];
  const int el = threadIdx.x & (DBR_EL - 1), q = threadIdx.x / DBR_EL;
  const int e = blockIdx.x * DBR_EL + el;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < E) {
    int b = q;
    for (; b + 3 * DBR_CH < nb; b += 4 * DBR_CH) {
      a0 += part[(size_t)b * E + e]; a1 += part[(size_t)(b + DBR_CH) * E + e];
      a2 += part[(size_t)(b + 2 * DBR_CH) * E + e]; a3 += part[(size_t)(b + 3 * DBR_CH) * E + e];
    }
    for (; b < nb; b += DBR_CH) a0 += part[(size_t)b * E + e];
  }
  red[q][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0 && e < E) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < DBR_CH; ++k) t += red[k][el];
    dbias[e] = t;
  }
}

// ------------------------------------------------------------------------------------------------ n_tok <= 32: one wave per item
// With at most 32 tokens the score matrix is ONE 32x32 MFMA tile (v_mfma_f32_32x32x2_f32: exact fp32 products and sums).
// Everything is kept transposed so that the softmax axis lies inside a lane:
//   S^T[j][i] = sum_d K[j][d] Q[i][d]     row operand K (lane = key j), column operand Q (lane = query i)
//   -> accumulator: lane (i, hh) holds keys j = 8*(e>>2) + 4*hh + (e&3), e = 0..15: max / sum over e and the partner lane.
//   O^T[d][i] = sum_j V[j][d] P[i][j]     the reduction index runs over the keys in exactly the order the accumulator holds
//   them (step m <-> e = m), so P^T feeds the second MFMA without moving data; V is read as coalesced 128-byte rows.
// The backward needs P and dS with the roles of the lanes swapped (lane = key) for dK / dV: two 32x33 tiles per wave in LDS.
#define AM_WAVES 4
#ifndef ATT_BWD_BLOCKS_PER_CU
#define ATT_BWD_BLOCKS_PER_CU 3      /* 4 (128 registers, 22 spilled): 457 / 139 / 70 us vs 458 / 126 / 75 at hw = 1600 / 400 / 100: no gain */
#endif
// wave-uniform pointer in SGPRs: the per-lane part of an address stays a 32-bit offset (global_load saddr + voffset)
__device__ __forceinline__ const float* am_uniform(const float* p) {
  uint64_t a = reinterpret_cast<uint64_t>(p);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int am_key(int m, int hh) { return 8 * (m >> 2) + 4 * hh + (m & 3); }

// Rows of one operand ([n][32] floats: q, k, v or dO of one head) -> LDS tile [32][AM_TS], loaded cooperatively (8 lanes
// per 128-byte row, three passes for 24 rows), optionally scaled and rotated on the way; rows >= n are zero-filled.
#define AM_TS 36
template <int NIF = 4>
__device__ __forceinline__ void am_stage_rows(float* __restrict__ tile, const float* __restrict__ ubase, unsigned row_stride,
                                              const float* __restrict__ rc, const float* __restrict__ rs, float scale, int n, int lane) {
  // ubase: wave-uniform pointer to row 0 of the operand (am_uniform); the per-lane part of every address is a 32-bit offset,
  // so nothing 64-bit per lane has to stay live across the item loop
  // NIF of the four 1-KiB row groups are requested together (4 KB in flight per wave; one at a time, sixteen waves per CU keep only
  // ~4 MB in flight chip-wide and the kernel sits at a latency-bound 2.9 TB/s); the rotation tables are fetched pass by pass.
  // NIF = 2 halves the staging registers where the caller is short of them.
#pragma unroll
  for (int k0 = 0; k0 < 4; k0 += NIF) {
    float4 xs[NIF];
#pragma unroll
    for (int k = 0; k < NIF; ++k) {
      const int idx = lane + 64 * (k0 + k);
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      xs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < n) xs[k] = *reinterpret_cast<const float4*>(ubase + ((unsigned)r * row_stride + (unsigned)c4));
    }
#pragma unroll
    for (int k = 0; k < NIF; ++k) {
      const int idx = lane + 64 * (k0 + k);
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      float4 x = xs[k];
      x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale;
      if (rc && r < n) {
        const float4 c = *reinterpret_cast<const float4*>(rc + r * DH + c4), sn = *reinterpret_cast<const float4*>(rs + r * DH + c4);
        x = make_float4(x.x * c.x - x.y * sn.x, x.y * c.y + x.x * sn.y, x.z * c.z - x.w * sn.z, x.w * c.w + x.z * sn.w);
      }
      *reinterpret_cast<float4*>(tile + r * AM_TS + c4) = x;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// the 16 values of row `li` this lane feeds to the MFMA steps of a product that contracts over the 32 channels: step m takes the
// channel pair (m, 16 + m), i.e. lane half hh holds channels d = 16 hh + m -- 16 CONSECUTIVE floats, four conflict-free ds_read_b128
// (any pairing of channels into steps is valid as long as both operands use it; the first version, d = 2 m + hh, needed sixteen
// ds_read_b32 at a row stride of 36 floats = 4-way bank conflicts, a fifth of the temporal-attention kernels' LDS cycles)
#define AM_D(m, hh) (16 * (hh) + (m))
__device__ __forceinline__ void am_sel(const float* __restrict__ tile, int li, int hh, float* sel) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(tile + li * AM_TS + 16 * hh + 4 * q);
    sel[4 * q] = v.x; sel[4 * q + 1] = v.y; sel[4 * q + 2] = v.z; sel[4 * q + 3] = v.w;
  }
}
// four of them (steps 4 g .. 4 g + 3)
__device__ __forceinline__ void am_sel4(const float* __restrict__ tile, int li, int hh, int g, float* sel) {
  const float4 v = *reinterpret_cast<const float4*>(tile + li * AM_TS + 16 * hh + 4 * g);
  sel[0] = v.x; sel[1] = v.y; sel[2] = v.z; sel[3] = v.w;
}
// softmax over the keys of query `li` from the transposed score tile (masked beyond n); returns P^T in place
__device__ __forceinline__ void am_softmax(f32x16& sT, const float* __restrict__ brow, int n, int hh, bool tok) {
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int j = am_key(e, hh);
    float v = sT[e];
    if (brow && tok && j < n) v += brow[j];
    v = j < n ? v : -INFINITY;
    sT[e] = v;
    mx = fmaxf(mx, v);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float l = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { sT[e] = expf(sT[e] - mx); l += sT[e]; }
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
#pragma unroll
  for (int e = 0; e < 16; ++e) sT[e] *= inv;
}

// NT: the token count as a compile-time constant (24 frames: every temporal attention of the base models) or 0 = p.d.n_tok. With a
// constant n every row / key guard below folds (row groups are 8 rows, keys come in runs of 4 per lane half): the generic kernel spends
// 85 predicated regions and ~90 selects per item on them -- it is bound by instruction issue (~8 K cycles per item and SIMD), not by memory.
template <int NT>
__global__ __launch_bounds__(64 * AM_WAVES, 4) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                                          const float* __restrict__ rsin, const float* __restrict__ bias,
                                                                          float* __restrict__ out, AttnP p) {
  __shared__ __attribute__((aligned(16))) float tiles[AM_WAVES][2][32 * AM_TS];
  const int n = NT ? NT : p.d.n_tok;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
  const bool tok = li < n;
  // planes of out for the to_out projection instead of the fp32 tensor (the MFMA backward does not read out): rows of P sum to 1,
  // so |out| <= max|v| <= max|qkv|
  float fps = 1.0f;
  if (p.pl_hi && p.pl_lo) {
    fps = scale_from_amax(amax_record_read(p.rec_qkv));
    if (blockIdx.x == 0 && threadIdx.x == 0) p.pl_scale[0] = fps;
  }
  float am = 0.f;
  float* Tq = tiles[wave][0];
  float* Tk = tiles[wave][1];
  for (int64_t item = (int64_t)blockIdx.x * AM_WAVES + wave; item < p.n_items; item += (int64_t)gridDim.x * AM_WAVES) {
    const int h = (int)(item % p.d.heads);
    const int64_t unit = item / p.d.heads;
    const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    const int64_t rowl = row0 + (int64_t)(tok ? li : 0) * p.d.st;
    const float* qb = am_uniform(qkv + row0 * p.RW + h * DH);          // q of token 0 of this item; k at + HD, v at + 2 HD
    const unsigned tstride = (unsigned)(p.d.st * p.RW);
    am_stage_rows(Tq, qb, tstride, rcos, rsin, p.scale, n, lane);
    am_stage_rows(Tk, qb + p.HD, tstride, rcos, rsin, 1.0f, n, lane);
    // V elements for the second product: step m needs V[key(m, hh)][d = li] (coalesced 128-byte rows)
    float va[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int j = am_key(m, hh);
      va[m] = j < n ? qb[(unsigned)j * tstride + (unsigned)(2 * p.HD + li)] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    float qs[16], ks[16];
    am_sel(Tq, li, hh, qs);
    am_sel(Tk, li, hh, ks);
    f32x16 sT;
#pragma unroll
    for (int e = 0; e < 16; ++e) sT[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[m], qs[m], sT, 0, 0, 0);
    am_softmax(sT, bias ? bias + ((int64_t)h * n + (tok ? li : 0)) * n : nullptr, n, hh, tok);
    f32x16 oT;
#pragma unroll
    for (int e = 0; e < 16; ++e) oT[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(va[m], sT[m], oT, 0, 0, 0);
    if (tok) {
      float* orow = out + rowl * p.HD + h * DH;
      float4 vv[4];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        vv[e4] = make_float4(oT[4 * e4], oT[4 * e4 + 1], oT[4 * e4 + 2], oT[4 * e4 + 3]);
        if (!p.pl_hi) *reinterpret_cast<float4*>(orow + 8 * e4 + 4 * hh) = vv[e4];
        am = amax4(am, vv[e4]);
      }
      if (p.pl_hi) am_store_row_planes(p.pl_hi, p.pl_lo, rowl * p.HD + h * DH, hh, vv, fps);      // instead of out
    }
    __builtin_amdgcn_wave_barrier();      // the next item overwrites the tiles
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * AM_WAVES + wave);
}

// 32 < n_tok <= 64 (the super-resolution model attends over 48 frames): the same one-wave-per-(unit, head) formulation with keys and
// queries in two 32-wide tiles each. Q and K of all 64 rows are staged once ([64][AM_TS] per wave and operand); per query tile the two
// score tiles (32 MFMA steps), one softmax over both, and the two P V products into one accumulator (32 steps).
__global__ __launch_bounds__(64 * AM_WAVES, 2) void attn_fwd_mfma64_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                                            const float* __restrict__ rsin, const float* __restrict__ bias,
                                                                            float* __restrict__ out, AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = p.d.n_tok;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
  float fps = 1.0f;
  if (p.pl_hi && p.pl_lo) {
    fps = scale_from_amax(amax_record_read(p.rec_qkv));
    if (blockIdx.x == 0 && threadIdx.x == 0) p.pl_scale[0] = fps;
  }
  float am = 0.f;
  float* Tq = smem + wave * (2 * 64 * AM_TS);
  float* Tk = Tq + 64 * AM_TS;
  for (int64_t item = (int64_t)blockIdx.x * AM_WAVES + wave; item < p.n_items; item += (int64_t)gridDim.x * AM_WAVES) {
    const int h = (int)(item % p.d.heads);
    const int64_t unit = item / p.d.heads;
    const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    const float* qb = am_uniform(qkv + row0 * p.RW + h * DH);
    const unsigned tstride = (unsigned)(p.d.st * p.RW);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float* rc = rcos ? rcos + half * 32 * DH : nullptr;
      const float* rs = rsin ? rsin + half * 32 * DH : nullptr;
      am_stage_rows(Tq + half * 32 * AM_TS, qb + (unsigned)(half * 32) * tstride, tstride, rc, rs, p.scale, n - half * 32, lane);
      am_stage_rows(Tk + half * 32 * AM_TS, qb + (unsigned)(half * 32) * tstride + p.HD, tstride, rc, rs, 1.0f, n - half * 32, lane);
    }
    // V elements of both key tiles: step m of tile jt needs V[32 jt + key(m, hh)][d = li]
    float va[2][16];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int j = 32 * jt + am_key(m, hh);
        va[jt][m] = j < n ? qb[(unsigned)j * tstride + (unsigned)(2 * p.HD + li)] : 0.f;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      const int qi = 32 * it + li;                     // this lane's query
      const bool tok = qi < n;
      if (32 * it >= n) break;
      float qs[16];
      am_sel(Tq + it * 32 * AM_TS, li, hh, qs);
      f32x16 sT[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        float ks[16];
        am_sel(Tk + jt * 32 * AM_TS, li, hh, ks);
#pragma unroll
        for (int e = 0; e < 16; ++e) sT[jt][e] = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) sT[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[m], qs[m], sT[jt], 0, 0, 0);
      }
      // softmax over the 64 keys of query qi: 32 in this lane (two tiles x 16), 32 in lane ^ 32
      const float* brow = bias ? bias + ((int64_t)h * n + (tok ? qi : 0)) * n : nullptr;
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = 32 * jt + am_key(e, hh);
          float v = sT[jt][e];
          if (brow && tok && j < n) v += brow[j];
          v = j < n ? v : -INFINITY;
          sT[jt][e] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float l = 0.f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { sT[jt][e] = expf(sT[jt][e] - mx); l += sT[jt][e]; }
      l += __shfl_xor(l, 32);
      const float inv = 1.0f / l;
      f32x16 oT;
#pragma unroll
      for (int e = 0; e < 16; ++e) oT[e] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int m = 0; m < 16; ++m) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(va[jt][m], sT[jt][m] * inv, oT, 0, 0, 0);
      // both lanes of a pair (l, l + 32) hold the same query, so `tok` is uniform over the pair (am_store_row_planes needs that)
      if (tok) {
        const int64_t rowl = row0 + (int64_t)qi * p.d.st;
        float* orow = out + rowl * p.HD + h * DH;
        float4 vv[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          vv[e4] = make_float4(oT[4 * e4], oT[4 * e4 + 1], oT[4 * e4 + 2], oT[4 * e4 + 3]);
          if (!p.pl_hi) *reinterpret_cast<float4*>(orow + 8 * e4 + 4 * hh) = vv[e4];
          am = amax4(am, vv[e4]);
        }
        if (p.pl_hi) am_store_row_planes(p.pl_hi, p.pl_lo, rowl * p.HD + h * DH, hh, vv, fps);
      }
    }
    __builtin_amdgcn_wave_barrier();      // the next item overwrites the tiles
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * AM_WAVES + wave);
}

// n_tok > 64 without rotation and bias (the mid spatial attention: 100 tokens at the bench size, 400 in the super-resolution model): one wave
// per (unit, head, tile of 32 queries), the keys walked in tiles of 32 with an online softmax -- the thread-per-row kernel took 89 us for the
// 768 items of a sampling step and 614 us for the super-resolution model's 384 items of 400 tokens. No LDS: every operand of the two exact-fp32
// products of a tile is loaded straight into the accumulator layout -- a lane owns one token and the feature runs 8 c + 4 hh + (0..3), which is
// what S^T = K Q^T wants of q and k (four 16-byte loads of the token's row each), and v column d = li of the 16 keys of its lane half (coalesced
// 128-byte rows); P^T feeds O^T += V^T P^T in place; the running maximum and sum of a query live in its lane pair.
__global__ __launch_bounds__(64 * AM_WAVES, 3) void attn_fwd_mfma_tiled_kernel(const float* __restrict__ qkv, float* __restrict__ out, AttnP p) {
  const int n = p.d.n_tok, ntile = (n + 31) >> 5;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
  const int64_t nwork = p.n_items * ntile;
  const unsigned tstride = (unsigned)(p.d.st * p.RW);
  float am = 0.f;
  for (int64_t wk = (int64_t)blockIdx.x * AM_WAVES + wave; wk < nwork; wk += (int64_t)gridDim.x * AM_WAVES) {
    const int64_t item = wk / ntile;
    const int it = (int)(wk - item * ntile);
    const int h = (int)(item % p.d.heads);
    const int64_t unit = item / p.d.heads;
    const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    const float* qb = am_uniform(qkv + row0 * p.RW + h * DH);          // q of token 0 of this item; k at + HD, v at + 2 HD
    const int qi = 32 * it + li;
    const bool tok = qi < n;
    float qs[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 v4 = tok ? *reinterpret_cast<const float4*>(qb + (unsigned)qi * tstride + 8 * c + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
      qs[4 * c] = v4.x * p.scale; qs[4 * c + 1] = v4.y * p.scale; qs[4 * c + 2] = v4.z * p.scale; qs[4 * c + 3] = v4.w * p.scale;
    }
    f32x16 oT;
#pragma unroll
    for (int e = 0; e < 16; ++e) oT[e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
    for (int jt = 0; jt < ntile; ++jt) {
      const int kj = 32 * jt + li;
      float ks[16], va[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 v4 = kj < n ? *reinterpret_cast<const float4*>(qb + (unsigned)kj * tstride + (unsigned)(p.HD + 8 * c + 4 * hh)) : make_float4(0.f, 0.f, 0.f, 0.f);
        ks[4 * c] = v4.x; ks[4 * c + 1] = v4.y; ks[4 * c + 2] = v4.z; ks[4 * c + 3] = v4.w;
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int j = 32 * jt + am_key(m, hh);
        va[m] = j < n ? qb[(unsigned)j * tstride + (unsigned)(2 * p.HD + li)] : 0.f;
      }
      f32x16 sT;
#pragma unroll
      for (int e = 0; e < 16; ++e) sT[e] = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[m], qs[m], sT, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = 32 * jt + am_key(e, hh) < n ? sT[e] : -INFINITY;
        sT[e] = v;
        mx = fmaxf(mx, v);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);                       // finite: every key tile holds at least one key
      const float corr = expf(m_run - m_new);                     // (0 for the first tile)
      float l = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { sT[e] = expf(sT[e] - m_new); l += sT[e]; }
      l += __shfl_xor(l, 32);
      l_run = l_run * corr + l;
      m_run = m_new;
#pragma unroll
      for (int e = 0; e < 16; ++e) oT[e] *= corr;
#pragma unroll
      for (int m = 0; m < 16; ++m) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(va[m], sT[m], oT, 0, 0, 0);
    }
    if (tok) {
      const float inv = 1.0f / l_run;
      float* orow = out + (row0 + (int64_t)qi * p.d.st) * p.HD + h * DH;
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const float4 vv = make_float4(oT[4 * e4] * inv, oT[4 * e4 + 1] * inv, oT[4 * e4 + 2] * inv, oT[4 * e4 + 3] * inv);
        *reinterpret_cast<float4*>(orow + 8 * e4 + 4 * hh) = vv;
        am = amax4(am, vv);
      }
    }
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * AM_WAVES + wave);
}

// The backward of the same case (n_tok > 64, no rotation, no bias), one block of four waves per (unit, head) item, nothing n x n stored and
// every product on the exact-fp32 matrix instruction with operands loaded straight into the layout the product wants (rows of q / k / v / dO as
// four 16-byte loads of a token's features, or column d = li of 16 tokens):
//   phase A, a wave per tile of 32 queries: pass 1 over the key tiles for the softmax statistics (m_i, 1 / l_i); delta_i = <dO_i, O_i>; pass 2:
//     S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T (dS^T feeds the product in place); statistics to LDS;
//   phase B, a wave per tile of 32 keys, over the query tiles: S = Q K^T, dP = dO V^T, P and dS from the statistics, dV^T += dO^T P, dK^T += Q^T dS.
// The thread-per-row kernel (attn_bwd_big_kernel: 246 us for the 768 items of 100 tokens of a training step) remains for rotation / bias.
// STAGED (n <= 128, the training case): q (scaled), k, v, dO of the item are staged once in LDS tiles [128][36] and every operand is read from
// there (a global round trip per product stage of a tile left the matrix pipe idle: 107 us; staged: see DESIGN.md); larger n reads global memory.
#define ATT_TILED_MAXTOK 576
#define ATT_TILED_STAGE 128
#define ATT_TILED_TS 36
template <bool STAGED>
__global__ __launch_bounds__(64 * AM_WAVES, 2) void attn_bwd_mfma_tiled_kernel(const float* __restrict__ qkv, const float* __restrict__ fout,
                                                                                const float* __restrict__ dout, float* __restrict__ dqkv, AttnP p) {
  __shared__ __attribute__((aligned(16))) float Ms[STAGED ? ATT_TILED_STAGE : ATT_TILED_MAXTOK], Ls[STAGED ? ATT_TILED_STAGE : ATT_TILED_MAXTOK],
      Dl[STAGED ? ATT_TILED_STAGE : ATT_TILED_MAXTOK];
  __shared__ __attribute__((aligned(16))) float Stg[STAGED ? 4 * ATT_TILED_STAGE * ATT_TILED_TS : 4];      // q | k | v | dO
  const int n = p.d.n_tok, ntile = (n + 31) >> 5;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
  const unsigned tq = (unsigned)(p.d.st * p.RW), to = (unsigned)(p.d.st * p.HD);
  float am = 0.f;
  for (int64_t item = blockIdx.x; item < p.n_items; item += gridDim.x) {
    const int h = (int)(item % p.d.heads);
    const int64_t unit = item / p.d.heads;
    const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    const float* qb = am_uniform(qkv + row0 * p.RW + h * DH);           // q of token 0; k at + HD, v at + 2 HD
    const float* gb = am_uniform(dout + row0 * p.HD + h * DH);
    const float* ob = am_uniform(fout + row0 * p.HD + h * DH);
    float* db = dqkv + row0 * p.RW + h * DH;
    // row `t` of a [token][feature] operand as the 16 values (features 8 c + 4 hh + 0..3) a lane feeds to a product over the features
    // (STAGED: `base` is matched to its LDS tile -- q, k, v at column offsets 0, HD, 2 HD of qkv; dO -- and rows beyond n hold zeros)
    auto tile_of = [&](const float* base, unsigned col0) -> const float* {
      return Stg + (base == gb ? 3 : col0 == 0 ? 0 : col0 == (unsigned)p.HD ? 1 : 2) * (ATT_TILED_STAGE * ATT_TILED_TS);
    };
    auto row16 = [&](const float* base, unsigned stride, int t, unsigned col0, float mul, float (&v)[16]) {
      if (STAGED) {
        const float* T = tile_of(base, col0) + t * ATT_TILED_TS;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 v4 = *reinterpret_cast<const float4*>(T + 8 * c + 4 * hh);
          v[4 * c] = v4.x; v[4 * c + 1] = v4.y; v[4 * c + 2] = v4.z; v[4 * c + 3] = v4.w;
        }
        return;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 v4 = t < n ? *reinterpret_cast<const float4*>(base + (unsigned)t * stride + col0 + 8 * c + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[4 * c] = v4.x * mul; v[4 * c + 1] = v4.y * mul; v[4 * c + 2] = v4.z * mul; v[4 * c + 3] = v4.w * mul;
      }
    };
    // column d = li of the 16 tokens t0 + am_key(m, hh): what a lane feeds to a product over the tokens
    auto col16 = [&](const float* base, unsigned stride, int t0, unsigned col0, float mul, float (&v)[16]) {
      if (STAGED) {
        const float* T = tile_of(base, col0) + t0 * ATT_TILED_TS + li;
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = T[am_key(m, hh) * ATT_TILED_TS];
        return;
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int t = t0 + am_key(m, hh);
        v[m] = t < n ? base[(unsigned)t * stride + col0 + li] * mul : 0.f;
      }
    };
    __syncthreads();                                  // the previous item's phase B has finished with the statistics (and the tiles)
    if (STAGED) {                                     // 4 tiles x 128 rows x 8 chunks of 16 bytes; q scaled here; rows beyond n zero
      for (int i = threadIdx.x; i < 4 * ATT_TILED_STAGE * 8; i += 64 * AM_WAVES) {
        const int tl = i >> 10, r = (i >> 3) & (ATT_TILED_STAGE - 1), ch = i & 7;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) v4 = tl < 3 ? *reinterpret_cast<const float4*>(qb + (unsigned)r * tq + (unsigned)(tl * p.HD + 4 * ch)) : *reinterpret_cast<const float4*>(gb + (unsigned)r * to + 4 * ch);
        if (tl == 0) { v4.x *= p.scale; v4.y *= p.scale; v4.z *= p.scale; v4.w *= p.scale; }
        *reinterpret_cast<float4*>(Stg + (tl * ATT_TILED_STAGE + r) * ATT_TILED_TS + 4 * ch) = v4;
      }
      __syncthreads();
    }
    // ------------------------------------------------------------------ phase A
#pragma unroll 1
    for (int it = wave; it < ntile; it += AM_WAVES) {
      const int qi = 32 * it + li;
      float qs[16], gs[16];
      row16(qb, tq, qi, 0, p.scale, qs);
      row16(gb, to, qi, 0, 1.0f, gs);
      float delta = 0.f;
      {
#pragma unroll
        for (int c = 0; c < 4; ++c) {                 // (the forward output: always from global memory)
          const float4 o4 = qi < n ? *reinterpret_cast<const float4*>(ob + (unsigned)qi * to + 8 * c + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
          delta = fmaf(gs[4 * c], o4.x, delta); delta = fmaf(gs[4 * c + 1], o4.y, delta);
          delta = fmaf(gs[4 * c + 2], o4.z, delta); delta = fmaf(gs[4 * c + 3], o4.w, delta);
        }
        delta += __shfl_xor(delta, 32);
      }
      float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
      for (int jt = 0; jt < ntile; ++jt) {
        float ks[16];
        row16(qb, tq, 32 * jt + li, p.HD, 1.0f, ks);
        f32x16 sT;
#pragma unroll
        for (int e = 0; e < 16; ++e) sT[e] = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[m], qs[m], sT, 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = 32 * jt + am_key(e, hh) < n ? sT[e] : -INFINITY;
          sT[e] = v;
          mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        float l = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) l += expf(sT[e] - m_new);
        l += __shfl_xor(l, 32);
        l_run = l_run * expf(m_run - m_new) + l;
        m_run = m_new;
      }
      const float inv_l = 1.0f / l_run;
      f32x16 dqT;
#pragma unroll
      for (int e = 0; e < 16; ++e) dqT[e] = 0.f;
#pragma unroll 1
      for (int jt = 0; jt < ntile; ++jt) {
        f32x16 sT, dpT;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sT[e] = 0.f; dpT[e] = 0.f; }
        {
          float ks[16], vs[16];
          row16(qb, tq, 32 * jt + li, p.HD, 1.0f, ks);
          row16(qb, tq, 32 * jt + li, 2 * p.HD, 1.0f, vs);
#pragma unroll
          for (int m = 0; m < 16; ++m) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[m], qs[m], sT, 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 16; ++m) dpT = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[m], gs[m], dpT, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = 32 * jt + am_key(e, hh) < n ? expf(sT[e] - m_run) * inv_l : 0.f;
          sT[e] = pe * (dpT[e] - delta);              // dS^T
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          float kc[16];
          col16(qb, tq, 32 * jt, p.HD, 1.0f, kc);
#pragma unroll
          for (int m = 0; m < 16; ++m) dqT = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[m], sT[m], dqT, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (qi < n) {
        float* dr = db + (unsigned)qi * tq;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 vv = make_float4(dqT[4 * c] * p.scale, dqT[4 * c + 1] * p.scale, dqT[4 * c + 2] * p.scale, dqT[4 * c + 3] * p.scale);
          *reinterpret_cast<float4*>(dr + 8 * c + 4 * hh) = vv;
          am = amax4(am, vv);
        }
        if (hh == 0) { Ms[qi] = m_run; Ls[qi] = inv_l; Dl[qi] = delta; }
      }
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase B
#pragma unroll 1
    for (int jt = wave; jt < ntile; jt += AM_WAVES) {
      const int kj = 32 * jt + li;
      float kb[16], vb[16];
      row16(qb, tq, kj, p.HD, 1.0f, kb);
      row16(qb, tq, kj, 2 * p.HD, 1.0f, vb);
      f32x16 dkT, dvT;
#pragma unroll
      for (int e = 0; e < 16; ++e) { dkT[e] = 0.f; dvT[e] = 0.f; }
#pragma unroll 1
      for (int it = 0; it < ntile; ++it) {
        f32x16 S, dP;
        {
          float qa[16], ga[16];
          row16(qb, tq, 32 * it + li, 0, p.scale, qa);
          row16(gb, to, 32 * it + li, 0, 1.0f, ga);
#pragma unroll
          for (int e = 0; e < 16; ++e) { S[e] = 0.f; dP[e] = 0.f; }
#pragma unroll
          for (int m = 0; m < 16; ++m) S = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m], kb[m], S, 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 16; ++m) dP = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[m], vb[m], dP, 0, 0, 0);
        }
        // rows of S are the queries 32 it + 8 c + 4 hh + (0..3): their statistics, four at a time
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r0 = 32 * it + 8 * c + 4 * hh;
          float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f), l4 = m4, d4 = m4;
          if (r0 < n) {                                // (n is a multiple of 4 or the tail rows read statistics nobody wrote: masked below)
            m4 = *reinterpret_cast<const float4*>(Ms + r0); l4 = *reinterpret_cast<const float4*>(Ls + r0); d4 = *reinterpret_cast<const float4*>(Dl + r0);
          }
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = 4 * c + j;
            const float pe = r0 + j < n ? expf(S[e] - mm[j]) * ll[j] : 0.f;
            S[e] = pe;                                 // P
            dP[e] = pe * (dP[e] - dd[j]);              // dS
          }
        }
        __builtin_amdgcn_sched_barrier(0);             // (the column loads stay behind the row operands: registers)
        {
          float gc[16];
          col16(gb, to, 32 * it, 0, 1.0f, gc);
#pragma unroll
          for (int m = 0; m < 16; ++m) dvT = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[m], S[m], dvT, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          float qc[16];
          col16(qb, tq, 32 * it, 0, p.scale, qc);
#pragma unroll
          for (int m = 0; m < 16; ++m) dkT = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[m], dP[m], dkT, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kj < n) {
        float* dr = db + (unsigned)kj * tq;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 kk = make_float4(dkT[4 * c], dkT[4 * c + 1], dkT[4 * c + 2], dkT[4 * c + 3]);
          const float4 vv = make_float4(dvT[4 * c], dvT[4 * c + 1], dvT[4 * c + 2], dvT[4 * c + 3]);
          *reinterpret_cast<float4*>(dr + p.HD + 8 * c + 4 * hh) = kk;
          *reinterpret_cast<float4*>(dr + 2 * p.HD + 8 * c + 4 * hh) = vv;
          am = amax4(amax4(am, kk), vv);
        }
      }
    }
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * AM_WAVES + wave);
}

// element (row j, channel li) of a rotated q / k row, read column-wise (coalesced 128-byte rows): the rotation partner sits in
// the neighbouring lane
__device__ __forceinline__ float am_rot_elem(float x, const float* __restrict__ rc, const float* __restrict__ rs, int j, int li, bool ok) {
  if (!rc) return x;
  const float partner = __shfl_xor(x, 1);
  const float c = ok ? rc[j * DH + li] : 1.f, sn = ok ? rs[j * DH + li] : 0.f;
  return (li & 1) ? x * c + partner * sn : x * c - partner * sn;
}
// gradient of the rotation on a run of four consecutive channels d0..d0+3 held by one lane (pairs are inside the run)
__device__ __forceinline__ float4 am_unrotate4(float4 g, const float* __restrict__ rc, const float* __restrict__ rs, int d0) {
  if (!rc) return g;
  const float4 c = *reinterpret_cast<const float4*>(rc + d0), s = *reinterpret_cast<const float4*>(rs + d0);
  return make_float4(g.x * c.x + g.y * s.x, g.y * c.y - g.x * s.y, g.z * c.z + g.w * s.z, g.w * c.w - g.z * s.w);
}

template <int NT>
__global__ __launch_bounds__(64 * AM_WAVES, ATT_BWD_BLOCKS_PER_CU) void attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                                       const float* __restrict__ rsin, const float* __restrict__ bias,
                                                                       const float* __restrict__ fout, const float* __restrict__ dout,
                                                                       float* __restrict__ dqkv, float* __restrict__ dbias, AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = NT ? NT : p.d.n_tok;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
  constexpr int WAVE_LDS = 2 * 32 * AM_TS + 32;
  float* Ta = smem + wave * WAVE_LDS;                // two staged operand tiles (q, k then v, dO), then P^T / dS^T ([j][i]) in the same
  float* Tb = Ta + 32 * AM_TS;                       // storage once the staged operands have been consumed; delta[32]
  float* dl = Tb + 32 * AM_TS;
  float* Pt = Ta;
  float* St = Tb;
  float* dBs = smem + AM_WAVES * WAVE_LDS;            // [heads][n][n] when dbias
  // Relative-position-bias gradient = sum of dS over the units. With heads == AM_WAVES a wave always works on the head `wave`
  // (item = 4 (block + k grid) + wave), so it sums its items' dS tiles in REGISTERS, in item order; otherwise LDS atomics.
  const bool db_regs = dbias && p.d.heads == AM_WAVES;
  f32x16 dbacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) dbacc[e] = 0.f;
  if (dbias && !db_regs) {
    for (int e = threadIdx.x; e < p.d.heads * n * n; e += 64 * AM_WAVES) dBs[e] = 0.f;
    __syncthreads();
  }
  const bool tok = li < n;
  float am = 0.f;                                    // max |dqkv| this lane has stored
  // Planes output: with A = max|qkv|, G = max|dout|, P row sums 1:  |dV| <= n G;  |dP| <= 32 G A;  sum_j |dS_ij| <= 2 * 32 G A and
  // sum_i |dS_ij| <= n * that;  |dK| <= sqrt2 * scale * n * 64 G A^2 (the rotation keeps pair norms), |dQ| the same without n.
  // A bound this loose (2^10 and more above the true maximum) still leaves the planes ~2^-28 max|dqkv| accurate: DESIGN.md.
  float ps = 1.0f;
  if (p.pl_hi && p.pl_lo) {
    const float A = amax_record_read(p.rec_qkv), G = amax_record_read(p.rec_dout);
    ps = scale_from_amax(1.01f * fmaxf((float)n * G, 1.4143f * p.scale * (float)n * 64.f * G * A * A));
    if (blockIdx.x == 0 && threadIdx.x == 0) p.pl_scale[0] = ps;
  }
  for (int64_t item = (int64_t)blockIdx.x * AM_WAVES + wave; item < p.n_items; item += (int64_t)gridDim.x * AM_WAVES) {
    const int h = (int)(item % p.d.heads);
    const int64_t unit = item / p.d.heads;
    const int uo = (int)(unit / p.d.n_ui), ui = (int)(unit - (int64_t)uo * p.d.n_ui);
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    const float* qb = am_uniform(qkv + row0 * p.RW + h * DH);          // q of token 0 of this item; k at + HD, v at + 2 HD
    const float* gb = am_uniform(dout + row0 * p.HD + h * DH);
    const float* fb = am_uniform(fout + row0 * p.HD + h * DH);
    float* db = const_cast<float*>(am_uniform(dqkv + row0 * p.RW + h * DH));
    const int64_t pbase = row0 * p.RW + h * DH;                        // the same element offset inside the planes
    const unsigned lrow = (unsigned)(tok ? li : 0);
    // key / query index of step m for this lane = 8*(m>>2) + (m&3) + hz with hz = 4*hh, made opaque per item: otherwise the
    // compiler hoists all 48 per-lane row offsets (as 64-bit pairs) out of the item loop and the kernel needs 270 registers
    unsigned hz = 4u * (unsigned)hh;
    asm volatile("" : "+v"(hz));
    const unsigned tstride = (unsigned)(p.d.st * p.RW), gstride = (unsigned)(p.d.st * p.HD);
    f32x16 pT, dsT;
    // Phase order (round 2): V / dO first, Q / K second, so that the scaled + rotated Q and K tiles are still in LDS when the dQ and dK
    // products need their columns. Round 1 staged Q / K first and re-read K, Q and dO column-wise from global memory for the three
    // gradient products, re-applying the rotation per element (two table loads each); now only dO is re-read. Still two tiles per wave.
    {
      // dP^T[j][i] = sum_d V[j][d] dO[i][d]
      am_stage_rows(Ta, qb + 2 * p.HD, tstride, nullptr, nullptr, 1.0f, n, lane);
      am_stage_rows(Tb, gb, gstride, nullptr, nullptr, 1.0f, n, lane);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < 16; ++e) dsT[e] = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {          // four steps at a time: bounded live registers (the kernel runs at 128 VGPRs)
        float va[4], gg[4];
        am_sel4(Ta, li, hh, g4, va);
        am_sel4(Tb, li, hh, g4, gg);
#pragma unroll
        for (int q = 0; q < 4; ++q) dsT = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q], gg[q], dsT, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      __builtin_amdgcn_wave_barrier();                 // every lane has taken its V / dO values: the tiles may be overwritten
      am_stage_rows<2>(Ta, qb, tstride, rcos, rsin, p.scale, n, lane);
      am_stage_rows<2>(Tb, qb + p.HD, tstride, rcos, rsin, 1.0f, n, lane);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < 16; ++e) pT[e] = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float qa[4], kb[4];
        am_sel4(Ta, li, hh, g4, qa);
        am_sel4(Tb, li, hh, g4, kb);
#pragma unroll
        for (int q = 0; q < 4; ++q) pT = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[q], qa[q], pT, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      am_softmax(pT, bias ? bias + ((int64_t)h * n + (tok ? li : 0)) * n : nullptr, n, hh, tok);
      // delta_i = <dO_i, O_i> = sum_j P_ij dP_ij: both factors are in this lane's accumulators (16 keys here, 16 in lane ^ 32; masked
      // keys have P = 0) -- the forward output is not read at all (round 1 re-read it: 157 MB per level-0 launch)
      float delta = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) delta = fmaf(pT[e], dsT[e], delta);
      delta += __shfl_xor(delta, 32);
#pragma unroll
      for (int e = 0; e < 16; ++e) dsT[e] = pT[e] * (dsT[e] - delta);
    }
    // dQ^T[d][i] = sum_j K[j][d] dS^T[j][i]: K columns from the staged (rotated) tile, dS^T straight from the accumulator
    {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float ka[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = 4 * g4 + q;
          const unsigned j = (unsigned)(8 * (m >> 2) + (m & 3)) + hz;
          ka[q] = Tb[j * AM_TS + (unsigned)li];        // rows >= n are zero-filled by the staging
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[q], dsT[4 * g4 + q], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tok) {
        float* drow = db + lrow * tstride;
        float4 gqv[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int d0 = 8 * e4 + 4 * hh;
          float4 gq = am_unrotate4(make_float4(acc[4 * e4], acc[4 * e4 + 1], acc[4 * e4 + 2], acc[4 * e4 + 3]), rcos ? rcos + li * DH : nullptr, rsin ? rsin + li * DH : nullptr, d0);
          gq = make_float4(gq.x * p.scale, gq.y * p.scale, gq.z * p.scale, gq.w * p.scale);
          gqv[e4] = gq;
          if (!p.pl_hi) *reinterpret_cast<float4*>(drow + d0) = gq;
          am = amax4(am, gq);
        }
        if (p.pl_hi) am_store_row_planes(p.pl_hi, p.pl_lo, pbase + (int64_t)(lrow * tstride), hh, gqv, ps);
      }
    }
    // dS with the lane roles swapped (lane = key) goes through LDS, over the K tile every lane has finished with; the
    // relative-position-bias gradient is dS itself
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = am_key(e, hh);
      St[j * AM_TS + li] = dsT[e];
      if (db_regs) dbacc[e] += dsT[e];                 // entries with li >= n or j >= n are never written out
      else if (dbias && tok && j < n) atomicAdd(&dBs[(h * n + li) * n + j], dsT[e]);
    }
    __builtin_amdgcn_wave_barrier();
    // dK^T[d][j] = sum_i Q[i][d] dS[i][j]: Q columns from the staged (scaled, rotated) tile
    {
      f32x16 dk;
#pragma unroll
      for (int e = 0; e < 16; ++e) dk[e] = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float qa[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = 4 * g4 + q;
          const unsigned i = (unsigned)(8 * (m >> 2) + (m & 3)) + hz;
          qa[q] = Ta[i * AM_TS + (unsigned)li];
        }
        // this lane's row of dS, four consecutive queries as ONE 16-byte read (four ds_read_b32 at a row stride of 36 floats were 2-way
        // bank-conflicted: 36 li mod 64 takes 16 values for 32 lanes)
        const float4 sb4 = *reinterpret_cast<const float4*>(St + li * AM_TS + 8 * g4 + hz);
        const float sb[4] = {sb4.x, sb4.y, sb4.z, sb4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) dk = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[q], sb[q], dk, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tok) {
        float* drow = db + (lrow * tstride + (unsigned)p.HD);
        float4 gkv[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int d0 = 8 * e4 + 4 * hh;
          const float4 gk = am_unrotate4(make_float4(dk[4 * e4], dk[4 * e4 + 1], dk[4 * e4 + 2], dk[4 * e4 + 3]), rcos ? rcos + li * DH : nullptr, rsin ? rsin + li * DH : nullptr, d0);
          gkv[e4] = gk;
          if (!p.pl_hi) *reinterpret_cast<float4*>(drow + d0) = gk;
          am = amax4(am, gk);
        }
        if (p.pl_hi) am_store_row_planes(p.pl_hi, p.pl_lo, pbase + (int64_t)(lrow * tstride + (unsigned)p.HD), hh, gkv, ps);
      }
    }
    // P with the lane roles swapped, over the Q tile; dV^T[d][j] = sum_i dO[i][d] P[i][j] (dO columns are the one global re-read left)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) Pt[am_key(e, hh) * AM_TS + li] = pT[e];
    __builtin_amdgcn_wave_barrier();
    {
      f32x16 dv;
#pragma unroll
      for (int e = 0; e < 16; ++e) dv[e] = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float ga[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = 4 * g4 + q;
          const unsigned i = (unsigned)(8 * (m >> 2) + (m & 3)) + hz;
          ga[q] = i < (unsigned)n ? gb[i * gstride + (unsigned)li] : 0.f;
        }
        const float4 pb4 = *reinterpret_cast<const float4*>(Pt + li * AM_TS + 8 * g4 + hz);
        const float pb[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) dv = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[q], pb[q], dv, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tok) {
        float* drow = db + (lrow * tstride + (unsigned)(2 * p.HD));
        float4 gvv[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const float4 gv = make_float4(dv[4 * e4], dv[4 * e4 + 1], dv[4 * e4 + 2], dv[4 * e4 + 3]);
          gvv[e4] = gv;
          if (!p.pl_hi) *reinterpret_cast<float4*>(drow + 8 * e4 + 4 * hh) = gv;
          am = amax4(am, gv);
        }
        if (p.pl_hi) am_store_row_planes(p.pl_hi, p.pl_lo, pbase + (int64_t)(lrow * tstride + (unsigned)(2 * p.HD)), hh, gvv, ps);
      }
    }
    __builtin_amdgcn_wave_barrier();      // the next item overwrites this wave's tiles
  }
  if (p.amax_rec) wave_amax_emit(am, p.amax_rec, (int)blockIdx.x * AM_WAVES + wave);
  if (dbias) {         // dbias: this block's partial [heads][n][n]; attn_dbias_reduce_kernel sums the blocks in index order
    float* part = dbias + (size_t)blockIdx.x * p.d.heads * n * n;
    if (db_regs) {
      if (tok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = am_key(e, hh);
          if (j < n) part[(wave * n + li) * n + j] = dbacc[e];
        }
      }
    } else {
      __syncthreads();
      for (int e = threadIdx.x; e < p.d.heads * n * n; e += 64 * AM_WAVES) part[e] = dBs[e];
    }
  }
}

static void attn_fwd_mfma64_launch(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out,
                                   const AttnP& p, hipStream_t st) {
  const size_t lds = (size_t)AM_WAVES * 2 * 64 * AM_TS * sizeof(float);       // 73.7 KB: two blocks per CU
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute((const void*)attn_fwd_mfma64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  int64_t nb = (p.n_items + AM_WAVES - 1) / AM_WAVES;
  if (nb > 4096) nb = 4096;
  attn_fwd_mfma64_kernel<<<(unsigned)nb, 64 * AM_WAVES, lds, st>>>(qkv, rot_cos, rot_sin, bias, out, p);
}
static int attn_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}
// Grid of the one-wave-per-item kernels: exactly the blocks that are RESIDENT together (blocks per CU x CUs), each wave walking over
// its items with a grid stride. The first versions launched up to 4096 / 2048 blocks: several rounds of the resident set with the last
// one partly empty (2048 blocks on 768 slots = 2.67 rounds), and waves with 3 or 4 items each (a 28 % imbalance at the 40 x 40
// level); with 768 / 1024 blocks every wave gets 12-17 items, i.e. at most one item of imbalance. wdno_debug 41: the old caps (A/B).
static int64_t attn_grid(int64_t items, int blocks_per_cu, int64_t old_cap) {
  int64_t nb = (items + AM_WAVES - 1) / AM_WAVES;
  const int64_t cap = wdno_debug_mode == 41 ? old_cap : (int64_t)blocks_per_cu * attn_num_cus();
  return nb > cap ? cap : nb;
}
static int attn_fill(AttnP& p, const wdno_attn_desc* d, float scale, int threads) {
  if (!d || d->n_uo <= 0 || d->n_ui <= 0 || d->n_tok <= 0 || d->heads <= 0) return WDNO_EINVAL;
  p.d = *d; p.scale = scale; p.HD = d->heads * DH; p.RW = 3 * p.HD;
  p.ipb = threads / d->n_tok;
  if (p.ipb < 1) p.ipb = 1;
  p.kst = d->n_tok * DH + 8;          // +8 floats: items start on different 32-byte LDS slots
  p.n_items = (int64_t)d->n_uo * d->n_ui * d->heads;
  p.amax_rec = nullptr;
  p.pl_hi = p.pl_lo = nullptr; p.rec_qkv = p.rec_dout = nullptr; p.pl_scale = nullptr;
  return WDNO_OK;
}
extern "C" int wdno_attn_fwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out,
                             const wdno_attn_desc* d, float scale, wdno_stream_t s) {
  return wdno_attn_fwd_amax(qkv, rot_cos, rot_sin, bias, out, nullptr, d, scale, s);
}
// the thread-per-row kernels do not track the amax of what they write: a sweep over the result fills the record instead
static int attn_amax_sweep(int rc, const float* t, const wdno_attn_desc* d, int width, float* amax_rec, wdno_stream_t s) {
  if (rc != WDNO_OK || !amax_rec) return rc;
  return wdno_amax_record(t, (int64_t)d->n_uo * d->n_ui * d->n_tok * width, amax_rec, s);
}
static int attn_fwd_rows(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out, AttnP& p,
                         const wdno_attn_desc* d, wdno_stream_t s);
extern "C" int wdno_attn_fwd_amax(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out,
                                  float* amax_rec, const wdno_attn_desc* d, float scale, wdno_stream_t s) {
  AttnP p;
  int rc = attn_fill(p, d, scale, ATT_THREADS);
  if (rc) return rc;
  if (d->n_tok > 1024) return WDNO_EUNSUPPORTED;
  if (d->n_tok <= 32 && wdno_debug_mode != 5) {                  // one 32x32 MFMA tile per (unit, head): one wave per item
    const int64_t nb = attn_grid(p.n_items, 4, 4096);
    p.amax_rec = amax_rec;
    (d->n_tok == 24 && wdno_debug_mode != 44 ? attn_fwd_mfma_kernel<24> : attn_fwd_mfma_kernel<0>)<<<(unsigned)nb, 64 * AM_WAVES, 0, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, p);
    return wdno_check_launch();
  }
  if (d->n_tok <= 64 && wdno_debug_mode != 5) {                  // two 32-wide tiles of keys and of queries per item
    p.amax_rec = amax_rec;
    attn_fwd_mfma64_launch(qkv, rot_cos, rot_sin, bias, out, p, as_stream(s));
    return wdno_check_launch();
  }
  if (!rot_cos && !bias && wdno_debug_mode != 5 && wdno_debug_mode != 67 && wdno_debug_mode != 69) {      // (debug 69: forward and backward thread per row)      // key tiles with an online softmax, one wave per tile of 32 queries (debug 67: thread per row)
    const int ntile = (d->n_tok + 31) / 32;
    const int64_t nb = attn_grid(p.n_items * ntile, 3, (int64_t)3 * attn_num_cus());
    p.amax_rec = amax_rec;
    attn_fwd_mfma_tiled_kernel<<<(unsigned)nb, 64 * AM_WAVES, 0, as_stream(s)>>>(qkv, out, p);
    return wdno_check_launch();
  }
  return attn_amax_sweep(attn_fwd_rows(qkv, rot_cos, rot_sin, bias, out, p, d, s), out, d, p.HD, amax_rec, s);
}
// forward with out delivered ONLY as fp16 planes for the to_out projection (MFMA path only: n_tok <= 32; `out` is not written)
extern "C" int wdno_attn_fwd_planes(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out,
                                    void* out_hi, void* out_lo, float* out_scale, float* amax_rec, const float* rec_qkv,
                                    const wdno_attn_desc* d, float scale, wdno_stream_t s) {
  AttnP p;
  int rc = attn_fill(p, d, scale, ATT_THREADS);
  if (rc) return rc;
  if (d->n_tok > 64 || !out_hi || (out_lo && (!out_scale || !rec_qkv))) return WDNO_EUNSUPPORTED;      // out_lo == NULL: one bf16 plane
  const int64_t nb = attn_grid(p.n_items, 4, 4096);
  p.amax_rec = amax_rec;
  p.pl_hi = (_Float16*)out_hi; p.pl_lo = (_Float16*)out_lo; p.rec_qkv = rec_qkv; p.pl_scale = out_scale;
  if (d->n_tok > 32) {
    attn_fwd_mfma64_launch(qkv, rot_cos, rot_sin, bias, out, p, as_stream(s));
    return wdno_check_launch();
  }
  (d->n_tok == 24 && wdno_debug_mode != 44 ? attn_fwd_mfma_kernel<24> : attn_fwd_mfma_kernel<0>)<<<(unsigned)nb, 64 * AM_WAVES, 0, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, p);
  return wdno_check_launch();
}
static int attn_fwd_rows(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out, AttnP& p,
                         const wdno_attn_desc* d, wdno_stream_t s) {
  size_t lds = (size_t)p.ipb * p.kst * sizeof(float);
  if (lds > 160 * 1024) return WDNO_EUNSUPPORTED;
  int64_t blocks = (p.n_items + p.ipb - 1) / p.ipb;
  if (blocks > 0x7fffffff) return WDNO_EUNSUPPORTED;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_fwd_kernel<<<(unsigned)blocks, ATT_THREADS, lds, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, p);
  return wdno_check_launch();
}
// blocks of the backward launch (= number of dbias partials) and the workspace that holds them
static int64_t attn_bwd_blocks(const wdno_attn_desc* d) {
  const int64_t items = (int64_t)d->n_uo * d->n_ui * d->heads;
  if (d->n_tok <= 32 && wdno_debug_mode != 5) return attn_grid(items, ATT_BWD_BLOCKS_PER_CU, 2048);
  int ipb = ATT_BWD_THREADS / d->n_tok;
  if (ipb < 1) ipb = 1;
  int64_t groups = (items + ipb - 1) / ipb;
  return groups > 1024 ? 1024 : groups;
}
extern "C" size_t wdno_attn_bwd_ws_bytes(const wdno_attn_desc* d) {
  if (!d || d->n_tok <= 0 || d->heads <= 0) return 0;
  return (size_t)attn_bwd_blocks(d) * d->heads * d->n_tok * d->n_tok * sizeof(float);
}
static int attn_dbias_reduce(const float* part, int64_t nb, float* dbias, const wdno_attn_desc* d, wdno_stream_t s) {
  const int E = d->heads * d->n_tok * d->n_tok;
  attn_dbias_reduce_kernel<<<(unsigned)((E + DBR_EL - 1) / DBR_EL), 256, 0, as_stream(s)>>>(part, (int)nb, dbias, E);
  return wdno_check_launch();
}
extern "C" int wdno_attn_bwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                             const float* dout, float* dqkv, float* dbias, const wdno_attn_desc* d, float scale, void* ws, size_t ws_bytes,
                             wdno_stream_t s) {
  return wdno_attn_bwd_amax(qkv, rot_cos, rot_sin, bias, out, dout, dqkv, dbias, nullptr, d, scale, ws, ws_bytes, s);
}
static int attn_bwd_rows(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                         const float* dout, float* dqkv, float* dbias, AttnP& p, const wdno_attn_desc* d, wdno_stream_t s);
extern "C" int wdno_attn_bwd_amax(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                                  const float* dout, float* dqkv, float* dbias, float* amax_rec, const wdno_attn_desc* d, float scale,
                                  void* ws, size_t ws_bytes, wdno_stream_t s) {
  AttnP p;
  int rc = attn_fill(p, d, scale, ATT_BWD_THREADS);
  if (rc) return rc;
  const int n = d->n_tok;
  if (dbias && (!ws || ws_bytes < wdno_attn_bwd_ws_bytes(d))) return WDNO_EWORKSPACE;
  float* part = dbias ? (float*)ws : nullptr;           // the kernels write per-block partials; the reduce below writes dbias
  // one wave per item on MFMA tiles (debug 5: thread-per-row kernel below); 116 registers and 9 KB of LDS per wave, so four
  // waves per SIMD hide its five dependent load phases (0.94 vs 1.17 ms at the 40 x 40 level)
  if (n <= 32 && wdno_debug_mode != 5) {
    size_t lds2 = ((size_t)AM_WAVES * (2 * 32 * AM_TS + 32) + (dbias && d->heads != AM_WAVES ? (size_t)d->heads * n * n : 0)) * sizeof(float);      // (heads == AM_WAVES: the partial lives in registers)
    const int64_t nb = attn_bwd_blocks(d);                   // also the number of dbias partials
    p.amax_rec = amax_rec;
    (d->n_tok == 24 && wdno_debug_mode != 44 ? attn_bwd_mfma_kernel<24> : attn_bwd_mfma_kernel<0>)<<<(unsigned)nb, 64 * AM_WAVES, lds2, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, dout, dqkv, part, p);
    rc = wdno_check_launch();
    return (rc || !dbias) ? rc : attn_dbias_reduce(part, nb, dbias, d, s);
  }
  // more than 64 tokens without rotation / bias (the mid spatial block): key / query tiles on the exact-fp32 matrix instruction (debug 68: thread per row)
  if (n > 64 && n <= ATT_TILED_MAXTOK && !rot_cos && !bias && !dbias && out && wdno_debug_mode != 5 && wdno_debug_mode != 68 && wdno_debug_mode != 69) {
    int64_t nb = p.n_items;
    const int64_t cap = 2LL * attn_num_cus();
    if (nb > cap) nb = cap;
    p.amax_rec = amax_rec;
    if (n <= ATT_TILED_STAGE) attn_bwd_mfma_tiled_kernel<true><<<(unsigned)nb, 64 * AM_WAVES, 0, as_stream(s)>>>(qkv, out, dout, dqkv, p);
    else attn_bwd_mfma_tiled_kernel<false><<<(unsigned)nb, 64 * AM_WAVES, 0, as_stream(s)>>>(qkv, out, dout, dqkv, p);
    return wdno_check_launch();
  }
  rc = attn_amax_sweep(attn_bwd_rows(qkv, rot_cos, rot_sin, bias, out, dout, dqkv, part, p, d, s), dqkv, d, p.RW, amax_rec, s);
  return (rc || !dbias) ? rc : attn_dbias_reduce(part, attn_bwd_blocks(d), dbias, d, s);
}
// dqkv as fp16 planes for the qkv projection's gradient kernels (MFMA path only: n_tok <= 32)
extern "C" int wdno_attn_bwd_planes(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                                    const float* dout, void* dqkv_hi, void* dqkv_lo, float* dqkv_scale, float* dbias,
                                    const float* rec_qkv, const float* rec_dout, const wdno_attn_desc* d, float scale, void* ws, size_t ws_bytes,
                                    wdno_stream_t s) {
  AttnP p;
  int rc = attn_fill(p, d, scale, ATT_BWD_THREADS);
  if (rc) return rc;
  const int n = d->n_tok;
  if (n > 32 || wdno_debug_mode == 5 || !dqkv_hi || (dqkv_lo && (!dqkv_scale || !rec_qkv || !rec_dout))) return WDNO_EUNSUPPORTED;      // dqkv_lo == NULL: one bf16 plane, no scale
  if (dbias && (!ws || ws_bytes < wdno_attn_bwd_ws_bytes(d))) return WDNO_EWORKSPACE;
  float* part = dbias ? (float*)ws : nullptr;
  size_t lds2 = ((size_t)AM_WAVES * (2 * 32 * AM_TS + 32) + (dbias && d->heads != AM_WAVES ? (size_t)d->heads * n * n : 0)) * sizeof(float);      // (heads == AM_WAVES: the partial lives in registers)
  const int64_t nb = attn_bwd_blocks(d);
  p.pl_hi = (_Float16*)dqkv_hi; p.pl_lo = (_Float16*)dqkv_lo; p.rec_qkv = rec_qkv; p.rec_dout = rec_dout; p.pl_scale = dqkv_scale;
  (d->n_tok == 24 && wdno_debug_mode != 44 ? attn_bwd_mfma_kernel<24> : attn_bwd_mfma_kernel<0>)<<<(unsigned)nb, 64 * AM_WAVES, lds2, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, dout, nullptr, part, p);
  rc = wdno_check_launch();
  return (rc || !dbias) ? rc : attn_dbias_reduce(part, nb, dbias, d, s);
}
static int attn_bwd_rows(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                         const float* dout, float* dqkv, float* dbias, AttnP& p, const wdno_attn_desc* d, wdno_stream_t s) {
  const int n = d->n_tok;
  // 129 .. 576 tokens: nothing n x n is stored (attn_bwd_big_kernel) -- and 64 .. 128 tokens too when the kernel takes the case (no rotation, no bias):
  // the thread-per-row kernel keeps P and dS (2 n^2 floats: 81 KB at the mid block's 100 tokens) in LDS, one block per CU; 27 KB and five blocks per CU
  // here: 305 -> 246 us for the mid spatial attention of the smoke U-Net (debug 61: the old routing)
  if (n > ATT_BWD_THREADS || (wdno_debug_mode != 61 && n >= 64 && !rot_cos && !bias && !dbias)) {
    if (n > ATT_BIG_MAXTOK || rot_cos || bias || dbias) return WDNO_EUNSUPPORTED;
    const size_t lds = ((size_t)2 * n * DH + 3 * n) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)attn_bwd_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int64_t nb = p.n_items;
    const int64_t cap = 4LL * attn_num_cus();
    if (nb > cap) nb = cap;
    attn_bwd_big_kernel<<<(unsigned)nb, ATT_BIG_THREADS, lds, as_stream(s)>>>(qkv, out, dout, dqkv, p);
    return wdno_check_launch();
  }
  size_t lds = ((size_t)p.ipb * (2 * p.kst + 2 * n * (n + 1)) + (dbias ? (size_t)d->heads * n * n : 0)) * sizeof(float);
  if (lds > 160 * 1024) return WDNO_EUNSUPPORTED;
  int64_t groups = (p.n_items + p.ipb - 1) / p.ipb;
  int64_t blocks = groups;
  if (dbias && blocks > 1024) blocks = 1024;               // = attn_bwd_blocks(d): one dbias partial per block (grid-stride over the groups)
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_kernel<<<(unsigned)blocks, ATT_BWD_THREADS, lds, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, dout, dqkv, dbias, p);
  return wdno_check_launch();
}

// ================================================================================================ linear attention
// pass 1: per (unit, column = head*32+d): max over tokens and sum of exp -> kstats[unit][col][2].
// A unit has up to 1600 tokens but there are only ~200 units: the token range is cut into `chunks` pieces (one block each,
// online-softmax partials into a scratch area) and a second tiny launch merges them, so the sweep fills the chip.
__device__ __forceinline__ void softmax_merge(float& m, float& l, float m2, float l2) {
  float mn = fmaxf(m, m2);
  float a = (m == -INFINITY) ? 0.f : l * expf(m - mn);
  float b = (m2 == -INFINITY) ? 0.f : l2 * expf(m2 - mn);
  l = a + b; m = mn;
}
__global__ __launch_bounds__(256) void linattn_kstats_kernel(const float* __restrict__ qkv, float* __restrict__ part, int n, int HD, int chunks) {
  __shared__ float rm[256], rl[256];
  const int unit = blockIdx.x / chunks, chunk = blockIdx.x - unit * chunks;
  const int per = (n + chunks - 1) / chunks;
  const int jb = chunk * per, je = min(n, jb + per);
  const int col = threadIdx.x % HD, rg = threadIdx.x / HD, nrg = 256 / HD;
  const float* kp = qkv + ((int64_t)unit * n) * (3 * HD) + HD + col;
  float m = -INFINITY, l = 0.f;
  // eight rows per round: the loads are requested together and the running sum is rescaled once per round (row by row the
  // online softmax is a chain of load -> exp -> exp with one 512-byte request in flight per wave)
  for (int j = jb + rg; j < je; j += 8 * nrg) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = j + u * nrg < je ? kp[(int64_t)(j + u * nrg) * 3 * HD] : -INFINITY;
    float mn = m;
#pragma unroll
    for (int u = 0; u < 8; ++u) mn = fmaxf(mn, v[u]);
    float add = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) add += expf(v[u] - mn);        // exp(-inf) = 0 for the rows past the end
    l = (m == -INFINITY ? 0.f : l * expf(m - mn)) + add;
    m = mn;
  }
  rm[threadIdx.x] = m; rl[threadIdx.x] = l;
  __syncthreads();
  if (rg == 0) {
    for (int t = 1; t < nrg; ++t) softmax_merge(m, l, rm[t * HD + col], rl[t * HD + col]);
    part[((int64_t)blockIdx.x * HD + col) * 2 + 0] = m;
    part[((int64_t)blockIdx.x * HD + col) * 2 + 1] = l;
  }
}
__global__ __launch_bounds__(256) void linattn_kstats_merge_kernel(const float* __restrict__ part, float* __restrict__ kstats, int64_t cols, int HD, int chunks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (unit, col)
  if (i >= cols) return;
  const int64_t unit = i / HD;
  const int col = (int)(i - unit * HD);
  float m = -INFINITY, l = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float* q = part + (((unit * chunks + c) * HD) + col) * 2;
    softmax_merge(m, l, q[0], q[1]);
  }
  kstats[i * 2] = m;
  kstats[i * 2 + 1] = l;
}

// pass 2 (MODE 0): ctx[d][e]  = sum_n softmax_n(k)[n][d] * v[n][e]        A = exp(k - m)/l, B = v
//        (MODE 1): dctx[d][e] = sum_n qs[n][d] * dout[n][e]                A = scale*softmax_d(q), B = dout ; also T[d]
// one block (4 waves) per (unit, head); every MFMA consumes two tokens.
template <int MODE>
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const float* __restrict__ qkv, const float* __restrict__ other,
                                                           const float* __restrict__ kstats, const float* __restrict__ ctx_in,
                                                           float* __restrict__ ctx_out, float* __restrict__ tvec,
                                                           int n, int heads, float scale, int chunks = 1, float* __restrict__ part = nullptr) {
  // chunks > 1 (few (unit, head) pairs, many tokens: the Burgers U-Net has 16 x 4 of them over 4096 tokens -- 64 blocks on 256 CUs): the
  // token range is cut into `chunks` pieces, one block each, partial sums go to part[block][32][32] and linattn_ctx_merge_kernel adds them
  __shared__ float red[4][DH][KST];
  const int HD = heads * DH, RW = 3 * HD;
  const int uh = blockIdx.x / chunks, chunk = blockIdx.x - uh * chunks;
  const int unit = uh / heads, h = uh - unit * heads;
  const int clen = ((n + chunks - 1) / chunks + 7) & ~7;
  const int t0 = chunk * clen, t1 = min(n, t0 + clen);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, dd = lane & 31, hh = lane >> 5;
  float km = 0.f, kinvl = 1.f;
  if (MODE == 0) {
    km = kstats[((int64_t)unit * HD + h * DH + dd) * 2 + 0];
    kinvl = 1.0f / kstats[((int64_t)unit * HD + h * DH + dd) * 2 + 1];
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* base = qkv + ((int64_t)unit * n) * RW + h * DH + dd;
  const float* ob = MODE == 1 ? other + ((int64_t)unit * n) * HD + h * DH + dd : nullptr;
  // eight token pairs per round: all sixteen row loads of a round are requested before the first is used (one pair per
  // iteration leaves a single 256-byte request in flight per wave and the kernel at ~1.5 TB/s)
  constexpr int UN = 8;
  for (int j0 = t0 + wave * 2; j0 < t1; j0 += 8 * UN) {
    float ra[UN], rb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + 8 * u + hh;
      const bool ok = j < t1;
      if (MODE == 0) {
        ra[u] = ok ? base[(int64_t)j * RW + HD] : 0.f;
        rb[u] = ok ? base[(int64_t)j * RW + 2 * HD] : 0.f;
      } else {
        ra[u] = ok ? base[(int64_t)j * RW] : 0.f;
        rb[u] = ok ? ob[(int64_t)j * HD] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool ok = j0 + 8 * u + hh < t1;
      float a;
      if (MODE == 0) {
        a = ok ? expf(ra[u] - km) * kinvl : 0.f;
      } else {
        float mx = group_max<32>(ra[u]);
        float ex = expf(ra[u] - mx);
        float sm = group_sum<32>(ex);
        a = ok ? scale * ex / sm : 0.f;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, rb[u], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hh][dd] = acc[e];
  __syncthreads();
  // 1024 outputs / 256 threads
  if (chunks > 1) {
    float* po = part + (int64_t)blockIdx.x * DH * DH;
    for (int o = threadIdx.x; o < DH * DH; o += 256) {
      int r = o >> 5, c = o & 31;
      po[o] = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
    }
    return;
  }
  float* co = ctx_out + ((int64_t)unit * heads + h) * DH * DH;
  const float* ci = MODE == 1 ? ctx_in + ((int64_t)unit * heads + h) * DH * DH : nullptr;
  for (int o = threadIdx.x; o < DH * DH; o += 256) {
    int r = o >> 5, c = o & 31;
    float v = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
    co[o] = v;
    if (MODE == 1) red[0][r][c] = v * ci[o];   // each (r,c) is owned by exactly one thread
  }
  if (MODE == 1) {
    __syncthreads();
    if (threadIdx.x < DH) {
      float t = 0.f;
      for (int c = 0; c < DH; ++c) t += red[0][threadIdx.x][c];
      tvec[((int64_t)unit * heads + h) * DH + threadIdx.x] = t;
    }
  }
}

// sum of the chunk partials of linattn_ctx_kernel in chunk order; MODE 1 also T[d] = sum_e dctx[d][e] ctx[d][e]
template <int MODE>
__global__ __launch_bounds__(256) void linattn_ctx_merge_kernel(const float* __restrict__ part, const float* __restrict__ ctx_in,
                                                                 float* __restrict__ ctx_out, float* __restrict__ tvec, int chunks) {
  __shared__ float red[DH][KST];
  const int64_t uh = blockIdx.x;
  const float* p0 = part + uh * chunks * DH * DH;
  float* co = ctx_out + uh * DH * DH;
  const float* ci = MODE == 1 ? ctx_in + uh * DH * DH : nullptr;
  for (int o = threadIdx.x; o < DH * DH; o += 256) {
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += p0[c * DH * DH + o];
    co[o] = v;
    if (MODE == 1) red[o >> 5][o & 31] = v * ci[o];
  }
  if (MODE == 1) {
    __syncthreads();
    if (threadIdx.x < DH) {
      float t = 0.f;
      for (int c = 0; c < DH; ++c) t += red[threadIdx.x][c];
      tvec[uh * DH + threadIdx.x] = t;
    }
  }
}
// pieces of the token range per (unit, head): enough blocks for ~2 per CU, at least 128 tokens each, at most LA_MAXCHUNKS
#define LA_MAXCHUNKS 16
static int la_ctx_chunks(int64_t units, int heads, int n_tok) {
  const int64_t uh = units * heads;
  int64_t c = (2 * (int64_t)attn_num_cus() + uh - 1) / uh;
  if (c > n_tok / 128) c = n_tok / 128;
  if (c > LA_MAXCHUNKS) c = LA_MAXCHUNKS;
  if (c < 1 || wdno_debug_mode == 52) c = 1;                  // debug 52: one block per (unit, head) (A/B)
  return (int)c;
}
template <int MODE>
static void la_ctx_launch(const float* qkv, const float* other, const float* kstats, const float* ctx_in, float* ctx_out, float* tvec, float* part,
                          int64_t units, int n_tok, int heads, float scale, hipStream_t st) {
  const int chunks = part ? la_ctx_chunks(units, heads, n_tok) : 1;
  linattn_ctx_kernel<MODE><<<(unsigned)(units * heads * chunks), 256, 0, st>>>(qkv, other, kstats, ctx_in, ctx_out, tvec, n_tok, heads, scale, chunks, part);
  if (chunks > 1) linattn_ctx_merge_kernel<MODE><<<(unsigned)(units * heads), 256, 0, st>>>(part, ctx_in, ctx_out, tvec, chunks);
}

// pass 3: out[n][e] = scale * sum_d ctx[d][e] * softmax_d(q[n])[d]; one thread per (token, head)
__global__ __launch_bounds__(256) void linattn_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx, float* __restrict__ out,
                                   int n, int heads, float scale, float* __restrict__ amax_rec) {
  extern __shared__ __attribute__((aligned(16))) float cs[];   // [heads][32][32]
  const int HD = heads * DH, RW = 3 * HD;
  const int unit = blockIdx.x;
  const int h = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < heads * DH * DH; e += blockDim.x) cs[e] = ctx[(int64_t)unit * heads * DH * DH + e];
  __syncthreads();
  const int jt = blockIdx.y * 64 + (threadIdx.x & 63);
  const bool ok = jt < n;
  const int j = ok ? jt : n - 1;           // lanes past the last token redo it (all lanes stay active for the amax reduction) and store nothing
  const float4* qp = reinterpret_cast<const float4*>(qkv + ((int64_t)unit * n + j) * RW + h * DH);
  float q[DH];
#pragma unroll
  for (int e = 0; e < 8; ++e) { float4 v = qp[e]; q[4 * e] = v.x; q[4 * e + 1] = v.y; q[4 * e + 2] = v.z; q[4 * e + 3] = v.w; }
  float mx = q[0];
#pragma unroll
  for (int e = 1; e < DH; ++e) mx = fmaxf(mx, q[e]);
  float sm = 0.f;
#pragma unroll
  for (int e = 0; e < DH; ++e) { q[e] = expf(q[e] - mx); sm += q[e]; }
  float f = scale / sm;
  float o[DH];
#pragma unroll
  for (int e = 0; e < DH; ++e) o[e] = 0.f;
  const float4* cp = reinterpret_cast<const float4*>(cs + h * DH * DH);
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    float w = q[d] * f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float4 c = cp[d * 8 + e];
      o[4 * e] = fmaf(w, c.x, o[4 * e]); o[4 * e + 1] = fmaf(w, c.y, o[4 * e + 1]);
      o[4 * e + 2] = fmaf(w, c.z, o[4 * e + 2]); o[4 * e + 3] = fmaf(w, c.w, o[4 * e + 3]);
    }
  }
  float4* op = reinterpret_cast<float4*>(out + ((int64_t)unit * n + j) * HD + h * DH);
  float am = 0.f;
  if (ok) {
#pragma unroll
  for (int e = 0; e < 8; ++e) op[e] = make_float4(o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]);
  }
#pragma unroll
  for (int e = 0; e < DH; ++e) am = fmaxf(am, fabsf(o[e]));
  if (amax_rec) wave_amax_emit(am, amax_rec, (int)((blockIdx.y * gridDim.x + blockIdx.x) * heads + h));
}

// backward pass B2: one thread per (token, head):
//   dq = scale * qsm * (dqs - <qsm, dqs>),  dqs[d] = sum_e dout[e] ctx[d][e]
//   dv[e] = sum_d ks[d] dctx[d][e] ; dk[d] = ks[d] * (sum_e v[e] dctx[d][e] - T[d])
__global__ __launch_bounds__(256) void linattn_bwd_tok_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ kstats,
                                       const float* __restrict__ ctx, const float* __restrict__ dctx, const float* __restrict__ tvec,
                                       float* __restrict__ dqkv, int n, int heads, float scale, float* __restrict__ amax_rec) {
  extern __shared__ __attribute__((aligned(16))) float sm_[];
  const int HD = heads * DH, RW = 3 * HD;
  float* cs = sm_;                       // [heads][32][32]
  float* ds = cs + heads * DH * DH;      // [heads][32][32]
  float* km = ds + heads * DH * DH;      // [heads*32]
  float* kl = km + HD;                   // 1/l
  float* tv = kl + HD;
  const int unit = blockIdx.x;
  const int h = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < heads * DH * DH; e += blockDim.x) {
    cs[e] = ctx[(int64_t)unit * heads * DH * DH + e];
    ds[e] = dctx[(int64_t)unit * heads * DH * DH + e];
  }
  for (int e = threadIdx.x; e < HD; e += blockDim.x) {
    km[e] = kstats[((int64_t)unit * HD + e) * 2];
    kl[e] = 1.0f / kstats[((int64_t)unit * HD + e) * 2 + 1];
    tv[e] = tvec[(int64_t)unit * HD + e];
  }
  __shared__ unsigned blk_amax;
  if (threadIdx.x == 0) blk_amax = 0u;
  __syncthreads();
  const int j = blockIdx.y * 64 + (threadIdx.x & 63);
  if (j >= n) return;                       // (a finished wave no longer counts at the barrier below)
  float am = 0.f;
  const int64_t row = (int64_t)unit * n + j;
  const float4* cp = reinterpret_cast<const float4*>(cs + h * DH * DH);
  const float4* dp = reinterpret_cast<const float4*>(ds + h * DH * DH);
  float* wq = dqkv + row * RW + h * DH;
  {  // ---- dq
    float q[DH], go[DH];
    const float4* qp = reinterpret_cast<const float4*>(qkv + row * RW + h * DH);
    const float4* gp = reinterpret_cast<const float4*>(dout + row * HD + h * DH);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float4 v = qp[e]; q[4 * e] = v.x; q[4 * e + 1] = v.y; q[4 * e + 2] = v.z; q[4 * e + 3] = v.w;
      float4 g = gp[e]; go[4 * e] = g.x; go[4 * e + 1] = g.y; go[4 * e + 2] = g.z; go[4 * e + 3] = g.w;
    }
    float mx = q[0];
#pragma unroll
    for (int e = 1; e < DH; ++e) mx = fmaxf(mx, q[e]);
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < DH; ++e) { q[e] = expf(q[e] - mx); sm += q[e]; }
    float inv = 1.0f / sm, dot = 0.f;
    float dqs[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float4 c = cp[d * 8 + e];
        a = fmaf(go[4 * e], c.x, a); a = fmaf(go[4 * e + 1], c.y, a); a = fmaf(go[4 * e + 2], c.z, a); a = fmaf(go[4 * e + 3], c.w, a);
      }
      q[d] *= inv;
      dqs[d] = a;
      dot = fmaf(q[d], a, dot);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float4 v = make_float4(scale * q[4 * e] * (dqs[4 * e] - dot), scale * q[4 * e + 1] * (dqs[4 * e + 1] - dot),
                                   scale * q[4 * e + 2] * (dqs[4 * e + 2] - dot), scale * q[4 * e + 3] * (dqs[4 * e + 3] - dot));
      reinterpret_cast<float4*>(wq)[e] = v;
      am = amax4(am, v);
    }
  }
  {  // ---- dk, dv
    float ks[DH], vv[DH], dv[DH];
    const float4* kp = reinterpret_cast<const float4*>(qkv + row * RW + HD + h * DH);
    const float4* vp = reinterpret_cast<const float4*>(qkv + row * RW + 2 * HD + h * DH);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float4 v = kp[e]; ks[4 * e] = v.x; ks[4 * e + 1] = v.y; ks[4 * e + 2] = v.z; ks[4 * e + 3] = v.w;
      float4 u = vp[e]; vv[4 * e] = u.x; vv[4 * e + 1] = u.y; vv[4 * e + 2] = u.z; vv[4 * e + 3] = u.w;
    }
#pragma unroll
    for (int e = 0; e < DH; ++e) { ks[e] = expf(ks[e] - km[h * DH + e]) * kl[h * DH + e]; dv[e] = 0.f; }
    float dk[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      float a = 0.f, w = ks[d];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float4 c = dp[d * 8 + e];
        a = fmaf(vv[4 * e], c.x, a); a = fmaf(vv[4 * e + 1], c.y, a); a = fmaf(vv[4 * e + 2], c.z, a); a = fmaf(vv[4 * e + 3], c.w, a);
        dv[4 * e] = fmaf(w, c.x, dv[4 * e]); dv[4 * e + 1] = fmaf(w, c.y, dv[4 * e + 1]);
        dv[4 * e + 2] = fmaf(w, c.z, dv[4 * e + 2]); dv[4 * e + 3] = fmaf(w, c.w, dv[4 * e + 3]);
      }
      dk[d] = w * (a - tv[h * DH + d]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float4 gk = make_float4(dk[4 * e], dk[4 * e + 1], dk[4 * e + 2], dk[4 * e + 3]);
      const float4 gv = make_float4(dv[4 * e], dv[4 * e + 1], dv[4 * e + 2], dv[4 * e + 3]);
      reinterpret_cast<float4*>(wq + HD)[e] = gk;
      reinterpret_cast<float4*>(wq + 2 * HD)[e] = gv;
      am = amax4(amax4(am, gk), gv);
    }
  }
  if (amax_rec) {     // lanes past the last token have left: an LDS atomic per lane instead of a wave shuffle, one global atomic per block
    atomicMax(&blk_amax, __float_as_uint(am));
    __syncthreads();
    if (threadIdx.x == 0)
      atomicMax(reinterpret_cast<unsigned*>(amax_rec) + ((blockIdx.y * gridDim.x + blockIdx.x) & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE, blk_amax);
  }
}

// The same backward pass on MFMA tiles: one wave per (unit, head, 32-token chunk), the four waves of a block take four
// consecutive chunks of one (unit, head) and share its ctx / dctx tiles. The thread-per-token kernel above needs 256 + 182
// registers (one wave per SIMD) and every lane walks its own 1.5-KB-strided rows; here the rows are staged through LDS by
// coalesced 128-byte reads (am_stage_rows) and the three 32 x 32 x 32 products run on v_mfma_f32_32x32x2_f32, computed
// TRANSPOSED so that lane (token, half) ends up with 16 of the token's 32 channels (runs of four: 16-byte stores):
//   dqs^T[d][t] = sum_e ctx[d][e]  dout[t][e]       a^T[d][t] = sum_e dctx[d][e] v[t][e]       dv^T[e][t] = sum_d dctx[d][e] ks[t][d]
// The channel softmax of q and the <qsm, dqs> dot product need the other half of the row: one lane ^ 32 exchange each.
#define LAM_TILE (32 * AM_TS)
__global__ __launch_bounds__(256, 2) void linattn_bwd_tok_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                       const float* __restrict__ kstats, const float* __restrict__ ctx,
                                                                       const float* __restrict__ dctx, const float* __restrict__ tvec,
                                                                       float* __restrict__ dqkv, int n, int heads, float scale,
                                                                       float* __restrict__ amax_rec, _Float16* __restrict__ pl_hi = nullptr,
                                                                       _Float16* __restrict__ pl_lo = nullptr, const float* __restrict__ rec_qkv = nullptr,
                                                                       const float* __restrict__ rec_dout = nullptr,
                                                                       const float* __restrict__ rec_dctx = nullptr, float* __restrict__ pl_scale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  float* Tc = lsm;                       // ctx  [d][e]
  float* Td = Tc + LAM_TILE;             // dctx [d][e]
  float* km = Td + LAM_TILE;             // per d: max, 1 / sum of the token softmax of k; T[d]
  float* kl = km + DH;
  float* tv = kl + DH;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
  float* T1 = tv + DH + wave * 3 * LAM_TILE;      // q, then dout
  float* T2 = T1 + LAM_TILE;                      // k
  float* T3 = T2 + LAM_TILE;                      // v
  const int HD = heads * DH, RW = 3 * HD;
  const int unit = blockIdx.x / heads, h = blockIdx.x - unit * heads;
  {
    const float* cs = ctx + ((int64_t)unit * heads + h) * DH * DH;
    const float* ds = dctx + ((int64_t)unit * heads + h) * DH * DH;
    for (int e = threadIdx.x; e < DH * DH / 4; e += 256) {
      const int r = e >> 3, c4 = (e & 7) * 4;
      *reinterpret_cast<float4*>(Tc + r * AM_TS + c4) = reinterpret_cast<const float4*>(cs)[e];
      *reinterpret_cast<float4*>(Td + r * AM_TS + c4) = reinterpret_cast<const float4*>(ds)[e];
    }
    if (threadIdx.x < DH) {
      const int64_t col = (int64_t)unit * HD + h * DH + threadIdx.x;
      km[threadIdx.x] = kstats[col * 2];
      kl[threadIdx.x] = 1.0f / kstats[col * 2 + 1];
      tv[threadIdx.x] = tvec[col];
    }
  }
  __syncthreads();
  float am = 0.f;
  // Planes output (wdno_linattn_bwd_planes): with A = max|qkv|, G = max|dout|, D = max|dctx| (measured), ks, qsm <= 1 and |ctx| <= A:
  // |dq| <= 2 scale 32 A G,  |dk| <= 2 * 32 A D,  |dv| <= 32 D.
  float ps = 1.0f;
  if (pl_hi && pl_lo) {
    const float A = amax_record_read(rec_qkv), G = amax_record_read(rec_dout), D = amax_record_read(rec_dctx);
    ps = scale_from_amax(1.01f * fmaxf(64.f * scale * A * G, fmaxf(64.f * A * D, 32.f * D)));
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) pl_scale[0] = ps;
  }
  const int t0 = (blockIdx.y * 4 + wave) * 32;
  if (t0 < n) {                                     // (no block-wide barrier below)
    const int nv = min(32, n - t0);
    const bool tok = li < nv;
    const int64_t row0 = (int64_t)unit * n + t0;
    const float* qb = am_uniform(qkv + row0 * RW + h * DH);
    const float* gb = am_uniform(dout + row0 * HD + h * DH);
    float* db = const_cast<float*>(am_uniform(dqkv + row0 * RW + h * DH)) + (unsigned)(tok ? li : 0) * (unsigned)RW;
    const int64_t pbase = row0 * RW + h * DH + (int64_t)((unsigned)(tok ? li : 0) * (unsigned)RW);
    am_stage_rows(T1, qb, (unsigned)RW, nullptr, nullptr, 1.0f, nv, lane);
    am_stage_rows(T2, qb + HD, (unsigned)RW, nullptr, nullptr, 1.0f, nv, lane);
    am_stage_rows(T3, qb + 2 * HD, (unsigned)RW, nullptr, nullptr, 1.0f, nv, lane);
    __builtin_amdgcn_wave_barrier();
    // channel softmax of this lane's token: channels d = 8*e4 + 4*hh + c here, the other sixteen in lane ^ 32
    float qsm[16];
    {
      float mx = -INFINITY;
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const float4 t = *reinterpret_cast<const float4*>(T1 + li * AM_TS + 8 * e4 + 4 * hh);
        qsm[4 * e4] = t.x; qsm[4 * e4 + 1] = t.y; qsm[4 * e4 + 2] = t.z; qsm[4 * e4 + 3] = t.w;
        mx = fmaxf(fmaxf(mx, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sm = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { qsm[e] = expf(qsm[e] - mx); sm += qsm[e]; }
      sm += __shfl_xor(sm, 32);
      const float inv = 1.0f / sm;
#pragma unroll
      for (int e = 0; e < 16; ++e) qsm[e] *= inv;
    }
    __builtin_amdgcn_wave_barrier();                // every lane has read its q row: the tile takes dout now
    am_stage_rows(T1, gb, (unsigned)HD, nullptr, nullptr, 1.0f, nv, lane);
    __builtin_amdgcn_wave_barrier();
    {   // ---- dq = scale * qsm * (dqs - <qsm, dqs>)
      float ca[16], ga[16];
      am_sel(Tc, li, hh, ca);
      am_sel(T1, li, hh, ga);
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[m], ga[m], acc, 0, 0, 0);
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) dot = fmaf(qsm[e], acc[e], dot);
      dot += __shfl_xor(dot, 32);
      float4 vv[4];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const float4 v = make_float4(scale * qsm[4 * e4] * (acc[4 * e4] - dot), scale * qsm[4 * e4 + 1] * (acc[4 * e4 + 1] - dot),
                                     scale * qsm[4 * e4 + 2] * (acc[4 * e4 + 2] - dot), scale * qsm[4 * e4 + 3] * (acc[4 * e4 + 3] - dot));
        vv[e4] = v;
        if (tok && !pl_hi) *reinterpret_cast<float4*>(db + 8 * e4 + 4 * hh) = v;
        am = amax4(am, v);
      }
      if (tok && pl_hi) am_store_row_planes(pl_hi, pl_lo, pbase, hh, vv, ps);
    }
    {   // ---- dk[d] = ks[d] * (sum_e v[e] dctx[d][e] - T[d]),  ks = exp(k - max) / sum
      float da[16], va[16];
      am_sel(Td, li, hh, da);
      am_sel(T3, li, hh, va);
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[m], va[m], acc, 0, 0, 0);
      float4 vk[4];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int d0 = 8 * e4 + 4 * hh;
        const float4 kk = *reinterpret_cast<const float4*>(T2 + li * AM_TS + d0);
        const float4 m4 = *reinterpret_cast<const float4*>(km + d0), l4 = *reinterpret_cast<const float4*>(kl + d0);
        const float4 t4 = *reinterpret_cast<const float4*>(tv + d0);
        const float4 v = make_float4(expf(kk.x - m4.x) * l4.x * (acc[4 * e4] - t4.x), expf(kk.y - m4.y) * l4.y * (acc[4 * e4 + 1] - t4.y),
                                     expf(kk.z - m4.z) * l4.z * (acc[4 * e4 + 2] - t4.z), expf(kk.w - m4.w) * l4.w * (acc[4 * e4 + 3] - t4.w));
        vk[e4] = v;
        if (tok && !pl_hi) *reinterpret_cast<float4*>(db + HD + d0) = v;
        am = amax4(am, v);
      }
      if (tok && pl_hi) am_store_row_planes(pl_hi, pl_lo, pbase + HD, hh, vk, ps);
    }
    {   // ---- dv[e] = sum_d ks[d] dctx[d][e]: row operand = dctx read by columns (lane = e), column operand = ks[t][2m + hh]
      float da[16], ka[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int d = AM_D(m, hh);
        da[m] = Td[d * AM_TS + li];
        ka[m] = expf(T2[li * AM_TS + d] - km[d]) * kl[d];
      }
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[m], ka[m], acc, 0, 0, 0);
      float4 vd[4];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const float4 v = make_float4(acc[4 * e4], acc[4 * e4 + 1], acc[4 * e4 + 2], acc[4 * e4 + 3]);
        vd[e4] = v;
        if (tok && !pl_hi) *reinterpret_cast<float4*>(db + 2 * HD + 8 * e4 + 4 * hh) = v;
        am = amax4(am, v);
      }
      if (tok && pl_hi) am_store_row_planes(pl_hi, pl_lo, pbase + 2 * HD, hh, vd, ps);
    }
  }
  if (amax_rec) wave_amax_emit(am, amax_rec, (int)((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave));
}

// pass 3 on MFMA tiles (same layout as linattn_bwd_tok_mfma_kernel): out^T[e][t] = sum_d ctx[d][e] qs[t][d], qs = scale * softmax_d(q).
// Row operand = ctx read by columns (lane = e), column operand = qs[t][2m + hh]; the two halves of a token's softmax sit in
// lanes t and t + 32.
__global__ __launch_bounds__(256, 2) void linattn_out_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                                   float* __restrict__ out, int n, int heads, float scale,
                                                                   float* __restrict__ amax_rec, _Float16* __restrict__ pl_hi = nullptr,
                                                                   _Float16* __restrict__ pl_lo = nullptr, const float* __restrict__ rec_qkv = nullptr,
                                                                   float* __restrict__ pl_scale = nullptr) {
  __shared__ __attribute__((aligned(16))) float Tc[LAM_TILE];
  __shared__ __attribute__((aligned(16))) float Tq[4][LAM_TILE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
  const int HD = heads * DH, RW = 3 * HD;
  const int unit = blockIdx.x / heads, h = blockIdx.x - unit * heads;
  {
    const float* cs = ctx + ((int64_t)unit * heads + h) * DH * DH;
    for (int e = threadIdx.x; e < DH * DH / 4; e += 256) {
      const int r = e >> 3, c4 = (e & 7) * 4;
      *reinterpret_cast<float4*>(Tc + r * AM_TS + c4) = reinterpret_cast<const float4*>(cs)[e];
    }
  }
  __syncthreads();
  float ops = 1.0f;
  if (pl_hi && pl_lo) {
    ops = scale_from_amax(1.01f * scale * amax_record_read(rec_qkv));
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) pl_scale[0] = ops;
  }
  float am = 0.f;
  const int t0 = (blockIdx.y * 4 + wave) * 32;
  if (t0 < n) {
    const int nv = min(32, n - t0);
    const bool tok = li < nv;
    const int64_t row0 = (int64_t)unit * n + t0;
    float* T1 = Tq[wave];
    am_stage_rows(T1, am_uniform(qkv + row0 * RW + h * DH), (unsigned)RW, nullptr, nullptr, 1.0f, nv, lane);
    __builtin_amdgcn_wave_barrier();
    float qs[16], ca[16];
    am_sel(T1, li, hh, qs);
    float mx = qs[0];
#pragma unroll
    for (int m = 1; m < 16; ++m) mx = fmaxf(mx, qs[m]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sm = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) { qs[m] = expf(qs[m] - mx); sm += qs[m]; }
    sm += __shfl_xor(sm, 32);
    const float f = scale / sm;
#pragma unroll
    for (int m = 0; m < 16; ++m) { qs[m] *= f; ca[m] = Tc[AM_D(m, hh) * AM_TS + li]; }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[m], qs[m], acc, 0, 0, 0);
    float* orow = out + (row0 + (tok ? li : 0)) * HD + h * DH;
    float4 vo[4];
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const float4 v = make_float4(acc[4 * e4], acc[4 * e4 + 1], acc[4 * e4 + 2], acc[4 * e4 + 3]);
      vo[e4] = v;
      if (tok && !pl_hi) *reinterpret_cast<float4*>(orow + 8 * e4 + 4 * hh) = v;
      am = amax4(am, v);
    }
    // planes for the to_out projection INSTEAD of the fp32 tensor (the backward works from ctx, not from out):
    // qs sums to `scale` over d and |ctx| <= max|v|, so |out| <= scale * max|qkv|
    if (tok && pl_hi) am_store_row_planes(pl_hi, pl_lo, (row0 + li) * HD + h * DH, hh, vo, ops);
  }
  if (amax_rec) wave_amax_emit(am, amax_rec, (int)((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave));
}

static int la_check(int64_t units, int n, int heads) {
  if (units <= 0 || n <= 0 || heads <= 0 || units > 0x7fffffff / 8) return WDNO_EINVAL;
  if (heads != 1 && heads != 2 && heads != 4) return WDNO_EUNSUPPORTED;   // 64*heads threads per block
  return WDNO_OK;
}
// backward workspace: dctx [units,heads,32,32] + T [units,heads,32]
extern "C" size_t wdno_linattn_ws_bytes(int64_t units, int heads) {
  const size_t chunk_part = units * heads <= 2 * (int64_t)attn_num_cus() ? (size_t)units * heads * LA_MAXCHUNKS * DH * DH : 0;      // partial contexts of a chunked token range
  return ((size_t)units * heads * DH * DH + (size_t)units * heads * DH + chunk_part) * sizeof(float);
}
extern "C" int wdno_linattn_fwd(const float* qkv, float* out, float* kstats, float* ctx, int64_t units, int n_tok, int heads,
                                float scale, wdno_stream_t s) {
  return wdno_linattn_fwd_amax(qkv, out, kstats, ctx, nullptr, units, n_tok, heads, scale, s);
}
extern "C" int wdno_linattn_fwd_amax(const float* qkv, float* out, float* kstats, float* ctx, float* amax_rec, int64_t units, int n_tok,
                                     int heads, float scale, wdno_stream_t s) {
  int rc = la_check(units, n_tok, heads);
  if (rc) return rc;
  hipStream_t st = as_stream(s);
  // partial statistics go through `ctx` (written only afterwards by the context kernel): chunks * HD * 2 <= 32 * HD floats per unit
  int chunks = n_tok / 64;
  chunks = chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks);
  if (chunks == 1) {
    linattn_kstats_kernel<<<(unsigned)units, 256, 0, st>>>(qkv, kstats, n_tok, heads * DH, 1);
  } else {
    linattn_kstats_kernel<<<(unsigned)(units * chunks), 256, 0, st>>>(qkv, ctx, n_tok, heads * DH, chunks);
    const int64_t cols = units * heads * DH;
    linattn_kstats_merge_kernel<<<(unsigned)cdiv64(cols, 256), 256, 0, st>>>(ctx, kstats, cols, heads * DH, chunks);
  }
  // (chunk partials of the context go through `out`, which only the output kernel below writes: units * heads * chunks * 1024 floats
  // <= units * n_tok * HD because chunks <= n_tok / 128)
  la_ctx_launch<0>(qkv, nullptr, kstats, nullptr, ctx, nullptr, out, units, n_tok, heads, scale, st);
  if (wdno_debug_mode != 5) {            // debug 5: the thread-per-token kernel
    linattn_out_mfma_kernel<<<dim3((unsigned)(units * heads), (unsigned)cdiv(n_tok, 128)), 256, 0, st>>>(qkv, ctx, out, n_tok, heads, scale, amax_rec);
    return wdno_check_launch();
  }
  size_t lds = (size_t)heads * DH * DH * sizeof(float);
  linattn_out_kernel<<<dim3((unsigned)units, (unsigned)cdiv(n_tok, 64)), 64 * heads, lds, st>>>(qkv, ctx, out, n_tok, heads, scale, amax_rec);
  return wdno_check_launch();
}
extern "C" int wdno_linattn_bwd(const float* qkv, const float* dout, const float* kstats, const float* ctx, float* dqkv,
                                void* ws, size_t ws_bytes, int64_t units, int n_tok, int heads, float scale, wdno_stream_t s) {
  return wdno_linattn_bwd_amax(qkv, dout, kstats, ctx, dqkv, nullptr, ws, ws_bytes, units, n_tok, heads, scale, s);
}
extern "C" int wdno_linattn_bwd_amax(const float* qkv, const float* dout, const float* kstats, const float* ctx, float* dqkv,
                                     float* amax_rec, void* ws, size_t ws_bytes, int64_t units, int n_tok, int heads, float scale,
                                     wdno_stream_t s) {
  int rc = la_check(units, n_tok, heads);
  if (rc) return rc;
  if (ws_bytes < wdno_linattn_ws_bytes(units, heads)) return WDNO_EWORKSPACE;
  hipStream_t st = as_stream(s);
  float* dctx = (float*)ws;
  float* tvec = dctx + (size_t)units * heads * DH * DH;
  float* cpart = ws_bytes >= wdno_linattn_ws_bytes(units, heads) && units * heads <= 2 * (int64_t)attn_num_cus() ? tvec + (size_t)units * heads * DH : nullptr;
  la_ctx_launch<1>(qkv, dout, nullptr, ctx, dctx, tvec, cpart, units, n_tok, heads, scale, st);
  if (wdno_debug_mode != 5) {            // debug 5: the thread-per-token kernel
    const size_t lds2 = ((size_t)(2 + 4 * 3) * LAM_TILE + 3 * DH) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)linattn_bwd_tok_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); attr_done = true; }
    linattn_bwd_tok_mfma_kernel<<<dim3((unsigned)(units * heads), (unsigned)cdiv(n_tok, 128)), 256, lds2, st>>>(qkv, dout, kstats, ctx, dctx, tvec,
                                                                                                          dqkv, n_tok, heads, scale, amax_rec);
    return wdno_check_launch();
  }
  size_t lds = ((size_t)2 * heads * DH * DH + 3 * heads * DH) * sizeof(float);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)linattn_bwd_tok_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  linattn_bwd_tok_kernel<<<dim3((unsigned)units, (unsigned)cdiv(n_tok, 64)), 64 * heads, lds, st>>>(qkv, dout, kstats, ctx, dctx, tvec, dqkv,
                                                                                                n_tok, heads, scale, amax_rec);
  return wdno_check_launch();
}

// dqkv as fp16 planes for the gradient kernels of the qkv projection (MFMA token kernel only). rec_dctx: a zeroed amax record, receives
// max|dctx| (a sweep over the small dctx tensor between the two launches).
extern "C" int wdno_linattn_bwd_planes(const float* qkv, const float* dout, const float* kstats, const float* ctx, void* dqkv_hi,
                                       void* dqkv_lo, float* dqkv_scale, const float* rec_qkv, const float* rec_dout, float* rec_dctx,
                                       void* ws, size_t ws_bytes, int64_t units, int n_tok, int heads, float scale, wdno_stream_t s) {
  int rc = la_check(units, n_tok, heads);
  if (rc) return rc;
  if (!dqkv_hi || (dqkv_lo && (!dqkv_scale || !rec_qkv || !rec_dout || !rec_dctx))) return WDNO_EINVAL;      // dqkv_lo == NULL: one bf16 plane
  if (ws_bytes < wdno_linattn_ws_bytes(units, heads)) return WDNO_EWORKSPACE;
  hipStream_t st = as_stream(s);
  float* dctx = (float*)ws;
  float* tvec = dctx + (size_t)units * heads * DH * DH;
  float* cpart = ws_bytes >= wdno_linattn_ws_bytes(units, heads) && units * heads <= 2 * (int64_t)attn_num_cus() ? tvec + (size_t)units * heads * DH : nullptr;
  la_ctx_launch<1>(qkv, dout, nullptr, ctx, dctx, tvec, cpart, units, n_tok, heads, scale, st);
  if (dqkv_lo) {
    rc = wdno_amax_record(dctx, (int64_t)units * heads * DH * DH, rec_dctx, s);
    if (rc) return rc;
  }
  const size_t lds2 = ((size_t)(2 + 4 * 3) * LAM_TILE + 3 * DH) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)linattn_bwd_tok_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); attr_done = true; }
  linattn_bwd_tok_mfma_kernel<<<dim3((unsigned)(units * heads), (unsigned)cdiv(n_tok, 128)), 256, lds2, st>>>(
      qkv, dout, kstats, ctx, dctx, tvec, nullptr, n_tok, heads, scale, nullptr, (_Float16*)dqkv_hi, (_Float16*)dqkv_lo, rec_qkv, rec_dout, rec_dctx,
      dqkv_scale);
  return wdno_check_launch();
}

// forward with out delivered as fp16 planes only (MFMA output kernel)
extern "C" int wdno_linattn_fwd_planes(const float* qkv, void* out_hi, void* out_lo, float* out_scale, float* kstats, float* ctx,
                                       const float* rec_qkv, int64_t units, int n_tok, int heads, float scale, wdno_stream_t s) {
  int rc = la_check(units, n_tok, heads);
  if (rc) return rc;
  if (!out_hi || (out_lo && (!out_scale || !rec_qkv))) return WDNO_EINVAL;      // out_lo == NULL: one bf16 plane
  hipStream_t st = as_stream(s);
  int chunks = n_tok / 64;
  chunks = chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks);
  if (chunks == 1) {
    linattn_kstats_kernel<<<(unsigned)units, 256, 0, st>>>(qkv, kstats, n_tok, heads * DH, 1);
  } else {
    linattn_kstats_kernel<<<(unsigned)(units * chunks), 256, 0, st>>>(qkv, ctx, n_tok, heads * DH, chunks);
    const int64_t cols = units * heads * DH;
    linattn_kstats_merge_kernel<<<(unsigned)cdiv64(cols, 256), 256, 0, st>>>(ctx, kstats, cols, heads * DH, chunks);
  }
  // (chunk partials go through the hi plane: units * heads * chunks * 4096 bytes <= units * n_tok * HD * 2 because chunks <= n_tok / 128)
  la_ctx_launch<0>(qkv, nullptr, kstats, nullptr, ctx, nullptr, (float*)out_hi, units, n_tok, heads, scale, st);
  linattn_out_mfma_kernel<<<dim3((unsigned)(units * heads), (unsigned)cdiv(n_tok, 128)), 256, 0, st>>>(
      qkv, ctx, nullptr, n_tok, heads, scale, nullptr, (_Float16*)out_hi, (_Float16*)out_lo, rec_qkv, out_scale);
  return wdno_check_launch();
}
