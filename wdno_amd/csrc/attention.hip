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
};

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
  if (dbias) {
    __syncthreads();
    for (int e = threadIdx.x; e < p.d.heads * n * n; e += ATT_BWD_THREADS)
      if (dBs[e] != 0.f) atomicAdd(&dbias[e], dBs[e]);
  }
}

static int attn_fill(AttnP& p, const wdno_attn_desc* d, float scale, int threads) {
  if (!d || d->n_uo <= 0 || d->n_ui <= 0 || d->n_tok <= 0 || d->heads <= 0) return WDNO_EINVAL;
  p.d = *d; p.scale = scale; p.HD = d->heads * DH; p.RW = 3 * p.HD;
  p.ipb = threads / d->n_tok;
  if (p.ipb < 1) p.ipb = 1;
  p.kst = d->n_tok * DH + 8;          // +8 floats: items start on different 32-byte LDS slots
  p.n_items = (int64_t)d->n_uo * d->n_ui * d->heads;
  return WDNO_OK;
}
extern "C" int wdno_attn_fwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out,
                             const wdno_attn_desc* d, float scale, wdno_stream_t s) {
  AttnP p;
  int rc = attn_fill(p, d, scale, ATT_THREADS);
  if (rc) return rc;
  if (d->n_tok > 1024) return WDNO_EUNSUPPORTED;
  size_t lds = (size_t)p.ipb * p.kst * sizeof(float);
  if (lds > 160 * 1024) return WDNO_EUNSUPPORTED;
  int64_t blocks = (p.n_items + p.ipb - 1) / p.ipb;
  if (blocks > 0x7fffffff) return WDNO_EUNSUPPORTED;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_fwd_kernel<<<(unsigned)blocks, ATT_THREADS, lds, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, p);
  return wdno_check_launch();
}
extern "C" int wdno_attn_bwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                             const float* dout, float* dqkv, float* dbias, const wdno_attn_desc* d, float scale, wdno_stream_t s) {
  AttnP p;
  int rc = attn_fill(p, d, scale, ATT_BWD_THREADS);
  if (rc) return rc;
  const int n = d->n_tok;
  if (n > ATT_BWD_THREADS) return WDNO_EUNSUPPORTED;      // training never attends over more than 100 tokens
  size_t lds = ((size_t)p.ipb * (2 * p.kst + 2 * n * (n + 1)) + (dbias ? (size_t)d->heads * n * n : 0)) * sizeof(float);
  if (lds > 160 * 1024) return WDNO_EUNSUPPORTED;
  int64_t groups = (p.n_items + p.ipb - 1) / p.ipb;
  int64_t blocks = groups;
  if (dbias && blocks > 1024) blocks = 1024;               // bound the number of global atomic flushes
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
  for (int j = jb + rg; j < je; j += nrg) {
    float v = kp[(int64_t)j * 3 * HD];
    float mn = fmaxf(m, v);
    l = l * expf(m - mn) + expf(v - mn);
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
                                                           int n, int heads, float scale) {
  __shared__ float red[4][DH][KST];
  const int HD = heads * DH, RW = 3 * HD;
  const int unit = blockIdx.x / heads, h = blockIdx.x - unit * heads;
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
  for (int j0 = wave * 2; j0 < n; j0 += 8) {
    int j = j0 + hh;
    bool ok = j < n;
    float a, b;
    if (MODE == 0) {
      float kv = ok ? base[(int64_t)j * RW + HD] : 0.f;
      a = ok ? expf(kv - km) * kinvl : 0.f;
      b = ok ? base[(int64_t)j * RW + 2 * HD] : 0.f;
    } else {
      float qv = ok ? base[(int64_t)j * RW] : 0.f;
      float mx = group_max<32>(qv);
      float ex = expf(qv - mx);
      float sm = group_sum<32>(ex);
      a = ok ? scale * ex / sm : 0.f;
      b = ok ? ob[(int64_t)j * HD] : 0.f;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hh][dd] = acc[e];
  __syncthreads();
  // 1024 outputs / 256 threads
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

// pass 3: out[n][e] = scale * sum_d ctx[d][e] * softmax_d(q[n])[d]; one thread per (token, head)
__global__ __launch_bounds__(256) void linattn_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx, float* __restrict__ out,
                                   int n, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) float cs[];   // [heads][32][32]
  const int HD = heads * DH, RW = 3 * HD;
  const int unit = blockIdx.x;
  const int h = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < heads * DH * DH; e += blockDim.x) cs[e] = ctx[(int64_t)unit * heads * DH * DH + e];
  __syncthreads();
  const int j = blockIdx.y * 64 + (threadIdx.x & 63);
  if (j >= n) return;
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
#pragma unroll
  for (int e = 0; e < 8; ++e) op[e] = make_float4(o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]);
}

// backward pass B2: one thread per (token, head):
//   dq = scale * qsm * (dqs - <qsm, dqs>),  dqs[d] = sum_e dout[e] ctx[d][e]
//   dv[e] = sum_d ks[d] dctx[d][e] ; dk[d] = ks[d] * (sum_e v[e] dctx[d][e] - T[d])
__global__ __launch_bounds__(256) void linattn_bwd_tok_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ kstats,
                                       const float* __restrict__ ctx, const float* __restrict__ dctx, const float* __restrict__ tvec,
                                       float* __restrict__ dqkv, int n, int heads, float scale) {
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
  __syncthreads();
  const int j = blockIdx.y * 64 + (threadIdx.x & 63);
  if (j >= n) return;
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
    for (int e = 0; e < 8; ++e)
      reinterpret_cast<float4*>(wq)[e] = make_float4(scale * q[4 * e] * (dqs[4 * e] - dot), scale * q[4 * e + 1] * (dqs[4 * e + 1] - dot),
                                                     scale * q[4 * e + 2] * (dqs[4 * e + 2] - dot), scale * q[4 * e + 3] * (dqs[4 * e + 3] - dot));
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
      reinterpret_cast<float4*>(wq + HD)[e] = make_float4(dk[4 * e], dk[4 * e + 1], dk[4 * e + 2], dk[4 * e + 3]);
      reinterpret_cast<float4*>(wq + 2 * HD)[e] = make_float4(dv[4 * e], dv[4 * e + 1], dv[4 * e + 2], dv[4 * e + 3]);
    }
  }
}

static int la_check(int64_t units, int n, int heads) {
  if (units <= 0 || n <= 0 || heads <= 0 || units > 0x7fffffff / 8) return WDNO_EINVAL;
  if (heads != 1 && heads != 2 && heads != 4) return WDNO_EUNSUPPORTED;   // 64*heads threads per block
  return WDNO_OK;
}
// backward workspace: dctx [units,heads,32,32] + T [units,heads,32]
extern "C" size_t wdno_linattn_ws_bytes(int64_t units, int heads) {
  return ((size_t)units * heads * DH * DH + (size_t)units * heads * DH) * sizeof(float);
}
extern "C" int wdno_linattn_fwd(const float* qkv, float* out, float* kstats, float* ctx, int64_t units, int n_tok, int heads,
                                float scale, wdno_stream_t s) {
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
  linattn_ctx_kernel<0><<<(unsigned)(units * heads), 256, 0, st>>>(qkv, nullptr, kstats, nullptr, ctx, nullptr, n_tok, heads, scale);
  size_t lds = (size_t)heads * DH * DH * sizeof(float);
  linattn_out_kernel<<<dim3((unsigned)units, (unsigned)cdiv(n_tok, 64)), 64 * heads, lds, st>>>(qkv, ctx, out, n_tok, heads, scale);
  return wdno_check_launch();
}
extern "C" int wdno_linattn_bwd(const float* qkv, const float* dout, const float* kstats, const float* ctx, float* dqkv,
                                void* ws, size_t ws_bytes, int64_t units, int n_tok, int heads, float scale, wdno_stream_t s) {
  int rc = la_check(units, n_tok, heads);
  if (rc) return rc;
  if (ws_bytes < wdno_linattn_ws_bytes(units, heads)) return WDNO_EWORKSPACE;
  hipStream_t st = as_stream(s);
  float* dctx = (float*)ws;
  float* tvec = dctx + (size_t)units * heads * DH * DH;
  linattn_ctx_kernel<1><<<(unsigned)(units * heads), 256, 0, st>>>(qkv, dout, nullptr, ctx, dctx, tvec, n_tok, heads, scale);
  size_t lds = ((size_t)2 * heads * DH * DH + 3 * heads * DH) * sizeof(float);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)linattn_bwd_tok_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  linattn_bwd_tok_kernel<<<dim3((unsigned)units, (unsigned)cdiv(n_tok, 64)), 64 * heads, lds, st>>>(qkv, dout, kstats, ctx, dctx, tvec, dqkv,
                                                                                                n_tok, heads, scale);
  return wdno_check_launch();
}
