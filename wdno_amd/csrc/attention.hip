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
struct AttnP {
  wdno_attn_desc d;
  float scale;
  int RW;   // 3*heads*32
  int HD;   // heads*32
};

__device__ __forceinline__ float rot_fwd(float v, float cs, float sn, int dd) {
  float partner = __shfl_xor(v, 1, 64);
  return v * cs + ((dd & 1) ? partner : -partner) * sn;
}
__device__ __forceinline__ float rot_bwd(float g, float cs, float sn, int dd) {
  float partner = __shfl_xor(g, 1, 64);
  return g * cs + ((dd & 1) ? -partner : partner) * sn;
}

#define ATT_MAXJ 8   // 64 * 8 = 512 tokens max

__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                       const float* __restrict__ rsin, const float* __restrict__ bias,
                                                       float* __restrict__ out, AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = p.d.n_tok;
  float* Ks = smem;                 // [n][33]
  float* Vs = Ks + n * KST;         // [n][33]
  float* Ps = Vs + n * KST;         // [n]
  float* Qs = Ps + ((n + 3) & ~3);  // [32]
  const int lane = threadIdx.x, dd = lane & 31, half = lane >> 5;
  const int h = blockIdx.x % p.d.heads;
  const int unit = blockIdx.x / p.d.heads;
  const int uo = unit / p.d.n_ui, ui = unit - uo * p.d.n_ui;
  const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
  for (int j = half; j < n; j += 2) {
    const float* rp = qkv + (row0 + (int64_t)j * p.d.st) * p.RW + h * DH + dd;
    float kv = rp[p.HD];
    if (rcos) kv = rot_fwd(kv, rcos[j * DH + dd], rsin[j * DH + dd], dd);
    Ks[j * KST + dd] = kv;
    Vs[j * KST + dd] = rp[2 * p.HD];
  }
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    float qv = qkv[(row0 + (int64_t)i * p.d.st) * p.RW + h * DH + dd] * p.scale;
    if (rcos) qv = rot_fwd(qv, rcos[i * DH + dd], rsin[i * DH + dd], dd);
    if (half == 0) Qs[dd] = qv;
    __syncthreads();
    float sreg[ATT_MAXJ];
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < ATT_MAXJ; ++jj) {
      int j = lane + 64 * jj;
      float sv = -INFINITY;
      if (j < n) {
        sv = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) sv = fmaf(Qs[e], Ks[j * KST + e], sv);
        if (bias) sv += bias[((int64_t)h * n + i) * n + j];
      }
      sreg[jj] = sv;
      m = fmaxf(m, sv);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < ATT_MAXJ; ++jj) {
      int j = lane + 64 * jj;
      float pv = (j < n) ? expf(sreg[jj] - m) : 0.f;
      sreg[jj] = pv;
      l += pv;
    }
    l = wave_sum(l);
    float inv = 1.0f / l;
#pragma unroll
    for (int jj = 0; jj < ATT_MAXJ; ++jj) {
      int j = lane + 64 * jj;
      if (j < n) Ps[j] = sreg[jj] * inv;
    }
    __syncthreads();
    float acc = 0.f;
    for (int j = half; j < n; j += 2) acc = fmaf(Ps[j], Vs[j * KST + dd], acc);
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0) out[(row0 + (int64_t)i * p.d.st) * p.HD + h * DH + dd] = acc;
    __syncthreads();
  }
}

// backward: one wavefront per (unit, head) item, grid-strided so that the bias gradient is accumulated in LDS and
// flushed with one atomic per entry per block.
__global__ __launch_bounds__(64) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                       const float* __restrict__ rsin, const float* __restrict__ bias,
                                                       const float* __restrict__ dout, float* __restrict__ dqkv,
                                                       float* __restrict__ dbias, AttnP p, int units_total) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = p.d.n_tok;
  const int nn = n * KST;
  float* Qs = smem;
  float* Ks = Qs + nn;
  float* Vs = Ks + nn;
  float* Os = Vs + nn;     // dO
  float* dKs = Os + nn;
  float* dVs = dKs + nn;
  float* Ps = dVs + nn;               // [n]
  float* Ss = Ps + ((n + 3) & ~3);    // dS [n]
  float* dBs = Ss + ((n + 3) & ~3);   // [n*n] when dbias
  const int lane = threadIdx.x, dd = lane & 31, half = lane >> 5;
  const int h = blockIdx.x % p.d.heads;
  const int ustride = gridDim.x / p.d.heads;
  if (dbias)
    for (int e = lane; e < n * n; e += 64) dBs[e] = 0.f;
  for (int unit = blockIdx.x / p.d.heads; unit < units_total; unit += ustride) {
    const int uo = unit / p.d.n_ui, ui = unit - uo * p.d.n_ui;
    const int64_t row0 = (int64_t)uo * p.d.so + (int64_t)ui * p.d.si;
    __syncthreads();
    for (int j = half; j < n; j += 2) {
      const int64_t row = row0 + (int64_t)j * p.d.st;
      const float* rp = qkv + row * p.RW + h * DH + dd;
      float qv = rp[0] * p.scale, kv = rp[p.HD];
      if (rcos) {
        float cs = rcos[j * DH + dd], sn = rsin[j * DH + dd];
        qv = rot_fwd(qv, cs, sn, dd);
        kv = rot_fwd(kv, cs, sn, dd);
      }
      Qs[j * KST + dd] = qv;
      Ks[j * KST + dd] = kv;
      Vs[j * KST + dd] = rp[2 * p.HD];
      Os[j * KST + dd] = dout[row * p.HD + h * DH + dd];
      dKs[j * KST + dd] = 0.f;
      dVs[j * KST + dd] = 0.f;
    }
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      float sreg[ATT_MAXJ], dpreg[ATT_MAXJ];
      float m = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < ATT_MAXJ; ++jj) {
        int j = lane + 64 * jj;
        float sv = -INFINITY, dp = 0.f;
        if (j < n) {
          sv = 0.f;
#pragma unroll
          for (int e = 0; e < DH; ++e) {
            sv = fmaf(Qs[i * KST + e], Ks[j * KST + e], sv);
            dp = fmaf(Os[i * KST + e], Vs[j * KST + e], dp);
          }
          if (bias) sv += bias[((int64_t)h * n + i) * n + j];
        }
        sreg[jj] = sv; dpreg[jj] = dp;
        m = fmaxf(m, sv);
      }
      m = wave_max(m);
      float l = 0.f;
#pragma unroll
      for (int jj = 0; jj < ATT_MAXJ; ++jj) {
        int j = lane + 64 * jj;
        float pv = (j < n) ? expf(sreg[jj] - m) : 0.f;
        sreg[jj] = pv;
        l += pv;
      }
      l = wave_sum(l);
      float inv = 1.0f / l, delta = 0.f;
#pragma unroll
      for (int jj = 0; jj < ATT_MAXJ; ++jj) { sreg[jj] *= inv; delta = fmaf(sreg[jj], dpreg[jj], delta); }
      delta = wave_sum(delta);
#pragma unroll
      for (int jj = 0; jj < ATT_MAXJ; ++jj) {
        int j = lane + 64 * jj;
        if (j < n) {
          float ds = sreg[jj] * (dpreg[jj] - delta);
          Ps[j] = sreg[jj];
          Ss[j] = ds;
          if (dbias) dBs[i * n + j] += ds;
        }
      }
      __syncthreads();
      float dq = 0.f;
      const float qi = Qs[i * KST + dd], oi = Os[i * KST + dd];
      for (int j = half; j < n; j += 2) {
        float ds = Ss[j], pj = Ps[j];
        dq = fmaf(ds, Ks[j * KST + dd], dq);
        dKs[j * KST + dd] = fmaf(ds, qi, dKs[j * KST + dd]);
        dVs[j * KST + dd] = fmaf(pj, oi, dVs[j * KST + dd]);
      }
      dq += __shfl_xor(dq, 32, 64);
      if (rcos) dq = rot_bwd(dq, rcos[i * DH + dd], rsin[i * DH + dd], dd);
      if (half == 0) dqkv[(row0 + (int64_t)i * p.d.st) * p.RW + h * DH + dd] = dq * p.scale;
      __syncthreads();
    }
    for (int j = half; j < n; j += 2) {
      const int64_t row = row0 + (int64_t)j * p.d.st;
      float dk = dKs[j * KST + dd];
      if (rcos) dk = rot_bwd(dk, rcos[j * DH + dd], rsin[j * DH + dd], dd);
      float* wp = dqkv + row * p.RW + h * DH + dd;
      wp[p.HD] = dk;
      wp[2 * p.HD] = dVs[j * KST + dd];
    }
  }
  if (dbias) {
    __syncthreads();
    for (int e = lane; e < n * n; e += 64) atomicAdd(&dbias[(int64_t)h * n * n + e], dBs[e]);
  }
}

static int attn_check(const wdno_attn_desc* d) {
  if (!d || d->n_uo <= 0 || d->n_ui <= 0 || d->n_tok <= 0 || d->heads <= 0) return WDNO_EINVAL;
  return WDNO_OK;
}
extern "C" int wdno_attn_fwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out,
                             const wdno_attn_desc* d, float scale, wdno_stream_t s) {
  int rc = attn_check(d);
  if (rc) return rc;
  if (d->n_tok > 64 * ATT_MAXJ) return WDNO_EUNSUPPORTED;
  AttnP p;
  p.d = *d; p.scale = scale; p.HD = d->heads * DH; p.RW = 3 * p.HD;
  size_t lds = ((size_t)2 * d->n_tok * KST + ((d->n_tok + 3) & ~3) + DH) * sizeof(float);
  if (lds > 160 * 1024) return WDNO_EUNSUPPORTED;
  int64_t blocks = (int64_t)d->n_uo * d->n_ui * d->heads;
  if (blocks > 0x7fffffff) return WDNO_EUNSUPPORTED;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_fwd_kernel<<<(unsigned)blocks, 64, lds, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, out, p);
  return wdno_check_launch();
}
extern "C" int wdno_attn_bwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* dout,
                             float* dqkv, float* dbias, const wdno_attn_desc* d, float scale, wdno_stream_t s) {
  int rc = attn_check(d);
  if (rc) return rc;
  if (d->n_tok > 64 * ATT_MAXJ) return WDNO_EUNSUPPORTED;
  AttnP p;
  p.d = *d; p.scale = scale; p.HD = d->heads * DH; p.RW = 3 * p.HD;
  const int n = d->n_tok;
  size_t lds = ((size_t)6 * n * KST + 2 * ((n + 3) & ~3) + (dbias ? (size_t)n * n : 0)) * sizeof(float);
  if (lds > 160 * 1024) return WDNO_EUNSUPPORTED;
  int64_t units = (int64_t)d->n_uo * d->n_ui;
  int64_t ub = units;
  if (dbias && ub > 1024) ub = 1024;     // bound the number of atomic flushes
  if (units > 0x7fffffff) return WDNO_EUNSUPPORTED;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_kernel<<<(unsigned)(ub * d->heads), 64, lds, as_stream(s)>>>(qkv, rot_cos, rot_sin, bias, dout, dqkv, dbias, p, (int)units);
  return wdno_check_launch();
}

// ================================================================================================ linear attention
// pass 1: per (unit, column = head*32+d): max over tokens and sum of exp -> kstats[unit][col][2]
__global__ __launch_bounds__(256) void linattn_kstats_kernel(const float* __restrict__ qkv, float* __restrict__ kstats, int n, int HD) {
  __shared__ float rm[256], rl[256];
  const int unit = blockIdx.x;
  const int col = threadIdx.x % HD, rg = threadIdx.x / HD, nrg = 256 / HD;
  const float* kp = qkv + ((int64_t)unit * n) * (3 * HD) + HD + col;
  float m = -INFINITY, l = 0.f;
  for (int j = rg; j < n; j += nrg) {
    float v = kp[(int64_t)j * 3 * HD];
    float mn = fmaxf(m, v);
    l = l * expf(m - mn) + expf(v - mn);
    m = mn;
  }
  rm[threadIdx.x] = m; rl[threadIdx.x] = l;
  __syncthreads();
  if (rg == 0) {
    for (int t = 1; t < nrg; ++t) {
      float m2 = rm[t * HD + col], l2 = rl[t * HD + col];
      float mn = fmaxf(m, m2);
      float a = (m == -INFINITY) ? 0.f : l * expf(m - mn);
      float b = (m2 == -INFINITY) ? 0.f : l2 * expf(m2 - mn);
      l = a + b; m = mn;
    }
    kstats[((int64_t)unit * HD + col) * 2 + 0] = m;
    kstats[((int64_t)unit * HD + col) * 2 + 1] = l;
  }
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
  linattn_kstats_kernel<<<(unsigned)units, 256, 0, st>>>(qkv, kstats, n_tok, heads * DH);
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
