// norm.hip -- GroupNorm(+scale/shift)+SiLU and channel LayerNorm on channels-last tensors (HBM-bound).
//
// GroupNorm [N, S, C] with G groups (G = 1 is a whole-sample reduction of up to 1024*64*64 elements for Burgers):
//   pass 1  gn_partial  : per (sample, row-chunk) per-channel sum / sum-of-squares, accumulated in fp64
//   pass 2  gn_finalize : per sample: group mean / rstd (fp64), folded per-channel affine  y = act(a[c]*x + b[c])
//   pass 3  gn_apply    : one read of x, one write of y
// Backward mirrors it: per-channel sums of dz and dz*xhat, a per-sample finalize producing the parameter
// gradients and the two group means, then one elementwise pass for dx.
#include "common.h"
#include <type_traits>
extern int wdno_debug_mode;      // 58 / 59: GroupNorm timing experiments -- the forward statistics pass / the backward reduction pass is launched TWICE (same results): the step-time difference is what the pass costs inside the step

#define GN_MAXC 1024

// Element type of an activation operand: float, or bf16 storage (single-product mode, WDNO_CONV_MATH=bf16: the convolution in front of a norm
// writes its output -- and the data gradient that reaches a norm's backward -- as bf16, train_diffusion.py:61-62 mixed-precision semantics;
// the arithmetic here stays fp32 / fp64). e = element offset of four / eight consecutive channels.
struct gn_bf16 {};
template <typename T> struct gn_ld;
template <> struct gn_ld<float> {
  static __device__ __forceinline__ float4 ld4(const void* p, int64_t e) { return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + e); }
};
template <> struct gn_ld<gn_bf16> {
  static __device__ __forceinline__ float4 ld4(const void* p, int64_t e) {
    const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(p) + e);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  }
};
template <typename T> __device__ __forceinline__ void gn_ld8(const void* p, int64_t e, float (&v)[8]) {
  const float4 a = gn_ld<T>::ld4(p, e), b = gn_ld<T>::ld4(p, e + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

static inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int gn_chunks(int64_t S, int64_t N, int C) {
  int64_t c = cdiv64(S, 64);
  if (c > 64) c = 64;
  if (c < 1) c = 1;
  // few samples x few rows of many channels (the deep levels of the Burgers U-Net: 16 samples x 64 pixels x 1024 channels was 16 blocks,
  // 34 us per backward reduction): finer chunks, down to one round of loads (4 rows per row group) each, until ~256 blocks exist
  int txp = pow2ceil(C >> 2);
  if (txp > 256) txp = 256;
  const int nty = 256 / (txp < 1 ? 1 : txp);
  // (and with one or two LARGE samples -- super-resolution sampling at batch 2: 307 200 pixels each -- more than 64 chunks: 128 blocks were 38.6 us per
  // statistics pass over 157 MB)
  const int64_t cap = N > 0 && 512 / N > 64 ? 512 / N : 64;
  while (c < cap && N * c < 256 && S / (2 * c) >= 4 * nty) c *= 2;
  return (int)c;
}

// ---------------------------------------------------------------------------------------------- per-channel partial sums
// MODE 0: (sum x, sum x^2)         MODE 1: (sum dz, sum dz*xhat) with dz = dy * act'(a x + b)
template <int MODE, typename XT = float, typename DT = float>
__global__ __launch_bounds__(256) void gn_partial_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                          const float* __restrict__ cb /*[N][C][4]*/, const float* __restrict__ gb /*[N][G][4]*/,
                                                          double* __restrict__ part, int64_t S, int C, int cg, int G, int txp,
                                                          int64_t rows_per_chunk, int silu, float* __restrict__ mx = nullptr) {
  // mx (optional): per (sample, chunk) max|dz| and max|xhat| (MODE 1) or max|x| (MODE 0) -- what the finalize kernels need to
  // bound |dx| / |y| before the apply pass runs
  __shared__ double red[256 * 8];
  __shared__ float redm[8];
  float m0[4] = {0.f, 0.f, 0.f, 0.f}, m1[4] = {0.f, 0.f, 0.f, 0.f};
  const int n = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int C4 = C >> 2;
  const int tx = threadIdx.x & (txp - 1), ty = threadIdx.x / txp, nty = 256 / txp;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > S) r1 = S;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (tx < C4) {
    float ca[4], cbb[4], mean[4], rstd[4];
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int c = tx * 4 + j;
        ca[j] = cb[((int64_t)n * C + c) * 4 + 0];
        cbb[j] = cb[((int64_t)n * C + c) * 4 + 1];
        int g = c / cg;
        mean[j] = gb[((int64_t)n * G + g) * 4 + 0];
        rstd[j] = gb[((int64_t)n * G + g) * 4 + 1];
      }
    }
    const int64_t e0 = ((int64_t)n * S) * C + tx * 4;             // element offset of this thread's four channels in row 0 of the sample
    // four rows per round (eight of bf16 storage: the same bytes in flight -- with four, the bf16 form of the backward reduction was SLOWER than
    // the fp32 one, 105 vs 93 us per launch at Burgers batch 256), all loads requested before the first is used (one row at a time leaves
    // ~4 MB in flight chip-wide)
    constexpr int RIF = std::is_same<XT, gn_bf16>::value ? 8 : 4;
    for (int64_t rb = r0 + ty; rb < r1; rb += RIF * nty) {
      float4 vv[RIF], dd[RIF];
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        const int64_t r = rb + u * nty;
        const bool ok = r < r1;
        vv[u] = ok ? gn_ld<XT>::ld4(x, e0 + r * C) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1) dd[u] = ok ? gn_ld<DT>::ld4(dy, e0 + r * C) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        if (rb + u * nty >= r1) break;
        const float4 v = vv[u];
        float xv[4] = {v.x, v.y, v.z, v.w};
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { s0[j] += (double)xv[j]; s1[j] += (double)xv[j] * (double)xv[j]; m0[j] = fmaxf(m0[j], fabsf(xv[j])); }
        } else {
          const float4 d = dd[u];
          float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float dz = dv[j];
            if (silu) dz *= silu_grad_f(ca[j] * xv[j] + cbb[j]);
            float xh = (xv[j] - mean[j]) * rstd[j];
            s0[j] += (double)dz;
            s1[j] += (double)dz * (double)xh;
            m0[j] = fmaxf(m0[j], fabsf(dz)); m1[j] = fmaxf(m1[j], fabsf(xh));
          }
        }
      }
    }
  }
  // reduce over the row groups: red[ty][tx][8]
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[(ty * txp + tx) * 8 + j] = s0[j]; red[(ty * txp + tx) * 8 + 4 + j] = s1[j]; }
  __syncthreads();
  if (ty == 0 && tx < C4) {
    for (int t = 1; t < nty; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] += red[(t * txp + tx) * 8 + j]; s1[j] += red[(t * txp + tx) * 8 + 4 + j]; }
    double* o = part + (((int64_t)n * nchunk + chunk) * C + tx * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j * 2] = s0[j]; o[j * 2 + 1] = s1[j]; }
  }
  if (mx) {                       // block maxima of |dz| and |xhat| (MODE 1) / of |x| (MODE 0)
    float a = fmaxf(fmaxf(m0[0], m0[1]), fmaxf(m0[2], m0[3])), b = fmaxf(fmaxf(m1[0], m1[1]), fmaxf(m1[2], m1[3]));
    a = wave_max(a); b = wave_max(b);
    if ((threadIdx.x & 63) == 0) { redm[(threadIdx.x >> 6) * 2] = a; redm[(threadIdx.x >> 6) * 2 + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float* o = mx + ((int64_t)n * nchunk + chunk) * 2;
      o[0] = fmaxf(fmaxf(redm[0], redm[2]), fmaxf(redm[4], redm[6]));
      o[1] = fmaxf(fmaxf(redm[1], redm[3]), fmaxf(redm[5], redm[7]));
    }
  }
}

// Per-channel totals of the chunk partials of sample n: sa[c], sb[c] for c < C (arrays of GN_MAXC doubles in LDS). All 256
// threads take part -- for C <= 128 a channel's chunk list is cut into 256 / C pieces that are folded through LDS -- so the
// chains of dependent loads are nchunk * C / 256 long instead of nchunk (these one-block-per-sample kernels are pure latency:
// 8.6 and 16.6 us per launch before, ~100 launches per train step).
// (cb0, cn: the channels cb0 .. cb0 + cn - 1 only, results at sa / sb[0 .. cn) -- the finalize kernels run one block per (sample, slice of groups))
__device__ __forceinline__ void gn_chunk_totals(const double* __restrict__ part, int n, int nchunk, int Call, double* sa, double* sb, int cb0 = 0, int cn = -1) {
  const int C = cn < 0 ? Call : cn;
  part += cb0 * 2;
  const int nsub = C <= 128 ? 256 / C : 1;
  for (int idx = threadIdx.x; idx < C * nsub; idx += 256) {
    const int q = idx / C, c = idx - q * C;
    const int k1 = (q + 1) * nchunk / nsub;
    int k = q * nchunk / nsub;
    double a0 = 0, b0 = 0, a1 = 0, b1 = 0;             // two chains (even / odd chunks), as before -- but the loads of EIGHT chunks are
    // requested before the first is added: the adds were chained behind one L2 round trip per pair of chunks (16 chunks per thread at
    // C = 64: ~6 of the 8.8 us of a launch that 8 blocks make 30 times per step, forward and backward)
    for (; k + 7 < k1; k += 8) {
      const double* q0 = part + (((int64_t)n * nchunk + k) * Call + c) * 2;
      double va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { va[u] = q0[(int64_t)u * Call * 2]; vb[u] = q0[(int64_t)u * Call * 2 + 1]; }
#pragma unroll
      for (int u = 0; u < 8; u += 2) { a0 += va[u]; b0 += vb[u]; a1 += va[u + 1]; b1 += vb[u + 1]; }
    }
    for (; k + 1 < k1; k += 2) {
      const double* q0 = part + (((int64_t)n * nchunk + k) * Call + c) * 2;
      const double* q1 = q0 + (int64_t)Call * 2;
      a0 += q0[0]; b0 += q0[1]; a1 += q1[0]; b1 += q1[1];
    }
    if (k < k1) {
      const double* q0 = part + (((int64_t)n * nchunk + k) * Call + c) * 2;
      a0 += q0[0]; b0 += q0[1];
    }
    sa[idx] = a0 + a1; sb[idx] = b0 + b1;               // idx = q * C + c
  }
  __syncthreads();
  if (nsub > 1) {
    const int c = threadIdx.x;
    double a = 0, b = 0;
    if (c < C)
      for (int q = 0; q < nsub; ++q) { a += sa[q * C + c]; b += sb[q * C + c]; }
    __syncthreads();
    if (c < C) { sa[c] = a; sb[c] = b; }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- forward finalize
// one block per sample. writes stats[n][g] = (mean, rstd), cb[n][c] = (a, b, k1, 0), gb[n][g] = (mean, rstd, 0, 0)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, const float* __restrict__ stats_in,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ ss, float* __restrict__ stats_out,
                                                           float* __restrict__ cb, float* __restrict__ gb, int64_t S, int C, int G,
                                                           int nchunk, float eps, const float* __restrict__ mx = nullptr,
                                                           float* __restrict__ bound_rec = nullptr) {
  __shared__ double chs[GN_MAXC], chq[GN_MAXC];
  __shared__ float gmean[GN_MAXC], grstd[GN_MAXC];
  const int n = blockIdx.x, cg = C / G;
  // round 6: gridDim.y slices of the groups (their statistics are independent): with one block per sample the launch was a chain of dependent loads
  // over nchunk x C partials -- 7.5 us per norm, 60 such launches per training step; a slice of one group of 8 channels reads 8 KB of them.
  // LDS tables are indexed from the slice's first channel / group.
  const int gps = (G + (int)gridDim.y - 1) / (int)gridDim.y, gs0 = (int)blockIdx.y * gps, gs1 = min(G, gs0 + gps);
  if (gs0 >= gs1) return;
  const int cs0 = gs0 * cg, csn = (gs1 - gs0) * cg;
  gamma += cs0; beta += cs0;
  if (ss) ss += cs0;
  cb += (int64_t)cs0 * 4; gb += (int64_t)gs0 * 4;
  if (stats_out) stats_out += (int64_t)gs0 * 2;
  if (stats_in) stats_in += (int64_t)gs0 * 2;
  const int Call = C, Gall = G;
  C = csn; G = gs1 - gs0;
  if (part) {
    gn_chunk_totals(part, n, nchunk, Call, chs, chq, cs0, csn);
    if (cg > 64) {
      // few, wide groups (GroupNorm(1, C) of the Burgers U-Net): the whole block reduces each group
      __shared__ double ra[256], rb[256];
      for (int g = 0; g < G; ++g) {
        double a = 0, b = 0;
        for (int c = g * cg + threadIdx.x; c < (g + 1) * cg; c += 256) { a += chs[c]; b += chq[c]; }
        ra[threadIdx.x] = a; rb[threadIdx.x] = b;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
          if (threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
          __syncthreads();
        }
        if (threadIdx.x == 0) { chs[g * cg] = ra[0]; chq[g * cg] = rb[0]; }      // group totals parked in the first channel slot
        __syncthreads();
      }
    }
    for (int g = threadIdx.x; g < G; g += 256) {
      double a = 0, b = 0;
      if (cg > 64) { a = chs[g * cg]; b = chq[g * cg]; }
      else for (int c = g * cg; c < (g + 1) * cg; ++c) { a += chs[c]; b += chq[c]; }
      double m = (double)cg * (double)S;
      double mean = a / m;
      double var = b / m - mean * mean;
      if (var < 0) var = 0;
      float rstd = (float)(1.0 / sqrt(var + (double)eps));
      gmean[g] = (float)mean; grstd[g] = rstd;
      stats_out[((int64_t)n * Gall + g) * 2 + 0] = (float)mean;
      stats_out[((int64_t)n * Gall + g) * 2 + 1] = rstd;
    }
  } else {
    for (int g = threadIdx.x; g < G; g += 256) {
      gmean[g] = stats_in[((int64_t)n * Gall + g) * 2 + 0];
      grstd[g] = stats_in[((int64_t)n * Gall + g) * 2 + 1];
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    float* o = gb + ((int64_t)n * Gall + g) * 4;
    o[0] = gmean[g]; o[1] = grstd[g]; o[2] = 0.f; o[3] = 0.f;
  }
  float ma = 0.f, mb = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    int g = c / cg;
    float sc1 = ss ? ss[(int64_t)n * 2 * Call + c] + 1.0f : 1.0f;
    float sh = ss ? ss[(int64_t)n * 2 * Call + Call + c] : 0.0f;
    float k1 = grstd[g] * gamma[c];
    float a = k1 * sc1;
    float b = (beta[c] - gmean[g] * k1) * sc1 + sh;
    float* o = cb + ((int64_t)n * Call + c) * 4;
    o[0] = a; o[1] = b; o[2] = a; o[3] = 0.f;   // k1*(s+1) == a
    ma = fmaxf(ma, fabsf(a)); mb = fmaxf(mb, fabsf(b));
  }
  if (bound_rec) {          // |act(a x + b)| <= max|a| max|x| + max|b|: known before the apply pass (gn_apply_planes_kernel)
    float mxx = 0.f;
    for (int k = threadIdx.x; k < nchunk; k += 256) mxx = fmaxf(mxx, mx[((int64_t)n * nchunk + k) * 2]);
    __shared__ float bm[3][4];
    float v[3] = {wave_max(ma), wave_max(mb), wave_max(mxx)};
    if ((threadIdx.x & 63) == 0)
      for (int q = 0; q < 3; ++q) bm[q][threadIdx.x >> 6] = v[q];
    __syncthreads();
    if (threadIdx.x == 0) {
      float t[3];
      for (int q = 0; q < 3; ++q) t[q] = fmaxf(fmaxf(bm[q][0], bm[q][1]), fmaxf(bm[q][2], bm[q][3]));
      atomicMax(reinterpret_cast<unsigned*>(bound_rec) + (n & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE, __float_as_uint(t[0] * t[2] + t[1]));
    }
  }
}

template <typename XT = float>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ x, const float* __restrict__ cb, float* __restrict__ y,
                                                        int64_t S, int C, int silu, float* __restrict__ amax_rec) {
  const int n = blockIdx.y;
  const int C4 = C >> 2;
  const int64_t total4 = S * C4;
  const int64_t xe = (int64_t)n * S * C;
  float4* yp = reinterpret_cast<float4*>(y + (int64_t)n * S * C);
  const float4* cbp = reinterpret_cast<const float4*>(cb + (int64_t)n * C * 4);
  const int64_t stride = (int64_t)gridDim.x * 256;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    int c4 = (int)(i % C4);
    float4 v = gn_ld<XT>::ld4(x, xe + i * 4);
    float4 k0 = cbp[c4 * 4 + 0], k1 = cbp[c4 * 4 + 1], k2 = cbp[c4 * 4 + 2], k3 = cbp[c4 * 4 + 3];
    float4 o;
    o.x = k0.x * v.x + k0.y; o.y = k1.x * v.y + k1.y; o.z = k2.x * v.z + k2.y; o.w = k3.x * v.w + k3.y;
    if (silu) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
    yp[i] = o;
    am = amax4(am, o);
  }
  if (amax_rec) amax_record_emit(am, amax_rec, blockIdx.y * gridDim.x + blockIdx.x);
}

// ---------------------------------------------------------------------------------------------- backward finalize / apply
// one block per sample: parameter-gradient pieces and the two group means
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ ss,
                                                               float* __restrict__ gb, float* __restrict__ dgb, float* __restrict__ dss,
                                                               int64_t S, int C, int G, int nchunk, const float* __restrict__ cb = nullptr,
                                                               const float* __restrict__ mx = nullptr, float* __restrict__ bound_rec = nullptr) {
  __shared__ double chA[GN_MAXC], chB[GN_MAXC];
  const int n = blockIdx.x, cg = C / G;
  // a block = (sample, slice of groups), as gn_finalize_kernel
  const int gps = (G + (int)gridDim.y - 1) / (int)gridDim.y, gs0 = (int)blockIdx.y * gps, gs1 = min(G, gs0 + gps);
  if (gs0 >= gs1) return;
  const int cs0 = gs0 * cg, csn = (gs1 - gs0) * cg;
  gamma += cs0; beta += cs0;
  if (ss) ss += cs0;
  if (dss) dss += cs0;
  dgb += cs0; gb += (int64_t)gs0 * 4;
  if (cb) cb += (int64_t)cs0 * 4;
  const int Call = C, Gall = G;
  C = csn; G = gs1 - gs0;
  gn_chunk_totals(part, n, nchunk, Call, chA, chB, cs0, csn);
  for (int c = threadIdx.x; c < C; c += 256) {
    const double a = chA[c], b = chB[c];
    float sc1 = ss ? ss[(int64_t)n * 2 * Call + c] + 1.0f : 1.0f;
    if (dss) {
      dss[(int64_t)n * 2 * Call + c] = (float)((double)gamma[c] * b + (double)beta[c] * a);   // d scale
      dss[(int64_t)n * 2 * Call + Call + c] = (float)a;                                          // d shift
    }
    dgb[((int64_t)n * 2 + 0) * Call + c] = (float)((double)sc1 * b);   // d gamma (this sample)
    dgb[((int64_t)n * 2 + 1) * Call + c] = (float)((double)sc1 * a);   // d beta
    double w = (double)gamma[c] * (double)sc1;
    chA[c] = w * a; chB[c] = w * b;
  }
  __syncthreads();
  if (cg > 64) {                       // few, wide groups: block-wide reduction per group (see gn_finalize_kernel)
    __shared__ double ra[256], rb[256];
    for (int g = 0; g < G; ++g) {
      double a = 0, b = 0;
      for (int c = g * cg + threadIdx.x; c < (g + 1) * cg; c += 256) { a += chA[c]; b += chB[c]; }
      ra[threadIdx.x] = a; rb[threadIdx.x] = b;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
        __syncthreads();
      }
      if (threadIdx.x == 0) { chA[g * cg] = ra[0]; chB[g * cg] = rb[0]; }
      __syncthreads();
    }
  }
  for (int g = threadIdx.x; g < G; g += 256) {
    double a = 0, b = 0;
    if (cg > 64) { a = chA[g * cg]; b = chB[g * cg]; }
    else for (int c = g * cg; c < (g + 1) * cg; ++c) { a += chA[c]; b += chB[c]; }
    double m = (double)cg * (double)S;
    float* o = gb + ((int64_t)n * Gall + g) * 4;
    float rstd = o[1];
    o[2] = (float)(a / m) * rstd;
    o[3] = (float)(b / m) * rstd;
  }
  if (bound_rec) {
    // |dx| = |k.z dz - g.z - xhat g.w| <= max|k.z| max|dz| + max|g.z| + max|xhat| max|g.w|: an upper bound of max|dx| that is known
    // BEFORE the apply pass, so that pass can write the (hi, lo) fp16 planes of dx directly (gn_bwd_apply_planes_kernel)
    __syncthreads();
    float mdz = 0.f, mxh = 0.f, mk = 0.f, mz = 0.f, mw = 0.f;
    for (int k = threadIdx.x; k < nchunk; k += 256) {
      mdz = fmaxf(mdz, mx[((int64_t)n * nchunk + k) * 2]); mxh = fmaxf(mxh, mx[((int64_t)n * nchunk + k) * 2 + 1]);
    }
    for (int c = threadIdx.x; c < C; c += 256) mk = fmaxf(mk, fabsf(cb[((int64_t)n * Call + c) * 4 + 2]));
    for (int g = threadIdx.x; g < G; g += 256) {
      mz = fmaxf(mz, fabsf(gb[((int64_t)n * Gall + g) * 4 + 2])); mw = fmaxf(mw, fabsf(gb[((int64_t)n * Gall + g) * 4 + 3]));
    }
    __shared__ float bm[5][4];
    float v[5] = {wave_max(mdz), wave_max(mxh), wave_max(mk), wave_max(mz), wave_max(mw)};
    if ((threadIdx.x & 63) == 0)
      for (int q = 0; q < 5; ++q) bm[q][threadIdx.x >> 6] = v[q];
    __syncthreads();
    if (threadIdx.x == 0) {
      float t[5];
      for (int q = 0; q < 5; ++q) t[q] = fmaxf(fmaxf(bm[q][0], bm[q][1]), fmaxf(bm[q][2], bm[q][3]));
      const float bound = t[2] * t[0] + t[3] + t[1] * t[4];
      atomicMax(reinterpret_cast<unsigned*>(bound_rec) + (n & (WDNO_AMAX_SLOTS - 1)) * WDNO_AMAX_STRIDE, __float_as_uint(bound));
    }
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ cb, const float* __restrict__ gb,
                                                            float* __restrict__ dx, int64_t S, int C, int cg, int G, int silu,
                                                            float* __restrict__ amax_rec) {
  const int n = blockIdx.y;
  const int C4 = C >> 2;
  const int64_t total4 = S * C4;
  const float4* xp = reinterpret_cast<const float4*>(x + (int64_t)n * S * C);
  const float4* dp = reinterpret_cast<const float4*>(dy + (int64_t)n * S * C);
  float4* op = reinterpret_cast<float4*>(dx + (int64_t)n * S * C);
  const float4* cbp = reinterpret_cast<const float4*>(cb + (int64_t)n * C * 4);
  const float4* gbp = reinterpret_cast<const float4*>(gb + (int64_t)n * G * 4);
  const int64_t stride = (int64_t)gridDim.x * 256;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    int c4 = (int)(i % C4);
    float4 v = xp[i], d = dp[i];
    float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w}, o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c = c4 * 4 + j;
      float4 k = cbp[c];
      float4 gq = gbp[c / cg];
      float dz = dv[j];
      if (silu) dz *= silu_grad_f(k.x * xv[j] + k.y);
      float xh = (xv[j] - gq.x) * gq.y;
      o[j] = k.z * dz - gq.z - xh * gq.w;
    }
    op[i] = make_float4(o[0], o[1], o[2], o[3]);
    am = fmaxf(fmaxf(am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (amax_rec) amax_record_emit(am, amax_rec, blockIdx.y * gridDim.x + blockIdx.x);
}

// y as (hi, lo) fp16 planes (scale from the bound gn_finalize_kernel left in `rec`): the output of a Block whose only reader is the
// next convolution. A thread owns 8 channels.
typedef _Float16 gn_half8 __attribute__((ext_vector_type(8)));
template <typename XT = float>
__global__ __launch_bounds__(256) void gn_apply_planes_kernel(const void* __restrict__ x, const float* __restrict__ cb,
                                                               _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                               float* __restrict__ scale_out, const float* __restrict__ rec,
                                                               int64_t S, int C, int silu) {
  const int n = blockIdx.y;
  const int C8 = C >> 3;
  const int64_t total8 = S * C8;
  const bool single = lo == nullptr;                 // one bf16 plane, no scale (WDNO_CONV_MATH=bf16)
  const float s = single ? 1.0f : scale_from_amax(amax_record_read(rec));
  if (!single && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale_out[0] = s;
  const int64_t xe = (int64_t)n * S * C;
  _Float16* hp = hi + (int64_t)n * S * C;
  _Float16* lp = single ? nullptr : lo + (int64_t)n * S * C;
  const float4* cbp = reinterpret_cast<const float4*>(cb + (int64_t)n * C * 4);
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int c0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % C8) * 8;
  float ka[8], kb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float4 k = cbp[c0 + j]; ka[j] = k.x; kb[j] = k.y; }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += stride) {
    float xv[8];
    gn_ld8<XT>(x, xe + i * 8, xv);
    gn_half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float o = ka[j] * xv[j] + kb[j];
      if (silu) o = silu_f(o);
      _Float16 th, tl;
      plane_pack(o, s, single, th, tl);
      h[j] = th; l[j] = tl;
    }
    *reinterpret_cast<gn_half8*>(hp + i * 8) = h;
    if (!single) *reinterpret_cast<gn_half8*>(lp + i * 8) = l;
  }
}

// y = act(a x + b) + res as BOTH the fp32 tensor and its (hi, lo) fp16 planes: the tail of a ResnetBlock with an identity skip
// (unet.py:167-176, conv3d.py:286-300). The sum is the next block's skip (fp32) and the operand of its first convolution (planes): this
// replaces the apply pass, the add and the split (28 bytes per element over three launches) by one pass of 16. Scale from
// max|a| max|x| + max|b| (gn_finalize_kernel, in `rec`) + max|res| (the amax record of res).
template <typename XT = float>
__global__ __launch_bounds__(256) void gn_apply_add_planes_kernel(const void* __restrict__ x, const float* __restrict__ cb,
                                                                   const float* __restrict__ res, float* __restrict__ y,
                                                                   _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                                   float* __restrict__ scale_out, const float* __restrict__ rec,
                                                                   const float* __restrict__ rec_res, float* __restrict__ amax_out,
                                                                   int64_t S, int C, int silu) {
  const int n = blockIdx.y;
  const int C8 = C >> 3;
  const int64_t total8 = S * C8;
  const bool planes = hi != nullptr;                 // hi == nullptr: the fp32 sum only (its reader is a LayerNorm, not a convolution)
  const bool single = lo == nullptr;                 // one bf16 plane, no scale (WDNO_CONV_MATH=bf16)
  const float s = single ? 1.0f : scale_from_amax(amax_record_read(rec) + amax_record_read(rec_res));
  if (!single && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale_out[0] = s;
  const int64_t xe = (int64_t)n * S * C;
  const float* rp = res + (int64_t)n * S * C;
  float* yp = y + (int64_t)n * S * C;
  _Float16* hp = planes ? hi + (int64_t)n * S * C : nullptr;
  _Float16* lp = single ? nullptr : lo + (int64_t)n * S * C;
  const float4* cbp = reinterpret_cast<const float4*>(cb + (int64_t)n * C * 4);
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int c0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % C8) * 8;      // fixed per thread: (gridDim.x * 256) % C8 == 0
  float ka[8], kb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float4 k = cbp[c0 + j]; ka[j] = k.x; kb[j] = k.y; }
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += stride) {
    float xv[8];
    gn_ld8<XT>(x, xe + i * 8, xv);
    const float4 r0 = *reinterpret_cast<const float4*>(rp + i * 8), r1 = *reinterpret_cast<const float4*>(rp + i * 8 + 4);
    const float rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    float ov[8];
    gn_half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float o = ka[j] * xv[j] + kb[j];
      if (silu) o = silu_f(o);
      o += rv[j];
      ov[j] = o;
      am = fmaxf(am, fabsf(o));
      _Float16 th, tl;
      plane_pack(o, s, single, th, tl);
      h[j] = th; l[j] = tl;
    }
    *reinterpret_cast<float4*>(yp + i * 8) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    *reinterpret_cast<float4*>(yp + i * 8 + 4) = make_float4(ov[4], ov[5], ov[6], ov[7]);
    if (planes) *reinterpret_cast<gn_half8*>(hp + i * 8) = h;
    if (planes && !single) *reinterpret_cast<gn_half8*>(lp + i * 8) = l;
  }
  if (amax_out) amax_record_emit(am, amax_out, blockIdx.y * gridDim.x + blockIdx.x);
}

// dx as (hi, lo) fp16 planes (scale from the bound gn_bwd_finalize_kernel left in `rec`) + per-block column sums of dx (the bias
// gradient of the convolution in front of the norm). A thread owns 8 channels: 16-byte plane stores; its channel group is fixed
// ((gridDim.x * 256) % (C / 8) == 0), so the column sums stay in registers until the block reduces them.
template <typename XT = float, typename DT = float>
__global__ __launch_bounds__(256) void gn_bwd_apply_planes_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                                   const float* __restrict__ cb, const float* __restrict__ gb,
                                                                   _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                                   float* __restrict__ scale_out, const float* __restrict__ rec,
                                                                   double* __restrict__ csp, int64_t S, int C, int cg, int G, int silu) {
  __shared__ double red[256][9];
  const int n = blockIdx.y;
  const int C8 = C >> 3;
  const int64_t total8 = S * C8;
  const bool single = lo == nullptr;                 // one bf16 plane, no scale (WDNO_CONV_MATH=bf16)
  const float s = single ? 1.0f : scale_from_amax(amax_record_read(rec));
  if (!single && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale_out[0] = s;
  const int64_t xe = (int64_t)n * S * C;
  _Float16* hp = hi + (int64_t)n * S * C;
  _Float16* lp = single ? nullptr : lo + (int64_t)n * S * C;
  const float4* cbp = reinterpret_cast<const float4*>(cb + (int64_t)n * C * 4);
  const float4* gbp = reinterpret_cast<const float4*>(gb + (int64_t)n * G * 4);
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int c0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % C8) * 8;
  float4 kk[8], gq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { kk[j] = cbp[c0 + j]; gq[j] = gbp[(c0 + j) / cg]; }
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += stride) {
    float xv[8], dv[8];
    gn_ld8<XT>(x, xe + i * 8, xv);
    gn_ld8<DT>(dy, xe + i * 8, dv);
    gn_half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dz = dv[j];
      if (silu) dz *= silu_grad_f(kk[j].x * xv[j] + kk[j].y);
      const float xh = (xv[j] - gq[j].x) * gq[j].y;
      const float o = kk[j].z * dz - gq[j].z - xh * gq[j].w;
      acc[j] += (double)o;
      _Float16 th, tl;
      plane_pack(o, s, single, th, tl);
      h[j] = th; l[j] = tl;
    }
    *reinterpret_cast<gn_half8*>(hp + i * 8) = h;
    if (!single) *reinterpret_cast<gn_half8*>(lp + i * 8) = l;
  }
  // block reduction of the column sums: threads with the same channel group are C8 apart
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
  __syncthreads();
  if (threadIdx.x < C8) {
    for (int t = threadIdx.x + C8; t < 256; t += C8)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += red[t][j];
    double* o = csp + ((int64_t)n * gridDim.x + blockIdx.x) * C + c0;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = acc[j];
  }
}

// ---------------------------------------------------------------------------------------------- host
// ws layout: [ double part[N][nchunk][C][2] | float cb[N][C][4] | float gb[N][G][4] ]
extern "C" size_t wdno_groupnorm_ws_bytes(int64_t N, int64_t S, int C, int G) {
  size_t part = (size_t)N * gn_chunks(S, N, C) * C * 2 * sizeof(double);
  return part + (size_t)N * C * 4 * sizeof(float) + (size_t)N * G * 4 * sizeof(float) + 64;
}
// `stats` holds, behind the [N][G] (mean, rstd) pairs, the two tables the forward finalize derives from them -- cb [N][C][4] (per-channel affine)
// and gb [N][G][4] -- so that the backward reads them instead of re-deriving them with one more 8-block launch per norm (30 x 6.7 us per smoke
// training step; a launch that small costs its full duration, measured by removing the equally small column sums: -0.13 ms per step).
extern "C" size_t wdno_groupnorm_stats_floats(int64_t N, int C, int G) { return (size_t)N * G * 2 + (size_t)N * C * 4 + (size_t)N * G * 4; }
static inline float* gn_cb(float* stats, int64_t N, int G) { return stats + (size_t)N * G * 2; }
static inline float* gn_gb(float* stats, int64_t N, int C, int G) { return stats + (size_t)N * G * 2 + (size_t)N * C * 4; }
// slices of groups per sample for the finalize kernels: every group its own block while the groups are narrow (the statistics of wide groups --
// GroupNorm(1, C) of the Burgers U-Net -- are reduced by a whole block)
static inline unsigned gn_slices(int C, int G) { return (C / G > 64 || wdno_debug_mode == 71) ? 1u : (unsigned)(G < 32 ? G : 32); }      // debug 71: one block per sample (round 5, the A/B)
static int gn_check(int64_t N, int64_t S, int C, int G) {
  if (N <= 0 || S <= 0 || C <= 0 || G <= 0 || N > 65535) return WDNO_EINVAL;
  if ((C & 3) || C > GN_MAXC || (C % G) != 0) return WDNO_EUNSUPPORTED;
  return WDNO_OK;
}
// x_bf16 / dy_bf16 (the _t entry points): the tensor is bf16 storage (gn_bf16); arithmetic and every other operand unchanged
#define GN_PARTIAL0(XB, ...) do { if (XB) gn_partial_kernel<0, gn_bf16><<<__VA_ARGS__; else gn_partial_kernel<0, float><<<__VA_ARGS__; } while (0)
extern "C" int wdno_groupnorm_act_fwd_amax_t(const void* x, int x_bf16, const float* gamma, const float* beta, const float* ss, float* y,
                                      float* stats, float* amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                      void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = gn_check(N, S, C, G);
  if (rc) return rc;
  if (ws_bytes < wdno_groupnorm_ws_bytes(N, S, C, G)) return WDNO_EWORKSPACE;
  const int nchunk = gn_chunks(S, N, C);
  double* part = (double*)ws;
  float* cb = gn_cb(stats, N, G);
  float* gb = gn_gb(stats, N, C, G);
  const int txp = pow2ceil(C / 4);
  const int64_t rpc = cdiv64(S, nchunk);
  hipStream_t st = as_stream(s);
  for (int rep_ = 0; rep_ < (wdno_debug_mode == 58 ? 2 : 1); ++rep_)
    GN_PARTIAL0(x_bf16, dim3(nchunk, (unsigned)N), 256, 0, st>>>(x, nullptr, nullptr, nullptr, part, S, C, C / G, G, txp, rpc, 0));
  gn_finalize_kernel<<<dim3((unsigned)N, gn_slices(C, G)), 256, 0, st>>>(part, nullptr, gamma, beta, ss, stats, cb, gb, S, C, G, nchunk, eps);
  int gx = stream_grid(S * (C / 4), 256);
  if (gx > 512) gx = 512;
  if (x_bf16) gn_apply_kernel<gn_bf16><<<dim3(gx, (unsigned)N), 256, 0, st>>>(x, cb, y, S, C, silu, amax_rec);
  else gn_apply_kernel<float><<<dim3(gx, (unsigned)N), 256, 0, st>>>(x, cb, y, S, C, silu, amax_rec);
  return wdno_check_launch();
}
extern "C" int wdno_groupnorm_act_fwd_amax(const float* x, const float* gamma, const float* beta, const float* ss, float* y,
                                      float* stats, float* amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                      void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_groupnorm_act_fwd_amax_t(x, 0, gamma, beta, ss, y, stats, amax_rec, N, S, C, G, eps, silu, ws, ws_bytes, s);
}
extern "C" int wdno_groupnorm_act_fwd(const float* x, const float* gamma, const float* beta, const float* ss, float* y,
                                      float* stats, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                      void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_groupnorm_act_fwd_amax(x, gamma, beta, ss, y, stats, nullptr, N, S, C, G, eps, silu, ws, ws_bytes, s);
}
extern "C" int wdno_groupnorm_act_bwd_amax(const float* x, const float* dy, const float* gamma, const float* beta, const float* ss,
                                      const float* stats, float* dx, float* dgb_partial, float* dss, float* amax_rec,
                                      int64_t N, int64_t S, int C, int G, int silu, void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = gn_check(N, S, C, G);
  if (rc) return rc;
  if (ws_bytes < wdno_groupnorm_ws_bytes(N, S, C, G)) return WDNO_EWORKSPACE;
  const int nchunk = gn_chunks(S, N, C);
  double* part = (double*)ws;
  const float* cb = gn_cb(const_cast<float*>(stats), N, G);              // the forward's tables (wdno_groupnorm_stats_floats)
  float* gb = gn_gb(const_cast<float*>(stats), N, C, G);      // entries 2, 3 of a group are this backward's own (the two group means): written by gn_bwd_finalize_kernel
  const int txp = pow2ceil(C / 4);
  const int64_t rpc = cdiv64(S, nchunk);
  hipStream_t st = as_stream(s);
  for (int rep_ = 0; rep_ < (wdno_debug_mode == 59 ? 2 : 1); ++rep_) gn_partial_kernel<1><<<dim3(nchunk, (unsigned)N), 256, 0, st>>>(x, dy, cb, gb, part, S, C, C / G, G, txp, rpc, silu);
  gn_bwd_finalize_kernel<<<dim3((unsigned)N, gn_slices(C, G)), 256, 0, st>>>(part, gamma, beta, ss, gb, dgb_partial, dss, S, C, G, nchunk);
  int gx = stream_grid(S * (C / 4), 256);
  if (gx > 512) gx = 512;
  gn_bwd_apply_kernel<<<dim3(gx, (unsigned)N), 256, 0, st>>>(x, dy, cb, gb, dx, S, C, C / G, G, silu, amax_rec);
  return wdno_check_launch();
}
extern "C" int wdno_groupnorm_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* ss,
                                      const float* stats, float* dx, float* dgb_partial, float* dss,
                                      int64_t N, int64_t S, int C, int G, int silu, void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_groupnorm_act_bwd_amax(x, dy, gamma, beta, ss, stats, dx, dgb_partial, dss, nullptr, N, S, C, G, silu, ws, ws_bytes, s);
}


extern "C" size_t wdno_groupnorm_fwd_planes_ws_bytes(int64_t N, int64_t S, int C, int G) {
  return wdno_groupnorm_ws_bytes(N, S, C, G) + (size_t)N * gn_chunks(S, N, C) * 2 * sizeof(float) + 128;
}
extern "C" int wdno_groupnorm_act_fwd_planes_t(const void* x, int x_bf16, const float* gamma, const float* beta, const float* ss, void* y_hi, void* y_lo,
                                             float* y_scale, float* stats, float* bound_rec, int64_t N, int64_t S, int C, int G, float eps,
                                             int silu, void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = gn_check(N, S, C, G);
  if (rc) return rc;
  const int C8 = C / 8;
  if ((C & 7) || C8 > 256 || (C8 & (C8 - 1))) return WDNO_EUNSUPPORTED;
  if (ws_bytes < wdno_groupnorm_fwd_planes_ws_bytes(N, S, C, G)) return WDNO_EWORKSPACE;
  const int nchunk = gn_chunks(S, N, C);
  double* part = (double*)ws;
  float* cb = gn_cb(stats, N, G);
  float* gb = gn_gb(stats, N, C, G);
  float* mx = (float*)((char*)ws + ((wdno_groupnorm_ws_bytes(N, S, C, G) + 63) & ~(size_t)63));
  const int txp = pow2ceil(C / 4);
  const int64_t rpc = cdiv64(S, nchunk);
  hipStream_t st = as_stream(s);
  for (int rep_ = 0; rep_ < (wdno_debug_mode == 58 ? 2 : 1); ++rep_)
    GN_PARTIAL0(x_bf16, dim3(nchunk, (unsigned)N), 256, 0, st>>>(x, nullptr, nullptr, nullptr, part, S, C, C / G, G, txp, rpc, 0, y_lo ? mx : nullptr));
  gn_finalize_kernel<<<dim3((unsigned)N, gn_slices(C, G)), 256, 0, st>>>(part, nullptr, gamma, beta, ss, stats, cb, gb, S, C, G, nchunk, eps, mx, y_lo ? bound_rec : nullptr);
  int gx = stream_grid(S * (C / 8), 256);
  if (gx > 512) gx = 512;
  if (x_bf16) gn_apply_planes_kernel<gn_bf16><<<dim3(gx, (unsigned)N), 256, 0, st>>>(x, cb, (_Float16*)y_hi, (_Float16*)y_lo, y_scale, bound_rec, S, C, silu);
  else gn_apply_planes_kernel<float><<<dim3(gx, (unsigned)N), 256, 0, st>>>(x, cb, (_Float16*)y_hi, (_Float16*)y_lo, y_scale, bound_rec, S, C, silu);
  return wdno_check_launch();
}
extern "C" int wdno_groupnorm_act_fwd_planes(const float* x, const float* gamma, const float* beta, const float* ss, void* y_hi, void* y_lo,
                                             float* y_scale, float* stats, float* bound_rec, int64_t N, int64_t S, int C, int G, float eps,
                                             int silu, void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_groupnorm_act_fwd_planes_t(x, 0, gamma, beta, ss, y_hi, y_lo, y_scale, stats, bound_rec, N, S, C, G, eps, silu, ws, ws_bytes, s);
}

extern "C" int wdno_groupnorm_act_add_fwd_planes_t(const void* x, int x_bf16, const float* gamma, const float* beta, const float* ss, const float* residual,
                                                 const float* res_rec, float* y, void* y_hi, void* y_lo, float* y_scale, float* stats,
                                                 float* bound_rec, float* y_amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                                 void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = gn_check(N, S, C, G);
  if (rc) return rc;
  const int C8 = C / 8;
  if ((C & 7) || C8 > 256 || (C8 & (C8 - 1))) return WDNO_EUNSUPPORTED;
  if (!residual || !y || (y_lo && (!y_hi || !res_rec || !bound_rec || !y_scale))) return WDNO_EINVAL;      // y_hi == NULL: fp32 sum only
  if (ws_bytes < wdno_groupnorm_fwd_planes_ws_bytes(N, S, C, G)) return WDNO_EWORKSPACE;
  const int nchunk = gn_chunks(S, N, C);
  double* part = (double*)ws;
  float* cb = gn_cb(stats, N, G);
  float* gb = gn_gb(stats, N, C, G);
  float* mx = (float*)((char*)ws + ((wdno_groupnorm_ws_bytes(N, S, C, G) + 63) & ~(size_t)63));
  const int txp = pow2ceil(C / 4);
  const int64_t rpc = cdiv64(S, nchunk);
  hipStream_t st = as_stream(s);
  for (int rep_ = 0; rep_ < (wdno_debug_mode == 58 ? 2 : 1); ++rep_)
    GN_PARTIAL0(x_bf16, dim3(nchunk, (unsigned)N), 256, 0, st>>>(x, nullptr, nullptr, nullptr, part, S, C, C / G, G, txp, rpc, 0, y_lo ? mx : nullptr));
  gn_finalize_kernel<<<dim3((unsigned)N, gn_slices(C, G)), 256, 0, st>>>(part, nullptr, gamma, beta, ss, stats, cb, gb, S, C, G, nchunk, eps, mx, y_lo ? bound_rec : nullptr);
  int gx = stream_grid(S * (C / 8), 256);
  if (gx > 512) gx = 512;
  if (x_bf16) gn_apply_add_planes_kernel<gn_bf16><<<dim3(gx, (unsigned)N), 256, 0, st>>>(x, cb, residual, y, (_Float16*)y_hi, (_Float16*)y_lo, y_scale, bound_rec,
                                                                                       res_rec, y_amax_rec, S, C, silu);
  else gn_apply_add_planes_kernel<float><<<dim3(gx, (unsigned)N), 256, 0, st>>>(x, cb, residual, y, (_Float16*)y_hi, (_Float16*)y_lo, y_scale, bound_rec, res_rec,
                                                                           y_amax_rec, S, C, silu);
  return wdno_check_launch();
}
extern "C" int wdno_groupnorm_act_add_fwd_planes(const float* x, const float* gamma, const float* beta, const float* ss, const float* residual,
                                                 const float* res_rec, float* y, void* y_hi, void* y_lo, float* y_scale, float* stats,
                                                 float* bound_rec, float* y_amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                                 void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_groupnorm_act_add_fwd_planes_t(x, 0, gamma, beta, ss, residual, res_rec, y, y_hi, y_lo, y_scale, stats, bound_rec, y_amax_rec, N, S, C, G, eps, silu,
                                             ws, ws_bytes, s);
}

__global__ __launch_bounds__(PRS_THREADS) void gn_bwd_tail_kernel(const double* __restrict__ csp, float* __restrict__ dx_colsum, int nb, int C, int nbx,
                                                                     const float* __restrict__ dgb, float* __restrict__ dgb_sum, int N) {
  if ((int)blockIdx.x < nbx) {
    partial_rows_sum_body<double>(csp, dx_colsum, nb, C, (int)blockIdx.x);
    return;
  }
  // sum over the samples of the per-sample (d gamma, d beta) pieces: 32 columns x 32 row groups per block, four sums per thread (one thread
  // per column walking all N rows was 256 dependent loads at batch 256: most of this kernel's 88 us there)
  partial_rows_sum_body<float>(dgb, dgb_sum, N, 2 * C, (int)blockIdx.x - nbx);
}

/* ---- backward with dx delivered as fp16 (hi, lo) planes: see include/wdno_hip.h ---- */
static inline int gn_planes_grid(int64_t N, int64_t S, int C) {
  int gx = stream_grid(S * (C / 8), 256);
  int cap = (int)(2048 / N);         // blocks per sample: every block leaves one row of column-sum partials, ~2048 rows in total
  if (cap < 2) cap = 2;              // (at batch 256 the final reduction over 32768 rows cost 36 us per GroupNorm)
  if (cap > 128) cap = 128;
  if (gx > cap) gx = cap;
  return gx;
}
extern "C" size_t wdno_groupnorm_bwd_planes_ws_bytes(int64_t N, int64_t S, int C, int G) {
  return wdno_groupnorm_ws_bytes(N, S, C, G) + (size_t)N * gn_chunks(S, N, C) * 2 * sizeof(float) + (size_t)N * 128 * C * sizeof(double) + 64;
}
// where the column-sum partials of dx live inside ws (doubles [rows][C]) when the backward is called with dx_colsum == NULL
extern "C" int wdno_groupnorm_bwd_planes_tail(int64_t N, int64_t S, int C, int G, size_t* csp_offset, int* rows) {
  int rc = gn_check(N, S, C, G);
  if (rc) return rc;
  WDNO_REQUIRE(csp_offset && rows);
  *csp_offset = (wdno_groupnorm_ws_bytes(N, S, C, G) + 63) & ~(size_t)63;
  *rows = (int)N * gn_planes_grid(N, S, C);
  return WDNO_OK;
}
extern "C" int wdno_groupnorm_act_bwd_planes_t(const void* x, int x_bf16, const void* dy, int dy_bf16, const float* gamma, const float* beta, const float* ss,
                                             const float* stats, void* dx_hi, void* dx_lo, float* dx_scale, float* dx_colsum,
                                             float* dgb_partial, float* dgb_sum, float* dss, float* bound_rec, int64_t N, int64_t S, int C, int G, int silu,
                                             void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = gn_check(N, S, C, G);
  if (rc) return rc;
  const int C8 = C / 8;
  if ((C & 7) || C8 > 256 || (C8 & (C8 - 1))) return WDNO_EUNSUPPORTED;     // a thread keeps one group of 8 channels
  if (ws_bytes < wdno_groupnorm_bwd_planes_ws_bytes(N, S, C, G)) return WDNO_EWORKSPACE;
  const int nchunk = gn_chunks(S, N, C);
  double* part = (double*)ws;
  const float* cb = gn_cb(const_cast<float*>(stats), N, G);
  float* gb = gn_gb(const_cast<float*>(stats), N, C, G);      // entries 2, 3 of a group are this backward's own (the two group means): written by gn_bwd_finalize_kernel
  char* tail = (char*)ws + ((wdno_groupnorm_ws_bytes(N, S, C, G) + 63) & ~(size_t)63);
  double* csp = (double*)tail;
  float* mx = (float*)(csp + (size_t)N * 128 * C);
  const int txp = pow2ceil(C / 4);
  const int64_t rpc = cdiv64(S, nchunk);
  hipStream_t st = as_stream(s);
#define GN_BWD_TYPES(KERNEL, ...) do { \
    if (x_bf16 && dy_bf16) KERNEL(gn_bf16, gn_bf16, __VA_ARGS__); else if (x_bf16) KERNEL(gn_bf16, float, __VA_ARGS__); \
    else if (dy_bf16) KERNEL(float, gn_bf16, __VA_ARGS__); else KERNEL(float, float, __VA_ARGS__); } while (0)
#define GN_K_PARTIAL1(XT, DT, ...) gn_partial_kernel<1, XT, DT><<<dim3(nchunk, (unsigned)N), 256, 0, st>>>(__VA_ARGS__)
#define GN_K_BWD_APPLY(XT, DT, ...) gn_bwd_apply_planes_kernel<XT, DT><<<dim3(gx, (unsigned)N), 256, 0, st>>>(__VA_ARGS__)
  for (int rep_ = 0; rep_ < (wdno_debug_mode == 59 ? 2 : 1); ++rep_)
    GN_BWD_TYPES(GN_K_PARTIAL1, x, dy, cb, gb, part, S, C, C / G, G, txp, rpc, silu, dx_lo ? mx : nullptr);
  gn_bwd_finalize_kernel<<<dim3((unsigned)N, gn_slices(C, G)), 256, 0, st>>>(part, gamma, beta, ss, gb, dgb_partial, dss, S, C, G, nchunk, cb, mx, dx_lo ? bound_rec : nullptr);
  const int gx = gn_planes_grid(N, S, C);
  GN_BWD_TYPES(GN_K_BWD_APPLY, x, dy, cb, gb, (_Float16*)dx_hi, (_Float16*)dx_lo, dx_scale, bound_rec, csp, S, C, C / G, G, silu);
  // one launch for both reductions that end the backward: the column sums of dx (partials of the apply pass) and, when dgb_sum is given, the
  // sum over the samples of the per-sample parameter-gradient pieces (was a colsum_rows launch of the caller: same fp64 sum in row order)
  // (dx_colsum == NULL: the caller sums the partials later with everything else that ends the backward -- wdno_rows_sum_multi over the
  // N * gx rows at wdno_groupnorm_bwd_planes_tail's offset of ws, and over the N rows of dgb_partial)
  const int nbx = dx_colsum ? cdiv(C, 32) : 0, nby = dgb_sum ? cdiv(2 * C, 32) : 0;
  if (nbx + nby) gn_bwd_tail_kernel<<<nbx + nby, PRS_THREADS, 0, st>>>(csp, dx_colsum, (int)N * gx, C, nbx, dgb_partial, dgb_sum, (int)N);
  return wdno_check_launch();
}

extern "C" int wdno_groupnorm_act_bwd_planes(const float* x, const float* dy, const float* gamma, const float* beta, const float* ss,
                                             const float* stats, void* dx_hi, void* dx_lo, float* dx_scale, float* dx_colsum,
                                             float* dgb_partial, float* dgb_sum, float* dss, float* bound_rec, int64_t N, int64_t S, int C, int G, int silu,
                                             void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_groupnorm_act_bwd_planes_t(x, 0, dy, 0, gamma, beta, ss, stats, dx_hi, dx_lo, dx_scale, dx_colsum, dgb_partial, dgb_sum, dss, bound_rec, N, S, C, G,
                                         silu, ws, ws_bytes, s);
}

// ---------------------------------------------------------------------------------------------- channel LayerNorm
// rows [P][C]; a row is handled by TPR lanes holding VPL float4 each. y = (x - mean) / sqrt(var + eps) * g
template <int TPR, int VPL, bool BWD>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ dy, float* __restrict__ out,
                                                         float* __restrict__ dg_part, int64_t P, int C, float eps,
                                                         const float* __restrict__ add_to, float* __restrict__ dx_amax,
                                                         _Float16* __restrict__ y_hi = nullptr, _Float16* __restrict__ y_lo = nullptr,
                                                         float* __restrict__ y_scale = nullptr) {
  constexpr int RPB = 256 / TPR;
  __shared__ float red[BWD ? 256 * VPL * 4 : 1];
  const int C4 = C >> 2;
  const int lane = threadIdx.x % TPR, rloc = threadIdx.x / TPR;
  float4 gv[VPL];
  bool cok[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    int c4 = lane + v * TPR;
    cok[v] = c4 < C4;
    gv[v] = cok[v] ? reinterpret_cast<const float4*>(g)[c4] : make_float4(0, 0, 0, 0);
  }
  float dgacc[VPL][4];
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) dgacc[v][j] = 0.f;
  const float invC = 1.0f / (float)C;
  const int64_t rstride = (int64_t)gridDim.x * RPB;
  float ps = 1.0f;                 // forward with planes output: |y| = |xhat g| <= sqrt(C) max|g|, a scale every block derives by itself
  if (!BWD && y_hi && y_lo) {
    float mg = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) mg = amax4(mg, gv[v]);
    ps = scale_from_amax(sqrtf((float)C) * group_max<TPR>(mg));
    if (blockIdx.x == 0 && threadIdx.x == 0) y_scale[0] = ps;
  }
  float am = 0.f;       // max|y| (forward: left in the amax record dg_part points to, if any) / max|dx| (backward: in dx_amax, if any)
  // all lanes of a row group iterate together (uniform trip count per group)
  for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < P; r0 += rstride) {
    int64_t r = r0 + rloc;
    bool rok = r < P;
    float4 xv[VPL];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      xv[v] = (rok && cok[v]) ? reinterpret_cast<const float4*>(x + r * C)[lane + v * TPR] : make_float4(0, 0, 0, 0);
      s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
    }
    float mean = group_sum<TPR>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      if (cok[v]) {
        xv[v].x -= mean; xv[v].y -= mean; xv[v].z -= mean; xv[v].w -= mean;
        q += (xv[v].x * xv[v].x + xv[v].y * xv[v].y) + (xv[v].z * xv[v].z + xv[v].w * xv[v].w);
      }
    }
    float var = group_sum<TPR>(q) * invC;
    float rstd = 1.0f / sqrtf(var + eps);
    if (!BWD) {
#pragma unroll
      for (int v = 0; v < VPL; ++v)
        if (rok && cok[v]) {
          float4 o;
          o.x = xv[v].x * rstd * gv[v].x; o.y = xv[v].y * rstd * gv[v].y;
          o.z = xv[v].z * rstd * gv[v].z; o.w = xv[v].w * rstd * gv[v].w;
          if (y_hi) {
            typedef _Float16 half4v __attribute__((ext_vector_type(4)));
            const float t[4] = {o.x, o.y, o.z, o.w};
            half4v h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) { _Float16 th, tl; plane_pack(t[j], ps, y_lo == nullptr, th, tl); h[j] = th; l[j] = tl; }
            reinterpret_cast<half4v*>(y_hi + r * C)[lane + v * TPR] = h;
            if (y_lo) reinterpret_cast<half4v*>(y_lo + r * C)[lane + v * TPR] = l;
            continue;
          }
          reinterpret_cast<float4*>(out + r * C)[lane + v * TPR] = o;
          am = amax4(am, o);
        }
    } else {
      float4 dv[VPL];
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        dv[v] = (rok && cok[v]) ? reinterpret_cast<const float4*>(dy + r * C)[lane + v * TPR] : make_float4(0, 0, 0, 0);
        // xhat in xv, dxhat = dy * g
        xv[v].x *= rstd; xv[v].y *= rstd; xv[v].z *= rstd; xv[v].w *= rstd;
        dgacc[v][0] += dv[v].x * xv[v].x; dgacc[v][1] += dv[v].y * xv[v].y;
        dgacc[v][2] += dv[v].z * xv[v].z; dgacc[v][3] += dv[v].w * xv[v].w;
        dv[v].x *= gv[v].x; dv[v].y *= gv[v].y; dv[v].z *= gv[v].z; dv[v].w *= gv[v].w;
        m1 += (dv[v].x + dv[v].y) + (dv[v].z + dv[v].w);
        m2 += (dv[v].x * xv[v].x + dv[v].y * xv[v].y) + (dv[v].z * xv[v].z + dv[v].w * xv[v].w);
      }
      m1 = group_sum<TPR>(m1) * invC;
      m2 = group_sum<TPR>(m2) * invC;
#pragma unroll
      for (int v = 0; v < VPL; ++v)
        if (rok && cok[v]) {
          float4 o;
          o.x = rstd * (dv[v].x - m1 - xv[v].x * m2); o.y = rstd * (dv[v].y - m1 - xv[v].y * m2);
          o.z = rstd * (dv[v].z - m1 - xv[v].z * m2); o.w = rstd * (dv[v].w - m1 - xv[v].w * m2);
          if (add_to) {          // the gradient arriving at x over the skip connection of Residual(PreNorm(fn)): one add launch less
            const float4 t = reinterpret_cast<const float4*>(add_to + r * C)[lane + v * TPR];
            o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
          }
          reinterpret_cast<float4*>(out + r * C)[lane + v * TPR] = o;
          am = amax4(am, o);
        }
    }
  }
  if (!BWD && dg_part) amax_record_emit(am, dg_part, blockIdx.x);
  if (BWD && dx_amax) amax_record_emit(am, dx_amax, blockIdx.x);
  if (BWD) {
    // reduce dg over the RPB row groups of the block, write one partial row per block
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((rloc * TPR + lane) * VPL + v) * 4 + j] = dgacc[v][j];
    __syncthreads();
    if (rloc == 0) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        if (!cok[v]) continue;
        float a[4] = {0, 0, 0, 0};
        for (int t = 0; t < RPB; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] += red[((t * TPR + lane) * VPL + v) * 4 + j];
        reinterpret_cast<float4*>(dg_part + (int64_t)blockIdx.x * C)[lane + v * TPR] = make_float4(a[0], a[1], a[2], a[3]);
      }
    }
  }
}

static inline int ln_blocks(int64_t P, int rpb) {
  int64_t nb = cdiv64(P, rpb);
  if (nb > 1024) nb = 1024;
  return (int)nb;
}
static inline void ln_shape(int C, int& tpr, int& vpl) {
  int c4 = C / 4;
  if (c4 <= 64) { tpr = pow2ceil(c4); if (tpr < 2) tpr = 2; vpl = 1; }
  else { tpr = 64; vpl = pow2ceil(cdiv(c4, 64)); }
}
template <bool BWD>
static int ln_launch(const float* x, const float* g, const float* dy, float* out, float* dgp, int64_t P, int C, float eps, hipStream_t st,
                     const float* add_to = nullptr, float* dx_amax = nullptr, _Float16* y_hi = nullptr, _Float16* y_lo = nullptr,
                     float* y_scale = nullptr) {
  int tpr, vpl;
  ln_shape(C, tpr, vpl);
  int nb = ln_blocks(P, 256 / tpr);
#define LN_CASE(T, V) layernorm_kernel<T, V, BWD><<<nb, 256, 0, st>>>(x, g, dy, out, dgp, P, C, eps, add_to, dx_amax, y_hi, y_lo, y_scale)
  if (vpl == 1) {
    switch (tpr) {
      case 2: LN_CASE(2, 1); break;
      case 4: LN_CASE(4, 1); break;
      case 8: LN_CASE(8, 1); break;
      case 16: LN_CASE(16, 1); break;
      case 32: LN_CASE(32, 1); break;
      default: LN_CASE(64, 1); break;
    }
  } else if (vpl == 2) LN_CASE(64, 2);
  else if (vpl == 4) LN_CASE(64, 4);
  else return WDNO_EUNSUPPORTED;
#undef LN_CASE
  return WDNO_OK;
}
extern "C" int wdno_layernorm_fwd_amax(const float* x, const float* g, float* y, float* amax_rec, int64_t P, int C, float eps,
                                       wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && C >= 4);
  if ((C & 3) || C > 1024) return WDNO_EUNSUPPORTED;
  int rc = ln_launch<false>(x, g, nullptr, y, amax_rec, P, C, eps, as_stream(s));
  if (rc) return rc;
  return wdno_check_launch();
}
extern "C" int wdno_layernorm_fwd_planes(const float* x, const float* g, void* y_hi, void* y_lo, float* y_scale, int64_t P, int C, float eps,
                                         wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && C >= 8);
  if ((C & 7) || C > 1024) return WDNO_EUNSUPPORTED;
  int rc = ln_launch<false>(x, g, nullptr, nullptr, nullptr, P, C, eps, as_stream(s), nullptr, nullptr, (_Float16*)y_hi, (_Float16*)y_lo, y_scale);
  if (rc) return rc;
  return wdno_check_launch();
}
extern "C" int wdno_layernorm_fwd(const float* x, const float* g, float* y, int64_t P, int C, float eps, wdno_stream_t s) {
  return wdno_layernorm_fwd_amax(x, g, y, nullptr, P, C, eps, s);
}
extern "C" size_t wdno_layernorm_bwd_ws_bytes(int64_t P, int C) { return (size_t)1024 * C * sizeof(float); }
extern "C" int wdno_layernorm_bwd_add_amax(const float* x, const float* g, const float* dy, const float* add_to, float* dx, float* dg,
                                           float* amax_rec, int64_t P, int C, float eps, void* ws, size_t ws_bytes, wdno_stream_t s) {
  WDNO_REQUIRE(P > 0 && C >= 4);
  if ((C & 3) || C > 1024) return WDNO_EUNSUPPORTED;
  if (ws_bytes < wdno_layernorm_bwd_ws_bytes(P, C)) return WDNO_EWORKSPACE;
  int tpr, vpl;
  ln_shape(C, tpr, vpl);
  int nb = ln_blocks(P, 256 / tpr);
  int rc = ln_launch<true>(x, g, dy, dx, (float*)ws, P, C, eps, as_stream(s), add_to, amax_rec);
  if (rc) return rc;
  partial_rows_sum_kernel<float><<<cdiv(C, 32), PRS_THREADS, 0, as_stream(s)>>>((const float*)ws, dg, nb, C);
  return wdno_check_launch();
}
extern "C" int wdno_layernorm_bwd_add(const float* x, const float* g, const float* dy, const float* add_to, float* dx, float* dg, int64_t P,
                                      int C, float eps, void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_layernorm_bwd_add_amax(x, g, dy, add_to, dx, dg, nullptr, P, C, eps, ws, ws_bytes, s);
}
extern "C" int wdno_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* dg, int64_t P, int C,
                                  float eps, void* ws, size_t ws_bytes, wdno_stream_t s) {
  return wdno_layernorm_bwd_add(x, g, dy, nullptr, dx, dg, P, C, eps, ws, ws_bytes, s);
}
