// dwt.hip -- single-level separable wavelet filter banks on gfx950 (HBM-bound stencil passes).
//
// Every transform on the WDNO path is a tensor product of two 1-axis primitives:
//   analysis : y_b[k] = sum_m f_b[L-1-m] * X(2k + m - off)            (b = lo, hi; stride-2 correlation)
//   synthesis: x[n]   = sum_{m : (n+off-m) even} lo[K] g_lo[m] + hi[K] g_hi[m],  K = (n+off-m)/2
// with off = L-2 and zero extension (mode 'zero'), or off = L/2-1 and periodic extension (mode
// 'periodization', odd lengths extended by repeating the last sample). The adjoints needed for guidance
// back-propagation are the same two primitives with reversed taps.
//
// One launch per axis; the band index and the (possibly padded) coefficient packing of coef_to_tensor are
// folded into the store/load addressing of the pass that touches the coefficient tensor, so no separate
// packing kernel exists. Taps travel as kernel arguments (<= 16 per filter).
#include "common.h"

#define WDNO_MAXL 16
#define WDNO_MAXCOMP 5

struct Taps { float lo[WDNO_MAXL]; float hi[WDNO_MAXL]; };

// mixed-radix outer index -> two linear addresses
struct OuterMap {
  int ncomp;
  int n[WDNO_MAXCOMP];
  int64_t sa[WDNO_MAXCOMP];  // strides on the "signal side" tensor
  int64_t sc[WDNO_MAXCOMP];  // strides on the "coefficient side" tensor
};

struct AxisPass {
  OuterMap om;
  int64_t outer;        // product of om.n
  int N;                // signal length along the axis (logical, before odd extension)
  int M;                // coefficient length along the axis
  int inner;            // contiguous trailing extent shared by both sides
  int64_t sig_stride;   // stride of the axis on the signal side (elements)
  int64_t coef_stride;  // stride of the axis on the coefficient side
  int64_t band_stride;  // stride between lo and hi on the coefficient side
  int mode, L, off;
  int odd;              // periodization with odd N: extended length N+1
};

__device__ __forceinline__ void outer_addr(const OuterMap& om, int64_t o, int64_t& a, int64_t& c) {
  a = 0; c = 0;
#pragma unroll
  for (int i = WDNO_MAXCOMP - 1; i >= 0; --i) {
    if (i < om.ncomp) {
      int64_t q = o / om.n[i];
      int r = (int)(o - q * om.n[i]);
      a += r * om.sa[i];
      c += r * om.sc[i];
      o = q;
    }
  }
}

// signal -> (lo, hi)
__global__ __launch_bounds__(256) void analysis_kernel(const float* __restrict__ sig, float* __restrict__ coef, AxisPass p, Taps t) {
  int64_t total = p.outer * p.M * p.inner;
  int64_t stride = (int64_t)gridDim.x * 256;
  int Next = p.N + p.odd;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    int i = (int)(idx % p.inner);
    int64_t r = idx / p.inner;
    int k = (int)(r % p.M);
    int64_t o = r / p.M;
    int64_t a, c;
    outer_addr(p.om, o, a, c);
    const float* sp = sig + a + i;
    float lo = 0.f, hi = 0.f;
    int j0 = 2 * k - p.off;
    for (int m = 0; m < p.L; ++m) {
      int j = j0 + m;
      float v;
      if (p.mode == 1) {
        v = (j >= 0 && j < p.N) ? sp[(int64_t)j * p.sig_stride] : 0.f;
      } else {
        j %= Next;
        if (j < 0) j += Next;
        if (j > p.N - 1) j = p.N - 1;
        v = sp[(int64_t)j * p.sig_stride];
      }
      lo = fmaf(t.lo[p.L - 1 - m], v, lo);
      hi = fmaf(t.hi[p.L - 1 - m], v, hi);
    }
    float* cp = coef + c + (int64_t)k * p.coef_stride + i;
    cp[0] = lo;
    cp[p.band_stride] = hi;
  }
}

// (lo, hi) -> signal. For the adjoint of a periodized analysis of odd length (odd = 1) the sample of the
// repeated position N is folded back onto N-1.
__device__ __forceinline__ float synth_point(const float* __restrict__ cp, const AxisPass& p, const Taps& t, int n) {
  float acc = 0.f;
  int base = n + p.off;
  int Next = p.N + p.odd;   // periodization period on the signal side
  for (int m = (base & 1); m < p.L; m += 2) {
    int kk = base - m;      // even
    if (p.mode == 1) {
      kk >>= 1;             // arithmetic shift; negative stays negative
      if (kk < 0 || kk >= p.M) continue;
    } else {
      kk %= Next;
      if (kk < 0) kk += Next;
      kk >>= 1;
    }
    const float* q = cp + (int64_t)kk * p.coef_stride;
    acc = fmaf(q[0], t.lo[m], acc);
    acc = fmaf(q[p.band_stride], t.hi[m], acc);
  }
  return acc;
}
__global__ __launch_bounds__(256) void synthesis_kernel(const float* __restrict__ coef, float* __restrict__ sig, AxisPass p, Taps t) {
  int64_t total = p.outer * p.N * p.inner;
  int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    int i = (int)(idx % p.inner);
    int64_t r = idx / p.inner;
    int n = (int)(r % p.N);
    int64_t o = r / p.N;
    int64_t a, c;
    outer_addr(p.om, o, a, c);
    const float* cp = coef + c + i;
    float v = synth_point(cp, p, t, n);
    if (p.odd && n == p.N - 1) v += synth_point(cp, p, t, p.N);
    sig[a + (int64_t)n * p.sig_stride + i] = v;
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int validate(const wdno_dwt_desc* d) {
  if (!d || d->nd < 1 || d->nd > 3 || (d->mode != 0 && d->mode != 1)) return WDNO_EINVAL;
  if (d->L < 2 || d->L > WDNO_MAXL || (d->L & 1) || d->n_img <= 0) return WDNO_EINVAL;
  for (int a = 3 - d->nd; a < 3; ++a) {
    int N = d->in_dims[a], M = d->out_dims[a];
    if (N <= 0 || M <= 0) return WDNO_EINVAL;
    if (d->mode == 0) {
      if (M != (N + 1) / 2 || N + (N & 1) < d->L) return WDNO_EUNSUPPORTED;
    }
    // mode zero: forward has M == (N+L-1)/2 ; inverse/adjoints may be called with the natural N = 2M-L+2
  }
  return WDNO_OK;
}

struct Plan {
  int nax;
  int ax[3];           // axis ids (0=T,1=H,2=W in the 3-slot convention), in forward (analysis) order
  int64_t sig_elems[4];   // number of elements of the intermediate tensor before pass i (i=0: the signal)
};

// Intermediate layouts (forward order T, H, W; nb = number of bands produced so far):
//   after pass T : [img][bt][To][H][W]
//   after pass H : [img][bt][To][bh][Ho][W]
//   after pass W : the coefficient tensor (strided), band = bt*4 + bh*2 + bw (missing axes contribute 0 bits)
static size_t ws_elems(const wdno_dwt_desc* d) {
  int T = d->in_dims[0], H = d->in_dims[1], W = d->in_dims[2];
  int To = d->out_dims[0], Ho = d->out_dims[1];
  if (d->nd == 1) return 0;
  if (d->nd == 2) return (size_t)d->n_img * 2 * Ho * W;
  size_t t1 = (size_t)d->n_img * 2 * To * H * W;
  size_t t2 = (size_t)d->n_img * 2 * To * 2 * Ho * W;
  return t1 + t2;
}
extern "C" size_t wdno_dwt_ws_bytes(const wdno_dwt_desc* d) {
  if (validate(d) != WDNO_OK) return 0;
  return ws_elems(d) * sizeof(float) + 256;
}

static void fill_taps(Taps& t, const float* lo, const float* hi, int L, bool reverse) {
  for (int i = 0; i < WDNO_MAXL; ++i) { t.lo[i] = 0.f; t.hi[i] = 0.f; }
  for (int i = 0; i < L; ++i) {
    t.lo[i] = reverse ? lo[L - 1 - i] : lo[i];
    t.hi[i] = reverse ? hi[L - 1 - i] : hi[i];
  }
}

static void set_outer(OuterMap& om, int ncomp, const int* n, const int64_t* sa, const int64_t* sc, int64_t& outer) {
  om.ncomp = ncomp;
  outer = 1;
  for (int i = 0; i < WDNO_MAXCOMP; ++i) {
    om.n[i] = i < ncomp ? n[i] : 1;
    om.sa[i] = i < ncomp ? sa[i] : 0;
    om.sc[i] = i < ncomp ? sc[i] : 0;
    if (i < ncomp) outer *= n[i];
  }
}

// Build the pass descriptors shared by all four entry points. `analysis_dir`: true when data flows
// signal -> coefficients (dwt_fwd, dwt_inv_adjoint), false for the reverse direction.
// sig = the signal-side tensor of the whole transform, coef = the packed coefficient tensor.
static int run(const float* src, float* dst, const wdno_dwt_desc* d, const float* fl, const float* fh, bool reverse_taps,
               bool analysis_dir, bool odd_rule, void* ws, size_t ws_bytes, hipStream_t st) {
  if (ws_elems(d) * sizeof(float) > ws_bytes) return WDNO_EWORKSPACE;
  Taps taps;
  fill_taps(taps, fl, fh, d->L, reverse_taps);
  const int T = d->in_dims[0], H = d->in_dims[1], W = d->in_dims[2];
  const int To = d->out_dims[0], Ho = d->out_dims[1], Wo = d->out_dims[2];
  const int img = d->n_img;
  const int off = d->mode == 1 ? d->L - 2 : d->L / 2 - 1;
  float* t1 = (float*)ws;                                   // [img][2][To][H][W]      (nd == 3)
  float* t2 = d->nd == 3 ? t1 + (size_t)img * 2 * To * H * W : t1;  // [img][bt][To][2][Ho][W]
  auto base = [&](int N, int M) {
    AxisPass p;
    p.N = N; p.M = M; p.mode = d->mode; p.L = d->L; p.off = off;
    p.odd = (d->mode == 0 && odd_rule && (N & 1)) ? 1 : 0;
    return p;
  };
  // contiguous signal strides
  const int64_t xs1 = W, xs0 = (int64_t)H * W, xsi = (int64_t)T * H * W;
  auto launch = [&](bool ana, const float* a_in, float* a_out, const AxisPass& p) {
    int64_t items = p.outer * (int64_t)(ana ? p.M : p.N) * p.inner;
    int grid = stream_grid(items, 256);
    if (ana) analysis_kernel<<<grid, 256, 0, st>>>(a_in, a_out, p, taps);
    else synthesis_kernel<<<grid, 256, 0, st>>>(a_in, a_out, p, taps);
  };

  // --- describe the three possible passes; "signal side" (sa) is the input of the analysis direction
  AxisPass pT, pH, pW;
  bool hasT = d->nd == 3, hasH = d->nd >= 2;
  if (hasT) {  // signal [img][T][H*W] <-> t1 [img][2][To][H*W]
    pT = base(T, To);
    int n[1] = {img};
    int64_t sa[1] = {xsi};
    int64_t sc[1] = {(int64_t)2 * To * H * W};
    set_outer(pT.om, 1, n, sa, sc, pT.outer);
    pT.inner = H * W; pT.sig_stride = xs0; pT.coef_stride = (int64_t)H * W; pT.band_stride = (int64_t)To * H * W;
  }
  const int nbt = hasT ? 2 : 1;            // bands already present before the H pass
  const int Tcur = hasT ? To : 1;
  if (hasH) {  // signal-side [img*nbt*Tcur][H][W] <-> t2 [img*nbt*Tcur][2][Ho][W]
    pH = base(H, Ho);
    int n[1] = {img * nbt * Tcur};
    int64_t sa[1] = {(int64_t)H * W};
    int64_t sc[1] = {(int64_t)2 * Ho * W};
    set_outer(pH.om, 1, n, sa, sc, pH.outer);
    pH.inner = W; pH.sig_stride = W; pH.coef_stride = W; pH.band_stride = (int64_t)Ho * W;
  }
  {  // W pass: signal-side rows (img, bt, to, bh, ho) of length W  <->  packed coefficient tensor
    pW = base(W, Wo);
    const int nbh = hasH ? 2 : 1;
    const int Hcur = hasH ? Ho : 1;
    int n[5] = {img, nbt, Tcur, nbh, Hcur};
    // signal side = t2 layout [img][bt][To][bh][Ho][W] (or the raw signal when nd == 1)
    int64_t sa[5] = {(int64_t)nbt * Tcur * nbh * Hcur * W, (int64_t)Tcur * nbh * Hcur * W, (int64_t)nbh * Hcur * W, (int64_t)Hcur * W, (int64_t)W};
    // band index: 3-D = bt*4 + bh*2 + bw ('aaa'..'ddd'); 2-D = bw*2 + bh (pywt dwt2 order LL,'da','ad','dd'); 1-D = bw
    const int64_t bsH = (d->nd == 2) ? d->cs_band : 2 * d->cs_band;
    const int64_t bsW = (d->nd == 2) ? 2 * d->cs_band : d->cs_band;
    int64_t sc[5] = {d->cs_img, 4 * d->cs_band, d->cs0, bsH, d->cs1};
    if (!hasT) { sc[1] = 0; sc[2] = 0; }
    if (!hasH) { sc[3] = 0; sc[4] = 0; }
    set_outer(pW.om, 5, n, sa, sc, pW.outer);
    pW.inner = 1; pW.sig_stride = 1; pW.coef_stride = 1; pW.band_stride = bsW;
  }

  if (analysis_dir) {
    const float* cur = src;
    if (hasT) { launch(true, cur, t1, pT); cur = t1; }
    if (hasH) { launch(true, cur, t2, pH); cur = t2; }
    launch(true, cur, dst, pW);
  } else {
    // coefficients -> signal: W, then H, then T
    float* wout = hasH ? t2 : dst;
    launch(false, src, wout, pW);
    if (hasH) {
      float* hout = hasT ? t1 : dst;
      launch(false, t2, hout, pH);
      if (hasT) launch(false, t1, dst, pT);
    }
  }
  return wdno_check_launch();
}

extern "C" int wdno_dwt_fwd(const float* x, float* coef, const wdno_dwt_desc* d, const float* f, void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = validate(d);
  if (rc) return rc;
  for (int a = 3 - d->nd; a < 3; ++a)
    if (d->mode == 1 && d->out_dims[a] != (d->in_dims[a] + d->L - 1) / 2) return WDNO_EINVAL;
  return run(x, coef, d, f, f + d->L, false, true, true, ws, ws_bytes, as_stream(s));
}
extern "C" int wdno_dwt_inv(const float* coef, float* x, const wdno_dwt_desc* d, const float* f, void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = validate(d);
  if (rc) return rc;
  for (int a = 3 - d->nd; a < 3; ++a) {
    int want = d->mode == 1 ? 2 * d->out_dims[a] - d->L + 2 : 2 * d->out_dims[a];
    if (d->in_dims[a] != want) return WDNO_EINVAL;   // the inverse always produces the natural (even) length
  }
  return run(coef, x, d, f + 2 * d->L, f + 3 * d->L, false, false, false, ws, ws_bytes, as_stream(s));
}
// d(coef) from d(x) for x = inv(coef): analysis-shaped with reversed reconstruction taps
extern "C" int wdno_dwt_inv_adjoint(const float* dx, float* dcoef, const wdno_dwt_desc* d, const float* f, void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = validate(d);
  if (rc) return rc;
  for (int a = 3 - d->nd; a < 3; ++a) {
    int want = d->mode == 1 ? 2 * d->out_dims[a] - d->L + 2 : 2 * d->out_dims[a];
    if (d->in_dims[a] != want) return WDNO_EINVAL;
  }
  return run(dx, dcoef, d, f + 2 * d->L, f + 3 * d->L, true, true, false, ws, ws_bytes, as_stream(s));
}
// d(x) from d(coef) for coef = fwd(x): synthesis-shaped with reversed decomposition taps
extern "C" int wdno_dwt_fwd_adjoint(const float* dcoef, float* dx, const wdno_dwt_desc* d, const float* f, void* ws, size_t ws_bytes, wdno_stream_t s) {
  int rc = validate(d);
  if (rc) return rc;
  for (int a = 3 - d->nd; a < 3; ++a)
    if (d->mode == 1 && d->out_dims[a] != (d->in_dims[a] + d->L - 1) / 2) return WDNO_EINVAL;
  return run(dcoef, dx, d, f, f + d->L, true, false, true, ws, ws_bytes, as_stream(s));
}
