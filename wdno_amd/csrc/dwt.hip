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
#include <type_traits>
#include <algorithm>
#include <cstdlib>

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

// ------------------------------------------------------------------------------------------------ fused single-launch transforms
// One launch per 2-D / 3-D transform, no workspace round trip ("fuse all axes in LDS").
// A block owns a tile of coefficients (analysis) / signal samples (synthesis) of one image and runs the per-axis passes back to
// back: the pass along the SLOWEST transformed axis lives in registers (it needs no neighbours across lanes: a thread owns one
// column, its loads / stores are coalesced over the contiguous axis and touch global memory), the other passes go through LDS.
//     analysis : global --T (regs)--> S1 --H--> S2 --W--> packed coefficients          (2-D: global --H (regs)--> S2 --W--> coef)
//     synthesis: packed coefficients --W--> S1 --H--> S2 --T (regs)--> global          (2-D: coef --W--> S1 --H (regs)--> global)
// Boundary rules (zero extension / periodization / repeated last sample of an odd length) are applied where a pass reads its
// operand from global memory, so the LDS passes are unconditional. Pass order and the order of every fmaf chain are those of the
// per-axis kernels above, hence the results are BIT-IDENTICAL to them (tests compare with torch.equal).
// Halo rows/frames of neighbouring tiles are re-read through L2: logical tile ids are laid out so that one XCD (= one L2) owns a
// contiguous range of tiles, i.e. whole images.
extern int wdno_debug_mode;      // 11: force the per-axis passes (A/B and bit-equality tests)

// n / d for 0 <= n, n * d < 2^32 as one multiply-high (m = floor(2^32 / d) + 1): the item loops below decode (row, column)
// from a linear index several times per item, and a hardware integer division is ~30 VALU instructions on gfx950 -- with them
// the 3-D synthesis kernel was ALU-bound at 50 us for 38 MB.
struct FastDiv { unsigned d, m; };
static inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = (unsigned)d;
  f.m = d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1ull) : 0u;
  return f;
}
__device__ __forceinline__ int fd_div(int n, FastDiv f) { return f.d == 1 ? n : (int)__umulhi((unsigned)n, f.m); }

struct FusedGeom {
  FastDiv dFW, dNH, dWo, dQW, dEH, dNQH2, dRW, dQW4;      // (dRW, dQW4: the staged W pass of the 2-D synthesis)
  int n_img, T, H, W, To, Ho, Wo;
  int64_t cs_img, cs_band, cs0, cs1;
  int off, odd_t, odd_h, odd_w;
  int NH;                 // analysis: coefficient rows per tile; synthesis: q rows per tile (q = (n + off) >> 1)
  int tiles_t, tiles_h, n_blocks;
  int FW;                 // analysis: columns incl. boundary extension (2 Wo + L - 2); synthesis: 2 * QW
  int qt0, qh0, qw0;      // synthesis: first q along each axis (off >> 1)
  int QH, QT;             // synthesis: number of q rows / frames in total
  int debug;              // tools only: 12 / 13 / 14 skip the W / H / T pass of the synthesis kernel (timing breakdown, wrong results)
  // PACKED analysis (wdno_dwt_fwd_packed): image = (outer, inner) = (img / img_inner, img % img_inner), coefficient base = outer * cs_outer +
  // inner * cs_img, every coefficient divided (IEEE) by resc[inner * 2^ND + band] on its way out
  // Rows are written in full: row_w (a multiple of 4, >= Wo) columns, zeros (/ resc) beyond the coefficients, as aligned 16-byte stores.
  FastDiv dInner;
  int img_inner, row_w;
  int64_t cs_outer;
  const float* resc;
};

__device__ __forceinline__ int xcd_tile(int bid, int nb) {
  return (nb & 7) ? bid : (bid & 7) * (nb >> 3) + (bid >> 3);
}
// Periodic wrap without an integer division: every index a valid item asks for lies within one period of [0, P) (|j| < 2 P is
// guaranteed by L <= P, checked on the host); items of a ragged last tile may ask for more and are clamped (their results are
// never stored), so that no read leaves the tensor.
__device__ __forceinline__ int wrap_period(int j, int P) {
  j = j < 0 ? j + P : j;
  j = j >= P ? j - P : j;
  j = j >= P ? j - P : j;
  return min(max(j, 0), P - 1);
}
template <int MODE>
__device__ __forceinline__ int amap(int j, int N, int odd) {      // signal index read by an analysis pass: [0, N) or -1 (= zero)
  if (MODE == 1) return (j >= 0 && j < N) ? j : -1;
  j = wrap_period(j, N + odd);
  return j > N - 1 ? N - 1 : j;
}
template <int MODE>
__device__ __forceinline__ int smap(int K, int M) {               // coefficient index read by a synthesis pass
  if (MODE == 1) return (K >= 0 && K < M) ? K : -1;
  return wrap_period(K, M);
}

// ND = 3: register pass along T, NK coefficient frames per block.  ND = 2: register pass along H, NK coefficient rows per item.
template <int ND, int L, int MODE, int NK, bool PACKED = false>
__global__ __launch_bounds__(256) void dwt_analysis_fused_kernel(const float* __restrict__ x, float* __restrict__ coef, FusedGeom g, Taps t) {
  extern __shared__ float lds[];
  int b = xcd_tile(blockIdx.x, g.n_blocks);
  const int th = b % g.tiles_h;
  b /= g.tiles_h;
  const int tt = (ND == 3) ? b % g.tiles_t : 0;
  const int img = (ND == 3) ? b / g.tiles_t : b;
  const int FW = g.FW, NH = g.NH, off = g.off;
  const int kh0 = th * NH;
  const int nh = min(NH, g.Ho - kh0);
  const float* __restrict__ xi = x + (int64_t)img * g.T * g.H * g.W;
  constexpr int NP = (ND == 3) ? 2 * NK : 1;
  float* S2;                                     // [NP][2][NH][FW]
  int kt0 = 0;
  if (ND == 3) {
    const int FH = 2 * NH + L - 2;               // signal rows under the tile: row r <-> h = 2 kh0 - off + r
    const int rows = 2 * nh + L - 2;
    float* S1 = lds;                             // [2 NK][FH][FW]
    S2 = lds + 2 * NK * FH * FW;
    kt0 = tt * NK;
    const int64_t fs = (int64_t)g.H * g.W;
    // (measured and dropped: batches of 2 / 3 items with all their frame loads requested first and 32-bit offsets: 25.4-25.9 us against 17.8)
    for (int it = threadIdx.x; it < rows * FW; it += 256) {
      const int r = fd_div(it, g.dFW), c = it - r * FW;
      const int h = amap<MODE>(2 * kh0 - off + r, g.H, g.odd_h);
      const int w = amap<MODE>(c - off, g.W, g.odd_w);
      float lo[NK], hi[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) { lo[k] = 0.f; hi[k] = 0.f; }
      if (h >= 0 && w >= 0) {
        const float* col = xi + (int64_t)h * g.W + w;
#pragma unroll
        for (int jj = 0; jj < 2 * NK + L - 2; ++jj) {
          const int tj = amap<MODE>(2 * kt0 - off + jj, g.T, g.odd_t);
          const float v = tj >= 0 ? col[tj * fs] : 0.f;
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            const int m = jj - 2 * k;
            if (m >= 0 && m < L) {
              lo[k] = fmaf(t.lo[L - 1 - m], v, lo[k]);
              hi[k] = fmaf(t.hi[L - 1 - m], v, hi[k]);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        S1[(k * FH + r) * FW + c] = lo[k];
        S1[((NK + k) * FH + r) * FW + c] = hi[k];
      }
    }
    __syncthreads();
    // pass H: an item is a COLUMN (plane p, column c) of S1: its nh outputs slide a window of L values down the column (two new reads per
    // output instead of L, one index decode per column instead of one per output); every sum in the order m = 0 .. L - 1 as before
    for (int it = threadIdx.x; it < NP * FW; it += 256) {
      const int p = fd_div(it, g.dFW), c = it - p * FW;
      const float* col = S1 + p * FH * FW + c;
      float* o0 = S2 + (p * 2 * NH) * FW + c;
      float w[L];
#pragma unroll
      for (int m = 0; m < L - 2; ++m) w[m + 2] = col[m * FW];
      col += (L - 2) * FW;
      for (int kh = 0; kh < nh; ++kh) {
#pragma unroll
        for (int m = 0; m < L - 2; ++m) w[m] = w[m + 2];
        w[L - 2] = col[0]; w[L - 1] = col[FW];
        col += 2 * FW;
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int m = 0; m < L; ++m) {
          lo = fmaf(t.lo[L - 1 - m], w[m], lo);
          hi = fmaf(t.hi[L - 1 - m], w[m], hi);
        }
        o0[kh * FW] = lo;
        o0[(NH + kh) * FW] = hi;
      }
    }
  } else {
    // 2-D: register pass along H straight from global memory. With periodization (MODE 0) a row of S2 holds ONE period (FW = 2 Wo columns,
    // set by the host): column c <-> w = (c - off) mod period, and the W pass below wraps its reads -- the L - 2 boundary columns of the
    // extended row were 272 items for 256 threads at W = 128 (a second round of global loads for 16 threads). The source rows of an item
    // are consecutive modulo the period: one wrap, then increments (16 boundary maps and 64-bit address products per item before).
    S2 = lds;
    const int ngroups = (nh + NK - 1) / NK;
    const int PH = g.H + g.odd_h;
    for (int it = threadIdx.x; it < ngroups * FW; it += 256) {
      const int gq = fd_div(it, g.dFW);
      const int c = it - gq * FW, k0 = gq * NK;
      const int w = amap<MODE>(c - off, g.W, g.odd_w);
      float lo[NK], hi[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) { lo[k] = 0.f; hi[k] = 0.f; }
      if (w >= 0) {
        const int hb = 2 * (kh0 + k0) - off;
        int hv = MODE == 0 ? wrap_period(hb, PH) : hb;              // MODE 0: position within the period; MODE 1: the row index itself
        float v[2 * NK + L - 2];
#pragma unroll
        for (int jj = 0; jj < 2 * NK + L - 2; ++jj) {
          int hj;
          if (MODE == 0) { hj = min(hv, g.H - 1); hv = hv + 1 == PH ? 0 : hv + 1; }
          else { hj = (hv >= 0 && hv < g.H) ? hv : -1; ++hv; }
          v[jj] = hj >= 0 ? xi[hj * g.W + w] : 0.f;                  // (host: an image has < 2^31 elements)
        }
#pragma unroll
        for (int jj = 0; jj < 2 * NK + L - 2; ++jj) {
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            const int m = jj - 2 * k;
            if (m >= 0 && m < L) {
              lo[k] = fmaf(t.lo[L - 1 - m], v[jj], lo[k]);
              hi[k] = fmaf(t.hi[L - 1 - m], v[jj], hi[k]);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (k0 + k < nh) {
          S2[(k0 + k) * FW + c] = lo[k];
          S2[(NH + k0 + k) * FW + c] = hi[k];
        }
    }
  }
  __syncthreads();
  // pass W: LDS -> packed coefficients. float2 reads (index 2 kw + m): conflict-free, and m ascends inside each pair
  float* __restrict__ ci = coef + (int64_t)img * g.cs_img;
  const float* __restrict__ rs = nullptr;
  if (PACKED) {
    const int io = fd_div(img, g.dInner), ii = img - io * g.img_inner;
    ci = coef + (int64_t)io * g.cs_outer + (int64_t)ii * g.cs_img;
    rs = g.resc + ii * (1 << ND);
  }
  const int Wo = g.Wo;
  constexpr bool PERIOD_ROWS = ND == 2 && MODE == 0;          // S2 rows hold one period: float2 index kw + i wraps at Wo
  // an item = FOUR consecutive coefficients of one row: a window of L / 2 + 3 float2 reads (instead of 4 L / 2) and one row decode
  constexpr int HLW = L / 2;
  const int Wo4 = PACKED ? g.row_w >> 2 : (Wo + 3) >> 2;
  const int rmax = PERIOD_ROWS ? Wo - 1 : (FW >> 1) - 1;       // last float2 of a row
  for (int it = threadIdx.x; it < NP * 2 * NH * Wo4; it += 256) {
    int q = fd_div(it, g.dQW4);
    const int kw0 = (it - q * Wo4) * 4;
    const int q2 = fd_div(q, g.dNH), kh = q - q2 * NH;
    if (kh >= nh) continue;
    const int bh = q2 & 1, p = q2 >> 1;
    const int bt = (ND == 3) ? p / NK : 0;
    const int kt = (ND == 3) ? kt0 + p % NK : 0;
    if (ND == 3 && kt >= g.To) continue;
    const float2* row0 = reinterpret_cast<const float2*>(S2 + ((p * 2 + bh) * NH + kh) * FW);
    float2 v[HLW + 3];
#pragma unroll
    for (int i = 0; i < HLW + 3; ++i) {
      int idx = kw0 + i;
      if (PERIOD_ROWS) idx = idx >= Wo ? idx - Wo : idx;
      v[i] = row0[min(idx, rmax)];                             // (past the row only for the coefficients kw >= Wo of a ragged last group: not stored)
    }
    const int band_lo = (ND == 3) ? bt * 4 + bh * 2 : bh;
    const int band_hi = (ND == 3) ? band_lo + 1 : bh + 2;
    float* o = ci + (int64_t)kt * g.cs0 + (int64_t)(kh0 + kh) * g.cs1 + kw0;
    float r_lo = 1.f, r_hi = 1.f;
    if (PACKED) { r_lo = rs[band_lo]; r_hi = rs[band_hi]; }
    float pl[4], ph[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float lo = 0.f, hi = 0.f;
#pragma unroll
      for (int i = 0; i < HLW; ++i) {
        lo = fmaf(t.lo[L - 1 - 2 * i], v[u + i].x, lo);
        hi = fmaf(t.hi[L - 1 - 2 * i], v[u + i].x, hi);
        lo = fmaf(t.lo[L - 2 - 2 * i], v[u + i].y, lo);
        hi = fmaf(t.hi[L - 2 - 2 * i], v[u + i].y, hi);
      }
      if (PACKED) {
        const bool in = kw0 + u < Wo;
        pl[u] = (in ? lo : 0.f) / r_lo; ph[u] = (in ? hi : 0.f) / r_hi;
        continue;
      }
      if (kw0 + u < Wo) {
        o[band_lo * g.cs_band + u] = lo;
        o[band_hi * g.cs_band + u] = hi;
      }
    }
    if (PACKED) {
      *reinterpret_cast<float4*>(o + band_lo * g.cs_band) = make_float4(pl[0], pl[1], pl[2], pl[3]);
      *reinterpret_cast<float4*>(o + band_hi * g.cs_band) = make_float4(ph[0], ph[1], ph[2], ph[3]);
    }
  }
}

// q = (n + off) >> 1, r = (n + off) & 1: x[n] = sum_i lo[q - i] g_lo[r + 2 i] + hi[q - i] g_hi[r + 2 i], i < L/2.
// ND = 3: register pass along T with NQ q-frames (2 NQ signal frames) per block. ND = 2: register pass along H, NQ q-rows per item.
template <int ND, int L, int MODE, int NQ>
__global__ __launch_bounds__(256) void dwt_synthesis_fused_kernel(const float* __restrict__ coef, float* __restrict__ x, FusedGeom g, Taps t) {
  extern __shared__ float lds[];
  constexpr int E = L / 2 - 1;
  constexpr int ET = NQ + E;
  int b = xcd_tile(blockIdx.x, g.n_blocks);
  const int th = b % g.tiles_h;
  b /= g.tiles_h;
  const int tt = (ND == 3) ? b % g.tiles_t : 0;
  const int img = (ND == 3) ? b / g.tiles_t : b;
  const int NW = g.FW, QW = NW >> 1, NQH = g.NH, off = g.off;
  const int EH = NQH + E;
  const int qh_start = g.qh0 + th * NQH;
  const int nqh = min(NQH, g.qh0 + g.QH - qh_start);       // valid q rows of this tile
  const int qt_start = (ND == 3) ? g.qt0 + tt * NQ : 0;
  const float* __restrict__ ci = coef + (int64_t)img * g.cs_img;
  float* __restrict__ xi = x + (int64_t)img * g.T * g.H * g.W;
  constexpr int NP = (ND == 3) ? 2 * ET : 1;
  float* S1 = lds;                                           // [NP][2][EH][NW]
  // pass W: packed coefficients -> S1
  const int eh_used = nqh + E;
  // Branch-free body: every load is issued unconditionally from a clamped (always valid) address and masked by a select, so that
  // the loads of several items are in flight together (with early-outs the loop was a chain of dependent L2 round trips: 22 of the
  // kernel's 42 us at [32, 8, 18, 34, 34]). A masked term contributes fmaf(0, tap, a) = a, i.e. the skipped fmaf of the per-axis kernel.
  // WB items per round: their WB * L loads are all requested before the first is used (a thread has 6 items at the Burgers shape; with the
  // compiler's 2-item unroll that was three dependent round trips to L2 / HBM per block).
  constexpr int WB = (ND == 2) ? 3 : 2;
  constexpr int HL = L / 2;
  const int n_w = NP * 2 * EH * QW;
  if (g.debug != 12)
  for (int it0 = threadIdx.x; it0 < n_w; it0 += 256 * WB) {
    float cl[WB][HL], ch[WB][HL];
    unsigned okm[WB];
    int dst[WB];
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const bool live = it0 + 256 * u < n_w;
      const int it = live ? it0 + 256 * u : 0;                  // (a slot past the end decodes item 0: every address stays inside the image)
      int q = fd_div(it, g.dQW);
      const int ql = it - q * QW;
      const int q2 = fd_div(q, g.dEH), eh = q - q2 * EH;
      const int bh = q2 & 1, p = q2 >> 1;
      const int bt = (ND == 3) ? p / ET : 0;
      const int Kt = (ND == 3) ? smap<MODE>(qt_start - E + p % ET, g.To) : 0;
      const int Kh = smap<MODE>(qh_start - E + eh, g.Ho);
      const bool rowok = live && Kt >= 0 && Kh >= 0;
      const int band_lo = (ND == 3) ? bt * 4 + bh * 2 : bh;
      const int band_hi = (ND == 3) ? band_lo + 1 : bh + 2;
      const int base = rowok ? (int)((int64_t)Kt * g.cs0 + (int64_t)Kh * g.cs1) : 0;        // (host: cs_img < 2^31)
      const float* rl = ci + band_lo * g.cs_band;
      const float* rh = ci + band_hi * g.cs_band;
      okm[u] = 0;
      dst[u] = (live && eh < eh_used) ? ((p * 2 + bh) * EH + eh) * NW + 2 * ql : -1;
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const int Kw = smap<MODE>(g.qw0 + ql - i, g.Wo);
        const int o = base + (Kw >= 0 ? Kw : 0);
        cl[u][i] = rl[o]; ch[u][i] = rh[o];
        okm[u] |= (rowok && Kw >= 0) ? (1u << i) : 0u;
      }
    }
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const bool ok = (okm[u] >> i) & 1u;
        const float vl = ok ? cl[u][i] : 0.f, vh = ok ? ch[u][i] : 0.f;
        a0 = fmaf(vl, t.lo[2 * i], a0);
        a0 = fmaf(vh, t.hi[2 * i], a0);
        a1 = fmaf(vl, t.lo[2 * i + 1], a1);
        a1 = fmaf(vh, t.hi[2 * i + 1], a1);
      }
      if (dst[u] >= 0) *reinterpret_cast<float2*>(S1 + dst[u]) = make_float2(a0, a1);
    }
  }
  __syncthreads();
  const int wshift = off & 1;                                // position pw <-> n_w = pw - (off & 1)
  if (ND == 3) {
    float* S2 = lds + NP * 2 * EH * NW;                      // [2 ET][2 NQH][NW]
    if (g.debug != 13)
#pragma unroll 2
    for (int it = threadIdx.x; it < NP * 2 * NQH * NW; it += 256) {
      int q = fd_div(it, g.dFW);
      const int pw = it - q * NW;
      const int p = fd_div(q, g.dNQH2), ph = q - p * 2 * NQH;
      if (ph >= 2 * nqh) continue;
      const int ql = ph >> 1, r = ph & 1;
      const float* cl = S1 + ((p * 2 + 0) * EH + ql + E) * NW + pw;
      const float* ch = S1 + ((p * 2 + 1) * EH + ql + E) * NW + pw;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < L / 2; ++i) {
        acc = fmaf(cl[-i * NW], r ? t.lo[2 * i + 1] : t.lo[2 * i], acc);
        acc = fmaf(ch[-i * NW], r ? t.hi[2 * i + 1] : t.hi[2 * i], acc);
      }
      S2[(p * 2 * NQH + ph) * NW + pw] = acc;
    }
    __syncthreads();
    const int64_t fs = (int64_t)g.H * g.W;
    if (g.debug != 14)
    for (int it = threadIdx.x; it < 2 * nqh * NW; it += 256) {
      const int ph = fd_div(it, g.dFW), pw = it - ph * NW;
      const int nw = pw - wshift;
      const int nhh = 2 * (qh_start + (ph >> 1)) + (ph & 1) - off;
      if (nw < 0 || nw >= g.W || nhh < 0 || nhh >= g.H) continue;
      float cl[ET], ch[ET];
#pragma unroll
      for (int e = 0; e < ET; ++e) {
        cl[e] = S2[(e * 2 * NQH + ph) * NW + pw];
        ch[e] = S2[((ET + e) * 2 * NQH + ph) * NW + pw];
      }
      float* o = xi + (int64_t)nhh * g.W + nw;
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < L / 2; ++i) {
            acc = fmaf(cl[qq - i + E], t.lo[r + 2 * i], acc);
            acc = fmaf(ch[qq - i + E], t.hi[r + 2 * i], acc);
          }
          const int nt = 2 * (qt_start + qq) + r - off;
          if (nt >= 0 && nt < g.T) o[nt * fs] = acc;
        }
      }
    }
  } else {
    // register pass along H: items (group of NQ q-rows, pw); EH rows of S1 hold the tile's coefficient rows (after the W pass)
    const int ngroups = (nqh + NQ - 1) / NQ;
    for (int it = threadIdx.x; it < ngroups * NW; it += 256) {
      const int gq = fd_div(it, g.dFW);
      const int pw = it - gq * NW, q0 = gq * NQ;
      const int nw = pw - wshift;
      if (nw < 0 || nw >= g.W) continue;
      float cl[ET], ch[ET];
#pragma unroll
      for (int e = 0; e < ET; ++e) {
        const bool ok = q0 + e < eh_used;
        cl[e] = ok ? S1[(q0 + e) * NW + pw] : 0.f;
        ch[e] = ok ? S1[(EH + q0 + e) * NW + pw] : 0.f;
      }
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < L / 2; ++i) {
            acc = fmaf(cl[qq - i + E], t.lo[r + 2 * i], acc);
            acc = fmaf(ch[qq - i + E], t.hi[r + 2 * i], acc);
          }
          const int nhh = 2 * (qh_start + q0 + qq) + r - off;
          if (q0 + qq < nqh && nhh >= 0 && nhh < g.H) xi[(int64_t)nhh * g.W + nw] = acc;
        }
      }
    }
  }
}


// Round 3: 2-D synthesis with the W pass fed from LDS. Ablation of the kernel above at the Burgers shape ([128, 4, 80, 64] -> [128, 160, 128],
// bior2.4, periodization; 14.8 us for 21 MB; tools/bench_dwt.py): without its loads 8.1 us, without the H pass 9.3, without the W pass 5.6.
// All 1280 blocks are resident at once (5 per CU) and the kernel lasts as long as ONE block's chain load -> W pass -> H pass -> store:
// what shortens it is fewer dependent round trips, not fewer instructions (a variant whose blocks walk several tiles with the next tile's
// rows prefetched into registers was slower for every tile count: 15.3 / 17.2 / 19.8 / 24.5 us at 1 / 2 / 3 / 5 tiles per block).
//   * staging: the tile's raw coefficient rows go to LDS -- row (band pair bh, coefficient row eh, lo / hi) holds the RW = 4 ceil(QW / 4)
//     + L / 2 - 1 coefficients K = qw0 - (L / 2 - 1) + p with the boundary rule applied (wrapped or zero). A wave takes whole rows (row
//     index and map are wave-uniform), a lane the same columns of every row; all loads of a thread are in flight together.
//   * W pass: an item is FOUR consecutive output pairs of one row: a window of L / 2 + 3 values per band from LDS instead of 4 * L / 2
//     global loads with a boundary map each.
//   * H pass: as above (register pass over NQ q-rows per item).
// Every sum keeps the order i = 0, 1, .. of the per-axis kernel: bit-identical (tests/test_gpu_dwt_fused.py).
template <int L, int MODE, int NQ, int PMAX>
__global__ __launch_bounds__(256) void dwt_synthesis2_kernel(const float* __restrict__ coef, float* __restrict__ x, FusedGeom g, Taps t) {
  extern __shared__ float lds[];
  constexpr int E = L / 2 - 1, ET = NQ + E, HL = L / 2, RMAX = 2 * NQ + E;       // rows per wave: 4 EH / 4, EH <= 2 NQ + E (host)
  const int NW = g.FW, QW = NW >> 1, NQH = g.NH, off = g.off;
  const int EH = NQH + E;
  const int QW4 = (QW + 3) >> 2, RW = 4 * QW4 + HL - 1;
  float* S1 = lds;                                            // [2 bh][EH][NW]
  float* R = lds + 2 * EH * NW;                               // [2 bh][EH][lo, hi][RW]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wshift = off & 1;
  // columns of this lane in a staged row (two per pass of 128; RW <= 128 PMAX by the host's choice): the same for every tile
  int kw[PMAX][2];
  bool colok[PMAX][2];
#pragma unroll
  for (int pp = 0; pp < PMAX; ++pp)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pidx = 128 * pp + 64 * h + lane;
      const int Kw = smap<MODE>(g.qw0 - (HL - 1) + min(pidx, RW - 1), g.Wo);
      colok[pp][h] = Kw >= 0 && pidx < RW;
      kw[pp][h] = Kw >= 0 ? Kw : 0;
    }
  constexpr int npp = PMAX;
  float v[PMAX][2][RMAX];
  auto tile_of = [&](int tile, int& img, int& qh_start) {
    const int b = xcd_tile(tile, g.n_blocks);
    const int th = b % g.tiles_h;
    img = b / g.tiles_h;
    qh_start = g.qh0 + th * NQH;
  };
  auto fetch = [&](int tile) {
    int img, qh_start;
    tile_of(tile, img, qh_start);
    const float* __restrict__ ci = coef + (int64_t)img * g.cs_img;
#pragma unroll
    for (int pp = 0; pp < PMAX; ++pp) {
      if (pp >= npp) break;
#pragma unroll
      for (int rr = 0; rr < RMAX; ++rr) {
        const int r = wave + 4 * rr;                          // r = (bh * EH + eh) * 2 + band
        if (r >= 4 * EH) break;
        const int band = r & 1, q2 = r >> 1;
        const int bh = q2 >= EH ? 1 : 0, eh = q2 - bh * EH;
        const int Kh = smap<MODE>(qh_start - E + eh, g.Ho);
        const float* row = ci + (bh + 2 * band) * g.cs_band + (int64_t)(Kh >= 0 ? Kh : 0) * g.cs1;
        v[pp][0][rr] = row[kw[pp][0]];
        v[pp][1][rr] = row[kw[pp][1]];
      }
    }
  };
  const int tile = blockIdx.x;
  fetch(tile);
  {
    int img, qh_start;
    tile_of(tile, img, qh_start);
    const int nqh = min(NQH, g.qh0 + g.QH - qh_start);
    const int eh_used = nqh + E;
    float* __restrict__ xi = x + (int64_t)img * g.T * g.H * g.W;
    // registers -> R
#pragma unroll
    for (int pp = 0; pp < PMAX; ++pp) {
      if (pp >= npp) break;
#pragma unroll
      for (int rr = 0; rr < RMAX; ++rr) {
        const int r = wave + 4 * rr;
        if (r >= 4 * EH) break;
        const int q2 = r >> 1;
        const int eh = q2 >= EH ? q2 - EH : q2;
        const bool rowok = smap<MODE>(qh_start - E + eh, g.Ho) >= 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pidx = 128 * pp + 64 * h + lane;
          if (pidx < RW) R[r * RW + pidx] = (colok[pp][h] && rowok) ? v[pp][h][rr] : 0.f;
        }
      }
    }
    __syncthreads();
    // pass W: R -> S1
    for (int it = threadIdx.x; it < 2 * EH * QW4; it += 256) {
      const int q2 = fd_div(it, g.dQW4), j = it - q2 * QW4;    // q2 = bh * EH + eh
      const int eh = q2 >= EH ? q2 - EH : q2;
      if (eh >= eh_used) continue;
      const float* wl = R + (q2 * 2) * RW + 4 * j;
      const float* wh = wl + RW;
      float cl[HL + 3], ch[HL + 3];
#pragma unroll
      for (int e = 0; e < HL + 3; ++e) { cl[e] = wl[e]; ch[e] = wh[e]; }
      float o8[8];
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          const float vl = cl[tq + HL - 1 - i], vh = ch[tq + HL - 1 - i];
          a0 = fmaf(vl, t.lo[2 * i], a0);
          a0 = fmaf(vh, t.hi[2 * i], a0);
          a1 = fmaf(vl, t.lo[2 * i + 1], a1);
          a1 = fmaf(vh, t.hi[2 * i + 1], a1);
        }
        o8[2 * tq] = a0; o8[2 * tq + 1] = a1;
      }
      float* d = S1 + q2 * NW + 8 * j;
      if (4 * j + 3 < QW && (NW & 3) == 0) {
        *reinterpret_cast<float4*>(d) = make_float4(o8[0], o8[1], o8[2], o8[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
      } else {
#pragma unroll
        for (int tq = 0; tq < 4; ++tq)
          if (4 * j + tq < QW) { d[2 * tq] = o8[2 * tq]; d[2 * tq + 1] = o8[2 * tq + 1]; }
      }
    }
    __syncthreads();
    // pass H (registers): items (group of NQ q-rows, pw)
    const int ngroups = (nqh + NQ - 1) / NQ;
    for (int it = threadIdx.x; it < ngroups * NW; it += 256) {
      const int gq = fd_div(it, g.dFW);
      const int pw = it - gq * NW, q0 = gq * NQ;
      const int nw = pw - wshift;
      if (nw < 0 || nw >= g.W) continue;
      float cl[ET], ch[ET];
#pragma unroll
      for (int e = 0; e < ET; ++e) {
        const bool ok = q0 + e < eh_used;
        cl[e] = ok ? S1[(q0 + e) * NW + pw] : 0.f;
        ch[e] = ok ? S1[(EH + q0 + e) * NW + pw] : 0.f;
      }
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < L / 2; ++i) {
            acc = fmaf(cl[qq - i + E], t.lo[r + 2 * i], acc);
            acc = fmaf(ch[qq - i + E], t.hi[r + 2 * i], acc);
          }
          const int nhh = 2 * (qh_start + q0 + qq) + r - off;
          if (q0 + qq < nqh && nhh >= 0 && nhh < g.H) xi[nhh * g.W + nw] = acc;
        }
      }
    }
  }
}

// Round 3: 3-D synthesis that STREAMS over the coefficient frames of its tile. The kernel above keeps the W-pass and H-pass results of
// all ET = NQ + E coefficient frames of a tile in LDS (12 planes for bior1.3), which leaves room for 2 q-rows per tile at 40 KB: 4896 blocks
// for [32, 8, 18, 34, 34], each coefficient read 3x through L2 and every W-pass item computed 3x (halo 6/4 in T times 4/2 in H). Here only ONE
// frame is in LDS at a time (W pass -> S1, H pass -> S2, 20 KB for 8 q-rows); the T pass is an accumulation in registers: a thread owns
// ITEMS output columns (ph, pw) and 2 NQ time samples of each, and every staged frame adds its lo / hi contribution to the <= L/2 q-frames it
// belongs to. Frames are walked in DESCENDING order so that each accumulator receives its terms in the order i = 0, 1, .. of the
// per-axis kernel (K = q - i): results stay BIT-IDENTICAL to it (tests/test_gpu_dwt_fused.py). wdno_debug 45: the kernel above (A/B).
// The kernel is bound by VALU issue, not by memory (4 blocks per CU run concurrently; ~2.5 K wave instructions per wave at 4 cycles each were 19 of
// its 25 us): so (a) everything that does not depend on the frame -- the (band, row, column) decode of a thread's W- and H-pass items, the
// boundary maps, the LDS offsets -- is computed once before the frame loop, (b) an H-pass item produces BOTH rows 2 ql, 2 ql + 1 from one set
// of L reads, (c) the global loads of frame e - 1 are issued before the H / T passes of frame e. 30.0 -> 25.3 (c) -> 16.1 us (a, b) at
// [32,8,18,34,34]; 70.5 -> 56.4 -> 32.5 us at [10,8,34,66,66].
template <int L, int MODE, int NQ, int ITEMS, int WMAX>
__global__ __launch_bounds__(256) void dwt_synthesis3_stream_kernel(const float* __restrict__ coef, float* __restrict__ x, FusedGeom g, Taps t) {
  extern __shared__ float lds[];
  constexpr int E = L / 2 - 1;
  constexpr int ET = NQ + E;
  constexpr int HL = L / 2;
  int b = xcd_tile(blockIdx.x, g.n_blocks);
  const int th = b % g.tiles_h;
  b /= g.tiles_h;
  const int tt = b % g.tiles_t;
  const int img = b / g.tiles_t;
  const int NW = g.FW, QW = NW >> 1, NQH = g.NH, off = g.off;
  const int EH = NQH + E;
  const int qh_start = g.qh0 + th * NQH;
  const int nqh = min(NQH, g.qh0 + g.QH - qh_start);
  const int qt_start = g.qt0 + tt * NQ;
  const int eh_used = nqh + E;
  const float* __restrict__ ci = coef + (int64_t)img * g.cs_img;
  float* __restrict__ xi = x + (int64_t)img * g.T * g.H * g.W;
  float* S1 = lds;                              // [2 bt][2 bh][EH][NW]
  float* S2 = lds + 4 * EH * NW;                // [2 bt][2 NQH][NW]
  const int n_items = 2 * nqh * NW;             // output columns (ph, pw) of this tile
  // W-pass items of this thread: the same (band pair, coefficient row, column pair) in every frame, so the index work is done ONCE and a
  // frame is only its offset Kt * cs0 (the decode per frame was half of the pass's instructions). The loads of frame e - 1 are issued
  // before the H / T passes of frame e and wait in registers: a block's chain of six exposed global-load latencies (one per frame; the
  // blocks of a CU all run concurrently, so the kernel lasts as long as ONE block's chain) becomes one.
  int w_dst[WMAX], w_off[WMAX][HL];
  unsigned w_ok[WMAX];
  const int n_w = 4 * EH * QW;
#pragma unroll
  for (int k = 0; k < WMAX; ++k) {
    const int it = threadIdx.x + 256 * k;
    int q = fd_div(it, g.dQW);
    const int ql = it - q * QW;
    const int q2 = fd_div(q, g.dEH), eh = q - q2 * EH;
    const int bh = q2 & 1, bt = q2 >> 1;
    const int Kh = smap<MODE>(qh_start - E + eh, g.Ho);
    const bool rowok = Kh >= 0 && it < n_w;
    const int src = rowok ? (int)((bt * 4 + bh * 2) * g.cs_band + (int64_t)Kh * g.cs1) : 0;      // (host: cs_img < 2^31)
    w_dst[k] = (it < n_w && eh < eh_used) ? (q2 * EH + eh) * NW + 2 * ql : -1;
    w_ok[k] = 0;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int Kw = smap<MODE>(g.qw0 + ql - i, g.Wo);
      w_off[k][i] = src + (rowok && Kw >= 0 ? Kw : 0);
      w_ok[k] |= (rowok && Kw >= 0) ? (1u << i) : 0u;
    }
  }
  float pl[WMAX][HL], ph_[WMAX][HL];
  auto fetch = [&](int e) {
    const int Kt = smap<MODE>(qt_start - E + e, g.To);
    const float* fl = ci + (int64_t)(Kt >= 0 ? Kt : 0) * g.cs0;          // block-uniform bases: the loads are saddr + 32-bit offset
    const float* fh = fl + g.cs_band;
#pragma unroll
    for (int k = 0; k < WMAX; ++k) {
      if (256 * k >= n_w) continue;             // block-uniform: item slots past the tile's count neither load nor compute
#pragma unroll
      for (int i = 0; i < HL; ++i) { pl[k][i] = fl[w_off[k][i]]; ph_[k][i] = fh[w_off[k][i]]; }
    }
  };
  // H-pass items of this thread: (bt, q-row, column) -> BOTH output rows 2 ql, 2 ql + 1 from the same L reads of S1 (each chain in the
  // order of the per-axis kernel); offsets fixed for all frames
  constexpr int HMAX = ITEMS;                   // 2 NQH NW <= 256 ITEMS by the host's choice of NQH
  int h_src[HMAX], h_dst[HMAX];
#pragma unroll
  for (int k = 0; k < HMAX; ++k) {
    const int it = threadIdx.x + 256 * k;
    const int q = fd_div(it, g.dFW);            // bt * NQH + ql
    const int pw = it - q * NW;
    const int bt = q >= NQH ? 1 : 0, ql = q - bt * NQH;
    const bool ok = it < 2 * NQH * NW && ql < nqh;
    h_src[k] = ok ? ((bt * 2) * EH + ql + E) * NW + pw : -1;
    h_dst[k] = (bt * 2 * NQH + 2 * ql) * NW + pw;
  }
  float acc[ITEMS][NQ][2];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j)
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) { acc[j][qq][0] = 0.f; acc[j][qq][1] = 0.f; }
  fetch(ET - 1);
#pragma unroll
  for (int e = ET - 1; e >= 0; --e) {
    const int Kt = smap<MODE>(qt_start - E + e, g.To);
    const bool live = Kt >= 0;                  // a frame outside the tensor contributes fmaf(0, tap, acc) = acc (block-uniform)
    if (live) {
      // pass W: the prefetched coefficients of frame Kt -> S1
#pragma unroll
      for (int k = 0; k < WMAX; ++k) {
        if (256 * k >= n_w) continue;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          const bool ok = (w_ok[k] >> i) & 1u;
          const float cl = ok ? pl[k][i] : 0.f, ch = ok ? ph_[k][i] : 0.f;
          a0 = fmaf(cl, t.lo[2 * i], a0);
          a0 = fmaf(ch, t.hi[2 * i], a0);
          a1 = fmaf(cl, t.lo[2 * i + 1], a1);
          a1 = fmaf(ch, t.hi[2 * i + 1], a1);
        }
        if (w_dst[k] >= 0) *reinterpret_cast<float2*>(S1 + w_dst[k]) = make_float2(a0, a1);
      }
    }
    if (e > 0) fetch(e - 1);
    if (!live) continue;
    __syncthreads();
    // pass H: S1 -> S2
#pragma unroll
    for (int k = 0; k < HMAX; ++k) {
      if (h_src[k] < 0) continue;
      const float* cl = S1 + h_src[k];
      const float* ch = cl + EH * NW;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const float vl = cl[-i * NW], vh = ch[-i * NW];
        a0 = fmaf(vl, t.lo[2 * i], a0);
        a0 = fmaf(vh, t.hi[2 * i], a0);
        a1 = fmaf(vl, t.lo[2 * i + 1], a1);
        a1 = fmaf(vh, t.hi[2 * i + 1], a1);
      }
      S2[h_dst[k]] = a0;
      S2[h_dst[k] + NW] = a1;
    }
    __syncthreads();
    // pass T: this frame's term of every q-frame it belongs to (i = qq + E - e in [0, E])
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int it = threadIdx.x + 256 * j;
      if (it < n_items) {
        const float cl = S2[it], ch = S2[2 * NQH * NW + it];        // (ph, pw) = it / NW, it % NW: S2 rows are NW wide, so the item index is the offset
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          const int i = qq + E - e;
          if (i >= 0 && i <= E) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              acc[j][qq][r] = fmaf(cl, t.lo[r + 2 * i], acc[j][qq][r]);
              acc[j][qq][r] = fmaf(ch, t.hi[r + 2 * i], acc[j][qq][r]);
            }
          }
        }
      }
    }
  }
  const int wshift = off & 1;
  const int64_t fs = (int64_t)g.H * g.W;
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int it = threadIdx.x + 256 * j;
    if (it >= n_items) continue;
    const int ph = fd_div(it, g.dFW), pw = it - ph * NW;
    const int nw = pw - wshift;
    const int nhh = 2 * (qh_start + (ph >> 1)) + (ph & 1) - off;
    if (nw < 0 || nw >= g.W || nhh < 0 || nhh >= g.H) continue;
    float* o = xi + (int64_t)nhh * g.W + nw;
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int nt = 2 * (qt_start + qq) + r - off;
        if (nt >= 0 && nt < g.T) o[nt * fs] = acc[j][qq][r];
      }
  }
}

// LDS budget per block (<= 64 KB: the default dynamic-LDS limit, no function attribute needed -> graph-capture safe). Smaller
// tiles = more blocks per CU to hide the latency of the global reads, at the price of more halo re-reads through L2.
static size_t fused_lds_budget(long default_kb) {
  static long env_kb = -1;
  if (env_kb < 0) {
    const char* e = getenv("WDNO_DWT_LDS_KB");        // A/B switch (tools/bench_dwt.py)
    env_kb = e ? atol(e) : 0;
  }
  long kb = env_kb > 0 ? env_kb : default_kb;
  if (kb < 8) kb = 8;
  if (kb > 64) kb = 64;
  return (size_t)kb * 1024;
}

struct PackedStore { int img_inner, row_w; int64_t cs_outer; const float* resc; };     // wdno_dwt_fwd_packed

template <int ND, int L, int MODE>
static bool fused_analysis(const float* src, float* dst, const wdno_dwt_desc* d, const Taps& taps, bool odd_rule, hipStream_t st, const PackedStore* pk = nullptr) {
  FusedGeom g = {};
  g.n_img = d->n_img;
  g.T = d->in_dims[0]; g.H = d->in_dims[1]; g.W = d->in_dims[2];
  g.To = d->out_dims[0]; g.Ho = d->out_dims[1]; g.Wo = d->out_dims[2];
  g.cs_img = d->cs_img; g.cs_band = d->cs_band; g.cs0 = d->cs0; g.cs1 = d->cs1;
  g.off = MODE == 1 ? L - 2 : L / 2 - 1;
  const bool odd = MODE == 0 && odd_rule;
  g.odd_t = odd && (g.T & 1); g.odd_h = odd && (g.H & 1); g.odd_w = odd && (g.W & 1);
  g.FW = (ND == 2 && MODE == 0) ? 2 * g.Wo : 2 * g.Wo + L - 2;        // 2-D periodization: one period per LDS row (see the kernel)
  if ((int64_t)g.T * g.H * g.W >= (1ll << 31)) return false;
  constexpr int NK = (ND == 3) ? 3 : 4;
  size_t lds;
  if (ND == 3) {
    // floats: 2 NK FW (FH + 2 NH) with FH = 2 NH + L - 2
    const size_t per_row = (size_t)2 * NK * g.FW * sizeof(float);
    const long nh_max = ((long)(fused_lds_budget(32) / per_row) - (L - 2)) / 4;
    if (nh_max < 1) return false;
    const int tiles = cdiv(g.Ho, (int)std::min<long>(nh_max, g.Ho));
    g.NH = cdiv(g.Ho, tiles);
    g.tiles_h = tiles;
    g.tiles_t = cdiv(g.To, NK);
    lds = per_row * (size_t)(4 * g.NH + L - 2);
  } else {
    const size_t per_row = (size_t)2 * g.FW * sizeof(float);
    long nh_max = (long)(fused_lds_budget(32) / per_row);
    nh_max = std::min<long>(nh_max / NK * NK, 2 * NK);        // two register groups per block: plenty of blocks for small images
    if (nh_max < NK) return false;
    g.NH = (int)nh_max;
    g.tiles_h = cdiv(g.Ho, g.NH);
    g.tiles_t = 1;
    lds = per_row * (size_t)g.NH;
  }
  const int64_t nb = (int64_t)g.n_img * g.tiles_t * g.tiles_h;
  if (nb > 0x7fffffff) return false;
  g.n_blocks = (int)nb;
  g.dFW = make_fastdiv(g.FW); g.dNH = make_fastdiv(g.NH); g.dWo = make_fastdiv(g.Wo); g.dQW4 = make_fastdiv((g.Wo + 3) / 4);
  if ((int64_t)2 * NK * 2 * g.NH * std::max(g.FW, g.Wo) * (int64_t)std::max(g.FW, g.NH) >= (1ll << 32)) return false;   // fd_div range
  if (pk) {
    if constexpr (ND == 3 && MODE == 1) {        // the smoke pipeline's transform (zero mode, 3-D); other shapes: not instantiated
      g.img_inner = pk->img_inner; g.cs_outer = pk->cs_outer; g.resc = pk->resc; g.row_w = pk->row_w;
      g.dInner = make_fastdiv(g.img_inner);
      g.dQW4 = make_fastdiv(g.row_w / 4);
      if (g.row_w < g.Wo || (g.row_w & 3) || ((g.cs_img | g.cs_band | g.cs0 | g.cs1 | g.cs_outer) & 3) || ((uintptr_t)dst & 15)) return false;
      if ((int64_t)2 * NK * 2 * g.NH * g.row_w * (int64_t)std::max(g.FW, g.NH) >= (1ll << 32)) return false;   // fd_div range
      if ((int64_t)g.n_img * g.img_inner >= (1ll << 32)) return false;
      dwt_analysis_fused_kernel<ND, L, MODE, NK, true><<<(int)nb, 256, lds, st>>>(src, dst, g, taps);
      return true;
    }
    return false;
  }
  dwt_analysis_fused_kernel<ND, L, MODE, NK><<<(int)nb, 256, lds, st>>>(src, dst, g, taps);
  return true;
}

template <int ND, int L, int MODE>
static bool fused_synthesis(const float* src, float* dst, const wdno_dwt_desc* d, const Taps& taps, hipStream_t st) {
  FusedGeom g = {};
  g.n_img = d->n_img;
  g.T = d->in_dims[0]; g.H = d->in_dims[1]; g.W = d->in_dims[2];
  g.To = d->out_dims[0]; g.Ho = d->out_dims[1]; g.Wo = d->out_dims[2];
  g.cs_img = d->cs_img; g.cs_band = d->cs_band; g.cs0 = d->cs0; g.cs1 = d->cs1;
  g.off = MODE == 1 ? L - 2 : L / 2 - 1;
  constexpr int E = L / 2 - 1;
  auto qcount = [&](int N) { return ((N - 1 + g.off) >> 1) - (g.off >> 1) + 1; };
  g.qt0 = g.qh0 = g.qw0 = g.off >> 1;
  g.QH = qcount(g.H);
  g.QT = qcount(g.T);
  g.FW = 2 * qcount(g.W);
  if (g.cs_img >= (1ll << 31) || (int64_t)g.T * g.H * g.W >= (1ll << 31)) return false;      // 32-bit offsets within an image
  constexpr int NQ = (ND == 3) ? 4 : 4;
  size_t lds;
  if (ND == 3 && wdno_debug_mode != 45) {
    // streaming kernel: q-rows per tile from the register budget (ITEMS columns of 2 NQ samples per thread) and a 32 KB LDS budget
    auto stream = [&](auto ITEMS_C, auto NQ_C) -> bool {
      constexpr int ITEMS = decltype(ITEMS_C)::value, NQS = decltype(NQ_C)::value;
      const int NW = g.FW;
      int nqh = (ITEMS * 256) / (2 * NW);
      const int lds_rows = (int)(fused_lds_budget(32) / ((size_t)NW * sizeof(float)));      // 4 (NQH + E) + 4 NQH rows of NW floats
      nqh = std::min(nqh, (lds_rows - 4 * E) / 8);
      nqh = std::min(nqh, g.QH);
      if (nqh < 1 || g.cs_img >= (1ll << 31)) return false;
      const int tiles = cdiv(g.QH, nqh);
      g.NH = cdiv(g.QH, tiles);
      g.tiles_h = tiles;
      g.tiles_t = cdiv(g.QT, NQS);
      const size_t lds = (size_t)(4 * (g.NH + E) + 4 * g.NH) * NW * sizeof(float);
      const int64_t nbs = (int64_t)g.n_img * g.tiles_t * g.tiles_h;
      const int n_w = 4 * (g.NH + E) * (NW / 2);               // W-pass items of a frame: WMAX per thread, held in registers
      if (nbs > 0x7fffffff || n_w > 8 * 256 || (int64_t)4 * (g.NH + E) * NW * (int64_t)std::max(NW, 2 * (g.NH + E)) >= (1ll << 32)) return false;
      g.n_blocks = (int)nbs;
      g.dFW = make_fastdiv(NW); g.dQW = make_fastdiv(NW / 2); g.dEH = make_fastdiv(g.NH + E); g.dNQH2 = make_fastdiv(2 * g.NH);
      g.debug = 0;
      if (n_w <= 4 * 256) dwt_synthesis3_stream_kernel<L, MODE, NQS, ITEMS, 4><<<(int)nbs, 256, lds, st>>>(src, dst, g, taps);
      else dwt_synthesis3_stream_kernel<L, MODE, NQS, ITEMS, 8><<<(int)nbs, 256, lds, st>>>(src, dst, g, taps);
      return true;
    };
    // tile shape: 2 columns per thread, NQ = 4 q-frames per tile. Measured on [32,8,18,34,34] / [10,8,34,66,66] with this kernel (us):
    // 2 col NQ 4: 16.1 / 33.3; 3 col: 17.2 / 42.0; 4 col: 19.7 / 37.4; 2 col NQ 8: 17.8 / 31.8; 3 col NQ 8: 21.5 / 47.1
    const bool ok = stream(std::integral_constant<int, 2>{}, std::integral_constant<int, NQ>{});
    if (ok) return true;
  }
  if (ND == 3) {
    // floats: 2 ET NW (2 EH + 2 NQH), EH = NQH + E
    const size_t per_row = (size_t)2 * (NQ + E) * g.FW * sizeof(float);
    const long nq_max = ((long)(fused_lds_budget(40) / per_row) - 2 * E) / 4;
    if (nq_max < 1) return false;
    const int tiles = cdiv(g.QH, (int)std::min<long>(nq_max, g.QH));
    g.NH = cdiv(g.QH, tiles);
    g.tiles_h = tiles;
    g.tiles_t = cdiv(g.QT, NQ);
    lds = per_row * (size_t)(4 * g.NH + 2 * E);
  } else {
    const size_t per_row = (size_t)2 * g.FW * sizeof(float);
    long nq_max = (long)(fused_lds_budget(40) / per_row) - E;
    nq_max = std::min<long>(nq_max / NQ * NQ, 2 * NQ);
    if (nq_max < NQ) return false;
    g.NH = (int)nq_max;                                       // NH + E <= 2 NQ + E <= 16: the staged W pass keeps one register per row of a wave
    g.tiles_h = cdiv(g.QH, g.NH);
    g.tiles_t = 1;
    lds = per_row * (size_t)(g.NH + E) + (size_t)4 * (g.NH + E) * (4 * ((g.FW / 2 + 3) / 4) + L / 2 - 1) * sizeof(float);      // S1 + the staged raw rows
  }
  const int64_t nb = (int64_t)g.n_img * g.tiles_t * g.tiles_h;
  if (nb > 0x7fffffff) return false;
  g.n_blocks = (int)nb;
  g.dFW = make_fastdiv(g.FW); g.dQW = make_fastdiv(g.FW / 2); g.dEH = make_fastdiv(g.NH + E); g.dNQH2 = make_fastdiv(2 * g.NH);
  g.dQW4 = make_fastdiv((g.FW / 2 + 3) / 4); g.dRW = make_fastdiv(4 * ((g.FW / 2 + 3) / 4) + L / 2 - 1);
  g.debug = wdno_debug_mode;
  if (ND == 2 && wdno_debug_mode != 50) {                     // debug 50: the one-tile-per-block kernel (A/B)
    const int RW = 4 * ((g.FW / 2 + 3) / 4) + L / 2 - 1;
    if (g.NH + E <= 2 * NQ + E && RW <= 256) {
      const int grid = (int)nb;
      if (RW <= 128) dwt_synthesis2_kernel<L, MODE, NQ, 1><<<grid, 256, lds, st>>>(src, dst, g, taps);
      else dwt_synthesis2_kernel<L, MODE, NQ, 2><<<grid, 256, lds, st>>>(src, dst, g, taps);
      return true;
    }
  }
  if ((int64_t)2 * (NQ + E) * 2 * (g.NH + E) * g.FW * (int64_t)std::max(g.FW, 2 * (g.NH + E)) >= (1ll << 32)) return false;   // fd_div range
  dwt_synthesis_fused_kernel<ND, L, MODE, NQ><<<(int)nb, 256, lds, st>>>(src, dst, g, taps);
  return true;
}

template <int ND, int L>
static bool fused_dispatch2(bool analysis_dir, const float* src, float* dst, const wdno_dwt_desc* d, const Taps& taps, bool odd_rule, hipStream_t st,
                            const PackedStore* pk) {
  if (d->mode == 1) return analysis_dir ? fused_analysis<ND, L, 1>(src, dst, d, taps, odd_rule, st, pk) : fused_synthesis<ND, L, 1>(src, dst, d, taps, st);
  return analysis_dir ? fused_analysis<ND, L, 0>(src, dst, d, taps, odd_rule, st, pk) : fused_synthesis<ND, L, 0>(src, dst, d, taps, st);
}
template <int ND>
static bool fused_dispatch(bool analysis_dir, const float* src, float* dst, const wdno_dwt_desc* d, const Taps& taps, bool odd_rule, hipStream_t st,
                           const PackedStore* pk) {
  switch (d->L) {
    case 2: return fused_dispatch2<ND, 2>(analysis_dir, src, dst, d, taps, odd_rule, st, pk);
    case 4: return fused_dispatch2<ND, 4>(analysis_dir, src, dst, d, taps, odd_rule, st, pk);
    case 6: return fused_dispatch2<ND, 6>(analysis_dir, src, dst, d, taps, odd_rule, st, pk);
    case 8: return fused_dispatch2<ND, 8>(analysis_dir, src, dst, d, taps, odd_rule, st, pk);
    case 10: return fused_dispatch2<ND, 10>(analysis_dir, src, dst, d, taps, odd_rule, st, pk);
    default: return false;
  }
}
// true = the transform was launched as ONE fused kernel; false = not covered (1-D, long filters, rows that do not fit LDS, the
// odd-length synthesis-shaped adjoint), the caller falls back to the per-axis passes.
static bool fused_try(bool analysis_dir, const float* src, float* dst, const wdno_dwt_desc* d, const Taps& taps, bool odd_rule, hipStream_t st,
                      const PackedStore* pk = nullptr) {
  if ((wdno_debug_mode == 11 && !pk) || d->nd < 2) return false;
  if (!analysis_dir && d->mode == 0 && odd_rule)
    for (int a = 3 - d->nd; a < 3; ++a)
      if (d->in_dims[a] & 1) return false;
  if (!analysis_dir && d->mode == 0)
    for (int a = 3 - d->nd; a < 3; ++a)
      if (d->in_dims[a] != 2 * d->out_dims[a]) return false;
  return d->nd == 3 ? fused_dispatch<3>(analysis_dir, src, dst, d, taps, odd_rule, st, pk) : fused_dispatch<2>(analysis_dir, src, dst, d, taps, odd_rule, st, pk);
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
  Taps taps;
  fill_taps(taps, fl, fh, d->L, reverse_taps);
  if (fused_try(analysis_dir, src, dst, d, taps, odd_rule, st)) return wdno_check_launch();
  if (ws_elems(d) * sizeof(float) > ws_bytes) return WDNO_EWORKSPACE;
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
// Analysis with the coefficients stored straight into a larger, differently ordered tensor and divided by a per-channel constant: image
// i = (outer, inner) = (i / img_inner, i % img_inner) writes at outer * cs_outer + inner * d->cs_img (+ band * cs_band + k0 * cs0 + k1 * cs1 + k2),
// value / rescaler[inner * 8 + band] (IEEE division); rows are written row_w (>= the coefficient count, % 4 == 0) columns wide, 0 / rescaler beyond
// the coefficients (whole 16-byte stores: strides and `state` 16-byte aligned). The smoke task's state [B][pad_t][8 F + 2][pad_x][pad_x] takes the F fields of a sample
// with cs_outer = pad_t C pad_x^2, cs_img = 8 pad_x^2, cs_band = pad_x^2, cs0 = C pad_x^2, cs1 = pad_x (data_2d.py:156-221: cat, pad, permute,
// / RESCALER -- the padding and the two condition channels are wdno_pack_smoke_fill's). ONE fused launch, 3-D zero mode only: anything the fused
// kernel does not take is WDNO_EUNSUPPORTED (the caller then transforms into a coefficient tensor and packs, wdno_pack_smoke_fields).
extern "C" int wdno_dwt_fwd_packed(const float* x, float* state, const wdno_dwt_desc* d, const float* f, int img_inner, int64_t cs_outer, int row_w,
                                   const float* rescaler, wdno_stream_t s) {
  int rc = validate(d);
  if (rc) return rc;
  WDNO_REQUIRE(x && state && f && rescaler && img_inner > 0 && d->n_img % img_inner == 0);
  if (d->nd != 3 || d->mode != 1) return WDNO_EUNSUPPORTED;
  for (int a = 0; a < 3; ++a)
    if (d->out_dims[a] != (d->in_dims[a] + d->L - 1) / 2) return WDNO_EINVAL;
  Taps taps;
  fill_taps(taps, f, f + d->L, d->L, false);
  PackedStore ps = {img_inner, row_w, cs_outer, rescaler};
  if (!fused_try(true, x, state, d, taps, true, as_stream(s), &ps)) return WDNO_EUNSUPPORTED;
  return wdno_check_launch();
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
