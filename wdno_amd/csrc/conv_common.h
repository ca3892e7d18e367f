// conv_common.h -- geometry helpers shared by the exact-fp32 and the 3xfp16-split implicit-GEMM convolution kernels.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define LDS_STRIDE 36   // BK + 4 floats: 144-byte rows -> 16 consecutive rows hit 16 distinct 16-B slots

struct ConvP {
  wdno_conv_geom g;
  int R;          // kw * C
  int nchunk;     // ceil(R / BK)
  int nsteps;     // kd*kh*nchunk
  int64_t P;      // N*OD*OH*OW
  int identity_out;
  int debug;      // diagnostics only (tools/bench_conv.py): 1 = skip global loads, 2 = skip MFMAs
  int tiles_n;
  int ntiles;
  int ksplit;     // register-staged split kernels: tap rows dealt to blockIdx.y = 0 .. ksplit-1, partial sums added with atomics (1 = off)
  float* amax_rec; // optional amax record of the output (common.h), filled by the epilogue; nullptr = not wanted
  // tap-resident kernel (conv_h3t.hip): the reduction of a tile cut into tsplit runs of stages, one block each; run s writes its raw partial
  // sums to split_ws[s][P][K] and conv_split_reduce_kernel adds them in order (1 = off). split_ws / split_ws_bytes: what the caller lent.
  int tsplit;
  float* split_ws;
  size_t split_ws_bytes;
  // structural zeros of x (include/wdno_hip.h: wdno_zero_box), used by the 7-wide tap-resident kernel: the first zb_blocks 16-channel blocks of
  // x are zero wherever d >= zb_d or h >= zb_h, so their stages with such a source plane / row for every pixel of a tile are not run
  // (0 blocks = off; stages whose source plane / row lies outside the grid for the whole tile are skipped for every block then, too)
  int zb_blocks, zb_d, zb_h;
  int out_bf16;   // single-product mode: y is bf16 storage [rows][K] (round-to-nearest-even of the fp32 result; no residual, no split reduction)
};

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // bijective remap so that each of the 8 XCDs (block b runs on XCD b % 8) gets a contiguous range of tiles
  int q = nwg >> 3, r = nwg & 7;
  int xcd = bid & 7, idx = bid >> 3;
  int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

__device__ __forceinline__ int64_t out_row(const wdno_conv_geom& g, int64_t p) {
  int ow = (int)(p % g.OW); int64_t t = p / g.OW;
  int oh = (int)(t % g.OH); t /= g.OH;
  int od = (int)(t % g.OD);
  int64_t n = t / g.OD;
  return ((n * g.YD + (od * g.osd + g.ood)) * g.YH + (oh * g.osh + g.ooh)) * g.YW + (ow * g.osw + g.oow);
}

static inline int check_geom(const wdno_conv_geom* g) {
  if (!g) return WDNO_EINVAL;
  if (g->N <= 0 || g->D <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->K <= 0) return WDNO_EINVAL;
  if (g->OD <= 0 || g->OH <= 0 || g->OW <= 0 || g->kd <= 0 || g->kh <= 0 || g->kw <= 0) return WDNO_EINVAL;
  if (g->sd <= 0 || g->sh <= 0 || g->sw <= 0 || g->osd <= 0 || g->osh <= 0 || g->osw <= 0) return WDNO_EINVAL;
  if ((g->C & 3) || (g->K & 3)) return WDNO_EUNSUPPORTED;   // rows must be 16-byte multiples (callers pad)
  if ((g->OD - 1) * g->osd + g->ood >= g->YD || (g->OH - 1) * g->osh + g->ooh >= g->YH || (g->OW - 1) * g->osw + g->oow >= g->YW)
    return WDNO_EINVAL;
  return WDNO_OK;
}
extern int wdno_debug_mode;
// conv_wgrad_h3d.hip: LDS-DMA variant of the split-fp16 weight-gradient kernel and its split plan
void wdno_wgrad_h3d_plan(const wdno_conv_geom* g, int* bm, int* bn, int* splits, int* pix_per_split, int* splitpair = nullptr);
int wdno_conv_wgrad_h3_dma(const void* xh, const void* xl, const void* dyh, const void* dyl, const float* sx, const float* sdy,
                           const void* table, float* wsf, const wdno_conv_geom* g, hipStream_t st);
// conv_h3d.hip: LDS-DMA variant of the split-fp16 forward / data-gradient kernel (WDNO_EUNSUPPORTED -> use conv_h3.hip's)
int wdno_conv_fwd_h3_dma(const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                         const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st, int stages);
// conv_h3t.hip: tap-resident variant for stride-1 convolutions on equal grids (called by conv_h3d.hip with the tile shape it chose)
bool wdno_conv_h3t_takes(const wdno_conv_geom& g);
int wdno_conv_h3t_split(const wdno_conv_geom& g, int64_t P, int cus);
int wdno_conv_fwd_h3_tap(int shape, const void* xh, const void* xl, const void* wh, const void* wl, const float* sx, const float* sw,
                         const float* bias, const float* residual, float* y, ConvP& p, hipStream_t st);
#ifdef __HIPCC__
// s_waitcnt lgkmcnt(0) + s_barrier. The wait is the BUILTIN so that the compiler's own s_waitcnt bookkeeping sees it: behind an
// asm-only wait it still counts the scalar loads of the previous tile's epilogue as possibly outstanding at the head of the
// step loop, and (scalar loads return out of order) protects the first MFMA of every step with lgkmcnt(0) -- a wait for the
// LDS reads issued just before it.
// bf16 storage of two accumulator runs (channels 8 e + 4 hh .. + 3 and 8 (e + 1) + 4 hh .. + 3 of one pixel, the two lane halves hh = 0 / 1
// holding the two halves of each group of eight): the halves trade runs (v_permlane32_swap), so that the lower half ends up with the eight
// channels of group e and the upper half with those of group e + 1 -- ONE 16-byte store per lane instead of two 8-byte ones (with the
// 8-byte stores the 256 x 128 single-plane kernel of the Burgers U-Net lost 2.7 %: 356 vs 366 us per launch).
__device__ __forceinline__ uint4 bf16x8_from_runs(float4 va, float4 vb) {
  const uint2 A = bf16x4_pack(va), B = bf16x4_pack(vb);
  const auto sx = __builtin_amdgcn_permlane32_swap(A.x, B.x, false, false);
  const auto sy = __builtin_amdgcn_permlane32_swap(A.y, B.y, false, false);
  return make_uint4(sx[0], sy[0], sx[1], sy[1]);
}
__device__ __forceinline__ void lgkm0_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  asm volatile("s_barrier" : : : "memory");
}
#endif
static inline void fill_params(ConvP& p, const wdno_conv_geom* g) {
  p.g = *g;
  p.amax_rec = nullptr;
  p.ksplit = 1;
  p.tsplit = 1;
  p.split_ws = nullptr;
  p.split_ws_bytes = 0;
  p.zb_blocks = 0; p.zb_d = 0; p.zb_h = 0;
  p.out_bf16 = 0;
  p.debug = wdno_debug_mode;
  p.R = g->kw * g->C;
  p.nchunk = cdiv(p.R, BK);
  p.nsteps = g->kd * g->kh * p.nchunk;
  p.P = (int64_t)g->N * g->OD * g->OH * g->OW;
  p.identity_out = (g->YD == g->OD && g->YH == g->OH && g->YW == g->OW && g->osd == 1 && g->osh == 1 && g->osw == 1 &&
                    g->ood == 0 && g->ooh == 0 && g->oow == 0) ? 1 : 0;
}

