/* wdno_hip.h -- C ABI of libwdno_hip.so: hand-written gfx950 (MI355X) kernels for the WDNO hot path.
 *
 * The reference (AI4Science-WestlakeU/wdno) has no FFI of its own: its hot path is stock torch.nn + three
 * third-party wavelet packages. Each entry point below therefore cites the reference *operator* it replaces
 * (file:line relative to the reference root). INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a device pointer to contiguous fp32 unless stated; int64 for timestep indices;
 *   - "CL" = channels-last activations [N, (D,) H, W, C]; API layout = the reference's own tensor layout;
 *   - every call enqueues on `stream` and returns immediately: 0 on success, <0 = WDNO_E* (wdno_strerror);
 *   - no call allocates, synchronises or throws. Workspaces are caller-provided (size queried by *_ws_bytes).
 */
#ifndef WDNO_HIP_H
#define WDNO_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wdno_stream_t; /* hipStream_t */

enum { WDNO_OK = 0, WDNO_EINVAL = -1, WDNO_ELAUNCH = -2, WDNO_EUNSUPPORTED = -3, WDNO_EWORKSPACE = -4 };

/* "amax record": WDNO_AMAX_FLOATS floats in device memory (WDNO_AMAX_SLOTS slots, one per 64-byte line so that the atomic
 * updates of a launch do not serialise), zeroed by the caller, whose maximum is max|x| of one tensor. The
 * fp32-equivalent convolutions scale their fp16 split by it. wdno_amax_record fills one with a sweep over x; the *_amax
 * forms of the kernels that WRITE activation tensors (group norm, layer norm, concat, add) fill one for their output on
 * the way (amax_rec may be NULL: not wanted), which saves that sweep. */
#define WDNO_AMAX_SLOTS 64
#define WDNO_AMAX_STRIDE 16
#define WDNO_AMAX_FLOATS (WDNO_AMAX_SLOTS * WDNO_AMAX_STRIDE)
const char* wdno_strerror(int code);
int wdno_version(void);
/* last hip error string seen by the library on this thread (diagnostics only) */
const char* wdno_last_hip_error(void);
/* diagnostics only: ablation switches of the convolution kernels (0 = off) */
int wdno_set_debug(int mode);

/* ------------------------------------------------------------------------------------------------ wavelets
 * Separable single-level filter banks. mode: 0 = periodization, 1 = zero padding. Filters: 4 x L floats
 * (dec_lo, dec_hi, rec_lo, rec_hi) on the HOST (copied into kernel arguments; L <= 16).
 *
 * wdno_dwt_fwd : x [n_img, dims...] -> coef [n_img, 2^nd, out dims...] with sub-bands stacked on axis 1 in
 *   pywt order (letters over axes slowest..fastest, a before d): 1-D (lo,hi); 2-D (LL,'da','ad','dd') =
 *   coef_to_tensor order burgers/wave_trans.py:43-62 ; 3-D 'aaa'..'ddd' = smoke/wave_trans_2d.py:55-58.
 *   Replaces pytorch_wavelets.DWTForward/DWT1DForward (burgers/wave_trans.py:94-107, data_burgers_1d.py:72)
 *   and ptwt.wavedec3 (smoke/wave_trans_2d.py:129). The packed output may be strided/padded:
 *   coef element (img, band, i0,i1,i2) lives at img*cs_img + band*cs_band + i0*cs0 + i1*cs1 + i2.
 * wdno_dwt_inv : the inverse (DWTInverse / DWT1DInverse / ptwt.waverec3), reading coef with the same strides.
 * wdno_dwt_inv_adjoint : d(loss)/d(coef) given d(loss)/d(x) for x = wdno_dwt_inv(coef)  (guidance back-prop,
 *   burgers/ddpm_burgers/model_utils.py:35-50, smoke/inference_2d.py:30-66).
 * wdno_dwt_fwd_adjoint : d(loss)/d(x) given d(loss)/d(coef) for coef = wdno_dwt_fwd(x).
 */
typedef struct {
  int nd;             /* 1, 2 or 3 transformed axes (the trailing ones) */
  int mode;           /* 0 periodization, 1 zero */
  int L;              /* filter length */
  int n_img;          /* product of all leading dims */
  int in_dims[3];     /* signal extent along each transformed axis (unused leading entries = 1) */
  int out_dims[3];    /* coefficient extent along each axis */
  int64_t cs_img, cs_band, cs0, cs1;   /* coefficient tensor strides (elements); innermost stride is 1.
                                          the signal tensor x is contiguous [n_img, in_dims...] */
} wdno_dwt_desc;
size_t wdno_dwt_ws_bytes(const wdno_dwt_desc* d);
int wdno_dwt_fwd(const float* x, float* coef, const wdno_dwt_desc* d, const float* filters_host, void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_dwt_inv(const float* coef, float* x, const wdno_dwt_desc* d, const float* filters_host, void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_dwt_fwd_adjoint(const float* dcoef, float* dx, const wdno_dwt_desc* d, const float* filters_host, void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_dwt_inv_adjoint(const float* dx, float* dcoef, const wdno_dwt_desc* d, const float* filters_host, void* ws, size_t ws_bytes, wdno_stream_t s);
/* 3-D zero-mode analysis whose coefficients go straight into a larger, differently ordered tensor, divided by a per-channel constant (the smoke
 * task's network input: smoke/ddpm/data_2d.py:156-221 does cat + pad + permute + `/ RESCALER` on the transform's output; here the transform's store
 * does it). Image i = (outer, inner) = (i / img_inner, i % img_inner) is stored at state + outer * cs_outer + inner * d->cs_img + band * d->cs_band +
 * k0 * d->cs0 + k1 * d->cs1 + k2, as value / rescaler[inner * 8 + band] (IEEE division: the bits of torch's `coef / RESCALER`). Rows are written
 * row_w columns wide (row_w >= out_dims[2], row_w % 4 == 0; 0 / rescaler beyond the coefficients; all strides % 4 == 0, state 16-byte aligned:
 * aligned 16-byte stores). One fused launch;
 * WDNO_EUNSUPPORTED for anything it does not take (nd != 3, mode != zero, filters longer than 10 taps, rows too wide for LDS): the caller then
 * transforms into a coefficient tensor (wdno_dwt_fwd) and packs (wdno_pack_smoke_fields). Nothing but the rows [k0 < out_dims[0]][k1 < out_dims[1]] of
 * the 8 channels of an image is written: wdno_pack_smoke_fields(coef = NULL, ...) fills the rest of the state. */
int wdno_dwt_fwd_packed(const float* x, float* state, const wdno_dwt_desc* d, const float* filters_host, int img_inner, int64_t cs_outer, int row_w,
                        const float* rescaler, wdno_stream_t s);
/* nearest x2 up-sampling of coefficient tensors (burgers/ddpm_burgers/wavelet_utils.py:5-16,
 * smoke/ddpm/wave_utils.py:1-14): in [outer, a, mid, b, c] -> out [outer, a*fa, mid, b*fb, c*fc]. */
int wdno_upsample_coef(const float* in, float* out, int64_t outer, int a, int mid, int b, int c, int fa, int fb, int fc, wdno_stream_t s);
/* Dataset packing of the smoke task for a batch, one launch (smoke/ddpm/data_2d.py:156-221, Smoke_wave.__getitem__ of a base-resolution model):
 *   state[b][f][c][h][w] = value / rescaler[c],   state [B][pad_t][8 F + 2][pad_x][pad_x]
 *   c <  8 F    : coef[sim][c / 8][c % 8][f][h][w] inside the [nt][nx][nx] box, 0 in the padding           (coef [.][F][8][nt][nx][nx])
 *   c == 8 F    : init_coef[sim][f / (pad_t / 4)][h][w] inside [nx][nx], 0 outside                          (the 4 sub-bands of the DWT of rho(t = 0))
 *   c == 8 F + 1: smokeout[sim][h >= pad_x / 2][f] for f < nt, 0 beyond                                     (smokeout [.][2][nt])
 * sim = idx[b] (device int64; NULL: b) selects the simulation inside resident stores whose per-simulation strides (in elements) are given --
 * the stores may hold the files exactly as the offline transform wrote them ([5][4][nx][nx] initial coefficients: field 0 is read).
 * IEEE division: bit-identical to the torch formulation. pad_t % 4 == 0, pad_x % 4 == 0. */
int wdno_pack_smoke_state(const float* coef, int64_t coef_sim_stride, const float* init_coef, int64_t init_sim_stride, const float* smokeout,
                          int64_t so_sim_stride, const int64_t* idx, const float* rescaler, float* state, int64_t B, int F, int nt, int nx,
                          int pad_t, int pad_x, wdno_stream_t s);
/* The same with the two condition channels transformed inside the launch from the physical inputs (the online pipeline fields -> DWT -> state,
 * smoke/wave_trans_2d.py:150-170 + data_2d.py:156-221): rho0 [.][H0][W0] = the density field at t = 0 (e.g. a view into the fields tensor),
 * curve [.][T0] = the smoke-out curve; zero-mode analysis with the decomposition filters (host arrays of L <= 16 taps).
 * coef == NULL: the rows [f < nt][h < nx] of the 8 F channels are left untouched (wdno_dwt_fwd_packed wrote them pad_x wide, already divided); the
 * zero rows / frames around them and the two condition channels are written -- together the two launches produce the same bits as the one-tensor form. */
int wdno_pack_smoke_fields(const float* coef, int64_t coef_sim_stride, const float* rho0, int64_t rho0_sim_stride, const float* curve,
                           int64_t curve_sim_stride, const float* dec_lo_host, const float* dec_hi_host, int L, const float* rescaler,
                           float* state, int64_t B, int F, int nt, int nx, int pad_t, int pad_x, int H0, int W0, int T0, wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ layout
 * API layout [N, C, S] (S = product of spatial dims) <-> channels-last [N, S, Cp] (Cp >= C, zero padded). */
int wdno_nc_to_cl(const float* src, float* dst, int64_t N, int C, int64_t S, int Cp, wdno_stream_t s);
int wdno_cl_to_nc(const float* src, float* dst, int64_t N, int C, int64_t S, int Cp, wdno_stream_t s);
/* (a | b) delivered only as the fp16 planes [P][Ca + Cb] (lo == NULL: one bf16 plane) of the convolutions that read it -- the up-path
 * torch.cat((x, h.pop()), dim=1) of conv3d.py:340-346 / unet.py:268-274, read by the next ResnetBlock's first convolution and 1 x 1 skip
 * projection only. Scale from the amax records of a and b (left in scale_out[0]). Ca, Cb multiples of 8. */
int wdno_concat2_cl_planes(const float* a, int Ca, const float* b, int Cb, const float* rec_a, const float* rec_b, void* hi, void* lo,
                           float* scale_out, int64_t P, wdno_stream_t s);
int wdno_concat2_cl(const float* a, int Ca, const float* b, int Cb, float* out, int64_t P, wdno_stream_t s);
int wdno_concat2_cl_amax(const float* a, int Ca, const float* b, int Cb, float* out, float* amax_rec, int64_t P, wdno_stream_t s);
int wdno_split2_cl(const float* in, float* a, int Ca, float* b, int Cb, int64_t P, wdno_stream_t s);
/* nearest x2 in H and W of CL [N, H, W, C] (nn.Upsample, burgers/ddpm_burgers/unet.py:35-39) and its adjoint */
int wdno_upsample2x_cl_fwd(const float* in, float* out, int64_t N, int H, int W, int C, wdno_stream_t s);
int wdno_upsample2x_cl_bwd(const float* dout, float* din, int64_t N, int H, int W, int C, wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ convolution
 * Implicit-GEMM convolution on MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
 * Replaces nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d / nn.Linear forward, data-gradient and weight-gradient
 * (burgers/ddpm_burgers/unet.py:133,162,190-192,233-234,317,336,361,369 ;
 *  smoke/.../video_diffusion_pytorch_conv3d.py:159-163,192,216,238-239,291-292,393,471).
 *
 * Input x is CL [N, D, H, W, C] (C % 4 == 0). Weights are pre-packed as wp[kd][kh][K][kw*C] (innermost run
 * kw*C is contiguous in both x and wp). Output pixel (n, od, oh, ow) reads input rows
 *   d = od*sd - pd + dz, h = oh*sh - ph + dy, w = ow*sw - pw + dx
 * and is stored at physical position (n, od*osd+ood, oh*osh+ooh, ow*osw+oow) of y [N, YD, YH, YW, K]
 * (the placement terms express the 4 parity classes of a stride-2 transposed convolution).
 */
typedef struct {
  int N, D, H, W, C;
  int OD, OH, OW, K;
  int kd, kh, kw;
  int sd, sh, sw;
  int pd, ph, pw;
  int YD, YH, YW;
  int osd, osh, osw, ood, ooh, oow;
} wdno_conv_geom;
/* y = conv(x, wp) (+ bias[K]) (+ residual, same layout as y). bias/residual may be NULL. */
int wdno_conv_fwd(const float* x, const float* wp, const float* bias, const float* residual, float* y,
                  const wdno_conv_geom* g, wdno_stream_t s);
/* fp32-equivalent convolution on the fp16 matrix cores ("3 x fp16 split", conv_h3.hip): an fp32 tensor is pre-split by
 * wdno_amax_record + wdno_split_f16 into two fp16 planes hi, lo [rows][C8] (C8 = C rounded up to 8) and a power-of-two scale;
 * wdno_conv_fwd_f16x3 evaluates ah*bh + ah*bl + al*bh with fp32 accumulation. Same geometry contract as wdno_conv_fwd
 * with g->C = C8; sx / sw are the device scalars written by wdno_split_f16 for the activation / packed-weight operand. */
int wdno_amax(const float* x, int64_t n, float* amax_zeroed, wdno_stream_t s);            /* one float (weights) */
int wdno_amax_record(const float* x, int64_t n, float* rec_zeroed, wdno_stream_t s);     /* an amax record (activations) */
/* amax_rec: the amax record of x (from wdno_amax_record or from the kernel that produced x) */
int wdno_split_f16(const float* x, const float* amax_rec, void* hi, void* lo, float* scale_out, int64_t rows, int C, int C8, wdno_stream_t s);
/* wdno_split_f16 that also returns colsum_out[C8] = sum over rows (the bias gradient when x is dy), in the same pass.
 * WDNO_EUNSUPPORTED unless C8 / 8 is a power of two <= 256 (use wdno_split_f16 + wdno_colsum then).
 * colsum_out == NULL (here and in wdno_cast_bf16_colsum): only the partials are written -- doubles [ws_bytes / (8 C8)][C8] in ws -- and the
 * caller sums them later (wdno_rows_sum_multi, is_double = 1). */
size_t wdno_split_colsum_ws_bytes(int64_t rows, int C8);
int wdno_split_f16_colsum(const float* x, const float* amax_rec, void* hi, void* lo, float* scale_out, float* colsum_out,
                          void* ws, size_t ws_bytes, int64_t rows, int C, int C8, wdno_stream_t s);
/* raw weight [K][C][kd][kh][kw] -> split planes of the packed operand in one launch. mode 0: forward operand
 * [kd][kh][A>=K][kw][B>=C]; mode 1: data-gradient operand [kd][kh][A>=C][kw][B>=K] with flipped taps. amax = max|w| (device).
 * modes 2 + 2 py + px: forward operand of parity class (py, px) of the (1,4,4) / stride (1,2,2) transposed convolution (the stride-1
 * (1,2,2) convolution that produces the output pixels of that parity; Upsample of video_diffusion_pytorch_conv3d.py:96-98), gathered
 * from the ConvTranspose3d weight w[C = in][K = out][1][4][4] itself; kd, kh, kw = 1, 2, 2.
 * modes 10 / 11: the forward operand of a 1 x 1 projection -- to_qkv [A = 384][B = C] / to_out [A = C][B = 128] of a temporal attention block with
 * 4 heads of 32 -- in the FRAGMENT ORDER wdno_tattn_fused_fwd reads at C = 128 / 256 (csrc/attn_fused_wide.hip streams its weights: the 64 lanes of
 * one matrix-operand load then read 1 KB of contiguous memory). Same bytes as mode 0, 16-byte groups permuted (csrc/conv_h3.hip: pack_split_body). */
int wdno_pack_split_weight(const float* w, const float* amax, void* hi, void* lo, float* scale_out, int K, int C, int kd, int kh, int kw,
                           int A, int B, int mode, wdno_stream_t s);
/* Multi-tensor forms of wdno_amax and wdno_pack_split_weight for the per-step refresh of all weights: one launch for any
 * number of items, arguments in a device-resident table (all pointers are device pointers). amax outputs must be zeroed. */
typedef struct { const void* x; int64_t n; void* out; } wdno_amax_item;
typedef struct {
  const void* w; const void* amax; void* hi; void* lo; void* scale_out;
  int K, C, kd, kh, kw, A, B, mode;
} wdno_wsplit_item;
int wdno_amax_multi(const void* table, int n_items, int blocks_per_item, wdno_stream_t s);
int wdno_pack_split_weight_multi(const void* table, int n_items, int blocks_per_item, wdno_stream_t s);
int wdno_conv_fwd_f16x3(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                        const float* bias, const float* residual, float* y, const wdno_conv_geom* g, wdno_stream_t s);
/* the same, and max|y| over the outputs this launch wrote is merged into amax_rec (an amax record, or NULL) */
int wdno_conv_fwd_f16x3_amax(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                             const float* bias, const float* residual, float* y, float* amax_rec, const wdno_conv_geom* g,
                             wdno_stream_t s);
/* Structural zeros of the activation operand, stated by the caller: x[n][d][h][w][c] == 0 for every c < channels at every pixel with d >= d0
 * or h >= h0 or w >= w0 (the zero padding of the wavelet-coefficient channels that p_losses / the sampler impose, smoke/ddpm/diffusion_2d.py:
 * 1008-1033: 18 x 34 x 34 coefficients in a 24 x 40 x 40 tensor). A hint about DATA: results are those of the plain entry points bit for bit
 * (a skipped term is an exact zero); a kernel that can use it -- the 7 x 7 x 7 stem of the smoke U-Net, video_diffusion_pytorch_conv3d.py:393 --
 * does not run the reduction stages of whole 16-channel blocks below `channels` whose source plane or row is >= d0 / h0 for every pixel of a tile. */
typedef struct { int channels, d0, h0, w0; } wdno_zero_box;
int wdno_conv_fwd_f16x3_zbox(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                             const float* bias, const float* residual, float* y, float* amax_rec, const wdno_conv_geom* g,
                             const wdno_zero_box* zb, wdno_stream_t s);
/* single-product (bf16) form with both options: zb may be NULL; y_bf16 != 0: y is bf16 storage [rows][K] (round-to-nearest-even of the fp32
 * result; residual must be NULL) -- mixed-precision activation storage between a convolution and the GroupNorm that is its only reader
 * (burgers/ddpm_burgers/train_diffusion.py:61-62: accelerate mixed precision), read by the wdno_groupnorm_*_t entry points. */
int wdno_conv_fwd_bf16_ex(const void* x16, const void* wp16, const float* bias, const float* residual, void* y, int y_bf16, float* amax_rec,
                          const wdno_conv_geom* g, const wdno_zero_box* zb, wdno_stream_t s);
/* the same with a caller-lent workspace: layers of few pixels x many channels (the 8 x 8 / 16 x 16 levels of the Burgers U-Net, unet.py:150-181)
   cut the reduction of a tile into four runs of stages, one block each; the runs' partial sums go through `ws` and are added in a fixed order.
   wdno_conv_fwd_split_ws_bytes(g): bytes such a geometry wants (0 = it never splits; ws may then be NULL). */
int wdno_conv_fwd_f16x3_ws(const void* xh, const void* xl, const float* sx, const void* wph, const void* wpl, const float* sw,
                           const float* bias, const float* residual, float* y, float* amax_rec, const wdno_conv_geom* g,
                           void* ws, size_t ws_bytes, wdno_stream_t s);
size_t wdno_conv_fwd_split_ws_bytes(const wdno_conv_geom* g);
/* weight gradient on the same split planes (g->C = C8 of x, g->K = K8 of dy); dwp [kd][kh][K8][kw*C8] fp32 */
size_t wdno_conv_wgrad_f16x3_ws_bytes(const wdno_conv_geom* g);
/* pixel_table: [N*OD*OH*OW] 16-byte records from wdno_conv_pixel_table (depends on the geometry only; callers cache it) */
int wdno_conv_pixel_table(void* table, const wdno_conv_geom* g, wdno_stream_t s);
int wdno_conv_wgrad_f16x3(const void* xh, const void* xl, const float* sx, const void* dyh, const void* dyl, const float* sdy,
                          const void* pixel_table, float* dwp, void* ws, size_t ws_bytes, const wdno_conv_geom* g, wdno_stream_t s);
/* Same, with the result in the layout of the parameter: dw[Kn][Cn][kd][kh][kw] (Kn <= g->K, Cn <= g->C; channel padding dropped),
 * written by the split reduction itself. ws >= wdno_conv_wgrad_f16x3_ws_bytes(g) always (also when there is one split). */
int wdno_conv_wgrad_f16x3_param(const void* xh, const void* xl, const float* sx, const void* dyh, const void* dyl, const float* sdy,
                                const void* pixel_table, float* dw, int Kn, int Cn, void* ws, size_t ws_bytes,
                                const wdno_conv_geom* g, wdno_stream_t s);
/* The same weight gradient in two steps, for callers that collect the split reductions of a whole backward pass (autograd through every
 * nn.Conv*d of the U-Nets, unet.py:129-259 / conv3d.py:189-230: only the optimiser reads a weight gradient): wdno_conv_wgrad_partials runs the
 * partial-sum kernels into ws ([splits][kd*kh][K8][kw*C8]) and fills `item`; wdno_wgrad_reduce_multi then performs the ordered reductions of any
 * number of items in one launch (items travel by value in the kernel arguments: no table upload, capturable in a HIP graph) -- the same additions
 * in the same order as wdno_conv_wgrad_f16x3_param / _bf16_param, i.e. bit-identical results. ws must stay alive and unmodified in between.
 * xl == NULL (and sx == sdy == NULL): single bf16 planes. `items` is a HOST array. */
#define WDNO_WGRAD_REDUCE_MAX 40
typedef struct wdno_wgrad_reduce_item {
  const float* ws; float* dw;      /* partial sums [splits][n]; destination in the parameter's layout dw[Kn][Cn][kd][kh][kw] */
  int n, splits;                   /* n = kd*kh*K8*kw*C8 */
  int K8, C8, kw, ntap, Kn, Cn;
  int tiled;                       /* which of the two reduction bodies (chosen by the library) */
  int reserved;
} wdno_wgrad_reduce_item;
int wdno_conv_wgrad_partials(const void* xh, const void* xl, const float* sx, const void* dyh, const void* dyl, const float* sdy,
                             const void* pixel_table, float* dw, int Kn, int Cn, void* ws, size_t ws_bytes,
                             const wdno_conv_geom* g, wdno_wgrad_reduce_item* item, wdno_stream_t s);
int wdno_wgrad_reduce_multi(const wdno_wgrad_reduce_item* items, int n_items, wdno_stream_t s);
/* dwp[kd][kh][K][kw*C] = sum over output pixels of dy (x) shifted x. ws: caller workspace. */
size_t wdno_conv_wgrad_ws_bytes(const wdno_conv_geom* g);
int wdno_conv_wgrad(const float* x, const float* dy, float* dwp, void* ws, size_t ws_bytes,
                    const wdno_conv_geom* g, wdno_stream_t s);
/* The sums over rows that END a backward pass -- bias gradients from the column-sum partials of a dy sweep, GroupNorm's d(gamma) / d(beta) from the
 * per-sample pieces -- feed parameters only (nn.Conv*d.bias, nn.GroupNorm.weight / .bias: unet.py:129-181, conv3d.py:189-204), so a caller may
 * collect them and run ONE launch when the backward has returned: out[j] = sum_r part[r * stride + col0 + j], j < ncols, fp64 accumulation in the
 * fixed order of the single-tensor launches they replace (partial_rows_sum: bit-identical). `items` is a HOST array (they travel by value in the
 * kernel arguments); the partial matrices must stay alive and unmodified until the launch has run. */
#define WDNO_ROWS_SUM_MAX 64
typedef struct wdno_rows_sum_item {
  const void* part; float* out;
  int rows, stride, col0, ncols;
  int is_double, reserved;
} wdno_rows_sum_item;
int wdno_rows_sum_multi(const wdno_rows_sum_item* items, int n_items, wdno_stream_t s);
/* out[C] = sum_p in[p][C]  (bias gradients and other per-channel reductions). ws >= wdno_colsum_ws_bytes. */
size_t wdno_colsum_ws_bytes(int64_t P, int C);
int wdno_colsum(const float* in, float* out, int64_t P, int C, void* ws, size_t ws_bytes, wdno_stream_t s);

/* ---- single-product bf16 convolutions (BASELINE.json configs[1]: "bf16 on 1 x MI355X, batch 256"; the reference's knob is
 * Trainer(amp=True, mixed_precision_type=...) burgers/ddpm_burgers/train_diffusion.py:62,71-74). Activations, weights and output
 * gradients are rounded to ONE bf16 plane [rows][C8] (round-to-nearest-even, no scale) and multiplied on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulators; master weights, bias, residual, outputs and all reductions stay fp32.
 * Same geometry contract and packed-weight layout as the f16x3 entry points (pack with wdno_pack_split_weight and lo == NULL,
 * amax == NULL, scale_out == NULL: one bf16 plane in `hi`; wdno_wsplit_item likewise). Accuracy is bf16-class (2^-8 per
 * operand): tests/test_gpu_bf16.py documents the tolerance against the oracle; it cannot meet the fp32 path's 1e-5. */
int wdno_cast_bf16(const float* x, void* out16, int64_t rows, int C, int C8, wdno_stream_t s);
int wdno_cast_bf16_colsum(const float* x, void* out16, float* colsum_out, void* ws, size_t ws_bytes, int64_t rows, int C, int C8, wdno_stream_t s);
int wdno_conv_fwd_bf16(const void* x16, const void* wp16, const float* bias, const float* residual, float* y, float* amax_rec,
                       const wdno_conv_geom* g, wdno_stream_t s);
/* dw[Kn][Cn][kd][kh][kw] (the parameter's layout); workspace size = wdno_conv_wgrad_f16x3_ws_bytes(g) */
int wdno_conv_wgrad_bf16_param(const void* x16, const void* dy16, const void* pixel_table, float* dw, int Kn, int Cn, void* ws,
                               size_t ws_bytes, const wdno_conv_geom* g, wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ normalisation
 * GroupNorm (+ optional (scale+1, shift) modulation) + optional SiLU on CL [N, S, C]:
 *   y = act( (GN(x)*gamma + beta) * (ss[n, c] + 1) + ss[n, C + c] ),  ss = [N, 2C] or NULL.
 * burgers/ddpm_burgers/unet.py:139-148 ; smoke/.../video_diffusion_pytorch_conv3d.py:196-204.
 * stats [N, G, 2] (mean, rstd) is written by fwd and read by bwd.
 */
size_t wdno_groupnorm_ws_bytes(int64_t N, int64_t S, int C, int G);
/* Round 3: `stats` of every GroupNorm entry point is a buffer of wdno_groupnorm_stats_floats(N, C, G) floats: the [N][G] (mean, rstd) pairs
 * followed by the per-channel / per-group tables the forward derives from them, which the backward entry points read back (one launch
 * fewer per norm). The forward that wrote it and the backward that reads it must see the same gamma / beta / scale_shift. */
size_t wdno_groupnorm_stats_floats(int64_t N, int C, int G);
int wdno_groupnorm_act_fwd(const float* x, const float* gamma, const float* beta, const float* ss, float* y,
                           float* stats, int64_t N, int64_t S, int C, int G, float eps, int silu,
                           void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_groupnorm_act_fwd_amax(const float* x, const float* gamma, const float* beta, const float* ss, float* y,
                                float* stats, float* amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                void* ws, size_t ws_bytes, wdno_stream_t s);
/* dx, dgamma_beta_partial [N, 2, C] (per-sample; caller sums over N), dss [N, 2C] or NULL */
int wdno_groupnorm_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* ss,
                           const float* stats, float* dx, float* dgb_partial, float* dss,
                           int64_t N, int64_t S, int C, int G, int silu, void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_groupnorm_act_bwd_amax(const float* x, const float* dy, const float* gamma, const float* beta, const float* ss,
                                const float* stats, float* dx, float* dgb_partial, float* dss, float* amax_rec,
                                int64_t N, int64_t S, int C, int G, int silu, void* ws, size_t ws_bytes, wdno_stream_t s);
/* The same backward with dx delivered as the fp16 (hi, lo) planes the split convolution kernels read (dx of the norm is the dy of
 * the convolution in front of it, Block.forward: unet.py:80-101, conv3d.py:186-206) -- no fp32 dx, no separate amax / split passes.
 * The scale comes from an upper bound of max|dx| that the per-sample finalize derives from per-channel maxima of the reduction pass
 * (bound_rec: a zeroed amax record, receives the bound). dx_colsum [C] = column sums of dx (the bias gradient of that convolution).
 * C / 8 must be a power of two <= 256 (WDNO_EUNSUPPORTED otherwise: use wdno_groupnorm_act_bwd_amax + wdno_split_f16). */
size_t wdno_groupnorm_bwd_planes_ws_bytes(int64_t N, int64_t S, int C, int G);
/* Forward with y delivered as planes (the output of block1 of a ResnetBlock, read by block2's convolution only: unet.py:103-118,
 * conv3d.py:208-230); scale from the bound max|a| max|x| + max|b| of the folded per-channel affine. Same channel restriction. */
size_t wdno_groupnorm_fwd_planes_ws_bytes(int64_t N, int64_t S, int C, int G);
int wdno_groupnorm_act_fwd_planes(const float* x, const float* gamma, const float* beta, const float* ss, void* y_hi, void* y_lo,
                                  float* y_scale, float* stats, float* bound_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                  void* ws, size_t ws_bytes, wdno_stream_t s);
/* y = act(GroupNorm(x)) + residual as BOTH the fp32 tensor y and its planes: the tail of a ResnetBlock with an identity skip
 * (unet.py:167-176, conv3d.py:286-300: `h + self.res_conv(x)` with res_conv = Identity), whose sum is the next block's skip and the
 * operand of its first convolution. res_rec = amax record of residual (the scale bound is the norm's bound + max|residual|);
 * y_amax_rec (optional) receives the amax record of y. y_hi == NULL: the fp32 sum only (one pass instead of apply + add) for sums
 * whose reader is not a convolution. Workspace: wdno_groupnorm_fwd_planes_ws_bytes. */
int wdno_groupnorm_act_add_fwd_planes(const float* x, const float* gamma, const float* beta, const float* ss, const float* residual,
                                      const float* res_rec, float* y, void* y_hi, void* y_lo, float* y_scale, float* stats,
                                      float* bound_rec, float* y_amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                      void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_groupnorm_act_bwd_planes(const float* x, const float* dy, const float* gamma, const float* beta, const float* ss,
                                  const float* stats, void* dx_hi, void* dx_lo, float* dx_scale, float* dx_colsum,
                                  float* dgb_partial, float* dgb_sum, float* dss, float* bound_rec, int64_t N, int64_t S, int C, int G, int silu,
                                  void* ws, size_t ws_bytes, wdno_stream_t s);
/* dx_colsum == NULL in the two backward entry points: the column sums of dx are left as partials -- doubles [rows][C] at byte offset csp_offset of ws
 * (this call) -- for wdno_rows_sum_multi; likewise dgb_sum == NULL leaves the per-sample pieces dgb_partial [N][2 C] to be summed there. */
int wdno_groupnorm_bwd_planes_tail(int64_t N, int64_t S, int C, int G, size_t* csp_offset, int* rows);
/* The same four entry points for bf16-stored activations (single-product mode): x_bf16 / dy_bf16 != 0 say that x / dy is bf16 storage -- what
 * wdno_conv_fwd_bf16_ex(y_bf16 = 1) wrote. Statistics, affine, activation and every output are computed exactly as above (fp32 / fp64). */
int wdno_groupnorm_act_fwd_amax_t(const void* x, int x_bf16, const float* gamma, const float* beta, const float* ss, float* y,
                                  float* stats, float* amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                  void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_groupnorm_act_fwd_planes_t(const void* x, int x_bf16, const float* gamma, const float* beta, const float* ss, void* y_hi, void* y_lo,
                                    float* y_scale, float* stats, float* bound_rec, int64_t N, int64_t S, int C, int G, float eps,
                                    int silu, void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_groupnorm_act_add_fwd_planes_t(const void* x, int x_bf16, const float* gamma, const float* beta, const float* ss, const float* residual,
                                        const float* res_rec, float* y, void* y_hi, void* y_lo, float* y_scale, float* stats,
                                        float* bound_rec, float* y_amax_rec, int64_t N, int64_t S, int C, int G, float eps, int silu,
                                        void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_groupnorm_act_bwd_planes_t(const void* x, int x_bf16, const void* dy, int dy_bf16, const float* gamma, const float* beta, const float* ss,
                                    const float* stats, void* dx_hi, void* dx_lo, float* dx_scale, float* dx_colsum,
                                    float* dgb_partial, float* dgb_sum, float* dss, float* bound_rec, int64_t N, int64_t S, int C, int G, int silu,
                                    void* ws, size_t ws_bytes, wdno_stream_t s);
/* dgb_sum (round 3, optional): [2 C] = the sum over the samples of dgb_partial [N][2][C] (d gamma | d beta), produced by the launch that
 * also finishes dx_colsum; NULL = the caller reduces dgb_partial itself. */
/* Channel LayerNorm over C of CL rows [P, C], gain only (unet.py:55-65, conv3d.py:165-174) */
int wdno_layernorm_fwd(const float* x, const float* g, float* y, int64_t P, int C, float eps, wdno_stream_t s);
int wdno_layernorm_fwd_amax(const float* x, const float* g, float* y, float* amax_rec, int64_t P, int C, float eps, wdno_stream_t s);
/* y as fp16 (hi, lo) planes for the to_qkv projection that is its only reader (PreNorm: unet.py:67-78, conv3d.py:176-184); the scale
 * follows from |y| <= sqrt(C) max|g|, no data pass. C must be a multiple of 8. */
int wdno_layernorm_fwd_planes(const float* x, const float* g, void* y_hi, void* y_lo, float* y_scale, int64_t P, int C, float eps,
                              wdno_stream_t s);
size_t wdno_layernorm_bwd_ws_bytes(int64_t P, int C);
int wdno_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* dg, int64_t P, int C,
                       float eps, void* ws, size_t ws_bytes, wdno_stream_t s);
/* the same with dx += add_to (the gradient that reaches x over the skip connection of Residual(PreNorm(fn)), unet.py:18-24,67-78,
 * conv3d.py:131-137,176-184): saves the separate accumulation pass autograd would issue. add_to may be NULL. */
int wdno_layernorm_bwd_add(const float* x, const float* g, const float* dy, const float* add_to, float* dx, float* dg, int64_t P, int C,
                           float eps, void* ws, size_t ws_bytes, wdno_stream_t s);
/* ... and max|dx| left in an amax record (dx is the dy of the projection in front of the next Residual block). amax_rec may be NULL. */
int wdno_layernorm_bwd_add_amax(const float* x, const float* g, const float* dy, const float* add_to, float* dx, float* dg, float* amax_rec,
                                int64_t P, int C, float eps, void* ws, size_t ws_bytes, wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ attention
 * qkv rows are [row][3*heads*32] = (q | k | v), each [heads][32]; outputs are [row][heads*32].
 * Row index of (unit (uo, ui), token j) = uo*so + ui*si + j*st.
 *
 * Softmax attention over <= 1024 tokens (burgers unet.py:240-259 ; smoke conv3d.py:294-353): optional rotary
 * tables rot_cos/rot_sin [n][32] (q and k), optional additive bias [heads][n][n]; q is scaled by `scale`.
 */
typedef struct {
  int n_uo, n_ui, n_tok, heads;
  int64_t so, si, st;
} wdno_attn_desc;
int wdno_attn_fwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out,
                  const wdno_attn_desc* d, float scale, wdno_stream_t s);
/* out = the forward result (saved); dqkv (same layout as qkv); dbias [heads][n][n] (written, not accumulated) or NULL. n_tok <= 128.
 * The relative-position-bias gradient is DETERMINISTIC: every block leaves its partial sum in the workspace (wdno_attn_bwd_ws_bytes,
 * needed only when dbias != NULL) and a second launch adds the partials in block order; with heads == 4 and n_tok <= 32 a wave sums
 * its items in registers in item order (no atomics anywhere). */
size_t wdno_attn_bwd_ws_bytes(const wdno_attn_desc* d);
int wdno_attn_bwd(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                  const float* dout, float* dqkv, float* dbias, const wdno_attn_desc* d, float scale, void* ws, size_t ws_bytes,
                  wdno_stream_t s);
/* the same with an amax record (or NULL) for the tensor written: out / dqkv */
int wdno_attn_fwd_amax(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out, float* amax_rec,
                       const wdno_attn_desc* d, float scale, wdno_stream_t s);
int wdno_attn_bwd_amax(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                       const float* dout, float* dqkv, float* dbias, float* amax_rec, const wdno_attn_desc* d, float scale,
                       void* ws, size_t ws_bytes, wdno_stream_t s);
/* The same with dqkv delivered as fp16 (hi, lo) planes [rows][3*heads*32] for the gradient kernels of the qkv projection (its only
 * reader: conv3d.py:232-300). The scale is derived in the kernel from the amax records of qkv and dout (an upper bound of max|dqkv|,
 * see csrc/attention.hip) and left in dqkv_scale[0]. n_tok <= 32 only (WDNO_EUNSUPPORTED otherwise). */
int wdno_attn_bwd_planes(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, const float* out,
                         const float* dout, void* dqkv_hi, void* dqkv_lo, float* dqkv_scale, float* dbias,
                         const float* rec_qkv, const float* rec_dout, const wdno_attn_desc* d, float scale, void* ws, size_t ws_bytes,
                         wdno_stream_t s);
/* Forward with out delivered ONLY as the fp16 planes of the to_out projection (|out| <= max|qkv|: scale from rec_qkv); `out` is not
 * written. n_tok <= 64 (WDNO_EUNSUPPORTED beyond). The n_tok <= 32 backward (wdno_attn_bwd*, MFMA path) does not read out: delta =
 * sum_j P dP is formed in registers; for 33..64 tokens the backward does read it, so a caller that needs gradients keeps the fp32 form
 * there (the super-resolution sampler, 48 frames, does not). */
int wdno_attn_fwd_planes(const float* qkv, const float* rot_cos, const float* rot_sin, const float* bias, float* out, void* out_hi,
                         void* out_lo, float* out_scale, float* amax_rec, const float* rec_qkv, const wdno_attn_desc* d, float scale,
                         wdno_stream_t s);
/* Linear attention (unet.py:203-223 ; conv3d.py:241-258): q softmax over the 32 head channels, k softmax over
 * tokens, ctx = k^T v, out = ctx^T q * scale. units x n_tok rows, contiguous. ws holds k statistics and ctx. */
size_t wdno_linattn_ws_bytes(int64_t units, int heads);
int wdno_linattn_fwd(const float* qkv, float* out, float* kstats /*[units,heads,32,2]*/, float* ctx /*[units,heads,32,32]*/,
                     int64_t units, int n_tok, int heads, float scale, wdno_stream_t s);
int wdno_linattn_bwd(const float* qkv, const float* dout, const float* kstats, const float* ctx, float* dqkv,
                     void* ws, size_t ws_bytes, int64_t units, int n_tok, int heads, float scale, wdno_stream_t s);
int wdno_linattn_fwd_amax(const float* qkv, float* out, float* kstats, float* ctx, float* amax_rec, int64_t units, int n_tok, int heads,
                          float scale, wdno_stream_t s);
int wdno_linattn_bwd_amax(const float* qkv, const float* dout, const float* kstats, const float* ctx, float* dqkv, float* amax_rec,
                          void* ws, size_t ws_bytes, int64_t units, int n_tok, int heads, float scale, wdno_stream_t s);
/* The same with dqkv delivered as fp16 (hi, lo) planes (see wdno_attn_bwd_planes); the scale bound uses the amax records of qkv and
 * dout and the measured max|dctx| (rec_dctx: a zeroed record the call fills). */
int wdno_linattn_bwd_planes(const float* qkv, const float* dout, const float* kstats, const float* ctx, void* dqkv_hi, void* dqkv_lo,
                            float* dqkv_scale, const float* rec_qkv, const float* rec_dout, float* rec_dctx, void* ws, size_t ws_bytes,
                            int64_t units, int n_tok, int heads, float scale, wdno_stream_t s);
/* Forward with out delivered ONLY as fp16 planes (the backward does not read out); |out| <= scale * max|qkv|. */
int wdno_linattn_fwd_planes(const float* qkv, void* out_hi, void* out_lo, float* out_scale, float* kstats, float* ctx,
                            const float* rec_qkv, int64_t units, int n_tok, int heads, float scale, wdno_stream_t s);

/* The whole temporal-attention block of the smoke U-Net's first level as ONE launch (csrc/attn_fused.hip): replaces
 * Residual(PreNorm(dim, EinopsToAndFrom('b c f h w', 'b (h w) f c', Attention(dim, heads, 32, rotary_emb)))) --
 * smoke/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:165-174 (LayerNorm), :277-353 (Attention.forward: to_qkv, scale,
 * rotary, + pos_bias, softmax, to_out), :176-183 / :259-275 (PreNorm, the einops wrapper), Residual's add.
 *   x, y: CL [n_batch, n_tok, hw, C] fp32 (frames before pixels; y = x + to_out(attention(LayerNorm(x)))); gamma [C];
 *   wq_* / wo_*: the packed forward operands (fp16 hi / lo planes + scale) of to_qkv [3*heads*32][C] and to_out [C][heads*32] as
 *   wdno_pack_split_weight writes them; rot_cos / rot_sin [n_tok][32] or NULL; bias [heads][n_tok][n_tok] or NULL;
 *   amax_rec: optional amax record of y; qkv_out: optional [rows][3*heads*32] raw projections for a backward pass that wants them;
 *   rec_v: optional zeroed amax record that receives max|v| (wdno_tattn_fused_bwd: the plane scale of the attention output).
 * wdno_tattn_fused_takes: 1 for the shapes the kernels are built for (C = 64, 4 heads, 24 frames -- the base-resolution model -- or 48 --
 * the super-resolution model of inference_2d.py; 48 frames forward only: qkv_out must be NULL and wdno_tattn_fused_bwd refuses them; C = 128 /
 * 256 with 24 frames -- the deeper levels, csrc/attn_fused_wide.hip -- forward only, qkv_out must be NULL, rec_v is left untouched, and
 * wq_* / wo_* are the operands of pack modes 10 / 11), else 0 (callers then run the block layer by layer). */
int wdno_tattn_fused_takes(int C, int n_tok, int heads);
int wdno_tattn_fused_fwd(const float* x, const float* gamma, float eps, const void* wq_hi, const void* wq_lo, const float* wq_scale,
                         const void* wo_hi, const void* wo_lo, const float* wo_scale, const float* rot_cos, const float* rot_sin,
                         const float* bias, float* y, float* amax_rec, float* qkv_out, float* rec_v,
                         int64_t n_batch, int n_tok, int64_t hw, int C, int heads, float scale, wdno_stream_t s);
/* Backward of the same block as ONE launch + an ordered reduction (csrc/attn_fused_bwd.hip): the backward of conv3d.py:165-174 and
 * :277-353 through autograd in the reference. Only x is needed from the forward: LayerNorm, projections, scores and attention output
 * are recomputed per sequence.
 *   wq_*: the packed FORWARD operand of to_qkv (as in the forward), wo_*: the packed DATA-GRADIENT operand of to_out (W_out^T
 *   [heads*32][C], wdno_pack_split_weight mode 1); dy: gradient of y, CL like x; rec_dy: its amax record; rec_v: the record the forward
 *   launch filled;
 *   dx: gradient of x (the residual path included), amax_rec: optional amax record of dx;
 *   grads: wdno_tattn_fused_bwd_grads() floats = [ dW_qkv [3*heads*32][C] | dW_out [C][heads*32] | dgamma [C] | dbias [heads][n_tok][n_tok] ],
 *   each the sum over the launch's blocks of per-block partials in `ws` (wdno_tattn_fused_bwd_ws_bytes() bytes) taken in block order:
 *   two launches on the same inputs return the same bits. */
size_t wdno_tattn_fused_bwd_ws_bytes(void);
int wdno_tattn_fused_bwd_grads(void);
int wdno_tattn_fused_bwd(const float* x, const float* dy, const float* gamma, float eps, const void* wq_hi, const void* wq_lo,
                         const float* wq_scale, const void* wo_hi, const void* wo_lo, const float* wo_scale, const float* rot_cos,
                         const float* rot_sin, const float* bias, const float* rec_dy, const float* rec_v, float* dx, float* amax_rec,
                         float* grads, void* ws, size_t ws_bytes,
                         int64_t n_batch, int n_tok, int64_t hw, int C, int heads, float scale, wdno_stream_t s);

/* The SpatialLinearAttention block of the 64-channel levels FORWARD, for passes that need no gradient (csrc/linattn_fused.hip): replaces
 * Residual(PreNorm(dim, SpatialLinearAttention(dim, heads))) -- smoke/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:165-174
 * (LayerNorm), :232-258 (to_qkv, softmax over features / tokens, context, to_out) and Residual's add -- by two passes over the tokens and a
 * merge; the [rows][384] projections are never written.
 *   x, y: CL [units][n_tok][C] fp32 (units = batch * frames, n_tok = H * W); gamma [C]; wq_* / wo_*: packed forward operands of to_qkv
 *   [3*heads*32][C] and to_out [C][heads*32]; bias_out [C] or NULL; amax_rec: optional amax record of y; ws: wdno_lattn_fused_ws_bytes bytes;
 *   ctx_out [units][heads][32][32] / kstat_out [units][heads][2][32] (max over the tokens of k, 1 / Z): optional, what the backward call reads.
 * wdno_lattn_fused_takes: 1 for C = 64, 4 heads, n_tok >= 32. */
int wdno_lattn_fused_takes(int C, int heads, int n_tok);
size_t wdno_lattn_fused_ws_bytes(int64_t units, int n_tok);
int wdno_lattn_fused_fwd(const float* x, const float* gamma, float eps, const void* wq_hi, const void* wq_lo, const float* wq_scale,
                         const void* wo_hi, const void* wo_lo, const float* wo_scale, const float* bias_out, float* y, float* amax_rec,
                         float* ctx_out, float* kstat_out, void* ws, size_t ws_bytes, int64_t units, int n_tok, int C, int heads, float scale,
                         wdno_stream_t s);

/* Backward of the same block (csrc/linattn_fused_bwd.hip): one reduction pass over the tokens (dctx), a merge, one pass that recomputes the
 * projections per token tile and writes dx, and the ordered sum of the per-block weight-gradient partials. Needs x, the ctx / kstat tensors the
 * forward call left, dy with its amax record, the packed forward operand of to_qkv and the packed DATA-GRADIENT operand of to_out (W_out^T).
 *   grads: wdno_lattn_fused_bwd_grads() floats = [ dW_qkv [3*heads*32][C] | dW_out [C][heads*32] | dgamma [C] | db_out [C] ]. */
int wdno_lattn_fused_bwd_grads(void);
size_t wdno_lattn_fused_bwd_ws_bytes(int64_t units, int n_tok);
int wdno_lattn_fused_bwd(const float* x, const float* dy, const float* gamma, float eps, const void* wq_hi, const void* wq_lo,
                         const float* wq_scale, const void* wot_hi, const void* wot_lo, const float* wot_scale, const float* ctx,
                         const float* kstat, const float* rec_dy, float* dx, float* amax_rec, float* grads, void* ws, size_t ws_bytes,
                         int64_t units, int n_tok, int C, int heads, float scale, wdno_stream_t s);

/* relative-position bias of the temporal attention (conv3d.py:74-112): bias[h][i][j] = W[bucket[i][j]][h] with W [num_buckets, heads]
 * (nn.Embedding weight) and bucket [n, n] int64 (host-built integer table); and dW from d(bias). One launch each. */
int wdno_relpos_bias_fwd(const float* w, const int64_t* bucket, float* out, int n, int heads, wdno_stream_t s);
int wdno_relpos_bias_bwd(const float* dbias, const int64_t* bucket, float* dw, int n, int heads, int num_buckets, wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ nn.Linear on a few rows
 * (time-embedding MLPs; conv3d.py:118-133, 286-296; unet.py:151-165). P <= 1024 rows, C % 4 == 0, strides in floats. The weight
 * is the reference's own [K][C] tensor (row stride w_stride), no packing. y [P][Kp] is zero in columns K..Kp-1. The data
 * gradient is the same call on the transposed weight. dw [K][C] contiguous; db [K] or NULL. */
int wdno_linear_rows_fwd(const float* x, int x_stride, const float* w, int w_stride, const float* bias, float* y, int P, int C, int K,
                         int Kp, wdno_stream_t s);
int wdno_linear_rows_wgrad(const float* x, int x_stride, const float* dy, int dy_stride, float* dw, float* db, int P, int C, int K,
                           wdno_stream_t s);

/* All projections of ONE input x [P][C] (contiguous) in one launch: the scale/shift projections of every ResnetBlock read the same
 * activated time embedding (conv3d.py:118-133 `self.mlp`, unet.py:151-165). The layers are a device-resident table; k_start = number of
 * output features of the layers before this one, F = their total. Outputs are concatenations:
 *   y  : block i = [P][K_i] at float offset P * k_start_i          dy: the same layout
 *   dw : [F][C] (rows k_start_i .. + K_i = the gradient of weight i),  db: [F]
 *   dx : [P][C] = sum over all layers; ws of wdno_linear_multi_dgrad_ws_bytes(F, P, C) bytes (partial sums, added in a fixed order). */
typedef struct wdno_linear_item {
  const void* w;        /* [K][C] fp32, contiguous */
  const void* bias;     /* [K] or NULL */
  int K, k_start;
} wdno_linear_item;
int wdno_linear_multi_fwd(const void* table, int n_items, int F, const float* x, float* y, int P, int C, wdno_stream_t s);
int wdno_linear_multi_wgrad(const void* table, int n_items, int F, const float* x, const float* dy, float* dw, float* db, int P, int C,
                            wdno_stream_t s);
size_t wdno_linear_multi_dgrad_ws_bytes(int F, int P, int C);
int wdno_linear_multi_dgrad(const void* table, int n_items, int F, const float* dy, float* dx, int P, int C, void* ws, size_t ws_bytes,
                            wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ pointwise
 * act: 0 = SiLU, 1 = GELU(erf). */
int wdno_act_fwd(const float* x, float* y, int64_t n, int act, wdno_stream_t s);
int wdno_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int act, wdno_stream_t s);
int wdno_add(const float* a, const float* b, float* out, int64_t n, wdno_stream_t s);
int wdno_add_amax(const float* a, const float* b, float* out, float* amax_rec, int64_t n, wdno_stream_t s);
/* out[b, :] = cat(sin(t_b f_k), cos(t_b f_k)); freqs[dim/2] = exp(-k ln(theta)/(dim/2-1)) is a device table built by
 * the caller with the reference's own host arithmetic (unet.py:88-96, conv3d.py:144-151) */
int wdno_sinusoidal_emb(const int64_t* t, const float* freqs, float* out, int B, int dim, wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ diffusion operator
 * All tensors in the reference's API layout: smoke [B, F, C, H, W]; Burgers [B, C, H, W] (F = 1).
 * Conditioning predicate descriptor (diffusion_2d.py:1008-1033 ; diffusion_1d.py:276-288, order pad,u0,uT,f,low):
 */
typedef struct {
  int tree;              /* 0 = smoke (wavelet), 1 = Burgers (wavelet) */
  int B, F, C, H, W;
  int cT, cH, cW;        /* padded_shape / coef_shape: valid coefficient extent (smoke: T',H',W'; Burgers: H',W') */
  int cond_pad, cond_a, cond_b, cond_c, cond_low;
  /* smoke  : cond_a = is_condition_control (channels 24:40), init-density channel C-2 is always conditioned;
   * Burgers: cond_a = u0, cond_b = uT, cond_c = f ; u0 rows [0, u_rows), uT rows [H-uT_rows, H) */
  int u_rows, uT_rows;
} wdno_cond_desc;
/* training: x = sqrt_ac[t_b]*x0 + sqrt_1mac[t_b]*noise, then conditions imposed on x (clean values from x0) and the
 * same regions of the target zeroed. x_out and target_out may not alias the inputs. */
int wdno_q_sample_cond(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                       float* x_out, float* target_out, const wdno_cond_desc* c, wdno_stream_t s);
/* sampling: impose the conditions in place; `src` holds the clean values at conditioned positions (same shape as x) */
int wdno_apply_cond(float* x, const float* src, const wdno_cond_desc* c, wdno_stream_t s);
/* loss = sum_e (out-target)^2 * wc[c(e)] * wb[b(e)] * inv_count ; grad = dloss/dout (written if grad != NULL).
 * (diffusion_1d.py:640-645 ; diffusion_2d.py:1045-1050). loss is a device scalar. per_sample = F*C*H*W, inner = H*W; channel of element e = (e / inner) % C. */
int wdno_weighted_mse(const float* out, const float* target, const float* wc, const float* wb, float inv_count,
                      float* loss, float* grad, int64_t B, int64_t per_sample, int C, int64_t inner,
                      void* ws, size_t ws_bytes, wdno_stream_t s);
size_t wdno_weighted_mse_ws_bytes(int64_t n);
/* grad = d(loss)/d(out) * gscale[0] (gscale: device scalar = upstream gradient, NULL -> 1) */
int wdno_weighted_mse_bwd(const float* out, const float* target, const float* wc, const float* wb, float inv_count,
                          const float* gscale, float* grad, int64_t B, int64_t per_sample, int C, int64_t inner, wdno_stream_t s);
/* coefficient tables are the registered buffers [T]; t is per-sample [B].
 * p_sample (diffusion_1d.py:242-258 ; diffusion_2d.py:757-785): x_start = clamp(c1 x - c2 eps), mean = m1 x_start + m2 x,
 * x_next = mean + exp(0.5 logvar) * noise  (noise == NULL -> t == 0 step). */
int wdno_p_sample_update(const float* x, const float* eps, const float* noise, const int64_t* t,
                         const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, const float* pm1, const float* pm2,
                         const float* plogvar, float* x_next, float* x_start, int64_t B, int64_t per_sample, int clamp, wdno_stream_t s);
/* DDIM (diffusion_1d.py:419-435 ; diffusion_2d.py:894-911): x_start = clamp(c1 x - c2 eps); eps' = (c1 x - x_start)/c2;
 * x_next = x_start*sqrt_an + c*eps' + sigma*noise. last step (noise == NULL): x_next = x_start. */
int wdno_ddim_update(const float* x, const float* eps, const float* noise, const int64_t* t,
                     const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, float sqrt_an, float c, float sigma,
                     float* x_next, float* x_start, int64_t B, int64_t per_sample, wdno_stream_t s);
/* the same update with (sqrt_an, c, sigma) read from three device floats: every step of a sampling loop is then the SAME launch,
 * which is what lets one captured HIP graph be replayed for the whole loop (diffusion_1d.py:376-460, diffusion_2d.py:851-933) */
int wdno_ddim_update_dev(const float* x, const float* eps, const float* noise, const int64_t* t,
                         const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, const float* coef_dev,
                         float* x_next, float* x_start, int64_t B, int64_t per_sample, wdno_stream_t s);

/* ------------------------------------------------------------------------------------------------ trainer step
 * Flat-buffer optimiser (train_diffusion.py:117,211-216 ; diffusion_2d.py:1159,1283-1291).
 * wdno_sumsq: out[0] (+)= sum g^2 (double accumulate, device scalar float).
 * wdno_adam_clip_step: g *= min(1, max_norm/(sqrt(sumsq)+1e-6)) then torch.optim.Adam update. */
size_t wdno_sumsq_ws_bytes(int64_t n);
int wdno_sumsq(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, wdno_stream_t s);
int wdno_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                        float grad_scale, float lr, float beta1, float beta2, float eps, int step, wdno_stream_t s);
/* Gradient gather: one launch copies every per-parameter gradient tensor into its span of the flat gradient buffer (items with
 * src == NULL are zero-filled): what DistributedDataParallel's bucket copies / AccumulateGrad's adds do one tensor at a time. The
 * table of n_items wdno_copy_item lives in device memory. */
typedef struct { const void* src; void* dst; int64_t n; } wdno_copy_item;
int wdno_gather_items(const void* table, int n_items, int blocks_per_item, wdno_stream_t s);
/* ema = ema*beta + p*(1-beta)  (ema_pytorch lerp) */
int wdno_ema_update(float* ema, const float* p, int64_t n, float beta, wdno_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* WDNO_HIP_H */
