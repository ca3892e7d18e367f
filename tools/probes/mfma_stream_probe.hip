// mfma_stream_probe.hip -- minimal reproduction ATTEMPT of the hand-over finding of csrc/attn_fused.h (it does NOT reproduce: 0 of 20 launches
// differ in every configuration on MI355X, also with the exact-fp32 products of the real kernel behind the projection and three blocks per CU --
// the finding needs the register pressure of the real kernel, 210 registers with matrix-operand registers re-used as load destinations right
// behind the matrix instructions; tools/probes/lattn_wide_repro.py runs the real kernels): a wave multiplies token planes read from
// LDS (ds_read_b128, the A operand) with weight fragments streamed from global memory (the B operand) on v_mfma_f32_32x32x16_f16, three products per
// fragment pair, two fragment sets in flight alternately (the loads of set B issued before the matrix instructions of set A, handed over by the
// waits the compiler derives). Every launch computes the same thing; the program launches it REPS times and counts launches whose output differs
// from the first, once with two blocks per CU and once with one (64 KB of dummy LDS).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_stream_probe tools/probes/mfma_stream_probe.hip && ./mfma_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define C 256
#define AST (C + 8)
#define NP (C / 32)

template <int MODE>      // 0: overlapped hand-over (compiler waits); 1: serial (full wait before the matrix instructions); +2: 64 KB of dummy LDS
__global__ __launch_bounds__(256, 2) void probe_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl, const _Float16* __restrict__ wh,
                                                        const _Float16* __restrict__ wl, float* __restrict__ out, int tiles) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[32 * AST];
  __shared__ __attribute__((aligned(16))) _Float16 Al[32 * AST];
  __shared__ float Dummy[(MODE & 2) ? 16384 : 1];
  if ((MODE & 2) && tiles < 0) Dummy[threadIdx.x] = 1.f;
  const int tid = threadIdx.x, h = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
  const unsigned woff = (unsigned)(h * 3 * NP * 2 * 64 + lane) * 8u;
  f32x16 ak, av, ctx;
  for (int e = 0; e < 16; ++e) ctx[e] = 0.f;
  for (int tile = 0; tile < tiles; ++tile) {
    for (int e = 0; e < 16; ++e) { ak[e] = 0.f; av[e] = 0.f; }
    // planes of this tile: a different 32-row window of the source for every (block, tile)
    const int r0 = (int)((blockIdx.x * 7 + tile * 13) % 64);
    for (int i = tid; i < 32 * (C / 8); i += 256) {
      const int r = i / (C / 8), c8 = i - r * (C / 8);
      *reinterpret_cast<half8*>(Ah + r * AST + 8 * c8) = *reinterpret_cast<const half8*>(xh + (r0 + r) * C + 8 * c8);
      *reinterpret_cast<half8*>(Al + r * AST + 8 * c8) = *reinterpret_cast<const half8*>(xl + (r0 + r) * C + 8 * c8);
    }
    __syncthreads();
    half8 w0[2][2][2], w1[2][2][2];
    auto wload = [&](half8 (&w)[2][2][2], int t) {
#pragma unroll
      for (int kv = 0; kv < 2; ++kv) {
        const unsigned o = woff + (unsigned)((kv + 1) * NP + t) * 1024u;
        w[kv][0][0] = *reinterpret_cast<const half8*>(wh + o); w[kv][0][1] = *reinterpret_cast<const half8*>(wh + o + 512);
        w[kv][1][0] = *reinterpret_cast<const half8*>(wl + o); w[kv][1][1] = *reinterpret_cast<const half8*>(wl + o + 512);
      }
    };
    auto mm3 = [&](half8 a_h, half8 a_l, half8 b_h, half8 b_l, f32x16 c) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_l, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, b_h, c, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_h, c, 0, 0, 0);
    };
    auto wmma = [&](const half8 (&w)[2][2][2], int t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const half8 ah = *reinterpret_cast<const half8*>(Ah + li * AST + 32 * t + 16 * hh + 8 * s);
        const half8 al = *reinterpret_cast<const half8*>(Al + li * AST + 32 * t + 16 * hh + 8 * s);
        ak = mm3(ah, al, w[0][0][s], w[0][1][s], ak);
        av = mm3(ah, al, w[1][0][s], w[1][1][s], av);
      }
    };
    if (MODE & 1) {
#pragma unroll 1
      for (int t = 0; t < NP; t += 2) {
        wload(w0, t);
        wload(w1, t + 1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0[0][0][0]), "+v"(w0[0][0][1]), "+v"(w0[0][1][0]), "+v"(w0[0][1][1]), "+v"(w0[1][0][0]), "+v"(w0[1][0][1]),
                     "+v"(w0[1][1][0]), "+v"(w0[1][1][1]) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w1[0][0][0]), "+v"(w1[0][0][1]), "+v"(w1[0][1][0]), "+v"(w1[0][1][1]), "+v"(w1[1][0][0]), "+v"(w1[1][0][1]),
                     "+v"(w1[1][1][0]), "+v"(w1[1][1][1]) :: "memory");
        wmma(w0, t);
        wmma(w1, t + 1);
        asm volatile("" : "+v"(ak), "+v"(av) :: "memory");
      }
    } else {
      wload(w0, 0);
#pragma unroll 1
      for (int t = 0; t < NP; t += 2) {
        wload(w1, t + 1);
        wmma(w0, t);
        if (t + 2 < NP) wload(w0, t + 2);
        wmma(w1, t + 1);
      }
    }
    __syncthreads();
    // the long exact-fp32 products of the real kernel (64 cycles each): what the other block's waves on this SIMD run into
#pragma unroll
    for (int e = 0; e < 16; ++e) { ak[e] *= 1e-9f; av[e] *= 1e-9f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[r], av[r], ctx, 0, 0, 0);
  }
  float* o = out + ((size_t)blockIdx.x * 256 + tid) * 32;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o[e] = ak[e]; o[16 + e] = ctx[e]; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>
static int run(const char* name, int grid, int reps, const _Float16* xh, const _Float16* xl, const _Float16* wh, const _Float16* wl, float* out, size_t out_floats) {
  std::vector<float> first(out_floats), cur(out_floats);
  int differ = 0;
  size_t worst = 0;
  for (int r = 0; r < reps; ++r) {
    probe_kernel<MODE><<<grid, 256>>>(xh, xl, wh, wl, out, 6);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    if (hipMemcpy(r ? cur.data() : first.data(), out, out_floats * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (r) {
      size_t nd = 0;
      for (size_t i = 0; i < out_floats; ++i) nd += memcmp(&first[i], &cur[i], 4) != 0;
      differ += nd != 0;
      if (nd > worst) worst = nd;
    }
  }
  printf("%-44s grid %4d: %2d of %d launches differ from the first (most differing values in one launch: %zu of %zu)\n", name, grid, differ, reps - 1, worst, out_floats);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t nx = 96 * C, nw = 384 * C;
  std::vector<_Float16> hx(nx), lx(nx), hw(nw), lw(nw);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (size_t i = 0; i < nx; ++i) { float v = rnd() * 16000.f; hx[i] = (_Float16)v; lx[i] = (_Float16)(v - (float)hx[i]); }
  for (size_t i = 0; i < nw; ++i) { float v = rnd() * 16000.f; hw[i] = (_Float16)v; lw[i] = (_Float16)(v - (float)hw[i]); }
  _Float16 *xh, *xl, *wh, *wl;
  float* out;
  const int grid = 3 * cus;
  const size_t out_floats = (size_t)grid * 256 * 32;
  CK(hipMalloc(&xh, nx * 2)); CK(hipMalloc(&xl, nx * 2)); CK(hipMalloc(&wh, nw * 2)); CK(hipMalloc(&wl, nw * 2)); CK(hipMalloc(&out, out_floats * 4));
  CK(hipMemcpy(xh, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(xl, lx.data(), nx * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(wh, hw.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(wl, lw.data(), nw * 2, hipMemcpyHostToDevice));
  printf("%s, %d CUs\n", prop.name, cus);
  const int reps = 21;
  if (run<0>("overlapped hand-over, two blocks per CU", grid, reps, xh, xl, wh, wl, out, out_floats)) return 1;
  if (run<2>("overlapped hand-over, one block per CU", grid, reps, xh, xl, wh, wl, out, out_floats)) return 1;
  if (run<1>("serial hand-over, two blocks per CU", grid, reps, xh, xl, wh, wl, out, out_floats)) return 1;
  if (run<0>("overlapped hand-over, grid = CUs", cus, reps, xh, xl, wh, wl, out, out_floats)) return 1;
  if (run<0>("overlapped hand-over, grid = 2 CUs", 2 * cus, reps, xh, xl, wh, wl, out, out_floats)) return 1;
  return 0;
}
