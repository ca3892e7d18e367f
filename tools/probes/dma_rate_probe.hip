// L2 -> LDS streaming rate of one CU (gfx950): LDS-DMA (buffer_load_dwordx4 ... lds) against register loads + ds_write_b128.
// One 256-thread block per CU re-reads a small (L2-resident) window; every wave issues `depth` 1-KiB requests before waiting.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/dma_rate_probe.hip -o tools/probes/dma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int4v rsrc_of(const void* ptr, unsigned bytes) {
  uint64_t a = reinterpret_cast<uint64_t>(ptr);
  int4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

template <int MODE, int WAVES, int PAT>
__global__ __launch_bounds__(64 * WAVES) void stream(const char* src, unsigned window, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const char* base = src + (size_t)blockIdx.x * window;
  int4v rs = rsrc_of(base, window);
  asm volatile("s_nop 4" : "+s"(rs));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 8192;
  unsigned acc = 0;
  int off = PAT == 0 ? wave * 8192 + lane * 16 : wave * 8192 + (lane >> 2) * 128 + (lane & 3) * 16;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // PAT 1: instruction u takes the first half of 16 lines, the second halves come a whole round (8 KB per wave) later
        // PAT 2: instructions 2v, 2v+1 take the two halves of the same 16 lines
        int o = PAT == 0 ? (off + u * 1024) & (window - 1)
              : PAT == 1 ? (off + (u & 3) * 2048 + (u >> 2) * 64 + (it & 1) * 0) & (window - 1)
                         : (off + (u >> 1) * 2048 + (u & 1) * 64) & (window - 1);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(o), "s"(lds0 + u * 1024), "s"(rs) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    } else {
      int4v v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int o = (off + u * 1024) & (window - 1);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[u]) : "v"(o), "s"(rs) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<int4v*>(smem + wave * 8192 + u * 1024 + lane * 16) = v[u];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= (unsigned)v[u].x;
      }
    }
    off += WAVES * 8192;
  }
  if (MODE != 2) acc = *reinterpret_cast<unsigned*>(smem + wave * 8192 + lane * 16);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int WAVES, int PAT>
static void run(const char* name, const char* src, unsigned window, unsigned* sink) {
  const int iters = 2000, blocks = 256;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipFuncSetAttribute((const void*)stream<MODE, WAVES, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * 8192);
  stream<MODE, WAVES, PAT><<<blocks, 64 * WAVES, WAVES * 8192>>>(src, window, 50, sink);
  (void)hipEventRecord(a);
  stream<MODE, WAVES, PAT><<<blocks, 64 * WAVES, WAVES * 8192>>>(src, window, iters, sink);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double bytes = (double)blocks * WAVES * 8192.0 * iters;
  printf("%-44s waves %d : %8.3f ms  %7.2f TB/s chip  %6.1f B/ns per CU\n", name, WAVES, ms, bytes / ms / 1e9, bytes / blocks / (ms * 1e6));
}

int main() {
  const unsigned window = 64 * 1024;          // per block; 256 blocks -> 16 MB total, stays in L2 / MALL
  char* src; unsigned* sink;
  (void)hipMalloc(&src, (size_t)256 * window); (void)hipMalloc(&sink, 64);
  (void)hipMemset(src, 1, (size_t)256 * window);
  run<0, 4, 0>("LDS-DMA, 1 KiB contiguous per instruction", src, window, sink);
  run<0, 4, 1>("LDS-DMA, half lines (other half 4 instr later)", src, window, sink);
  run<0, 4, 2>("LDS-DMA, half lines (other half next instr)", src, window, sink);
  run<1, 4, 0>("register load + ds_write_b128", src, window, sink);
  run<2, 4, 0>("register load only", src, window, sink);
  run<0, 8, 0>("LDS-DMA, 1 KiB contiguous per instruction", src, window, sink);
  run<0, 8, 1>("LDS-DMA, half lines (other half 4 instr later)", src, window, sink);
  run<0, 8, 2>("LDS-DMA, half lines (other half next instr)", src, window, sink);
  return 0;
}
