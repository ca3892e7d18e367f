// How expensive is "one atomicMax per block at the end of a streaming kernel" on gfx950, as a function of how many blocks
// there are and how the target words are spread? (amax records, common.h). Each block streams `per_block` float4 of a
// buffer and then issues one atomicMax to rec[(block % slots) * stride_words].
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/atomic_probe.hip -o tools/probes/atomic_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void sweep(const float4* __restrict__ x, int64_t n4, unsigned* rec, int slots, int stride_words, int mode) {
  __shared__ float red[4];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t k0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; k0 < n4; k0 += stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int64_t k = k0 + u * 256; v[u] = k < n4 ? x[k] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (mode == 0) atomicMax(rec + (size_t)(blockIdx.x % slots) * stride_words, __float_as_uint(m));
    else if (mode == 1) reinterpret_cast<float*>(rec)[blockIdx.x] = m;            // plain store of a partial
    // mode 2: nothing
  }
}

int main() {
  const int64_t n = 78643200 / 4;        // 78.6 MB, the [8,24,40,40,64] activation
  float* x; unsigned* rec;
  hipMalloc(&x, n * 4); hipMalloc(&rec, 1 << 22);
  hipMemset(x, 0, n * 4); hipMemset(rec, 0, 1 << 22);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  struct { int grid, slots, stride, mode; } cases[] = {
    {2048, 1, 1, 2}, {2048, 1, 1, 1}, {2048, 1, 1, 0}, {2048, 64, 1, 0}, {2048, 64, 16, 0}, {2048, 64, 64, 0}, {2048, 64, 1024, 0},
    {4096, 1, 1, 2}, {4096, 64, 1, 0}, {4096, 64, 64, 0}, {4096, 64, 1024, 0}, {4096, 1, 1, 1},
    {1024, 1, 1, 2}, {1024, 64, 1, 0}, {1024, 1, 1, 0}, {512, 1, 1, 2}, {512, 64, 1, 0}, {512, 1, 1, 0}, {256, 1, 1, 2}, {256, 1, 1, 0},
  };
  for (auto& c : cases) {
    for (int w = 0; w < 3; ++w) sweep<<<c.grid, 256>>>((const float4*)x, n / 4, rec, c.slots, c.stride, c.mode);
    hipEventRecord(a);
    for (int it = 0; it < 20; ++it) sweep<<<c.grid, 256>>>((const float4*)x, n / 4, rec, c.slots, c.stride, c.mode);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("grid %5d slots %3d stride %5d mode %d : %7.2f us  (%.2f TB/s)\n", c.grid, c.slots, c.stride, c.mode, ms * 50, n * 4 / (ms * 50e-6) / 1e12);
  }
  return 0;
}
