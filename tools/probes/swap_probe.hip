// v_permlane32_swap semantics probe (gfx950): which lanes of which operand are exchanged?
// hipcc --offload-arch=gfx950 -O3 swap_probe.hip -o swap_probe && ./swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
  float a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  o[threadIdx.x] = __uint_as_float(r[0]); o[64 + threadIdx.x] = __uint_as_float(r[1]);
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("first result : lane0 %g lane31 %g lane32 %g lane63 %g\n", h[0], h[31], h[32], h[63]);
  printf("second result: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[64], h[95], h[96], h[127]);
  return 0;
}
