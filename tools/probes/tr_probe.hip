// probe of ds_read_b64_tr_b16 semantics on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x;
  int byte_off = 0;
  if (mode == 0) byte_off = 0;
  else if (mode == 1) byte_off = l * 8;
  else if (mode == 2) byte_off = (l & 15) * 2 + (l >> 4) * 128;
  else if (mode == 3) byte_off = (l & 15) * 32;            // each lane points at a different 16-half row
  else if (mode == 4) byte_off = (l & 3) * 8 + (l >> 2) * 32;
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)((__attribute__((address_space(3))) char*)lds + byte_off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 5; ++mode) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  l%02d: %4d %4d %4d %4d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
