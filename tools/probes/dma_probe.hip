// LDS-DMA semantics probe (gfx950): buffer_load_dwordx4 ... offen lds
//   * destination = M0 base + instruction offset + lane * 16 ?
//   * lanes whose buffer offset is out of range: zeros written, or LDS left untouched ?
//   * does the instruction `offset:` field shift the LDS destination, the global source, or both ?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/dma_probe.hip -o tools/probes/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int int4v __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* src, unsigned src_bytes, unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned lds[1024];      // 4 KB
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xAAAAAAAAu;
  __syncthreads();
  uint64_t a = reinterpret_cast<uint64_t>(src);
  int4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)src_bytes);
  r.w = 0x00020000;
  int lane = threadIdx.x;
  // per-lane source offsets: reversed order so that source != destination order; lanes 5 and 40 out of range
  int off = (63 - lane) * 16;
  if (lane == 5 || lane == 40) off = 0x7ffffff0;
  unsigned base = (unsigned)(uintptr_t)lds;     // LDS byte address of the array (low 32 bits of the generic->local pointer)
  base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds);
  unsigned m0v = base + 1024;                   // land the piece 1 KB into the array
  if (mode == 0)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" : : "v"(off), "s"(m0v), "s"(r) : "memory");
  else
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen offset:32 lds\n\ts_waitcnt vmcnt(0)" : : "v"(off), "s"(m0v), "s"(r) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  unsigned h[64 * 4 + 64];
  for (int i = 0; i < 64 * 4 + 64; ++i) h[i] = 0x1000u + i;          // word i holds 0x1000 + i
  unsigned *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4096);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, 64 * 16, o, mode);
    unsigned r[1024];
    hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    printf("mode %d (%s)\n", mode, mode ? "offset:32" : "no offset");
    int first = -1, last = -1;
    for (int i = 0; i < 1024; ++i) if (r[i] != 0xAAAAAAAAu) { if (first < 0) first = i; last = i; }
    printf("  words written: first %d last %d (expected 256..511 if dest = m0 + lane*16)\n", first, last);
    for (int l : {0, 1, 5, 40, 62, 63}) {
      int w = 256 + l * 4 + (mode ? 8 : 0);
      printf("  lane %2d -> lds words [%d..]: %08x %08x %08x %08x (source word index if linear: %d)\n", l, w, r[w], r[w + 1], r[w + 2], r[w + 3], (63 - l) * 4);
    }
    if (mode) printf("  words 256..263: %08x %08x ... (untouched = aaaaaaaa)\n", r[256], r[257]);
  }
  return 0;
}
