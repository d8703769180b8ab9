// What does the instruction offset of global_load_lds_dwordx4 add to - the global address, the LDS address, or both?
//   hipcc --offload-arch=gfx950 -O2 -o scripts/_abl/lds_dma_offset_probe scripts/lds_dma_offset_probe.hip && scripts/_abl/lds_dma_offset_probe
// One wave copies 1 KB (16 B per lane) from global element offset `voff` with `offset:1024` and M0 = the LDS base; the LDS block (4 KB)
// is then dumped: where the data landed and which global bytes arrived.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* g, unsigned* out) {
  __shared__ unsigned sm[1024];   // 4 KB
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)sm;
  const unsigned voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(g), "s"(lds0) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = sm[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;   // word i holds i: byte offset = 4 i
  unsigned *g, *o;
  hipMalloc(&g, 4096 * 4); hipMalloc(&o, 1024 * 4);
  hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(g, o);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 1024; ++i) if (r[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
  printf("LDS words written: [%d, %d] (byte offset %d); first word holds global word %u (global byte offset %u)\n", first, last, first * 4, r[first], r[first] * 4);
  printf("=> instruction offset added to LDS address: %s; to global address: %s\n", first * 4 == 1024 ? "yes" : "no", r[first] * 4 == 1024 ? "yes" : "no");
  return 0;
}
