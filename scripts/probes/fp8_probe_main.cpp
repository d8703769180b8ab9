// Feasibility probe (BASELINE config 5): the 256x256 phase-interleaved GEMM schedule with v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, unit
// scales) on random bytes.  K is counted in fp8 elements; the kernel sees rows of K bytes as K/2 "bf16 elements" - same bytes, same DMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aigv-assessor_amd/csrc scripts/probes/fp8_probe_main.cpp scripts/probes/gemm256_fp8_probe.hip -o /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels.h"
int main() {
  struct Shape { int M, N, K; };
  const Shape shapes[] = {{8192, 8192, 8192}, {8704, 28672, 4096}, {8192, 4096, 14336}};
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    uint8_t *A, *W; uint16_t* C;
    hipMalloc((void**)&A, (size_t)M * K); hipMalloc((void**)&W, (size_t)N * K); hipMalloc((void**)&C, (size_t)M * N * 2);
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<uint8_t> ha((size_t)M * K), hw((size_t)N * K);
      if (pass == 0) { for (auto& v : ha) v = 0x38; for (auto& v : hw) v = 0x38; }   // e4m3 1.0 x 1.0 -> C = K exactly
      else { srand(1); for (auto& v : ha) { v = rand() & 0x7f; if ((v & 0x7f) >= 0x7e) v = 0x38; v |= (rand() & 1) << 7; } for (auto& v : hw) { v = rand() & 0x7f; if ((v & 0x7f) >= 0x7e) v = 0x30; v |= (rand() & 1) << 7; } }
      hipMemcpy(A, ha.data(), ha.size(), hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size(), hipMemcpyHostToDevice);
      GemmArgs a{};
      a.A = (const bf16_t*)A; a.lda = K / 2; a.W = (const bf16_t*)W; a.ldw = K / 2; a.C = (bf16_t*)C; a.ldc = N; a.M = M; a.N = N; a.K = K / 2;
      for (int i = 0; i < 3; ++i) aigv_launch_gemm256(a, EPI_STORE, 0);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      const int iters = 10;
      for (int i = 0; i < iters; ++i) aigv_launch_gemm256(a, EPI_STORE, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
      uint16_t c0; hipMemcpy(&c0, C + 12345, 2, hipMemcpyDeviceToHost);
      float cf; uint32_t u = (uint32_t)c0 << 16; memcpy(&cf, &u, 4);
      printf("fp8 e4m3 %s  M=%d N=%d K=%d: %8.1f us  %7.1f TFLOP/s   C[12345]=%g%s\n", pass == 0 ? "ones  " : "random", M, N, K, ms * 1e3,
             2.0 * M * N * K / (ms * 1e-3) / 1e12, cf, pass == 0 ? (cf == (float)K ? " (= K: ok)" : " (EXPECTED K)") : "");
    }
    hipFree(A); hipFree(W); hipFree(C);
  }
  return 0;
}
