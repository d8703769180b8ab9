// Probe for the GEMM inner loop's energy lever: does a 128x128 wave tile (one wave per SIMD, 256 accumulators,
// 0.25 KB of LDS fragment reads per MFMA) sustain more FLOP/s than the shipped 128x64 wave tile (two waves per SIMD, 128
// accumulators, 0.375 KB per MFMA) when BOTH run nothing but `ds_read_b128` + `v_mfma_f32_16x16x32_bf16` on random operands held in
// LDS - the inner loop of gemm256_kernel without its DMA, barriers and epilogue?  The chip lowers its clock under MFMA load
// (MI355X_MICROARCH.md, DVFS give-back), so the answer is read from wall time AND the in-kernel clock
// (d s_memtime / d s_memrealtime x 100 MHz), not from cycle counts.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_abl/mfma_lds_probe scripts/mfma_lds_probe.hip   (build container), then run the binary on the MI355X
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

// MT x NT MFMA tiles (16x16) per wave; one K-step = 32 deep: MT A fragments + NT B fragments read, MT*NT MFMAs issued.
// LDS image: 64 KB of random bf16, fragment reads walk it with the XOR swizzle of the shipped kernel (conflict-free b128 reads).
template <int MT, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters,
                                                                       unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += WAVES * 64) ((uint4*)smem)[i] = src[(blockIdx.x * 131 + i) & 4095];
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane offsets inside a 128-row x 128-B unit (row = tile*16 + fr, 16-B chunk (kh*4 + fq) ^ (row & 7))
  int offA[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) offA[kh] = fr * 128 + (((kh * 4 + fq) ^ sw) * 16);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const char* base = smem + ((it + wave) & 1) * 32768;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[m] = *(const bf16x8*)(base + offA[kh] + (m & 7) * 2048);
#pragma unroll
      for (int n = 0; n < NT; ++n) fb[n] = *(const bf16x8*)(base + 16384 + offA[kh] + (n & 7) * 2048);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[n], fa[m], acc[m][n], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same wave tile built from v_mfma_f32_32x32x16_bf16 (MT x NT tiles of 32x32, K-step 16): identical LDS fragment bytes per FLOP,
// half the MFMA instructions, half the A/B operand register reads, accumulators twice as large per instruction (round 3: is the MFMA
// SHAPE an energy lever?).  Lane layout of a fragment: row = lane & 31, 16-B chunk = 2 * k16 + (lane >> 5).
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MT, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe32_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters,
                                                                         unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += WAVES * 64) ((uint4*)smem)[i] = src[(blockIdx.x * 131 + i) & 4095];
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5, sw = fr & 7;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  int off[4];
#pragma unroll
  for (int k16 = 0; k16 < 4; ++k16) off[k16] = fr * 128 + (((k16 * 2 + fh) ^ sw) * 16);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const char* base = smem + ((it + wave) & 1) * 32768;
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[m] = *(const bf16x8*)(base + off[k16] + (m & 3) * 4096);
#pragma unroll
      for (int n = 0; n < NT; ++n) fb[n] = *(const bf16x8*)(base + 16384 + off[k16] + (n & 3) * 4096);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[n], fa[m], acc[m][n], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[m][n][e];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// 128x128 wave tile, ONE wave per SIMD, software-pipelined by hand: the 16 fragment reads of K-step k+1 are spread between the 64 MFMAs
// of K-step k (one ds_read_b128 behind every four MFMAs, pinned with sched_group_barrier).  Question (round 3): does the lowest
// LDS-bytes-per-MFMA form (0.25 KB) beat the shipped two-waves-per-SIMD form once its issue order is fixed by hand?
__global__ __launch_bounds__(256, 1) void probe_pipe128_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters,
                                                                unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += 256) ((uint4*)smem)[i] = src[(blockIdx.x * 131 + i) & 4095];
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
  f32x4 acc[8][8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  int off[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) off[kh] = fr * 128 + (((kh * 4 + fq) ^ sw) * 16);
  bf16x8 fa[2][8], fb[2][8];
  const char* base0 = smem + (wave & 1) * 32768;
#pragma unroll
  for (int m = 0; m < 8; ++m) { fa[0][m] = *(const bf16x8*)(base0 + off[0] + m * 2048); fb[0][m] = *(const bf16x8*)(base0 + 16384 + off[0] + m * 2048); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int cur = kh, nxt = kh ^ 1;
      const char* nb = smem + ((it + wave + (kh == 1)) & 1) * 32768;   // K-step after this one: other half-tile, or the next buffer
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        fa[nxt][m] = *(const bf16x8*)(nb + off[nxt] + m * 2048);
        fb[nxt][m] = *(const bf16x8*)(nb + 16384 + off[nxt] + m * 2048);
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur][n], fa[cur][m], acc[m][n], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
#pragma unroll
  for (int m = 0; m < 8; ++m) s += (float)fa[0][m][0] + (float)fb[0][m][0];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// register-only forms (no LDS reads inside the loop): what the MFMA pipe alone sustains for each shape
template <int SHAPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe_reg_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters,
                                                                           unsigned long long* __restrict__ stamps) {
  const int tid = threadIdx.x;
  bf16x8 fa[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fa[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + i) & 4095]);
    fb[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + 4 + i) & 4095]);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[(n + kh) & 3], fa[(m + kh) & 3], acc[m][n], 0, 0, 0);
      asm volatile("" : "+v"(fa[0]), "+v"(fb[0]));
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[(n + k16) & 3], fa[(m + k16) & 3], acc[m][n], 0, 0, 0);
      asm volatile("" : "+v"(fa[0]), "+v"(fb[0]));
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[m][n][e];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename K>
void run_generic(const char* name, K kern, int waves, double flops, const uint4* src, float* out, unsigned long long* stamps, int iters) {
  const int blocks = 256;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 65536, 0, src, out, iters, stamps);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 65536, 0, src, out, iters, stamps);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<unsigned long long> h(2 * blocks);
  CHECK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
  double clk = 0;
  for (int b = 0; b < blocks; ++b) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;   // MHz
  clk /= blocks;
  printf("%-60s %7.3f ms  %7.1f TFLOP/s  in-kernel clock %6.0f MHz\n", name, ms, flops * blocks / (ms * 1e-3) / 1e12, clk);
}

template <int MT, int NT, int WAVES>
void run(const char* name, const uint4* src, float* out, unsigned long long* stamps, int iters) {
  const int blocks = 256;
  CHECK(hipFuncSetAttribute((const void*)probe_kernel<MT, NT, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe_kernel<MT, NT, WAVES>), dim3(blocks), dim3(WAVES * 64), 65536, 0, src, out, iters, stamps);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe_kernel<MT, NT, WAVES>), dim3(blocks), dim3(WAVES * 64), 65536, 0, src, out, iters, stamps);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<unsigned long long> h(2 * blocks);
  CHECK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
  double clk = 0;
  for (int b = 0; b < blocks; ++b) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;   // MHz
  clk /= blocks;
  const double flops = 2.0 * 16 * 16 * 32 * (double)MT * NT * 2 * iters * WAVES * blocks;
  const double reads_per_mfma = (double)(MT + NT) / (MT * NT);
  printf("%-44s %7.3f ms  %7.1f TFLOP/s  in-kernel clock %6.0f MHz  LDS fragment KB per MFMA %.3f  (VGPR+AGPR budget %d)\n", name, ms,
         flops / (ms * 1e-3) / 1e12, clk, reads_per_mfma, WAVES == 8 ? 256 : 512);
}

int main() {
  std::vector<uint16_t> h(4096 * 8);
  srand(1);
  for (auto& v : h) {   // random bf16 in [-1, 1): sign, exponent 0x3f.. range, random mantissa
    const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
    uint32_t u;
    memcpy(&u, &f, 4);
    v = (uint16_t)(u >> 16);
  }
  uint4* src;
  float* out;
  unsigned long long* stamps;
  CHECK(hipMalloc((void**)&src, 65536));
  CHECK(hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice));
  CHECK(hipMalloc((void**)&out, sizeof(float) * 256 * 512));
  CHECK(hipMalloc((void**)&stamps, sizeof(unsigned long long) * 512));
  const int iters = 6000;
  for (int rep = 0; rep < 3; ++rep) {
    run<8, 4, 8>("128x64 wave tile, 8 waves (2 per SIMD)", src, out, stamps, iters);
    run<8, 8, 4>("128x128 wave tile, 4 waves (1 per SIMD)", src, out, stamps, iters / 2);
    run<4, 4, 8>("64x64 wave tile, 8 waves", src, out, stamps, iters * 2);
    // MFMA shape (round 3): per workgroup and iteration 8 waves x 128x64x64 either way
    const double fl = 2.0 * 128 * 64 * 64 * 8 * (double)iters;
    run_generic("128x64 wave tile from 32x32x16 MFMAs + LDS reads, 8 waves", probe32_kernel<4, 2, 8>, 8, fl, src, out, stamps, iters);
    run_generic("128x128 wave tile, 1 wave per SIMD, hand-pipelined reads", probe_pipe128_kernel, 4, 2.0 * 128 * 128 * 64 * 4 * (double)(iters / 2), src, out, stamps, iters / 2);
    run_generic("registers only, 16x16x32 (32 MFMAs per K-64), 8 waves", probe_reg_kernel<16, 8>, 8, fl, src, out, stamps, iters);
    run_generic("registers only, 32x32x16 (16 MFMAs per K-64... x2), 8 waves", probe_reg_kernel<32, 8>, 8, fl, src, out, stamps, iters);
  }
  return 0;
}
