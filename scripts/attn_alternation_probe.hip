// Would a role-alternating 8-wave attention kernel (waves w and w + 4 of a SIMD one segment apart, s_barrier between segments) reach segment lengths
// near their ideals now that the fragment reads can be pinned ahead of their MFMAs?  (round 3; the first such kernel - profiles/r3_attn8_negative.txt -
// read 1884 / 1695 cycles for segments whose ideals are 1024 / ~750.)  Synthetic d = 128 tile: MSEG = 16 "P V" + 16 "K Q^T" v_mfma_f32_32x32x16_bf16
// with their A operands read from LDS four MFMAs ahead; VSEG = the softmax's instruction mix on 32 scores per lane (16 v_max3, 32 v_fma, 32 v_exp,
// 32 v_add, 16 v_cvt_pk).  One workgroup of 8 waves per CU.  Modes: 0 alternate (group g runs MSEG when (it + g) is even), 1 in phase (both groups MSEG,
// then both VSEG: the convoy), 2 MSEG only, 3 VSEG only.  Reported: cycles per interval (s_memtime, 100 MHz ticks x clock ratio) and ms.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o scripts/_abl/attn_alternation_probe scripts/attn_alternation_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(const uint4* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  for (int i = tid; i < 65536 / 16; i += 512) ((uint4*)smem)[i] = src[(blockIdx.x * 131 + i) & 4095];
  __syncthreads();
  f32x16 oacc[4], sacc[2];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) sacc[i][e] = 0.001f * (lane + e + i);
  bf16x8 qf[8], pf[4];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) qf[i][e] = (__bf16)(0.01f * (lane + i + e));
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) pf[i][e] = (__bf16)(0.02f * (lane + i - e));
  float m_run = 0.f, l_run = 0.f;
  const int off = lane * 16;                                   // conflict-free: a fragment read is 1 KB contiguous over the wave
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const bool mseg = MODE == 2 ? true : MODE == 3 ? false : MODE == 1 ? (it & 1) == 0 : ((it + grp) & 1) == 0;
    const char* base = smem + ((it >> 1) & 1) * 32768;
    if (mseg) {
      // 32 MFMAs, A operands from LDS four ahead: 16 x "P V" (four accumulator chains), then 16 x "K Q^T" (two chains)
      bf16x8 fr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fr[i] = *(const bf16x8*)(base + off + i * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int idx = 0; idx < 32; ++idx) {
        if (idx < 16) oacc[idx & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[idx & 3], pf[idx >> 2], oacc[idx & 3], 0, 0, 0);
        else sacc[idx & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[idx & 3], qf[(idx - 16) >> 1], sacc[idx & 1], 0, 0, 0);
        if (idx + 4 < 32) fr[idx & 3] = *(const bf16x8*)(base + off + ((idx + 4) & 31) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // the softmax's VALU mix on the 32 scores of this lane
      float tmax = m_run;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; e += 2) tmax = fmaxf(fmaxf(tmax, sacc[st][e]), sacc[st][e + 1]);
      m_run = tmax;
      const float nmc = -tmax * 0.1f;
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f32x8 pw;
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[st][8 * q + j], 0.1f, nmc));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[st][8 * q + j + 1], 0.1f, nmc));
            ps0 += p0; ps1 += p1; pw[j] = p0; pw[j + 1] = p1;
          }
          pf[st * 2 + q] = __builtin_convertvector(pw, bf16x8);
        }
      l_run += ps0 + ps1;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[st][e] = 0.001f * e;      // (the next K Q^T starts from fresh accumulators)
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = l_run + m_run;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += oacc[i][e];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += sacc[i][e];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, const uint4* src, float* out, unsigned long long* stamps, int iters) {
  CHECK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 98304, 0, src, out, iters, stamps);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 98304, 0, src, out, iters, stamps);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  std::vector<unsigned long long> h(512);
  CHECK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 512, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (int b = 0; b < 256; ++b) cyc += (double)h[2 * b];
  cyc /= 256;
  printf("%-44s %8.3f ms   %8.0f cycles per interval\n", name, ms, cyc / iters);
}

int main() {
  uint4* src; float* out; unsigned long long* stamps;
  CHECK(hipMalloc((void**)&src, 65536)); CHECK(hipMemset(src, 0x3c, 65536));
  CHECK(hipMalloc((void**)&out, sizeof(float) * 256 * 512));
  CHECK(hipMalloc((void**)&stamps, sizeof(unsigned long long) * 512));
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    run<2>("MSEG only (both groups, every interval)", src, out, stamps, iters);
    run<3>("VSEG only", src, out, stamps, iters);
    run<1>("in phase: MSEG | VSEG | MSEG ...", src, out, stamps, iters);
    run<0>("alternating: group 0 MSEG while group 1 VSEG", src, out, stamps, iters);
  }
  return 0;
}
