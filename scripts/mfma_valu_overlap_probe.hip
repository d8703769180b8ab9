// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?  (round 3: the prefill-attention kernel's MFMA and
// VALU segments add up at two waves per SIMD, and neither static nor phase-wise s_setprio changes that.)
// One workgroup of 8 waves per CU = two waves per SIMD (waves w and w + 4 share a SIMD).  Roles per wave: M = a stream of independent
// v_mfma_f32_32x32x16_bf16 (4 accumulator chains), V = a stream of independent v_fma_f32 (16 chains), X = v_exp_f32 + v_fma_f32 mix,
// idle = exits at once.  Reported: wall time of the launch for (M, idle), (V, idle), (M, V), (M, M), (V, V), and the same with the
// 16x16x32 shape - if M + V takes max(M, V) the pipes overlap across waves, if it takes M + V they serialise.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_abl/mfma_valu_overlap_probe scripts/mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { IDLE = 0, MF32 = 1, MF16 = 2, VFMA = 3, VEXP = 4, MIX4 = 5, MIX8 = 6, MIX6S = 7, VCVT = 8, VMAX3 = 9, VEXPO = 10, VPKFMA = 11, VADD = 12 };

template <int ROLE>
__device__ __forceinline__ float work(int iters, float seed) {
  if constexpr (ROLE == MF32) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    return s;
  } else if constexpr (ROLE == MF16) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = seed;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    return s;
  } else if constexpr (ROLE == VFMA) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0000001f, 1e-9f);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
  } else if constexpr (ROLE == VEXP) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * 1e-3f + i * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0f;
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
  }
  if constexpr (ROLE == VCVT) {          // v_cvt_pk_bf16_f32 (two floats -> packed bf16), 16 independent chains
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          typedef __attribute__((ext_vector_type(2))) float f2;
          typedef __attribute__((ext_vector_type(2))) __bf16 b2;
          const b2 r = __builtin_convertvector(f2{v[i], v[(i + 1) & 15]}, b2);
          v[i] = __uint_as_float(__builtin_bit_cast(unsigned, r));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
  }
  if constexpr (ROLE == VMAX3 || ROLE == VADD) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if constexpr (ROLE == VMAX3) v[i] = fmaxf(fmaxf(v[i], v[(i + 5) & 15]), seed);
          else v[i] = v[i] + 1e-9f;
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
  }
  if constexpr (ROLE == VEXPO) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * 1e-3f + i * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
  }
  if constexpr (ROLE == VPKFMA) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f2{seed + i, seed - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], f2{1.0000001f, 1.0000001f}, f2{1e-9f, 1e-9f});
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    return s;
  }
  if constexpr (ROLE == MIX4 || ROLE == MIX8 || ROLE == MIX6S) {
    // ONE wave's own stream: every v_mfma_f32_32x32x16_bf16 followed by K independent v_fma_f32 (MIX6S: pinned with sched_group_barrier)
    constexpr int K = ROLE == MIX4 ? 4 : ROLE == MIX8 ? 8 : 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < K; ++j) { const int q = (u * 4 * K + i * K + j) & 15; v[q] = __builtin_fmaf(v[q], 1.0000001f, 1e-9f); }
        }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
      }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
  }
  return seed;
}

// waves 0-3 take role A, waves 4-7 role B (w and w + 4 share a SIMD)
template <int A, int B>
__global__ __launch_bounds__(512, 1) void probe(float* out, int itA, int itB, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float r;
  if (wave < 4) r = work<A>(itA, seed);
  else r = work<B>(itB, seed);
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int A, int B>
float run(float* out, int itA, int itB) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<A, B>), dim3(256), dim3(512), 0, 0, out, itA, itB, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<A, B>), dim3(256), dim3(512), 0, 0, out, itA, itB, 1.0f);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

int main() {
  float* out;
  CHECK(hipMalloc((void**)&out, 4096));
  // iteration counts chosen so that each role alone takes about the same time: MF32 16 MFMAs x 32 cycles = 512 cycles / iteration,
  // MF16 32 x 16 = 512, VFMA 128 x 4 = 512, VEXP 64 exp (8 or 16 cycles each) + 64 sub
  const int n = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    printf("M32 alone        %8.3f ms\n", run<MF32, IDLE>(out, n, 0));
    printf("M16 alone        %8.3f ms\n", run<MF16, IDLE>(out, n, 0));
    printf("VFMA alone       %8.3f ms\n", run<VFMA, IDLE>(out, n, 0));
    printf("VEXP alone       %8.3f ms\n", run<VEXP, IDLE>(out, n, 0));
    printf("M32 + VFMA       %8.3f ms\n", run<MF32, VFMA>(out, n, n));
    printf("M16 + VFMA       %8.3f ms\n", run<MF16, VFMA>(out, n, n));
    printf("M32 + VEXP       %8.3f ms\n", run<MF32, VEXP>(out, n, n));
    printf("M32 + M32        %8.3f ms\n", run<MF32, MF32>(out, n, n));
    printf("VFMA + VFMA      %8.3f ms\n", run<VFMA, VFMA>(out, n, n));
    printf("VEXP + VFMA      %8.3f ms\n", run<VEXP, VFMA>(out, n, n));
    printf("v_cvt_pk_bf16_f32 alone %8.3f ms   with M32 on the partner wave %8.3f ms\n", run<VCVT, IDLE>(out, n, 0), run<MF32, VCVT>(out, n, n));
    printf("v_max3_f32 alone        %8.3f ms   with M32 on the partner wave %8.3f ms\n", run<VMAX3, IDLE>(out, n, 0), run<MF32, VMAX3>(out, n, n));
    printf("v_add_f32 alone         %8.3f ms   with M32 on the partner wave %8.3f ms\n", run<VADD, IDLE>(out, n, 0), run<MF32, VADD>(out, n, n));
    printf("v_exp_f32 alone         %8.3f ms   with M32 on the partner wave %8.3f ms\n", run<VEXPO, IDLE>(out, n, 0), run<MF32, VEXPO>(out, n, n));
    printf("v_pk_fma_f32 alone      %8.3f ms   with M32 on the partner wave %8.3f ms\n", run<VPKFMA, IDLE>(out, n, 0), run<MF32, VPKFMA>(out, n, n));
    printf("one wave per SIMD: MFMA + 4 fma interleaved   %8.3f ms\n", run<MIX4, IDLE>(out, n, 0));
    printf("one wave per SIMD: MFMA + 6 fma interleaved   %8.3f ms\n", run<MIX6S, IDLE>(out, n, 0));
    printf("one wave per SIMD: MFMA + 8 fma interleaved   %8.3f ms\n", run<MIX8, IDLE>(out, n, 0));
    printf("two waves per SIMD, both MFMA + 4 fma         %8.3f ms\n", run<MIX4, MIX4>(out, n, n));
  }
  return 0;
}
