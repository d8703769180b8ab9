// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the scorer hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(8))) uint16_t u16x8;
typedef __attribute__((ext_vector_type(4))) uint16_t u16x4;

#define AIGV_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
// one bf16 rounding point of the reference's eager bf16 path
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// erf GELU, x * 0.5 * (1 + erf(x / sqrt 2)), for the GEMM epilogues (VALU-bound there: libm erff is ~38 instructions with both
// of its branches live in a wave, this is 15).  With a = min(|x|, 6):
//   Phi(-a) = 2^-(1 + a R(a)),  R = degree-7 fit of (-log2 Phi(-a) - 1) / a on [0, 6]  (one polynomial, one v_exp_f32)
//   erf(a / sqrt 2) = 1 - 2 Phi(-a), sign copied from x; the final (1 + erf) keeps the reference formula's cancellation
// Checked over ALL bf16 inputs against torch's CPU bf16 GELU (tests/test_gpu_ops.py): identical bf16 results except for
// a handful of last-place differences, the deep negative tail (x < -4, |gelu| < 3e-5, where the reference's own 1 + erf has
// lost its digits) and outputs below the smallest normal.
__device__ __forceinline__ float gelu_fast(float x) {
  const float a = fminf(fabsf(x), 6.0f);
  float r = 2.128495187e-07f;
  r = fmaf(r, a, -3.116951266e-06f);
  r = fmaf(r, a, -1.507227055e-05f);
  r = fmaf(r, a, 7.047377466e-04f);
  r = fmaf(r, a, -7.908032378e-03f);
  r = fmaf(r, a, 5.311047467e-02f);
  r = fmaf(r, a, 4.590370103e-01f);
  r = fmaf(r, a, 1.151112778e+00f);
  const float h = __builtin_amdgcn_exp2f(fmaf(-r, a, -1.0f));   // Phi(-a)
  const float e = copysignf(fmaf(-2.0f, h, 1.0f), x);            // erf(x / sqrt 2)
  return (x * 0.5f) * (1.0f + e);
}
// x * sigmoid(x) with hardware exp2 / rcp (about 3 ulp in fp32, i.e. ~2e-4 of the results round to the neighbouring
// bf16 value relative to an exactly rounded evaluation - the same order as the vectorised CPU kernels' own exp error);
// 6 VALU instructions instead of 27 for expf + IEEE division: the SwiGLU epilogue is VALU-bound
__device__ __forceinline__ float silu_f(float x) {
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x);   // exp(-x); overflow -> inf -> rcp -> 0 -> x*0
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---- two-at-a-time forms for the GEMM epilogues (VALU-bound): gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at
// the rate of their scalar forms, and v_cvt_pk_bf16_f32 rounds two values per instruction.  Per component these are the
// same IEEE operations as the scalar helpers above - results are bit-identical to them.
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf2(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ f32x2 unpack_bf2(uint32_t p) { return f32x2{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 rbf2(f32x2 v) { return unpack_bf2(pack_bf2(v)); }
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 a = f32x2{fminf(fabsf(x.x), 6.0f), fminf(fabsf(x.y), 6.0f)};
  f32x2 r = (f32x2)(2.128495187e-07f);
  r = __builtin_elementwise_fma(r, a, (f32x2)(-3.116951266e-06f));
  r = __builtin_elementwise_fma(r, a, (f32x2)(-1.507227055e-05f));
  r = __builtin_elementwise_fma(r, a, (f32x2)(7.047377466e-04f));
  r = __builtin_elementwise_fma(r, a, (f32x2)(-7.908032378e-03f));
  r = __builtin_elementwise_fma(r, a, (f32x2)(5.311047467e-02f));
  r = __builtin_elementwise_fma(r, a, (f32x2)(4.590370103e-01f));
  r = __builtin_elementwise_fma(r, a, (f32x2)(1.151112778e+00f));
  const f32x2 t = __builtin_elementwise_fma(-r, a, (f32x2)(-1.0f));
  const f32x2 h = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  f32x2 e = __builtin_elementwise_fma((f32x2)(-2.0f), h, (f32x2)(1.0f));
  e = f32x2{copysignf(e.x, x.x), copysignf(e.y, x.y)};
  return (x * 0.5f) * (e + 1.0f);
}
__device__ __forceinline__ f32x2 silu2(f32x2 x) {
  const f32x2 t = x * -1.4426950408889634f;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
  return x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// async 16-byte-per-lane global -> LDS copy; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// The same LDS-DMA issued from inline asm (per-lane 64-bit source address): hipcc does not see it, so it neither counts it
// in its vmcnt bookkeeping nor inserts a conservative vmcnt(0) in front of later LDS reads - the caller owns the wait
// (s_waitcnt vmcnt(N) + barrier before the data is read).  M0 = LDS destination base, saved and restored.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_wave_base)
      : "memory");
}

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (saddr form): one VGPR per
// stream instead of a 64-bit address pair, and invisible to hipcc's vmcnt bookkeeping (the schedule counts by hand).
// M0 (the LDS destination base) is written in the same statement that uses it and restored afterwards.
__device__ __forceinline__ void glds16_saddr(const char* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}


// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, DEVICE): a process that drives several devices must set it on
// each of them.  One instance per kernel instantiation (a function-local static); bit d of `done` = set on device d.  Devices >= 64
// set it on every launch.  Thread-safe: a concurrent first launch sets the attribute twice, which is harmless.
struct LdsAttrOnce {
  std::atomic<unsigned long long> done{0};
  hipError_t ensure(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return e;
  }
};
