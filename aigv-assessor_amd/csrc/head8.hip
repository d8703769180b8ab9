// e4m3 form of the decode GEMVs: the fp8 mode of the InternLM2 linears (aigv_set_precision, include/aigv_amd.h; BASELINE config 5)
// applied to the q_len = 1 steps of generate() (reference loop: modeling_internlm2.py:1126-1163; the reference has no fp8 path, the
// arithmetic is the one oracle/fp8.py states: activations quantised per token row, weights per output channel, fp32 accumulation of
// exact e4m3 products, y = bf16((acc * row scale) * channel scale)).
//
// A decode step is a weight stream; with one byte per weight the stream is half as long.  Structure = head.hip's skinny_kernel in
// its decode forms (one x row tile of R <= 4 rows, 4 waves = 4 K slices, sub-slab forms P, RoPE / KV-append / SwiGLU / residual
// epilogues) with three differences:
//   * W fragments are 32 e4m3 bytes per lane and k-step (two 16-byte loads), the MFMA is v_mfma_scale_f32_16x16x128_f8f6f4 with
//     unit block scales (K = 128 per instruction);
//   * every workgroup quantises the x rows itself, in front of its K loop and behind its first weight loads: optional RMSNorm
//     (rmsnorm_quant_fp8_kernel's arithmetic, so the same bytes), row amax, e4m3 bytes + row scale into LDS - the B fragments
//     come from there;
//   * the epilogue multiplies the fp32 sums by the row scale, then by the channel scale, before the bf16 rounding points of the
//     bf16 form.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

enum { SK_RESID = 1, SK_SWIGLU = 2, SK_ROPE_KV = 7 };   // head.hip's numbering

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) int i32x2;

__device__ __forceinline__ v8i_t cat8(v4i_t l, v4i_t h) { return v8i_t{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]}; }

// NCH = K / 2048: 16-byte chunks of an x row per thread (256 threads).  NORM needs the norm weight too (NCH <= 3: the hidden widths).
template <int EPI, bool NORM, int NCH, int P>
__global__ __launch_bounds__(256) void skinny8_kernel(const bf16_t* __restrict__ x, int ldx, int R, const uint8_t* __restrict__ W8, int ldw,
                                                      const float* __restrict__ w_scale, int N, int K, const bf16_t* __restrict__ resid, int ldr,
                                                      bf16_t* __restrict__ out, int ldo, const AigvRopeKv rk, const bf16_t* __restrict__ norm_w,
                                                      float eps) {
  constexpr int NS = EPI == SK_RESID ? 1 : 2;
  constexpr int RS = 16 / P;
  extern __shared__ __attribute__((aligned(16))) uint8_t xs8[];   // the quantised x rows [R][K]
  __shared__ float part[3][NS][4][64];
  __shared__ float red[4], redm[4], sx[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = fr % RS, sp = fr / RS;
  const int n0 = EPI == SK_ROPE_KV ? (int)(blockIdx.x / (64 / RS)) * 128 + (int)(blockIdx.x % (64 / RS)) * RS
                 : EPI == SK_SWIGLU ? (int)(blockIdx.x / P) * 32 + (int)(blockIdx.x % P) * RS
                                    : blockIdx.x * RS;
  constexpr int SLAB_STEP = EPI == SK_ROPE_KV ? 64 : 16;
  const int kper = K / (4 * P), kbeg = (wave * P + sp) * kper;      // bytes = elements; kper % 128 == 0 checked by the launcher

  const uint8_t* wrow[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) wrow[s] = W8 + (size_t)min(n0 + s * SLAB_STEP + sr, N - 1) * ldw + kbeg + fq * 32;
  f32x4 acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- x rows -> (RMSNorm) -> e4m3 + row scale, in LDS.  Load order matters (vmcnt retires in issue order): row 0 of x and the norm
  // weight first, then the first two weight k-steps, which stay in flight behind the statistics.
  u16x8 raw[NCH], gw[NORM ? NCH : 1];
#pragma unroll
  for (int c = 0; c < NCH; ++c) raw[c] = *(const u16x8*)(x + ((threadIdx.x + c * 256) << 3));
  if constexpr (NORM) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) gw[c] = *(const u16x8*)(norm_w + ((threadIdx.x + c * 256) << 3));
  }
  constexpr int PF = 2;                                              // kper >= 256 checked by the launcher
  v4i_t wpf[PF][NS][2];
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      wpf[u][s][0] = *(const v4i_t*)(wrow[s] + 128 * u);
      wpf[u][s][1] = *(const v4i_t*)(wrow[s] + 128 * u + 16);
    }
  for (int r = 0; r < R; ++r) {
    if (r > 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) raw[c] = *(const u16x8*)(x + (size_t)r * ldx + ((threadIdx.x + c * 256) << 3));
    }
    float rstd = 1.f;
    if constexpr (NORM) {
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = bf2f(raw[c][e]); sq += v * v; }
      rstd = rsqrtf(block_sum_256(sq, red) / (float)K + eps);
    }
    auto value = [&](int c, int e) {                                // the bf16 value the bf16 path would feed the linear
      if constexpr (NORM) return rbf(bf2f(gw[c][e]) * rbf(bf2f(raw[c][e]) * rstd));
      else return bf2f(raw[c][e]);
    };
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(value(c, e)));
    amax = wave_max(amax);
    __syncthreads();                                                 // redm of the previous row has been read
    if (lane == 0) redm[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 1.0f;
    if (threadIdx.x == 0) sx[r] = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(value(c, 0) * inv, value(c, 1) * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(value(c, 2) * inv, value(c, 3) * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(value(c, 4) * inv, value(c, 5) * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(value(c, 6) * inv, value(c, 7) * inv, hi, true);
      *(i32x2*)(xs8 + (size_t)r * K + ((threadIdx.x + c * 256) << 3)) = i32x2{lo, hi};
    }
  }
  __syncthreads();

  // ---- K loop: lane (fr, fq) holds W row sr / x row sr over K sub-range sp, bytes [32 fq, 32 fq + 32) of every 128-byte k-step ----
  const uint8_t* xf0 = xs8 + (size_t)min(sr, R - 1) * K + kbeg + fq * 32;
  auto xfrag = [&](int k) { return cat8(*(const v4i_t*)(xf0 + k), *(const v4i_t*)(xf0 + k + 16)); };
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const v8i_t xf = xfrag(128 * u);
#pragma unroll
    for (int s = 0; s < NS; ++s)
      acc[s] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(wpf[u][s][0], wpf[u][s][1]), xf, acc[s], 0, 0, 0, 127, 0, 127);
  }
  int k = 128 * PF;
  auto run = [&](auto dtag) {
    constexpr int DD = decltype(dtag)::value;
    for (; k + 128 * DD <= kper; k += 128 * DD) {
      v4i_t wf[DD][NS][2];
#pragma unroll
      for (int u = 0; u < DD; ++u)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          wf[u][s][0] = *(const v4i_t*)(wrow[s] + k + 128 * u);
          wf[u][s][1] = *(const v4i_t*)(wrow[s] + k + 128 * u + 16);
        }
#pragma unroll
      for (int u = 0; u < DD; ++u) {
        const v8i_t xf = xfrag(k + 128 * u);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          acc[s] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(wf[u][s][0], wf[u][s][1]), xf, acc[s], 0, 0, 0, 127, 0, 127);
      }
    }
  };
  constexpr int D8 = NS == 1 ? 8 : 4;                                // 256 bytes per lane in flight, as the bf16 form
  run(std::integral_constant<int, D8>{});
  if constexpr (D8 > 4) run(std::integral_constant<int, 4>{});
  run(std::integral_constant<int, 2>{});
  run(std::integral_constant<int, 1>{});

  // ---- combine the four K slices (fixed order), then the P diagonal blocks ----
  if (wave > 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[wave - 1][s][e][lane] = acc[s][e];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[s][e] += (part[0][s][e][lane] + part[1][s][e][lane]) + part[2][s][e][lane];
  if constexpr (P == 2) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s][e] += __shfl(acc[s][e], (lane + 40) & 63, 64);
  } else if constexpr (P == 4) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[s][e];
        acc[s][e] = ((v + __shfl(v, (lane + 20) & 63, 64)) + __shfl(v, (lane + 40) & 63, 64)) + __shfl(v, (lane + 60) & 63, 64);
      }
  }
  const bool own = P == 1 || (fr < RS && fq < RS / 4);
  const int r = own ? fr : R;
  if (r >= R) return;
  // (acc * row scale) * channel scale: oracle/fp8.py's order
  const float rs = sx[r];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4 cs = *(const f32x4*)(w_scale + n0 + s * SLAB_STEP + 4 * fq);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[s][e] = (acc[s][e] * rs) * cs[e];
  }
  if constexpr (EPI == SK_ROPE_KV) {
    const int hs = blockIdx.x / (64 / RS), d = (int)(blockIdx.x % (64 / RS)) * RS + 4 * fq;   // head slot; dims d .. d+3 and d+64 .. d+67
    const int slot = hs % (rk.g + 2), gi = hs / (rk.g + 2);
    const int p = rk.pos[r];
    u16x4 olo, ohi;
    if (slot <= rk.g) {
      const u16x4 co = *(const u16x4*)(rk.cos + (size_t)p * 64 + d), si = *(const u16x4*)(rk.sin + (size_t)p * 64 + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x1 = rbf(acc[0][e]), x2 = rbf(acc[1][e]), cc = bf2f(co[e]), ss = bf2f(si[e]);
        olo[e] = f2bf(rbf(x1 * cc) + rbf(-x2 * ss));
        ohi[e] = f2bf(rbf(x2 * cc) + rbf(x1 * ss));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { olo[e] = f2bf(acc[0][e]); ohi[e] = f2bf(acc[1][e]); }
    }
    bf16_t* dst = slot < rk.g ? out + (size_t)r * ldo + (size_t)hs * 128 + d
                              : (slot == rk.g ? rk.kc : rk.vc) + (((size_t)rk.seq[r] * rk.n_kv + gi) * rk.cap + p) * 128 + d;
    *(u16x4*)dst = olo;
    *(u16x4*)(dst + 64) = ohi;
  } else if constexpr (EPI == SK_SWIGLU) {
    const int n = (int)(blockIdx.x / P) * 16 + (int)(blockIdx.x % P) * RS + 4 * fq;
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = rbf(acc[0][e]), u = rbf(acc[1][e]);
      o[e] = f2bf(rbf(silu_f(g)) * u);
    }
    *(u16x4*)(out + (size_t)r * ldo + n) = o;
  } else {
    const int n = n0 + 4 * fq;
    const u16x4 rr = *(const u16x4*)(resid + (size_t)r * ldr + n);
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(rbf(bf2f(rr[e]) + rbf(acc[0][e])));
    *(u16x4*)(out + (size_t)r * ldo + n) = o;
  }
}

template <int EPI, bool NORM, int NCH>
hipError_t launch8p(int p, int blocks, size_t lds, hipStream_t s, const bf16_t* x, int ldx, int R, const uint8_t* W8, int ldw, const float* w_scale, int N,
                    int K, const bf16_t* resid, int ldr, bf16_t* out, int ldo, const AigvRopeKv& rk, const bf16_t* norm_w, float eps) {
  // the quantised rows of a long-K GEMV (4 x 16384 bytes) exceed the default dynamic-LDS limit
#define GO(PP)                                                                                                                                        \
  do {                                                                                                                                                \
    static LdsAttrOnce lds_attr;                                                                                                                     \
    if (hipError_t e = lds_attr.ensure((const void*)skinny8_kernel<EPI, NORM, NCH, PP>, 4 * 16384); e != hipSuccess) return e; \
    hipLaunchKernelGGL((skinny8_kernel<EPI, NORM, NCH, PP>), dim3(blocks), dim3(256), lds, s, x, ldx, R, W8, ldw, w_scale, N, K, resid, ldr, out, ldo, \
                       rk, norm_w, eps);                                                                                                              \
  } while (0)
  if (p == 1) GO(1); else if (p == 2) GO(2); else GO(4);
#undef GO
  return hipGetLastError();
}

}  // namespace

// epi: 1 residual (wo / w2), 2 swiglu (w1|w3 interleaved in 16-row blocks), 7 wqkv with RoPE + KV-cache append (rk).  norm_w != null:
// x is the raw residual stream and the RMSNorm in front of the linear is applied here (wqkv, w1|w3).  R <= 16 / p and <= 4 rows;
// K = 2048 j with j in {2, 3} (with a norm) or {2, 3, 7, 8}; ldw in bytes.
hipError_t aigv_launch_skinny_fp8(const bf16_t* x, int ldx, int R, const uint8_t* W8, int ldw, const float* w_scale, int N, int K, const bf16_t* resid,
                                  int ldr, bf16_t* out, int ldo, int epi, const AigvRopeKv* rk, const bf16_t* norm_w, float eps, int p, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  if ((p != 1 && p != 2 && p != 4) || R > 4 || R > 16 / p || !aigv_skinny_fp8_supported(K, norm_w != nullptr) || K % (512 * p) || K / (4 * p) < 256 ||
      (ldx % 8) || (ldw % 16) || (ldo % 4) || !x || !W8 || !w_scale || !out)
    return hipErrorInvalidValue;
  const int rs = 16 / p, nch = K / 2048;
  const size_t lds = (size_t)R * K;
  const AigvRopeKv none{};
  if (epi == SK_ROPE_KV) {
    if (!rk || !norm_w || N % 128 || N != rk->n_kv * (rk->g + 2) * 128) return hipErrorInvalidValue;
    const int blocks = N / 128 * (64 / rs);
    return nch == 2 ? launch8p<SK_ROPE_KV, true, 2>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, nullptr, 0, out, ldo, *rk, norm_w, eps)
                    : launch8p<SK_ROPE_KV, true, 3>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, nullptr, 0, out, ldo, *rk, norm_w, eps);
  }
  if (epi == SK_SWIGLU) {
    if (!norm_w || N % 32) return hipErrorInvalidValue;
    const int blocks = N / 32 * p;
    return nch == 2 ? launch8p<SK_SWIGLU, true, 2>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, nullptr, 0, out, ldo, none, norm_w, eps)
                    : launch8p<SK_SWIGLU, true, 3>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, nullptr, 0, out, ldo, none, norm_w, eps);
  }
  if (epi == SK_RESID) {
    if (norm_w || !resid || N % rs || (ldr % 4)) return hipErrorInvalidValue;
    const int blocks = N / rs;
    switch (nch) {
      case 2: return launch8p<SK_RESID, false, 2>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, resid, ldr, out, ldo, none, nullptr, 0.f);
      case 3: return launch8p<SK_RESID, false, 3>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, resid, ldr, out, ldo, none, nullptr, 0.f);
      case 7: return launch8p<SK_RESID, false, 7>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, resid, ldr, out, ldo, none, nullptr, 0.f);
      case 8: return launch8p<SK_RESID, false, 8>(p, blocks, lds, s, x, ldx, R, W8, ldw, w_scale, N, K, resid, ldr, out, ldo, none, nullptr, 0.f);
    }
  }
  return hipErrorInvalidValue;
}

bool aigv_skinny_fp8_supported(int K, bool with_norm) {
  const int nch = K / 2048;
  if (K % 2048) return false;
  return with_norm ? (nch == 2 || nch == 3) : (nch == 2 || nch == 3 || nch == 7 || nch == 8);
}
