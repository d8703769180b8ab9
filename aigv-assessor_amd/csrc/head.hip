// Skinny (<= 64 rows) weight-streaming GEMM for gfx950, used where the weights are read once per call and the
// kernel is HBM-bound:  the lm-head on the answer rows with a fused vocabulary argmax
// (reference: modeling_internlm2.py:1094-1096 + modeling_internvl_chat.py:451-463,488), the q_len = 1 decode
// steps of generate() (modeling_internlm2.py:1126-1163), the motion projector (M = clips), and the score head
// (modeling_internvl_chat.py:43-94).
//
// One workgroup = 4 waves = one 16-row slab of W (two slabs for SwiGLU); the waves split K four ways, stream
// their W fragments straight from global memory into MFMA A-operands (no LDS round trip: every weight byte
// is used once), keep the x fragments (L2-resident) as B-operands, and combine partial sums through LDS.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

#ifndef SKINNY_DEPTH2
#define SKINNY_DEPTH2 8      // k-steps in flight of the two-slab one-row-tile forms (SwiGLU w1|w3, RoPE wqkv)
#endif
enum { SK_STORE = 0, SK_RESID = 1, SK_SWIGLU = 2, SK_GELU = 3, SK_ARGMAX = 4, SK_RELU = 5, SK_LS_RESID = 6, SK_ROPE_KV = 7 };

// SK_ROPE_KV: the decode step's wqkv GEMV with RoPE and the KV-cache append in its epilogue (head_dim 128).  A workgroup takes the
// two 16-row slabs of W that rotate_half pairs - dims 16 sub .. and 64 + 16 sub .. of one head slot - so a lane ends up holding
// x_i and x_{i+64} of the same token: query slots are rotated and stored to the qkv row, the K slot is rotated and the V slot
// copied straight into the caches (modeling_internlm2.py:247-261, 397-402; same three bf16 roundings as rope_kernel).
struct RopeKvArgs {
  const int32_t* pos; const int32_t* seq;      // position / cache sequence of every x row (device)
  const bf16_t* cos; const bf16_t* sin;        // [max_pos, 64]
  bf16_t* kc; bf16_t* vc;                      // [seq][kv head][cap][128]
  int g, n_kv, cap;
};

// NORM: the GEMV applies the RMSNorm in front of it (attention_norm before wqkv, ffn_norm before w1|w3; modeling_internlm2.py:138-143)
// to its x rows itself: x is the raw residual stream, every workgroup normalises the R <= 4 rows into LDS with rmsnorm_kernel's exact
// arithmetic (same thread -> chunk mapping, same reduction tree) and takes its B fragments from there.  Redundant work per
// workgroup (R x K elements), but no launch: the statistics' L2 round trip and barriers run behind the first weight loads.
struct NormArgs { const bf16_t* w; float eps; };

__device__ __forceinline__ unsigned int ord_f32(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// NWV: waves per workgroup = K slices.  8 for the decode GEMVs whose N gives at most one workgroup per CU (wo, w2: 256 slabs of 16
// rows): twice the loads in flight per CU, which is what bounds a weight stream at this occupancy (in-box A/B, scripts/decode_gemv_bench.py:
// wo 7.9 -> 7.5 us, w2 26.6 -> 24.8 us = 4.7 TB/s; wqkv with 384 slabs is faster with 4 waves, 12.5 vs 13.2 us).
// NT: the weight fragments are loaded non-temporally (each byte is used once: no point in keeping it in L2 / Infinity Cache, and a
// once-read stream lands sooner with the nt policy - MI355X_MICROARCH.md 'nt-weights').
template <bool NT>
__device__ __forceinline__ bf16x8 wload(const bf16_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load((const bf16x8*)p);
  else return *(const bf16x8*)p;
}

// P: sub-slab form of a decode GEMV (one x row tile, R <= 16 / P rows).  A workgroup takes RS = 16 / P rows of W per slab instead
// of 16, and the MFMA's other rows / columns carry P - 1 further K sub-ranges of the SAME rows: A row m = W row (m mod RS) over
// sub-range (m / RS), B column n = x row (n mod RS) over sub-range (n / RS), so D[m][n] is a partial product wherever m / RS ==
// n / RS and the result is the sum of those P diagonal blocks (a shuffle per block).  Every lane still loads distinct weight
// bytes; what changes is the GRID: N / RS workgroups of 1 / P the bytes - a width whose 16-row slabs come to a non-integer
// number of workgroups per CU (w1|w3 of InternLM2-7B: 3.5; fused wqkv: 0.75) runs at the rate of the next integer
// (scripts/gemv_balance_probe.py: 3.5 per CU 4.97 TB/s, 3 per CU 5.34, 4 per CU 5.45).
template <int RT, int EPI, int NWV = 4, bool NT = false, int NC = 0, int P = 1>   // NC: fused-norm form, K = NC x 2048 (0 = off)
__global__ __launch_bounds__(NWV * 64) void skinny_kernel(const bf16_t* __restrict__ x, int ldx, int R,
                                                     const bf16_t* __restrict__ W, int ldw, int N, int K,
                                                     const bf16_t* __restrict__ bias, const bf16_t* __restrict__ resid,
                                                     int ldr, bf16_t* __restrict__ out, int ldo,
                                                     unsigned long long* __restrict__ packed,
                                                     const bf16_t* __restrict__ ls, const RopeKvArgs rk = RopeKvArgs{},
                                                     const NormArgs na = NormArgs{}) {
  constexpr bool NORM = NC > 0;
  static_assert(!NORM || (RT == 1 && NWV == 4), "the fused norm is the decode form: one row tile, 256 threads (rmsnorm_kernel's reduction)");
  extern __shared__ __attribute__((aligned(16))) bf16_t xs[];   // NORM: the normalised x rows [R][K]
  // W slabs (16 rows each) per workgroup.  The lm-head on the answer rows (40+ x rows) is bound by re-reading the x fragments
  // from L2 once per workgroup, not by streaming W: four slabs per workgroup share them.
  constexpr int NS = (EPI == SK_SWIGLU || EPI == SK_ROPE_KV) ? 2 : (EPI == SK_ARGMAX && RT >= 2) ? 4 : 1;
  static_assert(P == 1 || ((P == 2 || P == 4) && RT == 1 && EPI != SK_ARGMAX), "sub-slab forms: one row tile, 8 or 4 rows per slab");
  constexpr int RS = 16 / P;                         // W rows per slab (= x rows the form can take)
  __shared__ float part[NWV - 1][NS][RT][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = fr % RS, sp = fr / RS;              // this lane's slab row / x row, and its K sub-range (P == 1: fr, 0)
  // first W row of the workgroup's slab(s).  SK_ROPE_KV: block = (head slot, RS-dim group), its two slabs are 64 rows apart;
  // SK_SWIGLU: w1 / w3 alternate in 16-row blocks, block = (32-row pair, RS-row group), the slabs 16 rows apart
  const int n0 = EPI == SK_ROPE_KV ? (int)(blockIdx.x / (64 / RS)) * 128 + (int)(blockIdx.x % (64 / RS)) * RS
                 : EPI == SK_SWIGLU ? (int)(blockIdx.x / P) * 32 + (int)(blockIdx.x % P) * RS
                                    : blockIdx.x * RS * NS;
  constexpr int SLAB_STEP = EPI == SK_ROPE_KV ? 64 : 16;
  const int kper = K / (NWV * P), kbeg = (wave * P + sp) * kper;      // K % (32 * NWV * P) == 0 checked by the launcher

  const bf16_t* wrow[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) wrow[s] = W + (size_t)min(n0 + s * SLAB_STEP + sr, N - 1) * ldw + kbeg + fq * 8;
  const bf16_t* xrow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) xrow[t] = x + (size_t)min(t * 16 + sr, R - 1) * ldx + kbeg + fq * 8;

  f32x4 acc[NS][RT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DEPTH k-steps (DEPTH x 16 B per lane per operand) are loaded before their MFMAs so several loads are in flight; a decode
  // GEMV (one x row tile, few workgroups per CU when N is small) needs the deeper form to cover the HBM latency
  constexpr int DEPTH = RT == 1 ? (NS == 1 ? 16 : SKINNY_DEPTH2) : (RT == 2 && NS == 1) ? 8 : 4;   // (two row tiles: InternViT's 32 leftover rows per pass)
  int k = 0;
  const int xs_off = min(sr, R - 1) * K + kbeg + fq * 8;          // NORM: this lane's fragment origin in xs
  if constexpr (NORM) {
    // Load order matters (vmcnt retires in issue order): the x rows and the norm weight - a few L2 hits - go out FIRST, then the
    // first DEPTH weight k-steps; the statistics then wait only for the former, with the weight stream in flight behind them.
    // Rows past R repeat row R - 1 (never normalised); thread -> chunk mapping and sums as rmsnorm_kernel.
    u16x8 xr[4][NC], gw[NC];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) xr[r][c] = *(const u16x8*)(x + (size_t)min(r, R - 1) * ldx + ((threadIdx.x + c * 256) << 3));
#pragma unroll
    for (int c = 0; c < NC; ++c) gw[c] = *(const u16x8*)(na.w + ((threadIdx.x + c * 256) << 3));
    bf16x8 wf[DEPTH][NS];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
#pragma unroll
      for (int s = 0; s < NS; ++s) wf[u][s] = wload<NT>(wrow[s] + 32 * u);
    {
      __shared__ float red[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r == 0 || r < R) {     // (row 0 unconditionally: hipcc otherwise sinks the norm-weight loads into the branch, behind the weight stream)
          float sq = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = bf2f(xr[r][c][e]); sq += v * v; }
          const float rstd = rsqrtf(block_sum_256(sq, red) / (float)K + na.eps);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(gw[c][e]) * rbf(bf2f(xr[r][c][e]) * rstd));
            *(u16x8*)(xs + (size_t)r * K + ((threadIdx.x + c * 256) << 3)) = o;
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const bf16x8 xf = *(const bf16x8*)(xs + xs_off + 32 * u);
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][s], xf, acc[s][0], 0, 0, 0);
    }
    k = 32 * DEPTH;
  }
  // the rest of the K slice in groups of DEPTH, then 8 / 4 / 1 k-steps: a slice that is not a multiple of DEPTH (K = 14336 over 8
  // waves: 56 k-steps) used to finish one load at a time
  auto run = [&](auto dtag) {
    constexpr int DD = decltype(dtag)::value;
    for (; k + 32 * DD <= kper; k += 32 * DD) {
      bf16x8 wf[DD][NS], xf[DD][RT];
#pragma unroll
      for (int u = 0; u < DD; ++u) {
#pragma unroll
        for (int s = 0; s < NS; ++s) wf[u][s] = wload<NT>(wrow[s] + k + 32 * u);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          if constexpr (NORM) xf[u][t] = *(const bf16x8*)(xs + xs_off + k + 32 * u);
          else xf[u][t] = *(const bf16x8*)(xrow[t] + k + 32 * u);
        }
      }
#pragma unroll
      for (int u = 0; u < DD; ++u)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int t = 0; t < RT; ++t)
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][s], xf[u][t], acc[s][t], 0, 0, 0);
    }
  };
  run(std::integral_constant<int, DEPTH>{});
  if constexpr (DEPTH > 8) run(std::integral_constant<int, 8>{});
  if constexpr (DEPTH > 4) run(std::integral_constant<int, 4>{});
  run(std::integral_constant<int, 1>{});

  // ---- combine the four K slices: waves 1..3 publish, wave 0 sums in a fixed order -------------------------
  if (wave > 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[wave - 1][s][t][e][lane] = acc[s][t][e];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (NWV == 4) {
          acc[s][t][e] += (part[0][s][t][e][lane] + part[1][s][t][e][lane]) + part[2][s][t][e][lane];
        } else {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < NWV - 1; ++w) sum += part[w][s][t][e][lane];   // fixed order
          acc[s][t][e] += sum;
        }
      }

  // sub-slab forms: sum the P diagonal blocks (sub-range p lives in lanes fr = p RS + i, fq = p RS / 4 + j / 4), in order
  if constexpr (P == 2) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s][0][e] += __shfl(acc[s][0][e], (lane + 40) & 63, 64);
  } else if constexpr (P == 4) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[s][0][e];
        acc[s][0][e] = ((v + __shfl(v, (lane + 20) & 63, 64)) + __shfl(v, (lane + 40) & 63, 64)) + __shfl(v, (lane + 60) & 63, 64);
      }
  }
  const bool own = P == 1 || (fr < RS && fq < RS / 4);   // lanes that hold finished sums

  // lane owns x row r = 16t + fr and W rows n = n0 + 4*fq + e
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int r = own ? t * 16 + fr : R;
    if constexpr (EPI == SK_ARGMAX) {
      unsigned long long best = 0ull;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + sl * 16 + 4 * fq + e;
          if (n < N) {
            const unsigned long long key = ((unsigned long long)ord_f32(rbf(acc[sl][t][e])) << 32) | (0xFFFFFFFFu - (unsigned)n);
            best = key > best ? key : best;
          }
        }
      unsigned long long o = __shfl_xor(best, 16, 64); best = o > best ? o : best;
      o = __shfl_xor(best, 32, 64); best = o > best ? o : best;
      if (fq == 0 && r < R) atomicMax(packed + r, best);
    } else if constexpr (EPI == SK_ROPE_KV) {
      if (r < R) {
        const int hs = blockIdx.x / (64 / RS), d = (int)(blockIdx.x % (64 / RS)) * RS + 4 * fq;   // head slot; dims d .. d+3 and d+64 .. d+67
        const int slot = hs % (rk.g + 2), gi = hs / (rk.g + 2);
        const int p = rk.pos[r];
        u16x4 olo, ohi;
        if (slot <= rk.g) {
          const u16x4 co = *(const u16x4*)(rk.cos + (size_t)p * 64 + d), si = *(const u16x4*)(rk.sin + (size_t)p * 64 + d);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x1 = rbf(acc[0][t][e]), x2 = rbf(acc[1][t][e]), cc = bf2f(co[e]), ss = bf2f(si[e]);
            olo[e] = f2bf(rbf(x1 * cc) + rbf(-x2 * ss));
            ohi[e] = f2bf(rbf(x2 * cc) + rbf(x1 * ss));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { olo[e] = f2bf(acc[0][t][e]); ohi[e] = f2bf(acc[1][t][e]); }
        }
        bf16_t* dst = slot < rk.g ? out + (size_t)r * ldo + (size_t)hs * 128 + d
                                  : (slot == rk.g ? rk.kc : rk.vc) + (((size_t)rk.seq[r] * rk.n_kv + gi) * rk.cap + p) * 128 + d;
        *(u16x4*)dst = olo;
        *(u16x4*)(dst + 64) = ohi;
      }
    } else if constexpr (EPI == SK_SWIGLU) {
      if (r < R) {
        const int n = (int)(blockIdx.x / P) * 16 + (int)(blockIdx.x % P) * RS + 4 * fq;
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = rbf(acc[0][t][e]), u = rbf(acc[1][t][e]);
          o[e] = f2bf(rbf(silu_f(g)) * u);
        }
        *(u16x4*)(out + (size_t)r * ldo + n) = o;
      }
    } else {
      const int n = n0 + 4 * fq;
      if (r < R && n < N) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[0][t][e] + (bias ? bf2f(bias[n + e]) : 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
        if constexpr (EPI == SK_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rbf(gelu_fast(v[e]));
        }
        if constexpr (EPI == SK_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (EPI == SK_LS_RESID) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rbf(v[e] * bf2f(ls[n + e]));
        }
        if constexpr (EPI == SK_RESID || EPI == SK_LS_RESID) {
          const u16x4 rr = *(const u16x4*)(resid + (size_t)r * ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rbf(bf2f(rr[e]) + v[e]);
        }
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *(u16x4*)(out + (size_t)r * ldo + n) = o;
      }
    }
  }
}

__global__ void unpack_argmax_kernel(const unsigned long long* __restrict__ packed, int64_t* __restrict__ idx,
                                     float* __restrict__ val, int R) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const unsigned long long p = packed[r];
  idx[r] = (int64_t)(0xFFFFFFFFu - (unsigned)(p & 0xFFFFFFFFull));
  if (val) {
    unsigned int u = (unsigned)(p >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    val[r] = __uint_as_float(u);
  }
}

// ---- score head ------------------------------------------------------------------------------------------
// x = hidden[:, -4, :] (post final norm); if ANY NaN is present in the batch slice the reference applies
// nan_to_num(nan=0, posinf=1e9, neginf=-1e9) to every row (modeling_internvl_chat.py:469-473); then a chain
// of Linear+ReLU with a bf16 rounding after each Linear (:82-94).  The wide layers run on the skinny GEMM
// (weights streamed once for all clips); the narrow tail (fan-in < 128) runs in one small kernel.
__global__ __launch_bounds__(256) void score_guard_kernel(const bf16_t* __restrict__ x, int ldx, int B, int H,
                                                          bf16_t* __restrict__ out) {
  __shared__ int any_nan;
  if (threadIdx.x == 0) any_nan = 0;
  __syncthreads();
  int nan_here = 0;
  for (int i = threadIdx.x; i < B * H; i += 256) {
    const float v = bf2f(x[(size_t)(i / H) * ldx + (i % H)]);
    nan_here |= (v != v);
  }
  if (nan_here) atomicOr(&any_nan, 1);
  __syncthreads();
  const bool fix = any_nan != 0;
  for (int i = threadIdx.x; i < B * H; i += 256) {
    float v = bf2f(x[(size_t)(i / H) * ldx + (i % H)]);
    if (fix) {
      if (v != v) v = 0.f;
      else if (isinf(v)) v = v > 0 ? 1e9f : -1e9f;
    }
    out[i] = f2bf(v);
  }
}

// remaining narrow layers: one workgroup per clip; activations ping-pong in LDS (all dims <= 1024 here)
__global__ __launch_bounds__(256) void score_tail_kernel(const ScoreHeadArgs a, int first_layer) {
  __shared__ float buf[2][1024];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d0 = a.dims[first_layer];
  for (int i = threadIdx.x; i < d0; i += 256) buf[0][i] = bf2f(a.x[(size_t)b * a.ldx + i]);
  __syncthreads();
  int cur = 0;
  for (int L = first_layer; L < a.n_layers; ++L) {
    const int din = a.dims[L], dout = a.dims[L + 1];
    const bf16_t* w = a.w[L];
    const bf16_t* bb = a.b[L];
    for (int o = wave; o < dout; o += 4) {
      float s = 0.f;
      for (int i = lane; i < din; i += 64) s += bf2f(w[(size_t)o * din + i]) * buf[cur][i];
      s = wave_sum(s);
      if (lane == 0) buf[cur ^ 1][o] = fmaxf(rbf(s + bf2f(bb[o])), 0.f);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (threadIdx.x == 0) a.score[b] = buf[cur][0];
}

template <int EPI>
hipError_t launch_skinny(const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K, const bf16_t* bias,
                         const bf16_t* resid, int ldr, bf16_t* out, int ldo, unsigned long long* packed,
                         hipStream_t s, const bf16_t* ls = nullptr, int p = 1) {
  const int rt = (R + 15) / 16;
  if (p > 1) {   // sub-slab forms of the decode GEMVs (skinny_kernel's P): 16 / p rows per workgroup and slab
    const int rs = 16 / p;
    if ((p != 2 && p != 4) || R > rs || K % (128 * p) || N % (EPI == SK_SWIGLU ? 32 : rs)) return hipErrorInvalidValue;
    if constexpr (EPI == SK_STORE || EPI == SK_RESID || EPI == SK_SWIGLU) {
      const int blocks = EPI == SK_SWIGLU ? N / 32 * p : N / rs;
      if (p == 2) hipLaunchKernelGGL((skinny_kernel<1, EPI, 4, false, 0, 2>), dim3(blocks), dim3(256), 0, s, x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, packed, ls);
      else hipLaunchKernelGGL((skinny_kernel<1, EPI, 4, false, 0, 4>), dim3(blocks), dim3(256), 0, s, x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, packed, ls);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  const int ns = (EPI == SK_SWIGLU) ? 2 : (EPI == SK_ARGMAX && rt >= 2) ? 4 : 1;   // = NS of the kernel
  const int blocks = (N + 16 * ns - 1) / (16 * ns);
  // (non-temporal weight loads were measured SLOWER on the 8B decode shapes - wqkv 12.5 -> 14.8 us, w2 24.9 -> 29.9 us, w1|w3 47.9 -> 54.9 us -
  //  and the form was removed: plain cache policy everywhere)
#define GO(RT) hipLaunchKernelGGL((skinny_kernel<RT, EPI>), dim3(blocks), dim3(256), 0, s, x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, packed, ls)
  if constexpr (EPI == SK_STORE || EPI == SK_RESID) {
    constexpr int max8 = 256;   // at most one slab per CU
    if (rt == 1 && K % 256 == 0 && blocks <= max8 && p != 0) {   // a decode GEMV with about one slab per CU: 8 K slices per workgroup (p == 0: the caller wants ONE form for every row count)
      hipLaunchKernelGGL((skinny_kernel<1, EPI, 8>), dim3(blocks), dim3(512), 0, s, x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, packed, ls);
      return hipGetLastError();
    }
  }
  switch (rt) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    case 4: GO(4); break;
    default: return hipErrorInvalidValue;
  }
#undef GO
  return hipGetLastError();
}

}  // namespace

// widths the fused-norm GEMVs are built for (K = 2 x 2048: InternLM2-7B, 3 x 2048: InternLM2-20B)
bool aigv_skinny_norm_fusable(int K) { return K == 4096 || K == 6144; }

// the decode step's wqkv projection with RoPE + KV-cache append in the epilogue (SK_ROPE_KV above); x rows = one new token per sequence
hipError_t aigv_launch_skinny_rope_kv(const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K, bf16_t* qkv, int ldo,
                                      const int32_t* pos, const int32_t* seq, const bf16_t* cos, const bf16_t* sin, bf16_t* kc, bf16_t* vc,
                                      int g, int n_kv, int cap, int head_dim, hipStream_t s, const bf16_t* norm_w, float norm_eps, int p) {
  if (R <= 0) return hipSuccess;
  if (R > 64 || K % 128 || (ldx % 8) || (ldw % 8) || (ldo % 4) || head_dim != 128 || N != n_kv * (g + 2) * 128 || !pos || !seq || !cos ||
      !sin || !kc || !vc)
    return hipErrorInvalidValue;
  RopeKvArgs rk{pos, seq, cos, sin, kc, vc, g, n_kv, cap};
  const int rt = (R + 15) / 16, blocks = N / 32 * p;
  if ((p != 1 && p != 2 && p != 4) || R > (p == 1 ? 64 : 16 / p) || K % (128 * p)) return hipErrorInvalidValue;
  if (norm_w) {   // x = the raw residual rows; attention_norm applied by the kernel (NormArgs above)
    if (R > 4 || !aigv_skinny_norm_fusable(K)) return hipErrorInvalidValue;
#define GO(NC, PP) hipLaunchKernelGGL((skinny_kernel<1, SK_ROPE_KV, 4, false, NC, PP>), dim3(blocks), dim3(256), (size_t)R * K * sizeof(bf16_t), s, x, ldx, R, W, ldw, N, K, \
                                      nullptr, nullptr, 0, qkv, ldo, nullptr, nullptr, rk, NormArgs{norm_w, norm_eps})
    if (K == 4096) { if (p == 1) GO(2, 1); else if (p == 2) GO(2, 2); else GO(2, 4); }
    else { if (p == 1) GO(3, 1); else if (p == 2) GO(3, 2); else GO(3, 4); }
#undef GO
    return hipGetLastError();
  }
  if (p != 1) {
    if (p == 2) hipLaunchKernelGGL((skinny_kernel<1, SK_ROPE_KV, 4, false, 0, 2>), dim3(blocks), dim3(256), 0, s, x, ldx, R, W, ldw, N, K, nullptr, nullptr, 0, qkv, ldo, nullptr, nullptr, rk);
    else hipLaunchKernelGGL((skinny_kernel<1, SK_ROPE_KV, 4, false, 0, 4>), dim3(blocks), dim3(256), 0, s, x, ldx, R, W, ldw, N, K, nullptr, nullptr, 0, qkv, ldo, nullptr, nullptr, rk);
    return hipGetLastError();
  }
#define GO(RT) hipLaunchKernelGGL((skinny_kernel<RT, SK_ROPE_KV, 4, false>), dim3(blocks), dim3(256), 0, s, x, ldx, R, W, ldw, N, K, nullptr, nullptr, 0, qkv, ldo, nullptr, nullptr, rk)
  switch (rt) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    case 4: GO(4); break;
    default: return hipErrorInvalidValue;
  }
#undef GO
  return hipGetLastError();
}

// decode: SwiGLU(x_n W13^T) with x_n = RMSNorm(x) * norm_w computed by the kernel itself (R <= 4 rows; NormArgs above)
hipError_t aigv_launch_skinny_swiglu_normed(const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K, bf16_t* out, int ldo,
                                            const bf16_t* norm_w, float norm_eps, hipStream_t s, int p) {
  if (R <= 0) return hipSuccess;
  if (R > 4 || !aigv_skinny_norm_fusable(K) || (ldx % 8) || (ldw % 8) || (ldo % 4) || (N % 32) || !norm_w || (p != 1 && p != 2 && p != 4)) return hipErrorInvalidValue;
#define GO(NC, PP) hipLaunchKernelGGL((skinny_kernel<1, SK_SWIGLU, 4, false, NC, PP>), dim3(N / 32 * p), dim3(256), (size_t)R * K * sizeof(bf16_t), s, x, ldx, R, W, ldw, N, K, \
                                      nullptr, nullptr, 0, out, ldo, nullptr, nullptr, RopeKvArgs{}, NormArgs{norm_w, norm_eps})
  if (K == 4096) { if (p == 1) GO(2, 1); else if (p == 2) GO(2, 2); else GO(2, 4); }
  else { if (p == 1) GO(3, 1); else if (p == 2) GO(3, 2); else GO(3, 4); }
#undef GO
  return hipGetLastError();
}

// epi: 0 store(+bias) | 1 residual(+bias) | 2 swiglu (16-row interleaved w1/w3, out is N/2 wide) | 3 gelu(+bias)
//      6 layer-scale + residual (+bias)
hipError_t aigv_launch_skinny_gemm(const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K,
                                   const bf16_t* bias, const bf16_t* resid, int ldr, bf16_t* out, int ldo, int epi,
                                   hipStream_t s, const bf16_t* ls, int p) {
  if (R <= 0) return hipSuccess;
  if (R > 64 || K % 128 || (ldx % 8) || (ldw % 8) || (ldo % 4)) return hipErrorInvalidValue;
  if (epi != SK_SWIGLU && N % 4) return hipErrorInvalidValue;
  if (epi == SK_SWIGLU && N % 32) return hipErrorInvalidValue;
  if (p > 1 && epi != SK_STORE && epi != SK_RESID && epi != SK_SWIGLU) return hipErrorInvalidValue;
  switch (epi) {
    case SK_STORE: return launch_skinny<SK_STORE>(x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, nullptr, s, nullptr, p);
    case SK_RESID: return launch_skinny<SK_RESID>(x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, nullptr, s, nullptr, p);
    case SK_SWIGLU: return launch_skinny<SK_SWIGLU>(x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, nullptr, s, nullptr, p);
    case SK_GELU: return launch_skinny<SK_GELU>(x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, nullptr, s);
    case SK_LS_RESID:
      if (!ls || !resid) return hipErrorInvalidValue;
      return launch_skinny<SK_LS_RESID>(x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, nullptr, s, ls);
  }
  return hipErrorInvalidValue;
}

// lm-head logits of R rows as the reference holds them before its .float(): the bf16 output of the matmul (modeling_internlm2.py:
// 1095-1096).  out is [R, ldo] bf16 with ldo >= V rounded up to 4 (the store epilogue writes 4 columns per lane; columns >= V of a
// row are padding).  Used where the whole distribution is needed - sampling in generate() - the greedy paths use the fused argmax.
hipError_t aigv_launch_lm_head_logits(const bf16_t* h, int R, int H, const bf16_t* W, int V, bf16_t* out, int ldo, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  if (R > 64 || H % 128 || ldo < ((V + 3) / 4) * 4 || (ldo % 4)) return hipErrorInvalidValue;
  return launch_skinny<SK_STORE>(h, H, R, W, H, V, H, nullptr, nullptr, 0, out, ldo, nullptr, s);
}

hipError_t aigv_launch_lm_head_argmax(const bf16_t* h, int R, int H, const bf16_t* W, int V,
                                      unsigned long long* packed, int64_t* out_idx, float* out_val, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  if (R > 64 || H % 128) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(packed, 0, sizeof(unsigned long long) * R, s);
  if (e != hipSuccess) return e;
  e = launch_skinny<SK_ARGMAX>(h, H, R, W, H, V, H, nullptr, nullptr, 0, nullptr, 0, packed, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(unpack_argmax_kernel, dim3((R + 63) / 64), dim3(64), 0, s, packed, out_idx, out_val, R);
  return hipGetLastError();
}

// scratch: 3 * B * max(dims) bf16 (guarded input + two ping-pong activations)
hipError_t aigv_launch_score_head(const ScoreHeadArgs& a, bf16_t* scratch, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  if (a.n_layers < 1 || a.n_layers > 8 || a.B > 64 || !scratch) return hipErrorInvalidValue;
  int maxd = 0;
  for (int i = 0; i <= a.n_layers; ++i) {
    if (a.dims[i] <= 0) return hipErrorInvalidValue;
    maxd = a.dims[i] > maxd ? a.dims[i] : maxd;
  }
  bf16_t* xg = scratch;
  bf16_t* pp[2] = {scratch + (size_t)a.B * maxd, scratch + (size_t)2 * a.B * maxd};
  hipLaunchKernelGGL(score_guard_kernel, dim3(1), dim3(256), 0, s, a.x, a.ldx, a.B, a.dims[0], xg);
  const bf16_t* cur = xg;
  int ld = a.dims[0], L = 0, flip = 0;
  for (; L < a.n_layers; ++L) {
    const int din = a.dims[L], dout = a.dims[L + 1];
    if (din % 128 || dout % 4) break;
    hipError_t e = launch_skinny<SK_RELU>(cur, ld, a.B, a.w[L], din, dout, din, a.b[L], nullptr, 0, pp[flip], dout, nullptr, s);
    if (e != hipSuccess) return e;
    cur = pp[flip];
    ld = dout;
    flip ^= 1;
  }
  if (L < a.n_layers) {
    if (a.dims[L] > 1024) return hipErrorInvalidValue;
    for (int i = L + 1; i <= a.n_layers; ++i)
      if (a.dims[i] > 1024) return hipErrorInvalidValue;
    ScoreHeadArgs t = a;
    t.x = cur;
    t.ldx = ld;
    hipLaunchKernelGGL(score_tail_kernel, dim3(a.B), dim3(256), 0, s, t, L);
  } else {
    return hipErrorInvalidValue;   // the last layer (fan-out 1) always runs in the tail kernel
  }
  return hipGetLastError();
}
