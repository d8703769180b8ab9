// Row-wise and element-wise kernels of the scorer hot path (all HBM-bound: 16-byte vector accesses,
// fp32 statistics, bf16 rounding points as in the reference's eager path).
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXC = 8;  // 16-byte chunks per thread (256 threads) -> rows up to 16384 elements

// ---- LayerNorm -----------------------------------------------------------------------------------------
// reference: nn.LayerNorm in InternVisionEncoderLayer (modeling_intern_vit.py:208-209) and as the first
// stage of mlp1 / motion_mlp (modeling_internvl_chat.py:238-249).  fp32 statistics, one bf16 rounding.
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int ldx,
                                                        const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                        bf16_t* __restrict__ y, int ldy, int H, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int nchunk = H >> 3;
  float v[MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u16x8 raw = *(const u16x8*)(x + (size_t)row * ldx + (ch << 3));
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[c][e] = bf2f(raw[e]); sum += v[c][e]; }
    }
  }
  const float mean = block_sum_256(sum, red) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    if (threadIdx.x + c * 256 < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(block_sum_256(sq, red) / (float)H + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u16x8 ww = *(const u16x8*)(w + (ch << 3));
      const u16x8 bb = *(const u16x8*)(b + (ch << 3));
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf((v[c][e] - mean) * rstd * bf2f(ww[e]) + bf2f(bb[e]));
      *(u16x8*)(y + (size_t)row * ldy + (ch << 3)) = o;
    }
  }
}

// wave-per-row variant for rows of 512*CH elements (ViT hidden 1024 -> CH = 2): no LDS, no barriers, 4 rows per block
template <int CH>
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const bf16_t* __restrict__ x, int ldx,
                                                             const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                             bf16_t* __restrict__ y, int ldy, int rows, float eps) {
  constexpr int H = 512 * CH;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[CH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const u16x8 raw = *(const u16x8*)(x + (size_t)row * ldx + ((lane + c * 64) << 3));
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[c][e] = bf2f(raw[e]); sum += v[c][e]; }
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
  const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (lane + c * 64) << 3;
    const u16x8 ww = *(const u16x8*)(w + col);
    const u16x8 bb = *(const u16x8*)(b + col);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf((v[c][e] - mean) * rstd * bf2f(ww[e]) + bf2f(bb[e]));
    *(u16x8*)(y + (size_t)row * ldy + col) = o;
  }
}

// ---- RMSNorm: fp32 normalise -> bf16 -> * weight -> bf16 (modeling_internlm2.py:138-143) --------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, int ldx,
                                                      const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int ldy,
                                                      int H, float eps, const int32_t* __restrict__ row_idx) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int srow = row_idx ? row_idx[row] : row;
  const int nchunk = H >> 3;
  float v[MAXC][8];
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u16x8 raw = *(const u16x8*)(x + (size_t)srow * ldx + (ch << 3));
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[c][e] = bf2f(raw[e]); sq += v[c][e] * v[c][e]; }
    }
  }
  const float rstd = rsqrtf(block_sum_256(sq, red) / (float)H + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u16x8 ww = *(const u16x8*)(w + (ch << 3));
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(ww[e]) * rbf(v[c][e] * rstd));
      *(u16x8*)(y + (size_t)row * ldy + (ch << 3)) = o;
    }
  }
}

// RMSNorm fused with the fp8 row quantisation of its result (fp8 mode of the InternLM2 linears): the bf16 value the plain kernel
// would store is quantised from registers - bit-identical to rmsnorm_kernel followed by quant_fp8_rows_kernel, one pass instead of two.
__global__ __launch_bounds__(256) void rmsnorm_quant_fp8_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ w,
                                                                uint8_t* __restrict__ q, int ldq, float* __restrict__ scale, int H, float eps) {
  __shared__ float red[4];
  __shared__ float redm[4];
  const int row = blockIdx.x;
  const int nchunk = H >> 3;
  float v[MAXC][8];
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u16x8 raw = *(const u16x8*)(x + (size_t)row * ldx + (ch << 3));
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[c][e] = bf2f(raw[e]); sq += v[c][e] * v[c][e]; }
    }
  }
  const float rstd = rsqrtf(block_sum_256(sq, red) / (float)H + eps);
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u16x8 ww = *(const u16x8*)(w + (ch << 3));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = rbf(bf2f(ww[e]) * rbf(v[c][e] * rstd));
        amax = fmaxf(amax, fabsf(v[c][e]));
      }
    }
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 1.0f;
  if (threadIdx.x == 0) scale[row] = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
      typedef __attribute__((ext_vector_type(2))) int i32x2;
      *(i32x2*)(q + (size_t)row * ldq + (ch << 3)) = i32x2{lo, hi};
    }
  }
}

// ---- drop cls + pixel_shuffle v2 (modeling_internvl_chat.py:492-506,520-527): pre-projector tokens -------
// out token (i2, j2) of frame f = concat over (di, dj) of vit[f, 1 + (2*i2+di)*grid + 2*j2+dj, :]
__global__ void pixel_shuffle_kernel(const bf16_t* __restrict__ vit, int grid, int Hv, bf16_t* __restrict__ out,
                                     int rows) {
  const int cpr = (4 * Hv) >> 3;  // chunks per output row
  const size_t total = (size_t)rows * cpr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / cpr), ch = (int)(i - (size_t)row * cpr);
    const int g2 = grid >> 1, per = g2 * g2;
    const int f = row / per, t = row - f * per, i2 = t / g2, j2 = t - i2 * g2;
    const int col = ch << 3, part = col / Hv, di = part >> 1, dj = part & 1;
    const size_t tok = (size_t)f * (grid * grid + 1) + 1 + (size_t)(2 * i2 + di) * grid + (2 * j2 + dj);
    *(u16x8*)(out + (size_t)row * 4 * Hv + col) = *(const u16x8*)(vit + tok * Hv + (col - part * Hv));
  }
}

// ---- im2col for Conv2d(C, Hv, k=P, s=P) (modeling_intern_vit.py:78-80,97) -------------------------------
// out[(f*g + py)*g + px][c*P*P + iy*P + ix] = frames[f][c][py*P + iy][px*P + ix]; columns >= C*P*P are 0.
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ frames, int C, int S, int P, int Kp,
                                                     bf16_t* __restrict__ out) {
  const int g = S / P;
  const int patch = blockIdx.x;            // f*g*g + py*g + px
  const int f = patch / (g * g), r = patch - f * g * g, py = r / g, px = r - py * g;
  const int K = C * P * P;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    bf16_t val = 0;
    if (k < K) {
      const int c = k / (P * P), rr = k - c * P * P, iy = rr / P, ix = rr - iy * P;
      val = frames[(((size_t)f * C + c) * S + (py * P + iy)) * S + (px * P + ix)];
    }
    out[(size_t)patch * Kp + k] = val;
  }
}

__global__ void cls_rows_kernel(const bf16_t* __restrict__ cls_pos, bf16_t* __restrict__ x, int tokens_per_frame,
                                int H) {
  const int f = blockIdx.x;
  for (int c = threadIdx.x; c < (H >> 3); c += blockDim.x)
    *(u16x8*)(x + (size_t)f * tokens_per_frame * H + (c << 3)) = *(const u16x8*)(cls_pos + (c << 3));
}

// dst[i, :] = src[idx[i], :]  (compact copies of the few rows the last decoder layer still has to finish)
__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, int ld, const int32_t* __restrict__ idx,
                                   bf16_t* __restrict__ dst, int H) {
  const int i = blockIdx.x;
  const bf16_t* row = src + (size_t)idx[i] * ld;
  for (int c = threadIdx.x; c < (H >> 3); c += blockDim.x) *(u16x8*)(dst + (size_t)i * H + (c << 3)) = *(const u16x8*)(row + (c << 3));
}

// dst[idx[i], :] = src[i, :]  (the inverse: finished rows back to their places; duplicate indices carry identical rows)
__global__ void scatter_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ idx, bf16_t* __restrict__ dst, int ld, int H) {
  const int i = blockIdx.x;
  bf16_t* row = dst + (size_t)idx[i] * ld;
  for (int c = threadIdx.x; c < (H >> 3); c += blockDim.x) *(u16x8*)(row + (c << 3)) = *(const u16x8*)(src + (size_t)i * H + (c << 3));
}

// ---- RoPE in place (modeling_internlm2.py:247-261): out = bf16(bf16(x*cos) + bf16(rot(x)*sin)) ---------
// qkv row layout: n_groups x slots_per_group x D; slots [0, n_rot) of every group are rotated (q heads + K).
// cos/sin tables are [max_pos, D/2] bf16 (the second half of the reference's table repeats the first).
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, int ld, const int32_t* __restrict__ pos,
                                                   const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                                                   int tokens, int n_rot, int slots, int n_groups, int D, int first_rot) {
  const int half = D >> 1, cph = half >> 3;            // 16-byte chunks per half head
  const int per_tok = n_groups * n_rot * cph;
  const size_t total = (size_t)tokens * per_tok;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / per_tok);
    int r = (int)(i - (size_t)t * per_tok);
    const int c = r % cph; r /= cph;
    const int s = r % n_rot, gidx = r / n_rot;
    bf16_t* base = qkv + (size_t)t * ld + (size_t)(gidx * slots + first_rot + s) * D + (c << 3);
    const size_t tb = (size_t)pos[t] * half + (c << 3);
    const u16x8 lo = *(const u16x8*)base, hi = *(const u16x8*)(base + half);
    const u16x8 co = *(const u16x8*)(cs + tb), si = *(const u16x8*)(sn + tb);
    u16x8 olo, ohi;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x1 = bf2f(lo[e]), x2 = bf2f(hi[e]), cc = bf2f(co[e]), ss = bf2f(si[e]);
      olo[e] = f2bf(rbf(x1 * cc) + rbf(-x2 * ss));
      ohi[e] = f2bf(rbf(x2 * cc) + rbf(x1 * ss));
    }
    *(u16x8*)base = olo;
    *(u16x8*)(base + half) = ohi;
  }
}

// ---- token embedding gather + visual / motion scatter (modeling_internvl_chat.py:324,351-378) ------------
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ slot,
                                                    const bf16_t* __restrict__ emb, const bf16_t* __restrict__ vis,
                                                    const bf16_t* __restrict__ motion, int n_vis,
                                                    bf16_t* __restrict__ out, int H) {
  const int t = blockIdx.x;
  const int s = slot[t];
  const bf16_t* src = s < 0 ? emb + (size_t)ids[t] * H : (s < n_vis ? vis + (size_t)s * H : motion + (size_t)(s - n_vis) * H);
  for (int c = threadIdx.x; c < (H >> 3); c += 256)
    *(u16x8*)(out + (size_t)t * H + (c << 3)) = *(const u16x8*)(src + (c << 3));
}

// ---- KV cache append -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_store_kernel(const bf16_t* __restrict__ qkv, int ld,
                                                       const int32_t* __restrict__ seq_of_tok,
                                                       const int32_t* __restrict__ pos, bf16_t* __restrict__ kc,
                                                       bf16_t* __restrict__ vc, int n_groups, int g, int D, int cap) {
  const int t = blockIdx.x;
  const int sq = seq_of_tok[t], p = pos[t];
  const int cpd = D >> 3;
  for (int i = threadIdx.x; i < n_groups * 2 * cpd; i += 256) {
    const int c = i % cpd, r = i / cpd, which = r & 1, h = r >> 1;
    const bf16_t* src = qkv + (size_t)t * ld + (size_t)(h * (g + 2) + g + which) * D + (c << 3);
    bf16_t* dst = (which ? vc : kc) + (((size_t)sq * n_groups + h) * cap + p) * D + (c << 3);
    *(u16x8*)dst = *(const u16x8*)src;
  }
}

// Decode step: RoPE of the new token's q heads and K in place, then K / V appended to the cache - one launch instead of two
// (a decode layer is a chain of small launches; each costs its latency).  One workgroup per token; chunk c of slot s.
__global__ __launch_bounds__(256) void rope_kv_store_kernel(bf16_t* __restrict__ qkv, int ld, const int32_t* __restrict__ seq_of_tok,
                                                            const int32_t* __restrict__ pos, const bf16_t* __restrict__ cs,
                                                            const bf16_t* __restrict__ sn, bf16_t* __restrict__ kc,
                                                            bf16_t* __restrict__ vc, int n_groups, int g, int D, int cap) {
  const int t = blockIdx.x;
  const int sq = seq_of_tok[t], p = pos[t];
  const int half = D >> 1, cph = half >> 3, slots = g + 2;
  // rotated slots: q heads 0..g-1 and K (slot g); V (slot g+1) is copied as is
  for (int i = threadIdx.x; i < n_groups * (g + 1) * cph; i += 256) {
    const int c = i % cph, r = i / cph, s = r % (g + 1), gi = r / (g + 1);
    bf16_t* base = qkv + (size_t)t * ld + (size_t)(gi * slots + s) * D + (c << 3);
    const size_t tb = (size_t)p * half + (c << 3);
    const u16x8 lo = *(const u16x8*)base, hi = *(const u16x8*)(base + half);
    const u16x8 co = *(const u16x8*)(cs + tb), si = *(const u16x8*)(sn + tb);
    u16x8 olo, ohi;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x1 = bf2f(lo[e]), x2 = bf2f(hi[e]), cc = bf2f(co[e]), ss = bf2f(si[e]);
      olo[e] = f2bf(rbf(x1 * cc) + rbf(-x2 * ss));
      ohi[e] = f2bf(rbf(x2 * cc) + rbf(x1 * ss));
    }
    if (s < g) {
      *(u16x8*)base = olo;
      *(u16x8*)(base + half) = ohi;
    } else {   // K: straight into the cache (the qkv row's K slot is not read again in a decode step)
      bf16_t* dst = kc + (((size_t)sq * n_groups + gi) * cap + p) * D + (c << 3);
      *(u16x8*)dst = olo;
      *(u16x8*)(dst + half) = ohi;
    }
  }
  for (int i = threadIdx.x; i < n_groups * (D >> 3); i += 256) {
    const int c = i % (D >> 3), gi = i / (D >> 3);
    *(u16x8*)(vc + (((size_t)sq * n_groups + gi) * cap + p) * D + (c << 3)) =
        *(const u16x8*)(qkv + (size_t)t * ld + (size_t)(gi * slots + g + 1) * D + (c << 3));
  }
}

// ---- fp8 row quantisation (groundwork for BASELINE config 5; not on the bf16 scoring path) ------------------------------------
// One workgroup per row: amax over the row, scale = amax / 448 (e4m3 max), q = e4m3_rne(x * (448 / amax)) - OCP e4m3 (gfx950's
// v_cvt_pk_fp8_f32), the same three fp32 operations as the torch oracle in tests/test_gpu_ops.py; an all-zero row gets scale 1.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ x, int ldx, int K, uint8_t* __restrict__ q,
                                                             int ldq, float* __restrict__ scale) {
  __shared__ float red[4];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* xr = x + (size_t)row * ldx;
  float amax = 0.f;
  for (int c = threadIdx.x; c < (K >> 3); c += 256) {
    const u16x8 v = *(const u16x8*)(xr + (c << 3));
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(v[e])));
  }
  amax = wave_max(amax);
  if (lane == 0) red[wave] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 1.0f;
  if (threadIdx.x == 0) scale[row] = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
  uint8_t* qr = q + (size_t)row * ldq;
  for (int c = threadIdx.x; c < (K >> 3); c += 256) {
    const u16x8 v = *(const u16x8*)(xr + (c << 3));
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[0]) * inv, bf2f(v[1]) * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[2]) * inv, bf2f(v[3]) * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[4]) * inv, bf2f(v[5]) * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[6]) * inv, bf2f(v[7]) * inv, hi, true);
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    *(i32x2*)(qr + (c << 3)) = i32x2{lo, hi};
  }
}

// Host bookkeeping arrays travel as KERNEL ARGUMENTS (<= 4 KB), not as memcpys: a pageable hipMemcpyAsync makes the
// host wait for the stream, which would stop the CPU from running ahead of the GPU.
__global__ __launch_bounds__(256) void seqpos_kernel(const SmallInts cu, int n_seq, int32_t* __restrict__ pos,
                                                     int32_t* __restrict__ seq, int32_t* __restrict__ cu_out, int tokens) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t <= n_seq) cu_out[t] = cu.v[t];
  if (t >= tokens) return;
  int b = 0;
  while (b + 1 < n_seq && t >= cu.v[b + 1]) ++b;
  pos[t] = t - cu.v[b] + cu.v[AIGV_SMALL_INTS / 2 + b];   // second half of the argument: per-sequence position offsets
  seq[t] = b;
}

__global__ void write_ints_kernel(const SmallInts src, int n, int32_t* __restrict__ dst) {
  const int i = threadIdx.x;
  if (i < n) dst[i] = src.v[i];
}

__global__ void advance_kernel(int32_t* a, int32_t* b, int n) {
  const int i = threadIdx.x;
  if (i < n) { a[i] += 1; b[i] += 1; }
}

// End-of-sequence bookkeeping of a greedy / sampling loop, on the device (HF's loop: next = next * unfinished + pad * (1 - unfinished);
// unfinished &= next not in eos; stop when no sequence is unfinished): tok[b] in: the step's raw token, out: the token the loop
// emits; state[b] = 1 once sequence b has emitted an end token; state[n] counts the emitted columns that still held a live sequence
// (= the length HF's loop returns).  One wave (n <= 64).
struct EosIds { long long v[8]; int n; };
__global__ __launch_bounds__(64) void decode_eos_kernel(long long* __restrict__ tok, int32_t* __restrict__ state, int n, const EosIds eos,
                                                        long long pad) {
  const int i = threadIdx.x;
  const bool in = i < n;
  const int was_done = in ? state[i] : 1;
  const bool live = __any(in && !was_done);
  if (in) {
    const long long t = was_done ? pad : tok[i];
    bool hit = false;
    for (int k = 0; k < eos.n; ++k) hit |= (t == eos.v[k]);
    tok[i] = t;
    state[i] = was_done | (int)hit;
  }
  if (i == 0 && live) state[n] += 1;
}

}  // namespace

hipError_t aigv_launch_seqpos(const int32_t* cu_host, int n_seq, int32_t* pos, int32_t* seq, int32_t* cu_dev, int tokens,
                              hipStream_t s, const int32_t* pos_offset_host) {
  if (n_seq <= 0 || n_seq + 1 > AIGV_SMALL_INTS / 2 || tokens <= 0) return hipErrorInvalidValue;
  SmallInts a{};
  for (int i = 0; i <= n_seq; ++i) a.v[i] = cu_host[i];
  if (pos_offset_host)
    for (int i = 0; i < n_seq; ++i) a.v[AIGV_SMALL_INTS / 2 + i] = pos_offset_host[i];
  hipLaunchKernelGGL(seqpos_kernel, dim3((tokens + 255) / 256), dim3(256), 0, s, a, n_seq, pos, seq, cu_dev, tokens);
  return hipGetLastError();
}

// Beam search: slot i of the NEW cache takes the first len[i] positions of slot parent[i] of the current one, for every layer and kv head
// (dst and src are different buffers: a gather, never in place).  grid (16-byte chunks of a head's live region, kv heads, layers * n).
namespace {
__global__ __launch_bounds__(256) void kv_reorder_kernel(const bf16_t* __restrict__ sk, const bf16_t* __restrict__ sv, bf16_t* __restrict__ dk,
                                                         bf16_t* __restrict__ dv, const int32_t* __restrict__ parent,
                                                         const int32_t* __restrict__ lens, int n, int nkv, int cap, int D, size_t kv_layer) {
  const int layer = blockIdx.z / n, slot = blockIdx.z % n, head = blockIdx.y;
  const int src = parent[slot];
  const size_t live = (size_t)lens[slot] * D / 8;          // 16-byte chunks to copy (D % 8 == 0)
  const size_t so = (size_t)layer * kv_layer + ((size_t)src * nkv + head) * cap * D;
  const size_t dO = (size_t)layer * kv_layer + ((size_t)slot * nkv + head) * cap * D;
  const uint4* k0 = (const uint4*)(sk + so);
  const uint4* v0 = (const uint4*)(sv + so);
  uint4* k1 = (uint4*)(dk + dO);
  uint4* v1 = (uint4*)(dv + dO);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < live; i += (size_t)gridDim.x * blockDim.x) {
    k1[i] = k0[i];
    v1[i] = v0[i];
  }
}
}  // namespace

hipError_t aigv_launch_kv_reorder(const bf16_t* sk, const bf16_t* sv, bf16_t* dk, bf16_t* dv, const int32_t* parent, const int32_t* lens, int n,
                                  int layers, int nkv, int cap, int D, size_t kv_layer, int max_len, hipStream_t s) {
  if (n <= 0 || layers <= 0 || nkv <= 0 || cap <= 0 || D % 8 || max_len <= 0 || max_len > cap || !sk || !sv || !dk || !dv || !parent || !lens ||
      sk == dk || sv == dv || (long)layers * n > 65535)
    return hipErrorInvalidValue;
  const size_t chunks = (size_t)max_len * D / 8;
  const int gx = (int)std::min<size_t>((chunks + 255) / 256, 64);
  hipLaunchKernelGGL(kv_reorder_kernel, dim3(gx, nkv, layers * n), dim3(256), 0, s, sk, sv, dk, dv, parent, lens, n, nkv, cap, D, kv_layer);
  return hipGetLastError();
}

hipError_t aigv_launch_write_ints(const int32_t* host, int n, int32_t* dst, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  for (int off = 0; off < n; off += AIGV_SMALL_INTS) {
    SmallInts a{};
    const int m = n - off < AIGV_SMALL_INTS ? n - off : AIGV_SMALL_INTS;
    for (int i = 0; i < m; ++i) a.v[i] = host[off + i];
    hipLaunchKernelGGL(write_ints_kernel, dim3(1), dim3(AIGV_SMALL_INTS), 0, s, a, m, dst + off);
  }
  return hipGetLastError();
}

hipError_t aigv_launch_decode_eos(int64_t* tok, int32_t* state, int n, const int64_t* eos_host, int n_eos, int64_t pad, hipStream_t s) {
  if (n <= 0 || n > 64 || n_eos < 0 || n_eos > 8 || !tok || !state || (n_eos > 0 && !eos_host)) return hipErrorInvalidValue;
  EosIds e{};
  e.n = n_eos;
  for (int k = 0; k < n_eos; ++k) e.v[k] = (long long)eos_host[k];
  hipLaunchKernelGGL(decode_eos_kernel, dim3(1), dim3(64), 0, s, (long long*)tok, state, n, e, (long long)pad);
  return hipGetLastError();
}

hipError_t aigv_launch_advance(int32_t* a, int32_t* b, int n, hipStream_t s) {
  if (n <= 0 || n > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1024), 0, s, a, b, n);
  return hipGetLastError();
}

hipError_t aigv_launch_layernorm(const bf16_t* x, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* y, int ldy,
                                 int rows, int H, float eps, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (H % 8 || H > MAXC * 256 * 8) return hipErrorInvalidValue;
  if (H == 1024) hipLaunchKernelGGL(layernorm_wave_kernel<2>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, w, b, y, ldy, rows, eps);
  else if (H == 2048) hipLaunchKernelGGL(layernorm_wave_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, w, b, y, ldy, rows, eps);
  else hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, s, x, ldx, w, b, y, ldy, H, eps);
  return hipGetLastError();
}

hipError_t aigv_launch_rmsnorm(const bf16_t* x, int ldx, const bf16_t* w, bf16_t* y, int ldy, int rows, int H,
                               float eps, const int32_t* row_idx, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (H % 8 || H > MAXC * 256 * 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, s, x, ldx, w, y, ldy, H, eps, row_idx);
  return hipGetLastError();
}

hipError_t aigv_launch_rmsnorm_quant_fp8(const bf16_t* x, int ldx, const bf16_t* w, uint8_t* q, int ldq, float* scale, int rows, int H, float eps,
                                         hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (H % 8 || H > MAXC * 256 * 8 || ldq % 8 || !q || !scale) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rmsnorm_quant_fp8_kernel, dim3(rows), dim3(256), 0, s, x, ldx, w, q, ldq, scale, H, eps);
  return hipGetLastError();
}

hipError_t aigv_launch_pixel_shuffle(const bf16_t* vit, int grid, int Hv, bf16_t* out, int frames, hipStream_t s) {
  if (frames <= 0) return hipSuccess;
  if (Hv % 8 || grid % 2) return hipErrorInvalidValue;
  const int rows = frames * (grid / 2) * (grid / 2);
  hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(2048), dim3(256), 0, s, vit, grid, Hv, out, rows);
  return hipGetLastError();
}

hipError_t aigv_launch_im2col(const bf16_t* frames, int F, int C, int S, int P, int Kp, bf16_t* out, hipStream_t s) {
  if (F <= 0) return hipSuccess;
  if (S % P || Kp < C * P * P) return hipErrorInvalidValue;
  const int g = S / P;
  hipLaunchKernelGGL(im2col_kernel, dim3(F * g * g), dim3(256), 0, s, frames, C, S, P, Kp, out);
  return hipGetLastError();
}

hipError_t aigv_launch_cls_rows(const bf16_t* cls_pos, bf16_t* x, int F, int tokens_per_frame, int H, hipStream_t s) {
  if (F <= 0) return hipSuccess;
  hipLaunchKernelGGL(cls_rows_kernel, dim3(F), dim3(128), 0, s, cls_pos, x, tokens_per_frame, H);
  return hipGetLastError();
}

hipError_t aigv_launch_gather_rows(const bf16_t* src, int ld, const int32_t* idx, int n, bf16_t* dst, int H, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (H % 8 || ld % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, s, src, ld, idx, dst, H);
  return hipGetLastError();
}

hipError_t aigv_launch_scatter_rows(const bf16_t* src, const int32_t* idx, int n, bf16_t* dst, int ld, int H, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (H % 8 || ld % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, s, src, idx, dst, ld, H);
  return hipGetLastError();
}

hipError_t aigv_launch_rope(bf16_t* qkv, int ld, const int32_t* pos, const bf16_t* cos, const bf16_t* sin,
                            int tokens, int n_rot, int slots, int n_groups, int D, hipStream_t s, int first_rot) {
  if (tokens <= 0 || n_rot <= 0) return hipSuccess;
  if (D % 16 || first_rot < 0 || first_rot + n_rot > slots) return hipErrorInvalidValue;
  const size_t total = (size_t)tokens * n_groups * n_rot * (D / 16);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(rope_kernel, dim3(blocks), dim3(256), 0, s, qkv, ld, pos, cos, sin, tokens, n_rot, slots,
                     n_groups, D, first_rot);
  return hipGetLastError();
}

hipError_t aigv_launch_quant_fp8_rows(const bf16_t* x, int ldx, int rows, int K, uint8_t* q, int ldq, float* scale, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (K <= 0 || K % 8 || ldx % 8 || ldq % 8 || !x || !q || !scale) return hipErrorInvalidValue;
  hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3(rows), dim3(256), 0, s, x, ldx, K, q, ldq, scale);
  return hipGetLastError();
}

hipError_t aigv_launch_rope_kv_store(bf16_t* qkv, int ld, const int32_t* seq_of_tok, const int32_t* pos, const bf16_t* cos,
                                     const bf16_t* sin, bf16_t* kc, bf16_t* vc, int tokens, int n_groups, int g, int D, int cap,
                                     hipStream_t s) {
  if (tokens <= 0) return hipSuccess;
  if (D % 16) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rope_kv_store_kernel, dim3(tokens), dim3(256), 0, s, qkv, ld, seq_of_tok, pos, cos, sin, kc, vc, n_groups, g, D, cap);
  return hipGetLastError();
}

hipError_t aigv_launch_embed(const int64_t* ids, const int32_t* slot, const bf16_t* emb, const bf16_t* vis,
                             const bf16_t* motion, int n_vis, bf16_t* out, int tokens, int H, hipStream_t s) {
  if (tokens <= 0) return hipSuccess;
  if (H % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_kernel, dim3(tokens), dim3(256), 0, s, ids, slot, emb, vis, motion, n_vis, out, H);
  return hipGetLastError();
}

hipError_t aigv_launch_kv_store(const bf16_t* qkv, int ld, const int32_t* seq_of_tok, const int32_t* pos,
                                bf16_t* kc, bf16_t* vc, int tokens, int n_groups, int g, int D, int cap,
                                hipStream_t s) {
  if (tokens <= 0) return hipSuccess;
  hipLaunchKernelGGL(kv_store_kernel, dim3(tokens), dim3(256), 0, s, qkv, ld, seq_of_tok, pos, kc, vc, n_groups, g, D,
                     cap);
  return hipGetLastError();
}
