// Fused (flash-style) attention forward for gfx950, bf16 in / fp32 accumulate, packed varlen layout.
//   D = 64  non-causal : InternViT multi-head attention   (reference: modeling_intern_vit.py:143-160 / flash :162-177)
//   D = 128 causal GQA : InternLM2 prefill attention       (reference: modeling_internlm2.py:355-440 / flash :444-614)
//
// Work split: grid = (q blocks of 128 rows, q heads, sequences); 4 waves per workgroup, each wave owns
// 32 query rows; the workgroup streams 64-key K/V tiles through a double-buffered LDS ring filled by LDS-DMA
// (global_load_lds, 16 B/lane): tile t+1 is in flight while tile t is multiplied; one barrier per tile.
//
// Non-causal key counts 64 j + 1 (InternViT: cls + 1024 patches), opt-in: the loop covers keys 1.. in full tiles, key 0 is merged in the epilogue ("lead key").
//
// MFMA formulation ("key on the row, query on the lane"):
//   S^T[key, q] = K · Q^T      v_mfma_f32_32x32x16_bf16, A = K rows from LDS (ds_read_b128, XOR-swizzled),
//                              B = Q^T fragments held in registers for the whole kernel.
//     -> each lane owns ONE query column, so softmax statistics are lane-local (+ one lane^32 exchange)
//   O^T[d, q] += V^T · P^T     the S^T accumulator (converted to bf16) IS the B operand of this product
//                              (no LDS round trip); A = V^T read with ds_read_b64_tr_b16 from the row-major
//                              V tile, in the accumulator's permuted key order
//                              (key = 16s + 8(j>>2) + 4h + (j&3), guide §3 "accumulator tile as operand").
//
// Numerics: the q pre-scale rounds to bf16 (exact for d^-1/2 = 2^-3); the score matrix stays fp32 up to the softmax
// (aigv_set_attention_numerics 0, the default since round 5) or rounds to bf16 where the reference's eager path rounds it
// (RS, mode 1: once after q k^T, InternLM2 again after / sqrt(d)); the softmax runs in fp32, P rounds to bf16 before P·V (un-normalised; the row is divided at
// the end), the output rounds to bf16.  With RS the kernel sits ~4x closer to the eager bf16 result than that result sits to
// fp64 truth; without it, it is at least as accurate against fp64 truth as the eager path (tests/test_gpu_ops.py).
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "kernels.h"
#include <type_traits>
#include "attn_lay.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int KT = 64;    // keys per tile

// Exchange between lane l and lane l ^ 32 (the two halves of a query's key range) as ONE VALU instruction (v_permlane32_swap: both halves
// of the wave receive the low half's and the high half's value) - __shfl_xor(x, 32) compiles to ds_bpermute_b32, an LDS round trip.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ float xhalf_max(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Diagnostic build only (-DAIGV_ATTN_STAMP, scripts/attn_stamp.py): where a wave's cycles go, summed over all waves of all launches.
// Slots per head dim (0: d = 64, 1: d = 128): 0 waves, 1 total, 2 prologue (Q fragments, first DMA), 3 wait for the tile + barrier,
// 4 DMA issue, 5 S^T = K Q^T incl. fragment reads and the row maximum, 6 softmax + P V, 7 epilogue, 8 tiles computed.  The stamps
// (s_memtime, an lgkmcnt wait each) perturb the schedule by ~10 %: read shares, not absolute times.  Never defined in the product build.
#ifdef AIGV_ATTN_STAMP
constexpr int STAMP_REPL = 2048;     // replicas of the accumulators, indexed by workgroup: a single hot address would serialise the atomics
__device__ unsigned long long g_attn_stamp[2][STAMP_REPL][16];
#define STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define STAMP_ADD(slot, a, b) do { if (lane == 0) st_acc[slot] += (b) - (a); } while (0)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

// NW = waves per workgroup (32 query rows each).  More waves share one K/V tile: the LDS-DMA issue cost per wave and tile
// (the dominant overhead next to the MFMAs) halves going from 4 to 8 waves.
// NB = K/V ring depth: tile t is multiplied while tiles t+1 .. t+NB-2 are in flight behind a counted vmcnt.
// RS = the reference's rounding points of the SCORE matrix (AttnArgs::round_scores): s1 = bf16(q k^T) and, where the division that
// follows is not a power of two (InternLM2: / sqrt(128)), s2 = bf16(s1 / post_div); the softmax then runs in fp32 on those values.
template <int D, bool CAUSAL, int NW, int NB, bool RS>
__global__ __launch_bounds__(NW * 64, (D == 64 && NW == 4) ? 4 : 2) void attn_fwd_kernel(const AttnArgs p) {
  constexpr int QB = NW * 32;                // query rows per workgroup
  constexpr int ROWB = Lay<D>::ROWB;
  constexpr int CPR = D / 8;                 // 16-byte chunks per row
  constexpr int NKS = D / 16;                // k-steps of the S^T product
  constexpr int NDT = D / 32;                // 32-row tiles of O^T
  // ring of NB K/V buffers filled by LDS-DMA (no staging registers, no ds_write pass)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
#ifdef AIGV_ATTN_STAMP
  unsigned long long st_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  STAMP(t_begin);
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs, so the linear id is remapped (bijectively) to
  // give each XCD a contiguous run of (sequence, head, query block) - the query blocks of one head, and the heads of one GQA
  // group, then stream the same K/V through ONE L2 instead of eight.
  // a launch restricted to the query rows >= q_begin (the tail of a row range split between the two kernels) spans only those blocks
  const int qb0 = p.q_begin / QB;
  const int nqb = (p.max_len + QB - 1) / QB - qb0;
  int v;
  {
    const int total = (int)gridDim.x, bid = blockIdx.x, xcd = bid & 7, q8 = total >> 3, r8 = total & 7;
    v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  int qb = v % nqb, grp = v / nqb;
  if (CAUSAL && p.n_heads > p.n_kv_heads) {
    // Causal GQA (InternLM2): work grows with the query block, and an XCD's run of workgroups is dispatched in order - with all blocks of
    // one head in front of the next head's, the launch ends on the LAST heads' heaviest blocks (list-scheduling a clip's 16 heads x 17
    // blocks on an XCD's 64 slots: 108 tile-times against 89 ideal).  Order inside the run: one GQA group at a time (its heads read the
    // same K/V: the L2 sharing stays), the group's query blocks heaviest first, the group's heads side by side -> 93 tile-times.
    const int gq = p.n_heads / p.n_kv_heads, per = nqb * gq;
    const int t = v / per, r = v - t * per;
    qb = r / gq;
    grp = t * gq + (r - qb * gq);
  }
  if (qb0 == 0 && !CAUSAL && (p.max_len % QB) != 0 && (p.max_len % QB) <= 32 && nqb > 1 && ((int)gridDim.x % (8 * nqb)) == 0) {
    // Non-causal with a nearly empty last query block per head (ViT: 1025 rows): each XCD runs its full blocks first and the
    // cheap ragged ones (key-split below) at the end, so the launch drains on short workgroups instead of on full ones.
    const int chunk = (int)gridDim.x >> 3, gpc = chunk / nqb, j = blockIdx.x >> 3, x = blockIdx.x & 7;
    const int full = gpc * (nqb - 1);
    const int gl = j < full ? j / (nqb - 1) : j - full;
    qb = j < full ? j % (nqb - 1) : nqb - 1;
    grp = x * gpc + gl;
  }
  const int hq = grp % p.n_heads, seq = grp / p.n_heads;
  const int g = p.n_heads / p.n_kv_heads;
  const int hk = hq / g;
  const int row0 = p.cu[seq];
  const int len = p.cu[seq + 1] - row0;          // queries (= keys appended this call)
  // causal work grows with the query block index: launch the heaviest blocks first
  const int q0 = (qb0 + (CAUSAL ? nqb - 1 - qb : qb)) * QB;
  if (q0 >= len) return;
  if (p.q_tail > 0 && q0 + QB <= len - p.q_tail) return;   // this block's rows are not consumed (last-layer row trimming)
  const int kv_off = p.kv_off ? p.kv_off[seq] : p.kv_len_offset;   // keys in front of this sequence's first query row
  const int kv_all = len + kv_off;                // keys visible in total (plain prefill: offset 0)
  // Lead key (non-causal, key count = 64 j + 1: InternViT's 1025 = cls + 1024 patches): the tile loop would spend a whole masked tile on the
  // one left-over key - 1 / 17 of a workgroup's work.  With AttnArgs::lead_key the loop runs over keys 1.. in full, unmasked tiles and key 0 is
  // merged at the end like one more softmax state (a dot product, two exp2 and a scaled row add per query row, in the epilogue where the loop's
  // registers are free).  OPT-IN: -2 % on the InternViT shape in-step and statistically neutral on the 13 reference-recorded clips (mean 2.96
  // against 3.19 bf16 ulps from the reference's scores), but another fp32 summation order re-rolls each clip's rounding noise, and the recorded
  // per-batch numbers of the default path are kept as they are (profiles/r4_attn_stagger_negative.txt, 6).
  const int lead = (!CAUSAL && p.lead_key && kv_off == 0 && (kv_all % KT) == 1) ? 1 : 0;
  const int kv_len = kv_all - lead;               // keys the tile loop covers (K / V tile bases advance by `lead` rows)
  // Ragged last block of a NON-causal sequence with at most 32 rows (ViT: 1025 = 8 x 128 + 1): instead of one wave doing the
  // whole key range for those rows while three idle - the block would last as long as a full one - all waves take the SAME
  // query slice and every NW-th key tile each; their softmax states are merged through LDS at the end (key split).
  const bool ksplit = !CAUSAL && NW == 4 && p.q_tail == 0 && (len - q0) <= 32;   // (NW = 4: three published states fit the ring)
  const int qw = ksplit ? q0 : q0 + wave * 32;    // first query row of this wave

  // ---- Q^T fragments: lane (c,h) holds Q[qw+c][16*ks + 8h + j] --------------------------------------
  // Prologue order: the query rows (and their rotary rows) are REQUESTED first, then the first K/V tiles' LDS-DMA goes out, and only
  // then the query math runs - it waits for the loads with the DMA already in flight behind them (one memory latency instead of two).
  bf16x8 qf[NKS];
  const int qr0 = qw + c;
  const bool q_ok = qr0 < len;
  u16x8 raw[NKS], rco[NKS / 2], rsi[NKS / 2];
  {
    const bf16_t* qp = p.q + (size_t)(row0 + (q_ok ? qr0 : 0)) * p.ldq + (size_t)(hq / g) * p.q_group_stride + (hq % g) * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) raw[ks] = *(const u16x8*)(qp + 16 * ks + 8 * h);
    if (p.rope_cos) {
      // rotate_half pairs dimension i with i + D/2: k-steps ks and ks + NKS/2 of the same lane (modeling_internlm2.py:247-261)
      const int kv0 = p.kv_off ? p.kv_off[seq] : p.kv_len_offset;
      const size_t tb = (size_t)(p.rope_pos_is_row ? kv0 + (q_ok ? qr0 : 0) : p.rope_pos[row0 + (q_ok ? qr0 : 0)]) * (D / 2);
#pragma unroll
      for (int ks = 0; ks < NKS / 2; ++ks) {
        rco[ks] = *(const u16x8*)(p.rope_cos + tb + 16 * ks + 8 * h);
        rsi[ks] = *(const u16x8*)(p.rope_sin + tb + 16 * ks + 8 * h);
      }
    }
  }

  // ---- tile range ----------------------------------------------------------------------------------
  int n_tiles = (kv_len + KT - 1) / KT;
  if (CAUSAL) {
    const int last_q = min(q0 + QB, len) - 1 + kv_off;   // last visible key index of the block
    n_tiles = min(n_tiles, last_q / KT + 1);
  }
  const size_t kv_seq = p.kv_seq_stride ? (size_t)seq * p.kv_seq_stride : 0;
  const bf16_t* k_row0 = p.k + (p.kv_seq_stride ? kv_seq : (size_t)row0 * p.ldk) + (size_t)hk * p.kv_head_stride;
  const bf16_t* v_row0 = p.v + (p.kv_seq_stride ? kv_seq : (size_t)row0 * p.ldv) + (size_t)hk * p.kv_head_stride;
  const bf16_t* kbase = k_row0 + (size_t)lead * p.ldk;
  const bf16_t* vbase = v_row0 + (size_t)lead * p.ldv;

  // LDS-DMA staging: a wave-instruction writes 1 KB = RPI rows linearly, so the bank swizzles are applied to the per-lane
  // SOURCE chunk (the same XOR the fragment reads apply).  Keys past kv_len re-read the last valid row (masked later).
  constexpr int RPI = 1024 / ROWB;            // rows per wave-instruction (8 for D=64, 4 for D=128)
  constexpr int IPW = KT / RPI / NW;          // wave-instructions per wave per operand
  constexpr bool SPREAD_DMA = D == 128 && NKS % IPW == 0 && NKS / IPW >= 1;   // see the loop: requests issued between the MFMAs of S^T
  const int s_r = lane / CPR, s_c = lane % CPR;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
  // per-lane 32-bit source offsets inside a tile (loop invariant); the tile base advances as a scalar
  unsigned koff[IPW], voff[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int r = (wave_u * IPW + i) * RPI + s_r;
    koff[i] = (unsigned)r * (unsigned)p.ldk * 2u + Lay<D>::kchunk(r, s_c) * 16;
    voff[i] = (unsigned)r * (unsigned)p.ldv * 2u + Lay<D>::vchunk(r, s_c) * 16;
  }
  auto stage = [&](int kt, int buf) {
    const unsigned dK = lds0 + buf * (2 * KT * ROWB) + wave_u * IPW * 1024, dV = dK + KT * ROWB;
    const char* kb = (const char*)(kbase + (size_t)kt * KT * p.ldk);
    const char* vb = (const char*)(vbase + (size_t)kt * KT * p.ldv);
    if (kt * KT + KT <= kv_len) {
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        glds16_saddr(kb, koff[i], dK + i * 1024);
        glds16_saddr(vb, voff[i], dV + i * 1024);
      }
    } else {          // ragged last tile: rows past kv_len re-read the last valid row
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        const int r = (wave_u * IPW + i) * RPI + s_r;
        const int rr = min(r, kv_len - 1 - kt * KT);
        glds16_saddr(kb, (unsigned)rr * (unsigned)p.ldk * 2u + Lay<D>::kchunk(r, s_c) * 16, dK + i * 1024);
        glds16_saddr(vb, (unsigned)rr * (unsigned)p.ldv * 2u + Lay<D>::vchunk(r, s_c) * 16, dV + i * 1024);
      }
    }
  };

  // each wave issues 2*IPW LDS-DMA instructions per tile; nothing else in the loop touches vmcnt
#pragma unroll
  for (int t0 = 0; t0 < NB - 1; ++t0)
    if (t0 < n_tiles) stage(t0, t0);
  // ---- the query math (RoPE, pre-scale), behind the first DMA ------------------------------------------------
  {
    if (p.rope_cos) {
#pragma unroll
      for (int ks = 0; ks < NKS / 2; ++ks) {
        const u16x8 co = rco[ks], si = rsi[ks];
        const u16x8 lo = raw[ks], hi = raw[ks + NKS / 2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x1 = bf2f(lo[e]), x2 = bf2f(hi[e]), cc = bf2f(co[e]), ss = bf2f(si[e]);
          raw[ks][e] = f2bf(rbf(x1 * cc) + rbf(-x2 * ss));
          raw[ks + NKS / 2][e] = f2bf(rbf(x2 * cc) + rbf(x1 * ss));
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (p.q_prescale != 1.0f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[ks][e] = f2bf(bf2f(raw[ks][e]) * p.q_prescale);
      }
      if (!q_ok) raw[ks] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8, raw[ks]);
    }
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // exp2(s*sc - m*sc) = exp((s - m)/post_div).  RS with a rounded division (round_div): the scores the softmax sees are already divided.
  const bool round_div = RS && p.round_scores == 2;
  const float inv_div = 1.0f / p.post_div;
  const float sc = round_div ? 1.4426950408889634f : 1.4426950408889634f / p.post_div;

  // transposed-read lane constants: 16-lane group gi = lane>>4 -> d columns 16*(gi&1).., key rows 4*(gi>>1)..
  const int li = lane & 15, gi = lane >> 4;
  const int tr_key = 4 * (gi >> 1) + (li >> 2);       // + 32*st + 16*s + 8*jh
  const int tr_dcol = 16 * (gi & 1) + 4 * (li & 3);   // + 32*dt

  STAMP(t_loop);
  STAMP_ADD(2, t_begin, t_loop);
  for (int kt = 0; kt < n_tiles; ++kt) {
    STAMP(t0);
    // tile kt must have landed; up to NB-2 younger tiles may stay in flight
    const int younger = min(NB - 2, n_tiles - 1 - kt);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * IPW) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile kt visible to every wave; everyone is done reading buffer (kt-1) % NB, which is refilled now
    STAMP(t1);
    STAMP_ADD(3, t0, t1);
    const char* sK = smem + (kt % NB) * (2 * KT * ROWB);
    const char* sV = sK + KT * ROWB;
    const int key0 = kt * KT;
    // wave-uniform skips: tiles entirely in this wave's causal future; waves whose 32 query rows all lie past the sequence
    // (1025 = 8 x 128 + 1 rows per ViT frame: three of the last workgroup's four waves).  Such a wave still stages its share
    // of every tile and joins the barriers, but leaves its SIMD's issue slots to the co-resident workgroups.
    const bool skip = (CAUSAL && key0 > qw + 31 + kv_off) || (qw >= len || (p.q_tail > 0 && qw + 32 <= len - p.q_tail)) || (ksplit && (kt % NW) != wave);
    // The LDS-DMA requests of tile kt + NB - 1.  d = 128: not in one burst behind the barrier (the stamps priced that burst at 11 % of a
    // wave's lifetime: eight 1-KB requests queue at the CU's address unit) but one behind every second MFMA of S^T = K Q^T; a wave that
    // skips the tile, and the ragged last tile of a sequence, keep the burst.  d = 64 (half the requests, half the MFMAs to hide them
    // behind) measured 5 % slower spread out and keeps the burst (profiles/r3_attn_interleave.txt).
    const int st_kt = kt + NB - 1, st_buf = st_kt % NB;
    const bool st_any = st_kt < n_tiles;
    const bool st_spread = SPREAD_DMA && st_any && !skip && (st_kt * KT + KT <= kv_len);
    if (st_any && !st_spread) stage(st_kt, st_buf);
    STAMP(t2);
    STAMP_ADD(4, t1, t2);
    if (skip) continue;

    // ---- S^T = K · Q^T: the two 32-key chains alternate (no MFMA waits for its predecessor's result), the K fragments are read four
    // MFMAs ahead; sched_barrier pins that order (left alone, hipcc reads a fragment right in front of the MFMA that needs it) ----
    f32x16 sacc[2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[st][e] = 0.f;
    {
      const unsigned dK = lds0 + st_buf * (2 * KT * ROWB) + wave_u * IPW * 1024, dV = dK + KT * ROWB;
      const char* kb = (const char*)(kbase + (size_t)st_kt * KT * p.ldk);
      const char* vb = (const char*)(vbase + (size_t)st_kt * KT * p.ldv);
      auto kread = [&](int idx) {
        const int kr = (idx & 1) * 32 + c;
        return *(const bf16x8*)(sK + kr * ROWB + Lay<D>::kchunk(kr, 2 * (idx >> 1) + h) * 16);
      };
      auto qk = [&](auto with_dma) {
        bf16x8 kf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kf[i] = kread(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < 2 * NKS; ++idx) {
          sacc[idx & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[idx & 3], qf[idx >> 1], sacc[idx & 1], 0, 0, 0);
          if (idx + 4 < 2 * NKS) kf[idx & 3] = kread(idx + 4);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (decltype(with_dma)::value) {
            constexpr int EVERY = NKS / IPW;          // 2 * IPW requests over 2 * NKS MFMAs
            if (idx % EVERY == EVERY - 1) {
              const int u = idx / EVERY, i = u >> 1;
              if (u & 1) glds16_saddr(vb, voff[i], dV + i * 1024);
              else glds16_saddr(kb, koff[i], dK + i * 1024);
            }
          }
        }
      };
      if constexpr (SPREAD_DMA) {
        if (st_spread) qk(std::true_type{});
        else qk(std::false_type{});
      } else {
        // (d = 64: the compiler's own order of these eight MFMAs and reads, pinned only by read-ahead groups)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
          for (int st = 0; st < 2; ++st) sacc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kread(2 * ks + st), qf[ks], sacc[st], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < 2 * NKS - 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }

    // ---- online softmax on the fp32 scores (one FMA + one exp2 per element) ---------------------------------
    // p = exp2(s * c - m * c), c = log2(e) / post_div.  The mask is only evaluated on tiles that need it (the
    // causal diagonal / the ragged last tile); the O rescale is skipped when no row maximum moved (wave-uniform).
    if constexpr (RS) {
      // the reference's score matrix is a bf16 tensor (modeling_internlm2.py:417: matmul -> bf16, / sqrt(d) -> bf16;
      // modeling_intern_vit.py:153: (q * scale) @ k^T -> bf16): the same two roundings here, pairwise (one v_cvt_pk_bf16_f32 + two
      // unpacks per pair and rounding; scalar multiplies: packed fp32 math does not overlap with the matrix pipe).  Measured on the
      // benched batch at full depth (tests/manual/attention_numerics_study.py): WITHOUT these roundings a correct evaluation sits 4 bf16
      // ulps (mean) from the reference's scores, with them 1.5 - the reference's own spread between host thread counts.
      // (two straight-line forms: left inside the pair loop, the wave-uniform test becomes sixteen branches per tile)
      if (round_div) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            uint32_t pk = pack_bf2(f32x2{sacc[st][e], sacc[st][e + 1]});
            float a = __uint_as_float(pk << 16) * inv_div, b = __uint_as_float(pk & 0xffff0000u) * inv_div;
            pk = pack_bf2(f32x2{a, b});
            sacc[st][e] = __uint_as_float(pk << 16); sacc[st][e + 1] = __uint_as_float(pk & 0xffff0000u);
          }
      } else {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const uint32_t pk = pack_bf2(f32x2{sacc[st][e], sacc[st][e + 1]});
            sacc[st][e] = __uint_as_float(pk << 16); sacc[st][e + 1] = __uint_as_float(pk & 0xffff0000u);
          }
      }
    }
    const int qpos = qw + c + kv_off;   // index of the last key this query may see (causal)
    const bool need_mask = (key0 + KT > kv_len) || (CAUSAL && key0 + KT - 1 > qw + kv_off);
    float tmax = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + st * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const bool vis = key < kv_len && (!CAUSAL || key <= qpos);
          const float sv = vis ? sacc[st][e] : -INFINITY;
          sacc[st][e] = sv;
          tmax = fmaxf(tmax, sv);
        }
    } else {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[st][e]);
    }
    STAMP(t3);
    STAMP_ADD(5, t2, t3);
    // Lazy rescale: the running reference m_run only moves when some row's maximum has grown by more than 2^8 in the exp2
    // domain (a new row maximum turns up in almost every tile, a jump of 8 octaves almost never after the first).  Until
    // then p = exp2((s - m_run) * c) may exceed 1 (< 2^8): harmless in fp32 / bf16, and the final division by l uses the
    // same reference, so the result is the same softmax.  (NaN-safe: -inf - -inf compares false -> no move, mc = 0.)
    // The test runs on the lane's OWN maximum (the other half of the row's keys sits in lane ^ 32): some lane sees a jump exactly when the
    // row's maximum jumps, so the exchange between the halves is only paid on the rare tiles that move m (same bits as exchanging first).
    if (__any((tmax - m_run) * sc > 8.0f)) {
      const float m_new = fmaxf(m_run, xhalf_max(tmax));
      // rows with no visible key yet keep m = -inf; guard the exp argument (m_run = -inf -> alpha = 0)
      const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * sc);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
      m_run = m_new;
    }
    const float mc = (m_run == -INFINITY) ? 0.f : m_run * sc;
    // (the loop is VALU-issue-bound: SQ counters in profiles/r1_attn_sq.txt, phase stamps in profiles/r3_attn_stamps.txt); raw
    // v_exp_f32: p underflows to 0, no fix-up needed
    // One v_fma_f32 / v_add_f32 per score (the file is built with -fno-slp-vectorize so that hipcc does not re-pack them into
    // v_pk_*_f32: the packed forms halve the instruction count but not the issue time - the guide prices them as an anti-lever beside
    // MFMAs).  Plain C, not inline asm: hipcc inserts the wait state a VALU read of a v_exp_f32 result needs only for instructions it
    // can see (an asm v_add_f32 right behind the v_exp_f32 summed stale registers).  Two running sums.
    // Quarter-wise: the softmax of keys [16 q, 16 q + 16) is followed by their share of O^T += V^T P^T, and the schedule below moves each
    // quarter's MFMAs between the NEXT quarter's softmax instructions (inside one wave a few independent VALU instructions behind every
    // MFMA are almost free: profiles/r3_mfma_valu_overlap_probe.txt).
    float ps0 = 0.f, ps1 = 0.f;
    const float nmc = -mc;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        typedef __attribute__((ext_vector_type(8))) float f32x8;
        f32x8 pw;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[st][8 * s2 + j], sc, nmc));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[st][8 * s2 + j + 1], sc, nmc));
          ps0 += p0;
          ps1 += p1;
          pw[j] = p0;
          pw[j + 1] = p1;
        }
        const bf16x8 pf = __builtin_convertvector(pw, bf16x8);   // four v_cvt_pk_bf16_f32
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int dcol = 32 * dt + tr_dcol;
          const int k_lo = 32 * st + 16 * s2 + tr_key, k_hi = k_lo + 8;
          const char* a_lo = sV + k_lo * ROWB + Lay<D>::vchunk(k_lo, dcol >> 3) * 16 + (dcol & 7) * 2;
          const char* a_hi = sV + k_hi * ROWB + Lay<D>::vchunk(k_hi, dcol >> 3) * 16 + (dcol & 7) * 2;
          const s16x4 v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)a_lo);
          const s16x4 v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)a_hi);
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 vv = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc[dt], 0, 0, 0);
        }
      }
    }
    l_run += xhalf_sum(ps0 + ps1);
    // quarter 0's softmax (8 fma, 8 exp, 8 add, 4 cvt), then per MFMA of quarters 0..2 its two transposed reads and 28 / NDT of the
    // next quarter's VALU instructions, then quarter 3's MFMAs
    __builtin_amdgcn_sched_group_barrier(0x002, 28, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);            // the reads of an MFMA are issued two MFMAs ahead of it
#pragma unroll
    for (int i = 0; i < 3 * NDT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 28 / NDT, 0);
    }
#pragma unroll
    for (int i = 0; i < NDT - 2; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#ifdef AIGV_ATTN_STAMP
    asm volatile("" :: "v"(oacc[0][0]));     // the stamp below must not move in front of the last MFMA's result
    {
      STAMP(t4);
      STAMP_ADD(6, t3, t4);
      if (lane == 0) st_acc[8] += 1;
    }
#endif
  }
  STAMP(t_epi);

  if (ksplit) {
    // ---- merge the NW key-split states: waves 1.. publish (m, l, O^T) in the K/V ring (free now), wave 0 combines in wave order ----
    __syncthreads();
    constexpr int PER_LANE = NDT * 16 + 2;               // floats per lane
    float* pub = (float*)smem;
    if (wave > 0) {
      float* mine = pub + ((size_t)(wave - 1) * 64 + lane) * PER_LANE;
      mine[0] = m_run; mine[1] = l_run;
#pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[2 + i * 16 + e] = oacc[i][e];
    }
    __syncthreads();
    if (wave > 0) return;
    float M = m_run;
    for (int w = 1; w < NW; ++w) M = fmaxf(M, pub[((size_t)(w - 1) * 64 + lane) * PER_LANE]);
    const float a0 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - M) * sc);
    l_run *= a0;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[i][e] *= a0;
    for (int w = 1; w < NW; ++w) {
      const float* o = pub + ((size_t)(w - 1) * 64 + lane) * PER_LANE;
      const float aw = (o[0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((o[0] - M) * sc);
      l_run += o[1] * aw;
#pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] += o[2 + i * 16 + e] * aw;
    }
    m_run = M;     // (the lead-key merge below continues from the merged state)
  }

  if (lead) {
    // ---- the lead key (see above), merged like one more softmax state: M = max(m, s0), O = O e^(m - M) + v0 e^(s0 - M), l likewise.
    // Its K elements sit at this lane's query-fragment positions, its V elements at this lane's O^T positions; the loop's registers are free now.
    u16x8 k0r[NKS];
    u16x4 v0r[NDT * 4];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) k0r[ks] = *(const u16x8*)(k_row0 + 16 * ks + 8 * h);
#pragma unroll
    for (int i = 0; i < NDT * 4; ++i) v0r[i] = *(const u16x4*)(v_row0 + 32 * (i >> 2) + 8 * (i & 3) + 4 * h);
    float s0 = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const u16x8 qq = __builtin_bit_cast(u16x8, qf[ks]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s0 = __builtin_fmaf(bf2f(qq[j]), bf2f(k0r[ks][j]), s0);
    }
    s0 = xhalf_sum(s0);                      // the other half of the row's dimensions sits in lane ^ 32
    if constexpr (RS) {
      if (p.round_scores) s0 = rbf(s0);
      if (p.round_scores == 2) s0 = rbf(s0 * inv_div);
    }
    const float M = fmaxf(m_run, s0);
    const float a0 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - M) * sc);
    const float a1 = __builtin_amdgcn_exp2f((s0 - M) * sc);
    l_run = l_run * a0 + a1;
#pragma unroll
    for (int i = 0; i < NDT * 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) oacc[i >> 2][4 * (i & 3) + e] = oacc[i >> 2][4 * (i & 3) + e] * a0 + a1 * bf2f(v0r[i][e]);
  }

  // ---- normalise and store: lane (c,h) owns O[qw+c][32*dt + 8*(e>>2) + 4h + (e&3)] -------------------------
  const int qr = qw + c;
  if (qr < len) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    bf16_t* op = p.o + (size_t)(row0 + qr) * p.ldo + (size_t)hq * D;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(oacc[dt][4 * e4 + e] * inv);
        *(u16x4*)(op + 32 * dt + 8 * e4 + 4 * h) = o;
      }
  }
#ifdef AIGV_ATTN_STAMP
  {
    STAMP(t_end);
    STAMP_ADD(7, t_epi, t_end);
    if (lane == 0) {
      st_acc[1] = t_end - t_begin;
      st_acc[0] = 1;
      for (int i = 0; i < 9; ++i) atomicAdd(&g_attn_stamp[D == 64 ? 0 : 1][blockIdx.x % STAMP_REPL][i], st_acc[i]);
    }
  }
#endif
}

// ---- decode attention (q_len = 1 per sequence) against the KV cache: split-KV, two passes ------------------------
// reference: InternLM2Attention.forward with a cache (modeling_internlm2.py:397-424).  The cache is tiny next to the
// weights (131 KB per token per clip), so the kernel only has to be parallel and dependency-free:
//   pass 1  grid (key chunks of 128, kv heads, sequences): scores of the G grouped query heads for 128 keys (two
//           threads per key, 64 dims each), chunk-local softmax statistics, P.V partial sums for 128 dims
//   pass 2  grid (kv heads, sequences): merge the chunk partials (max / sum rescale) and write bf16
// Rounding points as the eager path: score -> bf16, / sqrt(d) -> bf16, softmax fp32, P -> bf16 (un-normalised).
constexpr int DC = 128;   // keys per chunk

template <int G>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(const bf16_t* __restrict__ q, int ldq, int q_group_stride,
                                                                  const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                                  const int32_t* __restrict__ kv_lens, int cap, float post_div,
                                                                  float* __restrict__ ws, int max_chunks) {
  constexpr int D = 128;
  __shared__ float sQ[G][D];
  __shared__ float sS[G][DC];          // scores, then probabilities
  __shared__ float sM[G], sL[G];
  __shared__ float sAcc[4][G][D];
  const int chunk = blockIdx.x, hk = blockIdx.y, seq = blockIdx.z, n_kv = gridDim.y;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kv_len = kv_lens[seq];
  const int key0 = chunk * DC;
  float* wbase = ws + ((((size_t)seq * n_kv + hk) * max_chunks + chunk) * G) * (D + 2);
  if (key0 >= kv_len) return;                                       // pass 2 only visits chunks below kv_len
  const int nkeys = min(DC, kv_len - key0);
  for (int i = t; i < G * D; i += 256) sQ[i / D][i % D] = bf2f(q[(size_t)seq * ldq + (size_t)hk * q_group_stride + i]);
  __syncthreads();
  const bf16_t* kb = kc + (((size_t)seq * n_kv + hk) * cap + key0) * D;
  const bf16_t* vb = vc + (((size_t)seq * n_kv + hk) * cap + key0) * D;
  {  // scores: thread -> (key = t>>1, half = t&1)
    const int kk = t >> 1, half = t & 1;
    float dot[G];
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2) dot[j2] = 0.f;
    if (kk < nkeys) {
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        const u16x8 raw = *(const u16x8*)(kb + (size_t)kk * D + half * 64 + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float kvv = bf2f(raw[e]);
#pragma unroll
          for (int j2 = 0; j2 < G; ++j2) dot[j2] += kvv * sQ[j2][half * 64 + c8 * 8 + e];
        }
      }
    }
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2) {
      const float tot = dot[j2] + __shfl_xor(dot[j2], 1, 64);
      if (half == 0) {
        float sc = rbf(tot);
        if (post_div != 1.0f) sc = rbf(sc / post_div);
        sS[j2][kk] = kk < nkeys ? sc : -INFINITY;
      }
    }
  }
  __syncthreads();
  for (int j2 = wave; j2 < G; j2 += 4) {   // chunk softmax statistics: one wave per head
    const float a = sS[j2][lane], b = sS[j2][lane + 64];
    const float m = wave_max(fmaxf(a, b));
    const float pa = __expf(a - m), pb = __expf(b - m);
    const float l = wave_sum(pa + pb);
    sS[j2][lane] = rbf(pa);
    sS[j2][lane + 64] = rbf(pb);
    if (lane == 0) { sM[j2] = m; sL[j2] = l; }
  }
  __syncthreads();
  {  // P.V: thread -> dims (2*d2, 2*d2+1), key slice = wave (32 keys)
    const int d2 = lane;
    float acc[G][2];
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2) acc[j2][0] = acc[j2][1] = 0.f;
    const int kbeg = wave * 32, kend = min(kbeg + 32, nkeys);
    // eight V rows in flight per step: the loop is a chain of L2 latencies, not of FMAs (rows past kend re-read the last valid
    // row and get p = 0: keys are summed in the same order as a plain loop)
    for (int k0 = kbeg; k0 < kend; k0 += 8) {
      uint32_t raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) raw[u] = *(const uint32_t*)(vb + (size_t)min(k0 + u, kend - 1) * D + 2 * d2);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float v0 = bf2f((bf16_t)(raw[u] & 0xffff)), v1 = bf2f((bf16_t)(raw[u] >> 16));
#pragma unroll
        for (int j2 = 0; j2 < G; ++j2) {
          const float pr = (k0 + u < kend) ? sS[j2][k0 + u] : 0.f;
          acc[j2][0] += pr * v0;
          acc[j2][1] += pr * v1;
        }
      }
    }
#pragma unroll
    for (int j2 = 0; j2 < G; ++j2) { sAcc[wave][j2][2 * d2] = acc[j2][0]; sAcc[wave][j2][2 * d2 + 1] = acc[j2][1]; }
  }
  __syncthreads();
  for (int i = t; i < G * D; i += 256) {
    const int j2 = i / D, d = i % D;
    wbase[(size_t)j2 * (D + 2) + 2 + d] = (sAcc[0][j2][d] + sAcc[1][j2][d]) + (sAcc[2][j2][d] + sAcc[3][j2][d]);
  }
  if (t < G) { wbase[(size_t)t * (D + 2)] = sM[t]; wbase[(size_t)t * (D + 2) + 1] = sL[t]; }
}

constexpr int MERGE_MAX_CHUNKS = 128;   // capacity / DC the merge kernel holds in LDS (a 16 384-token cache)

// pass 2: grid (kv heads x pairs of query heads, sequences); a workgroup merges the chunk partials of two query heads (256 outputs,
// one per thread): chunk maxima -> global maximum, chunk weights exp(m - M), denominator (chunks in order), then the weighted sum
// of the chunk accumulators with sixteen loads in flight (the chunk loop is a chain of L2 latencies, not of FMAs).
template <int G>
__global__ __launch_bounds__(256) void attn_decode_merge_kernel(const float* __restrict__ ws, int max_chunks,
                                                                const int32_t* __restrict__ kv_lens, bf16_t* __restrict__ o,
                                                                int ldo) {
  constexpr int D = 128;
  constexpr int HP = G >= 2 ? 2 : 1;                  // query heads per workgroup
  constexpr int PAIRS = (G + HP - 1) / HP;
  __shared__ float sM[HP][MERGE_MAX_CHUNKS], sW[HP][MERGE_MAX_CHUNKS];   // chunk maxima, then chunk weights exp(m - M)
  __shared__ float sL[HP];
  const int hk = blockIdx.x / PAIRS, j0 = (blockIdx.x % PAIRS) * HP, seq = blockIdx.y, n_kv = gridDim.x / PAIRS;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nch = (kv_lens[seq] + DC - 1) / DC;
  const float* base = ws + (((size_t)seq * n_kv + hk) * max_chunks) * G * (D + 2);
  for (int i = t; i < nch * HP; i += 256) {
    const int ch = i / HP, jj = i - ch * HP;
    if (j0 + jj < G) {
      const float* pp = base + ((size_t)ch * G + j0 + jj) * (D + 2);
      sM[jj][ch] = pp[0];
      sW[jj][ch] = pp[1];
    }
  }
  __syncthreads();
  if (wave < HP && j0 + wave < G) {   // one wave per head
    const int jj = wave;
    float m = -INFINITY;
    for (int ch = lane; ch < nch; ch += 64) m = fmaxf(m, sM[jj][ch]);
    m = wave_max(m);
    float l = 0.f;
    for (int c0 = 0; c0 < nch; c0 += 64) {
      const int ch = c0 + lane;
      const float w = ch < nch ? __expf(sM[jj][ch] - m) : 0.f;
      const float lw = ch < nch ? sW[jj][ch] * w : 0.f;
      if (ch < nch) sW[jj][ch] = w;
      l += wave_sum(lw);
    }
    if (lane == 0) sL[jj] = l;
  }
  __syncthreads();
  const int jj = t / D, d = t % D, j2 = j0 + jj;
  if (jj >= HP || j2 >= G) return;
  float A = 0.f;
  int ch = 0;
  for (; ch + 16 <= nch; ch += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = base[((size_t)(ch + u) * G + j2) * (D + 2) + 2 + d];
#pragma unroll
    for (int u = 0; u < 16; ++u) A += v[u] * sW[jj][ch + u];
  }
  {   // the last partial group with clamped (re-read, weight 0) loads instead of a one-at-a-time tail
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = ch < nch ? base[((size_t)min(ch + u, nch - 1) * G + j2) * (D + 2) + 2 + d] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) A += ch + u < nch ? v[u] * sW[jj][ch + u] : 0.f;
  }
  const float L = sL[jj];
  o[(size_t)seq * ldo + (size_t)(hk * G + j2) * D + d] = f2bf(L > 0.f ? A / L : 0.f);
}

}  // namespace

#ifdef AIGV_ATTN_STAMP
extern "C" int aigv_debug_attn_stamps(unsigned long long* out32, int reset) {   // diagnostic build only: 2 x 16 accumulators
  static unsigned long long host[2][STAMP_REPL][16];
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (out32) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamp), sizeof host) != hipSuccess) return -2;
    for (int d = 0; d < 2; ++d)
      for (int i = 0; i < 16; ++i) {
        unsigned long long t = 0;
        for (int r = 0; r < STAMP_REPL; ++r) t += host[d][r][i];
        out32[d * 16 + i] = t;
      }
  }
  if (reset) {
    memset(host, 0, sizeof host);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), host, sizeof host) != hipSuccess) return -2;
  }
  return 0;
}
#endif

const char* aigv_attn_check(const AttnArgs& a, int head_dim) {
  if (head_dim != 64 && head_dim != 128) return "attention: head_dim must be 64 or 128";
  if (a.n_seq <= 0 || a.max_len <= 0) return "attention: empty problem";
  if (a.n_kv_heads <= 0 || a.n_heads % a.n_kv_heads) return "attention: n_heads must be a multiple of n_kv_heads";
  if ((a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) || (a.ldo % 4) || (a.q_group_stride % 8) || (a.kv_head_stride % 8))
    return "attention: strides must keep 16-byte alignment";
  if (!a.q || !a.k || !a.v || !a.o || !a.cu) return "attention: null operand";
  if ((a.kv_len_offset != 0 || a.kv_off) && !a.kv_seq_stride) return "attention: a key offset needs K/V in cache layout (kv_seq_stride)";
  if (a.kv_len_offset < 0 || (a.kv_seq_stride % 8)) return "attention: bad key offset / cache stride";
  if ((a.rope_cos != nullptr) != (a.rope_sin != nullptr) || (a.rope_cos && !a.rope_pos)) return "attention: query RoPE needs positions, cos and sin";
  return nullptr;
}


// NB = 2: deeper rings (3, 4 buffers) measured 5-15 % slower on the ViT shape - they cost resident workgroups (LDS), and
// with four workgroups per CU the wait for the next tile is already covered by the others' work
template <int D, bool CAUSAL, int NW, bool RS, int NB = 2>
static hipError_t launch_attn_rs(const AttnArgs& a, hipStream_t s) {
  constexpr int LDS = NB * 2 * KT * (D * 2);
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)attn_fwd_kernel<D, CAUSAL, NW, NB, RS>, LDS); e != hipSuccess) return e;
  const int nqb = (a.max_len + NW * 32 - 1) / (NW * 32) - a.q_begin / (NW * 32);
  hipLaunchKernelGGL((attn_fwd_kernel<D, CAUSAL, NW, NB, RS>), dim3(nqb * a.n_heads * a.n_seq), dim3(NW * 64), LDS, s, a);
  return hipGetLastError();
}
template <int D, bool CAUSAL, int NW>
static hipError_t launch_attn(const AttnArgs& a, hipStream_t s) {
  return a.round_scores ? launch_attn_rs<D, CAUSAL, NW, true>(a, s) : launch_attn_rs<D, CAUSAL, NW, false>(a, s);
}

static hipError_t launch_attention32(const AttnArgs& a, int head_dim, hipStream_t s) {
  // 4 waves (128 query rows) per workgroup.  With the XCD-aware block order the K/V stream of a head is shared in L2, and
  // 8-wave workgroups (half the K/V reads, half the resident workgroups) measured equal or slower on every headline shape
  // (scripts/attn_bench.py with AB_WAVES=1); the 8-wave form stays reachable through aigv_tune_attention for such A/Bs.
  const int nw = a.waves == 8 ? 8 : 4;
  if (head_dim == 64) {
    if (a.causal) return nw == 8 ? launch_attn<64, true, 8>(a, s) : launch_attn<64, true, 4>(a, s);
    return nw == 8 ? launch_attn<64, false, 8>(a, s) : launch_attn<64, false, 4>(a, s);
  }
  if (a.causal) return nw == 8 ? launch_attn<128, true, 8>(a, s) : launch_attn<128, true, 4>(a, s);
  return nw == 8 ? launch_attn<128, false, 8>(a, s) : launch_attn<128, false, 4>(a, s);
}

// The one kernel above runs every shape.  A software-pipelined 64-rows-per-wave kernel (round 2, attention64.hip) won the isolated
// A/B by 7 % and lost the in-step one by 6 % twice (profiles/r2_attn_ab.txt, r2_attn_inmodel_ab.txt) and was removed in round 3; so
// did three-deep K/V rings (profiles/r3_attn_ring_negative.txt) and an 8-wave kernel with role-alternating halves (round 3,
// profiles/r3_attn8_negative.txt).  aigv_tune_attention: 4 / 8 waves per workgroup (A/B only).
hipError_t aigv_launch_attention(const AttnArgs& a_in, int head_dim, hipStream_t s) {
  AttnArgs a = a_in;
  {
    // A power-of-two query pre-scale (InternViT: d^-1/2 = 2^-3) commutes exactly with the bf16 rounding of q and with the fp32 dot
    // products, so it is folded into the softmax's exp2 scale instead of being applied to every query element: the same bits
    // (the scores the kernel sees are 2^k times larger, their scale 2^k times smaller), ~100 fewer VALU instructions per wave.
    int ex = 0;
    if (a.q_prescale > 0.f && a.q_prescale != 1.0f && frexpf(a.q_prescale, &ex) == 0.5f && ex > -60 && ex < 60) {
      a.post_div = a.post_div / a.q_prescale;
      a.q_prescale = 1.0f;
    }
    // round_scores: 1 = the score matrix rounds to bf16 once (a power-of-two division commutes with the rounding and stays folded into
    // the exp2 scale), 2 = and again after the division (InternLM2: sqrt(128) is no power of two)
    if (a.round_scores) {
      a.round_scores = (a.post_div > 0.f && frexpf(a.post_div, &ex) == 0.5f) ? 1 : 2;
    }
  }
  return launch_attention32(a, head_dim, s);
}

size_t aigv_attention_decode_ws_floats(int n_seq, int n_kv, int g, int cap) {
  return (size_t)n_seq * n_kv * ((cap + DC - 1) / DC) * g * (128 + 2);
}

hipError_t aigv_launch_attention_decode(const bf16_t* q, int ldq, int q_group_stride, const bf16_t* kc,
                                        const bf16_t* vc, const int32_t* kv_lens, int cap, bf16_t* o, int ldo,
                                        int n_seq, int n_kv, int g, int head_dim, float post_div, int max_kv_len,
                                        float* ws, hipStream_t s) {
  if (head_dim != 128 || !ws || max_kv_len <= 0 || max_kv_len > cap || (cap + DC - 1) / DC > MERGE_MAX_CHUNKS) return hipErrorInvalidValue;
  const int max_chunks = (cap + DC - 1) / DC;
  dim3 grid1((max_kv_len + DC - 1) / DC, n_kv, n_seq), grid2(n_kv * (g >= 2 ? (g + 1) / 2 : 1), n_seq);
#define DEC(G)                                                                                                              \
  hipLaunchKernelGGL((attn_decode_partial_kernel<G>), grid1, dim3(256), 0, s, q, ldq, q_group_stride, kc, vc, kv_lens, cap, \
                     post_div, ws, max_chunks);                                                                              \
  hipLaunchKernelGGL((attn_decode_merge_kernel<G>), grid2, dim3(256), 0, s, ws, max_chunks, kv_lens, o, ldo)
  switch (g) {
    case 1: DEC(1); break;
    case 2: DEC(2); break;
    case 3: DEC(3); break;
    case 4: DEC(4); break;
    case 5: DEC(5); break;
    case 6: DEC(6); break;
    case 7: DEC(7); break;
    case 8: DEC(8); break;
    default: return hipErrorInvalidValue;
  }
#undef DEC
  return hipGetLastError();
}
