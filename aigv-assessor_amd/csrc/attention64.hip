// Fused attention forward for gfx950, second structure: ONE wave per SIMD, 64 query rows per wave, the whole 512-entry register
// file per lane.  Same operands, layouts and numerics as attention.hip (packed varlen, bf16 in / fp32 accumulate, S^T = K.Q^T so
// that softmax statistics are lane-local, the S^T accumulator as the B operand of O^T += V^T.P^T, lazy rescale); what changes is
// the work split and the schedule:
//
//   * a workgroup = 4 waves = one 256-row query block; a wave owns TWO 32-row sub-blocks, rows q0 + 32 (w + 4 j), j = 0, 1 (the
//     interleave gives every wave one early and one late sub-block, so causal blocks end within a tile of each other).  Every K
//     fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16) read from LDS feeds two MFMAs instead of one: half the
//     LDS bytes per FLOP of the 32-row form.
//   * software pipeline over 32-key half tiles u = (tile, half), one barrier per 64-key tile:
//       stage 1   S(u+1) = K(u+1).Q^T   (16 MFMAs at D = 128)   beside   p = exp2(S(u) c - m c), row sums, bf16 packing
//       stage 2   O += V(u)^T.P(u)^T    (16 MFMAs)              beside   row maxima of S(u+1), the lazy-rescale decision
//     so the softmax VALU work of one half tile runs in the issue slots the matrix pipe leaves free while it multiplies the
//     neighbouring one (one MFMA 32x32x16 occupies the pipe for 32 cycles and the issue port for 8).  Half tiles keep the two
//     live score blocks at 32 + 32 registers per lane.
//   * K tiles live in a 3-deep LDS ring (K(t) is read from the second half of tile t-1's iteration to the first half of tile
//     t's), V tiles in a 2-deep one, filled by LDS-DMA a full tile period ahead of their use: at the top of tile t every wave
//     waits for its own pieces of K(t+1) / V(t), one barrier publishes them and releases K(t-1)'s and V(t-1)'s buffers, which
//     the DMAs of K(t+2) / V(t+1) then refill.
//   * mask arithmetic only on the tiles that need it (causal diagonal region, ragged last tile), outside the pipelined stages.
// Short query blocks (< 256 rows) run with the rows they have (a wave without rows only moves its share of the tiles); the old
// 32-row kernel keeps the cases it is better at (tiny sequences, the single left-over row of a 1025-row ViT frame with its
// key-split form) - aigv_launch_attention decides.
#include "common.h"
#include "kernels.h"
#include "attn_lay.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int KT = 64;    // keys per tile
// Timing ablations (wrong results; scripts/attn_ablate.py builds variants with -DATTN64_ABL=<bits>): 4 no DMA in the loop, 8 no P.V
// MFMAs, 16 no Q.K MFMAs, 32 no LDS fragment reads, 64 no softmax, 128 no row maxima, 256 one key tile per block, 512 no query
// load, 1024 no output store.  0 in every build that ships.
#ifndef ATTN64_ABL
#define ATTN64_ABL 0
#endif
// "no key seen yet" for the running softmax reference: a large FINITE negative, so that no (-inf) - (-inf) ever arises - the file
// is compiled with -fno-honor-nans (without it every fmaxf on an MFMA result costs an extra canonicalising v_max_f32)
constexpr float M_NONE = -1.0e30f;

// the value the same lane index holds in the other half-wave, combined with this lane's: one v_permlane32_swap
__device__ __forceinline__ float half_max(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct Blk {
  int lane, c, h, wave;
  int row0, len, q_lim, q0, kv_off, kv_len, n_tiles, hq, g;
  const bf16_t *kbase, *vbase;
  char* smem;
  unsigned lds0;
  float sc;
};

template <int D>
struct Stager {   // LDS-DMA of one K or V tile: 4 waves x IPW wave-instructions of 1 KB
  static constexpr int ROWB = Lay<D>::ROWB, CPR = D / 8, RPI = 1024 / ROWB, IPW = KT / RPI / 4, TILE = KT * ROWB;
  // Nothing lane-dependent is kept across the tile loop: hipcc would spill such loop-invariant offsets under the loop's register
  // pressure and reload them each tile behind its own s_waitcnt vmcnt(0), which serialises the DMA issue.  The per-lane source
  // offset is rebuilt from the lane id on every call (a handful of VALU instructions per tile); `lane` is passed through an
  // empty asm so that the compiler cannot hoist the arithmetic out of the loop again.
  int wave_u;
  __device__ __forceinline__ void init(int wave_u_) { wave_u = wave_u_; }
  // chunk(r0 + i RPI, c) = chunk(r0, c) ^ DELTA(i) for both swizzles (K: 4 i at D = 64 and 128; V: i at D = 128, 0 at D = 64), so
  // the i-th offset of a lane is its first one plus i row groups with the chunk bits flipped by a constant: two VALU
  // instructions per wave-instruction
  template <bool IS_K>
  __device__ __forceinline__ void tile(const AttnArgs& p, const Blk& b, int kt, int slot) const {
    int lane = b.lane;
    asm volatile("" : "+v"(lane));
    const int s_r = lane / CPR, s_c = lane % CPR;
    const unsigned ld2 = (unsigned)(IS_K ? p.ldk : p.ldv) * 2u;
    const unsigned dst = b.lds0 + ((IS_K ? 0 : 3) + slot) * TILE + wave_u * IPW * 1024;
    const char* base = (const char*)((IS_K ? b.kbase : b.vbase) + (size_t)kt * KT * (IS_K ? p.ldk : p.ldv));
    const int r0 = wave_u * IPW * RPI + s_r;
    if (kt * KT + KT <= b.kv_len) {
      const unsigned c16 = (unsigned)(IS_K ? Lay<D>::kchunk(r0, s_c) : Lay<D>::vchunk(r0, s_c)) * 16u;
      const unsigned row = (unsigned)r0 * ld2;
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        constexpr int KD = 4, VD = D == 128 ? 1 : 0;
        const unsigned delta = (unsigned)(i * (IS_K ? KD : VD)) * 16u;
        glds16_saddr(base, row + (unsigned)(i * RPI) * ld2 + (c16 ^ delta), dst + i * 1024);
      }
    } else {       // ragged last tile: keys past kv_len re-read the last valid row (masked later)
      const int last = b.kv_len - 1 - kt * KT;
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        const int r = r0 + i * RPI;
        const int ch = IS_K ? Lay<D>::kchunk(r, s_c) : Lay<D>::vchunk(r, s_c);
        glds16_saddr(base, (unsigned)min(r, last) * ld2 + ch * 16, dst + i * 1024);
      }
    }
  }
  __device__ __forceinline__ void k(const AttnArgs& p, const Blk& b, int kt, int slot) const { tile<true>(p, b, kt, slot); }
  __device__ __forceinline__ void v(const AttnArgs& p, const Blk& b, int kt, int slot) const { tile<false>(p, b, kt, slot); }
};

// every wave waits for its own DMA pieces, then one barrier: the tiles are visible to all and the buffers read last tile are free
__device__ __forceinline__ void tile_fence() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

template <int D, bool CAUSAL, int NQ>
__device__ __forceinline__ void attn64_run(const AttnArgs& p, const Blk& b, const Stager<D>& dma) {
  constexpr int ROWB = Lay<D>::ROWB, NKS = D / 16, NDT = D / 32, TILE = KT * ROWB;
  const int c = b.c, h = b.h, lane = b.lane;
  const float sc = b.sc;
  const int qs[2] = {b.q0 + 32 * b.wave, b.q0 + 32 * (b.wave + 4)};   // first row of each sub-block

  // ---- Q^T fragments: lane (c,h) holds Q[qs[j]+c][16*ks + 8h + e], rotated / pre-scaled as attention.hip does ----
  bf16x8 qf[NQ][NKS];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int qr = qs[j] + c;
    const bool ok = qr < b.q_lim && !(ATTN64_ABL & 512);
    const bf16_t* qp = p.q + (size_t)(b.row0 + (ok ? qr : 0)) * p.ldq + (size_t)(b.hq / b.g) * p.q_group_stride + (b.hq % b.g) * D;
    u16x8 raw[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) raw[ks] = *(const u16x8*)(qp + 16 * ks + 8 * h);
    if (p.rope_cos) {
      // rotate_half pairs dimension i with i + D/2: k-steps ks and ks + NKS/2 of the same lane (modeling_internlm2.py:247-261)
      const size_t tb = (size_t)p.rope_pos[b.row0 + (ok ? qr : 0)] * (D / 2);
#pragma unroll
      for (int ks = 0; ks < NKS / 2; ++ks) {
        const u16x8 co = *(const u16x8*)(p.rope_cos + tb + 16 * ks + 8 * h), si = *(const u16x8*)(p.rope_sin + tb + 16 * ks + 8 * h);
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
      if (!ok || (ATTN64_ABL & 512)) raw[ks] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      qf[j][ks] = __builtin_bit_cast(bf16x8, raw[ks]);
    }
  }

  f32x16 oacc[NQ][NDT];
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    m_run[j] = M_NONE; l_run[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[j][i][e] = 0.f;
  }
  // transposed-read lane constants: 16-lane group gi = lane>>4 -> d columns 16*(gi&1).., key rows 4*(gi>>1)..
  const int li = lane & 15, gi = lane >> 4;
  const int tr_key = 4 * (gi >> 1) + (li >> 2);       // + 32*st + 16*s2 + 8*jh
  const int tr_dcol = 16 * (gi & 1) + 4 * (li & 3);   // + 32*dt

  // ---- the pieces of a half tile (32 keys) -------------------------------------------------------------------------------------
  // K fragments of one half tile (lane (c,h): K[st*32 + c][16 ks + 8h ..]): loaded a stage ahead of the MFMAs that use them
  auto kload = [&](bf16x8 (&kf)[NKS], int slot, int st) {
    const char* sK = b.smem + slot * TILE;
    const int kr = st * 32 + c;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kf[ks] = *(const bf16x8*)(sK + kr * ROWB + Lay<D>::kchunk(kr, 2 * ks + h) * 16);
  };
  auto qk = [&](f32x16 (&s)[NQ], const bf16x8 (&kf)[NKS]) {   // S^T = K . Q^T for both sub-blocks: every K fragment feeds NQ MFMAs
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < NQ; ++j) s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[j][ks], s[j], 0, 0, 0);
  };
  auto mask = [&](f32x16 (&s)[NQ], int kt, int st) {   // -inf on keys a row may not see; only on the half tiles that hold such keys
    const int key0 = kt * KT + st * 32;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const bool need = (key0 + 32 > b.kv_len) || (CAUSAL && key0 + 31 > qs[j] + b.kv_off);   // wave-uniform
      if (!need) continue;
      const int qpos = qs[j] + c + b.kv_off;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const bool vis = key < b.kv_len && (!CAUSAL || key <= qpos);
        s[j][e] = vis ? s[j][e] : -INFINITY;
      }
    }
  };
  auto rowmax = [&](const f32x16 (&s)[NQ], float (&mn)[NQ]) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      float t0 = fmaxf(fmaxf(s[j][0], s[j][1]), s[j][2]), t1 = fmaxf(fmaxf(s[j][3], s[j][4]), s[j][5]);
#pragma unroll
      for (int e = 6; e + 3 < 16; e += 4) {          // v_max3_f32: two new scores per instruction
        t0 = fmaxf(fmaxf(t0, s[j][e]), s[j][e + 1]);
        t1 = fmaxf(fmaxf(t1, s[j][e + 2]), s[j][e + 3]);
      }
      t0 = fmaxf(fmaxf(t0, s[j][14]), s[j][15]);
      mn[j] = fmaxf(fmaxf(t0, t1), m_run[j]);
      mn[j] = half_max(mn[j]);
    }
  };
  // Lazy rescale (attention.hip): the running reference moves only when some row's maximum has grown by more than 2^8 in the exp2
  // domain; until then p may exceed 1 (< 2^8) and the final division by l uses the same reference.  Called after the pending
  // half tile's P.V is complete and before the next one is exponentiated, so everything at the old reference is scaled exactly once.
  auto rescale = [&](const float (&mn)[NQ]) {
    bool need = false;
#pragma unroll
    for (int j = 0; j < NQ; ++j) need |= (mn[j] - m_run[j]) * sc > 8.0f;
    if (__any(need)) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const float alpha = __builtin_amdgcn_exp2f((m_run[j] - mn[j]) * sc);    // m_run = M_NONE, mn finite: 2^-huge = 0
        l_run[j] *= alpha;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[j][i][e] *= alpha;
        m_run[j] = mn[j];
      }
    }
  };
  // p = exp2(s c - m c) on element PAIRS: one v_pk_fma_f32 and one v_pk_add_f32 per two scores (at two waves per SIMD the loop is
  // bound by the number of VALU instructions issued, SQ counters in profiles/), raw v_exp_f32: p underflows to 0, no fix-up
  auto softmax = [&](f32x16 (&s)[NQ], bf16x8 (&pf)[NQ][2]) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const float nmc = -m_run[j] * sc;          // (m_run = M_NONE only while every score seen is -inf: exp2(-inf) = 0)
      const f32x2 sc2 = f32x2{sc, sc}, nmc2 = f32x2{nmc, nmc};
      f32x2 acc2 = f32x2{0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 t = __builtin_elementwise_fma(f32x2{s[j][e], s[j][e + 1]}, sc2, nmc2);
        const f32x2 pp = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        acc2 += pp;
        s[j][e] = pp.x; s[j][e + 1] = pp.y;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f32x8 pw;
#pragma unroll
        for (int e = 0; e < 8; ++e) pw[e] = s[j][8 * s2 + e];
        pf[j][s2] = __builtin_convertvector(pw, bf16x8);   // four v_cvt_pk_bf16_f32
      }
      l_run[j] += acc2.x + acc2.y;   // per-lane partial (this half-wave's keys); the halves are added once, at the end
    }
  };
  // V^T fragments: lane constants per (d tile, low / high 8-key group) once; a read then costs one add of the scalar slot base
  // (the 16-key step s2 rides in the instruction's offset field)
  unsigned vlane[NDT][2];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      const int dcol = 32 * dt + tr_dcol, key = tr_key + 8 * hl;      // (+ 32 st + 16 s2: multiples of 16 leave the swizzle alone)
      vlane[dt][hl] = b.lds0 + 3 * TILE + key * ROWB + Lay<D>::vchunk(key, dcol >> 3) * 16 + (dcol & 7) * 2;
    }
  auto vload = [&](s16x8 (&vf)[2][NDT], int slot, int st) {       // the V^T fragments of one half tile, a stage ahead of their MFMAs
    const unsigned vb = slot * TILE + st * 32 * ROWB;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const s16x4 v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(size_t)(vlane[dt][0] + vb + s2 * 16 * ROWB));
        const s16x4 v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(size_t)(vlane[dt][1] + vb + s2 * 16 * ROWB));
        vf[s2][dt] = s16x8{v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
      }
  };
  auto pv = [&](const bf16x8 (&pf)[NQ][2], const s16x8 (&vf)[2][NDT]) {   // O^T += V^T . P^T: every V^T fragment feeds NQ MFMAs
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int j = 0; j < NQ; ++j)
          oacc[j][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[s2][dt]), pf[j][s2], oacc[j][dt], 0, 0, 0);
  };
  // half tile (kt, st) with the next half tile (kt2, st2) behind it; kslot2 = LDS slot of K(kt2).  The two stages are ONE basic
  // block (the scheduler may interleave matrix and vector work freely); the rare cases - a next half tile that needs its mask, a
  // moving softmax reference - branch only behind it.  The row maxima of the next scores are therefore taken speculatively on the
  // unmasked values and redone after masking where a mask applies (causal diagonal region, ragged last tile).
  // kf holds the K fragments of the NEXT half tile (kt2, st2) on entry and those of the one after it, (kt3, st3) in K slot
  // kslot3, on exit (load_k3 = false at the end of the sequence): LDS reads are issued one stage ahead of their MFMAs.
  auto step = [&](f32x16 (&cur)[NQ], f32x16 (&nxt)[NQ], bf16x8 (&kf)[NKS], int kt, int st, int kt2, int st2, int kslot3, int st3,
                  bool load_k3) {
    bf16x8 pf[NQ][2];
    s16x8 vf[2][NDT];
    float mn[NQ];
    if constexpr (!(ATTN64_ABL & 32)) vload(vf, kt & 1, st);        // stage 1: V^T fragments of this half tile on their way ...
    if constexpr (!(ATTN64_ABL & 16)) qk(nxt, kf);                  //          matrix pipe on S(next) ...
    if constexpr (!(ATTN64_ABL & 64)) softmax(cur, pf);             //          ... VALU on P(this)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ATTN64_ABL & 32)) { if (load_k3) kload(kf, kslot3, st3); }   // stage 2: K fragments of the half tile after next ...
    if constexpr (!(ATTN64_ABL & 8)) pv(pf, vf);                   //          matrix pipe on P(this).V(this) ...
    if constexpr (!(ATTN64_ABL & 128)) rowmax(nxt, mn);            //          ... VALU on the row maxima of S(next)
    else { for (int j = 0; j < NQ; ++j) mn[j] = m_run[j]; }
    const int key0 = kt2 * KT + st2 * 32;
    bool any_mask = key0 + 32 > b.kv_len;
#pragma unroll
    for (int j = 0; j < NQ; ++j) any_mask |= CAUSAL && key0 + 31 > qs[j] + b.kv_off;
    if (any_mask) {               // wave-uniform
      mask(nxt, kt2, st2);
      rowmax(nxt, mn);
    }
    rescale(mn);
  };

  // ---- prologue: S(0, 0) ----------------------------------------------------------------------------------------------------
  f32x16 sA[NQ], sB[NQ];
  bf16x8 kf[NKS];
  tile_fence();                   // K(0), V(0), K(1) were issued by the caller
  {
    kload(kf, 0, 0);
    qk(sA, kf);
    kload(kf, 0, 1);
    mask(sA, 0, 0);
    float mn[NQ];
    rowmax(sA, mn);
    rescale(mn);
  }
  // Tiles 0 .. n-2 run both half steps with the next half tile behind each; the last tile is peeled (no branch inside the loop
  // that would leave the score / output registers in two states at its end).
  int kslot = 0;                  // kt % 3
  int kt = 0;
  for (; kt + 1 < b.n_tiles; ++kt) {
    // K(kt+1) and V(kt) have landed and are visible; K(kt-1)'s and V(kt-1)'s buffers are refilled
    tile_fence();
    const int kslot1 = kslot == 2 ? 0 : kslot + 1, kslot2 = kslot1 == 2 ? 0 : kslot1 + 1;
    if constexpr (!(ATTN64_ABL & 4)) {
      if (kt + 2 < b.n_tiles) dma.k(p, b, kt + 2, kslot2);
      dma.v(p, b, kt + 1, (kt + 1) & 1);
    }
    step(sA, sB, kf, kt, 0, kt, 1, kslot1, 0, true);        // this (kt,0), next (kt,1) [kf], then (kt+1,0) from K(kt+1)
    step(sB, sA, kf, kt, 1, kt + 1, 0, kslot1, 1, true);    // this (kt,1), next (kt+1,0) [kf], then (kt+1,1)
    kslot = kslot1;
  }
  tile_fence();
  step(sA, sB, kf, kt, 0, kt, 1, 0, 0, false);
  {
    bf16x8 pf[NQ][2];
    s16x8 vf[2][NDT];
    vload(vf, kt & 1, 1);
    softmax(sB, pf);
    pv(pf, vf);
  }

  // ---- normalise and store: lane (c,h) owns O[qs[j]+c][32*dt + 8*(e>>2) + 4h + (e&3)] ------------------------------------------
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const float l = half_sum(l_run[j]);
    const int qr = qs[j] + c;
    if (qr < b.q_lim && !(ATTN64_ABL & 1024)) {
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      bf16_t* op = p.o + (size_t)(b.row0 + qr) * p.ldo + (size_t)b.hq * D;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          u16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(oacc[j][dt][4 * e4 + e] * inv);
          *(u16x4*)(op + 32 * dt + 8 * e4 + 4 * h) = o;
        }
    }
  }
}

// NQW = 32-row sub-blocks per wave: 2 at D = 64 (256-row query blocks), 1 at D = 128 (128-row blocks: two sub-blocks' O^T and Q^T
// alone would take 192 of the 256 registers a lane has at two waves per SIMD, and hipcc keeps MFMA results it must touch with the
// VALU out of the accumulator file only below that budget - beyond it every score would be copied accumulator -> VGPR)
template <int D, bool CAUSAL, int NQW>
__global__ __launch_bounds__(256, 2) void attn_fwd64_kernel(const AttnArgs p) {
  constexpr int QB = 128 * NQW;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K slots 0 1 2 | V slots 0 1
  Blk b;
  const int tid = threadIdx.x;
  b.lane = tid & 63; b.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  b.c = b.lane & 31; b.h = b.lane >> 5;
  const int rows_all = p.q_end > 0 ? min(p.max_len, p.q_end) : p.max_len;
  const int nqb = (rows_all + QB - 1) / QB;
  // XCD-aware block order (attention.hip): each XCD gets a contiguous run of (sequence, head, query block), so the query blocks of
  // a head - and the heads of a GQA group - stream the same K/V through one L2
  int v;
  {
    const int total = (int)gridDim.x, bid = blockIdx.x, xcd = bid & 7, q8 = total >> 3, r8 = total & 7;
    v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int qb = v % nqb, grp = v / nqb;
  b.hq = grp % p.n_heads;
  const int seq = grp / p.n_heads;
  b.g = p.n_heads / p.n_kv_heads;
  const int hk = b.hq / b.g;
  b.row0 = p.cu[seq];
  b.len = p.cu[seq + 1] - b.row0;
  b.q_lim = p.q_end > 0 ? min(b.len, p.q_end) : b.len;
  b.q0 = (CAUSAL ? nqb - 1 - qb : qb) * QB;        // causal work grows with the block index: heaviest first
  if (b.q0 >= b.q_lim) return;
  if (p.q_tail > 0 && b.q0 + QB <= b.len - p.q_tail) return;   // none of this block's rows is consumed (last-layer row trimming)
  b.kv_off = p.kv_off ? p.kv_off[seq] : p.kv_len_offset;
  b.kv_len = b.len + b.kv_off;
  b.n_tiles = (b.kv_len + KT - 1) / KT;
  if (CAUSAL) b.n_tiles = min(b.n_tiles, (min(b.q0 + QB, b.q_lim) - 1 + b.kv_off) / KT + 1);
  if (ATTN64_ABL & 256) b.n_tiles = 1;
  const size_t kv_seq = p.kv_seq_stride ? (size_t)seq * p.kv_seq_stride : 0;
  b.kbase = p.k + (p.kv_seq_stride ? kv_seq : (size_t)b.row0 * p.ldk) + (size_t)hk * p.kv_head_stride;
  b.vbase = p.v + (p.kv_seq_stride ? kv_seq : (size_t)b.row0 * p.ldv) + (size_t)hk * p.kv_head_stride;
  b.smem = smem;
  b.lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
  b.sc = 1.4426950408889634f / p.post_div;        // exp2(s c - m c) = exp((s - m) / post_div)

  Stager<D> dma;
  dma.init(b.wave);
  dma.k(p, b, 0, 0);
  dma.v(p, b, 0, 0);
  if (b.n_tiles > 1) dma.k(p, b, 1, 1);

  const int first1 = b.q0 + 32 * (b.wave + 4), first0 = b.q0 + 32 * b.wave;
  if (NQW == 2 && first1 < b.q_lim) {
    attn64_run<D, CAUSAL, NQW>(p, b, dma);
  } else if (first0 < b.q_lim) {
    attn64_run<D, CAUSAL, 1>(p, b, dma);
  } else {
    // a wave without rows (short last block): it still moves its share of every tile and joins every barrier
    tile_fence();
    for (int kt = 0; kt < b.n_tiles; ++kt) {
      tile_fence();
      if (kt + 2 < b.n_tiles) dma.k(p, b, kt + 2, (kt + 2) % 3);
      if (kt + 1 < b.n_tiles) dma.v(p, b, kt + 1, (kt + 1) & 1);
    }
  }
}

template <int D, bool CAUSAL>
hipError_t launch64(const AttnArgs& a, hipStream_t s) {
  constexpr int NQW = D == 64 ? 2 : 1, QB = 128 * NQW;
  constexpr int LDS = 5 * KT * (D * 2);
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)attn_fwd64_kernel<D, CAUSAL, NQW>, LDS); e != hipSuccess) return e;
  const int rows = a.q_end > 0 ? (a.max_len < a.q_end ? a.max_len : a.q_end) : a.max_len;
  const int nqb = (rows + QB - 1) / QB;
  hipLaunchKernelGGL((attn_fwd64_kernel<D, CAUSAL, NQW>), dim3(nqb * a.n_heads * a.n_seq), dim3(256), LDS, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t aigv_launch_attention64(const AttnArgs& a, int head_dim, hipStream_t s) {
  if (head_dim == 64) return a.causal ? launch64<64, true>(a, s) : launch64<64, false>(a, s);
  return a.causal ? launch64<128, true>(a, s) : launch64<128, false>(a, s);
}
