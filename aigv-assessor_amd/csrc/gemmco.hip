// bf16 GEMM, 256x128x64 tile, 4 waves, TWO co-resident workgroups per CU (gfx950).
//
//   C[M,N] = epilogue(A[M,K] . W[N,K]^T), same operand layouts, epilogues and - bit for bit - the same results as gemm256.hip.
//
// Why a third tile kernel: gemm256.hip owns the whole CU (8 waves, 128 KB of LDS), so while a tile is in its prologue (tile mapping,
// first DMA round trip) or its epilogue (GELU / residual arithmetic, stores and their acknowledgement) the matrix pipes idle.  At
// K = 1024 (InternViT: 16 K-tiles per output tile) that is 23-36 % of a tile's lifetime (profiles/r3_gemm_stamps.txt).  Here a workgroup
// is four waves, one per SIMD, with a 256 x 128 output tile and an 80 KB LDS ring, so two workgroups share a CU: each SIMD holds one
// wave of either, the hardware issues the older wave's MFMAs first and the younger fills the gaps, and one workgroup's prologue /
// epilogue runs under the other's K loop.
//
// Bits: every output element is the same chain of v_mfma_f32_16x16x32_bf16 over the same 32-deep K chunks in the same order, with the
// same operand roles and lane positions as in gemm256.hip, and the epilogues apply the same operations at the same rounding points -
// torch.equal between the two kernels is a test (tests/test_gpu_ops.py).
//
// Geometry
//   4 waves = 2 (M, "group" g = wave >> 1) x 2 (N, wc = wave & 1); wave tile 128 x 64 = (2 m-halves x 4) x (2 n-halves x 2) MFMA tiles,
//   128 accumulator VGPRs.  A K-tile (64 deep) is 4 phases of 16 MFMAs, one 64 x 32 output quadrant each; the n-half order alternates
//   with the K-tile's parity so that the last phase of a tile and the first phase of the next never share a fragment register:
//     even tile: (mh0,nh0) (mh0,nh1) (mh1,nh1) (mh1,nh0)      odd tile: (mh0,nh1) (mh0,nh0) (mh1,nh0) (mh1,nh1)
//   LDS: a ring of 10 units x 8 KB.  A unit = 64 rows x 128 B (one 64-deep K-tile of 64 rows), filled by 8 global_load_lds
//   wave-instructions (2 per wave: full 128-B lines), 16-B chunk index XOR-swizzled with (row & 7) on the SOURCE address and on the
//   ds_read_b128 fragment reads.  Six units per K-tile, streamed in the order
//     A(g0,mh0) A(g1,mh0) W(wc0) W(wc1) | A(g0,mh1) A(g1,mh1)           unit u of the stream lives in slot u % 10
//
// Schedule of one wave, K-tile t (R = fragment reads, M = 16 MFMAs; the reads run one phase ahead of the MFMAs that use them):
//   lgkmcnt(0) vmcnt(4) s_barrier "Y(t)"   issue the first four units of tile t+1 (8 DMA)
//   R(phase 0, t)  M(phase 3, t-1)  R(phase 1, t)  M(phase 0, t)
//   lgkmcnt(0) vmcnt(8) s_barrier "X(t)"   issue the last two units of tile t+1 (4 DMA)
//   R(phase 2, t)  M(phase 1, t)  M(phase 2, t)                               (phase 3 reads nothing new)
// Every unit is requested a whole K-tile before the barrier that publishes it.
// Hazards
//   RAW  the first four units of tile t are issued behind Y(t-1) (tile 0: in the prologue) and the only DMA younger than them in front
//        of Y(t) are the last two units of tile t (4 instructions, issued behind X(t-1)): vmcnt(4) means this wave's parts of the four
//        have landed, and the barrier that every wave's have.  The last two units of tile t: the 8 instructions issued behind Y(t) are
//        younger, vmcnt(8) in front of X(t); on the last K-tile nothing younger exists and the wait is vmcnt(0).
//   WAR  a slot is refilled ten units later.  A(*, mh0) of tile t (read in R(phase 0, t) only) by the last two units of tile t+1, issued
//        behind X(t), which follows every wave's lgkmcnt(0); the other four units of tile t (last read in R(phase 2, t)) by the first
//        four units of tile t+2, issued behind Y(t+1), which follows every wave's lgkmcnt(0) behind R(phase 2, t).
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int CM = 256, CN = 128, CK = 64;
constexpr int CUNIT = 64 * 128;             // bytes per unit (64 rows x 64 bf16)
constexpr int CRING = 10;                   // ring slots
constexpr int CO_LDS = CRING * CUNIT;       // 80 KB: two workgroups fill a CU's 160 KB exactly
constexpr int CO_GROUP_M = 4;

#define CO_BARRIER() asm volatile("s_barrier" ::: "memory")

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <class F, int... Is>
__device__ __forceinline__ void co_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void co_for(F&& f) { co_for_impl(f, std::make_integer_sequence<int, N>{}); }

// LDS-DMA as in common.h's glds16_saddr, with M0 declared clobbered instead of saved and restored (two scalar moves fewer per request)
__device__ __forceinline__ void co_glds16(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory", "m0");
}

// VAR 0: the reads, DMA requests and MFMAs of a K-tile in blocks (first version, kept for A/B); VAR 1: interleaved - one fragment read or
// DMA request between consecutive MFMAs, so that a wave's own matrix work covers its load issue
// VAR 6: the LONE form for launches of at most one workgroup per CU (one clip's wo / w2 bodies: 256 tiles of 256 x 128 where the 256 x 256 kernel
//        has 128): eight waves, waves 0-3 multiply (and never issue a DMA request), waves 4-7 only issue the LDS-DMA requests - a lone 4-wave
//        workgroup has no neighbour to cover the ~60 issue cycles of each request, here the loader waves take them off the multiplying waves'
//        instruction streams.  Same ring, same barriers, same MFMA chains: the same bits.
template <int EPI, int VAR>
__global__ __launch_bounds__(VAR == 6 ? 512 : 256, 2) void gemmco_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool LONE = VAR == 6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = LONE && wave_id >= 4;
  const int wave = loader ? wave_id - 4 : wave_id;   // index among the multiplying (or the loading) waves
  const int g = wave >> 1, wc = wave & 1;

  // ---- tile mapping: XCD-aware bijective remap, then groups of CO_GROUP_M row tiles sweep the column tiles (or column groups sweep the rows) ----
  const int nbm = p.row_tab ? (p.tab_halves + 1) / 2 : (p.M + CM - 1) / CM, nbn = p.N / CN;
  const int nwg = nbm * nbn;
  int wg;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  if (p.order == 0) {
    const int per_group = CO_GROUP_M * nbn;
    const int grp = wg / per_group, first_m = grp * CO_GROUP_M;
    const int gsz = min(nbm - first_m, CO_GROUP_M);
    const int in_g = wg - grp * per_group;
    tm = first_m + in_g % gsz; tn = in_g / gsz;
  } else {
    const int gn = p.order;
    const int per_group = gn * nbm;
    const int grp = wg / per_group, first_n = grp * gn;
    const int gsz = min(nbn - first_n, gn);
    const int in_g = wg - grp * per_group;
    tn = first_n + in_g % gsz; tm = in_g / gsz;
  }
  const int n0 = tn * CN;
  // the two 128-row halves of the tile: (base row, valid rows), wave-uniform; a wave multiplies and stores rows of half g only but
  // helps to load both
  int hbs[2], hvs[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int b, v;
    if (p.row_tab) {
      const int hi = 2 * tm + h;
      b = 0; v = 0;
      if (hi < p.tab_halves) { b = p.row_tab[2 * hi]; v = p.row_tab[2 * hi + 1]; }
    } else {
      b = tm * CM + h * 128; v = min(p.M - b, 128);
      if (v <= 0) { b = p.M - 1; v = 0; }
    }
    hbs[h] = __builtin_amdgcn_readfirstlane(b);
    hvs[h] = __builtin_amdgcn_readfirstlane(v);
  }
  const int hb = g ? hbs[1] : hbs[0], hv = g ? hvs[1] : hvs[0];
  const int hend = hb + hv;
  auto relc = [&](int r) __attribute__((always_inline)) { return max(min(r, hv - 1), 0); };

  // ---- LDS-DMA: this wave fills rows 16 * wave .. + 15 of every unit (two 8-row wave-instructions of full 128-B lines) ----
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;
  const int nk = p.K / CK;
  const char* tileA[2] = {(const char*)(p.A + (size_t)hbs[0] * p.lda), (const char*)(p.A + (size_t)hbs[1] * p.lda)};
  const char* tileW = (const char*)(p.W + (size_t)n0 * p.ldw);
  unsigned offA[2][2][2];   // [half][mh][instr]
  unsigned offW[2];         // [instr], W rows of column group 0 (group 1: + 64 rows on the scalar base)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave * 16 + i * 8 + lr;   // row inside the unit
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int row = max(min(mh * 64 + r, hvs[h] - 1), 0);   // rows past the half's valid count are never stored: any readable row
        offA[h][mh][i] = (unsigned)row * (unsigned)p.lda * 2u + lc * 16;
      }
    offW[i] = (unsigned)r * (unsigned)p.ldw * 2u + lc * 16;
  }
  const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem + wave * 2048;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using I5 = std::integral_constant<int, 5>;
  // stream position POS of a tile: 0 A(g0,mh0)  1 A(g1,mh0)  2 W(0)  3 W(1)  4 A(g0,mh1)  5 A(g1,mh1); `slot` is wave-uniform
  auto dma = [&](int tile, auto POS, int slot) __attribute__((always_inline)) {
    constexpr int i = decltype(POS)::value;
    const unsigned dst = lds0 + slot * CUNIT;
    const size_t kb = (size_t)tile * (CK * 2);
    if constexpr (i == 2 || i == 3) {
      const char* base = tileW + (size_t)(i - 2) * 64 * p.ldw * 2 + kb;
      glds16_saddr(base, offW[0], dst);
      glds16_saddr(base, offW[1], dst + 1024);
    } else {
      constexpr int h = i & 1, mh = i >> 2;
      glds16_saddr(tileA[h] + kb, offA[h][mh][0], dst);
      glds16_saddr(tileA[h] + kb, offA[h][mh][1], dst + 1024);
    }
  };
  auto dma1 = [&](int tile, auto POS, auto INSTR, int slot) __attribute__((always_inline)) {   // one of the two requests of a unit
    constexpr int i = decltype(POS)::value, instr = decltype(INSTR)::value;
    const unsigned dst = lds0 + slot * CUNIT + instr * 1024;
    const size_t kb = (size_t)tile * (CK * 2);
    if constexpr (i == 2 || i == 3) co_glds16(tileW + (size_t)(i - 2) * 64 * p.ldw * 2 + kb, offW[instr], dst);
    else co_glds16(tileA[i & 1] + kb, offA[i & 1][i >> 2][instr], dst);
  };
  auto wrap = [](int s) __attribute__((always_inline)) { return s >= CRING ? s - CRING : s; };

  // ---- fragment read offsets (bytes inside a unit) ----
  const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
  int offF[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) offF[kh] = fr * 128 + (((kh * 4 + fq) ^ sw) * 16);   // + fragment row block * 2048

  f32x4 acc[2][4][2][2];   // [mh][mt][nh][nt]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][4][2];   // A(mh) [mt][kh]
  bf16x8 fb[2][2][2];   // W(nh) [nt][kh]

  auto read_a = [&](auto MH, int slot) __attribute__((always_inline)) {
    constexpr int mh = decltype(MH)::value;
    const char* sb = smem + slot * CUNIT;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fa[mh][mt][kh] = *(const bf16x8*)(sb + offF[kh] + mt * 2048);
  };
  auto read_b = [&](auto NH, int slot) __attribute__((always_inline)) {
    constexpr int nh = decltype(NH)::value;
    const char* sb = smem + slot * CUNIT + nh * 4096;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fb[nh][nt][kh] = *(const bf16x8*)(sb + offF[kh] + nt * 2048);
  };
  auto mfma16 = [&](auto MH, auto NH) __attribute__((always_inline)) {
    constexpr int mh = decltype(MH)::value, nh = decltype(NH)::value;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mh][mt][nh][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nh][nt][kh], fa[mh][mt][kh], acc[mh][mt][nh][nt], 0, 0, 0);
  };

#define CO_SB() __builtin_amdgcn_sched_barrier(0)
  // one K-tile.  PAR = tile parity (n-half order), FIRST = there is no previous tile whose last phase is still to be multiplied;
  // `base` = slot of the tile's first unit (6 t mod 10)
  auto tile_body = [&](int t, int base, auto PAR, auto FIRST) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    using NA = std::integral_constant<int, par>;        // n-half of phases 0 and 3
    using NB = std::integral_constant<int, par ^ 1>;    // n-half of phases 1 and 2 (= the previous tile's phases 0 and 3)
    const int sA0 = wrap(base + g), sW = wrap(base + 2 + wc), sA1 = wrap(base + 4 + g);
    // ---- Y(t): every read of tile t-1 is done; the first four units of tile t have landed (4 younger DMA: its last two units) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    CO_SB(); CO_BARRIER(); CO_SB();
    if (t + 1 < nk) {   // first four units of tile t+1 -> the slots of tile t-1's last four
      dma(t + 1, I0{}, wrap(base + 6)); dma(t + 1, I1{}, wrap(base + 7)); dma(t + 1, I2{}, wrap(base + 8)); dma(t + 1, I3{}, wrap(base + 9));
    }
    CO_SB();
    read_a(I0{}, sA0);
    read_b(NA{}, sW);
    CO_SB();
    if constexpr (!decltype(FIRST)::value) mfma16(I1{}, NB{});   // phase 3 of tile t-1: (mh1, that tile's first n-half)
    CO_SB();
    read_b(NB{}, sW);
    CO_SB();
    mfma16(I0{}, NA{});
    CO_SB();
    // ---- X(t): every read of A(*, mh0) of tile t is done; the last two units of tile t have landed (8 younger DMA, if any) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CO_SB(); CO_BARRIER(); CO_SB();
    if (t + 1 < nk) { dma(t + 1, I4{}, base); dma(t + 1, I5{}, wrap(base + 1)); }   // -> the slots of A(*, mh0) of tile t
    CO_SB();
    read_a(I1{}, sA1);
    CO_SB();
    mfma16(I0{}, NB{});
    CO_SB();
    mfma16(I1{}, NB{});
    CO_SB();
  };

  // single fragment reads / MFMAs for the interleaved schedule.  Read order = MFMA order (kh-major): A(mt 0..3), W(nt 0..1) of k-half 0,
  // then of k-half 1
  auto read_a1 = [&](auto MH, int slot, auto J) __attribute__((always_inline)) {   // J = kh * 4 + mt
    constexpr int mh = decltype(MH)::value, j = decltype(J)::value;
    fa[mh][j & 3][j >> 2] = *(const bf16x8*)(smem + slot * CUNIT + offF[j >> 2] + (j & 3) * 2048);
  };
  auto read_b1 = [&](auto NH, int slot, auto J) __attribute__((always_inline)) {   // J = kh * 2 + nt
    constexpr int nh = decltype(NH)::value, j = decltype(J)::value;
    fb[nh][j & 1][j >> 1] = *(const bf16x8*)(smem + slot * CUNIT + nh * 4096 + offF[j >> 1] + (j & 1) * 2048);
  };
  auto mfma1 = [&](auto MH, auto NH, auto I) __attribute__((always_inline)) {   // I = kh * 8 + mt * 2 + nt: every accumulator takes k-half 0 before k-half 1, as in mfma16
    constexpr int mh = decltype(MH)::value, nh = decltype(NH)::value, i = decltype(I)::value;
    constexpr int kh = i >> 3, mt = (i >> 1) & 3, nt = i & 1;
    acc[mh][mt][nh][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nh][nt][kh], fa[mh][mt][kh], acc[mh][mt][nh][nt], 0, 0, 0);
  };
  // Diagnostic builds only (-DAIGV_CO_DIAG, scripts/gemmco_diag.py): VAR 2 = no DMA requests and no vmcnt waits inside the K loop,
  // VAR 3 = no workgroup barriers either (results are garbage: what the loop costs without its memory stream / its rendezvous)
  //                                                       VAR 4 = every DMA request but no vmcnt wait (issue + traffic without the landing
  //                                                       latency), VAR 5 = the same with the A units only (4 of 6)
  // (VAR 6: the multiplying waves run the no-DMA stream for real - the loader waves own the requests and the vmcnt waits)
  constexpr bool DIAG_NODMA = VAR == 2 || VAR == 3 || LONE, DIAG_NOBAR = VAR == 3, DIAG_NOWAIT = VAR == 4 || VAR == 5, DIAG_AONLY = VAR == 5;
  auto tile_body_i = [&](int t, int base, auto PAR, auto FIRST) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    constexpr bool first = decltype(FIRST)::value;
    using NA = std::integral_constant<int, par>;
    using NB = std::integral_constant<int, par ^ 1>;
    const int sA0 = wrap(base + g), sW = wrap(base + 2 + wc), sA1 = wrap(base + 4 + g);
    const bool more = !DIAG_NODMA && t + 1 < nk;
    const int s6 = wrap(base + 6), s7 = wrap(base + 7), s8 = wrap(base + 8), s9 = wrap(base + 9), s1 = wrap(base + 1);
    // ---- Y(t) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!DIAG_NODMA && !DIAG_NOWAIT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    CO_SB(); if constexpr (!DIAG_NOBAR) CO_BARRIER(); CO_SB();
    // phase 3 of tile t-1 beside the reads of phase 0 and the first four units of tile t+1
    co_for<16>([&](auto I) {
      constexpr int i = decltype(I)::value;
      // reads 0..11: A k0 (4), W k0 (2), A k1 (4), W k1 (2)
      if constexpr (i < 4) read_a1(I0{}, sA0, std::integral_constant<int, i>{});
      else if constexpr (i < 6) read_b1(NA{}, sW, std::integral_constant<int, i - 4>{});
      else if constexpr (i < 10) read_a1(I0{}, sA0, std::integral_constant<int, i - 2>{});
      else if constexpr (i < 12) read_b1(NA{}, sW, std::integral_constant<int, i - 8>{});
      CO_SB();
      if constexpr (!first) mfma1(I1{}, NB{}, I);
      CO_SB();
      if constexpr ((i & 1) != 0) {
        if (more) {
          constexpr int u = i >> 2;
          using IN = std::integral_constant<int, (i >> 1) & 1>;
          if constexpr (u == 0) dma1(t + 1, I0{}, IN{}, s6);
          if constexpr (u == 1) dma1(t + 1, I1{}, IN{}, s7);
          if constexpr (u == 2 && !DIAG_AONLY) dma1(t + 1, I2{}, IN{}, s8);
          if constexpr (u == 3 && !DIAG_AONLY) dma1(t + 1, I3{}, IN{}, s9);
        }
        CO_SB();
      }
    });
    // phase 0 beside the reads of phase 1
    co_for<16>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (i < 4) { read_b1(NB{}, sW, I); CO_SB(); }
      mfma1(I0{}, NA{}, I);
      CO_SB();
    });
    // ---- X(t) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!DIAG_NOWAIT) {
      if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CO_SB(); if constexpr (!DIAG_NOBAR) CO_BARRIER(); CO_SB();
    // phase 1 beside the reads of phase 2 and the last two units of tile t+1
    co_for<16>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (i < 8) { read_a1(I1{}, sA1, I); CO_SB(); }
      mfma1(I0{}, NB{}, I);
      CO_SB();
      if constexpr ((i & 1) != 0 && i < 8) {
        if (more) {
          using IN = std::integral_constant<int, (i >> 1) & 1>;
          if constexpr ((i >> 2) == 0) dma1(t + 1, I4{}, IN{}, base);
          else dma1(t + 1, I5{}, IN{}, s1);
        }
        CO_SB();
      }
    });
    // phase 2
    co_for<16>([&](auto I) { mfma1(I1{}, NB{}, I); });
    CO_SB();
  };
  auto body = [&](int t, int base, auto PAR, auto FIRST) __attribute__((always_inline)) {
    if constexpr (VAR == 0) tile_body(t, base, PAR, FIRST); else tile_body_i(t, base, PAR, FIRST);
  };

  // ---- prologue: the six units of tile 0 ----
  if (!LONE || loader) { dma(0, I0{}, 0); dma(0, I1{}, 1); dma(0, I2{}, 2); dma(0, I3{}, 3); dma(0, I4{}, 4); dma(0, I5{}, 5); }
  if constexpr (LONE) {
    if (loader) {   // the loader waves: the waits, barriers and requests of tile_body_i, nothing else
      int lbase = 0;
      for (int t = 0; t < nk; ++t) {
        const bool more = t + 1 < nk;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        CO_SB(); CO_BARRIER(); CO_SB();   // Y(t)
        if (more) {
          dma(t + 1, I0{}, wrap(lbase + 6)); dma(t + 1, I1{}, wrap(lbase + 7)); dma(t + 1, I2{}, wrap(lbase + 8)); dma(t + 1, I3{}, wrap(lbase + 9));
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        CO_SB(); CO_BARRIER(); CO_SB();   // X(t)
        if (more) { dma(t + 1, I4{}, lbase); dma(t + 1, I5{}, wrap(lbase + 1)); }
        lbase = wrap(lbase + 6);
      }
      CO_BARRIER();   // the multiplying waves' "ring is free" barrier
      return;
    }
  }

  using T = std::true_type;
  using F = std::false_type;
  body(0, 0, I0{}, T{});
  int base = 6, t = 1;
  for (; t + 1 < nk; t += 2) {   // (odd, even) pairs: base advances by 12 = 2 mod 10
    body(t, base, I1{}, F{});
    body(t + 1, wrap(base + 6), I0{}, F{});
    base = wrap(base + 2);
  }
  if (t < nk) body(t, base, I1{}, F{});
  // the last tile's phase 3: (mh1, its first n-half)
  CO_SB();
  if ((nk - 1) & 1) mfma16(I1{}, I1{}); else mfma16(I1{}, I0{});
  CO_SB();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (VAR == 4 || VAR == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (diagnostic variants without in-loop waits)
  CO_BARRIER();   // every wave is past its last fragment read: the ring is free for the epilogue's staging
  CO_SB();

  // ---- LDS-staged epilogue (the operations and rounding points of gemm256.hip's epilogues) -------------------------------
  // Each wave owns 64 rows x 144 B of LDS.  Stage 1 applies the part of the epilogue that is a function of the accumulator only (bias,
  // rounding, GELU, layer-scale, SwiGLU) and writes bf16; stage 2 re-reads whole row segments, adds residual / position rows and
  // stores 16 B per lane.  The residual comes straight from global memory here: its latency hides under the co-resident workgroup.
  constexpr int ROWP = 144;
  constexpr int OC = (EPI == EPI_SWIGLU) ? 32 : 64;
  constexpr int CPR = OC / 8;
  char* st = smem + wave * (64 * ROWP);
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  const int ncol = n0 + wc * 64;
  u32x2 bcol[2][2];
  u32x2 scol[2][2];
  if constexpr (EPI == EPI_LS_RESID) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) scol[nh][nt] = *(const u32x2*)(p.ls + ncol + nh * 32 + nt * 16 + fq * 4);
  }
  if constexpr (EPI != EPI_SWIGLU) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bcol[nh][nt] = p.bias ? *(const u32x2*)(p.bias + ncol + nh * 32 + nt * 16 + fq * 4) : u32x2{0u, 0u};
  }
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      char* rowp = st + (mt * 16 + fr) * ROWP;
      if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          u32x2 o;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2 gt = f32x2{acc[mh][mt][nh][0][2 * h], acc[mh][mt][nh][0][2 * h + 1]};
            f32x2 up = f32x2{acc[mh][mt][nh][1][2 * h], acc[mh][mt][nh][1][2 * h + 1]};
            gt = rbf2(gt);
            up = rbf2(up);
            o[h] = pack_bf2(rbf2(silu2(gt)) * up);
          }
          *(u32x2*)(rowp + (nh * 16 + fq * 4) * 2) = o;
        }
      } else {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int cl = nh * 32 + nt * 16 + fq * 4;
            u32x2 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 v = f32x2{acc[mh][mt][nh][nt][2 * h], acc[mh][mt][nh][nt][2 * h + 1]};
              if (p.bias) v += unpack_bf2(bcol[nh][nt][h]);
              if constexpr (EPI == EPI_GELU) v = gelu_fast2(rbf2(v));
              if constexpr (EPI == EPI_LS_RESID) v = rbf2(v) * unpack_bf2(scol[nh][nt][h]);
              o[h] = pack_bf2(v);
            }
            *(u32x2*)(rowp + cl * 2) = o;
          }
      }
    }
    constexpr int RPI = 64 / CPR;
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {
      const int r = i * RPI + lane / CPR, ch = lane % CPR;
      const int m = hb + mh * 64 + r;
      u16x8 val = *(const u16x8*)(st + r * ROWP + ch * 16);
      if (m < hend) {
        const int n = (EPI == EPI_SWIGLU ? ncol / 2 : ncol) + ch * 8;
        size_t orow = (size_t)m;
        if constexpr (EPI == EPI_PATCH) {
          const int f = m / p.np, pi = m - f * p.np;
          orow = (size_t)m + f + 1;
          const u16x8 ps = *(const u16x8*)(p.pos + (size_t)(pi + 1) * p.N + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = f2bf(bf2f(val[e]) + bf2f(ps[e]));
        }
        if constexpr (EPI == EPI_RESID || EPI == EPI_LS_RESID) {
          const u16x8 rs = *(const u16x8*)(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = f2bf(bf2f(rs[e]) + bf2f(val[e]));
        }
        *(u16x8*)(p.C + orow * p.ldc + n) = val;
      }
    }
  }
  (void)relc;
}

template <int EPI, int VAR>
hipError_t launch_co_v(const GemmArgs& a, hipStream_t s) {
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemmco_kernel<EPI, VAR>, CO_LDS); e != hipSuccess) return e;
  const int nbm = a.row_tab ? (a.tab_halves + 1) / 2 : (a.M + CM - 1) / CM, nbn = a.N / CN;
  GemmArgs b = a;
  // tile order as in gemm256.hip: small weights -> an XCD owns row tiles and sweeps W; large weights -> an XCD owns a slice of W
  const bool big_w = (size_t)a.N * (size_t)a.K >= ((size_t)32 << 20);
  b.order = a.order_sel > 0 ? a.order_sel - 1 : (big_w && nbm >= 32 ? 8 : 0);
  hipLaunchKernelGGL((gemmco_kernel<EPI, VAR>), dim3(nbm * nbn), dim3(VAR == 6 ? 512 : 256), CO_LDS, s, b);
  return hipGetLastError();
}
// GemmArgs::variant_sel: 0 = the shipped schedule, 1 + v = schedule variant v (A/B: AIGV_TUNE_GEMM256_VARIANT / aigv_tune_gemm bits 4..6)
template <int EPI>
hipError_t launch_co(const GemmArgs& a, hipStream_t s) {
#ifdef AIGV_CO_DIAG
  if (a.variant_sel == 3) return launch_co_v<EPI, 2>(a, s);
  if (a.variant_sel == 4) return launch_co_v<EPI, 3>(a, s);
  if (a.variant_sel == 5) return launch_co_v<EPI, 4>(a, s);
  if (a.variant_sel == 6) return launch_co_v<EPI, 5>(a, s);
#endif
  if (a.variant_sel == 7) return launch_co_v<EPI, 6>(a, s);   // the lone form (eight waves, four of them loaders)
  return (a.variant_sel == 1) ? launch_co_v<EPI, 0>(a, s) : launch_co_v<EPI, 1>(a, s);
}

}  // namespace

bool aigv_gemmco_supported(const GemmArgs& a) { return a.N % CN == 0 && a.K % CK == 0 && a.M >= 1 && (!a.row_tab || a.tab_halves >= 1); }

hipError_t aigv_launch_gemmco(const GemmArgs& a, int epi, hipStream_t s) {
  if (!aigv_gemmco_supported(a)) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_STORE: return launch_co<EPI_STORE>(a, s);
    case EPI_GELU: return launch_co<EPI_GELU>(a, s);
    case EPI_LS_RESID: return launch_co<EPI_LS_RESID>(a, s);
    case EPI_RESID: return launch_co<EPI_RESID>(a, s);
    case EPI_SWIGLU: return launch_co<EPI_SWIGLU>(a, s);
    case EPI_PATCH: return launch_co<EPI_PATCH>(a, s);
  }
  return hipErrorInvalidValue;
}
