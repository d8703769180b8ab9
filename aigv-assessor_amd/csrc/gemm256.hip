// bf16 GEMM, 256x256x64 tile, 8 waves, phase-interleaved schedule for gfx950 (one workgroup per CU).
//
//   C[M,N] = epilogue(A[M,K] . W[N,K]^T), same operand layouts and epilogues as gemm.hip.
//
// Why a second kernel: the 128x128 / one-barrier-per-K-step kernel tops out near 900 TFLOP/s (every wave
// stalls together on the vmcnt(0)+barrier of each K step).  Here each SIMD holds two waves that alternate
// roles: while one multiplies (16 MFMAs), the other reads its next fragments from LDS and issues the LDS-DMA
// prefetch, separated by raw s_barriers; the DMA runs ~4 phases ahead behind a COUNTED vmcnt.
//
// Geometry
//   8 waves = 2 (M, "group" g = wave>>2) x 4 (N); wave tile 128 x 64 = (2 m-halves x 4) x (2 n-halves x 2) MFMA
//   16x16x32 tiles; 128 accumulator VGPRs.  A K-tile (64 deep) takes 4 phases, one 64x32 output quadrant each:
//     j=0: A(mh0) x B(nh0)   j=1: A(mh0) x B(nh1)   j=2: A(mh1) x B(nh1)   j=3: A(mh1) x B(nh0)
//   LDS: 2 buffers x 4 units x 16 KB = 128 KB.  A unit is what ONE phase reads (all waves):
//     V0 = A rows {g*128 + 0..63}, V1 = A rows {g*128 + 64..127}, V2 = W rows {wc*64 + 0..31}, V3 = W rows {wc*64 + 32..63}
//   each 128 rows x 128 B, filled by 16 global_load_lds wave-instructions (2 per wave), 16-B chunk index
//   XOR-swizzled with (row & 7) on the SOURCE address and on the ds_read_b128 fragment reads.
//
// Schedule (program order of one wave; group 1 runs one barrier behind group 0, so on every SIMD one wave is in its
// MFMA segment while the other is in its load segment):
//   LOAD(p):  ds_read the fragments phase p needs (j=0: A(mh0)+B(nh0), j=1: B(nh1), j=2: A(mh1), j=3: none)
//             issue the LDS-DMA of stream unit 6+p  (stream order per tile: V0, V2, V3, V1)
//             s_waitcnt vmcnt(8)                    (4 units stay in flight)
//   s_barrier ; MFMA(p): 16 x v_mfma_f32_16x16x32_bf16 ; s_barrier
// Hazards (intervals between consecutive barriers are numbered; group g runs LOAD(p) in interval 2p+g):
//   RAW  a unit read in LOAD(q) was waited for (vmcnt) by every wave at the end of its LOAD(q-1), i.e. before the
//        barrier that precedes the reader's interval - also across the one-barrier stagger.
//   WAR  unit X of tile t is last read in LOAD(4t+{0,0,1,2}) for {V0,V2,V3,V1}; its refill for tile t+2 is issued in
//        LOAD(4t+{2,3,4,5}) - at least 3 intervals after the last reader's interval, whose ds_reads completed
//        (lgkmcnt(0)) right after the barrier that ended it.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int UNIT = 128 * 128;            // bytes per unit (128 rows x 64 bf16)
constexpr int BUF = 4 * UNIT;              // one K-tile
constexpr int STAGE_ROWP = 144;            // residual epilogue: 16 staged rows x 144 B per wave, outside the K ring
constexpr int STAGE_BYTES = 16 * STAGE_ROWP;
constexpr int LDS_BYTES = 2 * BUF + 8 * STAGE_BYTES;   // 128 KB ring + 18 KB
constexpr int GROUP_M256 = 4;
// row tiles of a launch: from the half-tile table when there is one (GemmArgs::row_tab), else from M
inline int row_tiles(const GemmArgs& a) { return a.row_tab ? (a.tab_halves + 1) / 2 : (a.M + TM - 1) / TM; }
}
// tile order of a plain launch: GemmArgs::order_sel 0 = by weight size, 1 = row groups, 1 + g = groups of g column tiles
static inline int tile_order(const GemmArgs& a, bool big_w, int nbm) { return a.order_sel > 0 ? a.order_sel - 1 : (big_w && nbm >= 32 ? 4 : 0); }
namespace {

#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// Diagnostic build only (-DAIGV_GEMM_STAMP, scripts/gemm_stamp.py): cycles per wave in the prologue (tile mapping, first DMA, wait for the
// first units), the K loop and the epilogue, per epilogue kind and K-tile count class, summed over all waves of all launches.
// g_gemm_stamp[EPI][nk <= 16 ? 0 : 1][replica][0 waves, 1 total, 2 prologue, 3 K loop, 4 epilogue, 5 K tiles].  Never defined in the product build.
#ifdef AIGV_GEMM_STAMP
constexpr int GSTAMP_REPL = 1024;
__device__ unsigned long long g_gemm_stamp[8][2][GSTAMP_REPL][8];
#define GSTAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define GSTAMP(var)
#endif

__device__ __forceinline__ void wait_vm(int n) {   // n in {0,2,4,6,8}, wave-uniform
  if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// VAR bit 0: balanced LDS reads (B(nh0) of the NEXT tile is read in phase j=3 into an alternate register set; waits
//            become vmcnt(8,8,6,-)), steady-state tiles run without issue/wait branches
// VAR bit 1: no s_setprio around the MFMA cluster
// VAR bit 2: epilogue staged through LDS: each wave transposes its tile so that global loads/stores are 16 B per lane
//            over whole 128-B row segments (half the store instructions of the 8-B-per-lane direct form)
// FP8 (EPI_STORE only): A / W rows hold e4m3 bytes; the 128-byte K-tile rows, the LDS image and the DMA stream are unchanged (a
//      64-deep bf16 K-tile and a 128-deep fp8 K-tile are the same bytes), a phase issues eight v_mfma_scale_f32_16x16x128_f8f6f4 with
//      unit block scales instead of sixteen bf16 MFMAs, and the epilogue multiplies by the per-row and per-column fp32 scales.
// FUSE: one launch holds the BODY tiles of a row plan (workgroups [0, p.fuse_body_wg): epilogue EPI, the table's first p.tab_halves halves)
//      and the split-K SLICES of its tail tiles behind them (p.k_slices workgroups per tile of the p.fuse_tail_halves halves that follow in
//      the table: fp32 slabs, as EPI_PARTIAL) - for launches whose body does not fill its last round of CUs (one or two clips), where a
//      separate slice launch would run behind a half-empty chip.  Same slices, same slabs, same finalize pass: not one bit differs from the
//      two-launch form.
// The 16-byte row-segment store of the staged epilogues.  Product build: a plain global_store_dwordx4.  A/B build (-DAIGV_STORE_POLICY_AB,
// scripts/store_policy_ab.py): GemmArgs::variant_sel 5 / 6 / 7 pick the cache policy of that store - nt / sc1 / sc1 nt -
// per launch, so that one process can interleave the arms (the schedule is variant 1, the shipped one, in all of them).  Same bits.
__device__ __forceinline__ void store_row_segment(bf16_t* dst, const u16x8& val, int policy) {
#ifdef AIGV_STORE_POLICY_AB
  if (policy == 5) { asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(val) : "memory"); return; }
  if (policy == 6) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(val) : "memory"); return; }
  if (policy == 7) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(dst), "v"(val) : "memory"); return; }
#endif
  (void)policy;
  *(u16x8*)dst = val;
}

template <int EPI, int VAR, bool FP8 = false, bool FUSE = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs p) {
  static_assert(!FP8 || ((VAR & 4) != 0 && EPI != EPI_PATCH), "the fp8 form uses the LDS-staged epilogue (or writes split-K slabs)");
  static_assert(!FUSE || (!FP8 && EPI != EPI_PARTIAL && EPI != EPI_PATCH), "the fused form: bf16, a body epilogue");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  GSTAMP(gs_begin);

  // ---- tile mapping: XCD-aware bijective remap, then groups of GROUP_M256 row-tiles sweep the column tiles ----
  const int nbm = p.row_tab ? (p.tab_halves + 1) / 2 : (p.M + TM - 1) / TM, nbn = p.N / TN;
  const int n_body = nbm * nbn;
  // (FUSE) the blocks behind the body tiles are the tail tiles' K slices.  The role follows the RAW block index - the long body tiles are
  // dispatched first and dealt evenly over the XCDs - and each role has its own XCD-aware remap (blocks with equal index mod 8 share an XCD
  // within a role either way).
  const bool is_slice = FUSE && (int)blockIdx.x >= n_body;
  const int nwg = is_slice ? ((p.fuse_tail_halves + 1) / 2) * nbn * p.k_slices : n_body;
  int wg;
  {
    const int bid = is_slice ? (int)blockIdx.x - n_body : (int)blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const bool partial = (EPI == EPI_PARTIAL) || is_slice;
  int slice_idx = (EPI == EPI_PARTIAL) ? (int)blockIdx.y : 0;
  int tm, tn;
  if (is_slice) {   // K slice `slice_idx` of tail tile (tm, tn): the slices of a tile are neighbours
    slice_idx = wg % p.k_slices;
    const int tile = wg / p.k_slices;
    tn = tile % nbn; tm = tile / nbn;
  } else if (p.order == 0) {
    const int per_group = GROUP_M256 * nbn;
    const int grp = wg / per_group, first_m = grp * GROUP_M256;
    const int gsz = min(nbm - first_m, GROUP_M256);
    const int in_g = wg - grp * per_group;
    tm = first_m + in_g % gsz; tn = in_g / gsz;
  } else {
    const int gn = p.order;                     // column tiles per group
    const int per_group = gn * nbm;
    const int grp = wg / per_group, first_n = grp * gn;
    const int gsz = min(nbn - first_n, gn);
    const int in_g = wg - grp * per_group;
    tn = first_n + in_g % gsz; tm = in_g / gsz;
  }
  const int n0 = tn * TN;
  // The 128-row HALF of the tile this wave group works on (group g loads, multiplies and stores rows g*128 .. g*128+127 only): base
  // row `hb` and valid rows `hv`, wave-uniform.  Without a table the halves are rows tm*256 + g*128 of A; with GemmArgs::row_tab
  // every half is an independent (base row, valid rows) entry, so ONE launch covers whole tiles of many sequences whose rows start
  // anywhere (rows are independent: a row's bits do not depend on where in a tile it sits).
  int hb, hv;
  if (p.row_tab) {
    const int h = 2 * tm + g;
    hb = 0; hv = 0;
    if (is_slice) {   // the tail halves follow the body halves in the table
      if (h < p.fuse_tail_halves) { hb = p.row_tab[2 * (p.tab_halves + h)]; hv = p.row_tab[2 * (p.tab_halves + h) + 1]; }
    } else if (h < p.tab_halves) { hb = p.row_tab[2 * h]; hv = p.row_tab[2 * h + 1]; }
  } else {
    hb = tm * TM + g * 128; hv = min(p.M - hb, 128);
    if (hv <= 0) { hb = p.M - 1; hv = 0; }
  }
  hb = __builtin_amdgcn_readfirstlane(hb);
  hv = __builtin_amdgcn_readfirstlane(hv);
  const int hend = hb + hv;                                     // first row past the half: rows m >= hend are never stored
  auto relc = [&](int r) { return max(min(r, hv - 1), 0); };   // row inside the half, clamped to a readable one (rows >= hv are never stored)

  // ---- LDS-DMA source pointers: this wave fills unit rows 16*wave .. 16*wave+15 (two 8-row wave-instructions) ----
  // unit row r -> tile row:  V0: (r>>6)*128 + (r&63)   V1: +64   V2: (r>>5)*64 + (r&31)   V3: +32
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;
  // wave-uniform tile bases (SGPRs) + 32-bit per-lane byte offsets: global_load_lds saddr + voffset, no 64-bit VALU
  // split-K (EPI_PARTIAL): blockIdx.y picks the K slice
  const int nk = partial ? p.K / TK / p.k_slices : p.K / TK;
  const size_t kbase = partial ? (size_t)slice_idx * nk * TK : 0;
  const char* tileA = (const char*)(p.A + (size_t)hb * p.lda + kbase);
  const char* tileW = (const char*)(p.W + (size_t)n0 * p.ldw + kbase);
  unsigned off[4][2];   // [unit][instr], at k = 0
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave * 16 + i * 8 + lr;
    const int ra0 = r & 63, rb0 = (r >> 5) * 64 + (r & 31);     // (r >> 6 == g: a wave fills A rows of its own half only)
    off[0][i] = (unsigned)relc(ra0) * (unsigned)p.lda * 2u + lc * 16;
    off[1][i] = (unsigned)relc(ra0 + 64) * (unsigned)p.lda * 2u + lc * 16;
    off[2][i] = (unsigned)rb0 * (unsigned)p.ldw * 2u + lc * 16;
    off[3][i] = (unsigned)(rb0 + 32) * (unsigned)p.ldw * 2u + lc * 16;
  }
  const int n_units = 4 * nk;
  // one unit = two wave-instructions; `unit` and the LDS destination are compile-time / wave-uniform
  const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem + wave * 2048;
  auto dma = [&](int tile, int unit) {
    const char* base = (unit < 2 ? tileA : tileW) + (size_t)tile * (TK * 2);
    const unsigned dst = lds0 + (tile & 1) * BUF + unit * UNIT;
    glds16_saddr(base, off[unit][0], dst);
    glds16_saddr(base, off[unit][1], dst + 1024);
  };
  // stream position o within a tile: 0 -> V0, 1 -> V2, 2 -> V3, 3 -> V1 (o is a compile-time constant at every call)
  auto issue = [&](int tile, int o) {
    if (tile >= nk) return;
    dma(tile, (o == 0) ? 0 : (o == 1) ? 2 : (o == 2) ? 3 : 1);
  };

  // ---- residual prefetch (EPI_RESID / EPI_LS_RESID with the staged epilogue) ----
  // The wave's 128 x 64 residual sub-tile goes through LDS-DMA as two 8 KB halves (64 rows x 128 B, 16-B chunk XOR (row & 7)
  // on the source side): half mh = 0 into the K buffer the last tile does not use, issued from that tile's phases 1 and 2
  // (its units were last read >= 3 intervals earlier, the refill rule of the header); half mh = 1 into the last tile's own
  // buffer once the K loop has drained.  The HBM latency of the residual is then hidden behind MFMA work / the first half.
  // (the fp8 form adds its residual in stage 2 of the staged epilogue instead: the prefetching path has no registers left for the scales)
  constexpr bool RESID_PF = ((VAR & 4) != 0) && (EPI == EPI_RESID || EPI == EPI_LS_RESID) && !FP8;
  const char* rtile = RESID_PF ? (const char*)(p.resid + (size_t)hb * p.ldr + n0 + wc * 64) : nullptr;
  auto resid_dma = [&](int mh, int i0, int cnt) {   // instructions i0 .. i0+cnt-1 of half mh (8 rows each)
    const int buf = (mh == 0) ? ((nk - 1) & 1) ^ 1 : ((nk - 1) & 1);
    const unsigned dst = (unsigned)(size_t)(LDS_AS char*)smem + buf * BUF + wave * 8192;
#pragma unroll
    for (int i = i0; i < i0 + cnt; ++i) {
      const int r = i * 8 + lr;                                            // row inside the half
      const int row = relc(mh * 64 + r);                                   // clamp: rows past the half's valid count are never stored
      glds16_saddr(rtile, (unsigned)row * (unsigned)p.ldr * 2u + (unsigned)(((lane & 7) ^ (r & 7)) * 16), dst + i * 1024);
    }
  };

  // ---- fragment read offsets (bytes inside a unit) ----
  const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
  int offA[2], offB[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    // bf16: k-half kh, 16-B chunk fq of it; fp8: k-group fq owns bytes [32 fq, 32 fq + 32) = chunks 2 fq, 2 fq + 1
    const int phys = (FP8 ? (2 * fq + kh) : (kh * 4 + fq)) ^ sw;
    offA[kh] = (g * 64 + fr) * 128 + phys * 16;     // + mt*16*128
    offB[kh] = (wc * 32 + fr) * 128 + phys * 16;    // + nt*16*128
  }

  f32x4 acc[2][4][2][2];   // [mh][mt][nh][nt]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fb0[2][2][2], fb1[2][2];   // A(mh) [mt][kh]; B(nh0) [set][nt][kh]; B(nh1) [nt][kh]
  constexpr bool BAL = (VAR & 1) != 0;
  constexpr bool PRIO = (VAR & 2) == 0;

  auto read_b0 = [&](const char* sb, auto SET) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fb0[decltype(SET)::value][nt][kh] = *(const bf16x8*)(sb + 2 * UNIT + offB[kh] + nt * 2048);
  };
  auto read_b1 = [&](const char* sb) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fb1[nt][kh] = *(const bf16x8*)(sb + 3 * UNIT + offB[kh] + nt * 2048);
  };
  auto read_a = [&](const char* sb, int unit) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) fa[mt][kh] = *(const bf16x8*)(sb + unit * UNIT + offA[kh] + mt * 2048);
  };
  typedef int v8i_t __attribute__((ext_vector_type(8)));
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  auto cat = [](bf16x8 lo, bf16x8 hi) {   // two 16-B fragment reads = the 32 e4m3 bytes of one lane's k-group
    const v4i_t l = __builtin_bit_cast(v4i_t, lo), h = __builtin_bit_cast(v4i_t, hi);
    return v8i_t{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
  };
  auto mfma16 = [&](auto J, auto SET) {
    constexpr int j = decltype(J)::value, st = decltype(SET)::value;
    if constexpr (FP8) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const v8i_t fav = cat(fa[mt][0], fa[mt][1]);
          if constexpr (j == 0) acc[0][mt][0][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat(fb0[st][nt][0], fb0[st][nt][1]), fav, acc[0][mt][0][nt], 0, 0, 0, 127, 0, 127);
          if constexpr (j == 1) acc[0][mt][1][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat(fb1[nt][0], fb1[nt][1]), fav, acc[0][mt][1][nt], 0, 0, 0, 127, 0, 127);
          if constexpr (j == 2) acc[1][mt][1][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat(fb1[nt][0], fb1[nt][1]), fav, acc[1][mt][1][nt], 0, 0, 0, 127, 0, 127);
          if constexpr (j == 3) acc[1][mt][0][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat(fb0[st][nt][0], fb0[st][nt][1]), fav, acc[1][mt][0][nt], 0, 0, 0, 127, 0, 127);
        }
      return;
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if constexpr (j == 0) acc[0][mt][0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[st][nt][kh], fa[mt][kh], acc[0][mt][0][nt], 0, 0, 0);
          if constexpr (j == 1) acc[0][mt][1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[nt][kh], fa[mt][kh], acc[0][mt][1][nt], 0, 0, 0);
          if constexpr (j == 2) acc[1][mt][1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[nt][kh], fa[mt][kh], acc[1][mt][1][nt], 0, 0, 0);
          if constexpr (j == 3) acc[1][mt][0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[st][nt][kh], fa[mt][kh], acc[1][mt][0][nt], 0, 0, 0);
        }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // one K-tile = 4 phases.  SET = register set holding this tile's B(nh0); STEADY = every issue of the tile is in range
  auto tile_body = [&](int t, auto SET, auto STEADY) {
    constexpr int st = decltype(SET)::value;
    constexpr bool steady = decltype(STEADY)::value;
    using NSET = std::integral_constant<int, st ^ 1>;
    const char* sb = smem + (t & 1) * BUF;
    const char* sbn = smem + ((t + 1) & 1) * BUF;
    // after LOAD(p) the stream must have landed up to need(p); allowed in-flight units = min(6+p, n_units-1) - need(p)
    auto wait_after = [&](int ph, int need) {
      if constexpr (steady) { (void)ph; (void)need; }
      else wait_vm(2 * max(0, min(6 + ph, n_units - 1) - need));
    };
#define PHASE_TAIL(J)                                 \
    __builtin_amdgcn_sched_barrier(0);                \
    RAW_BARRIER();                                    \
    __builtin_amdgcn_sched_barrier(0);                \
    mfma16(J{}, SET);                               \
    __builtin_amdgcn_sched_barrier(0);                \
    RAW_BARRIER();                                    \
    __builtin_amdgcn_sched_barrier(0);
    const int ph = 4 * t;
    // ---- j = 0 ----
    if constexpr (!BAL) { read_b0(sb, SET); __builtin_amdgcn_sched_barrier(0); }
    read_a(sb, 0);
    if constexpr (steady) dma(t + 1, 3); else issue(t + 1, 2);
    if constexpr (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_after(ph, ph + 2);
    PHASE_TAIL(I0)
    // the last tile of a residual epilogue: the K stream has nothing left to issue; phases 1 and 2 issue the first residual half
    // instead (4 LDS-DMA each).  The stream itself has fully landed after phase 1's wait (vmcnt(4) = only those 4 younger).
    const bool last_r = RESID_PF && !steady && (t == nk - 1) && !is_slice;   // (a fused K slice stores fp32 sums: no residual to fetch)
    // ---- j = 1 ----
    read_b1(sb);
    if constexpr (steady) dma(t + 1, 1); else issue(t + 1, 3);
    if constexpr (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (last_r) { resid_dma(0, 0, 4); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else wait_after(ph + 1, ph + 3);
    PHASE_TAIL(I1)
    // ---- j = 2 ----
    read_a(sb, 1);
    if constexpr (steady) dma(t + 2, 0); else issue(t + 2, 0);
    if constexpr (steady) { if constexpr (BAL) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else if (last_r) resid_dma(0, 4, 4);
    else wait_after(ph + 2, BAL ? ph + 5 : ph + 4);
    PHASE_TAIL(I2)
    // ---- j = 3 ----
    if constexpr (BAL) { if (steady || t + 1 < nk) read_b0(sbn, NSET{}); }
    if constexpr (steady) dma(t + 2, 2); else issue(t + 2, 1);
    if constexpr (steady) { if constexpr (!BAL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else if (!last_r) wait_after(ph + 3, BAL ? ph + 4 : ph + 5);
    PHASE_TAIL(I3)
#undef PHASE_TAIL
  };

  // ---- prologue: stream units 0..5, then make units 0,1 (V0, V2 of tile 0) visible ----
  issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3); issue(1, 0); issue(1, 1);
  wait_vm(2 * max(0, min(6, n_units) - 2));
  RAW_BARRIER();
  if constexpr (BAL) read_b0(smem, I0{});   // B(nh0) of tile 0; later tiles get theirs in the previous tile's phase 3
  if (g == 1) RAW_BARRIER();   // stagger: group 1 runs one barrier behind group 0

  using T = std::true_type;
  using F = std::false_type;
  GSTAMP(gs_loop);
  int t = 0;
  for (; t + 3 < nk; t += 2) {   // tiles t, t+1 issue units of tiles <= t+3: all in range
    tile_body(t, I0{}, T{});
    tile_body(t + 1, I1{}, T{});
  }
  for (; t < nk; t += 2) {
    tile_body(t, I0{}, F{});
    if (t + 1 < nk) tile_body(t + 1, I1{}, F{});
  }
  if (g == 0) RAW_BARRIER();   // balance the stagger barrier
#ifdef AIGV_GEMM_STAMP
  asm volatile("" :: "v"(acc[0][0][0][0][0]), "v"(acc[1][3][1][1][3]));
  const unsigned long long gs_epi = __builtin_readcyclecounter();
  auto gstamp_finish = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the epilogue's stores have been acknowledged: what the workgroup's exit waits for
    const unsigned long long gs_end = __builtin_readcyclecounter();
    if (lane == 0) {
      unsigned long long* d = g_gemm_stamp[EPI < 8 ? EPI : 7][nk <= 16 ? 0 : 1][blockIdx.x % GSTAMP_REPL];
      atomicAdd(d + 0, 1ull); atomicAdd(d + 1, gs_end - gs_begin); atomicAdd(d + 2, gs_loop - gs_begin);
      atomicAdd(d + 3, gs_epi - gs_loop); atomicAdd(d + 4, gs_end - gs_epi); atomicAdd(d + 5, (unsigned long long)nk);
    }
  };
#define GSTAMP_FINISH() gstamp_finish()
#else
#define GSTAMP_FINISH()
#endif

  auto store_partial = [&]() __attribute__((always_inline)) {
    // ---- split-K slice: fp32 partial sums into slab `slice_idx` ([M][N], the layout gemm_finalize_kernel sums) ----
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = hb + mh * 64 + mt * 16 + fr;
        if (m >= hend) continue;
        // slab rows: the output row itself, or (table form) the compact row (half index) * 128 + row inside the half
        const size_t srow = p.row_tab ? (size_t)((2 * tm + g) * 128 - hb) + m : (size_t)m;
        const size_t slab_rows = p.row_tab ? (size_t)(FUSE ? (p.fuse_tail_halves + 1) / 2 : nbm) * TM : (size_t)p.M;
        float* row = p.part + ((size_t)slice_idx * slab_rows + srow) * p.N + n0 + wc * 64 + fq * 4;
        // fp8: the slabs hold SCALED partial sums ((acc * row scale) * column scale is linear in acc), so the finalize pass is the bf16 one
        const float rs = FP8 ? p.row_scale[m] : 1.f;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            f32x4 v = acc[mh][mt][nh][nt];
            if constexpr (FP8) v = (v * rs) * *(const f32x4*)(p.col_scale + n0 + wc * 64 + nh * 32 + nt * 16 + fq * 4);
            *(f32x4*)(row + nh * 32 + nt * 16) = v;
          }
      }
  };
  if constexpr (FUSE) {
    if (is_slice) {
      store_partial();
      GSTAMP_FINISH();
      return;
    }
  }
  if constexpr (EPI == EPI_PARTIAL) {
    store_partial();
    GSTAMP_FINISH();
    return;
  } else if constexpr (RESID_PF) {
    // ---- residual epilogue: residual from LDS (prefetched), output staged 16 rows at a time in the wave's private region ----
    // No compiler-visible global LOAD may appear here: hipcc would wait for it with a vmcnt that also drains the residual DMA
    // still in flight.  The per-column bias / layer-scale vectors are therefore loaded by inline asm as well and tied to the wait.
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    // (the asm outputs are defined by exactly one asm statement each - no zero-init, no conditional: a phi would let the
    //  compiler copy a register the load has not written yet)
    u32x2 bcol[2][2], scol[2][2];
    const bool has_bias = p.bias != nullptr;
    const bf16_t* bvec = has_bias ? p.bias : p.W;   // without a bias: any readable address, the values are not used
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + wc * 64 + nh * 32 + nt * 16 + fq * 4;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(bcol[nh][nt]) : "v"(bvec + n) : "memory");
        if constexpr (EPI == EPI_LS_RESID) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(scol[nh][nt]) : "v"(p.ls + n) : "memory");
      }
    resid_dma(1, 0, 8);
    // everything older than the 8 DMA just issued has landed: bias, layer-scale, residual half 0
    if constexpr (EPI == EPI_LS_RESID)
      asm volatile("s_waitcnt vmcnt(8)"
                   : "+v"(bcol[0][0]), "+v"(bcol[0][1]), "+v"(bcol[1][0]), "+v"(bcol[1][1]), "+v"(scol[0][0]), "+v"(scol[0][1]),
                     "+v"(scol[1][0]), "+v"(scol[1][1])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(bcol[0][0]), "+v"(bcol[0][1]), "+v"(bcol[1][0]), "+v"(bcol[1][1]) : : "memory");
    char* st = smem + 2 * BUF + wave * STAGE_BYTES;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      // half 1: its 8 DMA are older than the 8 stores of half 0, so at most 8 outstanding operations means they have landed
      if (mh == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      const char* rl = smem + ((mh == 0) ? (((nk - 1) & 1) ^ 1) : ((nk - 1) & 1)) * BUF + wave * 8192;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int cl = nh * 32 + nt * 16 + fq * 4;
            const int r = mt * 16 + fr;
            const u32x2 rr = *(const u32x2*)(rl + r * 128 + (((cl >> 3) ^ (r & 7)) * 16) + (fq & 1) * 8);
            u32x2 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // element pairs (0,1), (2,3): packed fp32 math, one cvt_pk per rounding point
              f32x2 v = f32x2{acc[mh][mt][nh][nt][2 * h], acc[mh][mt][nh][nt][2 * h + 1]};
              if (has_bias) v += unpack_bf2(bcol[nh][nt][h]);
              v = rbf2(v);
              if constexpr (EPI == EPI_LS_RESID) v = rbf2(v * unpack_bf2(scol[nh][nt][h]));
              o[h] = pack_bf2(unpack_bf2(rr[h]) + v);
            }
            *(u32x2*)(st + fr * STAGE_ROWP + cl * 2) = o;
          }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = i * 8 + (lane >> 3), ch = lane & 7;
          const int m = hb + mh * 64 + mt * 16 + r;
          const u16x8 val = *(const u16x8*)(st + r * STAGE_ROWP + ch * 16);
          if (m < hend) store_row_segment(p.C + (size_t)m * p.ldc + n0 + wc * 64 + ch * 8, val, p.variant_sel);
        }
      }
    }
    GSTAMP_FINISH();
    return;
  } else if constexpr ((VAR & 4) != 0) {
    // ---- LDS-staged epilogue ------------------------------------------------------------------------------------
    // All DMA has landed (the tail waits reach vmcnt(0)) and every wave is past its last fragment read (final barriers), so
    // the tile buffers are free.  Each wave owns 64 rows x 144 B of LDS (row padded by 16 B: ds_write_b64 of the MFMA layout
    // and ds_read_b128 of the row layout stay 16-B aligned and at most 2-way conflicted).  Stage 1 applies the part of the
    // epilogue that is a function of the accumulator only (bias, rounding, GELU, layer-scale, SwiGLU) and writes bf16;
    // stage 2 re-reads whole row segments, adds residual / position rows and stores 16 B per lane.
    constexpr int ROWP = 144;
    constexpr int OC = (EPI == EPI_SWIGLU) ? 32 : 64;       // output columns per wave
    constexpr int CPR = OC / 8;                             // 16-B chunks per staged row
    char* st = smem + wave * (64 * ROWP);
    // per-column bias depends on (nh, nt, fq) only: loaded once, not once per row block.  (EPI_RESID / EPI_LS_RESID take the
    // prefetching epilogue above when VAR has bit 2, so this path carries no layer-scale.)
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    static_assert(FP8 || (EPI != EPI_RESID && EPI != EPI_LS_RESID), "bf16 residual epilogues use the prefetching path");
    u32x2 bcol[2][2];
    u32x2 scol[2][2];   // layer-scale (fp8 LS_RESID only)
    if constexpr (EPI == EPI_LS_RESID) {
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) scol[nh][nt] = *(const u32x2*)(p.ls + n0 + wc * 64 + nh * 32 + nt * 16 + fq * 4);
    }
    if constexpr (EPI != EPI_SWIGLU) {
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int n = n0 + wc * 64 + nh * 32 + nt * 16 + fq * 4;
          bcol[nh][nt] = p.bias ? *(const u32x2*)(p.bias + n) : u32x2{0u, 0u};
        }
    }
    f32x4 wsc[2][2];   // fp8: per-column scales of this lane's 4 + 4 + 4 + 4 columns
    if constexpr (FP8) {
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wsc[nh][nt] = *(const f32x4*)(p.col_scale + n0 + wc * 64 + nh * 32 + nt * 16 + fq * 4);
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        char* rowp = st + (mt * 16 + fr) * ROWP;
        // element pairs (0,1), (2,3): packed fp32 math, one v_cvt_pk_bf16_f32 per rounding point
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            u32x2 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 gt = f32x2{acc[mh][mt][nh][0][2 * h], acc[mh][mt][nh][0][2 * h + 1]};
              f32x2 up = f32x2{acc[mh][mt][nh][1][2 * h], acc[mh][mt][nh][1][2 * h + 1]};
              if constexpr (FP8) {
                const float rs = p.row_scale[hb + relc(mh * 64 + mt * 16 + fr)];
                gt = (gt * rs) * f32x2{wsc[nh][0][2 * h], wsc[nh][0][2 * h + 1]};
                up = (up * rs) * f32x2{wsc[nh][1][2 * h], wsc[nh][1][2 * h + 1]};
              }
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
              const int cl = nh * 32 + nt * 16 + fq * 4;       // column inside the wave's 64
              u32x2 o;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                f32x2 v = f32x2{acc[mh][mt][nh][nt][2 * h], acc[mh][mt][nh][nt][2 * h + 1]};
                if constexpr (FP8) {   // (acc * row scale) * column scale, fp32
                  const float rs = p.row_scale[hb + relc(mh * 64 + mt * 16 + fr)];
                  v = (v * rs) * f32x2{wsc[nh][nt][2 * h], wsc[nh][nt][2 * h + 1]};
                }
                if (p.bias) v += unpack_bf2(bcol[nh][nt][h]);   // wave-uniform branch
                if constexpr (EPI == EPI_GELU) v = gelu_fast2(rbf2(v));
                if constexpr (EPI == EPI_LS_RESID) v = rbf2(v) * unpack_bf2(scol[nh][nt][h]);
                o[h] = pack_bf2(v);
              }
              *(u32x2*)(rowp + cl * 2) = o;
            }
        }
      }
      // stage 2: lane -> (row = i*RPI + lane / CPR, chunk = lane % CPR)
      constexpr int RPI = 64 / CPR;                          // rows per wave-instruction (8 or 16)
#pragma unroll
      for (int i = 0; i < 64 / RPI; ++i) {
        const int r = i * RPI + lane / CPR, ch = lane % CPR;
        const int m = hb + mh * 64 + r;
        u16x8 val = *(const u16x8*)(st + r * ROWP + ch * 16);
        if (m < hend) {
          const int n = (EPI == EPI_SWIGLU ? (n0 + wc * 64) / 2 : n0 + wc * 64) + ch * 8;
          size_t orow = (size_t)m;
          if constexpr (EPI == EPI_PATCH) {
            const int f = m / p.np, pi = m - f * p.np;
            orow = (size_t)m + f + 1;
            const u16x8 ps = *(const u16x8*)(p.pos + (size_t)(pi + 1) * p.N + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = f2bf(bf2f(val[e]) + bf2f(ps[e]));
          }
          if constexpr (EPI == EPI_RESID || EPI == EPI_LS_RESID) {   // fp8 form only (bf16 takes the prefetching path)
            const u16x8 rs = *(const u16x8*)(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = f2bf(bf2f(rs[e]) + bf2f(val[e]));
          }
          store_row_segment(p.C + orow * p.ldc + n, val, p.variant_sel);
        }
      }
    }
    GSTAMP_FINISH();
    return;
  }
  // ---- epilogue: lane owns C[m][n .. n+3] ----
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = hb + mh * 64 + mt * 16 + fr;
      if (m >= hend) continue;
      if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const int n = (n0 + wc * 64 + nh * 32) / 2 + fq * 4;
          u16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gt = rbf(acc[mh][mt][nh][0][e]), up = rbf(acc[mh][mt][nh][1][e]);
            o[e] = f2bf(rbf(silu_f(gt)) * up);
          }
          *(u16x4*)(p.C + (size_t)m * p.ldc + n) = o;
        }
      } else {
        size_t orow = (size_t)m;
        const bf16_t* posrow = nullptr;
        if constexpr (EPI == EPI_PATCH) {
          const int f = m / p.np, pi = m - f * p.np;
          orow = (size_t)m + f + 1;
          posrow = p.pos + (size_t)(pi + 1) * p.N;
        }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wc * 64 + nh * 32 + nt * 16 + fq * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mh][mt][nh][nt][e];
            if (p.bias) {
              const u16x4 b = *(const u16x4*)(p.bias + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += bf2f(b[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rbf(gelu_fast(v[e]));
            }
            if constexpr (EPI == EPI_LS_RESID) {
              const u16x4 s = *(const u16x4*)(p.ls + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rbf(v[e] * bf2f(s[e]));
            }
            if constexpr (EPI == EPI_LS_RESID || EPI == EPI_RESID) {
              const u16x4 r = *(const u16x4*)(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rbf(bf2f(r[e]) + v[e]);
            }
            if constexpr (EPI == EPI_PATCH) {
              const u16x4 ps = *(const u16x4*)(posrow + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rbf(v[e] + bf2f(ps[e]));
            }
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
            *(u16x4*)(p.C + orow * p.ldc + n) = o;
          }
      }
    }
  GSTAMP_FINISH();
}

template <int EPI, int VAR>
hipError_t launch256(const GemmArgs& a, hipStream_t s) {
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemm256_kernel<EPI, VAR>, LDS_BYTES); e != hipSuccess) return e;
  const int nbm = row_tiles(a), nbn = a.N / TN;
  GemmArgs b = a;
  // Tile order.  The XCD-aware remap hands each XCD a contiguous run of tiles; WHICH operand an XCD then owns decides what is fetched
  // eight times over.  Row groups (order 0): an XCD owns 4 row tiles and sweeps all of W - right when W is small (InternViT: 2-8 MB).
  // Column groups (order g): an XCD owns a slice of W and sweeps the rows - each weight byte is wanted by one XCD only, a few times in
  // quick succession, and the operand all XCDs share is A, which the Infinity Cache holds; for the large InternLM2 matrices (w1|w3
  // 235 MB, w2 117 MB): w2 718 -> 694 us, w1|w3 1398 -> 1380 us isolated, 101.5-101.9 -> 100.4 ms of GEMM time per step
  // (profiles/r2_gemm_tile_order.txt; the L2<->fabric byte count is the same either way - what changes is how much of it reaches HBM).
  // (measured at M = 8704; with M = 4281 - one 16-frame clip at the 26B widths - the row order was 0.3 % ahead, so short problems keep it)
  b.order = tile_order(a, (size_t)a.N * (size_t)a.K >= ((size_t)32 << 20), nbm);
  hipLaunchKernelGGL((gemm256_kernel<EPI, VAR>), dim3(nbm * nbn), dim3(512), LDS_BYTES, s, b);
  return hipGetLastError();
}

template <int VAR>
hipError_t launch256v(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch256<EPI_STORE, VAR>(a, s);
    case EPI_GELU: return launch256<EPI_GELU, VAR>(a, s);
    case EPI_LS_RESID: return launch256<EPI_LS_RESID, VAR>(a, s);
    case EPI_RESID: return launch256<EPI_RESID, VAR>(a, s);
    case EPI_SWIGLU: return launch256<EPI_SWIGLU, VAR>(a, s);
    case EPI_PATCH: return launch256<EPI_PATCH, VAR>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

#ifdef AIGV_GEMM_STAMP
extern "C" int aigv_debug_gemm_stamps(unsigned long long* out /*[8][2][8]*/, int reset) {   // diagnostic build only
  static unsigned long long host[8][2][GSTAMP_REPL][8];
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (out) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamp), sizeof host) != hipSuccess) return -2;
    for (int e = 0; e < 8; ++e)
      for (int c = 0; c < 2; ++c)
        for (int i = 0; i < 8; ++i) {
          unsigned long long t = 0;
          for (int r = 0; r < GSTAMP_REPL; ++r) t += host[e][c][r][i];
          out[(e * 2 + c) * 8 + i] = t;
        }
  }
  if (reset) {
    for (auto& a : host) for (auto& b : a) for (auto& c : b) for (auto& d : c) d = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp), host, sizeof host) != hipSuccess) return -2;
  }
  return 0;
}
#endif

// (default schedule = variant 1: balanced reads + no s_setprio + LDS-staged epilogue, fastest in interleaved A/B: profiles/r1_gemm_variants.txt)

bool aigv_gemm256_supported(const GemmArgs& a) { return a.N % TN == 0 && a.K % TK == 0 && a.M >= 1 && (!a.row_tab || a.tab_halves >= 1); }

// split-K slices of the 256 kernel: grid.y = a.k_slices workgroups per tile, each writes fp32 partial sums of its K range into
// a.part[slice][M][N] (summed in slice order by gemm_finalize_kernel, gemm.hip)
// fp8: a.K / a.lda / a.ldw arrive in e4m3 ELEMENTS; the kernel addresses the same bytes as pairs (its bf16_t unit)
template <int EPI>
hipError_t launch256_fp8(const GemmArgs& b, hipStream_t s) {
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemm256_kernel<EPI, 7, true>, LDS_BYTES); e != hipSuccess) return e;
  const int nbm = row_tiles(b), nbn = b.N / TN;
  GemmArgs c = b;   // same tile-order rule as the bf16 launches (b.K counts byte pairs here: N x K x 2 = the weight bytes)
  c.order = tile_order(b, (size_t)b.N * (size_t)b.K * 2 >= ((size_t)64 << 20), nbm);
  hipLaunchKernelGGL((gemm256_kernel<EPI, 7, true>), dim3(nbm * nbn), dim3(512), LDS_BYTES, s, c);
  return hipGetLastError();
}

hipError_t aigv_launch_gemm256_fp8(const GemmArgs& a, int epi, hipStream_t s) {
  if (a.M < 1 || a.N % TN || a.K % 128 || (a.lda % 16) || (a.ldw % 16) || !a.row_scale || !a.col_scale || !a.A || !a.W || !a.C)
    return hipErrorInvalidValue;
  if ((epi == EPI_RESID || epi == EPI_LS_RESID) && (!a.resid || a.ldr % 8)) return hipErrorInvalidValue;
  if (epi == EPI_LS_RESID && !a.ls) return hipErrorInvalidValue;
  if (epi == EPI_SWIGLU && a.bias) return hipErrorInvalidValue;
  GemmArgs b = a;
  b.K = a.K / 2; b.lda = a.lda / 2; b.ldw = a.ldw / 2;
  switch (epi) {
    case EPI_STORE: return launch256_fp8<EPI_STORE>(b, s);
    case EPI_GELU: return launch256_fp8<EPI_GELU>(b, s);
    case EPI_LS_RESID: return launch256_fp8<EPI_LS_RESID>(b, s);
    case EPI_RESID: return launch256_fp8<EPI_RESID>(b, s);
    case EPI_SWIGLU: return launch256_fp8<EPI_SWIGLU>(b, s);
    default: return hipErrorInvalidValue;
  }
}

// split-K slices of the fp8 form (a.K / lda / ldw in e4m3 elements; a.part / a.k_slices filled in; K / 128 divisible by the slices)
hipError_t aigv_launch_gemm256_fp8_partial(const GemmArgs& a, hipStream_t s) {
  if (a.M < 1 || a.N % TN || a.K % 128 || (a.lda % 16) || (a.ldw % 16) || !a.row_scale || !a.col_scale || !a.A || !a.W || !a.part || a.k_slices < 1 ||
      (a.K / 128) % a.k_slices)
    return hipErrorInvalidValue;
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemm256_kernel<EPI_PARTIAL, 7, true>, LDS_BYTES); e != hipSuccess) return e;
  GemmArgs b = a;
  b.K = a.K / 2; b.lda = a.lda / 2; b.ldw = a.ldw / 2;
  const int nbm = row_tiles(a), nbn = a.N / TN;
  hipLaunchKernelGGL((gemm256_kernel<EPI_PARTIAL, 7, true>), dim3(nbm * nbn, a.k_slices), dim3(512), LDS_BYTES, s, b);
  return hipGetLastError();
}

hipError_t aigv_launch_gemm256_partial(const GemmArgs& a, hipStream_t s) {
  if (!aigv_gemm256_supported(a) || a.k_slices < 1 || (a.K / TK) % a.k_slices || !a.part) return hipErrorInvalidValue;
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemm256_kernel<EPI_PARTIAL, 7>, LDS_BYTES); e != hipSuccess) return e;
  const int nbm = row_tiles(a), nbn = a.N / TN;
  hipLaunchKernelGGL((gemm256_kernel<EPI_PARTIAL, 7>), dim3(nbm * nbn, a.k_slices), dim3(512), LDS_BYTES, s, a);
  return hipGetLastError();
}

// body tiles + the K slices of the tail tiles in ONE launch (gemm256_kernel<.., FUSE>): a.row_tab = body halves (a.tab_halves, an even
// count) followed by a.fuse_tail_halves tail halves; a.part / a.k_slices as for aigv_launch_gemm256_partial.  The caller runs the finalize
// pass over the tail table afterwards.
template <int EPI>
static hipError_t launch256_fused(const GemmArgs& a, hipStream_t s) {
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemm256_kernel<EPI, 7, false, true>, LDS_BYTES); e != hipSuccess) return e;
  const int nbm = (a.tab_halves + 1) / 2, nbn = a.N / TN;
  GemmArgs b = a;
  b.order = tile_order(a, (size_t)a.N * (size_t)a.K >= ((size_t)32 << 20), nbm);
  b.fuse_body_wg = nbm * nbn;
  const int grid = nbm * nbn + ((a.fuse_tail_halves + 1) / 2) * nbn * a.k_slices;
  hipLaunchKernelGGL((gemm256_kernel<EPI, 7, false, true>), dim3(grid), dim3(512), LDS_BYTES, s, b);
  return hipGetLastError();
}

hipError_t aigv_launch_gemm256_fused(const GemmArgs& a, int epi, hipStream_t s) {
  if (!aigv_gemm256_supported(a) || !a.row_tab || (a.tab_halves & 1) || a.fuse_tail_halves < 1 || a.k_slices < 2 || (a.K / TK) % a.k_slices || !a.part)
    return hipErrorInvalidValue;
  switch (epi) {
    case EPI_STORE: return launch256_fused<EPI_STORE>(a, s);
    case EPI_GELU: return launch256_fused<EPI_GELU>(a, s);
    case EPI_LS_RESID: return launch256_fused<EPI_LS_RESID>(a, s);
    case EPI_RESID: return launch256_fused<EPI_RESID>(a, s);
    case EPI_SWIGLU: return launch256_fused<EPI_SWIGLU>(a, s);
  }
  return hipErrorInvalidValue;
}

hipError_t aigv_launch_gemm256(const GemmArgs& a, int epi, hipStream_t s) {
  // schedule variants kept for in-process A/B (scripts/gemm_bench.py, profiles/r1_gemm_variants.txt):
  //   1 (default) balanced reads + no s_setprio + LDS-staged epilogue;  3: the same with the direct 8-B epilogue;
  //   0: first schedule (12/4/8/0 reads, s_setprio pairs, direct epilogue)
  switch (a.variant_sel > 0 ? a.variant_sel - 1 : 1) {
    case 0: return launch256v<0>(a, epi, s);
    case 3: return launch256v<3>(a, epi, s);
    default: return launch256v<7>(a, epi, s);
  }
}
