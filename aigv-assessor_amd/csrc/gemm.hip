// bf16 GEMM  C[M,N] = epilogue(A[M,K] · W[N,K]^T)  for gfx950: MFMA 16x16x32 bf16, fp32 accumulate.
//
// This one kernel carries ~95 % of the scorer's FLOPs (ViT qkv/proj/fc1/fc2, projector, LLM
// wqkv/wo/w1|w3/w2).  Both operands are K-contiguous (activations [M,K]; nn.Linear weights [N,K]),
// so A and W tiles are staged the same way.
//
//   tile 128x128x64, 256 threads = 4 waves (2x2), each wave a 64x64 output = 4x4 MFMA tiles
//   LDS  2 buffers x (A 16 KB + W 16 KB) = 64 KB  -> 2 workgroups per CU
//   staging: global_load_lds dwordx4 (16 B/lane, 1 KB per wave-instruction = 8 rows x 128 B).  The
//     LDS image is lane-linear, so the bank-conflict swizzle (16-B chunk ^= row&7) is applied to the
//     per-lane SOURCE address and again on the ds_read_b128 fragment reads (guide rule 21 / T2).
//   pipeline: tile t+1 is in flight (LDS-DMA) while tile t is multiplied; one barrier per K step.
//   MFMA operand order is swapped (W first) so that a lane owns 4 consecutive output COLUMNS of one
//     row: epilogue loads/stores are 8-byte vectors and per-column vectors (bias, layer-scale) are
//     one 8-byte load.
//   blockIdx -> tile: XCD-aware bijective remap (blocks b, b+8 share an XCD/L2), then groups of 8
//     row-tiles sweep the column tiles so co-resident workgroups share A and W panels in L2.
//
// Epilogues reproduce the rounding points of the reference's eager bf16 path:
// every Linear output rounds to bf16 (bias added in fp32 first), activations round, layer-scale
// rounds, residual adds round.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int GROUP_M = 8;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- tile mapping -------------------------------------------------------------------------
  // rows: tile tm covers rows [tm * 128, ..) of the matrix, or - with a half-tile table (GemmArgs::row_tab, the per-sequence row plans
  // of api.hip) - the table's half tm: `valid` rows from its base row
  const int nbm = p.row_tab ? p.tab_halves : (p.M + BM - 1) / BM, nbn = p.N / BN;
  const int nwg = nbm * nbn;
  int wg;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int per_group = GROUP_M * nbn;
  const int group = wg / per_group, first_m = group * GROUP_M;
  const int gsz = min(nbm - first_m, GROUP_M);
  const int in_g = wg - group * per_group;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int n0 = tn * BN;
  int m0 = tm * BM, mend = p.M;
  if (p.row_tab) {
    m0 = __builtin_amdgcn_readfirstlane(p.row_tab[2 * tm]);
    mend = m0 + __builtin_amdgcn_readfirstlane(p.row_tab[2 * tm + 1]);
  }

  // ---- staging addresses ----------------------------------------------------------------------
  // wave-instruction i of this wave fills rows (wave*4+i)*8 .. +7 of the tile; lane -> row lane>>3,
  // 16-B chunk lane&7 of the LDS row; it fetches global chunk (lane&7) ^ (row&7).
  const int lr = lane >> 3;
  const int lc = (lane & 7) ^ lr;
  const bf16_t* srcA[4];
  const bf16_t* srcW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = (wave * 4 + i) * 8 + lr;
    const int ra = min(m0 + rl, mend - 1);  // clamp: rows past the end are computed on a copy, never stored
    srcA[i] = p.A + (size_t)ra * p.lda + lc * 8;
    srcW[i] = p.W + (size_t)(n0 + rl) * p.ldw + lc * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* base = smem + buf * STAGE_BYTES + wave * 4 * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(srcA[i] + k0, base + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(srcW[i] + k0, base + TILE_BYTES + i * 1024);
  };

  // ---- fragment read addresses ------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  // row (within the 128-row tile) of fragment t of this wave: w*64 + t*16 + fr ; (row & 7) == (fr & 7)
  const int sw = fr & 7;
  int offA[2], offW[2];  // byte offset inside a tile for k-half 0/1, fragment 0 (add t*16*128 per fragment)
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int phys = (kh * 4 + fq) ^ sw;
    offA[kh] = (wm * 64 + fr) * 128 + phys * 16;
    offW[kh] = (wn * 64 + fr) * 128 + phys * 16;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // split-K (EPI_PARTIAL): blockIdx.y picks the K slice
  const int nk = (EPI == EPI_PARTIAL) ? p.K / BK / p.k_slices : p.K / BK;
  const int kbase = (EPI == EPI_PARTIAL) ? (int)blockIdx.y * nk * BK : 0;
  stage(0, kbase);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t landed for every wave; everyone is done reading the other buffer
    if (t + 1 < nk) stage((t + 1) & 1, kbase + (t + 1) * BK);
    const char* sA = smem + (t & 1) * STAGE_BYTES;
    const char* sW = sA + TILE_BYTES;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      bf16x8 a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(sA + offA[kh] + i * 16 * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = *(const bf16x8*)(sW + offW[kh] + j * 16 * 128);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns C[m][n .. n+3], m = ..+fr, n = ..+4*fq ---------------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= mend) continue;
    if constexpr (EPI == EPI_PARTIAL) {
      float* slab = p.part + ((size_t)blockIdx.y * p.M + m) * p.N;
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(slab + n0 + wn * 64 + j * 16 + fq * 4) = acc[i][j];
    } else if constexpr (EPI == EPI_SWIGLU) {
      // W rows are interleaved in 16-row blocks: even block = w1 (gate), odd block = w3 (up)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int n = (n0 + wn * 64) / 2 + jp * 16 + fq * 4;
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = rbf(acc[i][2 * jp][e]), u = rbf(acc[i][2 * jp + 1][e]);
          o[e] = f2bf(rbf(silu_f(g)) * u);
        }
        *(u16x4*)(p.C + (size_t)m * p.ldc + n) = o;
      }
    } else {
      size_t orow = (size_t)m;
      const bf16_t* posrow = nullptr;
      if constexpr (EPI == EPI_PATCH) {
        const int f = m / p.np, pi = m - f * p.np;
        orow = (size_t)m + f + 1;                      // skip one class-token row per frame
        posrow = p.pos + (size_t)(pi + 1) * p.N;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fq * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
        if (p.bias) {
          const u16x4 b = *(const u16x4*)(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(b[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);  // the Linear's bf16 output
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
}

template <int EPI>
hipError_t launch(const GemmArgs& a, hipStream_t s) {
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemm_bf16_kernel<EPI>, 2 * STAGE_BYTES); e != hipSuccess) return e;
  if (a.row_tab && (EPI == EPI_PARTIAL || EPI == EPI_PATCH)) return hipErrorInvalidValue;   // the table form has no slab / patch row mapping here
  const int nbm = a.row_tab ? a.tab_halves : (a.M + BM - 1) / BM, nbn = a.N / BN;
  if (nbm <= 0) return hipSuccess;
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(nbm * nbn), dim3(256), 2 * STAGE_BYTES, s, a);
  return hipGetLastError();
}

// ---- split-K finalize: sum the slabs in slice order, then the same epilogues as the GEMM kernels ------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_finalize_kernel(const GemmArgs p, const float* __restrict__ part, int S) {
  const int nq = p.N / 4;                       // float4 groups per row
  // slab rows: the output rows themselves, or (GemmArgs::row_tab, the 256 kernel's half-tile table) compact rows half * 128 + r
  const size_t srows = p.row_tab ? (size_t)((p.tab_halves + 1) / 2) * 256 : (size_t)p.M;
  const size_t total = (p.row_tab ? (size_t)p.tab_halves * 128 : (size_t)p.M) * nq, slab = srows * p.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int sm = (int)(i / nq), n = (int)(i - (size_t)sm * nq) * 4;
    int m = sm;
    if (p.row_tab) {
      const int h = sm >> 7, r = sm & 127;
      if (r >= p.row_tab[2 * h + 1]) continue;
      m = p.row_tab[2 * h] + r;
    }
    f32x4 a = *(const f32x4*)(part + (size_t)sm * p.N + n);
    for (int sidx = 1; sidx < S; ++sidx) {
      const f32x4 b = *(const f32x4*)(part + sidx * slab + (size_t)sm * p.N + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += b[e];
    }
    if constexpr (EPI == EPI_SWIGLU) {
      // 16-row interleave: columns [32b, 32b+16) gate, [32b+16, 32b+32) up -> output column 16b + (n % 16)
      const int blk = n >> 5, within = n & 31;
      if (within >= 16) continue;               // the gate thread also reads its up partner
      f32x4 u = *(const f32x4*)(part + (size_t)sm * p.N + n + 16);
      for (int sidx = 1; sidx < S; ++sidx) {
        const f32x4 b = *(const f32x4*)(part + sidx * slab + (size_t)sm * p.N + n + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] += b[e];
      }
      u16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(rbf(silu_f(rbf(a[e]))) * rbf(u[e]));
      *(u16x4*)(p.C + (size_t)m * p.ldc + blk * 16 + within) = o;
    } else {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a[e];
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
        const u16x4 sc = *(const u16x4*)(p.ls + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e] * bf2f(sc[e]));
      }
      if constexpr (EPI == EPI_LS_RESID || EPI == EPI_RESID) {
        const u16x4 r = *(const u16x4*)(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(bf2f(r[e]) + v[e]);
      }
      u16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
      *(u16x4*)(p.C + (size_t)m * p.ldc + n) = o;
    }
  }
}

template <int EPI>
void launch_finalize(const GemmArgs& a, const float* part, int S, hipStream_t s) {
  const size_t total = (a.row_tab ? (size_t)a.tab_halves * 128 : (size_t)a.M) * (a.N / 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(gemm_finalize_kernel<EPI>, dim3(blocks), dim3(256), 0, s, a, part, S);
}

}  // namespace

const char* aigv_gemm_check(const GemmArgs& a, int epi) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return "gemm: empty problem";
  if (a.N % BN) return "gemm: N must be a multiple of 128";
  if (a.K % BK) return "gemm: K must be a multiple of 64";
  if (a.lda < a.K || a.ldw < a.K || (a.lda % 8) || (a.ldw % 8) || (a.ldc % 4)) return "gemm: bad leading dimension";
  if (!a.A || !a.W || !a.C) return "gemm: null operand";
  if ((epi == EPI_LS_RESID) && (!a.ls || !a.resid)) return "gemm: layer-scale/residual epilogue needs ls and resid";
  if ((epi == EPI_RESID) && !a.resid) return "gemm: residual epilogue needs resid";
  if ((epi == EPI_PATCH) && (!a.pos || a.np <= 0 || a.M % a.np)) return "gemm: patch epilogue needs pos, np | M";
  if (epi == EPI_SWIGLU && a.ldc < a.N / 2) return "gemm: swiglu output is N/2 wide";
  if (epi < 0 || epi >= EPI_COUNT) return "gemm: unknown epilogue";
  return nullptr;
}

// the fp8 form's split-K: scaled fp32 slabs from the e4m3 kernel, then the same fixed-order finalize pass as the bf16 path
hipError_t aigv_launch_gemm_splitk_fp8(const GemmArgs& a, int epi, int k_slices, float* ws, hipStream_t s) {
  if (k_slices < 2 || !ws) return hipErrorInvalidValue;
  GemmArgs b = a;
  b.part = ws;
  b.k_slices = k_slices;
  hipError_t e = aigv_launch_gemm256_fp8_partial(b, s);
  if (e != hipSuccess) return e;
  switch (epi) {
    case EPI_STORE: launch_finalize<EPI_STORE>(a, ws, k_slices, s); break;
    case EPI_GELU: launch_finalize<EPI_GELU>(a, ws, k_slices, s); break;
    case EPI_LS_RESID: launch_finalize<EPI_LS_RESID>(a, ws, k_slices, s); break;
    case EPI_RESID: launch_finalize<EPI_RESID>(a, ws, k_slices, s); break;
    case EPI_SWIGLU: launch_finalize<EPI_SWIGLU>(a, ws, k_slices, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// the second half of a split-K GEMM on its own: sum the k_slices slabs in slice order + epilogue `epi` (a.row_tab / a.tab_halves = the rows
// the slabs hold, as the slice launch addressed them)
hipError_t aigv_launch_gemm_finalize(const GemmArgs& a, int epi, int k_slices, const float* ws, hipStream_t s) {
  if (k_slices < 2 || !ws) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_STORE: launch_finalize<EPI_STORE>(a, ws, k_slices, s); break;
    case EPI_GELU: launch_finalize<EPI_GELU>(a, ws, k_slices, s); break;
    case EPI_LS_RESID: launch_finalize<EPI_LS_RESID>(a, ws, k_slices, s); break;
    case EPI_RESID: launch_finalize<EPI_RESID>(a, ws, k_slices, s); break;
    case EPI_SWIGLU: launch_finalize<EPI_SWIGLU>(a, ws, k_slices, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t aigv_launch_gemm_splitk(const GemmArgs& a, int epi, int k_slices, float* ws, hipStream_t s, bool tile256) {
  if (k_slices < 2 || (a.K / BK) % k_slices || !ws || epi == EPI_PATCH || epi >= EPI_COUNT) return hipErrorInvalidValue;
  if (!tile256 && a.row_tab) return hipErrorInvalidValue;   // the 128 kernel's slabs have no table mapping
  if (tile256) {
    GemmArgs b = a;
    b.part = ws;
    b.k_slices = k_slices;
    hipError_t e = aigv_launch_gemm256_partial(b, s);
    if (e != hipSuccess) return e;
    switch (epi) {
      case EPI_STORE: launch_finalize<EPI_STORE>(a, ws, k_slices, s); break;
      case EPI_GELU: launch_finalize<EPI_GELU>(a, ws, k_slices, s); break;
      case EPI_LS_RESID: launch_finalize<EPI_LS_RESID>(a, ws, k_slices, s); break;
      case EPI_RESID: launch_finalize<EPI_RESID>(a, ws, k_slices, s); break;
      case EPI_SWIGLU: launch_finalize<EPI_SWIGLU>(a, ws, k_slices, s); break;
    }
    return hipGetLastError();
  }
  static LdsAttrOnce lds_attr;
  if (hipError_t e = lds_attr.ensure((const void*)gemm_bf16_kernel<EPI_PARTIAL>, 2 * STAGE_BYTES); e != hipSuccess) return e;
  GemmArgs b = a;
  b.part = ws;
  b.k_slices = k_slices;
  const int nbm = (a.M + BM - 1) / BM, nbn = a.N / BN;
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI_PARTIAL>, dim3(nbm * nbn, k_slices), dim3(256), 2 * STAGE_BYTES, s, b);
  switch (epi) {
    case EPI_STORE: launch_finalize<EPI_STORE>(a, ws, k_slices, s); break;
    case EPI_GELU: launch_finalize<EPI_GELU>(a, ws, k_slices, s); break;
    case EPI_LS_RESID: launch_finalize<EPI_LS_RESID>(a, ws, k_slices, s); break;
    case EPI_RESID: launch_finalize<EPI_RESID>(a, ws, k_slices, s); break;
    case EPI_SWIGLU: launch_finalize<EPI_SWIGLU>(a, ws, k_slices, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t aigv_launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch<EPI_STORE>(a, s);
    case EPI_GELU: return launch<EPI_GELU>(a, s);
    case EPI_LS_RESID: return launch<EPI_LS_RESID>(a, s);
    case EPI_RESID: return launch<EPI_RESID>(a, s);
    case EPI_SWIGLU: return launch<EPI_SWIGLU>(a, s);
    case EPI_PATCH: return launch<EPI_PATCH>(a, s);
  }
  return hipErrorInvalidValue;
}
