// SlowFast-R50 motion branch for gfx950: frames -> [B, 2304] motion feature.
//
// Replaces internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:97-133 (pack_pathway_output) and :135-193 (class slowfast:
// blocks 0..4 of pytorchvideo's slowfast_r50, repeat_interleave(4), AvgPool3d((8|32,7,7), stride 1), AdaptiveAvgPool3d(1), concat).
// pytorchvideo is not vendored by the reference: the architecture is the published R50 8x8 one (see oracle/slowfast.py for the
// restatement this is tested against; parity with the real package is UNPINNED in this container).
//
// Design: activations are channels-last [B, T, H, W, C] bf16 with a row stride (ld) so that the fast->slow fusion writes
// straight into the tail channels of the slow pathway's buffer (no concat copy).  Every convolution is ONE implicit-GEMM kernel
// on v_mfma_f32_16x16x32_bf16: rows = output positions, K = taps x Cin gathered on the fly (a tap is a contiguous Cin vector in
// this layout), eval-mode BatchNorm folded into the weights / an fp32 bias on the host, residual add + ReLU in the epilogue.
// The branch is ~0.05 TFLOP per clip (0.15 % of the scorer): the kernel is sized for simplicity, not for the MFMA roofline.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/aigv_amd.h"
#include "common.h"
#include "kernels.h"

namespace {

struct ConvArgs {
  const bf16_t* in;      // [B, Ti, Hi, Wi, ld_in], channels [0, Cin) used
  const bf16_t* w;       // [CoutPad16, Kp]: k = ((dt * kh + dy) * kw + dx) * Cin + ci, zero padded to Kp (multiple of 64)
  const float* bias;     // [Cout] (folded BatchNorm shift)
  const bf16_t* res;     // optional residual [rows, ld_res]
  bf16_t* out;           // [rows, ld_out], written at channel offset c_off
  int ld_in, Cin, Ti, Hi, Wi;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int To, Ho, Wo, Cout, CoutPad, Kp;
  int ld_res, ld_out, c_off, relu;
  long rows;             // B * To * Ho * Wo
  float* part;           // split-K: fp32 slabs [k_slices][rows][Cout] (k_slices > 1), finished by conv_finalize_kernel
  int k_slices;          // gridDim.z; Kp / 64 divisible by it
};

// Tile: 128 output positions x BN output channels per workgroup (4 waves x 32 positions), K in steps of 64 (two MFMA K-blocks per
// barrier pair).  LDS rows are 128 B, the 16-byte chunk c of row r sits in slot c ^ (r & 7) so that the fragment reads (16 rows x one
// chunk per quarter wave) spread over all banks.  The MFMA takes the WEIGHT fragment as its first operand, so the accumulator holds
// 4 consecutive channels of one position per lane -> 8-byte stores.
template <int BN>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(ConvArgs p) {
  constexpr int BM = 128, BK = 64, NT = BN / 16, WJ = (BN + 31) / 32;
  __shared__ __attribute__((aligned(16))) bf16_t sA[BM * BK];
  __shared__ __attribute__((aligned(16))) bf16_t sW[BN * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  // split-K: the deep layers of the slow pathway have ~100 tiles and K up to 6144 - slice z takes K range [kb, ke)
  const int kspan = p.Kp / p.k_slices, kb = blockIdx.z * kspan;
  const int kc = tid & 7;    // which 8-wide K chunk of the 64 this thread gathers
  const int r0 = tid >> 3;   // rows r0 + 32 j
  long base[4];
  int t0[4], y0[4], x0[4];
  bool rv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long m = m0 + r0 + 32 * j;
    rv[j] = m < p.rows;
    const long mm = rv[j] ? m : 0;
    const int x = (int)(mm % p.Wo);
    const long r1 = mm / p.Wo;
    const int y = (int)(r1 % p.Ho);
    const long r2 = r1 / p.Ho;
    const int t = (int)(r2 % p.To);
    const long b = r2 / p.To;
    t0[j] = t * p.st - p.pt;
    y0[j] = y * p.sh - p.ph;
    x0[j] = x * p.sw - p.pw;
    base[j] = b * p.Ti * p.Hi * p.Wi;
  }
  // this thread's position inside the tap grid, advanced by 64 K-elements per step
  int c = kb + kc * 8, dx = 0, dy = 0, dt = 0;
  auto normalise = [&]() {
    while (c >= p.Cin) {
      c -= p.Cin;
      if (++dx == p.kw) {
        dx = 0;
        if (++dy == p.kh) { dy = 0; ++dt; }
      }
    }
  };
  normalise();
  auto gather = [&](int j) -> u16x8 {
    u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!rv[j] || dt >= p.kt) return z;
    const int tt = t0[j] + dt, yy = y0[j] + dy, xx = x0[j] + dx;
    if ((unsigned)tt >= (unsigned)p.Ti || (unsigned)yy >= (unsigned)p.Hi || (unsigned)xx >= (unsigned)p.Wi) return z;
    return *(const u16x8*)(p.in + (base[j] + ((long)tt * p.Hi + yy) * p.Wi + xx) * p.ld_in + c);
  };
  bool wv[WJ];
  const bf16_t* wsrc[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int wrow = r0 + 32 * j;
    wv[j] = wrow < BN && (n0 + wrow) < p.CoutPad;
    wsrc[j] = p.w + (size_t)(n0 + (wv[j] ? wrow : 0)) * p.Kp + kb + kc * 8;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  u16x8 ra[4], rw[WJ];
#pragma unroll
  for (int j = 0; j < 4; ++j) ra[j] = gather(j);
#pragma unroll
  for (int j = 0; j < WJ; ++j) rw[j] = wv[j] ? *(const u16x8*)wsrc[j] : zero;
  const int slot = (kc ^ (r0 & 7)) * 8;   // (r0 + 32 j) & 7 == r0 & 7
  for (int k0 = 0; k0 < kspan; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) *(u16x8*)&sA[(r0 + 32 * j) * BK + slot] = ra[j];
#pragma unroll
    for (int j = 0; j < WJ; ++j)
      if (r0 + 32 * j < BN) *(u16x8*)&sW[(r0 + 32 * j) * BK + slot] = rw[j];
    __syncthreads();
    if (k0 + BK < kspan) {   // prefetch the next step while the MFMAs run
      c += BK;
      normalise();
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[j] = gather(j);
#pragma unroll
      for (int j = 0; j < WJ; ++j) rw[j] = wv[j] ? *(const u16x8*)(wsrc[j] + k0 + BK) : zero;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rs = ((h * 4 + (lane >> 4)) ^ (lane & 7)) * 8;   // fragment rows are (16 i + lane & 15): & 7 == lane & 7
      bf16x8 a[2], w[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)&sA[(wave * 32 + i * 16 + (lane & 15)) * BK + rs];
#pragma unroll
      for (int n = 0; n < NT; ++n) w[n] = *(const bf16x8*)&sW[(n * 16 + (lane & 15)) * BK + rs];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[n], a[i], acc[i][n], 0, 0, 0);
    }
  }
  // epilogue: acc[i][n][e] = channel n0 + 16 n + 4 (lane >> 4) + e of position m0 + 32 wave + 16 i + (lane & 15)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long m = m0 + wave * 32 + i * 16 + (lane & 15);
    if (m >= p.rows) continue;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int ch = n0 + n * 16 + (lane >> 4) * 4;
      if (ch >= p.Cout) continue;   // Cout is a multiple of 4
      if (p.k_slices > 1) {
        *(f32x4*)(p.part + ((size_t)blockIdx.z * p.rows + m) * p.Cout + ch) = acc[i][n];
        continue;
      }
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][n][e] + p.bias[ch + e];
      if (p.res) {
        const u16x4 r = *(const u16x4*)(p.res + m * p.ld_res + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf2f(r[e]);
      }
      u16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(p.relu ? fmaxf(v[e], 0.f) : v[e]);
      *(u16x4*)(p.out + m * p.ld_out + p.c_off + ch) = o;
    }
  }
}

// split-K tail: slabs summed in slice order, then the same bias / residual / ReLU epilogue (deterministic)
__global__ __launch_bounds__(256) void conv_finalize_kernel(ConvArgs p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over rows * Cout / 4
  const int cq = p.Cout >> 2;
  if (i >= p.rows * cq) return;
  const long m = i / cq;
  const int ch = (int)(i - m * cq) * 4;
  f32x4 v = *(const f32x4*)(p.part + (size_t)m * p.Cout + ch);
  for (int z = 1; z < p.k_slices; ++z) v += *(const f32x4*)(p.part + ((size_t)z * p.rows + m) * p.Cout + ch);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += p.bias[ch + e];
  if (p.res) {
    const u16x4 r = *(const u16x4*)(p.res + m * p.ld_res + ch);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bf2f(r[e]);
  }
  u16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf(p.relu ? fmaxf(v[e], 0.f) : v[e]);
  *(u16x4*)(p.out + m * p.ld_out + p.c_off + ch) = o;
}

// frames [B*T, 3, H, W] bf16 NCHW (the tensor the ViT also reads) -> channels-last with 4 channels per pixel (3 real + 1 zero) for the
// fast pathway (all T frames) and the slow pathway (frames slow_idx[0..Ts)), modeling_internvl_chat.py:97-133.  The stems read this
// buffer as [.., W/2, 8]: two neighbouring pixels form one 8-channel vector (see Builder::conv, stem_pairs).
struct FrameIdx { int v[64]; };
__global__ __launch_bounds__(256) void sf_repack_kernel(const bf16_t* __restrict__ frames, int T, int Ts, long hw, bf16_t* __restrict__ fast,
                                                        bf16_t* __restrict__ slow, FrameIdx idx, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over B * (T + Ts) * hw output pixels
  if (i >= total) return;
  const long px = i % hw;
  const long f = i / hw;
  const int tt = (int)(f % (T + Ts));
  const long b = f / (T + Ts);
  const int src_t = tt < T ? tt : idx.v[tt - T];
  const bf16_t* s = frames + ((b * T + src_t) * 3) * hw + px;
  u16x4 o = {s[0], s[hw], s[2 * hw], 0};
  bf16_t* d = tt < T ? fast + ((b * T + tt) * hw + px) * 4 : slow + ((b * Ts + (tt - T)) * hw + px) * 4;
  *(u16x4*)d = o;
}

// MaxPool3d([1,3,3], stride [1,2,2], pad [0,1,1]) on channels-last data, 8 channels per thread
__global__ __launch_bounds__(256) void sf_maxpool_kernel(const bf16_t* __restrict__ in, int C, int Hi, int Wi, int Ho, int Wo,
                                                         bf16_t* __restrict__ out, int ld_out, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over frames * Ho * Wo * (C / 8)
  if (i >= total) return;
  const int cv = C >> 3;
  const int c8 = (int)(i % cv);
  const long px = i / cv;
  const int x = (int)(px % Wo);
  const long r = px / Wo;
  const int y = (int)(r % Ho);
  const long f = r / Ho;
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = 2 * y - 1 + dy;
    if ((unsigned)yy >= (unsigned)Hi) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = 2 * x - 1 + dx;
      if ((unsigned)xx >= (unsigned)Wi) continue;
      const u16x8 v = *(const u16x8*)(in + ((f * Hi + yy) * Wi + xx) * C + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], bf2f(v[e]));
    }
  }
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(m[e]);
  *(u16x8*)(out + px * ld_out + c8 * 8) = o;
}

// repeat_interleave(4) + AvgPool3d((k,7,7), stride 1) + AdaptiveAvgPool3d(1) == one weighted mean with separable weights
// (how many pooling windows cover each position), modeling_internvl_chat.py:183-189.  A workgroup owns 32 channels of one clip:
// thread = (8-channel chunk, one of 64 position slices), slices reduced through LDS.
struct PoolW { float t[32], y[32], x[32]; };
__global__ __launch_bounds__(256) void sf_pool_kernel(const bf16_t* __restrict__ in, int T, int H, int W, int C, PoolW w,
                                                      bf16_t* __restrict__ out, int ld_out, int c_off) {
  __shared__ float red[64][4][8];
  const int chunk = threadIdx.x & 3, slice = threadIdx.x >> 2;
  const int c8 = blockIdx.x * 4 + chunk;
  const int b = blockIdx.y;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c8 * 8 < C) {
    const bf16_t* s = in + (long)b * T * H * W * C + c8 * 8;
    const int n = T * H * W;
    for (int i = slice; i < n; i += 64) {
      const int x = i % W, r = i / W, y = r % H, t = r / H;
      const float wt = w.t[t] * w.y[y] * w.x[x];
      const u16x8 v = *(const u16x8*)(s + (long)i * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wt * bf2f(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[slice][chunk][e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 32 && (blockIdx.x * 4 + (threadIdx.x >> 3)) * 8 < C) {   // one thread per output channel of this workgroup
    const int ch = threadIdx.x >> 3, e = threadIdx.x & 7;
    float v = 0.f;
    for (int q = 0; q < 64; ++q) v += red[q][ch][e];
    out[(long)b * ld_out + c_off + (blockIdx.x * 4 + ch) * 8 + e] = f2bf(v);
  }
}

constexpr long SPLITK_MAX_ROWS = 8192;   // per launch; only the deep, narrow-M layers are split

// K slices for a conv with `tiles` output tiles: enough workgroups to hide the gather latency, slices of >= 4 K-steps
int conv_k_slices(long rows, int CoutPad, int Kp) {   // rows of ONE clip
  if (rows > SPLITK_MAX_ROWS || Kp < 1024) return 1;
  const int bn = CoutPad >= 64 ? 64 : CoutPad >= 32 ? 32 : 16;
  const long tiles = ((rows + 127) / 128) * ((CoutPad + bn - 1) / bn);
  if (tiles >= 64) return 1;
  const int steps = Kp / 64;
  for (int S : {8, 6, 4, 3, 2})
    if (steps % S == 0 && steps / S >= 4 && tiles * S <= 512) return S;
  return 1;
}

// S: K slices decided by the caller (from the PER-CLIP shape, so that a clip's bits do not depend on its batch mates); 1 without workspace
hipError_t launch_conv(const ConvArgs& a_in, hipStream_t s, int S = 1, float* ws = nullptr, size_t ws_floats = 0) {
  ConvArgs a = a_in;
  if (a.rows <= 0) return hipSuccess;
  if (a.Cin % 8 || a.ld_in % 8 || a.Cout % 4 || a.ld_out % 4 || a.c_off % 4 || a.Kp % 64 || a.CoutPad % 16 || (a.res && a.ld_res % 4) ||
      a.Kp < a.kt * a.kh * a.kw * a.Cin)
    return hipErrorInvalidValue;
  if (!ws || S < 1 || (a.Kp / 64) % S || (size_t)S * a.rows * a.Cout > ws_floats) S = 1;
  a.k_slices = S;
  a.part = S > 1 ? ws : nullptr;
  const unsigned gx = (unsigned)((a.rows + 127) / 128);
  if (a.CoutPad >= 64) hipLaunchKernelGGL(conv3d_mfma_kernel<64>, dim3(gx, (a.CoutPad + 63) / 64, S), dim3(256), 0, s, a);
  else if (a.CoutPad >= 32) hipLaunchKernelGGL(conv3d_mfma_kernel<32>, dim3(gx, (a.CoutPad + 31) / 32, S), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv3d_mfma_kernel<16>, dim3(gx, 1, S), dim3(256), 0, s, a);
  if (S > 1) {
    const long total = a.rows * (a.Cout / 4);
    hipLaunchKernelGGL(conv_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

inline uint16_t bf16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_float_host(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct HostTensor {
  std::vector<float> v;
  std::vector<int64_t> shape;
};

enum OpKind { OP_REPACK, OP_CONV, OP_MAXPOOL, OP_POOL };
struct Op {
  OpKind kind;
  ConvArgs a{};          // OP_CONV: pointers are filled at run time from the buffer ids below
  int in_buf = -1, out_buf = -1, res_buf = -1;
  size_t w_off = 0, b_off = 0;
  long rows_per_clip = 0;
  int k_slices = 1;
  // OP_MAXPOOL / OP_POOL
  int C = 0, T = 0, Hi = 0, Wi = 0, Ho = 0, Wo = 0, ld_out = 0, c_off = 0;
  PoolW pw{};
};

int sf_fail(int code, const char* fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  aigv_set_error(buf);
  return code;
}

}  // namespace

struct aigv_slowfast {
  int device = 0, Bcap = 0, T = 0, Ts = 0, H = 0, W = 0;
  bool finalized = false;
  std::map<std::string, HostTensor> host;
  std::vector<Op> ops;
  std::vector<size_t> buf_elems;   // per clip
  std::vector<bf16_t*> bufs;
  bf16_t* d_w = nullptr;
  float* d_b = nullptr;
  float* d_part = nullptr;   // split-K slabs
  size_t part_floats = 0;
  FrameIdx slow_idx{};
  double flops_per_clip = 0;
};

namespace {

enum { B_IN0 = 0, B_IN1, B_STEM0, B_STEM1, B_X00, B_X01, B_X10, B_X11, B_A0, B_A1, B_B0, B_B1, B_S0, B_S1, B_COUNT };

struct Builder {
  aigv_slowfast* sf;
  std::vector<uint16_t> w;   // packed bf16 weights, all convs
  std::vector<float> b;      // folded biases
  std::string missing;

  const HostTensor* get(const std::string& name) {
    auto it = sf->host.find(name);
    if (it == sf->host.end()) {
      if (missing.size() < 400) missing += (missing.empty() ? "" : ", ") + name;
      return nullptr;
    }
    return &it->second;
  }
  void need(int buf, size_t elems) {
    if (sf->buf_elems[buf] < elems) sf->buf_elems[buf] = elems;
  }
  // fold conv (no bias) + eval BatchNorm (eps 1e-5) -> packed [CoutPad16, Kp] bf16 + fp32 bias; returns false when a tensor is missing or mis-shaped
  bool conv(const std::string& conv_name, const std::string& norm_name, int in_buf, int ld_in, int cin_eff, int Ti, int Hi, int Wi, int kt, int kh,
            int kw, int st, int sh, int sw, int pt, int ph, int pw, int cout, int out_buf, int ld_out, int c_off, int res_buf, int ld_res, bool relu,
            int* To_, int* Ho_, int* Wo_, bool stem_pairs = false) {
    const HostTensor* W = get(conv_name + ".weight");
    const HostTensor* g = get(norm_name + ".weight");
    const HostTensor* be = get(norm_name + ".bias");
    const HostTensor* mu = get(norm_name + ".running_mean");
    const HostTensor* var = get(norm_name + ".running_var");
    const int To = (Ti + 2 * pt - kt) / st + 1, Ho = (Hi + 2 * ph - kh) / sh + 1, Wo = (Wi + 2 * pw - kw) / sw + 1;
    *To_ = To; *Ho_ = Ho; *Wo_ = Wo;
    if (!W || !g || !be || !mu || !var) return false;
    const int cin_real = W->shape.size() == 5 ? (int)W->shape[1] : -1;
    if (W->shape.size() != 5 || W->shape[0] != cout || (stem_pairs ? cin_real != 3 : cin_real != cin_eff) || W->shape[2] != kt || W->shape[3] != kh || W->shape[4] != kw ||
        (int)g->v.size() != cout || (int)be->v.size() != cout || (int)mu->v.size() != cout || (int)var->v.size() != cout) {
      if (missing.size() < 400) missing += (missing.empty() ? "" : ", ") + conv_name + " (bad shape)";
      return false;
    }
    const int taps = kt * kh * kw;
    // stem_pairs: a 3-channel input with a [kt,7,7] kernel, stride 2 and pad 3 along x.  The input buffer holds 4 channels per pixel,
    // read as 8-channel PAIRS of pixels: output x needs pixels 2x-3 .. 2x+3 = pairs x-2 .. x+1, i.e. a 4-tap stride-1 pad-2 conv over
    // pairs whose weight for (pair tap q, parity s, channel ci) is the original tap dx = 2q + s - 1 (zero for dx = -1) - K shrinks from
    // 49 x 8 to 28 x 8 per frame tap against padding the 3 channels to 8, with no change to the kernel.
    const int kw_k = stem_pairs ? 4 : kw, taps_k = kt * kh * kw_k;
    const int K = taps_k * cin_eff, Kp = (K + 63) / 64 * 64, coutPad = (cout + 15) / 16 * 16;
    Op op;
    op.kind = OP_CONV;
    op.w_off = w.size();
    op.b_off = b.size();
    w.resize(w.size() + (size_t)coutPad * Kp, 0);
    b.resize(b.size() + cout, 0.f);
    for (int co = 0; co < cout; ++co) {
      const float scale = g->v[co] / std::sqrt(var->v[co] + 1e-5f);
      b[op.b_off + co] = be->v[co] - mu->v[co] * scale;
      uint16_t* dst = &w[op.w_off + (size_t)co * Kp];
      for (int ci = 0; ci < cin_real; ++ci)
        for (int tap = 0; tap < taps; ++tap) {
          int k = tap * cin_eff + ci;
          if (stem_pairs) {
            const int q = tap % kw + 1;   // dx + 1 = 2 * pair tap + parity
            k = ((tap / kw) * 4 + (q >> 1)) * 8 + (q & 1) * 4 + ci;
          }
          dst[k] = bf16_host(W->v[((size_t)co * cin_real + ci) * taps + tap] * scale);
        }
    }
    ConvArgs& a = op.a;
    a.ld_in = ld_in; a.Cin = cin_eff; a.Ti = Ti; a.Hi = Hi; a.Wi = stem_pairs ? Wi / 2 : Wi;
    a.kt = kt; a.kh = kh; a.kw = kw_k; a.st = st; a.sh = sh; a.sw = stem_pairs ? 1 : sw; a.pt = pt; a.ph = ph; a.pw = stem_pairs ? 2 : pw;
    a.To = To; a.Ho = Ho; a.Wo = Wo; a.Cout = cout; a.CoutPad = coutPad; a.Kp = Kp;
    a.ld_res = ld_res; a.ld_out = ld_out; a.c_off = c_off; a.relu = relu ? 1 : 0;
    op.in_buf = in_buf; op.out_buf = out_buf; op.res_buf = res_buf;
    op.rows_per_clip = (long)To * Ho * Wo;
    need(out_buf, (size_t)op.rows_per_clip * ld_out);
    sf->flops_per_clip += 2.0 * op.rows_per_clip * cout * (double)taps * cin_real;
    sf->ops.push_back(op);
    return true;
  }
};

void cover_weights(float* w, int n, int k) {   // how many stride-1 windows of length k cover each of n positions
  for (int i = 0; i < n; ++i) {
    const int lo = i - k + 1 > 0 ? i - k + 1 : 0, hi = i < n - k ? i : n - k;
    w[i] = (float)(hi - lo + 1);
  }
}

}  // namespace

extern "C" {

int aigv_slowfast_create(int device, int max_clips, int frames_per_clip, int height, int width, aigv_slowfast** out) {
  if (!out) return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_create: null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return sf_fail(AIGV_ERR_HIP, "aigv_slowfast_create: no HIP device is visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_create: device %d out of range (%d visible)", device, ndev);
  // the head pools need T/4*4 >= 8 repeated slow frames and a final 7x7 map: T a multiple of 4 in [8, 32]; H, W multiples of 32, >= 224
  if (max_clips <= 0 || frames_per_clip < 8 || frames_per_clip > 32 || frames_per_clip % 4 || height < 224 || width < 224 || height % 32 ||
      width % 32 || height > 1024 || width > 1024)
    return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_create: needs 8 <= T <= 32 with T %% 4 == 0 and 224 <= H, W <= 1024 multiples of 32 (got T=%d %dx%d)",
                   frames_per_clip, height, width);
  aigv_slowfast* sf = new (std::nothrow) aigv_slowfast();
  if (!sf) return sf_fail(AIGV_ERR_ALLOC, "out of host memory");
  sf->device = device; sf->Bcap = max_clips; sf->T = frames_per_clip; sf->Ts = frames_per_clip / 4; sf->H = height; sf->W = width;
  // torch.linspace(0, T - 1, T // 4).long() (modeling_internvl_chat.py:109-111): fp32 start + i * step, truncated; the second half is
  // computed from the end (end - (steps - 1 - i) * step) exactly as torch's kernel does
  const int n = sf->Ts;
  const float step = n > 1 ? (float)(sf->T - 1) / (float)(n - 1) : 0.f;
  for (int i = 0; i < n; ++i) sf->slow_idx.v[i] = i < n / 2 ? (int)(0.f + step * i) : (int)((float)(sf->T - 1) - step * (float)(n - 1 - i));
  *out = sf;
  return 0;
}

void aigv_slowfast_destroy(aigv_slowfast* sf) {
  if (!sf) return;
  hipSetDevice(sf->device);
  hipDeviceSynchronize();
  for (bf16_t* p : sf->bufs) hipFree(p);
  hipFree(sf->d_w);
  hipFree(sf->d_b);
  hipFree(sf->d_part);
  delete sf;
}

int aigv_slowfast_load_weight(aigv_slowfast* sf, const char* name, const void* host_data, const int64_t* shape, int ndim, int dtype) {
  if (!sf || !name || !host_data || !shape || ndim <= 0 || ndim > 5) return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_load_weight: bad argument");
  if (sf->finalized) return sf_fail(AIGV_ERR_STATE, "aigv_slowfast_load_weight(%s): already finalized", name);
  std::string key(name);
  for (const char* pre : {"slowfast_model.feature_extraction.", "feature_extraction.", "blocks."}) {
    const size_t n = strlen(pre);
    if (key.compare(0, n, pre) == 0) { key = key.substr(n); break; }
  }
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_load_weight(%s): bad shape", name);
    n *= (size_t)shape[i];
    t.shape.push_back(shape[i]);
  }
  t.v.resize(n);
  if (dtype == AIGV_F32) memcpy(t.v.data(), host_data, n * 4);
  else if (dtype == AIGV_BF16) for (size_t i = 0; i < n; ++i) t.v[i] = bf16_to_float_host(((const uint16_t*)host_data)[i]);
  else return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_load_weight(%s): dtype %d", name, dtype);
  sf->host[key] = std::move(t);
  return 0;
}

int aigv_slowfast_finalize(aigv_slowfast* sf) {
  if (!sf) return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_finalize: null handle");
  if (sf->finalized) return 0;
  if (hipSetDevice(sf->device) != hipSuccess) return sf_fail(AIGV_ERR_HIP, "hipSetDevice failed");
  sf->ops.clear();
  sf->buf_elems.assign(B_COUNT, 0);
  sf->flops_per_clip = 0;
  Builder bd{sf, {}, {}, {}};
  const int T[2] = {sf->Ts, sf->T};
  const int IN[2] = {B_IN0, B_IN1}, STEM[2] = {B_STEM0, B_STEM1}, X[2][2] = {{B_X00, B_X01}, {B_X10, B_X11}}, A[2] = {B_A0, B_A1},
            Bq[2] = {B_B0, B_B1}, S[2] = {B_S0, B_S1};
  {
    Op op; op.kind = OP_REPACK;
    sf->ops.push_back(op);
    bd.need(B_IN0, (size_t)sf->Ts * sf->H * sf->W * 4);
    bd.need(B_IN1, (size_t)sf->T * sf->H * sf->W * 4);
  }
  bool ok = true;
  int h = 0, w = 0, to = 0;
  int ld[2] = {64 + 16, 8};   // row stride of the current block input per pathway
  // block 0: stems + max-pool, then the first fusion
  for (int p = 0; p < 2; ++p) {
    const int kt = p ? 5 : 1, cout = p ? 8 : 64;
    const std::string pre = "0.multipathway_blocks." + std::to_string(p);
    ok &= bd.conv(pre + ".conv", pre + ".norm", IN[p], 8, 8, T[p], sf->H, sf->W, kt, 7, 7, 1, 2, 2, kt / 2, 3, 3, cout, STEM[p], cout, 0, -1, 0, true, &to, &h, &w, true);
    Op mp; mp.kind = OP_MAXPOOL;
    mp.in_buf = STEM[p]; mp.out_buf = X[p][0]; mp.C = cout; mp.T = T[p]; mp.Hi = h; mp.Wi = w;
    mp.Ho = (h + 2 - 3) / 2 + 1; mp.Wo = (w + 2 - 3) / 2 + 1; mp.ld_out = ld[p];
    bd.need(X[p][0], (size_t)T[p] * mp.Ho * mp.Wo * ld[p]);
    sf->ops.push_back(mp);
    if (p == 1) { h = mp.Ho; w = mp.Wo; }
  }
  int sp_h = h, sp_w = w;   // spatial size entering res2
  ok &= bd.conv("0.multipathway_fusion.conv_fast_to_slow", "0.multipathway_fusion.norm", X[1][0], 8, 8, T[1], sp_h, sp_w, 7, 1, 1, 4, 1, 1, 3, 0, 0, 16,
                X[0][0], ld[0], 64, -1, 0, true, &to, &h, &w);
  if (to != T[0]) return sf_fail(AIGV_ERR_ARG, "fast->slow fusion yields %d frames, slow pathway has %d", to, T[0]);
  int cur[2] = {0, 0};
  int cin[2] = {80, 8};
  static const int depths[4] = {3, 4, 6, 3};
  for (int s = 0; s < 4; ++s) {
    int out_h = sp_h, out_w = sp_w;
    for (int p = 0; p < 2; ++p) {
      const int inner = (p ? 8 : 64) << s, cout = inner * 4;
      const int fuse_ch = (s < 3 && p == 0) ? 2 * (32 << s) : 0;
      const int ld_out = cout + fuse_ch;
      const int kt_a = (p == 1 || s >= 2) ? 3 : 1;
      int bh = sp_h, bw = sp_w;
      for (int blk = 0; blk < depths[s]; ++blk) {
        const int stv = (blk == 0 && s > 0) ? 2 : 1;
        const std::string pre = std::to_string(s + 1) + ".multipathway_blocks." + std::to_string(p) + ".res_blocks." + std::to_string(blk) + ".";
        const int xin = X[p][cur[p]], xout = X[p][cur[p] ^ 1];
        const int c_in = blk == 0 ? cin[p] : cout, ld_in = blk == 0 ? ld[p] : ld_out;
        int t1, h1, w1, h2, w2;
        int res_buf = xin, res_ld = ld_in;
        if (blk == 0) {
          ok &= bd.conv(pre + "branch1_conv", pre + "branch1_norm", xin, ld_in, c_in, T[p], bh, bw, 1, 1, 1, 1, stv, stv, 0, 0, 0, cout, S[p], cout, 0, -1, 0,
                        false, &t1, &h1, &w1);
          res_buf = S[p]; res_ld = cout;
        }
        ok &= bd.conv(pre + "branch2.conv_a", pre + "branch2.norm_a", xin, ld_in, c_in, T[p], bh, bw, kt_a, 1, 1, 1, 1, 1, kt_a / 2, 0, 0, inner, A[p], inner, 0,
                      -1, 0, true, &t1, &h1, &w1);
        ok &= bd.conv(pre + "branch2.conv_b", pre + "branch2.norm_b", A[p], inner, inner, T[p], bh, bw, 1, 3, 3, 1, stv, stv, 0, 1, 1, inner, Bq[p], inner, 0, -1,
                      0, true, &t1, &h2, &w2);
        ok &= bd.conv(pre + "branch2.conv_c", pre + "branch2.norm_c", Bq[p], inner, inner, T[p], h2, w2, 1, 1, 1, 1, 1, 1, 0, 0, 0, cout, xout, ld_out, 0, res_buf,
                      res_ld, true, &t1, &h1, &w1);
        bh = h2; bw = w2;
        cur[p] ^= 1;
      }
      cin[p] = ld_out; ld[p] = ld_out;
      out_h = bh; out_w = bw;
    }
    sp_h = out_h; sp_w = out_w;
    if (s < 3) {
      const int cf = 32 << s, cs = 256 << s;
      const std::string pre = std::to_string(s + 1) + ".multipathway_fusion";
      ok &= bd.conv(pre + ".conv_fast_to_slow", pre + ".norm", X[1][cur[1]], cf, cf, T[1], sp_h, sp_w, 7, 1, 1, 4, 1, 1, 3, 0, 0, 2 * cf, X[0][cur[0]], ld[0], cs,
                    -1, 0, true, &to, &h, &w);
    }
  }
  if (!ok) return sf_fail(AIGV_ERR_STATE, "aigv_slowfast_finalize: missing or mis-shaped tensors: %s%s", bd.missing.c_str(), bd.missing.size() >= 400 ? " ..." : "");
  if (sp_h < 7 || sp_w < 7 || sp_h > 32 || sp_w > 32) return sf_fail(AIGV_ERR_ARG, "final map %dx%d outside the head pool's range", sp_h, sp_w);
  for (int p = 0; p < 2; ++p) {   // head pools: temporal window 8 (slow) / 32 (fast) over the 4x repeated frames, 7x7 spatial, stride 1
    Op op; op.kind = OP_POOL;
    op.in_buf = X[p][cur[p]]; op.C = p ? 256 : 2048; op.T = T[p]; op.Hi = sp_h; op.Wi = sp_w; op.ld_out = 2304; op.c_off = p ? 2048 : 0;
    const int L = 4 * T[p], k = p ? 32 : 8;
    if (L < k) return sf_fail(AIGV_ERR_ARG, "too few frames for the head pool");
    float rep[128];
    cover_weights(rep, L, k);
    float ty[32], tx[32];
    cover_weights(ty, sp_h, 7);
    cover_weights(tx, sp_w, 7);
    const float nt = (float)(L - k + 1) * k, ny = (float)(sp_h - 6) * 7, nx = (float)(sp_w - 6) * 7;
    for (int t = 0; t < T[p]; ++t) op.pw.t[t] = (rep[4 * t] + rep[4 * t + 1] + rep[4 * t + 2] + rep[4 * t + 3]) / nt;
    for (int y = 0; y < sp_h; ++y) op.pw.y[y] = ty[y] / ny;
    for (int x = 0; x < sp_w; ++x) op.pw.x[x] = tx[x] / nx;
    sf->ops.push_back(op);
  }
  // device memory
  auto dmalloc = [&](void** p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 16); };
  if (dmalloc((void**)&sf->d_w, bd.w.size() * 2) != hipSuccess || dmalloc((void**)&sf->d_b, bd.b.size() * 4) != hipSuccess)
    return sf_fail(AIGV_ERR_ALLOC, "aigv_slowfast_finalize: hipMalloc of %zu weight bytes failed", bd.w.size() * 2);
  if (hipMemcpy(sf->d_w, bd.w.data(), bd.w.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(sf->d_b, bd.b.data(), bd.b.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
    return sf_fail(AIGV_ERR_HIP, "aigv_slowfast_finalize: weight upload failed");
  sf->bufs.assign(B_COUNT, nullptr);
  for (int i = 0; i < B_COUNT; ++i) {
    const size_t bytes = sf->buf_elems[i] * (size_t)sf->Bcap * 2 + 64;
    if (hipMalloc((void**)&sf->bufs[i], bytes) != hipSuccess) return sf_fail(AIGV_ERR_ALLOC, "aigv_slowfast_finalize: hipMalloc(%zu) failed", bytes);
    if (hipMemset(sf->bufs[i], 0, bytes) != hipSuccess) return sf_fail(AIGV_ERR_HIP, "hipMemset failed");
  }
  // split-K workspace: the largest slab set any eligible conv can ask for at full capacity
  size_t need = 0;
  for (Op& op : sf->ops) {
    if (op.kind != OP_CONV) continue;
    op.k_slices = conv_k_slices(op.rows_per_clip, op.a.CoutPad, op.a.Kp);
    if (op.k_slices > 1) need = std::max(need, (size_t)op.k_slices * op.rows_per_clip * sf->Bcap * op.a.Cout);
  }
  sf->part_floats = need;
  if (sf->part_floats && hipMalloc((void**)&sf->d_part, sf->part_floats * sizeof(float)) != hipSuccess)
    return sf_fail(AIGV_ERR_ALLOC, "aigv_slowfast_finalize: hipMalloc of the split-K workspace failed");
  sf->host.clear();
  sf->finalized = true;
  return 0;
}

double aigv_slowfast_flops_per_clip(const aigv_slowfast* sf) { return sf ? sf->flops_per_clip : 0.0; }

int aigv_slowfast_forward(aigv_slowfast* sf, const void* frames_nchw_bf16, int clips, void* feature_bf16, void* stream) {
  if (!sf || !frames_nchw_bf16 || !feature_bf16) return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_forward: null argument");
  if (!sf->finalized) return sf_fail(AIGV_ERR_STATE, "aigv_slowfast_forward: call aigv_slowfast_finalize first");
  if (clips <= 0 || clips > sf->Bcap) return sf_fail(AIGV_ERR_ARG, "aigv_slowfast_forward: %d clips, capacity %d", clips, sf->Bcap);
  if (hipSetDevice(sf->device) != hipSuccess) return sf_fail(AIGV_ERR_HIP, "aigv_slowfast_forward: hipSetDevice(%d) failed", sf->device);
  hipStream_t s = (hipStream_t)stream;
  for (const Op& op : sf->ops) {
    hipError_t e = hipSuccess;
    switch (op.kind) {
      case OP_REPACK: {
        const long hw = (long)sf->H * sf->W, total = (long)clips * (sf->T + sf->Ts) * hw;
        hipLaunchKernelGGL(sf_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)frames_nchw_bf16, sf->T, sf->Ts, hw,
                           sf->bufs[B_IN1], sf->bufs[B_IN0], sf->slow_idx, total);
        e = hipGetLastError();
        break;
      }
      case OP_CONV: {
        ConvArgs a = op.a;
        a.in = sf->bufs[op.in_buf]; a.out = sf->bufs[op.out_buf]; a.res = op.res_buf >= 0 ? sf->bufs[op.res_buf] : nullptr;
        a.w = sf->d_w + op.w_off; a.bias = sf->d_b + op.b_off;
        a.rows = op.rows_per_clip * clips;
        e = launch_conv(a, s, op.k_slices, sf->d_part, sf->part_floats);
        break;
      }
      case OP_MAXPOOL: {
        const long total = (long)clips * op.T * op.Ho * op.Wo * (op.C / 8);
        hipLaunchKernelGGL(sf_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, sf->bufs[op.in_buf], op.C, op.Hi, op.Wi, op.Ho, op.Wo,
                           sf->bufs[op.out_buf], op.ld_out, total);
        e = hipGetLastError();
        break;
      }
      case OP_POOL: {
        hipLaunchKernelGGL(sf_pool_kernel, dim3((op.C + 31) / 32, clips), dim3(256), 0, s, sf->bufs[op.in_buf], op.T, op.Hi, op.Wi, op.C, op.pw,
                           (bf16_t*)feature_bf16, op.ld_out, op.c_off);
        e = hipGetLastError();
        break;
      }
    }
    if (e != hipSuccess) return sf_fail(e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP, "aigv_slowfast_forward: launch failed: %s", hipGetErrorString(e));
  }
  return 0;
}

// One convolution with the kernel the branch is built from (tests / profiling): x [B, Ti, Hi, Wi, ld_in] channels-last bf16,
// w [ceil16(Cout), Kp] packed as above, dims = {Ti,Hi,Wi, kt,kh,kw, st,sh,sw, pt,ph,pw}
int aigv_op_conv3d(const void* x, int ld_in, int Cin, int B, const int* dims, const void* w_packed, int Kp, const float* bias, int Cout, const void* res,
                   int ld_res, void* out, int ld_out, int c_off, int relu, void* stream) {
  if (!x || !dims || !w_packed || !bias || !out) return sf_fail(AIGV_ERR_ARG, "aigv_op_conv3d: null argument");
  ConvArgs a{};
  a.in = (const bf16_t*)x; a.ld_in = ld_in; a.Cin = Cin; a.Ti = dims[0]; a.Hi = dims[1]; a.Wi = dims[2];
  a.kt = dims[3]; a.kh = dims[4]; a.kw = dims[5]; a.st = dims[6]; a.sh = dims[7]; a.sw = dims[8]; a.pt = dims[9]; a.ph = dims[10]; a.pw = dims[11];
  if (a.kt <= 0 || a.kh <= 0 || a.kw <= 0 || a.st <= 0 || a.sh <= 0 || a.sw <= 0 || a.pt < 0 || a.ph < 0 || a.pw < 0 || B <= 0 || Cin <= 0 || Cout <= 0 ||
      ld_in < Cin || ld_out < c_off + Cout || (res && ld_res < Cout) || a.Ti + 2 * a.pt < a.kt || a.Hi + 2 * a.ph < a.kh || a.Wi + 2 * a.pw < a.kw)
    return sf_fail(AIGV_ERR_ARG, "aigv_op_conv3d: bad geometry");
  a.To = (a.Ti + 2 * a.pt - a.kt) / a.st + 1; a.Ho = (a.Hi + 2 * a.ph - a.kh) / a.sh + 1; a.Wo = (a.Wi + 2 * a.pw - a.kw) / a.sw + 1;
  a.w = (const bf16_t*)w_packed; a.Kp = Kp; a.bias = bias; a.Cout = Cout; a.CoutPad = (Cout + 15) / 16 * 16;
  a.res = (const bf16_t*)res; a.ld_res = ld_res; a.out = (bf16_t*)out; a.ld_out = ld_out; a.c_off = c_off; a.relu = relu;
  a.rows = (long)B * a.To * a.Ho * a.Wo;
  hipError_t e = launch_conv(a, (hipStream_t)stream);
  if (e != hipSuccess)
    return sf_fail(e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP,
                   "aigv_op_conv3d (Cin=%d Cout=%d Kp=%d; needs Cin, ld_in %% 8 == 0, Cout, ld_out, c_off %% 4 == 0, Kp %% 64 == 0): %s", Cin, Cout, Kp, hipGetErrorString(e));
  return 0;
}

}  // extern "C"
