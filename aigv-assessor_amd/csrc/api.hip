// C ABI (include/aigv_amd.h) over the gfx950 kernels: context, weight store, workspaces and the
// orchestration of the scorer hot path.  Host-side C++ only; every device op is one of the hand-written
// kernels in gemm.hip / attention.hip / rowops.hip / head.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/aigv_amd.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct VitLayer {
  const bf16_t *ls1, *ls2, *qkv_w, *qkv_b, *qn, *kn, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *n1w, *n1b, *n2w, *n2b;
};
struct LlmLayer {
  const bf16_t *wqkv, *wo, *w13, *w2, *an, *fn;
};

struct LlmLayerFp8 {   // e4m3 copies of a decoder layer's weights, one fp32 scale per output channel (aigv_set_precision)
  uint8_t *wqkv = nullptr, *wo = nullptr, *w13 = nullptr, *w2 = nullptr;
  float *s_wqkv = nullptr, *s_wo = nullptr, *s_w13 = nullptr, *s_w2 = nullptr;
};

struct ProfRec {
  int cls;
  hipEvent_t a, b;
  double flops, bytes;
};

// Row plan of one pass (round 4): how the rows of the activation matrices divide into INDEPENDENT sequences (InternViT frames,
// InternLM2 clips) and, from that alone, which kernel form every row runs in - so that a row's bits never depend on its batch mates:
//   body: rows [0, 256 * floor(L / 256)) of every sequence -> full-K 256x256 kernel, whole tiles addressed through the half-tile table;
//   tail: the remaining < 256 rows of every sequence      -> one or two (ragged) half tiles; run with a split-K factor that is a
//                                                            function of the GEMM's (N, K) only (1 = inside the body's launch);
//   tiny: tails of <= TINY_TAIL rows (InternViT: 1025 = 4 * 256 + 1) -> the weight-streaming skinny kernel in its fixed 4-slice form.
struct RowPlan {
  struct Tiny { int row0, stride_rows, count; };   // rows row0 + i * stride_rows, i < count
  std::vector<Tiny> tiny;
  int rows = 0, body_halves = 0, tail_halves = 0, tail_rows = 0;
  int32_t* d_tab = nullptr;     // device: (base row, valid rows) per half, body halves first
  int cap_halves = 0;
};

}  // namespace

struct aigv_ctx {
  aigv_config cfg{};
  int device = 0;
  std::string err;
  std::unordered_map<std::string, DevBuf> w;
  std::vector<void*> allocs;      // weight-side device memory: derived weights, e4m3 copies
  std::vector<void*> ws_allocs;   // capacity-sized workspaces (aigv_ctx_create / aigv_ctx_resize)
  bool ws_phase = false;          // dalloc books into ws_allocs while the workspaces are being allocated
  bool finalized = false;
  // derived sizes
  int np = 0, S = 0, Kp = 0, grid = 0, ntok = 0, proj_in = 0, qkv_out = 0, head_dim = 0, vit_head_dim = 0, g = 0;
  // derived weights
  std::vector<VitLayer> vit;
  std::vector<LlmLayer> llm;
  const bf16_t *patch_w = nullptr, *patch_b = nullptr, *pos = nullptr, *cls_pos = nullptr;
  const bf16_t *tok_emb = nullptr, *final_norm = nullptr, *lm_head = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
  const bf16_t *p_ln_w[2] = {nullptr, nullptr}, *p_ln_b[2] = {nullptr, nullptr}, *p_w1[2] = {nullptr, nullptr},
               *p_b1[2] = {nullptr, nullptr}, *p_w2[2] = {nullptr, nullptr}, *p_b2[2] = {nullptr, nullptr};  // 0 mlp1, 1 motion_mlp
  ScoreHeadArgs score{};
  // workspaces
  bf16_t *v_col = nullptr, *v_x = nullptr, *v_t = nullptr, *v_qkv = nullptr, *v_ao = nullptr, *v_h = nullptr;
  int32_t* v_cu = nullptr;
  bf16_t *p_t = nullptr, *p_mid = nullptr;
  bf16_t *l_h = nullptr, *l_t = nullptr, *l_qkv = nullptr, *l_ao = nullptr, *l_ffn = nullptr, *l_rows = nullptr;
  int32_t *l_pos = nullptr, *l_seq = nullptr, *l_cu = nullptr, *l_rowidx = nullptr, *l_rowidx2 = nullptr, *l_kvlen = nullptr;
  unsigned long long* l_packed = nullptr;
  int32_t* l_neg1 = nullptr;   // max_tokens x int32 -1: the "plain text token" slot map of aigv_llm_extend
  // fp8 mode of the InternLM2 prefill GEMMs (aigv_set_precision): weights quantised once, activations per row on the fly
  bool fp8_llm = false;
  bool llm_lin_dirty = false;  // an InternLM2 linear (wqkv / wo / w1 / w3 / w2) was (re)loaded since the e4m3 copies were made
  std::vector<LlmLayerFp8> llm8;
  uint8_t* q8 = nullptr;       // [max_tokens, max(H, I)] e4m3 activations of the GEMM about to run
  float* q8_scale = nullptr;   // [max_tokens]
  RowPlan rp_vit, rp_llm;           // row plans of the InternViT frames of the current chunk and of the current prefill's clips
  const RowPlan* cur_rp = nullptr;  // plan the InternLM2 layer helpers run under (null: batch-level dispatch, aigv_llm_extend)
  bool trim_last_layer = true;
  int attn_round_scores = AIGV_ATTENTION_NUMERICS_DEFAULT;   // prefill attention: 1 = the reference's bf16 rounding points of the score matrix, 0 = fp32 scores (default since round 5: profiles/r5_parity_stats.txt)
  int gemm_mode = -1;          // GEMM tile choice of this context: -1 = the process default (aigv_tune_gemm), else 0 / 1 / 2 / 3 / 4
  // the other experiment knobs of this context (aigv_ctx_tune): -1 = follow the process default (aigv_tune_*)
  int t_order = -1, t_variant = -1, t_attn_waves = -1, t_skinny_p = -1, t_body_tile = -1, t_co_kmax = -1;
  int t_tail_slices = -1, t_lead_key = -1, t_decode_fused = -1, t_decode_fp8 = -1, t_skinny_p8 = -1, t_fuse_tails = -1, t_lone_body = -1;
  size_t splitk_floats = 0;
  // fp32 slabs of the split-K row bands, owned by the context.  Two of them: aigv_vit_forward launches on `splitk_ws_vit`, every other entry
  // point on `splitk_ws` - so ONE visual front (InternViT on a stream of its own: InternVLChatModel.prefetch) may run beside ONE
  // projector / InternLM2 pass of the same context.  Within each half the rule stays: one launch stream at a time.
  float* splitk_ws = nullptr;
  float* splitk_ws_vit = nullptr;
  bool on_vit_front = false;   // host-side: set while aigv_vit_forward enqueues (SplitkVitScope)
  bf16_t* l_trim = nullptr;   // last-layer row trimming: compact [64, H] x 2 (attention out, normed) + [64, I], reused per 64 consumed rows
  bf16_t* l_trim_h = nullptr; // ... and the consumed rows' hidden states [max_out_rows + max_seqs + 64, H]
  bf16_t* l_score_ws = nullptr;
  bf16_t *kc = nullptr, *vc = nullptr;   // [layer][seq][kv head][cap][D]
  bf16_t *kc_alt = nullptr, *vc_alt = nullptr;   // second cache of the same size, made by the first aigv_kv_reorder (beam search gathers into it, then the two swap)
  int32_t* beam_ints = nullptr;                  // [2 * max_seqs]: parent slots | live lengths of a reorder
  float* dec_ws = nullptr;
  int32_t *dec_pos = nullptr, *dec_seq = nullptr, *dec_kvlen = nullptr, *dec_slot = nullptr;   // device-side decode state
  std::vector<int32_t> h_dec;
  std::vector<int32_t> h_pos, h_seq, h_rowidx, h_kvlen;
  int kv_seqs = 0;
  bool kv_valid = false;
  // profiling
  bool prof = false;
  int gemm_cls = AIGV_PROF_GEMM;   // class the GEMM launches are booked under: AIGV_PROF_GEMM_VIT inside aigv_vit_forward / aigv_project
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> ev_pool;
};

namespace {

int fail(aigv_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  g_err = buf;
  return code;
}

#define HIPCHK(c, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) return fail(c, AIGV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define TRY(x)            \
  do {                    \
    int r_ = (x);         \
    if (r_ != 0) return r_; \
  } while (0)

template <typename T>
int dalloc(aigv_ctx* c, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = (count ? count : 1) * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(c, AIGV_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  e = hipMemset(p, 0, bytes);
  if (e != hipSuccess) return fail(c, AIGV_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
  (c->ws_phase ? c->ws_allocs : c->allocs).push_back(p);
  *out = (T*)p;
  return 0;
}

inline uint16_t f32_to_bf16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_f32_host(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int roundup(int x, int m) { return (x + m - 1) / m * m; }

// ---- profiling brackets --------------------------------------------------------------------------------
hipEvent_t get_event(aigv_ctx* c) {
  if (!c->ev_pool.empty()) {
    hipEvent_t e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
struct ProfScope {
  aigv_ctx* c;
  hipStream_t s;
  ProfRec r{};
  bool on;
  ProfScope(aigv_ctx* c_, int cls, double flops, double bytes, hipStream_t s_) : c(c_), s(s_), on(c_ && c_->prof) {
    if (!on) return;
    r.cls = cls; r.flops = flops; r.bytes = bytes;
    r.a = get_event(c); r.b = get_event(c);
    if (!r.a || !r.b) { on = false; return; }
    hipEventRecord(r.a, s);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(r.b, s);
    c->recs.push_back(r);
  }
};

struct RowPlanScope {   // the InternLM2 layer helpers run under `rp` inside the scope
  aigv_ctx* c;
  RowPlanScope(aigv_ctx* c_, const RowPlan* rp) : c(c_) { c->cur_rp = rp; }
  ~RowPlanScope() { c->cur_rp = nullptr; }
};

struct SplitkVitScope {   // split-K launches inside the scope use the InternViT half of the context's scratch
  aigv_ctx* c;
  explicit SplitkVitScope(aigv_ctx* c_) : c(c_) { c->on_vit_front = true; }
  ~SplitkVitScope() { c->on_vit_front = false; }
};

struct GemmClassScope {   // GEMM launches inside the scope are booked under `cls` (per-class roofline entries of bench.py)
  aigv_ctx* c;
  int keep;
  GemmClassScope(aigv_ctx* c_, int cls) : c(c_), keep(c_->gemm_cls) { c->gemm_cls = cls; }
  ~GemmClassScope() { c->gemm_cls = keep; }
};

// ---- GEMM dispatch: split the rows over the tile kernels by a wave-quantisation cost model -----------------------
// Experiment knobs.  The kernel files hold no mutable state: every launch carries its selectors (GemmArgs::order_sel / variant_sel,
// AttnArgs::waves), filled in here from the context's own setting (aigv_ctx_tune / aigv_set_gemm_mode) or, where the context leaves a
// knob at -1 and for the context-free aigv_op_* entry points, from these process defaults (aigv_tune_*: tests and A/B scripts).
// gemm_mode: 0 = row plans (scoring pass) / cost model (op level), 1 = the 128x128 kernel, 2 = the 256x256 kernel, 3 = batch-level cost model,
//            4 = the co-resident 256x128 kernel
// co_kmax:   GEMMs with K <= co_kmax run on the co-resident 256x128 kernel (gemmco.hip) in modes 0 / 3 - a function of K alone, so a row's
//            kernel form never depends on its batch; 0 = never
constexpr int CO_KMAX_DEFAULT = 0;   // measured: profiles/r5_gemmco.txt
struct Tune {
  int gemm_mode = 0, order_sel = 0, variant_sel = 0, attn_waves = 0, skinny_p = 0, body_tile = 0, co_kmax = CO_KMAX_DEFAULT;
  // context-only experiment knobs (aigv_ctx_tune): split-K factor of the tails (0 = the per-shape rule), lead-key attention form for 64 j + 1 keys,
  // fused-norm decode GEMVs (1 = on), e4m3 decode GEMVs in fp8 mode (1 = on), form of the e4m3 decode GEMVs (0 = per-GEMV defaults)
  int tail_slices = 0, lead_key = 0, decode_fused = 1, decode_fp8 = 1, skinny_p8 = 0;
  int fuse_tails = 0;   // tail K slices inside the body's launch: 0 = when the body leaves CUs idle, 1 = never, 2 = always (same bits each way)
  int lone_body = 1;    // bodies of <= 128 tiles on the co-resident kernel's LONE form: 0 = by fill, 1 = never (default: no in-step gain measured), 2 = always (same bits each way)
};
Tune g_tune;
// the knobs in force for a call: the context's own setting, else the process default
int resolved_gemm_mode(const aigv_ctx* c) { return (c && c->gemm_mode >= 0) ? c->gemm_mode : g_tune.gemm_mode; }
Tune tune_of(const aigv_ctx* c) {
  Tune t = g_tune;
  if (c) {
    if (c->gemm_mode >= 0) t.gemm_mode = c->gemm_mode;
    if (c->t_order >= 0) t.order_sel = c->t_order;
    if (c->t_variant >= 0) t.variant_sel = c->t_variant;
    if (c->t_attn_waves >= 0) t.attn_waves = c->t_attn_waves;
    if (c->t_skinny_p >= 0) t.skinny_p = c->t_skinny_p;
    if (c->t_body_tile >= 0) t.body_tile = c->t_body_tile;
    if (c->t_co_kmax >= 0) t.co_kmax = c->t_co_kmax;
    if (c->t_tail_slices >= 0) t.tail_slices = c->t_tail_slices;
    if (c->t_lead_key >= 0) t.lead_key = c->t_lead_key;
    if (c->t_decode_fused >= 0) t.decode_fused = c->t_decode_fused;
    if (c->t_decode_fp8 >= 0) t.decode_fp8 = c->t_decode_fp8;
    if (c->t_skinny_p8 >= 0) t.skinny_p8 = c->t_skinny_p8;
    if (c->t_fuse_tails >= 0) t.fuse_tails = c->t_fuse_tails;
    if (c->t_lone_body >= 0) t.lone_body = c->t_lone_body;
  }
  return t;
}
GemmArgs tuned(const aigv_ctx* c, const GemmArgs& a) {
  GemmArgs b = a;
  const Tune t = tune_of(c);
  b.order_sel = t.order_sel; b.variant_sel = t.variant_sel;
  return b;
}
// Model constants, microseconds on one MI355X (scripts/gemm_overhead.py: time vs K at fixed M, N):
//   a full round of the 256 kernel (256 tiles, one per CU) takes nk * kt256 + fix256; a full round of the 128 kernel
//   (512 tiles, two co-resident workgroups per CU) nk * KT128 + FIX128, a last round of <= 256 tiles LONE128 of that.
double g_rate256 = 1.46;                 // throughput of the 256 kernel relative to the 128 kernel: kt256 = 2 * KT128 / rate
constexpr double KT128 = 0.95, FIX128 = 6.0, LONE128 = 0.70, FIX256 = 9.0, LAUNCH_GAP = 2.0;
double kt256() { return 2.0 * KT128 / g_rate256; }

double t256(long row_tiles, int N, int nk) {
  if (row_tiles <= 0) return 0;
  const long tiles = row_tiles * (N / 256);
  return (double)((tiles + 255) / 256) * (nk * kt256() + FIX256);
}
double t128_blocks(long blocks, double nk_each) {
  const double rt = nk_each * KT128 + FIX128;
  const long full = blocks / 512, part = blocks % 512;
  return full * rt + (part == 0 ? 0.0 : part <= 256 ? LONE128 * rt : rt);
}
double t128(int rows, int N, int nk) { return rows <= 0 ? 0 : t128_blocks((long)((rows + 127) / 128) * (N / 128), nk); }
// slabs read once + the bf16 result (and residual) at ~3 TB/s, plus the launch
double t_finalize(long rows, int N, int S) { return (double)rows * N * (4.0 * S + 4.0) / 3.0e6 + 3.0; }

// fp32 scratch for split-K slices.  A context owns its own (allocated at aigv_ctx_create on the context's device, SPLITK_MAX_FLOATS:
// the planner never asks for more).  The context-free single-operator entry points (aigv_op_gemm: tests, benches) use one scratch
// per DEVICE, created on first use under a lock and kept for the process lifetime - never freed or regrown, so no launch ever
// waits on a device synchronisation, and a second device never sees memory of the first.
constexpr size_t SPLITK_MAX_FLOATS = (size_t)64 << 20;   // 256 MB
constexpr int MAX_DEVICES = 64;
float* g_op_splitk_ws[MAX_DEVICES] = {};
std::mutex g_op_splitk_lock;

int splitk_scratch(aigv_ctx* c, size_t need_floats, float** out) {
  if (need_floats > SPLITK_MAX_FLOATS) return fail(c, AIGV_ERR_ARG, "split-K scratch: %zu floats exceed the planner's cap", need_floats);
  if (c) {
    if (need_floats > c->splitk_floats) return fail(c, AIGV_ERR_STATE, "split-K scratch: %zu floats exceed the context's %zu", need_floats, c->splitk_floats);
    *out = c->on_vit_front ? c->splitk_ws_vit : c->splitk_ws;
    return 0;
  }
  int dev = 0;
  HIPCHK(c, hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return fail(c, AIGV_ERR_ARG, "device %d out of range", dev);
  std::lock_guard<std::mutex> g(g_op_splitk_lock);
  if (!g_op_splitk_ws[dev]) HIPCHK(c, hipMalloc((void**)&g_op_splitk_ws[dev], SPLITK_MAX_FLOATS * sizeof(float)));
  *out = g_op_splitk_ws[dev];
  return 0;
}
constexpr int SPLITS[] = {2, 3, 4, 6, 8};

// rows that do not fill whole rounds are latency-bound on their K loop: split K over S workgroups per tile (fp32 slabs +
// a fixed-order finalize pass).  Best S for the 128 kernel / for `row_tiles` x 256 rows on the 256 kernel; S = 1: no split.
double best_split128(int rows, int N, int nk, int epi, int* S_out) {
  *S_out = 1;
  double best = t128(rows, N, nk);
  if (epi == EPI_PATCH) return best;
  const long tiles = (long)((rows + 127) / 128) * (N / 128);
  for (int S : SPLITS) {
    if (nk % S || nk / S < 4 || (size_t)S * rows * N > SPLITK_MAX_FLOATS) continue;
    const double t = t128_blocks(tiles * S, (double)nk / S) + t_finalize(rows, N, S) + LAUNCH_GAP;
    if (t < best) { best = t; *S_out = S; }
  }
  return best;
}
double best_split256(long row_tiles, int N, int nk, int* S_out) {
  *S_out = 0;
  double best = 1e30;
  for (int S : SPLITS) {
    if (nk % S || nk / S < 4 || (size_t)S * row_tiles * 256 * N > SPLITK_MAX_FLOATS) continue;
    const long blocks = row_tiles * (N / 256) * S;
    const double t = (double)((blocks + 255) / 256) * ((double)nk / S * kt256() + FIX256) + t_finalize(row_tiles * 256, N, S) + LAUNCH_GAP;
    if (t < best) { best = t; *S_out = S; }
  }
  return best;
}

int skinny_epi(int epi) {   // GEMM epilogue -> skinny-kernel epilogue (same rounding points); -1 if none
  switch (epi) {
    case EPI_STORE: return 0;
    case EPI_GELU: return 3;
    case EPI_LS_RESID: return 6;
    case EPI_RESID: return 1;
    case EPI_SWIGLU: return 2;
  }
  return -1;
}
// a <= 64-row remainder streamed through the skinny kernel costs ~ the weight bytes at HBM rate
double t_skinny(int rows, int N, int K) {
  if (rows <= 0) return 0;
  if (rows > 64 || K % 128) return 1e30;
  return (double)N * K * 2.0 / 4.5e6 + 4.0;
}

#define GEMM_PROF(c, a, s) ProfScope ps(c, (c) ? (c)->gemm_cls : AIGV_PROF_GEMM, 2.0 * (a).M * (double)(a).N * (a).K, \
                                       2.0 * ((double)(a).M * (a).K + (double)(a).N * (a).K + (double)(a).M * (a).N), s)

int launch_one(aigv_ctx* c, const GemmArgs& a, int epi, bool use256, hipStream_t s) {
  GEMM_PROF(c, a, s);
  HIPCHK(c, use256 ? aigv_launch_gemm256(tuned(c, a), epi, s) : aigv_launch_gemm(a, epi, s));
  return 0;
}

// the co-resident 256x128 kernel (gemmco.hip): every row in full K, ragged row counts and half-tile tables included
int launch_co(aigv_ctx* c, const GemmArgs& a, int epi, hipStream_t s) {
  GEMM_PROF(c, a, s);
  HIPCHK(c, aigv_launch_gemmco(tuned(c, a), epi, s));
  return 0;
}
// does this GEMM run on the co-resident kernel?  A function of the mode and of K only.
bool use_co(const aigv_ctx* c, const GemmArgs& a, int mode) {
  if (!aigv_gemmco_supported(a)) return false;
  return mode == 4 || ((mode == 0 || mode == 3) && a.K <= tune_of(c).co_kmax);
}

int launch_splitk(aigv_ctx* c, const GemmArgs& a, int epi, int S, bool tile256, hipStream_t s) {
  float* ws = nullptr;
  TRY(splitk_scratch(c, (size_t)S * a.M * a.N, &ws));
  GEMM_PROF(c, a, s);
  HIPCHK(c, aigv_launch_gemm_splitk(tuned(c, a), epi, S, ws, s, tile256));
  return 0;
}

GemmArgs row_slice(const GemmArgs& a, int row0, int rows) {
  GemmArgs b = a;
  b.M = rows;
  b.A = a.A + (size_t)row0 * a.lda;
  b.C = a.C + (size_t)row0 * a.ldc;
  if (a.resid) b.resid = a.resid + (size_t)row0 * a.ldr;
  return b;
}

// Rows are independent, so the GEMM is cut into up to three row bands, each on the kernel that wastes least:
//   top:  R x 256 rows on the 256x256 kernel, R chosen so that it runs whole rounds
//   mid:  Q x 256 rows on the 256x256 kernel with split-K (a partial round made of K slices)
//   last: the remaining rows on the weight-streaming skinny kernel (<= 64 rows) or the 128x128 kernel (plain or split-K)
struct GemmPlan {
  int top_tiles = 0;      // R; -1: the whole problem (ragged last tile included) in one launch of the 256 kernel
  int mid_tiles = 0;      // Q
  int mid_slices = 0;     // split-K factor of the mid band
  int last_rows = 0;
  int last_kind = 0;      // 0 none, 1 skinny, 2 the 128 kernel
  int last_slices = 1;    // split-K factor of the last band on the 128 kernel (1: plain)
  double est_us = 0;
};

GemmPlan plan_gemm(int M, int N, int K, int epi, int mode) {
  GemmPlan pl;
  const int nk = K / 64, sk = skinny_epi(epi);
  const bool ok256 = (N % 256 == 0);
  if (mode == 1 || !ok256) { pl.last_rows = M; pl.last_kind = 2; pl.est_us = t128(M, N, nk); return pl; }
  pl.top_tiles = -1;
  pl.est_us = t256((M + 255) / 256, N, nk);
  if (mode == 2) return pl;
  if (epi == EPI_PATCH) {
    if (t128(M, N, nk) < pl.est_us) { pl = GemmPlan(); pl.last_rows = M; pl.last_kind = 2; pl.est_us = t128(M, N, nk); }
    return pl;
  }
  const int full_tiles = M / 256;
  for (int R = 0; R <= full_tiles; ++R) {
    for (int Q = 0; Q <= 8 && R + Q <= full_tiles; ++Q) {
      const int rem = M - (R + Q) * 256;
      int S256 = 0, S128 = 1, last = 0;
      double t = t256(R, N, nk);
      if (Q > 0) {
        const double tm = best_split256(Q, N, nk, &S256);
        if (!S256) continue;
        t += tm;
      }
      if (rem > 0) {
        const double tk = best_split128(rem, N, nk, epi, &S128);
        const double ts = sk >= 0 ? t_skinny(rem, N, K) : 1e30;
        last = ts < tk ? 1 : 2;
        if (last == 1) S128 = 1;
        t += ts < tk ? ts : tk;
      }
      t += LAUNCH_GAP * ((R > 0) + (Q > 0) + (rem > 0) - 1);
      if (t < pl.est_us) {
        pl.est_us = t; pl.top_tiles = R; pl.mid_tiles = Q; pl.mid_slices = S256; pl.last_rows = rem; pl.last_kind = last;
        pl.last_slices = S128;
      }
    }
  }
  return pl;
}

// Columns are independent too: N = 256 j + 128 (InternViT-6B: 3200, 9600) would put the whole GEMM on the 128 kernel; instead the
// first 256 j columns take the row-band plan and only the last 128 columns run on the 128 kernel.
int split_columns(int M, int N, int K, int epi, int mode) {   // width of the right-hand 128-kernel band, 0 = no column split
  if (mode != 0 || N % 256 != 128 || N < 384 || epi == EPI_PATCH || epi == EPI_SWIGLU) return 0;
  const int nk = K / 64;
  const double whole = t128(M, N, nk);
  const double split = plan_gemm(M, N - 128, K, epi, mode).est_us + t128(M, 128, nk) + LAUNCH_GAP;
  return split < whole ? 128 : 0;
}

GemmArgs col_slice(const GemmArgs& a, int n0, int n) {
  GemmArgs b = a;
  b.N = n;
  b.W = a.W + (size_t)n0 * a.ldw;
  b.C = a.C + n0;
  if (a.bias) b.bias = a.bias + n0;
  if (a.ls) b.ls = a.ls + n0;
  if (a.resid) b.resid = a.resid + n0;
  return b;
}

int run_gemm(aigv_ctx* c, const GemmArgs& a, int epi, hipStream_t s) {
  if (const char* m = aigv_gemm_check(a, epi)) return fail(c, AIGV_ERR_ARG, "%s (M=%d N=%d K=%d epi=%d)", m, a.M, a.N, a.K, epi);
  if (use_co(c, a, resolved_gemm_mode(c))) return launch_co(c, a, epi, s);
  const int mode = resolved_gemm_mode(c) == 3 || resolved_gemm_mode(c) == 4 ? 0 : resolved_gemm_mode(c);
  if (const int right = split_columns(a.M, a.N, a.K, epi, mode)) {
    TRY(run_gemm(c, col_slice(a, 0, a.N - right), epi, s));
    return launch_one(c, col_slice(a, a.N - right, right), epi, false, s);
  }
  const GemmPlan pl = plan_gemm(a.M, a.N, a.K, epi, mode);
  if (pl.top_tiles < 0) return launch_one(c, a, epi, true, s);
  int row = 0;
  if (pl.top_tiles > 0) {
    TRY(launch_one(c, row_slice(a, 0, pl.top_tiles * 256), epi, true, s));
    row = pl.top_tiles * 256;
  }
  if (pl.mid_tiles > 0) {
    TRY(launch_splitk(c, row_slice(a, row, pl.mid_tiles * 256), epi, pl.mid_slices, true, s));
    row += pl.mid_tiles * 256;
  }
  if (row < a.M) {
    const GemmArgs bot = row_slice(a, row, a.M - row);
    if (pl.last_kind == 1) {
      const int sk = skinny_epi(epi);
      ProfScope ps(c, c ? c->gemm_cls : AIGV_PROF_GEMM, 2.0 * bot.M * (double)a.N * a.K, 2.0 * (double)a.N * a.K, s);
      hipError_t e = aigv_launch_skinny_gemm(bot.A, bot.lda, bot.M, bot.W, bot.ldw, bot.N, bot.K, bot.bias, bot.resid, bot.ldr,
                                            bot.C, bot.ldc, sk, s, bot.ls);
      if (e != hipSuccess) return fail(c, AIGV_ERR_HIP, "skinny remainder (M=%d N=%d K=%d): %s", bot.M, bot.N, bot.K, hipGetErrorString(e));
      return 0;
    }
    if (pl.last_slices > 1) return launch_splitk(c, bot, epi, pl.last_slices, false, s);
    return launch_one(c, bot, epi, false, s);
  }
  return 0;
}

// ---- per-sequence row plans (struct RowPlan above) ------------------------------------------------------------------------------------
constexpr int TINY_TAIL = 4;

// cu[0..n_seq]: row offsets of the sequences inside the activation matrices.  Builds the plan and uploads its table (kernel-argument
// writes on `s`: no host synchronisation).  The table is written by EVERY pass, never skipped for a plan "already on the device": a pass
// captured into a HIP graph must carry its own table writes (a replay after a pass of another shape would otherwise run on that pass's table),
// and a replayed graph rewrites the device table behind the host's back - so there is no host-side notion of what the device table holds.
int build_row_plan(aigv_ctx* c, RowPlan& rp, const int32_t* cu, int n_seq, hipStream_t s) {
  rp.tiny.clear();
  std::vector<int32_t> body, tail;
  bool uniform = true;
  for (int b = 1; b < n_seq; ++b) uniform = uniform && (cu[b + 1] - cu[b] == cu[1] - cu[0]);
  rp.tail_rows = 0;
  for (int b = 0; b < n_seq; ++b) {
    const int L = cu[b + 1] - cu[b], nb = L / 256, rem = L % 256, r0 = cu[b] + nb * 256;
    for (int j = 0; j < 2 * nb; ++j) { body.push_back(cu[b] + j * 128); body.push_back(128); }
    if (rem == 0) continue;
    if (rem <= TINY_TAIL) {
      if (!uniform) rp.tiny.push_back({r0, 1, rem});
      else if (b == 0)
        for (int j = 0; j < rem; ++j) rp.tiny.push_back({r0 + j, L, n_seq});
      continue;
    }
    tail.push_back(r0); tail.push_back(std::min(rem, 128));
    if (rem > 128) { tail.push_back(r0 + 128); tail.push_back(rem - 128); }
    rp.tail_rows += rem;
  }
  rp.body_halves = (int)body.size() / 2;
  rp.tail_halves = (int)tail.size() / 2;
  rp.rows = cu[n_seq];
  if (rp.body_halves + rp.tail_halves > rp.cap_halves)
    return fail(c, AIGV_ERR_STATE, "row plan: %d half tiles exceed the table's %d", rp.body_halves + rp.tail_halves, rp.cap_halves);
  body.insert(body.end(), tail.begin(), tail.end());
  if (!body.empty()) HIPCHK(c, aigv_launch_write_ints(body.data(), (int)body.size(), rp.d_tab, s));
  return 0;
}

// Split-K factor of the TAIL half tiles of a GEMM: a function of (N, K) only - never of the batch or of the number of clips: the largest
// S in {2, 4, 8} that leaves every slice >= 8 K-tiles and keeps ONE tail row tile's slices (N / 256 x S workgroups) within half a round of
// the chip (two tail row tiles - four clips' 128-row remainders - then come to about one round; eight clips to two).  1 = the tail rides in
// the body's launch.  `forced` > 0 (AIGV_TUNE_TAIL_SLICES, experiments): one factor for every shape it divides.
int tail_slices(int N, int K, int forced) {
  const int tn = N / 256, nk = K / 64;
  int best = 1;
  for (int S : {2, 4, 8})
    if (nk % S == 0 && nk / S >= 8 && 2 * tn * S <= 256) best = S;
  if (forced == 1 || (forced > 1 && nk % forced == 0 && nk / forced >= 4)) best = forced;
  return best;
}

int launch_tab(aigv_ctx* c, const GemmArgs& a, int epi, const int32_t* tab, int halves, int rows, int S, hipStream_t s) {
  if (halves <= 0) return 0;
  GemmArgs b = tuned(c, a);
  b.row_tab = tab; b.tab_halves = halves;
  GemmArgs pf = a; pf.M = rows;            // the rows this launch really computes (profile records only)
  GEMM_PROF(c, pf, s);
  if (S <= 1) {
    HIPCHK(c, aigv_launch_gemm256(b, epi, s));
    return 0;
  }
  float* ws = nullptr;
  TRY(splitk_scratch(c, (size_t)S * ((halves + 1) / 2) * 256 * a.N, &ws));
  HIPCHK(c, aigv_launch_gemm_splitk(b, epi, S, ws, s, true));
  return 0;
}

int run_tiny_tails(aigv_ctx* c, const GemmArgs& a, int epi, const RowPlan& rp, hipStream_t s);

// A GEMM whose rows follow the row plan `rp` (mode 0); modes 1 / 2 / 4 run every row on one tile kernel in full K (also batch-invariant).
int run_gemm_rows(aigv_ctx* c, const GemmArgs& a, int epi, const RowPlan& rp, hipStream_t s) {
  if (const char* m = aigv_gemm_check(a, epi)) return fail(c, AIGV_ERR_ARG, "%s (M=%d N=%d K=%d epi=%d)", m, a.M, a.N, a.K, epi);
  if (a.M != rp.rows) return fail(c, AIGV_ERR_STATE, "row plan covers %d rows, the GEMM has %d", rp.rows, a.M);
  const int mode = resolved_gemm_mode(c);
  if (mode == 3) return run_gemm(c, a, epi, s);   // rounds 1-3: batch-level cost-model dispatch (A/B only: not batch-invariant)
  if (mode == 4) return use_co(c, a, mode) ? launch_co(c, a, epi, s) : launch_one(c, a, epi, false, s);
  if (use_co(c, a, mode)) {
    // short-K GEMMs (InternViT): body AND tail half tiles in one launch of the co-resident kernel, full K; tiny tails on the skinny kernel as below
    if (rp.body_halves + rp.tail_halves > 0) {
      GemmArgs b = a;
      b.row_tab = rp.d_tab; b.tab_halves = rp.body_halves + rp.tail_halves;
      GemmArgs pf = a; pf.M = rp.body_halves * 128 + rp.tail_rows;
      GEMM_PROF(c, pf, s);
      HIPCHK(c, aigv_launch_gemmco(tuned(c, b), epi, s));
    }
    return run_tiny_tails(c, a, epi, rp, s);
  }
  if (mode == 1 || a.N < 256) return launch_one(c, a, epi, false, s);
  if (mode == 2 && a.N % 256 == 0) return launch_one(c, a, epi, true, s);
  if (a.N % 256) {   // N = 256 j + 128 (InternViT-6B: 3200, 9600): the last 128 columns of every row on the 128 kernel, full K
    if (epi == EPI_SWIGLU) return launch_one(c, a, epi, false, s);
    TRY(run_gemm_rows(c, col_slice(a, 0, a.N - 128), epi, rp, s));
    return launch_one(c, col_slice(a, a.N - 128, 128), epi, false, s);
  }
  const int S = tail_slices(a.N, a.K, tune_of(c).tail_slices);
  // The body rows may run on either tile kernel: both sum every output element over the full K in the same order, so not one bit moves
  // (tests/test_gpu_ops.py pins that).  Shipped: always the 256 tiles - for one clip, whose wo / w2 / ViT proj / fc2 bodies are only 128
  // tiles, the 128 kernel (512 tiles, two per CU) was expected to win by the cost model and measured 1-3 % slower per clip
  // (profiles/r4_negatives.txt, 6); body_tile = 2 keeps the 128 form reachable for tests.
  const bool body128 = rp.body_halves > 0 && tune_of(c).body_tile == 2;
  const bool tails_apart = rp.tail_halves > 0 && S > 1;
  // A body of at most 128 tiles (one clip's wo / w2, eight frames' proj / fc2) leaves half the CUs idle: the co-resident kernel's LONE form
  // (gemmco.hip VAR 6: 256 x 128 tiles, one 8-wave workgroup per CU, four of the waves only issue the LDS-DMA requests) gives every CU a
  // tile - the same bits as the 256 kernel, 18-24 % less time on these bodies in isolation (scripts/gemm_body_ab.py) and NOTHING inside the
  // one-clip forward (37.9-38.1 ms per clip either way: profiles/r5_loop_shape.txt; the tail's K slices can no longer ride in the idle half of
  // the chip, and the SlowFast branch's side-stream kernels lose the CUs the half-empty bodies left them).  Off by default (lone_body = 1).
  const long body_tiles = (long)(rp.body_halves / 2) * (a.N / 256);
  const int lone_knob = tune_of(c).lone_body;   // 0 = by fill, 1 = never, 2 = whenever the shapes allow
  const bool lone = !body128 && lone_knob != 1 && rp.body_halves > 0 && a.N % 128 == 0 &&
                    (lone_knob == 2 || (body_tiles <= 128 && (!tails_apart || a.K / 64 >= 128) && (lone_knob != 3 || tails_apart) && (lone_knob != 4 || !tails_apart)));
  if (lone) {
    GemmArgs b = tuned(c, a);
    b.row_tab = rp.d_tab; b.tab_halves = tails_apart ? rp.body_halves : rp.body_halves + rp.tail_halves;
    b.variant_sel = 7;
    GemmArgs pf = a; pf.M = rp.body_halves * 128 + (tails_apart ? 0 : rp.tail_rows);
    GEMM_PROF(c, pf, s);
    HIPCHK(c, aigv_launch_gemmco(b, epi, s));
  } else if (body128) {
    {
      GemmArgs b = a;
      b.row_tab = rp.d_tab; b.tab_halves = rp.body_halves;
      GemmArgs pf = a; pf.M = rp.body_halves * 128;
      GEMM_PROF(c, pf, s);
      HIPCHK(c, aigv_launch_gemm(b, epi, s));
    }
    if (rp.tail_halves > 0 && !tails_apart) TRY(launch_tab(c, a, epi, rp.d_tab + 2 * rp.body_halves, rp.tail_halves, rp.tail_rows, 1, s));
  } else if (!tails_apart) {
    TRY(launch_tab(c, a, epi, rp.d_tab, rp.body_halves + rp.tail_halves, rp.body_halves * 128 + rp.tail_rows, 1, s));
  } else {
    // A body that leaves part of its last round of CUs idle (one or two clips) takes the tail's K slices into its own launch: the same
    // slices, slabs and finalize pass as the two-launch form - not one bit differs, so the choice may follow the fill (fuse_tails).
    const int tn = a.N / 256;
    const long body_wg = (long)(rp.body_halves / 2) * tn, slice_wg = (long)((rp.tail_halves + 1) / 2) * tn * S;
    const size_t need = (size_t)S * ((rp.tail_halves + 1) / 2) * 256 * a.N;
    const int fuse_knob = tune_of(c).fuse_tails;   // 0 = by fill, 1 = never, 2 = whenever the shapes allow
    const bool by_fill = body_wg % 256 != 0 && body_wg % 256 + slice_wg <= 320;
    if (!lone && rp.body_halves > 0 && (rp.body_halves & 1) == 0 && need <= (c ? c->splitk_floats : SPLITK_MAX_FLOATS) && fuse_knob != 1 && (by_fill || fuse_knob == 2)) {
      float* ws = nullptr;
      TRY(splitk_scratch(c, need, &ws));
      GemmArgs b = tuned(c, a);
      b.row_tab = rp.d_tab; b.tab_halves = rp.body_halves; b.fuse_tail_halves = rp.tail_halves; b.part = ws; b.k_slices = S;
      GemmArgs pf = a; pf.M = rp.body_halves * 128 + rp.tail_halves * 128;
      GEMM_PROF(c, pf, s);
      HIPCHK(c, aigv_launch_gemm256_fused(b, epi, s));
      GemmArgs f = a;
      f.row_tab = rp.d_tab + 2 * rp.body_halves; f.tab_halves = rp.tail_halves;
      HIPCHK(c, aigv_launch_gemm_finalize(f, epi, S, ws, s));
      return run_tiny_tails(c, a, epi, rp, s);
    }
    if (!lone) TRY(launch_tab(c, a, epi, rp.d_tab, rp.body_halves, rp.body_halves * 128, 1, s));
  }
  if (tails_apart) {
    const size_t per_pair = (size_t)S * 256 * a.N;
    const size_t cap = c ? c->splitk_floats : SPLITK_MAX_FLOATS;
    const int max_halves = (int)std::min<size_t>(cap / per_pair, 4096) * 2;
    if (max_halves < 2) return fail(c, AIGV_ERR_STATE, "split-K scratch too small for one tail tile (N=%d, %d slices)", a.N, S);
    for (int h0 = 0; h0 < rp.tail_halves; h0 += max_halves) {
      const int nh = std::min(max_halves, rp.tail_halves - h0);
      TRY(launch_tab(c, a, epi, rp.d_tab + 2 * (rp.body_halves + h0), nh, nh * 128, S, s));   // (profile: ragged halves counted as full)
    }
  }
  return run_tiny_tails(c, a, epi, rp, s);
}

// tails of <= TINY_TAIL rows (InternViT: 1025 = 4 * 256 + 1) on the weight-streaming skinny kernel in its fixed 4-slice form
int run_tiny_tails(aigv_ctx* c, const GemmArgs& a, int epi, const RowPlan& rp, hipStream_t s) {
  const int sk = skinny_epi(epi);
  for (const RowPlan::Tiny& t : rp.tiny) {
    if (sk < 0 || a.K % 128) return fail(c, AIGV_ERR_STATE, "no skinny form for epilogue %d / K=%d (tiny sequence tails)", epi, a.K);
    const size_t ldx = (size_t)t.stride_rows * a.lda, ldo = (size_t)t.stride_rows * a.ldc, ldr = (size_t)t.stride_rows * a.ldr;
    if (ldx > 0x7fffffffu || ldo > 0x7fffffffu || ldr > 0x7fffffffu) return fail(c, AIGV_ERR_STATE, "tiny-tail row stride overflows");
    for (int i0 = 0; i0 < t.count; i0 += 64) {
      const int R = std::min(64, t.count - i0);
      const size_t r0 = (size_t)t.row0 + (size_t)i0 * t.stride_rows;
      ProfScope ps(c, c ? c->gemm_cls : AIGV_PROF_GEMM, 2.0 * R * (double)a.N * a.K, 2.0 * (double)a.N * a.K, s);
      hipError_t e = aigv_launch_skinny_gemm(a.A + r0 * a.lda, (int)ldx, R, a.W, a.ldw, a.N, a.K, a.bias, a.resid ? a.resid + r0 * a.ldr : nullptr,
                                            (int)ldr, a.C + r0 * a.ldc, (int)ldo, sk, s, a.ls, 0);
      if (e != hipSuccess) return fail(c, AIGV_ERR_HIP, "skinny tiny tails (R=%d N=%d K=%d): %s", R, a.N, a.K, hipGetErrorString(e));
    }
  }
  return 0;
}

// every row in full K on one tile kernel (the 256 kernel where its shape rules allow): batch-invariant for any row layout
int run_gemm_full(aigv_ctx* c, const GemmArgs& a, int epi, hipStream_t s) {
  if (const char* m = aigv_gemm_check(a, epi)) return fail(c, AIGV_ERR_ARG, "%s (M=%d N=%d K=%d epi=%d)", m, a.M, a.N, a.K, epi);
  const int mode = resolved_gemm_mode(c);
  if (use_co(c, a, mode)) return launch_co(c, a, epi, s);
  if (mode == 1 || a.N < 256 || (a.N % 256 && epi == EPI_SWIGLU)) return launch_one(c, a, epi, false, s);
  if (a.N % 256) {
    TRY(launch_one(c, col_slice(a, 0, a.N - 128), epi, true, s));
    return launch_one(c, col_slice(a, a.N - 128, 128), epi, false, s);
  }
  return launch_one(c, a, epi, true, s);
}

// InternLM2 linears: under the current pass's row plan (aigv_llm_prefill), else the batch-level dispatch (aigv_llm_extend)
int run_llm_gemm(aigv_ctx* c, const GemmArgs& a, int epi, hipStream_t s) {
  return c->cur_rp ? run_gemm_rows(c, a, epi, *c->cur_rp, s) : run_gemm(c, a, epi, s);
}

GemmArgs gemm_args(const bf16_t* A, int lda, const bf16_t* W, int ldw, bf16_t* C, int ldc, int M, int N, int K) {
  GemmArgs a{};
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  return a;
}

// One InternLM2 linear in fp8 mode: quantise the bf16 activation rows (per-row amax / 448), then the e4m3 form of the 256 kernel with
// the bf16 path's epilogue.  The quantisation pass is outside the profiled launch (it is not GEMM work).
int run_gemm_fp8(aigv_ctx* c, const bf16_t* A, int lda, int K, const uint8_t* W8, const float* w_scale, bf16_t* C, int ldc, int T, int N,
                 int epi, const bf16_t* resid, int ldr, hipStream_t s) {
  hipError_t e = hipSuccess;
  if (A) {   // A == nullptr: c->q8 / c->q8_scale already hold the quantised rows (RMSNorm fused with the quantisation)
    e = aigv_launch_quant_fp8_rows(A, lda, T, K, c->q8, K, c->q8_scale, s);
    if (e != hipSuccess) return fail(c, AIGV_ERR_HIP, "fp8 activation quantisation (T=%d K=%d): %s", T, K, hipGetErrorString(e));
  }
  GemmArgs a = tuned(c, GemmArgs{});
  a.A = (const bf16_t*)c->q8; a.lda = K; a.W = (const bf16_t*)W8; a.ldw = K; a.C = C; a.ldc = ldc; a.M = T; a.N = N; a.K = K;
  a.row_scale = c->q8_scale; a.col_scale = w_scale; a.resid = resid; a.ldr = ldr;
  ProfScope ps(c, AIGV_PROF_GEMM_FP8, 2.0 * T * (double)N * K, (double)T * K + (double)N * K + 2.0 * T * (epi == EPI_SWIGLU ? N / 2 : N), s);
  // ONE launch over all rows, every row in full K on the one e4m3 tile kernel: an output element's sum then runs over K in the same order
  // wherever its row sits, so a clip's bits do not depend on its batch mates - in this mode too (round 6; until round 5 the partial last
  // round of a batch ran as K slices, which tied a row's summation order to the size of the batch).
  e = aigv_launch_gemm256_fp8(a, epi, s);
  if (e != hipSuccess) return fail(c, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP, "fp8 gemm (M=%d N=%d K=%d epi=%d): %s", T, N, K, epi, hipGetErrorString(e));
  return 0;
}


int run_skinny(aigv_ctx* c, const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K, const bf16_t* bias,
               const bf16_t* resid, int ldr, bf16_t* out, int ldo, int epi, hipStream_t s, int p = 1) {
  ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * R * (double)N * K, 2.0 * (double)N * K, s);
  // p = 0 (the scoring pass: last-layer consumed rows, motion_mlp, tiny sequence tails): the fixed 4-slice form whatever the row count, so
  // that a row's bits do not depend on its batch; p = 1 / 2 / 4 are the decode step's forms (decode_forms).
  hipError_t e = aigv_launch_skinny_gemm(x, ldx, R, W, ldw, N, K, bias, resid, ldr, out, ldo, epi, s, nullptr, p);
  if (e != hipSuccess) return fail(c, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP,
                                   "skinny gemm (R=%d N=%d K=%d epi=%d): %s", R, N, K, epi, hipGetErrorString(e));
  return 0;
}

int need(aigv_ctx* c, const std::string& name, size_t elems, const bf16_t** out) {
  auto it = c->w.find(name);
  if (it == c->w.end()) return fail(c, AIGV_ERR_STATE, "weight '%s' was never loaded", name.c_str());
  if (it->second.bytes != elems * 2)
    return fail(c, AIGV_ERR_STATE, "weight '%s' has %zu elements, expected %zu", name.c_str(), it->second.bytes / 2, elems);
  *out = (const bf16_t*)it->second.p;
  return 0;
}

}  // namespace


void aigv_set_error(const char* msg) { g_err = msg ? msg : ""; }

// ==========================================================================================================
extern "C" {

int aigv_abi_version(void) { return AIGV_ABI_VERSION; }
int aigv_sizeof_config(void) { return (int)sizeof(aigv_config); }

const char* aigv_last_error(const aigv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

void aigv_clear_hip_error(void) { (void)hipGetLastError(); }

// Everything whose size depends on the capacities of aigv_config (frames, tokens, sequences, output rows, KV): activations, index
// arrays, split-K scratch, KV caches.  Booked in ws_allocs so that aigv_ctx_resize can replace them without touching the weights.
static int alloc_workspaces(aigv_ctx* c) {
  const aigv_config& k = c->cfg;
  hipError_t e = hipSuccess;
  int rc = 0;
  c->ws_phase = true;
  c->kc = c->vc = nullptr;   // (no KV capacity: no caches)
  c->kc_alt = c->vc_alt = nullptr; c->beam_ints = nullptr;
  c->dec_ws = nullptr; c->dec_pos = c->dec_seq = c->dec_kvlen = c->dec_slot = nullptr;
  const size_t vr = (size_t)k.vit_chunk * c->S;
  const size_t pr = (size_t)k.vit_chunk * c->ntok;
  const size_t T = (size_t)k.max_tokens;
  do {
    if ((rc = dalloc(c, &c->v_col, (size_t)k.vit_chunk * c->np * c->Kp))) break;
    if ((rc = dalloc(c, &c->v_x, vr * k.vit_hidden))) break;
    if ((rc = dalloc(c, &c->v_t, vr * k.vit_hidden))) break;
    if ((rc = dalloc(c, &c->v_qkv, vr * 3 * k.vit_hidden))) break;
    if ((rc = dalloc(c, &c->v_ao, vr * k.vit_hidden))) break;
    if ((rc = dalloc(c, &c->v_h, vr * k.vit_inter))) break;
    if ((rc = dalloc(c, &c->v_cu, (size_t)k.vit_chunk + 1))) break;
    if ((rc = dalloc(c, &c->p_t, pr * c->proj_in))) break;
    if ((rc = dalloc(c, &c->p_mid, pr * k.llm_hidden))) break;
    if ((rc = dalloc(c, &c->l_h, T * k.llm_hidden))) break;
    if ((rc = dalloc(c, &c->l_t, T * k.llm_hidden))) break;
    if ((rc = dalloc(c, &c->l_qkv, T * c->qkv_out))) break;
    if ((rc = dalloc(c, &c->l_ao, T * k.llm_hidden))) break;
    if ((rc = dalloc(c, &c->l_ffn, T * k.llm_inter))) break;
    if ((rc = dalloc(c, &c->l_rows, (size_t)(k.max_out_rows + k.max_seqs + 64) * k.llm_hidden))) break;
    if ((rc = dalloc(c, &c->l_pos, T))) break;
    if ((rc = dalloc(c, &c->l_seq, T))) break;
    if ((rc = dalloc(c, &c->l_cu, (size_t)k.max_seqs + 1))) break;
    if ((rc = dalloc(c, &c->l_rowidx, (size_t)k.max_out_rows + k.max_seqs + 64))) break;
    if ((rc = dalloc(c, &c->l_rowidx2, (size_t)k.max_out_rows + k.max_seqs + 64))) break;
    if ((rc = dalloc(c, &c->l_kvlen, (size_t)k.max_seqs))) break;
    if ((rc = dalloc(c, &c->l_packed, (size_t)64))) break;
    if ((rc = dalloc(c, &c->l_trim, (size_t)64 * (2 * k.llm_hidden + k.llm_inter)))) break;
    if ((rc = dalloc(c, &c->l_trim_h, (size_t)(k.max_out_rows + k.max_seqs + 64) * k.llm_hidden))) break;
    if ((rc = dalloc(c, &c->l_neg1, (size_t)k.max_tokens))) break;
    c->rp_vit = RowPlan(); c->rp_llm = RowPlan(); c->cur_rp = nullptr;
    c->rp_vit.cap_halves = (int)(vr / 128) + 2 * k.vit_chunk + 2;
    c->rp_llm.cap_halves = (int)(T / 128) + 2 * k.max_seqs + 2;
    if ((rc = dalloc(c, &c->rp_vit.d_tab, (size_t)2 * c->rp_vit.cap_halves))) break;
    if ((rc = dalloc(c, &c->rp_llm.d_tab, (size_t)2 * c->rp_llm.cap_halves))) break;
    {   // split-K slabs: the planner's cap, or less when no GEMM of this context can reach it (8 slices x most rows x widest N)
      const size_t widest = (size_t)std::max(std::max(std::max(2 * k.llm_inter, c->qkv_out), std::max(k.vit_inter, 3 * k.vit_hidden)), k.llm_hidden);
      const size_t rows = std::max((size_t)k.max_tokens, (size_t)k.vit_chunk * c->S);
      c->splitk_floats = std::min(SPLITK_MAX_FLOATS, (size_t)8 * rows * widest);
      void* p = nullptr;
      if (hipMalloc(&p, c->splitk_floats * sizeof(float)) != hipSuccess) { rc = fail(c, AIGV_ERR_ALLOC, "hipMalloc(split-K scratch) failed"); break; }
      c->ws_allocs.push_back(p);
      c->splitk_ws = (float*)p;
      // the InternViT half (same size: `rows` / `widest` above cover its GEMMs)
      if (hipMalloc(&p, c->splitk_floats * sizeof(float)) != hipSuccess) { rc = fail(c, AIGV_ERR_ALLOC, "hipMalloc(split-K scratch, InternViT) failed"); break; }
      c->ws_allocs.push_back(p);
      c->splitk_ws_vit = (float*)p;
    }
    if (hipMemset(c->l_neg1, 0xFF, (size_t)k.max_tokens * sizeof(int32_t)) != hipSuccess) { rc = fail(c, AIGV_ERR_HIP, "hipMemset failed"); break; }
    {
      int maxd = k.llm_hidden;
      for (int i = 0; i < k.n_score_layers; ++i) maxd = std::max(maxd, (int)k.score_dims[i]);
      if ((rc = dalloc(c, &c->l_score_ws, (size_t)3 * 64 * maxd))) break;
    }
    if (k.kv_capacity > 0) {
      const size_t per = (size_t)k.llm_layers * k.max_seqs * k.llm_kv_heads * k.kv_capacity * c->head_dim;
      if ((rc = dalloc(c, &c->kc, per))) break;
      if ((rc = dalloc(c, &c->vc, per))) break;
      if ((rc = dalloc(c, &c->dec_ws, aigv_attention_decode_ws_floats(k.max_seqs, k.llm_kv_heads, c->g, k.kv_capacity)))) break;
      if ((rc = dalloc(c, &c->dec_pos, (size_t)k.max_seqs))) break;
      if ((rc = dalloc(c, &c->dec_seq, (size_t)k.max_seqs))) break;
      if ((rc = dalloc(c, &c->dec_kvlen, (size_t)k.max_seqs))) break;
      if ((rc = dalloc(c, &c->dec_slot, (size_t)k.max_seqs))) break;
    }
    std::vector<int32_t> cu(k.vit_chunk + 1);
    for (int i = 0; i <= k.vit_chunk; ++i) cu[i] = i * c->S;
    e = hipMemcpy(c->v_cu, cu.data(), cu.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(c, AIGV_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
  } while (0);
  c->ws_phase = false;
  return rc;
}

int aigv_ctx_create(int device, const aigv_config* cfg, aigv_ctx** out) {
  if (!cfg || !out) return fail(nullptr, AIGV_ERR_ARG, "aigv_ctx_create: null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, AIGV_ERR_HIP, "aigv_ctx_create: no HIP device is visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, AIGV_ERR_ARG, "aigv_ctx_create: device %d out of range (%d visible)", device, ndev);
  const aigv_config& k = *cfg;
  // ---- shape constraints of the kernels ----
  if (k.vit_hidden <= 0 || k.vit_heads <= 0 || k.vit_hidden % k.vit_heads) return fail(nullptr, AIGV_ERR_ARG, "bad ViT head split");
  if (k.vit_hidden / k.vit_heads != 64 && k.vit_hidden / k.vit_heads != 128)
    return fail(nullptr, AIGV_ERR_ARG, "ViT head_dim %d not supported (64 or 128)", k.vit_hidden / k.vit_heads);
  if (k.llm_hidden <= 0 || k.llm_heads <= 0 || k.llm_hidden % k.llm_heads || k.llm_hidden / k.llm_heads != 128)
    return fail(nullptr, AIGV_ERR_ARG, "LLM head_dim must be 128");
  if (k.llm_kv_heads <= 0 || k.llm_heads % k.llm_kv_heads) return fail(nullptr, AIGV_ERR_ARG, "bad GQA split");
  if (k.llm_heads / k.llm_kv_heads > 8)   // the decode attention is instantiated for 1..8 query heads per KV head (InternLM2-8B: 4, -20B: 6): say so here, not at the first decode step
    return fail(nullptr, AIGV_ERR_ARG, "GQA groups of more than 8 query heads per KV head are not supported (%d / %d)", k.llm_heads, k.llm_kv_heads);
  if (k.vit_hidden % 128 || k.vit_inter % 128 || k.llm_hidden % 128 || k.llm_inter % 128)
    return fail(nullptr, AIGV_ERR_ARG, "hidden/intermediate sizes must be multiples of 128");
  if (k.image_size % k.patch_size || k.shuffle != 2 || ((k.image_size / k.patch_size) % 2))
    return fail(nullptr, AIGV_ERR_ARG, "image/patch/shuffle combination not supported");
  if (k.motion_dim % 128) return fail(nullptr, AIGV_ERR_ARG, "motion_dim must be a multiple of 128");
  if (k.n_score_layers < 1 || k.n_score_layers > 8) return fail(nullptr, AIGV_ERR_ARG, "n_score_layers out of range");
  if (k.max_frames <= 0 || k.vit_chunk <= 0 || k.max_tokens <= 0 || k.max_seqs <= 0 || k.max_out_rows <= 0)
    return fail(nullptr, AIGV_ERR_ARG, "capacities must be positive");

  aigv_ctx* c = new (std::nothrow) aigv_ctx();
  if (!c) return fail(nullptr, AIGV_ERR_ALLOC, "out of host memory");
  c->cfg = k;
  c->device = device;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { delete c; return fail(nullptr, AIGV_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e)); }
  c->grid = k.image_size / k.patch_size;
  c->np = c->grid * c->grid;
  c->S = c->np + 1;
  c->Kp = roundup(k.num_channels * k.patch_size * k.patch_size, 64);
  c->ntok = c->np / (k.shuffle * k.shuffle);
  c->proj_in = k.vit_hidden * k.shuffle * k.shuffle;
  c->head_dim = k.llm_hidden / k.llm_heads;
  c->vit_head_dim = k.vit_hidden / k.vit_heads;
  c->g = k.llm_heads / k.llm_kv_heads;
  c->qkv_out = (k.llm_heads + 2 * k.llm_kv_heads) * c->head_dim;

  int rc = alloc_workspaces(c);
  if (rc) {
    g_err = c->err;
    aigv_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

void aigv_ctx_destroy(aigv_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (void* p : c->allocs) hipFree(p);
  for (void* p : c->ws_allocs) hipFree(p);
  for (auto& kv : c->w) hipFree(kv.second.p);
  for (auto& r : c->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (auto e : c->ev_pool) hipEventDestroy(e);
  delete c;
}

int aigv_ctx_resize(aigv_ctx* c, const aigv_config* cfg) {
  if (!c || !cfg) return fail(c, AIGV_ERR_ARG, "aigv_ctx_resize: null argument");
  aigv_config a = c->cfg, b = *cfg;   // the model must be the same: compare with the capacity fields levelled
  a.max_frames = b.max_frames; a.vit_chunk = b.vit_chunk; a.max_tokens = b.max_tokens; a.max_seqs = b.max_seqs; a.max_out_rows = b.max_out_rows;
  a.kv_capacity = b.kv_capacity; a.max_positions = b.max_positions;
  if (memcmp(&a, &b, sizeof(aigv_config)) != 0) return fail(c, AIGV_ERR_ARG, "aigv_ctx_resize: only the capacities may change (create a new context for another model)");
  if (b.max_frames <= 0 || b.vit_chunk <= 0 || b.max_tokens <= 0 || b.max_seqs <= 0 || b.max_out_rows <= 0)
    return fail(c, AIGV_ERR_ARG, "capacities must be positive");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  for (void* p : c->ws_allocs) hipFree(p);
  c->ws_allocs.clear();
  const bool had_q8 = c->q8 != nullptr;
  if (had_q8) {   // the e4m3 activation rows are sized by max_tokens too (they live on the weight side: aigv_set_precision made them)
    for (void* p : {(void*)c->q8, (void*)c->q8_scale}) {
      auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
      if (it != c->allocs.end()) c->allocs.erase(it);
      hipFree(p);
    }
    c->q8 = nullptr; c->q8_scale = nullptr;
  }
  const int old_pos = c->cfg.max_positions;
  c->cfg = b;
  c->kv_valid = false;
  int rc = alloc_workspaces(c);
  if (!rc && had_q8) {
    rc = dalloc(c, &c->q8, (size_t)b.max_tokens * (size_t)std::max(b.llm_hidden, b.llm_inter));
    if (!rc) rc = dalloc(c, &c->q8_scale, (size_t)b.max_tokens);
  }
  if (rc) return rc;   // the context is unusable after a failed resize: destroy it
  if (b.max_positions != old_pos) c->finalized = false;   // the rotary tables must be loaded again at the new length, then aigv_finalize_weights
  return 0;
}

int aigv_load_weight(aigv_ctx* c, const char* name, const void* data, const int64_t* shape, int ndim, int dtype,
                     int on_device) {
  if (!c || !name || !data || !shape || ndim <= 0) return fail(c, AIGV_ERR_ARG, "aigv_load_weight: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return fail(c, AIGV_ERR_ARG, "aigv_load_weight(%s): bad shape", name);
    n *= (size_t)shape[i];
  }
  const std::string key(name);
  const aigv_config& k = c->cfg;
  const bool is_patch = key == "vision_model.embeddings.patch_embedding.weight";
  const bool is_w1 = key.find("feed_forward.w1.weight") != std::string::npos;
  const bool is_w3 = key.find("feed_forward.w3.weight") != std::string::npos;
  std::vector<uint16_t> host;
  const void* src = data;           // bf16 source (host or device)
  bool src_dev = on_device != 0;
  if (dtype == AIGV_F32 || is_patch) {
    // stage through the host: convert and/or repack
    std::vector<uint8_t> raw;
    const void* hsrc = data;
    const size_t esz = dtype == AIGV_F32 ? 4 : 2;
    if (on_device) {
      raw.resize(n * esz);
      HIPCHK(c, hipMemcpy(raw.data(), data, n * esz, hipMemcpyDeviceToHost));
      hsrc = raw.data();
    }
    host.resize(n);
    if (dtype == AIGV_F32) for (size_t i = 0; i < n; ++i) host[i] = f32_to_bf16_host(((const float*)hsrc)[i]);
    else memcpy(host.data(), hsrc, n * 2);
    if (is_patch) {  // [Hv, C, P, P] -> [Hv, Kp] zero padded
      const size_t kk = (size_t)k.num_channels * k.patch_size * k.patch_size;
      if (n != (size_t)k.vit_hidden * kk) return fail(c, AIGV_ERR_ARG, "patch_embedding.weight has the wrong size");
      std::vector<uint16_t> padded((size_t)k.vit_hidden * c->Kp, 0);
      for (int r = 0; r < k.vit_hidden; ++r) memcpy(&padded[(size_t)r * c->Kp], &host[(size_t)r * kk], kk * 2);
      host.swap(padded);
      n = host.size();
    }
    src = host.data();
    src_dev = false;
  } else if (dtype != AIGV_BF16) {
    return fail(c, AIGV_ERR_ARG, "aigv_load_weight(%s): unknown dtype %d", name, dtype);
  }

  if (is_w1 || is_w3) {
    // w1 / w3 [I, H] are stored interleaved in 16-row blocks (even block = w1, odd = w3) for the SwiGLU epilogue
    if (ndim != 2 || shape[0] != k.llm_inter || shape[1] != k.llm_hidden)
      return fail(c, AIGV_ERR_ARG, "aigv_load_weight(%s): expected [%d,%d]", name, k.llm_inter, k.llm_hidden);
    std::string fused = key.substr(0, key.find("feed_forward.")) + "feed_forward.w13.weight";
    auto it = c->w.find(fused);
    if (it == c->w.end()) {
      DevBuf b;
      b.bytes = (size_t)2 * k.llm_inter * k.llm_hidden * 2;
      hipError_t e = hipMalloc(&b.p, b.bytes);
      if (e != hipSuccess) return fail(c, AIGV_ERR_ALLOC, "hipMalloc(%zu) for %s: %s", b.bytes, fused.c_str(), hipGetErrorString(e));
      it = c->w.emplace(fused, b).first;
    }
    const size_t blk = (size_t)16 * k.llm_hidden * 2;  // bytes of one 16-row block
    char* dst = (char*)it->second.p + (is_w3 ? blk : 0);
    HIPCHK(c, hipMemcpy2D(dst, 2 * blk, src, blk, blk, k.llm_inter / 16, src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    c->w[key + "#seen"] = DevBuf{};  // marker (no storage)
    c->finalized = false;
    c->llm_lin_dirty = true;
    return 0;
  }

  auto it = c->w.find(key);
  if (it != c->w.end() && it->second.bytes != n * 2) {
    hipFree(it->second.p);
    c->w.erase(it);
    it = c->w.end();
  }
  if (it == c->w.end()) {
    DevBuf b;
    b.bytes = n * 2;
    hipError_t e = hipMalloc(&b.p, b.bytes);
    if (e != hipSuccess) return fail(c, AIGV_ERR_ALLOC, "hipMalloc(%zu) for %s: %s", b.bytes, name, hipGetErrorString(e));
    it = c->w.emplace(key, b).first;
  }
  HIPCHK(c, hipMemcpy(it->second.p, src, n * 2, src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  c->finalized = false;
  if (key.find("language_model.model.layers.") == 0 &&
      (key.find(".attention.wqkv.weight") != std::string::npos || key.find(".attention.wo.weight") != std::string::npos ||
       key.find(".feed_forward.w2.weight") != std::string::npos))
    c->llm_lin_dirty = true;
  return 0;
}

int aigv_finalize_weights(aigv_ctx* c) {
  if (!c) return fail(c, AIGV_ERR_ARG, "null ctx");
  HIPCHK(c, hipSetDevice(c->device));
  // The e4m3 copies follow the bf16 InternLM2 linears: they are dropped (and the context returns to bf16) only when one of those was
  // reloaded since they were made.  Any other reload - rotary tables after a capacity change, a new score head - keeps them and the mode.
  const bool keep_fp8 = !c->llm8.empty() && !c->llm_lin_dirty && c->q8 != nullptr;
  if (!c->llm8.empty() && !keep_fp8) {   // weights were (re)loaded: the e4m3 copies are stale - drop them; aigv_set_precision quantises again
    HIPCHK(c, hipDeviceSynchronize());
    auto drop = [&](void* p) {
      if (!p) return;
      auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
      if (it != c->allocs.end()) c->allocs.erase(it);
      hipFree(p);
    };
    for (auto& q : c->llm8) { drop(q.wqkv); drop(q.wo); drop(q.w13); drop(q.w2); drop(q.s_wqkv); drop(q.s_wo); drop(q.s_w13); drop(q.s_w2); }
    drop(c->q8); drop(c->q8_scale);
    c->q8 = nullptr; c->q8_scale = nullptr;
    c->llm8.clear();
  }
  if (!keep_fp8) c->fp8_llm = false;
  c->llm_lin_dirty = false;
  const aigv_config& k = c->cfg;
  const size_t Hv = k.vit_hidden, Iv = k.vit_inter, H = k.llm_hidden, I = k.llm_inter;
  const std::string e = "vision_model.embeddings.";
  TRY(need(c, e + "patch_embedding.weight", Hv * c->Kp, &c->patch_w));
  TRY(need(c, e + "patch_embedding.bias", Hv, &c->patch_b));
  TRY(need(c, e + "position_embedding", (size_t)c->S * Hv, &c->pos));
  const bf16_t* cls = nullptr;
  TRY(need(c, e + "class_embedding", Hv, &cls));
  {  // class token row = bf16(cls + pos[0])  (modeling_intern_vit.py:100-106)
    std::vector<uint16_t> a(Hv), b(Hv), o(Hv);
    HIPCHK(c, hipMemcpy(a.data(), cls, Hv * 2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(b.data(), c->pos, Hv * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < Hv; ++i) o[i] = f32_to_bf16_host(bf16_to_f32_host(a[i]) + bf16_to_f32_host(b[i]));
    const int64_t shp[1] = {(int64_t)Hv};
    TRY(aigv_load_weight(c, "derived.cls_pos", o.data(), shp, 1, AIGV_BF16, 0));
    TRY(need(c, "derived.cls_pos", Hv, &c->cls_pos));
  }
  c->vit.assign(k.vit_layers, VitLayer{});
  for (int i = 0; i < k.vit_layers; ++i) {
    const std::string p = "vision_model.encoder.layers." + std::to_string(i) + ".";
    VitLayer& L = c->vit[i];
    TRY(need(c, p + "ls1", Hv, &L.ls1));
    TRY(need(c, p + "ls2", Hv, &L.ls2));
    TRY(need(c, p + "attn.qkv.weight", 3 * Hv * Hv, &L.qkv_w));
    if (k.vit_qkv_bias) TRY(need(c, p + "attn.qkv.bias", 3 * Hv, &L.qkv_b));
    if (k.vit_qk_norm) {
      TRY(need(c, p + "attn.q_norm.weight", Hv, &L.qn));
      TRY(need(c, p + "attn.k_norm.weight", Hv, &L.kn));
    }
    TRY(need(c, p + "attn.proj.weight", Hv * Hv, &L.proj_w));
    TRY(need(c, p + "attn.proj.bias", Hv, &L.proj_b));
    TRY(need(c, p + "mlp.fc1.weight", Iv * Hv, &L.fc1_w));
    TRY(need(c, p + "mlp.fc1.bias", Iv, &L.fc1_b));
    TRY(need(c, p + "mlp.fc2.weight", Hv * Iv, &L.fc2_w));
    TRY(need(c, p + "mlp.fc2.bias", Hv, &L.fc2_b));
    TRY(need(c, p + "norm1.weight", Hv, &L.n1w));
    TRY(need(c, p + "norm2.weight", Hv, &L.n2w));
    if (!k.vit_norm_rms) {
      TRY(need(c, p + "norm1.bias", Hv, &L.n1b));
      TRY(need(c, p + "norm2.bias", Hv, &L.n2b));
    }
  }
  TRY(need(c, "language_model.model.tok_embeddings.weight", (size_t)k.vocab * H, &c->tok_emb));
  TRY(need(c, "language_model.model.norm.weight", H, &c->final_norm));
  TRY(need(c, "language_model.output.weight", (size_t)k.vocab * H, &c->lm_head));
  TRY(need(c, "rope.cos", (size_t)k.max_positions * c->head_dim / 2, &c->rope_cos));
  TRY(need(c, "rope.sin", (size_t)k.max_positions * c->head_dim / 2, &c->rope_sin));
  c->llm.assign(k.llm_layers, LlmLayer{});
  for (int i = 0; i < k.llm_layers; ++i) {
    const std::string p = "language_model.model.layers." + std::to_string(i) + ".";
    LlmLayer& L = c->llm[i];
    TRY(need(c, p + "attention.wqkv.weight", (size_t)c->qkv_out * H, &L.wqkv));
    TRY(need(c, p + "attention.wo.weight", H * H, &L.wo));
    if (!c->w.count(p + "feed_forward.w1.weight#seen") || !c->w.count(p + "feed_forward.w3.weight#seen"))
      return fail(c, AIGV_ERR_STATE, "layer %d: feed_forward.w1/w3 were not both loaded", i);
    TRY(need(c, p + "feed_forward.w13.weight", 2 * I * H, &L.w13));
    TRY(need(c, p + "feed_forward.w2.weight", H * I, &L.w2));
    TRY(need(c, p + "attention_norm.weight", H, &L.an));
    TRY(need(c, p + "ffn_norm.weight", H, &L.fn));
  }
  const char* pn[2] = {"mlp1", "motion_mlp"};
  const size_t pin[2] = {(size_t)c->proj_in, (size_t)k.motion_dim};
  for (int j = 0; j < 2; ++j) {
    const std::string p = std::string(pn[j]) + ".";
    TRY(need(c, p + "0.weight", pin[j], &c->p_ln_w[j]));
    TRY(need(c, p + "0.bias", pin[j], &c->p_ln_b[j]));
    TRY(need(c, p + "1.weight", H * pin[j], &c->p_w1[j]));
    TRY(need(c, p + "1.bias", H, &c->p_b1[j]));
    TRY(need(c, p + "3.weight", H * H, &c->p_w2[j]));
    TRY(need(c, p + "3.bias", H, &c->p_b2[j]));
  }
  c->score = ScoreHeadArgs{};
  c->score.n_layers = k.n_score_layers;
  c->score.dims[0] = (int)H;
  for (int j = 0; j < k.n_score_layers; ++j) {
    c->score.dims[j + 1] = k.score_dims[j];
    const std::string p = "mlpscore.fc" + std::to_string(j + 1) + ".";
    TRY(need(c, p + "weight", (size_t)c->score.dims[j + 1] * c->score.dims[j], &c->score.w[j]));
    TRY(need(c, p + "bias", (size_t)c->score.dims[j + 1], &c->score.b[j]));
  }
  c->finalized = true;
  return 0;
}

// ---- InternViT ---------------------------------------------------------------------------------------------
static int vit_norm(aigv_ctx* c, const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows, hipStream_t s) {
  const aigv_config& k = c->cfg;
  if (k.vit_norm_rms) HIPCHK(c, aigv_launch_rmsnorm(x, k.vit_hidden, w, y, k.vit_hidden, rows, k.vit_hidden, k.vit_eps, nullptr, s));
  else HIPCHK(c, aigv_launch_layernorm(x, k.vit_hidden, w, b, y, k.vit_hidden, rows, k.vit_hidden, k.vit_eps, s));
  return 0;
}

int aigv_vit_forward(aigv_ctx* c, const void* frames, int n_frames, void* out_tokens, void* stream) {
  if (!c || !frames || !out_tokens) return fail(c, AIGV_ERR_ARG, "aigv_vit_forward: null argument");
  if (!c->finalized) return fail(c, AIGV_ERR_STATE, "aigv_vit_forward: call aigv_finalize_weights first");
  const aigv_config& k = c->cfg;
  if (n_frames <= 0 || n_frames > k.max_frames) return fail(c, AIGV_ERR_ARG, "n_frames %d outside 1..%d", n_frames, k.max_frames);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  GemmClassScope gcls(c, AIGV_PROF_GEMM_VIT);
  SplitkVitScope svs(c);
  const int Hv = k.vit_hidden, Iv = k.vit_inter;
  int n_layers = k.vit_layers;
  if (k.select_layer != -1) n_layers = k.select_layer < 0 ? k.vit_layers + 1 + k.select_layer : k.select_layer;
  if (n_layers < 0 || n_layers > k.vit_layers) return fail(c, AIGV_ERR_ARG, "select_layer %d out of range", k.select_layer);
  const size_t frame_elems = (size_t)k.num_channels * k.image_size * k.image_size;
  for (int f0 = 0; f0 < n_frames; f0 += k.vit_chunk) {
    const int F = std::min(k.vit_chunk, n_frames - f0);
    const int rows = F * c->S;
    const bf16_t* fr = (const bf16_t*)frames + (size_t)f0 * frame_elems;
    {   // every frame is a sequence of S rows; the table is written by every pass (build_row_plan)
      std::vector<int32_t> cu(F + 1);
      for (int i = 0; i <= F; ++i) cu[i] = i * c->S;
      TRY(build_row_plan(c, c->rp_vit, cu.data(), F, s));
    }
    const RowPlan& rp = c->rp_vit;
    // patch embed: im2col + GEMM with the bias / position / row-remap epilogue (modeling_intern_vit.py:95-107)
    HIPCHK(c, aigv_launch_im2col(fr, F, k.num_channels, k.image_size, k.patch_size, c->Kp, c->v_col, s));
    HIPCHK(c, aigv_launch_cls_rows(c->cls_pos, c->v_x, F, c->S, Hv, s));
    {
      GemmArgs a = gemm_args(c->v_col, c->Kp, c->patch_w, c->Kp, c->v_x, Hv, F * c->np, Hv, c->Kp);
      a.bias = c->patch_b; a.pos = c->pos; a.np = c->np;
      // (every row in full K on ONE tile kernel chosen by the shape of a frame: batch-invariant)
      TRY(launch_one(c, a, EPI_PATCH, Hv % 256 == 0 && resolved_gemm_mode(c) != 1, s));
    }
    for (int li = 0; li < n_layers; ++li) {
      const VitLayer& L = c->vit[li];
      TRY(vit_norm(c, c->v_x, L.n1w, L.n1b, c->v_t, rows, s));
      {
        GemmArgs a = gemm_args(c->v_t, Hv, L.qkv_w, Hv, c->v_qkv, 3 * Hv, rows, 3 * Hv, Hv);
        a.bias = L.qkv_b;
        TRY(run_gemm_rows(c, a, EPI_STORE, rp, s));
      }
      if (k.vit_qk_norm) {  // full-width RMSNorm of q and k, in place (modeling_intern_vit.py:148-151)
        HIPCHK(c, aigv_launch_rmsnorm(c->v_qkv, 3 * Hv, L.qn, c->v_qkv, 3 * Hv, rows, Hv, k.vit_eps, nullptr, s));
        HIPCHK(c, aigv_launch_rmsnorm(c->v_qkv + Hv, 3 * Hv, L.kn, c->v_qkv + Hv, 3 * Hv, rows, Hv, k.vit_eps, nullptr, s));
      }
      {
        AttnArgs a{};
        a.q = c->v_qkv; a.k = c->v_qkv + Hv; a.v = c->v_qkv + 2 * Hv;
        a.ldq = a.ldk = a.ldv = 3 * Hv;
        a.o = c->v_ao; a.ldo = Hv;
        a.cu = c->v_cu; a.n_seq = F; a.max_len = c->S;
        a.n_heads = a.n_kv_heads = k.vit_heads;
        a.q_group_stride = a.kv_head_stride = c->vit_head_dim;
        a.causal = 0; a.post_div = 1.0f; a.q_prescale = 1.0f / sqrtf((float)c->vit_head_dim);
        a.round_scores = c->attn_round_scores; a.waves = tune_of(c).attn_waves; a.lead_key = tune_of(c).lead_key;
        a.uniform_len = 1;
        if (const char* m = aigv_attn_check(a, c->vit_head_dim)) return fail(c, AIGV_ERR_ARG, "%s", m);
        ProfScope ps(c, AIGV_PROF_ATTN_VIT, 4.0 * F * (double)c->S * c->S * Hv, 2.0 * 4 * rows * (double)Hv, s);
        HIPCHK(c, aigv_launch_attention(a, c->vit_head_dim, s));
      }
      {
        GemmArgs a = gemm_args(c->v_ao, Hv, L.proj_w, Hv, c->v_x, Hv, rows, Hv, Hv);
        a.bias = L.proj_b; a.ls = L.ls1; a.resid = c->v_x; a.ldr = Hv;
        TRY(run_gemm_rows(c, a, EPI_LS_RESID, rp, s));
      }
      TRY(vit_norm(c, c->v_x, L.n2w, L.n2b, c->v_t, rows, s));
      {
        GemmArgs a = gemm_args(c->v_t, Hv, L.fc1_w, Hv, c->v_h, Iv, rows, Iv, Hv);
        a.bias = L.fc1_b;
        TRY(run_gemm_rows(c, a, EPI_GELU, rp, s));
      }
      {
        GemmArgs a = gemm_args(c->v_h, Iv, L.fc2_w, Iv, c->v_x, Hv, rows, Hv, Iv);
        a.bias = L.fc2_b; a.ls = L.ls2; a.resid = c->v_x; a.ldr = Hv;
        TRY(run_gemm_rows(c, a, EPI_LS_RESID, rp, s));
      }
    }
    HIPCHK(c, aigv_launch_pixel_shuffle(c->v_x, c->grid, Hv, (bf16_t*)out_tokens + (size_t)f0 * c->ntok * c->proj_in, F, s));
  }
  return 0;
}

int aigv_project(aigv_ctx* c, const void* tokens, int rows, void* out, void* stream) {
  if (!c || !tokens || !out) return fail(c, AIGV_ERR_ARG, "aigv_project: null argument");
  if (!c->finalized) return fail(c, AIGV_ERR_STATE, "aigv_project: call aigv_finalize_weights first");
  if (rows <= 0) return fail(c, AIGV_ERR_ARG, "aigv_project: rows must be positive");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const aigv_config& k = c->cfg;
  const int H = k.llm_hidden, Pin = c->proj_in;
  GemmClassScope gcls(c, AIGV_PROF_GEMM_VIT);
  const int chunk = k.vit_chunk * c->ntok;
  for (int r0 = 0; r0 < rows; r0 += chunk) {
    const int R = std::min(chunk, rows - r0);
    const bf16_t* x = (const bf16_t*)tokens + (size_t)r0 * Pin;
    HIPCHK(c, aigv_launch_layernorm(x, Pin, c->p_ln_w[0], c->p_ln_b[0], c->p_t, Pin, R, Pin, 1e-5f, s));
    {
      GemmArgs a = gemm_args(c->p_t, Pin, c->p_w1[0], Pin, c->p_mid, H, R, H, Pin);
      a.bias = c->p_b1[0];
      TRY(run_gemm_full(c, a, EPI_GELU, s));
    }
    {
      GemmArgs a = gemm_args(c->p_mid, H, c->p_w2[0], H, (bf16_t*)out + (size_t)r0 * H, H, R, H, H);
      a.bias = c->p_b2[0];
      TRY(run_gemm_full(c, a, EPI_STORE, s));
    }
  }
  return 0;
}

int aigv_motion_project(aigv_ctx* c, const void* motion_feature, int n_clips, void* out, void* stream) {
  if (!c || !motion_feature || !out) return fail(c, AIGV_ERR_ARG, "aigv_motion_project: null argument");
  if (!c->finalized) return fail(c, AIGV_ERR_STATE, "aigv_motion_project: call aigv_finalize_weights first");
  if (n_clips <= 0 || n_clips > c->cfg.max_seqs) return fail(c, AIGV_ERR_ARG, "n_clips %d outside 1..%d", n_clips, c->cfg.max_seqs);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const aigv_config& k = c->cfg;
  const int H = k.llm_hidden, Md = k.motion_dim;
  // scratch: p_t (>= ntok*proj_in elements) holds the normalised feature, p_mid the hidden layer
  if ((size_t)n_clips * Md > (size_t)k.vit_chunk * c->ntok * c->proj_in || n_clips > k.vit_chunk * c->ntok)
    return fail(c, AIGV_ERR_STATE, "motion batch does not fit the projector workspace");
  HIPCHK(c, aigv_launch_layernorm((const bf16_t*)motion_feature, Md, c->p_ln_w[1], c->p_ln_b[1], c->p_t, Md, n_clips, Md, 1e-5f, s));
  for (int r0 = 0; r0 < n_clips; r0 += 64) {
    const int R = std::min(64, n_clips - r0);
    TRY(run_skinny(c, c->p_t + (size_t)r0 * Md, Md, R, c->p_w1[1], Md, H, Md, c->p_b1[1], nullptr, 0,
                   c->p_mid + (size_t)r0 * H, H, 3 /*gelu*/, s, 0));
    TRY(run_skinny(c, c->p_mid + (size_t)r0 * H, H, R, c->p_w2[1], H, H, H, c->p_b2[1], nullptr, 0,
                   (bf16_t*)out + (size_t)r0 * H, H, 0 /*store*/, s, 0));
  }
  return 0;
}

// ---- InternLM2 -----------------------------------------------------------------------------------------------
// rows whose final hidden state is consumed: [score rows (one per clip) | logit rows], validated and uploaded to l_rowidx
static int upload_out_rows(aigv_ctx* c, const int32_t* score_rows, bool with_score, int B, const int32_t* logit_rows, int R,
                           int total_rows, hipStream_t s) {
  const aigv_config& k = c->cfg;
  if (R > k.max_out_rows) return fail(c, AIGV_ERR_ARG, "%d logit rows exceed max_out_rows %d", R, k.max_out_rows);
  c->h_rowidx.clear();
  for (int i = 0; i < (with_score ? B : 0); ++i) c->h_rowidx.push_back(score_rows[i]);
  for (int i = 0; i < R; ++i) c->h_rowidx.push_back(logit_rows[i]);
  for (int v : c->h_rowidx)
    if (v < 0 || v >= total_rows) return fail(c, AIGV_ERR_ARG, "output row index %d outside 0..%d", v, total_rows - 1);
  if (!c->h_rowidx.empty()) HIPCHK(c, aigv_launch_write_ints(c->h_rowidx.data(), (int)c->h_rowidx.size(), c->l_rowidx, s));
  return 0;
}

// final RMSNorm + heads on the consumed rows.  compact: `hidden` already holds exactly those rows, in l_rowidx order.
static int final_rows(aigv_ctx* c, float* score, int B, int R, int64_t* argmax, const bf16_t* hidden, bool compact, hipStream_t s) {
  const aigv_config& k = c->cfg;
  const int H = k.llm_hidden;
  const int nS = score ? B : 0;
  const int n = nS + R;
  if (n == 0) return 0;
  // final RMSNorm only on the rows that are consumed (modeling_internlm2.py:984)
  HIPCHK(c, aigv_launch_rmsnorm(hidden, H, c->final_norm, c->l_rows, H, n, H, k.rms_eps, compact ? nullptr : c->l_rowidx, s));
  for (int b0 = 0; b0 < nS; b0 += 64) {
    // NB: the reference's NaN guard looks at the whole batch slice; batches above 64 clips are guarded per 64
    ScoreHeadArgs a = c->score;
    a.x = c->l_rows + (size_t)b0 * H; a.ldx = H; a.B = std::min(64, nS - b0); a.score = score + b0;
    HIPCHK(c, aigv_launch_score_head(a, c->l_score_ws, s));
  }
  for (int r0 = 0; r0 < R; r0 += 64) {
    const int rr = std::min(64, R - r0);
    ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * rr * (double)k.vocab * H, 2.0 * (double)k.vocab * H, s);
    HIPCHK(c, aigv_launch_lm_head_argmax(c->l_rows + (size_t)(nS + r0) * H, rr, H, c->lm_head, k.vocab, c->l_packed,
                                         argmax + r0, nullptr, s));
  }
  return 0;
}

// ---- the two halves of an InternLM2 decoder layer around its attention, shared by aigv_llm_prefill and aigv_llm_extend (rows of
// c->l_h, T of them).  fp8 mode: every linear of the layer in e4m3 except the post-attention half of the LAST layer (wo, w1|w3, w2
// there act on the few consumed rows - weight streaming, nothing for fp8 MFMA to gain - and stay bf16 with or without row trimming).
// attention_norm -> wqkv -> RoPE on K in place (one of g + 2 slots per group; the query heads are rotated by the attention kernel as
// it loads them)
static int llm_layer_qkv(aigv_ctx* c, int li, int T, hipStream_t s) {
  const aigv_config& k = c->cfg;
  const LlmLayer& L = c->llm[li];
  const int H = k.llm_hidden, D = c->head_dim, g = c->g, nkv = k.llm_kv_heads;
  if (c->fp8_llm) {
    HIPCHK(c, aigv_launch_rmsnorm_quant_fp8(c->l_h, H, L.an, c->q8, H, c->q8_scale, T, H, k.rms_eps, s));
    TRY(run_gemm_fp8(c, nullptr, H, H, c->llm8[li].wqkv, c->llm8[li].s_wqkv, c->l_qkv, c->qkv_out, T, c->qkv_out, EPI_STORE, nullptr, 0, s));
  } else {
    HIPCHK(c, aigv_launch_rmsnorm(c->l_h, H, L.an, c->l_t, H, T, H, k.rms_eps, nullptr, s));
    TRY(run_llm_gemm(c, gemm_args(c->l_t, H, L.wqkv, H, c->l_qkv, c->qkv_out, T, c->qkv_out, H), EPI_STORE, s));
  }
  HIPCHK(c, aigv_launch_rope(c->l_qkv, c->qkv_out, c->l_pos, c->rope_cos, c->rope_sin, T, 1, g + 2, nkv, D, s, g));
  return 0;
}

// h += wo(attention output);  h += w2(silu(w1 n) * w3 n), n = ffn_norm(h)
static int llm_layer_post(aigv_ctx* c, int li, int T, hipStream_t s) {
  const aigv_config& k = c->cfg;
  const LlmLayer& L = c->llm[li];
  const int H = k.llm_hidden, I = k.llm_inter;
  if (c->fp8_llm && li != k.llm_layers - 1) {
    const LlmLayerFp8& Q = c->llm8[li];
    TRY(run_gemm_fp8(c, c->l_ao, H, H, Q.wo, Q.s_wo, c->l_h, H, T, H, EPI_RESID, c->l_h, H, s));
    HIPCHK(c, aigv_launch_rmsnorm_quant_fp8(c->l_h, H, L.fn, c->q8, H, c->q8_scale, T, H, k.rms_eps, s));
    TRY(run_gemm_fp8(c, nullptr, H, H, Q.w13, Q.s_w13, c->l_ffn, I, T, 2 * I, EPI_SWIGLU, nullptr, 0, s));
    TRY(run_gemm_fp8(c, c->l_ffn, I, I, Q.w2, Q.s_w2, c->l_h, H, T, H, EPI_RESID, c->l_h, H, s));
    return 0;
  }
  {
    GemmArgs a = gemm_args(c->l_ao, H, L.wo, H, c->l_h, H, T, H, H);
    a.resid = c->l_h; a.ldr = H;
    TRY(run_llm_gemm(c, a, EPI_RESID, s));
  }
  HIPCHK(c, aigv_launch_rmsnorm(c->l_h, H, L.fn, c->l_t, H, T, H, k.rms_eps, nullptr, s));
  TRY(run_llm_gemm(c, gemm_args(c->l_t, H, L.w13, H, c->l_ffn, I, T, 2 * I, H), EPI_SWIGLU, s));
  GemmArgs a = gemm_args(c->l_ffn, I, L.w2, I, c->l_h, H, T, H, I);
  a.resid = c->l_h; a.ldr = H;
  TRY(run_llm_gemm(c, a, EPI_RESID, s));
  return 0;
}

int aigv_llm_prefill(aigv_ctx* c, const int64_t* ids, const int32_t* slot, const int32_t* cu, int B, const void* vis,
                     int n_vis, const void* motion, const int32_t* score_rows, float* score, const int32_t* logit_rows,
                     int R, int64_t* argmax, int keep_kv, void* stream) {
  if (!c || !ids || !slot || !cu) return fail(c, AIGV_ERR_ARG, "aigv_llm_prefill: null argument");
  if (!c->finalized) return fail(c, AIGV_ERR_STATE, "aigv_llm_prefill: call aigv_finalize_weights first");
  const aigv_config& k = c->cfg;
  if (B <= 0 || B > k.max_seqs) return fail(c, AIGV_ERR_ARG, "n_clips %d outside 1..%d", B, k.max_seqs);
  if (cu[0] != 0) return fail(c, AIGV_ERR_ARG, "cu_seqlens[0] must be 0");
  int max_len = 0;
  for (int b = 0; b < B; ++b) {
    const int len = cu[b + 1] - cu[b];
    if (len <= 0) return fail(c, AIGV_ERR_ARG, "clip %d has an empty token sequence", b);
    if (len > k.max_positions) return fail(c, AIGV_ERR_ARG, "clip %d: %d tokens exceed the RoPE table (%d)", b, len, k.max_positions);
    max_len = std::max(max_len, len);
  }
  const int T = cu[B];
  if (T > k.max_tokens) return fail(c, AIGV_ERR_ARG, "%d packed tokens exceed max_tokens %d", T, k.max_tokens);
  if ((score && !score_rows) || (R > 0 && (!logit_rows || !argmax))) return fail(c, AIGV_ERR_ARG, "output rows/buffers inconsistent");
  if (n_vis < 0 || (n_vis > 0 && !vis)) return fail(c, AIGV_ERR_ARG, "visual tokens missing");
  if (keep_kv && (k.kv_capacity <= 0 || max_len >= k.kv_capacity))
    return fail(c, AIGV_ERR_STATE, "keep_kv needs kv_capacity > longest prompt (%d vs %d)", k.kv_capacity, max_len);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const int H = k.llm_hidden, I = k.llm_inter, D = c->head_dim, g = c->g, nkv = k.llm_kv_heads;

  if (B + 1 > AIGV_SMALL_INTS / 2) return fail(c, AIGV_ERR_ARG, "at most %d clips per prefill call", AIGV_SMALL_INTS / 2 - 1);
  // positions / sequence ids / cu_seqlens are produced on the device from cu passed as a kernel argument
  HIPCHK(c, aigv_launch_seqpos(cu, B, c->l_pos, c->l_seq, c->l_cu, T, s));
  // every clip is a sequence: the kernel form of a row follows from its place in its own clip (struct RowPlan)
  TRY(build_row_plan(c, c->rp_llm, cu, B, s));
  RowPlanScope rps(c, &c->rp_llm);   // (fp8 mode: its bf16 GEMMs - the post-attention half of the last layer - follow the per-clip plan too)

  HIPCHK(c, aigv_launch_embed(ids, slot, c->tok_emb, (const bf16_t*)vis, (const bf16_t*)motion, n_vis, c->l_h, T, H, s));
  TRY(upload_out_rows(c, score_rows, score != nullptr, B, logit_rows, R, T, s));
  // Row trimming: after the last layer's attention every row is independent, and only the consumed rows (one score row per
  // clip + the answer rows) are read afterwards (stage2_eval.py:940-941; modeling_internvl_chat.py:469-481).  When they are
  // few, the last layer finishes just those rows: attention for the query blocks that contain them, then wo / MLP on a
  // compact copy through the weight-streaming skinny kernel.  K/V of the last layer still cover every row (keep_kv).
  const int n_out = (int)c->h_rowidx.size();
  int q_tail = 0;
  for (int v : c->h_rowidx) {
    int b = 0;
    while (cu[b + 1] <= v) ++b;
    q_tail = std::max(q_tail, cu[b + 1] - v);
  }
  // The rule is PER CLIP, so that a clip's consumed rows take the same kernels whatever its batch mates are: the consumed rows of a clip
  // with at most TRIM_ROWS of them finish on the weight-streaming kernel (its fixed form), those of a clip with more on the tile kernels
  // with all the other rows.  All clips small (the eval scripts' case): `trim` - the last layer runs for the consumed rows only.  Mixed
  // batch: the last layer runs for every row AND the small clips' consumed rows are finished on the streaming path and put back in place.
  constexpr int TRIM_ROWS = 16;
  std::vector<int32_t> small_rows;
  {
    std::vector<int> per_clip(B, 0), clip_of(c->h_rowidx.size());
    for (size_t i = 0; i < c->h_rowidx.size(); ++i) {
      int b = 0;
      while (cu[b + 1] <= c->h_rowidx[i]) ++b;
      clip_of[i] = b;
      ++per_clip[b];
    }
    for (size_t i = 0; i < c->h_rowidx.size(); ++i)
      if (per_clip[clip_of[i]] <= TRIM_ROWS) small_rows.push_back(c->h_rowidx[i]);
  }
  const bool trim_ok = c->trim_last_layer && n_out > 0 && H % 128 == 0 && I % 128 == 0;
  const bool trim = trim_ok && (int)small_rows.size() == n_out;
  const bool mixed = trim_ok && !trim && !small_rows.empty();
  if (mixed) HIPCHK(c, aigv_launch_write_ints(small_rows.data(), (int)small_rows.size(), c->l_rowidx2, s));
  double attn_flops = 0;
  for (int b = 0; b < B; ++b) { const double L = cu[b + 1] - cu[b]; attn_flops += 4.0 * (L * (L + 1) / 2) * D * k.llm_heads; }
  const size_t kv_layer = (size_t)k.max_seqs * nkv * k.kv_capacity * D;
  for (int li = 0; li < k.llm_layers; ++li) {
    const LlmLayer& L = c->llm[li];
    TRY(llm_layer_qkv(c, li, T, s));
    if (keep_kv)
      HIPCHK(c, aigv_launch_kv_store(c->l_qkv, c->qkv_out, c->l_seq, c->l_pos, c->kc + li * kv_layer, c->vc + li * kv_layer, T,
                                     nkv, g, D, k.kv_capacity, s));
    // a pass that only fills the cache (no output rows) needs nothing of the last layer beyond its K/V
    if (n_out == 0 && c->trim_last_layer && li == k.llm_layers - 1) break;
    {
      AttnArgs a{};
      a.q = c->l_qkv; a.k = c->l_qkv + (size_t)g * D; a.v = c->l_qkv + (size_t)(g + 1) * D;
      a.ldq = a.ldk = a.ldv = c->qkv_out;
      a.o = c->l_ao; a.ldo = H;
      a.cu = c->l_cu; a.n_seq = B; a.max_len = max_len;
      a.n_heads = k.llm_heads; a.n_kv_heads = nkv;
      a.q_group_stride = a.kv_head_stride = (g + 2) * D;
      a.causal = 1; a.post_div = sqrtf((float)D); a.q_prescale = 1.0f;
      a.round_scores = c->attn_round_scores; a.waves = tune_of(c).attn_waves;
      a.rope_pos = c->l_pos; a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin;
      a.rope_pos_is_row = 1;   // l_pos[t] = t - cu[seq] (+ the cached length): aigv_launch_seqpos
      const bool last_trim = trim && li == k.llm_layers - 1;
      a.q_tail = last_trim ? q_tail : 0;
      if (const char* m = aigv_attn_check(a, D)) return fail(c, AIGV_ERR_ARG, "%s", m);
      ProfScope ps(c, AIGV_PROF_ATTN_LLM, last_trim ? 0.0 : attn_flops, 2.0 * T * ((double)c->qkv_out + H), s);
      HIPCHK(c, aigv_launch_attention(a, D, s));
    }
    if ((trim || mixed) && li == k.llm_layers - 1) {
      // 64 rows at a time through the weight-streaming kernel in its fixed 4-slice form (p = 0: the same bits whatever the row count);
      // the finished rows collect in l_trim_h, in l_rowidx (mixed: l_rowidx2) order
      const int32_t* idx = trim ? c->l_rowidx : c->l_rowidx2;
      const int n_fin = trim ? n_out : (int)small_rows.size();
      bf16_t *t_ao = c->l_trim, *t_n = t_ao + (size_t)64 * H, *t_ffn = t_n + (size_t)64 * H, *t_h = c->l_trim_h;
      for (int r0 = 0; r0 < n_fin; r0 += 64) {
        const int nr = std::min(64, n_fin - r0);
        bf16_t* h = t_h + (size_t)r0 * H;
        HIPCHK(c, aigv_launch_gather_rows(c->l_ao, H, idx + r0, nr, t_ao, H, s));
        HIPCHK(c, aigv_launch_gather_rows(c->l_h, H, idx + r0, nr, h, H, s));
        TRY(run_skinny(c, t_ao, H, nr, L.wo, H, H, H, nullptr, h, H, h, H, 1, s, 0));
        HIPCHK(c, aigv_launch_rmsnorm(h, H, L.fn, t_n, H, nr, H, k.rms_eps, nullptr, s));
        TRY(run_skinny(c, t_n, H, nr, L.w13, H, 2 * I, H, nullptr, nullptr, 0, t_ffn, I, 2, s, 0));
        TRY(run_skinny(c, t_ffn, I, nr, L.w2, I, H, I, nullptr, h, H, h, H, 1, s, 0));
      }
      if (trim) {
        TRY(final_rows(c, score, B, R, argmax, t_h, true, s));
        break;
      }
      TRY(llm_layer_post(c, li, T, s));                                                          // every row on the tile kernels ...
      HIPCHK(c, aigv_launch_scatter_rows(t_h, c->l_rowidx2, n_fin, c->l_h, H, H, s));            // ... the small clips' consumed rows put back
      break;
    }
    TRY(llm_layer_post(c, li, T, s));
  }
  if (!trim) TRY(final_rows(c, score, B, R, argmax, c->l_h, false, s));
  if (keep_kv) {
    c->h_kvlen.resize(B);
    c->h_dec.resize((size_t)4 * B);   // pos | seq | visible kv length | slot, uploaded once; advanced on the device
    for (int b = 0; b < B; ++b) {
      const int len = cu[b + 1] - cu[b];
      c->h_kvlen[b] = len;
      c->h_dec[b] = len; c->h_dec[B + b] = b; c->h_dec[2 * B + b] = len + 1; c->h_dec[3 * B + b] = -1;
    }
    HIPCHK(c, aigv_launch_write_ints(c->h_dec.data(), B, c->dec_pos, s));
    HIPCHK(c, aigv_launch_write_ints(c->h_dec.data() + B, B, c->dec_seq, s));
    HIPCHK(c, aigv_launch_write_ints(c->h_dec.data() + 2 * B, B, c->dec_kvlen, s));
    HIPCHK(c, aigv_launch_write_ints(c->h_dec.data() + 3 * B, B, c->dec_slot, s));
    c->kv_seqs = B;
    c->kv_valid = true;
  } else {
    c->kv_valid = false;
  }
  return 0;
}

// Continue the sequences kept by aigv_llm_prefill(keep_kv = 1) with new TEXT tokens: causal attention of the new rows over the
// cached keys plus themselves.  commit = 0 leaves the cache lengths where they were, so several continuations of ONE prefix
// (the four quality-perspective questions behind the same video tokens) can be scored one after the other.
int aigv_llm_extend(aigv_ctx* c, const int64_t* ids, const int32_t* cu, int B, const int32_t* score_rows, float* score,
                    const int32_t* logit_rows, int R, int64_t* argmax, int commit, void* stream) {
  if (!c || !ids || !cu) return fail(c, AIGV_ERR_ARG, "aigv_llm_extend: null argument");
  if (!c->kv_valid) return fail(c, AIGV_ERR_STATE, "aigv_llm_extend: no KV state (run aigv_llm_prefill with keep_kv)");
  const aigv_config& k = c->cfg;
  if (B != c->kv_seqs) return fail(c, AIGV_ERR_ARG, "aigv_llm_extend: %d sequences, the cache holds %d", B, c->kv_seqs);
  if (B + 1 > AIGV_SMALL_INTS / 2) return fail(c, AIGV_ERR_ARG, "at most %d clips per call", AIGV_SMALL_INTS / 2 - 1);
  if (cu[0] != 0) return fail(c, AIGV_ERR_ARG, "cu_seqlens[0] must be 0");
  int max_new = 0;
  for (int b = 0; b < B; ++b) {
    const int n = cu[b + 1] - cu[b];
    if (n <= 0) return fail(c, AIGV_ERR_ARG, "clip %d: no new tokens", b);
    if (c->h_kvlen[b] + n > k.kv_capacity || c->h_kvlen[b] + n > k.max_positions)
      return fail(c, AIGV_ERR_STATE, "clip %d: KV cache / RoPE table exhausted (%d cached + %d new, capacity %d)", b, c->h_kvlen[b], n, k.kv_capacity);
    max_new = std::max(max_new, n);
  }
  const int T = cu[B];
  if (T > k.max_tokens) return fail(c, AIGV_ERR_ARG, "%d new tokens exceed max_tokens %d", T, k.max_tokens);
  if ((score && !score_rows) || (R > 0 && (!logit_rows || !argmax))) return fail(c, AIGV_ERR_ARG, "output rows/buffers inconsistent");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const int H = k.llm_hidden, D = c->head_dim, g = c->g, nkv = k.llm_kv_heads;
  // positions continue after the cached tokens; the cached lengths are the per-sequence key offsets of the attention
  HIPCHK(c, aigv_launch_seqpos(cu, B, c->l_pos, c->l_seq, c->l_cu, T, s, c->h_kvlen.data()));
  HIPCHK(c, aigv_launch_write_ints(c->h_kvlen.data(), B, c->l_kvlen, s));
  HIPCHK(c, aigv_launch_embed(ids, c->l_neg1, c->tok_emb, nullptr, nullptr, 0, c->l_h, T, H, s));
  TRY(upload_out_rows(c, score_rows, score != nullptr, B, logit_rows, R, T, s));
  double attn_flops = 0;
  for (int b = 0; b < B; ++b) { const double n = cu[b + 1] - cu[b]; attn_flops += 4.0 * n * (c->h_kvlen[b] + (n + 1) / 2) * D * k.llm_heads; }
  const size_t kv_layer = (size_t)k.max_seqs * nkv * k.kv_capacity * D;
  for (int li = 0; li < k.llm_layers; ++li) {
    // (fp8 mode: the linears aigv_llm_prefill runs in e4m3 run in e4m3 here too, so a continuation scores like the same tokens inside
    // one prefill of that mode)
    TRY(llm_layer_qkv(c, li, T, s));
    HIPCHK(c, aigv_launch_kv_store(c->l_qkv, c->qkv_out, c->l_seq, c->l_pos, c->kc + li * kv_layer, c->vc + li * kv_layer, T, nkv, g, D,
                                   k.kv_capacity, s));
    {
      AttnArgs a{};
      a.q = c->l_qkv; a.ldq = c->qkv_out;
      a.k = c->kc + li * kv_layer; a.v = c->vc + li * kv_layer;
      a.ldk = a.ldv = D; a.kv_head_stride = k.kv_capacity * D; a.kv_seq_stride = (size_t)nkv * k.kv_capacity * D;
      a.kv_off = c->l_kvlen;
      a.o = c->l_ao; a.ldo = H;
      a.cu = c->l_cu; a.n_seq = B; a.max_len = max_new;
      a.n_heads = k.llm_heads; a.n_kv_heads = nkv;
      a.q_group_stride = (g + 2) * D;
      a.causal = 1; a.post_div = sqrtf((float)D); a.q_prescale = 1.0f;
      a.round_scores = c->attn_round_scores; a.waves = tune_of(c).attn_waves;
      a.rope_pos = c->l_pos; a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin;
      a.rope_pos_is_row = 1;   // l_pos[t] = t - cu[seq] (+ the cached length): aigv_launch_seqpos
      if (const char* m = aigv_attn_check(a, D)) return fail(c, AIGV_ERR_ARG, "%s", m);
      ProfScope ps(c, AIGV_PROF_ATTN_LLM, attn_flops, 2.0 * T * ((double)c->qkv_out + H), s);
      HIPCHK(c, aigv_launch_attention(a, D, s));
    }
    TRY(llm_layer_post(c, li, T, s));
  }
  TRY(final_rows(c, score, B, R, argmax, c->l_h, false, s));
  if (commit) {
    for (int b = 0; b < B; ++b) {
      const int len = c->h_kvlen[b] + (cu[b + 1] - cu[b]);
      c->h_kvlen[b] = len;
      c->h_dec[b] = len; c->h_dec[2 * B + b] = len + 1;
    }
    HIPCHK(c, aigv_launch_write_ints(c->h_dec.data(), B, c->dec_pos, s));
    HIPCHK(c, aigv_launch_write_ints(c->h_dec.data() + 2 * B, B, c->dec_kvlen, s));
  }
  return 0;
}

// Replicate the kept sequences: slots [0, B) -> [B, 2B), ... so that `copies` different continuations of every sequence can
// be extended in ONE pass (the decoder weights are then streamed once for all of them).
int aigv_kv_fork(aigv_ctx* c, int copies, void* stream) {
  if (!c) return fail(c, AIGV_ERR_ARG, "aigv_kv_fork: null context");
  if (!c->kv_valid) return fail(c, AIGV_ERR_STATE, "aigv_kv_fork: no KV state (run aigv_llm_prefill with keep_kv)");
  const aigv_config& k = c->cfg;
  const int B = c->kv_seqs;
  if (copies < 1 || (long)B * copies > k.max_seqs) return fail(c, AIGV_ERR_ARG, "aigv_kv_fork: %d x %d sequences exceed max_seqs %d", B, copies, k.max_seqs);
  if (copies == 1) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t slot = (size_t)k.llm_kv_heads * k.kv_capacity * c->head_dim;      // elements per sequence and layer
  const size_t kv_layer = (size_t)k.max_seqs * slot;
  for (int li = 0; li < k.llm_layers; ++li)
    for (int cpy = 1; cpy < copies; ++cpy) {
      HIPCHK(c, hipMemcpyAsync(c->kc + li * kv_layer + (size_t)cpy * B * slot, c->kc + li * kv_layer, (size_t)B * slot * sizeof(bf16_t), hipMemcpyDeviceToDevice, s));
      HIPCHK(c, hipMemcpyAsync(c->vc + li * kv_layer + (size_t)cpy * B * slot, c->vc + li * kv_layer, (size_t)B * slot * sizeof(bf16_t), hipMemcpyDeviceToDevice, s));
    }
  const int N = B * copies;
  c->h_kvlen.resize(N);
  c->h_dec.resize((size_t)4 * N);
  for (int i = 0; i < N; ++i) {
    const int len = c->h_kvlen[i % B];
    c->h_kvlen[i] = len;
    c->h_dec[i] = len; c->h_dec[N + i] = i; c->h_dec[2 * N + i] = len + 1; c->h_dec[3 * N + i] = -1;
  }
  HIPCHK(c, aigv_launch_write_ints(c->h_dec.data(), N, c->dec_pos, s));
  HIPCHK(c, aigv_launch_write_ints(c->h_dec.data() + N, N, c->dec_seq, s));
  HIPCHK(c, aigv_launch_write_ints(c->h_dec.data() + 2 * N, N, c->dec_kvlen, s));
  HIPCHK(c, aigv_launch_write_ints(c->h_dec.data() + 3 * N, N, c->dec_slot, s));
  c->kv_seqs = N;
  return 0;
}

// Beam search over the kept sequences: sequence i continues from what sequence parent[i] has cached (its first len[i] positions); the
// device-side positions of aigv_decode_step are untouched (all beams of a search have one length).  A gather into a second cache that
// is allocated on first use (workspace: freed and re-made by aigv_ctx_resize), after which the two caches swap.
int aigv_kv_reorder(aigv_ctx* c, const int32_t* parent, const int32_t* len, int n, void* stream) {
  if (!c || !parent || !len) return fail(c, AIGV_ERR_ARG, "aigv_kv_reorder: null argument");
  if (!c->kv_valid) return fail(c, AIGV_ERR_STATE, "aigv_kv_reorder: no KV state (run aigv_llm_prefill with keep_kv)");
  const aigv_config& k = c->cfg;
  if (n != c->kv_seqs) return fail(c, AIGV_ERR_ARG, "aigv_kv_reorder: %d sequences, the cache holds %d", n, c->kv_seqs);
  int max_len = 0;
  for (int i = 0; i < n; ++i) {
    if (parent[i] < 0 || parent[i] >= n) return fail(c, AIGV_ERR_ARG, "aigv_kv_reorder: parent[%d] = %d outside 0..%d", i, parent[i], n - 1);
    if (len[i] <= 0 || len[i] > k.kv_capacity) return fail(c, AIGV_ERR_ARG, "aigv_kv_reorder: len[%d] = %d outside 1..%d", i, len[i], k.kv_capacity);
    max_len = std::max(max_len, len[i]);
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  if (!c->kc_alt || !c->vc_alt || !c->beam_ints) {
    // The second cache is made on first use.  No memset: the gather below writes every position a later step reads, and a
    // blocking null-stream memset would not be ordered against a gather on a non-blocking stream `s`.  All three or none: a partial
    // failure frees what it got, so the next call starts over instead of finding one pointer set.
    const size_t per = (size_t)k.llm_layers * k.max_seqs * k.llm_kv_heads * k.kv_capacity * c->head_dim;
    void *a = nullptr, *b = nullptr, *i = nullptr;
    const hipError_t e1 = hipMalloc(&a, per * sizeof(bf16_t));
    const hipError_t e2 = e1 == hipSuccess ? hipMalloc(&b, per * sizeof(bf16_t)) : e1;
    const hipError_t e3 = e2 == hipSuccess ? hipMalloc(&i, (size_t)2 * k.max_seqs * sizeof(int32_t)) : e2;
    if (e3 != hipSuccess) {
      if (a) hipFree(a);
      if (b) hipFree(b);
      if (i) hipFree(i);
      c->kc_alt = c->vc_alt = nullptr; c->beam_ints = nullptr;
      return fail(c, AIGV_ERR_ALLOC, "aigv_kv_reorder: second KV cache (2 x %zu bytes): %s", per * sizeof(bf16_t), hipGetErrorString(e3));
    }
    c->kc_alt = (bf16_t*)a; c->vc_alt = (bf16_t*)b; c->beam_ints = (int32_t*)i;
    c->ws_allocs.push_back(a); c->ws_allocs.push_back(b); c->ws_allocs.push_back(i);
  }
  HIPCHK(c, aigv_launch_write_ints(parent, n, c->beam_ints, s));
  HIPCHK(c, aigv_launch_write_ints(len, n, c->beam_ints + k.max_seqs, s));
  const size_t kv_layer = (size_t)k.max_seqs * k.llm_kv_heads * k.kv_capacity * c->head_dim;
  hipError_t e = aigv_launch_kv_reorder(c->kc, c->vc, c->kc_alt, c->vc_alt, c->beam_ints, c->beam_ints + k.max_seqs, n, k.llm_layers, k.llm_kv_heads,
                                        k.kv_capacity, c->head_dim, kv_layer, max_len, s);
  if (e != hipSuccess) return fail(c, AIGV_ERR_HIP, "aigv_kv_reorder (n=%d): %s", n, hipGetErrorString(e));
  std::swap(c->kc, c->kc_alt);
  std::swap(c->vc, c->vc_alt);
  return 0;
}

int aigv_set_precision(aigv_ctx* c, int mode) {
  if (!c) return fail(c, AIGV_ERR_ARG, "aigv_set_precision: null context");
  if (mode != AIGV_PRECISION_BF16 && mode != AIGV_PRECISION_FP8_LLM) return fail(c, AIGV_ERR_ARG, "aigv_set_precision: unknown mode %d", mode);
  if (mode == AIGV_PRECISION_BF16) { c->fp8_llm = false; return 0; }
  if (c->llm.empty()) return fail(c, AIGV_ERR_STATE, "aigv_set_precision: call aigv_finalize_weights first");
  const aigv_config& k = c->cfg;
  const int H = k.llm_hidden, I = k.llm_inter, Q = c->qkv_out;
  if (H % 256 || Q % 256 || (2 * I) % 256 || H % 128 || I % 128)
    return fail(c, AIGV_ERR_ARG, "aigv_set_precision: the fp8 kernel needs output widths in multiples of 256 and depths in multiples of 128 (H=%d I=%d qkv=%d)", H, I, Q);
  HIPCHK(c, hipSetDevice(c->device));
  if (c->llm8.empty()) {   // quantise once: one row of W[N, K] = one output channel
    std::vector<LlmLayerFp8> q(k.llm_layers);
    TRY(dalloc(c, &c->q8, (size_t)k.max_tokens * (size_t)std::max(H, I)));
    TRY(dalloc(c, &c->q8_scale, (size_t)k.max_tokens));
    for (int li = 0; li < k.llm_layers; ++li) {
      const LlmLayer& L = c->llm[li];
      struct { const bf16_t* w; int n, kk; uint8_t** q; float** sc; } items[4] = {
          {L.wqkv, Q, H, &q[li].wqkv, &q[li].s_wqkv}, {L.wo, H, H, &q[li].wo, &q[li].s_wo},
          {L.w13, 2 * I, H, &q[li].w13, &q[li].s_w13}, {L.w2, H, I, &q[li].w2, &q[li].s_w2}};
      for (auto& it : items) {
        TRY(dalloc(c, it.q, (size_t)it.n * it.kk));
        TRY(dalloc(c, it.sc, (size_t)it.n));
        hipError_t e = aigv_launch_quant_fp8_rows(it.w, it.kk, it.n, it.kk, *it.q, it.kk, *it.sc, nullptr);
        if (e != hipSuccess) return fail(c, AIGV_ERR_HIP, "weight quantisation failed: %s", hipGetErrorString(e));
      }
    }
    HIPCHK(c, hipDeviceSynchronize());
    c->llm8 = std::move(q);
  }
  c->fp8_llm = true;
  return 0;
}

int aigv_set_row_trimming(aigv_ctx* c, int on) {
  if (!c) return fail(c, AIGV_ERR_ARG, "aigv_set_row_trimming: null context");
  c->trim_last_layer = on != 0;
  return 0;
}

int aigv_ctx_tune(aigv_ctx* c, int knob, int value) {
  if (!c) return fail(c, AIGV_ERR_ARG, "aigv_ctx_tune: null context");
  switch (knob) {
    case AIGV_TUNE_GEMM_MODE: return aigv_set_gemm_mode(c, value);
    case AIGV_TUNE_GEMM256_ORDER:
      if (value < -1 || value > 15) break;
      c->t_order = value; return 0;
    case AIGV_TUNE_GEMM256_VARIANT:
      if (value < -1 || value > 7) break;
      c->t_variant = value; return 0;
    case AIGV_TUNE_ATTN_WAVES:
      if (value != -1 && value != 0 && value != 4 && value != 8) break;
      c->t_attn_waves = value; return 0;
    case AIGV_TUNE_SKINNY_P:
      if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && !(value >= 1000 && value < 1000 + 4096)) break;
      c->t_skinny_p = value; return 0;
    case AIGV_TUNE_BODY_TILE:
      if (value < -1 || value > 2) break;
      c->t_body_tile = value; return 0;
    case AIGV_TUNE_CO_KMAX:
      if (value < -1 || value > 65536 || (value > 0 && value % 64)) break;
      c->t_co_kmax = value; return 0;
    case AIGV_TUNE_TAIL_SLICES:
      if (value < -1 || value > 16) break;
      c->t_tail_slices = value; return 0;
    case AIGV_TUNE_ATTN_LEAD_KEY:
      if (value < -1 || value > 1) break;
      c->t_lead_key = value; return 0;
    case AIGV_TUNE_DECODE_FUSED:
      if (value < -1 || value > 1) break;
      c->t_decode_fused = value; return 0;
    case AIGV_TUNE_DECODE_FP8:
      if (value < -1 || value > 1) break;
      c->t_decode_fp8 = value; return 0;
    case AIGV_TUNE_FUSE_TAILS:
      if (value < -1 || value > 2) break;
      c->t_fuse_tails = value; return 0;
    case AIGV_TUNE_LONE_BODY:
      if (value < -1 || value > 4) break;
      c->t_lone_body = value; return 0;
    case AIGV_TUNE_SKINNY_P8:
      if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4) break;
      c->t_skinny_p8 = value; return 0;
    default: return fail(c, AIGV_ERR_ARG, "aigv_ctx_tune: unknown knob %d", knob);
  }
  return fail(c, AIGV_ERR_ARG, "aigv_ctx_tune: value %d out of range for knob %d", value, knob);
}

int aigv_get_attention_numerics(const aigv_ctx* c) { return c ? c->attn_round_scores : AIGV_ATTENTION_NUMERICS_DEFAULT; }

int aigv_set_attention_numerics(aigv_ctx* c, int mode) {
  if (!c) return fail(c, AIGV_ERR_ARG, "aigv_set_attention_numerics: null context");
  if (mode != 0 && mode != 1) return fail(c, AIGV_ERR_ARG, "aigv_set_attention_numerics: 0 (fp32 scores) or 1 (the reference's bf16 score matrix)");
  c->attn_round_scores = mode;
  return 0;
}

int aigv_set_gemm_mode(aigv_ctx* c, int mode) {
  if (!c) return fail(c, AIGV_ERR_ARG, "aigv_set_gemm_mode: null context");
  if (mode < -1 || mode > 4) return fail(c, AIGV_ERR_ARG, "aigv_set_gemm_mode: mode must be -1 (process default), 0 (row plans), 1 (128 tile), 2 (256 tile), 3 (batch-level dispatch) or 4 (co-resident 256x128 tile)");
  c->gemm_mode = mode;
  return 0;
}

// Which form (P = 1, 2, 4: 16, 8, 4 rows of W per workgroup and slab) each decode GEMV runs in.  A GEMV's rate depends on how
// evenly its workgroups fill the 256 CUs and on how many there are (scripts/gemv_balance_probe.py; in-box sweep of the 8B decode
// step, profiles/r2_decode_forms.txt: wqkv 1 -> 4: -0.16 ms/token, w1|w3 1 -> 2: -0.16, 1 -> 4: -0.24, w2 1 -> 2: -0.08, 1 -> 4: -0.04,
// wo: no change): the finest form the batch allows whose K sub-range per wave still fills one 8-deep load group; long-K GEMVs
// (w2) stop at 8 rows.
static int pick_form(int max_p, int K) {
  int best = 1;
  for (int p = 2; p <= max_p; p *= 2) {
    if (K % (128 * p) || K / (4 * p) < 256 || (K > 8192 && p > 2)) break;
    best = p;
  }
  return best;
}

static void decode_forms(aigv_ctx* c, int B, int* pq, int* po, int* p13, int* p2) {
  const aigv_config& k = c->cfg;
  const int max_p = B <= 4 ? 4 : B <= 8 ? 2 : 1;
  if (resolved_gemm_mode(c) == 1) {   // batch-invariant bits: ONE form whatever the batch (p = 1 here; run_skinny then pins the K split too)
    *pq = *po = *p13 = *p2 = 1;
    return;
  }
  *pq = *po = *p13 = pick_form(max_p, k.llm_hidden);
  *p2 = pick_form(max_p, k.llm_inter);
  // AIGV_TUNE_SKINNY_P (experiments): one value for all four, or - value = 1000 + a packed "wqkv, wo, w1|w3, w2" word with 3 bits each
  // (1 / 2 / 4) - one per GEMV (scripts/decode_bench.py; the sweep of profiles/r2_decode_forms.txt)
  if (const int sp = tune_of(c).skinny_p) {
    if (sp < 1000) *pq = *po = *p13 = *p2 = std::min(sp, max_p);
    else {
      int* out[4] = {pq, po, p13, p2};
      for (int i = 0; i < 4; ++i) {
        const int v = ((sp - 1000) >> (3 * i)) & 7;
        if ((v == 1 || v == 2 || v == 4) && v <= max_p) *out[i] = v;
      }
    }
  }
}

int aigv_decode_step(aigv_ctx* c, const int64_t* ids, int64_t* next, void* stream) {
  if (!c || !ids || !next) return fail(c, AIGV_ERR_ARG, "aigv_decode_step: null argument");
  if (!c->kv_valid) return fail(c, AIGV_ERR_STATE, "aigv_decode_step: no KV state (run aigv_llm_prefill with keep_kv)");
  const aigv_config& k = c->cfg;
  const int B = c->kv_seqs, H = k.llm_hidden, I = k.llm_inter, D = c->head_dim, g = c->g, nkv = k.llm_kv_heads;
  if (B > 64) return fail(c, AIGV_ERR_ARG, "decode supports at most 64 clips per step");
  for (int b = 0; b < B; ++b)
    if (c->h_kvlen[b] + 1 > k.kv_capacity || c->h_kvlen[b] + 1 > k.max_positions)
      return fail(c, AIGV_ERR_STATE, "clip %d: KV cache / RoPE table exhausted at %d tokens", b, c->h_kvlen[b]);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  // positions of the new tokens (= current lengths) and the visible KV lengths live on the device (dec_pos, dec_kvlen)
  // and are advanced by a one-block kernel at the end of the step: no host copies or syncs inside a decode step
  int max_vis = 0;
  for (int b = 0; b < B; ++b) max_vis = std::max(max_vis, c->h_kvlen[b] + 1);
  HIPCHK(c, aigv_launch_embed(ids, c->dec_slot, c->tok_emb, nullptr, nullptr, 0, c->l_h, B, H, s));
  const size_t kv_layer = (size_t)k.max_seqs * nkv * k.kv_capacity * D;
  // Up to 4 sequences: attention_norm / ffn_norm are applied by the GEMV that consumes them (NormArgs in head.hip; same bits), 6
  // launches per layer instead of 8.  AIGV_TUNE_DECODE_FUSED = 0 keeps the separate norm kernels (A/B).
  const Tune tn_ = tune_of(c);
  const bool fused = tn_.decode_fused != 0 && B <= 4 && aigv_skinny_norm_fusable(H);
  // sub-slab forms of the four GEMVs (skinny_kernel's P; head.hip): chosen so that the workgroups come to a whole number per CU.
  // AIGV_TUNE_SKINNY_P / aigv_tune_skinny force one value everywhere it is legal (or one per GEMV: decode_forms).
  int pq, po, p13, p2;
  decode_forms(c, B, &pq, &po, &p13, &p2);
  // fp8 mode: the linears the prefill runs in e4m3 stream their e4m3 copies here too (head8.hip: half the bytes per token; the
  // x rows are normalised and quantised by the GEMV itself) - up to 4 sequences and for the widths the kernel is built for; larger
  // batches decode from the bf16 weights.  AIGV_TUNE_DECODE_FP8 = 0 keeps the bf16 GEMVs (A/B); AIGV_TUNE_SKINNY_P8 = one form for all four.
  const bool f8 = tn_.decode_fp8 != 0 && c->fp8_llm && B <= 4 && aigv_skinny_fp8_supported(H, true) && aigv_skinny_fp8_supported(I, false) && D == 128;
  int q8[4] = {4, 2, 4, 2};
  {
    if (const int v = tn_.skinny_p8)
      for (int i = 0; i < 4; ++i) q8[i] = v;
    if (I / (4 * q8[3]) < 256 || I % (512 * q8[3])) q8[3] = 1;
  }
  for (int li = 0; li < k.llm_layers; ++li) {
    const LlmLayer& L = c->llm[li];
    if (f8) {
      const LlmLayerFp8& Q = c->llm8[li];
      const bool post8 = li != k.llm_layers - 1;   // the post-attention half of the last layer stays bf16, as in the prefill
      const AigvRopeKv rk{c->dec_pos, c->dec_seq, c->rope_cos, c->rope_sin, c->kc + li * kv_layer, c->vc + li * kv_layer, g, nkv, k.kv_capacity};
      {
        ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * B * (double)c->qkv_out * H, (double)c->qkv_out * H, s);
        HIPCHK(c, aigv_launch_skinny_fp8(c->l_h, H, B, Q.wqkv, H, Q.s_wqkv, c->qkv_out, H, nullptr, 0, c->l_qkv, c->qkv_out, 7, &rk, L.an, k.rms_eps, q8[0], s));
      }
      HIPCHK(c, aigv_launch_attention_decode(c->l_qkv, c->qkv_out, (g + 2) * D, c->kc + li * kv_layer, c->vc + li * kv_layer,
                                             c->dec_kvlen, k.kv_capacity, c->l_ao, H, B, nkv, g, D, sqrtf((float)D), max_vis,
                                             c->dec_ws, s));
      if (post8) {
        ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * B * ((double)H * H + 3.0 * H * I), (double)H * H + 3.0 * H * I, s);
        HIPCHK(c, aigv_launch_skinny_fp8(c->l_ao, H, B, Q.wo, H, Q.s_wo, H, H, c->l_h, H, c->l_h, H, 1, nullptr, nullptr, 0.f, q8[1], s));
        HIPCHK(c, aigv_launch_skinny_fp8(c->l_h, H, B, Q.w13, H, Q.s_w13, 2 * I, H, nullptr, 0, c->l_ffn, I, 2, nullptr, L.fn, k.rms_eps, q8[2], s));
        HIPCHK(c, aigv_launch_skinny_fp8(c->l_ffn, I, B, Q.w2, I, Q.s_w2, H, I, c->l_h, H, c->l_h, H, 1, nullptr, nullptr, 0.f, q8[3], s));
      } else {
        TRY(run_skinny(c, c->l_ao, H, B, L.wo, H, H, H, nullptr, c->l_h, H, c->l_h, H, 1, s, po));
        if (fused) {
          ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * B * 2.0 * I * H, 2.0 * 2.0 * I * H, s);
          HIPCHK(c, aigv_launch_skinny_swiglu_normed(c->l_h, H, B, L.w13, H, 2 * I, H, c->l_ffn, I, L.fn, k.rms_eps, s, p13));
        } else {
          HIPCHK(c, aigv_launch_rmsnorm(c->l_h, H, L.fn, c->l_t, H, B, H, k.rms_eps, nullptr, s));
          TRY(run_skinny(c, c->l_t, H, B, L.w13, H, 2 * I, H, nullptr, nullptr, 0, c->l_ffn, I, 2, s, p13));
        }
        TRY(run_skinny(c, c->l_ffn, I, B, L.w2, I, H, I, nullptr, c->l_h, H, c->l_h, H, 1, s, p2));
      }
      continue;
    }
    if (!fused) HIPCHK(c, aigv_launch_rmsnorm(c->l_h, H, L.an, c->l_t, H, B, H, k.rms_eps, nullptr, s));
    {   // wqkv with RoPE + KV-cache append in its epilogue: one launch instead of GEMV + rope / store
      ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * B * (double)c->qkv_out * H, 2.0 * (double)c->qkv_out * H, s);
      HIPCHK(c, aigv_launch_skinny_rope_kv(fused ? c->l_h : c->l_t, H, B, L.wqkv, H, c->qkv_out, H, c->l_qkv, c->qkv_out, c->dec_pos, c->dec_seq, c->rope_cos,
                                           c->rope_sin, c->kc + li * kv_layer, c->vc + li * kv_layer, g, nkv, k.kv_capacity, D, s,
                                           fused ? L.an : nullptr, k.rms_eps, pq));
    }
    HIPCHK(c, aigv_launch_attention_decode(c->l_qkv, c->qkv_out, (g + 2) * D, c->kc + li * kv_layer, c->vc + li * kv_layer,
                                           c->dec_kvlen, k.kv_capacity, c->l_ao, H, B, nkv, g, D, sqrtf((float)D), max_vis,
                                           c->dec_ws, s));
    TRY(run_skinny(c, c->l_ao, H, B, L.wo, H, H, H, nullptr, c->l_h, H, c->l_h, H, 1, s, po));
    if (fused) {
      ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * B * 2.0 * I * H, 2.0 * 2.0 * I * H, s);
      HIPCHK(c, aigv_launch_skinny_swiglu_normed(c->l_h, H, B, L.w13, H, 2 * I, H, c->l_ffn, I, L.fn, k.rms_eps, s, p13));
    } else {
      HIPCHK(c, aigv_launch_rmsnorm(c->l_h, H, L.fn, c->l_t, H, B, H, k.rms_eps, nullptr, s));
      TRY(run_skinny(c, c->l_t, H, B, L.w13, H, 2 * I, H, nullptr, nullptr, 0, c->l_ffn, I, 2, s, p13));
    }
    TRY(run_skinny(c, c->l_ffn, I, B, L.w2, I, H, I, nullptr, c->l_h, H, c->l_h, H, 1, s, p2));
  }
  HIPCHK(c, aigv_launch_rmsnorm(c->l_h, H, c->final_norm, c->l_rows, H, B, H, k.rms_eps, nullptr, s));
  HIPCHK(c, aigv_launch_lm_head_argmax(c->l_rows, B, H, c->lm_head, k.vocab, c->l_packed, next, nullptr, s));
  HIPCHK(c, aigv_launch_advance(c->dec_pos, c->dec_kvlen, B, s));
  for (int b = 0; b < B; ++b) c->h_kvlen[b] += 1;
  return 0;
}

int aigv_out_row_logits(aigv_ctx* c, int first_row, int n_rows, void* logits_bf16, int ldo, void* stream) {
  if (!c || !logits_bf16) return fail(c, AIGV_ERR_ARG, "aigv_out_row_logits: null argument");
  if (!c->finalized) return fail(c, AIGV_ERR_STATE, "aigv_out_row_logits: call aigv_finalize_weights first");
  const aigv_config& k = c->cfg;
  const int cap = k.max_out_rows + k.max_seqs + 64;
  if (first_row < 0 || n_rows <= 0 || first_row + n_rows > cap) return fail(c, AIGV_ERR_ARG, "aigv_out_row_logits: rows %d..%d outside 0..%d", first_row, first_row + n_rows - 1, cap - 1);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  for (int r0 = 0; r0 < n_rows; r0 += 64) {
    const int rr = std::min(64, n_rows - r0);
    ProfScope ps(c, AIGV_PROF_SKINNY, 2.0 * rr * (double)k.vocab * k.llm_hidden, 2.0 * (double)k.vocab * k.llm_hidden, s);
    hipError_t e = aigv_launch_lm_head_logits(c->l_rows + (size_t)(first_row + r0) * k.llm_hidden, rr, k.llm_hidden, c->lm_head, k.vocab,
                                              (bf16_t*)logits_bf16 + (size_t)r0 * ldo, ldo, s);
    if (e != hipSuccess) return fail(c, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP, "lm-head logits (rows=%d ldo=%d): %s", rr, ldo, hipGetErrorString(e));
  }
  return 0;
}

int aigv_out_row_hidden(aigv_ctx* c, int first_row, int n_rows, void* hidden_bf16, int ldo, void* stream) {
  if (!c || !hidden_bf16) return fail(c, AIGV_ERR_ARG, "aigv_out_row_hidden: null argument");
  if (!c->finalized) return fail(c, AIGV_ERR_STATE, "aigv_out_row_hidden: call aigv_finalize_weights first");
  const aigv_config& k = c->cfg;
  const int cap = k.max_out_rows + k.max_seqs + 64;
  if (first_row < 0 || n_rows <= 0 || first_row + n_rows > cap || ldo < k.llm_hidden)
    return fail(c, AIGV_ERR_ARG, "aigv_out_row_hidden: rows %d..%d outside 0..%d or ldo %d < %d", first_row, first_row + n_rows - 1, cap - 1, ldo, k.llm_hidden);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy2DAsync(hidden_bf16, (size_t)ldo * sizeof(bf16_t), c->l_rows + (size_t)first_row * k.llm_hidden, (size_t)k.llm_hidden * sizeof(bf16_t),
                             (size_t)k.llm_hidden * sizeof(bf16_t), n_rows, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int aigv_decode_eos(aigv_ctx* c, int64_t* tokens, int32_t* state, const int64_t* eos_ids, int n_eos, int64_t pad_id, void* stream) {
  if (!c || !tokens || !state) return fail(c, AIGV_ERR_ARG, "aigv_decode_eos: null argument");
  if (!c->kv_valid) return fail(c, AIGV_ERR_STATE, "aigv_decode_eos: no KV state (run aigv_llm_prefill with keep_kv)");
  if (n_eos < 0 || n_eos > 8 || (n_eos > 0 && !eos_ids)) return fail(c, AIGV_ERR_ARG, "aigv_decode_eos: 0..8 end-of-sequence ids");
  if (c->kv_seqs > 64) return fail(c, AIGV_ERR_ARG, "aigv_decode_eos: at most 64 sequences");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, aigv_launch_decode_eos(tokens, state, c->kv_seqs, eos_ids, n_eos, pad_id, (hipStream_t)stream));
  return 0;
}

// ---- single operators ----------------------------------------------------------------------------------------
int aigv_op_gemm(const void* A, int lda, const void* W_, int ldw, void* C, int ldc, const void* bias, const void* ls,
                 const void* resid, int ldr, const void* pos, int np, int M, int N, int K, int epi, void* stream) {
  GemmArgs a = gemm_args((const bf16_t*)A, lda, (const bf16_t*)W_, ldw, (bf16_t*)C, ldc, M, N, K);
  a.bias = (const bf16_t*)bias; a.ls = (const bf16_t*)ls; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  a.pos = (const bf16_t*)pos; a.np = np;
  return run_gemm(nullptr, a, epi, (hipStream_t)stream);
}

// aigv_op_gemm with the rows divided into independent sequences (cu_host[0..n_seq], cu[0] = 0, cu[n_seq] = M): the dispatch the scoring
// pass uses (struct RowPlan).  Test entry point: allocates the plan's table per call and synchronises the stream before freeing it.
int aigv_op_gemm_rows(const void* A, int lda, const void* W_, int ldw, void* C, int ldc, const void* bias, const void* ls,
                      const void* resid, int ldr, const int32_t* cu_host, int n_seq, int N, int K, int epi, void* stream) {
  if (!cu_host || n_seq < 1 || cu_host[0] != 0) return fail(nullptr, AIGV_ERR_ARG, "aigv_op_gemm_rows: bad cu_seqlens");
  for (int b = 0; b < n_seq; ++b)
    if (cu_host[b + 1] <= cu_host[b]) return fail(nullptr, AIGV_ERR_ARG, "aigv_op_gemm_rows: empty sequence %d", b);
  if (epi == EPI_PATCH) return fail(nullptr, AIGV_ERR_ARG, "aigv_op_gemm_rows: no patch epilogue");
  const int M = cu_host[n_seq];
  GemmArgs a = gemm_args((const bf16_t*)A, lda, (const bf16_t*)W_, ldw, (bf16_t*)C, ldc, M, N, K);
  a.bias = (const bf16_t*)bias; a.ls = (const bf16_t*)ls; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  RowPlan rp;
  rp.cap_halves = M / 128 + 2 * n_seq + 2;
  HIPCHK(nullptr, hipMalloc((void**)&rp.d_tab, (size_t)2 * rp.cap_halves * sizeof(int32_t)));
  int rc = build_row_plan(nullptr, rp, cu_host, n_seq, (hipStream_t)stream);
  if (!rc) rc = run_gemm_rows(nullptr, a, epi, rp, (hipStream_t)stream);
  hipStreamSynchronize((hipStream_t)stream);
  hipFree(rp.d_tab);
  return rc;
}

int aigv_op_gemm_splitk(const void* A, int lda, const void* W_, int ldw, void* C, int ldc, const void* bias, const void* ls,
                        const void* resid, int ldr, int M, int N, int K, int epi, int k_slices, void* ws_f32, void* stream) {
  GemmArgs a = gemm_args((const bf16_t*)A, lda, (const bf16_t*)W_, ldw, (bf16_t*)C, ldc, M, N, K);
  a.bias = (const bf16_t*)bias; a.ls = (const bf16_t*)ls; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  if (const char* m = aigv_gemm_check(a, epi)) return fail(nullptr, AIGV_ERR_ARG, "%s", m);
  hipError_t e = aigv_launch_gemm_splitk(a, epi, k_slices, (float*)ws_f32, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP, "split-K gemm: %s", hipGetErrorString(e));
  return 0;
}

int aigv_op_gemm_splitk256(const void* A, int lda, const void* W_, int ldw, void* C, int ldc, const void* bias, const void* ls,
                           const void* resid, int ldr, int M, int N, int K, int epi, int k_slices, void* ws_f32, void* stream) {
  GemmArgs a = gemm_args((const bf16_t*)A, lda, (const bf16_t*)W_, ldw, (bf16_t*)C, ldc, M, N, K);
  a.bias = (const bf16_t*)bias; a.ls = (const bf16_t*)ls; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  if (const char* m = aigv_gemm_check(a, epi)) return fail(nullptr, AIGV_ERR_ARG, "%s", m);
  hipError_t e = aigv_launch_gemm_splitk(a, epi, k_slices, (float*)ws_f32, (hipStream_t)stream, true);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP, "split-K gemm (256 tile): %s", hipGetErrorString(e));
  return 0;
}

int aigv_op_quant_fp8_rows(const void* x_bf16, int ldx, int rows, int K, void* q_e4m3, int ldq, float* row_scale, void* stream) {
  hipError_t e = aigv_launch_quant_fp8_rows((const bf16_t*)x_bf16, ldx, rows, K, (uint8_t*)q_e4m3, ldq, row_scale, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP, "fp8 row quantisation (rows=%d K=%d): %s", rows, K, hipGetErrorString(e));
  return 0;
}

int aigv_op_gemm_fp8(const void* A_e4m3, int lda, const void* W_e4m3, int ldw, void* C, int ldc, const float* row_scale,
                     const float* col_scale, const void* bias, const void* ls, const void* resid, int ldr, int M, int N, int K, int epi,
                     int k_slices, void* ws_f32, void* stream) {
  GemmArgs a{};
  a.A = (const bf16_t*)A_e4m3; a.lda = lda; a.W = (const bf16_t*)W_e4m3; a.ldw = ldw; a.C = (bf16_t*)C; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K; a.bias = (const bf16_t*)bias; a.row_scale = row_scale; a.col_scale = col_scale;
  a.ls = (const bf16_t*)ls; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  const int n_out = epi == EPI_SWIGLU ? N / 2 : N;
  if (ldc < n_out || (ldc % 8) || lda < K || ldw < K)
    return fail(nullptr, AIGV_ERR_ARG, "aigv_op_gemm_fp8: bad leading dimension (M=%d N=%d K=%d)", M, N, K);
  hipError_t e = k_slices > 1 ? aigv_launch_gemm_splitk_fp8(a, epi, k_slices, (float*)ws_f32, (hipStream_t)stream)
                              : aigv_launch_gemm256_fp8(a, epi, (hipStream_t)stream);
  if (e != hipSuccess)
    return fail(nullptr, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP,
                "fp8 gemm (M=%d N=%d K=%d epi=%d; needs N %% 256 == 0, K %% 128 == 0, 16-byte row strides, both scale vectors, epi in "
                "{store, gelu, ls_resid, resid, swiglu}): %s", M, N, K, epi, hipGetErrorString(e));
  return 0;
}

int aigv_op_skinny_gemm(const void* x, int ldx, int R, const void* W_, int ldw, int N, int K, const void* bias,
                        const void* resid, int ldr, void* out, int ldo, int epi, void* stream) {
  return run_skinny(nullptr, (const bf16_t*)x, ldx, R, (const bf16_t*)W_, ldw, N, K, (const bf16_t*)bias,
                    (const bf16_t*)resid, ldr, (bf16_t*)out, ldo, epi, (hipStream_t)stream, g_tune.skinny_p ? g_tune.skinny_p : 1);
}

int aigv_op_skinny_gemm_fp8(const void* x, int ldx, int R, const void* W_e4m3, int ldw, const float* w_scale, int N, int K, const void* resid,
                            int ldr, void* out, int ldo, int epi, const void* norm_w, float eps, int p, void* stream) {
  if (epi != 1 && epi != 2) return fail(nullptr, AIGV_ERR_ARG, "aigv_op_skinny_gemm_fp8: epi must be 1 (residual) or 2 (swiglu)");
  hipError_t e = aigv_launch_skinny_fp8((const bf16_t*)x, ldx, R, (const uint8_t*)W_e4m3, ldw, w_scale, N, K, (const bf16_t*)resid, ldr, (bf16_t*)out, ldo,
                                        epi, nullptr, (const bf16_t*)norm_w, eps, p, (hipStream_t)stream);
  if (e != hipSuccess)
    return fail(nullptr, e == hipErrorInvalidValue ? AIGV_ERR_ARG : AIGV_ERR_HIP, "fp8 skinny gemm (R=%d N=%d K=%d epi=%d p=%d): %s", R, N, K, epi, p, hipGetErrorString(e));
  return 0;
}

int aigv_op_layernorm(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, int rows, int H, float eps,
                      void* stream) {
  HIPCHK(nullptr, aigv_launch_layernorm((const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, ldy, rows, H,
                                        eps, (hipStream_t)stream));
  return 0;
}

int aigv_op_rmsnorm(const void* x, int ldx, const void* w, void* y, int ldy, int rows, int H, float eps,
                    const int32_t* row_idx, void* stream) {
  HIPCHK(nullptr, aigv_launch_rmsnorm((const bf16_t*)x, ldx, (const bf16_t*)w, (bf16_t*)y, ldy, rows, H, eps, row_idx,
                                      (hipStream_t)stream));
  return 0;
}

int aigv_op_rope(void* qkv, int ld, const int32_t* pos, const void* cos, const void* sin, int tokens, int n_rot, int slots,
                 int n_groups, int head_dim, void* stream) {
  HIPCHK(nullptr, aigv_launch_rope((bf16_t*)qkv, ld, pos, (const bf16_t*)cos, (const bf16_t*)sin, tokens, n_rot, slots,
                                   n_groups, head_dim, (hipStream_t)stream));
  return 0;
}

int aigv_op_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                      const int32_t* cu, int n_seq, int max_len, int n_heads, int n_kv_heads, int q_group_stride,
                      int kv_head_stride, int head_dim, int causal, float post_div, float q_prescale, void* stream) {
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.v = (const bf16_t*)v; a.ldv = ldv;
  a.o = (bf16_t*)o; a.ldo = ldo; a.cu = cu; a.n_seq = n_seq; a.max_len = max_len; a.n_heads = n_heads;
  a.n_kv_heads = n_kv_heads; a.q_group_stride = q_group_stride; a.kv_head_stride = kv_head_stride;
  a.causal = causal & 1; a.uniform_len = (causal >> 1) & 1; a.post_div = post_div; a.q_prescale = q_prescale;
  a.round_scores = (causal >> 2) & 1; a.lead_key = (causal >> 3) & 1; a.waves = g_tune.attn_waves;
  if (const char* m = aigv_attn_check(a, head_dim)) return fail(nullptr, AIGV_ERR_ARG, "%s", m);
  HIPCHK(nullptr, aigv_launch_attention(a, head_dim, (hipStream_t)stream));
  return 0;
}

int aigv_op_attention_rope(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                           const int32_t* cu, int n_seq, int max_len, int n_heads, int n_kv_heads, int q_group_stride,
                           int kv_head_stride, int head_dim, int causal, float post_div, float q_prescale, const int32_t* pos,
                           const void* cos, const void* sin, void* stream) {
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.v = (const bf16_t*)v; a.ldv = ldv;
  a.o = (bf16_t*)o; a.ldo = ldo; a.cu = cu; a.n_seq = n_seq; a.max_len = max_len; a.n_heads = n_heads;
  a.n_kv_heads = n_kv_heads; a.q_group_stride = q_group_stride; a.kv_head_stride = kv_head_stride;
  a.causal = causal & 1; a.uniform_len = (causal >> 1) & 1; a.post_div = post_div; a.q_prescale = q_prescale;
  a.round_scores = (causal >> 2) & 1; a.lead_key = (causal >> 3) & 1; a.waves = g_tune.attn_waves;
  a.rope_pos = pos; a.rope_cos = (const bf16_t*)cos; a.rope_sin = (const bf16_t*)sin;
  if (const char* m = aigv_attn_check(a, head_dim)) return fail(nullptr, AIGV_ERR_ARG, "%s", m);
  HIPCHK(nullptr, aigv_launch_attention(a, head_dim, (hipStream_t)stream));
  return 0;
}

int aigv_op_pixel_shuffle(const void* vit_out, int grid, int vit_hidden, void* out, int n_frames, void* stream) {
  HIPCHK(nullptr, aigv_launch_pixel_shuffle((const bf16_t*)vit_out, grid, vit_hidden, (bf16_t*)out, n_frames, (hipStream_t)stream));
  return 0;
}

int aigv_op_im2col(const void* frames, int n_frames, int channels, int image_size, int patch, int kp, void* out,
                   void* stream) {
  HIPCHK(nullptr, aigv_launch_im2col((const bf16_t*)frames, n_frames, channels, image_size, patch, kp, (bf16_t*)out,
                                     (hipStream_t)stream));
  return 0;
}

int aigv_op_lm_head_argmax(const void* h, int rows, int hidden, const void* W_, int vocab, void* scratch_u64, int64_t* idx,
                           float* val, void* stream) {
  HIPCHK(nullptr, aigv_launch_lm_head_argmax((const bf16_t*)h, rows, hidden, (const bf16_t*)W_, vocab,
                                             (unsigned long long*)scratch_u64, idx, val, (hipStream_t)stream));
  return 0;
}

int aigv_op_frame_ingest(const void* hwc_u8, int n_frames, int height, int width, const float* mean, const float* stdv,
                         void* out_nchw, void* stream) {
  HIPCHK(nullptr, aigv_launch_frame_ingest((const uint8_t*)hwc_u8, n_frames, height, width, mean, stdv, (bf16_t*)out_nchw,
                                           (hipStream_t)stream));
  return 0;
}

int aigv_op_frame_resize_ingest(const void* hwc_u8, int n_frames, int in_h, int in_w, int out_h, int out_w, const float* mean,
                                const float* stdv, void* tmp_u8, void* out_u8_hwc, void* out_nchw, void* stream) {
  if (!hwc_u8 || !tmp_u8 || (!out_u8_hwc && !out_nchw) || n_frames < 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 ||
      (out_nchw && (!mean || !stdv)))
    return fail(nullptr, AIGV_ERR_ARG, "aigv_op_frame_resize_ingest: bad argument (frames %d, %dx%d -> %dx%d)", n_frames, in_h, in_w, out_h, out_w);
  if (in_h > 16384 || in_w > 16384 || out_h > 16384 || out_w > 16384)
    return fail(nullptr, AIGV_ERR_ARG, "aigv_op_frame_resize_ingest: sizes above 16384 are not supported");
  // Pillow (12.2, observed against the live package: tests/manual/fuzz_resize.py) runs the VERTICAL pass first for frames more than 100 times taller than wide
  // that shrink vertically - the intermediate uint8 image, and so the result, differs from the horizontal-first order implemented here.  Not a video
  // geometry: refused rather than answered differently from Pillow.
  if (in_w != out_w && in_h != out_h && in_h > out_h && (long)in_h > 100L * in_w)
    return fail(nullptr, AIGV_ERR_ARG, "aigv_op_frame_resize_ingest: %dx%d frames (more than 100 times taller than wide) are not supported: Pillow orders its passes differently there", in_h, in_w);
  HIPCHK(nullptr, aigv_launch_frame_resize_ingest((const uint8_t*)hwc_u8, n_frames, in_h, in_w, out_h, out_w, mean, stdv,
                                                  (uint8_t*)tmp_u8, (uint8_t*)out_u8_hwc, (bf16_t*)out_nchw, (hipStream_t)stream));
  return 0;
}

// ---- measurement -----------------------------------------------------------------------------------------------
int aigv_plan_gemm(int M, int N, int K, int epi, int* plan, double* est_us) {
  if (!plan || M <= 0 || N <= 0 || K <= 0 || N % 128 || K % 64 || epi < 0 || epi >= EPI_COUNT)
    return fail(nullptr, AIGV_ERR_ARG, "aigv_plan_gemm: bad problem M=%d N=%d K=%d epi=%d", M, N, K, epi);
  const int pmode = g_tune.gemm_mode == 3 ? 0 : g_tune.gemm_mode;
  const int right = split_columns(M, N, K, epi, pmode);
  const GemmPlan pl = plan_gemm(M, N - right, K, epi, pmode);
  plan[6] = right;
  plan[0] = pl.top_tiles; plan[1] = pl.mid_tiles; plan[2] = pl.mid_slices; plan[3] = pl.last_rows; plan[4] = pl.last_kind;
  plan[5] = pl.last_slices;
  if (est_us) *est_us = pl.est_us + (right ? t128(M, right, K / 64) + LAUNCH_GAP : 0.0);
  return 0;
}

int aigv_tune_skinny(int p) {
  if (p != 0 && p != 1 && p != 2 && p != 4) return fail(nullptr, AIGV_ERR_ARG, "aigv_tune_skinny: 0 (default), 1, 2 or 4, got %d", p);
  g_tune.skinny_p = p;
  return 0;
}

int aigv_tune_attention(int waves) {
  if (waves != 0 && waves != 4 && waves != 8)
    return fail(nullptr, AIGV_ERR_ARG, "aigv_tune_attention: 0 (default), 4 or 8 waves per workgroup, got %d", waves);
  g_tune.attn_waves = waves;
  return 0;
}

int aigv_tune_gemm(int mode, double rate256) {
  // mode = kernel choice (0 auto, 1 128-tile, 2 256-tile) + 16 * (256-kernel schedule variant 0..3, experiments)
  // mode bits 4..6: 0 = keep the default schedule, 1 + v = select 256-kernel schedule variant v (0..3)
  // bits 10..13: tile order of the 256 kernel for every shape (default 0: by weight size, gemm256.hip): 1 = row groups, 1 + g = groups of g column tiles
  // bits 14..15: tile kernel of a row plan's body: 0 by fill (default), 1 = 256 tiles, 2 = 128 tiles
  g_tune.body_tile = (mode >> 14) & 3;
  g_tune.order_sel = (mode >> 10) & 15;
  mode &= 1023;
  const int vsel = mode >> 4;
  mode &= 15;
  if (mode < 0 || mode > 4 || vsel < 0 || vsel > 7)
    return fail(nullptr, AIGV_ERR_ARG, "aigv_tune_gemm: mode must be 0 (auto), 1 (128 tile), 2 (256 tile), 3 (batch-level dispatch in the scoring pass) or 4 (co-resident 256x128 tile)");
  if (vsel > 0) g_tune.variant_sel = vsel;
  g_tune.gemm_mode = mode;
  if (rate256 > 0) g_rate256 = rate256;
  return 0;
}

// process default of a context-only knob (the context-free aigv_op_* entry points and contexts that left it at -1 follow it): tests and A/B scripts
int aigv_tune_default(int knob, int value) {
  switch (knob) {
    case AIGV_TUNE_TAIL_SLICES: if (value < 0 || value > 16) break; g_tune.tail_slices = value; return 0;
    case AIGV_TUNE_FUSE_TAILS: if (value < 0 || value > 2) break; g_tune.fuse_tails = value; return 0;
    case AIGV_TUNE_LONE_BODY: if (value < 0 || value > 4) break; g_tune.lone_body = value; return 0;   // (same range as aigv_ctx_tune)
    case AIGV_TUNE_ATTN_LEAD_KEY: if (value < 0 || value > 1) break; g_tune.lead_key = value; return 0;
    case AIGV_TUNE_CO_KMAX: return aigv_tune_co_gemm(value);
    default: return fail(nullptr, AIGV_ERR_ARG, "aigv_tune_default: knob %d has no process default here (aigv_tune_gemm / _attention / _skinny set the others)", knob);
  }
  return fail(nullptr, AIGV_ERR_ARG, "aigv_tune_default: value %d out of range for knob %d", value, knob);
}

int aigv_tune_co_gemm(int kmax) {
  if (kmax < 0 || kmax > 65536 || kmax % 64) return fail(nullptr, AIGV_ERR_ARG, "aigv_tune_co_gemm: K threshold must be 0 (off) or a multiple of 64");
  g_tune.co_kmax = kmax;
  return 0;
}

int aigv_prof_enable(aigv_ctx* c, int on) {
  if (!c) return fail(c, AIGV_ERR_ARG, "null ctx");
  c->prof = on != 0;
  return 0;
}

int aigv_prof_read(aigv_ctx* c, int cls, int64_t* launches, double* total_ms, double* flops, double* bytes) {
  if (!c || cls < 0 || cls >= AIGV_PROF_COUNT) return fail(c, AIGV_ERR_ARG, "aigv_prof_read: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  int64_t n = 0;
  double ms = 0, fl = 0, by = 0;
  std::vector<ProfRec> keep;
  for (auto& r : c->recs) {
    if (r.cls != cls) { keep.push_back(r); continue; }
    HIPCHK(c, hipEventSynchronize(r.b));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, r.a, r.b));
    ms += t; fl += r.flops; by += r.bytes; ++n;
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->recs.swap(keep);
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  return 0;
}

}  // extern "C"
