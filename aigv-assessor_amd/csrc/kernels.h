// Internal launcher declarations (C++ side of the library; the public C ABI is include/aigv_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

// ---- GEMM ---------------------------------------------------------------------------------------
enum GemmEpilogue {
  EPI_STORE = 0,     // C = bf16(acc + bias?)
  EPI_GELU = 1,      // C = bf16(gelu(bf16(acc + bias?)))   erf GELU, common.h gelu_fast
  EPI_LS_RESID = 2,  // C = bf16(resid + bf16(bf16(acc + bias?) * ls))
  EPI_RESID = 3,     // C = bf16(resid + bf16(acc + bias?))
  EPI_SWIGLU = 4,    // W = 16-row interleave of (w1, w3); C[M, N/2] = bf16(bf16(silu(bf16(g))) * bf16(u))
  EPI_PATCH = 5,     // patch-embed: C[m + m/np + 1] = bf16(bf16(acc + bias) + pos[m % np + 1])
  EPI_COUNT = 6,
  EPI_PARTIAL = 6    // internal (split-K tails): fp32 partial sums of K slice blockIdx.y -> part[slice][M][N]
};

struct GemmArgs {
  const bf16_t* A; int lda;     // activations [M, K], row stride lda
  const bf16_t* W; int ldw;     // weights [N, K] (nn.Linear layout), row stride ldw
  bf16_t* C; int ldc;           // output, row stride ldc
  const bf16_t* bias;           // [N] or null
  const bf16_t* ls;             // [N] layer-scale (EPI_LS_RESID)
  const bf16_t* resid; int ldr; // residual [M, N]
  const bf16_t* pos;            // [np+1, N] position table (EPI_PATCH)
  int M, N, K;
  int np;                       // patches per frame (EPI_PATCH)
  float* part;                  // EPI_PARTIAL: fp32 slabs [k_slices][M][N]
  int k_slices;                 // EPI_PARTIAL: grid.y; each slice covers K / k_slices (a multiple of 64)
  // fp8 operands (aigv_launch_gemm256_fp8): A / W hold e4m3 bytes (K, lda, ldw counted in PAIRS of bytes so that the rows are the
  // same 128-byte K-tiles), C = bf16(acc * row_scale[m] * col_scale[n] + bias)
  const float* row_scale;
  const float* col_scale;
  // gemm256 only: the rows of a launch as a table of 128-row HALF tiles, (base row, valid rows 0..128) int32 pairs on the device; row
  // tile i = halves 2i, 2i+1.  M is then only the row count of the buffers (slabs of the split-K form are [k_slices][tiles*256][N]).
  const int32_t* row_tab; int tab_halves;
  // fused body + tail-slice launch (aigv_launch_gemm256_fused): the table continues with fuse_tail_halves tail halves; fuse_body_wg is filled
  // in by the launcher
  int fuse_tail_halves, fuse_body_wg;
  // per-launch tuning selectors (0 = the default; set from the context's / the process's knobs by the dispatcher in api.hip - the kernel files
  // hold no mutable state): order_sel 1 = row groups, 1 + g = groups of g column tiles (0: by weight size); variant_sel 1 + v = schedule
  // variant v of the 256 kernel (0: the shipped one)
  int order_sel, variant_sel;
  int order;                    // gemm256 tile order: 0 = groups of 4 ROW tiles sweep the column tiles (an XCD owns rows), g > 0 = groups of g COLUMN tiles sweep the rows (an XCD owns a slice of W)
};

const char* aigv_gemm_check(const GemmArgs& a, int epi);   // nullptr if the shapes fit the kernel
hipError_t aigv_launch_gemm(const GemmArgs& a, int epi, hipStream_t s);           // 128x128 tile kernel (gemm.hip)
// split-K for latency-bound tails: k_slices x the tiles of the 128 kernel write fp32 slabs, then one pass sums them in a
// fixed order and applies epilogue `epi` (deterministic; ws holds k_slices*M*N floats)
// (tile256: the slices come from the 256x256 kernel - needs N % 256 == 0)
hipError_t aigv_launch_gemm_splitk(const GemmArgs& a, int epi, int k_slices, float* ws, hipStream_t s, bool tile256 = false);
hipError_t aigv_launch_gemm256_partial(const GemmArgs& a, hipStream_t s);   // a.part / a.k_slices filled in
// e4m3 operands on the block-scaled fp8 MFMA (unit block scales; per-row x per-column fp32 scales applied to the accumulator before the
// epilogue proper); epi = STORE, GELU, LS_RESID, RESID or SWIGLU with the bf16 kernel's rounding points
hipError_t aigv_launch_gemm256_fp8(const GemmArgs& a, int epi, hipStream_t s);
hipError_t aigv_launch_gemm256_fp8_partial(const GemmArgs& a, hipStream_t s);   // a.part / a.k_slices filled in; slabs hold scaled sums
hipError_t aigv_launch_gemm_splitk_fp8(const GemmArgs& a, int epi, int k_slices, float* ws, hipStream_t s);
// bf16 rows -> e4m3 bytes + one fp32 scale per row (amax / 448); q = e4m3_rne(x * (448 / amax))
// RMSNorm whose result goes straight to e4m3 rows + row scales (== aigv_launch_rmsnorm then aigv_launch_quant_fp8_rows, bit for bit)
hipError_t aigv_launch_rmsnorm_quant_fp8(const bf16_t* x, int ldx, const bf16_t* w, uint8_t* q, int ldq, float* scale, int rows, int H, float eps,
                                         hipStream_t s);
hipError_t aigv_launch_quant_fp8_rows(const bf16_t* x, int ldx, int rows, int K, uint8_t* q, int ldq, float* scale, hipStream_t s);
bool aigv_gemm256_supported(const GemmArgs& a);
// one launch: the body tiles of a row plan with epilogue `epi` + the split-K slices (fp32 slabs a.part) of its tail tiles; then
// aigv_launch_gemm_finalize over the tail table sums the slabs (the second half of aigv_launch_gemm_splitk)
hipError_t aigv_launch_gemm256_fused(const GemmArgs& a, int epi, hipStream_t s);
hipError_t aigv_launch_gemm_finalize(const GemmArgs& a, int epi, int k_slices, const float* ws, hipStream_t s);
hipError_t aigv_launch_gemm256(const GemmArgs& a, int epi, hipStream_t s);        // 256x256 phase-interleaved kernel
// 256x128 tile, four waves, two co-resident workgroups per CU (gemmco.hip): same bits as the 256 kernel; takes the half-tile table too
bool aigv_gemmco_supported(const GemmArgs& a);
hipError_t aigv_launch_gemmco(const GemmArgs& a, int epi, hipStream_t s);

// ---- attention ------------------------------------------------------------------------------------
// Packed varlen layout: sequence s occupies rows cu[s] .. cu[s+1]-1 of the token-major buffers.
// q/k/v point at column 0 of head 0 of their operand inside a (possibly fused) row:
//   q head hq  at column (hq / g) * q_group_stride + (hq % g) * D        (g = n_heads / n_kv_heads)
//   kv head hk at column hk * kv_head_stride
struct AttnArgs {
  const bf16_t* q; int ldq;
  const bf16_t* k; int ldk;
  const bf16_t* v; int ldv;
  bf16_t* o; int ldo;           // [tokens, n_heads*D]
  const int32_t* cu;            // [n_seq+1] cumulative lengths (device)
  int n_seq, max_len;
  int n_heads, n_kv_heads;
  int q_group_stride, kv_head_stride;
  int kv_len_offset;            // keys that precede the first query row of every sequence (0: plain prefill)
  const int32_t* kv_off;        // per-sequence form of kv_len_offset (device, [n_seq]); overrides it when non-null
  size_t kv_seq_stride;         // 0: K/V rows are packed like the query rows (row0 * ldk); else K/V of sequence s start at
                                // s * kv_seq_stride elements (a KV cache [seq][kv head][capacity][D]: ldk = D, kv_head_stride = cap * D)
  int causal;
  float post_div;               // score = bf16(bf16(q.k) / post_div)  (LLM: sqrt(d); ViT: 1, q is pre-scaled)
  int round_scores;             // != 0: the two bf16 roundings of the line above are applied, as the reference's eager path does; 0: scores stay fp32
  float q_prescale;             // q <- bf16(q * q_prescale)           (ViT: d^-1/2; LLM: 1)
  // RoPE applied to the QUERY rows as they are loaded (same three bf16 roundings as rope_kernel); K must already be rotated.
  // null = queries are used as stored.  rope_pos: position of every packed query row; tables [max_pos, D/2] bf16.
  const int32_t* rope_pos; const bf16_t* rope_cos; const bf16_t* rope_sin;
  int rope_pos_is_row;          // != 0: the position of query row r of a sequence IS kv offset + r (plain prefill / continuation): the kernel computes it
                                // instead of loading rope_pos (one dependent global load fewer in front of the cos / sin rows in every workgroup's prologue)
  int q_tail;                   // > 0: only the last q_tail query rows of every sequence are computed (others left unwritten)
  int uniform_len;              // every sequence has max_len rows (InternViT frames): lets the dispatcher split the query rows between kernels
  // row range of ONE kernel launch (set by aigv_launch_attention when it splits the rows between the two kernels; 0 = no limit):
  int waves;                    // 0 = default (4 waves per workgroup); 4 / 8 forced (A/B: aigv_tune_attention / aigv_ctx_tune)
  int lead_key;                 // != 0: non-causal key counts 64 j + 1 run as full tiles over keys 1.. + key 0 merged in the epilogue (opt-in; attention.hip "lead key")
  int q_begin;                  // the launch computes query rows >= q_begin (a multiple of the workgroup's 128 rows) only; 0 everywhere at present
};
const char* aigv_attn_check(const AttnArgs& a, int head_dim);
hipError_t aigv_launch_attention(const AttnArgs& a, int head_dim, hipStream_t s);
// decode: one query row per sequence (fused qkv row), KV cache [seq][kv head][cap][D]; split-KV two-pass kernel,
// ws = aigv_attention_decode_ws_floats(...) floats of scratch
size_t aigv_attention_decode_ws_floats(int n_seq, int n_kv, int g, int cap);
hipError_t aigv_launch_attention_decode(const bf16_t* q, int ldq, int q_group_stride, const bf16_t* kc,
                                        const bf16_t* vc, const int32_t* kv_lens, int cap, bf16_t* o, int ldo,
                                        int n_seq, int n_kv, int g, int head_dim, float post_div, int max_kv_len,
                                        float* ws, hipStream_t s);

// ---- row-wise / elementwise ---------------------------------------------------------------------
// LayerNorm over rows of length H (fp32 statistics, bf16 out).
hipError_t aigv_launch_layernorm(const bf16_t* x, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* y, int ldy,
                                 int rows, int H, float eps, hipStream_t s);
// RMSNorm (fp32 normalise -> bf16 -> * weight -> bf16).  row_idx (optional) gathers input rows.
hipError_t aigv_launch_rmsnorm(const bf16_t* x, int ldx, const bf16_t* w, bf16_t* y, int ldy, int rows, int H,
                               float eps, const int32_t* row_idx, hipStream_t s);
// cls drop + pixel-shuffle v2 gather (pre-projector tokens; the frame-DP all-gather payload)
hipError_t aigv_launch_pixel_shuffle(const bf16_t* vit, int grid, int Hv, bf16_t* out, int frames, hipStream_t s);
// im2col for the patch-embed GEMM: frames NCHW bf16 -> [F*g*g, Kp] (k = c*P*P + py*P + px, zero padded)
hipError_t aigv_launch_im2col(const bf16_t* frames, int F, int C, int S, int P, int Kp, bf16_t* out, hipStream_t s);
// x[f*(np+1)] = cls_pos (class token + its position row, precomputed) for every frame
hipError_t aigv_launch_gather_rows(const bf16_t* src, int ld, const int32_t* idx, int n, bf16_t* dst, int H, hipStream_t s);
hipError_t aigv_launch_scatter_rows(const bf16_t* src, const int32_t* idx, int n, bf16_t* dst, int ld, int H, hipStream_t s);   // dst[idx[i]] = src[i]
hipError_t aigv_launch_cls_rows(const bf16_t* cls_pos, bf16_t* x, int F, int tokens_per_frame, int H, hipStream_t s);
// RoPE in place on the fused qkv rows: per kv group, slots 0..g (q heads and K) are rotated.
// slots [first_rot, first_rot + n_rot) of every group are rotated in place (q heads + K: first_rot 0, n_rot g + 1; K only: g, 1)
hipError_t aigv_launch_rope(bf16_t* qkv, int ld, const int32_t* pos, const bf16_t* cos, const bf16_t* sin,
                            int tokens, int n_rot, int slots, int n_groups, int D, hipStream_t s, int first_rot = 0);
// token embedding + visual/motion scatter: slot[t] < 0 -> tok_emb[ids[t]]; < n_vis -> vis[slot]; else motion
hipError_t aigv_launch_embed(const int64_t* ids, const int32_t* slot, const bf16_t* emb, const bf16_t* vis,
                             const bf16_t* motion, int n_vis, bf16_t* out, int tokens, int H, hipStream_t s);
// skinny (R <= 64) weight-streaming GEMM; epi: 0 store(+bias) 1 residual 2 swiglu 3 gelu(+bias) 6 layer-scale+residual
hipError_t aigv_launch_skinny_gemm(const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K,
                                   const bf16_t* bias, const bf16_t* resid, int ldr, bf16_t* out, int ldo, int epi,
                                   hipStream_t s, const bf16_t* ls = nullptr, int p = 1);
// p: 0 = 16-row slabs with FOUR K slices for every row count (the form does not depend on R: batch-invariant bits; aigv_set_gemm_mode 1);
// 1 = 16-row slabs, eight K slices for a one-tile GEMV with at most one slab per CU; 2 / 4 = the sub-slab forms (8 / 4 rows per workgroup and slab, R <= 8 / 4, store / residual / swiglu only):
// same result up to fp32 summation order, 2x / 4x the workgroups - for widths whose 16-row slabs leave CUs unevenly loaded
// e4m3 form of the decode GEMVs (head8.hip; fp8 mode of the InternLM2 linears).  epi: 1 residual, 2 swiglu, 7 wqkv with RoPE + KV append
// (rk); norm_w != null: the RMSNorm in front of the linear is applied by the kernel.  R <= 4 rows; W8 [N][ldw bytes] e4m3, w_scale [N]
struct AigvRopeKv {
  const int32_t* pos; const int32_t* seq;      // position / cache sequence of every x row (device)
  const bf16_t* cos; const bf16_t* sin;        // [max_pos, 64]
  bf16_t* kc; bf16_t* vc;                      // [seq][kv head][cap][128]
  int g, n_kv, cap;
};
bool aigv_skinny_fp8_supported(int K, bool with_norm);
hipError_t aigv_launch_skinny_fp8(const bf16_t* x, int ldx, int R, const uint8_t* W8, int ldw, const float* w_scale, int N, int K, const bf16_t* resid,
                                  int ldr, bf16_t* out, int ldo, int epi, const AigvRopeKv* rk, const bf16_t* norm_w, float eps, int p, hipStream_t s);
bool aigv_skinny_norm_fusable(int K);   // hidden widths the fused-norm decode GEMVs exist for
// decode: SwiGLU(RMSNorm(x) W13^T) with the norm applied by the GEMV itself (R <= 4 rows, K <= 8192); same bits as rmsnorm + skinny swiglu
hipError_t aigv_launch_skinny_swiglu_normed(const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K, bf16_t* out, int ldo,
                                            const bf16_t* norm_w, float norm_eps, hipStream_t s, int p = 1);
// decode: wqkv projection of the new tokens with RoPE (q heads, K) and the K / V cache append in the GEMV's epilogue (head_dim 128)
hipError_t aigv_launch_skinny_rope_kv(const bf16_t* x, int ldx, int R, const bf16_t* W, int ldw, int N, int K, bf16_t* qkv, int ldo,
                                      const int32_t* pos, const int32_t* seq, const bf16_t* cos, const bf16_t* sin, bf16_t* kc, bf16_t* vc,
                                      int g, int n_kv, int cap, int head_dim, hipStream_t s, const bf16_t* norm_w = nullptr, float norm_eps = 0.f, int p = 1);
// lm-head on R gathered rows + argmax over the vocabulary (first maximal index, bf16-rounded logits)
hipError_t aigv_launch_lm_head_argmax(const bf16_t* h, int R, int H, const bf16_t* W, int V,
                                      unsigned long long* packed, int64_t* out_idx, float* out_val, hipStream_t s);
// lm-head logits (bf16, the matmul output the reference upcasts) of R rows into out[R, ldo], ldo >= roundup(V, 4)
hipError_t aigv_launch_lm_head_logits(const bf16_t* h, int R, int H, const bf16_t* W, int V, bf16_t* out, int ldo, hipStream_t s);
// score head: x[B,H] -> chain of Linear+ReLU (bf16 rounding after each Linear), NaN/Inf guard on x
struct ScoreHeadArgs {
  const bf16_t* x; int ldx; int B;
  int n_layers; int dims[9];            // dims[0] = H, dims[i+1] = out of layer i
  const bf16_t* w[8]; const bf16_t* b[8];
  float* score;                         // [B] (the bf16 result widened to fp32)
};
// scratch: 3 * B * max(dims) bf16; B <= 64 per call
hipError_t aigv_launch_score_head(const ScoreHeadArgs& a, bf16_t* scratch, hipStream_t s);
// KV-cache append for decode: copies the K/V slots of fused qkv rows into [n_seq, n_kv, cap, D] caches
// decode: RoPE (q heads in place, K) + K / V cache append in one launch
hipError_t aigv_launch_rope_kv_store(bf16_t* qkv, int ld, const int32_t* seq_of_tok, const int32_t* pos, const bf16_t* cos,
                                     const bf16_t* sin, bf16_t* kc, bf16_t* vc, int tokens, int n_groups, int g, int D, int cap,
                                     hipStream_t s);
hipError_t aigv_launch_kv_store(const bf16_t* qkv, int ld, const int32_t* seq_of_tok, const int32_t* pos,
                                bf16_t* kc, bf16_t* vc, int tokens, int n_groups, int g, int D, int cap,
                                hipStream_t s);
// frame ingest: uint8 HWC RGB -> (u/255 - mean)/std -> bf16 NCHW (torchvision ToTensor + Normalize + bf16 cast)
hipError_t aigv_launch_frame_ingest(const uint8_t* hwc, int n_frames, int H, int W, const float* mean, const float* stdv,
                                    bf16_t* out, hipStream_t s);
// uint8 HWC frames at any resolution -> Pillow-exact BICUBIC resize -> uint8 HWC (out_u8, may be null) and / or normalised
// bf16 NCHW (out_nchw, may be null); tmp_u8 holds the horizontal pass: n_frames * in_h * out_w * 3 bytes
hipError_t aigv_launch_frame_resize_ingest(const uint8_t* hwc, int n_frames, int in_h, int in_w, int out_h, int out_w,
                                           const float* mean, const float* stdv, uint8_t* tmp_u8, uint8_t* out_u8, bf16_t* out_nchw,
                                           hipStream_t s);
// records a message for aigv_last_error(NULL) from translation units other than api.hip (thread-local, like every handle-less error)
void aigv_set_error(const char* msg);
// a[i] += 1, b[i] += 1 for i < n (decode bookkeeping kept on the device: positions and visible KV lengths)
hipError_t aigv_launch_advance(int32_t* a, int32_t* b, int n, hipStream_t s);
hipError_t aigv_launch_decode_eos(int64_t* tok, int32_t* state, int n, const int64_t* eos_host, int n_eos, int64_t pad, hipStream_t s);
// small host int arrays passed by value as kernel arguments (no memcpy, no implicit host/stream sync)
#define AIGV_SMALL_INTS 256
struct SmallInts { int32_t v[AIGV_SMALL_INTS]; };
// pos[t] = t - cu[seq(t)] (+ pos_offset[seq(t)]), seq[t], and a device copy of cu[0..n_seq]; at most 127 sequences
hipError_t aigv_launch_seqpos(const int32_t* cu_host, int n_seq, int32_t* pos, int32_t* seq, int32_t* cu_dev, int tokens,
                              hipStream_t s, const int32_t* pos_offset_host = nullptr);
hipError_t aigv_launch_write_ints(const int32_t* host, int n, int32_t* dst, hipStream_t s);
// beam search: new KV cache slot i = the first lens[i] positions of slot parent[i] of the current cache (all layers, all kv heads); dst != src
hipError_t aigv_launch_kv_reorder(const bf16_t* sk, const bf16_t* sv, bf16_t* dk, bf16_t* dv, const int32_t* parent, const int32_t* lens, int n,
                                  int layers, int nkv, int cap, int D, size_t kv_layer, int max_len, hipStream_t s);
