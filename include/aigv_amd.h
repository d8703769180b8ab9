/* aigv_amd.h — C ABI of libaigv_amd.so: the MI355X (gfx950) forward path of the AIGV-Assessor scorer.
 *
 * The reference has no FFI: its hot path sits behind the Python class InternVLChatModel
 * (internvl/model/internvl_chat_eval2/modeling_internvl_chat.py).  This header is the boundary a
 * maintainer binds instead (ctypes stub in INTEGRATION.md); each entry point names the reference
 * code it replaces.  Plain pointers and sizes only — no torch types.  All tensor pointers are DEVICE
 * pointers (bf16 = raw uint16 bits) unless marked host; outputs are caller-allocated; work is enqueued
 * on the caller's HIP stream and nothing synchronises except where noted.  Every function returns
 * 0 on success or a negative aigv_status; aigv_last_error() gives the message.  Never throws.
 *
 * Packed sequence convention: the B clips of a batch are concatenated without padding; clip b owns
 * token rows cu_seqlens[b] .. cu_seqlens[b+1]-1 and its positions restart at 0 — what the reference
 * computes for un-padded rows (modeling_internlm2.py:907-912) and for left/right padded batches
 * (:1141-1147).
 *
 * Threading: the intended deployment is one process per GPU (the reference's eval loop is single-threaded too).  A context is
 * not re-entrant - HOST calls on one context must not overlap - but it owns all the device state it uses (weights, workspaces,
 * split-K slab scratch), so several contexts, also on different devices of one process, are independent.
 * Streams: a context's device state comes in two halves.  aigv_vit_forward uses the InternViT workspaces, the InternViT row-plan
 * table and an InternViT split-K scratch of its own; every other entry point (aigv_project, aigv_motion_project, aigv_llm_prefill /
 * _extend, decode) uses the rest.  So ONE aigv_vit_forward may be in flight on one stream beside ONE pass of the other half on another
 * stream (the next clip's visual front beside this clip's InternLM2 pass); within a half, work must be stream-ordered (one launch
 * stream at a time, or the caller's events between them).
 * The context-free aigv_op_* entry points share one split-K scratch per DEVICE (created on first use, never regrown): overlap
 * them only from one stream per device.  Frame-resize coefficient tables are cached per (device, size pair) and immutable.
 */
#ifndef AIGV_AMD_H
#define AIGV_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIGV_ABI_VERSION 1

typedef struct aigv_ctx aigv_ctx;

enum aigv_status {
  AIGV_OK = 0,
  AIGV_ERR_ARG = -1,      /* bad argument / shape the kernels do not cover */
  AIGV_ERR_HIP = -2,      /* a HIP call failed */
  AIGV_ERR_STATE = -3,    /* weights missing, capacity exceeded, no KV state ... */
  AIGV_ERR_ALLOC = -4
};

enum aigv_dtype { AIGV_BF16 = 0, AIGV_F32 = 1 };

/* Model + capacity description.  Field meanings follow the reference's config classes
 * (configuration_intern_vit.py:20-119, configuration_internlm2.py:26-150, configuration_internvl_chat.py). */
typedef struct aigv_config {
  /* InternViT */
  int32_t vit_hidden, vit_inter, vit_heads, vit_layers;
  int32_t image_size, patch_size, num_channels;
  int32_t vit_norm_rms;      /* 0: nn.LayerNorm (ViT-300M), 1: InternRMSNorm (ViT-6B) */
  int32_t vit_qk_norm;       /* full-width QK RMSNorm (modeling_intern_vit.py:148-151) */
  int32_t vit_qkv_bias;
  float vit_eps;
  int32_t select_layer;      /* -1 = last (modeling_internvl_chat.py:509-518) */
  int32_t shuffle;           /* 1/downsample_ratio (2) */
  /* InternLM2 */
  int32_t llm_hidden, llm_inter, llm_heads, llm_kv_heads, llm_layers, vocab;   /* head width 128; 1..8 query heads per KV head (8B: 4, 20B: 6); widths multiples of 128 */
  float rms_eps;
  int32_t max_positions;     /* rows of the RoPE tables the host uploads */
  /* heads */
  int32_t motion_dim;        /* SlowFast feature width (2304) */
  int32_t n_score_layers;    /* 5 */
  int32_t score_dims[8];     /* 1024,256,64,16,1 */
  /* capacity (workspaces are allocated once, in aigv_ctx_create) */
  int32_t max_frames;        /* frames per aigv_vit_forward call (processed in chunks of vit_chunk) */
  int32_t vit_chunk;         /* frames per ViT pass (workspace size) */
  int32_t max_tokens;        /* packed tokens per aigv_llm_prefill call */
  int32_t max_seqs;          /* clips per call */
  int32_t max_out_rows;      /* lm-head rows per call (answer rows), <= 64 per launch, looped */
  int32_t kv_capacity;       /* per-clip KV-cache length for decode (0 = no decode support) */
} aigv_config;

/* ---- lifetime ------------------------------------------------------------------------------------ */
int aigv_abi_version(void);
int aigv_sizeof_config(void);                        /* sizeof(aigv_config): binding self-check */
int aigv_ctx_create(int device, const aigv_config* cfg, aigv_ctx** out);
void aigv_ctx_destroy(aigv_ctx* ctx);
/* Change the CAPACITIES of a context (max_frames, vit_chunk, max_tokens, max_seqs, max_out_rows, kv_capacity, max_positions; every
 * other field of `cfg` must equal the context's): the workspaces are re-allocated, the loaded weights - and in fp8 mode their e4m3
 * copies and the mode itself - stay where they are.  Kept KV state is dropped.  When max_positions changed the rotary tables
 * ("rope.cos" / "rope.sin") must be loaded again at the new length and aigv_finalize_weights called before the next pass; that
 * finalize keeps the fp8 mode (see aigv_set_precision).  Synchronises the device.  A failed resize (out of memory) leaves the
 * context unusable: destroy it. */
int aigv_ctx_resize(aigv_ctx* ctx, const aigv_config* cfg);
const char* aigv_last_error(const aigv_ctx* ctx);     /* ctx may be NULL (creation errors) */
/* Read and clear the HIP runtime's sticky last-error state of the calling thread.  The launchers report hipGetLastError() after every launch, so an error
 * left behind by something ELSE - a stream capture the caller's framework abandoned (hipErrorStreamCaptureInvalidated stays pending after the failed
 * hipStreamEndCapture) - would be blamed on the next launch of this library.  A host that falls back from a failed capture to eager launches calls this first. */
void aigv_clear_hip_error(void);

/* Weight upload.  `name` is the reference state-dict key, e.g.
 * "vision_model.encoder.layers.3.attn.qkv.weight", "language_model.model.layers.0.attention.wqkv.weight",
 * "mlp1.1.bias", "mlpscore.fc1.weight"; plus two host-precomputed tables "rope.cos" / "rope.sin"
 * [max_positions, head_dim/2] (modeling_internlm2.py:176-194) and the position table already resized to this
 * ctx's image size (modeling_intern_vit.py:87-93).  Data is bf16 (or f32, converted); `on_device` says where
 * `data` lives.  The ctx keeps its own repacked copy (padded patch kernel, interleaved w1|w3).  Synchronous. */
int aigv_load_weight(aigv_ctx* ctx, const char* name, const void* data, const int64_t* shape, int ndim,
                     int dtype, int on_device);
int aigv_finalize_weights(aigv_ctx* ctx);             /* checks that every tensor the config needs is present */

/* ---- hot path ------------------------------------------------------------------------------------ */
/* InternVisionModel.forward + cls drop + pixel_shuffle(v2)  (modeling_intern_vit.py:95-107,216-228,324-362;
 * modeling_internvl_chat.py:492-527).  frames: [F,3,S,S] bf16 NCHW.  out: [F*ntok, shuffle^2*vit_hidden] bf16
 * pre-projector tokens — the payload of the frame-DP all-gather. */
int aigv_vit_forward(aigv_ctx* ctx, const void* frames, int n_frames, void* out_tokens, void* stream);
/* mlp1: LayerNorm -> Linear -> GELU -> Linear (modeling_internvl_chat.py:238-243,529). rows x proj_in -> rows x llm_hidden */
int aigv_project(aigv_ctx* ctx, const void* tokens, int rows, void* out, void* stream);
/* motion_mlp on the SlowFast feature (modeling_internvl_chat.py:244-249,345). [B, motion_dim] -> [B, llm_hidden] */
int aigv_motion_project(aigv_ctx* ctx, const void* motion_feature, int n_clips, void* out, void* stream);

/* InternVLChatModel.forward from the embedding scatter onwards (modeling_internvl_chat.py:324-488,
 * modeling_internlm2.py:868-998,1094-1096):
 *   ids[T] int64 packed token ids; slot[T] int32: -1 = text token (embedding row ids[t]); 0..n_vis-1 = row of
 *   `vis`; n_vis + b = row b of `motion`.  cu_seqlens: HOST int32[B+1].
 *   score_rows[B]  (HOST int32, packed row index of hidden[:, -4] per clip) -> score[B] float (device), may be NULL
 *   logit_rows[R]  (HOST int32, packed row indices whose next-token argmax is wanted) -> argmax[R] int64 (device)
 *   keep_kv != 0 stores K/V of every layer into the ctx KV cache for aigv_decode_step.                      */
int aigv_llm_prefill(aigv_ctx* ctx, const int64_t* ids, const int32_t* slot, const int32_t* cu_seqlens, int n_clips,
                     const void* vis, int n_vis, const void* motion, const int32_t* score_rows, float* score,
                     const int32_t* logit_rows, int n_logit_rows, int64_t* argmax, int keep_kv, void* stream);
/* Continue the sequences kept by aigv_llm_prefill(keep_kv = 1) with new TEXT tokens (no visual slots): ids = DEVICE int64 of the
 * packed new tokens, cu = HOST int32[n_clips + 1] over the new tokens only; rows index the packed new tokens.  The new rows attend
 * causally to the cached keys and to themselves (positions continue after the cache).  commit = 0 leaves the cache lengths
 * unchanged, so several continuations of ONE prefix - the four quality-perspective questions behind the same video tokens
 *  - can be scored one after the other; commit = 1 appends them (a longer chat turn before aigv_decode_step).
 * Replaces re-running modeling_internvl_chat.py:306-488 from the first token for every question. */
int aigv_llm_extend(aigv_ctx* ctx, const int64_t* ids, const int32_t* cu, int n_clips, const int32_t* score_rows, float* score,
                    const int32_t* logit_rows, int n_logit_rows, int64_t* argmax, int commit, void* stream);

/* Replicate the n kept sequences `copies` times (cache slots [0, n) -> [n, 2n), ...; needs n * copies <= max_seqs): the copies can
 * then take DIFFERENT continuations in one aigv_llm_extend call over n * copies sequences (sequence c * n + b continues clip b),
 * which streams the decoder weights once for all of them. */
int aigv_kv_fork(aigv_ctx* ctx, int copies, void* stream);
/* Beam search (HF generate(num_beams > 1): the cache reorder of GenerationMixin._beam_search, transformers/generation/utils.py; the
 * reference inherits it through language_model.generate, modeling_internvl_chat.py:798-809).  After this call kept sequence i holds
 * what sequence parent[i] had cached - its first len[i] positions - in every layer; parent / len are HOST arrays of n = the number of
 * kept sequences (aigv_llm_prefill(keep_kv) x aigv_kv_fork).  The positions aigv_decode_step keeps on the device are not touched: the
 * beams of one search all have the same length.  The first call allocates a second KV cache of the context's size (the gather is never
 * in place; the two caches swap). */
int aigv_kv_reorder(aigv_ctx* ctx, const int32_t* parent, const int32_t* len, int n, void* stream);

/* Arithmetic of the InternLM2 linears (aigv_llm_prefill, aigv_llm_extend, aigv_decode_step; BASELINE config 5).  AIGV_PRECISION_BF16 (default) is the reference's
 * dtype flow.  AIGV_PRECISION_FP8_LLM: wqkv, wo, w1|w3, w2 of every decoder layer - except wo / w1|w3 / w2 of the LAST layer, which act on
 * the few consumed rows - run on the e4m3 MFMA: weights quantised once per output channel (scale = amax / 448), activations per token row
 * on the fly (aigv_op_quant_fp8_rows), fp32 accumulation, the bf16 path's epilogues and rounding points after the scaled accumulator.
 * Attention, norms, RoPE, residual stream, lm-head and score head stay bf16.  aigv_llm_extend runs the same e4m3 linears as the
 * prefill (a continuation scores like the same tokens inside one prefill of this mode); aigv_decode_step streams the e4m3 copies too
 * (same rule: all but the post-attention half of the last layer; the token rows are normalised and quantised inside the GEMVs) for up
 * to 4 sequences and hidden / intermediate widths of 2048 j (j = 2, 3 / 2, 3, 7, 8), else it decodes from the bf16 weights.  The reference
 * has no fp8 path: results move by the quantisation noise (oracle/fp8.py restates this mode).  EXPERIMENTAL: over the 37 clips recorded from the
 * reference the mode's SCORES correlate with the reference's at SRCC 0.72 (bf16 path: 0.985) - not usable as scores on that evidence.
 * Every e4m3 linear of a pass is ONE launch in full K (round 6), so in this mode too a clip's bits do not depend on its batch mates.  First call
 * quantises the weights (extra memory: one byte per InternLM2 linear weight).  Needs H, qkv width, 2*I multiples of 256.
 * aigv_finalize_weights after a reload of an InternLM2 linear (wqkv / wo / w1 / w3 / w2 of any layer) drops the e4m3 copies and returns the
 * context to bf16: set the mode again after it.  A finalize that follows other uploads only (the rotary tables after aigv_ctx_resize, a
 * score head, ViT weights) keeps the copies and the mode. */
enum aigv_precision { AIGV_PRECISION_BF16 = 0, AIGV_PRECISION_FP8_LLM = 1 };
int aigv_set_precision(aigv_ctx* ctx, int mode);
/* Last-layer row trimming in aigv_llm_prefill (default on): when at most 16 rows per clip are consumed (score rows + logit rows), the
 * last decoder layer computes attention only for the query blocks holding them and finishes wo / MLP / final norm on a compact
 * copy of those rows.  Rows are independent after attention, so the outputs are those of the untrimmed pass (up to the fp32
 * summation order of the kernel that runs the few rows); off = every row through every layer, as the reference does. */
int aigv_set_row_trimming(aigv_ctx* ctx, int on);
/* Numerics of the prefill attention (InternViT and InternLM2): 0 (default) = the score matrix stays fp32 up to the softmax; 1 = it carries
 * the reference's rounding points - s = bf16(q k^T), InternLM2 also bf16(s / sqrt(d)) (modeling_internlm2.py:417,
 * modeling_intern_vit.py:153).  At op level form 1 sits 4x closer to the reference's eager bf16 result; END TO END, over the 37 clips the
 * imported reference was recorded on, it is NOT closer to the reference's scores (3.09 against 2.80 bf16 ulps mean; the reference moves
 * 2.56 against itself with the host's thread count), it is farther from the reference's fp32 scores (3.59 against 2.01) and it costs
 * 1.9 ms of a 116 ms step: profiles/r5_parity_stats.txt.  P is rounded un-normalised in both forms (no measurable effect). */
#define AIGV_ATTENTION_NUMERICS_DEFAULT 0
int aigv_set_attention_numerics(aigv_ctx* ctx, int mode);
/* The mode in force for `ctx`; ctx == NULL: the mode a fresh context starts in (AIGV_ATTENTION_NUMERICS_DEFAULT; needs no GPU - the host
 * tests hold INTEGRATION.md's "default" sentence against it). */
int aigv_get_attention_numerics(const aigv_ctx* ctx);
/* GEMM tile choice of THIS context: -1 = follow the process default set by aigv_tune_gemm (the state after aigv_ctx_create),
 * 0 = the per-sequence row plan (aigv_op_gemm_rows: the default; a clip's / frame's bits do not depend on its batch mates),
 * 1 = every row on the 128x128 kernel, 2 = every row on the 256x256 kernel wherever its shape rules allow, 4 = every row on the
 * co-resident 256x128 kernel (all three in full K, so batch-invariant too and bit-identical with one another: test aliases).  The
 * co-resident kernel is otherwise NOT used: AIGV_TUNE_CO_KMAX defaults to 0 (it lost the in-step A/B, profiles/r5_gemmco.txt); an experiment
 * that raises the knob sends the GEMMs with K <= its value there in modes 0 and 3.  Split-K scratch is per context too.  aigv_llm_extend (continuations of a kept
 * prefix) still uses the batch-level cost-model dispatch of aigv_op_gemm. */
int aigv_set_gemm_mode(aigv_ctx* ctx, int mode);

/* One greedy decode step for every clip of the last keep_kv prefill (generate(): modeling_internvl_chat.py:769-811,
 * modeling_internlm2.py:1126-1163).  ids[B] int64 device (the previous tokens) -> next[B] int64 device. */
int aigv_decode_step(aigv_ctx* ctx, const int64_t* ids, int64_t* next, void* stream);
/* The full next-token distribution of the rows the last aigv_llm_prefill / aigv_llm_extend / aigv_decode_step consumed: lm-head
 * logits of their final hidden states (kept in the context, in the order [score rows | logit rows]; a decode step keeps its
 * n_clips rows) as the bf16 values the reference upcasts with .float() (modeling_internlm2.py:1095-1096).
 * logits: DEVICE bf16 [n_rows, ldo], ldo >= vocab rounded up to a multiple of 4 (columns >= vocab are padding).  For
 * generate() with do_sample (HF sampling needs the distribution; the greedy paths use the fused argmax and never call this). */
int aigv_out_row_logits(aigv_ctx* ctx, int first_row, int n_rows, void* logits_bf16, int ldo, void* stream);
/* The final hidden states (after the last RMSNorm, modeling_internlm2.py:984) of the same rows, bf16 [n_rows, ldo >= llm_hidden]: for the
 * score rows this is the reference's hidden_states[-1][:, -4, :], the input of its score head (modeling_internvl_chat.py:469-481) - the
 * slice of `output_hidden_states=True` the eval path consumes.  Rows in the order [score rows | logit rows]. */
int aigv_out_row_hidden(aigv_ctx* ctx, int first_row, int n_rows, void* hidden_bf16, int ldo, void* stream);
/* End-of-sequence bookkeeping of generate()'s token loop on the device (the reference defers to HF's loop: next = next * unfinished +
 * pad * (1 - unfinished); unfinished &= next not in eos_token_id; stop when every sequence has finished - modeling_internvl_chat.py:
 * 798-809).  tokens: DEVICE int64[n] (n = sequences of the kept KV state), in: the step's raw tokens (aigv_decode_step's `next`, or the
 * prefill's argmax), out: the tokens the loop emits (pad_id for finished sequences).  state: DEVICE int32[n + 1], zeroed by the caller
 * before the first token: state[b] = 1 once sequence b has emitted an end token, state[n] = number of emitted columns in which some
 * sequence was still live (the output length HF returns).  eos_ids: HOST, at most 8.  The host reads `state` only every few tokens,
 * so the loop runs without a per-token host synchronisation. */
int aigv_decode_eos(aigv_ctx* ctx, int64_t* tokens, int32_t* state, const int64_t* eos_ids, int n_eos, int64_t pad_id, void* stream);

/* ---- single operators (parity tests call these through the same ABI) --------------------------------- */
/* C = epilogue(A[M,K] . W[N,K]^T); epi: 0 store, 1 gelu, 2 layerscale+residual, 3 residual, 4 swiglu, 5 patch */
int aigv_op_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* ls,
                 const void* resid, int ldr, const void* pos, int np, int M, int N, int K, int epi, void* stream);
/* The same GEMM with its M = cu_host[n_seq] rows divided into n_seq independent sequences (HOST int32 cu_host[0..n_seq], cu[0] = 0; epi
 * 0..4): the dispatch of the scoring pass.  Every sequence's rows [0, 256 * floor(L / 256)) run in full K as whole tiles addressed
 * through a half-tile table on the 256x256 kernel (AIGV_TUNE_BODY_TILE = 2: on the 128x128 kernel, which sums every element in the same
 * order - same bits, a test alias), its remaining rows as (ragged) half tiles with a split-K factor that depends on (N, K)
 * only, remainders of <= 4 rows on the weight-streaming kernel in its fixed form - a row's result depends on its own sequence alone,
 * never on the other sequences of the call.  Synchronises the stream (test entry point). */
int aigv_op_gemm_rows(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* ls,
                      const void* resid, int ldr, const int32_t* cu_host, int n_seq, int N, int K, int epi, void* stream);
/* the split-K form used for latency-bound row tails: k_slices x tiles write fp32 slabs into ws_f32 (k_slices*M*N floats),
 * one pass sums them in slice order and applies the epilogue (epi 0..4) */
int aigv_op_gemm_splitk(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* ls,
                        const void* resid, int ldr, int M, int N, int K, int epi, int k_slices, void* ws_f32, void* stream);
/* the same with the slices computed by the 256x256 kernel (N % 256 == 0): whole row tiles that would not fill a round */
int aigv_op_gemm_splitk256(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* ls,
                           const void* resid, int ldr, int M, int N, int K, int epi, int k_slices, void* ws_f32, void* stream);
/* fp8 groundwork for BASELINE config 5 (NOT used by the bf16 scoring path; the reference has no fp8 path, so these two are defined by
 * their own arithmetic and tested against a torch restatement of it):
 *   aigv_op_quant_fp8_rows: bf16 [rows, K] -> OCP e4m3 bytes [rows, K] + row_scale[rows] = amax / 448 (1 for an all-zero row);
 *                           q = e4m3_rne(x * (448 / amax)), three fp32 operations, round-to-nearest-even.  K % 8 == 0.
 *   aigv_op_gemm_fp8:       C[M, N] = bf16((sum_k A[m,k] W[n,k]) * row_scale[m] * col_scale[n] + bias[n]) with e4m3 A [M, K] and
 *                           W [N, K] (K contiguous), products exact and accumulated in fp32 on v_mfma_scale_f32_16x16x128_f8f6f4
 *                           (unit block scales) with the 256x256 schedule of the bf16 kernel.  N % 256 == 0, K % 128 == 0,
 *                           lda / ldw in bytes and multiples of 16; row_scale / col_scale DEVICE float.  epi = AIGV_EPI_STORE, _GELU,
 *                           _LS_RESID, _RESID or _SWIGLU: the scaled accumulator takes the place of the bf16 kernel's accumulator,
 *                           every later rounding point is that of aigv_op_gemm. */
int aigv_op_quant_fp8_rows(const void* x_bf16, int ldx, int rows, int K, void* q_e4m3, int ldq, float* row_scale, void* stream);
int aigv_op_gemm_fp8(const void* A_e4m3, int lda, const void* W_e4m3, int ldw, void* C, int ldc, const float* row_scale,
                     const float* col_scale, const void* bias, const void* ls, const void* resid, int ldr, int M, int N, int K, int epi,
                     int k_slices, void* ws_f32, void* stream);   /* k_slices > 1: split-K, ws_f32 = k_slices * M * N floats, K/128 % k_slices == 0 */
int aigv_op_skinny_gemm(const void* x, int ldx, int R, const void* W, int ldw, int N, int K, const void* bias,
                        const void* resid, int ldr, void* out, int ldo, int epi, void* stream);
/* The e4m3 form of a decode GEMV (fp8 mode; what aigv_decode_step runs per linear after aigv_set_precision(fp8)): R <= 4 bf16 rows x
 * [R, K] are (optionally RMS-normalised with norm_w / eps, then) quantised per row inside the kernel, W_e4m3 [N, ldw bytes] with one
 * fp32 scale per output channel, out = epilogue(bf16((acc * row scale) * channel scale)).  epi: 1 residual, 2 swiglu (w1|w3 in 16-row
 * blocks; needs norm_w).  K = 2048 j, j in {2, 3} with a norm, {2, 3, 7, 8} without; p = 1 / 2 / 4: 16 / 8 / 4 rows of W per workgroup
 * (R <= 16 / p). */
int aigv_op_skinny_gemm_fp8(const void* x, int ldx, int R, const void* W_e4m3, int ldw, const float* w_scale, int N, int K, const void* resid,
                            int ldr, void* out, int ldo, int epi, const void* norm_w, float eps, int p, void* stream);
int aigv_op_layernorm(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, int rows, int H,
                      float eps, void* stream);
int aigv_op_rmsnorm(const void* x, int ldx, const void* w, void* y, int ldy, int rows, int H, float eps,
                    const int32_t* row_idx, void* stream);
int aigv_op_rope(void* qkv, int ld, const int32_t* pos, const void* cos, const void* sin, int tokens, int n_rot,
                 int slots, int n_groups, int head_dim, void* stream);
/* q/k/v as in kernels.h AttnArgs; cu is a DEVICE int32[n_seq+1].  causal: bit 0 = causal mask; bit 1 = "every sequence has
 * exactly max_len rows" (InternViT frames), which lets the dispatcher give a short left-over query block to the key-split kernel;
 * bit 2 = the score matrix rounds to bf16 as in the reference's eager path (aigv_set_attention_numerics mode 1); bit 3 = the
 * lead-key form for non-causal key counts 64 j + 1 (full tiles over keys 1.., key 0 merged in the epilogue; opt-in, off in the scoring pass). */
int aigv_op_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                      const int32_t* cu, int n_seq, int max_len, int n_heads, int n_kv_heads, int q_group_stride,
                      int kv_head_stride, int head_dim, int causal, float post_div, float q_prescale, void* stream);
/* the same with RoPE applied to the QUERY rows as they are loaded (pos: DEVICE int32 position of every packed row; cos/sin:
 * DEVICE bf16 [max_pos, head_dim/2]; three bf16 roundings as aigv_op_rope).  K must already be rotated in memory. */
int aigv_op_attention_rope(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                           const int32_t* cu, int n_seq, int max_len, int n_heads, int n_kv_heads, int q_group_stride,
                           int kv_head_stride, int head_dim, int causal, float post_div, float q_prescale, const int32_t* pos,
                           const void* cos, const void* sin, void* stream);
int aigv_op_pixel_shuffle(const void* vit_out, int grid, int vit_hidden, void* out, int n_frames, void* stream);
int aigv_op_im2col(const void* frames, int n_frames, int channels, int image_size, int patch, int kp, void* out,
                   void* stream);
int aigv_op_lm_head_argmax(const void* h, int rows, int hidden, const void* W, int vocab, void* scratch_u64,
                           int64_t* idx, float* val, void* stream);

/* Frame ingest: uint8 [F,H,W,3] RGB frames already at the model resolution -> bf16 NCHW
 * pixel_values = bf16((u/255 - mean[c]) / std[c])  (torchvision ToTensor + Normalize of dataset.py:267-274 and the
 * bf16 cast of stage2_eval.py:932).  mean/std: HOST float[3]. */
int aigv_op_frame_ingest(const void* hwc_u8, int n_frames, int height, int width, const float* mean, const float* stdv,
                         void* out_nchw, void* stream);

/* Frame resize + ingest: uint8 [F,in_h,in_w,3] RGB frames at the video's resolution -> Pillow's BICUBIC
 * `Image.resize((out_w, out_h))` (the reference's dynamic_preprocess tile, internvl/train/dataset.py:702-738 with max_num = 1;
 * stage2_eval.py:453-456), bit-exact with Pillow's 8-bit ImagingResample (22-bit fixed-point coefficients, horizontal pass
 * stored as uint8, then the vertical pass) -> uint8 [F,out_h,out_w,3] in out_u8_hwc (may be NULL) and / or the normalised bf16
 * NCHW pixel_values of aigv_op_frame_ingest in out_nchw (may be NULL).  tmp_u8: DEVICE scratch of F*in_h*out_w*3 bytes.
 * Coefficient tables are computed on the host (double precision, as Pillow does) and cached on the device per size pair; the
 * first call for a size pair synchronises on their upload.  mean/std: HOST float[3].
 * Checked against the live package on random sizes from 8 x 8 to 1200 x 2000 (tests/manual/fuzz_resize.py, Pillow 12.2).  One regime is REFUSED
 * (AIGV_ERR_ARG): frames more than 100 times taller than wide that shrink vertically - there Pillow runs its vertical pass first and the
 * uint8 intermediate makes the order visible; not a video geometry. */
int aigv_op_frame_resize_ingest(const void* hwc_u8, int n_frames, int in_h, int in_w, int out_h, int out_w, const float* mean,
                                const float* stdv, void* tmp_u8, void* out_u8_hwc, void* out_nchw, void* stream);

/* ---- tuning knobs: TESTS AND EXPERIMENTS ONLY ---------------------------------------------------------------------------------------
 * Every knob lives in the CONTEXT (aigv_ctx_tune; value -1 = follow the process default): two contexts of one process can run different
 * kernel forms side by side, and the kernel files hold no mutable state - each launch carries its selectors.  aigv_tune_gemm /
 * aigv_tune_attention / aigv_tune_skinny set the PROCESS defaults, which apply to the context-free aigv_op_* entry points and to
 * contexts that left a knob at -1 (not thread-safe: in-process A/B scripts and tests that must force a kernel form).  A deployment calls
 * none of them: what a context needs per instance is aigv_set_gemm_mode / aigv_set_precision / aigv_set_row_trimming /
 * aigv_set_attention_numerics above. */
enum aigv_tune_knob {
  AIGV_TUNE_GEMM_MODE = 0,       /* = aigv_set_gemm_mode */
  AIGV_TUNE_GEMM256_ORDER = 1,   /* tile order of the 256 kernel: 0 by weight size, 1 row groups, 1 + g groups of g column tiles */
  AIGV_TUNE_GEMM256_VARIANT = 2, /* 0 the shipped schedule, 1 + v schedule variant v (0..3); 5..7: the 256x256 kernel runs its shipped schedule, the co-resident kernel a diagnostic form (AIGV_CO_DIAG builds only) */
  AIGV_TUNE_ATTN_WAVES = 3,      /* prefill attention: 0 default, 4 / 8 waves per workgroup */
  AIGV_TUNE_SKINNY_P = 4,        /* decode GEMV form: 0 per-shape default, 1 / 2 / 4; 1000 + (wqkv | wo << 3 | w1w3 << 6 | w2 << 9) = one form per GEMV */
  AIGV_TUNE_BODY_TILE = 5,       /* tile kernel of a row plan's body rows: 0 / 1 = 256x256 (shipped), 2 = 128x128 (same bits, slower) */
  AIGV_TUNE_CO_KMAX = 6,         /* GEMMs with K <= value run on the co-resident 256x128 kernel (same bits as the 256x256 one): 0 never (default) */
  AIGV_TUNE_TAIL_SLICES = 7,     /* split-K factor of the row plans' tail half tiles: 0 = the per-(N, K) rule, 1 = inside the body's launch, S = one factor wherever it divides */
  AIGV_TUNE_ATTN_LEAD_KEY = 8,   /* InternViT attention (64 j + 1 keys): 1 = full tiles over keys 1.. + key 0 merged in the epilogue (another summation order) */
  AIGV_TUNE_DECODE_FUSED = 9,    /* decode: 1 (default) = RMSNorm inside the GEMV that consumes it, 0 = separate norm kernels */
  AIGV_TUNE_DECODE_FP8 = 10,     /* decode in fp8 mode: 1 (default) = e4m3 GEMVs, 0 = bf16 GEMVs */
  AIGV_TUNE_SKINNY_P8 = 11,      /* form of the e4m3 decode GEMVs: 0 per-GEMV defaults, 1 / 2 / 4 */
  AIGV_TUNE_FUSE_TAILS = 12,     /* the tail tiles' K slices inside the body's launch: 0 = when the body leaves CUs idle (default), 1 = never, 2 = always; same bits */
  AIGV_TUNE_LONE_BODY = 13       /* row-plan bodies of <= 128 tiles on the 256x128 kernel's one-workgroup-per-CU form: 0 = by fill, 1 = never (default), 2 = always, 3 / 4 = by fill for GEMMs with / without split-K tails; same bits */
};
int aigv_ctx_tune(aigv_ctx* ctx, int knob, int value);
/* GEMM tile-kernel selection: mode 0 = cost model (default), 1 = always the 128x128 kernel, 2 = always the 256x256
 * phase-interleaved kernel where N % 256 == 0, 4 = always the co-resident 256x128 kernel; rate256 > 0 overrides the model's relative throughput of the 256 kernel. */
/* mode bits 4..6: 1 + v selects schedule variant v of the 256 kernel (0 = keep); bits 10..13: tile order of the 256 kernel, 0 = by weight
 * size (default), 1 = row groups, 1 + g = groups of g column tiles; bits 14..15: tile kernel of a row plan's body (AIGV_TUNE_BODY_TILE). */
int aigv_tune_gemm(int mode, double rate256);
/* Process default of AIGV_TUNE_CO_KMAX: 0 = the co-resident 256x128 kernel is never chosen by the dispatcher, else the largest K it takes. */
int aigv_tune_co_gemm(int kmax);
/* Process default of AIGV_TUNE_TAIL_SLICES / _FUSE_TAILS / _LONE_BODY / _ATTN_LEAD_KEY / _CO_KMAX (same value ranges as aigv_ctx_tune, without the -1). */
int aigv_tune_default(int knob, int value);
/* The row bands run_gemm would cut an M x N x K problem into (host logic only, no GPU): plan[0] = row tiles (x256 rows) on the
 * 256x256 kernel in whole rounds, or -1 = the whole problem in one launch of that kernel; plan[1] = row tiles on the 256x256
 * kernel with split-K, plan[2] = their K slices; plan[3] = remaining rows, plan[4] = their kernel (0 none, 1 skinny
 * weight-streaming, 2 the 128x128 kernel), plan[5] = split-K slices on the 128x128 kernel (1 = none); plan[6] = width of a right-hand
 * column band that runs on the 128x128 kernel over all rows (N = 256 j + 128: plan[0..5] then describe the first 256 j columns;
 * 0 = no column split); `plan` holds 7 ints; est_us = the model's time. */
int aigv_plan_gemm(int M, int N, int K, int epi, int* plan, double* est_us);
/* Prefill-attention kernel form (process-wide; experiments and tests): 0 = default (attention.hip: 4 waves x 32 query rows per
 * workgroup, two-deep K/V ring); 4 or 8 = that many waves per workgroup (three-deep rings lost the in-step A/B and were removed:
 * profiles/r3_attn_ring_negative.txt). */
int aigv_tune_attention(int waves);
/* Form of the decode GEMVs (process-wide; experiments and tests): 0 = default (per-shape choice in aigv_decode_step, 16-row slabs
 * in aigv_op_skinny_gemm); 1 / 2 / 4 = 16 / 8 / 4 rows of W per workgroup and slab wherever legal (R <= 16 / p, epi store /
 * residual / swiglu).  Results agree up to fp32 summation order. */
int aigv_tune_skinny(int p);

/* ---- measurement ------------------------------------------------------------------------------------ */
/* When enabled every GEMM / attention launch of the hot path is bracketed by HIP events on the launch stream. */
/* AIGV_PROF_GEMM = the bf16 tile-kernel GEMMs of the InternLM2 pass, AIGV_PROF_GEMM_VIT = those of InternViT and the mlp1 projector
 * (aigv_vit_forward / aigv_project) - two classes because the short-K InternViT shapes run at a different fraction of the MFMA peak. */
enum aigv_prof_class { AIGV_PROF_GEMM = 0, AIGV_PROF_ATTN_VIT = 1, AIGV_PROF_ATTN_LLM = 2, AIGV_PROF_SKINNY = 3,
                       AIGV_PROF_GEMM_FP8 = 4, AIGV_PROF_GEMM_VIT = 5, AIGV_PROF_COUNT = 6 };
int aigv_prof_enable(aigv_ctx* ctx, int on);
/* Synchronises the recorded events and returns (and clears) launches, total milliseconds and algorithmic
 * FLOPs (2*M*N*K; attention 4*sum(len_q*len_kv_visible)*d*heads) and bytes per class. */
int aigv_prof_read(aigv_ctx* ctx, int cls, int64_t* launches, double* total_ms, double* flops, double* bytes);

/* ---- SlowFast-R50 motion branch ---------------------------------------------------------------------
 * Replaces the reference's ``slowfast`` module + pack_pathway_output (internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:97-193):
 * frames -> [clips, 2304] motion feature, the input of motion_mlp (aigv_motion_project).  Its own handle: the branch shares nothing
 * with the scorer context but the frames tensor.  Weights arrive by their state-dict names under
 * ``slowfast_model.feature_extraction.`` (that prefix, ``feature_extraction.`` or pytorchvideo's ``blocks.`` are all accepted):
 * conv ``.weight`` [Cout, Cin, kt, kh, kw] and BatchNorm ``.weight/.bias/.running_mean/.running_var``; finalize folds the eval-mode
 * norms into the convs, packs for the kernel, uploads, and reports missing / mis-shaped tensors by name.
 *   frames: DEVICE bf16 [clips * T, 3, H, W] NCHW, clip-major (the pixel_values tensor of the scorer); feature: DEVICE bf16 [clips, 2304].
 *   T a multiple of 4 in [8, 32] (slow pathway = frames linspace(0, T-1, T/4).long()); H, W multiples of 32 in [224, 1024].
 * Parity with pytorchvideo itself cannot be pinned offline (oracle/slowfast.py restates the published architecture). */
typedef struct aigv_slowfast aigv_slowfast;
int aigv_slowfast_create(int device, int max_clips, int frames_per_clip, int height, int width, aigv_slowfast** out);
void aigv_slowfast_destroy(aigv_slowfast* sf);
int aigv_slowfast_load_weight(aigv_slowfast* sf, const char* name, const void* host_data, const int64_t* shape, int ndim, int dtype);
int aigv_slowfast_finalize(aigv_slowfast* sf);
int aigv_slowfast_forward(aigv_slowfast* sf, const void* frames_nchw_bf16, int clips, void* feature_bf16, void* stream);
double aigv_slowfast_flops_per_clip(const aigv_slowfast* sf);
/* one convolution with the branch's implicit-GEMM kernel (tests / profiling): x channels-last bf16 [B, Ti, Hi, Wi, ld_in];
 * w_packed bf16 [ceil16(Cout), Kp], k = ((dt * kh + dy) * kw + dx) * Cin + ci zero padded to Kp (multiple of 64); bias fp32 [Cout];
 * dims = {Ti, Hi, Wi, kt, kh, kw, st, sh, sw, pt, ph, pw}; out[row, c_off + c] = relu?(conv + bias + res[row, c]) */
int aigv_op_conv3d(const void* x, int ld_in, int Cin, int B, const int* dims, const void* w_packed, int Kp, const float* bias, int Cout,
                   const void* res, int ld_res, void* out, int ld_out, int c_off, int relu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIGV_AMD_H */
