/*
 * valle_hip.h -- C ABI of libvalle_hip.so: the MI355X (gfx950) kernels behind the
 * AR+NAR forward of e-c-k-e-r/vall-e.
 *
 * The reference is pure Python/PyTorch: its "FFI" for this path is the set of torch
 * operators its modules call.  Each entry point below replaces one such call site (cited
 * as reference file:line, relative to the reference repo root) and is what a maintainer
 * would bind through ctypes (see INTEGRATION.md for the stubs).
 *
 * Conventions
 *   - every function returns 0 on success, a negative VALLE_E_* code otherwise;
 *     valle_last_error() returns a static string describing the last failure of the
 *     calling thread;
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `dtype` selects the storage/compute type of activations and weights:
 *     VALLE_F32 (parity path: fp32 storage, fp32 MFMA accumulate) or VALLE_BF16
 *     (throughput path: bf16 storage, fp32 accumulate);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, nothing synchronises;
 *   - no entry point allocates or frees device memory; workspaces are caller-owned.
 *
 * Activation layout: token-packed.  A batch of B utterances with lengths T_b is one
 * row-major matrix [M = sum T_b][d]; `cu_seqlens` (int32[B+1]) holds the row offsets.
 * (The reference right-pads to [B][Tmax][d] and masks, models/base.py:112-118, :1204-1225;
 * under its all-causal masks, base.py:1233, packed and padded give the same rows.)
 *
 * Weight layouts (prepared once by the host from the reference's state-dict names):
 *   w_qkv  [3d][d]   rows = q_proj | k_proj | v_proj (arch/llama.py:159-161); inside every
 *                    q and k head the 64 rows are re-ordered (0,32,1,33,...,31,63) so the two
 *                    elements a RoPE rotation mixes (arch/llama.py:48-61) are adjacent;
 *                    q.k dot products are invariant under this common permutation.
 *   w_o    [d][d]    o_proj (arch/llama.py:162)
 *   w_gu   [2f][d]   rows interleaved gate_0, up_0, gate_1, up_1, ... (arch/llama.py:486-487)
 *   w_down [d][f]    down_proj (arch/llama.py:488)
 *   emb    [R][d]    all embedding tables + `sep` concatenated row-wise (models/base.py:420-450)
 */
#ifndef VALLE_HIP_H
#define VALLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VALLE_F32  0
#define VALLE_BF16 1

#define VALLE_OK            0
#define VALLE_E_ARG        -1   /* bad argument (shape, alignment, null pointer)            */
#define VALLE_E_UNSUPPORTED -2  /* valid in the reference, not built here (message says what) */
#define VALLE_E_HIP        -3   /* a HIP runtime call failed                                 */
#define VALLE_E_WORKSPACE  -4   /* caller workspace too small                                */

/* GEMM epilogues */
#define VALLE_EPI_STORE  0      /* C = A.W^T (+ bias)                                          */
#define VALLE_EPI_RESID  1      /* C = resid + A.W^T                                           */
#define VALLE_EPI_GEGLU  2      /* C[:, j] = gelu_erf(acc[:, 2j]) * acc[:, 2j+1]  (N/2 columns) */
#define VALLE_EPI_F32OUT 3      /* C (fp32) = A.W^T + bias, whatever `dtype` is                */
#define VALLE_EPI_GELU   6      /* C = gelu_erf(A.W^T + bias): nn.Linear + nn.GELU of the version >= 7 encoder / decoder FFNs  */
#define VALLE_EPI_RESID_BIAS 7  /* C = resid + (A.W^T + bias): the residual FFN of the AudioEncoder (models/base_v2.py:149)    */

const char* valle_last_error(void);
int valle_version(void);

/* ---------------------------------------------------------------- operator level ---- */

/* Summed embedding gather.  out[r] = sum_j emb[idx[r][j]] over the non-negative entries of
 * idx[r][0..7], added in index order in the storage dtype (the reference's Python sum()).
 * Replaces Base.inputs_to_embeddings + AudioEmbedding.forward + _join
 * (models/base.py:774-922, :147-175, :101-110): per-level prom/resp tables, text/lang/
 * rvq-level/len tables and the `sep` rows are all rows of the one concatenated table. */
int valle_embed_sum(const void* emb, const int32_t* idx, void* out, int rows, int d, int dtype, void* stream);

/* RMSNorm: out = w * cast(x * rsqrt(mean(x^2) + eps)) -- arch/llama.py:64-77 (weight applied
 * after the down-cast). */
int valle_rmsnorm(const void* x, const void* w, void* out, int rows, int d, float eps, int dtype, void* stream);

/* C[M][N] = A[M][K] . W[N][K]^T with an epilogue (nn.Linear call sites:
 * arch/llama.py:307-309, :475, :492; models/base.py:217).  K % 32 == 0.  `bias` (N, same dtype;
 * fp32 for EPI_F32OUT) and `resid` ([M][ldc]) may be NULL when the epilogue does not use them.
 * ldc = leading dimension of C (and resid), in elements. */
int valle_gemm(const void* A, const void* W, const void* bias, const void* resid, void* C,
               int M, int N, int K, int ldc, int epilogue, int dtype, void* stream);

/* q/k/v projection with RoPE in the epilogue, for passes without a KV cache (bf16, head_dim 64,
 * M >= 64, d % 64 == 0, K % 64 == 0): qkv[M][3d] = A[M][K] . w_qkv[3d][K]^T, the q and k parts
 * rotated by rope[pos[m]] -- bit-identical to valle_gemm(EPI_STORE) followed by valle_rope_kv
 * (arch/llama.py:307-309 + :48-61). */
int valle_gemm_qkv_rope(const void* A, const void* w_qkv, const float* rope, const int32_t* pos, void* qkv,
                        int M, int d, int K, void* stream);

/* RMSNorm + nn.Linear in one launch for the dense bf16 passes (arch/llama.py:64-77 ahead of :307-309 and :492):
 * C = epilogue((A . Wf^T) * rsqrt(mean(A[m]^2) + eps)), where Wf[n][k] = W[n][k] * norm_weight[k] (rounded to bf16
 * once, on the host).  K is the whole row (the normalised width).  bf16 only, M >= 64, K % 64 == 0.
 * epilogue: VALLE_EPI_STORE or VALLE_EPI_GEGLU; with rope + pos (both or neither; STORE only, N = 3 d_model, ldc = N)
 * the q and k parts are rotated as in valle_gemm_qkv_rope.  Differs from valle_rmsnorm followed by valle_gemm only in
 * where the roundings to bf16 fall (the normalised row is never rounded): same tolerance class as the decode kernels. */
int valle_gemm_normed(const void* A, const void* Wf, void* C, int M, int N, int K, int ldc, int epilogue,
                      float eps, const float* rope, const int32_t* pos, void* stream);

/* RoPE on the q and k parts of packed qkv rows [M][3d] (in place, pair-interleaved head
 * layout, see w_qkv) using the fp32 table rope[pos][Dh/2][2] = (cos, sin)
 * (arch/llama.py:82-132, :48-61), then optionally copies the rotated k and v rows into a
 * KV cache laid out [slot][H][cache_T][Dh]: row r of utterance b goes to
 * slot cache_slot[b], position cache_pos0[b] + (r - cu_seqlens[b]).  kcache may be NULL. */
int valle_rope_kv(void* qkv, const int32_t* pos, const float* rope, const int32_t* cu_seqlens,
                  int B, int M, int H, int Dh, void* kcache, void* vcache, int cache_T,
                  const int32_t* cache_slot, const int32_t* cache_pos0, int dtype, void* stream);

/* Causal self-attention over packed qkv rows [M][3d] -> out [M][d]; softmax scale 1/sqrt(Dh).
 * Replaces SDPA with the additive causal+padding mask (arch/llama.py:448-456, :689-730).
 * `causal` = 0 gives full (non-causal) attention inside each utterance (noncausal_masks). */
int valle_attn_prefill(const void* qkv, void* out, const int32_t* cu_seqlens, int B, int M, int max_T,
                       int H, int Dh, int causal, int dtype, void* stream);

/* The same attention on separate q, k, v tensors of shape (B, H, T, Dh) with arbitrary batch / head / row strides
 * (elements; the last dimension contiguous) -- the form the reference's attention plug-in point passes
 * (arch/llama.py:396-403 `fused_attn_func(q, k, v, causal=, softmax_scale=, dropout_p=)`, backend registry
 * arch/attention/__init__.py:11-198; (B, T, H, Dh) tensors of the flash_attn branch, llama.py:328-339, are the same call with
 * the head and row strides swapped).  *_strides = {batch, head, row}.  seqlens (device int32 [B], optional): utterance b
 * attends over its first seqlens[b] rows only and rows at or beyond seqlens[b] of `out` are not written -- the padding the
 * reference's own fused kernel leaves unmasked (arch/attention/fused.py has no mask argument).  bf16, Dh == 64. */
int valle_attn_prefill_hnd(const void* q, const void* k, const void* v, void* out, int B, int H, int T, int Dh,
                           const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                           const int64_t* out_strides, int causal, float softmax_scale, const int32_t* seqlens,
                           int dtype, void* stream);

/* One-query-per-utterance attention against the KV cache (the KV-cache restatement of the
 * reference's full recompute, models/base.py:521 use_cache=False).  q rows come from qkv [B][3d];
 * utterance b reads ctx_len[b] cached positions of slot cache_slot[b]. */
int valle_attn_decode(const void* qkv, const void* kcache, const void* vcache, const int32_t* ctx_len,
                      const int32_t* cache_slot, void* out, int B, int H, int Dh, int cache_T,
                      int dtype, void* stream);

/* Sampler parameters -- Base.sample's kwargs (models/base.py:1288-1309). */
typedef struct {
    float temperature;              /* <= 0: argmax                                          */
    int   top_k;                    /* <= 0: off                                             */
    float top_p;                    /* >= 1: off                                             */
    float min_p;                    /* <= 0: off                                             */
    float top_no;                   /* <= 0: off; reference default 1.0 (base.py:1307)       */
    float repetition_penalty;       /* 1.0: off                                              */
    float repetition_penalty_decay;
    float length_penalty;           /* 0.0: off; AR rows only                                */
    int   stop_token;
    int   rep_pen_passes;           /* 1; 2 reproduces NAR-len's second in-place pass        */
    uint64_t seed;                  /* Philox seed for temperature > 0                       */
    uint64_t offset;                /* Philox offset (advance per call)                      */
    /* dynamic temperature (samplers.py:128-141), on when |temperature - min_temperature| >= 0.001
     * (base.py:1389): the reference computes it in Python doubles, hence the double copies        */
    double temperature64;
    double min_temperature64;       /* set equal to temperature64 for "off" (base.py:1313-1314)     */
    /* DRY (samplers.py:248-275), on when dry_multiplier > 0 and the row has previous tokens        */
    double dry_multiplier;
    double dry_base;
    int    dry_allowed_length;      /* >= 2 makes the reference's loop a no-op; kept for fidelity   */
    int    filter_passes;           /* 0 / 1: top-k / top-p once; 2 reproduces NAR-len's second in-place pass
                                     * (only when min_p is off: min-p filters out of place, samplers.py:90)   */
    int    score_unscaled;          /* != 0: the version >= 7 sampler's score (models/base_v2.py:1398-1410) -- softmax of the
                                     * processed logits BEFORE the temperature, at the drawn token; 0: after it           */
} valle_sample_params;

/* Sampler chain on fp32 logits [rows][ld] (first `vocab` columns valid): repetition penalty
 * (samplers.py:16-50) -> length penalty (:56-61, AR rows) -> min-p (:74-90) -> top-k/top-p
 * (:93-125) -> dynamic temperature (:128-141) or /temperature -> top-n-sigma (:164-172) -> DRY
 * (:248-275) -> argmax or categorical
 * (base.py:1419-1422) -> score = softmax(processed)[token] (base.py:1425-1428).
 * Row r belongs to utterance row_item[r]; its previous tokens are prev[prev_off[i] ..
 * prev_off[i] + prev_len[i]).  nar_mode != 0: row r (row_in_item[r]) is penalised only by
 * previous tokens at positions < row_in_item[r] (samplers.py:38-39).  AR rows with exactly one
 * previous token get no penalty (the reference classifies them as NAR, samplers.py:22).
 * Outputs: ids[rows] (int32), scores[rows] (fp32, may be NULL). */
int valle_sample(const float* logits, int ld, int vocab, int rows, const int32_t* row_item,
                 const int32_t* row_in_item, const int32_t* prev, const int32_t* prev_off,
                 const int32_t* prev_len, int nar_mode, const valle_sample_params* params,
                 int32_t* ids, float* scores, void* stream);

/* Mirostat state of one utterance (samplers.py:197-238; the dict the reference threads through Base.sample(mirostat=[...])):
 * in: n, tau, eta and the running values (max_surprise = 2 tau, the others 0 on a fresh state); out: the running values,
 * the drawn token and k = the size of the candidate set of this draw (-1: the reference's compute_k divides by zero on
 * these logits -- fewer than 101 columns with non-zero probability -- and the state is left as it was). */
typedef struct {
    double n, tau, eta, max_surprise, error_surprise, running_total_surprise, index_surprise;
    int32_t token, k;
} valle_mirostat_state;

/* valle_sample whose final pick is the mirostat draw (models/base.py:1404-1407): after the sampler chain, row r (one row
 * per utterance: row_item == NULL or rows == utterances) is sorted, k = compute_k(softmax, n, max_surprise) + 1, the token is
 * drawn from the softmax of the k largest logits (Philox stream of params), states[r] is updated.  vocab >= 101. */
int valle_sample_mirostat(const float* logits, int ld, int vocab, int rows, const int32_t* prev, const int32_t* prev_off,
                          const int32_t* prev_len, const valle_sample_params* params, valle_mirostat_state* states,
                          int32_t* ids, float* scores, void* stream);

/* Classifier-free guidance (samplers.py:177-191) on fp32 logits, per utterance i over its
 * rows [row_off[i], row_off[i+1]): s = neg + (pos - neg) * strength; if rescale > 0,
 * s *= rescale * std(pos)/std(s) + (1 - rescale) with std the unbiased std over ALL
 * rows x vocab of that utterance.  Written in place into `pos`.  One workgroup per utterance. */
int valle_cfg_combine(float* pos, const float* neg, int ld, int vocab, const int32_t* row_off, int B,
                      float strength, float rescale, void* stream);

/* valle_cfg_combine (call sites models/ar_nar.py:433, :574) with the work of an utterance spread over 8 workgroups (NAR-len: 64 utterances x 450 rows left 3/4 of
 * the chip idle on one workgroup each): two launches, the row chunks' partial sums (double) go through `workspace`
 * (valle_cfg_combine_workspace_bytes(B), 8-byte aligned) and are added up in chunk order by every workgroup of the utterance.
 * Same formula; the sums are added in another order than valle_cfg_combine's (fp64: the factor agrees to ~1e-7). */
size_t valle_cfg_combine_workspace_bytes(int B);
int valle_cfg_combine_ws(float* pos, const float* neg, int ld, int vocab, const int32_t* row_off, int B,
                         float strength, float rescale, void* workspace, size_t workspace_bytes, void* stream);

/* The guidance of the AR loop (models/ar_nar.py:785-800 -> samplers.py:177-191): utterance i's rows are
 * [row_start[i], row_start[i] + row_count[i]) of two READ-ONLY row matrices (the per-step logit histories of the
 * conditional and the null pass; the statistics run over all of them) and only the combination of its LAST row is written,
 * to out[i][0..vocab) (ld_out floats per row). */
int valle_cfg_combine_last(const float* pos, const float* neg, int ld, int vocab, const int32_t* row_start,
                           const int32_t* row_count, int B, float strength, float rescale, float* out, int ld_out,
                           void* stream);

/* One host-driven AR step (the prefill's row, every step of the guidance loop): valle_sample in AR mode on logits [B][ld]
 * with prev = seq[b][:seq_len[b]], then the append of models/ar_nar.py:837-848 on the device: seq[b][seq_len[b]] = token,
 * seq_len[b] += 1, stopped[b] |= (token == stop_token), and ctx_len[b] += 1 / ctx_len2[b] += 1 for the caches that ran
 * the token through the stack (either may be NULL).  ids [B] and scores [B] (may be NULL) as valle_sample. */
int valle_sample_append(const float* logits, int ld, int vocab, int B, const valle_sample_params* params,
                        int32_t* seq, int seq_cap, int32_t* seq_len, int32_t* ctx_len, int32_t* ctx_len2,
                        int32_t* stopped, int32_t* ids, float* scores, void* stream);

/* Beam search step (models/base.py:1412-1415 with samplers.py:148-160; loop at models/ar_nar.py:821-842): the sampler chain
 * of valle_sample in AR mode (prev = seq[r][:seq_len[r]]) on logits [rows][ld], then the beam_width best (row, token) pairs
 * of the flattened processed logits, best first.  Out: ids[i] = token of candidate i, scores[i] = its processed logit (may be
 * NULL), forced [beam_width][ld] = -inf except 0 at ids[i] -- a greedy valle_sample_append on it appends candidate i to
 * sequence i, as the reference does whichever row the candidate came from.  proc [rows][ld]: scratch (processed logits). */
int valle_beam_pick(const float* logits, int ld, int vocab, int rows, int beam_width,
                    const valle_sample_params* params, const int32_t* seq, int seq_cap, const int32_t* seq_len,
                    float* proc, float* forced, int32_t* ids, float* scores, void* stream);

/* Integer table patch: dst[rows[i]][c0 + c] = src[i][s0 + c] + add for i < n, c < ncols (rows == NULL: row i; dst int32 or,
 * dst_is_i64 != 0, int64).  The host plans the gather tables of a dense pass before the codes exist (Base.inputs,
 * models/base.py:560-735); the device fills in the codes as they are sampled (the torch.cat / tensor rebuilds of
 * models/ar_nar.py:523-525, :575-584, :363). */
int valle_scatter_cols(void* dst, int dst_is_i64, int ld_dst, const int32_t* rows, const int32_t* src, int ld_src,
                       int n, int c0, int s0, int ncols, int add, void* stream);

/* End of a NAR-len demasking step (models/ar_nar.py:457-462): codes[j] = ids[j] where is_masked[j]; scores[j] = sc[j], or
 * with masked_only (sampling_scores_masked_only) sc[j] where is_masked[j] and 1.0 elsewhere. */
int valle_narlen_commit(int32_t* codes, float* scores, const uint8_t* is_masked, const int32_t* ids, const float* sc,
                        int n, int masked_only, void* stream);

/* NAR-len masking step (models/ar_nar.py:358-365): for utterance i with scores
 * s[off[i] .. off[i+1]), sets codes[j] = mask_token at the k[i] lowest-score positions
 * (ties broken by lower index) and writes is_masked[j] = (codes[j] == mask_token).
 * max_len = the longest utterance (host-side bound, <= 8192). */
int valle_mask_lowest(const float* scores, int32_t* codes, uint8_t* is_masked, const int32_t* off,
                      const int32_t* k, int B, int max_len, int mask_token, void* stream);

/* dst[0 .. n_words) = value (32-bit words), ordered on `stream` (hipMemsetD32Async, no compute kernel).  No counterpart in the
 * reference (its torch.zeros / torch.full / torch.ones calls, e.g. models/ar_nar.py:337-339, :696): the host mirror
 * initialises its device tables with it so that the tensor carrier launches no kernel of its own on the path. */
int valle_fill32(void* dst, uint32_t value, size_t n_words, void* stream);

/* ------------------------------------------------------------------- model level ---- */

typedef struct {
    int d_model, n_heads, n_layers, d_ff;
    int dtype;
    float rms_eps;
    int max_pos;            /* rows of the rope table */
} valle_config;

typedef struct {
    const void *ln1, *w_qkv, *w_o, *ln2, *w_gu, *w_down;
    /* bf16 decode copies in streaming order (may be NULL; required for the fast VALLE_BF16 decode path): the same
     * matrices tiled as [N/16][K/64][2][64 lanes][8]: lane l = fq*16 + fr of tile (nt, ks, half) holds
     * W[nt*16 + fr][ks*64 + half*32 + fq*8 .. +8], i.e. exactly the MFMA A fragment, so one wave-instruction
     * reads 1 KiB of contiguous memory and a wave's whole slice is contiguous.
     * pn_qkv / pn_gu: w_qkv / w_gu with the preceding RMSNorm weight multiplied in (W[n][k] * ln[k], rounded to
     * bf16 once): the decode kernels apply 1/rms(x) to the accumulator instead of normalising x first.
     * p_o / p_down are the streaming-order image of the NARROW-TILE row order of w_o / w_down, a [4 N][K/4] matrix
     * W4 with W4[(n/4)*16 + q*4 + n%4][k] = W[n][q*(K/4) + k], q = 0..3 (vall-e_amd/models/base.py narrow_tile_rows):
     * a 16-row tile then holds 4 output columns x the 4 quarters of K, and N/4 workgroups each own 4 columns. */
    const void *p_o, *p_down, *pn_qkv, *pn_gu;
    /* bf16 dense-pass copies (may be NULL): w_qkv / w_gu row-major with the preceding RMSNorm weight multiplied in, as
     * pn_qkv / pn_gu but not tiled; with them valle_forward_dense runs valle_gemm_normed and no separate RMSNorm. */
    const void *f_qkv, *f_gu;
} valle_layer_weights;

typedef struct valle_model valle_model;

/* Builds a host-side descriptor holding the device pointers (no device allocation, no copy).
 * heads: n_cls classifier heads (models/base.py:179-231), head k = (w [n_k][d], b [n_k] fp32 or NULL);
 * head_pn_host: optional streaming-order copies of the head weights (rows padded to a multiple of 16) with the
 * final norm weight multiplied in (see valle_layer_weights.pn_*). */
int valle_model_create(const valle_config* cfg, const void* emb, const float* rope,
                       const valle_layer_weights* layers_host, const void* final_norm,
                       int n_cls, const void* const* head_w_host, const float* const* head_b_host,
                       const int32_t* head_n_host, const void* const* head_pn_host, valle_model** out);
void valle_model_destroy(valle_model* m);

/* Workspace bytes valle_forward_dense needs for M packed rows and n_out output rows. */
size_t valle_dense_workspace_bytes(const valle_model* m, int M, int n_out);

/* One full (non-incremental) pass: embedding sums -> n_layers x {RMSNorm, QKV, RoPE, causal
 * attention, o_proj, RMSNorm, GEGLU MLP} -> final RMSNorm -> classifier `head` on the rows
 * listed in out_rows (packed row ids) -> fp32 logits [n_out][ld_logits].
 * = Base.forward (models/base.py:1184-1278) for one homogeneous classifier level.
 * If kcache != NULL the rotated K and V of every layer are also written to the caches
 * ([n_layers][slots][H][cache_T][Dh], layer stride cache_layer_stride elements): the AR prefill.
 * hidden_out (optional, [n_out][d] in dtype) receives the normalised hidden rows. */
int valle_forward_dense(valle_model* m, const int32_t* emb_idx, const int32_t* pos,
                        const int32_t* cu_seqlens, int B, int M, int max_T, int causal,
                        void* kcache, void* vcache, int cache_T, size_t cache_layer_stride,
                        const int32_t* cache_slot, const int32_t* cache_pos0,
                        const int32_t* out_rows, int n_out, int head, float* logits, int ld_logits,
                        void* hidden_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------- version >= 7 model ("v2": models/base_v2.py) ---- */

/* valle_gemm for FEW rows in bf16 (a decode step's encoder / decoder linears: 16 .. a few hundred rows): one workgroup per
 * 16 output columns, K split over its waves, row-major weights streamed as they lie.  Same arguments and epilogues as
 * valle_gemm (STORE, RESID, F32OUT, GELU, RESID_BIAS; not GEGLU); K must be a multiple of 128.  An output's K sum is the
 * sum of four partial sums, so the last bits can differ from valle_gemm's. */
int valle_gemm_skinny(const void* A, const void* W, const void* bias, const void* resid, void* C, int M, int N, int K, int ldc,
                      int epilogue, int dtype, void* stream);

/* G products of ONE shape in one launch: group g reads A + g * a_group_stride, W + g * w_group_stride, bias + g *
 * bias_group_stride, resid + g * resid_group_stride and writes C + g * c_group_stride (strides in ELEMENTS of the operand's
 * own type: bf16, or fp32 for the bias and C of EPI_F32OUT; A / W strides multiples of 8, resid / C of 4; a stride of 0
 * shares the operand).  The AudioDecoder's per-level FFNs and heads (models/base_v2.py:199-225: `self.head[level]` applied
 * level by level) as one launch per linear instead of one per level and linear; every group's arithmetic is valle_gemm_skinny's. */
int valle_gemm_skinny_grouped(const void* A, size_t a_group_stride, const void* W, size_t w_group_stride, const void* bias,
                              size_t bias_group_stride, const void* resid, size_t resid_group_stride, void* C,
                              size_t c_group_stride, int G, int M, int N, int K, int ldc, int epilogue, int dtype, void* stream);

/* nn.LayerNorm over the last dimension (with weight and bias, eps inside the square root): the AudioDecoder's per-level
 * norm (models/base_v2.py:184).  x, w, b, out in dtype. */
int valle_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int d, float eps, int dtype, void* stream);

/* G LayerNorms of the same rows: w, b = [G][d], out = [G][rows][d] (the per-level norms that open the AudioDecoder's heads,
 * models/base_v2.py:203-206, in one launch). */
int valle_layernorm_grouped(const void* x, const void* w, const void* b, void* out, int rows, int d, int G, float eps, int dtype,
                            void* stream);

/* First half of AudioEncoder.forward (models/base_v2.py:138-147) for n code frames of L <= 32 levels: out[(i * L + l)][:] =
 * LayerNorm(table[row0[l] + codes[l * ld_codes + i]][:] + pos[l][:]) -- codes level-major int32 on the device, table = the
 * concatenated embedding table (row0[l] = first row of level l's table, n_rows[l] = its row count: HOST arrays of L ints;
 * a code outside its table is clamped into it), pos [L][d], norm weight / bias [d]; src_scratch: n * L int32 on the
 * device (the gathered row ids). */
int valle_v2_encode_rows(const void* table, const int32_t* row0_host, const int32_t* n_rows_host, int L, const int32_t* codes,
                         int ld_codes, int n, const void* pos, const void* norm_w, const void* norm_b, void* out,
                         int32_t* src_scratch, int d, float eps, int dtype, void* stream);

/* Second half (models/base_v2.py:153-157): out[dst_rows[i]][:] = sum_l lw[l] * y[(i * L + l)][:] (lw = softmax of the level
 * weights, fp32 [L] on the device; products rounded to dtype as the reference's elementwise multiply is); dst_rows == NULL:
 * row i.  This is also how encoded frames are placed into their rows of the packed sequence. */
int valle_v2_level_sum(const void* y, const float* lw, int L, int n, const int32_t* dst_rows, void* out, int d, int dtype,
                       void* stream);

/* Classifier-free guidance on (levels, rows, vocab) logits (samplers.py:177-191 with a 3-D input): for utterance i, level l
 * and vocabulary column v, s = neg + (pos - neg) * strength over the rows [row_off[i], row_off[i+1]); with rescale > 0,
 * s *= rescale * std_rows(pos) / std_rows(s) + (1 - rescale), std = the unbiased std over the ROWS of that column.
 * pos / neg: [L][rows_total][ld] fp32, written in place into pos. */
int valle_cfg_combine_cols(float* pos, const float* neg, int ld, int vocab, int L, int rows_total, const int32_t* row_off,
                           int B, float strength, float rescale, void* stream);

/* valle_forward_dense for a model whose embeddings are computed outside (the version >= 7 AudioEncoder): x_in [M][d] in
 * dtype replaces the embedding gather, the stack runs as in valle_forward_dense, and hidden_out [n_out][d] receives the
 * out_rows of the last layer's output -- normalised by the final RMSNorm when final_norm != 0, raw otherwise
 * (per_level_normalization moves that norm into the decoder, models/base_v2.py:495).  No head is applied.
 * kv_lens (optional, [B], 1 <= kv_lens[b] <= rows of b): only the first kv_lens[b] rows of utterance b are keys; the rows
 * behind them are queries alone -- the padding rows of the reference's right-padded batch, whose last one the version >= 7
 * duration head reads (models/base_v2.py:1298-1300: logit[..., -1, :] of the PADDED batch). */
int valle_forward_layers(valle_model* m, const void* x_in, const int32_t* pos, const int32_t* cu_seqlens, int B, int M,
                         int max_T, int causal, void* kcache, void* vcache, int cache_T, size_t cache_layer_stride,
                         const int32_t* cache_slot, const int32_t* cache_pos0, const int32_t* out_rows, int n_out,
                         int final_norm, const int32_t* kv_lens, void* hidden_out, void* workspace, size_t workspace_bytes,
                         void* stream);

/* valle_forward_dense that also returns the LAST layer's raw attention products of chosen rows: scores[r][h][j] =
 * q[score_rows[r]][h] . k[j][h] / sqrt(head_dim) for every row j of that row's utterance, fp32 [n_score][H][ld_scores]
 * (ld_scores >= max_T) -- what the reference's attention hands back under output_attentions (models/arch/llama.py:429,
 * :469-477: the scaled products BEFORE mask and softmax; for a bf16 model the product is rounded to bf16 first, as
 * torch.matmul's output is).  The entropix sampler reads the newest row of the last layer (samplers.py:294, :482);
 * the earlier layers' maps are not materialised. */
int valle_forward_dense_scores(valle_model* m, const int32_t* emb_idx, const int32_t* pos,
                               const int32_t* cu_seqlens, int B, int M, int max_T, int causal,
                               void* kcache, void* vcache, int cache_T, size_t cache_layer_stride,
                               const int32_t* cache_slot, const int32_t* cache_pos0,
                               const int32_t* out_rows, int n_out, int head, float* logits, int ld_logits,
                               void* hidden_out, void* workspace, size_t workspace_bytes, void* stream,
                               const int32_t* score_rows, int n_score, float* scores, int ld_scores);

/* Device-resident AR state (one per batch). All arrays are caller-allocated device memory. */
typedef struct {
    int B;
    int max_steps;              /* capacity of seq (tokens per utterance)                     */
    int32_t* seq;               /* [B][max_steps] generated tokens (level 0)                  */
    int32_t* seq_len;           /* [B] tokens generated so far                                */
    int32_t* ctx_len;           /* [B] cached positions per utterance                         */
    int32_t* resp_pos0;         /* [B] RoPE position of response token 0 (0 unless prefixed)  */
    int32_t* cache_slot;        /* [B]                                                        */
    int32_t* stopped;           /* [B] 1 once the stop token was sampled                      */
    float*   scores;            /* [B] score of the last sampled token (may be NULL)          */
    void*    kcache;            /* [n_layers][slots][H][cache_T][Dh]                          */
    void*    vcache;
    int      cache_T;
    size_t   cache_layer_stride;
    int32_t  resp_emb_row0;     /* row of resps_emb[0][0] in the concatenated table           */
    int      head;              /* classifier index (AR:0:0 = 0)                              */
    int      vocab;             /* columns of that head                                       */
    int      early_exit;        /* 1: an utterance whose stop token was sampled leaves the step -- no KV append, no KV
                                 * read (its attention workgroups return), nothing appended to seq, seq_len frozen and
                                 * ctx_len[b] stored as ~ctx_len (negative: the mark the decode kernels test).  The reference keeps stepping finished utterances until all have stopped
                                 * (models/ar_nar.py:837-848; its own to-do, docs/README.md:85); what it appends behind a
                                 * stop token is cut off again (:870-872), so results are identical -- except for
                                 * refine_on_stop (:874-883), which re-reads those rows: leave 0 there                  */
    int      max_ctx;           /* HOST-side upper bound on ctx_len[b] over the batch at the time of the call
                                 * (required, > 0): valle_ar_decode refuses n_steps that would write past
                                 * cache_T or read past the RoPE table -- the kernels do not check        */
} valle_ar_state;

/* One KV-cache decode step of the stack for embeddings computed outside (the version >= 7 AR loop, models/ar_nar_v2.py:504-572;
 * the reference recomputes the whole sequence per step, use_cache=False at models/base_v2.py:539): x_in [B][d] in dtype = the
 * AudioEncoder's row of every utterance's newest frame, pos_in [B] its position id.  Of the state, B / ctx_len / cache_slot /
 * kcache / vcache / cache_T / cache_layer_stride / max_ctx are read; the row's key and value are appended and ctx_len[b] is
 * incremented.  hidden_out [B][d]: the last hidden state, through the final RMSNorm when final_norm != 0. */
int valle_ar_decode_layers(valle_model* m, const valle_ar_state* st, const void* x_in, const int32_t* pos_in, void* hidden_out,
                           int final_norm, void* workspace, size_t workspace_bytes, void* stream);


size_t valle_decode_workspace_bytes(const valle_model* m, int B);

/* n_steps AR decode steps, enqueued back to back on `stream` with no host synchronisation:
 * each step embeds every utterance's newest token (seq[b][seq_len[b]-1]), runs the stack on
 * that one row per utterance against the KV cache, applies head `st->head`, samples
 * (valle_sample semantics, AR mode, prev = seq[b][:seq_len[b]]), appends the token and
 * updates seq_len / ctx_len / stopped on the device.  = one iteration of the loop at
 * models/ar_nar.py:748-848 per step, with the recompute replaced by the cache.
 * logits_out (optional, [n_steps][B][ld_logits] fp32) keeps every step's raw logits.
 * sp == NULL (n_steps must be 1): the step stops after the head; the caller combines/samples the
 * logits itself and appends the token (classifier-free-guidance path, ar_nar.py:785-800). */
int valle_ar_decode(valle_model* m, const valle_ar_state* st, const valle_sample_params* sp,
                    int n_steps, float* logits_out, int ld_logits,
                    void* workspace, size_t workspace_bytes, void* stream);

/* n_steps steps of the classifier-free-guidance AR loop (models/ar_nar.py:785-808) with no host round trip: per step, one
 * logits-only decode step on the conditional state and one on the null state (they share seq / seq_len / stopped and
 * differ in the caches, resp_pos0 and ctx_len), whose logits land in row hist_row0 + step of every utterance's
 * [hist_rows][ld] block of hist_pos / hist_neg; valle_cfg_combine_last over rows [row_start[b], row_start[b] +
 * row_count0[b] + step) -> last + i * last_step_stride ([B][ld]; stride 0 reuses one buffer); valle_sample_append with sp
 * (Philox offset sp->offset + i) bumping both ctx_len.  first_step >= 1 is the index of the first of these steps in the
 * whole loop (step 0 samples from the prefill's rows, host side).  The caller checks `stopped` between calls. */
int valle_ar_decode_cfg(valle_model* m, const valle_ar_state* st_pos, const valle_ar_state* st_neg,
                        const valle_sample_params* sp, int n_steps, int first_step,
                        float* hist_pos, float* hist_neg, int ld, int hist_rows, int hist_row0,
                        const int32_t* row_start, const int32_t* row_count0, float strength, float rescale,
                        float* last, size_t last_step_stride,
                        void* workspace, size_t workspace_bytes, void* stream);

/* One logits-only decode step (valle_ar_decode with sp == NULL, n_steps 1) that also returns the last layer's raw attention
 * products of every utterance's newest row against its ctx_len[b] + 1 cached keys: scores [B][H][ld_scores] fp32,
 * ld_scores >= max_ctx + 1 (same values as valle_forward_dense_scores gives for that row of a recompute).
 * = Base.forward(..., output_attentions=True) of the entropix AR loop (models/ar_nar.py:778-783). */
int valle_ar_decode_scores(valle_model* m, const valle_ar_state* st, float* logits_out, int ld_logits,
                           float* scores, int ld_scores, void* workspace, size_t workspace_bytes, void* stream);

/* calculate_entropix_metrics (samplers.py:281-321) as sample_entropix calls it (:482): per utterance b, on the logits row
 * logit_row[b] (NULL: row b) of logits [..][ld] and its scores [b][H][ld_scores] over n_keys[b] keys:
 * out[b][0..5] = logits entropy, logits varentropy (bits), attention entropy (mean over heads), its unbiased variance over
 * heads (NaN -> 0), agreement, interaction strength; out is [B][8] fp32 (6, 7 unused). */
int valle_entropix_metrics(const float* logits, int ld, int vocab, const int32_t* logit_row, const float* scores,
                           int ld_scores, int H, const int32_t* n_keys, int B, float* out, void* stream);

/* acc[b][v] (+)= sum over rows [row_off[b], row_off[b+1]) of log_softmax(logits[row])[v] (accumulate = 0 overwrites): the
 * adaptive entropix branch scores a draw by the sum over ALL rows of the sequence of log_softmax[row][token]
 * (samplers.py:516, :527-545); the decode loop keeps that sum per utterance and adds the newest row each step. */
int valle_logsoftmax_colsum(const float* logits, int ld, int vocab, const int32_t* row_off, int B, float* acc,
                            int ld_acc, int accumulate, void* stream);

/* _sample_entropix (samplers.py:419-462) per utterance on logits row logit_row[b]: probabilities at temperature[b], the
 * top_k[b] largest in ascending order, the top_p[b] cut of :452, n_draws[b] <= max_draws draws argmax(p / q) with
 * q ~ Exponential(1) (noise == NULL: Philox of (seed, offset); else q = noise[(b * max_draws + draw) * ld_noise + i] for
 * ascending position i), and of those draws the one with the largest colsum[b][token] (first on ties; samplers.py:547-550;
 * colsum NULL: the first draw).  temperature / top_k / top_p arrive clamped as :433-436 does (the host does the branch
 * arithmetic of :489-523).  ids [B] the winner, draws [B][max_draws] every draw's token, forced (optional) [B][ld_forced] =
 * -inf except 0 at the winner -- a greedy valle_sample_append on it appends it. */
int valle_entropix_pick(const float* logits, int ld, int vocab, const int32_t* logit_row, int B,
                        const float* temperature, const int32_t* top_k, const float* top_p, const int32_t* n_draws,
                        int max_draws, const float* colsum, int ld_colsum, const float* noise, int ld_noise,
                        uint64_t seed, uint64_t offset, int32_t* ids, int32_t* draws, float* forced, int ld_forced,
                        void* stream);

/* valle_ar_decode for callers that cannot allocate device memory (code under HIP graph capture, serving loops and
 * harnesses that forbid hipMalloc) and so cannot bring a workspace.  NO COUNTERPART IN THE REFERENCE: the reference has
 * no such entry; it exists for those callers only.  Same arguments, results and errors as valle_ar_decode without the
 * workspace pair: the step's activations live in a static device array of valle_noalloc_scratch_bytes() = 2 MiB inside
 * the library (every GPU the library runs on holds its own copy until the process ends).  Batches whose
 * valle_decode_workspace_bytes() exceeds that are refused with VALLE_E_WORKSPACE (32 utterances at d_model 1024 in bf16
 * need 1.3 MB).  Calls no hipMalloc / hipFree; the array's device address on a GPU is taken in valle_model_create (with that
 * GPU current, outside any stream capture), and a stream of another GPU than the model's is refused.  All calls on one GPU share the array: calls that overlap on one GPU
 * (two streams, two threads) race -- serialise them or use valle_ar_decode with separate workspaces. */
size_t valle_noalloc_scratch_bytes(void);
int valle_ar_decode_noalloc(valle_model* m, const valle_ar_state* st, const valle_sample_params* sp,
                            int n_steps, float* logits_out, int ld_logits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VALLE_HIP_H */
