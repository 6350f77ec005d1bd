// Model-level orchestration: one dense pass (Base.forward, models/base.py:1184-1278) and the
// KV-cache AR decode loop (models/ar_nar.py:748-848).  Host code only enqueues kernels.
#include "common.h"
#include <mutex>

extern "C" int valle_attn_prefill(const void* qkv, void* out, const int32_t* cu_seqlens, int B, int M, int max_T,
                                  int H, int Dh, int causal, int dtype, void* stream) {
    if (dtype == VALLE_BF16 && Dh == 64 && max_T > 0 && qkv && out && cu_seqlens && M > 0)
        return valle_attn_flash_bf16(qkv, out, cu_seqlens, B, max_T, H, causal, (hipStream_t)stream);
    return valle_attn_prefill_rowwise(qkv, out, cu_seqlens, B, M, H, Dh, causal, dtype, stream);
}

// ------------------------------------------------------------------ static scratch of valle_ar_decode_noalloc (below)
#define VALLE_NOALLOC_SCRATCH_BYTES (2u << 20)
static __device__ __align__(256) unsigned char scratch[VALLE_NOALLOC_SCRATCH_BYTES];
#define VALLE_MAX_DEVICES 64
static void* g_scratch_ptr[VALLE_MAX_DEVICES];       // per device: the device address of that GPU's copy of `scratch`
static std::mutex g_scratch_mu;

// Once per device, with that device current: the array's address ON THIS GPU.  Called from valle_model_create (the
// library's set-up call: no stream capture can be active there) and, for a device no model was created on, from the
// entry point's first call.
static int scratch_resolve(int dev, void** out) {
    VALLE_REQUIRE(dev >= 0 && dev < VALLE_MAX_DEVICES, VALLE_E_UNSUPPORTED, "valle_ar_decode_noalloc: device %d", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        VALLE_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        VALLE_REQUIRE(p != nullptr, VALLE_E_HIP, "valle_ar_decode_noalloc: no device address for the scratch array");
        g_scratch_ptr[dev] = p;
    }
    if (out) *out = g_scratch_ptr[dev];
    return VALLE_OK;
}

extern "C" void valle_model_destroy(valle_model* m);
extern "C" int valle_model_create(const valle_config* cfg, const void* emb, const float* rope,
                                  const valle_layer_weights* layers_host, const void* final_norm,
                                  int n_cls, const void* const* head_w_host, const float* const* head_b_host,
                                  const int32_t* head_n_host, const void* const* head_pn_host, valle_model** out) {
    VALLE_REQUIRE(cfg && emb && rope && layers_host && final_norm && out, VALLE_E_ARG, "valle_model_create: null pointer");
    VALLE_REQUIRE(cfg->n_heads > 0 && cfg->d_model % cfg->n_heads == 0 && cfg->d_model / cfg->n_heads == 64,
                  VALLE_E_UNSUPPORTED, "valle_model_create: head_dim must be 64 (d_model %d, heads %d)", cfg->d_model, cfg->n_heads);
    VALLE_REQUIRE(cfg->d_model % 32 == 0 && cfg->d_ff % 32 == 0, VALLE_E_ARG, "valle_model_create: d_model/d_ff must be multiples of 32");
    VALLE_REQUIRE(cfg->dtype == VALLE_F32 || cfg->dtype == VALLE_BF16, VALLE_E_ARG, "valle_model_create: unknown dtype %d", cfg->dtype);
    VALLE_REQUIRE(cfg->n_layers > 0 && n_cls >= 0 && cfg->max_pos > 0, VALLE_E_ARG,
                  "valle_model_create: n_layers %d, n_cls %d, max_pos %d", cfg->n_layers, n_cls, cfg->max_pos);
    VALLE_REQUIRE(n_cls == 0 || (head_w_host && head_n_host), VALLE_E_ARG, "valle_model_create: head tables missing");
    valle_model* m = (valle_model*)calloc(1, sizeof(valle_model));
    VALLE_REQUIRE(m, VALLE_E_ARG, "valle_model_create: out of host memory");
    m->cfg = *cfg;
    m->emb = emb; m->rope = rope; m->final_norm = final_norm; m->n_cls = n_cls;
    m->layers = (valle_layer_weights*)malloc(sizeof(valle_layer_weights) * cfg->n_layers);
    m->head_w = (const void**)malloc(sizeof(void*) * (n_cls > 0 ? n_cls : 1));
    m->head_b = (const float**)malloc(sizeof(float*) * (n_cls > 0 ? n_cls : 1));
    m->head_n = (int32_t*)malloc(sizeof(int32_t) * (n_cls > 0 ? n_cls : 1));
    m->head_pn = (const void**)malloc(sizeof(void*) * (n_cls > 0 ? n_cls : 1));
    if (!m->layers || !m->head_w || !m->head_b || !m->head_n || !m->head_pn) {
        valle_model_destroy(m);
        valle_set_error("valle_model_create: out of host memory");
        return VALLE_E_ARG;
    }
    memcpy(m->layers, layers_host, sizeof(valle_layer_weights) * cfg->n_layers);
    for (int k = 0; k < n_cls; ++k) {
        m->head_w[k] = head_w_host[k];
        m->head_b[k] = head_b_host ? head_b_host[k] : nullptr;
        m->head_n[k] = head_n_host[k];
        m->head_pn[k] = head_pn_host ? head_pn_host[k] : nullptr;
    }
    // the no-allocation entry point's scratch address on this GPU is taken here, outside any stream capture
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) {
        const int rc = scratch_resolve(dev, nullptr);
        if (rc != VALLE_OK) { valle_model_destroy(m); return rc; }
        m->device = dev;
    } else {
        m->device = -1;
    }
    *out = m;
    return VALLE_OK;
}

extern "C" void valle_model_destroy(valle_model* m) {
    if (!m) return;
    free(m->layers); free((void*)m->head_w); free((void*)m->head_b); free(m->head_n); free((void*)m->head_pn); free(m);
}

// ------------------------------------------------------------------ buffer initialisation without a compute kernel
// 32-bit fill on the stream (hipMemsetD32Async): the host mirror initialises its device tables with it, so that the
// allocator carrier (torch) launches no fill / arange kernel of its own on the product path.
extern "C" int valle_fill32(void* dst, uint32_t value, size_t n_words, void* stream) {
    VALLE_REQUIRE(dst || n_words == 0, VALLE_E_ARG, "valle_fill32: null pointer");
    if (n_words == 0) return VALLE_OK;
    VALLE_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)dst, (int)value, n_words, (hipStream_t)stream));
    return VALLE_OK;
}

// ------------------------------------------------------------------ small helpers
template <typename T>
__global__ __launch_bounds__(128) void gather_rows_kernel(const T* __restrict__ x, const int32_t* __restrict__ rows,
                                                          T* __restrict__ out, int d) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int r = blockIdx.x, src = rows[r];
    for (int c = threadIdx.x * VEC; c < d; c += 128 * VEC)
        *reinterpret_cast<vec_t*>(out + (size_t)r * d + c) = *reinterpret_cast<const vec_t*>(x + (size_t)src * d + c);
}

// decode step prologue: embed each utterance's newest token, derive its RoPE position
template <typename T>
__global__ __launch_bounds__(128) void dec_prepare_kernel(const T* __restrict__ emb, int emb_row0,
                                                          const int32_t* __restrict__ seq, int seq_cap,
                                                          const int32_t* __restrict__ seq_len,
                                                          const int32_t* __restrict__ resp_pos0,
                                                          T* __restrict__ x, int32_t* __restrict__ pos,
                                                          int32_t* __restrict__ cu, int d, int B) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int b = blockIdx.x;
    const int n = seq_len[b];
    const int tok = seq[(size_t)b * seq_cap + (n - 1)];
    if (threadIdx.x == 0) {
        pos[b] = (resp_pos0 ? resp_pos0[b] : 0) + n - 1;
        cu[b] = b;
        if (b == B - 1) cu[B] = B;
    }
    const T* src = emb + (size_t)(emb_row0 + tok) * d;
    for (int c = threadIdx.x * VEC; c < d; c += 128 * VEC)
        *reinterpret_cast<vec_t*>(x + (size_t)b * d + c) = *reinterpret_cast<const vec_t*>(src + c);
}

// wave-streaming decode path (dec_linear.hip): x in fragment order
__global__ __launch_bounds__(128) void dec_prepare_frag_kernel(const bf16* __restrict__ emb, int emb_row0,
                                                               const int32_t* __restrict__ seq, int seq_cap,
                                                               const int32_t* __restrict__ seq_len,
                                                               const int32_t* __restrict__ resp_pos0,
                                                               bf16* __restrict__ xf, int32_t* __restrict__ pos, int d,
                                                               const float* __restrict__ rope_tab, float* __restrict__ rope_m) {
    const int b = blockIdx.x;
    const int n = seq_len[b];
    const int tok = seq[(size_t)b * seq_cap + (n - 1)];
    const int p = (resp_pos0 ? resp_pos0[b] : 0) + n - 1;
    if (threadIdx.x == 0) pos[b] = p;
    if (threadIdx.x < 64) rope_m[b * 64 + threadIdx.x] = rope_tab[(size_t)p * 64 + threadIdx.x];   // this position's (cos, sin) row
    const bf16* src = emb + (size_t)(emb_row0 + tok) * d;
    for (int c = threadIdx.x * 8; c < d; c += 128 * 8)
        *reinterpret_cast<bf16x8*>(xf + (size_t)(b >> 4) * 16 * d + dec_frag_off<1>(b & 15, c, d)) = *reinterpret_cast<const bf16x8*>(src + c);
}

// decode step on embeddings computed outside (version >= 7: the AudioEncoder's row of the newest frame): x_in [B][d] row-major,
// pos_in [B] -> the same outputs as the two kernels above
template <typename T>
__global__ __launch_bounds__(128) void dec_prepare_x_kernel(const T* __restrict__ x_in, const int32_t* __restrict__ pos_in,
                                                            T* __restrict__ x, bf16* __restrict__ xf, int32_t* __restrict__ pos,
                                                            int32_t* __restrict__ cu, int d, int B,
                                                            const float* __restrict__ rope_tab, float* __restrict__ rope_m) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int b = blockIdx.x;
    const int p = pos_in[b];
    if (threadIdx.x == 0) {
        pos[b] = p;
        if (cu) { cu[b] = b; if (b == B - 1) cu[B] = B; }
    }
    if (rope_m && threadIdx.x < 64) rope_m[b * 64 + threadIdx.x] = rope_tab[(size_t)p * 64 + threadIdx.x];
    for (int c = threadIdx.x * VEC; c < d; c += 128 * VEC) {
        const vec_t v = *reinterpret_cast<const vec_t*>(x_in + (size_t)b * d + c);
        if constexpr (sizeof(T) == 2) {
            if (xf) { *reinterpret_cast<bf16x8*>(xf + (size_t)(b >> 4) * 16 * d + dec_frag_off<1>(b & 15, c, d)) = v; continue; }
        }
        *reinterpret_cast<vec_t*>(x + (size_t)b * d + c) = v;
    }
}

// fragment-order x of the wave-streaming path -> row-major rows
__global__ __launch_bounds__(128) void dec_unfrag_kernel(const bf16* __restrict__ xf, bf16* __restrict__ out, int d) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x * 8; c < d; c += 128 * 8)
        *reinterpret_cast<bf16x8*>(out + (size_t)b * d + c) = *reinterpret_cast<const bf16x8*>(xf + (size_t)(b >> 4) * 16 * d + dec_frag_off<1>(b & 15, c, d));
}

__global__ void dec_bump_ctx_kernel(int32_t* __restrict__ ctx_len, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) ctx_len[b] += 1;
}

struct Carver {
    char* p; size_t left; bool ok;
    Carver(void* base, size_t n) : p((char*)base), left(n), ok(true) {}
    void* take(size_t bytes) {
        bytes = align_up(bytes, 256);
        if (bytes > left) { ok = false; return nullptr; }
        void* r = p; p += bytes; left -= bytes; return r;
    }
};

extern "C" size_t valle_dense_workspace_bytes(const valle_model* m, int M, int n_out) {
    const size_t es = dtype_size(m->cfg.dtype), d = m->cfg.d_model, f = m->cfg.d_ff;
    size_t t = 0;
    t += align_up((size_t)M * d * es, 256) * 2;        // x, h
    t += align_up((size_t)M * 3 * d * es, 256);        // qkv
    t += align_up((size_t)M * f * es, 256);            // ff
    t += align_up((size_t)(n_out > 0 ? n_out : 1) * d * es, 256) * 2;  // selected rows (raw, normed)
    return t;
}

static int forward_dense_impl(valle_model* m, const int32_t* emb_idx, const int32_t* pos,
                              const int32_t* cu_seqlens, int B, int M, int max_T, int causal,
                              void* kcache, void* vcache, int cache_T, size_t cache_layer_stride,
                              const int32_t* cache_slot, const int32_t* cache_pos0,
                              const int32_t* out_rows, int n_out, int head, float* logits, int ld_logits,
                              void* hidden_out, void* workspace, size_t workspace_bytes, void* stream,
                              const int32_t* score_rows, int n_score, float* scores, int ld_scores,
                              const void* x_in = nullptr, int final_norm = 1, const int32_t* kv_lens = nullptr) {
    VALLE_REQUIRE(m && (emb_idx || x_in) && pos && cu_seqlens && workspace, VALLE_E_ARG, "valle_forward_dense: null pointer");
    VALLE_REQUIRE(B > 0 && M > 0, VALLE_E_ARG, "valle_forward_dense: empty batch (B=%d, M=%d)", B, M);
    VALLE_REQUIRE(n_out == 0 || out_rows, VALLE_E_ARG, "valle_forward_dense: out_rows missing");
    // the last layer runs its output projection and MLP on the n_out gathered rows inside buffers carved for M rows
    VALLE_REQUIRE(n_out >= 0 && n_out <= M, VALLE_E_ARG, "valle_forward_dense: n_out %d outside 0..M (%d)", n_out, M);
    VALLE_REQUIRE(logits == nullptr || (head >= 0 && head < m->n_cls && ld_logits >= m->head_n[head]), VALLE_E_ARG,
                  "valle_forward_dense: bad head %d / ld_logits %d", head, ld_logits);
    const valle_config& c = m->cfg;
    const int d = c.d_model, f = c.d_ff, H = c.n_heads, Dh = d / H, dt = c.dtype;
    const size_t es = dtype_size(dt);
    Carver ws(workspace, workspace_bytes);
    void* x = ws.take((size_t)M * d * es);
    void* h = ws.take((size_t)M * d * es);
    void* qkv = ws.take((size_t)M * 3 * d * es);
    void* ff = ws.take((size_t)M * f * es);
    void* sel = ws.take((size_t)(n_out > 0 ? n_out : 1) * d * es);
    void* seln = ws.take((size_t)(n_out > 0 ? n_out : 1) * d * es);
    VALLE_REQUIRE(ws.ok, VALLE_E_WORKSPACE, "valle_forward_dense: workspace %zu B too small (need %zu)", workspace_bytes,
                  valle_dense_workspace_bytes(m, M, n_out));
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define RUN(call) do { rc = (call); if (rc != VALLE_OK) return rc; } while (0)
    if (x_in) VALLE_HIP_CHECK(hipMemcpyAsync(x, x_in, (size_t)M * d * es, hipMemcpyDeviceToDevice, s));   // the stack works in place
    else RUN(valle_embed_sum(m->emb, emb_idx, x, M, d, dt, s));
    for (int l = 0; l < c.n_layers; ++l) {
        const valle_layer_weights& w = m->layers[l];
        // bf16 with folded weight copies: the two RMSNorms ride in the q/k/v and gate/up GEMMs (valle_gemm_normed)
        const bool fold = dt == VALLE_BF16 && w.f_qkv && w.f_gu && M >= 64 && d % 64 == 0;
        if (!fold) RUN(valle_rmsnorm(x, w.ln1, h, M, d, c.rms_eps, dt, s));
        void* kc = kcache ? (char*)kcache + (size_t)l * cache_layer_stride * es : nullptr;
        void* vc = vcache ? (char*)vcache + (size_t)l * cache_layer_stride * es : nullptr;
        if (dt == VALLE_BF16 && !kcache && Dh == 64 && d % 64 == 0 && M >= 64) {   // (a zero padding row is fine: rinv = 1/sqrt(eps), times 0)
            // passes without a KV cache (NAR levels, NAR-len steps): RoPE rides in the projection's epilogue
            if (fold) RUN(valle_gemm_normed(x, w.f_qkv, qkv, M, 3 * d, d, 3 * d, VALLE_EPI_STORE, c.rms_eps, m->rope, pos, s));
            else RUN(valle_gemm_qkv_rope(h, w.w_qkv, m->rope, pos, qkv, M, d, d, s));
        } else {
            if (fold) RUN(valle_gemm_normed(x, w.f_qkv, qkv, M, 3 * d, d, 3 * d, VALLE_EPI_STORE, c.rms_eps, nullptr, nullptr, s));
            else RUN(valle_gemm(h, w.w_qkv, nullptr, nullptr, qkv, M, 3 * d, d, 3 * d, VALLE_EPI_STORE, dt, s));
            RUN(valle_rope_kv(qkv, pos, m->rope, cu_seqlens, B, M, H, Dh, kc, vc, cache_T, cache_slot, cache_pos0, dt, s));
        }
        if (scores && l == c.n_layers - 1)      // output_attentions: the last layer's raw scaled products of the chosen rows
            RUN(valle_attn_row_scores_dense(qkv, d, score_rows, n_score, cu_seqlens, B, H, scores, ld_scores, dt, s));
        if (kv_lens) RUN(valle_attn_prefill_rowwise_kv(qkv, h, cu_seqlens, kv_lens, B, M, H, Dh, causal, dt, s));
        else RUN(valle_attn_prefill(qkv, h, cu_seqlens, B, M, max_T, H, Dh, causal, dt, s));
        // Behind the last layer's attention only the rows the caller asked for are read again (the other rows' K/V have
        // been consumed): output projection and MLP of the last layer run on those rows alone -- the same arithmetic per
        // row, 1 row of ~311 per utterance in the prefill, the response rows (~60 %) in the NAR passes
        void* xr = x; void* hr = h; int Mr = M;
        if (l == c.n_layers - 1 && n_out > 0) {
            if (dt == VALLE_BF16) {
                gather_rows_kernel<bf16><<<n_out, 128, 0, s>>>((const bf16*)x, out_rows, (bf16*)sel, d);
                gather_rows_kernel<bf16><<<n_out, 128, 0, s>>>((const bf16*)h, out_rows, (bf16*)seln, d);
            } else {
                gather_rows_kernel<float><<<n_out, 128, 0, s>>>((const float*)x, out_rows, (float*)sel, d);
                gather_rows_kernel<float><<<n_out, 128, 0, s>>>((const float*)h, out_rows, (float*)seln, d);
            }
            VALLE_LAUNCH_CHECK();
            xr = sel; hr = seln; Mr = n_out;
        }
        const bool fold2 = fold && Mr >= 64;
        RUN(valle_gemm(hr, w.w_o, nullptr, xr, xr, Mr, d, d, d, VALLE_EPI_RESID, dt, s));
        if (fold2) RUN(valle_gemm_normed(xr, w.f_gu, ff, Mr, 2 * f, d, f, VALLE_EPI_GEGLU, c.rms_eps, nullptr, nullptr, s));
        else {
            RUN(valle_rmsnorm(xr, w.ln2, hr, Mr, d, c.rms_eps, dt, s));
            RUN(valle_gemm(hr, w.w_gu, nullptr, nullptr, ff, Mr, 2 * f, d, f, VALLE_EPI_GEGLU, dt, s));
        }
        RUN(valle_gemm(ff, w.w_down, nullptr, xr, xr, Mr, d, f, d, VALLE_EPI_RESID, dt, s));
    }
    if (n_out > 0) {                                       // sel = the requested rows of the last layer's output
        void* dst = hidden_out ? hidden_out : seln;
        if (final_norm) RUN(valle_rmsnorm(sel, m->final_norm, dst, n_out, d, c.rms_eps, dt, s));
        else VALLE_HIP_CHECK(hipMemcpyAsync(dst, sel, (size_t)n_out * d * es, hipMemcpyDeviceToDevice, s));
        if (logits)
            RUN(valle_gemm(dst, m->head_w[head], m->head_b[head], nullptr, logits, n_out, m->head_n[head], d, ld_logits,
                           VALLE_EPI_F32OUT, dt, s));
    }
    return VALLE_OK;
}

extern "C" int valle_forward_dense(valle_model* m, const int32_t* emb_idx, const int32_t* pos,
                                   const int32_t* cu_seqlens, int B, int M, int max_T, int causal,
                                   void* kcache, void* vcache, int cache_T, size_t cache_layer_stride,
                                   const int32_t* cache_slot, const int32_t* cache_pos0,
                                   const int32_t* out_rows, int n_out, int head, float* logits, int ld_logits,
                                   void* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_dense_impl(m, emb_idx, pos, cu_seqlens, B, M, max_T, causal, kcache, vcache, cache_T, cache_layer_stride,
                              cache_slot, cache_pos0, out_rows, n_out, head, logits, ld_logits, hidden_out, workspace,
                              workspace_bytes, stream, nullptr, 0, nullptr, 0);
}

extern "C" int valle_forward_dense_scores(valle_model* m, const int32_t* emb_idx, const int32_t* pos,
                                          const int32_t* cu_seqlens, int B, int M, int max_T, int causal,
                                          void* kcache, void* vcache, int cache_T, size_t cache_layer_stride,
                                          const int32_t* cache_slot, const int32_t* cache_pos0,
                                          const int32_t* out_rows, int n_out, int head, float* logits, int ld_logits,
                                          void* hidden_out, void* workspace, size_t workspace_bytes, void* stream,
                                          const int32_t* score_rows, int n_score, float* scores, int ld_scores) {
    VALLE_REQUIRE(score_rows && scores && n_score > 0, VALLE_E_ARG, "valle_forward_dense_scores: score rows / buffer missing");
    VALLE_REQUIRE(ld_scores >= max_T, VALLE_E_ARG, "valle_forward_dense_scores: ld_scores %d < max_T %d", ld_scores, max_T);
    VALLE_REQUIRE(m && m->cfg.d_model / m->cfg.n_heads == 64, VALLE_E_UNSUPPORTED, "valle_forward_dense_scores: head_dim 64 only");
    return forward_dense_impl(m, emb_idx, pos, cu_seqlens, B, M, max_T, causal, kcache, vcache, cache_T, cache_layer_stride,
                              cache_slot, cache_pos0, out_rows, n_out, head, logits, ld_logits, hidden_out, workspace,
                              workspace_bytes, stream, score_rows, n_score, scores, ld_scores);
}

// The stack alone, on embeddings computed outside (version >= 7: the AudioEncoder's rows, models/base_v2.py:725-799), with
// or without the final RMSNorm (per_level_normalization moves it into the AudioDecoder, base_v2.py:495); no head.
extern "C" int valle_forward_layers(valle_model* m, const void* x_in, const int32_t* pos, const int32_t* cu_seqlens, int B,
                                    int M, int max_T, int causal, void* kcache, void* vcache, int cache_T,
                                    size_t cache_layer_stride, const int32_t* cache_slot, const int32_t* cache_pos0,
                                    const int32_t* out_rows, int n_out, int final_norm, const int32_t* kv_lens,
                                    void* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
    VALLE_REQUIRE(x_in && hidden_out && n_out > 0, VALLE_E_ARG, "valle_forward_layers: x_in, hidden_out and out_rows are required");
    VALLE_REQUIRE(!kv_lens || !kcache, VALLE_E_ARG, "valle_forward_layers: kv_lens (padding rows) and a KV cache do not go together");
    return forward_dense_impl(m, nullptr, pos, cu_seqlens, B, M, max_T, causal, kcache, vcache, cache_T, cache_layer_stride,
                              cache_slot, cache_pos0, out_rows, n_out, 0, nullptr, 0, hidden_out, workspace, workspace_bytes,
                              stream, nullptr, 0, nullptr, 0, x_in, final_norm, kv_lens);
}

extern "C" size_t valle_decode_workspace_bytes(const valle_model* m, int B) {
    const size_t es = dtype_size(m->cfg.dtype), d = m->cfg.d_model, f = m->cfg.d_ff;
    size_t t = 0;
    t += align_up((size_t)B * d * es, 256) * 3;        // x, h, att
    t += align_up((size_t)B * 3 * d * es, 256);        // qkv
    t += align_up((size_t)B * f * es, 256);            // ff
    t += align_up((size_t)B * 2048 * sizeof(float), 256);  // logits
    t += align_up((size_t)(B + 1) * sizeof(int32_t), 256) * 2 + align_up((size_t)B * sizeof(int32_t), 256);
    t += align_up((size_t)32 * d * es, 256) * 2 + align_up((size_t)32 * f * es, 256);  // its fragment-order x, att, ff (16-row tiles)
    t += align_up((size_t)32 * 64 * sizeof(float), 256);                // RoPE rows of the current positions
    return t;
}

static int ar_decode_impl(valle_model* m, const valle_ar_state* st, const valle_sample_params* sp,
                          int n_steps, float* logits_out, int ld_logits,
                          void* workspace, size_t workspace_bytes, void* stream,
                          float* scores = nullptr, int ld_scores = 0,
                          const void* x_in = nullptr, const int32_t* pos_in = nullptr, void* hidden_out = nullptr, int final_norm = 1) {
    VALLE_REQUIRE(m && st && workspace, VALLE_E_ARG, "valle_ar_decode: null pointer");
    const bool outside = x_in != nullptr;          // embeddings in, hidden state out: no token sequence, no head, no sampler
    VALLE_REQUIRE(sp != nullptr || n_steps == 1, VALLE_E_ARG, "valle_ar_decode: without sampler params only one step can run");
    VALLE_REQUIRE(n_steps >= 1, VALLE_E_ARG, "valle_ar_decode: n_steps %d", n_steps);
    VALLE_REQUIRE(st->B > 0 && st->ctx_len && st->kcache && st->vcache && st->cache_slot && (outside || (st->seq && st->seq_len && st->stopped)),
                  VALLE_E_ARG, "valle_ar_decode: incomplete state");
    // every step appends one key per utterance at ctx_len[b] and reads the RoPE row of its position (<= ctx_len[b]):
    // the host's bound on the contexts keeps both inside the cache and the table
    VALLE_REQUIRE(st->max_ctx > 0, VALLE_E_ARG, "valle_ar_decode: valle_ar_state.max_ctx (host bound on ctx_len) is required");
    VALLE_REQUIRE((long)st->max_ctx + n_steps <= st->cache_T, VALLE_E_ARG,
                  "valle_ar_decode: %d cached positions + %d steps exceed cache_T %d", st->max_ctx, n_steps, st->cache_T);
    VALLE_REQUIRE((long)st->max_ctx + n_steps <= m->cfg.max_pos, VALLE_E_ARG,
                  "valle_ar_decode: %d cached positions + %d steps exceed the RoPE table (%d rows)", st->max_ctx, n_steps, m->cfg.max_pos);
    VALLE_REQUIRE(outside || (st->head >= 0 && st->head < m->n_cls && st->vocab == m->head_n[st->head] && st->vocab <= 2048),
                  VALLE_E_ARG, "valle_ar_decode: bad head/vocab");
    VALLE_REQUIRE(logits_out == nullptr || ld_logits >= st->vocab, VALLE_E_ARG, "valle_ar_decode: ld_logits too small");
    const valle_config& c = m->cfg;
    const int d = c.d_model, f = c.d_ff, H = c.n_heads, Dh = d / H, dt = c.dtype, B = st->B;
    const size_t es = dtype_size(dt);
    Carver ws(workspace, workspace_bytes);
    void* x = ws.take((size_t)B * d * es);
    void* h = ws.take((size_t)B * d * es);
    void* att = ws.take((size_t)B * d * es);
    void* qkv = ws.take((size_t)B * 3 * d * es);
    void* ff = ws.take((size_t)B * f * es);
    float* lg_ws = (float*)ws.take((size_t)B * 2048 * sizeof(float));
    int32_t* pos = (int32_t*)ws.take((size_t)(B + 1) * sizeof(int32_t));
    int32_t* cu = (int32_t*)ws.take((size_t)(B + 1) * sizeof(int32_t));
    // the wave-streaming path keeps its activations as 16-row fragment-order matrices (one or two of them)
    bf16* xf = (bf16*)ws.take((size_t)32 * d * es);
    bf16* attf = (bf16*)ws.take((size_t)32 * d * es);
    bf16* fff = (bf16*)ws.take((size_t)32 * f * es);
    float* rope_m = (float*)ws.take((size_t)32 * 64 * sizeof(float));   // (cos, sin) row of every utterance's current position
    VALLE_REQUIRE(ws.ok, VALLE_E_WORKSPACE, "valle_ar_decode: workspace %zu B too small (need %zu)", workspace_bytes,
                  valle_decode_workspace_bytes(m, B));
    hipStream_t s = (hipStream_t)stream;
    const bool wavepath = (dt == VALLE_BF16) && B <= 32 && (d == 256 || d == 512 || d == 768 || d == 1024) && f == 4 * d &&
                          m->layers[0].pn_qkv && m->layers[0].p_o && m->layers[0].pn_gu && m->layers[0].p_down && (outside || m->head_pn[st->head]);
    // early exit of finished utterances: only with the fused sampler (it freezes their seq_len / ctx_len) on the fused path
    const bool early = st->early_exit != 0 && sp != nullptr && wavepath;
    int rc;
    for (int step = 0; step < n_steps; ++step) {
        float* lg = logits_out ? logits_out + (size_t)step * B * ld_logits : lg_ws;
        const int ldl = logits_out ? ld_logits : 2048;
        if (outside) {
            if (dt == VALLE_BF16)
                dec_prepare_x_kernel<bf16><<<B, 128, 0, s>>>((const bf16*)x_in, pos_in, (bf16*)x, wavepath ? xf : nullptr, pos, cu, d, B, m->rope, wavepath ? rope_m : nullptr);
            else
                dec_prepare_x_kernel<float><<<B, 128, 0, s>>>((const float*)x_in, pos_in, (float*)x, nullptr, pos, cu, d, B, m->rope, nullptr);
        } else if (wavepath && step > 0) {
            // the previous step's sampler already embedded its token (SampleArgs.nx_*)
        } else if (wavepath)
            dec_prepare_frag_kernel<<<B, 128, 0, s>>>((const bf16*)m->emb, st->resp_emb_row0, st->seq, st->max_steps, st->seq_len, st->resp_pos0, xf, pos, d, m->rope, rope_m);
        else if (dt == VALLE_BF16)
            dec_prepare_kernel<bf16><<<B, 128, 0, s>>>((const bf16*)m->emb, st->resp_emb_row0, st->seq, st->max_steps, st->seq_len, st->resp_pos0, (bf16*)x, pos, cu, d, B);
        else
            dec_prepare_kernel<float><<<B, 128, 0, s>>>((const float*)m->emb, st->resp_emb_row0, st->seq, st->max_steps, st->seq_len, st->resp_pos0, (float*)x, pos, cu, d, B);
        VALLE_LAUNCH_CHECK();
        for (int l = 0; l < c.n_layers; ++l) {
            const valle_layer_weights& w = m->layers[l];
            void* kc = (char*)st->kcache + (size_t)l * st->cache_layer_stride * es;
            void* vc = (char*)st->vcache + (size_t)l * st->cache_layer_stride * es;
            if (wavepath) {
                DecWArgs a{};
                a.x = xf; a.W = (const bf16*)w.pn_qkv; a.out = qkv; a.ldo = 3 * d; a.B = B; a.N = 3 * d; a.K = d;
                a.eps = c.rms_eps; a.norm = 1;
                a.rope_m = rope_m; a.ctx_len = st->ctx_len; a.slot = st->cache_slot;
                a.kc = (bf16*)kc; a.vc = (bf16*)vc; a.cache_T = st->cache_T; a.H = H; a.d = d;
                RUN(valle_dec_wave_linear(a, DEC_EPI_QKV, s));
                if (scores && l == c.n_layers - 1)
                    RUN(valle_attn_row_scores_cache(qkv, (size_t)3 * d, kc, st->ctx_len, st->cache_slot, 1, B, H, st->cache_T, scores, ld_scores, dt, s));
                RUN(valle_attn_decode_bf16(qkv, (size_t)3 * d, kc, vc, st->ctx_len, st->cache_slot, attf, B, H, st->cache_T, 1, 1, s));
                DecWArgs o{};
                o.x = attf; o.W = (const bf16*)w.p_o; o.resid = xf; o.out = xf; o.B = B; o.N = d; o.K = d;
                RUN(valle_dec_wave_linear(o, VALLE_EPI_RESID, s));
                DecWArgs g{};
                g.x = xf; g.W = (const bf16*)w.pn_gu; g.out = fff; g.B = B; g.N = 2 * f; g.K = d;
                g.eps = c.rms_eps; g.norm = 1;
                RUN(valle_dec_wave_linear(g, VALLE_EPI_GEGLU, s));
                DecWArgs dn{};
                dn.x = fff; dn.W = (const bf16*)w.p_down; dn.resid = xf; dn.out = xf; dn.B = B; dn.N = d; dn.K = f;
                RUN(valle_dec_wave_linear(dn, VALLE_EPI_RESID, s));
            } else {
                RUN(valle_rmsnorm(x, w.ln1, h, B, d, c.rms_eps, dt, s));
                RUN(valle_gemm(h, w.w_qkv, nullptr, nullptr, qkv, B, 3 * d, d, 3 * d, VALLE_EPI_STORE, dt, s));
                RUN(valle_rope_kv(qkv, pos, m->rope, cu, B, B, H, Dh, kc, vc, st->cache_T, st->cache_slot, st->ctx_len, dt, s));
                if (scores && l == c.n_layers - 1)
                    RUN(valle_attn_row_scores_cache(qkv, (size_t)3 * d, kc, st->ctx_len, st->cache_slot, 1, B, H, st->cache_T, scores, ld_scores, dt, s));
                RUN(valle_attn_decode_launch(qkv, kc, vc, st->ctx_len, st->cache_slot, att, B, H, Dh, st->cache_T, 1, dt, s));
                RUN(valle_gemm(att, w.w_o, nullptr, x, x, B, d, d, d, VALLE_EPI_RESID, dt, s));
                RUN(valle_rmsnorm(x, w.ln2, h, B, d, c.rms_eps, dt, s));
                RUN(valle_gemm(h, w.w_gu, nullptr, nullptr, ff, B, 2 * f, d, f, VALLE_EPI_GEGLU, dt, s));
                RUN(valle_gemm(ff, w.w_down, nullptr, x, x, B, d, f, d, VALLE_EPI_RESID, dt, s));
            }
        }
        if (outside) {
            // the newest row of the last hidden state, raw or through the final norm; the caller's decoder and sampler follow
            void* raw = final_norm ? h : hidden_out;
            if (wavepath) dec_unfrag_kernel<<<B, 128, 0, s>>>(xf, (bf16*)raw, d);
            else VALLE_HIP_CHECK(hipMemcpyAsync(raw, x, (size_t)B * d * es, hipMemcpyDeviceToDevice, s));
            if (final_norm) RUN(valle_rmsnorm(h, m->final_norm, hidden_out, B, d, c.rms_eps, dt, s));
            dec_bump_ctx_kernel<<<ceil_div(B, 64), 64, 0, s>>>(st->ctx_len, B);       // the row's key / value now sit in the cache
            VALLE_LAUNCH_CHECK();
            break;
        }
        if (wavepath) {
            DecWArgs hd{};
            hd.x = xf; hd.W = (const bf16*)m->head_pn[st->head]; hd.bias = m->head_b[st->head]; hd.out = lg; hd.ldo = ldl;
            hd.B = B; hd.N = st->vocab; hd.K = d; hd.eps = c.rms_eps; hd.norm = 1;
            RUN(valle_dec_wave_linear(hd, VALLE_EPI_F32OUT, s));
        } else {
            RUN(valle_rmsnorm(x, m->final_norm, h, B, d, c.rms_eps, dt, s));
            RUN(valle_gemm(h, m->head_w[st->head], m->head_b[st->head], nullptr, lg, B, st->vocab, d, ldl, VALLE_EPI_F32OUT, dt, s));
        }
        if (sp == nullptr) break;   // logits only: the caller samples and appends (CFG path)
        SampleArgs sa{};
        sa.logits = lg; sa.ld = ldl; sa.vocab = st->vocab; sa.nar_mode = 0; sa.p = *sp;
        sa.ids = wavepath ? cu : pos;  // scratch: the token also lands in seq
        if (wavepath && step + 1 < n_steps) {
            sa.nx_emb = (const bf16*)m->emb; sa.nx_row0 = st->resp_emb_row0; sa.nx_x = xf;
            sa.nx_pos = pos; sa.nx_pos0 = st->resp_pos0; sa.nx_d = d; sa.nx_rope_tab = m->rope; sa.nx_rope = rope_m;
        }
        sa.scores = st->scores;
        sa.seq = st->seq; sa.seq_cap = st->max_steps; sa.seq_len = st->seq_len; sa.ctx_len = st->ctx_len; sa.stopped = st->stopped;
        sa.prev_is_seq = 1; sa.step = (uint64_t)step; sa.early_exit = early ? 1 : 0;
        RUN(valle_sample_launch(sa, B, s));
    }
    return VALLE_OK;
}

extern "C" int valle_ar_decode(valle_model* m, const valle_ar_state* st, const valle_sample_params* sp,
                               int n_steps, float* logits_out, int ld_logits,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return ar_decode_impl(m, st, sp, n_steps, logits_out, ld_logits, workspace, workspace_bytes, stream);
}

extern "C" int valle_ar_decode_scores(valle_model* m, const valle_ar_state* st, float* logits_out, int ld_logits,
                                      float* scores, int ld_scores, void* workspace, size_t workspace_bytes, void* stream) {
    VALLE_REQUIRE(logits_out && scores, VALLE_E_ARG, "valle_ar_decode_scores: logits / scores buffer missing");
    VALLE_REQUIRE(st && ld_scores >= st->max_ctx + 1, VALLE_E_ARG, "valle_ar_decode_scores: ld_scores too small for max_ctx + 1 keys");
    VALLE_REQUIRE(m && m->cfg.d_model / m->cfg.n_heads == 64, VALLE_E_UNSUPPORTED, "valle_ar_decode_scores: head_dim 64 only");
    return ar_decode_impl(m, st, nullptr, 1, logits_out, ld_logits, workspace, workspace_bytes, stream, scores, ld_scores);
}

// One decode step of the stack on embeddings computed outside (version >= 7 AR loop: the AudioEncoder's row of the newest frame
// in, the last hidden state out, models/ar_nar_v2.py:504-572 with the KV cache the reference leaves unused, base_v2.py:539).
extern "C" int valle_ar_decode_layers(valle_model* m, const valle_ar_state* st, const void* x_in, const int32_t* pos_in,
                                      void* hidden_out, int final_norm, void* workspace, size_t workspace_bytes, void* stream) {
    VALLE_REQUIRE(x_in && pos_in && hidden_out, VALLE_E_ARG, "valle_ar_decode_layers: x_in, pos_in and hidden_out are required");
    return ar_decode_impl(m, st, nullptr, 1, nullptr, 0, workspace, workspace_bytes, stream, nullptr, 0, x_in, pos_in, hidden_out, final_norm);
}

// ------------------------------------------------------------------ guided AR steps without a host round trip
extern "C" int valle_ar_decode_cfg(valle_model* m, const valle_ar_state* st_pos, const valle_ar_state* st_neg,
                                   const valle_sample_params* sp, int n_steps, int first_step,
                                   float* hist_pos, float* hist_neg, int ld, int hist_rows, int hist_row0,
                                   const int32_t* row_start, const int32_t* row_count0, float strength, float rescale,
                                   float* last, size_t last_step_stride,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    VALLE_REQUIRE(m && st_pos && st_neg && sp && hist_pos && hist_neg && row_start && row_count0 && last && workspace, VALLE_E_ARG,
                  "valle_ar_decode_cfg: null pointer");
    VALLE_REQUIRE(n_steps >= 1 && first_step >= 1, VALLE_E_ARG, "valle_ar_decode_cfg: n_steps %d, first_step %d (step 0 samples from the prefill)", n_steps, first_step);
    VALLE_REQUIRE(st_pos->B == st_neg->B && st_pos->seq == st_neg->seq && st_pos->seq_len == st_neg->seq_len &&
                  st_pos->stopped == st_neg->stopped && st_pos->vocab == st_neg->vocab, VALLE_E_ARG,
                  "valle_ar_decode_cfg: the two states must share the sequences and differ in the caches only");
    VALLE_REQUIRE(ld >= st_pos->vocab && hist_row0 + first_step + n_steps <= hist_rows, VALLE_E_ARG,
                  "valle_ar_decode_cfg: history of %d rows too short for rows %d + %d + %d", hist_rows, hist_row0, first_step, n_steps);
    const int B = st_pos->B;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    for (int i = 0; i < n_steps; ++i) {
        const int g = first_step + i;                                   // the step's index in the whole loop
        valle_ar_state a = *st_pos, b = *st_neg;
        a.max_ctx += i; b.max_ctx += i; a.early_exit = 0; b.early_exit = 0;
        // the step's logits land in their history rows: row (hist_row0 + g) of every utterance's [hist_rows][ld] block
        const size_t off = (size_t)(hist_row0 + g) * ld;
        RUN(ar_decode_impl(m, &a, nullptr, 1, hist_pos + off, hist_rows * ld, workspace, workspace_bytes, stream));
        RUN(ar_decode_impl(m, &b, nullptr, 1, hist_neg + off, hist_rows * ld, workspace, workspace_bytes, stream));
        float* out = last + (size_t)i * last_step_stride;
        // guidance over every response row so far (samplers.py:177-191: the statistics run over rows x vocab), last row out
        RUN(valle_cfg_combine_last_add(hist_pos, hist_neg, ld, st_pos->vocab, row_start, row_count0, g, B, strength, rescale, out, ld, s));
        SampleArgs sa{};
        sa.logits = out; sa.ld = ld; sa.vocab = st_pos->vocab; sa.nar_mode = 0; sa.p = *sp;
        sa.ids = (int32_t*)workspace;                                   // scratch (the workspace's head is rewritten by the next step's prologue)
        sa.scores = st_pos->scores;
        sa.seq = st_pos->seq; sa.seq_cap = st_pos->max_steps; sa.seq_len = st_pos->seq_len; sa.stopped = st_pos->stopped;
        sa.ctx_len = st_pos->ctx_len; sa.ctx_len2 = st_neg->ctx_len;    // both caches ran the previous token through the stack
        sa.prev_is_seq = 1; sa.step = (uint64_t)i;
        RUN(valle_sample_launch(sa, B, s));
    }
    return VALLE_OK;
}

// ------------------------------------------------------------------ decode without a caller workspace
// valle_ar_decode_noalloc: valle_ar_decode with the step's activations in a static device array instead of a caller
// workspace, for callers that cannot allocate (HIP graph capture, serving loops that forbid hipMalloc).  Nothing here
// allocates: the array lives in the code object (each GPU the library runs on holds its own copy until the process ends).
extern "C" size_t valle_noalloc_scratch_bytes(void) { return VALLE_NOALLOC_SCRATCH_BYTES; }

extern "C" int valle_ar_decode_noalloc(valle_model* m, const valle_ar_state* st, const valle_sample_params* sp,
                                       int n_steps, float* logits_out, int ld_logits, void* stream) {
    VALLE_REQUIRE(m && st, VALLE_E_ARG, "valle_ar_decode_noalloc: null pointer");
    VALLE_REQUIRE(st->B > 0, VALLE_E_ARG, "valle_ar_decode_noalloc: empty batch");
    const size_t need = valle_decode_workspace_bytes(m, st->B);
    VALLE_REQUIRE(need <= VALLE_NOALLOC_SCRATCH_BYTES, VALLE_E_WORKSPACE,
                  "valle_ar_decode_noalloc: %d utterances at d_model %d need %zu B of scratch, the static array holds %u B; "
                  "use valle_ar_decode with a caller workspace", st->B, m->cfg.d_model, need, VALLE_NOALLOC_SCRATCH_BYTES);
    // the array of the GPU the model was created on (its weights live there, so must the stream); a descriptor made
    // before any device was current falls back to the current device, resolved now (call once outside capture then)
    int dev = m->device;
    if (dev < 0) VALLE_HIP_CHECK(hipGetDevice(&dev));
    void* ptr = nullptr;
    const int rcs = scratch_resolve(dev, &ptr);
    if (rcs != VALLE_OK) return rcs;
    if (stream) {                                    // a stream of another GPU would run the kernels on another GPU's array
        int sdev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess)
            VALLE_REQUIRE(sdev == dev, VALLE_E_ARG, "valle_ar_decode_noalloc: stream belongs to device %d, the model to device %d", sdev, dev);
    }
    // every byte of the workspace that a step reads was written earlier in the same step (or by the previous step's
    // sampler), so nothing has to be cleared; calls that overlap on one GPU share the array and race
    return ar_decode_impl(m, st, sp, n_steps, logits_out, ld_logits, ptr, VALLE_NOALLOC_SCRATCH_BYTES, stream);
}
