// Entropix sampling (reference samplers.py:281-566; wired at models/base.py:1325-1342, models/ar_nar.py:782, :807).
//   attn_row_scores   the LAST layer's raw scaled products q.k / sqrt(Dh) of chosen rows -- what the reference's attention
//                     returns under output_attentions (arch/llama.py:429, :469-477: before mask and softmax)
//   entropix_metrics  calculate_entropix_metrics (samplers.py:281-321) on the newest row
//   logsoftmax_colsum sum over an utterance's rows of log_softmax[row][v] (the adaptive branch's score, :527-545)
//   entropix_pick     _sample_entropix (:419-462), n draws, the best by that score (:547-550)
// The branch logic between the metrics and the pick (samplers.py:489-523) is a handful of Python floats per utterance and
// stays on the host, as in the reference (which moves everything to the CPU for it, models/base.py:1328-1339).
#include "common.h"

#define EX_THREADS 256
#define EX_MAXV 2048
#define EX_MAXH 64

__device__ __forceinline__ float ex_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < EX_THREADS / 64; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float ex_block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < EX_THREADS / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ------------------------------------------------------------------------------------------------ raw scores of rows
// scores[r][h][j] = scale * sum_e q[r][h][e] k[j][h][e], j < n_keys(r), head_dim 64.
//   dense (cu != NULL):  row r is packed row q_rows[r] of a [M][q_row_stride] matrix, its keys are all rows of its
//                        utterance (the reference returns the products unmasked)
//   cache (cu == NULL):  row r = utterance r, keys 0 .. ctx_len[r] + ctx_extra - 1 of cache slot slot[r]
template <typename T>
__global__ __launch_bounds__(EX_THREADS) void attn_row_scores_kernel(const T* __restrict__ q, size_t q_row_stride,
                                                                     const int32_t* __restrict__ q_rows,
                                                                     const T* __restrict__ k, size_t k_row_stride,
                                                                     size_t k_head_stride, size_t k_item_stride,
                                                                     const int32_t* __restrict__ cu, int B,
                                                                     const int32_t* __restrict__ ctx_len,
                                                                     const int32_t* __restrict__ slot, int ctx_extra,
                                                                     float* __restrict__ out, int ld, int H, float scale) {
    const int r = blockIdx.x, h = blockIdx.y;
    int n_keys;
    const T* qp;
    const T* kp;
    if (cu != nullptr) {
        const int m = q_rows ? q_rows[r] : r;
        int b = 0;
        while (b + 1 < B && cu[b + 1] <= m) ++b;
        n_keys = cu[b + 1] - cu[b];
        qp = q + (size_t)m * q_row_stride + h * 64;
        kp = k + (size_t)cu[b] * k_row_stride + (size_t)h * k_head_stride;
    } else {
        const int c = ctx_len[r];
        n_keys = (c < 0 ? ~c : c) + ctx_extra;               // (a finished utterance's context is stored negated)
        qp = q + (size_t)r * q_row_stride + h * 64;
        kp = k + (size_t)(slot ? slot[r] : r) * k_item_stride + (size_t)h * k_head_stride;
    }
    float qv[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) qv[e] = to_f32<T>(qp[e]);
    float* o = out + ((size_t)r * H + h) * ld;
    for (int j = threadIdx.x; j < n_keys; j += EX_THREADS) {
        const T* kr = kp + (size_t)j * k_row_stride;
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 64; ++e) acc = fmaf(qv[e], to_f32<T>(kr[e]), acc);
        // a low-precision model's product is a tensor of that type (torch.matmul output) before the division by sqrt(Dh)
        o[j] = to_f32<T>(from_f32<T>(acc)) * scale;
    }
}

int valle_attn_row_scores_dense(const void* qkv, int d, const int32_t* rows, int n_rows, const int32_t* cu, int B, int H,
                                float* out, int ld, int dtype, hipStream_t s) {
    if (n_rows <= 0) return VALLE_OK;
    dim3 grid(n_rows, H);
    if (dtype == VALLE_BF16)
        attn_row_scores_kernel<bf16><<<grid, EX_THREADS, 0, s>>>((const bf16*)qkv, (size_t)3 * d, rows, (const bf16*)qkv + d, (size_t)3 * d, 64, 0,
                                                                 cu, B, nullptr, nullptr, 0, out, ld, H, 0.125f);
    else
        attn_row_scores_kernel<float><<<grid, EX_THREADS, 0, s>>>((const float*)qkv, (size_t)3 * d, rows, (const float*)qkv + d, (size_t)3 * d, 64, 0,
                                                                  cu, B, nullptr, nullptr, 0, out, ld, H, 0.125f);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

int valle_attn_row_scores_cache(const void* q, size_t q_row_stride, const void* kcache, const int32_t* ctx_len,
                                const int32_t* slot, int ctx_extra, int B, int H, int cache_T, float* out, int ld, int dtype,
                                hipStream_t s) {
    dim3 grid(B, H);
    const size_t hs = (size_t)cache_T * 64, is = (size_t)H * hs;
    if (dtype == VALLE_BF16)
        attn_row_scores_kernel<bf16><<<grid, EX_THREADS, 0, s>>>((const bf16*)q, q_row_stride, nullptr, (const bf16*)kcache, 64, hs, is,
                                                                 nullptr, B, ctx_len, slot, ctx_extra, out, ld, H, 0.125f);
    else
        attn_row_scores_kernel<float><<<grid, EX_THREADS, 0, s>>>((const float*)q, q_row_stride, nullptr, (const float*)kcache, 64, hs, is,
                                                                  nullptr, B, ctx_len, slot, ctx_extra, out, ld, H, 0.125f);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// ------------------------------------------------------------------------------------------------ metrics
// One workgroup per utterance.  out[b][0..5] = logits entropy, logits varentropy (base 2), attention entropy (mean over
// heads of the entropy over keys of softmax(scores)), its unbiased variance over heads (NaN -> 0), agreement (mean
// |p - mean over heads|), interaction strength (mean |score|).
__global__ __launch_bounds__(EX_THREADS) void entropix_metrics_kernel(const float* __restrict__ logits, int ld, int vocab,
                                                                      const int32_t* __restrict__ logit_row,
                                                                      const float* __restrict__ scores, int ld_s, int H,
                                                                      const int32_t* __restrict__ n_keys,
                                                                      float* __restrict__ out) {
    __shared__ float red[EX_THREADS / 64];
    __shared__ float hmax[EX_MAXH], hinv[EX_MAXH], hent[EX_MAXH];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float LN2 = 0.69314718056f;                           // samplers.py:278
    const float* lg = logits + (size_t)(logit_row ? logit_row[b] : b) * ld;
    float mx = -INFINITY;
    for (int v = tid; v < vocab; v += EX_THREADS) mx = fmaxf(mx, lg[v]);
    mx = ex_block_max(mx, red);
    float se = 0.f;
    for (int v = tid; v < vocab; v += EX_THREADS) se += expf(lg[v] - mx);
    se = ex_block_sum(se, red);
    const float lse = mx + logf(se);
    float e = 0.f;
    for (int v = tid; v < vocab; v += EX_THREADS) { const float lp = lg[v] - lse; e += expf(lp) * lp; }
    const float ent = -ex_block_sum(e, red) / LN2;
    float ve = 0.f;
    for (int v = tid; v < vocab; v += EX_THREADS) { const float lp = lg[v] - lse, t = lp / LN2 + ent; ve += expf(lp) * t * t; }
    const float vent = ex_block_sum(ve, red);

    const int T = n_keys[b];
    const float* sc = scores + (size_t)b * H * ld_s;
    float inter = 0.f;
    for (int h = 0; h < H; ++h) {
        const float* s = sc + (size_t)h * ld_s;
        float m = -INFINITY, ab = 0.f;
        for (int j = tid; j < T; j += EX_THREADS) { m = fmaxf(m, s[j]); ab += fabsf(s[j]); }
        m = ex_block_max(m, red);
        inter += ex_block_sum(ab, red);
        float z = 0.f;
        for (int j = tid; j < T; j += EX_THREADS) z += expf(s[j] - m);
        z = ex_block_sum(z, red);
        float he = 0.f;
        for (int j = tid; j < T; j += EX_THREADS) {
            const float p = expf(s[j] - m) / z;
            he += p * log2f(fminf(fmaxf(p, 1e-10f), 1.0f));
        }
        he = -ex_block_sum(he, red);
        if (tid == 0) { hmax[h] = m; hinv[h] = 1.0f / z; hent[h] = he; }
    }
    __syncthreads();
    float ag = 0.f;
    for (int j = tid; j < T; j += EX_THREADS) {
        float mean = 0.f;
        for (int h = 0; h < H; ++h) mean += expf(sc[(size_t)h * ld_s + j] - hmax[h]) * hinv[h];
        mean /= (float)H;
        for (int h = 0; h < H; ++h) ag += fabsf(expf(sc[(size_t)h * ld_s + j] - hmax[h]) * hinv[h] - mean);
    }
    ag = ex_block_sum(ag, red);
    if (tid == 0) {
        float me = 0.f;
        for (int h = 0; h < H; ++h) me += hent[h];
        me /= (float)H;
        float var = 0.f;
        for (int h = 0; h < H; ++h) var += (hent[h] - me) * (hent[h] - me);
        var = H > 1 ? var / (float)(H - 1) : 0.f;                 // torch.var of one head is NaN, replaced by 0 (:307)
        float* o = out + (size_t)b * 8;
        o[0] = ent; o[1] = vent; o[2] = me; o[3] = var;
        o[4] = ag / ((float)H * (float)T); o[5] = inter / ((float)H * (float)T);
        o[6] = 0.f; o[7] = 0.f;
    }
}

extern "C" int valle_entropix_metrics(const float* logits, int ld, int vocab, const int32_t* logit_row, const float* scores,
                                      int ld_scores, int H, const int32_t* n_keys, int B, float* out, void* stream) {
    VALLE_REQUIRE(logits && scores && n_keys && out, VALLE_E_ARG, "valle_entropix_metrics: null pointer");
    VALLE_REQUIRE(vocab > 1 && ld >= vocab && H >= 1 && H <= EX_MAXH && ld_scores > 0, VALLE_E_ARG,
                  "valle_entropix_metrics: vocab %d ld %d heads %d (max %d)", vocab, ld, H, EX_MAXH);
    if (B <= 0) return VALLE_OK;
    entropix_metrics_kernel<<<B, EX_THREADS, 0, (hipStream_t)stream>>>(logits, ld, vocab, logit_row, scores, ld_scores, H, n_keys, out);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// ------------------------------------------------------------------------------------------------ column sums of log-softmax
// acc[b][v] (+)= sum over rows [row_off[b], row_off[b+1]) of (logits[row][v] - logsumexp(logits[row])).
__global__ __launch_bounds__(EX_THREADS) void logsoftmax_colsum_kernel(const float* __restrict__ logits, int ld, int vocab,
                                                                       const int32_t* __restrict__ row_off,
                                                                       float* __restrict__ acc, int ld_acc, int accumulate) {
    __shared__ float red[EX_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    constexpr int PER = EX_MAXV / EX_THREADS;
    float a[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int v = tid + i * EX_THREADS; a[i] = (accumulate && v < vocab) ? acc[(size_t)b * ld_acc + v] : 0.f; }
    for (int row = row_off[b]; row < row_off[b + 1]; ++row) {
        const float* lg = logits + (size_t)row * ld;
        float x[PER];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int v = tid + i * EX_THREADS; x[i] = v < vocab ? lg[v] : -INFINITY; mx = fmaxf(mx, x[i]); }
        mx = ex_block_max(mx, red);
        float se = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) se += expf(x[i] - mx);      // exp(-inf) = 0 past the vocabulary
        se = ex_block_sum(se, red);
        const float lse = mx + logf(se);
#pragma unroll
        for (int i = 0; i < PER; ++i) if (tid + i * EX_THREADS < vocab) a[i] += x[i] - lse;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int v = tid + i * EX_THREADS; if (v < vocab) acc[(size_t)b * ld_acc + v] = a[i]; }
}

extern "C" int valle_logsoftmax_colsum(const float* logits, int ld, int vocab, const int32_t* row_off, int B, float* acc,
                                       int ld_acc, int accumulate, void* stream) {
    VALLE_REQUIRE(logits && row_off && acc, VALLE_E_ARG, "valle_logsoftmax_colsum: null pointer");
    VALLE_REQUIRE(vocab > 1 && vocab <= EX_MAXV && ld >= vocab && ld_acc >= vocab, VALLE_E_ARG, "valle_logsoftmax_colsum: vocab %d", vocab);
    if (B <= 0) return VALLE_OK;
    logsoftmax_colsum_kernel<<<B, EX_THREADS, 0, (hipStream_t)stream>>>(logits, ld, vocab, row_off, acc, ld_acc, accumulate);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// ------------------------------------------------------------------------------------------------ the pick
// Philox4x32-10 -> one uniform in (0,1]
__device__ __forceinline__ float ex_philox_uniform(uint64_t seed, uint64_t offset, uint32_t subseq) {
    uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), subseq, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    return ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// One workgroup per utterance: probs = softmax(logit / temperature) of the newest row; the top_k largest taken in
// ASCENDING order; those whose smaller neighbours already hold more than top_p dropped; renormalised; n draws
// argmax(p / q), q ~ Exponential(1) (noise != NULL: q given per draw and ascending position -- tests; else Philox);
// the draw with the largest colsum[token] wins (first on ties).  temperature / top_k / top_p arrive clamped.
// Out: ids[b] = the winner, draws[b][0..n) = every draw's token, forced[b][:] = -inf except 0 at the winner.
__global__ __launch_bounds__(EX_THREADS) void entropix_pick_kernel(const float* __restrict__ logits, int ld, int vocab,
                                                                   const int32_t* __restrict__ logit_row,
                                                                   const float* __restrict__ temperature,
                                                                   const int32_t* __restrict__ top_k,
                                                                   const float* __restrict__ top_p,
                                                                   const int32_t* __restrict__ n_draws,
                                                                   const float* __restrict__ colsum, int ld_cs,
                                                                   const float* __restrict__ noise, int ld_noise, int max_draws,
                                                                   uint64_t seed, uint64_t offset,
                                                                   int32_t* __restrict__ ids, int32_t* __restrict__ draws,
                                                                   float* __restrict__ forced, int ld_forced) {
    __shared__ __attribute__((aligned(16))) float skey[EX_MAXV];
    __shared__ __attribute__((aligned(16))) int sidx[EX_MAXV];
    __shared__ float asc[1024];
    __shared__ float red[EX_THREADS / 64];
    __shared__ float s_best;
    __shared__ int s_arg;
    const int b = blockIdx.x, tid = threadIdx.x, V = vocab;
    const float* lg = logits + (size_t)(logit_row ? logit_row[b] : b) * ld;
    const float temp = temperature[b];
    float mx = -INFINITY;
    for (int v = tid; v < V; v += EX_THREADS) { const float x = lg[v] / temp; skey[v] = x; mx = fmaxf(mx, x); }
    mx = ex_block_max(mx, red);
    float se = 0.f;
    for (int v = tid; v < V; v += EX_THREADS) se += expf(skey[v] - mx);
    se = ex_block_sum(se, red);
    for (int v = tid; v < EX_MAXV; v += EX_THREADS) {
        skey[v] = v < V ? expf(skey[v] - mx) / se : -1.0f;        // probabilities; padding sorts last
        sidx[v] = v;
    }
    __syncthreads();
    // bitonic sort, descending by (probability, then lower index first)
    for (int k2 = 2; k2 <= EX_MAXV; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < EX_MAXV; t += EX_THREADS) {
                const int u = t ^ j;
                if (u > t) {
                    const float x = skey[t], y = skey[u];
                    const int xi = sidx[t], yi = sidx[u];
                    const bool x_first = (x > y) || (x == y && xi < yi);
                    const bool desc = (t & k2) == 0;
                    if (desc ? !x_first : x_first) { skey[t] = y; skey[u] = x; sidx[t] = yi; sidx[u] = xi; }
                }
            }
            __syncthreads();
        }
    int k = top_k[b];
    k = k < 1 ? 1 : (k > 1024 ? 1024 : k);
    k = k > V ? V : k;
    if (tid == 0) {                                            // ascending cumulative sum, top-p cut, total
        const float tp = top_p[b];
        float run = 0.f, tot = 0.f;
        for (int i = 0; i < k; ++i) {
            const float p = skey[k - 1 - i];
            run += p;
            const float kept = (run - p > tp) ? 0.f : p;
            asc[i] = kept;
            tot += kept;
        }
        for (int i = 0; i < k; ++i) asc[i] /= tot;
    }
    __syncthreads();
    const int n = n_draws[b] < 1 ? 1 : (n_draws[b] > max_draws ? max_draws : n_draws[b]);
    float best_score = -INFINITY;
    int best_tok = 0;
    for (int dr = 0; dr < n; ++dr) {
        float bv = -INFINITY;
        int bi = 1 << 30;
        for (int i = tid; i < k; i += EX_THREADS) {
            float q;
            if (noise) q = i < ld_noise ? noise[((size_t)b * max_draws + dr) * ld_noise + i] : 1.0f;
            else q = -logf(ex_philox_uniform(seed, offset, (uint32_t)((b * max_draws + dr) * 1024 + i)));
            const float val = asc[i] / q;
            if (val > bv) { bv = val; bi = i; }                 // ascending i within a thread: the first maximum stays
        }
        const float wm = ex_block_max(bv, red);
        if (tid == 0) s_arg = 1 << 30;
        __syncthreads();
        if (bv == wm) atomicMin(&s_arg, bi);                    // torch.argmax: the first maximum
        __syncthreads();
        const int tok = sidx[k - 1 - s_arg];
        if (tid == 0) {
            draws[(size_t)b * max_draws + dr] = tok;
            s_best = colsum ? colsum[(size_t)b * ld_cs + tok] : 0.f;
        }
        __syncthreads();
        if (dr == 0 || s_best > best_score) { best_score = s_best; best_tok = tok; }
        __syncthreads();
    }
    if (tid == 0) ids[b] = best_tok;
    if (forced)
        for (int v = tid; v < ld_forced; v += EX_THREADS) forced[(size_t)b * ld_forced + v] = (v == best_tok) ? 0.f : -INFINITY;
}

extern "C" int valle_entropix_pick(const float* logits, int ld, int vocab, const int32_t* logit_row, int B,
                                   const float* temperature, const int32_t* top_k, const float* top_p, const int32_t* n_draws,
                                   int max_draws, const float* colsum, int ld_colsum, const float* noise, int ld_noise,
                                   uint64_t seed, uint64_t offset, int32_t* ids, int32_t* draws, float* forced, int ld_forced,
                                   void* stream) {
    VALLE_REQUIRE(logits && temperature && top_k && top_p && n_draws && ids && draws, VALLE_E_ARG, "valle_entropix_pick: null pointer");
    VALLE_REQUIRE(vocab > 1 && vocab <= EX_MAXV && ld >= vocab, VALLE_E_ARG, "valle_entropix_pick: vocab %d (max %d), ld %d", vocab, EX_MAXV, ld);
    VALLE_REQUIRE(max_draws >= 1 && max_draws <= 16, VALLE_E_ARG, "valle_entropix_pick: max_draws %d", max_draws);
    VALLE_REQUIRE(!colsum || ld_colsum >= vocab, VALLE_E_ARG, "valle_entropix_pick: ld_colsum too small");
    VALLE_REQUIRE(!noise || ld_noise >= 1, VALLE_E_ARG, "valle_entropix_pick: ld_noise");
    VALLE_REQUIRE(!forced || ld_forced >= vocab, VALLE_E_ARG, "valle_entropix_pick: ld_forced too small");
    if (B <= 0) return VALLE_OK;
    entropix_pick_kernel<<<B, EX_THREADS, 0, (hipStream_t)stream>>>(logits, ld, vocab, logit_row, temperature, top_k, top_p, n_draws,
                                                                    colsum, ld_colsum, noise, ld_noise, max_draws, seed, offset,
                                                                    ids, draws, forced, ld_forced);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}
