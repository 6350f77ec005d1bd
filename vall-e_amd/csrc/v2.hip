// Kernels of the version >= 7 model family ("v2": reference models/base_v2.py, models/ar_nar_v2.py) that the
// version-5 path has no counterpart for: nn.LayerNorm, the AudioEncoder's gather + level position + norm and its
// weighted level sum, and the guidance on (levels, rows, vocab) logits.  The encoder / decoder FFNs and heads are
// valle_gemm calls (epilogues GELU, RESID_BIAS, STORE + bias, F32OUT); the llama stack is valle_forward_layers.
#include "common.h"

// One wave per row (4 rows per 256-thread workgroup), fp32 statistics: mean, then the centred sum of squares (the
// biased variance nn.LayerNorm uses), y = (x - mean) * rstd * w + b rounded to the storage type once (torch computes
// the row in fp32 for low-precision inputs too).  src == nullptr: the row is x[r]; else the row is
// round(table[src[r]] + pos[(r % L)]) -- the AudioEncoder's embedding lookup plus level position (base_v2.py:138-147).
template <typename T, bool GATHER>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const int32_t* __restrict__ src,
                                                        const T* __restrict__ pos, int L,
                                                        const T* __restrict__ w, const T* __restrict__ b,
                                                        T* __restrict__ out, int rows, int d, float eps) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    // grouped form (blockIdx.y = group): the same rows under another (w, b), written rows * d further on
    w += (size_t)blockIdx.y * d; b += (size_t)blockIdx.y * d; out += (size_t)blockIdx.y * rows * d;
    const T* xr = GATHER ? x + (size_t)src[r] * d : x + (size_t)r * d;
    const T* pr = GATHER ? pos + (size_t)(r % L) * d : nullptr;
    auto load = [&](int c, float (&f)[VEC]) {
        vec_t v = *reinterpret_cast<const vec_t*>(xr + c);
        if constexpr (GATHER) {
            vec_t p = *reinterpret_cast<const vec_t*>(pr + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = to_f32<T>(from_f32<T>(to_f32<T>(v[e]) + to_f32<T>(p[e])));
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = to_f32<T>(v[e]);
        }
    };
    float s = 0.f;
    for (int c = lane * VEC; c < d; c += 64 * VEC) {
        float f[VEC];
        load(c, f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += f[e];
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
    for (int c = lane * VEC; c < d; c += 64 * VEC) {
        float f[VEC];
        load(c, f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float dl = f[e] - mean; ss += dl * dl; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)d + eps);
    for (int c = lane * VEC; c < d; c += 64 * VEC) {
        float f[VEC];
        load(c, f);
        vec_t g = *reinterpret_cast<const vec_t*>(w + c);
        vec_t bb = *reinterpret_cast<const vec_t*>(b + c);
        vec_t o;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            o[e] = from_f32<T>(__fadd_rn(__fmul_rn(__fmul_rn(f[e] - mean, rstd), to_f32<T>(g[e])), to_f32<T>(bb[e])));
        *reinterpret_cast<vec_t*>(out + (size_t)r * d + c) = o;
    }
}

extern "C" int valle_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int d, float eps, int dtype,
                               void* stream) {
    VALLE_REQUIRE(x && w && b && out, VALLE_E_ARG, "valle_layernorm: null pointer");
    VALLE_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0, VALLE_E_ARG, "valle_layernorm: d=%d must be a multiple of 8", d);
    if (rows == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VALLE_BF16)
        layernorm_kernel<bf16, false><<<ceil_div(rows, 4), 256, 0, s>>>((const bf16*)x, nullptr, nullptr, 1, (const bf16*)w, (const bf16*)b, (bf16*)out, rows, d, eps);
    else if (dtype == VALLE_F32)
        layernorm_kernel<float, false><<<ceil_div(rows, 4), 256, 0, s>>>((const float*)x, nullptr, nullptr, 1, (const float*)w, (const float*)b, (float*)out, rows, d, eps);
    else
        VALLE_REQUIRE(false, VALLE_E_ARG, "valle_layernorm: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// G LayerNorms of the same rows (the AudioDecoder's per-level norms, base_v2.py:203-206): w, b = [G][d], out = [G][rows][d]
extern "C" int valle_layernorm_grouped(const void* x, const void* w, const void* b, void* out, int rows, int d, int G, float eps,
                                       int dtype, void* stream) {
    VALLE_REQUIRE(x && w && b && out, VALLE_E_ARG, "valle_layernorm_grouped: null pointer");
    VALLE_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0 && G >= 1 && G <= 65535, VALLE_E_ARG, "valle_layernorm_grouped: rows=%d d=%d G=%d", rows, d, G);
    if (rows == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(ceil_div(rows, 4), G);
    if (dtype == VALLE_BF16)
        layernorm_kernel<bf16, false><<<grid, 256, 0, s>>>((const bf16*)x, nullptr, nullptr, 1, (const bf16*)w, (const bf16*)b, (bf16*)out, rows, d, eps);
    else if (dtype == VALLE_F32)
        layernorm_kernel<float, false><<<grid, 256, 0, s>>>((const float*)x, nullptr, nullptr, 1, (const float*)w, (const float*)b, (float*)out, rows, d, eps);
    else
        VALLE_REQUIRE(false, VALLE_E_ARG, "valle_layernorm_grouped: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// row (i * L + l) of the encoder's input reads table row row0[l] + codes[l][i] (the per-level tables sit in one
// concatenated table; row0 / n_rows travel in the kernel arguments)
#define V2_MAX_LEVELS 32
struct V2Rows { int32_t row0[V2_MAX_LEVELS]; int32_t n_rows[V2_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void v2_src_rows_kernel_k(const int32_t* __restrict__ codes, int ld_codes, int n, int L,
                                                            V2Rows t, int32_t* __restrict__ src) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n * L) return;
    const int i = r / L, l = r - i * L;
    int c = codes[(size_t)l * ld_codes + i];
    c = c < 0 ? 0 : (c >= t.n_rows[l] ? t.n_rows[l] - 1 : c);        // a code outside its table must not leave it
    src[r] = t.row0[l] + c;
}

extern "C" int valle_v2_encode_rows(const void* table, const int32_t* row0_host, const int32_t* n_rows_host, int L,
                                    const int32_t* codes, int ld_codes, int n, const void* pos, const void* norm_w,
                                    const void* norm_b, void* out, int32_t* src_scratch, int d, float eps, int dtype,
                                    void* stream) {
    VALLE_REQUIRE(table && row0_host && n_rows_host && codes && pos && norm_w && norm_b && out && src_scratch, VALLE_E_ARG,
                  "valle_v2_encode_rows: null pointer");
    VALLE_REQUIRE(L > 0 && L <= V2_MAX_LEVELS && n >= 0 && ld_codes >= n && d > 0 && d % 8 == 0, VALLE_E_ARG,
                  "valle_v2_encode_rows: L=%d n=%d ld_codes=%d d=%d", L, n, ld_codes, d);
    if (n == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    V2Rows t{};
    for (int l = 0; l < L; ++l) {
        VALLE_REQUIRE(n_rows_host[l] > 0, VALLE_E_ARG, "valle_v2_encode_rows: empty table for level %d", l);
        t.row0[l] = row0_host[l]; t.n_rows[l] = n_rows_host[l];
    }
    const int rows = n * L;
    v2_src_rows_kernel_k<<<ceil_div(rows, 256), 256, 0, s>>>(codes, ld_codes, n, L, t, src_scratch);
    if (dtype == VALLE_BF16)
        layernorm_kernel<bf16, true><<<ceil_div(rows, 4), 256, 0, s>>>((const bf16*)table, src_scratch, (const bf16*)pos, L, (const bf16*)norm_w, (const bf16*)norm_b, (bf16*)out, rows, d, eps);
    else if (dtype == VALLE_F32)
        layernorm_kernel<float, true><<<ceil_div(rows, 4), 256, 0, s>>>((const float*)table, src_scratch, (const float*)pos, L, (const float*)norm_w, (const float*)norm_b, (float*)out, rows, d, eps);
    else
        VALLE_REQUIRE(false, VALLE_E_ARG, "valle_v2_encode_rows: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// out[dst[i]] = sum_l round(y[i * L + l] * lw[l]): the products are rounded to the storage type (an elementwise multiply in
// the reference), the sum over the levels runs in fp32 and is rounded once (torch.sum's accumulation type)
template <typename T>
__global__ __launch_bounds__(128) void v2_level_sum_kernel(const T* __restrict__ y, const float* __restrict__ lw, int L,
                                                           const int32_t* __restrict__ dst, T* __restrict__ out, int d) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int i = blockIdx.x;
    const int r = dst ? dst[i] : i;
    for (int c = threadIdx.x * VEC; c < d; c += 128 * VEC) {
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int l = 0; l < L; ++l) {
            const vec_t v = *reinterpret_cast<const vec_t*>(y + ((size_t)i * L + l) * d + c);
            const float wl = lw[l];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += to_f32<T>(from_f32<T>(__fmul_rn(to_f32<T>(v[e]), wl)));
        }
        vec_t o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(acc[e]);
        *reinterpret_cast<vec_t*>(out + (size_t)r * d + c) = o;
    }
}

extern "C" int valle_v2_level_sum(const void* y, const float* lw, int L, int n, const int32_t* dst_rows, void* out, int d,
                                  int dtype, void* stream) {
    VALLE_REQUIRE(y && lw && out, VALLE_E_ARG, "valle_v2_level_sum: null pointer");
    VALLE_REQUIRE(L > 0 && n >= 0 && d > 0 && d % 8 == 0, VALLE_E_ARG, "valle_v2_level_sum: L=%d n=%d d=%d", L, n, d);
    if (n == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VALLE_BF16) v2_level_sum_kernel<bf16><<<n, 128, 0, s>>>((const bf16*)y, lw, L, dst_rows, (bf16*)out, d);
    else if (dtype == VALLE_F32) v2_level_sum_kernel<float><<<n, 128, 0, s>>>((const float*)y, lw, L, dst_rows, (float*)out, d);
    else VALLE_REQUIRE(false, VALLE_E_ARG, "valle_v2_level_sum: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// Guidance on (levels, rows, vocab) logits (samplers.py:177-191 with a 3-D input: dims = (1,), the ROWS): one workgroup per
// (utterance, level, 64 vocabulary columns); a lane owns a column, the eight waves split the rows (wave w: rows w, w + 8, ...).
// Unbiased standard deviations by mean-then-centred-squares; one row gives nan, as torch.std does.  An utterance of up to
// 8 x CC_MAXR rows keeps its column's values in registers between the three passes (one read of pos / neg instead of three);
// longer ones re-read them.  Both forms do the same additions in the same order.
#define CC_MAXR 64
__global__ __launch_bounds__(512) void cfg_combine_cols_kernel(float* __restrict__ pos, const float* __restrict__ neg, int ld,
                                                               int vocab, int rows_total, const int32_t* __restrict__ row_off,
                                                               int L, float strength, float rescale) {
    const int i = blockIdx.y / L, l = blockIdx.y - i * L;
    const int r0 = row_off[i], n = row_off[i + 1] - r0;
    const int c = threadIdx.x & 63, v = blockIdx.x * 64 + c, w = threadIdx.x >> 6;
    const bool live = v < vocab;
    float* p = pos + ((size_t)l * rows_total + r0) * ld + v;
    const float* q = neg + ((size_t)l * rows_total + r0) * ld + v;
    __shared__ float red[2][8][64];
    auto fold = [&](float a, float b, float& oa, float& ob) {          // sums over the 8 waves, in wave order
        __syncthreads();
        red[0][w][c] = a; red[1][w][c] = b;
        __syncthreads();
        oa = 0.f; ob = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { oa += red[0][u][c]; ob += red[1][u][c]; }
    };
    auto mix = [&](float a, float b) { return __fadd_rn(b, __fmul_rn(__fadd_rn(a, -b), strength)); };
    if (n <= 8 * CC_MAXR) {                                            // (uniform per workgroup)
        float pa[CC_MAXR], sa[CC_MAXR];
#pragma unroll
        for (int k = 0; k < CC_MAXR; ++k) {
            const int r = w + 8 * k;
            float a = 0.f, b = 0.f;
            if (live && r < n) { a = p[(size_t)r * ld]; b = q[(size_t)r * ld]; }
            pa[k] = a; sa[k] = mix(a, b);
        }
        float sp = 0.f, ssum = 0.f;
#pragma unroll
        for (int k = 0; k < CC_MAXR; ++k)
            if (live && w + 8 * k < n) { sp += pa[k]; ssum += sa[k]; }
        float tp, ts;
        fold(sp, ssum, tp, ts);
        const float mp = tp / (float)n, ms = ts / (float)n;
        float factor = 1.f;
        if (rescale > 0.f) {
            float vp = 0.f, vs = 0.f;
#pragma unroll
            for (int k = 0; k < CC_MAXR; ++k)
                if (live && w + 8 * k < n) { vp += (pa[k] - mp) * (pa[k] - mp); vs += (sa[k] - ms) * (sa[k] - ms); }
            float qp, qs;
            fold(vp, vs, qp, qs);
            factor = __fadd_rn(__fmul_rn(rescale, sqrtf(qp / (float)(n - 1)) / sqrtf(qs / (float)(n - 1))), 1.f - rescale);
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < CC_MAXR; ++k) {
                const int r = w + 8 * k;
                if (r < n) p[(size_t)r * ld] = rescale > 0.f ? __fmul_rn(sa[k], factor) : sa[k];
            }
        }
        return;
    }
    float sp = 0.f, ssum = 0.f;
    if (live)
        for (int r = w; r < n; r += 8) {
            const float a = p[(size_t)r * ld], b = q[(size_t)r * ld];
            sp += a; ssum += mix(a, b);
        }
    float tp, ts;
    fold(sp, ssum, tp, ts);
    const float mp = tp / (float)n, ms = ts / (float)n;
    float factor = 1.f;
    if (rescale > 0.f) {
        float vp = 0.f, vs = 0.f;
        if (live)
            for (int r = w; r < n; r += 8) {
                const float a = p[(size_t)r * ld], sv = mix(a, q[(size_t)r * ld]);
                vp += (a - mp) * (a - mp); vs += (sv - ms) * (sv - ms);
            }
        float qp, qs;
        fold(vp, vs, qp, qs);
        factor = __fadd_rn(__fmul_rn(rescale, sqrtf(qp / (float)(n - 1)) / sqrtf(qs / (float)(n - 1))), 1.f - rescale);
    }
    if (live)
        for (int r = w; r < n; r += 8) {
            const float sv = mix(p[(size_t)r * ld], q[(size_t)r * ld]);
            p[(size_t)r * ld] = rescale > 0.f ? __fmul_rn(sv, factor) : sv;
        }
}

extern "C" int valle_cfg_combine_cols(float* pos, const float* neg, int ld, int vocab, int L, int rows_total,
                                      const int32_t* row_off, int B, float strength, float rescale, void* stream) {
    VALLE_REQUIRE(pos && neg && row_off, VALLE_E_ARG, "valle_cfg_combine_cols: null pointer");
    VALLE_REQUIRE(ld >= vocab && vocab > 0 && L > 0 && B >= 0 && rows_total >= 0, VALLE_E_ARG,
                  "valle_cfg_combine_cols: ld=%d vocab=%d L=%d B=%d", ld, vocab, L, B);
    if (B == 0 || rows_total == 0) return VALLE_OK;
    VALLE_REQUIRE((long)B * L <= 65535, VALLE_E_ARG, "valle_cfg_combine_cols: B * L = %ld exceeds the grid", (long)B * L);
    dim3 grid(ceil_div(vocab, 64), B * L);
    cfg_combine_cols_kernel<<<grid, 512, 0, (hipStream_t)stream>>>(pos, neg, ld, vocab, rows_total, row_off, L, strength, rescale);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}
