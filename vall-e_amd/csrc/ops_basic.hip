// HBM-bound elementwise / gather / reduction kernels of the AR+NAR forward (gfx950).
//   embed_sum      models/base.py:774-922 (+ AudioEmbedding :147-175, _join :101-110)
//   rmsnorm        models/arch/llama.py:64-77
//   rope_kv        models/arch/llama.py:48-61, :82-132 (+ KV-cache append)
//   attn_rowwise   models/arch/llama.py:448-456 (one query row per workgroup; parity path + decode)
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void valle_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* valle_last_error(void) { return g_err; }
extern "C" int valle_version(void) { return 1; }

// ------------------------------------------------------------------------ embed_sum
// One workgroup per output row; every lane owns 16-byte chunks of the row, so each of the
// <= 8 table rows is read with fully coalesced 16 B/lane loads.  Terms are added in index
// order and rounded to the storage type after each add (== the reference's Python sum()).
template <typename T>
__global__ __launch_bounds__(128) void embed_sum_kernel(const T* __restrict__ emb, const int32_t* __restrict__ idx,
                                                        T* __restrict__ out, int d) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int r = blockIdx.x;
    int id[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) id[j] = idx[(size_t)r * 8 + j];
    for (int c = threadIdx.x * VEC; c < d; c += 128 * VEC) {
        float acc[VEC];
        bool any = false;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (id[j] < 0) continue;
            vec_t v = *reinterpret_cast<const vec_t*>(emb + (size_t)id[j] * d + c);
            if (!any) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = to_f32<T>(v[e]);
                any = true;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = to_f32<T>(from_f32<T>(acc[e] + to_f32<T>(v[e])));
            }
        }
        vec_t o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(acc[e]);
        *reinterpret_cast<vec_t*>(out + (size_t)r * d + c) = o;
    }
}

extern "C" int valle_embed_sum(const void* emb, const int32_t* idx, void* out, int rows, int d, int dtype, void* stream) {
    VALLE_REQUIRE(emb && idx && out, VALLE_E_ARG, "valle_embed_sum: null pointer");
    VALLE_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0, VALLE_E_ARG, "valle_embed_sum: d=%d must be a multiple of 8", d);
    if (rows == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VALLE_BF16)
        embed_sum_kernel<bf16><<<rows, 128, 0, s>>>((const bf16*)emb, idx, (bf16*)out, d);
    else if (dtype == VALLE_F32)
        embed_sum_kernel<float><<<rows, 128, 0, s>>>((const float*)emb, idx, (float*)out, d);
    else
        VALLE_REQUIRE(false, VALLE_E_ARG, "valle_embed_sum: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// -------------------------------------------------------------------------- rmsnorm
// One wave per row (4 rows per 256-thread workgroup); 16-byte loads; fp32 statistics.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ out, int rows, int d, float eps) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* xr = x + (size_t)r * d;
    float ss = 0.f;
    for (int c = lane * VEC; c < d; c += 64 * VEC) {
        vec_t v = *reinterpret_cast<const vec_t*>(xr + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float f = to_f32<T>(v[e]);
            ss += f * f;
        }
    }
    ss = wave_sum(ss);
    const float rinv = 1.0f / sqrtf(ss / (float)d + eps);
    for (int c = lane * VEC; c < d; c += 64 * VEC) {
        vec_t v = *reinterpret_cast<const vec_t*>(xr + c);
        vec_t g = *reinterpret_cast<const vec_t*>(w + c);
        vec_t o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            T n = from_f32<T>(__fmul_rn(to_f32<T>(v[e]), rinv));
            o[e] = from_f32<T>(__fmul_rn(to_f32<T>(g[e]), to_f32<T>(n)));
        }
        *reinterpret_cast<vec_t*>(out + (size_t)r * d + c) = o;
    }
}

extern "C" int valle_rmsnorm(const void* x, const void* w, void* out, int rows, int d, float eps, int dtype, void* stream) {
    VALLE_REQUIRE(x && w && out, VALLE_E_ARG, "valle_rmsnorm: null pointer");
    VALLE_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0, VALLE_E_ARG, "valle_rmsnorm: d=%d must be a multiple of 8", d);
    if (rows == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = ceil_div(rows, 4);
    if (dtype == VALLE_BF16)
        rmsnorm_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, (const bf16*)w, (bf16*)out, rows, d, eps);
    else if (dtype == VALLE_F32)
        rmsnorm_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)w, (float*)out, rows, d, eps);
    else
        VALLE_REQUIRE(false, VALLE_E_ARG, "valle_rmsnorm: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// -------------------------------------------------------------------------- rope_kv
// One workgroup per packed row.  q and k heads are stored pair-interleaved (element 2j is the
// reference's dim j, element 2j+1 its dim j+32), so a rotation touches two adjacent values:
//   y_j      = x_j cos_j - x_{j+32} sin_j
//   y_{j+32} = x_{j+32} cos_j + x_j sin_j          (arch/llama.py:48-61 with emb = cat(f, f))
// Products are rounded separately (no FMA contraction) as the reference's two multiplies are.
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(T* __restrict__ qkv, const int32_t* __restrict__ pos,
                                                      const float* __restrict__ rope, const int32_t* __restrict__ cu,
                                                      int B, int H, int Dh, T* __restrict__ kc, T* __restrict__ vc,
                                                      int cache_T, const int32_t* __restrict__ slot,
                                                      const int32_t* __restrict__ pos0) {
    const int r = blockIdx.x;
    const int d = H * Dh;
    const int half = Dh >> 1;
    T* row = qkv + (size_t)r * 3 * d;
    const float* tab = rope + (size_t)pos[r] * Dh;  // [half][2]
    __shared__ int s_b;
    if (kc != nullptr && threadIdx.x == 0) {
        int lo = 0, hi = B;  // largest b with cu[b] <= r
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (cu[mid] <= r) lo = mid; else hi = mid;
        }
        s_b = lo;
    }
    __syncthreads();
    size_t cbase = 0;
    if (kc != nullptr) {
        const int b = s_b;
        const int sl = slot ? slot[b] : b;
        const int p = (pos0 ? pos0[b] : 0) + (r - cu[b]);
        cbase = ((size_t)sl * H * cache_T + p) * Dh;  // + h * cache_T * Dh
    }
    // q and k: d/2 pairs each
    for (int t = threadIdx.x; t < d; t += 256) {
        const int which = t >= (d >> 1);              // 0: q, 1: k
        const int pi = t - which * (d >> 1);          // pair index within q or k
        const int h = pi / half, j = pi - h * half;
        T* p2 = row + which * d + h * Dh + 2 * j;
        typedef T vec2 __attribute__((ext_vector_type(2)));
        vec2 v = *reinterpret_cast<vec2*>(p2);
        const float c = to_f32<T>(from_f32<T>(tab[2 * j])), s = to_f32<T>(from_f32<T>(tab[2 * j + 1]));
        const float a = to_f32<T>(v[0]), bb = to_f32<T>(v[1]);
        vec2 o;
        o[0] = from_f32<T>(__fadd_rn(__fmul_rn(a, c), __fmul_rn(-bb, s)));
        o[1] = from_f32<T>(__fadd_rn(__fmul_rn(bb, c), __fmul_rn(a, s)));
        *reinterpret_cast<vec2*>(p2) = o;
        if (which && kc != nullptr)
            *reinterpret_cast<vec2*>(kc + cbase + (size_t)h * cache_T * Dh + 2 * j) = o;
    }
    if (vc != nullptr) {
        for (int t = threadIdx.x * 2; t < d; t += 512) {
            const int h = t / Dh, j = t - h * Dh;
            typedef T vec2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<vec2*>(vc + cbase + (size_t)h * cache_T * Dh + j) =
                *reinterpret_cast<const vec2*>(row + 2 * d + t);
        }
    }
}

extern "C" int valle_rope_kv(void* qkv, const int32_t* pos, const float* rope, const int32_t* cu_seqlens,
                             int B, int M, int H, int Dh, void* kcache, void* vcache, int cache_T,
                             const int32_t* cache_slot, const int32_t* cache_pos0, int dtype, void* stream) {
    VALLE_REQUIRE(qkv && pos && rope, VALLE_E_ARG, "valle_rope_kv: null pointer");
    VALLE_REQUIRE(Dh % 2 == 0 && H > 0, VALLE_E_ARG, "valle_rope_kv: bad head shape H=%d Dh=%d", H, Dh);
    VALLE_REQUIRE((kcache == nullptr) == (vcache == nullptr), VALLE_E_ARG, "valle_rope_kv: kcache and vcache go together");
    VALLE_REQUIRE(kcache == nullptr || (cu_seqlens && cache_T > 0 && B > 0), VALLE_E_ARG, "valle_rope_kv: cache write needs cu_seqlens/cache_T");
    if (M == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VALLE_BF16)
        rope_kv_kernel<bf16><<<M, 256, 0, s>>>((bf16*)qkv, pos, rope, cu_seqlens, B, H, Dh, (bf16*)kcache, (bf16*)vcache, cache_T, cache_slot, cache_pos0);
    else if (dtype == VALLE_F32)
        rope_kv_kernel<float><<<M, 256, 0, s>>>((float*)qkv, pos, rope, cu_seqlens, B, H, Dh, (float*)kcache, (float*)vcache, cache_T, cache_slot, cache_pos0);
    else
        VALLE_REQUIRE(false, VALLE_E_ARG, "valle_rope_kv: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// --------------------------------------------------------------------- attn_rowwise
// One workgroup (4 waves) per (query row, head); Dh = 64.  A wave's 64 lanes form 8 groups of
// 8 lanes; a group owns one key per iteration (its 8 lanes each hold 8 of the 64 dims, so one
// wave-instruction reads 8 whole K rows = 1 KiB contiguous in the cache layout).  Every group
// runs its own online softmax; groups, then waves, are merged at the end.  HBM-streaming shape
// for the KV-cache decode; also the fp32 parity path for dense passes.
struct AttnRowArgs {
    const void* q;  size_t q_row_stride;           // q row of query i: q + i*q_row_stride (+ h*Dh)
    const void* k;  const void* v;                 // K/V bases
    size_t kv_row_stride, kv_head_stride;          // element strides between keys / heads
    const int32_t* item_of_row;                    // NULL: row == item (decode)
    const int32_t* cu;                             // dense: packed row offsets per item
    const int32_t* ctx_len;                        // decode: keys per item
    const int32_t* slot;                           // decode: cache slot per item (NULL: item)
    size_t kv_item_stride;                         // decode: elements between slots; dense: unused
    void* out; size_t out_row_stride;
    int causal;
    int B;
    int ctx_extra;                                 // decode: keys = ctx_len + ctx_extra
};

template <typename T, bool DECODE, int NW, int UN>
__global__ __launch_bounds__(NW * 64) void attn_rowwise_kernel(AttnRowArgs a) {
    constexpr int DH = 64;
    constexpr int VEC = 8;
    const int row = blockIdx.x, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 3, sub = lane & 7;
    int n_keys;
    const T *kb, *vb;
    if (DECODE) {
        const int sl = a.slot ? a.slot[row] : row;
        n_keys = a.ctx_len[row] + a.ctx_extra;
        kb = (const T*)a.k + sl * a.kv_item_stride + h * a.kv_head_stride;
        vb = (const T*)a.v + sl * a.kv_item_stride + h * a.kv_head_stride;
    } else {
        int lo = 0, hi = a.B;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (a.cu[mid] <= row) lo = mid; else hi = mid;
        }
        const int start = a.cu[lo];
        n_keys = a.causal ? (row - start + 1) : (a.cu[lo + 1] - start);
        if (a.ctx_len) n_keys = min(n_keys, a.ctx_len[lo]);                  // dense: optional key limit per item (padding rows behind it)
        kb = (const T*)a.k + (size_t)start * a.kv_row_stride + h * a.kv_head_stride;
        vb = (const T*)a.v + (size_t)start * a.kv_row_stride + h * a.kv_head_stride;
    }
    float q[VEC];
    {
        const T* qp = (const T*)a.q + (size_t)row * a.q_row_stride + h * DH + sub * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) q[e] = to_f32<T>(qp[e]) * 0.125f;  // 1/sqrt(64), exact
    }
    float m = -INFINITY, l = 0.f, o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = 0.f;

    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    // UN keys per group per iteration: 2*UN 16-byte loads in flight per lane
    for (int k0 = wave * 8 + g; k0 < n_keys; k0 += NW * 8 * UN) {
        vec_t kv[UN], vv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int key = k0 + NW * 8 * u;
            key = key < n_keys ? key : n_keys - 1;       // clamp: the load stays unconditional
            kv[u] = *reinterpret_cast<const vec_t*>(kb + (size_t)key * a.kv_row_stride + sub * VEC);
            vv[u] = *reinterpret_cast<const vec_t*>(vb + (size_t)key * a.kv_row_stride + sub * VEC);
        }
        float s[UN];
        float mn = m;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) t = fmaf(q[e], to_f32<T>(kv[u][e]), t);
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            s[u] = (k0 + NW * 8 * u < n_keys) ? t : -INFINITY;
            mn = fmaxf(mn, s[u]);
        }
        const float alpha = expf(m - mn);    // m = -inf on the first pass: exp(-inf) = 0 (s[0] is always valid)
        l *= alpha;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] *= alpha;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float p = expf(s[u] - mn);
            l += p;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = fmaf(p, to_f32<T>(vv[u][e]), o[e]);
        }
        m = mn;
    }
    // merge the 8 groups of the wave (lanes with equal `sub`)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, m2);
        const float a1 = (m == -INFINITY) ? 0.f : expf(m - mn);
        const float a2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float o2 = __shfl_xor(o[e], off, 64);
            o[e] = o[e] * a1 + o2 * a2;
        }
        m = mn;
    }
    __shared__ float sm_m[NW], sm_l[NW], sm_o[NW][DH];
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sm_o[wave][sub * VEC + e] = o[e];
        if (sub == 0) { sm_m[wave] = m; sm_l[wave] = l; }
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        float mm = sm_m[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mm = fmaxf(mm, sm_m[w]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float sc = (sm_m[w] == -INFINITY) ? 0.f : expf(sm_m[w] - mm);
            num += sm_o[w][threadIdx.x] * sc;
            den += sm_l[w] * sc;
        }
        ((T*)a.out)[(size_t)row * a.out_row_stride + h * DH + threadIdx.x] = from_f32<T>(num / den);
    }
}

extern "C" int valle_attn_prefill_rowwise(const void* qkv, void* out, const int32_t* cu_seqlens, int B, int M,
                                          int H, int Dh, int causal, int dtype, void* stream) {
    return valle_attn_prefill_rowwise_kv(qkv, out, cu_seqlens, nullptr, B, M, H, Dh, causal, dtype, stream);
}

// kv_lens (optional, per utterance, >= 1): a row reads at most the first kv_lens[b] rows of its utterance as keys -- the
// reference's key-padding mask on a right-padded batch (rows behind kv_lens[b] are padding: queries, never keys)
int valle_attn_prefill_rowwise_kv(const void* qkv, void* out, const int32_t* cu_seqlens, const int32_t* kv_lens, int B, int M,
                                  int H, int Dh, int causal, int dtype, void* stream) {
    VALLE_REQUIRE(qkv && out && cu_seqlens, VALLE_E_ARG, "valle_attn_prefill: null pointer");
    VALLE_REQUIRE(Dh == 64, VALLE_E_UNSUPPORTED, "valle_attn_prefill: head_dim %d (only 64 is built)", Dh);
    if (M == 0) return VALLE_OK;
    const int d = H * Dh;
    const size_t es = dtype_size(dtype);
    AttnRowArgs a{};
    a.q = qkv; a.q_row_stride = 3 * d;
    a.k = (const char*)qkv + (size_t)d * es; a.v = (const char*)qkv + (size_t)2 * d * es;
    a.kv_row_stride = 3 * d; a.kv_head_stride = Dh;
    a.cu = cu_seqlens; a.out = out; a.out_row_stride = d; a.causal = causal; a.B = B; a.ctx_len = kv_lens;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(M, H);
    if (dtype == VALLE_BF16) attn_rowwise_kernel<bf16, false, 4, 4><<<grid, 256, 0, s>>>(a);
    else if (dtype == VALLE_F32) attn_rowwise_kernel<float, false, 4, 4><<<grid, 256, 0, s>>>(a);
    else VALLE_REQUIRE(false, VALLE_E_ARG, "valle_attn_prefill: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

int valle_attn_decode_launch(const void* qkv, const void* kcache, const void* vcache, const int32_t* ctx_len,
                             const int32_t* cache_slot, void* out, int B, int H, int Dh, int cache_T,
                             int ctx_extra, int dtype, void* stream) {
    VALLE_REQUIRE(qkv && kcache && vcache && ctx_len && out, VALLE_E_ARG, "valle_attn_decode: null pointer");
    VALLE_REQUIRE(Dh == 64, VALLE_E_UNSUPPORTED, "valle_attn_decode: head_dim %d (only 64 is built)", Dh);
    if (B == 0) return VALLE_OK;
    const int d = H * Dh;
    AttnRowArgs a{};
    a.q = qkv; a.q_row_stride = 3 * d;
    a.k = kcache; a.v = vcache;
    a.kv_row_stride = Dh; a.kv_head_stride = (size_t)cache_T * Dh; a.kv_item_stride = (size_t)H * cache_T * Dh;
    a.ctx_len = ctx_len; a.slot = cache_slot;
    a.out = out; a.out_row_stride = d; a.causal = 1; a.B = B; a.ctx_extra = ctx_extra;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(B, H);
    if (dtype == VALLE_BF16)      // the streaming bf16 kernel (attn_decode.hip), row-major output
        return valle_attn_decode_bf16(qkv, (size_t)3 * d, kcache, vcache, ctx_len, cache_slot, out, B, H, cache_T, ctx_extra, 0, s);
    if (dtype == VALLE_F32) attn_rowwise_kernel<float, true, 8, 4><<<grid, 512, 0, s>>>(a);
    else VALLE_REQUIRE(false, VALLE_E_ARG, "valle_attn_decode: unknown dtype %d", dtype);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_attn_decode(const void* qkv, const void* kcache, const void* vcache, const int32_t* ctx_len,
                                 const int32_t* cache_slot, void* out, int B, int H, int Dh, int cache_T,
                                 int dtype, void* stream) {
    return valle_attn_decode_launch(qkv, kcache, vcache, ctx_len, cache_slot, out, B, H, Dh, cache_T, 0, dtype, stream);
}
