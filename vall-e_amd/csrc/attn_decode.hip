// KV-cache attention of the AR decode step (gfx950, bf16, head_dim 64): one new query per
// (utterance, head) against ctx cached keys -- the KV-cache restatement of the reference's full
// recompute (models/base.py:521 use_cache=False; SDPA call at models/arch/llama.py:448-456).
//
// HBM-bound: every cached K and V row is read exactly once.  One workgroup (8 waves) per
// (utterance, head); the keys of a pass are dealt to the waves in contiguous ranges and ALL of a
// wave's K and V rows are requested up front (MAXC x 2 x 16 B per lane in flight), then consumed in
// arrival order.  Lane mapping: 8 lanes share a key (16 B = 8 dims each), so one wave-instruction
// reads 8 whole 128-B rows = 1 KiB contiguous of the [slot][H][cache_T][64] cache.
#include "common.h"

struct AttnDecArgs {
    const bf16* q; size_t q_row_stride;          // q of utterance b: q + b*q_row_stride + h*64
    const bf16* k; const bf16* v;                // caches [slot][H][cache_T][64]
    const int32_t* ctx_len; const int32_t* slot;
    bf16* out;                                   // [B][H*64] row-major, or fragment order (dec_linear.hip)
    int H, cache_T, ctx_extra, out_frag;
};

__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_ror8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); }

template <int MAXC, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_bf16_kernel(AttnDecArgs a) {
    const int b = blockIdx.x, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 3, sub = lane & 7;
    const int sl = a.slot ? a.slot[b] : b;
    const int n_keys = a.ctx_len[b] + a.ctx_extra;
    const size_t base = ((size_t)sl * a.H + h) * a.cache_T * 64 + sub * 8;
    const bf16* kb = a.k + base;
    const bf16* vb = a.v + base;
    uint4 qbits = *reinterpret_cast<const uint4*>(a.q + (size_t)b * a.q_row_stride + h * 64 + sub * 8);
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;

    for (int p0 = 0; p0 < n_keys; p0 += NW * 8 * MAXC) {
        const int nk = min(n_keys - p0, NW * 8 * MAXC);                    // keys of this pass
        const int per = ((nk + NW * 8 - 1) / (NW * 8)) * 8;                // keys per wave (multiple of 8)
        const int k_lo = p0 + wave * per;
        const int k_hi = min(p0 + nk, k_lo + per);
        const int kmax = k_hi > k_lo ? k_hi - 1 : n_keys - 1;             // a wave without keys re-reads the last row
        bf16x8 kv[MAXC], vv[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            int key = k_lo + c * 8 + g;
            key = key < k_hi ? key : kmax;                                  // clamped: the load stays unconditional
            kv[c] = *reinterpret_cast<const bf16x8*>(kb + (size_t)key * 64);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            int key = k_lo + c * 8 + g;
            key = key < k_hi ? key : kmax;
            vv[c] = *reinterpret_cast<const bf16x8*>(vb + (size_t)key * 64);
        }
        __builtin_amdgcn_sched_barrier(0);          // every K and V row of the pass is in flight before any use
        asm volatile("" : "+v"(qbits.x), "+v"(qbits.y), "+v"(qbits.z), "+v"(qbits.w));   // keeps the wait for q behind the loads above
        const bf16x8 qv = __builtin_bit_cast(bf16x8, qbits);
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (float)qv[e] * 0.125f;          // 1/sqrt(64), exact
        float s[MAXC];
        float mloc = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t = fmaf(q[e], (float)kv[c][e], t);
            t += dpp_xor1(t);
            t += dpp_xor2(t);
            t += dpp_half_mirror(t);
            s[c] = (k_lo + c * 8 + g < k_hi) ? t : -INFINITY;
            mloc = fmaxf(mloc, s[c]);
        }
        mloc = fmaxf(mloc, dpp_ror8(mloc));                                 // over the 8 key groups of the wave
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mn = fmaxf(m, mloc);
        if (mn != -INFINITY) {                                              // wave-uniform
            const float alpha = expf(m - mn);                               // m = -inf -> 0
            l *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const float p = expf(s[c] - mn);                            // masked keys: exp(-inf) = 0
                l += p;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(p, (float)vv[c][e], o[e]);
            }
            m = mn;
        }
    }
    // m is wave-uniform: the 8 key groups simply add up
    l += dpp_ror8(l);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = o[e];
        t += dpp_ror8(t);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        o[e] = t;
    }
    __shared__ float sm_m[NW], sm_l[NW], sm_o[NW][64];
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sm_o[wave][sub * 8 + e] = o[e];
        if (sub == 0) { sm_m[wave] = m; sm_l[wave] = l; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // cross-wave merge by wave 0: lane w < NW owns wave w's rescale factor (one exponential per lane instead of NW
        // dependent ones per lane), broadcast by readlane; every LDS operand is requested before the first use
        const int lw = threadIdx.x < NW ? threadIdx.x : NW - 1;
        const float mw = sm_m[lw], lw_l = sm_l[lw];
        float ov[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) ov[w] = sm_o[w][threadIdx.x];
        float mm = mw;
#pragma unroll
        for (int o = NW >> 1; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));   // lanes >= NW mirror lane NW-1: harmless
        mm = __shfl(mm, 0, 64);
        const float scl = (mw == -INFINITY) ? 0.f : expf(mw - mm);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float sc = __shfl(scl, w, 64);
            num += ov[w] * sc;
            den += __shfl(lw_l, w, 64) * sc;
        }
        const int col = h * 64 + threadIdx.x;
        const size_t off = a.out_frag ? (size_t)(b >> 4) * 16 * a.H * 64 + (size_t)(((((col >> 5) << 6) + (((col >> 3) & 3) << 4) + (b & 15)) << 3) + (col & 7))
                                      : (size_t)b * a.H * 64 + col;
        a.out[off] = (bf16)(num / den);
    }
}

int valle_attn_decode_bf16(const void* q, size_t q_row_stride, const void* kcache, const void* vcache,
                           const int32_t* ctx_len, const int32_t* cache_slot, void* out, int B, int H, int cache_T,
                           int ctx_extra, int out_frag, hipStream_t s) {
    VALLE_REQUIRE(q && kcache && vcache && ctx_len && out, VALLE_E_ARG, "valle_attn_decode: null pointer");
    if (B == 0) return VALLE_OK;
    AttnDecArgs a{};
    a.q = (const bf16*)q; a.q_row_stride = q_row_stride; a.k = (const bf16*)kcache; a.v = (const bf16*)vcache;
    a.ctx_len = ctx_len; a.slot = cache_slot; a.out = (bf16*)out; a.H = H; a.cache_T = cache_T;
    a.ctx_extra = ctx_extra; a.out_frag = out_frag;
    dim3 grid(B, H);
    // Long caches: 8 waves x 6 chunks = 384 keys per pass.  At 94 VGPRs two such workgroups share a CU, so one's loads
    // overlap the other's arithmetic; the 16-wave form (768 keys per pass, one workgroup per CU) measured 1-8 % slower
    // from 330 to 2,000 cached positions (tools/run_attn_dec_variants.sh; VALLE_ATTN_DEC=0 brings it back, dev); 4 waves
    // x 6 or x 12 chunks are no faster, and at 2,000 positions x 32 utterances every form takes 49.4 us = 5.3 TB/s.
    static int variant = -1;
    if (variant < 0) { const char* e = getenv("VALLE_ATTN_DEC"); variant = e ? atoi(e) : 1; }
    if (cache_T <= 8 * 8 * 8) attn_decode_bf16_kernel<8, 8><<<grid, 512, 0, s>>>(a);
    else if (variant == 0) attn_decode_bf16_kernel<6, 16><<<grid, 1024, 0, s>>>(a);
    else attn_decode_bf16_kernel<6, 8><<<grid, 512, 0, s>>>(a);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}
