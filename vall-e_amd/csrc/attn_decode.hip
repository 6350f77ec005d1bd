// KV-cache attention of the AR decode step (gfx950, bf16, head_dim 64): one new query per
// (utterance, head) against ctx cached keys -- the KV-cache restatement of the reference's full
// recompute (models/base.py:521 use_cache=False; SDPA call at models/arch/llama.py:448-456).
//
// HBM-bound: every cached K and V row is read exactly once.  One workgroup (NW waves) per
// (utterance, head).  Keys go to the waves in chunks of 8 (chunk j of a pass: wave j % NW); a pass is
// MAXC chunks per wave and the NEXT pass's K and V rows are requested before the current pass is
// consumed (two register buffers), so up to 2 x MAXC x 2 x 16 B per lane are in flight.  Lane mapping:
// 8 lanes share a key (16 B = 8 dims each), so one wave-instruction reads 8 whole 128-B rows = 1 KiB
// contiguous of the [slot][H][cache_T][64] cache.  Arithmetic on packed bf16 pairs (v_dot2_f32_bf16):
// q.k over dim pairs, P.V over KEY pairs (two chunks' V dwords interleaved by v_perm, the two probabilities
// rounded to bf16 as the reference's bf16 SDPA does before its P.V matmul).
#include "common.h"

#ifdef VALLE_NO_NT
#define LOAD_KV(p) (*(p))
#else
#define LOAD_KV(p) __builtin_nontemporal_load(p)   // every cached row is read once per step
#endif
struct AttnDecArgs {
    const bf16* q; size_t q_row_stride;          // q of utterance b: q + b*q_row_stride + h*64
    const bf16* k; const bf16* v;                // caches [slot][H][cache_T][64]
    const int32_t* ctx_len; const int32_t* slot;
    bf16* out;                                   // [B][H*64] row-major, or fragment order (dec_linear.hip)
    int H, cache_T, ctx_extra, out_frag;
#ifdef VALLE_STAMPS
    unsigned long long* stamps;
#endif
};
#ifdef VALLE_STAMPS
unsigned long long* valle_stamp_buf = nullptr;
int valle_stamp_launch = 0;
#endif

__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_ror8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); }


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// K and V rows of one pass: chunk c of wave w covers keys p0 + (c*NW + w)*8 .. +8; rows past `bound` re-read row bound-1
template <int MAXC, int NW>
__device__ __forceinline__ void attn_dec_issue(const bf16* kb, const bf16* vb, int p0, int wave, int g, int bound,
                                               u32x4 (&kk)[MAXC], u32x4 (&vv)[MAXC]) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        int key = p0 + (c * NW + wave) * 8 + g;
        key = key < bound ? key : (bound > 0 ? bound - 1 : 0);    // (an empty context reads row 0 and masks it)
        kk[c] = LOAD_KV(reinterpret_cast<const u32x4*>(kb + (size_t)key * 64));
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        int key = p0 + (c * NW + wave) * 8 + g;
        key = key < bound ? key : (bound > 0 ? bound - 1 : 0);
        vv[c] = LOAD_KV(reinterpret_cast<const u32x4*>(vb + (size_t)key * 64));
    }
}

// online-softmax update with one pass (scores in log2 units)
template <int MAXC, int NW>
__device__ __forceinline__ void attn_dec_consume(int p0, int wave, int g, int n_keys, const u32x4& qb,
                                                 u32x4 (&kk)[MAXC], u32x4 (&vv)[MAXC], float& m, float& l, float (&o)[8]) {
    float s[MAXC];
    float mloc = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int key = p0 + (c * NW + wave) * 8 + g;
        float t = dot2bf(kk[c].x, qb.x, 0.f);
        t = dot2bf(kk[c].y, qb.y, t);
        t = dot2bf(kk[c].z, qb.z, t);
        t = dot2bf(kk[c].w, qb.w, t);
        t += dpp_xor1(t);
        t += dpp_xor2(t);
        t += dpp_half_mirror(t);
        s[c] = key < n_keys ? t * 1.44269504088896341f : -INFINITY;
        mloc = fmaxf(mloc, s[c]);
        if (key >= n_keys) vv[c] = u32x4{0u, 0u, 0u, 0u};     // rows past the context may hold anything (0 x NaN)
    }
    mloc = fmaxf(mloc, dpp_ror8(mloc));                                     // over the 8 key groups of the wave
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mn = fmaxf(m, mloc);
    if (mn != -INFINITY) {                                                  // wave-uniform
        const float alpha = __builtin_amdgcn_exp2f(m - mn);                 // m = -inf -> 0
        l *= alpha;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
        for (int c = 0; c < MAXC; c += 2) {
            // (an odd chunk count pairs its last chunk with an empty one: the same arithmetic as a masked chunk)
            const bool pair = c + 1 < MAXC;
            const int c1 = pair ? c + 1 : c;
            const float pa = __builtin_amdgcn_exp2f(s[c] - mn), pb = pair ? __builtin_amdgcn_exp2f(s[c1] - mn) : 0.f;   // masked keys: 0
            l += pa + pb;
            const bf16x2_t pp = {(bf16)pa, (bf16)pb};
            const uint32_t ppu = __builtin_bit_cast(uint32_t, pp);
            const uint32_t va[4] = {vv[c].x, vv[c].y, vv[c].z, vv[c].w};
            const uint32_t vb2[4] = {pair ? vv[c1].x : 0u, pair ? vv[c1].y : 0u, pair ? vv[c1].z : 0u, pair ? vv[c1].w : 0u};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t lo = __builtin_amdgcn_perm(vb2[i], va[i], 0x05040100u);   // (v_c[2i],   v_c+1[2i])
                const uint32_t hi = __builtin_amdgcn_perm(vb2[i], va[i], 0x07060302u);   // (v_c[2i+1], v_c+1[2i+1])
                o[2 * i] = dot2bf(lo, ppu, o[2 * i]);
                o[2 * i + 1] = dot2bf(hi, ppu, o[2 * i + 1]);
            }
        }
        m = mn;
    }
}

__device__ __forceinline__ void attn_dec_scale_q(u32x4& qb) {      // q * 1/sqrt(64): exact in bf16
    bf16x8 qv = __builtin_bit_cast(bf16x8, qb);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = (bf16)((float)qv[e] * 0.125f);
    qb = __builtin_bit_cast(u32x4, qv);
}

// the whole context in one pass of MC chunks per wave (n_keys <= MC x NW x 8)
template <int MC, int NW>
__device__ __forceinline__ void attn_dec_single(const bf16* kb, const bf16* vb, int wave, int g, int n_keys, u32x4 qb,
                                                float& m, float& l, float (&o)[8]) {
    u32x4 kk[MC], vv[MC];
    attn_dec_issue<MC, NW>(kb, vb, 0, wave, g, n_keys, kk, vv);
    attn_dec_scale_q(qb);
    attn_dec_consume<MC, NW>(0, wave, g, n_keys, qb, kk, vv, m, l, o);
}

template <int MAXC, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_bf16_kernel(const bf16* __restrict__ pQ, const bf16* __restrict__ pK,
                                                                   const bf16* __restrict__ pV, const int32_t* __restrict__ pCtx,
                                                                   const int32_t* __restrict__ pSlot, bf16* __restrict__ pOut,
                                                                   int H, int cache_T, AttnDecArgs a) {
    // the first 14 argument dwords arrive in SGPRs (kernarg preload, see the Makefile)
#ifdef VALLE_STAMPS
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = 0, t2 = 0;
#endif
    constexpr int PK = NW * 8 * MAXC;                                       // keys per pass
    const int b = blockIdx.x, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 3, sub = lane & 7;
    // q first: vector loads return in order, so a q requested behind the K/V rows would arrive behind all of them and the
    // first score could not start before the whole pass has landed
    u32x4 qb = *reinterpret_cast<const u32x4*>(pQ + (size_t)b * a.q_row_stride + h * 64 + (lane & 7) * 8);
    const int sl = pSlot ? pSlot[b] : b;
    const int ctx = pCtx[b];
    if (ctx < 0) return;        // finished utterance (valle_ar_state.early_exit stores ~ctx_len): the whole workgroup leaves
                                // before any barrier, its K/V rows stay in HBM, its output row keeps its last value
    const int n_keys = min(ctx + a.ctx_extra, cache_T);      // (a length beyond the cache rows is the caller's error; never read past them)
    const int bound = n_keys;   // (a host bound on the batch's longest context, so that the loads need not wait for ctx_len[b],
                                // measured no faster: 8.74 vs 8.70 us)
    const size_t base = ((size_t)sl * H + h) * cache_T * 64 + sub * 8;
    const bf16* kb = pK + base;
    const bf16* vb = pV + base;
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    // A context that fits one pass of fewer chunks takes a straight-line body of exactly that many (a pass of MAXC chunks
    // requests and scores MAXC x NW x 8 keys whatever the context: at 311..761 keys of 768 up to 60 % of that was padding).
    // The choice is uniform over the workgroup; every body is branch-free, so loads are still consumed as they land.
#ifndef ATTN_DEC_NO_SHORT     // (dev: the one-size body alone, for the comparison in profiles/README.md)
    if constexpr (MAXC == 12) {
        switch ((bound + NW * 8 - 1) / (NW * 8)) {          // chunks per wave the context needs
            case 0: case 1: case 2: case 3: case 4: attn_dec_single<4, NW>(kb, vb, wave, g, n_keys, qb, m, l, o); break;
            case 5: case 6: attn_dec_single<6, NW>(kb, vb, wave, g, n_keys, qb, m, l, o); break;
            case 7: attn_dec_single<7, NW>(kb, vb, wave, g, n_keys, qb, m, l, o); break;
            case 8: attn_dec_single<8, NW>(kb, vb, wave, g, n_keys, qb, m, l, o); break;
            case 9: attn_dec_single<9, NW>(kb, vb, wave, g, n_keys, qb, m, l, o); break;
            case 10: attn_dec_single<10, NW>(kb, vb, wave, g, n_keys, qb, m, l, o); break;
            case 11: attn_dec_single<11, NW>(kb, vb, wave, g, n_keys, qb, m, l, o); break;
            default: break;
        }
    }
    const bool general = MAXC != 12 || bound > NW * 8 * 11;
#else
    const bool general = true;
#endif
    u32x4 kA[MAXC], vA[MAXC], kB[MAXC], vB[MAXC];
    if (general) {
    attn_dec_issue<MAXC, NW>(kb, vb, 0, wave, g, bound, kA, vA);
    attn_dec_scale_q(qb);
    }

    for (int p0 = 0; general && p0 < bound; p0 += 2 * PK) {
        const bool more = p0 + PK < bound;                                  // wave-uniform
        if (more) attn_dec_issue<MAXC, NW>(kb, vb, p0 + PK, wave, g, bound, kB, vB);
#ifdef VALLE_STAMPS
        if (p0 == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memrealtime(); }
#endif
        attn_dec_consume<MAXC, NW>(p0, wave, g, n_keys, qb, kA, vA, m, l, o);
        if (more) {
            if (p0 + 2 * PK < bound) attn_dec_issue<MAXC, NW>(kb, vb, p0 + 2 * PK, wave, g, bound, kA, vA);
            attn_dec_consume<MAXC, NW>(p0 + PK, wave, g, n_keys, qb, kB, vB, m, l, o);
        }
    }
#ifdef VALLE_STAMPS
    t2 = __builtin_amdgcn_s_memrealtime();
#endif
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
        const float scl = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - mm);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float sc = __shfl(scl, w, 64);
            num += ov[w] * sc;
            den += __shfl(lw_l, w, 64) * sc;
        }
        const int col = h * 64 + threadIdx.x;
        const size_t off = a.out_frag ? (size_t)(b >> 4) * 16 * H * 64 + dec_frag_off<DEC_S_O>(b & 15, col, H * 64)
                                      : (size_t)b * H * 64 + col;
        pOut[off] = (bf16)(num / den);   // (as a non-temporal store: 9.33 vs 8.78 us)
#ifdef VALLE_STAMPS
        if (a.stamps) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) { unsigned long long* q = a.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4; q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; }
        }
#endif
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
#ifdef VALLE_STAMPS
    a.stamps = valle_stamp_next();
#endif
    dim3 grid(B, H);
    // Up to one workgroup per CU (B x H <= 256) a workgroup keeps 2 x 12 chunks per wave in flight; with more workgroups
    // two of them share a CU (6-chunk passes, 138 VGPRs) and one's loads overlap the other's arithmetic.  Measured
    // (tools/micro_dec, tools/time_configs.py): 16 utterances at 537 keys 8.70 vs 9.04 us per launch for 12 vs 6 chunks,
    // 32 utterances at ~2,000 keys 808 vs 830 us per decode step for 6 vs 12; 16-wave workgroups lose in both (10.2-11.7 us).
#ifdef ATTN_DEC_MAXC      // dev override (tools/micro_dec builds)
    attn_decode_bf16_kernel<ATTN_DEC_MAXC, ATTN_DEC_NW><<<grid, ATTN_DEC_NW * 64, 0, s>>>(a.q, a.k, a.v, a.ctx_len, a.slot, a.out, a.H, a.cache_T, a);
#else
    if (B * H <= 256) attn_decode_bf16_kernel<12, 8><<<grid, 512, 0, s>>>(a.q, a.k, a.v, a.ctx_len, a.slot, a.out, a.H, a.cache_T, a);
    else attn_decode_bf16_kernel<6, 8><<<grid, 512, 0, s>>>(a.q, a.k, a.v, a.ctx_len, a.slot, a.out, a.H, a.cache_T, a);
#endif
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}
