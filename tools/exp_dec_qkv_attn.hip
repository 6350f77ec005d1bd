// EXPERIMENT, NOT BUILT INTO THE LIBRARY (kept for the record; needs QkvAttnArgs = {x, W, ssq_in, eps, rope_m, ctx_len,
// slot, kc, vc, cache_T, H, B, n_tiles, exch, epoch, err_flag, out}).  Measured on MI355X, bench configs[1]:
// 385.0 us per decode step against 387.3 us for the two separate launches -- the hand-off works, but the projection's
// loads, the published words and the polls all go through the same per-CU memory pipeline as the 137 KB cache
// stream, so whichever is issued second waits for the first (in-kernel timestamps: projection 3.5 us, cache stream
// 8.9 us, merge 1.8 us, back to back).  See profiles/README.md.
// AR decode step, first two phases of a layer in ONE launch (gfx950, bf16, head_dim 64, 12..16 utterances --
// one projection tile per workgroup needs B * H >= 3 d / 16):
//   RMSNorm + QKV projection + RoPE + KV-cache append   (arch/llama.py:64-77, :307-309, :48-61)
//   KV-cache attention of the new token                  (SDPA at arch/llama.py:448-456)
// = dec_linear.hip's QKV epilogue and attn_decode.hip's kernel, with the kernel boundary between them
// replaced by a hand-off INSIDE the launch.
//
// Why this seam and no other: a kernel boundary costs ~2.9 us on this chip and an in-launch all-to-all of the
// 32 KB activation matrix costs more (profiles/README.md), but a (utterance, head) attention workgroup needs
// only 384 bytes from the projection -- its q row and the newest k and v rows -- and nothing else keeps it
// from starting: every CACHED key and value can be requested at launch.  So each workgroup
//   1. (waves 4..15) puts its three quarters of the K/V cache of its (utterance, head) in flight,
//   2. (waves 0..3) computes one 16-column tile of the QKV projection for all utterances -- K split over
//      the 4 waves, reduced through LDS -- applies 1/rms, RoPE, appends k/v to the cache for later steps
//      and publishes the tile for THIS step as flag-in-data words (4 bytes of data + the 4-byte epoch per
//      8-byte word, stored sc1; no drain, no counter),
//   3. (waves 2..3, before they queue any cache load: a wave's loads return in order) polls the 96 words
//      of its (utterance, head) until their epoch matches (sc1 loads, bounded); waves 0..3 then request
//      their quarter of the cache, and
//   4. all 16 waves run the streaming attention over the cached keys, wave 0 adds the one new key.
// The projection and the hand-off hide under the cache stream, which is the HBM-bound part.
//
// Progress: every workgroup publishes before it polls, and the grid (B*H workgroups of 1024 threads, at most
// 256) is resident at once on a 256-CU chip with room for two such workgroups per CU, so no workgroup
// waits for one that cannot start.  Polls are bounded all the same: on expiry the workgroup raises
// err_flag and goes on with what it has, so a launch always drains.
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ int qa_frag_off(int m, int k) { return ((((k >> 5) << 6) + (((k >> 3) & 3) << 4) + m) << 3) + (k & 7); }
__device__ __forceinline__ float qa_dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float qa_dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float qa_dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float qa_dpp_ror8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); }

// SW = 64-wide k steps per wave of the projection (K = 4 * SW * 64): 4 for d_model 1024, 1 for 256.
// MAXC = cached-key chunks a lane keeps in flight (16 waves x 8 keys x MAXC keys per pass).  In waves 0..3 the
// projection comes first and the cache registers after it, so the register budget (128 at 16 waves per workgroup)
// is the larger of the two roles, not their sum.  Both branches pass the same two workgroup barriers.
template <int SW, int MAXC>
__global__ __launch_bounds__(1024) void dec_qkv_attn_kernel(QkvAttnArgs a) {
    constexpr int K = 4 * SW * 64;
    constexpr int NW = 16;
    const int b = blockIdx.x, h = blockIdx.y;
    const int wg = b * a.H + h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 3, sub = lane & 7;
    __shared__ float part[4][16][17];              // [wave][output column][utterance]
    __shared__ float s_q[64], s_k[64], s_v[64];
    __shared__ float sm_m[NW], sm_l[NW], sm_o[NW][64];

// workgroup barrier that orders LDS only: __syncthreads() also waits for every outstanding global load (its fence
// is vmcnt(0)), which here would be the whole cache stream
#define QA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- cached keys of this (utterance, head)
    const int sl = a.slot ? a.slot[b] : b;
    const int n_keys = a.ctx_len[b];              // cached positions; the new token goes to position n_keys
    const size_t cbase = ((size_t)sl * a.H + h) * a.cache_T * 64 + sub * 8;
    const bf16* kb = a.kc + cbase;
    const bf16* vb = a.vc + cbase;
    bf16x8 kv[MAXC], vv[MAXC];
    int k_lo = 0, k_hi = 0;
    auto request_cache = [&](int p0) {
        const int nk = min(n_keys - p0, NW * 8 * MAXC);
        const int per = ((nk + NW * 8 - 1) / (NW * 8)) * 8;
        k_lo = p0 + wave * per;
        k_hi = min(p0 + nk, k_lo + per);
        const int kmax = k_hi > k_lo ? k_hi - 1 : (n_keys > 0 ? n_keys - 1 : 0);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            int key = k_lo + c * 8 + g;
            key = key < k_hi ? key : kmax;
            kv[c] = *reinterpret_cast<const bf16x8*>(kb + (size_t)key * 64);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            int key = k_lo + c * 8 + g;
            key = key < k_hi ? key : kmax;
            vv[c] = *reinterpret_cast<const bf16x8*>(vb + (size_t)key * 64);
        }
    };

    if (wave < 4) {
        // ================= projection tile `wg` (if any), publish, poll; only then this wave's cache share =================
        const int tile = wg;
        const bool has_tile = tile < a.n_tiles;   // workgroup-uniform
        const int fr = lane & 15, fq = lane >> 4;
        const int n = tile * 16 + fq * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float scale = 1.f;
        f32x4 rp = {};
        int cpos = 0, csl = 0;
        if (has_tile) {
            const int m = fr, mc = m < a.B ? m : a.B - 1;
            f32x4 sq[SW];
            if (wave == 0) {                      // epilogue operands first
                cpos = a.ctx_len[mc];
                csl = a.slot ? a.slot[mc] : mc;
                rp = *reinterpret_cast<const f32x4*>(a.rope_m + (size_t)mc * 64 + (n & 63));   // (cos,sin) of pairs n/2, n/2+1
#pragma unroll
                for (int i = 0; i < SW; ++i) sq[i] = *reinterpret_cast<const f32x4*>(a.ssq_in + (size_t)m * (K / 16) + fq * (SW * 4) + i * 4);
            }
            const bf16* xp = a.x + ((size_t)wave * SW * 128 + lane) * 8;
            const bf16* wp = a.W + ((size_t)tile * (4 * SW) + wave * SW) * 1024 + lane * 8;
            bf16x8 x0[SW], x1[SW], w0[SW], w1[SW];
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                x0[s] = *reinterpret_cast<const bf16x8*>(xp + s * 1024);
                x1[s] = *reinterpret_cast<const bf16x8*>(xp + s * 1024 + 512);
            }
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                w0[s] = *reinterpret_cast<const bf16x8*>(wp + s * 1024);
                w1[s] = *reinterpret_cast<const bf16x8*>(wp + s * 1024 + 512);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (wave == 0) {
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < SW; ++i) ss += (sq[i][0] + sq[i][1]) + (sq[i][2] + sq[i][3]);
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                scale = 1.0f / sqrtf(ss * (1.0f / (float)K) + a.eps);
            }
#pragma unroll
            for (int s = 0; s < SW; ++s) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[s], x0[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s], x1[s], acc, 0, 0, 0);
            }
            if (wave != 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][fq * 4 + r][fr] = acc[r];
            }
        }
        QA_LDS_BARRIER();                         // barrier 1: the partial sums are in LDS
        if (has_tile && wave == 0 && fr < a.B) {
            const int m = fr;
#pragma unroll
            for (int w = 1; w < 4; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += part[w][fq * 4 + r][fr];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= scale;
            const int d = a.H * 64;
            const int which = n / d, c = n - which * d, hh = c >> 6, e = c & 63;
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16)acc[r];
            if (which < 2) {                      // RoPE on the two adjacent pairs (arch/llama.py:48-61)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float cs = (float)(bf16)rp[2 * p], sn = (float)(bf16)rp[2 * p + 1];
                    const float u = (float)ov[2 * p], v = (float)ov[2 * p + 1];
                    ov[2 * p] = (bf16)(__fadd_rn(__fmul_rn(u, cs), __fmul_rn(-v, sn)));
                    ov[2 * p + 1] = (bf16)(__fadd_rn(__fmul_rn(v, cs), __fmul_rn(u, sn)));
                }
            }
            // this step's consumers: flag-in-data words, [utterance][q|k|v][head][32 words of 2 bf16]
            const uint2 bits = __builtin_bit_cast(uint2, ov);
            u64* dst = a.exch + (((size_t)m * 3 + which) * a.H + hh) * 32 + (e >> 1);
            __hip_atomic_store(dst, ((u64)a.epoch << 32) | bits.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, ((u64)a.epoch << 32) | bits.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (which != 0) {                     // later steps read it from the cache
                bf16* cp = (which == 1 ? a.kc : a.vc) + (((size_t)csl * a.H + hh) * a.cache_T + cpos) * 64 + e;
                *reinterpret_cast<bf16x4*>(cp) = ov;
            }
        }
        request_cache(0);
        // ---- hand-off: q, new k, new v of this (utterance, head).  The poll is queued BEHIND the cache stream of this
        // CU on purpose: the words were published microseconds ago, and their data is needed no sooner than the cache
        if (threadIdx.x >= 128 && threadIdx.x < 224) {
            const int t = threadIdx.x - 128;
            const int which = t >> 5, j = t & 31;
            const u64* src = a.exch + (((size_t)b * 3 + which) * a.H + h) * 32 + j;
            u64 wv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((unsigned)(wv >> 32) != a.epoch) {
                __builtin_amdgcn_s_sleep(1);
                wv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > (1u << 22)) { atomicAdd(a.err_flag, 1); break; }   // bounded: the launch always drains
            }
            const bf16x2 pr = __builtin_bit_cast(bf16x2, (unsigned)wv);
            float* dstf = which == 0 ? s_q : (which == 1 ? s_k : s_v);
            dstf[2 * j] = (float)pr[0];
            dstf[2 * j + 1] = (float)pr[1];
        }
    } else {
        QA_LDS_BARRIER();                         // barrier 1: the projection's loads had this CU's memory pipeline to themselves
        request_cache(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    QA_LDS_BARRIER();                             // barrier 2: q, k_new, v_new are in LDS

    // ---- streaming attention over the cached keys (attn_decode.hip), then the new key
    float q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = s_q[sub * 8 + e] * 0.125f;              // 1/sqrt(64), exact
    float mrun = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int p0 = 0; p0 < n_keys; p0 += NW * 8 * MAXC) {
        if (p0 > 0) request_cache(p0);
        float s[MAXC];
        float mloc = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t = fmaf(q[e], (float)kv[c][e], t);
            t += qa_dpp_xor1(t);
            t += qa_dpp_xor2(t);
            t += qa_dpp_half_mirror(t);
            s[c] = (k_lo + c * 8 + g < k_hi) ? t : -INFINITY;
            mloc = fmaxf(mloc, s[c]);
        }
        mloc = fmaxf(mloc, qa_dpp_ror8(mloc));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mn = fmaxf(mrun, mloc);
        if (mn != -INFINITY) {                                                  // wave-uniform
            const float alpha = expf(mrun - mn);
            l *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const float p = expf(s[c] - mn);
                l += p;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(p, (float)vv[c][e], o[e]);
            }
            mrun = mn;
        }
    }
    if (wave == 0) {                              // the new token's own key (position n_keys): one more key for wave 0
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t = fmaf(q[e], s_k[sub * 8 + e], t);
        t += qa_dpp_xor1(t);
        t += qa_dpp_xor2(t);
        t += qa_dpp_half_mirror(t);                // every key group holds the same score: wave-uniform
        const float mn = fmaxf(mrun, t);
        const float alpha = expf(mrun - mn);       // mrun = -inf -> 0
        const float p = (g == 0) ? expf(t - mn) : 0.f;
        l = l * alpha + p;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(p, s_v[sub * 8 + e], o[e] * alpha);
        mrun = mn;
    }
    // ---- merge the waves' partial softmax states (m is wave-uniform: the 8 key groups simply add up)
    l += qa_dpp_ror8(l);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = o[e];
        t += qa_dpp_ror8(t);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        o[e] = t;
    }
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sm_o[wave][sub * 8 + e] = o[e];
        if (sub == 0) { sm_m[wave] = mrun; sm_l[wave] = l; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
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
        const int col = h * 64 + threadIdx.x;
        a.out[qa_frag_off(b, col)] = (bf16)(num / den);
    }
}

int valle_dec_qkv_attn(const QkvAttnArgs& a, hipStream_t s) {
    const int d = a.H * 64;
    VALLE_REQUIRE(a.B >= 1 && a.B <= 16, VALLE_E_UNSUPPORTED, "fused QKV+attention step takes up to 16 utterances, got %d", a.B);
    VALLE_REQUIRE(d == 256 || d == 1024, VALLE_E_UNSUPPORTED, "fused QKV+attention step: d_model %d", d);
    VALLE_REQUIRE(a.n_tiles == 3 * d / 16 && a.n_tiles <= a.B * a.H, VALLE_E_ARG, "fused QKV+attention step: %d tiles for %d workgroups", a.n_tiles, a.B * a.H);
    VALLE_REQUIRE(a.x && a.W && a.ssq_in && a.rope_m && a.ctx_len && a.kc && a.vc && a.exch && a.err_flag && a.out && a.epoch != 0,
                  VALLE_E_ARG, "fused QKV+attention step: null pointer / zero epoch");
    dim3 grid(a.B, a.H);
    if (d == 1024) dec_qkv_attn_kernel<4, 6><<<grid, 1024, 0, s>>>(a);
    else dec_qkv_attn_kernel<1, 6><<<grid, 1024, 0, s>>>(a);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}
