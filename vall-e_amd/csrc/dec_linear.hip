// AR decode linears, wave-streaming form (gfx950, bf16, up to 32 utterances per step).
//
// out[m][n] = f( scale[m] * sum_k W[n][k] x[m][k] )    m < 32 utterances, one new token each.
// nn.Linear call sites replaced (one row per utterance, KV-cache step of models/ar_nar.py:748-848):
// arch/llama.py:307-309 (q/k/v), :475 (o_proj), :492 (gate/up/down), models/base.py:217 (AR head).
//
// HBM-bound: the weights are read exactly once per step for the whole batch.  One WAVE owns a slice of a
// tile's weight rows and streams it straight into registers (the pre-tiled streaming order of valle_hip.h
// makes each wave-instruction a contiguous 1 KiB read whose bytes are already the MFMA A fragment).  The
// <= 16 activation rows are kept in the matching "fragment order" by every producer on the decode path, so
// the B operand is 1 KiB-contiguous too and needs no LDS staging (dec_frag_off in common.h).
//
// Tile shapes.  A launch streams fastest when every CU pulls an equal share of the weights, so a linear with
// few output columns (o_proj, down_proj: N = d_model) uses NARROW tiles: with S = 4 a workgroup owns C = 16/S = 4
// output columns and the 16 rows of its MFMA A operand are (K quarter ks, column c) pairs; the B operand's 16
// columns are (K quarter ks', utterance m') pairs for one group of C utterances, and only the diagonal blocks
// ks == ks' of the 16 x 16 product are kept (S MFMAs per k step, one per utterance group, all on the same A
// registers).  N = 1,024 then gives 256 workgroups instead of 64.  S = 1 is the plain 16-column tile.
// A tile's K' = K/S range is further split over the NWK waves of the workgroup; all partial sums (waves x
// diagonal blocks) are reduced through LDS once.
//
// RMSNorm (arch/llama.py:64-77) is folded in: the norm weight is multiplied into W when the streaming copy
// is built (valle_layer_weights.pn_*), and 1/rms(x[m]) -- computed from the activation registers the wave
// holds anyway (v_dot2 on bf16 pairs), summed over the waves through LDS -- scales the accumulator.
#include "common.h"

// weights are read once per step by one CU: non-temporal loads keep them from displacing the activations in the
// caches (measured on the layer chain, tools/micro_dec: 28.4 -> 28.1 us per layer)
#ifdef VALLE_NT_ST      // dev: outputs by non-temporal stores (measured: see profiles/README.md)
#define STORE_OUT(p, v) __builtin_nontemporal_store((v), (p))
#else
#define STORE_OUT(p, v) (*(p) = (v))
#endif
#ifdef VALLE_NO_NT_W
#define LOAD_W(p) (*(p))
#else
#define LOAD_W(p) __builtin_nontemporal_load(p)
#endif

template <int EPI, int STEPS, int NWK, bool NORM, int MT, int S>
__global__ __launch_bounds__(64 * NWK) void dec_wave_kernel(const bf16* __restrict__ pW, const bf16* __restrict__ pX,
                                                            const bf16* pResid, void* pOut, int B, int N, float eps,
                                                            DecWArgs a) {
    // the first 11 argument dwords arrive in SGPRs (kernarg preload, see the Makefile): the weight and activation
    // loads go out without waiting for a kernarg fetch; everything in `a` is needed behind the streams only
#ifdef VALLE_STAMPS
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = 0, t2 = 0;
#endif
    static_assert(!NORM || S == 1, "the norm prologue is built for plain tiles");
    constexpr int C = 16 / S;                         // output columns per tile
    constexpr int KS = STEPS * NWK;                   // 64-wide k steps per (virtual) weight row
    constexpr int K = KS * 64 * S;                    // the linear's K
    constexpr int P = NWK * S;                        // partial sums per output element
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tile = blockIdx.x;
    const int n = tile * C + fq * 4;                  // this lane's 4 consecutive output columns (lanes with fq*4 < C)
    const bool col_lane = fq * 4 < C;
    [[maybe_unused]] __shared__ __attribute__((aligned(16))) float part[P > 1 ? P : 1][16][S == 1 ? C + 1 : C];
    [[maybe_unused]] __shared__ float part_ss[NWK][16];

    f32x4 bs = {};
    const bf16* wp = pW + ((size_t)tile * KS + wave * STEPS) * 1024 + lane * 8;
    bf16x8 w0[STEPS], w1[STEPS];
    // utterances come in tiles of 16 (MT = 1 for up to 16, 2 for up to 32): the weight registers are loaded once
    // and reused, every tile has its own fragment-order activation matrix (pX + mt * 16 * K)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + fr;                   // this lane's utterance
        const int mc = m < B ? m : B - 1;             // clamped for side loads
        // ---- 1. activations first (L2-resident, back quickly), then the weight stream
        const bf16* xp = pX + (size_t)mt * 16 * K + ((size_t)wave * STEPS * 2 * S * 64 + lane) * 8;
        bf16x8 xv[STEPS * 2][S];
#pragma unroll
        for (int s = 0; s < STEPS * 2; ++s)
#pragma unroll
            for (int g = 0; g < S; ++g) xv[s][g] = *reinterpret_cast<const bf16x8*>(xp + (s * S + g) * 512);
        __builtin_amdgcn_sched_barrier(0);
        if (mt == 0) {
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                w0[s] = LOAD_W(reinterpret_cast<const bf16x8*>(wp + s * 1024));
                w1[s] = LOAD_W(reinterpret_cast<const bf16x8*>(wp + s * 1024 + 512));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 2. epilogue operands: behind the streams (their pointers come from the kernarg segment), back long
        // before the weights
        bf16x4 rs = {};
        f32x4 rp = {};
        int cpos = 0, csl = 0;
        if constexpr (EPI == VALLE_EPI_RESID) {
            if (wave == 0 && col_lane) rs = *reinterpret_cast<const bf16x4*>(pResid + (size_t)mt * 16 * N + dec_frag_off<1>(fr, n, N));
        } else if constexpr (EPI == VALLE_EPI_F32OUT) {
            if (mt == 0 && a.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bs[r] = (n + r < N) ? a.bias[n + r] : 0.f;
            }
        } else if constexpr (EPI == DEC_EPI_QKV) {
            cpos = a.ctx_len[mc];
            csl = a.slot ? a.slot[mc] : mc;
            rp = *reinterpret_cast<const f32x4*>(a.rope_m + (size_t)mc * 64 + (n & 63));   // (cos,sin) of pairs n/2, n/2+1
        }
        // ---- 3. sum of squares of this wave's K share of row m (the activations are back first)
        [[maybe_unused]] float ss = 0.f;
        if constexpr (NORM) {
#pragma unroll
            for (int s = 0; s < STEPS * 2; ++s) {
                const bf16x8 v = xv[s][0];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2 pr = {v[2 * e], v[2 * e + 1]};
                    ss = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ss, false);
                }
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if constexpr (NWK > 1) {
                if (mt > 0) __syncthreads();
                if (fq == 0) part_ss[wave][fr] = ss;
            }
        }
#ifdef VALLE_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- 4. the product
        f32x4 acc[S];
#pragma unroll
        for (int g = 0; g < S; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
#pragma unroll
            for (int g = 0; g < S; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[s], xv[2 * s][g], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < S; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s], xv[2 * s + 1][g], acc[g], 0, 0, 0);
        }
        f32x4 sum = acc[0];
        if constexpr (P > 1) {
            // acc[g][r] is (row fq*4 + r, column fr) of the product for utterance group g: row = (K part, column c),
            // column = (K part, utterance); the diagonal blocks go to LDS as part[wave*S + K part][utterance][c .. c+3]
            if (mt > 0 && !(NORM && NWK > 1)) __syncthreads();   // the previous tile's partial sums have been consumed
            const int ks_i = (fq * 4) / C, ks_j = fr / C;
            if (ks_i == ks_j) {
#pragma unroll
                for (int g = 0; g < S; ++g) {
                    float* q = &part[wave * S + ks_i][g * C + (fr % C)][(fq * 4) % C];
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r] = acc[g][r];
                }
            }
            __syncthreads();
            if constexpr (S == 4) {
                // 4 columns x P partial sums per utterance: the four lane groups share the partial sums, then fold
                if (wave == 0) {
                    sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int p = 0; p < P; p += 4) sum += *reinterpret_cast<const f32x4*>(&part[p + fq][fr][0]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sum[r] += __shfl_xor(sum[r], 16, 64);
                        sum[r] += __shfl_xor(sum[r], 32, 64);
                    }
                }
            } else if (wave == 0 && col_lane) {
                sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[r] += part[p][fr][fq * 4 + r];
            }
        }
#ifdef VALLE_STAMPS
        t2 = __builtin_amdgcn_s_memrealtime();
#endif
        if (wave != 0 || !col_lane || m >= B) continue;   // (with NWK == 1 every wave is wave 0 of its own workgroup)
        if constexpr (NORM) {
            if constexpr (NWK > 1) {
                ss = 0.f;
#pragma unroll
                for (int w = 0; w < NWK; ++w) ss += part_ss[w][fr];
            }
            const float scale = 1.0f / sqrtf(ss * (1.0f / (float)K) + eps);
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[r] *= scale;
        }
        // ---- 5. epilogue: lane = utterance m, columns n .. n+3
        if constexpr (EPI == VALLE_EPI_RESID) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)((float)rs[r] + (float)(bf16)sum[r]);
            STORE_OUT(reinterpret_cast<bf16x4*>((bf16*)pOut + (size_t)mt * 16 * N + dec_frag_off<1>(fr, n, N)), o);
        } else if constexpr (EPI == VALLE_EPI_GEGLU) {
            bf16x2 o;
            o[0] = (bf16)((float)(bf16)gelu_erf_bf16out((float)(bf16)sum[0]) * (float)(bf16)sum[1]);
            o[1] = (bf16)((float)(bf16)gelu_erf_bf16out((float)(bf16)sum[2]) * (float)(bf16)sum[3]);
            STORE_OUT(reinterpret_cast<bf16x2*>((bf16*)pOut + (size_t)mt * 16 * (N >> 1) + dec_frag_off<DEC_S_DOWN>(fr, n >> 1, N >> 1)), o);
        } else if constexpr (EPI == VALLE_EPI_F32OUT) {
            float* o = (float*)pOut + (size_t)m * a.ldo + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < N) o[r] = sum[r] + bs[r];
        } else if constexpr (EPI == DEC_EPI_QKV) {
            const int d = a.d;
            const int which = n / d, c = n - which * d, h = c >> 6, e = c & 63;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)sum[r];
            if (which < 2) {                          // RoPE on the two adjacent pairs (arch/llama.py:48-61)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float cs = (float)(bf16)rp[2 * p], sn = (float)(bf16)rp[2 * p + 1];
                    const float u = (float)o[2 * p], v = (float)o[2 * p + 1];
                    o[2 * p] = (bf16)(__fadd_rn(__fmul_rn(u, cs), __fmul_rn(-v, sn)));
                    o[2 * p + 1] = (bf16)(__fadd_rn(__fmul_rn(v, cs), __fmul_rn(u, sn)));
                }
            }
            if (which == 0) {
                STORE_OUT(reinterpret_cast<bf16x4*>((bf16*)pOut + (size_t)m * a.ldo + n), o);
            } else if (cpos >= 0) {                    // a negative ctx_len marks a finished utterance (early exit): no append
                bf16* cp = (which == 1 ? a.kc : a.vc) + (((size_t)csl * a.H + h) * a.cache_T + cpos) * 64 + e;
                STORE_OUT(reinterpret_cast<bf16x4*>(cp), o);
            }
        }
    }
#ifdef VALLE_STAMPS
    if (a.stamps && wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { unsigned long long* q = a.stamps + (size_t)blockIdx.x * 4; q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; }
    }
#endif
}

template <int EPI, int STEPS, int NWK, bool NORM, int S>
static int dec_wave_launch(const DecWArgs& a_in, hipStream_t s) {
#ifdef VALLE_STAMPS
    DecWArgs a = a_in; a.stamps = valle_stamp_next();
#else
    const DecWArgs& a = a_in;
#endif
    const int grid = ceil_div(a.N, 16 / S);
    if (a.B <= 16) dec_wave_kernel<EPI, STEPS, NWK, NORM, 1, S><<<grid, 64 * NWK, 0, s>>>(a.W, a.x, a.resid, a.out, a.B, a.N, a.eps, a);
    else dec_wave_kernel<EPI, STEPS, NWK, NORM, 2, S><<<grid, 64 * NWK, 0, s>>>(a.W, a.x, a.resid, a.out, a.B, a.N, a.eps, a);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// Shapes per linear (K' = K / S is split into NWK wave shares of STEPS 64-wide steps).  Measured on the decode-layer chain
// (tools/micro_dec, B = 16, d_model 1,024; see profiles/README.md): only the full width is tuned, the other presets of the
// reference (config.py:462-469: 256 "quarter", 512 "half", 768) take the same forms scaled down.
#ifndef DEC_O_STEPS       // dev overrides (tools/micro_dec builds): -DDEC_O_STEPS=.. -DDEC_O_NWK=.. etc.
#define DEC_O_STEPS 2
#define DEC_O_NWK 2
#endif
#ifndef DEC_DN_STEPS
#define DEC_DN_STEPS 2
#define DEC_DN_NWK 8
#endif
int valle_dec_wave_linear(const DecWArgs& a, int epilogue, hipStream_t s) {
    VALLE_REQUIRE(a.B >= 1 && a.B <= 32, VALLE_E_UNSUPPORTED, "wave-streaming decode linears take 1..32 utterances, got %d", a.B);
    VALLE_REQUIRE(a.x && a.W && a.out, VALLE_E_ARG, "decode linear: null pointer");
    const int K = a.K;
    const bool two = a.B <= 16;       // K = d_model on two waves up to 16 utterances, on one above (B = 32: 40.1 vs 40.8 us per layer)
    if (a.norm) {
#define DEC_NORM_CASES(EPI) \
        if (K == 256) return dec_wave_launch<EPI, 4, 1, true, 1>(a, s); \
        if (K == 512) return dec_wave_launch<EPI, 8, 1, true, 1>(a, s); \
        if (K == 768) return dec_wave_launch<EPI, 12, 1, true, 1>(a, s); \
        if (K == 1024) return two ? dec_wave_launch<EPI, 8, 2, true, 1>(a, s) : dec_wave_launch<EPI, 16, 1, true, 1>(a, s);
        switch (epilogue) {
            case VALLE_EPI_GEGLU: DEC_NORM_CASES(VALLE_EPI_GEGLU) break;
            case VALLE_EPI_F32OUT: DEC_NORM_CASES(VALLE_EPI_F32OUT) break;
            case DEC_EPI_QKV: DEC_NORM_CASES(DEC_EPI_QKV) break;
        }
    } else if (epilogue == VALLE_EPI_RESID) {
        VALLE_REQUIRE(a.resid != nullptr, VALLE_E_ARG, "decode linear: residual missing");
        VALLE_REQUIRE(a.N % 4 == 0, VALLE_E_ARG, "decode linear: N must be a multiple of 4");
        constexpr int S = DEC_S_DOWN;   // == DEC_S_O: narrow tiles, K' = K / 4
        static_assert(DEC_S_DOWN == 4 && DEC_S_O == 4, "the shape table below is written for S = 4");
        if (K == 256) return dec_wave_launch<VALLE_EPI_RESID, 1, 1, false, S>(a, s);       // K' = 64
        if (K == 512) return dec_wave_launch<VALLE_EPI_RESID, 1, 2, false, S>(a, s);       // 128
        if (K == 768) return dec_wave_launch<VALLE_EPI_RESID, 1, 3, false, S>(a, s);       // 192
        if (K == 1024) return dec_wave_launch<VALLE_EPI_RESID, DEC_O_STEPS, DEC_O_NWK, false, S>(a, s);   // 256
        if (K == 2048) return dec_wave_launch<VALLE_EPI_RESID, 2, 4, false, S>(a, s);      // 512
        if (K == 3072) return dec_wave_launch<VALLE_EPI_RESID, 2, 6, false, S>(a, s);      // 768
        if (K == 4096) return dec_wave_launch<VALLE_EPI_RESID, DEC_DN_STEPS, DEC_DN_NWK, false, S>(a, s); // 1,024
    }
    valle_set_error("decode linear: combination not built (K=%d norm=%d epilogue=%d)", K, a.norm, epilogue);
    return VALLE_E_UNSUPPORTED;
}
