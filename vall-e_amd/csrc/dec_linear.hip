// AR decode linears, wave-streaming form (gfx950, bf16, up to 32 utterances per step).
//
// out[m][n] = f( scale[m] * sum_k W[n][k] x[m][k] )    m < 32 utterances, one new token each.
// nn.Linear call sites replaced (one row per utterance, KV-cache step of models/ar_nar.py:748-848):
// arch/llama.py:307-309 (q/k/v), :475 (o_proj), :492 (gate/up/down), models/base.py:217 (AR head).
//
// HBM-bound: the weights are read exactly once per step for the whole batch.  One WAVE owns a tile of
// 16 weight rows x KW columns and streams it straight into registers (the pre-tiled streaming order of
// valle_hip.h makes each wave-instruction a contiguous 1 KiB read whose bytes are already the MFMA A
// fragment).  The <= 16 activation rows are kept in the same "fragment order" by every producer on the
// decode path, so the B operand is 1 KiB-contiguous too and needs no LDS staging, no barrier:
//     xfrag[(k/32)*64 + ((k/8)%4)*16 + m][k%8]  = x[m][k]     (one such [16][K] matrix per 16 utterances)
// A tile's K range is split over the NWK waves of a workgroup (each streams its share into registers) and the
// partial sums are reduced through LDS once: 2 waves at K = d_model, 8 at K = d_ff (shorter load queues and MFMA
// chains per wave beat the single-wave form by ~6 % of the layer).
//
// RMSNorm (arch/llama.py:64-77) is folded in: the norm weight is multiplied into W when the streaming
// copy is built (valle_layer_weights.pn_*), and 1/rms(x[m]) scales the accumulator in the epilogue.
// sum_k x[m][k]^2 arrives as per-tile partial sums written by whichever kernel produced x.
#include "common.h"

__device__ __forceinline__ int frag_off(int m, int k) { return ((((k >> 5) << 6) + (((k >> 3) & 3) << 4) + m) << 3) + (k & 7); }

template <int EPI, int STEPS, int NWK, bool NORM, int MT>
__global__ __launch_bounds__(64 * NWK) void dec_wave_kernel(DecWArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tile = blockIdx.x;
    const int n = tile * 16 + fq * 4;                 // this lane's 4 consecutive output columns
    constexpr int KS = STEPS * NWK;                   // 64-wide k steps per weight row
    constexpr int K = KS * 64;
    [[maybe_unused]] __shared__ float part[NWK > 1 ? NWK : 1][16][17];

    f32x4 bs = {};
    if constexpr (EPI == VALLE_EPI_F32OUT) {
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bs[r] = (n + r < a.N) ? a.bias[n + r] : 0.f;
        }
    }
    const bf16* wp = a.W + ((size_t)tile * KS + wave * STEPS) * 1024 + lane * 8;
    bf16x8 w0[STEPS], w1[STEPS];
    // utterances come in tiles of 16 (MT = 1 for up to 16, 2 for up to 32): the weight registers are loaded once
    // and reused, every tile has its own fragment-order activation matrix (a.x + mt * 16 * K)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + fr;                   // this lane's utterance
        const int mc = m < a.B ? m : a.B - 1;         // clamped for side loads
        // ---- 1. epilogue operands: issued first so they are back long before the weight stream ends
        bf16x4 rs = {};
        f32x4 rp = {};
        int cpos = 0, csl = 0;
        if constexpr (EPI == VALLE_EPI_RESID) {
            if (wave == 0) rs = *reinterpret_cast<const bf16x4*>(a.resid + (size_t)mt * 16 * a.N + frag_off(fr, n));
        } else if constexpr (EPI == DEC_EPI_QKV) {
            cpos = a.ctx_len[mc];
            csl = a.slot ? a.slot[mc] : mc;
            rp = *reinterpret_cast<const f32x4*>(a.rope_m + (size_t)mc * 64 + (n & 63));   // (cos,sin) of pairs n/2, n/2+1
        }
        // ---- 2. per-row sum of squares: partial sums per 16-column tile, left by the producer of x
        // (ssq[m][K/16]); this lane adds tiles fq*STEPS .. +STEPS of its row
        f32x4 sq[NORM ? KS / 4 : 1];
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < KS / 4; ++i)
                sq[i] = *reinterpret_cast<const f32x4*>(a.ssq_in + (size_t)m * (KS * 4) + fq * KS + i * 4);
        }
        // ---- 3. activations first (L2-resident, back quickly), then the weight stream: a wave can have 64
        // vector-memory instructions outstanding, exactly the 32 + 32 of a K = 1024 tile
        const bf16* xp = a.x + (size_t)mt * 16 * K + ((size_t)wave * STEPS * 128 + lane) * 8;
        bf16x8 x0[STEPS], x1[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            x0[s] = *reinterpret_cast<const bf16x8*>(xp + s * 1024);
            x1[s] = *reinterpret_cast<const bf16x8*>(xp + s * 1024 + 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mt == 0) {
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                w0[s] = *reinterpret_cast<const bf16x8*>(wp + s * 1024);
                w1[s] = *reinterpret_cast<const bf16x8*>(wp + s * 1024 + 512);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float scale = 1.f;
        if constexpr (NORM) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < KS / 4; ++i) ss += (sq[i][0] + sq[i][1]) + (sq[i][2] + sq[i][3]);
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            scale = 1.0f / sqrtf(ss * (1.0f / (float)K) + a.eps);
        }
        // ---- 4. the product
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[s], x0[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s], x1[s], acc, 0, 0, 0);
        }
        if constexpr (NWK > 1) {
            if (mt > 0) __syncthreads();              // the previous tile's partial sums have been consumed
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][fq * 4 + r][fr] = acc[r];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 1; w < NWK; ++w)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += part[w][fq * 4 + r][fr];
            }
        }
        if (wave != 0 || m >= a.B) continue;          // (with NWK == 1 every wave is wave 0 of its own workgroup)
        if constexpr (NORM) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= scale;
        }
        // ---- 5. epilogue: lane = utterance m, columns n .. n+3
        if constexpr (EPI == VALLE_EPI_RESID) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)((float)rs[r] + (float)(bf16)acc[r]);
            *reinterpret_cast<bf16x4*>((bf16*)a.out + (size_t)mt * 16 * a.N + frag_off(fr, n)) = o;
            if (a.ssq_out) {                          // this tile's share of sum_n x[m][n]^2 for the next RMSNorm
                float q2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) q2 = fmaf((float)o[r], (float)o[r], q2);
                q2 += __shfl_xor(q2, 16, 64);
                q2 += __shfl_xor(q2, 32, 64);
                if (fq == 0) a.ssq_out[(size_t)m * (a.N >> 4) + tile] = q2;
            }
        } else if constexpr (EPI == VALLE_EPI_GEGLU) {
            bf16x2 o;
            o[0] = (bf16)((float)(bf16)gelu_erf((float)(bf16)acc[0]) * (float)(bf16)acc[1]);
            o[1] = (bf16)((float)(bf16)gelu_erf((float)(bf16)acc[2]) * (float)(bf16)acc[3]);
            *reinterpret_cast<bf16x2*>((bf16*)a.out + (size_t)mt * 16 * (a.N >> 1) + frag_off(fr, n >> 1)) = o;
        } else if constexpr (EPI == VALLE_EPI_F32OUT) {
            float* o = (float*)a.out + (size_t)m * a.ldo + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < a.N) o[r] = acc[r] + bs[r];
        } else if constexpr (EPI == DEC_EPI_QKV) {
            const int d = a.d;
            const int which = n / d, c = n - which * d, h = c >> 6, e = c & 63;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[r];
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
                *reinterpret_cast<bf16x4*>((bf16*)a.out + (size_t)m * a.ldo + n) = o;
            } else {
                bf16* cp = (which == 1 ? a.kc : a.vc) + (((size_t)csl * a.H + h) * a.cache_T + cpos) * 64 + e;
                *reinterpret_cast<bf16x4*>(cp) = o;
            }
        }
    }
}

template <int EPI, int STEPS, int NWK, bool NORM>
static int dec_wave_launch(const DecWArgs& a, hipStream_t s) {
    if (a.B <= 16) dec_wave_kernel<EPI, STEPS, NWK, NORM, 1><<<ceil_div(a.N, 16), 64 * NWK, 0, s>>>(a);
    else dec_wave_kernel<EPI, STEPS, NWK, NORM, 2><<<ceil_div(a.N, 16), 64 * NWK, 0, s>>>(a);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// K-split of one 16-row tile over the waves of a workgroup.  Measured on the decode-layer chain (tools/micro_dec,
// tools/run_dec_variants.sh, B = 16): 2 waves for K = 1024 and 8 waves for K = 4096 give 30.0 us per layer against
// 31.8 us for (1, 4); more waves lose again to the LDS reduction; above 16 utterances K = 1024 stays on one wave
// (B = 32: 40.1 vs 40.8 us per layer).  VALLE_DEC_NWK (1 | 2) / VALLE_DEC_NWK_DN (4 | 8) override (dev).
static int dec_variant(int B) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VALLE_DEC_NWK"); v = e ? atoi(e) : 0; }
    return v ? v : (B <= 16 ? 2 : 1);      // two 16-utterance tiles per wave (B > 16): one wave per tile is 40.1 vs 40.8 us
}
static int dec_variant_dn() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VALLE_DEC_NWK_DN"); v = e ? atoi(e) : 8; }
    return v;
}
#define DEC_K1024(EPI, NORM) \
    switch (dec_variant(a.B)) { \
        case 1: return dec_wave_launch<EPI, 16, 1, NORM>(a, s); \
        default: return dec_wave_launch<EPI, 8, 2, NORM>(a, s); \
    }

// K = d_model in {256, 512, 768, 1024} (the reference's "quarter", "half", 0.75 and "full" sizes, config.py:462-469) for
// the norm-prologue linears (QKV, gate/up, head) and o_proj; down_proj K = d_ff = 4 d_model.  Only the full size is tuned.
#define DEC_KSMALL(EPI, NORM) \
    if (K == 256) return dec_wave_launch<EPI, 4, 1, NORM>(a, s); \
    if (K == 512) return dec_wave_launch<EPI, 8, 1, NORM>(a, s); \
    if (K == 768) return dec_wave_launch<EPI, 12, 1, NORM>(a, s);
int valle_dec_wave_linear(const DecWArgs& a, int epilogue, hipStream_t s) {
    VALLE_REQUIRE(a.B >= 1 && a.B <= 32, VALLE_E_UNSUPPORTED, "wave-streaming decode linears take 1..32 utterances, got %d", a.B);
    VALLE_REQUIRE(a.x && a.W && a.out, VALLE_E_ARG, "decode linear: null pointer");
    const int K = a.K;
    if (a.norm) {
        switch (epilogue) {
            case VALLE_EPI_GEGLU: DEC_KSMALL(VALLE_EPI_GEGLU, true) if (K == 1024) DEC_K1024(VALLE_EPI_GEGLU, true) break;
            case VALLE_EPI_F32OUT: DEC_KSMALL(VALLE_EPI_F32OUT, true) if (K == 1024) DEC_K1024(VALLE_EPI_F32OUT, true) break;
            case DEC_EPI_QKV: DEC_KSMALL(DEC_EPI_QKV, true) if (K == 1024) DEC_K1024(DEC_EPI_QKV, true) break;
        }
    } else if (epilogue == VALLE_EPI_RESID) {
        VALLE_REQUIRE(a.resid != nullptr, VALLE_E_ARG, "decode linear: residual missing");
        DEC_KSMALL(VALLE_EPI_RESID, false)
        if (K == 1024) DEC_K1024(VALLE_EPI_RESID, false)
        if (K == 2048) return dec_wave_launch<VALLE_EPI_RESID, 8, 4, false>(a, s);
        if (K == 3072) return dec_wave_launch<VALLE_EPI_RESID, 12, 4, false>(a, s);
        if (K == 4096) switch (dec_variant_dn()) {
            case 4: return dec_wave_launch<VALLE_EPI_RESID, 16, 4, false>(a, s);
            default: return dec_wave_launch<VALLE_EPI_RESID, 8, 8, false>(a, s);
        }
    }
    valle_set_error("decode linear: combination not built (K=%d norm=%d epilogue=%d)", K, a.norm, epilogue);
    return VALLE_E_UNSUPPORTED;
}
