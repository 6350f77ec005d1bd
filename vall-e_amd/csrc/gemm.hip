// MFMA GEMMs of the AR+NAR forward (gfx950).
//   valle_gemm        dense passes (prefill, NAR levels, NAR-len steps): C = A . W^T, epilogues
//   valle_dec_linear  AR decode step, B <= 32 rows: weight-streaming, RMSNorm prologue fused
// nn.Linear call sites replaced: models/arch/llama.py:307-309 (q/k/v), :475 (o_proj),
// :492 (gate/up/down), models/base.py:217 (classifier heads).
//
// Orientation: the MFMA's A operand is the WEIGHT tile (rows n), the B operand the activation
// tile (rows m):  D[n][m] = sum_k W[n][k] X[m][k].  In the 16x16 C/D layout a lane then holds
// 4 consecutive n of one m, so the adjacent pairs this path needs -- a RoPE pair (w_qkv is
// pair-interleaved) and a (gate, up) pair (w_gu is interleaved) -- sit in one lane's registers.
#include "common.h"

#define TN 128
#define TM 128
#define LDS_ROW_BYTES 80   // 64 B of k-data + 16 B pad (keeps 16-B alignment, spreads banks)

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ X, const T* __restrict__ W,
                                                      const void* __restrict__ bias, const T* resid,
                                                      void* Cout, int M, int N, int K, int ldc) {
    constexpr bool IS_BF16 = sizeof(T) == 2;
    constexpr int BK = IS_BF16 ? 32 : 16;            // 64 bytes of k per row either way
    __shared__ __attribute__((aligned(16))) char lds[2][2][TN * LDS_ROW_BYTES];  // [buf][W|X]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: 512 16-byte chunks per operand tile, 2 per thread
    uint4 rw[2], rx[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * (16 / (int)sizeof(T));
            const int n = n0 + row, m = m0 + row;
            rw[i] = (n < N) ? *reinterpret_cast<const uint4*>(W + (size_t)n * K + kt * BK + kc) : make_uint4(0, 0, 0, 0);
            rx[i] = (m < M) ? *reinterpret_cast<const uint4*>(X + (size_t)m * K + kt * BK + kc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, ch = c & 3;
            *reinterpret_cast<uint4*>(&lds[buf][0][row * LDS_ROW_BYTES + ch * 16]) = rw[i];
            *reinterpret_cast<uint4*>(&lds[buf][1][row * LDS_ROW_BYTES + ch * 16]) = rx[i];
        }
    };
    const int nt = K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nt) load_tile(kt + 1);
        const char* Ws = lds[buf][0];
        const char* Xs = lds[buf][1];
        if constexpr (IS_BF16) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(Ws + (wn * 64 + i * 16 + fr) * LDS_ROW_BYTES + fq * 16);
                b[i] = *reinterpret_cast<const bf16x8*>(Xs + (wm * 64 + i * 16 + fr) * LDS_ROW_BYTES + fq * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = *reinterpret_cast<const float*>(Ws + (wn * 64 + i * 16 + fr) * LDS_ROW_BYTES + (ks * 4 + fq) * 4);
                    b[i] = *reinterpret_cast<const float*>(Xs + (wm * 64 + i * 16 + fr) * LDS_ROW_BYTES + (ks * 4 + fq) * 4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds n = nb + 0..3 (consecutive) for one m
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nb = n0 + wn * 64 + i * 16 + fq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + fr;
            if (m >= M || nb >= N) continue;
            f32x4 v = acc[i][j];
            if constexpr (EPI == VALLE_EPI_F32OUT) {
                float* C = (float*)Cout;
                const float* bf = (const float*)bias;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < N) C[(size_t)m * ldc + nb + r] = v[r] + (bf ? bf[nb + r] : 0.f);
            } else if constexpr (EPI == VALLE_EPI_GEGLU) {
                T* C = (T*)Cout;
                typedef T vec2 __attribute__((ext_vector_type(2)));
                vec2 o;
                // reference: act(gate) and up are each rounded to the storage type before the product
                o[0] = from_f32<T>(to_f32<T>(from_f32<T>(gelu_erf(to_f32<T>(from_f32<T>(v[0]))))) * to_f32<T>(from_f32<T>(v[1])));
                o[1] = from_f32<T>(to_f32<T>(from_f32<T>(gelu_erf(to_f32<T>(from_f32<T>(v[2]))))) * to_f32<T>(from_f32<T>(v[3])));
                *reinterpret_cast<vec2*>(C + (size_t)m * ldc + (nb >> 1)) = o;
            } else {
                T* C = (T*)Cout;
                const T* bt = (const T*)bias;
                if (nb + 3 < N && (ldc & 3) == 0) {
                    typedef T vec4 __attribute__((ext_vector_type(4)));
                    vec4 o;
                    if constexpr (EPI == VALLE_EPI_RESID) {
                        const vec4 rr = *reinterpret_cast<const vec4*>(resid + (size_t)m * ldc + nb);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(to_f32<T>(rr[r]) + to_f32<T>(from_f32<T>(v[r])));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r] + (bt ? to_f32<T>(bt[nb + r]) : 0.f));
                    }
                    *reinterpret_cast<vec4*>(C + (size_t)m * ldc + nb) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (nb + r >= N) continue;
                        float o = v[r];
                        if constexpr (EPI == VALLE_EPI_RESID) o = to_f32<T>(resid[(size_t)m * ldc + nb + r]) + to_f32<T>(from_f32<T>(o));
                        else if (bt) o += to_f32<T>(bt[nb + r]);
                        C[(size_t)m * ldc + nb + r] = from_f32<T>(o);
                    }
                }
            }
        }
    }
}

template <typename T>
static int gemm_dispatch(const void* A, const void* W, const void* bias, const void* resid, void* C,
                         int M, int N, int K, int ldc, int epi, hipStream_t s) {
    dim3 grid(ceil_div(N, TN), ceil_div(M, TM));
    switch (epi) {
        case VALLE_EPI_STORE:  gemm_nt_kernel<T, VALLE_EPI_STORE><<<grid, 256, 0, s>>>((const T*)A, (const T*)W, bias, (const T*)resid, C, M, N, K, ldc); break;
        case VALLE_EPI_RESID:  gemm_nt_kernel<T, VALLE_EPI_RESID><<<grid, 256, 0, s>>>((const T*)A, (const T*)W, bias, (const T*)resid, C, M, N, K, ldc); break;
        case VALLE_EPI_GEGLU:  gemm_nt_kernel<T, VALLE_EPI_GEGLU><<<grid, 256, 0, s>>>((const T*)A, (const T*)W, bias, (const T*)resid, C, M, N, K, ldc); break;
        case VALLE_EPI_F32OUT: gemm_nt_kernel<T, VALLE_EPI_F32OUT><<<grid, 256, 0, s>>>((const T*)A, (const T*)W, bias, (const T*)resid, C, M, N, K, ldc); break;
        default: valle_set_error("valle_gemm: unknown epilogue %d", epi); return VALLE_E_ARG;
    }
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_gemm(const void* A, const void* W, const void* bias, const void* resid, void* C,
                          int M, int N, int K, int ldc, int epilogue, int dtype, void* stream) {
    VALLE_REQUIRE(A && W && C, VALLE_E_ARG, "valle_gemm: null pointer");
    VALLE_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 32 == 0, VALLE_E_ARG, "valle_gemm: M=%d N=%d K=%d (K must be a multiple of 32)", M, N, K);
    VALLE_REQUIRE(epilogue != VALLE_EPI_RESID || resid, VALLE_E_ARG, "valle_gemm: EPI_RESID needs resid");
    VALLE_REQUIRE(epilogue != VALLE_EPI_GEGLU || (N % 4 == 0 && ldc % 2 == 0), VALLE_E_ARG, "valle_gemm: EPI_GEGLU needs N %% 4 == 0");
    VALLE_REQUIRE(ldc >= (epilogue == VALLE_EPI_GEGLU ? N / 2 : N), VALLE_E_ARG, "valle_gemm: ldc %d too small", ldc);
    if (M == 0) return VALLE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VALLE_BF16) return gemm_dispatch<bf16>(A, W, bias, resid, C, M, N, K, ldc, epilogue, s);
    if (dtype == VALLE_F32) return gemm_dispatch<float>(A, W, bias, resid, C, M, N, K, ldc, epilogue, s);
    valle_set_error("valle_gemm: unknown dtype %d", dtype);
    return VALLE_E_ARG;
}

// ===================================================================== decode linears
// out[B<=32][N] = f(rmsnorm?(x)[B][K] . W[N][K]^T): one workgroup per 16 weight rows, the 4
// waves split K; every lane's weight loads are issued before anything else (the stream is
// read exactly once, straight to VGPRs), the RMSNorm of the <=32 activation rows is recomputed
// per workgroup from L2 while they fly.  k-slot permutation: a lane reads 32 contiguous bytes
// of its weight row per MFMA pair (elements kb+16g+{0..7} feed MFMA 0, +{8..15} MFMA 1) and
// the activation fragment uses the same permutation, so the sum over k is unchanged.

template <int EPI, bool NORM, int MT, int STEPS>
__global__ __launch_bounds__(256) void dec_linear_kernel(DecArgs a) {
    constexpr int NW = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int K = a.K;
    const int kw = K / NW;                 // k range of this wave
    const int kb0 = wave * kw;
    __shared__ float s_rinv[32];
    __shared__ float s_part[NW][MT][16][17];

    // ---- 1. issue the weight stream (16 rows x kw), 32 B per lane per 64-k step
    const int n_row = n0 + fr;
    const bf16* wrow = a.W + (size_t)(n_row < a.N ? n_row : a.N - 1) * K + kb0 + fq * 16;
    bf16x8 w0[STEPS], w1[STEPS];           // STEPS = K / 4 waves / 64: the whole slice in registers
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        w0[s] = *reinterpret_cast<const bf16x8*>(wrow + s * 64);
        w1[s] = *reinterpret_cast<const bf16x8*>(wrow + s * 64 + 8);
    }
    // ---- 2. RMSNorm statistics of the B rows (wave w takes rows w, w+4, ...)
    if constexpr (NORM) {
        for (int m = wave; m < a.B; m += NW) {
            float ss = 0.f;
            for (int c = lane * 8; c < K; c += 512) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.x + (size_t)m * K + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; ss += f * f; }
            }
            ss = wave_sum(ss);
            if (lane == 0) s_rinv[m] = 1.0f / sqrtf(ss / (float)K + a.eps);
        }
        __syncthreads();
    }
    // ---- 3. MFMAs
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = t * 16 + fr;
                bf16x8 x0, x1;
                if (m < a.B) {
                    const bf16* xp = a.x + (size_t)m * K + kb0 + s * 64 + fq * 16;
                    x0 = *reinterpret_cast<const bf16x8*>(xp);
                    x1 = *reinterpret_cast<const bf16x8*>(xp + 8);
                    if constexpr (NORM) {
                        const float ri = s_rinv[m];
                        const bf16* gp = a.norm_w + kb0 + s * 64 + fq * 16;
                        const bf16x8 g0 = *reinterpret_cast<const bf16x8*>(gp);
                        const bf16x8 g1 = *reinterpret_cast<const bf16x8*>(gp + 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            x0[e] = (bf16)((float)g0[e] * (float)(bf16)((float)x0[e] * ri));
                            x1[e] = (bf16)((float)g1[e] * (float)(bf16)((float)x1[e] * ri));
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { x0[e] = (bf16)0.f; x1[e] = (bf16)0.f; }
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[s], x0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s], x1, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- 4. cross-wave reduction through LDS
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_part[wave][t][fq * 4 + r][fr] = acc[t][r];
    __syncthreads();
    // thread -> (m, pair of n): 16*MT rows x 8 pairs
    for (int idx = tid; idx < 16 * MT * 8; idx += 256) {
        const int m = idx & (16 * MT - 1), np = idx / (16 * MT);
        if (m >= a.B) continue;
        const int t = m >> 4, mm = m & 15;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { v0 += s_part[w][t][2 * np][mm]; v1 += s_part[w][t][2 * np + 1][mm]; }
        const int n = n0 + 2 * np;
        if (n >= a.N) continue;
        if constexpr (EPI == VALLE_EPI_F32OUT) {
            float* o = (float*)a.out + (size_t)m * a.ldo;
            o[n] = v0 + (a.bias ? a.bias[n] : 0.f);
            if (n + 1 < a.N) o[n + 1] = v1 + (a.bias ? a.bias[n + 1] : 0.f);
        } else if constexpr (EPI == VALLE_EPI_RESID) {
            bf16* o = (bf16*)a.out + (size_t)m * a.ldo;
            const bf16* rs = a.resid + (size_t)m * a.ldo;
            bf16x2 ov;
            ov[0] = (bf16)((float)rs[n] + (float)(bf16)v0);
            ov[1] = (bf16)((float)rs[n + 1] + (float)(bf16)v1);
            *reinterpret_cast<bf16x2*>(o + n) = ov;
        } else if constexpr (EPI == VALLE_EPI_GEGLU) {
            bf16* o = (bf16*)a.out + (size_t)m * a.ldo;
            o[n >> 1] = (bf16)((float)(bf16)gelu_erf((float)(bf16)v0) * (float)(bf16)v1);
        } else if constexpr (EPI == DEC_EPI_QKV) {
            const int d = a.d;
            const int which = n / d, c = n - which * d, h = c >> 6, e = c & 63;
            bf16 o0 = (bf16)v0, o1 = (bf16)v1;
            if (which < 2) {
                const float* tab = a.rope + (size_t)a.pos[m] * 64 + (e >> 1) * 2;
                const float cs = (float)(bf16)tab[0], sn = (float)(bf16)tab[1];
                const float x0 = (float)o0, x1 = (float)o1;
                o0 = (bf16)(__fadd_rn(__fmul_rn(x0, cs), __fmul_rn(-x1, sn)));
                o1 = (bf16)(__fadd_rn(__fmul_rn(x1, cs), __fmul_rn(x0, sn)));
            }
            bf16x2 ov; ov[0] = o0; ov[1] = o1;
            if (which == 0) {
                *reinterpret_cast<bf16x2*>((bf16*)a.out + (size_t)m * a.ldo + n) = ov;
            } else {
                const int sl = a.slot ? a.slot[m] : m;
                bf16* cp = (which == 1 ? a.kc : a.vc) + (((size_t)sl * a.H + h) * a.cache_T + a.ctx_len[m]) * 64 + e;
                *reinterpret_cast<bf16x2*>(cp) = ov;
            }
        } else {
            bf16* o = (bf16*)a.out + (size_t)m * a.ldo;
            bf16x2 ov; ov[0] = (bf16)v0; ov[1] = (bf16)v1;
            *reinterpret_cast<bf16x2*>(o + n) = ov;
        }
    }
}

template <int EPI, bool NORM, int STEPS>
static int dec_launch2(const DecArgs& a, hipStream_t s) {
    const int grid = ceil_div(a.N, 16);
    if (a.B <= 16) dec_linear_kernel<EPI, NORM, 1, STEPS><<<grid, 256, 0, s>>>(a);
    else dec_linear_kernel<EPI, NORM, 2, STEPS><<<grid, 256, 0, s>>>(a);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// bf16 only.  RMSNorm-prologue linears (QKV, gate/up, heads) take K = d_model in {256, 1024};
// residual linears (o_proj, down_proj) K in {256, 1024, 4096}.
int valle_dec_linear_bf16(const DecArgs& a, int epilogue, hipStream_t s) {
    VALLE_REQUIRE(a.B >= 1 && a.B <= 32, VALLE_E_UNSUPPORTED, "decode linears are built for 1..32 utterances per step, got %d", a.B);
    VALLE_REQUIRE(a.N % 2 == 0 || epilogue == VALLE_EPI_F32OUT, VALLE_E_ARG, "decode linear: odd N needs the fp32 epilogue");
    const bool norm = a.norm_w != nullptr;
    const int K = a.K;
    if (norm && (K == 256 || K == 1024)) {
        switch (epilogue) {
            case VALLE_EPI_GEGLU:  return K == 256 ? dec_launch2<VALLE_EPI_GEGLU, true, 1>(a, s) : dec_launch2<VALLE_EPI_GEGLU, true, 4>(a, s);
            case VALLE_EPI_F32OUT: return K == 256 ? dec_launch2<VALLE_EPI_F32OUT, true, 1>(a, s) : dec_launch2<VALLE_EPI_F32OUT, true, 4>(a, s);
            case DEC_EPI_QKV:      return K == 256 ? dec_launch2<DEC_EPI_QKV, true, 1>(a, s) : dec_launch2<DEC_EPI_QKV, true, 4>(a, s);
        }
    } else if (!norm && epilogue == VALLE_EPI_RESID) {
        if (K == 256) return dec_launch2<VALLE_EPI_RESID, false, 1>(a, s);
        if (K == 1024) return dec_launch2<VALLE_EPI_RESID, false, 4>(a, s);
        if (K == 4096) return dec_launch2<VALLE_EPI_RESID, false, 16>(a, s);
    }
    valle_set_error("decode linear: combination not built (K=%d norm=%d epilogue=%d)", K, (int)norm, epilogue);
    return VALLE_E_UNSUPPORTED;
}
