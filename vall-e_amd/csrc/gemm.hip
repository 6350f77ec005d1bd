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

// Shared epilogue of the dense GEMM kernels: this wave's 64(n) x 64(m) block, acc[i][j] = tile (n i, m j);
// a lane holds n = nb .. nb+3 (consecutive) for one m.
template <typename T, int EPI, int NI = 4, int MJ = 4>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[NI][MJ], int nw0, int mw0, int fr, int fq,
                                              const void* __restrict__ bias, const T* resid, void* Cout,
                                              int M, int N, int ldc) {
    // epilogue: lane holds n = nb + 0..3 (consecutive) for one m
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int nb = nw0 + i * 16 + fq * 4;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int m = mw0 + j * 16 + fr;
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

    gemm_epilogue<T, EPI>(acc, n0 + wn * 64, m0 + wm * 64, fr, fq, bias, resid, Cout, M, N, ldc);
}

// ---------------------------------------------------------------------------------------------
// bf16 dense GEMM with LDS-DMA staging: 128 x 128 tile, BK = 64, two LDS buffers, one barrier per
// K-step.  global_load_lds writes 64 lanes x 16 B contiguously (1 KiB per wave-instruction = 8 tile
// rows of 128 B), so the LDS image is linear; bank conflicts of the fragment reads are removed by
// XOR-swizzling the 16-B chunk index with (row & 7) on the GLOBAL source address and again on the read.
typedef __attribute__((address_space(3))) void lds_void;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_glds_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                             const void* __restrict__ bias, const bf16* resid,
                                                             void* Cout, int M, int N, int K, int ldc) {
    __shared__ __attribute__((aligned(1024))) char lds[2][2][128 * 128];     // [buf][W | X][128 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: each operand tile is 16 wave-instructions of 8 rows; this wave issues pieces wave*4 .. +3
    const bf16* wsrc[4];
    const bf16* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (row & 7);
        int n = n0 + row; n = n < N ? n : N - 1;          // rows past the edge re-read the last row (never stored)
        int m = m0 + row; m = m < M ? m : M - 1;
        wsrc[j] = W + (size_t)n * K + ch * 8;
        xsrc[j] = X + (size_t)m * K + ch * 8;
    }
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds(wsrc[j] + kt * 64, (lds_void*)&lds[buf][0][(wave * 4 + j) * 1024], 16, 0, 0);
            __builtin_amdgcn_global_load_lds(xsrc[j] + kt * 64, (lds_void*)&lds[buf][1][(wave * 4 + j) * 1024], 16, 0, 0);
        }
    };
    // fragment read offsets (bytes) of this lane inside a tile, per 16-row sub-tile i and k-half hf
    int aoff[4][2], boff[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int ra = wn * 64 + i * 16 + fr, rb = wm * 64 + i * 16 + fr;
            aoff[i][hf] = ra * 128 + (((hf * 4 + fq) ^ (ra & 7)) << 4);
            boff[i][hf] = rb * 128 + (((hf * 4 + fq) ^ (rb & 7)) << 4);
        }
    const int nt = K / 64;
    stage(0, 0);
    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // tile kt landed; everyone is done reading the other buffer
        if (kt + 1 < nt) stage(kt + 1, buf ^ 1);
        const char* Ws = lds[buf][0];
        const char* Xs = lds[buf][1];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(Ws + aoff[i][hf]);
                b[i] = *reinterpret_cast<const bf16x8*>(Xs + boff[i][hf]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    gemm_epilogue<bf16, EPI>(acc, n0 + wn * 64, m0 + wm * 64, fr, fq, bias, resid, Cout, M, N, ldc);
}

// Narrow-N variant (o_proj / down_proj, N = d_model): 64(n) x 128(m) tile, the 4 waves side by side along m
// (each 64 x 32), 24 KB per LDS stage -> 3 workgroups per CU and twice the workgroups, so a 1,024-column
// GEMM fills the chip in whole rounds.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_glds_n64_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                                 const void* __restrict__ bias, const bf16* resid,
                                                                 void* Cout, int M, int N, int K, int ldc) {
    __shared__ __attribute__((aligned(1024))) char lds[2][(64 + 128) * 128];   // [buf][W 64 rows | X 128 rows] x 128 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 128;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging: 24 pieces of 8 rows (W: 0..7, X: 8..23); this wave issues pieces wave*6 .. +5
    const bf16* src[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int piece = wave * 6 + j;
        const int row = piece * 8 + (lane >> 3);            // 0..63 = W rows, 64..191 = X rows
        const int ch = (lane & 7) ^ (row & 7);
        if (piece < 8) { int n = n0 + row; n = n < N ? n : N - 1; src[j] = W + (size_t)n * K + ch * 8; }
        else { int m = m0 + row - 64; m = m < M ? m : M - 1; src[j] = X + (size_t)m * K + ch * 8; }
    }
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            __builtin_amdgcn_global_load_lds(src[j] + kt * 64, (lds_void*)&lds[buf][(wave * 6 + j) * 1024], 16, 0, 0);
    };
    int aoff[4][2], boff[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = i * 16 + fr;
            aoff[i][hf] = ra * 128 + (((hf * 4 + fq) ^ (ra & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = 64 + wave * 32 + j * 16 + fr;    // (rb & 7) == ((rb - 64) & 7): the swizzle of the X row
            boff[j][hf] = rb * 128 + (((hf * 4 + fq) ^ (rb & 7)) << 4);
        }
    }
    const int nt = K / 64;
    stage(0, 0);
    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nt) stage(kt + 1, buf ^ 1);
        const char* Ls = lds[buf];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Ls + aoff[i][hf]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Ls + boff[j][hf]);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    gemm_epilogue<bf16, EPI, 4, 2>(acc, n0, m0 + wave * 32, fr, fq, bias, resid, Cout, M, N, ldc);
}

static int gemm_glds_dispatch(const void* A, const void* W, const void* bias, const void* resid, void* C,
                              int M, int N, int K, int ldc, int epi, hipStream_t s) {
    if (N <= 1024 && epi == VALLE_EPI_RESID) {      // narrow outputs: smaller tile, whole rounds of workgroups
        dim3 g64(ceil_div(N, 64), ceil_div(M, 128));
        gemm_glds_n64_bf16_kernel<VALLE_EPI_RESID><<<g64, 256, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        VALLE_LAUNCH_CHECK();
        return VALLE_OK;
    }
    dim3 grid(ceil_div(N, 128), ceil_div(M, 128));
#define GLDS_CASE(E) case E: gemm_glds_bf16_kernel<E><<<grid, 256, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc); break;
    switch (epi) {
        GLDS_CASE(VALLE_EPI_STORE) GLDS_CASE(VALLE_EPI_RESID) GLDS_CASE(VALLE_EPI_GEGLU) GLDS_CASE(VALLE_EPI_F32OUT)
        default: valle_set_error("valle_gemm: unknown epilogue %d", epi); return VALLE_E_ARG;
    }
#undef GLDS_CASE
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
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
    if (dtype == VALLE_BF16 && K % 64 == 0 && M >= 64) return gemm_glds_dispatch(A, W, bias, resid, C, M, N, K, ldc, epilogue, s);
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

template <int EPI, bool NORM, int MT, int STEPS, int NW>
__global__ __launch_bounds__(NW * 64) void dec_linear_kernel(DecArgs a) {
    constexpr int NT = NW * 64;
    constexpr int KW = STEPS * 64;          // k range of one wave
    constexpr int K = KW * NW;
    constexpr int XS = K + 8;               // LDS row stride (elements): +16 B shifts banks per row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kb0 = wave * KW;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    float* s_part = reinterpret_cast<float*>(dyn_lds);                       // [NW][MT][16][17]
    float* s_ss = s_part + NW * MT * 16 * 17;                                // [32]
    bf16* xs = reinterpret_cast<bf16*>(s_ss + 32);                           // NORM: [MT*16][XS]

    // ---- 1. activation loads are issued FIRST: vector-memory results return in issue order, so
    // the (L2-resident) activations land while the weight stream is still in flight
    constexpr int CPR = K / 8;                           // 16-B chunks per row
    constexpr int NCH = NORM ? MT * 16 * CPR / NT : 1;   // chunks per thread
    constexpr int RED = CPR >= 64 ? 64 : CPR;            // lanes that share a row
    bf16x8 xc[NCH], gc[NCH];
    bf16x8 x0[NORM ? 1 : STEPS][MT], x1[NORM ? 1 : STEPS][MT];
    if constexpr (NORM) {
        if (tid < 32) s_ss[tid] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NT * i, row = c / CPR, col = (c % CPR) * 8;
            xc[i] = *reinterpret_cast<const bf16x8*>(a.x + (size_t)(row < a.B ? row : a.B - 1) * K + col);
            gc[i] = *reinterpret_cast<const bf16x8*>(a.norm_w + col);
        }
    } else {
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = t * 16 + fr;
                const bf16* xp = a.x + (size_t)(m < a.B ? m : a.B - 1) * K + kb0 + s * 64 + fq * 8;
                x0[s][t] = *reinterpret_cast<const bf16x8*>(xp);
                x1[s][t] = *reinterpret_cast<const bf16x8*>(xp + 32);
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 1b. the weight stream: W is pre-tiled [N/16][K/64][2][64][8] (valle_hip.h), so this wave's
    // slice is 2*STEPS KiB of contiguous memory read 1 KiB per instruction
    const bf16* wt = a.W + ((size_t)blockIdx.x * (K / 64) + wave * STEPS) * 1024 + lane * 8;
    bf16x8 w0[STEPS], w1[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        w0[s] = *reinterpret_cast<const bf16x8*>(wt + s * 1024);
        w1[s] = *reinterpret_cast<const bf16x8*>(wt + s * 1024 + 512);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. activation operand
    if constexpr (NORM) {
        // all threads: sum of squares per row (atomics in LDS), normalise, -> LDS
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NT * i, row = c / CPR;
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)xc[i][e]; ss += f * f; }
#pragma unroll
            for (int o = RED / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if ((lane & (RED - 1)) == 0) atomicAdd(&s_ss[row], ss);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NT * i, row = c / CPR, col = (c % CPR) * 8;
            const float ri = __builtin_amdgcn_rsqf(s_ss[row] * (1.0f / (float)K) + a.eps);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)gc[i][e] * (float)(bf16)((float)xc[i][e] * ri));
            *reinterpret_cast<bf16x8*>(xs + row * XS + col) = o;
        }
        __syncthreads();
    }
    // ---- 3. MFMAs (rows m >= B produce garbage columns that are never stored)
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            bf16x8 b0, b1;
            if constexpr (NORM) {
                const bf16* xp = xs + (t * 16 + fr) * XS + kb0 + s * 64 + fq * 8;
                b0 = *reinterpret_cast<const bf16x8*>(xp);
                b1 = *reinterpret_cast<const bf16x8*>(xp + 32);
            } else {
                b0 = x0[s][t]; b1 = x1[s][t];
            }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[s], b0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s], b1, acc[t], 0, 0, 0);
        }
    }
    // ---- 4. cross-wave reduction through LDS
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_part[((wave * MT + t) * 16 + fq * 4 + r) * 17 + fr] = acc[t][r];
    __syncthreads();
    // thread -> (m, pair of n): 16*MT rows x 8 pairs
    for (int idx = tid; idx < 16 * MT * 8; idx += NT) {
        const int m = idx & (16 * MT - 1), np = idx / (16 * MT);
        if (m >= a.B) continue;
        const int t = m >> 4, mm = m & 15;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            v0 += s_part[((w * MT + t) * 16 + 2 * np) * 17 + mm];
            v1 += s_part[((w * MT + t) * 16 + 2 * np + 1) * 17 + mm];
        }
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
                const float x0f = (float)o0, x1f = (float)o1;
                o0 = (bf16)(__fadd_rn(__fmul_rn(x0f, cs), __fmul_rn(-x1f, sn)));
                o1 = (bf16)(__fadd_rn(__fmul_rn(x1f, cs), __fmul_rn(x0f, sn)));
            }
            bf16x2 ov; ov[0] = o0; ov[1] = o1;
            if (which == 0) {
                *reinterpret_cast<bf16x2*>((bf16*)a.out + (size_t)m * a.ldo + n) = ov;
            } else {
                const int sl = a.slot ? a.slot[m] : m;
                bf16* cp = (which == 1 ? a.kc : a.vc) + (((size_t)sl * a.H + h) * a.cache_T + a.ctx_len[m]) * 64 + e;
                *reinterpret_cast<bf16x2*>(cp) = ov;
            }
        }
    }
}

template <int EPI, bool NORM, int STEPS, int NW>
static int dec_launch2(const DecArgs& a, hipStream_t s) {
    const int grid = ceil_div(a.N, 16);
    constexpr int K = STEPS * 64 * NW;
    auto lds = [&](int MT) { return (size_t)(NW * MT * 16 * 17 + 32) * 4 + (NORM ? (size_t)MT * 16 * (K + 8) * 2 : 0); };
    if (a.B <= 16) dec_linear_kernel<EPI, NORM, 1, STEPS, NW><<<grid, NW * 64, lds(1), s>>>(a);
    else {
        static bool attr_set = false;   // > 64 KiB of dynamic LDS needs the opt-in once per kernel
        if (!attr_set && lds(2) > 65536) {
            VALLE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_linear_kernel<EPI, NORM, 2, STEPS, NW>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(2)));
            attr_set = true;
        }
        dec_linear_kernel<EPI, NORM, 2, STEPS, NW><<<grid, NW * 64, lds(2), s>>>(a);
    }
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
            case VALLE_EPI_GEGLU:  return K == 256 ? dec_launch2<VALLE_EPI_GEGLU, true, 1, 4>(a, s) : dec_launch2<VALLE_EPI_GEGLU, true, 4, 4>(a, s);
            case VALLE_EPI_F32OUT: return K == 256 ? dec_launch2<VALLE_EPI_F32OUT, true, 1, 4>(a, s) : dec_launch2<VALLE_EPI_F32OUT, true, 4, 4>(a, s);
            case DEC_EPI_QKV:      return K == 256 ? dec_launch2<DEC_EPI_QKV, true, 1, 4>(a, s) : dec_launch2<DEC_EPI_QKV, true, 4, 4>(a, s);
        }
    } else if (!norm && epilogue == VALLE_EPI_RESID) {
        if (K == 256) return dec_launch2<VALLE_EPI_RESID, false, 1, 4>(a, s);
        if (K == 1024) return dec_launch2<VALLE_EPI_RESID, false, 2, 8>(a, s);
        if (K == 4096) return dec_launch2<VALLE_EPI_RESID, false, 8, 8>(a, s);
    }
    valle_set_error("decode linear: combination not built (K=%d norm=%d epilogue=%d)", K, (int)norm, epilogue);
    return VALLE_E_UNSUPPORTED;
}
