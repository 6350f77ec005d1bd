// MFMA GEMMs of the AR+NAR forward (gfx950).
//   valle_gemm        dense passes (prefill, NAR levels, NAR-len steps): C = A . W^T, epilogues
// (the AR decode step's linears are in dec_linear.hip)
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
                                              int M, int N, int ldc, const float* rope_rows = nullptr) {
    if constexpr (EPI == GEMM_EPI_ROPE) {
        // QKV projection: q and k columns are pair-interleaved, a lane's 4 columns are two RoPE pairs
        // (arch/llama.py:48-61); same arithmetic as rope_kv_kernel: operands rounded to the storage type, the two
        // products rounded separately.  rope_rows (LDS): the (cos, sin) rows of this wave's positions (its first m row first),
        // [row][68 floats], staged by the kernel -- a row serves every head, so it is fetched once per workgroup.
        T* C = (T*)Cout;
        typedef T vec4 __attribute__((ext_vector_type(4)));
        const int d = N / 3;
        f32x4 rp[NI][MJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j)
                rp[i][j] = *reinterpret_cast<const f32x4*>(rope_rows + (j * 16 + fr) * 68 + (i & 3) * 16 + fq * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nb = nw0 + i * 16 + fq * 4;
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const int m = mw0 + j * 16 + fr;
                if (m >= M || nb >= N) continue;
                vec4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(acc[i][j][r]);
                if (nb < 2 * d) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float cs = to_f32<T>(from_f32<T>(rp[i][j][2 * q])), sn = to_f32<T>(from_f32<T>(rp[i][j][2 * q + 1]));
                        const float u = to_f32<T>(o[2 * q]), w = to_f32<T>(o[2 * q + 1]);
                        o[2 * q] = from_f32<T>(__fadd_rn(__fmul_rn(u, cs), __fmul_rn(-w, sn)));
                        o[2 * q + 1] = from_f32<T>(__fadd_rn(__fmul_rn(w, cs), __fmul_rn(u, sn)));
                    }
                }
                *reinterpret_cast<vec4*>(C + (size_t)m * ldc + nb) = o;
            }
        }
        return;
    }
    // epilogue: lane holds n = nb + 0..3 (consecutive) for one m
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int nb = nw0 + i * 16 + fq * 4;
        // the four columns' bias once per column group (one 8-byte load; it does not depend on the row tile j)
        [[maybe_unused]] float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == VALLE_EPI_RESID_BIAS || EPI == VALLE_EPI_GELU || EPI == VALLE_EPI_STORE) {
            const T* bt = (const T*)bias;
            if (bt != nullptr && nb < N) {
                if (nb + 3 < N && sizeof(T) == 2 && (reinterpret_cast<uintptr_t>(bt) & 7) == 0) {
                    typedef T vec4 __attribute__((ext_vector_type(4)));
                    const vec4 b4 = *reinterpret_cast<const vec4*>(bt + nb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = to_f32<T>(b4[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = nb + r < N ? to_f32<T>(bt[nb + r]) : 0.f;
                }
            }
        }
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
                o[0] = from_f32<T>(to_f32<T>(from_f32<T>(gelu_act<T>(to_f32<T>(from_f32<T>(v[0]))))) * to_f32<T>(from_f32<T>(v[1])));
                o[1] = from_f32<T>(to_f32<T>(from_f32<T>(gelu_act<T>(to_f32<T>(from_f32<T>(v[2]))))) * to_f32<T>(from_f32<T>(v[3])));
                *reinterpret_cast<vec2*>(C + (size_t)m * ldc + (nb >> 1)) = o;
            } else {
                T* C = (T*)Cout;
                if (nb + 3 < N && (ldc & 3) == 0) {
                    typedef T vec4 __attribute__((ext_vector_type(4)));
                    vec4 o;
                    if constexpr (EPI == VALLE_EPI_RESID) {
                        const vec4 rr = *reinterpret_cast<const vec4*>(resid + (size_t)m * ldc + nb);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(to_f32<T>(rr[r]) + to_f32<T>(from_f32<T>(v[r])));
                    } else if constexpr (EPI == VALLE_EPI_RESID_BIAS) {
                        const vec4 rr = *reinterpret_cast<const vec4*>(resid + (size_t)m * ldc + nb);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(to_f32<T>(rr[r]) + to_f32<T>(from_f32<T>(v[r] + bv[r])));
                    } else if constexpr (EPI == VALLE_EPI_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(gelu_act<T>(to_f32<T>(from_f32<T>(v[r] + bv[r]))));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r] + bv[r]);
                    }
                    *reinterpret_cast<vec4*>(C + (size_t)m * ldc + nb) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (nb + r >= N) continue;
                        float o = v[r];
                        if constexpr (EPI == VALLE_EPI_RESID) o = to_f32<T>(resid[(size_t)m * ldc + nb + r]) + to_f32<T>(from_f32<T>(o));
                        else if constexpr (EPI == VALLE_EPI_RESID_BIAS) o = to_f32<T>(resid[(size_t)m * ldc + nb + r]) + to_f32<T>(from_f32<T>(o + bv[r]));
                        else if constexpr (EPI == VALLE_EPI_GELU) o = gelu_act<T>(to_f32<T>(from_f32<T>(o + bv[r])));
                        else if constexpr (EPI == VALLE_EPI_STORE) o += bv[r];
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

// NORM variants (RMSNorm folded into the weights, models/arch/llama.py:64-78 ahead of q/k/v and gate/up): the GEMM runs on
// the un-normalised rows, every wave adds up the squares of the activation fragments it reads anyway (K = the whole row),
// and the accumulators are scaled by 1/rms(row) before the epilogue rounds them.
__device__ __forceinline__ float ssq8(const bf16x8& v, float acc) {
    // (pairs by shufflevector: a bit_cast of the fragment to a 4-dword vector made the compiler read dword 0 four times)
    const bf16x2 p0 = __builtin_shufflevector(v, v, 0, 1), p1 = __builtin_shufflevector(v, v, 2, 3);
    const bf16x2 p2 = __builtin_shufflevector(v, v, 4, 5), p3 = __builtin_shufflevector(v, v, 6, 7);
    acc = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(p3, p3, acc, false);
    return acc;
}
// a lane's sum covers the k columns of its fq; the row's total is the sum over the 4 lanes fr + 16 fq
__device__ __forceinline__ float row_rinv(float ss, int K, float eps) {
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    return 1.0f / sqrtf(ss * (1.0f / (float)K) + eps);
}

#ifndef GLDS_SWZ_MAX_NX
#define GLDS_SWZ_MAX_NX 24
#endif
template <int EPI, bool NORM = false>
__global__ __launch_bounds__(256) void gemm_glds_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                             const void* __restrict__ bias, const bf16* resid,
                                                             void* Cout, int M, int N, int K, int ldc, float eps = 0.f) {
    __shared__ __attribute__((aligned(1024))) char lds[2][2][128 * 128];     // [buf][W | X][128 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    // (workgroups dealt to XCDs by row tile when there are few column tiles: see tile_nm in gemm_pp256_bf16_kernel)
    int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
    {
        const int nxt = gridDim.x, lin = blockIdx.x + nxt * blockIdx.y;
        if (nxt <= GLDS_SWZ_MAX_NX && lin < nxt * ((int)gridDim.y & ~7)) {
            const int sl = lin >> 3;
            n0 = (sl % nxt) * 128; m0 = ((lin & 7) + 8 * (sl / nxt)) * 128;
        }
    }

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
    // EPI_ROPE: the (cos, sin) rows of this workgroup's 128 positions are requested now (8 x 16 B per thread), kept in
    // registers through the main loop and handed to the epilogue through LDS
    [[maybe_unused]] f32x4 rrow[8];
    if constexpr (EPI == GEMM_EPI_ROPE) {
        const int32_t* pos = reinterpret_cast<const int32_t*>(resid);
        const float* tab = reinterpret_cast<const float*>(bias);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int idx = tid + 256 * t;                 // row = idx / 16, 16-byte piece = idx % 16
            int m = m0 + (idx >> 4); m = m < M ? m : M - 1;
            rrow[t] = *reinterpret_cast<const f32x4*>(tab + (size_t)pos[m] * 64 + (idx & 15) * 4);
        }
    }
    const int nt = K / 64;
    [[maybe_unused]] float ssq[4] = {0.f, 0.f, 0.f, 0.f};
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
            if constexpr (NORM) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ssq[j] = ssq8(b[j], ssq[j]);
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
    if constexpr (NORM) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float rinv = row_rinv(ssq[j], K, eps);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] *= rinv;
        }
    }
    if constexpr (EPI == GEMM_EPI_ROPE) {
        float* rope_rows = reinterpret_cast<float*>(&lds[0][0][0]);     // 128 x 68 floats = 34 KB of the 64 KB
        __syncthreads();                                   // the last tile has been read by every wave
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int idx = tid + 256 * t;
            *reinterpret_cast<f32x4*>(rope_rows + (idx >> 4) * 68 + (idx & 15) * 4) = rrow[t];
        }
        __syncthreads();
        gemm_epilogue<bf16, EPI>(acc, n0 + wn * 64, m0 + wm * 64, fr, fq, bias, resid, Cout, M, N, ldc, rope_rows + wm * 64 * 68);
    } else {
        gemm_epilogue<bf16, EPI>(acc, n0 + wn * 64, m0 + wm * 64, fr, fq, bias, resid, Cout, M, N, ldc);
    }
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

// ---------------------------------------------------------------------------------------------
// Wide GEMMs (QKV, gate/up): 256 x 256 tile, BK = 64, 8 waves in two groups that ping-pong on the matrix
// core.  Every wave owns a 128(n) x 64(m) block of the output = four 64 x 32 quadrants; per K-tile it
// alternates a READ slot (fragment reads for the next quadrant, or the LDS-DMA of a later tile) and an MFMA
// slot (one quadrant, 16 MFMAs), with a workgroup barrier after every slot.  Group 1 (waves 4-7, which share
// their SIMDs with waves 0-3) runs one slot behind group 0, so while one wave of a SIMD issues MFMAs the other
// one reads LDS.  Two 64-KB LDS buffers (swizzled linear image as above); tile t+1 is staged into the other
// buffer during the first three read slots of tile t -- that buffer's last fragment reads (tile t-1) were
// waited for (lgkmcnt(0) before the barrier that ends a read slot) before tile t began -- and is waited for
// (vmcnt(0)) in the 4th read slot, two barriers before its first read.
// MQR = m rows per wave: 64 (256-row tiles) or 48 (192-row tiles: the second B half is one 16-row tile instead of two --
// for row counts whose 256-row tiling leaves CUs idle, see gemm_glds_dispatch)
// PERSIST: a one-dimensional grid of at most one workgroup per CU, every workgroup walks tiles t = blockIdx.x, + gridDim.x, ...
// (t = n tile + n tiles x m tile, the order a two-dimensional grid is dispatched in).  The first K-tile of the NEXT output
// tile is staged during the last K-tile of this one (its buffer is free by then) and the second one behind the last reads,
// in front of the epilogue, so a tile after the first starts without a dispatch, without a cold prologue and with the
// epilogue's stores still in flight.  K / 64 must be even (the buffers alternate across the tile boundary).
#ifndef PP_SWZ_MAX_NX
#define PP_SWZ_MAX_NX 16
#endif
template <int EPI, bool NORM = false, int MQR = 64, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm_pp256_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                              const void* __restrict__ bias, const bf16* resid,
                                                              void* Cout, int M, int N, int K, int ldc, float eps = 0.f) {
    // [buf][W 256 rows | X 256 rows] x 128 B; behind them, in the persistent RoPE form, what the tile's (cos, sin) rows need beyond buffer 1
    constexpr int ROPE_TAIL = (EPI == GEMM_EPI_ROPE && PERSIST && MQR * 4 * 68 * 4 > 65536) ? MQR * 4 * 68 * 4 - 65536 : 0;
    __shared__ __attribute__((aligned(1024))) char lds[2 * 65536 + ROPE_TAIL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 3;
    const int nh = wi & 1, mq = (wi >> 1) + 2 * grp;                       // n half (128 rows), m quarter (64 rows)
    const int fr = lane & 15, fq = lane >> 4;
    static_assert(MQR == 64 || MQR == 48, "m rows per wave");
    constexpr int TMR = 4 * MQR, MJ = MQR / 16, PPW = (32 + TMR / 8) / 8;    // tile rows, 16-row m tiles per wave, pieces per wave
    // Tile t (in dispatch order; workgroup b lands on XCD b mod 8) -> (n tile, m tile).  Few n tiles (o_proj / down_proj: 4,
    // q/k/v: 12): the plain order n = t mod nxt deals the workgroups that share an m tile's activation rows to as many XCDs
    // as there are n tiles, and every one of those L2s fetches the rows (down_proj at 12,176 rows: 441 MB over the fabric for
    // 108 MB of operands); instead XCD x takes the m tiles x, x + 8, ... with ALL their n tiles, so a row tile is fetched once.
    // (Many n tiles -- gate/up: 32 -- keep the plain order: 8 m tiles x 4 n tiles per XCD and round is the smaller footprint.)
    const int nxt = (N + 255) >> 8, nyt = (M + TMR - 1) / TMR;
    [[maybe_unused]] const int ntiles = nxt * nyt;
    const int nswz = nxt <= PP_SWZ_MAX_NX ? nxt * (nyt & ~7) : 0;          // tiles [0, nswz) are dealt by XCD
    auto tile_nm = [&](int t, int& tn0, int& tm0) {
        if (t < nswz) { const int sl = t >> 3; tn0 = (sl % nxt) * 256; tm0 = ((t & 7) + 8 * (sl / nxt)) * TMR; }
        else { tn0 = (t % nxt) * 256; tm0 = (t / nxt) * TMR; }
    };
    int tile = PERSIST ? (int)blockIdx.x : (int)(blockIdx.x + gridDim.x * blockIdx.y);
    int n0, m0;
    tile_nm(tile, n0, m0);

    f32x4 acc[8][MJ];

    // staging: a K-tile is 64 pieces of 8 rows (W: 0..31, X: 32..63); this wave issues pieces wave*8 .. +7
    const bf16* src[8];
    auto set_src = [&](int tn0, int tm0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int piece = wave * PPW + (j < PPW ? j : PPW - 1);
            const int row = piece * 8 + (lane >> 3);                       // 0..255 W rows, 256..256+TMR-1 X rows
            const int ch = (lane & 7) ^ (row & 7);
            if (piece < 32) { int n = tn0 + row; n = n < N ? n : N - 1; src[j] = W + (size_t)n * K + ch * 8; }
            else { int m = tm0 + row - 256; m = m < M ? m : M - 1; src[j] = X + (size_t)m * K + ch * 8; }
        }
    };
    set_src(n0, m0);
    auto stage_part = [&](int kt, int buf, int j0, int j1) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j >= j0 && j < j1 && j < PPW)
                __builtin_amdgcn_global_load_lds(src[j] + kt * 64, (lds_void*)&lds[buf * 65536 + (wave * PPW + j) * 1024], 16, 0, 0);
    };
    auto stage = [&](int kt, int buf) { stage_part(kt, buf, 0, 8); };
    // fragment addressing: row = const + fr, so (row & 7) == (fr & 7) and the swizzle depends on the k-half only
    const int sw0 = ((fq ^ (fr & 7)) << 4), sw1 = (((4 + fq) ^ (fr & 7)) << 4);
    const int abase = (nh * 128 + fr) * 128, bbase = (256 + mq * MQR + fr) * 128;
    bf16x8 A[4][2], B0[2][2], B1[2][2];
    auto read_a = [&](const char* L, int i) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            A[rt][0] = *reinterpret_cast<const bf16x8*>(L + abase + (i * 64 + rt * 16) * 128 + sw0);
            A[rt][1] = *reinterpret_cast<const bf16x8*>(L + abase + (i * 64 + rt * 16) * 128 + sw1);
        }
    };
    auto read_b = [&](const char* L, int j, bf16x8 (&Bf)[2][2]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            if (j * 2 + ct >= MJ) continue;
            Bf[ct][0] = *reinterpret_cast<const bf16x8*>(L + bbase + (j * 32 + ct * 16) * 128 + sw0);
            Bf[ct][1] = *reinterpret_cast<const bf16x8*>(L + bbase + (j * 32 + ct * 16) * 128 + sw1);
        }
    };
    [[maybe_unused]] float ssq[4] = {0.f, 0.f, 0.f, 0.f};                 // per 16-row m tile j * 2 + ct
    auto sq_b = [&](int j, bf16x8 (&Bf)[2][2]) {                           // (issued in an MFMA slot, behind the MFMAs)
        if constexpr (NORM) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                if (j * 2 + ct < MJ) ssq[j * 2 + ct] = ssq8(Bf[ct][1], ssq8(Bf[ct][0], ssq[j * 2 + ct]));
        }
    };
    auto quad = [&](int i, int j, bf16x8 (&Bf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    if (j * 2 + ct < MJ)
                        acc[i * 4 + rt][j * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rt][hf], Bf[ct][hf], acc[i * 4 + rt][j * 2 + ct], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define PP_END_READ()  do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define PP_END_MFMA()  __builtin_amdgcn_s_barrier()

    // EPI_ROPE (q/k/v projection, bias = RoPE table, resid = positions): the (cos, sin) rows of the tile's positions are
    // requested now (16 B pieces, TMR * 16 of them over 512 threads), live in registers through the main loop and reach the
    // epilogue through LDS
    constexpr int RPT = TMR * 16 / 512;
    [[maybe_unused]] f32x4 rrow[RPT];
    const int nt = K / 64;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  for (;;) {                                                               // tiles of this workgroup (one unless PERSIST)
    [[maybe_unused]] const bool has_next = PERSIST && tile + (int)gridDim.x < ntiles;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (NORM) { ssq[0] = ssq[1] = ssq[2] = ssq[3] = 0.f; }
    if constexpr (EPI == GEMM_EPI_ROPE) {
        const int32_t* pos = reinterpret_cast<const int32_t*>(resid);
        const float* tab = reinterpret_cast<const float*>(bias);
#pragma unroll
        for (int t = 0; t < RPT; ++t) {
            const int idx = tid + 512 * t;                 // row = idx / 16, 16-byte piece = idx % 16
            int m = m0 + (idx >> 4); m = m < M ? m : M - 1;
            rrow[t] = *reinterpret_cast<const f32x4*>(tab + (size_t)pos[m] * 64 + (idx & 15) * 4);
        }
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();                            // group 1 runs one slot behind
    for (int kt = 0; kt < nt; ++kt) {
        const char* L = lds + (kt & 1) * 65536;
        // tile kt+1 goes into the OTHER buffer, whose last reads (tile kt-1) ended before this iteration began;
        // its 8 DMA pieces per wave are spread over the read slots (a piece costs ~60-100 cycles to issue)
        bool st = kt >= 1 && kt + 1 < nt;
        int ktn = kt + 1;
        if constexpr (PERSIST) {
            if (has_next && kt + 1 == nt) {                                // K-tile 0 of the NEXT output tile (nt is even: buffer 0)
                int n2, m2;
                tile_nm(tile + (int)gridDim.x, n2, m2);
                set_src(n2, m2);
                st = true; ktn = 0;
            }
        }
        read_a(L, 0); read_b(L, 0, B0);                                   // R1
        if (st) stage_part(ktn, (kt + 1) & 1, 0, 2);
        PP_END_READ();
        quad(0, 0, B0);                                                   // M1
        sq_b(0, B0);
        PP_END_MFMA();
        read_b(L, 1, B1);                                                 // R2
        if (st) stage_part(ktn, (kt + 1) & 1, 2, 5);
        PP_END_READ();
        quad(0, 1, B1);                                                   // M2
        sq_b(1, B1);
        PP_END_MFMA();
        read_a(L, 1);                                                     // R3
        if (st) stage_part(ktn, (kt + 1) & 1, 5, 8);
        PP_END_READ();
        quad(1, 1, B1);                                                   // M3
        PP_END_MFMA();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // R4: tile kt+1 has landed (my pieces)
        __builtin_amdgcn_s_barrier();
        quad(1, 0, B0);                                                   // M4
        PP_END_MFMA();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    // PERSIST: every wave's pieces of the next tile's K-tile 0 have landed in buffer 0 (own vmcnt(0) in R4, then barriers) and
    // buffer 1 has been read for the last time: its K-tile 1 goes out now, in front of the epilogue's stores
    if constexpr (PERSIST && EPI != GEMM_EPI_ROPE) { if (has_next) stage(1, 1); }
    if constexpr (NORM) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const float rinv = row_rinv(ssq[j], K, eps);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] *= rinv;
        }
    }
    if constexpr (EPI == GEMM_EPI_ROPE) {
        // TMR x 68 floats: 52 KB at 192-row tiles, 70 KB at 256.  PERSIST: from buffer 1 on (buffer 0 holds the next tile's
        // K-tile 0; 256-row tiles run 4 KB past buffer 1 into ROPE_TAIL); K-tile 1 is staged once every wave has taken its rows
        float* rope_rows = reinterpret_cast<float*>(lds + (PERSIST ? 65536 : 0));
        __syncthreads();                                                   // the last tile has been read by every wave
#pragma unroll
        for (int t = 0; t < RPT; ++t) {
            const int idx = tid + 512 * t;
            *reinterpret_cast<f32x4*>(rope_rows + (idx >> 4) * 68 + (idx & 15) * 4) = rrow[t];
        }
        __syncthreads();
        gemm_epilogue<bf16, EPI, 8, MJ>(acc, n0 + nh * 128, m0 + mq * MQR, fr, fq, bias, resid, Cout, M, N, ldc, rope_rows + mq * MQR * 68);
        if constexpr (PERSIST) {
            if (has_next) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stage(1, 1);
            }
        }
    } else {
        gemm_epilogue<bf16, EPI, 8, MJ>(acc, n0 + nh * 128, m0 + mq * MQR, fr, fq, bias, resid, Cout, M, N, ldc);
    }
    if constexpr (!PERSIST) break;
    else {
        if (!has_next) break;
        tile += (int)gridDim.x;
        tile_nm(tile, n0, m0);
    }
  }
#undef PP_END_READ
#undef PP_END_MFMA
}

// Persistent form of the ping-pong kernel: when a launch is several rounds of one-workgroup-per-CU tiles and K / 64 is even.
// VALLE_GEMM_PERSIST=0 (dev switch, read once) keeps the one-tile-per-workgroup launches for A/B runs.
static int pp_persist_grid(int ntiles, int K) {
    static const int enabled = [] { const char* e = getenv("VALLE_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    if (!enabled || ntiles <= n_cu || (K / 64) % 2 != 0 || K < 128) return 0;
    return n_cu;
}

static int gemm_glds_dispatch(const void* A, const void* W, const void* bias, const void* resid, void* C,
                              int M, int N, int K, int ldc, int epi, hipStream_t s) {
    // down_proj (K = d_ff, narrow N): 64 K-tiles amortise the 256^2 ping-pong kernel's fill and epilogue even
    // though N = 1024 gives only 4 x ceil(M/256) workgroups (measured 113.6 vs 133.5 us at M = 12,176); o_proj (K = 1,024)
    // gains too once the 192-row tiles fill the chip in one round (43.1 vs 47.1 us for the 64 x 128 kernel, tools/micro_gemm)
    if (epi == VALLE_EPI_RESID && K >= 1024 && M >= 2048 && N >= 256) {
        // one workgroup per CU: what counts is the number of rounds of 256 workgroups.  192-row tiles do 3/4 of the MFMA work
        // of a 256-row tile in ~0.85 of its time; they win where they save a round or fill idle CUs (M = 12,176, N = 1,024:
        // 192 workgroups of 256 rows leave 64 CUs idle, 256 workgroups of 192 rows fill the chip)
        const int nx = ceil_div(N, 256);
        const float r256 = (float)ceil_div(nx * ceil_div(M, 256), 256), r192 = 0.85f * (float)ceil_div(nx * ceil_div(M, 192), 256);
        // (several rounds -- the NAR-len pass of 64 utterances has 4 -- run as one persistent launch)
        if (r192 < r256) {
            dim3 g192(nx, ceil_div(M, 192));
            const int pg = pp_persist_grid(g192.x * g192.y, K);
            if (pg) gemm_pp256_bf16_kernel<VALLE_EPI_RESID, false, 48, true><<<pg, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
            else gemm_pp256_bf16_kernel<VALLE_EPI_RESID, false, 48><<<g192, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        } else {
            dim3 g256(nx, ceil_div(M, 256));
            const int pg = pp_persist_grid(g256.x * g256.y, K);
            if (pg) gemm_pp256_bf16_kernel<VALLE_EPI_RESID, false, 64, true><<<pg, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
            else gemm_pp256_bf16_kernel<VALLE_EPI_RESID><<<g256, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        }
        VALLE_LAUNCH_CHECK();
        return VALLE_OK;
    }
    // the second linear of the version >= 7 encoder FFN (bias + residual, base_v2.py:120-134) at dense-pass row counts: the
    // ping-pong tiles, 192 or 256 rows by rounds (57,600 x 1,024 x 2,048: 381 -> 324 us; NAR-len of all levels 383.0 -> 380.8
    // ms).  The GELU linear was tried on them too: 570 -> 497 us alone on random operands, but 383 -> 385 ms in the NAR-len
    // run (the erf of every output sets the pace on real activations) -- left on the 128 x 128 tiles; bias alone likewise
    // (N = d_model at 7,200 rows: 45 us there against 51 here).
    if (epi == VALLE_EPI_RESID_BIAS && K >= 1024 && M >= 2048 && N >= 256) {
        const int nx = ceil_div(N, 256);
        const float r256 = (float)ceil_div(nx * ceil_div(M, 256), 256), r192 = 0.85f * (float)ceil_div(nx * ceil_div(M, 192), 256);
        const bool t192 = r192 < r256;
        const dim3 g(nx, ceil_div(M, t192 ? 192 : 256));
        const int pg = pp_persist_grid(g.x * g.y, K);
#define PP_CASE(E) case E: \
        if (t192 && pg) gemm_pp256_bf16_kernel<E, false, 48, true><<<pg, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc); \
        else if (t192) gemm_pp256_bf16_kernel<E, false, 48><<<g, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc); \
        else if (pg) gemm_pp256_bf16_kernel<E, false, 64, true><<<pg, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc); \
        else gemm_pp256_bf16_kernel<E><<<g, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc); \
        break;
        switch (epi) { PP_CASE(VALLE_EPI_RESID_BIAS) }
#undef PP_CASE
        VALLE_LAUNCH_CHECK();
        return VALLE_OK;
    }
    if (N <= 1024 && epi == VALLE_EPI_RESID) {      // narrow outputs: smaller tile, whole rounds of workgroups
        dim3 g64(ceil_div(N, 64), ceil_div(M, 128));
        gemm_glds_n64_bf16_kernel<VALLE_EPI_RESID><<<g64, 256, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        VALLE_LAUNCH_CHECK();
        return VALLE_OK;
    }
    // Wide outputs only: at K = 1024 (16 K-tiles) a one-workgroup-per-CU kernel pays its pipeline fill and its
    // epilogue once per tile with nothing to overlap them, so it wins where the grid is several whole rounds of
    // 256 workgroups (gate/up: +12 % over the 128^2 kernel) and loses on the QKV shape (2.25 rounds: -10 %).
    if (N >= 4096 && M >= 1024 && (epi == VALLE_EPI_STORE || epi == VALLE_EPI_GEGLU)) {
        dim3 g256(ceil_div(N, 256), ceil_div(M, 256));
        const int pg = pp_persist_grid(g256.x * g256.y, K);
        if (epi == VALLE_EPI_STORE && pg) gemm_pp256_bf16_kernel<VALLE_EPI_STORE, false, 64, true><<<pg, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        else if (epi == VALLE_EPI_STORE) gemm_pp256_bf16_kernel<VALLE_EPI_STORE><<<g256, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        else if (pg) gemm_pp256_bf16_kernel<VALLE_EPI_GEGLU, false, 64, true><<<pg, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        else gemm_pp256_bf16_kernel<VALLE_EPI_GEGLU><<<g256, 512, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc);
        VALLE_LAUNCH_CHECK();
        return VALLE_OK;
    }
    dim3 grid(ceil_div(N, 128), ceil_div(M, 128));
#define GLDS_CASE(E) case E: gemm_glds_bf16_kernel<E><<<grid, 256, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc); break;
    switch (epi) {
        GLDS_CASE(VALLE_EPI_STORE) GLDS_CASE(VALLE_EPI_RESID) GLDS_CASE(VALLE_EPI_GEGLU) GLDS_CASE(VALLE_EPI_F32OUT)
        GLDS_CASE(VALLE_EPI_GELU) GLDS_CASE(VALLE_EPI_RESID_BIAS)
        default: valle_set_error("valle_gemm: unknown epilogue %d", epi); return VALLE_E_ARG;
    }
#undef GLDS_CASE
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_gemm_qkv_rope(const void* A, const void* Wqkv, const float* rope, const int32_t* pos, void* qkv,
                                   int M, int d, int K, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    VALLE_REQUIRE(A && Wqkv && rope && pos && qkv, VALLE_E_ARG, "valle_gemm_qkv_rope: null pointer");
    VALLE_REQUIRE(d % 64 == 0 && K % 64 == 0 && M >= 64, VALLE_E_UNSUPPORTED, "valle_gemm_qkv_rope: M=%d d=%d K=%d", M, d, K);
    const int N = 3 * d;
    dim3 grid(ceil_div(N, 128), ceil_div(M, 128));
    gemm_glds_bf16_kernel<GEMM_EPI_ROPE><<<grid, 256, 0, s>>>((const bf16*)A, (const bf16*)Wqkv, rope, (const bf16*)pos, qkv, M, N, K, N);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_gemm_normed(const void* A, const void* Wf, void* C, int M, int N, int K, int ldc, int epilogue,
                                 float eps, const float* rope, const int32_t* pos, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    VALLE_REQUIRE(A && Wf && C, VALLE_E_ARG, "valle_gemm_normed: null pointer");
    VALLE_REQUIRE(M >= 64 && N > 0 && K > 0 && K % 64 == 0, VALLE_E_UNSUPPORTED, "valle_gemm_normed: M=%d N=%d K=%d (M >= 64, K a multiple of 64)", M, N, K);
    VALLE_REQUIRE(epilogue == VALLE_EPI_STORE || (epilogue == VALLE_EPI_GEGLU && !rope), VALLE_E_ARG, "valle_gemm_normed: epilogue %d", epilogue);
    VALLE_REQUIRE(!rope == !pos, VALLE_E_ARG, "valle_gemm_normed: rope and pos come together");
    VALLE_REQUIRE(epilogue != VALLE_EPI_GEGLU || (N % 4 == 0 && ldc % 2 == 0), VALLE_E_ARG, "valle_gemm_normed: EPI_GEGLU needs N %% 4 == 0");
    VALLE_REQUIRE(ldc >= (epilogue == VALLE_EPI_GEGLU ? N / 2 : N), VALLE_E_ARG, "valle_gemm_normed: ldc %d too small", ldc);
    VALLE_REQUIRE(!rope || (N % 192 == 0 && ldc == N), VALLE_E_ARG, "valle_gemm_normed: the RoPE epilogue takes N = 3 d_model, d_model a multiple of 64");
    const bf16* X = (const bf16*)A; const bf16* W = (const bf16*)Wf;
    if (epilogue == VALLE_EPI_GEGLU && N >= 4096 && M >= 1024) {
        dim3 g256(ceil_div(N, 256), ceil_div(M, 256));
        const int pg = pp_persist_grid(g256.x * g256.y, K);
        if (pg) gemm_pp256_bf16_kernel<VALLE_EPI_GEGLU, true, 64, true><<<pg, 512, 0, s>>>(X, W, nullptr, nullptr, C, M, N, K, ldc, eps);
        else gemm_pp256_bf16_kernel<VALLE_EPI_GEGLU, true><<<g256, 512, 0, s>>>(X, W, nullptr, nullptr, C, M, N, K, ldc, eps);
    } else {
        dim3 grid(ceil_div(N, 128), ceil_div(M, 128));
        if (epilogue == VALLE_EPI_GEGLU) gemm_glds_bf16_kernel<VALLE_EPI_GEGLU, true><<<grid, 256, 0, s>>>(X, W, nullptr, nullptr, C, M, N, K, ldc, eps);
        else if (rope) {
            // 256 x 192 ping-pong tiles where they make (nearly) whole rounds of 256 workgroups (M = 12,176, d = 1,024:
            // 12 x 64 = 3.0 rounds, 95 vs 105 us without the epilogue); else the 128 x 128 kernel
            const float rounds = N % 256 == 0 ? (float)(N / 256) * (float)ceil_div(M, 192) / 256.f : 0.f;
            const float frac = rounds - (float)(int)rounds;
            // many rounds (48,704 rows: 11.9 of 192-row tiles, 8.95 of 256-row ones): the persistent 256-row tiles, priced as
            // in gemm_glds_dispatch (a 192-row tile costs 0.85 of a 256-row one; 476 -> 4xx us there)
            const int t256 = N % 256 == 0 ? (N / 256) * ceil_div(M, 256) : 0;
            static const int rope256 = [] { const char* e = getenv("VALLE_GEMM_ROPE256"); return e ? atoi(e) : 1; }();   // dev switch
            const int pg256 = (t256 && rope256) ? pp_persist_grid(t256, K) : 0;
            if (pg256 && (float)ceil_div(t256, pg256) < 0.85f * (float)ceil_div((N / 256) * ceil_div(M, 192), pg256)) {
                gemm_pp256_bf16_kernel<GEMM_EPI_ROPE, true, 64, true><<<pg256, 512, 0, s>>>(X, W, rope, (const bf16*)pos, C, M, N, K, ldc, eps);
            } else if (rounds >= 2.f && (frac == 0.f || frac > 0.7f)) {
                dim3 g192(N / 256, ceil_div(M, 192));
                const int pg = pp_persist_grid(g192.x * g192.y, K);
                if (pg) gemm_pp256_bf16_kernel<GEMM_EPI_ROPE, true, 48, true><<<pg, 512, 0, s>>>(X, W, rope, (const bf16*)pos, C, M, N, K, ldc, eps);
                else gemm_pp256_bf16_kernel<GEMM_EPI_ROPE, true, 48><<<g192, 512, 0, s>>>(X, W, rope, (const bf16*)pos, C, M, N, K, ldc, eps);
            } else {
                gemm_glds_bf16_kernel<GEMM_EPI_ROPE, true><<<grid, 256, 0, s>>>(X, W, rope, (const bf16*)pos, C, M, N, K, ldc, eps);
            }
        }
        else gemm_glds_bf16_kernel<VALLE_EPI_STORE, true><<<grid, 256, 0, s>>>(X, W, nullptr, nullptr, C, M, N, K, ldc, eps);
    }
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Few rows, bf16 (M up to a few hundred): the 128 x 128 tile kernels leave most of the chip idle there (N = 2,048 is 16
// workgroups).  One workgroup per 16 output columns instead; its four waves split K, every wave streams its share of the 16
// weight rows straight from the row-major matrix -- a lane's 16 bytes ARE its MFMA A fragment (row fr, k = fq * 8 ..) -- and
// of the activation rows likewise (B fragment), the four partial sums meet in LDS.  Weights are read once per 16 rows of M
// (from L2 behind the first tile).  Used where the caller asks for it (valle_gemm_skinny): the version >= 7 encoder /
// decoder linears of a decode step.
// Grouped form (blockIdx.y = group g): G independent products of one shape in one launch -- the eight per-level FFNs and
// heads of the version >= 7 AudioDecoder (base_v2.py:215-225); every operand of group g lies a fixed stride behind group 0's.
struct SkinnyGroups { size_t x, w, bias_bytes, resid, c_bytes; };   // strides between groups: elements (x, w, resid) or bytes
template <int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                               const void* __restrict__ bias, const bf16* resid, void* Cout,
                                                               int M, int N, int K, int ldc, SkinnyGroups gs) {
    {
        const size_t g = blockIdx.y;
        X += g * gs.x; W += g * gs.w;
        if (bias) bias = (const char*)bias + g * gs.bias_bytes;
        if (resid) resid += g * gs.resid;
        Cout = (char*)Cout + g * gs.c_bytes;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nrow = min(n0 + fr, N - 1);                       // clamped: rows past N are computed and never stored
    const int kw = K >> 2, kbase = wave * kw;                    // K % 128 == 0 (host check): a wave's share is whole 32-wide steps
    __shared__ float part[4][16][17];
    const bf16* wp = W + (size_t)nrow * K + kbase + fq * 8;
    // a wave's weight share stays in registers across the row tiles when it fits 16 steps (K <= 2,048)
    const bool held = kw <= 512;
    bf16x8 wreg[16];
    if (held) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u * 32 < kw) wreg[u] = *reinterpret_cast<const bf16x8*>(wp + u * 32);
    }
    // row tiles of 16: blockIdx.z takes every gridDim.z-th one (few output columns and many rows: the launch fills the chip)
    const int mt0 = blockIdx.z;
    for (int mt = mt0; mt * 16 < M; mt += gridDim.z) {
        const int mc = min(mt * 16 + fr, M - 1);
        const bf16* xp = X + (size_t)mc * K + kbase + fq * 8;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (held) {
#pragma unroll
            for (int u4 = 0; u4 < 16; u4 += 4) {
                if (u4 * 32 < kw) {                              // kw is a multiple of 32; whole groups of four while it is one of 128
                    bf16x8 b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) b[u] = (u4 + u) * 32 < kw ? *reinterpret_cast<const bf16x8*>(xp + (u4 + u) * 32) : bf16x8{};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if ((u4 + u) * 32 < kw) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[u4 + u], b[u], acc, 0, 0, 0);
                }
            }
        } else {
            int k = 0;
            for (; k + 128 <= kw; k += 128) {                    // four steps' loads in flight
                bf16x8 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = *reinterpret_cast<const bf16x8*>(wp + k + 32 * u);
                    b[u] = *reinterpret_cast<const bf16x8*>(xp + k + 32 * u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u], acc, 0, 0, 0);
            }
            for (; k < kw; k += 32) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(wp + k);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(xp + k);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            }
        }
        if (mt > mt0) __syncthreads();                           // the previous tile's partial sums have been consumed
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][fr][fq * 4 + r] = acc[r];      // acc[r] = (column n0 + fq * 4 + r, row mt * 16 + fr)
        __syncthreads();
        if (wave == 0) {
            f32x4 sum[1][1];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sum[0][0][r] = ((part[0][fr][fq * 4 + r] + part[1][fr][fq * 4 + r]) + part[2][fr][fq * 4 + r]) + part[3][fr][fq * 4 + r];
            gemm_epilogue<bf16, EPI, 1, 1>(sum, n0, mt * 16, fr, fq, bias, resid, Cout, M, N, ldc);
        }
    }
}

static int gemm_skinny_launch(const void* A, const void* W, const void* bias, const void* resid, void* C, int M, int N, int K,
                              int ldc, int epilogue, int G, const SkinnyGroups& gs, hipStream_t s) {
    // the row tiles are dealt to blockIdx.z while the launch has fewer workgroups than the chip has CUs (weights then come
    // from L2 for all but the first reader; measured on the version >= 7 encoder at 128 rows, see profiles/README.md)
    const int cols = ceil_div(N, 16) * G, tiles = ceil_div(M, 16);
    int mz = 256 / cols;
    mz = mz < 1 ? 1 : (mz > tiles ? tiles : mz);
    const dim3 grid(ceil_div(N, 16), G, mz);
#define SK_CASE(E) case E: gemm_skinny_bf16_kernel<E><<<grid, 256, 0, s>>>((const bf16*)A, (const bf16*)W, bias, (const bf16*)resid, C, M, N, K, ldc, gs); break;
    switch (epilogue) {
        SK_CASE(VALLE_EPI_STORE) SK_CASE(VALLE_EPI_RESID) SK_CASE(VALLE_EPI_F32OUT) SK_CASE(VALLE_EPI_GELU) SK_CASE(VALLE_EPI_RESID_BIAS)
        default: valle_set_error("valle_gemm_skinny: epilogue %d is not built", epilogue); return VALLE_E_ARG;
    }
#undef SK_CASE
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_gemm_skinny(const void* A, const void* W, const void* bias, const void* resid, void* C,
                                 int M, int N, int K, int ldc, int epilogue, int dtype, void* stream) {
    VALLE_REQUIRE(A && W && C, VALLE_E_ARG, "valle_gemm_skinny: null pointer");
    VALLE_REQUIRE(dtype == VALLE_BF16, VALLE_E_UNSUPPORTED, "valle_gemm_skinny: bf16 only (use valle_gemm)");
    VALLE_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 128 == 0, VALLE_E_ARG, "valle_gemm_skinny: M=%d N=%d K=%d (K must be a multiple of 128)", M, N, K);
    VALLE_REQUIRE((epilogue != VALLE_EPI_RESID && epilogue != VALLE_EPI_RESID_BIAS) || resid, VALLE_E_ARG, "valle_gemm_skinny: residual missing");
    VALLE_REQUIRE((epilogue != VALLE_EPI_GELU && epilogue != VALLE_EPI_RESID_BIAS) || bias, VALLE_E_ARG, "valle_gemm_skinny: EPI_GELU / EPI_RESID_BIAS need a bias");
    VALLE_REQUIRE(ldc >= N, VALLE_E_ARG, "valle_gemm_skinny: ldc %d too small", ldc);
    if (M == 0) return VALLE_OK;
    return gemm_skinny_launch(A, W, bias, resid, C, M, N, K, ldc, epilogue, 1, SkinnyGroups{}, (hipStream_t)stream);
}

extern "C" int valle_gemm_skinny_grouped(const void* A, size_t a_group_stride, const void* W, size_t w_group_stride,
                                         const void* bias, size_t bias_group_stride, const void* resid, size_t resid_group_stride,
                                         void* C, size_t c_group_stride, int G, int M, int N, int K, int ldc, int epilogue,
                                         int dtype, void* stream) {
    VALLE_REQUIRE(A && W && C, VALLE_E_ARG, "valle_gemm_skinny_grouped: null pointer");
    VALLE_REQUIRE(dtype == VALLE_BF16, VALLE_E_UNSUPPORTED, "valle_gemm_skinny_grouped: bf16 only");
    VALLE_REQUIRE(G >= 1 && G <= 65535, VALLE_E_ARG, "valle_gemm_skinny_grouped: G=%d", G);
    VALLE_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 128 == 0, VALLE_E_ARG, "valle_gemm_skinny_grouped: M=%d N=%d K=%d (K must be a multiple of 128)", M, N, K);
    VALLE_REQUIRE((epilogue != VALLE_EPI_RESID && epilogue != VALLE_EPI_RESID_BIAS) || resid, VALLE_E_ARG, "valle_gemm_skinny_grouped: residual missing");
    VALLE_REQUIRE((epilogue != VALLE_EPI_GELU && epilogue != VALLE_EPI_RESID_BIAS) || bias, VALLE_E_ARG, "valle_gemm_skinny_grouped: EPI_GELU / EPI_RESID_BIAS need a bias");
    VALLE_REQUIRE(ldc >= N, VALLE_E_ARG, "valle_gemm_skinny_grouped: ldc %d too small", ldc);
    // the fragment loads are 16-byte vectors: every group's operands must keep that alignment
    VALLE_REQUIRE(a_group_stride % 8 == 0 && w_group_stride % 8 == 0 && resid_group_stride % 4 == 0 && c_group_stride % 4 == 0, VALLE_E_ARG,
                  "valle_gemm_skinny_grouped: group strides must keep 16-byte (A, W) / 8-byte (resid, C) alignment");
    if (M == 0) return VALLE_OK;
    SkinnyGroups gs{};
    gs.x = a_group_stride; gs.w = w_group_stride; gs.resid = resid_group_stride;
    gs.bias_bytes = bias_group_stride * (epilogue == VALLE_EPI_F32OUT ? sizeof(float) : sizeof(bf16));
    gs.c_bytes = c_group_stride * (epilogue == VALLE_EPI_F32OUT ? sizeof(float) : sizeof(bf16));
    return gemm_skinny_launch(A, W, bias, resid, C, M, N, K, ldc, epilogue, G, gs, (hipStream_t)stream);
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
        case VALLE_EPI_GELU:   gemm_nt_kernel<T, VALLE_EPI_GELU><<<grid, 256, 0, s>>>((const T*)A, (const T*)W, bias, (const T*)resid, C, M, N, K, ldc); break;
        case VALLE_EPI_RESID_BIAS: gemm_nt_kernel<T, VALLE_EPI_RESID_BIAS><<<grid, 256, 0, s>>>((const T*)A, (const T*)W, bias, (const T*)resid, C, M, N, K, ldc); break;
        default: valle_set_error("valle_gemm: unknown epilogue %d", epi); return VALLE_E_ARG;
    }
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_gemm(const void* A, const void* W, const void* bias, const void* resid, void* C,
                          int M, int N, int K, int ldc, int epilogue, int dtype, void* stream) {
    VALLE_REQUIRE(A && W && C, VALLE_E_ARG, "valle_gemm: null pointer");
    VALLE_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 32 == 0, VALLE_E_ARG, "valle_gemm: M=%d N=%d K=%d (K must be a multiple of 32)", M, N, K);
    VALLE_REQUIRE((epilogue != VALLE_EPI_RESID && epilogue != VALLE_EPI_RESID_BIAS) || resid, VALLE_E_ARG, "valle_gemm: EPI_RESID needs resid");
    VALLE_REQUIRE((epilogue != VALLE_EPI_GELU && epilogue != VALLE_EPI_RESID_BIAS) || bias, VALLE_E_ARG, "valle_gemm: EPI_GELU / EPI_RESID_BIAS need a bias");
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
