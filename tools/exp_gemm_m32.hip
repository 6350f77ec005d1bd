// Dev experiment (not part of the library): the 256 x 256 ping-pong GEMM main loop of csrc/gemm.hip
// (gemm_pp256_bf16_kernel, plain store, no norm) with v_mfma_f32_32x32x16_bf16 in place of 16x16x32 -- half the MFMA
// issues and half the operand-register reads per FLOP, the same LDS reads (24 ds_read_b128 per wave and K-tile), the
// same slots and barriers.  Question: do the dense GEMMs, which run with the clock held down, gain from it?
// Build (from vall-e_amd/csrc, after `make`):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize ../../tools/exp_gemm_m32.hip build/*.o -o ../../tools/exp_gemm_m32
// (build/*.o without the micro_* objects; see `make exp_m32`)
#include "../vall-e_amd/csrc/common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

__global__ void fill_rand(uint16_t* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (uint16_t)(((h & 1u) << 15) | ((118u + ((h >> 1) & 7u)) << 7) | ((h >> 8) & 0x7fu));
    }
}

// Every wave owns 128 (n) x 64 (m) of the 256 x 256 tile = four 64 x 32 quadrants of two 32 x 32 MFMA tiles each.
// LDS image as in the library: rows of 128 B (one K-tile of 64), 16-B chunk c of row r stored at chunk c ^ (r & 7).
// A 32x32x16 operand lane holds row (lane % 32), k = 16 * kk + 8 * (lane / 32) .. +7  ->  chunk 2 * kk + lane / 32.
__global__ __launch_bounds__(512) void pp256_m32_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W, bf16* C,
                                                        int M, int N, int K, int ldc, int flags) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 3;
    const int nh = wi & 1, mq = (wi >> 1) + 2 * grp;
    const int l32 = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 256, m0 = blockIdx.y * 256;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bf16* src[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int piece = wave * 8 + j;
        const int row = piece * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (row & 7);
        if (piece < 32) { int n = n0 + row; n = n < N ? n : N - 1; src[j] = W + (size_t)n * K + ch * 8; }
        else { int m = m0 + row - 256; m = m < M ? m : M - 1; src[j] = X + (size_t)m * K + ch * 8; }
    }
    auto stage_part = [&](int kt, int buf, int j0, int j1) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j >= j0 && j < j1)
                __builtin_amdgcn_global_load_lds(src[j] + kt * 64, (lds_void_t*)&lds[buf * 65536 + (wave * 8 + j) * 1024], 16, 0, 0);
    };
    int sw[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) sw[kk] = ((kk * 2 + lh) ^ (l32 & 7)) << 4;
    const int abase = (nh * 128 + l32) * 128, bbase = (256 + mq * 64 + l32) * 128;
    // slots go by K halves, not by m halves: an MFMA slot = one 64-row n half x both 32-row m halves x two k steps =
    // 8 MFMAs on FOUR independent accumulators (a 32x32x16 MFMA is 16 passes long: with two accumulators in flight the next
    // dependent one waits -- first form of this experiment, 320-364 us on the gate/up shape)
    bf16x8 A[2][4], Bf[2][4];
    auto read_a = [&](const char* L, int i, int k0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) A[rt][k0 + kk] = *reinterpret_cast<const bf16x8*>(L + abase + (i * 64 + rt * 32) * 128 + sw[k0 + kk]);
    };
    auto read_b = [&](const char* L, int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) Bf[j][k0 + kk] = *reinterpret_cast<const bf16x8*>(L + bbase + (j * 32) * 128 + sw[k0 + kk]);
    };
    auto quad = [&](int i, int k0) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + rt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[rt][k0 + kk], Bf[j][k0 + kk], acc[i * 2 + rt][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define PP_END_READ()  do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define PP_END_MFMA()  __builtin_amdgcn_s_barrier()
    const int nt = K / 64;
    stage_part(0, 0, 0, 8);
    if (nt > 1) stage_part(1, 1, 0, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nt; ++kt) {
        const char* L = lds + (kt & 1) * 65536;
        const bool st = kt >= 1 && kt + 1 < nt;
        read_a(L, 0, 0); read_b(L, 0);                                     // R1
        if (st) stage_part(kt + 1, (kt + 1) & 1, 0, 2);
        PP_END_READ();
        quad(0, 0);                                                        // M1
        PP_END_MFMA();
        read_a(L, 0, 2); read_b(L, 2);                                     // R2
        if (st) stage_part(kt + 1, (kt + 1) & 1, 2, 5);
        PP_END_READ();
        quad(0, 2);                                                        // M2
        PP_END_MFMA();
        read_a(L, 1, 0);                                                   // R3
        if (st) stage_part(kt + 1, (kt + 1) & 1, 5, 8);
        PP_END_READ();
        quad(1, 0);                                                        // M3
        PP_END_MFMA();
        read_a(L, 1, 2);                                                   // R4 (+ tile kt+1 has landed: my pieces)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_END_READ();
        quad(1, 2);                                                        // M4
        PP_END_MFMA();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    if (flags & 1) {            // main loop only: no stores (every accumulator still feeds the condition)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) C[0] = (bf16)t;
        return;
    }
    // accumulator register 4 q + r of a 32 x 32 tile = (n = 8 q + 4 (lane / 32) + r, m = lane % 32)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + mq * 64 + j * 32 + l32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nb = n0 + nh * 128 + t * 32 + q * 8 + lh * 4;
                if (m >= M || nb >= N) continue;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[t][j][q * 4 + r];
                *reinterpret_cast<bf16x4*>(C + (size_t)m * ldc + nb) = o;
            }
        }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 12176;
    struct Shape { const char* name; int N, K; } shapes[] = {{"gu shape   N=8192 K=1024", 8192, 1024}, {"qkv shape  N=3072 K=1024", 3072, 1024},
                                                              {"down shape N=1024 K=4096", 1024, 4096}, {"o shape    N=1024 K=1024", 1024, 1024}};
    bf16 *A, *W, *C, *C2;
    CK(hipMalloc(&A, (size_t)M * 4096 * 2)); CK(hipMalloc(&W, (size_t)8192 * 4096 * 2));
    CK(hipMalloc(&C, (size_t)M * 8192 * 2)); CK(hipMalloc(&C2, (size_t)M * 8192 * 2));
    fill_rand<<<1024, 256>>>((uint16_t*)A, (size_t)M * 4096, 1); fill_rand<<<1024, 256>>>((uint16_t*)W, (size_t)8192 * 4096, 2);
    CK(hipDeviceSynchronize());
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        const dim3 grid((sh.N + 255) / 256, (M + 255) / 256);
        int flags = 0;
        auto mine = [&]() { pp256_m32_kernel<<<grid, 512, 0, s>>>(A, W, C, M, sh.N, sh.K, sh.N, flags); return 0; };
        auto lib = [&]() { return valle_gemm(A, W, nullptr, nullptr, C2, M, sh.N, sh.K, sh.N, VALLE_EPI_STORE, VALLE_BF16, s); };
        CK(hipMemsetAsync(C, 0xff, (size_t)M * sh.N * 2, s));
        mine(); if (lib()) { printf("error: %s\n", valle_last_error()); return 1; }
        CK(hipStreamSynchronize(s)); CK(hipGetLastError());
        std::vector<uint16_t> h1((size_t)M * sh.N), h2((size_t)M * sh.N);
        CK(hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), C2, h2.size() * 2, hipMemcpyDeviceToHost));
        auto tof = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
        double num = 0, den = 0; size_t differ = 0;
        for (size_t i = 0; i < h1.size(); ++i) { const double a = tof(h1[i]), b = tof(h2[i]); num += (a - b) * (a - b); den += b * b; differ += h1[i] != h2[i]; }
        printf("%-26s rel-L2 against the library kernel %.3e, %zu of %zu values differ in the last bit or more\n", sh.name, std::sqrt(num / den), differ, h1.size());
        for (int round = 0; round < 2; ++round)
            for (int which = 0; which < 3; ++which) {
                flags = which == 2 ? 1 : 0;
                const int reps = 20;
                for (int i = 0; i < 3; ++i) { if (which == 1) lib(); else mine(); }
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) { if (which == 1) lib(); else mine(); }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps;
                printf("  %-34s %8.1f us  %7.1f TFLOP/s\n", which == 1 ? "library (16x16x32, its own tiling)" : which == 2 ? "32x32x16, main loop only (no stores)" : "32x32x16, 256 x 256 tiles", us, 2.0 * M * sh.N * sh.K / us * 1e-6);
            }
    }
    return 0;
}
