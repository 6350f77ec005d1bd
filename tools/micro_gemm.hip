// Dev micro-benchmark (not part of the library): the dense-pass GEMM shapes of one NAR level
// (B=16 x T=761 = 12,176 rows) as valle_forward_dense launches them in bf16: q/k/v (+ RoPE) and gate/up through
// valle_gemm_normed (RMSNorm folded in), o_proj and down_proj through valle_gemm; random operands.
// MICRO_GEMM_UNFUSED=1: the q/k/v and gate/up GEMMs without the folded norm (valle_gemm_qkv_rope / valle_gemm).
#include "../vall-e_amd/csrc/common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void fill_rand(uint16_t* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (uint16_t)(((h & 1u) << 15) | ((118u + ((h >> 1) & 7u)) << 7) | ((h >> 8) & 0x7fu));
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 12176;
    struct Shape { const char* name; int N, K, epi; } shapes[] = {
        {"qkv   N=3072 K=1024 store", 3072, 1024, VALLE_EPI_STORE}, {"o     N=1024 K=1024 resid", 1024, 1024, VALLE_EPI_RESID},
        {"gu    N=8192 K=1024 geglu", 8192, 1024, VALLE_EPI_GEGLU}, {"down  N=1024 K=4096 resid", 1024, 4096, VALLE_EPI_RESID},
        // the gate/up shape through the same 256 x 256 kernel without norm and GEGLU (what the epilogue costs); not in the total
        {"(gu shape, plain store)", 8192, 1024, -1}, {"(gu shape, GEGLU, no norm)", 8192, 1024, -2}};
    bf16 *A, *W, *C, *R;
    CK(hipMalloc(&A, (size_t)M * 4096 * 2)); CK(hipMalloc(&W, (size_t)8192 * 4096 * 2));
    CK(hipMalloc(&C, (size_t)M * 8192 * 2)); CK(hipMalloc(&R, (size_t)M * 8192 * 2));
    fill_rand<<<1024, 256>>>((uint16_t*)A, (size_t)M * 4096, 1); fill_rand<<<1024, 256>>>((uint16_t*)W, (size_t)8192 * 4096, 2);
    fill_rand<<<1024, 256>>>((uint16_t*)R, (size_t)M * 8192, 3);
    CK(hipDeviceSynchronize());
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot = 0;
    float* rope; int32_t* pos;
    CK(hipMalloc(&rope, 1024 * 64 * 4)); CK(hipMemset(rope, 0, 1024 * 64 * 4));
    CK(hipMalloc(&pos, (size_t)M * 4)); CK(hipMemset(pos, 0, (size_t)M * 4));
    const bool unfused = getenv("MICRO_GEMM_UNFUSED") != nullptr;
    const char* only = getenv("MICRO_GEMM_ONLY");          // "qkv" | "o" | "gu" | "down": that GEMM alone (PMC passes per shape)
    for (auto& sh : shapes) {
        if (only && (sh.epi < 0 || strncmp(sh.name, only, strlen(only)) != 0 || sh.name[strlen(only)] != ' ')) continue;
        const int ldc = (sh.epi == VALLE_EPI_GEGLU || sh.epi == -2) ? sh.N / 2 : sh.N;
        auto run = [&]() -> int {
            if (sh.epi == -1) return valle_gemm(A, W, nullptr, nullptr, C, M, sh.N, sh.K, ldc, VALLE_EPI_STORE, VALLE_BF16, s);
            if (sh.epi == -2) return valle_gemm(A, W, nullptr, nullptr, C, M, sh.N, sh.K, ldc, VALLE_EPI_GEGLU, VALLE_BF16, s);
            if (sh.epi == VALLE_EPI_STORE)
                return unfused ? valle_gemm_qkv_rope(A, W, rope, pos, C, M, sh.N / 3, sh.K, s)
                               : valle_gemm_normed(A, W, C, M, sh.N, sh.K, ldc, VALLE_EPI_STORE, 1e-6f, rope, pos, s);
            if (sh.epi == VALLE_EPI_GEGLU && !unfused) return valle_gemm_normed(A, W, C, M, sh.N, sh.K, ldc, VALLE_EPI_GEGLU, 1e-6f, nullptr, nullptr, s);
            return valle_gemm(A, W, nullptr, R, C, M, sh.N, sh.K, ldc, sh.epi, VALLE_BF16, s);
        };
        for (int i = 0; i < 3; ++i)
            if (run()) { printf("error: %s\n", valle_last_error()); return 1; }
        CK(hipStreamSynchronize(s));
        const int reps = 20;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tf = 2.0 * M * sh.N * sh.K / us * 1e-6;
        printf("%-28s %8.1f us  %7.1f TFLOP/s\n", sh.name, us, tf);
        if (sh.epi >= 0) tot += us;
    }
    printf("layer GEMMs total %.1f us\n", tot);
    return 0;
}
