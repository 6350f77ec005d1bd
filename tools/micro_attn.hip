// Dev micro-benchmark (not part of the library): the dense-pass flash attention alone, compiled together with
// attn_flash.hip so that its compile-time knobs (-DFA_ROW=...) can be swept:
//   hipcc ... -DFA_ROW=80 tools/micro_attn.hip <library objects without attn_flash.o> -o tools/micro_attn_80
#include "../vall-e_amd/csrc/attn_flash.hip"
#include <vector>
#include <stdlib.h>

__global__ void fill_rand(uint16_t* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (uint16_t)(((h & 1u) << 15) | ((118u + ((h >> 1) & 7u)) << 7) | ((h >> 8) & 0x7fu));
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 761, B = argc > 2 ? atoi(argv[2]) : 16, H = 16, d = H * 64;
    const int causal = argc > 3 ? atoi(argv[3]) : 1;
    bf16 *qkv, *out; int32_t* cu;
    CK(hipMalloc(&qkv, (size_t)B * T * 3 * d * 2)); CK(hipMalloc(&out, (size_t)B * T * d * 2)); CK(hipMalloc(&cu, (B + 1) * 4));
    fill_rand<<<1024, 256>>>((uint16_t*)qkv, (size_t)B * T * 3 * d, 7);
    std::vector<int32_t> h(B + 1); for (int i = 0; i <= B; ++i) h[i] = i * T;
    CK(hipMemcpy(cu, h.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (valle_attn_flash_bf16(qkv, out, cu, B, T, H, causal, s)) { printf("error %s\n", valle_last_error()); return 1; }
    CK(hipStreamSynchronize(s));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) valle_attn_flash_bf16(qkv, out, cu, B, T, H, causal, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("FA_ROW=%d %s attention B=%d T=%d: %.1f us, %.1f TFLOP/s\n", FA_ROW, causal ? "causal" : "full", B, T, us, (double)B * H * 4.0 * T * T * 64 / (causal ? 2 : 1) / us * 1e-6);
    return 0;
}
