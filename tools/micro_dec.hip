// Dev micro-benchmark (not part of the library): time the decode linears and a pure streaming
// read over a weight set larger than the Infinity Cache, so every launch reads HBM-cold data.
#include "../vall-e_amd/csrc/common.h"
#include <vector>
#include <stdio.h>

template <int BYTES_PER_BLOCK>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ p, uint4* out) {
    const uint4* base = p + (size_t)blockIdx.x * (BYTES_PER_BLOCK / 16);
    uint4 acc = make_uint4(0, 0, 0, 0);
    constexpr int N = BYTES_PER_BLOCK / 16 / 256;
    uint4 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = base[threadIdx.x + 256 * i];
#pragma unroll
    for (int i = 0; i < N; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[blockIdx.x] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const int d = 1024, f = 4096, B = 16, L = 24;   // 24 "layers" of weights = 805 MB > 256 MiB MALL
    std::vector<bf16*> wqkv(L), wo(L), wgu(L), wdn(L);
    for (int l = 0; l < L; ++l) {
        CK(hipMalloc(&wqkv[l], (size_t)3 * d * d * 2)); CK(hipMemset(wqkv[l], 0x11, (size_t)3 * d * d * 2));
        CK(hipMalloc(&wo[l], (size_t)d * d * 2));       CK(hipMemset(wo[l], 0x11, (size_t)d * d * 2));
        CK(hipMalloc(&wgu[l], (size_t)2 * f * d * 2));  CK(hipMemset(wgu[l], 0x11, (size_t)2 * f * d * 2));
        CK(hipMalloc(&wdn[l], (size_t)d * f * 2));      CK(hipMemset(wdn[l], 0x11, (size_t)d * f * 2));
    }
    bf16 *x, *h, *ff, *nw; uint4* sink;
    CK(hipMalloc(&x, B * d * 2)); CK(hipMalloc(&h, B * d * 2)); CK(hipMalloc(&ff, B * f * 2)); CK(hipMalloc(&nw, d * 2));
    CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemset(x, 0x3c, B * d * 2)); CK(hipMemset(ff, 0x3c, B * f * 2)); CK(hipMemset(nw, 0x3c, d * 2));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    auto timeit = [&](const char* name, double mb, auto fn) {
        for (int l = 0; l < L; ++l) fn(l);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) for (int l = 0; l < L; ++l) fn(l);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (reps * L);
        printf("%-34s %7.2f us/launch  %6.2f TB/s\n", name, us, mb / us);
    };
    timeit("stream 16.8MB (512 blk x 32KB)", 16.78, [&](int l) { stream_kernel<32768><<<512, 256, 0, s>>>((const uint4*)wgu[l], sink); });
    timeit("stream 16.8MB (1024 blk x 16KB)", 16.78, [&](int l) { stream_kernel<16384><<<1024, 256, 0, s>>>((const uint4*)wgu[l], sink); });
    timeit("stream 16.8MB (256 blk x 64KB)", 16.78, [&](int l) { stream_kernel<65536><<<256, 256, 0, s>>>((const uint4*)wgu[l], sink); });
    timeit("stream 8.4MB (256 blk x 32KB)", 8.39, [&](int l) { stream_kernel<32768><<<256, 256, 0, s>>>((const uint4*)wdn[l], sink); });
    timeit("stream 6.3MB (192 blk x 32KB)", 6.29, [&](int l) { stream_kernel<32768><<<192, 256, 0, s>>>((const uint4*)wqkv[l], sink); });
    timeit("stream 2MB (64 blk x 32KB)", 2.10, [&](int l) { stream_kernel<32768><<<64, 256, 0, s>>>((const uint4*)wo[l], sink); });
    timeit("stream 8.4MB (64 blk x 128KB)", 8.39, [&](int l) { stream_kernel<131072><<<64, 256, 0, s>>>((const uint4*)wdn[l], sink); });
    timeit("stream 2MB (256 blk x 8KB)", 2.10, [&](int l) { stream_kernel<8192><<<256, 256, 0, s>>>((const uint4*)wo[l], sink); });
    auto mk = [&](const bf16* xx, const bf16* norm, const bf16* W, const bf16* resid, void* out, int ldo, int N, int K) {
        DecArgs a{}; a.x = xx; a.norm_w = norm; a.eps = 1e-6f; a.W = W; a.resid = resid; a.out = out; a.ldo = ldo; a.B = B; a.N = N; a.K = K; return a; };
    timeit("dec gate/up GEGLU (16.8MB)", 16.78, [&](int l) { valle_dec_linear_bf16(mk(x, nw, wgu[l], nullptr, ff, f, 2 * f, d), VALLE_EPI_GEGLU, s); });
    timeit("dec down RESID (8.4MB)", 8.39, [&](int l) { valle_dec_linear_bf16(mk(ff, nullptr, wdn[l], x, x, d, d, f), VALLE_EPI_RESID, s); });
    timeit("dec o_proj RESID (2MB)", 2.10, [&](int l) { valle_dec_linear_bf16(mk(h, nullptr, wo[l], x, x, d, d, d), VALLE_EPI_RESID, s); });
    {   // decode attention over a KV cache set larger than the Infinity Cache
        const int H = 16, CT = 768, NL = 6;
        const size_t per = (size_t)B * H * CT * 64;
        bf16 *kc, *vc, *qkvb, *att; int32_t* ctx;
        CK(hipMalloc(&kc, per * NL * 2)); CK(hipMalloc(&vc, per * NL * 2)); CK(hipMemset(kc, 0x3c, per * NL * 2)); CK(hipMemset(vc, 0x3c, per * NL * 2));
        CK(hipMalloc(&qkvb, B * 3 * d * 2)); CK(hipMemset(qkvb, 0x3c, B * 3 * d * 2)); CK(hipMalloc(&att, B * d * 2));
        CK(hipMalloc(&ctx, B * 4));
        for (int cl : {311, 536, 760}) {
            std::vector<int32_t> hc(B, cl); CK(hipMemcpy(ctx, hc.data(), B * 4, hipMemcpyHostToDevice));
            char nm[64]; snprintf(nm, sizeof nm, "attn decode ctx=%d", cl);
            timeit(nm, (double)B * H * cl * 256 / 1e6, [&](int l) { valle_attn_decode_launch(qkvb, kc + per * (l % NL), vc + per * (l % NL), ctx, nullptr, att, B, H, 64, CT, 0, VALLE_BF16, s); });
        }
    }
    printf("last error: %s\n", valle_last_error());
    return 0;
}
