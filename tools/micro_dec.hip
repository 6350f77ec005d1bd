// Dev micro-benchmark (not part of the library): the AR decode layer chain
// (QKV -> decode attention -> o_proj -> gate/up -> down) on weights / KV caches larger than the
// Infinity Cache, eager and as a hipGraph, plus each kernel alone in a dependent chain and the
// pure-streaming floor for the same byte counts.  Per-kernel durations: run under
//   rocprofv3 --kernel-trace --stats -- ./tools/micro_dec
#include "../vall-e_amd/csrc/common.h"
#include <vector>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

template <int BYTES_PER_BLOCK, int NT>
__global__ __launch_bounds__(NT) void stream2_kernel(const uint4* __restrict__ p, uint4* out, size_t block_stride16) {
    const uint4* base = p + (size_t)blockIdx.x * block_stride16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    constexpr int N = BYTES_PER_BLOCK / 16 / NT;
    uint4 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = base[threadIdx.x + NT * i];
#pragma unroll
    for (int i = 0; i < N; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[blockIdx.x] = acc;
}

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

// cache warmer: block (r, which) reads the first `bytes` of region r of the K (which = 0) or V cache with plain loads
__global__ __launch_bounds__(256) void kv_prefetch_kernel(const uint4* __restrict__ k, const uint4* __restrict__ v, size_t region16, int n16, uint4* out) {
    const uint4* base = (blockIdx.y ? v : k) + (size_t)blockIdx.x * region16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < n16; i += 256 * 4) {
        uint4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (i + j * 256 < n16) ? base[i + j * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc.x ^= t[j].x; acc.y ^= t[j].y; acc.z ^= t[j].z; acc.w ^= t[j].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[blockIdx.x] = acc;
}

// pseudo-random bf16 in (-2, 2) with random mantissas: realistic bit toggling (memset patterns run faster)
__global__ void fill_rand(uint16_t* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (uint16_t)(((h & 1u) << 15) | ((118u + ((h >> 1) & 7u)) << 7) | ((h >> 8) & 0x7fu));
    }
}
#define FILL(ptr, bytes, seed) fill_rand<<<1024, 256>>>((uint16_t*)(ptr), (size_t)(bytes) / 2, (seed))

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_ != 0) { printf("valle error %d: %s (line %d)\n", rc_, valle_last_error(), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int d = 1024, f = 4096, H = 16, L = 24;   // 24 layers of weights = 805 MB > 256 MiB MALL
    const int B = argc > 1 ? atoi(argv[1]) : 16;
    const int ctx = argc > 2 ? atoi(argv[2]) : 536;
    const int CT = (ctx + 1 <= 768 ? 768 : ((ctx + 1 + 63) / 64) * 64) + (getenv("MICRO_CT_PAD") ? atoi(getenv("MICRO_CT_PAD")) : 0);     // cache rows per (utterance, head): holds ctx + the new key
    if (B < 1 || B > 32 || ctx < 1 || ctx > 8192) { printf("B in 1..32, ctx in 1..8192\n"); return 1; }
    (void)(argc > 3 ? atoi(argv[3]) : 0);                       // (argv[3] kept for old command lines)
    const bool force_eager = argc > 4 && atoi(argv[4]) != 0;   // counter runs: one dispatch record per kernel
    std::vector<bf16*> wqkv(L), wo(L), wgu(L), wdn(L), kc(L), vc(L);
    const size_t per = (size_t)B * H * CT * 64;
    for (int l = 0; l < L; ++l) {
        CK(hipMalloc(&wqkv[l], (size_t)3 * d * d * 2)); FILL(wqkv[l], (size_t)3 * d * d * 2, 11 + l);
        CK(hipMalloc(&wo[l], (size_t)d * d * 2));       FILL(wo[l], (size_t)d * d * 2, 111 + l);
        CK(hipMalloc(&wgu[l], (size_t)2 * f * d * 2));  FILL(wgu[l], (size_t)2 * f * d * 2, 211 + l);
        CK(hipMalloc(&wdn[l], (size_t)d * f * 2));      FILL(wdn[l], (size_t)d * f * 2, 311 + l);
        CK(hipMalloc(&kc[l], per * 2)); FILL(kc[l], per * 2, 411 + l);
        CK(hipMalloc(&vc[l], per * 2)); FILL(vc[l], per * 2, 511 + l);
    }
    bf16 *x, *att, *ff, *nw, *qkv; uint4* sink; float* rope; int32_t *pos, *ctxl;
    CK(hipMalloc(&x, 32 * d * 2)); CK(hipMalloc(&att, 32 * d * 2)); CK(hipMalloc(&ff, 32 * f * 2)); CK(hipMalloc(&nw, d * 2));
    CK(hipMalloc(&qkv, B * 3 * d * 2)); CK(hipMalloc(&rope, 1024 * 64 * 4)); CK(hipMalloc(&pos, B * 4)); CK(hipMalloc(&ctxl, B * 4));
    CK(hipMalloc(&sink, 1 << 20));
    float* ssq; CK(hipMalloc(&ssq, 32 * 256 * 4)); CK(hipMemset(ssq, 0, 32 * 256 * 4));
    FILL(x, 32 * d * 2, 1); FILL(att, 32 * d * 2, 2); FILL(ff, 32 * f * 2, 3); FILL(nw, d * 2, 4);
    FILL(qkv, B * 3 * d * 2, 5); CK(hipDeviceSynchronize()); CK(hipMemset(rope, 0, 1024 * 64 * 4)); CK(hipMemset(pos, 0, B * 4));
    { std::vector<int32_t> hc(B, ctx); CK(hipMemcpy(ctxl, hc.data(), B * 4, hipMemcpyHostToDevice)); }
    // MICRO_STOPPED=n: the first n utterances count as finished (early exit): their attention workgroups leave at once
    if (getenv("MICRO_STOPPED")) { std::vector<int32_t> hc(B, ctx); for (int i = 0; i < atoi(getenv("MICRO_STOPPED")) && i < B; ++i) hc[i] = ~ctx;
        CK(hipMemcpy(ctxl, hc.data(), B * 4, hipMemcpyHostToDevice)); }
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = force_eager ? 2 : 20;

    auto mkw = [&](const bf16* xx, int norm, const bf16* W, const bf16* resid, void* out, int ldo, int N, int K) {
        DecWArgs a{}; a.x = xx; a.norm = norm; a.eps = 1e-6f; a.W = W; a.resid = resid; a.out = out; a.ldo = ldo; a.B = B; a.N = N; a.K = K; return a; };
    auto w_qkv = [&](int l) { DecWArgs a = mkw(x, 1, wqkv[l], nullptr, qkv, 3 * d, 3 * d, d);
        a.rope_m = rope; a.ctx_len = ctxl; a.slot = nullptr; a.kc = kc[l]; a.vc = vc[l]; a.cache_T = CT; a.H = H; a.d = d;
        return valle_dec_wave_linear(a, DEC_EPI_QKV, s); };
    auto w_o = [&](int l) { return valle_dec_wave_linear(mkw(att, 0, wo[l], x, x, d, d, d), VALLE_EPI_RESID, s); };
    auto w_gu = [&](int l) { return valle_dec_wave_linear(mkw(x, 1, wgu[l], nullptr, ff, f, 2 * f, d), VALLE_EPI_GEGLU, s); };
    auto w_dn = [&](int l) { return valle_dec_wave_linear(mkw(ff, 0, wdn[l], x, x, d, d, f), VALLE_EPI_RESID, s); };
    auto w_attn = [&](int l) { return valle_attn_decode_bf16(qkv, 3 * d, kc[l], vc[l], ctxl, nullptr, att, B, H, CT, 1, 1, s); };
    // MICRO_ALIAS=n: every layer uses the buffers of layer (l % n) -- with n = 1 or 2 the whole working set stays in the
    // Infinity Cache: the time a perfect prefetcher could buy
    if (getenv("MICRO_ALIAS")) { const int n = atoi(getenv("MICRO_ALIAS")); for (int l = 0; l < L; ++l) { wqkv[l] = wqkv[l % n]; wo[l] = wo[l % n]; wgu[l] = wgu[l % n]; wdn[l] = wdn[l % n]; if (!getenv("MICRO_ALIAS_W")) { kc[l] = kc[l % n]; vc[l] = vc[l % n]; } } }   // MICRO_ALIAS_W: weights only
    auto wlayer = [&](int l) { int rc = w_qkv(l); rc |= w_attn(l); rc |= w_o(l); rc |= w_gu(l); rc |= w_dn(l); return rc; };

    // times `fn(l)` for l in 0..L-1 as one dependent chain; graph = captured once, replayed
    auto timeit = [&](const char* name, double mb, bool graph, auto fn) -> int {
        float ms;
        if (force_eager) graph = false;
        for (int l = 0; l < L; ++l) RC(fn(l));
        CK(hipStreamSynchronize(s));
        if (!graph) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) for (int l = 0; l < L; ++l) fn(l);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
        } else {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < L; ++l) fn(l);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * L);
        printf("%-40s %-6s %7.2f us  %6.2f TB/s\n", name, graph ? "graph" : "eager", us, mb / us);
        return 0;
    };
    const double mb_qkv = 6.29, mb_o = 2.10, mb_gu = 16.78, mb_dn = 8.39, mb_att = (double)B * H * (ctx + 1) * 256 / 1e6;
    printf("B=%d ctx=%d\n", B, ctx);
    if (timeit("stream 16.8MB (256 blk x 64KB)", 16.78, true, [&](int l) { stream_kernel<65536><<<256, 256, 0, s>>>((const uint4*)wgu[l], sink); return 0; })) return 1;
    if (timeit("stream 8.4MB (256 blk x 32KB)", 8.39, true, [&](int l) { stream_kernel<32768><<<256, 256, 0, s>>>((const uint4*)wdn[l], sink); return 0; })) return 1;
    if (timeit("stream 6.3MB (192 blk x 32KB)", 6.29, true, [&](int l) { stream_kernel<32768><<<192, 256, 0, s>>>((const uint4*)wqkv[l], sink); return 0; })) return 1;
    if (timeit("stream 2MB (64 blk x 32KB)", 2.10, true, [&](int l) { stream_kernel<32768><<<64, 256, 0, s>>>((const uint4*)wo[l], sink); return 0; })) return 1;
    // KV-shaped streams over ONE layer's K cache (B*H regions of CT*128 B; the buffer holds exactly B*H*CT*128 B)
    if (B == 16 && CT == 768) {
        if (timeit("stream 25MB (256 blk x 96KB, 512thr)", 25.2, true, [&](int l) { stream2_kernel<98304, 512><<<256, 512, 0, s>>>((const uint4*)kc[l], sink, (size_t)98304 / 16); return 0; })) return 1;
        if (timeit("stream 25MB (256 blk x 96KB, 256thr)", 25.2, true, [&](int l) { stream2_kernel<98304, 256><<<256, 256, 0, s>>>((const uint4*)kc[l], sink, (size_t)98304 / 16); return 0; })) return 1;
        if (timeit("stream 25MB (512 blk x 48KB, 256thr)", 25.2, true, [&](int l) { stream2_kernel<49152, 256><<<512, 256, 0, s>>>((const uint4*)kc[l], sink, (size_t)49152 / 16); return 0; })) return 1;
        if (timeit("stream 25MB (1024 blk x 24KB, 256thr)", 25.2, true, [&](int l) { stream2_kernel<24576, 256><<<1024, 256, 0, s>>>((const uint4*)kc[l], sink, (size_t)24576 / 16); return 0; })) return 1;
    }
    {
        if (timeit("wave qkv+rope+append", mb_qkv, true, w_qkv)) return 1;
        if (timeit("wave attention", mb_att, true, w_attn)) return 1;
        if (timeit("wave o_proj", mb_o, true, w_o)) return 1;
        if (timeit("wave gate/up GEGLU", mb_gu, true, w_gu)) return 1;
        if (timeit("wave down", mb_dn, true, w_dn)) return 1;
    }
#ifdef VALLE_STAMPS
    {   // in-kernel timeline of a few layers in the middle of a warm eager chain (100 MHz ticks -> us)
        const int NL = 3, NK = 5 * NL;
        unsigned long long* sb; CK(hipMalloc(&sb, (size_t)NK * 4096 * 8));
        for (int r = 0; r < 3; ++r) for (int l = 0; l < L; ++l) RC(wlayer(l));
        CK(hipStreamSynchronize(s));
        CK(hipMemset(sb, 0, (size_t)NK * 4096 * 8));
        for (int l = 0; l < 8; ++l) RC(wlayer(l));
        valle_stamp_buf = sb; valle_stamp_launch = 0;
        for (int l = 8; l < 8 + NL; ++l) RC(wlayer(l));
        valle_stamp_buf = nullptr;
        for (int l = 8 + NL; l < L; ++l) RC(wlayer(l));
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> hs((size_t)NK * 4096);
        CK(hipMemcpy(hs.data(), sb, hs.size() * 8, hipMemcpyDeviceToHost));
        const char* names[5] = {"qkv", "attn", "o", "gu", "down"};
        unsigned long long origin = ~0ull; double prev_end = 0;
        for (int i = 0; i < 1024; ++i) if (hs[i * 4]) origin = std::min(origin, hs[i * 4]);
        printf("launch  nWG | first-start (gap after prev end) | start spread | med loads-landed | med compute | med store-ack | last end   [us, rel. first launch]\n");
        for (int k = 0; k < NK; ++k) {
            std::vector<double> st, ld, cp, sa; double first = 1e30, lasts = 0, last_end = 0;
            for (int i = 0; i < 1024; ++i) {
                const unsigned long long* q = &hs[((size_t)k * 1024 + i) * 4];
                if (!q[0]) continue;
                const double a0 = (q[0] - origin) * 0.01, a1 = (q[1] - origin) * 0.01, a2 = (q[2] - origin) * 0.01, a3 = (q[3] - origin) * 0.01;
                first = std::min(first, a0); lasts = std::max(lasts, a0); last_end = std::max(last_end, a3);
                st.push_back(a0); ld.push_back(a1 - a0); cp.push_back(a2 - a1); sa.push_back(a3 - a2);
            }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
            auto mx = [](std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
            printf("%-5s %5zu | %7.2f (%5.2f) | %5.2f | %5.2f (max %5.2f) | %5.2f (max %5.2f) | %5.2f (max %5.2f) | %7.2f  dur %5.2f\n", names[k % 5], st.size(), first,
                   k ? first - prev_end : 0.0, lasts - first, med(ld), mx(ld), med(cp), mx(cp), med(sa), mx(sa), last_end, last_end - first);
            prev_end = last_end;
        }
        return 0;
    }
#endif
    const double mb_layer = mb_qkv + mb_o + mb_gu + mb_dn + mb_att;
    if (timeit("WAVE LAYER (5 kernels)", mb_layer, false, wlayer)) return 1;
    if (timeit("WAVE LAYER (5 kernels)", mb_layer, true, wlayer)) return 1;
    if (getenv("MICRO_PREFETCH")) {
        // side stream warms the Infinity Cache with the NEXT layer's K/V rows while this layer runs; paced by one event per
        // layer recorded on the main stream (mode 1: behind the QKV launch, 2: behind the attention launch, 3: events only,
        // no prefetch kernel -- the price of the pacing alone)
        const int mode = atoi(getenv("MICRO_PREFETCH"));
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        std::vector<hipEvent_t> ev(L); for (auto& evl : ev) CK(hipEventCreateWithFlags(&evl, hipEventDisableTiming));
        hipEvent_t join; CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        const size_t region16 = (size_t)CT * 128 / 16; const int n16 = (ctx + 1) * 128 / 16;
        auto pace = [&](int l) {
            (void)hipEventRecord(ev[l], s); (void)hipStreamWaitEvent(s2, ev[l], 0);
            const int n = (l + 1) % L;
            if (mode != 3) kv_prefetch_kernel<<<dim3(B * H, 2), 256, 0, s2>>>((const uint4*)kc[n], (const uint4*)vc[n], region16, n16, sink);
            if (l == L - 1) { (void)hipEventRecord(join, s2); (void)hipStreamWaitEvent(s, join, 0); }
        };
        auto player = [&](int l) { int rc = w_qkv(l); if (mode != 2) pace(l); rc |= w_attn(l); if (mode == 2) pace(l); rc |= w_o(l); rc |= w_gu(l); rc |= w_dn(l); return rc; };
        if (timeit("WAVE LAYER + next-layer KV prefetch", mb_layer, false, player)) return 1;
        if (timeit("WAVE LAYER + next-layer KV prefetch", mb_layer, true, player)) return 1;
    }
    printf("last error: %s\n", valle_last_error());
    return 0;
}
