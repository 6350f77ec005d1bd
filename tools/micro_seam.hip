// Dev micro-benchmark (not part of the library): what does an all-to-all hand-off INSIDE one launch cost on
// this chip, in the shape the AR decode step would need (every workgroup publishes a slice of a small
// activation matrix, everybody then reads the whole matrix)?  Compared with the ~2.9 us a kernel boundary
// costs (launch floor + store tail), this decides whether fusing decode phases into a persistent launch pays.
// Hand-off form (MI355X_MICROARCH.md, visibility table, row 3): payload stored sc1 (whole 128-B lines per wave
// store), every storing wave drains vmcnt, barrier, ONE lane adds to an agent-scope counter; consumers poll the
// counter with an sc1 load (one lane, bounded), barrier, then read the payload with sc1 loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void store_sc1_16(uint4* p, uint4 v) {
    // two 8-byte agent-scope relaxed stores = global_store_dwordx2 ... sc1 each
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long load_sc1_8(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// payload_bytes: size of the exchanged matrix; producers: the first n_prod workgroups each write payload/n_prod bytes;
// consumers: the first n_cons workgroups each read the whole payload.  iters hand-offs back to back.
__global__ __launch_bounds__(512) void seam_kernel(unsigned long long* payload, unsigned* counters, unsigned* timeout,
                                                   int payload_bytes, int n_prod, int n_cons, int iters, unsigned long long* sink) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long* buf = payload + (size_t)(it & 1) * (payload_bytes / 8);
        if (wg < n_prod) {
            const int per = payload_bytes / 8 / n_prod;                 // 8-byte words per producer
            for (int i = tid; i < per; i += 512)
                __hip_atomic_store(buf + (size_t)wg * per + i, ((unsigned long long)(it + 1) << 32) | (unsigned)(wg * per + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && wg < n_prod) __hip_atomic_fetch_add(counters + it, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wg < n_cons) {
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(counters + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n_prod) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 2000000u) { atomicAdd(timeout, 1u); break; }          // bounded: never hangs
                }
            }
            __syncthreads();
            const int words = payload_bytes / 8;
            for (int i = tid; i < words; i += 512) {
                const unsigned long long x = load_sc1_8(buf + i);
                if ((x >> 32) != (unsigned long long)(it + 1)) atomicAdd(timeout + 1, 1u);   // stale word
                acc ^= x;
            }
        }
    }
    if (acc == 0x1234567ull) sink[wg] = acc;
}

// Flag-in-data variant (the "LL" protocol of collective libraries): every 8-byte word carries 4 bytes of data and
// the 4-byte epoch, stored sc1 with no drain, no counter, no barrier; a consumer lane re-loads ITS words (sc1) until
// the epoch matches.  One memory round trip less than counter + poll + payload load.
__global__ __launch_bounds__(512) void seam_ll_kernel(unsigned long long* payload, unsigned* timeout, int payload_bytes,
                                                      int n_prod, int n_cons, int iters, unsigned long long* sink) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long* buf = payload + (size_t)(it & 1) * (payload_bytes / 8);
        const unsigned long long epoch = (unsigned long long)(it + 1);
        if (wg < n_prod) {
            const int per = payload_bytes / 8 / n_prod;
            for (int i = tid; i < per; i += 512)
                __hip_atomic_store(buf + (size_t)wg * per + i, (epoch << 32) | (unsigned)(wg * per + i + (unsigned)acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wg < n_cons) {
            const int words = payload_bytes / 8;
            for (int i0 = 0; i0 < words; i0 += 512 * 8) {              // 8 words per lane in flight, then check, re-load the late ones
                unsigned long long x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int i = i0 + j * 512 + tid; x[j] = i < words ? load_sc1_8(buf + i) : (epoch << 32); }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + j * 512 + tid;
                    unsigned spins = 0;
                    while ((x[j] >> 32) < epoch) {                                  // a consumer-only workgroup may be lapped: newer is fine
                        x[j] = load_sc1_8(buf + i);
                        if (++spins > 20000u) { atomicAdd(timeout, 1u); break; }        // bounded: never hangs
                    }
                    acc ^= x[j] & 0xffffffffull;
                }
            }
        }
    }
    if (acc == 0x1234567ull) sink[wg] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const int iters = 400;
    unsigned long long *payload, *sink; unsigned *counters, *tmo;
    CK(hipMalloc(&payload, 2 * (256 << 10))); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&counters, iters * 4)); CK(hipMalloc(&tmo, 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int bytes, prod, cons; } cases[] = {
        {"x    32 KB: 64 -> 256 workgroups", 32 << 10, 64, 256}, {"x    32 KB: 256 -> 256", 32 << 10, 256, 256},
        {"ff  128 KB: 256 -> 64", 128 << 10, 256, 64}, {"ff  128 KB: 256 -> 256", 128 << 10, 256, 256},
        {"flag only (64 B): 256 -> 256", 64 * 8 * 4, 256, 256}};
    for (auto& c : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(counters, 0, iters * 4, s)); CK(hipMemsetAsync(tmo, 0, 16, s)); CK(hipMemsetAsync(payload, 0, 2 * (256 << 10), s));
            CK(hipEventRecord(e0, s));
            seam_kernel<<<256, 512, 0, s>>>(payload, counters, tmo, c.bytes, c.prod, c.cons, iters, sink);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        unsigned h[4]; CK(hipMemcpy(h, tmo, 16, hipMemcpyDeviceToHost));
        printf("%-34s %6.2f us per hand-off   (timeouts %u, stale words %u)\n", c.name, best * 1e3 / iters, h[0], h[1]);
    }
    struct Case2 { const char* name; int bytes, prod, cons; } ll[] = {
        {"LL  x 32 KB data (64 KB words): 64 -> 256", 64 << 10, 64, 256}, {"LL  x 32 KB data: 256 -> 256", 64 << 10, 256, 256},
        {"LL  x 32 KB data: 256 -> 192", 64 << 10, 256, 192},
        {"LL  ff 128 KB data (256 KB words): 256 -> 64", 256 << 10, 256, 64}, {"LL  4 KB data: 256 -> 256", 8 << 10, 256, 256}};
    unsigned long long* big; CK(hipMalloc(&big, 2 * (256 << 10)));
    for (auto& c : ll) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(tmo, 0, 16, s)); CK(hipMemsetAsync(big, 0, 2 * (256 << 10), s));
            CK(hipEventRecord(e0, s));
            seam_ll_kernel<<<256, 512, 0, s>>>(big, tmo, c.bytes, c.prod, c.cons, iters, sink);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        unsigned h[4]; CK(hipMemcpy(h, tmo, 16, hipMemcpyDeviceToHost));
        printf("%-48s %6.2f us per hand-off   (timeouts %u)\n", c.name, best * 1e3 / iters, h[0]);
        fflush(stdout);
    }
    return 0;
}
