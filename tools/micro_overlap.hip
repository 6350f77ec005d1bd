// Dev micro-benchmark (not part of the library): do two dependent chains of small streaming
// kernels, issued on two HIP streams, overlap on one MI355X?  Prices the idea of running the
// AR decode step as two half-batch chains so one chain's launch/latency floor hides under the other.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int BYTES_PER_BLOCK, int NT>
__global__ __launch_bounds__(NT) void stream_kernel(const uint4* __restrict__ p, uint4* out) {
    const uint4* base = p + (size_t)blockIdx.x * (BYTES_PER_BLOCK / 16);
    uint4 acc = make_uint4(0, 0, 0, 0);
    constexpr int N = BYTES_PER_BLOCK / 16 / NT;
    uint4 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = base[threadIdx.x + NT * i];
#pragma unroll
    for (int i = 0; i < N; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[blockIdx.x] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const int L = 48;                               // 48 x 16.8 MB = 805 MB > MALL
    const size_t SZ = 16u << 20;
    std::vector<uint4*> w(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&w[l], SZ)); CK(hipMemset(w[l], 0x11, SZ)); }
    uint4* sink; CK(hipMalloc(&sink, 1 << 20));
    hipStream_t s[4]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1, ej[4]; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& x : ej) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    const int reps = 10;
    // chain of L kernels per rep; variant: nstreams chains, each streaming `mb` MB per kernel; second chain reads
    // the SAME buffer as the first (what a half-batch split would do: the second read can come from the MALL)
    auto run = [&](const char* name, int nstreams, int which, bool same_buf, bool use_graph) {
        auto launch = [&](int st, int l) {
            const uint4* p = w[same_buf ? l : (l + st * 7) % L];
            switch (which) {
                case 0: stream_kernel<8192, 256><<<256, 256, 0, s[st]>>>(p, sink); break;           // 2 MB
                case 1: stream_kernel<32768, 256><<<192, 256, 0, s[st]>>>(p, sink); break;          // 6.3 MB
                case 2: stream_kernel<32768, 256><<<256, 256, 0, s[st]>>>(p, sink); break;          // 8.4 MB
                case 3: stream_kernel<65536, 256><<<256, 256, 0, s[st]>>>(p, sink); break;          // 16.8 MB
            }
        };
        for (int l = 0; l < L; ++l) for (int st = 0; st < nstreams; ++st) launch(st, l);
        CK(hipDeviceSynchronize());
        float ms;
        if (!use_graph) {
            CK(hipEventRecord(e0, s[0]));
            for (int st = 1; st < nstreams; ++st) { CK(hipStreamWaitEvent(s[st], e0, 0)); }
            for (int r = 0; r < reps; ++r) for (int l = 0; l < L; ++l) for (int st = 0; st < nstreams; ++st) launch(st, l);
            for (int st = 1; st < nstreams; ++st) { CK(hipEventRecord(ej[st], s[st])); CK(hipStreamWaitEvent(s[0], ej[st], 0)); }
            CK(hipEventRecord(e1, s[0]));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        } else {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
            CK(hipEventRecord(ej[0], s[0]));
            for (int st = 1; st < nstreams; ++st) { CK(hipStreamWaitEvent(s[st], ej[0], 0)); }
            for (int l = 0; l < L; ++l) for (int st = 0; st < nstreams; ++st) launch(st, l);
            for (int st = 1; st < nstreams; ++st) { CK(hipEventRecord(ej[st], s[st])); CK(hipStreamWaitEvent(s[0], ej[st], 0)); }
            CK(hipStreamEndCapture(s[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s[0])); CK(hipStreamSynchronize(s[0]));
            CK(hipEventRecord(e0, s[0]));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s[0]));
            CK(hipEventRecord(e1, s[0]));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        const double mbs[] = {2.10, 6.29, 8.39, 16.78};
        const double us = ms * 1e3 / (reps * L);
        printf("%-44s %7.2f us per chain position (%d kernels)  %6.2f TB/s of distinct bytes\n", name, us, nstreams,
               mbs[which] * (same_buf ? 1 : nstreams) / us);
        return 0;
    };
    const char* nm[] = {"2MB", "6.3MB", "8.4MB", "16.8MB"};
    for (int which = 0; which < 4; ++which) {
        char b[96];
        snprintf(b, sizeof b, "%s  1 stream eager", nm[which]); run(b, 1, which, true, false);
        snprintf(b, sizeof b, "%s  1 stream graph", nm[which]); run(b, 1, which, true, true);
        snprintf(b, sizeof b, "%s  2 streams eager, distinct buffers", nm[which]); run(b, 2, which, false, false);
        snprintf(b, sizeof b, "%s  2 branches graph, same buffers", nm[which]); run(b, 2, which, true, true);
        snprintf(b, sizeof b, "%s  2 branches graph, distinct buffers", nm[which]); run(b, 2, which, false, true);
        snprintf(b, sizeof b, "%s  4 branches graph, same buffers", nm[which]); run(b, 4, which, true, true);
    }
    return 0;
}
