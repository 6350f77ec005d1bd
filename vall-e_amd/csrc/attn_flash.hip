// Flash-style causal self-attention for the dense passes (prefill, NAR levels, NAR-len steps), bf16,
// head_dim 64, token-packed variable-length batches.  Replaces SDPA + additive mask at
// models/arch/llama.py:448-456 / :689-730.
//
// One workgroup = 4 waves = 64 query rows of one (utterance, head); K/V tiles of 64 keys go
// through LDS (register-prefetched one tile ahead).  Per wave the score tile is computed
// TRANSPOSED, S^T = K.Q^T (MFMA A = K rows, B = Q rows): a lane then owns ONE query (its
// column) and 16 keys of the tile, so the online-softmax state (m, l) and the rescale factor are
// per-lane scalars and P^T is already the B operand of O^T += V^T.P^T -- P never leaves
// registers.  The V^T operand is read straight out of the row-major V tile with the gfx950
// transposing LDS read (ds_read_b64_tr_b16).
#include "common.h"

#define FA_ROW 72   // LDS row stride in elements (144 B: 16-B aligned, shifts banks per row)

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__global__ __launch_bounds__(256) void attn_flash_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                              const int32_t* __restrict__ cu, int H, int causal) {
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int start = cu[b], T = cu[b + 1] - start;
    if (qt * 64 >= T) return;
    const int d = H * 64, ld = 3 * d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * FA_ROW];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * FA_ROW];

    const int q = qt * 64 + wave * 16 + fr;                 // this lane's query (column of S^T)
    const int qc = q < T ? q : T - 1;
    const bf16* qp = qkv + (size_t)(start + qc) * ld + h * 64 + fq * 8;
    bf16x8 qf[2];
    qf[0] = *reinterpret_cast<const bf16x8*>(qp);
    qf[1] = *reinterpret_cast<const bf16x8*>(qp + 32);

    const int n_kt = causal ? qt + 1 : (T + 63) / 64;
    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float sc = 0.125f * 1.4426950408889634f;          // 1/sqrt(64) * log2(e)

    uint4 kr[2], vr[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            int key = kt * 64 + row;
            key = key < T ? key : T - 1;
            const bf16* p = qkv + (size_t)(start + key) * ld + d + h * 64 + ch * 8;
            kr[i] = *reinterpret_cast<const uint4*>(p);
            vr[i] = *reinterpret_cast<const uint4*>(p + d);
        }
    };
    gload(0);
    for (int kt = 0; kt < n_kt; ++kt) {
        __syncthreads();                                     // the previous tile is fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            *reinterpret_cast<uint4*>(&Ks[row * FA_ROW + ch * 8]) = kr[i];
            *reinterpret_cast<uint4*>(&Vs[row * FA_ROW + ch * 8]) = vr[i];
        }
        __syncthreads();
        if (kt + 1 < n_kt) gload(kt + 1);
        // ---- S^T = K . Q^T : 4 key sub-tiles x 2 k-steps
        f32x4 st[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            st[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ks[(nt * 16 + fr) * FA_ROW + ks * 32 + fq * 8]);
                st[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], st[nt], 0, 0, 0);
            }
        }
        // ---- scale, mask, online softmax (state is per lane = per query)
        const bool diag = causal ? (kt == qt) : false;
        const bool tail = (kt + 1) * 64 > T;
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 64 + nt * 16 + fq * 4 + r;
                float v = st[nt][r] * sc;
                if ((tail && key >= T) || (diag && key > q)) v = -INFINITY;
                st[nt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        float ls = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[nt][r] - mn);
                st[nt][r] = p;
                ls += p;
            }
        l = l * alpha + ls;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        // ---- P^T as the B operand: k-slot (fq, j) <-> key 32ks + 16*(j>>2) + 4fq + (j&3)
        bf16x8 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[ks][j] = (bf16)st[2 * ks + (j >> 2)][j & 3];
        // ---- O^T += V^T . P^T : A operand through the transposing LDS read, same key mapping
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int krow = ks * 32 + fq * 4 + (fr >> 2);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int col = dt * 16 + (fr & 3) * 4;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&Vs[krow * FA_ROW + col]);
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&Vs[(krow + 16) * FA_ROW + col]);
                bf16x8 vt;
#pragma unroll
                for (int j = 0; j < 4; ++j) { vt[j] = lo[j]; vt[4 + j] = hi[j]; }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pf[ks], o[dt], 0, 0, 0);
            }
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (q < T) {
        const float inv = 1.0f / l;
        bf16* op = out + (size_t)(start + q) * d + h * 64 + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16)(o[dt][r] * inv);
            *reinterpret_cast<bf16x4*>(op + dt * 16) = ov;
        }
    }
}

int valle_attn_flash_bf16(const void* qkv, void* out, const int32_t* cu_seqlens, int B, int max_T, int H, int causal,
                          hipStream_t s) {
    dim3 grid(ceil_div(max_T, 64), H, B);
    attn_flash_bf16_kernel<<<grid, 256, 0, s>>>((const bf16*)qkv, (bf16*)out, cu_seqlens, H, causal);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}
