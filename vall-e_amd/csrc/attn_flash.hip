// Flash-style causal self-attention for the dense passes (prefill, NAR levels, NAR-len steps), bf16,
// head_dim 64, token-packed variable-length batches.  Replaces SDPA + additive mask at
// models/arch/llama.py:448-456 / :689-730.
//
// One workgroup = 4 waves = 64 or 128 query rows of one (utterance, head); K/V tiles of 64 keys go
// through a double-buffered LDS image (register-prefetched one tile ahead).  Per wave the score tile is computed
// TRANSPOSED, S^T = K.Q^T (MFMA A = K rows, B = Q rows): a lane then owns ONE query (its
// column) and 16 keys of the tile, so the online-softmax state (m, l) and the rescale factor are
// per-lane scalars and P^T is already the B operand of O^T += V^T.P^T -- P never leaves
// registers.  The V^T operand is read straight out of the row-major V tile with the gfx950
// transposing LDS read (ds_read_b64_tr_b16).
#include "common.h"

#ifndef FA_ROW
#define FA_ROW 72   // LDS row stride in elements (144 B: 16-B aligned, shifts banks per row)
#endif

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// QB = query blocks of 16 per wave: a workgroup covers 64*QB queries; every K/V fragment read from LDS
// feeds QB MFMAs.  K/V tiles are double-buffered in LDS: one barrier per tile.  Four waves per SIMD (128 VGPRs; two
// values spill outside the loop): the softmax is a long dependent chain per tile, and residency is what hides it
// (2 / 3 / 4 waves per SIMD: 68 / 57 / 54 us at 16 x 761 tokens).
// Where the rows live.  Token-packed qkv (the model's own layout): q / k / v = qkv + {0, d, 2d}, row stride 3d, head
// stride 64, utterance b starts at row cu[b].  Strided (B, H, T, 64) tensors (the reference's attention plug-in point,
// arch/llama.py:396-403): one batch / head / row stride per tensor, T keys per utterance or seqlens[b] of them.
struct FlashAddr {
    const bf16 *q, *k, *v; bf16* out;
    long q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;   // elements
    const int32_t* cu;          // packed: row offsets [B + 1]; NULL: strided
    const int32_t* seqlens;     // strided: valid rows per utterance (NULL: T_fixed)
    int T_fixed;
    float scale_log2;           // softmax scale x log2(e)
};

template <int QB>
__global__ __launch_bounds__(256, 4) void attn_flash_bf16_kernel(FlashAddr A, int causal) {
    constexpr int QT = 64 * QB;                               // queries per workgroup
    // causal: the last query tiles have the most key tiles -- dispatch them first so the grid's tail is made of short ones
    // Workgroup lin (dispatch order: it lands on XCD lin mod 8) -> (query tile, head, utterance).  The query tiles of one
    // (utterance, head) share its K/V rows: in the plain order they go to as many XCDs as there are tiles and every one of those
    // L2s fetches the rows (6 tiles at 761 tokens: 300 MB over the fabric for 50 MB of K/V per pass); instead XCD x takes the
    // (utterance, head) units x, x + 8, ... with all their query tiles.  FA_NO_XCD_ORDER: the plain order (dev A/B builds).
    int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
#ifndef FA_NO_XCD_ORDER
    {
        const int nq = gridDim.x, H = gridDim.y, nu8 = (int)(gridDim.y * gridDim.z) & ~7;
        const int lin = blockIdx.x + nq * (blockIdx.y + H * blockIdx.z);
        if (lin < nq * nu8) {
            const int sl = lin >> 3, unit = (lin & 7) + 8 * (sl / nq);
            qi = sl % nq; h = unit % H; b = unit / H;
        }
    }
#endif
    const int qt = causal ? (int)gridDim.x - 1 - qi : qi;
    int start = 0, T;
    if (A.cu) { start = A.cu[b]; T = A.cu[b + 1] - start; }
    else T = A.seqlens ? min(A.seqlens[b], A.T_fixed) : A.T_fixed;
    if (qt * QT >= T) return;
    const long brow = A.cu ? (long)start : 0, bidx = A.cu ? 0 : (long)b;     // packed: row offset; strided: batch index
    const bf16* qbase = A.q + bidx * A.q_bs + brow * A.q_rs + (long)h * A.q_hs;
    const bf16* kbase = A.k + bidx * A.k_bs + brow * A.k_rs + (long)h * A.k_hs;
    const bf16* vbase = A.v + bidx * A.v_bs + brow * A.v_rs + (long)h * A.v_hs;
    bf16* obase = A.out + bidx * A.o_bs + brow * A.o_rs + (long)h * A.o_hs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    __shared__ __attribute__((aligned(16))) bf16 Ks[2][64 * FA_ROW];
    __shared__ __attribute__((aligned(16))) bf16 Vs[2][64 * FA_ROW];

    int q[QB];                                                // this lane's queries (columns of S^T)
    bf16x8 qf[QB][2];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        q[j] = qt * QT + (wave * QB + j) * 16 + fr;
        const int qc = q[j] < T ? q[j] : T - 1;
        const bf16* qp = qbase + (long)qc * A.q_rs + fq * 8;
        qf[j][0] = *reinterpret_cast<const bf16x8*>(qp);
        qf[j][1] = *reinterpret_cast<const bf16x8*>(qp + 32);
    }
    const int q_hi = qt * QT + (wave * QB + QB) * 16 - 1;     // largest query of this wave
    const int q_lo = qt * QT + wave * QB * 16;                // smallest
    const int last_q = min(qt * QT + QT, T) - 1;              // largest query of the workgroup
    const int n_kt = causal ? last_q / 64 + 1 : (T + 63) / 64;
    float m[QB], l[QB];
    f32x4 o[QB][4];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        m[j] = -INFINITY; l[j] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float sc = A.scale_log2;                            // softmax scale (1/sqrt(64) in the model) x log2(e); > 0

    // the next tile's K/V rows wait in four named registers (as arrays behind a lambda the compiler kept them in
    // scratch memory, which put the global-load latency back on the critical path of every tile)
    uint4 kr0, kr1, vr0, vr1;
    const int lrow = tid >> 3, lch = tid & 7;                 // this thread stages rows lrow and lrow + 32, 16-byte piece lch
    const bf16* kl = kbase + lch * 8;
    const bf16* vl = vbase + lch * 8;
#define FA_GLOAD(KT) do { \
        int key0 = (KT) * 64 + lrow, key1 = key0 + 32; \
        key0 = key0 < T ? key0 : T - 1; key1 = key1 < T ? key1 : T - 1; \
        kr0 = *reinterpret_cast<const uint4*>(kl + (long)key0 * A.k_rs); vr0 = *reinterpret_cast<const uint4*>(vl + (long)key0 * A.v_rs); \
        kr1 = *reinterpret_cast<const uint4*>(kl + (long)key1 * A.k_rs); vr1 = *reinterpret_cast<const uint4*>(vl + (long)key1 * A.v_rs); \
    } while (0)
    FA_GLOAD(0);
    // per-lane LDS offsets (elements) inside a tile, fixed for the whole loop: every access below is base + constant
    const int woff = lrow * FA_ROW + lch * 8;
    const int koff = fr * FA_ROW + fq * 8;
    const int voff = (fq * 4 + (fr >> 2)) * FA_ROW + (fr & 3) * 4;
    for (int kt = 0; kt < n_kt; ++kt) {
        bf16* Kb = Ks[kt & 1];
        bf16* Vb = Vs[kt & 1];
        *reinterpret_cast<uint4*>(Kb + woff) = kr0;
        *reinterpret_cast<uint4*>(Vb + woff) = vr0;
        *reinterpret_cast<uint4*>(Kb + woff + 32 * FA_ROW) = kr1;
        *reinterpret_cast<uint4*>(Vb + woff + 32 * FA_ROW) = vr1;
        __syncthreads();          // tile kt visible; the other buffer was last read in iteration kt-1, which every wave has left
        if (kt + 1 < n_kt) FA_GLOAD(kt + 1);
        if (causal && kt * 64 > q_hi) continue;              // wave-uniform: this wave's queries see none of these keys
        const bool diag = causal && (kt * 64 + 63 > q_lo);   // some (query, key) pair of this wave is masked
        const bool tail = (kt + 1) * 64 > T;
        // ---- S^T = K . Q^T : 4 key sub-tiles x 2 k-steps, each K fragment feeds QB query blocks
        f32x4 st[QB][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Kb + koff + nt * 16 * FA_ROW);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Kb + koff + nt * 16 * FA_ROW + 32);
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                st[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[j][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                st[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[j][1], st[j][nt], 0, 0, 0);
            }
        }
        // ---- mask (edge tiles only), online softmax in the log2 domain; state is per lane = per query
        bf16x8 pf[QB][2];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            if (diag || tail) {
                // key = kt*64 + fq*4 + c with c = nt*16 + r a compile-time constant: masked <=> c > lim
                int lim = 64;
                if (diag) lim = q[j] - kt * 64 - fq * 4;
                if (tail) lim = min(lim, T - 1 - kt * 64 - fq * 4);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nt * 16 + r > lim) st[j][nt][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(st[j][0][0], st[j][0][1]), fmaxf(st[j][0][2], st[j][0][3]));
#pragma unroll
            for (int nt = 1; nt < 4; ++nt)
                mx = fmaxf(mx, fmaxf(fmaxf(st[j][nt][0], st[j][nt][1]), fmaxf(st[j][nt][2], st[j][nt][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[j], mx * sc);            // sc > 0: max commutes with the scale
            const float ms = (mn == -INFINITY) ? 0.f : mn;    // fully masked so far (rows past T): keep arguments finite
            const float alpha = __builtin_amdgcn_exp2f(m[j] - ms);
            // exponent arguments and the row sum on the packed-fp32 pipe (v_pk_fma_f32 / v_pk_add_f32: two lanes of
            // work per instruction), the exponentials themselves are quarter-rate scalars
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 sc2 = {sc, sc}, nms2 = {-ms, -ms};
            f32x2 ls2 = {0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 a = {st[j][nt][r], st[j][nt][r + 1]};
                    a = __builtin_elementwise_fma(a, sc2, nms2);
                    f32x2 p = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                    st[j][nt][r] = p[0];
                    st[j][nt][r + 1] = p[1];
                    ls2 += p;
                }
            l[j] = l[j] * alpha + (ls2[0] + ls2[1]);
            m[j] = mn;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[j][dt][r] *= alpha;
            // P^T as the B operand: k-slot (fq, e) <-> key 32ks + 16*(e>>2) + 4fq + (e&3)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[j][ks][e] = (bf16)st[j][2 * ks + (e >> 2)][e & 3];
        }
        // ---- O^T += V^T . P^T : A operand through the transposing LDS read, same key mapping
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Vb + voff + ks * 32 * FA_ROW + dt * 16));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Vb + voff + (ks * 32 + 16) * FA_ROW + dt * 16));
                bf16x8 vt;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vt[e] = lo[e]; vt[4 + e] = hi[e]; }
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    o[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pf[j][ks], o[j][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        float lt = l[j];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        if (q[j] < T) {
            const float inv = 1.0f / lt;
            bf16* op = obase + (long)q[j] * A.o_rs + fq * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (bf16)(o[j][dt][r] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 16) = ov;
            }
        }
    }
}

#undef FA_GLOAD

static int flash_launch(const FlashAddr& A, int B, int max_T, int H, int causal, hipStream_t s) {
    if (max_T > 256) {
        dim3 grid(ceil_div(max_T, 128), H, B);
        attn_flash_bf16_kernel<2><<<grid, 256, 0, s>>>(A, causal);
    } else {
        dim3 grid(ceil_div(max_T, 64), H, B);
        attn_flash_bf16_kernel<1><<<grid, 256, 0, s>>>(A, causal);
    }
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

int valle_attn_flash_bf16(const void* qkv, void* out, const int32_t* cu_seqlens, int B, int max_T, int H, int causal,
                          hipStream_t s) {
    const long d = (long)H * 64;
    FlashAddr A{};
    A.q = (const bf16*)qkv; A.k = A.q + d; A.v = A.q + 2 * d; A.out = (bf16*)out;
    A.q_rs = A.k_rs = A.v_rs = 3 * d; A.o_rs = d;
    A.q_hs = A.k_hs = A.v_hs = A.o_hs = 64;
    A.cu = cu_seqlens; A.scale_log2 = 0.125f * 1.4426950408889634f;
    return flash_launch(A, B, max_T, H, causal, s);
}

extern "C" int valle_attn_prefill_hnd(const void* q, const void* k, const void* v, void* out, int B, int H, int T, int Dh,
                                      const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                      const int64_t* out_strides, int causal, float softmax_scale, const int32_t* seqlens,
                                      int dtype, void* stream) {
    VALLE_REQUIRE(q && k && v && out && q_strides && k_strides && v_strides && out_strides, VALLE_E_ARG, "valle_attn_prefill_hnd: null pointer");
    VALLE_REQUIRE(B > 0 && H > 0 && T > 0, VALLE_E_ARG, "valle_attn_prefill_hnd: empty shape (B=%d, H=%d, T=%d)", B, H, T);
    VALLE_REQUIRE(softmax_scale > 0.f, VALLE_E_ARG, "valle_attn_prefill_hnd: softmax_scale must be positive");
    if (dtype != VALLE_BF16 || Dh != 64) {
        valle_set_error("valle_attn_prefill_hnd: built for bf16 and head_dim 64 (got dtype %d, head_dim %d)", dtype, Dh);
        return VALLE_E_UNSUPPORTED;
    }
    const int64_t* st[4] = {q_strides, k_strides, v_strides, out_strides};
    const void* ptrs[4] = {q, k, v, out};
    for (int i = 0; i < 4; ++i) {
        VALLE_REQUIRE(st[i][0] % 4 == 0 && st[i][1] % 4 == 0 && st[i][2] % 4 == 0 && (i == 3 || (st[i][0] % 8 == 0 && st[i][1] % 8 == 0 && st[i][2] % 8 == 0)),
                      VALLE_E_ARG, "valle_attn_prefill_hnd: strides must keep rows 16-byte aligned (8 bytes for out)");
        VALLE_REQUIRE(((uintptr_t)ptrs[i] & (i == 3 ? 7 : 15)) == 0, VALLE_E_ARG, "valle_attn_prefill_hnd: misaligned tensor");
    }
    FlashAddr A{};
    A.q = (const bf16*)q; A.k = (const bf16*)k; A.v = (const bf16*)v; A.out = (bf16*)out;
    A.q_bs = q_strides[0]; A.q_hs = q_strides[1]; A.q_rs = q_strides[2];
    A.k_bs = k_strides[0]; A.k_hs = k_strides[1]; A.k_rs = k_strides[2];
    A.v_bs = v_strides[0]; A.v_hs = v_strides[1]; A.v_rs = v_strides[2];
    A.o_bs = out_strides[0]; A.o_hs = out_strides[1]; A.o_rs = out_strides[2];
    A.seqlens = seqlens; A.T_fixed = T; A.scale_log2 = softmax_scale * 1.4426950408889634f;
    return flash_launch(A, B, T, H, causal, (hipStream_t)stream);
}
