// Shared device/host helpers for libvalle_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../include/valle_hip.h"

typedef __bf16 bf16;
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// ---- host-side error plumbing -------------------------------------------------------
void valle_set_error(const char* fmt, ...);

#define VALLE_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            valle_set_error(__VA_ARGS__);         \
            return (code);                        \
        }                                         \
    } while (0)

#define VALLE_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            valle_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VALLE_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define VALLE_LAUNCH_CHECK() VALLE_HIP_CHECK(hipGetLastError())

// ---- device helpers -----------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact erf GELU (transformers ACT2FN["gelu"], reference models/base.py:479)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same function where its result is rounded to bf16 at once (the bf16 epilogues): erfc by the rational form of Abramowitz &
// Stegun 7.1.26 (|error| < 1.5e-7 on erf, and no cancellation for negative x: 1 + erf(z) = erfc(-z) is formed directly) -- one
// v_rcp_f32, one v_exp_f32 and a dozen FMAs instead of the ~60 instructions of erff.  Over all 65,280 finite bf16 inputs the
// rounded result differs from the correctly rounded erf GELU in 169 cases by one bf16 step (the fp32 erf form: 114 cases;
// tools/gelu_bf16_study.py); the fp32 path keeps erff.
__device__ __forceinline__ float gelu_erf_bf16out(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc = p * t * __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
    const float hx = 0.5f * x;
    return x >= 0.f ? hx * (2.0f - erfc) : hx * erfc;
}
template <typename T> __device__ __forceinline__ float gelu_act(float x) {
    if constexpr (sizeof(T) == 2) return gelu_erf_bf16out(x);
    else return gelu_erf(x);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
static inline size_t dtype_size(int dtype) { return dtype == VALLE_BF16 ? 2 : 4; }

// ---- internal launchers shared between translation units ------------------------------
struct valle_model {
    valle_config cfg;
    const void* emb;
    const float* rope;
    valle_layer_weights* layers;
    const void* final_norm;
    int n_cls;
    const void** head_w;
    const float** head_b;
    const void** head_pn;
    int32_t* head_n;
    int device;                 // the GPU that was current in valle_model_create (-1: none)
};

struct SampleArgs {
    const float* logits; int ld, vocab;
    const int32_t* row_item; const int32_t* row_in_item;
    const int32_t* prev; const int32_t* prev_off; const int32_t* prev_len;
    int nar_mode;
    valle_sample_params p;
    int32_t* ids; float* scores;
    // optional fused AR bookkeeping (valle_ar_decode): row == utterance
    int32_t* seq; int seq_cap; int32_t* seq_len; int32_t* ctx_len; int32_t* stopped;
    int32_t* ctx_len2;          // optional second context counter (the null-prompt cache of the guidance loop)
    int early_exit;             // rows whose utterance has stopped[] set append nothing and keep seq_len / ctx_len
    int prev_is_seq;            // prev tokens of row i = seq[i*seq_cap .. + seq_len[i])
    uint64_t step;              // added to the Philox offset
    valle_mirostat_state* mirostat;   // per row: the pick is the mirostat draw (samplers.py:197-238)
    float* proc_out;            // beam search: the processed logits of row r go to proc_out[r*ld ..] and nothing is picked
    // optional fused prologue of the NEXT decode step (wave-streaming path): embed the sampled token into the
    // fragment-order x and its RoPE position
    const bf16* nx_emb; int nx_row0; bf16* nx_x; int32_t* nx_pos; const int32_t* nx_pos0; int nx_d;
    const float* nx_rope_tab; float* nx_rope;   // RoPE table [P][64] -> the new position's row, [32][64]
};
int valle_sample_launch(const SampleArgs& a, int rows, hipStream_t s);
int valle_cfg_combine_last_add(const float* pos, const float* neg, int ld, int vocab, const int32_t* row_start,
                               const int32_t* row_count, int count_add, int B, float strength, float rescale, float* out,
                               int ld_out, hipStream_t s);

// dense QKV projection with RoPE applied in the epilogue (gemm.hip; internal epilogue code, bf16, head_dim 64):
// qkv[m][0:2d] rotated by position pos[m] exactly as valle_rope_kv does, qkv[m][2d:3d] stored as is
#define GEMM_EPI_ROPE 5

#define DEC_EPI_QKV 10
// wave-streaming decode linears (dec_linear.hip): activations in fragment order, norm weight folded into W.
// Fragment order of a [16 utterances][K] activation matrix read by a linear with tile split S (dec_linear.hip):
// K is cut into S parts of K' = K/S; a 1-KiB block = (32-wide k block b of K', utterance group g of C = 16/S) holds, for
// MFMA lane (kq, j = part*C + m'), the 8 values x[g*C + m'][part*K' + b*32 + kq*8 ..].  S = 1 is the plain
// [(k/32)][(k/8)%4][m][k%8] order.
#define DEC_S_O 4       // o_proj reads the attention output in this order
#define DEC_S_DOWN 4    // down_proj reads the GEGLU output in this order
template <int S>
__host__ __device__ __forceinline__ int dec_frag_off(int m, int k, int K) {
    constexpr int C = 16 / S;
    const int Kp = K / S, part = k / Kp, kk = k - part * Kp;
    return (((((kk >> 5) * S + m / C) << 6) + (((kk >> 3) & 3) << 4) + part * C + (m % C)) << 3) + (k & 7);
}
struct DecWArgs {
    const bf16* x; const bf16* W; const float* bias; const bf16* resid;
    void* out; int ldo;
    int B, N, K;
    float eps; int norm;
    const float* rope_m;        // QKV: [32][64] (cos, sin) row of every utterance's position
    const int32_t* ctx_len; const int32_t* slot;
    bf16* kc; bf16* vc; int cache_T; int H; int d;
#ifdef VALLE_STAMPS
    unsigned long long* stamps;
#endif
};
#ifdef VALLE_STAMPS
// dev-only in-kernel timestamps (tools/micro_dec_stamps): [launch][1024 workgroups][4] s_memrealtime ticks (100 MHz)
extern unsigned long long* valle_stamp_buf;
extern int valle_stamp_launch;
static inline unsigned long long* valle_stamp_next() { return valle_stamp_buf ? valle_stamp_buf + (size_t)(valle_stamp_launch++) * 4096 : nullptr; }
#endif
int valle_dec_wave_linear(const DecWArgs& a, int epilogue, hipStream_t s);

int valle_attn_decode_bf16(const void* q, size_t q_row_stride, const void* kcache, const void* vcache,
                           const int32_t* ctx_len, const int32_t* cache_slot, void* out, int B, int H, int cache_T,
                           int ctx_extra, int out_frag, hipStream_t s);

// raw scaled attention products of chosen rows (entropix.hip): [rows][H][ld] fp32, see attn_row_scores_kernel
int valle_attn_row_scores_dense(const void* qkv, int d, const int32_t* rows, int n_rows, const int32_t* cu, int B, int H,
                                float* out, int ld, int dtype, hipStream_t s);
int valle_attn_row_scores_cache(const void* q, size_t q_row_stride, const void* kcache, const int32_t* ctx_len,
                                const int32_t* slot, int ctx_extra, int B, int H, int cache_T, float* out, int ld, int dtype,
                                hipStream_t s);

// attention launchers (ops_basic.hip)
extern "C" int valle_attn_prefill_rowwise(const void* qkv, void* out, const int32_t* cu_seqlens, int B, int M,
                                          int H, int Dh, int causal, int dtype, void* stream);
int valle_attn_prefill_rowwise_kv(const void* qkv, void* out, const int32_t* cu_seqlens, const int32_t* kv_lens, int B, int M,
                                  int H, int Dh, int causal, int dtype, void* stream);
int valle_attn_decode_launch(const void* qkv, const void* kcache, const void* vcache, const int32_t* ctx_len,
                             const int32_t* cache_slot, void* out, int B, int H, int Dh, int cache_T,
                             int ctx_extra, int dtype, void* stream);
int valle_attn_flash_bf16(const void* qkv, void* out, const int32_t* cu_seqlens, int B, int max_T, int H, int causal,
                          hipStream_t s);
