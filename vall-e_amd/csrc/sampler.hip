// Sampler chain, classifier-free guidance and NAR-len masking (gfx950).
//   sample        models/base.py:1280-1430 + samplers.py:16-172
//   cfg_combine   samplers.py:177-191
//   mask_lowest   models/ar_nar.py:358-365
#include "common.h"

#define SAMP_MAXV 2048
#define SAMP_THREADS 256


__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < SAMP_THREADS / 64; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < SAMP_THREADS / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}

// Philox4x32-10 -> one uniform in (0,1]
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint32_t subseq) {
    uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), subseq, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    return ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(SAMP_THREADS) void sample_kernel(SampleArgs a) {
    __shared__ __attribute__((aligned(16))) float vals[SAMP_MAXV];
    __shared__ __attribute__((aligned(16))) float skey[SAMP_MAXV];
    __shared__ __attribute__((aligned(16))) int sidx[SAMP_MAXV];
    __shared__ float red[SAMP_THREADS / 64];
    __shared__ double red64[SAMP_THREADS / 64];
    __shared__ int s_flag;
    __shared__ float s_f;
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const int V = a.vocab;
    const int item = a.row_item ? a.row_item[row] : row;
    const float* lg = a.logits + (size_t)row * a.ld;
    for (int v = tid; v < V; v += SAMP_THREADS) vals[v] = lg[v];
    __syncthreads();

    // previous tokens of this row's utterance
    const int32_t* prev = nullptr;
    int n_prev = 0;
    if (a.prev_is_seq) {
        n_prev = a.seq_len[item];
        prev = a.seq + (size_t)item * a.seq_cap;
    } else if (a.prev != nullptr) {
        n_prev = a.prev_len[item];
        prev = a.prev + a.prev_off[item];
    }

    // Ordered application of one operand per previous token (repetition penalty: divide, DRY: subtract).  The tokens
    // and their operands (computed in parallel by all threads) are staged through LDS in tiles; wave 0 then takes 64
    // positions at a time, one per lane.  Positions with different tokens commute and are applied together; positions
    // of the SAME token must keep their sequence order (x / d1 / d2 is not x / d2 / d1 in floating point), so every
    // pending lane bids for its token with its lane number (LDS min), the lowest -- earliest -- bidder applies its
    // operand and clears the bid, and the others try again.  A 2,000-token history costs ~10 us.
    constexpr int NK = SAMP_MAXV / SAMP_THREADS;
    const int nkv = (V + SAMP_THREADS - 1) / SAMP_THREADS;
    __shared__ int claim[SAMP_MAXV];
    for (int v = tid; v < SAMP_MAXV; v += SAMP_THREADS) claim[v] = 0x7fffffff;
    auto ordered_apply = [&](int limit, bool subtract, auto operand) {
        for (int t0 = 0; t0 < limit; t0 += SAMP_MAXV) {
            const int nt = min(limit - t0, SAMP_MAXV);
            __syncthreads();
            for (int i = tid; i < nt; i += SAMP_THREADS) {
                sidx[i] = prev[t0 + i];
                skey[i] = operand(t0 + i);
            }
            __syncthreads();
            if (tid < 64) {
                for (int base = 0; base < nt; base += 64) {
                    const int i = base + tid;
                    const int tok = i < nt ? sidx[i] : -1;
                    const float dv = i < nt ? skey[i] : 1.f;
                    bool pending = (unsigned)tok < (unsigned)V;
                    while (__ballot(pending)) {
                        if (pending) atomicMin(&claim[tok], tid);
                        if (pending && claim[tok] == tid) {
                            const float cur = vals[tok];
                            vals[tok] = subtract ? cur - dv : cur / dv;
                            claim[tok] = 0x7fffffff;
                            pending = false;
                        }
                    }
                }
            }
        }
        __syncthreads();
    };

    // ---- repetition penalty (samplers.py:16-50): for every previous token, in order, logit[token] /= factor * distance^decay
    if (a.p.repetition_penalty != 1.0f && prev != nullptr && n_prev > 0) {
        int limit;
        if (a.nar_mode) limit = a.row_in_item ? a.row_in_item[row] : 0;  // only tokens before this row
        else limit = (n_prev == 1) ? 0 : n_prev;                           // 1 row & 1 prev => "NAR": start=1 empties the slice
        if (limit > n_prev) limit = n_prev;
        const double factor = (double)a.p.repetition_penalty, decay = (double)a.p.repetition_penalty_decay;
        const float fdiv = (float)factor;
        if (decay == 0.0) {
            // every position divides by the same number, so the divisions of a column commute: count the column's
            // occurrences (LDS integer adds) and divide that many times -- the sequence of operations on a logit is the
            // ordered loop's (x / f / f ...), without its serial part (a 2,000-token history: ~1 us against ~11)
            __syncthreads();
            for (int v = tid; v < SAMP_MAXV; v += SAMP_THREADS) sidx[v] = 0;
            __syncthreads();
            for (int i = tid; i < limit; i += SAMP_THREADS) {
                const int tok = prev[i];
                if ((unsigned)tok < (unsigned)V) atomicAdd(&sidx[tok], 1);
            }
            __syncthreads();
            for (int v = tid; v < V; v += SAMP_THREADS) {
                const int cnt = sidx[v] * a.p.rep_pen_passes;
                float x = vals[v];
                for (int c = 0; c < cnt; ++c) x = x / fdiv;
                vals[v] = x;
            }
            __syncthreads();
        } else {
            for (int pass = 0; pass < a.p.rep_pen_passes; ++pass)
                ordered_apply(limit, false, [&](int i) { return (float)(factor * pow((double)(n_prev - i), decay)); });
        }
    }
    // ---- length penalty, AR rows only (samplers.py:56-61; length = len(prev) + 1, base.py:1377)
    if (!a.nar_mode && a.p.length_penalty != 0.0f && (prev != nullptr || a.prev_is_seq) && a.p.stop_token < V) {
        if (tid == 0) {
            const float div = (float)pow((double)(n_prev + 1), (double)a.p.length_penalty);
            vals[a.p.stop_token] = vals[a.p.stop_token] / div;
        }
        __syncthreads();
    }

    // Order statistics without sorting.  Every column gets a 64-bit key that orders like a stable descending sort of the
    // logits does -- (monotone image of the float) << 32 | ~index, so a larger logit, then a lower index, is a larger key
    // and no two keys are equal.  "The k-th largest key" and "the lowest-ranked column whose preceding softmax mass is
    // still <= P" are then found by a bit-by-bit binary search over the key, each probe one block reduction over the
    // 4-5 columns a thread holds in registers: 43 probes (32 value bits + 11 index bits) instead of a 66-phase bitonic sort.
    typedef unsigned long long u64;
    u64 key[NK];
    float ev[NK];
    auto load_keys = [&](float mx) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int v = tid + SAMP_THREADS * k;
            if (v < V) {
                const float x = vals[v];
                unsigned u = __float_as_uint(x);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                key[k] = ((u64)u << 32) | (unsigned)(~v);
                ev[k] = expf(x - mx);
            } else { key[k] = 0; ev[k] = 0.f; }
        }
    };
    __shared__ int cnt_red[SAMP_THREADS / 64];
    auto block_count = [&](int c) {
        c = (int)wave_sum((float)c);                 // <= 2048: exact in fp32
        __syncthreads();
        if ((tid & 63) == 0) cnt_red[tid >> 6] = c;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int i = 0; i < SAMP_THREADS / 64; ++i) t += cnt_red[i];
        return t;
    };
    // The k-th largest key (1 <= k <= V; = the largest T with #(key >= T) >= k): radix select, 4 bits per pass over the 32
    // value bits and 3 + 4 + 4 over the 11 index bits.  A pass counts the keys that match the digits chosen so far into 16
    // LDS bins (integer adds: order-free), one barrier, every thread then scans the bins from the top.  11 barriers instead
    // of the 2 x 43 of a bit-by-bit search with a block reduction per probe (14 -> ~2 us at 1,025 columns).
    __shared__ int hist[11][16];
    auto kth_largest_key = [&](int k) {
        __syncthreads();                             // the previous call's bins have been read by every thread
        if (tid < 11 * 16) (&hist[0][0])[tid] = 0;
        __syncthreads();
        u64 prefix = 0, mask = 0;
        int k_rem = k;
#pragma unroll
        for (int p = 0; p < 11; ++p) {
            const int shift = p < 8 ? 60 - 4 * p : (p == 8 ? 8 : (p == 9 ? 4 : 0));
            const int width = p == 8 ? 3 : 4;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int v = tid + SAMP_THREADS * kk;
                if (kk < nkv && v < V && (key[kk] & mask) == prefix)
                    atomicAdd(&hist[p][(int)((key[kk] >> shift) & ((1u << width) - 1))], 1);
            }
            __syncthreads();
            int cum = 0, d = (1 << width) - 1;
            for (; d > 0; --d) {                     // the highest digit whose count, with those above it, reaches k_rem
                const int c = hist[p][d];
                if (cum + c >= k_rem) break;
                cum += c;
            }
            k_rem -= cum;
            prefix |= (u64)d << shift;
            mask |= (u64)((1u << width) - 1) << shift;
        }
        return prefix | 0xFFFFF800ull;               // bits 31..11 are ones in every key of the row
    };
    // ---- min-p (samplers.py:74-90): drop p < min_p * p_max, but never the 32 largest
    if (a.p.min_p > 0.0f) {
        float mx = -INFINITY;
        for (int v = tid; v < V; v += SAMP_THREADS) mx = fmaxf(mx, vals[v]);
        mx = block_reduce_max(mx, red);
        load_keys(mx);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) se += ev[k];
        se = block_reduce_sum(se, red);
        const float thr = a.p.min_p * (1.0f / se);  // p_max = exp(0)/se
        const u64 k32 = V > 32 ? kth_largest_key(32) : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int v = tid + SAMP_THREADS * k;
            if (v < V && key[k] < k32 && ev[k] / se < thr) vals[v] = -INFINITY;
        }
        __syncthreads();
    }
    // ---- top-k (samplers.py:102-106): keep logits >= the k-th largest value (ties kept); top-p (:108-123): softmax of
    // what is left, drop a column when the mass ranked before it exceeds top_p (the first column always stays)
    // filter_passes = 2 replays the reference's in-place filter for NAR-len's second sample call (ar_nar.py:447-453 samples
    // twice on the same tensors; top-p on the already filtered, renormalised row can drop more columns)
    for (int fpass = 0; fpass < (a.p.filter_passes > 1 ? a.p.filter_passes : 1) && (a.p.top_k > 0 || a.p.top_p < 1.0f); ++fpass) {
        float mx = -INFINITY;
        for (int v = tid; v < V; v += SAMP_THREADS) mx = fmaxf(mx, vals[v]);
        mx = block_reduce_max(mx, red);
        load_keys(mx);
        u64 kth_hi = 0;                              // value part of the k-th largest key; columns with a smaller value go
        if (a.p.top_k > 0) {
            int k = a.p.top_k < 1 ? 1 : a.p.top_k;
            if (k > V) k = V;
            kth_hi = kth_largest_key(k) >> 32;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
                if ((key[kk] >> 32) < kth_hi) ev[kk] = 0.f;      // exp(-inf - max) = 0 in the softmax that follows
        }
        u64 cut = 0;                                 // columns with key < cut are dropped by top-p
        if (a.p.top_p < 1.0f) {
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) se += ev[k];
            se = block_reduce_sum(se, red);
            const float budget = a.p.top_p * se;     // mass ranked before a column must not exceed this
            // columns that still carry mass: those top-k kept (ties included); without top-k the whole row
            int n_live = V;
            if (a.p.top_k > 0 || V <= 64) {
                int c = 0;
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) c += (kk < nkv && tid + SAMP_THREADS * kk < V && (key[kk] >> 32) >= kth_hi) ? 1 : 0;
                n_live = block_count(c);
            }
            if (n_live <= 64) {
                // Few live columns (top-k in front of top-p, the usual pairing): what the reference does, literally -- sort
                // them in descending order (one per lane of wave 0, rank = number of larger keys), add their masses up one
                // after the other in that order, and cut where the mass in front of a column exceeds the budget
                // (samplers.py:108-123: sort, cumsum, shift).  ~2 us against the 43 block-wide probes of the search below.
                __shared__ u64 c_key[64], c_sorted[64], c_cut;
                __shared__ float c_ev[64], c_evs[64];
                __shared__ int c_n;
                if (tid == 0) c_n = 0;
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < NK; ++kk)
                    if (kk < nkv && tid + SAMP_THREADS * kk < V && (key[kk] >> 32) >= kth_hi) {
                        const int slot = atomicAdd(&c_n, 1);      // any order: the sort below undoes it
                        c_key[slot] = key[kk]; c_ev[slot] = ev[kk];
                    }
                __syncthreads();
                if (tid < 64) {
                    const u64 myk = tid < n_live ? c_key[tid] : 0ull;
                    const float mye = tid < n_live ? c_ev[tid] : 0.f;
                    int rank = 0;
                    for (int j = 0; j < n_live; ++j) rank += c_key[j] > myk ? 1 : 0;
                    if (tid < n_live) { c_sorted[rank] = myk; c_evs[rank] = mye; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (tid == 0) {
                        float run = 0.f;
                        u64 cutk = 0;
                        for (int r = 0; r < n_live; ++r) {
                            if (!(run <= budget)) { cutk = r > 0 ? c_sorted[r - 1] : ~0ull; break; }   // rank r and all behind it go
                            run += c_evs[r];
                        }
                        c_cut = cutk;
                    }
                }
                __syncthreads();
                cut = c_cut;
            } else {
                // smallest T with sum(e_u : key_u > T) <= budget
                u64 T = ~0ull;
                for (int bit = 63; bit >= 0; --bit) {
                    if (bit < 32 && bit > 10) continue;
                    const u64 cand = T & ~(1ull << bit);
                    float g = 0.f;
#pragma unroll
                    for (int kk = 0; kk < NK; ++kk) g += (key[kk] > cand) ? ev[kk] : 0.f;
                    g = block_reduce_sum(g, red);
                    if (g <= budget) T = cand;
                }
                cut = T;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int v = tid + SAMP_THREADS * kk;
            if (v < V && ((key[kk] >> 32) < kth_hi || key[kk] < cut)) vals[v] = -INFINITY;
        }
        __syncthreads();
    }
    // ---- dynamic temperature (samplers.py:128-141) or plain temperature (base.py:1389-1392)
    if (a.p.temperature64 != a.p.min_temperature64) {
        float mxr = -INFINITY;
        for (int v = tid; v < V; v += SAMP_THREADS) mxr = fmaxf(mxr, vals[v]);
        mxr = block_reduce_max(mxr, red);
        double se = 0.0;                       // the reference sums Python doubles of exp(fp32 difference)
        for (int v = tid; v < V; v += SAMP_THREADS) se += exp((double)(vals[v] - mxr));
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) red64[tid >> 6] = se;
        __syncthreads();
        se = 0.0;
        for (int i = 0; i < SAMP_THREADS / 64; ++i) se += red64[i];
        const double p_max = 1.0 / se;
        const double T = a.p.temperature64, Tmin = a.p.min_temperature64;
        const float dyn = (float)(T - (T - Tmin) / (1.0 + exp(-10.0 * (p_max - 0.5))));
        for (int v = tid; v < V; v += SAMP_THREADS) vals[v] = vals[v] / dyn;
        __syncthreads();
    } else if (a.p.temperature > 0.0f) {
        for (int v = tid; v < V; v += SAMP_THREADS) vals[v] = vals[v] / a.p.temperature;
        __syncthreads();
    }
    // ---- top-n-sigma, n = 1 whenever top_no > 0 (base.py:1395-1396 passes no n; samplers.py:164-172)
    if (a.p.top_no > 0.0f) {
        float s = 0.f, mx = -INFINITY;
        int bad = 0;
        for (int v = tid; v < V; v += SAMP_THREADS) {
            const float x = vals[v];
            if (!(fabsf(x) <= 3.0e38f)) bad = 1;  // -inf / nan present: torch.std gives nan -> everything is masked
            s += x; mx = fmaxf(mx, x);
        }
        if (tid == 0) s_flag = 0;
        __syncthreads();
        if (bad) s_flag = 1;
        s = block_reduce_sum(s, red);
        mx = block_reduce_max(mx, red);
        const int any_bad = s_flag;
        const float mean = s / (float)V;
        float ss = 0.f;
        for (int v = tid; v < V; v += SAMP_THREADS) { const float dlt = vals[v] - mean; ss += dlt * dlt; }
        ss = block_reduce_sum(ss, red);
        const float sd = sqrtf(ss / (float)(V - 1));
        const float thr = mx - sd;
        __syncthreads();
        for (int v = tid; v < V; v += SAMP_THREADS) {
            if (any_bad || !(vals[v] >= thr)) vals[v] = -INFINITY;
        }
        __syncthreads();
    }
    // ---- DRY (samplers.py:248-275 as it runs: one entry per POSITION of prev, in order; the loop stops at the
    // first entry shorter than allowed_length, and position 0 always has length 1, so allowed_length >= 2 is a no-op)
    if (a.p.dry_multiplier > 0.0 && prev != nullptr && n_prev > 0 && a.p.dry_allowed_length < 2) {
        const int allowed = a.p.dry_allowed_length;
        const int cap = allowed > 50 ? allowed : 50;
        ordered_apply(n_prev, true, [&](int i) {
            int length = 1;
            while (length < cap) {
                const int j = i - length;
                if (j < 0) break;
                if (prev[j] != prev[n_prev - length - 1]) break;
                ++length;
            }
            return (float)(a.p.dry_multiplier * pow(a.p.dry_base, (double)(length - allowed)));
        });
    }
    if (a.proc_out != nullptr) {        // beam search picks across rows (valle_beam_pick): hand the processed row over
        __syncthreads();
        for (int v = tid; v < V; v += SAMP_THREADS) a.proc_out[(size_t)row * a.ld + v] = vals[v];
        return;
    }
    // ---- pick: argmax (lowest index on ties) or categorical
    float mx = -INFINITY;
    for (int v = tid; v < V; v += SAMP_THREADS) mx = fmaxf(mx, vals[v]);
    mx = block_reduce_max(mx, red);
    float se = 0.f;
    for (int v = tid; v < V; v += SAMP_THREADS) se += expf(vals[v] - mx);
    se = block_reduce_sum(se, red);   // all -inf: exp(nan) -> nan, like the reference's softmax
    int tok;
    if (a.mirostat != nullptr) {
        // samplers.py:197-238.  Descending sort of the processed row (bitonic, in LDS; equal logits: lower index first)
        for (int v = tid; v < SAMP_MAXV; v += SAMP_THREADS) { skey[v] = v < V ? vals[v] : -INFINITY; sidx[v] = v; }
        __syncthreads();
        for (int k2 = 2; k2 <= SAMP_MAXV; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < SAMP_MAXV; t += SAMP_THREADS) {
                    const int u = t ^ j;
                    if (u > t) {
                        const float x = skey[t], y = skey[u];
                        const int xi = sidx[t], yi = sidx[u];
                        const bool x_first = (x > y) || (x == y && xi < yi) || (y != y && x == x);
                        const bool desc = (t & k2) == 0;
                        if (desc ? !x_first : x_first) { skey[t] = y; skey[u] = x; sidx[t] = yi; sidx[u] = xi; }
                    }
                }
                __syncthreads();
            }
        // prob_original = softmax(sorted) in fp32 (mx, se of the whole row are the sorted row's too)
        __shared__ int s_k;
        valle_mirostat_state* st = a.mirostat + item;
        if (tid == 0) {
            double num = 0.0, den = 0.0;
            bool zero = V < 101;
            for (int i = 0; i < 100 && !zero; ++i) {                       // compute_k (:198-211), Python float arithmetic
                const double p0 = (double)(expf(skey[i] - mx) / se), p1 = (double)(expf(skey[i + 1] - mx) / se);
                if (p1 == 0.0) { zero = true; break; }
                const double b = p0 / p1, t = (double)(i + 2) / (double)(i + 1);
                num += log(b) * log(t);
                den += log(t) * log(t);
            }
            // degenerate results are reported as the reference meets them (k < 0, decoded by Base._sample_mirostat):
            // -1 a division by zero (ZeroDivisionError), -2 a power of a negative base = a complex number in Python
            // (round() raises TypeError; NaN here), -3 an infinite value (OverflowError), -4 an empty candidate slice
            int k = -1;
            if (!zero) {
                const double s_hat = num / den, eps = s_hat - 1.0;
                const double dn = 1.0 - pow(st->n, -eps);
                if (s_hat != 0.0 && dn != 0.0) {
                    const double kf = pow((eps * pow(2.0, st->max_surprise)) / dn, 1.0 / s_hat);
                    if (kf != kf) k = -2;
                    else if (kf > 1.7e308 || kf < -1.7e308) k = -3;
                    else {
                        const double kr = rint(kf) + 1.0;                  // round(): half to even, like rint
                        // sorted[0:k]: k above V keeps the row, k <= 0 counts from the end (Python slice)
                        const double ke = kr > (double)V ? (double)V : (kr <= 0.0 ? (double)V + kr : kr);
                        k = ke >= 1.0 ? (int)ke : -4;
                    }
                }
            }
            s_k = k;
        }
        __syncthreads();
        const int k = s_k;
        if (k < 0) {                                                   // ZeroDivisionError in the reference: nothing is drawn
            if (tid == 0) { st->k = k; st->token = 0; a.ids[row] = 0; if (a.scores) a.scores[row] = 0.f; }
            return;
        }
        float sek = 0.f;
        for (int i = tid; i < k; i += SAMP_THREADS) sek += expf(skey[i] - mx);
        sek = block_reduce_sum(sek, red);
        if (tid == 0) {
            // draw from softmax(sorted[:k]) by inverse CDF over the sorted order (torch.multinomial's stream is torch's own)
            const float u = philox_uniform(a.p.seed, a.p.offset + a.step, (uint32_t)row) * sek;
            float run = 0.f; int pick = k - 1;
            for (int i = 0; i < k; ++i) { run += expf(skey[i] - mx); if (run >= u) { pick = i; break; } }
            const double p_pick = (double)(expf(skey[pick] - mx) / se);
            const double surprise = log2(1.0 / p_pick);
            st->index_surprise = surprise;
            st->running_total_surprise += surprise;
            st->error_surprise = surprise - st->tau;
            st->max_surprise -= st->eta * st->error_surprise;
            st->token = sidx[pick];
            st->k = k;
            s_flag = sidx[pick];
        }
        __syncthreads();
        tok = s_flag;
    } else if (a.p.temperature <= 0.0f) {
        int best = 1 << 30;
        for (int v = tid; v < V; v += SAMP_THREADS) if (vals[v] == mx && v < best) best = v;
        if (tid == 0) s_flag = 1 << 30;
        __syncthreads();
        atomicMin(&s_flag, best);
        __syncthreads();
        tok = s_flag;
        if (tok >= V) tok = 0;          // all -inf / nan row: torch.argmax returns 0
    } else {
        // inverse-CDF draw over the index order: e[v] = exp(logit - max), first v with cumsum(e)[v] >= u * sum(e);
        // chunked scan (each thread owns a contiguous run of columns)
        const float u = philox_uniform(a.p.seed, a.p.offset + a.step, (uint32_t)row) * se;
        const int chunk = (V + SAMP_THREADS - 1) / SAMP_THREADS;
        const int b0 = tid * chunk, b1 = min(b0 + chunk, V);
        float run = 0.f;
        int last_valid_loc = -1;
        for (int v = b0; v < b1; ++v) {
            const float e = expf(vals[v] - mx);
            if (e > 0.f) last_valid_loc = v;
            run += e;
            skey[v] = run;                                   // running sum inside the chunk
        }
        __shared__ float draw_tot[SAMP_THREADS];
        draw_tot[tid] = run;
        if (tid == 0) s_flag = 1 << 30;
        __syncthreads();
        float before = 0.f;
        for (int t = 0; t < tid; ++t) before += draw_tot[t];  // chunk offsets, summed in index order
        int pick = 1 << 30;
        for (int v = b0; v < b1; ++v)
            if (before + skey[v] >= u) { pick = v; break; }
        atomicMin(&s_flag, pick);
        // fallback when rounding leaves the total below u: the last column with non-zero mass
        int lv = last_valid_loc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lv = max(lv, __shfl_xor(lv, o, 64));
        __shared__ int draw_lv[SAMP_THREADS / 64];
        if ((tid & 63) == 0) draw_lv[tid >> 6] = lv;
        __syncthreads();
        if (tid == 0 && s_flag >= V) {
            int m = 0;
            for (int w = 0; w < SAMP_THREADS / 64; ++w) m = max(m, draw_lv[w]);
            s_flag = m;
        }
        __syncthreads();
        tok = s_flag;
    }
    // version >= 7 sampler (models/base_v2.py:1398-1410): the score is the softmax of the processed logits BEFORE the
    // temperature, at the token drawn from logits / temperature
    float se_unscaled = 0.f;
    const bool unscaled = a.p.score_unscaled != 0 && a.p.temperature > 0.0f && a.p.temperature64 == a.p.min_temperature64 && a.mirostat == nullptr;
    if (unscaled) {
        for (int v = tid; v < V; v += SAMP_THREADS) se_unscaled += expf((vals[v] - mx) * a.p.temperature);
        se_unscaled = block_reduce_sum(se_unscaled, red);
    }
    if (tid == 0) {
        a.ids[row] = tok;
        const float sc = unscaled ? expf((vals[tok] - mx) * a.p.temperature) / se_unscaled : expf(vals[tok] - mx) / se;
        if (a.scores) a.scores[row] = sc;
        if (a.seq != nullptr && !(a.early_exit && a.stopped[item])) {  // fused AR append (models/ar_nar.py:837-842)
            const int n = a.seq_len[item];
            if (n < a.seq_cap) a.seq[(size_t)item * a.seq_cap + n] = tok;
            a.seq_len[item] = n + 1;
            if (a.ctx_len) {
                const int ctx1 = a.ctx_len[item] + 1;
                // early exit: a finished utterance's context is stored as ~ctx_len (negative) -- the decode kernels read
                // ctx_len anyway and leave on a negative value, with no further flag to fetch
                a.ctx_len[item] = (a.early_exit && tok == a.p.stop_token) ? ~ctx1 : ctx1;
            }
            if (a.ctx_len2) a.ctx_len2[item] = a.ctx_len2[item] + 1;     // second cache of the guidance loop
            if (tok == a.p.stop_token) a.stopped[item] = 1;
        }
    }
    (void)s_f;
    if (a.nx_x != nullptr) {        // = dec_prepare_frag_kernel (model.hip) for the token just appended
        const bf16* src = a.nx_emb + (size_t)(a.nx_row0 + tok) * a.nx_d;
        for (int c = tid * 8; c < a.nx_d; c += SAMP_THREADS * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c);
            *reinterpret_cast<bf16x8*>(a.nx_x + (size_t)(item >> 4) * 16 * a.nx_d + dec_frag_off<1>(item & 15, c, a.nx_d)) = v;
        }
        const int npos = (a.nx_pos0 ? a.nx_pos0[item] : 0) + n_prev;                  // position of token index n_prev
        if (tid == 0) a.nx_pos[item] = npos;
        if (tid < 64 && a.nx_rope) a.nx_rope[item * 64 + tid] = a.nx_rope_tab[(size_t)npos * 64 + tid];
    }
}

// Greedy fast path: ONE WAVE per row, the row lives in registers (NJ values per lane), every reduction is a
// wave reduction -- no LDS, no barriers, a few hundred instructions.  Covers what the AR decode loop and the
// NAR levels run by default: argmax (temperature <= 0) with or without top-n-sigma and nothing else switched
// on.  Same formulas as sample_kernel (token = lowest index of the row maximum; 0 when the row holds inf/nan
// under top-n-sigma; score = softmax(processed)[token]), same fused AR bookkeeping and next-step prologue.
template <int NJ>
__global__ __launch_bounds__(64) void sample_greedy_wave_kernel(SampleArgs a) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int V = a.vocab;
    const int item = a.row_item ? a.row_item[row] : row;
    const float* lg = a.logits + (size_t)row * a.ld;
    float x[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int v = lane + 64 * j; x[j] = v < V ? lg[v] : -INFINITY; }
    int n_prev = 0;
    if (a.prev_is_seq) n_prev = a.seq_len[item];
    else if (a.prev != nullptr) n_prev = a.prev_len[item];
    // the append's state is requested now, beside the logits: read behind the argmax, each of these loads would be one
    // more dependent round trip (~1 us) in front of the next step's embedding row
    const bool append = a.seq != nullptr;                       // AR: one row per utterance
    int st_stopped = 0, st_n = 0, st_ctx = 0, st_ctx2 = 0;
    if (append) {
        st_stopped = a.stopped[item];
        st_n = a.seq_len[item];
        if (a.ctx_len) st_ctx = a.ctx_len[item];
        if (a.ctx_len2) st_ctx2 = a.ctx_len2[item];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) mx = fmaxf(mx, x[j]);
    mx = wave_max(mx);
    bool all_masked = false;
    if (a.p.top_no > 0.0f) {                                   // samplers.py:164-172, n = 1
        float sum = 0.f;
        int bad = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (lane + 64 * j < V) { if (!(fabsf(x[j]) <= 3.0e38f)) bad = 1; sum += x[j]; }
        sum = wave_sum(sum);
        all_masked = __any(bad);
        const float mean = sum / (float)V;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (lane + 64 * j < V) { const float dlt = x[j] - mean; ss += dlt * dlt; }
        ss = wave_sum(ss);
        const float thr = mx - sqrtf(ss / (float)(V - 1));
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (all_masked || !(x[j] >= thr)) x[j] = -INFINITY;
        if (all_masked) mx = -INFINITY;
    }
    float se = 0.f;
    int best = 1 << 30;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int v = lane + 64 * j;
        if (v < V) {
            se += expf(x[j] - mx);                              // all -inf: exp(nan), like the reference's softmax
            if (x[j] == mx && v < best) best = v;
        }
    }
    se = wave_sum(se);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
    int tok = best;
    if (tok >= V || all_masked) tok = 0;                        // torch.argmax of an all -inf / nan row
    if (a.nx_x != nullptr) {                                    // next step's fragment-order x and position (its loads go
        const bf16* src = a.nx_emb + (size_t)(a.nx_row0 + tok) * a.nx_d;        // out before the bookkeeping stores below)
        const int npos = (a.nx_pos0 ? a.nx_pos0[item] : 0) + n_prev;
        float rp = 0.f;
        if (a.nx_rope) rp = a.nx_rope_tab[(size_t)npos * 64 + lane];
        for (int c = lane * 8; c < a.nx_d; c += 64 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c);
            *reinterpret_cast<bf16x8*>(a.nx_x + (size_t)(item >> 4) * 16 * a.nx_d + dec_frag_off<1>(item & 15, c, a.nx_d)) = v;
        }
        if (lane == 0) a.nx_pos[item] = npos;
        if (a.nx_rope) a.nx_rope[item * 64 + lane] = rp;
    }
    if (lane == 0) {
        a.ids[row] = tok;
        if (a.scores) a.scores[row] = 1.0f / se;               // exp(x[tok] - mx) = 1; nan when the row is all -inf / nan
        if (append && !(a.early_exit && st_stopped)) {         // fused AR append (models/ar_nar.py:837-842)
            if (st_n < a.seq_cap) a.seq[(size_t)item * a.seq_cap + st_n] = tok;
            a.seq_len[item] = st_n + 1;
            if (a.ctx_len) {
                const int ctx1 = st_ctx + 1;
                // early exit: a finished utterance's context is stored as ~ctx_len (negative) -- the decode kernels read
                // ctx_len anyway and leave on a negative value, with no further flag to fetch
                a.ctx_len[item] = (a.early_exit && tok == a.p.stop_token) ? ~ctx1 : ctx1;
            }
            if (a.ctx_len2) a.ctx_len2[item] = st_ctx2 + 1;    // second cache of the guidance loop
            if (tok == a.p.stop_token) a.stopped[item] = 1;
        }
    }
}

static bool sample_is_plain_greedy(const SampleArgs& a) {
    const valle_sample_params& p = a.p;
    const bool has_prev = a.prev != nullptr || a.prev_is_seq;
    return p.temperature <= 0.0f && p.temperature64 == p.min_temperature64 && p.top_k <= 0 && p.top_p >= 1.0f &&
           p.min_p <= 0.0f && (p.repetition_penalty == 1.0f || !has_prev) && (p.length_penalty == 0.0f || a.nar_mode || !has_prev) &&
           !(p.dry_multiplier > 0.0 && has_prev && p.dry_allowed_length < 2);
}

int valle_sample_launch(const SampleArgs& a, int rows, hipStream_t s) {
    if (a.proc_out == nullptr && a.mirostat == nullptr && sample_is_plain_greedy(a)) {
        if (a.vocab <= 64 * 17) sample_greedy_wave_kernel<17><<<rows, 64, 0, s>>>(a);
        else sample_greedy_wave_kernel<32><<<rows, 64, 0, s>>>(a);
    } else {
        sample_kernel<<<rows, SAMP_THREADS, 0, s>>>(a);
    }
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_sample(const float* logits, int ld, int vocab, int rows, const int32_t* row_item,
                            const int32_t* row_in_item, const int32_t* prev, const int32_t* prev_off,
                            const int32_t* prev_len, int nar_mode, const valle_sample_params* params,
                            int32_t* ids, float* scores, void* stream) {
    VALLE_REQUIRE(logits && params && ids, VALLE_E_ARG, "valle_sample: null pointer");
    VALLE_REQUIRE(vocab > 1 && vocab <= SAMP_MAXV && ld >= vocab, VALLE_E_ARG, "valle_sample: vocab %d (max %d), ld %d", vocab, SAMP_MAXV, ld);
    VALLE_REQUIRE(prev == nullptr || (prev_off && prev_len), VALLE_E_ARG, "valle_sample: prev needs prev_off/prev_len");
    VALLE_REQUIRE(params->rep_pen_passes >= 1 && params->rep_pen_passes <= 2, VALLE_E_ARG, "valle_sample: rep_pen_passes must be 1 or 2");
    VALLE_REQUIRE(params->filter_passes >= 0 && params->filter_passes <= 2, VALLE_E_ARG, "valle_sample: filter_passes must be 0, 1 or 2");
    if (rows == 0) return VALLE_OK;
    SampleArgs a{};
    a.logits = logits; a.ld = ld; a.vocab = vocab; a.row_item = row_item; a.row_in_item = row_in_item;
    a.prev = prev; a.prev_off = prev_off; a.prev_len = prev_len; a.nar_mode = nar_mode; a.p = *params;
    a.ids = ids; a.scores = scores;
    return valle_sample_launch(a, rows, (hipStream_t)stream);
}

extern "C" int valle_sample_mirostat(const float* logits, int ld, int vocab, int rows, const int32_t* prev,
                                     const int32_t* prev_off, const int32_t* prev_len, const valle_sample_params* params,
                                     valle_mirostat_state* states, int32_t* ids, float* scores, void* stream) {
    VALLE_REQUIRE(logits && params && ids && states, VALLE_E_ARG, "valle_sample_mirostat: null pointer");
    VALLE_REQUIRE(vocab >= 101 && vocab <= SAMP_MAXV && ld >= vocab, VALLE_E_ARG, "valle_sample_mirostat: vocab %d (101..%d), ld %d", vocab, SAMP_MAXV, ld);
    VALLE_REQUIRE(prev == nullptr || (prev_off && prev_len), VALLE_E_ARG, "valle_sample_mirostat: prev needs prev_off/prev_len");
    VALLE_REQUIRE(params->rep_pen_passes == 1 && params->filter_passes >= 0 && params->filter_passes <= 1, VALLE_E_ARG, "valle_sample_mirostat: rep_pen_passes / filter_passes");
    if (rows == 0) return VALLE_OK;
    SampleArgs a{};
    a.logits = logits; a.ld = ld; a.vocab = vocab; a.prev = prev; a.prev_off = prev_off; a.prev_len = prev_len; a.nar_mode = 0;
    a.p = *params; a.ids = ids; a.scores = scores; a.mirostat = states;
    return valle_sample_launch(a, rows, (hipStream_t)stream);
}

// ---------------------------------------------------------------------- cfg_combine
// LAST = false: utterance i owns rows [row_a[i], row_a[i + 1]), every row is rewritten in place.
// LAST = true: utterance i owns rows [row_a[i], row_a[i] + row_n[i]) of two read-only row matrices (the logit histories of
// the AR guidance loop) and only its last row's combination is written, to out_last[i].
template <bool LAST>
__global__ __launch_bounds__(1024) void cfg_combine_kernel(float* __restrict__ pos, const float* __restrict__ neg, int ld,
                                                           int vocab, const int32_t* __restrict__ row_a,
                                                           const int32_t* __restrict__ row_n, float* __restrict__ out_last,
                                                           int ld_out, float strength, float rescale, int count_add = 0) {
    const int i = blockIdx.x;
    const int r0 = row_a[i], r1 = LAST ? r0 + row_n[i] + count_add : row_a[i + 1];
    const long n = (long)(r1 - r0) * vocab;
    if (n <= 0) return;
    __shared__ double red[2][16];
    __shared__ double stat[4];
    auto block_sum2 = [&](double a, double b, double& oa, double& ob) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
        __syncthreads();
        oa = 0; ob = 0;
        for (int w = 0; w < 16; ++w) { oa += red[0][w]; ob += red[1][w]; }
    };
    // one pass for the statistics (sums and sums of squares in double: n <= a few 1e5 values of O(10), the cancellation in
    // sum(x^2) - sum(x)^2 / n costs ~1e-10 relative), one pass to write; rows x columns, no per-element division
    float factor = 1.0f;
    if (rescale > 0.f) {
        double sp = 0, sm = 0, qp = 0, qm = 0;
        for (int r = r0; r < r1; ++r) {
            const float* pr = pos + (size_t)r * ld;
            const float* nr = neg + (size_t)r * ld;
            for (int v = threadIdx.x; v < vocab; v += 1024) {
                const float p = pr[v], q = nr[v];
                const double sd = (double)(q + (p - q) * strength);
                sp += p; qp += (double)p * p; sm += sd; qm += sd * sd;
            }
        }
        double tp, tm, tqp, tqm;
        block_sum2(sp, sm, tp, tm);
        block_sum2(qp, qm, tqp, tqm);
        const double vp = (tqp - tp * tp / n) / (n - 1), vm = (tqm - tm * tm / n) / (n - 1);
        const float sdp = (float)sqrt(vp > 0 ? vp : 0), sdm = (float)sqrt(vm > 0 ? vm : 0);
        factor = rescale * (sdp / sdm) + (1.0f - rescale);
    }
    __syncthreads();
    for (int r = LAST ? r1 - 1 : r0; r < r1; ++r) {
        float* pr = pos + (size_t)r * ld;
        const float* nr = neg + (size_t)r * ld;
        float* dst = LAST ? out_last + (size_t)i * ld_out : pr;
        for (int v = threadIdx.x; v < vocab; v += 1024) {
            const float p = pr[v], q = nr[v];
            float s = q + (p - q) * strength;
            if (rescale > 0.f) s = s * factor;
            dst[v] = s;
        }
    }
    (void)stat;
}

// The same combination over whole utterances (NAR-len: 450 rows x 1,025 columns each) with CFG_CH workgroups per utterance
// instead of one (64 utterances on one workgroup each left 3/4 of the chip idle: 446 us per call).  Phase 0: chunk c of
// utterance i adds up its rows' four sums (double) into cfg_part[i][c]; phase 1: every chunk adds the CFG_CH partial sums up
// in order -- the same value in every workgroup of the utterance -- and rewrites its rows.  The partial sums live in the
// caller's workspace (valle_cfg_combine_ws).
#define CFG_CH 8
template <int PHASE>
__global__ __launch_bounds__(1024) void cfg_combine_split_kernel(float* __restrict__ pos, const float* __restrict__ neg, int ld,
                                                                 int vocab, const int32_t* __restrict__ row_a, double* part,
                                                                 float strength, float rescale) {
    const int i = blockIdx.x, c = blockIdx.y;
    const int ra = row_a[i], rb = row_a[i + 1];
    const long n = (long)(rb - ra) * vocab;
    if (n <= 0) return;
    const int per = (rb - ra + CFG_CH - 1) / CFG_CH;
    const int r0 = ra + c * per, r1 = min(r0 + per, rb);
    __shared__ double red[4][16];
    if constexpr (PHASE == 0) {
        double sp = 0, sm = 0, qp = 0, qm = 0;
        for (int r = r0; r < r1; ++r) {
            const float* pr = pos + (size_t)r * ld;
            const float* nr = neg + (size_t)r * ld;
            for (int v = threadIdx.x; v < vocab; v += 1024) {
                const float p = pr[v], q = nr[v];
                const double sd = (double)(q + (p - q) * strength);
                sp += p; qp += (double)p * p; sm += sd; qm += sd * sd;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sp += __shfl_xor(sp, o, 64); sm += __shfl_xor(sm, o, 64); qp += __shfl_xor(qp, o, 64); qm += __shfl_xor(qm, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            red[0][w] = sp; red[1][w] = sm; red[2][w] = qp; red[3][w] = qm;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            double t = 0;
            for (int w = 0; w < 16; ++w) t += red[threadIdx.x][w];
            part[((size_t)i * CFG_CH + c) * 4 + threadIdx.x] = t;
        }
    } else {
        float factor = 1.0f;
        if (rescale > 0.f) {
            double t[4] = {0, 0, 0, 0};
            for (int cc = 0; cc < CFG_CH; ++cc)
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] += part[((size_t)i * CFG_CH + cc) * 4 + k];
            const double vp = (t[2] - t[0] * t[0] / n) / (n - 1), vm = (t[3] - t[1] * t[1] / n) / (n - 1);
            const float sdp = (float)sqrt(vp > 0 ? vp : 0), sdm = (float)sqrt(vm > 0 ? vm : 0);
            factor = rescale * (sdp / sdm) + (1.0f - rescale);
        }
        for (int r = r0; r < r1; ++r) {
            float* pr = pos + (size_t)r * ld;
            const float* nr = neg + (size_t)r * ld;
            for (int v = threadIdx.x; v < vocab; v += 1024) {
                const float p = pr[v], q = nr[v];
                float sv = q + (p - q) * strength;
                if (rescale > 0.f) sv = sv * factor;
                pr[v] = sv;
            }
        }
    }
}

extern "C" int valle_cfg_combine(float* pos, const float* neg, int ld, int vocab, const int32_t* row_off, int B,
                                 float strength, float rescale, void* stream) {
    VALLE_REQUIRE(pos && neg && row_off, VALLE_E_ARG, "valle_cfg_combine: null pointer");
    VALLE_REQUIRE(vocab > 0 && ld >= vocab, VALLE_E_ARG, "valle_cfg_combine: bad vocab/ld");
    if (B == 0) return VALLE_OK;
    cfg_combine_kernel<false><<<B, 1024, 0, (hipStream_t)stream>>>(pos, neg, ld, vocab, row_off, nullptr, nullptr, 0, strength, rescale);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" size_t valle_cfg_combine_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * CFG_CH * 4 * sizeof(double); }

extern "C" int valle_cfg_combine_ws(float* pos, const float* neg, int ld, int vocab, const int32_t* row_off, int B,
                                    float strength, float rescale, void* workspace, size_t workspace_bytes, void* stream) {
    VALLE_REQUIRE(pos && neg && row_off && workspace, VALLE_E_ARG, "valle_cfg_combine_ws: null pointer");
    VALLE_REQUIRE(vocab > 0 && ld >= vocab, VALLE_E_ARG, "valle_cfg_combine_ws: bad vocab/ld");
    VALLE_REQUIRE(B >= 0 && B <= 65535, VALLE_E_ARG, "valle_cfg_combine_ws: %d utterances", B);
    VALLE_REQUIRE(workspace_bytes >= valle_cfg_combine_workspace_bytes(B) && ((uintptr_t)workspace & 7) == 0, VALLE_E_WORKSPACE,
                  "valle_cfg_combine_ws: workspace %zu B too small or misaligned (need %zu)", workspace_bytes, valle_cfg_combine_workspace_bytes(B));
    if (B == 0) return VALLE_OK;
    const dim3 g(B, CFG_CH);
    double* part = (double*)workspace;
    if (rescale > 0.f) cfg_combine_split_kernel<0><<<g, 1024, 0, (hipStream_t)stream>>>(pos, neg, ld, vocab, row_off, part, strength, rescale);
    cfg_combine_split_kernel<1><<<g, 1024, 0, (hipStream_t)stream>>>(pos, neg, ld, vocab, row_off, part, strength, rescale);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

extern "C" int valle_cfg_combine_last(const float* pos, const float* neg, int ld, int vocab, const int32_t* row_start,
                                      const int32_t* row_count, int B, float strength, float rescale, float* out, int ld_out,
                                      void* stream) {
    VALLE_REQUIRE(pos && neg && row_start && row_count && out, VALLE_E_ARG, "valle_cfg_combine_last: null pointer");
    VALLE_REQUIRE(vocab > 0 && ld >= vocab && ld_out >= vocab, VALLE_E_ARG, "valle_cfg_combine_last: bad vocab/ld");
    if (B == 0) return VALLE_OK;
    cfg_combine_kernel<true><<<B, 1024, 0, (hipStream_t)stream>>>((float*)pos, neg, ld, vocab, row_start, row_count, out, ld_out, strength, rescale);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// (internal: the device-side guidance loop of valle_ar_decode_cfg counts the rows as row_count[i] + count_add)
int valle_cfg_combine_last_add(const float* pos, const float* neg, int ld, int vocab, const int32_t* row_start,
                               const int32_t* row_count, int count_add, int B, float strength, float rescale, float* out,
                               int ld_out, hipStream_t s) {
    cfg_combine_kernel<true><<<B, 1024, 0, s>>>((float*)pos, neg, ld, vocab, row_start, row_count, out, ld_out, strength, rescale, count_add);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// ---------------------------------------------------------------------- mask_lowest
#define MASK_MAXN 8192
__global__ __launch_bounds__(256) void mask_lowest_kernel(const float* __restrict__ scores, int32_t* __restrict__ codes,
                                                          uint8_t* __restrict__ is_masked, const int32_t* __restrict__ off,
                                                          const int32_t* __restrict__ kk, int mask_token) {
    __shared__ float s[MASK_MAXN];
    const int i = blockIdx.x;
    const int o = off[i], n = off[i + 1] - o;
    const int k = kk[i];
    for (int j = threadIdx.x; j < n; j += 256) s[j] = scores[o + j];
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 256) {
        const float x = s[j];
        int rank = 0;
        for (int t = 0; t < n; ++t) {
            const float y = s[t];
            rank += (y < x) || (y == x && t < j);
        }
        int c = codes[o + j];
        if (rank < k) c = mask_token;
        codes[o + j] = c;
        is_masked[o + j] = (c == mask_token);
    }
}

extern "C" int valle_mask_lowest(const float* scores, int32_t* codes, uint8_t* is_masked, const int32_t* off,
                                 const int32_t* k, int B, int max_len, int mask_token, void* stream) {
    VALLE_REQUIRE(scores && codes && is_masked && off && k, VALLE_E_ARG, "valle_mask_lowest: null pointer");
    VALLE_REQUIRE(max_len >= 0 && max_len <= MASK_MAXN, VALLE_E_UNSUPPORTED, "valle_mask_lowest: %d positions per utterance (max %d)", max_len, MASK_MAXN);
    if (B == 0) return VALLE_OK;
    mask_lowest_kernel<<<B, 256, 0, (hipStream_t)stream>>>(scores, codes, is_masked, off, k, mask_token);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// ---------------------------------------------------------------------- host-driven AR step: sample + append
extern "C" int valle_sample_append(const float* logits, int ld, int vocab, int B, const valle_sample_params* params,
                                   int32_t* seq, int seq_cap, int32_t* seq_len, int32_t* ctx_len, int32_t* ctx_len2,
                                   int32_t* stopped, int32_t* ids, float* scores, void* stream) {
    VALLE_REQUIRE(logits && params && seq && seq_len && stopped && ids, VALLE_E_ARG, "valle_sample_append: null pointer");
    VALLE_REQUIRE(vocab > 1 && vocab <= SAMP_MAXV && ld >= vocab && seq_cap > 0, VALLE_E_ARG, "valle_sample_append: vocab %d, ld %d, seq_cap %d", vocab, ld, seq_cap);
    VALLE_REQUIRE(params->rep_pen_passes >= 1 && params->rep_pen_passes <= 2 && params->filter_passes >= 0 && params->filter_passes <= 2,
                  VALLE_E_ARG, "valle_sample_append: rep_pen_passes / filter_passes out of range");
    if (B == 0) return VALLE_OK;
    SampleArgs a{};
    a.logits = logits; a.ld = ld; a.vocab = vocab; a.nar_mode = 0; a.p = *params;
    a.ids = ids; a.scores = scores;
    a.seq = seq; a.seq_cap = seq_cap; a.seq_len = seq_len; a.ctx_len = ctx_len; a.ctx_len2 = ctx_len2; a.stopped = stopped;
    a.prev_is_seq = 1;
    return valle_sample_launch(a, B, (hipStream_t)stream);
}

// ---------------------------------------------------------------------- beam search (models/base.py:1412-1415)
// top-W of the flattened [rows][vocab] processed logits (samplers.py:148-160), best first, ties to the lower flat index;
// candidate i = (row, token) gives ids[i] = token, scores[i] = its logit, and forced[i][..] = -inf except 0 at the token,
// so that a greedy valle_sample_append on `forced` appends token i to sequence i -- the reference's (naive) re-batching
// hands candidate i to sequence i whichever row it came from (models/ar_nar.py:837-842).
__global__ __launch_bounds__(256) void beam_pick_kernel(float* __restrict__ proc, int ld, int V, int rows, int W,
                                                        float* __restrict__ forced, int32_t* __restrict__ ids,
                                                        float* __restrict__ scores) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const int tid = threadIdx.x;
    for (int i = 0; i < W; ++i) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int t = tid; t < rows * V; t += 256) {
            const float x = proc[(size_t)(t / V) * ld + (t % V)];
            if (x > best || (x == best && t < bi)) { best = x; bi = t; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { s_val[tid >> 6] = best; s_idx[tid >> 6] = bi; }
        __syncthreads();
        best = s_val[0]; bi = s_idx[0];
        for (int w = 1; w < 4; ++w) if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) { best = s_val[w]; bi = s_idx[w]; }
        if (bi == 0x7fffffff) bi = 0;                       // a matrix of NaNs compares false everywhere
        const int tok = bi % V;
        if (tid == 0) { ids[i] = tok; if (scores) scores[i] = best; proc[(size_t)(bi / V) * ld + tok] = -INFINITY; }
        for (int v = tid; v < V; v += 256) forced[(size_t)i * ld + v] = (v == tok) ? 0.f : -INFINITY;
        __syncthreads();
    }
}

extern "C" int valle_beam_pick(const float* logits, int ld, int vocab, int rows, int beam_width,
                               const valle_sample_params* params, const int32_t* seq, int seq_cap, const int32_t* seq_len,
                               float* proc, float* forced, int32_t* ids, float* scores, void* stream) {
    VALLE_REQUIRE(logits && params && seq && seq_len && proc && forced && ids, VALLE_E_ARG, "valle_beam_pick: null pointer");
    VALLE_REQUIRE(vocab > 1 && vocab <= SAMP_MAXV && ld >= vocab && rows >= 1 && beam_width >= 1 && beam_width <= rows * vocab,
                  VALLE_E_ARG, "valle_beam_pick: vocab %d, ld %d, rows %d, beam_width %d", vocab, ld, rows, beam_width);
    SampleArgs a{};
    a.logits = logits; a.ld = ld; a.vocab = vocab; a.nar_mode = 0; a.p = *params;
    a.ids = ids; a.seq = const_cast<int32_t*>(seq); a.seq_cap = seq_cap; a.seq_len = const_cast<int32_t*>(seq_len);
    a.prev_is_seq = 1; a.proc_out = proc;
    int rc = valle_sample_launch(a, rows, (hipStream_t)stream);
    if (rc != VALLE_OK) return rc;
    beam_pick_kernel<<<1, 256, 0, (hipStream_t)stream>>>(proc, ld, vocab, rows, beam_width, forced, ids, scores);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

// ---------------------------------------------------------------------- integer bookkeeping of the NAR / NAR-len loops
template <typename D>
__global__ __launch_bounds__(256) void scatter_cols_kernel(D* __restrict__ dst, int ld_dst, const int32_t* __restrict__ rows,
                                                           const int32_t* __restrict__ src, int ld_src, int n, int c0, int s0,
                                                           int ncols, int add) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * ncols) return;
    const int i = t / ncols, c = t - i * ncols;
    const int r = rows ? rows[i] : i;
    dst[(size_t)r * ld_dst + c0 + c] = (D)(src[(size_t)i * ld_src + s0 + c] + add);
}

extern "C" int valle_scatter_cols(void* dst, int dst_is_i64, int ld_dst, const int32_t* rows, const int32_t* src, int ld_src,
                                  int n, int c0, int s0, int ncols, int add, void* stream) {
    VALLE_REQUIRE(dst && src, VALLE_E_ARG, "valle_scatter_cols: null pointer");
    VALLE_REQUIRE(n >= 0 && ncols >= 0 && c0 >= 0 && s0 >= 0 && c0 + ncols <= ld_dst && s0 + ncols <= ld_src, VALLE_E_ARG,
                  "valle_scatter_cols: columns out of range (c0 %d + %d of %d, s0 %d + %d of %d)", c0, ncols, ld_dst, s0, ncols, ld_src);
    if (n == 0 || ncols == 0) return VALLE_OK;
    const int grid = ceil_div(n * ncols, 256);
    if (dst_is_i64) scatter_cols_kernel<long long><<<grid, 256, 0, (hipStream_t)stream>>>((long long*)dst, ld_dst, rows, src, ld_src, n, c0, s0, ncols, add);
    else scatter_cols_kernel<int32_t><<<grid, 256, 0, (hipStream_t)stream>>>((int32_t*)dst, ld_dst, rows, src, ld_src, n, c0, s0, ncols, add);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}

__global__ __launch_bounds__(256) void narlen_commit_kernel(int32_t* __restrict__ codes, float* __restrict__ scores,
                                                            const uint8_t* __restrict__ is_masked, const int32_t* __restrict__ ids,
                                                            const float* __restrict__ sc, int n, int masked_only) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const bool m = is_masked[j] != 0;
    if (m) codes[j] = ids[j];                                        // ar_nar.py:457: unmasked positions keep their token
    scores[j] = masked_only ? (m ? sc[j] : 1.0f) : sc[j];            // ar_nar.py:459-462
}

extern "C" int valle_narlen_commit(int32_t* codes, float* scores, const uint8_t* is_masked, const int32_t* ids, const float* sc,
                                   int n, int masked_only, void* stream) {
    VALLE_REQUIRE(codes && scores && is_masked && ids && sc, VALLE_E_ARG, "valle_narlen_commit: null pointer");
    if (n <= 0) return VALLE_OK;
    narlen_commit_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(codes, scores, is_masked, ids, sc, n, masked_only);
    VALLE_LAUNCH_CHECK();
    return VALLE_OK;
}
