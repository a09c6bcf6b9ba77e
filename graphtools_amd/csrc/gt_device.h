// Device-side helpers for gfx950 (wave64): order-preserving float keys, wave-level
// bitonic sorting networks held in registers, wave reductions.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GT_WAVE 64

// ---- order-preserving 32-bit image of a float (larger float <=> larger unsigned) --------------
__device__ __forceinline__ uint32_t f32_ord(float v) {
    uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_f32(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}
// candidate key: high word = score (bigger = closer), low word = ~index so that, for equal scores,
// the smaller index sorts first in a descending sort.  A real key is never 0.
__device__ __forceinline__ uint64_t cand_pack(float score, uint32_t idx) {
    return (uint64_t(f32_ord(score)) << 32) | uint64_t(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ uint32_t cand_index(uint64_t key) { return 0xFFFFFFFFu - uint32_t(key); }
__device__ __forceinline__ float cand_score(uint64_t key) { return ord_f32(uint32_t(key >> 32)); }

// agent-scope (L2-served) 8-byte accesses: candidate lists are written and re-read by different
// lanes of the same wave at different times; going through L2 on both sides keeps that coherent
// without any dependence on the per-CU L1 (MI355X_MICROARCH: L1 is never refreshed by stores).
__device__ __forceinline__ void st_agent_u64(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_agent_u64(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- wave-level bitonic sort, descending, NT keys per lane; element e = t*64 + lane -----------
// (32-bit keys: one shuffle, v_max_u32 / v_min_u32 and a select per step - half the work of a 64-bit key)
template <int NT, typename K>
__device__ __forceinline__ void wave_bitonic_desc(K (&key)[NT], const int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * NT; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int jj = j >> 6;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if ((t & jj) == 0) {
                        const int u = t | jj;
                        const bool desc = (((t << 6) & k) == 0);
                        const K a = key[t], b = key[u];
                        const K mx = a > b ? a : b, mn = a > b ? b : a;
                        key[t] = desc ? mx : mn;
                        key[u] = desc ? mn : mx;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const K a = key[t];
                    K o;
                    if constexpr (sizeof(K) == 8) o = K(__shfl_xor((unsigned long long)a, j));
                    else o = K(__shfl_xor((unsigned int)a, j));
                    const int e = (t << 6) | lane;
                    const bool desc = ((e & k) == 0);
                    const bool lower = ((lane & j) == 0);
                    const bool want_max = (lower == desc);
                    const K mx = a > o ? a : o, mn = a > o ? o : a;
                    key[t] = want_max ? mx : mn;
                }
            }
        }
    }
}

// ---- wave-level bitonic sort, ascending on the 128-bit pair (hi, lo) ---------------------------
__device__ __forceinline__ bool pair_gt(uint64_t ah, uint64_t al, uint64_t bh, uint64_t bl) {
    return (ah > bh) || (ah == bh && al > bl);
}
template <int NT>
__device__ __forceinline__ void wave_bitonic_asc_pair(uint64_t (&hi)[NT], uint64_t (&lo)[NT], const int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * NT; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int jj = j >> 6;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if ((t & jj) == 0) {
                        const int u = t | jj;
                        const bool asc = (((t << 6) & k) == 0);
                        const uint64_t ah = hi[t], al = lo[t], bh = hi[u], bl = lo[u];
                        const bool a_gt = pair_gt(ah, al, bh, bl);
                        const bool swap = (a_gt == asc);
                        hi[t] = swap ? bh : ah;
                        lo[t] = swap ? bl : al;
                        hi[u] = swap ? ah : bh;
                        lo[u] = swap ? al : bl;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const uint64_t ah = hi[t], al = lo[t];
                    const uint64_t oh = __shfl_xor((unsigned long long)ah, j);
                    const uint64_t ol = __shfl_xor((unsigned long long)al, j);
                    const int e = (t << 6) | lane;
                    const bool asc = ((e & k) == 0);
                    const bool lower = ((lane & j) == 0);
                    const bool want_min = (lower == asc);
                    const bool a_gt = pair_gt(ah, al, oh, ol);
                    // want_min: keep the smaller of (a, o); else keep the larger
                    const bool take_other = (want_min == a_gt);
                    hi[t] = take_other ? oh : ah;
                    lo[t] = take_other ? ol : al;
                }
            }
        }
    }
}

// ---- wave reductions ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// exclusive prefix count of a predicate within the wave + total
__device__ __forceinline__ int wave_prefix_count(bool pred, int lane, int& total) {
    const unsigned long long m = __ballot(pred);
    total = __popcll(m);
    return __popcll(m & ((1ull << lane) - 1ull));
}
