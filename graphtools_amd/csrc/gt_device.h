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

// ---- canonical float64 dot product of the exact stages -----------------------------------------------------------
// Every exact squared distance is (|x|^2 + (-2 x.y)) + |y|^2 with x.y and the norms summed in ONE fixed order, so that
// the distance of a pair is the same bits wherever it is evaluated (either row's re-rank, the repairs, the radius rows)
// and a row's distance to itself is exactly 0.  The order: 16 partial sums - element k goes to sum (k >> 2) & 15, in
// increasing k, one fma each - combined by the tree (l, l + 8), (l, l + 4), (l, l + 2), (0, 1).  Sixteen independent
// accumulators instead of one dependent fma chain per row; it is also the order a group of 16 lanes produces when lane l
// takes the elements 4 l ... 4 l + 3 of every 64 (one 16-byte load per lane) and sums by rotations of 8, 4, 2, 1 lanes.
__device__ __forceinline__ double gt_tree16(const double (&a)[16]) {
    double b[8], c[4];
#pragma unroll
    for (int l = 0; l < 8; ++l) b[l] = a[l] + a[l + 8];
#pragma unroll
    for (int l = 0; l < 4; ++l) c[l] = b[l] + b[l + 4];
    return (c[0] + c[2]) + (c[1] + c[3]);
}
template <typename TX, typename TY>
__device__ __forceinline__ double gt_dot16(const TX* __restrict__ x, const TY* __restrict__ y, const int d) {
    double a[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) a[l] = 0.0;
    int k0 = 0;
    // (general path - rows that are not 16-byte aligned float32: the scheduling barriers keep the compiler from hoisting
    //  all 64 loads of a group in front of the arithmetic, which would cost the kernels around it their occupancy)
    for (; k0 + 64 <= d; k0 += 64) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = k0 + 4 * q;
            a[q] = fma(double(x[k + 0]), double(y[k + 0]), a[q]);
            a[q] = fma(double(x[k + 1]), double(y[k + 1]), a[q]);
            a[q] = fma(double(x[k + 2]), double(y[k + 2]), a[q]);
            a[q] = fma(double(x[k + 3]), double(y[k + 3]), a[q]);
            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (k0 < d) {
        // the last, partial group of 64: whole quadruples without a test per element (d is uniform: so are the branches)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = k0 + 4 * q;
            if (k + 4 <= d) {
                a[q] = fma(double(x[k + 0]), double(y[k + 0]), a[q]);
                a[q] = fma(double(x[k + 1]), double(y[k + 1]), a[q]);
                a[q] = fma(double(x[k + 2]), double(y[k + 2]), a[q]);
                a[q] = fma(double(x[k + 3]), double(y[k + 3]), a[q]);
            } else if (k < d) {
                for (int e = 0; k + e < d; ++e) a[q] = fma(double(x[k + e]), double(y[k + e]), a[q]);
            }
            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    return gt_tree16(a);
}
// the same bits with a handful of live registers (one partial sum at a time, the tree folded as they come): for rare
// paths inside kernels whose occupancy matters more
template <typename TX, typename TY>
__device__ __forceinline__ double gt_dot16_lean(const TX* __restrict__ x, const TY* __restrict__ y, const int d) {
    auto part = [&](const int q) {
        double s = 0.0;
        for (int k = 4 * q; k < d; k += 64) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < d) s = fma(double(x[k + e]), double(y[k + e]), s);
        }
        return s;
    };
    auto c = [&](const int l) { return (part(l) + part(l + 8)) + (part(l + 4) + part(l + 12)); };
    const double e0 = c(0) + c(2);
    const double e1 = c(1) + c(3);
    return e0 + e1;
}
// the same with 16-byte loads of a float32 row (d a multiple of 4, y 16-byte aligned)
template <typename TX>
__device__ __forceinline__ double gt_dot16_f4(const TX* __restrict__ x, const float* __restrict__ y, const int d) {
    double a[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) a[l] = 0.0;
    const float4* y4 = reinterpret_cast<const float4*>(y);
    int k0 = 0;
    for (; k0 + 64 <= d; k0 += 64) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = y4[(k0 >> 2) + q];
            a[q] = fma(double(x[k0 + 4 * q + 0]), double(v.x), a[q]);
            a[q] = fma(double(x[k0 + 4 * q + 1]), double(v.y), a[q]);
            a[q] = fma(double(x[k0 + 4 * q + 2]), double(v.z), a[q]);
            a[q] = fma(double(x[k0 + 4 * q + 3]), double(v.w), a[q]);
        }
    }
    if (k0 < d) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (k0 + 4 * q < d) {
                const float4 v = y4[(k0 >> 2) + q];
                a[q] = fma(double(x[k0 + 4 * q + 0]), double(v.x), a[q]);
                a[q] = fma(double(x[k0 + 4 * q + 1]), double(v.y), a[q]);
                a[q] = fma(double(x[k0 + 4 * q + 2]), double(v.z), a[q]);
                a[q] = fma(double(x[k0 + 4 * q + 3]), double(v.w), a[q]);
            }
    }
    return gt_tree16(a);
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
                    // keep the own key when it is the one wanted: one compare, the lane-pattern mask folded in on the
                    // scalar side, one select (instead of max, min and a select between them)
                    key[t] = ((a > o) == want_max) ? a : o;
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

// The same result through ONE 64-bit key per entry: the low bits of hi (a float64 bit pattern) give way to the entry's
// position, the network moves and compares one word instead of a 128-bit pair, and the pairs are fetched by position at
// the end.  Two entries whose hi agree above those bits could come out in the wrong order: the sorted sequence is checked
// for such neighbours and, when there is one (exact ties, distances within 2^-44 of each other), the pair network
// runs instead - so the outcome is always the one of wave_bitonic_asc_pair.  Entries (kInf, 0xFFFFFFFF) mean "none" and
// sort last; lo must fit 32 bits.
template <int NT>
__device__ __forceinline__ void wave_sort_asc_pair_fast(uint64_t (&hi)[NT], uint64_t (&lo)[NT], const int lane) {
    static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8, "up to 512 entries");
    constexpr int PB = NT == 1 ? 6 : NT == 2 ? 7 : NT == 4 ? 8 : 9;
    constexpr uint64_t PM = (1ull << PB) - 1ull;
    constexpr uint64_t kNoneHi = 0x7FF0000000000000ull;
    uint64_t ck[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bool none = hi[t] == kNoneHi && lo[t] == 0xFFFFFFFFull;
        ck[t] = none ? 0ull : ~((hi[t] & ~PM) | uint64_t(t * 64 + lane));   // descending complement = ascending key
    }
    wave_bitonic_desc<NT>(ck, lane);
    bool clash = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        uint64_t nx = __shfl_down((unsigned long long)ck[t], 1);
        const uint64_t edge = (t < NT - 1) ? __shfl((unsigned long long)ck[t < NT - 1 ? t + 1 : t], 0) : 0ull;
        if (lane == 63) nx = edge;
        clash |= ck[t] != 0ull && nx != 0ull && ((~ck[t]) >> PB) == ((~nx) >> PB);
    }
    if (__ballot(clash) != 0ull) {   // wave-uniform, rare
        wave_bitonic_asc_pair<NT>(hi, lo, lane);
        return;
    }
    uint64_t nh[NT];
    uint32_t nl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t p = uint32_t(~ck[t]) & uint32_t(PM);
        const int sl = int(p & 63u), sr = int(p >> 6);
        uint64_t h = kNoneHi;
        uint32_t l = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < NT; ++r) {
            const uint64_t hr = __shfl((unsigned long long)hi[r], sl);
            const uint32_t lr = __shfl(uint32_t(lo[r]), sl);
            if (sr == r && ck[t] != 0ull) {
                h = hr;
                l = lr;
            }
        }
        nh[t] = h;
        nl[t] = l;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        hi[t] = nh[t];
        lo[t] = uint64_t(nl[t]);
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
