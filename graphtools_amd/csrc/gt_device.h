// Device-side helpers for gfx950 (wave64): order-preserving float keys, wave-level
// bitonic sorting networks held in registers, wave reductions.
#pragma once
#include <type_traits>
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
// ---- lane exchanges without the LDS -------------------------------------------------------------
// value of lane (lane ^ J): DPP row operations inside a row of 16 lanes (quad_perm for 1 and 2, half-mirror + quad reversal
// for 4, a rotation by 8 for 8), the gfx950 row / half swaps for 16 and 32.  ds_bpermute (what __shfl_xor compiles to) takes a
// trip through the LDS pipeline with an s_waitcnt per step of a sorting network; these are VALU moves.
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(const uint32_t v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "one lane bit");
    if constexpr (J == 1) return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0xB1, 0xF, 0xF, true));
    else if constexpr (J == 2) return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x4E, 0xF, 0xF, true));
    else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, int(v), 0x141, 0xF, 0xF, true);   // lane ^ 7
        return uint32_t(__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, true));        // ... ^ 3
    } else if constexpr (J == 8) return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x128, 0xF, 0xF, true));
    else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (__lane_id() & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (__lane_id() & 32) ? r[0] : r[1];
    }
}
template <int J>
__device__ __forceinline__ uint64_t lane_xor_u64(const uint64_t v) {
    return uint64_t(lane_xor_u32<J>(uint32_t(v))) | (uint64_t(lane_xor_u32<J>(uint32_t(v >> 32))) << 32);
}
template <int J>
__device__ __forceinline__ uint64_t lane_xor_any(const uint64_t v) { return lane_xor_u64<J>(v); }
template <int J>
__device__ __forceinline__ uint32_t lane_xor_any(const uint32_t v) { return lane_xor_u32<J>(v); }
// J = 16 or 32: the pair of values the two partner lanes hold, the same in both - lo from the lane with bit J clear
template <int J, typename K>
__device__ __forceinline__ void lane_pair_values(const K v, K& lo, K& hi) {
    static_assert(J == 16 || J == 32, "row / half swaps");
    if constexpr (sizeof(K) == 8) {
        const uint32_t v0 = uint32_t(uint64_t(v)), v1 = uint32_t(uint64_t(v) >> 32);
        if constexpr (J == 16) {
            const auto r0 = __builtin_amdgcn_permlane16_swap(v0, v0, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(v1, v1, false, false);
            lo = K(uint64_t(r0[0]) | (uint64_t(r1[0]) << 32));
            hi = K(uint64_t(r0[1]) | (uint64_t(r1[1]) << 32));
        } else {
            const auto r0 = __builtin_amdgcn_permlane32_swap(v0, v0, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(v1, v1, false, false);
            lo = K(uint64_t(r0[0]) | (uint64_t(r1[0]) << 32));
            hi = K(uint64_t(r0[1]) | (uint64_t(r1[1]) << 32));
        }
    } else {
        if constexpr (J == 16) {
            const auto r = __builtin_amdgcn_permlane16_swap(uint32_t(v), uint32_t(v), false, false);
            lo = K(r[0]);
            hi = K(r[1]);
        } else {
            const auto r = __builtin_amdgcn_permlane32_swap(uint32_t(v), uint32_t(v), false, false);
            lo = K(r[0]);
            hi = K(r[1]);
        }
    }
}

// ---- wave-level bitonic sort, descending, NT keys per lane; element e = t*64 + lane -----------
// (32-bit keys: one exchange, one compare and a select per step - half the work of a 64-bit key)
template <int J, int NT, typename K>
__device__ __forceinline__ void wave_bitonic_desc_step(K (&key)[NT], const int lane, const int k) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const K a = key[t];
        const int e = (t << 6) | lane;
        const bool desc = ((e & k) == 0);
        if constexpr (J >= 16) {
            // both partners see the same (lo, hi): one compare decides for both.  The lower lane keeps the larger key when
            // the run descends; equal keys are the same value either way
            K lo, hi;
            lane_pair_values<J>(a, lo, hi);
            const bool upper = (lane & J) != 0;
            key[t] = (((lo > hi) == desc) != upper) ? lo : hi;
        } else {
            K o;
            if constexpr (sizeof(K) == 8) o = K(lane_xor_u64<J>(uint64_t(a)));
            else o = K(lane_xor_u32<J>(uint32_t(a)));
            const bool lower = ((lane & J) == 0);
            const bool want_max = (lower == desc);
            // keep the own key when it is the one wanted: one compare, the lane-pattern mask folded in on the scalar
            // side, one select (instead of max, min and a select between them)
            key[t] = ((a > o) == want_max) ? a : o;
        }
    }
}
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
                switch (j) {   // (constant after unrolling)
                    case 32: wave_bitonic_desc_step<32>(key, lane, k); break;
                    case 16: wave_bitonic_desc_step<16>(key, lane, k); break;
                    case 8: wave_bitonic_desc_step<8>(key, lane, k); break;
                    case 4: wave_bitonic_desc_step<4>(key, lane, k); break;
                    case 2: wave_bitonic_desc_step<2>(key, lane, k); break;
                    default: wave_bitonic_desc_step<1>(key, lane, k); break;
                }
            }
        }
    }
}

// ---- wave-level bitonic sort, ascending on the 128-bit pair (hi, lo) ---------------------------
template <typename L>
__device__ __forceinline__ bool pair_gt(uint64_t ah, L al, uint64_t bh, L bl) {
    return (ah > bh) || (ah == bh && al > bl);
}
template <int NT, typename L>   // L: uint64_t or uint32_t
__device__ __forceinline__ void wave_bitonic_asc_pair(uint64_t (&hi)[NT], L (&lo)[NT], const int lane) {
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
                        const uint64_t ah = hi[t], bh = hi[u];
                        const L al = lo[t], bl = lo[u];
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
                    const uint64_t ah = hi[t];
                    const L al = lo[t];
                    uint64_t oh;
                    L ol;
                    switch (j) {   // (constant after unrolling)
                        case 32: oh = lane_xor_u64<32>(ah); ol = lane_xor_any<32>(al); break;
                        case 16: oh = lane_xor_u64<16>(ah); ol = lane_xor_any<16>(al); break;
                        case 8: oh = lane_xor_u64<8>(ah); ol = lane_xor_any<8>(al); break;
                        case 4: oh = lane_xor_u64<4>(ah); ol = lane_xor_any<4>(al); break;
                        case 2: oh = lane_xor_u64<2>(ah); ol = lane_xor_any<2>(al); break;
                        default: oh = lane_xor_u64<1>(ah); ol = lane_xor_any<1>(al); break;
                    }
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
// x / lds_x (optional, with the LDS slots): a 64-bit payload per entry that follows its entry (NT values per lane).
__device__ __forceinline__ float wave_max_f32(float v);
// Q32: the composite keys of the network are 32 bits - the key as a 23 / 24-bit fixed-point fraction of the row's own key range
// (float32 minimum and maximum of the wave, widened) above the position bits - instead of the key's own upper bits in 64: half
// the instructions per exchange.  The mapping is monotone, so only entries that fall into the same step of the range (and
// true ties) can come out in the wrong order: they are detected below and the row takes the exact pair network.
template <int NT, typename L, bool Q32 = false>
__device__ __forceinline__ void wave_sort_asc_pair_fast(uint64_t (&hi)[NT], L (&lo)[NT], const int lane,
                                                        uint64_t* lds_hi = nullptr, uint32_t* lds_lo = nullptr,
                                                        uint64_t* x = nullptr, uint64_t* lds_x = nullptr) {
    static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8, "up to 512 entries");
    constexpr int PB = NT == 1 ? 6 : NT == 2 ? 7 : NT == 4 ? 8 : 9;
    constexpr uint64_t PM = (1ull << PB) - 1ull;
    constexpr uint64_t kNoneHi = 0x7FF0000000000000ull;
    using CK = typename std::conditional<Q32, uint32_t, uint64_t>::type;
    CK ck[NT];
    if constexpr (Q32) {
        constexpr int QB = 32 - PB;   // bits of the fraction
        float fmin_ = INFINITY, fmax_ = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool none = hi[t] == kNoneHi && lo[t] == L(0xFFFFFFFFu);
            const float f = float(__longlong_as_double((long long)hi[t]));   // (keys are non-negative)
            if (!none) {
                fmin_ = fminf(fmin_, f);
                fmax_ = fmaxf(fmax_, f);
            }
        }
        fmin_ = -wave_max_f32(-fmin_);
        fmax_ = wave_max_f32(fmax_);
        const double base = double(fmin_) * (1.0 - 0x1p-20);
        const double top = double(fmax_) * (1.0 + 0x1p-20);
        const double span = top - base;
        const double scale = (span > 0.0 && span < INFINITY) ? double(1u << QB) / span : 0.0;
        constexpr uint32_t qmax = (1u << QB) - 2u;   // (all ones with the last position is the complement of the "no entry" key)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool none = hi[t] == kNoneHi && lo[t] == L(0xFFFFFFFFu);
            const double rel = (__longlong_as_double((long long)hi[t]) - base) * scale;
            uint32_t q = rel > 0.0 ? (rel < double(qmax) ? uint32_t(rel) : qmax) : 0u;
            ck[t] = none ? 0u : ~((q << PB) | uint32_t(t * 64 + lane));   // descending complement = ascending key
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool none = hi[t] == kNoneHi && lo[t] == L(0xFFFFFFFFu);
            ck[t] = none ? 0ull : ~((hi[t] & ~PM) | uint64_t(t * 64 + lane));   // descending complement = ascending key
        }
    }
    wave_bitonic_desc<NT>(ck, lane);
    bool clash = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        CK nx;
        if constexpr (Q32) nx = __shfl_down(ck[t], 1);
        else nx = __shfl_down((unsigned long long)ck[t], 1);
        CK edge = 0;
        if constexpr (Q32) edge = (t < NT - 1) ? __shfl(ck[t < NT - 1 ? t + 1 : t], 0) : 0u;
        else edge = (t < NT - 1) ? __shfl((unsigned long long)ck[t < NT - 1 ? t + 1 : t], 0) : 0ull;
        if (lane == 63) nx = edge;
        clash |= ck[t] != 0 && nx != 0 && (CK(~ck[t]) >> PB) == (CK(~nx) >> PB);
    }
    if (__ballot(clash) != 0ull) {   // wave-uniform, rare
        if (x) {
            // the payloads are parked by original position next to their entries' lo; after the pair network every slot
            // looks its payload up by lo (unique within a row: a database row is a candidate once)
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                lds_lo[r * 64 + lane] = uint32_t(lo[r]);
                lds_x[r * 64 + lane] = x[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        wave_bitonic_asc_pair<NT>(hi, lo, lane);
        if (x) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint64_t found = 0ull;
                const uint32_t want = uint32_t(lo[t]);
                if (want != 0xFFFFFFFFu)
                    for (int q = 0; q < 64 * NT; ++q)
                        if (lds_lo[q] == want) found = lds_x[q];
                x[t] = found;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    uint64_t nh[NT];
    uint32_t nl[NT];
    if (lds_hi) {
        // the wave's own 64 NT (hi, lo) slots in the LDS: every entry is parked at its position, every sorted slot picks
        // its entry up (2 NT writes and reads instead of 3 NT^2 lane permutes; LDS operations of one wave stay in order)
#pragma unroll
        for (int r = 0; r < NT; ++r) {
            lds_hi[r * 64 + lane] = hi[r];
            lds_lo[r * 64 + lane] = uint32_t(lo[r]);
            if (x) lds_x[r * 64 + lane] = x[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint32_t p = uint32_t(~ck[t]) & uint32_t(PM);
            const bool some = ck[t] != 0;
            nh[t] = some ? lds_hi[p] : kNoneHi;
            nl[t] = some ? lds_lo[p] : 0xFFFFFFFFu;
            if (x) x[t] = some ? lds_x[p] : 0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
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
                if (sr == r && ck[t] != 0) {
                    h = hr;
                    l = lr;
                }
            }
            nh[t] = h;
            nl[t] = l;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        hi[t] = nh[t];
        lo[t] = L(nl[t]);
    }
}

// ---- workgroup -> work mapping over the 8 XCDs -------------------------------------------------
// Workgroups are dealt to the XCDs round robin (blockIdx & 7).  Work item b of `nb`, where neighbouring items touch the same
// rows: XCD x takes chunk x of every run of 8 chunks of `chunk` consecutive items - the neighbours share an L2, and all XCDs
// walk the same region of the (cell-sorted) point set at the same time, so that regions whose rows cost more (longer
// candidate lists) are spread over the whole chip instead of landing on one XCD.  chunk <= 0: one contiguous eighth each.
__device__ __forceinline__ int64_t gt_xcd_item(const int64_t bidx, const int64_t nb, const int chunk) {
    int64_t start = 0, n = nb, b = bidx;
    if (chunk > 0) {
        const int64_t span = int64_t(8) * chunk, whole = nb / span * span;
        if (bidx < whole) {
            const int64_t g = bidx >> 3;
            return (g / chunk) * span + (bidx & 7) * chunk + g % chunk;
        }
        start = whole;
        n = nb - whole;
        b = bidx - whole;
    }
    const int64_t xcd = b & 7, base = n >> 3, rem = n & 7;
    return start + xcd * base + (xcd < rem ? xcd : rem) + (b >> 3);
}

// ---- wave reductions ------------------------------------------------------------------------
__device__ __forceinline__ double lane_xor_f64(const double v, const int o) {   // o: a constant after unrolling
    const uint64_t b = uint64_t(__double_as_longlong(v));
    uint64_t r;
    switch (o) {
        case 32: r = lane_xor_u64<32>(b); break;
        case 16: r = lane_xor_u64<16>(b); break;
        case 8: r = lane_xor_u64<8>(b); break;
        case 4: r = lane_xor_u64<4>(b); break;
        case 2: r = lane_xor_u64<2>(b); break;
        default: r = lane_xor_u64<1>(b); break;
    }
    return __longlong_as_double((long long)r);
}
__device__ __forceinline__ uint32_t lane_xor_b32(const uint32_t v, const int o) {
    switch (o) {
        case 32: return lane_xor_u32<32>(v);
        case 16: return lane_xor_u32<16>(v);
        case 8: return lane_xor_u32<8>(v);
        case 4: return lane_xor_u32<4>(v);
        case 2: return lane_xor_u32<2>(v);
        default: return lane_xor_u32<1>(v);
    }
}
// PRECONDITION of every helper built on lane_xor_* (these reductions, the bitonic networks): the whole wave executes the call
// CONVERGED - all 64 lanes active.  The DPP forms zero-fill what an inactive partner lane would have delivered (bound_ctrl) and
// the permlane swaps leave an inactive lane's value in place, so a reduction inside a divergent branch silently drops addends.
// Kernels therefore take their early exits per WAVE (`if (row >= n) return;` with one row per wave) or mask inside the call
// (zero addends), never per lane around it.  gt_dbg_lane_ops (gt_debug.hip) checks every form against __shfl_xor on the GPU.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += lane_xor_f64(v, o);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __uint_as_float(lane_xor_b32(__float_as_uint(v), o)));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += int(lane_xor_b32(uint32_t(v), o));
    return v;
}
// exclusive prefix count of a predicate within the wave + total
__device__ __forceinline__ int wave_prefix_count(bool pred, int lane, int& total) {
    const unsigned long long m = __ballot(pred);
    total = __popcll(m);
    return __popcll(m & ((1ull << lane) - 1ull));
}
