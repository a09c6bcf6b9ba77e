// Exact dense graph (TraditionalGraph.build_kernel, graphtools/graphs.py:1514-1610, + symmetrise,
// anisotropy and P from graphtools/base.py:557-592, 629-646) for one GPU.
//
//   D      : from data  -> float64 difference form sqrt(sum_k (x_ik - x_jk)^2), sequential in k (scipy pdist)
//            precomputed -> the caller's n x n matrix, dtype preserved
//   bw_i   : (knn+1)-th smallest entry of row i (self's 0 included) * scale, or the user bandwidth * scale
//   K0_ij  : exp(-(D_ij / bw_i)^decay), NaN -> 1, < thresh -> 0      (arithmetic in D's dtype, like numpy)
//   K      : symmetrised tile pair by tile pair (both (I,J) and (J,I) are produced by one workgroup, so the
//            update can be done IN PLACE on D), then anisotropy, then P = K / rowsum.
//
// HBM-bound: per element 1 read + 1 write for K, 1-2 reads + 1 write for P (row sums re-read the row while
// it is L2 resident).  The from-data variant adds 3*d float64 flops per element on the vector pipe.
#include <cfloat>
#include <memory>
#include <type_traits>

#include "gt_common.h"
#include "gt_hostcopy.h"
#include "gt_device.h"
#include "gt_knn.h"

namespace {

constexpr int TS = 64;        // tile side
constexpr int TSP = TS + 1;   // padded LDS row
constexpr int SG = 8;         // super-tile side (tiles)

// xcut: beyond this scaled distance the reference's result is exactly 0 - below `thresh` (then zeroed), or an
// underflow of exp in T - so the transcendental work (the compute bound of this otherwise HBM-bound kernel) is
// skipped for almost every entry of a dense matrix.  The host leaves a 1 % margin in the exponent (dense_xcut).
template <typename T>
__device__ __forceinline__ T affinity_t(T dist, T bw, T decay, T xcut) {
    // cheap pre-test without the division (xcut carries a 1 % margin in the exponent, the factor below costs 1e-5 of it)
    if (dist > bw * (xcut * T(1.00001))) return T(0);
    const T x = dist / bw;
    if (x > xcut) return T(0);
    T w;
    if constexpr (sizeof(T) == 4) {
        w = expf(-powf(x, decay));
    } else {
        w = exp(-pow(x, decay));
    }
    return (w != w) ? T(1) : w;
}

// unsymmetrised kernel entry of a precomputed matrix: 0 a distance (alpha-decay affinity), 1 an affinity (taken as is,
// graphs.py:1532-1536), 2 an adjacency (diagonal set to 1, graphs.py:1537-1545)
template <typename T>
__device__ __forceinline__ T k0_t(T v, T bwv, T decay, T xcut, int pass, bool on_diag) {
    // (pass is the same for the whole launch; bwv = the row's bandwidth, loaded once by the caller)
    if (pass == 0) return affinity_t<T>(v, bwv, decay, xcut);
    return (pass == 2 && on_diag) ? T(1) : v;
}

template <typename T>
__device__ __forceinline__ T merge_t(T a, T b, int symm, T theta) {
    switch (symm) {
        case GT_SYMM_ADD: return (a + b) / T(2);
        case GT_SYMM_MUL: return a * b;
        case GT_SYMM_MNN: return theta * (a < b ? a : b) + (T(1) - theta) * (a < b ? b : a);
        default: return a;
    }
}

// (row-streaming form of the '+' rule, below: the kept affinities of all rows as one flat list)
constexpr int ROWS_LDS_CAP = 1024;   // non-zeros of a row parked in LDS before the row's one reservation (beyond: direct appends)

struct TriList {
    uint32_t* i;
    uint32_t* j;
    float* a;
    unsigned long long* cursor;   // entries appended so far (may run past cap: the host checks)
    unsigned long long cap;
    // where a row's own entries sit (its one reservation): the write pass places them from here instead of reading the row again;
    // own_cnt = kNoOwnList: the row's entries are not in one piece (more of them than the LDS list holds)
    unsigned long long* own_base;
    uint32_t* own_cnt;
};
constexpr uint32_t kNoOwnList = 0xFFFFFFFFu;

// ---- bandwidth from a precomputed distance matrix: (knn+1)-th smallest of every row --------------
template <typename T, int KL>
__global__ __launch_bounds__(256) void dense_bandwidth_kernel(const T* __restrict__ D, const int64_t n, const int kth,
                                                              const double scale, double* __restrict__ bw) {
    __shared__ double cand[256 * KL];
    __shared__ int red[4];
    const int64_t i = blockIdx.x;
    const int tid = threadIdx.x;
    const T* row = D + i * n;
    double best[KL];
#pragma unroll
    for (int t = 0; t < KL; ++t) best[t] = INFINITY;
#define GT_BW_INSERT(V_)                                                                          \
    {                                                                                             \
        const double v_ = double(V_);                                                             \
        if (v_ < best[KL - 1]) {                                                                  \
            best[KL - 1] = v_;                                                                    \
            _Pragma("unroll") for (int t = KL - 1; t > 0; --t) {                                  \
                if (best[t] < best[t - 1]) {                                                      \
                    const double tmp = best[t];                                                   \
                    best[t] = best[t - 1];                                                        \
                    best[t - 1] = tmp;                                                            \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
    }
    // The scan is a pure HBM stream: 16-byte loads, four of them in flight per thread (one load at a time leaves the
    // kernel latency-bound at a tenth of the bandwidth).  Rows are 16-byte aligned when n is a multiple of the vector.
    constexpr int VW = 16 / int(sizeof(T));
    typedef T vecT __attribute__((ext_vector_type(VW)));
    if ((n % VW) == 0) {
        const vecT* rv = reinterpret_cast<const vecT*>(row);
        const int64_t nv = n / VW;
        for (int64_t j0 = 0; j0 < nv; j0 += 256 * 4) {
            vecT v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = j0 + u * 256 + tid;
                if (j < nv) {
                    v[u] = rv[j];
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e) v[u][e] = T(INFINITY);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < VW; ++e) GT_BW_INSERT(v[u][e]);
            }
        }
    } else {
        for (int64_t j = tid; j < n; j += 256) GT_BW_INSERT(row[j]);
    }
#undef GT_BW_INSERT
#pragma unroll
    for (int t = 0; t < KL; ++t) cand[tid * KL + t] = best[t];
    __syncthreads();
    // kth smallest (1-based) of the 256*KL candidates by bitwise search on the float64 pattern (values >= 0)
    unsigned long long v = 0ull;
    for (int b = 63; b >= 0; --b) {
        const unsigned long long trial = v | ((1ull << b) - 1ull);
        int c = 0;
        for (int e = tid; e < 256 * KL; e += 256) {
            const double x = cand[e];
            const unsigned long long key = (x > 0.0) ? (unsigned long long)__double_as_longlong(x) : 0ull;
            c += (key <= trial) ? 1 : 0;
        }
        c = wave_sum_i32(c);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = c;
        __syncthreads();
        const int total = red[0] + red[1] + red[2] + red[3];
        if (total < kth) v |= (1ull << b);
    }
    if (tid == 0) bw[i] = __longlong_as_double((long long)v) * scale;
}

// kth smallest of a row in ONE streaming pass (kth <= 256): the row is read once (the two-pass kernel of round 3 read its 800 KB
// at N = 2e5 twice from the HBM - 256 rows in flight per XCD do not stay in a 4 MB L2).
//   steps 0, 1   (the first 2 x 256 x 16 / sizeof(T) values) stay in registers, every thread takes the minimum of
//                its own; the kth smallest of the 256 minima is an upper bound `ub` of the answer with at least kth
//                stored values at or below it (the two-pass kernel's argument); the stored values <= ub (a few dozen) start
//                the candidate list;
//   later steps  append what is <= ub to the list (LDS atomic for the slot: a handful per step); every 8th step the list
//                is looked at - beyond half its capacity it is cut back to its kth smallest, which lowers ub;
//   the end      the kth smallest of the list is the row's: every value <= the final ub was appended while ub was at
//                least that large, and survives every cut.
// The selections run on the values' own bit patterns (32 search steps for float32) over a few values per thread - the
// first version searched 8192 LDS entries with 64-bit patterns and two barriers per step: 2.7 TB/s instead of 6.
// Rows with mass ties at the bound (list overflow) are flagged for the generic kernel.
#ifndef GT_DENSE_PREFETCH
#define GT_DENSE_PREFETCH 0
#endif
#ifndef GT_DENSE_EMIT_WAVES
#define GT_DENSE_EMIT_WAVES 6   // waves per SIMD the listing variant of the one-pass bandwidth kernel is cut for (80 VGPRs; 128 uncut)
#endif
struct EmitArgs {
    float rfac, decay, thresh, xcut;
    double* own_sum;
    TriList tl;
    uint32_t* flags;
    uint32_t* scan_rows;     // rows that need dense_rows_scan_kernel (their list overflowed here)
    uint32_t* scan_count;
};

template <typename T, bool EMIT>
__global__ __launch_bounds__(256, EMIT ? GT_DENSE_EMIT_WAVES : 1) void dense_bandwidth_1pass_kernel(const T* __restrict__ D, const int64_t n, const int kth,
                                                                    const double scale, double* __restrict__ bw,
                                                                    uint32_t* __restrict__ redo, const EmitArgs em) {
    // EMIT (row-streaming form of the '+' rule, float32): the list keeps, WITH their columns, every value within rfac x the
    // running bound - rfac = bandwidth scale x the scaled distance beyond which an affinity is zero (+ rounding room) - so that
    // when the row's bandwidth is known the row's kept affinities are all in the list: they are evaluated, summed (own half of
    // the row sum) and appended to the flat list the transposition reads - the matrix is not read a second time for them.
    // A row whose list overflows is redone by the generic bandwidth kernel and scanned on its own (em.scan_rows).
    constexpr int VW = 16 / int(sizeof(T));
    constexpr int PT = 4 * VW;                 // values per thread and step
    constexpr int CHUNK = 256 * PT;            // values per step of the workgroup
    constexpr int LCAP = EMIT ? 2048 : 1024;   // candidate list
    constexpr int BITS = int(sizeof(T)) * 8;
    typedef typename std::conditional<sizeof(T) == 4, uint32_t, unsigned long long>::type key_t;
    __shared__ T lst[LCAP];
    __shared__ uint32_t lcolv[EMIT ? LCAP : 1];
    __shared__ int red[2][4];
    __shared__ int cnt;
    const T rfac = EMIT ? T(em.rfac) : T(1);
#define GT_BW_COL(ST_, U_) (vec ? uint32_t(((ST_) * (256 * 4) + ((U_) / VW) * 256 + tid) * VW + (U_) % VW) \
                                : uint32_t((ST_) * CHUNK + int64_t(U_) * 256 + tid))
    const int64_t i = blockIdx.x;
    const int tid = threadIdx.x;
    const T* row = D + i * n;
    typedef T vecT __attribute__((ext_vector_type(VW)));
    const bool vec = (n % VW) == 0;
    const vecT* rv = reinterpret_cast<const vecT*>(row);
    const int64_t nv = n / VW;
    auto key_of = [](T x) -> key_t {   // order-preserving for x >= 0 (distances); anything <= 0 sorts first
        if (!(x > T(0))) return key_t(0);
        if constexpr (sizeof(T) == 4) return key_t(__float_as_uint(x));
        else return key_t((unsigned long long)__double_as_longlong(x));
    };
    // kth smallest key (1-based) among the keys k[0 .. E) of all threads (invalid slots: the all-ones key)
    auto select = [&](const key_t* k, const int E) -> key_t {
        key_t v = 0;
        int par = 0;
        for (int b = BITS - 1; b >= 0; --b, par ^= 1) {
            const key_t trial = v | ((key_t(1) << b) - key_t(1));
            int c = 0;
            for (int e = 0; e < E; ++e) c += (k[e] <= trial) ? 1 : 0;
            c = wave_sum_i32(c);
            if ((tid & 63) == 0) red[par][tid >> 6] = c;
            __syncthreads();   // (one barrier per step: the next step writes the other half of `red`)
            if (red[par][0] + red[par][1] + red[par][2] + red[par][3] < kth) v |= (key_t(1) << b);
        }
        return v;
    };
    const key_t kInvalid = ~key_t(0);
    if (tid == 0) cnt = 0;
    bool nan_seen = false;   // EMIT: a NaN distance is an affinity of 1 in the reference (graphs.py:1593) - such a row is scanned on its own
    T ub = T(INFINITY);
    T mn = T(INFINITY);
    T x0[PT];   // the values of step 0, held until the bound is known
#pragma unroll
    for (int u = 0; u < PT; ++u) x0[u] = T(INFINITY);
    bool bad = false;
    const int64_t steps = vec ? (nv + 256 * 4 - 1) / (256 * 4) : (n + CHUNK - 1) / CHUNK;
    // (GT_DENSE_PREFETCH: the loads of step st + 1 are issued before step st is looked at - a second set of 16 registers; the
    //  workgroup's 16 KB per step then overlap its own list work instead of leaving that to the other workgroups of the CU)
    auto load_step = [&](const int64_t st_, vecT* v_) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = st_ * (256 * 4) + u * 256 + tid;
            if (j < nv) {
                v_[u] = rv[j];
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) v_[u][e] = T(INFINITY);
            }
        }
    };
    vecT vpre[4];
    if (GT_DENSE_PREFETCH && vec) load_step(0, vpre);
    for (int64_t st = 0; st < steps; ++st) {
        T x[PT];
        if (vec) {
            vecT v[4];
            if (GT_DENSE_PREFETCH) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = vpre[u];
                if (st + 1 < steps) load_step(st + 1, vpre);
            } else {
                load_step(st, v);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VW; ++e) x[u * VW + e] = v[u][e];
        } else {
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                const int64_t j = st * CHUNK + int64_t(u) * 256 + tid;
                x[u] = j < n ? row[j] : T(INFINITY);
            }
        }
        if (EMIT) {
#pragma unroll
            for (int u = 0; u < PT; ++u) nan_seen |= x[u] != x[u];
        }
        if (st < 2) {
#pragma unroll
            for (int u = 0; u < PT; ++u) mn = x[u] < mn ? x[u] : mn;
            if (st == 0) {
#pragma unroll
                for (int u = 0; u < PT; ++u) x0[u] = x[u];
                if (steps > 1) continue;
            }
            // the bound from the 256 minima, then the values of these steps at or below it start the list
            const key_t kmn = (mn < T(INFINITY)) ? key_of(mn) : kInvalid;
            __syncthreads();
            const key_t kb = select(&kmn, 1);
            if constexpr (sizeof(T) == 4) ub = __uint_as_float(uint32_t(kb));
            else ub = __longlong_as_double((long long)kb);
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                if (x0[u] < T(INFINITY) && (EMIT ? x0[u] <= ub * rfac : key_of(x0[u]) <= kb)) {
                    const int pos = atomicAdd(&cnt, 1);
                    if (pos < LCAP) {
                        lst[pos] = x0[u];
                        if (EMIT) lcolv[pos] = GT_BW_COL(int64_t(0), u);
                    }
                }
                if (st == 1 && x[u] < T(INFINITY) && (EMIT ? x[u] <= ub * rfac : key_of(x[u]) <= kb)) {
                    const int pos = atomicAdd(&cnt, 1);
                    if (pos < LCAP) {
                        lst[pos] = x[u];
                        if (EMIT) lcolv[pos] = GT_BW_COL(int64_t(1), u);
                    }
                }
            }
            __syncthreads();
            if (cnt > LCAP) {
                bad = true;
                break;
            }
            continue;
        }
        {
            const T rb = EMIT ? ub * rfac : ub;
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                if (x[u] <= rb) {
                    const int pos = atomicAdd(&cnt, 1);
                    if (pos < LCAP) {
                        lst[pos] = x[u];
                        if (EMIT) lcolv[pos] = GT_BW_COL(st, u);
                    }
                }
            }
        }
        if (!((st & 7) == 7)) continue;
        __syncthreads();
        const int m = cnt;   // (uniform: read behind the barrier)
        if (m > LCAP) {      // appends were dropped
            bad = true;
            break;
        }
        if (m > LCAP / 2) {
            key_t kb;
            {
                key_t k4[LCAP / 256];
#pragma unroll
                for (int q = 0; q < LCAP / 256; ++q) k4[q] = (q * 256 + tid < m) ? key_of(lst[q * 256 + tid]) : kInvalid;
                kb = select(k4, LCAP / 256);
            }
            if constexpr (sizeof(T) == 4) ub = __uint_as_float(uint32_t(kb));
            else ub = __longlong_as_double((long long)kb);
            // the entries are read again (not held across the selection: registers), then the list is rebuilt
            T xv[LCAP / 256];
            uint32_t cv[LCAP / 256];
#pragma unroll
            for (int q = 0; q < LCAP / 256; ++q) {
                const bool in = q * 256 + tid < m;
                xv[q] = in ? lst[q * 256 + tid] : T(INFINITY);
                cv[q] = (EMIT && in) ? lcolv[q * 256 + tid] : 0xFFFFFFFFu;
            }
            __syncthreads();
            if (tid == 0) cnt = 0;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < LCAP / 256; ++q) {
                const bool in = q * 256 + tid < m;
                // (the plain list holds keys' values: anything <= 0 is stored as the key says; the emitting list keeps the
                //  value itself - its affinity is evaluated from it)
                if (in && (EMIT ? xv[q] <= ub * rfac : key_of(xv[q]) <= kb)) {
                    const int pos = atomicAdd(&cnt, 1);
                    if (EMIT) {
                        lst[pos] = xv[q];
                        lcolv[pos] = cv[q];
                    } else if constexpr (sizeof(T) == 4) {
                        lst[pos] = __uint_as_float(uint32_t(key_of(xv[q])));
                    } else {
                        lst[pos] = __longlong_as_double((long long)key_of(xv[q]));
                    }
                }
            }
            __syncthreads();
            if (cnt > LCAP / 2) {   // ties at the kth value fill the list: the generic kernel's case
                bad = true;
                break;
            }
        }
    }
    __syncthreads();
    if (!bad && cnt > LCAP) bad = true;
    if (EMIT) {
        if (__syncthreads_or(nan_seen ? 1 : 0)) bad = true;
    }
    if (bad) {
        if (tid == 0) {
            bw[i] = -1.0;
            atomicAdd(redo, 1u);
            if (EMIT) em.scan_rows[atomicAdd(em.scan_count, 1u)] = uint32_t(i);
        }
        return;
    }
    const int m = cnt;
    key_t k4[LCAP / 256];
#pragma unroll
    for (int q = 0; q < LCAP / 256; ++q) k4[q] = (q * 256 + tid < m) ? key_of(lst[q * 256 + tid]) : kInvalid;
    const key_t kb = select(k4, LCAP / 256);
    double r;
    if constexpr (sizeof(T) == 4) r = double(__uint_as_float(uint32_t(kb)));
    else r = __longlong_as_double((long long)kb);
    if (tid == 0) bw[i] = r * scale;
    if constexpr (EMIT && sizeof(T) == 4) {
        // the row's kept affinities, from the list (dense_rows_scan_kernel's arithmetic and outputs)
        __shared__ double reds[4];
        __shared__ uint32_t nzc, base_lo, base_hi;
        const float bwi = float(r * scale);
        float av[LCAP / 256];
        double sown = 0.0;
        uint32_t nh = 0;
        bool diag_seen = false;
        if (tid == 0) nzc = 0u;
#pragma unroll
        for (int q = 0; q < LCAP / 256; ++q) {
            av[q] = 0.f;
            if (q * 256 + tid < m) {
                float a = affinity_t<float>(lst[q * 256 + tid], bwi, em.decay, em.xcut);
                if (a < em.thresh) a = 0.f;
                av[q] = a;
                sown += double(a);
                nh += a != 0.f ? 1u : 0u;
                diag_seen |= a != 0.f && lcolv[q * 256 + tid] == uint32_t(i);
            }
        }
        sown = wave_sum_f64(sown);
        if ((tid & 63) == 0) reds[tid >> 6] = sown;
        __syncthreads();
        const uint32_t slot = nh ? atomicAdd(&nzc, nh) : 0u;
        __syncthreads();
        if (tid == 0) {
            em.own_sum[i] = (reds[0] + reds[1]) + (reds[2] + reds[3]);
            const unsigned long long g = nzc ? atomicAdd(em.tl.cursor, (unsigned long long)nzc) : 0ull;   // the row's ONE reservation
            em.tl.own_base[i] = g;
            em.tl.own_cnt[i] = nzc;
            base_lo = uint32_t(g);
            base_hi = uint32_t(g >> 32);
        }
        __syncthreads();
        unsigned long long g = ((unsigned long long)base_lo | ((unsigned long long)base_hi << 32)) + slot;
#pragma unroll
        for (int q = 0; q < LCAP / 256; ++q)
            if (av[q] != 0.f) {
                if (g < em.tl.cap) {
                    em.tl.i[g] = uint32_t(i);
                    em.tl.j[g] = lcolv[q * 256 + tid];
                    em.tl.a[g] = av[q];
                }
                ++g;
            }
        if (!__syncthreads_or(diag_seen ? 1 : 0) && tid == 0) atomicOr(em.flags, GT_FLAG_ZERO_DIAGONAL);
    }
#undef GT_BW_COL
}

// generic (any kth): bitwise search straight over the row (64 passes; the row stays L2 resident)
template <typename T>
__global__ __launch_bounds__(256) void dense_bandwidth_generic_kernel(const T* __restrict__ D, const int64_t n,
                                                                      const int kth, const double scale,
                                                                      double* __restrict__ bw, const int only_flagged) {
    __shared__ int red[4];
    const int64_t i = blockIdx.x;
    const int tid = threadIdx.x;
    if (only_flagged && !(bw[i] < 0.0)) return;   // block-uniform: redo pass after the two-pass kernel
    const T* row = D + i * n;
    unsigned long long v = 0ull;
    for (int b = 63; b >= 0; --b) {
        const unsigned long long trial = v | ((1ull << b) - 1ull);
        int c = 0;
        for (int64_t j = tid; j < n; j += 256) {
            const double x = double(row[j]);
            const unsigned long long key = (x > 0.0) ? (unsigned long long)__double_as_longlong(x) : 0ull;
            c += (key <= trial) ? 1 : 0;
        }
        c = wave_sum_i32(c);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = c;
        __syncthreads();
        const int total = red[0] + red[1] + red[2] + red[3];
        if (total < kth) v |= (1ull << b);
    }
    if (tid == 0) bw[i] = __longlong_as_double((long long)v) * scale;
}

// ---- bandwidth from data: exact difference-form distance of the (knn+1) nearest candidates ---------
template <typename T>
__global__ __launch_bounds__(256) void dense_bandwidth_from_knn_kernel(const T* __restrict__ X, const int64_t n,
                                                                       const int d, const uint32_t* __restrict__ cand_j,
                                                                       const int MP, const int kth, const double scale,
                                                                       double* __restrict__ bw) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const T* xi = X + i * d;
    double m = 0.0;
    for (int c = 0; c < kth; ++c) {
        const T* xj = X + int64_t(cand_j[i * MP + c]) * d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) {
            const double diff = double(xi[k]) - double(xj[k]);
            s += diff * diff;
        }
        const double dist = sqrt(s);
        m = dist > m ? dist : m;
    }
    bw[i] = m * scale;
}

__global__ void dense_user_bandwidth_kernel(const double* __restrict__ user, const int64_t len, const int64_t n,
                                            const double scale, double* __restrict__ bw) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) bw[i] = (len == 1 ? user[0] : user[i]) * scale;
}

// ---- tile-pair kernel ------------------------------------------------------------------------------
// grid.x enumerates tile pairs (bi <= bj).  TC = compute/output dtype, TD = dtype of the stored distances.
template <typename TD, typename TC, typename TX, bool FROM_DATA>
__global__ __launch_bounds__(256) void dense_kernel_tiles(const TD* __restrict__ D, const TX* __restrict__ X, const int d,
                                                          const int64_t n, const int nb, const double* __restrict__ bw,
                                                          const double decay_d, const double thresh_d, const int symm,
                                                          const double theta_d, TC* __restrict__ Kout,
                                                          uint32_t* __restrict__ flags, const double xcut_d, const int pass,
                                                          double* __restrict__ rowsum, const int rs_mode) {
    // rs_mode (float32 16-byte path, rowsum given): 0 K is written and the row sums accumulated on the way; 1 NOTHING is written,
    // only the row sums (and the zero-diagonal flag) are formed; 2 the row sums are final: the tiles are computed again and
    // P = K / rowsum is written instead of K.  1 + 2 = the operator in place without ever storing K: 4 N^2 + 8 N^2 bytes instead
    // of 8 N^2 (K) + 8 N^2 (normalisation pass) - the distance tiles are read twice, the affinities computed twice (cheap: the
    // transcendental work is skipped beyond the cut).
    // rowsum (optional, pre-zeroed; float32 16-byte path only): sum of |K_ij| of every row, accumulated tile by tile - one
    // float64 atomic per row and tile (the 16 lanes that hold a row's 64 outputs reduce first) - so that the row sums
    // behind P and the degrees do not need a pass of their own over the N x N matrix (4 N^2 bytes of HBM reads)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC* sA = reinterpret_cast<TC*>(smem_raw);      // [TS][TSP]  K0 of tile (I,J): sA[i][j]
    TC* sB = sA + TS * TSP;                        // [TS][TSP]  K0 of tile (J,I): sB[j][i]
    double* xI = reinterpret_cast<double*>(sB + TS * TSP);   // FROM_DATA: [TS][KC] chunks
    // Tile pairs are visited in super-tiles of SG x SG tiles: the workgroups in flight together then cover SG adjacent
    // 256-byte segments of every row they touch - for the (I,J) tiles AND for the transposed (J,I) tiles, whose rows
    // would otherwise be hit one isolated segment at a time (DRAM page misses on half of the traffic).  Positions
    // below the diagonal exit at once.
    const int nbs = (nb + SG - 1) / SG;
    const int64_t st = int64_t(blockIdx.x) / (SG * SG);
    const int within = int(int64_t(blockIdx.x) % (SG * SG));
    const int sbi = int(st / nbs), sbj = int(st % nbs);
    if (sbi > sbj) return;
    const int bi = sbi * SG + within / SG;
    const int bj = sbj * SG + within % SG;
    if (bi > bj || bj >= nb) return;
    const int64_t I0 = int64_t(bi) * TS, J0 = int64_t(bj) * TS;
    const int tx = threadIdx.x & 63;
    const int ty = threadIdx.x >> 6;
    const TC decay = TC(decay_d), thresh = TC(thresh_d), theta = TC(theta_d), xcut = TC(xcut_d);
    const bool diag = (bi == bj);
    constexpr bool kVecOK = !FROM_DATA && sizeof(TD) == 4 && sizeof(TC) == 4;
    const bool vec4 = kVecOK && (n % 4) == 0;

    if (FROM_DATA) {
        constexpr int KC = 32, KCP = 33;   // padded stride: conflict-free per-lane rows
        double* xJ = xI + TS * KCP;
        double acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0;
        for (int k0 = 0; k0 < d; k0 += KC) {
            __syncthreads();
            for (int e = threadIdx.x; e < TS * KC; e += 256) {
                const int r = e / KC, k = e % KC;
                const int64_t gi = I0 + r, gj = J0 + r;
                xI[r * KCP + k] = (gi < n && k0 + k < d) ? double(X[gi * d + k0 + k]) : 0.0;
                xJ[r * KCP + k] = (gj < n && k0 + k < d) ? double(X[gj * d + k0 + k]) : 0.0;
            }
            __syncthreads();
            const int kc = (d - k0) < KC ? (d - k0) : KC;
            for (int k = 0; k < kc; ++k) {
                const double xj = xJ[tx * KCP + k];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double diff = xI[(ty + 4 * r) * KCP + k] - xj;
                    acc[r] += diff * diff;
                }
            }
        }
        // D is symmetric here: K0_ij uses bw_i, K0_ji uses bw_j
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ty + 4 * r, j = tx;
            const int64_t gi = I0 + i, gj = J0 + j;
            TC ka = TC(0), kb = TC(0);
            if (gi < n && gj < n) {
                const TC dist = TC(sqrt(acc[r]));
                ka = affinity_t<TC>(dist, TC(bw[gi]), decay, xcut);
                kb = affinity_t<TC>(dist, TC(bw[gj]), decay, xcut);
                if (ka < thresh) ka = TC(0);
                if (kb < thresh) kb = TC(0);
                if (dist == TC(0) && gi != gj) atomicOr(flags, GT_FLAG_DUPLICATES);
            }
            sA[i * TSP + j] = ka;
            sB[j * TSP + i] = kb;
        }
    } else if (vec4) {
        // float32 distances, n a multiple of 4: 16-byte loads and stores, one address computation per four elements
        // (the scalar form below spends ~90 instructions per element, most of them on addressing)
        if constexpr (kVecOK) {
            const int tx4 = threadIdx.x & 15, ty16 = threadIdx.x >> 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ty16 + 16 * r;
                {   // tile (I,J): row I0+i, columns J0 + 4 tx4 ..
                    const int64_t gi = I0 + i, gj0 = J0 + 4 * tx4;
                    float ka[4] = {0.f, 0.f, 0.f, 0.f};
                    if (gi < n && gj0 < n) {
                        const float4 v = *reinterpret_cast<const float4*>(D + gi * n + gj0);
                        const float bwi = float(bw[gi]);
                        ka[0] = k0_t<float>(v.x, bwi, decay, xcut, pass, gi == gj0);
                        ka[1] = k0_t<float>(v.y, bwi, decay, xcut, pass, gi == gj0 + 1);
                        ka[2] = k0_t<float>(v.z, bwi, decay, xcut, pass, gi == gj0 + 2);
                        ka[3] = k0_t<float>(v.w, bwi, decay, xcut, pass, gi == gj0 + 3);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ka[e] < thresh) ka[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) sA[i * TSP + 4 * tx4 + e] = ka[e];
                }
                if (!diag) {   // tile (J,I): row J0+i, columns I0 + 4 tx4 ..
                    const int64_t gj = J0 + i, gi0 = I0 + 4 * tx4;
                    float kb[4] = {0.f, 0.f, 0.f, 0.f};
                    if (gj < n && gi0 < n) {
                        const float4 v = *reinterpret_cast<const float4*>(D + gj * n + gi0);
                        const float bwj = float(bw[gj]);
                        kb[0] = k0_t<float>(v.x, bwj, decay, xcut, pass, gj == gi0);
                        kb[1] = k0_t<float>(v.y, bwj, decay, xcut, pass, gj == gi0 + 1);
                        kb[2] = k0_t<float>(v.z, bwj, decay, xcut, pass, gj == gi0 + 2);
                        kb[3] = k0_t<float>(v.w, bwj, decay, xcut, pass, gj == gi0 + 3);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (kb[e] < thresh) kb[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) sB[i * TSP + 4 * tx4 + e] = kb[e];
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ty16 + 16 * r;
                {   // output tile (I,J)
                    const int64_t gi = I0 + i, gj0 = J0 + 4 * tx4;
                    double part = 0.0;
                    if (gi < n && gj0 < n) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = sA[i * TSP + 4 * tx4 + e];
                            const float b = diag ? sA[(4 * tx4 + e) * TSP + i] : sB[(4 * tx4 + e) * TSP + i];
                            o[e] = merge_t<float>(a, b, symm, theta);
                        }
                        if (rs_mode == 2) {
                            double sd = rowsum[gi];
                            if (sd == 0.0) sd = 1.0;   // sklearn _handle_zeros_in_scale
                            const float sf = float(sd);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = o[e] / sf;
                        }
                        if (rs_mode != 1) *reinterpret_cast<float4*>(Kout + gi * n + gj0) = make_float4(o[0], o[1], o[2], o[3]);
                        part = double(fabsf(o[0])) + double(fabsf(o[1])) + double(fabsf(o[2])) + double(fabsf(o[3]));
                        if (rowsum && rs_mode != 2 && diag && gi >= gj0 && gi < gj0 + 4 && o[gi - gj0] == 0.f)
                            atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
                    }
                    if (rowsum && rs_mode != 2) {   // (uniform: the 16 lanes of a row's group reduce together, in or out of range)
#pragma unroll
                        for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 16);
                        if (tx4 == 0 && gi < n) atomicAdd(rowsum + gi, part);
                    }
                }
                if (!diag) {   // output tile (J,I)
                    const int64_t gj = J0 + i, gi0 = I0 + 4 * tx4;
                    double part = 0.0;
                    if (gj < n && gi0 < n) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float b = sB[i * TSP + 4 * tx4 + e];
                            const float a = sA[(4 * tx4 + e) * TSP + i];
                            o[e] = merge_t<float>(b, a, symm, theta);
                        }
                        if (rs_mode == 2) {
                            double sd = rowsum[gj];
                            if (sd == 0.0) sd = 1.0;
                            const float sf = float(sd);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = o[e] / sf;
                        }
                        if (rs_mode != 1) *reinterpret_cast<float4*>(Kout + gj * n + gi0) = make_float4(o[0], o[1], o[2], o[3]);
                        part = double(fabsf(o[0])) + double(fabsf(o[1])) + double(fabsf(o[2])) + double(fabsf(o[3]));
                    }
                    if (rowsum && rs_mode != 2) {
#pragma unroll
                        for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 16);
                        if (tx4 == 0 && gj < n) atomicAdd(rowsum + gj, part);
                    }
                }
            }
        }
        return;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ty + 4 * r;
            {   // tile (I,J): row I0+i, column J0+tx
                const int64_t gi = I0 + i, gj = J0 + tx;
                TC ka = TC(0);
                if (gi < n && gj < n) {
                    ka = k0_t<TC>(TC(D[gi * n + gj]), TC(bw[gi]), decay, xcut, pass, gi == gj);
                    if (ka < thresh) ka = TC(0);
                }
                sA[i * TSP + tx] = ka;
            }
            if (!diag) {   // tile (J,I): row J0+i, column I0+tx
                const int64_t gj = J0 + i, gi = I0 + tx;
                TC kb = TC(0);
                if (gi < n && gj < n) {
                    kb = k0_t<TC>(TC(D[gj * n + gi]), TC(bw[gj]), decay, xcut, pass, gi == gj);
                    if (kb < thresh) kb = TC(0);
                }
                sB[i * TSP + tx] = kb;
            }
        }
    }
    __syncthreads();
    // ---- symmetrise + write (both tiles are fully consumed above: safe to overwrite D in place) ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = ty + 4 * r;
        {   // output tile (I,J): element (i, tx)
            const int64_t gi = I0 + i, gj = J0 + tx;
            if (gi < n && gj < n) {
                const TC a = sA[i * TSP + tx];
                const TC b = diag ? sA[tx * TSP + i] : sB[tx * TSP + i];
                Kout[gi * n + gj] = merge_t<TC>(a, b, symm, theta);
            }
        }
        if (!diag) {   // output tile (J,I): element (row j = i-th row of J block, column tx of I block)
            const int64_t gj = J0 + i, gi = I0 + tx;
            if (gi < n && gj < n) {
                const TC b = sB[i * TSP + tx];
                const TC a = sA[tx * TSP + i];
                Kout[gj * n + gi] = merge_t<TC>(b, a, symm, theta);
            }
        }
    }
}


// ---- row-streaming form of the '+' rule on a float32 distance matrix ------------------------------------------------------
// K = (K0 + K0^T) / 2 needs K0_ji next to K0_ij.  The tile-pair kernel reads both tiles - 256-byte row segments of 64 rows each,
// which the HBM serves at 4.3 TB/s whatever the kernel does with them (a read-only tile pass takes as long).  But K0 is SPARSE
// after the threshold (decay 40, thresh 1e-4: ~170 of 200 000 entries per row): the transposed half can travel as a list.
//   S1  dense_rows_scan_kernel    one workgroup per row streams the row (16-byte loads): own row sum, non-zeros {i, j, K0_ij}
//                                 collected in LDS and appended to a flat list (one reservation per row)
//   T   tri_count / scan / tri_scatter   the list transposed: for every row i the entries {j, K0_ji} ("incoming")
//   S3  dense_rows_write_kernel   one workgroup per row: row sum = (own + incoming) / 2; the incoming entries are merged with
//                                 the row's own values first (their distances are still there: the row is written by this
//                                 workgroup only) and parked.  Then nothing of the matrix is needed any more: zeros are
//                                 streamed over it (dense_zero_rows_kernel), every row's own entries - they are in the list -
//                                 are placed as (K0_ij + 0) / 2, or / row sum when P is what is wanted, and the parked values
//                                 overwrite their places.  (The first form of this pass streamed every row once more and
//                                 recomputed its affinities: option dense_rows_reread, and what a row does whose entries are
//                                 not in one piece of the list.)
// Every element sees the operations of the tile-pair kernel in the same order ((a + b) / 2, then / float(row sum)): same bits
// up to the summation order of the float64 row sums.  Bytes: 4 N^2 read (S1, inside the bandwidth pass) + 4 N^2 written (S3)
// instead of 4 N^2 + 8 N^2 + 8 N^2 (bandwidths, tile pairs, normalisation pass): the matrix is read ONCE and written once.

__global__ __launch_bounds__(256) void dense_rows_scan_kernel(const float* __restrict__ D, const int64_t n,
                                                              const double* __restrict__ bw, const double decay_d,
                                                              const double thresh_d, const double xcut_d,
                                                              double* __restrict__ own_sum, const TriList tl,
                                                              uint32_t* __restrict__ flags, const uint32_t* __restrict__ row_list) {
    __shared__ uint32_t lcol[ROWS_LDS_CAP];
    __shared__ float lval[ROWS_LDS_CAP];
    __shared__ uint32_t lcount, lbase_lo, lbase_hi;
    __shared__ double red[4];
    const int64_t i = row_list ? int64_t(row_list[blockIdx.x]) : int64_t(blockIdx.x);
    const float decay = float(decay_d), thresh = float(thresh_d), xcut = float(xcut_d);
    const float bwi = float(bw[i]);
    if (threadIdx.x == 0) lcount = 0u;
    __syncthreads();
    const float4* rv = reinterpret_cast<const float4*>(D + i * n);
    const int64_t nv = n / 4;
    double s = 0.0;
    for (int64_t j0 = 0; j0 < nv; j0 += 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + u * 256 + threadIdx.x;
            v[u] = (j < nv) ? rv[j] : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + u * 256 + threadIdx.x;
            float a[4];
            a[0] = affinity_t<float>(v[u].x, bwi, decay, xcut);
            a[1] = affinity_t<float>(v[u].y, bwi, decay, xcut);
            a[2] = affinity_t<float>(v[u].z, bwi, decay, xcut);
            a[3] = affinity_t<float>(v[u].w, bwi, decay, xcut);
            int nh = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a[e] < thresh) a[e] = 0.f;
                nh += a[e] != 0.f ? 1 : 0;
            }
            if (j < nv && nh) {   // (rare: a handful of the row's entries survive the threshold)
                s += (double(a[0]) + double(a[1])) + (double(a[2]) + double(a[3]));
                const uint32_t slot = atomicAdd(&lcount, uint32_t(nh));
                if (slot + uint32_t(nh) <= uint32_t(ROWS_LDS_CAP)) {
                    uint32_t k = slot;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (a[e] != 0.f) {
                            lcol[k] = uint32_t(4 * j + e);
                            lval[k] = a[e];
                            ++k;
                        }
                } else {
                    // a row with more non-zeros than the LDS list holds: appended directly (correct, slow)
                    unsigned long long g = atomicAdd(tl.cursor, (unsigned long long)nh);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (a[e] != 0.f) {
                            if (g < tl.cap) {
                                tl.i[g] = uint32_t(i);
                                tl.j[g] = uint32_t(4 * j + e);
                                tl.a[g] = a[e];
                            }
                            ++g;
                        }
                    // (its slots of the LDS list stay unused: marked below)
                    for (uint32_t k = slot; k < slot + uint32_t(nh) && k < uint32_t(ROWS_LDS_CAP); ++k) lcol[k] = 0xFFFFFFFFu;
                }
            }
        }
    }
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const uint32_t nl = lcount < uint32_t(ROWS_LDS_CAP) ? lcount : uint32_t(ROWS_LDS_CAP);
    if (threadIdx.x == 0) {
        own_sum[i] = (red[0] + red[1]) + (red[2] + red[3]);
        const unsigned long long g = nl ? atomicAdd(tl.cursor, (unsigned long long)nl) : 0ull;   // the row's ONE reservation
        tl.own_base[i] = g;
        tl.own_cnt[i] = lcount <= uint32_t(ROWS_LDS_CAP) ? nl : kNoOwnList;
        lbase_lo = uint32_t(g);
        lbase_hi = uint32_t(g >> 32);
    }
    __syncthreads();
    const unsigned long long base = (unsigned long long)lbase_lo | ((unsigned long long)lbase_hi << 32);
    bool diag_seen = false;
    for (uint32_t k = threadIdx.x; k < nl; k += 256) {
        const uint32_t c = lcol[k];
        const unsigned long long g = base + k;
        if (g < tl.cap) {   // (unused slots travel as zeros to column 0xFFFFFFFF: skipped by the consumers)
            tl.i[g] = uint32_t(i);
            tl.j[g] = c;
            tl.a[g] = c == 0xFFFFFFFFu ? 0.f : lval[k];
        }
        diag_seen |= c == uint32_t(i);
    }
    // K_ii = K0_ii: zero when the row's own diagonal entry did not survive (base.py:553 warns)
    if (lcount <= uint32_t(ROWS_LDS_CAP)) {
        if (!__syncthreads_or(diag_seen ? 1 : 0) && threadIdx.x == 0) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
    } else if (threadIdx.x == 0) {
        const float aii = affinity_t<float>(D[i * n + i], bwi, decay, xcut);
        if (aii < thresh) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
    }
}

__global__ __launch_bounds__(256) void tri_count_kernel(const uint32_t* __restrict__ tj, const int64_t total,
                                                        uint32_t* __restrict__ incount) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t < total && tj[t] != 0xFFFFFFFFu) atomicAdd(&incount[tj[t]], 1u);
}

// exclusive scan of n counts by ONE workgroup (n is a row count: at most a few million)
__global__ __launch_bounds__(1024) void rows_scan_counts_kernel(const uint32_t* __restrict__ cnt, const int64_t n,
                                                                unsigned long long* __restrict__ ptr) {
    __shared__ unsigned long long part[1024];
    const int64_t per = (n + 1023) / 1024;
    const int64_t a = int64_t(threadIdx.x) * per, b = a + per < n ? a + per : n;
    unsigned long long s = 0ull;
    for (int64_t k = a; k < b; ++k) s += cnt[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0ull;
        for (int t = 0; t < 1024; ++t) {
            const unsigned long long v = part[t];
            part[t] = run;
            run += v;
        }
        ptr[n] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int64_t k = a; k < b; ++k) {
        ptr[k] = run;
        run += cnt[k];
    }
}

__global__ __launch_bounds__(256) void tri_scatter_kernel(const uint32_t* __restrict__ ti, const uint32_t* __restrict__ tj,
                                                          const float* __restrict__ ta, const int64_t total,
                                                          const unsigned long long* __restrict__ inptr,
                                                          uint32_t* __restrict__ cursor, uint32_t* __restrict__ in_col,
                                                          float* __restrict__ in_val) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= total) return;
    const uint32_t j = tj[t];
    if (j == 0xFFFFFFFFu) return;
    const unsigned long long k = inptr[j] + atomicAdd(&cursor[j], 1u);
    in_col[k] = ti[t];
    in_val[k] = ta[t];
}

// zeros streamed over the rows whose entries the placement launch will write (the others were written whole by the merge launch)
__global__ __launch_bounds__(256) void dense_zero_rows_kernel(float* __restrict__ out, const int64_t n,
                                                              const uint32_t* __restrict__ own_cnt) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v z;
    z[0] = z[1] = z[2] = z[3] = 0.f;
    const int64_t nv = n / 4;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (own_cnt[i] == kNoOwnList) continue;   // (workgroup-uniform)
        f4v* ov = reinterpret_cast<f4v*>(out + i * n);
        for (int64_t j0 = threadIdx.x; j0 < nv; j0 += 256 * 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u * 256 < nv) __builtin_nontemporal_store(z, ov + j0 + u * 256);
        }
    }
}

// The write pass.  PART 0: one launch - every row is read again, its affinities recomputed and written, the merged entries placed
// over them (the first form of this pass: 8 N^2 bytes).  PART 1, dense_zero_rows_kernel, PART 2 - three launches, 4 N^2 bytes: the
// rows' kept affinities are known (the list the transposition read), so the matrix is not read again - (1) the incoming entries
// are merged with the row's own values and the row sums formed while the distances are still there, (2) zeros are streamed over
// the matrix (a pure store stream: 7 TB/s), (3) every row's own entries take their places, (K0_ij + 0) / 2 like the streamed
// form computes them, then the merged ones.  A row whose own entries are not in one piece of the list (own_cnt = kNoOwnList)
// is written whole by launch 1, PART 0's way, and left alone by the other two.
template <bool DIVIDE, int PART>   // DIVIDE: P = K / row sum is written instead of K
__global__ __launch_bounds__(256) void dense_rows_write_kernel(const float* __restrict__ D, const int64_t n,
                                                               const double* __restrict__ bw, const double decay_d,
                                                               const double thresh_d, const double xcut_d,
                                                               const double* __restrict__ own_sum,
                                                               const unsigned long long* __restrict__ inptr,
                                                               const uint32_t* __restrict__ in_col, float* __restrict__ in_val,
                                                               double* __restrict__ rowsum, float* __restrict__ out,
                                                               const TriList own) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    __shared__ double red[4];
    __shared__ float sf_s;
    const int64_t i = blockIdx.x;
    const float decay = float(decay_d), thresh = float(thresh_d), xcut = float(xcut_d);
    const float bwi = float(bw[i]);
    const unsigned long long p0 = inptr[i], p1 = inptr[i + 1];
    // the entries that have a transposed partner: merged now, while the row's distances are still in place.  Row sum of K (what
    // numpy sums is the float32 K): every entry contributes (K0_ij + 0) / 2 = own / 2 unless it has a partner - those contribute
    // their merged, float32-rounded value instead.  Float64 sums of exactly the values the tile-pair kernel sums.
    const float* drow = D + i * n;
    float* orow = out + i * n;
    if constexpr (PART == 2) {
        const uint32_t n_own2 = own.own_cnt[i];
        if (n_own2 == kNoOwnList) return;   // (workgroup-uniform)
        const float sf2 = float(rowsum[i] == 0.0 ? 1.0 : rowsum[i]);
        const unsigned long long b2 = own.own_base[i];
        for (uint32_t k = threadIdx.x; k < n_own2; k += 256) {
            float o = merge_t<float>(own.a[b2 + k], 0.f, GT_SYMM_ADD, 1.f);
            if (DIVIDE) o = o / sf2;
            orow[own.j[b2 + k]] = o;
        }
        // a partnered entry is written twice, by different waves: the merged value must land last.  Every wave waits until its
        // own stores are ACKNOWLEDGED by the L2 (vmcnt counts stores down when the L2 has them) before it meets the others at
        // the barrier - the second store to an address is then issued after the first is in place, whatever path it takes.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (unsigned long long k = p0 + threadIdx.x; k < p1; k += 256) orow[in_col[k]] = in_val[k];
        return;
    }
    double corr = 0.0;
    for (unsigned long long k = p0 + threadIdx.x; k < p1; k += 256) {
        const uint32_t j = in_col[k];
        float a = affinity_t<float>(drow[j], bwi, decay, xcut);
        if (a < thresh) a = 0.f;
        const float o = merge_t<float>(a, in_val[k], GT_SYMM_ADD, 1.f);
        in_val[k] = o;
        corr += double(o) - double(merge_t<float>(a, 0.f, GT_SYMM_ADD, 1.f));
    }
    corr = wave_sum_f64(corr);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = corr;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double rs = own_sum[i] / 2.0 + ((red[0] + red[1]) + (red[2] + red[3]));
        rowsum[i] = rs;
        sf_s = float(rs == 0.0 ? 1.0 : rs);   // sklearn _handle_zeros_in_scale
    }
    __syncthreads();
    const float sf = sf_s;
    if (DIVIDE)
        for (unsigned long long k = p0 + threadIdx.x; k < p1; k += 256) in_val[k] = in_val[k] / sf;
    if (PART == 1 && own.own_cnt[i] != kNoOwnList) return;   // (workgroup-uniform)
    __syncthreads();   // (in place: every read of the row's distances above comes before the first write below)
    const float4* rv = reinterpret_cast<const float4*>(drow);
    float4* ov = reinterpret_cast<float4*>(out + i * n);
    const int64_t nv = n / 4;
    for (int64_t j0 = 0; j0 < nv; j0 += 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + u * 256 + threadIdx.x;
            if (j < nv) {
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(rv + j));
                v[u] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + u * 256 + threadIdx.x;
            if (j < nv) {
                float a[4];
                a[0] = affinity_t<float>(v[u].x, bwi, decay, xcut);
                a[1] = affinity_t<float>(v[u].y, bwi, decay, xcut);
                a[2] = affinity_t<float>(v[u].z, bwi, decay, xcut);
                a[3] = affinity_t<float>(v[u].w, bwi, decay, xcut);
                f4v t;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (a[e] < thresh) a[e] = 0.f;
                    float o = merge_t<float>(a[e], 0.f, GT_SYMM_ADD, 1.f);
                    if (DIVIDE) o = o / sf;
                    t[e] = o;
                }
                __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(ov + j));
            }
        }
    }
    // the streamed row is complete before the merged entries go over it: every wave waits for the L2's acknowledgement of its
    // own (non-temporal) stores - vmcnt(0) - and only then meets the others at the barrier, so the second store to an address is
    // issued after the first is in place.  (The barrier alone does not wait on vmcnt outside tgsplit mode; NOT __threadfence():
    // a device-scope release writes the XCD's L2 back - once per row, 1.9 x the time.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (unsigned long long k = p0 + threadIdx.x; k < p1; k += 256) orow[in_col[k]] = in_val[k];
}

// ---- row sums / anisotropy / normalisation ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dense_rowsum_kernel(const T* __restrict__ K, const int64_t n, const int use_abs,
                                                           double* __restrict__ out, uint32_t* __restrict__ flags) {
    __shared__ double red[4];
    const int64_t i = blockIdx.x;
    const T* row = K + i * n;
    double s = 0.0;
    constexpr int VW = 16 / int(sizeof(T));
    typedef T vecT __attribute__((ext_vector_type(VW)));
    if ((n % VW) == 0) {
        // 16-byte loads, four in flight per thread (a pure HBM stream)
        const vecT* rv = reinterpret_cast<const vecT*>(row);
        const int64_t nv = n / VW;
        for (int64_t j0 = 0; j0 < nv; j0 += 256 * 4) {
            vecT v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = j0 + u * 256 + threadIdx.x;
                if (j < nv) {
                    v[u] = rv[j];
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e) v[u][e] = T(0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const double x = double(v[u][e]);
                    s += use_abs ? fabs(x) : x;
                }
        }
    } else {
        for (int64_t j = threadIdx.x; j < n; j += 256) {
            const double v = double(row[j]);
            s += use_abs ? fabs(v) : v;
        }
    }
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[i] = red[0] + red[1] + red[2] + red[3];
        if (flags && row[i] == T(0)) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dense_anisotropy_kernel(T* __restrict__ K, const int64_t n,
                                                               const double* __restrict__ deg, const double alpha) {
    const int64_t i = blockIdx.y;
    const int64_t j = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (j >= n) return;
    if constexpr (sizeof(T) == 4) {
        // numpy: float32 K / (outer(d, d) ** alpha) with d float32
        const float q = powf(float(deg[i]) * float(deg[j]), float(alpha));
        K[i * n + j] = K[i * n + j] / q;
    } else {
        K[i * n + j] = K[i * n + j] / pow(deg[i] * deg[j], alpha);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dense_normalize_kernel(const T* __restrict__ K, const int64_t n,
                                                              const double* __restrict__ rowsum, T* __restrict__ P) {
    const int64_t i = blockIdx.y;
    const int64_t j = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (j >= n) return;
    double s = rowsum[i];
    if (s == 0.0) s = 1.0;   // sklearn _handle_zeros_in_scale
    P[i * n + j] = T(K[i * n + j] / T(s));
}

// P = K / rowsum, float32, four values per thread (in place allowed: every thread rewrites what it read)
__global__ __launch_bounds__(256) void dense_normalize4_kernel(const float4* __restrict__ K, const int64_t n4,
                                                               const double* __restrict__ rowsum, float4* __restrict__ P) {
    // one workgroup per row, four 16-byte loads in flight per thread (a pure HBM stream)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int64_t i = blockIdx.x;
    double s = rowsum[i];
    if (s == 0.0) s = 1.0;   // sklearn _handle_zeros_in_scale
    const float sf = float(s);
    const float4* kr = K + i * n4;
    float4* pr = P + i * n4;
    for (int64_t j0 = 0; j0 < n4; j0 += 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + u * 256 + threadIdx.x;
            if (j < n4) {   // (read once, rewritten once: no reuse to keep in the caches)
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(kr + j));
                v[u] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + u * 256 + threadIdx.x;
            if (j < n4) {
                v[u].x = v[u].x / sf;
                v[u].y = v[u].y / sf;
                v[u].z = v[u].z / sf;
                v[u].w = v[u].w / sf;
                f4v t;
                t[0] = v[u].x; t[1] = v[u].y; t[2] = v[u].z; t[3] = v[u].w;
                __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(pr + j));
            }
        }
    }
}

struct DenseState {
    DevBuf bw, bw_user, rowsum, deg, work_in, work_k, work_p, flags, redo;
    DevBuf own_sum, tri_i, tri_j, tri_a, tri_cursor, incount, inptr, in_col, in_val, scan_rows, own_base, own_cnt;   // row-streaming form
};

// Scaled distance beyond which exp(-x^decay) is exactly 0 in the reference: it falls below `thresh` (zeroed,
// graphs.py:1598-1600) or, with thresh == 0, exp underflows to 0 in the result dtype.  1 % margin in the exponent:
// the value at the cut is >= 8 % (thresh = 1e-4) away from the decision, rounding errors are ~1e-6.
static double dense_xcut(double decay, double thresh, bool f32) {
    if (!(decay > 0.0)) return INFINITY;
    const double p_under = f32 ? 104.5 : 746.0;
    double p_cut = p_under;
    if (thresh > 0.0 && thresh < 1.0) p_cut = std::min(p_under, -std::log(thresh));
    if (thresh >= 1.0) return INFINITY;
    return std::pow(p_cut * 1.01, 1.0 / decay);
}

template <typename TD, typename TC, typename TX, bool FROM_DATA>
int launch_tiles(gt_ctx* ctx, const TD* D, const TX* X, int d, int64_t n, const double* bw, double decay, double thresh,
                 int symm, double theta, TC* Kout, uint32_t* flags, int pass = 0, double* rowsum = nullptr, int rs_mode = 0) {
    const int nb = int(ceil_div64(n, TS));
    const int64_t nbs = ceil_div64(nb, SG);
    const int64_t pairs = nbs * nbs * SG * SG;   // grid positions (super-tile order, lower triangle exits)
    if (pairs >= (int64_t(1) << 31)) GT_FAIL(ctx, GT_E_LIMIT, "dense graph: too many tiles for one launch");
    size_t lds = size_t(2) * TS * TSP * sizeof(TC);
    if (FROM_DATA) lds += size_t(2) * TS * 33 * sizeof(double);
    auto kern = dense_kernel_tiles<TD, TC, TX, FROM_DATA>;
    GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    int(lds)));
    hipLaunchKernelGGL(kern, dim3((unsigned)pairs), dim3(256), lds, ctx->stream, D, X, d, n, nb, bw, decay, thresh, symm,
                       theta, Kout, flags, dense_xcut(decay, thresh, sizeof(TC) == 4), pass, rowsum, rs_mode);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// ---- out-of-sample extension of an exact graph (TraditionalGraph.build_kernel_to_data, graphs.py:1612-1678) -----------
// pdx = cdist(Y, X) (float64 difference form, sequential in k like scipy), bandwidth_y = the knn-th smallest of row y (or
// the caller's) * scale, K_yj = exp(-(pdx_yj / bandwidth_y)^decay), NaN -> 1, < thresh -> 0.  One workgroup per
// 64 x 64 tile; both row blocks are staged through LDS as float64.
template <typename TX>
__global__ __launch_bounds__(256) void dense_extend_tiles(const TX* __restrict__ Y, const int64_t m, const TX* __restrict__ X,
                                                          const int64_t n, const int d, const double* __restrict__ bw,
                                                          const double decay, const double thresh, const double xcut,
                                                          double* __restrict__ Kout) {
    constexpr int KC = 32, KCP = 33;
    __shared__ double yI[TS * KCP], xJ[TS * KCP];
    const int64_t I0 = int64_t(blockIdx.y) * TS, J0 = int64_t(blockIdx.x) * TS;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0;
    for (int k0 = 0; k0 < d; k0 += KC) {
        __syncthreads();
        for (int e = threadIdx.x; e < TS * KC; e += 256) {
            const int r = e / KC, k = e % KC;
            const int64_t gi = I0 + r, gj = J0 + r;
            yI[r * KCP + k] = (gi < m && k0 + k < d) ? double(Y[gi * d + k0 + k]) : 0.0;
            xJ[r * KCP + k] = (gj < n && k0 + k < d) ? double(X[gj * d + k0 + k]) : 0.0;
        }
        __syncthreads();
        const int kc = (d - k0) < KC ? (d - k0) : KC;
        for (int k = 0; k < kc; ++k) {
            const double xj = xJ[tx * KCP + k];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double diff = yI[(ty + 4 * r) * KCP + k] - xj;
                acc[r] += diff * diff;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t gi = I0 + ty + 4 * r, gj = J0 + tx;
        if (gi < m && gj < n) {
            double kv = affinity_t<double>(sqrt(acc[r]), bw[gi], decay, xcut);
            if (kv < thresh) kv = 0.0;
            Kout[gi * n + gj] = kv;
        }
    }
}

// bandwidth of the extension rows from their exact nearest candidates (difference-form distances, like the build)
template <typename T>
__global__ __launch_bounds__(256) void dense_bandwidth_ext_kernel(const T* __restrict__ Y, const int64_t m, const T* __restrict__ X,
                                                                  const int d, const uint32_t* __restrict__ cand_j, const int MP,
                                                                  const int kth, const double scale, double* __restrict__ bw) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= m) return;
    const T* yi = Y + i * d;
    double mx = 0.0;
    for (int c = 0; c < kth; ++c) {
        const T* xj = X + int64_t(cand_j[i * MP + c]) * d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) {
            const double diff = double(yi[k]) - double(xj[k]);
            s += diff * diff;
        }
        const double dist = sqrt(s);
        mx = dist > mx ? dist : mx;
    }
    bw[i] = mx * scale;
}

template <typename T>
int finish_dense(gt_ctx* ctx, DenseState& st, T* K, T* P, int64_t n, double anisotropy, bool have_rowsum = false) {
    GT_HIP(ctx, st.rowsum.reserve(size_t(n) * sizeof(double)));
    if (have_rowsum) {
        // the tile kernel accumulated the row sums (and the zero-diagonal flag): only P is left, 16 bytes per thread
        StageSpan span(ctx, "dense_normalize");
        if (P) {
            if constexpr (sizeof(T) == 4) {
                if (n % 4 == 0) {
                    hipLaunchKernelGGL(dense_normalize4_kernel, dim3((unsigned)n), dim3(256), 0,
                                       ctx->stream, reinterpret_cast<const float4*>(K), n / 4, st.rowsum.as<double>(),
                                       reinterpret_cast<float4*>(P));
                    GT_HIP(ctx, hipGetLastError());
                    return GT_OK;
                }
            }
            hipLaunchKernelGGL(dense_normalize_kernel<T>, dim3((unsigned)ceil_div64(n, 256), (unsigned)n), dim3(256), 0,
                               ctx->stream, K, n, st.rowsum.as<double>(), P);
            GT_HIP(ctx, hipGetLastError());
        }
        return GT_OK;
    }
    {
        StageSpan span(ctx, "dense_normalize");
        if (anisotropy != 0.0) {
            GT_HIP(ctx, st.deg.reserve(size_t(n) * sizeof(double)));
            hipLaunchKernelGGL(dense_rowsum_kernel<T>, dim3((unsigned)n), dim3(256), 0, ctx->stream, K, n, 0,
                               st.deg.as<double>(), (uint32_t*)nullptr);
            hipLaunchKernelGGL(dense_anisotropy_kernel<T>, dim3((unsigned)ceil_div64(n, 256), (unsigned)n), dim3(256), 0,
                               ctx->stream, K, n, st.deg.as<double>(), anisotropy);
        }
        hipLaunchKernelGGL(dense_rowsum_kernel<T>, dim3((unsigned)n), dim3(256), 0, ctx->stream, K, n, 1,
                           st.rowsum.as<double>(), st.flags.as<uint32_t>());
        if (P)
            hipLaunchKernelGGL(dense_normalize_kernel<T>, dim3((unsigned)ceil_div64(n, 256), (unsigned)n), dim3(256), 0,
                               ctx->stream, K, n, st.rowsum.as<double>(), P);
        GT_HIP(ctx, hipGetLastError());
    }
    return GT_OK;
}

}  // namespace

extern "C" int gt_dense_graph_build(gt_ctx* ctx, const void* X_or_D, int64_t n, int32_t d, int32_t dtype,
                                    int32_t on_device, int32_t precomputed, int32_t knn, double decay, double thresh,
                                    const double* bandwidth, int64_t bandwidth_len, double bandwidth_scale,
                                    int32_t kernel_symm, double theta, double anisotropy, int32_t inplace, void* out_K,
                                    void* out_P, int32_t out_on_device, uint32_t* flags) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    if (!X_or_D || n <= 0) GT_FAIL(ctx, GT_E_ARG, "gt_dense_graph_build: empty input");
    if (dtype != GT_F32 && dtype != GT_F64) GT_FAIL(ctx, GT_E_ARG, "dtype must be GT_F32 or GT_F64");
    if (precomputed < 0 || precomputed > GT_PRECOMPUTED_ADJACENCY) GT_FAIL(ctx, GT_E_ARG, "precomputed must be 0 ... 3");
    // 2 / 3: the caller's matrix IS the unsymmetrised kernel (adjacency: with the diagonal set to 1) - no bandwidth, no decay
    const int pass = precomputed >= GT_PRECOMPUTED_AFFINITY ? precomputed - 1 : 0;
    if (pass) {
        bandwidth_len = 0;
        decay = 1.0;
    }
    if (std::isnan(decay)) GT_FAIL(ctx, GT_E_ARG, "`decay` must be provided for a TraditionalGraph");
    if (bandwidth_len != 0 && bandwidth_len != 1 && bandwidth_len != n)
        GT_FAIL(ctx, GT_E_ARG, "bandwidth must have 1 or n entries");
    if (!pass && bandwidth_len == 0 && (knn < 0 || int64_t(knn) + 1 > n)) GT_FAIL(ctx, GT_E_ARG, "knn + 1 exceeds n_samples");
    if (inplace && !(precomputed && on_device)) GT_FAIL(ctx, GT_E_ARG, "inplace needs a device-resident distance matrix");
    DenseState st;
    int rc = GT_OK;
    // result dtype: float64 from data, D's dtype from distances (a float64 bandwidth VECTOR promotes to float64)
    const bool out_f64 = (!precomputed) || dtype == GT_F64 || bandwidth_len > 1;
    if (inplace && out_f64 != (dtype == GT_F64)) GT_FAIL(ctx, GT_E_ARG, "inplace needs equal input and output dtypes");
    const size_t out_esz = out_f64 ? 8 : 4;
    const size_t in_esz = dtype == GT_F64 ? 8 : 4;
    auto cleanup = [&]() {
        for (DevBuf* b : {&st.bw, &st.bw_user, &st.rowsum, &st.deg, &st.work_in, &st.work_k, &st.work_p, &st.flags, &st.redo,
                          &st.own_sum, &st.tri_i, &st.tri_j, &st.tri_a, &st.tri_cursor, &st.incount, &st.inptr, &st.in_col, &st.in_val, &st.own_base, &st.own_cnt,
                          &st.scan_rows})
            b->release();
    };
#define DENSE_TRY(expr)            \
    do {                           \
        rc = (expr);               \
        if (rc != GT_OK) {         \
            cleanup();             \
            return rc;             \
        }                          \
    } while (0)
#define DENSE_HIP(expr)                                                                   \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            /* (as GT_HIP: running out of device memory is a LIMIT of this build, named as such - callers take another route) */ \
            const bool _oom = (_e == hipErrorOutOfMemory);                                \
            ctx->set_error(std::string(_oom ? "the working set of this build does not fit the GPU's memory - " : "") + \
                           std::string(#expr) + ": " + hipGetErrorString(_e));            \
            cleanup();                                                                    \
            return _oom ? GT_E_LIMIT : GT_E_HIP;                                          \
        }                                                                                 \
    } while (0)

    DENSE_HIP(st.bw.reserve(size_t(n) * sizeof(double)));
    if (pass) DENSE_HIP(hipMemsetAsync(st.bw.p, 0, size_t(n) * sizeof(double), ctx->stream));
    DENSE_HIP(st.flags.reserve(sizeof(uint32_t)));
    DENSE_HIP(hipMemsetAsync(st.flags.p, 0, sizeof(uint32_t), ctx->stream));
    const void* in_dev = X_or_D;
    if (!precomputed) {
        // binds the points (device copy + norms + padded working copy for the kNN bandwidth search)
        DENSE_TRY(gt_set_points(ctx, X_or_D, n, d, dtype, on_device));
        in_dev = ctx->X;
    } else if (!on_device) {
        DENSE_HIP(st.work_in.reserve(size_t(n) * n * in_esz));
        DENSE_TRY(gt_copy_from_host(ctx, st.work_in.p, X_or_D, size_t(n) * n * in_esz));
        in_dev = st.work_in.p;
    }
    // Row-streaming form (float32 distances, the '+' rule, no anisotropy, whole quads): the transposed half of the sparse
    // thresholded kernel travels as a list, every pass over the matrix reads and writes whole rows.  P alone wanted: K is never
    // stored; K wanted: the normalisation pass follows as before.
    const bool rows_ok = precomputed && !pass && dtype == GT_F32 && !out_f64 && kernel_symm == GT_SYMM_ADD && anisotropy == 0.0 &&
                         (n % 4) == 0 && n < (int64_t(1) << 32) &&
                         (ctx->dense_rows > 0 || (ctx->dense_rows < 0 && n >= 16384));
    bool rows_listed = false;   // the bandwidth pass has listed the kept affinities already
    TriList tl;
    tl.i = tl.j = nullptr;
    tl.a = nullptr;
    tl.cursor = nullptr;
    tl.cap = 0;
    tl.own_base = nullptr;
    tl.own_cnt = nullptr;
    auto rows_alloc = [&]() -> int {
        if (tl.i) return GT_OK;
        const unsigned long long cap = ctx->dense_rows_cap > 0 ? (unsigned long long)ctx->dense_rows_cap
            : (unsigned long long)std::min<double>(double(n) * double(n), std::max<double>(double(n) * 1024.0, double(1 << 24)));
        GT_HIP(ctx, st.own_sum.reserve(size_t(n) * sizeof(double)));
        GT_HIP(ctx, st.tri_i.reserve(size_t(cap) * sizeof(uint32_t)));
        GT_HIP(ctx, st.tri_j.reserve(size_t(cap) * sizeof(uint32_t)));
        GT_HIP(ctx, st.tri_a.reserve(size_t(cap) * sizeof(float)));
        GT_HIP(ctx, st.tri_cursor.reserve(sizeof(unsigned long long)));
        GT_HIP(ctx, hipMemsetAsync(st.tri_cursor.p, 0, sizeof(unsigned long long), ctx->stream));
        GT_HIP(ctx, st.scan_rows.reserve(size_t(n + 1) * sizeof(uint32_t)));
        GT_HIP(ctx, hipMemsetAsync(st.scan_rows.as<uint32_t>() + n, 0, sizeof(uint32_t), ctx->stream));
        tl.i = st.tri_i.as<uint32_t>();
        tl.j = st.tri_j.as<uint32_t>();
        tl.a = st.tri_a.as<float>();
        tl.cursor = st.tri_cursor.as<unsigned long long>();
        tl.cap = cap;
        GT_HIP(ctx, st.own_base.reserve(size_t(n) * sizeof(unsigned long long)));
        GT_HIP(ctx, st.own_cnt.reserve(size_t(n) * sizeof(uint32_t)));
        tl.own_base = st.own_base.as<unsigned long long>();
        tl.own_cnt = st.own_cnt.as<uint32_t>();
        return GT_OK;
    };
    // ---- bandwidth ----
    if (!pass) {
        StageSpan span(ctx, "dense_bandwidth");
        if (bandwidth_len > 0) {
            DENSE_HIP(st.bw_user.reserve(size_t(bandwidth_len) * sizeof(double)));
            DENSE_HIP(hipMemcpyAsync(st.bw_user.p, bandwidth, size_t(bandwidth_len) * sizeof(double),
                                     hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(dense_user_bandwidth_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream,
                               st.bw_user.as<double>(), bandwidth_len, n, bandwidth_scale, st.bw.as<double>());
        } else if (!precomputed) {
            if (ctx->DP == 0) {
                ctx->set_error("exact graph from data: this feature count is not supported on the HIP path (reduce with n_pca)");
                cleanup();
                return GT_E_LIMIT;
            }
            DENSE_TRY(gt_knn_candidates(ctx, 0, n, false, knn + 1));
            KnnWork* k = ctx->knn;
            if (dtype == GT_F32)
                hipLaunchKernelGGL(dense_bandwidth_from_knn_kernel<float>, dim3((unsigned)ceil_div64(n, 256)), dim3(256),
                                   0, ctx->stream, (const float*)in_dev, n, d, k->cand_j.as<uint32_t>(), k->MP, knn + 1,
                                   bandwidth_scale, st.bw.as<double>());
            else
                hipLaunchKernelGGL(dense_bandwidth_from_knn_kernel<double>, dim3((unsigned)ceil_div64(n, 256)), dim3(256),
                                   0, ctx->stream, (const double*)in_dev, n, d, k->cand_j.as<uint32_t>(), k->MP, knn + 1,
                                   bandwidth_scale, st.bw.as<double>());
        } else {
            const int kth = knn + 1;
            DENSE_HIP(st.redo.reserve(sizeof(uint32_t)));
            DENSE_HIP(hipMemsetAsync(st.redo.p, 0, sizeof(uint32_t), ctx->stream));
            uint32_t n_redo = (kth > 256) ? 1u : 0u;
            if (kth <= 256) {
                // one streaming read of the matrix
                if (dtype == GT_F32 && rows_ok && ctx->dense_rows_fused != 0) {
                    // the row-streaming form follows: this pass also lists the rows' kept affinities (no second read for them)
                    DENSE_TRY(rows_alloc());
                    EmitArgs em;
                    const double xc = dense_xcut(decay, thresh, true);
                    em.rfac = float(bandwidth_scale * xc * 1.0001);
                    em.decay = float(decay);
                    em.thresh = float(thresh);
                    em.xcut = float(xc);
                    em.own_sum = st.own_sum.as<double>();
                    em.tl = tl;
                    em.flags = st.flags.as<uint32_t>();
                    em.scan_rows = st.scan_rows.as<uint32_t>();
                    em.scan_count = st.scan_rows.as<uint32_t>() + n;
                    hipLaunchKernelGGL((dense_bandwidth_1pass_kernel<float, true>), dim3((unsigned)n), dim3(256), 0, ctx->stream,
                                       (const float*)in_dev, n, kth, bandwidth_scale, st.bw.as<double>(), st.redo.as<uint32_t>(), em);
                    rows_listed = true;
                } else if (dtype == GT_F32) {
                    hipLaunchKernelGGL((dense_bandwidth_1pass_kernel<float, false>), dim3((unsigned)n), dim3(256), 0, ctx->stream,
                                       (const float*)in_dev, n, kth, bandwidth_scale, st.bw.as<double>(), st.redo.as<uint32_t>(), EmitArgs());
                } else {
                    hipLaunchKernelGGL((dense_bandwidth_1pass_kernel<double, false>), dim3((unsigned)n), dim3(256), 0, ctx->stream,
                                       (const double*)in_dev, n, kth, bandwidth_scale, st.bw.as<double>(), st.redo.as<uint32_t>(), EmitArgs());
                }
                DENSE_HIP(hipGetLastError());
                DENSE_HIP(hipMemcpyAsync(&n_redo, st.redo.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                DENSE_HIP(hipStreamSynchronize(ctx->stream));
            }
            if (n_redo > 0) {
                const int only = (kth <= 256) ? 1 : 0;
                if (dtype == GT_F32)
                    hipLaunchKernelGGL(dense_bandwidth_generic_kernel<float>, dim3((unsigned)n), dim3(256), 0, ctx->stream,
                                       (const float*)in_dev, n, kth, bandwidth_scale, st.bw.as<double>(), only);
                else
                    hipLaunchKernelGGL(dense_bandwidth_generic_kernel<double>, dim3((unsigned)n), dim3(256), 0, ctx->stream,
                                       (const double*)in_dev, n, kth, bandwidth_scale, st.bw.as<double>(), only);
            }
        }
        DENSE_HIP(hipGetLastError());
    }
    // ---- kernel tiles ----
    bool fused_rowsum = false;
    void* K_dev = nullptr;
    bool rows_done = false, rows_wrote_p = false;
    if (rows_ok) {
        const bool direct_p = out_P && !out_K;   // the operator alone
        void* target = nullptr;
        if (direct_p) {
            if (out_on_device) {
                target = out_P;
            } else {
                DENSE_HIP(st.work_p.reserve(size_t(n) * n * out_esz));
                target = st.work_p.p;
            }
        } else if (inplace) {
            target = const_cast<void*>(in_dev);
        } else if (out_K && out_on_device) {
            target = out_K;
        } else {
            DENSE_HIP(st.work_k.reserve(size_t(n) * n * out_esz));
            target = st.work_k.p;
        }
        std::unique_ptr<StageSpan> span(new StageSpan(ctx, "dense_rows_scan"));
        if (rows_listed) StageSpan mark(ctx, "dense_rows_listed");   // (the bandwidth pass listed the kept affinities: no pass over the matrix here)
        DENSE_TRY(rows_alloc());
        const unsigned long long cap = tl.cap;
        DENSE_HIP(st.incount.reserve(size_t(n) * 2 * sizeof(uint32_t)));   // counts, then the scatter's cursors
        DENSE_HIP(hipMemsetAsync(st.incount.p, 0, size_t(n) * 2 * sizeof(uint32_t), ctx->stream));
        DENSE_HIP(st.inptr.reserve(size_t(n + 1) * sizeof(unsigned long long)));
        DENSE_HIP(st.rowsum.reserve(size_t(n) * sizeof(double)));
        const double xc = dense_xcut(decay, thresh, true);
        if (!rows_listed) {
            hipLaunchKernelGGL(dense_rows_scan_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, (const float*)in_dev, n,
                               st.bw.as<double>(), decay, thresh, xc, st.own_sum.as<double>(), tl, st.flags.as<uint32_t>(),
                               (const uint32_t*)nullptr);
        } else {
            // (listed by the bandwidth pass; the rows it gave up on - bandwidths from the generic kernel by now - are scanned here)
            uint32_t n_scan = 0;
            DENSE_HIP(hipMemcpyAsync(&n_scan, st.scan_rows.as<uint32_t>() + n, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            DENSE_HIP(hipStreamSynchronize(ctx->stream));
            if (n_scan > 0)
                hipLaunchKernelGGL(dense_rows_scan_kernel, dim3(n_scan), dim3(256), 0, ctx->stream, (const float*)in_dev, n,
                                   st.bw.as<double>(), decay, thresh, xc, st.own_sum.as<double>(), tl, st.flags.as<uint32_t>(),
                                   st.scan_rows.as<uint32_t>());
        }
        DENSE_HIP(hipGetLastError());
        unsigned long long total = 0;
        DENSE_HIP(hipMemcpyAsync(&total, st.tri_cursor.p, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
        DENSE_HIP(hipStreamSynchronize(ctx->stream));
        span.reset();
        if (total <= cap) {
            StageSpan span2(ctx, "dense_kernel");
            DENSE_HIP(st.in_col.reserve(size_t(std::max<unsigned long long>(total, 1)) * sizeof(uint32_t)));
            DENSE_HIP(st.in_val.reserve(size_t(std::max<unsigned long long>(total, 1)) * sizeof(float)));
            uint32_t* incount = st.incount.as<uint32_t>();
            std::unique_ptr<StageSpan> tspan(new StageSpan(ctx, "dense_rows_transpose"));
            if (total > 0)
                hipLaunchKernelGGL(tri_count_kernel, dim3((unsigned)ceil_div64(int64_t(total), 256)), dim3(256), 0, ctx->stream,
                                   tl.j, int64_t(total), incount);
            hipLaunchKernelGGL(rows_scan_counts_kernel, dim3(1), dim3(1024), 0, ctx->stream, incount, n,
                               st.inptr.as<unsigned long long>());
            if (total > 0)
                hipLaunchKernelGGL(tri_scatter_kernel, dim3((unsigned)ceil_div64(int64_t(total), 256)), dim3(256), 0, ctx->stream,
                                   tl.i, tl.j, tl.a, int64_t(total), st.inptr.as<unsigned long long>(), incount + n,
                                   st.in_col.as<uint32_t>(), st.in_val.as<float>());
            tspan.reset();
#define GT_ROWS_PART(DIV_, PART_)                                                                                              \
    hipLaunchKernelGGL((dense_rows_write_kernel<DIV_, PART_>), dim3((unsigned)n), dim3(256), 0, ctx->stream, (const float*)in_dev, n, \
                       st.bw.as<double>(), decay, thresh, xc, st.own_sum.as<double>(), st.inptr.as<unsigned long long>(),     \
                       st.in_col.as<uint32_t>(), st.in_val.as<float>(), st.rowsum.as<double>(), (float*)target, tl)
            {
                StageSpan mark(ctx, "dense_rows_placed");   // (marker: the write pass reads no row - 4 N^2 bytes, not 8)
                if (direct_p) GT_ROWS_PART(true, 1); else GT_ROWS_PART(false, 1);
                hipLaunchKernelGGL(dense_zero_rows_kernel, dim3((unsigned)std::min<int64_t>(n, int64_t(ctx->n_cu) * 16)), dim3(256), 0,
                                   ctx->stream, (float*)target, n, tl.own_cnt);
                if (direct_p) GT_ROWS_PART(true, 2); else GT_ROWS_PART(false, 2);
            }
#undef GT_ROWS_PART
            DENSE_HIP(hipGetLastError());
            rows_done = true;
            rows_wrote_p = direct_p;
            fused_rowsum = true;   // (the row sums are final)
            K_dev = direct_p ? nullptr : target;
        } else {
            // more kept entries than the list was sized for (a kernel that is not sparse): the tile-pair form below; the matrix
            // is untouched so far.  (the zero-diagonal flag may have been raised by rows the scan saw: recomputed there)
            DENSE_HIP(hipMemsetAsync(st.flags.p, 0, sizeof(uint32_t), ctx->stream));
        }
        for (DevBuf* b : {&st.tri_i, &st.tri_j, &st.tri_a}) b->release();
    }
    if (rows_done) {
        // (nothing left for the tile kernels)
    } else if (inplace) {
        K_dev = const_cast<void*>(in_dev);
    } else if (out_K && out_on_device) {
        K_dev = out_K;
    } else {
        DENSE_HIP(st.work_k.reserve(size_t(n) * n * out_esz));
        K_dev = st.work_k.p;
    }
    if (!rows_done) {
        StageSpan span(ctx, "dense_kernel");
        uint32_t* fl = st.flags.as<uint32_t>();
        const double* bw = st.bw.as<double>();
        if (!precomputed) {
            if (dtype == GT_F32)
                DENSE_TRY((launch_tiles<double, double, float, true>(ctx, nullptr, (const float*)in_dev, d, n, bw, decay,
                                                                     thresh, kernel_symm, theta, (double*)K_dev, fl)));
            else
                DENSE_TRY((launch_tiles<double, double, double, true>(ctx, nullptr, (const double*)in_dev, d, n, bw, decay,
                                                                      thresh, kernel_symm, theta, (double*)K_dev, fl)));
        } else if (dtype == GT_F64) {
            DENSE_TRY((launch_tiles<double, double, double, false>(ctx, (const double*)in_dev, nullptr, 0, n, bw, decay,
                                                                   thresh, kernel_symm, theta, (double*)K_dev, fl, pass)));
        } else if (out_f64) {
            DENSE_TRY((launch_tiles<float, double, double, false>(ctx, (const float*)in_dev, nullptr, 0, n, bw, decay,
                                                                  thresh, kernel_symm, theta, (double*)K_dev, fl, pass)));
        } else {
            // float32 in, float32 out, whole quads, no anisotropy: the tile kernel accumulates the row sums as it writes K
            fused_rowsum = anisotropy == 0.0 && (n % 4) == 0 && ctx->dense_fused_rowsum != 0;
            if (fused_rowsum) {
                DENSE_HIP(st.rowsum.reserve(size_t(n) * sizeof(double)));
                DENSE_HIP(hipMemsetAsync(st.rowsum.p, 0, size_t(n) * sizeof(double), ctx->stream));
            }
            DENSE_TRY((launch_tiles<float, float, double, false>(ctx, (const float*)in_dev, nullptr, 0, n, bw, decay,
                                                                 thresh, kernel_symm, theta, (float*)K_dev, fl, pass,
                                                                 fused_rowsum ? st.rowsum.as<double>() : nullptr)));
        }
    }
    // ---- anisotropy + P ----
    void* P_dev = nullptr;
    if (out_P) {
        if (out_on_device) {
            P_dev = out_P;
        } else {
            DENSE_HIP(st.work_p.reserve(size_t(n) * n * out_esz));
            P_dev = st.work_p.p;
        }
    }
    if (out_f64)
        DENSE_TRY(finish_dense<double>(ctx, st, (double*)K_dev, (double*)P_dev, n, anisotropy));
    else if (!rows_wrote_p)   // (the row-streaming form for P alone: P is written already)
        DENSE_TRY(finish_dense<float>(ctx, st, (float*)K_dev, (float*)P_dev, n, anisotropy, fused_rowsum));
    if (out_K && !out_on_device) DENSE_TRY(gt_copy_to_host(ctx, out_K, K_dev, size_t(n) * n * out_esz));
    if (out_P && !out_on_device) DENSE_TRY(gt_copy_to_host(ctx, out_P, P_dev, size_t(n) * n * out_esz));
    uint32_t fl = 0;
    DENSE_HIP(hipMemcpyAsync(&fl, st.flags.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    DENSE_HIP(hipStreamSynchronize(ctx->stream));
    if (!precomputed && ctx->knn && ctx->knn->gflags.p) {   // (a search that was refused before its first launch leaves no flags)
        uint32_t kfl = 0;
        DENSE_HIP(hipMemcpy(&kfl, ctx->knn->gflags.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        fl |= (kfl & GT_FLAG_DUPLICATES);
    }
    if (flags) *flags = fl;
    // keep degree (row sums of K) and bandwidth for gt_dense_fetch_vec
    ctx->dense_n = n;
    {
        hipError_t e1 = ctx->dense_degree.reserve(size_t(n) * sizeof(double));
        hipError_t e2 = ctx->dense_bw.reserve(size_t(n) * sizeof(double));
        if (e1 == hipSuccess && e2 == hipSuccess) {
            (void)hipMemcpy(ctx->dense_degree.p, st.rowsum.p, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice);
            (void)hipMemcpy(ctx->dense_bw.p, st.bw.p, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice);
        }
    }
    cleanup();
    return GT_OK;
#undef DENSE_TRY
#undef DENSE_HIP
}

extern "C" int gt_dense_extend(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device, int32_t knn, double decay,
                               double thresh, const double* bandwidth, int64_t bandwidth_len, double bandwidth_scale,
                               double* out_K, int32_t out_on_device) {
    if (!ctx || !Y || m <= 0 || !out_K) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "gt_dense_extend: no points bound (build the graph from data first)");
    if (ctx->metric != 0) GT_FAIL(ctx, GT_E_LIMIT, "gt_dense_extend: euclidean metric only");
    if (std::isnan(decay)) GT_FAIL(ctx, GT_E_ARG, "`decay` must be provided for a TraditionalGraph");
    if (bandwidth_len != 0 && bandwidth_len != 1 && bandwidth_len != m)
        GT_FAIL(ctx, GT_E_ARG, "bandwidth must have 1 or n_samples_y entries");
    if (bandwidth_len == 0 && (knn < 1 || int64_t(knn) > ctx->n)) GT_FAIL(ctx, GT_E_ARG, "knn must be in [1, n_samples]");
    const int64_t n = ctx->n;
    const int d = ctx->d;
    DevBuf bw, bw_user, kbuf;
    int rc = GT_OK;
    auto cleanup = [&]() {
        bw.release();
        bw_user.release();
        kbuf.release();
    };
#define EXT_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            ctx->set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
            cleanup();                                                         \
            return GT_E_HIP;                                                   \
        }                                                                      \
    } while (0)
#define EXT_TRY(expr)      \
    do {                   \
        rc = (expr);       \
        if (rc != GT_OK) { \
            cleanup();     \
            return rc;     \
        }                  \
    } while (0)
    // the query matrix on the device, in the points' dtype (also the working copy the candidate pass needs)
    EXT_TRY(gt_prepare_queries(ctx, Y, m, y_on_device));
    KnnWork* k = ctx->knn;
    EXT_HIP(bw.reserve(size_t(m) * sizeof(double)));
    {
        StageSpan span(ctx, "dense_bandwidth");
        if (bandwidth_len > 0) {
            EXT_HIP(bw_user.reserve(size_t(bandwidth_len) * sizeof(double)));
            EXT_HIP(hipMemcpyAsync(bw_user.p, bandwidth, size_t(bandwidth_len) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(dense_user_bandwidth_kernel, dim3((unsigned)ceil_div64(m, 256)), dim3(256), 0, ctx->stream,
                               bw_user.as<double>(), bandwidth_len, m, bandwidth_scale, bw.as<double>());
        } else {
            if (ctx->DP == 0) {
                ctx->set_error("exact graph extension: this feature count is not supported on the HIP path (reduce with n_pca)");
                cleanup();
                return GT_E_LIMIT;
            }
            EXT_TRY(gt_knn_candidates(ctx, 0, m, true, knn));
            k = ctx->knn;
            if (ctx->dtype == GT_F32)
                hipLaunchKernelGGL(dense_bandwidth_ext_kernel<float>, dim3((unsigned)ceil_div64(m, 256)), dim3(256), 0, ctx->stream,
                                   (const float*)k->Qraw.p, m, (const float*)ctx->X, d, k->cand_j.as<uint32_t>(), k->MP, knn,
                                   bandwidth_scale, bw.as<double>());
            else
                hipLaunchKernelGGL(dense_bandwidth_ext_kernel<double>, dim3((unsigned)ceil_div64(m, 256)), dim3(256), 0, ctx->stream,
                                   (const double*)k->Qraw.p, m, (const double*)ctx->X, d, k->cand_j.as<uint32_t>(), k->MP, knn,
                                   bandwidth_scale, bw.as<double>());
        }
        EXT_HIP(hipGetLastError());
    }
    double* K_dev = out_K;
    if (!out_on_device) {
        EXT_HIP(kbuf.reserve(size_t(m) * size_t(n) * sizeof(double)));
        K_dev = kbuf.as<double>();
    }
    {
        StageSpan span(ctx, "dense_kernel");
        const dim3 grid((unsigned)ceil_div64(n, TS), (unsigned)ceil_div64(m, TS));
        const double xcut = dense_xcut(decay, thresh, false);
        if (ctx->dtype == GT_F32)
            hipLaunchKernelGGL(dense_extend_tiles<float>, grid, dim3(256), 0, ctx->stream, (const float*)k->Qraw.p, m,
                               (const float*)ctx->X, n, d, bw.as<double>(), decay, thresh, xcut, K_dev);
        else
            hipLaunchKernelGGL(dense_extend_tiles<double>, grid, dim3(256), 0, ctx->stream, (const double*)k->Qraw.p, m,
                               (const double*)ctx->X, n, d, bw.as<double>(), decay, thresh, xcut, K_dev);
        EXT_HIP(hipGetLastError());
    }
    if (!out_on_device) EXT_TRY(gt_copy_to_host(ctx, out_K, K_dev, size_t(m) * size_t(n) * sizeof(double)));
    EXT_HIP(hipStreamSynchronize(ctx->stream));
    cleanup();
    return GT_OK;
#undef EXT_HIP
#undef EXT_TRY
}

extern "C" int gt_dense_fetch_vec(gt_ctx* ctx, int32_t which, double* out_host) {
    if (!ctx || !out_host) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->dense_n <= 0) GT_FAIL(ctx, GT_E_STATE, "gt_dense_fetch_vec: no dense graph built");
    const void* src = which == GT_VEC_BANDWIDTH ? ctx->dense_bw.p : ctx->dense_degree.p;
    GT_HIP(ctx, hipMemcpy(out_host, src, size_t(ctx->dense_n) * sizeof(double), hipMemcpyDeviceToHost));
    return GT_OK;
}
