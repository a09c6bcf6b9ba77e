// Sparse tail of the kNN graph build: bandwidths, radius pass bookkeeping, alpha-decay affinities with
// thresholding, symmetrisation (transpose by triplet exchange + per-row sort/merge), anisotropy and the
// row-stochastic diffusion operator.  All HBM-bound streaming kernels; one wave per matrix row.
//
// Reference semantics (graphtools/graphs.py:886-911, 450-559; base.py:557-592, 629-666):
//   bw_i  = max(D[i, knn] * scale, eps)            (or the user bandwidth * scale)
//   r_i   = bw_i * (-ln thresh)^(1/decay)
//   K0_ij = exp(-(D_ij / bw_i)^decay), NaN -> 1, kept iff >= thresh, for every j with D_ij <= r_i
//   K     = sym(K0);  K_ij /= (q_i q_j)^alpha;  P = diag(1/sum_j|K_ij|) K;  degree = K 1
#include <cfloat>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "gt_common.h"
#include "gt_hostcopy.h"
#include "gt_device.h"
#include "gt_knn.h"
#include "gt_knn_select.h"
#include "gt_graph_state.h"

void gt_free_graph_state(gt_ctx* ctx) {
    if (!ctx->graph) return;
    GraphState* g = ctx->graph;
    for (DevBuf* b : {&g->bw, &g->rec_s, &g->bwpos, &g->sC, &g->midrows, &g->bw_user, &g->rowsrc, &g->hugerows, &g->tablen, &g->spmm_in, &g->spmm_out, &g->lenN, &g->lenT, &g->cursor, &g->off, &g->outlen, &g->indptr,
                      &g->degree, &g->over_rows, &g->over_count, &g->rthr, &g->rlists, &g->rcounts, &g->rK, &g->rmax,
                      &g->ownercnt, &g->ownerpos, &g->cnt_sorted, &g->pos_sorted, &g->scan_tmp, &g->selfbuf, &g->splits_dev, &g->edges, &g->Ukey, &g->Uval, &g->Vkey, &g->Vval,
                      &g->bincnt, &g->binoff, &g->ucol, &g->uval, &g->bigrows, &g->bigcount, &g->bigscratch_k, &g->bigscratch_v, &g->bigsoff, &g->aniso_tmp, &g->scan_own,
                      &g->indices, &g->Kdata, &g->Pdata, &g->flags, &g->deg_caller})
        b->release();
    delete g;
    ctx->graph = nullptr;
}

// internal bit of GraphState::flags (never handed to the caller): merge_pairs_slots_kernel saw the same column twice in a union row
static constexpr uint32_t kFlagPairDupColumn = 0x40000000u;

namespace {

constexpr int kMaxWorld = 64;

struct Splits {
    int64_t s[kMaxWorld + 1];
    int world;
};

__device__ __forceinline__ int owner_of(const Splits& sp, int64_t row) {
    int o = 0;
    for (int r = 1; r < sp.world; ++r) o += (row >= sp.s[r]) ? 1 : 0;
    return o;
}

// dtype the distances are formed in from the float64 keys: the points' own (scikit-learn rounds a float32 set's distances to
// float32, graphs.py:883), or float64 whatever the points are ("distance_dtype" = "float64": scipy pdist / cdist semantics,
// what the exact graph built through this path needs - graphs.py:1552)
static inline int gt_dist_dtype(const gt_ctx* ctx) { return ctx->dist_f64 ? GT_F64 : ctx->dtype; }

// the symmetrisation rule on one pair of values (base.py:557-577)
__device__ __forceinline__ double merge_values(double a, double b, int symm, double theta) {
    switch (symm) {
        case GT_SYMM_ADD: return (a + b) / 2;
        case GT_SYMM_MUL: return a * b;
        case GT_SYMM_MNN: return theta * fmin(a, b) + (1 - theta) * fmax(a, b);
        default: return a;
    }
}

// idecay > 0: `decay` is that whole number (1 ... 128, the usual case: the default is 40) and the power is formed by
// repeated squaring - a handful of multiplications instead of the library pow, which was most of the affinity kernel's
// instructions (PMC: VALU busy 88 %).  The squarings' roundings add up to < 2^(bits of idecay) ulp of the power, i.e. a
// relative 1e-14 of exp(-power) at the threshold: far inside the 1e-5 the values are held to, and the same function is
// used wherever an affinity is formed, so every path agrees bit for bit.
__device__ __forceinline__ double affinity(double dist, double bw, double decay, int idecay) {
    const double x = dist / bw;
    double p;
    if (idecay > 0) {   // (uniform)
        double b = x;
        p = (idecay & 1) ? x : 1.0;
        for (int n = idecay >> 1; n; n >>= 1) {
            b *= b;
            if (n & 1) p *= b;
        }
    } else {
        p = pow(x, decay);
    }
    double w = exp(-p);
    return (w != w) ? 1.0 : w;   // NaN -> 1 (graphs.py:505-506)
}
__host__ __device__ inline int gt_whole_decay(double decay) {
    return (decay >= 1.0 && decay <= 128.0 && decay == double(int(decay))) ? int(decay) : 0;
}

constexpr uint32_t kNoDest = 0xFFFFFFFFu;   // posj of a kept entry that is not sent (destination bins, below)

// Tables by sorted position (KnnWork::tab_sorted): what affinity_slots_kernel reads per slot and per partner, one record each
struct SlotRec {      // by slot
    int32_t i;        // the row; -1: not that launch's row (a row of the radius pass, a table from a repair pass)
    uint32_t n;       // entries of its table
    double bwi;       // its bandwidth
};
struct BwPos {        // by row: ONE gather per table entry fetches the partner's bandwidth and its sorted position
    double bw;
    uint32_t pos, pad;
};

// ---- A0: bandwidth, radius, classification ------------------------------------------------------
__global__ __launch_bounds__(256) void bandwidth_kernel(
    const int64_t nloc, const int64_t r0, const int MP, const int kprime, const int dtype, const int metric,
    const double* __restrict__ cand_d2,
    const double* __restrict__ d2_lb, const double* __restrict__ qnorm, const double* __restrict__ qnorm_full,
    const int64_t qoff, const double* __restrict__ ymax2p,
    const ErrModel err, const double* __restrict__ bw_user, const int64_t bw_len, const double bw_scale,
    const int use_radius, const double radius_factor, double* __restrict__ bw_out, int32_t* __restrict__ rowsrc,
    int32_t* __restrict__ over_rows, uint32_t* __restrict__ over_count, float* __restrict__ rthr,
    const int32_t* __restrict__ trow, SlotRec* __restrict__ rec_s, BwPos* __restrict__ bwpos,
    const uint32_t* __restrict__ cand_n, const uint8_t* __restrict__ keyt_ok) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= nloc) return;
    const int64_t ti = trow ? int64_t(trow[i]) : i;   // (KnnWork::tab_sorted: the row's table is at its sorted position)
    double bw;
    if (bw_len == 0)
        bw = gt_key_to_dist(cand_d2[ti * MP + (kprime - 1)], dtype, metric) * bw_scale;
    else
        bw = (bw_len == 1 ? bw_user[0] : bw_user[r0 + i]) * bw_scale;
    bw = fmax(bw, DBL_EPSILON);
    bw_out[i] = bw;
    int32_t src = -1;
    if (use_radius) {
        const double r = bw * radius_factor * (1.0 + 1e-6);
        const double r2 = (metric == 1) ? r : r * r;   // key space: squared distance / cosine distance
        if (!(r2 < d2_lb[ti])) {
            const uint32_t slot = atomicAdd(over_count, 1u);
            over_rows[slot] = int32_t(qoff + i);
            src = int32_t(slot);
            const double qn = qnorm[qoff + i];
            const double y2 = ymax2p[0];
            const double e = gt_err_bound(err, qn, y2);
            // every row within the radius scores at least this much (scaled score units; qn, y2: norms over the
            // scored columns, the cosine bound also needs the full ones - see bound_of_score in rerank_kernel)
            const double smin = (metric == 1) ? (1.0 - r2 - 0.5 * (qnorm_full[qoff + i] - qn) - 0.5 * ymax2p[1])
                                              : 0.5 * (qn - r2);
            const double x = (smin - e - 1e-9 * (qn + y2)) / err.inv_sc2;
            float f = float(x);
            if (double(f) >= x) f = nextafterf(f, -INFINITY);
            rthr[slot] = f;
        }
    }
    rowsrc[i] = src;
    if (rec_s) {
        SlotRec r;
        r.i = (src < 0 && keyt_ok[ti] != 0) ? int32_t(i) : -1;
        r.n = cand_n[ti];
        r.bwi = bw;
        rec_s[ti] = r;
        BwPos b;
        b.bw = bw;
        b.pos = uint32_t(ti);
        b.pad = 0u;
        bwpos[i] = b;
    }
}

__global__ void max_u32_kernel(const uint32_t* __restrict__ v, const int64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) m = max(m, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// knn_max: rows whose s-th nearest neighbour still lies inside their radius, for the (up to 4) table sizes s the
// reference's search-expansion loop would try (graphs.py:916-940: "update_idx" after a search with search_knn = s).
// Distances and radii are compared as the reference holds them (float64 images of the input-dtype distances).
__global__ __launch_bounds__(256) void stage_count_kernel(const int64_t nloc, const int MP, const int dtype, const int metric,
                                                          const double* __restrict__ cand_d2, const double* __restrict__ bw,
                                                          const double radius_factor, const int4 stages, const int nstage,
                                                          uint32_t* __restrict__ counts) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int st[4] = {stages.x, stages.y, stages.z, stages.w};
    bool un[4] = {false, false, false, false};
    if (i < nloc) {
        const double r = bw[i] * radius_factor;
        for (int t = 0; t < nstage; ++t) un[t] = gt_key_to_dist(cand_d2[i * MP + (st[t] - 1)], dtype, metric) < r;
    }
    for (int t = 0; t < nstage; ++t) {
        const unsigned long long m = __ballot(un[t]);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counts[t], uint32_t(__popcll(m)));
    }
}

// ---- A1: affinities + counts ----------------------------------------------------------------------
// One wave per local row.  Table rows: K overwrites cand_d2 in place (-1 marks a dropped slot).
// Radius rows: exact float64 distance for every collected candidate, K into rK.
// RADIUS: the launch covers the rows of the radius pass only (row_list, nloc of them) - their exact float64 distances
// need registers the table rows of the main launch should not pay for with occupancy
// PAIRS (table rows): 0 the plain kernel, 1 pair-resolved rows whose tables carry the transposed keys, 2 pair-resolved rows
// whose tables came from a repair pass (the few: transposed keys from the dot products); a launch of 1 or 2 skips the other's rows
template <typename T, bool RADIUS, int PAIRS>
__global__ __launch_bounds__(256) void affinity_kernel(
    const int32_t* __restrict__ row_list, const int64_t nrows, const int64_t nloc, const int64_t r0, const T* __restrict__ X, const int d, const double* __restrict__ xn,
    const T* __restrict__ Qm, const double* __restrict__ qnorm, const int64_t qoff, const int dtype, const int metric, const int MP, const int limit, double* __restrict__ cand_d2, const uint32_t* __restrict__ cand_j,
    const uint32_t* __restrict__ cand_n, const int32_t* __restrict__ rowsrc, const uint64_t* __restrict__ rlists,
    const uint32_t* __restrict__ rcounts, const int32_t rcap, double* __restrict__ rK, const double* __restrict__ bw,
    const double decay, const int binary, const double thresh, const int count_owners, const Splits sp,
    int32_t* __restrict__ lenN, int32_t* __restrict__ ownercnt, int32_t* __restrict__ tablen,
    const int pairs, const double* __restrict__ cand_d2t, const uint8_t* __restrict__ keyt_ok, const double rf_guard,
    const int32_t* __restrict__ tperm, const int32_t* __restrict__ trow, uint32_t* __restrict__ posj,
    const int64_t* __restrict__ sC, int32_t* __restrict__ lenN_s, const int64_t bw_i0, const int self_rank) {
    // bw_i0: where this launch's row 0 sits in bw (0: bw holds the launch's rows - one rank; r0: bw holds the bandwidths of ALL
    // rows, gathered over the ranks - the pair-resolved tail of a row-sharded build, `pairs` = 2: partners j are global rows, and
    // only the entries that TRAVEL - the one-sided ones, stored non-negative - are counted per owner; `pairs` = 3: ... and of those
    // only the ones whose partner lives on ANOTHER rank (self_rank: this one) - the local ones go through the rank's own
    // destination bins, their destination - the partner's local row - noted in posj like the single-rank build's)
    // posj / sC (with trow, table rows): the destinations of the pair-resolved tail are looked up here (see
    // affinity_slots_kernel: this launch serves the few rows whose tables came from a repair pass)
    // tperm / trow (KnnWork::tab_sorted): the tables lie by sorted position - tperm: slot -> row, for the launch over all the
    // rows (wave wi takes slot wi: the tables stream, and the partners whose bandwidths a stretch of slots gathers are a
    // cache-resident neighbourhood); trow: row -> slot, for the listed rows
    // pairs (single-rank builds with the '+' rule, see graph_finish_pairs): the row settles its MUTUAL pairs itself.  For a
    // kept entry (i, j) the value row j gives the same pair, K0(j, i), follows from the key row j holds for it - cand_d2t,
    // or the dot product where the table came from a repair pass - and row j's bandwidth, through the very function row j
    // uses: when it passes the threshold too the entry is stored as MINUS the merged value (final; nothing is sent), else
    // as the plain value (the transposed half travels to row j as before).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    double* xs = reinterpret_cast<double*>(smem_raw) + size_t(w) * d;
    const int idecay = gt_whole_decay(decay);
    const int64_t wi = int64_t(blockIdx.x) * (blockDim.x >> 6) + w;
    if (wi >= nrows) return;
    const int64_t i = row_list ? int64_t(row_list[wi]) - qoff : (tperm ? int64_t(tperm[wi]) : wi);   // (a list holds rows of the query matrix: qoff + i)
    const int64_t ti = row_list ? (trow ? int64_t(trow[i]) : i) : (tperm ? wi : i);                  // where the row's table is
    const int32_t src = rowsrc[i];
    if (RADIUS != (src >= 0)) return;   // (the other launch's row)
    if (!RADIUS && PAIRS != 0 && (PAIRS == 1) != (keyt_ok[ti] != 0)) return;   // (the other pair-resolving launch's row)
    const double bwi = bw[bw_i0 + i];
    int kept = 0;
    int owner_cnt = 0;   // lane o accumulates the count for owner o (world <= 64)
    if constexpr (!RADIUS) {
        uint32_t n = cand_n[ti];
        if (n > uint32_t(limit)) n = uint32_t(limit);
        uint32_t* cand_j_w = const_cast<uint32_t*>(cand_j);
        constexpr bool have_t = PAIRS == 1;
        double qn_i = 0.0;
        if constexpr (PAIRS == 2) {
            // a table from a repair pass: the transposed keys are formed from the dot products (the query row in the LDS)
            const T* xrow = Qm + (qoff + i) * int64_t(d);
            for (int k = lane; k < d; k += 64) xs[k] = double(xrow[k]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            qn_i = qnorm[qoff + i];
        }
        // one chunk of 64 entries, its loads done: e = entry of this lane, (d2, j) its key and column, d2t / bwj the transposed
        // key and the partner's bandwidth where the caller fetched them ahead (PAIRS == 1)
        auto chunk = [&](const uint32_t e, const double d2, const uint32_t j, double d2t, double bwj) {
            bool keep = false;
            double kvv = -1.0;
            if (e < n) {
                double kv = 1.0;
                if (!binary) kv = affinity(gt_key_to_dist(d2, dtype, metric), bwi, decay, idecay);
                keep = binary || (kv >= thresh);
                kvv = kv;
                if constexpr (PAIRS != 0) {
                    // (row j drops the pair when its distance exceeds bw_j x radius_factor: beyond that by a relative 1e-9 -
                    //  six orders above the rounding of the affinity, which falls monotonically with the distance - the
                    //  transposed value need not be formed to know that it is below the threshold)
                    if (keep) {
                        if constexpr (!have_t) {
                            const double dot = gt_dot16(xs, X + int64_t(j) * d, d);
                            d2t = gt_pair_key(xn[j], dot, qn_i, metric);
                            bwj = bw[j];
                        }
                        const double dt = gt_key_to_dist(d2t, dtype, metric);
                        if (!(dt > bwj * rf_guard)) {
                            const double kvt = affinity(dt, bwj, decay, idecay);
                            if (kvt >= thresh) kvv = -merge_values(kv, kvt, GT_SYMM_ADD, 1.0);
                        }
                    }
                }
            }
            // the kept entries move to the front of the row, in order (in place: a chunk is read before it is written, and
            // only at or below where it was read) - the table is sorted by distance, so they are a prefix anyway and nothing
            // moves, but the passes over the rows rely on it
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const uint32_t slot = uint32_t(kept) + uint32_t(__popcll(km & ((1ull << lane) - 1ull)));
                cand_d2[ti * MP + slot] = kvv;
                if (slot != e) cand_j_w[ti * MP + slot] = j;
                if (posj) {
                    // (tables by sorted position: the partner's; a rank of a sharded build: the partner's local row, if it has one)
                    const int64_t jl = int64_t(j) - r0;
                    const uint32_t pj = trow ? uint32_t(trow[j]) : ((jl >= 0 && jl < nloc) ? uint32_t(jl) : kNoDest);
                    posj[sC[ti] + slot] = kvv >= 0.0 ? pj : kNoDest;
                }
            }
            kept += __popcll(km);
            if (count_owners) {
                int o = (keep && (pairs < 2 || kvv >= 0.0)) ? owner_of(sp, j) : -1;
                if (pairs == 3 && o == self_rank) o = -1;
                for (int r = 0; r < sp.world; ++r) {
                    const int c = __popcll(__ballot(o == r));
                    if (lane == r) owner_cnt += c;
                }
            }
        };
        if constexpr (have_t) {
            // The row is a chain of dependent reads (entry -> column -> the partner's bandwidth) and a million rows hide each
            // other's latency only eight to a SIMD: two chunks' keys, columns and transposed keys are fetched at once and the
            // partners' bandwidths right behind them - for every entry, kept or not (nine in ten are) - before any arithmetic.
            for (uint32_t e0 = 0; e0 < n; e0 += 128) {
                const uint32_t ea = e0 + lane, eb = ea + 64;
                double d2a = 0.0, d2b = 0.0, ta = 0.0, tb = 0.0, bja = 1.0, bjb = 1.0;
                uint32_t ja = 0, jb = 0;
                if (ea < n) {
                    d2a = cand_d2[ti * MP + ea];
                    ja = cand_j[ti * MP + ea];
                    ta = cand_d2t[ti * MP + ea];
                }
                if (eb < n) {
                    d2b = cand_d2[ti * MP + eb];
                    jb = cand_j[ti * MP + eb];
                    tb = cand_d2t[ti * MP + eb];
                }
                if (ea < n) bja = bw[ja];
                if (eb < n) bjb = bw[jb];
                chunk(ea, d2a, ja, ta, bja);
                if (e0 + 64 < n) chunk(eb, d2b, jb, tb, bjb);   // (uniform)
            }
        } else {
            for (uint32_t e0 = 0; e0 < n; e0 += 64) {
                const uint32_t e = e0 + lane;
                double d2 = 0.0;
                uint32_t j = 0;
                if (e < n) {
                    d2 = cand_d2[ti * MP + e];
                    j = cand_j[ti * MP + e];
                }
                chunk(e, d2, j, 0.0, 1.0);
            }
        }
        if (posj)   // (the row's stretch of posj is as long as its table: nothing travels from the slots behind the kept ones)
            for (int64_t e = sC[ti] + kept + lane; e < sC[ti + 1]; e += 64) posj[e] = kNoDest;
        // the consumers read the kept prefix and nothing else
        if (lane == 0) tablen[i] = int32_t(kept);
    } else {
        const T* xrow = Qm + (qoff + i) * int64_t(d);
        for (int k = lane; k < d; k += 64) xs[k] = double(xrow[k]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const double qn = qnorm[qoff + i];
        uint32_t n = rcounts[src];
        if (n > uint32_t(rcap)) n = uint32_t(rcap);
        uint64_t* lp = const_cast<uint64_t*>(rlists) + size_t(src) * rcap;
        double* kp = rK + size_t(src) * rcap;
        for (uint32_t e0 = 0; e0 < n; e0 += 64) {
            const uint32_t e = e0 + lane;
            bool keep = false;
            uint32_t j = 0;
            uint64_t lkey = 0ull;
            double kvv = -1.0;
            if (e < n) {
                lkey = lp[e];
                j = cand_index(lkey);
                const T* y = X + int64_t(j) * d;
                const double dot = gt_dot16(xs, y, d);   // (the canonical order of the exact stages)
                const double t = gt_pair_key(qn, dot, xn[j], metric);
                const double kv = affinity(gt_key_to_dist(t, dtype, metric), bwi, decay, idecay);
                keep = kv >= thresh;
                kvv = kv;
                if (pairs && keep) {
                    const double tt = gt_pair_key(xn[j], dot, qn, metric);
                    const double kvt = affinity(gt_key_to_dist(tt, dtype, metric), bw[j], decay, idecay);
                    if (kvt >= thresh) kvv = -merge_values(kv, kvt, GT_SYMM_ADD, 1.0);
                }
            }
            // (kept entries to the front, as above; the list is in no particular order)
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const uint32_t slot = uint32_t(kept) + uint32_t(__popcll(km & ((1ull << lane) - 1ull)));
                kp[slot] = kvv;
                if (slot != e) lp[slot] = lkey;
            }
            kept += __popcll(km);
            if (count_owners) {
                int o = (keep && (pairs < 2 || kvv >= 0.0)) ? owner_of(sp, j) : -1;
                if (pairs == 3 && o == self_rank) o = -1;
                for (int r = 0; r < sp.world; ++r) {
                    const int c = __popcll(__ballot(o == r));
                    if (lane == r) owner_cnt += c;
                }
            }
        }
    }
    if (lane == 0) lenN[i] = kept;
    if (lane == 0 && trow && lenN_s) lenN_s[trow[i]] = kept;   // (tables by sorted position: by slot too - every launch writes it)
    if constexpr (RADIUS) {
        if (lane == 0) const_cast<uint32_t*>(rcounts)[src] = uint32_t(kept);   // the list now ends behind its kept entries
    }
    if (count_owners && lane < sp.world) ownercnt[int64_t(lane) * nloc + i] = owner_cnt;   // owner-major layout
}

// ---- A1s: the affinity pass over tables that lie by sorted position (KnnWork::tab_sorted) -----------------------------
// Same arithmetic, same in-place result as affinity_kernel<T, false, 1> (the pair-resolving launch over the rows whose tables
// carry the transposed keys) - but a row there is a chain of dependent reads (row number -> its header -> its table -> the
// partners' bandwidths) that eight waves per SIMD cannot hide (SQ counters, round 5: waves parked 66 % of their cycles, VALU
// busy 14 %).  Here a wave walks `rpw` consecutive slots: everything a row needs before its table - row number, count, flags,
// bandwidth - lies BY SLOT and is fetched two rows ahead, the first 128 entries of the next row's table (addresses known from
// the slot alone) while the current row's partners' bandwidths are on their way: one exposed round trip per row instead of four.
struct SlotTab {
    double d2a, d2b, ta, tb;
    uint32_t ja, jb;
};
__global__ __launch_bounds__(256) void affinity_slots_kernel(
    const int64_t nslots, const int rpw, const SlotRec* __restrict__ rec_s, const BwPos* __restrict__ bwpos, const int dtype,
    const int metric, const int MP, const int limit, double* __restrict__ cand_d2, uint32_t* __restrict__ cand_j,
    const double* __restrict__ cand_d2t, const double decay, const int binary, const double thresh, const double rf_guard,
    const int count_owners, int32_t* __restrict__ lenN, int32_t* __restrict__ ownercnt, int32_t* __restrict__ tablen,
    uint32_t* __restrict__ posj, const int64_t* __restrict__ sC, int32_t* __restrict__ lenN_s, const Splits sp, const int self_rank) {
    // count_owners = 2 (a rank of a row-sharded build, GraphState::pairs_shard_bins: slots are the rank's rows, the partner records
    // lie by the row numbers of ALL rows, a partner on another rank has no position - kNoDest): per owner the one-sided entries
    // whose partner lives on ANOTHER rank (ownercnt owner-major [world][nslots], what emit_triplets_kernel sends)
    // posj / sC (optional, together): the destination lookup of the pair-resolved tail (bin_count_kernel) is done HERE - the
    // partner's sorted position arrives with its bandwidth, every kept entry's destination goes to posj[sC[slot] + its place in
    // the row] (sC: scan of the TABLE lengths, known before this pass; kNoDest for a settled pair and for the slots behind the
    // kept ones); posj_hist_kernel counts the bins from the packed array.  (Counting them here, in an LDS histogram per
    // workgroup, cost the pass what bin_count_kernel had taken; a separate gather of the positions +0.45 ms: C3, round 5.)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const int idecay = gt_whole_decay(decay);
    const int64_t t0 = (int64_t(blockIdx.x) * 4 + w) * rpw;
    const int64_t t1 = t0 + rpw < nslots ? t0 + rpw : nslots;
    struct Hdr {
        SlotRec r;
        int64_t c0;   // start of the row's stretch of posj
    };
    auto load_hdr = [&](const int64_t t, Hdr& h) {
        if (t < t1) {   // (uniform)
            h.r = rec_s[t];
            if (h.r.n > uint32_t(limit)) h.r.n = uint32_t(limit);
            if (h.r.i < 0) h.r.n = 0u;
            if (posj) h.c0 = sC[t];
        }
    };
    auto load_tab = [&](const int64_t t, const Hdr& h, SlotTab& x) {
        if (t < t1) {   // (the row's defined slots: its header arrived a row earlier)
            const size_t o = size_t(t) * MP + lane;
            if (uint32_t(lane) < h.r.n) {
                x.d2a = cand_d2[o];
                x.ja = cand_j[o];
                x.ta = cand_d2t[o];
            }
            if (uint32_t(lane) + 64u < h.r.n) {
                x.d2b = cand_d2[o + 64];
                x.jb = cand_j[o + 64];
                x.tb = cand_d2t[o + 64];
            }
        }
    };
    // three rows in flight: the table of row t + 2 and the header of row t + 3 are requested, then the partners' records of
    // row t + 1 (its columns arrived while row t - 1 was worked on), then row t - everything it needs is in registers - is done
    struct SlotBw {
        BwPos a, b;
    };
    auto load_bw = [&](const int64_t t, const Hdr& h, const SlotTab& x, SlotBw& g) {
        g.a.bw = g.b.bw = 1.0;
        g.a.pos = g.b.pos = 0u;
        if (t < t1) {
            if (uint32_t(lane) < h.r.n) g.a = bwpos[x.ja];
            if (uint32_t(lane) + 64u < h.r.n) g.b = bwpos[x.jb];
        }
    };
    Hdr h0 = {{0, 0u, 1.0}, 0}, h1 = h0, h2 = h0, h3 = h0;
    SlotTab x0 = {0.0, 0.0, 0.0, 0.0, 0u, 0u}, x1 = x0, x2 = x0;
    SlotBw g0 = {{1.0, 0u, 0u}, {1.0, 0u, 0u}}, g1 = g0;
    load_hdr(t0, h0);
    load_hdr(t0 + 1, h1);
    load_hdr(t0 + 2, h2);
    load_tab(t0, h0, x0);
    load_tab(t0 + 1, h1, x1);
    load_bw(t0, h0, x0, g0);
    for (int64_t t = t0; t < t1; ++t) {
        load_tab(t + 2, h2, x2);
        load_hdr(t + 3, h3);
        load_bw(t + 1, h1, x1, g1);
        const uint32_t n = h0.r.n;
        if (h0.r.i >= 0) {   // (else: the radius launch's / the listed launch's row)
            const int64_t i = h0.r.i;
            const double bwi = h0.r.bwi;
            int kept = 0;
            int owner_cnt = 0;   // (count_owners = 2: lane o counts what leaves for rank o)
            auto chunk = [&](const uint32_t e, const double d2, const uint32_t j, const double d2t, const double bwj,
                             const uint32_t pj) {
                bool keep = false;
                double kvv = -1.0;
                if (e < n) {
                    double kv = 1.0;
                    if (!binary) kv = affinity(gt_key_to_dist(d2, dtype, metric), bwi, decay, idecay);
                    keep = binary || (kv >= thresh);
                    kvv = kv;
                    if (keep) {   // (see affinity_kernel: the pair is settled here when row j keeps it too)
                        const double dt = gt_key_to_dist(d2t, dtype, metric);
                        if (!(dt > bwj * rf_guard)) {
                            const double kvt = affinity(dt, bwj, decay, idecay);
                            if (kvt >= thresh) kvv = -merge_values(kv, kvt, GT_SYMM_ADD, 1.0);
                        }
                    }
                }
                const unsigned long long km = __ballot(keep);
                if (keep) {
                    const uint32_t slot = uint32_t(kept) + uint32_t(__popcll(km & ((1ull << lane) - 1ull)));
                    cand_d2[size_t(t) * MP + slot] = kvv;
                    if (slot != e) cand_j[size_t(t) * MP + slot] = j;
                    if (posj) posj[h0.c0 + slot] = kvv >= 0.0 ? pj : kNoDest;
                }
                kept += __popcll(km);
                if (count_owners == 2) {
                    // (the few entries whose partner lives elsewhere: most chunks hold none)
                    const bool leaves = keep && kvv >= 0.0 && pj == kNoDest;
                    if (__ballot(leaves) != 0ull) {
                        const int o = leaves ? owner_of(sp, j) : -1;
                        for (int r = 0; r < sp.world; ++r) {
                            const int cnt_r = __popcll(__ballot(o == r && r != self_rank));
                            if (lane == r) owner_cnt += cnt_r;
                        }
                    }
                }
            };
            chunk(uint32_t(lane), x0.d2a, x0.ja, x0.ta, g0.a.bw, g0.a.pos);
            if (n > 64u) chunk(uint32_t(lane) + 64u, x0.d2b, x0.jb, x0.tb, g0.b.bw, g0.b.pos);   // (uniform)
            for (uint32_t e0 = 128u; e0 < n; e0 += 64u) {   // (the few rows beyond 128 entries: as they come)
                const uint32_t e = e0 + lane;
                double d2 = 0.0, d2t = 0.0;
                uint32_t j = 0;
                BwPos bp = {1.0, 0u, 0u};
                if (e < n) {
                    d2 = cand_d2[size_t(t) * MP + e];
                    j = cand_j[size_t(t) * MP + e];
                    d2t = cand_d2t[size_t(t) * MP + e];
                    bp = bwpos[j];
                }
                chunk(e, d2, j, d2t, bp.bw, bp.pos);
            }
            if (posj)   // (the row's stretch is as long as its table)
                for (uint32_t e = uint32_t(kept) + lane; e < n; e += 64u) posj[h0.c0 + e] = kNoDest;
            if (lane == 0) {
                tablen[i] = int32_t(kept);
                lenN[i] = kept;
                lenN_s[t] = kept;   // (by slot too: what the passes over the sorted rows read)
                if (count_owners == 1) ownercnt[i] = kept;   // (one rank: every kept entry is its own)
            }
            if (count_owners == 2 && lane < sp.world) ownercnt[int64_t(lane) * nslots + i] = owner_cnt;
        }
        h0 = h1;
        h1 = h2;
        h2 = h3;
        x0 = x1;
        x1 = x2;
        g0 = g1;
    }
}

// the records affinity_slots_kernel reads, for a rank of a row-sharded build (slot = local row): rec_s by local row, bwpos by
// the row numbers of ALL rows - a partner's bandwidth from the all-gather and its local row, kNoDest where another rank owns it
__global__ __launch_bounds__(256) void shard_slot_records_kernel(const int64_t nloc, const int64_t r0, const int64_t n_total,
                                                                 const double* __restrict__ bw_all, const uint32_t* __restrict__ cand_n,
                                                                 const uint8_t* __restrict__ keyt_ok, const int32_t* __restrict__ rowsrc,
                                                                 SlotRec* __restrict__ rec_s, BwPos* __restrict__ bwpos) {
    const int64_t j = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (j < n_total) {
        BwPos b;
        b.bw = bw_all[j];
        b.pos = (j >= r0 && j < r0 + nloc) ? uint32_t(j - r0) : kNoDest;
        b.pad = 0u;
        bwpos[j] = b;
    }
    if (j < nloc) {
        SlotRec r;
        r.i = (rowsrc[j] < 0 && keyt_ok[j] != 0) ? int32_t(j) : -1;   // (-1: the row of another launch - radius pass, repaired table)
        r.n = cand_n[j];
        r.bwi = bw_all[r0 + j];
        rec_s[j] = r;
    }
}

// destinations per bin from the packed posj (fused destination lookup): a workgroup counts a contiguous stretch in the LDS
__global__ __launch_bounds__(256) void posj_hist_kernel(const uint32_t* __restrict__ posj, const int64_t total, const int shift,
                                                        const int nbins, int32_t* __restrict__ bincnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int32_t* hist = reinterpret_cast<int32_t*>(smem_raw);
    for (int b = threadIdx.x; b < nbins; b += 256) hist[b] = 0;
    __syncthreads();
    const int64_t per = (((total + gridDim.x - 1) / gridDim.x) + 3) & ~int64_t(3);
    const int64_t e0 = int64_t(blockIdx.x) * per, e1 = e0 + per < total ? e0 + per : total;
    const uint4* p4 = reinterpret_cast<const uint4*>(posj + e0);   // (per is a multiple of 4: 16-byte aligned)
    const int64_t n4 = e1 > e0 ? (e1 - e0) / 4 : 0;
    for (int64_t q = threadIdx.x; q < n4; q += 256) {
        const uint4 v = p4[q];
        if (v.x != kNoDest) atomicAdd(&hist[v.x >> shift], 1);
        if (v.y != kNoDest) atomicAdd(&hist[v.y >> shift], 1);
        if (v.z != kNoDest) atomicAdd(&hist[v.z >> shift], 1);
        if (v.w != kNoDest) atomicAdd(&hist[v.w >> shift], 1);
    }
    for (int64_t e = e0 + n4 * 4 + threadIdx.x; e < e1; e += 256) {
        const uint32_t v = posj[e];
        if (v != kNoDest) atomicAdd(&hist[v >> shift], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (hist[b] != 0) atomicAdd(&bincnt[b], hist[b]);
}

// ---- A2t: transposed triplets, bucketed by destination rank --------------------------------------
__global__ __launch_bounds__(256) void emit_triplets_kernel(
    const int64_t nloc, const int64_t r0, const int MP, const double* __restrict__ cand_k,
    const uint32_t* __restrict__ cand_j, const int32_t* __restrict__ rowsrc, const uint64_t* __restrict__ rlists,
    const uint32_t* __restrict__ rcounts, const int32_t rcap, const double* __restrict__ rK, const Splits sp,
    const int64_t* __restrict__ ownerpos, const int32_t* __restrict__ tablen, Triplet* __restrict__ out, const int skip_owner) {
    // skip_owner (-1: none): entries for that rank - this one - do not travel (the pair-resolved tail of a sharded build takes its
    // local one-sided entries through the rank's own destination bins; the affinity pass counted accordingly)
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int32_t src = rowsrc[i];
    uint32_t n;
    const double* kv;
    if (src < 0) {
        n = uint32_t(tablen[i]);
        kv = cand_k + i * MP;
    } else {
        n = rcounts[src];
        if (n > uint32_t(rcap)) n = uint32_t(rcap);
        kv = rK + size_t(src) * rcap;
    }
    // lane r tracks how many triplets of this row already went to owner r; the row's first slot in bucket r is
    // ownerpos[r * nloc + i] (exclusive scan of the owner-major count array: deterministic, no atomics)
    int64_t my_base = (lane < sp.world) ? ownerpos[int64_t(lane) * nloc + i] : 0;
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {
        const uint32_t e = e0 + lane;
        double v = -1.0;
        uint32_t j = 0;
        if (e < n) {
            v = kv[e];
            j = (src < 0) ? cand_j[i * MP + e] : cand_index(rlists[size_t(src) * rcap + e]);
        }
        const bool keep = v >= 0.0;
        int o = keep ? owner_of(sp, j) : -1;
        if (o == skip_owner) o = -1;
        for (int r = 0; r < sp.world; ++r) {
            const unsigned long long m = __ballot(o == r);
            if (m == 0ull) continue;
            const int64_t base = __shfl(my_base, r);
            if (lane == r) my_base += __popcll(m);
            if (o == r) {
                const int64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
                Triplet t;
                t.row = j;
                t.col = uint32_t(r0 + i);
                t.val = v;
                out[pos] = t;
            }
        }
    }
}

// Single rank, cell-sorted row order available (gt_order.hip): the send buffer is laid out in THAT order - row perm[p]'s
// triplets follow row perm[p-1]'s - so that the passes over the received triplets touch the union rows of one
// neighbourhood while they are still in the L2 (a row's targets are its neighbours, i.e. rows of the same cells).
__global__ __launch_bounds__(256) void gather_counts_kernel(const int32_t* __restrict__ cnt, const int32_t* __restrict__ perm,
                                                            const int64_t n, int32_t* __restrict__ out) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p < n) out[p] = cnt[perm[p]];
}
__global__ __launch_bounds__(256) void scatter_positions_kernel(const int64_t* __restrict__ pos_sorted,
                                                                const int32_t* __restrict__ perm, const int64_t n,
                                                                int64_t* __restrict__ pos) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p < n) pos[perm[p]] = pos_sorted[p];
    if (p == n) pos[n] = pos_sorted[n];
}

// ---- R1: count received triplets per local row ---------------------------------------------------
__global__ __launch_bounds__(256) void count_recv_kernel(const Triplet* __restrict__ recv, const int64_t n_recv,
                                                         const int64_t r0, int32_t* __restrict__ lenT,
                                                         int32_t* __restrict__ slot) {
    // the value the counter held is this triplet's place among the row's received entries: the fill pass needs no
    // second round of atomics (the order within a row is arbitrary either way - the merge sorts by column)
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < n_recv; t += int64_t(gridDim.x) * 256)
        slot[t] = atomicAdd(&lenT[int64_t(recv[t].row) - r0], 1);
}

// ---- scans (int32 in -> int64 exclusive out, with total in out[n]) --------------------------------
__global__ __launch_bounds__(256) void scan_block_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                         const int64_t n, int64_t* __restrict__ out,
                                                         int64_t* __restrict__ block_sums) {
    __shared__ int64_t wsum[4];
    const int64_t base = int64_t(blockIdx.x) * 1024;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t v[4];
    int64_t tsum = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t idx = base + int64_t(threadIdx.x) * 4 + u;
        v[u] = idx < n ? int64_t(a[idx]) + (b ? int64_t(b[idx]) : 0) : 0;
        tsum += v[u];
    }
    // inclusive scan of tsum across the wave
    int64_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int64_t woff = 0;
    for (int r = 0; r < w; ++r) woff += wsum[r];
    int64_t run = woff + inc - tsum;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t idx = base + int64_t(threadIdx.x) * 4 + u;
        if (idx < n) out[idx] = run;
        run += v[u];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = woff + inc;
}

__global__ void scan_sums_kernel(int64_t* __restrict__ block_sums, const int64_t nb) {
    // one wave: exclusive scan of the block sums in chunks of 64 (a serial loop of a few thousand dependent loads took
    // 0.1 ms per scan, four scans per graph)
    const int lane = threadIdx.x;
    int64_t run = 0;
    for (int64_t i0 = 0; i0 < nb; i0 += 64) {
        const int64_t i = i0 + lane;
        const int64_t v = i < nb ? block_sums[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (i < nb) block_sums[i] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) block_sums[nb] = run;
}

__global__ __launch_bounds__(256) void scan_add_kernel(int64_t* __restrict__ out, const int64_t n,
                                                       const int64_t* __restrict__ block_sums, const int64_t nb) {
    const int64_t base = int64_t(blockIdx.x) * 1024;
    const int64_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t idx = base + int64_t(threadIdx.x) * 4 + u;
        if (idx < n) out[idx] += add;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_sums[nb];
}

// ---- A2n: N part of the union rows ---------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_rows_kernel(
    const int64_t nloc, const int MP, const double* __restrict__ cand_k, const uint32_t* __restrict__ cand_j,
    const int32_t* __restrict__ rowsrc, const uint64_t* __restrict__ rlists, const uint32_t* __restrict__ rcounts,
    const int32_t rcap, const double* __restrict__ rK, const int64_t* __restrict__ off,
    const int32_t* __restrict__ tablen, UEntry* __restrict__ U, const int32_t* __restrict__ relabel) {
    // relabel (optional): the context's rows are a renumbering of the caller's (gt_points_cell_sort) - the union rows get
    // the caller's column numbers, so that the sort by column, the row sums and the CSR are the caller's
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int32_t src = rowsrc[i];
    uint32_t n;
    const double* kv;
    if (src < 0) {
        n = uint32_t(tablen[i]);
        kv = cand_k + i * MP;
    } else {
        n = rcounts[src];
        if (n > uint32_t(rcap)) n = uint32_t(rcap);
        kv = rK + size_t(src) * rcap;
    }
    int64_t pos = off[i];
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {
        const uint32_t e = e0 + lane;
        double v = -1.0;
        uint32_t j = 0;
        if (e < n) {
            v = kv[e];
            j = (src < 0) ? cand_j[i * MP + e] : cand_index(rlists[size_t(src) * rcap + e]);
        }
        const bool keep = v >= 0.0;
        int total;
        const int p = wave_prefix_count(keep, lane, total);
        if (keep) U[pos + p] = UEntry{(relabel ? uint32_t(relabel[j]) : j) << 1, 0u, v};
        pos += total;
    }
}

// ---- R2: T part of the union rows ----------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_recv_kernel(const Triplet* __restrict__ recv, const int64_t n_recv,
                                                        const int64_t r0, const int64_t* __restrict__ off,
                                                        const int32_t* __restrict__ lenN,
                                                        const int32_t* __restrict__ slot,
                                                        UEntry* __restrict__ U, const int32_t* __restrict__ relabel) {
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < n_recv; t += int64_t(gridDim.x) * 256) {
        const Triplet tr = recv[t];
        const int64_t il = int64_t(tr.row) - r0;
        const int64_t pos = off[il] + lenN[il] + slot[t];
        U[pos] = UEntry{((relabel ? uint32_t(relabel[tr.col]) : tr.col) << 1) | 1u, 0u, tr.val};
    }
}

// ---- single rank, cell-sorted order available: the transpose goes through DESTINATION BINS -------------------------
// count_recv / fill_recv pay one device-scope atomic and one scattered 16-byte write per received triplet (58 M of each
// at N = 10^6: 2.9 + 2.7 ms, both fabric-bound - a device-scope atomic is executed at the memory side, the union rows of
// unrelated destinations share no cache line).  With every row of the graph on this device the exchange is a local
// permutation, and the cell-sorted order of the points (gt_order.hip) makes it a NEARLY local one: the targets of a row
// are its neighbours, i.e. rows of the same few cells.  So
//   * the union rows are laid out in sorted order (row at sorted position p = point perm[p]); positions are cut into bins
//     of 2^shift rows (512: the triplets of a bin are a few hundred KB, its union rows under a MB - L2-resident);
//   * bin_count_kernel:  triplets per destination bin (histogram in the LDS, one global atomic per workgroup and bin);
//   * bin_emit_kernel:   the triplets, written bin by bin (a workgroup reserves its run inside each bin with ONE returning
//                        atomic per bin it touches, places its triplets through LDS cursors);
//   * bin_fill_kernel:   one workgroup per bin counts its triplets per row, scans, and writes both halves of its union
//                        rows - every counter and cursor lives in the LDS, the writes land in the bin's own window;
//   * sort / merge / long rows run unchanged on the sorted row space; compact_kernel maps position -> row on the way out.
// Entry order inside a union row depends on the atomics' arrival order; the per-row sort keys (column, tag) are unique,
// so K does not.
template <typename F>
__device__ __forceinline__ void for_kept_entries(const int64_t i, const int64_t ti, const int lane, const int MP, const double* __restrict__ cand_k,
                                                 const uint32_t* __restrict__ cand_j, const int32_t* __restrict__ rowsrc,
                                                 const uint64_t* __restrict__ rlists, const uint32_t* __restrict__ rcounts,
                                                 const int32_t rcap, const double* __restrict__ rK,
                                                 const int32_t* __restrict__ tablen, F&& f) {
    const int32_t src = rowsrc[i];
    uint32_t n;
    const double* kv;
    if (src < 0) {
        n = uint32_t(tablen[i]);
        kv = cand_k + ti * MP;   // (ti: where the row's table is - the row, or its sorted position: KnnWork::tab_sorted)
    } else {
        n = rcounts[src];
        if (n > uint32_t(rcap)) n = uint32_t(rcap);
        kv = rK + size_t(src) * rcap;
    }
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {   // (wave-uniform trip count: f may use wave-wide operations)
        const uint32_t e = e0 + lane;
        double v = -1.0;
        uint32_t j = 0;
        if (e < n) {
            v = kv[e];
            j = (src < 0) ? cand_j[ti * MP + e] : cand_index(rlists[size_t(src) * rcap + e]);
        }
        // (a kept value is positive; pair-resolved builds store a settled mutual pair as minus its final value: such an
        //  entry stays in its row and nothing of it is sent - but it keeps its slot in the enumeration of the kept entries)
        f(e < n, e < n && v >= 0.0, j, v);
    }
}

constexpr int kEmitRows = 128;   // sorted rows per workgroup of bin_emit_kernel

__global__ __launch_bounds__(256) void bin_count_kernel(const int64_t nloc, const int MP, const double* __restrict__ cand_k,
                                                        const uint32_t* __restrict__ cand_j, const int32_t* __restrict__ rowsrc,
                                                        const uint64_t* __restrict__ rlists, const uint32_t* __restrict__ rcounts,
                                                        const int32_t rcap, const double* __restrict__ rK,
                                                        const int32_t* __restrict__ tablen, const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ pos, const int shift, const int nbins,
                                                        const int64_t* __restrict__ sN, uint32_t* __restrict__ posj,
                                                        int32_t* __restrict__ bincnt, const int tab_sorted) {
    // posj: the sorted position of every kept entry's column, in the order of the kept entries of the sorted rows (row p
    // at sN[p]) - the ONE pass that gathers pos[j] (a random 4-byte read per entry); bin_emit_kernel streams it
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int32_t* hist = reinterpret_cast<int32_t*>(smem_raw);
    for (int b = threadIdx.x; b < nbins; b += 256) hist[b] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t p = int64_t(blockIdx.x) * 4 + w; p < nloc; p += int64_t(gridDim.x) * 4) {
        int64_t at = sN[p];
        const int64_t i = perm[p];
        for_kept_entries(i, tab_sorted ? p : i, lane, MP, cand_k, cand_j, rowsrc, rlists, rcounts, rcap, rK, tablen,
                         [&](bool slot, bool send, uint32_t j, double) {
                             int total;
                             const int q = wave_prefix_count(slot, lane, total);
                             if (send) {
                                 const uint32_t pj = uint32_t(pos[j]);
                                 posj[at + q] = pj;
                                 atomicAdd(&hist[pj >> shift], 1);
                             } else if (slot) {
                                 posj[at + q] = kNoDest;   // (a settled pair: nothing travels)
                             }
                             at += total;
                         });
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (hist[b] != 0) atomicAdd(&bincnt[b], hist[b]);
}

__global__ __launch_bounds__(256) void bin_emit_kernel(const int64_t nloc, const int MP, const double* __restrict__ cand_k,
                                                       const uint32_t* __restrict__ cand_j, const int32_t* __restrict__ rowsrc,
                                                       const uint64_t* __restrict__ rlists, const uint32_t* __restrict__ rcounts,
                                                       const int32_t rcap, const double* __restrict__ rK,
                                                       const int32_t* __restrict__ tablen, const int32_t* __restrict__ perm,
                                                       const int64_t* __restrict__ sN, const uint32_t* __restrict__ posj,
                                                       const int shift, const int nbins,
                                                       const int64_t* __restrict__ binoff, int32_t* __restrict__ bincur,
                                                       Triplet* __restrict__ out, const int tab_sorted) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int32_t* hist = reinterpret_cast<int32_t*>(smem_raw);   // entries of this workgroup per bin, then its cursor there
    int32_t* base = hist + nbins;                           // first slot of this workgroup's run inside the bin
    for (int b = threadIdx.x; b < nbins; b += 256) hist[b] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t p0 = int64_t(blockIdx.x) * kEmitRows;
    const int64_t p1 = p0 + kEmitRows < nloc ? p0 + kEmitRows : nloc;
    // (the destinations of the workgroup's rows are one contiguous stretch of posj)
    for (int64_t e = sN[p0] + threadIdx.x; e < sN[p1]; e += 256) {
        const uint32_t pj = posj[e];
        if (pj != kNoDest) atomicAdd(&hist[pj >> shift], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (hist[b] != 0) {
            base[b] = atomicAdd(&bincur[b], hist[b]);
            hist[b] = 0;
        }
    __syncthreads();
    for (int64_t p = p0 + w; p < p1; p += 4) {
        const uint32_t i = uint32_t(perm[p]);
        int64_t at = sN[p];
        for_kept_entries(int64_t(i), tab_sorted ? p : int64_t(i), lane, MP, cand_k, cand_j, rowsrc, rlists, rcounts, rcap, rK, tablen,
                         [&](bool slot, bool send, uint32_t, double v) {
                             int total;
                             const int q = wave_prefix_count(slot, lane, total);
                             const int64_t mine = at + q;
                             at += total;
                             if (send) {
                                 const uint32_t pj = posj[mine];
                                 const int b = int(pj >> shift);
                                 const int slot = base[b] + atomicAdd(&hist[b], 1);
                                 Triplet t;
                                 t.row = pj;   // destination: the SORTED POSITION of point j
                                 t.col = i;    // column: the point itself
                                 t.val = v;
                                 out[binoff[b] + slot] = t;
                             }
                         });
    }
}

// bin_emit_kernel for builds whose affinity pass looked the destinations up (GraphState::pairs_fused: every row is a table row,
// tables / counts / posj by slot).  The same triplets; a row here is table -> LDS cursor -> store: the header (row, count, start
// of its stretch of posj) is fetched two rows ahead, values and destinations one row ahead, and the bin's offset - a dependent
// global read per entry there - is folded into the run's base when the run is reserved.
__global__ __launch_bounds__(256) void bin_emit_slots_kernel(const int64_t nloc, const int MP, const double* __restrict__ cand_k,
                                                             const int32_t* __restrict__ lenN_s, const int32_t* __restrict__ perm,
                                                             const int64_t* __restrict__ sC, const uint32_t* __restrict__ posj,
                                                             const int shift, const int nbins, const int64_t* __restrict__ binoff,
                                                             int32_t* __restrict__ bincur, Triplet* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int32_t* hist = reinterpret_cast<int32_t*>(smem_raw);          // entries of this workgroup per bin, then its cursor there
    uint32_t* base = reinterpret_cast<uint32_t*>(hist + nbins);    // first slot of this workgroup's run: binoff[b] + its start inside the bin
    for (int b = threadIdx.x; b < nbins; b += 256) hist[b] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const int64_t p0 = int64_t(blockIdx.x) * kEmitRows;
    const int64_t p1 = p0 + kEmitRows < nloc ? p0 + kEmitRows : nloc;
    for (int64_t e = sC[p0] + threadIdx.x; e < sC[p1]; e += 256) {
        const uint32_t pj = posj[e];
        if (pj != kNoDest) atomicAdd(&hist[pj >> shift], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (hist[b] != 0) {
            base[b] = uint32_t(binoff[b]) + uint32_t(atomicAdd(&bincur[b], hist[b]));
            hist[b] = 0;
        }
    __syncthreads();
    struct Hdr {
        int64_t c0;
        int32_t i, n;
    };
    struct Dat {
        double va, vb;
        uint32_t pa, pb;
    };
    auto load_hdr = [&](const int64_t p, Hdr& h) {
        if (p < p1) {
            h.c0 = sC[p];
            h.i = perm[p];
            h.n = lenN_s[p];
        }
    };
    auto load_dat = [&](const int64_t p, const Hdr& h, Dat& x) {
        x.va = x.vb = -1.0;
        x.pa = x.pb = kNoDest;
        if (p < p1) {
            if (lane < h.n) {
                x.va = cand_k[size_t(p) * MP + lane];
                x.pa = posj[h.c0 + lane];
            }
            if (lane + 64 < h.n) {
                x.vb = cand_k[size_t(p) * MP + 64 + lane];
                x.pb = posj[h.c0 + 64 + lane];
            }
        }
    };
    auto send = [&](const uint32_t i, const double v, const uint32_t pj) {
        if (v >= 0.0 && pj != kNoDest) {   // (a settled pair is stored negative: nothing of it travels)
            const int b = int(pj >> shift);
            const uint32_t slot = base[b] + uint32_t(atomicAdd(&hist[b], 1));
            Triplet t;
            t.row = pj;   // destination: the SORTED POSITION of point j
            t.col = i;    // column: the point itself
            t.val = v;
            out[slot] = t;
        }
    };
    Hdr h0 = {0, 0, 0}, h1 = h0, h2 = h0, h3 = h0;
    Dat x0 = {-1.0, -1.0, kNoDest, kNoDest}, x1 = x0, x2 = x0;
    const int64_t pw = p0 + w;
    load_hdr(pw, h0);
    load_hdr(pw + 4, h1);
    load_hdr(pw + 8, h2);
    load_dat(pw, h0, x0);
    load_dat(pw + 4, h1, x1);
    for (int64_t p = pw; p < p1; p += 4) {
        load_hdr(p + 12, h3);
        load_dat(p + 8, h2, x2);
        send(uint32_t(h0.i), x0.va, x0.pa);
        send(uint32_t(h0.i), x0.vb, x0.pb);
        for (int e = 128 + lane; e < h0.n; e += 64) send(uint32_t(h0.i), cand_k[size_t(p) * MP + e], posj[h0.c0 + e]);
        h0 = h1;
        h1 = h2;
        h2 = h3;
        x0 = x1;
        x1 = x2;
    }
}

constexpr int kBigRow = 512;    // longer rows go to sort_merge_long_kernel
constexpr uint32_t kFusedHugeRow = 2u;
constexpr int kPairHugeRow = 1024;   // pair-resolved tail: union rows beyond this go to the segmented sort (see merge_long_final_kernel)
// What pairs_len_kernel hands the final merge, written by bin_fill_kernel itself where it is given (fused-destination builds: the
// kernel knows every union row's length the moment it has placed the row - a separate pass over the rows cost 0.11 ms)
struct PairsOut {
    int32_t* outlen;       // [row] final length of the row (null: not wanted)
    int32_t* biglist;      // rows for merge_long_final_kernel from [0] upwards, rows beyond the register sorts from [nloc - 1] downwards
    uint32_t* bigcount;    // [0] long rows, [1] rows of 129 ... kBigRow entries, [2] flags, [4] huge rows, [6..7] their entries
    int32_t* midlist;      // rows of 129 ... kBigRow entries
};

// One workgroup per bin.  sN: exclusive scan of the own-entry counts in sorted order (lenNs), binoff: of the bins'
// triplet counts - the bin's union rows start at sN[first row] + binoff[bin].
template <int NT>   // threads per workgroup (the kernel waits on load -> LDS atomic -> store chains: more of them in flight)
__global__ __launch_bounds__(NT) void bin_fill_kernel(const int64_t nloc, const int MP, const double* __restrict__ cand_k,
                                                       const uint32_t* __restrict__ cand_j, const int32_t* __restrict__ rowsrc,
                                                       const uint64_t* __restrict__ rlists, const uint32_t* __restrict__ rcounts,
                                                       const int32_t rcap, const double* __restrict__ rK,
                                                       const int32_t* __restrict__ tablen, const int32_t* __restrict__ perm,
                                                       const int shift, const int nbins, const int64_t* __restrict__ binoff,
                                                       const Triplet* __restrict__ trip, const int32_t* __restrict__ lenNs,
                                                       const int64_t* __restrict__ sN, int64_t* __restrict__ off,
                                                       UEntry* __restrict__ U, uint32_t* __restrict__ ucol,
                                                       double* __restrict__ uval, const PairsOut po) {
    // ucol / uval given (fused tail): the received entries go there as (column, value) in separate arrays instead of U
    __shared__ uint32_t mid_n, mid_base;
    __shared__ int32_t mid_rows[1024];   // (R <= 4096 rows per bin in principle; lists beyond 1024 medium rows of one bin go out one by one)
    if (po.outlen && threadIdx.x == 0) mid_n = 0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int R = 1 << shift;
    int32_t* cnt = reinterpret_cast<int32_t*>(smem_raw);   // [R] received entries per row, then the cursor of its T part
    int32_t* excl = cnt + R;                               // [R] start of the row inside the bin's window
    __shared__ int32_t wsum[NT / 64];
    const int b = blockIdx.x;
    const int64_t p0 = int64_t(b) << shift;
    const int nr = int(nloc - p0 < R ? nloc - p0 : R);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = threadIdx.x; r < R; r += NT) cnt[r] = 0;
    __syncthreads();
    const int64_t t0 = binoff[b], t1 = binoff[b + 1];
    {
        // (eight, then four loads in flight per thread: the loop is bound by the latency of load -> LDS atomic otherwise)
        int64_t t = t0 + threadIdx.x;
        for (; t + 15 * NT < t1; t += 16 * NT) {
            uint32_t r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = trip[t + u * NT].row;
#pragma unroll
            for (int u = 0; u < 16; ++u) atomicAdd(&cnt[int64_t(r[u]) - p0], 1);
        }
        for (; t + 7 * NT < t1; t += 8 * NT) {
            uint32_t r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = trip[t + u * NT].row;
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&cnt[int64_t(r[u]) - p0], 1);
        }
        for (; t + 3 * NT < t1; t += 4 * NT) {
            const uint32_t r0 = trip[t].row, r1 = trip[t + NT].row, r2 = trip[t + 2 * NT].row, r3 = trip[t + 3 * NT].row;
            atomicAdd(&cnt[int64_t(r0) - p0], 1);
            atomicAdd(&cnt[int64_t(r1) - p0], 1);
            atomicAdd(&cnt[int64_t(r2) - p0], 1);
            atomicAdd(&cnt[int64_t(r3) - p0], 1);
        }
        for (; t < t1; t += NT) atomicAdd(&cnt[int64_t(trip[t].row) - p0], 1);
    }
    __syncthreads();
    // exclusive scan of lenNs + cnt over the bin's rows: thread x owns the `per` consecutive rows from x per (R >= 256 is a
    // power of two: with more threads than rows the first R threads own one row each)
    const int per = R >= NT ? R / NT : 1;
    const int r_first = threadIdx.x * per;
    int32_t mine = 0;
    for (int u = 0; u < per; ++u) {
        const int r = r_first + u;
        if (r < nr) mine += lenNs[p0 + r] + cnt[r];
    }
    int32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int32_t run = inc - mine;
    for (int q = 0; q < w; ++q) run += wsum[q];
    // off[]: start of the whole union row (own + received entries: where its merged form goes); the received halves
    // alone are packed in U, row p's at off[p] - sN[p]
    const int64_t ubase = sN[p0] + t0;
    int32_t lnrun = 0;   // own entries of the bin's rows before r_first
    {
        int32_t lmine = 0;
        for (int u = 0; u < per; ++u) {
            const int r = r_first + u;
            if (r < nr) lmine += lenNs[p0 + r];
        }
        int32_t linc = lmine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t up = __shfl_up(linc, o);
            if (lane >= o) linc += up;
        }
        __syncthreads();   // (wsum is reused)
        if (lane == 63) wsum[w] = linc;
        __syncthreads();
        lnrun = linc - lmine;
        for (int q = 0; q < w; ++q) lnrun += wsum[q];
    }
    for (int u = 0; u < per; ++u) {
        const int r = r_first + u;
        if (r < nr) {
            const int32_t ln = lenNs[p0 + r], lt = cnt[r];
            excl[r] = run;
            cnt[r] = run - lnrun;   // cursor of the row's received half inside the bin's stretch of U
            off[p0 + r] = ubase + run;
            run += ln + lt;
            lnrun += ln;
            if (po.outlen) {   // (what pairs_len_kernel did in a pass of its own)
                const int32_t L = ln + lt;
                const int32_t row = perm[p0 + r];
                po.outlen[row] = L;
                if (L > kPairHugeRow) {
                    atomicOr(po.bigcount + 2, kFusedHugeRow);
                    po.biglist[nloc - 1 - int64_t(atomicAdd(po.bigcount + 4, 1u))] = row;
                    atomicAdd(reinterpret_cast<unsigned long long*>(po.bigcount + 6), (unsigned long long)L);
                } else if (L > kBigRow) {
                    po.biglist[atomicAdd(po.bigcount, 1u)] = row;
                } else if (L > 128) {
                    const uint32_t q = atomicAdd(&mid_n, 1u);
                    if (q < 1024u) mid_rows[q] = row;
                    else po.midlist[atomicAdd(po.bigcount + 1, 1u)] = row;
                }
            }
        }
    }
    if (po.outlen) {   // the bin's medium rows: one atomic for all of them
        __syncthreads();
        const uint32_t nm = mid_n < 1024u ? mid_n : 1024u;
        if (threadIdx.x == 0) mid_base = nm ? atomicAdd(po.bigcount + 1, nm) : 0u;
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < nm; q += NT) po.midlist[mid_base + q] = mid_rows[q];
    }
    if (b == nbins - 1 && threadIdx.x == NT - 1) off[nloc] = ubase + run;
    __syncthreads();
    {
        int64_t t = t0 + threadIdx.x;
        if (ucol) {
            for (; t + 15 * NT < t1; t += 16 * NT) {
                Triplet a[16];
                int sl[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) a[u] = trip[t + u * NT];
#pragma unroll
                for (int u = 0; u < 16; ++u) sl[u] = atomicAdd(&cnt[int64_t(a[u].row) - p0], 1);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    ucol[t0 + sl[u]] = a[u].col;
                    uval[t0 + sl[u]] = a[u].val;
                }
            }
            for (; t + 7 * NT < t1; t += 8 * NT) {
                Triplet a[8];
                int sl[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = trip[t + u * NT];
#pragma unroll
                for (int u = 0; u < 8; ++u) sl[u] = atomicAdd(&cnt[int64_t(a[u].row) - p0], 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ucol[t0 + sl[u]] = a[u].col;
                    uval[t0 + sl[u]] = a[u].val;
                }
            }
        }
        for (; t + 3 * NT < t1; t += 4 * NT) {
            const Triplet a0 = trip[t], a1 = trip[t + NT], a2 = trip[t + 2 * NT], a3 = trip[t + 3 * NT];
            const int s0 = atomicAdd(&cnt[int64_t(a0.row) - p0], 1);
            const int s1 = atomicAdd(&cnt[int64_t(a1.row) - p0], 1);
            const int s2 = atomicAdd(&cnt[int64_t(a2.row) - p0], 1);
            const int s3 = atomicAdd(&cnt[int64_t(a3.row) - p0], 1);
            if (ucol) {
                ucol[t0 + s0] = a0.col; uval[t0 + s0] = a0.val;
                ucol[t0 + s1] = a1.col; uval[t0 + s1] = a1.val;
                ucol[t0 + s2] = a2.col; uval[t0 + s2] = a2.val;
                ucol[t0 + s3] = a3.col; uval[t0 + s3] = a3.val;
            } else {
                U[t0 + s0] = UEntry{(a0.col << 1) | 1u, 0u, a0.val};
                U[t0 + s1] = UEntry{(a1.col << 1) | 1u, 0u, a1.val};
                U[t0 + s2] = UEntry{(a2.col << 1) | 1u, 0u, a2.val};
                U[t0 + s3] = UEntry{(a3.col << 1) | 1u, 0u, a3.val};
            }
        }
        for (; t < t1; t += NT) {
            const Triplet tr = trip[t];
            const int slot = atomicAdd(&cnt[int64_t(tr.row) - p0], 1);
            if (ucol) {
                ucol[t0 + slot] = tr.col;
                uval[t0 + slot] = tr.val;
            } else {
                U[t0 + slot] = UEntry{(tr.col << 1) | 1u, 0u, tr.val};
            }
        }
    }
}

__global__ __launch_bounds__(256) void scatter_i32_kernel(const int32_t* __restrict__ in, const int32_t* __restrict__ perm,
                                                          const int64_t n, int32_t* __restrict__ out) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p < n) out[perm[p]] = in[p];
}

// ---- S5: per-row sort by column + merge of (K0, K0^T) pairs ---------------------------------------

// merge a sorted key/value sequence held in registers (position p = t*64 + lane); returns the number of
// merged entries written to (Vk, Vv).  Keys: (column << 1) | tag, kNoKey where there is no entry (columns are below
// 2^31 - 1, so no key looks like it).
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
template <int NT>
__device__ __forceinline__ int merge_sorted_regs(const uint32_t (&hi)[NT], const uint64_t (&lo)[NT], const int lane,
                                                 const int symm, const double theta, uint32_t* __restrict__ Vk,
                                                 double* __restrict__ Vv) {
    int count = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t key = hi[t];
        const uint64_t val = lo[t];
        uint32_t pk = __shfl_up(key, 1);
        const uint32_t pk_edge = (t > 0) ? __shfl(hi[t > 0 ? t - 1 : 0], 63) : kNoKey;
        if (lane == 0) pk = pk_edge;
        uint32_t nk = __shfl_down(key, 1);
        uint64_t nv = __shfl_down((unsigned long long)val, 1);
        const uint32_t nk_edge = (t < NT - 1) ? __shfl(hi[t < NT - 1 ? t + 1 : t], 0) : kNoKey;
        const uint64_t nv_edge = (t < NT - 1) ? __shfl((unsigned long long)lo[t < NT - 1 ? t + 1 : t], 0) : 0ull;
        if (lane == 63) {
            nk = nk_edge;
            nv = nv_edge;
        }
        const bool valid = key != kNoKey;
        const uint32_t col = key >> 1;
        const bool first = valid && (pk == kNoKey || (pk >> 1) != col);
        const bool pair = first && nk != kNoKey && (nk >> 1) == col;
        const int tag = int(key & 1u);
        const double v = __longlong_as_double((long long)val);
        const double a = tag == 0 ? v : 0.0;
        const double b = tag == 1 ? v : (pair ? __longlong_as_double((long long)nv) : 0.0);
        const double m = merge_values(a, b, symm, theta);
        const bool emit = first && m != 0.0;
        int total;
        const int p = wave_prefix_count(emit, lane, total);
        if (emit) {
            Vk[count + p] = col;
            Vv[count + p] = m;
        }
        count += total;
    }
    return count;
}

// One wave sorts a union row of L <= 64 NT entries by (column, tag) and merges the pairs.  Sort keys are packed into
// one word - (column, tag) above the entry's position in the row - so the bitonic network moves one register per entry
// instead of a key/value pair; the values are fetched by position afterwards (the row was read a moment ago).
// K = uint32_t where key and position fit 32 bits together (sort_key_fits_u32: up to 4 M columns for rows of up to 512
// entries), else uint64_t.
template <int NT>
struct SortPos {
    static constexpr int bits = NT == 1 ? 6 : NT == 2 ? 7 : NT == 4 ? 8 : NT == 8 ? 9 : NT == 16 ? 10 : 11;
    static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8 || NT == 16 || NT == 32, "rows of 64 ... 2048 entries");
};
// largest key of the build (2 ncols - 1) must stay below the all-ones pattern of its field
static inline bool sort_key_fits_u32(int64_t ncols, int nt) {
    int bits = 6;
    while ((64 << (bits - 6)) < 64 * nt) ++bits;
    return 2 * ncols - 1 < (int64_t(1) << (32 - bits)) - 1;
}
// Where the entries of a union row come from.  Exchange path: all L of them from U.  Bin path: the first ln are the row's
// own kept entries, read where the affinity pass left them (the kept prefix of its table row or radius list), the
// received ones behind them from U - the own half of a union row is never copied.
struct RowSrc {
    const UEntry* U;
    int ln;
    const double* kv;
    const uint32_t* cj;
    const uint64_t* rl;
    __device__ __forceinline__ uint32_t key(const int p) const {
        return p < ln ? ((cj ? cj[p] : cand_index(rl[p])) << 1) : U[p - ln].key;
    }
    __device__ __forceinline__ double val(const int p) const { return p < ln ? kv[p] : U[p - ln].val; }
};
struct UnionSrc {
    const UEntry* U;
    const int64_t* sN;        // nullptr: exchange path (U holds whole rows at off[])
    const int32_t* lenNs;
    const int32_t* perm;
    const int32_t* rowsrc;
    const double* cand_k;
    const uint32_t* cand_j;
    int MP;
    const uint64_t* rlists;
    const double* rK;
    int32_t rcap;
};
__device__ __forceinline__ RowSrc make_row_src(const UnionSrc& us, const int64_t row, const int64_t o0) {
    RowSrc r;
    if (!us.sN) {
        r.U = us.U + o0;
        r.ln = 0;
        r.kv = nullptr;
        r.cj = nullptr;
        r.rl = nullptr;
        return r;
    }
    const int64_t i = us.perm[row];
    r.ln = us.lenNs[row];
    r.U = us.U + (o0 - us.sN[row]);
    const int32_t src = us.rowsrc[i];
    if (src < 0) {
        r.kv = us.cand_k + i * us.MP;
        r.cj = us.cand_j + i * us.MP;
        r.rl = nullptr;
    } else {
        r.kv = us.rK + size_t(src) * us.rcap;
        r.cj = nullptr;
        r.rl = us.rlists + size_t(src) * us.rcap;
    }
    return r;
}

template <typename K, int NT>
__device__ __forceinline__ int sort_merge_row(const RowSrc& U, const int L, const int lane, const int symm,
                                              const double theta,
                                              uint32_t* __restrict__ Vk, double* __restrict__ Vv) {
    constexpr int PB = SortPos<NT>::bits;
    K pk[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = t * 64 + lane;
        // descending sort of the complement = ascending sort of the key; 0 (no entry) sorts last
        pk[t] = (p < L) ? K(~((K(U.key(p)) << PB) | K(p))) : K(0);
    }
    wave_bitonic_desc<NT, K>(pk, lane);
    uint32_t hi[NT];
    uint64_t lo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        hi[t] = kNoKey;
        lo[t] = 0ull;
        if (pk[t] != K(0)) {
            const K x = K(~pk[t]);
            hi[t] = uint32_t(x >> PB);
            lo[t] = (uint64_t)__double_as_longlong(U.val(int(uint32_t(x) & ((1u << PB) - 1u))));
        }
    }
    return merge_sorted_regs<NT>(hi, lo, lane, symm, theta, Vk, Vv);
}

// (kBigRow: defined in front of bin_fill_kernel)
constexpr int kHugeRow = 2048;  // and beyond that to the global-memory sort (big_sort_kernel)

__global__ __launch_bounds__(256) void sort_merge_kernel(const int64_t nloc, const int64_t* __restrict__ off,
                                                         const UnionSrc us,
                                                         const int symm, const double theta, uint32_t* __restrict__ Vkey,
                                                         double* __restrict__ Vval, int32_t* __restrict__ outlen,
                                                         int32_t* __restrict__ bigrows, uint32_t* __restrict__ bigcount,
                                                         const int key32) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * (blockDim.x >> 6) + w;
    if (i >= nloc) return;
    const int64_t o0 = off[i];
    const int64_t L64 = off[i + 1] - o0;
    if (L64 > kBigRow) {
        if (lane == 0) {
            const uint32_t slot = atomicAdd(bigcount, 1u);
            bigrows[slot] = int32_t(i);
        }
        return;
    }
    const int L = int(L64);
    const RowSrc U = make_row_src(us, i, o0);
    uint32_t* Vk = Vkey + o0;
    double* Vv = Vval + o0;
    int c;
    if (key32) {   // (uniform over the launch)
        if (L <= 64)
            c = sort_merge_row<uint32_t, 1>(U, L, lane, symm, theta, Vk, Vv);
        else if (L <= 128)
            c = sort_merge_row<uint32_t, 2>(U, L, lane, symm, theta, Vk, Vv);
        else if (L <= 256)
            c = sort_merge_row<uint32_t, 4>(U, L, lane, symm, theta, Vk, Vv);
        else
            c = sort_merge_row<uint32_t, 8>(U, L, lane, symm, theta, Vk, Vv);
    } else {
        if (L <= 64)
            c = sort_merge_row<uint64_t, 1>(U, L, lane, symm, theta, Vk, Vv);
        else if (L <= 128)
            c = sort_merge_row<uint64_t, 2>(U, L, lane, symm, theta, Vk, Vv);
        else if (L <= 256)
            c = sort_merge_row<uint64_t, 4>(U, L, lane, symm, theta, Vk, Vv);
        else
            c = sort_merge_row<uint64_t, 8>(U, L, lane, symm, theta, Vk, Vv);
    }
    if (lane == 0) outlen[i] = c;
}

// Rows of kBigRow < L <= kHugeRow entries (hub rows of the transpose): same register sort with 16 / 32 keys per lane in
// a kernel of its own, so that its register budget does not cut the occupancy of the common case.  Persistent waves
// walk the list sort_merge_kernel left in bigrows; what is longer still is compacted to the front of hugerows.
// (NT = 16: the rows of up to 1024 entries - and the listing of the huge ones -, NT = 32: those of 1025 ... 2048: two launches,
//  so that the common range is not compiled against the 32-key network's 260 registers and its 272 B of scratch)
template <int NT>
__global__ __launch_bounds__(64) void sort_merge_long_kernel(const int64_t* __restrict__ off, const UnionSrc us,
                                                             const int symm,
                                                             const double theta, uint32_t* __restrict__ Vkey,
                                                             double* __restrict__ Vval, int32_t* __restrict__ outlen,
                                                             const int32_t* __restrict__ bigrows,
                                                             const uint32_t* __restrict__ bigcount,
                                                             int32_t* __restrict__ hugerows, uint32_t* __restrict__ hugecount) {
    const int lane = threadIdx.x;
    const uint32_t nbig = *bigcount;
    for (uint32_t b = blockIdx.x; b < nbig; b += gridDim.x) {
        const int64_t i = bigrows[b];
        const int64_t o0 = off[i];
        const int64_t L64 = off[i + 1] - o0;
        if (L64 > kHugeRow) {
            if (NT == 16 && lane == 0) hugerows[atomicAdd(hugecount, 1u)] = int32_t(i);
            continue;
        }
        const int L = int(L64);
        if ((L <= 1024) != (NT == 16)) continue;   // (the other launch's row; wave-uniform)
        const RowSrc U = make_row_src(us, i, o0);
        const int c = sort_merge_row<uint64_t, NT>(U, L, lane, symm, theta, Vkey + o0, Vval + o0);
        if (lane == 0) outlen[i] = c;
    }
}

// huge rows: bitonic sort in global scratch (one workgroup per row), then a separate merge kernel
__global__ __launch_bounds__(1024) void big_sort_kernel(const int32_t* __restrict__ bigrows, const int64_t* __restrict__ off,
                                                        const UnionSrc us,
                                                        const int64_t* __restrict__ scratch_off,
                                                        uint32_t* __restrict__ Sk, double* __restrict__ Sv) {
    const int64_t i = bigrows[blockIdx.x];
    const int64_t o0 = off[i];
    const int64_t L = off[i + 1] - o0;
    const int64_t s0 = scratch_off[blockIdx.x];
    const int64_t P = scratch_off[blockIdx.x + 1] - s0;   // power of two >= L
    uint32_t* k = Sk + s0;
    double* v = Sv + s0;
    const RowSrc U = make_row_src(us, i, o0);
    for (int64_t p = threadIdx.x; p < P; p += 1024) {
        k[p] = p < L ? U.key(int(p)) : 0xFFFFFFFFu;
        v[p] = p < L ? U.val(int(p)) : 0.0;
    }
    __syncthreads();
    for (int64_t kk = 2; kk <= P; kk <<= 1) {
        for (int64_t j = kk >> 1; j > 0; j >>= 1) {
            for (int64_t p = threadIdx.x; p < P; p += 1024) {
                const int64_t q = p ^ j;
                if (q > p) {
                    const bool asc = ((p & kk) == 0);
                    const uint32_t a = k[p], b = k[q];
                    if ((a > b) == asc) {
                        k[p] = b;
                        k[q] = a;
                        const double t = v[p];
                        v[p] = v[q];
                        v[q] = t;
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(64) void big_merge_kernel(const int32_t* __restrict__ bigrows, const int64_t* __restrict__ off,
                                                       const int64_t* __restrict__ scratch_off,
                                                       const uint32_t* __restrict__ Sk, const double* __restrict__ Sv,
                                                       const int symm, const double theta, uint32_t* __restrict__ Vkey,
                                                       double* __restrict__ Vval, int32_t* __restrict__ outlen) {
    const int lane = threadIdx.x;
    const int64_t i = bigrows[blockIdx.x];
    const int64_t o0 = off[i];
    const int64_t L = off[i + 1] - o0;
    const uint32_t* k = Sk + scratch_off[blockIdx.x];
    const double* v = Sv + scratch_off[blockIdx.x];
    int count = 0;
    for (int64_t p0 = 0; p0 < L; p0 += 64) {
        const int64_t p = p0 + lane;
        bool emit = false;
        uint32_t col = 0;
        double m = 0.0;
        if (p < L) {
            const uint32_t key = k[p];
            col = key >> 1;
            const bool first = (p == 0) || ((k[p - 1] >> 1) != col);
            const bool pair = first && (p + 1 < L) && ((k[p + 1] >> 1) == col);
            const int tag = int(key & 1u);
            const double a = tag == 0 ? v[p] : 0.0;
            const double b = tag == 1 ? v[p] : (pair ? v[p + 1] : 0.0);
            m = merge_values(a, b, symm, theta);
            emit = first && m != 0.0;
        }
        int total;
        const int pp = wave_prefix_count(emit, lane, total);
        if (emit) {
            Vkey[o0 + count + pp] = col;
            Vval[o0 + count + pp] = m;
        }
        count += total;
    }
    if (lane == 0) outlen[i] = count;
}

// ---- S6: compaction to CSR, degrees, anisotropy, P -----------------------------------------------
__global__ __launch_bounds__(256) void compact_kernel(const int64_t nloc, const int64_t r0, const int64_t* __restrict__ off,
                                                      const int32_t* __restrict__ outlen, const int64_t* __restrict__ indptr,
                                                      const uint32_t* __restrict__ Vkey, const double* __restrict__ Vval,
                                                      int32_t* __restrict__ indices, double* __restrict__ Kdata,
                                                      double* __restrict__ degree, uint32_t* __restrict__ flags,
                                                      const int32_t* __restrict__ perm, double* __restrict__ Pdata,
                                                      const int32_t* __restrict__ relabel = nullptr) {
    // perm: the merged rows are in sorted order (bin path: merged row p is row perm[p] of K); nullptr: in row order.
    // Pdata: the row-normalised operator is written along (no anisotropy: K is final here) - normalize_kernel's
    // arithmetic, same summation order
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t p = int64_t(blockIdx.x) * (blockDim.x >> 6) + w;
    if (p >= nloc) return;
    const int64_t i = perm ? int64_t(perm[p]) : p;
    const int64_t s = off[p];
    const int64_t dst = indptr[i];
    const int n = outlen[p];
    double sum = 0.0, asum = 0.0;
    bool has_diag = false;
    const int64_t self = relabel ? int64_t(relabel[r0 + i]) : r0 + i;   // the row's own column (the caller's numbering)
    for (int e = lane; e < n; e += 64) {
        const uint32_t c = Vkey[s + e];
        const double v = Vval[s + e];
        indices[dst + e] = int32_t(c);
        Kdata[dst + e] = v;
        sum += v;
        asum += fabs(v);
        has_diag |= (int64_t(c) == self) && (v != 0.0);
    }
    sum = wave_sum_f64(sum);
    const bool any_diag = __ballot(has_diag) != 0ull;
    if (lane == 0) {
        degree[i] = sum;
        if (!any_diag) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
    }
    if (Pdata) {
        asum = wave_sum_f64(asum);
        for (int e = lane; e < n; e += 64) {
            const double v = Vval[s + e];
            Pdata[dst + e] = (asum != 0.0) ? v / asum : v;
        }
    }
}

// =====================================================================================================================
// Final-CSR merge kernels of the pair-resolved tail (below; symmetrisation '+', single rank): K, P and the degrees are written
// ONCE, straight into the final CSR - the row pointers are scanned before the merge.
//   merge_final_kernel   one wave per row, in ROW order (the CSR is written front to back): register sort of the union by
//                        (column, tag), merge, indices and K written at the row's place, the row sum accumulated in
//                        compact_kernel's order (entry e in lane e % 64, increasing e, then the xor tree) through one LDS
//                        permute per 64 sorted entries, P = K / sum written behind it
//   merge_long_final_kernel  the rows of 513 ... 2048 entries (persistent waves, 16 / 32 keys per lane); longer rows go to
//                        the segmented sort (symm_huge)
// Results are bit-identical to sort_merge_kernel + compact_kernel (tests/test_gpu_symm_bins.py).  (A count-first variant for
// kernels WITHOUT settled pairs - a counting pass over every column in front of the same merge - was measured at 6.7 ms
// against 5.8 ms for sort + compact at N = 1e6 and removed in round 5; so was a fixed-slot layout for the received halves:
// the in-degrees of the unsymmetrised kernel are heavy-tailed - N = 1e6 mix: median 40, mean 72, 99.9 % 749, maximum 1416.)
struct FusedSrc {
    const int32_t* pos;       // row -> sorted position
    const int32_t* lenN;      // own kept entries per row
    const int64_t* off;       // [n + 1] start of the union row of every sorted position (own + received entries)
    const int64_t* sN;        // [n + 1] scan of the own entries in sorted order: the received half of position p starts at
                              //         off[p] - sN[p] in ucol / uval
    const int32_t* rowsrc;
    const double* cand_k;
    const uint32_t* cand_j;
    int MP;
    const uint64_t* rlists;
    const double* rK;
    int32_t rcap;
    const uint32_t* ucol;     // received entries, packed row by row in sorted order: columns ...
    const double* uval;       // ... and values
    int tab_sorted;           // the tables lie by sorted position (KnnWork::tab_sorted), else by row
    // a rank of a row-sharded build on renumbered points (graph_finish_pairs_shard): the tables hold the context's row numbers,
    // the CSR gets the caller's - cmap: context row -> caller's row (nullptr: the same), row0: context row of local row 0
    const int32_t* cmap;
    int64_t row0;
    __device__ __forceinline__ int64_t caller_row(const int64_t i) const { return cmap ? int64_t(cmap[row0 + i]) : i; }
};
struct RowSrc3 {
    int ln;
    const double* kv;
    const uint32_t* cj;
    const uint64_t* rl;
    const uint32_t* uc;
    const double* uv;
    const int32_t* cmap;   // (FusedSrc::cmap; the received entries carry the caller's columns already)
    __device__ __forceinline__ uint32_t key(const int p) const {
        if (p >= ln) return (uc[p - ln] << 1) | 1u;
        const uint32_t c = cj ? cj[p] : cand_index(rl[p]);
        return (cmap ? uint32_t(cmap[c]) : c) << 1;
    }
    __device__ __forceinline__ double val(const int p) const { return p < ln ? kv[p] : uv[p - ln]; }
};
__device__ __forceinline__ RowSrc3 make_row_src3(const FusedSrc& fs, const int64_t i, const int64_t p, int& lt) {
    RowSrc3 r;
    r.ln = fs.lenN[i];
    r.cmap = fs.cmap;
    const int64_t o0 = fs.off[p];
    lt = int(fs.off[p + 1] - o0) - r.ln;
    const int32_t src = fs.rowsrc[i];
    if (src < 0) {
        const int64_t ti = fs.tab_sorted ? p : i;
        r.kv = fs.cand_k + ti * fs.MP;
        r.cj = fs.cand_j + ti * fs.MP;
        r.rl = nullptr;
    } else {
        r.kv = fs.rK + size_t(src) * fs.rcap;
        r.cj = nullptr;
        r.rl = fs.rlists + size_t(src) * fs.rcap;
    }
    r.uc = fs.ucol + (o0 - fs.sN[p]);
    r.uv = fs.uval + (o0 - fs.sN[p]);
    return r;
}

// (kFusedHugeRow, kPairHugeRow: defined in front of bin_fill_kernel)
// pair-resolved tail: union rows beyond this go to the segmented sort (symm_huge).  1024, not kHugeRow: the 32-keys-per-lane
// register network that served 1025 ... 2048 entries needs 260 VGPRs and 272 B of scratch - ONE such row took ~1 ms (a hundred
// thousand unrolled instructions at one wave per SIMD), longer than the merge of the million short rows it ran beside.

// merge of a sorted (key, value) sequence held in registers straight into the final CSR row at dst; returns the row sum
// in compact_kernel's order.  Keys: (column << 1) | tag, kNoKey where there is no entry.
template <int NT>
__device__ __forceinline__ double merge_sorted_final(const uint32_t (&hi)[NT], const uint64_t (&lo)[NT], const int lane,
                                                     const int64_t row, int32_t* __restrict__ indices,
                                                     double* __restrict__ Kdata, double* __restrict__ Pdata,
                                                     const int64_t dst, bool& any_diag) {
    int count = 0;
    double msave[NT];
    int esave[NT];   // final index of the lane's entry of chunk t, -1: none
    double lsum = 0.0;   // this lane's partial sum: the entries e with e % 64 == lane, in increasing e
    bool has_diag = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t key = hi[t];
        const uint64_t val = lo[t];
        uint32_t pk = __shfl_up(key, 1);
        const uint32_t pk_edge = (t > 0) ? __shfl(hi[t > 0 ? t - 1 : 0], 63) : kNoKey;
        if (lane == 0) pk = pk_edge;
        uint32_t nk = __shfl_down(key, 1);
        uint64_t nv = __shfl_down((unsigned long long)val, 1);
        const uint32_t nk_edge = (t < NT - 1) ? __shfl(hi[t < NT - 1 ? t + 1 : t], 0) : kNoKey;
        const uint64_t nv_edge = (t < NT - 1) ? __shfl((unsigned long long)lo[t < NT - 1 ? t + 1 : t], 0) : 0ull;
        if (lane == 63) {
            nk = nk_edge;
            nv = nv_edge;
        }
        const bool valid = key != kNoKey;
        const uint32_t col = key >> 1;
        const bool first = valid && (pk == kNoKey || (pk >> 1) != col);
        const bool pair = first && nk != kNoKey && (nk >> 1) == col;
        const int tag = int(key & 1u);
        const double v = __longlong_as_double((long long)val);
        const double a = tag == 0 ? v : 0.0;
        const double b = tag == 1 ? v : (pair ? __longlong_as_double((long long)nv) : 0.0);
        // (pair-resolved builds: an own entry stored negative is final already - minus the merged value of a mutual pair)
        const double m = (tag == 0 && v < 0.0) ? -v : merge_values(a, b, GT_SYMM_ADD, 1.0);
        const bool emit = first;   // ('+': a merged value is never 0 - both parts are >= thresh or absent)
        int total;
        const int pp = wave_prefix_count(emit, lane, total);
        const int e = count + pp;
        if (emit) {
            indices[dst + e] = int32_t(col);
            Kdata[dst + e] = m;
            has_diag |= int64_t(col) == row && m != 0.0;
        }
        msave[t] = m;
        esave[t] = emit ? e : -1;
        // the value goes to lane e % 64 (emitters land on `total` consecutive lanes from count % 64, the others fill the
        // rest: a permutation), which adds it to its partial sum - entries reach a lane in increasing e
        const int dl = emit ? (e & 63) : ((count + total + (lane - pp)) & 63);
        const unsigned long long mb = (unsigned long long)__double_as_longlong(m);
        const int rlo = __builtin_amdgcn_ds_permute(dl << 2, int(uint32_t(mb)));
        const int rhi = __builtin_amdgcn_ds_permute(dl << 2, int(uint32_t(mb >> 32)));
        const double got = __longlong_as_double((long long)((uint64_t(uint32_t(rhi)) << 32) | uint64_t(uint32_t(rlo))));
        if (((lane - count) & 63) < total) lsum += got;
        count += total;
    }
    const double sum = wave_sum_f64(lsum);
    any_diag = __ballot(has_diag) != 0ull;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (esave[t] >= 0) Pdata[dst + esave[t]] = (sum != 0.0) ? msave[t] / sum : msave[t];
    return sum;
}

template <typename K, int NT>
__device__ __forceinline__ double sort_merge_final_row(const RowSrc3& U, const int L, const int lane, const int64_t row,
                                                       int32_t* __restrict__ indices, double* __restrict__ Kdata,
                                                       double* __restrict__ Pdata, const int64_t dst, bool& any_diag) {
    constexpr int PB = SortPos<NT>::bits;
    K pk[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = t * 64 + lane;
        pk[t] = (p < L) ? K(~((K(U.key(p)) << PB) | K(p))) : K(0);
    }
    wave_bitonic_desc<NT, K>(pk, lane);
    uint32_t hi[NT];
    uint64_t lo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        hi[t] = kNoKey;
        lo[t] = 0ull;
        if (pk[t] != K(0)) {
            const K x = K(~pk[t]);
            hi[t] = uint32_t(x >> PB);
            lo[t] = (uint64_t)__double_as_longlong(U.val(int(uint32_t(x) & ((1u << PB) - 1u))));
        }
    }
    return merge_sorted_final<NT>(hi, lo, lane, row, indices, Kdata, Pdata, dst, any_diag);
}

__global__ __launch_bounds__(256) void merge_final_kernel(const int64_t nloc, const FusedSrc fs, const int64_t* __restrict__ indptr,
                                                          int32_t* __restrict__ indices, double* __restrict__ Kdata,
                                                          double* __restrict__ Pdata, double* __restrict__ degree,
                                                          uint32_t* __restrict__ flags, const int key32,
                                                          const int32_t* __restrict__ list) {
    // list (optional): the launch covers these rows (nloc of them) - the rows of 129 ... kBigRow entries that
    // merge_pairs_slots_kernel leaves out
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wi = int64_t(blockIdx.x) * (blockDim.x >> 6) + w;
    if (wi >= nloc) return;
    const int64_t i = list ? int64_t(list[wi]) : wi;
    const int64_t p = fs.pos[i];
    int lt;
    const RowSrc3 U = make_row_src3(fs, i, p, lt);
    const int L = U.ln + lt;
    if (L > kBigRow) return;   // (merge_long_final_kernel)
    const int64_t dst = indptr[i];
    bool any_diag = false;
    double sum;
    const int64_t ic = fs.caller_row(i);   // (the diagonal is where the column is the row's own number - the caller's)
    if (key32) {   // (uniform over the launch)
        if (L <= 64) sum = sort_merge_final_row<uint32_t, 1>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
        else if (L <= 128) sum = sort_merge_final_row<uint32_t, 2>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
        else if (L <= 256) sum = sort_merge_final_row<uint32_t, 4>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
        else sum = sort_merge_final_row<uint32_t, 8>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
    } else {
        if (L <= 64) sum = sort_merge_final_row<uint64_t, 1>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
        else if (L <= 128) sum = sort_merge_final_row<uint64_t, 2>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
        else if (L <= 256) sum = sort_merge_final_row<uint64_t, 4>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
        else sum = sort_merge_final_row<uint64_t, 8>(U, L, lane, ic, indices, Kdata, Pdata, dst, any_diag);
    }
    if (lane == 0) {
        degree[i] = sum;
        if (!any_diag) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
    }
}

// merge_final_kernel for builds whose affinity pass looked the destinations up (GraphState::pairs_fused: every row is a table
// row; tables, counts, union rows by slot).  A wave walks `rpw` consecutive slots with the next rows' headers and the first
// 128 (key, value) pairs of the next row in flight.  A union row of a pair-resolved build holds no column twice (see
// huge_gather_kernel), so rows of up to 128 entries take a short path: sort the composite keys, pick the values up by position
// through the LDS, store - entry e of the sorted row IS entry e of the CSR row, and lane l's share of the row sum is its own
// entries in turn: the sum and every value come out as merge_sorted_final forms them.  Longer rows are left to
// merge_final_kernel (pairs_len_kernel lists them) and merge_long_final_kernel: their register networks in this loop would
// cost every row its occupancy.
template <typename K>
__global__ __launch_bounds__(256) void merge_pairs_slots_kernel(const int64_t nloc, const int rpw, const FusedSrc fs,
                                                                const int32_t* __restrict__ lenN_s, const int32_t* __restrict__ perm,
                                                                const int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                                                double* __restrict__ Kdata, double* __restrict__ Pdata,
                                                                double* __restrict__ degree, uint32_t* __restrict__ flags) {
    __shared__ double park_all[4][128];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    double* park = park_all[w];
    const int64_t t0 = (int64_t(blockIdx.x) * 4 + w) * rpw;
    const int64_t t1 = t0 + rpw < nloc ? t0 + rpw : nloc;
    struct Hdr {
        int64_t o0, sn, dst;
        int32_t i, ln, L;
    };
    struct Dat {
        double va, vb;
        uint32_t ka, kb;
    };
    auto load_hdr = [&](const int64_t p, Hdr& h) {
        if (p < t1) {
            h.i = perm[p];
            h.ln = lenN_s[p];
            h.o0 = fs.off[p];
            h.L = int32_t(fs.off[p + 1] - h.o0);
            h.sn = fs.sN[p];
        }
    };
    auto load_dst = [&](const int64_t p, Hdr& h) {
        if (p < t1) h.dst = indptr[h.i];
    };
    auto load_one = [&](const int64_t p, const Hdr& h, const int q, uint32_t& key, double& val) {
        key = kNoKey;
        val = 0.0;
        if (q < h.L) {
            if (q < h.ln) {
                const uint32_t c = fs.cand_j[size_t(p) * fs.MP + q];
                key = (fs.cmap ? uint32_t(fs.cmap[c]) : c) << 1;
                val = fs.cand_k[size_t(p) * fs.MP + q];
            } else {
                const int64_t r = (h.o0 - h.sn) + (q - h.ln);
                key = (fs.ucol[r] << 1) | 1u;
                val = fs.uval[r];
            }
        }
    };
    auto load_dat = [&](const int64_t p, const Hdr& h, Dat& x) {
        x.ka = x.kb = kNoKey;
        x.va = x.vb = 0.0;
        if (p < t1 && h.L <= 128) {   // (a longer row reads itself where it is sorted)
            load_one(p, h, lane, x.ka, x.va);
            load_one(p, h, lane + 64, x.kb, x.vb);
        }
    };
    // (headers three rows ahead, offsets and entries two: one row of sorting does not cover a round trip)
    Hdr h0 = {0, 0, 0, 0, 0, 0}, h1 = h0, h2 = h0, h3 = h0;
    Dat x0 = {0.0, 0.0, kNoKey, kNoKey}, x1 = x0, x2 = x0;
    load_hdr(t0, h0);
    load_hdr(t0 + 1, h1);
    load_hdr(t0 + 2, h2);
    load_dst(t0, h0);
    load_dst(t0 + 1, h1);
    load_dat(t0, h0, x0);
    load_dat(t0 + 1, h1, x1);
    for (int64_t p = t0; p < t1; ++p) {
        load_hdr(p + 3, h3);
        load_dst(p + 2, h2);
        load_dat(p + 2, h2, x2);
        const int L = h0.L;
        const int64_t row = h0.i, dst = h0.dst;
        const int64_t rowc = fs.caller_row(row);
        if (L <= 128) {
            K pk[2];
            pk[0] = (lane < L) ? K(~((K(x0.ka) << 7) | K(lane))) : K(0);
            pk[1] = (lane + 64 < L) ? K(~((K(x0.kb) << 7) | K(lane + 64))) : K(0);
            park[lane] = x0.va;
            park[lane + 64] = x0.vb;
            if (L <= 64) {   // (uniform)
                K p1[1] = {pk[0]};
                wave_bitonic_desc<1, K>(p1, lane);
                pk[0] = p1[0];
            } else {
                wave_bitonic_desc<2, K>(pk, lane);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double m[2];
            bool has_diag = false;
            double lsum = 0.0;
            uint32_t colv[2] = {0u, 0u};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                m[t] = 0.0;
                if (pk[t] != K(0)) {   // (the entries come out in front: slot e of the sorted row is entry e of the CSR row)
                    const K x = K(~pk[t]);
                    const uint32_t key = uint32_t(x >> 7);
                    const double v = park[int(uint32_t(x) & 127u)];
                    const int tag = int(key & 1u);
                    const uint32_t col = key >> 1;
                    colv[t] = col;
                    m[t] = (tag == 0 && v < 0.0) ? -v : merge_values(tag == 0 ? v : 0.0, tag == 1 ? v : 0.0, GT_SYMM_ADD, 1.0);
                    const int e = t * 64 + lane;
                    indices[dst + e] = int32_t(col);
                    Kdata[dst + e] = m[t];
                    has_diag |= int64_t(col) == rowc && m[t] != 0.0;
                    lsum += m[t];
                }
            }
            const double sum = wave_sum_f64(lsum);
            const bool any_diag = __ballot(has_diag) != 0ull;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (pk[t] != K(0)) Pdata[dst + t * 64 + lane] = (sum != 0.0) ? m[t] / sum : m[t];
            // "a union row of a pair-resolved build holds no column twice" is what this path rests on: rows i and j must agree
            // on whether their pair is mutual.  Should they ever not (round-5 advisor: a repaired row next to a partner that was
            // not), the same column sits in two neighbouring entries of the sorted row - seen here with one shuffle per half, and
            // the build is done again the general way (graph_finish_pairs returns 0) instead of emitting a duplicate column.
            {
                uint32_t n0 = uint32_t(__shfl_down(int(colv[0]), 1));
                const uint32_t first1 = uint32_t(__shfl(int(colv[1]), 0));
                if (lane == 63) n0 = first1;
                const uint32_t n1 = uint32_t(__shfl_down(int(colv[1]), 1));
                const bool dup = (lane + 1 < L && colv[0] == n0) || (lane < 63 && 64 + lane + 1 < L && colv[1] == n1);
                if (__ballot(dup) != 0ull && lane == 0) atomicOr(flags, kFlagPairDupColumn);
            }
            if (lane == 0) {
                degree[row] = sum;
                if (!any_diag) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // (the next row parks into the same slots)
        }   // (longer rows: merge_final_kernel over the listed rows, merge_long_final_kernel)
        h0 = h1;
        h1 = h2;
        h2 = h3;
        x0 = x1;
        x1 = x2;
    }
}

// NT = 16: the rows of 513 ... 1024 entries (135 VGPRs, no scratch).  Longer rows take the segmented sort (kPairHugeRow): the
// kernel that also held the 32-keys-per-lane network needed 265 VGPRs and 272 B of scratch.
template <int NT>
__global__ __launch_bounds__(64) void merge_long_final_kernel(const FusedSrc fs, const int64_t* __restrict__ indptr,
                                                              int32_t* __restrict__ indices, double* __restrict__ Kdata,
                                                              double* __restrict__ Pdata, double* __restrict__ degree,
                                                              uint32_t* __restrict__ flags, const int32_t* __restrict__ biglist,
                                                              const uint32_t* __restrict__ bigcount) {
    const int lane = threadIdx.x;
    const uint32_t nbig = *bigcount;
    for (uint32_t bb = blockIdx.x; bb < nbig; bb += gridDim.x) {
        const int64_t i = biglist[bb];
        const int64_t p = fs.pos[i];
        int lt;
        const RowSrc3 U = make_row_src3(fs, i, p, lt);
        const int L = U.ln + lt;
        const int64_t dst = indptr[i];
        bool any_diag = false;
        const double sum = sort_merge_final_row<uint64_t, NT>(U, L, lane, fs.caller_row(i), indices, Kdata, Pdata, dst, any_diag);
        if (lane == 0) {
            degree[i] = sum;
            if (!any_diag) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
        }
    }
}

// ---- union rows beyond the register sorts, pair-resolved tail -----------------------------------------------------------
// A row of more than kHugeRow entries - a hub of the transpose, or an isolated point whose radius takes in thousands of rows -
// used to refute the pair-resolved tail for the point set: the whole build was done again the general way (twenty isolated
// points in a million: 2 x the time).  Such rows are now finished behind the others: gathered (column, final value) into a
// scratch segment each, sorted by column by rocPRIM's segmented radix sort, written to their place in the CSR with the row
// sum in the order every other path uses (lane l adds the entries l, l + 64, ... in turn, then the wave's tree).
__global__ __launch_bounds__(64) void huge_gather_kernel(const FusedSrc fs, const int32_t* __restrict__ hugelist, const uint32_t nhuge,
                                                         unsigned long long* __restrict__ cursor, uint32_t* __restrict__ seg_begin,
                                                         uint32_t* __restrict__ seg_end, uint32_t* __restrict__ keys,
                                                         double* __restrict__ vals) {
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= nhuge) return;
    const int64_t i = hugelist[-int64_t(b)];   // (the list grows downwards from its last slot)
    const int64_t p = fs.pos[i];
    int lt;
    const RowSrc3 U = make_row_src3(fs, i, p, lt);
    const int L = U.ln + lt;
    unsigned long long base = 0ull;
    if (lane == 0) base = atomicAdd(cursor, (unsigned long long)L);
    base = __shfl((unsigned long long)base, 0);
    if (lane == 0) {
        seg_begin[b] = uint32_t(base);
        seg_end[b] = uint32_t(base) + uint32_t(L);
    }
    for (int q = lane; q < L; q += 64) {
        const uint32_t key = U.key(q);
        const double v = U.val(q);
        const int tag = int(key & 1u);
        // (no column occurs twice in a pair-resolved union row: an own entry stored negative is the settled value of a mutual
        //  pair, every other entry is one side of a pair whose other side is absent)
        const double m = (tag == 0 && v < 0.0) ? -v : merge_values(tag == 0 ? v : 0.0, tag == 1 ? v : 0.0, GT_SYMM_ADD, 1.0);
        keys[base + q] = key >> 1;
        vals[base + q] = m;
    }
}

__global__ __launch_bounds__(64) void huge_finalize_kernel(const int32_t* __restrict__ hugelist, const uint32_t nhuge,
                                                           const uint32_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_end,
                                                           const uint32_t* __restrict__ keys, const double* __restrict__ vals,
                                                           const int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                                           double* __restrict__ Kdata, double* __restrict__ Pdata,
                                                           double* __restrict__ degree, uint32_t* __restrict__ flags,
                                                           const int32_t* __restrict__ cmap, const int64_t row0) {
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= nhuge) return;
    const int64_t i = hugelist[-int64_t(b)];
    const int64_t ic = cmap ? int64_t(cmap[row0 + i]) : i;   // (FusedSrc::caller_row)
    const uint32_t s0 = seg_begin[b];
    const int L = int(seg_end[b] - s0);
    const int64_t dst = indptr[i];
    double lsum = 0.0;
    bool has_diag = false;
    for (int e = lane; e < L; e += 64) {
        const uint32_t col = keys[s0 + e];
        const double m = vals[s0 + e];
        indices[dst + e] = int32_t(col);
        Kdata[dst + e] = m;
        has_diag |= int64_t(col) == ic && m != 0.0;
        lsum += m;
    }
    const double sum = wave_sum_f64(lsum);
    const bool any_diag = __ballot(has_diag) != 0ull;
    for (int e = lane; e < L; e += 64) {
        const double m = vals[s0 + e];
        Pdata[dst + e] = (sum != 0.0) ? m / sum : m;
    }
    if (lane == 0) {
        degree[i] = sum;
        if (!any_diag) atomicOr(flags, GT_FLAG_ZERO_DIAGONAL);
    }
}

__global__ __launch_bounds__(256) void anisotropy_kernel(const int64_t nloc, const int64_t r0,
                                                         const int64_t* __restrict__ indptr,
                                                         const int32_t* __restrict__ indices, double* __restrict__ Kdata,
                                                         const double* __restrict__ degree_all, const double alpha,
                                                         double* __restrict__ degree_out,
                                                         const int32_t* __restrict__ rowid) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int64_t s = indptr[i], e1 = indptr[i + 1];
    const double qi = degree_all[rowid ? int64_t(rowid[r0 + i]) : r0 + i];   // (rowid: renumbered points, degree_all in the caller's numbering)
    double sum = 0.0;
    for (int64_t e = s + lane; e < e1; e += 64) {
        const double v = Kdata[e] / pow(qi * degree_all[indices[e]], alpha);
        Kdata[e] = v;
        sum += v;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) degree_out[i] = sum;
}

// D^-1/2 K D^-1/2 (BaseGraph.diff_aff, base.py:668-698) on the structure of K: the reference forms it as two sparse
// products with the diagonal matrix 1 / sqrt(degree) - (D K) D - i.e. every entry is rounded after each of the two
// multiplications, left factor first
__global__ __launch_bounds__(256) void diff_aff_kernel(const int64_t nloc, const int64_t r0, const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices, const double* __restrict__ Kdata,
                                                       const double* __restrict__ degree_all, double* __restrict__ out,
                                                       const int32_t* __restrict__ rowid) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int64_t s = indptr[i], e1 = indptr[i + 1];
    const double di = 1.0 / sqrt(degree_all[rowid ? int64_t(rowid[r0 + i]) : r0 + i]);
    for (int64_t e = s + lane; e < e1; e += 64) {
        const double left = di * Kdata[e];
        out[e] = left * (1.0 / sqrt(degree_all[indices[e]]));
    }
}

__global__ __launch_bounds__(256) void normalize_kernel(const int64_t nloc, const int64_t* __restrict__ indptr,
                                                        const double* __restrict__ Kdata, double* __restrict__ Pdata) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int64_t s = indptr[i], e1 = indptr[i + 1];
    double sum = 0.0;
    for (int64_t e = s + lane; e < e1; e += 64) sum += fabs(Kdata[e]);
    sum = wave_sum_f64(sum);
    // sklearn inplace_csr_row_normalize_l1: rows with zero sum are left untouched
    for (int64_t e = s + lane; e < e1; e += 64) Pdata[e] = (sum != 0.0) ? Kdata[e] / sum : Kdata[e];
}

// out[idx[i]] = in[i] (the degrees of a renumbered point set in the caller's numbering)
__global__ __launch_bounds__(256) void scatter_f64_kernel(const double* __restrict__ in, const int32_t* __restrict__ idx,
                                                          const int64_t n, double* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[idx[i]] = in[i];
}

int exclusive_scan(gt_ctx* ctx, const int32_t* a, const int32_t* b, int64_t n, int64_t* out, DevBuf& tmp) {
    const int64_t nb = ceil_div64(n, 1024);
    GT_HIP(ctx, tmp.reserve(size_t(nb + 1) * sizeof(int64_t)));
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, a, b, n, out, tmp.as<int64_t>());
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(64), 0, ctx->stream, tmp.as<int64_t>(), nb);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, out, n, tmp.as<int64_t>(), nb);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// ---- C0: ingest an arbitrary CSR kernel (gt_csr_graph_build) into the radius-list layout the pipeline reads ----
__global__ __launch_bounds__(256) void csr_ingest_kernel(const int64_t n, const int64_t* __restrict__ indptr,
                                                         const int32_t* __restrict__ indices,
                                                         const double* __restrict__ data, const int32_t rcap,
                                                         const int count_owners, uint64_t* __restrict__ rlists,
                                                         double* __restrict__ rK, uint32_t* __restrict__ rcounts,
                                                         int32_t* __restrict__ rowsrc, int32_t* __restrict__ lenN,
                                                         int32_t* __restrict__ ownercnt) {
    const int lane = threadIdx.x & 63;
    const int64_t i = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t s0 = indptr[i];
    const int32_t len = int32_t(indptr[i + 1] - s0);
    for (int32_t e = lane; e < len; e += 64) {
        rlists[size_t(i) * rcap + e] = cand_pack(0.f, uint32_t(indices[s0 + e]));
        rK[size_t(i) * rcap + e] = data[s0 + e];
    }
    if (lane == 0) {
        rcounts[i] = uint32_t(len);
        rowsrc[i] = int32_t(i);
        lenN[i] = len;
        if (count_owners) ownercnt[i] = len;
    }
}

// out[r] = v[r * stride], r < count (the bucket edges of the owner-major scan)
__global__ void gather_strided_i64_kernel(const int64_t* __restrict__ v, const int64_t stride, const int count,
                                          int64_t* __restrict__ out) {
    for (int r = threadIdx.x; r < count; r += blockDim.x) out[r] = v[int64_t(r) * stride];
}

Splits make_splits(const GraphState* g) {
    Splits sp;
    sp.world = g->world;
    for (int r = 0; r <= g->world; ++r) sp.s[r] = g->splits[r];
    return sp;
}

template <typename T>
void launch_affinity(gt_ctx* ctx, GraphState* g, KnnWork* k, int binary, double decay, double thresh, int count_owners) {
    // (rows per workgroup: a workgroup's wave slots are handed on when its LAST wave is done, and the rows' costs differ)
    const int wpb = 1;
    const size_t lds = size_t(wpb) * ctx->d * sizeof(double);
#define GT_AFFINITY_LAUNCH(RADIUS_, PAIRS_, LIST_, NROWS_, POSJ_)                                                                  \
    hipLaunchKernelGGL((affinity_kernel<T, RADIUS_, PAIRS_>), dim3((unsigned)ceil_div64(NROWS_, wpb)), dim3(64 * wpb), lds, ctx->stream, \
                       LIST_, int64_t(NROWS_), g->nloc, g->r0, (const T*)ctx->X, ctx->d, ctx->xn.as<double>(),              \
                       (const T*)g->Qmat, g->qnorm, g->qoff, gt_dist_dtype(ctx), ctx->metric, k->MP, g->limit,            \
                       k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), k->cand_n.as<uint32_t>(), g->rowsrc.as<int32_t>(), \
                       g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(),                   \
                       bwp, decay, binary, thresh, count_owners, make_splits(g), g->lenN.as<int32_t>(),                      \
                       g->ownercnt.as<int32_t>(), g->tablen.as<int32_t>(), g->pairs ? (g->pairs_shard ? (g->pairs_shard_bins ? 3 : 2) : 1) : 0, \
                       g->pairs ? k->cand_d2t.as<double>() : (const double*)nullptr,                                       \
                       g->pairs ? k->keyt_ok.as<uint8_t>() : (const uint8_t*)nullptr, g->radius_factor * (1.0 + 1e-9),      \
                       tperm, trow, POSJ_, g->sC.as<int64_t>(), k->tab_sorted ? g->cnt_sorted.as<int32_t>() : (int32_t*)nullptr, \
                       bw_i0, g->rank)
    // (row-sharded pair-resolved build: the bandwidths of ALL rows, this rank's from row r0 on - gt_graph_set_bandwidths)
    const double* bwp = g->pairs_shard ? g->bw_all.as<double>() : g->bw.as<double>();
    const int64_t bw_i0 = g->pairs_shard ? g->r0 : 0;
    // (... whose local one-sided entries go through the rank's own destination bins: their destinations by the scan of the tables)
    uint32_t* shard_posj = g->pairs_shard_bins ? g->cursor.as<uint32_t>() : (uint32_t*)nullptr;
    const int32_t* tperm = k->tab_sorted ? k->qorder.as<int32_t>() : (const int32_t*)nullptr;
    const int32_t* trow = k->tab_sorted ? k->sh_invperm.as<int32_t>() : (const int32_t*)nullptr;
    if (g->pairs && k->tab_sorted) {
        // (eight consecutive slots per wave: 2 ... 8 measured alike on C3, 16 slower - the longer run of a wave's last rows)
        const int rpw = 8;
        const bool fz = g->pairs_fused;
        hipLaunchKernelGGL(affinity_slots_kernel, dim3((unsigned)ceil_div64(g->nloc, int64_t(4) * rpw)), dim3(256), 0, ctx->stream,
                           g->nloc, rpw, (const SlotRec*)g->rec_s.p, (const BwPos*)g->bwpos.p, gt_dist_dtype(ctx), ctx->metric, k->MP,
                           g->limit, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), k->cand_d2t.as<double>(), decay, binary,
                           thresh, g->radius_factor * (1.0 + 1e-9), fz ? 0 : count_owners, g->lenN.as<int32_t>(), g->ownercnt.as<int32_t>(),
                           g->tablen.as<int32_t>(), fz ? g->cursor.as<uint32_t>() : (uint32_t*)nullptr, g->sC.as<int64_t>(),
                           g->cnt_sorted.as<int32_t>(), make_splits(g), g->rank);
        if (k->nokeyt_n > 0)
            GT_AFFINITY_LAUNCH(false, 2, k->nokeyt_rows.as<int32_t>(), int64_t(k->nokeyt_n),   // (qoff = 0 here)
                               fz ? g->cursor.as<uint32_t>() : (uint32_t*)nullptr);
        if (fz)
            hipLaunchKernelGGL(posj_hist_kernel, dim3(2048), dim3(256), size_t(g->bin_count) * sizeof(int32_t), ctx->stream,
                               g->cursor.as<uint32_t>(), g->sc_total, g->bin_shift, g->bin_count, g->bincnt.as<int32_t>());
    } else if (g->pairs_shard_bins && k->keyt_valid) {
        // a rank of a sharded build whose tables carry the transposed keys and whose local one-sided entries go through its own
        // bins: the slot kernel (slot = local row; the records graph_begin_b prepared: shard_slot_records_kernel)
        const int rpw = 8;
        hipLaunchKernelGGL(affinity_slots_kernel, dim3((unsigned)ceil_div64(g->nloc, int64_t(4) * rpw)), dim3(256), 0, ctx->stream,
                           g->nloc, rpw, (const SlotRec*)g->rec_s.p, (const BwPos*)g->bwpos.p, gt_dist_dtype(ctx), ctx->metric, k->MP,
                           g->limit, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), k->cand_d2t.as<double>(), decay, binary,
                           thresh, g->radius_factor * (1.0 + 1e-9), count_owners ? 2 : 0, g->lenN.as<int32_t>(), g->ownercnt.as<int32_t>(),
                           g->tablen.as<int32_t>(), shard_posj, g->sC.as<int64_t>(), g->lenN.as<int32_t>(), make_splits(g), g->rank);
        if (k->nokeyt_n > 0) GT_AFFINITY_LAUNCH(false, 2, k->nokeyt_rows.as<int32_t>(), int64_t(k->nokeyt_n), shard_posj);
    } else if (g->pairs && g->pairs_shard && !k->keyt_valid) {
        // (a rank whose rows went through the classic pass: no table carries transposed keys - every kept entry's comes from its
        //  dot product; keyt_ok is all zeros, graph_begin_b)
        GT_AFFINITY_LAUNCH(false, 2, (const int32_t*)nullptr, g->nloc, shard_posj);
    } else if (g->pairs) {
        GT_AFFINITY_LAUNCH(false, 1, (const int32_t*)nullptr, g->nloc, shard_posj);
        // (the list holds rows of the query matrix, qoff + i: qoff = 0 on one rank, r0 on a rank of a sharded build)
        if (k->nokeyt_n > 0) GT_AFFINITY_LAUNCH(false, 2, k->nokeyt_rows.as<int32_t>(), int64_t(k->nokeyt_n), shard_posj);
    } else {
        GT_AFFINITY_LAUNCH(false, 0, (const int32_t*)nullptr, g->nloc, (uint32_t*)nullptr);
    }
    if (g->n_over > 0) GT_AFFINITY_LAUNCH(true, 0, g->over_rows.as<int32_t>(), g->n_over, (uint32_t*)nullptr);
    if (g->pairs_shard_bins)
        hipLaunchKernelGGL(posj_hist_kernel, dim3(2048), dim3(256), size_t(g->bin_count) * sizeof(int32_t), ctx->stream,
                           g->cursor.as<uint32_t>(), g->sc_total, g->bin_shift, g->bin_count, g->bincnt.as<int32_t>());
#undef GT_AFFINITY_LAUNCH
}

}  // namespace

int gt_launch_max_u32(gt_ctx* ctx, const uint32_t* v, int64_t n, uint32_t* out) {
    hipLaunchKernelGGL(max_u32_kernel, dim3(64), dim3(256), 0, ctx->stream, v, n, out);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_exclusive_scan_i32(gt_ctx* ctx, const int32_t* a, int64_t n, int64_t* out) {
    DevBuf tmp;
    int rc = exclusive_scan(ctx, a, nullptr, n, out, tmp);
    if (rc == GT_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->set_error(std::string("scan: ") + hipGetErrorString(e));
            rc = GT_E_HIP;
        }
    }
    tmp.release();
    return rc;
}

// ================================================================================================
static double candidate_hint(const gt_ctx* ctx, const gt_knn_params* params, double thresh, bool use_radius) {
    double hint = 1.0;
    if (use_radius && params->bandwidth_len == 0) {
        const double rf = std::pow(-1.0 * std::log(thresh), 1.0 / params->decay) * params->bandwidth_scale;
        hint = (ctx->metric == 1) ? rf : rf * rf;
    } else if (use_radius) {
        hint = -1.0;   // negative: no early stop in the re-rank (gt_knn.h)
    }
    return hint;
}

// ---- row-sharded symmetric candidate pass: the stages around the host's collectives (gt_knn_shard.cpp) -------------
int gt_knn_shard_plan(gt_ctx* ctx, int world, int rank, const int64_t* splits, int need_m, double rkf, int32_t* applies,
                      int64_t* n_pad_sorted, int64_t* sorted_splits);
int gt_knn_shard_seed(gt_ctx* ctx, float* thr_local, int64_t* far_local, double* racc_local);
int gt_knn_shard_collect(gt_ctx* ctx, const float* thr_all, int64_t far_total, const double* racc_total, int32_t* applies,
                         int64_t* send_counts);
int gt_knn_shard_emit(gt_ctx* ctx, void* send_buf);
int gt_knn_shard_finish(gt_ctx* ctx, const void* recv, int64_t n_recv);

extern "C" int gt_graph_sym_plan(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                                 const int64_t* row_splits, int32_t* applies, int64_t* n_pad_sorted, int64_t* sorted_splits) {
    if (!ctx || !params || !applies || !n_pad_sorted || !sorted_splits || !row_splits) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    *applies = 0;
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world) GT_FAIL(ctx, GT_E_ARG, "gt_graph_sym_plan: bad world/rank");
    if (row_splits[0] != 0 || row_splits[world] != ctx->n) GT_FAIL(ctx, GT_E_ARG, "row_splits must cover [0, n]");
    // the cases whose table depth and re-rank hint graph_begin_impl derives without further state
    if (params->knn < 1 || params->knn_max > 0 || int64_t(params->knn) + 1 > ctx->n) return GT_OK;
    const bool binary = std::isnan(params->decay) || params->thresh == 1.0;
    double thresh = params->thresh;
    if (!binary) {
        if (thresh <= 0) return GT_OK;
        if (thresh < DBL_EPSILON) thresh = DBL_EPSILON;
    }
    const double hint = candidate_hint(ctx, params, thresh, !binary);
    return gt_knn_shard_plan(ctx, world, rank, row_splits, params->knn + 1, hint, applies, n_pad_sorted, sorted_splits);
}
extern "C" int gt_graph_sym_seed(gt_ctx* ctx, void* thr_local, int64_t* far_local, double* radius_local) {
    if (!ctx || !far_local || !radius_local) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    return gt_knn_shard_seed(ctx, static_cast<float*>(thr_local), far_local, radius_local);
}
extern "C" int gt_graph_sym_collect(gt_ctx* ctx, const void* thr_all, int64_t far_total, const double* radius_total,
                                    int32_t* applies, int64_t* send_counts) {
    if (!ctx || !thr_all || !radius_total || !applies || !send_counts) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    return gt_knn_shard_collect(ctx, static_cast<const float*>(thr_all), far_total, radius_total, applies, send_counts);
}
extern "C" int gt_graph_sym_emit(gt_ctx* ctx, void* send_buf) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    return gt_knn_shard_emit(ctx, send_buf);
}
extern "C" int gt_graph_sym_finish(gt_ctx* ctx, const void* recv, int64_t n_recv) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    return gt_knn_shard_finish(ctx, recv, n_recv);
}

// Row-sharded build on renumbered points (gt_points_cell_sort; row_splits from gt_points_shard_splits): the candidate lists
// of the rank's own rows, collected by the rank alone (gt_knn_shard.cpp gt_knn_shard_local) - no exchange.  applies = 0:
// gt_graph_begin will run the classic candidate pass for these rows instead (every rank decides for itself).
int gt_knn_shard_local(gt_ctx* ctx, int64_t r0, int64_t r1, int need_m, double rkf, int32_t* applies);
extern "C" int gt_graph_shard_local(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                                    const int64_t* row_splits, int32_t* applies) {
    if (!ctx || !params || !applies || !row_splits) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    *applies = 0;
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world) GT_FAIL(ctx, GT_E_ARG, "gt_graph_shard_local: bad world/rank");
    if (row_splits[0] != 0 || row_splits[world] != ctx->n) GT_FAIL(ctx, GT_E_ARG, "row_splits must cover [0, n]");
    if (row_splits[rank + 1] <= row_splits[rank]) return GT_OK;
    if (params->knn < 1 || params->knn_max > 0 || int64_t(params->knn) + 1 > ctx->n) return GT_OK;
    const bool binary = std::isnan(params->decay) || params->thresh == 1.0;
    double thresh = params->thresh;
    if (!binary) {
        if (thresh <= 0) return GT_OK;
        if (thresh < DBL_EPSILON) thresh = DBL_EPSILON;
    }
    const double hint = candidate_hint(ctx, params, thresh, !binary);
    return gt_knn_shard_local(ctx, row_splits[rank], row_splits[rank + 1], params->knn + 1, hint, applies);
}

__global__ __launch_bounds__(256) void gather_f64_kernel(const double* __restrict__ in, const int32_t* __restrict__ idx,
                                                         const int64_t n, double* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

// gt_graph_begin in two halves.  graph_begin_a: the candidate tables of the owned rows and their bandwidths (+ the replay of the
// reference's knn_max loop); graph_begin_b: radius pass, affinities, the counts of what travels.  One call runs both; a rank of a
// row-sharded build may stop between them (gt_graph_bandwidth_local) so that the caller can gather the bandwidths of ALL rows -
// what the pair-resolved tail needs to know of a partner - and hand them back (gt_graph_set_bandwidths) before the second half.
// *done (graph_begin_a): the call has ended (gt_graph_stage_counts wanted the counts only).
static bool shard_pairs_static(const gt_ctx* ctx, const gt_knn_params* params, int64_t nloc) {
    // what every rank of a build decides alike: the '+' rule over a decaying kernel, nothing the tail does not serve
    const bool binary = std::isnan(params->decay) || params->thresh == 1.0;
    return ctx->symm_pairs != 0 && ctx->symm_pairs_shard != 0 && !binary && params->knn_max <= 0 &&
           params->kernel_symm == GT_SYMM_ADD && params->anisotropy == 0.0 && ctx->n < (int64_t(1) << 31) && nloc < (int64_t(1) << 31);
}
static int graph_begin_a(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                         const int64_t* row_splits, int64_t* send_counts, bool external, int64_t m_ext, bool want_keyt, bool* done) {
    *done = false;
    if (!ctx || !params) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->n <= 0) GT_FAIL(ctx, GT_E_STATE, "gt_graph_begin: no points bound");
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || !row_splits || !send_counts)
        GT_FAIL(ctx, GT_E_ARG, "gt_graph_begin: bad world/rank/row_splits");
    if (row_splits[0] != 0 || row_splits[world] != (external ? m_ext : ctx->n))
        GT_FAIL(ctx, GT_E_ARG, "row_splits must cover [0, n]");
    for (int r = 0; r < world; ++r)
        if (row_splits[r + 1] < row_splits[r]) GT_FAIL(ctx, GT_E_ARG, "row_splits must be ascending");
    if (params->knn < 1) GT_FAIL(ctx, GT_E_ARG, "knn must be >= 1");
    if (!ctx->graph) ctx->graph = new GraphState();
    GraphState* g = ctx->graph;
    g->p = *params;
    g->world = world;
    g->rank = rank;
    g->splits.assign(row_splits, row_splits + world + 1);
    g->r0 = row_splits[rank];
    g->r1 = row_splits[rank + 1];
    g->nloc = g->r1 - g->r0;
    g->n_total = ctx->n;   // columns: the bound points
    g->begun = false;
    g->finished = false;
    g->aniso_applied = false;
    g->external = external;
    g->n_over = 0;
    g->radius_retries = 0;
    g->rcap = 0;
    if (g->nloc <= 0) GT_FAIL(ctx, GT_E_ARG, "gt_graph_begin: this rank owns no rows");
    const bool binary = std::isnan(params->decay) || params->thresh == 1.0;
    double thresh = params->thresh;
    if (!binary) {
        if (thresh <= 0 && params->knn_max <= 0)
            GT_FAIL(ctx, GT_E_ARG, "thresh <= 0 needs knn_max (use the exact dense graph instead)");
        if (thresh < DBL_EPSILON) thresh = DBL_EPSILON;   // graphs.py:628-629
    }
    g->p.thresh = thresh;
    // build_kernel: knn + 1 neighbours (self included, graphs.py:783-784); build_kernel_to_data(Y): knn (:854-855)
    const int kprime = external ? params->knn : params->knn + 1;
    if (int64_t(kprime) > ctx->n) GT_FAIL(ctx, GT_E_ARG, "knn + 1 exceeds n_samples");
    int need = kprime;
    bool use_radius = !binary;
    // knn_max: table sizes the reference's search-expansion loop tries (graphs.py:882, 916-940): 6 k', 36 k', ... capped
    // at knn_max' = knn_max (+1 with self).  The exact tables are built as deep as knn_max' when the kernels can hold
    // it (447 neighbours), else as deep as the largest step below - enough whenever the loop ends in its radius branch.
    int64_t km = 0;
    int kst[4] = {0, 0, 0, 0};
    int n_kst = 0;
    const int kMaxTable = 448;
    if (!binary && params->knn_max > 0) {
        km = std::max<int64_t>(kprime, std::min<int64_t>(external ? params->knn_max : params->knn_max + 1, ctx->n));
        for (int64_t sv = std::min<int64_t>(int64_t(kprime) * 6, km); n_kst < 4; sv = std::min<int64_t>(sv * 6, km)) {
            kst[n_kst++] = int(std::min<int64_t>(sv, 1 << 30));
            if (sv >= km) break;
        }
        need = int(km);
        if (km > kMaxTable || world > 1) {
            // sharded builds replay the loop on the row counts of ALL ranks: the host sums the ranks' counts
            // (gt_graph_stage_counts) and hands the totals back (gt_graph_set_stage_totals); without that exchange a sharded
            // build caps every row, as it always used to
            if (world > 1 && km <= kMaxTable) {
                if (!ctx->stage_counts_only && !ctx->stage_totals_valid) n_kst = 0;
            } else {
                need = 0;
                for (int t = 0; t < n_kst; ++t)
                    if (kst[t] <= kMaxTable && kst[t] < km) need = kst[t];
                if (need < kprime || world > 1)
                    GT_FAIL(ctx, GT_E_LIMIT, "knn_max beyond 447 neighbours is only supported when the search ends in its radius branch");
            }
        }
        use_radius = false;
    }
    g->need_m = need;
    if (params->bandwidth_len != 0 && params->bandwidth_len != 1 && params->bandwidth_len != (external ? m_ext : ctx->n))
        GT_FAIL(ctx, GT_E_ARG, "bandwidth must have 1 or n_samples entries");
    if (params->bandwidth_len > 0 && !params->bandwidth) GT_FAIL(ctx, GT_E_ARG, "bandwidth pointer is NULL");

    // ---- kNN candidates for the owned rows ----
    {
        // hint for the arithmetic choice of the main pass: rows out to radius_factor x bandwidth will be needed
        // (it also lets the re-rank stop after its first batch of candidates); a caller-given bandwidth is not tied to
        // the k-th neighbour: no hint, every candidate is evaluated
        const double hint = candidate_hint(ctx, params, thresh, use_radius);
        // A build that will take the pair-resolved tail (the static half of g->pairs below; the other half - the symmetric
        // pass wrote the transposed keys - is what makes the re-rank honour the request) asks for its tables by sorted
        // position: every pass of the tail walks the rows in that order.
        if (!ctx->knn) ctx->knn = new KnnWork();
        ctx->knn->want_tab_sorted = ctx->in_graph_build && ctx->symm_pairs == 2 && ctx->symm_pair_ok != 0 && ctx->symm_bins != 0 &&
                                    world == 1 && !external && !binary && params->knn_max <= 0 &&
                                    params->kernel_symm == GT_SYMM_ADD && params->anisotropy == 0.0 && g->r0 == 0 &&
                                    g->nloc == ctx->n && !ctx->presorted && g->nloc < (int64_t(1) << 31) &&
                                    (ctx->symm_bins > 0 || g->nloc >= 65536);
        // (a rank of a row-sharded build that will take the pair-resolved tail: its re-rank writes the transposed keys too)
        ctx->knn->want_keyt_shard = want_keyt;
        // (... and so does the re-rank of the classic pass in a single-rank build whose parameters are the tail's - isotropic data
        //  never reach the symmetric pass, their tail was the general one until round 6)
        ctx->knn->want_keyt_classic = ctx->in_graph_build && ctx->symm_pairs != 0 && ctx->symm_pair_ok != 0 && ctx->symm_bins != 0 &&
                                      world == 1 && !external && !binary && params->knn_max <= 0 &&
                                      params->kernel_symm == GT_SYMM_ADD && params->anisotropy == 0.0 && g->r0 == 0 &&
                                      g->nloc == ctx->n && !ctx->presorted && g->nloc < (int64_t(1) << 31) &&
                                      (ctx->symm_bins > 0 || g->nloc >= 65536);
        const int rc_knn = gt_knn_candidates(ctx, g->r0, g->nloc, external, need, hint);
        ctx->knn->want_tab_sorted = false;
        ctx->knn->want_keyt_shard = false;
        ctx->knn->want_keyt_classic = false;
        if (rc_knn != GT_OK) return rc_knn;
    }
    KnnWork* k = ctx->knn;
    g->Qmat = external ? k->Qraw.p : ctx->X;
    g->qnorm = external ? k->qn.as<double>() : ctx->xn.as<double>();
    g->qoff = g->r0;
    g->limit = binary ? kprime : (params->knn_max > 0 ? need : k->MP);

    GT_HIP(ctx, g->bw.reserve(size_t(g->nloc) * sizeof(double)));
    if (k->tab_sorted) {
        GT_HIP(ctx, g->cnt_sorted.reserve(size_t(g->nloc) * sizeof(int32_t)));   // (lenN by slot, written by the affinity launches)
        GT_HIP(ctx, g->rec_s.reserve(size_t(g->nloc) * sizeof(SlotRec)));
        GT_HIP(ctx, g->bwpos.reserve(size_t(g->nloc) * sizeof(BwPos)));
    }
    GT_HIP(ctx, g->rowsrc.reserve(size_t(g->nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->lenN.reserve(size_t(g->nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->tablen.reserve(size_t(g->nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->lenT.reserve(size_t(g->nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->cursor.reserve(size_t(g->nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->over_rows.reserve(size_t(g->nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->rthr.reserve(size_t(g->nloc) * sizeof(float)));
    GT_HIP(ctx, g->over_count.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, g->rmax.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, g->ownercnt.reserve(size_t(int64_t(world) * g->nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->flags.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(g->over_count.p, 0, sizeof(uint32_t), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(g->flags.p, 0, sizeof(uint32_t), ctx->stream));
    if (params->bandwidth_len > 0) {
        GT_HIP(ctx, g->bw_user.reserve(size_t(params->bandwidth_len) * sizeof(double)));
        GT_HIP(ctx, hipMemcpyAsync(g->bw_user.p, params->bandwidth, size_t(params->bandwidth_len) * sizeof(double),
                                   hipMemcpyHostToDevice, ctx->stream));
        if (ctx->presorted && !external && params->bandwidth_len == ctx->n) {
            // one bandwidth per row of the CALLER's numbering: into the context's
            GT_HIP(ctx, g->deg_caller.reserve(size_t(ctx->n) * sizeof(double)));
            GT_HIP(ctx, hipMemcpyAsync(g->deg_caller.p, g->bw_user.p, size_t(ctx->n) * sizeof(double), hipMemcpyDeviceToDevice,
                                       ctx->stream));
            hipLaunchKernelGGL(gather_f64_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream,
                               g->deg_caller.as<double>(), ctx->vperm.as<int32_t>(), ctx->n, g->bw_user.as<double>());
            GT_HIP(ctx, hipGetLastError());
        }
    }
    g->radius_factor = binary ? 0.0 : std::pow(-1.0 * std::log(thresh), 1.0 / params->decay);   // graphs.py:902-904
    // repairs (radius pass) run on the accurate arithmetic of the working copy, whatever the main pass used
    const ErrModel err_model = gt_err_model(ctx, ctx->prec);
    // norms the candidate pass saw (partial norms for wide data): they bound the scores of everything inside a radius
    const double* qn_bound = !ctx->wide ? g->qnorm : (external ? k->qn_sel.as<double>() : ctx->xn_sel.as<double>());
    // tables by sorted position (KnnWork::tab_sorted, asked for above): row -> slot
    const int32_t* trow = k->tab_sorted ? k->sh_invperm.as<int32_t>() : (const int32_t*)nullptr;
    {
        StageSpan span(ctx, "affinity");
        hipLaunchKernelGGL(bandwidth_kernel, dim3((unsigned)ceil_div64(g->nloc, 256)), dim3(256), 0, ctx->stream, g->nloc,
                           g->r0, k->MP, kprime, gt_dist_dtype(ctx), ctx->metric, k->cand_d2.as<double>(), k->d2_lb.as<double>(),
                           qn_bound, g->qnorm, g->qoff, ctx->ymax.as<double>(), err_model, g->bw_user.as<double>(),
                           params->bandwidth_len, params->bandwidth_scale, use_radius ? 1 : 0, g->radius_factor,
                           g->bw.as<double>(), g->rowsrc.as<int32_t>(), g->over_rows.as<int32_t>(),
                           g->over_count.as<uint32_t>(), g->rthr.as<float>(), trow, trow ? (SlotRec*)g->rec_s.p : (SlotRec*)nullptr,
                           trow ? (BwPos*)g->bwpos.p : (BwPos*)nullptr, k->cand_n.as<uint32_t>(), k->keyt_ok.as<uint8_t>());
        GT_HIP(ctx, hipGetLastError());
    }
    if (n_kst > 0) {
        // knn_max does not always cap (graphs.py:916-976): the reference escalates search_knn = 6 k', 36 k', ... <= knn_max'
        // while more than a tenth of the rows still have their whole table inside their radius, and what is left over at
        // the end is searched out to knn_max' only if the escalation got there - otherwise by RADIUS, without a cap.
        // The exact tables answer every step of that loop they are deep enough for; in the radius case the build
        // continues like one without knn_max (rows whose radius reaches past their table take the radius pass).
        int n_avail = 0;
        while (n_avail < n_kst && kst[n_avail] <= need) ++n_avail;
        GT_HIP(ctx, g->rmax.reserve(4 * sizeof(uint32_t)));
        GT_HIP(ctx, hipMemsetAsync(g->rmax.p, 0, 4 * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(stage_count_kernel, dim3((unsigned)ceil_div64(g->nloc, 256)), dim3(256), 0, ctx->stream, g->nloc,
                           k->MP, gt_dist_dtype(ctx), ctx->metric, k->cand_d2.as<double>(), g->bw.as<double>(), g->radius_factor,
                           make_int4(kst[0], kst[1], kst[2], kst[3]), n_avail, g->rmax.as<uint32_t>());
        GT_HIP(ctx, hipGetLastError());
        uint32_t un[4] = {0, 0, 0, 0};
        GT_HIP(ctx, hipMemcpyAsync(un, g->rmax.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int64_t rows_total = g->nloc;
        if (ctx->stage_counts_only) {
            // gt_graph_stage_counts: this rank's counts are all the caller wants of this pass
            ctx->stage_n = n_avail;
            for (int t = 0; t < 4; ++t) ctx->stage_local[t] = t < n_avail ? int64_t(un[t]) : 0;
            for (int r = 0; r < world; ++r) send_counts[r] = 0;
            *done = true;
            return GT_OK;
        }
        if (world > 1) {
            // the counts over the rows of every rank (the reference's loop looks at the whole point set)
            for (int t = 0; t < 4; ++t) un[t] = uint32_t(std::min<int64_t>(ctx->stage_totals[t], 0xFFFFFFFFll));
            rows_total = ctx->n;
        }
        // replay of the loop: t = step the remaining rows were last searched with, `next` = the table size tried next
        int t = 0;
        int64_t next = std::min<int64_t>(int64_t(kst[0]) * 6, km);
        uint32_t remaining = un[0];
        bool blind = n_avail < 1;
        while (!blind && int64_t(remaining) > rows_total / 10 && double(next) < double(ctx->n) / 2.0 && next < km) {
            ++t;
            if (t >= n_avail) {
                blind = true;
                break;
            }
            remaining = un[t];
            next = std::min<int64_t>(next * 6, km);
        }
        const bool capped = !blind && remaining > 0 && next == km;
        if (blind || (capped && need != km))
            GT_FAIL(ctx, GT_E_LIMIT, "knn_max beyond 447 neighbours is only supported when the search ends in its radius branch");
        if (remaining > 0 && !capped) {
            // the reference gives up by radius search: no cap
            use_radius = true;
            g->limit = k->MP;
            GT_HIP(ctx, hipMemsetAsync(g->over_count.p, 0, sizeof(uint32_t), ctx->stream));
            StageSpan span(ctx, "affinity");
            hipLaunchKernelGGL(bandwidth_kernel, dim3((unsigned)ceil_div64(g->nloc, 256)), dim3(256), 0, ctx->stream, g->nloc,
                               g->r0, k->MP, kprime, gt_dist_dtype(ctx), ctx->metric, k->cand_d2.as<double>(), k->d2_lb.as<double>(),
                               qn_bound, g->qnorm, g->qoff, ctx->ymax.as<double>(), err_model, g->bw_user.as<double>(),
                               params->bandwidth_len, params->bandwidth_scale, 1, g->radius_factor,
                               g->bw.as<double>(), g->rowsrc.as<int32_t>(), g->over_rows.as<int32_t>(),
                               g->over_count.as<uint32_t>(), g->rthr.as<float>(), trow, trow ? (SlotRec*)g->rec_s.p : (SlotRec*)nullptr,
                           trow ? (BwPos*)g->bwpos.p : (BwPos*)nullptr, k->cand_n.as<uint32_t>(), k->keyt_ok.as<uint8_t>());
            GT_HIP(ctx, hipGetLastError());
        } else if (need != km) {
            // every row finished inside the tables: nothing is capped, nothing is missing
            g->limit = need;
        }
    }
    return GT_OK;
}

static int graph_begin_b(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int64_t* send_counts, bool external) {
    GraphState* g = ctx->graph;
    KnnWork* k = ctx->knn;
    const bool binary = std::isnan(params->decay) || params->thresh == 1.0;
    const double thresh = g->p.thresh;   // (clamped by the first half)
    uint32_t n_over = 0;
    int64_t sc_total = 0;
    // (a rank of a sharded build about to take the pair-resolved tail: the same scan - its local one-sided entries go through
    //  destination bins of its own, graph_finish_pairs_shard)
    const bool shard_pairs_next = g->bw_all_valid && !external && ctx->symm_bins != 0 && shard_pairs_static(ctx, params, g->nloc);
    const bool scan_tables = k->tab_sorted || shard_pairs_next;
    if (scan_tables) {
        // (tables by sorted position: the scan of their lengths - where the fused destination count of the affinity pass puts a
        //  row's destinations; the total comes back with the count of the radius rows)
        GT_HIP(ctx, g->sC.reserve(size_t(g->nloc + 1) * sizeof(int64_t)));
        GT_TRY(exclusive_scan(ctx, reinterpret_cast<const int32_t*>(k->cand_n.as<uint32_t>()), nullptr, g->nloc, g->sC.as<int64_t>(),
                              g->scan_tmp));
    }
    {
        ReadBack rb(ctx);
        if (scan_tables) GT_HIP(ctx, rb.add(&sc_total, g->sC.as<int64_t>() + g->nloc, sizeof(int64_t)));
        GT_HIP(ctx, rb.add(&n_over, g->over_count.p, sizeof(uint32_t)));
        GT_HIP(ctx, rb.sync());
    }
    g->n_over = n_over;
    if (n_over > 0) {
        // ---- radius pass over the rows whose table is not provably complete out to their radius ----
        const int bq = gt_select_bq(ctx->DP);
        const int64_t nover_pad = ceil_div64(n_over, bq) * bq;
        int64_t cap = 1024;
        GT_HIP(ctx, g->rcounts.reserve(size_t(nover_pad) * sizeof(uint32_t)));
        for (;;) {
            if (cap > ctx->n_pad) cap = ctx->n_pad;
            {
                // the radius lists are [rows][cap] (+ as many affinities): say what a too-wide kernel asks for instead of
                // letting the allocator fail
                const double need_gb = double(nover_pad) * double(cap) * 16.0 / 1073741824.0;
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need_gb * 1073741824.0 > 0.9 * double(total_b)) {
                    char msg[320];
                    std::snprintf(msg, sizeof(msg),
                                  "radius pass: %u rows with up to %lld candidates inside their kernel radius need %.1f GB of lists "
                                  "(GPU: %.1f GB) - the kernel reaches too far for the sparse path (raise thresh or decay, lower "
                                  "bandwidth_scale, or build the exact dense graph)", n_over, (long long)cap, need_gb,
                                  double(total_b) / 1073741824.0);
                    GT_FAIL(ctx, GT_E_LIMIT, msg);
                }
            }
            GT_HIP(ctx, g->rlists.reserve(size_t(nover_pad) * size_t(cap) * sizeof(uint64_t)));
            SelectArgs sa;
            sa.dp = ctx->DP;
            sa.prec = ctx->prec;
            sa.mode = 1;
            sa.Yp = ctx->Yp.as<float>();
            sa.hneg = ctx->hneg.as<float>();
            sa.n_pad = ctx->n_pad;
            sa.Qp = g->external ? k->Qp.as<float>() : ctx->Yp.as<float>();
            sa.qrows = g->over_rows.as<int32_t>();
            sa.q0 = 0;
            sa.nq = int32_t(n_over);
            sa.lists = g->rlists.as<uint64_t>();
            sa.counts = g->rcounts.as<uint32_t>();
            sa.thr_in = g->rthr.as<float>();
            sa.cap = int32_t(cap);
            {
                StageSpan span(ctx, "radius");
                GT_TRY(gt_launch_select(ctx, sa));
            }
            GT_HIP(ctx, hipMemsetAsync(g->rmax.p, 0, sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(max_u32_kernel, dim3(64), dim3(256), 0, ctx->stream, g->rcounts.as<uint32_t>(),
                               int64_t(n_over), g->rmax.as<uint32_t>());
            uint32_t rmax = 0;
            GT_HIP(ctx, hipMemcpyAsync(&rmax, g->rmax.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (int64_t(rmax) <= cap) break;
            if (cap >= ctx->n_pad) GT_FAIL(ctx, GT_E_STATE, "radius pass: inconsistent candidate count");
            cap = std::max<int64_t>(cap * 8, int64_t(rmax) + 64);
            g->radius_retries++;
        }
        g->rcap = int32_t(cap);
        GT_HIP(ctx, g->rK.reserve(size_t(n_over) * size_t(cap) * sizeof(double)));
    }
    // ---- affinities + per-row / per-destination counts ----
    const int count_owners = (params->kernel_symm != GT_SYMM_NONE) ? 1 : 0;
    // pair-resolved symmetrisation (graph_finish_pairs): a whole single-rank build (gt_graph_build) with the '+' rule whose
    // tables carry the transposed keys and whose transpose will go through the destination bins
    g->pairs = ctx->in_graph_build && ctx->symm_pairs != 0 && ctx->symm_pair_ok != 0 && ctx->symm_bins != 0 && world == 1 &&
               !external && !binary && params->knn_max <= 0 && params->kernel_symm == GT_SYMM_ADD && params->anisotropy == 0.0 &&
               k->keyt_valid && k->ordered && k->nq == g->nloc && g->r0 == 0 && !ctx->presorted &&
               g->nloc < (int64_t(1) << 31) && (ctx->symm_bins > 0 || g->nloc >= 65536);
    if (k->tab_sorted && !g->pairs) GT_FAIL(ctx, GT_E_STATE, "graph build: tables by sorted position without the pair-resolved tail");
    g->pairs_fused = false;
    // ... or a rank of a row-sharded build that was handed the bandwidths of all rows (gt_graph_set_bandwidths): it settles its
    // mutual pairs itself - whichever rank the partner lives on - and only the one-sided entries travel (graph_finish_pairs_shard).
    // Every rank takes this branch or none: the conditions are the call's parameters, and a rank whose tables carry no
    // transposed keys (classic pass, repaired rows) forms them from the dot products.
    g->pairs_shard = g->bw_all_valid && !external && shard_pairs_static(ctx, params, g->nloc);
    g->bw_all_valid = false;   // (the bandwidths belong to one build)
    g->pairs_shard_bins = false;
    if (g->pairs_shard) {
        g->pairs = true;
        if (!k->keyt_valid) {
            GT_HIP(ctx, k->keyt_ok.reserve(size_t(g->nloc)));
            GT_HIP(ctx, hipMemsetAsync(k->keyt_ok.p, 0, size_t(g->nloc), ctx->stream));
            k->nokeyt_n = 0;
        }
        // Most one-sided entries of a rank point at rows of the SAME rank (it owns whole cells): those stay out of the exchange
        // and go through destination bins of the rank's own, as on one GPU - the affinity pass notes each one's local
        // destination row (posj, by the scan of the tables) and counts per owner only what leaves.  Needs every row to be a
        // table row (no row of the radius pass); else every one-sided entry travels and is placed by atomics.
        g->pairs_shard_bins = shard_pairs_next && n_over == 0 && sc_total > 0 && sc_total < (int64_t(1) << 31);
        if (g->pairs_shard_bins) {
            int shift = 9;
            if (ctx->symm_bin_shift > 0) shift = ctx->symm_bin_shift;
            while (ceil_div64(g->nloc, int64_t(1) << shift) > 4096) ++shift;
            // (a rank's share of the rows is small: bins of 128 rows still give the fill pass, one workgroup per bin, its thousand)
            while (ctx->symm_bin_shift <= 0 && shift > 7 && ceil_div64(g->nloc, int64_t(1) << shift) < 1024) --shift;
            g->bin_shift = shift;
            g->bin_count = int32_t(ceil_div64(g->nloc, int64_t(1) << shift));
            GT_HIP(ctx, g->bincnt.reserve(size_t(2 * g->bin_count) * sizeof(int32_t)));
            GT_HIP(ctx, hipMemsetAsync(g->bincnt.p, 0, size_t(2 * g->bin_count) * sizeof(int32_t), ctx->stream));
            GT_HIP(ctx, g->cursor.reserve(size_t(sc_total) * sizeof(uint32_t)));   // posj, by sC
            g->sc_total = sc_total;
            if (k->keyt_valid) {   // (the affinity pass will run as affinity_slots_kernel: its records)
                GT_HIP(ctx, g->rec_s.reserve(size_t(g->nloc) * sizeof(SlotRec)));
                GT_HIP(ctx, g->bwpos.reserve(size_t(ctx->n) * sizeof(BwPos)));
                hipLaunchKernelGGL(shard_slot_records_kernel, dim3((unsigned)ceil_div64(ctx->n, 256)), dim3(256), 0, ctx->stream, g->nloc,
                                   g->r0, ctx->n, g->bw_all.as<double>(), k->cand_n.as<uint32_t>(), k->keyt_ok.as<uint8_t>(),
                                   g->rowsrc.as<int32_t>(), (SlotRec*)g->rec_s.p, (BwPos*)g->bwpos.p);
                GT_HIP(ctx, hipGetLastError());
            }
        }
    }
    if (g->pairs && !g->pairs_shard) {
        // the destination bins of the tail (graph_finish_pairs): rows per bin, bins; their counters start at zero
        int shift = 9;
        if (ctx->symm_bin_shift > 0) shift = ctx->symm_bin_shift;
        while (ceil_div64(g->nloc, int64_t(1) << shift) > 4096) ++shift;
        g->bin_shift = shift;
        g->bin_count = int32_t(ceil_div64(g->nloc, int64_t(1) << shift));
        GT_HIP(ctx, g->bincnt.reserve(size_t(2 * g->bin_count) * sizeof(int32_t)));
        GT_HIP(ctx, hipMemsetAsync(g->bincnt.p, 0, size_t(2 * g->bin_count) * sizeof(int32_t), ctx->stream));
        // with the tables by sorted position and no row of the radius pass, the affinity pass counts the destinations itself
        g->pairs_fused = k->tab_sorted && n_over == 0 && sc_total > 0 && sc_total < (int64_t(1) << 31);
        if (g->pairs_fused) GT_HIP(ctx, g->cursor.reserve(size_t(sc_total) * sizeof(uint32_t)));   // posj, by sC
        g->sc_total = sc_total;
    }
    if (ctx->dbg_select & 2048)
        std::fprintf(stderr, "[gt] pairs %d: in_build %d opt %d ok %d bins %d world %d ext %d bin %d kmax %lld symm %d aniso %g metric %d keyt %d ordered %d nq %lld nloc %lld r0 %lld\n",
                     int(g->pairs), ctx->in_graph_build, ctx->symm_pairs, ctx->symm_pair_ok, ctx->symm_bins, world, int(external), int(binary),
                     (long long)params->knn_max, params->kernel_symm, params->anisotropy, ctx->metric, int(k->keyt_valid), int(k->ordered),
                     (long long)k->nq, (long long)g->nloc, (long long)g->r0);
    {
        StageSpan span(ctx, "affinity");
        if (ctx->dtype == GT_F32)
            launch_affinity<float>(ctx, g, k, binary ? 1 : 0, params->decay, thresh, count_owners);
        else
            launch_affinity<double>(ctx, g, k, binary ? 1 : 0, params->decay, thresh, count_owners);
        GT_HIP(ctx, hipGetLastError());
    }
    g->send_counts_host.assign(world, 0);
    if (count_owners && g->pairs_fused) {
        // the pair-resolved tail that follows sizes its buffers by this: the entries of all tables bound the kept ones, which the
        // tail counts itself (the scan of the counts by slot) - no gather / scan / scatter / read-back of the owners' counts here
        g->send_counts_host[0] = g->sc_total;
    } else if (count_owners) {
        // exclusive scan of the owner-major counts: slot of every (row, owner) pair inside the bucketed send buffer
        GT_HIP(ctx, g->ownerpos.reserve(size_t(int64_t(world) * g->nloc + 1) * sizeof(int64_t)));
        const bool sorted_buf = world == 1 && !external && k->ordered && k->nq == g->nloc && g->r0 == 0;
        if (sorted_buf) {
            // buffer in cell-sorted row order (see gather_counts_kernel)
            const int32_t* perm = k->qorder.as<int32_t>();
            GT_HIP(ctx, g->cnt_sorted.reserve(size_t(g->nloc) * sizeof(int32_t)));
            GT_HIP(ctx, g->pos_sorted.reserve(size_t(g->nloc + 1) * sizeof(int64_t)));
            hipLaunchKernelGGL(gather_counts_kernel, dim3((unsigned)ceil_div64(g->nloc, 256)), dim3(256), 0, ctx->stream,
                               g->ownercnt.as<int32_t>(), perm, g->nloc, g->cnt_sorted.as<int32_t>());
            GT_HIP(ctx, hipGetLastError());
            GT_TRY(exclusive_scan(ctx, g->cnt_sorted.as<int32_t>(), nullptr, g->nloc, g->pos_sorted.as<int64_t>(), g->scan_tmp));
            hipLaunchKernelGGL(scatter_positions_kernel, dim3((unsigned)ceil_div64(g->nloc + 1, 256)), dim3(256), 0, ctx->stream,
                               g->pos_sorted.as<int64_t>(), perm, g->nloc, g->ownerpos.as<int64_t>());
            GT_HIP(ctx, hipGetLastError());
        } else {
            GT_TRY(exclusive_scan(ctx, g->ownercnt.as<int32_t>(), nullptr, int64_t(world) * g->nloc, g->ownerpos.as<int64_t>(),
                                  g->scan_tmp));
        }
        std::vector<int64_t> edge(world + 1, 0);
        // (sorted buffer, one rank: row 0 does not sit at slot 0 - only the total, behind the last row, is an edge)
        if (world > 1) {
            // the world + 1 bucket edges in ONE read-back (a copy of 8 bytes each cost 20 us: 0.17 ms per rank at world 8)
            GT_HIP(ctx, g->edges.reserve(size_t(world + 1) * sizeof(int64_t)));
            hipLaunchKernelGGL(gather_strided_i64_kernel, dim3(1), dim3(64), 0, ctx->stream, g->ownerpos.as<int64_t>(), g->nloc,
                               world + 1, g->edges.as<int64_t>());
            GT_HIP(ctx, hipGetLastError());
            GT_HIP(ctx, hipMemcpyAsync(edge.data(), g->edges.p, size_t(world + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        } else {
            for (int r = sorted_buf ? world : 0; r <= world; ++r)
                GT_HIP(ctx, hipMemcpyAsync(&edge[r], g->ownerpos.as<int64_t>() + int64_t(r) * g->nloc, sizeof(int64_t),
                                           hipMemcpyDeviceToHost, ctx->stream));
        }
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int r = 0; r < world; ++r) g->send_counts_host[r] = edge[r + 1] - edge[r];
    }
    for (int r = 0; r < world; ++r) send_counts[r] = g->send_counts_host[r];
    g->begun = true;
    return GT_OK;
}

static int graph_begin_impl(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                            const int64_t* row_splits, int64_t* send_counts, bool external, int64_t m_ext) {
    bool done = false;
    if (ctx && ctx->graph) ctx->graph->half_begun = false, ctx->graph->bw_all_valid = false;   // (a build from its start)
    GT_TRY(graph_begin_a(ctx, params, world, rank, row_splits, send_counts, external, m_ext, false, &done));
    if (done) return GT_OK;
    return graph_begin_b(ctx, params, world, send_counts, external);
}

static bool same_build(const GraphState* g, const gt_knn_params* p, int32_t world, int32_t rank, const int64_t* row_splits) {
    if (g->world != world || g->rank != rank) return false;
    for (int r = 0; r <= world; ++r)
        if (g->splits[r] != row_splits[r]) return false;
    const gt_knn_params& q = g->half_params;
    const bool decay_same = (std::isnan(p->decay) && std::isnan(q.decay)) || p->decay == q.decay;
    return p->knn == q.knn && decay_same && p->thresh == q.thresh && p->knn_max == q.knn_max && p->kernel_symm == q.kernel_symm &&
           p->theta == q.theta && p->anisotropy == q.anisotropy && p->bandwidth_len == q.bandwidth_len &&
           p->bandwidth_scale == q.bandwidth_scale && p->bandwidth == q.bandwidth;
}

extern "C" int gt_graph_begin(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                              const int64_t* row_splits, int64_t* send_counts) {
    if (!ctx) return GT_E_ARG;
    if (ctx->graph && ctx->graph->half_begun) {
        // the second half of a build whose first half gt_graph_bandwidth_local ran (its stages belong to this build)
        GraphState* g = ctx->graph;
        g->half_begun = false;
        if (!params || !row_splits || !send_counts) return GT_E_ARG;
        GT_HIP(ctx, hipSetDevice(ctx->device));
        if (!same_build(g, params, world, rank, row_splits))
            GT_FAIL(ctx, GT_E_STATE, "gt_graph_begin: not the build gt_graph_bandwidth_local started (parameters, world, rank or row_splits differ)");
        const int rc = graph_begin_b(ctx, params, world, send_counts, false);
        ctx->stage_totals_valid = 0;
        return rc;
    }
    // (the stages of a sharded symmetric pass that is about to be consumed belong to this build)
    // (... and so do those of a first attempt that the pair-resolved tail refuted: gt_graph_build keeps them, the stage times
    //  then say what the build cost)
    if (!(ctx->knn && (ctx->knn->sh_stage == 5 || ctx->knn->sh_stage == 6)) && !ctx->keep_stages) ctx->reset_stages();
    const int rc = graph_begin_impl(ctx, params, world, rank, row_splits, send_counts, false, 0);
    ctx->stage_totals_valid = 0;   // (the totals belong to one build)
    return rc;
}

// Row-sharded build, pair-resolved tail (include/graphtools_amd.h): the first half of gt_graph_begin for this rank's rows - candidate
// tables, re-rank (with the transposed keys), bandwidths - and the rank's bandwidths into bw_local_dev (device, float64
// [row_splits[rank + 1] - row_splits[rank]], on the library's stream).  applies = 0: nothing was done (the build's parameters
// are not the pair-resolved tail's: every rank gets the same answer) - gt_graph_begin runs the whole build as it always did.
extern "C" int gt_graph_bandwidth_local(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                                        const int64_t* row_splits, double* bw_local_dev, int32_t* applies) {
    if (!ctx || !params || !row_splits || !applies) return GT_E_ARG;
    *applies = 0;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world) GT_FAIL(ctx, GT_E_ARG, "gt_graph_bandwidth_local: bad world/rank");
    if (ctx->graph) ctx->graph->half_begun = false, ctx->graph->bw_all_valid = false;
    const int64_t nloc = row_splits[rank + 1] - row_splits[rank];
    if (nloc <= 0 || params->knn < 1 || !shard_pairs_static(ctx, params, nloc)) return GT_OK;
    if (!bw_local_dev) GT_FAIL(ctx, GT_E_ARG, "gt_graph_bandwidth_local: bw_local_dev is NULL");
    if (!(ctx->knn && (ctx->knn->sh_stage == 5 || ctx->knn->sh_stage == 6)) && !ctx->keep_stages) ctx->reset_stages();
    std::vector<int64_t> sendc(size_t(world), 0);
    bool done = false;
    GT_TRY(graph_begin_a(ctx, params, world, rank, row_splits, sendc.data(), false, 0, true, &done));
    GraphState* g = ctx->graph;
    GT_HIP(ctx, hipMemcpyAsync(bw_local_dev, g->bw.p, size_t(g->nloc) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    g->half_begun = true;
    g->half_params = *params;
    *applies = 1;
    return GT_OK;
}

// ... and the bandwidths of ALL rows (device, float64 [n], in the order of the context's rows: the ranks' slices in rank order),
// copied on the library's stream: the caller keeps bw_all_dev alive until gt_graph_begin has returned.
extern "C" int gt_graph_set_bandwidths(gt_ctx* ctx, const double* bw_all_dev) {
    if (!ctx || !bw_all_dev) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->half_begun) GT_FAIL(ctx, GT_E_STATE, "gt_graph_set_bandwidths: call gt_graph_bandwidth_local first");
    GT_HIP(ctx, g->bw_all.reserve(size_t(ctx->n) * sizeof(double)));
    GT_HIP(ctx, hipMemcpyAsync(g->bw_all.p, bw_all_dev, size_t(ctx->n) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    g->bw_all_valid = true;
    return GT_OK;
}

// Row-sharded builds with knn_max (graphs.py:916-976: the search-expansion loop escalates while more than a tenth of ALL rows
// still have their whole table inside their radius): this rank's counts for the loop's steps.  The caller sums them over the
// ranks and hands the sums to gt_graph_set_stage_totals before gt_graph_begin; n_counts = 0: nothing to exchange (no knn_max,
// one rank, or a table depth the kernels cannot hold).  Runs the candidate search of the owned rows (gt_graph_begin runs it
// again: the price of following the reference's branches in this rare configuration).
extern "C" int gt_graph_stage_counts(gt_ctx* ctx, const gt_knn_params* params, int32_t world, int32_t rank,
                                     const int64_t* row_splits, int64_t* counts4, int32_t* n_counts) {
    if (!ctx || !params || !counts4 || !n_counts) return GT_E_ARG;
    *n_counts = 0;
    for (int t = 0; t < 4; ++t) counts4[t] = 0;
    if (world <= 1 || params->knn_max <= 0 || std::isnan(params->decay)) return GT_OK;
    std::vector<int64_t> sendc(size_t(world), 0);
    ctx->reset_stages();
    ctx->stage_counts_only = 1;
    ctx->stage_n = 0;
    const int rc = graph_begin_impl(ctx, params, world, rank, row_splits, sendc.data(), false, 0);
    ctx->stage_counts_only = 0;
    if (ctx->graph) ctx->graph->begun = false;
    if (rc != GT_OK) return rc;
    *n_counts = ctx->stage_n;
    for (int t = 0; t < ctx->stage_n && t < 4; ++t) counts4[t] = ctx->stage_local[t];
    return GT_OK;
}

extern "C" int gt_graph_set_stage_totals(gt_ctx* ctx, const int64_t* totals4, int32_t n_counts) {
    if (!ctx || !totals4 || n_counts < 0 || n_counts > 4) return GT_E_ARG;
    for (int t = 0; t < 4; ++t) ctx->stage_totals[t] = t < n_counts ? totals4[t] : 0;
    ctx->stage_totals_valid = n_counts > 0 ? 1 : 0;
    return GT_OK;
}

extern "C" int gt_graph_extend(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device, const gt_knn_params* params,
                               int64_t* out_nnz, uint32_t* flags) {
    if (!ctx || !params) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    GT_TRY(gt_prepare_queries(ctx, Y, m, y_on_device));
    gt_knn_params p = *params;
    p.kernel_symm = GT_SYMM_NONE;   // K_yx is rectangular: no symmetrisation, no anisotropy (graphs.py:819-982)
    p.anisotropy = 0.0;
    int64_t splits[2] = {0, m};
    int64_t sendc[1] = {0};
    GT_TRY(graph_begin_impl(ctx, &p, 1, 0, splits, sendc, true, m));
    uint32_t fl = 0;
    GT_TRY(gt_graph_finish(ctx, nullptr, 0, out_nnz, &fl));
    if (flags) *flags = fl & ~uint32_t(GT_FLAG_ZERO_DIAGONAL | GT_FLAG_DUPLICATES);
    return GT_OK;
}

extern "C" int gt_graph_emit(gt_ctx* ctx, void* send_buf_dev) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->begun) GT_FAIL(ctx, GT_E_STATE, "gt_graph_emit: call gt_graph_begin first");
    int64_t total = 0;
    for (int r = 0; r < g->world; ++r) total += g->send_counts_host[r];
    if (total == 0) return GT_OK;
    if (!send_buf_dev) GT_FAIL(ctx, GT_E_ARG, "gt_graph_emit: send buffer is NULL");
    KnnWork* k = ctx->knn;
    StageSpan span(ctx, "symmetrize");
    hipLaunchKernelGGL(emit_triplets_kernel, dim3((unsigned)ceil_div64(g->nloc, 4)), dim3(256), 0, ctx->stream, g->nloc,
                       g->r0, k->MP, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(),
                       g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(), make_splits(g),
                       g->ownerpos.as<int64_t>(), g->tablen.as<int32_t>(), (Triplet*)send_buf_dev,
                       g->pairs_shard_bins ? g->rank : -1);
    GT_HIP(ctx, hipGetLastError());
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

static int finish_normalize(gt_ctx* ctx, GraphState* g, const double* degree_all_dev);

// bins: single rank, every row local, cell-sorted order available - the transpose is built from the tables through
// destination bins (see bin_count_kernel) and recv_buf_dev is not read
static int graph_finish_impl(gt_ctx* ctx, const void* recv_buf_dev, int64_t n_recv, bool bins, int64_t* out_nnz, uint32_t* flags) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->begun) GT_FAIL(ctx, GT_E_STATE, "gt_graph_finish: call gt_graph_begin first");
    if (n_recv < 0 || (n_recv > 0 && !recv_buf_dev)) GT_FAIL(ctx, GT_E_ARG, "gt_graph_finish: bad receive buffer");
    KnnWork* k = ctx->knn;
    const Triplet* recv = (const Triplet*)recv_buf_dev;
    const int64_t nloc = g->nloc;
    // renumbered points (gt_points_cell_sort): rows and triplets are the context's, the CSR gets the caller's columns
    const int32_t* relabel = (ctx->presorted && !g->external && !bins) ? ctx->vperm.as<int32_t>() : nullptr;
    g->relabelled = relabel != nullptr;
    {
        StageSpan span(ctx, "symmetrize");
        HostTrace tr_all(ctx, "finish: symmetrize");
        GT_HIP(ctx, g->off.reserve(size_t(nloc + 1) * sizeof(int64_t)));
        int rc = GT_OK;
        int64_t total_u = 0;
        const int32_t* perm = bins ? k->qorder.as<int32_t>() : nullptr;
        int shift = 9, nbins = 0;
        if (bins) {
            // rows per bin: 512, more only where the two histograms of an emitting workgroup would outgrow 32 KB of LDS
            if (ctx->symm_bin_shift > 0) shift = ctx->symm_bin_shift;
            while (ceil_div64(nloc, int64_t(1) << shift) > 4096) ++shift;
            nbins = int(ceil_div64(nloc, int64_t(1) << shift));
            StageSpan span_bins(ctx, "symm_bins");   // (nested in "symmetrize": its share, and the sign that this path ran)
            GT_HIP(ctx, k->sh_invperm.reserve(size_t(nloc) * sizeof(int32_t)));
            GT_TRY(gt_sym_invperm(ctx, perm, k->sh_invperm.as<int32_t>()));
            // own entries in sorted order and their scan; triplets per bin and their scan
            GT_HIP(ctx, g->cnt_sorted.reserve(size_t(nloc) * sizeof(int32_t)));
            GT_HIP(ctx, g->pos_sorted.reserve(size_t(nloc + 1) * sizeof(int64_t)));
            GT_HIP(ctx, g->bincnt.reserve(size_t(2 * nbins) * sizeof(int32_t)));
            GT_HIP(ctx, g->binoff.reserve(size_t(nbins + 1) * sizeof(int64_t)));
            GT_HIP(ctx, hipMemsetAsync(g->bincnt.p, 0, size_t(2 * nbins) * sizeof(int32_t), ctx->stream));
            hipLaunchKernelGGL(gather_counts_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream,
                               g->lenN.as<int32_t>(), perm, nloc, g->cnt_sorted.as<int32_t>());
            GT_TRY(exclusive_scan(ctx, g->cnt_sorted.as<int32_t>(), nullptr, nloc, g->pos_sorted.as<int64_t>(), g->scan_tmp));
            {
                // every kept entry is sent once: the scan's total is the number of triplets
                hipError_t e = hipMemcpyAsync(&n_recv, g->pos_sorted.as<int64_t>() + nloc, sizeof(int64_t), hipMemcpyDeviceToHost,
                                              ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) {
                    ctx->set_error(std::string("scan: ") + hipGetErrorString(e));
                    return GT_E_HIP;
                }
            }
            GT_HIP(ctx, g->cursor.reserve(size_t(std::max<int64_t>(n_recv, 1)) * sizeof(uint32_t)));   // posj
            hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(nloc, 64), 2048)), dim3(256),
                               size_t(nbins) * sizeof(int32_t), ctx->stream, nloc, k->MP, k->cand_d2.as<double>(),
                               k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(), g->rlists.as<uint64_t>(),
                               g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(), g->tablen.as<int32_t>(), perm,
                               k->sh_invperm.as<int32_t>(), shift, nbins, g->pos_sorted.as<int64_t>(), g->cursor.as<uint32_t>(),
                               g->bincnt.as<int32_t>(), 0);
            GT_HIP(ctx, hipGetLastError());
            GT_TRY(exclusive_scan(ctx, g->bincnt.as<int32_t>(), nullptr, nbins, g->binoff.as<int64_t>(), g->scan_tmp));
            total_u = 2 * n_recv;   // every kept entry once in its own row, once in its column's
        } else {
            GT_HIP(ctx, hipMemsetAsync(g->lenT.p, 0, size_t(nloc) * sizeof(int32_t), ctx->stream));
            GT_HIP(ctx, g->cursor.reserve(size_t(std::max<int64_t>(n_recv, 1)) * sizeof(int32_t)));   // slot of every received triplet
            if (n_recv > 0) {
                int64_t blocks = std::min<int64_t>(ceil_div64(n_recv, 256), 16384);
                hipLaunchKernelGGL(count_recv_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, recv, n_recv, g->r0,
                                   g->lenT.as<int32_t>(), g->cursor.as<int32_t>());
            }
            rc = exclusive_scan(ctx, g->lenN.as<int32_t>(), g->lenT.as<int32_t>(), nloc, g->off.as<int64_t>(), g->scan_tmp);
            if (rc == GT_OK) {
                hipError_t e = hipMemcpyAsync(&total_u, g->off.as<int64_t>() + nloc, sizeof(int64_t), hipMemcpyDeviceToHost,
                                              ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) {
                    ctx->set_error(std::string("scan: ") + hipGetErrorString(e));
                    rc = GT_E_HIP;
                }
            }
            GT_TRY(rc);
        }
        g->nnz0 = total_u - n_recv;
        HostTrace tr_u(ctx, "finish: fill + merge");
        // union rows (bin path: their received halves only - the own halves are read from the tables)
        GT_HIP(ctx, g->Ukey.reserve(size_t(bins ? std::max<int64_t>(n_recv, 1) : total_u) * sizeof(UEntry)));
        GT_HIP(ctx, g->Vkey.reserve(size_t(total_u) * sizeof(uint32_t)));
        GT_HIP(ctx, g->Vval.reserve(size_t(total_u) * sizeof(double)));
        GT_HIP(ctx, g->outlen.reserve(size_t(nloc) * sizeof(int32_t)));
        GT_HIP(ctx, g->bigrows.reserve(size_t(nloc) * sizeof(int32_t)));
        GT_HIP(ctx, g->hugerows.reserve(size_t(nloc) * sizeof(int32_t)));
        GT_HIP(ctx, g->bigcount.reserve(2 * sizeof(uint32_t)));   // [0] rows > kBigRow, [1] rows > kHugeRow
        GT_HIP(ctx, hipMemsetAsync(g->bigcount.p, 0, 2 * sizeof(uint32_t), ctx->stream));
        if (bins) {
            GT_HIP(ctx, g->selfbuf.reserve(size_t(std::max<int64_t>(n_recv, 1)) * sizeof(Triplet)));
            StageSpan span_bins(ctx, "symm_bins");
            const size_t emit_lds = size_t(2 * nbins) * sizeof(int32_t);
            hipLaunchKernelGGL(bin_emit_kernel, dim3((unsigned)ceil_div64(nloc, kEmitRows)), dim3(256), emit_lds, ctx->stream,
                               nloc, k->MP, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(),
                               g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(),
                               g->tablen.as<int32_t>(), perm, g->pos_sorted.as<int64_t>(), g->cursor.as<uint32_t>(), shift, nbins,
                               g->binoff.as<int64_t>(), g->bincnt.as<int32_t>() + nbins, (Triplet*)g->selfbuf.p, 0);
            GT_HIP(ctx, hipGetLastError());
#define GT_BIN_FILL(NT_)                                                                                                     \
    hipLaunchKernelGGL(bin_fill_kernel<NT_>, dim3((unsigned)nbins), dim3(NT_), size_t(2) * (size_t(1) << shift) * sizeof(int32_t), \
                       ctx->stream, nloc, k->MP, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(),                          \
                       g->rowsrc.as<int32_t>(), g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap,               \
                       g->rK.as<double>(), g->tablen.as<int32_t>(), perm, shift, nbins, g->binoff.as<int64_t>(),            \
                       (const Triplet*)g->selfbuf.p, g->cnt_sorted.as<int32_t>(), g->pos_sorted.as<int64_t>(),              \
                       g->off.as<int64_t>(), g->Ukey.as<UEntry>(), (uint32_t*)nullptr, (double*)nullptr,                        \
                       PairsOut{nullptr, nullptr, nullptr, nullptr})
            GT_BIN_FILL(256);
#undef GT_BIN_FILL
            GT_HIP(ctx, hipGetLastError());
        } else {
            hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc, k->MP,
                               k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(),
                               g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(),
                               g->off.as<int64_t>(), g->tablen.as<int32_t>(), g->Ukey.as<UEntry>(), relabel);
            if (n_recv > 0) {
                int64_t blocks = std::min<int64_t>(ceil_div64(n_recv, 256), 16384);
                hipLaunchKernelGGL(fill_recv_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, recv, n_recv, g->r0,
                                   g->off.as<int64_t>(), g->lenN.as<int32_t>(), g->cursor.as<int32_t>(), g->Ukey.as<UEntry>(),
                                   relabel);
            }
        }
        UnionSrc us;
        us.U = g->Ukey.as<UEntry>();
        us.sN = bins ? g->pos_sorted.as<int64_t>() : nullptr;
        us.lenNs = g->cnt_sorted.as<int32_t>();
        us.perm = perm;
        us.rowsrc = g->rowsrc.as<int32_t>();
        us.cand_k = k->cand_d2.as<double>();
        us.cand_j = k->cand_j.as<uint32_t>();
        us.MP = k->MP;
        us.rlists = g->rlists.as<uint64_t>();
        us.rK = g->rK.as<double>();
        us.rcap = g->rcap;
        {
            StageSpan span_m(ctx, "symm_merge");   // (nested in "symmetrize")
            hipLaunchKernelGGL(sort_merge_kernel, dim3((unsigned)ceil_div64(nloc, 1)), dim3(64), 0, ctx->stream, nloc,
                               g->off.as<int64_t>(), us, g->p.kernel_symm, g->p.theta,
                               g->Vkey.as<uint32_t>(), g->Vval.as<double>(), g->outlen.as<int32_t>(), g->bigrows.as<int32_t>(),
                               g->bigcount.as<uint32_t>(), (ctx->symm_key32 != 0 && sort_key_fits_u32(g->n_total, 8)) ? 1 : 0);
            hipLaunchKernelGGL(sort_merge_long_kernel<16>, dim3(4096), dim3(64), 0, ctx->stream, g->off.as<int64_t>(),
                               us, g->p.kernel_symm, g->p.theta, g->Vkey.as<uint32_t>(),
                               g->Vval.as<double>(), g->outlen.as<int32_t>(), g->bigrows.as<int32_t>(), g->bigcount.as<uint32_t>(),
                               g->hugerows.as<int32_t>(), g->bigcount.as<uint32_t>() + 1);
            hipLaunchKernelGGL(sort_merge_long_kernel<32>, dim3(2048), dim3(64), 0, ctx->stream, g->off.as<int64_t>(),
                               us, g->p.kernel_symm, g->p.theta, g->Vkey.as<uint32_t>(),
                               g->Vval.as<double>(), g->outlen.as<int32_t>(), g->bigrows.as<int32_t>(), g->bigcount.as<uint32_t>(),
                               g->hugerows.as<int32_t>(), g->bigcount.as<uint32_t>() + 1);
        }
        GT_HIP(ctx, hipGetLastError());
        uint32_t nbig = 0;   // rows beyond the register sorts
        GT_HIP(ctx, hipMemcpyAsync(&nbig, g->bigcount.as<uint32_t>() + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HostTrace tr_b(ctx, "finish: long rows + compact");
        if (nbig > 0) {
            // rows longer than kHugeRow: global-memory bitonic sort, one workgroup per row
            std::vector<int32_t> rows(nbig);
            std::vector<int64_t> offh(nloc + 1);
            {
                // copies on the library's stream (the null stream would synchronise with every other stream of the process)
                GT_HIP(ctx, hipMemcpyAsync(rows.data(), g->hugerows.p, size_t(nbig) * sizeof(int32_t), hipMemcpyDeviceToHost,
                                           ctx->stream));
                GT_HIP(ctx, hipMemcpyAsync(offh.data(), g->off.p, size_t(nloc + 1) * sizeof(int64_t), hipMemcpyDeviceToHost,
                                           ctx->stream));
                GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
            std::vector<int64_t> soff(nbig + 1, 0);
            for (uint32_t b = 0; b < nbig; ++b) {
                const int64_t L = offh[rows[b] + 1] - offh[rows[b]];
                int64_t P = 1;
                while (P < L) P <<= 1;
                soff[b + 1] = soff[b] + P;
            }
            if (std::getenv("GT_TRACE")) {
                int64_t lmax = 0;
                for (uint32_t b = 0; b < nbig; ++b) lmax = std::max<int64_t>(lmax, offh[rows[b] + 1] - offh[rows[b]]);
                std::fprintf(stderr, "[gt_trace] long rows: %u, longest %lld entries (received %lld triplets)\n", nbig,
                             (long long)lmax, (long long)n_recv);
            }
            DevBuf& soff_dev = g->bigsoff;   // persistent: no hipMalloc / hipFree per call
            GT_HIP(ctx, soff_dev.reserve(size_t(nbig + 1) * sizeof(int64_t)));
            GT_HIP(ctx, hipMemcpyAsync(soff_dev.p, soff.data(), size_t(nbig + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                                       ctx->stream));
            GT_HIP(ctx, g->bigscratch_k.reserve(size_t(soff[nbig]) * sizeof(uint32_t)));
            GT_HIP(ctx, g->bigscratch_v.reserve(size_t(soff[nbig]) * sizeof(double)));
            hipLaunchKernelGGL(big_sort_kernel, dim3(nbig), dim3(1024), 0, ctx->stream, g->hugerows.as<int32_t>(),
                               g->off.as<int64_t>(), us, soff_dev.as<int64_t>(),
                               g->bigscratch_k.as<uint32_t>(), g->bigscratch_v.as<double>());
            hipLaunchKernelGGL(big_merge_kernel, dim3(nbig), dim3(64), 0, ctx->stream, g->hugerows.as<int32_t>(),
                               g->off.as<int64_t>(), soff_dev.as<int64_t>(), g->bigscratch_k.as<uint32_t>(),
                               g->bigscratch_v.as<double>(), g->p.kernel_symm, g->p.theta, g->Vkey.as<uint32_t>(),
                               g->Vval.as<double>(), g->outlen.as<int32_t>());
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) {
                ctx->set_error(std::string("big-row sort: ") + hipGetErrorString(e));
                return GT_E_HIP;
            }
        }
        // ---- compact to CSR ----
        GT_HIP(ctx, g->indptr.reserve(size_t(nloc + 1) * sizeof(int64_t)));
        const int32_t* outlen_rows = g->outlen.as<int32_t>();
        if (bins) {
            // merged lengths back in row order (lenT is free on this path)
            hipLaunchKernelGGL(scatter_i32_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream,
                               g->outlen.as<int32_t>(), perm, nloc, g->lenT.as<int32_t>());
            GT_HIP(ctx, hipGetLastError());
            outlen_rows = g->lenT.as<int32_t>();
        }
        rc = exclusive_scan(ctx, outlen_rows, nullptr, nloc, g->indptr.as<int64_t>(), g->scan_tmp);
        int64_t nnz = 0;
        if (rc == GT_OK) {
            hipError_t e = hipMemcpyAsync(&nnz, g->indptr.as<int64_t>() + nloc, sizeof(int64_t), hipMemcpyDeviceToHost,
                                          ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) {
                ctx->set_error(std::string("scan: ") + hipGetErrorString(e));
                rc = GT_E_HIP;
            }
        }
        GT_TRY(rc);
        g->nnz = nnz;
        GT_HIP(ctx, g->indices.reserve(size_t(nnz) * sizeof(int32_t)));
        GT_HIP(ctx, g->Kdata.reserve(size_t(nnz) * sizeof(double)));
        GT_HIP(ctx, g->Pdata.reserve(size_t(nnz) * sizeof(double)));
        GT_HIP(ctx, g->degree.reserve(size_t(nloc) * sizeof(double)));
        {
            StageSpan span_c(ctx, "symm_compact");   // (nested in "symmetrize")
            hipLaunchKernelGGL(compact_kernel, dim3((unsigned)ceil_div64(nloc, 1)), dim3(64), 0, ctx->stream, nloc, g->r0,
                               g->off.as<int64_t>(), g->outlen.as<int32_t>(), g->indptr.as<int64_t>(), g->Vkey.as<uint32_t>(),
                               g->Vval.as<double>(), g->indices.as<int32_t>(), g->Kdata.as<double>(), g->degree.as<double>(),
                               g->flags.as<uint32_t>(), perm, g->p.anisotropy == 0.0 ? g->Pdata.as<double>() : nullptr, relabel);
        }
        GT_HIP(ctx, hipGetLastError());
    }
    g->finished = true;
    if (g->p.anisotropy != 0.0) {
        // sharded build: the caller must all-gather the degrees and call gt_graph_anisotropy (K and every flag of the
        // build are final here, only the anisotropic rescaling and P are still to come)
        if (g->world == 1) GT_TRY(finish_normalize(ctx, g, g->degree.as<double>()));
    }   // (no anisotropy: compact_kernel has written P next to K)
    uint32_t fl = 0, kfl = 0;
    GT_HIP(ctx, hipMemcpyAsync(&fl, g->flags.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipMemcpyAsync(&kfl, k->gflags.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    fl |= kfl;
    if (k->n_fallback > 0) fl |= GT_FLAG_FALLBACK_ROWS;
    if (g->n_over > 0) fl |= GT_FLAG_RADIUS_ROWS;
    if (out_nnz) *out_nnz = g->nnz;
    if (flags) *flags = fl;
    return GT_OK;
}

// ---- pair-resolved tail ('+' rule, single rank) ---------------------------------------------------------------------
// affinity_kernel has settled every mutual pair in its own row (negative = final value), so only the one-sided entries
// are transposed: 44 M instead of 72 M triplets at N = 1e6, no pair ever meets its partner in a union row - a row's
// final length is its own kept entries plus what it receives, known once the bins are filled - and the merge sorts and
// writes K and P straight into the CSR (merge_final_kernel).
__global__ __launch_bounds__(256) void pairs_len_kernel(const int64_t nloc, const int32_t* __restrict__ pos,
                                                        const int64_t* __restrict__ off, int32_t* __restrict__ outlen,
                                                        int32_t* __restrict__ biglist, uint32_t* __restrict__ bigcount,
                                                        uint32_t* __restrict__ fflags, int32_t* __restrict__ midlist) {
    // midlist (optional): the rows of 129 ... kBigRow entries, counted in bigcount[1] (one atomic per wave)
    // bigcount: [0] rows for merge_long_final_kernel (listed from biglist[0] upwards), [2] = fflags, [4] rows beyond the
    // register sorts (listed from biglist[nloc - 1] downwards), [6..7] their entries in all (64 bit)
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    int64_t L = 0;
    if (i < nloc) {
        const int64_t p = pos[i];
        L = off[p + 1] - off[p];
        outlen[i] = int32_t(L);
    }
    if (midlist) {   // (uniform; one atomic per workgroup: sixteen thousand on one address took 0.18 ms)
        __shared__ uint32_t wcnt[4], wbase;
        const bool mid = L > 128 && L <= kBigRow;
        const unsigned long long mm = __ballot(mid);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) wcnt[w] = uint32_t(__popcll(mm));
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            wbase = tot ? atomicAdd(bigcount + 1, tot) : 0u;
        }
        __syncthreads();
        uint32_t base = wbase;
        for (int q = 0; q < w; ++q) base += wcnt[q];
        if (mid) midlist[base + uint32_t(__popcll(mm & ((1ull << lane) - 1ull)))] = int32_t(i);
    }
    if (i >= nloc) return;
    if (L > kBigRow) {
        if (L > kPairHugeRow) {
            atomicOr(fflags, kFusedHugeRow);
            biglist[nloc - 1 - int64_t(atomicAdd(bigcount + 4, 1u))] = int32_t(i);
            atomicAdd(reinterpret_cast<unsigned long long*>(bigcount + 6), (unsigned long long)L);
        } else {
            biglist[atomicAdd(bigcount, 1u)] = int32_t(i);
        }
    }
}

// the merge launches of the pair-resolved tails (graph_finish_pairs, graph_finish_pairs_shard): union rows -> K, P, degrees in the
// CSR.  slots: every row is a table row and the rows are walked by slot (merge_pairs_slots_kernel; perm_slots: slot -> row,
// lenN_slots: own entries by slot; the rows of 129 ... kBigRow entries are listed in g->midrows, n_mid of them)
static int launch_pair_merges(gt_ctx* ctx, GraphState* g, const FusedSrc& fs, const int64_t nloc, const bool fused, const uint32_t n_mid,
                              const uint32_t n_huge, const unsigned long long huge_total, const int32_t* perm_slots,
                              const int32_t* lenN_slots) {
    {
        StageSpan span_m(ctx, "symm_merge");
        // the few long rows (one wave each, 270 registers: a thousand waves for half a millisecond) run on a side stream
        // next to the merge of the others - the two launches write different rows; the main stream was drained by the
        // read-back above, and takes the side stream's completion back before anything looks at the result
        if (!ctx->side_stream) {
            // (highest priority: the thousand long-row waves are the shorter job and must not queue behind the million short rows)
            int prio_lo = 0, prio_hi = 0;
            ctx->side_stream = gt_handle_take_stream(ctx->device, true);   // (parked by a closed context, gt_devpool.cpp)
            if (!ctx->side_stream) {
                GT_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
                GT_HIP(ctx, hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, prio_hi));
            }
            GT_HIP(ctx, hipEventCreateWithFlags(&ctx->side_event, hipEventDisableTiming));
        }
        hipLaunchKernelGGL(merge_long_final_kernel<16>, dim3(2048), dim3(64), 0, ctx->side_stream, fs, g->indptr.as<int64_t>(),
                           g->indices.as<int32_t>(), g->Kdata.as<double>(), g->Pdata.as<double>(), g->degree.as<double>(),
                           g->flags.as<uint32_t>(), g->bigrows.as<int32_t>(), g->bigcount.as<uint32_t>());
        GT_HIP(ctx, hipGetLastError());
        if (n_huge > 0) {
            // the rows beyond the register sorts: gather -> segmented sort by column -> their place in the CSR, on the side
            // stream as well (they write their own rows of the CSR: nothing the merge of the others reads or writes)
            StageSpan span_h(ctx, "symm_huge");
            const size_t H = size_t(huge_total);
            GT_HIP(ctx, g->bigscratch_k.reserve(2 * H * sizeof(uint32_t)));
            GT_HIP(ctx, g->bigscratch_v.reserve(2 * H * sizeof(double)));
            GT_HIP(ctx, g->bigsoff.reserve((2 * size_t(n_huge) + 2) * sizeof(uint32_t) + sizeof(unsigned long long)));
            uint32_t* kin = g->bigscratch_k.as<uint32_t>();
            uint32_t* kout = kin + H;
            double* vin = g->bigscratch_v.as<double>();
            double* vout = vin + H;
            unsigned long long* cursor = g->bigsoff.as<unsigned long long>();
            uint32_t* seg_begin = reinterpret_cast<uint32_t*>(cursor + 1);
            uint32_t* seg_end = seg_begin + n_huge;
            GT_HIP(ctx, hipMemsetAsync(cursor, 0, sizeof(unsigned long long), ctx->side_stream));
            const int32_t* hugelist = g->bigrows.as<int32_t>() + (nloc - 1);
            hipLaunchKernelGGL(huge_gather_kernel, dim3(n_huge), dim3(64), 0, ctx->side_stream, fs, hugelist, n_huge, cursor, seg_begin,
                               seg_end, kin, vin);
            GT_HIP(ctx, hipGetLastError());
            int bits = 1;
            while (bits < 32 && (int64_t(1) << bits) < g->n_total) ++bits;
            size_t tmp_bytes = 0;
            GT_HIP(ctx, rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, unsigned(H), n_huge, seg_begin,
                                                            seg_end, 0u, unsigned(bits), ctx->side_stream));
            GT_HIP(ctx, g->hugerows.reserve(tmp_bytes));
            GT_HIP(ctx, rocprim::segmented_radix_sort_pairs(g->hugerows.p, tmp_bytes, kin, kout, vin, vout, unsigned(H), n_huge,
                                                            seg_begin, seg_end, 0u, unsigned(bits), ctx->side_stream));
            hipLaunchKernelGGL(huge_finalize_kernel, dim3(n_huge), dim3(64), 0, ctx->side_stream, hugelist, n_huge, seg_begin, seg_end, kout,
                               vout, g->indptr.as<int64_t>(), g->indices.as<int32_t>(), g->Kdata.as<double>(),
                               g->Pdata.as<double>(), g->degree.as<double>(), g->flags.as<uint32_t>(), fs.cmap, fs.row0);
            GT_HIP(ctx, hipGetLastError());
        }
        if (fused && n_mid > 0)   // (the rows of 129 ... kBigRow entries: next to the short rows' kernel on the main stream)
            hipLaunchKernelGGL(merge_final_kernel, dim3(n_mid), dim3(64), 0, ctx->side_stream, int64_t(n_mid), fs, g->indptr.as<int64_t>(),
                               g->indices.as<int32_t>(), g->Kdata.as<double>(), g->Pdata.as<double>(), g->degree.as<double>(),
                               g->flags.as<uint32_t>(),
                               (ctx->symm_key32 != 0 && sort_key_fits_u32(g->n_total, 8)) ? 1 : 0, g->midrows.as<int32_t>());
        GT_HIP(ctx, hipEventRecord(ctx->side_event, ctx->side_stream));
        const int key32 = (ctx->symm_key32 != 0 && sort_key_fits_u32(g->n_total, 8)) ? 1 : 0;
        if (fused) {
            const int rpw = 8;
            // (the short path's composite keys: column, tag and 7 position bits)
            if (ctx->symm_key32 != 0 && sort_key_fits_u32(g->n_total, 2))
                hipLaunchKernelGGL(merge_pairs_slots_kernel<uint32_t>, dim3((unsigned)ceil_div64(nloc, int64_t(4) * rpw)), dim3(256), 0,
                                   ctx->stream, nloc, rpw, fs, lenN_slots, perm_slots, g->indptr.as<int64_t>(),
                                   g->indices.as<int32_t>(), g->Kdata.as<double>(), g->Pdata.as<double>(), g->degree.as<double>(),
                                   g->flags.as<uint32_t>());
            else
                hipLaunchKernelGGL(merge_pairs_slots_kernel<uint64_t>, dim3((unsigned)ceil_div64(nloc, int64_t(4) * rpw)), dim3(256), 0,
                                   ctx->stream, nloc, rpw, fs, lenN_slots, perm_slots, g->indptr.as<int64_t>(),
                                   g->indices.as<int32_t>(), g->Kdata.as<double>(), g->Pdata.as<double>(), g->degree.as<double>(),
                                   g->flags.as<uint32_t>());
        } else
        hipLaunchKernelGGL(merge_final_kernel, dim3((unsigned)ceil_div64(nloc, 1)), dim3(64), 0, ctx->stream, nloc, fs,
                           g->indptr.as<int64_t>(), g->indices.as<int32_t>(), g->Kdata.as<double>(), g->Pdata.as<double>(),
                           g->degree.as<double>(), g->flags.as<uint32_t>(), key32, (const int32_t*)nullptr);
        GT_HIP(ctx, hipGetLastError());
        GT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_event, 0));
    }
    return GT_OK;
}

// returns 1: K and P are complete; 0: a union row beyond the register sorts - the caller rebuilds without this path
static int graph_finish_pairs(gt_ctx* ctx, int64_t* out_nnz, uint32_t* flags) {
    GraphState* g = ctx->graph;
    KnnWork* k = ctx->knn;
    const int64_t nloc = g->nloc;
    const int64_t n_own = g->send_counts_host[0];   // kept entries of all rows (an upper bound of what is sent)
    if (n_own <= 0) return 0;
    StageSpan span(ctx, "symmetrize");
    const int32_t* perm = k->qorder.as<int32_t>();
    const int tabs = k->tab_sorted ? 1 : 0;   // the tables lie by sorted position (and k->sh_invperm is there already)
    const bool fused = g->pairs_fused;        // ... and the affinity pass has counted the destinations (posj by g->sC, bincnt)
    const int shift = g->bin_shift, nbins = g->bin_count;   // (set, and the counters cleared, where g->pairs was decided)
    GT_HIP(ctx, k->sh_invperm.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->cnt_sorted.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->pos_sorted.reserve(size_t(nloc + 1) * sizeof(int64_t)));
    GT_HIP(ctx, g->binoff.reserve(size_t(nbins + 1) * sizeof(int64_t)));
    if (!fused) GT_HIP(ctx, g->cursor.reserve(size_t(n_own) * sizeof(uint32_t)));   // posj
    GT_HIP(ctx, g->selfbuf.reserve(size_t(n_own) * sizeof(Triplet)));
    GT_HIP(ctx, g->ucol.reserve(size_t(n_own) * sizeof(uint32_t)));
    GT_HIP(ctx, g->uval.reserve(size_t(n_own) * sizeof(double)));
    GT_HIP(ctx, g->off.reserve(size_t(nloc + 1) * sizeof(int64_t)));
    GT_HIP(ctx, g->outlen.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->bigrows.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->bigcount.reserve(8 * sizeof(uint32_t)));   // [0] long rows, [1] rows of 129 ... kBigRow entries, [2] flags, [4] huge rows, [6..7] their entries (pairs_len_kernel)
    if (fused) GT_HIP(ctx, g->midrows.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->indptr.reserve(size_t(nloc + 1) * sizeof(int64_t)));
    GT_HIP(ctx, g->degree.reserve(size_t(nloc) * sizeof(double)));
    GT_HIP(ctx, hipMemsetAsync(g->bigcount.p, 0, 8 * sizeof(uint32_t), ctx->stream));
    uint32_t* fflags = g->bigcount.as<uint32_t>() + 2;
    // posj: by the scan of the own entries - or, counted by the affinity pass, by the scan of the tables' lengths
    const int64_t* sP = fused ? g->sC.as<int64_t>() : g->pos_sorted.as<int64_t>();
    {
        StageSpan span_bins(ctx, "symm_bins");
        if (!tabs) GT_TRY(gt_sym_invperm(ctx, perm, k->sh_invperm.as<int32_t>()));
        if (!tabs)   // (tables by sorted position: the affinity launches wrote the counts by slot)
            hipLaunchKernelGGL(gather_counts_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream,
                               g->lenN.as<int32_t>(), perm, nloc, g->cnt_sorted.as<int32_t>());
        GT_TRY(exclusive_scan(ctx, g->cnt_sorted.as<int32_t>(), nullptr, nloc, g->pos_sorted.as<int64_t>(), g->scan_tmp));
        if (!fused)
            hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(nloc, 64), 2048)), dim3(256),
                               size_t(nbins) * sizeof(int32_t), ctx->stream, nloc, k->MP, k->cand_d2.as<double>(),
                               k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(), g->rlists.as<uint64_t>(),
                               g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(), g->tablen.as<int32_t>(), perm,
                               k->sh_invperm.as<int32_t>(), shift, nbins, g->pos_sorted.as<int64_t>(), g->cursor.as<uint32_t>(),
                               g->bincnt.as<int32_t>(), tabs);
        GT_TRY(exclusive_scan(ctx, g->bincnt.as<int32_t>(), nullptr, nbins, g->binoff.as<int64_t>(), g->scan_tmp));
        if (fused && n_own < (int64_t(1) << 32))
            hipLaunchKernelGGL(bin_emit_slots_kernel, dim3((unsigned)ceil_div64(nloc, kEmitRows)), dim3(256),
                               size_t(2 * nbins) * sizeof(int32_t), ctx->stream, nloc, k->MP, k->cand_d2.as<double>(),
                               g->cnt_sorted.as<int32_t>(), perm, sP, g->cursor.as<uint32_t>(), shift, nbins, g->binoff.as<int64_t>(),
                               g->bincnt.as<int32_t>() + nbins, (Triplet*)g->selfbuf.p);
        else
        hipLaunchKernelGGL(bin_emit_kernel, dim3((unsigned)ceil_div64(nloc, kEmitRows)), dim3(256), size_t(2 * nbins) * sizeof(int32_t),
                           ctx->stream, nloc, k->MP, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(),
                           g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(),
                           g->tablen.as<int32_t>(), perm, sP, g->cursor.as<uint32_t>(), shift, nbins,
                           g->binoff.as<int64_t>(), g->bincnt.as<int32_t>() + nbins, (Triplet*)g->selfbuf.p, tabs);
        hipLaunchKernelGGL(bin_fill_kernel<256>, dim3((unsigned)nbins), dim3(256), size_t(2) * (size_t(1) << shift) * sizeof(int32_t),
                           ctx->stream, nloc, k->MP, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(),
                           g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(),
                           g->tablen.as<int32_t>(), perm, shift, nbins, g->binoff.as<int64_t>(), (const Triplet*)g->selfbuf.p,
                           g->cnt_sorted.as<int32_t>(), g->pos_sorted.as<int64_t>(), g->off.as<int64_t>(), (UEntry*)nullptr,
                           g->ucol.as<uint32_t>(), g->uval.as<double>(),
                           fused ? PairsOut{g->outlen.as<int32_t>(), g->bigrows.as<int32_t>(), g->bigcount.as<uint32_t>(), g->midrows.as<int32_t>()}
                                 : PairsOut{nullptr, nullptr, nullptr, nullptr});
        GT_HIP(ctx, hipGetLastError());
    }
    // final row lengths (own + received: nothing merges), CSR offsets in row order (fused builds: bin_fill_kernel wrote them)
    if (!fused)
    hipLaunchKernelGGL(pairs_len_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream, nloc,
                       k->sh_invperm.as<int32_t>(), g->off.as<int64_t>(), g->outlen.as<int32_t>(), g->bigrows.as<int32_t>(),
                       g->bigcount.as<uint32_t>(), fflags, fused ? g->midrows.as<int32_t>() : (int32_t*)nullptr);
    GT_HIP(ctx, hipGetLastError());
    GT_TRY(exclusive_scan(ctx, g->outlen.as<int32_t>(), nullptr, nloc, g->indptr.as<int64_t>(), g->scan_tmp));
    int64_t nnz = 0;
    uint32_t ff = 0, n_mid = 0;
    uint32_t huge_host[4] = {0, 0, 0, 0};   // [0] rows beyond the register sorts, [2..3] their entries
    int64_t n_kept = n_own;   // (fused builds: n_own is the bound the buffers were sized by, the kept entries are counted here)
    {
        uint32_t bc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (pairs_len_kernel's counters, one copy)
        ReadBack rb(ctx);
        if (fused) GT_HIP(ctx, rb.add(&n_kept, g->pos_sorted.as<int64_t>() + nloc, sizeof(int64_t)));
        GT_HIP(ctx, rb.add(&nnz, g->indptr.as<int64_t>() + nloc, sizeof(int64_t)));
        GT_HIP(ctx, rb.add(bc, g->bigcount.p, 8 * sizeof(uint32_t)));
        GT_HIP(ctx, rb.sync());
        n_mid = bc[1];
        ff = bc[2];
        for (int q = 0; q < 4; ++q) huge_host[q] = bc[4 + q];
    }
    const uint32_t n_huge = huge_host[0];
    const unsigned long long huge_total = (unsigned long long)huge_host[2] | ((unsigned long long)huge_host[3] << 32);
    if (ctx->dbg_select & 2048) {
        std::vector<int32_t> ol(nloc);
        std::vector<int64_t> oh(nloc + 1);
        (void)hipMemcpy(ol.data(), g->outlen.p, size_t(nloc) * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(oh.data(), g->off.p, size_t(nloc + 1) * 8, hipMemcpyDeviceToHost);
        int64_t mx = 0, arg = 0, neg = 0, nonmono = 0;
        for (int64_t i = 0; i < nloc; ++i) {
            if (ol[i] > mx) mx = ol[i], arg = i;
            if (ol[i] < 0) ++neg;
            if (oh[i + 1] < oh[i]) ++nonmono;
        }
        uint32_t nbig = 0;
        (void)hipMemcpy(&nbig, g->bigcount.p, sizeof(uint32_t), hipMemcpyDeviceToHost);
        {
            std::vector<int64_t> bo(nbins + 1);
            (void)hipMemcpy(bo.data(), g->binoff.p, size_t(nbins + 1) * 8, hipMemcpyDeviceToHost);
            std::vector<int64_t> bs(nbins);
            for (int b = 0; b < nbins; ++b) bs[b] = bo[b + 1] - bo[b];
            std::sort(bs.begin(), bs.end());
            std::fprintf(stderr, "[gt] destination bins: %d of %d rows, triplets per bin min %lld median %lld mean %.0f p90 %lld p99 %lld max %lld\n", nbins,
                         1 << shift, (long long)bs[0], (long long)bs[nbins / 2], double(bo[nbins]) / nbins, (long long)bs[nbins * 9 / 10],
                         (long long)bs[nbins * 99 / 100], (long long)bs[nbins - 1]);
        }
        int64_t hist[6] = {0, 0, 0, 0, 0, 0};   // rows of up to 64, 128, 256, 512, 1024, more entries
        for (int64_t i = 0; i < nloc; ++i) ++hist[ol[i] <= 64 ? 0 : ol[i] <= 128 ? 1 : ol[i] <= 256 ? 2 : ol[i] <= 512 ? 3 : ol[i] <= 1024 ? 4 : 5];
        std::fprintf(stderr, "[gt] pair-resolved tail: flags %u, nnz %lld, own entries %lld; longest row %lld (row %lld), negative %lld, off not monotone at %lld places, off[n] %lld; "
                     "%u rows for the long-row kernel; rows by length <=64 %lld, <=128 %lld, <=256 %lld, <=512 %lld, <=1024 %lld, more %lld\n",
                     ff, (long long)nnz, (long long)n_own, (long long)mx, (long long)arg, (long long)neg, (long long)nonmono, (long long)oh[nloc], nbig,
                     (long long)hist[0], (long long)hist[1], (long long)hist[2], (long long)hist[3], (long long)hist[4], (long long)hist[5]);
    }
    // (huge rows: finished behind the others, below - unless they are beyond a 32-bit scratch or the option says no, then
    //  the caller builds the general way as it used to)
    if ((ff & ~kFusedHugeRow) != 0 || ((ff & kFusedHugeRow) && (ctx->symm_pair_huge == 0 || huge_total >= (1ull << 31)))) return 0;
    GT_HIP(ctx, g->indices.reserve(size_t(nnz) * sizeof(int32_t)));
    GT_HIP(ctx, g->Kdata.reserve(size_t(nnz) * sizeof(double)));
    GT_HIP(ctx, g->Pdata.reserve(size_t(nnz) * sizeof(double)));
    FusedSrc fs;
    fs.pos = k->sh_invperm.as<int32_t>();
    fs.lenN = g->lenN.as<int32_t>();
    fs.off = g->off.as<int64_t>();
    fs.sN = g->pos_sorted.as<int64_t>();
    fs.rowsrc = g->rowsrc.as<int32_t>();
    fs.cand_k = k->cand_d2.as<double>();
    fs.cand_j = k->cand_j.as<uint32_t>();
    fs.MP = k->MP;
    fs.rlists = g->rlists.as<uint64_t>();
    fs.rK = g->rK.as<double>();
    fs.rcap = g->rcap;
    fs.ucol = g->ucol.as<uint32_t>();
    fs.uval = g->uval.as<double>();
    fs.tab_sorted = tabs;
    fs.cmap = nullptr;
    fs.row0 = 0;
    GT_TRY(launch_pair_merges(ctx, g, fs, nloc, fused, n_mid, n_huge, huge_total, perm, g->cnt_sorted.as<int32_t>()));
    uint32_t fl = 0, kfl = 0;
    {
        ReadBack rb(ctx);
        GT_HIP(ctx, rb.add(&fl, g->flags.p, sizeof(uint32_t)));
        GT_HIP(ctx, rb.add(&kfl, k->gflags.p, sizeof(uint32_t)));
        GT_HIP(ctx, rb.sync());
    }
    if (fl & kFlagPairDupColumn) {
        // (never seen: two rows disagreed on whether their pair is mutual - the general tail builds the graph)
        if (ctx->dbg_select & 2048) std::fprintf(stderr, "[gt] pair-resolved tail: a union row holds a column twice - rebuilt the general way\n");
        return 0;
    }
    g->nnz0 = n_kept;
    g->nnz = nnz;
    g->finished = true;
    fl |= kfl;
    if (k->n_fallback > 0) fl |= GT_FLAG_FALLBACK_ROWS;
    if (g->n_over > 0) fl |= GT_FLAG_RADIUS_ROWS;
    if (out_nnz) *out_nnz = g->nnz;
    if (flags) *flags = fl;
    return 1;
}

// ---- pair-resolved tail of a rank of a row-sharded build ------------------------------------------------------------------
// The rank's affinity pass (graph_begin_b, pairs_shard) has settled every mutual pair in its own row - minus the merged value - with
// the partner's bandwidth from the all-gather, wherever the partner lives; what arrived here are the one-sided entries of other
// rows (and of this rank's own: they went through the exchange like everybody's) whose transposed half belongs to a local row.
// No pair meets its partner in a union row: a row's final length is its kept entries plus what it received, the CSR offsets are
// one scan away, and the merge kernels of the single-rank tail sort each union row by the CALLER's column numbers and write
// K, P and the degree straight into the CSR - no union buffer of all entries, no merge of duplicates, no compaction pass.
__global__ __launch_bounds__(256) void iota_i32_kernel(const int64_t n, int32_t* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = int32_t(i);
}
__global__ __launch_bounds__(256) void fill_recv_pairs_kernel(const Triplet* __restrict__ recv, const int64_t n_recv, const int64_t r0,
                                                              const int64_t* __restrict__ off, const int64_t* __restrict__ sN,
                                                              const int32_t* __restrict__ slot, uint32_t* __restrict__ ucol,
                                                              double* __restrict__ uval, const int32_t* __restrict__ relabel) {
    // (the received half of local row il starts at off[il] - sN[il] of ucol / uval: FusedSrc)
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < n_recv; t += int64_t(gridDim.x) * 256) {
        const Triplet tr = recv[t];
        const int64_t il = int64_t(tr.row) - r0;
        const int64_t pos = (off[il] - sN[il]) + slot[t];
        ucol[pos] = relabel ? uint32_t(relabel[tr.col]) : tr.col;
        uval[pos] = tr.val;
    }
}

// ... with destination bins of the rank's own (GraphState::pairs_shard_bins): what arrives are the few entries of OTHER ranks' rows -
// counted per bin, placed behind the local ones by the bins' cursors
__global__ __launch_bounds__(256) void recv_hist_kernel(const Triplet* __restrict__ recv, const int64_t n_recv, const int64_t r0,
                                                        const int shift, const int nbins, int32_t* __restrict__ bincnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int32_t* hist = reinterpret_cast<int32_t*>(smem_raw);
    for (int b = threadIdx.x; b < nbins; b += 256) hist[b] = 0;
    __syncthreads();
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < n_recv; t += int64_t(gridDim.x) * 256)
        atomicAdd(&hist[(int64_t(recv[t].row) - r0) >> shift], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (hist[b] != 0) atomicAdd(&bincnt[b], hist[b]);
}
constexpr int kRecvEmitChunk = 8192;   // received triplets per workgroup of recv_bin_emit_kernel
__global__ __launch_bounds__(256) void recv_bin_emit_kernel(const Triplet* __restrict__ recv, const int64_t n_recv, const int64_t r0,
                                                            const int shift, const int nbins, const int64_t* __restrict__ binoff,
                                                            int32_t* __restrict__ bincur, Triplet* __restrict__ out,
                                                            const int32_t* __restrict__ relabel) {
    // (as bin_emit_kernel: the workgroup counts its chunk per bin in the LDS, reserves its run inside each bin it touches with
    //  ONE returning atomic, places through LDS cursors - a global atomic per triplet on a thousand cursors took a millisecond)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int32_t* hist = reinterpret_cast<int32_t*>(smem_raw);
    int32_t* base = hist + nbins;
    for (int b = threadIdx.x; b < nbins; b += 256) hist[b] = 0;
    __syncthreads();
    const int64_t t0 = int64_t(blockIdx.x) * kRecvEmitChunk;
    const int64_t t1 = t0 + kRecvEmitChunk < n_recv ? t0 + kRecvEmitChunk : n_recv;
    for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) atomicAdd(&hist[(int64_t(recv[t].row) - r0) >> shift], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (hist[b] != 0) {
            base[b] = atomicAdd(&bincur[b], hist[b]);
            hist[b] = 0;
        }
    __syncthreads();
    for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) {
        Triplet tr = recv[t];
        const int64_t pl = int64_t(tr.row) - r0;
        const int b = int(pl >> shift);
        const int slot = base[b] + atomicAdd(&hist[b], 1);
        tr.row = uint32_t(pl);                                    // destination: the local row
        if (relabel) tr.col = uint32_t(relabel[tr.col]);          // column: the caller's number of the sender's row
        out[binoff[b] + slot] = tr;
    }
}
__global__ __launch_bounds__(256) void caller_rows_kernel(const int64_t nloc, const int64_t r0, const int32_t* __restrict__ relabel,
                                                          int32_t* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < nloc) out[i] = relabel ? relabel[r0 + i] : int32_t(r0 + i);
}

static int graph_finish_pairs_shard(gt_ctx* ctx, const void* recv_buf_dev, int64_t n_recv, int64_t* out_nnz, uint32_t* flags) {
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    KnnWork* k = ctx->knn;
    if (n_recv < 0 || (n_recv > 0 && !recv_buf_dev)) GT_FAIL(ctx, GT_E_ARG, "gt_graph_finish: bad receive buffer");
    const Triplet* recv = (const Triplet*)recv_buf_dev;
    const int64_t nloc = g->nloc;
    const int32_t* relabel = ctx->presorted ? ctx->vperm.as<int32_t>() : nullptr;
    g->relabelled = relabel != nullptr;
    const bool slots = g->n_over == 0;   // every row is a table row: the short rows take merge_pairs_slots_kernel
    const bool bins = g->pairs_shard_bins;
    const int64_t recv_cap = std::max<int64_t>(n_recv + (bins ? g->sc_total : 0), 1);   // entries the received halves can hold
    StageSpan span(ctx, "symmetrize");
    GT_HIP(ctx, g->ident.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->off.reserve(size_t(nloc + 1) * sizeof(int64_t)));
    GT_HIP(ctx, g->pos_sorted.reserve(size_t(nloc + 1) * sizeof(int64_t)));
    GT_HIP(ctx, g->outlen.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->bigrows.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->midrows.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, g->bigcount.reserve(8 * sizeof(uint32_t)));
    GT_HIP(ctx, g->indptr.reserve(size_t(nloc + 1) * sizeof(int64_t)));
    GT_HIP(ctx, g->degree.reserve(size_t(nloc) * sizeof(double)));
    GT_HIP(ctx, g->ucol.reserve(size_t(recv_cap) * sizeof(uint32_t)));
    GT_HIP(ctx, g->uval.reserve(size_t(recv_cap) * sizeof(double)));
    GT_HIP(ctx, hipMemsetAsync(g->bigcount.p, 0, 8 * sizeof(uint32_t), ctx->stream));
    uint32_t* fflags = g->bigcount.as<uint32_t>() + 2;
    hipLaunchKernelGGL(iota_i32_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream, nloc, g->ident.as<int32_t>());
    // own entries, scanned in row order: sN (the received half of row p starts at off[p] - sN[p] of ucol / uval)
    GT_TRY(exclusive_scan(ctx, g->lenN.as<int32_t>(), nullptr, nloc, g->pos_sorted.as<int64_t>(), g->scan_tmp));
    int64_t n_placed = n_recv;   // entries of the received halves
    if (bins) {
        StageSpan span_bins(ctx, "symm_bins");
        const int shift = g->bin_shift, nbins = g->bin_count;
        GT_HIP(ctx, g->binoff.reserve(size_t(nbins + 1) * sizeof(int64_t)));
        GT_HIP(ctx, g->selfbuf.reserve(size_t(recv_cap) * sizeof(Triplet)));
        GT_HIP(ctx, g->colid.reserve(size_t(nloc) * sizeof(int32_t)));
        hipLaunchKernelGGL(caller_rows_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream, nloc, g->r0, relabel,
                           g->colid.as<int32_t>());
        if (n_recv > 0) {
            const int64_t blocks = std::min<int64_t>(ceil_div64(n_recv, 1024), 1024);
            hipLaunchKernelGGL(recv_hist_kernel, dim3((unsigned)blocks), dim3(256), size_t(nbins) * sizeof(int32_t), ctx->stream, recv,
                               n_recv, g->r0, shift, nbins, g->bincnt.as<int32_t>());
        }
        GT_HIP(ctx, hipGetLastError());
        GT_TRY(exclusive_scan(ctx, g->bincnt.as<int32_t>(), nullptr, nbins, g->binoff.as<int64_t>(), g->scan_tmp));
        hipLaunchKernelGGL(bin_emit_slots_kernel, dim3((unsigned)ceil_div64(nloc, kEmitRows)), dim3(256),
                           size_t(2 * nbins) * sizeof(int32_t), ctx->stream, nloc, k->MP, k->cand_d2.as<double>(),
                           g->lenN.as<int32_t>(), g->colid.as<int32_t>(), g->sC.as<int64_t>(), g->cursor.as<uint32_t>(), shift, nbins,
                           g->binoff.as<int64_t>(), g->bincnt.as<int32_t>() + nbins, (Triplet*)g->selfbuf.p);
        if (n_recv > 0) {
            hipLaunchKernelGGL(recv_bin_emit_kernel, dim3((unsigned)ceil_div64(n_recv, kRecvEmitChunk)), dim3(256),
                               size_t(2 * nbins) * sizeof(int32_t), ctx->stream, recv, n_recv, g->r0, shift, nbins,
                               g->binoff.as<int64_t>(), g->bincnt.as<int32_t>() + nbins, (Triplet*)g->selfbuf.p, relabel);
        }
        hipLaunchKernelGGL(bin_fill_kernel<256>, dim3((unsigned)nbins), dim3(256), size_t(2) * (size_t(1) << shift) * sizeof(int32_t),
                           ctx->stream, nloc, k->MP, k->cand_d2.as<double>(), k->cand_j.as<uint32_t>(), g->rowsrc.as<int32_t>(),
                           g->rlists.as<uint64_t>(), g->rcounts.as<uint32_t>(), g->rcap, g->rK.as<double>(),
                           g->tablen.as<int32_t>(), g->ident.as<int32_t>(), shift, nbins, g->binoff.as<int64_t>(),
                           (const Triplet*)g->selfbuf.p, g->lenN.as<int32_t>(), g->pos_sorted.as<int64_t>(), g->off.as<int64_t>(),
                           (UEntry*)nullptr, g->ucol.as<uint32_t>(), g->uval.as<double>(),
                           PairsOut{g->outlen.as<int32_t>(), g->bigrows.as<int32_t>(), g->bigcount.as<uint32_t>(), g->midrows.as<int32_t>()});
        GT_HIP(ctx, hipGetLastError());
    } else {
        GT_HIP(ctx, g->cursor.reserve(size_t(std::max<int64_t>(n_recv, 1)) * sizeof(int32_t)));   // slot of every received triplet
        GT_HIP(ctx, hipMemsetAsync(g->lenT.p, 0, size_t(nloc) * sizeof(int32_t), ctx->stream));
        if (n_recv > 0) {
            const int64_t blocks = std::min<int64_t>(ceil_div64(n_recv, 256), 16384);
            hipLaunchKernelGGL(count_recv_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, recv, n_recv, g->r0,
                               g->lenT.as<int32_t>(), g->cursor.as<int32_t>());
        }
        GT_HIP(ctx, hipGetLastError());
        // union rows (own + received), scanned in row order: off
        GT_TRY(exclusive_scan(ctx, g->lenN.as<int32_t>(), g->lenT.as<int32_t>(), nloc, g->off.as<int64_t>(), g->scan_tmp));
        hipLaunchKernelGGL(pairs_len_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream, nloc, g->ident.as<int32_t>(),
                           g->off.as<int64_t>(), g->outlen.as<int32_t>(), g->bigrows.as<int32_t>(), g->bigcount.as<uint32_t>(), fflags,
                           slots ? g->midrows.as<int32_t>() : (int32_t*)nullptr);
        GT_HIP(ctx, hipGetLastError());
        if (n_recv > 0) {
            const int64_t blocks = std::min<int64_t>(ceil_div64(n_recv, 256), 16384);
            hipLaunchKernelGGL(fill_recv_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, recv, n_recv, g->r0,
                               g->off.as<int64_t>(), g->pos_sorted.as<int64_t>(), g->cursor.as<int32_t>(), g->ucol.as<uint32_t>(),
                               g->uval.as<double>(), relabel);
            GT_HIP(ctx, hipGetLastError());
        }
    }
    // (the rows of the CSR are the local rows in their order: its offsets are the union rows')
    GT_HIP(ctx, hipMemcpyAsync(g->indptr.p, g->off.p, size_t(nloc + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
    int64_t nnz = 0, n_kept = 0;
    uint32_t bc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        ReadBack rb(ctx);
        GT_HIP(ctx, rb.add(&nnz, g->off.as<int64_t>() + nloc, sizeof(int64_t)));
        GT_HIP(ctx, rb.add(&n_kept, g->pos_sorted.as<int64_t>() + nloc, sizeof(int64_t)));
        GT_HIP(ctx, rb.add(bc, g->bigcount.p, 8 * sizeof(uint32_t)));
        if (bins) GT_HIP(ctx, rb.add(&n_placed, g->binoff.as<int64_t>() + g->bin_count, sizeof(int64_t)));
        GT_HIP(ctx, rb.sync());
    }
    if (nnz != n_kept + n_placed) GT_FAIL(ctx, GT_E_STATE, "gt_graph_finish: received triplets for rows this rank does not own");
    const uint32_t n_mid = bc[1], ff = bc[2], n_huge = bc[4];
    const unsigned long long huge_total = (unsigned long long)bc[6] | ((unsigned long long)bc[7] << 32);
    if ((ff & ~kFusedHugeRow) != 0 || ((ff & kFusedHugeRow) && huge_total >= (1ull << 31)))
        GT_FAIL(ctx, GT_E_LIMIT, "gt_graph_finish: a union row beyond what the pair-resolved tail of a sharded build holds (set symmetrize_pairs_shard=0)");
    GT_HIP(ctx, g->indices.reserve(size_t(std::max<int64_t>(nnz, 1)) * sizeof(int32_t)));
    GT_HIP(ctx, g->Kdata.reserve(size_t(std::max<int64_t>(nnz, 1)) * sizeof(double)));
    GT_HIP(ctx, g->Pdata.reserve(size_t(std::max<int64_t>(nnz, 1)) * sizeof(double)));
    FusedSrc fs;
    fs.pos = g->ident.as<int32_t>();
    fs.lenN = g->lenN.as<int32_t>();
    fs.off = g->off.as<int64_t>();
    fs.sN = g->pos_sorted.as<int64_t>();
    fs.rowsrc = g->rowsrc.as<int32_t>();
    fs.cand_k = k->cand_d2.as<double>();
    fs.cand_j = k->cand_j.as<uint32_t>();
    fs.MP = k->MP;
    fs.rlists = g->rlists.as<uint64_t>();
    fs.rK = g->rK.as<double>();
    fs.rcap = g->rcap;
    fs.ucol = g->ucol.as<uint32_t>();
    fs.uval = g->uval.as<double>();
    fs.tab_sorted = 0;
    fs.cmap = relabel;
    fs.row0 = g->r0;
    GT_TRY(launch_pair_merges(ctx, g, fs, nloc, slots, n_mid, n_huge, huge_total, g->ident.as<int32_t>(), g->lenN.as<int32_t>()));
    uint32_t fl = 0, kfl = 0;
    {
        ReadBack rb(ctx);
        GT_HIP(ctx, rb.add(&fl, g->flags.p, sizeof(uint32_t)));
        GT_HIP(ctx, rb.add(&kfl, k->gflags.p, sizeof(uint32_t)));
        GT_HIP(ctx, rb.sync());
    }
    if (fl & kFlagPairDupColumn)
        GT_FAIL(ctx, GT_E_STATE, "gt_graph_finish: two ranks disagreed on a mutual pair (a column twice in a union row; set symmetrize_pairs_shard=0)");
    g->nnz0 = n_kept;
    g->nnz = nnz;
    g->finished = true;
    fl |= kfl;
    if (k->n_fallback > 0) fl |= GT_FLAG_FALLBACK_ROWS;
    if (g->n_over > 0) fl |= GT_FLAG_RADIUS_ROWS;
    if (out_nnz) *out_nnz = g->nnz;
    if (flags) *flags = fl;
    return GT_OK;
}

extern "C" int gt_graph_finish(gt_ctx* ctx, const void* recv_buf_dev, int64_t n_recv, int64_t* out_nnz, uint32_t* flags) {
    if (ctx && ctx->graph && ctx->graph->begun && ctx->graph->pairs_shard)
        return graph_finish_pairs_shard(ctx, recv_buf_dev, n_recv, out_nnz, flags);
    return graph_finish_impl(ctx, recv_buf_dev, n_recv, false, out_nnz, flags);
}

static int finish_normalize(gt_ctx* ctx, GraphState* g, const double* degree_all_dev) {
    StageSpan span(ctx, "normalize");
    const int64_t nloc = g->nloc;
    if (degree_all_dev && g->p.anisotropy != 0.0) {
        g->aniso_applied = true;
        // degree_all_dev is indexed by GLOBAL row; for world == 1 the local degree vector is global
        DevBuf& tmp = g->aniso_tmp;
        GT_HIP(ctx, tmp.reserve(size_t(nloc) * sizeof(double)));
        // renumbered points: the CSR holds the caller's column numbers, degree_all_dev is indexed by them (a sharded caller
        // hands it over that way; the local vector of a single-rank build is scattered here)
        const int32_t* rowid = g->relabelled ? ctx->vperm.as<int32_t>() : nullptr;
        if (rowid && degree_all_dev == g->degree.as<double>()) {
            GT_HIP(ctx, g->deg_caller.reserve(size_t(g->n_total) * sizeof(double)));
            hipLaunchKernelGGL(scatter_f64_kernel, dim3((unsigned)ceil_div64(nloc, 256)), dim3(256), 0, ctx->stream,
                               g->degree.as<double>(), rowid + g->r0, nloc, g->deg_caller.as<double>());
            degree_all_dev = g->deg_caller.as<double>();
        }
        hipLaunchKernelGGL(anisotropy_kernel, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc, g->r0,
                           g->indptr.as<int64_t>(), g->indices.as<int32_t>(), g->Kdata.as<double>(), degree_all_dev,
                           g->p.anisotropy, tmp.as<double>(), rowid);
        hipError_t e = hipMemcpyAsync(g->degree.p, tmp.p, size_t(nloc) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->set_error(std::string("anisotropy: ") + hipGetErrorString(e));
            return GT_E_HIP;
        }
    }
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc,
                       g->indptr.as<int64_t>(), g->Kdata.as<double>(), g->Pdata.as<double>());
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

extern "C" int gt_graph_anisotropy(gt_ctx* ctx, const double* degree_all_dev) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->finished) GT_FAIL(ctx, GT_E_STATE, "gt_graph_anisotropy: call gt_graph_finish first");
    if (!degree_all_dev) GT_FAIL(ctx, GT_E_ARG, "gt_graph_anisotropy: degree vector is NULL");
    // the rescaling overwrites K in place: a second call would apply it twice
    if (g->aniso_applied) GT_FAIL(ctx, GT_E_STATE, "gt_graph_anisotropy: the anisotropy of this build has been applied already");
    GT_TRY(finish_normalize(ctx, g, degree_all_dev));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

extern "C" int gt_graph_diff_aff(gt_ctx* ctx, const double* degree_all_dev, double* out, int32_t on_device) {
    if (!ctx || !out) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->finished) GT_FAIL(ctx, GT_E_STATE, "gt_graph_diff_aff: no finished graph");
    if (!degree_all_dev && g->world != 1) GT_FAIL(ctx, GT_E_ARG, "gt_graph_diff_aff: a sharded build needs the degrees of all rows");
    if (g->external) GT_FAIL(ctx, GT_E_STATE, "gt_graph_diff_aff: the device holds a rectangular kernel");
    const double* deg = degree_all_dev ? degree_all_dev : g->degree.as<double>();
    const int32_t* rowid = g->relabelled ? ctx->vperm.as<int32_t>() : nullptr;
    if (rowid && !degree_all_dev) {   // (one rank, renumbered points: the degrees in the caller's numbering)
        GT_HIP(ctx, g->deg_caller.reserve(size_t(g->n_total) * sizeof(double)));
        hipLaunchKernelGGL(scatter_f64_kernel, dim3((unsigned)ceil_div64(g->nloc, 256)), dim3(256), 0, ctx->stream,
                           g->degree.as<double>(), rowid + g->r0, g->nloc, g->deg_caller.as<double>());
        deg = g->deg_caller.as<double>();
    }
    double* od = out;
    DevBuf tmp;
    if (!on_device) {
        GT_HIP(ctx, tmp.reserve(size_t(std::max<int64_t>(g->nnz, 1)) * sizeof(double)));
        od = tmp.as<double>();
    }
    hipLaunchKernelGGL(diff_aff_kernel, dim3((unsigned)ceil_div64(g->nloc, 4)), dim3(256), 0, ctx->stream, g->nloc,
                       (degree_all_dev || rowid) ? g->r0 : int64_t(0), g->indptr.as<int64_t>(), g->indices.as<int32_t>(),
                       g->Kdata.as<double>(), deg, od, rowid);
    int rc = GT_OK;
    if (hipGetLastError() != hipSuccess) {
        ctx->set_error("gt_graph_diff_aff: launch failed");
        rc = GT_E_HIP;
    }
    if (rc == GT_OK && !on_device) rc = gt_copy_to_host(ctx, out, od, size_t(g->nnz) * sizeof(double));
    hipError_t es = hipStreamSynchronize(ctx->stream);
    tmp.release();
    if (rc == GT_OK && es != hipSuccess) {
        ctx->set_error(std::string("gt_graph_diff_aff: ") + hipGetErrorString(es));
        rc = GT_E_HIP;
    }
    return rc;
}

extern "C" int gt_graph_build(gt_ctx* ctx, const gt_knn_params* params, int64_t* out_nnz, uint32_t* flags) {
    if (!ctx) return GT_E_ARG;
    int64_t splits[2] = {0, ctx->n};
    int64_t sendc[1] = {0};
    for (int attempt = 0;; ++attempt) {
        ctx->in_graph_build = 1;
        const int rc_b = gt_graph_begin(ctx, params, 1, 0, splits, sendc);
        ctx->in_graph_build = 0;
        ctx->keep_stages = 0;
        if (rc_b != GT_OK) return rc_b;
        if (!ctx->graph->pairs) break;
        if (sendc[0] == 0) break;   // (no kept entry at all: nothing is settled in the tables, the general tail returns the empty graph)
        const int rc = graph_finish_pairs(ctx, out_nnz, flags);
        if (rc < 0) return rc;
        if (rc == 1) {
            ctx->graph->bins_used = true;
            return GT_OK;
        }
        // a union row beyond the register sorts (a hub of the transpose): the tables hold settled pairs the general tail
        // cannot read - this point set is built again, now and from here on, the general way
        ctx->symm_pair_ok = 0;
        ctx->keep_stages = 1;   // (the second attempt's stage times are added to the first's)
        if (attempt > 0) GT_FAIL(ctx, GT_E_STATE, "gt_graph_build: the pair-resolved path was taken twice");
    }
    GraphState* g = ctx->graph;
    KnnWork* k = ctx->knn;
    // every row is here: with the cell-sorted order of the points at hand the transpose is built through destination
    // bins instead of the triplet exchange (option symmetrize_bins: -1 auto, 0 off, 1 on where possible)
    const bool bins = ctx->symm_bins != 0 && sendc[0] > 0 && k->ordered && k->nq == g->nloc && g->r0 == 0 && !g->external && !ctx->presorted &&
                      g->nloc < (int64_t(1) << 31) && (ctx->symm_bins > 0 || g->nloc >= 65536);
    g->bins_used = bins;
    if (bins) return graph_finish_impl(ctx, nullptr, 0, true, out_nnz, flags);
    if (sendc[0] > 0) {
        GT_HIP(ctx, g->selfbuf.reserve(size_t(sendc[0]) * sizeof(Triplet)));
        GT_TRY(gt_graph_emit(ctx, g->selfbuf.p));
    }
    return gt_graph_finish(ctx, sendc[0] > 0 ? g->selfbuf.p : nullptr, sendc[0], out_nnz, flags);
}

/* Symmetrisation, anisotropy and row normalisation of a caller-assembled square kernel (CSR, unique ascending or
 * unordered columns per row, non-negative values): the tail of BaseGraph._build_kernel + BaseGraph.P for graphs whose
 * unsymmetrised kernel is a composition of kNN kernels (MNNGraph.build_kernel, graphs.py:1857-1936). */
extern "C" int gt_csr_graph_build(gt_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* indices,
                                  const double* data, int32_t kernel_symm, double theta, double anisotropy,
                                  int64_t* out_nnz, uint32_t* flags) {
    if (!ctx || !indptr || n <= 0) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->reset_stages();
    if (n >= (int64_t(1) << 31)) GT_FAIL(ctx, GT_E_LIMIT, "gt_csr_graph_build: too many rows");
    if (indptr[0] != 0) GT_FAIL(ctx, GT_E_ARG, "gt_csr_graph_build: indptr[0] must be 0");
    const int64_t nnz0 = indptr[n];
    int64_t maxlen = 1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t len = indptr[i + 1] - indptr[i];
        if (len < 0) GT_FAIL(ctx, GT_E_ARG, "gt_csr_graph_build: indptr must be ascending");
        maxlen = std::max(maxlen, len);
    }
    if (nnz0 > 0 && (!indices || !data)) GT_FAIL(ctx, GT_E_ARG, "gt_csr_graph_build: indices / data are NULL");
    if (double(n) * double(maxlen) * 16.0 > 128e9)
        GT_FAIL(ctx, GT_E_LIMIT, "gt_csr_graph_build: padded row layout would exceed 128 GB");
    if (!ctx->knn) ctx->knn = new KnnWork();
    KnnWork* k = ctx->knn;
    k->n_fallback = 0;
    k->n_fallback_exhaustive = 0;
    GT_HIP(ctx, k->gflags.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(k->gflags.p, 0, sizeof(uint32_t), ctx->stream));
    if (!ctx->graph) ctx->graph = new GraphState();
    GraphState* g = ctx->graph;
    g->p = gt_knn_params{};
    g->p.knn = 1;
    g->p.kernel_symm = kernel_symm;
    g->p.theta = theta;
    g->p.anisotropy = anisotropy;
    g->world = 1;
    g->rank = 0;
    g->splits = {0, n};
    g->r0 = 0;
    g->r1 = n;
    g->nloc = n;
    g->n_total = n;
    g->begun = false;
    g->finished = false;
    g->aniso_applied = false;
    g->external = true;   // rows are not tied to bound points
    g->n_over = 0;
    g->radius_retries = 0;
    g->rcap = int32_t(maxlen);
    const int count_owners = (kernel_symm != GT_SYMM_NONE) ? 1 : 0;
    DevBuf d_indptr, d_indices, d_data;
    int rc = GT_OK;
    do {
        hipError_t e = d_indptr.reserve(size_t(n + 1) * sizeof(int64_t));
        if (e == hipSuccess) e = d_indices.reserve(size_t(std::max<int64_t>(nnz0, 1)) * sizeof(int32_t));
        if (e == hipSuccess) e = d_data.reserve(size_t(std::max<int64_t>(nnz0, 1)) * sizeof(double));
        if (e == hipSuccess) e = g->rlists.reserve(size_t(n) * size_t(maxlen) * sizeof(uint64_t));
        if (e == hipSuccess) e = g->rK.reserve(size_t(n) * size_t(maxlen) * sizeof(double));
        if (e == hipSuccess) e = g->rcounts.reserve(size_t(n) * sizeof(uint32_t));
        if (e == hipSuccess) e = g->rowsrc.reserve(size_t(n) * sizeof(int32_t));
        if (e == hipSuccess) e = g->lenN.reserve(size_t(n) * sizeof(int32_t));
        if (e == hipSuccess) e = g->tablen.reserve(size_t(n) * sizeof(int32_t));
        if (e == hipSuccess) e = g->lenT.reserve(size_t(n) * sizeof(int32_t));
        if (e == hipSuccess) e = g->cursor.reserve(size_t(n) * sizeof(int32_t));
        if (e == hipSuccess) e = g->ownercnt.reserve(size_t(n) * sizeof(int32_t));
        if (e == hipSuccess) e = g->bw.reserve(size_t(n) * sizeof(double));
        if (e == hipSuccess) e = g->flags.reserve(sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(g->flags.p, 0, sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(g->bw.p, 0, size_t(n) * sizeof(double), ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_indptr.p, indptr, size_t(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nnz0 > 0)
            e = hipMemcpyAsync(d_indices.p, indices, size_t(nnz0) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nnz0 > 0)
            e = hipMemcpyAsync(d_data.p, data, size_t(nnz0) * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            StageSpan span(ctx, "symmetrize");
            hipLaunchKernelGGL(csr_ingest_kernel, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0, ctx->stream, n,
                               d_indptr.as<int64_t>(), d_indices.as<int32_t>(), d_data.as<double>(), g->rcap, count_owners,
                               g->rlists.as<uint64_t>(), g->rK.as<double>(), g->rcounts.as<uint32_t>(),
                               g->rowsrc.as<int32_t>(), g->lenN.as<int32_t>(), g->ownercnt.as<int32_t>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->set_error(std::string("gt_csr_graph_build: ") + hipGetErrorString(e));
            rc = GT_E_HIP;
        }
    } while (0);
    d_indptr.release();
    d_indices.release();
    d_data.release();
    GT_TRY(rc);
    g->send_counts_host.assign(1, 0);
    if (count_owners) {
        GT_HIP(ctx, g->ownerpos.reserve(size_t(n + 1) * sizeof(int64_t)));
        GT_TRY(exclusive_scan(ctx, g->ownercnt.as<int32_t>(), nullptr, n, g->ownerpos.as<int64_t>(), g->scan_tmp));
        g->send_counts_host[0] = nnz0;
    }
    g->begun = true;
    const int64_t sendc = g->send_counts_host[0];
    if (sendc > 0) {
        GT_HIP(ctx, g->selfbuf.reserve(size_t(sendc) * sizeof(Triplet)));
        GT_TRY(gt_graph_emit(ctx, g->selfbuf.p));
    }
    return gt_graph_finish(ctx, sendc > 0 ? g->selfbuf.p : nullptr, sendc, out_nnz, flags);
}

extern "C" int gt_graph_rows(const gt_ctx* ctx, int64_t* row0, int64_t* row1, int64_t* nnz) {
    if (!ctx || !ctx->graph || !ctx->graph->finished) return GT_E_STATE;
    if (row0) *row0 = ctx->graph->r0;
    if (row1) *row1 = ctx->graph->r1;
    if (nnz) *nnz = ctx->graph->nnz;
    return GT_OK;
}

extern "C" int gt_graph_fetch_csr(gt_ctx* ctx, int32_t which, double* data, int32_t* indices, int64_t* indptr,
                                  int32_t on_device) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->finished) GT_FAIL(ctx, GT_E_STATE, "gt_graph_fetch_csr: no finished graph");
    const double* src = which == GT_CSR_P ? g->Pdata.as<double>() : g->Kdata.as<double>();
    if (!on_device) {
        // caller's numpy arrays: pipelined through pinned slots (gt_hostcopy.cpp)
        if (data) GT_TRY(gt_copy_to_host(ctx, data, src, size_t(g->nnz) * sizeof(double)));
        if (indices) GT_TRY(gt_copy_to_host(ctx, indices, g->indices.p, size_t(g->nnz) * sizeof(int32_t)));
        if (indptr) GT_TRY(gt_copy_to_host(ctx, indptr, g->indptr.p, size_t(g->nloc + 1) * sizeof(int64_t)));
        return GT_OK;
    }
    const hipMemcpyKind kind = hipMemcpyDeviceToDevice;
    if (data && g->nnz > 0) GT_HIP(ctx, hipMemcpyAsync(data, src, size_t(g->nnz) * sizeof(double), kind, ctx->stream));
    if (indices && g->nnz > 0)
        GT_HIP(ctx, hipMemcpyAsync(indices, g->indices.p, size_t(g->nnz) * sizeof(int32_t), kind, ctx->stream));
    if (indptr) GT_HIP(ctx, hipMemcpyAsync(indptr, g->indptr.p, size_t(g->nloc + 1) * sizeof(int64_t), kind, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

// K and P of the owned rows to the host in ONE pass over the link: structure and K values are copied, the P values are derived
// from them on the host by the copy lanes - P[e] = K[e] / degree[row], the very division the device made (bit-identical,
// tested) - so a third of the host-complete transfer (0.93 of 2.3 GB at C3) never crosses PCIe.  Falls back to a plain copy of
// the device's P when a value is negative (the |v| sums behind P then differ from the degrees).  All four outputs are the
// caller's host arrays (nnz / nnz / nloc + 1 / nnz entries).
extern "C" int gt_graph_fetch_kp(gt_ctx* ctx, double* K_data, int32_t* indices, int64_t* indptr, double* P_data) {
    if (!ctx || !K_data || !indices || !indptr || !P_data) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->finished) GT_FAIL(ctx, GT_E_STATE, "gt_graph_fetch_kp: no finished graph");
    if (g->p.anisotropy != 0.0 && !g->aniso_applied) GT_FAIL(ctx, GT_E_STATE, "gt_graph_fetch_kp: the anisotropy of this build is still to be applied");
    // (kept per thread: a fresh 8 MB vector is zeroed and faulted in on every call - a millisecond of the host-complete graph)
    static thread_local std::vector<double> deg;
    if (deg.size() < size_t(g->nloc)) deg.resize(size_t(g->nloc));
    {
        HostTrace t(ctx, "fetch_kp: indptr + degrees");
        GT_TRY(gt_copy_to_host(ctx, indptr, g->indptr.p, size_t(g->nloc + 1) * sizeof(int64_t)));
        GT_TRY(gt_copy_to_host(ctx, deg.data(), g->degree.p, size_t(g->nloc) * sizeof(double)));
    }
    {
        HostTrace t(ctx, "fetch_kp: indices");
        GT_TRY(gt_copy_to_host(ctx, indices, g->indices.p, size_t(g->nnz) * sizeof(int32_t)));
    }
    int negative = 0;
    {
        HostTrace t(ctx, "fetch_kp: K with P derived");
        GT_TRY(gt_fetch_kp_host(ctx, K_data, P_data, g->Kdata.as<double>(), g->nnz, reinterpret_cast<const long long*>(indptr),
                                deg.data(), g->nloc, &negative));
    }
    if (negative) GT_TRY(gt_copy_to_host(ctx, P_data, g->Pdata.p, size_t(g->nnz) * sizeof(double)));
    return GT_OK;
}

// ---- SpMM: rows of K / P times a dense matrix -----------------------------------------------------
// One wave per row, one lane per output column (chunks of 64 columns): the entries of the row are walked in order,
// (value, column) are wave-uniform loads, the gathered row of X is read coalesced.  HBM/L2-bound: nnz * ncols * 8 bytes
// of gathers.
__global__ __launch_bounds__(256) void spmm_rows_kernel(const int64_t nloc, const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, const double* __restrict__ data,
                                                        const double* __restrict__ X, const int64_t ncols,
                                                        double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int64_t e0 = indptr[i], e1 = indptr[i + 1];
    for (int64_t c0 = 0; c0 < ncols; c0 += 64) {
        const int64_t c = c0 + lane;
        double acc = 0.0;
        if (c < ncols) {
            for (int64_t e = e0; e < e1; ++e) {
                const double a = data[e];
                const double x = X[int64_t(indices[e]) * ncols + c];
                acc = acc + a * x;   // -ffp-contract=off: multiply, then add (scipy's csr_matvecs)
            }
            out[i * ncols + c] = acc;
        }
    }
}

extern "C" int gt_graph_spmm(gt_ctx* ctx, int32_t which, const double* X, int64_t ncols, double* out, int32_t on_device) {
    if (!ctx || !X || !out || ncols <= 0) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->finished) GT_FAIL(ctx, GT_E_STATE, "gt_graph_spmm: no finished graph");
    const int64_t n = g->n_total;
    const double* Xd = X;
    double* od = out;
    if (!on_device) {
        GT_HIP(ctx, g->spmm_in.reserve(size_t(n) * ncols * sizeof(double)));
        GT_HIP(ctx, g->spmm_out.reserve(size_t(g->nloc) * ncols * sizeof(double)));
        GT_TRY(gt_copy_from_host(ctx, g->spmm_in.p, X, size_t(n) * ncols * sizeof(double)));
        Xd = g->spmm_in.as<double>();
        od = g->spmm_out.as<double>();
    }
    const double* vals = which == GT_CSR_P ? g->Pdata.as<double>() : g->Kdata.as<double>();
    {
        StageSpan span(ctx, "spmm");
        hipLaunchKernelGGL(spmm_rows_kernel, dim3((unsigned)ceil_div64(g->nloc, 4)), dim3(256), 0, ctx->stream, g->nloc,
                           g->indptr.as<int64_t>(), g->indices.as<int32_t>(), vals, Xd, ncols, od);
        GT_HIP(ctx, hipGetLastError());
    }
    if (!on_device) GT_TRY(gt_copy_to_host(ctx, out, od, size_t(g->nloc) * ncols * sizeof(double)));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

extern "C" int gt_graph_fetch_vec(gt_ctx* ctx, int32_t which, double* out, int32_t on_device) {
    if (!ctx || !out) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->begun) GT_FAIL(ctx, GT_E_STATE, "gt_graph_fetch_vec: no graph");
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const void* src = nullptr;
    if (which == GT_VEC_BANDWIDTH)
        src = g->bw.p;
    else if (which == GT_VEC_DEGREE && g->finished)
        src = g->degree.p;
    if (!src) GT_FAIL(ctx, GT_E_ARG, "gt_graph_fetch_vec: unknown or unavailable vector");
    GT_HIP(ctx, hipMemcpyAsync(out, src, size_t(g->nloc) * sizeof(double), kind, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

extern "C" int gt_graph_stats(const gt_ctx* ctx, int64_t* out4) {
    if (!ctx || !out4 || !ctx->graph) return GT_E_STATE;
    out4[0] = ctx->knn ? ctx->knn->n_fallback : 0;
    out4[1] = ctx->graph->n_over;
    out4[2] = ctx->graph->nnz0;
    out4[3] = ctx->graph->radius_retries;
    return GT_OK;
}

// ---- dense copy of the owned rows of K or P (the exact graph built through the sparse path: TraditionalGraph from data) ----
// zero fill of the whole output as ONE linear stream (16-byte non-temporal stores, consecutive workgroups on consecutive 4 KB
// pieces: a wave per row - 800 KB rows at N = 2e5 - keeps thousands of distant streams open and reached 3.5 TB/s), then the
// entries of every row scattered over it
__global__ __launch_bounds__(256) void dense_zero_kernel(void* __restrict__ out, const size_t bytes) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4* o = reinterpret_cast<u32x4*>(out);
    const size_t n16 = bytes / 16;
    const u32x4 z = {0u, 0u, 0u, 0u};
    const size_t stride = size_t(gridDim.x) * 256;
    size_t f = size_t(blockIdx.x) * 256 + threadIdx.x;
    for (; f + 3 * stride < n16; f += 4 * stride) {
        __builtin_nontemporal_store(z, o + f);
        __builtin_nontemporal_store(z, o + f + stride);
        __builtin_nontemporal_store(z, o + f + 2 * stride);
        __builtin_nontemporal_store(z, o + f + 3 * stride);
    }
    for (; f < n16; f += stride) __builtin_nontemporal_store(z, o + f);
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) reinterpret_cast<unsigned char*>(out)[n16 * 16 + threadIdx.x] = 0;
}

template <typename TO>
__global__ __launch_bounds__(256) void csr_to_dense_kernel(const int64_t nloc, const int64_t ncols, const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices, const double* __restrict__ data,
                                                           TO* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (i >= nloc) return;
    TO* row = out + i * ncols;
    for (int64_t e = indptr[i] + lane; e < indptr[i + 1]; e += 64) row[indices[e]] = TO(data[e]);
}

extern "C" int gt_graph_to_dense(gt_ctx* ctx, int32_t which, void* out, int32_t out_dtype, int32_t out_on_device) {
    if (!ctx || !out) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->finished) GT_FAIL(ctx, GT_E_STATE, "gt_graph_to_dense: no finished graph");
    if (which != GT_CSR_K && which != GT_CSR_P) GT_FAIL(ctx, GT_E_ARG, "which must be GT_CSR_K or GT_CSR_P");
    if (out_dtype != GT_F32 && out_dtype != GT_F64) GT_FAIL(ctx, GT_E_ARG, "out_dtype must be GT_F32 or GT_F64");
    const int64_t nloc = g->nloc, ncols = g->n_total;
    const size_t esz = out_dtype == GT_F32 ? 4 : 8;
    const size_t bytes = size_t(nloc) * size_t(ncols) * esz;
    DevBuf tmp;
    void* dst = out;
    if (!out_on_device) {
        GT_HIP(ctx, tmp.reserve(bytes));
        dst = tmp.p;
    }
    const double* data = which == GT_CSR_K ? g->Kdata.as<double>() : g->Pdata.as<double>();
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)
        hipLaunchKernelGGL(dense_zero_kernel, dim3((unsigned)std::min<size_t>(ceil_div64(int64_t(bytes / 16) + 1, 256), size_t(1) << 20)),
                           dim3(256), 0, ctx->stream, dst, bytes);
    else
        GT_HIP(ctx, hipMemsetAsync(dst, 0, bytes, ctx->stream));
    if (out_dtype == GT_F32)
        hipLaunchKernelGGL(csr_to_dense_kernel<float>, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc, ncols,
                           g->indptr.as<int64_t>(), g->indices.as<int32_t>(), data, (float*)dst);
    else
        hipLaunchKernelGGL(csr_to_dense_kernel<double>, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc, ncols,
                           g->indptr.as<int64_t>(), g->indices.as<int32_t>(), data, (double*)dst);
    hipError_t e = hipGetLastError();
    int rc = GT_OK;
    if (e == hipSuccess && !out_on_device) rc = gt_copy_to_host(ctx, out, dst, bytes);
    if (e == hipSuccess && rc == GT_OK) e = hipStreamSynchronize(ctx->stream);
    tmp.release();
    if (e != hipSuccess) {
        ctx->set_error(std::string("gt_graph_to_dense: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    return rc;
}
