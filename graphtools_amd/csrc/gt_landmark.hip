// Landmark operator algebra (LandmarkGraph, graphtools/graphs.py:1169-1182, 1232-1246) and the random
// landmark assignment (graphs.py:1200-1213) on the device.
//
// With S the one-hot N x L cluster matrix:  pmn = S^T K (L x N), pnm = pmn^T (N x L); both are row-L1
// normalised (p^mn, p^nm); landmark_op = p^mn . p^nm (dense L x L), transitions = p^nm.
// K is symmetric (every symmetrisation mode produces bitwise-symmetric values), so row n of pnm is the
// cluster-wise sum of ROW n of K - a purely row-local operation on the row-sharded CSR:
//   a_n[c]   = sum_{j in cluster c} K[n, j]                    (sorted by c; sequential sum in j order)
//   p^nm[n]  = a_n / sum_c a_n[c]
//   M[c, :] += a_n[c] * p^nm[n, :]   ,   R[c] += a_n[c]         (partials over the owned rows n)
//   landmark_op[c, :] = M[c, :] / R[c]                          (after summing partials over ranks)
// M is accumulated per landmark row in LDS (one workgroup per landmark, ds_add_f64), so the only global
// atomics are the counting-sort cursors that build the transposed structure.
#include "gt_common.h"
#include "gt_hostcopy.h"
#include "gt_device.h"
#include "gt_graph_state.h"

#include <vector>

struct LandmarkState {
    int32_t L = 0;
    int64_t nloc = 0, tnnz = 0;
    DevBuf clusters, tlen, tptr, scol, sval, tcol, tval, tnorm, rowsum, ccount, cptr, ccur, prow, pval, M, R, bigrows,
        bigcount;
};

void gt_free_landmark_state(gt_ctx* ctx) {
    LandmarkState* l = reinterpret_cast<LandmarkState*>(ctx->landmark);
    if (!l) return;
    for (DevBuf* b : {&l->clusters, &l->tlen, &l->tptr, &l->scol, &l->sval, &l->tcol, &l->tval, &l->tnorm, &l->rowsum,
                      &l->ccount, &l->cptr, &l->ccur, &l->prow, &l->pval, &l->M, &l->R, &l->bigrows, &l->bigcount})
        b->release();
    delete l;
    ctx->landmark = nullptr;
}

int gt_exclusive_scan_i32(gt_ctx* ctx, const int32_t* a, int64_t n, int64_t* out);  // gt_sparse.hip

namespace {

constexpr int kRowCap = 512;   // rows up to this many entries are aggregated by one wave in registers/LDS
constexpr int kBigStage = 512; // entries of a longer row staged in LDS at a time (aggregate_big_rows_kernel)

// sort one row by (cluster, column) and emit (cluster, sum) pairs; returns the number of pairs
template <int NT>
__device__ __forceinline__ int aggregate_row(const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                             const int L_row, const int32_t* __restrict__ clusters, const int lane,
                                             uint64_t* __restrict__ s_key, double* __restrict__ s_val,
                                             int32_t* __restrict__ out_c, double* __restrict__ out_v) {
    uint64_t hi[NT], lo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = t * 64 + lane;
        hi[t] = ~0ull;
        lo[t] = 0ull;
        if (p < L_row) {
            const int32_t j = cols[p];
            hi[t] = (uint64_t(uint32_t(clusters[j])) << 32) | uint64_t(uint32_t(j));
            lo[t] = (uint64_t)__double_as_longlong(vals[p]);
        }
    }
    wave_bitonic_asc_pair<NT>(hi, lo, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s_key[t * 64 + lane] = hi[t];
        s_val[t * 64 + lane] = __longlong_as_double((long long)lo[t]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int count = 0;
    for (int p0 = 0; p0 < L_row; p0 += 64) {
        const int p = p0 + lane;
        bool start = false;
        uint32_t c = 0;
        double sum = 0.0;
        if (p < L_row) {
            c = uint32_t(s_key[p] >> 32);
            start = (p == 0) || (uint32_t(s_key[p - 1] >> 32) != c);
            if (start) {
                int q = p;
                while (q < L_row && uint32_t(s_key[q] >> 32) == c) {
                    sum += s_val[q];
                    ++q;
                }
            }
        }
        int total;
        const int pos = wave_prefix_count(start, lane, total);
        if (start) {
            out_c[count + pos] = int32_t(c);
            out_v[count + pos] = sum;
        }
        count += total;
    }
    __builtin_amdgcn_wave_barrier();
    return count;
}

__global__ __launch_bounds__(256) void aggregate_rows_kernel(const int64_t nloc, const int64_t* __restrict__ indptr,
                                                             const int32_t* __restrict__ indices,
                                                             const double* __restrict__ Kdata,
                                                             const int32_t* __restrict__ clusters,
                                                             int32_t* __restrict__ scol, double* __restrict__ sval,
                                                             int32_t* __restrict__ tlen, int32_t* __restrict__ bigrows,
                                                             uint32_t* __restrict__ bigcount) {
    __shared__ uint64_t s_key_all[4 * kRowCap];
    __shared__ double s_val_all[4 * kRowCap];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int64_t s = indptr[i];
    const int64_t L64 = indptr[i + 1] - s;
    if (L64 > kRowCap) {
        if (lane == 0) bigrows[atomicAdd(bigcount, 1u)] = int32_t(i);
        return;
    }
    const int Lr = int(L64);
    uint64_t* sk = s_key_all + w * kRowCap;
    double* sv = s_val_all + w * kRowCap;
    int c;
    if (Lr <= 64)
        c = aggregate_row<1>(indices + s, Kdata + s, Lr, clusters, lane, sk, sv, scol + s, sval + s);
    else if (Lr <= 128)
        c = aggregate_row<2>(indices + s, Kdata + s, Lr, clusters, lane, sk, sv, scol + s, sval + s);
    else if (Lr <= 256)
        c = aggregate_row<4>(indices + s, Kdata + s, Lr, clusters, lane, sk, sv, scol + s, sval + s);
    else
        c = aggregate_row<8>(indices + s, Kdata + s, Lr, clusters, lane, sk, sv, scol + s, sval + s);
    if (lane == 0) tlen[i] = c;
}

// long rows: dense accumulator over the L landmarks in LDS (one workgroup per row)
__global__ __launch_bounds__(256) void aggregate_big_rows_kernel(const int32_t* __restrict__ bigrows,
                                                                 const int64_t* __restrict__ indptr,
                                                                 const int32_t* __restrict__ indices,
                                                                 const double* __restrict__ Kdata,
                                                                 const int32_t* __restrict__ clusters, const int L,
                                                                 int32_t* __restrict__ scol, double* __restrict__ sval,
                                                                 int32_t* __restrict__ tlen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* acc = reinterpret_cast<double*>(smem_raw);   // [L]
    __shared__ int wtot[4];
    __shared__ int base_sh;
    __shared__ int32_t st_c[kBigStage];
    __shared__ double st_v[kBigStage];
    const int64_t i = bigrows[blockIdx.x];
    const int64_t s = indptr[i], e = indptr[i + 1];
    for (int c = threadIdx.x; c < L; c += 256) acc[c] = 0.0;
    if (threadIdx.x == 0) base_sh = 0;
    __syncthreads();
    // a cluster's sum belongs to ONE thread (cluster % 256), which meets the row's entries in column order: the sums of the
    // wave-per-row path, bit for bit, whatever the scheduling (LDS atomics summed in arrival order: the last bit varied from
    // run to run).  The entries are staged by all threads, then every thread walks the stage (broadcast reads).
    for (int64_t p0 = s; p0 < e; p0 += kBigStage) {
        const int m = int(e - p0 < int64_t(kBigStage) ? e - p0 : int64_t(kBigStage));
        for (int q = threadIdx.x; q < m; q += 256) {
            st_c[q] = clusters[indices[p0 + q]];
            st_v[q] = Kdata[p0 + q];
        }
        __syncthreads();
        for (int q = 0; q < m; ++q) {
            const int c = st_c[q];
            if ((c & 255) == int(threadIdx.x)) acc[c] += st_v[q];
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c0 = 0; c0 < L; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const bool nz = c < L && acc[c] != 0.0;
        int total;
        const int pos = wave_prefix_count(nz, lane, total);
        if (lane == 0) wtot[w] = total;
        __syncthreads();
        int off = base_sh;
        for (int r = 0; r < w; ++r) off += wtot[r];
        if (nz) {
            scol[s + off + pos] = c;
            sval[s + off + pos] = acc[c];
        }
        __syncthreads();
        if (threadIdx.x == 0) base_sh += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) tlen[i] = base_sh;
}

// compact the per-row aggregates into CSR, normalise, count entries per landmark
__global__ __launch_bounds__(256) void compact_transitions_kernel(const int64_t nloc, const int64_t* __restrict__ indptr,
                                                                  const int32_t* __restrict__ tlen,
                                                                  const int64_t* __restrict__ tptr,
                                                                  const int32_t* __restrict__ scol,
                                                                  const double* __restrict__ sval,
                                                                  int32_t* __restrict__ tcol, double* __restrict__ tval,
                                                                  double* __restrict__ tnorm, int32_t* __restrict__ ccount) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    const int64_t s = indptr[i];
    const int64_t dst = tptr[i];
    const int n = tlen[i];
    double sum = 0.0;
    for (int e = lane; e < n; e += 64) sum += fabs(sval[s + e]);
    sum = wave_sum_f64(sum);
    for (int e = lane; e < n; e += 64) {
        const int32_t c = scol[s + e];
        const double v = sval[s + e];
        tcol[dst + e] = c;
        tval[dst + e] = v;
        tnorm[dst + e] = (sum != 0.0) ? v / sum : v;
        atomicAdd(&ccount[c], 1);
    }
}

// transposed structure: for every landmark the (row, value) pairs of pnm
__global__ __launch_bounds__(256) void scatter_by_landmark_kernel(const int64_t nloc, const int64_t* __restrict__ tptr,
                                                                  const int32_t* __restrict__ tcol,
                                                                  const double* __restrict__ tval,
                                                                  const int64_t* __restrict__ cptr, int32_t* __restrict__ ccur,
                                                                  int32_t* __restrict__ prow, double* __restrict__ pval) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t i = int64_t(blockIdx.x) * 4 + w;
    if (i >= nloc) return;
    for (int64_t e = tptr[i] + lane; e < tptr[i + 1]; e += 64) {
        const int32_t c = tcol[e];
        const int slot = atomicAdd(&ccur[c], 1);
        prow[cptr[c] + slot] = int32_t(i);
        pval[cptr[c] + slot] = tval[e];
    }
}

// one workgroup per landmark c: M[c, :] = sum_n pnm[n, c] * pnm_hat[n, :], R[c] = sum_n pnm[n, c]
__global__ __launch_bounds__(256) void landmark_rows_kernel(const int L, const int64_t* __restrict__ cptr,
                                                            const int32_t* __restrict__ prow, const double* __restrict__ pval,
                                                            const int64_t* __restrict__ tptr, const int32_t* __restrict__ tcol,
                                                            const double* __restrict__ tnorm, double* __restrict__ M,
                                                            double* __restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* acc = reinterpret_cast<double*>(smem_raw);   // [L]
    __shared__ double rsum[4];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < L; k += 256) acc[k] = 0.0;
    __syncthreads();
    double r = 0.0;
    for (int64_t e = cptr[c] + w; e < cptr[c + 1]; e += 4) {
        const int64_t n = prow[e];
        const double v = pval[e];
        if (lane == 0) r += v;
        for (int64_t q = tptr[n] + lane; q < tptr[n + 1]; q += 64) atomicAdd(&acc[tcol[q]], v * tnorm[q]);
    }
    r = wave_sum_f64(r);
    if (lane == 0) rsum[w] = r;
    __syncthreads();
    for (int k = threadIdx.x; k < L; k += 256) M[size_t(c) * L + k] = acc[k];
    if (threadIdx.x == 0) R[c] = rsum[0] + rsum[1] + rsum[2] + rsum[3];
}

__global__ __launch_bounds__(256) void landmark_scale_kernel(double* __restrict__ M, const double* __restrict__ R,
                                                             const int L) {
    const int c = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= L) return;
    const double r = R[c];
    if (r != 0.0) M[size_t(c) * L + k] = M[size_t(c) * L + k] / r;
}

// ---- nearest landmark ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nearest_landmark_kernel(const T* __restrict__ X, const int64_t row0,
                                                               const int64_t nrows, const int d,
                                                               const int64_t* __restrict__ landmarks, const int L,
                                                               const int mode, const int is_f32,
                                                               int32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lm = reinterpret_cast<double*>(smem_raw);   // [LC][d] chunk of landmark rows (float64)
    constexpr int LC = 32;
    double* lnorm = lm + size_t(LC) * d;                // [LC]
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const bool live = i < nrows;
    const T* xi = X + (row0 + (live ? i : 0)) * d;
    double xn = 0.0;
    if (mode == 1) {
        for (int k = 0; k < d; ++k) {
            const double v = double(xi[k]);
            xn = fma(v, v, xn);
        }
    }
    double best = INFINITY;
    float best32 = INFINITY;
    int best_j = 0;
    for (int j0 = 0; j0 < L; j0 += LC) {
        __syncthreads();
        const int lc = (L - j0) < LC ? (L - j0) : LC;
        for (int e = threadIdx.x; e < lc * d; e += 256) {
            const int r = e / d, k = e % d;
            lm[r * d + k] = double(X[landmarks[j0 + r] * d + k]);
        }
        __syncthreads();
        if (mode == 1) {
            for (int r = threadIdx.x; r < lc; r += 256) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s = fma(lm[r * d + k], lm[r * d + k], s);
                lnorm[r] = s;
            }
            __syncthreads();
        }
        if (live) {
            for (int r = 0; r < lc; ++r) {
                if (mode == 0) {
                    // scipy cdist: float64 difference form, sequential in k
                    double s = 0.0;
                    for (int k = 0; k < d; ++k) {
                        const double diff = double(xi[k]) - lm[r * d + k];
                        s += diff * diff;
                    }
                    const double dist = sqrt(s);
                    if (dist < best) {
                        best = dist;
                        best_j = j0 + r;
                    }
                } else {
                    // sklearn euclidean_distances: ((-2 x.y) + |x|^2) + |y|^2 in float64, clamped, rounded to the
                    // input dtype, sqrt in that dtype (sklearn:metrics/pairwise.py:373-406, 582-596)
                    double dot = 0.0;
                    for (int k = 0; k < d; ++k) dot = fma(double(xi[k]), lm[r * d + k], dot);
                    double d2 = (-2.0 * dot + xn) + lnorm[r];
                    if (is_f32) {
                        float f = float(d2);
                        f = f > 0.f ? f : 0.f;
                        const float dist = sqrtf(f);
                        if (dist < best32) {
                            best32 = dist;
                            best_j = j0 + r;
                        }
                    } else {
                        d2 = d2 > 0.0 ? d2 : 0.0;
                        const double dist = sqrt(d2);
                        if (dist < best) {
                            best = dist;
                            best_j = j0 + r;
                        }
                    }
                }
            }
        }
    }
    if (live) out[i] = best_j;
}

}  // namespace

extern "C" int gt_landmark_build(gt_ctx* ctx, const int32_t* clusters, int32_t n_landmark, double* out_M, double* out_R,
                                 int32_t out_on_device, int64_t* out_transitions_nnz) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GraphState* g = ctx->graph;
    if (!g || !g->finished) GT_FAIL(ctx, GT_E_STATE, "gt_landmark_build: no finished kNN graph on this context");
    if (!clusters || n_landmark < 1) GT_FAIL(ctx, GT_E_ARG, "gt_landmark_build: bad cluster labels");
    if (g->p.kernel_symm == GT_SYMM_NONE)
        GT_FAIL(ctx, GT_E_ARG, "gt_landmark_build needs a symmetric kernel (kernel_symm != None)");
    if (size_t(n_landmark) * sizeof(double) > 150 * 1024)
        GT_FAIL(ctx, GT_E_LIMIT, "gt_landmark_build: n_landmark > 19200 is not supported");
    if (!ctx->landmark) ctx->landmark = new LandmarkState();
    LandmarkState* l = reinterpret_cast<LandmarkState*>(ctx->landmark);
    const int L = n_landmark;
    const int64_t nloc = g->nloc, n = ctx->n, nnz = g->nnz;
    l->L = L;
    l->nloc = nloc;
    StageSpan span(ctx, "landmark");
    GT_HIP(ctx, l->clusters.reserve(size_t(n) * sizeof(int32_t)));
    GT_HIP(ctx, hipMemcpyAsync(l->clusters.p, clusters, size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    GT_HIP(ctx, l->tlen.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, l->tptr.reserve(size_t(nloc + 1) * sizeof(int64_t)));
    GT_HIP(ctx, l->scol.reserve(size_t(nnz) * sizeof(int32_t)));
    GT_HIP(ctx, l->sval.reserve(size_t(nnz) * sizeof(double)));
    GT_HIP(ctx, l->bigrows.reserve(size_t(nloc) * sizeof(int32_t)));
    GT_HIP(ctx, l->bigcount.reserve(sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(l->bigcount.p, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(aggregate_rows_kernel, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc,
                       g->indptr.as<int64_t>(), g->indices.as<int32_t>(), g->Kdata.as<double>(), l->clusters.as<int32_t>(),
                       l->scol.as<int32_t>(), l->sval.as<double>(), l->tlen.as<int32_t>(), l->bigrows.as<int32_t>(),
                       l->bigcount.as<uint32_t>());
    GT_HIP(ctx, hipGetLastError());
    uint32_t nbig = 0;
    GT_HIP(ctx, hipMemcpyAsync(&nbig, l->bigcount.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nbig > 0) {
        const size_t lds = size_t(L) * sizeof(double);
        GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(aggregate_big_rows_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(aggregate_big_rows_kernel, dim3(nbig), dim3(256), lds, ctx->stream, l->bigrows.as<int32_t>(),
                           g->indptr.as<int64_t>(), g->indices.as<int32_t>(), g->Kdata.as<double>(),
                           l->clusters.as<int32_t>(), L, l->scol.as<int32_t>(), l->sval.as<double>(), l->tlen.as<int32_t>());
        GT_HIP(ctx, hipGetLastError());
    }
    GT_TRY(gt_exclusive_scan_i32(ctx, l->tlen.as<int32_t>(), nloc, l->tptr.as<int64_t>()));
    int64_t tnnz = 0;
    GT_HIP(ctx, hipMemcpyAsync(&tnnz, l->tptr.as<int64_t>() + nloc, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    l->tnnz = tnnz;
    GT_HIP(ctx, l->tcol.reserve(size_t(tnnz) * sizeof(int32_t)));
    GT_HIP(ctx, l->tval.reserve(size_t(tnnz) * sizeof(double)));
    GT_HIP(ctx, l->tnorm.reserve(size_t(tnnz) * sizeof(double)));
    GT_HIP(ctx, l->ccount.reserve(size_t(L) * sizeof(int32_t)));
    GT_HIP(ctx, l->ccur.reserve(size_t(L) * sizeof(int32_t)));
    GT_HIP(ctx, l->cptr.reserve(size_t(L + 1) * sizeof(int64_t)));
    GT_HIP(ctx, l->prow.reserve(size_t(tnnz) * sizeof(int32_t)));
    GT_HIP(ctx, l->pval.reserve(size_t(tnnz) * sizeof(double)));
    GT_HIP(ctx, l->M.reserve(size_t(L) * L * sizeof(double)));
    GT_HIP(ctx, l->R.reserve(size_t(L) * sizeof(double)));
    GT_HIP(ctx, hipMemsetAsync(l->ccount.p, 0, size_t(L) * sizeof(int32_t), ctx->stream));
    GT_HIP(ctx, hipMemsetAsync(l->ccur.p, 0, size_t(L) * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(compact_transitions_kernel, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc,
                       g->indptr.as<int64_t>(), l->tlen.as<int32_t>(), l->tptr.as<int64_t>(), l->scol.as<int32_t>(),
                       l->sval.as<double>(), l->tcol.as<int32_t>(), l->tval.as<double>(), l->tnorm.as<double>(),
                       l->ccount.as<int32_t>());
    GT_TRY(gt_exclusive_scan_i32(ctx, l->ccount.as<int32_t>(), L, l->cptr.as<int64_t>()));
    hipLaunchKernelGGL(scatter_by_landmark_kernel, dim3((unsigned)ceil_div64(nloc, 4)), dim3(256), 0, ctx->stream, nloc,
                       l->tptr.as<int64_t>(), l->tcol.as<int32_t>(), l->tval.as<double>(), l->cptr.as<int64_t>(),
                       l->ccur.as<int32_t>(), l->prow.as<int32_t>(), l->pval.as<double>());
    const size_t lds = size_t(L) * sizeof(double);
    GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(landmark_rows_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(landmark_rows_kernel, dim3((unsigned)L), dim3(256), lds, ctx->stream, L, l->cptr.as<int64_t>(),
                       l->prow.as<int32_t>(), l->pval.as<double>(), l->tptr.as<int64_t>(), l->tcol.as<int32_t>(),
                       l->tnorm.as<double>(), l->M.as<double>(), l->R.as<double>());
    GT_HIP(ctx, hipGetLastError());
    const hipMemcpyKind kind = out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (out_M) GT_HIP(ctx, hipMemcpyAsync(out_M, l->M.p, size_t(L) * L * sizeof(double), kind, ctx->stream));
    if (out_R) GT_HIP(ctx, hipMemcpyAsync(out_R, l->R.p, size_t(L) * sizeof(double), kind, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (out_transitions_nnz) *out_transitions_nnz = tnnz;
    return GT_OK;
}

extern "C" int gt_landmark_scale(gt_ctx* ctx, double* M_inout, const double* R, int32_t n_landmark, int32_t on_device) {
    if (!ctx || !M_inout || !R || n_landmark < 1) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    const int L = n_landmark;
    DevBuf dm, dr;
    DevBufScope scratch{&dm, &dr};
    double* Md = M_inout;
    const double* Rd = R;
    if (!on_device) {
        GT_HIP(ctx, dm.reserve(size_t(L) * L * sizeof(double)));
        GT_HIP(ctx, dr.reserve(size_t(L) * sizeof(double)));
        GT_HIP(ctx, hipMemcpyAsync(dm.p, M_inout, size_t(L) * L * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        GT_HIP(ctx, hipMemcpyAsync(dr.p, R, size_t(L) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        Md = dm.as<double>();
        Rd = dr.as<double>();
    }
    hipLaunchKernelGGL(landmark_scale_kernel, dim3((unsigned)ceil_div64(L, 256), (unsigned)L), dim3(256), 0, ctx->stream, Md,
                       Rd, L);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !on_device)
        e = hipMemcpyAsync(M_inout, Md, size_t(L) * L * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->set_error(std::string("gt_landmark_scale: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    return GT_OK;
}

extern "C" int gt_landmark_fetch_transitions(gt_ctx* ctx, double* data, int32_t* indices, int64_t* indptr,
                                             int32_t on_device) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    LandmarkState* l = reinterpret_cast<LandmarkState*>(ctx->landmark);
    if (!l || l->nloc == 0) GT_FAIL(ctx, GT_E_STATE, "gt_landmark_fetch_transitions: call gt_landmark_build first");
    if (!on_device) {
        if (data) GT_TRY(gt_copy_to_host(ctx, data, l->tnorm.p, size_t(l->tnnz) * sizeof(double)));
        if (indices) GT_TRY(gt_copy_to_host(ctx, indices, l->tcol.p, size_t(l->tnnz) * sizeof(int32_t)));
        if (indptr) GT_TRY(gt_copy_to_host(ctx, indptr, l->tptr.p, size_t(l->nloc + 1) * sizeof(int64_t)));
        return GT_OK;
    }
    const hipMemcpyKind kind = hipMemcpyDeviceToDevice;
    if (data && l->tnnz > 0) GT_HIP(ctx, hipMemcpyAsync(data, l->tnorm.p, size_t(l->tnnz) * sizeof(double), kind, ctx->stream));
    if (indices && l->tnnz > 0)
        GT_HIP(ctx, hipMemcpyAsync(indices, l->tcol.p, size_t(l->tnnz) * sizeof(int32_t), kind, ctx->stream));
    if (indptr) GT_HIP(ctx, hipMemcpyAsync(indptr, l->tptr.p, size_t(l->nloc + 1) * sizeof(int64_t), kind, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

// first index among the nearest that tie: labels[i] = min { idx[i][c] : dist[i][c] == dist[i][0] } (tables sorted by distance)
__global__ __launch_bounds__(256) void first_nearest_kernel(const int64_t* __restrict__ idx, const double* __restrict__ dist,
                                                            const int64_t m, const int k, int64_t* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= m) return;
    const double d0 = dist[i * k];
    int64_t best = idx[i * k];
    for (int c = 1; c < k; ++c)
        if (dist[i * k + c] == d0 && idx[i * k + c] < best) best = idx[i * k + c];
    out[i] = best;
}

extern "C" int gt_knn_first_nearest(gt_ctx* ctx, const void* Y, int64_t m, int32_t y_on_device, int32_t k, int64_t* out_labels,
                                    uint32_t* flags) {
    if (!ctx) return GT_E_ARG;
    if (!Y || m <= 0 || k < 1 || !out_labels) GT_FAIL(ctx, GT_E_ARG, "gt_knn_first_nearest: bad arguments");
    GT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d_idx, d_dist, d_out;
    DevBufScope scratch{&d_idx, &d_dist, &d_out};   // (released on every exit, the early GT_HIP returns included)
    GT_HIP(ctx, d_idx.reserve(size_t(m) * k * sizeof(int64_t)));
    GT_HIP(ctx, d_dist.reserve(size_t(m) * k * sizeof(double)));
    GT_HIP(ctx, d_out.reserve(size_t(m) * sizeof(int64_t)));
    int rc = gt_knn_search(ctx, 0, 0, Y, m, y_on_device, k, d_idx.as<int64_t>(), d_dist.as<double>(), 1, flags);
    if (rc == GT_OK) {
        hipLaunchKernelGGL(first_nearest_kernel, dim3((unsigned)ceil_div64(m, 256)), dim3(256), 0, ctx->stream, d_idx.as<int64_t>(),
                           d_dist.as<double>(), m, int(k), d_out.as<int64_t>());
        if (hipGetLastError() != hipSuccess) {
            ctx->set_error("gt_knn_first_nearest: launch failed");
            rc = GT_E_HIP;
        }
    }
    if (rc == GT_OK) rc = gt_copy_to_host(ctx, out_labels, d_out.p, size_t(m) * sizeof(int64_t));
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (rc == GT_OK && es != hipSuccess) {
        ctx->set_error(std::string("gt_knn_first_nearest: ") + hipGetErrorString(es));
        rc = GT_E_HIP;
    }
    return rc;
}

extern "C" int gt_nearest_landmark(gt_ctx* ctx, int64_t row0, int64_t row1, const int64_t* landmarks, int32_t n_landmark,
                                   int32_t mode, int32_t* out_clusters) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->n <= 0 || !ctx->X) GT_FAIL(ctx, GT_E_STATE, "gt_nearest_landmark: no points bound");
    if (!landmarks || !out_clusters || n_landmark < 1 || row0 < 0 || row1 > ctx->n || row1 <= row0)
        GT_FAIL(ctx, GT_E_ARG, "gt_nearest_landmark: bad arguments");
    for (int j = 0; j < n_landmark; ++j)
        if (landmarks[j] < 0 || landmarks[j] >= ctx->n) GT_FAIL(ctx, GT_E_ARG, "gt_nearest_landmark: landmark out of range");
    const int64_t nrows = row1 - row0;
    DevBuf lmk, out;
    DevBufScope scratch{&lmk, &out};
    GT_HIP(ctx, lmk.reserve(size_t(n_landmark) * sizeof(int64_t)));
    GT_HIP(ctx, out.reserve(size_t(nrows) * sizeof(int32_t)));
    std::vector<int64_t> lm_ctx;
    if (ctx->presorted) {
        // renumbered points (gt_points_cell_sort): [row0, row1) are rows of the context, the landmarks are the CALLER's row
        // numbers - looked up through the inverse of the renumbering
        std::vector<int32_t> vperm(size_t(ctx->n)), inv(size_t(ctx->n));
        GT_HIP(ctx, hipMemcpyAsync(vperm.data(), ctx->vperm.p, size_t(ctx->n) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int64_t v = 0; v < ctx->n; ++v) inv[size_t(vperm[size_t(v)])] = int32_t(v);
        lm_ctx.resize(size_t(n_landmark));
        for (int j = 0; j < n_landmark; ++j) lm_ctx[size_t(j)] = inv[size_t(landmarks[j])];
        landmarks = lm_ctx.data();
    }
    GT_HIP(ctx, hipMemcpyAsync(lmk.p, landmarks, size_t(n_landmark) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = (size_t(32) * ctx->d + 32) * sizeof(double);
    {
        StageSpan span(ctx, "landmark_assign");
        if (ctx->dtype == GT_F32) {
            auto kern = nearest_landmark_kernel<float>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(nrows, 256)), dim3(256), lds, ctx->stream,
                               (const float*)ctx->X, row0, nrows, ctx->d, lmk.as<int64_t>(), n_landmark, mode, 1,
                               out.as<int32_t>());
        } else {
            auto kern = nearest_landmark_kernel<double>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(nrows, 256)), dim3(256), lds, ctx->stream,
                               (const double*)ctx->X, row0, nrows, ctx->d, lmk.as<int64_t>(), n_landmark, mode, 0,
                               out.as<int32_t>());
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(out_clusters, out.p, size_t(nrows) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->set_error(std::string("gt_nearest_landmark: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    return GT_OK;
}
