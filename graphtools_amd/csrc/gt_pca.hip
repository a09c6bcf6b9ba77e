// Tall-matrix products for the PCA pre-reduction in front of the graph build (reference: Data._reduce_data,
// graphtools/base.py:227-294 -> sklearn PCA(svd_solver="randomized"): a randomized range finder with power iterations).
//
// The data matrix X (n x d float32, n up to millions, d up to thousands) stays on the device; every step of the
// randomized SVD is one of three products with a thin matrix of at most KP = 128 columns:
//   gt_pca_matmul    Y = S W - 1 sub^T      S = X (n x d) or a thin buffer (n x KP), W (rows(S) x k) from the host
//   gt_pca_tmatmul   Z = X^T Y  (d x k)     + the column sums of Y (centring correction on the host)
//   gt_pca_gram      C = Y^T Y  (k x k)     float64 accumulation
// The thin factors (d x k, k x k) go to the host, where the reference's own LAPACK calls (QR, eigh) finish them
// (graphtools_amd/_pca.py mirrors sklearn.utils.extmath.randomized_svd step by step).
// Arithmetic: v_mfma_f32_32x32x2_f32 - exact float32 products, float32 accumulation, like the sgemm sklearn runs on
// float32 data; partial sums over row slabs are combined in float64.  Roofline: 2 n d k flop per product against the
// 157 TF float32 MFMA peak (k >= 32: compute-bound; the 4 n d bytes of X stream from HBM once per product).
#include "gt_common.h"
#include "gt_device.h"
#include "gt_hostcopy.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int KP = 128;        // padded thin width (columns of W / Y)
constexpr int CH = 64;         // inner-dimension chunk per LDS stage
constexpr int SLAB = 2048;     // rows per partial sum of the transposed products

struct PcaState {
    const float* X = nullptr;   // n x d, device
    DevBuf X_own;
    int64_t n = 0;
    int d = 0;
    DevBuf Y[2];                // thin buffers n x KP float32
    DevBuf W, sub;              // device copies of the host factors (rows x KP float32, zero padded), [KP]
    DevBuf part, out64;         // partial sums of the transposed products, float64 results
};

// ---- Y = S W - 1 sub^T ------------------------------------------------------------------------------------------------
// One workgroup = 128 rows (4 waves x one 32-row tile), all KP columns (4 tiles of 32 per wave).  The inner dimension is
// walked in chunks of 64: the chunk of W sits in LDS (64 x KP floats), the rows of S come straight from global memory -
// lane (i, h) holds the 32 consecutive values S[row0 + i][c0 + 32 h ..] and feeds them as the k-pair (s, 32 + s) of the
// 32 x 32 x 2 MFMA, so every lane reads 128 contiguous bytes per chunk.
__global__ __launch_bounds__(256) void tall_matmul_kernel(const float* __restrict__ S, const int64_t n, const int ds,
                                                          const int64_t lds, const float* __restrict__ W,
                                                          const float* __restrict__ sub, float* __restrict__ Y) {
    __shared__ __attribute__((aligned(16))) float wl[CH * KP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
    const int64_t row0 = int64_t(blockIdx.x) * 128 + w * 32;
    const int64_t row = row0 + i < n ? row0 + i : n - 1;
    const float* srow = S + row * lds;
    f32x16 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const int nch = (ds + CH - 1) / CH;
    const bool vec = (lds % 4 == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0);
    for (int c = 0; c < nch; ++c) {
        const int c0 = c * CH;
        __syncthreads();
        // W is zero padded to a multiple of CH rows
        for (int f = tid; f < CH * KP / 4; f += 256)
            reinterpret_cast<float4*>(wl)[f] = reinterpret_cast<const float4*>(W + size_t(c0) * KP)[f];
        float a[32];
        const int k0 = c0 + 32 * h;
        if (vec && k0 + 32 <= ds) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(srow + k0 + 4 * q);
                a[4 * q + 0] = v.x;
                a[4 * q + 1] = v.y;
                a[4 * q + 2] = v.z;
                a[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) a[s] = (k0 + s < ds) ? srow[k0 + s] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; ++s) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float b = wl[(32 * h + s) * KP + 32 * ct + i];
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b, acc[ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float sb = sub ? sub[32 * ct + i] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (rr < n) Y[rr * KP + 32 * ct + i] = acc[ct][r] - sb;
        }
    }
}

// ---- partial Z = X^T Y over a slab of rows ---------------------------------------------------------------------------
// grid (feature blocks of 128, row slabs of SLAB).  Wave w owns features f0 + 32 w .. + 32 and all KP columns.  Per chunk
// of 64 rows X[64][128 features] and Y[64][KP] sit in LDS; A[i = feature][k = row] is read column-wise from the X image.
__global__ __launch_bounds__(256) void tall_tmatmul_kernel(const float* __restrict__ X, const int64_t n, const int d,
                                                           const int64_t ldx, const float* __restrict__ Y,
                                                           float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xl = sm;                 // [CH][128 + 4]
    float* yl = sm + CH * 132;      // [CH][KP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
    const int f0 = blockIdx.x * 128;
    const int64_t r_begin = int64_t(blockIdx.y) * SLAB;
    const int64_t r_end = r_begin + SLAB < n ? r_begin + SLAB : n;
    f32x16 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += CH) {
        __syncthreads();
        for (int f = tid; f < CH * 128; f += 256) {
            const int rr = f >> 7, cc = f & 127;
            const int64_t r = r0 + rr;
            xl[rr * 132 + cc] = (r < r_end && f0 + cc < d) ? X[r * ldx + f0 + cc] : 0.f;
        }
        for (int f = tid; f < CH * KP / 4; f += 256) {
            const int rr = f / (KP / 4);
            const int64_t r = r0 + rr;
            reinterpret_cast<float4*>(yl)[f] = (r < r_end) ? reinterpret_cast<const float4*>(Y + r * KP)[f % (KP / 4)]
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const float a = xl[(32 * h + s) * 132 + 32 * w + i];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float b = yl[(32 * h + s) * KP + 32 * ct + i];
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ct], 0, 0, 0);
            }
        }
    }
    float* po = part + (size_t(blockIdx.y) * gridDim.x * 128 + size_t(blockIdx.x) * 128) * KP;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
            po[size_t(fr) * KP + 32 * ct + i] = acc[ct][r];
        }
}

// out[f][c] = sum over slabs of part[slab][f][c] in float64 (fixed order: deterministic)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ part, const int64_t per_slab,
                                                           const int nslab, double* __restrict__ out) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= per_slab) return;
    double s = 0.0;
    for (int b = 0; b < nslab; ++b) s += double(part[size_t(b) * per_slab + e]);
    out[e] = s;
}

// column sums of a thin buffer: partial per slab (float64), reduced by reduce_cols_kernel
__global__ __launch_bounds__(128) void colsum_kernel(const float* __restrict__ Y, const int64_t n, double* __restrict__ part) {
    const int64_t r_begin = int64_t(blockIdx.x) * SLAB;
    const int64_t r_end = r_begin + SLAB < n ? r_begin + SLAB : n;
    double s = 0.0;
    for (int64_t r = r_begin; r < r_end; ++r) s += double(Y[r * KP + threadIdx.x]);
    part[size_t(blockIdx.x) * KP + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void reduce_cols_kernel(const double* __restrict__ part, const int64_t per, const int nb,
                                                          double* __restrict__ out) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= per) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[size_t(b) * per + e];
    out[e] = s;
}

// ---- C = Y^T Y, float64 accumulation ---------------------------------------------------------------------------------
// One workgroup per slab; thread (ta, tb) of a 16 x 16 grid owns the 8 x 8 block C[8 ta .. +8][8 tb .. +8]; rows are
// staged through LDS 32 at a time.
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ Y, const int64_t n, double* __restrict__ part) {
    __shared__ float yl[32][KP + 1];
    const int tid = threadIdx.x, ta = tid >> 4, tb = tid & 15;
    const int64_t r_begin = int64_t(blockIdx.x) * SLAB;
    const int64_t r_end = r_begin + SLAB < n ? r_begin + SLAB : n;
    double c[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) c[a][b] = 0.0;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
        __syncthreads();
        for (int f = tid; f < 32 * KP; f += 256) {
            const int rr = f >> 7, cc = f & 127;
            yl[rr][cc] = (r0 + rr < r_end) ? Y[(r0 + rr) * KP + cc] : 0.f;
        }
        __syncthreads();
        for (int rr = 0; rr < 32; ++rr) {
            double va[8], vb[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) va[a] = double(yl[rr][8 * ta + a]);
#pragma unroll
            for (int b = 0; b < 8; ++b) vb[b] = double(yl[rr][8 * tb + b]);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) c[a][b] = fma(va[a], vb[b], c[a][b]);
        }
    }
    double* po = part + size_t(blockIdx.x) * KP * KP;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) po[size_t(8 * ta + a) * KP + 8 * tb + b] = c[a][b];
}

// column sums and sums of squares of X (float64): mean and total variance
__global__ __launch_bounds__(256) void col_moments_kernel(const float* __restrict__ X, const int64_t n, const int d,
                                                          const int64_t ldx, const int64_t rows_per_block,
                                                          double* __restrict__ part) {
    const int c = int(blockIdx.x) * 256 + threadIdx.x;
    const int64_t r0 = int64_t(blockIdx.y) * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    if (c >= d) return;
    double s = 0.0, q = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        const double v = double(X[r * ldx + c]);
        s += v;
        q = fma(v, v, q);
    }
    part[(size_t(blockIdx.y) * 2 + 0) * d + c] = s;
    part[(size_t(blockIdx.y) * 2 + 1) * d + c] = q;
}

PcaState* state_of(gt_ctx* ctx) { return reinterpret_cast<PcaState*>(ctx->pca); }

int upload_factor(gt_ctx* ctx, PcaState* p, const double* W, int rows, int k, const double* sub) {
    const int rows_pad = (rows + CH - 1) / CH * CH;
    std::vector<float> wf(size_t(rows_pad) * KP, 0.f);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < k; ++c) wf[size_t(r) * KP + c] = float(W[size_t(r) * k + c]);
    GT_HIP(ctx, p->W.reserve(wf.size() * sizeof(float)));
    GT_HIP(ctx, hipMemcpyAsync(p->W.p, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    float sf[KP];
    for (int c = 0; c < KP; ++c) sf[c] = (sub && c < k) ? float(sub[c]) : 0.f;
    GT_HIP(ctx, p->sub.reserve(KP * sizeof(float)));
    GT_HIP(ctx, hipMemcpyAsync(p->sub.p, sf, sizeof(sf), hipMemcpyHostToDevice, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host staging buffers go out of scope
    return GT_OK;
}

}  // namespace

void gt_free_pca_state(gt_ctx* ctx) {
    PcaState* p = state_of(ctx);
    if (!p) return;
    for (DevBuf* b : {&p->X_own, &p->Y[0], &p->Y[1], &p->W, &p->sub, &p->part, &p->out64}) b->release();
    delete p;
    ctx->pca = nullptr;
}

extern "C" int gt_pca_begin(gt_ctx* ctx, const float* X, int64_t n, int32_t d, int32_t x_on_device, double* mean_out,
                            double* sumsq_centered_out) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (!X || n < 2 || d < 1) GT_FAIL(ctx, GT_E_ARG, "gt_pca_begin: need a matrix with at least 2 rows");
    gt_free_pca_state(ctx);
    PcaState* p = new PcaState();
    ctx->pca = p;
    p->n = n;
    p->d = d;
    if (x_on_device) {
        p->X = X;
    } else {
        GT_HIP(ctx, p->X_own.reserve(size_t(n) * d * sizeof(float)));
        GT_TRY(gt_copy_from_host(ctx, p->X_own.p, X, size_t(n) * d * sizeof(float)));
        p->X = p->X_own.as<float>();
    }
    GT_HIP(ctx, p->Y[0].reserve(size_t(n) * KP * sizeof(float)));
    GT_HIP(ctx, p->Y[1].reserve(size_t(n) * KP * sizeof(float)));
    // column sums / sums of squares in float64, a few thousand rows per partial
    const int64_t rpb = 4096;
    const int nby = int(ceil_div64(n, rpb));
    GT_HIP(ctx, p->out64.reserve(std::max<size_t>(size_t(nby) * 2 * d, size_t(KP) * KP) * sizeof(double) + size_t(2) * d * sizeof(double)));
    double* partm = p->out64.as<double>();
    hipLaunchKernelGGL(col_moments_kernel, dim3((unsigned)ceil_div64(d, 256), (unsigned)nby), dim3(256), 0, ctx->stream, p->X,
                       n, d, int64_t(d), rpb, partm);
    GT_HIP(ctx, hipGetLastError());
    std::vector<double> host(size_t(nby) * 2 * d);
    GT_HIP(ctx, hipMemcpyAsync(host.data(), partm, host.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < d; ++c) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < nby; ++b) {
            s += host[(size_t(b) * 2 + 0) * d + c];
            q += host[(size_t(b) * 2 + 1) * d + c];
        }
        const double mean = s / double(n);
        if (mean_out) mean_out[c] = mean;
        if (sumsq_centered_out) sumsq_centered_out[c] = std::max(0.0, q - s * mean);
    }
    return GT_OK;
}

extern "C" int gt_pca_matmul(gt_ctx* ctx, int32_t src, const double* W, int32_t wrows, int32_t k, const double* sub,
                             int32_t dst) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    PcaState* p = state_of(ctx);
    if (!p) GT_FAIL(ctx, GT_E_STATE, "gt_pca_matmul: call gt_pca_begin first");
    if (!W || k < 1 || k > KP || src < 0 || src > 2 || dst < 1 || dst > 2 || src == dst)
        GT_FAIL(ctx, GT_E_ARG, "gt_pca_matmul: bad arguments (at most 128 columns; source and destination must differ)");
    if (wrows != (src == 0 ? p->d : wrows) || wrows < 1 || (src != 0 && wrows > KP))
        GT_FAIL(ctx, GT_E_ARG, "gt_pca_matmul: the factor must have one row per column of the source");
    GT_TRY(upload_factor(ctx, p, W, wrows, k, sub));
    const float* S = src == 0 ? p->X : p->Y[src - 1].as<float>();
    const int64_t lds = src == 0 ? int64_t(p->d) : int64_t(KP);
    StageSpan span(ctx, "pca_matmul");
    hipLaunchKernelGGL(tall_matmul_kernel, dim3((unsigned)ceil_div64(p->n, 128)), dim3(256), 0, ctx->stream, S, p->n,
                       wrows, lds, p->W.as<float>(), sub ? p->sub.as<float>() : nullptr,
                       p->Y[dst - 1].as<float>());
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

extern "C" int gt_pca_tmatmul(gt_ctx* ctx, int32_t ybuf, int32_t k, double* out, double* colsum) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    PcaState* p = state_of(ctx);
    if (!p) GT_FAIL(ctx, GT_E_STATE, "gt_pca_tmatmul: call gt_pca_begin first");
    if (!out || k < 1 || k > KP || ybuf < 1 || ybuf > 2) GT_FAIL(ctx, GT_E_ARG, "gt_pca_tmatmul: bad arguments");
    const int nfb = (p->d + 127) / 128;
    const int nslab = int(ceil_div64(p->n, SLAB));
    const size_t per_slab = size_t(nfb) * 128 * KP;
    GT_HIP(ctx, p->part.reserve(std::max(per_slab * nslab * sizeof(float), size_t(nslab) * KP * KP * sizeof(double))));
    GT_HIP(ctx, p->out64.reserve(std::max(per_slab, size_t(KP) * KP) * sizeof(double) + KP * sizeof(double)));
    const float* Yb = p->Y[ybuf - 1].as<float>();
    {
        StageSpan span(ctx, "pca_tmatmul");
        const size_t lds = size_t(CH) * 132 * 4 + size_t(CH) * KP * 4;
        GT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tall_tmatmul_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(tall_tmatmul_kernel, dim3((unsigned)nfb, (unsigned)nslab), dim3(256), lds, ctx->stream, p->X, p->n,
                           p->d, int64_t(p->d), Yb, p->part.as<float>());
        GT_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div64(int64_t(per_slab), 256)), dim3(256), 0, ctx->stream,
                           p->part.as<float>(), int64_t(per_slab), nslab, p->out64.as<double>());
        GT_HIP(ctx, hipGetLastError());
    }
    std::vector<double> host(per_slab);
    GT_HIP(ctx, hipMemcpyAsync(host.data(), p->out64.p, per_slab * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < p->d; ++f)
        for (int c = 0; c < k; ++c) out[size_t(f) * k + c] = host[size_t(f) * KP + c];
    if (colsum) {
        double* cpart = reinterpret_cast<double*>(p->part.p);
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nslab), dim3(128), 0, ctx->stream, Yb, p->n, cpart);
        GT_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_cols_kernel, dim3(1), dim3(256), 0, ctx->stream, cpart, int64_t(KP), nslab, p->out64.as<double>());
        GT_HIP(ctx, hipGetLastError());
        double cs[KP];
        GT_HIP(ctx, hipMemcpyAsync(cs, p->out64.p, sizeof(cs), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int c = 0; c < k; ++c) colsum[c] = cs[c];
    }
    return GT_OK;
}

extern "C" int gt_pca_gram(gt_ctx* ctx, int32_t ybuf, int32_t k, double* out) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    PcaState* p = state_of(ctx);
    if (!p) GT_FAIL(ctx, GT_E_STATE, "gt_pca_gram: call gt_pca_begin first");
    if (!out || k < 1 || k > KP || ybuf < 1 || ybuf > 2) GT_FAIL(ctx, GT_E_ARG, "gt_pca_gram: bad arguments");
    const int nslab = int(ceil_div64(p->n, SLAB));
    GT_HIP(ctx, p->part.reserve(size_t(nslab) * KP * KP * sizeof(double)));
    GT_HIP(ctx, p->out64.reserve(size_t(KP) * KP * sizeof(double)));
    {
        StageSpan span(ctx, "pca_gram");
        hipLaunchKernelGGL(gram_kernel, dim3((unsigned)nslab), dim3(256), 0, ctx->stream, p->Y[ybuf - 1].as<float>(), p->n,
                           reinterpret_cast<double*>(p->part.p));
        GT_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_cols_kernel, dim3((unsigned)ceil_div64(KP * KP, 256)), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const double*>(p->part.p), int64_t(KP) * KP, nslab, p->out64.as<double>());
        GT_HIP(ctx, hipGetLastError());
    }
    std::vector<double> host(size_t(KP) * KP);
    GT_HIP(ctx, hipMemcpyAsync(host.data(), p->out64.p, host.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) out[size_t(a) * k + b] = host[size_t(a) * KP + b];
    return GT_OK;
}

// rows of a thin buffer, first k columns, float32 row-major [n][k]; on_device: `out` is device memory
extern "C" int gt_pca_fetch(gt_ctx* ctx, int32_t ybuf, int32_t k, float* out, int32_t on_device) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    PcaState* p = state_of(ctx);
    if (!p) GT_FAIL(ctx, GT_E_STATE, "gt_pca_fetch: call gt_pca_begin first");
    if (!out || k < 1 || k > KP || ybuf < 1 || ybuf > 2) GT_FAIL(ctx, GT_E_ARG, "gt_pca_fetch: bad arguments");
    GT_HIP(ctx, hipMemcpy2DAsync(out, size_t(k) * sizeof(float), p->Y[ybuf - 1].p, size_t(KP) * sizeof(float),
                                 size_t(k) * sizeof(float), size_t(p->n), on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                 ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

extern "C" int gt_pca_end(gt_ctx* ctx) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    gt_free_pca_state(ctx);
    return GT_OK;
}
