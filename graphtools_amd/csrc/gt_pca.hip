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
// Thin matrices (W, the thin buffers) are stored with their columns PERMUTED: column c = 32 ct + i sits at position
// 4 i + ct, so that the four values a lane feeds to its four column tiles (ct = 0..3) are 16 contiguous bytes - one
// ds_read_b128 per MFMA step instead of four ds_read_b32, and a lane's four results of a row leave as one float4.
__host__ __device__ inline int cperm(int c) { return 4 * (c & 31) + (c >> 5); }
constexpr int CH = 64;         // inner-dimension chunk per LDS stage
constexpr int SLAB = 2048;     // rows per partial sum of the transposed products

struct PcaState {
    const float* X = nullptr;   // n rows of ldx floats on the device (ldx = d rounded up to a multiple of 64, zero filled)
    DevBuf X_own;
    int64_t n = 0;
    int d = 0;
    int64_t ldx = 0;
    DevBuf Y[2];                // thin buffers n x KP float32
    DevBuf W, sub;              // device copies of the host factors (rows x KP float32, zero padded), [KP]
    DevBuf part, out64;         // partial sums of the transposed products, float64 results
};

// ---- Y = S W - 1 sub^T ------------------------------------------------------------------------------------------------
// One workgroup = 128 rows (4 waves x one 32-row tile), all KP columns (4 tiles of 32 per wave).  The inner dimension is
// walked in chunks of 64: the chunk of W sits in LDS (64 x KP floats), the rows of S come straight from global memory -
// lane (i, h) holds the 32 consecutive values S[row0 + i][c0 + 32 h ..] and feeds them as the k-pair (s, 32 + s) of the
// 32 x 32 x 2 MFMA, so every lane reads 128 contiguous bytes per chunk.
// S: rows of `lds` floats, lds a multiple of CH = 64 and 16-byte aligned (gt_pca_begin pads X when it has to), nch chunks.
__global__ __launch_bounds__(256) void tall_matmul_kernel(const float* __restrict__ S, const int64_t n, const int nch,
                                                          const int64_t lds, const float* __restrict__ W,
                                                          const float* __restrict__ sub, float* __restrict__ Y) {
    // two LDS images of the W chunk: the next chunk (and the next slice of the rows) is fetched into registers while the
    // MFMAs of the current one run, and parked behind them - one barrier per chunk
    __shared__ __attribute__((aligned(16))) float wl[2][CH * KP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
    const int64_t row0 = int64_t(blockIdx.x) * 128 + w * 32;
    const int64_t row = row0 + i < n ? row0 + i : n - 1;
    const float4* srow = reinterpret_cast<const float4*>(S + row * lds + 32 * h);
    const float4* wsrc = reinterpret_cast<const float4*>(W) + tid;
    f32x16 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    float4 a4[8], an4[8], wst[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        wst[q] = wsrc[256 * q];
        a4[q] = srow[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) reinterpret_cast<float4*>(wl[0])[tid + 256 * q] = wst[q];
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const float* wc = wl[c & 1];
        const int cn = c + 1 < nch ? c + 1 : c;   // (the last round re-reads its own chunk: no branch around the loads)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            wst[q] = wsrc[size_t(cn) * (CH * KP / 4) + 256 * q];
            an4[q] = srow[size_t(cn) * (CH / 4) + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float av[4] = {a4[q].x, a4[q].y, a4[q].z, a4[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 b = *reinterpret_cast<const float4*>(wc + (32 * h + 4 * q + e) * KP + 4 * i);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.w, acc[3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            reinterpret_cast<float4*>(wl[(c + 1) & 1])[tid + 256 * q] = wst[q];
            a4[q] = an4[q];
        }
        __syncthreads();
    }
    const float4 sb = sub ? *reinterpret_cast<const float4*>(sub + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (rr < n)
            *reinterpret_cast<float4*>(Y + rr * KP + 4 * i) =
                make_float4(acc[0][r] - sb.x, acc[1][r] - sb.y, acc[2][r] - sb.z, acc[3][r] - sb.w);
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
    // the next chunk of rows is fetched into registers while the MFMAs of the current one run (thread t moves the 16-byte
    // pieces t, t + 256, ... of both images)
    const bool vec = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (f0 + 128 <= d);
    float4 xs_[8], ys_[8];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int f = tid + 256 * q;          // piece: row f / 32, columns 4 (f % 32) ..
            const int rr = f >> 5, c4 = (f & 31) * 4;
            const int64_t r = r0 + rr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < r_end) {
                if (vec) {
                    v = *reinterpret_cast<const float4*>(X + r * ldx + f0 + c4);
                } else {
                    const float* xr = X + r * ldx;
                    v.x = (f0 + c4 + 0 < d) ? xr[f0 + c4 + 0] : 0.f;
                    v.y = (f0 + c4 + 1 < d) ? xr[f0 + c4 + 1] : 0.f;
                    v.z = (f0 + c4 + 2 < d) ? xr[f0 + c4 + 2] : 0.f;
                    v.w = (f0 + c4 + 3 < d) ? xr[f0 + c4 + 3] : 0.f;
                }
            }
            xs_[q] = v;
            ys_[q] = (r < r_end) ? reinterpret_cast<const float4*>(Y + r * KP)[f & 31] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int f = tid + 256 * q;
            const int rr = f >> 5, c4 = (f & 31) * 4;
            *reinterpret_cast<float4*>(xl + rr * 132 + c4) = xs_[q];
            *reinterpret_cast<float4*>(yl + rr * KP + c4) = ys_[q];
        }
    };
    fetch(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += CH) {
        __syncthreads();   // everyone is done with the previous images
        park();
        __syncthreads();
        if (r0 + CH < r_end) fetch(r0 + CH);
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const float a = xl[(32 * h + s) * 132 + 32 * w + i];
            const float4 b = *reinterpret_cast<const float4*>(yl + (32 * h + s) * KP + 4 * i);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.w, acc[3], 0, 0, 0);
        }
    }
    // columns stay in the permuted order of the thin buffer (the host unpermutes)
    float* po = part + (size_t(blockIdx.y) * gridDim.x * 128 + size_t(blockIdx.x) * 128) * KP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int fr = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
        *reinterpret_cast<float4*>(po + size_t(fr) * KP + 4 * i) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
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

// first k columns of a thin buffer as a dense [n][k] matrix
__global__ __launch_bounds__(256) void compact_cols_kernel(const float* __restrict__ Y, const int64_t n, const int k,
                                                           float* __restrict__ out) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= n * k) return;
    out[e] = Y[(e / k) * KP + cperm(int(e % k))];
}

// rows padded with zeros to a multiple of 64 floats (and 16-byte aligned): the products then run on whole chunks only
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ X, const int64_t n, const int d,
                                                       const int64_t ldx, float* __restrict__ out) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= n * ldx) return;
    const int64_t r = e / ldx;
    const int c = int(e % ldx);
    out[e] = c < d ? X[r * d + c] : 0.f;
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

// thin_rows: the factor multiplies a thin buffer from the right - its rows follow the buffer's permuted column order
int upload_factor(gt_ctx* ctx, PcaState* p, const double* W, int rows, int k, const double* sub, bool thin_rows) {
    const int rows_pad = ((thin_rows ? KP : rows) + CH - 1) / CH * CH;
    std::vector<float> wf(size_t(rows_pad) * KP, 0.f);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < k; ++c) wf[size_t(thin_rows ? cperm(r) : r) * KP + cperm(c)] = float(W[size_t(r) * k + c]);
    GT_HIP(ctx, p->W.reserve(wf.size() * sizeof(float)));
    GT_HIP(ctx, hipMemcpyAsync(p->W.p, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    float sf[KP];
    for (int c = 0; c < KP; ++c) sf[c] = 0.f;
    for (int c = 0; sub && c < k; ++c) sf[cperm(c)] = float(sub[c]);
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
    ctx->reset_stages();
    PcaState* p = new PcaState();
    ctx->pca = p;
    p->n = n;
    p->d = d;
    p->ldx = (int64_t(d) + CH - 1) / CH * CH;
    const bool need_pad = p->ldx != d;
    if (x_on_device && !need_pad && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        p->X = X;
    } else if (!need_pad) {
        GT_HIP(ctx, p->X_own.reserve(size_t(n) * d * sizeof(float)));
        if (x_on_device)
            GT_HIP(ctx, hipMemcpyAsync(p->X_own.p, X, size_t(n) * d * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        else
            GT_TRY(gt_copy_from_host(ctx, p->X_own.p, X, size_t(n) * d * sizeof(float)));
        p->X = p->X_own.as<float>();
    } else {
        // stage the caller's rows (in the first thin buffer's memory when it is large enough), then pad
        DevBuf raw;
        const float* src = X;
        if (!x_on_device) {
            GT_HIP(ctx, raw.reserve(size_t(n) * d * sizeof(float)));
            GT_TRY(gt_copy_from_host(ctx, raw.p, X, size_t(n) * d * sizeof(float)));
            src = raw.as<float>();
        }
        GT_HIP(ctx, p->X_own.reserve(size_t(n) * p->ldx * sizeof(float)));
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)ceil_div64(n * p->ldx, 256)), dim3(256), 0, ctx->stream, src, n, d,
                           p->ldx, p->X_own.as<float>());
        GT_HIP(ctx, hipGetLastError());
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        raw.release();
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
                       n, d, p->ldx, rpb, partm);
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
    GT_TRY(upload_factor(ctx, p, W, wrows, k, sub, src != 0));
    const float* S = src == 0 ? p->X : p->Y[src - 1].as<float>();
    const int64_t lds = src == 0 ? p->ldx : int64_t(KP);
    StageSpan span(ctx, "pca_matmul");
    hipLaunchKernelGGL(tall_matmul_kernel, dim3((unsigned)ceil_div64(p->n, 128)), dim3(256), 0, ctx->stream, S, p->n,
                       int(lds / CH), lds, p->W.as<float>(), sub ? p->sub.as<float>() : nullptr, p->Y[dst - 1].as<float>());
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
                           int(p->ldx), p->ldx, Yb, p->part.as<float>());
        GT_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div64(int64_t(per_slab), 256)), dim3(256), 0, ctx->stream,
                           p->part.as<float>(), int64_t(per_slab), nslab, p->out64.as<double>());
        GT_HIP(ctx, hipGetLastError());
    }
    std::vector<double> host(per_slab);
    GT_HIP(ctx, hipMemcpyAsync(host.data(), p->out64.p, per_slab * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < p->d; ++f)
        for (int c = 0; c < k; ++c) out[size_t(f) * k + c] = host[size_t(f) * KP + cperm(c)];
    if (colsum) {
        double* cpart = reinterpret_cast<double*>(p->part.p);
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nslab), dim3(128), 0, ctx->stream, Yb, p->n, cpart);
        GT_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_cols_kernel, dim3(1), dim3(256), 0, ctx->stream, cpart, int64_t(KP), nslab, p->out64.as<double>());
        GT_HIP(ctx, hipGetLastError());
        double cs[KP];
        GT_HIP(ctx, hipMemcpyAsync(cs, p->out64.p, sizeof(cs), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int c = 0; c < k; ++c) colsum[c] = cs[cperm(c)];
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
        for (int b = 0; b < k; ++b) out[size_t(a) * k + b] = host[size_t(cperm(a)) * KP + cperm(b)];
    return GT_OK;
}

// rows of a thin buffer, first k columns, float32 row-major [n][k]; on_device: `out` is device memory
extern "C" int gt_pca_fetch(gt_ctx* ctx, int32_t ybuf, int32_t k, float* out, int32_t on_device) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    PcaState* p = state_of(ctx);
    if (!p) GT_FAIL(ctx, GT_E_STATE, "gt_pca_fetch: call gt_pca_begin first");
    if (!out || k < 1 || k > KP || ybuf < 1 || ybuf > 2) GT_FAIL(ctx, GT_E_ARG, "gt_pca_fetch: bad arguments");
    float* dense = out;
    if (!on_device) {
        GT_HIP(ctx, p->part.reserve(size_t(p->n) * k * sizeof(float)));
        dense = p->part.as<float>();
    }
    hipLaunchKernelGGL(compact_cols_kernel, dim3((unsigned)ceil_div64(p->n * int64_t(k), 256)), dim3(256), 0, ctx->stream,
                       p->Y[ybuf - 1].as<float>(), p->n, k, dense);
    GT_HIP(ctx, hipGetLastError());
    if (!on_device) return gt_copy_to_host(ctx, out, dense, size_t(p->n) * k * sizeof(float));
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
