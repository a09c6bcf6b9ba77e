// float64 helpers for tall thin matrices (n rows, at most 128 columns, row-major, device resident) next to the sparse
// operators of a finished graph: the pieces of sklearn.utils.extmath.randomized_svd on `diff_aff` that the spectral
// landmark front end needs (reference graphs.py:1215-1230) besides gt_graph_spmm -
//   gt_thin_scale_rows   A[r][:] *= v[r]^p             (diff_aff = D^-1/2 K D^-1/2 applied as scale, K-SpMM, scale)
//   gt_thin_gram         C = A^T A  (k x k, to the host)  for the Cholesky-QR of a power-iteration block
//   gt_thin_rmul         B = A R    (R k x m from the host)
// HBM-bound streaming kernels; the driver is graphtools_amd/_spectral.py.
#include "gt_common.h"
#include "gt_device.h"

#include <vector>

namespace {

constexpr int TK = 128;     // most columns a thin matrix may have here
constexpr int TSLAB = 1024; // rows per partial Gram

__global__ __launch_bounds__(256) void scale_rows_kernel(double* __restrict__ A, const int64_t n, const int k,
                                                         const double* __restrict__ v, const double p) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= n * k) return;
    A[e] *= pow(v[e / k], p);
}

// partial C = A^T A over a slab of rows: thread (ta, tb) of a 16 x 16 grid owns an 8 x 8 block, rows staged 16 at a time
__global__ __launch_bounds__(256) void thin_gram_kernel(const double* __restrict__ A, const int64_t n, const int k,
                                                        double* __restrict__ part) {
    __shared__ double al[16][TK + 1];
    const int tid = threadIdx.x, ta = tid >> 4, tb = tid & 15;
    const int64_t r_begin = int64_t(blockIdx.x) * TSLAB;
    const int64_t r_end = r_begin + TSLAB < n ? r_begin + TSLAB : n;
    double c[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) c[a][b] = 0.0;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 16) {
        __syncthreads();
        for (int f = tid; f < 16 * TK; f += 256) {
            const int rr = f >> 7, cc = f & 127;
            al[rr][cc] = (r0 + rr < r_end && cc < k) ? A[(r0 + rr) * k + cc] : 0.0;
        }
        __syncthreads();
        for (int rr = 0; rr < 16; ++rr) {
            double va[8], vb[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) va[a] = al[rr][8 * ta + a];
#pragma unroll
            for (int b = 0; b < 8; ++b) vb[b] = al[rr][8 * tb + b];
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) c[a][b] = fma(va[a], vb[b], c[a][b]);
        }
    }
    double* po = part + size_t(blockIdx.x) * TK * TK;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) po[size_t(8 * ta + a) * TK + 8 * tb + b] = c[a][b];
}

__global__ __launch_bounds__(256) void thin_reduce_kernel(const double* __restrict__ part, const int64_t per, const int nb,
                                                          double* __restrict__ out) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= per) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[size_t(b) * per + e];
    out[e] = s;
}

// B[r][c] = sum_j A[r][j] R[j][c]: R (k x m) in LDS, one row per 8 threads (32 rows per workgroup), fixed summation order
__global__ __launch_bounds__(256) void thin_rmul_kernel(const double* __restrict__ A, const int64_t n, const int k,
                                                        const double* __restrict__ R, const int m, double* __restrict__ B) {
    extern __shared__ double sm[];
    double* rl = sm;               // [k][m]
    double* al = sm + size_t(k) * m;   // [32][k]
    const int tid = threadIdx.x;
    for (int f = tid; f < k * m; f += 256) rl[f] = R[f];
    const int64_t r0 = int64_t(blockIdx.x) * 32;
    for (int f = tid; f < 32 * k; f += 256) {
        const int64_t r = r0 + f / k;
        al[f] = r < n ? A[r * k + f % k] : 0.0;
    }
    __syncthreads();
    const int rr = tid >> 3, cg = tid & 7;
    const int64_t r = r0 + rr;
    if (r >= n) return;
    for (int c = cg; c < m; c += 8) {
        double acc = 0.0;
        for (int j = 0; j < k; ++j) acc = fma(al[rr * k + j], rl[j * m + c], acc);
        B[r * m + c] = acc;
    }
}

}  // namespace

extern "C" int gt_thin_scale_rows(gt_ctx* ctx, double* A_dev, int64_t n, int32_t k, const double* v_dev, double power) {
    if (!ctx || !A_dev || !v_dev || n < 1 || k < 1) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)ceil_div64(n * int64_t(k), 256)), dim3(256), 0, ctx->stream, A_dev, n, k,
                       v_dev, power);
    GT_HIP(ctx, hipGetLastError());
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

extern "C" int gt_thin_gram(gt_ctx* ctx, const double* A_dev, int64_t n, int32_t k, double* out_host) {
    if (!ctx || !A_dev || !out_host || n < 1 || k < 1) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (k > TK) GT_FAIL(ctx, GT_E_LIMIT, "gt_thin_gram: at most 128 columns");
    const int nslab = int(ceil_div64(n, TSLAB));
    DevBuf part, out;
    GT_HIP(ctx, part.reserve(size_t(nslab) * TK * TK * sizeof(double)));
    GT_HIP(ctx, out.reserve(size_t(TK) * TK * sizeof(double)));
    hipLaunchKernelGGL(thin_gram_kernel, dim3((unsigned)nslab), dim3(256), 0, ctx->stream, A_dev, n, k, part.as<double>());
    hipLaunchKernelGGL(thin_reduce_kernel, dim3((unsigned)ceil_div64(TK * TK, 256)), dim3(256), 0, ctx->stream, part.as<double>(),
                       int64_t(TK) * TK, nslab, out.as<double>());
    hipError_t e = hipGetLastError();
    std::vector<double> host(size_t(TK) * TK);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), out.p, host.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    part.release();
    out.release();
    if (e != hipSuccess) {
        ctx->set_error(std::string("gt_thin_gram: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) out_host[size_t(a) * k + b] = host[size_t(a) * TK + b];
    return GT_OK;
}

extern "C" int gt_thin_rmul(gt_ctx* ctx, const double* A_dev, int64_t n, int32_t k, const double* R_host, int32_t m,
                            double* B_dev) {
    if (!ctx || !A_dev || !R_host || !B_dev || n < 1 || k < 1 || m < 1) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    if (k > TK || m > TK) GT_FAIL(ctx, GT_E_LIMIT, "gt_thin_rmul: at most 128 columns");
    if (A_dev == B_dev) GT_FAIL(ctx, GT_E_ARG, "gt_thin_rmul: source and destination must differ");
    DevBuf R;
    GT_HIP(ctx, R.reserve(size_t(k) * m * sizeof(double)));
    GT_HIP(ctx, hipMemcpyAsync(R.p, R_host, size_t(k) * m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = (size_t(k) * m + size_t(32) * k) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(thin_rmul_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(lds));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(thin_rmul_kernel, dim3((unsigned)ceil_div64(n, 32)), dim3(256), lds, ctx->stream, A_dev, n, k,
                           R.as<double>(), m, B_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    R.release();
    if (e != hipSuccess) {
        ctx->set_error(std::string("gt_thin_rmul: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    return GT_OK;
}
