// Point preprocessing: the padded float32 working copy consumed by the MFMA candidate kernels, the
// float64 squared row norms used by the exact re-rank (scikit-learn computes them in float64 as well,
// sklearn:metrics/_pairwise_distances_reduction/_base.pyx.tp:45-83) and the -|y|^2/2 accumulator seeds.
// HBM-bound, one pass over the data.
#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn.h"

#include <algorithm>
#include <utility>
#include <vector>

namespace {

// `sel` (optional): the working copy holds columns sel[0 .. dw) of X instead of its first dw = min(d, DP) columns
template <typename T>
__global__ __launch_bounds__(256) void pad_convert_kernel(const T* __restrict__ X, int64_t n, int d, int DP,
                                                          int64_t n_pad, float* __restrict__ Yp,
                                                          const int32_t* __restrict__ sel, int dw) {
    const int64_t total4 = n_pad * DP / 4;
    const int dp4 = DP / 4;
    for (int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x; f < total4; f += int64_t(gridDim.x) * 256) {
        const int64_t r = f / dp4;
        const int c = int(f % dp4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) {
            const T* src = X + r * int64_t(d);
            if (c + 0 < dw) v.x = float(src[sel ? sel[c + 0] : c + 0]);
            if (c + 1 < dw) v.y = float(src[sel ? sel[c + 1] : c + 1]);
            if (c + 2 < dw) v.z = float(src[sel ? sel[c + 2] : c + 2]);
            if (c + 3 < dw) v.w = float(src[sel ? sel[c + 3] : c + 3]);
        }
        reinterpret_cast<float4*>(Yp)[f] = v;
    }
}

// per-column sums and sums of squares (float64): thread = one column of a 256-column slab, block = a chunk of rows
template <typename T>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* __restrict__ X, int64_t n, int d, int64_t rows_per_block,
                                                        double* __restrict__ stat) {
    const int c = int(blockIdx.x) * 256 + threadIdx.x;
    const int64_t r0 = int64_t(blockIdx.y) * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    if (c >= d) return;
    double s = 0.0, q = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        const double v = double(X[r * int64_t(d) + c]);
        s += v;
        q = fma(v, v, q);
    }
    atomicAdd(&stat[c], s);
    atomicAdd(&stat[d + c], q);
}

// max |x| over the matrix (for the power-of-two scale of the split-float16 working copy) and non-finite flags
// (out_bits[1]: bit 0 = a NaN was seen, bit 1 = an infinity)
template <typename T>
__global__ __launch_bounds__(256) void max_abs_kernel(const T* __restrict__ X, const int64_t total,
                                                      unsigned long long* __restrict__ out_bits) {
    double m = 0.0;
    bool nan = false;
    for (int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x; f < total; f += int64_t(gridDim.x) * 256) {
        const double v = fabs(double(X[f]));
        nan |= v != v;
        m = v > m ? v : m;   // NaN never wins, infinity does
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, lane_xor_f64(m, o));
    // (one atomic per workgroup: same-address atomics of thousands of waves serialise)
    __shared__ double wm[4];
    __shared__ int wn[4];
    const int w = threadIdx.x >> 6;
    const bool wave_nan = __ballot(nan) != 0ull;   // (all lanes vote: outside the branch)
    if ((threadIdx.x & 63) == 0) {
        wm[w] = m;
        wn[w] = wave_nan ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(out_bits, (unsigned long long)__double_as_longlong(fmax(fmax(wm[0], wm[1]), fmax(wm[2], wm[3]))));
        if (wn[0] | wn[1] | wn[2] | wn[3]) atomicOr(out_bits + 1, 1ull);
    }
}

// (float32 data, 16-byte loads: four independent maxima per thread - the scalar form above runs at 1 TB/s)
__global__ __launch_bounds__(256) void max_abs_f4_kernel(const float4* __restrict__ X4, const int64_t total4,
                                                         unsigned long long* __restrict__ out_bits) {
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    bool nan = false;
    const int64_t stride = int64_t(gridDim.x) * 256;
    auto take = [&](const float4 v) {
        const float a = fabsf(v.x), b = fabsf(v.y), c = fabsf(v.z), e = fabsf(v.w);
        nan |= (a != a) | (b != b) | (c != c) | (e != e);
        m0 = a > m0 ? a : m0;   // NaN never wins, infinity does
        m1 = b > m1 ? b : m1;
        m2 = c > m2 ? c : m2;
        m3 = e > m3 ? e : m3;
    };
    int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    for (; f + 3 * stride < total4; f += 4 * stride) {   // four loads in flight per thread
        const float4 v0 = X4[f], v1 = X4[f + stride], v2 = X4[f + 2 * stride], v3 = X4[f + 3 * stride];
        take(v0);
        take(v1);
        take(v2);
        take(v3);
    }
    for (; f < total4; f += stride) take(X4[f]);
    m0 = m0 > m1 ? m0 : m1;
    m2 = m2 > m3 ? m2 : m3;
    float mf = m0 > m2 ? m0 : m2;
    mf = wave_max_f32(mf);
    // ONE atomic per workgroup: 16 384 waves on one address serialise (7 ns each: the kernel took 0.2 ms for 256 MB)
    __shared__ float wm[4];
    __shared__ int wn[4];
    const int w = threadIdx.x >> 6;
    const bool wave_nan = __ballot(nan) != 0ull;   // (all lanes vote: outside the branch)
    if ((threadIdx.x & 63) == 0) {
        wm[w] = mf;
        wn[w] = wave_nan ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        atomicMax(out_bits, (unsigned long long)__double_as_longlong(double(a)));
        if (wn[0] | wn[1] | wn[2] | wn[3]) atomicOr(out_bits + 1, 1ull);
    }
}

// the same for float32 rows that fill the padded width exactly (d == DP, a multiple of 8, no column selection, 16-byte
// aligned): eight features per thread - two 16-byte loads, one 16-byte store per plane
__global__ __launch_bounds__(256) void pad_split_f16_v8_kernel(const float4* __restrict__ X4, const int64_t n, const int DP,
                                                               const int64_t n_pad, const float sc,
                                                               uint4* __restrict__ Yh, uint4* __restrict__ Yc) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int c8 = DP / 8;
    const int64_t total = n_pad * c8;
    for (int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x; f < total; f += int64_t(gridDim.x) * 256) {
        const int64_t r = f / c8;
        const int c = int(f - r * c8);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r < n) {
            const float4 a = X4[r * int64_t(2 * c8) + 2 * c], b = X4[r * int64_t(2 * c8) + 2 * c + 1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = v[e] * sc;              // exact: sc is a power of two
            const _Float16 h = _Float16(x);
            hi[e] = h;
            lo[e] = _Float16(x - float(h));         // (exact in float32: the residual of a float32 against its float16 rounding)
        }
        uint4 hb, lb;
        __builtin_memcpy(&hb, &hi, 16);
        __builtin_memcpy(&lb, &lo, 16);
        Yh[r * int64_t(2 * c8) + c] = hb;           // row = hi plane | lo plane
        Yh[r * int64_t(2 * c8) + c8 + c] = lb;
        if (Yc) Yc[f] = hb;
    }
}

// split-float16 working copy: row = hi plane (DP halves) | lo plane (DP halves), x*sc = hi + lo + O(2^-22 |x*sc|)
template <typename T>
__global__ __launch_bounds__(256) void pad_split_f16_kernel(const T* __restrict__ X, int64_t n, int d, int DP,
                                                            int64_t n_pad, double sc, _Float16* __restrict__ Yh,
                                                            _Float16* __restrict__ Yc, const int32_t* __restrict__ sel,
                                                            int dw) {
    const int64_t total = n_pad * DP;
    for (int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x; f < total; f += int64_t(gridDim.x) * 256) {
        const int64_t r = f / DP;
        const int c = int(f % DP);
        double v = 0.0;
        if (r < n && c < dw) v = double(X[r * int64_t(d) + (sel ? sel[c] : c)]) * sc;   // exact: sc is a power of two
        const _Float16 hi = _Float16(v);
        const _Float16 lo = _Float16(v - double(hi));
        Yh[r * int64_t(2 * DP) + c] = hi;
        Yh[r * int64_t(2 * DP) + DP + c] = lo;
        if (Yc) Yc[f] = hi;   // compact copy of the hi plane (rows of 2*DP bytes) for the single-chain pass
    }
}

// sklearn.preprocessing.normalize(X, "l2"): x / sqrt(sum x^2), rows with zero norm untouched
template <typename T>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const T* __restrict__ X, T* __restrict__ out, int64_t n, int d) {
    const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (r >= n) return;
    const T* src = X + r * int64_t(d);
    double acc = 0.0;
    for (int k = 0; k < d; ++k) {
        const double v = double(src[k]);
        acc = fma(v, v, acc);
    }
    T nrm = T(sqrt(acc));
    if (nrm == T(0)) nrm = T(1);
    T* dst = out + r * int64_t(d);
    for (int k = 0; k < d; ++k) dst[k] = src[k] / nrm;
}

template <typename T>
__global__ __launch_bounds__(256) void row_norm_kernel(const T* __restrict__ X, int64_t n, int d, int64_t n_pad,
                                                       double* __restrict__ xn, float* __restrict__ hneg,
                                                       const double sc2, unsigned long long* __restrict__ ymax2_bits,
                                                       const double sc, unsigned long long* __restrict__ lomax2_bits,
                                                       const int32_t* __restrict__ sel, const int dw,
                                                       double* __restrict__ xn_sel) {
    // xn: squared norm over all d columns (exact stages).  With `sel` the candidate pass sees only columns sel[0..dw):
    // its seeds, the norm bound and the residual bound come from the partial norm (xn_sel).
    // One thread per row, columns accumulated in the canonical order of the exact stages (gt_dot16: a row's distance to
    // itself must come out 0); the
    // block's 256 rows are staged through LDS in chunks of 32 columns so that the global reads are coalesced.
    constexpr int CH = 32;
    __shared__ T chunk[256][CH + 1];
    const int64_t r0 = int64_t(blockIdx.x) * 256;
    const int64_t r = r0 + threadIdx.x;
    double acc = 0.0, accs = 0.0, lo2 = 0.0;
    double a16[16];   // the canonical partial sums (gt_device.h gt_dot16): column k goes to sum (k >> 2) & 15
#pragma unroll
    for (int l = 0; l < 16; ++l) a16[l] = 0.0;
    for (int c0 = 0; c0 < d; c0 += CH) {
        const int cw = d - c0 < CH ? d - c0 : CH;
        __syncthreads();
        for (int f = threadIdx.x; f < 256 * cw; f += 256) {
            const int rr = f / cw, cc = f % cw;
            chunk[rr][cc] = (r0 + rr < n) ? X[(r0 + rr) * int64_t(d) + c0 + cc] : T(0);
        }
        __syncthreads();
        if (r < n) {
            if ((c0 & 32) == 0) {   // (uniform: columns 0 ... 31 of a group of 64 feed sums 0 ... 7, the others 8 ... 15)
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (k < cw) {
                        const double v = double(chunk[threadIdx.x][k]);
                        a16[k >> 2] = fma(v, v, a16[k >> 2]);
                    }
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (k < cw) {
                        const double v = double(chunk[threadIdx.x][k]);
                        a16[8 + (k >> 2)] = fma(v, v, a16[8 + (k >> 2)]);
                    }
            }
            for (int k = 0; k < cw; ++k) {
                const double v = double(chunk[threadIdx.x][k]);
                if (lomax2_bits && !sel) {
                    // exact residual of the float16 rounding of the scaled value (what the hi-plane-only pass drops)
                    const double vs = v * sc;
                    const double res = vs - double(_Float16(vs));
                    lo2 = fma(res, res, lo2);
                }
            }
        }
    }
    acc = gt_tree16(a16);
    if (r < n) {
        const T* src = X + r * int64_t(d);
        accs = acc;
        if (sel) {
            accs = 0.0;
            for (int k = 0; k < dw; ++k) {
                const double v = double(src[sel[k]]);
                accs = fma(v, v, accs);
                if (lomax2_bits) {
                    const double vs = v * sc;
                    const double res = vs - double(_Float16(vs));
                    lo2 = fma(res, res, lo2);
                }
            }
            xn_sel[r] = accs;
        }
        xn[r] = acc;
        if (hneg) hneg[r] = float(-0.5 * accs * sc2);
    } else if (r < n_pad) {
        if (hneg) hneg[r] = -INFINITY;
    }
    // block max -> one atomic per wave
    double m = (r < n) ? accs : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && ymax2_bits) atomicMax(ymax2_bits, (unsigned long long)__double_as_longlong(m));
    if (ymax2_bits) {
        // [1]: the largest FULL squared norm (differs from [0] for wide data; the cosine bounds need it)
        double mf = (r < n) ? acc : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mf = fmax(mf, __shfl_xor(mf, o));
        if ((threadIdx.x & 63) == 0) atomicMax(ymax2_bits + 1, (unsigned long long)__double_as_longlong(mf));
    }
    if (lomax2_bits) {
        double l = (r < n) ? lo2 : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l = fmax(l, __shfl_xor(l, o));
        if ((threadIdx.x & 63) == 0) atomicMax(lomax2_bits, (unsigned long long)__double_as_longlong(l));
    }
}

// The same outputs for float32 rows with d a multiple of 4 and at most 64, all columns (no selection): FOUR LANES PER ROW,
// lane c reads quad c of every 64-byte sector (one 16-byte load each, the row's 256 bytes in one instruction of the four
// lanes) - no LDS staging, 16 rows per wave in flight.  The lane's four accumulators are the partial sums c, c + 4, c + 8,
// c + 12 of gt_dot16, the lane tree is gt_tree16's: xn comes out bit for bit as above (a row's distance to itself must be 0).
// The float16 residuals are summed per lane and then across the four lanes (an upper bound's rounding order is free).
__global__ __launch_bounds__(256) void row_norm4_kernel(const float* __restrict__ X, const int64_t n, const int d,
                                                        const int64_t n_pad, double* __restrict__ xn, float* __restrict__ hneg,
                                                        const double sc2, unsigned long long* __restrict__ ymax2_bits,
                                                        const double sc, unsigned long long* __restrict__ lomax2_bits) {
    const int lane = threadIdx.x & 63, c = lane & 3;
    double m = 0.0, l = 0.0;   // running maxima of this lane group (the atomics on ONE address are paid once per workgroup)
    for (int64_t r = int64_t(blockIdx.x) * 64 + (threadIdx.x >> 2); r < n_pad; r += int64_t(gridDim.x) * 64) {
        if (r < n) {
            const float4* row = reinterpret_cast<const float4*>(X + r * int64_t(d));
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (16 * i + 4 * c < d) ? row[4 * i + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            double lo2 = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (16 * i + 4 * c < d) {
                    const double e[4] = {double(v[i].x), double(v[i].y), double(v[i].z), double(v[i].w)};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        a[i] = fma(e[k], e[k], a[i]);
                        if (lomax2_bits) {
                            const double vs = e[k] * sc;
                            const double res = vs - double(_Float16(vs));
                            lo2 = fma(res, res, lo2);
                        }
                    }
                }
            const double cc = (a[0] + a[2]) + (a[1] + a[3]);   // b[c] + b[c + 4] of gt_tree16
            const double w2 = cc + lane_xor_f64(cc, 2);
            const double acc = w2 + lane_xor_f64(w2, 1);
            lo2 += lane_xor_f64(lo2, 2);
            lo2 += lane_xor_f64(lo2, 1);
            if (c == 0) {
                xn[r] = acc;
                if (hneg) hneg[r] = float(-0.5 * acc * sc2);
            }
            m = fmax(m, acc);
            l = fmax(l, lo2);
        } else if (c == 0 && hneg) {
            hneg[r] = -INFINITY;
        }
    }
    __shared__ double red[2][4];
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) {
        m = fmax(m, lane_xor_f64(m, o));
        l = fmax(l, lane_xor_f64(l, o));
    }
    if (lane == 0) {
        red[0][threadIdx.x >> 6] = m;
        red[1][threadIdx.x >> 6] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mm = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
        const double ll = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
        if (ymax2_bits) {
            atomicMax(ymax2_bits, (unsigned long long)__double_as_longlong(mm));
            atomicMax(ymax2_bits + 1, (unsigned long long)__double_as_longlong(mm));
        }
        if (lomax2_bits) atomicMax(lomax2_bits, (unsigned long long)__double_as_longlong(ll));
    }
}

}  // namespace

int gt_normalize_rows(gt_ctx* ctx, const void* X, void* out, int64_t n, int d, int dtype) {
    const int64_t nb = ceil_div64(n, 256);
    if (dtype == GT_F32)
        hipLaunchKernelGGL(normalize_rows_kernel<float>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const float*)X,
                           (float*)out, n, d);
    else
        hipLaunchKernelGGL(normalize_rows_kernel<double>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const double*)X,
                           (double*)out, n, d);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

int gt_max_abs(gt_ctx* ctx, const void* Xdev, int64_t total, int dtype, double* out_host, uint32_t* nonfinite) {
    DevBuf& tmp = ctx->small_tmp;
    GT_HIP(ctx, tmp.reserve(64));
    GT_HIP(ctx, hipMemsetAsync(tmp.p, 0, 2 * sizeof(double), ctx->stream));
    int64_t blocks = std::min<int64_t>(ceil_div64(total, 256), 4096);
    if (dtype == GT_F32 && (total & 3) == 0 && (reinterpret_cast<uintptr_t>(Xdev) & 15) == 0)
        hipLaunchKernelGGL(max_abs_f4_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(total / 4, 256), 4096)), dim3(256), 0,
                           ctx->stream, (const float4*)Xdev, total / 4, (unsigned long long*)tmp.p);
    else if (dtype == GT_F32)
        hipLaunchKernelGGL(max_abs_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)Xdev,
                           total, (unsigned long long*)tmp.p);
    else
        hipLaunchKernelGGL(max_abs_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const double*)Xdev,
                           total, (unsigned long long*)tmp.p);
    unsigned long long host[2] = {0ull, 0ull};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, tmp.p, sizeof(host), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->set_error(std::string("max_abs: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    std::memcpy(out_host, &host[0], sizeof(double));
    if (nonfinite) *nonfinite = uint32_t(host[1] & 1ull) | (std::isinf(*out_host) ? 2u : 0u);
    return GT_OK;
}

int gt_fail_nonfinite(gt_ctx* ctx, uint32_t flags, int dtype) {
    if (flags & 1u)
        ctx->set_error("Input X contains NaN.");
    else
        ctx->set_error(std::string("Input X contains infinity or a value too large for dtype('") +
                       (dtype == GT_F32 ? "float32" : "float64") + "').");
    return GT_E_NONFINITE;
}

// power-of-two scale that puts max|x| * sc into [2^13, 2^14): float16 hi parts stay far from overflow
double gt_f16_scale(double maxabs) {
    if (!(maxabs > 0.0) || !std::isfinite(maxabs)) return 1.0;
    int e = 0;
    (void)std::frexp(maxabs, &e);   // maxabs = m * 2^e, m in [0.5, 1)
    return std::ldexp(1.0, 14 - e);
}

int gt_prep_matrix(gt_ctx* ctx, const void* Xdev, int64_t n, int d, int dtype, int DP, int64_t n_pad, float* Yp,
                   double* xn, float* hneg, double* ymax2, int prec, double sc, double* lomax2, void* Yc,
                   const int32_t* sel, int dsel, double* xn_sel) {
    const int dw = sel ? dsel : (d < DP || DP == 0 ? d : DP);
    if (ymax2) GT_HIP(ctx, hipMemsetAsync(ymax2, 0, 2 * sizeof(double), ctx->stream));   // [0] scored norms, [1] full norms
    if (lomax2) GT_HIP(ctx, hipMemsetAsync(lomax2, 0, sizeof(double), ctx->stream));
    if (Yp && prec == 1) {
        const int64_t total = n_pad * DP;
        int64_t blocks = std::min<int64_t>(ceil_div64(total, 256), 16384);
        if (dtype == GT_F32 && !sel && d == DP && (DP & 7) == 0 && (reinterpret_cast<uintptr_t>(Xdev) & 15) == 0 &&
            sc <= 3.0e38 && sc >= 1.0e-38)
            hipLaunchKernelGGL(pad_split_f16_v8_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(total / 8, 256), 16384)),
                               dim3(256), 0, ctx->stream, (const float4*)Xdev, n, DP, n_pad, float(sc), (uint4*)Yp, (uint4*)Yc);
        else if (dtype == GT_F32)
            hipLaunchKernelGGL(pad_split_f16_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                               (const float*)Xdev, n, d, DP, n_pad, sc, (_Float16*)Yp, (_Float16*)Yc, sel, dw);
        else
            hipLaunchKernelGGL(pad_split_f16_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                               (const double*)Xdev, n, d, DP, n_pad, sc, (_Float16*)Yp, (_Float16*)Yc, sel, dw);
        GT_HIP(ctx, hipGetLastError());
    } else if (Yp) {
        const int64_t total4 = n_pad * DP / 4;
        int64_t blocks = ceil_div64(total4, 256);
        if (blocks > 8192) blocks = 8192;
        if (dtype == GT_F32)
            hipLaunchKernelGGL(pad_convert_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                               (const float*)Xdev, n, d, DP, n_pad, Yp, sel, dw);
        else
            hipLaunchKernelGGL(pad_convert_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                               (const double*)Xdev, n, d, DP, n_pad, Yp, sel, dw);
        GT_HIP(ctx, hipGetLastError());
    }
    const int64_t rows = hneg ? n_pad : n;
    const int64_t nb = ceil_div64(rows, 256);
    if (dtype == GT_F32 && !sel && (d & 3) == 0 && d <= 64 && (reinterpret_cast<uintptr_t>(Xdev) & 15) == 0) {
        hipLaunchKernelGGL(row_norm4_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(rows, 64), 4096)), dim3(256), 0, ctx->stream,
                           (const float*)Xdev, n, d, rows, xn, hneg, sc * sc, (unsigned long long*)ymax2, sc,
                           (unsigned long long*)lomax2);
        GT_HIP(ctx, hipGetLastError());
        return GT_OK;
    }
    if (dtype == GT_F32)
        hipLaunchKernelGGL(row_norm_kernel<float>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const float*)Xdev, n,
                           d, n_pad, xn, hneg, sc * sc, (unsigned long long*)ymax2, sc, (unsigned long long*)lomax2, sel, dw, xn_sel);
    else
        hipLaunchKernelGGL(row_norm_kernel<double>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const double*)Xdev,
                           n, d, n_pad, xn, hneg, sc * sc, (unsigned long long*)ymax2, sc, (unsigned long long*)lomax2, sel, dw, xn_sel);
    GT_HIP(ctx, hipGetLastError());
    return GT_OK;
}

// Wide data: pick the `want` columns of largest variance (ties by index) and upload their indices.
int gt_select_columns(gt_ctx* ctx, int want) {
    const int d = ctx->d;
    GT_HIP(ctx, ctx->colstat.reserve(size_t(2) * d * sizeof(double)));
    GT_HIP(ctx, hipMemsetAsync(ctx->colstat.p, 0, size_t(2) * d * sizeof(double), ctx->stream));
    const int64_t rows_per_block = 2048;
    dim3 grid((unsigned)ceil_div64(d, 256), (unsigned)ceil_div64(ctx->n, rows_per_block));
    if (ctx->dtype == GT_F32)
        hipLaunchKernelGGL(col_stats_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)ctx->X, ctx->n, d,
                           rows_per_block, ctx->colstat.as<double>());
    else
        hipLaunchKernelGGL(col_stats_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double*)ctx->X, ctx->n, d,
                           rows_per_block, ctx->colstat.as<double>());
    GT_HIP(ctx, hipGetLastError());
    std::vector<double> st(size_t(2) * d);
    GT_HIP(ctx, hipMemcpyAsync(st.data(), ctx->colstat.p, st.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<std::pair<double, int>> var(d);
    for (int c = 0; c < d; ++c) {
        const double mean = st[c] / double(ctx->n);
        double v = st[d + c] / double(ctx->n) - mean * mean;
        if (!(v > 0.0)) v = 0.0;   // also NaN
        var[c] = {-v, c};
    }
    std::sort(var.begin(), var.end());
    std::vector<int32_t> sel(want);
    for (int k = 0; k < want; ++k) sel[k] = var[k].second;
    std::sort(sel.begin(), sel.end());   // ascending columns: friendlier gathers
    GT_HIP(ctx, ctx->sel_idx.reserve(size_t(want) * sizeof(int32_t)));
    GT_HIP(ctx, hipMemcpyAsync(ctx->sel_idx.p, sel.data(), size_t(want) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dsel = want;
    return GT_OK;
}

int gt_prep_points(gt_ctx* ctx) {
    ctx->sc = 1.0;
    if (ctx->prec == 1) {
        // euclidean: gt_set_points has just measured max|x| together with its finiteness check
        if (ctx->metric == 1) GT_TRY(gt_max_abs(ctx, ctx->X, ctx->n * int64_t(ctx->d), ctx->dtype, &ctx->maxabs, nullptr));
        ctx->sc = gt_f16_scale(ctx->maxabs);
    }
    StageSpan span(ctx, "prep", 2);
    const int bn = gt_select_bn_for(ctx->DP);
    ctx->n_pad = ceil_div64(ctx->n, bn) * bn;
    GT_HIP(ctx, ctx->Yp.reserve(size_t(ctx->n_pad) * ctx->DP * sizeof(float)));
    GT_HIP(ctx, ctx->xn.reserve(size_t(ctx->n) * sizeof(double)));
    GT_HIP(ctx, ctx->hneg.reserve(size_t(ctx->n_pad) * sizeof(float)));
    GT_HIP(ctx, ctx->ymax.reserve(sizeof(double)));
    GT_HIP(ctx, ctx->lomax_dev.reserve(sizeof(double)));
    const bool want_hi = ctx->prec == 1 && ctx->fast_mode != 0;
    if (want_hi) GT_HIP(ctx, ctx->Yc.reserve(size_t(ctx->n_pad) * ctx->DP * sizeof(_Float16)));
    if (ctx->wide) GT_HIP(ctx, ctx->xn_sel.reserve(size_t(ctx->n) * sizeof(double)));
    GT_TRY(gt_prep_matrix(ctx, ctx->X, ctx->n, ctx->d, ctx->dtype, ctx->DP, ctx->n_pad, ctx->Yp.as<float>(),
                          ctx->xn.as<double>(), ctx->hneg.as<float>(), ctx->ymax.as<double>(), ctx->prec, ctx->sc,
                          ctx->prec == 1 ? ctx->lomax_dev.as<double>() : nullptr, want_hi ? ctx->Yc.p : nullptr,
                          ctx->wide ? ctx->sel_idx.as<int32_t>() : nullptr, ctx->dsel,
                          ctx->wide ? ctx->xn_sel.as<double>() : nullptr));
    ctx->lomax = 0.0;
    if (ctx->prec == 1) {
        double lo2 = 0.0;
        GT_HIP(ctx, hipMemcpyAsync(&lo2, ctx->lomax_dev.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->lomax = std::sqrt(lo2) / ctx->sc;
    }
    return GT_OK;
}
