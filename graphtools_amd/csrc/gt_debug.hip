// Development / unit-test hooks (not part of the public ABI in include/graphtools_amd.h): they expose
// the device primitives so that tests on the GPU box can check them in isolation.
#include "gt_common.h"
#include "gt_device.h"
#include "gt_knn.h"

namespace {

template <int NT>
__global__ void dbg_sort_desc_kernel(const uint64_t* in, int n, uint64_t* out) {
    const int lane = threadIdx.x;
    uint64_t key[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int e = t * 64 + lane;
        key[t] = e < n ? in[e] : 0ull;
    }
    wave_bitonic_desc<NT>(key, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t * 64 + lane] = key[t];
}

template <int NT>
__global__ void dbg_sort_pair_kernel(const uint64_t* in_hi, const uint64_t* in_lo, int n, uint64_t* out_hi,
                                     uint64_t* out_lo) {
    const int lane = threadIdx.x;
    uint64_t hi[NT], lo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int e = t * 64 + lane;
        hi[t] = e < n ? in_hi[e] : ~0ull;
        lo[t] = e < n ? in_lo[e] : ~0ull;
    }
    wave_bitonic_asc_pair<NT>(hi, lo, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        out_hi[t * 64 + lane] = hi[t];
        out_lo[t * 64 + lane] = lo[t];
    }
}

// C[i][j] of one 32x32x2 MFMA chain: A = a[32][K], B = b[K][32] (b given as bt[32][K], i.e. B^T)
__global__ void dbg_mfma_kernel(const float* a, const float* bt, int K, float* c) {
    const int lane = threadIdx.x;
    const int li = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < K / 2; ++s) {
        const float av = a[li * K + h * (K / 2) + s];
        const float bv = bt[li * K + h * (K / 2) + s];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        c[row * 32 + li] = acc[r];
    }
}

// every lane exchange of gt_device.h against ds_bpermute (__shfl_xor): mismatches per form -> bad[0 .. 9]
__global__ void dbg_lane_ops_kernel(const uint32_t seed, uint32_t* bad) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = (uint32_t(blockIdx.x) * 64u + uint32_t(lane) + 1u) * 2654435761u ^ seed;
    const uint64_t v64 = (uint64_t(v) << 32) | uint64_t(~v * 40503u);
    auto chk = [&](int slot, bool ok) {
        if (!ok) atomicAdd(bad + slot, 1u);
    };
    chk(0, lane_xor_u32<1>(v) == uint32_t(__shfl_xor(int(v), 1)));
    chk(1, lane_xor_u32<2>(v) == uint32_t(__shfl_xor(int(v), 2)));
    chk(2, lane_xor_u32<4>(v) == uint32_t(__shfl_xor(int(v), 4)));
    chk(3, lane_xor_u32<8>(v) == uint32_t(__shfl_xor(int(v), 8)));
    chk(4, lane_xor_u32<16>(v) == uint32_t(__shfl_xor(int(v), 16)));
    chk(5, lane_xor_u32<32>(v) == uint32_t(__shfl_xor(int(v), 32)));
    chk(6, lane_xor_u64<16>(v64) == uint64_t(__shfl_xor((unsigned long long)v64, 16)) &&
               lane_xor_u64<4>(v64) == uint64_t(__shfl_xor((unsigned long long)v64, 4)));
    // reductions: against the same tree built from __shfl_xor (bitwise: same order of additions)
    const double x = double(int(v >> 8)) * 1e-3;
    double r = x;
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
    chk(7, wave_sum_f64(x) == r);
    float m = float(int(v >> 9));
    const float m0 = m;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    chk(8, wave_max_f32(m0) == m);
    int si = int(v >> 12);
    const int si0 = si;
    for (int o = 32; o > 0; o >>= 1) si += __shfl_xor(si, o);
    chk(9, wave_sum_i32(si0) == si);
}

}  // namespace

// self-test of the DPP / permlane lane exchanges and the reductions built on them (full waves, as their precondition demands)
extern "C" int gt_dbg_lane_ops(gt_ctx* ctx, uint32_t seed, uint32_t* bad10_host) {
    if (!ctx || !bad10_host) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf b;
    GT_HIP(ctx, b.reserve(10 * sizeof(uint32_t)));
    GT_HIP(ctx, hipMemsetAsync(b.p, 0, 10 * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(dbg_lane_ops_kernel, dim3(64), dim3(256), 0, ctx->stream, seed, b.as<uint32_t>());
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(bad10_host, b.p, 10 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    b.release();
    if (e != hipSuccess) {
        ctx->set_error(std::string("gt_dbg_lane_ops: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    return GT_OK;
}

extern "C" int gt_dbg_sort_desc(gt_ctx* ctx, const uint64_t* keys_host, int n, int nt, uint64_t* out_host) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf in, out;
    GT_HIP(ctx, in.reserve(size_t(64) * nt * 8));
    GT_HIP(ctx, out.reserve(size_t(64) * nt * 8));
    GT_HIP(ctx, hipMemcpy(in.p, keys_host, size_t(n) * 8, hipMemcpyHostToDevice));
    switch (nt) {
        case 1: hipLaunchKernelGGL(dbg_sort_desc_kernel<1>, dim3(1), dim3(64), 0, ctx->stream, in.as<uint64_t>(), n, out.as<uint64_t>()); break;
        case 2: hipLaunchKernelGGL(dbg_sort_desc_kernel<2>, dim3(1), dim3(64), 0, ctx->stream, in.as<uint64_t>(), n, out.as<uint64_t>()); break;
        case 8: hipLaunchKernelGGL(dbg_sort_desc_kernel<8>, dim3(1), dim3(64), 0, ctx->stream, in.as<uint64_t>(), n, out.as<uint64_t>()); break;
        case 32: hipLaunchKernelGGL(dbg_sort_desc_kernel<32>, dim3(1), dim3(64), 0, ctx->stream, in.as<uint64_t>(), n, out.as<uint64_t>()); break;
        default: GT_FAIL(ctx, GT_E_ARG, "nt");
    }
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GT_HIP(ctx, hipMemcpy(out_host, out.p, size_t(64) * nt * 8, hipMemcpyDeviceToHost));
    in.release();
    out.release();
    return GT_OK;
}

extern "C" int gt_dbg_sort_pair(gt_ctx* ctx, const uint64_t* hi_host, const uint64_t* lo_host, int n, int nt,
                                uint64_t* out_hi, uint64_t* out_lo) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf ih, il, oh, ol;
    const size_t bytes = size_t(64) * nt * 8;
    GT_HIP(ctx, ih.reserve(bytes));
    GT_HIP(ctx, il.reserve(bytes));
    GT_HIP(ctx, oh.reserve(bytes));
    GT_HIP(ctx, ol.reserve(bytes));
    GT_HIP(ctx, hipMemcpy(ih.p, hi_host, size_t(n) * 8, hipMemcpyHostToDevice));
    GT_HIP(ctx, hipMemcpy(il.p, lo_host, size_t(n) * 8, hipMemcpyHostToDevice));
    switch (nt) {
        case 1: hipLaunchKernelGGL(dbg_sort_pair_kernel<1>, dim3(1), dim3(64), 0, ctx->stream, ih.as<uint64_t>(), il.as<uint64_t>(), n, oh.as<uint64_t>(), ol.as<uint64_t>()); break;
        case 2: hipLaunchKernelGGL(dbg_sort_pair_kernel<2>, dim3(1), dim3(64), 0, ctx->stream, ih.as<uint64_t>(), il.as<uint64_t>(), n, oh.as<uint64_t>(), ol.as<uint64_t>()); break;
        case 4: hipLaunchKernelGGL(dbg_sort_pair_kernel<4>, dim3(1), dim3(64), 0, ctx->stream, ih.as<uint64_t>(), il.as<uint64_t>(), n, oh.as<uint64_t>(), ol.as<uint64_t>()); break;
        case 8: hipLaunchKernelGGL(dbg_sort_pair_kernel<8>, dim3(1), dim3(64), 0, ctx->stream, ih.as<uint64_t>(), il.as<uint64_t>(), n, oh.as<uint64_t>(), ol.as<uint64_t>()); break;
        default: GT_FAIL(ctx, GT_E_ARG, "nt");
    }
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GT_HIP(ctx, hipMemcpy(out_hi, oh.p, bytes, hipMemcpyDeviceToHost));
    GT_HIP(ctx, hipMemcpy(out_lo, ol.p, bytes, hipMemcpyDeviceToHost));
    ih.release();
    il.release();
    oh.release();
    ol.release();
    return GT_OK;
}

extern "C" int gt_dbg_mfma(gt_ctx* ctx, const float* a_host, const float* bt_host, int K, float* c_host) {
    if (!ctx) return GT_E_ARG;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf a, b, c;
    GT_HIP(ctx, a.reserve(size_t(32) * K * 4));
    GT_HIP(ctx, b.reserve(size_t(32) * K * 4));
    GT_HIP(ctx, c.reserve(size_t(32) * 32 * 4));
    GT_HIP(ctx, hipMemcpy(a.p, a_host, size_t(32) * K * 4, hipMemcpyHostToDevice));
    GT_HIP(ctx, hipMemcpy(b.p, bt_host, size_t(32) * K * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dbg_mfma_kernel, dim3(1), dim3(64), 0, ctx->stream, a.as<float>(), b.as<float>(), K, c.as<float>());
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GT_HIP(ctx, hipMemcpy(c_host, c.p, size_t(32) * 32 * 4, hipMemcpyDeviceToHost));
    a.release();
    b.release();
    c.release();
    return GT_OK;
}

// raw candidate lists of the most recent kNN call: counts [nq] and the first `width` keys of each list
extern "C" int gt_dbg_fetch_lists(gt_ctx* ctx, int64_t nq, int width, uint32_t* counts_host, uint64_t* keys_host) {
    if (!ctx || !ctx->knn) return GT_E_STATE;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    KnnWork* k = ctx->knn;
    const size_t lcap = size_t(64) * k->nt;
    GT_HIP(ctx, hipMemcpy(counts_host, k->counts.p, size_t(nq) * 4, hipMemcpyDeviceToHost));
    GT_HIP(ctx, hipMemcpy2D(keys_host, size_t(width) * 8, k->lists.p, lcap * 8, size_t(width) * 8, size_t(nq),
                            hipMemcpyDeviceToHost));
    return GT_OK;
}

extern "C" int gt_dbg_fetch_cand(gt_ctx* ctx, int64_t nq, double* d2_host, uint32_t* j_host, double* lb_host) {
    if (!ctx || !ctx->knn) return GT_E_STATE;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    KnnWork* k = ctx->knn;
    GT_HIP(ctx, hipMemcpy(d2_host, k->cand_d2.p, size_t(nq) * k->MP * 8, hipMemcpyDeviceToHost));
    GT_HIP(ctx, hipMemcpy(j_host, k->cand_j.p, size_t(nq) * k->MP * 4, hipMemcpyDeviceToHost));
    GT_HIP(ctx, hipMemcpy(lb_host, k->d2_lb.p, size_t(nq) * 8, hipMemcpyDeviceToHost));
    return GT_OK;
}

extern "C" int gt_dbg_fetch_prof(gt_ctx* ctx, int64_t nwaves, unsigned long long* out_host) {
    if (!ctx || !ctx->knn || !ctx->knn->prof.p) return GT_E_STATE;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    GT_HIP(ctx, hipMemcpy(out_host, ctx->knn->prof.p, size_t(nwaves) * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return GT_OK;
}

// development: raw buffers of the symmetric candidate pass (gt_sym.hip) of the most recent kNN call
//   which: 0 thr (float [n]) 1 perm (int32 [n]) 3 list lengths (uint32 [n]; may exceed the capacity) 13 table lengths (uint32 [n])
//          4 tile counts of launch A (int32 [blocks]) 5 sorted cell ids (uint32 [n])
extern "C" int gt_dbg_fetch_sym(gt_ctx* ctx, int32_t which, int64_t count, void* out_host) {
    if (!ctx || !ctx->knn) return GT_E_STATE;
    GT_HIP(ctx, hipSetDevice(ctx->device));
    KnnWork* k = ctx->knn;
    const void* src = nullptr;
    size_t esz = 4;
    switch (which) {
        case 0: src = k->thr_final.p; break;
        case 1: src = k->qorder.p; break;
        case 3: src = k->tcounts.p; break;
        case 4: src = k->sym_tile_cnt.p; break;
        case 5: src = ctx->order_cell.as<uint32_t>() + ctx->n; break;
        case 6: src = k->sym_tiles.p; break;
        case 8: src = k->sym_qcount.p; break;  // entries noted per wave region of the two-stage collect
        case 9: src = k->counts.p; break;      // rows launch A kept per sorted position
        case 10: src = k->sym_thrh.p; break;   // two-stage collect: partial-distance thresholds, half seeds, row radii term
        case 11: src = k->sym_hh.p; break;
        case 12: src = k->sym_gh.p; break;
        case 14: src = k->sym_zc.p; break;     // unit skipping of the two-stage collect: centres [n_pad / 32][16] (float)
        case 15: src = k->sym_zrn.p; break;    //   {radius, need} [n_pad / 32][2] (float)
        case 13: src = k->cand_n.p; break;     // entries of every exact table (by slot in builds with the tables by sorted position)
        case 16: src = k->lists.p; esz = 8; break;   // rows launch A kept: uint64 keys [n_pad][64] (cand_index = sorted position)
        case 7: src = k->sym_work.p; break;    // nbr [L][M] | start [L] | end [L]   // [blocks][tile_stride] tile lists of launch A (count = entries)
        default: return GT_E_ARG;
    }
    if (!src) return GT_E_STATE;
    GT_HIP(ctx, hipMemcpy(out_host, src, size_t(count) * esz, hipMemcpyDeviceToHost));
    return GT_OK;
}
