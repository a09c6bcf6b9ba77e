// Development microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 with NACC independent
// accumulators per wave and W waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int nacc, double flop_per_mfma, int wg_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    const int iters = 4000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 100, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = double(grid) * 4 * iters * 8 * nacc;
    printf("%s nacc=%d wg/cu=%d: %.3f ms, %.1f TF\n", name, nacc, wg_per_cu, ms, mf * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run("32x32x2", k32<1>, 1, 4096.0, w);
        run("32x32x2", k32<2>, 2, 4096.0, w);
        run("32x32x2", k32<4>, 4, 4096.0, w);
        run("16x16x4", k16<4>, 4, 2048.0, w);
        run("16x16x4", k16<8>, 8, 2048.0, w);
    }
    return 0;
}
