// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library (MI355X_MICROARCH.md:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Kernels with KNOWN traffic over a buffer far beyond the L2
// and the 256 MiB Infinity Cache:
//   stream16   every lane reads 16 B, coalesced                 ->  bytes = N x 16
//   gather16   every lane reads 16 B at a random 64-B slot      ->  one sector per lane: 64 B if the fabric request is a
//              64-B sector, 128 B if a whole L2 line is fetched - the counters cannot tell, the RATE can: more than
//              6.3 TB/s / 128 B = 49 G gathers/s is only possible with 64-B requests
//   gather8    the same with 8-byte loads (the bandwidth gathers of affinity_kernel)
//   gather16x4 four lanes share a 64-B slot (the row gathers of the re-rank: 4 x 16 B)
//   store16    every lane writes 16 B, coalesced                ->  WRITE_SIZE against N x 16
//   scatter16  every lane writes 16 B to a random 64-B slot     (the triplet scatters of the symmetrisation)
// usage: fetch_calib [GiB of buffer = 8] [reps = 3]; run plain for the rates, under `rocprofv3 --pmc FETCH_SIZE` /
// `--pmc WRITE_SIZE` / `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum` (separate passes) for the counters.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void stream16_kernel(const uint4* __restrict__ buf, const uint64_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int BYTES, int SHARE>   // SHARE lanes read consecutive pieces of one 64-byte slot
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ buf, const uint64_t slots, const uint64_t per_thread,
                                                      uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t t = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    for (uint64_t k = 0; k < per_thread; ++k) {
        const uint64_t slot = mix64((t / SHARE) * per_thread + k + 1) % slots;
        const char* p = buf + slot * 64 + (t % SHARE) * BYTES;
        if (BYTES == 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        } else {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            acc ^= v.x ^ v.y;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void store16_kernel(uint4* __restrict__ buf, const uint64_t n16) {
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256)
        buf[i] = make_uint4(uint32_t(i), 1u, 2u, 3u);
}

__global__ __launch_bounds__(256) void scatter16_kernel(char* __restrict__ buf, const uint64_t slots, const uint64_t per_thread) {
    const uint64_t t = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    for (uint64_t k = 0; k < per_thread; ++k) {
        const uint64_t slot = mix64(t * per_thread + k + 1) % slots;
        *reinterpret_cast<uint4*>(buf + slot * 64) = make_uint4(uint32_t(t), uint32_t(k), 2u, 3u);
    }
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const uint64_t bytes = uint64_t(gib * 1024.0 * 1024.0 * 1024.0) & ~uint64_t(4095);
    char* buf = nullptr;
    uint32_t* sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t n16 = bytes / 16, slots = bytes / 64;
    const unsigned grid = 256 * 32;
    const uint64_t threads = uint64_t(grid) * 256, per_thread = 128;   // 2.7e8 accesses per gather launch
    printf("{\"buffer_GiB\": %.2f, \"kernels\": {", gib);
    for (int kind = 0; kind < 6; ++kind) {
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            switch (kind) {
                case 0: hipLaunchKernelGGL(stream16_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, n16, sink); break;
                case 1: hipLaunchKernelGGL((gather_kernel<16, 1>), dim3(grid), dim3(256), 0, 0, buf, slots, per_thread, sink); break;
                case 2: hipLaunchKernelGGL((gather_kernel<8, 1>), dim3(grid), dim3(256), 0, 0, buf, slots, per_thread, sink); break;
                case 3: hipLaunchKernelGGL((gather_kernel<16, 4>), dim3(grid), dim3(256), 0, 0, buf, slots, per_thread, sink); break;
                case 4: hipLaunchKernelGGL(store16_kernel, dim3(grid), dim3(256), 0, 0, (uint4*)buf, n16); break;
                case 5: hipLaunchKernelGGL(scatter16_kernel, dim3(grid), dim3(256), 0, 0, buf, slots, per_thread); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const char* names[6] = {"stream16_kernel", "gather_kernel<16, 1>", "gather_kernel<8, 1>", "gather_kernel<16, 4>", "store16_kernel",
                                "scatter16_kernel"};
        const double accesses = (kind == 0 || kind == 4) ? double(n16) : (kind == 3 ? double(threads / 4) * per_thread : double(threads) * per_thread);
        const double useful = (kind == 0 || kind == 4) ? double(bytes) : double(threads) * per_thread * (kind == 2 ? 8.0 : 16.0);
        printf("%s\"%s\": {\"ms\": %.4f, \"distinct_slots_or_units\": %.0f, \"useful_bytes\": %.0f, \"G_units_per_s\": %.2f, "
               "\"TBps_if_64B_per_unit\": %.3f, \"TBps_if_128B_per_unit\": %.3f, \"useful_TBps\": %.3f}",
               kind ? ", " : "", names[kind], best, accesses, useful, accesses / best / 1e6, accesses * 64.0 / best / 1e9,
               accesses * 128.0 / best / 1e9, useful / best / 1e9);
    }
    printf("}}\n");
    return 0;
}
