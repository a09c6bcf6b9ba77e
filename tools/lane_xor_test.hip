#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int J>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v) {
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
        return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, true);
    } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);
    else if constexpr (J == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
__global__ void k(uint32_t* out) {
    const uint32_t v = threadIdx.x * 3 + 7;
    out[0 * 64 + threadIdx.x] = lane_xor32<1>(v);
    out[1 * 64 + threadIdx.x] = lane_xor32<2>(v);
    out[2 * 64 + threadIdx.x] = lane_xor32<4>(v);
    out[3 * 64 + threadIdx.x] = lane_xor32<8>(v);
    out[4 * 64 + threadIdx.x] = lane_xor32<16>(v);
    out[5 * 64 + threadIdx.x] = lane_xor32<32>(v);
}
int main() {
    uint32_t* d;
    hipMalloc(&d, 6 * 64 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[6 * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s = 0; s < 6; ++s)
        for (int l = 0; l < 64; ++l)
            if (h[s * 64 + l] != uint32_t((l ^ (1 << s)) * 3 + 7)) { if (bad < 10) printf("bad s=%d l=%d got %u\n", s, l, h[s*64+l]); ++bad; }
    printf("bad=%d\n", bad);
    return bad != 0;
}
