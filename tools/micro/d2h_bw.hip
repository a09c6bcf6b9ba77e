// Development: what the device -> host link of the box sustains (pinned destination), one copy and several concurrent streams.
// hipcc --offload-arch=gfx950 -O2 tools/micro/d2h_bw.hip -o /tmp/d2h_bw && /tmp/d2h_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
    const size_t total = size_t(1) << 30;
    char* dev = nullptr;
    char* host = nullptr;
    if (hipMalloc(&dev, total) != hipSuccess || hipHostMalloc(&host, total, hipHostMallocDefault) != hipSuccess) return 1;
    (void)hipMemset(dev, 1, total);
    (void)hipDeviceSynchronize();
    for (int ns : {1, 2, 4, 8, 16}) {
        std::vector<hipStream_t> st(ns);
        for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (size_t chunk : {size_t(8) << 20, size_t(64) << 20, total / ns}) {
            double best = 0.0;
            for (int rep = 0; rep < 3; ++rep) {
                const auto t0 = std::chrono::steady_clock::now();
                size_t off = 0;
                int k = 0;
                while (off < total) {
                    const size_t len = chunk < total - off ? chunk : total - off;
                    (void)hipMemcpyAsync(host + off, dev + off, len, hipMemcpyDeviceToHost, st[k % ns]);
                    off += len;
                    ++k;
                }
                for (auto& s : st) (void)hipStreamSynchronize(s);
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                const double gbs = double(total) / sec / 1e9;
                if (gbs > best) best = gbs;
            }
            std::printf("streams %2d chunk %5zu MiB: %.1f GB/s\n", ns, chunk >> 20, best);
        }
        for (auto& s : st) (void)hipStreamDestroy(s);
    }
    return 0;
}
