// Process-wide cache of large device allocations (see gt_common.h, DevBuf).
#include <algorithm>
#include <map>
#include <mutex>

#include "gt_common.h"

namespace {

constexpr size_t kMinPooled = size_t(1) << 20;
constexpr int kMaxDevices = 64;

struct Pool {
    std::mutex mu;
    std::multimap<size_t, void*> blocks[kMaxDevices];   // by size
    size_t cached[kMaxDevices] = {};
    // bytes that may stay parked per device: GT_POOL_MAX_GB, else an eighth of THAT device's memory (at most 32 GB) - the
    // rest of the process (torch, RCCL) cannot see what sits here.  Queried per device, with the device current.
    size_t lim[kMaxDevices] = {};
    bool lim_known[kMaxDevices] = {};
    size_t limit(int d) {   // (called with mu held and device d current)
        if (!lim_known[d]) {
            const char* s = std::getenv("GT_POOL_MAX_GB");
            size_t cap = size_t(32) << 30;
            if (s) {
                const double gb = std::atof(s);
                cap = gb <= 0 ? size_t(0) : size_t(gb * double(size_t(1) << 30));
            } else {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) cap = std::min(cap, total_b / 8);
            }
            lim[d] = cap;
            lim_known[d] = true;
        }
        return lim[d];
    }
};

Pool& pool() {
    static Pool* p = new Pool();   // never destroyed: the HIP runtime may be gone when static destructors run
    return *p;
}

int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return -1;
    return d;
}

void flush_device(Pool& P, int d) {
    for (auto& kv : P.blocks[d]) (void)hipFree(kv.second);
    P.blocks[d].clear();
    P.cached[d] = 0;
}

}  // namespace

hipError_t gt_pool_alloc(void** p, size_t bytes, size_t* got) {
    Pool& P = pool();
    const int d = current_device();
    if (d >= 0 && bytes >= kMinPooled) {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.blocks[d].lower_bound(bytes);
        if (it != P.blocks[d].end() && it->first <= 2 * bytes) {
            *p = it->second;
            *got = it->first;
            P.cached[d] -= it->first;
            P.blocks[d].erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && d >= 0) {
        // out of memory with blocks parked: give them back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(P.mu);
            flush_device(P, d);
        }
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) (void)hipGetLastError();   // handed to the caller as a return value: leave no stale error behind
    *got = bytes;
    return e;
}

void gt_pool_free(void* p, size_t bytes) {
    if (!p) return;
    Pool& P = pool();
    const int d = current_device();
    if (d >= 0 && bytes >= kMinPooled) {
        // hipFree would have waited for the device; a parked block can be handed to another context (another stream,
        // another thread) at once, so work still queued on it must be finished first.  Blocks are let go when a buffer
        // grows or a context closes - never inside the steady state of a build.
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lock(P.mu);
        if (P.cached[d] + bytes <= P.limit(d)) {
            P.blocks[d].emplace(bytes, p);
            P.cached[d] += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

extern "C" int gt_release_cached_memory(void) {
    Pool& P = pool();
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    std::lock_guard<std::mutex> lock(P.mu);
    for (int d = 0; d < kMaxDevices; ++d) {
        if (P.blocks[d].empty()) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        flush_device(P, d);
    }
    if (have_prev) (void)hipSetDevice(prev);
    return GT_OK;
}
