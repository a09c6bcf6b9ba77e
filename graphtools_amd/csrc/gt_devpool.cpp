// Process-wide cache of large device allocations (see gt_common.h, DevBuf).
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "gt_common.h"

namespace {

// Every size is pooled: a fresh context reserves ~100 buffers and the small ones (counters, per-bin tables) cost 20-300 us each
// through hipMalloc - 2.5 ms of a 21 ms build on a new context (round 4, tools/host_complete_probe.py).  Small requests are
// rounded up to 4 KiB so that they find each other again.
constexpr size_t kMinPooled = 1;
constexpr size_t kSmallRound = 4096;
thread_local int t_quiesced = 0;   // > 0: the caller has synchronised the device for a batch of releases (gt_pool_quiesced_*)
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

void gt_pool_quiesced_begin() {
    (void)hipDeviceSynchronize();
    ++t_quiesced;
}
void gt_pool_quiesced_end() { --t_quiesced; }

hipError_t gt_pool_alloc(void** p, size_t bytes, size_t* got) {
    Pool& P = pool();
    const int d = current_device();
    if (bytes < kSmallRound) bytes = kSmallRound;
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
        if (t_quiesced <= 0) (void)hipDeviceSynchronize();
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

// ---- HIP handles (streams, events) parked per device ---------------------------------------------------------------------
// A context makes a stream, a high-priority side stream and ~150 events (two per stage span); created afresh they cost a new
// context ~2.5 ms of its first build (round 4: 21 ms against 18.4 on a warm context).  A closing context - synchronised - parks
// them here, the next one takes them over.
namespace {
struct Handles {
    std::mutex mu;
    std::vector<hipStream_t> streams[kMaxDevices], side_streams[kMaxDevices];
    std::vector<hipEvent_t> events[kMaxDevices];
};
Handles& handles() {
    static Handles* h = new Handles();
    return *h;
}
}  // namespace

hipStream_t gt_handle_take_stream(int device, bool side) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    Handles& H = handles();
    std::lock_guard<std::mutex> lock(H.mu);
    auto& v = side ? H.side_streams[device] : H.streams[device];
    if (v.empty()) return nullptr;
    hipStream_t s = v.back();
    v.pop_back();
    return s;
}

void gt_handle_park_stream(int device, bool side, hipStream_t s) {
    if (!s) return;
    if (device >= 0 && device < kMaxDevices) {
        Handles& H = handles();
        std::lock_guard<std::mutex> lock(H.mu);
        auto& v = side ? H.side_streams[device] : H.streams[device];
        if (v.size() < 16) {
            v.push_back(s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

hipEvent_t gt_handle_take_event(int device) {
    if (device >= 0 && device < kMaxDevices) {
        Handles& H = handles();
        std::lock_guard<std::mutex> lock(H.mu);
        if (!H.events[device].empty()) {
            hipEvent_t e = H.events[device].back();
            H.events[device].pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void gt_handle_park_events(int device, std::vector<hipEvent_t>& ev) {
    if (device >= 0 && device < kMaxDevices) {
        Handles& H = handles();
        std::lock_guard<std::mutex> lock(H.mu);
        while (!ev.empty() && H.events[device].size() < 8192) {
            H.events[device].push_back(ev.back());
            ev.pop_back();
        }
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear();
}
