// Copies between PAGEABLE host memory (numpy arrays of the caller) and the device.
//
// hipMemcpy into a FRESH host array (np.empty: no page touched yet) moves 15-25 GB/s on the MI355X host - the
// runtime pins the destination page by page, single-threaded, taking the first-touch faults on the way - against
// 56 GB/s into resident memory.  The results of a graph build at N = 1e6 are 2.3 GB (CSR K
// and P), always copied into fresh arrays.  Here large device-to-host copies are cut into 8 MiB chunks and dealt to
// kLanes host threads; every lane owns a HIP stream and two pinned slots and overlaps the DMA of its next chunk
// with the memcpy of the current one from the slot into the caller's memory, where the page faults are then taken
// in parallel: 50 GB/s into fresh memory.  Small copies and host-to-device copies take the plain runtime path.
//
// The pinned slots and streams are created once per process and device (128 MiB pinned) and never freed: the HIP
// runtime may already be gone when static destructors run.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>

#include <sys/mman.h>
#include <unistd.h>

#include "gt_common.h"
#include "gt_hostcopy.h"

namespace {

constexpr int kMaxLanes = 32;
// development overrides: GT_COPY_LANES (0 = plain hipMemcpy for every size), GT_COPY_SLOT_MB
int env_int(const char* name, int dflt, int lo, int hi) {
    const char* v = std::getenv(name);
    if (!v) return dflt;
    const int x = std::atoi(v);
    return x < lo ? lo : x > hi ? hi : x;
}
const int kLanes = env_int("GT_COPY_LANES", 16, 0, kMaxLanes);
const size_t kSlotBytes = size_t(env_int("GT_COPY_SLOT_MB", 8, 1, 64)) << 20;
constexpr size_t kMinPipelined = size_t(32) << 20;
constexpr int kMaxDevices = 64;

struct Lane {
    void* slot[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
};

struct Pipe {
    Lane lanes[kMaxLanes];
    std::mutex busy;   // one pipelined copy at a time per device
};

std::mutex g_create;
Pipe* g_pipes[kMaxDevices];

hipError_t pipe_for(int device, Pipe** out) {
    std::lock_guard<std::mutex> lock(g_create);
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_pipes[device]) {
        Pipe* p = new Pipe();
        for (int l = 0; l < kLanes; ++l) {
            Lane& ln = p->lanes[l];
            hipError_t e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking);
            for (int s = 0; s < 2 && e == hipSuccess; ++s) {
                e = hipHostMalloc(&ln.slot[s], kSlotBytes, hipHostMallocDefault);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev[s], hipEventDisableTiming);
            }
            if (e != hipSuccess) return e;   // leaks the partial pipe; the caller reports the error
        }
        g_pipes[device] = p;
    }
    *out = g_pipes[device];
    return hipSuccess;
}

// chunk c of the copy covers bytes [c * kSlotBytes, min(bytes, (c+1) * kSlotBytes)); lane l takes c = l, l + kLanes, ...
// post (optional): called by the lane's thread for every chunk once it sits in the caller's memory (byte offset, length) -
// host work that rides along with the transfer (gt_fetch_kp_host: P = K / degree while the next chunks are on the link)
typedef std::function<void(size_t, size_t)> ChunkFn;

void lane_d2h(int device, Lane* ln, int lane, char* dst, const char* src, size_t bytes, std::atomic<int>* err,
              const ChunkFn* post) {
    hipError_t e = hipSetDevice(device);
    const size_t nchunks = (bytes + kSlotBytes - 1) / kSlotBytes;
    auto issue = [&](size_t c, int s) {
        const size_t off = c * kSlotBytes, len = std::min(kSlotBytes, bytes - off);
        hipError_t r = hipMemcpyAsync(ln->slot[s], src + off, len, hipMemcpyDeviceToHost, ln->stream);
        if (r == hipSuccess) r = hipEventRecord(ln->ev[s], ln->stream);
        return r;
    };
    int s = 0;
    if (e == hipSuccess && size_t(lane) < nchunks) e = issue(size_t(lane), 0);
    for (size_t c = size_t(lane); c < nchunks && e == hipSuccess; c += kLanes, s ^= 1) {
        if (c + kLanes < nchunks) e = issue(c + kLanes, s ^ 1);   // its previous content was consumed one step ago
        if (e == hipSuccess) e = hipEventSynchronize(ln->ev[s]);
        if (e != hipSuccess) break;
        const size_t off = c * kSlotBytes, len = std::min(kSlotBytes, bytes - off);
        std::memcpy(dst + off, ln->slot[s], len);
        if (post) (*post)(off, len);
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ln->stream);
        err->store(int(e));
    }
}

int pipelined_d2h(gt_ctx* ctx, void* dst, const void* src, size_t bytes, const ChunkFn* post = nullptr) {
    Pipe* p = nullptr;
    GT_HIP(ctx, pipe_for(ctx->device, &p));
    std::lock_guard<std::mutex> lock(p->busy);
    std::atomic<int> err(0);
    std::thread th[kMaxLanes];
    const int nl = int(std::min<size_t>(size_t(kLanes), (bytes + kSlotBytes - 1) / kSlotBytes));
    int started = 0;
    try {
        for (; started < nl; ++started)
            th[started] = std::thread(lane_d2h, ctx->device, &p->lanes[started], started, static_cast<char*>(dst),
                                      static_cast<const char*>(src), bytes, &err, post);
    } catch (const std::system_error&) {
        // the process cannot start another thread: this thread takes over the lanes that did not get one
        for (int l = started; l < nl; ++l)
            lane_d2h(ctx->device, &p->lanes[l], l, static_cast<char*>(dst), static_cast<const char*>(src), bytes, &err, post);
    }
    for (int l = 0; l < started; ++l) th[l].join();
    if (err.load() != 0) {
        ctx->set_error(std::string("pipelined host copy: ") + hipGetErrorString(hipError_t(err.load())));
        return GT_E_HIP;
    }
    return GT_OK;
}

// RESIDENT destinations (round 5).  The lanes exist because the runtime's copy into untouched pages is slow; into pages that
// are already resident a plain hipMemcpy runs at the link's rate (measured on the MI355X host: 56.0 GB/s into a resident
// malloc'd gigabyte, 57.1 into registered or hipHostMalloc'd memory, 53 in 8 MiB pieces - the lanes, with their staging memcpy,
// reach 46).  The Python binding recycles the big result arrays (graphtools_amd/_hip.py `_HostPool`), so from the second
// graph of a process on the destinations ARE resident: the copy then goes direct, in 64 MiB pieces when host work rides along
// (worker threads take the pieces as they land).  Residency is sampled with mincore() - one page every 2 MiB and the last.
bool dst_resident(const void* p, size_t bytes) {
    static const bool off = std::getenv("GT_COPY_DIRECT") != nullptr && std::atoi(std::getenv("GT_COPY_DIRECT")) == 0;
    if (off || bytes == 0) return false;
    const size_t page = size_t(sysconf(_SC_PAGESIZE));
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(p) & ~(uintptr_t(page) - 1);
    const uintptr_t a1 = (reinterpret_cast<uintptr_t>(p) + bytes - 1) & ~(uintptr_t(page) - 1);
    unsigned char v = 0;
    for (uintptr_t a = a0; a <= a1; a += (size_t(2) << 20)) {
        if (mincore(reinterpret_cast<void*>(a), page, &v) != 0 || !(v & 1)) return false;
    }
    if (mincore(reinterpret_cast<void*>(a1), page, &v) != 0 || !(v & 1)) return false;
    return true;
}

int direct_d2h(gt_ctx* ctx, void* dst, const void* src, size_t bytes, const ChunkFn* post) {
    if (std::getenv("GT_TRACE")) std::fprintf(stderr, "[gt_trace] direct device -> host copy of %.1f MB%s\n", double(bytes) / 1e6, post ? " (host work rides along)" : "");
    if (!post) {
        GT_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return GT_OK;
    }
    // copies of 64 MiB; the host work is dealt in 4 MiB items (whoever is free takes the next one that has landed): behind the
    // last copy there is a sixteenth of a piece left per worker, not a whole piece for one of them
    static const size_t piece = size_t(env_int("GT_COPY_PIECE_MB", 64, 8, 4096)) << 20;
    const size_t item = size_t(4) << 20;
    const size_t npieces = (bytes + piece - 1) / piece, nitems = (bytes + item - 1) / item;
    std::mutex m;
    std::condition_variable cv;
    size_t landed = 0;     // bytes that have landed
    bool failed = false;
    std::atomic<size_t> next(0);
    // (six workers: P = K / degree over a C3 graph behind the copy took 17.8 ms with 4 or 8 of them, 18.5-19.6 with 16, 19.0 with
    //  32 - the copy alone is 16.6 ms at the link's rate, and threads beyond what the arithmetic needs only get in its way)
    static const int kWorkers = env_int("GT_COPY_WORKERS", 6, 1, kMaxLanes);
    const int nw = int(std::min<size_t>(size_t(kWorkers), nitems));
    std::thread th[kMaxLanes];
    auto worker = [&]() {
        for (;;) {
            const size_t it = next.fetch_add(1);
            if (it >= nitems) return;
            const size_t off = it * item, len = std::min(item, bytes - off);
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return landed >= off + len || failed; });
                if (failed) return;
            }
            (*post)(off, len);
        }
    };
    int started = 0;
    try {
        for (; started < nw; ++started) th[started] = std::thread(worker);
    } catch (const std::system_error&) {
    }
    hipError_t e = hipSuccess;
    for (size_t c = 0; c < npieces && e == hipSuccess; ++c) {
        const size_t off = c * piece, len = std::min(piece, bytes - off);
        e = hipMemcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost);
        {
            std::lock_guard<std::mutex> lock(m);
            if (e == hipSuccess) landed = off + len;
            else failed = true;
        }
        cv.notify_all();
    }
    if (e == hipSuccess) worker();   // (this thread helps with what is left - all of it when no worker could be started)
    for (int l = 0; l < started; ++l) th[l].join();
    if (e != hipSuccess) {
        ctx->set_error(std::string("direct host copy: ") + hipGetErrorString(e));
        return GT_E_HIP;
    }
    return GT_OK;
}

}  // namespace

int gt_copy_to_host(gt_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (bytes == 0) return GT_OK;
    if (bytes >= kMinPipelined && dst_resident(dst_host, bytes)) {
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the producer kernels of src_dev
        return direct_d2h(ctx, dst_host, src_dev, bytes, nullptr);
    }
    if (bytes < kMinPipelined || kLanes == 0) {
        GT_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GT_OK;
    }
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the producer kernels of src_dev
    return pipelined_d2h(ctx, dst_host, src_dev, bytes);
}

// K (values) to the host with P derived on the way: P[e] = K[e] / degree[row of e] - the division the device made when it
// wrote its own P (compact_kernel / merge_final_kernel: v / asum, asum = the row sum in the degrees' summation order; IEEE
// division on both sides: the same bits, tests/test_gpu_dropin.py) - computed by the copy lanes on every chunk of K as it
// lands, while later chunks are still on the link.  The 0.93 GB of P values of a C3 graph then never cross PCIe (a third of
// the host-complete build's transfer).  indptr / degree: host arrays already fetched; *negative: set when a value < 0 was
// seen (|v| sums differ from the degrees then: the caller fetches P itself).
int gt_fetch_kp_host(gt_ctx* ctx, double* K_host, double* P_host, const double* K_dev, long long nnz_, const long long* indptr_,
                     const double* degree, long long nrows_, int* negative) {
    const int64_t nnz = nnz_, nrows = nrows_;
    const int64_t* indptr = reinterpret_cast<const int64_t*>(indptr_);
    *negative = 0;
    if (nnz <= 0) return GT_OK;
    std::atomic<int> neg(0);
    const ChunkFn post = [&](size_t off, size_t len) {
        const int64_t e0 = int64_t(off / sizeof(double)), e1 = e0 + int64_t(len / sizeof(double));
        int64_t row = int64_t(std::upper_bound(indptr, indptr + nrows + 1, e0) - indptr) - 1;   // indptr[row] <= e0 < indptr[row + 1]
        bool bad = false;
        for (int64_t e = e0; e < e1;) {
            while (row + 1 <= nrows && indptr[row + 1] <= e) ++row;
            const int64_t end = std::min<int64_t>(e1, indptr[row + 1]);
            const double s = degree[row];
            if (s != 0.0) {
                for (; e < end; ++e) {
                    const double v = K_host[e];
                    bad |= v < 0.0;
                    P_host[e] = v / s;
                }
            } else {
                for (; e < end; ++e) P_host[e] = K_host[e];
            }
        }
        if (bad) neg.store(1);
    };
    const size_t bytes = size_t(nnz) * sizeof(double);
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes >= kMinPipelined && dst_resident(K_host, bytes) && dst_resident(P_host, bytes)) {
        GT_TRY(direct_d2h(ctx, K_host, K_dev, bytes, &post));
    } else if (bytes < kMinPipelined || kLanes == 0) {
        GT_HIP(ctx, hipMemcpy(K_host, K_dev, bytes, hipMemcpyDeviceToHost));
        post(0, bytes);
    } else {
        GT_TRY(pipelined_d2h(ctx, K_host, K_dev, bytes, &post));
    }
    *negative = neg.load();
    return GT_OK;
}

namespace {

// host -> device through the lanes' pinned slots: the lane's thread copies a chunk of the caller's (pageable) array into a
// slot while the DMA of its previous chunk is still on the link.  A plain hipMemcpy of a pageable numpy array moved 28 GB/s
// (256 MB of points: 9 ms of the host-complete build).
void lane_h2d(int device, Lane* ln, int lane, char* dst, const char* src, size_t bytes, std::atomic<int>* err) {
    hipError_t e = hipSetDevice(device);
    const size_t nchunks = (bytes + kSlotBytes - 1) / kSlotBytes;
    bool used[2] = {false, false};
    int s = 0;
    for (size_t c = size_t(lane); c < nchunks && e == hipSuccess; c += kLanes, s ^= 1) {
        if (used[s]) e = hipEventSynchronize(ln->ev[s]);   // the DMA that last read this slot
        if (e != hipSuccess) break;
        const size_t off = c * kSlotBytes, len = std::min(kSlotBytes, bytes - off);
        std::memcpy(ln->slot[s], src + off, len);
        e = hipMemcpyAsync(dst + off, ln->slot[s], len, hipMemcpyHostToDevice, ln->stream);
        if (e == hipSuccess) e = hipEventRecord(ln->ev[s], ln->stream);
        used[s] = true;
    }
    const hipError_t e2 = hipStreamSynchronize(ln->stream);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) err->store(int(e));
}

int pipelined_h2d(gt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    Pipe* p = nullptr;
    GT_HIP(ctx, pipe_for(ctx->device, &p));
    std::lock_guard<std::mutex> lock(p->busy);
    std::atomic<int> err(0);
    std::thread th[kMaxLanes];
    const int nl = int(std::min<size_t>(size_t(kLanes), (bytes + kSlotBytes - 1) / kSlotBytes));
    int started = 0;
    try {
        for (; started < nl; ++started)
            th[started] = std::thread(lane_h2d, ctx->device, &p->lanes[started], started, static_cast<char*>(dst),
                                      static_cast<const char*>(src), bytes, &err);
    } catch (const std::system_error&) {
        for (int l = started; l < nl; ++l)
            lane_h2d(ctx->device, &p->lanes[l], l, static_cast<char*>(dst), static_cast<const char*>(src), bytes, &err);
    }
    for (int l = 0; l < started; ++l) th[l].join();
    if (err.load() != 0) {
        ctx->set_error(std::string("pipelined host-to-device copy: ") + hipGetErrorString(hipError_t(err.load())));
        return GT_E_HIP;
    }
    return GT_OK;
}

}  // namespace

int gt_copy_from_host(gt_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (bytes == 0) return GT_OK;
    // (measured at 256 MB from a resident numpy array: the plain copy 9 ms, the lanes 17 ms - the threads' staging memcpy
    //  costs more than the runtime's own pinned staging saves; kept behind GT_H2D_PIPELINED=1 for other hosts)
    if (bytes >= kMinPipelined && kLanes > 0 && std::getenv("GT_H2D_PIPELINED") != nullptr) {
        GT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (whatever still reads the destination)
        return pipelined_h2d(ctx, dst_dev, src_host, bytes);
    }
    GT_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    GT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GT_OK;
}

// ---- host helper of the MNN composition (graphs.py MNNGraph._assemble_kernel0) --------------------------------------
// Copies the rows of one CSR block (batch-local ids) into their rows of the assembled kernel: entry e of local row r
// goes to cursor[rows_global[r]] + (e - indptr[r]), its column becomes cols_global[indices[e]], its value is scaled
// by scale[r] when given.  cursor[] is advanced by the row lengths.  Rows are independent: split over host threads.
extern "C" int gt_host_place_block(int64_t nrows, const int64_t* indptr, const int32_t* indices, const double* data,
                                   const int64_t* rows_global, const int64_t* cols_global, const double* scale,
                                   int64_t* cursor, int32_t* out_indices, double* out_data) {
    if (nrows < 0 || !indptr || !rows_global || !cols_global || !cursor || !out_indices || !out_data) return GT_E_ARG;
    auto work = [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t g = rows_global[r];
            int64_t dst = cursor[g];
            const double s = scale ? scale[r] : 1.0;
            for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e, ++dst) {
                out_indices[dst] = int32_t(cols_global[indices[e]]);
                out_data[dst] = scale ? data[e] * s : data[e];
            }
            cursor[g] = dst;
        }
    };
    const int64_t nnz = indptr[nrows] - indptr[0];
    int nt = int(std::min<int64_t>(16, std::max<int64_t>(1, nnz / (int64_t(1) << 20))));
    if (nt <= 1) {
        work(0, nrows);
        return GT_OK;
    }
    std::thread th[16];
    int started = 0;
    try {
        for (; started < nt; ++started)
            th[started] = std::thread(work, nrows * started / nt, nrows * (started + 1) / nt);
    } catch (const std::system_error&) {
        work(nrows * started / nt, nrows);
    }
    for (int t = 0; t < started; ++t) th[t].join();
    return GT_OK;
}
