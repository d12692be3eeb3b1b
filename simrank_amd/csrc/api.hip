// Plumbing half of the C ABI: errors, device/memory/stream/event wrappers, graph upload,
// tuning knobs.  Kernels live in spmm.hip (sparse legs, evidence, identity) and dense.hip
// (densify + f32 MFMA GEMM).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace simrank {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

Tuning& tuning() {
    static Tuning t;
    return t;
}

static std::mutex& tuning_mutex() {
    static std::mutex m;
    return m;
}

Tuning tuning_snapshot() {
    std::lock_guard<std::mutex> lock(tuning_mutex());
    return tuning();
}

}  // namespace simrank


namespace simrank {
// ---- balanced tiling: 32-row blocks whose cost (entries) exceeds `balance` times the mean
// are cut into aligned halves, down to single rows, so that no wave is left with a tile
// many times the others' (a power-law row order sorted by length puts all long rows
// in a few blocks).  For the upper-triangle leg 2: the (panel, workgroup) launch list.
// Appends the launch order of the plain legs to tile_row0 (after its n_tiles + 1 entries): the groups
// of four tiles a workgroup takes, most entries first.  In ascending row-length order that is the
// reverse order; when the nodes were dealt to shards the long rows sit at the end of every shard.
static void append_group_order(const int32_t* rowptr, std::vector<int32_t>& tile_row0, int64_t n_tiles) {
    const int64_t groups = (n_tiles + 3) / 4;
    std::vector<std::pair<int64_t, int32_t>> key((size_t)groups);
    for (int64_t gidx = 0; gidx < groups; ++gidx) {
        const int64_t lo = tile_row0[(size_t)(4 * gidx)];
        const int64_t hi = tile_row0[(size_t)std::min<int64_t>(4 * gidx + 4, n_tiles)];
        key[(size_t)gidx] = {-(int64_t)(rowptr[hi] - rowptr[lo]), (int32_t)(groups - 1 - gidx)};
    }
    std::sort(key.begin(), key.end());
    for (const auto& k : key) tile_row0.push_back((int32_t)(groups - 1 - k.second));
}

int64_t build_tiles(const int32_t* rowptr, int64_t n_rows, int64_t nnz, int64_t balance,
                    std::vector<int32_t>& tile_row0, std::vector<int32_t>& sym_map, bool sym_descending) {
    tile_row0.clear();
    sym_map.clear();
    if (balance <= 0 || nnz <= 0) return 0;
    const int64_t nblk = (n_rows + 31) / 32;
    const int64_t limit = std::max<int64_t>(balance * ((nnz + nblk - 1) / nblk), 256);
    std::vector<std::pair<int64_t, int64_t>> stack;
    for (int64_t b = 0; b < nblk; ++b) {
        stack.clear();
        stack.emplace_back(b * 32, std::min<int64_t>(n_rows, b * 32 + 32));
        while (!stack.empty()) {
            const auto [lo, hi] = stack.back();
            stack.pop_back();
            if (rowptr[hi] - rowptr[lo] <= limit || hi - lo <= 1) {
                tile_row0.push_back((int32_t)lo);
            } else {
                const int64_t mid = lo + (hi - lo + 1) / 2;
                stack.emplace_back(mid, hi);       // popped second: tiles stay in row order
                stack.emplace_back(lo, mid);
            }
        }
    }
    tile_row0.push_back((int32_t)n_rows);
    const int64_t n_tiles = (int64_t)tile_row0.size() - 1;
    append_group_order(rowptr, tile_row0, n_tiles);
    if (n_rows < 64) return n_tiles;
    std::vector<std::vector<int32_t>> lists(8);
    size_t t_end = 0;                            // tiles with row0 < 32 (p + 1)
    std::vector<int64_t> groups_of((size_t)nblk);
    for (int64_t pnl = 0; pnl < nblk; ++pnl) {
        while (t_end < (size_t)n_tiles && tile_row0[t_end] < 32 * (pnl + 1)) ++t_end;
        groups_of[(size_t)pnl] = ((int64_t)t_end + 3) / 4;
    }
    // panel p needs the row groups up to its own: the late panels are the big ones.  Descending: an XCD
    // starts with panels of ~nblk/4 workgroups (one panel at a time in its L2) and ends on the small ones
    for (int64_t k = 0; k < nblk; ++k) {
        const int64_t pnl = sym_descending ? nblk - 1 - k : k;
        std::vector<int32_t>& l = lists[size_t(pnl & 7)];
        for (int64_t rt = groups_of[(size_t)pnl] - 1; rt >= 0; --rt) {     // heavy (late) groups first
            l.push_back((int32_t)pnl);
            l.push_back((int32_t)rt);
        }
    }
    size_t longest = 0;
    for (const auto& l : lists) longest = std::max(longest, l.size() / 2);
    sym_map.assign(longest * 8 * 2, -1);
    for (size_t x = 0; x < 8; ++x)
        for (size_t k = 0; k < lists[x].size() / 2; ++k) {
            sym_map[2 * (8 * k + x)] = lists[x][2 * k];
            sym_map[2 * (8 * k + x) + 1] = lists[x][2 * k + 1];
        }
    return n_tiles;
}

// ---- device block pool.  hipMalloc maps a 17 GiB matrix in 0.4-0.5 s on some boxes (1.6-2.0 s of a config-5
// set-up against 0.12 s for its four updates: profiles/r03_setup_before_pool.log), so blocks of at least
// kPoolMin bytes are kept when their owner lets go of them and handed to the next request they fit (smallest
// cached block of at least the size asked for and at most 1/8 larger).  Per device at most SIMRANK_POOL_GIB
// (default 56: one config-5 plan) GiB at rest, least recently freed blocks leave first; a failed hipMalloc anywhere behind this
// interface empties the pool and tries once more, so cached blocks never cause an out-of-memory error the
// process would not have had without them.  Cached memory is invisible to other allocators of the process
// (torch's caching allocator, the caller's own hipMalloc): simrank_pool_trim() hands it back.
// (round 6) SMALL blocks — the few dozen arrays of a graph object and its plans, the counters, the tickets — are kept too, in
// size classes (2^k x {1, 1.25, 1.5, 1.75}, from 512 bytes): a fit made ~60 hipMalloc / hipFree pairs of them, 5 ms of the
// 60 ms of a MovieLens-shaped fit in the frees alone.  At most kSmallLimit bytes at rest per device; simrank_pool_trim
// returns them with the rest.
namespace {
constexpr size_t kPoolMin = size_t(64) << 20;
constexpr size_t kSmallLimit = size_t(1) << 30;
constexpr int kSmallClasses = 4 * 40;
inline size_t small_class_bytes(int c) { return (size_t(4 + (c & 3)) << (c >> 2)) * 128; }      // class 0 = 512 bytes
inline int small_class_of(size_t bytes) {
    int c = 0;
    while (small_class_bytes(c) < bytes) ++c;
    return c;
}
struct PoolBlock { void* ptr; size_t bytes; uint64_t age; };
struct PoolState {
    std::mutex m;
    std::vector<std::vector<PoolBlock>> cached;      // per device
    std::vector<size_t> cached_bytes;
    std::vector<std::vector<std::vector<void*>>> small;   // per device and size class
    std::vector<size_t> small_bytes;
    std::unordered_map<void*, std::pair<size_t, int>> live;   // blocks handed out: ptr -> (bytes, device)
    uint64_t clock = 0;
    size_t limit = size_t(56) << 30;      // what one config-5 plan holds: three 17 GiB matrices + its counts
    PoolState() {
        if (const char* e = getenv("SIMRANK_POOL_GIB")) limit = size_t(std::max(0ll, atoll(e))) << 30;
    }
    void grow(int dev) {
        if ((int)cached.size() <= dev) {
            cached.resize(size_t(dev) + 1);
            cached_bytes.resize(size_t(dev) + 1, 0);
            small.resize(size_t(dev) + 1);
            small_bytes.resize(size_t(dev) + 1, 0);
        }
        if (small[size_t(dev)].empty()) small[size_t(dev)].resize(kSmallClasses);
    }
};
PoolState& pool() {
    static PoolState* p = new PoolState;     // (never destroyed: frees at exit would race the runtime's teardown)
    return *p;
}
void pool_trim_locked(PoolState& P, int device) {
    int restore = -1;
    (void)hipGetDevice(&restore);
    for (int d = 0; d < (int)P.cached.size(); ++d) {
        if (device >= 0 && d != device) continue;
        if (P.cached[size_t(d)].empty()) continue;
        (void)hipSetDevice(d);
        for (PoolBlock& b : P.cached[size_t(d)]) (void)hipFree(b.ptr);
        P.cached[size_t(d)].clear();
        P.cached_bytes[size_t(d)] = 0;
    }
    for (int d = 0; d < (int)P.small.size(); ++d) {
        if ((device >= 0 && d != device) || P.small_bytes[size_t(d)] == 0) continue;
        (void)hipSetDevice(d);
        for (std::vector<void*>& l : P.small[size_t(d)]) {
            for (void* q : l) (void)hipFree(q);
            l.clear();
        }
        P.small_bytes[size_t(d)] = 0;
    }
    if (restore >= 0) (void)hipSetDevice(restore);
}
}  // namespace

static int pool_alloc_raw(void** dptr, size_t bytes) {
    *dptr = nullptr;
    PoolState& P = pool();
    int dev = 0;
    SR_HIP(hipGetDevice(&dev));
    if (bytes >= kPoolMin) {
        std::lock_guard<std::mutex> lock(P.m);
        P.grow(dev);
        std::vector<PoolBlock>& c = P.cached[size_t(dev)];
        int best = -1;
        for (int i = 0; i < (int)c.size(); ++i)
            if (c[size_t(i)].bytes >= bytes && c[size_t(i)].bytes - bytes <= bytes / 8 &&
                (best < 0 || c[size_t(i)].bytes < c[size_t(best)].bytes))
                best = i;
        if (best >= 0) {
            const PoolBlock b = c[size_t(best)];
            c.erase(c.begin() + best);
            P.cached_bytes[size_t(dev)] -= b.bytes;
            P.live[b.ptr] = {b.bytes, dev};
            *dptr = b.ptr;
            return SIMRANK_OK;
        }
    } else if (P.limit > 0) {
        const int cls = small_class_of(std::max<size_t>(bytes, 1));
        bytes = small_class_bytes(cls);
        std::lock_guard<std::mutex> lock(P.m);
        P.grow(dev);
        std::vector<void*>& l = P.small[size_t(dev)][size_t(cls)];
        if (!l.empty()) {
            *dptr = l.back();
            l.pop_back();
            P.small_bytes[size_t(dev)] -= bytes;
            P.live[*dptr] = {bytes, dev};
            return SIMRANK_OK;
        }
    }
    hipError_t e = hipMalloc(dptr, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(P.m);
            pool_trim_locked(P, dev);
        }
        e = hipMalloc(dptr, bytes);
    }
    if (e != hipSuccess) {
        *dptr = nullptr;
        set_error("hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? SIMRANK_ERR_ALLOC : SIMRANK_ERR_HIP;
    }
    if (P.limit > 0 || bytes >= kPoolMin) {
        std::lock_guard<std::mutex> lock(P.m);
        P.live[*dptr] = {bytes, dev};
    }
    return SIMRANK_OK;
}

// SIMRANK_POOL_POISON=1 (tests): every block handed out is filled with 0xFF bytes first — NaN as f32 and as fp16, 255 as
// a count — so that a kernel which lets padding rows / columns of an un-zeroed matrix reach a sum, a count, a top-k
// or a store shows (fresh hipMalloc memory is usually zero and hides it; advisor, round 4)
static bool pool_poison() {
    static const bool on = [] { const char* e = getenv("SIMRANK_POOL_POISON"); return e && *e && *e != '0'; }();
    return on;
}

int pool_alloc(void** dptr, size_t bytes) {
    const int rc = pool_alloc_raw(dptr, bytes);
    if (rc == SIMRANK_OK && pool_poison() && *dptr) {
        SR_HIP(hipMemset(*dptr, 0xFF, bytes));
        SR_HIP(hipDeviceSynchronize());
    }
    return rc;
}

int pool_free(void* ptr) {
    if (!ptr) return SIMRANK_OK;
    PoolState& P = pool();
    size_t bytes = 0;
    int dev = -1;
    {
        std::lock_guard<std::mutex> lock(P.m);
        auto it = P.live.find(ptr);
        if (it != P.live.end()) {
            bytes = it->second.first;
            dev = it->second.second;
            P.live.erase(it);
        }
    }
    if (bytes && bytes < kPoolMin && P.limit > 0) {
        // a small block: back to its size class (as below: nothing queued on the device may still use it)
        int cur = 0;
        SR_HIP(hipGetDevice(&cur));
        if (cur != dev) SR_HIP(hipSetDevice(dev));
        const hipError_t e = hipDeviceSynchronize();
        if (cur != dev) (void)hipSetDevice(cur);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lock(P.m);
            P.grow(dev);
            if (P.small_bytes[size_t(dev)] + bytes <= kSmallLimit) {
                P.small[size_t(dev)][size_t(small_class_of(bytes))].push_back(ptr);
                P.small_bytes[size_t(dev)] += bytes;
                return SIMRANK_OK;
            }
        } else {
            (void)hipGetLastError();
        }
        SR_HIP(hipFree(ptr));
        return SIMRANK_OK;
    }
    if (bytes && bytes <= P.limit) {
        // like hipFree: nothing queued anywhere on the device may still use the block when the next owner gets it
        int cur = 0;
        SR_HIP(hipGetDevice(&cur));
        if (cur != dev) SR_HIP(hipSetDevice(dev));
        const hipError_t e = hipDeviceSynchronize();
        if (cur != dev) (void)hipSetDevice(cur);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lock(P.m);
            P.grow(dev);
            std::vector<PoolBlock>& c = P.cached[size_t(dev)];
            while (!c.empty() && P.cached_bytes[size_t(dev)] + bytes > P.limit) {      // least recently freed first
                size_t oldest = 0;
                for (size_t i = 1; i < c.size(); ++i)
                    if (c[i].age < c[oldest].age) oldest = i;
                if (cur != dev) (void)hipSetDevice(dev);
                (void)hipFree(c[oldest].ptr);
                if (cur != dev) (void)hipSetDevice(cur);
                P.cached_bytes[size_t(dev)] -= c[oldest].bytes;
                c.erase(c.begin() + (long)oldest);
            }
            c.push_back({ptr, bytes, ++P.clock});
            P.cached_bytes[size_t(dev)] += bytes;
            return SIMRANK_OK;
        }
        (void)hipGetLastError();
    }
    SR_HIP(hipFree(ptr));
    return SIMRANK_OK;
}

void pool_trim(int device) {
    PoolState& P = pool();
    std::lock_guard<std::mutex> lock(P.m);
    pool_trim_locked(P, device);
}

void pool_stats(int device, int64_t* cached_bytes, int64_t* cached_blocks, int64_t* limit_bytes) {
    PoolState& P = pool();
    std::lock_guard<std::mutex> lock(P.m);
    int64_t b = 0, n = 0;
    for (int d = 0; d < (int)P.cached.size(); ++d)
        if (device < 0 || d == device) {
            b += (int64_t)P.cached_bytes[size_t(d)] + (int64_t)P.small_bytes[size_t(d)];
            n += (int64_t)P.cached[size_t(d)].size();
            for (const std::vector<void*>& l : P.small[size_t(d)]) n += (int64_t)l.size();
        }
    if (cached_bytes) *cached_bytes = b;
    if (cached_blocks) *cached_blocks = n;
    if (limit_bytes) *limit_bytes = (int64_t)P.limit;
}
}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_abi_version(void) { return SIMRANK_ABI_VERSION; }
const char* simrank_last_error(void) { return g_err; }

int simrank_device_count(int* count) {
    SR_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return SIMRANK_ERR_NO_DEVICE;
    }
    *count = n;
    return SIMRANK_OK;
}

int simrank_set_device(int device) {
    SR_HIP(hipSetDevice(device));
    return SIMRANK_OK;
}

int simrank_device_info(int device, char* name, int name_len, int64_t* total_bytes,
                        int* compute_units, char* arch, int arch_len) {
    hipDeviceProp_t p;
    SR_HIP(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) snprintf(name, name_len, "%s", p.name);
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", p.gcnArchName);
    if (total_bytes) *total_bytes = (int64_t)p.totalGlobalMem;
    if (compute_units) *compute_units = p.multiProcessorCount;
    return SIMRANK_OK;
}

int simrank_malloc(void** dptr, size_t bytes) {
    SR_REQUIRE(dptr, "dptr is NULL");
    *dptr = nullptr;
    if (bytes == 0) return SIMRANK_OK;
    return pool_alloc(dptr, bytes);
}

int simrank_free(void* dptr) {
    if (dptr) return pool_free(dptr);
    return SIMRANK_OK;
}

int simrank_pool_trim(int device) {
    pool_trim(device);
#ifndef SIMRANK_HOST_ONLY
    handback_release_slabs(device);
#endif
    return SIMRANK_OK;
}

int simrank_pool_stats(int device, int64_t* cached_bytes, int64_t* cached_blocks, int64_t* limit_bytes) {
    pool_stats(device, cached_bytes, cached_blocks, limit_bytes);
    return SIMRANK_OK;
}

int simrank_memset(void* dptr, int v, size_t bytes, void* stream) {
    if (bytes) SR_HIP(hipMemsetAsync(dptr, v, bytes, as_stream(stream)));
    return SIMRANK_OK;
}

static int copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, void* stream) {
    if (bytes == 0) return SIMRANK_OK;
    SR_REQUIRE(dst && src, "NULL pointer in copy of %zu bytes", bytes);
    SR_HIP(hipMemcpyAsync(dst, src, bytes, kind, as_stream(stream)));
    SR_HIP(hipStreamSynchronize(as_stream(stream)));
    return SIMRANK_OK;
}

int simrank_memcpy_h2d(void* d, const void* s, size_t b, void* st) {
    return copy_sync(d, s, b, hipMemcpyHostToDevice, st);
}
int simrank_memcpy_d2h(void* d, const void* s, size_t b, void* st) {
    return copy_sync(d, s, b, hipMemcpyDeviceToHost, st);
}
int simrank_memcpy_d2d(void* d, const void* s, size_t b, void* st) {
    return copy_sync(d, s, b, hipMemcpyDeviceToDevice, st);
}

int simrank_download_f64(double* dst, int64_t ld_dst, const float* src, int64_t ld_src,
                         int64_t n_rows, int64_t n_cols, void* stream) {
    SR_REQUIRE(n_rows >= 0 && n_cols >= 0 && ld_dst >= n_cols && ld_src >= n_cols, "bad shape");
    if (n_rows == 0 || n_cols == 0) return SIMRANK_OK;
    SR_REQUIRE(dst && src, "NULL pointer");
    // staged through pinned slabs so the PCIe copy of slab k+1 overlaps the f32->f64
    // widening of slab k on the host.  The two slabs (128 MB each at most, and their events) are
    // kept for the life of the process: allocating pinned memory costs more than a small download.
    struct PinCache {
        float* pin[2] = {nullptr, nullptr};
        hipEvent_t done[2];
        size_t cap = 0;
        bool have_events = false;
    };
    static std::mutex pin_mutex;
    static PinCache caches[16];                       // one per device (events are per device)
    std::lock_guard<std::mutex> lock(pin_mutex);
    int dev = 0;
    SR_HIP(hipGetDevice(&dev));
    PinCache& pc = caches[dev & 15];
    float** pin = pc.pin;
    hipEvent_t* done = pc.done;
    size_t& pin_cap = pc.cap;
    bool& have_events = pc.have_events;
    const int64_t row_bytes = n_cols * 4;
    const int64_t slab_rows = std::max<int64_t>(1, std::min<int64_t>(n_rows, (int64_t(128) << 20) / row_bytes));
    const size_t need = size_t(slab_rows) * row_bytes;
    const int64_t n_threads = std::max<unsigned>(1, std::min<unsigned>(32, std::thread::hardware_concurrency()));
    if (!have_events) {
        for (int i = 0; i < 2; ++i) SR_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        have_events = true;
    }
    if (need > pin_cap) {
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            pin[i] = nullptr;
        }
        pin_cap = 0;
        for (int i = 0; i < 2; ++i) SR_HIP(hipHostMalloc((void**)&pin[i], need, hipHostMallocPortable));
        pin_cap = need;
    }
    hipStream_t st = as_stream(stream);
    auto issue = [&](int64_t r0, int buf) -> hipError_t {
        int64_t nr = std::min(slab_rows, n_rows - r0);
        hipError_t e = hipMemcpy2DAsync(pin[buf], n_cols * 4, src + r0 * ld_src, ld_src * 4,
                                        n_cols * 4, nr, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        return hipEventRecord(done[buf], st);
    };
    int rc = SIMRANK_OK;
    hipError_t e = issue(0, 0);
    int buf = 0;
    for (int64_t r0 = 0; r0 < n_rows && e == hipSuccess; r0 += slab_rows, buf ^= 1) {
        if (r0 + slab_rows < n_rows) e = issue(r0 + slab_rows, buf ^ 1);
        if (e != hipSuccess) break;
        e = hipEventSynchronize(done[buf]);
        if (e != hipSuccess) break;
        int64_t nr = std::min(slab_rows, n_rows - r0);
        // widen on several host threads: one core moves ~1 GB/s, PCIe Gen5 delivers ~50
        const float* slab = pin[buf];
        auto widen = [&](int64_t ra, int64_t rb) {
            for (int64_t r = ra; r < rb; ++r) {
                const float* s = slab + r * n_cols;
                double* d = dst + (r0 + r) * ld_dst;
                for (int64_t c = 0; c < n_cols; ++c) d[c] = (double)s[c];
            }
        };
        const int64_t nt = std::max<int64_t>(1, std::min<int64_t>({n_threads, nr, (nr * n_cols) >> 16}));
        if (nt == 1) {
            widen(0, nr);
        } else {
            std::vector<std::thread> pool;
            for (int64_t t = 0; t < nt; ++t)
                pool.emplace_back(widen, nr * t / nt, nr * (t + 1) / nt);
            for (auto& th : pool) th.join();
        }
    }
    if (e != hipSuccess) {
        set_error("simrank_download_f64: %s", hipGetErrorString(e));
        rc = SIMRANK_ERR_HIP;
    }
    (void)hipStreamSynchronize(st);
    return rc;
}

int simrank_read_counters(const unsigned long long* counters, int32_t n, unsigned long long* sum,
                          void* stream) {
    SR_REQUIRE(counters && sum && n > 0 && n <= 65536, "bad counter read");
    // through a pinned slab kept per device: a pageable 8 KiB read-back costs ~50 us, and the
    // loop of SimRank.py:130 needs the count on the host after every update
    struct Slab { unsigned long long* host = nullptr; int32_t cap = 0; };
    static std::mutex m;
    static Slab slabs[16];
    std::lock_guard<std::mutex> lock(m);
    int dev = 0;
    SR_HIP(hipGetDevice(&dev));
    Slab& sl = slabs[dev & 15];
    if (n > sl.cap) {
        if (sl.host) (void)hipHostFree(sl.host);
        sl.host = nullptr;
        sl.cap = 0;
        SR_HIP(hipHostMalloc((void**)&sl.host, size_t(n) * sizeof(unsigned long long), hipHostMallocPortable));
        sl.cap = n;
    }
    hipStream_t st = as_stream(stream);
    SR_HIP(hipMemcpyAsync(sl.host, counters, size_t(n) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SR_HIP(hipStreamSynchronize(st));
    unsigned long long total = 0;
    for (int32_t i = 0; i < n; ++i) total += sl.host[i];
    *sum = total;
    return SIMRANK_OK;
}

// The count of an update WITHOUT stopping the stream (the loop of SimRank.py:129-140 as plan.hip runs it, for
// callers that drive the legs themselves): _fetch queues the copy of the counters into pinned slot `slot` of the
// caller's COUNTER SET and records an event behind it, _wait returns their sum once that copy has landed — whatever
// was queued behind the fetch (update k + 1) keeps running.  The set belongs to one engine: round 4 kept the slots in a
// per-device table, and two fits on one device read each other's counts (advisor, round 4).
}  // extern "C"
struct simrank_counter_set {
    struct Slot { unsigned long long* host = nullptr; int32_t cap = 0, n = 0; hipEvent_t ev = nullptr; } slot[4];
};
extern "C" {

int simrank_counters_create(simrank_counter_set** out) {
    SR_REQUIRE(out, "out is NULL");
    *out = new simrank_counter_set;
    return SIMRANK_OK;
}

int simrank_counters_destroy(simrank_counter_set* set) {
    if (!set) return SIMRANK_OK;
    for (auto& sl : set->slot) {
        if (sl.ev) { (void)hipEventSynchronize(sl.ev); (void)hipEventDestroy(sl.ev); }
        if (sl.host) (void)hipHostFree(sl.host);
    }
    delete set;
    return SIMRANK_OK;
}

int simrank_counters_fetch(simrank_counter_set* set, const unsigned long long* counters, int32_t n, int32_t slot, void* stream) {
    SR_REQUIRE(set && counters && n > 0 && n <= 65536 && slot >= 0 && slot < 4, "bad counter fetch");
    auto& sl = set->slot[slot];
    if (!sl.ev) SR_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (n > sl.cap) {
        if (sl.host) (void)hipHostFree(sl.host);
        sl.host = nullptr;
        sl.cap = 0;
        SR_HIP(hipHostMalloc((void**)&sl.host, size_t(n) * sizeof(unsigned long long), hipHostMallocPortable));
        sl.cap = n;
    }
    sl.n = n;
    SR_HIP(hipMemcpyAsync(sl.host, counters, size_t(n) * sizeof(unsigned long long), hipMemcpyDeviceToHost, as_stream(stream)));
    SR_HIP(hipEventRecord(sl.ev, as_stream(stream)));
    return SIMRANK_OK;
}

int simrank_counters_wait(simrank_counter_set* set, int32_t slot, unsigned long long* sum) {
    SR_REQUIRE(set && sum && slot >= 0 && slot < 4, "bad counter wait");
    const auto& sl = set->slot[slot];
    SR_REQUIRE(sl.ev && sl.host && sl.n > 0, "counter slot %d was never fetched into", slot);
    SR_HIP(hipEventSynchronize(sl.ev));
    unsigned long long total = 0;
    for (int32_t i = 0; i < sl.n; ++i) total += sl.host[i];
    *sum = total;
    return SIMRANK_OK;
}

int simrank_stream_create(void** stream) {
    SR_REQUIRE(stream, "stream is NULL");
    hipStream_t s;
    SR_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return SIMRANK_OK;
}
int simrank_stream_destroy(void* stream) {
    if (stream) SR_HIP(hipStreamDestroy(as_stream(stream)));
    return SIMRANK_OK;
}
int simrank_stream_synchronize(void* stream) {
    SR_HIP(hipStreamSynchronize(as_stream(stream)));
    return SIMRANK_OK;
}
int simrank_event_create(void** event) {
    SR_REQUIRE(event, "event is NULL");
    hipEvent_t e;
    SR_HIP(hipEventCreate(&e));
    *event = e;
    return SIMRANK_OK;
}
int simrank_event_destroy(void* event) {
    if (event) SR_HIP(hipEventDestroy((hipEvent_t)event));
    return SIMRANK_OK;
}
int simrank_event_record(void* event, void* stream) {
    SR_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    return SIMRANK_OK;
}
int simrank_event_synchronize(void* event) {
    SR_HIP(hipEventSynchronize((hipEvent_t)event));
    return SIMRANK_OK;
}
int simrank_event_elapsed_ms(void* start, void* stop, float* ms) {
    SR_REQUIRE(ms, "ms is NULL");
    SR_HIP(hipEventSynchronize((hipEvent_t)stop));
    SR_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SIMRANK_OK;
}

// ---------------------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------------------
int simrank_graph_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                         const int32_t* col, const float* rowscale, simrank_graph** out) {
    return graph_create_with(tuning_snapshot(), n_rows, n_cols, nnz, rowptr, col, rowscale, out);
}

}  // extern "C"

namespace simrank {
// (the plans of the set-up — dense sets, one-launch plan(s) — are independent of each other and of the uploads:
// they are built on threads of their own, simrank_graph_create of config 5 0.05 -> 0.03 s)
int graph_create_with(const Tuning& tun, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                      const int32_t* col, const float* rowscale, simrank_graph** out,
                      const std::function<int(simrank_graph*)>* after_base) {
    SR_REQUIRE(out, "out is NULL");
    *out = nullptr;
    SR_REQUIRE(n_rows > 0 && n_cols > 0 && nnz >= 0, "bad graph shape %lld x %lld, nnz %lld",
               (long long)n_rows, (long long)n_cols, (long long)nnz);
    SR_REQUIRE(n_rows < (int64_t(1) << 31) && n_cols < (int64_t(1) << 31) &&
                   nnz < (int64_t(1) << 31),
               "graph too large for 32-bit indices");
    SR_REQUIRE(rowptr && rowscale && (col || nnz == 0), "NULL CSR array");
    SR_REQUIRE(rowptr[0] == 0 && rowptr[n_rows] == nnz, "rowptr does not span [0, nnz]");
    int32_t max_row = 0;
    for (int64_t a = 0; a < n_rows; ++a) {
        SR_REQUIRE(rowptr[a + 1] >= rowptr[a], "rowptr not monotone at row %lld", (long long)a);
        max_row = std::max(max_row, rowptr[a + 1] - rowptr[a]);
        for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) {
            SR_REQUIRE(col[j] >= 0 && col[j] < n_cols, "column index %d out of range at row %lld",
                       col[j], (long long)a);
            SR_REQUIRE(j == rowptr[a] || col[j] > col[j - 1],
                       "columns of row %lld not strictly ascending", (long long)a);
        }
    }
    simrank_graph* g = new simrank_graph;
    g->tun = tun;
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    g->max_row_nnz = max_row;
    // plan builders on their own threads (each sets the device, keeps its own error text)
    int dev = 0;
#ifndef SIMRANK_HOST_ONLY
    (void)hipGetDevice(&dev);
#endif
    struct Job { int rc = SIMRANK_OK; std::string err; std::thread th; double ms = 0; };
    Job jobs[3];
    const bool timed = std::getenv("SIMRANK_TIME_BUILD") != nullptr;     // diagnostic: builder durations on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    auto spawn = [&](Job& j, int (*fn)(simrank_graph*, const int32_t*, const int32_t*, const float*)) {
        j.th = std::thread([&j, fn, g, rowptr, col, rowscale, dev, &since]() {
#ifndef SIMRANK_HOST_ONLY
            (void)hipSetDevice(dev);
#endif
            const double t0 = since();
            j.rc = fn(g, rowptr, col, rowscale);
            j.ms = since() - t0;
            if (j.rc) j.err = simrank_last_error();
        });
    };
    const bool big = nnz >= 20000;         // (threads cost more than they save on small graphs)
    auto dense_job = [](simrank_graph* gg, const int32_t* rp, const int32_t* cl, const float*) { return build_dense_plan(gg, rp, cl); };
    int n_jobs = 0;
    if (big) {
        if (g->tun.fuse) g->fused_build.store(1, std::memory_order_release);      // (before the dense builder can look)
        if (g->tun.dense_min > 0 && nnz > 0) spawn(jobs[n_jobs++], dense_job);
#ifdef SIMRANK_EXPERIMENT_FUSED2
        if (g->tun.fuse == 2 && nnz > 0) spawn(jobs[n_jobs++], build_fused2_plan);
#endif
        if (g->tun.fuse) {
            auto fused_job = [](simrank_graph* gg, const int32_t* rp, const int32_t* cl, const float* rs) {
                const int rc2 = build_fused_plan(gg, rp, cl, rs);
                gg->fused_build.store(2, std::memory_order_release);
                return rc2;
            };
            spawn(jobs[n_jobs++], fused_job);
        }
    }
    // (the plan builders above read the validated CSR only: the transposed pattern, the hub list and the tiles are worked out
    // here, beside them)
    std::vector<int32_t> t_rowptr(size_t(n_cols) + 1, 0);
    // (the transposed pattern serves the evidence counts, which ignore rows without weight: `G > 0` in SimRank.py:315 — it
    // lists live rows only)
    for (int64_t a = 0; a < n_rows; ++a)
        if (rowscale[a] > 0.f)
            for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) t_rowptr[size_t(col[j]) + 1]++;
    for (int64_t i = 0; i < n_cols; ++i) t_rowptr[i + 1] += t_rowptr[i];
    std::vector<int32_t> t_col(std::max<size_t>(1, size_t(nnz)));
    // t_pos[j]: where entry j = (a, i) of the CSR sits in column i's list of the transposed pattern — the list is
    // ascending in a, so its part from there on holds exactly the rows b >= a (evidence counts, upper triangle)
    std::vector<int32_t> t_pos(std::max<size_t>(1, size_t(nnz)), 0);
    {
        std::vector<int32_t> cur(t_rowptr.begin(), t_rowptr.end() - 1);
        for (int64_t a = 0; a < n_rows; ++a)
            if (rowscale[a] > 0.f)
                for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) {
                    t_pos[size_t(j)] = cur[col[j]];
                    t_col[cur[col[j]]++] = (int32_t)a;
                }
    }
    // HUB columns of the evidence counts (round 5; spmm.hip evidence_hub_kernel): a column with d live rows costs the
    // LDS-counter kernel d^2 / 2 paths (~1e11 paths/s) and one more column of the 0/1 image costs the matrix-core
    // kernel n_rows^2 multiply-adds (~1e15/s): from d >= 0.014 n_rows on (tuning "ev_hub", in 1/1000 of n_rows; 0 = off)
    // the column's pairs are counted as an i8 product P_H . P_H^T instead.  On a power-law pattern a few dozen columns
    // carry nearly all of the paths; on a MovieLens-like one nearly every column qualifies.
    std::vector<int32_t> hubidx(size_t(n_cols), -1);
    if (g->tun.ev_hub > 0 && nnz > 0) {
        const int64_t thr = std::max<int64_t>(48, (n_rows * g->tun.ev_hub + 999) / 1000);
        std::vector<int32_t> cand;
        for (int64_t i = 0; i < n_cols; ++i)
            if (t_rowptr[size_t(i) + 1] - t_rowptr[size_t(i)] >= thr) cand.push_back((int32_t)i);
        std::sort(cand.begin(), cand.end(), [&](int32_t x, int32_t y) {
            const int32_t dx = t_rowptr[size_t(x) + 1] - t_rowptr[size_t(x)], dy = t_rowptr[size_t(y) + 1] - t_rowptr[size_t(y)];
            return dx != dy ? dx > dy : x < y;
        });
        if (cand.size() > 4096) cand.resize(4096);
        if (cand.size() >= 8) {
            for (size_t h = 0; h < cand.size(); ++h) hubidx[size_t(cand[h])] = (int32_t)h;
            g->ev_hubs = (int32_t)((cand.size() + 31) / 32 * 32);
        }
    }
    std::vector<int32_t> tile_row0, sym_map;
    g->n_tiles = (int32_t)build_tiles(rowptr, n_rows, nnz, g->tun.balance, tile_row0, sym_map, g->tun.sym_desc != 0);
    g->sym_blocks = (int32_t)(sym_map.size() / 2);
    auto up = [&](void** d, const void* h, size_t bytes) -> int {
        size_t alloc = std::max<size_t>(bytes, 16);
        hipError_t e = plan_alloc(d, alloc);
        if (e == hipSuccess && bytes) e = plan_upload(*d, h, bytes);
        if (e != hipSuccess) {
            set_error("graph upload: %s", hipGetErrorString(e));
            (void)hipGetLastError();
            return SIMRANK_ERR_HIP;
        }
        return SIMRANK_OK;
    };
    int rc = up((void**)&g->rowptr, rowptr, size_t(n_rows + 1) * 4);
    if (!rc) rc = up((void**)&g->col, col, size_t(nnz) * 4);
    if (!rc && n_cols <= 65536 && nnz > 0) {
        std::vector<uint16_t> c16(col, col + nnz);
        rc = up((void**)&g->col16, c16.data(), size_t(nnz) * 2);
    }
    if (!rc) rc = up((void**)&g->rowscale, rowscale, size_t(n_rows) * 4);
    if (!rc) rc = up((void**)&g->t_rowptr, t_rowptr.data(), size_t(n_cols + 1) * 4);
    if (!rc) rc = up((void**)&g->t_col, t_col.data(), size_t(nnz) * 4);
    if (!rc) rc = up((void**)&g->t_pos, t_pos.data(), size_t(nnz) * 4);
    if (!rc && g->ev_hubs) rc = up((void**)&g->ev_hubidx, hubidx.data(), size_t(n_cols) * 4);
    if (!rc && g->n_tiles) rc = up((void**)&g->tile_row0, tile_row0.data(), tile_row0.size() * 4);
    if (!rc && g->sym_blocks) rc = up((void**)&g->sym_map, sym_map.data(), sym_map.size() * 4);
    if (!rc && after_base && *after_base) rc = (*after_base)(g);     // (the builders are still at work on their threads)
    const double t_base = since();
    for (int i = 0; i < n_jobs; ++i) {
        jobs[i].th.join();
        if (timed) std::fprintf(stderr, "simrank_graph_create: builder %d took %.1f ms\n", i, jobs[i].ms);
        if (!rc && jobs[i].rc) {
            rc = jobs[i].rc;
            set_error("%s", jobs[i].err.c_str());
        }
    }
    if (!big) {
        if (!rc && g->tun.fuse) rc = build_fused_plan(g, rowptr, col, rowscale);    // (also without entries: fp16-held updates have no other path)
        if (g->tun.fuse) g->fused_build.store(2, std::memory_order_release);
        if (!rc && g->tun.dense_min > 0 && nnz > 0) rc = build_dense_plan(g, rowptr, col);
#ifdef SIMRANK_EXPERIMENT_FUSED2
        if (!rc && g->tun.fuse == 2 && g->nnz > 0) rc = build_fused2_plan(g, rowptr, col, rowscale);
#endif
    }
    if (timed)
        std::fprintf(stderr, "simrank_graph_create: n %lld nnz %lld: transposed pattern + tiles + base uploads %.1f ms, all %.1f ms\n",
                     (long long)n_rows, (long long)nnz, t_base, since());
    if (rc) {
        simrank_graph_destroy(g);
        return rc;
    }
    *out = g;
    return SIMRANK_OK;
}
}  // namespace simrank

extern "C" {

int simrank_graph_destroy(simrank_graph* g) {
    if (!g) return SIMRANK_OK;
    plan_free(g->rowptr);
    plan_free(g->col);
    plan_free(g->col16);
    plan_free(g->rowscale);
    plan_free(g->t_rowptr);
    plan_free(g->t_col);
    plan_free(g->t_pos);
    plan_free(g->ev_hubidx);
    plan_free(g->ev_hub_image);
#ifndef SIMRANK_HOST_ONLY
    if (g->ev_hub_ready) (void)hipEventDestroy((hipEvent_t)g->ev_hub_ready);
#endif
    plan_free(g->tile_row0);
    plan_free(g->sym_map);
    free_dense_plan(g->dense);
    free_fused_plan(g->fused);
#ifdef SIMRANK_EXPERIMENT_FUSED2
    free_fused2_plan(g->fused2);
#endif
    delete g;
    return SIMRANK_OK;
}

int simrank_graph_set_dense_terms(simrank_graph* g, int32_t terms) {
    SR_REQUIRE(g, "graph is NULL");
    SR_REQUIRE(terms == 1 || terms == 3, "dense terms must be 3 (exact) or 1 (fp16 operand)");
    g->tun.dense_terms = terms;
    return SIMRANK_OK;
}

int simrank_graph_shape(const simrank_graph* g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz) {
    SR_REQUIRE(g, "graph is NULL");
    if (n_rows) *n_rows = g->n_rows;
    if (n_cols) *n_cols = g->n_cols;
    if (nnz) *nnz = g->nnz;
    return SIMRANK_OK;
}

// ---------------------------------------------------------------------------------------
// tuning
// ---------------------------------------------------------------------------------------
int simrank_set_tuning(const char* key, int64_t value) {
    SR_REQUIRE(key, "key is NULL");
    std::lock_guard<std::mutex> lock(tuning_mutex());
    Tuning& t = tuning();
    if (!strcmp(key, "panel")) {
        SR_REQUIRE(value == 0 || value == 16 || value == 32 || value == 64 || value == 128 ||
                       value == 256,
                   "panel must be 0, 16, 32, 64, 128 or 256");
        t.panel = value;
    } else if (!strcmp(key, "xcd_map")) {
        t.xcd_map = value ? 1 : 0;
    } else if (!strcmp(key, "tile")) {
        SR_REQUIRE(value == 0 || value == 16 || value == 32 || value == 64, "tile must be 0, 16, 32 or 64");
        t.tile = value;
    } else if (!strcmp(key, "huge")) {
        SR_REQUIRE(value >= 64 && value <= (1 << 30), "huge must be >= 64");
        t.huge = value;
    } else if (!strcmp(key, "triangle")) {
        t.triangle = value ? 1 : 0;
    } else if (!strcmp(key, "stream_nt")) {
        t.stream_nt = value ? 1 : 0;
    } else if (!strcmp(key, "balance")) {
        SR_REQUIRE(value >= 0 && value <= 1024, "balance must be 0 (uniform tiles) .. 1024");
        t.balance = value;
    } else if (!strcmp(key, "dense_min")) {
        SR_REQUIRE(value >= 0 && value <= 128, "dense_min must be 0 (off) .. 128");
        t.dense_min = value;
    } else if (!strcmp(key, "probe_mask")) {
        SR_REQUIRE(value < 0 || getenv("SIMRANK_ENABLE_PROBES"),
                   "probe knobs give WRONG results (measurement only): set SIMRANK_ENABLE_PROBES=1 to use them");
        t.probe_mask = value < 0 ? -1 : value;
    } else if (!strcmp(key, "dense_terms")) {
        SR_REQUIRE(value == 1 || value == 3, "dense_terms must be 3 (exact) or 1 (fp16 operand)");
        t.dense_terms = value;
    } else if (!strcmp(key, "probe_flags")) {
        SR_REQUIRE(value == 0 || getenv("SIMRANK_ENABLE_PROBES"),
                   "probe knobs give WRONG results (measurement only): set SIMRANK_ENABLE_PROBES=1 to use them");
        t.probe_flags = value & 63;
    } else if (!strcmp(key, "addr32")) {
        t.addr32 = value ? 1 : 0;
    } else if (!strcmp(key, "lean")) {
        SR_REQUIRE(value == 0 || value == 1, "lean must be 0 or 1");
        t.lean = value;
    } else if (!strcmp(key, "ids16")) {
        t.ids16 = value ? 1 : 0;
    } else if (!strcmp(key, "sym_desc")) {
        t.sym_desc = value ? 1 : 0;
    } else if (!strcmp(key, "dense_sym")) {
        t.dense_sym = value < 0 ? -1 : (value ? 1 : 0);
    } else if (!strcmp(key, "fuse")) {
#ifdef SIMRANK_EXPERIMENT_FUSED2
        SR_REQUIRE(value >= 0 && value <= 2, "fuse must be 0 (two launches), 1 (one launch) or 2 (one persistent launch)");
#else
        SR_REQUIRE(value == 0 || value == 1, "fuse must be 0 (two launches) or 1 (one launch)");
#endif
        t.fuse = value;
    } else if (!strcmp(key, "ev_hub")) {
        SR_REQUIRE(value >= 0 && value <= 1000, "ev_hub must be 0 (off) .. 1000 (thousandths of the row count)");
        t.ev_hub = value;
    } else if (!strcmp(key, "fuse_min")) {
        SR_REQUIRE(value == 0 || (value >= 2 && value <= 128), "fuse_min must be 2 .. 128, or 0 (quads that pay)");
        t.fuse_min = value;
    } else if (!strcmp(key, "fuse_pays")) {
        SR_REQUIRE(value >= -1 && value <= 8192, "fuse_pays must be -1 (automatic) .. 8192");
        t.fuse_pays = value;
    } else if (!strcmp(key, "fuse_unit")) {
        SR_REQUIRE(value >= 4 && value <= (1 << 20), "fuse_unit must be >= 4");
        t.fuse_unit = value;
    } else if (!strcmp(key, "fuse_store")) {
        SR_REQUIRE(value >= 0 && value <= 3, "fuse_store must be 0 .. 3");
        t.fuse_store = value;
    } else if (!strcmp(key, "fuse_meta_nt")) {
        t.fuse_meta_nt = value ? 1 : 0;
    } else if (!strcmp(key, "fuse_order")) {
        SR_REQUIRE(value >= 0 && value <= 64, "fuse_order must be 0 .. 64");
        t.fuse_order = value;
    } else if (!strcmp(key, "fuse_max_rows")) {
        SR_REQUIRE(value >= 0, "fuse_max_rows must be >= 0");
        t.fuse_max_rows = value;
    } else if (!strcmp(key, "fuse_rows")) {
        SR_REQUIRE(value >= 64 && value <= (int64_t(1) << 40), "fuse_rows must be >= 64");
        t.fuse_rows = value;
    } else if (!strcmp(key, "ev_tri")) {
        t.ev_tri = value ? 1 : 0;
    } else if (!strcmp(key, "fuse_shards")) {
        t.fuse_shards = value ? 1 : 0;
    } else if (!strcmp(key, "fuse_cap")) {
        SR_REQUIRE(value >= 1000 && value <= (int64_t(1) << 40), "fuse_cap must be >= 1000");
        t.fuse_cap = value;
    } else if (!strcmp(key, "fuse_wgs")) {
        SR_REQUIRE(value >= 1 && value <= 8, "fuse_wgs must be 1 .. 8");
        t.fuse_wgs = value;
    } else if (!strcmp(key, "fuse_group")) {
        SR_REQUIRE(value >= 1 && value <= 4, "fuse_group must be 1 .. 4");
        t.fuse_group = value;
    } else if (!strcmp(key, "fuse_sym")) {
        SR_REQUIRE(value >= -1 && value <= 1, "fuse_sym must be -1 (automatic), 0 or 1");
        t.fuse_sym = value;
    } else if (!strcmp(key, "fuse_steps")) {
        SR_REQUIRE(value >= -1 && value <= (1 << 20), "fuse_steps must be >= 0, or -1 (by the operand's size)");
        t.fuse_steps = value;
    } else if (!strcmp(key, "fuse_dens")) {
        SR_REQUIRE(value >= 0 && value <= (1 << 20), "fuse_dens must be >= 0");
        t.fuse_dens = value;
    } else if (!strcmp(key, "dense_cols")) {
        SR_REQUIRE(value >= 1 && value <= (1 << 20), "dense_cols must be >= 1");
        t.dense_cols = value;
    } else {
        SR_REQUIRE(false, "unknown tuning key '%s'", key);
    }
    return SIMRANK_OK;
}

int simrank_get_tuning(const char* key, int64_t* value) {
    SR_REQUIRE(key && value, "NULL argument");
    const Tuning t = tuning_snapshot();
    if (!strcmp(key, "panel")) *value = t.panel;
    else if (!strcmp(key, "xcd_map")) *value = t.xcd_map;
    else if (!strcmp(key, "stream_nt")) *value = t.stream_nt;
    else if (!strcmp(key, "tile")) *value = t.tile;
    else if (!strcmp(key, "triangle")) *value = t.triangle;
    else if (!strcmp(key, "huge")) *value = t.huge;
    else if (!strcmp(key, "balance")) *value = t.balance;
    else if (!strcmp(key, "dense_min")) *value = t.dense_min;
    else if (!strcmp(key, "dense_cols")) *value = t.dense_cols;
    else if (!strcmp(key, "dense_sym")) *value = t.dense_sym;
    else if (!strcmp(key, "fuse")) *value = t.fuse;
    else if (!strcmp(key, "ev_hub")) *value = t.ev_hub;
    else if (!strcmp(key, "fuse_min")) *value = t.fuse_min;
    else if (!strcmp(key, "fuse_pays")) *value = t.fuse_pays;
    else if (!strcmp(key, "fuse_steps")) *value = t.fuse_steps;
    else if (!strcmp(key, "fuse_dens")) *value = t.fuse_dens;
    else if (!strcmp(key, "fuse_unit")) *value = t.fuse_unit;
    else if (!strcmp(key, "fuse_group")) *value = t.fuse_group;
    else if (!strcmp(key, "fuse_sym")) *value = t.fuse_sym;
    else if (!strcmp(key, "fuse_cap")) *value = t.fuse_cap;
    else if (!strcmp(key, "fuse_shards")) *value = t.fuse_shards;
    else if (!strcmp(key, "ev_tri")) *value = t.ev_tri;
    else if (!strcmp(key, "fuse_rows")) *value = t.fuse_rows;
    else if (!strcmp(key, "fuse_wgs")) *value = t.fuse_wgs;
    else if (!strcmp(key, "fuse_max_rows")) *value = t.fuse_max_rows;
    else if (!strcmp(key, "fuse_order")) *value = t.fuse_order;
    else if (!strcmp(key, "fuse_store")) *value = t.fuse_store;
    else if (!strcmp(key, "fuse_meta_nt")) *value = t.fuse_meta_nt;
    else if (!strcmp(key, "sym_desc")) *value = t.sym_desc;
    else if (!strcmp(key, "ids16")) *value = t.ids16;
    else if (!strcmp(key, "lean")) *value = t.lean;
    else if (!strcmp(key, "addr32")) *value = t.addr32;
    else if (!strcmp(key, "probe_flags")) *value = t.probe_flags;
    else if (!strcmp(key, "dense_terms")) *value = t.dense_terms;
    else if (!strcmp(key, "probe_mask")) *value = t.probe_mask;
    else SR_REQUIRE(false, "unknown tuning key '%s'", key);
    return SIMRANK_OK;
}

}  // extern "C"
