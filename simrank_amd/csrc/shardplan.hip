// K10 behind the C ABI ("step(...) includes K10 when P>1"): the loop of SimRank.fit with S split by
// COLUMN BLOCK over `world` GPUs, one process per GPU, for callers that bind the library directly.  The reference
// (SimRank.py:129-140, :351-362, :443-454) is one NumPy process and has no counterpart; what one update computes is
// its two `.dot`s, `C *`, `Evidence *`, the prior blend, `fill_diagonal` and the test of `_converged` (:74).
//
// One update of a rank (driver.Side / driver.Solver._update do the same over torch.distributed):
//   leg 1       simrank_spmm(transpose_out, t_block = mb, t_pad): (W . S_block)^T straight into the chunks of the
//               all-to-all, one launch per STAGE (a slice of the rank's columns)
//   exchange 1  per stage: chunk h of the slice goes to rank h; what arrives from rank h lands at the rows of the
//               leg-2 operand that rank's columns own — consecutive rows, so no renaming of the graph is needed
//               (all_to_all_single could not scatter like that; ncclSend / ncclRecv groups can).  RCCL runs on a
//               stream of its own: it waits for the stage's kernel by an event and moves the slice while the next
//               stage computes; leg 2 waits for the last stage.
//   leg 2       full form: simrank_spmm with the fused epilogue; half form: simrank_spmm_shard (tiles i <= j), then
//   exchange 2  the packed mirrored tiles, equal chunks, and simrank_shard_unpack puts them in place
//   count       the striped counters summed over the ranks on the device (ncclAllReduce), copied to pinned memory;
//               simrank_shardplan_run queues update k + 1 before it reads the count of update k, like the single plan
//
// A communicator is RCCL (dlopen, so the library does not depend on it), or an in-process group of virtual ranks on
// one device whose exchanges are device copies queued by ONE host thread (simrank_comm_local_group: the ranks advance
// in turn), or — round 6 — a THREAD group (simrank_comm_thread_group): one host thread per rank, each running the very
// code path of an RCCL rank (one plan, a stream of its own for the exchanges, stage events, hops, the grouped sends and
// receives, the all-reduced count) over a transport with RCCL's interface whose sends and receives rendezvous between
// the threads and move the bytes with peer copies behind the sender's event.  That is how the loop an 8-GPU run executes
// meets ranks that run AT THE SAME TIME on the one GPU available (tests/test_gpu_shardplan.py).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

#include "common.h"

static_assert(sizeof(ncclUniqueId) == SIMRANK_COMM_ID_BYTES, "RCCL's unique id is SIMRANK_COMM_ID_BYTES long");

namespace simrank {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* name = std::getenv("SIMRANK_RCCL_LIB");
        for (const char* cand : {name, "librccl.so.1", "librccl.so"}) {
            if (!cand || !*cand) continue;
            r.lib = dlopen(cand, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        auto sym = [](const char* s) { return dlsym(r.lib, s); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv &&
               r.AllReduce && r.GetErrorString;
    });
    return &r;
}

static const char* comm_error_string(ncclResult_t r) {
    if (r == ncclSystemError) return "a peer rank did not arrive in time or failed (in-process transport), or RCCL's system error";
    Rccl* R = rccl();
    return R->ok ? R->GetErrorString(r) : "transport error";
}

#define SR_RCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            ::simrank::set_error("%s failed: %s (%s:%d)", #call, comm_error_string(r_), __FILE__, __LINE__); \
            return SIMRANK_ERR_HIP;                                                            \
        }                                                                                      \
    } while (0)

// ---- the thread group's transport: RCCL's interface (the entries of `Rccl` the loops call) between the host threads of one
// process.  A send is POSTED (source pointer, byte count, an event recorded on the sender's stream at that point); the matching
// receive — the next one from that peer, in order, as RCCL matches them — makes ITS stream wait for that event, copies, and
// records a second event; the sender's stream then waits for that one (the buffer is the sender's again when the call has
// completed in stream order, RCCL's rule).  ncclGroupEnd posts every send of the group before it waits for anything, so
// ranks that issue the same groups in the same order cannot deadlock; a peer that never arrives turns into an error after
// SIMRANK_THREAD_COMM_TIMEOUT seconds (default 120) on every rank of the group instead of a hang.
struct TMsg {
    const void* src;
    size_t bytes;
    hipEvent_t ready, done;
    bool finished = false;
};
struct TGroup {
    int32_t world = 0, alive = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<TMsg*>> box;      // [src * world + dst]: posted, not yet received
    bool failed = false;
    double timeout_s = 120.0;
};
struct TComm {
    TGroup* g = nullptr;
    int32_t rank = 0;
    struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t st; };
    std::vector<Op> ops;                     // of the open group
    std::deque<hipEvent_t> used;             // events handed out, oldest first (recycled once they have completed)
    unsigned long long* tmp = nullptr;       // all-reduce: what the peers sent, [world][tmp_count]
    size_t tmp_count = 0;
};
static thread_local int t_depth = 0;
static thread_local std::vector<TComm*> t_open;

static size_t dt_bytes(ncclDataType_t dt) {
    switch (dt) {
        case ncclHalf: return 2;
        case ncclUint64: case ncclInt64: case ncclFloat64: return 8;
        case ncclInt8: case ncclUint8: return 1;
        default: return 4;
    }
}
static hipEvent_t t_event(TComm* c) {
    if (c->used.size() > 64 && hipEventQuery(c->used.front()) == hipSuccess) {
        hipEvent_t e = c->used.front();
        c->used.pop_front();
        c->used.push_back(e);
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    c->used.push_back(e);
    return e;
}
static ncclResult_t t_fail(TGroup* g) {
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->failed = true;
    }
    g->cv.notify_all();
    return ncclSystemError;
}
static ncclResult_t t_flush(TComm* c) {
    TGroup* g = c->g;
    const int32_t P = g->world;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(g->timeout_s);
    std::vector<std::pair<TMsg*, hipStream_t>> mine;
    std::vector<TComm::Op> ops;
    ops.swap(c->ops);
    // 1. every send of the group is posted before anything waits
    for (const TComm::Op& op : ops) {
        if (!op.send) continue;
        TMsg* m = new TMsg{op.buf, op.bytes, t_event(c), t_event(c)};
        if (!m->ready || !m->done || hipEventRecord(m->ready, op.st) != hipSuccess) { delete m; return t_fail(g); }
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->box[size_t(c->rank) * P + op.peer].push_back(m);
        }
        mine.push_back({m, op.st});
    }
    g->cv.notify_all();
    // 2. the receives, in order: the next message that peer posted for this rank
    for (const TComm::Op& op : ops) {
        if (op.send) continue;
        TMsg* m = nullptr;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            std::deque<TMsg*>& q = g->box[size_t(op.peer) * P + c->rank];
            if (!g->cv.wait_until(lk, deadline, [&] { return g->failed || !q.empty(); }) || g->failed) {
                g->failed = true;
                lk.unlock();
                g->cv.notify_all();
                return ncclSystemError;
            }
            m = q.front();
            q.pop_front();
        }
        if (m->bytes != op.bytes || hipStreamWaitEvent(op.st, m->ready, 0) != hipSuccess ||
            hipMemcpyAsync(op.buf, m->src, op.bytes, hipMemcpyDeviceToDevice, op.st) != hipSuccess ||
            hipEventRecord(m->done, op.st) != hipSuccess)
            return t_fail(g);
        {
            std::lock_guard<std::mutex> lk(g->mu);
            m->finished = true;
        }
        g->cv.notify_all();
    }
    // 3. a send has completed (in stream order) when its receiver has copied
    for (auto& ms : mine) {
        TMsg* m = ms.first;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            if (!g->cv.wait_until(lk, deadline, [&] { return g->failed || m->finished; }) || g->failed) {
                g->failed = true;
                lk.unlock();
                g->cv.notify_all();
                return ncclSystemError;          // (the message stays with the group: its receiver may still hold it)
            }
        }
        if (hipStreamWaitEvent(ms.second, m->done, 0) != hipSuccess) return t_fail(g);
        delete m;
    }
    return ncclSuccess;
}
static ncclResult_t t_group_start() {
    ++t_depth;
    return ncclSuccess;
}
static ncclResult_t t_group_end() {
    if (t_depth > 0 && --t_depth > 0) return ncclSuccess;
    std::vector<TComm*> open;
    open.swap(t_open);
    ncclResult_t r = ncclSuccess;
    for (TComm* c : open) {
        const ncclResult_t rc = t_flush(c);
        if (rc != ncclSuccess) r = rc;
    }
    return r;
}
static ncclResult_t t_queue(TComm* c, bool send, void* buf, size_t bytes, int peer, hipStream_t st) {
    if (peer < 0 || peer >= c->g->world || peer == c->rank) return ncclInvalidArgument;
    if (c->ops.empty() && std::find(t_open.begin(), t_open.end(), c) == t_open.end()) t_open.push_back(c);
    c->ops.push_back({send, buf, bytes, peer, st});
    if (t_depth == 0) return t_group_end();              // an ungrouped call is a group of one
    return ncclSuccess;
}
static ncclResult_t t_send(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    return t_queue(reinterpret_cast<TComm*>(comm), true, const_cast<void*>(buf), count * dt_bytes(dt), peer, st);
}
static ncclResult_t t_recv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    return t_queue(reinterpret_cast<TComm*>(comm), false, buf, count * dt_bytes(dt), peer, st);
}
#ifndef SIMRANK_HOST_ONLY
__global__ void t_sum_u64_kernel(unsigned long long* out, const unsigned long long* mine, const unsigned long long* others,
                                 int world, int me, int count, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned long long t = mine[i];
    for (int h = 0; h < world; ++h)
        if (h != me) t += others[size_t(h) * stride + i];
    out[i] = t;
}
#endif
// the sum over the ranks of `count` 64-bit counters, in place or not: everybody sends to everybody, then adds in rank order
static ncclResult_t t_all_reduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                                 hipStream_t st) {
    TComm* c = reinterpret_cast<TComm*>(comm);
    if (dt != ncclUint64 || op != ncclSum || t_depth != 0) return ncclInvalidArgument;
    const int32_t P = c->g->world;
    if (c->tmp_count < count) {
        if (c->tmp) (void)hipFree(c->tmp);
        c->tmp = nullptr;
        if (hipMalloc((void**)&c->tmp, size_t(P) * count * 8) != hipSuccess) return ncclSystemError;
        c->tmp_count = count;
    }
    t_group_start();
    for (int32_t h = 0; h < P; ++h) {
        if (h == c->rank) continue;
        t_send(send, count, dt, h, comm, st);
        t_recv(c->tmp + size_t(h) * c->tmp_count, count, dt, h, comm, st);
    }
    const ncclResult_t r = t_group_end();
    if (r != ncclSuccess) return r;
#ifndef SIMRANK_HOST_ONLY
    hipLaunchKernelGGL(t_sum_u64_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, (unsigned long long*)recv,
                       (const unsigned long long*)send, c->tmp, P, c->rank, (int)count, (int)c->tmp_count);
    if (hipGetLastError() != hipSuccess) return ncclSystemError;
#endif
    return ncclSuccess;
}
static const char* t_error_string(ncclResult_t r) { return comm_error_string(r); }
static Rccl* thread_api() {
    static Rccl t;
    static std::once_flag once;
    std::call_once(once, [] {
        t.GroupStart = t_group_start;
        t.GroupEnd = t_group_end;
        t.Send = t_send;
        t.Recv = t_recv;
        t.AllReduce = t_all_reduce;
        t.GetErrorString = t_error_string;
        t.ok = true;
    });
    return &t;
}

struct LocalGroup {
    int32_t world = 0;
    int32_t alive = 0;
};

struct PlanPrepView {                      // one group's share of a PlanPrep / BiPlanPrep
    const std::vector<int32_t>* ord;
    const std::vector<int32_t>* inv;
    const std::vector<int32_t>* rp;
    const std::vector<int32_t>* cl;
    const std::vector<float>* rs;
};

constexpr float kHalfScale = 16384.0f;       // what fp16-held matrices are scaled by (include/simrank_hip.h, SCALE)
constexpr float kWireScale = 16384.0f;       // what values on an fp16 wire are multiplied by (engine.HipOps.WIRE_SCALE)

static int64_t env_pad(const char* name) {
    const char* v = std::getenv(name);
    const int64_t p = v && *v ? std::atoll(v) : 96;      // three 128-byte lines ("The row pitch")
    return p < 0 ? 0 : p / 4 * 4;
}
// leading dimension of a row-major block of `cols` elements of `elem` bytes (engine.HipOps.pitch)
static int64_t pitch(int64_t cols, int64_t elem) {
    const int64_t unit = 16 / elem;
    int64_t ld = (std::max<int64_t>(cols, 1) + unit - 1) / unit * unit;
    if (ld >= 4096 && (ld & (ld - 1)) == 0) ld += env_pad("SIMRANK_PITCH_PAD");
    return ld;
}
static int64_t row_pad(int64_t block_rows) {             // driver.row_pad
    return block_rows >= 1024 && block_rows % 256 == 0 ? env_pad("SIMRANK_ROW_PAD") / 32 * 32 : 0;
}
static void part(int64_t n, int32_t world, int32_t rank, int64_t* lo, int64_t* hi) {      // ingest.partition
    const int64_t b = (n + world - 1) / world;
    *lo = std::min<int64_t>(n, rank * b);
    *hi = std::min<int64_t>(n, *lo + b);
}
static int64_t span(int64_t n, int32_t world, int32_t rank) {
    int64_t lo, hi;
    part(n, world, rank, &lo, &hi);
    return hi - lo;
}
// driver.stage_widths: equal pieces rounded up to whole 32-column panels, the last one shorter or empty
static int64_t stage_width(int64_t n_cols, int32_t n_stages, int32_t s, int32_t align = 32) {
    int64_t q = (n_cols + n_stages - 1) / n_stages;
    q = (q + align - 1) / align * align;
    return std::max<int64_t>(0, std::min<int64_t>(q, n_cols - s * q));
}
static int64_t stage_col0(int64_t n_cols, int32_t n_stages, int32_t s, int32_t align = 32) {
    int64_t c = 0;
    for (int32_t t = 0; t < s; ++t) c += stage_width(n_cols, n_stages, t, align);
    return c;
}

}  // namespace simrank

struct simrank_comm {
    int32_t rank = 0, world = 1;
    simrank::LocalGroup* group = nullptr;    // in-process group whose ranks ONE thread advances in turn, or
    ncclComm_t nccl = nullptr;               // RCCL (or the thread group's handle of this rank, a TComm)
    simrank::Rccl* api = nullptr;            // ... and the calls that go with `nccl`: RCCL's own, or the thread transport's
    simrank::TComm* thread = nullptr;        // set for a rank of a thread group (owned)
    bool owned = false;
};

struct simrank_shardplan {
    simrank_comm* comm = nullptr;
    int32_t rank = 0, world = 1;
    int64_t n = 0, mb = 0, m_lo = 0, m_hi = 0, Lm = 0;
    // the OPERAND of leg 1: this rank's column block of S itself (the directed classes: k = n, Lk = Lm, src = NULL), or of the
    // OTHER group's matrix (the two-matrix classes, simrank_shardbiplan_*: k = that group's size, Lk = this rank's block of
    // it, src = that group's plan: S1 <- W12 . S2 . W12^T reads S2's blocks, SimRank.py:297-302)
    int64_t k = 0, Lk = 0;
    simrank_shardplan* src = nullptr;
    int64_t ld = 0, pad = 0, send_ld = 0, recv_ld = 0, ld_ev = 0;
    simrank_graph* g = nullptr;
    float* S[2] = {nullptr, nullptr};       // the rank's column block of the iterate, row-major n x Lm, ping-pong
    float* send = nullptr;                  // exchange 1: Lm columns x (n + world pad) floats, chunked per stage
    float* recv = nullptr;                  //             the leg-2 operand, n rows x (Lm + pad)
    // asymmetric iterates (a prior that is not symmetric, SimRank.py:453 / :488 / :491): leg 2 is leg 1's launch on `recv` — its
    // product leaves transposed through `send2`, a second all-to-all lands it in `recv2`, the epilogue runs as a pass of its own
    int32_t asym = 0;
    float* send2 = nullptr;                 // Lm columns x (n + world pad) floats, chunked per stage
    float* recv2 = nullptr;                 // n rows x (Lm + pad): W S W^T's columns of this rank, raw
    float* sh_send = nullptr;               // exchange 2 (half form): world chunks of packed mirrored tiles
    float* sh_recv = nullptr;
    int64_t sh_chunk = 0;
    // the half-form leg 2 (and exchange 2) in stages of column tiles, heaviest (last tiles) first: column tile j packs j mirrored
    // tiles per source shard into slots j (j - 1) / 2 ..., so tiles [lo, hi) own a contiguous slot range of every chunk; the
    // buffers hold those ranges stage-major (per stage one piece per rank) and a stage's tiles leave while the next computes
    // (driver.Side.sh_stages).  Empty: the leg is one launch, its tiles leave after it.
    struct Stage2 { int32_t tile_lo, tile_hi; int64_t off, chunk; };
    std::vector<Stage2> sh_stages;
    std::vector<hipEvent_t> staged2;        // RCCL worlds: "this stage's kernel is done"
    uint16_t* wire[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // fp16 shadows of send / recv / sh_send / sh_recv / send2 / recv2
    uint8_t* ev = nullptr;
    float* prior = nullptr;
    int32_t* inv = nullptr;                 // device: position of caller's node i in the solver's order
    std::vector<int32_t> ord;               // host: ord[position] = caller's node
    unsigned long long* counters = nullptr;
    unsigned long long* host_counters[2] = {nullptr, nullptr};
    hipEvent_t counted[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipStream_t xstream = nullptr;          // RCCL's stream for exchange 1 (stages overlap the kernels)
    std::vector<hipEvent_t> staged;         // [n_stages] "this stage's kernel is done" + [1] "all stages have arrived"
    // storage_fp16: S, the transposed product and the leg-2 operand HELD in fp16 on 64-column panels (half.hip):
    // S[] = n rows x Lm columns, `send` = Lm rows x n columns (what leg 1 stores), `recv` = n rows x Lm columns
    int32_t half = 0;
    int64_t rows_pad = 0, rows_pad_t = 0;   // padded rows of an n-row / an Lm-row panel
    float* hand[2] = {nullptr, nullptr};    // hand-back scratch of an fp16 plan: f32 panel-blocked, f32 row-major
    float coef = 0.8f, lbd = 0.f;
    int32_t restrict_support = 0, half_form = 0, n_stages = 1, wire_fp16 = 0;
    int cur = 0;
    int32_t updates = 0;
    // simrank_shardplan_set_timing: HIP events at the boundaries of an update (of the FIRST local plan), on the stream each
    // piece runs on — the kernels' stream and the exchanges' stream
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, hipEvent_t>> marks;       // (tag, event) in issue order
};

namespace simrank {

enum { kMarkUpdate0 = 0, kMarkK0, kMarkK1, kMarkX0, kMarkX1, kMarkLeg2a, kMarkLeg2b, kMarkY0, kMarkY1, kMarkUpdate1 };
static int mark(simrank_shardplan* p, hipStream_t st, int tag) {
    if (!p->timing || p->ev_pool.empty()) return SIMRANK_OK;
    hipEvent_t e = p->ev_pool.back();
    p->ev_pool.pop_back();
    p->marks.push_back({tag, e});
    SR_HIP(hipEventRecord(e, st));
    return SIMRANK_OK;
}
#define SR_MARK(p, st, tag) do { const int rm_ = mark(p, st, tag); if (rm_) return rm_; } while (0)

// ---- one all-to-all: what every local plan sends to / receives from every rank, as lists of pieces (element counts;
// rank s's i-th piece for rank d is rank d's i-th piece from rank s) ----
struct Piece {               // `reps` runs of n elements, `stride` elements apart (the panels of a blocked matrix)
    char* ptr;
    int64_t n;
    int64_t reps, stride;
};
struct Route {
    std::vector<std::vector<Piece>> out, in;          // [peer][piece]
    const float* send_base = nullptr;       // the f32 buffers the pieces lie in, and their fp16 shadows (wire_fp16)
    float* recv_base = nullptr;
    uint16_t* send_h = nullptr;
    uint16_t* recv_h = nullptr;
    explicit Route(int32_t world) : out(world), in(world) {}
    void add_out(int32_t h, const void* ptr, int64_t n, int64_t reps = 1, int64_t stride = 0) {
        if (n && reps) out[h].push_back(Piece{(char*)const_cast<void*>(ptr), n, reps, stride});
    }
    void add_in(int32_t h, void* ptr, int64_t n, int64_t reps = 1, int64_t stride = 0) {
        if (n && reps) in[h].push_back(Piece{(char*)ptr, n, reps, stride});
    }
};

// elem: bytes per element of the buffers (4: f32 — narrowed to fp16 on the way when the plan's wire is fp16; 2: fp16-held)
static int all_to_all(simrank_shardplan* const* plans, int32_t n_local, std::vector<Route>& routes, hipStream_t st, int elem) {
    simrank_shardplan* p0 = plans[0];
    const int32_t P = p0->world;
    const bool wire = p0->wire_fp16 != 0 && elem == 4;
    // fp16 wire: narrow every outgoing piece into the shadow at the same element offset
    if (wire)
        for (int32_t i = 0; i < n_local; ++i)
            for (int32_t h = 0; h < P; ++h)
                for (const Piece& pc : routes[i].out[h]) {
                    const float* src = (const float*)pc.ptr;
                    const int rc = simrank_narrow_h16(src, routes[i].send_h + (src - routes[i].send_base), pc.n, kWireScale, st);
                    if (rc) return rc;
                }
    const size_t bytes = wire ? 2 : size_t(elem);
    auto src_of = [&](Route& r, const Piece& pc) -> const void* {
        return wire ? (const void*)(r.send_h + ((const float*)pc.ptr - r.send_base)) : (const void*)pc.ptr;
    };
    auto dst_of = [&](Route& r, const Piece& pc) -> void* {
        return wire ? (void*)(r.recv_h + ((float*)pc.ptr - r.recv_base)) : (void*)pc.ptr;
    };
    auto matched = [&](const std::vector<Piece>& a, const std::vector<Piece>& b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); ++i)
            if (a[i].n != b[i].n || a[i].reps != b[i].reps) return false;
        return true;
    };
    if (p0->comm->group) {
        for (int32_t s = 0; s < P; ++s)
            for (int32_t d = 0; d < P; ++d) {
                SR_REQUIRE(matched(routes[s].out[d], routes[d].in[s]), "ranks %d and %d disagree about their chunks", s, d);
                for (size_t i = 0; i < routes[s].out[d].size(); ++i) {
                    const Piece& a = routes[s].out[d][i];
                    const Piece& b = routes[d].in[s][i];
                    if (a.reps == 1)
                        SR_HIP(hipMemcpyAsync(dst_of(routes[d], b), src_of(routes[s], a), size_t(a.n) * bytes,
                                              hipMemcpyDeviceToDevice, st));
                    else                                 // (one strided copy for all the panels of a chunk)
                        SR_HIP(hipMemcpy2DAsync(b.ptr, size_t(b.stride) * bytes, a.ptr, size_t(a.stride) * bytes,
                                                size_t(a.n) * bytes, size_t(a.reps), hipMemcpyDeviceToDevice, st));
                }
            }
    } else {
        Rccl* R = p0->comm->api;
        Route& r = routes[0];
        const int32_t me = p0->rank;
        // the chunk a rank addresses to itself never touches the fabric
        SR_REQUIRE(matched(r.out[me], r.in[me]), "own chunk: pieces differ");
        for (size_t i = 0; i < r.out[me].size(); ++i) {
            const Piece& a = r.out[me][i];
            const Piece& b = r.in[me][i];
            if (a.reps == 1)
                SR_HIP(hipMemcpyAsync(dst_of(r, b), src_of(r, a), size_t(a.n) * bytes, hipMemcpyDeviceToDevice, st));
            else
                SR_HIP(hipMemcpy2DAsync(b.ptr, size_t(b.stride) * bytes, a.ptr, size_t(a.stride) * bytes, size_t(a.n) * bytes,
                                        size_t(a.reps), hipMemcpyDeviceToDevice, st));
        }
        const ncclDataType_t dt = bytes == 2 ? ncclHalf : ncclFloat;
        SR_RCCL(R->GroupStart());
        for (int32_t h = 0; h < P; ++h) {
            if (h == me) continue;
            // (a strided piece = one send / receive per panel: each is a contiguous run of >= 0.5 MB at the sizes this mode
            // is for, and lands where leg 2 reads it — no staging copy on either side)
            for (const Piece& pc : r.out[h])
                for (int64_t k = 0; k < pc.reps; ++k)
                    SR_RCCL(R->Send(pc.reps == 1 ? src_of(r, pc) : (const void*)(pc.ptr + size_t(k) * size_t(pc.stride) * bytes),
                                    size_t(pc.n), dt, h, p0->comm->nccl, st));
            for (const Piece& pc : r.in[h])
                for (int64_t k = 0; k < pc.reps; ++k)
                    SR_RCCL(R->Recv(pc.reps == 1 ? dst_of(r, pc) : (void*)(pc.ptr + size_t(k) * size_t(pc.stride) * bytes),
                                    size_t(pc.n), dt, h, p0->comm->nccl, st));
        }
        SR_RCCL(R->GroupEnd());
    }
    if (wire)
        for (int32_t i = 0; i < n_local; ++i)
            for (int32_t h = 0; h < P; ++h)
                for (const Piece& pc : routes[i].in[h]) {
                    float* dst = (float*)pc.ptr;
                    const int rc = simrank_widen_h16(routes[i].recv_h + (dst - routes[i].recv_base), dst, pc.n, kWireScale, st);
                    if (rc) return rc;
                }
    return SIMRANK_OK;
}

static int check_group(simrank_shardplan* const* plans, int32_t n_local) {
    SR_REQUIRE(plans && n_local >= 1 && plans[0], "no plans");
    simrank_shardplan* p0 = plans[0];
    if (p0->comm->group) {
        SR_REQUIRE(n_local == p0->world, "an in-process group advances all its %d plans together (%d given)", p0->world, n_local);
        for (int32_t i = 0; i < n_local; ++i)
            SR_REQUIRE(plans[i] && plans[i]->comm->group == p0->comm->group && plans[i]->rank == i &&
                           plans[i]->stream == p0->stream && plans[i]->n == p0->n &&
                           plans[i]->half_form == p0->half_form && plans[i]->n_stages == p0->n_stages &&
                           plans[i]->sh_stages.size() == p0->sh_stages.size() &&
                           plans[i]->wire_fp16 == p0->wire_fp16 && plans[i]->half == p0->half && plans[i]->asym == p0->asym,
                       "plans[%d] is not rank %d of the same in-process group, stream and options", i, i);
    } else {
        SR_REQUIRE(n_local == 1, "a process of a multi-process world holds one plan");
    }
    return SIMRANK_OK;
}

static void fill_epilogue(simrank_shardplan* p, double eps, int32_t exact_count, simrank_epilogue* ep) {
    *ep = simrank_epilogue{};
    ep->coef = p->coef;
    ep->lbd = p->lbd;
    ep->evidence = p->ev;
    ep->ld_evidence = p->ld_ev;
    ep->apriori = p->prior;
    ep->ld_apriori = p->ld;
    ep->previous = p->S[p->cur];
    ep->ld_previous = p->ld;
    ep->eps = eps;
    ep->n_changed = p->counters;
    ep->diag_col0 = p->m_lo;
    ep->set_diag = 1;
    ep->symmetric = 0;
    if (p->half) ep->ld_evidence = ep->ld_apriori = ep->ld_previous = 0;      // (panel-blocked: the padded rows say it all)
    ep->restrict_support = p->restrict_support;
    ep->count_any = exact_count ? 0 : 1;
}

// One update on every local plan, queued; its count lands in pinned slot `slot` of every plan.
static int update(simrank_shardplan* const* plans, int32_t n_local, double eps, int32_t exact_count, int slot) {
    simrank_shardplan* p0 = plans[0];
    const int32_t P = p0->world, S = p0->n_stages;
    const bool local = p0->comm->group != nullptr;
    hipStream_t xs = local ? p0->stream : p0->xstream;
    // X -> (W X)^T in column stages, the chunks of each stage leaving for their ranks behind its kernel.  second = false: leg 1
    // (X = the rank's block of the operand matrix, Lk columns; what arrives is the leg-2 operand `recv`).  second = true
    // (asymmetric iterates only): leg 2 as the same launch on `recv` (Lm columns) — W . Tt is the TRANSPOSE of the wanted
    // block, so its product travels exactly like leg 1's and lands in `recv2`.
    const int32_t walign = p0->half ? 64 : 32;            // stage widths: whole panels of the operand
    auto transposed_leg = [&](bool second) -> int {
        for (int32_t s = 0; s < S; ++s) {
            std::vector<Route> routes(n_local, Route(P));
            if (!second) SR_MARK(p0, p0->stream, kMarkK0);
            for (int32_t i = 0; i < n_local; ++i) {
                simrank_shardplan* p = plans[i];
                const int64_t L = second ? p->Lm : p->Lk;
                const int64_t w = stage_width(L, S, s, walign), c0 = stage_col0(L, S, s, walign);
                const simrank_shardplan* sp = p->src ? p->src : p;      // whose block leg 1 reads
                Route& r = routes[i];
                if (p->half) {
                    // fp16-held: leg 1 stores rows [c0, c0 + w) of every 64-column panel of (W.S_block)^T (Lm rows x n
                    // columns); the panels of rank h's columns go to rank h and land, panel by panel, at the rows of its
                    // leg-2 operand this rank's columns own
                    uint16_t* T = reinterpret_cast<uint16_t*>(p->send);
                    uint16_t* X2 = reinterpret_cast<uint16_t*>(p->recv);
                    if (w) {
                        const int rc = simrank_spmm_blocked_h16(p->g, reinterpret_cast<uint16_t*>(p->S[p->cur]) + (c0 / 64) * p->rows_pad * 64,
                                                                p->rows_pad, w, T + c0 * 64, p->rows_pad_t, 1, nullptr, 0,
                                                                kHalfScale, p->stream);
                        if (rc) return rc;
                    }
                    for (int32_t h = 0; h < P; ++h) {
                        int64_t lo, hi;
                        part(p->n, P, h, &lo, &hi);
                        r.add_out(h, T + ((lo / 64) * p->rows_pad_t + c0) * 64, w * 64, (hi - lo) / 64, p->rows_pad_t * 64);
                        const int64_t wh = stage_width(hi - lo, S, s, walign), ch = stage_col0(hi - lo, S, s, walign);
                        r.add_in(h, X2 + (lo + ch) * 64, wh * 64, p->Lm / 64, p->rows_pad * 64);
                    }
                } else {
                    const float* X = second ? p->recv + c0 : sp->S[sp->cur] + c0;
                    const int64_t ldx = second ? p->recv_ld : sp->ld;
                    float* outb = second ? p->send2 : p->send;
                    float* inb = second ? p->recv2 : p->recv;
                    const int64_t send_off = c0 * p->send_ld;
                    if (w) {
                        const int rc = simrank_spmm(p->g, X, ldx, w, outb + send_off, 0, 1, p->mb, p->pad, nullptr, p->stream);
                        if (rc) return rc;
                    }
                    r.send_base = outb; r.recv_base = inb;
                    r.send_h = p->wire[second ? 4 : 0]; r.recv_h = p->wire[second ? 5 : 1];
                    for (int32_t h = 0; h < P; ++h) {
                        r.add_out(h, outb + send_off + int64_t(h) * w * (p->mb + p->pad), w * (span(p->n, P, h) + p->pad));
                        // what rank h computed: the columns of ITS block — of the operand matrix (leg 1: k nodes) or of the
                        // result itself (second: n nodes) — which are rows here
                        int64_t k_lo, k_hi;
                        part(second ? p->n : p->k, P, h, &k_lo, &k_hi);
                        const int64_t wh = stage_width(k_hi - k_lo, S, s, walign), ch = stage_col0(k_hi - k_lo, S, s, walign);
                        // (a rank without columns receives rows of zero length)
                        r.add_in(h, inb + (k_lo + ch) * p->recv_ld, p->Lm ? wh * p->recv_ld : 0);
                    }
                }
                if (!second && i == n_local - 1) SR_MARK(p0, p0->stream, kMarkK1);
                if (!local) {                                // RCCL's stream waits for this stage's kernel only
                    SR_HIP(hipEventRecord(p->staged[s], p->stream));
                    SR_HIP(hipStreamWaitEvent(xs, p->staged[s], 0));
                }
            }
            if (!second) SR_MARK(p0, xs, kMarkX0);
            const int rc = all_to_all(plans, n_local, routes, xs, p0->half ? 2 : 4);
            if (rc) return rc;
            if (!second) SR_MARK(p0, xs, kMarkX1);
        }
        return SIMRANK_OK;
    };
    SR_MARK(p0, p0->stream, kMarkUpdate0);
    {
        const int rc1 = transposed_leg(false);               // leg 1 + exchange 1, stage by stage
        if (rc1) return rc1;
    }
    // every RCCL call of this communicator is issued on ITS stream, in one order; `hop` makes one stream wait for the other
    auto hop = [&](hipStream_t from, hipStream_t to, hipEvent_t ev) -> int {
        if (from == to) return SIMRANK_OK;
        SR_HIP(hipEventRecord(ev, from));
        SR_HIP(hipStreamWaitEvent(to, ev, 0));
        return SIMRANK_OK;
    };
    int rc = local ? SIMRANK_OK : hop(xs, p0->stream, p0->staged[S]);      // leg 2 reads what the last stage delivered
    if (rc) return rc;
    // leg 2 with the fused epilogue and count
    SR_MARK(p0, p0->stream, kMarkLeg2a);
    if (p0->asym) {
        // asymmetric iterates: leg 2's product goes round once more (transposed_leg), then the epilogue — coefficient, evidence,
        // prior, diagonal, the exact count — over what arrived
        rc = transposed_leg(true);
        if (!rc && !local) rc = hop(xs, p0->stream, p0->staged[S]);
        if (rc) return rc;
    }
    for (int32_t i = 0; i < n_local; ++i) {
        simrank_shardplan* p = plans[i];
        simrank_epilogue ep;
        fill_epilogue(p, eps, exact_count, &ep);
        if (p->asym && p->Lm) {
            rc = simrank_epilogue_apply(p->recv2, p->recv_ld, p->S[p->cur ^ 1], p->ld, p->n, p->Lm, &ep, p->stream);
        } else if (!p->Lm) {
            SR_HIP(hipMemsetAsync(p->counters, 0, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, p->stream));
        } else if (p->half) {
            rc = simrank_spmm_blocked_h16(p->g, p->recv, p->rows_pad, p->Lm, p->S[p->cur ^ 1], p->rows_pad, 0, &ep, p->rows_pad,
                                          kHalfScale, p->stream);
        } else if (p->half_form && !p->sh_stages.empty()) {
            continue;                                    // (stage by stage, below)
        } else if (p->half_form) {
            rc = simrank_spmm_shard(p->g, p->recv, p->recv_ld, p->S[p->cur ^ 1], p->ld, &ep, p->rank, P, p->sh_send,
                                    p->sh_chunk, p->stream);
        } else {
            rc = simrank_spmm(p->g, p->recv, p->recv_ld, p->Lm, p->S[p->cur ^ 1], p->ld, 0, 0, 0, &ep, p->stream);
        }
        if (rc) return rc;
    }
    // the mirrored tiles of one stage (or of the whole leg): equal chunks, the one a rank addresses to itself is empty
    auto exchange_mirrors = [&](int64_t off, int64_t chunk) -> int {
        std::vector<Route> routes(n_local, Route(P));
        for (int32_t i = 0; i < n_local; ++i) {
            simrank_shardplan* p = plans[i];
            Route& r = routes[i];
            r.send_base = p->sh_send; r.recv_base = p->sh_recv;
            r.send_h = p->wire[2]; r.recv_h = p->wire[3];
            for (int32_t h = 0; h < P; ++h) {
                if (h == p->rank) continue;
                r.add_out(h, p->sh_send + off + int64_t(h) * chunk, chunk);
                r.add_in(h, p->sh_recv + off + int64_t(h) * chunk, chunk);
            }
        }
        return all_to_all(plans, n_local, routes, xs, 4);
    };
    const bool staged2 = p0->half_form && !p0->sh_stages.empty();
    if (staged2) {
        // half-form leg 2 stage by stage, heaviest tiles first: a stage's mirrored tiles leave (on RCCL's stream, behind an
        // event) while the next stage computes; the counters are zeroed by the first stage only
        for (size_t k = 0; k < p0->sh_stages.size(); ++k) {
            for (int32_t i = 0; i < n_local; ++i) {
                simrank_shardplan* p = plans[i];
                const simrank_shardplan::Stage2& sg = p->sh_stages[k];
                simrank_epilogue ep;
                fill_epilogue(p, eps, exact_count, &ep);
                rc = simrank_spmm_shard_stage(p->g, p->recv, p->recv_ld, p->S[p->cur ^ 1], p->ld, &ep, p->rank, P,
                                              p->sh_send + sg.off, sg.chunk, sg.tile_lo, sg.tile_hi, k == 0 ? 1 : 0, p->stream);
                if (rc) return rc;
                if (!local) {
                    SR_HIP(hipEventRecord(p->staged2[k], p->stream));
                    SR_HIP(hipStreamWaitEvent(xs, p->staged2[k], 0));
                }
            }
            if (p0->sh_stages[k].chunk) {
                rc = exchange_mirrors(p0->sh_stages[k].off, p0->sh_stages[k].chunk);
                if (rc) return rc;
            }
        }
    }
    SR_MARK(p0, p0->stream, kMarkLeg2b);
    if (!local) {
        rc = hop(p0->stream, xs, p0->staged[0]);
        if (rc) return rc;
        SR_MARK(p0, xs, kMarkY0);
        if (P > 1)                                       // the count of the whole update, on every rank
            SR_RCCL(p0->comm->api->AllReduce(p0->counters, p0->counters, SIMRANK_CHANGED_SLOTS, ncclUint64, ncclSum,
                                             p0->comm->nccl, xs));
    }
    if (p0->half_form && !staged2) {
        // exchange 2 after the whole leg: the packed mirrored tiles
        if (local) SR_MARK(p0, xs, kMarkY0);
        rc = exchange_mirrors(0, p0->sh_chunk);
        if (rc) return rc;
        if (local) SR_MARK(p0, xs, kMarkY1);
    }
    if (!local) {
        SR_MARK(p0, xs, kMarkY1);
        rc = hop(xs, p0->stream, p0->staged[S]);
        if (rc) return rc;
    }
    for (int32_t i = 0; i < n_local; ++i) {
        simrank_shardplan* p = plans[i];
        SR_HIP(hipMemcpyAsync(p->host_counters[slot], p->counters, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS,
                              hipMemcpyDeviceToHost, p->stream));
        SR_HIP(hipEventRecord(p->counted[slot], p->stream));
        if (p->half_form && !p->sh_stages.empty()) {
            for (const simrank_shardplan::Stage2& sg : p->sh_stages) {
                if (!sg.chunk) continue;
                rc = simrank_shard_unpack_stage(p->S[p->cur ^ 1], p->ld, p->sh_recv + sg.off, sg.chunk, p->rank, P, p->n,
                                                sg.tile_lo, sg.tile_hi, p->stream);
                if (rc) return rc;
            }
        } else if (p->half_form) {
            rc = simrank_shard_unpack(p->S[p->cur ^ 1], p->ld, p->sh_recv, p->sh_chunk, p->rank, P, p->n, p->stream);
            if (rc) return rc;
        }
    }
    SR_MARK(p0, p0->stream, kMarkUpdate1);
    return SIMRANK_OK;
}

// the count of the update that used `slot`, over all ranks; waits for that update only
static int read_count(simrank_shardplan* const* plans, int32_t n_local, int slot, unsigned long long* sum) {
    unsigned long long t = 0;
    for (int32_t i = 0; i < n_local; ++i) {
        SR_HIP(hipEventSynchronize(plans[i]->counted[slot]));
        for (int k = 0; k < SIMRANK_CHANGED_SLOTS; ++k) t += plans[i]->host_counters[slot][k];
    }
    *sum = t;
    return SIMRANK_OK;
}

static void flip(simrank_shardplan* const* plans, int32_t n_local) {
    for (int32_t i = 0; i < n_local; ++i) plans[i]->cur ^= 1;
}

// the rank's block with its rows in the caller's order, on the device (in the idle ping-pong partner)
static int block_rows_in_callers_order(simrank_shardplan* p, float** out) {
    if (p->half) {
        // fp16-held: widen into an f32 panel-blocked scratch copy, then out of that layout and the solver's row order
        const size_t wide = size_t((p->Lm + 31) / 32) * size_t(p->rows_pad) * 32 * 4, rowm = size_t(p->n) * size_t(p->ld) * 4;
        if (!p->hand[0]) SR_HIP(pool_hip_alloc((void**)&p->hand[0], wide));
        if (!p->hand[1]) SR_HIP(pool_hip_alloc((void**)&p->hand[1], rowm));
        int rc = simrank_widen_blocked_h16(p->S[p->cur], p->rows_pad, p->hand[0], p->rows_pad, p->n, p->Lm, kHalfScale, p->stream);
        if (!rc) rc = simrank_permute_layout(p->hand[0], 32, p->rows_pad, p->hand[1], p->ld, 0, p->n, p->Lm, p->inv, nullptr, 4, p->stream);
        *out = p->hand[1];
        return rc;
    }
    float* tmp = p->S[p->cur ^ 1];
    if (p->Lm) {
        const int rc = simrank_permute(p->S[p->cur], p->ld, tmp, p->ld, p->n, p->Lm, p->inv, nullptr, 4, p->stream);
        if (rc) return rc;
    }
    *out = tmp;
    return SIMRANK_OK;
}

}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_comm_unique_id(void* id_bytes) {
    SR_REQUIRE(id_bytes, "id is NULL");
    Rccl* R = rccl();
    SR_REQUIRE(R->ok, "RCCL could not be loaded (librccl.so.1; set SIMRANK_RCCL_LIB)");
    ncclUniqueId id;
    SR_RCCL(R->GetUniqueId(&id));
    std::memcpy(id_bytes, &id, sizeof id);
    return SIMRANK_OK;
}

int simrank_comm_create(const void* id_bytes, int32_t rank, int32_t world, simrank_comm** out) {
    SR_REQUIRE(id_bytes && out && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    *out = nullptr;
    Rccl* R = rccl();
    SR_REQUIRE(R->ok, "RCCL could not be loaded (librccl.so.1; set SIMRANK_RCCL_LIB)");
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    ncclComm_t c = nullptr;
    SR_RCCL(R->CommInitRank(&c, world, id, rank));
    simrank_comm* k = new simrank_comm;
    k->rank = rank; k->world = world; k->nccl = c; k->owned = true; k->api = R;
    *out = k;
    return SIMRANK_OK;
}

int simrank_comm_adopt(void* rccl_comm, int32_t rank, int32_t world, simrank_comm** out) {
    SR_REQUIRE(rccl_comm && out && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    SR_REQUIRE(rccl()->ok, "RCCL could not be loaded (librccl.so.1; set SIMRANK_RCCL_LIB)");
    simrank_comm* k = new simrank_comm;
    k->rank = rank; k->world = world; k->nccl = (ncclComm_t)rccl_comm; k->owned = false; k->api = rccl();
    *out = k;
    return SIMRANK_OK;
}

int simrank_comm_local_group(int32_t world, simrank_comm** out) {
    SR_REQUIRE(out && world >= 1 && world <= 64, "an in-process group has 1 .. 64 ranks");
    LocalGroup* g = new LocalGroup;
    g->world = world;
    g->alive = world;
    for (int32_t r = 0; r < world; ++r) {
        simrank_comm* k = new simrank_comm;
        k->rank = r; k->world = world; k->group = g;
        out[r] = k;
    }
    return SIMRANK_OK;
}

int simrank_comm_thread_group(int32_t world, simrank_comm** out) {
    SR_REQUIRE(out && world >= 1 && world <= 64, "a thread group has 1 .. 64 ranks");
    TGroup* g = new TGroup;
    g->world = world;
    g->alive = world;
    g->box.resize(size_t(world) * size_t(world));
    if (const char* e = std::getenv("SIMRANK_THREAD_COMM_TIMEOUT")) g->timeout_s = std::max(1.0, std::atof(e));
    for (int32_t r = 0; r < world; ++r) {
        simrank_comm* k = new simrank_comm;
        TComm* t = new TComm;
        t->g = g;
        t->rank = r;
        k->rank = r; k->world = world; k->thread = t; k->nccl = reinterpret_cast<ncclComm_t>(t); k->api = thread_api();
        out[r] = k;
    }
    return SIMRANK_OK;
}

int simrank_comm_destroy(simrank_comm* c) {
    if (!c) return SIMRANK_OK;
    if (c->group && --c->group->alive == 0) delete c->group;
    if (c->thread) {
        TComm* t = c->thread;
        for (hipEvent_t e : t->used) (void)hipEventDestroy(e);
        if (t->tmp) (void)hipFree(t->tmp);
        TGroup* g = t->g;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            last = --g->alive == 0;
        }
        if (last) {
            for (auto& q : g->box)
                for (TMsg* m : q) delete m;
            delete g;
        }
        delete t;
    } else if (c->nccl && c->owned) {
        (void)rccl()->CommDestroy(c->nccl);
    }
    delete c;
    return SIMRANK_OK;
}

int simrank_shardplan_destroy(simrank_shardplan* p) {
    if (!p) return SIMRANK_OK;
    if (p->xstream) (void)hipStreamSynchronize(p->xstream);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (void* b : {(void*)p->S[0], (void*)p->S[1], (void*)p->send, (void*)p->recv, (void*)p->sh_send, (void*)p->sh_recv,
                    (void*)p->wire[0], (void*)p->wire[1], (void*)p->wire[2], (void*)p->wire[3], (void*)p->wire[4],
                    (void*)p->wire[5], (void*)p->ev,
                    (void*)p->prior, (void*)p->inv, (void*)p->counters, (void*)p->hand[0], (void*)p->hand[1], (void*)p->send2,
                    (void*)p->recv2})
        (void)pool_free(b);
    for (int i = 0; i < 2; ++i) {
        if (p->host_counters[i]) (void)hipHostFree(p->host_counters[i]);
        if (p->counted[i]) (void)hipEventDestroy(p->counted[i]);
    }
    for (hipEvent_t e : p->staged) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->staged2) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    for (auto& m : p->marks) (void)hipEventDestroy(m.second);
    if (p->xstream) (void)hipStreamDestroy(p->xstream);
    simrank_graph_destroy(p->g);
    delete p;
    return SIMRANK_OK;
}

}  // extern "C"

namespace simrank {
// One rank's share of ONE similarity matrix: everything simrank_shardplan_create and simrank_shardbiplan_create have in
// common.  `pat` is the pattern in the solver's order, n rows x k columns (k = n: the directed classes).
struct SideIn {
    int64_t n = 0, k = 0, nnz = 0;
    const PlanPrepView* pat = nullptr;
    float coef = 0.8f, lbd = 0.f;
    const float* apriori = nullptr;          // HOST n x n, the caller's order
    int64_t ld_apriori = 0;
    int32_t evidence = 0;                    // 1: counts of `pat` itself
    bool half_form = false, fp16 = false, asym = false;
    int32_t stages = 0, wire_fp16 = 0;
};

static int create_side(const SideIn& in, simrank_comm* comm, void* stream, simrank_shardplan** out) {
    SR_REQUIRE(!(in.half_form && in.asym), "a plan cannot be both in the half form of leg 2 and asymmetric");
    *out = nullptr;
    const int32_t P = comm->world;
    const int64_t n = in.n, k = in.k;
    const bool fp16 = in.fp16;
    const std::vector<int32_t>& ord = *in.pat->ord;
    const std::vector<int32_t>& inv = *in.pat->inv;
    simrank_shardplan* p = new simrank_shardplan;
    p->comm = comm; p->rank = comm->rank; p->world = P;
    p->n = n;
    p->k = k;
    p->stream = as_stream(stream);
    p->coef = in.coef; p->lbd = in.lbd;
    p->half_form = in.half_form ? 1 : 0;
    p->wire_fp16 = in.wire_fp16 ? 1 : 0;
    p->asym = in.asym ? 1 : 0;
    p->mb = (n + P - 1) / P;
    part(n, P, p->rank, &p->m_lo, &p->m_hi);
    p->Lm = p->m_hi - p->m_lo;
    {
        int64_t k_lo, k_hi;
        part(k, P, p->rank, &k_lo, &k_hi);
        p->Lk = k_hi - k_lo;
    }
    p->ld = pitch(p->Lm, 4);
    p->ld_ev = pitch(p->Lm, 1);
    p->pad = row_pad(p->mb);
    p->send_ld = int64_t(P) * (p->mb + p->pad);          // (>= n + P pad: only the last chunk can be short)
    p->recv_ld = std::max<int64_t>(1, p->Lm + p->pad);
    // stages of exchange 1: by the width of the LARGEST block of the operand, so every rank takes the same number (driver.auto_stages)
    p->n_stages = in.stages > 0 ? in.stages : int32_t(std::max<int64_t>(1, std::min<int64_t>(4, ((k + P - 1) / P) / 2048)));
    p->ord = ord;
    auto fail = [&](int code) { simrank_shardplan_destroy(p); return code; };
    p->half = fp16 ? 1 : 0;
    p->rows_pad = (n + 7) / 8 * 8 + 8;
    p->rows_pad_t = (p->Lm + 7) / 8 * 8 + 8;
    int rc;
    {
        Tuning t = tuning_snapshot();
        if (fp16) t.fuse_unit = int64_t(1) << 20;        // (half.hip runs whole blocks: no units whose sums meet in memory)
        rc = graph_create_with(t, n, k, in.nnz, in.pat->rp->data(), in.pat->cl->data(), in.pat->rs->data(), &p->g);
    }
    if (rc) return fail(rc);
    if (fp16 && !p->g->fused) {
        set_error("storage_fp16 needs the one-launch plan (tuning fuse = 1) and a graph that has one");
        return fail(SIMRANK_ERR_INVALID);
    }
#define SP_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                              \
            return fail(e_ == hipErrorOutOfMemory ? SIMRANK_ERR_ALLOC : SIMRANK_ERR_HIP);         \
        }                                                                                         \
    } while (0)
    auto dev = [&](void** b, size_t bytes) -> hipError_t {
        hipError_t e = pool_hip_alloc(b, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) e = hipMemsetAsync(*b, 0, std::max<size_t>(bytes, 16), p->stream);
        return e;
    };
    const size_t blk = fp16 ? size_t(p->Lm / 64) * size_t(p->rows_pad) * 128 : size_t(n) * size_t(p->ld) * 4;
    SP_HIP(dev((void**)&p->S[0], blk));
    SP_HIP(dev((void**)&p->S[1], blk));
    const size_t send_floats = size_t(std::max<int64_t>(1, p->Lk)) * size_t(p->send_ld);     // Lk columns of (W.S_block)^T
    const size_t recv_floats = size_t(k) * size_t(p->recv_ld);                               // the leg-2 operand: k rows
    if (fp16) {
        SP_HIP(dev((void**)&p->send, size_t(n / 64) * size_t(p->rows_pad_t) * 128));      // (W.S_block)^T: Lm rows x n columns
        SP_HIP(dev((void**)&p->recv, blk));                                                // leg-2 operand: n rows x Lm columns
    } else {
        SP_HIP(dev((void**)&p->send, send_floats * 4));
        SP_HIP(dev((void**)&p->recv, recv_floats * 4));
    }
    if (p->asym) {
        const size_t s2 = size_t(std::max<int64_t>(1, p->Lm)) * size_t(p->send_ld), r2 = size_t(n) * size_t(p->recv_ld);
        SP_HIP(dev((void**)&p->send2, s2 * 4));
        SP_HIP(dev((void**)&p->recv2, r2 * 4));
        if (p->wire_fp16) {
            SP_HIP(dev((void**)&p->wire[4], s2 * 2));
            SP_HIP(dev((void**)&p->wire[5], r2 * 2));
        }
    }
    if (p->wire_fp16) {
        SP_HIP(dev((void**)&p->wire[0], send_floats * 2));
        SP_HIP(dev((void**)&p->wire[1], recv_floats * 2));
    }
    if (p->half_form) {
        const int64_t t = p->mb / 32;
        p->sh_chunk = std::max<int64_t>(1, t * (t - 1) / 2 * 1024);
        const size_t sh = size_t(P) * size_t(p->sh_chunk);
        // leg 2 in as many stages as exchange 1 (at most one per two column tiles), cut at T sqrt(k / S) for equal slot counts
        const int64_t want = std::min<int64_t>(p->n_stages, t / 2);
        if (want > 1) {
            std::vector<int64_t> cuts;
            for (int64_t k = 1; k < want; ++k)
                cuts.push_back(std::min<int64_t>(t, std::max<int64_t>(2, (int64_t)std::llround(double(t) * std::sqrt(double(k) / double(want))))));
            cuts.push_back(t);
            std::sort(cuts.begin(), cuts.end());
            cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
            int64_t lo = 0;
            for (int64_t hi : cuts) {
                const int64_t s_lo = lo * (lo - 1) / 2, s_hi = hi * (hi - 1) / 2;
                p->sh_stages.push_back({(int32_t)lo, (int32_t)hi, int64_t(P) * s_lo * 1024, (s_hi - s_lo) * 1024});
                lo = hi;
            }
            std::reverse(p->sh_stages.begin(), p->sh_stages.end());
        }
        SP_HIP(dev((void**)&p->sh_send, sh * 4));
        SP_HIP(dev((void**)&p->sh_recv, sh * 4));
        if (p->wire_fp16) {
            SP_HIP(dev((void**)&p->wire[2], sh * 2));
            SP_HIP(dev((void**)&p->wire[3], sh * 2));
        }
    }
    SP_HIP(dev((void**)&p->counters, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS));
    for (int i = 0; i < 2; ++i) {
        SP_HIP(hipHostMalloc((void**)&p->host_counters[i], sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, hipHostMallocPortable));
        SP_HIP(hipEventCreateWithFlags(&p->counted[i], hipEventDisableTiming));
    }
    if (!comm->group) {
        SP_HIP(hipStreamCreateWithFlags(&p->xstream, hipStreamNonBlocking));
        p->staged.resize(size_t(p->n_stages) + 1, nullptr);
        for (hipEvent_t& e : p->staged) SP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->staged2.resize(p->sh_stages.size(), nullptr);
        for (hipEvent_t& e : p->staged2) SP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    SP_HIP(dev((void**)&p->inv, size_t(n) * 4));
    SP_HIP(hipMemcpyAsync(p->inv, inv.data(), size_t(n) * 4, hipMemcpyHostToDevice, p->stream));
    SP_HIP(hipStreamSynchronize(p->stream));             // (inv may be a host vector about to go away)
    if (in.evidence && p->Lm) {
        // common in-neighbour counts of the rank's columns (SimRank.py:311-320), 1 - 2^-count in the epilogue
        if (fp16) {                                      // (32-column panels of n padded rows, as the single plan's)
            SP_HIP(dev((void**)&p->ev, size_t((p->Lm + 31) / 32) * size_t(p->rows_pad) * 32));
            rc = simrank_evidence_counts_blocked(p->g, p->m_lo, p->Lm, p->ev, p->rows_pad, p->stream);
            if (rc) return fail(rc);
        } else {
            SP_HIP(dev((void**)&p->ev, size_t(n) * size_t(p->ld_ev)));
            rc = simrank_evidence_counts(p->g, p->m_lo, p->Lm, p->ev, p->ld_ev, p->stream);
            if (rc) return fail(rc);
            int64_t live = 0, total = 1;
            rc = simrank_evidence_live_segments(p->ev, p->ld_ev, 0, n, p->Lm, &live, &total, p->stream);
            if (rc) return fail(rc);
            p->restrict_support = 2 * live < total ? 1 : 0;
        }
    }
    if (in.apriori && p->Lm) {
        // the rank's columns of the prior in the solver's order: block[i][j] = A[ord[i]][ord[m_lo + j]]
        std::vector<float> host(size_t(n) * size_t(p->ld), 0.f);
        for (int64_t i = 0; i < n; ++i) {
            const float* src = in.apriori + int64_t(ord[(size_t)i]) * in.ld_apriori;
            float* dst = host.data() + i * p->ld;
            for (int64_t j = 0; j < p->Lm; ++j) dst[j] = src[ord[(size_t)(p->m_lo + j)]];
        }
        SP_HIP(dev((void**)&p->prior, blk));
        SP_HIP(hipMemcpyAsync(p->prior, host.data(), blk, hipMemcpyHostToDevice, p->stream));
        SP_HIP(hipStreamSynchronize(p->stream));
    }
#undef SP_HIP
    if (p->Lm) {
        rc = fp16 ? simrank_fill_identity_blocked_h16(p->S[0], n, p->Lm, p->rows_pad, p->m_lo, kHalfScale, p->stream)
                  : simrank_fill_identity(p->S[0], n, p->Lm, p->ld, p->m_lo, p->stream);
        if (rc) return fail(rc);
    }
    *out = p;
    return SIMRANK_OK;
}
}  // namespace simrank

extern "C" {

int simrank_shardplan_create(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* rowscale,
                             const simrank_shardplan_options* opt, simrank_comm* comm, void* stream,
                             simrank_shardplan** out) {
    SR_REQUIRE(out, "out is NULL");
    *out = nullptr;
    SR_REQUIRE(opt && comm, "options / communicator missing");
    const int32_t P = comm->world;
    SR_REQUIRE(opt->leg2_form >= -1 && opt->leg2_form <= 1 && opt->stages >= 0 && opt->stages <= 64, "bad options");
    const bool fits_half = P > 1 && n % (32 * int64_t(P)) == 0;
    SR_REQUIRE(opt->leg2_form != 1 || fits_half || (P == 1 && n % 32 == 0),
               "the half form of leg 2 needs n to be a multiple of 32 x ranks");
    const bool fp16 = opt->storage_fp16 != 0;
    // fp16-held matrices (half.hip): every block whole 64-column panels, leg 2 in its full form, no prior, no fp16 wire
    // (the exchange moves the fp16 values themselves)
    SR_REQUIRE(!fp16 || (n % (64 * int64_t(P)) == 0 && opt->leg2_form != 1 && !opt->apriori && !opt->wire_fp16),
               "storage_fp16 on shards needs n %% (64 x ranks) == 0, leg 2 in its full form, no prior and the f32 wire option off");
    // a prior that is not symmetric: asymmetric iterates — f32, leg 2 in its full form (there is no mirror image to share);
    // the second exchange carries leg 2's product back to the ranks that own its columns
    const bool asym = opt->apriori && opt->ld_apriori >= n && !prior_symmetric(opt->apriori, opt->ld_apriori, n);
    SR_REQUIRE(!asym || (!fp16 && opt->leg2_form != 1),
               "a prior that is not symmetric needs f32 matrices and leg 2 in its full form");
    const bool half = !fp16 && !asym && (opt->leg2_form == 1 || (opt->leg2_form == -1 && fits_half && P >= 8));
    PlanPrep pp;
    int rc = shard_prepare(n, nnz, rowptr, col, rowscale, opt->apriori, opt->ld_apriori, opt->reorder != 0, half ? P : 1, &pp, true);
    if (rc) return rc;
    PlanPrepView view{&pp.ord, &pp.inv, &pp.rp, &pp.cl, &pp.rs};
    SideIn in;
    in.n = in.k = n;
    in.nnz = nnz;
    in.pat = &view;
    in.coef = opt->coef; in.lbd = opt->lbd;
    in.apriori = opt->apriori; in.ld_apriori = opt->ld_apriori;
    in.evidence = opt->evidence ? 1 : 0;
    in.half_form = half; in.fp16 = fp16; in.asym = pp.asym;
    in.stages = opt->stages; in.wire_fp16 = opt->wire_fp16;
    return create_side(in, comm, stream, out);
}

int simrank_shardplan_set_timing(simrank_shardplan* p, int32_t updates) {
    SR_REQUIRE(p && updates >= 0 && updates <= 4096, "bad timing arguments");
    for (auto& m : p->marks) p->ev_pool.push_back(m.second);
    p->marks.clear();
    const size_t per_update = size_t(4 * p->n_stages + 8);
    while (p->ev_pool.size() < per_update * size_t(updates)) {
        hipEvent_t e;
        SR_HIP(hipEventCreate(&e));
        p->ev_pool.push_back(e);
    }
    p->timing = updates > 0;
    return SIMRANK_OK;
}

int simrank_shardplan_timings(simrank_shardplan* p, double* ms, int32_t n_ms, int32_t* updates) {
    SR_REQUIRE(p && ms && n_ms >= 6, "bad timing arguments");
    SR_HIP(hipStreamSynchronize(p->stream));
    if (p->xstream) SR_HIP(hipStreamSynchronize(p->xstream));
    // ms[0] leg 1 (the stages' kernels), [1] exchange 1 (its stages on the exchange stream), [2] what the kernels' stream
    // waited between the last stage's kernel and leg 2, [3] leg 2, [4] all-reduce of the count + exchange 2, [5] the update
    double acc[6] = {0, 0, 0, 0, 0, 0};
    int32_t n_upd = 0;
    hipEvent_t open[10] = {};
    hipEvent_t last_k1 = nullptr;
    auto el = [&](hipEvent_t a, hipEvent_t b) -> double {
        float t = 0.f;
        if (a && b && hipEventElapsedTime(&t, a, b) != hipSuccess) { (void)hipGetLastError(); t = 0.f; }
        return (double)t;
    };
    for (auto& m : p->marks) {
        switch (m.first) {
            case kMarkUpdate0: open[kMarkUpdate0] = m.second; break;
            case kMarkK0: open[kMarkK0] = m.second; break;
            case kMarkK1: acc[0] += el(open[kMarkK0], m.second); last_k1 = m.second; break;
            case kMarkX0: open[kMarkX0] = m.second; break;
            case kMarkX1: acc[1] += el(open[kMarkX0], m.second); break;
            case kMarkLeg2a: acc[2] += el(last_k1, m.second); open[kMarkLeg2a] = m.second; break;
            case kMarkLeg2b: acc[3] += el(open[kMarkLeg2a], m.second); break;
            case kMarkY0: open[kMarkY0] = m.second; break;
            case kMarkY1: acc[4] += el(open[kMarkY0], m.second); break;
            case kMarkUpdate1: acc[5] += el(open[kMarkUpdate0], m.second); ++n_upd; break;
            default: break;
        }
    }
    for (int i = 0; i < 6; ++i) ms[i] = n_upd ? acc[i] / n_upd : 0.0;
    if (updates) *updates = n_upd;
    for (auto& m : p->marks) p->ev_pool.push_back(m.second);
    p->marks.clear();
    return SIMRANK_OK;
}

int simrank_shardplan_reset(simrank_shardplan* const* plans, int32_t n_local) {
    int rc = check_group(plans, n_local);
    if (rc) return rc;
    for (int32_t i = 0; i < n_local; ++i) {
        simrank_shardplan* p = plans[i];
        p->cur = 0;
        p->updates = 0;
        if (p->Lm) {
            rc = p->half ? simrank_fill_identity_blocked_h16(p->S[0], p->n, p->Lm, p->rows_pad, p->m_lo, kHalfScale, p->stream)
                         : simrank_fill_identity(p->S[0], p->n, p->Lm, p->ld, p->m_lo, p->stream);
            if (rc) return rc;
        }
    }
    return SIMRANK_OK;
}

int simrank_shardplan_step(simrank_shardplan* const* plans, int32_t n_local, double eps, int32_t exact_count,
                           int64_t* n_changed) {
    int rc = check_group(plans, n_local);
    if (rc) return rc;
    rc = update(plans, n_local, eps, exact_count, 0);
    if (rc) return rc;
    flip(plans, n_local);
    for (int32_t i = 0; i < n_local; ++i) ++plans[i]->updates;
    if (n_changed) {
        unsigned long long c = 0;
        rc = read_count(plans, n_local, 0, &c);
        if (rc) return rc;
        *n_changed = (int64_t)c;
    }
    return SIMRANK_OK;
}

int simrank_shardplan_run(simrank_shardplan* const* plans, int32_t n_local, int32_t iterations, double eps,
                          int32_t* updates_done, int32_t* converged_at) {
    int rc = check_group(plans, n_local);
    if (rc) return rc;
    SR_REQUIRE(iterations >= 0, "iterations < 0");
    rc = simrank_shardplan_reset(plans, n_local);
    if (rc) return rc;
    int32_t conv = -1, done = 0;
    if (iterations > 0 && !(1.0 > eps)) {
        conv = 0;           // loop index 0 compares S_0 = I with the zero matrix: "converged" unless 1 > eps
    } else if (iterations > 0) {
        rc = update(plans, n_local, eps, 0, 1);
        if (rc) return rc;
        for (int32_t k = 1;; ++k) {
            flip(plans, n_local);                    // S[cur] = result of update k
            done = k;
            if (k == iterations) break;              // the reference makes no test after its last update
            // update k + 1 goes out before the count of update k is known only while a rank's update is short (common.h
            // kSpeculateBelow, on the rows a rank computes per leg: n / ranks columns of n rows)
            const bool spec = plans[0]->n / std::max(1, plans[0]->world / 2) < kSpeculateBelow;
            if (spec) {
                rc = update(plans, n_local, eps, 0, (k + 1) & 1);
                if (rc) return rc;
            }
            unsigned long long c = 0;
            rc = read_count(plans, n_local, k & 1, &c);           // the same number on every rank
            if (rc) return rc;
            if (c == 0) {                            // converged at loop index k; a speculative update is dropped
                conv = k;
                break;
            }
            if (!spec) {
                rc = update(plans, n_local, eps, 0, (k + 1) & 1);
                if (rc) return rc;
            }
        }
    }
    for (int32_t i = 0; i < n_local; ++i) {
        SR_HIP(hipStreamSynchronize(plans[i]->stream));
        if (plans[i]->xstream) SR_HIP(hipStreamSynchronize(plans[i]->xstream));
        plans[i]->updates = done;
    }
    if (updates_done) *updates_done = done;
    if (converged_at) *converged_at = conv;
    return SIMRANK_OK;
}

int simrank_shardplan_block_f64(simrank_shardplan* p, double* dst, int64_t ld) {
    SR_REQUIRE(p && (dst || !p->Lm) && ld >= p->Lm, "bad result arguments");
    if (!p->Lm) return SIMRANK_OK;
    float* tmp = nullptr;
    int rc = block_rows_in_callers_order(p, &tmp);
    if (!rc) rc = simrank_download_f64(dst, ld, tmp, p->ld, p->n, p->Lm, p->stream);
    return rc;
}

int simrank_shardplan_columns(const simrank_shardplan* p, int32_t* ids) {
    SR_REQUIRE(p && (ids || !p->Lm), "bad arguments");
    for (int64_t j = 0; j < p->Lm; ++j) ids[j] = p->ord[(size_t)(p->m_lo + j)];
    return SIMRANK_OK;
}

int simrank_shardplan_result_f64(simrank_shardplan* const* plans, int32_t n_local, int32_t root, double* dst, int64_t ld) {
    int rc = check_group(plans, n_local);
    if (rc) return rc;
    simrank_shardplan* p0 = plans[0];
    const int32_t P = p0->world;
    const int64_t n = p0->n;
    SR_REQUIRE(root >= 0 && root < P, "root out of range");
    const bool local = p0->comm->group != nullptr;
    const bool i_am_root = local || p0->rank == root;
    SR_REQUIRE(!i_am_root || (dst && ld >= n), "bad result arguments");
    auto scatter = [&](const std::vector<double>& blockv, int64_t lo, int64_t width) {
        for (int64_t i = 0; i < n; ++i) {
            const double* src = blockv.data() + i * width;
            double* row = dst + i * ld;
            for (int64_t j = 0; j < width; ++j) row[p0->ord[(size_t)(lo + j)]] = src[j];
        }
    };
    if (local) {
        for (int32_t i = 0; i < n_local; ++i) {
            simrank_shardplan* p = plans[i];
            if (!p->Lm) continue;
            std::vector<double> host(size_t(n) * size_t(p->Lm));
            rc = simrank_shardplan_block_f64(p, host.data(), p->Lm);
            if (rc) return rc;
            scatter(host, p->m_lo, p->Lm);
        }
        return SIMRANK_OK;
    }
    Rccl* R = p0->comm->api;
    float* mine = nullptr;
    rc = block_rows_in_callers_order(p0, &mine);
    if (rc) return rc;
    // (the communicator's calls all go to its own stream; the hand-back is not overlapped with anything)
    SR_HIP(hipStreamSynchronize(p0->stream));
    hipStream_t xs = p0->xstream;
    if (!i_am_root) {
        if (p0->Lm) SR_RCCL(R->Send(mine, size_t(n) * size_t(p0->ld), ncclFloat, root, p0->comm->nccl, xs));
        SR_HIP(hipStreamSynchronize(xs));
        return SIMRANK_OK;
    }
    float* scratch = nullptr;
    const int64_t ld_max = pitch(p0->mb, 4);
    hipError_t e = pool_hip_alloc((void**)&scratch, size_t(n) * size_t(ld_max) * 4);
    if (e != hipSuccess) {
        set_error("simrank_shardplan_result_f64: %s", hipGetErrorString(e));
        return SIMRANK_ERR_ALLOC;
    }
    for (int32_t h = 0; h < P && !rc; ++h) {
        int64_t lo, hi;
        part(n, P, h, &lo, &hi);
        const int64_t w = hi - lo, ldh = pitch(w, 4);
        if (!w) continue;
        const float* src = mine;
        if (h != root) {
            ncclResult_t r = R->Recv(scratch, size_t(n) * size_t(ldh), ncclFloat, h, p0->comm->nccl, xs);
            if (r != ncclSuccess || hipStreamSynchronize(xs) != hipSuccess) {
                set_error("ncclRecv failed: %s", R->GetErrorString(r));
                rc = SIMRANK_ERR_HIP;
                break;
            }
            src = scratch;
        }
        std::vector<double> host(size_t(n) * size_t(w));
        rc = simrank_download_f64(host.data(), w, src, ldh, n, w, p0->stream);
        if (!rc) scatter(host, lo, w);
    }
    (void)hipStreamSynchronize(p0->stream);
    (void)pool_free(scratch);
    return rc;
}

int simrank_shardplan_topk(simrank_shardplan* const* plans, int32_t n_local, int32_t root, int32_t k, int32_t exclude_diag,
                           int32_t* idx_host, float* val_host) {
    int rc = check_group(plans, n_local);
    if (rc) return rc;
    simrank_shardplan* p0 = plans[0];
    const int32_t P = p0->world;
    const int64_t n = p0->n;
    SR_REQUIRE(root >= 0 && root < P && k > 0 && k <= 1024, "bad top-k arguments");
    const bool local = p0->comm->group != nullptr;
    const bool i_am_root = local || p0->rank == root;
    SR_REQUIRE(!i_am_root || (idx_host && val_host), "bad top-k arguments");
    // every rank: the k best of ITS columns for every row (rows in the solver's order, ids = the caller's); root: merge
    auto kk_of = [&](int32_t h) { return (int32_t)std::min<int64_t>(k, span(n, P, h)); };
    std::vector<std::vector<int32_t>> cand_idx(P);
    std::vector<std::vector<float>> cand_val(P);
    Rccl* R = local ? nullptr : p0->comm->api;
    for (int32_t i = 0; i < n_local; ++i) {
        simrank_shardplan* p = plans[i];
        const int32_t kk = kk_of(p->rank);
        if (!kk) continue;
        int32_t* ids_dev = nullptr;
        int32_t* idx_dev = nullptr;
        float* val_dev = nullptr;
        hipError_t e = pool_hip_alloc((void**)&ids_dev, size_t(p->Lm) * 4);
        if (e == hipSuccess) e = pool_hip_alloc((void**)&idx_dev, size_t(n) * kk * 4);
        if (e == hipSuccess) e = pool_hip_alloc((void**)&val_dev, size_t(n) * kk * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(ids_dev, p->ord.data() + p->m_lo, size_t(p->Lm) * 4, hipMemcpyHostToDevice, p->stream);
        const float* src = p->S[p->cur];
        if (e == hipSuccess && p->half) {                // fp16-held: an f32 row-major copy in the solver's row order
            const size_t wide = size_t((p->Lm + 31) / 32) * size_t(p->rows_pad) * 32 * 4, rowm = size_t(n) * size_t(p->ld) * 4;
            if (!p->hand[0]) e = pool_hip_alloc((void**)&p->hand[0], wide);
            if (e == hipSuccess && !p->hand[1]) e = pool_hip_alloc((void**)&p->hand[1], rowm);
            if (e == hipSuccess) {
                rc = simrank_widen_blocked_h16(p->S[p->cur], p->rows_pad, p->hand[0], p->rows_pad, n, p->Lm, kHalfScale, p->stream);
                if (!rc) rc = simrank_permute_layout(p->hand[0], 32, p->rows_pad, p->hand[1], p->ld, 0, n, p->Lm, nullptr, nullptr, 4, p->stream);
                src = p->hand[1];
            }
        }
        if (e == hipSuccess && !rc)
            rc = simrank_topk_rows_ids(src, p->ld, n, p->Lm, p->m_lo, ids_dev, kk, exclude_diag, idx_dev, val_dev, p->stream);
        if (e == hipSuccess && !rc) {
            if (i_am_root) {
                cand_idx[p->rank].resize(size_t(n) * kk);
                cand_val[p->rank].resize(size_t(n) * kk);
                e = hipMemcpyAsync(cand_idx[p->rank].data(), idx_dev, size_t(n) * kk * 4, hipMemcpyDeviceToHost, p->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(cand_val[p->rank].data(), val_dev, size_t(n) * kk * 4, hipMemcpyDeviceToHost, p->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
            } else {
                e = hipStreamSynchronize(p->stream);
                ncclResult_t r1 = R->Send(idx_dev, size_t(n) * kk, ncclInt32, root, p->comm->nccl, p->xstream);
                ncclResult_t r2 = r1 == ncclSuccess ? R->Send(val_dev, size_t(n) * kk, ncclFloat, root, p->comm->nccl, p->xstream) : r1;
                if (r2 != ncclSuccess) { set_error("ncclSend failed: %s", R->GetErrorString(r2)); rc = SIMRANK_ERR_HIP; }
                if (e == hipSuccess) e = hipStreamSynchronize(p->xstream);
            }
        } else {
            (void)hipStreamSynchronize(p->stream);
        }
        (void)pool_free(ids_dev); (void)pool_free(idx_dev); (void)pool_free(val_dev);
        if (e != hipSuccess) {
            set_error("simrank_shardplan_topk: %s", hipGetErrorString(e));
            (void)hipGetLastError();
            return SIMRANK_ERR_HIP;
        }
        if (rc) return rc;
    }
    if (!i_am_root) return SIMRANK_OK;
    if (!local) {                                        // the other ranks' candidates
        for (int32_t h = 0; h < P; ++h) {
            const int32_t kk = kk_of(h);
            if (h == root || !kk) continue;
            int32_t* idx_dev = nullptr;
            float* val_dev = nullptr;
            hipError_t e = pool_hip_alloc((void**)&idx_dev, size_t(n) * kk * 4);
            if (e == hipSuccess) e = pool_hip_alloc((void**)&val_dev, size_t(n) * kk * 4);
            if (e == hipSuccess) {
                ncclResult_t r1 = R->Recv(idx_dev, size_t(n) * kk, ncclInt32, h, p0->comm->nccl, p0->xstream);
                ncclResult_t r2 = r1 == ncclSuccess ? R->Recv(val_dev, size_t(n) * kk, ncclFloat, h, p0->comm->nccl, p0->xstream) : r1;
                if (r2 != ncclSuccess) { set_error("ncclRecv failed: %s", R->GetErrorString(r2)); rc = SIMRANK_ERR_HIP; }
                cand_idx[h].resize(size_t(n) * kk);
                cand_val[h].resize(size_t(n) * kk);
                if (!rc) e = hipMemcpyAsync(cand_idx[h].data(), idx_dev, size_t(n) * kk * 4, hipMemcpyDeviceToHost, p0->xstream);
                if (!rc && e == hipSuccess) e = hipMemcpyAsync(cand_val[h].data(), val_dev, size_t(n) * kk * 4, hipMemcpyDeviceToHost, p0->xstream);
                if (e == hipSuccess) e = hipStreamSynchronize(p0->xstream);
            }
            (void)pool_free(idx_dev); (void)pool_free(val_dev);
            if (e != hipSuccess) {
                set_error("simrank_shardplan_topk: %s", hipGetErrorString(e));
                return SIMRANK_ERR_HIP;
            }
            if (rc) return rc;
        }
    }
    // merge: largest first, ties by the lower id; row r of the solver's order is the caller's node ord[r]
    std::vector<std::pair<float, int32_t>> row;
    for (int64_t r = 0; r < n; ++r) {
        row.clear();
        for (int32_t h = 0; h < P; ++h) {
            const int32_t kk = kk_of(h);
            for (int32_t j = 0; j < kk; ++j) {
                const int32_t id = cand_idx[h][size_t(r) * kk + j];
                if (id >= 0) row.emplace_back(cand_val[h][size_t(r) * kk + j], id);
            }
        }
        const size_t take = std::min<size_t>(size_t(k), row.size());
        std::partial_sort(row.begin(), row.begin() + take, row.end(), [](const std::pair<float, int32_t>& a, const std::pair<float, int32_t>& b) {
            return a.first > b.first || (a.first == b.first && a.second < b.second);
        });
        int32_t* io = idx_host + int64_t(p0->ord[(size_t)r]) * k;
        float* vo = val_host + int64_t(p0->ord[(size_t)r]) * k;
        for (int32_t j = 0; j < k; ++j) {
            io[j] = size_t(j) < take ? row[j].second : -1;
            vo[j] = size_t(j) < take ? row[j].first : 0.f;
        }
    }
    return SIMRANK_OK;
}

int simrank_shardplan_info(const simrank_shardplan* p, int64_t* n, int64_t* col_lo, int64_t* col_hi, int32_t* half_form,
                           int32_t* stages, int32_t* updates) {
    SR_REQUIRE(p, "plan is NULL");
    if (n) *n = p->n;
    if (col_lo) *col_lo = p->m_lo;
    if (col_hi) *col_hi = p->m_hi;
    if (half_form) *half_form = p->half_form;
    if (stages) *stages = p->n_stages;
    if (updates) *updates = p->updates;
    return SIMRANK_OK;
}


}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// The two-matrix classes on shards (round 5; SimRank.py:288-302, :410-424, :478-492 with S1 and S2 split by column block):
// a pair of the plans above, one per group, each reading the OTHER group's blocks in its leg 1 —
//     S1 <- E1 * C1 * W12.S2.W12^T (+ lbd1 A1), diag <- 1        exchange 1 (+ 2 in the half form) of group 1
//     S2 <- E2 * C2 * W21.S1.W21^T (+ lbd2 A2), diag <- 1        from the NEW S1 (Gauss-Seidel, :300-302): exchanges of group 2
// — two exchanges per loop body in strict order, the loop ends when NEITHER matrix moved (:289).  Same kernels, chunk
// layouts, node orders (ascending row length per group, dealt to the shards where that group runs its half form) and
// evidence rules (options.strict_reference: Evidence_N1 on both updates, quirk Q2) as the Python driver's Sides and as
// simrank_biplan_* on one GPU.  Asymmetric priors too (round 5): both iterates asymmetric, leg 2 = leg 1's operation again with a second exchange.
// ---------------------------------------------------------------------------------------------------------------------
struct simrank_shardbiplan {
    simrank_shardplan* side[2] = {nullptr, nullptr};
    int32_t broadcast_error = 0;
    int64_t n1 = 0, n2 = 0;
    int32_t updates = 0;
};

namespace simrank {
static int bi_sides(simrank_shardbiplan* const* bps, int32_t n_local, int w, std::vector<simrank_shardplan*>& out) {
    SR_REQUIRE(bps && n_local >= 1, "no plans");
    out.resize((size_t)n_local);
    for (int32_t i = 0; i < n_local; ++i) {
        SR_REQUIRE(bps[i] && bps[i]->side[0] && bps[i]->side[1], "plans[%d] is NULL", i);
        out[(size_t)i] = bps[i]->side[w];
    }
    return check_group(out.data(), n_local);
}

// one loop body on every local pair: both updates queued; group w's count lands in pinned slot `slot` of its plans
static int bi_iteration(simrank_shardbiplan* const* bps, int32_t n_local, double eps, int32_t exact_count, int slot) {
    SR_REQUIRE(!bps[0]->broadcast_error, "operands could not be broadcast together with shapes (%lld,%lld) (%lld,%lld) ",
               (long long)bps[0]->n1, (long long)bps[0]->n1, (long long)bps[0]->n2, (long long)bps[0]->n2);
    std::vector<simrank_shardplan*> s;
    for (int w = 0; w < 2; ++w) {
        int rc = bi_sides(bps, n_local, w, s);
        if (!rc) rc = update(s.data(), n_local, eps, exact_count, slot);
        if (rc) return rc;
        flip(s.data(), n_local);             // (the group-2 update of the same loop body reads the new S1)
    }
    return SIMRANK_OK;
}

static int bi_counts(simrank_shardbiplan* const* bps, int32_t n_local, int slot, unsigned long long* c1, unsigned long long* c2) {
    std::vector<simrank_shardplan*> s;
    int rc = bi_sides(bps, n_local, 0, s);
    if (!rc) rc = read_count(s.data(), n_local, slot, c1);
    if (!rc) rc = bi_sides(bps, n_local, 1, s);
    if (!rc) rc = read_count(s.data(), n_local, slot, c2);
    return rc;
}
}  // namespace simrank

extern "C" {

int simrank_shardbiplan_destroy(simrank_shardbiplan* bp) {
    if (!bp) return SIMRANK_OK;
    simrank_shardplan_destroy(bp->side[0]);
    simrank_shardplan_destroy(bp->side[1]);
    delete bp;
    return SIMRANK_OK;
}

int simrank_shardbiplan_create(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12, const int32_t* col12,
                               const float* rowscale1, const float* rowscale2, const simrank_biplan_options* opt,
                               int32_t leg2_form, int32_t stages, int32_t wire_fp16, simrank_comm* comm, void* stream,
                               simrank_shardbiplan** out) {
    SR_REQUIRE(out, "out is NULL");
    *out = nullptr;
    SR_REQUIRE(opt && comm, "options / communicator missing");
    SR_REQUIRE(leg2_form >= -1 && leg2_form <= 1 && stages >= 0 && stages <= 64, "bad options");
    const int32_t P = comm->world;
    const int64_t ns[2] = {n1, n2};
    // the half form group by group: where a group's size is a multiple of 32 x ranks (and asked for, or 8 ranks on)
    // (a prior of either group that is not symmetric makes BOTH iterates asymmetric: full form, second exchange)
    const bool asym = (opt->apriori1 && opt->ld_apriori1 >= n1 && !prior_symmetric(opt->apriori1, opt->ld_apriori1, n1)) ||
                      (opt->apriori2 && opt->ld_apriori2 >= n2 && !prior_symmetric(opt->apriori2, opt->ld_apriori2, n2));
    SR_REQUIRE(!asym || leg2_form != 1, "a prior that is not symmetric needs leg 2 in its full form");
    bool half[2];
    for (int w = 0; w < 2; ++w) {
        const bool fits = (P > 1 && ns[w] % (32 * int64_t(P)) == 0) || (P == 1 && ns[w] % 32 == 0);
        half[w] = !asym && fits && (leg2_form == 1 || (leg2_form == -1 && P >= 8));
    }
    BiPlanPrep pp;
    int rc = shard_biplan_prepare(n1, n2, nnz, rowptr12, col12, rowscale1, rowscale2, opt, half[0] ? P : 1, half[1] ? P : 1, &pp,
                                  true);
    if (rc) return rc;
    simrank_shardbiplan* bp = new simrank_shardbiplan;
    bp->n1 = n1; bp->n2 = n2;
    auto fail = [&](int code) { simrank_shardbiplan_destroy(bp); return code; };
    const float* priors[2] = {opt->apriori1, opt->apriori2};
    const int64_t lds[2] = {opt->ld_apriori1, opt->ld_apriori2};
    for (int w = 0; w < 2; ++w) {
        PlanPrepView view{&pp.ord[w], &pp.inv[w], &pp.rp[w], &pp.cl[w], &pp.rs[w]};
        SideIn in;
        in.n = ns[w];
        in.k = ns[w ^ 1];
        in.nnz = nnz;
        in.pat = &view;
        in.coef = w == 0 ? opt->c1 : opt->c2;
        in.lbd = w == 0 ? opt->lbd1 : opt->lbd2;
        in.apriori = priors[w];
        in.ld_apriori = lds[w];
        const bool q2 = opt->evidence && opt->strict_reference && w == 1;       // Evidence_N1 on the group-2 update
        in.evidence = (opt->evidence && !q2) ? 1 : 0;
        in.half_form = half[w];
        in.asym = pp.asym;
        in.stages = stages;
        in.wire_fp16 = wire_fp16;
        rc = create_side(in, comm, stream, &bp->side[w]);
        if (rc) return fail(rc);
        simrank_shardplan* p = bp->side[w];
        if (q2 && n1 != n2 && n1 != 1) {
            bp->broadcast_error = 1;             // NumPy raises when the first group-2 update runs (quirk Q2)
        } else if (q2 && p->Lm) {
            hipError_t e = pool_hip_alloc((void**)&p->ev, size_t(p->n) * size_t(p->ld_ev));
            if (e != hipSuccess) { set_error("evidence counts: %s", hipGetErrorString(e)); return fail(SIMRANK_ERR_ALLOC); }
            if (n1 == 1 && n2 != 1) {
                // the 1 x 1 Evidence_N1 broadcasts: the one group-1 node's count (with itself) gates every element
                const int cnt = rowscale1[0] != 0.f ? (int)std::min<int64_t>(255, nnz) : 0;
                e = hipMemsetAsync(p->ev, cnt, size_t(p->n) * size_t(p->ld_ev), p->stream);
                if (e != hipSuccess) { set_error("evidence counts: %s", hipGetErrorString(e)); return fail(SIMRANK_ERR_HIP); }
            } else {
                // n1 = n2: element (i, j) of the group-2 update is multiplied by Evidence_N1[i][j], positions in the caller's
                // order: the counts of the group-1 pattern with its rows taken in THIS group's solver order
                std::vector<int32_t> rp((size_t)n1 + 1, 0), cl((size_t)std::max<int64_t>(1, nnz));
                std::vector<float> rs((size_t)n1);
                for (int64_t r = 0; r < n1; ++r) {
                    const int32_t src = pp.ord[1][(size_t)r];
                    const int32_t b = rowptr12[src], e2 = rowptr12[src + 1];
                    std::copy(col12 + b, col12 + e2, cl.data() + rp[(size_t)r]);
                    std::sort(cl.data() + rp[(size_t)r], cl.data() + rp[(size_t)r] + (e2 - b));
                    rp[(size_t)r + 1] = rp[(size_t)r] + (e2 - b);
                    rs[(size_t)r] = rowscale1[src];
                }
                simrank_graph* g1 = nullptr;
                rc = simrank_graph_create(n1, n2, nnz, rp.data(), cl.data(), rs.data(), &g1);
                if (rc) return fail(rc);
                e = hipMemsetAsync(p->ev, 0, size_t(p->n) * size_t(p->ld_ev), p->stream);
                rc = e == hipSuccess ? simrank_evidence_counts(g1, p->m_lo, p->Lm, p->ev, p->ld_ev, p->stream) : SIMRANK_ERR_HIP;
                (void)hipStreamSynchronize(p->stream);
                simrank_graph_destroy(g1);
                if (rc) return fail(rc);
            }
            int64_t live = 0, total = 1;
            rc = simrank_evidence_live_segments(p->ev, p->ld_ev, 0, p->n, p->Lm, &live, &total, p->stream);
            if (rc) return fail(rc);
            p->restrict_support = 2 * live < total ? 1 : 0;
        }
    }
    bp->side[0]->src = bp->side[1];
    bp->side[1]->src = bp->side[0];
    *out = bp;
    return SIMRANK_OK;
}

int simrank_shardbiplan_side(simrank_shardbiplan* bp, int32_t group, simrank_shardplan** out) {
    SR_REQUIRE(bp && out && (group == 1 || group == 2), "bad arguments");
    *out = bp->side[group - 1];
    return SIMRANK_OK;
}

int simrank_shardbiplan_reset(simrank_shardbiplan* const* bps, int32_t n_local) {
    std::vector<simrank_shardplan*> s;
    for (int w = 0; w < 2; ++w) {
        int rc = bi_sides(bps, n_local, w, s);
        if (!rc) rc = simrank_shardplan_reset(s.data(), n_local);
        if (rc) return rc;
    }
    for (int32_t i = 0; i < n_local; ++i) bps[i]->updates = 0;
    return SIMRANK_OK;
}

int simrank_shardbiplan_step(simrank_shardbiplan* const* bps, int32_t n_local, double eps, int32_t exact_count,
                             int64_t* changed1, int64_t* changed2) {
    SR_REQUIRE(bps && n_local >= 1 && bps[0], "no plans");
    int rc = bi_iteration(bps, n_local, eps, exact_count, 0);
    if (rc) return rc;
    for (int32_t i = 0; i < n_local; ++i) ++bps[i]->updates;
    if (changed1 || changed2) {
        unsigned long long c1 = 0, c2 = 0;
        rc = bi_counts(bps, n_local, 0, &c1, &c2);
        if (rc) return rc;
        if (changed1) *changed1 = (int64_t)c1;
        if (changed2) *changed2 = (int64_t)c2;
    }
    return SIMRANK_OK;
}

int simrank_shardbiplan_run(simrank_shardbiplan* const* bps, int32_t n_local, int32_t iterations, double eps,
                            int32_t* updates_done, int32_t* converged_at) {
    SR_REQUIRE(bps && n_local >= 1 && bps[0], "no plans");
    SR_REQUIRE(iterations >= 0, "iterations < 0");
    int rc = simrank_shardbiplan_reset(bps, n_local);
    if (rc) return rc;
    int32_t conv = -1, done = 0;
    if (iterations > 0 && !(1.0 > eps)) {
        conv = 0;           // loop index 0 compares the identities with zero matrices: "converged" unless 1 > eps
    } else {
        // (the counts of a loop body are read before the next one is queued: two exchanges per body set the pace here,
        // not the host's round trip)
        for (int32_t k = 0; k < iterations; ++k) {
            rc = bi_iteration(bps, n_local, eps, 0, k & 1);
            if (rc) return rc;
            done = k + 1;
            if (done == iterations) break;               // the reference makes no test after its last iteration
            unsigned long long c1 = 0, c2 = 0;
            rc = bi_counts(bps, n_local, k & 1, &c1, &c2);
            if (rc) return rc;
            if (c1 == 0 && c2 == 0) {                    // SimRank.py:289: both groups
                conv = done;
                break;
            }
        }
    }
    std::vector<simrank_shardplan*> s;
    for (int w = 0; w < 2; ++w) {
        rc = bi_sides(bps, n_local, w, s);
        if (rc) return rc;
        for (simrank_shardplan* p : s) {
            SR_HIP(hipStreamSynchronize(p->stream));
            if (p->xstream) SR_HIP(hipStreamSynchronize(p->xstream));
            p->updates = done;
        }
    }
    for (int32_t i = 0; i < n_local; ++i) bps[i]->updates = done;
    if (updates_done) *updates_done = done;
    if (converged_at) *converged_at = conv;
    return SIMRANK_OK;
}

}  // extern "C"
