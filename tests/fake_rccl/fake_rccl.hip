// fake_rccl.hip -- TEST-ONLY stand-in for librccl.so (never shipped, never loaded by default).
//
// Why: RCCL refuses two ranks on one GPU, and the development boxes have one GPU.  The slab exchange behind the C ABI
// (sph_taichi_amd/csrc/sph_comm.hip) had therefore only ever run as ONE rank talking to itself; the ordering it relies
// on BETWEEN ranks had never met a second process.  This library exports the eleven RCCL entry points sph_comm.hip binds
// (same prototypes: it includes the real <rccl/rccl.h>, so the compiler checks them) and moves the bytes between
// PROCESSES that share one GPU.  `SPH_RCCL_LIB=<path to this .so>` makes sph_comm.hip dlopen it instead of librccl.
//
// What it models -- the part of RCCL's behaviour the protocol depends on:
//   * every operation is enqueued on the caller's stream and completes IN STREAM ORDER, without the host waiting:
//     a receive makes the stream wait ON THE DEVICE (a one-lane kernel spinning on a flag in mapped pinned memory, the
//     way RCCL's own kernels spin on their peers' flags) until the bytes are there; the host call returns at once;
//   * the ops of one ncclGroupStart/End progress together (no order between them inside the group);
//   * a send to / receive from oneself inside one group is a device-to-device copy;
//   * ncclAllReduce(sum) adds the contributions in rank order on every rank (bit-identical results everywhere).
// What it does not model: xGMI (bytes travel GPU -> pinned host -> a shared mapping in /tmp -> pinned host -> GPU, so
// bandwidth and latency are meaningless), RCCL's channels / kernels / topology, error recovery, more than 8 ranks.
//
// Mechanics.  ncclGetUniqueId invents a file name; ncclCommInitRank maps that file (sparse: per directed rank pair a
// ring of FAKE_RCCL_SLOTS slots of FAKE_RCCL_SLOT_BYTES bytes) and waits until all ranks have arrived.  Per
// communicator one HELPER THREAD moves bytes between pinned staging buffers and the rings; it never calls HIP.
//   send : stream: D2H copy into a staging buffer, then a kernel sets the buffer's READY flag
//          helper: waits for READY, pushes the buffer chunk by chunk into the ring (waits for free slots)
//   recv : helper: pulls the chunks into a staging buffer, sets its DONE flag
//          stream: a kernel waits for DONE, H2D copy, a kernel marks the buffer free
// A group's chunks go round-robin over its ops, so two ranks that both send more than a ring holds cannot wait for each
// other.  Every wait -- helper or device -- gives up after FAKE_RCCL_TIMEOUT_S (default 60) and latches an error that
// the next API call reports: a protocol bug ends as a failed test, not as a hung GPU box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int MAXW = 8;
constexpr int MAX_SLOTS = 16;
constexpr size_t RED_MAX = 1 << 16;  // bytes per all-reduce (the re-cut histogram of a 400-layer tank is 3.2 KB)
constexpr int NENT = 48;             // staging buffers per communicator

struct ShmHeader {
    std::atomic<uint32_t> arrived;
    uint32_t world, slots;
    uint64_t slot_bytes;
    std::atomic<uint64_t> red_posted[MAXW];  // number of all-reduces this rank has contributed to
    std::atomic<uint64_t> red_done[MAXW];    // ... and has finished reading
    alignas(64) uint8_t red_data[MAXW][RED_MAX];
};
struct ChanHeader {  // one directed pair src -> dst
    alignas(64) std::atomic<uint64_t> head;  // chunks published
    alignas(64) std::atomic<uint64_t> tail;  // chunks consumed
    uint64_t len[MAX_SLOTS];    // bytes of the chunk in a slot
    uint64_t total[MAX_SLOTS];  // bytes of the whole message the chunk belongs to (size mismatches are reported, not guessed)
};
static_assert(std::atomic<uint64_t>::is_always_lock_free, "the rings are shared between processes");

struct Entry {          // a pinned staging buffer and its three flags (in mapped pinned memory, visible to both sides)
    void* host = nullptr;
    size_t cap = 0;
    volatile uint32_t* ready = nullptr;  // stream -> helper: the D2H copy of op `seq` is complete
    volatile uint32_t* done = nullptr;   // helper -> stream: the bytes of op `seq` are in the buffer
    volatile uint32_t* busy = nullptr;   // 1 from acquisition until the last reader is through (cleared by the helper or by a kernel)
};

enum OpKind { OP_SEND, OP_RECV, OP_ALLREDUCE };
struct Op {
    OpKind kind;
    int peer;
    size_t bytes;
    int entry;
    uint32_t seq;
    ncclDataType_t dtype;
    size_t count;
    // only while the group is being assembled
    const void* src;
    void* dst;
    hipStream_t stream;
};
struct Group { std::vector<Op> ops; };

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct ncclComm {
    int rank = 0, world = 1, device = 0;
    uint8_t* shm = nullptr;
    size_t shm_bytes = 0;
    uint32_t slots = 4;
    uint64_t slot_bytes = 1 << 18;
    double timeout_s = 60.0;
    Entry ent[NENT];
    uint32_t* cells = nullptr;  // pinned, mapped: 3 per entry + the device-side error latch
    std::vector<void*> graveyard;
    std::thread helper;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Group> q;
    bool stop = false;
    std::atomic<int> err{0};
    char errmsg[256] = "";
    uint32_t seq = 0;
    uint64_t red_epoch = 0;  // helper-side count of all-reduces

    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(shm); }
    ChanHeader* chan(int src, int dst) const {
        return reinterpret_cast<ChanHeader*>(shm + sizeof(ShmHeader)) + (src * world + dst);
    }
    uint8_t* chan_data(int src, int dst, uint64_t slot) const {
        uint8_t* base = shm + sizeof(ShmHeader) + sizeof(ChanHeader) * (size_t)world * world;
        return base + ((size_t)(src * world + dst) * slots + slot) * slot_bytes;
    }
    void fail(const char* fmt, ...) {
        if (err.exchange(1) != 0) return;  // keep the first message
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(errmsg, sizeof(errmsg), fmt, ap);
        va_end(ap);
        fprintf(stderr, "[fake_rccl rank %d/%d] %s\n", rank, world, errmsg);
        fflush(stderr);
    }
};

namespace {

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local ncclComm* g_group_comm = nullptr;
thread_local char g_last[256] = "";

__global__ void k_set(uint32_t* cell, uint32_t v) { __hip_atomic_store(cell, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// the device-side wait of a receive: one lane polls a flag in host memory, sleeping between polls; gives up after
// `ticks` of the 100 MHz wall clock and latches the error (the stream then carries on with undefined bytes -- the host
// reports the error at its next call)
__global__ void k_wait(const uint32_t* cell, uint32_t v, uint32_t* err, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(cell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != v) {
        __builtin_amdgcn_s_sleep(100);
        if (wall_clock64() - t0 > ticks) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

template <class F>
bool spin_until(ncclComm* c, F&& cond, const char* what) {
    const double t0 = now_s();
    unsigned n = 0;
    while (!cond()) {
        if (c->err.load()) return false;
        if ((++n & 1023u) == 0) {
            if (now_s() - t0 > c->timeout_s) { c->fail("timeout (%.0f s) waiting for %s", c->timeout_s, what); return false; }
            std::this_thread::yield();
        }
    }
    return true;
}

// ---- helper thread -----------------------------------------------------------------------------------------------------
bool push_chunk(ncclComm* c, const Op& op, size_t k) {
    ChanHeader* ch = c->chan(c->rank, op.peer);
    const uint64_t h = ch->head.load(std::memory_order_relaxed);
    char what[96];
    snprintf(what, sizeof(what), "a free slot towards rank %d (is that rank receiving?)", op.peer);
    if (!spin_until(c, [&] { return h - ch->tail.load(std::memory_order_acquire) < c->slots; }, what)) return false;
    const size_t off = k * c->slot_bytes, len = op.bytes - off < c->slot_bytes ? op.bytes - off : c->slot_bytes;
    const uint64_t s = h % c->slots;
    memcpy(c->chan_data(c->rank, op.peer, s), (const uint8_t*)c->ent[op.entry].host + off, len);
    ch->len[s] = len;
    ch->total[s] = op.bytes;
    ch->head.store(h + 1, std::memory_order_release);
    return true;
}

bool pull_chunk(ncclComm* c, const Op& op, size_t k) {
    ChanHeader* ch = c->chan(op.peer, c->rank);
    const uint64_t t = ch->tail.load(std::memory_order_relaxed);
    char what[96];
    snprintf(what, sizeof(what), "a message from rank %d (is that rank sending?)", op.peer);
    if (!spin_until(c, [&] { return ch->head.load(std::memory_order_acquire) > t; }, what)) return false;
    const size_t off = k * c->slot_bytes, len = op.bytes - off < c->slot_bytes ? op.bytes - off : c->slot_bytes;
    const uint64_t s = t % c->slots;
    if (ch->len[s] != len || ch->total[s] != op.bytes) {
        c->fail("recv of %zu bytes from rank %d meets a message of %llu bytes (chunk %llu): the two sides disagree on the size",
                op.bytes, op.peer, (unsigned long long)ch->total[s], (unsigned long long)ch->len[s]);
        return false;
    }
    memcpy((uint8_t*)c->ent[op.entry].host + off, c->chan_data(op.peer, c->rank, s), len);
    ch->tail.store(t + 1, std::memory_order_release);
    return true;
}

template <class T>
void add_into(void* acc, const void* x, size_t n) {
    T* a = (T*)acc;
    const T* b = (const T*)x;
    for (size_t i = 0; i < n; ++i) a[i] += b[i];
}

bool do_allreduce(ncclComm* c, const Op& op) {
    ShmHeader* h = c->hdr();
    const uint64_t e = c->red_epoch++;
    Entry& en = c->ent[op.entry];
    if (!spin_until(c, [&] { return *en.ready == op.seq; }, "the all-reduce input's copy to the host")) return false;
    // everybody must have finished READING the previous epoch before anybody overwrites its contribution
    if (!spin_until(c, [&] { for (int r = 0; r < c->world; ++r) if (h->red_done[r].load(std::memory_order_acquire) < e) return false; return true; },
                    "the other ranks to finish the previous all-reduce"))
        return false;
    memcpy(h->red_data[c->rank], en.host, op.bytes);
    h->red_posted[c->rank].store(e + 1, std::memory_order_release);
    if (!spin_until(c, [&] { for (int r = 0; r < c->world; ++r) if (h->red_posted[r].load(std::memory_order_acquire) < e + 1) return false; return true; },
                    "the other ranks' all-reduce contributions (did every rank call it?)"))
        return false;
    memset(en.host, 0, op.bytes);
    for (int r = 0; r < c->world; ++r) {  // rank order on every rank: identical sums everywhere
        switch (op.dtype) {
            case ncclFloat64: add_into<double>(en.host, h->red_data[r], op.count); break;
            case ncclFloat32: add_into<float>(en.host, h->red_data[r], op.count); break;
            case ncclInt64: case ncclUint64: add_into<uint64_t>(en.host, h->red_data[r], op.count); break;
            case ncclInt32: case ncclUint32: add_into<uint32_t>(en.host, h->red_data[r], op.count); break;
            default: c->fail("all-reduce of data type %d is not modelled", (int)op.dtype); return false;
        }
    }
    h->red_done[c->rank].store(e + 1, std::memory_order_release);
    std::atomic_thread_fence(std::memory_order_release);
    *en.done = op.seq;
    return true;
}

void run_group(ncclComm* c, Group& g) {
    for (Op& op : g.ops)
        if (op.kind == OP_ALLREDUCE && !do_allreduce(c, op)) return;
    std::vector<bool> started(g.ops.size(), false);
    for (size_t k = 0;; ++k) {
        bool any = false;
        for (size_t i = 0; i < g.ops.size(); ++i) {
            Op& op = g.ops[i];
            if (op.kind != OP_SEND || k * c->slot_bytes >= op.bytes) continue;
            any = true;
            if (!started[i]) {
                Entry& en = c->ent[op.entry];
                if (!spin_until(c, [&] { return *en.ready == op.seq; }, "a send buffer's copy to the host (is the stream stuck behind an event?)")) return;
                started[i] = true;
            }
            if (!push_chunk(c, op, k)) return;
            if ((k + 1) * c->slot_bytes >= op.bytes) *c->ent[op.entry].busy = 0;  // staging buffer free again
        }
        for (Op& op : g.ops) {
            if (op.kind != OP_RECV || k * c->slot_bytes >= op.bytes) continue;
            any = true;
            if (!pull_chunk(c, op, k)) return;
            if ((k + 1) * c->slot_bytes >= op.bytes) {
                std::atomic_thread_fence(std::memory_order_release);
                *c->ent[op.entry].done = op.seq;  // the waiting kernel lets the stream continue
            }
        }
        if (!any) break;
    }
}

void helper_main(ncclComm* c) {
    for (;;) {
        Group g;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->stop || !c->q.empty(); });
            if (c->q.empty()) return;  // (stop: drain first)
            g = std::move(c->q.front());
            c->q.pop_front();
        }
        if (!c->err.load()) run_group(c, g);
    }
}

// ---- API thread ----------------------------------------------------------------------------------------------------------
int acquire_entry(ncclComm* c, size_t bytes) {
    const double t0 = now_s();
    for (;;) {
        int best = -1;
        for (int i = 0; i < NENT; ++i) {
            if (*c->ent[i].busy) continue;
            if (c->ent[i].cap >= bytes) { best = i; break; }
            if (best < 0) best = i;
        }
        if (best >= 0) {
            Entry& e = c->ent[best];
            if (e.cap < bytes) {  // grow: the old buffer is parked until ncclCommDestroy (hipHostFree may synchronise the device)
                if (e.host) c->graveyard.push_back(e.host);
                size_t cap = 4096;
                while (cap < bytes) cap <<= 1;
                if (hipHostMalloc(&e.host, cap, hipHostMallocCoherent) != hipSuccess) { e.host = nullptr; e.cap = 0; c->fail("hipHostMalloc(%zu) failed", cap); return -1; }
                e.cap = cap;
            }
            *e.busy = 1;
            return best;
        }
        if (c->err.load()) return -1;
        if (now_s() - t0 > c->timeout_s) { c->fail("timeout waiting for a free staging buffer (%d in flight)", NENT); return -1; }
        std::this_thread::yield();
    }
}

uint32_t* dev_cell(volatile uint32_t* host_cell) {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, (void*)host_cell, 0) != hipSuccess) return (uint32_t*)host_cell;
    return (uint32_t*)d;
}

ncclResult_t check(ncclComm* c) {
    if (c->cells && c->cells[3 * NENT]) c->fail("a device-side wait timed out (%.0f s): the bytes of a receive / all-reduce never arrived", c->timeout_s);
    if (c->err.load()) { snprintf(g_last, sizeof(g_last), "%s", c->errmsg); return ncclInternalError; }
    return ncclSuccess;
}

#define HIPCHK(c, expr)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) { (c)->fail("%s: %s", #expr, hipGetErrorString(e__)); return ncclUnhandledCudaError; } \
    } while (0)

ncclResult_t submit(ncclComm* c, std::vector<Op>& ops) {
    ncclResult_t rc = check(c);
    if (rc != ncclSuccess) return rc;
    const unsigned long long ticks = (unsigned long long)(c->timeout_s * 1e8);
    uint32_t* err_cell = dev_cell(c->cells + 3 * NENT);
    // a send to oneself meets the receive from oneself of the same group: device-to-device
    std::vector<Op*> self_s, self_r;
    for (Op& op : ops) {
        if (op.kind == OP_SEND && op.peer == c->rank) self_s.push_back(&op);
        if (op.kind == OP_RECV && op.peer == c->rank) self_r.push_back(&op);
    }
    if (self_s.size() != self_r.size()) { c->fail("%zu sends to self but %zu receives from self in one group", self_s.size(), self_r.size()); return ncclInvalidArgument; }
    for (size_t i = 0; i < self_s.size(); ++i) {
        if (self_s[i]->bytes != self_r[i]->bytes) { c->fail("send to self of %zu bytes meets a receive of %zu", self_s[i]->bytes, self_r[i]->bytes); return ncclInvalidArgument; }
        if (self_s[i]->bytes) HIPCHK(c, hipMemcpyAsync(self_r[i]->dst, self_s[i]->src, self_s[i]->bytes, hipMemcpyDeviceToDevice, self_r[i]->stream));
    }
    Group g;
    // the sends' copies first, then the waits of the receives: nothing in a group waits for anything else in it
    const OpKind order[3] = {OP_ALLREDUCE, OP_SEND, OP_RECV};
    for (const OpKind kind : order)
        for (Op& op : ops) {
            if (op.kind != kind) continue;
            if (kind != OP_ALLREDUCE && (op.peer == c->rank || op.bytes == 0)) continue;
            op.entry = acquire_entry(c, op.bytes);
            if (op.entry < 0) return ncclInternalError;
            op.seq = ++c->seq;
            Entry& e = c->ent[op.entry];
            if (op.kind == OP_SEND || op.kind == OP_ALLREDUCE) {
                HIPCHK(c, hipMemcpyAsync(e.host, op.src, op.bytes, hipMemcpyDeviceToHost, op.stream));
                hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, op.stream, dev_cell(e.ready), op.seq);
                HIPCHK(c, hipGetLastError());
            }
            if (op.kind == OP_RECV || op.kind == OP_ALLREDUCE) {
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, op.stream, dev_cell(e.done), op.seq, err_cell, ticks);
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipMemcpyAsync(op.dst, e.host, op.bytes, hipMemcpyHostToDevice, op.stream));
                hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, op.stream, dev_cell(e.busy), 0u);
                HIPCHK(c, hipGetLastError());
            }
            g.ops.push_back(op);
        }
    if (!g.ops.empty()) {
        std::lock_guard<std::mutex> lk(c->mu);
        c->q.push_back(std::move(g));
        c->cv.notify_one();
    }
    return ncclSuccess;
}

ncclResult_t add_op(ncclComm* c, Op op) {
    if (!c) return ncclInvalidArgument;
    if (g_depth > 0) {
        if (g_group_comm && g_group_comm != c) { c->fail("one communicator per group in this stand-in"); return ncclInvalidArgument; }
        g_group_comm = c;
        g_ops.push_back(op);
        return ncclSuccess;
    }
    std::vector<Op> one{op};
    return submit(c, one);
}

}  // namespace

extern "C" {

const char* ncclGetErrorString(ncclResult_t r) {
    if (r != ncclSuccess && g_last[0]) return g_last;
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclSystemError: return "system error (fake_rccl)";
        case ncclInternalError: return "internal error (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        default: return "error (fake_rccl)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned r = 0;
    if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&r, sizeof(r), 1, f) != 1) r = 0; fclose(f); }
    const char* dir = getenv("FAKE_RCCL_DIR");
    snprintf(id->internal, sizeof(id->internal), "%s/fake_rccl_%d_%08x", dir ? dir : "/tmp", (int)getpid(), r);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
    if (!c || !count) return ncclInvalidArgument;
    *count = c->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int* rank) {
    if (!c || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
        c->cv.notify_one();
    }
    if (c->helper.joinable()) c->helper.join();
    for (int i = 0; i < NENT; ++i)
        if (c->ent[i].host) (void)hipHostFree(c->ent[i].host);
    for (void* p : c->graveyard) (void)hipHostFree(p);
    if (c->cells) (void)hipHostFree(c->cells);
    if (c->shm) munmap(c->shm, c->shm_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAXW || rank < 0 || rank >= nranks) {
        snprintf(g_last, sizeof(g_last), "fake_rccl: %d ranks (this stand-in models 1..%d), rank %d", nranks, MAXW, rank);
        return ncclInvalidArgument;
    }
    ncclComm* c = new ncclComm();
    c->rank = rank; c->world = nranks;
    (void)hipGetDevice(&c->device);
    if (const char* e = getenv("FAKE_RCCL_SLOT_BYTES")) c->slot_bytes = strtoull(e, nullptr, 10);
    if (const char* e = getenv("FAKE_RCCL_SLOTS")) c->slots = (uint32_t)atoi(e);
    if (const char* e = getenv("FAKE_RCCL_TIMEOUT_S")) c->timeout_s = atof(e);
    if (c->slot_bytes < 16) c->slot_bytes = 16;
    c->slot_bytes = (c->slot_bytes + 15) & ~(uint64_t)15;
    if (c->slots < 2) c->slots = 2;
    if (c->slots > MAX_SLOTS) c->slots = MAX_SLOTS;
    if (c->timeout_s < 1.0) c->timeout_s = 1.0;
    id.internal[sizeof(id.internal) - 1] = 0;
    c->shm_bytes = sizeof(ShmHeader) + sizeof(ChanHeader) * (size_t)nranks * nranks + (size_t)nranks * nranks * c->slots * c->slot_bytes;
    const int fd = open(id.internal, O_RDWR | O_CREAT, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) {
        snprintf(g_last, sizeof(g_last), "fake_rccl: cannot create %s (%zu bytes): %s", id.internal, c->shm_bytes, strerror(errno));
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void* p = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { snprintf(g_last, sizeof(g_last), "fake_rccl: mmap failed: %s", strerror(errno)); delete c; return ncclSystemError; }
    c->shm = (uint8_t*)p;  // (a fresh file is all zero: every counter starts at 0)
    ShmHeader* h = c->hdr();
    if (rank == 0) { h->world = (uint32_t)nranks; h->slots = c->slots; h->slot_bytes = c->slot_bytes; }
    h->arrived.fetch_add(1, std::memory_order_acq_rel);
    const bool all = spin_until(c, [&] { return h->arrived.load(std::memory_order_acquire) >= (uint32_t)nranks; }, "the other ranks in ncclCommInitRank");
    if (rank == 0) unlink(id.internal);  // the mappings keep the file alive; nothing is left behind in /tmp
    if (!all || h->world != (uint32_t)nranks || h->slots != c->slots || h->slot_bytes != c->slot_bytes) {
        snprintf(g_last, sizeof(g_last), "fake_rccl: %s", all ? "the ranks disagree on the world size or the FAKE_RCCL_* ring settings" : c->errmsg);
        munmap(c->shm, c->shm_bytes);
        c->shm = nullptr;
        delete c;
        return ncclInternalError;
    }
    if (hipHostMalloc((void**)&c->cells, (3 * NENT + 1) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        snprintf(g_last, sizeof(g_last), "fake_rccl: hipHostMalloc of the flag cells failed");
        munmap(c->shm, c->shm_bytes);
        delete c;
        return ncclUnhandledCudaError;
    }
    memset(c->cells, 0, (3 * NENT + 1) * sizeof(uint32_t));
    for (int i = 0; i < NENT; ++i) {
        c->ent[i].ready = c->cells + 3 * i;
        c->ent[i].done = c->cells + 3 * i + 1;
        c->ent[i].busy = c->cells + 3 * i + 2;
    }
    c->helper = std::thread(helper_main, c);
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) { snprintf(g_last, sizeof(g_last), "fake_rccl: ncclGroupEnd without ncclGroupStart"); return ncclInvalidArgument; }
    if (--g_depth > 0) return ncclSuccess;
    ncclComm* c = g_group_comm;
    g_group_comm = nullptr;
    std::vector<Op> ops;
    ops.swap(g_ops);
    if (!c || ops.empty()) return ncclSuccess;
    return submit(c, ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (!c || peer < 0 || peer >= c->world || dtype_size(t) == 0 || (count > 0 && !buf)) return ncclInvalidArgument;
    Op op{};
    op.kind = OP_SEND; op.peer = peer; op.bytes = count * dtype_size(t); op.src = buf; op.stream = s; op.dtype = t; op.count = count;
    return add_op(c, op);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (!c || peer < 0 || peer >= c->world || dtype_size(t) == 0 || (count > 0 && !buf)) return ncclInvalidArgument;
    Op op{};
    op.kind = OP_RECV; op.peer = peer; op.bytes = count * dtype_size(t); op.dst = buf; op.stream = s; op.dtype = t; op.count = count;
    return add_op(c, op);
}

ncclResult_t ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t red, ncclComm_t c, hipStream_t s) {
    if (!c || !sendbuf || !recvbuf || count == 0 || dtype_size(t) == 0) return ncclInvalidArgument;
    if (red != ncclSum) { c->fail("only ncclSum is modelled"); return ncclInvalidArgument; }
    if (count * dtype_size(t) > RED_MAX) { c->fail("all-reduce of %zu bytes exceeds the stand-in's %zu", count * dtype_size(t), RED_MAX); return ncclInvalidArgument; }
    Op op{};
    op.kind = OP_ALLREDUCE; op.peer = -1; op.bytes = count * dtype_size(t); op.src = sendbuf; op.dst = recvbuf; op.stream = s; op.dtype = t; op.count = count;
    return add_op(c, op);
}

}  // extern "C"
