// sph_comm.hip -- the slab exchange of SURVEY 8(e) behind the C ABI: RCCL point-to-point over xGMI, enqueued on
// the device.  The reference has no multi-GPU path; these entry points are what a host in any language
// binds to run one context per GPU: sph_comm_create (ncclCommInitRank from a unique id the host distributes by
// whatever means it has -- torch.distributed, MPI, a file), sph_slab_announce / sph_slab_incoming (the record
// counts of the next exchange, sent a whole sweep phase before the payload exists), sph_slab_exchange
// (ncclGroupStart; ncclSend / ncclRecv to the <= 2 x-neighbours; ncclGroupEnd on a communication stream that waits
// for the halo packers' event -- the host is not woken -- and that the context's stream waits for before the next
// sort), sph_comm_swap (fixed-size band refresh of the DFSPH sweeps) and sph_comm_all_reduce (16 shape-matching
// sums per body, the re-cut histogram, the conservation guard).
//
// librccl is opened with dlopen at the first sph_comm_* call: a single-GPU deployment of libsph_hip.so has no
// link-time dependency on it.
#include "sph_internal.h"

#include <dlfcn.h>

// The few RCCL declarations this file needs, spelled out: the library is opened with dlopen, and a single-GPU build of
// libsph_hip.so should not need the RCCL headers either.  Values as in <rccl/rccl.h> (checked against it below whenever
// that header is on the include path).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == 128 && ncclSuccess == 0 && ncclUint8 == 1 && ncclInt32 == 2 && ncclInt64 == 4 &&
              ncclFloat64 == 8 && ncclSum == 0, "the local RCCL declarations below no longer match <rccl/rccl.h>");
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclInt32 = 2, ncclInt64 = 4, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

struct RcclApi {
    void* handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
};

static RcclApi g_rccl = {};
static thread_local char g_comm_err[256] = "";

static int rccl_load() {
    if (g_rccl.handle) return 0;
    // SPH_RCCL_LIB (tests): open THIS library instead -- tests/fake_rccl/libfake_rccl.so, a stand-in that lets several
    // processes sharing one GPU run the exchange (RCCL itself refuses two ranks on a device).  Never a silent fallback:
    // if the variable is set and the library does not load, that is the error.
    const char* forced = getenv("SPH_RCCL_LIB");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    if (forced && forced[0]) h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    else
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) {
        snprintf(g_comm_err, sizeof(g_comm_err), "%s not found (%s)", forced && forced[0] ? forced : "librccl", dlerror());
        return SPH_E_STATE;
    }
    RcclApi a = {};
    a.handle = h;
#define SPH_SYM(field, name) \
    *(void**)(&a.field) = dlsym(h, name); \
    if (!a.field) { snprintf(g_comm_err, sizeof(g_comm_err), "librccl lacks %s", name); dlclose(h); return SPH_E_STATE; }
    SPH_SYM(GetUniqueId, "ncclGetUniqueId")
    SPH_SYM(CommInitRank, "ncclCommInitRank")
    SPH_SYM(CommDestroy, "ncclCommDestroy")
    SPH_SYM(GroupStart, "ncclGroupStart")
    SPH_SYM(GroupEnd, "ncclGroupEnd")
    SPH_SYM(Send, "ncclSend")
    SPH_SYM(Recv, "ncclRecv")
    SPH_SYM(AllReduce, "ncclAllReduce")
    SPH_SYM(GetErrorString, "ncclGetErrorString")
    SPH_SYM(CommCount, "ncclCommCount")
    SPH_SYM(CommUserRank, "ncclCommUserRank")
#undef SPH_SYM
    g_rccl = a;
    return 0;
}

struct SphComm {
    SphContext* ctx;
    ncclComm_t comm;
    int rank, world;
    int device;             // (kept here: sph_comm_destroy may run after the context is gone)
    hipStream_t stream;     // communication stream (beside the context's main and side streams)
    hipEvent_t ev_in;       // main stream -> communication stream
    hipEvent_t ev_done;     // communication stream -> main stream
    hipEvent_t ev_cnt;      // the announced counts have reached pinned memory
    hipEvent_t ev_t0, ev_t1;  // timing of the last payload exchange (halo bucket)
    bool timed_open;
    int* d_cnt;             // [4] device: out-left, out-right, in-left, in-right
    int* h_cnt;             // [4] pinned: in-left, in-right, out-left, out-right
    bool announced;
    bool lonely;            // the pending announcement had no neighbour: no message, no event
    int ann_out[2];         // what was announced (must equal what sph_slab_exchange then sends)
    double halo_ms;         // accumulated payload-exchange time on the communication stream
    long exchanges;
};

#define SPH_NCCL(c, expr)                                                                        \
    do {                                                                                         \
        ncclResult_t r__ = (expr);                                                               \
        if (r__ != ncclSuccess) {                                                                \
            snprintf((c)->err, sizeof((c)->err), "%s: %s", #expr, g_rccl.GetErrorString(r__));    \
            return SPH_E_STATE;                                                                  \
        }                                                                                        \
    } while (0)
// inside ncclGroupStart .. ncclGroupEnd: a failing call must not leave the group open
#define SPH_NCCL_G(c, expr)                                                                      \
    do {                                                                                         \
        ncclResult_t r__ = (expr);                                                               \
        if (r__ != ncclSuccess) {                                                                \
            snprintf((c)->err, sizeof((c)->err), "%s: %s", #expr, g_rccl.GetErrorString(r__));    \
            (void)g_rccl.GroupEnd();                                                             \
            return SPH_E_STATE;                                                                  \
        }                                                                                        \
    } while (0)

extern "C" {

const char* sph_comm_last_error(void) { return g_comm_err; }

int32_t sph_comm_available(void) { return rccl_load(); }

int32_t sph_comm_unique_id(uint8_t* out128) {
    if (!out128) return SPH_E_INVALID;
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the unique id around as 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) {
        snprintf(g_comm_err, sizeof(g_comm_err), "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
        return SPH_E_STATE;
    }
    memcpy(out128, &id, 128);
    return 0;
}

int32_t sph_comm_destroy(SphComm* m) {
    if (!m) return 0;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm) (void)g_rccl.CommDestroy(m->comm);
    for (hipEvent_t e : {m->ev_in, m->ev_done, m->ev_cnt, m->ev_t0, m->ev_t1})
        if (e) (void)hipEventDestroy(e);
    if (m->d_cnt) (void)hipFree(m->d_cnt);
    if (m->h_cnt) (void)hipHostFree(m->h_cnt);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return 0;
}

int32_t sph_comm_create(SphContext* c, const uint8_t* id128, int32_t rank, int32_t world, SphComm** out) {
    if (!c || !id128 || !out || world < 1 || rank < 0 || rank >= world) return SPH_E_INVALID;
    *out = nullptr;
    int rc = rccl_load();
    if (rc) return sph_fail(c, rc, g_comm_err);
    SPH_HIP(c, hipSetDevice(c->device));
    SphComm* m = new SphComm();
    memset(m, 0, sizeof(*m));
    m->ctx = c; m->rank = rank; m->world = world; m->device = c->device;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = g_rccl.CommInitRank(&m->comm, world, id, rank);
    if (r != ncclSuccess) {
        snprintf(c->err, sizeof(c->err), "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
        m->comm = nullptr;
        sph_comm_destroy(m);
        return SPH_E_STATE;
    }
    bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&m->ev_cnt, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreate(&m->ev_t0) == hipSuccess && hipEventCreate(&m->ev_t1) == hipSuccess;
    ok = ok && hipMalloc((void**)&m->d_cnt, 4 * sizeof(int)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&m->h_cnt, 4 * sizeof(int), hipHostMallocMapped) == hipSuccess;
    if (!ok) {
        sph_comm_destroy(m);
        return sph_fail(c, SPH_E_NOMEM, "sph_comm_create: stream / event / buffer allocation failed");
    }
    *out = m;
    return 0;
}

static bool peer_ok(const SphComm* m, int p) { return p >= -1 && p < m->world; }

// close the timing bracket of the previous payload exchange (its events have long completed)
static void harvest_halo(SphComm* m) {
    if (!m->timed_open) return;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, m->ev_t0, m->ev_t1) == hipSuccess) { m->halo_ms += ms; m->exchanges += 1; m->timed_open = false; }
}

int32_t sph_slab_announce(SphContext* c, SphComm* m, int32_t left, int32_t right, int32_t n_to_left, int32_t n_to_right) {
    if (!c || !m || m->ctx != c || !peer_ok(m, left) || !peer_ok(m, right) || n_to_left < 0 || n_to_right < 0) return SPH_E_INVALID;
    SPH_HIP(c, hipSetDevice(c->device));
    if (m->announced) return sph_fail(c, SPH_E_STATE, "sph_slab_announce: the previous announcement was not consumed by sph_slab_exchange");
    m->ann_out[0] = n_to_left; m->ann_out[1] = n_to_right;
    if (left < 0 && right < 0) {  // a rank without neighbours (world = 1): nothing to send, nothing to wait for
        m->h_cnt[0] = m->h_cnt[1] = 0;
        m->announced = true;
        m->lonely = true;
        return 0;
    }
    m->lonely = false;
    m->h_cnt[2] = n_to_left; m->h_cnt[3] = n_to_right;  // (pinned; the previous announcement's copy was waited for by its exchange)
    SPH_HIP(c, hipMemcpyAsync(m->d_cnt, m->h_cnt + 2, 2 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    SPH_HIP(c, hipMemsetAsync(m->d_cnt + 2, 0, 2 * sizeof(int), m->stream));
    SPH_NCCL(c, g_rccl.GroupStart());
    if (left >= 0) {
        SPH_NCCL_G(c, g_rccl.Send(m->d_cnt + 0, 1, ncclInt32, left, m->comm, m->stream));
        SPH_NCCL_G(c, g_rccl.Recv(m->d_cnt + 2, 1, ncclInt32, left, m->comm, m->stream));
    }
    if (right >= 0) {
        SPH_NCCL_G(c, g_rccl.Send(m->d_cnt + 1, 1, ncclInt32, right, m->comm, m->stream));
        SPH_NCCL_G(c, g_rccl.Recv(m->d_cnt + 3, 1, ncclInt32, right, m->comm, m->stream));
    }
    SPH_NCCL(c, g_rccl.GroupEnd());
    SPH_HIP(c, hipMemcpyAsync(m->h_cnt, m->d_cnt + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    SPH_HIP(c, hipEventRecord(m->ev_cnt, m->stream));
    m->announced = true;
    return 0;
}

int32_t sph_slab_incoming(SphContext* c, SphComm* m, int32_t* n_from_left, int32_t* n_from_right) {
    if (!c || !m || m->ctx != c || !n_from_left || !n_from_right) return SPH_E_INVALID;
    SPH_HIP(c, hipSetDevice(c->device));
    if (!m->announced) return sph_fail(c, SPH_E_STATE, "sph_slab_incoming before sph_slab_announce");
    if (!m->lonely) SPH_HIP(c, hipEventSynchronize(m->ev_cnt));  // posted a sweep phase ago: normally complete
    *n_from_left = m->h_cnt[0];
    *n_from_right = m->h_cnt[1];
    return 0;
}

int32_t sph_slab_exchange(SphContext* c, SphComm* m, int32_t left, int32_t right, const void* send_left, int32_t n_to_left,
                          const void* send_right, int32_t n_to_right, void* recv_left, int32_t n_from_left, void* recv_right,
                          int32_t n_from_right, int32_t after_packers) {
    if (!c || !m || m->ctx != c || !peer_ok(m, left) || !peer_ok(m, right)) return SPH_E_INVALID;
    SPH_HIP(c, hipSetDevice(c->device));
    if (!m->announced) return sph_fail(c, SPH_E_STATE, "sph_slab_exchange before sph_slab_announce");
    if (n_to_left != m->ann_out[0] || n_to_right != m->ann_out[1])
        return sph_fail(c, SPH_E_STATE, "sph_slab_exchange: the counts sent differ from the counts announced");
    if (left < 0 && right < 0) {
        if (!m->lonely) return sph_fail(c, SPH_E_STATE, "sph_slab_exchange: neighbours differ from the announcement's");
        m->announced = false;
        return 0;
    }
    if (m->lonely) return sph_fail(c, SPH_E_STATE, "sph_slab_exchange: neighbours differ from the announcement's");
    SPH_HIP(c, hipEventSynchronize(m->ev_cnt));
    if ((left >= 0 && n_from_left != m->h_cnt[0]) || (right >= 0 && n_from_right != m->h_cnt[1]))
        return sph_fail(c, SPH_E_INVALID, "sph_slab_exchange: receive counts differ from sph_slab_incoming");
    if ((left >= 0 && ((n_to_left > 0 && !send_left) || (n_from_left > 0 && !recv_left))) ||
        (right >= 0 && ((n_to_right > 0 && !send_right) || (n_from_right > 0 && !recv_right))))
        return sph_fail(c, SPH_E_INVALID, "sph_slab_exchange: null buffer");
    harvest_halo(m);
    // The payload exists once the halo packers of sph_slab_forces have run (their event) -- the interior force sweep
    // on the main stream may still be running -- or, for the synchronous packers, now.
    if (after_packers) {
        SPH_HIP(c, hipStreamWaitEvent(m->stream, c->ev_pack, 0));
    } else {
        SPH_HIP(c, hipEventRecord(m->ev_in, c->stream));
        SPH_HIP(c, hipStreamWaitEvent(m->stream, m->ev_in, 0));
    }
    const size_t rec = 48;  // xm + vf + aux records of a packed range
    SPH_HIP(c, hipEventRecord(m->ev_t0, m->stream));
    SPH_NCCL(c, g_rccl.GroupStart());
    if (left >= 0) {
        if (n_to_left > 0) SPH_NCCL_G(c, g_rccl.Send(send_left, (size_t)n_to_left * rec, ncclUint8, left, m->comm, m->stream));
        if (n_from_left > 0) SPH_NCCL_G(c, g_rccl.Recv(recv_left, (size_t)n_from_left * rec, ncclUint8, left, m->comm, m->stream));
    }
    if (right >= 0) {
        if (n_to_right > 0) SPH_NCCL_G(c, g_rccl.Send(send_right, (size_t)n_to_right * rec, ncclUint8, right, m->comm, m->stream));
        if (n_from_right > 0) SPH_NCCL_G(c, g_rccl.Recv(recv_right, (size_t)n_from_right * rec, ncclUint8, right, m->comm, m->stream));
    }
    SPH_NCCL(c, g_rccl.GroupEnd());
    SPH_HIP(c, hipEventRecord(m->ev_t1, m->stream));
    m->timed_open = true;
    SPH_HIP(c, hipEventRecord(m->ev_done, m->stream));
    SPH_HIP(c, hipStreamWaitEvent(c->stream, m->ev_done, 0));  // whatever the host enqueues next (the insert + sort) follows the exchange
    m->announced = false;
    return 0;
}

int32_t sph_comm_swap(SphContext* c, SphComm* m, int32_t left, int32_t right, const void* send_left, int64_t bytes_to_left,
                      const void* send_right, int64_t bytes_to_right, void* recv_left, int64_t bytes_from_left, void* recv_right,
                      int64_t bytes_from_right) {
    if (!c || !m || m->ctx != c || !peer_ok(m, left) || !peer_ok(m, right) || bytes_to_left < 0 || bytes_to_right < 0 ||
        bytes_from_left < 0 || bytes_from_right < 0)
        return SPH_E_INVALID;
    SPH_HIP(c, hipSetDevice(c->device));
    SPH_HIP(c, hipEventRecord(m->ev_in, c->stream));
    SPH_HIP(c, hipStreamWaitEvent(m->stream, m->ev_in, 0));
    SPH_NCCL(c, g_rccl.GroupStart());
    if (left >= 0) {
        if (bytes_to_left > 0) SPH_NCCL_G(c, g_rccl.Send(send_left, (size_t)bytes_to_left, ncclUint8, left, m->comm, m->stream));
        if (bytes_from_left > 0) SPH_NCCL_G(c, g_rccl.Recv(recv_left, (size_t)bytes_from_left, ncclUint8, left, m->comm, m->stream));
    }
    if (right >= 0) {
        if (bytes_to_right > 0) SPH_NCCL_G(c, g_rccl.Send(send_right, (size_t)bytes_to_right, ncclUint8, right, m->comm, m->stream));
        if (bytes_from_right > 0) SPH_NCCL_G(c, g_rccl.Recv(recv_right, (size_t)bytes_from_right, ncclUint8, right, m->comm, m->stream));
    }
    SPH_NCCL(c, g_rccl.GroupEnd());
    SPH_HIP(c, hipEventRecord(m->ev_done, m->stream));
    SPH_HIP(c, hipStreamWaitEvent(c->stream, m->ev_done, 0));
    return 0;
}

int32_t sph_comm_all_reduce(SphContext* c, SphComm* m, void* dev, int32_t n, int32_t dtype) {
    if (!c || !m || m->ctx != c || !dev || n <= 0 || (dtype != 0 && dtype != 1)) return SPH_E_INVALID;
    SPH_HIP(c, hipSetDevice(c->device));
    SPH_HIP(c, hipEventRecord(m->ev_in, c->stream));
    SPH_HIP(c, hipStreamWaitEvent(m->stream, m->ev_in, 0));
    SPH_NCCL(c, g_rccl.AllReduce(dev, dev, (size_t)n, dtype == 0 ? ncclFloat64 : ncclInt64, ncclSum, m->comm, m->stream));
    SPH_HIP(c, hipEventRecord(m->ev_done, m->stream));
    SPH_HIP(c, hipStreamWaitEvent(c->stream, m->ev_done, 0));
    return 0;
}

int32_t sph_comm_sync(SphContext* c, SphComm* m) {
    if (!c || !m || m->ctx != c) return SPH_E_INVALID;
    SPH_HIP(c, hipSetDevice(c->device));
    SPH_HIP(c, hipStreamSynchronize(m->stream));
    harvest_halo(m);
    return 0;
}

// What the communication library itself says about this communicator (ncclCommCount / ncclCommUserRank), not what the
// host passed to sph_comm_create: a bench line can show that RCCL saw the N ranks of the job.
int32_t sph_comm_info(SphContext* c, SphComm* m, int32_t* rank, int32_t* world) {
    if (!c || !m || m->ctx != c || !rank || !world) return SPH_E_INVALID;
    int r = -1, w = -1;
    SPH_NCCL(c, g_rccl.CommCount(m->comm, &w));
    SPH_NCCL(c, g_rccl.CommUserRank(m->comm, &r));
    *rank = r;
    *world = w;
    return 0;
}

int32_t sph_comm_halo_time(SphContext* c, SphComm* m, double* ms, int64_t* exchanges) {
    if (!c || !m || m->ctx != c || !ms || !exchanges) return SPH_E_INVALID;
    SPH_HIP(c, hipSetDevice(c->device));
    SPH_HIP(c, hipStreamSynchronize(m->stream));
    harvest_halo(m);
    *ms = m->halo_ms;
    *exchanges = m->exchanges;
    return 0;
}

}  // extern "C"
