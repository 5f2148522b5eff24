// lps_comm.hip — the one collective of the multi-GPU path: rank 0's packed SNP table (and, optionally, reference slices) reaches the other
// GPUs of the node by ncclBroadcast (RCCL over xGMI); after that every GPU phases its own contigs with no further exchange (SURVEY.md §8e;
// the reference's analogue is the OpenMP loop over chromosomes, src/phase/PhasingProcess.cpp:113-173, which shares one parsed SnpParser).
//
// librccl.so is opened lazily (dlopen) when the first communicator is created: a single-GPU run never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "lps_common.h"

namespace {
struct Rccl {
    void *so = nullptr; std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr; decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr; decltype(&ncclCommCount) CommCount = nullptr; decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl &rccl() {
    static Rccl r; static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.so) break; }
        if (!r.so) { r.err = std::string("librccl.so not found: ") + dlerror(); return; }
#define RCCL_SYM(f) r.f = (decltype(r.f))dlsym(r.so, "nccl" #f); if (!r.f) { r.err = "librccl.so lacks nccl" #f; return; }
        RCCL_SYM(GetUniqueId) RCCL_SYM(CommInitRank) RCCL_SYM(CommInitAll) RCCL_SYM(CommDestroy) RCCL_SYM(CommCount) RCCL_SYM(Broadcast) RCCL_SYM(GetErrorString)
#undef RCCL_SYM
    });
    return r;
}
thread_local std::string g_comm_err;
}  // namespace

struct lps_comm {
    ncclComm_t comm = nullptr; int device = 0, rank = 0, n_ranks = 1; hipStream_t stream = nullptr;
    uint8_t *buf = nullptr; size_t cap = 0;          // device staging of lps_comm_bcast (host buffers)
    uint8_t *pin = nullptr; size_t pin_cap = 0;
};

extern "C" {

const char *lps_comm_last_error(void) { return g_comm_err.c_str(); }

int lps_comm_unique_id(uint8_t id[128]) {
    Rccl &R = rccl();
    if (!R.err.empty()) { g_comm_err = R.err; return -1; }
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id size");
    ncclUniqueId u; const ncclResult_t rc = R.GetUniqueId(&u);
    if (rc != ncclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + R.GetErrorString(rc); return -1; }
    memcpy(id, &u, 128);
    return 0;
}

static lps_comm *comm_finish(lps_comm *c) {
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        g_comm_err = "hipStreamCreate failed";
        if (c->comm) (void)rccl().CommDestroy(c->comm);
        delete c; return nullptr;
    }
    return c;
}

lps_comm *lps_comm_create(int device, int n_ranks, int rank, const uint8_t id[128]) {
    Rccl &R = rccl();
    if (!R.err.empty()) { g_comm_err = R.err; return nullptr; }
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !id) { g_comm_err = "lps_comm_create: bad rank / rank count"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { g_comm_err = "lps_comm_create: hipSetDevice failed"; return nullptr; }
    lps_comm *c = new lps_comm(); c->device = device; c->rank = rank; c->n_ranks = n_ranks;
    ncclUniqueId u; memcpy(&u, id, 128);
    const ncclResult_t rc = R.CommInitRank(&c->comm, n_ranks, u, rank);
    if (rc != ncclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + R.GetErrorString(rc); delete c; return nullptr; }
    return comm_finish(c);
}

int lps_comm_create_all(int n_devices, const int *devices, lps_comm **out) {
    Rccl &R = rccl();
    if (!R.err.empty()) { g_comm_err = R.err; return -1; }
    if (n_devices < 1 || n_devices > 64 || !devices || !out) { g_comm_err = "lps_comm_create_all: bad arguments"; return -1; }
    ncclComm_t comms[64];
    const ncclResult_t rc = R.CommInitAll(comms, n_devices, devices);
    if (rc != ncclSuccess) { g_comm_err = std::string("ncclCommInitAll: ") + R.GetErrorString(rc); return -1; }
    for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
    for (int i = 0; i < n_devices; ++i) {
        lps_comm *c = new lps_comm(); c->comm = comms[i]; c->device = devices[i]; c->rank = i; c->n_ranks = n_devices;
        out[i] = comm_finish(c);
        if (!out[i]) {                                                   // nothing half-built stays behind: the communicators made so far and the handles not yet wrapped
            for (int k = 0; k < i; ++k) { lps_comm_destroy(out[k]); out[k] = nullptr; }
            for (int k = i + 1; k < n_devices; ++k) (void)R.CommDestroy(comms[k]);
            return -1;
        }
    }
    return 0;
}

int lps_comm_size(lps_comm *c) {
    if (!c) return -1;
    int n = 0; if (rccl().CommCount(c->comm, &n) != ncclSuccess) return -1;
    return n;
}
int lps_comm_rank(lps_comm *c) { return c ? c->rank : -1; }

void lps_comm_destroy(lps_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->buf) (void)hipFree(c->buf);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int lps_comm_bcast_device(lps_comm *c, void *dev_buf, int64_t n_bytes, int root, double *ms) {
    if (!c || n_bytes < 0 || (n_bytes && !dev_buf) || root < 0 || root >= c->n_ranks) { g_comm_err = "lps_comm_bcast_device: bad arguments"; return -1; }
    Rccl &R = rccl();
    if (hipSetDevice(c->device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return -1; }
    hipEvent_t e0 = nullptr, e1 = nullptr; bool timed = ms != nullptr;
    if (timed) timed = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, c->stream) == hipSuccess;
    const ncclResult_t rc = R.Broadcast(dev_buf, dev_buf, (size_t)n_bytes, ncclUint8, root, c->comm, c->stream);
    if (timed) timed = hipEventRecord(e1, c->stream) == hipSuccess;
    const hipError_t he = hipStreamSynchronize(c->stream);
    if (ms) { float f = -1.f; if (!timed || hipEventElapsedTime(&f, e0, e1) != hipSuccess) f = -1.f; *ms = f; }   // -1: the events failed, not a measurement
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc != ncclSuccess) { g_comm_err = std::string("ncclBroadcast: ") + R.GetErrorString(rc); return -2; }
    if (he != hipSuccess) { g_comm_err = std::string("ncclBroadcast stream: ") + hipGetErrorString(he); return -2; }
    return 0;
}

// The table stays where the collective put it: root uploads `host_src` into the communicator's device buffer, ncclBroadcast fills the same buffer
// on the other ranks, and *dev_out is that buffer on every rank (valid until the next broadcast on this communicator or lps_comm_destroy) -
// what lps_set_variants_device takes.  No copy back to the host anywhere.
int lps_comm_bcast_to_device(lps_comm *c, const void *host_src, int64_t n_bytes, int root, void **dev_out, double *ms) {
    if (!c || n_bytes < 0 || !dev_out || root < 0 || root >= c->n_ranks || (c->rank == root && n_bytes && !host_src)) { g_comm_err = "lps_comm_bcast_to_device: bad arguments"; return -1; }
    *dev_out = nullptr;
    if (n_bytes == 0) return 0;
    if (hipSetDevice(c->device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return -1; }
    if ((size_t)n_bytes > c->cap) { if (c->buf) (void)hipFree(c->buf); c->buf = nullptr; c->cap = 0; if (hipMalloc((void **)&c->buf, (size_t)n_bytes + 256) != hipSuccess) { g_comm_err = "lps_comm_bcast_to_device: device allocation failed"; return -1; } c->cap = (size_t)n_bytes; }
    if (c->rank == root && hipMemcpy(c->buf, host_src, (size_t)n_bytes, hipMemcpyHostToDevice) != hipSuccess) { g_comm_err = "lps_comm_bcast_to_device: H2D failed"; return -1; }
    const int rc = lps_comm_bcast_device(c, c->buf, n_bytes, root, ms);
    if (rc) return rc;
    *dev_out = c->buf;
    return 0;
}

int lps_comm_bcast(lps_comm *c, void *host_buf, int64_t n_bytes, int root, double *ms) {
    if (!c || n_bytes < 0 || (n_bytes && !host_buf)) { g_comm_err = "lps_comm_bcast: bad arguments"; return -1; }
    if (n_bytes == 0) return 0;
    if (hipSetDevice(c->device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return -1; }
    if ((size_t)n_bytes > c->cap) { if (c->buf) (void)hipFree(c->buf); c->buf = nullptr; c->cap = 0; if (hipMalloc((void **)&c->buf, (size_t)n_bytes + 256) != hipSuccess) { g_comm_err = "lps_comm_bcast: device staging allocation failed"; return -1; } c->cap = (size_t)n_bytes; }
    if (c->rank == root && hipMemcpy(c->buf, host_buf, (size_t)n_bytes, hipMemcpyHostToDevice) != hipSuccess) { g_comm_err = "lps_comm_bcast: H2D failed"; return -1; }
    const int rc = lps_comm_bcast_device(c, c->buf, n_bytes, root, ms);
    if (rc) return rc;
    if (c->rank != root && hipMemcpy(host_buf, c->buf, (size_t)n_bytes, hipMemcpyDeviceToHost) != hipSuccess) { g_comm_err = "lps_comm_bcast: D2H failed"; return -1; }
    return 0;
}

}  // extern "C"
