// tic_comm.hip - the path's only collective (north star): an all-gather of per-frame compressed sizes
// across the ranks of a node (one process per GPU), on RCCL, behind the C-ABI - no torch in the product.
//
// librccl.so is opened on first use (it is half a gigabyte; single-GPU users never load it) and is not a link-time dependency:
// types, enumerators and prototypes come from <rccl/rccl.h>, the six entry points are looked up with dlsym and every pointer's type is
// decltype(&ncclXxx) - a prototype that drifts in a later rccl.h fails the build instead of the call.
//
// Rendezvous: rank 0 creates the RCCL unique id and publishes it as a small file; the other ranks poll for it
// (tic_rdv_publish / tic_rdv_wait below).  The caller names the file - something unique to the launch AND to the communicator,
// e.g. /tmp/tic_rdv_<MASTER_PORT>_<launcher pid>_<launcher start time>_<sequence number> (tinyimgcodec_amd/distributed.py).
// The file is created exclusively (O_CREAT|O_EXCL|O_NOFOLLOW, mode 0600: never through a symlink, never over somebody else's
// file) under a temporary name and renamed into place; rank 0 removes whatever an earlier, crashed launch left under either
// name first, removes the file on every failure after publishing, and removes it once the first collective has completed.
// A reader accepts a file only if it carries the magic, the expected payload size and a publishing time not older than the
// reader's own process (a leftover of an earlier launch is ignored, not believed).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE // dladdr
#endif
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>

// ---- RCCL's types and enumerators come from its own header (rccl.h is on the build box; round 5 wrote them out by hand and trusted a
// version gate).  The LIBRARY is still opened at run time (dlopen below): nothing here links against it, the header only supplies
// ncclComm_t, ncclUniqueId, ncclResult_t, ncclDataType_t, ncclRedOp_t and the prototypes the function pointers are checked against.
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == 128, "the rendezvous file carries a 128-byte RCCL unique id");
static_assert(NCCL_MAJOR == 2, "written against the NCCL 2 API (tic_comm_create refuses another major version at run time too)");

#include "../../include/tinyimgcodec_hip.h"
#include "tic_hooks.h"

extern "C" __attribute__((visibility("hidden"))) int tic_ctx_device(const tic_ctx *ctx);
extern "C" __attribute__((visibility("hidden"))) void *tic_ctx_stream(const tic_ctx *ctx);

struct tic_comm {
    tic_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    void *lib = nullptr;
    ncclComm_t comm = nullptr;
    void *d_send = nullptr, *d_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
    std::string err;
    int version = 0; // NCCL_VERSION_CODE of the loaded library (0: no library loaded - single rank)
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static thread_local std::string g_comm_err;
static int comm_fail(tic_comm *c, int code, const char *what, const char *detail) {
    std::string m = std::string(what) + (detail ? std::string(": ") + detail : std::string());
    if (c) c->err = m;
    g_comm_err = m;
    return code;
}

static int ensure_bufs(tic_comm *c, size_t send_bytes, size_t recv_bytes) {
    if (send_bytes > c->send_cap) {
        if (c->d_send) (void)hipFree(c->d_send);
        c->d_send = nullptr;
        c->send_cap = 0;
        if (hipMalloc(&c->d_send, send_bytes) != hipSuccess) return comm_fail(c, TIC_E_HIP, "hipMalloc failed", nullptr);
        c->send_cap = send_bytes;
    }
    if (recv_bytes > c->recv_cap) {
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_recv = nullptr;
        c->recv_cap = 0;
        if (hipMalloc(&c->d_recv, recv_bytes) != hipSuccess) return comm_fail(c, TIC_E_HIP, "hipMalloc failed", nullptr);
        c->recv_cap = recv_bytes;
    }
    return TIC_OK;
}

// ---- rendezvous file ----------------------------------------------------------------------------------------------------
namespace {
struct RdvHeader {
    char magic[8];          // "TICRDV1\0"
    uint64_t published_ns;  // CLOCK_REALTIME at publication
    uint64_t payload_bytes;
};
const char kRdvMagic[8] = {'T', 'I', 'C', 'R', 'D', 'V', '1', 0};

uint64_t now_realtime_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
// CLOCK_REALTIME at which this process started (0 if /proc cannot tell): start time in clock ticks since boot against the uptime.
uint64_t process_start_realtime_ns() {
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return 0;
    char buf[2048];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')'); // the command name may contain anything: fields are counted behind it
    if (!p) return 0;
    unsigned long long start_ticks = 0;
    int field = 2;
    for (p++; *p && field < 22; p++)
        if (*p == ' ') field++;
    if (field != 22 || sscanf(p, "%llu", &start_ticks) != 1) return 0;
    double up = 0.0;
    f = fopen("/proc/uptime", "r");
    if (!f) return 0;
    const int ok = fscanf(f, "%lf", &up);
    fclose(f);
    const long hz = sysconf(_SC_CLK_TCK);
    if (ok != 1 || hz <= 0) return 0;
    const double age_s = up - (double)start_ticks / (double)hz;
    const uint64_t now = now_realtime_ns();
    if (age_s < 0 || (uint64_t)(age_s * 1e9) > now) return 0;
    return now - (uint64_t)(age_s * 1e9);
}
} // namespace

extern "C" {

// Publishes `payload` under `path` (see the head of this file).  Returns TIC_OK or TIC_E_ARG with the reason in tic_comm_last_error(NULL).
int tic_rdv_publish(const char *path, const void *payload, size_t bytes) {
    if (!path || !*path || (!payload && bytes)) return comm_fail(nullptr, TIC_E_ARG, "bad rendezvous arguments", nullptr);
    const std::string p(path), tmp = p + ".tmp";
    (void)unlink(p.c_str());   // leftovers of an earlier launch that used the same name
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_NOFOLLOW | O_WRONLY | O_CLOEXEC, 0600);
    if (fd < 0) return comm_fail(nullptr, TIC_E_ARG, "cannot create the rendezvous file", (tmp + ": " + strerror(errno)).c_str());
    RdvHeader h;
    memcpy(h.magic, kRdvMagic, 8);
    h.published_ns = now_realtime_ns();
    h.payload_bytes = bytes;
    bool ok = write(fd, &h, sizeof h) == (ssize_t)sizeof h && (bytes == 0 || write(fd, payload, bytes) == (ssize_t)bytes);
    ok = (fsync(fd) == 0) && ok;
    ok = (close(fd) == 0) && ok;
    if (!ok || rename(tmp.c_str(), p.c_str()) != 0) {
        (void)unlink(tmp.c_str());
        return comm_fail(nullptr, TIC_E_ARG, "cannot publish the rendezvous file", p.c_str());
    }
    return TIC_OK;
}

// Waits (at most timeout_ms) for a rendezvous file at `path` whose payload has `bytes` bytes and which was published no earlier
// than `not_before_ns` (CLOCK_REALTIME; 0 = this process's own start), and copies the payload out.
int tic_rdv_wait(const char *path, void *payload, size_t bytes, int timeout_ms, uint64_t not_before_ns) {
    if (!path || !*path || (!payload && bytes)) return comm_fail(nullptr, TIC_E_ARG, "bad rendezvous arguments", nullptr);
    if (not_before_ns == 0) { // (1 = any age: for names that are unique to the launch, e.g. inside the launcher's private directory)
        const uint64_t st = process_start_realtime_ns();
        not_before_ns = st > 2000000000ull ? st - 2000000000ull : 0; // (clock-tick granularity of the start time)
    }
    const uint64_t t_end = now_realtime_ns() + (uint64_t)(timeout_ms < 0 ? 0 : timeout_ms) * 1000000ull;
    bool saw_stale = false;
    long nap_ns = 50 * 1000; // polls double from 50 us to 20 ms: a collective of the file communicator completes in well under a millisecond
    for (;;) {
        const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        if (fd >= 0) {
            RdvHeader h;
            struct stat st;
            // (a file somebody else planted under a predictable name in a shared /tmp is not ours to believe)
            const bool whole = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && (size_t)st.st_size == sizeof h + bytes &&
                               read(fd, &h, sizeof h) == (ssize_t)sizeof h && memcmp(h.magic, kRdvMagic, 8) == 0 && h.payload_bytes == bytes;
            if (whole && h.published_ns >= not_before_ns) {
                const bool got = bytes == 0 || read(fd, payload, bytes) == (ssize_t)bytes;
                close(fd);
                if (got) return TIC_OK;
            } else {
                if (whole) saw_stale = true;
                close(fd);
            }
        }
        if (now_realtime_ns() >= t_end) break;
        struct timespec ts = {0, nap_ns};
        nanosleep(&ts, nullptr);
        if (nap_ns < 20 * 1000 * 1000) nap_ns *= 2;
    }
    return comm_fail(nullptr, TIC_E_ARG, saw_stale ? "only a stale rendezvous file (older than this process) was found" : "rendezvous file did not appear", path);
}

const char *tic_comm_last_error(const tic_comm *c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

int tic_comm_destroy(tic_comm *c) {
    if (!c) return TIC_OK;
    (void)hipSetDevice(tic_ctx_device(c->ctx));
    if (c->comm && c->CommDestroy) (void)c->CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    // (librccl stays loaded: unloading a library that owns GPU state at exit is asking for trouble)
    delete c;
    return TIC_OK;
}

int tic_comm_create(tic_ctx *ctx, int rank, int world, const char *rendezvous_path, tic_comm **out) {
    return tic_comm_create_ex(ctx, rank, world, rendezvous_path, 0, 120000, out);
}

// not_before_ns: the oldest publication time (CLOCK_REALTIME) a reader accepts - the LAUNCHER's start time when the ranks start at
// different times (a rank restarted or spawned long after rank 0 published), 0 = this process's own start, 1 = any age.
int tic_comm_create_ex(tic_ctx *ctx, int rank, int world, const char *rendezvous_path, uint64_t not_before_ns, int timeout_ms, tic_comm **out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return comm_fail(nullptr, TIC_E_ARG, "bad communicator arguments", nullptr);
    tic_comm *c = new tic_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    *out = nullptr;
    if (hipSetDevice(tic_ctx_device(ctx)) != hipSuccess) { delete c; return comm_fail(nullptr, TIC_E_HIP, "hipSetDevice failed", nullptr); }
    // TIC_COMM_FORCE_RCCL: a single rank goes through RCCL too (library load, communicator, collectives on the context's stream):
    // the only way to exercise this file on a one-GPU box (tests/test_gpu_parity.py::test_rccl_single_rank_smoke)
    const bool forced = world == 1 && tic::test_hook("TIC_COMM_FORCE_RCCL") != nullptr; // (tic_hooks.h: off unless TIC_TEST_HOOKS=1)
    if (world > 1 || forced) {
        if (world > 1 && (!rendezvous_path || !*rendezvous_path)) { delete c; return comm_fail(nullptr, TIC_E_ARG, "rendezvous path required for world > 1", nullptr); }
        // The RCCL that belongs to the HIP runtime this library runs on: the one in the same directory.  (A bare soname would
        // return whatever librccl.so.1 the process loaded first - e.g. the copy bundled with a PyTorch wheel, which brings up a
        // second HSA runtime and fails with "no ROCm-capable device is detected".)
        {
            Dl_info di;
            if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &di) && di.dli_fname) {
                std::string dir(di.dli_fname);
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) c->lib = dlopen((dir.substr(0, slash) + "/librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
            }
        }
        if (!c->lib) c->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!c->lib) c->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!c->lib) { const char *e = dlerror(); delete c; return comm_fail(nullptr, TIC_E_NODEVICE, "cannot load librccl", e); }
#define SYM(field, name)                                                                                  \
    c->field = reinterpret_cast<decltype(c->field)>(dlsym(c->lib, name));                                  \
    if (!c->field) { delete c; return comm_fail(nullptr, TIC_E_NODEVICE, "librccl lacks a symbol", name); }
        // Types and prototypes are rccl.h's of the build box (NCCL_MAJOR 2, static_assert above); the library found at RUN time may be
        // another one: a library of another MAJOR version may have changed any of them, so it is refused here instead of being called.
        {
            decltype(&ncclGetVersion) GetVersion = reinterpret_cast<decltype(&ncclGetVersion)>(dlsym(c->lib, "ncclGetVersion"));
            int ver = 0;
            if (!GetVersion || GetVersion(&ver) != ncclSuccess) { delete c; return comm_fail(nullptr, TIC_E_NODEVICE, "librccl does not report its version", "ncclGetVersion"); }
            const int major = ver >= 10000 ? ver / 10000 : ver / 1000; // NCCL_VERSION_CODE: X*10000 + Y*100 + Z (X*1000 + ... up to 2.8)
            c->version = ver;
            if (major != 2) {
                char buf[96];
                snprintf(buf, sizeof buf, "version code %d (major %d); this build binds the RCCL 2.x C API", ver, major);
                delete c;
                return comm_fail(nullptr, TIC_E_NODEVICE, "unsupported librccl", buf);
            }
        }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather")
        SYM(AllReduce, "ncclAllReduce")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        ncclUniqueId id;
        memset(&id, 0, sizeof id);
        const std::string path(rendezvous_path ? rendezvous_path : "");
        bool published = false;
        if (forced && path.empty()) {
            ncclResult_t r = c->GetUniqueId(&id);
            if (r != ncclSuccess) { const char *e = c->GetErrorString(r); delete c; return comm_fail(nullptr, TIC_E_HIP, "ncclGetUniqueId failed", e); }
        } else if (rank == 0) {
            ncclResult_t r = c->GetUniqueId(&id);
            if (r != ncclSuccess) { const char *e = c->GetErrorString(r); delete c; return comm_fail(nullptr, TIC_E_HIP, "ncclGetUniqueId failed", e); }
            const int rc = tic_rdv_publish(path.c_str(), &id, sizeof id);
            if (rc != TIC_OK) { delete c; return rc; }
            published = true;
        } else {
            const int rc = tic_rdv_wait(path.c_str(), &id, sizeof id, timeout_ms > 0 ? timeout_ms : 120000, not_before_ns);
            if (rc != TIC_OK) { delete c; return rc; }
        }
        (void)hipGetLastError(); // RCCL reads the thread's last HIP error: a stale one from an earlier, unrelated call would fail it
        ncclResult_t r = c->CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) {
            const char *e = c->GetErrorString(r);
            if (published) (void)unlink(path.c_str());
            delete c;
            return comm_fail(nullptr, TIC_E_HIP, "ncclCommInitRank failed", e);
        }
    }
    *out = c;
    if (c->comm) { // everybody has read the id once a first collective has completed: rank 0 removes the file
        double one = 1.0;
        int rc = tic_comm_allreduce_max(c, &one, 1);
        if (rank == 0 && rendezvous_path && *rendezvous_path) (void)unlink(rendezvous_path);
        if (rc != TIC_OK) { *out = nullptr; std::string keep = c->err; tic_comm_destroy(c); return comm_fail(nullptr, rc, "first collective failed", keep.c_str()); }
    }
    return TIC_OK;
}

int tic_comm_rank(const tic_comm *c) { return c ? c->rank : -1; }
int tic_comm_rccl_version(const tic_comm *c) { return c ? c->version : -1; }
int tic_comm_world(const tic_comm *c) { return c ? c->world : 0; }

int tic_gather_sizes(tic_comm *c, const uint64_t *mine, int n_mine, uint64_t *all) {
    if (!c || n_mine < 0 || (n_mine > 0 && (!mine || !all))) return comm_fail(c, TIC_E_ARG, "bad gather arguments", nullptr);
    if (n_mine == 0) return TIC_OK;
    const size_t sb = (size_t)n_mine * sizeof(uint64_t);
    if (!c->comm) { // a single rank without RCCL
        memcpy(all, mine, sb);
        return TIC_OK;
    }
    if (hipSetDevice(tic_ctx_device(c->ctx)) != hipSuccess) return comm_fail(c, TIC_E_HIP, "hipSetDevice failed", nullptr);
    int rc = ensure_bufs(c, sb, sb * (size_t)c->world);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)tic_ctx_stream(c->ctx);
    if (hipMemcpyAsync(c->d_send, mine, sb, hipMemcpyHostToDevice, st) != hipSuccess) return comm_fail(c, TIC_E_HIP, "upload failed", nullptr);
    (void)hipGetLastError();
    ncclResult_t r = c->AllGather(c->d_send, c->d_recv, (size_t)n_mine, ncclUint64, c->comm, st);
    if (r != ncclSuccess) return comm_fail(c, TIC_E_HIP, "ncclAllGather failed", c->GetErrorString(r));
    if (hipMemcpyAsync(all, c->d_recv, sb * (size_t)c->world, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return comm_fail(c, TIC_E_HIP, "download failed", nullptr);
    return TIC_OK;
}

int tic_comm_allreduce_max(tic_comm *c, double *vals, int n) {
    if (!c || n < 0 || (n > 0 && !vals)) return comm_fail(c, TIC_E_ARG, "bad all-reduce arguments", nullptr);
    if (n == 0 || !c->comm) return TIC_OK;
    if (hipSetDevice(tic_ctx_device(c->ctx)) != hipSuccess) return comm_fail(c, TIC_E_HIP, "hipSetDevice failed", nullptr);
    const size_t sb = (size_t)n * sizeof(double);
    int rc = ensure_bufs(c, sb, sb);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)tic_ctx_stream(c->ctx);
    if (hipMemcpyAsync(c->d_send, vals, sb, hipMemcpyHostToDevice, st) != hipSuccess) return comm_fail(c, TIC_E_HIP, "upload failed", nullptr);
    (void)hipGetLastError();
    ncclResult_t r = c->AllReduce(c->d_send, c->d_recv, (size_t)n, ncclFloat64, ncclMax, c->comm, st);
    if (r != ncclSuccess) return comm_fail(c, TIC_E_HIP, "ncclAllReduce failed", c->GetErrorString(r));
    if (hipMemcpyAsync(vals, c->d_recv, sb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return comm_fail(c, TIC_E_HIP, "download failed", nullptr);
    return TIC_OK;
}

} // extern "C"
