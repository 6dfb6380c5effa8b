// tic_comm.hip - the path's only collective (north star): an all-gather of per-frame compressed sizes
// across the ranks of a node (one process per GPU), on RCCL, behind the C-ABI - no torch in the product.
//
// librccl.so is opened on first use (it is half a gigabyte; single-GPU users never load it).  Rendezvous: rank 0 creates
// the RCCL unique id and publishes it as a small file (written under a temporary name, then renamed); the other ranks poll
// for it.  The caller names the file - something unique to the launch, e.g. /tmp/tic_rdv_<MASTER_PORT>_<launcher pid>.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE // dladdr
#endif
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>

#include "../../include/tinyimgcodec_hip.h"

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
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
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

extern "C" {

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
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return comm_fail(nullptr, TIC_E_ARG, "bad communicator arguments", nullptr);
    tic_comm *c = new tic_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    *out = nullptr;
    if (hipSetDevice(tic_ctx_device(ctx)) != hipSuccess) { delete c; return comm_fail(nullptr, TIC_E_HIP, "hipSetDevice failed", nullptr); }
    // TIC_COMM_FORCE_RCCL: a single rank goes through RCCL too (library load, communicator, collectives on the context's stream):
    // the only way to exercise this file on a one-GPU box (tests/test_gpu_parity.py::test_rccl_single_rank_smoke)
    const bool forced = world == 1 && getenv("TIC_COMM_FORCE_RCCL") != nullptr;
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
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather")
        SYM(AllReduce, "ncclAllReduce")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        ncclUniqueId id;
        memset(&id, 0, sizeof id);
        const std::string path(rendezvous_path ? rendezvous_path : ""), tmp = path + ".tmp";
        if (forced && path.empty()) {
            ncclResult_t r = c->GetUniqueId(&id);
            if (r != ncclSuccess) { const char *e = c->GetErrorString(r); delete c; return comm_fail(nullptr, TIC_E_HIP, "ncclGetUniqueId failed", e); }
        } else if (rank == 0) {
            ncclResult_t r = c->GetUniqueId(&id);
            if (r != ncclSuccess) { const char *e = c->GetErrorString(r); delete c; return comm_fail(nullptr, TIC_E_HIP, "ncclGetUniqueId failed", e); }
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) { if (f) fclose(f); delete c; return comm_fail(nullptr, TIC_E_ARG, "cannot write the rendezvous file", tmp.c_str()); }
            fclose(f);
            if (rename(tmp.c_str(), path.c_str()) != 0) { delete c; return comm_fail(nullptr, TIC_E_ARG, "cannot publish the rendezvous file", path.c_str()); }
        } else {
            bool got = false;
            for (int tries = 0; tries < 6000 && !got; tries++) { // up to 2 minutes
                struct stat st;
                if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == sizeof id) {
                    FILE *f = fopen(path.c_str(), "rb");
                    if (f) {
                        got = fread(&id, sizeof id, 1, f) == 1;
                        fclose(f);
                    }
                }
                if (!got) {
                    struct timespec ts = {0, 20 * 1000 * 1000};
                    nanosleep(&ts, nullptr);
                }
            }
            if (!got) { delete c; return comm_fail(nullptr, TIC_E_ARG, "rendezvous file did not appear", path.c_str()); }
        }
        (void)hipGetLastError(); // RCCL reads the thread's last HIP error: a stale one from an earlier, unrelated call would fail it
        ncclResult_t r = c->CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) { const char *e = c->GetErrorString(r); delete c; return comm_fail(nullptr, TIC_E_HIP, "ncclCommInitRank failed", e); }
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
