// RCCL behind the C ABI: the one collective of the multi-GPU path - the all-gather of the decoded id blocks -
// issued by the library itself on a stream the caller names.  librccl is resolved at run time (dlopen: the copy the process already holds,
// e.g. the one PyTorch-ROCm loaded, else the system's), so the library links against nothing new; the rendezvous (who the ranks are, how the
// 128-byte unique id travels from rank 0 to the others) stays with the host - torch.distributed's store in markushgrapher_amd/dist.py,
// anything else in another embedding.  No counterpart in the reference (single device, utils/ocsr/utils_evaluation.py:140).
#include "mg_kernels.h"
#include "../../include/mgrapher.h"
#include <string>
#include <string.h>

#ifndef MG_EMU
#include <dlfcn.h>
#endif

namespace mg { int fail_msg(int code, const char* msg); }

struct mg_dist {
    void* comm = nullptr;
    int rank = 0, world = 1;
};

namespace {

#ifndef MG_EMU
struct UniqueId128 { char b[128]; };           // ncclUniqueId: an opaque 128-byte array, passed BY VALUE to ncclCommInitRank
typedef int (*GetUniqueId_t)(void*);
typedef int (*CommInitRankV_t)(void**, int, UniqueId128, int);
typedef int (*AllGather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*CommDestroy_t)(void*);
typedef const char* (*GetErrorString_t)(int);

struct Rccl {
    void* h = nullptr;
    GetUniqueId_t get_id = nullptr;
    CommInitRankV_t init = nullptr;
    AllGather_t all_gather = nullptr;
    CommDestroy_t destroy = nullptr;
    GetErrorString_t err = nullptr;
    std::string why;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) { x.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (x.h) break; }       // the copy already in the process first
        if (!x.h) for (const char* n : names) { x.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (x.h) break; }
        if (!x.h) { x.why = "librccl.so not found (dlopen)"; return x; }
        x.get_id = (GetUniqueId_t)dlsym(x.h, "ncclGetUniqueId");
        x.init = (CommInitRankV_t)dlsym(x.h, "ncclCommInitRank");
        x.all_gather = (AllGather_t)dlsym(x.h, "ncclAllGather");
        x.destroy = (CommDestroy_t)dlsym(x.h, "ncclCommDestroy");
        x.err = (GetErrorString_t)dlsym(x.h, "ncclGetErrorString");
        if (!x.get_id || !x.init || !x.all_gather || !x.destroy) x.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        return x;
    }();
    return r;
}
int rccl_fail(const char* what, int rc) {
    std::string m = std::string(what) + ": " + (rccl().err ? rccl().err(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
    return mg::fail_msg(MG_E_HIP, m.c_str());
}
#endif

}  // namespace

extern "C" {

int mg_dist_unique_id(void* out, int bytes) {
#ifdef MG_EMU
    (void)out; (void)bytes;
    return mg::fail_msg(MG_E_UNSUPPORTED, "mg_dist_unique_id: no RCCL in the emulator build");
#else
    if (!out || bytes < 128) return mg::fail_msg(MG_E_ARG, "mg_dist_unique_id: a 128-byte buffer is needed");
    Rccl& r = rccl();
    if (!r.why.empty()) return mg::fail_msg(MG_E_UNSUPPORTED, ("mg_dist_unique_id: " + r.why).c_str());
    const int rc = r.get_id(out);
    return rc == 0 ? MG_OK : rccl_fail("ncclGetUniqueId", rc);
#endif
}

int mg_dist_create(const void* unique_id, int bytes, int rank, int world, mg_dist** out) {
#ifdef MG_EMU
    (void)unique_id; (void)bytes; (void)rank; (void)world; (void)out;
    return mg::fail_msg(MG_E_UNSUPPORTED, "mg_dist_create: no RCCL in the emulator build");
#else
    if (!unique_id || bytes < 128 || !out || world < 1 || rank < 0 || rank >= world) return mg::fail_msg(MG_E_ARG, "mg_dist_create: bad argument");
    Rccl& r = rccl();
    if (!r.why.empty()) return mg::fail_msg(MG_E_UNSUPPORTED, ("mg_dist_create: " + r.why).c_str());
    UniqueId128 id;
    memcpy(id.b, unique_id, 128);
    mg_dist* d = new mg_dist();
    d->rank = rank; d->world = world;
    const int rc = r.init(&d->comm, world, id, rank);         // collective: every rank of the group calls it with rank 0's id
    if (rc != 0) { delete d; return rccl_fail("ncclCommInitRank", rc); }
    *out = d;
    return MG_OK;
#endif
}

int mg_dist_allgather(mg_dist* d, void* stream, const void* send, void* recv, size_t bytes_per_rank) {
#ifdef MG_EMU
    (void)d; (void)stream; (void)send; (void)recv; (void)bytes_per_rank;
    return mg::fail_msg(MG_E_UNSUPPORTED, "mg_dist_allgather: no RCCL in the emulator build");
#else
    if (!d || !d->comm || !send || !recv) return mg::fail_msg(MG_E_ARG, "mg_dist_allgather: null argument");
    const int rc = rccl().all_gather(send, recv, bytes_per_rank, /* ncclInt8 */ 0, d->comm, (hipStream_t)stream);
    return rc == 0 ? MG_OK : rccl_fail("ncclAllGather", rc);
#endif
}

void mg_dist_destroy(mg_dist* d) {
    if (!d) return;
#ifndef MG_EMU
    if (d->comm && rccl().destroy) (void)rccl().destroy(d->comm);
#endif
    delete d;
}

}  // extern "C"
