// comm.hip -- R1: the one collective of the path, behind the C ABI for hosts that have no
// torch.distributed (the Go/cgo drop-in): RCCL all-gather of per-rank sketches over xGMI.
//
// One process per GPU.  RCCL is resolved at RUN time (dlopen "librccl.so.1", then "librccl.so"): the
// library has no link-time dependency on it, a process that never calls polyhip_comm_* never loads it,
// and inside a PyTorch process the already loaded librccl is the one that gets used.
// Rendezvous is the host's business: rank 0 calls polyhip_comm_unique_id and hands the 128 bytes to the
// other ranks by whatever channel it has (the Go driver: a pipe / env / file), then every rank calls
// polyhip_comm_init_rank.  One in-flight collective per communicator (serialise calls on a comm).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace polyhip {
namespace r1 {

typedef struct { char internal[128]; } UniqueId; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;                               // ncclComm_t
constexpr int NCCL_UINT8 = 1;                     // ncclUint8 (ncclDataType_t)
constexpr int NCCL_UINT32 = 3;                    // ncclUint32

struct Api {
    void *handle = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    char why[256] = "";
};

static Api &api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle)
                break;
        }
        if (!a.handle) {
            snprintf(a.why, sizeof a.why, "dlopen(librccl.so.1): %s", dlerror());
            return;
        }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
        a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(a.handle, "ncclBroadcast"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(a.handle, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(a.handle, "ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.Broadcast || !a.GroupStart ||
            !a.GroupEnd) {
            snprintf(a.why, sizeof a.why,
                     "librccl lacks one of ncclGetUniqueId/CommInitRank/CommDestroy/AllGather/Broadcast/GroupStart/GroupEnd");
            a.handle = nullptr;
        }
    });
    return a;
}

static int rccl_error(const char *what, int rc)
{
    Api &a = api();
    return set_error(POLYHIP_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, a.GetErrorString ? a.GetErrorString(rc) : "?");
}

} // namespace r1
} // namespace polyhip

using namespace polyhip;

struct polyhip_comm {
    r1::Comm comm;
    int rank, nranks;
};

extern "C" {

int polyhip_comm_unique_id(uint8_t id[128])
{
    PH_REQUIRE(id, "polyhip_comm_unique_id: null pointer");
    r1::Api &a = r1::api();
    if (!a.handle)
        return set_error(POLYHIP_ERR_HIP, "polyhip_comm: RCCL unavailable: %s", a.why);
    r1::UniqueId u;
    const int rc = a.GetUniqueId(&u);
    if (rc != 0)
        return r1::rccl_error("ncclGetUniqueId", rc);
    memcpy(id, u.internal, 128);
    return POLYHIP_OK;
}

int polyhip_comm_init_rank(const uint8_t id[128], int rank, int nranks, polyhip_comm **out)
{
    PH_REQUIRE(id && out, "polyhip_comm_init_rank: null pointer");
    PH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "polyhip_comm_init_rank: rank %d of %d", rank, nranks);
    r1::Api &a = r1::api();
    if (!a.handle)
        return set_error(POLYHIP_ERR_HIP, "polyhip_comm: RCCL unavailable: %s", a.why);
    r1::UniqueId u;
    memcpy(u.internal, id, 128);
    polyhip_comm *c = new polyhip_comm{nullptr, rank, nranks};
    const int rc = a.CommInitRank(&c->comm, nranks, u, rank); // uses the calling thread's current HIP device
    if (rc != 0) {
        delete c;
        return r1::rccl_error("ncclCommInitRank", rc);
    }
    *out = c;
    return POLYHIP_OK;
}

int polyhip_comm_destroy(polyhip_comm *c)
{
    if (!c)
        return POLYHIP_OK;
    r1::Api &a = r1::api();
    int rc = 0;
    if (a.handle && c->comm)
        rc = a.CommDestroy(c->comm);
    delete c;
    return rc == 0 ? POLYHIP_OK : r1::rccl_error("ncclCommDestroy", rc);
}

int polyhip_comm_rank(const polyhip_comm *c) { return c ? c->rank : -1; }
int polyhip_comm_size(const polyhip_comm *c) { return c ? c->nranks : -1; }

int polyhip_allgather_sketches_dev(polyhip_comm *c, const uint32_t *d_local, uint64_t n_local, uint32_t s,
                                   uint32_t *d_all, polyhip_stream_t stream)
{
    PH_REQUIRE(c && c->comm, "polyhip_allgather_sketches: null communicator");
    PH_REQUIRE((d_local && d_all) || n_local == 0, "polyhip_allgather_sketches: null pointer");
    // Always enter the collective, also with an empty contribution: a rank that returned early would leave the
    // others waiting.  ncclAllGather needs the SAME count on every rank (see the header: pad ragged shards).
    r1::Api &a = r1::api();
    const int rc = a.AllGather(d_local, d_all, (size_t)n_local * s, r1::NCCL_UINT32, c->comm, as_stream(stream));
    if (rc != 0)
        return r1::rccl_error("ncclAllGather", rc);
    return POLYHIP_OK;
}

// Ragged all-gather in place: rank r's segment of d_buf is bytes [offsets[r], offsets[r+1]) and every rank ends up with
// all of them.  One grouped call of nranks broadcasts (each rank is the root of its own segment) -- ncclAllGather needs
// equal counts, and padding an inverted index's parts to the largest would move bytes nobody needs.
int polyhip_allgatherv_dev(polyhip_comm *c, void *d_buf, const uint64_t *offsets, polyhip_stream_t stream)
{
    PH_REQUIRE(c && c->comm, "polyhip_allgatherv: null communicator");
    PH_REQUIRE(offsets, "polyhip_allgatherv: null offsets");
    for (int r = 0; r < c->nranks; ++r)
        PH_REQUIRE(offsets[r] <= offsets[r + 1], "polyhip_allgatherv: offsets not ascending at rank %d", r);
    PH_REQUIRE(d_buf || offsets[c->nranks] == offsets[0], "polyhip_allgatherv: null buffer");
    r1::Api &a = r1::api();
    int rc = a.GroupStart();
    if (rc != 0)
        return r1::rccl_error("ncclGroupStart", rc);
    int first_bad = 0;
    for (int r = 0; r < c->nranks; ++r) {
        uint8_t *seg = static_cast<uint8_t *>(d_buf) + offsets[r];
        const int e = a.Broadcast(seg, seg, (size_t)(offsets[r + 1] - offsets[r]), r1::NCCL_UINT8, r, c->comm, as_stream(stream));
        if (e != 0 && first_bad == 0)
            first_bad = e;
    }
    rc = a.GroupEnd(); // always close the group, also after a failed member
    if (first_bad != 0)
        return r1::rccl_error("ncclBroadcast", first_bad);
    if (rc != 0)
        return r1::rccl_error("ncclGroupEnd", rc);
    return POLYHIP_OK;
}

} // extern "C"
