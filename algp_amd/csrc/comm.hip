// comm.hip -- the one collective of the sharded greedy loop, behind the C ABI (RCCL over xGMI).
//
// The reference's greedy loop (agent.py:313-354) evaluates every candidate independently given the factor of the
// sampled set, so the candidate list shards over the GPUs of a node (one process per GPU, one algp_ctx each, the
// factor replicated).  Per pick each rank resolves its own best candidate (algp_best_candidate: lazily, on its
// shard) and contributes the 16-byte pair (utility, pool index) to ONE ncclAllGather on the context's stream; a
// one-wave kernel takes the first maximum in rank order (= np.argmax over the concatenated scores, agent.py:349,
// shards being contiguous in rank order) and every rank commits that winner to its shard (a rank that does not own
// it rebuilds its row from the replicated factor on the device).  No other communication exists on the path.
// RCCL is opened with dlopen at algp_comm_init: the library loads and every single-GPU entry point works without it.
#include "common.h"
#include <dlfcn.h>
#include <link.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <rccl/rccl.h>

namespace algp {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// ONE copy of RCCL per process: a second one (PyTorch ships its own librccl.so next to the ROCm one) corrupts the heap
// at exit.  So: the copy that is already mapped (a process that imported torch first), else $ALGP_RCCL_PATH, else the
// system's.  (A process that opens the system's copy here and imports torch afterwards ends up with two.)
static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
        *(std::string*)data = info->dlpi_name;
        return 1;
    }
    return 0;
}

static RcclApi* rccl_api(std::string* why) {
    static RcclApi api;
    static bool tried = false;
    static std::string err;
    if (!tried) {
        tried = true;
        std::string loaded;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        const char* envp = getenv("ALGP_RCCL_PATH");
        const char* names[] = {loaded.empty() ? nullptr : loaded.c_str(), envp, "librccl.so.1", "librccl.so",
                               "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            err = std::string("dlopen(librccl.so): ") + (dlerror() ? dlerror() : "not found");
        } else {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
            api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) {
                err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
                api.handle = nullptr;
            }
        }
    }
    if (!api.handle) {
        if (why) *why = err;
        return nullptr;
    }
    return &api;
}

// pair[0] = the local best utility (-inf without a candidate), pair[1] = its pool index as a double (exact below 2^53)
__global__ void pack_best_kernel(const double* val, const int64_t* pos, const int64_t* cidx, double* pair) {
    const int64_t p = *pos;
    pair[0] = p >= 0 ? *val : -INFINITY;
    pair[1] = p >= 0 ? (double)cidx[p] : -1.0;
}
// first maximum in rank order (NaN never wins): out = (utility, pool index, owning rank)
__global__ void first_max_kernel(const double* pairs, int nranks, double* out) {
    double bv = -INFINITY, bi = -1.0, br = -1.0;
    for (int r = 0; r < nranks; ++r) {
        const double v = pairs[2 * r], i = pairs[2 * r + 1];
        if (i >= 0.0 && (bi < 0.0 || v > bv)) { bv = v; bi = i; br = (double)r; }
    }
    out[0] = bv;
    out[1] = bi;
    out[2] = br;
}

int comm_unique_id(void* out128, std::string* why) {
    RcclApi* api = rccl_api(why);
    if (!api) return ALGP_ERR_HIP;
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) {
        if (why) *why = std::string("ncclGetUniqueId: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed");
        return ALGP_ERR_HIP;
    }
    memcpy(out128, &id, NCCL_UNIQUE_ID_BYTES);
    return ALGP_OK;
}

int comm_init(algp_ctx* c, int nranks, int rank, const void* unique_id128) {
    std::string why;
    RcclApi* api = rccl_api(&why);
    if (!api) return fail(c, ALGP_ERR_HIP, "comm_init: " + why);
    if (c->comm) {
        api->CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm;
    const ncclResult_t r = api->CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess)
        return fail(c, ALGP_ERR_HIP, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed"));
    c->comm = comm;
    c->comm_nranks = nranks;
    c->comm_rank = rank;
    return ensure(c, c->commbuf, sizeof(double) * (2 + 2 * (size_t)nranks + 4));
}

void comm_destroy(algp_ctx* c) {
    if (!c->comm) return;
    RcclApi* api = rccl_api(nullptr);
    if (api) api->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->comm_nranks = 1;
    c->comm_rank = 0;
}

// (val_dev, pos_dev): the local argmax as the lazy greedy left it on the device; returns the global winner
int comm_gather_winner(algp_ctx* c, const double* val_dev, const int64_t* pos_dev, const int64_t* cidx_dev, double* winner3) {
    RcclApi* api = rccl_api(nullptr);
    if (!api || !c->comm) return fail(c, ALGP_ERR_STATE, "greedy_sharded: call algp_comm_init first");
    double* pair = (double*)c->commbuf.p;
    double* all = pair + 2;
    double* out = all + 2 * c->comm_nranks;
    hipLaunchKernelGGL(pack_best_kernel, dim3(1), dim3(1), 0, c->stream, val_dev, pos_dev, cidx_dev, pair);
    ALGP_HIP(hipGetLastError());
    const ncclResult_t r = api->AllGather(pair, all, 2, ncclDouble, (ncclComm_t)c->comm, c->stream);
    if (r != ncclSuccess)
        return fail(c, ALGP_ERR_HIP, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed"));
    hipLaunchKernelGGL(first_max_kernel, dim3(1), dim3(1), 0, c->stream, all, c->comm_nranks, out);
    ALGP_HIP(hipGetLastError());
    ALGP_HIP(hipMemcpyAsync(winner3, out, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    return ALGP_OK;
}

}  // namespace algp
