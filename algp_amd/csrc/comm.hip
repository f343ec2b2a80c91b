// comm.hip -- the one collective of the sharded greedy loop, behind the C ABI (RCCL over xGMI).
//
// The reference's greedy loop (agent.py:313-354) evaluates every candidate independently given the factor of the
// sampled set, so the candidate list shards over the GPUs of a node (one process per GPU, one algp_ctx each, the
// factor replicated).  Per pick each rank resolves its own best candidate on the device (argmax -> refresh of the rows
// whose bound can still win -> argmax, see api.hip) and contributes the 24-byte triple (utility, pool index, status) to
// ONE all-gather on the context's stream; a one-thread kernel takes the first maximum in rank order (= np.argmax over
// the concatenated scores, agent.py:349, shards being contiguous in rank order) and the worst status, and the 40-byte
// result is the pick's ONLY read-back.  The status word is what keeps the ranks together: a rank that cannot score
// (no solve, an allocation that failed, ...) still takes part in the gather and reports its error code there, so every
// rank returns that error instead of waiting for a peer that left.  Every rank then commits the same winner to its shard
// (a rank that does not own it rebuilds its row from the replicated factor on the device).
// Transports: RCCL (algp_comm_init; opened with dlopen, so the library loads and every single-GPU entry point works
// without it), a caller-supplied host all-gather (algp_comm_init_host: MPI, gloo, ... -- also what lets two ranks share
// ONE card in the tests, which RCCL refuses), or none (one rank: the same kernels without the gather).
#include "common.h"
#include <algorithm>
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
        // $ALGP_RCCL_PATH, when set, is the only file tried (besides a copy the process has mapped already)
        const char* envp = getenv("ALGP_RCCL_PATH");
        const bool only_env = envp && *envp;
        const char* names[] = {loaded.empty() ? nullptr : loaded.c_str(), envp, only_env ? nullptr : "librccl.so.1",
                               only_env ? nullptr : "librccl.so", only_env ? nullptr : "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
            const char* e = dlerror();                  // one call per failure: dlerror() clears the state it returns
            err = std::string("dlopen(") + n + "): " + (e ? e : "not found");
        }
        if (!api.handle) {
            if (err.empty()) err = "dlopen(librccl.so): no candidate path";
        } else {
            err.clear();
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

// triple[0] = the local best utility (-inf without a candidate), [1] = its pool index as a double (exact below 2^53;
// -1: none), [2] = status: 0 fine | 1 the best row still lags behind the committed picks (its utility is only an upper
// bound: one more refresh round) | >= 2 the ALGP_ERR_* code this rank failed with.  pos == null: no candidate.
__global__ void pack_best_kernel(const double* val, const int64_t* pos, const int64_t* cidx, const int* fresh, int npicks,
                                 int status, double* triple) {
    const int64_t p = pos ? *pos : -1;
    triple[0] = p >= 0 ? *val : -INFINITY;
    triple[1] = p >= 0 ? (double)cidx[p] : -1.0;
    triple[2] = status != 0 ? (double)status : ((p >= 0 && fresh && fresh[p] < npicks) ? 1.0 : 0.0);
}
// out = (utility, pool index, owning rank, status, first rank with a non-zero status): the first maximum in rank order
// over the ranks that have a candidate -- a NaN utility never wins (the local argmax skips NaN as well), -inf does when
// nothing else is on offer -- and the largest status word (error codes are >= 2, so they outrank "one more round")
__global__ void first_max_kernel(const double* triples, int nranks, double* out) {
    double bv = -INFINITY, bi = -1.0, br = -1.0, st = 0.0, bad = -1.0;
    for (int r = 0; r < nranks; ++r) {
        const double v = triples[3 * r], i = triples[3 * r + 1], s = triples[3 * r + 2];
        if (s != 0.0) {
            if (bad < 0.0) bad = (double)r;
            if (!(s <= st)) st = s;                     // a NaN status counts as a failure too
        }
        if (i >= 0.0 && v == v && (bi < 0.0 || v > bv)) { bv = v; bi = i; br = (double)r; }
    }
    out[0] = bv;
    out[1] = bi;
    out[2] = br;
    out[3] = st == st ? st : (double)ALGP_ERR_HIP;
    out[4] = bad;
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
    c->host_gather = nullptr;
    c->comm_nranks = nranks;
    c->comm_rank = rank;
    return ALGP_OK;
}

int comm_init_host(algp_ctx* c, int nranks, int rank, algp_allgather_fn fn, void* user) {
    comm_destroy(c);
    c->host_gather = fn;
    c->host_gather_user = user;
    c->comm_nranks = nranks;
    c->comm_rank = rank;
    return ALGP_OK;
}

void comm_destroy(algp_ctx* c) {
    if (c->comm) {
        RcclApi* api = rccl_api(nullptr);
        if (api) api->CommDestroy((ncclComm_t)c->comm);
    }
    c->comm = nullptr;
    c->host_gather = nullptr;
    c->host_gather_user = nullptr;
    c->comm_nranks = 1;
    c->comm_rank = 0;
}

// The exchange of one pick.  (val_dev, pos_dev): the local argmax as the kernels before left it on the device (null: this
// rank has no candidate to offer); status: 0 or the ALGP_ERR_* code this rank failed with while preparing it.  Everything
// is stream-ordered; the single synchronisation is the read-back of rec5 = (utility, pool index, owner, status, first
// rank with a non-zero status), identical on every rank.  Returns non-zero only when the exchange ITSELF failed.
int comm_pick_exchange(algp_ctx* c, const double* val_dev, const int64_t* pos_dev, const int64_t* cidx_dev,
                       const int* fresh_dev, int npicks, int status, double* rec5) {
    const int nr = c->comm_nranks;
    ALGP_TRY(ensure(c, c->commbuf, sizeof(double) * (3 + 3 * (size_t)nr + 5)));
    double* triple = (double*)c->commbuf.p;
    double* all = triple + 3;
    double* out = all + 3 * nr;
    hipLaunchKernelGGL(pack_best_kernel, dim3(1), dim3(1), 0, c->stream, val_dev, pos_dev, cidx_dev, fresh_dev, npicks, status,
                       triple);
    ALGP_HIP(hipGetLastError());
    const double* gathered = all;
    if (c->comm) {
        RcclApi* api = rccl_api(nullptr);
        if (!api) return fail(c, ALGP_ERR_STATE, "greedy_sharded: the RCCL communicator has no library behind it");
        const ncclResult_t r = api->AllGather(triple, all, 3, ncclDouble, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess)
            return fail(c, ALGP_ERR_HIP, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed"));
    } else if (c->host_gather) {
        // the caller's transport works on host memory: the triple goes down, the gathered triples come back up
        std::vector<double> send(3), recv(3 * (size_t)nr);
        ALGP_HIP(hipMemcpyAsync(send.data(), triple, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        ALGP_HIP(hipStreamSynchronize(c->stream));
        c->n_syncs++;
        const int rc = c->host_gather(c->host_gather_user, send.data(), recv.data(), (int64_t)(3 * sizeof(double)));
        if (rc != 0) return fail(c, ALGP_ERR_HIP, "greedy_sharded: the caller's all-gather returned " + std::to_string(rc));
        ALGP_HIP(hipMemcpyAsync(all, recv.data(), 3 * sizeof(double) * nr, hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipStreamSynchronize(c->stream));       // recv goes out of scope
        c->n_syncs++;
    } else {
        gathered = triple;                               // one rank: the same reduction over its own triple
    }
    hipLaunchKernelGGL(first_max_kernel, dim3(1), dim3(1), 0, c->stream, gathered, c->comm || c->host_gather ? nr : 1, out);
    ALGP_HIP(hipGetLastError());
    ALGP_HIP(hipMemcpyAsync(rec5, out, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    return ALGP_OK;
}

// test hook: first_max_kernel over a caller-made buffer of `nranks` triples (fabricated 8-rank cases on one GPU)
int comm_debug_first_max(algp_ctx* c, const double* triples, int nranks, double* out5) {
    ALGP_TRY(ensure(c, c->commbuf, sizeof(double) * (3 + 3 * (size_t)std::max(nranks, c->comm_nranks) + 5)));
    double* all = (double*)c->commbuf.p + 3;
    double* out = all + 3 * std::max(nranks, c->comm_nranks);
    ALGP_HIP(hipMemcpyAsync(all, triples, 3 * sizeof(double) * nranks, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(first_max_kernel, dim3(1), dim3(1), 0, c->stream, all, nranks, out);
    ALGP_HIP(hipGetLastError());
    ALGP_HIP(hipMemcpyAsync(out5, out, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    return ALGP_OK;
}

}  // namespace algp
