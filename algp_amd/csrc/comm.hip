// comm.hip -- the one collective of the sharded greedy loop, behind the C ABI (RCCL over xGMI).
//
// The reference's greedy loop (agent.py:313-354) evaluates every candidate independently given the factor of the
// sampled set, so the candidate list shards over the GPUs of a node (one process per GPU, one algp_ctx each, the
// factor replicated).  Per pick each rank resolves its own best candidate on the device (argmax -> refresh of the rows
// whose bound can still win -> argmax, see api.hip) and contributes (utility, pool index, status) + that candidate's
// statistic and row of V^T (comm_payload_bytes: 32 B + one row, ~80 KB at N = 10 000 fp64) to
// ONE all-gather on the context's stream; a one-thread kernel takes the first maximum in rank order (= np.argmax over
// the concatenated scores, agent.py:349, shards being contiguous in rank order) and the worst status, and the 40-byte
// result is the pick's ONLY read-back.  The status word is what keeps the ranks together: a rank that cannot score
// (no solve, an allocation that failed, ...) still takes part in the gather and reports its error code there, so every
// rank returns that error instead of waiting for a peer that left.  Every rank then commits the same winner to its shard
// (a rank that does not own it copies the owner's row out of the gather buffer: no second collective, no rebuild).
// Transports: RCCL (algp_comm_init; opened with dlopen, so the library loads and every single-GPU entry point works
// without it), a caller-supplied host all-gather (algp_comm_init_host: MPI, gloo, ... -- also what lets two ranks share
// ONE card in the tests, which RCCL refuses), or none (one rank: the same kernels without the gather).
#include "common.h"
#include <mutex>
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

// (Loaded once per process, under a lock: two contexts of one process may attach their communicators from two threads at
// the same time -- the eight-rank tests do -- and the second must not see a half-initialised table.  Round 5's bare flag
// let it: "comm_init: " with an empty reason on one rank, the others waiting for it in ncclCommInitRank.)
static RcclApi* rccl_api(std::string* why) {
    static RcclApi api;
    static bool tried = false;
    static std::string err;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        std::string loaded;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        // $ALGP_RCCL_PATH, when set, is the ONLY file tried -- also in a process that has another librccl mapped already
        // (PyTorch's): whoever sets it means that file.  Otherwise the mapped copy first (one RCCL per process), then the usual names.
        const char* envp = getenv("ALGP_RCCL_PATH");
        const bool only_env = envp && *envp;
        const char* names[] = {only_env ? envp : nullptr, only_env || loaded.empty() ? nullptr : loaded.c_str(),
                               only_env ? nullptr : "librccl.so.1", only_env ? nullptr : "librccl.so",
                               only_env ? nullptr : "/opt/rocm/lib/librccl.so.1"};
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

// What a rank contributes to the pick's all-gather, `comm_payload_bytes` per rank:
//   double[0] = its best utility (-inf without a candidate), [1] = that candidate's pool index as a double (exact below
//   2^53; -1: none), [2] = status: 0 fine | 1 the best row still lags behind the committed picks (its utility is only an
//   upper bound: one more refresh round) | >= 2 the ALGP_ERR_* code this rank failed with;
//   bytes 24..31: the candidate's statistic (pv or s) in the context's element type;
//   bytes 32.. : its row of V^T (Npad + MAX_APPEND elements; zero beyond the active columns).
// The row is what makes a remote commit a copy: every rank commits the winner by appending the winner's row to its own
// shard (agent.py:352-354 with the candidates sharded) -- a rank that does not own the winner used to rebuild that row
// from the replicated factor (a kernel row + a 0.42-ms forward substitution + three kernels per pick, on 7 of 8 ranks);
// now it takes the owner's bits from the gather ($ALGP_GATHER_ROWS=0: the 32-byte header only, rows rebuilt).
__global__ void pack_best_kernel(const double* val, const int64_t* pos, const int64_t* cidx, const int* fresh, int npicks,
                                 int status, const int* sticky, double* triple) {
    const int64_t p = pos ? *pos : -1;
    if (status == 0 && sticky && *sticky != 0) status = ALGP_ERR_HIP;     // a one-launch kernel of this rank gave up earlier
    triple[0] = p >= 0 ? *val : -INFINITY;
    triple[1] = p >= 0 ? (double)cidx[p] : -1.0;
    triple[2] = status != 0 ? (double)status : ((p >= 0 && fresh && fresh[p] < npicks) ? 1.0 : 0.0);
}
template <typename T>
__global__ __launch_bounds__(256) void pack_row_kernel(const int64_t* pos, const T* Vt, int64_t ldv, int64_t ncols, int64_t rowlen,
                                                       const T* dstat, char* payload) {
    const int64_t p = pos ? *pos : -1;
    T* row = (T*)(payload + 32);
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < rowlen) row[j] = (p >= 0 && j < ncols) ? Vt[p * ldv + j] : (T)0;
    if (j == 0) *(T*)(payload + 24) = p >= 0 ? dstat[p] : (T)0;
}
// out = (utility, pool index, owning rank, status, first rank with a non-zero status): the first maximum in rank order
// over the ranks that have a candidate -- a NaN utility never wins (the local argmax skips NaN as well), -inf does when
// nothing else is on offer -- and the largest status word (error codes are >= 2, so they outrank "one more round")
__host__ __device__ inline void first_max_reduce(const char* payloads, int64_t stride, int nranks, double* out) {
    double bv = -INFINITY, bi = -1.0, br = -1.0, st = 0.0, bad = -1.0;
    for (int r = 0; r < nranks; ++r) {
        const double* t = (const double*)(payloads + (int64_t)r * stride);
        const double v = t[0], i = t[1], s = t[2];
        if (s != 0.0) {
            if (bad < 0.0) bad = (double)r;
            if (!(s <= st)) st = s;                     // a NaN status counts as a failure too
        }
        // equal utilities: the smaller pool index, whichever rank offers it -- with contiguous shards in rank order that IS
        // the first rank; with any other owner map (algp_comm_set_owners: e.g. site q on rank q mod n) it keeps the pick
        // equal to np.argmax over the candidates in pool order
        if (i >= 0.0 && v == v && (bi < 0.0 || v > bv || (v == bv && i < bi))) { bv = v; bi = i; br = (double)r; }
    }
    out[0] = bv;
    out[1] = bi;
    out[2] = br;
    out[3] = st == st ? st : (double)ALGP_ERR_HIP;
    out[4] = bad;
}
// the RCCL transport and the one-rank case reduce where the payloads are, on the device; the host transport has them in
// host memory when the caller's gather returns and reduces there (pick_exchange)
__global__ void first_max_kernel(const char* payloads, int64_t stride, int nranks, double* out) {
    first_max_reduce(payloads, stride, nranks, out);
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
    return comm_reserve(c);
}

static bool gather_rows_on() {
    static const bool on = env_switch("ALGP_GATHER_ROWS", true);
    return on;
}
// bytes a rank contributes per pick: the 32-byte header, and the row when rows travel (fixed by the train set's size, which
// every rank shares -- the factor is replicated)
size_t comm_payload_bytes(const algp_ctx* c) {
    const bool multi = c->comm || c->host_gather;
    if (!multi || !gather_rows_on()) return 32;
    return 32 + (size_t)round_up((int64_t)((c->Npad + MAX_APPEND) * (int64_t)c->es), 16);
}
// Buffers of the exchange for the current train-set size: [own payload | nranks payloads | 5-double record] on the device,
// pinned staging for the host transport.  Called when a communicator is attached and whenever the train set changes
// (algp_set_train), so that no allocation is left for the middle of a pick.
int comm_reserve(algp_ctx* c) {
    const size_t pb = comm_payload_bytes(c);
    const int nr = c->comm_nranks;
    // a train set that grows by a few rows per planning step crosses a 128-row boundary every few steps: reserve 12.5 % more
    // than this size asks for whenever the buffers have to grow (a re-allocation of the pinned staging costs ~15 ms)
    const size_t need_dev = pb * (size_t)(nr + 1) + 5 * sizeof(double);
    if (!c->commbuf.p || c->commbuf.cap < need_dev) ALGP_TRY(ensure(c, c->commbuf, need_dev + need_dev / 8));
    if (c->host_gather) {
        const size_t need = pb * (size_t)(nr + 1);
        if (c->comm_host_cap < need) {
            if (c->comm_host) hipHostFree(c->comm_host);
            c->comm_host = nullptr;
            c->comm_host_cap = 0;
            const size_t want = need + need / 8;
            if (hipHostMalloc(&c->comm_host, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return fail(c, ALGP_ERR_OOM, "comm: hipHostMalloc(" + std::to_string(want) + ") for the host transport's staging failed");
            }
            c->comm_host_cap = want;
        }
    }
    return ALGP_OK;
}

int comm_init_host(algp_ctx* c, int nranks, int rank, algp_allgather_fn fn, void* user) {
    comm_destroy(c);
    c->host_gather = fn;
    c->host_gather_user = user;
    c->comm_nranks = nranks;
    c->comm_rank = rank;
    return comm_reserve(c);
}

void comm_destroy(algp_ctx* c) {
    if (c->comm) {
        RcclApi* api = rccl_api(nullptr);
        if (api) api->CommDestroy((ncclComm_t)c->comm);
    }
    if (c->comm_host) hipHostFree(c->comm_host);
    c->comm_host = nullptr;
    c->comm_host_cap = 0;
    if (c->rowx_host) hipHostFree(c->rowx_host);
    c->rowx_host = nullptr;
    c->rowx_host_cap = 0;
    c->site_owner.clear();
    c->comm = nullptr;
    c->host_gather = nullptr;
    c->host_gather_user = nullptr;
    c->comm_nranks = 1;
    c->comm_rank = 0;
}

// The exchange of one pick.  (val_dev, pos_dev): the local argmax as the kernels before left it on the device (null: this
// rank has no candidate to offer); status: 0 or the ALGP_ERR_* code this rank failed with while preparing it.  Everything
// is stream-ordered; the single synchronisation (RCCL transport: the read-back of the record; host transport: the payload's
// way down to host memory, where the record is then computed) is the read-back of rec5 = (utility, pool index, owner, status, first rank with a non-zero
// status), identical on every rank.  *winner_payload: where the owner's payload sits in this rank's gather buffer (device).
// No rank-local failure returns before the collective has been issued: a pack launch that fails turns into this rank's
// status word (written from the host); only a failure of the transport itself (the collective call, the copies around the
// caller's gather) returns early -- there is no exchange left to report it through.
template <typename T>
static int pick_exchange(algp_ctx* c, const double* val_dev, const int64_t* pos_dev, const int64_t* cidx_dev, const int* fresh_dev,
                         int npicks, int status, double* rec5, const char** winner_payload) {
    const int nr = c->comm_nranks;
    const bool multi = c->comm || c->host_gather;
    const size_t pb = comm_payload_bytes(c);
    const size_t need = pb * (size_t)(nr + 1) + 5 * sizeof(double);
    if (!c->commbuf.p || c->commbuf.cap < need) {
        // not reserved for this size (a train set that changed without algp_set_train, or a reservation that failed and was
        // ignored): the one allocation that can still precede the collective
        ALGP_TRY(comm_reserve(c));
    }
    char* own = (char*)c->commbuf.p;
    char* all = own + pb;
    double* out = (double*)(all + pb * (size_t)nr);
    const int* sticky = (const int*)((const double*)c->scal.p + SC_STALL);
    hipLaunchKernelGGL(pack_best_kernel, dim3(1), dim3(1), 0, c->stream, val_dev, pos_dev, cidx_dev, fresh_dev, npicks, status,
                       sticky, (double*)own);
    hipError_t pe = hipGetLastError();
    if (pe == hipSuccess && pb > 32) {
        const int64_t rowlen = (int64_t)(pb - 32) / (int64_t)sizeof(T);
        const bool have_row = pos_dev && c->Vt.p && c->solved;
        hipLaunchKernelGGL(pack_row_kernel<T>, dim3((unsigned)((rowlen + 255) / 256)), dim3(256), 0, c->stream,
                           have_row ? pos_dev : (const int64_t*)nullptr, (const T*)c->Vt.p, c->ldv, c->ncols, rowlen,
                           (const T*)c->dstat.p, own);
        pe = hipGetLastError();
    }
    if (c->debug_fail_next_pack) {                              // algp_debug_fail_at(2): as if the pack launch had failed
        pe = hipErrorLaunchFailure;
        c->debug_fail_next_pack = 0;
    }
    if (pe != hipSuccess) {
        // the launch failed: this rank still joins the gather, with the failure as its status word
        const double t[4] = {-INFINITY, -1.0, (double)ALGP_ERR_HIP, 0.0};
        c->err = std::string("greedy: packing the local best failed: ") + hipGetErrorString(pe);
        (void)hipMemcpyAsync(own, t, sizeof(t), hipMemcpyHostToDevice, c->stream);
        (void)hipStreamSynchronize(c->stream);                  // t is a stack array
    }
    const char* gathered = all;
    if (c->comm) {
        RcclApi* api = rccl_api(nullptr);
        if (!api) return fail(c, ALGP_ERR_STATE, "greedy_sharded: the RCCL communicator has no library behind it");
        const ncclResult_t r = api->AllGather(own, all, pb, ncclChar, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess)
            return fail(c, ALGP_ERR_HIP, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed"));
    } else if (c->host_gather) {
        // the caller's transport works on host memory: the payload goes down, the gathered payloads come back up
        char* hs = (char*)c->comm_host;
        char* hr = hs + pb;
        ALGP_HIP(hipMemcpyAsync(hs, own, pb, hipMemcpyDeviceToHost, c->stream));
        ALGP_HIP(hipStreamSynchronize(c->stream));
        c->n_syncs++;
        const int rc = c->host_gather(c->host_gather_user, hs, hr, (int64_t)pb);
        if (rc != 0) return fail(c, ALGP_ERR_HIP, "greedy_sharded: the caller's all-gather returned " + std::to_string(rc));
        // the payloads are in host memory: the reduction of first_max_kernel runs here, and only the winner's payload (its
        // statistic and its row of V^T, what commit_enqueue reads) goes back up -- one row instead of one per rank, no
        // read-back of the record, one synchronisation less per pick
        first_max_reduce(hr, (int64_t)pb, nr, rec5);
        const int owner = (int)rec5[2];
        const bool have = pb > 32 && owner >= 0 && owner < nr;
        if (have) ALGP_HIP(hipMemcpyAsync(all + (size_t)owner * pb, hr + (size_t)owner * pb, pb, hipMemcpyHostToDevice, c->stream));
        if (winner_payload) *winner_payload = have ? all + (size_t)owner * pb : nullptr;
        return ALGP_OK;
    } else {
        gathered = own;                                  // one rank: the same reduction over its own payload
    }
    hipLaunchKernelGGL(first_max_kernel, dim3(1), dim3(1), 0, c->stream, gathered, (int64_t)pb, multi ? nr : 1, out);
    ALGP_HIP(hipGetLastError());
    ALGP_HIP(hipMemcpyAsync(rec5, out, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    if (winner_payload) {
        const int owner = (int)rec5[2];
        *winner_payload = (pb > 32 && owner >= 0 && owner < (multi ? nr : 1)) ? gathered + (size_t)owner * pb : nullptr;
    }
    return ALGP_OK;
}
int comm_pick_exchange(algp_ctx* c, const double* val_dev, const int64_t* pos_dev, const int64_t* cidx_dev,
                       const int* fresh_dev, int npicks, int status, double* rec5, const char** winner_payload) {
    return c->dtype == ALGP_F64 ? pick_exchange<double>(c, val_dev, pos_dev, cidx_dev, fresh_dev, npicks, status, rec5, winner_payload)
                                : pick_exchange<float>(c, val_dev, pos_dev, cidx_dev, fresh_dev, npicks, status, rec5, winner_payload);
}

// ---- the factor update's row exchange (api.hip: exchange_new_rows) ------------------------------------------------------
// The active-learning loop on sharded candidates (agent.py:125-229 with the loop of agent.py:313-354 cut into shards): the
// sites a planning step adds to the train set -- the picks and the mobile readings along the chosen path, agent.py:66-82 --
// are candidates of exactly one rank each, and that rank's row of V^T IS the site's new row of the replicated factor left
// of the tail block.  Two collectives per factor update: a 32-byte agreement word per rank (status, first changed row, train
// size, a hash of the plan -- so that every rank takes the same branch, and a rank that cannot take part says so instead
// of leaving its peers in the second collective), then ONE all-gather of the rows (`cap` rows per rank, cap = the largest
// number any rank owns; the plan is computed identically everywhere from the train set and the owner map).
int comm_agree(algp_ctx* c, const double mine[4], std::vector<double>& all) {
    const int nr = c->comm_nranks;
    all.assign((size_t)nr * 4, 0.0);
    if (c->host_gather) {
        std::vector<double> send(mine, mine + 4);
        const int rc = c->host_gather(c->host_gather_user, send.data(), all.data(), 32);
        if (rc != 0) return fail(c, ALGP_ERR_HIP, "factorize_update: the caller's all-gather returned " + std::to_string(rc));
        return ALGP_OK;
    }
    if (!c->comm) return fail(c, ALGP_ERR_STATE, "factorize_update: no communicator");
    RcclApi* api = rccl_api(nullptr);
    if (!api) return fail(c, ALGP_ERR_STATE, "factorize_update: the RCCL communicator has no library behind it");
    ALGP_TRY(ensure(c, c->commbuf, std::max<size_t>(c->commbuf.cap, 32 * (size_t)(nr + 1))));
    char* own = (char*)c->commbuf.p;
    ALGP_HIP(hipMemcpyAsync(own, mine, 32, hipMemcpyHostToDevice, c->stream));
    const ncclResult_t r = api->AllGather(own, own + 32, 32, ncclChar, (ncclComm_t)c->comm, c->stream);
    if (r != ncclSuccess)
        return fail(c, ALGP_ERR_HIP, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed"));
    ALGP_HIP(hipMemcpyAsync(all.data(), own + 32, 32 * (size_t)nr, hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    return ALGP_OK;
}
// room for [own rows | every rank's rows] on the device (and in pinned memory for the host transport); grows geometrically
int comm_rows_reserve(algp_ctx* c, size_t bytes_per_rank) {
    const size_t need = bytes_per_rank * (size_t)(c->comm_nranks + 1);
    if (!c->rowx.p || c->rowx.cap < need) ALGP_TRY(ensure(c, c->rowx, need + need / 4));
    if (c->host_gather && c->rowx_host_cap < need) {
        if (c->rowx_host) hipHostFree(c->rowx_host);
        c->rowx_host = nullptr;
        c->rowx_host_cap = 0;
        const size_t want = need + need / 4;
        if (hipHostMalloc(&c->rowx_host, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, ALGP_ERR_OOM, "comm: hipHostMalloc(" + std::to_string(want) + ") for the row exchange's staging failed");
        }
        c->rowx_host_cap = want;
    }
    return ALGP_OK;
}
// rowx[0 : bytes) of every rank -> rowx[bytes : bytes * (nranks + 1)) in rank order, stream-ordered on c->stream.
// used_bytes[r] (or null): how much of rank r's slice carries rows -- the host transport stages only that much up to the
// device (the plan gives every rank `cap` rows, a rank that owns fewer sends padding) and takes this rank's own slice from
// the device copy it already holds.
int comm_rows_gather(algp_ctx* c, size_t bytes_per_rank, const size_t* used_bytes) {
    const int nr = c->comm_nranks;
    char* own = (char*)c->rowx.p;
    char* all = own + bytes_per_rank;
    if (c->comm) {
        RcclApi* api = rccl_api(nullptr);
        if (!api) return fail(c, ALGP_ERR_STATE, "factorize_update: the RCCL communicator has no library behind it");
        const ncclResult_t r = api->AllGather(own, all, bytes_per_rank, ncclChar, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess)
            return fail(c, ALGP_ERR_HIP, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed"));
        return ALGP_OK;
    }
    if (!c->host_gather) return fail(c, ALGP_ERR_STATE, "factorize_update: no communicator");
    char* hs = (char*)c->rowx_host;
    char* hr = hs + bytes_per_rank;
    ALGP_HIP(hipMemcpyAsync(hs, own, bytes_per_rank, hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    const int rc = c->host_gather(c->host_gather_user, hs, hr, (int64_t)bytes_per_rank);
    if (rc != 0) return fail(c, ALGP_ERR_HIP, "factorize_update: the caller's all-gather returned " + std::to_string(rc));
    if (!used_bytes) {
        ALGP_HIP(hipMemcpyAsync(all, hr, bytes_per_rank * (size_t)nr, hipMemcpyHostToDevice, c->stream));
        return ALGP_OK;
    }
    for (int r = 0; r < nr; ++r) {
        const size_t ub = std::min(used_bytes[r], bytes_per_rank);
        if (ub == 0) continue;
        if (r == c->comm_rank) ALGP_HIP(hipMemcpyAsync(all + bytes_per_rank * (size_t)r, own, ub, hipMemcpyDeviceToDevice, c->stream));
        else ALGP_HIP(hipMemcpyAsync(all + bytes_per_rank * (size_t)r, hr + bytes_per_rank * (size_t)r, ub, hipMemcpyHostToDevice, c->stream));
    }
    return ALGP_OK;
}

// test hook: first_max_kernel over a caller-made buffer of `nranks` triples (fabricated 8-rank cases on one GPU)
int comm_debug_first_max(algp_ctx* c, const double* triples, int nranks, double* out5) {
    DevBuf tmp;
    ALGP_TRY(ensure(c, tmp, sizeof(double) * (3 * (size_t)nranks + 5)));
    double* all = (double*)tmp.p;
    double* out = all + 3 * nranks;
    int rc = ALGP_OK;
    if (hipMemcpyAsync(all, triples, 3 * sizeof(double) * nranks, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = ALGP_ERR_HIP;
    if (rc == ALGP_OK) {
        hipLaunchKernelGGL(first_max_kernel, dim3(1), dim3(1), 0, c->stream, (const char*)all, (int64_t)(3 * sizeof(double)), nranks, out);
        if (hipGetLastError() != hipSuccess) rc = ALGP_ERR_HIP;
    }
    if (rc == ALGP_OK && hipMemcpyAsync(out5, out, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = ALGP_ERR_HIP;
    hipStreamSynchronize(c->stream);
    hipFree(tmp.p);
    c->dev_bytes -= (int64_t)tmp.cap;
    return rc == ALGP_OK ? ALGP_OK : fail(c, ALGP_ERR_HIP, "debug_first_max: HIP call failed");
}

}  // namespace algp
