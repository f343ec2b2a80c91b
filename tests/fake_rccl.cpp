// TEST DOUBLE, not a product file: the five RCCL entry points libalgp_hip.so binds with dlsym (comm.hip: rccl_api), implemented
// over POSIX shared memory so that SEVERAL ranks of the library can run its RCCL transport -- algp_comm_init +
// algp_greedy_sharded: device buffers, the ncclChar all-gather of the payloads, ONE read-back per pick -- on the ONE GPU a
// session has (real RCCL refuses two ranks on one device).  tests/test_rccl_transport.py builds it with
// hipcc into a temporary directory and points $ALGP_RCCL_PATH at it.  The all-gather is stream-ordered the blunt way:
// synchronise the stream, copy the rank's bytes to its slot, barrier, copy every slot back up, barrier.  Every wait is
// bounded (60 s): a broken protocol fails the test instead of hanging the box.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <mutex>
#include <set>
#include <string>

static std::mutex g_ids_mu;
static std::set<std::string> g_ids_used;                        // "<id>#<rank>" joined by this process (ranks may be threads of one process)

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                       // 0 = ncclSuccess
typedef int ncclDataType_t;                                     // 0 = ncclChar (the only type the library sends)
struct FakeComm;
typedef FakeComm* ncclComm_t;

constexpr int MAX_RANKS = 16;
constexpr size_t SLOT = 4 << 20;                               // per rank: a pick's payload, or a factor update's rows (cap rows of the kept width)
struct Shared {
    int count, gen, calls, used;                                // used: every rank has joined once -- a second join with this id is refused
    char pad[48];
    char slots[MAX_RANKS][SLOT];
};
struct FakeComm {
    Shared* sh;
    int nranks, rank;
    char name[64];
};

static double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static bool barrier(FakeComm* c) {
    Shared* s = c->sh;
    const int g = __atomic_load_n(&s->gen, __ATOMIC_ACQUIRE);
    if (__atomic_fetch_add(&s->count, 1, __ATOMIC_ACQ_REL) == c->nranks - 1) {
        __atomic_store_n(&s->count, 0, __ATOMIC_RELEASE);
        __atomic_fetch_add(&s->gen, 1, __ATOMIC_ACQ_REL);
        return true;
    }
    const double t0 = now_s();
    while (__atomic_load_n(&s->gen, __ATOMIC_ACQUIRE) == g) {
        if (now_s() - t0 > 60.0) return false;
        usleep(20);
    }
    return true;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/algp_fake_rccl_%d_%ld", (int)getpid(), (long)(now_s() * 1e6));
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') return 4;   // ncclInvalidArgument
    // a real ncclUniqueId is SINGLE-USE (the root's bootstrap listener is gone after the first init): refuse a second join
    // of this process with the same id, and a join of a segment every rank has already joined once
    {
        std::lock_guard<std::mutex> lk(g_ids_mu);
        const std::string key = std::string(id.internal) + "#" + std::to_string(rank);
        if (!g_ids_used.insert(key).second) return 3;
    }
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;                                       // ncclSystemError
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return 2; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    FakeComm* c = new FakeComm;
    c->sh = (Shared*)p;
    c->nranks = nranks;
    c->rank = rank;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->name[sizeof(c->name) - 1] = 0;
    if (__atomic_load_n(&c->sh->used, __ATOMIC_ACQUIRE)) { munmap(p, sizeof(Shared)); delete c; return 3; }
    if (!barrier(c)) return 3;                                  // ncclInternalError: a rank never arrived
    if (!barrier(c)) return 3;                                  // (everybody has checked `used` before anybody sets it)
    __atomic_store_n(&c->sh->used, 1, __ATOMIC_RELEASE);
    *comm = c;
    return 0;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dtype, ncclComm_t c, hipStream_t stream) {
    if (dtype != 0 || count > SLOT) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;   // ncclUnhandledCudaError
    if (hipMemcpy(c->sh->slots[c->rank], send, count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!barrier(c)) return 3;
    for (int r = 0; r < c->nranks; ++r)
        if (hipMemcpy((char*)recv + (size_t)r * count, c->sh->slots[r], count, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (c->rank == 0) __atomic_fetch_add(&c->sh->calls, 1, __ATOMIC_RELAXED);
    if (!barrier(c)) return 3;                                  // nobody refills a slot before everybody has read it
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    munmap(c->sh, sizeof(Shared));
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case 0: return "no error";
        case 1: return "unhandled HIP error (test double)";
        case 2: return "system error (test double: shared memory)";
        case 3: return "internal error (test double: a rank did not reach the barrier within 60 s, or a unique id was used twice)";
        case 4: return "invalid argument (test double)";
    }
    return "unknown";
}

}  // extern "C"
