// chol_dag.hip -- the Cholesky factorisation S = L L^T as ONE dependency-driven launch.
//
// Replaces the reference's LU inside np.linalg.inv / np.linalg.slogdet (utils.py:193, 300).  The blocked
// right-looking factorisation of potrf.hip is a chain of short launches (diagonal block on one workgroup,
// K = 128 panel launches, one K = 512 update per 512 columns): at N = 10 000 the chip is mostly idle
// (19.7 TFLOP/s, round 1).  Here the same arithmetic is a list of tile tasks over 128 x 128 tiles
//     CHAIN(k)          L_kk = chol(S_kk), X_kk = inv(L_kk) (diag.h) by the team's leader (the first workgroup to arrive,
//                       its CU to itself), then the two tile products the next diagonal block waits for, TRSM(k+1,k)
//                       and UPD(k+1,k+1,k,k+1), in 32-row strips by the team's four helpers (see the kernel)
//     TRSM(i,k)         L_ik = S_ik X_kk^T                                   (MFMA tile product, K = 128)
//     UPD(i,j,k0,k1)    S_ij -= L_i,[k0,k1) L_j,[k0,k1)^T                    (MFMA tile product, K = 128 (k1-k0))
//     TU(i,k)           TRSM(i,k), then UPD(i,k+1,k,k+1) by the same workgroup: every row's step-to-step recurrence
// executed by persistent workgroups that draw tasks from ONE ordered list with an atomic ticket.  A task waits
// (bounded spin on per-tile version counters) until its inputs are final; because the list is a topological
// order and tickets are handed out in list order, every task's producers are already held by running
// workgroups: no deadlock whatever the residency, dispatch order or XCD placement.
// The order is a list schedule computed on the host (critical-path priorities, simulated on as many workers
// as the chip holds workgroups), so look-ahead happens by itself: the diagonal chain runs ahead while the
// K = 512 updates of older block columns fill every other workgroup.
// Updates of a tile by the W + (0..3) columns just left of it are single K = 128 steps (they sit on or near
// the critical path); everything older is applied in batches -- K = 512 next to that window, K = 1024 and K = 2048
// further back (dag_build_schedule): one read + one write of the tile and one ticket / wait / publish per batch
// keeps the update MFMA-bound instead of bound by the tile's own traffic.  Each tile receives its updates in
// ascending k, so the result is bitwise reproducible and independent of the schedule.
// Hand-off between workgroups (MI355X: per-XCD L2s that are not coherent with each other, per-CU L1s): producer =
// every tile element is stored WRITE-THROUGH (sc1: nothing stays dirty in the XCD's L2, so no L2 write-back is needed
// before the flag -- a release fence writes back every dirty line of the XCD, and with 64 workgroups per XCD writing
// tiles that was several microseconds per task), every wave drains its stores, workgroup barrier, relaxed agent-scope
// store of the tile's version; consumer = one wave polls (relaxed agent-scope loads), one agent-scope acquire (this
// CU's L1), drain, barrier, plain loads.
#include "common.h"
#include "mfma.h"
#include "diag.h"
#include <algorithm>
#include <queue>
#include <stdlib.h>
#include <stdio.h>

namespace algp {

enum { DAG_CHAIN = 0, DAG_TRSM = 1, DAG_UPD = 2, DAG_TU = 3 };
#ifndef ALGP_DAG_STRIP_ROWS
#define ALGP_DAG_STRIP_ROWS 32
#endif
constexpr int DAG_SR = ALGP_DAG_STRIP_ROWS;                   // rows of a strip: 16 or 32
constexpr int DAG_STRIPS = 128 / DAG_SR;                      // row strips of a 128-row tile product on the chain
constexpr int DAG_TEAM = 1 + DAG_STRIPS;                      // workgroups on the diagonal chain: leader + one helper per strip
constexpr int DAG_CTRL = 32;                                  // control words
struct DagTask {
    int type, i, j, kk;                                        // kk = (k0 << 16) | k1
};                                                             // DAG_TU: TRSM(i,j) and then UPD(i,j+1,j) by the same workgroup

template <typename T>
struct DagArgs {
    T* L;
    int64_t ld;
    T* invD;
    T* P;                                                      // the row panel below the factor (tile rows nt .. nt+mt-1), or null
    int64_t ldp;
    int no_team;                                               // solve-only list: L is final, every workgroup draws tickets
    const DagTask* tasks;
    int ntasks, nt;
    int* ver;                                                  // (nt + mt) x nt tile versions (number of column steps applied)
    int* ctrl;                                                 // [0] ticket, [1] abort code, [3] arrival order, [8..8+DAG_TEAM) the team's CUs
    int* cnt;                                                  // per column step: strips published of the five team products
    double* pivots;                                            // the nt * 128 pivots d_j (their logarithms are summed by dag_finish_kernel)
    int* info;
    unsigned long long spin_limit;                             // ticks (100 MHz) a wait may spin before it aborts the launch
    int skip_publish;                                          // test hook: the bulk task with this ticket never publishes (-1: none)
    unsigned long long* stats;                                 // or null: [0] ticks (100 MHz) spent in UPD products, summed over
                                                               // workgroups, [1] their K = 128 steps, [2] ticks in TRSM products, [3] their count
};

// Diagnostic builds only (tools/dag_test.hip): per-workgroup progress words the host can read while the launch runs.
#ifdef ALGP_DAG_DEBUG
__device__ int g_dag_dbg[4 * 1024];
#define DAG_DBG(slot, val) do { if (threadIdx.x == 0 && blockIdx.x < 1024) __hip_atomic_store(&g_dag_dbg[4 * blockIdx.x + (slot)], (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
// per task: 100 MHz stamps at ticket / inputs ready / computed / published, and the workgroup + XCD that ran it
__device__ unsigned long long g_dag_trace[4 * 262144];
__device__ unsigned long long g_dag_chain[16 * 1024];         // per chain step: stamps inside the fused task
#define DAG_CHAINT(k, slot) do { if (threadIdx.x == 0 && (k) < 1024) g_dag_chain[16 * (k) + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ int g_dag_who[262144];
#define DAG_TRACE(t, slot) do { if (threadIdx.x == 0 && (t) < 262144) g_dag_trace[4 * (t) + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long g_dag_phase[8];                 // K = 128 tile products: ticks until the C loads are issued / main loop / until the stores are issued (x 10^6) + drain, count
#define DAG_PHASE(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define DAG_WHO(t) do { if (threadIdx.x == 0 && (t) < 262144) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_dag_who[t] = (int)((blockIdx.x << 4) | (xcc & 15)); } } while (0)
#else
#define DAG_DBG(slot, val) do { } while (0)
#define DAG_TRACE(t, slot) do { } while (0)
#define DAG_WHO(t) do { } while (0)
#define DAG_CHAINT(k, slot) do { } while (0)
#define DAG_PHASE(var) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// ---- 128 x 128 tile product acc += A B^T, A and B row-major with k contiguous: the four-stage LDS-DMA
// pipeline of gemm.hip (three 64-byte k-tiles in flight while one is multiplied) as a device routine ----
template <typename T>
__device__ __forceinline__ void tile_mainloop(char* smem, const T* A0, int64_t lda, const T* B0, int64_t ldb, int nkt,
                                              typename MF<T>::acc_t (&acc)[4][4]) {
    using F = MF<T>;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC;
    constexpr int BK = 4 * EPC;
    constexpr int NST = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int srow = lane >> 2;
    // chunk XOR ((row >> 2) & 2): conflict-free for the four 16-lane groups a ds_read_b128 is served in (gemm.hip)
    const int schunk = (lane & 3) ^ ((lane >> 4) & 2);
    const T* Ag = A0 + (int64_t)(32 * wave + srow) * lda + schunk * EPC;
    const T* Bg = B0 + (int64_t)(32 * wave + srow) * ldb + schunk * EPC;
    auto stage = [&](int st, int kt) {
        char* As = smem + st * 16384 + wave * 2048;
        char* Bs = As + 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)(Ag + (int64_t)(16 * i) * lda + (int64_t)kt * BK),
                                             (lds_vp)(As + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(Bg + (int64_t)(16 * i) * ldb + (int64_t)kt * BK),
                                             (lds_vp)(Bs + i * 1024), 16, 0, 0);
        }
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int coff = ((fg ^ ((fr >> 2) & 2)) << 4);
    const int aoff = (wr * 64 + fr) * 64 + coff;
    const int boff = (wc * 64 + fr) * 64 + coff;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nkt) stage(t, t);
    auto fread = [&](int st, chunk_t (&a)[4], chunk_t (&b)[4]) {
        const char* As = smem + st * 16384;
        const char* Bs = As + 8192;
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const chunk_t*>(As + aoff + t * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const chunk_t*>(Bs + boff + t * 1024);
    };
    auto fmac = [&](const chunk_t (&a)[4], const chunk_t (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < EPC; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = F::mfma(a[i][e], b[j][e], acc[i][j]);
    };
    auto arrive = [&](int younger) {
        if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F78);       // vmcnt(8)
        else if (younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F74);  // vmcnt(4)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        arrive(nkt - 1 - kt);
        chunk_t a[4], b[4];
        fread(st, a, b);
        if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, kt + NST - 1);
        fmac(a, b);
        st = (st + 1 == NST) ? 0 : st + 1;
    }
}


// ---- the three building blocks of a task, called by all 256 threads ----
// Wait until every listed word has reached its target: lane l of wave 0 polls dependency l (addr / want are per-lane
// values; addr == nullptr: nothing to wait for in this lane).  Returns false when the launch is being aborted (a spin
// ran into its time limit here or elsewhere); the caller then leaves its task loop.  `ticket`: what the abort word
// records as ticket + 1 (bulk tasks: their ticket >= 0; the team's waits of column step k pass -2 - k: never zero).
template <typename T>
__device__ __forceinline__ bool dag_wait(const DagArgs<T>& g, int ticket, const int* addr, int want, int* s_ok) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) {
        const bool mine = addr != nullptr;
        const int* a = mine ? addr : g.ctrl;
        bool ok = false;
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 0;; ++spins) {
            const int v = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = __all(!mine || v >= want);
            if (ok) break;
            if (__hip_atomic_load(&g.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if ((spins & 255) == 255 && __builtin_amdgcn_s_memrealtime() - t_start > g.spin_limit) {   // 2 s at 100 MHz by default
                if (lane == 0) atomicCAS(&g.ctrl[1], 0, ticket + 1);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (ok) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) *s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = __builtin_amdgcn_readfirstlane(*s_ok) != 0;
    __syncthreads();                                           // s_ok may be rewritten by the next wait
    return ok;
}
template <typename T>
__device__ __forceinline__ const int* dag_ver(const DagArgs<T>& g, int i, int j) { return g.ver + (int64_t)i * g.nt + j; }
// Tile row i: a block row of the factor (i < nt) or of the panel below it (the candidates' V^T, or the identity that
// becomes L^-T); both are row-major with their own leading dimension.
template <typename T>
__device__ __forceinline__ T* dag_row(const DagArgs<T>& g, int i, int64_t& ldr) {
    if (i < g.nt) { ldr = g.ld; return g.L + (int64_t)i * 128 * g.ld; }
    ldr = g.ldp;
    return g.P + (int64_t)(i - g.nt) * 128 * g.ldp;
}
// Publish tile (i, j) at version `ver`: every wave's (write-through) stores drained, barrier, version store.
template <typename T>
__device__ __forceinline__ void dag_publish(const DagArgs<T>& g, int i, int j, int ver) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) {                                           // a wave-uniform branch (see the note at the ticket)
#if !ALGP_DAG_WT
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (lane == 0)
            __hip_atomic_store(g.ver + (int64_t)i * g.nt + j, ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// UPD: tile (i, j) -= L_i,[k0,k1) L_j,[k0,k1)^T.  TRSM (upd == false): tile (i, j) <- tile (i, j) X_jj^T, in place.
template <typename T>
__device__ __forceinline__ void dag_tile_op(const DagArgs<T>& g, char* smem, bool upd, int ti, int tj, int k0, int k1) {
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    const int lane = threadIdx.x & 63, fr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int64_t ldr;
    T* Ri = dag_row(g, ti, ldr);                               // tile row ti (factor or panel)
    T* Cij = Ri + (int64_t)tj * 128;
    DAG_PHASE(ph0);
    acc_t acc[4][4];
    const T* A0;
    const T* B0;
    int64_t ldb;
    int nkt;
    if (upd) {
        // acc = -C, then acc += A B^T, then C = -acc
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] = -Cij[gi * ldr + wc * 64 + j * 16 + fr];
            }
        A0 = Ri + (int64_t)k0 * 128;
        B0 = g.L + (int64_t)tj * 128 * g.ld + (int64_t)k0 * 128;
        ldb = g.ld;
        nkt = (k1 - k0) * (128 / (4 * F::EPC));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;
        A0 = Cij;                                              // in place: every read of the tile precedes the stores
        B0 = g.invD + (int64_t)tj * 128 * 128;
        ldb = 128;
        nkt = 128 / (4 * F::EPC);
    }
    DAG_PHASE(ph1);
    tile_mainloop<T>(smem, A0, ldr, B0, ldb, nkt, acc);
    DAG_PHASE(ph2);
    const T sgn = upd ? (T)-1 : (T)1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) st_wt(&Cij[gi * ldr + wc * 64 + j * 16 + fr], sgn * acc[i][j][r]);
        }
#ifdef ALGP_DAG_DEBUG
    // Where a K = 128 product spends its time (tools/dag_test.hip prints the means): "issuing" the 64 loads / 64 stores
    // of a lane takes 3-5 us each, and that is the memory system's back-pressure, not the instruction count -- with
    // the tile moved in 16-byte pieces (operands of the MFMA exchanged so that a lane holds row pieces) the stores took
    // as long (fp32) or twice as long (fp64: half-line pieces) and the loads, no longer overlapped, 12-15 us.
    DAG_PHASE(ph3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DAG_PHASE(ph4);
    if (threadIdx.x == 0 && nkt == 128 / (4 * F::EPC)) {
        atomicAdd(&g_dag_phase[upd ? 0 : 4], ph1 - ph0);
        atomicAdd(&g_dag_phase[upd ? 1 : 5], ph2 - ph1);
        atomicAdd(&g_dag_phase[upd ? 2 : 6], (ph3 - ph2) * 1000000ull + (ph4 - ph3));
        atomicAdd(&g_dag_phase[upd ? 3 : 7], 1ull);
    }
#endif
}

// One of `parts` workgroups that share a tile publishes its part: drained as in dag_publish, then an agent-scope add
// to the tile's arrival counter; the workgroup whose add comes last publishes the tile's version for everybody else.
template <typename T>
__device__ __forceinline__ void dag_publish_part(const DagArgs<T>& g, int* counter, int parts, int i, int j, int ver) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) {
#if !ALGP_DAG_WT
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (lane == 0) {
            const int before = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == parts - 1)
                __hip_atomic_store(g.ver + (int64_t)i * g.nt + j, ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- SR x 128 strip product acc += A_strip B^T (A_strip: SR = 16 or 32 rows, B: 128 rows, both k-contiguous): the
// chain's K = 128 products cut into 128 / SR row strips, one per helper workgroup.  Same LDS-DMA pipeline as
// tile_mainloop; a stage is the strip of A (32 rows: 4 KB, wave w stages rows 8w..8w+7 with its lower 32 lanes, the
// upper lanes' copies land in the unused half of the wave's 1 KB; 16 rows: 1 KB, all of it by wave 0) + 8 KB of B;
// wave w owns output columns 32w..32w+31 (SR / 16 x 2 MFMA tiles).  (Sixteen-row strips on eight helpers,
// -DALGP_DAG_STRIP_ROWS=16, were measured in round 4: 7.21-7.24 vs 7.26-7.28 ms in fp64, 4.53-4.55 vs 4.42-4.59 ms in
// fp32 at N = 10 000 -- within the spread.  So was a one-shot form of the fp32 strip product -- all fragments of A and B
// loaded straight into registers, one wait, no LDS, no k-loop: 5.3 + 5.4 us per step for the two products against 4.4 + 4.5
// by the in-kernel stamps.  A strip product on the chain is the latency of fetching tiles another CU has just written
// through to memory, not pipeline structure.)
template <typename T, int SR>
__device__ __forceinline__ void strip_mainloop(char* smem, const T* A0, int64_t lda, const T* B0, int64_t ldb, int nkt,
                                               typename MF<T>::acc_t (&acc)[SR / 16][2]) {
    using F = MF<T>;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC;
    constexpr int BK = 4 * EPC;
    constexpr int NST = 4, STB = 12288, RT = SR / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = lane >> 2;
    const int bchunk = (lane & 3) ^ ((lane >> 4) & 2);
    const int arow = SR == 32 ? 8 * wave + (srow & 7) : srow;
    const int achunk = (lane & 3) ^ ((arow >> 2) & 2);
    const T* Ag = A0 + (int64_t)arow * lda + achunk * EPC;
    const T* Bg = B0 + (int64_t)(32 * wave + srow) * ldb + bchunk * EPC;
    const bool stages_a = SR == 32 || wave == 0;                   // wave-uniform
    auto stage = [&](int st, int kt) {
        char* As = smem + st * STB + (SR == 32 ? wave * 1024 : 0);
        char* Bs = smem + st * STB + 4096 + wave * 2048;
        if (stages_a) __builtin_amdgcn_global_load_lds((glb_vp)(Ag + (int64_t)kt * BK), (lds_vp)As, 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(Bg + (int64_t)(16 * i) * ldb + (int64_t)kt * BK),
                                             (lds_vp)(Bs + i * 1024), 16, 0, 0);
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int coff = ((fg ^ ((fr >> 2) & 2)) << 4);
    // 32 rows: A row 16 i + fr sits in the 1 KB of wave (2 i + (fr >> 3)), local row fr & 7; 16 rows: row fr of the one KB
    const int aoff = SR == 32 ? (fr >> 3) * 1024 + (fr & 7) * 64 + coff : fr * 64 + coff;
    const int boff = 4096 + (32 * wave + fr) * 64 + coff;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nkt) stage(t, t);
    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int younger = nkt - 1 - kt;
        if (stages_a) {                                             // three DMA per stage in flight from this wave
            if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F76);       // vmcnt(6)
            else if (younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F73);  // vmcnt(3)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        } else {                                                    // two
            if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F74);       // vmcnt(4)
            else if (younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F72);  // vmcnt(2)
            else __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        chunk_t a[RT], b[2];
        const char* base = smem + st * STB;
#pragma unroll
        for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const chunk_t*>(base + aoff + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const chunk_t*>(base + boff + j * 1024);
        if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, kt + NST - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < EPC; ++e)
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][j] = F::mfma(a[i][e], b[j][e], acc[i][j]);
        st = (st + 1 == NST) ? 0 : st + 1;
    }
}
// strip `part` (rows SR part .. SR part + SR - 1) of: UPD  tile (ti, tj) -= L_(ti,k) L_(tj,k)^T
//                                                      TRSM tile (ti, k) <- tile (ti, k) X_kk^T  (in place; tj unused)
template <typename T>
__device__ __forceinline__ void dag_strip_op(const DagArgs<T>& g, char* smem, bool upd, int part, int ti, int tj, int k) {
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    constexpr int SR = DAG_SR, RT = SR / 16;
    const int lane = threadIdx.x & 63, fr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    T* Lik = g.L + ((int64_t)ti * 128 + SR * part) * g.ld + (int64_t)k * 128;      // the strip of tile (ti, k)
    T* Out = upd ? g.L + ((int64_t)ti * 128 + SR * part) * g.ld + (int64_t)tj * 128 : Lik;
    acc_t acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = i * 16 + F::row_of(lane, r);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][r] = upd ? -Out[gi * g.ld + wave * 32 + j * 16 + fr] : (T)0;
        }
    const T* B0 = upd ? g.L + (int64_t)tj * 128 * g.ld + (int64_t)k * 128 : g.invD + (int64_t)k * 128 * 128;
    strip_mainloop<T, SR>(smem, Lik, g.ld, B0, upd ? g.ld : 128, 128 / (4 * F::EPC), acc);
    const T sgn = upd ? (T)-1 : (T)1;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = i * 16 + F::row_of(lane, r);
#pragma unroll
            for (int j = 0; j < 2; ++j) st_wt(&Out[gi * g.ld + wave * 32 + j * 16 + fr], sgn * acc[i][j][r]);
        }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void chol_dag_kernel(DagArgs<T> g) {
    __shared__ __attribute__((aligned(1024))) union {
        DiagShared<T> diag;
        char gemm[4 * 16384];
    } sm;
    __shared__ int s_ticket, s_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nt = g.nt;

    // ---- roles: the first DAG_TEAM workgroups to arrive are the chain team.  The leader (first) factors every diagonal
    // block.  The DAG_STRIPS helpers cut the K = 128 products around the diagonal into row strips, one per helper:
    // the two the next diagonal block waits for, TRSM(k+1,k) and UPD(k+1,k+1,k), and -- while the leader factors
    // that block -- the three that feed the chain's next step, TRSM(k+2,k), UPD(k+2,k+1,k), UPD(k+2,k+2,k) (left to
    // bulk workgroups these arrived 10-30 us late at every step).  Each member publishes which CU it sits on and the
    // workgroup that shares that CU retires after its current task: with the CU to themselves the leaf arithmetic
    // and the products run twice as fast (diagonal block 87 -> 44 us), and a few idle slots of 512 cost the bulk nothing.
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int cu_key = (int)((((hwid >> 8) & 0xff) | ((xcc & 0xf) << 8)) + 1);       // (se, sh, cu, xcc), never 0
    if (tid == 0) s_ticket = atomicAdd(&g.ctrl[3], 1);
    __syncthreads();
    const int role = __builtin_amdgcn_readfirstlane(s_ticket);
    __syncthreads();
    if (role < DAG_TEAM && !g.no_team) {
        if (tid == 0) __hip_atomic_store(&g.ctrl[8 + role], cu_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_setprio(3);
        if (role == 0) {
            for (int k = 0; k < nt; ++k) {
                if (!dag_wait<T>(g, -2 - k, lane == 0 ? dag_ver(g, k, k) : nullptr, k, &s_ok)) break;
                DAG_CHAINT(k, 0);
                T* Akk = g.L + (int64_t)k * 128 * g.ld + (int64_t)k * 128;
                diag128_factor<T, true>(sm.diag, Akk, g.ld, g.invD + (int64_t)k * 128 * 128, nullptr, false, g.info,
                                        (int64_t)k * 128, g.pivots + (int64_t)k * 128);
                DAG_CHAINT(k, 1);
                dag_publish<T>(g, k, k, k + 1);
                DAG_CHAINT(k, 2);
            }
            return;
        }
        const int part = role - 1;
        const bool stamp = role == 1;
        for (int k = 0; k + 1 < nt; ++k) {
            int* cnt = g.cnt + 5 * k;
            // L_(k+1)k = S_(k+1)k X_kk^T: needs X_kk and the tile updated k times
            if (!dag_wait<T>(g, -2 - k, lane == 0 ? dag_ver(g, k, k) : (lane == 1 ? dag_ver(g, k + 1, k) : nullptr),
                             lane == 0 ? k + 1 : k, &s_ok)) break;
            if (stamp) DAG_CHAINT(k, 3);
            dag_strip_op<T>(g, sm.gemm, false, part, k + 1, k, k);
            if (stamp) DAG_CHAINT(k, 4);
            dag_publish_part<T>(g, cnt + 0, DAG_STRIPS, k + 1, k, k + 1);
            if (stamp) DAG_CHAINT(k, 5);
            // S_(k+1)(k+1) -= L_(k+1)k L_(k+1)k^T: needs every strip of L_(k+1)k
            if (!dag_wait<T>(g, -2 - k, lane == 0 ? cnt + 0 : (lane == 1 ? dag_ver(g, k + 1, k + 1) : nullptr),
                             lane == 0 ? DAG_STRIPS : k, &s_ok)) break;
            if (stamp) DAG_CHAINT(k, 6);
            dag_strip_op<T>(g, sm.gemm, true, part, k + 1, k + 1, k);
            if (stamp) DAG_CHAINT(k, 7);
            dag_publish_part<T>(g, cnt + 1, DAG_STRIPS, k + 1, k + 1, k + 1);
            if (stamp) DAG_CHAINT(k, 8);
            if (k + 2 >= nt) continue;
            // row k+2, in the shadow of the next diagonal block: L_(k+2)k, then its updates of (k+2,k+1) and (k+2,k+2)
            if (!dag_wait<T>(g, -2 - k, lane == 0 ? dag_ver(g, k + 2, k) : nullptr, k, &s_ok)) break;
            dag_strip_op<T>(g, sm.gemm, false, part, k + 2, k, k);
            dag_publish_part<T>(g, cnt + 2, DAG_STRIPS, k + 2, k, k + 1);
            if (stamp) DAG_CHAINT(k, 9);
            if (!dag_wait<T>(g, -2 - k, lane == 0 ? cnt + 2 : (lane == 1 ? dag_ver(g, k + 2, k + 1) : nullptr),
                             lane == 0 ? DAG_STRIPS : k, &s_ok)) break;
            dag_strip_op<T>(g, sm.gemm, true, part, k + 2, k + 1, k);
            dag_publish_part<T>(g, cnt + 3, DAG_STRIPS, k + 2, k + 1, k + 1);
            if (stamp) DAG_CHAINT(k, 10);
            if (!dag_wait<T>(g, -2 - k, lane == 0 ? dag_ver(g, k + 2, k + 2) : nullptr, k, &s_ok)) break;
            dag_strip_op<T>(g, sm.gemm, true, part, k + 2, k + 2, k);
            dag_publish_part<T>(g, cnt + 4, DAG_STRIPS, k + 2, k + 2, k + 1);
            if (stamp) DAG_CHAINT(k, 11);
        }
        return;
    }

    // ---- everybody else draws tickets from the list ----
    unsigned long long st_upd = 0, st_trsm = 0;
    unsigned st_steps = 0, st_ntrsm = 0;
    for (;;) {
        if (tid < 64) {
            // a workgroup that shares its CU with a team member retires
            bool beside = lane < DAG_TEAM && __hip_atomic_load(&g.ctrl[8 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cu_key;
            beside = __any(beside);
            if (tid == 0) s_ticket = beside ? 0x7fffffff : atomicAdd(&g.ctrl[0], 1);
        }
        __syncthreads();                                       // also: every wave is done with the previous task's LDS
        // wave-uniform in fact, and told so to the compiler: with a (formally) divergent loop exit hipcc lets the
        // lanes that stay in the loop run ahead into the next iteration's barrier before lane 0 has published
        const int t = __builtin_amdgcn_readfirstlane(s_ticket);
        DAG_DBG(0, t);
        DAG_DBG(1, 1);
        if (t >= g.ntasks) break;
        DAG_TRACE(t, 0);
        DAG_WHO(t);
        const DagTask task = g.tasks[t];
        const int type = task.type, ti = task.i, tj = task.j, k0 = task.kk >> 16, k1 = task.kk & 0xffff;

        if (type == DAG_TU) {
            // every row's recurrence from one column step to the next: L_ik = S_ik X_kk^T, then S_i,k+1 -= L_ik L_k+1,k^T
            // (TRSM(i,k+1) needs it).  As two tasks the pair cost two tickets and a hand-off through HBM per step and
            // was the longest path of the whole graph (76 x 78 us for the last row); one workgroup does both, the
            // second product reading the tile it has just written.
            const int k = tj;
            if (!dag_wait<T>(g, t, lane == 0 ? dag_ver(g, ti, k) : (lane == 1 ? dag_ver(g, k, k) : nullptr),
                             lane == 0 ? k : k + 1, &s_ok)) break;
            DAG_TRACE(t, 1);
            unsigned long long tick0 = __builtin_amdgcn_s_memrealtime();
            dag_tile_op<T>(g, sm.gemm, false, ti, k, k, k + 1);
            st_trsm += __builtin_amdgcn_s_memrealtime() - tick0;
            ++st_ntrsm;
            if (t != g.skip_publish) dag_publish<T>(g, ti, k, k + 1);
            if (!dag_wait<T>(g, t, lane == 0 ? dag_ver(g, ti, k + 1) : (lane == 1 ? dag_ver(g, k + 1, k) : nullptr),
                             lane == 0 ? k : k + 1, &s_ok)) break;
            tick0 = __builtin_amdgcn_s_memrealtime();
            dag_tile_op<T>(g, sm.gemm, true, ti, k + 1, k, k + 1);
            st_upd += __builtin_amdgcn_s_memrealtime() - tick0;
            ++st_steps;
            DAG_TRACE(t, 2);
            if (t != g.skip_publish) dag_publish<T>(g, ti, k + 1, k + 1);
            DAG_TRACE(t, 3);
            continue;
        }
        // ---- wait for the inputs: lane l polls dependency l ----
        int ndeps, di = ti, dj = tj, want = 0;
        if (type == DAG_TRSM) {
            ndeps = 2;
            if (lane == 0) want = tj;                          // tile (i,k) updated k times
            else { di = tj; dj = tj; want = tj + 1; }          // diagonal block k factored
        } else {
            const int n = k1 - k0;
            ndeps = 1 + (ti == tj ? n : 2 * n);
            if (lane == 0) want = k0;
            else if (lane <= n) { dj = k0 + lane - 1; want = dj + 1; }
            else { di = tj; dj = k0 + lane - n - 1; want = dj + 1; }
        }
        if (!dag_wait<T>(g, t, lane < ndeps ? dag_ver(g, di, dj) : nullptr, want, &s_ok)) break;
        DAG_DBG(1, 2);
        DAG_TRACE(t, 1);
        const unsigned long long tick0 = __builtin_amdgcn_s_memrealtime();
        dag_tile_op<T>(g, sm.gemm, type == DAG_UPD, ti, tj, k0, k1);
        const unsigned long long ticks = __builtin_amdgcn_s_memrealtime() - tick0;
        if (type == DAG_UPD) { st_upd += ticks; st_steps += (unsigned)(k1 - k0); }
        else { st_trsm += ticks; ++st_ntrsm; }
        DAG_DBG(1, 3);
        DAG_TRACE(t, 2);
        if (t != g.skip_publish) dag_publish<T>(g, ti, tj, (type == DAG_UPD) ? k1 : tj + 1);
        DAG_DBG(1, 4);
        DAG_TRACE(t, 3);
    }
    if (g.stats && tid == 0) {
        atomicAdd(g.stats + 0, st_upd);
        atomicAdd(g.stats + 1, (unsigned long long)st_steps);
        atomicAdd(g.stats + 2, st_trsm);
        atomicAdd(g.stats + 3, (unsigned long long)st_ntrsm);
    }
    DAG_DBG(1, 5);
}

// log det = sum of the logarithms of the n pivots, in a fixed order (thread t takes pivots t, t + 256, ...; the 256 partial
// sums are added in thread order): the same bits in every run.  Abort code -> info.
__global__ __launch_bounds__(256) void dag_finish_kernel(const double* pivots, int n, const int* ctrl, double* logdet_acc, int* info) {
    __shared__ double part[256];
    double s = 0;
    for (int j = threadIdx.x; j < n; j += 256) s += log(pivots[j]);
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int t = 0; t < 256; ++t) tot += part[t];
        if (logdet_acc) *logdet_acc += tot;
        if (ctrl[1] != 0) *info = -2147483647 - 1;             // stalled: reported as a HIP-level failure by the caller
    }
}

// ---------------------------------------------------------------------------------------------
// Host side: the task list in list-schedule order.
// ---------------------------------------------------------------------------------------------
struct DagSchedule {
    int64_t nt = 0;
    int window = 0;
    std::vector<DagTask> tasks;
    float makespan = 0;                                        // of the simulation, microseconds
};

// What the list covers besides the nt x nt factor: `mt` more tile rows below it (the "panel") that are carried along as
// extra block rows of the factorisation -- TRSM / UPD tasks without a diagonal -- so that P <- P L^-T comes out of the
// same launch.  mode 1: a dense panel (the candidates' B^T -> V^T, reference utils.py:300-301); mode 2: the identity
// (mt == nt; tile row e is zero left of column e and stays so: only tiles (e, j >= e) exist) -> L^-T for the MLL
// gradient (reference models.py:147-148).  solve_only: the factor is final already, the list holds the panel's tasks only.
struct DagShape {
    int nt = 0, mt = 0, mode = 0;
    bool solve_only = false;
    // mode 1, round 6: the first `pshort` panel tile rows leave their LAST column tile out of the list -- a train set that ends a
    // few columns into that tile has them solved by the tail kernel afterwards (api_candidates.hip: fit_and_solve), the list
    // does not pay 79 K-steps per tile row for 16 columns.  (The tile row that carries y - ybar keeps every column: z is whole.)
    int pshort = 0;
    int pcols(int e) const { return (mode == 1 && e < pshort) ? nt - 1 : nt; }
    // mode 2: tile rows behind the nt identity rows are dense (mt = nt + 1 in algp_fit_step: the row that carries y - ybar -> z)
    int pstart(int e) const { return (mode == 2 && e < nt) ? e : 0; }      // first non-zero tile column of panel row e
    bool operator==(const DagShape& o) const { return nt == o.nt && mt == o.mt && mode == o.mode && solve_only == o.solve_only && pshort == o.pshort; }
};

constexpr int DAG_FAR8 = 4, DAG_FAR16 = 16;                    // see the batches in dag_build_schedule
static int batched_until(int j, int W) {                       // columns [0, kf) of tile column j arrive in batches of 4
    if (j < W) return 0;
    return 4 * ((j - W) / 4);
}

void dag_build_schedule(const DagShape& shape, int W, int workers, DagSchedule& out) {
    const int nt = shape.nt, mt = shape.mt, R = nt + mt;
    const bool solve_only = shape.solve_only;
    struct Node {
        DagTask t;
        float dur, prio;
        int npred;
        bool on_chain = false;
        std::vector<int> succ;
        std::vector<int> succ_start;                                        // nodes that may start once this one HAS STARTED
    };
    std::vector<Node> nodes;
    std::vector<int> last_writer((size_t)R * nt, -1), trsm((size_t)R * nt, -1), diag(nt, -1);
    auto add = [&](int type, int i, int j, int k0, int k1, float dur) {
        Node n;
        n.t = DagTask{type, i, j, (k0 << 16) | k1};
        n.dur = dur;
        n.prio = 0;
        n.npred = 0;
        nodes.push_back(std::move(n));
        return (int)nodes.size() - 1;
    };
    auto edge = [&](int from, int to) {
        if (from < 0) return;
        nodes[from].succ.push_back(to);
        nodes[to].npred++;
    };
    // `to` only needs `from` to have an earlier ticket (it waits for it half-way through, holding its workgroup): in
    // the simulation it may start as soon as `from` has started
    auto edge_after_start = [&](int from, int to) {
        if (from < 0) return;
        nodes[from].succ_start.push_back(to);
        nodes[to].npred++;
    };
    // durations in microseconds as measured at N = 10 000 (bulk workgroups share a CU's matrix cores in pairs, the
    // chain team's members have their CUs to themselves); only their proportions matter
#ifdef ALGP_DAG_DEBUG
    auto knob = [](const char* name, float dflt) { return getenv(name) ? (float)atof(getenv(name)) : dflt; };   // what-if runs of the probe
    const float D_DIAG = knob("DAG_D_DIAG", 45.f), D_STRIP = knob("DAG_D_STRIP", 10.f), D_OP = knob("DAG_D_OP", 34.f), D_OVH = knob("DAG_D_OVH", 5.f);
#else
    const float D_DIAG = 45.f, D_STRIP = 10.f, D_OP = 34.f, D_OVH = 5.f;
#endif
    int prev_h5 = -1;
    for (int k = 0; k < nt; ++k) {
        // the chain team's work of column step k: never ticketed, but part of the simulation, where each link starts
        // the moment its inputs are there (on_chain)
        auto chain = [&](int i, int j, float dur) {
            const int v = add(DAG_CHAIN, i, j, k, k + 1, dur);
            nodes[v].on_chain = true;
            return v;
        };
        int d = -1;
        if (!solve_only) {
            d = chain(k, k, D_DIAG);
            edge(last_writer[(size_t)k * nt + k], d);
            if (k + 1 < nt) {
                const int h1 = chain(k + 1, k, D_STRIP);                         // TRSM(k+1,k)
                edge(d, h1);
                edge(prev_h5, h1);                                               // the helpers work in sequence
                edge(last_writer[(size_t)(k + 1) * nt + k], h1);
                trsm[(size_t)(k + 1) * nt + k] = h1;
                const int h2 = chain(k + 1, k + 1, D_STRIP);                     // UPD(k+1,k+1,k)
                edge(h1, h2);
                edge(last_writer[(size_t)(k + 1) * nt + (k + 1)], h2);
                last_writer[(size_t)(k + 1) * nt + (k + 1)] = h2;
                prev_h5 = h2;
                if (k + 2 < nt) {
                    const int h3 = chain(k + 2, k, D_STRIP);                     // TRSM(k+2,k)
                    edge(h2, h3);
                    edge(last_writer[(size_t)(k + 2) * nt + k], h3);
                    trsm[(size_t)(k + 2) * nt + k] = h3;
                    const int h4 = chain(k + 2, k + 1, D_STRIP);                 // UPD(k+2,k+1,k)
                    edge(h3, h4);
                    edge(last_writer[(size_t)(k + 2) * nt + (k + 1)], h4);
                    last_writer[(size_t)(k + 2) * nt + (k + 1)] = h4;
                    const int h5 = chain(k + 2, k + 2, D_STRIP);                 // UPD(k+2,k+2,k)
                    edge(h4, h5);
                    edge(last_writer[(size_t)(k + 2) * nt + (k + 2)], h5);
                    last_writer[(size_t)(k + 2) * nt + (k + 2)] = h5;
                    prev_h5 = h5;
                }
            }
        }
        auto row_task = [&](int i) {
            if (k + 1 >= (i >= nt ? shape.pcols(i - nt) : nt)) {             // the row's last column: nothing to update to its right
                const int tr = add(DAG_TRSM, i, k, k, k + 1, D_OP + D_OVH);
                edge(d, tr);
                edge(last_writer[(size_t)i * nt + k], tr);
                trsm[(size_t)i * nt + k] = tr;
                return;
            }
            // TRSM(i,k) and UPD(i,k+1,k) are one ticketed task (DAG_TU): in the simulation the ticketed node is the first
            // product (L_ik is published when it ends) and the second a continuation that starts when its other inputs
            // are there.  The ticket is handed out only after the second product's other input, the tile's last update,
            // has its ticket (edge_after_start), so that everything a task waits for -- at its start or half-way --
            // is held by a workgroup that is already running: the no-deadlock argument of the header stays intact.
            const int tr = add(DAG_TU, i, k, k, k + 1, D_OP + D_OVH);
            edge(d, tr);
            edge(last_writer[(size_t)i * nt + k], tr);
            trsm[(size_t)i * nt + k] = tr;
            edge_after_start(last_writer[(size_t)i * nt + (k + 1)], tr);
            edge_after_start(trsm[(size_t)(k + 1) * nt + k], tr);           // ... and after the team's TRSM(k+1,k) has started:
                                                                             // every ticketed ancestor of L_(k+1)k is then done
            const int u = add(DAG_CHAIN, i, k + 1, k, k + 1, D_OP + 0.5f * D_OVH);
            nodes[u].on_chain = true;
            edge(tr, u);
            edge(last_writer[(size_t)i * nt + (k + 1)], u);
            edge(trsm[(size_t)(k + 1) * nt + k], u);
            last_writer[(size_t)i * nt + (k + 1)] = u;
        };
        if (!solve_only)
            for (int i = k + 3; i < nt; ++i) row_task(i);
        for (int e = 0; e < mt; ++e)                                         // panel rows: always ticketed, from their first column on
            if (shape.pstart(e) <= k && k < shape.pcols(e)) row_task(nt + e);
        auto upd = [&](int i, int j, int k0, int k1) {
            const int u = add(DAG_UPD, i, j, k0, k1, D_OVH + D_OP * (k1 - k0));
            edge(last_writer[(size_t)i * nt + j], u);
            for (int kk = k0; kk < k1; ++kk) {
                edge(trsm[(size_t)i * nt + kk], u);
                if (j != i) edge(trsm[(size_t)j * nt + kk], u);
            }
            last_writer[(size_t)i * nt + j] = u;
        };
        // rows that receive UPD(i, j, [k0, k1)): the factor's rows i >= j (the team's rows k+1, k+2 excepted for single
        // steps) and every panel row that has started before k1 (its batch begins at its first column)
        auto upd_rows = [&](int j, int k0, int k1, bool single) {
            if (!solve_only)
                for (int i = j; i < nt; ++i)
                    if (!single || i > k + 2) upd(i, j, k0, k1);
        };
        // The panel's far updates in batches of their own: a panel row gates nothing but itself (its own TU recurrence, 79
        // steps of ~90 us, is far shorter than the launch), so its batches are as long as the columns allow -- sixteens (K = 2048)
        // from column 0, at most one eight and one four next to the near window -- where the factor's
        // rows keep the shorter batches that let the chain run ahead.  In the folded launch of config 4's rank share (79 + 98
        // tile rows) 41 % of the workgroup time sat in panel batches of K = 512 / 1024 at 34.1 / 33.0 us per 128-step against
        // 31.8 for K = 2048 (tools/dag_test.hip 79 3 d 98): one read + write of the tile and one ticket / wait / publish per batch.
        auto panel_far = [&](int j) {
            const int kf = batched_until(j, W);
            if (k >= kf) return;
            const int pf8 = 8 * (kf / 8), pf16 = 16 * (pf8 / 16);
            int k0 = -1;
            if (k < pf16) { if ((k + 1) % 16 == 0) k0 = k - 15; }     // (a task polls 1 + 2 n <= 64 version words: n <= 31)
            else if (k < pf8) { if ((k + 1) % 8 == 0) k0 = k - 7; }
            else if ((k + 1) % 4 == 0) k0 = k - 3;
            if (k0 < 0) return;
            for (int e = 0; e < mt; ++e) {
                const int ps = shape.pstart(e);
                if (ps < k + 1 && j < shape.pcols(e)) upd(nt + e, j, std::max(k0, ps), k + 1);
            }
        };
        // A panel row is on nobody's critical path: the single steps of its tile (i, j) -- columns [kf, j-1), one to four of
        // them -- are ONE task of K = 128 .. 512, handed out when the last of those columns is solved (step j-2); the factor's
        // own rows keep their single steps, which gate the chain.  (23.7 % of the folded list's tasks were such single steps.)
        auto panel_near = [&](int j) {
            const int kf = batched_until(j, W);
            for (int e = 0; e < mt; ++e) {
                const int k0 = std::max(kf, shape.pstart(e));
                if (k0 < j - 1 && j < shape.pcols(e)) upd(nt + e, j, k0, j - 1);
            }
        };
        if (k + 2 < nt) panel_near(k + 2);
        for (int j = k + 1; j < nt; ++j) {
            if (mt > 0) panel_far(j);
            const int kf = batched_until(j, W);
            if (k >= kf) {
                if (j != k + 1) upd_rows(j, k, k + 1, true);                 // single step (column k+1 went with the TRSM: DAG_TU)
            } else {
                // Batches of four column steps (K = 512) next to the window, of eight (K = 1024) from DAG_FAR8 steps before
                // it, of sixteen (K = 2048) from DAG_FAR16 steps before that: one read + write of the tile and one ticket /
                // wait / publish per batch.  Measured at N = 10 000 (fours only -> this): 7.62 -> 7.35 ms in fp64, 5.14 ->
                // 4.80 ms in fp32, whose tile products are half as long and so feel the fixed cost per task twice as much
                // (sixteens everywhere: 8.15 ms -- a batch can only start when its last column is solved).
                const int kf8 = std::max(0, 8 * ((kf - DAG_FAR8) / 8));      // [kf16, kf8) in eights, [kf8, kf) in fours
                const int kf16 = std::max(0, 16 * ((kf8 - DAG_FAR16) / 16)); // [0, kf16) in sixteens
                if (k < kf16) {
                    if ((k + 1) % 16 == 0) upd_rows(j, k - 15, k + 1, false);
                } else if (k < kf8) {
                    if ((k + 1) % 8 == 0) upd_rows(j, k - 7, k + 1, false);
                } else if ((k + 1) % 4 == 0) {
                    upd_rows(j, k - 3, k + 1, false);
                }
            }
        }
    }
    const int n = (int)nodes.size();
    // bottom levels (longest path to the end) in reverse generation order (generation order is topological)
    for (int v = n - 1; v >= 0; --v) {
        float b = 0;
        for (int s : nodes[v].succ) b = std::max(b, nodes[s].prio);
        nodes[v].prio = b + nodes[v].dur;
    }
#ifdef ALGP_DAG_DEBUG
    {
        // critical path: follow the successor with the largest bottom level from the node with the largest one
        int v = 0;
        for (int u = 0; u < n; ++u) if (nodes[u].prio > nodes[v].prio) v = u;
        fprintf(stderr, "critical path %.0f us:", nodes[v].prio);
        float by_type[4] = {0, 0, 0, 0};
        int cnt = 0;
        for (;;) {
            by_type[nodes[v].t.type] += nodes[v].dur;
            if (cnt++ < 12 || nodes[v].succ.empty())
                fprintf(stderr, " %s(%d,%d,%d..%d)", nodes[v].t.type == 0 ? "CHAIN" : nodes[v].t.type == 1 ? "TRSM" : nodes[v].t.type == 2 ? "UPD" : "TU", nodes[v].t.i, nodes[v].t.j,
                        nodes[v].t.kk >> 16, nodes[v].t.kk & 0xffff);
            else if (cnt == 14) fprintf(stderr, " ...");
            if (nodes[v].succ.empty()) break;
            int b = nodes[v].succ[0];
            for (int s2 : nodes[v].succ) if (nodes[s2].prio > nodes[b].prio) b = s2;
            v = b;
        }
        fprintf(stderr, "\n  time on the path by type: chain/continuations %.0f trsm %.0f upd %.0f trsm+upd %.0f\n", by_type[0], by_type[1], by_type[2], by_type[3]);
    }
    std::vector<float> t_start(n, 0.f);
#define DAG_SIM_START(v) t_start[v] = now
#else
#define DAG_SIM_START(v) do { } while (0)
#endif
    // list scheduling of the ticketed tasks on the workers left beside the team; chain links start the moment they
    // are ready
    typedef std::pair<float, int> PI;
    std::priority_queue<PI> ready;                                           // max bottom level first
    std::priority_queue<PI, std::vector<PI>, std::greater<PI>> running;      // earliest finish first
    out.tasks.clear();
    out.tasks.reserve(n);
    float now = 0;
    int freew = solve_only ? workers : workers - 2 * DAG_TEAM, started = 0;  // the team and its retired CU neighbours
    std::vector<int> just_started;
    auto release = [&](int v) {                                              // all inputs of v are there
        if (nodes[v].on_chain) {
            DAG_SIM_START(v);
            running.push(PI(now + nodes[v].dur, v));
            ++started;
            just_started.push_back(v);
        } else {
            ready.push(PI(nodes[v].prio, v));
        }
    };
    auto drain_started = [&]() {                                             // edge_after_start successors
        while (!just_started.empty()) {
            const int v = just_started.back();
            just_started.pop_back();
            for (int s2 : nodes[v].succ_start)
                if (--nodes[s2].npred == 0) release(s2);
        }
    };
    for (int v = 0; v < n; ++v)
        if (nodes[v].npred == 0) release(v);
    drain_started();
    while (started < n) {
        while (freew > 0 && !ready.empty()) {
            const int v = ready.top().second;
            ready.pop();
            out.tasks.push_back(nodes[v].t);
            ++started;
            DAG_SIM_START(v);
            running.push(PI(now + nodes[v].dur, v));
            --freew;
            just_started.push_back(v);
            drain_started();
        }
        if (running.empty()) break;                                          // cannot happen for a DAG
        now = running.top().first;
        while (!running.empty() && running.top().first <= now) {
            const int v = running.top().second;
            running.pop();
            if (!nodes[v].on_chain) ++freew;
            for (int s2 : nodes[v].succ)
                if (--nodes[s2].npred == 0) release(s2);
            drain_started();
        }
    }
    out.nt = nt;
    out.window = W;
    out.makespan = now;
#ifdef ALGP_DAG_DEBUG
    {
        // the simulated machine per 500 us: ticketed workers busy, and the diagonal block the chain has reached
        const int nb = (int)(now / 500.f) + 1;
        std::vector<double> busy(nb, 0.0);
        std::vector<int> chain_at(nb, 0);
        double work = 0;
        for (int v = 0; v < n; ++v) {
            if (nodes[v].on_chain) {
                if (nodes[v].t.i == nodes[v].t.j && nodes[v].dur > 40.f) chain_at[std::min(nb - 1, (int)(t_start[v] / 500.f))] = nodes[v].t.i;
                continue;
            }
            work += nodes[v].dur;
            for (float t = t_start[v]; t < t_start[v] + nodes[v].dur;) {
                const int b = std::min(nb - 1, (int)(t / 500.f));
                const float e = std::min(t_start[v] + nodes[v].dur, (b + 1) * 500.f);
                busy[b] += e - t;
                t = e;
            }
        }
        fprintf(stderr, "  simulated makespan %.0f us; ticketed work %.0f workgroup-us = %.0f us on %d workers\n  per 500 us, busy workers / chain step:",
                now, work, work / (workers - 2 * DAG_TEAM), workers - 2 * DAG_TEAM);
        for (int b = 0; b < nb; ++b) fprintf(stderr, " %.0f/%d", busy[b] / 500.0, chain_at[b]);
        fprintf(stderr, "\n");
    }
#endif
}

// One launch of the task list for `shape`: the factorisation of A (unless shape.solve_only) and, with a panel, P <- P L^-T
// for the mt x 128 rows of P in the same launch.  The per-launch state (pivots | control words | team counters | tile
// versions) is zeroed -- or, for lists whose tiles do not all start at version 0 (a final factor, identity rows that
// begin at their own column), copied from a template kept beside the task list.
template <typename T>
static int dag_launch(algp_ctx* c, const DagShape& shape, T* A, int64_t ld, T* invD, T* P, int64_t ldp, double* logdet_acc,
                      int* info) {
    const int nt = shape.nt, mt = shape.mt, R = nt + mt;
    const int W = 2;     // fine (K=128) steps next to a tile's own column; >= 2: the team owns rows k+1, k+2.  With the K = 1024 / 2048
                         // batches of round 3: W = 2 / 3 / 4 / 6 -> 7.28 / 7.32 / 7.28 / 7.53 ms in fp64, 4.49 / 4.58 / 4.86 / 5.10 ms in fp32
    const size_t nver = (size_t)R * nt;
    const size_t state_bytes = round_up((int64_t)(sizeof(int) * (nver + (DAG_CTRL + 3) / 4 * 4 + 5 * (size_t)nt) + sizeof(double) * 128 * nt + 16), 16);
    const size_t ver_off = sizeof(double) * 128 * nt + sizeof(int) * ((DAG_CTRL + 3) / 4 * 4 + 5 * (size_t)nt);
    DagCache* dc = nullptr;
    for (auto& e : c->dag_cache)
        if (e.nt == nt && e.mt == mt && e.mode == shape.mode && e.solve_only == (shape.solve_only ? 1 : 0) && e.pshort == shape.pshort) dc = &e;
    if (!dc) {
        hipDeviceProp_t prop;
        ALGP_HIP(hipGetDeviceProperties(&prop, c->device));
        DagSchedule sched;
        const int workers = 2 * prop.multiProcessorCount;      // two workgroups per CU: the grid of the launch below
        dag_build_schedule(shape, W, workers, sched);
        if (c->dag_cache.size() >= 6) {                        // keep the six most recent shapes
            ALGP_HIP(hipStreamSynchronize(c->cur));
            for (DevBuf* b : {&c->dag_cache.front().tasks, &c->dag_cache.front().init})
                if (b->p) { (void)hipFree(b->p); c->dev_bytes -= (int64_t)b->cap; }
            c->dag_cache.erase(c->dag_cache.begin());
        }
        DagCache e;
        e.nt = nt;
        e.mt = mt;
        e.mode = shape.mode;
        e.solve_only = shape.solve_only ? 1 : 0;
        e.pshort = shape.pshort;
        e.workers = workers;
        e.ntasks = (int)sched.tasks.size();
        ALGP_TRY(ensure(c, e.tasks, sizeof(DagTask) * std::max<size_t>(sched.tasks.size(), 1)));
        ALGP_HIP(hipMemcpy(e.tasks.p, sched.tasks.data(), sizeof(DagTask) * sched.tasks.size(), hipMemcpyHostToDevice));
        if (shape.solve_only || shape.mode == 2) {
            std::vector<char> st(state_bytes, 0);
            int* ver = (int*)(st.data() + ver_off);
            if (shape.solve_only)                              // every tile of the factor is final
                for (int i = 0; i < nt; ++i)
                    for (int j = 0; j <= i; ++j) ver[(size_t)i * nt + j] = j + 1;
            for (int r = 0; r < mt; ++r)                       // a panel row counts its column steps from its first column
                for (int j = shape.pstart(r); j < nt; ++j) ver[(size_t)(nt + r) * nt + j] = shape.pstart(r);
            int rc = ensure(c, e.init, state_bytes);
            if (rc != ALGP_OK) { (void)hipFree(e.tasks.p); c->dev_bytes -= (int64_t)e.tasks.cap; return rc; }
            ALGP_HIP(hipMemcpy(e.init.p, st.data(), state_bytes, hipMemcpyHostToDevice));
        }
        c->dag_cache.push_back(e);
        dc = &c->dag_cache.back();
    }
    ALGP_TRY(ensure(c, c->dag_state, state_bytes));
    if (dc->init.p) ALGP_HIP(hipMemcpyAsync(c->dag_state.p, dc->init.p, state_bytes, hipMemcpyDeviceToDevice, c->cur));
    else ALGP_HIP(hipMemsetAsync(c->dag_state.p, 0, state_bytes, c->cur));
    DagArgs<T> g;
    g.L = A;
    g.ld = ld;
    g.invD = invD;
    g.P = P;
    g.ldp = ldp;
    g.no_team = shape.solve_only ? 1 : 0;
    g.tasks = (const DagTask*)dc->tasks.p;
    g.ntasks = dc->ntasks;
    g.nt = nt;
    g.pivots = (double*)c->dag_state.p;                        // doubles first (8-byte aligned)
    g.ctrl = (int*)((char*)c->dag_state.p + sizeof(double) * 128 * nt);
    g.cnt = g.ctrl + (DAG_CTRL + 3) / 4 * 4;
    g.ver = g.cnt + 5 * nt;
    g.info = info;
    g.spin_limit = 200000000ull;
    g.skip_publish = -1;
    if (c->debug_dag_stall_ticket >= 0) {                      // algp_debug_dag_stall: one launch with a lost publish
        g.skip_publish = c->debug_dag_stall_ticket;
        g.spin_limit = 20000000ull;                            // 0.2 s
        c->debug_dag_stall_ticket = -1;
    }
    g.stats = nullptr;
    if (c->prof_on) {                                          // in-kernel accounting only while profiling is enabled
        if (!c->dag_stats.p) {
            ALGP_TRY(ensure(c, c->dag_stats, 64));
            ALGP_HIP(hipMemsetAsync(c->dag_stats.p, 0, 64, c->cur));
        }
        g.stats = (unsigned long long*)c->dag_stats.p;
    }
    const double npad = 128.0 * nt, mrows = 128.0 * mt;
    // flop executed: N^3/3 for the factor, N^2 per dense panel row, N^3/3 in all for the identity panel
    const double flops = (shape.solve_only ? 0.0 : npad * npad * npad / 3.0) +
                         (shape.mode == 1 ? npad * npad * mrows : (shape.mode == 2 ? npad * npad * npad / 3.0 : 0.0));
    {
        ProfScope ps(c, mt > 0 ? ALGP_PROF_DAG_PANEL : ALGP_PROF_CHOL_DAG, flops, sizeof(T) * npad * (npad + mrows));
        int grid = dc->workers;                                // the machine the list schedule was simulated for
        const int team = shape.solve_only ? 0 : DAG_TEAM;
        if (grid > dc->ntasks + team) grid = dc->ntasks + team;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(chol_dag_kernel<T>, dim3(grid), dim3(256), 0, c->cur, g);
        ALGP_HIP(hipGetLastError());
    }
    // log-determinant from the pivots (factorising lists) and the abort word -> info
    hipLaunchKernelGGL(dag_finish_kernel, dim3(1), dim3(256), 0, c->cur, g.pivots, shape.solve_only ? 0 : 128 * nt, g.ctrl, logdet_acc, info);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

template <typename T>
int cholesky_dag(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info) {
    DagShape sh;
    sh.nt = (int)(npad / NB);
    return dag_launch<T>(c, sh, A, ld, invD, (T*)nullptr, 0, logdet_acc, info);
}
template int cholesky_dag<double>(algp_ctx*, double*, int64_t, int64_t, double*, double*, int*);
template int cholesky_dag<float>(algp_ctx*, float*, int64_t, int64_t, float*, double*, int*);

// The factorisation of A and P <- P L^-T in ONE launch: the mpad rows of P ride along as extra block rows of the task list
// (mode 1: dense rows, the candidates' B^T -> V^T; mode 2: P = I, npad x npad -> L^-T, zero tiles never touched).
template <typename T>
int cholesky_dag_panel(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info, T* P, int64_t ldp,
                       int64_t mpad, int mode, int pshort) {
    DagShape sh;
    sh.nt = (int)(npad / NB);
    sh.mt = (int)(mpad / NB);
    sh.mode = mode;
    sh.pshort = mode == 1 ? pshort : 0;
    return dag_launch<T>(c, sh, A, ld, invD, P, ldp, logdet_acc, info);
}
template int cholesky_dag_panel<double>(algp_ctx*, double*, int64_t, int64_t, double*, double*, int*, double*, int64_t, int64_t, int, int);
template int cholesky_dag_panel<float>(algp_ctx*, float*, int64_t, int64_t, float*, double*, int*, float*, int64_t, int64_t, int, int);

// P <- P L^-T against a factor that is final (same task list without the factorisation's own tasks)
template <typename T>
int solve_dag_panel(algp_ctx* c, const T* L, int64_t npad, int64_t ld, const T* invD, int* info, T* P, int64_t ldp, int64_t mpad,
                    int mode, int pshort) {
    DagShape sh;
    sh.nt = (int)(npad / NB);
    sh.mt = (int)(mpad / NB);
    sh.mode = mode;
    sh.pshort = mode == 1 ? pshort : 0;
    sh.solve_only = true;
    return dag_launch<T>(c, sh, const_cast<T*>(L), ld, const_cast<T*>(invD), P, ldp, (double*)nullptr, info);
}
template int solve_dag_panel<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, int*, double*, int64_t, int64_t, int, int);
template int solve_dag_panel<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, int*, float*, int64_t, int64_t, int, int);

void dag_release(algp_ctx* c) {
    for (auto& e : c->dag_cache) {
        if (e.tasks.p) (void)hipFree(e.tasks.p);
        if (e.init.p) (void)hipFree(e.init.p);
    }
    c->dag_cache.clear();
}

}  // namespace algp
