// tail.hip -- the few NEW columns of V^T after rows were appended to the factor (f1, the active-learning loop:
// reference agent.py:66-82 adds the picked sites and the mobile readings of a step, agent.py:210 refits from scratch).
//
//   X[:, c0:c1) = ( B[:, c0:c1) - X[:, 0:c0) L[c0:c1, 0:c0)^T ) inv(L[c0:c1, c0:c1))^T          c1 - c0 <= 64
//   (c0 = the train set's old size, c1 = its new size: exactly the appended rows, no alignment -- round 5; rounds 3-4 started
//   at the 16-column boundary below c0 and ended at the one above c1: 48 columns of MFMA work for 30 new rows)
//
// The blocked solve of potrf.hip works in 128-column tiles: after an append of ~32 rows it re-solves the whole open tail
// block -- one or two 128-wide output tiles whose products walk ALL of V^T (K = N: 40 GB at N = 50 000 x 100 000
// candidates) on the matrix cores at full tile width, 24 ms where the data take 8 to stream.  Columns left of c0 do not
// change when rows are appended (L's old rows do not), so only [c0, c1) is computed here, as a 64-wide product that is
// HBM-bound by construction: a workgroup owns 128 candidate rows, streams them once through a three-stage LDS-DMA
// pipeline (128-byte row pieces, see the kernel) against the new rows of L (32 of them, or all 64), and finishes the columns in its
// epilogue with the trailing block of the tail's explicit inverse (for a lower-triangular D, inv(D)[S, S] = inv(D[S, S])
// for every diagonal range S): the accumulator of the first product is, as it lies in registers, the B operand of the
// second (MFMA layouts, mfma.h), so nothing goes through LDS in between.
// Orientation: MFMA rows = the new columns (A operand = rows of L), MFMA columns = candidates (B operand = rows of V^T).
#include "common.h"
#include "mfma.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace algp {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

template <typename T>
struct TailArgs {
    T* X;                 // V^T, row-major, leading dimension ldx; rows padded to a multiple of 128
    int64_t ldx;
    const T* Lrows;       // L + c0 * ldl: the new rows of the factor (row q = train row c0 + q), k-contiguous
    int64_t ldl;
    int lrows_valid;      // rows of Lrows that may be read (c1 - c0 .. 64): rows beyond are clamped to the last valid one
    const T* E;           // inv(D)[o:, o:] of the tail's 128 x 128 inverse (leading dimension 128), o = c0 mod 128
    int64_t c0;           // first new column = K of the product (a multiple of 16)
    int w;                // new columns (<= 64)
    // split form (tail_part_kernel + tail_finish_kernel): the k range cut into `nsplit` chunks of `kt_per` k-tiles, unit
    // u = chunk * nrb + row block; partial accumulators to part[(u * 4 + wave) * nt + tile][2][4][64]
    int nsplit, kt_per, nrb;
    T* part;
};

// a stage: 32 NL rows of L (NL = 1 while the new columns fit two 16-row tiles -- the usual append of ~30 rows -- else 2: all 64)
// + 64 JT rows of V^T, 128 bytes each
constexpr int tail_nl(int NT) { return NT <= 2 ? 1 : 2; }
constexpr int tail_stb(int NT, int JT) { return 4096 * tail_nl(NT) + 8192 * JT; }
// stages: three -- one being read, one or two k-tiles in flight.  (Four, where the small stage allows them at the same
// occupancy, were measured: 7.74 against 7.77 ms for 32 new columns over config 5's 40 GB; so were 64-row workgroups at
// four per CU against 128-row ones at two: 7.77 both.  What is left above the 6.75 ms the same kernel takes with its
// products removed does not respond to deeper prefetch or more waves.)
constexpr int tail_nst(int NT, int JT) { return 3; }
// workgroups per CU: 160 KB of LDS
constexpr int tail_occ(int NT, int JT) { return (160 * 1024) / (tail_nst(NT, JT) * tail_stb(NT, JT)); }

// acc += L[c0 .. c0 + 64, k-tiles [kt0, kt1)] (x) X[m0 .. m0 + 128, the same k-tiles]^T for the workgroup's 128 candidate rows;
// with_ktail: the fp32 half tile behind the last full one as well.  Every wave of the workgroup calls it with the same range.
// NT: the 16-row tiles of L that carry new columns (ceil(w / 16)), a compile-time constant: the products of a k-tile are one
// straight run of MFMAs in which consecutive instructions never share an accumulator.
// JT: 16-candidate tiles per wave -- the workgroup owns 64 JT candidate rows (JT = 2: 60 or 72 KB of LDS, two workgroups
// per CU; JT = 1: 36 or 48 KB, four or three).
template <typename T, int NT, int JT>
__device__ __forceinline__ void tail_accumulate(const TailArgs<T>& g, int64_t m0, int kt0, int kt1, bool with_ktail,
                                                typename MF<T>::acc_t (&acc)[4][2], char* smem) {
    // k-tiles of 128 bytes per row (the GEMM's are 64): this kernel lives on HBM bandwidth, and with 64-byte pieces of
    // 128 x 512 different rows in flight it reached 2.7 TB/s (14.6 ms for the 40 GB of config 5) -- every piece opens a
    // DRAM page of its own.  Three stages of 20 / 24 KB (4 / 8 KB of L rows + 16 KB of V^T rows), one or two k-tiles in flight.
    constexpr int NST = tail_nst(NT, JT), NL = tail_nl(NT), STB = tail_stb(NT, JT), AB = 4096 * NL;
    using F = MF<T>;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC;
    constexpr int BK = 8 * EPC;                                    // elements per 128-byte row piece

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // DMA: one instruction moves 8 rows x 128 bytes; lane l -> row l >> 3 of the group, LDS slot l & 7.  LDS image
    // [row][8 slots of 16 B], slot = chunk ^ ((row >> 1) & 7): the 16 rows a quarter-wave reads at one chunk index fall
    // into 16 different 16-byte bank groups.  Wave w stages rows 8 NL w .. 8 NL (w + 1) - 1 of L and 16 JT w .. of V^T.
    const int r8 = lane >> 3, slot = lane & 7;
    int lr[NL];
    const T* Lg[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int row = 8 * NL * wave + 8 * i + r8;                // row of the 32 NL-row L part
        lr[i] = row < g.lrows_valid ? row : g.lrows_valid - 1;     // never read beyond the factor's rows; such rows are masked below
        Lg[i] = g.Lrows + (int64_t)lr[i] * g.ldl + (slot ^ ((row >> 1) & 7)) * EPC;
    }
    const T* Xg[2 * JT];
#pragma unroll
    for (int i = 0; i < 2 * JT; ++i) {
        const int row = 16 * JT * wave + 8 * i + r8;               // row of the workgroup's V^T part
        Xg[i] = g.X + (m0 + row) * g.ldx + (slot ^ ((row >> 1) & 7)) * EPC;
    }
    auto stage = [&](int st, int kt) {
        char* As = smem + st * STB + wave * (1024 * NL);
        char* Bs = smem + st * STB + AB + wave * (2048 * JT);
#pragma unroll
        for (int i = 0; i < NL; ++i) __builtin_amdgcn_global_load_lds((glb_vp)(Lg[i] + (int64_t)kt * BK), (lds_vp)(As + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2 * JT; ++i) __builtin_amdgcn_global_load_lds((glb_vp)(Xg[i] + (int64_t)kt * BK), (lds_vp)(Bs + i * 1024), 16, 0, 0);
    };
    const int fr = lane & 15, fg = lane >> 4;
    // chunk 4 h + fg of row (16 t + fr): slot = (4 h + fg) ^ ((row >> 1) & 7), and (row >> 1) & 7 = (fr >> 1) for every tile
    const int sw = (fr >> 1) & 7;
    const int aoff0 = fr * 128 + ((fg ^ sw) << 4), aoff1 = fr * 128 + (((4 + fg) ^ sw) << 4);
    const int boff0 = AB + (16 * JT * wave + fr) * 128 + ((fg ^ sw) << 4), boff1 = AB + (16 * JT * wave + fr) * 128 + (((4 + fg) ^ sw) << 4);

    // One barrier per k-tile, the fragments of tile kt + 1 read from LDS while the products of tile kt run (two register
    // sets): behind the barrier of step kt every wave holds tile kt in registers (its stage is free: refilled at once with
    // tile kt + 3) and tile kt + 1 has landed for everybody (read now, multiplied in the next step).  Round 5's first form
    // read a tile's fragments, waited for them, passed a second barrier, refilled and only then multiplied: with every
    // load served from cache it still took 6.9 ms for 32 new columns over config 5's 40 GB (4.1 ms of MFMA work; 2.8 ms
    // with the products removed as well) -- the matrix pipe idled through the LDS latency and two barriers per 16 MFMAs,
    // and the kernel was bound by that as much as by HBM (6.8 ms with the products removed: 5.9 TB/s).
    chunk_t fa[2][2][NT], fb[2][2][JT];                            // [register set][half of the k-tile][tile]
    auto read_frags = [&](auto set_c, int st_) {
        constexpr int set = decltype(set_c)::value;
        const char* base = smem + st_ * STB;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            fa[set][0][i] = *reinterpret_cast<const chunk_t*>(base + aoff0 + i * 2048);
            fa[set][1][i] = *reinterpret_cast<const chunk_t*>(base + aoff1 + i * 2048);
        }
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            fb[set][0][j] = *reinterpret_cast<const chunk_t*>(base + boff0 + j * 2048);
            fb[set][1][j] = *reinterpret_cast<const chunk_t*>(base + boff1 + j * 2048);
        }
    };
    // NL + 2 JT DMA instructions per tile and wave
    auto wait_landed = [&](int flying) {                           // all of this wave's DMA but the last `flying` tiles'
        constexpr int NI = NL + 2 * JT;                            // instructions per tile; vmcnt: bits 3:0 and 15:14 of the immediate
#define ALGP_VMCNT(n) (0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
        if (flying >= 3) __builtin_amdgcn_s_waitcnt(ALGP_VMCNT(3 * NI));
        else if (flying == 2) __builtin_amdgcn_s_waitcnt(ALGP_VMCNT(2 * NI));
        else if (flying == 1) __builtin_amdgcn_s_waitcnt(ALGP_VMCNT(NI));
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                       // vmcnt(0)
#undef ALGP_VMCNT
    };
    auto products = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) acc[i][j] = F::mfma(fa[set][h][i][e], fb[set][h][j][e], acc[i][j]);
        }
    };
    // step kt: set `cur` holds tile kt, stage st0 held it; tile kt + 1 lies in stage st1.  (Spreading the refill's DMA
    // instructions and the LDS reads over the products, one behind each MFMA, was measured as well: no faster -- 8.07 against
    // 7.89 ms -- and 60 more registers.)
    auto step = [&](auto cur_c, auto nxt_c, int kt, int st0, int st1) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): tile kt is in this wave's registers (the builtin, not
                                                                    // inline asm: the compiler's own wait insertion must see it, or it
                                                                    // waits again in front of the products -- for tile kt + 1's reads too)
        wait_landed(min(NST - 2, max(0, kt1 - 2 - kt)));            // tile kt + 1 has landed; tiles kt + 2 .. kt + NST - 1 may fly
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NST < kt1) stage(st0, kt + NST);                   // wave-uniform
        if (kt + 1 < kt1) read_frags(nxt_c, st1);
        __builtin_amdgcn_sched_barrier(0);                          // the LDS reads are issued before the products, not after
        products(cur_c);
    };
    if (kt0 < kt1) {
#pragma unroll
        for (int t = 0; t < NST; ++t)
            if (kt0 + t < kt1) stage(t, kt0 + t);
        wait_landed(min(NST - 1, kt1 - 1 - kt0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(std::integral_constant<int, 0>{}, 0);
        int st = 0;
        for (int kt = kt0; kt < kt1; kt += 2) {
            const int s1 = st == NST - 1 ? 0 : st + 1, s2 = s1 == NST - 1 ? 0 : s1 + 1;
            step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, kt, st, s1);
            if (kt + 1 < kt1) step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, kt + 1, s1, s2);
            st = s2;
        }
    }
    const int nkt = (int)(g.c0 / BK);
    const int ktail = (int)(g.c0 - (int64_t)nkt * BK);             // columns behind the last full k-tile: c0 need not be aligned
    if (with_ktail && ktail > 0) {
        // the first new column is the train set's old size, any number: the columns [nkt BK, c0) as plain masked loads, four
        // per MFMA step (lane group fg supplies k = k0 + 4 s + fg); columns from c0 on are the ones being computed: zero
        const int64_t k0 = (int64_t)nkt * BK;
        for (int s4 = 0; 4 * s4 < ktail; ++s4) {
            const int64_t kk = k0 + 4 * s4 + fg;
            const bool in = kk < g.c0;
            T b[JT], a[NT];
#pragma unroll
            for (int j = 0; j < JT; ++j) b[j] = in ? g.X[(m0 + 16 * JT * wave + 16 * j + fr) * g.ldx + kk] : (T)0;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                int row = 16 * i + fr;
                if (row >= g.lrows_valid) row = g.lrows_valid - 1;
                a[i] = in ? g.Lrows[(int64_t)row * g.ldl + kk] : (T)0;
            }
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i][j] = F::mfma(a[i], b[j], acc[i][j]);
        }
    }
}

// acc holds the full product for the workgroup's 128 candidates: X_new = (B - acc) inv(D)^T into the w new columns
template <typename T, int JT>
__device__ __forceinline__ void tail_epilogue(const TailArgs<T>& g, int64_t m0, typename MF<T>::acc_t (&acc)[4][2]) {
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15;
    // T = B - acc for the w new columns (element (new column 16 i + row_of, candidate 32 wave + 16 j + fr)); zero beyond w
    const int w = g.w;
    T* Xw = g.X + (m0 + 16 * JT * wave + fr) * g.ldx + g.c0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * i + F::row_of(lane, r);
                acc[i][j][r] = q < w ? Xw[(int64_t)(16 * j) * g.ldx + q] - acc[i][j][r] : (T)0;
            }
    // X_new^T = E T^T, E lower triangular by 16 x 16 blocks: tile row i takes E[i][i'] T[i'] for i' <= i.  A operand: lane
    // (row fr of tile i, k = row_of(lane, s)) -- the k order in which register s of the accumulator holds T's rows
#pragma unroll
    for (int i = 3; i >= 0; --i) {                                 // downwards: acc[i] is overwritten once nothing above needs it
        if (16 * i >= w) {                                         // wave-uniform
            continue;
        }
        acc_t o[JT];
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[j][r] = (T)0;
        const int qa = 16 * i + fr;
#pragma unroll
        for (int ip = 0; ip <= i; ++ip) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * ip + F::row_of(lane, s);
                const T e = (qa < w && k < w) ? g.E[qa * 128 + k] : (T)0;
#pragma unroll
                for (int j = 0; j < JT; ++j) o[j] = F::mfma(e, acc[ip][j][s], o[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = o[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * i + F::row_of(lane, r);
                if (q < w) Xw[(int64_t)(16 * j) * g.ldx + q] = acc[i][j][r];
            }
}

template <typename T, int NT, int JT>
__global__ __launch_bounds__(256, tail_occ(NT, JT)) void tail_cols_kernel(TailArgs<T> g) {
    using acc_t = typename MF<T>::acc_t;
    __shared__ __attribute__((aligned(1024))) char smem[tail_nst(NT, JT) * tail_stb(NT, JT)];
    const int64_t m0 = (int64_t)blockIdx.x * (64 * JT);
    acc_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;
    tail_accumulate<T, NT, JT>(g, m0, 0, (int)(g.c0 / (8 * MF<T>::EPC)), true, acc, smem);
    tail_epilogue<T, JT>(g, m0, acc);
}

// The same product with the k range cut into nsplit chunks: unit u = chunk * nrb + row block, workgroup b takes units b,
// b + gridDim.x, ... (chunk-major: the workgroups running at one moment mostly walk the same k chunk of the 64 new rows
// of L, which then stays in the XCDs' L2) and leaves each unit's accumulators in `part`.  Why: one workgroup per 128
// candidates gives 782 workgroups for config 5's 100 000 candidates on 512 slots (two per CU) -- the second round runs on
// half a machine (3.6 - 4.4 TB/s) --, and a rank's 12 500 candidates fill 98 of the 512 slots (1.9 TB/s).  nsplit is chosen
// by the host so that units / slots sits just below an integer.  Fixed assignment, fixed summation order in
// tail_finish_kernel: the same bits in every run.
template <typename T, int NT, int JT>
__global__ __launch_bounds__(256, tail_occ(NT, JT)) void tail_part_kernel(TailArgs<T> g) {
    using acc_t = typename MF<T>::acc_t;
    __shared__ __attribute__((aligned(1024))) char smem[tail_nst(NT, JT) * tail_stb(NT, JT)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int nt = NT;
    const int nkt = (int)(g.c0 / (8 * MF<T>::EPC));
    const int units = g.nsplit * g.nrb;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int chunk = u / g.nrb, rb = u - chunk * g.nrb;
        const int kt0 = chunk * g.kt_per, kt1 = min(nkt, kt0 + g.kt_per);
        acc_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;
        __syncthreads();                                           // the previous unit's last LDS reads are done
        tail_accumulate<T, NT, JT>(g, (int64_t)rb * (64 * JT), kt0, kt1, chunk == g.nsplit - 1, acc, smem);
        T* dst = g.part + ((int64_t)u * 4 + wave) * nt * (256 * JT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nt) {
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((i * JT + j) * 4 + r) * 64 + lane] = acc[i][j][r];
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);                        // the stores have left before the next unit's DMA is counted
    }
}
template <typename T, int JT>
__global__ __launch_bounds__(256) void tail_finish_kernel(TailArgs<T> g) {
    using acc_t = typename MF<T>::acc_t;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt = (g.w + 15) >> 4;
    const int rb = blockIdx.x;
    acc_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;
    for (int chunk = 0; chunk < g.nsplit; ++chunk) {               // ascending k: one fixed order of summation
        const T* src = g.part + (((int64_t)chunk * g.nrb + rb) * 4 + wave) * nt * (256 * JT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nt) {
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += src[((i * JT + j) * 4 + r) * 64 + lane];
            }
    }
    tail_epilogue<T, JT>(g, (int64_t)rb * (64 * JT), acc);
}

template <typename T, int JT>
static int tail_dispatch(algp_ctx* c, TailArgs<T>& g, int64_t mpad, int nkt, int w) {
    // One workgroup per 64 JT rows leaves the last round of workgroups on a part of the machine (or, for a rank's share of
    // the candidates, never fills it): cut the k range so that the units fill the slots evenly ($ALGP_TAIL_SPLIT=0: never).
    const int nt = (w + 15) / 16;
    const int nrb = (int)(mpad / (64 * JT)), slots = 256 * tail_occ(nt, JT);
    int best = 1;
    {
        static const bool split_on = env_switch("ALGP_TAIL_SPLIT", true);
        auto eff = [&](int s_) { const double r = (double)s_ * nrb / slots; return r / ceil(r); };
        double be = eff(1);
        // worth two launches and the partials' round trip only where the plain launch wastes more than 6 % of the machine
        for (int s_ = 2; split_on && be < 0.94 && s_ <= 16 && nkt / s_ >= 48; ++s_)
            if (eff(s_) > be + 0.02) { be = eff(s_); best = s_; }
    }
    if (best == 1) {
        switch (nt) {
            case 1: hipLaunchKernelGGL((tail_cols_kernel<T, 1, JT>), dim3((unsigned)nrb), dim3(256), 0, c->cur, g); break;
            case 2: hipLaunchKernelGGL((tail_cols_kernel<T, 2, JT>), dim3((unsigned)nrb), dim3(256), 0, c->cur, g); break;
            case 3: hipLaunchKernelGGL((tail_cols_kernel<T, 3, JT>), dim3((unsigned)nrb), dim3(256), 0, c->cur, g); break;
            default: hipLaunchKernelGGL((tail_cols_kernel<T, 4, JT>), dim3((unsigned)nrb), dim3(256), 0, c->cur, g); break;
        }
        ALGP_HIP(hipGetLastError());
        return ALGP_OK;
    }
    ALGP_TRY(ensure(c, c->tailPart, sizeof(T) * (size_t)best * (size_t)nrb * 4 * (size_t)nt * (256 * JT)));
    g.nsplit = best;
    g.kt_per = (nkt + best - 1) / best;
    g.nrb = nrb;
    g.part = (T*)c->tailPart.p;
    const dim3 pgrid((unsigned)std::min(slots, best * nrb));
    switch (nt) {
        case 1: hipLaunchKernelGGL((tail_part_kernel<T, 1, JT>), pgrid, dim3(256), 0, c->cur, g); break;
        case 2: hipLaunchKernelGGL((tail_part_kernel<T, 2, JT>), pgrid, dim3(256), 0, c->cur, g); break;
        case 3: hipLaunchKernelGGL((tail_part_kernel<T, 3, JT>), pgrid, dim3(256), 0, c->cur, g); break;
        default: hipLaunchKernelGGL((tail_part_kernel<T, 4, JT>), pgrid, dim3(256), 0, c->cur, g); break;
    }
    ALGP_HIP(hipGetLastError());
    hipLaunchKernelGGL((tail_finish_kernel<T, JT>), dim3((unsigned)nrb), dim3(256), 0, c->cur, g);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

// X[:, c0 : c0 + w) of the mpad rows of X <- the solution's new columns (see the header); c0 a multiple of 16, w <= 64, the
// columns inside ONE 128-column block of the factor, invD_blk = that block's explicit inverse (128 x 128, ld 128) -- or any
// 16-aligned range of at most 64 columns with E_window = the inverse of the 128 x 128 window of L at (c0, c0).
template <typename T>
int tail_cols_launch(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t ldl, int64_t lrows, const T* invD_blk,
                     int64_t c0, int w, const T* E_window) {
    if (mpad <= 0 || w <= 0) return ALGP_OK;
    constexpr int G = 16;
    if (mpad % 128 || w > 64 || (!E_window && (c0 % G || c0 / 128 != (c0 + w - 1) / 128)) || ldx % 4 || ldl % 4 || lrows < c0 + w)
        return fail(c, ALGP_ERR_BAD_ARG, "tail_cols: at most 64 columns (a 16-aligned range inside one 128-column block unless the "
                                         "inverse of the range's own window is supplied)");
    TailArgs<T> g;
    g.X = X;
    g.ldx = ldx;
    g.Lrows = L + c0 * ldl;
    g.ldl = ldl;
    g.lrows_valid = (int)std::min<int64_t>(64, lrows - c0);
    const int64_t o = c0 % 128;
    // E_window: inv(L[c0 : c0 + w, c0 : c0 + w)) with leading dimension 128 (the caller inverts a 128 x 128 window of L that
    // contains the range and passes the range's corner of it) -- for a range that starts at an arbitrary column or straddles
    // two 128-column blocks of the factor, where no stored block inverse covers it
    g.E = E_window ? E_window : invD_blk + o * 128 + o;
    g.c0 = c0;
    g.w = w;
    g.nsplit = 1;
    g.kt_per = 0;
    g.nrb = 0;
    g.part = nullptr;
    const int nkt = (int)(c0 / (16 / sizeof(T) * 8));
    ProfScope ps(c, klass, 2.0 * (double)mpad * (double)c0 * w, sizeof(T) * ((double)mpad * (double)c0 + 64.0 * (double)c0));
    // 128 candidate rows per workgroup, two workgroups per CU (64 rows at three or four per CU: the same time, EXPERIMENTS.md)
    return tail_dispatch<T, 2>(c, g, mpad, nkt, w);
}
template int tail_cols_launch<double>(algp_ctx*, int, double*, int64_t, int64_t, const double*, int64_t, int64_t, const double*, int64_t, int,
                                      const double*);
template int tail_cols_launch<float>(algp_ctx*, int, float*, int64_t, int64_t, const float*, int64_t, int64_t, const float*, int64_t, int,
                                     const float*);

}  // namespace algp
