// tail.hip -- the few NEW columns of V^T after rows were appended to the factor (f1, the active-learning loop:
// reference agent.py:66-82 adds the picked sites and the mobile readings of a step, agent.py:210 refits from scratch).
//
//   X[:, c0:c1) = ( B[:, c0:c1) - X[:, 0:c0) L[c0:c1, 0:c0)^T ) inv(L[c0:c1, c0:c1))^T          c1 - c0 <= 64
//
// The blocked solve of potrf.hip works in 128-column tiles: after an append of ~32 rows it re-solves the whole open tail
// block -- one or two 128-wide output tiles whose products walk ALL of V^T (K = N: 40 GB at N = 50 000 x 100 000
// candidates) on the matrix cores at full tile width, 24 ms where the data take 8 to stream.  Columns left of c0 do not
// change when rows are appended (L's old rows do not), so only [c0, c1) is computed here, as a 64-wide product that is
// HBM-bound by construction: a workgroup owns 128 candidate rows, streams them once through a three-stage LDS-DMA
// pipeline (128-byte row pieces, see the kernel) against the 64 new rows of L, and finishes the columns in its
// epilogue with the trailing block of the tail's explicit inverse (for a lower-triangular D, inv(D)[S, S] = inv(D[S, S])
// for every diagonal range S): the accumulator of the first product is, as it lies in registers, the B operand of the
// second (MFMA layouts, mfma.h), so nothing goes through LDS in between.
// Orientation: MFMA rows = the new columns (A operand = rows of L), MFMA columns = candidates (B operand = rows of V^T).
#include "common.h"
#include "mfma.h"

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
};

template <typename T>
__global__ __launch_bounds__(256, 2) void tail_cols_kernel(TailArgs<T> g) {
    // k-tiles of 128 bytes per row (the GEMM's are 64): this kernel lives on HBM bandwidth, and with 64-byte pieces of
    // 128 x 512 different rows in flight it reached 2.7 TB/s (14.6 ms for the 40 GB of config 5) -- every piece opens a
    // DRAM page of its own.  Three stages of 24 KB (8 KB of L rows + 16 KB of V^T rows), two k-tiles in flight.
    constexpr int NST = 3, STB = 24576;
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC;
    constexpr int BK = 8 * EPC;                                    // elements per 128-byte row piece
    __shared__ __attribute__((aligned(1024))) char smem[NST * STB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    // DMA: one instruction moves 8 rows x 128 bytes; lane l -> row l >> 3 of the group, LDS slot l & 7.  LDS image
    // [row][8 slots of 16 B], slot = chunk ^ ((row >> 1) & 7): the 16 rows a quarter-wave reads at one chunk index fall
    // into 16 different 16-byte bank groups.  Wave w stages rows 16 w .. 16 w + 15 of L and 32 w .. 32 w + 31 of V^T.
    const int r8 = lane >> 3, slot = lane & 7;
    int lr[2];
    const T* Lg[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * wave + 8 * i + r8;                    // row of the 64-row L part
        lr[i] = row < g.lrows_valid ? row : g.lrows_valid - 1;     // never read beyond the factor's rows; such rows are masked below
        Lg[i] = g.Lrows + (int64_t)lr[i] * g.ldl + (slot ^ ((row >> 1) & 7)) * EPC;
    }
    const T* Xg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * wave + 8 * i + r8;                    // row of the 128-row V^T part
        Xg[i] = g.X + (m0 + row) * g.ldx + (slot ^ ((row >> 1) & 7)) * EPC;
    }
    auto stage = [&](int st, int kt) {
        char* As = smem + st * STB + wave * 2048;
        char* Bs = smem + st * STB + 8192 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((glb_vp)(Lg[i] + (int64_t)kt * BK), (lds_vp)(As + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_vp)(Xg[i] + (int64_t)kt * BK), (lds_vp)(Bs + i * 1024), 16, 0, 0);
    };
    const int fr = lane & 15, fg = lane >> 4;
    // chunk 4 h + fg of row (16 t + fr): slot = (4 h + fg) ^ ((row >> 1) & 7), and (row >> 1) & 7 = (fr >> 1) for every tile
    const int sw = (fr >> 1) & 7;
    const int aoff0 = fr * 128 + ((fg ^ sw) << 4), aoff1 = fr * 128 + (((4 + fg) ^ sw) << 4);
    const int boff0 = 8192 + (32 * wave + fr) * 128 + ((fg ^ sw) << 4), boff1 = 8192 + (32 * wave + fr) * 128 + (((4 + fg) ^ sw) << 4);

    acc_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    const int nt = (g.w + 15) >> 4;                                // 16-row tiles of L that carry new columns (wave-uniform)
    const int nkt = (int)(g.c0 / BK);                              // c0 is a multiple of 16 elements: of BK for fp64; fp32 below
    const int ktail = (int)(g.c0 - (int64_t)nkt * BK);             // fp32 only: 16 elements left over (half a k-tile)
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nkt) stage(t, t);
    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (nkt - 1 - kt >= 1) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6): this tile landed, the next one may fly (6 DMA per tile)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* base = smem + st * STB;
        chunk_t a0[4], a1[4], b0[2], b1[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] = *reinterpret_cast<const chunk_t*>(base + aoff0 + i * 2048);
            a1[i] = *reinterpret_cast<const chunk_t*>(base + aoff1 + i * 2048);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b0[j] = *reinterpret_cast<const chunk_t*>(base + boff0 + j * 2048);
            b1[j] = *reinterpret_cast<const chunk_t*>(base + boff1 + j * 2048);
        }
        if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, kt + NST - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < EPC; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nt) {
                        acc[i][j] = F::mfma(a0[i][e], b0[j][e], acc[i][j]);
                        acc[i][j] = F::mfma(a1[i][e], b1[j][e], acc[i][j]);
                    }
        st = (st + 1 == NST) ? 0 : st + 1;
    }
    if (ktail > 0) {
        // fp32, c0 = 16 (mod 32): the last 16 columns as plain fragment loads (64 bytes per row: chunk fg of the row)
        __syncthreads();
        const int64_t k0 = (int64_t)nkt * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const chunk_t b = *reinterpret_cast<const chunk_t*>(g.X + (m0 + 32 * wave + 16 * j + fr) * g.ldx + k0 + fg * EPC);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = 16 * i + fr;
                if (row >= g.lrows_valid) row = g.lrows_valid - 1;
                const chunk_t a = *reinterpret_cast<const chunk_t*>(g.Lrows + (int64_t)row * g.ldl + k0 + fg * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[i][j] = F::mfma(a[e], b[e], acc[i][j]);
            }
        }
    }

    // T = B - acc for the w new columns (element (new column 16 i + row_of, candidate 32 wave + 16 j + fr)); zero beyond w
    const int w = g.w;
    T* Xw = g.X + (m0 + 32 * wave + fr) * g.ldx + g.c0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
        acc_t o[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
                for (int j = 0; j < 2; ++j) o[j] = F::mfma(e, acc[ip][j][s], o[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = o[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * i + F::row_of(lane, r);
                if (q < w) Xw[(int64_t)(16 * j) * g.ldx + q] = acc[i][j][r];
            }
}

// X[:, c0 : c0 + w) of the mpad rows of X <- the solution's new columns (see the header); c0 a multiple of 16, w <= 64, the
// columns inside ONE 128-column block of the factor; invD_blk = that block's explicit inverse (128 x 128, ld 128).
template <typename T>
int tail_cols_launch(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t ldl, int64_t lrows, const T* invD_blk,
                     int64_t c0, int w) {
    if (mpad <= 0 || w <= 0) return ALGP_OK;
    constexpr int G = 16;
    if (mpad % 128 || c0 % G || w > 64 || c0 / 128 != (c0 + w - 1) / 128 || ldx % 4 || ldl % 4 || lrows < c0 + w)
        return fail(c, ALGP_ERR_BAD_ARG, "tail_cols: columns must be a 16-aligned range of at most 64 inside one 128-column block");
    TailArgs<T> g;
    g.X = X;
    g.ldx = ldx;
    g.Lrows = L + c0 * ldl;
    g.ldl = ldl;
    g.lrows_valid = (int)std::min<int64_t>(64, lrows - c0);
    const int64_t o = c0 % 128;
    g.E = invD_blk + o * 128 + o;
    g.c0 = c0;
    g.w = w;
    ProfScope ps(c, klass, 2.0 * (double)mpad * (double)c0 * w, sizeof(T) * ((double)mpad * (double)c0 + 64.0 * (double)c0));
    hipLaunchKernelGGL(tail_cols_kernel<T>, dim3((unsigned)(mpad / 128)), dim3(256), 0, c->cur, g);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int tail_cols_launch<double>(algp_ctx*, int, double*, int64_t, int64_t, const double*, int64_t, int64_t, const double*, int64_t, int);
template int tail_cols_launch<float>(algp_ctx*, int, float*, int64_t, int64_t, const float*, int64_t, int64_t, const float*, int64_t, int);

}  // namespace algp
