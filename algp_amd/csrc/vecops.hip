// vecops.hip -- the HBM-bound tail of the path: row reductions over V^T (posterior variance and
// mean, greedy rank-1 updates), candidate scoring, argmax, and the fused kernel-GEMV for the
// mean-only posterior.  Replaces utils.py:301, 308 and the per-candidate slogdet loop of
// agent.py:317-347 (whose M fresh factorizations collapse to one pass over V^T per pick).
#include "common.h"
#include "vecops.h"
#include <algorithm>

namespace algp {

// ---------------------------------------------------------------------------------------------
// rows_reduce: one wave per row of V^T; 16-byte loads; ss[j] = sum v^2, dot[j] = sum v*w.
// ---------------------------------------------------------------------------------------------
// one wave, one row: lane-strided 16-byte loads, then a shuffle tree; lane 0 holds the sums.  Shared by the
// full-pass kernel and the lazy greedy refresh so that both produce bit-identical values.
template <typename T, bool HAS_W, bool HAS_SS>
__device__ __forceinline__ void wave_row_reduce(const T* row, const T* w, int64_t ncols, int lane, T& s2, T& sd) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int64_t nvec = ncols / VEC;           // full vectors; the tail (ncols % VEC) is handled scalar
    s2 = (T)0;
    sd = (T)0;
    for (int64_t v = lane; v < nvec; v += 64) {
        const vec_t x = *reinterpret_cast<const vec_t*>(row + v * VEC);
        if (HAS_W) {
            const vec_t y = *reinterpret_cast<const vec_t*>(w + v * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) sd += x[e] * y[e];
        }
        if (HAS_SS) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) s2 += x[e] * x[e];
        }
    }
    const int64_t t = nvec * VEC + lane;
    if (t < ncols) {
        const T x = row[t];
        if (HAS_W) sd += x * w[t];
        if (HAS_SS) s2 += x * x;
    }
    for (int o = 32; o > 0; o >>= 1) {
        if (HAS_SS) s2 += __shfl_down(s2, o, 64);
        if (HAS_W) sd += __shfl_down(sd, o, 64);
    }
}

// tri_c0 >= 0: Vt is UPPER triangular by 128-tiles (row j is zero -- or holds something else: the MI criterion keeps a
// factor's strictly-lower tiles there -- left of column 128 (j / 128)): row j is walked from max(128 (j / 128), tri_c0) on
template <typename T, bool HAS_W, bool HAS_SS>
__global__ __launch_bounds__(256) void rows_reduce_kernel(const T* Vt, int64_t rows, int64_t ldv, int64_t ncols,
                                                          const T* w, T* ss, T* dot, int64_t tri_c0) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t j = wave; j < rows; j += nw) {
        T s2, sd;
        const int64_t c0 = tri_c0 < 0 ? 0 : (j / 128 * 128 > tri_c0 ? j / 128 * 128 : tri_c0);
        wave_row_reduce<T, HAS_W, HAS_SS>(Vt + j * ldv + c0, HAS_W ? w + c0 : w, ncols - c0, lane, s2, sd);
        if (lane == 0) {
            if (HAS_SS) ss[j] = s2;
            if (HAS_W) dot[j] = sd;
        }
    }
}

// Dot product of a row with w in an order that sixteen waves together or one wave alone can follow, bit for bit: vector v
// (16 bytes) belongs to virtual lane v mod 1024, a virtual lane adds its vectors in ascending order (fused multiply-adds,
// element by element), each of the sixteen virtual waves folds its 64 lanes with the shuffle tree, and the sixteen partial
// sums are added left to right.  The lazy refresh of ONE row (the pick's own row, the best stale row: latency-bound, 117 us
// for a row of 50 000 with one wave) runs the sixteen virtual waves on sixteen real ones; the sweeps over many rows keep a
// wave per row with sixteen accumulators.  The elements behind the last full vector go to virtual wave 0.
template <typename T>
__device__ __forceinline__ void dot_vec(T& acc, const T* row, const T* w, int64_t v) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const vec_t x = *reinterpret_cast<const vec_t*>(row + v * VEC);
    const vec_t y = *reinterpret_cast<const vec_t*>(w + v * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc = __builtin_fma(x[e], y[e], acc);
}
template <typename T>
__device__ __forceinline__ T dot_tree(T s) {
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    return s;
}
// virtual wave vw's partial sum (valid in lane 0)
template <typename T>
__device__ __forceinline__ T row_dot_virtual_wave(const T* row, const T* w, int64_t ncols, int lane, int vw) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t nvec = ncols / VEC;
    T s = (T)0;
    for (int64_t v = vw * 64 + lane; v < nvec; v += 1024) dot_vec<T>(s, row, w, v);
    const int64_t t = nvec * VEC + lane;
    if (vw == 0 && t < ncols) s = __builtin_fma(row[t], w[t], s);
    return dot_tree<T>(s);
}
// the whole dot product by one wave (valid in lane 0)
template <typename T>
__device__ __forceinline__ T row_dot_one_wave(const T* row, const T* w, int64_t ncols, int lane) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t nvec = ncols / VEC;
    T acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = (T)0;
    for (int64_t base = 0; base < nvec; base += 1024) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t v = base + k * 64 + lane;
            if (v < nvec) dot_vec<T>(acc[k], row, w, v);
        }
    }
    const int64_t t = nvec * VEC + lane;
    if (t < ncols) acc[0] = __builtin_fma(row[t], w[t], acc[0]);
    T s = dot_tree<T>(acc[0]);
#pragma unroll
    for (int k = 1; k < 16; ++k) s += dot_tree<T>(acc[k]);
    return s;
}

// Column window [c0, c1) (multiples of 128) of every row: sum v^2, sum v u, sum v w; accumulate != 0 adds to the
// outputs.  Used by the incremental candidate solve: the sums over the kept columns of V^T are carried from
// step to step and only the new columns are read (the posterior mean needs V.z with z = u - ybar w).
template <typename T>
__global__ __launch_bounds__(256) void rows_reduce3_kernel(const T* Vt, int64_t rows, int64_t ldv, int64_t c0, int64_t c1,
                                                           const T* u, const T* w, T* oss, T* odu, T* odw, int accumulate) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t j = wave; j < rows; j += nw) {
        const T* row = Vt + j * ldv;
        T s2 = (T)0, su = (T)0, sw = (T)0;
        for (int64_t v = c0 / VEC + lane; v < c1 / VEC; v += 64) {
            const vec_t x = *reinterpret_cast<const vec_t*>(row + v * VEC);
            const vec_t a = *reinterpret_cast<const vec_t*>(u + v * VEC);
            const vec_t b = *reinterpret_cast<const vec_t*>(w + v * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s2 += x[e] * x[e];
                su += x[e] * a[e];
                sw += x[e] * b[e];
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            s2 += __shfl_down(s2, o, 64);
            su += __shfl_down(su, o, 64);
            sw += __shfl_down(sw, o, 64);
        }
        if (lane == 0) {
            if (accumulate) {
                oss[j] += s2;
                odu[j] += su;
                odw[j] += sw;
            } else {
                oss[j] = s2;
                odu[j] = su;
                odw[j] = sw;
            }
        }
    }
}

// ss = a_ss + t_ss ; dot = (a_du + t_du) - ybar (a_dw + t_dw)
template <typename T>
__global__ void combine3_kernel(int64_t M, const T* acc, const T* tmp, int64_t stride, T ybar, T* ss, T* dot) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    ss[j] = acc[j] + tmp[j];
    dot[j] = (acc[stride + j] + tmp[stride + j]) - ybar * (acc[2 * stride + j] + tmp[2 * stride + j]);
}

template <typename T>
int rows_reduce3_launch(algp_ctx* c, const T* Vt, int64_t rows, int64_t ldv, int64_t c0, int64_t c1, const T* u, const T* w,
                        T* out3, int64_t stride, int accumulate) {
    if (rows <= 0) return ALGP_OK;
    int64_t g = (rows + 3) / 4;
    if (g > 8192) g = 8192;
    ProfScope ps(c, ALGP_PROF_ROWS, 6.0 * rows * (double)(c1 - c0), sizeof(T) * (double)rows * (double)(c1 - c0));
    hipLaunchKernelGGL(rows_reduce3_kernel<T>, dim3((unsigned)g), dim3(256), 0, c->cur, Vt, rows, ldv, c0, c1, u, w, out3,
                       out3 + stride, out3 + 2 * stride, accumulate);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template <typename T>
int combine3_launch(algp_ctx* c, int64_t M, const T* acc, const T* tmp, int64_t stride, T ybar, T* ss, T* dot) {
    if (M <= 0) return ALGP_OK;
    hipLaunchKernelGGL(combine3_kernel<T>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->cur, M, acc, tmp, stride, ybar,
                       ss, dot);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int rows_reduce3_launch<double>(algp_ctx*, const double*, int64_t, int64_t, int64_t, int64_t, const double*,
                                         const double*, double*, int64_t, int);
template int rows_reduce3_launch<float>(algp_ctx*, const float*, int64_t, int64_t, int64_t, int64_t, const float*, const float*,
                                        float*, int64_t, int);
template int combine3_launch<double>(algp_ctx*, int64_t, const double*, const double*, int64_t, double, double*, double*);
template int combine3_launch<float>(algp_ctx*, int64_t, const float*, const float*, int64_t, float, float*, float*);

template <typename T>
int rows_reduce_launch(algp_ctx* c, const T* Vt, int64_t rows, int64_t ldv, int64_t ncols, const T* w, T* ss,
                       T* dot, int64_t tri_c0) {
    if (rows <= 0) return ALGP_OK;
    int64_t g = (rows + 3) / 4;
    if (g > 8192) g = 8192;
    const double share = tri_c0 < 0 ? 1.0 : 0.5;                   // an upper triangle is half the bytes
    ProfScope ps(c, ALGP_PROF_ROWS, share * 2.0 * rows * ncols * ((w ? 1 : 0) + (ss ? 1 : 0)), share * sizeof(T) * (double)rows * ncols);
    dim3 grid((unsigned)g), blk(256);
    if (w && ss) hipLaunchKernelGGL((rows_reduce_kernel<T, true, true>), grid, blk, 0, c->cur, Vt, rows, ldv, ncols, w, ss, dot, tri_c0);
    else if (w) hipLaunchKernelGGL((rows_reduce_kernel<T, true, false>), grid, blk, 0, c->cur, Vt, rows, ldv, ncols, w, ss, dot, tri_c0);
    else if (ss) hipLaunchKernelGGL((rows_reduce_kernel<T, false, true>), grid, blk, 0, c->cur, Vt, rows, ldv, ncols, w, ss, dot, tri_c0);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int rows_reduce_launch<double>(algp_ctx*, const double*, int64_t, int64_t, int64_t, const double*, double*, double*, int64_t);
template int rows_reduce_launch<float>(algp_ctx*, const float*, int64_t, int64_t, int64_t, const float*, float*, float*, int64_t);

// The row statistics the candidate solve left per column tile (gemm.hip, STATS: stat[(2 t + 0 / 1) * ld + row]) summed over
// the tiles in ascending order: ss = sum v^2, dot = sum v z of every row, without a second pass over V^T.
template <typename T>
__global__ __launch_bounds__(256) void rowstat_combine_kernel(const T* stat, int64_t ld, int ntiles, int64_t rows, T* ss, T* dot) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y;                                  // 0: sum v^2, 1: sum v z
    if (j >= rows) return;
    const T* p = stat + (int64_t)q * ld + j;
    T s = (T)0;
    int t = 0;
    for (; t + 8 <= ntiles; t += 8) {                          // eight loads in flight, added in tile order
        T v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[(int64_t)(2 * (t + e)) * ld];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; t < ntiles; ++t) s += p[(int64_t)(2 * t) * ld];
    (q == 0 ? ss : dot)[j] = s;
}
template <typename T>
int rowstat_combine_launch(algp_ctx* c, const T* stat, int64_t ld, int ntiles, int64_t rows, T* ss, T* dot) {
    if (rows <= 0) return ALGP_OK;
    ProfScope ps(c, ALGP_PROF_ROWS, 2.0 * rows * ntiles, sizeof(T) * 2.0 * (double)rows * ntiles);
    hipLaunchKernelGGL(rowstat_combine_kernel<T>, dim3((unsigned)((rows + 255) / 256), 2), dim3(256), 0, c->cur, stat, ld, ntiles, rows, ss, dot);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int rowstat_combine_launch<double>(algp_ctx*, const double*, int64_t, int, int64_t, double*, double*);
template int rowstat_combine_launch<float>(algp_ctx*, const float*, int64_t, int, int64_t, float*, float*);

// ---------------------------------------------------------------------------------------------
// candidate finalize: dstat = prior - ss (ordinary) | ss (unit row: [S^-1]_jj); mu = ybar + dot
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void cand_finalize_kernel(int64_t M, const int* ckind, const int64_t* cidx, const T* Cp, int64_t n_pool,
                                     T prior_const, const T* extra, const T* ss, const T* dot, T ybar, T* dstat,
                                     T* mu, unsigned char* alive) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const bool unit = ckind && ckind[j] >= 0;
    T prior = Cp ? Cp[cidx[j] * n_pool + cidx[j]] : prior_const;
    if (extra) prior += extra[j];
    dstat[j] = unit ? ss[j] : prior - ss[j];
    mu[j] = ybar + dot[j];
    alive[j] = 1;
}

template <typename T>
int cand_finalize_launch(algp_ctx* c, int64_t M, const int* ckind, const int64_t* cidx, const T* Cp, int64_t n_pool,
                         T prior_const, const T* extra, const T* ss, const T* dot, T ybar, T* dstat, T* mu,
                         unsigned char* alive) {
    if (M <= 0) return ALGP_OK;
    hipLaunchKernelGGL(cand_finalize_kernel<T>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->cur, M, ckind,
                       cidx, Cp, n_pool, prior_const, extra, ss, dot, ybar, dstat, mu, alive);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int cand_finalize_launch<double>(algp_ctx*, int64_t, const int*, const int64_t*, const double*, int64_t, double,
                                          const double*, const double*, const double*, double, double*, double*, unsigned char*);
template int cand_finalize_launch<float>(algp_ctx*, int64_t, const int*, const int64_t*, const float*, int64_t, float,
                                         const float*, const float*, const float*, float, float*, float*, unsigned char*);

// ---------------------------------------------------------------------------------------------
// scores (entropy gain per candidate, agent.py:341 after telescoping)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double entropy_utility(double d, bool unit, double ss, double delta) {
    return unit ? 0.5 * log1p(delta * d) : ENT_CONST + 0.5 * log(d + ss);
}

template <typename T>
__global__ void score_kernel(int64_t M, const int* ckind, const unsigned char* alive, const T* dstat, double ss,
                             double delta, const double* extra, double* out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    double u;
    if (!alive[j]) {
        u = -INFINITY;
    } else {
        u = entropy_utility((double)dstat[j], ckind[j] >= 0, ss, delta);
        if (extra) u += extra[j];
    }
    out[j] = u;
}

template <typename T>
int score_launch(algp_ctx* c, int64_t M, const int* ckind, const unsigned char* alive, const T* dstat, double ss,
                 double delta, const double* extra, double* out) {
    if (M <= 0) return ALGP_OK;
    ProfScope ps(c, ALGP_PROF_SCORE, 4.0 * M, (sizeof(T) + 13.0) * M);
    hipLaunchKernelGGL(score_kernel<T>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->cur, M, ckind, alive,
                       dstat, ss, delta, extra, out);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int score_launch<double>(algp_ctx*, int64_t, const int*, const unsigned char*, const double*, double, double,
                                  const double*, double*);
template int score_launch<float>(algp_ctx*, int64_t, const int*, const unsigned char*, const float*, double, double,
                                 const double*, double*);

// first maximum (np.argmax semantics, agent.py:349): larger value wins, ties go to the smaller index
__global__ __launch_bounds__(1024) void argmax_kernel(const double* s, int64_t M, double* out_val, int64_t* out_idx) {
    __shared__ double sv[16];
    __shared__ int64_t si[16];
    double bv = -INFINITY;
    int64_t bi = -1;
    for (int64_t j = threadIdx.x; j < M; j += 1024) {
        const double v = s[j];
        if (bi < 0 || v > bv) {     // j ascending per thread: strict > keeps the first maximum
            if (!(v != v)) { bv = v; bi = j; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(bv, o, 64);
        const int64_t oi = __shfl_down(bi, o, 64);
        if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (si[w] >= 0 && (bi < 0 || sv[w] > bv || (sv[w] == bv && si[w] < bi))) { bv = sv[w]; bi = si[w]; }
        *out_val = bv;
        *out_idx = bi;
    }
}

// The same in two launches for long score vectors: ARGMAX_G workgroups leave their first maxima (value, index), one wave
// folds them with the same rule -- which does not depend on how the entries were grouped.  (One workgroup walking config
// 5's 100 000 utilities: 25-45 us, twice per pick.)
constexpr int ARGMAX_G = 64;
__global__ __launch_bounds__(256) void argmax_part_kernel(const double* s, int64_t M, double* pv, int64_t* pi) {
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    double bv = -INFINITY;
    int64_t bi = -1;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < M; j += 256 * ARGMAX_G) {
        const double v = s[j];
        if (bi < 0 || v > bv) {     // j ascending per thread: strict > keeps the first maximum
            if (!(v != v)) { bv = v; bi = j; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(bv, o, 64);
        const int64_t oi = __shfl_down(bi, o, 64);
        if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (si[w] >= 0 && (bi < 0 || sv[w] > bv || (sv[w] == bv && si[w] < bi))) { bv = sv[w]; bi = si[w]; }
        pv[blockIdx.x] = bv;
        pi[blockIdx.x] = bi;
    }
}
__global__ __launch_bounds__(64) void argmax_final_kernel(const double* pv, const int64_t* pi, double* out_val, int64_t* out_idx) {
    double bv = pv[threadIdx.x];
    int64_t bi = pi[threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(bv, o, 64);
        const int64_t oi = __shfl_down(bi, o, 64);
        if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if (threadIdx.x == 0) {
        *out_val = bv;
        *out_idx = bi;
    }
}

int argmax_launch(algp_ctx* c, const double* s, int64_t M, double* out_val, int64_t* out_idx) {
    ProfScope ps(c, ALGP_PROF_SCORE, (double)M, 8.0 * M);
    static_assert(ARGMAX_G == 64, "argmax_final_kernel folds the partial maxima with one wave");
    if (M >= 32768) {
        ALGP_TRY(ensure(c, c->amax, 16 * ARGMAX_G));
        double* pv = (double*)c->amax.p;
        int64_t* pi = (int64_t*)(pv + ARGMAX_G);
        hipLaunchKernelGGL(argmax_part_kernel, dim3(ARGMAX_G), dim3(256), 0, c->cur, s, M, pv, pi);
        hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(64), 0, c->cur, (const double*)pv, (const int64_t*)pi, out_val, out_idx);
    } else {
        hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, c->cur, s, M, out_val, out_idx);
    }
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

// ---------------------------------------------------------------------------------------------
// greedy commit, per row j: r_j = (b'_j - V_j . l) * scale ; dstat_j -= r^2 (ordinary) / += r^2 (unit row);
// V^T[j][ncols] = r_j.   b'_j = C(pick, j) when the pick is a new site and row j is ordinary.
// ---------------------------------------------------------------------------------------------
// b'_j = C(pick, j) when the pick is a new site and row j is ordinary, else 0
template <typename T, int DP>
__device__ __forceinline__ T pick_bprime(bool unit, int64_t pj, const T* Xs, const T* Cp, int64_t n_pool,
                                         int64_t pick_pool, int pick_in_train, int kernel, T os, T noise) {
    T bp = (T)0;
    if (!pick_in_train && !unit) {
        if (Cp) {
            bp = Cp[pick_pool * n_pool + pj];
        } else {
            T r2 = (T)0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const T df = Xs[pick_pool * DP + d] - Xs[pj * DP + d];
                r2 += df * df;
            }
            if (kernel == ALGP_KERNEL_RBF) bp = os * kexp((T)-0.5 * r2);
            else {
                const T r = sqrt(r2) * (T)1.7320508075688772;
                bp = os * ((T)1 + r) * kexp(-r);
            }
            if (pj == pick_pool) bp += noise;
        }
    }
    return bp;
}

// z = u - ybar w ; rhs initialisation for the resumed substitutions: u[i] = y[i], w[i] = (i < n) for i >= k
template <typename T>
__global__ void uw_init_kernel(T* u, T* w, const T* y, int64_t k, int64_t n, int64_t npad) {
    const int64_t i = k + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    u[i] = y[i];
    w[i] = i < n ? (T)1 : (T)0;
}
template <typename T>
__global__ void uw_combine_kernel(T* z, const T* u, const T* w, T ybar, int64_t npad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) z[i] = u[i] - ybar * w[i];
}
template <typename T>
int uw_init_launch(algp_ctx* c, T* u, T* w, const T* y, int64_t k, int64_t n, int64_t npad) {
    if (npad <= k) return ALGP_OK;
    hipLaunchKernelGGL(uw_init_kernel<T>, dim3((unsigned)((npad - k + 255) / 256)), dim3(256), 0, c->cur, u, w, y, k, n, npad);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template <typename T>
int uw_combine_launch(algp_ctx* c, T* z, const T* u, const T* w, T ybar, int64_t npad) {
    hipLaunchKernelGGL(uw_combine_kernel<T>, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, c->cur, z, u, w, ybar, npad);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int uw_init_launch<double>(algp_ctx*, double*, double*, const double*, int64_t, int64_t, int64_t);
template int uw_init_launch<float>(algp_ctx*, float*, float*, const float*, int64_t, int64_t, int64_t);
template int uw_combine_launch<double>(algp_ctx*, double*, const double*, const double*, double, int64_t);
template int uw_combine_launch<float>(algp_ctx*, float*, const float*, const float*, float, int64_t);

// dst[r][0:ncols] = src[src_row[r]][0:ncols], or zeros where src_row[r] < 0 (one workgroup per row and 2048 columns: the
// row exchange of the sharded loop packs a handful of rows of 50 000).
// With lrow: rows with lrow[r] >= 0 are  Lb[lrow[r]][0:ncols] - lscale[r] * src[src_row[r]][0:ncols]
// (a second measurement of a site that already is train row lrow[r], see vt_rows_for_new_sites in api.hip).
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* src, int64_t lds, const int64_t* src_row, T* dst,
                                                          int64_t ldd, int64_t ncols, const int64_t* lrow, const T* lscale,
                                                          const T* Lb, int64_t ldl) {
    const int64_t r = blockIdx.x, sr = src_row[r];
    T* d = dst + r * ldd;
    const int64_t k0 = (int64_t)blockIdx.y * 2048 + threadIdx.x;
    ncols = ncols < ((int64_t)blockIdx.y + 1) * 2048 ? ncols : ((int64_t)blockIdx.y + 1) * 2048;
    if (sr < 0) {
        for (int64_t k = k0; k < ncols; k += 256) d[k] = (T)0;
    } else if (lrow && lrow[r] >= 0) {
        const T* sp = src + sr * lds;
        const int64_t lr = lrow[r];
        const T* lp = Lb + lr * ldl;
        const T sc = lscale[r];
        // row lr of L ends at its diagonal (the strict upper part of a diagonal block is scratch)
        for (int64_t k = k0; k < ncols; k += 256) d[k] = (k <= lr ? lp[k] : (T)0) - sc * sp[k];
    } else {
        const T* sp = src + sr * lds;
        for (int64_t k = k0; k < ncols; k += 256) d[k] = sp[k];
    }
}

template <typename T>
int gather_rows_launch(algp_ctx* c, const T* src, int64_t lds, const int64_t* src_row, T* dst, int64_t ldd, int64_t nrows,
                       int64_t ncols, const int64_t* lrow, const T* lscale, const T* Lb, int64_t ldl) {
    if (nrows <= 0 || ncols <= 0) return ALGP_OK;
    hipLaunchKernelGGL(gather_rows_kernel<T>, dim3((unsigned)nrows, (unsigned)((ncols + 2047) / 2048)), dim3(256), 0, c->cur, src, lds, src_row, dst, ldd, ncols,
                       lrow, lscale, Lb, ldl);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int gather_rows_launch<double>(algp_ctx*, const double*, int64_t, const int64_t*, double*, int64_t, int64_t, int64_t,
                                        const int64_t*, const double*, const double*, int64_t);
template int gather_rows_launch<float>(algp_ctx*, const float*, int64_t, const int64_t*, float*, int64_t, int64_t, int64_t,
                                       const int64_t*, const float*, const float*, int64_t);

// ---------------------------------------------------------------------------------------------
// Lazy greedy (entropy criterion).  The gain of a candidate never grows when more sites are sampled
// (information gain is submodular), so a utility computed before the last picks is an upper bound.  After
// a pick only the rows that can still win are brought up to date: the best stale row first (its fresh
// utility becomes the threshold), then every stale row whose bound reaches the threshold.  A refresh
// applies the missing picks in order with exactly the arithmetic of the full pass (row_dot_* +
// pick_bprime), so picks and values equal the full pass bit for bit; it costs one row of V^T per pick
// instead of a sweep over all M rows.
//   mode 0: row `pos` only (grid = 1 block)   mode 1: alive stale rows with scores >= scores[pos]
//   mode 2: every stale row (flush before anything reads the full state)
// pos_dev != null: the row is the one an argmax kernel has just left on the device (*pos_dev; < 0 = no row, nothing to
// do), so that argmax -> refresh -> refresh -> argmax is one stream-ordered chain without a host round trip.
// ---------------------------------------------------------------------------------------------
template <typename T, int DP, bool COOP>
__global__ __launch_bounds__(COOP ? 1024 : 256) void lazy_refresh_kernel(int64_t M, int mode, int64_t pos, const int64_t* pos_dev,
                                                           const LazyPick* picks,
                                                           int npicks, const int* ckind, const int64_t* cidx,
                                                           const T* Xs, const T* Cp, int64_t n_pool, int kernel, T os,
                                                           T noise, const T* prevrows, int64_t ldv, T* Vt, T* dstat,
                                                           int* fresh, const unsigned char* alive, double* scores,
                                                           double ss, double delta) {
    const int lane = threadIdx.x & 63;
    if (pos_dev) {
        pos = *pos_dev;
        if (pos < 0 && mode != 2) return;
    }
    if (COOP) {
        // one row (mode 0: row pos; mode 2 with M = 1: row 0), sixteen waves: wave w is virtual wave w of the dot product
        __shared__ T red[16];
        const int wv = threadIdx.x >> 6;
        const int64_t j = (mode == 0) ? pos : 0;
        const int f = fresh[j];
        if (f >= npicks) return;
        const bool unit = ckind[j] >= 0;
        const int64_t pj = cidx[j];
        T* row = Vt + j * ldv;
        T d = dstat[j];
        for (int q = f; q < npicks; ++q) {
            const LazyPick pk = picks[q];
            const T part = row_dot_virtual_wave<T>(row, prevrows + (int64_t)q * ldv, pk.ncols, lane, wv);
            if (lane == 0) red[wv] = part;
            __syncthreads();
            T sd = red[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) sd += red[k];
            const T bp = pick_bprime<T, DP>(unit, pj, Xs, Cp, n_pool, pk.pool_idx, (int)pk.in_train, kernel, os, noise);
            const T r = (bp - sd) * (T)pk.scale;
            d += unit ? r * r : -(r * r);
            if (threadIdx.x == 0) row[pk.ncols] = r;
            __threadfence();                                   // the next pick's dot product reads this entry
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            dstat[j] = d;
            fresh[j] = npicks;
            scores[j] = alive[j] ? entropy_utility((double)d, unit, ss, delta) : -INFINITY;
        }
        return;
    }
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const double thr = (mode == 1) ? scores[pos] : 0.0;        // row pos is up to date in mode 1: never rewritten here
    for (int64_t j = wave; j < M; j += nw) {
        const int f = fresh[j];
        if (f >= npicks) continue;
        if (mode == 1 && (!alive[j] || !(scores[j] >= thr))) continue;
        const bool unit = ckind[j] >= 0;
        const int64_t pj = cidx[j];
        T* row = Vt + j * ldv;
        T d = dstat[j];
        for (int q = f; q < npicks; ++q) {
            const LazyPick pk = picks[q];
            T sd = row_dot_one_wave<T>(row, prevrows + (int64_t)q * ldv, pk.ncols, lane);
            sd = __shfl(sd, 0, 64);
            const T bp = pick_bprime<T, DP>(unit, pj, Xs, Cp, n_pool, pk.pool_idx, (int)pk.in_train, kernel, os, noise);
            const T r = (bp - sd) * (T)pk.scale;
            d += unit ? r * r : -(r * r);
            if (lane == 0) row[pk.ncols] = r;
            __threadfence();                                   // the next pick's dot product reads this entry
        }
        if (lane == 0) {
            dstat[j] = d;
            fresh[j] = npicks;
            scores[j] = alive[j] ? entropy_utility((double)d, unit, ss, delta) : -INFINITY;
        }
    }
}

template <typename T>
int lazy_refresh_launch(algp_ctx* c, int64_t M, int mode, int64_t pos, const LazyPick* picks, int npicks, const int* ckind,
                        const int64_t* cidx, const T* Xs, const T* Cp, int64_t n_pool, int DP, int kernel, T os, T noise,
                        const T* prevrows, int64_t ldv, T* Vt, T* dstat, int* fresh, const unsigned char* alive,
                        double* scores, double ss, double delta, const int64_t* pos_dev) {
    if (M <= 0 || npicks <= 0) return ALGP_OK;
    const bool coop = (mode == 0 || (mode == 2 && M == 1));      // one row: sixteen waves share its dot products
    int64_t g = coop ? 1 : (M + 3) / 4;
    if (g > 65536) g = 65536;
    ProfScope ps(c, ALGP_PROF_ROWS, 0.0, 13.0 * M);
    dim3 grid((unsigned)g), blk(coop ? 1024 : 256);
#define ALGP_LR(DPV, CO)                                                                                          \
    hipLaunchKernelGGL((lazy_refresh_kernel<T, DPV, CO>), grid, blk, 0, c->cur, M, mode, pos, pos_dev, picks, npicks, ckind, \
                       cidx, Xs, Cp, n_pool, kernel, os, noise, prevrows, ldv, Vt, dstat, fresh, alive, scores, ss, delta)
#define ALGP_LR2(DPV)                                                                                             \
    do {                                                                                                          \
        if (coop) ALGP_LR(DPV, true);                                                                             \
        else ALGP_LR(DPV, false);                                                                                 \
    } while (0)
    if (DP == 2) ALGP_LR2(2);
    else if (DP == 4) ALGP_LR2(4);
    else ALGP_LR2(8);
#undef ALGP_LR2
#undef ALGP_LR
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int lazy_refresh_launch<double>(algp_ctx*, int64_t, int, int64_t, const LazyPick*, int, const int*, const int64_t*,
                                         const double*, const double*, int64_t, int, int, double, double, const double*,
                                         int64_t, double*, double*, int*, const unsigned char*, double*, double, double,
                                         const int64_t*);
template int lazy_refresh_launch<float>(algp_ctx*, int64_t, int, int64_t, const LazyPick*, int, const int*, const int64_t*,
                                        const float*, const float*, int64_t, int, int, float, float, const float*, int64_t,
                                        float*, float*, int*, const unsigned char*, double*, double, double, const int64_t*);

// ---------------------------------------------------------------------------------------------
// greedy commit bookkeeping on the device (one thread): from the winner's statistic d_c (posterior variance, or
// [S^-1]_cc for a train site) the scale of the appended row, the pick record the lazy refresh reads, the winner's
// local row retired, and (d_c, scale) for the host's single read-back.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void commit_finalize_kernel(const T* dsrc, int in_train, double ss, double delta, LazyPick* lp_out,
                                       int64_t pool_idx, int64_t ncols, unsigned char* alive_local, double* score_local,
                                       double* out2) {
    const double dc = (double)*dsrc;
    const double scale = in_train ? sqrt(-(delta / (1.0 + delta * dc))) : 1.0 / sqrt(dc + ss);
    LazyPick lp;
    lp.pool_idx = pool_idx;
    lp.ncols = ncols;
    lp.scale = scale;
    lp.in_train = in_train;
    lp.d = dc;
    *lp_out = lp;
    // the winner retires -- unless its scale is not finite (a non-positive variance under the square root): the caller
    // of algp_commit_pick then takes the pick back, and the site must stay selectable
    if (alive_local && scale == scale && !isinf(scale)) {
        *alive_local = 0;
        *score_local = -INFINITY;
    }
    out2[0] = dc;
    out2[1] = scale;
}
template <typename T>
int commit_finalize_launch(algp_ctx* c, const T* dsrc, int in_train, double ss, double delta, LazyPick* lp_out,
                           int64_t pool_idx, int64_t ncols, unsigned char* alive_local, double* score_local, double* out2) {
    hipLaunchKernelGGL(commit_finalize_kernel<T>, dim3(1), dim3(1), 0, c->cur, dsrc, in_train, ss, delta, lp_out, pool_idx,
                       ncols, alive_local, score_local, out2);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int commit_finalize_launch<double>(algp_ctx*, const double*, int, double, double, LazyPick*, int64_t, int64_t,
                                            unsigned char*, double*, double*);
template int commit_finalize_launch<float>(algp_ctx*, const float*, int, double, double, LazyPick*, int64_t, int64_t,
                                           unsigned char*, double*, double*);

// ---------------------------------------------------------------------------------------------
// MI criterion (agent.py:330-339).  Besides the entropy gain, the utility of candidate i needs H(Abar \ i) and H(all_i):
//   H(Abar \ i) = H(Abar) - CONST + 1/2 log [C_AbarAbar^-1]_ii                 (i not sampled)
//   H(all_i)    = H(all)  + 1/2 log(1 + d_i [(C + D_all)^-1]_ii),  d_i = ss (new site) | v_fused - sm (mobile-sampled)
// i.e. the DIAGONALS of two pool-wide inverses, P = C_AbarAbar^-1 and Q = (C + D_all)^-1.  Both are kept as their
// triangular factors' inverses X (P = X X^T), computed once per candidate solve; a committed pick c is a rank-1 change
//   removal of site c from Abar:   P <- P - p p^T / P_cc,                 p = P[:, c]
//   noise change d at site c:      Q <- Q - g q q^T,  g = d/(1 + d Q_cc),  q = Q[:, c]
// so each pick costs one pass over X (column c of X X^T = X times row c of X, rows_reduce_kernel) plus this kernel,
// which removes the earlier picks' rank-1 terms from that column, appends the new term u (sign sg) to the list and
// updates the diagonal and the entropy: O(n^2) per pick instead of two O(n^3) factorisations.
//   mode 0 (removal): u = p / sqrt(P_cc), sg = +1, H += -CONST + 1/2 log P_cc
//   mode 1 (noise):   u = q sqrt|g|,      sg = sign g, H += 1/2 log(1 + d Q_cc); also H_A += the pick's own entropy gain,
//                     read off the scale of its appended row (LazyPick): new site CONST - log scale, mobile-sampled
//                     1/2 log(-d) - log scale
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mi_rank1_kernel(int64_t m, const T* col0, T* U, int64_t ldu, double* sgn, int q, int64_t cpos,
                                                       int mode, double dlt, T* diag, double* H, double* H_A, const LazyPick* pick) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double pc = (double)col0[cpos];
    for (int r = 0; r < q; ++r) {
        const double uc = (double)U[(int64_t)r * ldu + cpos];
        pc -= sgn[r] * uc * uc;
    }
    double fac, sg;
    if (mode == 0) { fac = 1.0 / sqrt(pc); sg = 1.0; }
    else {
        const double g = dlt / (1.0 + dlt * pc);
        fac = sqrt(fabs(g));
        sg = g >= 0.0 ? 1.0 : -1.0;
    }
    if (j < m) {
        double cj = (double)col0[j];
        for (int r = 0; r < q; ++r) cj -= sgn[r] * (double)U[(int64_t)r * ldu + j] * (double)U[(int64_t)r * ldu + cpos];
        const double u = cj * fac;
        U[(int64_t)q * ldu + j] = (T)u;
        diag[j] = (T)((double)diag[j] - sg * u * u);
    }
    if (j == 0) {
        // every thread read sgn[0..q) and H before this block's thread 0 writes slot q / H: other blocks may still be
        // reading sgn[r < q] (untouched here); slot q and H are written once per launch
        sgn[q] = sg;
        *H += mode == 0 ? -ENT_CONST + 0.5 * log(pc) : 0.5 * log1p(dlt * pc);
        if (H_A) *H_A += (pick->in_train ? 0.5 * log(-dlt) : ENT_CONST) - log(pick->scale);
    }
}
template <typename T>
int mi_rank1_launch(algp_ctx* c, int64_t m, const T* col0, T* U, int64_t ldu, double* sgn, int q, int64_t cpos, int mode,
                    double dlt, T* diag, double* H, double* H_A, const LazyPick* pick) {
    hipLaunchKernelGGL(mi_rank1_kernel<T>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->cur, m, col0, U, ldu, sgn, q, cpos,
                       mode, dlt, diag, H, H_A, pick);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int mi_rank1_launch<double>(algp_ctx*, int64_t, const double*, double*, int64_t, double*, int, int64_t, int, double, double*,
                                     double*, double*, const LazyPick*);
template int mi_rank1_launch<float>(algp_ctx*, int64_t, const float*, float*, int64_t, double*, int, int64_t, int, double, float*,
                                    double*, double*, const LazyPick*);

// MI utility of every candidate from the device-resident terms: entropy gain + H(A) + H(Abar \ i) - H(all_i)
// Hs = (H(A), H(Abar), H(all)); posbar: pool index -> row of the complement matrix
template <typename T>
__global__ void mi_score_kernel(int64_t M, const int* ckind, const int64_t* cidx, const unsigned char* alive, const T* dstat,
                                double ss, double delta, const int64_t* posbar, const T* dP, const T* dQ, const double* Hs,
                                double* out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    double u = -INFINITY;
    if (alive[j]) {
        const bool unit = ckind[j] >= 0;
        const int64_t pj = cidx[j];
        u = entropy_utility((double)dstat[j], unit, ss, delta) + Hs[0];
        if (unit) u += Hs[1] - (Hs[2] + 0.5 * log1p(delta * (double)dQ[pj]));
        else u += (Hs[1] - ENT_CONST + 0.5 * log((double)dP[posbar[pj]])) - (Hs[2] + 0.5 * log1p(ss * (double)dQ[pj]));
    }
    out[j] = u;
}
template <typename T>
int mi_score_launch(algp_ctx* c, int64_t M, const int* ckind, const int64_t* cidx, const unsigned char* alive, const T* dstat,
                    double ss, double delta, const int64_t* posbar, const T* dP, const T* dQ, const double* Hs, double* out) {
    if (M <= 0) return ALGP_OK;
    ProfScope ps(c, ALGP_PROF_SCORE, 12.0 * M, (3.0 * sizeof(T) + 21.0) * M);
    hipLaunchKernelGGL(mi_score_kernel<T>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->cur, M, ckind, cidx, alive, dstat,
                       ss, delta, posbar, dP, dQ, Hs, out);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int mi_score_launch<double>(algp_ctx*, int64_t, const int*, const int64_t*, const unsigned char*, const double*, double,
                                     double, const int64_t*, const double*, const double*, const double*, double*);
template int mi_score_launch<float>(algp_ctx*, int64_t, const int*, const int64_t*, const unsigned char*, const float*, double,
                                    double, const int64_t*, const float*, const float*, const double*, double*);

// out[0] = fresh[*idx] (as a double; -1 when *idx < 0): rides with the argmax read-back of the lazy greedy
__global__ void fresh_at_kernel(const int* fresh, const int64_t* idx, double* out) {
    const int64_t i = *idx;
    *out = i >= 0 ? (double)fresh[i] : -1.0;
}
int fresh_at_launch(algp_ctx* c, const int* fresh, const int64_t* idx, double* out) {
    hipLaunchKernelGGL(fresh_at_kernel, dim3(1), dim3(1), 0, c->cur, fresh, idx, out);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

// ---------------------------------------------------------------------------------------------
// best_path block scoring (agent.py:358-403, entropy criterion): the gain of a whole path of mobile readings,
//   dH = H(A u P) - H(A) = |P| CONST + 1/2 log det G,   G = C_PP + sigma_m^2 I - V_P^T V_P   (|P| <= 64),
// from the rows of V^T the candidate solve left resident -- one workgroup per path, no factor update, no host round
// trip per path.  A path entry is a local candidate position; entries with lpos >= 0 are sites that already are
// train row lpos (a second, mobile reading of a statically sampled site): their column is C[A, j] = S e_lpos - d e_lpos,
// so V_j = L[lpos, :]^T - d_lpos u with u = the candidate's unit row.  Accumulation in fp64 for either dtype.
// ---------------------------------------------------------------------------------------------
constexpr int PATH_SITES = 64;
template <typename T, int DP>
__global__ __launch_bounds__(256) void path_score_kernel(const int64_t* cpos, const int64_t* lpos, int maxlen, const int64_t* cidx,
                                                         const T* Vt, int64_t ldv, int64_t ncols, const T* L, int64_t ldl,
                                                         const T* varA, const T* Xs, const T* Cp, int64_t n_pool, int kernel,
                                                         double os, double noise, double sm, double* out) {
    __shared__ double G[PATH_SITES][PATH_SITES + 1];
    __shared__ double R[PATH_SITES][33];
    __shared__ int64_t s_c[PATH_SITES], s_l[PATH_SITES];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int64_t* pc = cpos + (int64_t)blockIdx.x * maxlen;
    const int64_t* pl = lpos + (int64_t)blockIdx.x * maxlen;
    if (tid == 0) {
        int n = 0;
        for (int a = 0; a < maxlen && n < PATH_SITES; ++a)
            if (pc[a] >= 0) { s_c[n] = pc[a]; s_l[n] = pl[a]; ++n; }
        s_n = n;
    }
    __syncthreads();
    const int P = s_n;
    if (P == 0) {
        if (tid == 0) out[blockIdx.x] = 0.0;
        return;
    }
    // Gram matrix of the path's rows of V^T, 32 columns at a time; thread (ty, tx) owns G[ty + 16 i][tx + 16 j]
    const int ty = tid >> 4, tx = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int64_t k0 = 0; k0 < ncols; k0 += 32) {
        for (int e = tid; e < P * 32; e += 256) {
            const int a = e >> 5, k = e & 31;
            const int64_t kk = k0 + k;
            double v = 0.0;
            if (kk < ncols) {
                v = (double)Vt[s_c[a] * ldv + kk];
                const int64_t lp = s_l[a];
                if (lp >= 0) v = (kk <= lp ? (double)L[lp * ldl + kk] : 0.0) - (double)varA[lp] * v;
            }
            R[a][k] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = ty + 16 * i;
            if (a >= P) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = tx + 16 * j;
                if (b > a) continue;                            // lower triangle
                double s2 = 0.0;
#pragma unroll 8
                for (int k = 0; k < 32; ++k) s2 += R[a][k] * R[b][k];
                acc[i][j] += s2;
            }
        }
        __syncthreads();
    }
    // G = C_PP + sigma_m^2 I - Gram
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = ty + 16 * i;
        if (a >= P) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = tx + 16 * j;
            if (b > a) continue;
            const int64_t pa = cidx[s_c[a]], pb = cidx[s_c[b]];
            double cab;
            if (Cp) {
                cab = (double)Cp[pa * n_pool + pb];
            } else {
                double r2 = 0.0;
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    const double df = (double)Xs[pa * DP + d] - (double)Xs[pb * DP + d];
                    r2 += df * df;
                }
                if (kernel == ALGP_KERNEL_RBF) cab = os * kexp(-0.5 * r2);
                else {
                    const double r = sqrt(r2) * 1.7320508075688772;
                    cab = os * (1.0 + r) * kexp(-r);
                }
                if (pa == pb) cab += noise;
            }
            G[a][b] = cab + (a == b ? sm : 0.0) - acc[i][j];
        }
    }
    __syncthreads();
    // Cholesky of the P x P block in LDS, column by column; log det = 2 sum log diag
    double logdet = 0.0;
    for (int j = 0; j < P; ++j) {
        const double d = G[j][j];
        if (!(d > 0.0)) { logdet = NAN; break; }                   // wave-uniform: every thread reads the same G[j][j]
        const double rs = 1.0 / sqrt(d);
        logdet += log(d);
        __syncthreads();
        for (int i = j + 1 + tid; i < P; i += 256) G[i][j] *= rs;
        __syncthreads();
        for (int e = tid; e < (P - j - 1) * (P - j - 1); e += 256) {
            const int i = j + 1 + e / (P - j - 1), c2 = j + 1 + e % (P - j - 1);
            if (c2 <= i) G[i][c2] -= G[i][j] * G[c2][j];
        }
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = (double)P * ENT_CONST + 0.5 * logdet;
}

template <typename T>
int path_score_launch(algp_ctx* c, const int64_t* cpos, const int64_t* lpos, int npaths, int maxlen, const int64_t* cidx,
                      const T* Vt, int64_t ldv, int64_t ncols, const T* L, int64_t ldl, const T* varA, const T* Xs,
                      const T* Cp, int64_t n_pool, int DP, int kernel, double os, double noise, double sm, double* out) {
    if (npaths <= 0) return ALGP_OK;
    dim3 grid((unsigned)npaths), blk(256);
#define ALGP_PS(DPV)                                                                                                    \
    hipLaunchKernelGGL((path_score_kernel<T, DPV>), grid, blk, 0, c->cur, cpos, lpos, maxlen, cidx, Vt, ldv, ncols, L, ldl, \
                       varA, Xs, Cp, n_pool, kernel, os, noise, sm, out)
    if (DP == 2) ALGP_PS(2);
    else if (DP == 4) ALGP_PS(4);
    else ALGP_PS(8);
#undef ALGP_PS
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int path_score_launch<double>(algp_ctx*, const int64_t*, const int64_t*, int, int, const int64_t*, const double*, int64_t,
                                       int64_t, const double*, int64_t, const double*, const double*, const double*, int64_t, int,
                                       int, double, double, double, double*);
template int path_score_launch<float>(algp_ctx*, const int64_t*, const int64_t*, int, int, const int64_t*, const float*, int64_t,
                                      int64_t, const float*, int64_t, const float*, const float*, const float*, int64_t, int, int,
                                      double, double, double, double*);

// ---- paths of 65 .. 256 sites (config 5's field rows: up to ~250 sites on a 250 x 200 field, env.py:197-310) -------------
// The path's posterior block no longer fits LDS: its rows of V^T are gathered into a scratch (gather_rows_kernel, the
// second-reading transform included), the Gram matrices of a batch of paths are ONE batched MFMA product (gemm.hip), this
// kernel turns each into G = C_PP + sigma_m^2 I - Gram (identity on the padding) in place, and the blocks are factored
// as 2 x 2 tiles of 128 by the diagonal-block kernel + two batched tile products (api.hip: score_paths_big).
template <typename T, int DP>
__global__ __launch_bounds__(256) void path_assemble_kernel(const int64_t* cpos, int ppad, const int64_t* cidx, const T* Xs, const T* Cp,
                                                            int64_t n_pool, int kernel, double os, double noise, double sm, T* G) {
    __shared__ int64_t s_p[256];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int64_t* pc = cpos + (int64_t)blockIdx.x * ppad;
    if (tid == 0) {
        int n = 0;
        while (n < ppad && pc[n] >= 0) ++n;                        // the host packs a path's sites to the front
        s_n = n;
    }
    if (tid < ppad) s_p[tid] = pc[tid] >= 0 ? cidx[pc[tid]] : -1;
    __syncthreads();
    const int P = s_n;
    T* Gp = G + (int64_t)blockIdx.x * ppad * ppad;
    for (int e = tid; e < ppad * ppad; e += 256) {
        const int a = e / ppad, b = e - a * ppad;
        if ((a >> 7) < (b >> 7)) continue;                         // tiles above the diagonal are never read
        T v;
        if (a < P && b < P) {
            const int64_t pa = s_p[a], pb = s_p[b];
            double cab;
            if (Cp) {
                cab = (double)Cp[pa * n_pool + pb];
            } else {
                double r2 = 0.0;
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    const double df = (double)Xs[pa * DP + d] - (double)Xs[pb * DP + d];
                    r2 += df * df;
                }
                if (kernel == ALGP_KERNEL_RBF) cab = os * kexp(-0.5 * r2);
                else {
                    const double r = sqrt(r2) * 1.7320508075688772;
                    cab = os * (1.0 + r) * kexp(-r);
                }
                if (pa == pb) cab += noise;
            }
            v = (T)(cab + (a == b ? sm : 0.0) - (double)Gp[e]);
        } else {
            v = a == b ? (T)1 : (T)0;
        }
        Gp[e] = v;
    }
}
template <typename T>
int path_assemble_launch(algp_ctx* c, const int64_t* cpos, int batch, int ppad, const int64_t* cidx, const T* Xs, const T* Cp,
                         int64_t n_pool, int DP, int kernel, double os, double noise, double sm, T* G) {
    if (batch <= 0) return ALGP_OK;
    dim3 grid((unsigned)batch), blk(256);
#define ALGP_PA(DPV)                                                                                                      \
    hipLaunchKernelGGL((path_assemble_kernel<T, DPV>), grid, blk, 0, c->cur, cpos, ppad, cidx, Xs, Cp, n_pool, kernel, os, noise, sm, G)
    if (DP == 2) ALGP_PA(2);
    else if (DP == 4) ALGP_PA(4);
    else ALGP_PA(8);
#undef ALGP_PA
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int path_assemble_launch<double>(algp_ctx*, const int64_t*, int, int, const int64_t*, const double*, const double*, int64_t, int, int,
                                          double, double, double, double*);
template int path_assemble_launch<float>(algp_ctx*, const int64_t*, int, int, const int64_t*, const float*, const float*, int64_t, int, int,
                                         double, double, double, float*);
// out[p] = P CONST + 1/2 log det (NaN where a pivot was not positive)
__global__ void path_finish_kernel(const int64_t* cpos, int ppad, int batch, const double* logdet, const int* info, double* out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= batch) return;
    int n = 0;
    const int64_t* pc = cpos + (int64_t)p * ppad;
    while (n < ppad && pc[n] >= 0) ++n;
    out[p] = info[p] != 0 ? NAN : (double)n * ENT_CONST + 0.5 * logdet[p];
}
int path_finish_launch(algp_ctx* c, const int64_t* cpos, int ppad, int batch, const double* logdet, const int* info, double* out) {
    if (batch <= 0) return ALGP_OK;
    hipLaunchKernelGGL(path_finish_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, c->cur, cpos, ppad, batch, logdet, info, out);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

// ---------------------------------------------------------------------------------------------
// fused kernel-GEMV: mu_j = ybar + sum_a k(x_j, x_a) alpha_a, K never materialised (utils.py:301).
// One wave per output; lanes stride over the train set (coordinates and alpha are L2 resident).
// ---------------------------------------------------------------------------------------------
template <typename T, int DP>
__global__ __launch_bounds__(256) void kgemv_kernel(int64_t M, const int64_t* qidx, const T* Xs, int64_t N,
                                                    const int64_t* aidx, const T* alpha, int kernel, T os, T ybar,
                                                    T* mu) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t j = wave; j < M; j += nw) {
        const int64_t pj = qidx[j];
        T xj[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) xj[d] = Xs[pj * DP + d];
        T s = (T)0;
        for (int64_t a = lane; a < N; a += 64) {
            const int64_t pa = aidx[a];
            T r2 = (T)0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const T df = xj[d] - Xs[pa * DP + d];
                r2 += df * df;
            }
            T kv;
            if (kernel == ALGP_KERNEL_RBF) kv = os * kexp((T)-0.5 * r2);
            else {
                const T r = sqrt(r2) * (T)1.7320508075688772;
                kv = os * ((T)1 + r) * kexp(-r);
            }
            s += kv * alpha[a];
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) mu[j] = ybar + s;
    }
}

template <typename T>
int kgemv_launch(algp_ctx* c, int64_t M, const int64_t* qidx, const T* Xs, int DP, int64_t N, const int64_t* aidx,
                 const T* alpha, int kernel, T os, T ybar, T* mu) {
    if (M <= 0) return ALGP_OK;
    int64_t g = (M + 3) / 4;
    if (g > 8192) g = 8192;
    ProfScope ps(c, ALGP_PROF_KMAT, (double)M * N * (3.0 * DP + 4.0), sizeof(T) * (double)(M + N) * (DP + 1));
    dim3 grid((unsigned)g), blk(256);
    if (DP == 2) hipLaunchKernelGGL((kgemv_kernel<T, 2>), grid, blk, 0, c->cur, M, qidx, Xs, N, aidx, alpha, kernel, os, ybar, mu);
    else if (DP == 4) hipLaunchKernelGGL((kgemv_kernel<T, 4>), grid, blk, 0, c->cur, M, qidx, Xs, N, aidx, alpha, kernel, os, ybar, mu);
    else hipLaunchKernelGGL((kgemv_kernel<T, 8>), grid, blk, 0, c->cur, M, qidx, Xs, N, aidx, alpha, kernel, os, ybar, mu);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int kgemv_launch<double>(algp_ctx*, int64_t, const int64_t*, const double*, int, int64_t, const int64_t*,
                                  const double*, int, double, double, double*);
template int kgemv_launch<float>(algp_ctx*, int64_t, const int64_t*, const float*, int, int64_t, const int64_t*,
                                 const float*, int, float, float, float*);

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pad_identity_kernel(T* A, int64_t n, int64_t npad, int64_t ld) {
    // rows/cols >= n of an npad x npad matrix become identity (the interior is left alone)
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npad * npad) return;
    const int64_t i = e / npad, j = e - i * npad;
    if (i >= n || j >= n) A[i * ld + j] = (i == j) ? (T)1 : (T)0;
}
template <typename T>
int pad_identity_launch(algp_ctx* c, T* A, int64_t n, int64_t npad, int64_t ld) {
    const int64_t tot = npad * npad;
    hipLaunchKernelGGL(pad_identity_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->cur, A, n, npad, ld);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int pad_identity_launch<double>(algp_ctx*, double*, int64_t, int64_t, int64_t);
template int pad_identity_launch<float>(algp_ctx*, float*, int64_t, int64_t, int64_t);

// A[i][i] += v for i < n
template <typename T>
__global__ void add_diag_kernel(T* A, int64_t n, int64_t ld, T v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i * ld + i] += v;
}
template <typename T>
int add_diag_launch(algp_ctx* c, T* A, int64_t n, int64_t ld, T v) {
    if (n <= 0) return ALGP_OK;
    hipLaunchKernelGGL(add_diag_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->cur, A, n, ld, v);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int add_diag_launch<double>(algp_ctx*, double*, int64_t, int64_t, double);
template int add_diag_launch<float>(algp_ctx*, float*, int64_t, int64_t, float);

template <typename T>
__global__ void set_identity_kernel(T* A, int64_t npad, int64_t ld) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npad * npad) return;
    const int64_t i = e / npad, j = e - i * npad;
    A[i * ld + j] = (i == j) ? (T)1 : (T)0;
}
template <typename T>
int set_identity_launch(algp_ctx* c, T* A, int64_t npad, int64_t ld) {
    const int64_t tot = npad * npad;
    hipLaunchKernelGGL(set_identity_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->cur, A, npad, ld);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int set_identity_launch<double>(algp_ctx*, double*, int64_t, int64_t);
template int set_identity_launch<float>(algp_ctx*, float*, int64_t, int64_t);

// sum_i log L[i][i] for i < n, accumulated into *out (double)
// out[i] = sum_{k >= 128 (i / 128)} X[i][k] z[k] for an upper-triangular X (zero tiles left of the diagonal tile are never read):
// alpha = L^-T z from the X = L^-T a fit iteration's launch leaves behind.  A workgroup per row, the waves' quarters added in
// wave order (the same bits in every run).
template <typename T>
__global__ __launch_bounds__(256) void upper_gemv_kernel(const T* X, int64_t ld, int64_t n, const T* z, T* out) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ T red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x, k0 = i / 128 * 128;
    const T* row = X + i * ld;
    const int64_t nv = (n - k0) / VEC, per = (nv + 3) / 4;
    const int64_t v0 = wave * per, v1 = v0 + per < nv ? v0 + per : nv;
    T s = (T)0;
    for (int64_t v = v0 + lane; v < v1; v += 64) {
        const vec_t x = *reinterpret_cast<const vec_t*>(row + k0 + v * VEC);
        const vec_t a = *reinterpret_cast<const vec_t*>(z + k0 + v * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += x[e] * a[e];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = (red[0] + red[1]) + (red[2] + red[3]);
}
template <typename T>
int upper_gemv_launch(algp_ctx* c, const T* X, int64_t ld, int64_t n, const T* z, T* out) {
    if (n <= 0) return ALGP_OK;
    ProfScope ps(c, ALGP_PROF_TRSV, (double)n * n, sizeof(T) * (double)n * n / 2.0);
    hipLaunchKernelGGL(upper_gemv_kernel<T>, dim3((unsigned)n), dim3(256), 0, c->cur, X, ld, n, z, out);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int upper_gemv_launch<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, double*);
template int upper_gemv_launch<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, float*);

// acc3[q * stride + rows[e]] = 0 for q < 3: the carried row sums of candidates whose kept columns were zeroed
template <typename T>
__global__ void zero_rows3_kernel(T* acc3, int64_t stride, const int64_t* rows, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 3 * n) acc3[(e / n) * stride + rows[e % n]] = (T)0;
}
template <typename T>
int zero_rows3_launch(algp_ctx* c, T* acc3, int64_t stride, const int64_t* rows, int64_t n) {
    if (n <= 0) return ALGP_OK;
    hipLaunchKernelGGL(zero_rows3_kernel<T>, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, c->cur, acc3, stride, rows, n);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int zero_rows3_launch<double>(algp_ctx*, double*, int64_t, const int64_t*, int64_t);
template int zero_rows3_launch<float>(algp_ctx*, float*, int64_t, const int64_t*, int64_t);

// X[rows[r]][0:ncols] = 0 for n listed rows (a workgroup per row and 2048 columns): the candidates that became train sites
// in an incremental step restart as unit rows (one memset per row before: ~30 per step of config 5's loop, 0.2 ms in all).
template <typename T>
__global__ __launch_bounds__(256) void zero_listed_rows_kernel(T* X, int64_t ldx, const int64_t* rows, int64_t ncols) {
    T* d = X + rows[blockIdx.x] * ldx;
    const int64_t end = ncols < ((int64_t)blockIdx.y + 1) * 2048 ? ncols : ((int64_t)blockIdx.y + 1) * 2048;
    for (int64_t k = (int64_t)blockIdx.y * 2048 + threadIdx.x; k < end; k += 256) d[k] = (T)0;
}
template <typename T>
int zero_listed_rows_launch(algp_ctx* c, T* X, int64_t ldx, const int64_t* rows, int64_t n, int64_t ncols) {
    if (n <= 0 || ncols <= 0) return ALGP_OK;
    hipLaunchKernelGGL(zero_listed_rows_kernel<T>, dim3((unsigned)n, (unsigned)((ncols + 2047) / 2048)), dim3(256), 0, c->cur, X, ldx, rows, ncols);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int zero_listed_rows_launch<double>(algp_ctx*, double*, int64_t, const int64_t*, int64_t, int64_t);
template int zero_listed_rows_launch<float>(algp_ctx*, float*, int64_t, const int64_t*, int64_t, int64_t);

// Fixed assignment and fixed summation order: the log-determinant of an updated factor is the same number in every run.
// Workgroup b takes entries i = 256 (b + G q) + thread; its 4 wave sums go to part[4 b ..]; logdiag_sum_kernel adds the 4 G
// partials in index order.  (One workgroup walked the 50 000 diagonal entries of config 5's factor -- one cache line each --
// in 157 us of every planning step; 64 take 12.)
constexpr int LOGDIAG_G = 64;
template <typename T>
__global__ __launch_bounds__(256) void logdiag_kernel(const T* L, int64_t ld, int64_t n, double* part) {
    double v = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 256 * LOGDIAG_G) v += log((double)L[i * ld + i]);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[4 * blockIdx.x + (threadIdx.x >> 6)] = v;
}
__global__ void logdiag_sum_kernel(const double* part, double* out) {
    double s = 0;
    for (int q = 0; q < 4 * LOGDIAG_G; ++q) s += part[q];
    *out += s;
}
template <typename T>
int logdiag_launch(algp_ctx* c, const T* L, int64_t ld, int64_t n, double* out) {
    if (n <= 0) return ALGP_OK;
    ALGP_TRY(ensure(c, c->ldpart, sizeof(double) * 4 * LOGDIAG_G));
    hipLaunchKernelGGL(logdiag_kernel<T>, dim3(LOGDIAG_G), dim3(256), 0, c->cur, L, ld, n, (double*)c->ldpart.p);
    ALGP_HIP(hipGetLastError());
    hipLaunchKernelGGL(logdiag_sum_kernel, dim3(1), dim3(1), 0, c->cur, (const double*)c->ldpart.p, out);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int logdiag_launch<double>(algp_ctx*, const double*, int64_t, int64_t, double*);
template int logdiag_launch<float>(algp_ctx*, const float*, int64_t, int64_t, double*);

template <typename T>
__global__ void add_doubles_kernel(double* dst, const T* src, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = (double)src[e];
}
template <typename T>
int to_double_launch(algp_ctx* c, double* dst, const T* src, int64_t n) {
    if (n <= 0) return ALGP_OK;
    hipLaunchKernelGGL(add_doubles_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->cur, dst, src, n);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int to_double_launch<double>(algp_ctx*, double*, const double*, int64_t);
template int to_double_launch<float>(algp_ctx*, double*, const float*, int64_t);

}  // namespace algp
