// kmat.hip -- kernel-matrix build: ScaleKernel(RBF-ARD) / ScaleKernel(Matern-1.5) values, the
// per-point white-noise diagonal and the likelihood-noise diagonal in one pass.
//
// Replaces GPR.cov_mat (reference models.py:161-181): gpytorch materialises an n1 x n2 x D
// difference tensor, then np.diag(var) and np.eye(n) allocate two more dense n x n arrays just to
// add a diagonal (models.py:175-180).  Here every output element is produced once, in registers,
// and stored once: the kernel is HBM-write bound (s * n1 * n2 bytes; 4.6 TB/s of stores at the bench's sizes since
// round 4: 4 KiB-contiguous rows per workgroup, the rows' coordinates staged in LDS, the short fp64 exp of common.h --
// 2.5 -> 2.0 ms per step; non-temporal stores and deeper unrolling: no further change).
//
// Layout: coordinates are pre-scaled by 1/lengthscale and zero padded to DP in {2,4,8} columns
// (scale_coords), so the distance loop is fully unrolled.  A 256-thread workgroup produces a
// 32-row x (256*VEC)-column tile: each lane owns VEC = 16 B / sizeof(T) adjacent columns (its
// column coordinates stay in registers), the four waves sit side by side and walk the same 32 rows
// (coordinates are wave-uniform loads): every row of the tile is written as 4 KiB contiguous.
#include "common.h"

namespace algp {

template <typename T>
struct Vec16;
template <>
struct Vec16<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int N = 2;
};
template <>
struct Vec16<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int N = 4;
};

template <typename T>
struct KmatArgs {
    const T* X1;            // scaled coords of the row side   (n x DP)
    const T* X2;            // scaled coords of the column side
    const T* Cp;            // explicit pool covariance or null
    int64_t n_pool;
    const int64_t* ridx;    // row -> pool index (null: identity)
    const int64_t* cidx;    // col -> pool index (null: identity)
    int64_t rows, rows_pad, cols, cols_pad;
    int64_t row_base;       // first row handled by this launch (gridDim.y is limited to 65535)
    const T* diag_add;      // per-row value added where pool indices coincide (null: none)
    T noise_on_equal;       // added where pool indices coincide
    int same_pool;          // row and column indices address the same pool (equality is meaningful)
    const int* unit;        // unit[r] >= 0: row r is the unit vector e_{unit[r]} (null: none)
    int identity_pad;
    int64_t ident_shift;    // identity padding sits at column r + ident_shift (row windows of a larger matrix)
    int64_t col_shift;      // global column of local column 0 (unit rows compare global columns)
    int kernel;
    T outputscale;
    T* out;
    int64_t ldo;
};

template <typename T>
__device__ __forceinline__ T kval(int kernel, T os, T r2) {
    if (kernel == ALGP_KERNEL_RBF) return os * kexp((T)-0.5 * r2);
    const T r = sqrt(r2) * (T)1.7320508075688772;
    return os * ((T)1 + r) * kexp(-r);
}

// EQ: entries whose row and column address the same pool site receive a diagonal term (the symmetric builds); without it
// (the candidates' B^T, cross matrices) the index comparisons are not even compiled in -- they and the other per-element
// bookkeeping were half of the kernel's VALU instructions, and the kernel is VALU-issue bound (profiles/r04_valu_by_kernel.json).
template <typename T, int DP, bool EQ>
__global__ __launch_bounds__(256) void kmat_kernel(KmatArgs<T> a) {
    constexpr int VEC = Vec16<T>::N;
    using vec_t = typename Vec16<T>::type;
    // the four waves of a workgroup sit side by side on the same 32 rows: every row of the tile is 4 KiB contiguous
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = (((int64_t)blockIdx.x * 4 + wave) * 64 + lane) * VEC;
    const int64_t r0 = a.row_base + (int64_t)blockIdx.y * 32;

    T xc[VEC][DP];
    int64_t pc[VEC];
    bool live[VEC];                                                // column inside the matrix (not padding): per lane, per launch
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int64_t c = c0 + v;
        pc[v] = -1;
        live[v] = c < a.cols;
        if (live[v]) {
            pc[v] = a.cidx ? a.cidx[c] : c;
            if (!a.Cp) {
#pragma unroll
                for (int d = 0; d < DP; ++d) xc[v][d] = a.X2[pc[v] * DP + d];
            }
        } else {
#pragma unroll
            for (int d = 0; d < DP; ++d) xc[v][d] = (T)0;
        }
    }
    // the 32 rows' unit flags, pool indices and coordinates once per workgroup into LDS: the row loop then has no
    // dependent global loads (unit -> pool index -> coordinates were three serial scalar loads per row and wave)
    __shared__ int s_u[32];
    __shared__ int64_t s_p[32];
    __shared__ T s_x[32][DP];
    if (threadIdx.x < 32) {
        const int64_t r = r0 + threadIdx.x;
        int u = -1;
        int64_t pr = 0;
        if (r < a.rows) {
            u = a.unit ? a.unit[r] : -1;
            pr = a.ridx ? a.ridx[r] : r;
        }
        s_u[threadIdx.x] = u;
        s_p[threadIdx.x] = pr;
#pragma unroll
        for (int d = 0; d < DP; ++d) s_x[threadIdx.x][d] = (r < a.rows && u < 0 && !a.Cp) ? a.X1[pr * DP + d] : (T)0;
    }
    __syncthreads();
    if (c0 >= a.cols_pad) return;
    const int64_t cend = c0 + VEC;                                 // this lane's columns are [c0, cend)
#pragma unroll 2
    for (int rr = 0; rr < 32; ++rr) {
        const int64_t r = r0 + rr;
        if (r >= a.rows_pad) break;
        vec_t o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = (T)0;
        // the row's facts are wave-uniform: told so to the compiler, its branches become scalar branches
        const int u = __builtin_amdgcn_readfirstlane(s_u[rr]);
        if (r < a.rows && u < 0) {
            const int64_t pr = s_p[rr];
            if (a.Cp) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (live[v]) o[v] = a.Cp[pr * a.n_pool + pc[v]];
            } else {
                T xr[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) xr[d] = s_x[rr][d];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    T r2 = (T)0;
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        const T df = xr[d] - xc[v][d];
                        r2 += df * df;
                    }
                    const T val = kval<T>(a.kernel, a.outputscale, r2);
                    o[v] = live[v] ? val : (T)0;
                }
            }
            if (EQ) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (live[v] && a.same_pool && pr == pc[v]) {
                        o[v] += a.noise_on_equal;
                        // the per-row term only on the row's own diagonal entry: two train rows may address the
                        // same site (independent measurements), and their cross entry is C(i,i) without it
                        if (a.diag_add && c0 + v == r + a.ident_shift) o[v] += a.diag_add[r];
                    }
            }
        } else if (r < a.rows) {
            // a unit row e_u (a candidate that is a train site): one 1 in global column u
            const int64_t cu = (int64_t)u - a.col_shift;
            if (cu >= c0 && cu < cend && cu < a.cols) o[(int)(cu - c0)] = (T)1;
        }
        if (a.identity_pad) {
            // the padded diagonal: (r, r + shift) for rows and columns beyond the matrix
            const int64_t cd = r + a.ident_shift;
            if (cd >= c0 && cd < cend && (cd >= a.cols || r >= a.rows)) o[(int)(cd - c0)] = (T)1;
        }
        if (cend <= a.cols_pad) {
            *reinterpret_cast<vec_t*>(a.out + r * a.ldo + c0) = o;
        } else {                                                   // a column count that is no multiple of VEC (a window that starts
#pragma unroll                                                     // at an arbitrary column of V^T): never store beyond it
            for (int v = 0; v < VEC; ++v)
                if (c0 + v < a.cols_pad) a.out[r * a.ldo + c0 + v] = o[v];
        }
    }
}

template <typename T>
static int kmat_dispatch(algp_ctx* c, const KmatArgs<T>& a, int DP) {
    constexpr int VEC = Vec16<T>::N;
    if (a.rows_pad <= 0 || a.cols_pad <= 0) return ALGP_OK;
    const int64_t gx = (a.cols_pad + 256 * VEC - 1) / (256 * VEC);
    const int64_t gy = (a.rows_pad + 31) / 32;
    const double elems = (double)a.rows_pad * (double)a.cols_pad;
    ProfScope ps(c, ALGP_PROF_KMAT, elems * (3.0 * DP + 2.0), sizeof(T) * elems);
    const int64_t ymax = 65535;
    for (int64_t y0 = 0; y0 < gy; y0 += ymax) {
        KmatArgs<T> b = a;
        const int64_t ny = (gy - y0 < ymax) ? gy - y0 : ymax;
        b.row_base = y0 * 32;
        dim3 grid((unsigned)gx, (unsigned)ny);
        const bool eq = a.same_pool && (a.noise_on_equal != (T)0 || a.diag_add != nullptr);
        if (eq) {
            if (DP == 2) hipLaunchKernelGGL((kmat_kernel<T, 2, true>), grid, dim3(256), 0, c->cur, b);
            else if (DP == 4) hipLaunchKernelGGL((kmat_kernel<T, 4, true>), grid, dim3(256), 0, c->cur, b);
            else hipLaunchKernelGGL((kmat_kernel<T, 8, true>), grid, dim3(256), 0, c->cur, b);
        } else {
            if (DP == 2) hipLaunchKernelGGL((kmat_kernel<T, 2, false>), grid, dim3(256), 0, c->cur, b);
            else if (DP == 4) hipLaunchKernelGGL((kmat_kernel<T, 4, false>), grid, dim3(256), 0, c->cur, b);
            else hipLaunchKernelGGL((kmat_kernel<T, 8, false>), grid, dim3(256), 0, c->cur, b);
        }
    }
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

template <typename T>
int kmat_launch(algp_ctx* c, const KmatSrc& s, const int64_t* ridx, int64_t rows, int64_t rows_pad,
                const int64_t* cidx, int64_t cols, int64_t cols_pad, const T* diag_add, int add_noise_on_equal,
                const int* unit, int identity_pad, T* out, int64_t ldo, int64_t ident_shift, int64_t col_shift) {
    KmatArgs<T> a;
    a.X1 = (const T*)s.Xs;
    a.X2 = (const T*)s.Xs;
    a.Cp = (const T*)s.Cp;
    a.n_pool = s.n_pool;
    a.ridx = ridx;
    a.cidx = cidx;
    a.rows = rows; a.rows_pad = rows_pad; a.cols = cols; a.cols_pad = cols_pad;
    a.diag_add = diag_add;
    a.noise_on_equal = (T)(add_noise_on_equal ? s.noise : 0.0);
    a.same_pool = 1;
    a.unit = unit;
    a.identity_pad = identity_pad;
    a.ident_shift = ident_shift;
    a.col_shift = col_shift;
    a.kernel = s.kernel;
    a.outputscale = (T)s.outputscale;
    a.out = out;
    a.ldo = ldo;
    return kmat_dispatch<T>(c, a, s.DP);
}
template int kmat_launch<double>(algp_ctx*, const KmatSrc&, const int64_t*, int64_t, int64_t, const int64_t*, int64_t,
                                 int64_t, const double*, int, const int*, int, double*, int64_t, int64_t, int64_t);
template int kmat_launch<float>(algp_ctx*, const KmatSrc&, const int64_t*, int64_t, int64_t, const int64_t*, int64_t,
                                int64_t, const float*, int, const int*, int, float*, int64_t, int64_t, int64_t);

template <typename T>
int kmat_xy_launch(algp_ctx* c, const T* xs1, int64_t n1, const T* xs2, int64_t n2, int symmetric,
                   const T* diag_add, double add_noise, T* out, int64_t ldo) {
    KmatArgs<T> a;
    a.X1 = xs1;
    a.X2 = symmetric ? xs1 : xs2;
    a.Cp = nullptr;
    a.n_pool = 0;
    a.ridx = nullptr;
    a.cidx = nullptr;
    a.rows = n1; a.rows_pad = n1; a.cols = n2; a.cols_pad = round_up(n2, Vec16<T>::N);
    a.diag_add = symmetric ? diag_add : nullptr;
    a.noise_on_equal = (T)(symmetric ? add_noise : 0.0);
    a.same_pool = symmetric;
    a.unit = nullptr;
    a.identity_pad = 0;
    a.ident_shift = 0;
    a.col_shift = 0;
    a.kernel = c->hyp.kernel;
    a.outputscale = (T)c->hyp.outputscale;
    a.out = out;
    a.ldo = ldo;
    return kmat_dispatch<T>(c, a, c->hyp.DP);
}
template int kmat_xy_launch<double>(algp_ctx*, const double*, int64_t, const double*, int64_t, int, const double*,
                                    double, double*, int64_t);
template int kmat_xy_launch<float>(algp_ctx*, const float*, int64_t, const float*, int64_t, int, const float*, double,
                                   float*, int64_t);

// xs[i][d] = x[i][d] / lengthscale_d for d < D, 0 for D <= d < DP
struct InvLs {
    double v[MAXD];
};

template <typename T>
__global__ void scale_coords_kernel(const T* x, int64_t n, int D, int DP, InvLs inv_ls8, T* xs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * DP) return;
    const int64_t p = i / DP;
    const int d = (int)(i - p * DP);
    xs[i] = d < D ? x[p * D + d] * (T)inv_ls8.v[d] : (T)0;
}

template <typename T>
int scale_coords_launch(algp_ctx* c, const T* x, int64_t n, T* xs) {
    if (n <= 0) return ALGP_OK;
    InvLs il;
    for (int d = 0; d < MAXD; ++d) il.v[d] = c->hyp.inv_ls[d];
    const int64_t tot = n * c->hyp.DP;
    hipLaunchKernelGGL(scale_coords_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->cur, x, n,
                       c->hyp.D, c->hyp.DP, il, xs);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int scale_coords_launch<double>(algp_ctx*, const double*, int64_t, double*);
template int scale_coords_launch<float>(algp_ctx*, const float*, int64_t, float*);

}  // namespace algp
