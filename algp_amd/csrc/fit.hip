// fit.hip -- gradient of the exact-GP marginal log likelihood w.r.t. the D+2 log hyper-parameters
// (what gpytorch's autograd computes inside the reference's fit loop, models.py:145-158).
//
//   MLL = -1/2 y0' S^-1 y0 - 1/2 log det S - N/2 log 2pi,   S = K + diag(var) + sigma_n^2 I
//   dMLL/dtheta = 1/2 tr(W dS/dtheta),  W = alpha alpha' - S^-1
//   dS/dlog os = K ; dS/dlog ls_d = K .* u_d^2 (RBF) | 3 os e^{-a} u_d^2 (Matern-1.5), u_d = (x_d-x'_d)/ls_d
//   dS/dlog sigma_n^2 = sigma_n^2 I
// S^-1 = X X' with X = L^-T (trinv_upper + syrk_upper on the MFMA GEMM, both skipping X's zero half:
// N^3/6 + N^3/6 multiply-adds);
// the pairwise reduction below then streams the lower triangle of S^-1 once (HBM-bound, s*N^2/2
// bytes) while K and u_d are recomputed from the scaled coordinates.
#include "common.h"

namespace algp {

template <typename T, int DP>
__global__ __launch_bounds__(256) void mll_grad_kernel(const T* Sinv, int64_t ld, int64_t N, const T* Xs,
                                                       const int64_t* aidx, const T* alpha, int kernel, T os,
                                                       double* partial /* per workgroup: [0]=os, [1]=noise trace, [2..2+DP) = ls */) {
    // one workgroup = a 64-row x 64-col tile of the lower triangle; the grid enumerates only those
    // (blockIdx.x -> (bi, bj <= bi)).  The 64 rows' coordinates and alpha are staged in LDS once.
    int bi = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((int64_t)(bi + 1) * (bi + 2) / 2 <= (int64_t)blockIdx.x) ++bi;
    while ((int64_t)bi * (bi + 1) / 2 > (int64_t)blockIdx.x) --bi;
    const int bj = (int)(blockIdx.x - (int64_t)bi * (bi + 1) / 2);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // tx: column, ty: 4 row groups of 16
    __shared__ T s_x[64][DP];
    __shared__ T s_a[64];
    if (threadIdx.x < 64) {
        const int64_t i = (int64_t)bi * 64 + threadIdx.x;
        const bool ok = i < N;
        const int64_t pi = ok ? aidx[i] : 0;
#pragma unroll
        for (int d = 0; d < DP; ++d) s_x[threadIdx.x][d] = ok ? Xs[pi * DP + d] : (T)0;
        s_a[threadIdx.x] = ok ? alpha[i] : (T)0;
    }
    const int64_t j = (int64_t)bj * 64 + tx;
    double g_os = 0, g_tr = 0, g_ls[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) g_ls[d] = 0;
    T xj[DP], aj = (T)0;
#pragma unroll
    for (int d = 0; d < DP; ++d) xj[d] = (T)0;
    if (j < N) {
        const int64_t pj = aidx[j];
#pragma unroll
        for (int d = 0; d < DP; ++d) xj[d] = Xs[pj * DP + d];
        aj = alpha[j];
    }
    __syncthreads();
    if (j < N) {
        T sv[16];                                                // this thread's 16 entries of S^-1, loaded up front
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int64_t i = (int64_t)bi * 64 + ty * 16 + rr;
            sv[rr] = (i < N && j <= i) ? Sinv[i * ld + j] : (T)0;
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int li = ty * 16 + rr;
            const int64_t i = (int64_t)bi * 64 + li;
            if (i >= N || j > i) continue;
            T u2[DP], r2 = (T)0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const T df = s_x[li][d] - xj[d];
                u2[d] = df * df;
                r2 += u2[d];
            }
            const T w = s_a[li] * aj - sv[rr];
            const double m = (i == j) ? 1.0 : 2.0;               // symmetric: off-diagonal pairs count twice
            T kv, dk;                                            // dk * u_d^2 = dK/dlog ls_d
            if (kernel == ALGP_KERNEL_RBF) {
                kv = os * kexp((T)-0.5 * r2);
                dk = kv;
            } else {
                const T a = sqrt(r2) * (T)1.7320508075688772;
                const T e = kexp(-a);
                kv = os * ((T)1 + a) * e;
                dk = (T)3 * os * e;
            }
            g_os += m * (double)(w * kv);
            if (i == j) g_tr += (double)w;
#pragma unroll
            for (int d = 0; d < DP; ++d) g_ls[d] += m * (double)(w * dk * u2[d]);
        }
    }
    // block reduction: wave shuffles, LDS across the 4 waves, then one partial per block and quantity (summed in fixed
    // order by mll_grad_reduce_kernel: the gradient, and with it a whole fit trajectory, is the same in every run)
    auto wsum = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        return v;
    };
    __shared__ double s_red[4][2 + DP];
    g_os = wsum(g_os);
    g_tr = wsum(g_tr);
#pragma unroll
    for (int d = 0; d < DP; ++d) g_ls[d] = wsum(g_ls[d]);
    if (tx == 0) {
        s_red[ty][0] = g_os;
        s_red[ty][1] = g_tr;
#pragma unroll
        for (int d = 0; d < DP; ++d) s_red[ty][2 + d] = g_ls[d];
    }
    __syncthreads();
    if (threadIdx.x < 2 + DP)
        partial[(int64_t)blockIdx.x * (2 + DP) + threadIdx.x] =
            (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// out[q] += sum over blocks of partial[b][q], q < nq: one workgroup, thread-strided then shuffle tree
__global__ __launch_bounds__(256) void mll_grad_reduce_kernel(const double* partial, int64_t nblocks, int nq, double* out) {
    __shared__ double part[4];
    for (int q = 0; q < nq; ++q) {
        double v = 0;
        for (int64_t b = threadIdx.x; b < nblocks; b += 256) v += partial[b * nq + q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) out[q] += (part[0] + part[1]) + (part[2] + part[3]);
        __syncthreads();
    }
}

template <typename T>
int mll_grad_launch(algp_ctx* c, const T* Sinv, int64_t ld, int64_t N, const T* Xs, int DP, const int64_t* aidx,
                    const T* alpha, int kernel, T os, double* out_dev, double* partial) {
    if (N <= 0) return ALGP_OK;
    const unsigned nb = (unsigned)((N + 63) / 64);
    ProfScope ps(c, ALGP_PROF_KMAT, 0.5 * (double)N * N * (3.0 * DP + 8.0), sizeof(T) * 0.5 * (double)N * N);
    dim3 grid(nb * (nb + 1) / 2), blk(256);
    if (DP == 2) hipLaunchKernelGGL((mll_grad_kernel<T, 2>), grid, blk, 0, c->cur, Sinv, ld, N, Xs, aidx, alpha, kernel, os, partial);
    else if (DP == 4) hipLaunchKernelGGL((mll_grad_kernel<T, 4>), grid, blk, 0, c->cur, Sinv, ld, N, Xs, aidx, alpha, kernel, os, partial);
    else hipLaunchKernelGGL((mll_grad_kernel<T, 8>), grid, blk, 0, c->cur, Sinv, ld, N, Xs, aidx, alpha, kernel, os, partial);
    ALGP_HIP(hipGetLastError());
    hipLaunchKernelGGL(mll_grad_reduce_kernel, dim3(1), dim3(256), 0, c->cur, partial, (int64_t)grid.x, 2 + DP, out_dev);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int mll_grad_launch<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, int, const int64_t*,
                                     const double*, int, double, double*, double*);
template int mll_grad_launch<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, int, const int64_t*,
                                    const float*, int, float, double*, double*);

}  // namespace algp
