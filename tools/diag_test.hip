// Stand-alone check + timing of the 128 x 128 diagonal-block routine (algp_amd/csrc/diag.h) against a
// plain CPU Cholesky / triangular inverse, fp64 and fp32, with in-kernel cycle stamps.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DALGP_POTRF_STAMPS tools/diag_test.hip -o build/diag_test
#include "../algp_amd/csrc/diag.h"
#include <stdio.h>
#include <math.h>
#include <vector>

using namespace algp;

template <typename T, bool FACTOR>
__global__ __launch_bounds__(256) void diag_kernel(T* A, int64_t lda, T* inv_out, double* logdet_acc, int* info) {
    __shared__ DiagShared<T> sh;
    diag128_run<T, FACTOR>(sh, A, lda, inv_out, logdet_acc, true, info, 0);
}

template <typename T>
int run(const char* name, double tol) {
    const int n = 128, lda = 160;
    std::vector<double> A0(n * n), Lr(n * n, 0.0), Xr(n * n, 0.0);
    srand(7);
    // SPD with a decaying kernel + noise, not too well conditioned
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            A0[i * n + j] = exp(-0.02 * (i - j) * (i - j)) + (i == j ? 0.05 + 0.01 * (i % 7) : 0.0);
    // CPU reference
    std::vector<double> W = A0;
    double logdet = 0;
    for (int j = 0; j < n; ++j) {
        double d = W[j * n + j];
        for (int k = 0; k < j; ++k) d -= Lr[j * n + k] * Lr[j * n + k];
        Lr[j * n + j] = sqrt(d);
        logdet += log(d);
        for (int i = j + 1; i < n; ++i) {
            double s = W[i * n + j];
            for (int k = 0; k < j; ++k) s -= Lr[i * n + k] * Lr[j * n + k];
            Lr[i * n + j] = s / Lr[j * n + j];
        }
    }
    for (int c = 0; c < n; ++c)
        for (int i = c; i < n; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= Lr[i * n + k] * Xr[k * n + c];
            Xr[i * n + c] = s / Lr[i * n + i];
        }
    std::vector<T> hA(n * lda, (T)0), hInv(n * n), hL(n * lda);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) hA[i * lda + j] = (T)A0[i * n + j];
    T *dA, *dInv;
    double* dLd;
    int* dInfo;
    hipMalloc(&dA, sizeof(T) * n * lda);
    hipMalloc(&dInv, sizeof(T) * n * n);
    hipMalloc(&dLd, 8);
    hipMalloc(&dInfo, 4);
    int bad = 0;
    const char* names[8] = {"start", "loads issued", "p0 column in LDS", "p0 leaf", "p0 trsm", "p0 update", "all panels",
                            "tail stored"};
    for (int rep = 0; rep < 4; ++rep) {
        hipMemcpy(dA, hA.data(), sizeof(T) * n * lda, hipMemcpyHostToDevice);
        hipMemset(dLd, 0, 8);
        hipMemset(dInfo, 0, 4);
        hipMemset(dInv, 0xff, sizeof(T) * n * n);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((diag_kernel<T, true>), dim3(1), dim3(256), 0, 0, dA, (int64_t)lda, dInv, dLd, dInfo);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s rep %d: %.1f us (event)\n", name, rep, ms * 1e3);
        if (rep == 3) {
            unsigned long long st[64];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(algp::g_potrf_stamps), sizeof(st));
            printf("  whole block: %llu cycles\n", st[7] - st[0]);
            // wave 0 per 16-column panel: leaf | barrier + its share of the panel products | barrier (next diagonal block final)
            unsigned long long leaf = 0, crit = 0, bx = 0;
            for (int p = 0; p < 8; ++p) {
                leaf += st[8 + 3 * p + 1] - st[8 + 3 * p];
                crit += st[8 + 3 * p + 2] - st[8 + 3 * p + 1];
                if (p < 7) bx += st[8 + 3 * (p + 1)] - st[8 + 3 * p + 2];
            }
            unsigned long long sw[64];
            hipMemcpyFromSymbol(sw, HIP_SYMBOL(algp::g_potrf_stamps_w), sizeof(sw));
            printf("  per panel, cycles: wave 0 [leaf | B2 wait + its panel products | Bx wait]   bulk wave 1 [shadow work since Bx | B2 wait | section | Bx wait]\n");
            for (int p = 0; p < 8; ++p)
                printf("    p%d: w0 %5llu %5llu %5llu   w1 %5llu %5llu %5llu %5llu   (leaf ends %lld cycles after wave 1 reaches B2)\n", p,
                       st[8 + 3 * p + 1] - st[8 + 3 * p], st[8 + 3 * p + 2] - st[8 + 3 * p + 1], p < 7 ? st[8 + 3 * (p + 1)] - st[8 + 3 * p + 2] : 0ull,
                       p > 0 ? sw[4 * p] - sw[4 * (p - 1) + 3] : 0ull, sw[4 * p + 1] - sw[4 * p], sw[4 * p + 2] - sw[4 * p + 1], sw[4 * p + 3] - sw[4 * p + 2],
                       (long long)st[8 + 3 * p + 1] - (long long)sw[4 * p]);
            printf("  8 leaves %llu cycles; B2 + panel products %llu; wait for the next diagonal block (7x) %llu; start -> first leaf %llu; last leaf done -> end %llu\n",
                   leaf, crit, bx, st[8] - st[0], st[7] - st[8 + 3 * 7 + 1]);
        }
    }
    hipMemcpy(hL.data(), dA, sizeof(T) * n * lda, hipMemcpyDeviceToHost);
    hipMemcpy(hInv.data(), dInv, sizeof(T) * n * n, hipMemcpyDeviceToHost);
    double ld;
    int info;
    hipMemcpy(&ld, dLd, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&info, dInfo, 4, hipMemcpyDeviceToHost);
    double eL = 0, eX = 0, eU = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (j <= i) eL = fmax(eL, fabs((double)hL[i * lda + j] - Lr[i * n + j]));
            const double xr = (j <= i) ? Xr[i * n + j] : 0.0;
            eX = fmax(eX, fabs((double)hInv[i * n + j] - xr) / (1.0 + fabs(xr)));
            if (j > i) eU = fmax(eU, fabs((double)hInv[i * n + j]));
        }
    printf("%s: max|L-Lref| %.3e  max rel|X-Xref| %.3e  upper(inv) %.1e  logdet %.12f (ref %.12f) info %d\n", name, eL, eX,
           eU, ld, logdet, info);
    if (!(eL < tol) || !(eX < tol * 50) || eU != 0 || fabs(ld - logdet) > tol * 100 || info != 0) { printf("FAIL %s\n", name); bad = 1; }

    // inverse-only path on the factor just computed
    hipMemset(dInv, 0xff, sizeof(T) * n * n);
    hipLaunchKernelGGL((diag_kernel<T, false>), dim3(1), dim3(256), 0, 0, dA, (int64_t)lda, dInv, dLd, dInfo);
    hipMemcpy(hInv.data(), dInv, sizeof(T) * n * n, hipMemcpyDeviceToHost);
    eX = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double xr = (j <= i) ? Xr[i * n + j] : 0.0;
            eX = fmax(eX, fabs((double)hInv[i * n + j] - xr) / (1.0 + fabs(xr)));
        }
    printf("%s inverse-only: max rel|X-Xref| %.3e\n", name, eX);
    if (!(eX < tol * 50)) { printf("FAIL %s inverse-only\n", name); bad = 1; }

    // non-PD block: pivot 37 made negative -> info = 38
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) hA[i * lda + j] = (T)A0[i * n + j];
    hA[37 * lda + 37] = (T)-1.0;
    hipMemcpy(dA, hA.data(), sizeof(T) * n * lda, hipMemcpyHostToDevice);
    hipMemset(dInfo, 0, 4);
    hipLaunchKernelGGL((diag_kernel<T, true>), dim3(1), dim3(256), 0, 0, dA, (int64_t)lda, dInv, dLd, dInfo);
    hipMemcpy(&info, dInfo, 4, hipMemcpyDeviceToHost);
    printf("%s non-PD: info %d (want 38)\n", name, info);
    if (info != 38) { printf("FAIL %s info\n", name); bad = 1; }
    hipFree(dA); hipFree(dInv); hipFree(dLd); hipFree(dInfo);
    return bad;
}

int main() {
    int bad = run<double>("f64", 1e-12);
    bad |= run<float>("f32", 2e-4);
    printf(bad ? "DIAG TEST FAILED\n" : "DIAG TEST OK\n");
    return bad;
}
