// Diagnostic build of the 128 x 128 diagonal-block kernel with in-kernel cycle stamps
// (MI355X_MICROARCH.md: stamps only in a separate build; read shares, not the total).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DALGP_POTRF_STAMPS tools/potrf_stamp.hip -o gpurun_out/potrf_stamp
#include "../algp_amd/csrc/potrf.hip"
#include <stdio.h>
#include <vector>
namespace algp {
int fail(algp_ctx*, int code, const std::string&) { return code; }
void prof_begin(algp_ctx*, int, double, double) {}
void prof_end(algp_ctx*) {}
template <typename T>
int gemm_nt_launch(algp_ctx*, int, int64_t, int64_t, int64_t, T, const T*, int64_t, const T*, int64_t, T, const T*, int64_t, T*, int64_t, int) { return 0; }
template <typename T>
int gemm_nt_launch_batched(algp_ctx*, int, int64_t, int64_t, int64_t, T, const T*, int64_t, int64_t, const T*, int64_t, int64_t, T, const T*, int64_t, int64_t, T*, int64_t, int64_t, int, int) { return 0; }
template int gemm_nt_launch_batched<double>(algp_ctx*, int, int64_t, int64_t, int64_t, double, const double*, int64_t, int64_t, const double*, int64_t, int64_t, double, const double*, int64_t, int64_t, double*, int64_t, int64_t, int, int);
template int gemm_nt_launch_batched<float>(algp_ctx*, int, int64_t, int64_t, int64_t, float, const float*, int64_t, int64_t, const float*, int64_t, int64_t, float, const float*, int64_t, int64_t, float*, int64_t, int64_t, int, int);
int ensure(algp_ctx*, DevBuf& b, size_t bytes) { if (b.p) hipFree(b.p); hipMalloc(&b.p, bytes); b.cap = bytes; return 0; }
template int gemm_nt_launch<double>(algp_ctx*, int, int64_t, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t, double, const double*, int64_t, double*, int64_t, int);
template int gemm_nt_launch<float>(algp_ctx*, int, int64_t, int64_t, int64_t, float, const float*, int64_t, const float*, int64_t, float, const float*, int64_t, float*, int64_t, int);
}
int main() {
    const int n = 128;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 2.0 : 0.0) + exp(-0.05 * (i - j) * (i - j));
    double *dA, *dInv, *dLd; int* dInfo;
    hipMalloc(&dA, 8 * n * n); hipMalloc(&dInv, 8 * n * n); hipMalloc(&dLd, 8); hipMalloc(&dInfo, 4);
    hipMemset(dLd, 0, 8); hipMemset(dInfo, 0, 4);
    const char* names[18] = {"start", "loaded", "p0 begin", "p0 sweep done", "p0 writeback done", "p0 rank16 done", "all panels done", "logdet+L store done", "inv diag blocks", "", "row1", "row2", "row3", "row4", "row5", "row6", "row7", "inv stored"};
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), 8 * n * n, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((algp::potrf_diag_kernel<double, true>), dim3(1), dim3(256), 0, 0, dA, (int64_t)n, dInv, dLd, dInfo, (int64_t)0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[64];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(algp::g_potrf_stamps), sizeof(st));
        printf("rep %d: %.1f us total (event)\n", rep, ms * 1e3);
        if (rep == 2) {
            int prev = 0;
            for (int k = 1; k < 18; ++k) { if (k == 9) continue; printf("  %-22s +%8llu cycles (cum %8llu)\n", names[k], st[k] - st[prev], st[k] - st[0]); prev = k; }
        }
    }
    return 0;
}
