// rocblas_yardstick.cpp -- NOT part of the product (the library links no BLAS): what does the vendor's DGEMM / SGEMM reach on
// the shapes of the candidate solve, on the same card, in the same minute?  D = C - A B^T with A (m x k), B (n x k) row-major,
// i.e. column-major C^T (n x m) = C^T - B_cm^T ... written for rocBLAS as gemm(op_T, op_N) on the transposed problem.
// A yardstick for DESIGN section 7.1's ceiling statement: if the vendor kernel runs at the same 83-85 % on these operands,
// the gap to 78.6 TFLOP/s is the part's, not this kernel's.
//   hipcc -O2 tools/rocblas_yardstick.cpp -lrocblas -o build/tools/rocblas_yardstick
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        if ((x) != hipSuccess) { fprintf(stderr, "HIP error line %d\n", __LINE__); exit(1); } \
    } while (0)

template <typename T>
static void run(rocblas_handle h, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int reps) {
    T *A, *B, *C;
    CK(hipMalloc(&A, sizeof(T) * m * lda));
    CK(hipMalloc(&B, sizeof(T) * n * ldb));
    CK(hipMalloc(&C, sizeof(T) * m * ldc));
    std::vector<T> hA((size_t)m * lda), hB((size_t)n * ldb);
    unsigned s = 99;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (T)(((int)(s >> 8) % 2001 - 1000) * 1e-3); }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (T)(((int)(s >> 8) % 2001 - 1000) * 1e-3); }
    CK(hipMemcpy(A, hA.data(), sizeof(T) * m * lda, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), sizeof(T) * n * ldb, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, sizeof(T) * m * ldc));
    const T alpha = (T)-1, beta = (T)1;
    // row-major D (m x n) = C - A B^T  <=>  column-major D^T (n x m) = C^T - B_cm^T-view: B row-major (n x k) is column-major (k x n)
    // with leading dimension ldb -> op_T gives n x k; A row-major (m x k) is column-major (k x m) -> op_N
    auto call = [&]() {
        if constexpr (sizeof(T) == 8)
            return rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)n, (rocblas_int)m, (rocblas_int)k,
                                 (const double*)&alpha, (const double*)B, (rocblas_int)ldb, (const double*)A, (rocblas_int)lda,
                                 (const double*)&beta, (double*)C, (rocblas_int)ldc);
        else
            return rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)n, (rocblas_int)m, (rocblas_int)k,
                                 (const float*)&alpha, (const float*)B, (rocblas_int)ldb, (const float*)A, (rocblas_int)lda,
                                 (const float*)&beta, (float*)C, (rocblas_int)ldc);
    };
    for (int w = 0; w < 2; ++w)
        if (call() != rocblas_status_success) { printf("rocblas gemm failed\n"); return; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        call();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double tf = 2.0 * m * n * k / best / 1e9, peak = sizeof(T) == 8 ? 78.6 : 157.3;
    printf("rocBLAS %s  m %6lld n %5lld k %5lld (ld %lld / %lld / %lld): %8.3f ms  %6.1f TFLOP/s = %5.1f %% of %.1f\n", sizeof(T) == 8 ? "dgemm" : "sgemm",
           (long long)m, (long long)n, (long long)k, (long long)lda, (long long)ldb, (long long)ldc, best, tf, 100 * tf / peak, peak);
    fflush(stdout);
    CK(hipFree(A));
    CK(hipFree(B));
    CK(hipFree(C));
}

int main() {
    rocblas_handle h;
    if (rocblas_create_handle(&h) != rocblas_status_success) { printf("no rocblas handle\n"); return 1; }
    // the solve's update products: rows of one chunk / of all chunks x one 512-column block x the K of blocks 4, 10, 19
    const int64_t ldx = 10240, ldl = 10112;
    run<double>(h, 33408, 512, 2048, ldx, ldl, 512, 5);
    run<double>(h, 33408, 512, 5120, ldx, ldl, 512, 5);
    run<double>(h, 33408, 512, 9728, ldx, ldl, 512, 5);
    run<double>(h, 100096, 512, 5120, ldx, ldl, 512, 3);
    run<double>(h, 100096, 512, 9728, ldx, ldl, 512, 3);
    run<double>(h, 4096, 4096, 4096, 4096, 4096, 4096, 5);
    run<double>(h, 8192, 8192, 8192, 8192, 8192, 8192, 3);
    run<float>(h, 33408, 512, 5120, ldx, ldl, 512, 5);
    run<float>(h, 8192, 8192, 8192, 8192, 8192, 8192, 3);
    rocblas_destroy_handle(h);
    return 0;
}
