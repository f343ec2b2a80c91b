// row_piece_probe.hip -- how fast can HBM deliver a row-major matrix when a workgroup owns 128 (or 64) ROWS and walks them
// in pieces of P bytes per row (the access pattern of tail.hip: V^T is read once, candidate rows x k-tiles), as a function
// of P and of the bytes a workgroup keeps in flight?  No arithmetic beyond a sum that keeps the loads alive; occupancy is
// pinned by a dynamic-LDS request, as the real kernel's stages pin it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/row_piece_probe.hip -o /tmp/row_piece_probe && /tmp/row_piece_probe [rows] [cols]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));

// P: bytes per row piece; R: rows per workgroup; U: k-steps in flight (loads per thread in flight = U * R * P / 4096)
template <int P, int R, int U>
__global__ __launch_bounds__(256) void probe(const double* X, int64_t ldx, int64_t ncols, double* out) {
    extern __shared__ char smem[];
    constexpr int LPR = P / 16;                // lanes per row piece
    constexpr int RPP = 256 / LPR;             // rows per pass of the workgroup
    constexpr int NP = R / RPP;                // passes per k-step
    constexpr int EP = P / 8;                  // elements per piece
    const int t = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * R;
    const double* base = X + (m0 + t / LPR) * ldx + (t % LPR) * 2;
    v2 acc = {0.0, 0.0};
    const int64_t nk = ncols / EP;
    for (int64_t k = 0; k + U <= nk; k += U) {
        v2 x[U][NP];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int p = 0; p < NP; ++p) x[u][p] = *reinterpret_cast<const v2*>(base + (int64_t)p * RPP * ldx + (k + u) * EP);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc += x[u][p];
    }
    if (acc[0] + acc[1] == 123.456) out[blockIdx.x] = acc[0] + (double)smem[0];
}

template <int P, int R, int U>
static void run(const double* X, int64_t rows, int64_t ldx, int64_t ncols, double* out, int lds_kb) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)probe<P, R, U>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    const dim3 grid((unsigned)(rows / R)), blk(256);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<P, R, U>), grid, blk, lds_kb * 1024, 0, X, ldx, ncols, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)rows * (double)(ncols / (P / 8) / U * U) * P;
    printf("piece %4d B  rows/WG %3d  in flight %3d KB/WG  LDS %3d KB (%d WG/CU)  %8.3f ms  %7.1f GB/s\n", P, R, U * R * P / 1024, lds_kb,
           160 / lds_kb, best, bytes / best / 1e6);
    fflush(stdout);
}


// The GEMM's own request pattern (gemm.hip: stage()): P-byte row pieces by LDS-DMA, 16 bytes per lane, the P / 16 lanes of a
// piece in the order `lane ^ x` -- XORM 0: x = 0; 1: x = (row >> 2) & 3 (rounds 1-4); 2: x = (row >> 2) & 2 (round 5) -- U
// k-steps in flight per wave.  The bytes are only counted, never read back: for the FETCH_SIZE calibration
// (tools/fetch_calibrate.sh), where every launch moves a known number of bytes.
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;
template <int P, int XORM, int U>
__global__ __launch_bounds__(256) void probe_dma(const double* X, int64_t ldx, int64_t ncols, double* out) {
    extern __shared__ char smem[];
    constexpr int LPR = P / 16, RPW = 64 / LPR, NP = 128 / (4 * RPW), EP = P / 8;   // rows per wave and instruction; instructions per k-step
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int row = lane / LPR;
    const int x = XORM == 1 ? ((row >> 2) & 3) : XORM == 2 ? ((row >> 2) & 2) : 0;
    const int chunk = ((lane % LPR) ^ x) % LPR;
    const double* base = X + (m0 + wave * (128 / 4) + row) * ldx + chunk * 2;
    const int64_t nk = ncols / EP;
    for (int64_t k = 0; k < nk; ++k) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            __builtin_amdgcn_global_load_lds((glb_vp)(base + (int64_t)p * RPW * ldx + k * EP),
                                             (lds_vp)(smem + ((((k % U) * NP + p) & 15) * 4096) + wave * 1024), 16, 0, 0);
        if (NP * (U - 1) >= 8) __builtin_amdgcn_s_waitcnt(0x0F78);
        else if (NP * (U - 1) >= 4) __builtin_amdgcn_s_waitcnt(0x0F74);
        else __builtin_amdgcn_s_waitcnt(0x0F72);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (ncols < 0) out[blockIdx.x] = (double)smem[t];
}
template <int P, int XORM, int U>
static void run_dma(const double* X, int64_t rows, int64_t ldx, int64_t ncols, double* out, int lds_kb, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)probe_dma<P, XORM, U>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    float best = 1e30f;
    for (int it = 0; it < reps; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe_dma<P, XORM, U>), dim3((unsigned)(rows / 128)), dim3(256), lds_kb * 1024, 0, X, ldx, ncols, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)rows * (double)(ncols / (P / 8)) * P;
    printf("LDS-DMA piece %4d B  lane order %d  %d k-steps in flight  LDS %3d KB  %8.3f ms  %7.1f GB/s  bytes %.0f\n", P, XORM, U, lds_kb, best,
           bytes / best / 1e6, bytes);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 100096, ncols = argc > 2 ? atoll(argv[2]) : 50048;
    const int64_t ldx = ncols + 128;
    double *X, *out;
    CK(hipMalloc(&X, sizeof(double) * rows * ldx));
    CK(hipMalloc(&out, sizeof(double) * rows));
    CK(hipMemset(X, 0, sizeof(double) * rows * ldx));
    CK(hipDeviceSynchronize());
    printf("matrix %lld x %lld fp64 (%.1f GB), leading dimension %lld\n", (long long)rows, (long long)ncols, rows * ncols * 8e-9, (long long)ldx);
    if (argc > 3 && atoi(argv[3]) == 1) {
        // calibration set: ONE launch per shape (a counter pass attributes FETCH_SIZE per dispatch; the true byte count is
        // rows x ncols x 8 for every one of them)
        run_dma<64, 0, 3>(X, rows, ldx, ncols, out, 72, 1);
        run_dma<64, 1, 3>(X, rows, ldx, ncols, out, 72, 1);
        run_dma<64, 2, 3>(X, rows, ldx, ncols, out, 72, 1);
        run_dma<128, 0, 2>(X, rows, ldx, ncols, out, 72, 1);
        run_dma<256, 0, 2>(X, rows, ldx, ncols, out, 72, 1);
        run_dma<1024, 0, 1>(X, rows, ldx, ncols, out, 72, 1);
        run<64, 128, 6>(X, rows, ldx, ncols, out, 72);
        run<128, 128, 3>(X, rows, ldx, ncols, out, 72);
        run<1024, 64, 1>(X, rows, ldx, ncols, out, 72);
        CK(hipFree(X));
        CK(hipFree(out));
        return 0;
    }
    // two workgroups per CU (72 KB), as tail.hip's JT = 2
    run<64, 128, 6>(X, rows, ldx, ncols, out, 72);
    run<128, 128, 2>(X, rows, ldx, ncols, out, 72);
    run<128, 128, 3>(X, rows, ldx, ncols, out, 72);
    run<256, 128, 1>(X, rows, ldx, ncols, out, 72);
    run<256, 128, 2>(X, rows, ldx, ncols, out, 72);
    run<512, 128, 1>(X, rows, ldx, ncols, out, 72);
    run<256, 64, 2>(X, rows, ldx, ncols, out, 72);
    run<256, 64, 3>(X, rows, ldx, ncols, out, 72);
    run<512, 64, 1>(X, rows, ldx, ncols, out, 72);
    run<512, 64, 2>(X, rows, ldx, ncols, out, 72);
    run<1024, 64, 1>(X, rows, ldx, ncols, out, 72);
    // three / four per CU
    run<128, 128, 3>(X, rows, ldx, ncols, out, 48);
    run<256, 64, 2>(X, rows, ldx, ncols, out, 48);
    run<256, 64, 2>(X, rows, ldx, ncols, out, 36);
    run<512, 64, 1>(X, rows, ldx, ncols, out, 36);
    // one per CU
    run<256, 128, 3>(X, rows, ldx, ncols, out, 144);
    run<512, 128, 2>(X, rows, ldx, ncols, out, 144);
    CK(hipFree(X));
    CK(hipFree(out));
    return 0;
}
