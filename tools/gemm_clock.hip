// Diagnostic: the clock the chip holds inside the fp64 GEMM main loop (DVFS give-back check,
// MI355X_MICROARCH.md item 6).  Back-to-back launches on random operands for ~2 s first.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DALGP_GEMM_CLOCK tools/gemm_clock.hip -o gpurun_out/gemm_clock
#include "../algp_amd/csrc/gemm.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
namespace algp {
int fail(algp_ctx*, int code, const std::string&) { return code; }
int ensure(algp_ctx*, DevBuf& b, size_t bytes) { if (b.p) hipFree(b.p); hipMalloc(&b.p, bytes); b.cap = bytes; return 0; }
void prof_begin(algp_ctx*, int, double, double) {}
void prof_end(algp_ctx*) {}
}
int main() {
    algp_ctx c; hipStreamCreate(&c.stream); c.cur = c.stream;
    const int64_t n = 4096;
    double *A, *B, *C;
    hipMalloc(&A, 8 * n * n); hipMalloc(&B, 8 * n * n); hipMalloc(&C, 8 * n * n);
    hipLaunchKernelGGL(algp::fill_random_kernel<double>, dim3(n * n / 256), dim3(256), 0, c.stream, A, n * n, 1u);
    hipLaunchKernelGGL(algp::fill_random_kernel<double>, dim3(n * n / 256), dim3(256), 0, c.stream, B, n * n, 2u);
    hipLaunchKernelGGL(algp::fill_random_kernel<double>, dim3(n * n / 256), dim3(256), 0, c.stream, C, n * n, 3u);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int phase = 0; phase < 2; ++phase) {
        const int reps = phase == 0 ? 1000 : 20;     // ~2 s warm, then the measured burst
        hipEventRecord(e0, c.stream);
        for (int r = 0; r < reps; ++r)
            algp::gemm_nt_launch<double>(&c, 7, n, n, n, -1.0, A, n, B, n, 0.0, nullptr, 0, C, n, 0);
        hipEventRecord(e1, c.stream); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("phase %d: %d launches %.1f ms  %.2f TFLOP/s\n", phase, reps, ms, 2.0 * n * n * n * reps / ms * 1e-9);
    }
    std::vector<unsigned long long> h(2 * 1024);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(algp::g_gemm_clk), sizeof(unsigned long long) * 2 * 1024);
    std::vector<double> ghz;
    for (int i = 0; i < 1024; ++i) if (h[2 * i + 1]) ghz.push_back(0.1 * (double)h[2 * i] / (double)h[2 * i + 1]);
    std::sort(ghz.begin(), ghz.end());
    printf("in-kernel clock over %zu blocks: median %.3f GHz (min %.3f max %.3f); loop cycles median %llu\n", ghz.size(),
           ghz[ghz.size() / 2], ghz.front(), ghz.back(), h[2 * 512]);
    // MFMA-pipe utilisation at that clock: 64 MFMAs/wave/k-tile * 64 cycles * 2 waves/SIMD (2 blocks per CU)
    return 0;
}
