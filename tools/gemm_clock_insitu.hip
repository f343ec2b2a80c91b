// Diagnostic: per-workgroup main-loop cycles of the fp64 GEMM in the SHAPE of the candidate solve's outer update
// (A = 100 096 x K rows of V^T at stride 10 240, B = 512 rows of L, C = a 512-column slice updated in place),
// against the L2-resident square case of tools/gemm_clock.hip.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DALGP_GEMM_CLOCK tools/gemm_clock_insitu.hip -o gpurun_out/gemm_clock_insitu
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
    const int64_t M = 100096, ld = 10240, K = 5120, NL = 10240;
    double *V, *L;
    hipMalloc(&V, 8 * M * ld); hipMalloc(&L, 8 * NL * ld);
    hipLaunchKernelGGL(algp::fill_random_kernel<double>, dim3((unsigned)(M * ld / 256)), dim3(256), 0, c.stream, V, M * ld, 1u);
    hipLaunchKernelGGL(algp::fill_random_kernel<double>, dim3((unsigned)(NL * ld / 256)), dim3(256), 0, c.stream, L, NL * ld, 2u);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rows_case = 0; rows_case < 2; ++rows_case) {
        const int64_t m = rows_case == 0 ? M : 33408;             // all rows | one of three row chunks
        for (int phase = 0; phase < 2; ++phase) {
            const int reps = phase == 0 ? 40 : 10;
            hipEventRecord(e0, c.stream);
            for (int r = 0; r < reps; ++r)
                algp::gemm_nt_launch<double>(&c, 7, m, 512, K, -1.0, V, ld, L + K * ld, ld, 1.0, V + K, ld, V + K, ld, 0);
            hipEventRecord(e1, c.stream); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (phase == 1) printf("m=%lld: %d launches %.2f ms each  %.2f TFLOP/s\n", (long long)m, reps, ms / reps, 2.0 * m * 512 * K * reps / ms * 1e-9);
        }
        const int ntile = (int)(m / 128 * 4);
        std::vector<unsigned long long> h(2 * 8192);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(algp::g_gemm_clk), sizeof(unsigned long long) * 2 * 8192);
        std::vector<double> cyc;
        for (int i = 0; i < ntile && i < 8192; ++i) if (h[2 * i + 1]) cyc.push_back((double)h[2 * i] / (K / 16));
        std::sort(cyc.begin(), cyc.end());
        printf("  loop cycles per k-tile over %zu workgroups: p10 %.0f median %.0f p90 %.0f max %.0f (8192 = MFMA pipe full, 2 workgroups per CU)\n",
               cyc.size(), cyc[cyc.size() / 10], cyc[cyc.size() / 2], cyc[cyc.size() * 9 / 10], cyc.back());
    }
    return 0;
}
