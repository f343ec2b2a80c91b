// Microbenchmark: sustained fp64/fp32 MFMA and fp64 VALU-FMA rates on MI355X with the in-kernel
// clock (s_memtime / s_memrealtime, MI355X_MICROARCH.md "DVFS give-back" item 6).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename T, typename ACC, int WPS>
__global__ __launch_bounds__(256, WPS) void rate_kernel(T* out, const T* in, int iters, unsigned long long* clk) {
    ACC acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = (T)0;
    T a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (sizeof(T) == 8) acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
                else acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
            }
    }
    T s = 0;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void fma64_kernel(double* out, const double* in, int iters, unsigned long long* clk) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = in[(threadIdx.x + i) & 4095];
    const double a = in[(threadIdx.x * 3) & 4095], b = in[(threadIdx.x * 5 + 1) & 4095];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static double clock_ghz(unsigned long long* dclk, int grid) {
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), dclk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid; ++i) s += (double)h[2 * i] / (double)h[2 * i + 1];
    return s / grid * 0.1;   // s_memrealtime ticks at 100 MHz
}

template <typename T, typename ACC, int WPS>
void run(const char* name, int blocks_per_cu, int fill, int iters) {
    int grid = 256 * blocks_per_cu;
    T *out, *in; hipMalloc(&out, sizeof(T) * grid * 256); hipMalloc(&in, sizeof(T) * 4096);
    unsigned long long* clk; hipMalloc(&clk, 16 * grid);
    std::vector<T> h(4096);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = fill == 0 ? (T)0 : (T)(((int)(s >> 8) % 2001 - 1000) * 1e-3); }
    hipMemcpy(in, h.data(), sizeof(T) * 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate_kernel<T, ACC, WPS>), dim3(grid), dim3(256), 0, 0, out, in, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 * iters * 16.0 * (16 * 16 * 4 * 2);
        printf("%-16s %-6s blocks/CU=%d iters=%d rep=%d: %8.3f ms %7.2f TFLOP/s  in-kernel clock %.2f GHz\n", name,
               fill ? "random" : "zeros", blocks_per_cu, iters, rep, ms, flops / ms * 1e-9, clock_ghz(clk, grid));
    }
    hipFree(out); hipFree(in); hipFree(clk);
}

void run_fma(int blocks_per_cu, int iters) {
    int grid = 256 * blocks_per_cu;
    double *out, *in; hipMalloc(&out, 8 * grid * 256); hipMalloc(&in, 8 * 4096);
    unsigned long long* clk; hipMalloc(&clk, 16 * grid);
    std::vector<double> h(4096);
    unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3; }
    hipMemcpy(in, h.data(), 8 * 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma64_kernel, dim3(grid), dim3(256), 0, 0, out, in, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 256 * iters * 16.0 * 2;
        printf("v_fma_f64      random blocks/CU=%d iters=%d rep=%d: %8.3f ms %7.2f TFLOP/s  in-kernel clock %.2f GHz\n",
               blocks_per_cu, iters, rep, ms, flops / ms * 1e-9, clock_ghz(clk, grid));
    }
    hipFree(out); hipFree(in); hipFree(clk);
}

int main() {
    // WPS = launch-bound waves/SIMD: 1 -> 512 registers (hipcc puts the accumulators in AGPRs),
    //                                2 -> 256 registers (accumulators stay in arch VGPRs)
    run<double, v4d, 1>("mfma f64 agpr", 1, 1, 4000);
    run<double, v4d, 1>("mfma f64 agpr", 1, 0, 4000);
    run<double, v4d, 1>("mfma f64 agpr", 1, 1, 40000);
    run<double, v4d, 2>("mfma f64 vgpr", 1, 1, 4000);
    run<double, v4d, 2>("mfma f64 vgpr", 2, 1, 4000);
    run<double, v4d, 2>("mfma f64 vgpr", 2, 0, 4000);
    run<double, v4d, 2>("mfma f64 vgpr", 2, 1, 40000);
    run<float, v4f, 1>("mfma f32 agpr", 1, 1, 8000);
    run<float, v4f, 2>("mfma f32 vgpr", 2, 1, 8000);
    run_fma(2, 20000);
    run_fma(4, 20000);
    return 0;
}
