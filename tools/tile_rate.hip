// How fast the Cholesky's 128 x 128 tile product (tile_mainloop of algp_amd/csrc/chol_dag.hip) runs with 512 resident
// workgroups when its operand panels (a) are all different -- every byte from beyond the L2, as in the task list today --,
// (b) are the same for everybody (all L2 hits), (c) are shared by the workgroups of an XCD in an R x (64 / R) arrangement.
// Answers whether the update tasks are held back by the fabric.  Prints microseconds per K = 128 step and TFLOP/s.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/tile_rate.hip -o build/tile_rate
#include "../algp_amd/csrc/chol_dag.hip"
#include <stdio.h>
#include <vector>
namespace algp {
int fail(algp_ctx*, int code, const std::string& m) { fprintf(stderr, "fail: %s\n", m.c_str()); return code; }
void prof_begin(algp_ctx*, int, double, double) {}
void prof_end(algp_ctx*) {}
int ensure(algp_ctx*, DevBuf& b, size_t bytes) { if (b.p && b.cap >= bytes) return 0; if (b.p) hipFree(b.p); hipMalloc(&b.p, bytes); b.cap = bytes; return 0; }
}
using namespace algp;

// mode 0: distinct panels; 1: one pair of panels; 2: per XCD, R rows x 64/R columns; 3: as 2 with each workgroup starting
// at its own phase of the k range (the steady state of a task list: sharers are spread over a task's duration)
template <typename T>
__global__ __launch_bounds__(256, 2) void rate_kernel(const T* L, int64_t ld, int mode, int R, int ktiles, int reps, T* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * 16384];
    using F = MF<T>;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    const int w = blockIdx.x >> 3;                                 // workgroups go round-robin over the XCDs
    int ra, rb;
    if (mode == 0) { ra = blockIdx.x; rb = 512 + blockIdx.x; }
    else if (mode == 1) { ra = 0; rb = 1; }
    else { ra = xcc * 64 + w % R; rb = 512 + xcc * 64 + w / R; }
    typename F::acc_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;
    const int nk = ktiles * (128 / (4 * F::EPC));
    for (int rep = 0; rep < reps; ++rep) {
        int phase = 0;
        if (mode == 3) phase = ((w * 7919) % ktiles);
        // two calls so that a phase-shifted workgroup still reads the whole range once per rep
        const T* A0 = L + (int64_t)ra * 128 * ld;
        const T* B0 = L + (int64_t)rb * 128 * ld;
        if (phase) {
            tile_mainloop<T>(smem, A0 + phase * 128, ld, B0 + phase * 128, ld, (ktiles - phase) * (128 / (4 * F::EPC)), acc);
            __syncthreads();
            tile_mainloop<T>(smem, A0, ld, B0, ld, phase * (128 / (4 * F::EPC)), acc);
        } else {
            tile_mainloop<T>(smem, A0, ld, B0, ld, nk, acc);
        }
        __syncthreads();
    }
    T s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == (T)12345.678) sink[threadIdx.x] = s;
}

template <typename T>
void run(const char* name) {
    const int64_t rows = 1024 * 128, ld = 2048 + 64;              // 1024 row panels of K = 2048 (+ padding: not a power of two)
    T* L;
    hipMalloc(&L, rows * ld * sizeof(T));
    hipMemset(L, 0, rows * ld * sizeof(T));
    T* sink;
    hipMalloc(&sink, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int ktiles = 16, reps = 8;
    struct Cfg { int mode, R; const char* what; } cfgs[] = {
        {0, 0, "every panel different"}, {1, 0, "one pair of panels for all"}, {2, 8, "per XCD 8 x 8, in step"},
        {3, 8, "per XCD 8 x 8, phases spread"}, {3, 4, "per XCD 4 x 16, phases spread"}, {3, 16, "per XCD 16 x 4, phases spread"},
        {2, 2, "per XCD 2 x 32, in step"}, {3, 2, "per XCD 2 x 32, phases spread"}};
    for (const Cfg& c : cfgs) {
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            rate_kernel<T><<<512, 256>>>(L, ld, c.mode, c.R, ktiles, reps, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double steps = (double)ktiles * reps;                // K = 128 steps per workgroup
        const double us = best * 1e3 / steps;
        const double tf = 512.0 * steps * 2.0 * 128 * 128 * 128 / (best * 1e-3) / 1e12;
        printf("%s  %-32s  %6.2f us per K=128 step  %6.1f TFLOP/s  (%.0f GB/s of operand reads)\n", name, c.what, us, tf,
               512.0 * steps * 2 * 128 * 128 * sizeof(T) / (best * 1e-3) / 1e9);
    }
    hipFree(L);
    hipFree(sink);
}
int main() {
    run<double>("f64");
    run<float>("f32");
    return 0;
}
