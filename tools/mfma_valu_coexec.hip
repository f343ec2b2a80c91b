// mfma_valu_coexec.hip -- does an independent v_fma_f64 stream run BESIDE the fp64 MFMA stream on gfx950?
// On MI355X the fp64 vector peak equals the fp64 matrix peak (78.6 TFLOP/s): if the two pipes execute together a VALU
// side-tile could add to the GEMM's rate; if the matrix instruction occupies the same double-precision units, it cannot.
// Three shapes, all with two waves per SIMD, operands in registers only (no memory traffic inside the timed loop):
//   same-wave : every wave issues 16 MFMAs (64 x 64 accumulator tile) and R v_fma_f64 per MFMA on accumulators of its own,
//               interleaved by sched_group_barrier (R = 0 is the MFMA stream alone, MFMAS = 0 the VALU stream alone);
//   split-wave: 512-thread workgroups, waves 0-3 only MFMAs, waves 4-7 (their SIMD partners) only v_fma_f64;
// reported: time, matrix flop rate, vector flop rate, the sum, and the sum over the MFMA stream alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
// Under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES the same binary gives the counters.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef double v4 __attribute__((ext_vector_type(4)));

// R vector FMAs per MFMA inside one wave; MF = 1: the 16 MFMAs are there, 0: only the vector stream (16 * R FMAs per iteration)
template <int R, int MF>
__global__ __launch_bounds__(256, 2) void same_wave_kernel(const double* in, int iters, double* out) {
    v4 acc[16];
    double f[16];                                                  // the vector stream's accumulators (independent chains)
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] = (v4){0, 0, 0, 0};
        f[i] = in[(t + i) & 1023];
    }
    const double a0 = in[(t * 3) & 1023], b0 = in[(t * 5 + 1) & 1023], x = in[(t * 7 + 2) & 1023], y = in[(t * 11 + 3) & 1023];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MF) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[i], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < R; ++r) f[(i * R + r) & 15] = __builtin_fma(f[(i * R + r) & 15], x, y);
            if (MF) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
            if (R) __builtin_amdgcn_sched_group_barrier(0x002, R, 0);    // ... then R VALU instructions
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
    if (s == 123.456) out[blockIdx.x * 256 + t] = s;
}

// waves 0-3: MFMAs only; waves 4-7: RV vector FMAs per iteration (where the partner issues 16 MFMAs)
template <int RV>
__global__ __launch_bounds__(512, 1) void split_wave_kernel(const double* in, int iters, double* out) {
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const double a0 = in[(t * 3) & 1023], b0 = in[(t * 5 + 1) & 1023], x = in[(t * 7 + 2) & 1023], y = in[(t * 11 + 3) & 1023];
    double s = 0;
    if (wave < 4) {
        v4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (v4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double f[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = in[(t + i) & 1023];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < RV; ++r) f[r & 15] = __builtin_fma(f[r & 15], x, y);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += f[i];
    }
    if (s == 123.456) out[blockIdx.x * 512 + t] = s;
}

static float time_launch(void (*launch)(void*), void* arg) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, 0));
        launch(arg);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

struct Args {
    const double* in;
    double* out;
    int iters;
};
static double g_mfma_alone = 0.0;

template <int R, int MF>
static void run_same(const Args& a) {
    auto l = [](void* p) {
        const Args* q = (const Args*)p;
        hipLaunchKernelGGL((same_wave_kernel<R, MF>), dim3(512), dim3(256), 0, 0, q->in, q->iters, q->out);
    };
    const float ms = time_launch(l, (void*)&a);
    const double waves = 512.0 * 4.0;
    const double mflop = MF ? waves * a.iters * 16.0 * 2048.0 : 0.0;            // 16 x 16 x 4 x 2 per MFMA
    const double vflop = waves * a.iters * 16.0 * R * 64.0 * 2.0;               // 64 lanes x 2 per v_fma_f64
    const double mt = mflop / ms / 1e9, vt = vflop / ms / 1e9;
    if (MF && R == 0) g_mfma_alone = mt;
    printf("same wave   %2d v_fma_f64 per MFMA%s: %8.3f ms  matrix %6.2f + vector %6.2f = %6.2f TFLOP/s", R, MF ? "" : " (no MFMA)", ms, mt, vt, mt + vt);
    if (g_mfma_alone > 0) printf("  = %.3f x the MFMA stream alone", (mt + vt) / g_mfma_alone);
    printf("\n");
    fflush(stdout);
}

template <int RV>
static void run_split(const Args& a) {
    auto l = [](void* p) {
        const Args* q = (const Args*)p;
        hipLaunchKernelGGL((split_wave_kernel<RV>), dim3(256), dim3(512), 0, 0, q->in, q->iters, q->out);
    };
    const float ms = time_launch(l, (void*)&a);
    const double mflop = 256.0 * 4.0 * a.iters * 16.0 * 2048.0;
    const double vflop = 256.0 * 4.0 * a.iters * (double)RV * 64.0 * 2.0;
    const double mt = mflop / ms / 1e9, vt = vflop / ms / 1e9;
    printf("split waves: one MFMA wave (16 per iteration) + one VALU wave (%3d v_fma_f64 per iteration) per SIMD: %8.3f ms  matrix %6.2f + vector %6.2f = %6.2f TFLOP/s\n",
           RV, ms, mt, vt, mt + vt);
    fflush(stdout);
}

int main() {
    double *in, *out;
    CK(hipMalloc(&in, 8 * 1024));
    CK(hipMalloc(&out, 8 * 512 * 512));
    double h[1024];
    unsigned s = 4242;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = ((int)(s >> 8) % 2001 - 1000) * 1e-4;
    }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    Args a{in, out, 20000};
    printf("fp64 on gfx950: matrix peak 78.6 TFLOP/s (v_mfma_f64_16x16x4: 64 cycles per SIMD), vector peak 78.6 (v_fma_f64: 4 cycles)\n");
    run_same<0, 1>(a);
    run_same<1, 0>(a);
    run_same<1, 1>(a);
    run_same<2, 1>(a);
    run_same<4, 1>(a);
    run_same<8, 1>(a);
    run_same<16, 1>(a);
    a.iters = 10000;
    run_split<16>(a);                                              // 16 FMAs (64 cycles) beside 16 MFMAs (1024 cycles)
    run_split<64>(a);
    run_split<128>(a);
    run_split<256>(a);                                             // equal issue time on both sides
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
