// gemm_loop_probe.hip -- the k-loop of gemm_nt_kernel_dma4 (gemm.hip) taken apart: the same workgroup (256 threads, 4 waves
// x 4 x 4 MFMA tiles, four 16 KB LDS stages, 64-byte row pieces by LDS-DMA, one barrier per k-tile, two workgroups per CU,
// 512 workgroups), every operand byte from a 2 x 128-row panel that stays in L2, and compile-time switches that remove one
// part at a time.  Which part keeps the matrix pipe at 85 % when a bare MFMA loop reaches 98 %?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/gemm_loop_probe.hip -o /tmp/gemm_loop_probe && /tmp/gemm_loop_probe
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

typedef double v2 __attribute__((ext_vector_type(2)));
typedef double v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// switches
constexpr int F_DMA = 1, F_BARRIER = 2, F_READS = 4, F_MFMA = 8, F_VMWAIT = 16;

template <int FLAGS, int MFMAS_PER_TILE>
__global__ __launch_bounds__(256, 2) void loop_kernel(const double* A, const double* B, int64_t ld, int nkt, double* out) {
    constexpr int NST = 4;
    __shared__ __attribute__((aligned(1024))) char smem[NST * 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((lane >> 4) & 2);
    const double* Ag = A + (int64_t)(32 * wave + srow) * ld + schunk * 2;
    const double* Bg = B + (int64_t)(32 * wave + srow) * ld + schunk * 2;
    auto stage = [&](int st, int kt) {
        char* As = smem + st * 16384 + wave * 2048;
        char* Bs = As + 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)(Ag + (int64_t)(16 * i) * ld + (int64_t)kt * 8), (lds_vp)(As + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(Bg + (int64_t)(16 * i) * ld + (int64_t)kt * 8), (lds_vp)(Bs + i * 1024), 16, 0, 0);
        }
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int coff = ((fg ^ ((fr >> 2) & 2)) << 4);
    const int aoff = (wr * 64 + fr) * 64 + coff;
    const int boff = (wc * 64 + fr) * 64 + coff;
    v4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4){0, 0, 0, 0};
    // something in LDS and in the fragment registers whatever the switches
    for (int t = 0; t < NST; ++t) stage(t, t);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    v2 a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const v2*>(smem + aoff + t * 1024);
        b[t] = *reinterpret_cast<const v2*>(smem + 8192 + boff + t * 1024);
    }
    if (FLAGS & F_DMA)
        for (int t = 0; t < NST - 1; ++t) stage(t, t);
    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (FLAGS & F_VMWAIT) {
            if (FLAGS & F_DMA) __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8): two younger tiles may fly
        }
        if (FLAGS & F_BARRIER) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (FLAGS & F_READS) {
            const char* As = smem + st * 16384;
            const char* Bs = As + 8192;
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const v2*>(As + aoff + t * 1024);
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const v2*>(Bs + boff + t * 1024);
        }
        if (FLAGS & F_DMA) stage(st == 0 ? NST - 1 : st - 1, (kt + NST - 1) & 63);
        if (FLAGS & F_MFMA) {
#pragma unroll
            for (int rep = 0; rep < MFMAS_PER_TILE / 32; ++rep)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][0][0] += a[i][0] + b[i][1];
        }
        st = (st + 1 == NST) ? 0 : st + 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456) out[blockIdx.x * 256 + tid] = s;
}

// the k-tile 128 bytes wide: TWO stages of 32 KB (the same 64 KB), 8 DMA instructions and 16 LDS reads per k-tile and wave,
// 64 MFMAs; a stage is refilled once every wave holds its fragments in registers (second barrier).  ORDER 0: reads, wait,
// barrier, refill, products; ORDER 1: as 0 with the fragments of the second half-tile read behind the first 32 products.
template <int ORDER>
__global__ __launch_bounds__(256, 2) void loop128_kernel(const double* A, const double* B, int64_t ld, int nkt, double* out) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r8 = lane >> 3, slot = lane & 7;
    const double* Ag[4];
    const double* Bg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * wave + 8 * i + r8;
        Ag[i] = A + (int64_t)row * ld + (slot ^ ((row >> 1) & 7)) * 2;
        Bg[i] = B + (int64_t)row * ld + (slot ^ ((row >> 1) & 7)) * 2;
    }
    auto stage = [&](int st, int kt) {
        char* As = smem + st * 32768 + wave * 4096;
        char* Bs = As + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)(Ag[i] + (int64_t)kt * 16), (lds_vp)(As + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(Bg[i] + (int64_t)kt * 16), (lds_vp)(Bs + i * 1024), 16, 0, 0);
        }
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = (fr >> 1) & 7;
    const int off0 = fr * 128 + ((fg ^ sw) << 4), off1 = fr * 128 + (((4 + fg) ^ sw) << 4);
    const int abase = wr * 64 * 128, bbase = 16384 + wc * 64 * 128;
    v4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4){0, 0, 0, 0};
    stage(0, 0);
    stage(1, 1);
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        __builtin_amdgcn_s_waitcnt(0x0F78);                        // vmcnt(8): this tile landed, the next may fly
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* base = smem + st * 32768;
        v2 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0[t] = *reinterpret_cast<const v2*>(base + abase + off0 + t * 2048);
            b0[t] = *reinterpret_cast<const v2*>(base + bbase + off0 + t * 2048);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a1[t] = *reinterpret_cast<const v2*>(base + abase + off1 + t * 2048);
            b1[t] = *reinterpret_cast<const v2*>(base + bbase + off1 + t * 2048);
        }
        if (ORDER == 0) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage(st, (kt + 2) & 31);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i][e], b0[j][e], acc[i][j], 0, 0, 0);
        if (ORDER == 1) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage(st, (kt + 2) & 31);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i][e], b1[j][e], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456) out[blockIdx.x * 256 + tid] = s;
}

template <int ORDER>
static void run128(const char* what, const double* A, const double* B, int64_t ld, int nkt, double* out, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((loop128_kernel<ORDER>), dim3(grid), dim3(256), 0, 0, A, B, ld, nkt, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double flop = (double)grid * 4 * nkt * 64 * 2048.0;
    printf("%-78s %8.3f ms  %6.1f TFLOP/s = %5.1f %% of 78.6\n", what, best, flop / best / 1e9, 100 * flop / best / 1e9 / 78.6);
    fflush(stdout);
}

template <int FLAGS, int MPT>
static void run(const char* what, const double* A, const double* B, int64_t ld, int nkt, double* out, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((loop_kernel<FLAGS, MPT>), dim3(grid), dim3(256), 0, 0, A, B, ld, nkt, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double flop = (FLAGS & F_MFMA) ? (double)grid * 4 * nkt * MPT * 2048.0 : 0.0;
    printf("%-78s %8.3f ms  %6.1f TFLOP/s = %5.1f %% of 78.6   (%.0f cycles per k-tile and CU at 2.4 GHz)\n", what, best, flop / best / 1e9,
           100 * flop / best / 1e9 / 78.6, best * 1e-3 * 2.4e9 / ((double)nkt * grid / 512.0));
    fflush(stdout);
}

int main() {
    const int64_t ld = 1024;
    double *A, *B, *out;
    CK(hipMalloc(&A, sizeof(double) * 128 * ld));
    CK(hipMalloc(&B, sizeof(double) * 128 * ld));
    CK(hipMalloc(&out, sizeof(double) * 4096 * 256));
    CK(hipMemset(A, 0, sizeof(double) * 128 * ld));
    CK(hipMemset(B, 0, sizeof(double) * 128 * ld));
    const int nkt = 4000;
    printf("512 workgroups x %d k-tiles of 32 MFMAs per wave; operands: two 128-row panels of 8 KB per k-tile, 64 k-tiles long (L2)\n", nkt);
    run<F_MFMA, 32>("products only (fragments stay in registers)", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_BARRIER, 32>("products + the barrier", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_READS, 32>("products + the 8 LDS reads in front of them, no barrier", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_READS | F_BARRIER, 32>("products + LDS reads + barrier", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_DMA | F_VMWAIT, 32>("products + the refill's 4 DMA instructions + vmcnt wait, no barrier, no reads", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_DMA | F_VMWAIT | F_BARRIER, 32>("products + DMA + wait + barrier", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_DMA | F_VMWAIT | F_BARRIER | F_READS, 32>("the whole loop", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_DMA | F_VMWAIT | F_BARRIER | F_READS, 32>("the whole loop, one workgroup per CU (256 workgroups)", A, B, ld, nkt, out, 256);
    run<F_DMA | F_VMWAIT | F_BARRIER | F_READS, 32>("the whole loop without the products", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_DMA | F_VMWAIT | F_BARRIER | F_READS, 64>("the whole loop with 64 MFMAs per k-tile (as if the k-tile were 128 bytes wide)", A, B, ld, nkt, out, 512);
    run<F_MFMA | F_BARRIER, 64>("products + barrier, 64 MFMAs per barrier", A, B, ld, nkt, out, 512);
    run128<0>("128-byte k-tiles, two 32 KB stages: reads, wait, barrier, refill, 64 products", A, B, ld, nkt / 2, out, 512);
    run128<1>("128-byte k-tiles, two stages: the wait + barrier + refill behind the first 32 products", A, B, ld, nkt / 2, out, 512);
    return 0;
}
