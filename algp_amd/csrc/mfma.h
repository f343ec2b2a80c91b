// MFMA traits shared by the GEMM and the diagonal-block kernel.
//   fp64: v_mfma_f64_16x16x4_f64   C/D: row = (lane>>4) + 4*reg, col = lane&15
//   fp32: v_mfma_f32_16x16x4_f32   C/D: row = (lane>>4)*4 + reg, col = lane&15
// A operand: one scalar per lane, A[i = lane&15][k = lane>>4]; B: B[k = lane>>4][j = lane&15].
// Accumulators: keep kernels at <= 256 registers so that they stay in arch VGPRs.  Not because AccVGPR accumulators are
// slow -- rocBLAS's fp64 kernel holds 256 of them at 97 % of peak -- but because hipcc does not keep them there in place:
// round 1's microbenchmark ("AGPR accumulators run at half rate", profiles/r01_mfma_rate_microbench.txt) measured hipcc's
// own 128 v_accvgpr_write + 128 v_accvgpr_read per 16 MFMAs, and round 6's 256-accumulator kernel got 224-500 copies per
// k-tile (EXPERIMENTS.md).  Where hipcc does put a small kernel's accumulators into AccVGPRs without copies in the loop
// (tail_finish_kernel: 64 AGPRs, a 65-us epilogue kernel), that is fine.
#pragma once
#include <hip/hip_runtime.h>

namespace algp {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename T>
struct MF;
template <>
struct MF<double> {
    using acc_t = v4d;
    using chunk_t = v2d;
    static constexpr int EPC = 2;
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct MF<float> {
    using acc_t = v4f;
    using chunk_t = v4f;
    static constexpr int EPC = 4;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// Write-through store (global_store ... sc1): the bytes leave for memory at once instead of staying dirty in the XCD's
// L2, so a tile handed to workgroups on other XCDs needs no L2 write-back (buffer_wbl2, which drains EVERY dirty line
// of the XCD: several microseconds with 64 workgroups writing tiles) before its flag -- only the storing waves' own
// s_waitcnt vmcnt(0) (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores, drained, then the flag).
#ifndef ALGP_DAG_WT
#define ALGP_DAG_WT 1
#endif
template <typename T>
__device__ __forceinline__ void st_wt(T* p, T v) {
#if ALGP_DAG_WT
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}

}  // namespace algp
