// Dependent-chain latencies on a lone gfx950 wave (cycles of s_memtime per link): what the serial 16 x 16 leaf of the
// diagonal-block kernel (algp_amd/csrc/diag.h) can be built from -- and why its matrix-instruction form was not kept
// (rejected experiments).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/lat_bench.hip -o build/lat_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));
#define N 64
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
    const int lane = threadIdx.x;
    float x = seed + lane * 1e-3f;
    double xd = seed + lane * 1e-3;
    v4f c = {x, x, x, x};
    v4d cd = {xd, xd, xd, xd};
    v4f c2 = {x, x, x, x};
    float y = x, sm = x * 0.01f, sg = x * 0.02f;
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_nop 0" :: "v"(x), "v"(xd), "v"(c), "v"(cd));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) x = __builtin_fmaf(x, 1.0001f, 0.5f);                       // dependent fp32 FMA
        if (MODE == 1) xd = __builtin_fma(xd, 1.0001, 0.5);                        // dependent fp64 FMA
        if (MODE == 2) x = __builtin_amdgcn_rsqf(x) + 1.0f;                        // rsq + add
        if (MODE == 3) xd = __builtin_amdgcn_rsq(xd) + 1.0;                        // rsq f64 + add
        if (MODE == 4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0); // MFMA chained through the accumulator
        if (MODE == 5) cd = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, xd, cd, 0, 0, 0);
        if (MODE == 6) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0); x = c[i & 3] * 0.5f; }   // MFMA -> VALU -> MFMA operand
        if (MODE == 7) { cd = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, xd, cd, 0, 0, 0); xd = cd[i & 3] * 0.5; }
        if (MODE == 8) { const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i & 63)); x = __builtin_fmaf(x, 0.5f, s); }  // readlane -> VALU
        if (MODE == 9) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0);
                         const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c[i & 3]), i & 63)); x = s * 0.5f; }   // MFMA -> readlane -> VALU -> MFMA
        if (MODE == 10) x = __shfl(x, (lane + 1) & 63) + 1.0f;                     // ds_bpermute round trip
        if (MODE == 11) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0);
                          const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c[i & 3]), i & 63));
                          float r = __builtin_amdgcn_rsqf(s); r = r * (1.5f - 0.5f * s * r * r); x = c[(i + 1) & 3] * r; }   // the leaf's column chain
        if (MODE == 12) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0);
                          const float fx = (1.0f - c2[i & 3]) * y;
                          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, fx, c2, 0, 0, 0);
                          const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c[i & 3]), i & 63));
                          float r = __builtin_amdgcn_rsqf(s); r = r * (1.5f - 0.5f * s * r * r); y = r; x = c[(i + 1) & 3] * r; }   // + the inverse's MFMA
        if (MODE == 13) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0);
                          const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c[i & 3]), i & 63));
                          float r = __builtin_amdgcn_rsqf(s); r = r * (1.5f - 0.5f * s * r * r); 
                          const float fx = (1.0f - c2[i & 3]) * y;
                          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, fx, c2, 0, 0, 0);
                          y = r; x = c[(i + 1) & 3] * r; }   // the inverse's MFMA issued late (after the chain's VALU work)
        if (MODE >= 14 && MODE <= 17) {
            // the leaf's column as written in diag.h: w, nw from the row and sm;  MFMA C;  [MFMA S];  next pivot chain
            const float crow = c[i & 3];
            const float w = crow * sm, nw = crow * -sm;
            float cdv = 0.f;
            if (MODE != 17) cdv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c[(i + 1) & 3]), (i + 1) & 63));
            __builtin_amdgcn_sched_barrier(0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(nw, w, c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const float wn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), (i + 1) & 63));
            if (MODE != 15) {
                const float xx = (y - c2[i & 3]) * sg;
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xx, c2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float dn = __builtin_fmaf(-wn, wn, MODE == 17 ? 2.0f : cdv);
            float r = __builtin_amdgcn_rsqf(dn);
            if (MODE != 16) r = r * (1.5f - 0.5f * dn * r * r);
            __builtin_amdgcn_sched_barrier(0);
            sg = r * y;
            sm = sg * x;
        }
    }
    asm volatile("s_nop 0" :: "v"(x), "v"(xd), "v"(c), "v"(cd), "v"(c2), "v"(y), "v"(sm), "v"(sg));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = x + (float)xd + c[0] + c[1] + c[2] + c[3] + c2[0] + c2[1] + c2[2] + c2[3] + y + sm + sg + (float)(cd[0] + cd[1] + cd[2] + cd[3]);
    if (lane == 0) *cyc = t1 - t0;
}
int main() {
    float* o; unsigned long long* c;
    hipMalloc(&o, 256); hipMalloc(&c, 8);
    const char* names[] = {"fp32 fma", "fp64 fma", "rsq_f32 + add", "rsq_f64 + add", "mfma f32 16x16x4 acc chain", "mfma f64 16x16x4 acc chain",
                           "mfma f32 -> mul -> operand", "mfma f64 -> mul -> operand", "readlane -> fma", "mfma f32 -> readlane -> mul -> operand",
                           "ds_bpermute + add", "leaf column chain f32 (mfma, readlane, rsq+newton, mul)", "the same + second mfma (inverse) right after the first", "the same + second mfma after the chain's VALU work", "a 16 x 16 Cholesky column by MFMA: 2 mfma, next pivot by VALU beside them", "  without the inverse's mfma",
                           "  without the Newton step", "  without the diagonal's broadcast"};
#define RUN(M) { k<M><<<1, 64>>>(o, c, 1.5f); k<M><<<1, 64>>>(o, c, 1.5f); hipDeviceSynchronize(); unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-60s %6.1f cycles per link\n", names[M], h / (double)N); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
    return 0;
}
