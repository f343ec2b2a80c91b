// What a hand-off between two workgroups of the one-launch factorisation costs, in parts (chol_dag.hip: dag_publish /
// dag_wait / the strip products of the chain team).  Two workgroups on different XCDs; the producer writes a tile with
// write-through stores, drains, sets a flag; the consumer polls the flag the way dag_wait does, then fetches `bytes` of
// the tile into LDS by DMA with every load in flight at once, and reports when the last byte has arrived.  Variants of
// what is fetched: FRESH (just written by the other workgroup), OLD (written by the other workgroup long ago, never read
// here), WARM (read here once before: in this XCD's L2), and old bytes through the strip products' 3-deep pipeline of 8 KB stages.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/handoff_bench.hip -o build/handoff_bench && build/handoff_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int TILE_BYTES = 65536;
constexpr int ITERS = 64;
constexpr long long SPIN_LIMIT = 200000000;      // 2 s at 100 MHz

struct Args {
    float* tiles;        // ITERS + 1 tiles of 64 KB, 1 MB apart (a new one per iteration: nothing is ever L2-resident by accident)
    float* old_tiles;    // written once by the producer before the loop
    int* flag;           // producer -> consumer
    int* ack;            // consumer -> producer
    int* abort_word;
    long long* out;      // per iteration: [0] stores + drain, [1] time the flag was seen, [2] fetch fresh, [3] fetch old, [4] fetch warm, [5] old, pipelined, [6] values ok
    int mode_bytes;      // bytes fetched in the all-at-once variants
    int wide_stores;     // producer: 16-byte stores instead of 4-byte ones
    long long* stamps;   // producer's flag-store time per iteration
};

__device__ __forceinline__ bool spin_until(const int* p, int want, int* abort_word) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 0;; ++spins) {
        const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= want) return true;
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if ((spins & 255) == 255 && (long long)(__builtin_amdgcn_s_memrealtime() - t0) > SPIN_LIMIT) {
            atomicCAS(abort_word, 0, 1);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// every wave-load is 64 lanes x 16 bytes = 1 KB contiguous
__device__ __forceinline__ void dma_all(char* smem, const char* src, int bytes) {
    const int tid = threadIdx.x;
    for (int off = 0; off < bytes; off += 256 * 16)
        __builtin_amdgcn_global_load_lds((glb_vp)(src + off + tid * 16), (lds_vp)(smem + off + (tid >> 6) * 1024), 16, 0, 0);
}

__global__ __launch_bounds__(256, 1) void handoff_kernel(Args g) {
    __shared__ __attribute__((aligned(1024))) char smem[TILE_BYTES];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0) {
        // ---- producer ----
        for (int i = tid; i < (ITERS + 1) * (TILE_BYTES / 4); i += 256) {
            const int it = i / (TILE_BYTES / 4), e = i % (TILE_BYTES / 4);
            __hip_atomic_store(g.old_tiles + (size_t)it * 262144 + e, (float)(e & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < ITERS; ++it) {
            if (wave == 0) {
                bool ok = spin_until(g.ack, it, g.abort_word);
                if (lane == 0) s_ok = ok;
            }
            __syncthreads();
            if (!s_ok) return;
            __syncthreads();
            float* t = g.tiles + (size_t)it * 262144;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            if (g.wide_stores) {
                for (int off = tid * 4; off < TILE_BYTES / 4; off += 1024) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    f4 v = {(float)it, (float)it, (float)it, (float)it};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(t + off), "v"(v) : "memory");
                }
            } else {
                // the accumulator pattern of the tile products: a store instruction covers 4 rows x 16 floats
                for (int i = 0; i < 64; ++i) {
                    const int row = (i >> 2) * 8 + wave * 2 + ((lane >> 4) >> 1), col = (i & 3) * 32 + ((lane >> 4) & 1) * 16 + (lane & 15);
                    __hip_atomic_store(t + row * 128 + col, (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            __syncthreads();
            if (tid == 0) {
                g.out[it * 8 + 0] = (long long)(t1 - t0);
                g.stamps[it] = (long long)__builtin_amdgcn_s_memrealtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(g.flag, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // ---- consumer ----
    for (int it = 0; it < ITERS; ++it) {
        const char* fresh = (const char*)(g.tiles + (size_t)it * 262144);
        const char* old = (const char*)(g.old_tiles + (size_t)it * 262144);
        if (tid == 0) __hip_atomic_store(g.ack, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long t_seen = 0;
        if (wave == 0) {
            bool ok = spin_until(g.flag, it + 1, g.abort_word);
            t_seen = __builtin_amdgcn_s_memrealtime();
            if (ok) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (lane == 0) s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (g.mode_bytes > 0) dma_all(smem, fresh, g.mode_bytes);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const float seen = *reinterpret_cast<volatile float*>(smem + 4 * (tid & 255));
        __syncthreads();
        // old data, never read on this XCD
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        dma_all(smem, old, g.mode_bytes);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
        // the same bytes again: now in this XCD's L2
        dma_all(smem, old, g.mode_bytes);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
        // old bytes never read here (the second half of this iteration's 1 MB slot) through 8 KB stages, three in flight:
        // the strip products' pipeline
        const char* src = old + 524288;
        const unsigned long long t5 = __builtin_amdgcn_s_memrealtime();
        {
            const int nst = g.mode_bytes / 8192;                       // 8 KB per stage
            for (int s = 0; s < 3 && s < nst; ++s) dma_all(smem + s * 8192, src + s * 8192, 8192);
            for (int s = 0; s < nst; ++s) {
                const int younger = nst - 1 - s;
                if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F74);        // two DMA per stage and wave
                else if (younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F72);
                else __builtin_amdgcn_s_waitcnt(0x0F70);
                __builtin_amdgcn_s_barrier();
                if (s + 3 < nst) dma_all(smem + ((s + 3) & 3) * 8192, src + (s + 3) * 8192, 8192);
            }
        }
        __syncthreads();
        const unsigned long long t6 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            g.out[it * 8 + 1] = (long long)t_seen;
            g.out[it * 8 + 2] = (long long)(t1 - t0);
            g.out[it * 8 + 3] = (long long)(t3 - t2);
            g.out[it * 8 + 4] = (long long)(t4 - t3);
            g.out[it * 8 + 5] = (long long)(t6 - t5);
            g.out[it * 8 + 6] = seen == (float)it ? 1 : 0;
        }
        __syncthreads();
    }
}

int main() {
    Args g;
    CHECK(hipMalloc(&g.tiles, (size_t)(ITERS + 2) * 1048576));
    CHECK(hipMalloc(&g.old_tiles, (size_t)(ITERS + 2) * 1048576));
    CHECK(hipMemset(g.old_tiles, 0, (size_t)(ITERS + 2) * 1048576));
    int* words;
    CHECK(hipMalloc(&words, 4096));
    CHECK(hipMalloc(&g.out, ITERS * 8 * sizeof(long long)));
    CHECK(hipMalloc(&g.stamps, ITERS * sizeof(long long)));
    g.flag = words;
    g.ack = words + 64;
    g.abort_word = words + 128;
    printf("all times in us (100 MHz counter), median [p10 p90] over %d hand-offs, two workgroups on different XCDs\n", ITERS);
    for (int wide = 0; wide < 2; ++wide)
        for (int bytes : {8192, 16384, 32768, 65536}) {
            g.mode_bytes = bytes;
            g.wide_stores = wide;
            CHECK(hipMemset(words, 0, 4096));
            CHECK(hipMemset(g.out, 0, ITERS * 8 * sizeof(long long)));
            CHECK(hipMemset(g.tiles, 0, (size_t)(ITERS + 2) * 1048576));
            int m1 = -1;
            CHECK(hipMemcpy(g.ack, &m1, 4, hipMemcpyHostToDevice));
            handoff_kernel<<<2, 256>>>(g);
            CHECK(hipDeviceSynchronize());
            std::vector<long long> out(ITERS * 8), st(ITERS);
            CHECK(hipMemcpy(out.data(), g.out, out.size() * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(st.data(), g.stamps, st.size() * 8, hipMemcpyDeviceToHost));
            int ab = 0;
            CHECK(hipMemcpy(&ab, g.abort_word, 4, hipMemcpyDeviceToHost));
            if (ab) { printf("stalled\n"); return 1; }
            auto stat = [&](int col, bool rel) {
                std::vector<double> v;
                for (int it = 8; it < ITERS; ++it) v.push_back((rel ? out[it * 8 + col] - st[it] : out[it * 8 + col]) / 100.0);
                std::sort(v.begin(), v.end());
                static char buf[6][64];
                static int bi = 0;
                char* b = buf[bi++ % 6];
                snprintf(b, 64, "%5.2f [%5.2f %5.2f]", v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
                return b;
            };
            int okc = 0;
            for (int it = 0; it < ITERS; ++it) okc += (int)out[it * 8 + 6];
            printf("%s stores of the 64 KB tile, %2d KB fetched: drain %s  flag store->seen %s  fetch fresh %s  old %s  L2-warm %s  old, 8 KB stages 3 in flight %s  (values ok %d/%d)\n",
                   wide ? "16-byte" : " 4-byte", bytes / 1024, stat(0, false), stat(1, true), stat(2, false), stat(3, false), stat(4, false),
                   stat(5, false), okc, ITERS);
        }
    return 0;
}
