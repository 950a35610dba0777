// Development probe (not product code): throughput + correctness of fp32 atomic-add flavours on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/atomics_bench.hip -o tools/atomics_bench && ./tools/atomics_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

template <int FLAVOR, int PAIR>
__global__ void k_atomics(float *tab, uint32_t mask, uint32_t iters, uint32_t coherent) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t it = 0; it < iters; ++it) {
        // coherent=1: all lanes of a wave hit consecutive entries (coalesced lines); 0: fully random
        uint32_t idx = coherent ? (pcg(tid / 64 + it * 7919u) + (tid & 63)) : pcg(tid * 9781u + it * 6271u);
        idx &= mask;
        float *p = tab + (size_t)idx * 2;
#pragma unroll
        for (int f = 0; f < (PAIR ? 2 : 1); ++f) {
            if (FLAVOR == 0) unsafeAtomicAdd(p + f, 1.0f);
            if (FLAVOR == 1) __hip_atomic_fetch_add(p + f, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (FLAVOR == 2) __hip_atomic_fetch_add(p + f, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (FLAVOR == 3) atomicAdd(p + f, 1.0f);
            if (FLAVOR == 4) atomicAdd((unsigned int *)(p + f), 1u);
            if (FLAVOR == 5) { if (f == 0) atomicAdd((unsigned long long *)p, 0x0000000100000001ull); }
            if (FLAVOR == 6) { if (f == 0) unsafeAtomicAdd((double *)p, 1.0); }
        }
    }
}

template <int FLAVOR, int PAIR>
void run(const char *name, float *tab, size_t entries, uint32_t coherent) {
    const uint32_t blocks = 256 * 8, threads = 256, iters = 64;
    hipMemset(tab, 0, entries * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_atomics<FLAVOR, PAIR><<<blocks, threads>>>(tab, (uint32_t)entries - 1, 4, coherent);  // warm
    hipMemset(tab, 0, entries * 8);
    hipEventRecord(e0);
    k_atomics<FLAVOR, PAIR><<<blocks, threads>>>(tab, (uint32_t)entries - 1, iters, coherent);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * threads * iters * (PAIR && FLAVOR < 5 ? 2 : 1);
    // correctness: total must equal the number of adds (flavours 0-3: float 1.0 each)
    double total = -1;
    if (FLAVOR <= 3) {
        std::vector<float> h(entries * 2);
        hipMemcpy(h.data(), tab, entries * 8, hipMemcpyDeviceToHost);
        total = 0; for (float v : h) total += v;
    }
    printf("%-34s entries=%9zu coherent=%u pair=%d : %8.3f ms  %7.2f G atomics/s  sum_ok=%s\n", name, entries, coherent,
           PAIR, ms, n / ms / 1e6, FLAVOR <= 3 ? (total == n ? "yes" : "NO") : "n/a");
}

int main() {
    float *tab; hipMalloc(&tab, (size_t)(1 << 23) * 8);
    for (size_t entries : {(size_t)1 << 12, (size_t)1 << 19, (size_t)1 << 23}) {
        for (uint32_t coh : {0u, 1u}) {
            run<0, 1>("unsafeAtomicAdd(f32)", tab, entries, coh);
            run<1, 1>("hip_atomic relaxed agent", tab, entries, coh);
            run<2, 1>("hip_atomic relaxed workgroup", tab, entries, coh);
            run<3, 1>("atomicAdd(f32)", tab, entries, coh);
            run<4, 1>("atomicAdd(u32)", tab, entries, coh);
            run<5, 1>("atomicAdd(u64) one per pair", tab, entries, coh);
            run<6, 1>("unsafeAtomicAdd(f64) one per pair", tab, entries, coh);
            run<0, 0>("unsafeAtomicAdd(f32) single", tab, entries, coh);
        }
    }
    return 0;
}
