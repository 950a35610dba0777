// Development probe #2: do lanes that hit the SAME 8/16/32/64-byte slot in one instruction share an L2 atomic request?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
// GROUP lanes share one aligned slot of GROUP dwords; slots are random over the table
template <int GROUP>
__global__ void k_group(float *tab, uint32_t mask_dwords, uint32_t iters) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = tid / GROUP, sub = tid % GROUP;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t slot = pcg(g * 9781u + it * 6271u) & (mask_dwords / GROUP);
        unsafeAtomicAdd(tab + (size_t)slot * GROUP + sub, 1.0f);
    }
}
template <int GROUP>
void run(float *tab, size_t dwords) {
    const uint32_t blocks = 2048, threads = 256, iters = 64;
    hipMemset(tab, 0, dwords * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_group<GROUP><<<blocks, threads>>>(tab, (uint32_t)dwords - 1, 4);
    hipMemset(tab, 0, dwords * 4);
    hipEventRecord(e0);
    k_group<GROUP><<<blocks, threads>>>(tab, (uint32_t)dwords - 1, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * threads * iters;
    std::vector<float> h(dwords); hipMemcpy(h.data(), tab, dwords * 4, hipMemcpyDeviceToHost);
    double total = 0; for (float v : h) total += v;
    printf("group=%2d lanes/slot (%3d B)  dwords=%9zu : %7.3f ms  %7.2f G dword-atomics/s  %6.2f G slots/s  sum_ok=%s\n", GROUP,
           GROUP * 4, dwords, ms, n / ms / 1e6, n / GROUP / ms / 1e6, total == n ? "yes" : "NO");
}
int main() {
    float *tab; hipMalloc(&tab, (size_t)(1 << 24) * 4);
    for (size_t dwords : {(size_t)1 << 20, (size_t)1 << 24}) {
        run<1>(tab, dwords); run<2>(tab, dwords); run<4>(tab, dwords); run<8>(tab, dwords); run<16>(tab, dwords);
        run<32>(tab, dwords); run<64>(tab, dwords);
    }
    return 0;
}
