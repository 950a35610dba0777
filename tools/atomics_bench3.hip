// Development probe #3: where do fp32 atomic adds get served?  Shared table vs one private copy per XCD, at three
// footprints (hot: 32 KB, one hashed level: 4 MB, whole gradient table: 64 MB), scaling with the number of workgroups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xFu;
}
// 4 lanes share one aligned 16-byte slot (like the scatter's quads); slots random over `dwords`; mode 0: shared table,
// mode 1: copy = XCC id, mode 2: copy = blockIdx % 8
__global__ void k_add(float *tab, uint32_t dwords, uint32_t iters, int mode, uint32_t *xcc_hist) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = tid / 4, sub = tid % 4;
    const uint32_t xcc = xcc_id();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(xcc_hist + (blockIdx.x % 8) * 16 + xcc, 1u);
    float *t = tab + (mode == 0 ? 0 : (size_t)(mode == 1 ? xcc : blockIdx.x % 8) * dwords);
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t slot = pcg(g * 9781u + it * 6271u) & (dwords / 4 - 1);
        unsafeAtomicAdd(t + (size_t)slot * 4 + sub, 1.0f);
    }
}
int main() {
    const size_t max_dwords = (size_t)1 << 24;  // 64 MB per copy
    float *tab; hipMalloc(&tab, max_dwords * 4 * 8);
    uint32_t *hist; hipMalloc(&hist, 8 * 16 * 4); hipMemset(hist, 0, 8 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t dwords : {(size_t)1 << 13, (size_t)1 << 20, (size_t)1 << 24}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (uint32_t blocks : {64u, 256u, 1024u, 4096u}) {
                const uint32_t threads = 256, iters = 64;
                hipMemset(tab, 0, dwords * 4 * 8);
                k_add<<<blocks, threads>>>(tab, (uint32_t)dwords, 2, mode, nullptr);
                hipMemset(tab, 0, dwords * 4 * 8);
                hipEventRecord(e0);
                k_add<<<blocks, threads>>>(tab, (uint32_t)dwords, iters, mode, (dwords == ((size_t)1 << 20) && blocks == 1024 && mode == 1) ? hist : nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double n = (double)blocks * threads * iters;
                std::vector<float> h(dwords * 8); hipMemcpy(h.data(), tab, dwords * 4 * 8, hipMemcpyDeviceToHost);
                double total = 0; for (float v : h) total += v;
                printf("footprint %8zu KB/copy mode=%s blocks=%5u : %8.3f ms  %7.2f G requests/s  sum_ok=%s\n", dwords * 4 / 1024,
                       mode == 0 ? "shared " : mode == 1 ? "xcc-id " : "blk%8  ", blocks, ms, n / 4 / ms / 1e6, total == n ? "yes" : "NO");
            }
        }
    }
    uint32_t hh[128]; hipMemcpy(hh, hist, sizeof hh, hipMemcpyDeviceToHost);
    printf("blockIdx%%8 (rows) vs XCC id (cols):\n");
    for (int b = 0; b < 8; ++b) { for (int x = 0; x < 8; ++x) printf("%5u", hh[b * 16 + x]); printf("\n"); }
    return 0;
}
