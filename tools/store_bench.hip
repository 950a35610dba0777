// Development probe: what does the chip charge for appending 16-byte records to many per-wave regions?  Every wave owns
// R regions (like the emit's (wave, bin) regions of one or several levels); per iteration its 64 lanes append 64 records
// in pieces of PIECE consecutive records (PIECE = 1: every lane picks its own random region - the emit as it is; 4 / 8:
// PIECE lanes write one contiguous 64- / 128-byte piece to one region).  Same bytes in every variant.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
template <int PIECE>
__global__ __launch_bounds__(256) void k(uint4 *arena, uint32_t R, uint32_t cap, uint32_t iters) {
    extern __shared__ uint32_t cnt_all[];  // [4][R]
    const uint32_t lane = threadIdx.x & 63, w_in = threadIdx.x >> 6, gw = blockIdx.x * 4 + w_in;
    uint32_t *cnt = cnt_all + w_in * R;
    for (uint32_t r = lane; r < R; r += 64) cnt[r] = 0;
    __builtin_amdgcn_wave_barrier();
    uint4 *mine = arena + (size_t)gw * R * cap;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t g = lane / PIECE, j = lane % PIECE;
        const uint32_t region = pcg(gw * 7919u + it * 64u + g) % R;
        uint32_t slot = 0;
        if (j == 0) slot = atomicAdd(&cnt[region], (uint32_t)PIECE);
        slot = __shfl(slot, (int)(g * PIECE), 64) + j;
        if (slot < cap) mine[(size_t)region * cap + slot] = make_uint4(it, lane, region, slot);
    }
}
template <int PIECE>
void run(uint4 *arena, uint32_t waves, uint32_t R, uint32_t iters) {
    const uint32_t cap = (iters * 64 / R) * 2 + 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PIECE><<<waves / 4, 256, 4 * R * 4>>>(arena, R, cap, 8);
    (void)hipEventRecord(e0);
    k<PIECE><<<waves / 4, 256, 4 * R * 4>>>(arena, R, cap, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double recs = (double)waves * iters * 64;
    printf("piece %d  waves %5u  regions/wave %4u  open lines %7.1f MB : %8.3f ms  %7.1f G records/s  %.3f records/clk/CU  %.2f TB/s\n",
           PIECE, waves, R, (double)waves * R * 128 / 1e6, ms, recs / ms / 1e6, recs / ms / 1e6 / (2.4 * 256), recs * 16 / ms / 1e9);
}
int main() {
    const size_t bytes = (size_t)24 << 30;
    uint4 *arena; if (hipMalloc(&arena, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(arena, 0, bytes);
    for (uint32_t waves : {1536u, 3072u})
        for (uint32_t R : {64u, 512u}) {
            const uint32_t iters = (uint32_t)(((size_t)8 << 30) / 16 / 64 / waves);  // 8 GB of records per run
            run<1>(arena, waves, R, iters);
            run<4>(arena, waves, R, iters);
            run<8>(arena, waves, R, iters);
        }
    return 0;
}
