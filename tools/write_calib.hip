// What does rocprofv3's WRITE_SIZE report for the store patterns of the scatter's emit?  (MI355X_MICROARCH.md: "other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".)  Round 4's emit
// stores 12-byte records (global_store_dwordx3) in sorted runs - a wave's 64 lanes write 64 consecutive 12-byte slots,
// cut into ~8-record runs that start at arbitrary 12-byte offsets of their regions - where round 3 stored 16-byte records;
// its WRITE_SIZE came out at 282 GB per launch for ~23 GB of records.  Six kernels, each writing a KNOWN byte count once
// into an 8 GiB buffer (far beyond L2 + Infinity Cache):
//   k_w16_stream   16 B per lane, fully coalesced          k_w12_stream   12 B per lane, fully coalesced (768 B per wave)
//   k_w16_runs8    16-byte records in runs of 8 at random 16-byte-aligned places
//   k_w12_runs8    12-byte records in runs of 8 at random 4-byte-aligned places  (the emit's pattern)
//   k_w12_single   one 12-byte record per lane at a random place                 (the emit's direct appends)
//   k_w4_stream    4 B per lane coalesced (the gradient planes' binary16 pairs)
// Run under   rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -- tools/bin/write_calib   and divide each kernel's
// WRITE_SIZE (KB) by the bytes it wrote: profiles/write_calib_r04.txt.
//     hipcc --offload-arch=gfx950 -O3 -o tools/bin/write_calib tools/write_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct R12 { uint32_t a, b, c; };
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__global__ void k_w16_stream(uint4 *t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void k_w12_stream(R12 *t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = R12{(uint32_t)i, 1u, 2u};
}
__global__ void k_w4_stream(uint32_t *t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = (uint32_t)i;
}
// every group of 8 consecutive lanes writes 8 consecutive records at a random place (`per_lane` rounds)
__global__ void k_w16_runs8(uint4 *t, size_t n_slots, uint32_t per_lane) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, grp = id >> 3, in = id & 7;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const size_t base = mix(grp * per_lane + k) % (n_slots - 8);
        t[base + in] = make_uint4((uint32_t)id, k, 2u, 3u);
    }
}
__global__ void k_w12_runs8(R12 *t, size_t n_slots, uint32_t per_lane) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, grp = id >> 3, in = id & 7;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const size_t base = mix(grp * per_lane + k + 13) % (n_slots - 8);
        t[base + in] = R12{(uint32_t)id, k, 2u};
    }
}
__global__ void k_w12_single(R12 *t, size_t n_slots, uint32_t per_lane) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) t[mix(id * per_lane + k + 29) % n_slots] = R12{(uint32_t)id, k, 2u};
}

template <class F> static void timed(const char *name, double bytes, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("%-13s %.3f ms, %.3f GB written -> %.2f TB/s\n", name, ms, bytes / 1e9, bytes / ms / 1e9);
}

int main() {
    const size_t bytes = (size_t)8 << 30;
    void *t;
    if (hipMalloc(&t, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 0, bytes);
    hipDeviceSynchronize();
    const uint32_t blocks = 256 * 16, threads = 256, per_lane = 64;
    const double lanes = (double)blocks * threads;
    timed("k_w16_stream", (double)bytes, [&] { hipLaunchKernelGGL(k_w16_stream, dim3(blocks), dim3(threads), 0, 0, (uint4 *)t, bytes / 16); });
    timed("k_w12_stream", (double)(bytes / 12 * 12), [&] { hipLaunchKernelGGL(k_w12_stream, dim3(blocks), dim3(threads), 0, 0, (R12 *)t, bytes / 12); });
    timed("k_w4_stream", (double)bytes, [&] { hipLaunchKernelGGL(k_w4_stream, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)t, bytes / 4); });
    timed("k_w16_runs8", lanes * per_lane * 16, [&] { hipLaunchKernelGGL(k_w16_runs8, dim3(blocks), dim3(threads), 0, 0, (uint4 *)t, bytes / 16, per_lane); });
    timed("k_w12_runs8", lanes * per_lane * 12, [&] { hipLaunchKernelGGL(k_w12_runs8, dim3(blocks), dim3(threads), 0, 0, (R12 *)t, bytes / 12, per_lane); });
    timed("k_w12_single", lanes * per_lane * 12, [&] { hipLaunchKernelGGL(k_w12_single, dim3(blocks), dim3(threads), 0, 0, (R12 *)t, bytes / 12, per_lane); });
    hipDeviceSynchronize();
    return 0;
}
