// What does rocprofv3's FETCH_SIZE report for the access widths of the hash-grid gather?  (MI355X_MICROARCH.md: on gfx950
// it reports HALF the bytes of a wide coalesced streaming read - 128-byte requests tallied at 64 - and "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern".)  Three kernels over an 8 GiB table,
// far beyond L2 + Infinity Cache: a coalesced 16-byte stream (known: every byte once), random 16-byte gathers and random
// 8-byte gathers (one load per lane, every lane its own 128-byte line with probability ~1).  Run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/bin/fetch_calib
// and divide each kernel's FETCH_SIZE (KB) by its load count: profiles/fetch_calib_r03.txt.
//     hipcc --offload-arch=gfx950 -O3 -o tools/bin/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void k_stream16(const float4 *__restrict__ t, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n; i += stride) { const float4 v = t[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 1234.5f) *out = s;
}
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__global__ void k_gather16(const float4 *__restrict__ t, size_t n_slots, uint32_t per_lane, float *out) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (uint32_t k = 0; k < per_lane; ++k) { const float4 v = t[mix(id * per_lane + k) % n_slots]; s += v.x + v.w; }
    if (s == 1234.5f) *out = s;
}
__global__ void k_gather8(const float2 *__restrict__ t, size_t n_slots, uint32_t per_lane, float *out) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (uint32_t k = 0; k < per_lane; ++k) { const float2 v = t[mix(id * per_lane + k + 77) % n_slots]; s += v.x + v.y; }
    if (s == 1234.5f) *out = s;
}

int main() {
    const size_t bytes = (size_t)8 << 30;
    float4 *t; float *out;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const uint32_t blocks = 256 * 16, threads = 256, per_lane = 64;
    const double loads = (double)blocks * threads * per_lane;
    hipEventRecord(e0); hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(threads), 0, 0, t, bytes / 16, out); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("k_stream16: %.3f ms, %.1f GB read -> %.2f TB/s\n", ms, bytes / 1e9, bytes / ms / 1e9);
    hipEventRecord(e0); hipLaunchKernelGGL(k_gather16, dim3(blocks), dim3(threads), 0, 0, t, bytes / 16, per_lane, out); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("k_gather16: %.3f ms, %.0f loads -> %.2f G loads/s (x64 B = %.2f TB/s, x128 B = %.2f TB/s)\n", ms, loads,
           loads / ms / 1e6, loads * 64 / ms / 1e9, loads * 128 / ms / 1e9);
    hipEventRecord(e0); hipLaunchKernelGGL(k_gather8, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<float2 *>(t), bytes / 8, per_lane, out); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("k_gather8: %.3f ms, %.0f loads -> %.2f G loads/s (x64 B = %.2f TB/s, x128 B = %.2f TB/s)\n", ms, loads,
           loads / ms / 1e6, loads * 64 / ms / 1e9, loads * 128 / ms / 1e9);
    hipDeviceSynchronize();
    return 0;
}
