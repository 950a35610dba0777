// Development probe: LDS atomic-add throughput on gfx950 (fp32 vs u32 vs u64; random / conflict-free / same-address).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
template <int KIND, int PATTERN>
__global__ __launch_bounds__(1024) void k(float *out, uint32_t iters) {
    __shared__ unsigned long long acc64[8192];
    float *accf = reinterpret_cast<float *>(acc64);
    uint32_t *accu = reinterpret_cast<uint32_t *>(acc64);
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) acc64[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t a;
        if (PATTERN == 0) a = pcg(tid * 9781u + it * 6271u) & 16383u;          // random over 16K dwords
        else if (PATTERN == 1) a = ((it * 64u) & 16383u) + lane;                 // lane-linear: conflict free
        else if (PATTERN == 2) a = it & 16383u;                                  // all lanes same address
        else a = (pcg(tid * 9781u + it * 6271u) & 8191u) * 2u;                   // random 8-byte slots (even dwords)
        if (KIND == 0) atomicAdd(&accf[a], 1.0f);
        else if (KIND == 1) atomicAdd(&accu[a], 1u);
        else if (KIND == 2) atomicAdd(&acc64[a >> 1], 1ull);
        else { atomicAdd(&accf[a], 1.0f); atomicAdd(&accf[a ^ 1u], 1.0f); }      // the pair the reduce kernel issues
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = accf[threadIdx.x] + (float)acc64[threadIdx.x + 64];
}
template <int KIND, int PATTERN>
void run(const char *name, float *out) {
    const uint32_t blocks = 512, threads = 1024, iters = 2048;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND, PATTERN><<<blocks, threads>>>(out, 16);
    (void)hipEventRecord(e0);
    k<KIND, PATTERN><<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * threads * iters * (KIND == 3 ? 2 : 1);
    printf("%-44s %8.3f ms  %8.1f G lane-atomics/s  (%.2f per clk per CU @2.1GHz x256)\n", name, ms, lane_ops / ms / 1e6,
           lane_ops / ms / 1e6 / (2.1 * 256));
}
int main() {
    float *out; (void)hipMalloc(&out, 512 * 64 * 4);
    run<0, 0>("ds_add_f32  random", out);
    run<1, 0>("ds_add_u32  random", out);
    run<2, 3>("ds_add_u64  random 8-B slots", out);
    run<3, 3>("ds_add_f32 x2 (pair g0,g1) random slots", out);
    run<0, 1>("ds_add_f32  conflict-free", out);
    run<1, 1>("ds_add_u32  conflict-free", out);
    run<0, 2>("ds_add_f32  same address", out);
    run<1, 2>("ds_add_u32  same address", out);
    return 0;
}
