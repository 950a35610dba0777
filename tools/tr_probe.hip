// Probe of gfx950's transposing LDS read and of csrc/lds_transpose.h's tile round trip (what the MLP backward relies on).
//   hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/bin/tr_probe && tools/bin/tr_probe
// 1. raw exchange pattern: halfword i of LDS holds i, lane l reads the 8 bytes at 8 l: value j of lane l must be
//    64 (l >> 4) + 16 j + (l & 15);
// 2. a 32 x 32 tile held "lane = sample" (both value orders) written and read back "lane = feature", bit for bit;
// 3. what the round trip costs beside the identity product it replaces (cycles per tile of one wave, s_memtime);
// 4. the round trip's throughput on the whole chip against the row pitch of the tile image.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../make-it-3d_amd/csrc/lds_transpose.h"

using namespace mi3d_tr;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__host__ __device__ inline int rowmap(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

__global__ void k_raw(unsigned short *out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<lds_short4_ptr>(to_lds(img) + 8 * lane));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}

template <bool KIND_X>
__global__ void k_tile(const _Float16 *in /* [64][16] */, _Float16 *out /* [64][16] */, int reps, long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_ptr tile = to_lds(lds) + wave * kTileBytes;
    half8v a0, a1, b0, b1;
    for (int i = 0; i < 8; ++i) { a0[i] = in[lane * 16 + i]; a1[i] = in[lane * 16 + 8 + i]; }
    lds_ptr wr = tile + (KIND_X ? write_offset_x(lane) : write_offset_d(lane)), rd = tile + read_offset(lane);
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < 2 * reps - 1; ++r) {   // an odd number of transpositions = one (value order D; X: reps = 1 only)
        write_tile<KIND_X>(wr, a0, a1);
        read_tile(rd, b0, b1);
        a0 = b0; a1 = b1;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (wave == 0) {
        for (int i = 0; i < 8; ++i) { out[lane * 16 + i] = b0[i]; out[lane * 16 + 8 + i] = b1[i]; }
        if (lane == 0) cycles[0] = t1 - t0;
    }
}

// the identity product the round trip replaces (timing only)
__global__ void k_mfma(const _Float16 *in, _Float16 *out, int reps, long long *cycles) {
    const int lane = threadIdx.x & 63;
    half8v a0, a1, id0, id1;
    for (int i = 0; i < 8; ++i) {
        a0[i] = in[lane * 16 + i]; a1[i] = in[lane * 16 + 8 + i];
        id0[i] = (_Float16)((lane & 31) == rowmap(i, lane >> 5) ? 1.f : 0.f);
        id1[i] = (_Float16)((lane & 31) == rowmap(8 + i, lane >> 5) ? 1.f : 0.f);
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < 2 * reps - 1; ++r) {
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, id0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, id1, acc, 0, 0, 0);
        for (int q = 0; q < 8; ++q) { a0[q] = (_Float16)acc[q]; a1[q] = (_Float16)acc[8 + q]; }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x >> 6) == 0) {
        for (int i = 0; i < 8; ++i) { out[lane * 16 + i] = a0[i]; out[lane * 16 + 8 + i] = a1[i]; }
        if (lane == 0) cycles[0] = t1 - t0;
    }
}


// 4. throughput of the round trip on a full chip (8 waves per CU, as the MLP backward runs) against the row pitch of the
//    tile image - what the 8 bytes of padding buy, and whether another pitch would do better
template <int ROW>
__global__ __launch_bounds__(256) void k_pitch(const _Float16 *in, _Float16 *out, int reps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_ptr tile = to_lds(lds) + wave * (32 * ROW);
    half8v a0, a1;
    for (int i = 0; i < 8; ++i) { a0[i] = in[lane * 16 + i]; a1[i] = in[lane * 16 + 8 + i]; }
    lds_ptr wr = tile + ROW * (lane & 31) + 8 * (lane >> 5);
    const int t = lane & 15, g = lane >> 4;
    lds_ptr rd = tile + ROW * (4 * (lane >> 5) + (t >> 2)) + 8 * (4 * (g & 1) + (t & 3));
    for (int r = 0; r < reps; ++r) {
        *reinterpret_cast<lds_half4_ptr>(wr + 0) = __builtin_shufflevector(a0, a0, 0, 1, 2, 3);
        *reinterpret_cast<lds_half4_ptr>(wr + 16) = __builtin_shufflevector(a0, a0, 4, 5, 6, 7);
        *reinterpret_cast<lds_half4_ptr>(wr + 32) = __builtin_shufflevector(a1, a1, 0, 1, 2, 3);
        *reinterpret_cast<lds_half4_ptr>(wr + 48) = __builtin_shufflevector(a1, a1, 4, 5, 6, 7);
        asm volatile("" ::: "memory");
        const half4v p = tr_read(rd + 0 * 8 * ROW), q = tr_read(rd + 1 * 8 * ROW), u = tr_read(rd + 2 * 8 * ROW), v = tr_read(rd + 3 * 8 * ROW);
        asm volatile("" ::: "memory");
        a0 = __builtin_shufflevector(p, q, 0, 1, 2, 3, 4, 5, 6, 7);
        a1 = __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    if (blockIdx.x == 0 && wave == 0) for (int i = 0; i < 8; ++i) { out[lane * 16 + i] = a0[i]; out[lane * 16 + 8 + i] = a1[i]; }
}
template <int ROW>
static int pitch_case(const _Float16 *d_in, _Float16 *d_out) {
    const int reps = 4001;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_pitch<ROW>, dim3(512), dim3(256), 4 * 32 * ROW, 0, d_in, d_out, 11);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_pitch<ROW>, dim3(512), dim3(256), 4 * 32 * ROW, 0, d_in, d_out, reps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<_Float16> out(1024);
    hipMemcpy(out.data(), d_out, 2048, hipMemcpyDeviceToHost);
    int tb = 0;   // an odd number of transpositions of the D-order tile = one
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q) tb += (float)out[l * 16 + q] != (float)(32 * rowmap(q, l >> 5) + (l & 31));
    printf("row pitch %3d bytes: %.3f ms for %d round trips x 8 waves per CU = %.1f ns per round trip and CU%s\n", ROW, ms, reps,
           ms * 1e6 / (reps * 8.0), tb ? "  WRONG RESULT" : "");
    return tb != 0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main() {
    int bad = 0;
    unsigned short *d_raw; CK(hipMalloc(&d_raw, 256 * 2));
    hipLaunchKernelGGL(k_raw, dim3(1), dim3(64), 0, 0, d_raw);
    unsigned short raw[256]; CK(hipMemcpy(raw, d_raw, sizeof raw, hipMemcpyDeviceToHost));
    int raw_bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) raw_bad += raw[l * 4 + j] != 64 * (l >> 4) + 16 * j + (l & 15);
    printf("raw exchange pattern: %s\n", raw_bad ? "DIFFERENT" : "as assumed");
    if (raw_bad) {
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %3d %3d %3d %3d\n", l, raw[l * 4], raw[l * 4 + 1], raw[l * 4 + 2], raw[l * 4 + 3]);
        bad = 1;
    }
    _Float16 *d_in, *d_out; long long *d_cyc;
    CK(hipMalloc(&d_in, 1024 * 2)); CK(hipMalloc(&d_out, 1024 * 2)); CK(hipMalloc(&d_cyc, 8));
    for (int kind = 0; kind < 2; ++kind) {
        std::vector<_Float16> in(1024), out(1024);
        for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q) {   // the value = 32 sample + feature (exact in binary16 below 2048)
            const int s = l & 31, h = l >> 5, f = kind ? 16 * h + q : rowmap(q, h);
            in[l * 16 + q] = (_Float16)(float)(32 * s + f);
        }
        CK(hipMemcpy(d_in, in.data(), 2048, hipMemcpyHostToDevice));
        if (kind) hipLaunchKernelGGL(k_tile<true>, dim3(1), dim3(256), 4 * kTileBytes, 0, d_in, d_out, 1, d_cyc);
        else hipLaunchKernelGGL(k_tile<false>, dim3(1), dim3(256), 4 * kTileBytes, 0, d_in, d_out, 1, d_cyc);
        CK(hipMemcpy(out.data(), d_out, 2048, hipMemcpyDeviceToHost));
        int tb = 0;
        for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q)
            tb += (float)out[l * 16 + q] != (float)(32 * rowmap(q, l >> 5) + (l & 31));
        printf("tile round trip, value order %s: %s (%d of 1024 differ)\n", kind ? "X" : "D", tb ? "WRONG" : "exact", tb);
        if (tb) {
            bad = 1;
            for (int l = 0; l < 64; l += 9) { printf("  lane %2d:", l); for (int q = 0; q < 16; ++q) printf(" %4d", (int)(float)out[l * 16 + q]); printf("\n"); }
        }
    }
    // timing: 1 wave and 4 waves per workgroup, one workgroup (latency-bound chains, both) - per transposition
    for (int waves = 1; waves <= 4; waves *= 4) {
        long long c_lds = 0, c_mfma = 0;
        const int reps = 1000;
        hipLaunchKernelGGL(k_tile<false>, dim3(1), dim3(64 * waves), 4 * kTileBytes, 0, d_in, d_out, reps, d_cyc);
        CK(hipMemcpy(&c_lds, d_cyc, 8, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64 * waves), 0, 0, d_in, d_out, reps, d_cyc);
        CK(hipMemcpy(&c_mfma, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("%d wave(s) on one CU, dependent chain of %d transpositions: LDS %.1f, identity MFMA %.1f counter ticks each\n", waves,
               2 * reps - 1, (double)c_lds / (2 * reps - 1), (double)c_mfma / (2 * reps - 1));
    }
    {   // (the D-order input of the tile test is what d_in holds after the loop above? no: reload it)
        std::vector<_Float16> in(1024);
        for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q) in[l * 16 + q] = (_Float16)(float)(32 * (l & 31) + rowmap(q, l >> 5));
        CK(hipMemcpy(d_in, in.data(), 2048, hipMemcpyHostToDevice));
        bad |= pitch_case<64>(d_in, d_out);
        bad |= pitch_case<72>(d_in, d_out);
        bad |= pitch_case<80>(d_in, d_out);
        bad |= pitch_case<88>(d_in, d_out);
        bad |= pitch_case<104>(d_in, d_out);
        bad |= pitch_case<136>(d_in, d_out);
    }
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad;
}
