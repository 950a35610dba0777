// Multiresolution hash-grid encoding, forward and parameter-gradient backward, for MI355X (gfx950).
// Replaces the tiny-cuda-nn `Encoding` the reference instantiates at
// /root/reference/nerf/network_tcnn.py:54-65 (Part 2 of include/mi3d.h) and adds the multi-point form the
// fused field uses (P stencil points per sample, network_tcnn.py:115-128).
//
// FORWARD (gather).  workgroup = 4 waves, tile = 64 consecutive samples; lane = sample, wave w walks levels
// w, w+4, ... so the level is wave-uniform (level constants in SGPRs, uniform hashed/dense branch, and the 64
// lanes of one gather instruction hit ONE level's table with neighbouring samples of a ray).  The [64 x 2L]
// feature tile is transposed through LDS so the [rows, 2L] feature matrix is written in full 128-byte lines.
//
// BACKWARD (scatter).  Measured on MI355X (tools/atomics_bench*.hip, profiles/atomics_r01.txt): the L2 retires
// ~21 G atomic REQUESTS/s, a request being one aligned 64-byte block touched by one wave instruction, no matter
// how many of its 16 dwords are hit (1 lane: 21 G adds/s; 16 lanes: 320 G adds/s), any scope, any table size.
// The scatter is therefore organised around requests, not adds - see k_scatter below: lane quads put the
// 2 features x 2 x-neighbours of a corner (one 64-byte block 7 times out of 8, because x only enters the low
// bits of both the dense and the hashed index) into one request, and equal-cell runs of neighbouring samples
// are summed in registers before they leave.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include "../../include/mi3d.h"
#include "mi3d_grid.h"

using namespace mi3d;

namespace {

constexpr int kWave = 64;
constexpr int kTile = 64;   // samples per workgroup
constexpr int kWaves = 4;   // waves per workgroup
constexpr int kMaxFeat = MI3D_MAX_LEVELS * 2;
constexpr int kMaxPts = MI3D_MAX_POINTS;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Where the P evaluation points of a sample sit.  mode 0: the raw input (tcnn.Encoding: x already in [0,1]).
// mode 1: world-space stencil - point p = clamp(base + offs[p], -bound, bound), base = x for p < P0 else x2,
// then mapped to [0,1] as (pt + bound) / (2 bound)  (network_tcnn.py:106,117-122).
struct PointSet {
    const float *x, *x2;
    float offs[kMaxPts * 3];
    uint32_t P0, P;
    float bound;
    int mode;
};

__device__ __forceinline__ void load_bases(const PointSet &ps, uint32_t s, bool valid, float (&b)[2][3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        b[0][d] = valid ? ps.x[(size_t)s * 3 + d] : 0.f;
        b[1][d] = (valid && ps.x2 != nullptr) ? ps.x2[(size_t)s * 3 + d] : 0.f;
    }
}
__device__ __forceinline__ void point_of(const PointSet &ps, const float (&b)[2][3], uint32_t p, float (&q)[3]) {
    const int which = p < ps.P0 ? 0 : 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (ps.mode == 0) {
            q[d] = b[which][d];
        } else {
            const float w = clampf(b[which][d] + ps.offs[p * 3 + d], -ps.bound, ps.bound);
            q[d] = (w + ps.bound) / (2.0f * ps.bound);
        }
    }
}

// ---------------------------------------------------------------- forward
__global__ __launch_bounds__(kWave *kWaves) void k_grid_encode(PointSet ps, uint32_t n, const int32_t *count,
                                                                const float2 *__restrict__ table, GridTable T,
                                                                float *__restrict__ out) {
    __shared__ float tile[kTile * (kMaxFeat + 1)];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    if (count != nullptr) { const uint32_t c = (uint32_t)max(*count, 0); n = c < n ? c : n; }
    const uint32_t F = T.n_levels * 2, stride = F + 1;
    const uint32_t s0 = blockIdx.x * kTile, s = s0 + lane;
    if (s0 >= n) return;
    const bool valid = s < n;
    const uint32_t rows = (n - s0) < (uint32_t)kTile ? (n - s0) : (uint32_t)kTile;
    float base[2][3];
    load_bases(ps, s, valid, base);

    for (uint32_t p = 0; p < ps.P; ++p) {
        float q[3];
        point_of(ps, base, p, q);
        for (uint32_t l = wave; l < T.n_levels; l += kWaves) {
            const GridLevel L = T.level[l];
            float r0 = 0.f, r1 = 0.f;
            if (valid) {
                Corners c;
                grid_corners(L, q[0], q[1], q[2], c);
                const float2 *lvl = table + L.offset;
                float2 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = lvl[c.idx[k]];  // 8 independent 8-byte gathers in flight
#pragma unroll
                for (int k = 0; k < 8; ++k) { r0 += c.w[k] * v[k].x; r1 += c.w[k] * v[k].y; }
            }
            tile[lane * stride + 2 * l] = r0;
            tile[lane * stride + 2 * l + 1] = r1;
        }
        __syncthreads();
        // row of (sample i, point p) is i*P + p: one full 2L-float line per row
        for (uint32_t e = threadIdx.x; e < rows * F; e += blockDim.x) {
            const uint32_t i = e / F, f = e % F;
            out[((size_t)(s0 + i) * ps.P + p) * F + f] = tile[i * stride + f];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- backward
// One kernel, every level.  A QUAD of lanes serves one (sample, point, level): lane (dx, f) of the quad owns
// feature f of the two corners x+dx, so the 4 dwords of an x-neighbour pair leave in ONE request.  A wave
// instruction covers 16 CONSECUTIVE SAMPLES of the SAME stencil point; on levels whose cells are longer than a
// marching step, neighbouring quads sit in the same cell and are summed by a segmented scan across quads
// (lane strides 4, 8, 16, 32) before the last quad of each run issues the 4 requests.
template <int NV>
__device__ __forceinline__ bool quad_merge_runs(bool active, const CellKey &key, float (&v)[NV], int lane) {
    CellKey prev;
    prev.a = __shfl_up(key.a, 4, 64);
    prev.b = __shfl_up(key.b, 4, 64);
    prev.c = __shfl_up(key.c, 4, 64);
    const bool prev_active = __shfl_up((int)active, 4, 64) != 0;
    const bool joins_prev = active && prev_active && lane >= 4 && (key == prev);
    const unsigned long long joins = __ballot(joins_prev);
    if (joins != 0ull) {
        bool head = !joins_prev;
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {
            const bool up_head = __shfl_up((int)head, off, 64) != 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float up = __shfl_up(v[i], off, 64);
                if (!head && lane >= off) v[i] += up;
            }
            head = head || (lane < off) || up_head;
        }
    }
    const bool next_joins = (lane < 60) && (((joins >> (lane + 4)) & 1ull) != 0ull);
    return active && !next_joins;
}

__global__ __launch_bounds__(kWave *kWaves) void k_scatter(PointSet ps, uint32_t n, const int32_t *count,
                                                            const float *__restrict__ dout, GridTable T,
                                                            uint32_t merge_levels, uint32_t level_mask,
                                                            float *__restrict__ grad_table) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    if (count != nullptr) { const uint32_t c = (uint32_t)max(*count, 0); n = c < n ? c : n; }
    const uint32_t F = T.n_levels * 2;
    const uint32_t s0 = blockIdx.x * kTile;
    if (s0 >= n) return;
    const uint32_t sub = lane & 3u, dx = sub >> 1, f = sub & 1u;

    for (uint32_t g = 0; g < kTile / 16; ++g) {  // 16 samples per wave instruction
        const uint32_t s = s0 + g * 16 + (lane >> 2);
        const bool valid = s < n;
        if (!__any(valid)) break;
        float base[2][3];
        load_bases(ps, s, valid, base);
        for (uint32_t p = 0; p < ps.P; ++p) {
            float q[3];
            point_of(ps, base, p, q);
            const float *drow = dout + ((size_t)s * ps.P + p) * F + f;
            for (uint32_t l = wave; l < T.n_levels; l += kWaves) {
                if (!((level_mask >> l) & 1u)) continue;
                const GridLevel L = T.level[l];
                const float d = valid ? drow[2 * l] : 0.f;
                // a row whose two feature gradients are both zero (padding, masked samples) is skipped
                const float d_other = __shfl_xor(d, 1, 64);  // the quad's other feature (unconditional: all lanes)
                const bool has = valid && (d != 0.f || d_other != 0.f);
                if (!__any(has)) continue;
                uint32_t cx, cy, cz;
                float fx, fy, fz;
                grid_cell(q[0], L.scale, cx, fx);
                grid_cell(q[1], L.scale, cy, fy);
                grid_cell(q[2], L.scale, cz, fz);
                const float wx = dx ? fx : 1.0f - fx;
                float v[4];
#pragma unroll
                for (uint32_t yz = 0; yz < 4; ++yz)  // same multiplication order as the forward: (wx * wy) * wz
                    v[yz] = ((wx * ((yz & 1u) ? fy : 1.0f - fy)) * ((yz >> 1) ? fz : 1.0f - fz)) * d;
                bool owner = has;
                if (l < merge_levels) owner = quad_merge_runs<4>(has, CellKey{cx, cy, cz}, v, lane);
                if (owner) {
                    float *lvl = grad_table + (size_t)L.offset * 2 + f;
#pragma unroll
                    for (uint32_t yz = 0; yz < 4; ++yz) {
                        const uint32_t e = grid_entry(L, cx + dx, cy + (yz & 1u), cz + (yz >> 1));
                        unsafeAtomicAdd(lvl + (size_t)e * 2, v[yz]);
                    }
                }
            }
        }
    }
}

PointSet make_points(const float *x, const float *x2, const float *offsets_host, uint32_t P0, uint32_t P, float bound,
                     int mode) {
    PointSet ps;
    ps.x = x; ps.x2 = x2; ps.P0 = P0; ps.P = P; ps.bound = bound; ps.mode = mode;
    for (uint32_t i = 0; i < kMaxPts * 3; ++i) ps.offs[i] = (offsets_host && i < P * 3) ? offsets_host[i] : 0.f;
    return ps;
}

int launch_scatter(const PointSet &ps, uint32_t n, const int32_t *count, const float *dout, const GridTable &T,
                   uint32_t merge_levels, float *grad_params, hipStream_t st) {
    static const int force = getenv("MI3D_SCATTER") ? atoi(getenv("MI3D_SCATTER")) : -1;  // profiling A/B only
    if (force >= 0) merge_levels = (uint32_t)force;
    if (merge_levels > T.n_levels) merge_levels = T.n_levels;
    const char *lm = getenv("MI3D_SCATTER_LMASK");  // profiling only: restrict the scatter to a subset of levels
    const uint32_t level_mask = lm ? (uint32_t)strtoul(lm, nullptr, 0) : 0xFFFFFFFFu;
    hipLaunchKernelGGL(k_scatter, dim3((n + kTile - 1) / kTile), dim3(kWave * kWaves), 0, st, ps, n, count, dout, T,
                       merge_levels, level_mask, grad_params);
    return (int)hipGetLastError();
}

// levels whose cells are longer than one marching step `step01` (in [0,1] units) try to merge neighbours
uint32_t default_merge_levels(const GridTable &T, float step01) {
    uint32_t m = 0;
    for (uint32_t l = 0; l < T.n_levels; ++l)
        if (1.0f / (float)T.level[l].res >= 1.05f * step01) m = l + 1;
    return m;
}

}  // namespace

extern "C" {

uint32_t mi3d_hashgrid_levels(uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                              uint32_t log2_hashmap_size, uint32_t *offsets_host, uint32_t *resolutions_host,
                              float *scales_host) {
    GridTable T;
    if (n_levels > MI3D_MAX_LEVELS) return 0;
    const uint32_t total = build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    for (uint32_t i = 0; i < n_levels; ++i) {
        if (offsets_host) offsets_host[i] = T.level[i].offset;
        if (resolutions_host) resolutions_host[i] = T.level[i].res;
        if (scales_host) scales_host[i] = T.level[i].scale;
    }
    if (offsets_host) offsets_host[n_levels] = total;
    return total;
}

int mi3d_hashgrid_forward(const float *x, uint32_t n, const float *params, uint32_t n_levels,
                          uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size, float *out,
                          void *stream) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const PointSet ps = make_points(x, nullptr, nullptr, 1, 1, 1.0f, 0);
    hipLaunchKernelGGL(k_grid_encode, dim3((n + kTile - 1) / kTile), dim3(kWave * kWaves), 0, as_stream(stream), ps, n,
                       (const int32_t *)nullptr, reinterpret_cast<const float2 *>(params), T, out);
    return (int)hipGetLastError();
}

int mi3d_hashgrid_backward(const float *x, uint32_t n, const float *dout, uint32_t n_levels, uint32_t base_resolution,
                           float per_level_scale, uint32_t log2_hashmap_size, float *grad_params, void *stream) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const PointSet ps = make_points(x, nullptr, nullptr, 1, 1, 1.0f, 0);
    // no knowledge of the sampling step here: merge wherever a 1/512 step would linger in a cell
    return launch_scatter(ps, n, nullptr, dout, T, T.n_levels, grad_params,
                          as_stream(stream));
}

int mi3d_grid_encode_points(const float *x, const float *x2, uint32_t n, const int32_t *count,
                            const float *offsets_host, uint32_t P0, uint32_t P, float bound, const float *params,
                            uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                            uint32_t log2_hashmap_size, float *out, void *stream) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || P == 0 || P > MI3D_MAX_POINTS || P0 > P ||
        (P0 < P && x2 == nullptr))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const PointSet ps = make_points(x, x2, offsets_host, P0, P, bound, 1);
    hipLaunchKernelGGL(k_grid_encode, dim3((n + kTile - 1) / kTile), dim3(kWave * kWaves), 0, as_stream(stream), ps, n,
                       count, reinterpret_cast<const float2 *>(params), T, out);
    return (int)hipGetLastError();
}

int mi3d_grid_scatter_points(const float *x, const float *x2, uint32_t n, const int32_t *count,
                             const float *offsets_host, uint32_t P0, uint32_t P, float bound, const float *dout,
                             uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                             uint32_t log2_hashmap_size, float step, float *grad_params, void *stream) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || P == 0 || P > MI3D_MAX_POINTS || P0 > P ||
        (P0 < P && x2 == nullptr))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const PointSet ps = make_points(x, x2, offsets_host, P0, P, bound, 1);
    const float step01 = step > 0.f ? step / (2.0f * bound) : 1.0f / 512.0f;
    return launch_scatter(ps, n, count, dout, T, default_merge_levels(T, step01), grad_params, as_stream(stream));
}

}  // extern "C"
