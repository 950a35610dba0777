// Multiresolution hash-grid building blocks shared by hashgrid.hip (stand-alone tcnn.Encoding
// replacement) and field.hip (fused field).  Algorithm provenance: oracle/hashgrid_ref.c.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mi3d.h"
#include "mi3d_common.h"

namespace mi3d {

struct GridTable {
    GridLevel level[MI3D_MAX_LEVELS];
    uint32_t n_levels;
    uint32_t n_entries;
};

// Host: the level table exactly as tiny-cuda-nn's GridEncodingTemplated constructor lays it out.
inline uint32_t build_grid_table(GridTable &T, uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                                 uint32_t log2_hashmap_size) {
    const float l2 = log2f(per_level_scale);
    uint32_t offset = 0;
    T.n_levels = n_levels;
    for (uint32_t i = 0; i < n_levels && i < MI3D_MAX_LEVELS; ++i) {
        GridLevel &L = T.level[i];
        L.scale = exp2f((float)i * l2) * (float)base_resolution - 1.0f;
        L.res = (uint32_t)ceilf(L.scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2;
        uint32_t size = (powf((float)L.res, 3.0f) > (float)max_params) ? max_params : L.res * L.res * L.res;
        size = (size + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (size > cap) size = cap;
        L.size = size;
        L.offset = offset;
        // which dims the dense stride loop covers before the stride exceeds the level size
        uint32_t stride = 1, dims = 0;
        for (; dims < 3 && stride <= size; ++dims) stride *= L.res;
        L.dims = dims;
        L.hashed = size < stride ? 1u : 0u;
        offset += size;
    }
    T.n_entries = offset;
    return offset;
}

// The 8 lattice corners of one point at one level: entry indices (relative to the level) and weights.
// Corner k: bit0 -> +x, bit1 -> +y, bit2 -> +z (the order tcnn accumulates in).
struct Corners {
    uint32_t idx[8];
    float w[8];
};

__device__ __forceinline__ void grid_corners(const GridLevel &L, float x, float y, float z, Corners &c) {
    uint32_t cx, cy, cz;
    float fx, fy, fz;
    grid_cell(x, L.scale, cx, fx);
    grid_cell(y, L.scale, cy, fy);
    grid_cell(z, L.scale, cz, fz);
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    // weights in tcnn's multiplication order: ((1 * wx) * wy) * wz
    const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;
    c.w[0] = w00 * gz; c.w[1] = w10 * gz; c.w[2] = w01 * gz; c.w[3] = w11 * gz;
    c.w[4] = w00 * fz; c.w[5] = w10 * fz; c.w[6] = w01 * fz; c.w[7] = w11 * fz;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k)
        c.idx[k] = grid_entry(L, cx + (k & 1u), cy + ((k >> 1) & 1u), cz + ((k >> 2) & 1u));
}

// ---------------------------------------------------------------------------------------------
// Run merging in the scatter kernels.
//
// Neighbouring lanes hold neighbouring samples of one ray, so at the coarse and middle levels long runs of lanes
// fall into the SAME lattice cell and would send their corner values to the same addresses (measured: 206 ms for one
// 10.9 M-sample backward without merging).  The kernels in hashgrid.hip group lanes into runs of equal cell key and
// sum each run before it leaves the wave - by a segmented scan across quads (k_scatter) or through an LDS slab
// (k_scatter_runs, k_bin_emit).  Any partition into equal-key runs is valid, so inactive lanes simply break runs.
struct CellKey {
    uint32_t a, b, c;  // lattice cell (cx, cy, cz)
};
__device__ __forceinline__ bool operator==(const CellKey &p, const CellKey &q) {
    return p.a == q.a && p.b == q.b && p.c == q.c;
}

}  // namespace mi3d
