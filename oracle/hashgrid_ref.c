/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the multiresolution hash encoding the reference takes
 * from tiny-cuda-nn:  `tcnn.Encoding(n_input_dims=3, {"otype":"HashGrid", ...},
 * dtype=torch.float32)` at /root/reference/nerf/network_tcnn.py:54-65, called
 * at :107.
 *
 * PARITY UNPINNED: tiny-cuda-nn is not vendored in /root/reference and is
 * installed from git master with no version pin (README.md:43), it is not
 * installed in this image and there is no network.  What follows restates the
 * published algorithm of NVlabs/tiny-cuda-nn `include/tiny-cuda-nn/encodings/
 * grid.h` + `common_device.h` (grid_scale / grid_resolution / pos_fract /
 * grid_index / coherent-prime hash, linear interpolation, fp32 parameters,
 * float atomicAdd backward) from memory.  It is anchored only on the
 * reference's call site (input in [0,1]^3, 16 levels x 2 features, output
 * feature index = level*2 + f, one flat fp32 `params` tensor of 12 196 240
 * floats for the default config - checked in tests/test_oracle_cpu.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))
#define MAX_LEVELS 32

/* tcnn grid.h: grid_scale() -- exp2f(level * log2(per_level_scale)) * base - 1 */
static inline float grid_scale(uint32_t level, float log2_per_level_scale, uint32_t base_resolution) {
    return exp2f((float)level * log2_per_level_scale) * (float)base_resolution - 1.0f;
}
/* tcnn grid.h: grid_resolution() */
static inline uint32_t grid_resolution(float scale) { return (uint32_t)ceilf(scale) + 1; }

/* Level table exactly as GridEncodingTemplated's constructor builds it:
 * params_in_level = min(next_multiple(res^3, 8), 2^log2_hashmap_size); offsets = prefix sum.
 * offsets has n_levels+1 entries (in grid entries, not floats).  Returns total entries. */
ORACLE_API uint32_t ref_hashgrid_levels(uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                                        uint32_t log2_hashmap_size, uint32_t *offsets, uint32_t *resolutions,
                                        float *scales) {
    const float l2 = log2f(per_level_scale);
    uint32_t offset = 0;
    for (uint32_t i = 0; i < n_levels; i++) {
        const float scale = grid_scale(i, l2, base_resolution);
        const uint32_t res = grid_resolution(scale);
        const uint32_t max_params = 0xFFFFFFFFu / 2;
        uint32_t params = (powf((float)res, 3.0f) > (float)max_params) ? max_params : res * res * res;
        params = (params + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (params > cap) params = cap;
        offsets[i] = offset;
        if (resolutions) resolutions[i] = res;
        if (scales) scales[i] = scale;
        offset += params;
    }
    offsets[n_levels] = offset;
    return offset;
}

/* tcnn common_device.h: grid_index<3, CoherentPrime>() */
static inline uint32_t grid_index(uint32_t hashmap_size, uint32_t res, uint32_t px, uint32_t py, uint32_t pz) {
    const uint32_t p[3] = {px, py, pz};
    uint32_t stride = 1, index = 0;
    for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
        index += p[dim] * stride;
        stride *= res;
    }
    if (hashmap_size < stride) index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
    return index % hashmap_size;
}

typedef struct { uint32_t idx[8]; float w[8]; } corners_t;

/* tcnn pos_fract() + the 8-corner loop of kernel_grid (linear interpolation) */
static inline void corners(const float *x, float scale, uint32_t res, uint32_t hashmap_size, corners_t *c) {
    float pos[3]; uint32_t g[3];
    for (int d = 0; d < 3; d++) {
        const float p = fmaf(scale, x[d], 0.5f);
        const float fl = floorf(p);
        g[d] = (uint32_t)(int)fl;
        pos[d] = p - fl;
    }
    for (uint32_t k = 0; k < 8; k++) {
        float w = 1; uint32_t q[3];
        for (uint32_t d = 0; d < 3; d++) {
            if ((k & (1u << d)) == 0) { w *= 1 - pos[d]; q[d] = g[d]; }
            else                      { w *= pos[d];     q[d] = g[d] + 1; }
        }
        c->idx[k] = grid_index(hashmap_size, res, q[0], q[1], q[2]);
        c->w[k] = w;
    }
}

/* x: [n,3] in [0,1]; params: [total*F] fp32; out: [n, n_levels*F], feature = level*F + f. F = 2. */
ORACLE_API void ref_hashgrid_forward(const float *x, uint32_t n, const float *params, uint32_t n_levels,
                                     uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size,
                                     float *out) {
    uint32_t offsets[MAX_LEVELS + 1], res[MAX_LEVELS]; float scales[MAX_LEVELS];
    ref_hashgrid_levels(n_levels, base_resolution, per_level_scale, log2_hashmap_size, offsets, res, scales);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        for (uint32_t l = 0; l < n_levels; l++) {
            const uint32_t hs = offsets[l + 1] - offsets[l];
            const float *grid = params + (size_t)offsets[l] * 2;
            corners_t c; corners(x + i * 3, scales[l], res[l], hs, &c);
            float r0 = 0, r1 = 0;
            for (int k = 0; k < 8; k++) { /* result += weight * data, in corner order 0..7 */
                r0 += c.w[k] * grid[(size_t)c.idx[k] * 2 + 0];
                r1 += c.w[k] * grid[(size_t)c.idx[k] * 2 + 1];
            }
            out[(size_t)i * n_levels * 2 + l * 2 + 0] = r0;
            out[(size_t)i * n_levels * 2 + l * 2 + 1] = r1;
        }
    }
}

/* tcnn kernel_grid_backward: grad_params[off_l + idx][f] += w * dout[l*2+f]  (float atomics there;
 * sequential here, so the sum order is sample-major - compare with a tolerance). grad_params must be zeroed. */
ORACLE_API void ref_hashgrid_backward(const float *x, uint32_t n, const float *dout, uint32_t n_levels,
                                      uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size,
                                      float *grad_params) {
    uint32_t offsets[MAX_LEVELS + 1], res[MAX_LEVELS]; float scales[MAX_LEVELS];
    ref_hashgrid_levels(n_levels, base_resolution, per_level_scale, log2_hashmap_size, offsets, res, scales);
    /* levels are independent -> parallelise over levels without atomics */
#pragma omp parallel for schedule(dynamic, 1)
    for (int l = 0; l < (int)n_levels; l++) {
        const uint32_t hs = offsets[l + 1] - offsets[l];
        float *g = grad_params + (size_t)offsets[l] * 2;
        for (uint32_t i = 0; i < n; i++) {
            corners_t c; corners(x + (size_t)i * 3, scales[l], res[l], hs, &c);
            const float d0 = dout[(size_t)i * n_levels * 2 + l * 2 + 0], d1 = dout[(size_t)i * n_levels * 2 + l * 2 + 1];
            for (int k = 0; k < 8; k++) {
                g[(size_t)c.idx[k] * 2 + 0] += c.w[k] * d0;
                g[(size_t)c.idx[k] * 2 + 1] += c.w[k] * d1;
            }
        }
    }
}

/* Integer side only (bit-exact target for the HIP kernels): the 8 table indices
 * (relative to the level's base) per (sample, level). idx_out: [n, n_levels, 8] */
ORACLE_API void ref_hashgrid_indices(const float *x, uint32_t n, uint32_t n_levels, uint32_t base_resolution,
                                     float per_level_scale, uint32_t log2_hashmap_size, uint32_t *idx_out,
                                     float *w_out) {
    uint32_t offsets[MAX_LEVELS + 1], res[MAX_LEVELS]; float scales[MAX_LEVELS];
    ref_hashgrid_levels(n_levels, base_resolution, per_level_scale, log2_hashmap_size, offsets, res, scales);
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t l = 0; l < n_levels; l++) {
            corners_t c; corners(x + (size_t)i * 3, scales[l], res[l], offsets[l + 1] - offsets[l], &c);
            memcpy(idx_out + ((size_t)i * n_levels + l) * 8, c.idx, sizeof c.idx);
            if (w_out) memcpy(w_out + ((size_t)i * n_levels + l) * 8, c.w, sizeof c.w);
        }
}
