// Shared scalar math of the MI355X hot path: occupancy-grid DDA step, Morton codes, hash-grid
// indexing.  Everything here is `MI3D_HD` so tests/test_host_math.py can compile the very same
// functions for the host with g++ and compare them bit-for-bit with the oracle before a GPU is
// involved.  Semantics follow /root/reference/raymarching/src/raymarching.cu:19-81, :359-400 and
// tiny-cuda-nn's grid indexing (see oracle/hashgrid_ref.c for the provenance note).
//
// FMA policy: the product is compiled with -ffp-contract=off for this header's users in
// raymarching.hip; the fused multiply-adds that nvcc's default -fmad=true would form in the
// reference are spelled fmaf() so that voxel decisions are reproducible bit-for-bit.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MI3D_HD __host__ __device__ __forceinline__
#else
#define MI3D_HD static inline
#endif

namespace mi3d {

constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kRPi = 0.3183098861837907f;

MI3D_HD float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// ---- Morton codes (3 x 10 bit) -------------------------------------------------------------
MI3D_HD uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
MI3D_HD uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
MI3D_HD uint32_t compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// frexpf exponent of a non-negative finite float, by bit inspection (0 -> 0, like frexpf).
MI3D_HD int frexp_exponent(float v) {
    int e;
    (void)frexpf(v, &e);
    return e;
}

// ---- occupancy-grid marching ---------------------------------------------------------------
struct MarchRay {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float sx, sy, sz;  // 0.5 + 0.5*sign(d): 1 for non-negative directions (incl. +0), else 0
};
struct MarchGrid {
    const uint8_t *bits;
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Hm1;
    int C;
    uint32_t H;
    // uniform shortcuts that leave every result bit for bit (see march_step): H a power of two -> the reference's
    // double-precision products by 0.5 and H are exact scalings, done in float; one cascade -> level 0 always
    bool h_pow2, one_level;
    float half_H, mip_bound0, mip_rbound0;
};

MI3D_HD void march_ray_init(MarchRay &r, const float *o, const float *d) {
    r.ox = o[0]; r.oy = o[1]; r.oz = o[2];
    r.dx = d[0]; r.dy = d[1]; r.dz = d[2];
    r.rdx = 1.0f / r.dx; r.rdy = 1.0f / r.dy; r.rdz = 1.0f / r.dz;
    r.sx = 0.5f + 0.5f * copysignf(1.0f, r.dx);
    r.sy = 0.5f + 0.5f * copysignf(1.0f, r.dy);
    r.sz = 0.5f + 0.5f * copysignf(1.0f, r.dz);
}
MI3D_HD void march_grid_init(MarchGrid &g, const uint8_t *bits, float bound, float dt_gamma, uint32_t max_steps,
                             uint32_t C, uint32_t H) {
    g.bits = bits; g.bound = bound; g.dt_gamma = dt_gamma;
    g.dt_min = 2 * kSqrt3 / (float)max_steps;
    g.dt_max = 2 * kSqrt3 * (float)(1 << (C - 1)) / (float)H;
    g.rH = 1.0f / (float)H;
    g.H3 = (float)(H * H * H);
    g.Hf = (float)H;
    g.Hm1 = (float)(H - 1);
    g.C = (int)C; g.H = H;
    g.h_pow2 = H >= 2 && (H & (H - 1)) == 0;
    g.one_level = C == 1;
    g.half_H = 0.5f * (float)H;
    g.mip_bound0 = fminf(1.0f, bound);
    g.mip_rbound0 = 1.0f / g.mip_bound0;
}

// voxel coordinate along one axis: trunc(clamp(0.5*(p/mb + 1)*H, 0, H-1)).  The reference evaluates the
// outer product in double; the float fma result times 0.5 and H is re-rounded to float by clamp().
MI3D_HD int voxel_coord(float p, float mip_rbound, const MarchGrid &g) {
    const float u = fmaf(p, mip_rbound, 1.0f);
    // H = 2^k: 0.5 * u * H is u scaled by a power of two - exact in float and in double alike, so the float product IS
    // the double product rounded to float (u is O(1): no overflow, no underflow)
    if (g.h_pow2) return (int)clampf(u * g.half_H, 0.0f, g.Hm1);
    const double v = 0.5 * (double)u * (double)g.H;
    return (int)clampf((float)v, 0.0f, g.Hm1);
}

// One DDA iteration.  Occupied: returns true, (x,y,z) is the sample, dt its step, t advanced by dt.
// Empty: returns false and t has been advanced past the voxel's exit face.
MI3D_HD bool march_step(const MarchRay &r, const MarchGrid &g, float &t, float &x, float &y, float &z, float &dt) {
    x = clampf(fmaf(t, r.dx, r.ox), -g.bound, g.bound);
    y = clampf(fmaf(t, r.dy, r.oy), -g.bound, g.bound);
    z = clampf(fmaf(t, r.dz, r.oz), -g.bound, g.bound);
    dt = clampf(t * g.dt_gamma, g.dt_min, g.dt_max);

    // cascade level = max(level of the position, level of the step size), clamped to [0, C-1]; with one cascade that
    // is level 0 whatever the two exponents are
    int level = 0;
    float mip_bound = g.mip_bound0, mip_rbound = g.mip_rbound0;
    if (!g.one_level) {
        const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        int lp = frexp_exponent(mx);
        int ld = g.h_pow2 ? frexp_exponent(dt * g.half_H) : frexp_exponent((float)((double)(dt * g.Hf) * 0.5));
        lp = lp < 0 ? 0 : lp; ld = ld < 0 ? 0 : ld;
        level = lp > ld ? lp : ld;
        level = level > g.C - 1 ? g.C - 1 : level;
        mip_bound = fminf(scalbnf(1.0f, level), g.bound);
        mip_rbound = 1.0f / mip_bound;
    }
    const int nx = voxel_coord(x, mip_rbound, g);
    const int ny = voxel_coord(y, mip_rbound, g);
    const int nz = voxel_coord(z, mip_rbound, g);

    // the reference forms level*H^3 + morton in float before truncating (exact below 2^24)
    const uint32_t index = (uint32_t)((float)level * g.H3 + (float)morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const bool occ = (g.bits[index >> 3] >> (index & 7u)) & 1u;
    if (occ) {
        t += dt;
        return true;
    }
    // parametric distance to the exit face of this voxel, then step in dt units until past it
    const float tx = fmaf(((float)nx + r.sx) * g.rH * 2.0f - 1.0f, mip_bound, -x) * r.rdx;
    const float ty = fmaf(((float)ny + r.sy) * g.rH * 2.0f - 1.0f, mip_bound, -y) * r.rdy;
    const float tz = fmaf(((float)nz + r.sz) * g.rH * 2.0f - 1.0f, mip_bound, -z) * r.rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        t += clampf(t * g.dt_gamma, g.dt_min, g.dt_max);
    } while (t < tt);
    return false;
}

// first sample parameter: near + clamp(near*dt_gamma, dt_min, dt_max) * noise   (one fma)
MI3D_HD float march_t0(float near, float noise, const MarchGrid &g) {
    return fmaf(clampf(near * g.dt_gamma, g.dt_min, g.dt_max), noise, near);
}

// ---- multiresolution hash grid -------------------------------------------------------------
constexpr uint32_t kPrimeY = 2654435761u, kPrimeZ = 805459861u;

struct GridLevel {
    float scale;        // exp2(level*log2(per_level_scale))*base - 1
    uint32_t res;       // ceil(scale) + 1
    uint32_t offset;    // first entry of this level in the table
    uint32_t size;      // entries in this level
    uint32_t hashed;    // 1: coherent-prime hash, 0: dense (strided) indexing
    uint32_t dims;      // how many dims the dense stride loop covers (3 unless the stride overflowed size)
};

MI3D_HD uint32_t grid_entry(const GridLevel &L, uint32_t px, uint32_t py, uint32_t pz) {
    uint32_t index;
    if (L.hashed) {
        index = px ^ (py * kPrimeY) ^ (pz * kPrimeZ);
        // hashed levels are 2^k entries in every configuration the reference builds; keep the general path
        index = ((L.size & (L.size - 1u)) == 0u) ? (index & (L.size - 1u)) : (index % L.size);
    } else {
        index = px;
        if (L.dims > 1) index += py * L.res;
        if (L.dims > 2) index += pz * L.res * L.res;
        if (index >= L.size) {  // only the +1 corner at the upper boundary wraps (or out-of-range input)
            index -= L.size;
            if (index >= L.size) index %= L.size;
        }
    }
    return index;
}

// fractional position inside the level's lattice: pos = fma(scale, x, 0.5); cell = floor(pos); w = pos - cell
MI3D_HD void grid_cell(float x, float scale, uint32_t &cell, float &w) {
    const float p = fmaf(scale, x, 0.5f);
    const float fl = floorf(p);
    cell = (uint32_t)(int)fl;
    w = p - fl;
}

}  // namespace mi3d
