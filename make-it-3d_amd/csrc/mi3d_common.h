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

// ---- the scatter's 12-byte x-pair record (binary16 gradient planes; hashgrid.hip k_bin_emit / k_bin_reduce) ---------------
// The two x-neighbours of a (y, z) corner pair differ in their entry's low bits only, because x enters the hash with prime 1:
// e(x + 1) = e(x) ^ (2^t - 1) (masked to the level), t = 1 + trailing ones of cx (a +1 carry flips exactly those bits); a
// dense level's pair is e, e + 1 (t = 0).  One record carries both contributions - (1 - fx) w (a, b) and fx w (a, b), w = w_y w_z:
//   word 0 = entry & 8191 | t << 13 | (w_q >> 16) << 18 | (fx_q >> 16) << 25      word 1 = the raw binary16 pair (a, b)
//   word 2 = (w_q & 0xFFFF) | (fx_q & 0xFFFF) << 16          w_q = round(w 2^23), fx_q = round(fx 2^23), clamped below 2^23
// |w - w_q 2^-23| <= 2^-24 (2^-23 at w = 1, the clamp): the last bit of an fp32 weight near 1.
struct Row12 { uint32_t w0, w1, w2; };
constexpr float kFix23 = 8388608.0f, kUnfix23 = 1.0f / 8388608.0f;
MI3D_HD uint32_t fix23(float v) {   // v in [0, 1]
    const uint32_t q = (uint32_t)(v * kFix23 + 0.5f);
    return q < 8388607u ? q : 8388607u;
}
MI3D_HD uint32_t pair_flip_t(uint32_t cx) {   // hashed levels: how many low bits of the entry a +1 in x flips
    uint32_t t = 1;
    while (cx & 1u) { cx >>= 1; ++t; }
    return t;
}
MI3D_HD Row12 pack_row12(uint32_t e_local, uint32_t t, float w, float fx, uint32_t raw_pair) {
    const uint32_t wq = fix23(w), fq = fix23(fx);
    return Row12{e_local | (t << 13) | ((wq >> 16) << 18) | ((fq >> 16) << 25), raw_pair, (wq & 0xFFFFu) | ((fq & 0xFFFFu) << 16)};
}
MI3D_HD void unpack_row12_fields(uint32_t w0, uint32_t w2, uint32_t &e_local, uint32_t &t, float &w, float &fx) {
    e_local = w0 & 8191u;
    t = (w0 >> 13) & 31u;
    w = (float)((((w0 >> 18) & 127u) << 16) | (w2 & 0xFFFFu)) * kUnfix23;
    fx = (float)(((w0 >> 25) << 16) | (w2 >> 16)) * kUnfix23;
}

// ---- a 32 x 32 binary16 tile turned round through LDS (field.hip's MLP backward; csrc/lds_transpose.h holds the device side) ----
// Byte offsets of a lane inside its wave's tile image M[sample][feature]: a row = 32 features = 64 bytes, rows kTrRowBytes
// apart.  The lane's 16 values are four 8-byte chunks; value order "D" (an MFMA output tile: value q of lane half h = index
// (q & 3) + 8 (q >> 2) + 4 h) puts chunk c at features 8 c + 4 h, order "X" (rows as loaded: 16 h + q) at 16 h + 4 c.
// The transposing read (ds_read_b64_tr_b16) hands lane t of a 16-lane group, value j, what lane 4 j + (t >> 2) of the group
// loaded as its value t & 3; with the read offsets below (+ 8 kTrRowBytes c for chunk c) lane = feature gets value q =
// sample rowmap(q, h').  tests/test_host_math.py plays this through on the host; tools/tr_probe.hip on the chip.
constexpr int kTrRowBytes = 72;
MI3D_HD uint32_t tr_write_offset_d(int lane) { return (uint32_t)(kTrRowBytes * (lane & 31) + 8 * (lane >> 5)); }
MI3D_HD uint32_t tr_write_offset_x(int lane) { return (uint32_t)(kTrRowBytes * (lane & 31) + 32 * (lane >> 5)); }
MI3D_HD uint32_t tr_read_offset(int lane) {
    const int t = lane & 15, g = lane >> 4;
    return (uint32_t)(kTrRowBytes * (4 * (lane >> 5) + (t >> 2)) + 8 * (4 * (g & 1) + (t & 3)));
}

}  // namespace mi3d
