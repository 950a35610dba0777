// Test helper (NOT product code): compiles make-it-3d_amd/csrc/mi3d_common.h - the exact scalar
// math the HIP kernels use - for the HOST, so tests can compare it bit-for-bit with the oracle
// without a GPU.  Built by tests/test_host_math.py with g++ -ffp-contract=off.
#include <cstdint>
#include <cstring>
#include "mi3d_common.h"
using namespace mi3d;

extern "C" {

// sequential emulation of k_march_train (slabs in ray order)
void hm_march_train(const float* rays_o, const float* rays_d, const uint8_t* bits, float bound, float dt_gamma,
                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                    const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                    const float* noises) {
    MarchGrid g; march_grid_init(g, bits, bound, dt_gamma, max_steps, C, H);
    uint32_t total = 0;
    for (uint32_t n = 0; n < N; n++) {
        MarchRay r; march_ray_init(r, rays_o + n * 3, rays_d + n * 3);
        const float far = fars[n], t0 = march_t0(nears[n], noises[n], g);
        uint32_t count = 0; float t = t0, x, y, z, dt;
        while (t < far && count < max_steps) if (march_step(r, g, t, x, y, z, dt)) ++count;
        const uint32_t offset = total; total += count;
        rays[n * 3] = n; rays[n * 3 + 1] = offset; rays[n * 3 + 2] = count;
        if (count == 0 || offset + count > M) continue;
        float *px = xyzs + (size_t)offset * 3, *pd = dirs + (size_t)offset * 3, *pl = deltas + (size_t)offset * 2;
        t = t0; float last_t = t0; uint32_t step = 0;
        while (t < far && step < count) {
            if (march_step(r, g, t, x, y, z, dt)) {
                px[0] = x; px[1] = y; px[2] = z; pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                pl[0] = dt; pl[1] = t - last_t; last_t = t; px += 3; pd += 3; pl += 2; ++step;
            }
        }
    }
    counter[0] += total; counter[1] += N;
}

void hm_morton(const int32_t* c, uint32_t N, int32_t* out) {
    for (uint32_t n = 0; n < N; n++) out[n] = (int32_t)morton3d(c[n*3], c[n*3+1], c[n*3+2]);
}
void hm_morton_invert(const int32_t* in, uint32_t N, int32_t* c) {
    for (uint32_t n = 0; n < N; n++) { int32_t v = in[n];
        c[n*3] = compact_bits((uint32_t)(v >> 0)); c[n*3+1] = compact_bits((uint32_t)(v >> 1)); c[n*3+2] = compact_bits((uint32_t)(v >> 2)); }
}

// hash-grid corner indices / weights for one level, given the host-built level record
void hm_grid_corners(const float* x, uint32_t n, float scale, uint32_t res, uint32_t offset, uint32_t size,
                     uint32_t hashed, uint32_t dims, uint32_t* idx, float* w) {
    GridLevel L{scale, res, offset, size, hashed, dims};
    for (uint32_t i = 0; i < n; i++) {
        uint32_t c[3]; float f[3];
        for (int d = 0; d < 3; d++) grid_cell(x[i*3+d], scale, c[d], f[d]);
        for (uint32_t k = 0; k < 8; k++) {
            float wk = 1.f; uint32_t q[3];
            for (uint32_t d = 0; d < 3; d++) {
                if (k & (1u << d)) { wk *= f[d]; q[d] = c[d] + 1; } else { wk *= 1 - f[d]; q[d] = c[d]; }
            }
            idx[i*8+k] = grid_entry(L, q[0], q[1], q[2]); w[i*8+k] = wk;
        }
    }
}

// the scatter's 12-byte pair record: pack, unpack, and the x-pair rule on a hashed level of `size` (a power of two) entries
void hm_row12_roundtrip(const uint32_t* e_local, const uint32_t* t, const float* w, const float* fx, const uint32_t* raw,
                        uint32_t n, uint32_t* words, uint32_t* e_out, uint32_t* t_out, float* w_out, float* fx_out) {
    for (uint32_t i = 0; i < n; i++) {
        const Row12 r = pack_row12(e_local[i], t[i], w[i], fx[i], raw[i]);
        words[i*3] = r.w0; words[i*3+1] = r.w1; words[i*3+2] = r.w2;
        unpack_row12_fields(r.w0, r.w2, e_out[i], t_out[i], w_out[i], fx_out[i]);
    }
}
void hm_pair_rule(const uint32_t* cell, uint32_t n, uint32_t size, uint32_t* e0, uint32_t* e1, uint32_t* e1_from_rule) {
    GridLevel L{1.0f, 0u, 0u, size, 1u, 3u};
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t cx = cell[i*3], cy = cell[i*3+1], cz = cell[i*3+2];
        e0[i] = grid_entry(L, cx, cy, cz);
        e1[i] = grid_entry(L, cx + 1, cy, cz);
        const uint32_t t = pair_flip_t(cx);
        e1_from_rule[i] = e0[i] ^ (((t >= 32 ? 0xFFFFFFFFu : (1u << t) - 1u)) & (size - 1u));
    }
}
// the MLP backward's tile transposition through LDS: the three byte offsets of a lane (csrc/mi3d_common.h)
void hm_tr_offsets(uint32_t* wr_d, uint32_t* wr_x, uint32_t* rd, int32_t* row_bytes) {
    for (int lane = 0; lane < 64; lane++) {
        wr_d[lane] = tr_write_offset_d(lane); wr_x[lane] = tr_write_offset_x(lane); rd[lane] = tr_read_offset(lane);
    }
    *row_bytes = kTrRowBytes;
}
}
