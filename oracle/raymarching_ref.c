/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's ray-marching kernels,
 *   /root/reference/raymarching/src/raymarching.cu
 * one host function per CUDA kernel, same arithmetic, same operation order.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (make-it-3d_amd/) never does.
 *
 * PINNED: tests/test_reference_kernels_gpu.py runs the reference's own kernels
 * (the .cu above built for gfx950 by oracle/build_ref.py into oracle/_ref/) next
 * to this restatement on an MI355X - integer kernels and near/far bit-exact,
 * training march per-ray counts equal on every ray and sampled rays bit-exact,
 * composite forward/backward and the inference loop within 1e-5.
 *
 * Floating-point policy (DESIGN.md "FMA policy"): nvcc's default -fmad=true
 * contracts a*b+c into one fused multiply-add.  Every site where that
 * contraction changes an integer decision (sample position -> voxel index,
 * DDA skip distance) is written here as an explicit fmaf(); everything else is
 * compiled with -ffp-contract=off, so this file has exactly one meaning.
 * The HIP kernels use the same explicit fmaf() sites, which is what lets the
 * parity tests demand bit-equality on voxel indices / step counts.
 *
 * Pinning: validated against (a) golden vectors produced by the reference's
 * own kernels (hipified build of raymarching.cu run on MI355X, see
 * oracle/build_ref.py + tests/golden/make_golden_ref_gpu.py) and (b) the
 * reference's pure-PyTorch composite formula nerf/renderer.py:415-452
 * (tests/golden/make_golden_py.py).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* raymarching.cu:19 */
static const float SQRT3 = 1.7320508075688772f;
/* raymarching.cu:22 */
static const float RPI = 0.3183098861837907f;

/* raymarching.cu:30-32 */
static inline float signf_(float x) { return copysignf(1.0f, x); }
/* raymarching.cu:34-36 */
static inline float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* raymarching.cu:42-47 */
static inline int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:49-54  (dt*H is float, *0.5 promotes to double, narrowed back) */
static inline int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:56-63 */
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:65-71 */
static inline uint32_t morton3D_(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* raymarching.cu:73-81 */
static inline uint32_t morton3D_invert_(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

/* ---------------------------------------------------------------- utils */

/* raymarching.cu:91-145 */
ORACLE_API void ref_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                                       uint32_t N, float min_near, float *nears, float *fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float *o = rays_o + n * 3, *d = rays_d + n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];

        float near = (aabb[0] - ox) * rdx;
        float far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }

        float near_y = (aabb[1] - oy) * rdy;
        float far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }

        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;

        float near_z = (aabb[2] - oz) * rdz;
        float far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }

        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;

        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* raymarching.cu:162-198 (float results, tolerance-compared) */
ORACLE_API void ref_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                                 float *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const float *o = rays_o + n * 3, *d = rays_d + n * 3;
        const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float C = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[n * 2 + 0] = 2 * theta * RPI - 1;
        coords[n * 2 + 1] = phi * RPI;
    }
}

/* raymarching.cu:214-226 */
ORACLE_API void ref_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3D_((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1],
                                        (uint32_t)coords[n * 3 + 2]);
}

/* raymarching.cu:237-254 (note: `ind >> k` is an arithmetic shift of an int) */
ORACLE_API void ref_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n];
        coords[n * 3 + 0] = (int32_t)morton3D_invert_((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)morton3D_invert_((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)morton3D_invert_((uint32_t)(ind >> 2));
    }
}

/* raymarching.cu:267-289 */
ORACLE_API void ref_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        const float *g = grid + (size_t)n * 8;
        uint8_t bits = 0;
        for (uint8_t i = 0; i < 8; i++) bits |= (g[i] > density_thresh) ? ((uint8_t)1 << i) : 0;
        bitfield[n] = bits;
    }
}

/* ------------------------------------------------------------- training */

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, bound, dt_gamma, dt_min, dt_max, far;
    uint32_t C, H;
    const uint8_t *grid;
} march_ctx;

/* One DDA iteration shared by the count pass and the write pass
 * (raymarching.cu:359-400 == :427-479).  Returns 1 if the voxel is occupied
 * (x,y,z,dt valid, *t advanced by dt), 0 if the ray skipped empty space. */
static inline int march_step(const march_ctx *c, float *t, float *x, float *y, float *z, float *dt_out) {
    /* current point: nvcc contracts o + t*d */
    *x = clampf_(fmaf(*t, c->dx, c->ox), -c->bound, c->bound);
    *y = clampf_(fmaf(*t, c->dy, c->oy), -c->bound, c->bound);
    *z = clampf_(fmaf(*t, c->dz, c->oz), -c->bound, c->bound);

    const float dt = clampf_(*t * c->dt_gamma, c->dt_min, c->dt_max);

    int level = mip_from_pos(*x, *y, *z, (float)c->C);
    const int l2 = mip_from_dt(dt, (float)c->H, (float)c->C);
    if (l2 > level) level = l2;

    const float mip_bound = fminf(scalbnf(1.0f, level), c->bound);
    const float mip_rbound = 1 / mip_bound;

    /* 0.5 * (x*rb + 1) * H : float fma, then double product, narrowed to float by clamp() */
    const int nx = (int)clampf_((float)(0.5 * (double)fmaf(*x, mip_rbound, 1.0f) * (double)c->H), 0.0f, (float)(c->H - 1));
    const int ny = (int)clampf_((float)(0.5 * (double)fmaf(*y, mip_rbound, 1.0f) * (double)c->H), 0.0f, (float)(c->H - 1));
    const int nz = (int)clampf_((float)(0.5 * (double)fmaf(*z, mip_rbound, 1.0f) * (double)c->H), 0.0f, (float)(c->H - 1));

    /* level * H3 + morton evaluated in float, then truncated (raymarching.cu:378) */
    const uint32_t index = (uint32_t)((float)level * c->H3 + (float)morton3D_((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = c->grid[index / 8] & (1 << (index % 8));

    if (occ) {
        *dt_out = dt;
        *t += dt;
        return 1;
    }
    /* distance to the next voxel face (raymarching.cu:390-398) */
    const float tx = (fmaf(((float)nx + 0.5f + 0.5f * signf_(c->dx)) * c->rH * 2 - 1, mip_bound, -*x)) * c->rdx;
    const float ty = (fmaf(((float)ny + 0.5f + 0.5f * signf_(c->dy)) * c->rH * 2 - 1, mip_bound, -*y)) * c->rdy;
    const float tz = (fmaf(((float)nz + 0.5f + 0.5f * signf_(c->dz)) * c->rH * 2 - 1, mip_bound, -*z)) * c->rdz;
    const float tt = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        *t += clampf_(*t * c->dt_gamma, c->dt_min, c->dt_max);
    } while (*t < tt);
    return 0;
}

static inline void march_ctx_init(march_ctx *c, const float *o, const float *d, const uint8_t *grid, float bound,
                                  float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, float far) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1 / c->dx; c->rdy = 1 / c->dy; c->rdz = 1 / c->dz;
    c->rH = 1 / (float)H;
    c->H3 = (float)(H * H * H);
    c->bound = bound; c->dt_gamma = dt_gamma;
    c->dt_min = 2 * SQRT3 / max_steps;                 /* raymarching.cu:345 */
    c->dt_max = 2 * SQRT3 * (1 << (C - 1)) / H;        /* raymarching.cu:346 */
    c->far = far; c->C = C; c->H = H; c->grid = grid;
}

/* raymarching.cu:311-480.  The two atomicAdd()s (:405-406) are replaced by a
 * prefix sum in ray order, which is one of the legal outcomes of the atomics
 * (the one a single sequential thread would produce).  counter[0] += total
 * samples, counter[1] += N, exactly as the atomics leave them. */
ORACLE_API void ref_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                                     float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                     uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs,
                                     float *deltas, int32_t *rays, int32_t *counter, const float *noises) {
    /* pass 1: count (parallel) */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        march_ctx c;
        march_ctx_init(&c, rays_o + n * 3, rays_d + n * 3, grid, bound, dt_gamma, max_steps, C, H, fars[n]);
        float t0 = nears[n];
        t0 = fmaf(clampf_(t0 * dt_gamma, c.dt_min, c.dt_max), noises[n], t0); /* :351, contracted */
        float t = t0, x, y, z, dt;
        uint32_t num_steps = 0;
        while (t < c.far && num_steps < max_steps)
            if (march_step(&c, &t, &x, &y, &z, &dt)) num_steps++;
        rays[n * 3 + 2] = (int32_t)num_steps;
    }
    /* slab allocation in ray order */
    uint32_t point_index = (uint32_t)counter[0], ray_index = (uint32_t)counter[1];
    /* rays rows are written at ray_index..ray_index+N-1; tests always start from a zeroed counter */
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ns = rays[n * 3 + 2];
        rays[(ray_index + n) * 3 + 0] = (int32_t)n;
        rays[(ray_index + n) * 3 + 1] = (int32_t)point_index;
        rays[(ray_index + n) * 3 + 2] = ns;
        point_index += (uint32_t)ns;
    }
    counter[0] = (int32_t)point_index;
    counter[1] = (int32_t)(ray_index + N);

    /* pass 2: write (parallel) */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t off = (uint32_t)rays[(ray_index + n) * 3 + 1];
        const uint32_t num_steps = (uint32_t)rays[(ray_index + n) * 3 + 2];
        if (num_steps == 0) continue;
        if (off + num_steps > M) continue;
        march_ctx c;
        march_ctx_init(&c, rays_o + n * 3, rays_d + n * 3, grid, bound, dt_gamma, max_steps, C, H, fars[n]);
        float t0 = nears[n];
        t0 = fmaf(clampf_(t0 * dt_gamma, c.dt_min, c.dt_max), noises[n], t0);
        float t = t0, last_t = t0, x, y, z, dt;
        uint32_t step = 0;
        float *px = xyzs + (size_t)off * 3, *pd = dirs + (size_t)off * 3, *pl = deltas + (size_t)off * 2;
        while (t < c.far && step < num_steps) {
            if (march_step(&c, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            }
        }
    }
}

/* raymarching.cu:500-577 (SDF == 0) and :708-783 (SDF == 1: alpha = sigma, ws = 1 - T) */
static void composite_train_forward(int sdf, const float *sigmas, const float *rgbs, const float *deltas,
                                    const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                    float *weights_sum, float *depth, float *image) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                       num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = sdf ? s[0] : 1.0f - expf(-s[0] * dl[0]);
            const float weight = alpha * T;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            t += dl[1];
            d += weight * t;
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            s++; c += 3; dl += 2;
        }
        weights_sum[index] = sdf ? 1.0f - T : ws; /* :775 vs :572 */
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

ORACLE_API void ref_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                                 const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                                 float *weights_sum, float *depth, float *image) {
    composite_train_forward(0, sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
}
ORACLE_API void ref_composite_sdf_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                                     const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                                     float *weights_sum, float *depth, float *image) {
    composite_train_forward(1, sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
}

/* raymarching.cu:601-682 (SDF == 0) and :807-887 (SDF == 1) */
static void composite_train_backward(int sdf, const float *grad_weights_sum, const float *grad_image,
                                     const float *sigmas, const float *rgbs, const float *deltas,
                                     const int32_t *rays, const float *weights_sum, const float *image, uint32_t M,
                                     uint32_t N, float T_thresh, float *grad_sigmas, float *grad_rgbs) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                       num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gws = grad_weights_sum[index];
        const float *gi = grad_image + (size_t)index * 3;
        const float ws_final = weights_sum[index];
        const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
        const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
        float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * 3;
        float T = 1.0f, r = 0, g = 0, b = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = sdf ? s[0] : 1.0f - expf(-s[0] * dl[0]);
            const float weight = alpha * T;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
            /* identical expression in both variants (:662-667 == :867-872) */
            gs[0] = dl[0] * (gi[0] * (T * c[0] - (r_final - r)) + gi[1] * (T * c[1] - (g_final - g)) +
                             gi[2] * (T * c[2] - (b_final - b)) + gws * (1 - ws_final));
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; gs++; gc += 3;
        }
    }
}

ORACLE_API void ref_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                                  const float *sigmas, const float *rgbs, const float *deltas,
                                                  const int32_t *rays, const float *weights_sum, const float *image,
                                                  uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                                  float *grad_rgbs) {
    composite_train_backward(0, grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                             T_thresh, grad_sigmas, grad_rgbs);
}
ORACLE_API void ref_composite_sdf_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                                      const float *sigmas, const float *rgbs, const float *deltas,
                                                      const int32_t *rays, const float *weights_sum,
                                                      const float *image, uint32_t M, uint32_t N, float T_thresh,
                                                      float *grad_sigmas, float *grad_rgbs) {
    composite_train_backward(1, grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                             T_thresh, grad_sigmas, grad_rgbs);
}

/* ------------------------------------------------------------ inference */

/* raymarching.cu:906-1011.  Single pass from rays_t[index]; rows past the
 * ray's end keep whatever the caller put there (zeros: raymarching.py:402-404). */
ORACLE_API void ref_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                               const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                               uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *nears,
                               const float *fars, float *xyzs, float *dirs, float *deltas, const float *noises) {
    (void)nears;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int32_t index = rays_alive[n];
        march_ctx c;
        march_ctx_init(&c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, dt_gamma, max_steps,
                       C, H, fars[index]);
        float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3,
              *pl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        t = fmaf(clampf_(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t); /* :952 */
        float last_t = t, x, y, z, dt;
        uint32_t step = 0;
        while (t < c.far && step < n_step) {
            if (march_step(&c, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            }
        }
    }
}

/* raymarching.cu:1023-1115 (normals != NULL) and :1126-1213 (SDF: alpha = sigma, no normals) */
static void composite_infer(int sdf, uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                            float *rays_t, const float *sigmas, const float *rgbs, const float *normals,
                            const float *deltas, float *weights_sum, float *depth, float *image, float *normal) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        const float *s = sigmas + (size_t)n * n_step, *c = rgbs + (size_t)n * n_step * 3,
                    *nm = normals ? normals + (size_t)n * n_step * 3 : 0, *dl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        float d = depth[index], r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        float x_ = 0, y_ = 0, z_ = 0;
        if (nm) { x_ = normal[index * 3]; y_ = normal[index * 3 + 1]; z_ = normal[index * 3 + 2]; }
        float weight_sum = weights_sum[index];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = sdf ? s[0] : 1.0f - expf(-s[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t += dl[1];
            d += weight * t;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            if (nm) { x_ += weight * nm[0]; y_ += weight * nm[1]; z_ += weight * nm[2]; }
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; if (nm) nm += 3;
            step++;
        }
        if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
        if (nm || (!sdf && normal)) { normal[index * 3] = x_; normal[index * 3 + 1] = y_; normal[index * 3 + 2] = z_; }
    }
}

ORACLE_API void ref_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                                   float *rays_t, const float *sigmas, const float *rgbs, const float *normals,
                                   const float *deltas, float *weights_sum, float *depth, float *image,
                                   float *normal) {
    composite_infer(0, n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights_sum,
                    depth, image, normal);
}
ORACLE_API void ref_composite_sdf_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                                       float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                                       float *weights_sum, float *depth, float *image) {
    composite_infer(1, n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, 0, deltas, weights_sum, depth,
                    image, 0);
}
