// Occupancy-grid ray marching and volume compositing for MI355X (gfx950, wave64).
//
// Implements Part 1 of include/mi3d.h - the 13 entry points the reference binds in
// /root/reference/raymarching/src/bindings.cpp:5-23 - as native HIP:
//
//  * march_rays_train : persistent waves, one wave = 64 rays at a time (a 128x128 view = 256 batches already
//    fills the 256 CUs).  Count pass -> wave64 inclusive scan of the per-ray sample counts -> ONE atomic
//    per wave reserves the slab for all 64 rays (the reference issues two global atomics per ray,
//    raymarching.cu:405-406) -> write pass through per-ray LDS rings, flushed as whole rows in flat order.
//    rays[] rows are in ray order.
//  * composite_rays_train fwd/bwd : one wave per RAY.  Lanes load 64 consecutive samples of the ray
//    (coalesced; the reference reads with a stride of one ray length between lanes), transmittance
//    is a wave64 multiplicative prefix scan, early termination is a ballot + first-set-lane, sums are
//    wave reductions.  Same quadrature, same break rule (accumulate the crossing sample, then stop:
//    raymarching.cu:554-557), results agree with the sequential loop to fp32 rounding.
//  * utilities (near/far, sph, morton, packbits) and the inference march/composite: one thread per
//    element, vectorised loads where the layout allows.
//
// Compiled with -ffp-contract=off: fused multiply-adds are explicit (mi3d_common.h).
#include <hip/hip_runtime.h>

#include <float.h>

#include "../../include/mi3d.h"
#include "mi3d_common.h"
#include "mi3d_dev.h"

using namespace mi3d;

namespace {

constexpr int kWave = 64;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return (int)hipGetLastError(); }
inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
__device__ __forceinline__ uint32_t cdiv_dev(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- wave64 primitives
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

__device__ __forceinline__ float wave_scan_mul(float v, int lane) {  // inclusive
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float o = __shfl_up(v, off, kWave);
        if (lane >= off) v *= o;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {  // inclusive
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float o = __shfl_up(v, off, kWave);
        if (lane >= off) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v, int lane) {  // inclusive
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, kWave);
        if (lane >= off) v += o;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// ---------------------------------------------------------------- utilities

__global__ void k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                           const float *__restrict__ aabb, uint32_t N, float min_near, float *__restrict__ nears,
                           float *__restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float rdx = 1.0f / rays_d[n * 3], rdy = 1.0f / rays_d[n * 3 + 1], rdz = 1.0f / rays_d[n * 3 + 2];

    // slab test, axis by axis; a miss on any axis marks the ray with FLT_MAX on both ends
    float lo = (aabb[0] - ox) * rdx, hi = (aabb[3] - ox) * rdx;
    if (lo > hi) { const float s = lo; lo = hi; hi = s; }
    float lo2 = (aabb[1] - oy) * rdy, hi2 = (aabb[4] - oy) * rdy;
    if (lo2 > hi2) { const float s = lo2; lo2 = hi2; hi2 = s; }
    bool miss = (lo > hi2) || (lo2 > hi);
    if (!miss) {
        if (lo2 > lo) lo = lo2;
        if (hi2 < hi) hi = hi2;
        float lo3 = (aabb[2] - oz) * rdz, hi3 = (aabb[5] - oz) * rdz;
        if (lo3 > hi3) { const float s = lo3; lo3 = hi3; hi3 = s; }
        miss = (lo > hi3) || (lo3 > hi);
        if (!miss) {
            if (lo3 > lo) lo = lo3;
            if (hi3 < hi) hi = hi3;
            if (lo < min_near) lo = min_near;
        }
    }
    nears[n] = miss ? FLT_MAX : lo;
    fars[n] = miss ? FLT_MAX : hi;
}

__global__ void k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius,
                               uint32_t N, float *__restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    // far intersection of o + t d with the sphere |p| = radius
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float Cq = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * Cq)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    coords[n * 2] = 2 * atan2f(sqrtf(x * x + z * z), y) * kRPi - 1;
    coords[n * 2 + 1] = atan2f(z, x) * kRPi;
}

__global__ void k_morton3d(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3d((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}

__global__ void k_morton3d_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t code = indices[n];  // signed shifts, as the reference shifts an `int`
    coords[n * 3] = (int32_t)compact_bits((uint32_t)(code >> 0));
    coords[n * 3 + 1] = (int32_t)compact_bits((uint32_t)(code >> 1));
    coords[n * 3 + 2] = (int32_t)compact_bits((uint32_t)(code >> 2));
}

// one thread per output byte; the 8 floats are two aligned 16-byte loads
__global__ void k_packbits(const float4 *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bits) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 a = grid[(size_t)n * 2], b = grid[(size_t)n * 2 + 1];
    uint32_t v = 0;
    v |= (a.x > thresh) ? 1u : 0u;   v |= (a.y > thresh) ? 2u : 0u;
    v |= (a.z > thresh) ? 4u : 0u;   v |= (a.w > thresh) ? 8u : 0u;
    v |= (b.x > thresh) ? 16u : 0u;  v |= (b.y > thresh) ? 32u : 0u;
    v |= (b.z > thresh) ? 64u : 0u;  v |= (b.w > thresh) ? 128u : 0u;
    bits[n] = (uint8_t)v;
}

// ---------------------------------------------------------------- training march

// Persistent waves, LDS-staged sample compaction, whole-row stores.  A wave takes batches of 64 rays (lane = ray, batch
// w, w + W, ... of the W waves of the launch) and walks each batch twice, as the reference does (raymarching.cu:335-479:
// the slab of a ray must be known before its first sample can be written, and the walk itself is the reference's
// arithmetic, step by step - the integer voxel / Morton / bit indices are bit-exact).  What is different is where the
// samples go: every occupied step appends {x, y, z, dt, delta} to its lane's ring in LDS; as soon as one ring is full the
// wave flushes ALL staged samples in flat order - consecutive lanes write consecutive ROWS of a ray's slab with one
// 12-byte (xyz, dirs) / 8-byte (deltas) store each - instead of every lane trickling 4-byte stores into its own slab
// (64 scattered dwords per instruction, 8 instructions per step).
constexpr int kRing = 8;  // staged samples per ray

struct Row3 { float a, b, c; };
struct Row2 { float a, b; };

__global__ __launch_bounds__(kWave) void k_march_train(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ bits,
    float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
    const float *__restrict__ nears, const float *__restrict__ fars, float *__restrict__ xyzs,
    float *__restrict__ dirs, float *__restrict__ deltas, int32_t *__restrict__ rays, int32_t *counter,
    const float *__restrict__ noises, uint32_t rpw /* rays per wave: 16, 32 or 64 */) {
    __shared__ float ring[kWave * kRing * 5];      // [lane][slot]{x, y, z, dt, delta}
    __shared__ float rdir[kWave * 3];              // the lane's ray direction
    __shared__ uint32_t rdst[kWave], rpfx[kWave];  // first row of the lane's staged samples; its first flat index
    __shared__ uint8_t owner[kWave * kRing];       // flat staged index -> lane
    const int lane = lane_id();
    MarchGrid g;
    march_grid_init(g, bits, bound, dt_gamma, max_steps, C, H);
    // Small views run with FEWER rays per wave (the other lanes only help in the flush) so that every SIMD has a wave to
    // issue from - see mi3d_march_rays_train: 16 rays per wave at C2 = 1024 waves.
    const uint32_t n_batches = cdiv_dev(N, rpw);

    for (uint32_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        const uint32_t n = batch * rpw + lane;
        const bool live = (uint32_t)lane < rpw && n < N;
        MarchRay r;
        float far = 0.f, t0 = 0.f;
        if (live) {
            march_ray_init(r, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
            far = fars[n];
            t0 = march_t0(nears[n], noises[n], g);
        }

        // pass 1: how many occupied steps does this ray take?
        uint32_t count = 0;
        if (live) {
            float t = t0, x, y, z, dt;
            while (t < far && count < max_steps)
                if (march_step(r, g, t, x, y, z, dt)) ++count;
        }

        // slab reservation: wave scan + one atomic per wave (counter[0] += samples, counter[1] += rays)
        const uint32_t incl = wave_scan_add_u32(count, lane);
        const uint32_t wave_total = __shfl(incl, kWave - 1, kWave);
        uint32_t base = 0;
        if (lane == 0) {
            base = (uint32_t)atomicAdd(counter, (int)wave_total);
            const uint32_t rays_here = (N - batch * rpw) < rpw ? (N - batch * rpw) : rpw;
            atomicAdd(counter + 1, (int)rays_here);
        }
        base = __shfl(base, 0, kWave);
        const uint32_t offset = base + incl - count;
        if (live) {
            rays[(size_t)n * 3] = (int32_t)n;
            rays[(size_t)n * 3 + 1] = (int32_t)offset;
            rays[(size_t)n * 3 + 2] = (int32_t)count;
        }
        // overflowing rays are dropped, not an error (raymarching.cu:416)
        bool active = live && count != 0 && offset + count <= M;

        // pass 2: replay the same walk; samples go through the rings
        if (active) { rdir[lane * 3] = r.dx; rdir[lane * 3 + 1] = r.dy; rdir[lane * 3 + 2] = r.dz; }
        float t = t0, last_t = t0;
        uint32_t step = 0, written = 0, fill = 0;
        while (__any(active) || __any(fill != 0u)) {
            // lockstep walk until one ring is full (or nobody is left walking)
            while (__any(active) && !__any(fill == (uint32_t)kRing)) {
                if (active) {
                    if (t < far && step < count) {
                        float x, y, z, dt;
                        if (march_step(r, g, t, x, y, z, dt)) {
                            float *slot = ring + ((size_t)lane * kRing + fill) * 5;
                            slot[0] = x; slot[1] = y; slot[2] = z; slot[3] = dt; slot[4] = t - last_t;
                            last_t = t;
                            ++fill;
                            ++step;
                        }
                    } else {
                        active = false;
                    }
                }
            }
            // flush: flat index i -> (owner lane, slot); consecutive i of one owner are consecutive rows of its slab
            const uint32_t fincl = wave_scan_add_u32(fill, lane);
            const uint32_t total = __shfl(fincl, kWave - 1, kWave);
            const uint32_t pfx = fincl - fill;
            rdst[lane] = offset + written;
            rpfx[lane] = pfx;
            for (uint32_t j = 0; j < fill; ++j) owner[pfx + j] = (uint8_t)lane;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < total; i += kWave) {
                const uint32_t o = owner[i], j = i - rpfx[o];
                const size_t row = (size_t)rdst[o] + j;
                const float *slot = ring + ((size_t)o * kRing + j) * 5;
                *reinterpret_cast<Row3 *>(xyzs + row * 3) = Row3{slot[0], slot[1], slot[2]};
                *reinterpret_cast<Row3 *>(dirs + row * 3) = Row3{rdir[o * 3], rdir[o * 3 + 1], rdir[o * 3 + 2]};
                *reinterpret_cast<Row2 *>(deltas + row * 2) = Row2{slot[3], slot[4]};
            }
            __builtin_amdgcn_wave_barrier();
            written += fill;
            fill = 0;
        }
    }
}

// zero the `align` padding rows the Python wrapper exposes after the last sample
__global__ void k_march_zero_tail(const int32_t *__restrict__ counter, uint32_t align, uint32_t M,
                                  float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas) {
    const uint32_t m = (uint32_t)counter[0];
    if (m >= M) return;
    uint32_t pad = align - m % align;
    if (m + pad > M) pad = M - m;
    for (uint32_t i = threadIdx.x; i < pad * 3; i += blockDim.x) {
        xyzs[(size_t)m * 3 + i] = 0.f;
        dirs[(size_t)m * 3 + i] = 0.f;
    }
    for (uint32_t i = threadIdx.x; i < pad * 2; i += blockDim.x) deltas[(size_t)m * 2 + i] = 0.f;
}

// ---------------------------------------------------------------- training composite (wave per ray)

struct CompositeChunk {
    float alpha, w, T_incl, t_incl;
    bool active;      // this lane's sample contributes
    bool terminated;  // the ray stopped inside this chunk (wave-uniform)
};

// Shared by forward and backward: loads 64 consecutive samples, returns weights/transmittance.
// T_run / t_run are wave-uniform carries (transmittance and depth parameter before this chunk).
template <bool SDF>
__device__ __forceinline__ CompositeChunk composite_chunk(const float *__restrict__ sigmas,
                                                          const float *__restrict__ deltas, uint32_t i, bool valid,
                                                          int lane, float T_thresh, float &T_run, float &t_run) {
    CompositeChunk c;
    float sigma = 0.f, d0 = 0.f, d1 = 0.f;
    if (valid) {
        sigma = sigmas[i];
        const float2 dl = reinterpret_cast<const float2 *>(deltas)[i];
        d0 = dl.x; d1 = dl.y;
    }
    c.alpha = valid ? (SDF ? sigma : 1.0f - __expf(-sigma * d0)) : 0.f;
    const float P = wave_scan_mul(1.0f - c.alpha, lane);  // inclusive product of (1 - alpha)
    float P_excl = __shfl_up(P, 1, kWave);
    if (lane == 0) P_excl = 1.0f;
    c.T_incl = T_run * P;
    const float T_excl = T_run * P_excl;
    const float S = wave_scan_add(d1, lane);
    c.t_incl = t_run + S;

    // the reference accumulates the sample that drops T below the threshold, then stops
    const unsigned long long stop = __ballot(valid && (c.T_incl < T_thresh));
    c.terminated = stop != 0ull;
    const int last = c.terminated ? (int)__ffsll((long long)stop) - 1 : kWave - 1;
    c.active = valid && lane <= last;
    c.w = c.active ? c.alpha * T_excl : 0.f;

    T_run = __shfl(c.T_incl, last, kWave);  // transmittance after the last processed sample of the chunk
    t_run = t_run + __shfl(S, kWave - 1, kWave);
    return c;
}

template <bool SDF>
__global__ __launch_bounds__(256) void k_composite_train_fwd(
    const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
    const int32_t *__restrict__ rays, uint32_t M, uint32_t N, float T_thresh, float *__restrict__ weights_sum,
    float *__restrict__ depth, float *__restrict__ image) {
    const int lane = lane_id();
    const uint32_t row = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (row >= N) return;
    const uint32_t index = (uint32_t)rays[(size_t)row * 3], offset = (uint32_t)rays[(size_t)row * 3 + 1],
                   num_steps = (uint32_t)rays[(size_t)row * 3 + 2];

    float r = 0.f, g = 0.f, b = 0.f, d = 0.f, ws = 0.f, T_run = 1.0f, t_run = 0.f;
    if (num_steps != 0 && offset + num_steps <= M) {
        for (uint32_t base = 0; base < num_steps; base += kWave) {
            const uint32_t i = offset + base + lane;
            const bool valid = base + lane < num_steps;
            const CompositeChunk c = composite_chunk<SDF>(sigmas, deltas, i, valid, lane, T_thresh, T_run, t_run);
            if (c.active) {
                const float *col = rgbs + (size_t)i * 3;
                r += c.w * col[0]; g += c.w * col[1]; b += c.w * col[2];
                d += c.w * c.t_incl;
                ws += c.w;
            }
            if (c.terminated) break;
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); d = wave_sum(d); ws = wave_sum(ws);
        if (SDF) ws = 1.0f - T_run;
    }
    if (lane == 0) {
        weights_sum[index] = ws;
        depth[index] = d;
        image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
    }
}

template <bool SDF>
__global__ __launch_bounds__(256) void k_composite_train_bwd(
    const float *__restrict__ grad_ws, const float *__restrict__ grad_image, const float *__restrict__ sigmas,
    const float *__restrict__ rgbs, const float *__restrict__ deltas, const int32_t *__restrict__ rays,
    const float *__restrict__ weights_sum, const float *__restrict__ image, uint32_t M, uint32_t N, float T_thresh,
    float *__restrict__ grad_sigmas, float *__restrict__ grad_rgbs) {
    const int lane = lane_id();
    const uint32_t row = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (row >= N) return;
    const uint32_t index = (uint32_t)rays[(size_t)row * 3], offset = (uint32_t)rays[(size_t)row * 3 + 1],
                   num_steps = (uint32_t)rays[(size_t)row * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;

    const float gws = grad_ws[index], ws_final = weights_sum[index];
    const float gi0 = grad_image[(size_t)index * 3], gi1 = grad_image[(size_t)index * 3 + 1],
                gi2 = grad_image[(size_t)index * 3 + 2];
    const float rf = image[(size_t)index * 3], gf = image[(size_t)index * 3 + 1], bf = image[(size_t)index * 3 + 2];
    const float ws_term = gws * (1.0f - ws_final);

    float T_run = 1.0f, t_run = 0.f, r_run = 0.f, g_run = 0.f, b_run = 0.f;  // colour accumulated before chunk
    for (uint32_t base = 0; base < num_steps; base += kWave) {
        const uint32_t i = offset + base + lane;
        const bool valid = base + lane < num_steps;
        const CompositeChunk c = composite_chunk<SDF>(sigmas, deltas, i, valid, lane, T_thresh, T_run, t_run);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, d0 = 0.f;
        if (valid) {
            const float *col = rgbs + (size_t)i * 3;
            c0 = col[0]; c1 = col[1]; c2 = col[2];
            d0 = deltas[(size_t)i * 2];
        }
        // colour accumulated up to and including this sample
        const float sr = wave_scan_add(c.w * c0, lane), sg = wave_scan_add(c.w * c1, lane),
                    sb = wave_scan_add(c.w * c2, lane);
        if (c.active) {
            const float r = r_run + sr, g = g_run + sg, b = b_run + sb;
            float *gc = grad_rgbs + (size_t)i * 3;
            gc[0] = gi0 * c.w; gc[1] = gi1 * c.w; gc[2] = gi2 * c.w;
            grad_sigmas[i] = d0 * (gi0 * (c.T_incl * c0 - (rf - r)) + gi1 * (c.T_incl * c1 - (gf - g)) +
                                   gi2 * (c.T_incl * c2 - (bf - b)) + ws_term);
        }
        if (c.terminated) break;
        r_run += __shfl(sr, kWave - 1, kWave);
        g_run += __shfl(sg, kWave - 1, kWave);
        b_run += __shfl(sb, kWave - 1, kWave);
    }
}

// ---------------------------------------------------------------- inference (one thread per alive ray)

// March up to n_step occupied steps of ray `index` into slot n of the round's sample buffers.  zero_tail: the rows a
// finished ray leaves unused are zeroed here (the ctl loop re-uses its buffers; the reference op gets fresh torch.zeros)
__device__ __forceinline__ void march_infer_ray(uint32_t n, int32_t index, uint32_t n_step, const float *rays_t,
                                                const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                                                uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *bits,
                                                const float *fars, float *xyzs, float *dirs, float *deltas, float noise,
                                                bool zero_tail) {
    MarchGrid g;
    march_grid_init(g, bits, bound, dt_gamma, max_steps, C, H);
    MarchRay r;
    march_ray_init(r, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3);
    const float far = fars[index];
    float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3,
          *pl = deltas + (size_t)n * n_step * 2;
    float t = march_t0(rays_t[index], noise, g);
    float last_t = t, x, y, z, dt;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        if (march_step(r, g, t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            pl[0] = dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            ++step;
        }
    }
    if (zero_tail)
        for (; step < n_step; ++step) {
            px[0] = px[1] = px[2] = 0.f; pd[0] = pd[1] = pd[2] = 0.f; pl[0] = pl[1] = 0.f;
            px += 3; pd += 3; pl += 2;
        }
}

__global__ void k_march_infer(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                              const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                              const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                              uint32_t C, uint32_t H, const uint8_t *__restrict__ bits,
                              const float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs,
                              float *__restrict__ deltas, const float *__restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    march_infer_ray(n, rays_alive[n], n_step, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, bits, fars, xyzs,
                    dirs, deltas, noises[n], false);
}

template <bool SDF>
__device__ __forceinline__ void composite_infer_ray(uint32_t n, uint32_t n_step, float T_thresh,
                                                    int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
                                                    const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                    const float *__restrict__ normals, const float *__restrict__ deltas,
                                                    float *__restrict__ weights_sum, float *__restrict__ depth,
                                                    float *__restrict__ image, float *__restrict__ normal) {
    const int32_t index = rays_alive[n];
    const float *s = sigmas + (size_t)n * n_step, *c = rgbs + (size_t)n * n_step * 3,
                *dl = deltas + (size_t)n * n_step * 2;
    const float *nm = SDF ? nullptr : normals + (size_t)n * n_step * 3;

    float t = rays_t[index], d = depth[index], ws = weights_sum[index];
    float r = image[(size_t)index * 3], g = image[(size_t)index * 3 + 1], b = image[(size_t)index * 3 + 2];
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (!SDF) { nx = normal[(size_t)index * 3]; ny = normal[(size_t)index * 3 + 1]; nz = normal[(size_t)index * 3 + 2]; }

    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0.f) break;  // rows past the ray's end are zero: the ray is finished
        const float alpha = SDF ? s[0] : 1.0f - __expf(-s[0] * dl[0]);
        const float T = 1.0f - ws;  // transmittance carried across calls through weights_sum
        const float w = alpha * T;
        ws += w;
        t += dl[1];
        d += w * t;
        r += w * c[0]; g += w * c[1]; b += w * c[2];
        if (!SDF) { nx += w * nm[0]; ny += w * nm[1]; nz += w * nm[2]; nm += 3; }
        if (T < T_thresh) break;
        ++s; c += 3; dl += 2;
        ++step;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = ws;
    depth[index] = d;
    image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
    if (!SDF) { normal[(size_t)index * 3] = nx; normal[(size_t)index * 3 + 1] = ny; normal[(size_t)index * 3 + 2] = nz; }
}

template <bool SDF>
__global__ void k_composite_infer(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive,
                                  float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                  const float *__restrict__ rgbs, const float *__restrict__ normals,
                                  const float *__restrict__ deltas, float *__restrict__ weights_sum,
                                  float *__restrict__ depth, float *__restrict__ image, float *__restrict__ normal) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    composite_infer_ray<SDF>(n, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights_sum, depth,
                             image, normal);
}

// ---------------------------------------------------------------- inference loop driven from the device
// The reference's eval loop (nerf/renderer.py:526-551) asks the HOST for the number of alive rays every round
// (`rays_alive = rays_alive[rays_alive >= 0]` is a boolean-mask copy = a device synchronisation) because the round's
// launch sizes and n_step = max(min(N // n_alive, 8), 1) depend on it.  Here that state lives in a device control
// block and every kernel of a round reads it: the host launches rounds for an upper bound of the alive count and looks
// at the real one only every few rounds.
//   ctl[0] n_alive   ctl[1] n_step   ctl[2] rows = n_alive * n_step rounded up past `align` (raymarching.py:397-400)
//   ctl[3] marching steps done so far (the loop's `step`)   ctl[4] rounds done
enum : int { CTL_ALIVE = 0, CTL_STEP = 1, CTL_ROWS = 2, CTL_DONE = 3, CTL_ROUNDS = 4, CTL_SIZE = 8 };

__device__ __forceinline__ void infer_plan(int32_t n_alive, uint32_t N, uint32_t align, int32_t *ctl) {
    int32_t n_step = 0, rows = 0;
    if (n_alive > 0) {
        n_step = (int32_t)(N / (uint32_t)n_alive);
        n_step = n_step > 8 ? 8 : (n_step < 1 ? 1 : n_step);
        rows = n_alive * n_step;
        if (align > 0) rows += (int32_t)align - rows % (int32_t)align;
    }
    ctl[CTL_ALIVE] = n_alive; ctl[CTL_STEP] = n_step; ctl[CTL_ROWS] = rows;
}

__global__ void k_infer_begin(int32_t *__restrict__ ctl, int32_t *__restrict__ rays_alive, uint32_t N, uint32_t align) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) rays_alive[i] = (int32_t)i;
    if (i == 0) {
        infer_plan((int32_t)N, N, align, ctl);
        ctl[CTL_DONE] = 0; ctl[CTL_ROUNDS] = 0;
    }
}

__global__ void k_march_infer_ctl(const int32_t *__restrict__ ctl, const int32_t *__restrict__ rays_alive,
                                  const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                  const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                  uint32_t C, uint32_t H, const uint8_t *__restrict__ bits,
                                  const float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs,
                                  float *__restrict__ deltas, const float *__restrict__ noises) {
    const uint32_t n_alive = (uint32_t)ctl[CTL_ALIVE], n_step = (uint32_t)ctl[CTL_STEP], rows = (uint32_t)ctl[CTL_ROWS];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_alive) {
        // the jitter applies to the first round only (renderer.py:546: `perturb if step == 0 else False`)
        const float noise = (noises != nullptr && ctl[CTL_ROUNDS] == 0) ? noises[n] : 0.f;
        march_infer_ray(n, rays_alive[n], n_step, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, bits, fars,
                        xyzs, dirs, deltas, noise, true);
    }
    // the alignment rows behind the last ray read as finished samples
    for (uint32_t r = n_alive * n_step + n; r < rows; r += gridDim.x * blockDim.x) {
        xyzs[(size_t)r * 3] = xyzs[(size_t)r * 3 + 1] = xyzs[(size_t)r * 3 + 2] = 0.f;
        dirs[(size_t)r * 3] = dirs[(size_t)r * 3 + 1] = dirs[(size_t)r * 3 + 2] = 0.f;
        deltas[(size_t)r * 2] = deltas[(size_t)r * 2 + 1] = 0.f;
    }
}

__global__ void k_composite_infer_ctl(const int32_t *__restrict__ ctl, float T_thresh, int32_t *__restrict__ rays_alive,
                                      float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                      const float *__restrict__ rgbs, const float *__restrict__ normals,
                                      const float *__restrict__ deltas, float *__restrict__ weights_sum,
                                      float *__restrict__ depth, float *__restrict__ image, float *__restrict__ normal) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (uint32_t)ctl[CTL_ALIVE]) return;
    composite_infer_ray<false>(n, (uint32_t)ctl[CTL_STEP], T_thresh, rays_alive, rays_t, sigmas, rgbs, normals, deltas,
                               weights_sum, depth, image, normal);
}

// Order-preserving compaction of the rays still alive (entries >= 0) - what the boolean mask does - by ONE 1024-thread
// workgroup (N is a few 10^4): wave64 ballot ranks inside a wave, an LDS scan over the 16 waves, a running base across
// chunks.  The same kernel advances the loop state and plans the next round.
__device__ __forceinline__ void infer_plan2(int32_t n_alive, int32_t *ctl);   // (the budget plan, defined further down)
template <bool BUDGET>
__global__ __launch_bounds__(1024) void k_compact_alive_ctl(int32_t *__restrict__ ctl, const int32_t *__restrict__ in,
                                                            int32_t *__restrict__ out, uint32_t N, uint32_t align,
                                                            uint32_t max_steps) {
    __shared__ uint32_t wave_count[16];
    __shared__ uint32_t base_s;
    const uint32_t n_alive = (uint32_t)ctl[CTL_ALIVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_alive; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const int32_t v = i < n_alive ? in[i] : -1;
        const bool keep = v >= 0;
        const unsigned long long mask = __ballot(keep);
        const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = wave_count[w];
            before += w < wave ? c : 0u;
            total += c;
        }
        const uint32_t base = base_s;
        if (keep) out[base + before + rank] = v;
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int32_t done = n_alive != 0u ? ctl[CTL_DONE] + ctl[CTL_STEP] : ctl[CTL_DONE];
        int32_t alive = (int32_t)base_s;
        if (done >= (int32_t)max_steps) alive = 0;  // `while step < max_steps` of the reference loop
        if (BUDGET) {
            infer_plan2(alive, ctl);
            // never plan past the loop's end: the reference's `while step < max_steps` overshoots by at most n_step - 1 <= 7
            // steps; a budget round of hundreds of steps would let a ray that is still alive there (bound > 1, dt_gamma = 0)
            // take up to max_steps - 1 more - by an amount that depends on the budget.  With the cap a
            // ray takes at most max_steps steps, whatever the budget.
            const int32_t left = (int32_t)max_steps - done;
            if (alive > 0 && ctl[CTL_STEP] > left) ctl[CTL_STEP] = left > 1 ? left : 1;
        } else {
            infer_plan(alive, N, align, ctl);
        }
        ctl[CTL_DONE] = done;
        if (n_alive != 0u) ctl[CTL_ROUNDS] += 1;   // (a round launched after the last ray died is not a round of the loop)
    }
}

// ---------------------------------------------------------------- the same loop, compact rounds under a row budget
// (round 5; include/mi3d.h Part 1b, second half).  What the reference's round structure costs: n_step = clamp(N /
// n_alive, 1, 8) steps per alive ray, laid out at n * n_step, so a round never holds more than N rows - a 128 x 128 render
// is ~280 rounds x (march, gather, MLP, head, composite, compaction) of a few thousand rows each: launch latency, two
// orders of magnitude below what the training forward pushes through the same kernels per millisecond.  A ray's result
// does not depend on where its samples are cut into rounds: its t travels in rays_t, its transmittance in weights_sum
// (T = 1 - weights_sum at every step, raymarching.cu:1075), and every accumulator is read and written back exactly.  So a
// round here takes n_step = clamp(budget / n_alive, step_min, step_max) steps, and the march PACKS what the rays really
// emitted - count pass, wave64 scan, one atomic per wave on the round's row counter, write pass: the training march's
// scheme - so rays that miss, finish or terminate cost no rows.  The restart point of a ray is the march's own t (t_next),
// not the composite's running sum of the float differences deltas[.,1] (equal whenever those sums are exact, i.e. almost
// always): a ray's sample sequence is then EXACTLY the uninterrupted march's, whatever the budget - two budgets give
// bit-identical images (tests/test_raymarching_gpu.py).
// (ABI version 5: the budget loop's ctl is int32[16]; [8] = rows the march could not place because a ray's slab would have
//  passed rows_cap - a sticky count, zeroed by begin2, that the host looks at when it reads the block: the drop used to be
//  silent)
enum : int { CTL_BUDGET = 5, CTL_STEP_MIN = 6, CTL_STEP_MAX = 7, CTL_DROPPED = 8 };

__device__ __forceinline__ void infer_plan2(int32_t n_alive, int32_t *ctl) {
    int32_t n_step = 0;
    if (n_alive > 0) {
        n_step = ctl[CTL_BUDGET] / n_alive;
        n_step = n_step > ctl[CTL_STEP_MAX] ? ctl[CTL_STEP_MAX] : n_step;
        n_step = n_step < ctl[CTL_STEP_MIN] ? ctl[CTL_STEP_MIN] : n_step;
    }
    ctl[CTL_ALIVE] = n_alive; ctl[CTL_STEP] = n_step; ctl[CTL_ROWS] = 0;
}

__global__ void k_infer_begin2(int32_t *__restrict__ ctl, int32_t *__restrict__ rays_alive, uint32_t N, uint32_t budget,
                               uint32_t step_min, uint32_t step_max) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) rays_alive[i] = (int32_t)i;
    if (i == 0) {
        ctl[CTL_BUDGET] = (int32_t)budget; ctl[CTL_STEP_MIN] = (int32_t)step_min; ctl[CTL_STEP_MAX] = (int32_t)step_max;
        infer_plan2((int32_t)N, ctl);
        ctl[CTL_DONE] = 0; ctl[CTL_ROUNDS] = 0; ctl[CTL_DROPPED] = 0;
    }
}

// lane = alive slot.  Whole waves stay in the kernel (the scan needs every lane); a lane beyond n_alive counts zero.
__global__ __launch_bounds__(128) void k_march_infer_compact_ctl(
    int32_t *__restrict__ ctl, const int32_t *__restrict__ rays_alive, const float *__restrict__ rays_t,
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
    uint32_t C, uint32_t H, const uint8_t *__restrict__ bits, const float *__restrict__ fars, uint32_t rows_cap,
    float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas, int32_t *__restrict__ ray_slab,
    float *__restrict__ t_next, const float *__restrict__ noises) {
    const uint32_t n_alive = (uint32_t)ctl[CTL_ALIVE], n_step = (uint32_t)ctl[CTL_STEP];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= n_alive) return;   // (block-uniform)
    const int lane = lane_id();
    const bool live = n < n_alive;
    MarchGrid g;
    march_grid_init(g, bits, bound, dt_gamma, max_steps, C, H);
    MarchRay r;
    int32_t index = 0;
    float far = 0.f, t0 = 0.f;
    if (live) {
        index = rays_alive[n];
        march_ray_init(r, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3);
        far = fars[index];
        // the jitter applies to the first round only (renderer.py:546: `perturb if step == 0 else False`)
        const float noise = (noises != nullptr && ctl[CTL_ROUNDS] == 0) ? noises[n] : 0.f;
        t0 = march_t0(rays_t[index], noise, g);
    }
    // pass 1: how many occupied steps (at most n_step) does the ray take this round, and where does its march stand then
    uint32_t count = 0;
    float t = t0;
    if (live) {
        float x, y, z, dt;
        while (t < far && count < n_step)
            if (march_step(r, g, t, x, y, z, dt)) ++count;
    }
    const uint32_t incl = wave_scan_add_u32(count, lane);
    const uint32_t wave_total = __shfl(incl, kWave - 1, kWave);
    uint32_t base = 0;
    if (lane == 0 && wave_total != 0u) base = (uint32_t)atomicAdd(ctl + CTL_ROWS, (int)wave_total);
    base = __shfl(base, 0, kWave);
    uint32_t offset = base + incl - count;
    if (offset + count > rows_cap) {   // cannot happen with rows_cap >= max(budget, N step_min); if it does the ray emits
        if (count != 0u) atomicAdd(ctl + CTL_DROPPED, (int)count);   // nothing this round and the host is TOLD (ctl[8], sticky)
        count = 0;
    }
    if (!live) return;
    ray_slab[(size_t)n * 2] = (int32_t)offset;
    ray_slab[(size_t)n * 2 + 1] = (int32_t)count;
    t_next[n] = t;
    // pass 2: the same walk, written
    float *px = xyzs + (size_t)offset * 3, *pd = dirs + (size_t)offset * 3, *pl = deltas + (size_t)offset * 2;
    t = t0;
    float last_t = t0, x, y, z, dt;
    uint32_t step = 0;
    while (t < far && step < count) {
        if (march_step(r, g, t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            pl[0] = dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            ++step;
        }
    }
}

// composite_rays (raymarching.cu:1039-1114) over the ray's slab: same arithmetic, same break rule; the "rows past the
// ray's end are zero" sentinel of the n * n_step layout is the slab's row count here
__global__ void k_composite_infer_compact_ctl(const int32_t *__restrict__ ctl, float T_thresh,
                                              int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
                                              const int32_t *__restrict__ ray_slab, const float *__restrict__ t_next,
                                              const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                              const float *__restrict__ normals, const float *__restrict__ deltas,
                                              float *__restrict__ weights_sum, float *__restrict__ depth,
                                              float *__restrict__ image, float *__restrict__ normal) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (uint32_t)ctl[CTL_ALIVE]) return;
    const uint32_t n_step = (uint32_t)ctl[CTL_STEP];
    const int32_t index = rays_alive[n];
    const uint32_t offset = (uint32_t)ray_slab[(size_t)n * 2], count = (uint32_t)ray_slab[(size_t)n * 2 + 1];
    const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2,
                *nm = normals + (size_t)offset * 3;
    float t = rays_t[index], d = depth[index], ws = weights_sum[index];
    float r = image[(size_t)index * 3], g = image[(size_t)index * 3 + 1], b = image[(size_t)index * 3 + 2];
    float nx = normal[(size_t)index * 3], ny = normal[(size_t)index * 3 + 1], nz = normal[(size_t)index * 3 + 2];
    uint32_t step = 0;
    while (step < count) {
        const float alpha = 1.0f - __expf(-s[0] * dl[0]);
        const float T = 1.0f - ws;  // transmittance carried across rounds through weights_sum
        const float w = alpha * T;
        ws += w;
        t += dl[1];
        d += w * t;
        r += w * c[0]; g += w * c[1]; b += w * c[2];
        nx += w * nm[0]; ny += w * nm[1]; nz += w * nm[2];
        if (T < T_thresh) break;
        ++s; c += 3; dl += 2; nm += 3;
        ++step;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t_next[n];
    weights_sum[index] = ws;
    depth[index] = d;
    image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
    normal[(size_t)index * 3] = nx; normal[(size_t)index * 3 + 1] = ny; normal[(size_t)index * 3 + 2] = nz;
}

}  // namespace

// ================================================================ C ABI

extern "C" {

int mi3d_abi_version(void) { return 5; }
const char *mi3d_last_error_string(int err) { return hipGetErrorString((hipError_t)err); }

int mi3d_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                            float *nears, float *fars, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_near_far, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d, aabb, N,
                       min_near, nears, fars);
    return launch_status();
}

int mi3d_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                      void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_sph_from_ray, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d, radius, N,
                       coords);
    return launch_status();
}

int mi3d_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_morton3d, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), coords, N, indices);
    return launch_status();
}

int mi3d_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_morton3d_invert, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), indices, N, coords);
    return launch_status();
}

int mi3d_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_packbits, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(grid), N, density_thresh, bitfield);
    return launch_status();
}

int mi3d_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                          uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                          const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                          const float *noises, void *stream) {
    if (N == 0) return 0;
    // rays per wave.  The walk is ~150 instructions per step in a dependent chain, so a wave advances at one wave's issue
    // rate whatever its lane count: a view of 16 384 rays as 256 full waves would use one SIMD in four.  Halve the rays
    // per wave until every SIMD has a wave (1024), down to 16 rays each; more waves than SIMDs only add instructions
    // (measured at C2, tools/march_bench.py: 64 rays per wave 0.75 ms, 16: 0.69, 8: 0.79, 4: 1.48).  One wave's worth of
    // rays stays one wave (its slab order is then the ray order, which tests that compare two runs row by row rely on).
    uint32_t rpw = kWave;
    const uint32_t floor_rpw = (uint32_t)MI3D_TUNE(MI3D_T_MARCH_RPW_MIN, 16), want = (uint32_t)MI3D_TUNE(MI3D_T_MARCH_WAVES, 1024);
    while (N > (uint32_t)kWave && rpw > floor_rpw && cdiv(N, rpw) < want) rpw >>= 1;
    const uint32_t batches = cdiv(N, rpw), resident = 256u * 16u;  // persistent beyond 16 waves per CU
    hipLaunchKernelGGL(k_march_train, dim3(batches < resident ? batches : resident), dim3(kWave), 0, as_stream(stream),
                       rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                       counter, noises, rpw);
    return launch_status();
}

int mi3d_march_zero_tail(const int32_t *counter, uint32_t align, uint32_t M, float *xyzs, float *dirs, float *deltas,
                         void *stream) {
    if (align == 0 || M == 0) return 0;
    hipLaunchKernelGGL(k_march_zero_tail, dim3(1), dim3(256), 0, as_stream(stream), counter, align, M, xyzs, dirs,
                       deltas);
    return launch_status();
}

#define MI3D_COMPOSITE_FWD(NAME, SDF)                                                                               \
    int NAME(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M,          \
             uint32_t N, float T_thresh, float *weights_sum, float *depth, float *image, void *stream) {            \
        if (N == 0) return 0;                                                                                       \
        hipLaunchKernelGGL(k_composite_train_fwd<SDF>, dim3(cdiv(N, 4)), dim3(256), 0, as_stream(stream), sigmas,   \
                           rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);                          \
        return launch_status();                                                                                     \
    }
MI3D_COMPOSITE_FWD(mi3d_composite_rays_train_forward, false)
MI3D_COMPOSITE_FWD(mi3d_composite_sdf_rays_train_forward, true)

#define MI3D_COMPOSITE_BWD(NAME, SDF)                                                                               \
    int NAME(const float *grad_weights_sum, const float *grad_image, const float *sigmas, const float *rgbs,        \
             const float *deltas, const int32_t *rays, const float *weights_sum, const float *image, uint32_t M,    \
             uint32_t N, float T_thresh, float *grad_sigmas, float *grad_rgbs, void *stream) {                      \
        if (N == 0) return 0;                                                                                       \
        hipLaunchKernelGGL(k_composite_train_bwd<SDF>, dim3(cdiv(N, 4)), dim3(256), 0, as_stream(stream),           \
                           grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,      \
                           T_thresh, grad_sigmas, grad_rgbs);                                                       \
        return launch_status();                                                                                     \
    }
MI3D_COMPOSITE_BWD(mi3d_composite_rays_train_backward, false)
MI3D_COMPOSITE_BWD(mi3d_composite_sdf_rays_train_backward, true)

int mi3d_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                    const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                    float *dirs, float *deltas, const float *noises, void *stream) {
    (void)nears;  // read but unused by the reference kernel as well (raymarching.cu:943)
    if (n_alive == 0) return 0;
    hipLaunchKernelGGL(k_march_infer, dim3(cdiv(n_alive, 128)), dim3(128), 0, as_stream(stream), n_alive, n_step,
                       rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs,
                       deltas, noises);
    return launch_status();
}

int mi3d_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                        const float *sigmas, const float *rgbs, const float *normals, const float *deltas,
                        float *weights_sum, float *depth, float *image, float *normal, void *stream) {
    if (n_alive == 0) return 0;
    hipLaunchKernelGGL(k_composite_infer<false>, dim3(cdiv(n_alive, 128)), dim3(128), 0, as_stream(stream), n_alive,
                       n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights_sum, depth, image,
                       normal);
    return launch_status();
}

int mi3d_composite_sdf_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                            const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum,
                            float *depth, float *image, void *stream) {
    if (n_alive == 0) return 0;
    hipLaunchKernelGGL(k_composite_infer<true>, dim3(cdiv(n_alive, 128)), dim3(128), 0, as_stream(stream), n_alive,
                       n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, nullptr, deltas, weights_sum, depth, image,
                       nullptr);
    return launch_status();
}

/* ---- inference loop driven from the device (see k_infer_begin .. k_compact_alive_ctl above) */
int mi3d_infer_begin(int32_t *ctl, int32_t *rays_alive, uint32_t N, uint32_t align, void *stream) {
    if (N == 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_infer_begin, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), ctl, rays_alive, N, align);
    return launch_status();
}

int mi3d_march_rays_ctl(const int32_t *ctl, uint32_t n_alive_max, const int32_t *rays_alive, const float *rays_t,
                        const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t *grid, const float *fars, float *xyzs, float *dirs,
                        float *deltas, const float *noises, void *stream) {
    if (n_alive_max == 0) return 0;
    hipLaunchKernelGGL(k_march_infer_ctl, dim3(cdiv(n_alive_max, 128)), dim3(128), 0, as_stream(stream), ctl, rays_alive,
                       rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises);
    return launch_status();
}

int mi3d_composite_rays_ctl(const int32_t *ctl, uint32_t n_alive_max, float T_thresh, int32_t *rays_alive,
                            float *rays_t, const float *sigmas, const float *rgbs, const float *normals,
                            const float *deltas, float *weights_sum, float *depth, float *image, float *normal,
                            void *stream) {
    if (n_alive_max == 0) return 0;
    hipLaunchKernelGGL(k_composite_infer_ctl, dim3(cdiv(n_alive_max, 128)), dim3(128), 0, as_stream(stream), ctl, T_thresh,
                       rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights_sum, depth, image, normal);
    return launch_status();
}

int mi3d_compact_alive_ctl(int32_t *ctl, const int32_t *rays_alive_in, int32_t *rays_alive_out, uint32_t N,
                           uint32_t align, uint32_t max_steps, void *stream) {
    if (N == 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_compact_alive_ctl<false>, dim3(1), dim3(1024), 0, as_stream(stream), ctl, rays_alive_in,
                       rays_alive_out, N, align, max_steps);
    return launch_status();
}

/* ---- the same loop in compact rounds under a row budget (k_infer_begin2 .. k_composite_infer_compact_ctl above) */
int mi3d_infer_begin2(int32_t *ctl, int32_t *rays_alive, uint32_t N, uint32_t budget_rows, uint32_t step_min,
                      uint32_t step_max, void *stream) {
    if (N == 0 || budget_rows == 0 || step_min == 0 || step_max < step_min || N > 0x7FFFFFFFu || budget_rows > 0x7FFFFFFFu)
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_infer_begin2, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), ctl, rays_alive, N, budget_rows,
                       step_min, step_max);
    return launch_status();
}

int mi3d_march_rays_compact_ctl(int32_t *ctl, uint32_t n_alive_max, const int32_t *rays_alive, const float *rays_t,
                                const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                uint32_t C, uint32_t H, const uint8_t *grid, const float *fars, uint32_t rows_cap,
                                float *xyzs, float *dirs, float *deltas, int32_t *ray_slab, float *t_next,
                                const float *noises, void *stream) {
    if (n_alive_max == 0) return 0;
    hipLaunchKernelGGL(k_march_infer_compact_ctl, dim3(cdiv(n_alive_max, 128)), dim3(128), 0, as_stream(stream), ctl,
                       rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, rows_cap, xyzs, dirs,
                       deltas, ray_slab, t_next, noises);
    return launch_status();
}

int mi3d_composite_rays_compact_ctl(const int32_t *ctl, uint32_t n_alive_max, float T_thresh, int32_t *rays_alive,
                                    float *rays_t, const int32_t *ray_slab, const float *t_next, const float *sigmas,
                                    const float *rgbs, const float *normals, const float *deltas, float *weights_sum,
                                    float *depth, float *image, float *normal, void *stream) {
    if (n_alive_max == 0) return 0;
    hipLaunchKernelGGL(k_composite_infer_compact_ctl, dim3(cdiv(n_alive_max, 128)), dim3(128), 0, as_stream(stream), ctl,
                       T_thresh, rays_alive, rays_t, ray_slab, t_next, sigmas, rgbs, normals, deltas, weights_sum, depth,
                       image, normal);
    return launch_status();
}

int mi3d_compact_alive_ctl2(int32_t *ctl, const int32_t *rays_alive_in, int32_t *rays_alive_out, uint32_t N,
                            uint32_t max_steps, void *stream) {
    if (N == 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_compact_alive_ctl<true>, dim3(1), dim3(1024), 0, as_stream(stream), ctl, rays_alive_in,
                       rays_alive_out, N, 0u, max_steps);
    return launch_status();
}

}  // extern "C"
