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

#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "../../include/mi3d.h"
#include "mi3d_dev.h"
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
    float4 offs[kMaxPts];  // one aligned 16-byte record per point: ONE scalar load in the point loops
    uint32_t P0, P;
    float bound;
    int mode;
    float inv2b;  // 1 / (2 bound)
    int pow2b;    // 2 bound is a power of two: t * inv2b IS t / (2 bound) (one real number, correctly rounded either way)
};

__device__ __forceinline__ void load_bases(const PointSet &ps, uint32_t s, bool valid, float (&b)[2][3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        b[0][d] = valid ? ps.x[(size_t)s * 3 + d] : 0.f;
        b[1][d] = (valid && ps.x2 != nullptr) ? ps.x2[(size_t)s * 3 + d] : 0.f;
    }
}
__device__ __forceinline__ void point_of(const PointSet &ps, const float (&b)[2][3], uint32_t p, const float4 o,
                                         float (&q)[3]) {
    // (written so that the wave-uniform choices cost scalar instructions: one 16-byte scalar load for the offset - `o` is
    // ps.offs[p], which a point loop fetches one iteration ahead - and one select per coordinate for the base; indexing
    // b[which][d] made hipcc walk all six registers per coordinate)
    const bool second = p >= ps.P0;
    const float bx = second ? b[1][0] : b[0][0], by = second ? b[1][1] : b[0][1], bz = second ? b[1][2] : b[0][2];
    if (ps.mode == 0) {
        q[0] = bx; q[1] = by; q[2] = bz;
        return;
    }
    const float w[3] = {clampf(bx + o.x, -ps.bound, ps.bound), clampf(by + o.y, -ps.bound, ps.bound),
                        clampf(bz + o.z, -ps.bound, ps.bound)};
    // (the IEEE division is ~12 instructions per coordinate; the reference's bounds - 1, 2, ... - never need it)
    if (ps.pow2b) {
#pragma unroll
        for (int d = 0; d < 3; ++d) q[d] = (w[d] + ps.bound) * ps.inv2b;
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) q[d] = (w[d] + ps.bound) / (2.0f * ps.bound);
    }
}

__device__ __forceinline__ void point_of(const PointSet &ps, const float (&b)[2][3], uint32_t p, float (&q)[3]) {
    point_of(ps, b, p, ps.offs[p], q);
}

// Feature / gradient planes hold one pair per (level, row): fp32 (8 bytes) or - under torch.autocast(float16), where
// the MLP's first layer rounds its input to binary16 and its input gradient comes out of a binary16 GEMM anyway -
// binary16 (4 bytes): the same values the fp32 planes would hold after the rounding autocast applies to them.
__device__ __forceinline__ float2 plane_pair(const float *planes, int half, size_t i) {
    if (half) {
        const uint32_t u = reinterpret_cast<const uint32_t *>(planes)[i];
        return make_float2((float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xFFFFu)),
                           (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)));
    }
    return reinterpret_cast<const float2 *>(planes)[i];
}
typedef float f32x2_native __attribute__((ext_vector_type(2)));
template <bool NT = false>
__device__ __forceinline__ void store_plane_pair(float *planes, int half, size_t i, float a, float b) {
    if (half) {
        const uint32_t u = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)a) |
                           ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
        if (NT) __builtin_nontemporal_store(u, reinterpret_cast<uint32_t *>(planes) + i);
        else reinterpret_cast<uint32_t *>(planes)[i] = u;
    } else {
        f32x2_native v = {a, b};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x2_native *>(planes) + i);
        else reinterpret_cast<f32x2_native *>(planes)[i] = v;
    }
}

// ---------------------------------------------------------------- forward
__global__ __launch_bounds__(kWave *kWaves) void k_grid_encode(PointSet ps, uint32_t n, const int32_t *count,
                                                                const float2 *__restrict__ table, GridTable T,
                                                                float *__restrict__ out) {
    __shared__ float tile[kTile * (kMaxFeat + 1)];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t n_rows = n;  // the row stride between points is the caller's n, whatever *count says
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
        // row of (sample i, point p) is p*n + i (point-major): one full 2L-float line per row, 64 rows contiguous
        for (uint32_t e = threadIdx.x; e < rows * F; e += blockDim.x) {
            const uint32_t i = e / F, f = e % F;
            out[((size_t)p * n_rows + s0 + i) * F + f] = tile[i * stride + f];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- forward, level-major planes, levels tied to XCDs
// rocprofv3 on the row kernel above (profiles/pmc_r01.json): 191 GB of FETCH_SIZE per launch against 145 GB of
// algorithmic gather bytes - every XCD touches all 16 levels (48.8 MB) through a 4 MB L2, so the hashed levels miss and
// each 8-byte gather pulls a line across the fabric.  Here the (level, tile) work list is cut into 8 contiguous
// segments of equal modelled cost and the workgroups of XCD x (block b runs on XCD b % 8) walk segment x, one level at
// a time: the table an XCD gathers from at any moment is one level (<= 4 MB) and stays L2-resident.
// Output is level-major planes [L][P*n][2], row = p*n + s (point-major): the 64 lanes of a wave store 512 contiguous
// bytes per point, and the MLP kernels read them with x_plane_rows = P*n.
//
// Round 3 corrected two things this comment used to get wrong: the coarse levels were instruction-bound
// (grid_entry's general rule, an IEEE division and a select chain per coordinate: ~300 instructions per point and level -
// see level_fast below), and the segments were not walked one level at a time per XCD until the tiles were claimed
// instead of dealt (k_grid_encode_planes: 29.8 -> 22.5 ms per 141 M evaluations, table re-fetches 23.6 -> 9.8 GB).
// What bounds it (round 2, tools/kbench.py, C2 dense, every XCD on the same level, profiles/kbench_r02_*.json): levels
// 0-7 cost 1.15-1.3 ms each, then the cost climbs with the number of distinct lines a wave's 64 consecutive samples
// touch - 1.4 / 1.7 / 2.4 / 3.0 / 3.3 ms for levels 8-12 - and saturates at 3.5 ms for levels 13-15 (4.06 with eight
// separate 8-byte loads), where every lane is in its own line: 141 M evaluations x 8 corners x one 128-byte line from
// the L2 = 145 GB per level against the L2s' ~34.5 TB/s (MI355X_MICROARCH.md) = 4.2 ms.  The gather is L2->L1
// line-bandwidth bound (16x the bytes it uses), so what helps is not asking the L2 twice for a line:
//   PAIR   the x and x+1 corners of a (y, z) corner pair are neighbours in memory whenever their entry indices differ
//          in bit 0 only - every even cx on hashed levels (x enters the hash with prime 1), every even entry index on
//          dense ones: one 16-byte load of the aligned slot serves both (44.7 -> 39.5 ms per launch);
//   ORDER  where the pair straddles a slot, the x+1 corner still sits in the same 128-byte line 15 times out of 16;
//          its load is issued DIRECTLY behind the slot's, so it merges with the pending miss instead of finding the
//          line evicted again by the other 60 lanes' lines (32 KB of L1 against 64 KB of lines per point): 39.5 ->
//          35.6 ms, saturated levels 3.8 -> 3.5 ms;
//   NT     the planes are written once and read by another kernel: stored non-temporally they stop evicting the
//          level's 4 MB table from the XCD's 4 MB L2 (FETCH_SIZE showed 41 GB of table re-fetches per launch against
//          49 MB of tables): 34.6 -> 30.4 ms with binary16 planes (loading the positions non-temporally as well: no change).
// Measured and rejected: keeping a tile's 13 results in registers and storing them after the last point, so that no
// store sits between the gathers in the shared load/store counter (30.99 -> 31.45 ms, bit-identical); non-temporal
// loads on the saturated levels (3.8 -> 10.9 ms per level); more than 3 workgroups
// per CU (+2-6 ms); evaluating the stencil three points at a time with the +-eps x-neighbours adjacent and all 24 loads
// in flight (35.6 -> 36.0 ms: the coarse levels lose to the lower occupancy what the fine ones gain); keeping the corner values of the sample's own cell in registers for its +-eps neighbours (the
// coarse levels, where it applies, are served by the L1 anyway and the extra compares made them 5-15 % slower).
constexpr uint32_t kXcds = 8;
constexpr int kMaxSegs = 16;
#ifndef MI3D_ENCODE_TRIPLE
// x-group evaluation on the fine hashed levels in a kernel instance of their own (encode_group_hash; VERDICT round 4 item 4:
// "the x-triple row sharing").  Measured in round 5 (tools/kbench.py --what encode_r05, profiles/kbench_r05_encode_xgroup.json),
// bit-identical planes: per level 1.44 -> 1.35 ms (level 9), 1.87 -> 1.75 (10), 2.33 -> 2.24 (11), 2.69 -> 2.65 (12), nothing from
// level 13 on - 0.45 ms over all levels, less than the second launch costs (whole gather 22.3 -> 22.5-23.0 ms).  The fine
// levels do not speed up in proportion to the lines they no longer ask for: NOT taken; the tools build keeps the switch.
#define MI3D_ENCODE_TRIPLE 0
#endif

struct EncodeSeg { uint32_t level, tile0, tile1, wgs; };  // wgs: workgroups of the XCD that walk this segment
struct EncodePlan {
    uint32_t n_seg[kXcds];
    EncodeSeg seg[kXcds][kMaxSegs];
    uint32_t shared;   // 1: every XCD walks the SAME list - whole levels, in order - and all claim from one counter per level
};

template <bool PAIR>
__device__ __forceinline__ void gather_corners(const GridLevel &L, const float2 *__restrict__ lvl, uint32_t cx,
                                               uint32_t cy, uint32_t cz, float2 (&v)[8]) {
    if (!PAIR) {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            v[k] = lvl[grid_entry(L, cx + (k & 1u), cy + ((k >> 1) & 1u), cz + (k >> 2))];
        return;
    }
    uint32_t e0[4], e1[4];
    float4 t[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        e0[j] = grid_entry(L, cx, cy + (j & 1u), cz + (j >> 1));
        e1[j] = grid_entry(L, cx + 1u, cy + (j & 1u), cz + (j >> 1));
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        // the aligned 16-byte slot holding entry e0 (level bases and sizes are multiples of 8 entries) ...
        t[j] = *reinterpret_cast<const float4 *>(lvl + (e0[j] & ~1u));
        // ... and, right behind it, the x+1 corner where it is not the other half of that slot: 15 times out of 16 it
        // sits in the same 128-byte line (x only touches the low index bits), and a request issued while that line's
        // miss is still pending merges with it instead of fetching the line from the L2 a second time
        if ((e0[j] ^ e1[j]) != 1u) v[2 * j + 1] = lvl[e1[j]];
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const bool odd = e0[j] & 1u;
        v[2 * j] = odd ? make_float2(t[j].z, t[j].w) : make_float2(t[j].x, t[j].y);
        if ((e0[j] ^ e1[j]) == 1u) v[2 * j + 1] = odd ? make_float2(t[j].x, t[j].y) : make_float2(t[j].z, t[j].w);
    }
}

// Fast index paths.  grid_entry() is the general rule (any dims, any table size, any input): ~30 instructions and four
// uniform branches per corner pair, and with them the gather's coarse levels were INSTRUCTION-bound - ~300 issued
// instructions per (point, level) = the 0.84-1.05 ms those levels cost whether their table sat in the L1 or in LDS.
// Two level shapes cover every level the reference's configurations build, and a stencil point (PointSet mode 1) is
// clamped into [0, 1], so its cell coordinates never exceed res - 1:
//   kDense3    3-D strided index, cy * res and cz * res^2 in 24-bit multiplies.  The index wraps (grid_entry's
//              `index >= size`) only for the +1 corners of the box's last cells: ONE compare on the largest of the
//              four (y, z) bases decides for all eight corners, and a wave with such a lane takes the general path for
//              that point.  Everywhere else entries e and e + 1 are neighbours in memory: one 8-byte-aligned 16-byte
//              load per (y, z) pair, no select.
//   kHashPow2  power-of-two table: (cx ^ cy p1 ^ cz p2) & mask, the +1 bases by adding the prime; the x+1 corner is
//              the other half of the aligned 16-byte slot exactly when cx is even (x enters with prime 1): ONE
//              predicate for the four pairs, the odd lanes fetch their four x+1 corners behind the slots.
// Same entries, same weights, same order of the eight fused multiply-adds: the planes are bit-identical.
enum LevelKind : int { kGeneral = 0, kDense3 = 1, kHashPow2 = 2 };
struct LevelFast {
    int kind;
    uint32_t res2;     // res * res (kDense3)
    uint32_t last;     // size - 1: the mask (kHashPow2), the entry whose x+1 neighbour wraps (kDense3)
};
__host__ __device__ __forceinline__ LevelFast level_fast(const GridLevel &L, int mode) {
    LevelFast f = {kGeneral, L.res * L.res, L.size - 1u};
    if (mode == 0) return f;  // raw positions may lie outside [0, 1]
    if (L.hashed) {
        if ((L.size & (L.size - 1u)) == 0u && L.size >= 2u) f.kind = kHashPow2;
    } else if (L.dims == 3 && L.res >= 2u && (uint64_t)L.res * L.res * L.res < (1ull << 31) &&
               (uint64_t)L.res * L.res * L.res <= (uint64_t)L.size) {
        f.kind = kDense3;  // the largest corner index, res (1 + res + res^2), is below 2 size: it wraps at most once
    }
    return f;
}
struct __attribute__((packed, aligned(8))) EntryPair { float ax, ay, bx, by; };  // entries e, e + 1 of a level

#ifndef MI3D_MUL24
// Tried in round 6 and NOT taken (0): 24-bit multiplies in the emit.  The emit is bound by vector-instruction issue, and a
// 32-bit integer multiply (v_mul_lo_u32) was assumed to hold the issue port four times as long as v_mul_u32_u24.  (1) The
// gather table's slot hash - eight multiplies per table pass, ~40 per (tile, level) of the coarse role - by one 24-bit
// product; (2) the spatial hash's c * prime (needed exactly mod 2^32) as two signed 24-bit products of the prime's halves,
// c * (prime & 0x7FFFFF) + ((c * (prime >> 23)) << 23), three instructions.  Same entries, same gradient (7e-8 x max: the order
// of the table's float atomics).  Product-grade builds in one process on one arena, four interleaved rounds
// (profiles/scatter_ab_libs_r06_mul24.json): dense 49.83 -> 50.05 ms, real census 39.79 -> 40.28: slightly SLOWER - the 32-bit
// multiplies are not what the issue port waits for (consistent with MI3D_TIMING_FAKE_PASS1's small ceiling).
#define MI3D_MUL24 0
#endif
__device__ __forceinline__ uint32_t mul_prime(uint32_t c, uint32_t prime, bool c_fits_24) {
    // (SIGNED 24-bit products with the prime split at bit 23: a coordinate may be base - 1 = 0xFFFFFFFF at the box's lower
    //  face, and -1 * prime mod 2^32 comes out right this way; `c_fits_24` - uniform - says every coordinate of the level
    //  lies in [-2^23, 2^23))
    if (MI3D_MUL24 != 0 && c_fits_24)
        return (uint32_t)__mul24((int)c, (int)(prime & 0x7FFFFFu)) + ((uint32_t)__mul24((int)c, (int)(prime >> 23)) << 23);
    return c * prime;
}
__device__ __forceinline__ uint32_t merge_slot(uint32_t e) {   // 9-bit slot of the per-wave gather table (kMergeSlots = 512)
    if (MI3D_MUL24 != 0) return (__umul24(e, 0x9E3779u) >> 14) & 511u;
    return (e * 2654435761u) >> (32 - 9);
}

// four corner entries of a cell - the z-bit `zb` half of its eight, corner j = x-bit | y-bit << 1 - by the short routes
// (grid_entry's values)
__device__ __forceinline__ void corner_entries4(const GridLevel &L, const LevelFast &F, uint32_t cx, uint32_t cy,
                                                uint32_t cz, uint32_t zb, uint32_t (&e)[4]) {
    if (F.kind == kHashPow2) {
        const bool f24 = (L.res >> 22) == 0u;   // (cells <= res + 1 < 2^23)
        const uint32_t hy = mul_prime(cy, kPrimeY, f24), hz = mul_prime(cz + zb, kPrimeZ, f24), a = hy ^ hz, b = (hy + kPrimeY) ^ hz, cx1 = cx + 1u;
        e[0] = (a ^ cx) & F.last; e[1] = (a ^ cx1) & F.last; e[2] = (b ^ cx) & F.last; e[3] = (b ^ cx1) & F.last;
        return;
    }
    if (F.kind == kDense3) {  // an index wraps at most once (level_fast): min(e, e - size) in unsigned arithmetic
        const uint32_t a = cx + __umul24(cy, L.res) + __umul24(cz + zb, F.res2), b = a + L.res, size = F.last + 1u;
        e[0] = min(a, a - size); e[1] = min(a + 1u, a + 1u - size); e[2] = min(b, b - size); e[3] = min(b + 1u, b + 1u - size);
        return;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) e[j] = grid_entry(L, cx + (j & 1u), cy + (j >> 1), cz + zb);
}

// one corner's entry by the same routes
__device__ __forceinline__ uint32_t corner_entry1(const GridLevel &L, const LevelFast &F, uint32_t x, uint32_t y, uint32_t z) {
    if (F.kind == kHashPow2) {
        const bool f24 = (L.res >> 22) == 0u;   // (uniform; the coordinates lie in [-1, res + 1])
        return (x ^ mul_prime(y, kPrimeY, f24) ^ mul_prime(z, kPrimeZ, f24)) & F.last;
    }
    if (F.kind == kDense3) {
        const uint32_t a = x + __umul24(y, L.res) + __umul24(z, F.res2), size = F.last + 1u;
        return min(a, a - size);
    }
    return grid_entry(L, x, y, z);
}

// returns false if this wave has to take the general path for the point (kDense3: a lane in the box's last cells)
template <int KIND, bool PAIR>
__device__ __forceinline__ bool gather_corners_fast(const GridLevel &L, const LevelFast &F, const float2 *__restrict__ lvl,
                                                    uint32_t cx, uint32_t cy, uint32_t cz, float2 (&v)[8]) {
    if (KIND == kDense3) {
        const uint32_t yz0 = __umul24(cy, L.res) + __umul24(cz, F.res2);
        const uint32_t yz[4] = {yz0, yz0 + L.res, yz0 + F.res2, yz0 + L.res + F.res2};
        if (__any(cx + yz[3] >= F.last)) return false;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const EntryPair t = *reinterpret_cast<const EntryPair *>(lvl + (cx + yz[j]));
            v[2 * j] = make_float2(t.ax, t.ay);
            v[2 * j + 1] = make_float2(t.bx, t.by);
        }
        return true;
    }
    const uint32_t hy = cy * kPrimeY, hz = cz * kPrimeZ;
    const uint32_t hy1 = hy + kPrimeY, hz1 = hz + kPrimeZ;
    const uint32_t yz[4] = {hy ^ hz, hy1 ^ hz, hy ^ hz1, hy1 ^ hz1};
    if (!PAIR) {
        const uint32_t cx1 = cx + 1u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            v[2 * j] = lvl[(yz[j] ^ cx) & F.last];
            v[2 * j + 1] = lvl[(yz[j] ^ cx1) & F.last];
        }
        return true;
    }
    const bool cx_odd = cx & 1u;
    const uint32_t cx1 = cx + 1u, slot_mask = F.last & ~1u;
    float4 t[4];
    float2 u[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t e0 = yz[j] ^ cx;
        t[j] = *reinterpret_cast<const float4 *>(lvl + (e0 & slot_mask));
        // the x+1 corner of an odd cx sits elsewhere in the level - 15 times out of 16 in the same 128-byte line; issued
        // directly behind the slot it merges with the pending miss (see ORDER above)
        if (cx_odd) u[j] = lvl[(yz[j] ^ cx1) & F.last];
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const bool e0_odd = (yz[j] ^ cx) & 1u;
        const float2 lo = make_float2(t[j].x, t[j].y), hi = make_float2(t[j].z, t[j].w);
        v[2 * j] = e0_odd ? hi : lo;
        v[2 * j + 1] = cx_odd ? u[j] : (e0_odd ? lo : hi);
    }
    return true;
}

// Stencil points that differ in their x offset only - the sample, its +x and its -x neighbour; the +x / -x pair around the
// jittered position - have the SAME (y, z) rows on a hashed level: entry = (cx ^ row hash) & mask, and cx +- a few cells
// changes the low bits only, so the three points' corners of a row lie in the same 128-byte line (16 entries) with
// probability ~ 1 - d / 16 for d cells of epsilon (level 11: 0.82, level 13: 0.67, level 15: 0.37).  Evaluated one point
// at a time those lines are fetched from the L2 once PER POINT (a wave's 64 lanes touch ~32 KB of lines per point, twelve
// waves share a 32 KB L1: nothing survives from one point to the next), and the fine levels are bound by exactly that,
// the L2 -> L1 line rate.  Here the points of such a group are evaluated TOGETHER, row by row, the loads
// of one row issued back to back: a request for a line whose miss is still pending merges with it (the mechanism the
// straddling x + 1 corner already uses, ORDER above).  Same cells, same weights, same order of the fused multiply-adds per
// point: the planes are bit-identical; the row hash and the (y, z) cell are computed once per group.
template <bool NT>
__device__ __forceinline__ void encode_group_hash(const PointSet &ps, const float (&base)[2][3], const GridLevel &L,
                                                  const LevelFast &F, const float2 *__restrict__ lvl,
                                                  float *__restrict__ planes, int out_half, size_t plane0, uint32_t n,
                                                  uint32_t s, uint32_t p, uint32_t g /* 2 or 3, uniform */) {
    uint32_t cx[3], cy = 0, cz = 0;
    float fx[3], fy = 0.f, fz = 0.f;
#pragma unroll
    for (uint32_t i = 0; i < 3; ++i) {
        if (i < g) {
            float q[3];
            point_of(ps, base, p + i, ps.offs[p + i], q);
            grid_cell(q[0], L.scale, cx[i], fx[i]);
            if (i == 0) { grid_cell(q[1], L.scale, cy, fy); grid_cell(q[2], L.scale, cz, fz); }   // (equal for the whole group)
        } else { cx[i] = 0; fx[i] = 0.f; }
    }
    const uint32_t hy = cy * kPrimeY, hz = cz * kPrimeZ, hy1 = hy + kPrimeY, hz1 = hz + kPrimeZ;
    const uint32_t yz[4] = {hy ^ hz, hy1 ^ hz, hy ^ hz1, hy1 ^ hz1};
    const uint32_t slot_mask = F.last & ~1u;
    float4 t[3][4];
    float2 u[3][4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
#pragma unroll
        for (uint32_t i = 0; i < 3; ++i) {
            if (i < g) {
                t[i][j] = *reinterpret_cast<const float4 *>(lvl + ((yz[j] ^ cx[i]) & slot_mask));
                if (cx[i] & 1u) u[i][j] = lvl[(yz[j] ^ (cx[i] + 1u)) & F.last];
            }
        }
    }
    const float gy = 1.0f - fy, gz = 1.0f - fz;
#pragma unroll
    for (uint32_t i = 0; i < 3; ++i) {
        if (i < g) {
            const bool cx_odd = cx[i] & 1u;
            float2 v[8];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const bool e0_odd = (yz[j] ^ cx[i]) & 1u;
                const float2 lo = make_float2(t[i][j].x, t[i][j].y), hi = make_float2(t[i][j].z, t[i][j].w);
                v[2 * j] = e0_odd ? hi : lo;
                v[2 * j + 1] = cx_odd ? u[i][j] : (e0_odd ? lo : hi);
            }
            const float gx = 1.0f - fx[i];
            // weights in tcnn's multiplication order ((1 * wx) * wy) * wz, corners accumulated in its order (encode_points)
            const float w00 = gx * gy, w10 = fx[i] * gy, w01 = gx * fy, w11 = fx[i] * fy;
            const float w[8] = {w00 * gz, w10 * gz, w01 * gz, w11 * gz, w00 * fz, w10 * fz, w01 * fz, w11 * fz};
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { r0 += w[k] * v[k].x; r1 += w[k] * v[k].y; }
            store_plane_pair<NT>(planes, out_half, plane0 + (size_t)(p + i) * n + s, r0, r1);
        }
    }
}

// one (tile, level): the P points of the lane's sample -> plane pairs
template <int KIND, bool PAIR, bool NT, bool TRIPLE = false>
__device__ __forceinline__ void encode_points(const PointSet &ps, const float (&base)[2][3], const GridLevel &L,
                                              const LevelFast &F, const float2 *__restrict__ lvl,
                                              float *__restrict__ planes, int out_half, size_t plane0, uint32_t n,
                                              uint32_t s) {
    float4 o_next = ps.offs[0];
    for (uint32_t p = 0; p < ps.P; ++p) {
        if (TRIPLE && KIND == kHashPow2 && PAIR && ps.mode == 1) {
            // how many points from p on differ in their x offset only (uniform: the offsets are kernel arguments)
            uint32_t g = 1;
            while (g < 3u && p + g < ps.P && ps.offs[p + g].y == ps.offs[p].y && ps.offs[p + g].z == ps.offs[p].z &&
                   ((p + g) < ps.P0) == (p < ps.P0))
                ++g;
            if (g > 1u) {
                encode_group_hash<NT>(ps, base, L, F, lvl, planes, out_half, plane0, n, s, p, g);
                p += g - 1u;
                o_next = ps.offs[p + 1 < ps.P ? p + 1 : p];
                continue;
            }
        }
        float q[3];
        const float4 o = o_next;
        o_next = ps.offs[p + 1 < ps.P ? p + 1 : p];  // (scalar load, a point ahead of its use)
        point_of(ps, base, p, o, q);
        uint32_t cx, cy, cz;
        float fx, fy, fz;
        grid_cell(q[0], L.scale, cx, fx);
        grid_cell(q[1], L.scale, cy, fy);
        grid_cell(q[2], L.scale, cz, fz);
        float2 v[8];
        if (KIND == kGeneral || !gather_corners_fast<KIND, PAIR>(L, F, lvl, cx, cy, cz, v))
            gather_corners<PAIR>(L, lvl, cx, cy, cz, v);
        const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
        // weights in tcnn's multiplication order ((1 * wx) * wy) * wz, corners accumulated in its order
        const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;
        const float w[8] = {w00 * gz, w10 * gz, w01 * gz, w11 * gz, w00 * fz, w10 * fz, w01 * fz, w11 * fz};
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { r0 += w[k] * v[k].x; r1 += w[k] * v[k].y; }
        store_plane_pair<NT>(planes, out_half, plane0 + (size_t)p * n + s, r0, r1);
    }
}

#ifndef MI3D_ENCODE_STEAL
// k_grid_encode_planes: an XCD that has finished its list helps the others with their last segments.  Measured between
// product-grade builds in one process (tools/gather_ab_libs.py, profiles/gather_ab_libs_r06_steal.json): 22.25 -> 23.35 ms -
// SLOWER: a helper starts on a table its L2 does not hold, and the tiles it claimed finish after the owner would have
// finished them.  Not taken (0).
#define MI3D_ENCODE_STEAL 0
#endif
// next unclaimed tile of every plan segment; one slot per launch in flight (zeroed in-stream before the launch)
constexpr uint32_t kPlanSlots = 64;
__device__ uint32_t g_encode_next[kPlanSlots * kXcds * kMaxSegs];
// host side of the ring (mi3d_grid_encode_points_planes_counted): who used a slot last, and the event behind that launch
// (`broken`: the event behind the slot's last launch could not be recorded - nobody can tell any more when that launch is
//  over, so the slot is never handed out again: launches that draw it deal their tiles statically)
struct PlanSlot { bool used = false; bool broken = false; hipEvent_t done = nullptr; int device = -1; };
PlanSlot g_slots[kPlanSlots];
uint32_t g_slot_launches = 0;
std::mutex g_slot_mutex;

#ifdef MI3D_DEV
// tools build only: when each XCD started, and when it finished each of its segments (100 MHz wall clock), so
// tools/kbench.py can see how well make_encode_plan's cost model balances the XCDs
__device__ unsigned long long mi3d_dbg_encode[kXcds * (1 + kMaxSegs)];
#endif

template <bool PAIR, bool NT, bool TRIPLE = false>
__global__ __launch_bounds__(kWave *kWaves) void k_grid_encode_planes(PointSet ps, uint32_t n,
                                                                       const float2 *__restrict__ table, GridTable T,
                                                                       EncodePlan plan, float *__restrict__ planes,
                                                                       int out_half, const int32_t *__restrict__ count,
                                                                       uint32_t *__restrict__ next) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t xcd = blockIdx.x % kXcds, wg_in_xcd = blockIdx.x / kXcds;
    const size_t rows_total = (size_t)n * ps.P;
    // `count` (device, optional): only samples below *count are evaluated; the plane strides stay the caller's n
    const uint32_t n_eff = count ? ((uint32_t)max(*count, 0) < n ? (uint32_t)max(*count, 0) : n) : n;
#ifdef MI3D_DEV
    if (threadIdx.x == 0) atomicMin(&mi3d_dbg_encode[xcd * (1 + kMaxSegs)], (unsigned long long)wall_clock64());
#endif
    // one segment of XCD x's list: claim its tiles until none is left
    auto walk = [&](uint32_t x, uint32_t sg, bool own) __attribute__((always_inline)) {
        const EncodeSeg seg = plan.seg[x][sg];
        const uint32_t l = seg.level;
        const GridLevel L = T.level[l];
        const float2 *lvl = table + L.offset;
        const size_t plane0 = (size_t)l * rows_total;
        // a coarse level is latency-bound and wants every workgroup the launch has (6 per CU); a fine level thrashes the
        // L1 beyond 3 per CU - the surplus workgroups skip its segments (and, the segments being ordered coarse to fine,
        // retire once the coarse ones are done)
        if (wg_in_xcd >= seg.wgs) return;
        const LevelFast F = level_fast(L, ps.mode);
        // Tiles are CLAIMED, not dealt (next[] = the segment's next unclaimed tile; one claim in flight while the wave
        // works on the previous one).  Dealt statically (tile = first + k * stride) the waves of an XCD drifted apart -
        // the CU's oldest-first issue arbitration lets the first-dispatched workgroups finish a segment milliseconds
        // before the last-dispatched ones, they move on, and the XCD's 4 MB L2 then holds two or three levels' tables at
        // once (tools/kbench.py encode_xcds: level 7 still being walked at 28 ms of a 29.6 ms launch).
        uint32_t *ctr = next ? next + x * kMaxSegs + sg : nullptr;
        if (!ctr && !own) return;
        const uint32_t stride = seg.wgs * kWaves;
        auto claim = [&](uint32_t prev) __attribute__((always_inline)) {
            if (!ctr) return prev + stride;
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(ctr, 1u);
            return seg.tile0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        };
        uint32_t tile = ctr ? claim(0u) : seg.tile0 + wg_in_xcd * kWaves + wave;
        while (tile < seg.tile1) {
            const uint32_t s = tile * kTile + lane;
            tile = claim(tile);
            if (s >= n_eff) continue;
            float base[2][3];
            load_bases(ps, s, true, base);
            if (F.kind == kDense3) encode_points<kDense3, PAIR, NT>(ps, base, L, F, lvl, planes, out_half, plane0, n, s);
            else if (F.kind == kHashPow2) encode_points<kHashPow2, PAIR, NT, TRIPLE>(ps, base, L, F, lvl, planes, out_half, plane0, n, s);
            else encode_points<kGeneral, PAIR, NT>(ps, base, L, F, lvl, planes, out_half, plane0, n, s);
        }
    };
    // The XCD's own list; with MI3D_ENCODE_STEAL (measured: slower, off) then the LAST segment of every other XCD's - the
    // cost model that cuts the list leaves the XCDs finishing 19.8 to 22.0 ms into a 22 ms launch.  One loop, so that the
    // point loops are instantiated once.
    // (plan.shared: all eight XCDs walk the levels TOGETHER - segment sg of every XCD is its eighth of level sg's tiles, and an
    //  XCD that is through with its eighth claims from the others' before anybody moves on to the next level)
    const uint32_t n_own = plan.n_seg[xcd];
    const uint32_t n_walk = plan.shared ? n_own * (next ? kXcds : 1u) : n_own + ((MI3D_ENCODE_STEAL && next) ? kXcds - 1u : 0u);
    for (uint32_t i = 0; i < n_walk; ++i) {
        bool own;
        uint32_t x, sg;
        if (plan.shared) {
            const uint32_t k = next ? i % kXcds : 0u;
            sg = next ? i / kXcds : i;
            own = k == 0u;
            x = (xcd + k) % kXcds;
        } else {
            own = i < n_own;
            x = own ? xcd : (xcd + 1u + (i - n_own)) % kXcds;
            if (!own && plan.n_seg[x] == 0u) continue;
            sg = own ? i : plan.n_seg[x] - 1u;
        }
#ifdef MI3D_DEV
        if (!plan.shared && i > 0 && i <= n_own && threadIdx.x == 0)
            atomicMax(&mi3d_dbg_encode[xcd * (1 + kMaxSegs) + i], (unsigned long long)wall_clock64());
#endif
        walk(x, sg, own);
    }
#ifdef MI3D_DEV
    if (n_walk == n_own && threadIdx.x == 0)
        atomicMax(&mi3d_dbg_encode[xcd * (1 + kMaxSegs) + n_own], (unsigned long long)wall_clock64());
#endif
}

// The coarsest levels from LDS.  Levels whose tables fit a CU's LDS together (the default grid: level 0 = 4096 entries,
// level 1 = 12 167: 130 KB of 160) are staged once per workgroup - one 1024-thread workgroup per CU, persistent over the
// tiles - and every corner is a ds_read_b64: same cells, same weights, same order of the eight fused multiply-adds as
// k_grid_encode_planes, so the planes are bit-identical; what changes is that 8 corners cost 16 LDS cycles instead of 4-8
// vector-memory instructions of 16+ cycles each (these levels ran at 0.84 ms each from the L1).
constexpr int kLdsWaves = 16;
template <bool NT>
__global__ __launch_bounds__(kWave *kLdsWaves) void k_grid_encode_planes_lds(PointSet ps, uint32_t n,
                                                                             const float2 *__restrict__ table, GridTable T,
                                                                             uint32_t n_lds_levels,
                                                                             float *__restrict__ planes, int out_half,
                                                                             const int32_t *__restrict__ count) {
    extern __shared__ float2 lds_tab[];
    const uint32_t n_eff = count ? ((uint32_t)max(*count, 0) < n ? (uint32_t)max(*count, 0) : n) : n;
    const uint32_t total = T.level[n_lds_levels - 1].offset + T.level[n_lds_levels - 1].size;  // levels are contiguous from 0
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) lds_tab[i] = table[i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t n_tiles = (n + kTile - 1) / kTile;
    const size_t rows_total = (size_t)n * ps.P;
    for (uint32_t tile = blockIdx.x * kLdsWaves + wave; tile < n_tiles; tile += gridDim.x * kLdsWaves) {
        const uint32_t s = tile * kTile + lane;
        if (s >= n_eff) continue;
        float base[2][3];
        load_bases(ps, s, true, base);
        for (uint32_t l = 0; l < n_lds_levels; ++l) {
            const GridLevel L = T.level[l];
            const float2 *lvl = lds_tab + L.offset;
            const size_t plane0 = (size_t)l * rows_total;
            const LevelFast F = level_fast(L, ps.mode);
            if (F.kind == kDense3) encode_points<kDense3, false, NT>(ps, base, L, F, lvl, planes, out_half, plane0, n, s);
            else if (F.kind == kHashPow2) encode_points<kHashPow2, false, NT>(ps, base, L, F, lvl, planes, out_half, plane0, n, s);
            else encode_points<kGeneral, false, NT>(ps, base, L, F, lvl, planes, out_half, plane0, n, s);
        }
    }
}

#ifndef MI3D_ENCODE_SHARED_LEVELS
// 1: all eight XCDs walk the levels TOGETHER (each its eighth of a level's tiles first, then the others' eighths) instead of
// one contiguous stretch of the (level, tile) list per XCD cut by a cost model.  Round 6, product builds in one process
// (tools/gather_ab_libs.py; planes bit-identical): per-XCD stretches 21.89-22.73 ms (mean 22.27), together 22.47-22.53
// (profiles/gather_ab_libs_r06_shared_levels.json) - the perfect balance buys nothing once every level change is a change for
// the whole chip; with ONE claim counter per level for all XCDs 32.5 ms (profiles/gather_ab_libs_r06_shared_one_counter.json:
// ~75 M claims/s is what one address takes).  Not taken (0).
#define MI3D_ENCODE_SHARED_LEVELS 0
#endif
#ifndef MI3D_ENCODE_COST_FIT
#define MI3D_ENCODE_COST_FIT 6   // which round's fit of the per-level cost table make_encode_plan balances the XCDs with
#endif
// Relative cost of one tile of a level, as a function of x = (marching step) x (level scale) = how many cells of the
// level two consecutive samples of a ray are apart: the measured per-level times above, tabulated against x (C2: step
// 2 sqrt(3) / 1024 in a box of side 2) and interpolated, so other step sizes and grid configurations balance too.
inline double encode_level_cost(double x, bool dense_fast) {
    // round 3, after the short index routes (profiles/kbench_r03_encode_fast.json): ms per level at C2 with 6 workgroups
    // per CU up to x = 0.26 and 3 beyond.  A dense level on the short route is instruction-bound and flat.
    static const double xs[] = {0.0, 0.136, 0.19, 0.26, 0.36, 0.50, 0.69, 0.95, 1.30, 1.80, 2.50, 3.50};
#if MI3D_ENCODE_COST_FIT == 3
    if (dense_fast) return 0.44;
    static const double cs[] = {0.74, 0.76, 0.84, 0.94, 1.10, 1.50, 1.98, 2.55, 2.87, 3.03, 3.06, 3.08};
#else
    // round 6: re-fitted IN SITU - what a tile of each level costs its XCD while the other seven walk theirs, from the
    // per-XCD, per-segment timestamps of the whole gather (tools/encode_xcd_timeline.py, profiles/encode_xcd_timeline_r06*.json;
    // tile-weighted ms per 1000 tiles of one XCD x 170 / 8, segments of 20 000 tiles and more, two plans).  Round 3's table came from per-level launches of the whole chip; under it the
    // XCDs finished 20.3 to 21.9 ms into the launch; under this one 20.6 to 22.2 on another box (what an XCD's tile costs moves
    // by ~5 % with the box and with what the other XCDs are doing): 22.25-22.98 ms against 22.37-23.0 for round 3's table between
    // product builds in one process (profiles/gather_ab_libs_r06_cost_fit.json) - equal within the run-to-run spread.  The
    // balance of the XCDs is worth <= 0.9 ms (mean end 21.0 against the last one's 21.9) and a static table cannot have it.
    if (dense_fast) return 0.383;
    static const double cs[] = {0.67, 0.685, 0.80, 0.853, 0.982, 1.416, 1.819, 2.246, 2.656, 2.72, 2.895, 2.78};
#endif
    constexpr int N = sizeof(xs) / sizeof(xs[0]);
    if (x <= xs[0]) return cs[0];
    for (int i = 1; i < N; ++i)
        if (x < xs[i]) return cs[i - 1] + (cs[i] - cs[i - 1]) * (x - xs[i - 1]) / (xs[i] - xs[i - 1]);
    return cs[N - 1];
}
// workgroups per CU a level is walked with: measured per level (same file) - 0.84 ms at 6 against 1.06 at 3 for the
// coarse levels, 1.53 at 3 against 1.67 at 6 where a wave's lanes sit in different lines (x = cells per marching step)
inline uint32_t encode_level_wgs_per_cu(double x, uint32_t coarse, uint32_t fine) { return x < 0.30 ? coarse : fine; }

// The (level, tile) list cut into kXcds contiguous segments of equal modelled cost.
inline EncodePlan make_encode_plan(const GridTable &T, uint32_t n_tiles, float step01, int only_level,
                                   uint32_t wgs_coarse_per_xcd, uint32_t wgs_fine_per_xcd, uint32_t first_level = 0,
                                   int point_mode = 1, uint32_t level_mask = 0xFFFFFFFFu) {
    EncodePlan plan{};
    double cost[MI3D_MAX_LEVELS], total = 0.0;
    for (uint32_t l = 0; l < T.n_levels; ++l) {
        cost[l] = encode_level_cost((double)step01 * (double)T.level[l].scale,
                                    level_fast(T.level[l], point_mode).kind == kDense3);
        if (only_level >= 0) cost[l] = (int)l == only_level ? 1.0 : 0.0;
        if (!((level_mask >> l) & 1u)) cost[l] = 0.0;  // another launch's levels
        if (l < first_level) cost[l] = 0.0;  // served from LDS by k_grid_encode_planes_lds
        total += cost[l];
    }
#if MI3D_ENCODE_SHARED_LEVELS
    // all eight XCDs walk level l together, each its eighth of the tiles first (its own claim counter: ONE counter for the
    // whole chip was measured at 32.5 ms against 22.2 - ~75 M claims/s is what one address takes), then the others' eighths
    // (the table is in every L2 by then): no cost model, no XCD finishing early
    {
        plan.shared = 1u;
        for (uint32_t x = 0; x < kXcds; ++x)
            for (uint32_t l = 0; l < T.n_levels; ++l)
                if (cost[l] > 0.0 && plan.n_seg[x] < (uint32_t)kMaxSegs)
                    plan.seg[x][plan.n_seg[x]++] = EncodeSeg{l, (uint32_t)((uint64_t)n_tiles * x / kXcds), (uint32_t)((uint64_t)n_tiles * (x + 1) / kXcds), encode_level_wgs_per_cu((double)step01 * (double)T.level[l].scale,
                                                                                                  wgs_coarse_per_xcd, wgs_fine_per_xcd)};
        return plan;
    }
#endif
    const double share = total / kXcds;
    uint32_t x = 0;
    double filled = 0.0;  // cost already given to XCD x
    for (uint32_t l = 0; l < T.n_levels; ++l) {
        if (cost[l] <= 0.0) continue;
        uint32_t t0 = 0;
        while (t0 < n_tiles) {
            // tiles of this level that still fit XCD x's share (the last XCD takes whatever is left)
            const double room = share - filled;
            uint32_t take = (x + 1 == kXcds) ? n_tiles - t0 : (uint32_t)ceil(room / cost[l] * (double)n_tiles - 1e-9);
            if (take > n_tiles - t0) take = n_tiles - t0;
            if (take > 0 && plan.n_seg[x] < (uint32_t)kMaxSegs) {
                const uint32_t wgs = encode_level_wgs_per_cu((double)step01 * (double)T.level[l].scale,
                                                             wgs_coarse_per_xcd, wgs_fine_per_xcd);
                plan.seg[x][plan.n_seg[x]++] = EncodeSeg{l, t0, t0 + take, wgs};
                filled += cost[l] * (double)take / (double)n_tiles;
                t0 += take;
            }
            if (t0 < n_tiles || filled >= share - 1e-12) {
                if (x + 1 < kXcds) { ++x; filled = 0.0; }
                else if (take == 0) break;  // cannot happen: the last XCD takes everything
            }
        }
    }
    return plan;
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
                                                            uint32_t plane_rows, int planes_half, uint32_t n_rep,
                                                            size_t rep_stride, float *__restrict__ grad_table) {
    // n_rep > 1: grad_table is a stack of n_rep private copies (rep_stride floats apart) of the table prefix the
    // selected levels live in; workgroup b adds into copy b % n_rep.  The few lines of a coarse level are hit by every
    // workgroup, and atomics on ONE line serialise at the memory side (~0.3 us each, profiles/scatter_levels_r01.json);
    // spreading them over copies removes the serialisation, k_replica_reduce sums the copies afterwards.
    grad_table += (size_t)(blockIdx.x % n_rep) * rep_stride;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t n_rows = n;
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
            const size_t row = (size_t)p * n_rows + s;  // point-major rows
            const float *drow = dout + row * F + f;
            for (uint32_t l = wave; l < T.n_levels; l += kWaves) {
                if (!((level_mask >> l) & 1u)) continue;
                const GridLevel L = T.level[l];
                // dout is either [rows, 2L] or (plane_rows != 0) level-major planes [L][plane_rows][2]
                float d = 0.f;
                if (valid) {
                    if (plane_rows) {
                        const float2 dd = plane_pair(dout, planes_half, (size_t)l * plane_rows + row);
                        d = f ? dd.y : dd.x;
                    } else {
                        d = drow[2 * l];
                    }
                }
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

// ---------------------------------------------------------------- backward, coarse levels ("runs" kernel)
// Levels whose cells are longer than a marching step: consecutive samples of a ray (= consecutive lanes) sit in the
// same cell for a whole RUN of lanes.  Lane = sample, so cell and weights are computed once per point (the quad kernel
// above computes them four times); the 16 corner values of every lane go to an LDS slab, then the wave is re-used as
// 4 run-slots x 16 values: lane (r, i) sums value i over the lanes of run r (sequential LDS reads, deterministic
// order) and ONE atomic instruction adds the 4 runs' 16 dwords each - the 4 dwords of an x-neighbour pair leave in the
// same request, like the quads of the kernel above.
constexpr int kRunStride = 17;  // floats per lane in the slab (16 values, padded against bank conflicts)

__global__ __launch_bounds__(kWave *kWaves) void k_scatter_runs(PointSet ps, uint32_t n, const float *__restrict__ dout,
                                                                 GridTable T, uint32_t level_mask, uint32_t plane_rows,
                                                                 int planes_half, uint32_t n_rep, size_t rep_stride,
                                                                 float *__restrict__ grad_table) {
    __shared__ float slab_all[kWaves][kWave * kRunStride];
    __shared__ uint32_t cell_all[kWaves][kWave * 3];   // cell of every lane
    __shared__ uint32_t start_all[kWaves][kWave + 1];  // first lane of run r (runs are numbered in lane order)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    float *slab = slab_all[wave];
    uint32_t *cells = cell_all[wave], *starts = start_all[wave];
    grad_table += (size_t)(blockIdx.x % n_rep) * rep_stride;
    const uint32_t F = T.n_levels * 2;
    const uint32_t s = (blockIdx.x * kWaves + wave) * kWave + lane;
    const bool valid = s < n;
    if (!__any(valid)) return;
    float base[2][3];
    load_bases(ps, s, valid, base);
    const unsigned long long lt = (1ull << lane) - 1ull;

    for (uint32_t l = 0; l < T.n_levels; ++l) {
        if (!((level_mask >> l) & 1u)) continue;
        const GridLevel L = T.level[l];
        float *lvl = grad_table + (size_t)L.offset * 2;
        for (uint32_t p = 0; p < ps.P; ++p) {
            const size_t row = (size_t)p * n + s;  // point-major rows
            float d0 = 0.f, d1 = 0.f;
            if (valid) {
                const float2 dd = plane_rows ? plane_pair(dout, planes_half, (size_t)l * plane_rows + row)
                                             : *reinterpret_cast<const float2 *>(dout + row * F + 2 * l);
                d0 = dd.x; d1 = dd.y;
            }
            const bool has = valid && (d0 != 0.f || d1 != 0.f);
            const unsigned long long act = __ballot(has);
            if (act == 0ull) continue;
            float q[3];
            point_of(ps, base, p, q);
            uint32_t cx, cy, cz;
            float fx, fy, fz;
            grid_cell(q[0], L.scale, cx, fx);
            grid_cell(q[1], L.scale, cy, fy);
            grid_cell(q[2], L.scale, cz, fz);
            const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
            const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;  // the forward's product order
            const float wk[8] = {w00 * gz, w10 * gz, w01 * gz, w11 * gz, w00 * fz, w10 * fz, w01 * fz, w11 * fz};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                slab[lane * kRunStride + 2 * k] = has ? wk[k] * d0 : 0.f;
                slab[lane * kRunStride + 2 * k + 1] = has ? wk[k] * d1 : 0.f;
            }
            cells[lane * 3] = cx; cells[lane * 3 + 1] = cy; cells[lane * 3 + 2] = cz;
            // run heads: an active lane whose predecessor is inactive or sits in another cell
            const uint32_t px = __shfl_up(cx, 1, 64), py = __shfl_up(cy, 1, 64), pz = __shfl_up(cz, 1, 64);
            const bool prev_has = (lane > 0) && ((act >> (lane - 1)) & 1ull);
            const bool head = has && !(prev_has && px == cx && py == cy && pz == cz);
            const unsigned long long heads = __ballot(head);
            const uint32_t n_runs = (uint32_t)__popcll(heads);
            if (head) starts[__popcll(heads & lt)] = (uint32_t)lane;
            // a run ends where the next head begins or at the first inactive lane after its start
            __builtin_amdgcn_wave_barrier();
            const uint32_t i = lane & 15u, slot = lane >> 4;
            for (uint32_t r0 = 0; r0 < n_runs; r0 += 4) {
                const uint32_t r = r0 + slot;
                if (r < n_runs) {
                    const uint32_t first = starts[r];
                    // length: contiguous active, non-head lanes after `first`
                    const unsigned long long stop = (~act | heads) & ~((2ull << first) - 1ull);  // bits above `first`
                    const uint32_t end = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
                    float sum = 0.f;
                    for (uint32_t j = first; j < end; ++j) sum += slab[j * kRunStride + i];
                    const uint32_t k = i >> 1, f = i & 1u;
                    const uint32_t e = grid_entry(L, cells[first * 3] + (k & 1u), cells[first * 3 + 1] + ((k >> 1) & 1u),
                                                  cells[first * 3 + 2] + (k >> 2));
                    unsafeAtomicAdd(lvl + (size_t)e * 2 + f, sum);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

PointSet make_points(const float *x, const float *x2, const float *offsets_host, uint32_t P0, uint32_t P, float bound,
                     int mode) {
    PointSet ps;
    ps.x = x; ps.x2 = x2; ps.P0 = P0; ps.P = P; ps.bound = bound; ps.mode = mode;
    int ex = 0;
    const float two_b = 2.0f * bound;
    ps.pow2b = (two_b > 0.f && two_b >= 1.17549435e-38f && two_b <= 8.5e37f && frexpf(two_b, &ex) == 0.5f) ? 1 : 0;
    ps.inv2b = ps.pow2b ? 1.0f / two_b : 0.f;
    for (uint32_t i = 0; i < (uint32_t)kMaxPts; ++i) {
        const bool have = offsets_host != nullptr && i < P;
        ps.offs[i] = make_float4(have ? offsets_host[3 * i] : 0.f, have ? offsets_host[3 * i + 1] : 0.f,
                                 have ? offsets_host[3 * i + 2] : 0.f, 0.f);
    }
    return ps;
}

int launch_scatter(const PointSet &ps, uint32_t n, const int32_t *count, const float *dout, const GridTable &T,
                   uint32_t merge_levels, float *grad_params, hipStream_t st, uint32_t level_mask = 0xFFFFFFFFu,
                   uint32_t plane_rows = 0, uint32_t n_rep = 1, size_t rep_stride = 0, int planes_half = 0) {
    merge_levels = (uint32_t)MI3D_TUNE(MI3D_T_SCATTER_MERGE, merge_levels);
    if (merge_levels > T.n_levels) merge_levels = T.n_levels;
    level_mask &= (uint32_t)MI3D_TUNE(MI3D_T_SCATTER_LEVEL_MASK, 0x7FFFFFFF);
    level_mask &= (uint32_t)((1ull << T.n_levels) - 1);
    if (level_mask == 0) return 0;
    // levels where neighbouring samples share cells go through the runs kernel (it has no device-side count)
    const uint32_t runs_mask = (count == nullptr) ? (level_mask & ((1u << merge_levels) - 1u)) : 0u;
    if (runs_mask)
        hipLaunchKernelGGL(k_scatter_runs, dim3((n + kWave * kWaves - 1) / (kWave * kWaves)), dim3(kWave * kWaves), 0, st,
                           ps, n, dout, T, runs_mask, plane_rows, planes_half, n_rep, rep_stride, grad_params);
    if (level_mask & ~runs_mask)
        hipLaunchKernelGGL(k_scatter, dim3((n + kTile - 1) / kTile), dim3(kWave * kWaves), 0, st, ps, n, count, dout, T,
                           merge_levels, level_mask & ~runs_mask, plane_rows, planes_half, n_rep, rep_stride, grad_params);
    return (int)hipGetLastError();
}

// The record path sums a tile in the gather table only where it pays: on levels whose cells are at least this many
// marching steps long (divided by default_merge_levels' own 1.05); the others emit per-point x-pair records.  4 steps =
// levels 0-6 at C2.  The gathered role is bound by its instruction stream and its LDS atomics whatever the gradients
// are, the record role by records, i.e. by how many gradient pairs are not zero.  Round 3 measured the threshold at 16-byte
// records (3 steps: level 7 gathered) - a real field iteration preferred 4.2 (89.5 -> 86.4 ms) but the dense-gradient
// scatter paid 58 -> 66 ms.  With 12-byte binary16 records and the shared-face pass (round 4) level 7 as records wins
// on both: 13-point scatter + deferred point-0 pair, one box, tools/kbench.py --what scatter_ab: dense gradients 62.25 ->
// 61.57 ms, a real step's zero census 40.87 -> 39.77 ms, whole steps -1.9 ms (tools/step_ab.py, anchored A/B); 5.8 steps
// (level 6 as records too): real 39.1, dense 66.2 - not taken (profiles/kbench_r04_scatter_ab.json).
#ifndef MI3D_MERGE_STEPS_X10_DEFAULT
#define MI3D_MERGE_STEPS_X10_DEFAULT 42
#endif
inline float merge_steps() { return (float)MI3D_TUNE(MI3D_T_MERGE_STEPS_X10, MI3D_MERGE_STEPS_X10_DEFAULT) / 10.5f; }

// levels whose cells are longer than one marching step `step01` (in [0,1] units) try to merge neighbours
uint32_t default_merge_levels(const GridTable &T, float step01) {
    uint32_t m = 0;
    for (uint32_t l = 0; l < T.n_levels; ++l)
        if (1.0f / (float)T.level[l].res >= 1.05f * step01) m = l + 1;
    return m;
}


// ================================================================ binned scatter (no global atomics)
//
// The chip retires ~21 G atomic requests/s no matter what (profiles/atomics_r01.txt, atomics3_r01.txt): scope,
// footprint and per-XCD private copies change nothing, and requests that hit one line serialise on top of that.  On
// the hashed levels nothing merges (the 8 corners of every point land in 4 random 64-byte blocks), on the dense levels
// every workgroup hammers the same few lines.  Plain stores and LDS integer atomics are an order of magnitude cheaper,
// so the gradient goes through memory instead:
//   pass 1 (k_bin_emit)   every corner contribution becomes a 12-byte record {entry, g0, g1} appended to the region
//                         of its BIN (8192 consecutive entries of one level = 64 KB of gradient).  One private
//                         region per (wave, bin): an append is a wave-private LDS counter and a plain store.  The
//                         arena is wave-major so the regions a wave is filling at any moment (one level's bins) sit
//                         in a few MB: a handful of TLB pages, and with ~1500 emitting waves the lines being
//                         appended to fit the L2s (round 2, 16-byte pair records, 141 M evaluations with dense random
//                         gradients: 256 waves 175 ms, 512: 110, 768: 89, 1024: 79, 1536: 80, 2048: 86, 4096: 94; with
//                         the gradients of a real step - whole iteration of tools/field_bench.py - 768: 130, 1024: 118.5,
//                         1536: 114.0, 2048: 114.3, 3072: 115.6, 4096: 118.2).
//                         On the coarse levels (cells of 3 marching steps and more) a wave first SUMS what its
//                         tile contributes - in registers across the stencil points that share their base position's
//                         cell, then in a 512-slot LDS hash table across lanes - and emits one record per distinct
//                         entry: 0.10 G records per 70 M evaluations where per-point run merging left 1.66 G
//                         (tools/scatter_fill.py), which had made the coarse levels the larger half of both passes.
//                         What the profile showed on the way (rocprofv3 on tools/kbench.py --what scatter13, dev
//                         level mask): the emit is insensitive to the store pattern (LDS-staged 64-byte pieces:
//                         76 -> 84 ms; level-major order, one level's region lines open at a time: 76 -> 79 ms;
//                         tools/store_bench.hip gives the store path alone 75-184 G records/s, the emit runs at 95),
//                         to a 3x shorter instruction stream on the fine path (-1 ms) and to the per-point
//                         "s_waitcnt vmcnt(0)" drains (gfx950 counts loads and stores in one counter; all pairs of
//                         a (tile, level) are now fetched up front) - because the coarse role, not the fine one, was
//                         the long pole: fine levels alone 13.5 ms per slice, coarse alone 15.0, together 24.8.
//   pass 2 (k_bin_reduce) one workgroup per (bin, split) streams the bin's records and accumulates them in LDS in
//                         64-bit fixed point (LDS fp32 atomics retire 0.38 lanes/clk/CU, 64-bit integer ones 5.3:
//                         profiles/lds_atomics_r01.txt), then adds the 64 KB tile to the gradient table.
// Samples are processed in slices so the record arena (caller-provided workspace) stays bounded.
// Round 4 (DESIGN.md 3.2'): under autocast the pair records take 12 bytes (Row12: the raw binary16 pair + 23-bit fixed-point
// weights, mi3d_common.h); a second gradient pair for stencil point 0 - what an earlier backward pass through the same
// forward left behind - rides along (mi3d_grid_scatter_binned_plus); the coarse role sends a +-eps neighbour that left the
// base cell through a shared-face pass (face_pass: the near face into the register sums, the four far corners through the
// gather table); level 7 (cells of 3.9 marching steps at C2) is a record level now (merge_steps).
// Round 5 (DESIGN.md 3.2''): the emit turned out to be bound by vector-instruction issue (0.71 of its SIMDs' slots at three
// waves per SIMD), so what moved it was its instruction stream: the coarse role sums a run's lanes in registers before its
// gather-table flush (MI3D_RUN_MERGE), a tile's gradient pairs are picked by relative register indexing instead of select
// chains that hipcc had half-spilled to scratch (MI3D_DYN_IDX), and the reduce's record loads are unconditional, so that
// they really are in flight (MI3D_REDUCE_U).  Variants are judged between PRODUCT-grade builds loaded into one process
// (tools/scatter_ab_libs.py): the tools build, with every variant behind a run-time switch, allocates registers for all
// of them.
constexpr uint32_t kBinShift = 13, kBinEntries = 1u << kBinShift;
#ifndef MI3D_EMIT_FINE_WAVES_DEFAULT
#define MI3D_EMIT_FINE_WAVES_DEFAULT 1536
#endif
constexpr uint32_t kEmitWavesMax = MI3D_EMIT_FINE_WAVES_DEFAULT;
constexpr uint32_t kReduceWavesC = 16;  // waves of a reduce workgroup (= kReduceWaves below)

struct __attribute__((packed, aligned(4))) BinRecord {
    uint32_t entry;  // level-local entry index
    float g0, g1;
};
// One record for the TWO x-neighbours of a (y, z) corner pair on a fine level.  The four values of such a pair have rank
// one - (1 - fx) (a, b) for the corner at x, fx (a, b) for the one at x + 1, with (a, b) = w_y w_z (dfeature0, dfeature1)
// - and the two entries differ in their low bits only (hashed: e1 = e0 ^ (2^t - 1), t = 1 + trailing ones of cx, because x
// enters the hash with prime 1; dense: e1 = e0 + 1), so 16 bytes carry what two 12-byte records did: half the
// lane-stores and LDS counter updates of the emit and two thirds of the bytes.
// hdr = e0 | t << 19 (t = 0: dense "+1").  fx == 0 marks a single (only e0 receives (a, b)): pairs that straddle a bin.
struct __attribute__((aligned(16))) RowRecord {
    uint32_t hdr;
    float a, b, fx;
};
constexpr uint32_t kRowEntryBits = 19;
// ... and with BINARY16 gradient planes (torch.autocast: the pair (dfeature0, dfeature1) is two binary16 numbers) the same
// pair record takes 12 bytes: the gradient pair travels as its 32 raw bits, the weight w = w_y w_z and the x fraction as
// 23-bit fixed point (|error| <= 2^-24: the last bit of an fp32 weight near 1; the gradient beside it carries binary16's
// 2^-11), and the entry as its 13 bits inside the bin - the region a record lies in IS its bin:
//   word 0 = entry & 8191 | t << 13 | (w_q >> 16) << 18 | (fx_q >> 16) << 25      word 1 = raw binary16 pair
//   word 2 = (w_q & 0xFFFF) | (fx_q & 0xFFFF) << 16            w_q = round(w 2^23), fx_q = round(fx 2^23), both < 2^23
// The reduce forms (1 - fx) w (a, b) and fx w (a, b) from them.  fx_q == 0 marks a single, as fx == 0 does above.
// A quarter less of the arena, of the emit's stores and of the reduce's loads - which is what bounds the reduce (it
// reads its records at the HBM rate: 72 GB per dense 13-point pass with 16-byte records, profiles/pmc_r03.json).
// (Row12, pack_row12 / unpack_row12 and the x-pair rule pair_flip_t live in mi3d_common.h: tests/test_host_math.py
// compiles them for the host)
__device__ __forceinline__ void unpack_row12(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t &e_local, uint32_t &t, float &w,
                                             float &fx, float &dx, float &dy) {
    unpack_row12_fields(w0, w2, e_local, t, w, fx);
    dx = (float)__builtin_bit_cast(_Float16, (unsigned short)(w1 & 0xFFFFu));
    dy = (float)__builtin_bit_cast(_Float16, (unsigned short)(w1 >> 16));
}

struct BinPlan {
    uint32_t level_bin0[MI3D_MAX_LEVELS];   // first bin of each level (bins are numbered level by level)
    uint32_t level_cap[MI3D_MAX_LEVELS];    // records one (wave, bin) region of that level holds
    uint32_t level_waves[MI3D_MAX_LEVELS];  // emitting waves of that level (coarse levels: many, fine levels: 2048)
    uint64_t level_base[MI3D_MAX_LEVELS];   // BYTE offset of the level in the arena; inside: [wave][bin][cap] records
    uint32_t level_cnt0[MI3D_MAX_LEVELS];   // first entry of the level in counts[]; inside: [wave][bin]
    uint32_t level_max0[MI3D_MAX_LEVELS];   // first entry of the level in level_max[]; inside: [wave]
    uint64_t total_bytes;
    uint32_t row_mask;                      // fine levels stored as x-pair records (RowRecord, 16 bytes; Row12 with rec12)
    uint32_t rec12;                         // binary16 gradient planes: the pair records are 12 bytes (Row12)
    uint32_t level_split[MI3D_MAX_LEVELS];  // reduce workgroups that share one bin of the level
    uint32_t level_wg0[MI3D_MAX_LEVELS];    // first reduce workgroup of the level; inside: [bin][split]
    uint32_t n_reduce_wgs;
    uint32_t total_counts, total_max;
    uint32_t n_levels, n_bins;
    uint32_t claim0;                        // level_max[claim0 ..+1] as uint32: the two roles' tile-claim counters (k_bin_emit)
};

#ifndef MI3D_EMIT_COARSE_WAVES_DEFAULT
// Emitting waves of the coarse role = regions per coarse bin the reduce has to walk (a fixed cost per slice).  The chip holds
// 3072 emitting waves at a time (256 CUs x 4 SIMDs x 3), 1536 of them the fine role's: 3072 coarse waves are two full rounds
// of the other half.  Round 6, product-grade builds in one process on a placed arena (tools/scatter_ab_libs.py,
// profiles/scatter_ab_libs_r06_coarse_waves_{56,30}GiB.json; dense / real census, ms): 16384 (rounds 3-5) 47.8 / 41.1,
// 4096 48.7 / 43.3, 3072 47.1 / 40.6, 2048 53.9 / 48.3, 1536 47.0 / 40.6, 1024 55.5 / 49.3, 512 81.3 / 75.2 - whole rounds or
// many; with four slices (30 GiB) 16384 54.4 / 44.6, 3072 51.9 / 41.8, 1536 51.7 / 41.9.
#define MI3D_EMIT_COARSE_WAVES_DEFAULT 3072
#endif
#ifndef MI3D_HASHED_SLACK
// a hashed level's region capacity over the uniform share of its records (plan_for)
#define MI3D_HASHED_SLACK 1.25
#endif
#ifndef MI3D_LEVEL_PAD_BYTES
// Padding behind every fine (pair-record) level's block of the arena.  At C2 the nine fine levels' blocks are 4.319 GB each,
// i.e. 4 GiB + 24 MB: the regions a wave appends to at the same time - same (wave, bin), nine levels - lie almost exactly
// 4 GiB apart, and the emit's time turned out to depend on the arena's PHYSICAL base with a period of 4 GiB (a block whose
// base is shifted by 2 GiB mod 4 GiB: 56.0 ms instead of 48.0, profiles/scatter_placement_r06.json `shift`).  See
// Measured (tools/scatter_bimodal.py --shift --libs, profiles/scatter_placement_r06.json
// `shift_pads`): paddings of 64 MiB / 455 MiB / 1100 MiB move the pattern (1100 MiB: 46.0 ms where the product takes 48.2 - and
// 48.5 where it takes 46.7) but every variant still swings with the placement (53.5 against 54.5 on the worst block): the
// padding is NOT the fix; 0 stays.
#define MI3D_LEVEL_PAD_BYTES 0
#endif
__host__ __device__ inline uint32_t level_bins(const GridLevel &L) { return (L.size + kBinEntries - 1) / kBinEntries; }

inline uint32_t round_waves(uint64_t w, uint32_t cap_waves) {
    uint32_t nw = (uint32_t)(w < cap_waves ? w : cap_waves);
    nw = (nw + kWaves - 1) / kWaves * kWaves;
    return nw ? nw : kWaves;
}

// The plan for slices of n_slice samples.  Fine levels are emitted by at most 1536 waves (the lines being appended to
// must fit the L2s); the coarse levels' run merging is latency-bound and emits few records, so it gets up to 3072 (see MI3D_EMIT_COARSE_WAVES_DEFAULT).
// Region capacities - hashed levels: the uniform share of the UNMERGED record count plus 25 % (the hash spreads them
// evenly).  Dense levels: bins are spatial, a wave's samples cluster in few of them, and merging thins the records by
// an unknown factor: the share assumes a quarter of the geometric run length and two-fold imbalance.  A full region is
// not an error - the overflow goes to the table by atomics.
inline BinPlan plan_for(const GridTable &T, uint64_t n_slice, uint32_t P, float step01, uint32_t merge_levels,
                        bool half_planes = false) {
    const uint32_t fine_waves = (uint32_t)MI3D_TUNE(MI3D_T_EMIT_FINE_WAVES, kEmitWavesMax);
    const uint32_t coarse_waves = (uint32_t)MI3D_TUNE(MI3D_T_EMIT_COARSE_WAVES, MI3D_EMIT_COARSE_WAVES_DEFAULT);
    BinPlan p{};
    p.n_levels = T.n_levels;
    p.rec12 = half_planes ? 1u : 0u;
    const uint64_t tiles = (n_slice + kWave - 1) / kWave;
    for (uint32_t l = 0; l < T.n_levels; ++l) {
        const GridLevel &L = T.level[l];
        const uint32_t bins = level_bins(L);
        const bool merged = l < merge_levels;
        p.level_waves[l] = round_waves(tiles, merged ? coarse_waves : fine_waves);
        const double pts_per_wave = (double)n_slice * P / p.level_waves[l];
        double per = pts_per_wave * 8.0 / bins;
        if (L.hashed) {
            per *= MI3D_HASHED_SLACK;
        } else {
            double run = merged ? (1.0 / (double)L.res) / (1.5 * (double)step01) / 4.0 : 1.0;
            run = run < 1.0 ? 1.0 : run;
            per = per / run * 2.0;
        }
        // merged (coarse) levels emit one record per distinct entry a tile touched: measured 3-6 % of the per-contribution
        // count (tools/scatter_fill.py, profiles/scatter_fill_r02.json) - their regions get a fifth of it (round 2 sized
        // them for every contribution: 52 of the arena's 92 GiB at C2).  A full region is not an error (float atomics).
        if (merged) per *= 0.2;
        p.level_cap[l] = (uint32_t)per + 64u;
        // fine fp32 levels whose x-neighbour entries are derivable from each other: one 16-byte record per corner pair
        const bool row = !merged && L.size <= (1u << kRowEntryBits) &&
                         (!L.hashed || (L.size & (L.size - 1u)) == 0u);
        if (row) { p.row_mask |= 1u << l; p.level_cap[l] = p.level_cap[l] / 2u + 64u; }
        p.level_bin0[l] = p.n_bins;
        p.level_base[l] = p.total_bytes;
        p.level_cnt0[l] = p.total_counts;
        p.level_max0[l] = p.total_max;
        p.n_bins += bins;
        p.total_bytes += (uint64_t)p.level_waves[l] * bins * p.level_cap[l] *
                         (row ? (p.rec12 ? sizeof(Row12) : sizeof(RowRecord)) : sizeof(BinRecord));
        p.total_bytes = (p.total_bytes + 255u) / 256u * 256u;
        if (row) p.total_bytes += (uint64_t)MI3D_LEVEL_PAD_BYTES;   // (see MI3D_LEVEL_PAD_BYTES)
        p.total_counts += p.level_waves[l] * bins;
        p.total_max += p.level_waves[l];
    }
    p.claim0 = p.total_max;
    p.total_max += 4u;
    return p;
}
// How many reduce workgroups share a bin.  A bin's records are spread over the level's emitting waves, so `split`
// workgroups can each take every split-th group of 16 regions.  The bins are NOT equally heavy: a fine level puts ~4.4 M
// x-pair records into each of its 64 bins at C2, but level 0 is ONE bin that receives 80 M run-merged records, level 1
// two bins with 54 M each (tools/scatter_fill.py) - with the same split for every bin the reduce waited 12 ms for those
// few workgroups.  So the split follows the expected work per bin (region capacity x waves x the usual fill - 0.78 for
// per-point records, a few per cent for the gathered records of the coarse levels, whose regions are sized for the worst
// case; an x-pair record costs two single ones), normalised so that the average bin gets `base_split` workgroups.
inline void plan_reduce_splits(BinPlan &p, const GridTable &T, uint32_t base_split, uint32_t merge_levels) {
    double work[MI3D_MAX_LEVELS], total = 0.0;
    uint32_t bins_total = 0;
    for (uint32_t l = 0; l < T.n_levels; ++l) {
        const bool row = (p.row_mask >> l) & 1u;
        work[l] = (double)p.level_waves[l] * p.level_cap[l] * (row ? 0.78 * 2.0 : (l < merge_levels ? 0.05 : 0.78));
        total += work[l] * level_bins(T.level[l]);
        bins_total += level_bins(T.level[l]);
    }
    const double target = total / bins_total / base_split;  // work one reduce workgroup should get
    p.n_reduce_wgs = 0;
    for (uint32_t l = 0; l < T.n_levels; ++l) {
        uint32_t split = (uint32_t)(work[l] / target + 0.5);
        const uint32_t most = p.level_waves[l] / kReduceWavesC ? p.level_waves[l] / kReduceWavesC : 1u;  // >= 1 region per wave
        split = split < 1u ? 1u : (split > most ? most : split);
        p.level_split[l] = split;
        p.level_wg0[l] = p.n_reduce_wgs;
        p.n_reduce_wgs += level_bins(T.level[l]) * split;
    }
}

inline size_t bin_workspace_bytes(const BinPlan &p) {
    return (size_t)p.total_bytes + (size_t)p.total_counts * sizeof(uint32_t) +
           (size_t)p.total_max * sizeof(float);
}

// The slice length of a call over n samples: ceil(n / k) for the smallest k whose plan fits the workspace (k = 1, 2, 3 ...
// up to 64, doubling from there).  Round 2-5 halved the slice (k = 1, 2, 4 ...): at C2 an arena between 31 and 46 GiB then
// ran the four-slice plan of 24.7 GB although three slices (33 GB) fit - and every slice has a fixed cost (the emit's and
// the reduce's tails, the regions the reduce walks).  `plan` receives the slice's plan (which may still not fit: the caller
// checks and takes the atomic path).
inline uint64_t slice_for(const GridTable &T, uint64_t n, uint32_t P, float step01, uint32_t merge_levels, bool half_planes,
                          size_t workspace_bytes, BinPlan &plan) {
    uint64_t n_slice = n;
    plan = plan_for(T, n_slice, P, step01, merge_levels, half_planes);
    for (uint64_t k = 2; n_slice > kWave && bin_workspace_bytes(plan) > workspace_bytes; k = k < 64 ? k + 1 : 2 * k) {
        n_slice = (n + k - 1) / k;
        plan = plan_for(T, n_slice, P, step01, merge_levels, half_planes);
    }
    return n_slice;
}

__device__ __forceinline__ void emit_record(const BinPlan &plan, const GridLevel &L, uint32_t l, uint32_t gw, uint32_t e,
                                            float g0, float g1, uint32_t *fill, BinRecord *__restrict__ arena,
                                            float *__restrict__ grad_table, float &lmax) {
    const uint32_t b = e >> kBinShift;
    // a non-finite contribution never enters the fixed-point path: it goes to the table with float atomics, which puts
    // the inf / NaN on exactly the entries the reference's atomicAdd would have put it on (GradScaler then sees it)
    const bool finite = fabsf(g0) <= 3.4028234663852886e38f && fabsf(g1) <= 3.4028234663852886e38f;
    uint32_t slot = 0xFFFFFFFFu;
    if (finite) {
        slot = atomicAdd(&fill[plan.level_bin0[l] + b], 1u);  // wave-private LDS counter
        lmax = fmaxf(lmax, fmaxf(fabsf(g0), fabsf(g1)));
    }
    if (slot < plan.level_cap[l]) {
        BinRecord r{e, g0, g1};
        reinterpret_cast<BinRecord *>(reinterpret_cast<char *>(arena) + plan.level_base[l])[((size_t)gw * level_bins(L) + b) * plan.level_cap[l] + slot] = r;
    } else {  // region full (or non-finite): straight to the table
        float *dst = grad_table + ((size_t)L.offset + e) * 2;
        unsafeAtomicAdd(dst, g0);
        unsafeAtomicAdd(dst + 1, g1);
    }
}

// round-to-nearest-even of x (|x| < 2^51) as a 64-bit two's complement integer: adding 1.5 * 2^52 in binary64 leaves it in
// the low mantissa bits.  Same result as __float2ll_rn, 5 VALU operations instead of the ~13 of the software conversion
// (there is no native float -> int64), four times per record.
__device__ __forceinline__ unsigned long long fixed_point(float x) {
    const double d = (double)x + 6755399441055744.0;
    return (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll);
}

// Coarse levels (cells of 3 marching steps and more): the 64 samples of a tile x 13 stencil points x 8 corners fall on a
// few hundred distinct entries at most, so the wave first sums them in a small LDS hash table - entry -> two 64-bit
// fixed-point sums, scaled by a power of two from the tile's largest gradient - and then emits ONE record per distinct
// entry.  (Round 1-2 summed equal-cell runs of consecutive lanes per stencil point through an LDS slab: 7x fewer records
// than contributions on level 0, 1.6x on level 7 - 1.66 G records per slice at C2, more than the fine levels', and a
// handful of bins to reduce them in.  This merges across lanes, runs AND stencil points.)  A contribution that finds no
// slot within kMergeProbes goes out as a record of its own.
constexpr uint32_t kMergeSlots = 512, kMergeProbes = 16, kMergeEmpty = 0xFFFFFFFFu;
#ifndef MI3D_RUN_MERGE
// the coarse role's group flush sums a run's lanes in registers first (flush_group).  Round 5, one box, tools/kbench.py --what
// scatter_r05, three interleaved pairs each: dense gradients 54.4-54.6 -> 53.3-53.5 ms, real census 37.7-37.8 -> 36.3-36.4;
// the coarse role alone 36.3 -> 34.2 / 25.9 -> 24.2 (profiles/kbench_r05_scatter_run_merge.json)
#define MI3D_RUN_MERGE 1
#endif
#ifndef MI3D_DYN_IDX
// a tile's 13 gradient pairs wait in registers; the point loops pick theirs by a UNIFORM index.  Written as select chains
// (round 3: "picked by a select chain (p is uniform)") that is 12 v_cndmask per pick, and hipcc kept part of the array in
// scratch memory on top (96 bytes of private segment); indexed directly, hipcc uses gfx9's relative register addressing
// (s_set_gpr_idx_on + one move) and the scratch array goes (28 bytes).  Product-grade builds in ONE process
// (tools/scatter_ab_libs.py, profiles/scatter_ab_libs_r05.json), three interleaved rounds: dense gradients 52.56-52.69 ->
// 50.74-50.78 ms, real census 43.01-43.03 -> 40.67-40.77; same gradient (1.7e-8 x max: the order of the float atomics).
#define MI3D_DYN_IDX 1
#endif
#ifndef MI3D_TIMING_SEQ_FLUSH
// TIMING ONLY (round 6; the reduce then reads garbage): the fine role's flush stores a chunk's 512 sorted records as one
// contiguous 6 KB page of the wave's region instead of ~64 runs of ~8 records at the 64 bins' append points - the store
// pattern a page-structured arena would have.  What leaves the L2s today is 740 M partial 64-byte write requests for 1.72 G
// records per slice launch, and how well HBM takes them depends on where the arena lies (13.7-18.7 ms per slice,
// profiles/scatter_placement_r06.json); this build says what the emit costs when its stores are perfectly sequential, i.e.
// the most ANY re-organisation of the stores could buy.  Result: profiles/scatter_ab_libs_r06_seq_flush.json.
#define MI3D_TIMING_SEQ_FLUSH 0
#endif
#ifndef MI3D_EMIT_ORDER_DEFAULT
// bit 0 = the fine role walks its (tile, level) pairs level-major (see k_bin_emit); 0 = tile-major, the product's order
#define MI3D_EMIT_ORDER_DEFAULT 0
#endif
#ifndef MI3D_TIMING_FAKE_PASS1
// TIMING ONLY (round 6; a variant build for tools/scatter_ab_libs.py - the gradient is garbage):
// the fine role's pass 1 without its position / cell / hash arithmetic.  What the forward gather could hand over in an index
// plane (item 1a: entry, the y / z hash differences, three fractions - 16 bytes per (evaluation, fine level)) or a delta
// hash could shorten (1b) is exactly this arithmetic: point_of (~10 vector instructions), three grid_cell (12), two
// quarter-rate 32-bit multiplies (8 issue slots), the +1 bases and the flip (9).  Here it is replaced by what UNPACKING such
// a plane would cost (a few shifts, masks and three converts) on bits that are already in registers - i.e. the index plane
// with its loads for free: an upper bound of what either idea can buy.  Result: profiles/scatter_ab_libs_r06_fake_pass1.json.
#define MI3D_TIMING_FAKE_PASS1 0
#endif
#ifndef MI3D_MASK_FMA
// the coarse role's register sums by fused multiply-add with a 0 / 1 lane mask instead of select + add.  Round 5: in the
// tools build (both forms behind a run-time switch, profiles/kbench_r05_scatter_mask_fma.json) real census 36.55-37.09 ->
// 36.34-36.60 ms, dense 53.70-54.16 -> 53.46-53.52; between two PRODUCT-grade builds in one process
// (profiles/scatter_ab_libs_r05.json) no difference (43.02 vs 43.05, 52.62 vs 52.65): neutral, kept; same gradient to
// 1e-7 x max
#define MI3D_MASK_FMA 1
#endif
// Fine levels: the x-pair records of kChunkPts stencil points of a tile (64 lanes x 4 pairs each) are SORTED BY BIN in the
// wave's LDS before they leave, so that what goes to a region is a contiguous run of records (consecutive lanes store
// consecutive 16-byte slots) instead of one scattered 16-byte store per lane.  The staging area is the memory the coarse
// role uses for its gather table (a wave has one role): 2 * kMergeSlots 64-bit words = 512 records.
constexpr uint32_t kChunkPts = 2, kStageRecs = kWave * 4 * kChunkPts;
static_assert(kStageRecs * 16 == 2 * kMergeSlots * 8, "the staging area is the gather table's sums");
static_assert(3 * 64 <= kMergeSlots, "histogram, cursors and region deltas live in the gather table's keys");
__host__ __device__ inline uint32_t emit_wave_words(uint32_t n_bins) {  // 32-bit words of LDS per emitting wave (even)
    return 4u * kMergeSlots + kMergeSlots + ((n_bins + 1u) & ~1u);
}

template <bool HP>
__global__ __launch_bounds__(kWave *kWaves, 3) void k_bin_emit(PointSet ps, uint32_t s_begin, uint32_t s_end,
                                                             const float *__restrict__ dplanes, uint32_t plane_rows,
                                                             uint32_t n_rows, const float *__restrict__ extra0,
                                                             GridTable T, BinPlan plan, uint32_t merge_levels,
                                                             uint32_t mask_a, uint32_t waves_a, uint32_t mask_b,
                                                             uint32_t waves_b, uint32_t fine_level_major,
                                                             BinRecord *__restrict__ arena,
                                                             uint32_t *__restrict__ counts,
                                                             float *__restrict__ level_max,
                                                             float *__restrict__ grad_table) {
    extern __shared__ unsigned long long emit_lds[];  // per wave: gather-table sums, its keys, the region counters
    __shared__ float lmax_all[kWaves][MI3D_MAX_LEVELS];
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    // two roles in one launch: the first waves_a waves emit the levels of mask_a (the fine group, x-pair records per
    // point: dispatched first), the next waves_b waves the levels of mask_b (the coarse group, gathered per tile), which
    // share the machine with them - at C2 the two roles are about equally long
    uint32_t gw = blockIdx.x * kWaves + wave_in_wg;
    const bool role_b = gw >= waves_a;
    if (role_b) gw -= waves_a;
    const uint32_t level_mask = role_b ? mask_b : mask_a, n_waves = role_b ? waves_b : waves_a;
    unsigned long long *sums = emit_lds + (size_t)wave_in_wg * (emit_wave_words(plan.n_bins) / 2u);  // [kMergeSlots][2]
    uint32_t *keys = reinterpret_cast<uint32_t *>(sums + 2 * kMergeSlots);                            // [kMergeSlots]
    uint32_t *fill = keys + kMergeSlots;  // [n_bins] records appended so far by this wave
    for (uint32_t b = lane; b < plan.n_bins; b += kWave) fill[b] = 0;
    for (uint32_t i = lane; i < kMergeSlots; i += kWave) { keys[i] = kMergeEmpty; sums[2 * i] = 0ull; sums[2 * i + 1] = 0ull; }
    if (lane < MI3D_MAX_LEVELS) lmax_all[wave_in_wg][lane] = 0.f;
    if (gw >= n_waves) return;

    // The (tile, level) pairs of this wave.  Tile-major (the product's order for both roles): the positions of a tile are
    // loaded once for all its levels.  Level-major for the fine role is a development switch (csrc/mi3d_dev.h
    // MI3D_T_EMIT_ORDER): the wave then appends to the 64 regions of ONE level at a time, 12.6 MB of open lines chip-wide
    // instead of 100 MB - which doubles the rate of the bare store pattern (tools/store_bench.hip,
    // profiles/store_bench_r02.txt: 184 G vs 75 G records/s) but not of this kernel: measured 2 ms slower, twice.
    const uint32_t nl = (uint32_t)__popc(level_mask);
    const uint32_t span = n_waves * kWave, first = s_begin + gw * kWave;
    const uint32_t nt = first < s_end ? (s_end - first + span - 1) / span : 0u;
    const bool level_major = !role_b && (fine_level_major & 1u);
    // Tiles are CLAIMED from a per-role counter (round 6), not dealt: a tile's cost follows its non-zero gradient pairs,
    // and with tiles dealt statically (wave gw: tiles gw, gw + n_waves ...) the launch ended when the unluckiest of its
    // persistent waves did - the slots of all the others idle meanwhile, once per slice.  (Which wave emits a tile does
    // not matter to the sums: the reduce adds integers.)  The dealt order stays behind bit 0x100000 of the order word.
    const bool claim = !level_major && !(fine_level_major & 0x100000u);
    uint32_t *claim_ctr = reinterpret_cast<uint32_t *>(level_max + plan.claim0) + (role_b ? 1 : 0);
    const uint32_t n_tiles = (s_end - s_begin + kWave - 1) / kWave;
    uint32_t claimed = 0;
#ifdef MI3D_DEV  // tools build: 0x10000 = the coarse role's shared-face pass off (A/B against round 3's pair passes);
                 // 0x20000 toggles the run-merged group flush against its product default
    const bool face_on = !(fine_level_major & 0x10000u);
    const bool run_merge = ((fine_level_major & 0x20000u) != 0u) != (MI3D_RUN_MERGE != 0);
    const bool mask_fma = ((fine_level_major & 0x40000u) != 0u) != (MI3D_MASK_FMA != 0);   // 0x40000 flips the masked fma
    const bool dyn_idx = ((fine_level_major & 0x80000u) != 0u) != (MI3D_DYN_IDX != 0);     // 0x80000 flips the register indexing
#else
    constexpr bool face_on = true;
    constexpr bool run_merge = MI3D_RUN_MERGE != 0;
    constexpr bool mask_fma = MI3D_MASK_FMA != 0;
    constexpr bool dyn_idx = MI3D_DYN_IDX != 0;
#endif
    uint32_t cur_tile = 0xFFFFFFFFu, s = 0;
    bool valid = false;
    float b00 = 0.f, b01 = 0.f, b02 = 0.f, b10 = 0.f, b11 = 0.f, b12 = 0.f;  // the tile's positions
    for (uint32_t it = 0; claim || it < nl * nt; ++it) {
        uint32_t ti = level_major ? it % nt : it / nl;
        const uint32_t li = level_major ? it / nt : it % nl;
        if (claim) {
            if (li == 0u) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(claim_ctr, 1u);
                claimed = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            }
            if (claimed >= n_tiles) break;
            ti = claimed;
        }
        uint32_t rest = level_mask;
        for (uint32_t k = 0; k < li; ++k) rest &= rest - 1u;
        const uint32_t l = (uint32_t)__builtin_ctz(rest);  // the li-th level of the role's mask
        if (ti != cur_tile) {
            cur_tile = ti;
            s = claim ? s_begin + ti * kWave + lane : first + ti * span + lane;
            valid = s < s_end;
            float fresh[2][3];
            load_bases(ps, s, valid, fresh);
            b00 = fresh[0][0]; b01 = fresh[0][1]; b02 = fresh[0][2];
            b10 = fresh[1][0]; b11 = fresh[1][1]; b12 = fresh[1][2];
        }
        {
            const float base[2][3] = {{b00, b01, b02}, {b10, b11, b12}};
            const GridLevel L = T.level[l];
            const bool merge = l < merge_levels;
            float lmax = 0.f;
            // wave-uniform, hoisted out of the point loop: the level's region capacity, this wave's counters and regions
            const uint32_t cap = plan.level_cap[l];
            const LevelFast LF = level_fast(L, ps.mode);
            uint32_t *fill_l = fill + plan.level_bin0[l];
            RowRecord *region0 = reinterpret_cast<RowRecord *>(reinterpret_cast<char *>(arena) + plan.level_base[l]) +
                                 (size_t)gw * level_bins(L) * cap;
            // ALL gradient pairs of this (tile, level) are fetched before its first record is stored.  gfx950 has one
            // counter for vector loads and stores (vmcnt), and the compiler cannot count the conditional stores between a
            // load and its use - so a load consumed inside the point loop costs "s_waitcnt vmcnt(0)": a full drain of the
            // wave's scattered stores, once per point.  Up front it is one drain per 13 points; the pairs wait in
            // registers as raw bits and are picked by a select chain (p is uniform).
            // point-major rows: the pair of (sample s, point p) sits at plane[p * n_rows + s] - 512 contiguous bytes per wave
            // (unconditional loads from clamped rows - a predicated load makes the compiler wait for it on the spot; what
            // an invalid lane or a point >= P fetched is never looked at)
            const size_t prow = (size_t)l * plane_rows + (valid ? s : s_end - 1u);
            uint32_t raw0[kMaxPts], raw1[HP ? 1 : kMaxPts];
            if (HP) {
#pragma unroll
                for (uint32_t k = 0; k < (uint32_t)kMaxPts; ++k) {
                    const uint32_t kc = k < ps.P ? k : ps.P - 1u;
                    raw0[k] = reinterpret_cast<const uint32_t *>(dplanes)[prow + (size_t)kc * n_rows];
                    raw1[0] = 0u;
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < (uint32_t)kMaxPts; ++k) {
                    const uint32_t kc = k < ps.P ? k : ps.P - 1u;
                    const uint2 v = reinterpret_cast<const uint2 *>(dplanes)[prow + (size_t)kc * n_rows];
                    raw0[k] = v.x; raw1[k] = v.y;
                }
            }
            // A SECOND gradient pair for stencil point 0 (extra0: planes [L][n_rows][2] of the same element type, or null):
            // the point-0 planes an earlier backward pass through the same forward left behind (the reference's
            // latents.backward, nerf/sd.py:171, reaches sigma / albedo of point 0 only) ride along with this pass instead
            // of paying for a scatter of their own; the two pairs are added in fp32 before anything else looks at them.
            float ex0x = 0.f, ex0y = 0.f;
            if (extra0 != nullptr) {
                const size_t erow = (size_t)l * n_rows + (valid ? s : s_end - 1u);
                if (HP) {
                    const uint32_t u = reinterpret_cast<const uint32_t *>(extra0)[erow];
                    ex0x = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xFFFFu));
                    ex0y = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
                } else {
                    const float2 v = reinterpret_cast<const float2 *>(extra0)[erow];
                    ex0x = v.x; ex0y = v.y;
                }
            }
            // coarse level: the power-of-two scale of the wave's gather table, from the tile's largest gradient (every
            // contribution is a weight <= 1 times a gradient, 6656 of them per tile: |sum| < 2^13 2^40)
            float merge_scale = 0.f;
            double merge_unscale = 0.0;
            bool merge_any = false, dead = false;
            if (merge) {
                float tmax = 0.f;
                bool bad = false;
#pragma unroll
                for (uint32_t k = 0; k < (uint32_t)kMaxPts; ++k) {
                    float dx = HP ? (float)__builtin_bit_cast(_Float16, (unsigned short)(raw0[k] & 0xFFFFu))
                                           : __uint_as_float(raw0[k]);
                    float dy = HP ? (float)__builtin_bit_cast(_Float16, (unsigned short)(raw0[k] >> 16))
                                           : __uint_as_float(raw1[HP ? 0 : k]);
                    if (k == 0u) { dx += ex0x; dy += ex0y; }
                    if (valid && k < ps.P) {
                        bad |= !(fabsf(dx) <= 3.4028234663852886e38f && fabsf(dy) <= 3.4028234663852886e38f);
                        tmax = fmaxf(tmax, fmaxf(fabsf(dx), fabsf(dy)));
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64));
                if (__ballot(bad) != 0ull) {  // inf / NaN in the tile: its contributions go to the table one by one (float
                    dead = true;              // atomics, as the reference adds them) instead of through the fixed-point sums
                } else if (tmax > 0.f) {
                    int ex;
                    (void)frexpf(tmax, &ex);  // tmax < 2^ex
                    int k2 = 40 - ex;
                    k2 = k2 > 126 ? 126 : (k2 < -126 ? -126 : k2);
                    merge_scale = ldexpf(1.0f, k2);
                    merge_unscale = ldexp(1.0, -k2);
                    merge_any = true;
                }
            }
            uint32_t bx = 0, by = 0, bz = 0;  // the cell of the current group's base position
            float acc0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            bool acc_any = false;
            // The 8 corners of a cell into the wave's gather table (kMergeSlots).  All eight first probes are issued
            // before any of their results is looked at: an LDS compare-and-swap returns after ~120 cycles, and eight
            // dependent probe loops in a row (round 2) made the coarse role a chain of LDS latencies - 36 of the 61 ms of
            // a dense 13-point emit (rocprofv3 per level, profiles/kernel_trace_r03_scatter_levels.csv).  A probe that
            // finds its slot taken by another entry (rare at 512 slots for a few hundred entries) walks on alone.
            auto gather8 = [&](uint32_t ex, uint32_t ey, uint32_t ez, const float (&g0)[8], const float (&g1)[8]) __attribute__((always_inline)) {
                static_assert(kMergeSlots == 512, "the slot hash keeps 9 bits");
#ifdef MI3D_DEV  // tools build: what the gather table's atomics cost (timing only - the sums are wrong with either bit)
                const bool dbg_no_add = fine_level_major & 0x100u, dbg_no_cas = fine_level_major & 0x200u;
#else
                constexpr bool dbg_no_add = false, dbg_no_cas = false;
#endif
#pragma unroll
                for (uint32_t half = 0; half < 2; ++half) {  // two batches of four probes in flight (registers)
                    uint32_t e[4], slot[4], old[4];
                    corner_entries4(L, LF, ex, ey, ez, half, e);
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) slot[j] = merge_slot(e[j]);
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t k = 4 * half + j;
                        old[j] = ((g0[k] != 0.f || g1[k] != 0.f) && !dbg_no_cas) ? atomicCAS(&keys[slot[j]], kMergeEmpty, e[j]) : e[j];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t k = 4 * half + j;
                        if (!(g0[k] != 0.f || g1[k] != 0.f)) continue;
                        bool placed = old[j] == kMergeEmpty || old[j] == e[j];
                        for (uint32_t tries = 1; !placed && tries < kMergeProbes; ++tries) {
                            slot[j] = (slot[j] + 1u) & (kMergeSlots - 1u);
                            const uint32_t o = atomicCAS(&keys[slot[j]], kMergeEmpty, e[j]);
                            placed = o == kMergeEmpty || o == e[j];
                        }
                        if (placed) {
                            if (!dbg_no_add) {
                                atomicAdd(&sums[2 * slot[j]], fixed_point(g0[k] * merge_scale));
                                atomicAdd(&sums[2 * slot[j] + 1], fixed_point(g1[k] * merge_scale));
                            }
                        } else {
                            emit_record(plan, L, l, gw, e[j], g0[k], g1[k], fill, arena, grad_table, lmax);
                        }
                    }
                }
            };
            // four given entries into the gather table (the second half of what gather8 does per batch)
            auto gather4 = [&](const uint32_t (&e)[4], const float (&g0)[4], const float (&g1)[4]) __attribute__((always_inline)) {
                uint32_t slot[4], old[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) slot[j] = merge_slot(e[j]);
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)
                    old[j] = (g0[j] != 0.f || g1[j] != 0.f) ? atomicCAS(&keys[slot[j]], kMergeEmpty, e[j]) : e[j];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    if (!(g0[j] != 0.f || g1[j] != 0.f)) continue;
                    bool placed = old[j] == kMergeEmpty || old[j] == e[j];
                    for (uint32_t tries = 1; !placed && tries < kMergeProbes; ++tries) {
                        slot[j] = (slot[j] + 1u) & (kMergeSlots - 1u);
                        const uint32_t o = atomicCAS(&keys[slot[j]], kMergeEmpty, e[j]);
                        placed = o == kMergeEmpty || o == e[j];
                    }
                    if (placed) {
                        atomicAdd(&sums[2 * slot[j]], fixed_point(g0[j] * merge_scale));
                        atomicAdd(&sums[2 * slot[j] + 1], fixed_point(g1[j] * merge_scale));
                    } else {
                        emit_record(plan, L, l, gw, e[j], g0[j], g1[j], fill, arena, grad_table, lmax);
                    }
                }
            };
            // A +eps / -eps neighbour pair along AXIS whose point left the base cell sits in the cell NEXT to it: the four
            // corners on the face between the two cells ARE corners of the base cell - those contributions join the group's
            // register sums - and only the four corners of the far face go through the gather table: 12 LDS atomics per
            // pair pass instead of 24 (a lane cannot leave on both sides, so one far face per lane; the LDS atomics are 56 %
            // of this role, profiles/kbench_r03_scatter_diag.json).  fa: the lane's + point moved to base + 1 along AXIS,
            // fb: its - point moved to base - 1; everything else equal to the base cell.
            auto face_pass = [&](auto axis_tag, bool fa, bool fb, const float (&a0)[8], const float (&a1)[8],
                                 const float (&b0)[8], const float (&b1)[8]) __attribute__((always_inline)) {
                constexpr uint32_t AXIS = decltype(axis_tag)::value, m = 1u << AXIS;
                if (mask_fma) {
                    const float mfa = fa ? 1.0f : 0.0f, mfb = fb ? 1.0f : 0.0f;
#pragma unroll
                    for (uint32_t k = 0; k < 8; ++k) {
                        if (!(k & m)) { acc0[k | m] = fmaf(a0[k], mfa, acc0[k | m]); acc1[k | m] = fmaf(a1[k], mfa, acc1[k | m]); }
                        else { acc0[k & ~m] = fmaf(b0[k], mfb, acc0[k & ~m]); acc1[k & ~m] = fmaf(b1[k], mfb, acc1[k & ~m]); }
                    }
                } else {
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    if (!(k & m)) { acc0[k | m] += fa ? a0[k] : 0.f; acc1[k | m] += fa ? a1[k] : 0.f; }   // +: its near face
                    else { acc0[k & ~m] += fb ? b0[k] : 0.f; acc1[k & ~m] += fb ? b1[k] : 0.f; }          // -: its near face
                }
                }
                uint32_t e[4];
                float f0[4], f1[4];
                const uint32_t far_c = (AXIS == 0 ? bx : (AXIS == 1 ? by : bz)) + (fa ? 2u : 0xFFFFFFFFu);   // base + 2 / base - 1
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t lo = j & 1u, hi = j >> 1;   // the two other axes' bits, in x < y < z order
                    const uint32_t k1 = AXIS == 0 ? (1u | lo << 1 | hi << 2) : (AXIS == 1 ? (lo | 2u | hi << 2) : (lo | hi << 1 | 4u));
                    const uint32_t k0 = k1 & ~m;
                    f0[j] = fa ? a0[k1] : (fb ? b0[k0] : 0.f);
                    f1[j] = fa ? a1[k1] : (fb ? b1[k0] : 0.f);
                    const uint32_t x = AXIS == 0 ? far_c : bx + lo;
                    const uint32_t y = AXIS == 1 ? far_c : by + (AXIS == 0 ? lo : hi);
                    const uint32_t z = AXIS == 2 ? far_c : bz + hi;
                    e[j] = corner_entry1(L, LF, x, y, z);
                }
                gather4(e, f0, f1);
            };
            auto flush_group = [&]() __attribute__((always_inline)) {
                if (acc_any) {
                    if (run_merge) {
                        // Consecutive samples of a ray sit in ONE cell of a coarse level for a whole run of lanes (37
                        // marching steps per cell on level 0, 5 on level 6), so in the flush below most of a wave's lanes
                        // add into the SAME 8 slots: LDS atomics on one address serialise, and these 24 instructions - with
                        // every lane active - are where the gather table's time goes (an atomic here retires ~70 cycles per
                        // instruction and CU against ~12 for distinct addresses, profiles/lds_atomics_r01.txt).  So the
                        // register sums of a run are first added up ACROSS its lanes, in segments of at most 8 lanes that
                        // never cross an 8-lane boundary (three steps of row_shr, pure VALU: no LDS), and only a segment's
                        // last lane goes to the table - 8 x fewer lanes per address for 3 x 16 DPP moves + adds.
                        const uint32_t pbx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bx, 0x111, 0xf, 0xf, true);
                        const uint32_t pby = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)by, 0x111, 0xf, 0xf, true);
                        const uint32_t pbz = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bz, 0x111, 0xf, 0xf, true);
                        const bool head = (lane & 7) == 0 || pbx != bx || pby != by || pbz != bz;
                        const unsigned long long heads = __ballot(head);
                        const uint32_t first = 63u - (uint32_t)__builtin_clzll(heads & ((2ull << lane) - 1ull));
                        const uint32_t pos = (uint32_t)lane - first;                        // 0 .. 7 inside the segment
                        const bool tail = lane == 63 || (((heads >> 1) >> lane) & 1ull) != 0ull;
                        auto seg_step = [&](auto ctrl_tag, uint32_t off) __attribute__((always_inline)) {
                            constexpr int CTRL = decltype(ctrl_tag)::value;   // (the DPP control is an instruction immediate)
                            float u0[8], u1[8];
#pragma unroll
                            for (uint32_t k = 0; k < 8; ++k) {   // (moved under the full exec mask: a masked DPP source reads 0)
                                u0[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc0[k]), CTRL, 0xf, 0xf, true));
                                u1[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc1[k]), CTRL, 0xf, 0xf, true));
                            }
                            const bool take = pos >= off;
                            if (mask_fma) {
                                const float mt = take ? 1.0f : 0.0f;
#pragma unroll
                                for (uint32_t k = 0; k < 8; ++k) { acc0[k] = fmaf(u0[k], mt, acc0[k]); acc1[k] = fmaf(u1[k], mt, acc1[k]); }
                            } else {
#pragma unroll
                            for (uint32_t k = 0; k < 8; ++k) { acc0[k] += take ? u0[k] : 0.f; acc1[k] += take ? u1[k] : 0.f; }
                            }
                        };
                        seg_step(std::integral_constant<int, 0x111>{}, 1u);   // row_shr:1
                        seg_step(std::integral_constant<int, 0x112>{}, 2u);   // row_shr:2
                        seg_step(std::integral_constant<int, 0x114>{}, 4u);   // row_shr:4
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) { acc0[k] = tail ? acc0[k] : 0.f; acc1[k] = tail ? acc1[k] : 0.f; }
                    }
                    gather8(bx, by, bz, acc0, acc1);
#pragma unroll
                    for (uint32_t k = 0; k < 8; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
                }
                acc_any = false;
            };
            if (!merge && ((plan.row_mask >> l) & 1u)) {
                // ---- fine level: one 16-byte record per x-corner pair {e0 | t << 19, a, b, fx}, sorted by bin in LDS
                // kChunkPts stencil points at a time.  What round 2's counters and this round's per-level traces say
                // about the scattered version (one 16-byte store per lane and record): 123 G records/s per level
                // whatever the instruction stream - the rate at which the L2s take partial-line writes - and 1536
                // emitting waves at most, because every open line of every (wave, bin) region had to stay in the L2s
                // until its 8th record arrived.  Sorted runs leave as whole lines.
                uint4 *stage = reinterpret_cast<uint4 *>(sums);
                uint32_t *hist = keys, *cursor = keys + 64, *gdelta = keys + 128;
                const uint32_t mask = L.size - 1u;
                // binary16 planes: 12-byte records (Row12).  The first pass's point-0 pair (extra0) cannot be added into
                // point 0's raw bits, so it travels as a record of its own: one more "point" at point 0's position
                Row12 *region12 = reinterpret_cast<Row12 *>(reinterpret_cast<char *>(arena) + plan.level_base[l]) +
                                  (size_t)gw * level_bins(L) * cap;
                const bool extra_pt = HP && extra0 != nullptr;
                const uint32_t Pf = ps.P + (extra_pt ? 1u : 0u);
                uint32_t ex_raw = 0u;
                if (extra_pt) ex_raw = reinterpret_cast<const uint32_t *>(extra0)[(size_t)l * n_rows + (valid ? s : s_end - 1u)];
                auto to_table = [&](uint32_t e, float va, float vb) __attribute__((always_inline)) {  // straight to the table (float atomics)
                    float *dst = grad_table + ((size_t)L.offset + e) * 2;
                    unsafeAtomicAdd(dst, va); unsafeAtomicAdd(dst + 1, vb);
                };
                auto put_single = [&](uint32_t e, float va, float vb) __attribute__((always_inline)) {  // a record of its own, appended directly
                    const uint32_t bin = e >> kBinShift;
                    const uint32_t slot = atomicAdd(&fill_l[bin], 1u);
                    if (slot < cap) region0[__umul24(bin, cap) + slot] = RowRecord{e, va, vb, 0.f};
                    else to_table(e, va, vb);
                };
                // (binary16 planes: the single carries its whole weight wgt = (1 - fx) w or fx w, and fx_q = 0)
                auto put_single12 = [&](uint32_t e, float wgt, uint32_t raw, float dx, float dy) __attribute__((always_inline)) {
                    const uint32_t bin = e >> kBinShift;
                    const uint32_t slot = atomicAdd(&fill_l[bin], 1u);
                    if (slot < cap) region12[__umul24(bin, cap) + slot] = pack_row12(e & (kBinEntries - 1u), 0u, wgt, 0.f, raw);
                    else to_table(e, wgt * dx, wgt * dy);
                };
                // the four (y, z) corner pairs of a cell: e0 = entry of the x corner, e1 = of the x + 1 corner
                auto pair_entries = [&](uint32_t cx, uint32_t cy, uint32_t cz, uint32_t j, uint32_t &e0, uint32_t &e1) __attribute__((always_inline)) {
                    if (L.hashed) {  // e(x + 1) = e(x) ^ flip, flip = the bits a +1 carry changes in cx (x enters with prime 1)
                        e0 = (cx ^ ((cy + (j & 1u)) * kPrimeY) ^ ((cz + (j >> 1)) * kPrimeZ)) & mask;
                        e1 = e0 ^ ((cx ^ (cx + 1u)) & mask);
                    } else {
                        e0 = grid_entry(L, cx, cy + (j & 1u), cz + (j >> 1));
                        e1 = grid_entry(L, cx + 1u, cy + (j & 1u), cz + (j >> 1));
                    }
                };
                // a pair record needs both entries in one bin AND derivable from each other: hashed e1 = e0 ^ flip by
                // construction, dense e1 = e0 + 1 (not where grid_entry wraps the +1 corner to entry 0)
                auto pairable = [&](uint32_t e0, uint32_t e1) __attribute__((always_inline)) {
                    return (e0 >> kBinShift) == (e1 >> kBinShift) && (L.hashed || e1 == e0 + 1u);
                };
                hist[lane] = 0u;
                for (uint32_t p0 = 0; p0 < Pf; p0 += kChunkPts) {
                    // (per chunk point: its x cell and fractions, its gradient pair, the four x-corner entries e0 and which of
                    // them leave a pair record - kept for pass 2, which used to derive them a second time: with the
                    // hash's two 32-bit multiplies per corner pair that was a third of the role's 820 VALU instructions
                    // per chunk, and the role is issue-bound for its occupancy - SQ_ACTIVE_INST_ANY 0.45 of the wave
                    // cycles at 1.5 waves per SIMD, profiles/pmc_r03_emit_roles.json)
                    uint32_t ccx[kChunkPts], ce0[kChunkPts][4], cpm[kChunkPts];
                    float cfx[kChunkPts], cfy[kChunkPts], cfz[kChunkPts], cd0[kChunkPts], cd1[kChunkPts];
                    // the chunk's gradient pairs out of the tile's (p0 is uniform and even: one select chain per chunk)
                    uint32_t craw0[kChunkPts], craw1[kChunkPts];
                    static_assert(kChunkPts == 2, "the pick below walks the points two at a time");
                    craw0[0] = raw0[0]; craw0[1] = raw0[1]; craw1[0] = raw1[0]; craw1[1] = raw1[HP ? 0 : 1];
                    if (dyn_idx) {
                        const uint32_t ka = p0 < (uint32_t)kMaxPts ? p0 : 0u, kb = ka + 1 < (uint32_t)kMaxPts ? ka + 1 : ka;
                        craw0[0] = raw0[ka]; craw0[1] = raw0[kb]; craw1[0] = raw1[HP ? 0 : ka]; craw1[1] = raw1[HP ? 0 : kb];
                    } else {
#pragma unroll
                    for (uint32_t k = 2; k < (uint32_t)kMaxPts; k += 2) {
                        const uint32_t k1 = k + 1 < (uint32_t)kMaxPts ? k + 1 : k;
                        craw0[0] = (p0 == k) ? raw0[k] : craw0[0]; craw0[1] = (p0 == k) ? raw0[k1] : craw0[1];
                        craw1[0] = (p0 == k) ? raw1[HP ? 0 : k] : craw1[0]; craw1[1] = (p0 == k) ? raw1[HP ? 0 : k1] : craw1[1];
                    }
                    }
                    // pass 1: cells, entries, and how many records each bin gets
#pragma unroll
                    for (uint32_t c = 0; c < kChunkPts; ++c) {
                        const uint32_t p = p0 + c;  // uniform
                        if (extra_pt && p == ps.P) craw0[c] = ex_raw;   // the extra "point": point 0's position, the first pass's pair
                        const uint32_t r0 = craw0[c], r1 = craw1[c];
                        float2 d = HP
                            ? make_float2((float)__builtin_bit_cast(_Float16, (unsigned short)(r0 & 0xFFFFu)),
                                          (float)__builtin_bit_cast(_Float16, (unsigned short)(r0 >> 16)))
                            : make_float2(__uint_as_float(r0), __uint_as_float(r1));
                        if (!HP && p == 0u) { d.x += ex0x; d.y += ex0y; }   // (uniform; fp32 planes: summed in place)
                        const bool has = valid && p < Pf && (d.x != 0.f || d.y != 0.f);
                        const bool finite = fabsf(d.x) <= 3.4028234663852886e38f && fabsf(d.y) <= 3.4028234663852886e38f;
#if MI3D_TIMING_FAKE_PASS1   // TIMING ONLY (see the switch's comment above k_bin_emit): no position, no cells, no hash
                        uint32_t cy = 0, cz = 0;
                        const uint32_t fk = r0 ^ (r0 >> 9) ^ (s << 7) ^ (p * 0x51ED27u);
                        ccx[c] = fk >> 13;
                        cfx[c] = (float)(fk & 1023u) * (1.0f / 1024.0f);
                        cfy[c] = (float)((fk >> 10) & 1023u) * (1.0f / 1024.0f);
                        cfz[c] = (float)((fk >> 20) & 1023u) * (1.0f / 1024.0f);
#else
                        float q[3];
                        point_of(ps, base, p < ps.P ? p : 0u, q);
                        uint32_t cy, cz;
                        grid_cell(q[0], L.scale, ccx[c], cfx[c]);
                        grid_cell(q[1], L.scale, cy, cfy[c]);
                        grid_cell(q[2], L.scale, cz, cfz[c]);
#endif
                        cd0[c] = d.x; cd1[c] = d.y;
                        const bool ok = has && finite;
                        if (ok) lmax = fmaxf(lmax, fmaxf(fabsf(d.x), fabsf(d.y)));
                        uint32_t e1s[4], pm = 0u;
                        if (L.hashed) {  // e(x + 1) = e(x) ^ flip, flip = the bits a +1 carry changes in cx (x enters with prime 1)
#if MI3D_TIMING_FAKE_PASS1
                            (void)cy; (void)cz;
                            const uint32_t yz[4] = {fk, fk ^ 0x2F3A5u, fk ^ 0x5C1B3u, fk ^ 0x73216u};
#else
                            const bool f24 = (L.res >> 22) == 0u;
                            const uint32_t hy = mul_prime(cy, kPrimeY, f24), hz = mul_prime(cz, kPrimeZ, f24), hy1 = hy + kPrimeY, hz1 = hz + kPrimeZ;
                            const uint32_t yz[4] = {hy ^ hz, hy1 ^ hz, hy ^ hz1, hy1 ^ hz1};
#endif
                            const uint32_t flip = (ccx[c] ^ (ccx[c] + 1u)) & mask;
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) { ce0[c][j] = (ccx[c] ^ yz[j]) & mask; e1s[j] = ce0[c][j] ^ flip; }
                            pm = (flip >> kBinShift) == 0u ? 0xFu : 0u;   // both entries in one bin: the same answer for all four
                        } else {
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) {
                                pair_entries(ccx[c], cy, cz, j, ce0[c][j], e1s[j]);
                                pm |= pairable(ce0[c][j], e1s[j]) ? 1u << j : 0u;
                            }
                        }
                        cpm[c] = ok ? pm : 0u;
                        if (has) {
                            const float gx = 1.0f - cfx[c], gy = 1.0f - cfy[c], gz = 1.0f - cfz[c];
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) {
                                const uint32_t e0 = ce0[c][j], e1 = e1s[j];
                                if ((cpm[c] >> j) & 1u) {
                                    __hip_atomic_fetch_add(&hist[e0 >> kBinShift], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                } else {
                                    // rare: a pair that straddles two bins leaves as two singles right away; a non-finite
                                    // gradient goes to the table with float atomics - on exactly the entries the
                                    // reference's atomicAdd would have put the inf / NaN
                                    const float w = ((j & 1u) ? cfy[c] : gy) * ((j >> 1) ? cfz[c] : gz);
                                    const float a = w * d.x, b = w * d.y;
                                    if (finite && HP) { put_single12(e0, gx * w, r0, d.x, d.y); put_single12(e1, cfx[c] * w, r0, d.x, d.y); }
                                    else if (finite) { put_single(e0, gx * a, gx * b); put_single(e1, cfx[c] * a, cfx[c] * b); }
                                    else { to_table(e0, gx * a, gx * b); to_table(e1, cfx[c] * a, cfx[c] * b); }
                                }
                            }
                        }
                    }
#ifdef MI3D_DEV  // tools build: 0x1000 = cells, entries and the bin histogram only (timing only)
                    if (fine_level_major & 0x1000u) { hist[lane] = 0u; continue; }
#endif
                    __builtin_amdgcn_wave_barrier();
                    // lane = bin: where the bin's run starts in the staging area and in its region
                    const uint32_t cnt = hist[lane], g = fill_l[lane];
                    // inclusive prefix sum over the 64 bins in registers: four DPP steps inside the 16-lane rows (zeros
                    // shift in at a row start), then the three row totals.  (Six __shfl_up = six dependent LDS round
                    // trips per chunk; this phase was 15.6 of the fine role's 44 ms, kbench.py --what scatter_diag.)
                    uint32_t incl = cnt;
                    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
                    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
                    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
                    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
                    {
                        const uint32_t r0 = __builtin_amdgcn_readlane(incl, 15), r1 = __builtin_amdgcn_readlane(incl, 31),
                                       r2 = __builtin_amdgcn_readlane(incl, 47);
                        incl += (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
                    }
                    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
                    const uint32_t excl = incl - cnt;
                    cursor[lane] = excl;
                    // binary16 records: the slot's index within the LEVEL's regions of this wave, bin base included - and a
                    // vote whether every bin's run fits its region, so that the flush of a chunk that fits (all but a
                    // skewed tile's) stores without a bound check and a multiply per record
                    const bool all_fit = __ballot(lane < (int)level_bins(L) && g + cnt > cap) == 0ull;
                    gdelta[lane] = (HP ? __umul24((uint32_t)lane, cap) : 0u) + g - excl;
                    if (cnt) fill_l[lane] = g + cnt;
                    hist[lane] = 0u;
                    __builtin_amdgcn_wave_barrier();
#ifdef MI3D_DEV  // tools build: 0x2000 = stop after the scan (timing only)
                    if (fine_level_major & 0x2000u) continue;
#endif
                    // pass 2: the records, each to its place
#pragma unroll
                    for (uint32_t c = 0; c < kChunkPts; ++c) {
                        if (cpm[c]) {
                            const float gy = 1.0f - cfy[c], gz = 1.0f - cfz[c];
                            const uint32_t t = L.hashed ? (uint32_t)__builtin_ctz(~ccx[c]) + 1u : 0u;
                            const uint32_t fq = HP ? fix23(cfx[c]) : 0u;   // (per point, not per record)
                            const uint32_t hdr_pt = (t << 13) | ((fq >> 16) << 25), lo_pt = (fq & 0xFFFFu) << 16;
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) {
                                if ((cpm[c] >> j) & 1u) {
                                    const uint32_t e0 = ce0[c][j];
                                    const float w = ((j & 1u) ? cfy[c] : gy) * ((j >> 1) ? cfz[c] : gz);
                                    const uint32_t pos = atomicAdd(&cursor[e0 >> kBinShift], 1u);
                                    if (HP) {   // Row12 + the bin in the fourth word (the staging slot stays 16 bytes: one LDS write)
                                        const uint32_t wq = fix23(w);   // (pack_row12, with the point's share hoisted)
                                        stage[pos] = make_uint4((e0 & (kBinEntries - 1u)) | hdr_pt | ((wq >> 16) << 18), craw0[c],
                                                                (wq & 0xFFFFu) | lo_pt, e0 >> kBinShift);
                                    } else
                                    stage[pos] = make_uint4(e0 | (t << kRowEntryBits), __float_as_uint(w * cd0[c]),
                                                            __float_as_uint(w * cd1[c]), __float_as_uint(cfx[c]));
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#ifdef MI3D_DEV  // tools build: 0x4000 = stop after the records are staged (timing only)
                    if (fine_level_major & 0x4000u) continue;
#endif
                    // the sorted records leave: consecutive lanes, consecutive slots of a region
                    for (uint32_t i = lane; i < total; i += kWave) {
                        const uint4 rec = stage[i];
                        if (HP) {
#if MI3D_TIMING_SEQ_FLUSH   // TIMING ONLY: the chunk's sorted records leave as ONE contiguous 6 KB page (see the switch's comment)
                            const uint32_t bin = rec.w, at = (ti * 7u + (p0 >> 1)) * 512u + i + 0u * gdelta[bin];
#else
                            const uint32_t bin = rec.w, at = i + gdelta[bin];   // = bin * cap + slot
#endif
                            if (all_fit) {
#ifdef MI3D_DEV
                                if (!(fine_level_major & 0x800u))
#endif
                                region12[at] = Row12{rec.x, rec.y, rec.z};
                                continue;
                            }
                            const uint32_t slot = at - __umul24(bin, cap);
                            if (slot < cap) {
#ifdef MI3D_DEV  // tools build: 0x800 = the records are sorted but not stored (timing only)
                                if (!(fine_level_major & 0x800u))
#endif
                                region12[__umul24(bin, cap) + slot] = Row12{rec.x, rec.y, rec.z};
                            } else {  // region full: straight to the table
                                uint32_t el, tt;
                                float w, fx, dx, dy;
                                unpack_row12(rec.x, rec.y, rec.z, el, tt, w, fx, dx, dy);
                                const uint32_t e0 = (bin << kBinShift) | el;
                                const uint32_t e1 = L.hashed ? (e0 ^ (((1u << tt) - 1u) & mask)) : e0 + 1u;
                                to_table(e0, (1.0f - fx) * (w * dx), (1.0f - fx) * (w * dy));
                                if (fx != 0.f) to_table(e1, fx * (w * dx), fx * (w * dy));
                            }
                            continue;
                        }
                        const uint32_t e0 = rec.x & ((1u << kRowEntryBits) - 1u), bin = e0 >> kBinShift;
                        const uint32_t slot = i + gdelta[bin];
                        if (slot < cap) {
#ifdef MI3D_DEV  // tools build: 0x800 = the records are sorted but not stored (timing only)
                            if (!(fine_level_major & 0x800u))
#endif
                            reinterpret_cast<uint4 *>(region0)[__umul24(bin, cap) + slot] = rec;
                        } else {  // region full: straight to the table
                            const float a = __uint_as_float(rec.y), b = __uint_as_float(rec.z), fx = __uint_as_float(rec.w);
                            const uint32_t tt = rec.x >> kRowEntryBits;
                            const uint32_t e1 = L.hashed ? (e0 ^ (((1u << tt) - 1u) & mask)) : e0 + 1u;
                            to_table(e0, (1.0f - fx) * a, (1.0f - fx) * b);
                            if (fx != 0.f) to_table(e1, fx * a, fx * b);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else if (merge && !dead) {
                // ---- coarse level.  The stencil points of one base position mostly share its cell: those are summed in
                // registers (acc0 / acc1, flushed into the gather table when the group ends); a point that left the
                // base cell goes to the gather table directly - and that path, 8 probes and 16 atomics, executes for
                // the whole wave as soon as ONE lane needs it, i.e. practically for every neighbour point.  So the +eps
                // and -eps neighbours of an axis are evaluated TOGETHER: below ~0.5 cells of epsilon a sample cannot
                // leave its cell on both sides, one gather serves the pair (a second one runs only if some lane did).
                auto eval_point = [&](uint32_t p, uint32_t &cx, uint32_t &cy, uint32_t &cz, float (&v0)[8], float (&v1)[8])
                    __attribute__((always_inline)) {
                    uint32_t r0 = raw0[0], r1 = raw1[0];
                    if (dyn_idx) {   // (p is uniform: one relative register move instead of a 12-step select chain)
                        const uint32_t pc = p < (uint32_t)kMaxPts ? p : 0u;
                        r0 = raw0[pc]; r1 = raw1[HP ? 0 : pc];
                    } else {
#pragma unroll
                    for (uint32_t k = 1; k < (uint32_t)kMaxPts; ++k) { r0 = (p == k) ? raw0[k] : r0; r1 = (p == k) ? raw1[HP ? 0 : k] : r1; }
                    }
                    float2 d = HP
                        ? make_float2((float)__builtin_bit_cast(_Float16, (unsigned short)(r0 & 0xFFFFu)),
                                      (float)__builtin_bit_cast(_Float16, (unsigned short)(r0 >> 16)))
                        : make_float2(__uint_as_float(r0), __uint_as_float(r1));
                    if (p == 0u) { d.x += ex0x; d.y += ex0y; }   // (uniform)
                    float q[3];
                    point_of(ps, base, p, q);
                    float fx, fy, fz;
                    grid_cell(q[0], L.scale, cx, fx);
                    grid_cell(q[1], L.scale, cy, fy);
                    grid_cell(q[2], L.scale, cz, fz);
                    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
                    const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;  // forward's product order
                    const float wk[8] = {w00 * gz, w10 * gz, w01 * gz, w11 * gz, w00 * fz, w10 * fz, w01 * fz, w11 * fz};
#pragma unroll
                    for (uint32_t k = 0; k < 8; ++k) { v0[k] = wk[k] * d.x; v1[k] = wk[k] * d.y; }
                    return valid && (d.x != 0.f || d.y != 0.f);
                };
                for (uint32_t p = 0; p < ps.P;) {
                    if (p == 0u || p == ps.P0) {  // a new group of stencil points: around x, then around x2
                        flush_group();
                        float qb[3];
                        const int which = p < ps.P0 ? 0 : 1;
#pragma unroll
                        for (int dd = 0; dd < 3; ++dd) {
                            if (ps.mode == 0) {
                                qb[dd] = base[which][dd];
                            } else {
                                const float w = clampf(base[which][dd], -ps.bound, ps.bound);
                                qb[dd] = ps.pow2b ? (w + ps.bound) * ps.inv2b : (w + ps.bound) / (2.0f * ps.bound);
                            }
                        }
                        float unused;
                        grid_cell(qb[0], L.scale, bx, unused);
                        grid_cell(qb[1], L.scale, by, unused);
                        grid_cell(qb[2], L.scale, bz, unused);
                    }
                    const bool paired = p != 0u && p + 1u < ps.P && p + 1u != ps.P0;
                    uint32_t ax, ay, az, cx2 = 0, cy2 = 0, cz2 = 0;
                    float a0[8], a1[8], b0[8], b1[8];
                    const bool has_a = eval_point(p, ax, ay, az, a0, a1);
                    const bool same_a = has_a && ax == bx && ay == by && az == bz;
                    bool has_b = false, same_b = false;
                    if (paired) {
                        has_b = eval_point(p + 1u, cx2, cy2, cz2, b0, b1);
                        same_b = has_b && cx2 == bx && cy2 == by && cz2 == bz;
                    }
                    if (__ballot(same_a || same_b) != 0ull) {
                        acc_any = true;
                        if (mask_fma) {
                            // (one fused multiply-add per value by a 0 / 1 lane mask instead of select + add: the emit is
                            // bound by vector-instruction issue, DESIGN.md 3.2''; every value here is finite - a tile with a
                            // non-finite gradient took the `dead` route - so x * 0 is 0)
                            const float ma = same_a ? 1.0f : 0.0f;
#pragma unroll
                            for (uint32_t k = 0; k < 8; ++k) { acc0[k] = fmaf(a0[k], ma, acc0[k]); acc1[k] = fmaf(a1[k], ma, acc1[k]); }
                            if (paired) {
                                const float mb = same_b ? 1.0f : 0.0f;
#pragma unroll
                                for (uint32_t k = 0; k < 8; ++k) { acc0[k] = fmaf(b0[k], mb, acc0[k]); acc1[k] = fmaf(b1[k], mb, acc1[k]); }
                            }
                        } else {
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) {
                            acc0[k] += (same_a ? a0[k] : 0.f) + (same_b ? b0[k] : 0.f);
                            acc1[k] += (same_a ? a1[k] : 0.f) + (same_b ? b1[k] : 0.f);
                        }
                        }
                    }
                    bool ex_a = has_a && !same_a, ex_b = has_b && !same_b;
                    // the pair is (+eps, -eps) along one axis (the reference's stencil, network_tcnn.py:117-122: +x -x +y -y
                    // +z -z): a point that left the base cell is in the cell next to it - see face_pass
                    int axis = -1;
                    if (paired && ps.mode == 1 && face_on) {
                        const float4 oa = ps.offs[p], ob = ps.offs[p + 1u];
                        const int nz = (oa.x != 0.f) + (oa.y != 0.f) + (oa.z != 0.f);
                        if (nz == 1 && ob.x == -oa.x && ob.y == -oa.y && ob.z == -oa.z && oa.x + oa.y + oa.z > 0.f)
                            axis = oa.x != 0.f ? 0 : (oa.y != 0.f ? 1 : 2);
                    }
                    if (axis >= 0 && __ballot(ex_a || ex_b) != 0ull) {
                        const uint32_t ca = axis == 0 ? ax : (axis == 1 ? ay : az), cb = axis == 0 ? cx2 : (axis == 1 ? cy2 : cz2);
                        const uint32_t cbase = axis == 0 ? bx : (axis == 1 ? by : bz);
                        const bool rest_a = (axis == 0 || ax == bx) && (axis == 1 || ay == by) && (axis == 2 || az == bz);
                        const bool rest_b = (axis == 0 || cx2 == bx) && (axis == 1 || cy2 == by) && (axis == 2 || cz2 == bz);
                        const bool fa = ex_a && rest_a && ca == cbase + 1u;
                        const bool fb = ex_b && rest_b && cb + 1u == cbase && !fa;   // (one far face per lane and pass)
                        if (__ballot(fa || fb) != 0ull) {
                            acc_any = true;
                            if (axis == 0) face_pass(std::integral_constant<uint32_t, 0>{}, fa, fb, a0, a1, b0, b1);
                            else if (axis == 1) face_pass(std::integral_constant<uint32_t, 1>{}, fa, fb, a0, a1, b0, b1);
                            else face_pass(std::integral_constant<uint32_t, 2>{}, fa, fb, a0, a1, b0, b1);
                            ex_a = ex_a && !fa;
                            ex_b = ex_b && !fb;
                        }
                    }
                    if (__ballot(ex_a || ex_b) != 0ull) {
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) {  // in place (registers): zeros are skipped by gather8
                            a0[k] = ex_a ? a0[k] : (ex_b ? b0[k] : 0.f);
                            a1[k] = ex_a ? a1[k] : (ex_b ? b1[k] : 0.f);
                        }
                        gather8(ex_a ? ax : cx2, ex_a ? ay : cy2, ex_a ? az : cz2, a0, a1);
                        const bool both = ex_a && ex_b;
                        if (__ballot(both) != 0ull) {
#pragma unroll
                            for (uint32_t k = 0; k < 8; ++k) { b0[k] = both ? b0[k] : 0.f; b1[k] = both ? b1[k] : 0.f; }
                            gather8(cx2, cy2, cz2, b0, b1);
                        }
                    }
                    p += paired ? 2u : 1u;
                }
            } else
            for (uint32_t p = 0; p < ps.P; ++p) {  // fine levels without the pair structure, and tiles with a non-finite gradient
                uint32_t r0 = raw0[0], r1 = raw1[0];
#pragma unroll
                for (uint32_t k = 1; k < (uint32_t)kMaxPts; ++k) { r0 = (p == k) ? raw0[k] : r0; r1 = (p == k) ? raw1[HP ? 0 : k] : r1; }
                float2 d = HP
                    ? make_float2((float)__builtin_bit_cast(_Float16, (unsigned short)(r0 & 0xFFFFu)),
                                  (float)__builtin_bit_cast(_Float16, (unsigned short)(r0 >> 16)))
                    : make_float2(__uint_as_float(r0), __uint_as_float(r1));
                if (p == 0u) { d.x += ex0x; d.y += ex0y; }   // (uniform)
                const bool has = valid && (d.x != 0.f || d.y != 0.f);
                const unsigned long long act = __ballot(has);
                if (act == 0ull) continue;
                float q[3];
                point_of(ps, base, p, q);
                uint32_t cx, cy, cz;
                float fx, fy, fz;
                grid_cell(q[0], L.scale, cx, fx);
                grid_cell(q[1], L.scale, cy, fy);
                grid_cell(q[2], L.scale, cz, fz);
                const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
                const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;  // forward's product order
                const float wk[8] = {w00 * gz, w10 * gz, w01 * gz, w11 * gz, w00 * fz, w10 * fz, w01 * fz, w11 * fz};
                if (!merge) {  // fine level without the pair structure: every point emits its 8 corners
                    // (appending the two 12-byte records of an x-pair together - one counter update, 24 contiguous
                    // bytes - was measured: 106 -> 114 ms per 141 M evaluations; the store path is paid per lane-store)
                    if (has) {
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) {
                            const uint32_t e = grid_entry(L, cx + (k & 1u), cy + ((k >> 1) & 1u), cz + (k >> 2));
                            emit_record(plan, L, l, gw, e, wk[k] * d.x, wk[k] * d.y, fill, arena, grad_table, lmax);
                        }
                    }
                    continue;
                }
                {   // (merge && dead) a tile with a non-finite gradient: plain float atomics, entry by entry
                    if (has) {
#pragma unroll 1
                        for (uint32_t k = 0; k < 8; ++k) {  // (rare path: kept rolled, it must not cost the kernel registers)
                            const float wkk = ((k & 1u) ? fx : gx) * ((k & 2u) ? fy : gy) * ((k & 4u) ? fz : gz);
                            float *dst = grad_table + ((size_t)L.offset + grid_entry(L, cx + (k & 1u), cy + ((k >> 1) & 1u), cz + (k >> 2))) * 2;
                            unsafeAtomicAdd(dst, wkk * d.x);
                            unsafeAtomicAdd(dst + 1, wkk * d.y);
                        }
                    }
                }
            }
            if (merge) flush_group();
            if (merge && merge_any) {  // one record per distinct entry the tile touched on this level; the table is left empty
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = lane; i < kMergeSlots; i += kWave) {
                    const uint32_t e = keys[i];
                    if (e != kMergeEmpty) {
                        const long long a0 = (long long)sums[2 * i], a1 = (long long)sums[2 * i + 1];
                        keys[i] = kMergeEmpty; sums[2 * i] = 0ull; sums[2 * i + 1] = 0ull;
                        if (a0 != 0 || a1 != 0)
                            emit_record(plan, L, l, gw, e, (float)((double)a0 * merge_unscale), (float)((double)a1 * merge_unscale),
                                        fill, arena, grad_table, lmax);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
            if (lane == 0) lmax_all[wave_in_wg][l] = fmaxf(lmax_all[wave_in_wg][l], lmax);
        }
    }
    for (uint32_t b = lane; b < plan.n_bins; b += kWave) {
        uint32_t lvl = 0;
        for (uint32_t l = 0; l < plan.n_levels; ++l)
            if (plan.level_bin0[l] <= b) lvl = l;
        if (!((level_mask >> lvl) & 1u)) continue;
        const uint32_t c = fill[b], cap = plan.level_cap[lvl], bins = level_bins(T.level[lvl]);
        counts[plan.level_cnt0[lvl] + (size_t)gw * bins + (b - plan.level_bin0[lvl])] = c < cap ? c : cap;
    }
    if (lane < MI3D_MAX_LEVELS && lane < (int)plan.n_levels && ((level_mask >> lane) & 1u))
        level_max[plan.level_max0[lane] + gw] = lmax_all[wave_in_wg][lane];
}

constexpr int kReduceWaves = (int)kReduceWavesC;  // 1024-thread workgroups, one per CU: 128 KB of LDS accumulators each

// Every record value is scaled by a power of two 2^k chosen from the largest |value| any wave emitted for the level
// (|v| 2^k < 2^38, so 2^24 records cannot overflow), rounded to an integer and added with ds_add_u64.  The scaling is
// exact; what is dropped is whatever lies more than 38 binary digits below the level's largest contribution - far
// below what fp32 running sums (24 digits) resolve.
__global__ __launch_bounds__(kWave *kReduceWaves) void k_bin_reduce(const char *__restrict__ arena,
                                                                     const uint32_t *__restrict__ counts,
                                                                     const float *__restrict__ level_max, GridTable T,
                                                                     BinPlan plan, float *__restrict__ grad_table) {
    extern __shared__ unsigned long long acc[];  // [2][kBinEntries]: feature-major, so one atomic instruction's 64
    // addresses spread over every 8-byte bank pair (entry-major pairs would leave half of them unused per instruction)
    __shared__ float wg_max[kReduceWaves];
    // level_split[l] workgroups share a bin of level l (each takes every split-th group of emitting waves), see
    // plan_reduce_splits; a small pass (the point-0 pass of the SDS backward) gets fewer: zeroing and flushing the 128 KB
    // tile is most of its work
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave_in_wg = threadIdx.x / kWave;
    uint32_t lvl = 0;  // which level does this workgroup belong to?  (uniform scan of at most 16 entries)
    for (uint32_t l = 0; l < T.n_levels; ++l)
        if (plan.level_wg0[l] <= blockIdx.x) lvl = l;
    const uint32_t n_split = plan.level_split[lvl];
    const uint32_t lb = (blockIdx.x - plan.level_wg0[lvl]) / n_split, split = (blockIdx.x - plan.level_wg0[lvl]) % n_split;
    const uint32_t cap = plan.level_cap[lvl], n_waves = plan.level_waves[lvl], bins = level_bins(T.level[lvl]);
    for (uint32_t i = threadIdx.x; i < kBinEntries * 2; i += blockDim.x) acc[i] = 0ull;
    const bool row = (plan.row_mask >> lvl) & 1u;
    float scale;
    double unscale;
    {
        float m = 0.f;
        for (uint32_t r = threadIdx.x; r < n_waves; r += blockDim.x) m = fmaxf(m, level_max[plan.level_max0[lvl] + r]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (lane == 0) wg_max[wave_in_wg] = m;
        __syncthreads();
        m = 0.f;
        for (int w = 0; w < kReduceWaves; ++w) m = fmaxf(m, wg_max[w]);
        if (!(m > 0.f)) return;  // nothing but zeros was emitted for this level (uniform across the workgroup)
        int e;
        (void)frexpf(m, &e);  // m < 2^e
        int k = 38 - e;
        k = k > 126 ? 126 : (k < -126 ? -126 : k);
        scale = ldexpf(1.0f, k);
        unscale = ldexp(1.0, -k);
    }

    bool any = false;
#ifndef MI3D_REDUCE_U
// records in flight per lane.  Round 4 swept 4 / 8 / 16 and found no difference - there was none to find: the loads were
// predicated (`i < cnt ? src[i] : 0`), hipcc waited for each on the spot (eight `global_load_dwordx3` + `s_waitcnt vmcnt(0)`
// pairs in a row), nothing was ever in flight.  With unconditional loads from a clamped index they are issued back to
// back (round 5, read off the ISA).  Product-grade builds in one process (tools/scatter_ab_libs.py,
// profiles/scatter_ab_libs_r05_reduce.json): predicated 8: dense 51.13 / real 40.03 ms; in flight 8: 50.67 / 40.07; in
// flight 16: 51.6 / 41.1; in flight 4: **50.25 / 39.8** - the reduce is bound by its LDS atomics, not by the latency of
// its loads (16 waves per CU hid that already): a small gain, and the short batch wins.
#define MI3D_REDUCE_U 4
#endif
#ifndef MI3D_TIMING_REDUCE_NO_FLUSH
// TIMING ONLY: a reduce workgroup ends with up to 16 384 float atomics into the table (49 M per slice, at the chip's 21 G/s that
// would be 2.3 ms).  Without them (round 6, product builds in one process, three slices: profiles/scatter_ab_libs_r06_reduce_flush_33GiB.json)
// dense 47.51 -> 47.21 ms, captured real step 32.70 -> 32.65: they cost nothing - the wave ends behind them, the next workgroup's
// LDS atomics run meanwhile.  (2 / 3 workgroups per bin instead of 4, same file: nothing either.)
#define MI3D_TIMING_REDUCE_NO_FLUSH 0
#endif
#ifndef MI3D_REDUCE_BASE_SPLIT
// reduce workgroups per average bin for slices of 30 M evaluations and more (plan_reduce_splits).  Round 6, product-grade builds
// in one process (profiles/scatter_ab_libs_r06_reduce_split.json; dense / synthetic census / captured real step, ms): 3: 44.78 /
// 38.24 / 32.21, 4: 44.85 / 38.25 / 32.28, 6: 44.93 / 38.39 / 32.27, 8: 44.96 / 38.47 / 32.45 - flat: the reduce's tail is not where
// its time goes
#define MI3D_REDUCE_BASE_SPLIT 4
#endif
#ifndef MI3D_REDUCE_PIPE
// the next batch's record loads issued BEFORE this batch's LDS atomics (two register sets, loop unrolled by two; the ISA
// shows the older batch waited for with the newer one in flight).  Round 6, product-grade builds in one process
// (profiles/scatter_ab_libs_r06_reduce_pipe.json; dense / synthetic census / captured real step): off 46.34 / 37.04 / 31.88 ms,
// on 46.17 / 36.97 / 32.05, on with batches of 2: 46.14 / 36.87 / 31.82, of 8: 46.33 / 37.20 / 32.19 - nothing: the reduce does
// not wait for its loads, it is bound by the LDS atomics themselves.  Off.
#define MI3D_REDUCE_PIPE 0
#endif
    constexpr uint32_t U = MI3D_REDUCE_U;  // records per lane and batch
    // A region's records in batches of U per lane.  MI3D_REDUCE_PIPE: the NEXT batch's loads are issued before this batch's
    // LDS atomics (the loads come back in order, so the wave waits for the older batch only): a wave that loads, waits ~2 us
    // and then issues its 16 atomic instructions leaves the LDS idle whenever the CU's 16 waves happen to wait together.
    auto stream = [&](uint32_t cnt, auto load, auto add) __attribute__((always_inline)) {
        if (cnt == 0u) return;
        using Rec = decltype(load(0u));
        if (MI3D_REDUCE_PIPE) {
            // two register sets, the loop unrolled by two so that neither is ever copied (a copy would wait for its loads);
            // every load is unconditional from a clamped index (a conditional one is waited for where its block ends)
            Rec ra[U], rb[U];
            auto fetch = [&](Rec (&dst)[U], uint32_t at) __attribute__((always_inline)) {
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) { const uint32_t i = at + u * kWave + lane; dst[u] = load(i < cnt ? i : cnt - 1u); }
            };
            auto consume = [&](const Rec (&src)[U], uint32_t at) __attribute__((always_inline)) {
#pragma unroll
                for (uint32_t u = 0; u < U; ++u)
                    if (at + u * kWave + lane < cnt) add(src[u]);
            };
            constexpr uint32_t B = kWave * U;
            fetch(ra, 0u);
            for (uint32_t i0 = 0; i0 < cnt; i0 += 2u * B) {
                fetch(rb, i0 + B);
                consume(ra, i0);
                fetch(ra, i0 + 2u * B);
                consume(rb, i0 + B);
            }
        } else {
            for (uint32_t i0 = 0; i0 < cnt; i0 += kWave * U) {
                Rec rec[U];
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {
                    // (unconditional, from a clamped index: a predicated load makes hipcc wait for it on the spot - the eight
                    // records "in flight" were eight HBM round trips in a row, each behind its own s_waitcnt vmcnt(0), until
                    // round 5 read the ISA; what a lane beyond cnt fetched is never looked at)
                    const uint32_t i = i0 + u * kWave + lane;
                    rec[u] = load(i < cnt ? i : cnt - 1u);
                }
#pragma unroll
                for (uint32_t u = 0; u < U; ++u)
                    if (i0 + u * kWave + lane < cnt) add(rec[u]);
            }
        }
    };
    for (uint32_t r = split * kReduceWaves + wave_in_wg; r < n_waves; r += n_split * kReduceWaves) {
        uint32_t cnt = counts[plan.level_cnt0[lvl] + (size_t)r * bins + lb];
        cnt = cnt < cap ? cnt : cap;
        if (row && plan.rec12) {   // binary16 gradient planes: 12-byte pair records
            const Row12 *src = reinterpret_cast<const Row12 *>(arena + plan.level_base[lvl]) + ((size_t)r * bins + lb) * cap;
            const bool hashed = T.level[lvl].hashed;
            stream(cnt, [&](uint32_t i) { return src[i]; }, [&](const Row12 &rec) {
                uint32_t l0, t;
                float w, fx, dx, dy;
                unpack_row12(rec.w0, rec.w1, rec.w2, l0, t, w, fx, dx, dy);
                // (the level's power-of-two scale goes into the weight once - an exact scaling, it commutes with
                // every rounding below - instead of into each of the four products)
                const float ws = w * scale, a = ws * dx, bb = ws * dy, gx = 1.0f - fx;
                atomicAdd(&acc[l0], fixed_point(gx * a));
                atomicAdd(&acc[kBinEntries + l0], fixed_point(gx * bb));
                if (fx != 0.f) {  // a pair: the x + 1 corner sits in the same bin (singles carry fx = 0)
                    // (the flip is masked with the level's size first, as the emit and the 16-byte branch do: a
                    // table smaller than a bin - log2_hashmap_size < 13 - must not see carry bits beyond it)
                    const uint32_t l1 = hashed ? (l0 ^ (((1u << t) - 1u) & (T.level[lvl].size - 1u))) & (kBinEntries - 1u)
                                               : l0 + 1u;
                    atomicAdd(&acc[l1], fixed_point(fx * a));
                    atomicAdd(&acc[kBinEntries + l1], fixed_point(fx * bb));
                }
            });
        } else if (row) {
            const uint4 *src = reinterpret_cast<const uint4 *>(arena + plan.level_base[lvl]) + ((size_t)r * bins + lb) * cap;
            const GridLevel &L = T.level[lvl];
            stream(cnt, [&](uint32_t i) { return src[i]; }, [&](const uint4 &rec) {
                const uint32_t e0 = rec.x & ((1u << kRowEntryBits) - 1u), t = rec.x >> kRowEntryBits;
                const float a = __uint_as_float(rec.y), bb = __uint_as_float(rec.z), fx = __uint_as_float(rec.w);
                const float gx = 1.0f - fx;
                const uint32_t l0 = e0 & (kBinEntries - 1);
                atomicAdd(&acc[l0], fixed_point((gx * a) * scale));
                atomicAdd(&acc[kBinEntries + l0], fixed_point((gx * bb) * scale));
                if (fx != 0.f) {  // a pair: the x + 1 corner sits in the same bin (singles carry fx = 0)
                    const uint32_t e1 = L.hashed ? (e0 ^ (((1u << t) - 1u) & (L.size - 1u))) : e0 + 1u;
                    const uint32_t l1 = e1 & (kBinEntries - 1);
                    atomicAdd(&acc[l1], fixed_point((fx * a) * scale));
                    atomicAdd(&acc[kBinEntries + l1], fixed_point((fx * bb) * scale));
                }
            });
        } else {
            const BinRecord *src = reinterpret_cast<const BinRecord *>(arena + plan.level_base[lvl]) + ((size_t)r * bins + lb) * cap;
            stream(cnt, [&](uint32_t i) { return src[i]; }, [&](const BinRecord &rec) {
                const uint32_t local = rec.entry & (kBinEntries - 1);
                atomicAdd(&acc[local], fixed_point(rec.g0 * scale));
                atomicAdd(&acc[kBinEntries + local], fixed_point(rec.g1 * scale));
            });
        }
        any |= cnt != 0;
    }
    if (!__syncthreads_or((int)any)) return;
    const uint32_t e0 = lb * kBinEntries;
    const uint32_t live = T.level[lvl].size - e0 < kBinEntries ? T.level[lvl].size - e0 : kBinEntries;  // last bin of a level
    float *dst = grad_table + ((size_t)T.level[lvl].offset + e0) * 2;
#if MI3D_TIMING_REDUCE_NO_FLUSH   // TIMING ONLY (the gradient is lost): what the float atomics that end a reduce workgroup cost
    if (live == 0xFFFFFFFFu) dst[threadIdx.x] = (float)((double)(long long)acc[threadIdx.x] * unscale);
#else
    for (uint32_t i = threadIdx.x; i < live * 2; i += blockDim.x) {
        const long long a = (long long)acc[(i & 1u) * kBinEntries + (i >> 1)];
        if (a != 0) unsafeAtomicAdd(dst + i, (float)((double)a * unscale));
    }
#endif
}

// grad[i] += sum over the n_rep private copies
__global__ void k_replica_reduce(const float *__restrict__ rep, uint32_t n_rep, size_t rep_stride, uint32_t count,
                                 float *__restrict__ grad_table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (uint32_t r = 0; r < n_rep; ++r) s += rep[(size_t)r * rep_stride + i];
    if (s != 0.f) grad_table[i] += s;
}

constexpr uint32_t kReplicasDefault = 64;  // private table copies of the atomic fallback (same-line serialisation)

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

int mi3d_grid_encode_points_planes(const float *x, const float *x2, uint32_t n, const float *offsets_host, uint32_t P0,
                                   uint32_t P, float bound, const float *params, uint32_t n_levels,
                                   uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size, float step,
                                   void *out_planes, int out_half, void *stream) {
    return mi3d_grid_encode_points_planes_counted(x, x2, n, nullptr, offsets_host, P0, P, bound, params, n_levels,
                                                  base_resolution, per_level_scale, log2_hashmap_size, step, out_planes,
                                                  out_half, stream);
}

int mi3d_grid_encode_points_planes_counted(const float *x, const float *x2, uint32_t n, const int32_t *count,
                                           const float *offsets_host, uint32_t P0, uint32_t P, float bound,
                                           const float *params, uint32_t n_levels, uint32_t base_resolution,
                                           float per_level_scale, uint32_t log2_hashmap_size, float step, void *out_planes,
                                           int out_half, void *stream) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || P == 0 || P > MI3D_MAX_POINTS || P0 > P ||
        (P0 < P && x2 == nullptr))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const PointSet ps = make_points(x, x2, offsets_host, P0, P, bound, 1);
    const uint32_t tiles = (n + kTile - 1) / kTile;
    const float step01 = step > 0.f ? step / (2.0f * bound) : 1.0f / 512.0f;
    const int variant = MI3D_TUNE(MI3D_T_ENCODE_VARIANT, 3);
    uint32_t need = (tiles + kWaves - 1) / kWaves;  // workgroups one XCD needs to give every tile its own wave
    const uint32_t fine_cu = (uint32_t)MI3D_TUNE(MI3D_T_ENCODE_WGS_PER_CU, 3);
    const uint32_t coarse_cu = (uint32_t)MI3D_TUNE(MI3D_T_ENCODE_COARSE_WGS_PER_CU, 6);
    const uint32_t wgs_fine = need < 32 * fine_cu ? need : 32 * fine_cu;        // 32 CUs per XCD; persistent beyond
    uint32_t wgs_coarse = need < 32 * coarse_cu ? need : 32 * coarse_cu;
    if (wgs_coarse < wgs_fine) wgs_coarse = wgs_fine;
    const float2 *tab = reinterpret_cast<const float2 *>(params);
    float *out = reinterpret_cast<float *>(out_planes);
    hipStream_t st = as_stream(stream);
    // the longest prefix of levels whose tables fit the LDS of a CU together goes through the LDS kernel
    const int only_level = MI3D_TUNE(MI3D_T_ENCODE_ONLY_LEVEL, -1);
    uint32_t n_lds = 0;
    size_t lds_bytes = 0;
    while (n_lds < T.n_levels && (size_t)(T.level[n_lds].offset + T.level[n_lds].size) * sizeof(float2) <= (size_t)150 * 1024) {
        lds_bytes = (size_t)(T.level[n_lds].offset + T.level[n_lds].size) * sizeof(float2);
        ++n_lds;
    }
    if (MI3D_TUNE(MI3D_T_ENCODE_LDS_LEVELS, -1) >= 0) {  // dev: force the count (0 = off)
        n_lds = (uint32_t)MI3D_TUNE(MI3D_T_ENCODE_LDS_LEVELS, -1) < n_lds ? (uint32_t)MI3D_TUNE(MI3D_T_ENCODE_LDS_LEVELS, -1) : n_lds;
        lds_bytes = n_lds ? (size_t)(T.level[n_lds - 1].offset + T.level[n_lds - 1].size) * sizeof(float2) : 0;
    }
    if (only_level >= 0) n_lds = 0;
    if (n_lds > 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grid_encode_planes_lds<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const uint32_t wgs = (tiles + kLdsWaves - 1) / kLdsWaves;
        hipLaunchKernelGGL((k_grid_encode_planes_lds<true>), dim3(wgs < 256u ? wgs : 256u), dim3(kWave * kLdsWaves), lds_bytes,
                           st, ps, n, tab, T, n_lds, out, out_half, count);
        if (n_lds == T.n_levels) return (int)hipGetLastError();
    }
    // The launch (or launches) of the XCD-planned kernel.  With the x-group evaluation (encode_group_hash, TRIPLE) the fine
    // levels - hashed, x = cells per marching step >= 0.30: 3 workgroups per CU anyway - go to a kernel instance of their
    // own: holding three points' rows in flight takes ~2 x the registers, which the coarse levels' 6 workgroups per CU
    // cannot afford.  Each launch balances its own levels over the XCDs and claims tiles from its own counter slot.
    uint32_t fine_mask = 0u;
    for (uint32_t l = n_lds; l < T.n_levels; ++l) {
        const double x = (double)step01 * (double)T.level[l].scale;
        if (level_fast(T.level[l], 1).kind == kHashPow2 && encode_level_wgs_per_cu(x, 6u, 3u) == 3u) fine_mask |= 1u << l;
    }
    const int triple = MI3D_TUNE(MI3D_T_ENCODE_TRIPLE, MI3D_ENCODE_TRIPLE);
    const bool split = triple != 0 && fine_mask != 0u && (variant & 3) == 3;
    // One counter slot per launch, a ring of 64, zeroed in-stream right before the kernel; a hipEvent recorded behind the
    // kernel says when the slot is free again: a slot is handed out only once that event has completed (whatever the
    // stream) - otherwise the launch leaves the slot alone and deals
    // its tiles statically (next == nullptr: the round-2 order, same planes, a few ms slower at C2 size) - never two live
    // launches on one set of counters.  (Round 4 asked the previous owner's STREAM with hipStreamQuery: a destroyed
    // stream's handle is not a valid argument, and a launch captured into a hipGraph recorded the idle capture stream as
    // owner while the graph later replays elsewhere - ADVICE round 4.)  A launch that is being CAPTURED always deals
    // statically: a graph replays any number of times on any stream, it must not own a slot.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
    const bool claim_tiles = MI3D_TUNE(MI3D_T_ENCODE_STATIC_TILES, 0) == 0 && capturing == hipStreamCaptureStatusNone;
    auto acquire_slot = [&](uint32_t *&next, int &slot_used) -> int {
        next = nullptr;
        slot_used = -1;
        if (!claim_tiles) return 0;
        uint32_t *base = nullptr;
        hipError_t e = hipGetSymbolAddress(reinterpret_cast<void **>(&base), HIP_SYMBOL(g_encode_next));
        if (e != hipSuccess) return (int)e;
        std::lock_guard<std::mutex> guard(g_slot_mutex);
        const uint32_t slot = g_slot_launches++ % kPlanSlots;
        PlanSlot &ps_slot = g_slots[slot];
        // free = never used, or the event recorded behind its last launch has completed.  (Round 5 also took a slot whose
        // last user was "this stream" by comparing raw handles: a destroyed stream's address can come back as another
        // stream's while the old launch is still running - ADVICE round 5.  The event is the only witness now; a slot that
        // comes round on a stream with 64 gathers still in flight deals statically instead.)
        bool free_now = !ps_slot.used;
        if (!free_now && !ps_slot.broken && ps_slot.done != nullptr) {   // (hipEventQuery does not block; hipErrorNotReady is not a failure)
            free_now = hipEventQuery(ps_slot.done) == hipSuccess;
            if (!free_now) (void)hipGetLastError();
        }
        int dev_now = -1;
        (void)hipGetDevice(&dev_now);
        if (free_now && ps_slot.done != nullptr && ps_slot.device != dev_now) {
            // (an event belongs to the device it was created on: the ring is shared by the process's devices)
            free_now = hipEventQuery(ps_slot.done) == hipSuccess;
            if (free_now) { (void)hipEventDestroy(ps_slot.done); ps_slot.done = nullptr; } else (void)hipGetLastError();
        }
        if (free_now && ps_slot.done == nullptr) {
            if (hipEventCreateWithFlags(&ps_slot.done, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                ps_slot.done = nullptr;
                free_now = false;
            }
            ps_slot.device = dev_now;
        }
        if (free_now) {
            ps_slot.used = true;
            next = base + (size_t)slot * kXcds * kMaxSegs;
            e = hipMemsetAsync(next, 0, sizeof(uint32_t) * kXcds * kMaxSegs, st);
            if (e != hipSuccess) return (int)e;
            slot_used = (int)slot;
        }
        return 0;
    };
    auto slot_done = [&](int slot_used) {   // behind the kernel: the slot's counters are free once this event has completed
        if (slot_used >= 0) {
            std::lock_guard<std::mutex> guard(g_slot_mutex);
            if (hipEventRecord(g_slots[slot_used].done, st) != hipSuccess) {
                (void)hipGetLastError();
                g_slots[slot_used].broken = true;   // a stale event would report "complete" while the kernel still runs
            }
        }
    };
    const dim3 block(kWave * kWaves);
    int rc = 0;
    for (int part = 0; part < (split ? 2 : 1) && rc == 0; ++part) {
        const uint32_t mask = split ? (part == 0 ? ~fine_mask : fine_mask) : 0xFFFFFFFFu;
        const EncodePlan plan = make_encode_plan(T, tiles, step01, only_level, wgs_coarse, wgs_fine, n_lds, 1, mask);
        uint32_t segs = 0;
        for (uint32_t x = 0; x < kXcds; ++x) segs += plan.n_seg[x];
        if (segs == 0) continue;   // (no level of this class)
        const bool fine_part = split && part == 1;
        const dim3 grid((fine_part ? wgs_fine : wgs_coarse) * kXcds);
        uint32_t *next = nullptr;
        int slot_used = -1;
        rc = acquire_slot(next, slot_used);
        if (rc != 0) break;
#if defined(MI3D_DEV) || MI3D_ENCODE_TRIPLE   // (the product is built without the instance: measured, not taken - see MI3D_ENCODE_TRIPLE)
        if (fine_part)
            hipLaunchKernelGGL((k_grid_encode_planes<true, true, true>), grid, block, 0, st, ps, n, tab, T, plan, out, out_half, count, next);
        else
#endif
        if ((variant & 3) == 3)
            hipLaunchKernelGGL((k_grid_encode_planes<true, true>), grid, block, 0, st, ps, n, tab, T, plan, out, out_half, count, next);
        else if (variant & 1)
            hipLaunchKernelGGL((k_grid_encode_planes<true, false>), grid, block, 0, st, ps, n, tab, T, plan, out, out_half, count, next);
        else
            hipLaunchKernelGGL((k_grid_encode_planes<false, false>), grid, block, 0, st, ps, n, tab, T, plan, out, out_half, count, next);
        rc = (int)hipGetLastError();
        slot_done(slot_used);
    }
    return rc;
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

size_t mi3d_grid_scatter_binned_workspace(uint32_t n, uint32_t P, float bound, float step, uint32_t n_levels,
                                          uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || n == 0) return 0;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const float step01 = step > 0.f ? step / (2.0f * bound) : 1.0f / 512.0f;
    return bin_workspace_bytes(plan_for(T, n, P, step01, default_merge_levels(T, step01 * merge_steps())));
}

int mi3d_grid_scatter_binned(const float *x, const float *x2, uint32_t n, const float *offsets_host, uint32_t P0,
                             uint32_t P, float bound, const void *dout_planes_v, int dout_half, uint32_t n_levels,
                             uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size, float step,
                             void *workspace, size_t workspace_bytes, float *grad_params, void *stream) {
    return mi3d_grid_scatter_binned_plus(x, x2, n, offsets_host, P0, P, bound, dout_planes_v, nullptr, dout_half, n_levels,
                                         base_resolution, per_level_scale, log2_hashmap_size, step, workspace,
                                         workspace_bytes, grad_params, stream);
}

int mi3d_grid_scatter_binned_plus(const float *x, const float *x2, uint32_t n, const float *offsets_host, uint32_t P0,
                                  uint32_t P, float bound, const void *dout_planes_v, const void *extra_point0_planes,
                                  int dout_half, uint32_t n_levels, uint32_t base_resolution, float per_level_scale,
                                  uint32_t log2_hashmap_size, float step, void *workspace, size_t workspace_bytes,
                                  float *grad_params, void *stream) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || P == 0 || P > MI3D_MAX_POINTS || P0 > P ||
        (P0 < P && x2 == nullptr))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipStream_t st = as_stream(stream);
    const float *dout_planes = reinterpret_cast<const float *>(dout_planes_v);
    const float *extra0 = reinterpret_cast<const float *>(extra_point0_planes);
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const PointSet ps = make_points(x, x2, offsets_host, P0, P, bound, 1);
    const float step01 = step > 0.f ? step / (2.0f * bound) : 1.0f / 512.0f;
    // the record path run-merges only where it pays (cells at least 3 marching steps long); levels with shorter runs
    // emit per-point records (16-byte x-pair records where the level's entries allow it)
    const uint32_t merge_atomic = default_merge_levels(T, step01);
    const uint32_t merge_levels = default_merge_levels(T, step01 * merge_steps());
    const uint32_t plane_rows = n * P;

    // the slice: the samples are cut into the FEWEST equal slices whose record arena fits the workspace
    const uint32_t P_rec = P + ((extra0 != nullptr && dout_half) ? 1u : 0u);   // (binary16: the extra pair is a record of its own)
    BinPlan plan;
    const uint64_t n_slice = slice_for(T, n, P_rec, step01, merge_levels, dout_half != 0, workspace_bytes, plan);
    if (workspace == nullptr || bin_workspace_bytes(plan) > workspace_bytes) {
        // no usable workspace: the atomic kernels, with private copies of the table against same-line serialisation
        // when the caller's scratch at least holds those
        const uint32_t kReplicas = (uint32_t)MI3D_TUNE(MI3D_T_REPLICAS, kReplicasDefault);
        const size_t rep_bytes = (size_t)kReplicas * T.n_entries * 2 * sizeof(float);
        if (workspace != nullptr && workspace_bytes >= rep_bytes && rep_bytes < ((size_t)2 << 30)) {
            float *rep = reinterpret_cast<float *>(workspace);
            (void)hipMemsetAsync(rep, 0, rep_bytes, st);
            int err = launch_scatter(ps, n, nullptr, dout_planes, T, merge_atomic, rep, st, 0xFFFFFFFFu, plane_rows,
                                     kReplicas, (size_t)T.n_entries * 2, dout_half);
            if (!err && extra0 != nullptr)   // (the atomic kernels take one pair per point: point 0 again, on its own)
                err = launch_scatter(make_points(x, nullptr, offsets_host, 1, 1, bound, 1), n, nullptr, extra0, T,
                                     merge_atomic, rep, st, 0xFFFFFFFFu, n, kReplicas, (size_t)T.n_entries * 2, dout_half);
            hipLaunchKernelGGL(k_replica_reduce, dim3((T.n_entries * 2 + 255) / 256), dim3(256), 0, st, rep, kReplicas,
                               (size_t)T.n_entries * 2, T.n_entries * 2, grad_params);
            return err ? err : (int)hipGetLastError();
        }
        const int err = launch_scatter(ps, n, nullptr, dout_planes, T, merge_atomic, grad_params, st, 0xFFFFFFFFu,
                                       plane_rows, 1, 0, dout_half);
        if (err || extra0 == nullptr) return err;
        return launch_scatter(make_points(x, nullptr, offsets_host, 1, 1, bound, 1), n, nullptr, extra0, T, merge_atomic,
                              grad_params, st, 0xFFFFFFFFu, n, 1, 0, dout_half);
    }
    char *arena = reinterpret_cast<char *>(workspace);
    uint32_t *counts = reinterpret_cast<uint32_t *>(arena + plan.total_bytes);
    float *level_max = reinterpret_cast<float *>(counts + plan.total_counts);
    const size_t lds = (size_t)kWaves * emit_wave_words(plan.n_bins) * sizeof(uint32_t);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(dout_half ? reinterpret_cast<const void *>(k_bin_emit<true>)
                                            : reinterpret_cast<const void *>(k_bin_emit<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t lds_reduce = (size_t)kBinEntries * 2 * sizeof(unsigned long long);
    // per call: the attribute is per device and the call is a host-side table write (no static, re-entrant)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_reduce), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_reduce);
    // the fine levels (>= merge_levels) and the coarse ones are two roles of one emit launch
    const uint32_t all = (uint32_t)((1ull << T.n_levels) - 1);
    uint32_t coarse_mask = merge_levels ? (all & ((1u << merge_levels) - 1u)) : 0u;
    const uint32_t dev_mask = (uint32_t)MI3D_TUNE(MI3D_T_SCATTER_LEVEL_MASK, 0x7FFFFFFF);  // per-role timing (dev build)
    const uint32_t coarse_mask_all = coarse_mask;
    const uint32_t fine_mask = all & ~coarse_mask_all & dev_mask;
    coarse_mask &= dev_mask;
    const uint32_t fine_waves = fine_mask ? plan.level_waves[__builtin_ctz(fine_mask)] : 0u;
    const uint32_t coarse_waves = coarse_mask ? plan.level_waves[__builtin_ctz(coarse_mask)] : 0u;
    const uint32_t emit_order = (uint32_t)MI3D_TUNE(MI3D_T_EMIT_ORDER, MI3D_EMIT_ORDER_DEFAULT);
    for (uint64_t s0 = 0; s0 < n; s0 += n_slice) {
        const uint32_t s1 = (uint32_t)((s0 + n_slice < n) ? s0 + n_slice : n);
        if (fine_waves + coarse_waves)
        {
            const dim3 eg((fine_waves + coarse_waves) / kWaves), eb(kWave * kWaves);
            (void)hipMemsetAsync(level_max + plan.claim0, 0, 4 * sizeof(uint32_t), st);   // the roles' tile-claim counters
            if (dout_half)
                hipLaunchKernelGGL(k_bin_emit<true>, eg, eb, lds, st, ps, (uint32_t)s0, s1, dout_planes, plane_rows, n, extra0, T,
                                   plan, merge_levels, fine_mask, fine_waves, coarse_mask, coarse_waves, emit_order,
                                   reinterpret_cast<BinRecord *>(arena), counts, level_max, grad_params);
            else
                hipLaunchKernelGGL(k_bin_emit<false>, eg, eb, lds, st, ps, (uint32_t)s0, s1, dout_planes, plane_rows, n, extra0, T,
                                   plan, merge_levels, fine_mask, fine_waves, coarse_mask, coarse_waves, emit_order,
                                   reinterpret_cast<BinRecord *>(arena), counts, level_max, grad_params);
        }
        const uint32_t n_split = (uint64_t)(s1 - s0) * P >= 30000000ull ? (uint32_t)MI3D_REDUCE_BASE_SPLIT : ((uint64_t)(s1 - s0) * P >= 8000000ull ? 2u : 1u);
        plan_reduce_splits(plan, T, n_split, merge_levels);
        hipLaunchKernelGGL(k_bin_reduce, dim3(plan.n_reduce_wgs), dim3(kWave * kReduceWaves), lds_reduce, st, arena,
                           counts, level_max, T, plan, grad_params);
    }
    return (int)hipGetLastError();
}


// Host-side planning queries (no device work; they run without a GPU): what the two calls above will do.
int mi3d_grid_level_routes(uint32_t n_levels, uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size,
                           int stencil_points, int32_t *kinds) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || kinds == nullptr) return (int)hipErrorInvalidValue;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    for (uint32_t l = 0; l < T.n_levels; ++l) kinds[l] = level_fast(T.level[l], stencil_points ? 1 : 0).kind;
    return 0;
}

int mi3d_grid_encode_plan(uint32_t n, float bound, float step, uint32_t n_levels, uint32_t base_resolution,
                          float per_level_scale, uint32_t log2_hashmap_size, uint32_t *n_segments, uint32_t *segments) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || n_segments == nullptr || segments == nullptr)
        return (int)hipErrorInvalidValue;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const float step01 = step > 0.f ? step / (2.0f * bound) : 1.0f / 512.0f;
    uint32_t n_lds = 0;  // as mi3d_grid_encode_points_planes: the levels served from LDS are not in the plan
    while (n_lds < T.n_levels && (size_t)(T.level[n_lds].offset + T.level[n_lds].size) * sizeof(float2) <= (size_t)150 * 1024) ++n_lds;
    if (n_lds == T.n_levels) n_lds = 0;
    const EncodePlan plan = make_encode_plan(T, (n + kTile - 1) / kTile, step01, -1, 32 * 6, 32 * 3, n_lds);
    for (uint32_t x = 0; x < kXcds; ++x) {
        n_segments[x] = plan.n_seg[x];
        for (uint32_t i = 0; i < (uint32_t)kMaxSegs; ++i) {
            const EncodeSeg sg = i < plan.n_seg[x] ? plan.seg[x][i] : EncodeSeg{0u, 0u, 0u, 0u};
            segments[(x * kMaxSegs + i) * 3 + 0] = sg.level;
            segments[(x * kMaxSegs + i) * 3 + 1] = sg.tile0;
            segments[(x * kMaxSegs + i) * 3 + 2] = sg.tile1;
        }
    }
    return 0;
}

int mi3d_grid_scatter_plan(uint32_t n, uint32_t P, float bound, float step, uint32_t n_levels, uint32_t base_resolution,
                           float per_level_scale, uint32_t log2_hashmap_size, size_t workspace_bytes,
                           unsigned long long *out) {
    if (n_levels == 0 || n_levels > MI3D_MAX_LEVELS || P == 0 || P > MI3D_MAX_POINTS || n == 0 || out == nullptr)
        return (int)hipErrorInvalidValue;
    GridTable T;
    build_grid_table(T, n_levels, base_resolution, per_level_scale, log2_hashmap_size);
    const float step01 = step > 0.f ? step / (2.0f * bound) : 1.0f / 512.0f;
    const uint32_t merge_levels = default_merge_levels(T, step01 * merge_steps());
    BinPlan p;
    const uint64_t n_slice = slice_for(T, n, P, step01, merge_levels, false, workspace_bytes, p);  // as mi3d_grid_scatter_binned
    const uint64_t evals = n_slice * P;
    plan_reduce_splits(p, T, evals >= 30000000ull ? (uint32_t)MI3D_REDUCE_BASE_SPLIT : (evals >= 8000000ull ? 2u : 1u), merge_levels);
    out[0] = n_slice; out[1] = bin_workspace_bytes(p); out[2] = merge_levels; out[3] = p.n_reduce_wgs;
    out[4] = p.total_bytes; out[5] = p.total_counts;
    for (uint32_t l = 0; l < n_levels; ++l) {
        unsigned long long *o = out + 6 + 7 * l;
        o[0] = level_bins(T.level[l]); o[1] = p.level_cap[l]; o[2] = p.level_waves[l]; o[3] = (p.row_mask >> l) & 1u;
        o[4] = p.level_split[l]; o[5] = p.level_wg0[l]; o[6] = p.level_cnt0[l];
    }
    return 0;
}


}  // extern "C"

#ifdef MI3D_DEV
// tools build only: reset (reset != 0) or read the per-XCD timestamps of the last plane gather, and its plan
extern "C" int mi3d_dev_encode_times(unsigned long long *out /* [8 * 17] */, int reset) {
    unsigned long long h[kXcds * (1 + kMaxSegs)];
    if (reset) {
        for (uint32_t x = 0; x < kXcds; ++x) {
            h[x * (1 + kMaxSegs)] = ~0ull;
            for (int i = 1; i <= kMaxSegs; ++i) h[x * (1 + kMaxSegs) + i] = 0ull;
        }
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi3d_dbg_encode), h, sizeof(h));
    }
    const int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(mi3d_dbg_encode), sizeof(h));
    for (uint32_t i = 0; i < kXcds * (1 + kMaxSegs); ++i) out[i] = h[i];
    return rc;
}
#endif
