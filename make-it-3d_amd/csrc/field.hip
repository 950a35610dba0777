// The field's MLP on the MI355X matrix cores: Linear(32,64)-ReLU-Linear(64,64)-ReLU-Linear(64,4), forward and
// backward (input gradient + all weight/bias gradients) in ONE kernel each, activations never leave registers.
// Replaces the three nn.Linear / two ReLU launches per field evaluation of the reference's `sigma_net`
// (/root/reference/nerf/network_tcnn.py:13-32,67,107) - Part 4 of include/mi3d.h.
//
// Dataflow (wave64, v_mfma_f32_32x32x16_f16 or, in exact-fp32 mode, v_mfma_f32_32x32x2_f32).  One wave owns a
// TILE of 32 rows (field evaluations).  Every matrix product is D = A.B with a 32x32 output tile whose lane l holds
// column (l & 31) and, in register q, row rowmap(q, l >> 5) = (q & 3) + 8 (q >> 2) + 4 (l >> 5).  A "K-block" is 32
// contraction indices held as 16 values per lane: value q of lane-half h stands for index kmap(q, h).  Two facts
// make the whole network chain through registers with no LDS round trip and no cross-lane traffic:
//   (1) the hardware pairs A and B by K-SLOT, so any kmap works as long as both operands use it - in particular
//       kmap = rowmap: the accumulators of one layer ARE the next layer's operand (after bias/ReLU/convert);
//   (2) A and B fragments have the same lane layout (lane = row of A / column of B), so an activation tile held
//       as "lane = sample, values = features" can be the B operand (output: lane = sample, rows = out features,
//       "orientation 1") or the A operand (output: lane = out feature, rows = samples, "orientation 2").
// Orientation 2 is what the weight gradients need (contraction over samples), so the backward kernel runs the
// cheap 32-wide layers in both orientations instead of transposing tiles through LDS: 34 K-block products per tile
// (64 MFMAs in fp16 mode, 2048 matrix-pipe cycles per 32 samples) - the kernel stays bound by streaming its rows.
//
// Weights live in LDS as ready-made operand blocks (one 16-byte vector per lane per read, conflict-free), built
// once per workgroup from the fp32 master weights; fp16 mode rounds weights, biases and inter-layer activations to
// binary16 exactly where torch.autocast does for nn.Linear (oracle/field_ref.c half_mode).
//
// Weight gradients accumulate in registers across all tiles of a persistent wave, are reduced across the
// workgroup in LDS and leave with one atomic per element per workgroup.
#include <hip/hip_runtime.h>

#include "../../include/mi3d.h"
#include "mi3d_dev.h"
#include "lds_transpose.h"

// 1: the backward turns its binary16 tiles round through LDS (ds_read_b64_tr_b16, csrc/lds_transpose.h); 0: on the matrix
// core, as a product with an identity block (round 3)
#ifndef MI3D_MLP_LDS_TRANSPOSE
#define MI3D_MLP_LDS_TRANSPOSE 1
#endif
// development switches of the backward (variant builds, tools/build_dev.py): an instance of its own for the full input
// width; the next tile's rows requested late in the tile (after the last use of this tile's) instead of at its top
#ifndef MI3D_MLP_BWD_FULL
#define MI3D_MLP_BWD_FULL 1
#endif
#ifndef MI3D_MLP_BWD_LATE_PREFETCH
#define MI3D_MLP_BWD_LATE_PREFETCH 1
#endif
// TIMING ONLY (variant builds for tools/mlp_ab.py --timing-only; the gradients are wrong with either bit): what a group of
// the backward's vector instructions costs - 1 = no bias-gradient sums (40 dot products per tile), 2 = no ReLU masks on
// the gradients (64 packed operations per tile); 4 (round 6) = recompute + input gradient ONLY, in the
// "lane = sample" orientation throughout (14 matrix products per tile, no transposes, no weight-gradient registers) at
// THREE waves per SIMD: what a producer wave of a producer / consumer pair could at best cost (the input-gradient planes
// of this build are correct, every weight gradient is zero)
#ifndef MI3D_MLP_DO_LDS
// the 4-wide output gradient's tile turned round through the wave's LDS like every other binary16 tile, instead of on the
// matrix core (VERDICT r04 item 3 / DESIGN.md 7.3a): 42 MFMAs and 72 packed converts per tile instead of 43 / 80.  Both
// builds in one process (tools/mlp_ab.py, profiles/mlp_ab_r05_do_lds.json): every template instance's input-gradient
// planes bit-identical, weight gradients to 1.5e-6 (the order of their float atomics); 13 points 8.37 -> 8.34 ms, point-0
// pass 0.71 -> 0.71: the last matrix-core transpose of the binary16 kernels is gone, the time it took was already hidden.
#define MI3D_MLP_DO_LDS 1
#endif
#ifndef MI3D_MLP_BWD_TIMING_CUT
#define MI3D_MLP_BWD_TIMING_CUT 0
#endif

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerWG = 4;
constexpr int DIN = 32, HID = 64, DOUT = 4;
constexpr int NTH = HID / 32;  // 32-wide tiles across the hidden width

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using ushort8v = __attribute__((ext_vector_type(8))) unsigned short;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ int rowmap(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// ---------------------------------------------------------------- precision policies
struct F16 {
    using elem = _Float16;
    static constexpr int kUnits = 2;  // 16-byte vectors per lane per K-block (8 halfs each)
    static constexpr bool kLdsTranspose = MI3D_MLP_LDS_TRANSPOSE != 0;
    struct KB { half8 v[2]; };
    __device__ static __forceinline__ float round(float x) { return (float)(_Float16)x; }
    __device__ static __forceinline__ void set(KB &k, int q, float x) { k.v[q >> 3][q & 7] = (_Float16)x; }
    // values 2j, 2j+1 of the K-block from a packed binary16 pair
    __device__ static __forceinline__ void set_pair_bits(KB &k, int j, uint32_t bits) {
        const half2v hv = __builtin_bit_cast(half2v, bits);
        k.v[j >> 2][2 * (j & 3)] = hv[0];
        k.v[j >> 2][2 * (j & 3) + 1] = hv[1];
    }
    __device__ static __forceinline__ void mma(f32x16 &acc, const KB &a, const KB &b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v[0], b.v[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v[1], b.v[1], acc, 0, 0, 0);
    }
    // one K-block product = kSteps matrix instructions; issuing step s of several products before step s + 1 of any
    // keeps consecutive MFMAs on different accumulators
    [[maybe_unused]] static constexpr int kSteps = 2;
    __device__ static __forceinline__ void mma_step(f32x16 &acc, const KB &a, const KB &b, int s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v[s], b.v[s], acc, 0, 0, 0);
    }
    // only values q < 8 of both operands are non-zero
    __device__ static __forceinline__ void mma_lo(f32x16 &acc, const KB &a, const KB &b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v[0], b.v[0], acc, 0, 0, 0);
    }
    __device__ static __forceinline__ KB load_block(const char *blk, int lane) {
        KB k;
        k.v[0] = *reinterpret_cast<const half8 *>(blk + (0 * kWave + lane) * 16);
        k.v[1] = *reinterpret_cast<const half8 *>(blk + (1 * kWave + lane) * 16);
        return k;
    }
    // accumulator tile -> ReLU'd K-block: packed convert (the layer output is rounded to binary16 first, as autocast
    // does), packed max.  The K-block itself is the ReLU mask later on (value > 0 <=> bits != 0).
    __device__ static __forceinline__ KB relu(const f32x16 &acc) {
        KB k;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2 pr = {acc[2 * j], acc[2 * j + 1]};
            half2v hv = __builtin_convertvector(pr, half2v);
            hv = __builtin_elementwise_max(hv, half2v{(_Float16)0, (_Float16)0});
            k.v[j >> 2][2 * (j & 3)] = hv[0];
            k.v[j >> 2][2 * (j & 3) + 1] = hv[1];
        }
        return k;
    }
    // gradient tile masked by the ReLU of `act` (a K-block from relu()): packed convert, then per binary16 lane the
    // bit pattern times min(act bits, 1) in packed u16 arithmetic - the pattern itself where act != 0, else +0.  (Written as inline asm: the
    // optimiser canonicalises every C spelling of it into 16 compares + 16 selects + 16 scalar converts per K-block,
    // which made the masks a third of the kernel's instructions.)
    __device__ static __forceinline__ KB masked(const f32x16 &d, const KB &act) {
        KB k;
        const unsigned ones = 0x00010001u;
        using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;   // (whole registers in and out: see sum())
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u32x4 aw = __builtin_bit_cast(u32x4, act.v[u]);
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 pr = {d[8 * u + 2 * j], d[8 * u + 2 * j + 1]};
                const unsigned dv = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, half2v));
                unsigned m = aw[j];
                asm("v_pk_min_u16 %0, %1, %2\n\tv_pk_mul_lo_u16 %0, %0, %3"
                    : "=&v"(m) : "v"(m), "s"(ones), "v"(dv));
                o[j] = m;
            }
            k.v[u] = __builtin_bit_cast(half8, o);
        }
        return k;
    }
    // accumulator tile -> K-block, no activation (the values are binary16-exact already: products with the identity)
    __device__ static __forceinline__ KB cast(const f32x16 &acc) {
        KB k;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2 pr = {acc[2 * j], acc[2 * j + 1]};
            const half2v hv = __builtin_convertvector(pr, half2v);
            k.v[j >> 2][2 * (j & 3)] = hv[0];
            k.v[j >> 2][2 * (j & 3) + 1] = hv[1];
        }
        return k;
    }
    // fp32 sum of the 16 values of a K-block (v_dot2_f32_f16 against ones)
    __device__ static __forceinline__ float sum(const KB &k) {
        float s = 0.f;
        const half2v ones = {(_Float16)1, (_Float16)1};
        // (whole registers: picking the two halves of a pair out of the vector and putting them together again made
        // hipcc emit a v_bfi copy in front of 6 of every 8 dot products - 30 of the backward's ~360 VALU instructions)
        using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u32x4 w = __builtin_bit_cast(u32x4, k.v[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t wj = w[j];   // (a copy first: __builtin_bit_cast of the vector ELEMENT w[j] reads element 0 - hipcc 7.2)
                s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, wj), ones, s, false);
            }
        }
        return s;
    }
};

struct F32 {
    using elem = float;
    static constexpr int kUnits = 4;  // 4 floats each
    static constexpr bool kLdsTranspose = false;  // (the transposing LDS read moves 16-bit values)
    struct KB { float v[16]; };
    __device__ static __forceinline__ float round(float x) { return x; }
    __device__ static __forceinline__ void set(KB &k, int q, float x) { k.v[q] = x; }
    __device__ static __forceinline__ void set_pair_bits(KB &k, int j, uint32_t bits) {  // (binary16 planes: F16 mode only)
        const half2v hv = __builtin_bit_cast(half2v, bits);
        k.v[2 * j] = (float)hv[0];
        k.v[2 * j + 1] = (float)hv[1];
    }
    __device__ static __forceinline__ void mma(f32x16 &acc, const KB &a, const KB &b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q], b.v[q], acc, 0, 0, 0);
    }
    static constexpr int kSteps = 16;
    __device__ static __forceinline__ void mma_step(f32x16 &acc, const KB &a, const KB &b, int s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], acc, 0, 0, 0);
    }
    __device__ static __forceinline__ void mma_lo(f32x16 &acc, const KB &a, const KB &b) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q], b.v[q], acc, 0, 0, 0);
    }
    __device__ static __forceinline__ KB load_block(const char *blk, int lane) {
        KB k;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(blk + (u * kWave + lane) * 16);
            k.v[4 * u] = t[0]; k.v[4 * u + 1] = t[1]; k.v[4 * u + 2] = t[2]; k.v[4 * u + 3] = t[3];
        }
        return k;
    }
    __device__ static __forceinline__ KB relu(const f32x16 &acc) {
        KB k;
#pragma unroll
        for (int q = 0; q < 16; ++q) k.v[q] = fmaxf(acc[q], 0.f);
        return k;
    }
    __device__ static __forceinline__ KB masked(const f32x16 &d, const KB &act) {
        KB k;
#pragma unroll
        for (int q = 0; q < 16; ++q) k.v[q] = act.v[q] > 0.f ? d[q] : 0.f;
        return k;
    }
    __device__ static __forceinline__ KB cast(const f32x16 &acc) {
        KB k;
#pragma unroll
        for (int q = 0; q < 16; ++q) k.v[q] = acc[q];
        return k;
    }
    __device__ static __forceinline__ float sum(const KB &k) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += k.v[q];
        return s;
    }
};

template <class P> constexpr int block_bytes() { return P::kUnits * kWave * 16; }

// ---------------------------------------------------------------- operand blocks in LDS
// Block list (N = the dimension lanes run over, K = the contraction dimension; kind D: kmap = rowmap, kind X:
// kmap(q, h) = 16 h + q, the layout rows are loaded from memory in):
enum : int {
    B_W1 = 0,              // [tn]      N = hidden-1 feature, K = input feature (X)      Wm[N][K] = W1[N][K]
    B_W2 = B_W1 + NTH,     // [tn][tk]  N = hidden-2 feature, K = hidden-1 feature (D)   W2[N][K]
    B_W3 = B_W2 + NTH * NTH,   // [tk]  N = output (4, zero padded), K = hidden-2 (D)    W3[N][K]
    B_W3T = B_W3 + NTH,    // [tn]      N = hidden-2 feature, K = output index (X, < 4)  W3[K][N]
    B_W2T = B_W3T + NTH,   // [tn][tk]  N = hidden-1 feature, K = hidden-2 feature (D)   W2[K][N]
    B_W1T = B_W2T + NTH * NTH, // [tk]  N = input feature, K = hidden-1 feature (D)      W1[K][N]
    B_ID = B_W1T + NTH,    //           N = index j, K = index k (X): (j == k) - as the B operand of a tile held
                           //           "lane = sample" it hands back the tile "lane = index" (a transpose on the matrix core)
    B_FWD_COUNT = B_W3T,
    B_ALL_COUNT = B_ID + 1,
};

struct Weights {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
};

template <class P>
__device__ void build_blocks(char *lds, float *bias /* [HID + HID + 32] */, const Weights &w, int n_blocks) {
    using T = typename P::elem;
    for (int e = threadIdx.x; e < n_blocks * kWave * 16; e += blockDim.x) {
        const int blk = e / (kWave * 16), r = e % (kWave * 16), lane = r / 16, q = r % 16;
        const int nl = lane & 31, h = lane >> 5;
        const int kd = rowmap(q, h), kx = 16 * h + q;
        float v = 0.f;
        if (blk < B_W2) {
            const int tn = blk - B_W1;
            v = w.W1[(32 * tn + nl) * DIN + kx];
        } else if (blk < B_W3) {
            const int tn = (blk - B_W2) / NTH, tk = (blk - B_W2) % NTH;
            v = w.W2[(32 * tn + nl) * HID + 32 * tk + kd];
        } else if (blk < B_W3T) {
            const int tk = blk - B_W3;
            v = nl < DOUT ? w.W3[nl * HID + 32 * tk + kd] : 0.f;
        } else if (blk < B_W2T) {
            const int tn = blk - B_W3T;
            v = kx < DOUT ? w.W3[kx * HID + 32 * tn + nl] : 0.f;
        } else if (blk < B_W1T) {
            const int tn = (blk - B_W2T) / NTH, tk = (blk - B_W2T) % NTH;
            v = w.W2[(32 * tk + kd) * HID + 32 * tn + nl];
        } else if (blk < B_ID) {
            const int tk = blk - B_W1T;
            v = w.W1[(32 * tk + kd) * DIN + nl];
        } else {
            v = nl == kx ? 1.f : 0.f;
        }
        // value q of lane sits in 16-byte unit q / per_unit
        constexpr int per_unit = 16 / (int)sizeof(T);
        T *dst = reinterpret_cast<T *>(lds + (size_t)blk * block_bytes<P>() + ((q / per_unit) * kWave + lane) * 16);
        dst[q % per_unit] = (T)v;
    }
    for (int e = threadIdx.x; e < HID + HID + 32; e += blockDim.x) {
        float v = 0.f;
        if (e < HID) v = w.b1[e];
        else if (e < 2 * HID) v = w.b2[e - HID];
        else if (e - 2 * HID < DOUT) v = w.b3[e - 2 * HID];
        bias[e] = P::round(v);
    }
}

// ---------------------------------------------------------------- tile pieces
// rows of the tile as a K-block (kind X): lane (p, h) holds features 16 h .. 16 h + 15 of row row0 + p.
// x is either [n, DIN] rows or (plane_rows != 0) level-major planes [DIN/2][plane_rows][2]: feature pair (2l, 2l+1) of
// row r at x[(l plane_rows + r) 2]; plane_rows >= n lets a caller process a prefix of the rows of wider planes.
// The 16 floats are fetched raw (so the backward can have the next tile's in flight while it computes) and turned into
// the precision's K-block afterwards.
// Rows past n are read from row n - 1 instead of being predicated off (no exec-mask branches around the loads, so all
// of them are in flight together); their products are harmless: every weight-gradient term carries a factor dout,
// which IS zeroed for those rows, and their outputs are never stored.
// planes_half: the planes hold binary16 pairs (4 bytes per (level, row)) - see plane layouts in include/mi3d.h.
__device__ __forceinline__ void load_rows_raw(const float *__restrict__ x, size_t row, size_t n, int h,
                                              size_t plane_rows, float (&raw)[16], int planes_half = 0) {
    row = row < n ? row : n - 1;
    if (plane_rows != 0 && planes_half) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // 32 consecutive rows of one plane per load: 128 contiguous bytes per lane-half
            const uint32_t u = reinterpret_cast<const uint32_t *>(x)[(size_t)(8 * h + j) * plane_rows + row];
            raw[2 * j] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xFFFFu));
            raw[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
        }
    } else if (plane_rows == 0) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(x + row * DIN + 16 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 t = src[c];
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[4 * c + i] = t[i];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // 32 consecutive rows of one plane per load: 256 contiguous bytes per lane-half
            const f32x2 t = *reinterpret_cast<const f32x2 *>(x + ((size_t)(8 * h + j) * plane_rows + row) * 2);
            raw[2 * j] = t[0];
            raw[2 * j + 1] = t[1];
        }
    }
}
template <class P> __device__ __forceinline__ typename P::KB rows_kb(const float (&raw)[16]) {
    typename P::KB k;
#pragma unroll
    for (int q = 0; q < 16; ++q) P::set(k, q, raw[q]);
    return k;
}
// binary16 planes: the 8 dwords a lane fetches ARE its K-block (pair j = features 2j, 2j+1 of this lane-half) - the
// backward keeps them as bit patterns while they are in flight and re-interprets them, no conversion either way
[[maybe_unused]] __device__ __forceinline__ void load_rows_half(const float *__restrict__ x, size_t row, size_t n, int h,
                                               size_t plane_rows, uint32_t (&u)[8]) {
    row = row < n ? row : n - 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = reinterpret_cast<const uint32_t *>(x)[(size_t)(8 * h + j) * plane_rows + row];
}
template <class P> __device__ __forceinline__ typename P::KB rows_kb_half(const uint32_t (&u)[8]) {
    typename P::KB k;
#pragma unroll
    for (int j = 0; j < 8; ++j) P::set_pair_bits(k, j, u[j]);
    return k;
}
template <class P>
__device__ __forceinline__ typename P::KB load_rows_kb(const float *__restrict__ x, size_t row, size_t n, int h,
                                                       size_t plane_rows, int planes_half) {
    float raw[16];
    load_rows_raw(x, row, n, h, plane_rows, raw, planes_half);
    return rows_kb<P>(raw);
}

// [rows,4] output-side gradient as a K-block (kind X over the 4 outputs): only lane-half 0, q < 4 are non-zero
// (the zeroing of rows past n and of lane-half 1 happens at the use, one iteration after the load was issued)
__device__ __forceinline__ f32x4 load_dout_raw(const float *__restrict__ dout, size_t row, size_t n) {
    return *reinterpret_cast<const f32x4 *>(dout + (row < n ? row : n - 1) * DOUT);
}
template <class P> __device__ __forceinline__ typename P::KB dout_kb(const f32x4 &t, bool keep) {
    typename P::KB k;
#pragma unroll
    for (int q = 0; q < 16; ++q) P::set(k, q, (q < 4 && keep) ? t[q] : 0.f);
    return k;
}

__device__ __forceinline__ f32x16 splat(float v) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = v;
    return a;
}
// bias along the ROWS of the tile (orientation 1: rows = features)
__device__ __forceinline__ f32x16 bias_rows(const float *bias, int h) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = bias[rowmap(q, h)];
    return a;
}
// ---------------------------------------------------------------- forward
template <class P>
__global__ __launch_bounds__(kWave *kWavesPerWG, 2) void k_mlp_forward(const float *__restrict__ x, uint32_t x_planes, int x_half,
                                                                     uint32_t n, Weights w, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *bias = reinterpret_cast<float *>(lds + (size_t)B_FWD_COUNT * block_bytes<P>());
    build_blocks<P>(lds, bias, w, B_FWD_COUNT);
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), p = lane & 31, h = lane >> 5;
    const uint32_t wave = blockIdx.x * kWavesPerWG + threadIdx.x / kWave, n_waves = gridDim.x * kWavesPerWG;
    const uint32_t n_tiles = (n + 31) / 32;
    // the lane offset is laundered through an empty asm so every use is a fresh LDS read: hipcc would otherwise
    // hoist all (loop-invariant) operand blocks into registers and spill the accumulators
    auto blk = [&](int b) {
        int l = lane;
        asm volatile("" : "+v"(l));
        return P::load_block(lds + (size_t)b * block_bytes<P>(), l);
    };

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const size_t row = (size_t)tile * 32 + p;
        const bool valid = row < n;
        const typename P::KB X = load_rows_kb<P>(x, row, n, h, x_planes, x_half);
        typename P::KB H1[NTH], H2[NTH];
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            f32x16 acc = bias_rows(bias + 32 * t, h);
            P::mma(acc, blk(B_W1 + t), X);
            H1[t] = P::relu(acc);
        }
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            f32x16 acc = bias_rows(bias + HID + 32 * t, h);
#pragma unroll
            for (int tk = 0; tk < NTH; ++tk) P::mma(acc, blk(B_W2 + t * NTH + tk), H1[tk]);
            H2[t] = P::relu(acc);
        }
        f32x16 acc = bias_rows(bias + 2 * HID, h);
#pragma unroll
        for (int tk = 0; tk < NTH; ++tk) P::mma(acc, blk(B_W3 + tk), H2[tk]);
        if (valid && h == 0) {  // rows 0..3 of the tile = the four outputs, held by lane-half 0 in registers 0..3
            f32x4 o = {P::round(acc[0]), P::round(acc[1]), P::round(acc[2]), P::round(acc[3])};
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(out + row * DOUT));
        }
    }
}

// ---------------------------------------------------------------- backward
struct Grads {
    float *dW1, *db1, *dW2, *db2, *dW3, *db3;
};

// U = tiles (of 32 rows) a wave works on at once, WPS = waves per SIMD the register allocation is held to.
//
// One wave per SIMD (the weight-gradient tiles alone are 128 registers), so nothing hides a latency but the wave's own
// independent work.  Round 2 measured where the time went (tools/kbench.py): prefetching the next tile's rows gained
// 1 ms of 20, the other 18 are the dependency chain product -> convert / mask -> next product, ~250 cycles per product
// against 64 on the matrix pipe.  So the kernel is written in STAGES over U = 2 tiles x NTH = 2 output tiles: every
// stage first issues the matrix products of all its (tile, output-tile) pairs step by step - consecutive MFMAs go to
// different accumulators - and only then converts them, which gives the in-order wave four independent chains to
// overlap (the packed converts of one accumulator run under the MFMAs of the next).
template <class P, int U, int WPS, bool HP>
__global__ __launch_bounds__(kWave *kWavesPerWG, WPS) void k_mlp_backward(const float *__restrict__ x, uint32_t x_planes,
                                                                           int /*planes_half: HP*/, const float *__restrict__ dout,
                                                                           uint32_t n, Weights w, float *__restrict__ dx,
                                                                           uint32_t dx_planes, Grads g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *bias = reinterpret_cast<float *>(lds + (size_t)B_ALL_COUNT * block_bytes<P>());
    build_blocks<P>(lds, bias, w, B_ALL_COUNT);
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), p = lane & 31, h = lane >> 5;
    const uint32_t wave = blockIdx.x * kWavesPerWG + threadIdx.x / kWave, n_waves = gridDim.x * kWavesPerWG;
    const uint32_t n_tiles = (n + 31) / 32;
    using KB = typename P::KB;
    // the lane offset is laundered through an empty asm so every use is a fresh LDS read: hipcc would otherwise
    // hoist all (loop-invariant) operand blocks into registers and spill the accumulators
    auto blk = [&](int b) {
        int l = lane;
        asm volatile("" : "+v"(l));
        return P::load_block(lds + (size_t)b * block_bytes<P>(), l);
    };

    // weight-gradient tiles: lane = column (input-side feature), register q = row rowmap(q, h) (output-side feature)
    f32x16 gW2[NTH][NTH], gW1[NTH], gW3[NTH];
    float gb1[NTH], gb2[NTH], gb3 = 0.f;
#pragma unroll
    for (int a = 0; a < NTH; ++a) {
        gW1[a] = splat(0.f); gW3[a] = splat(0.f); gb1[a] = 0.f; gb2[a] = 0.f;
#pragma unroll
        for (int b = 0; b < NTH; ++b) gW2[a][b] = splat(0.f);
    }

    // the rows of the NEXT group of tiles are requested before the current group's products start and are consumed one
    // iteration later; tile u of a group is `u * n_waves` tiles further on, so every wave instruction still reads 32
    // consecutive rows
    float raw[HP ? 1 : U][16];
    uint32_t rawh[HP ? U : 1][8];
    f32x4 dor[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t r0 = ((size_t)wave + (size_t)u * n_waves) * 32 + p;
        if constexpr (HP) load_rows_half(x, r0, n, h, x_planes, rawh[u]);
        else load_rows_raw(x, r0, n, h, x_planes, raw[u]);
        dor[u] = load_dout_raw(dout, r0, n);
    }
    for (uint32_t tile = wave; tile < n_tiles; tile += U * n_waves) {
        size_t row[U];
        bool valid[U];
        KB X[U], dO[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            row[u] = ((size_t)tile + (size_t)u * n_waves) * 32 + p;
            valid[u] = row[u] < n;
            if constexpr (HP) X[u] = rows_kb_half<P>(rawh[u]);
            else X[u] = rows_kb<P>(raw[u]);
            dO[u] = dout_kb<P>(dor[u], valid[u] && h == 0);
        }
        if (tile + U * n_waves < n_tiles) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t rn = ((size_t)tile + (size_t)(U + u) * n_waves) * 32 + p;
                if constexpr (HP) load_rows_half(x, rn, n, h, x_planes, rawh[u]);
                else load_rows_raw(x, rn, n, h, x_planes, raw[u]);
                dor[u] = load_dout_raw(dout, rn, n);
            }
        }

        // ---- orientation 1 (lane = sample): recompute the activations (they double as their own ReLU masks)
        KB H1[U][NTH], H2[U][NTH], dH2[U][NTH], dH1[U][NTH];
        {
            f32x16 acc[U][NTH];
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u][t] = bias_rows(bias + 32 * t, h);
            {
                KB Wb[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W1 + t);
#pragma unroll
                for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                    for (int t = 0; t < NTH; ++t)
#pragma unroll
                        for (int u = 0; u < U; ++u) P::mma_step(acc[u][t], Wb[t], X[u], st);
            }
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) H1[u][t] = P::relu(acc[u][t]);
        }
        {
            f32x16 acc[U][NTH];
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u][t] = bias_rows(bias + HID + 32 * t, h);
#pragma unroll
            for (int tk = 0; tk < NTH; ++tk) {
                KB Wb[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W2 + t * NTH + tk);
#pragma unroll
                for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                    for (int t = 0; t < NTH; ++t)
#pragma unroll
                        for (int u = 0; u < U; ++u) P::mma_step(acc[u][t], Wb[t], H1[u][tk], st);
            }
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) H2[u][t] = P::relu(acc[u][t]);
        }
        // ---- orientation 1: input-side gradients  dH2 = W3^T dO, dH1 = W2^T dH2, dX = W1^T dH1
        {
            f32x16 acc[U][NTH];
            KB Wb[NTH];
#pragma unroll
            for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W3T + t);
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) { acc[u][t] = splat(0.f); P::mma_lo(acc[u][t], Wb[t], dO[u]); }
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) dH2[u][t] = P::masked(acc[u][t], H2[u][t]);
        }
        {
            f32x16 acc[U][NTH];
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u][t] = splat(0.f);
#pragma unroll
            for (int tk = 0; tk < NTH; ++tk) {
                KB Wb[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W2T + t * NTH + tk);
#pragma unroll
                for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                    for (int t = 0; t < NTH; ++t)
#pragma unroll
                        for (int u = 0; u < U; ++u) P::mma_step(acc[u][t], Wb[t], dH2[u][tk], st);
            }
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) dH1[u][t] = P::masked(acc[u][t], H1[u][t]);
        }
        {
            f32x16 acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = splat(0.f);
#pragma unroll
            for (int tk = 0; tk < NTH; ++tk) {
                const KB Wb = blk(B_W1T + tk);
#pragma unroll
                for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                    for (int u = 0; u < U; ++u) P::mma_step(acc[u], Wb, dH1[u][tk], st);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (valid[u] && !dx_planes) {  // register q = input feature rowmap(q, h): four runs of four features
                    float *dst = dx + row[u] * DIN + 4 * h;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 o = {acc[u][4 * c], acc[u][4 * c + 1], acc[u][4 * c + 2], acc[u][4 * c + 3]};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(dst + 8 * c));
                    }
                } else if (valid[u] && HP) {
                    // binary16 planes: one 4-byte store per (level, row); this IS the rounding torch.autocast gives the
                    // input gradient of the first nn.Linear (a binary16 GEMM output)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const size_t lvl = 4 * c + 2 * h;
                        const half2v a = __builtin_convertvector((f32x2){acc[u][4 * c], acc[u][4 * c + 1]}, half2v);
                        const half2v b = __builtin_convertvector((f32x2){acc[u][4 * c + 2], acc[u][4 * c + 3]}, half2v);
                        reinterpret_cast<uint32_t *>(dx)[lvl * dx_planes + row[u]] = __builtin_bit_cast(uint32_t, a);
                        reinterpret_cast<uint32_t *>(dx)[(lvl + 1) * dx_planes + row[u]] = __builtin_bit_cast(uint32_t, b);
                    }
                } else if (valid[u]) {
                    // level-major planes [DIN/2][dx_planes][2] (what the binned scatter reads): features (2l, 2l+1) of
                    // this row are one 8-byte store; lane-half h owns levels 4c + 2h and 4c + 2h + 1, 32 consecutive
                    // rows per store
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x2 a = {acc[u][4 * c], acc[u][4 * c + 1]}, b = {acc[u][4 * c + 2], acc[u][4 * c + 3]};
                        const size_t lvl = 4 * c + 2 * h;
                        __builtin_nontemporal_store(a, reinterpret_cast<f32x2 *>(dx + (lvl * dx_planes + row[u]) * 2));
                        __builtin_nontemporal_store(b, reinterpret_cast<f32x2 *>(dx + ((lvl + 1) * dx_planes + row[u]) * 2));
                    }
                }
            }
        }

        // ---- orientation 2 (lane = feature, registers = the tile's 32 samples): operands of the weight gradients.
        // (Stages are cut so that at most one set of U x NTH accumulators is live next to the 128 gradient registers.)
        KB H1p[U][NTH];
        {   // hidden-1 activations
            f32x16 acc[U][NTH];
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u][t] = splat(bias[32 * t + p]);
            KB Wb[NTH];
#pragma unroll
            for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W1 + t);
#pragma unroll
            for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                for (int t = 0; t < NTH; ++t)
#pragma unroll
                    for (int u = 0; u < U; ++u) P::mma_step(acc[u][t], X[u], Wb[t], st);
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) H1p[u][t] = P::relu(acc[u][t]);
        }
        KB dH2p[U][NTH];
        {
            KB H2p[U][NTH], dOp[U];
            {   // hidden-2 activations (the mask, and the operand of dW3)
                f32x16 acc[U][NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t)
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[u][t] = splat(bias[HID + 32 * t + p]);
#pragma unroll
                for (int tk = 0; tk < NTH; ++tk) {
                    KB Wb[NTH];
#pragma unroll
                    for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W2 + t * NTH + tk);
#pragma unroll
                    for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                        for (int t = 0; t < NTH; ++t)
#pragma unroll
                            for (int u = 0; u < U; ++u) P::mma_step(acc[u][t], H1[u][tk], Wb[t], st);
                }
#pragma unroll
                for (int t = 0; t < NTH; ++t)
#pragma unroll
                    for (int u = 0; u < U; ++u) H2p[u][t] = P::relu(acc[u][t]);
            }
            {   // dO with lane = output index, values = samples: dO (lane = sample) times the identity block hands it
                // back transposed (no gather loads); its sum over the samples is db3
                const KB Id = blk(B_ID);
                f32x16 tO[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { tO[u] = splat(0.f); P::mma_lo(tO[u], dO[u], Id); }
#pragma unroll
                for (int u = 0; u < U; ++u) { dOp[u] = P::cast(tO[u]); gb3 += P::sum(dOp[u]); }
            }
            // dW3[o][f] += sum_s dO[s][o] H2[s][f]
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                    for (int t = 0; t < NTH; ++t) P::mma_step(gW3[t], dOp[u], H2p[u][t], st);
            {   // gradient wrt hidden-2
                f32x16 d[U][NTH];
                KB Wb[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W3T + t);
#pragma unroll
                for (int t = 0; t < NTH; ++t)
#pragma unroll
                    for (int u = 0; u < U; ++u) { d[u][t] = splat(0.f); P::mma_lo(d[u][t], dO[u], Wb[t]); }
#pragma unroll
                for (int t = 0; t < NTH; ++t)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        dH2p[u][t] = P::masked(d[u][t], H2p[u][t]);
                        gb2[t] += P::sum(dH2p[u][t]);
                    }
            }
        }
        // dW2[i][j] += sum_s dH2[s][i] H1[s][j]
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                for (int ti = 0; ti < NTH; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NTH; ++tj) P::mma_step(gW2[ti][tj], dH2p[u][ti], H1p[u][tj], st);
        {   // the gradient wrt hidden-1 in orientation 2, the input rows with lane = input feature (X times the
            // identity block), and dW1[i][j] += sum_s dH1[s][i] X[s][j]
            KB dH1p[U][NTH], Xp[U];
            {
                f32x16 d[U][NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t)
#pragma unroll
                    for (int u = 0; u < U; ++u) d[u][t] = splat(0.f);
#pragma unroll
                for (int tk = 0; tk < NTH; ++tk) {
                    KB Wb[NTH];
#pragma unroll
                    for (int t = 0; t < NTH; ++t) Wb[t] = blk(B_W2T + t * NTH + tk);
#pragma unroll
                    for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                        for (int t = 0; t < NTH; ++t)
#pragma unroll
                            for (int u = 0; u < U; ++u) P::mma_step(d[u][t], dH2[u][tk], Wb[t], st);
                }
#pragma unroll
                for (int t = 0; t < NTH; ++t)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        dH1p[u][t] = P::masked(d[u][t], H1p[u][t]);
                        gb1[t] += P::sum(dH1p[u][t]);
                    }
            }
            {
                const KB Id = blk(B_ID);
                f32x16 tX[U];
#pragma unroll
                for (int u = 0; u < U; ++u) tX[u] = splat(0.f);
#pragma unroll
                for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                    for (int u = 0; u < U; ++u) P::mma_step(tX[u], X[u], Id, st);
#pragma unroll
                for (int u = 0; u < U; ++u) Xp[u] = P::cast(tX[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int st = 0; st < P::kSteps; ++st)
#pragma unroll
                    for (int t = 0; t < NTH; ++t) P::mma_step(gW1[t], dH1p[u][t], Xp[u], st);
        }
    }

    // ---- reduce the weight gradients across the workgroup in LDS, then one atomic per element per workgroup
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);
    constexpr int OFF_W1 = 0, OFF_B1 = OFF_W1 + HID * DIN, OFF_W2 = OFF_B1 + HID, OFF_B2 = OFF_W2 + HID * HID,
                  OFF_W3 = OFF_B2 + HID, OFF_B3 = OFF_W3 + DOUT * HID, TOTAL = OFF_B3 + DOUT;
    for (int e = threadIdx.x; e < TOTAL; e += blockDim.x) red[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = rowmap(q, h);
#pragma unroll
        for (int ti = 0; ti < NTH; ++ti) {
            atomicAdd(&red[OFF_W1 + (32 * ti + r) * DIN + p], gW1[ti][q]);
#pragma unroll
            for (int tj = 0; tj < NTH; ++tj) atomicAdd(&red[OFF_W2 + (32 * ti + r) * HID + 32 * tj + p], gW2[ti][tj][q]);
            if (r < DOUT) atomicAdd(&red[OFF_W3 + r * HID + 32 * ti + p], gW3[ti][q]);
        }
    }
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
        atomicAdd(&red[OFF_B1 + 32 * t + p], gb1[t]);
        atomicAdd(&red[OFF_B2 + 32 * t + p], gb2[t]);
    }
    if (p < DOUT) atomicAdd(&red[OFF_B3 + p], gb3);
    __syncthreads();
    for (int e = threadIdx.x; e < TOTAL; e += blockDim.x) {
        const float v = red[e];
        float *dst = e < OFF_B1 ? g.dW1 + (e - OFF_W1)
                   : e < OFF_W2 ? g.db1 + (e - OFF_B1)
                   : e < OFF_B2 ? g.dW2 + (e - OFF_W2)
                   : e < OFF_W3 ? g.db2 + (e - OFF_B2)
                   : e < OFF_B3 ? g.dW3 + (e - OFF_W3)
                                : g.db3 + (e - OFF_B3);
        if (v != 0.f) unsafeAtomicAdd(dst, v);
    }
}

// ================================================================ generic kernels (round 3)
// The same dataflow for every shape the reference's MLP class builds around this field (network_tcnn.py:13-32,67 takes
// num_layers and hidden_dim; BASELINE config 1 is Linear(8,32)-ReLU-Linear(32,4)): dim_in even and <= 32 (padded to one
// K-block with zero weights), dim_hidden 32 or 64, 2 or 3 layers, dim_out 4 - and a different register plan.
//
// What round 2's profile showed (rocprofv3 + the ISA of k_mlp_backward): 12.0 ms for 141 M rows whatever the staging -
// one tile at a time, two, prefetch or not - because the kernel was bound by INSTRUCTION ISSUE of its single wave per
// SIMD: 1175 instructions per 32-row tile, 40 % of them v_accvgpr_read / _mov.  hipcc puts every MFMA result into the
// accumulator file as soon as a kernel may use more than 256 registers (one wave per SIMD), and VALU cannot read AGPRs,
// so each of the 16 values of every product paid a move before its convert.  Here the kernels are held to TWO waves
// per SIMD (<= 256 registers, __launch_bounds__(256, 2)): the compiler then selects the VGPR form of every MFMA and
// there is no accumulator file at all, the second wave fills the issue slots the first one leaves under its matrix
// products, and the operands of the weight gradients (which need "lane = feature, registers = samples") come from
// TRANSPOSING the orientation-1 tiles on the matrix core (a product with an identity block: 2 MFMAs and 8 converts per
// K-block) instead of recomputing the layers in the second orientation (4 MFMAs, 16-32 converts / masks and a bias
// splat per K-block): 61 MFMAs and 676 instructions per tile.
//
// Round 4: binary16 tiles are turned round through the wave's own LDS instead (ds_read_b64_tr_b16, csrc/lds_transpose.h:
// 4 writes + 4 transposing reads per lane, nothing on the matrix core, no converts) - the kernel was issue-bound, and 20
// of its 61 MFMAs moved bits.  43 MFMAs and ~500 instructions per tile with the instruction diet described at the
// functions below (whole-register sums / masks, an instance of its own for the full input width, the next tile's rows
// requested after this tile's last use of its own): 9.2-9.5 -> 8.3 ms for 141 M rows in one process
// (tools/mlp_ab.py, profiles/mlp_ab_r04_final.json), every output of every instance bit-identical in the input
// gradient.  The exact-fp32 kernels keep the identity product.
template <int NTH, int LAYERS> struct Blk {
    static constexpr int W1 = 0;                                      // [tn]      as B_W1
    static constexpr int W2 = W1 + NTH;                               // [tn][tk]  as B_W2 (three layers only)
    static constexpr int W3 = W2 + (LAYERS == 3 ? NTH * NTH : 0);     // [tk]      the LAST layer (4 outputs)
    static constexpr int FWD_COUNT = W3 + NTH;
    static constexpr int W3T = FWD_COUNT;                             // [tn]
    static constexpr int W2T = W3T + NTH;                             // [tn][tk]
    static constexpr int W1T = W2T + (LAYERS == 3 ? NTH * NTH : 0);   // [tk]
    static constexpr int IDX = W1T + NTH;                             // identity, K kind X (transposes X and dO)
    static constexpr int IDD = IDX + 1;                               // identity, K kind D (transposes layer outputs)
    static constexpr int ALL_COUNT = IDD + 1;
    static constexpr int BIAS_TILES = 2 * NTH + 1;                    // b1[t], b2[t], b_last: [h][16] floats each
};

// Operand blocks (layouts as build_blocks above) for hidden width 32 NTH and input width din <= 32, plus the biases as
// ready-made accumulator tiles: tile i, lane-half h, register q = bias of row rowmap(q, h) - one broadcast 64-byte read
// per lane initialises an accumulator.
template <class P, int NTH, int LAYERS>
__device__ void build_blocks_g(char *lds, float *biasT, const Weights &w, int din, int n_blocks) {
    using T = typename P::elem;
    using B = Blk<NTH, LAYERS>;
    constexpr int H = 32 * NTH;
    for (int e = threadIdx.x; e < n_blocks * kWave * 16; e += blockDim.x) {
        const int blk = e / (kWave * 16), r = e % (kWave * 16), lane = r / 16, q = r % 16;
        const int nl = lane & 31, h = lane >> 5;
        const int kd = rowmap(q, h), kx = 16 * h + q;
        float v = 0.f;
        if (blk < B::W2) {
            const int tn = blk - B::W1;
            v = kx < din ? w.W1[(32 * tn + nl) * din + kx] : 0.f;
        } else if (blk < B::W3) {
            const int tn = (blk - B::W2) / NTH, tk = (blk - B::W2) % NTH;
            v = w.W2[(32 * tn + nl) * H + 32 * tk + kd];
        } else if (blk < B::W3T) {
            const int tk = blk - B::W3;
            v = nl < DOUT ? w.W3[nl * H + 32 * tk + kd] : 0.f;
        } else if (blk < B::W2T) {
            const int tn = blk - B::W3T;
            v = kx < DOUT ? w.W3[kx * H + 32 * tn + nl] : 0.f;
        } else if (blk < B::W1T) {
            const int tn = (blk - B::W2T) / NTH, tk = (blk - B::W2T) % NTH;
            v = w.W2[(32 * tk + kd) * H + 32 * tn + nl];
        } else if (blk < B::IDX) {
            const int tk = blk - B::W1T;
            v = nl < din ? w.W1[(32 * tk + kd) * din + nl] : 0.f;
        } else if (blk == B::IDX) {
            v = nl == kx ? 1.f : 0.f;
        } else {
            v = nl == kd ? 1.f : 0.f;
        }
        constexpr int per_unit = 16 / (int)sizeof(T);
        T *dst = reinterpret_cast<T *>(lds + (size_t)blk * block_bytes<P>() + ((q / per_unit) * kWave + lane) * 16);
        dst[q % per_unit] = (T)v;
    }
    for (int e = threadIdx.x; e < B::BIAS_TILES * 32; e += blockDim.x) {
        const int tile = e / 32, h = (e % 32) / 16, q = e % 16, row = rowmap(q, h);
        float v = 0.f;
        if (tile < NTH) v = w.b1[32 * tile + row];
        else if (tile < 2 * NTH) v = LAYERS == 3 ? w.b2[32 * (tile - NTH) + row] : 0.f;
        else v = row < DOUT ? w.b3[row] : 0.f;
        biasT[e] = P::round(v);
    }
}
__device__ __forceinline__ f32x16 bias_tile(const float *biasT, int tile, int h) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(biasT + (tile * 2 + h) * 16);
    f32x16 a;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 t = src[c];
        a[4 * c] = t[0]; a[4 * c + 1] = t[1]; a[4 * c + 2] = t[2]; a[4 * c + 3] = t[3];
    }
    return a;
}

// Rows of a tile for input width din <= 32.  Planes / features past din are read from the last valid one instead of
// being predicated off (their weights are zero, so whatever finite value they carry contributes nothing).
template <bool FULL = false>
__device__ __forceinline__ void load_rows_half_g(const float *__restrict__ x, size_t row, size_t n, int h,
                                                 size_t plane_rows, uint32_t last_plane, uint32_t (&u)[8]) {
    row = row < n ? row : n - 1;
    if constexpr (FULL) {   // all 16 planes: ONE 64-bit lane address, the planes a uniform stride apart
        const uint32_t *src = reinterpret_cast<const uint32_t *>(x) + ((size_t)(8 * h) * plane_rows + row);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = src[(size_t)j * plane_rows];
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t pl = (uint32_t)(8 * h + j) < last_plane ? (uint32_t)(8 * h + j) : last_plane;
        u[j] = reinterpret_cast<const uint32_t *>(x)[(size_t)pl * plane_rows + row];
    }
}
__device__ __forceinline__ void load_rows_raw_g(const float *__restrict__ x, size_t row, size_t n, int h,
                                                size_t plane_rows, uint32_t din, float (&raw)[16]) {
    row = row < n ? row : n - 1;
    const uint32_t last_plane = din / 2 - 1;
    if (plane_rows == 0 && din == DIN) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(x + row * DIN + 16 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 t = src[c];
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[4 * c + i] = t[i];
        }
    } else if (plane_rows == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t f = (uint32_t)(16 * h + q) < din ? (uint32_t)(16 * h + q) : din - 1;
            raw[q] = x[row * din + f];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t pl = (uint32_t)(8 * h + j) < last_plane ? (uint32_t)(8 * h + j) : last_plane;
            const f32x2 t = *reinterpret_cast<const f32x2 *>(x + ((size_t)pl * plane_rows + row) * 2);
            raw[2 * j] = t[0];
            raw[2 * j + 1] = t[1];
        }
    }
}

template <class P, int NTH, int LAYERS, bool HP>
__global__ __launch_bounds__(kWave *kWavesPerWG, 2) void k_mlp_fwd_g(const float *__restrict__ x, uint32_t x_planes,
                                                                        uint32_t n, uint32_t din, Weights w,
                                                                        float *__restrict__ out,
                                                                        const int32_t *__restrict__ count, uint32_t n_stride) {
    using B = Blk<NTH, LAYERS>;
    using KB = typename P::KB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *biasT = reinterpret_cast<float *>(lds + (size_t)B::FWD_COUNT * block_bytes<P>());
    build_blocks_g<P, NTH, LAYERS>(lds, biasT, w, (int)din, B::FWD_COUNT);
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), p = lane & 31, h = lane >> 5;
    const uint32_t wave = blockIdx.x * kWavesPerWG + threadIdx.x / kWave, n_waves = gridDim.x * kWavesPerWG;
    const uint32_t n_tiles = (n + 31) / 32;
    auto blk = [&](int b) {  // fresh LDS read at every use (see k_mlp_forward)
        int l = lane;
        asm volatile("" : "+v"(l));
        return P::load_block(lds + (size_t)b * block_bytes<P>(), l);
    };
    auto bias = [&](int tile) {
        int hh = h;
        asm volatile("" : "+v"(hh));
        return bias_tile(biasT, tile, hh);
    };
    const uint32_t last_plane = din / 2 - 1;
    // `count` (device, optional): rows are point-major, row = point * n_stride + sample, and only samples below *count
    // carry data - a tile all of whose rows are beyond it is skipped (its outputs are never read)
    const uint32_t c_rows = count ? (uint32_t)max(*count, 0) : 0xFFFFFFFFu;
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        if (count) {
            const uint32_t s0 = (tile * 32u) % n_stride;
            if (s0 >= c_rows && s0 + 32u <= n_stride) continue;
        }
        const size_t row = (size_t)tile * 32 + p;
        KB X;
        if constexpr (HP) {
            uint32_t u[8];
            load_rows_half_g(x, row, n, h, x_planes, last_plane, u);
            X = rows_kb_half<P>(u);
        } else {
            float raw[16];
            load_rows_raw_g(x, row, n, h, x_planes, din, raw);
            X = rows_kb<P>(raw);
        }
        KB H1[NTH], HL[NTH];
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            f32x16 acc = bias(t);
            P::mma(acc, blk(B::W1 + t), X);
            H1[t] = P::relu(acc);
        }
        if constexpr (LAYERS == 3) {
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
                f32x16 acc = bias(NTH + t);
#pragma unroll
                for (int tk = 0; tk < NTH; ++tk) P::mma(acc, blk(B::W2 + t * NTH + tk), H1[tk]);
                HL[t] = P::relu(acc);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NTH; ++t) HL[t] = H1[t];
        }
        f32x16 acc = bias(2 * NTH);
#pragma unroll
        for (int tk = 0; tk < NTH; ++tk) P::mma(acc, blk(B::W3 + tk), HL[tk]);
        if (row < n && h == 0) {
            f32x4 o = {P::round(acc[0]), P::round(acc[1]), P::round(acc[2]), P::round(acc[3])};
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(out + row * DOUT));
        }
    }
}

template <class P, int NTH, int LAYERS, bool HP, bool FULL>   // FULL: dim_in = 32 (plane indices and store guards constant)
__global__ __launch_bounds__(kWave *kWavesPerWG, (MI3D_MLP_BWD_TIMING_CUT & 4) ? 3 : 2) void k_mlp_bwd_g(const float *__restrict__ x, uint32_t x_planes,
                                                                        const float *__restrict__ dout, uint32_t n,
                                                                        uint32_t din, Weights w, float *__restrict__ dx,
                                                                        uint32_t dx_planes, Grads g) {
    using B = Blk<NTH, LAYERS>;
    using KB = typename P::KB;
    constexpr int H = 32 * NTH;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *biasT = reinterpret_cast<float *>(lds + (size_t)B::ALL_COUNT * block_bytes<P>());
    build_blocks_g<P, NTH, LAYERS>(lds, biasT, w, (int)din, B::ALL_COUNT);
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), p = lane & 31, h = lane >> 5;
    const uint32_t wave = blockIdx.x * kWavesPerWG + threadIdx.x / kWave, n_waves = gridDim.x * kWavesPerWG;
    const uint32_t n_tiles = (n + 31) / 32;
    // the lane offsets are laundered through an empty asm ONCE PER TILE: every tile re-reads its operand blocks from
    // LDS (hipcc would otherwise hoist the loop-invariant blocks into registers and spill), at one move per tile
    int lt = lane, ht = h;
    auto blk = [&](int b) { return P::load_block(lds + (size_t)b * block_bytes<P>(), lt); };
    auto blk_fresh = [&](int b) {  // the identity blocks serve seven products per tile: re-read at every use (8 registers)
        int l = lane;
        asm volatile("" : "+v"(l));
        return P::load_block(lds + (size_t)b * block_bytes<P>(), l);
    };
    auto bias = [&](int tile) { return bias_tile(biasT, tile, ht); };
    // A tile held "lane = sample" -> the same tile held "lane = index".  Binary16 tiles go through the wave's own 2.25 KB
    // of LDS (4 ds_write_b64 + 4 ds_read_b64_tr_b16 per lane, csrc/lds_transpose.h): bit moves.  Otherwise (exact fp32)
    // the tile times an identity block on the matrix core (exact: one non-zero product per output).
    [[maybe_unused]] mi3d_tr::lds_ptr tr_wr_d = nullptr, tr_wr_x = nullptr, tr_rd = nullptr;
    if constexpr (P::kLdsTranspose) {
        mi3d_tr::lds_ptr tile = mi3d_tr::to_lds(biasT + B::BIAS_TILES * 32) + (threadIdx.x / kWave) * mi3d_tr::kTileBytes;
        tr_wr_d = tile + mi3d_tr::write_offset_d(lane);
        tr_wr_x = tile + mi3d_tr::write_offset_x(lane);
        tr_rd = tile + mi3d_tr::read_offset(lane);
    }
    auto transpose = [&](const KB &a, int id_block) {
        if constexpr (P::kLdsTranspose) {
            KB r;
            if (id_block == B::IDX) mi3d_tr::write_tile<true>(tr_wr_x, a.v[0], a.v[1]);   // (a constant at every call)
            else mi3d_tr::write_tile<false>(tr_wr_d, a.v[0], a.v[1]);
            mi3d_tr::read_tile(tr_rd, r.v[0], r.v[1]);
            return r;
        } else {
            f32x16 t = splat(0.f);
            P::mma(t, a, blk_fresh(id_block));
            return P::cast(t);
        }
    };
    const uint32_t last_plane = FULL ? (uint32_t)(DIN / 2 - 1) : din / 2 - 1;

    // weight-gradient tiles: lane = column (input-side feature), register q = row rowmap(q, h) (output-side feature)
    f32x16 gW1[NTH], gW2[NTH][NTH], gW3[NTH];
    float gb1[NTH], gb2[NTH], gb3 = 0.f;
#pragma unroll
    for (int a = 0; a < NTH; ++a) {
        gW1[a] = splat(0.f); gW3[a] = splat(0.f); gb1[a] = 0.f; gb2[a] = 0.f;
#pragma unroll
        for (int b = 0; b < NTH; ++b) gW2[a][b] = splat(0.f);
    }

    // the NEXT tile's rows are requested before this tile's products start (the second wave of the SIMD hides issue
    // gaps, not a memory round trip per tile)
    float raw[HP ? 1 : 16];
    uint32_t rawh[HP ? 8 : 1];
    f32x4 dor;
    {
        const size_t r0 = (size_t)wave * 32 + p;
        if constexpr (HP) load_rows_half_g<FULL && P::kLdsTranspose>(x, r0, n, h, x_planes, last_plane, rawh);
        else load_rows_raw_g(x, r0, n, h, x_planes, din, raw);
        dor = load_dout_raw(dout, r0, n);
    }
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const size_t row = (size_t)tile * 32 + p;
        const bool valid = row < n;
        // the plane strides are laundered per iteration: hipcc otherwise keeps one 64-bit base pointer per plane and lane
        // alive across the loop (32 registers; they were what spilled) instead of two adds per access
        uint32_t xp = x_planes, dxp = dx_planes;
        asm volatile("" : "+s"(xp), "+s"(dxp), "+v"(lt), "+v"(ht));
        KB X;
        if constexpr (HP) X = rows_kb_half<P>(rawh);
        else X = rows_kb<P>(raw);
        const KB dO = dout_kb<P>(dor, valid && h == 0);
        auto prefetch = [&]() {
            if (tile + n_waves < n_tiles) {
                const size_t rn = ((size_t)tile + n_waves) * 32 + p;
                // (LDS transposes: the eight 64-bit plane indices of the prefetch are formed here from the laundered lane
                // half instead of living across the tile as lane constants - they were what spilled, 20 registers)
                if constexpr (HP) load_rows_half_g<FULL && P::kLdsTranspose>(x, rn, n, P::kLdsTranspose ? ht : h, xp, last_plane, rawh);
                else load_rows_raw_g(x, rn, n, P::kLdsTranspose ? ht : h, xp, din, raw);
                dor = load_dout_raw(dout, rn, n);
            }
        };
        constexpr bool late_prefetch = P::kLdsTranspose && HP && MI3D_MLP_BWD_LATE_PREFETCH != 0;
        if constexpr (!late_prefetch) prefetch();
        // ---- forward recompute, lane = sample (the activations double as their own ReLU masks).  The order below keeps
        // the live set small (it is what decides spills at 256 registers): every orientation-1 tile is transposed
        // as soon as its last orientation-1 use is over, and the hidden-1 gradient is formed directly in orientation 2,
        // where its mask (the transposed activations) already is.
        KB H1p[NTH], HL[NTH];
        KB dgH1[(MI3D_MLP_BWD_TIMING_CUT & 4) ? NTH : 1];   // (timing cut 4: hidden 1 kept in the "lane = sample" orientation)
        {
            KB H1[NTH];
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
                f32x16 acc = bias(t);
                P::mma(acc, blk(B::W1 + t), X);
                H1[t] = P::relu(acc);
                if constexpr ((MI3D_MLP_BWD_TIMING_CUT & 4) != 0) dgH1[t] = H1[t];
            }
            if constexpr (LAYERS == 3) {
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    f32x16 acc = bias(NTH + t);
#pragma unroll
                    for (int tk = 0; tk < NTH; ++tk) P::mma(acc, blk(B::W2 + t * NTH + tk), H1[tk]);
                    HL[t] = P::relu(acc);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NTH; ++t) HL[t] = H1[t];
            }
            if constexpr ((MI3D_MLP_BWD_TIMING_CUT & 4) == 0) {
#pragma unroll
            for (int t = 0; t < NTH; ++t) H1p[t] = transpose(H1[t], B::IDD);  // lane = hidden-1 feature, values = samples
            }
        }
        // ---- gradient wrt the last hidden layer (lane = sample); dW_last and db_last from the transposed operands
        KB dHL[NTH];
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            f32x16 acc = splat(0.f);
            P::mma_lo(acc, blk(B::W3T + t), dO);
            dHL[t] = (MI3D_MLP_BWD_TIMING_CUT & 2) ? P::cast(acc) : P::masked(acc, HL[t]);
        }
        if constexpr ((MI3D_MLP_BWD_TIMING_CUT & 4) != 0) {
            // recompute + dgrad only: dH1 = relu'(H1) (W2^T dH2) with the W2T block as the A operand (rows = hidden-1
            // feature, lane = sample - the layout H1 has), then dX = W1^T dH1
            if constexpr (late_prefetch) prefetch();
            f32x16 accx = splat(0.f);
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
                KB dH1;
                if constexpr (LAYERS == 3) {
                    f32x16 acc = splat(0.f);
#pragma unroll
                    for (int tk = 0; tk < NTH; ++tk) P::mma(acc, blk(B::W2T + t * NTH + tk), dHL[tk]);
                    dH1 = P::masked(acc, dgH1[t]);
                } else {
                    dH1 = dHL[t];
                }
                P::mma(accx, blk(B::W1T + t), dH1);
            }
            if constexpr (HP) {
                if (valid) {
                    uint32_t *dst = reinterpret_cast<uint32_t *>(dx) + row + (size_t)(2 * h) * dxp;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        dst[(size_t)(4 * c) * dxp] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){accx[4 * c], accx[4 * c + 1]}, half2v));
                        dst[(size_t)(4 * c + 1) * dxp] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){accx[4 * c + 2], accx[4 * c + 3]}, half2v));
                    }
                }
            }
            continue;
        }
        {
            KB dOp;   // lane = output index, values = the tile's samples
            if constexpr (P::kLdsTranspose && MI3D_MLP_DO_LDS != 0) {
                dOp = transpose(dO, B::IDX);   // (kind X over the 4 outputs: through the wave's LDS like every other tile)
            } else {
                f32x16 tO = splat(0.f);
                P::mma_lo(tO, dO, blk_fresh(B::IDX));
                dOp = P::cast(tO);
            }
            if (!(MI3D_MLP_BWD_TIMING_CUT & 1)) gb3 += P::sum(dOp);
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
                const KB HLp = LAYERS == 3 ? transpose(HL[t], B::IDD) : H1p[t];
                P::mma(gW3[t], dOp, HLp);   // dW_last[o][f] += sum_s dO[s][o] H_last[s][f]
            }
        }
        KB dH1p[NTH];  // lane = hidden-1 feature, values = samples
        if constexpr (LAYERS == 3) {
#pragma unroll
            for (int ti = 0; ti < NTH; ++ti) {   // dW2[i][j] += sum_s dH2[s][i] H1[s][j]
                const KB dHLp = transpose(dHL[ti], B::IDD);
                if (!(MI3D_MLP_BWD_TIMING_CUT & 1)) gb2[ti] += P::sum(dHLp);
#pragma unroll
                for (int tj = 0; tj < NTH; ++tj) P::mma(gW2[ti][tj], dHLp, H1p[tj]);
            }
#pragma unroll
            for (int t = 0; t < NTH; ++t) {      // dH1[s][f1] = relu'(.) sum_f2 dH2[s][f2] W2[f2][f1], output lane = f1
                f32x16 acc = splat(0.f);
#pragma unroll
                for (int tk = 0; tk < NTH; ++tk) P::mma(acc, dHL[tk], blk(B::W2T + t * NTH + tk));
                dH1p[t] = (MI3D_MLP_BWD_TIMING_CUT & 2) ? P::cast(acc) : P::masked(acc, H1p[t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NTH; ++t) dH1p[t] = transpose(dHL[t], B::IDD);
        }
        {   // dW1[i][j] += sum_s dH1[s][i] X[s][j]
            const KB Xp = transpose(X, B::IDX);
            if constexpr (late_prefetch) prefetch();   // (this tile's rows are dead: the next tile's land in their registers)
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
                if (!(MI3D_MLP_BWD_TIMING_CUT & 1)) gb1[t] += P::sum(dH1p[t]);
                P::mma(gW1[t], dH1p[t], Xp);
            }
        }
        {   // ---- input gradient dX = W1^T dH1, lane = sample again
            f32x16 acc = splat(0.f);
#pragma unroll
            for (int tk = 0; tk < NTH; ++tk) {
                const KB dH1 = LAYERS == 3 ? transpose(dH1p[tk], B::IDD) : dHL[tk];
                P::mma(acc, blk(B::W1T + tk), dH1);
            }
            const bool full = FULL || din == (uint32_t)DIN;  // uniform: the usual width stores without per-plane guards
            if constexpr (HP) {
            if (valid) {
                // binary16 planes: one 4-byte store per (level, row); this IS the rounding torch.autocast gives the
                // input gradient of the first nn.Linear (a binary16 GEMM output)
                uint32_t *dxh = reinterpret_cast<uint32_t *>(dx) + row;
                uint32_t pk[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    pk[2 * c] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){acc[4 * c], acc[4 * c + 1]}, half2v));
                    pk[2 * c + 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){acc[4 * c + 2], acc[4 * c + 3]}, half2v));
                }
                if (FULL && P::kLdsTranspose) {
                    uint32_t *dst = dxh + (size_t)(2 * h) * dxp;   // one lane address; the planes a uniform stride apart
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        dst[(size_t)(4 * c) * dxp] = pk[2 * c];
                        dst[(size_t)(4 * c + 1) * dxp] = pk[2 * c + 1];
                    }
                } else if (full) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        dxh[(size_t)(4 * c + 2 * h) * dxp] = pk[2 * c];
                        dxh[(size_t)(4 * c + 2 * h + 1) * dxp] = pk[2 * c + 1];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t lvl = 4 * c + 2 * h;
                        if (lvl <= last_plane) dxh[(size_t)lvl * dxp] = pk[2 * c];
                        if (lvl + 1 <= last_plane) dxh[(size_t)(lvl + 1) * dxp] = pk[2 * c + 1];
                    }
                }
            }
            } else {
            if (valid && !dxp) {  // register q = input feature rowmap(q, h): four runs of four features
                if (full) {
                    float *dst = dx + row * DIN + 4 * h;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 o = {acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(dst + 8 * c));
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if ((uint32_t)rowmap(q, h) < din) dx[row * din + rowmap(q, h)] = acc[q];
                }
            } else if (valid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x2 a = {acc[4 * c], acc[4 * c + 1]}, b = {acc[4 * c + 2], acc[4 * c + 3]};
                    const uint32_t lvl = 4 * c + 2 * h;
                    if (lvl <= last_plane)
                        __builtin_nontemporal_store(a, reinterpret_cast<f32x2 *>(dx + ((size_t)lvl * dxp + row) * 2));
                    if (lvl + 1 <= last_plane)
                        __builtin_nontemporal_store(b, reinterpret_cast<f32x2 *>(dx + ((size_t)(lvl + 1) * dxp + row) * 2));
                }
            }
            }
        }
    }

    // ---- reduce the weight gradients across the workgroup in LDS, then one atomic per element per workgroup
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);
    const int OFF_W1 = 0, OFF_B1 = OFF_W1 + H * (int)din, OFF_W2 = OFF_B1 + H,
              OFF_B2 = OFF_W2 + (LAYERS == 3 ? H * H : 0), OFF_W3 = OFF_B2 + (LAYERS == 3 ? H : 0),
              OFF_B3 = OFF_W3 + DOUT * H, TOTAL = OFF_B3 + DOUT;
    for (int e = threadIdx.x; e < TOTAL; e += blockDim.x) red[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = rowmap(q, h);
#pragma unroll
        for (int ti = 0; ti < NTH; ++ti) {
            if (p < (int)din) atomicAdd(&red[OFF_W1 + (32 * ti + r) * (int)din + p], gW1[ti][q]);
            if constexpr (LAYERS == 3) {
#pragma unroll
                for (int tj = 0; tj < NTH; ++tj) atomicAdd(&red[OFF_W2 + (32 * ti + r) * H + 32 * tj + p], gW2[ti][tj][q]);
            }
            if (r < DOUT) atomicAdd(&red[OFF_W3 + r * H + 32 * ti + p], gW3[ti][q]);
        }
    }
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
        atomicAdd(&red[OFF_B1 + 32 * t + p], gb1[t]);
        if constexpr (LAYERS == 3) atomicAdd(&red[OFF_B2 + 32 * t + p], gb2[t]);
    }
    if (p < DOUT) atomicAdd(&red[OFF_B3 + p], gb3);
    __syncthreads();
    for (int e = threadIdx.x; e < TOTAL; e += blockDim.x) {
        const float v = red[e];
        float *dst = e < OFF_B1 ? g.dW1 + (e - OFF_W1)
                   : e < OFF_W2 ? g.db1 + (e - OFF_B1)
                   : e < OFF_B2 ? g.dW2 + (e - OFF_W2)
                   : e < OFF_W3 ? g.db2 + (e - OFF_B2)
                   : e < OFF_B3 ? g.dW3 + (e - OFF_W3)
                                : g.db3 + (e - OFF_B3);
        if (v != 0.f) unsafeAtomicAdd(dst, v);
    }
}

// ---------------------------------------------------------------- field head (elementwise, one thread per sample)
// sigma, albedo and the two finite-difference normals from the MLP output of the P = 7 or 13 stencil points
// (network_tcnn.py:94-138, activation.py:5-18, nerf/utils.py:47-48) in one pass instead of ~40 elementwise launches
// over [n, P] tensors.  Point 0 is the sample, 1..6 its +-eps neighbours (+x,-x,+y,-y,+z,-z), 7..12 those of x2.
struct HeadArgs {
    const float *x, *x2;
    float offs[MI3D_MAX_POINTS * 3];
    uint32_t P;
    float bound, blob_density, two_r2, inv_2eps;
};

__device__ __forceinline__ float head_gauss(const HeadArgs &a, const float *base, uint32_t p) {
    float d2 = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float v = fminf(a.bound, fmaxf(-a.bound, base[d] + a.offs[p * 3 + d]));
        d2 += v * v;
    }
    return a.blob_density * expf(-d2 / a.two_r2);  // divide like torch does (network_tcnn.py:98)
}

// v / sqrt(clamp(|v|^2, 1e-20, 1e32)), NaN -> 0, +-inf -> +-FLT_MAX  (safe_normalize, then torch.nan_to_num)
__device__ __forceinline__ void head_normal(const float s[6], float inv_2eps, float n[3], float v[3], float &c) {
    v[0] = -((s[0] - s[1]) * inv_2eps); v[1] = -((s[2] - s[3]) * inv_2eps); v[2] = -((s[4] - s[5]) * inv_2eps);
    const float ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    c = fminf(1e32f, fmaxf(1e-20f, ss));
    const float r = 1.0f / sqrtf(c);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float t = v[d] * r;
        if (t != t) t = 0.f;
        else if (t > 3.4028234663852886e38f) t = 3.4028234663852886e38f;
        else if (t < -3.4028234663852886e38f) t = -3.4028234663852886e38f;
        n[d] = t;
    }
}

__global__ void k_head_forward(const float4 *__restrict__ h, HeadArgs a, uint32_t n, float *__restrict__ sigma,
                               float *__restrict__ albedo, float *__restrict__ normal, float *__restrict__ normal2,
                               const int32_t *__restrict__ count) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || (count && (int32_t)s >= *count)) return;  // (device row count, optional: rows beyond it carry nothing)
    float bx[3] = {a.x[(size_t)s * 3], a.x[(size_t)s * 3 + 1], a.x[(size_t)s * 3 + 2]};
    const float4 h0 = h[s];  // rows are point-major: (sample s, point p) at p*n + s, coalesced across the wave
    sigma[s] = expf(h0.x + head_gauss(a, bx, 0));
    albedo[(size_t)s * 3] = 1.0f / (1.0f + expf(-h0.y));
    albedo[(size_t)s * 3 + 1] = 1.0f / (1.0f + expf(-h0.z));
    albedo[(size_t)s * 3 + 2] = 1.0f / (1.0f + expf(-h0.w));
    float sg[6], nn[3], v[3], c;
#pragma unroll
    for (uint32_t p = 0; p < 6; ++p) sg[p] = expf(h[(size_t)(1 + p) * n + s].x + head_gauss(a, bx, 1 + p));
    head_normal(sg, a.inv_2eps, nn, v, c);
    normal[(size_t)s * 3] = nn[0]; normal[(size_t)s * 3 + 1] = nn[1]; normal[(size_t)s * 3 + 2] = nn[2];
    if (a.P == 13) {
        float b2[3] = {a.x2[(size_t)s * 3], a.x2[(size_t)s * 3 + 1], a.x2[(size_t)s * 3 + 2]};
#pragma unroll
        for (uint32_t p = 0; p < 6; ++p) sg[p] = expf(h[(size_t)(7 + p) * n + s].x + head_gauss(a, b2, 7 + p));
        head_normal(sg, a.inv_2eps, nn, v, c);
        normal2[(size_t)s * 3] = nn[0]; normal2[(size_t)s * 3 + 1] = nn[1]; normal2[(size_t)s * 3 + 2] = nn[2];
    }
}

// gradient of one normal wrt the six pre-activations u_p = h_p0 + gauss_p (sigma_p = exp(u_p), trunc_exp backward
// evaluates exp at min(u_p, 15)); dn is the upstream gradient of the (nan_to_num'ed) normal
__device__ __forceinline__ void head_normal_backward(const float u[6], float inv_2eps, const float dn_in[3], float du[6]) {
    float sg[6], nn[3], v[3], c;
#pragma unroll
    for (int p = 0; p < 6; ++p) sg[p] = expf(u[p]);
    head_normal(sg, inv_2eps, nn, v, c);
    const float ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float r = 1.0f / sqrtf(c);
    float dn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {  // nan_to_num passes the gradient where its input is finite
        const float t = v[d] * r;
        dn[d] = (t == t && fabsf(t) <= 3.4028234663852886e38f) ? dn_in[d] : 0.f;
    }
    const float dot = dn[0] * v[0] + dn[1] * v[1] + dn[2] * v[2];
    const bool inside = ss >= 1e-20f && ss <= 1e32f;  // clamp passes the gradient inside its range
    const float k = inside ? dot * (r / c) : 0.f;      // d(1/sqrt(c))/d(ss) * 2 v = -c^-1.5 v
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float dv = r * dn[d] - k * v[d];
        const float dg = -dv * inv_2eps;  // v = -(s+ - s-) * inv_2eps
        du[2 * d] = dg * expf(fminf(u[2 * d], 15.0f));
        du[2 * d + 1] = -dg * expf(fminf(u[2 * d + 1], 15.0f));
    }
}

// dh holds the rows of the first P_active points only ([P_active * n, 4], point-major): 1 when only sigma / albedo
// carry a gradient, 7 with the normal, 13 with the jittered normal - the MLP backward and the scatter behind it then
// run over that prefix of the stencil and nothing else.
__global__ void k_head_backward(const float4 *__restrict__ h, HeadArgs a, uint32_t n, uint32_t P_active,
                                const float *__restrict__ dsigma, const float *__restrict__ dalbedo,
                                const float *__restrict__ dnormal, const float *__restrict__ dnormal2,
                                float4 *__restrict__ dh) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    float bx[3] = {a.x[(size_t)s * 3], a.x[(size_t)s * 3 + 1], a.x[(size_t)s * 3 + 2]};
    const float4 h0 = h[s];
    float4 g0;
    g0.x = (dsigma ? dsigma[s] : 0.f) * expf(fminf(h0.x + head_gauss(a, bx, 0), 15.0f));
    const float a0 = 1.0f / (1.0f + expf(-h0.y)), a1 = 1.0f / (1.0f + expf(-h0.z)), a2 = 1.0f / (1.0f + expf(-h0.w));
    g0.y = dalbedo ? dalbedo[(size_t)s * 3] * a0 * (1.0f - a0) : 0.f;
    g0.z = dalbedo ? dalbedo[(size_t)s * 3 + 1] * a1 * (1.0f - a1) : 0.f;
    g0.w = dalbedo ? dalbedo[(size_t)s * 3 + 2] * a2 * (1.0f - a2) : 0.f;
    dh[s] = g0;
    if (P_active < 7) return;
    float u[6], du[6], dn[3];
#pragma unroll
    for (uint32_t p = 0; p < 6; ++p) u[p] = h[(size_t)(1 + p) * n + s].x + head_gauss(a, bx, 1 + p);
#pragma unroll
    for (int d = 0; d < 3; ++d) dn[d] = dnormal ? dnormal[(size_t)s * 3 + d] : 0.f;
    head_normal_backward(u, a.inv_2eps, dn, du);
#pragma unroll
    for (uint32_t p = 0; p < 6; ++p) dh[(size_t)(1 + p) * n + s] = make_float4(du[p], 0.f, 0.f, 0.f);
    if (a.P == 13 && P_active == 13) {
        float b2[3] = {a.x2[(size_t)s * 3], a.x2[(size_t)s * 3 + 1], a.x2[(size_t)s * 3 + 2]};
#pragma unroll
        for (uint32_t p = 0; p < 6; ++p) u[p] = h[(size_t)(7 + p) * n + s].x + head_gauss(a, b2, 7 + p);
#pragma unroll
        for (int d = 0; d < 3; ++d) dn[d] = dnormal2 ? dnormal2[(size_t)s * 3 + d] : 0.f;
        head_normal_backward(u, a.inv_2eps, dn, du);
#pragma unroll
        for (uint32_t p = 0; p < 6; ++p) dh[(size_t)(7 + p) * n + s] = make_float4(du[p], 0.f, 0.f, 0.f);
    }
}

HeadArgs make_head_args(const float *x, const float *x2, const float *offsets_host, uint32_t P, float bound,
                        float blob_density, float blob_radius, float epsilon) {
    HeadArgs a;
    a.x = x; a.x2 = x2; a.P = P; a.bound = bound; a.blob_density = blob_density;
    a.two_r2 = (float)(2.0 * (double)blob_radius * (double)blob_radius);
    a.inv_2eps = 0.5f / epsilon;
    for (uint32_t i = 0; i < MI3D_MAX_POINTS * 3; ++i) a.offs[i] = i < P * 3 ? offsets_host[i] : 0.f;
    return a;
}

template <class P> constexpr size_t lds_bytes(int n_blocks) {
    return (size_t)n_blocks * block_bytes<P>() + (HID + HID + 32) * sizeof(float);
}
template <class P> constexpr size_t lds_bytes_g(int n_blocks, int bias_tiles, bool transposes = false) {
    return (size_t)n_blocks * block_bytes<P>() + (size_t)bias_tiles * 32 * sizeof(float) +
           (transposes && P::kLdsTranspose ? (size_t)kWavesPerWG * mi3d_tr::kTileBytes : 0u);
}

int grid_for(uint32_t n, int wgs_per_cu) {
    const uint32_t tiles = (n + 31) / 32, wgs = (tiles + kWavesPerWG - 1) / kWavesPerWG;
    const uint32_t cap = 256 * (uint32_t)wgs_per_cu;  // persistent beyond that many workgroups per CU
    return (int)(wgs < cap ? (wgs ? wgs : 1) : cap);
}

// network_tcnn.py:13-32: dim_in = 2 x levels (even, <= 32), hidden 32 or 64, 4 outputs, 2 or 3 layers
bool dims_ok(uint32_t di, uint32_t dh, uint32_t dout, uint32_t layers) {
    return di >= 2 && di <= (uint32_t)DIN && di % 2 == 0 && (dh == 32 || dh == 64) && dout == DOUT &&
           (layers == 2 || layers == 3);
}

template <class P, int NTH, int LAYERS, bool HP>
void launch_fwd(dim3 grid, hipStream_t st, const float *x, uint32_t x_planes, uint32_t n, uint32_t din, const Weights &w,
                float *out, const int32_t *count, uint32_t n_stride) {
    using B = Blk<NTH, LAYERS>;
    hipLaunchKernelGGL((k_mlp_fwd_g<P, NTH, LAYERS, HP>), grid, dim3(kWave * kWavesPerWG),
                       lds_bytes_g<P>(B::FWD_COUNT, B::BIAS_TILES), st, x, x_planes, n, din, w, out, count, n_stride);
}
template <class P, int NTH, int LAYERS, bool HP>
void launch_bwd(dim3 grid, hipStream_t st, const float *x, uint32_t x_planes, const float *dout, uint32_t n, uint32_t din,
                const Weights &w, float *dx, uint32_t dx_planes, const Grads &g) {
    using B = Blk<NTH, LAYERS>;
    // (binary16 planes at the full input width get their own instance: constant plane indices, no store guards)
    if (MI3D_MLP_BWD_FULL != 0 && HP && P::kLdsTranspose && din == (uint32_t)DIN)
        hipLaunchKernelGGL((k_mlp_bwd_g<P, NTH, LAYERS, HP, MI3D_MLP_BWD_FULL != 0 && HP && P::kLdsTranspose>), grid, dim3(kWave * kWavesPerWG),
                           lds_bytes_g<P>(B::ALL_COUNT, B::BIAS_TILES, true), st, x, x_planes, dout, n, din, w, dx, dx_planes, g);
    else
        hipLaunchKernelGGL((k_mlp_bwd_g<P, NTH, LAYERS, HP, false>), grid, dim3(kWave * kWavesPerWG),
                           lds_bytes_g<P>(B::ALL_COUNT, B::BIAS_TILES, true), st, x, x_planes, dout, n, din, w, dx, dx_planes, g);
}
// runtime (hidden width, layer count) -> the template instance
#define MI3D_MLP_DISPATCH(FN, P, HP, nth, layers, ...)                     \
    do {                                                                   \
        if ((nth) == 2 && (layers) == 3) FN<P, 2, 3, HP>(__VA_ARGS__);     \
        else if ((nth) == 2) FN<P, 2, 2, HP>(__VA_ARGS__);                 \
        else if ((layers) == 3) FN<P, 1, 3, HP>(__VA_ARGS__);              \
        else FN<P, 1, 2, HP>(__VA_ARGS__);                                 \
    } while (0)

}  // namespace

extern "C" {

int mi3d_mlp_supported(uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out, uint32_t num_layers) {
    return dims_ok(dim_in, dim_hidden, dim_out, num_layers) ? 1 : 0;
}

int mi3d_mlp_forward(const void *xv, uint32_t x_plane_rows, int planes_half, uint32_t n, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *W3, const float *b3, uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out,
                     int half_mode, float *out, void *stream) {
    return mi3d_mlp_forward_counted(xv, x_plane_rows, planes_half, n, nullptr, n ? n : 1u, W1, b1, W2, b2, W3, b3, dim_in,
                                    dim_hidden, dim_out, half_mode, out, stream);
}

int mi3d_mlp_forward_counted(const void *xv, uint32_t x_plane_rows, int planes_half, uint32_t n, const int32_t *count,
                             uint32_t n_stride, const float *W1, const float *b1, const float *W2, const float *b2,
                             const float *W3, const float *b3, uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out,
                             int half_mode, float *out, void *stream) {
    if (n_stride == 0) return (int)hipErrorInvalidValue;
    const uint32_t layers = (W2 == nullptr && b2 == nullptr) ? 2u : 3u;
    if (!dims_ok(dim_in, dim_hidden, dim_out, layers) || (x_plane_rows != 0 && x_plane_rows < n) ||
        (planes_half && (x_plane_rows == 0 || !half_mode)) || W1 == nullptr || b1 == nullptr || W3 == nullptr ||
        b3 == nullptr || (layers == 3 && (W2 == nullptr || b2 == nullptr)))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    const float *x = reinterpret_cast<const float *>(xv);
    const Weights w{W1, b1, W2, b2, W3, b3};
    const int nth = (int)dim_hidden / 32;
    hipStream_t st = as_stream(stream);
    const dim3 grid(grid_for(n, MI3D_TUNE(MI3D_T_MLP_FWD_WGS_PER_CU, 5)));
#ifdef MI3D_DEV
    if (MI3D_TUNE(MI3D_T_MLP_BWD_VARIANT, 0) == 3 && nth == 2 && layers == 3 && dim_in == (uint32_t)DIN) {  // round 2's kernel
        const dim3 g2(grid_for(n, 2));
        if (half_mode)
            hipLaunchKernelGGL(k_mlp_forward<F16>, g2, dim3(kWave * kWavesPerWG), lds_bytes<F16>(B_FWD_COUNT), st, x,
                               x_plane_rows, planes_half, n, w, out);
        else
            hipLaunchKernelGGL(k_mlp_forward<F32>, g2, dim3(kWave * kWavesPerWG), lds_bytes<F32>(B_FWD_COUNT), st, x,
                               x_plane_rows, planes_half, n, w, out);
        return (int)hipGetLastError();
    }
#endif
    if (half_mode && planes_half) MI3D_MLP_DISPATCH(launch_fwd, F16, true, nth, layers, grid, st, x, x_plane_rows, n, dim_in, w, out, count, n_stride);
    else if (half_mode) MI3D_MLP_DISPATCH(launch_fwd, F16, false, nth, layers, grid, st, x, x_plane_rows, n, dim_in, w, out, count, n_stride);
    else MI3D_MLP_DISPATCH(launch_fwd, F32, false, nth, layers, grid, st, x, x_plane_rows, n, dim_in, w, out, count, n_stride);
    return (int)hipGetLastError();
}

int mi3d_mlp_backward(const void *xv, uint32_t x_plane_rows, int planes_half, const float *dout, uint32_t n, const float *W1, const float *b1,
                      const float *W2, const float *b2, const float *W3, const float *b3, uint32_t dim_in,
                      uint32_t dim_hidden, uint32_t dim_out, int half_mode, void *dxv, uint32_t dx_plane_rows, float *dW1, float *db1, float *dW2, float *db2, float *dW3, float *db3,
                      void *stream) {
    const uint32_t layers = (W2 == nullptr && b2 == nullptr) ? 2u : 3u;
    if (!dims_ok(dim_in, dim_hidden, dim_out, layers) || (x_plane_rows != 0 && x_plane_rows < n) ||
        (dx_plane_rows != 0 && dx_plane_rows < n) ||
        (planes_half && (x_plane_rows == 0 || dx_plane_rows == 0 || !half_mode)) || W1 == nullptr || b1 == nullptr ||
        W3 == nullptr || b3 == nullptr || dW1 == nullptr || db1 == nullptr || dW3 == nullptr || db3 == nullptr ||
        (layers == 3 && (W2 == nullptr || b2 == nullptr || dW2 == nullptr || db2 == nullptr)))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    const float *x = reinterpret_cast<const float *>(xv);
    float *dx = reinterpret_cast<float *>(dxv);
    const Weights w{W1, b1, W2, b2, W3, b3};
    const Grads g{dW1, db1, dW2, db2, dW3, db3};
    const int nth = (int)dim_hidden / 32;
    const dim3 grid(grid_for(n, MI3D_TUNE(MI3D_T_MLP_WGS_PER_CU, (MI3D_MLP_BWD_TIMING_CUT & 4) ? 3 : 2))), block(kWave * kWavesPerWG);
    hipStream_t st = as_stream(stream);
    const bool classic = nth == 2 && layers == 3 && dim_in == (uint32_t)DIN;
#ifdef MI3D_DEV
    if (half_mode && classic && MI3D_TUNE(MI3D_T_MLP_BWD_VARIANT, 0) == 3) {  // round 2's kernel (one wave per SIMD, staged)
        if (planes_half)
            hipLaunchKernelGGL((k_mlp_backward<F16, 2, 1, true>), grid, block, lds_bytes<F16>(B_ALL_COUNT), st, x, x_plane_rows, 1,
                               dout, n, w, dx, dx_plane_rows, g);
        else
            hipLaunchKernelGGL((k_mlp_backward<F16, 2, 1, false>), grid, block, lds_bytes<F16>(B_ALL_COUNT), st, x, x_plane_rows, 0,
                               dout, n, w, dx, dx_plane_rows, g);
        return (int)hipGetLastError();
    }
#endif
    if (half_mode && planes_half)
        MI3D_MLP_DISPATCH(launch_bwd, F16, true, nth, layers, grid, st, x, x_plane_rows, dout, n, dim_in, w, dx, dx_plane_rows, g);
    else if (half_mode)
        MI3D_MLP_DISPATCH(launch_bwd, F16, false, nth, layers, grid, st, x, x_plane_rows, dout, n, dim_in, w, dx, dx_plane_rows, g);
    else if (classic)   // exact fp32 at full width: 16-register K-blocks do not fit two waves per SIMD - one wave, one tile
        hipLaunchKernelGGL((k_mlp_backward<F32, 1, 1, false>), grid, block, lds_bytes<F32>(B_ALL_COUNT), st, x, x_plane_rows, 0,
                           dout, n, w, dx, dx_plane_rows, g);
    else
        MI3D_MLP_DISPATCH(launch_bwd, F32, false, nth, layers, grid, st, x, x_plane_rows, dout, n, dim_in, w, dx, dx_plane_rows, g);
    return (int)hipGetLastError();
}

int mi3d_field_head_forward(const float *h, const float *x, const float *x2, uint32_t n, const float *offsets_host,
                            uint32_t P, float bound, float blob_density, float blob_radius, float epsilon, float *sigma,
                            float *albedo, float *normal, float *normal2, void *stream) {
    return mi3d_field_head_forward_counted(h, x, x2, n, nullptr, offsets_host, P, bound, blob_density, blob_radius, epsilon,
                                           sigma, albedo, normal, normal2, stream);
}

int mi3d_field_head_forward_counted(const float *h, const float *x, const float *x2, uint32_t n, const int32_t *count,
                                    const float *offsets_host, uint32_t P, float bound, float blob_density,
                                    float blob_radius, float epsilon, float *sigma, float *albedo, float *normal,
                                    float *normal2, void *stream) {
    if ((P != 7 && P != 13) || (P == 13 && (x2 == nullptr || normal2 == nullptr))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    const HeadArgs a = make_head_args(x, x2, offsets_host, P, bound, blob_density, blob_radius, epsilon);
    hipLaunchKernelGGL(k_head_forward, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(h), a, n, sigma, albedo, normal, normal2, count);
    return (int)hipGetLastError();
}

int mi3d_field_head_backward(const float *h, const float *x, const float *x2, uint32_t n, const float *offsets_host,
                             uint32_t P, uint32_t P_active, float bound, float blob_density, float blob_radius,
                             float epsilon, const float *dsigma, const float *dalbedo, const float *dnormal,
                             const float *dnormal2, float *dh, void *stream) {
    if ((P != 7 && P != 13) || (P == 13 && x2 == nullptr)) return (int)hipErrorInvalidValue;
    if ((P_active != 1 && P_active != 7 && P_active != 13) || P_active > P) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    const HeadArgs a = make_head_args(x, x2, offsets_host, P, bound, blob_density, blob_radius, epsilon);
    hipLaunchKernelGGL(k_head_backward, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(h), a, n, P_active, dsigma, dalbedo, dnormal, dnormal2,
                       reinterpret_cast<float4 *>(dh));
    return (int)hipGetLastError();
}

}  // extern "C"
