// A 32 x 32 binary16 tile turned round inside one wave through LDS (gfx950: ds_read_b64_tr_b16).
//
// The MLP kernels of field.hip hold a tile in one of two orientations - "lane = sample, the lane's 16 values = 16
// features" or "lane = feature, values = samples" - and the weight gradients need the second one of every tile the
// forward / backward chain produced in the first.  Round 3 turned a tile on the matrix core (a product with an identity
// block: 2 MFMAs into a zeroed accumulator tile + 8 packed converts, 20 of the backward's 61 MFMAs per tile).  Here the
// wave writes the tile row-major into 2.25 KB of its own LDS (4 ds_write_b64 per lane) and reads it back through the
// transposing read (4 ds_read_b64_tr_b16 per lane): bit moves, no arithmetic, nothing on the matrix core.
//
// ds_read_b64_tr_b16: every lane supplies the address of 8 bytes (4 binary16 values); inside each group of 16 lanes the
// 16 x 4 values are exchanged so that lane t, value j receives what lane 4 j + (t >> 2) loaded as its value t & 3 - the
// group reads a [4][16] row-major block and every lane leaves with one COLUMN of it.  (tools/tr_probe.hip checks that on
// the chip, and the whole tile round trip below.)
//
// Tile image: M[sample s][feature f], a row = 32 features = 64 bytes, rows kRowBytes = 72 bytes apart (the 8 bytes of
// padding spread the 32 rows a half-wave writes in one instruction over all 32 banks; 64-byte rows would put them on
// two).  The value order of a lane is field.hip's: value q of lane half h is index rowmap(q, h) = (q & 3) + 8 (q >> 2)
// + 4 h ("kind D": what an MFMA output tile holds) or 16 h + q ("kind X": rows as they are loaded from memory).  Either
// way four consecutive values are four consecutive indices: one 8-byte chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mi3d_common.h"

namespace mi3d_tr {

constexpr int kRowBytes = mi3d::kTrRowBytes;   // (the index math lives in mi3d_common.h, where the host tests reach it)
constexpr int kTileBytes = 32 * kRowBytes;  // per wave

using half4v = __attribute__((ext_vector_type(4))) _Float16;
using half8v = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(3))) char *lds_ptr;
typedef __attribute__((address_space(3))) half4v *lds_half4_ptr;
using short4v = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) short4v *lds_short4_ptr;
// a generic pointer into the workgroup's LDS as a 32-bit LDS address (ds_* with immediate offsets from there on)
__device__ __forceinline__ lds_ptr to_lds(void *p) { return (lds_ptr)p; }
__device__ __forceinline__ half4v tr_read(lds_ptr p) {
    return __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_short4_ptr>(p)));
}

// byte offsets of a lane inside its wave's tile image (mi3d_common.h)
__device__ __forceinline__ uint32_t write_offset_d(int lane) { return mi3d::tr_write_offset_d(lane); }
__device__ __forceinline__ uint32_t write_offset_x(int lane) { return mi3d::tr_write_offset_x(lane); }
__device__ __forceinline__ uint32_t read_offset(int lane) { return mi3d::tr_read_offset(lane); }

// the lane's 16 values (two 8-value vectors, as field.hip's F16::KB holds them) into the image.  wr = tile + write_offset_*
template <bool KIND_X>
__device__ __forceinline__ void write_tile(lds_ptr wr, const half8v &v0, const half8v &v1) {
    constexpr int step = KIND_X ? 8 : 16;  // kind D: chunk c = features 8 c + 4 h ...; kind X: 16 h + 4 c ...
    *reinterpret_cast<lds_half4_ptr>(wr + 0 * step) = __builtin_shufflevector(v0, v0, 0, 1, 2, 3);
    *reinterpret_cast<lds_half4_ptr>(wr + 1 * step) = __builtin_shufflevector(v0, v0, 4, 5, 6, 7);
    *reinterpret_cast<lds_half4_ptr>(wr + 2 * step) = __builtin_shufflevector(v1, v1, 0, 1, 2, 3);
    *reinterpret_cast<lds_half4_ptr>(wr + 3 * step) = __builtin_shufflevector(v1, v1, 4, 5, 6, 7);
}

// ... and back, turned round: lane = feature (lane & 31), value q of lane half h' = sample rowmap(q, h').  rd = tile +
// read_offset.  A wave's LDS instructions execute in order, so the reads see the writes; the empty asm statements keep
// the COMPILER from moving either across the other (to it, a lane reading what another lane wrote is a race).
__device__ __forceinline__ void read_tile(lds_ptr rd, half8v &v0, half8v &v1) {
    asm volatile("" ::: "memory");
    const half4v a = tr_read(rd + 0 * 8 * kRowBytes);
    const half4v b = tr_read(rd + 1 * 8 * kRowBytes);
    const half4v c = tr_read(rd + 2 * 8 * kRowBytes);
    const half4v d = tr_read(rd + 3 * 8 * kRowBytes);
    asm volatile("" ::: "memory");
    v0 = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    v1 = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
}

}  // namespace mi3d_tr
