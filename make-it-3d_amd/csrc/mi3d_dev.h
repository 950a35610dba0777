// Development tunables.  The product library (make-it-3d_amd/build.py) is compiled WITHOUT MI3D_DEV: every
// MI3D_TUNE(i, dflt) is then the literal default and nothing in libmi3d.so reads the environment or carries a
// debug path.  tools/build_dev.py compiles the same sources with -DMI3D_DEV into tools/bin/libmi3d_dev.so, whose
// extra entry point mi3d_dev_set(index, value) lets the kernel micro-benchmarks under tools/ A/B launch
// geometries and kernel variants on the GPU box in one call.
#pragma once

#ifdef MI3D_DEV
extern "C" int mi3d_dev_tunable[32];
#define MI3D_TUNE(i, dflt) (mi3d_dev_tunable[(i)] >= 0 ? mi3d_dev_tunable[(i)] : (int)(dflt))
#else
#define MI3D_TUNE(i, dflt) ((int)(dflt))
#endif

// indices
enum : int {
    MI3D_T_ENCODE_VARIANT = 0,     // gather: bit 0 = 16-byte pair loads, bit 1 (with bit 0) = non-temporal plane stores
    MI3D_T_ENCODE_WGS_PER_CU = 1,  // gather: workgroups per CU on the fine segments
    MI3D_T_ENCODE_ONLY_LEVEL = 2,  // gather: >= 0: the plan holds this level alone (per-level timing)
    MI3D_T_EMIT_FINE_WAVES = 3,    // scatter: emitting waves of the fine (x-pair record) role
    MI3D_T_EMIT_COARSE_WAVES = 4,  // scatter: emitting waves of the coarse (gathered per tile) role
    MI3D_T_SCATTER_LEVEL_MASK = 5, // scatter: bit l = level l is emitted (per-role / per-level timing; the reduce still
                                   //          walks every level's stale counters: a constant)
    MI3D_T_MLP_BWD_VARIANT = 6,    // MLP: 0 = round 3's kernels, 3 = round 2's
    MI3D_T_MLP_WGS_PER_CU = 7,     // MLP backward: workgroups per CU
    MI3D_T_SCATTER_MERGE = 8,      // atomic scatter (fallback path): levels that run-merge
    MI3D_T_REPLICAS = 9,           // atomic scatter (fallback path): private table copies
    MI3D_T_EMIT_ORDER = 10,        // scatter emit, bit flags:
                                   //   0x0001  the fine role walks level-major (one level's regions open at a time)
                                   //   0x0100  coarse role: gather-table 64-bit adds off        } timing only:
                                   //   0x0200  coarse role: gather-table compare-and-swap off   } the sums are
                                   //   0x0800  fine role: records sorted in LDS but not stored  } wrong with any
                                   //   0x1000  fine role: stop a chunk after pass 1 (cells, entries, histogram)
                                   //   0x2000  fine role: stop a chunk after the bin prefix sum
                                   //   0x4000  fine role: stop a chunk after the records are staged
                                   //  0x10000  coarse role: shared-face pass off (round 3's pair passes; same sums)
    MI3D_T_MLP_FWD_WGS_PER_CU = 11,       // MLP forward: workgroups per CU
    MI3D_T_ENCODE_COARSE_WGS_PER_CU = 12, // gather: workgroups per CU on the coarse segments
    MI3D_T_MARCH_RPW_MIN = 13,     // march: smallest rays per wave the launch may choose
    MI3D_T_MARCH_WAVES = 14,       // march: persistent waves
    MI3D_T_MERGE_STEPS_X10 = 15,   // scatter: a level is gathered per tile (coarse role) if its cells are >= this / 10.5
                                   //          marching steps long (product: 42 -> levels 0-6 at C2)
    MI3D_T_ENCODE_LDS_LEVELS = 16, // gather: levels served from LDS (default: as many as fit)
    MI3D_T_ENCODE_STATIC_TILES = 17,  // gather: 1 = tiles dealt statically (round 2's order) instead of claimed
    MI3D_T_ENCODE_TRIPLE = 18,     // gather: 1 = the fine hashed levels in a launch of their own that evaluates the stencil
                                   //          points differing in x only (sample, +x, -x) together (encode_group_hash)
};
