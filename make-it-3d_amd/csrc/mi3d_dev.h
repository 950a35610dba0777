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
    MI3D_T_ENCODE_VARIANT = 0,     // bit 0: 16-byte pair loads, bit 1 (with bit 0): non-temporal plane stores
    MI3D_T_ENCODE_WGS_PER_CU = 1,
    MI3D_T_ENCODE_ONLY_LEVEL = 2,  // >= 0: the plan holds this level alone (per-level timing)
    MI3D_T_EMIT_FINE_WAVES = 3,
    MI3D_T_EMIT_COARSE_WAVES = 4,
    MI3D_T_SCATTER_LEVEL_MASK = 5,
    MI3D_T_MLP_BWD_VARIANT = 6,
    MI3D_T_MLP_WGS_PER_CU = 7,
    MI3D_T_SCATTER_MERGE = 8,
    MI3D_T_REPLICAS = 9,
    MI3D_T_EMIT_ORDER = 10,
    MI3D_T_MLP_FWD_WGS_PER_CU = 11,
    MI3D_T_ENCODE_COARSE_WGS_PER_CU = 12,
    MI3D_T_MARCH_RPW_MIN = 13,
    MI3D_T_MARCH_WAVES = 14,
    MI3D_T_ENCODE_STATIC_TILES = 17,  // gather: 1 = tiles dealt statically (round 2's order) instead of claimed
    MI3D_T_ENCODE_LDS_LEVELS = 16,  // gather: levels served from LDS (default: as many as fit)
    MI3D_T_MERGE_STEPS_X10 = 15,   // scatter: a level is gathered per tile if its cells are >= this / 10 marching steps        // 1: the fine emit role walks level-major (one level's regions open at a time)
};
