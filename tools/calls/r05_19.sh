#!/bin/bash
# round 5, GPU call 19: the fine role's flush with the record's place formed in pass 2 (-DMI3D_FLUSH_AT=1: one LDS read per
# flushed record instead of two dependent ones) against the product, product-grade builds in one process; parity on the variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_19
mkdir -p $O
# (the variant: in pass 2 `gd = gdelta[bin]` read beside the cursor claim and `(pos + gd) | bin << 26` staged as the fourth word; the flush
#  then takes `at = rec.w & 0x3FFFFFF`, `bin = rec.w >> 26` instead of `i + gdelta[rec.w]`; built with -DMI3D_FLUSH_AT=1 from a patch that was not kept)
timeout 500 python tools/scatter_ab_libs.py --libs tools/bin/libmi3d_base.so,tools/bin/libmi3d_flush_at.so --rounds 3 --out $O/scatter_ab_libs.json 2>&1 | tail -30
MI3D_LIB=$PWD/tools/bin/libmi3d_flush_at.so timeout 300 python -m pytest tests/test_grid_points_gpu.py tests/test_sds_step_gpu.py -q -x 2>&1 | tail -3
