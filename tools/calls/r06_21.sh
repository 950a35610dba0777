#!/bin/bash
# round 6, GPU call 21: 24-bit multiplies in the emit (MI3D_MUL24: the gather table's slot hash, the spatial hash's c * prime as
# two signed 24-bit products) against the 32-bit ones, product-grade builds in one process on one arena; scatter parity
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export MI3D_SCATTER_PLACEMENT_TRIALS=1
O=gpurun_out/r06_21
mkdir -p $O
timeout 600 python tools/scatter_ab_libs.py --libs tools/bin/libmi3d_mul24_0.so,make-it-3d_amd/csrc/libmi3d.so --rounds 4 --out $O/scatter_ab_libs_mul24.json 2>&1 | grep -v amdgpu | tail -32
unset MI3D_SCATTER_PLACEMENT_TRIALS
timeout 900 python -m pytest tests/test_grid_points_gpu.py tests/test_hashgrid_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -3
