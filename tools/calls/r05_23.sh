#!/bin/bash
# round 5, GPU call 23: smoke and the scatter / field / raymarching / MLP tests on the last build (a comment-only change to
# hashgrid.hip since call 22's full suite)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_23
mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python -m pytest tests/test_grid_points_gpu.py tests/test_field_gpu.py tests/test_raymarching_gpu.py tests/test_mlp_gpu.py tests/test_hashgrid_gpu.py -q -x 2>&1 | tail -3
