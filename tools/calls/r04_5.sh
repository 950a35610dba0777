#!/bin/bash
# round 4, GPU call 5: the coarse role's shared-face pass (tests, A/B dense / real census / real steps), refine trainer terms
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_5
mkdir -p $O
timeout 1200 python -m pytest tests/test_grid_points_gpu.py tests/test_field_gpu.py tests/test_fullsize_gpu.py tests/test_headline_parity_gpu.py tests/test_raster_gpu.py tests/test_sds_step_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 600 python tools/kbench.py --what scatter_ab --half-planes --iters 3 --out $O/kbench_scatter_ab.json > /dev/null 2> $O/kbench.err
timeout 900 python tools/step_ab.py --anchor --rounds 2 --steps 3 --configs "base:;noface:10=65536;m42:15=42;m42noface:15=42,10=65536" --out $O/step_ab.json > /dev/null 2> $O/step_ab.err
tail -4 $O/pytest.log; tail -2 $O/kbench.err
