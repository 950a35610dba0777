#!/bin/bash
# round 4, GPU call 3: 12-byte binary16 records + deferred point 0 (tests), the diffusion half's stock knobs, the scatter's
# role split on real steps (dev tunables), a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_3
mkdir -p $O
timeout 1200 python -m pytest tests/test_grid_points_gpu.py tests/test_sds_step_gpu.py tests/test_field_gpu.py tests/test_fullsize_gpu.py tests/test_headline_parity_gpu.py tests/test_reference_glue_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 300 python tools/sd_knobs.py --out $O/sd_knobs.json > /dev/null 2> $O/sd_knobs.err
timeout 600 python tools/step_ab.py --rounds 2 --steps 3 --configs "base:;m42:15=42;m58:15=58;m80:15=80;cw8k:4=8192;cw32k:4=32768;fw2k:3=2048;fw1k:3=1024" --out $O/step_ab_cap56.json > /dev/null 2> $O/step_ab.err
MI3D_SCATTER_WORKSPACE_GB=110 timeout 600 python tools/step_ab.py --rounds 2 --steps 3 --configs "base:;m42:15=42;m58:15=58" --out $O/step_ab_cap110.json > /dev/null 2>> $O/step_ab.err
timeout 900 python bench.py --steps 10 --warmup 3 --variant-steps 2 --no-cpu-baseline --no-reference-shaped > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -4 $O/pytest.log; tail -3 $O/bench.err
