#!/bin/bash
# round 4, GPU call 4: GroupNorm knob with the native backward; anchored A/B of the scatter's role split, fine waves and arena cap
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_4
mkdir -p $O
timeout 300 python tools/sd_knobs.py --out $O/sd_knobs.json > /dev/null 2> $O/sd_knobs.err
timeout 900 python tools/step_ab.py --anchor --rounds 2 --steps 3 --configs "base:;m42:15=42;m58:15=58;fw1k:3=1024;c32:cap=32;c24:cap=24;m58fw1k:15=58,3=1024" --out $O/step_ab.json > /dev/null 2> $O/step_ab.err
timeout 600 python -m pytest tests/test_sd_branches_cpu.py tests/test_sds_step_gpu.py -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log; tail -3 $O/step_ab.err
