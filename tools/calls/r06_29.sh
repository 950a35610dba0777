#!/bin/bash
# round 6, GPU call 29: whole steps in one process (dev build, one placed arena): 16384 coarse waves against 3072, and
# 3072 with the arena capped at 47 GiB (2 slices) / 31 GiB (3 slices, the new fewest-slices rule) / 24 GiB (4 slices);
# then the grid tests on the new product library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_29
mkdir -p $O
timeout 1200 python tools/step_ab.py --anchor --rounds 3 --steps 3 --configs "cw3072:4=3072;cw16384:4=16384;cap47:4=3072,cap=47;cap31:4=3072,cap=31;cap24:4=3072,cap=24" --out $O/step_ab_coarse_waves_caps.json > $O/log.txt 2>&1
python - <<PY
import json
d=json.load(open('$O/step_ab_coarse_waves_caps.json'))
for k,v in d['configs'].items():
    print(k, 'ms', round(v['ms'],2), 'scatter', round(v['scatter'],2), 'vs anchor ms', round(v.get('ms_vs_anchor',0),2), 'scatter', round(v.get('scatter_vs_anchor',0),2))
PY
tail -2 $O/log.txt
timeout 900 python -m pytest tests/test_grid_points_gpu.py -m gpu -x -q 2>&1 | tail -3
