#!/bin/bash
# round 6, GPU call 28: the hashed levels' region slack (1.25 = product) 1.5 / 1.1 at 3072 coarse waves, 64 GiB arena (2 slices for all)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_28
mkdir -p $O
LIBS=tools/bin/libmi3d_cw3072.so,tools/bin/libmi3d_slack15.so,tools/bin/libmi3d_slack11.so
MI3D_SCATTER_WORKSPACE_GB=64 timeout 900 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --out $O/scatter_ab_libs_slack.json > $O/log.txt 2>&1
python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_slack.json'))
for c in ('dense_ms','real_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
PY
tail -3 $O/log.txt
