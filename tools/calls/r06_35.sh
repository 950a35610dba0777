#!/bin/bash
# round 6, GPU call 35: which levels the coarse role takes, under claimed tiles (0-6 = product; 0-7; 0-5), product-grade builds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_35
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_ms38.so,tools/bin/libmi3d_ms58.so
MI3D_SCATTER_WORKSPACE_GB=64 timeout 1200 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --capture 8 --out $O/scatter_ab_libs_merge_levels.json > $O/log.txt 2>&1
python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_merge_levels.json'))
for c in ('dense_ms','real_ms','captured_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
print({k:v for k,v in d.items() if 'err' in k})
PY
tail -2 $O/log.txt
