#!/bin/bash
# round 6, GPU call 37: reduce workgroups per average bin (4 = product) 3 / 6 / 8, product-grade builds in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_37
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_split3.so,tools/bin/libmi3d_split6.so,tools/bin/libmi3d_split8.so
timeout 1200 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --capture 8 --out $O/scatter_ab_libs_reduce_split.json > $O/log.txt 2>&1
python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_reduce_split.json'))
for c in ('dense_ms','real_ms','captured_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
PY
tail -2 $O/log.txt
