#!/bin/bash
# round 6, GPU call 47: what the float atomics that end a reduce workgroup cost (timing-only build without them), and 2 / 3 reduce
# workgroups per bin, at three slices (33 GiB) and at four (25 GiB); product-grade builds in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_47
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_noflush.so,tools/bin/libmi3d_split2.so,tools/bin/libmi3d_split3.so
for GB in 33 25; do
MI3D_SCATTER_WORKSPACE_GB=$GB timeout 1200 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --capture 8 --out $O/scatter_ab_libs_flush_$GB.json > $O/log_$GB.txt 2>&1
python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_flush_$GB.json'))
print('$GB GiB')
for c in ('dense_ms','real_ms','captured_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
PY
done
