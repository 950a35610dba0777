#!/bin/bash
# round 6, GPU call 45: hashed-level region slack 1.25 (product) / 1.18 / 1.15 with THREE slices (33 GiB arena), product builds in one
# process; then the driver's command (side legs off) with slack 1.18 at a 31 GiB cap (3 slices, peak <= 56 GiB), 7 placement trials
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_45
mkdir -p $O
LIBS=make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_slk118.so,tools/bin/libmi3d_slk115.so
MI3D_SCATTER_WORKSPACE_GB=33 timeout 1200 python tools/scatter_ab_libs.py --libs $LIBS --rounds 3 --capture 8 --out $O/scatter_ab_libs_slack_3slices.json > $O/log.txt 2>&1
python - <<PY
import json
d=json.load(open('$O/scatter_ab_libs_slack_3slices.json'))
for c in ('dense_ms','real_ms','captured_ms'):
    print(c, {k.replace('libmi3d','').replace('.so',''):round(min(v),2) for k,v in d[c].items()})
PY
for i in 1 2; do
MI3D_LIB=tools/bin/libmi3d_slk118.so MI3D_SCATTER_PLACEMENT_TRIALS=7 MI3D_SCATTER_WORKSPACE_GB=31 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-shaped --variant-steps 0 > $O/bench_31_$i.json 2> $O/bench_31_$i.err
python - <<PY
import json
d=json.loads(open('$O/bench_31_$i.json').read().strip().splitlines()[-1])
print('31 GiB slack 1.18', round(d['ms_per_step'],2), 'scatter', round(d['kernels_ms_per_step']['scatter'],2), 'dense', round(d['scatter_dense_gradients']['ms'],2), 'peak', round(d['peak_mem_GiB'],1), d['scatter_arena_placement'][0]['candidates_ms'], 'valid', d['valid'])
PY
done
