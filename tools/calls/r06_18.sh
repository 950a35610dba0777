#!/bin/bash
# round 6, GPU call 18: can better arena placements be found by shifting (a spacer in front of a fresh arena)?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_18
mkdir -p $O
timeout 900 python tools/scatter_bimodal.py --shift 16 --shift-gib 6 --out $O/scatter_shift.json 2>&1 | grep "^{"
timeout 900 python tools/scatter_bimodal.py --shift 12 --shift-gib 1 --out $O/scatter_shift_1gib.json 2>&1 | grep "^{"
