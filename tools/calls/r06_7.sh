#!/bin/bash
# round 6, GPU call 7: product (tile-major fine role) vs level-major fine role over the same 10 freshly placed arenas
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_7
mkdir -p $O
timeout 900 python tools/scatter_bimodal.py --placements 8 --libs make-it-3d_amd/csrc/libmi3d.so --out $O/scatter_placements_bw.json 2>&1 | grep "^{"
