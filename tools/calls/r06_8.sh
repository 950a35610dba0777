#!/bin/bash
# round 6, GPU call 8: the dense scatter and plain fill / read bandwidth on 8 freshly placed arenas (two processes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_8
mkdir -p $O
timeout 900 python tools/scatter_bimodal.py --placements 8 --libs make-it-3d_amd/csrc/libmi3d.so --out $O/scatter_placements_bw.json 2>&1 | grep "^{"
timeout 900 python tools/scatter_bimodal.py --placements 8 --libs make-it-3d_amd/csrc/libmi3d.so --out $O/scatter_placements_bw2.json 2>&1 | grep "^{"
