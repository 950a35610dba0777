#!/bin/bash
# round 6, GPU call 33: the gather with / without the end-of-launch help between XCDs, product-grade builds in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_33
mkdir -p $O
timeout 600 python tools/gather_ab_libs.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_nosteal.so --out $O/gather_ab_libs_steal.json 2>&1 | grep -v amdgpu | tail -24
