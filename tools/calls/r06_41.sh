#!/bin/bash
# round 6, GPU call 41: the gather with all eight XCDs walking the levels together (one claim counter per level) against the
# per-XCD segments of equal modelled cost, product builds in one process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_41
timeout 600 python tools/gather_ab_libs.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_shared.so --out gpurun_out/r06_41/gather_ab_libs_shared.json 2>&1 | grep -v amdgpu | tail -22
