#!/bin/bash
# round 6, GPU call 39: the gather under the re-fitted cost table: per-XCD timeline (dev build) and product builds A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_39
timeout 600 python tools/encode_xcd_timeline.py --out gpurun_out/r06_39/encode_xcd_timeline_fit6.json 2>&1 | grep -v amdgpu | tail -32
timeout 600 python tools/gather_ab_libs.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_fit3.so --out gpurun_out/r06_39/gather_ab_libs_fit.json 2>&1 | grep -v amdgpu | tail -22
