#!/bin/bash
# round 6, GPU call 38: the gather's per-XCD, per-segment timeline against its plan (dev build)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_38
timeout 600 python tools/encode_xcd_timeline.py --out gpurun_out/r06_38/encode_xcd_timeline.json 2>&1 | grep -v amdgpu | tail -40
