#!/bin/bash
# round 6, GPU call 26: emitting waves of the coarse role (16384 = product) 8192 / 4096 / 32768, product-grade builds, one arena
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export MI3D_SCATTER_PLACEMENT_TRIALS=1
O=gpurun_out/r06_26
mkdir -p $O
timeout 900 python tools/scatter_ab_libs.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_cw8192.so,tools/bin/libmi3d_cw4096.so,tools/bin/libmi3d_cw32768.so --rounds 3 --out $O/scatter_ab_libs_coarse_waves.json 2>&1 | grep -v amdgpu | tail -48
