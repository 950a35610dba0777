#!/bin/bash
# round 4, GPU call 23: the probe of the transposing LDS read with the row-pitch sweep (standalone binary, built here:
#   hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/bin/tr_probe)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_23
timeout 40 tools/bin/tr_probe > gpurun_out/r04_23/tr_probe.txt 2>&1
cat gpurun_out/r04_23/tr_probe.txt
