#!/bin/bash
# round 5, GPU call 17: does the diffusion half's time depend on the emit's build?  (the last two driver-command lines show
# sd_guidance 24.6 ms where the earlier ones show 22.6-22.9, on different boxes.)  The same bench on ONE box under the product
# of before the register-indexing change (tools/bin/libmi3d_nofma.so) and under the final library, A B A B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_17
mkdir -p $O
for R in 1 2; do
  for V in old final; do
    L=make-it-3d_amd/csrc/libmi3d.so; [ $V = old ] && L=tools/bin/libmi3d_nofma.so
    MI3D_LIB=$PWD/$L timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --variant-steps 0 --no-cpu-baseline --no-reference-shaped > $O/bench_${V}_$R.json 2> $O/bench_${V}_$R.err
  done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_17/bench_*.json")):
    b = json.load(open(f)); k = b["kernels_ms_per_step"]
    print(f.split("/")[-1], round(b["ms_per_step"], 2), "scatter", round(k["scatter"], 2), "sd", round(k["sd_guidance"], 2), {a: round(v, 2) for a, v in b["phases_ms_per_step"].items()}, b["config"]["steps_run_total"], (b.get("clocks") or {}).get("sclk_mhz"))
P
