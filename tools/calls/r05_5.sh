#!/bin/bash
# round 5, GPU call 5: the gather's x-group evaluation (tools build, tunable 18) against the product: whole gather (three
# interleaved pairs, bit-exactness against the single-point route) and per level; the plane kernels' bit-identity test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_5
mkdir -p $O
timeout 300 python tools/kbench.py --what encode_r05 --half-planes --iters 3 --out $O/kbench_encode_r05.json > $O/kbench.log 2>&1
echo "kbench rc=$?"; tail -3 $O/kbench.log
python - <<'P'
import json
r = json.load(open("gpurun_out/r05_5/kbench_encode_r05.json"))["encode_r05"]
for k, v in r.items():
    print(f"{k:32s} {v:.3f}" if isinstance(v, float) else f"{k:32s} {v}")
P
