#!/bin/bash
# round 5, GPU call 12: relative register indexing of a tile's gradient pairs (dev bit 0x80000) against the select chains; (call 10 was the masked fma and the
# reduce's scale folded into the weight, against the product; scatter parity tests on the tools build's product defaults
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_12
mkdir -p $O
python tools/clock_log.py --out $O/clocks.jsonl --period 0.2 &
CL=$!
timeout 500 python tools/kbench.py --what scatter_r05 --half-planes --iters 3 --out $O/kbench_scatter_r05.json > $O/kbench.log 2>&1
echo "kbench rc=$?"
kill $CL
python - <<'P'
import json
r = json.load(open("gpurun_out/r05_12/kbench_scatter_r05.json"))
for k, v in r["scatter_r05_ms"].items():
    print(f"{k:40s} {v:.3f}" if isinstance(v, float) and v > 1e-3 else f"{k:40s} {v}")
P
timeout 300 python -m pytest tests/test_grid_points_gpu.py -q -x 2>&1 | tail -3
