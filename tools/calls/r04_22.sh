#!/bin/bash
# round 4, GPU call 22: the driver's bench command + smoke at the final HEAD (bench.py's labels / traffic_source changed
# after call 20's validation)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_22
mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -2 $O/smoke.log; tail -3 $O/bench.err; python - <<'P'
import json
b = json.load(open("gpurun_out/r04_22/bench_c2_dense.json"))
print(b["ms_per_step"], b.get("kernels_ms_per_step"), b.get("valid"), b["roofline"]["frac"], b["roofline"].get("traffic_source"))
P
