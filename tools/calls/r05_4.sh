#!/bin/bash
# round 5, GPU call 4: the whole GPU suite with the run-merged group flush as the product default and the autocast output
# bounds tightened to 3 x the recorded errors; the driver's bench command; clocks / power sampled beside the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_4
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
python tools/clock_log.py --out $O/clocks_bench.jsonl --period 0.2 &
CL=$!
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
kill $CL
tail -3 $O/bench.err; python - <<'P'
import json
b = json.load(open("gpurun_out/r05_4/bench_c2_dense.json"))
print(b["ms_per_step"], b.get("kernels_ms_per_step"), b.get("valid"), b["roofline"]["frac"])
print(b.get("variants_ms_per_step"))
cl = [json.loads(l) for l in open("gpurun_out/r05_4/clocks_bench.jsonl") if l.strip()]
s = [c for c in cl if "sclk_mhz" in c]
print(len(cl), "samples", cl[:1], "sclk min/max", min(c["sclk_mhz"] for c in s) if s else None, max(c["sclk_mhz"] for c in s) if s else None,
      "power max", max(c.get("power_w", 0) for c in cl))
P
