#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the round's first tree (ABI 4: compact-round eval loop, claim-counter events,
# 12-byte reduce mask fix, parity subset through the CPU oracle), the eval bench (reference rounds vs budget rounds), the
# driver's bench command (with the denoise+CLIP branch variant), smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 300 python tools/eval_bench.py --out $O/eval_bench.json > $O/eval_bench.log 2>&1
echo "eval rc=$?"; python - <<'P'
import json
try:
    r = json.load(open("gpurun_out/r05_1/eval_bench.json"))
    for k, v in r.items():
        print(k, (v["ms_median"], v["stats"]) if isinstance(v, dict) and "ms_median" in v else v)
except Exception as e:
    print("eval_bench:", e)
P
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_dense.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -4 $O/bench.err; python - <<'P'
import json
b = json.load(open("gpurun_out/r05_1/bench_c2_dense.json"))
print(b["ms_per_step"], b.get("kernels_ms_per_step"), b.get("valid"), b["roofline"]["frac"])
print(b.get("variants_ms_per_step")); print(b.get("guidance_branches"))
P
