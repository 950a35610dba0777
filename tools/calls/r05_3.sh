#!/bin/bash
# round 5, GPU call 3: the coarse role's run-merged group flush (dev bit 0x20000) against the product, alone and in the whole
# scatter, dense gradients and the real census; clocks / power sampled beside it (tools/clock_log.py)
#   python tools/build_dev.py   (before the call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_3
mkdir -p $O
python tools/clock_log.py --out $O/clocks.jsonl --period 0.2 &
CL=$!
timeout 500 python tools/kbench.py --what scatter_r05 --half-planes --iters 3 --out $O/kbench_scatter_r05.json > $O/kbench.log 2>&1
echo "kbench rc=$?"
kill $CL
python - <<'P'
import json
r = json.load(open("gpurun_out/r05_3/kbench_scatter_r05.json"))
for k, v in r["scatter_r05_ms"].items():
    print(f"{k:40s} {v:.3f}" if isinstance(v, float) and v > 1e-3 else f"{k:40s} {v}")
cl = [json.loads(l) for l in open("gpurun_out/r05_3/clocks.jsonl") if l.strip()]
print(len(cl), "clock samples;", cl[:2], cl[-1:])
for name, t0, t1 in r["scatter_r05_stamps"]:
    s = [c for c in cl if t0 <= c.get("t", 0) <= t1 and "sclk_mhz" in c]
    if s:
        print(f"{name:40s} sclk {sum(c['sclk_mhz'] for c in s)/len(s):7.0f} MHz  power {sum(c.get('power_w', 0) for c in s)/len(s):6.0f} W  ({len(s)} samples)")
P
