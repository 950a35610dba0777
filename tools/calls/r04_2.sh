#!/bin/bash
# round 4, GPU call 2: the whole GPU suite (deferred point-0 scatter, graph-replayed inference loop), a bench line, the
# scatter per role / level on a real-step zero census, eval-render before / after
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04_2
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cp gpurun_out/headline_parity.json $O/ 2>/dev/null
timeout 900 python bench.py --steps 10 --warmup 3 --variant-steps 2 --no-cpu-baseline --no-reference-shaped > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
timeout 600 python tools/kbench.py --what scatter_levels --half-planes --real-census --iters 2 --out $O/kbench_scatter_levels_real.json > /dev/null 2> $O/kbench.err
timeout 600 python tools/eval_bench.py --out $O/eval_bench.json > /dev/null 2> $O/eval_bench.err
tail -5 $O/pytest.log; tail -3 $O/bench.err
