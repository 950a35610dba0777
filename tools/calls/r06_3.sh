#!/bin/bash
# round 6, GPU call 3: the failing new test with its traceback; does the dense scatter's time depend on the sample count n
# (item 1d, under a kernel trace: which kernel moves); the fine role's pass 1 without its arithmetic = the upper bound of
# what an index plane / delta hash could buy (items 1a / 1b); the MLP backward's recompute + dgrad alone at 3 waves / SIMD
# (item 4's closing experiment).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_3
mkdir -p $O
timeout 300 python -m pytest tests/test_raymarching_gpu.py -q -x -k compact_budget 2>&1 | tail -40
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_sweep -o sw --output-format csv -- python $GRAFT_REPO_ROOT/tools/scatter_bimodal.py --sweep-n 40 --out $GRAFT_REPO_ROOT/$O/scatter_sweep_n.json 2>&1 | grep sweep )
python tools/scatter_bimodal.py --per-dispatch $O/trace_sweep > $O/scatter_sweep_n_dispatches.json 2>&1
rm -rf $O/trace_sweep
timeout 600 python tools/scatter_ab_libs.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_fake_pass1.so --rounds 3 --out $O/scatter_ab_libs_fake_pass1.json 2>&1 | tail -30
timeout 600 python tools/mlp_ab.py --libs make-it-3d_amd/csrc/libmi3d.so,tools/bin/libmi3d_mlp_dgrad_only.so --timing-only --out $O/mlp_ab_dgrad_only.json 2>&1 | tail -30
