set -x
cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_sds_step_gpu.py -x -q 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03_c2_dense.json 2> gpurun_out/bench_r03_c2_dense.err
tail -6 gpurun_out/bench_r03_c2_dense.err
for w in c2_pruned c4_views c5_refine; do
  python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-reference-shaped > gpurun_out/bench_r03_$w.json 2> gpurun_out/bench_r03_$w.err
  tail -2 gpurun_out/bench_r03_$w.err
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c4 -- python bench.py --workload c4_views --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/bench_c4_prof.err
python tools/trace_sum.py gpurun_out/prof_c4 --out gpurun_out/kernel_stats_r03_c4_views.csv | head -8
rm -rf gpurun_out/prof_c4
