set -x
python -m pytest tests/test_mlp_gpu.py tests/test_field_gpu.py tests/test_grid_points_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/c2_tests.log
python tools/kbench.py --what mlp --half-planes --out gpurun_out/kb_mlp_r03a.json > gpurun_out/kb_mlp_r03a.log 2>&1
python tools/kbench.py --what scatter_levels --half-planes --out gpurun_out/kb_scatter_levels_r03a.json > gpurun_out/kb_scatter_levels_r03a.log 2>&1
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err
tail -3 gpurun_out/c2_tests.log; tail -5 gpurun_out/bench_r03a.err
