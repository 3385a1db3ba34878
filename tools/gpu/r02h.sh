set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $R
(time timeout 300 python tools/kernel_bench.py resnet9 bert gpt2) > $R/kernel_bench.log 2>&1
(time timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_layer_shapes_gpu.py -m gpu -q -k "conv2d_implicit or rows_v2 or llama" 2>&1 | tail -8) > $R/pytest_sel.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline) > $R/bench_resnet9.json 2> $R/bench_resnet9.err
ls -la $R
