set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(time timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "conv2d_implicit or rows_v2 or odd_augmented or precondition or lambda_accum or k_tile_major" 2>&1 | tail -40) > gpurun_out/r02b/pytest_ops.log 2>&1
(time timeout 300 python tools/kernel_bench.py resnet9 bert gpt2) > gpurun_out/r02b/kernel_bench.log 2>&1
(time timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_layer_shapes_gpu.py tests/test_fullsize_gpu.py tests/test_widen.py -q 2>&1 | tail -40) > gpurun_out/r02b/pytest_pipeline.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline) > gpurun_out/r02b/bench_resnet9.json 2> gpurun_out/r02b/bench_resnet9.err
(time timeout 900 python bench.py --workload bert_base --n-train 4096 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0) > gpurun_out/r02b/bench_bert.json 2> gpurun_out/r02b/bench_bert.err
ls -la gpurun_out/r02b
