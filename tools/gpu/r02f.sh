set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
(time timeout 300 python tools/kernel_bench.py resnet9 bert) > gpurun_out/r02f/kernel_bench.log 2>&1
(time timeout 300 python tools/eigh_bench.py 769 3073 4096) > gpurun_out/r02f/eigh_bench.log 2>&1
(time timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -m gpu -q -s -k "full_size or eigh or conv2d_implicit" 2>&1 | tail -30) > gpurun_out/r02f/pytest_sel.log 2>&1
(time timeout 1200 python bench.py --steps 5 --warmup 2) > gpurun_out/r02f/bench_default.json 2> gpurun_out/r02f/bench_default.err
ls -la gpurun_out/r02f
