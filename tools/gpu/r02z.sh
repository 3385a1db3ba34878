set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02z
mkdir -p $R
(timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "rotate or lambda or precondition" 2>&1 | tail -6) > $R/pytest.log 2>&1
(timeout 200 python tools/rotate_bench.py) > $R/rotate_bench.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9.json 2> $R/bench_resnet9.err
(timeout 400 python -m pytest tests/test_fullsize_gpu.py -m gpu -q 2>&1 | tail -4) > $R/pytest_fullsize.log 2>&1
ls -la $R
