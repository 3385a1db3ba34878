set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $R
(timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "cov or syrk or score" 2>&1 | tail -4) > $R/pytest.log 2>&1
(timeout 200 python tools/cov_bench.py) > $R/cov_bench.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9.json 2> $R/bench_resnet9.err
ls -la $R
