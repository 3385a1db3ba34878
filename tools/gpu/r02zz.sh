set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02zz
mkdir -p $R
(time timeout 100 python bench.py --steps 5 --warmup 2 --no-extras) > $R/bench_headline.json 2> $R/bench_headline.err
(time timeout 110 python -m pytest tests/test_layer_shapes_gpu.py -m gpu -q -x 2>&1 | tail -4) > $R/pytest_layers.log 2>&1
ls -la $R
