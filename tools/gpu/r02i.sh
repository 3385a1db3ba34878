set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $R
(time timeout 900 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --miopen-find) > $R/bench_resnet9_find.json 2> $R/bench_resnet9_find.err
(time timeout 300 python tools/eigh_bench.py 769 3073) > $R/eigh_bench.log 2>&1
ls -la $R
