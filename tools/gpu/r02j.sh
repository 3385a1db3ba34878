set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $R
(time KF_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --n-train 8000 --n-query 200 --no-cpu-baseline) > $R/bench_2ranks_gloo.json 2> $R/bench_2ranks_gloo.err
(time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $R/pytest_gpu.log 2>&1
(time timeout 1200 python bench.py --steps 5 --warmup 2) > $R/bench_default.json 2> $R/bench_default.err
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $R/smoke.log 2>&1
ls -la $R
