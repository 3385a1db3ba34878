set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02m
mkdir -p $R
for ring in 0 4 5; do
(KF_SCORE_RING=$ring timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "score" 2>&1 | tail -15) > $R/pytest_score_ring$ring.log 2>&1
(KF_SCORE_RING=$ring timeout 300 python tools/kernel_bench.py resnet9 bert) > $R/kb_ring$ring.log 2>&1
done
(timeout 300 python tools/cov_bench.py) > $R/cov_big.log 2>&1
(KF_COV_TILE=128 timeout 300 python tools/cov_bench.py) > $R/cov_small.log 2>&1
(KF_SCORE_RING=4 KF_COV_TILE=128 timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9_ring4.json 2> $R/bench_resnet9_ring4.err
(KF_SCORE_RING=5 KF_COV_TILE=128 timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9_ring5.json 2> $R/bench_resnet9_ring5.err
ls -la $R
