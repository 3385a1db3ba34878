set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $R
(time timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "cov or syrk" 2>&1 | tail -25) > $R/pytest_cov.log 2>&1
(timeout 300 python tools/cov_bench.py) > $R/cov_big.log 2>&1
(KF_COV_TILE=128 timeout 300 python tools/cov_bench.py) > $R/cov_small.log 2>&1
(KF_SCORE_ITEMS=256 timeout 300 python tools/kernel_bench.py resnet9 bert) > $R/kb_256.log 2>&1
(KF_SCORE_ITEMS=512 timeout 300 python tools/kernel_bench.py resnet9 bert) > $R/kb_512.log 2>&1
(KF_SCORE_ITEMS=1024 timeout 300 python tools/kernel_bench.py resnet9 bert) > $R/kb_1024.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9.json 2> $R/bench_resnet9.err
ls -la $R
