set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02u
mkdir -p $R
(KF_PIPE=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "score or cov or syrk" 2>&1 | tail -8) > $R/pytest_pipe.log 2>&1
for pipe in 0 1; do
(KF_PIPE=$pipe timeout 300 python tools/kernel_bench.py resnet9 bert) > $R/kb_pipe$pipe.log 2>&1
(KF_PIPE=$pipe timeout 300 python tools/cov_bench.py) > $R/cov_pipe$pipe.log 2>&1
done
(KF_PIPE=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9_pipe1.json 2> $R/bench_resnet9_pipe1.err
ls -la $R
