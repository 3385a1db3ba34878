set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02r
mkdir -p $R
(time timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "syrk or cov or lambda or precondition or gemm" 2>&1 | tail -8) > $R/pytest_ops.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9.json 2> $R/bench_resnet9.err
(time timeout 900 python bench.py --workload bert_base --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 1) > $R/bench_bert.json 2> $R/bench_bert.err
(time timeout 900 python bench.py --workload gpt2_small --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 1) > $R/bench_gpt2.json 2> $R/bench_gpt2.err
ls -la $R
