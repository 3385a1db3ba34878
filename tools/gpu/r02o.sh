set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02o
mkdir -p $R
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "cov or syrk" 2>&1 | tail -25) > $R/pytest_cov.log 2>&1
(timeout 300 python tools/cov_bench.py) > $R/cov_bench.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9.json 2> $R/bench_resnet9.err
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_resnet -- python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --factor-reps 1) > $R/prof_resnet.log 2>&1
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_bert -- python $GRAFT_REPO_ROOT/bench.py --workload bert_base --n-train 1024 --n-query 64 --steps 1 --warmup 0 --no-cpu-baseline --factor-reps 1) > $R/prof_bert.log 2>&1
cd $R; for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -40 $f | cut -c1-200; done > $R/stats_head.txt
find $R -name "*kernel_trace.csv" -delete; find $R -name "*agent_info.csv" -delete
ls -la $R
