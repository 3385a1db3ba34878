set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r02a/pytest.log 2>&1
(time python bench.py --steps 3 --warmup 1) > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02a/prof_bert -- python $GRAFT_REPO_ROOT/bench.py --workload bert_base --n-train 1024 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 > $GRAFT_REPO_ROOT/gpurun_out/r02a/bench_bert_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02a/bench_bert_prof.err)
find gpurun_out/r02a/prof_bert -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02a/bert_kernel_stats.csv \;
find gpurun_out/r02a/prof_bert -type f ! -name "*stats*" -delete
ls -la gpurun_out/r02a
