set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
(time timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "eigh or conv2d_implicit or rows_v2 or odd_augmented or low_rank" 2>&1 | tail -30) > gpurun_out/r02c/pytest_ops.log 2>&1
(time timeout 600 python tools/eigh_bench.py) > gpurun_out/r02c/eigh_bench.log 2>&1
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02c/prof_kb -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py resnet9 bert gpt2 > $GRAFT_REPO_ROOT/gpurun_out/r02c/kernel_bench_prof.log 2>&1)
find gpurun_out/r02c/prof_kb -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02c/kernel_bench_kernel_stats.csv \;
rm -rf gpurun_out/r02c/prof_kb
(time timeout 900 python bench.py --workload gpt2_small --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0) > gpurun_out/r02c/bench_gpt2.json 2> gpurun_out/r02c/bench_gpt2.err
ls -la gpurun_out/r02c
