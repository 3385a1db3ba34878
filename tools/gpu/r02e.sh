set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
(time timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py tests/test_distributed_gpu.py -m gpu -q -x 2>&1 | tail -60) > gpurun_out/r02e/pytest_sel.log 2>&1
(time timeout 300 python tools/kernel_bench.py resnet9 bert) > gpurun_out/r02e/kernel_bench.log 2>&1
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e/prof_eigh -- python $GRAFT_REPO_ROOT/tools/eigh_bench.py 769 3073 > $GRAFT_REPO_ROOT/gpurun_out/r02e/eigh_prof.log 2>&1)
find gpurun_out/r02e/prof_eigh -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02e/eigh_kernel_stats.csv \;
rm -rf gpurun_out/r02e/prof_eigh
(time timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline) > gpurun_out/r02e/bench_resnet9.json 2> gpurun_out/r02e/bench_resnet9.err
(time timeout 900 python bench.py --workload gpt2_small --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0) > gpurun_out/r02e/bench_gpt2.json 2> gpurun_out/r02e/bench_gpt2.err
(time timeout 900 python bench.py --workload bert_base --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0) > gpurun_out/r02e/bench_bert.json 2> gpurun_out/r02e/bench_bert.err
ls -la gpurun_out/r02e
