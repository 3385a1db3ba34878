set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
(time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r02d/pytest_gpu.log 2>&1
(time timeout 600 python tools/eigh_bench.py) > gpurun_out/r02d/eigh_bench.log 2>&1
(time timeout 300 python tools/kernel_bench.py resnet9 bert gpt2) > gpurun_out/r02d/kernel_bench.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline) > gpurun_out/r02d/bench_resnet9.json 2> gpurun_out/r02d/bench_resnet9.err
(time timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --train-batch 2000) > gpurun_out/r02d/bench_resnet9_tb2000.json 2> gpurun_out/r02d/bench_resnet9_tb2000.err
(time timeout 900 python bench.py --workload gpt2_small --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0) > gpurun_out/r02d/bench_gpt2.json 2> gpurun_out/r02d/bench_gpt2.err
ls -la gpurun_out/r02d
