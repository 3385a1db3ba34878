#!/usr/bin/env bash
# round 3, call D: where the ResNet-9 pairwise step goes (kernel trace of the bench), 2-rank dry run of the self-spawning
# bench (gloo on one GPU), the two config tests with their final bounds, low-rank GPU tests.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03d_trace" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --n-train 10000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 --no-miopen-find ) > gpurun_out/r03d_trace.log 2>&1
find gpurun_out/r03d_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03d_kernel_stats.csv \;
find gpurun_out/r03d_trace -name "*kernel_trace.csv" -delete
( KF_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03d_two_ranks_gloo.log 2>&1
( timeout 900 python -m pytest tests/test_configs_gpu.py -q -s -k "assembled" ) > gpurun_out/r03d_configs.log 2>&1
( timeout 600 python -m pytest tests/test_widen.py -q -m gpu -k "low_rank" ) > gpurun_out/r03d_lowrank.log 2>&1
tail -n 3 gpurun_out/r03d_configs.log gpurun_out/r03d_lowrank.log
tail -c 600 gpurun_out/r03d_two_ranks_gloo.log
ls -la gpurun_out/r03d_kernel_stats.csv
