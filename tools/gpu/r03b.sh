#!/usr/bin/env bash
# round 3, call B: validation + A/B of the new kernels (dense conv Lambda, persistent per-sample-gradient kernel, covariance on
# the wave-role-split loop), the fixed config tests, regression of the stage tests, headline bench, eigensolver lane counts.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 400 python tools/engine_ab.py ) > gpurun_out/r03b_engine_ab.log 2>&1
( timeout 300 python tools/cov_bench.py ) > gpurun_out/r03b_cov_bench.log 2>&1
( timeout 600 python -m pytest tests/test_ops_gpu.py -q ) > gpurun_out/r03b_ops.log 2>&1
( timeout 900 python -m pytest tests/test_configs_gpu.py -q -s --durations=8 ) > gpurun_out/r03b_configs.log 2>&1
( timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py tests/test_widen.py -q --durations=8 ) > gpurun_out/r03b_stages.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03b_bench.log 2>&1
for lanes in 2 4; do ( timeout 200 python tools/eigh_bench.py multi 3073 12 $lanes ) >> gpurun_out/r03b_eigh_lanes.log 2>&1; done
tail -n 4 gpurun_out/r03b_ops.log gpurun_out/r03b_configs.log gpurun_out/r03b_stages.log gpurun_out/r03b_eigh_lanes.log
tail -c 400 gpurun_out/r03b_bench.log
