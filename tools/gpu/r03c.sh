#!/usr/bin/env bash
# round 3, call C: mixed-precision eigensolver (tests, A/B, multi-problem throughput), config tests with heuristic damping,
# the full default bench (full-size BERT, GPT-2 at 16 384 train), channels-last experiment for the ResNet-9 model passes.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "eigh" ) > gpurun_out/r03c_eigh_tests.log 2>&1
( timeout 600 python tools/eigh_bench.py 1152 2304 3073 4096 ) > gpurun_out/r03c_eigh_bench.log 2>&1
( timeout 300 python tools/eigh_bench.py multi 3073 12 4 ) >> gpurun_out/r03c_eigh_bench.log 2>&1
( timeout 900 python -m pytest tests/test_configs_gpu.py -q -s --durations=8 ) > gpurun_out/r03c_configs.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --channels-last ) > gpurun_out/r03c_bench_nhwc.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r03c_bench_default.log 2>&1
tail -n 3 gpurun_out/r03c_eigh_tests.log gpurun_out/r03c_configs.log
cat gpurun_out/r03c_eigh_bench.log
tail -c 300 gpurun_out/r03c_bench_nhwc.log
