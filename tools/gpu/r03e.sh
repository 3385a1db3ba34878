#!/usr/bin/env bash
# round 3, call E: LDS offset tables for the implicit-im2col requests (cov v2 / v3, per-sample gradients v2 / v3), XCD-aware
# Lambda kernel, reworked assembled-model test; A/B tools and the headline bench on the result.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -q ) > gpurun_out/r03e_ops.log 2>&1
( timeout 300 python tools/cov_bench.py ) > gpurun_out/r03e_cov_bench.log 2>&1
( timeout 400 python tools/engine_ab.py ) > gpurun_out/r03e_engine_ab.log 2>&1
( timeout 900 python -m pytest tests/test_configs_gpu.py -q -s -k "assembled" ) > gpurun_out/r03e_configs.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03e_bench.log 2>&1
tail -n 3 gpurun_out/r03e_ops.log gpurun_out/r03e_configs.log
cat gpurun_out/r03e_cov_bench.log
tail -n 14 gpurun_out/r03e_engine_ab.log
tail -c 500 gpurun_out/r03e_bench.log
