#!/usr/bin/env bash
# round 3, call A: new parity tests (assembled BERT / GPT-2, eigh 3073 / 4096, full-width Llama, GPT-2 shapes), engine A/B,
# regression of the ops touched by the new main loop, headline bench with both engines and a larger train batch.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 300 python tools/engine_ab.py ) > gpurun_out/r03a_engine_ab.log 2>&1
echo "engine_ab rc=$?" >> gpurun_out/r03a_engine_ab.log
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "score or rotate or lambda or precondition or gemm" ) > gpurun_out/r03a_ops.log 2>&1
( timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_layer_shapes_gpu.py -q -s --durations=20 ) > gpurun_out/r03a_configs.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03a_bench_e3.log 2>&1
( KF_ENGINE=2 timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03a_bench_e2.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --train-batch 2000 ) > gpurun_out/r03a_bench_e3_tb2000.log 2>&1
tail -5 gpurun_out/r03a_engine_ab.log gpurun_out/r03a_ops.log gpurun_out/r03a_configs.log
tail -c 600 gpurun_out/r03a_bench_e3.log
