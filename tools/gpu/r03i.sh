#!/usr/bin/env bash
# round 3, call I: lean per-item path of the persistent per-sample-gradient kernel (LDS row table, transposed accumulators,
# packed epilogue) + round-aware sample split of the 256-row covariance kernel: parity tests, A/B, bench.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x ) > gpurun_out/r03i_ops.log 2>&1
( timeout 300 python -m pytest tests/test_layer_shapes_gpu.py -q -k "bert-768 or gpt2-768x3073 or resnet" ) > gpurun_out/r03i_shapes.log 2>&1
( timeout 400 python tools/engine_ab.py ) > gpurun_out/r03i_engine_ab.log 2>&1
( timeout 200 python tools/cov_bench.py ) > gpurun_out/r03i_cov_bench.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03i_bench.log 2>&1
( timeout 400 python bench.py --workload gpt2_small --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/r03i_gpt2.log 2>&1
tail -n 3 gpurun_out/r03i_ops.log gpurun_out/r03i_shapes.log
grep -n "MISMATCH" gpurun_out/r03i_engine_ab.log | head
tail -n 22 gpurun_out/r03i_engine_ab.log
tail -n 12 gpurun_out/r03i_cov_bench.log
tail -c 600 gpurun_out/r03i_bench.log
