#!/usr/bin/env bash
# round 3, call K: per-sample gradients of long contractions on the 256 x 256 loop (tests + A/B) and the copy census of a
# ResNet-9 step (torch profiler with shapes and call sites).
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "rows or score" ) > gpurun_out/r03k_ops.log 2>&1
( timeout 300 python -m pytest tests/test_layer_shapes_gpu.py -q -k "gpt2 or llama" ) > gpurun_out/r03k_shapes.log 2>&1
( timeout 400 python tools/engine_ab.py ) > gpurun_out/r03k_engine_ab.log 2>&1
( KF_BENCH_PROFILE=gpurun_out/r03k_prof_resnet9.txt KF_BENCH_PROFILE_STACKS=1 timeout 300 python bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03k_resnet9.log 2>&1
tail -n 3 gpurun_out/r03k_ops.log gpurun_out/r03k_shapes.log
grep -n "MISMATCH" gpurun_out/r03k_engine_ab.log | head
grep -A5 "transformer score entry" gpurun_out/r03k_engine_ab.log
grep -A70 "copy-like operators by input shape" gpurun_out/r03k_prof_resnet9.txt | cut -c1-220
