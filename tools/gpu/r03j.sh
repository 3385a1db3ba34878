#!/usr/bin/env bash
# round 3, call J: where does a score step go OUTSIDE the entry points?  One extra step of each config under the torch profiler
# (bench.py KF_BENCH_PROFILE, after the timed region): per-kernel device totals.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( KF_BENCH_PROFILE=gpurun_out/r03j_prof_resnet9.txt timeout 300 python bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03j_resnet9.log 2>&1
( KF_BENCH_PROFILE=gpurun_out/r03j_prof_gpt2.txt timeout 400 python bench.py --workload gpt2_small --n-train 512 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03j_gpt2.log 2>&1
( KF_BENCH_PROFILE=gpurun_out/r03j_prof_bert.txt timeout 400 python bench.py --workload bert_base --n-train 2048 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03j_bert.log 2>&1
tail -c 300 gpurun_out/r03j_resnet9.log gpurun_out/r03j_gpt2.log gpurun_out/r03j_bert.log
ls -la gpurun_out/r03j_prof_*
