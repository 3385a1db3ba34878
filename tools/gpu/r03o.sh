#!/usr/bin/env bash
# round 3, call O: bench line after the roofline_lambda.traffic change (same kernel sources) + per-kernel traces of one BERT-base
# and one GPT-2-small step at bounded sizes (evidence for the other configs).
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03o_bench.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03o_trace_bert" -- python "$GRAFT_REPO_ROOT/bench.py" --workload bert_base --n-train 2048 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03o_trace_bert.log 2>&1
find gpurun_out/r03o_trace_bert -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03o_bert_n2048_kernel_stats.csv \;
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03o_trace_gpt2" -- python "$GRAFT_REPO_ROOT/bench.py" --workload gpt2_small --n-train 1024 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03o_trace_gpt2.log 2>&1
find gpurun_out/r03o_trace_gpt2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03o_gpt2_n1024_kernel_stats.csv \;
find gpurun_out/r03o_trace_bert gpurun_out/r03o_trace_gpt2 -name "*kernel_trace.csv" -delete
tail -c 900 gpurun_out/r03o_bench.log
head -n 12 gpurun_out/r03o_bert_n2048_kernel_stats.csv | cut -c1-160
head -n 12 gpurun_out/r03o_gpt2_n1024_kernel_stats.csv | cut -c1-160
