#!/usr/bin/env bash
# round 3, call H: train-batch sizes of the transformer extras (P is streamed once per train batch: a larger batch raises the
# score contraction's arithmetic intensity), bounded sizes.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for tb in 512 1024; do
  ( timeout 500 python bench.py --workload bert_base --n-train 16384 --train-batch $tb --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03h_bert_tb$tb.log 2>&1
done
for tb in 128 192; do
  ( timeout 500 python bench.py --workload gpt2_small --n-train 3072 --train-batch $tb --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03h_gpt2_tb$tb.log 2>&1
done
for f in gpurun_out/r03h_*.log; do echo $f; tail -c 2500 $f | grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"peak_hbm_gib": [0-9.]*\|Error.*' | head -5; done
