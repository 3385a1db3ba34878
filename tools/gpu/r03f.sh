#!/usr/bin/env bash
# round 3, call F: the record of the final code -- full GPU test suite, smoke, default bench (headline + targets + full-size
# BERT / 16k GPT-2), kernel trace and the three PMC passes of the bench command (summarised into profiles/pmc_resnet9.json).
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r03f_pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r03f_smoke.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r03f_bench_default.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03f_trace" -- $CMD ) > gpurun_out/r03f_trace.log 2>&1
find gpurun_out/r03f_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03f_resnet9_n4000_kernel_stats.csv \;
find gpurun_out/r03f_trace -name "*kernel_trace.csv" -delete
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03f_pmc_fetch" -- $CMD ) > gpurun_out/r03f_pmc1.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03f_pmc_write" -- $CMD ) > gpurun_out/r03f_pmc2.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03f_pmc_mfma" -- $CMD ) > gpurun_out/r03f_pmc3.log 2>&1
( python tools/pmc_summary.py resnet9 gpurun_out/r03f_pmc_resnet9.json gpurun_out/r03f_pmc_fetch gpurun_out/r03f_pmc_write gpurun_out/r03f_pmc_mfma ) > gpurun_out/r03f_pmc_summary.log 2>&1
find gpurun_out/r03f_pmc_fetch gpurun_out/r03f_pmc_write gpurun_out/r03f_pmc_mfma -name "*.csv" -size +8M -delete
tail -n 5 gpurun_out/r03f_pytest_gpu.log gpurun_out/r03f_smoke.log
tail -c 400 gpurun_out/r03f_bench_default.log
head -c 1500 gpurun_out/r03f_pmc_summary.log
