#!/usr/bin/env bash
# round 3, call N: the record of the final code -- full GPU test suite, smoke, kernel trace + the three PMC passes of the
# bench command (summarised into profiles/pmc_resnet9.json with the kernel-source hash), stall counters, then the default bench
# (headline + targets + full-size BERT / 16k GPT-2) reading that file, and the C5 slice (one Llama-3-8B MLP projection, full width).
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r03n_pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r03n_smoke.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03n_trace" -- $CMD ) > gpurun_out/r03n_trace.log 2>&1
find gpurun_out/r03n_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03n_resnet9_n4000_kernel_stats.csv \;
find gpurun_out/r03n_trace -name "*kernel_trace.csv" -delete
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03n_pmc_fetch" -- $CMD ) > gpurun_out/r03n_pmc1.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03n_pmc_write" -- $CMD ) > gpurun_out/r03n_pmc2.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03n_pmc_mfma" -- $CMD ) > gpurun_out/r03n_pmc3.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03n_pmc_stalls" -- python "$GRAFT_REPO_ROOT/tools/kernel_bench.py" resnet9 ) > gpurun_out/r03n_pmc4.log 2>&1
( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r03n_pmc_fetch gpurun_out/r03n_pmc_write gpurun_out/r03n_pmc_mfma ) > gpurun_out/r03n_pmc_summary.log 2>&1
cp profiles/pmc_resnet9.json gpurun_out/r03n_pmc_resnet9.json
( python tools/pmc_dump.py gpurun_out/r03n_pmc_stalls ) > gpurun_out/r03n_pmc_stalls.txt 2>&1
find gpurun_out/r03n_pmc_fetch gpurun_out/r03n_pmc_write gpurun_out/r03n_pmc_mfma gpurun_out/r03n_pmc_stalls -name "*.csv" -size +4M -delete
( timeout 1200 python bench.py ) > gpurun_out/r03n_bench_default.log 2>&1
( timeout 200 python tools/llama_layer.py up --skip-big-eigh ) > gpurun_out/r03n_llama_up.log 2>&1
( timeout 200 python tools/llama_layer.py down --skip-big-eigh ) > gpurun_out/r03n_llama_down.log 2>&1
tail -n 5 gpurun_out/r03n_pytest_gpu.log gpurun_out/r03n_smoke.log
head -c 1500 gpurun_out/r03n_pmc_summary.log
tail -c 500 gpurun_out/r03n_bench_default.log
tail -n 14 gpurun_out/r03n_llama_up.log
