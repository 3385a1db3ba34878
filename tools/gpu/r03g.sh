#!/usr/bin/env bash
# round 3, call G: kernel trace + the three PMC passes of the bench command on the FINAL kernel sources, summarised into
# profiles/pmc_resnet9.json, then the headline bench line reading it (roofline.traffic / mfma_util from the same code).
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "conv2d or lambda or race" ) > gpurun_out/r03g_ops.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03g_trace" -- $CMD ) > gpurun_out/r03g_trace.log 2>&1
find gpurun_out/r03g_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03g_resnet9_n4000_kernel_stats.csv \;
find gpurun_out/r03g_trace -name "*kernel_trace.csv" -delete
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03g_pmc_fetch" -- $CMD ) > gpurun_out/r03g_pmc1.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03g_pmc_write" -- $CMD ) > gpurun_out/r03g_pmc2.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03g_pmc_mfma" -- $CMD ) > gpurun_out/r03g_pmc3.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03g_pmc_stalls" -- python "$GRAFT_REPO_ROOT/tools/kernel_bench.py" resnet9 ) > gpurun_out/r03g_pmc4.log 2>&1
( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r03g_pmc_fetch gpurun_out/r03g_pmc_write gpurun_out/r03g_pmc_mfma ) > gpurun_out/r03g_pmc_summary.log 2>&1
cp profiles/pmc_resnet9.json gpurun_out/r03g_pmc_resnet9.json
( python tools/pmc_dump.py gpurun_out/r03g_pmc_stalls ) > gpurun_out/r03g_pmc_stalls.txt 2>&1
find gpurun_out/r03g_pmc_fetch gpurun_out/r03g_pmc_write gpurun_out/r03g_pmc_mfma gpurun_out/r03g_pmc_stalls -name "*.csv" -size +4M -delete
( timeout 400 python bench.py --steps 5 --warmup 2 --no-extras ) > gpurun_out/r03g_bench_headline.log 2>&1
tail -n 3 gpurun_out/r03g_ops.log
head -c 1200 gpurun_out/r03g_pmc_summary.log
head -n 30 gpurun_out/r03g_pmc_stalls.txt
tail -c 300 gpurun_out/r03g_bench_headline.log
