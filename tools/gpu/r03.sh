#!/usr/bin/env bash
# The GPU command lines of round 3, one sub-command each:  gpurun --timeout S -- 'bash tools/gpu/r03.sh <what>'
#   record       the record of the final code: full GPU suite, smoke, kernel trace + the three PMC passes of the bench command
#                (summarised into profiles/pmc_resnet9.json with the kernel-source hash), stall counters, default bench, C5 slice
#   traces       per-kernel totals of one BERT-base and one GPT-2-small step at bounded sizes
#   ab           A/B of the round-2 and round-3 kernels through the entry points, covariance kernels, a headline bench line
#   profile      one extra step of each single-GPU config under the torch profiler (KF_BENCH_PROFILE): where a step goes outside
#                the entry points, and the census of copy-like operators of a ResNet-9 step
#   train-batch  larger train batches for the transformer configs
# Everything is written under gpurun_out/ (scratch; what is kept is copied to profiles/ by hand).
set -u
what="${1:-record}"
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
pmc() {  # pmc <tag> <counters...>: one counter-collection pass of the bench command
    local tag="$1"; shift
    ( cd /tmp && timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03_pmc_$tag" -- $CMD ) > "gpurun_out/r03_pmc_$tag.log" 2>&1
}
case "$what" in
record)
    ( timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r03_pytest_gpu.log 2>&1
    ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r03_smoke.log 2>&1
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03_trace" -- $CMD ) > gpurun_out/r03_trace.log 2>&1
    find gpurun_out/r03_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_resnet9_n4000_kernel_stats.csv \;
    find gpurun_out/r03_trace -name "*kernel_trace.csv" -delete
    pmc fetch FETCH_SIZE
    pmc write WRITE_SIZE
    pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
    ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv \
        -d "$GRAFT_REPO_ROOT/gpurun_out/r03_pmc_stalls" -- python "$GRAFT_REPO_ROOT/tools/kernel_bench.py" resnet9 ) > gpurun_out/r03_pmc_stalls.log 2>&1
    ( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write gpurun_out/r03_pmc_mfma ) > gpurun_out/r03_pmc_summary.log 2>&1
    cp profiles/pmc_resnet9.json gpurun_out/r03_pmc_resnet9.json
    ( python tools/pmc_dump.py gpurun_out/r03_pmc_stalls ) > gpurun_out/r03_pmc_stalls.txt 2>&1
    find gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write gpurun_out/r03_pmc_mfma gpurun_out/r03_pmc_stalls -name "*.csv" -size +4M -delete
    ( timeout 1200 python bench.py ) > gpurun_out/r03_bench_default.log 2>&1
    ( timeout 200 python tools/llama_layer.py up --skip-big-eigh ) > gpurun_out/r03_llama_layer_up.log 2>&1
    ( timeout 200 python tools/llama_layer.py down --skip-big-eigh ) > gpurun_out/r03_llama_layer_down.log 2>&1
    tail -n 5 gpurun_out/r03_pytest_gpu.log gpurun_out/r03_smoke.log
    head -c 1500 gpurun_out/r03_pmc_summary.log
    tail -c 500 gpurun_out/r03_bench_default.log
    ;;
traces)
    for spec in "bert_base 2048" "gpt2_small 1024"; do
        set -- $spec
        ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03_trace_$1" -- \
            python "$GRAFT_REPO_ROOT/bench.py" --workload "$1" --n-train "$2" --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > "gpurun_out/r03_trace_$1.log" 2>&1
        find "gpurun_out/r03_trace_$1" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/r03_$1_n$2_kernel_stats.csv" \;
        find "gpurun_out/r03_trace_$1" -name "*kernel_trace.csv" -delete
        head -n 12 "gpurun_out/r03_$1_n$2_kernel_stats.csv" | cut -c1-160
    done
    ;;
ab)
    ( timeout 400 python tools/engine_ab.py ) > gpurun_out/r03_engine_ab.log 2>&1
    ( timeout 200 python tools/cov_bench.py ) > gpurun_out/r03_cov_bench.log 2>&1
    ( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03_bench_headline.log 2>&1
    grep -n "MISMATCH" gpurun_out/r03_engine_ab.log | head
    tail -n 30 gpurun_out/r03_engine_ab.log
    tail -n 12 gpurun_out/r03_cov_bench.log
    tail -c 600 gpurun_out/r03_bench_headline.log
    ;;
profile)
    ( KF_BENCH_PROFILE=gpurun_out/r03_prof_resnet9.txt KF_BENCH_PROFILE_STACKS=1 timeout 300 python bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03_prof_resnet9.log 2>&1
    ( KF_BENCH_PROFILE=gpurun_out/r03_prof_gpt2.txt timeout 400 python bench.py --workload gpt2_small --n-train 512 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03_prof_gpt2.log 2>&1
    ( KF_BENCH_PROFILE=gpurun_out/r03_prof_bert.txt timeout 400 python bench.py --workload bert_base --n-train 2048 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03_prof_bert.log 2>&1
    ls -la gpurun_out/r03_prof_*.txt
    ;;
train-batch)
    for spec in "bert_base 16384 512" "bert_base 16384 1024" "gpt2_small 3072 128" "gpt2_small 3072 192"; do
        set -- $spec
        ( timeout 500 python bench.py --workload "$1" --n-train "$2" --train-batch "$3" --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > "gpurun_out/r03_$1_tb$3.log" 2>&1
        tail -c 300 "gpurun_out/r03_$1_tb$3.log"
    done
    ;;
*)
    echo "unknown sub-command $what" >&2
    exit 2
    ;;
esac
