#!/usr/bin/env bash
# The GPU command lines of round 5, one sub-command each:  gpurun --timeout S -- 'bash tools/gpu/r05.sh <what>'
#   probe     lane semantics / LDS cycles of ds_read_b64_tr_b16 (tools/tr_probe.py), the K-major main loop stand-alone for its three
#             LDS images (tools/tn_gemm_test.py), the new kernel tests, tools/r05_ab.py ab under the kernel trace, and the PMC replays
#             of the transformer entry points (before: KF_TN=0, after: KF_TN=1)
#   check2    the persistent K-major gradient kernel: tests, A/B, counters, bounded GPT-2 / BERT bench lines
#   check3    re-validation, query passes, 4-block C5 slice with parity, 8 ranks over gloo, GPT-2 100 000 x 2 000
#   check4    fp32-row split, shared covariance increments, conv chunk experiment + counters, 4-block C5 slice, BERT / ResNet-9 lines
#   final     the record: full GPU suite + smoke, traces, counters, default bench line, 4-block C5 slice
#   final2    the same without the 4-block slice and the GPT-2 trace (re-record after the split-K rounding fix)
#   rccl      RCCL with ONE forced rank (KF_DIST_FORCE=1), side-stream test, two ranks on one GPU for the record, Llama cov / Lambda counters
#   pmc       only the PMC replays (after a kernel change) -> profiles/pmc_gpt2_small.json, pmc_bert_base.json
# Everything is written under gpurun_out/ (scratch; what is kept is copied to profiles/ by hand).
set -u
what="${1:-probe}"
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"

replay_pmc() {  # replay_pmc <out_base> <workload> <entry>: three counter passes of one entry point's calls
    local base="$1" w="$2" e="$3"
    mkdir -p "$base"
    for spec in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        set -- $spec; local tag="$1"; shift
        ( cd /tmp && timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$R/$base/${w}_${e}_$tag" -- \
            python "$R/tools/r05_ab.py" replay "$w" "$e" "$R/$base/${w}_${e}_meta.json" ) > "$base/${w}_${e}_$tag.log" 2>&1 || echo "pmc pass $w $e $tag failed"
    done
}

case "$what" in
probe)
    ( timeout 120 python tools/tr_probe.py ) > gpurun_out/r05_tr_probe.log 2>&1; echo "tr_probe rc $?"
    grep -v "^W0" gpurun_out/r05_tr_probe.log | tail -40
    ( timeout 200 python tools/tn_gemm_test.py ) > gpurun_out/r05_tn_gemm.log 2>&1; echo "tn_gemm_test rc $?"
    tail -45 gpurun_out/r05_tn_gemm.log
    ( timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "k_major or mixed_row or rows_v2 or two_segments or sequence_rows or half_tile" --durations=5 ) > gpurun_out/r05_probe_tests.log 2>&1
    tail -25 gpurun_out/r05_probe_tests.log
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r05_ab_trace" -- python "$R/tools/r05_ab.py" ab ) > gpurun_out/r05_ab.log 2>&1
    find gpurun_out/r05_ab_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_ab_kernel_stats.csv \;
    rm -rf gpurun_out/r05_ab_trace
    grep -v "^W0\|rocprof" gpurun_out/r05_ab.log | tail -30
    # counters: before (K-contiguous path) for GPT-2, after (K-major) for GPT-2 and BERT
    export KF_TN=0
    for e in score cov; do replay_pmc gpurun_out/r05_pmc_before gpt2_small $e; done
    unset KF_TN
    ( python tools/pmc_entry_summary.py gpt2_small gpurun_out/r05_pmc_gpt2_small_before.json gpurun_out/r05_pmc_before ) > gpurun_out/r05_pmc_before_summary.log 2>&1
    grep "^==" gpurun_out/r05_pmc_before_summary.log
    for w in gpt2_small bert_base; do
        for e in score cov lambda; do replay_pmc gpurun_out/r05_pmc $w $e; done
        ( python tools/pmc_entry_summary.py $w gpurun_out/r05_pmc_$w.json gpurun_out/r05_pmc ) > gpurun_out/r05_pmc_${w}_summary.log 2>&1
        grep "^==" gpurun_out/r05_pmc_${w}_summary.log
    done
    find gpurun_out/r05_pmc gpurun_out/r05_pmc_before -name "*.csv" -size +2M -delete
    ;;
check2)
    # the persistent K-major gradient kernel + folded bias columns: kernel tests, A/B under the kernel trace, counters, bounded benches
    ( timeout 200 python tools/tn_gemm_test.py ) > gpurun_out/r05_tn_gemm.log 2>&1; echo "tn_gemm_test rc $?"
    tail -12 gpurun_out/r05_tn_gemm.log
    ( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_layer_shapes_gpu.py -q --durations=8 ) > gpurun_out/r05_check2_tests.log 2>&1
    tail -20 gpurun_out/r05_check2_tests.log
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r05_ab2_trace" -- python "$R/tools/r05_ab.py" ab ) > gpurun_out/r05_ab2.log 2>&1
    find gpurun_out/r05_ab2_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_ab2_kernel_stats.csv \;
    rm -rf gpurun_out/r05_ab2_trace
    grep -v "^W0\|^E0\|rocprof" gpurun_out/r05_ab2.log | tail -24
    rm -rf gpurun_out/r05_pmc
    for w in gpt2_small bert_base; do
        for e in score cov lambda; do replay_pmc gpurun_out/r05_pmc $w $e; done
        ( python tools/pmc_entry_summary.py $w gpurun_out/r05_pmc_$w.json gpurun_out/r05_pmc ) > gpurun_out/r05_pmc_${w}_summary.log 2>&1
        grep "^==" gpurun_out/r05_pmc_${w}_summary.log
    done
    find gpurun_out/r05_pmc -name "*.csv" -size +2M -delete
    ( timeout 600 python bench.py --workload gpt2_small --n-train 4096 --n-fit 1024 --warm-n-train 256 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 1 ) > gpurun_out/r05_check2_gpt2.json 2> gpurun_out/r05_check2_gpt2.log
    python tools/bench_digest.py gpurun_out/r05_check2_gpt2.json || tail -c 2000 gpurun_out/r05_check2_gpt2.log
    ( timeout 600 python bench.py --workload bert_base --n-train 16384 --n-fit 2048 --warm-n-train 1024 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 1 ) > gpurun_out/r05_check2_bert.json 2> gpurun_out/r05_check2_bert.log
    python tools/bench_digest.py gpurun_out/r05_check2_bert.json || tail -c 2000 gpurun_out/r05_check2_bert.log
    ;;
check3)
    # folded bias columns (re-validation), query passes, multi-block C5 slice with parity, 8 ranks over gloo on one GPU, configs[3] at
    # its stated size
    ( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_layer_shapes_gpu.py -q -k "k_major or sequence_rows or rows_v2 or two_segments or layer_shape" --durations=5 ) > gpurun_out/r05_check3_ops.log 2>&1
    tail -6 gpurun_out/r05_check3_ops.log
    ( timeout 900 python -m pytest tests/test_configs_gpu.py -q -k "query_passes or plan_takes or low_rank or pairs" -s --durations=5 ) > gpurun_out/r05_check3_configs.log 2>&1
    grep -v "^W0\|amdgpu.ids" gpurun_out/r05_check3_configs.log | tail -12
    ( KF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 8 --n-train 1003 --n-query 37 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 --no-miopen-find ) > gpurun_out/r05_eight_ranks_gloo_resnet9.txt 2>&1
    grep "^{" gpurun_out/r05_eight_ranks_gloo_resnet9.txt | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('8 ranks gloo resnet9', r['value'], r['n_gpus'], json.dumps(r['exchanges'])[:900])" || tail -5 gpurun_out/r05_eight_ranks_gloo_resnet9.txt
    ( KF_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --workload gpt2_small --n-train 67 --n-query 19 --n-fit 35 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r05_eight_ranks_gloo_gpt2.txt 2>&1
    grep "^{" gpurun_out/r05_eight_ranks_gloo_gpt2.txt | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('8 ranks gloo gpt2', r['value'], r['n_gpus'], json.dumps(r['exchanges'])[:900])" || tail -5 gpurun_out/r05_eight_ranks_gloo_gpt2.txt
    ( timeout 900 python bench.py --workload llama_block --blocks 4 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r05_bench_llama_4blocks.json 2> gpurun_out/r05_bench_llama_4blocks.log
    python tools/bench_digest.py gpurun_out/r05_bench_llama_4blocks.json || tail -c 2000 gpurun_out/r05_bench_llama_4blocks.log
    ( timeout 1500 python bench.py --workload gpt2_small --n-train 100000 --n-query 2000 --n-fit 2048 --warm-n-train 512 --busy-n-train 2048 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r05_bench_gpt2_full_100k_x_2000.json 2> gpurun_out/r05_bench_gpt2_full_100k_x_2000.log
    python tools/bench_digest.py gpurun_out/r05_bench_gpt2_full_100k_x_2000.json || tail -c 2000 gpurun_out/r05_bench_gpt2_full_100k_x_2000.log
    ;;
check4)
    # fp32 rows on the bf16 engine, shared-input covariance increments (through the assembled BERT / GPT-2 pipelines), the conv
    # gradient -> score chunk experiment with counters, the 4-block C5 slice, BERT / ResNet-9 bench lines
    ( timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "fp32_rows or activation_cov or gradient_cov or sequence_covariance" --durations=5 ) > gpurun_out/r05_check4_ops.log 2>&1
    tail -6 gpurun_out/r05_check4_ops.log
    ( timeout 1200 python -m pytest tests/test_configs_gpu.py -q -k "plan_takes or assembled_model or late_layers" -s --durations=5 ) > gpurun_out/r05_check4_configs.log 2>&1
    grep -v "^W0\|amdgpu.ids" gpurun_out/r05_check4_configs.log | tail -14
    ( timeout 300 python tools/r05_ab.py convchunks ) > gpurun_out/r05_conv_chunks.log 2>&1
    grep -v "^W0\|amdgpu.ids" gpurun_out/r05_conv_chunks.log | tail -12
    for n in 1 3; do
        export KF_CONV_CHUNKS=$n
        replay_pmc gpurun_out/r05_pmc_chunks$n resnet9 score
        unset KF_CONV_CHUNKS
        ( python tools/pmc_entry_summary.py resnet9 gpurun_out/r05_pmc_resnet9_chunks$n.json gpurun_out/r05_pmc_chunks$n ) > gpurun_out/r05_pmc_chunks${n}_summary.log 2>&1
        grep "^==" gpurun_out/r05_pmc_chunks${n}_summary.log
    done
    find gpurun_out/r05_pmc_chunks1 gpurun_out/r05_pmc_chunks3 -name "*.csv" -size +2M -delete
    ( timeout 900 python bench.py --workload llama_block --blocks 4 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r05_bench_llama_4blocks.json 2> gpurun_out/r05_bench_llama_4blocks.log
    python tools/bench_digest.py gpurun_out/r05_bench_llama_4blocks.json || tail -c 2000 gpurun_out/r05_bench_llama_4blocks.log
    ( timeout 600 python bench.py --workload bert_base --n-train 16384 --n-fit 2048 --warm-n-train 1024 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 1 ) > gpurun_out/r05_check4_bert.json 2> gpurun_out/r05_check4_bert.log
    python tools/bench_digest.py gpurun_out/r05_check4_bert.json || tail -c 2000 gpurun_out/r05_check4_bert.log
    ( timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 ) > gpurun_out/r05_check4_resnet9.json 2> gpurun_out/r05_check4_resnet9.log
    python tools/bench_digest.py gpurun_out/r05_check4_resnet9.json || tail -c 2000 gpurun_out/r05_check4_resnet9.log
    ;;
final|final2)
    # the record of the round on the final sources: full GPU suite + smoke, kernel traces, counter passes (ResNet-9: the bench command
    # itself; GPT-2 / BERT: the replayed entry points), then -- with those summaries in place -- the default bench line and the
    # 4-block C5 slice
    ( timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r05_pytest_gpu.log 2>&1
    tail -18 gpurun_out/r05_pytest_gpu.log
    ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r05_smoke.log 2>&1
    tail -2 gpurun_out/r05_smoke.log
    CMD="python $R/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
    ( cd /tmp && KF_BENCH_BUSY=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r05_trace" -- $CMD ) > gpurun_out/r05_trace.log 2>&1
    find gpurun_out/r05_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_resnet9_n4000_kernel_stats.csv \;
    rm -rf gpurun_out/r05_trace
    for spec in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        set -- $spec; tag="$1"; shift
        ( cd /tmp && KF_BENCH_BUSY=0 timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$R/gpurun_out/r05_pmc_r9_$tag" -- $CMD ) > "gpurun_out/r05_pmc_r9_$tag.log" 2>&1
    done
    ( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r05_pmc_r9_fetch gpurun_out/r05_pmc_r9_write gpurun_out/r05_pmc_r9_mfma ) > gpurun_out/r05_pmc_resnet9_summary.log 2>&1
    cp profiles/pmc_resnet9.json gpurun_out/r05_pmc_resnet9.json
    head -c 1500 gpurun_out/r05_pmc_resnet9_summary.log
    find gpurun_out/r05_pmc_r9_fetch gpurun_out/r05_pmc_r9_write gpurun_out/r05_pmc_r9_mfma -name "*.csv" -size +2M -delete
    rm -rf gpurun_out/r05_pmc
    for w in gpt2_small bert_base; do
        for e in score cov lambda; do replay_pmc gpurun_out/r05_pmc $w $e; done
        ( python tools/pmc_entry_summary.py $w profiles/pmc_$w.json gpurun_out/r05_pmc ) > gpurun_out/r05_pmc_${w}_summary.log 2>&1
        cp profiles/pmc_$w.json gpurun_out/r05_pmc_$w.json
        grep "^==" gpurun_out/r05_pmc_${w}_summary.log
    done
    find gpurun_out/r05_pmc -name "*.csv" -size +2M -delete
    export KF_EIGH_STREAMS=1   # rocprofv3 segfaults when eight host threads launch the eigensolver's kernels at once
    traced="bert_base:2048 gpt2_small:1024"
    [ "$what" = "final2" ] && traced="bert_base:2048"   # (the re-record after the split-K fix: GPT-2's launches are unchanged)
    for w in $traced; do
        name="${w%%:*}"; n="${w##*:}"
        ( cd /tmp && KF_BENCH_BUSY=0 timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r05_trace_$name" -- \
            python "$R/bench.py" --workload "$name" --n-train "$n" --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 ) > "gpurun_out/r05_trace_$name.log" 2>&1
        find "gpurun_out/r05_trace_$name" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/r05_${name}_n${n}_kernel_stats.csv" \;
        rm -rf "gpurun_out/r05_trace_$name"
    done
    unset KF_EIGH_STREAMS
    ( timeout 1800 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.log
    python tools/bench_digest.py gpurun_out/r05_bench_default.json || tail -c 3000 gpurun_out/r05_bench_default.log
    [ "$what" = "final2" ] && exit 0
    ( timeout 900 python bench.py --workload llama_block --blocks 4 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r05_bench_llama_4blocks.json 2> gpurun_out/r05_bench_llama_4blocks.log
    python tools/bench_digest.py gpurun_out/r05_bench_llama_4blocks.json || tail -c 2000 gpurun_out/r05_bench_llama_4blocks.log
    ;;
pmc)
    for w in gpt2_small bert_base; do
        for e in score cov lambda; do replay_pmc gpurun_out/r05_pmc $w $e; done
        ( python tools/pmc_entry_summary.py $w gpurun_out/r05_pmc_$w.json gpurun_out/r05_pmc ) > gpurun_out/r05_pmc_${w}_summary.log 2>&1
        grep "^==" gpurun_out/r05_pmc_${w}_summary.log
    done
    find gpurun_out/r05_pmc -name "*.csv" -size +2M -delete
    ;;
rccl)
    # RCCL on the one-GPU box: a ONE-rank "nccl" group with every exchange forced (KF_DIST_FORCE=1), the gloo two-rank variant beside
    # it, the opt-in side-stream test; then -- for the record -- both ranks of the two-rank RCCL variant on the one GPU
    ( timeout 600 python -m pytest tests/test_distributed_gpu.py -q --durations=5 ) > gpurun_out/r05_rccl_one_rank.log 2>&1
    tail -12 gpurun_out/r05_rccl_one_rank.log
    ( timeout 400 python -m pytest tests/test_pipeline_gpu.py -q -k "side" ) > gpurun_out/r05_side_stream.log 2>&1
    tail -3 gpurun_out/r05_side_stream.log
    ( KF_TEST_RCCL_SHARED_GPU=1 NCCL_DEBUG=WARN timeout -k 10 240 python -m pytest tests/test_distributed_gpu.py -q -x -k "two_rank and nccl" ) \
        > gpurun_out/r05_rccl_two_ranks_one_gpu.log 2>&1
    echo "two ranks on one GPU over RCCL: rc $?"
    grep -i -m 12 "duplicate\|invalid usage\|ncclInvalid\|passed\|failed\|error" gpurun_out/r05_rccl_two_ranks_one_gpu.log | cut -c1-300
    # counters for the Llama slice's covariance / Lambda calls (its score path is the low-rank contraction: no dense score entry)
    for e in cov lambda; do replay_pmc gpurun_out/r05_pmc llama_block $e; done
    ( python tools/pmc_entry_summary.py llama_block gpurun_out/r05_pmc_llama_block.json gpurun_out/r05_pmc ) > gpurun_out/r05_pmc_llama_block_summary.log 2>&1
    grep "^==" gpurun_out/r05_pmc_llama_block_summary.log || tail -5 gpurun_out/r05_pmc_llama_block_summary.log
    find gpurun_out/r05_pmc -name "*.csv" -size +2M -delete
    ;;
tb)
    # BERT: train batch 512 (the bench's) against 1024 -- P is read once per launch, the item prologues / epilogues amortise
    for tb in 512 1024; do
        ( KF_BENCH_BUSY=0 timeout 500 python bench.py --workload bert_base --n-train 16384 --n-fit 2048 --warm-n-train 1024 --train-batch $tb \
            --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r05_bert_tb$tb.json 2> gpurun_out/r05_bert_tb$tb.log
        python tools/bench_digest.py gpurun_out/r05_bert_tb$tb.json | grep "pairs/s\|roofline:" || tail -5 gpurun_out/r05_bert_tb$tb.log
    done
    ;;
fb)
    # factor-fit batch: GPT-2 64 -> 128 sequences, BERT 256 -> 512 (the covariance / Lambda calls' prologues, epilogues and the
    # memset / finalize launches amortise over twice the rows); same n_fit as the default line, tiny score stage
    ( KF_BENCH_BUSY=0 timeout 500 python bench.py --workload gpt2_small --n-train 256 --n-query 64 --factor-batch 128 --steps 1 --warmup 1 \
        --no-cpu-baseline --factor-reps 1 ) > gpurun_out/r05_gpt2_fb128.json 2> gpurun_out/r05_gpt2_fb128.log
    python tools/bench_digest.py gpurun_out/r05_gpt2_fb128.json | grep "factor_fit\|roofline_cov:\|roofline_lambda" || tail -5 gpurun_out/r05_gpt2_fb128.log
    ( KF_BENCH_BUSY=0 timeout 500 python bench.py --workload bert_base --n-train 1024 --n-query 109 --n-fit 8192 --factor-batch 512 --steps 1 --warmup 1 \
        --no-cpu-baseline --factor-reps 1 ) > gpurun_out/r05_bert_fb512.json 2> gpurun_out/r05_bert_fb512.log
    python tools/bench_digest.py gpurun_out/r05_bert_fb512.json | grep "factor_fit\|roofline_cov\|roofline_lambda" || tail -5 gpurun_out/r05_bert_fb512.log
    ;;
fb2)
    for fb in 2000 1000; do
        ( KF_BENCH_BUSY=0 timeout 300 python bench.py --factor-batch $fb --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1 ) \
            > gpurun_out/r05_resnet9_fb$fb.json 2> gpurun_out/r05_resnet9_fb$fb.log
        python tools/bench_digest.py gpurun_out/r05_resnet9_fb$fb.json | grep "pairs/s\|factor_fit\|roofline_cov:\|roofline_lambda:" || tail -5 gpurun_out/r05_resnet9_fb$fb.log
    done
    ;;
final3)
    # re-record after the factor batches of GPT-2 / BERT doubled (kernel sources unchanged: the ResNet-9 summary stays): the replayed
    # entry points under the counters at the new batch sizes, then the default bench line with those summaries in place
    rm -rf gpurun_out/r05_pmc
    for w in gpt2_small bert_base; do
        for e in score cov lambda; do replay_pmc gpurun_out/r05_pmc $w $e; done
        ( python tools/pmc_entry_summary.py $w profiles/pmc_$w.json gpurun_out/r05_pmc ) > gpurun_out/r05_pmc_${w}_summary.log 2>&1
        cp profiles/pmc_$w.json gpurun_out/r05_pmc_$w.json
        grep "^==" gpurun_out/r05_pmc_${w}_summary.log
    done
    find gpurun_out/r05_pmc -name "*.csv" -size +2M -delete
    ( timeout 1800 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.log
    python tools/bench_digest.py gpurun_out/r05_bench_default.json || tail -c 3000 gpurun_out/r05_bench_default.log
    ;;
suite)
    ( timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r05_pytest_gpu.log 2>&1
    tail -18 gpurun_out/r05_pytest_gpu.log
    ;;
eig2)
    for lanes in 6 3 2; do
        ( timeout 400 python tools/eigh_bench.py multi 14336 6 $lanes ) 2>&1 | grep -v "^W0\|amdgpu.ids" | tee -a gpurun_out/r05_eigh_lanes_14336.log
    done
    ;;
eig)
    # does keeping several 14 336^2 eigenproblems in flight pay?  three problems on one lane / on three lanes (tools/eigh_bench.py multi)
    ( timeout 600 python -m pytest tests/test_distributed_gpu.py -q --durations=5 ) > gpurun_out/r05_rccl_one_rank.log 2>&1
    tail -4 gpurun_out/r05_rccl_one_rank.log
    for lanes in 1 3; do
        ( timeout 400 python tools/eigh_bench.py multi 14336 3 $lanes ) 2>&1 | grep -v "^W0\|amdgpu.ids" | tee -a gpurun_out/r05_eigh_lanes_14336.log
    done
    ;;
*)
    echo "unknown sub-command $what"; exit 2 ;;
esac
