#!/usr/bin/env bash
# The GPU command lines of round 6, one sub-command each:  gpurun --timeout S -- 'bash tools/gpu/r06.sh <what>'
#   rot       A/B of the blocked rotation's XCD grids (tools/r06_rot.py), twice, + FETCH_SIZE of the linear order against the grid
#   dist      the query-exchange modes end to end: 2 and 8 ranks over gloo on the one GPU (KF_QUERY_EXCHANGE=gather | replicate),
#             scores of both modes compared; the one-rank RCCL test
#   layers    per-layer times of the event-timed entry points of the ResNet-9 workload (tools/r06_layer_times.py)
#   llama32   configs[4] at FULL DEPTH on the one GPU: 32 Llama-3-8B decoder blocks (224 tracked projections, D = 6.98 G), rank-64
#             queries, 256 train x 16 query sequences of 512 tokens, one cold factor fit (96 eigenproblems of 14 336^2)
#   final     the record on the final sources: full GPU suite + smoke, kernel traces, counter passes, the driver-shaped bench line
# Everything is written under gpurun_out/ (scratch; what is kept is copied to profiles/ by hand).
set -u
what="${1:-final}"
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
rot)
    for pass in 1 2; do ( timeout 200 python tools/r06_rot.py ) 2>&1 | grep -v "^W0\|amdgpu.ids" | tee -a gpurun_out/r06_rotate_xcd_grid.log; done
    for g in 0 4; do
        ( cd /tmp && KF_ROT_FORCE=$g timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/r06_rot_pmc_$g" -- python "$R/tools/r06_rot.py" $g ) > gpurun_out/r06_rot_pmc_$g.log 2>&1
        python - "$g" <<'PY' | tee -a gpurun_out/r06_rotate_xcd_grid.log
import csv, glob, sys
g = sys.argv[1]
rows = []
for path in glob.glob(f"gpurun_out/r06_rot_pmc_{g}/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
by = {}
for r in rows:
    if "rotate_gemm_v3" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        key = (r["Grid_Size"],)
        by.setdefault(key, []).append(float(r["Counter_Value"]))
for key, vals in sorted(by.items()):
    print(f"gm={g} rotate_gemm_v3 grid {key[0]:>8s}: {len(vals):3d} launches, FETCH_SIZE x 2 (gfx950) = {2 * 1024 * sum(vals) / len(vals) / 1e9:.3f} GB per launch")
PY
        rm -rf "gpurun_out/r06_rot_pmc_$g"
    done
    ;;
dist)
    ( timeout 600 python -m pytest tests/test_distributed_gpu.py -q --durations=5 ) > gpurun_out/r06_rccl_one_rank.log 2>&1
    tail -6 gpurun_out/r06_rccl_one_rank.log
    for mode in gather replicate; do
        ( KF_QUERY_EXCHANGE=$mode KF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload gpt2_small --n-train 67 --n-query 19 --n-fit 35 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 --phase-split ) > gpurun_out/r06_two_ranks_gloo_gpt2_$mode.txt 2>&1
        tail -n 1 gpurun_out/r06_two_ranks_gloo_gpt2_$mode.txt | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('2 ranks gloo gpt2 $mode', r['value'], r['n_gpus'], json.dumps(r['exchanges'])[:700])" || tail -5 gpurun_out/r06_two_ranks_gloo_gpt2_$mode.txt
        cp bench_extras.json gpurun_out/r06_two_ranks_gloo_gpt2_${mode}_extras.json
        ( KF_QUERY_EXCHANGE=$mode KF_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --workload gpt2_small --n-train 67 --n-query 19 --n-fit 35 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r06_eight_ranks_gloo_gpt2_$mode.txt 2>&1
        tail -n 1 gpurun_out/r06_eight_ranks_gloo_gpt2_$mode.txt | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('8 ranks gloo gpt2 $mode', r['value'], r['n_gpus'], json.dumps(r['exchanges'])[:700])" || tail -5 gpurun_out/r06_eight_ranks_gloo_gpt2_$mode.txt
        ( KF_QUERY_EXCHANGE=$mode KF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 8 --n-train 1003 --n-query 37 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 --no-miopen-find ) > gpurun_out/r06_eight_ranks_gloo_resnet9_$mode.txt 2>&1
        tail -n 1 gpurun_out/r06_eight_ranks_gloo_resnet9_$mode.txt | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('8 ranks gloo resnet9 $mode', r['value'], r['n_gpus'], json.dumps(r['exchanges'])[:700])" || tail -5 gpurun_out/r06_eight_ranks_gloo_resnet9_$mode.txt
    done
    # the driver's own N > 1 command shape (auto plan; gloo stands in for RCCL on the one GPU), default line incl. other_configs.gpt2_small at tiny sizes is too long: the line's shape only
    ;;
final)
    ( timeout 2700 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r06_pytest_gpu.log 2>&1
    tail -18 gpurun_out/r06_pytest_gpu.log
    ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r06_smoke.log 2>&1
    tail -2 gpurun_out/r06_smoke.log
    CMD="python $R/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
    ( cd /tmp && KF_BENCH_BUSY=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r06_trace" -- $CMD ) > gpurun_out/r06_trace.log 2>&1
    find gpurun_out/r06_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_resnet9_n4000_kernel_stats.csv \;
    rm -rf gpurun_out/r06_trace
    for spec in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        set -- $spec; tag="$1"; shift
        ( cd /tmp && KF_BENCH_BUSY=0 timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$R/gpurun_out/r06_pmc_r9_$tag" -- $CMD ) > "gpurun_out/r06_pmc_r9_$tag.log" 2>&1
    done
    ( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r06_pmc_r9_fetch gpurun_out/r06_pmc_r9_write gpurun_out/r06_pmc_r9_mfma ) > gpurun_out/r06_pmc_resnet9_summary.log 2>&1
    cp profiles/pmc_resnet9.json gpurun_out/r06_pmc_resnet9.json
    head -c 1200 gpurun_out/r06_pmc_resnet9_summary.log
    find gpurun_out/r06_pmc_r9_fetch gpurun_out/r06_pmc_r9_write gpurun_out/r06_pmc_r9_mfma -name "*.csv" -size +2M -delete
    rm -rf gpurun_out/r06_pmc
    for w in gpt2_small bert_base; do
        for e in score cov lambda; do replay_pmc gpurun_out/r06_pmc $w $e; done
        ( python tools/pmc_entry_summary.py $w profiles/pmc_$w.json gpurun_out/r06_pmc ) > gpurun_out/r06_pmc_${w}_summary.log 2>&1
        cp profiles/pmc_$w.json gpurun_out/r06_pmc_$w.json
        grep "^==" gpurun_out/r06_pmc_${w}_summary.log
    done
    find gpurun_out/r06_pmc -name "*.csv" -size +2M -delete
    export KF_EIGH_STREAMS=1   # rocprofv3 segfaults when eight host threads launch the eigensolver's kernels at once
    for w in bert_base:2048 gpt2_small:1024; do
        name="${w%%:*}"; n="${w##*:}"
        ( cd /tmp && KF_BENCH_BUSY=0 timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r06_trace_$name" -- \
            python "$R/bench.py" --workload "$name" --n-train "$n" --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 ) > "gpurun_out/r06_trace_$name.log" 2>&1
        find "gpurun_out/r06_trace_$name" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/r06_${name}_n${n}_kernel_stats.csv" \;
        rm -rf "gpurun_out/r06_trace_$name"
    done
    unset KF_EIGH_STREAMS
    ( timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_bench_default.out 2> gpurun_out/r06_bench_default.err
    tail -n 1 gpurun_out/r06_bench_default.out | wc -c
    cp bench_extras.json gpurun_out/r06_bench_default_extras.json
    python tools/bench_digest.py bench_extras.json || tail -c 3000 gpurun_out/r06_bench_default.err
    ;;
llama32)
    # (first attempt, factor batches of 8 sequences: covariance + 352 eigenproblems went through, the Lambda stage ran out of the 288 GiB --
    #  fp32 eigenvectors 99 GB + their bf16 copies 49 GB + Lambda 28 GB + weights 29 GB + the autograd graph of 8 x 512 tokens through 32
    #  blocks; hence factor batches of 2, train batches of 4 and expandable allocator segments)
    ( KF_BENCH_BUSY=0 PYTORCH_ALLOC_CONF=expandable_segments:True PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 2700 \
        python bench.py --workload llama_block --blocks 32 --n-train 256 --n-query 16 --n-fit 64 --warm-n-train 16 \
        --factor-batch 2 --train-batch 4 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 --phase-split ) > gpurun_out/r06_bench_llama_32blocks.json 2> gpurun_out/r06_bench_llama_32blocks.log
    echo "rc $?"
    cp bench_extras.json gpurun_out/r06_bench_llama_32blocks_extras.json
    python tools/bench_digest.py bench_extras.json || tail -c 3000 gpurun_out/r06_bench_llama_32blocks.log
    ;;
layers)
    # per-layer times of the event-timed entry points (the psgdirect / psgstream steps that ran beside it measured two removed
    # experiments: profiles/r06_psg_register_stores_negative.log, commit 95a01bd)
    ( timeout 600 python tools/r06_layer_times.py resnet9 8000 ) 2>&1 | grep -v "^W0\|amdgpu.ids" | tail -60 > gpurun_out/r06_layer_times_resnet9.log
    tail -45 gpurun_out/r06_layer_times_resnet9.log
    ;;
suite)
    ( timeout 2700 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r06_pytest_gpu.log 2>&1
    tail -18 gpurun_out/r06_pytest_gpu.log
    ;;
*)
    echo "unknown sub-command $what"; exit 2 ;;
esac
