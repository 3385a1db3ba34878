#!/usr/bin/env bash
# The GPU command lines of round 4, one sub-command each:  gpurun --timeout S -- 'bash tools/gpu/r04.sh <what>'
#   check1    the new kernels: targeted GPU tests (Lambda rows engine, half / wide score tiles, race screens, eigensolver on product
#             covariances), then tools/r04_ab.py under rocprofv3 --kernel-trace --stats (score, lambda) and its eigh part
#   check2    low-rank kernels / C5 low-rank test + bounded benches of llama_block, gpt2_small, bert_base
#   check3    pairing + late-layer tests, half-tile tests, one default bench
#   check4    the multi-layer Llama slice test
#   sidestream  A/B of the hooks' kernels on a second stream (KF_SIDE_STREAM) on three workloads
#   suite     the full GPU suite + smoke
#   record    kernel trace + the three PMC passes of the bench command (-> profiles/pmc_resnet9.json), default bench line
#   issue     request schedules of the 256 x 256 loop: agreement tests, A/B, and (if the compiled default wins) the record
#   rerecord  kernel trace + PMC passes + headline bench line of the sources as they are
#   pmc_gpt2  kernel trace (+ PMC passes: they segfault inside rocprofv3 on this workload) of a bounded GPT-2-small run
#   gpt2_full configs[3] at its full train size on one GPU
#   traces    per-kernel totals of one BERT-base and one GPT-2-small step at bounded sizes
# Everything is written under gpurun_out/ (scratch; what is kept is copied to profiles/ by hand).
set -u
what="${1:-check1}"
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
pmc() {  # pmc <tag> <counters...>: one counter-collection pass of the bench command
    local tag="$1"; shift
    ( cd /tmp && timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_pmc_$tag" -- $CMD ) > "gpurun_out/r04_pmc_$tag.log" 2>&1
}
case "$what" in
check1)
    ( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "lambda or half_tile or race_screen or eigh or score_gemm_long or rotate_bf16" --durations=8 ) > gpurun_out/r04_check1_ops.log 2>&1
    tail -5 gpurun_out/r04_check1_ops.log
    ( timeout 600 python -m pytest tests/test_configs_gpu.py -q -k "eigh" --durations=8 -s ) > gpurun_out/r04_check1_eigh.log 2>&1
    tail -5 gpurun_out/r04_check1_eigh.log
    ( timeout 600 python -m pytest tests/test_layer_shapes_gpu.py -q --durations=5 ) > gpurun_out/r04_check1_shapes.log 2>&1
    tail -3 gpurun_out/r04_check1_shapes.log
    ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_ab_trace" -- python "$GRAFT_REPO_ROOT/tools/r04_ab.py" score lambda ) > gpurun_out/r04_ab_score_lambda.log 2>&1
    find gpurun_out/r04_ab_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_ab_kernel_stats.csv \;
    rm -rf gpurun_out/r04_ab_trace
    cat gpurun_out/r04_ab_score_lambda.log | grep -v "^W0\|rocprof" | tail -30
    ( KF_EIGH_VERBOSE=1 timeout 600 python tools/r04_ab.py eigh ) > gpurun_out/r04_ab_eigh.log 2>&1
    grep -v "kf_eigh\]" gpurun_out/r04_ab_eigh.log | tail -14
    ;;
check2)
    ( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "lowrank or low_rank or eigh_small" --durations=5 ) > gpurun_out/r04_check2_ops.log 2>&1
    tail -4 gpurun_out/r04_check2_ops.log
    ( timeout 900 python -m pytest tests/test_configs_gpu.py -q -x -k "low_rank" -s --durations=5 ) > gpurun_out/r04_check2_c5.log 2>&1
    tail -8 gpurun_out/r04_check2_c5.log
    ( timeout 600 python -m pytest tests/test_widen.py tests/test_distributed_gpu.py -q --durations=5 ) > gpurun_out/r04_check2_widen.log 2>&1
    tail -4 gpurun_out/r04_check2_widen.log
    for w in llama_block gpt2_small bert_base; do
        ( timeout 900 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 1 ) > gpurun_out/r04_check2_bench_$w.json 2> gpurun_out/r04_check2_bench_$w.log
        python - "$w" <<'PY'
import json, sys
w = sys.argv[1]
try:
    line = [l for l in open(f"gpurun_out/r04_check2_bench_{w}.json") if l.startswith("{")][-1]
    r = json.loads(line)
    print(w, "value", r["value"], "ms/step", r["ms_per_step"], "peak GiB", r["peak_hbm_gib"])
    print("  factor_fit", r["factor_fit"])
    for k in ("roofline", "roofline_cov", "roofline_lambda", "roofline_lambda_update"):
        v = r.get(k) or {}
        print("  ", k, {x: v.get(x) for x in ("achieved", "frac", "launches", "avg_launch_ms", "kernel_share_of_region", "model_share_of_region")})
except Exception as e:
    print(w, "FAILED", e)
    print(open(f"gpurun_out/r04_check2_bench_{w}.log").read()[-3000:])
PY
    done
    ;;
check3)
    ( timeout 900 python -m pytest tests/test_configs_gpu.py -q -x -k "late_layers or pairs" -s --durations=5 ) > gpurun_out/r04_check3_tests.log 2>&1
    tail -8 gpurun_out/r04_check3_tests.log
    ( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "half_tile or race_screen" ) > gpurun_out/r04_check3_ops.log 2>&1
    tail -3 gpurun_out/r04_check3_ops.log
    ( timeout 1500 python bench.py ) > gpurun_out/r04_check3_bench_default.json 2> gpurun_out/r04_check3_bench_default.log
    python tools/bench_digest.py gpurun_out/r04_check3_bench_default.json || tail -c 2000 gpurun_out/r04_check3_bench_default.log
    ;;
sidestream)
    # A/B of the score kernels on a second stream (KF_SCORE_SIDE_STREAM=0: in line with the model's backward pass)
    for side in 0 1; do
        ( KF_SCORE_SIDE_STREAM=$side timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r04_side${side}_resnet9.json 2> gpurun_out/r04_side${side}_resnet9.log
        ( KF_SCORE_SIDE_STREAM=$side timeout 600 python bench.py --workload gpt2_small --n-train 4096 --n-fit 1024 --warm-n-train 256 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r04_side${side}_gpt2.json 2> gpurun_out/r04_side${side}_gpt2.log
        ( KF_SCORE_SIDE_STREAM=$side timeout 600 python bench.py --workload bert_base --n-train 16384 --n-fit 2048 --warm-n-train 1024 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r04_side${side}_bert.json 2> gpurun_out/r04_side${side}_bert.log
        for w in resnet9 gpt2 bert; do echo "== side stream $side: $w"; python tools/bench_digest.py gpurun_out/r04_side${side}_$w.json | head -3 || tail -c 1500 gpurun_out/r04_side${side}_$w.log; done
    done
    ( timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_pipeline_gpu.py -q -x -k "pairs or goldens or late_layers" ) > gpurun_out/r04_side_tests.log 2>&1
    tail -3 gpurun_out/r04_side_tests.log
    ;;
check4)
    ( timeout 900 python -m pytest tests/test_configs_gpu.py -q -x -k "reduced_width" -s --durations=3 ) > gpurun_out/r04_check4_llama.log 2>&1
    tail -6 gpurun_out/r04_check4_llama.log
    ;;
suite)
    ( timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r04_pytest_gpu.log 2>&1
    tail -15 gpurun_out/r04_pytest_gpu.log
    ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r04_smoke.log 2>&1
    tail -2 gpurun_out/r04_smoke.log
    ;;
record)
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_trace" -- $CMD ) > gpurun_out/r04_trace.log 2>&1
    find gpurun_out/r04_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_resnet9_n4000_kernel_stats.csv \;
    rm -rf gpurun_out/r04_trace
    pmc fetch FETCH_SIZE
    pmc write WRITE_SIZE
    pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
    ( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write gpurun_out/r04_pmc_mfma ) > gpurun_out/r04_pmc_summary.log 2>&1
    cp profiles/pmc_resnet9.json gpurun_out/r04_pmc_resnet9.json
    find gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write gpurun_out/r04_pmc_mfma -name "*.csv" -size +4M -delete
    ( timeout 1500 python bench.py ) > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.log
    tail -c 3000 gpurun_out/r04_bench_default.json
    ;;
issue)
    # request schedules of the 256 x 256 loop (KF_PP_ISSUE): 1. they agree with the round-3 schedule and pass the race screen,
    # 2. A/B of the three under the kernel trace, 3. when the compiled default is the fastest (within 1 %): the record of the
    # final sources -- kernel trace + PMC passes + headline bench line (the other configs take 4 more minutes: not re-run)
    ( timeout 400 python -m pytest tests/test_ops_gpu.py -q -x -k "request_schedules or wave_role_split_loop_race_screen or lambda_conv2d_dense or rotate_bf16_tall" --durations=5 ) > gpurun_out/r04_issue_tests.log 2>&1
    tail -4 gpurun_out/r04_issue_tests.log
    grep -Eq "^[0-9]+ passed" gpurun_out/r04_issue_tests.log || { echo "tests not green: stop"; exit 0; }
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_issue_trace" -- python "$GRAFT_REPO_ROOT/tools/issue_ab.py" "$GRAFT_REPO_ROOT/gpurun_out/r04_issue_ab.json" ) > gpurun_out/r04_issue_ab.log 2>&1
    find gpurun_out/r04_issue_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_issue_ab_kernel_stats.csv \;
    rm -rf gpurun_out/r04_issue_trace
    grep -v "^W0\|rocprof" gpurun_out/r04_issue_ab.log | tail -22
    go=$(python - <<'PY'
import json, re
try:
    d = json.load(open("gpurun_out/r04_issue_ab.json"))
    default = int(re.search(r"PP_ISSUE_DEFAULT = (\d)", open("kronfluence_amd/csrc/kf_score_v2.hip").read()).group(1))
    t = {int(k): v for k, v in d["totals_ms"].items()}
    print("go" if d["all_equal"] and t[default] <= 1.01 * min(t.values()) else "stop")
except Exception as exc:
    print("stop", exc)
PY
)
    echo "decision: $go"
    [ "$go" = "go" ] || exit 0
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_trace" -- $CMD ) > gpurun_out/r04_trace.log 2>&1
    find gpurun_out/r04_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_resnet9_n4000_kernel_stats.csv \;
    rm -rf gpurun_out/r04_trace
    pmc fetch FETCH_SIZE
    pmc write WRITE_SIZE
    pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
    ( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write gpurun_out/r04_pmc_mfma ) > gpurun_out/r04_pmc_summary.log 2>&1
    cp profiles/pmc_resnet9.json gpurun_out/r04_pmc_resnet9.json
    find gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write gpurun_out/r04_pmc_mfma -name "*.csv" -size +4M -delete
    ( timeout 400 python bench.py --no-extras ) > gpurun_out/r04_bench_headline.json 2> gpurun_out/r04_bench_headline.log
    python tools/bench_digest.py gpurun_out/r04_bench_headline.json || tail -c 1500 gpurun_out/r04_bench_headline.log
    ;;
rerecord)
    # kernel trace + PMC passes + headline bench line of the sources as they are (after a change of the compiled default)
    ( timeout 200 python -m pytest tests/test_ops_gpu.py -q -x -k "wave_role_split_loop_race_screen" ) > gpurun_out/r04_rerecord_tests.log 2>&1
    tail -2 gpurun_out/r04_rerecord_tests.log
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_trace" -- $CMD ) > gpurun_out/r04_trace.log 2>&1
    find gpurun_out/r04_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_resnet9_n4000_kernel_stats.csv \;
    rm -rf gpurun_out/r04_trace
    pmc fetch FETCH_SIZE
    pmc write WRITE_SIZE
    pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
    ( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write gpurun_out/r04_pmc_mfma ) > gpurun_out/r04_pmc_summary.log 2>&1
    cp profiles/pmc_resnet9.json gpurun_out/r04_pmc_resnet9.json
    find gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write gpurun_out/r04_pmc_mfma -name "*.csv" -size +4M -delete
    ( timeout 400 python bench.py --no-extras ) > gpurun_out/r04_bench_headline.json 2> gpurun_out/r04_bench_headline.log
    python tools/bench_digest.py gpurun_out/r04_bench_headline.json || tail -c 1500 gpurun_out/r04_bench_headline.log
    ;;
pmc_gpt2)
    # the transformer kernels under the counters: three PMC passes + a kernel trace of a bounded GPT-2-small run
    export KF_EIGH_STREAMS=1   # one host thread launches kernels: rocprofv3 segfaulted twice with the eigen stage's eight
    G2="python $GRAFT_REPO_ROOT/bench.py --workload gpt2_small --n-train 512 --n-fit 256 --steps 1 --warmup 1 --warm-n-train 128 --no-cpu-baseline --factor-reps 0"
    ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_trace_g2" -- $G2 ) > gpurun_out/r04_trace_g2.log 2>&1
    find gpurun_out/r04_trace_g2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_gpt2_n512_kernel_stats.csv \;
    rm -rf gpurun_out/r04_trace_g2
    for spec in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        set -- $spec; tag="$1"; shift
        ( cd /tmp && timeout 500 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_pmcg2_$tag" -- $G2 ) > "gpurun_out/r04_pmcg2_$tag.log" 2>&1
    done
    ( python tools/pmc_summary.py gpt2_small profiles/pmc_gpt2_small.json gpurun_out/r04_pmcg2_fetch gpurun_out/r04_pmcg2_write gpurun_out/r04_pmcg2_mfma ) > gpurun_out/r04_pmcg2_summary.log 2>&1
    cp profiles/pmc_gpt2_small.json gpurun_out/r04_pmc_gpt2_small.json
    find gpurun_out/r04_pmcg2_fetch gpurun_out/r04_pmcg2_write gpurun_out/r04_pmcg2_mfma -name "*.csv" -size +4M -delete
    head -c 2500 gpurun_out/r04_pmcg2_summary.log
    ;;
gpt2_full)
    # configs[3] at its full train size on ONE GPU (the 8-GPU configuration's per-rank shard is an eighth of this): 100 000 x 1 024
    ( timeout 1500 python bench.py --workload gpt2_small --n-train 100000 --n-fit 2048 --warm-n-train 512 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r04_bench_gpt2_full_100k.json 2> gpurun_out/r04_bench_gpt2_full_100k.log
    python tools/bench_digest.py gpurun_out/r04_bench_gpt2_full_100k.json || tail -c 2000 gpurun_out/r04_bench_gpt2_full_100k.log
    ;;
traces)
    export KF_EIGH_STREAMS=1   # rocprofv3 segfaults when eight host threads launch the eigensolver's kernels at once
    for w in bert_base:2048 gpt2_small:1024; do
        name="${w%%:*}"; n="${w##*:}"
        ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_trace_$name" -- \
            python "$GRAFT_REPO_ROOT/bench.py" --workload "$name" --n-train "$n" --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 0 ) > "gpurun_out/r04_trace_$name.log" 2>&1
        find "gpurun_out/r04_trace_$name" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/r04_${name}_n${n}_kernel_stats.csv" \;
        rm -rf "gpurun_out/r04_trace_$name"
        tail -c 1500 "gpurun_out/r04_trace_$name.log"
    done
    ;;
*)
    echo "unknown sub-command $what"; exit 2 ;;
esac
