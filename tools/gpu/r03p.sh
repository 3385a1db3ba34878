#!/usr/bin/env bash
# round 3, call P: the 256 x 256 gradient kernel from K = 128 on (BERT's T): A/B against the 128 x 128 kernel (KF_PSG_PP_MIN_K=256),
# parity tests of the sequence paths, then the PMC passes + headline on these sources (profiles/pmc_resnet9.json).
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 300 python tools/engine_ab.py ) > gpurun_out/r03p_engine_ab_k128.log 2>&1
( KF_PSG_PP_MIN_K=256 timeout 300 python tools/engine_ab.py ) > gpurun_out/r03p_engine_ab_k256.log 2>&1
( timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_layer_shapes_gpu.py -q -x -k "rows or score or bert or gpt2 or llama or precondition" ) > gpurun_out/r03p_tests.log 2>&1
( timeout 300 python bench.py --workload bert_base --n-train 4096 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03p_bert_k128.log 2>&1
( KF_PSG_PP_MIN_K=256 timeout 300 python bench.py --workload bert_base --n-train 4096 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03p_bert_k256.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03p_pmc_fetch" -- $CMD ) > gpurun_out/r03p_pmc1.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03p_pmc_write" -- $CMD ) > gpurun_out/r03p_pmc2.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03p_pmc_mfma" -- $CMD ) > gpurun_out/r03p_pmc3.log 2>&1
( python tools/pmc_summary.py resnet9 profiles/pmc_resnet9.json gpurun_out/r03p_pmc_fetch gpurun_out/r03p_pmc_write gpurun_out/r03p_pmc_mfma ) > gpurun_out/r03p_pmc_summary.log 2>&1
cp profiles/pmc_resnet9.json gpurun_out/r03p_pmc_resnet9.json
find gpurun_out/r03p_pmc_fetch gpurun_out/r03p_pmc_write gpurun_out/r03p_pmc_mfma -name "*.csv" -size +4M -delete
( timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03p_bench.log 2>&1
for f in k128 k256; do echo "== $f"; grep -A5 "transformer score entry" gpurun_out/r03p_engine_ab_$f.log | cut -c1-150; done
tail -n 3 gpurun_out/r03p_tests.log
for f in gpurun_out/r03p_bert_k128.log gpurun_out/r03p_bert_k256.log gpurun_out/r03p_bench.log; do python - "$f" <<'PY'
import sys, json
s = open(sys.argv[1]).read(); i = s.rfind('{"metric')
if i < 0: print(sys.argv[1], "NO JSON", s[-300:])
else:
    d = json.loads(s[i:].strip().splitlines()[0]); print(sys.argv[1], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "avg_ms", d["roofline"]["avg_launch_ms"], "traffic", d["roofline"].get("traffic"))
PY
done
head -c 700 gpurun_out/r03p_pmc_summary.log
