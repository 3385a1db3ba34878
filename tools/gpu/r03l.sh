#!/usr/bin/env bash
# round 3, call L: wave-specialised per-sample-gradient kernel (loaders / storers, raw barriers) and the query-side
# preconditioner on the round-3 engines: parity tests, A/B, bench lines.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x ) > gpurun_out/r03l_ops.log 2>&1
( timeout 400 python -m pytest tests/test_layer_shapes_gpu.py tests/test_pipeline_gpu.py -q ) > gpurun_out/r03l_shapes.log 2>&1
( timeout 400 python tools/engine_ab.py ) > gpurun_out/r03l_engine_ab.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03l_bench.log 2>&1
( timeout 400 python bench.py --workload bert_base --n-train 2048 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03l_bert.log 2>&1
( KF_PRECOND_V3_OFF=1 timeout 400 python bench.py --workload bert_base --n-train 2048 --steps 1 --warmup 1 --no-cpu-baseline --factor-reps 0 ) > gpurun_out/r03l_bert_old_precond.log 2>&1
tail -n 4 gpurun_out/r03l_ops.log gpurun_out/r03l_shapes.log
grep -n "MISMATCH" gpurun_out/r03l_engine_ab.log | head
grep -A12 "implicit-im2col score entry" gpurun_out/r03l_engine_ab.log
for f in gpurun_out/r03l_bench.log gpurun_out/r03l_bert.log gpurun_out/r03l_bert_old_precond.log; do python - "$f" <<'PY'
import sys, json
s = open(sys.argv[1]).read(); i = s.rfind('{"metric')
if i < 0: print(sys.argv[1], "NO JSON", s[-300:])
else:
    d = json.loads(s[i:].strip().splitlines()[0]); print(sys.argv[1], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "avg_ms", d["roofline"]["avg_launch_ms"])
PY
done
