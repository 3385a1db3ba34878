#!/usr/bin/env bash
# round 3, call M: per-sample-gradient kernel with the DMA wait deferred to the top of the consuming k-step (the next item's
# first k-step in flight through the epilogue, counted vmcnt over the copy-out stores) -- parity tests and A/B against the
# wait-at-the-bottom form (KF_PSG_DEFER=0).
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x ) > gpurun_out/r03m_ops.log 2>&1
( timeout 400 python -m pytest tests/test_layer_shapes_gpu.py tests/test_fullsize_gpu.py -q ) > gpurun_out/r03m_shapes.log 2>&1
( timeout 300 python tools/engine_ab.py ) > gpurun_out/r03m_engine_ab_defer.log 2>&1
( KF_PSG_DEFER=0 timeout 300 python tools/engine_ab.py ) > gpurun_out/r03m_engine_ab_bottom.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03m_bench_defer.log 2>&1
( KF_PSG_DEFER=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/r03m_bench_bottom.log 2>&1
tail -n 4 gpurun_out/r03m_ops.log gpurun_out/r03m_shapes.log
grep -n "MISMATCH" gpurun_out/r03m_engine_ab_defer.log | head
for f in defer bottom; do echo "== $f"; grep -A10 "implicit-im2col score entry" gpurun_out/r03m_engine_ab_$f.log | cut -c1-140; grep -A4 "Lambda of a conv" gpurun_out/r03m_engine_ab_$f.log | cut -c1-140; done
for f in gpurun_out/r03m_bench_defer.log gpurun_out/r03m_bench_bottom.log; do python - "$f" <<'PY'
import sys, json
s = open(sys.argv[1]).read(); i = s.rfind('{"metric')
if i < 0: print(sys.argv[1], "NO JSON", s[-300:])
else:
    d = json.loads(s[i:].strip().splitlines()[0]); print(sys.argv[1], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "avg_ms", d["roofline"]["avg_launch_ms"], "fit", d["factor_fit"]["samples_per_sec"])
PY
done
