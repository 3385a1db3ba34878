"""Per-layer times of the event-timed entry points of one workload (default resnet9): bench.py's own HIP events, grouped by the
position of a call within its batch (the hooks fire in a fixed layer order), so that a stage's time can be read off layer by layer.
    gpurun -- 'python tools/r06_layer_times.py [workload] [n_train]'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KF_BENCH_BUSY", "0")
import bench
from kronfluence_amd.utils.state import State

real = bench._event_summary
SEEN = {}


def spy(events, peak, kernel, elapsed=None, other_kernel_ms=0.0):
    tag = kernel.split(":")[0][:40]
    if events and tag not in SEEN:
        SEEN[tag] = [(s.elapsed_time(e), f, b) for s, e, f, b in events]
    return real(events, peak, kernel, elapsed, other_kernel_ms)


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "resnet9"
    n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    bench._event_summary = spy
    state = State()
    import torch
    from kronfluence_amd import prepare_model
    torch.manual_seed(0)
    raw = bench.WORKLOADS[workload]["model"]()
    task, *_ = bench.workload_parts(bench.WORKLOADS[workload], raw)
    layers = bench.tracked_shapes(prepare_model(raw, task))
    period = len(layers)
    r = bench.run_workload(workload, state, n_train, None, steps=1, warmup=1, factor_reps=1, cpu_baseline=False)
    print(f"{workload}: {r['value']:.4g} pairs/s, fit {r['factor_fit']['seconds']}; tracked layers (O, I') in forward order: {layers}")
    for tag, calls in SEEN.items():
        if len(calls) % period:   # (activation + gradient calls interleave: grouped by their algorithmic flop count instead)
            print(f"== {tag}: {len(calls)} calls, mean {sum(c[0] for c in calls) / len(calls):.3f} ms; by algorithmic flops / bytes:")
            groups = {}
            for ms, fl, nb in calls:
                groups.setdefault((fl, nb), []).append(ms)
            for (fl, nb), mss in sorted(groups.items()):
                ms = sum(mss) / len(mss)
                print(f"   {len(mss):4d} calls {ms:8.3f} ms  {fl / ms / 1e9 if ms else 0:8.1f} TFLOP/s on {fl:.3g} flop, {nb / 1e6:8.1f} MB in")
            continue
        print(f"== {tag}: {len(calls)} calls = {len(calls) // period} batches x {period} layers (order of the hooks)")
        for pos in range(period):
            mine = calls[pos::period]
            ms = sum(c[0] for c in mine) / len(mine)
            fl = mine[0][1]
            print(f"   call {pos}: {ms:8.3f} ms  {fl / ms / 1e9 if ms else 0:8.1f} TFLOP/s on {fl:.3g} algorithmic flop")


if __name__ == "__main__":
    main()
