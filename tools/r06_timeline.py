"""Device timeline of ONE steady-state pairwise step of a small workload (default mnist_mlp): every kernel / copy with its start
offset and duration, and the host-side wall clock of the same step -- where a 22 ms MNIST step spends the time its 10 ms of kernels
do not account for.   gpurun -- 'python tools/r06_timeline.py [workload]'"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.autograd import DeviceType
from torch.profiler import ProfilerActivity, profile

import bench
from kronfluence_amd.utils.state import State


def timeline(step, count, queries):
    for _ in range(3):
        step(count, queries)
    torch.cuda.synchronize()
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        step(count, queries)
        torch.cuda.synchronize()
        walls.append(1e3 * (time.perf_counter() - t0))
    print("plain steps (ms):", [round(w, 2) for w in walls])
    many = os.environ.get("KF_TIMELINE_STEPS")
    if many:   # several consecutive steps, only the long activities and the long gaps: what a periodic slow step consists of
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            marks = []
            for _ in range(int(many)):
                t0 = time.perf_counter()
                step(count, queries)
                torch.cuda.synchronize()
                marks.append(round(1e3 * (time.perf_counter() - t0), 2))
        print("profiled steps (ms):", marks)
        dev = sorted((e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type == DeviceType.CUDA)
        first, last_end = dev[0][0], dev[0][0]
        for s, e, name in dev:
            if s - last_end > 1000 or e - s > 1000:
                print(f"{(s - first) / 1e3:9.3f} ms  +{(s - last_end) / 1e3:8.3f} ms gap  {(e - s) / 1e3:9.3f} ms  {name[:90]}")
            last_end = max(last_end, e)
        cpu = sorted(((e.time_range.end - e.time_range.start), e.time_range.start - first, e.name) for e in prof.events() if e.device_type == DeviceType.CPU)[-25:]
        print("longest host-side operators (ms, at ms):", [(round(d / 1e3, 1), round(at / 1e3, 1), n[:40]) for d, at, n in reversed(cpu)])
        return None
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        t0 = time.perf_counter()
        step(count, queries)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
    dev = sorted((e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type == DeviceType.CUDA)
    first = dev[0][0]
    print(f"profiled step: {wall:.2f} ms wall, {len(dev)} device activities, busy {sum(e - s for s, e, _ in dev) / 1e3:.2f} ms")
    last_end = first
    for s, e, name in dev:
        gap = s - last_end
        print(f"{(s - first) / 1e3:9.3f} ms  +{gap:8.1f} us gap  {e - s:9.1f} us  {name[:90]}")
        last_end = max(last_end, e)
    cpu = sorted(((e.time_range.end - e.time_range.start), e.name) for e in prof.events() if e.device_type == DeviceType.CPU)[-12:]
    print("longest host-side operators (us):", [(round(d), n[:40]) for d, n in reversed(cpu)])
    return None


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "mnist_mlp"
    bench._device_busy = timeline
    os.environ["KF_BENCH_BUSY"] = "1"
    bench.run_workload(workload, State(), None, None, steps=3, warmup=2, factor_reps=0, cpu_baseline=False)


if __name__ == "__main__":
    main()
