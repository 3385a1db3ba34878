"""Where the HOST time of a small pairwise step goes (MNIST-MLP: 88 % of a step the GPU waits for Python): cProfile over
``bench.run_workload`` with many timed steps, functions by own time and by cumulative time.

    gpurun -- 'python tools/r06_host_profile.py [workload] [steps]'
"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KF_BENCH_BUSY", "0")
import bench
from kronfluence_amd.utils.state import State


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "mnist_mlp"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    state = State()
    bench.run_workload(workload, state, None, None, steps=2, warmup=1, factor_reps=0, cpu_baseline=False)   # warm everything
    prof = cProfile.Profile()
    prof.enable()
    result = bench.run_workload(workload, state, None, None, steps=steps, warmup=1, factor_reps=0, cpu_baseline=False)
    prof.disable()
    print(f"{workload}: {result['ms_per_step']:.2f} ms per step over {steps} steps (profiled: slower than plain)")
    for key in ("tottime", "cumulative"):
        print(f"==== by {key}")
        pstats.Stats(prof).sort_stats(key).print_stats(45)


if __name__ == "__main__":
    main()
