"""A/B of the per-sample-gradient kernel's epilogue on the ResNet-9 layer shapes of bench.py (round 6): the bf16 tile transposed
through LDS (default) against 8-byte stores straight from the accumulators (KF_PSG_DIRECT=1, read per call).
    gpurun -- 'python tools/r06_psg_direct.py [--q 1000] [--b 1000]'
Per layer: milliseconds of the score entry point (pad + gradients + score GEMM) either way and the relative difference of the score
blocks (same products, same rounding, only the route to memory differs: what is left is the order of the split-K atomics).  Under rocprofv3 --kernel-trace --stats the two
kernels show up as psg_gemm_v3_kernel<0, false> / <0, true>."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"
LAYERS = [  # bench.py's resnet9(): (cin, cout, k, stride, padding, H)
    ("conv0   3->64  3x3 32x32", 3, 64, 3, 1, 1, 32), ("conv1  64->128 5x5 s2 32x32", 64, 128, 5, 2, 2, 32),
    ("conv2 128->128 3x3 16x16", 128, 128, 3, 1, 1, 16), ("conv4 128->256 3x3 16x16", 128, 256, 3, 1, 1, 16),
    ("conv5 256->256 3x3 8x8", 256, 256, 3, 1, 1, 8), ("conv7 256->128 3x3 p0 8x8", 256, 128, 3, 1, 0, 8),
]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--q", type=int, default=1000)
    ap.add_argument("--b", type=int, default=1000)
    args = ap.parse_args()
    q, b = args.q, args.b
    total = [0.0, 0.0]
    for name, cin, cout, k, s, p, h in LAYERS:
        conv = nn.Conv2d(cin, cout, k, stride=s, padding=p, bias=False)
        torch.manual_seed(1)
        x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
        o = (h + 2 * p - k) // s + 1
        g = torch.randn(b, cout, o, o, device=DEV).bfloat16()
        ip = cin * k * k
        pq = TiledQueries(torch.randn(q, cout, ip, device=DEV).bfloat16(), 0, conv_channels=cin)
        out = [torch.zeros(q, b, device=DEV), torch.zeros(q, b, device=DEV)]
        ms = []
        for mode in (0, 1):
            os.environ["KF_PSG_DIRECT"] = str(mode)
            ms.append(timed(lambda: ops.pairwise_score_conv2d(out[mode], 0, pq, g, x, conv)))
            out[mode].zero_()
            ops.pairwise_score_conv2d(out[mode], 0, pq, g, x, conv)
        torch.cuda.synchronize()
        same = float((out[0] - out[1]).norm() / out[0].norm())   # (split-K atomics land in any order: not bit-identical run to run)
        flops = 2.0 * q * b * cout * ip + 2.0 * b * o * o * cout * ip
        total[0] += ms[0]; total[1] += ms[1]
        print(f"{name:28s} lds {ms[0]:7.3f} ms {flops / ms[0] / 1e9:6.0f} TF/s | direct {ms[1]:7.3f} ms {flops / ms[1] / 1e9:6.0f} TF/s | "
              f"rel diff {same:.1e}", flush=True)
    os.environ.pop("KF_PSG_DIRECT", None)
    print(f"sum over the six shapes: lds {total[0]:.3f} ms, direct {total[1]:.3f} ms")


if __name__ == "__main__":
    main()
