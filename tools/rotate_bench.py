"""Timing of the eigenbasis rotations ``X @ Q`` (ops.rotate_bf16: bf16 in, bf16 out) at the ResNet-9 / BERT Lambda-stage shapes.

    gpurun -- 'python tools/rotate_bench.py'
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops

DEV = "cuda:0"
SHAPES = [("resnet conv2 patches 256000 x 1152", 256000, 1152), ("resnet conv1 patches 256000 x 1600", 256000, 1600),
          ("resnet conv5 patches 64000 x 2304", 64000, 2304), ("resnet grads 256000 x 128", 256000, 128),
          ("resnet grads 64000 x 256", 64000, 256), ("bert rows 8192 x 776 (769 padded)", 8192, 776),
          ("bert rows 8192 x 3072", 8192, 3072)]


def timed(fn, reps=5):
    for _ in range(2):
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
    for name, n, d in SHAPES:
        x = torch.randn(n, d, device=DEV).bfloat16()
        q = torch.linalg.qr(torch.randn(d, d, device=DEV))[0]
        q_t = q.t().contiguous().bfloat16()
        t = timed(lambda: ops.rotate_bf16(x, q_t))
        got = ops.rotate_bf16(x[:512], q_t).float()
        want = x[:512].float() @ q_t.float().t()
        err = float((got - want).norm() / want.norm())
        print(f"{name:40s} {t:7.3f} ms {2.0 * n * d * d / t / 1e9:6.0f} TF/s   rel err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
