"""Timing of the covariance entry points at the real layer shapes (HIP events on the launch stream).

    gpurun -- 'python tools/cov_bench.py'        (KF_COV_TILE=128 forces the 128 x 128-tile kernel)

ResNet-9 convolutions through kf_conv2d_cov_accum (implicit im2col), BERT / GPT-2 activations through kf_syrk_rows_bf16:
milliseconds per call and TFLOP/s on the algorithmic flops n d (d + 1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

from kronfluence_amd import ops

DEV = "cuda:0"
CONVS = [("conv1 64->128 k5 s2", 64, 128, 5, 2, 2, 32), ("conv2 128->128 16x16", 128, 128, 3, 1, 1, 16),
         ("conv5 256->256 8x8", 256, 256, 3, 1, 1, 8), ("conv 1x1 1152->128 16x16 (aligned taps)", 1152, 128, 1, 1, 0, 16), ("conv7 256->128 6x6 grid", 256, 128, 3, 1, 0, 8)]
SEQS = [("bert 768 T128 b64", 64, 128, 768), ("bert 3072 T128 b64", 64, 128, 3072), ("gpt2 768 T512 b16", 16, 512, 768),
        ("gpt2 3072 T512 b16", 16, 512, 3072)]


def timed(fn, reps=10):
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
    b = 1000
    for name, cin, cout, k, s, p, h in CONVS:
        conv = nn.Conv2d(cin, cout, k, stride=s, padding=p, bias=False)
        x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
        o = (h + 2 * p - k) // s + 1
        d = cin * k * k
        cov, count = torch.zeros(d, d, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        geometry = ops.conv2d_cov_geometry(x, conv)
        if geometry is None:
            print(f"{name:40s} not on the implicit path (materialised patches)", flush=True)
            continue
        t = timed(lambda: ops.conv2d_cov_accum(cov, count, x, conv, geometry))
        flops = float(b * o * o) * d * (d + 1)
        print(f"{name:40s} {t:7.3f} ms {flops / t / 1e9:6.0f} TF/s", flush=True)
    for name, bb, t_len, d_in in SEQS:
        x = torch.randn(bb, t_len, d_in, device=DEV).bfloat16()
        mask = (torch.rand(bb, t_len, device=DEV) < 0.9).to(torch.int64)
        d = d_in + 1
        cov, count = torch.zeros(d, d, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        t = timed(lambda: ops.linear_activation_cov(cov, count, x, mask, True))
        flops = float(bb * t_len) * d * (d + 1)
        print(f"{name:40s} {t:7.3f} ms {flops / t / 1e9:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
