"""Timing of the covariance entry points at the real layer shapes (HIP events on the launch stream).

    gpurun -- 'python tools/cov_bench.py'        (both engines: KF_COV_ENGINE=2 the 128-row / 4-wave kernel, 3 the 256-row
                                                  wave-role-split kernel; the two results are compared)

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


def both(fn, make_cov, flops, name):
    """Times ``fn(cov)`` under both engines and compares the accumulated matrices."""
    line, results = f"{name:40s}", {}
    for gen in (2, 3):
        os.environ["KF_COV_ENGINE"] = str(gen)
        cov = make_cov()
        t = timed(lambda: fn(cov))
        cov.zero_()
        fn(cov)
        results[gen] = cov.clone()
        line += f" engine{gen} {t:7.3f} ms {flops / t / 1e9:6.0f} TF/s |"
    diff = float((results[3] - results[2]).norm() / results[2].norm())
    sym = float((results[3] - results[3].t()).abs().max())
    print(f"{line} rel diff {diff:.1e} asym {sym:.1e}{'' if diff < 1e-5 else '   <-- MISMATCH'}", flush=True)
    os.environ.pop("KF_COV_ENGINE")


def main():
    b = 1000
    for name, cin, cout, k, s, p, h in CONVS:
        conv = nn.Conv2d(cin, cout, k, stride=s, padding=p, bias=False)
        x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
        o = (h + 2 * p - k) // s + 1
        d = cin * k * k
        count = torch.zeros(1, dtype=torch.int64, device=DEV)
        geometry = ops.conv2d_cov_geometry(x, conv)
        if geometry is None:
            print(f"{name:40s} not on the implicit path (materialised patches)", flush=True)
            continue
        both(lambda cov: ops.conv2d_cov_accum(cov, count, x, conv, geometry), lambda: torch.zeros(d, d, device=DEV),
             float(b * o * o) * d * (d + 1), name)
    for name, cout, h in [("grad cov 128 ch 16x16", 128, 16), ("grad cov 256 ch 16x16", 256, 16), ("grad cov 256 ch 8x8", 256, 8)]:
        g = torch.randn(b, cout, h, h, device=DEV).bfloat16()
        count = torch.zeros(1, dtype=torch.int64, device=DEV)
        both(lambda cov: ops.conv_gradient_cov(cov, count, g), lambda: torch.zeros(cout, cout, device=DEV),
             float(b * h * h) * cout * (cout + 1), name)
    for name, bb, t_len, d_in in SEQS:
        x = torch.randn(bb, t_len, d_in, device=DEV).bfloat16()
        mask = (torch.rand(bb, t_len, device=DEV) < 0.9).to(torch.int64)
        d = d_in + 1
        count = torch.zeros(1, dtype=torch.int64, device=DEV)
        both(lambda cov: ops.linear_activation_cov(cov, count, x, mask, True), lambda: torch.zeros(d, d, device=DEV),
             float(bb * t_len) * d * (d + 1), name)


if __name__ == "__main__":
    main()
