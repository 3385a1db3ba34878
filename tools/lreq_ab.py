"""A/B of where the LDS-DMA requests of the 64 x 64-wave loop are issued (KF_PP64_LREQ: how many of a k-tile's 6 requests go out in
the L segment, the rest between the MFMA groups) on the Lambda product:  gpurun -- 'python tools/lreq_ab.py'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops

DEV = "cuda:0"
CASES = [("gpt2 768x769 T=512 b=64", 64, 512, 768, 776, 769), ("gpt2 3072x769 T=512 b=64", 64, 512, 3072, 776, 769),
         ("gpt2 768x3073 T=512 b=64", 64, 512, 768, 3080, 3073), ("bert 768x769 T=128 b=256", 256, 128, 768, 776, 769),
         ("bert 3072x769 T=128 b=256", 256, 128, 3072, 776, 769), ("llama 4096x4096 T=512 b=16", 16, 512, 4096, 4096, 4096),
         ("resnet conv 256x2304 R=64 b=1000", 1000, 64, 256, 2304, 2304)]


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


torch.manual_seed(0)
for name, b, r, o, w, ip in CASES:
    gt = torch.randn(b, o, r, device=DEV).bfloat16()
    at = torch.randn(b, w, r, device=DEV).bfloat16()
    line, first = f"{name:34s}", None
    for lreq in (6, 4, 3, 2, 0):
        os.environ["KF_PP64_LREQ"] = str(lreq)
        lam = torch.zeros(o, ip, device=DEV)
        t = timed(lambda: ops.lambda_rows_accum(lam, gt, at))
        lam.zero_()
        ops.lambda_rows_accum(lam, gt, at)
        if first is None:
            first = lam.clone()
        d = float((lam - first).norm() / first.norm())
        line += f" L{lreq} {t * 1e3:6.0f} us {2.0 * b * r * o * ip / t / 1e9:5.0f} TF{'' if d < 1e-5 else ' MISMATCH ' + str(d)} |"
    print(line, flush=True)
os.environ.pop("KF_PP64_LREQ", None)
