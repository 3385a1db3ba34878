"""A/B of the work-item count of the 128-row covariance kernel (KF_COV_ITEMS) and of the two covariance engines at the factor
batches bench.py uses for the transformer configs:  gpurun -- 'python tools/cov_items_ab.py'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops

DEV = "cuda:0"
SEQS = [("bert 768 T128 b256", 256, 128, 768, True), ("bert 768 T128 b64", 64, 128, 768, True), ("bert 768 T128 b512", 512, 128, 768, True),
        ("gpt2 768 T512 b64", 64, 512, 768, True), ("gpt2 768 T512 b16", 16, 512, 768, True), ("gpt2 768 T512 b128", 128, 512, 768, True),
        ("1536 T128 b256", 256, 128, 1536, True), ("1536 T512 b64", 64, 512, 1536, False), ("llama 1024 T512 b8 (grad)", 8, 512, 1024, False)]

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


for name, bb, t_len, d_in, bias in SEQS:
    x = torch.randn(bb, t_len, d_in, device=DEV).bfloat16()
    d = d_in + int(bias)
    count = torch.zeros(1, dtype=torch.int64, device=DEV)
    cov = torch.zeros(d, d, device=DEV)
    flops = float(bb * t_len) * d * (d + 1)
    line = f"{name:30s}"
    for label, env in [("default", {})] + [(f"k>={m}", {"KF_COV_ENGINE": "2", "KF_COV_MIN_KSTEPS": str(m)}) for m in (8, 16, 32, 48, 64)] + [("v3", {"KF_COV_ENGINE": "3"})]:
        for k in ("KF_COV_ENGINE", "KF_COV_ITEMS", "KF_COV_MIN_KSTEPS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        t = timed(lambda: ops.linear_activation_cov(cov, count, x, None, bias))
        line += f" {label} {t * 1e3:6.0f} us ({flops / t / 1e9:4.0f}) |"
    print(line, flush=True)
