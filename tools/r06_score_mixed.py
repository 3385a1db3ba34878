"""A/B of the mixed row tiling of the score GEMM (round 6; KF_SCORE_MIXED, read per call) on BERT's shapes: 872 queries against 512
sequences -- one launch over 1 024 padded query rows (default) against 768 rows on the 256 x 256 loop + the last 104 rows on ONE
128 x 512 tile split over k (score_gemm_v5_kernel<1, 8>).
    gpurun -- 'python tools/r06_score_mixed.py [reps]'
Per shape: milliseconds of the score entry point (K-major gradients + score GEMM) either way and the relative difference of the
score blocks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"
CASES = [  # name, Q, b, T, O, I, bias
    ("bert 768x769 T=128 Q=872 b=512", 872, 512, 128, 768, 768, True),
    ("bert 3072x769 T=128 Q=872 b=512", 872, 512, 128, 3072, 768, True),
    ("bert 768x3073 T=128 Q=872 b=512", 872, 512, 128, 768, 3072, True),
    ("Q=600 b=1024 768x769 T=128", 600, 1024, 128, 768, 768, True),
    ("Q=300 b=512 768x769 T=128", 300, 512, 128, 768, 768, True),
    ("Q=872 b=384 768x769 T=128 (128 x 256 remainder)", 872, 384, 128, 768, 768, True),
]


def timed(fn, reps):
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    bad = 0
    for name, q, b, t_len, o, i, bias in CASES:
        torch.manual_seed(5)
        ipp = (i + int(bias) + 7) // 8 * 8
        g = torch.randn(b, t_len, o, device=DEV).bfloat16()
        a = torch.randn(b, t_len, i, device=DEV).bfloat16()
        tiled = TiledQueries(torch.randn(q, o, ipp, device=DEV).bfloat16(), 0)
        flops = 2.0 * q * b * o * (i + int(bias)) + 2.0 * b * t_len * o * (i + int(bias))
        line, outs = f"  {name:50s}", {}
        for label, mixed in (("one launch", "0"), ("mixed rows", "1")):
            os.environ["KF_SCORE_MIXED"] = mixed
            s = torch.zeros(q, b, device=DEV)
            t = timed(lambda: ops.pairwise_score_rows(s, 0, tiled, g, a, bias), reps)
            s.zero_()
            ops.pairwise_score_rows(s, 0, tiled, g, a, bias)
            torch.cuda.synchronize()
            outs[label] = s.clone()
            line += f" {label} {t:7.3f} ms {flops / t / 1e9:5.0f} TF/s |"
        d = float((outs["mixed rows"] - outs["one launch"]).norm() / outs["one launch"].norm())
        bad += d >= 1e-5
        print(f"{line} rel diff {d:.1e}{'' if d < 1e-5 else '   <-- MISMATCH'}", flush=True)
        del g, a, tiled
        torch.cuda.empty_cache()
    os.environ.pop("KF_SCORE_MIXED", None)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
