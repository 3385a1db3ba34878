"""A/B of the K-major per-sample-gradient kernel (round 6): one workgroup per (sample, tile) item with the tile transposed through
LDS (psg_gemm_tn_kernel, default) against 256 persistent workgroups streaming their items with register stores
(psg_gemm_tn_stream_kernel, KF_PSG_STREAM=1; read per call).
    gpurun -- 'python tools/r06_psg_stream.py [reps]'
Per shape: milliseconds of the score entry point (kf_pairwise_score_rows2: gradients + score GEMM) either way, the K-contiguous
path (KF_TN=0: transposed copies) as the independent reference, and the relative difference of the score blocks.  Under
rocprofv3 --kernel-trace --stats the two gradient kernels show up under their own names."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"
CASES = [  # name, Q, b0, b1, T, O, I, bias
    ("bert 768x769 T=128 Q=872 b=512", 872, 512, 0, 128, 768, 768, True),
    ("bert 3072x769 T=128 Q=872 b=512", 872, 512, 0, 128, 3072, 768, True),
    ("bert 768x3073 T=128 Q=872 b=512", 872, 512, 0, 128, 768, 3072, True),
    ("gpt2 768x769 T=512 Q=1024 b=2x128", 1024, 128, 128, 512, 768, 768, True),
    ("gpt2 2304x769 T=512 Q=1024 b=2x128", 1024, 128, 128, 512, 2304, 768, True),
    ("gpt2 768x3073 T=512 Q=1024 b=2x128", 1024, 128, 128, 512, 768, 3072, True),
    ("llama 4096x4096 T=512 Q=64 b=2x8 no bias", 64, 8, 8, 512, 4096, 4096, False),
    ("odd: 256x513 T=192 Q=40 b=37+5", 40, 37, 5, 192, 256, 512, True),
    ("odd: 512x256 T=128 Q=9 b=3 no bias", 9, 3, 0, 128, 512, 256, False),
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    bad = 0
    for name, q, b0, b1, t_len, o, i, bias in CASES:
        torch.manual_seed(3)
        ipp = (i + int(bias) + 7) // 8 * 8
        b = b0 + b1
        g = torch.randn(b, t_len, o, device=DEV).bfloat16()
        a = torch.randn(b, t_len, i, device=DEV).bfloat16()
        tiled = TiledQueries(torch.randn(q, o, ipp, device=DEV).bfloat16(), 0)
        second = (g[b0:], a[b0:]) if b1 else None
        flops = 2.0 * q * b * o * (i + int(bias)) + 2.0 * b * t_len * o * (i + int(bias))
        line, outs = f"  {name:44s}", {}
        for label, tn, stream in (("K-contig", "0", "0"), ("TN item/wg", "1", "0"), ("TN stream", "1", "1")):
            os.environ["KF_TN"], os.environ["KF_PSG_STREAM"] = tn, stream
            s = torch.zeros(q, b, device=DEV)
            t = timed(lambda: ops.pairwise_score_rows(s, 0, tiled, g[:b0], a[:b0], bias, second=second), reps)
            s.zero_()
            ops.pairwise_score_rows(s, 0, tiled, g[:b0], a[:b0], bias, second=second)
            torch.cuda.synchronize()
            outs[label] = s.clone()
            line += f" {label} {t:7.3f} ms {flops / t / 1e9:5.0f} TF/s |"
        d = max(float((outs[k] - outs["K-contig"]).norm() / outs["K-contig"].norm()) for k in outs)
        bad += d >= 1e-4
        print(f"{line} max rel diff {d:.1e}{'' if d < 1e-4 else '   <-- MISMATCH'}", flush=True)
        del g, a, tiled
        torch.cuda.empty_cache()
    os.environ.pop("KF_TN", None)
    os.environ.pop("KF_PSG_STREAM", None)
    print("all shapes agree" if not bad else f"{bad} shapes DISAGREE")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
