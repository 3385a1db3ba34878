"""Per-layer timing of the pairwise-score entry points at the real layer shapes (HIP events on the launch stream).

    gpurun -- 'python tools/kernel_bench.py [resnet9] [bert] [gpt2] [--q 1000] [--b 1000]'

For every layer: the v1 path (materialised patches / transposed gradient + kf_pairwise_score on the k-tile-major P) and the
v2 path (kf_pairwise_score_conv2d: implicit im2col, or kf_pairwise_score_rows) -- milliseconds per call, TFLOP/s on the
algorithmic flops 2 Q b O I' + 2 b R O I', and the agreement of the two score blocks."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"
RESNET9 = [  # (cin, cout, k, stride, padding, H)
    ("conv0 3->64 (C pad 8)", 3, 64, 3, 1, 1, 32), ("conv7 256->128 6x6 grid", 256, 128, 3, 1, 0, 8),
    ("conv1 64->128 k5 s2", 64, 128, 5, 2, 2, 32), ("conv2 128->128", 128, 128, 3, 1, 1, 16), ("conv4 128->256", 128, 256, 3, 1, 1, 16),
    ("conv5 256->256 8x8", 256, 256, 3, 1, 1, 8), ("conv 1x1 1152->128 (aligned)", 1152, 128, 1, 1, 0, 16),
]
SEQ = {  # (name, O, I, T)
    "bert": [("bert 768x769", 768, 768, 128), ("bert 3072x769", 3072, 768, 128), ("bert 768x3073", 768, 3072, 128)],
    "gpt2": [("gpt2 2304x769", 2304, 768, 512), ("gpt2 768x3073", 768, 3072, 512)],
}


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
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["resnet9", "bert"])
    ap.add_argument("--q", type=int, default=1000)
    ap.add_argument("--b", type=int, default=1000)
    args = ap.parse_args()
    q = args.q
    if "resnet9" in args.which:
        b = args.b
        for name, cin, cout, k, s, p, h in RESNET9:
            conv = nn.Conv2d(cin, cout, k, stride=s, padding=p, bias=False)
            x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
            o = (h + 2 * p - k) // s + 1
            g = torch.randn(b, cout, o, o, device=DEV).bfloat16()
            ip = cin * k * k
            pq = torch.randn(q, cout, ip, device=DEV).bfloat16()
            flops = 2.0 * q * b * cout * ip + 2.0 * b * o * o * cout * ip
            v1p = TiledQueries(pq, (-ip) % 8) if cout % 8 == 0 else pq
            v2p = TiledQueries(pq, 0, conv_channels=cin)
            s1, s2 = torch.zeros(q, b, device=DEV), torch.zeros(q, b, device=DEV)

            def v1(out=s1):
                patches = ops.im2col(x, conv, False, torch.bfloat16)
                if ip % 8:
                    patches = torch.nn.functional.pad(patches, (0, (-ip) % 8))
                rows = g.flatten(2).transpose(1, 2).contiguous()
                ops.pairwise_score(out, 0, v1p, rows, patches, False)

            def v2(out=s2):
                ops.pairwise_score_conv2d(out, 0, v2p, g, x, conv)

            t1, t2 = timed(v1), timed(v2)
            s1.zero_(); s2.zero_(); v1(); v2()
            err = float((s1 - s2).norm() / s1.norm())
            print(f"{name:26s} v1 {t1:7.3f} ms {flops / t1 / 1e9:6.0f} TF/s | v2 {t2:7.3f} ms {flops / t2 / 1e9:6.0f} TF/s | rel diff {err:.1e}", flush=True)
    for family in ("bert", "gpt2"):
        if family not in args.which:
            continue
        for name, o, i, t in SEQ[family]:
            b = max(8, args.b * 128 // t // 4)  # 250 sequences of 128 tokens by default
            g = torch.randn(b, t, o, device=DEV).bfloat16()
            a = torch.randn(b, t, i, device=DEV).bfloat16()
            ip = i + 1
            pad = (-ip) % 8
            pq = torch.randn(q, o, ip, device=DEV).bfloat16()
            tiled = TiledQueries(pq, pad)
            flops = 2.0 * q * b * o * ip + 2.0 * b * t * o * ip
            s1, s2 = torch.zeros(q, b, device=DEV), torch.zeros(q, b, device=DEV)

            def v1(out=s1):
                ap = torch.cat([a, a.new_ones(b, t, 1), a.new_zeros(b, t, pad)], dim=-1)
                ops.pairwise_score(out, 0, tiled, g, ap, False)

            def v2(out=s2):
                ops.pairwise_score_rows(out, 0, tiled, g, a, True)

            t1, t2 = timed(v1, 3), timed(v2, 3)
            s1.zero_(); s2.zero_(); v1(); v2()
            err = float((s1 - s2).norm() / s1.norm())
            print(f"{name:26s} b={b:4d} v1 {t1:7.3f} ms {flops / t1 / 1e9:6.0f} TF/s | v2 {t2:7.3f} ms {flops / t2 / 1e9:6.0f} TF/s | rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
