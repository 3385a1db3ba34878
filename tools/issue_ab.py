"""A/B of the request schedules of the 256 x 256 wave-role-split loop (csrc/kf_pingpong.h, template parameter ISSUE, run-time
override KF_PP_ISSUE): 0 = LDS-DMA requests in the L segments (round 3), 1 = between the MFMA groups of the M segments,
2 = as 1 with A0 left in the short L segment.  One process, the public entry points, HIP events on the launch stream; run under
``rocprofv3 --kernel-trace --stats`` the three instantiations of every kernel show up as separate rows.

    gpurun -- 'python tools/issue_ab.py [out.json]'

Prints one line per case (ms per call and arm, results compared with arm 0) and writes {"totals_ms": {...}, "best": k}.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"
ARMS = (0, 1, 2)


def timed(fn, reps=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def arm(k):
    os.environ["KF_PP_ISSUE"] = str(k)


def run_case(name, flops, call, result, totals, exact=False):
    """call(): the timed entry point; result(): a fresh result tensor of one call (compared across arms)."""
    line, outs = f"  {name:34s}", {}
    for k in ARMS:
        arm(k)
        t = timed(call)
        outs[k] = result()
        totals[k] += t
        line += f" issue{k} {t:7.3f} ms {flops / t / 1e9:5.0f} TF/s |"
    worst = 0.0
    for k in ARMS[1:]:
        if exact:
            worst = max(worst, float((outs[k].float() - outs[0].float()).abs().max()))
        else:
            worst = max(worst, float((outs[k] - outs[0]).abs().max() / outs[0].abs().max()))
    ok = worst == 0.0 if exact else worst <= 1e-5
    print(f"{line} max diff vs issue0 {worst:.1e}{'' if ok else '   <-- MISMATCH'}", flush=True)
    return ok


def main():
    torch.manual_seed(0)
    totals = {k: 0.0 for k in ARMS}
    ok = True
    print("== conv score entry points (pad + per-sample gradients + 256 x 256 score GEMM), Q = b = 1000")
    for name, cin, cout, ksz, st, pd, h in [("conv 128->128 16x16", 128, 128, 3, 1, 1, 16), ("conv 128->256 16x16", 128, 256, 3, 1, 1, 16),
                                            ("conv 256->256 8x8", 256, 256, 3, 1, 1, 8), ("conv 64->128 k5 s2", 64, 128, 5, 2, 2, 32)]:
        q = b = 1000
        conv = nn.Conv2d(cin, cout, ksz, stride=st, padding=pd, bias=False)
        x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
        o = (h + 2 * pd - ksz) // st + 1
        g = torch.randn(b, cout, o, o, device=DEV).bfloat16()
        ip = cin * ksz * ksz
        tiled = TiledQueries(torch.randn(q, cout, ip, device=DEV).bfloat16(), 0, conv_channels=cin)
        s = torch.zeros(q, b, device=DEV)

        def fresh():
            s.zero_()
            ops.pairwise_score_conv2d(s, 0, tiled, g, x, conv)
            return s.clone()

        ok &= run_case(name, 2.0 * q * b * cout * ip + 2.0 * b * o * o * cout * ip,
                       lambda: ops.pairwise_score_conv2d(s, 0, tiled, g, x, conv), fresh, totals)
        del x, g, tiled
    print("== rotations X @ Q (bf16 -> bf16)")
    for name, n, d in [("256000 x 1600^2", 256000, 1600), ("64000 x 2304^2", 64000, 2304), ("256000 x 1152^2", 256000, 1152)]:
        x = torch.randn(n, d, device=DEV).bfloat16()
        q_t = torch.linalg.qr(torch.randn(d, d, device=DEV))[0].t().contiguous().bfloat16()
        ok &= run_case(name, 2.0 * n * d * d, lambda: ops.rotate_bf16(x, q_t), lambda: ops.rotate_bf16(x, q_t), totals, exact=True)
        del x, q_t
    print("== covariance of conv inputs (implicit im2col, 256-row kernel)")
    for name, cin, ksz, st, pd, h, b in [("5x5 s2 64ch 32x32 (1600 rows)", 64, 5, 2, 2, 32, 1000), ("3x3 128ch 16x16 (1152 rows)", 128, 3, 1, 1, 16, 1000),
                                          ("3x3 256ch 8x8 (2304 rows)", 256, 3, 1, 1, 8, 1000)]:
        conv = nn.Conv2d(cin, 32, ksz, stride=st, padding=pd, bias=False)
        x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
        geometry = ops.conv2d_cov_geometry(x, conv)
        d = cin * ksz * ksz
        o = (h + 2 * pd - ksz) // st + 1
        cov, cnt = torch.zeros(d, d, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)

        def fresh():
            cov.zero_()
            ops.conv2d_cov_accum(cov, cnt, x, conv, geometry)
            return cov.clone()

        ok &= run_case(name, float(b * o * o) * d * (d + 1), lambda: ops.conv2d_cov_accum(cov, cnt, x, conv, geometry), fresh, totals)
        del x
    print("== transformer score entry points (transposes + per-sample gradients + score GEMM)")
    for name, q, b, t_len, o, i in [("gpt2 768x769 T=512 b=256 (pair)", 1024, 256, 512, 768, 768), ("gpt2 768x3073 T=512 b=256 (pair)", 1024, 256, 512, 768, 3072),
                                    ("bert 768x769 T=128 b=512", 872, 512, 128, 768, 768), ("bert 3072x769 T=128 b=512", 872, 512, 128, 3072, 768)]:
        ipp = (i + 1 + 7) // 8 * 8
        g = torch.randn(b, t_len, o, device=DEV).bfloat16()
        a = torch.randn(b, t_len, i, device=DEV).bfloat16()
        tiled = TiledQueries(torch.randn(q, o, ipp, device=DEV).bfloat16(), 0)
        s = torch.zeros(q, b, device=DEV)

        def fresh():
            s.zero_()
            ops.pairwise_score_rows(s, 0, tiled, g, a, True)
            return s.clone()

        ok &= run_case(name, 2.0 * q * b * o * (i + 1) + 2.0 * b * t_len * o * (i + 1),
                       lambda: ops.pairwise_score_rows(s, 0, tiled, g, a, True), fresh, totals, )
        del g, a, tiled
    os.environ.pop("KF_PP_ISSUE", None)
    best = min(totals, key=totals.get)
    summary = {"totals_ms": {str(k): round(v, 3) for k, v in totals.items()}, "best": best, "all_equal": bool(ok)}
    print(json.dumps(summary))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(summary, fh)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
