"""A/B of the two 256 x 256 LDS-DMA main loops (KF_ENGINE=2: round-2 lock-step loop, KF_ENGINE=3: wave-role-split loop of
csrc/kf_pingpong.h) through the public entry points, plus a correctness sweep of the new loop over k-tile counts and ragged
edges against a plain torch reference.

    gpurun -- 'python tools/engine_ab.py'            (add `rocprofv3 --kernel-trace --stats` in front for per-kernel times)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"


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


def engine(gen):
    os.environ["KF_ENGINE"] = str(gen)


def score_case(q, b, o, ip, r=64):
    """kf_pairwise_score on a k-tile-major P: v1 gradient kernel (common to both arms) + the 256 x 256 score GEMM."""
    g = torch.randn(b, r, o, device=DEV).bfloat16()
    a = torch.randn(b, r, ip, device=DEV).bfloat16()
    p = torch.randn(q, o, ip, device=DEV).bfloat16()
    tiled = TiledQueries(p, 0)
    out = {}
    for gen in (2, 3):
        engine(gen)
        s = torch.zeros(q, b, device=DEV)
        ops.pairwise_score(s, 0, tiled, g, a, False)
        out[gen] = s
    psg = torch.einsum("bro,bri->boi", g.float(), a.float()).bfloat16().float()
    want = torch.einsum("qoi,boi->qb", p.float(), psg)
    return [float((out[gen] - want).norm() / want.norm()) for gen in (2, 3)] + [float((out[3] - out[2]).abs().max() / want.abs().max())]


def main():
    torch.manual_seed(0)
    print("== correctness: score GEMM (rel_F vs torch fp32 on the same bf16 gradients) engine 2 / engine 3 / max|3-2|")
    # D = o * ip; KT = D / 64 -> 1, 2, 3, 5 k-tiles without split-K, then split-K shapes; ragged Q / b
    for q, b, o, ip in [(256, 256, 8, 8), (256, 256, 8, 16), (256, 256, 8, 24), (250, 300, 8, 40), (1000, 1000, 64, 64),
                        (513, 777, 128, 1152), (1000, 1000, 128, 1152), (1024, 2048, 256, 2304)]:
        for rep in range(2 if o * ip > 4096 else 1):
            e2, e3, d = score_case(q, b, o, ip)
            flag = "" if (e3 < 1e-4 and d < 1e-4) else "   <-- MISMATCH"
            print(f"  Q={q:5d} b={b:5d} D={o * ip:7d} (KT={o * ip // 64:5d}): {e2:.1e} / {e3:.1e} / {d:.1e}{flag}", flush=True)

    print("== correctness + timing: rotation GEMM X @ Q (bf16 -> bf16)")
    for name, n, d in [("256000 x 1152", 256000, 1152), ("256000 x 1600", 256000, 1600), ("131072 x 768", 131072, 768),
                       ("64000 x 2304", 64000, 2304), ("140000 x 128 (K=128)", 140000, 128 * 9)]:
        x = torch.randn(n, d, device=DEV).bfloat16()
        qm = torch.linalg.qr(torch.randn(d, d, device=DEV))[0]
        q_t = qm.t().contiguous().bfloat16()
        bias = torch.randn(d, device=DEV)
        res = {}
        for gen in (2, 3):
            engine(gen)
            t = timed(lambda: ops.rotate_bf16(x, q_t))
            got = ops.rotate_bf16(x, q_t, bias)
            rows = torch.cat([got[:300], got[n // 2:n // 2 + 300], got[-300:]]).float()
            xs = torch.cat([x[:300], x[n // 2:n // 2 + 300], x[-300:]]).float()
            want = xs @ q_t.float().t() + bias
            res[gen] = (t, float((rows - want).norm() / want.norm()))
        print(f"  {name:24s} engine2 {res[2][0]:7.3f} ms {2.0 * n * d * d / res[2][0] / 1e9:6.0f} TF/s err {res[2][1]:.1e} | "
              f"engine3 {res[3][0]:7.3f} ms {2.0 * n * d * d / res[3][0] / 1e9:6.0f} TF/s err {res[3][1]:.1e}", flush=True)

    print("== timing: score entry point (gradient kernel + score GEMM), Q = b = 1000")
    for name, o, ip, r in [("resnet 128x1152 R=256", 128, 1152, 256), ("resnet 256x2304 R=64", 256, 2304, 64),
                           ("resnet 128x1600 R=256", 128, 1600, 256), ("bert 768x776 R=128 b=250", 768, 776, 128)]:
        q = 1000
        b = 250 if "bert" in name else 1000
        g = torch.randn(b, r, o, device=DEV).bfloat16()
        a = torch.randn(b, r, ip, device=DEV).bfloat16()
        tiled = TiledQueries(torch.randn(q, o, ip, device=DEV).bfloat16(), 0)
        s = torch.zeros(q, b, device=DEV)
        line = f"  {name:26s}"
        for gen in (2, 3):
            engine(gen)
            t = timed(lambda: ops.pairwise_score(s, 0, tiled, g, a, False), 5)
            line += f" engine{gen} {t:7.3f} ms ({2.0 * q * b * o * ip / t / 1e9:6.0f} TF/s on the score flops)"
        print(line, flush=True)
    print("== implicit-im2col score entry point (pad + per-sample gradients v2 / v3 + score GEMM v2 / v3), Q = b = 1000")
    from torch import nn
    for name, cin, cout, k, st, pd, h in [("conv2 128->128 16x16", 128, 128, 3, 1, 1, 16), ("conv4 128->256 16x16", 128, 256, 3, 1, 1, 16),
                                           ("conv5 256->256 8x8", 256, 256, 3, 1, 1, 8), ("conv1 64->128 k5 s2", 64, 128, 5, 2, 2, 32)]:
        q = b = 1000
        conv = nn.Conv2d(cin, cout, k, stride=st, padding=pd, bias=False)
        x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
        o = (h + 2 * pd - k) // st + 1
        g = torch.randn(b, cout, o, o, device=DEV).bfloat16()
        ip = cin * k * k
        tiled = TiledQueries(torch.randn(q, cout, ip, device=DEV).bfloat16(), 0, conv_channels=cin)
        flops = 2.0 * q * b * cout * ip + 2.0 * b * o * o * cout * ip
        line, outs = f"  {name:24s}", {}
        for gen in (2, 3):
            engine(gen)
            s = torch.zeros(q, b, device=DEV)
            t = timed(lambda: ops.pairwise_score_conv2d(s, 0, tiled, g, x, conv), 5)
            s.zero_()
            ops.pairwise_score_conv2d(s, 0, tiled, g, x, conv)
            outs[gen] = s
            line += f" engine{gen} {t:7.3f} ms {flops / t / 1e9:6.0f} TF/s |"
        d = float((outs[3] - outs[2]).norm() / outs[2].norm())
        print(f"{line} rel diff {d:.1e}{'' if d < 1e-4 else '   <-- MISMATCH'}", flush=True)
    print("== transformer score entry point (kf_pairwise_score_rows: transposes + per-sample gradients + score GEMM)")
    for name, q, b, t_len, o, i in [("bert 768x769 T=128 b=512", 872, 512, 128, 768, 768), ("bert 3072x769 T=128 b=512", 872, 512, 128, 3072, 768),
                                    ("gpt2 768x769 T=512 b=128", 1024, 128, 512, 768, 768), ("gpt2 768x3073 T=512 b=128", 1024, 128, 512, 768, 3072)]:
        ipp = (i + 1 + 7) // 8 * 8
        g = torch.randn(b, t_len, o, device=DEV).bfloat16()
        a = torch.randn(b, t_len, i, device=DEV).bfloat16()
        tiled = TiledQueries(torch.randn(q, o, ipp, device=DEV).bfloat16(), 0)
        flops = 2.0 * q * b * o * (i + 1) + 2.0 * b * t_len * o * (i + 1)
        line, outs = f"  {name:28s}", {}
        for gen in (2, 3):
            engine(gen)
            s = torch.zeros(q, b, device=DEV)
            t = timed(lambda: ops.pairwise_score_rows(s, 0, tiled, g, a, True), 5)
            s.zero_()
            ops.pairwise_score_rows(s, 0, tiled, g, a, True)
            outs[gen] = s
            line += f" engine{gen} {t:7.3f} ms {flops / t / 1e9:6.0f} TF/s |"
        d = float((outs[3] - outs[2]).norm() / outs[2].norm())
        print(f"{line} rel diff {d:.1e}{'' if d < 1e-4 else '   <-- MISMATCH'}", flush=True)
    engine(3)

    print("== Lambda of a conv layer: factored form (im2col + 2 rotations + kf_lambda_accum) vs dense form (kf_lambda_conv2d_accum)")
    for name, cin, cout, k, st, pd, h in [("conv2 128->128 16x16", 128, 128, 3, 1, 1, 16), ("conv4 128->256 16x16", 128, 256, 3, 1, 1, 16),
                                           ("conv1 64->128 k5 s2", 64, 128, 5, 2, 2, 32)]:
        b = 1000
        conv = nn.Conv2d(cin, cout, k, stride=st, padding=pd, bias=False)
        x = torch.randn(b, cin, h, h, device=DEV).bfloat16()
        o = (h + 2 * pd - k) // st + 1
        r = o * o
        g = torch.randn(b, cout, o, o, device=DEV).bfloat16()
        ip = cin * k * k
        q_a = torch.linalg.qr(torch.randn(ip, ip, device=DEV))[0]
        q_g = torch.linalg.qr(torch.randn(cout, cout, device=DEV))[0]
        qa_t, qg_t = q_a.t().contiguous().bfloat16(), q_g.t().contiguous().bfloat16()
        lam_f, lam_d = torch.zeros(cout, ip, device=DEV), torch.zeros(cout, ip, device=DEV)

        def factored():
            patches = ops.im2col(x, conv, False, torch.bfloat16)
            rows = g.flatten(2).transpose(1, 2).contiguous()
            gt = ops.rotate_bf16(rows.reshape(b * r, cout), qg_t)
            at = ops.rotate_bf16(patches.reshape(b * r, ip), qa_t)
            ops.lambda_accum(lam_f, gt, at, b, r)

        geometry = ops.lambda_conv2d_geometry(tuple(x.shape), cout, conv)
        qa_perm, qg16 = ops.conv_patch_order_eigenvectors(q_a, cin, k * k), q_g.bfloat16().contiguous()

        def dense():
            ops.lambda_conv2d_accum(lam_d, ops.rotate_channels(g, qg16), x, geometry, qa_perm)

        tf, td = timed(factored, 5), timed(dense, 5)
        lam_f.zero_(); lam_d.zero_(); factored(); dense()
        d = float((lam_d - lam_f).norm() / lam_f.norm())
        f_alg = b * min(2.0 * cout * ip * (ip + cout) + 2.0 * r * cout * ip, 2.0 * r * (ip * ip + cout * cout + cout * ip))
        print(f"  {name:24s} factored {tf:7.3f} ms | dense {td:7.3f} ms ({f_alg / td / 1e9:6.0f} TF/s on F_lambda) | rel diff {d:.1e}"
              f"{'' if d < 3e-2 else '   <-- MISMATCH'}", flush=True)


if __name__ == "__main__":
    main()
