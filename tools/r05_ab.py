"""Round-5 measurements through the public entry points at the layer shapes of the transformer configs.

    ab [score] [cov]      one process: the score entry point (kf_pairwise_score_rows2, two train micro-batches per call as the
                          tracker pairs them) and the covariance entry point (kf_syrk_rows_bf16: activations with the bias column,
                          output gradients) on the K-contiguous path (``KF_TN=0``: transpose_rows + the round-3 / 4 kernels) and
                          on the K-major loop (``KF_TN=1``) with each LDS image; HIP events on the launch stream; results compared.
                          Run it under ``rocprofv3 --kernel-trace --stats`` for the per-kernel averages.
    ab convchunks         the convolution score entry point with its gradient -> score hand-over in 1 / 2 / 3 / 4 / 6 chunks (KF_CONV_CHUNKS)
    replay <workload> <entry> <meta.json>
                          the calls ONE train batch of the workload makes to one entry point (score | cov | lambda), one call per
                          distinct layer shape weighted as the model has them, for ``rocprofv3 --pmc`` passes: every dispatch of the
                          process belongs to that entry point, so bytes per call = sum over its kernels.  (The counter passes of
                          the whole bench command segfault inside rocprofv3 on the transformer workloads -- round 4 -- so the
                          entry points are replayed at the bench's own shapes and batch sizes instead.)  Writes the number of
                          calls and their algorithmic bytes / flops to <meta.json>.

    gpurun -- 'python tools/r05_ab.py ab'
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"

# (O, I, bias, layers of that shape per block) -- bench.py's models
LAYERS = {
    "gpt2_small": dict(T=512, q=1024, train_batch=128, pair=True, factor_batch=128,
                       shapes=[(2304, 768, True, 1), (768, 768, True, 1), (3072, 768, True, 1), (768, 3072, True, 1)]),
    "bert_base": dict(T=128, q=872, train_batch=512, pair=False, factor_batch=512,
                      shapes=[(768, 768, True, 4), (3072, 768, True, 1), (768, 3072, True, 1)]),
    "llama_block": dict(T=512, q=8, train_batch=8, pair=True, factor_batch=8,
                        shapes=[(4096, 4096, False, 2), (1024, 4096, False, 2), (14336, 4096, False, 2), (4096, 14336, False, 1)]),
}


def timed(fn, reps=5, warm=2):
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


def set_tn(tn, image=None):
    os.environ["KF_TN"] = str(tn)
    for key, value in (("KF_TN_IMG", image),):
        if value is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = str(value)


def rand(*shape):
    return torch.randn(*shape, device=DEV).bfloat16()


def score_ab():
    print("== score entry point (per-sample gradients + score GEMM); TF/s on 2 Q b O I' + 2 b R O I'")
    cases = [("gpt2 768x769 T=512 Q=1024 b=2x128", 1024, 128, 128, 512, 768, 768), ("gpt2 2304x769 T=512 Q=1024 b=2x128", 1024, 128, 128, 512, 2304, 768),
             ("gpt2 3072x769 T=512 Q=1024 b=2x128", 1024, 128, 128, 512, 3072, 768), ("gpt2 768x3073 T=512 Q=1024 b=2x128", 1024, 128, 128, 512, 768, 3072),
             ("bert 768x769 T=128 Q=872 b=512 (TN forced)", 872, 512, 0, 128, 768, 768), ("bert 3072x769 T=128 Q=872 b=512 (TN forced)", 872, 512, 0, 128, 3072, 768),
             ("llama 4096x4096 T=512 Q=64 b=2x8 no bias", 64, 8, 8, 512, 4096, 4096)]
    os.environ["KF_TN_MIN_R"] = "64"
    for name, q, b0, b1, t_len, o, i in cases:
        bias = "no bias" not in name
        ipp = (i + int(bias) + 7) // 8 * 8
        b = b0 + b1
        g, a = rand(b, t_len, o), rand(b, t_len, i)
        tiled = TiledQueries(rand(q, o, ipp), 0)
        flops = 2.0 * q * b * o * (i + int(bias)) + 2.0 * b * t_len * o * (i + int(bias))
        second = (g[b0:], a[b0:]) if b1 else None
        line, outs = f"  {name:46s}", {}
        for label, tn, image in (("K-contig", 0, None), ("TN img0", 1, 0), ("TN img1", 1, 1), ("TN img2", 1, 2)):
            set_tn(tn, image)
            s = torch.zeros(q, b, device=DEV)
            t = timed(lambda: ops.pairwise_score_rows(s, 0, tiled, g[:b0], a[:b0], bias, second=second))
            s.zero_()
            ops.pairwise_score_rows(s, 0, tiled, g[:b0], a[:b0], bias, second=second)
            outs[label] = s.clone()
            line += f" {label} {t:7.3f} ms {flops / t / 1e9:5.0f} TF/s |"
        d = max(float((outs[k] - outs["K-contig"]).norm() / outs["K-contig"].norm()) for k in outs)
        print(f"{line} max rel diff {d:.1e}{'' if d < 1e-4 else '   <-- MISMATCH'}", flush=True)
        del g, a, tiled
        torch.cuda.empty_cache()
    os.environ.pop("KF_TN_MIN_R", None)
    set_tn(1)


def cov_ab():
    print("== covariance entry point (kf_syrk_rows_bf16, unmasked rows); TF/s on n d (d + 1)")
    cases = [("gpt2 act 768+1 b=64 T=512", 64, 512, 768, True), ("gpt2 act 3072+1 b=64 T=512", 64, 512, 3072, True),
             ("gpt2 grad 768 b=64 T=512", 64, 512, 768, False), ("gpt2 grad 2304 b=64 T=512", 64, 512, 2304, False),
             ("gpt2 grad 3072 b=64 T=512", 64, 512, 3072, False), ("bert grad 768 b=256 T=128", 256, 128, 768, False),
             ("bert grad 3072 b=256 T=128", 256, 128, 3072, False), ("llama act 4096 b=8 T=512", 8, 512, 4096, False),
             ("llama grad 14336 b=8 T=512", 8, 512, 14336, False)]
    for name, b, t_len, d, bias in cases:
        x = rand(b, t_len, d)
        dd = d + int(bias)
        flops = float(b * t_len) * dd * (dd + 1)
        line, outs = f"  {name:32s}", {}
        for label, tn, image in (("K-contig", 0, None), ("TN img0", 1, 0), ("TN img1", 1, 1), ("TN img2", 1, 2)):
            set_tn(tn, image)
            cov, cnt = torch.zeros(dd, dd, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
            call = (lambda: ops.linear_activation_cov(cov, cnt, x, None, True)) if bias else (lambda: ops.linear_gradient_cov(cov, cnt, x, None, 1.0))
            t = timed(call, 5, 2)
            cov.zero_()
            call()
            outs[label] = cov.clone()
            line += f" {label} {t:7.3f} ms {flops / t / 1e9:5.0f} TF/s |"
        d_rel = max(float((outs[k] - outs["K-contig"]).norm() / outs["K-contig"].norm()) for k in outs)
        print(f"{line} max rel diff {d_rel:.1e}{'' if d_rel < 1e-4 else '   <-- MISMATCH'}", flush=True)
        del x
        torch.cuda.empty_cache()
    set_tn(1)


RESNET9 = [  # (name, cin, cout, k, stride, padding, H): the eight convolutions of bench.py's ResNet-9
    ("conv0 3->64", 3, 64, 3, 1, 1, 32), ("conv1 64->128 k5 s2", 64, 128, 5, 2, 2, 32), ("conv2 128->128", 128, 128, 3, 1, 1, 16),
    ("conv3 128->128", 128, 128, 3, 1, 1, 16), ("conv4 128->256", 128, 256, 3, 1, 1, 16), ("conv5 256->256 8x8", 256, 256, 3, 1, 1, 8),
    ("conv6 256->256 8x8", 256, 256, 3, 1, 1, 8), ("conv7 256->128 6x6", 256, 128, 3, 1, 0, 8),
]


def conv_case(cin, cout, k, s, p, h, q=1000, b=1000):
    from torch import nn

    conv = nn.Conv2d(cin, cout, k, stride=s, padding=p, bias=False)
    x = rand(b, cin, h, h)
    o = (h + 2 * p - k) // s + 1
    g = rand(b, cout, o, o)
    ip = cin * k * k
    tiled = TiledQueries(rand(q, cout, ip), 0, conv_channels=cin)
    flops = 2.0 * q * b * cout * ip + 2.0 * b * o * o * cout * ip
    nbytes = b * (cout * o * o + cin * h * h) * 2 + q * cout * ip * 2 + 2.0 * q * b * 4
    return conv, x, g, tiled, flops, nbytes


def conv_chunks():
    """the convolution gradient -> score hand-over in D-chunks through one chunk-sized workspace region
    (``KF_CONV_CHUNKS``), so that a chunk is still in the 256 MB Infinity Cache when the score GEMM reads it."""
    print("== kf_pairwise_score_conv2d, Q = b = 1000: gradient -> score hand-over in n chunks of output channels (ms per call)")
    total = {n: 0.0 for n in (1, 2, 3, 4, 6)}
    for name, cin, cout, k, s, p, h in RESNET9:
        conv, x, g, tiled, flops, _ = conv_case(cin, cout, k, s, p, h)
        line, outs = f"  {name:22s}", {}
        for n in total:
            os.environ["KF_CONV_CHUNKS"] = str(n)
            sc = torch.zeros(1000, 1000, device=DEV)
            t = timed(lambda: ops.pairwise_score_conv2d(sc, 0, tiled, g, x, conv))
            sc.zero_()
            ops.pairwise_score_conv2d(sc, 0, tiled, g, x, conv)
            outs[n] = sc.clone()
            total[n] += t
            line += f" {n}: {t:6.3f} ({flops / t / 1e9:4.0f} TF/s) |"
        d = max(float((outs[n] - outs[1]).norm() / outs[1].norm()) for n in outs)
        print(f"{line} max rel diff {d:.1e}{'' if d < 1e-4 else '   <-- MISMATCH'}", flush=True)
        del x, g, tiled
        torch.cuda.empty_cache()
    os.environ.pop("KF_CONV_CHUNKS", None)
    print("  all eight layers: " + ", ".join(f"{n} chunk(s) {t:.3f} ms" for n, t in total.items()))


def replay(workload, entry, meta_path):
    if workload == "resnet9":   # one kf_pairwise_score_conv2d call per convolution (KF_CONV_CHUNKS from the environment)
        calls, alg_bytes, alg_flops = 0, 0.0, 0.0
        for _, cin, cout, k, s, p, h in RESNET9:
            conv, x, g, tiled, flops, nbytes = conv_case(cin, cout, k, s, p, h)
            sc = torch.zeros(1000, 1000, device=DEV)
            ops.pairwise_score_conv2d(sc, 0, tiled, g, x, conv)
            torch.cuda.synchronize()
            calls += 1
            alg_bytes += nbytes
            alg_flops += flops
            del x, g, tiled
            torch.cuda.empty_cache()
        with open(meta_path, "w", encoding="utf-8") as handle:
            json.dump({"workload": workload, "entry": entry, "calls": calls, "algorithmic_bytes_per_call": alg_bytes / calls,
                       "algorithmic_flops_per_call": alg_flops / calls, "KF_CONV_CHUNKS": os.environ.get("KF_CONV_CHUNKS", "1")}, handle)
        return
    spec = LAYERS[workload]
    t_len, q = spec["T"], spec["q"]
    calls, alg_bytes, alg_flops = 0, 0.0, 0.0
    for o, i, bias, count in spec["shapes"]:
        ip = i + int(bias)
        ipp = (ip + 7) // 8 * 8
        if entry == "score":
            b0 = spec["train_batch"]
            b1 = b0 if spec["pair"] else 0
            b = b0 + b1
            g, a = rand(b, t_len, o), rand(b, t_len, i)
            tiled = TiledQueries(rand(q, o, ipp), 0)
            s = torch.zeros(q, b, device=DEV)
            second = (g[b0:], a[b0:]) if b1 else None
            for _ in range(count):
                ops.pairwise_score_rows(s, 0, tiled, g[:b0], a[:b0], bias, second=second)
            calls += count
            alg_bytes += count * (b * t_len * (o + i) * 2 + q * o * ip * 2 + 2.0 * q * b * 4)
            alg_flops += count * (2.0 * q * b * o * ip + 2.0 * b * t_len * o * ip)
        elif entry == "cov":
            b = spec["factor_batch"]
            g, a = rand(b, t_len, o), rand(b, t_len, i)
            ca, cg = torch.zeros(ip, ip, device=DEV), torch.zeros(o, o, device=DEV)
            cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
            for _ in range(count):
                ops.linear_activation_cov(ca, cnt, a, None, bias)
                ops.linear_gradient_cov(cg, cnt, g, None, 1.0)
            calls += 2 * count
            alg_bytes += count * b * t_len * (o + i) * 2
            alg_flops += count * float(b * t_len) * (ip * (ip + 1) + o * (o + 1))
        elif entry == "lambda":
            b = spec["factor_batch"]
            g, a = rand(b, t_len, o), rand(b, t_len, i)
            w = ipp
            # random "eigenvector" matrices: the traffic of the rotations does not depend on orthogonality (and torch.linalg.qr
            # segfaults under rocprofv3 --pmc)
            qa_t = torch.zeros(w, w, device=DEV)
            qa_t[:ip, :ip] = torch.randn(ip, ip, device=DEV) / ip ** 0.5
            bias_row = qa_t[:ip, i].contiguous() if bias else None
            qa_t = qa_t.bfloat16().contiguous()
            qg_t = (torch.randn(o, o, device=DEV) / o ** 0.5).bfloat16()
            lam = torch.zeros(o, ip, device=DEV)
            torch.cuda.synchronize()
            for _ in range(count):
                gt_t = ops.rotate_rows_transposed(g, qg_t)
                at_t = ops.rotate_rows_transposed(a, qa_t, bias_row)
                ops.lambda_rows_accum(lam, gt_t, at_t)
            calls += count
            alg_bytes += count * b * t_len * (o + i) * 2
            alg_flops += count * 2.0 * b * t_len * (ip * ip + o * o + o * ip)
        else:
            raise SystemExit(f"unknown entry {entry}")
        torch.cuda.synchronize()
        del g, a
        torch.cuda.empty_cache()
    with open(meta_path, "w", encoding="utf-8") as handle:
        json.dump({"workload": workload, "entry": entry, "calls": calls, "algorithmic_bytes_per_call": alg_bytes / calls,
                   "algorithmic_flops_per_call": alg_flops / calls, "KF_TN": os.environ.get("KF_TN", "1"),
                   "shapes": [list(s) for s in spec["shapes"]], "T": t_len, "queries": q, "train_batch": spec["train_batch"],
                   "paired": spec["pair"], "factor_batch": spec["factor_batch"]}, handle)


if __name__ == "__main__":
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "replay":
        replay(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        which = [w for w in sys.argv[1:] if w in ("score", "cov", "convchunks")] or ["score", "cov"]
        if "convchunks" in which:
            conv_chunks()
        if "score" in which:
            score_ab()
        if "cov" in which:
            cov_ab()
