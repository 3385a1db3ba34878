"""C5 feasibility slice: ONE Llama-3-8B MLP projection at FULL width -- up/gate (O = 14336, I = 4096) or down
(O = 4096, I = 14336), no bias, T = 512 tokens, bf16 -- through every stage of the hot path on one MI355X, timed:

    covariance SYRKs (4096^2 and 14336^2 fp32 accumulators)  ->  kf_eigh_f64 of both  ->  Lambda (bf16 rotations)
    ->  EK-FAC preconditioning of Q queries, truncated to rank k = 64 (the reference's setting for this model,
        examples/openwebtext)  ->  pairwise scores of a train batch against the low-rank queries.

    gpurun -- 'python tools/llama_layer.py [up|down] [--train 32] [--query 8] [--skip-big-eigh]'

Synthetic activations / gradients (randn) stand in for the hooked tensors; the arithmetic parity of exactly these stages at
1/8 width is tests/test_layer_shapes_gpu.py::test_layer_shape_stages_match_oracle[llama-*]."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops

DEV = "cuda:0"


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:58s} {dt * 1e3:10.1f} ms", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="up", choices=["up", "down"])
    ap.add_argument("--train", type=int, default=32)
    ap.add_argument("--query", type=int, default=8)
    ap.add_argument("--rank", type=int, default=64)
    ap.add_argument("--skip-big-eigh", action="store_true")
    args = ap.parse_args()
    o, i = (14336, 4096) if args.which == "up" else (4096, 14336)
    t, b, q, k = 512, args.train, args.query, args.rank
    gen = torch.Generator(device=DEV).manual_seed(0)
    g = (torch.randn(b, t, o, generator=gen, device=DEV) * 0.02).bfloat16()
    a = torch.randn(b, t, i, generator=gen, device=DEV).bfloat16()
    print(f"layer O = {o}, I = {i}, T = {t}; {b} train sequences, {q} queries, rank {k}; D = {o * i / 1e6:.1f} M")

    cov_a = torch.zeros(i, i, device=DEV)
    cov_g = torch.zeros(o, o, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    timed(f"activation covariance {i}^2 ({b * t} rows)", lambda: ops.linear_activation_cov(cov_a, cnt, a, None, False))
    timed(f"gradient covariance {o}^2 ({b * t} rows)", lambda: ops.linear_gradient_cov(cov_g, cnt.clone(), g, None))
    n = float(b * t)
    small, big = (cov_a, cov_g) if i < o else (cov_g, cov_a)
    ev_small = timed(f"kf_eigh_f64 {small.shape[0]}^2", lambda: ops.eigh(small, n))
    print(f"    sweeps {ev_small[2]}")
    if args.skip_big_eigh:
        q_big = torch.linalg.qr(torch.randn(big.shape[0], big.shape[0], device=DEV))[0]
        print(f"kf_eigh_f64 {big.shape[0]}^2 skipped: random orthogonal basis instead")
    else:
        ev_big = timed(f"kf_eigh_f64 {big.shape[0]}^2", lambda: ops.eigh(big, n))
        print(f"    sweeps {ev_big[2]}")
        q_big = ev_big[1].float()
    q_small = ev_small[1].float()
    q_a, q_g = (q_small, q_big) if i < o else (q_big, q_small)
    q_a, q_g = q_a.contiguous(), q_g.contiguous()

    qa_t16, qg_t16 = q_a.t().contiguous().bfloat16(), q_g.t().contiguous().bfloat16()
    lam = torch.zeros(o, i, device=DEV)

    def fit_lambda():
        gt = ops.rotate_bf16(g.reshape(b * t, o), qg_t16)
        at = ops.rotate_bf16(a.reshape(b * t, i), qa_t16)
        ops.lambda_accum(lam, gt, at, b, t)

    timed(f"Lambda ({b} sequences; bf16 rotations + squared product)", fit_lambda)
    lam_inv = ops.inv_lambda(lam, float(b), None)
    gq = (torch.randn(q, t, o, generator=gen, device=DEV) * 0.02).bfloat16()
    aq = torch.randn(q, t, i, generator=gen, device=DEV).bfloat16()
    p = timed(f"EK-FAC preconditioning of {q} queries (bf16 engine)",
              lambda: ops.precondition(gq, aq, False, q_g, q_a, lam_inv, out_dtype=torch.bfloat16, q_a_bf16=q_a.bfloat16().contiguous(),
                                       q_g_t_bf16=qg_t16, q_a_t_bf16=qa_t16))
    left, right = timed(f"rank-{k} factorisation of the {q} query gradients", lambda: ops.low_rank_factors(ops.cast(p, torch.float32), k))
    err = float((ops.low_rank_product(left, right) - p.float()).norm() / p.float().norm())
    print(f"    low-rank residual {err:.3f} (random queries have no low-rank structure; real ones do), "
          f"storage {left.numel() * 2 + right.numel() * 2 >> 20} MiB vs dense {p.numel() * 2 >> 20} MiB")
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    dense = ops.cast(ops.low_rank_product(left, right), torch.bfloat16)
    tiled = timed("expand low-rank queries + k-tile-major layout", lambda: TiledQueries(dense, 0))
    scores = torch.zeros(q, b, device=DEV)
    timed(f"pairwise scores, {q} x {b} (transposes + gradient kernel + score GEMM)", lambda: ops.pairwise_score_rows(scores, 0, tiled, g, a, False))
    timed("  (second call, warm)", lambda: ops.pairwise_score_rows(scores, 0, tiled, g, a, False))
    print(f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; scores finite: {bool(torch.isfinite(scores).all())}")


if __name__ == "__main__":
    main()
