"""Round-4 A/B measurements through the public entry points (HIP events on the launch stream; run it under
``rocprofv3 --kernel-trace --stats`` to get the per-kernel averages next to the entry-point times):

    score    the transformer score entry point (kf_pairwise_score_rows) at GPT-2 / BERT shapes with the half-tile score GEMM on
             the round-2 lock-step loop, the 256 x 128 loop for 64 x 64 waves (kf_pingpong64.h) and the 512 x 128 wave grid
    lambda   the Lambda update of a sequence layer: round-2 path (2 x rotate_bf16 + lambda_bf16_kernel) against the round-4 path
             (2 x rotate_rows_transposed + lambda_rows_kernel) at BERT / GPT-2 / Llama shapes; results compared
    eigh     kf_eigh_f64 on covariances AS THE PRODUCT STORES THEM -- fp32-accumulated (rank deficient and full rank), bf16
             exported -- factor-first (with the path counters) against the solver that carries V

    gpurun -- 'python tools/r04_ab.py [score] [lambda] [eigh]'
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

DEV = "cuda:0"


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


ENGINES = {"round2": {"KF_HALF_TILE_ENGINE": "2"}, "half64": {}, "wide": {"KF_WIDE_TILE": "1"}}


def set_engine(name):
    for key in ("KF_HALF_TILE_ENGINE", "KF_WIDE_TILE", "KF_SCORE_SHAPE"):
        os.environ.pop(key, None)
    os.environ.update(ENGINES[name])


def score():
    print("== transformer score entry point (transposes + per-sample gradients + score GEMM); TF/s on 2 Q b O I' + 2 b R O I'")
    cases = [("gpt2 768x769 T=512 Q=1024 b=128", 1024, 128, 512, 768, 768), ("gpt2 2304x769 T=512 Q=1024 b=128", 1024, 128, 512, 2304, 768),
             ("gpt2 768x3073 T=512 Q=1024 b=128", 1024, 128, 512, 768, 3072), ("bert 768x769 T=128 Q=872 b=512", 872, 512, 128, 768, 768),
             ("few queries 768x769 T=128 Q=100 b=512", 100, 512, 128, 768, 768)]
    for name, q, b, t_len, o, i in cases:
        ipp = (i + 1 + 7) // 8 * 8
        g = torch.randn(b, t_len, o, device=DEV).bfloat16()
        a = torch.randn(b, t_len, i, device=DEV).bfloat16()
        tiled = TiledQueries(torch.randn(q, o, ipp, device=DEV).bfloat16(), 0)
        flops = 2.0 * q * b * o * (i + 1) + 2.0 * b * t_len * o * (i + 1)
        line, outs = f"  {name:40s}", {}
        for eng in ENGINES:
            set_engine(eng)
            s = torch.zeros(q, b, device=DEV)
            t = timed(lambda: ops.pairwise_score_rows(s, 0, tiled, g, a, True))
            s.zero_()
            ops.pairwise_score_rows(s, 0, tiled, g, a, True)
            outs[eng] = s.clone()
            line += f" {eng} {t:7.3f} ms {flops / t / 1e9:6.0f} TF/s |"
        d = max(float((outs[e] - outs["round2"]).norm() / outs["round2"].norm()) for e in ENGINES)
        print(f"{line} max rel diff {d:.1e}{'' if d < 1e-4 else '   <-- MISMATCH'}", flush=True)
    set_engine("half64")


def lam():
    print("== Lambda update of a sequence layer: rotations + product; TF/s of the PRODUCT on 2 b R O I'")
    cases = [("gpt2 768x769 T=512 b=64", 64, 512, 768, 768), ("gpt2 2304x769 T=512 b=64", 64, 512, 2304, 768),
             ("gpt2 3072x769 T=512 b=64", 64, 512, 3072, 768), ("gpt2 768x3073 T=512 b=64", 64, 512, 768, 3072),
             ("bert 768x769 T=128 b=256", 256, 128, 768, 768), ("bert 3072x769 T=128 b=256", 256, 128, 3072, 768),
             ("llama 4096x4096 T=512 b=16", 16, 512, 4096, 4096), ("llama 14336x4096 T=512 b=8", 8, 512, 14336, 4096)]
    for name, b, r, o, i in cases:
        ip = i + 1
        w = ip + (-ip) % 8
        g = torch.randn(b, r, o, device=DEV).bfloat16()
        a = torch.randn(b, r, i, device=DEV).bfloat16()
        qa_t = torch.zeros(w, w, device=DEV)
        qa_t[:ip, :ip] = torch.linalg.qr(torch.randn(ip, ip, device=DEV))[0].t()
        bias_row = qa_t[:ip, i].contiguous()
        qa_t = qa_t.bfloat16().contiguous()
        qg_t = torch.linalg.qr(torch.randn(o, o, device=DEV))[0].t().contiguous().bfloat16()
        lam_old, lam_new = torch.zeros(o, ip, device=DEV), torch.zeros(o, ip, device=DEV)
        st = {}

        def old_rot():
            st["gt"] = ops.rotate_bf16(g.reshape(b * r, o), qg_t)
            st["at"] = ops.rotate_bf16(a.reshape(b * r, i), qa_t, bias_row)

        def new_rot():
            st["gt_t"] = ops.rotate_rows_transposed(g, qg_t)
            st["at_t"] = ops.rotate_rows_transposed(a, qa_t, bias_row)

        t_or, t_nr = timed(old_rot, 3, 1), timed(new_rot, 3, 1)
        t_op = timed(lambda: ops.lambda_accum(lam_old, st["gt"], st["at"], b, r), 3, 1)
        t_np = timed(lambda: ops.lambda_rows_accum(lam_new, st["gt_t"], st["at_t"]), 3, 1)
        lam_old.zero_(); lam_new.zero_()
        ops.lambda_accum(lam_old, st["gt"], st["at"], b, r)
        ops.lambda_rows_accum(lam_new, st["gt_t"], st["at_t"])
        d = float((lam_new - lam_old).norm() / lam_old.norm())
        fl = 2.0 * b * r * o * ip
        frot = 2.0 * b * r * (o * o + i * ip)
        print(f"  {name:30s} round2: rot {t_or:7.3f} ms ({frot / t_or / 1e9:5.0f} TF/s) product {t_op:7.3f} ms ({fl / t_op / 1e9:5.0f} TF/s) | "
              f"round4: rot {t_nr:7.3f} ms ({frot / t_nr / 1e9:5.0f}) product {t_np:7.3f} ms ({fl / t_np / 1e9:5.0f} TF/s) | rel diff {d:.1e}"
              f"{'' if d < 1e-4 else '   <-- MISMATCH'}", flush=True)
        del st, g, a
        torch.cuda.empty_cache()


def eigh():
    print("== kf_eigh_f64 on product-like covariances: factor-first (paths) vs the solver that carries V (KF_EIGH_CHOLESKY=0)")
    cases = [(769, 300), (769, 6000), (3073, 1500), (3073, 9000), (2304, 20000)]
    if "big" in sys.argv:
        cases.append((14336, 30000))
    for d, n in cases:
        gen = torch.Generator(device=DEV).manual_seed(d + n)
        x = torch.randn(n, d, generator=gen, device=DEV) * torch.logspace(0, -3, d, device=DEV)
        mix = torch.linalg.qr(torch.randn(d, d, generator=gen, device=DEV))[0]
        xm = x @ mix
        cov32 = torch.zeros(d, d, device=DEV)
        for start in range(0, n, 1024):   # accumulated in fp32, batch by batch, as the covariance stage does
            blk = xm[start:start + 1024]
            cov32 += blk.t() @ blk
        del x, xm, mix
        for label, cov, noise in (("fp32", cov32, 0.0), ("bf16", cov32.bfloat16().float(), ops.STORAGE_NOISE[torch.bfloat16])):
            line = f"  d={d:5d} n={n:5d} {label}:"
            ref_vals = None
            for mode in ("factor_first", "carry_V"):
                if d > 8000 and mode == "carry_V":
                    continue
                if mode == "carry_V":
                    os.environ["KF_EIGH_CHOLESKY"] = "0"
                else:
                    os.environ.pop("KF_EIGH_CHOLESKY", None)
                ops.eigh_stats(reset=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                evals, evecs, sweeps = ops.eigh(cov, float(n), noise_rel=noise)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                stats = ops.eigh_stats()
                s = 0.5 * (cov.double() + cov.double().t()) / n
                ortho = float((evecs.t() @ evecs - torch.eye(d, device=DEV, dtype=torch.float64)).abs().max())
                recon = float(((evecs * evals) @ evecs.t() - s).norm() / s.norm())
                if ref_vals is None:
                    ref_vals = torch.linalg.eigvalsh(s) if d <= 8000 else evals
                verr = float((evals - ref_vals).abs().max() / ref_vals.abs().max())
                line += f" {mode} {dt * 1e3:7.0f} ms {sweeps:2d} sweeps ortho {ortho:.1e} recon {recon:.1e} evals {verr:.1e} {stats} |"
                del evecs, s
            print(line, flush=True)
        os.environ.pop("KF_EIGH_CHOLESKY", None)
        del cov32
        torch.cuda.empty_cache()


if __name__ == "__main__":
    torch.manual_seed(0)
    which = [w for w in sys.argv[1:] if w in ("score", "lambda", "eigh")] or ["score", "lambda", "eigh"]
    if "score" in which:
        score()
    if "lambda" in which:
        lam()
    if "eigh" in which:
        eigh()
