"""bench.py -- headline benchmark of the MI355X EK-FAC hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Metric (BASELINE.json): query x train influence pairs/sec (``value``) and EKFAC factor-fit
samples/sec (``factor_fit`` object of the same JSON line).  One "step" is one complete pairwise
stage -- query measurement backward + EK-FAC preconditioning of every query gradient, one full
forward/backward pass over the (sharded) train set with the score kernels in the hooks, and the
scores brought to host memory -- on synthetic data ALREADY RESIDENT IN HBM.  For N > 1 the driver
launches this file under torch.distributed.run; ranks shard the train set (contiguous chunks) and
the queries (strided), exchange over RCCL (factor all-reduce, query-gradient all-gather, score-block
gather) and the job is timed barrier-to-barrier, MAX over ranks ("strong" scaling: the workload is
fixed as N grows).

Workloads (synthetic random-weight models of the BASELINE.json layer shapes):
  resnet9    configs[1] (DEFAULT): CIFAR-10 ResNet-9 (Conv2d tracked), 50 000 train x 1 000 query,
             bf16 autocast, bf16 query gradients (the reference's all_low_precision preset)
  mnist_mlp  configs[0]: 784-1024-1024-1024-10 MLP, 1 000 train x 100 query, fp32
  gpt2_small GPT-2-small-shaped decoder (12 x [c_attn 768->2304, c_proj 768->768, c_fc 768->3072, c_proj 3072->768],
             nn.Linear with bias, T = 512; D = 85.0 M), bf16; scaled down to 2 048 train x 256 query sequences
             (the reference's WikiText-2 example: 4.6 k x 481); only the 48 block Linears are tracked, as there

The ``roofline`` object times the dominant kernel launches (the pairwise-score contraction) with HIP
events on the launch stream inside the timed region; ``cpu_baseline`` times the CPU oracle
(``oracle/ekfac_ref.py``, a torch-CPU restatement of the reference) on a bounded sample of the same
workload on this box's host cores (rank 0, N = 1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from typing import Callable, Dict, List, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16); AMD's 5 PF figure is 2:1 sparse
PEAK_HBM_GBPS = 8000.0


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def mnist_mlp() -> nn.Module:
    return nn.Sequential(nn.Flatten(), nn.Linear(784, 1024), nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU(),
                         nn.Linear(1024, 1024), nn.ReLU(), nn.Linear(1024, 10))


class _Residual(nn.Module):
    def __init__(self, inner: nn.Module) -> None:
        super().__init__()
        self.inner = inner

    def forward(self, x):
        return x + self.inner(x)


class _Scale(nn.Module):
    def __init__(self, weight: float) -> None:
        super().__init__()
        self.weight = weight

    def forward(self, x):
        return x * self.weight


def resnet9() -> nn.Module:
    """Layer shapes of the CIFAR-10 ResNet-9 the reference's example analyses."""
    def block(cin, cout, k=3, stride=1, padding=1):
        return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False), nn.BatchNorm2d(cout), nn.ReLU())

    return nn.Sequential(
        block(3, 64), block(64, 128, k=5, stride=2, padding=2),
        _Residual(nn.Sequential(block(128, 128), block(128, 128))),
        block(128, 256), nn.MaxPool2d(2),
        _Residual(nn.Sequential(block(256, 256), block(256, 256))),
        block(256, 128, padding=0), nn.AdaptiveMaxPool2d((1, 1)), nn.Flatten(), nn.Linear(128, 10, bias=False), _Scale(0.2),
    )


class _GPT2Block(nn.Module):
    def __init__(self, width: int, heads: int) -> None:
        super().__init__()
        self.heads = heads
        self.ln_1, self.ln_2 = nn.LayerNorm(width), nn.LayerNorm(width)
        self.c_attn = nn.Linear(width, 3 * width)
        self.attn_proj = nn.Linear(width, width)
        self.c_fc = nn.Linear(width, 4 * width)
        self.mlp_proj = nn.Linear(4 * width, width)

    def forward(self, x):
        b, t, d = x.shape
        q, k, v = self.c_attn(self.ln_1(x)).split(d, dim=-1)
        q, k, v = (z.reshape(b, t, self.heads, d // self.heads).transpose(1, 2) for z in (q, k, v))
        y = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(b, t, d)
        x = x + self.attn_proj(y)
        return x + self.mlp_proj(F.gelu(self.c_fc(self.ln_2(x)), approximate="tanh"))


class GPT2(nn.Module):
    """GPT-2-shaped decoder with ``nn.Linear`` layers (the reference's example converts HF's ``Conv1D`` to ``Linear``
    first, examples/wikitext/pipeline.py:13-39); random init, no hub access."""

    def __init__(self, layers: int = 12, width: int = 768, heads: int = 12, vocab: int = 50257, positions: int = 512) -> None:
        super().__init__()
        self.wte, self.wpe = nn.Embedding(vocab, width), nn.Embedding(positions, width)
        self.h = nn.ModuleList(_GPT2Block(width, heads) for _ in range(layers))
        self.ln_f = nn.LayerNorm(width)
        self.lm_head = nn.Linear(width, vocab, bias=False)

    def forward(self, ids):
        x = self.wte(ids) + self.wpe(torch.arange(ids.shape[1], device=ids.device))
        for block in self.h:
            x = block(x)
        return self.lm_head(self.ln_f(x))

    def tracked_names(self) -> List[str]:
        return [f"h.{i}.{name}" for i in range(len(self.h)) for name in ("c_attn", "attn_proj", "c_fc", "mlp_proj")]


def gpt2_small() -> nn.Module:
    return GPT2()


def lm_loss(model, batch) -> torch.Tensor:
    """Summed next-token cross-entropy (examples/wikitext/analyze.py:85-103)."""
    ids = batch[0]
    logits = model(ids)[:, :-1]
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), ids[:, 1:].reshape(-1), reduction="sum")


def make_lm_task(tracked: List[str]):
    from kronfluence_amd import Task

    class LanguageModelingTask(Task):
        def compute_train_loss(self, batch, model, sample=False):
            if not sample:
                return lm_loss(model, batch)
            ids = batch[0]
            logits = model(ids)[:, :-1]
            flat = logits.reshape(-1, logits.shape[-1])
            with torch.no_grad():
                drawn = torch.multinomial(torch.softmax(flat.detach().float(), dim=-1), 1).flatten()
            return F.cross_entropy(flat.float(), drawn, reduction="sum")

        def compute_measurement(self, batch, model):
            return lm_loss(model, batch)

        def get_influence_tracked_modules(self):
            return tracked

    return LanguageModelingTask()


def synth_tokens(spec, n: int, seed: int, device) -> Tuple[torch.Tensor, ...]:
    gen = torch.Generator().manual_seed(seed)
    return (torch.randint(0, spec["vocab"], (n, spec["tokens"]), generator=gen).to(device),)


WORKLOADS = {
    "mnist_mlp": dict(model=mnist_mlp, shape=(1, 28, 28), classes=10, n_train=1000, n_query=100, amp=None,
                      factor_batch=1000, train_batch=1000, query_batch=100,
                      cpu_sample=dict(n_train=1000, n_query=100, n_fit=250)),
    "resnet9": dict(model=resnet9, shape=(3, 32, 32), classes=10, n_train=50_000, n_query=1000, amp=torch.bfloat16,
                    factor_batch=1000, train_batch=1000, query_batch=250,
                    cpu_sample=dict(n_train=192, n_query=32, n_fit=64)),
    "gpt2_small": dict(model=gpt2_small, lm=True, vocab=50257, tokens=512, n_train=2048, n_query=256, amp=torch.bfloat16,
                       factor_batch=16, train_batch=16, query_batch=16,
                       cpu_sample=dict(n_train=8, n_query=2, n_fit=4)),
}


def make_task():
    from kronfluence_amd import Task

    class ClassificationTask(Task):
        def compute_train_loss(self, batch, model, sample=False):
            x, y = batch
            logits = model(x)
            if sample:
                with torch.no_grad():
                    y = torch.multinomial(torch.softmax(logits.detach().float(), dim=-1), 1).flatten()
            return F.cross_entropy(logits.float(), y, reduction="sum")

        def compute_measurement(self, batch, model):
            return self.compute_train_loss(batch, model, sample=False)

    return ClassificationTask()


def synth(spec, n: int, seed: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn((n,) + spec["shape"], generator=gen)
    y = torch.randint(0, spec["classes"], (n,), generator=gen)
    return x.to(device), y.to(device)


def tracked_shapes(model) -> List[Tuple[int, int, int]]:
    """(O, I', R) per tracked layer -> D = sum O*I' for the algorithmic flop counts."""
    from kronfluence_amd.module.tracked_module import TrackedModule

    out = []
    for m in model.modules():
        if isinstance(m, TrackedModule):
            w = m.original_module.weight
            o = w.shape[0]
            ip = w[0].numel() + int(m.original_module.bias is not None)
            out.append((o, ip))
    return out


# ------------------------------------------------------------------------------------------------
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("KF_BENCH_WORKLOAD", "resnet9"), choices=sorted(WORKLOADS))
    ap.add_argument("--n-train", type=int, default=None)
    ap.add_argument("--n-query", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--factor-reps", type=int, default=1)
    args = ap.parse_args()

    from kronfluence_amd import FactorArguments, ScoreArguments, ops, prepare_model
    from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
    from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader, perform_eigendecomposition
    from kronfluence_amd.score.pairwise import compute_pairwise_scores_with_loaders
    from kronfluence_amd.utils.dataset import DistributedEvalSampler, DistributedSamplerWithStack, ResidentLoader
    from kronfluence_amd.utils.state import State

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback).")
    state = State()
    world, rank, dev = state.num_processes, state.process_index, state.device
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    spec = WORKLOADS[args.workload]
    n_train = args.n_train or spec["n_train"]
    n_query = args.n_query or spec["n_query"]

    torch.manual_seed(0)
    raw_model = spec["model"]()
    is_lm = bool(spec.get("lm"))
    task = make_lm_task(raw_model.tracked_names()) if is_lm else make_task()
    model = prepare_model(raw_model, task).to(dev)
    make_data = synth_tokens if is_lm else synth
    train = make_data(spec, n_train, 1, dev)
    query = make_data(spec, n_query, 2, dev)
    amp = spec["amp"]
    low = amp == torch.bfloat16  # the reference's all_low_precision preset: bf16 factors / gradients
    # (covariances stay fp32: at least the reference's precision, and the fp64 eigensolver converges faster on them)
    fargs = FactorArguments(use_empirical_fisher=True, amp_dtype=amp, **(dict(
        per_sample_gradient_dtype=torch.bfloat16, lambda_dtype=torch.bfloat16) if low else {}))
    per_dev_q = max(1, min(spec["query_batch"], -(-n_query // world)))
    # hold every preconditioned query gradient resident in HBM (P: n_query x D) -> ONE train pass per step
    accumulate = -(-n_query // (per_dev_q * world))
    sargs = ScoreArguments(amp_dtype=amp, query_gradient_accumulation_steps=accumulate,
                           score_dtype=torch.bfloat16 if amp == torch.bfloat16 else torch.float32,
                           precondition_dtype=torch.bfloat16 if amp == torch.bfloat16 else torch.float32)
    layers = tracked_shapes(model)
    D = sum(o * ip for o, ip in layers)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn: Callable[[], object]) -> Tuple[float, object]:
        barrier()
        t0 = time.perf_counter()
        out = fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    # -- shards: factor fit strided without padding; train contiguous chunks; queries strided
    def factor_loader():
        idx = list(DistributedEvalSampler(range(n_train), world, rank)) if world > 1 else None
        return ResidentLoader(train, spec["factor_batch"], idx)

    def train_loader():
        idx = list(DistributedSamplerWithStack(range(n_train), world, rank)) if world > 1 else None
        loader = ResidentLoader(train, spec["train_batch"], idx)
        return loader

    def query_loader():
        if world > 1:
            from torch.utils.data import DistributedSampler

            idx = list(DistributedSampler(range(n_query), world, rank, shuffle=False, drop_last=False))
        else:
            idx = None
        return ResidentLoader(query, per_dev_q, idx)

    # -- factor fit (cov + eigen + lambda), timed per sub-stage --------------------------------------
    fit_times = {"covariance": 0.0, "eigendecomposition": 0.0, "lambda": 0.0}
    for _ in range(args.factor_reps + 1 if args.factor_reps > 0 else 1):  # first pass = warm-up; --factor-reps 0: single cold pass
        t_cov, (_, cov) = timed(lambda: fit_covariance_matrices_with_loader(model, state, task, factor_loader(), fargs))
        if world > 1:  # the reference hands factors to the other ranks through the file system
            box = [cov]
            dist.broadcast_object_list(box, src=0)
            cov = box[0]
        t_eig, eig = timed(lambda: perform_eigendecomposition(cov, model, state, fargs))
        t_lam, (_, lam) = timed(lambda: fit_lambda_matrices_with_loader(model, state, task, factor_loader(), fargs, eig))
        if world > 1:
            box = [lam]
            dist.broadcast_object_list(box, src=0)
            lam = box[0]
        fit_times = {"covariance": t_cov, "eigendecomposition": t_eig, "lambda": t_lam}
    # inputs of the pairwise stage resident in HBM before the timed region starts (the stage API also
    # accepts the CPU dicts the factor stage returns; that adds one 37 MB H2D per call for mnist_mlp)
    factors = {k: {n: v.to(dev) for n, v in d.items()} for k, d in {**eig, **lam}.items()}
    fit_total = sum(fit_times.values())

    # -- pairwise stage: W warm-up steps, K timed steps ------------------------------------------------
    def step():
        # ResidentLoader.dataset keeps the FULL dataset length (remainder / padding logic needs it)
        return compute_pairwise_scores_with_loaders(factors, model, state, task, query_loader(), per_dev_q,
                                                    train_loader(), sargs, fargs, None)

    for _ in range(args.warmup):
        step()
    import gc

    gc.collect()
    gc.disable()  # no cyclic-GC pause between steps either (the stage loops already pause it inside a stage)
    ops.SCORE_EVENT_LOG = []
    seg0 = torch.cuda.memory_stats().get("segment.all.allocated", 0)
    barrier()
    t0 = time.perf_counter()
    scores = None
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        scores = step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
    barrier()
    elapsed = time.perf_counter() - t0
    new_segments = torch.cuda.memory_stats().get("segment.all.allocated", 0) - seg0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    events, ops.SCORE_EVENT_LOG = ops.SCORE_EVENT_LOG, None
    kernel_ms = sum(s.elapsed_time(e) for s, e, _ in events)
    kernel_flops = sum(f for _, _, f in events)
    launches = len(events)

    pairs = float(n_query) * float(n_train) * args.steps
    value = pairs / elapsed
    if rank == 0:
        assert scores["all_modules"].shape == (n_query, n_train), scores["all_modules"].shape
        assert bool(torch.isfinite(scores["all_modules"]).all())

    # -- CPU baseline: the oracle on this box's host cores, bounded sample ---------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ekfac_ref as ref

        cs = spec["cpu_sample"]
        ct, cq = min(cs["n_train"], n_train), min(cs["n_query"], n_query)
        cpu_model = spec["model"]()
        cpu_model.load_state_dict({k.replace(".original_module", ""): v.cpu() for k, v in model.state_dict().items()
                                   if "_constant" not in k})
        engine = ref.OracleEngine(cpu_model, module_names=raw_model.tracked_names() if is_lm else None)
        loss = lm_loss if is_lm else (lambda m, b: F.cross_entropy(m(b[0]), b[1], reduction="sum"))  # noqa: E731
        ctrain = tuple(t[:ct].cpu() for t in train)
        cquery = tuple(t[:cq].cpu() for t in query)

        def chunks(d, bs):
            return [tuple(t[i:i + bs] for t in d) for i in range(0, d[0].shape[0], bs)]

        cpu_eig = {k: {n: v.float() for n, v in d.items()} for k, d in eig.items()}
        cpu_lam = {k: {n: (v.float() if v.is_floating_point() else v) for n, v in d.items()} for k, d in lam.items()}
        tb = min(spec["train_batch"], 250)
        t0 = time.perf_counter()
        cscores = engine.pairwise_scores(chunks(cquery, min(cq, 100)), chunks(ctrain, tb), loss, loss, cpu_eig, cpu_lam, 1e-8)
        cpu_pair_s = time.perf_counter() - t0
        nf = min(cs["n_fit"], n_train)
        fit_sample = tuple(t[:nf].cpu() for t in train)
        t0 = time.perf_counter()
        ccov = engine.fit_covariance(chunks(fit_sample, tb), loss)
        engine.fit_lambda(chunks(fit_sample, tb), loss, cpu_eig)
        cpu_fit_s = time.perf_counter() - t0
        cpu = {
            "value": cq * ct / cpu_pair_s, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{cq} query x {ct} train pairwise stage (precondition + train pass) in {cpu_pair_s:.1f}s; "
                      f"factor fit (covariance+lambda, eigh excluded) on {nf} samples in {cpu_fit_s:.1f}s",
            "host_cpu_count": os.cpu_count(),
            "factor_fit_samples_per_sec": nf / cpu_fit_s,
        }
        if ct == n_train and cq == n_query:  # same workload, same factors: parity of the whole stage in the bench itself
            err = float((scores["all_modules"].double() - cscores.double()).norm() / cscores.double().norm())
            cpu["gpu_vs_cpu_scores_rel_F"] = err

    if rank == 0:
        achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        peak = PEAK_BF16_MFMA_TFLOPS if sargs.score_dtype == torch.bfloat16 else PEAK_FP32_MFMA_TFLOPS
        line = {
            "metric": "pairwise_influence_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "step_ms": [round(x, 2) for x in step_ms], "hipmalloc_segments_in_timed_region": new_segments,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if amp == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "n_train": n_train, "n_query": n_query, "tracked_layers": len(layers),
                       "D": D, "input_dtype": "bf16-autocast" if amp is not None else "f32",
                       "score_dtype": str(sargs.score_dtype), "query_gradient_accumulation_steps": accumulate,
                       "train_batch": spec["train_batch"], "query_batch": per_dev_q,
                       "parallelism": f"train-shard-dp{world}"},
            "roofline": {
                "bound": "mfma", "kernel": "kf_pairwise_score (score_r1_kernel | gemm_kernel + gemm_nt_bf16_kernel)",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": None,
                "launches": launches, "avg_launch_ms": kernel_ms / max(launches, 1),
                "algorithmic_flops_per_launch": kernel_flops / max(launches, 1),
                "kernel_share_of_step": (kernel_ms * 1e-3) / elapsed if elapsed > 0 else None,
            },
            "factor_fit": {
                "samples_per_sec": n_train / fit_total, "seconds": fit_times, "n_fit": n_train,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
