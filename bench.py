"""bench.py -- headline benchmark of the MI355X EK-FAC hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-extras]

Metric (BASELINE.json): query x train influence pairs/sec (``value``) and EKFAC factor-fit
samples/sec (``factor_fit`` object of the same JSON line).  One "step" is one complete pairwise
stage -- query measurement backward + EK-FAC preconditioning of every query gradient, one full
forward/backward pass over the (sharded) train set with the score kernels in the hooks, and the
scores brought to host memory -- on synthetic data ALREADY RESIDENT IN HBM.  For N > 1 the driver
launches this file under torch.distributed.run; ranks shard the train set (contiguous chunks) and
the queries (strided), exchange over RCCL (factor all-reduce, query-gradient all-gather, score-block
gather; eigenvector broadcasts) INSIDE the timed regions, and the job is timed barrier-to-barrier,
MAX over ranks ("strong" scaling: the workload is fixed as N grows).

Workloads (synthetic random-weight models of the BASELINE.json layer shapes):
  resnet9    configs[1] (DEFAULT headline: the config the metric is quoted on that fits one GPU):
             CIFAR-10 ResNet-9 (Conv2d tracked), 50 000 train x 1 000 query, bf16 autocast, bf16 query gradients
  mnist_mlp  configs[0]: 784-1024-1024-1024-10 MLP, 1 000 train x 100 query, fp32, damping 1e-8
  bert_base  configs[2]: 12-layer BERT-base-shaped encoder + pooler + 2-way head (74 tracked Linears, D = 85.6 M),
             T = 128 with random-length padding masks, 872 queries, fp32 factors / bf16 gradients (``--workload bert_base``
             alone defaults to 8 192 of the 67 349 train sequences; ``--n-train`` overrides)
  gpt2_small configs[3]: GPT-2-small-shaped decoder (48 tracked block Linears with bias, T = 512, D = 85.0 M), bf16 incl. the
             covariances (the reference's all-low-precision preset); 2 048 train x 1 024 query sequences by default
  llama_block configs[4] as a one-block slice: the seven projections of a Llama-3-8B decoder block at full width (D = 218 M),
             T = 512, rank-64 low-rank query gradients, covariances released as their eigendecompositions finish; 64 x 8 sequences

With the default workload the same JSON line also carries ``targets.mnist_mlp`` (N = 1; the north-star target: GPU pairs/s,
CPU-oracle pairs/s on the SAME full workload, their ratio and the GPU-vs-oracle score error at damping 1e-8) and
``other_configs``, one timed step each, measured in the same run: N = 1 -- bert_base at its FULL 67 349 x 872 and gpt2_small at
16 384 x 1 024 (factors fitted on a bounded prefix, ``factor_fit.n_fit``); N > 1 -- gpt2_small, the config the north-star
scaling target is stated on, sharded like the headline.  ``python bench.py --gpus N`` typed without a launcher re-executes itself
under ``torch.distributed.run`` (one rank per GPU, RCCL) and the line then carries ``exchanges``.

``roofline`` times the dominant kernel calls (the pairwise-score contraction ``kf_pairwise_score*``) with HIP events on
the launch stream inside the timed region; ``roofline_cov`` / ``roofline_lambda`` / ``roofline_lambda_update`` do the same for the
covariance calls, the Lambda kernels and the whole Lambda update of a hook during the factor fit; ``cpu_baseline`` times the CPU oracle (``oracle/ekfac_ref.py``) on a bounded
sample of the same workload on this box's host cores (rank 0, N = 1 only).
"""

from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time
from typing import Callable, Dict, List, Optional, Tuple

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
# models
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


# The MODEL's batch norm (eval mode: a per-channel affine map) on ATen's own kernel instead of MIOpen's: ``MIOpenBatchNormFwdInferSpatialEst``
# takes 0.32 ms per call on a [1000, C, H, W] bf16 batch -- 17 % of a ResNet-9 pairwise step (``device_busy.top_model_kernels``, round 6) --
# about five times the HBM time of its operands (tools/bn_probe.py: 427 us against 80 us for ATen's kernel at [1000, 64, 32, 32]).  Same module class, same arithmetic, a different library kernel: the same kind of
# model-side setting as ``torch.backends.cudnn.benchmark`` below.  ``--miopen-batchnorm`` switches back.
NATIVE_BATCH_NORM = True


class _BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        if not (NATIVE_BATCH_NORM and not self.training and x.is_cuda):
            return super().forward(x)
        # (this build ignores the ``cudnn_enabled`` argument of torch.batch_norm -- tools/bn_probe.py: MIOpen either way, 427 us --
        # so the global switch is flipped around the call: ``batch_norm_transform_input_kernel``, 80 us on the same tensor)
        enabled = torch.backends.cudnn.enabled
        torch.backends.cudnn.enabled = False
        try:
            return super().forward(x)
        finally:
            torch.backends.cudnn.enabled = enabled


def resnet9() -> nn.Module:
    """Layer shapes of the CIFAR-10 ResNet-9 the reference's example analyses."""
    def block(cin, cout, k=3, stride=1, padding=1):
        return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False), _BatchNorm2d(cout), nn.ReLU())

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


class _LlamaBlock(nn.Module):
    """One Llama-3-8B decoder block at FULL width (examples/openwebtext/pipeline.py loads Meta-Llama-3-8B; here random
    init): RMSNorm, grouped-query causal attention (32 heads, 8 KV heads), SwiGLU MLP -- seven bias-free Linears:
    q / o (4096, 4096), k / v (1024, 4096), gate / up (14336, 4096), down (4096, 14336)."""

    def __init__(self, width: int = 4096, heads: int = 32, kv_heads: int = 8, inter: int = 14336) -> None:
        super().__init__()
        self.heads, self.kv_heads, self.head_dim = heads, kv_heads, width // heads
        self.input_norm, self.post_norm = nn.RMSNorm(width, eps=1e-5), nn.RMSNorm(width, eps=1e-5)
        self.q_proj = nn.Linear(width, width, bias=False)
        self.k_proj = nn.Linear(width, kv_heads * self.head_dim, bias=False)
        self.v_proj = nn.Linear(width, kv_heads * self.head_dim, bias=False)
        self.o_proj = nn.Linear(width, width, bias=False)
        self.gate_proj = nn.Linear(width, inter, bias=False)
        self.up_proj = nn.Linear(width, inter, bias=False)
        self.down_proj = nn.Linear(inter, width, bias=False)

    def forward(self, x):
        b, t, d = x.shape
        h = self.input_norm(x)
        q = self.q_proj(h).reshape(b, t, self.heads, self.head_dim).transpose(1, 2)
        k = self.k_proj(h).reshape(b, t, self.kv_heads, self.head_dim).transpose(1, 2)
        v = self.v_proj(h).reshape(b, t, self.kv_heads, self.head_dim).transpose(1, 2)
        rep = self.heads // self.kv_heads
        y = F.scaled_dot_product_attention(q, k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1), is_causal=True)
        x = x + self.o_proj(y.transpose(1, 2).reshape(b, t, d))
        h = self.post_norm(x)
        return x + self.down_proj(F.silu(self.gate_proj(h)) * self.up_proj(h))


class LlamaSlice(nn.Module):
    """``blocks`` Llama-3-8B decoder blocks between a (reduced-vocabulary, untracked) embedding and head: the slice of
    configs[4] one GPU's worth of layers stands for.  Tracked: the seven projections of every block ("Linear layers only")."""

    def __init__(self, blocks: int = 1, width: int = 4096, vocab: int = 32000, heads: int = 32, kv_heads: int = 8,
                 inter: int = 14336) -> None:
        super().__init__()
        self.embed = nn.Embedding(vocab, width)
        self.layers = nn.ModuleList(_LlamaBlock(width, heads, kv_heads, inter) for _ in range(blocks))
        self.norm = nn.RMSNorm(width, eps=1e-5)
        self.lm_head = nn.Linear(width, vocab, bias=False)

    def forward(self, ids):
        x = self.embed(ids)
        for block in self.layers:
            x = block(x)
        return self.lm_head(self.norm(x))

    def tracked_names(self) -> List[str]:
        return [f"layers.{i}.{name}" for i in range(len(self.layers))
                for name in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")]


def llama_block() -> nn.Module:
    return LlamaSlice(blocks=WORKLOADS["llama_block"].get("blocks", 1))


class _BertLayer(nn.Module):
    def __init__(self, width: int, heads: int, inter: int) -> None:
        super().__init__()
        self.heads = heads
        self.query, self.key, self.value = nn.Linear(width, width), nn.Linear(width, width), nn.Linear(width, width)
        self.attn_out = nn.Linear(width, width)
        self.ln_attn = nn.LayerNorm(width)
        self.intermediate = nn.Linear(width, inter)
        self.output = nn.Linear(inter, width)
        self.ln_out = nn.LayerNorm(width)

    def forward(self, x, key_mask):
        b, t, d = x.shape
        q, k, v = (lin(x).reshape(b, t, self.heads, d // self.heads).transpose(1, 2) for lin in (self.query, self.key, self.value))
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=key_mask).transpose(1, 2).reshape(b, t, d)
        x = self.ln_attn(x + self.attn_out(y))
        return self.ln_out(x + self.output(F.gelu(self.intermediate(x))))


class Bert(nn.Module):
    """BERT-base-shaped sequence classifier (the reference's GLUE example loads ``bert-base-cased`` from the hub,
    examples/glue/pipeline.py:22-36; here random init): every ``nn.Linear`` is tracked as there -- per layer
    query / key / value / attention-output (768, 769), intermediate (3072, 769), output (768, 3073), plus the pooler
    (768, 769, applied to the [CLS] row: one row per sample) and the classifier (2, 769)."""

    def __init__(self, layers: int = 12, width: int = 768, heads: int = 12, inter: int = 3072, vocab: int = 28996,
                 positions: int = 512, labels: int = 2) -> None:
        super().__init__()
        self.word, self.position, self.kind = nn.Embedding(vocab, width), nn.Embedding(positions, width), nn.Embedding(2, width)
        self.ln_embed = nn.LayerNorm(width)
        self.layers = nn.ModuleList(_BertLayer(width, heads, inter) for _ in range(layers))
        self.pooler = nn.Linear(width, width)
        self.classifier = nn.Linear(width, labels)

    def forward(self, ids, mask):
        t = ids.shape[1]
        x = self.word(ids) + self.position(torch.arange(t, device=ids.device)) + self.kind.weight[0]
        x = self.ln_embed(x)
        key_mask = mask[:, None, None, :].to(torch.bool)  # padded keys are never attended to
        for layer in self.layers:
            x = layer(x, key_mask)
        return self.classifier(torch.tanh(self.pooler(x[:, 0])))


def bert_base() -> nn.Module:
    return Bert()


# ------------------------------------------------------------------------------------------------
# losses, tasks, data (plain functions are shared with the CPU oracle)
# ------------------------------------------------------------------------------------------------
def lm_loss(model, batch) -> torch.Tensor:
    """Summed next-token cross-entropy (examples/wikitext/analyze.py:85-103)."""
    ids = batch[0]
    logits = model(ids)[:, :-1]
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), ids[:, 1:].reshape(-1), reduction="sum")


def glue_loss(model, batch) -> torch.Tensor:
    ids, mask, labels = batch
    return F.cross_entropy(model(ids, mask).float(), labels, reduction="sum")


def glue_margin(model, batch) -> torch.Tensor:
    """Negative summed correct-class margin (examples/glue/analyze.py:107-127)."""
    ids, mask, labels = batch
    logits = model(ids, mask).float()
    rows = torch.arange(logits.shape[0], device=logits.device)
    correct = logits[rows, labels]
    others = logits.clone()
    others[rows, labels] = float("-inf")
    return -(correct - others.logsumexp(dim=-1)).sum()


def image_loss(model, batch) -> torch.Tensor:
    return F.cross_entropy(model(batch[0]).float(), batch[1], reduction="sum")


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


def make_glue_task():
    from kronfluence_amd import Task

    class TextClassificationTask(Task):
        def compute_train_loss(self, batch, model, sample=False):
            if not sample:
                return glue_loss(model, batch)
            ids, mask, _ = batch
            logits = model(ids, mask)
            with torch.no_grad():
                drawn = torch.multinomial(torch.softmax(logits.detach().float(), dim=-1), 1).flatten()
            return F.cross_entropy(logits.float(), drawn, reduction="sum")

        def compute_measurement(self, batch, model):
            return glue_margin(model, batch)

        def get_attention_mask(self, batch):
            return batch[1]

    return TextClassificationTask()


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


def synth_tokens(spec, n: int, seed: int, device) -> Tuple[torch.Tensor, ...]:
    gen = torch.Generator().manual_seed(seed)
    return (torch.randint(0, spec["vocab"], (n, spec["tokens"]), generator=gen).to(device),)


def synth_glue(spec, n: int, seed: int, device) -> Tuple[torch.Tensor, ...]:
    """Token ids, random-length padding masks (at least T/16 real tokens), binary labels."""
    gen = torch.Generator().manual_seed(seed)
    t = spec["tokens"]
    ids = torch.randint(0, spec["vocab"], (n, t), generator=gen)
    lengths = torch.randint(max(2, t // 16), t + 1, (n,), generator=gen)
    mask = (torch.arange(t)[None, :] < lengths[:, None]).to(torch.int64)
    labels = torch.randint(0, 2, (n,), generator=gen)
    return (ids.to(device), mask.to(device), labels.to(device))


def synth(spec, n: int, seed: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn((n,) + spec["shape"], generator=gen)
    y = torch.randint(0, spec["classes"], (n,), generator=gen)
    return x.to(device), y.to(device)


WORKLOADS = {
    "mnist_mlp": dict(model=mnist_mlp, kind="image", shape=(1, 28, 28), classes=10, n_train=1000, n_query=100, amp=None,
                      factor_batch=1000, train_batch=1000, query_batch=100,
                      cpu_sample=dict(n_train=1000, n_query=100, n_fit=250)),
    "resnet9": dict(model=resnet9, kind="image", shape=(3, 32, 32), classes=10, n_train=50_000, n_query=1000,
                    # (round 6: factor batches of 2 000 and train batches of 2 048 images -- whole 256-column tiles of the score GEMM, half the
                    #  launches: 83.3 -> 86.8 M pairs/s, covariance 0.447 -> 0.410 s, Lambda 0.573 -> 0.553 s; 4 096: 84.8 M)
                    amp=torch.bfloat16, factor_batch=2000, train_batch=2048, query_batch=250,
                    cpu_sample=dict(n_train=192, n_query=32, n_fit=64)),
    "bert_base": dict(model=bert_base, kind="glue", vocab=28996, tokens=128, n_train=8192, n_query=872, full_n_train=67_349,
                      amp=torch.bfloat16, fp32_factors=True, factor_batch=512, train_batch=512, query_batch=109,
                      cpu_sample=dict(n_train=16, n_query=4, n_fit=8)),
    "gpt2_small": dict(model=gpt2_small, kind="lm", vocab=50257, tokens=512, n_train=2048, n_query=1024,
                       full_n_train=100_000, full_n_query=2000, amp=torch.bfloat16, low_cov=True, factor_batch=128, train_batch=128,
                       query_batch=32, cpu_sample=dict(n_train=8, n_query=2, n_fit=4)),
    # configs[4] (OpenWebText Llama-3-8B, Linear layers only, 100k x 1k on 8 GPUs, AMP bf16) as a ONE-BLOCK slice at full
    # width: all seven projections tracked, T = 512, the reference's rank-64 low-rank query gradients
    # (examples/openwebtext/files/scores_raw/score_arguments.json), all-low-precision factors; the covariances (3 of
    # 14336^2) are released one by one as their eigendecompositions finish.  The vocabulary of the untracked embedding /
    # head is reduced to 32 000.
    "llama_block": dict(model=llama_block, kind="lm", vocab=32000, tokens=512, n_train=64, n_query=8, full_n_train=100_000,
                        full_n_query=1000, amp=torch.bfloat16, low_cov=True, factor_batch=8, train_batch=8, query_batch=4,
                        low_rank=64, release_covariances=True, blocks=1, cpu_sample=dict(n_train=2, n_query=1, n_fit=1)),
}


def factor_arguments(spec):
    """The FactorArguments ``bench.py`` runs a workload with (also used by tests/test_configs_gpu.py).  bf16 workloads use
    the reference's low-precision gradients (per_sample_gradient_dtype / lambda_dtype bf16); the covariances stay fp32 (BERT:
    "fp32 factors / bf16 grads") unless the workload says ``low_cov`` -- the reference's ``all_low_precision`` preset
    (utils/common/factor_arguments.py:38-47): hooked tensors are cast to bf16 ahead of the covariance update, so every
    layer (LayerNorm outputs under autocast are fp32) runs on the bf16 LDS-DMA covariance kernel; accumulation stays fp32."""
    from kronfluence_amd import FactorArguments

    amp = spec["amp"]
    extra = {}
    if amp == torch.bfloat16:
        extra.update(per_sample_gradient_dtype=torch.bfloat16, lambda_dtype=torch.bfloat16)
        if spec.get("low_cov"):
            extra.update(activation_covariance_dtype=torch.bfloat16, gradient_covariance_dtype=torch.bfloat16)
    return FactorArguments(use_empirical_fisher=True, amp_dtype=amp, **extra)


def score_arguments(spec, n_query: int, world: int, per_dev_q: int, passes: int = 1):
    """ScoreArguments of a workload: the preconditioned query gradients held resident in HBM (P: n_query x D) -> ONE train pass per
    step; ``passes`` > 1 when P does not fit (GPT-2-small at its stated 2 000 queries: 340 GB): the query batches are accumulated in
    ``passes`` groups, each followed by its own train pass (reference score/pairwise.py:133-293)."""
    from kronfluence_amd import ScoreArguments

    amp = spec["amp"]
    low = amp == torch.bfloat16
    accumulate = -(-(-(-n_query // (per_dev_q * world))) // max(1, passes))
    return ScoreArguments(amp_dtype=amp, query_gradient_accumulation_steps=accumulate,
                          score_dtype=torch.bfloat16 if low else torch.float32,
                          precondition_dtype=torch.bfloat16 if low else torch.float32,
                          query_gradient_low_rank=spec.get("low_rank"))


def workload_parts(spec, raw_model):
    """-> (Task, data maker, oracle train loss, oracle measurement, oracle mask fn, tracked names or None)."""
    kind = spec["kind"]
    if kind == "lm":
        names = raw_model.tracked_names()
        return make_lm_task(names), synth_tokens, lm_loss, lm_loss, None, names
    if kind == "glue":
        return make_glue_task(), synth_glue, glue_loss, glue_margin, (lambda batch: batch[1]), None
    return make_task(), synth, image_loss, image_loss, None, None


def tracked_shapes(model) -> List[Tuple[int, int]]:
    """(O, I') per tracked layer -> D = sum O*I' for the algorithmic flop counts."""
    from kronfluence_amd.module.tracked_module import TrackedModule

    out = []
    for m in model.modules():
        if isinstance(m, TrackedModule):
            w = m.original_module.weight
            out.append((w.shape[0], w[0].numel() + int(m.original_module.bias is not None)))
    return out


def _event_summary(events, peak_tflops: float, kernel: str, elapsed_s: Optional[float] = None, other_kernel_ms: float = 0.0) -> Optional[dict]:
    """Roofline object from ``[(start_event, end_event, algorithmic_flops, algorithmic_bytes)]``."""
    if not events:
        return None
    ms = sum(s.elapsed_time(e) for s, e, _, _ in events)
    flops = sum(f for _, _, f, _ in events)
    nbytes = sum(b for _, _, _, b in events)
    if ms <= 0:
        return None
    achieved = flops / (ms * 1e-3) / 1e12
    out = {
        "bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak_tflops, "unit": "TFLOP/s",
        "frac": achieved / peak_tflops, "traffic": None, "launches": len(events), "avg_launch_ms": ms / len(events),
        "algorithmic_flops_per_launch": flops / len(events), "algorithmic_bytes_per_launch": nbytes / len(events),
        "algorithmic_GBps": nbytes / (ms * 1e-3) / 1e9, "hbm_frac_of_8TBps": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
    }
    if elapsed_s:
        out["kernel_share_of_region"] = (ms * 1e-3) / elapsed_s
        # the rest of the region: the model's own forward / backward (MIOpen, hipBLASLt, attention), autograd and hook overhead --
        # outside the hand-written kernels but inside the metric; it caps what faster kernels can buy
        out["model_share_of_region"] = max(0.0, 1.0 - (ms + other_kernel_ms) * 1e-3 / elapsed_s)
    return out


def kernel_source_hash() -> str:
    """sha256 over the sources of the kernels a kept PMC profile reports (every HIP source and header but the eigensolver's):
    identifies the code the profile was taken on."""
    import glob
    import hashlib

    digest = hashlib.sha256()
    paths = sorted(glob.glob(os.path.join(ROOT, "kronfluence_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "kronfluence_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for path in paths:
        if os.path.basename(path) == "kf_eigh.hip":
            continue   # the eigensolver: none of its kernels is in the PMC summary (its evidence is the eigh logs under profiles/)
        with open(path, "rb") as handle:
            digest.update(os.path.basename(path).encode() + b"\0" + handle.read())
    return digest.hexdigest()


def _pmc_traffic(workload: str) -> Optional[dict]:
    """Per-launch HBM bytes of the hot kernels, from the kept ``rocprofv3 --pmc`` passes of the SAME bench command
    (``profiles/pmc_<workload>.json``, written by tools/pmc_summary.py; FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md).  The counters cannot be collected inside this process, so the numbers are only as current as that
    file: it records a hash of the kernel sources it was taken on, and a file taken on OTHER sources is refused (traffic
    null, with the reason) instead of being quoted.  None when no such file has been committed."""
    path = os.path.join(ROOT, "profiles", f"pmc_{workload}.json")
    if not os.path.exists(path):
        return None
    with open(path, encoding="utf-8") as handle:
        summary = json.load(handle)
    if summary.get("kernel_source_sha256") != kernel_source_hash():
        return {"stale": f"profiles/pmc_{workload}.json was taken on other kernel sources "
                         f"({str(summary.get('kernel_source_sha256'))[:12]} != {kernel_source_hash()[:12]}): not quoted"}
    return summary


# C5 parity bound: scores of the rank-64 bf16 path against the oracle's fp64 contraction of the SAME hooked tensors
# and factor pairs.  Every operand is bf16 (8 mantissa bits: 2e-3 per rounding) and the two row products U = G L, V = A' R^T are held
# in bf16 before the k-long dot, so the floor is a few 1e-3 (measured 4.3e-3 .. 5.8e-3 on 1 - 4 full-width blocks); a wrong kernel
# (a dropped k-tile, a mis-strided factor) shows up at 1e-1 and above.  Above the bound the EXTRA is marked failed (``ok: false``),
# never the headline; tests/test_configs_gpu.py::test_llama_full_width_block_bench_parity holds one full-width block to the same bound.
LOW_RANK_PARITY_BOUND = 2e-2


def _low_rank_parity(model, step: Callable[[int], object], n_sub: int) -> Optional[dict]:
    """C5 parity inside the bench: one more pairwise pass over the first ``n_sub`` train samples with plain torch
    hooks riding along; at every tracked layer's backward the hooked (activation, output gradient) and the low-rank factor pair the
    product holds for ALL queries go through the oracle's fp64 restatement of ``"qik,qko,b...i,b...o->qb"`` (module/linear.py:83-99;
    oracle/ekfac_ref.py, evaluated on the GPU in fp64), summed over layers, and are compared with the scores the pass returns.  The
    checker, outside every timed region."""
    try:
        from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
        from kronfluence_amd.utils.constants import ACCUMULATED_PRECONDITIONED_GRADIENT_NAME
        from oracle import ekfac_ref as ref

        want: Dict[str, torch.Tensor] = {}
        handles, seen = [], {"layers": 0}
        for m in [x for x in model.modules() if isinstance(x, TrackedModule)]:
            def fwd(mod, inputs, output, m=m):
                x = inputs[0].detach()

                def bwd(grad, x=x, m=m):
                    held = m.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME]
                    if m.current_mode != ModuleMode.PAIRWISE_SCORE or not isinstance(held, list):
                        return   # (the query passes run the same model: only the train pass is checked)
                    left, right = held
                    g = grad.detach().double()
                    a = x.to(grad.dtype).double()   # the product consumes the activation in the gradient's (autocast) dtype
                    block = ref.linear_pairwise_score_low_rank(left.double(), right.double(), a, g, m.original_module.bias is not None)
                    want["sum"] = block if "sum" not in want else want["sum"] + block
                    seen["layers"] += 1
                output.register_hook(bwd)
            handles.append(m.register_forward_hook(fwd))
        try:
            got = step(n_sub)["all_modules"].double()
        finally:
            for h in handles:
                h.remove()
        if "sum" not in want:
            return {"error": "no layer held a low-rank factor pair"}
        ref_scores = want["sum"].cpu()
        err = float((got - ref_scores).norm() / ref_scores.norm())
        return {"scores_rel_F_vs_fp64_low_rank_contraction": err, "bound": LOW_RANK_PARITY_BOUND, "ok": bool(err <= LOW_RANK_PARITY_BOUND),
                "queries": got.shape[0], "train_samples": got.shape[1],
                "layer_batches_checked": seen["layers"],
                "what": "scores of one extra pass vs the oracle's fp64 low-rank contraction (linear.py:83-99) on the hooked tensors and "
                        "the very factor pairs the product held, all tracked layers summed"}
    except Exception as error:
        return {"error": f"{type(error).__name__}: {error}"[:200]}


def _library_kernel_pattern():
    """Regex that matches the (demangled) name of a kernel of libkronfluence_hip.so: every ``*_kernel`` the HIP sources define, in
    the anonymous namespace or in ``kf::`` (torch's own kernels live in ``at::native::``)."""
    import glob
    import re

    names = set()
    for path in glob.glob(os.path.join(ROOT, "kronfluence_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "kronfluence_amd", "csrc", "*.h")):
        with open(path, encoding="utf-8") as handle:
            names.update(re.findall(r"void ([A-Za-z0-9_]+_kernel)\b", handle.read()))
    return re.compile(r"^(?:void )?(?:\(anonymous namespace\)::|kf::)(?:" + "|".join(sorted(names)) + r")\b")


def _device_busy(step: Callable[[int, int], object], count: int, queries: int) -> Optional[dict]:
    """Where a pairwise step spends its wall time on the device: one step over the first ``count`` train samples
    and the first ``queries`` queries un-profiled (wall clock), the same step again under the torch profiler (device activity only: roctracer kernel records), kernel
    durations summed by owner.  Everything runs on one stream, so the sum IS the busy time: ``device_busy_frac`` = all kernels and
    copies / wall; ``idle_frac`` = the rest -- the GPU waiting for the host (Python hooks, autograd, ctypes launches).  Outside
    the timed region; MIOpen / hipBLASLt heuristics are warm by then."""
    try:
        from torch.autograd import DeviceType
        from torch.profiler import ProfilerActivity, profile

        pattern = _library_kernel_pattern()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(count, queries)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(count, queries)
            torch.cuda.synchronize()
        ours = model = copies = 0.0
        launches_ours = launches_model = 0
        by_name: Dict[str, List[float]] = {}
        for e in prof.key_averages():
            if e.device_type != DeviceType.CUDA:
                continue
            us = float(e.self_device_time_total)
            if pattern.match(e.key):
                ours += us
                launches_ours += e.count
            elif e.key.startswith(("Memcpy", "Memset", "__amd_rocclr_")):
                copies += us
            else:
                model += us
                launches_model += e.count
                entry = by_name.setdefault(e.key, [0.0, 0])
                entry[0] += us
                entry[1] += e.count
        busy = (ours + model + copies) * 1e-6
        # who the "model" share is: MIOpen / hipBLASLt / ATen kernels by device time; the step runs after the
        # timed region, so MIOpen's find-mode searches are long over
        top_model = [{"kernel": k.split("(")[0][:96], "calls": int(v[1]), "seconds": v[0] * 1e-6, "frac_of_wall": v[0] * 1e-6 / wall}
                     for k, v in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:5]]
        # where the device waits: idle gaps (> 20 us) between consecutive device activities, summed by the activity that PRECEDES the gap
        spans = sorted((e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type == DeviceType.CUDA)
        gaps: Dict[str, List[float]] = {}
        frontier, last = None, ""
        for start, end, name in spans:
            if frontier is not None and start - frontier > 20.0:
                gaps.setdefault(last, []).append(start - frontier)
            if frontier is None or end > frontier:
                frontier, last = end, name
        top = sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:6]
        gap_total = sum(sum(v) for v in gaps.values()) * 1e-6
        return {"n_train": count, "n_query": queries, "wall_s": wall, "kf_kernel_s": ours * 1e-6, "model_kernel_s": model * 1e-6, "copy_s": copies * 1e-6,
                "idle_gaps_over_20us_s": gap_total,
                "largest_idle_after": [{"after": k.replace("(anonymous namespace)::", "").split("(")[0][:70], "gaps": len(v), "seconds": sum(v) * 1e-6} for k, v in top],
                "kf_kernel_launches": launches_ours, "model_kernel_launches": launches_model, "top_model_kernels": top_model,
                "device_busy_frac": busy / wall, "idle_frac": max(0.0, 1.0 - busy / wall),
                "kf_kernel_frac": ours * 1e-6 / wall, "model_kernel_frac": model * 1e-6 / wall,
                "method": "torch profiler (device activity) on one extra step after the timed region; kernel durations / the wall "
                          "clock of the same step un-profiled; kf = kernels of libkronfluence_hip.so, model = everything else "
                          "(MIOpen, hipBLASLt, attention, elementwise, softmax)"}
    except Exception as error:  # diagnostics must never take the measurement down
        return {"error": f"{type(error).__name__}: {error}"[:200]}


# ------------------------------------------------------------------------------------------------
def run_workload(name: str, state, n_train: Optional[int], n_query: Optional[int], steps: int, warmup: int,
                 factor_reps: int, cpu_baseline: bool, n_fit: Optional[int] = None, warm_n_train: Optional[int] = None,
                 busy_n_train: Optional[int] = None, query_passes: Optional[int] = None, phase_split: bool = False) -> dict:
    """``n_fit``: fit the factors on the first ``n_fit`` train samples only (the pairwise stage does not care how many samples
    the factors saw; used by the full-size extras to keep the default run within minutes -- reported in ``factor_fit.n_fit``).
    ``warm_n_train``: the warm-up steps score against the first ``warm_n_train`` train samples (one-time costs -- allocator
    growth, GEMM / MIOpen heuristics, the k-tile-major query layout code paths -- without paying a full-size step).
    ``busy_n_train``: train samples of the two extra steps that measure ``device_busy`` (default: all)."""
    from kronfluence_amd import ops, prepare_model
    from kronfluence_amd.utils import comm
    from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
    from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader, perform_eigendecomposition
    from kronfluence_amd.score.pairwise import compute_pairwise_scores_with_loaders
    from kronfluence_amd.utils.dataset import DistributedEvalSampler, DistributedSamplerWithStack, ResidentLoader

    world, rank, dev = state.num_processes, state.process_index, state.device
    spec = WORKLOADS[name]
    n_train = n_train or spec["n_train"]
    n_query = n_query or spec["n_query"]
    n_fit = min(n_fit or n_train, n_train)

    torch.manual_seed(0)
    raw_model = spec["model"]()
    task, make_data, oracle_loss, oracle_measure, oracle_mask, tracked = workload_parts(spec, raw_model)
    model = prepare_model(raw_model, task).to(dev)
    train = make_data(spec, n_train, 1, dev)
    query = make_data(spec, n_query, 2, dev)
    if spec.get("channels_last") and spec["kind"] == "image":
        model = model.to(memory_format=torch.channels_last)
        train = (train[0].contiguous(memory_format=torch.channels_last),) + tuple(train[1:])
        query = (query[0].contiguous(memory_format=torch.channels_last),) + tuple(query[1:])
    amp = spec["amp"]
    low = amp == torch.bfloat16
    fargs = factor_arguments(spec)
    per_dev_q = max(1, min(spec["query_batch"], -(-n_query // world)))
    layers = tracked_shapes(model)
    D = sum(o * ip for o, ip in layers)
    if query_passes is None:
        # dense query gradients (2 bytes each in the bf16 presets) must fit beside the model's own passes: at most 62 % of the device
        held = float(n_query) * D * (2 if low else 4) if not spec.get("low_rank") else 0.0
        query_passes = max(1, int(-(-held // (0.62 * torch.cuda.get_device_properties(dev).total_memory))))
    sargs = score_arguments(spec, n_query, world, per_dev_q, query_passes)
    accumulate = sargs.query_gradient_accumulation_steps

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
        idx = list(DistributedEvalSampler(range(n_fit), world, rank)) if world > 1 else (None if n_fit == n_train else list(range(n_fit)))
        return ResidentLoader(train, spec["factor_batch"], idx)

    def train_loader(count: int = n_train):
        subset = train if count == n_train else tuple(t[:count] for t in train)
        idx = list(DistributedSamplerWithStack(range(count), world, rank)) if world > 1 else None
        return ResidentLoader(subset, spec["train_batch"], idx)

    # N > 1, the query side (score/query_exchange.py): strided query shard + per-layer all-gather over xGMI ("gather", the
    # reference's way) or every rank preconditioning all queries itself ("replicate"); planned from bytes against flops
    exchange_plan = None
    if world > 1:
        from kronfluence_amd.score import query_exchange as qx

        def one_query_forward():
            with torch.autocast(device_type="cuda", enabled=amp is not None, dtype=amp):
                task.compute_measurement(batch=tuple(t[:1] for t in query), model=model)
        exchange_plan = qx.plan_query_exchange(
            layers, qx.probe_rows(model, one_query_forward), n_query, world,
            score_dtype=torch.bfloat16 if low else torch.float32, precondition_dtype=torch.bfloat16 if low else torch.float32,
            low_rank=spec.get("low_rank"), backend=dist.get_backend())
    replicate_queries = exchange_plan is not None and exchange_plan.mode == "replicate"

    def query_loader(count: int = n_query):
        subset = query if count == n_query else tuple(t[:count] for t in query)
        idx = None
        if world > 1 and not replicate_queries:
            from torch.utils.data import DistributedSampler

            idx = list(DistributedSampler(range(count), world, rank, shuffle=False, drop_last=False))
        loader = ResidentLoader(subset, per_dev_q, idx)
        loader.kf_replicated_queries = replicate_queries
        return loader

    # -- factor fit (cov + eigen + lambda), timed per sub-stage.  Every exchange (factor all-reduce, eigenvector
    #    broadcasts) happens INSIDE the timed calls; every rank keeps the reduced factors in HBM (no host round trip).
    fit_times = {}
    fit_events: Dict[str, list] = {}
    passes = factor_reps + 1 if factor_reps > 0 else 1  # first pass = warm-up; --factor-reps 0: single cold pass
    for index in range(passes):
        ops.EVENT_LOG = {} if index == passes - 1 else None
        comm.EXCHANGE_LOG = {} if (index == passes - 1 and world > 1) else None
        t_cov, (_, cov) = timed(lambda: fit_covariance_matrices_with_loader(model, state, task, factor_loader(), fargs,
                                                                             all_ranks=True, cpu=False))
        ops.eigh_stats(reset=True)
        t_eig, eig = timed(lambda: perform_eigendecomposition(cov, model, state, fargs, cpu=False,
                                                              release_covariances=bool(spec.get("release_covariances"))))
        eigh_paths = ops.eigh_stats()   # this rank's share of the 2L problems: factor-first solves / fall-backs / Cholesky retries
        t_lam, (_, lam) = timed(lambda: fit_lambda_matrices_with_loader(model, state, task, factor_loader(), fargs, eig,
                                                                         all_ranks=True, cpu=False))
        fit_times = {"covariance": t_cov, "eigendecomposition": t_eig, "lambda": t_lam}
        fit_events, ops.EVENT_LOG = (ops.EVENT_LOG or {}), None
        fit_exchanges, comm.EXCHANGE_LOG = comm.summary(comm.EXCHANGE_LOG), None
    factors = {k: {n: v.to(dev) for n, v in d.items()} for k, d in {**eig, **lam}.items()}
    eig_dims = sorted({int(v.shape[0]) for d in (eig["activation_eigenvalues"], eig["gradient_eigenvalues"]) for v in d.values()})
    # the eigendecomposition is reported in seconds beside its size proxy sum_l (I'_l^3 + O_l^3) over ALL 2L matrices
    # (what the reference solves; layers that share an input are solved once here: eigh_paths counts the problems actually solved)
    eig_sum_d3 = float(sum(int(v.shape[0]) ** 3 for d in (eig["activation_eigenvalues"], eig["gradient_eigenvalues"]) for v in d.values()))
    del cov
    fit_total = sum(fit_times.values())

    # -- pairwise stage: W warm-up steps, K timed steps ------------------------------------------------
    def step(count: int = n_train, queries: int = n_query):
        return compute_pairwise_scores_with_loaders(factors, model, state, task, query_loader(queries), per_dev_q,
                                                    train_loader(count), sargs, fargs, None)

    for _ in range(warmup):
        step(min(warm_n_train or n_train, n_train))
    gc.collect()
    gc.disable()  # no cyclic-GC pause between steps either (the stage loops already pause it inside a stage)
    ops.EVENT_LOG = {}
    comm.EXCHANGE_LOG = {} if world > 1 else None
    seg0 = torch.cuda.memory_stats().get("segment.all.allocated", 0)
    barrier()
    t0 = time.perf_counter()
    scores = None
    step_ms = []
    from kronfluence_amd.score import pairwise as pairwise_stage
    # the extras (one timed step of tens of seconds) also split the step into its query phase and train passes: ONE more device
    # synchronisation per held-query window; the headline's timed region is left untouched
    pairwise_stage.STAGE_LOG = {} if phase_split else None
    for _ in range(steps):
        ts = time.perf_counter()
        scores = step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
    barrier()
    elapsed = time.perf_counter() - t0
    phases, pairwise_stage.STAGE_LOG = pairwise_stage.STAGE_LOG, None
    new_segments = torch.cuda.memory_stats().get("segment.all.allocated", 0) - seg0
    gc.enable()
    score_events, ops.EVENT_LOG = ops.EVENT_LOG, None          # the timed region's calls only (the diagnostics below run more steps)
    score_exchanges, comm.EXCHANGE_LOG = comm.summary(comm.EXCHANGE_LOG), None
    busy = None
    if rank == 0 and world == 1 and os.environ.get("KF_BENCH_BUSY", "1") != "0":
        # a MINIATURE step -- at most 16 train batches against at most 4 query batches: the profiler drops device records on long
        # steps (GPT-2 at 2 048 x 2 000: 28 k of ~60 k kernels came back) -- with both phases of the stage in it
        busy = _device_busy(step, min(n_train, busy_n_train or 16 * spec["train_batch"]), min(n_query, 4 * per_dev_q))
    parity = None
    if rank == 0 and world == 1 and spec.get("low_rank") and os.environ.get("KF_BENCH_PARITY", "1") != "0":
        parity = _low_rank_parity(model, step, min(n_train, spec["train_batch"]))
    if os.environ.get("KF_BENCH_PROFILE") and rank == 0:
        # diagnostics only (outside the timed region): one more step under the torch profiler, per-kernel device totals to a file
        from torch.profiler import ProfilerActivity, profile
        count = int(os.environ.get("KF_BENCH_PROFILE_TRAIN", n_train))
        stacks = os.environ.get("KF_BENCH_PROFILE_STACKS") == "1"
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=stacks, record_shapes=stacks) as prof:
            step(min(count, n_train))
            torch.cuda.synchronize()
        with open(os.environ["KF_BENCH_PROFILE"], "w") as fh:
            fh.write(prof.key_averages().table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=90))
            if stacks:  # who copies: the copy-like operators by input shape and by Python call site
                copies = ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::pad", "aten::cat")
                rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in copies]
                rows.sort(key=lambda e: -e.device_time_total)
                fh.write("\n\n== copy-like operators by input shape (device us, calls)\n")
                for e in rows[:40]:
                    fh.write(f"{e.key:18s} {e.device_time_total:10.0f} {e.count:6d}  {e.input_shapes}\n")
                rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key in copies]
                rows.sort(key=lambda e: -e.device_time_total)
                fh.write("\n== copy-like operators by call site (device us, calls)\n")
                for e in rows[:25]:
                    frames = [f for f in e.stack if "site-packages/torch" not in f and "<built-in" not in f][:5]
                    fh.write(f"{e.key:18s} {e.device_time_total:10.0f} {e.count:6d}\n    " + "\n    ".join(frames) + "\n")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    torch.cuda.synchronize()
    pairs = float(n_query) * float(n_train) * steps
    value = pairs / elapsed
    if rank == 0:
        assert scores["all_modules"].shape == (n_query, n_train), scores["all_modules"].shape
        assert bool(torch.isfinite(scores["all_modules"]).all())
    peak_mem = torch.cuda.max_memory_allocated() / 2**30

    # -- CPU baseline: the oracle on this box's host cores, bounded sample ---------------------------
    cpu = None
    if rank == 0 and world == 1 and cpu_baseline:
        from oracle import ekfac_ref as ref

        cs = spec["cpu_sample"]
        ct, cq = min(cs["n_train"], n_train), min(cs["n_query"], n_query)
        cpu_model = spec["model"]()
        cpu_model.load_state_dict({k.replace(".original_module", ""): v.cpu() for k, v in model.state_dict().items()
                                   if "_constant" not in k})
        engine = ref.OracleEngine(cpu_model, module_names=tracked)
        ctrain = tuple(t[:ct].cpu() for t in train)
        cquery = tuple(t[:cq].cpu() for t in query)

        def chunks(d, bs):
            return [tuple(t[i:i + bs] for t in d) for i in range(0, d[0].shape[0], bs)]

        cpu_eig = {k: {n: v.float().cpu() for n, v in d.items()} for k, d in eig.items()}
        cpu_lam = {k: {n: (v.float() if v.is_floating_point() else v).cpu() for n, v in d.items()} for k, d in lam.items()}
        tb = min(spec["train_batch"], 250)
        t0 = time.perf_counter()
        cscores = engine.pairwise_scores(chunks(cquery, min(cq, 100)), chunks(ctrain, tb), oracle_measure, oracle_loss,
                                         cpu_eig, cpu_lam, 1e-8)
        cpu_pair_s = time.perf_counter() - t0
        nf = min(cs["n_fit"], n_train)
        fit_sample = tuple(t[:nf].cpu() for t in train)
        t0 = time.perf_counter()
        engine.fit_covariance(chunks(fit_sample, tb), oracle_loss, oracle_mask)
        engine.fit_lambda(chunks(fit_sample, tb), oracle_loss, cpu_eig)
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
            cpu["damping"] = 1e-8

    result = None
    if rank == 0:
        peak = PEAK_BF16_MFMA_TFLOPS if low else PEAK_FP32_MFMA_TFLOPS
        roofline = _event_summary(score_events.get("pairwise_score", []), peak,
                                  "kf_pairwise_score*: score_r1_kernel (one row per sample) | conv_pad_phases_kernel / "
                                  "transpose_rows_kernel + psg_gemm_v3_kernel / psg_gemm_pp_kernel (per-sample gradients) + "
                                  "score_gemm_v3_kernel (256 x 256; score_gemm_v4_kernel<TA,TB> for the 256 x 128 / 128 x 256 "
                                  "shapes) (score GEMM; the dominant kernel)", elapsed,
                                  other_kernel_ms=sum(a.elapsed_time(b) for a, b, _, _ in score_events.get("precondition", [])))
        if roofline is not None:
            pre = score_events.get("precondition", [])
            roofline["precondition_share_of_region"] = sum(a.elapsed_time(b) for a, b, _, _ in pre) * 1e-3 / elapsed if pre else 0.0
        traffic = _pmc_traffic(name)
        if traffic is not None and "stale" in traffic:
            if roofline is not None:
                roofline["traffic_source"] = traffic["stale"]
            traffic = None
        if roofline is not None and traffic is not None:
            roofline["traffic"] = traffic.get("kf_pairwise_score_bytes_per_launch")
            roofline["traffic_source"] = traffic.get("source")
            roofline["mfma_util"] = traffic.get("mfma_util")
            # counter bytes over algorithmic bytes of the SAME calls (the replayed entry points carry their own algorithmic figure: one
            # train batch against all queries; the timed region's average launch may differ, e.g. BERT's ragged last batch)
            algorithmic = traffic.get("kf_pairwise_score_algorithmic_bytes_per_launch") or roofline["algorithmic_bytes_per_launch"]
            if roofline["traffic"] and algorithmic:
                roofline["traffic_over_algorithmic"] = roofline["traffic"] / algorithmic
        roofline_cov = _event_summary(fit_events.get("syrk_accum", []), peak, "covariance calls: kf_syrk_accum | kf_syrk_rows_bf16 | "
                                      "kf_conv2d_cov_accum | kf_syrk_planes_bf16 (pad / transpose + cov_gemm_v2_kernel + cov_finalize_kernel; "
                                      "algorithmic bytes = one read of the rows handed over)", fit_times["covariance"])
        if roofline_cov is not None and traffic is not None and traffic.get("cov_call_bytes_per_launch") is not None:
            roofline_cov["traffic"] = traffic["cov_call_bytes_per_launch"]
            roofline_cov["traffic_over_algorithmic"] = traffic["cov_call_bytes_per_launch"] / traffic["cov_call_algorithmic_bytes_per_launch"]
            roofline_cov["traffic_source"] = "all kernels of a covariance call (memset + cov_gemm + finalize), per call: " + str(traffic.get("source"))
            roofline_cov["mfma_util"] = traffic.get("cov_gemm_mfma_util")
        elif roofline_cov is not None and traffic is not None and traffic.get("cov_gemm_bytes_per_launch") is not None:
            roofline_cov["traffic"] = traffic["cov_gemm_bytes_per_launch"]
            roofline_cov["traffic_source"] = "the covariance GEMM kernel with the most launches alone (cov_gemm_v3_kernel / cov_gemm_v2_kernel; per launch), same PMC passes as roofline.traffic"
            roofline_cov["mfma_util"] = traffic.get("cov_gemm_mfma_util")
        # covariance calls whose rows arrive in fp32 (LayerNorm outputs under autocast with fp32 factors: BERT) run on the exact-fp32
        # MFMA engine: their own line against that engine's peak
        roofline_cov_f32 = (_event_summary(fit_events.get("syrk_accum_f32", []), PEAK_FP32_MFMA_TFLOPS, "kf_syrk_accum on fp32 rows "
                                           "(syrk_kernel<F32>: v_mfma_f32_32x32x2_f32)", fit_times["covariance"]) if low else None)
        if not low:
            roofline_cov = _event_summary(fit_events.get("syrk_accum_f32", []) + fit_events.get("syrk_accum", []), peak,
                                          "covariance calls: kf_syrk_accum (fp32 rows)", fit_times["covariance"])
        roofline_lambda = _event_summary(fit_events.get("lambda_accum", []), peak, "kf_lambda_rows_accum / kf_lambda_accum (factored form: the product of "
                                         "the rotated factors, 2 b R O I' flops) | kf_lambda_conv2d_accum (dense form of a Conv2d layer: "
                                         "pad + psg_gemm + rotate_gemm_v3<sumsq>, 2 b R O I' + 2 b O I'^2 flops)", fit_times["lambda"])
        if roofline_lambda is not None and traffic is not None and traffic.get("kf_lambda_bytes_per_launch") is not None:
            roofline_lambda["traffic"] = traffic["kf_lambda_bytes_per_launch"]
            roofline_lambda["traffic_source"] = ("lambda_bf16_kernel / lambda_kernel (factored calls) and conv_pad_phases + psg_gemm_v3<1> + "
                                                 "rotate_gemm_v3<1> (dense calls) per call, same PMC passes as roofline.traffic")
            roofline_lambda["mfma_util"] = traffic.get("lambda_mfma_util")
        result = {
            "metric": "pairwise_influence_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "step_ms": [round(x, 2) for x in step_ms], "hipmalloc_segments_in_timed_region": new_segments,
            # wall seconds of the timed steps by phase (rank 0; only where ``phase_split``): the two terms of DESIGN.md section 6's time model
            "phase_seconds": ({k: v / steps for k, v in phases.items()} if phases else None),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if low else "f32",
            "data": "synthetic",
            "config": {"workload": name, "n_train": n_train, "n_query": n_query, "tracked_layers": len(layers),
                       "D": D, "input_dtype": "bf16-autocast" if amp is not None else "f32",
                       "score_dtype": str(sargs.score_dtype), "query_gradient_accumulation_steps": accumulate, "query_passes": query_passes,
                       **({"blocks": spec["blocks"]} if "blocks" in spec else {}),
                       "train_batch": spec["train_batch"], "factor_batch": spec["factor_batch"], "query_batch": per_dev_q,
                       "parallelism": f"train-shard-dp{world}",
                       **({"memory_format": "channels_last"} if spec.get("channels_last") and spec["kind"] == "image" else {}),
                       **({"model_batch_norm": "aten" if NATIVE_BATCH_NORM else "miopen"} if name == "resnet9" else {}),
                       **({"warmup_n_train": min(warm_n_train, n_train)} if warm_n_train else {}),
                       **({"scaled_from": {"n_train": spec.get("full_n_train"), "n_query": spec.get("full_n_query", spec["n_query"])}}
                          if n_train < spec.get("full_n_train", 0) else {})},
            "roofline": roofline,
            "roofline_cov": roofline_cov,
            "roofline_cov_f32": roofline_cov_f32,
            "roofline_lambda": roofline_lambda,
            # the WHOLE Lambda update of a hook (eigenbasis rotations included) against F_lambda, the cheaper of the two exact
            # formulations
            "roofline_lambda_update": _event_summary(fit_events.get("lambda_update", []), peak, "LambdaTracker backward hook: "
                                                     "rotations / per-sample gradient + squared product; algorithmic flops = F_lambda",
                                                     fit_times["lambda"]),
            # samples_per_sec divides by ALL of the fit, eigendecomposition included -- a fixed cost per model, so at a bounded
            # n_fit it is not a rate; the per-sample stages and the fixed seconds are therefore also given apart
            "factor_fit": {"samples_per_sec": n_fit / fit_total, "seconds": fit_times, "n_fit": n_fit,
                           "eigen_dims": eig_dims, "eigen_sum_d3": eig_sum_d3, "eigh_paths": eigh_paths,
                           "covariance_samples_per_sec": n_fit / fit_times["covariance"],
                           "lambda_samples_per_sec": n_fit / fit_times["lambda"],
                           "eigendecomposition_fixed_seconds": fit_times["eigendecomposition"]},
            "peak_hbm_gib": round(peak_mem, 1),
            # where the wall time of a step goes on the device: hand-written kernels / the model's own kernels / idle (host bound)
            "device_busy": busy,
            # low-rank workloads: the scores of one extra pass against the oracle's fp64 contraction on the hooked tensors
            "parity": parity,
            # rank 0's collectives (RCCL over xGMI; all inside the timed regions): seconds are stream time between events
            # around each call -- for the query all-gather only the wait still exposed after overlapping with backward
            "exchanges": ({"backend": dist.get_backend(), "ranks": world, "query_exchange": exchange_plan.mode,
                           "query_exchange_plan": exchange_plan.to_dict(), "factor_fit": fit_exchanges,
                           "pairwise_timed_steps": score_exchanges} if world > 1 else None),
            "cpu_baseline": cpu,
        }
    del factors, eig, lam, model, train, query, scores
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    return result


# ------------------------------------------------------------------------------------------------
# the printed line: compact by construction (the driver parses the LAST stdout line; r05's 30 KB object was not parsed)
# ------------------------------------------------------------------------------------------------
LINE_TARGET_BYTES = 8192
LINE_HARD_CAP_BYTES = 16384
EXTRAS_FILE = "bench_extras.json"

_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "mfma_util", "launches",
                  "avg_launch_ms", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "kernel_share_of_region",
                  "model_share_of_region")


def _sig(x, digits: int = 6):
    """Floats to ``digits`` significant digits (a 17-digit double is 2-3x the characters and none of the information)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _compact_roofline(r: Optional[dict], keys=_ROOFLINE_KEYS) -> Optional[dict]:
    """Numbers only (+ the two one-word strings ``bound`` / ``unit``): the kernel / method prose lives in DESIGN.md section 7."""
    if not isinstance(r, dict):
        return None
    return {k: r[k] for k in keys if k in r and (k in ("bound", "unit") or not isinstance(r[k], (str, dict, list)))}


def _compact_fit(f: Optional[dict]) -> Optional[dict]:
    if not isinstance(f, dict):
        return None
    return {"samples_per_sec": f.get("samples_per_sec"), "n_fit": f.get("n_fit"), "seconds": f.get("seconds"),
            "covariance_samples_per_sec": f.get("covariance_samples_per_sec"), "lambda_samples_per_sec": f.get("lambda_samples_per_sec")}


def _compact_busy(b: Optional[dict]) -> Optional[dict]:
    if not isinstance(b, dict):
        return None
    if "error" in b:
        return {"error": str(b["error"])[:120]}
    return {k: b.get(k) for k in ("idle_frac", "kf_kernel_frac", "model_kernel_frac")}


def _compact_exchanges(e: Optional[dict]) -> Optional[dict]:
    """``{backend, ranks, query_exchange, <stage>: {<kind>: [calls, bytes, seconds]}}``."""
    if not isinstance(e, dict):
        return None
    out = {k: e[k] for k in ("backend", "ranks", "query_exchange") if k in e}
    for stage in ("factor_fit", "pairwise_timed_steps"):
        if isinstance(e.get(stage), dict):
            out[stage] = {kind: [v.get("calls"), v.get("bytes"), v.get("seconds")] for kind, v in e[stage].items()}
    return out


def compact_line(full: dict) -> dict:
    """The ONE JSON object ``bench.py`` prints: the contract's headline keys, ``config``, numeric ``roofline`` (of the headline
    entry point) + the three factor-fit roofline fractions, ``cpu_baseline``, ``factor_fit``, ``targets.mnist_mlp`` and one short record per
    ``other_configs`` entry.  Everything else ``run_workload`` measures (per-kernel prose, ``device_busy`` gap tables, the extra
    rooflines, ``eigh_paths`` ...) is the FULL object, written to ``bench_extras.json`` beside this script and to stderr."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(full.get("config") or {})
    cfg.pop("scaled_from", None)
    line["config"] = cfg
    line["roofline"] = _compact_roofline(full.get("roofline"))
    if isinstance(full.get("roofline"), dict) and line["roofline"] is not None:
        line["roofline"]["kernel"] = "kf_pairwise_score* (entry point: per-sample-gradient GEMM + score GEMM)"
    cpu = full.get("cpu_baseline")
    line["cpu_baseline"] = (None if not isinstance(cpu, dict) else
                            {**{k: cpu.get(k) for k in ("value", "unit", "cores", "kind")}, "sample": str(cpu.get("sample", ""))[:160],
                             **({"factor_fit_samples_per_sec": cpu["factor_fit_samples_per_sec"]} if "factor_fit_samples_per_sec" in cpu else {})})
    line["factor_fit"] = _compact_fit(full.get("factor_fit"))
    for key in ("roofline_cov", "roofline_lambda", "roofline_lambda_update"):
        line[key] = _compact_roofline(full.get(key), ("frac", "traffic_over_algorithmic", "mfma_util", "avg_launch_ms"))
    line["device_busy"] = _compact_busy(full.get("device_busy"))
    line["peak_hbm_gib"] = full.get("peak_hbm_gib")
    if full.get("exchanges") is not None:
        line["exchanges"] = _compact_exchanges(full["exchanges"])
    mnist = (full.get("targets") or {}).get("mnist_mlp")
    if isinstance(mnist, dict):
        line["targets"] = {"mnist_mlp": {k: mnist.get(k) for k in ("ratio", "target_ratio", "scores_rel_F_vs_cpu_oracle", "target_rel",
                                                                  "gpu_pairs_per_sec", "cpu_pairs_per_sec", "cpu_cores", "ms_per_step")}}
    others = full.get("other_configs")
    if isinstance(others, dict):
        short = {}
        for name, r in others.items():
            if not isinstance(r, dict) or "error" in r:
                short[name] = {"error": str((r or {}).get("error"))[:160]}
                continue
            rc, roof = r.get("config") or {}, r.get("roofline") or {}
            parity = r.get("parity")
            short[name] = {
                "value": r.get("value"), "unit": r.get("unit"), "n_gpus": r.get("n_gpus"), "ms_per_step": r.get("ms_per_step"),
                "phase_seconds": r.get("phase_seconds"),
                "config": {k: rc[k] for k in ("workload", "n_train", "n_query", "blocks", "query_passes", "parallelism") if k in rc},
                "roofline": {k: roof.get(k) for k in ("frac", "traffic_over_algorithmic", "mfma_util") if k in roof},
                "roofline_cov_frac": (r.get("roofline_cov") or {}).get("frac"),
                "roofline_lambda_update_frac": (r.get("roofline_lambda_update") or {}).get("frac"),
                "factor_fit": {"n_fit": (r.get("factor_fit") or {}).get("n_fit"), "seconds": (r.get("factor_fit") or {}).get("seconds")},
                "device_busy": _compact_busy(r.get("device_busy")),
            }
            if isinstance(parity, dict):
                short[name]["parity"] = ({"error": str(parity["error"])[:120]} if "error" in parity else
                                         {k: parity.get(k) for k in ("scores_rel_F_vs_fp64_low_rank_contraction", "bound", "ok", "queries", "train_samples")
                                          if k in parity})
            if r.get("exchanges") is not None:
                short[name]["exchanges"] = _compact_exchanges(r["exchanges"])
        line["other_configs"] = short
    line["extras_file"] = EXTRAS_FILE
    return _sig(line)


def render_line(full: dict) -> str:
    """``json.dumps(compact_line(full))``, checked: one line, parses back, under the hard cap (sections are dropped in order of
    dispensability if an unforeseen entry ever pushes it over -- the headline keys, ``roofline`` and ``cpu_baseline`` never are)."""
    line = compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    for dispensable in ("other_configs", "exchanges", "targets", "device_busy", "roofline_lambda_update", "roofline_lambda", "roofline_cov"):
        if len(text) <= LINE_HARD_CAP_BYTES:
            break
        line[dispensable] = {"moved_to": EXTRAS_FILE}
        text = json.dumps(line, separators=(",", ":"))
    assert "\n" not in text and len(text) <= LINE_HARD_CAP_BYTES, len(text)
    assert json.loads(text)["metric"] == full.get("metric")
    return text


def emit(full: dict) -> None:
    """Full object -> ``bench_extras.json`` (beside this script; best effort) and stderr; compact line -> the LAST stdout line."""
    blob = json.dumps(full, default=str)
    try:
        with open(os.path.join(ROOT, EXTRAS_FILE), "w", encoding="utf-8") as handle:
            handle.write(blob + "\n")
    except OSError as error:
        print(f"[bench] could not write {EXTRAS_FILE}: {error}", file=sys.stderr)
    print("[bench extras] " + blob, file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(render_line(full), flush=True)


def _respawn_under_torchrun(gpus: int) -> None:
    """``python bench.py --gpus N`` typed WITHOUT a launcher (no ``WORLD_SIZE``): re-execute this very command line as
    N ranks of one node under ``torch.distributed.run`` (one process per GPU, RCCL) and relay its output -- rank 0 of the
    child job prints the JSON line."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL / cross-process tensors on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(command, env=env))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("KF_BENCH_WORKLOAD", "resnet9"), choices=sorted(WORKLOADS))
    ap.add_argument("--n-train", type=int, default=None)
    ap.add_argument("--n-query", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip targets.mnist_mlp / other_configs in the default run")
    ap.add_argument("--factor-reps", type=int, default=1)
    ap.add_argument("--train-batch", type=int, default=None, help="override the workload's train batch size")
    ap.add_argument("--factor-batch", type=int, default=None, help="override the workload's factor-fit batch size")
    ap.add_argument("--query-batch", type=int, default=None, help="override the workload's per-device query batch size")
    ap.add_argument("--n-fit", type=int, default=None, help="fit the factors on the first N train samples only (default: all)")
    ap.add_argument("--warm-n-train", type=int, default=None, help="warm-up steps score against the first N train samples only")
    ap.add_argument("--blocks", type=int, default=None, help="llama_block: decoder blocks of the slice (default 1; other_configs uses 2)")
    ap.add_argument("--query-passes", type=int, default=None, help="groups the query batches are accumulated in, one train pass each "
                    "(default: as few as fit 62 %% of the device memory)")
    ap.add_argument("--busy-n-train", type=int, default=None, help="train samples of the two extra steps behind ``device_busy`` (default: all)")
    ap.add_argument("--channels-last", action="store_true", help="image workloads: model and images in torch.channels_last (NHWC) memory "
                    "format -- MIOpen's bf16 convolution kernels are NHWC kernels; with NCHW tensors it transposes around each of them")
    ap.add_argument("--miopen-batchnorm", action="store_true", help="resnet9: the model's eval-mode batch norm on MIOpen's kernel again "
                    "(default: ATen's; see NATIVE_BATCH_NORM)")
    ap.add_argument("--phase-split", action="store_true", help="also report the step's query-phase / train-pass wall seconds (one more "
                    "device synchronisation per held-query window inside the timed region)")
    ap.add_argument("--no-miopen-find", action="store_true", help="leave torch.backends.cudnn.benchmark off (default: on -- MIOpen "
                    "searches its convolution kernels for the MODEL's own forward / backward during warm-up; ResNet-9 stage "
                    "907 -> 862 ms; nothing of the EK-FAC path is affected)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn_under_torchrun(args.gpus)
    if args.train_batch:
        WORKLOADS[args.workload]["train_batch"] = args.train_batch
    if args.factor_batch:
        WORKLOADS[args.workload]["factor_batch"] = args.factor_batch
    if args.query_batch:
        WORKLOADS[args.workload]["query_batch"] = args.query_batch
    if args.blocks:
        WORKLOADS["llama_block"]["blocks"] = args.blocks
    if args.channels_last:
        WORKLOADS[args.workload]["channels_last"] = True
    if args.miopen_batchnorm:
        global NATIVE_BATCH_NORM
        NATIVE_BATCH_NORM = False
    if not args.no_miopen_find:
        torch.backends.cudnn.benchmark = True

    from kronfluence_amd.utils.state import State

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback).")
    state = State()
    world, rank = state.num_processes, state.process_index
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    line = run_workload(args.workload, state, args.n_train, args.n_query, args.steps, args.warmup, args.factor_reps,
                        cpu_baseline=not args.no_cpu_baseline, n_fit=args.n_fit, warm_n_train=args.warm_n_train,
                        query_passes=args.query_passes, busy_n_train=args.busy_n_train, phase_split=args.phase_split)
    default_run = (args.workload == "resnet9" and args.n_train is None and args.n_query is None and not args.no_extras
                   and os.environ.get("KF_BENCH_EXTRAS", "1") != "0")
    if default_run and world == 1:
        # north-star target: >= 10x reference-CPU pairs/s on MNIST-MLP at 1 GPU with scores within 1e-4 -- the CPU oracle
        # runs the FULL 100 x 1000 workload on the same factors, so the score error is part of the same line
        m = run_workload("mnist_mlp", state, None, None, steps=10, warmup=2, factor_reps=1, cpu_baseline=True)
        cpu_rate = m["cpu_baseline"]["value"]
        line["targets"] = {"mnist_mlp": {
            "gpu_pairs_per_sec": m["value"], "ms_per_step": m["ms_per_step"], "cpu_pairs_per_sec": cpu_rate,
            "cpu_cores": m["cpu_baseline"]["cores"], "ratio": m["value"] / cpu_rate, "target_ratio": 10.0,
            "scores_rel_F_vs_cpu_oracle": m["cpu_baseline"].get("gpu_vs_cpu_scores_rel_F"), "damping": 1e-8,
            "target_rel": 1e-4, "roofline": m["roofline"], "factor_fit": m["factor_fit"], "device_busy": m["device_busy"],
        }}
    if default_run:
        # N = 1: BERT-base and GPT-2-small at bounded sizes.  N > 1: GPT-2-small only -- the config the north-star scaling
        # target (>= 6x strong scaling 1 -> 8) is stated on -- sharded like the headline, same fixed size at every N.
        others = ("bert_base", "gpt2_small", "llama_block") if world == 1 else ("gpt2_small",)
        # BERT-base at its FULL 67 349 x 872 (configs[2]); GPT-2-small at 16 384 x 1 024 sequences of 512 tokens (the score
        # contraction dominates the stage from there on; the full 100 k x 2 k is the 8-GPU configuration).  The factors are
        # fitted on a bounded prefix (n_fit) and the warm-up step scores a small prefix, so the default run stays within minutes.
        sizes = {"bert_base": dict(n_train=WORKLOADS["bert_base"]["full_n_train"], n_fit=8192, warm_n_train=1024, busy_n_train=8192),
                 "gpt2_small": dict(n_train=16384, n_fit=2048, warm_n_train=512, busy_n_train=2048),
                 # configs[4] as a one-block slice at full width (C5 proper is 32 blocks x 100k x 1k on 8 GPUs): ONE cold factor
                 # fit -- its 40 s are three 14336^2 eigendecompositions
                 "llama_block": dict(n_train=64, n_fit=64, warm_n_train=16, factor_reps=0, blocks=2)}
        extras: Dict[str, dict] = {}
        for other in others:
            try:
                # factor_reps=1: the reported fit is the second, warm one (the first GPT-2 covariance pass alone spends ~5 s in
                # first-touch allocations and GEMM heuristics)
                size = sizes[other]
                if "blocks" in size:
                    WORKLOADS[other]["blocks"] = size["blocks"]
                r = run_workload(other, state, size["n_train"], None, steps=1, warmup=1, factor_reps=size.get("factor_reps", 1), cpu_baseline=False,
                                 n_fit=size["n_fit"], warm_n_train=size["warm_n_train"], busy_n_train=size.get("busy_n_train"), phase_split=True)
                if rank == 0:
                    extras[other] = {k: r[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "phase_seconds", "scaling", "config", "roofline",
                                                      "roofline_cov", "roofline_cov_f32", "roofline_lambda", "roofline_lambda_update", "factor_fit",
                                                      "exchanges", "peak_hbm_gib", "device_busy", "parity")}
            except Exception as error:  # an extra must never take the headline down with it
                extras[other] = {"error": f"{type(error).__name__}: {error}"[:300]}
                gc.collect()
                torch.cuda.empty_cache()
        if rank == 0:
            line["other_configs"] = extras
    if rank == 0:
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
