"""Small offline fixtures shared by the golden generator, the oracle tests and the GPU parity tests.

Three models cover the three layer flavours of the hot path:
``mlp`` (2-D activations, one layer without bias), ``conv`` (Conv2d incl. stride, padding, a grouped
layer, bias / no-bias, followed by a Linear) and ``seq`` (``[b, T, d]`` Linear stack with a padding
mask).  Every fixture has more fitted rows than factor dimensions so the covariances are full rank
(or have a simple null vector) and the eigenbasis -- hence Lambda and the scores -- is well defined
up to sign.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

Batch = Tuple[torch.Tensor, ...]


class _SeqModel(nn.Module):
    def __init__(self, vocab: int = 20, width: int = 8) -> None:
        super().__init__()
        self.embed = nn.Embedding(vocab, width)
        self.fc1 = nn.Linear(width, 12)
        self.fc2 = nn.Linear(12, width, bias=False)
        self.head = nn.Linear(width, vocab)

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        h = self.embed(ids)
        h = h + self.fc2(torch.tanh(self.fc1(h)))
        return self.head(h)


class _SharedMLP(nn.Module):
    """One Linear applied three times per forward: exercises ``has_shared_parameters`` (per-sample gradients of
    the three uses are summed before Lambda / preconditioning / scoring)."""

    def __init__(self) -> None:
        super().__init__()
        self.first = nn.Linear(12, 16)
        self.shared = nn.Linear(16, 16)
        self.last = nn.Linear(16, 3, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = torch.relu(self.first(x))
        h = torch.relu(self.shared(h))
        h = torch.tanh(self.shared(h))
        h = torch.relu(self.shared(h))
        return self.last(h)


def is_regression(kind: str) -> bool:
    """``*_mse`` fixtures: the same three layer flavours with smooth activations, a summed squared-error loss and many
    more fitted rows than factor dimensions.  Cross-entropy gives the last layer's gradient covariance an exact null
    vector (softmax gradients sum to zero) and ReLU units that never fire give exact zero rows, so at the reference's
    DEFAULT damping 1e-8 ``1 / (Lambda / n + 1e-8)`` amplifies fp32 round-off by 1e8 there and even the reference
    disagrees with itself (fp32 vs fp64) at the 1e-2 level.  These fixtures are well conditioned, which makes the
    default damping testable at the north-star bound of 1e-4."""
    return kind.endswith("_mse")


def _act(kind: str) -> nn.Module:
    return nn.Tanh() if is_regression(kind) else nn.ReLU()


def make_model(kind: str, seed: int = 0) -> nn.Module:
    torch.manual_seed(seed)
    if kind == "mlp_mse":
        return nn.Sequential(nn.Linear(12, 16), nn.Tanh(), nn.Linear(16, 16, bias=False), nn.Tanh(), nn.Linear(16, 3))
    if kind == "conv_mse":
        return nn.Sequential(
            nn.Conv2d(3, 4, 3, padding=1, bias=False), nn.Tanh(),
            nn.Conv2d(4, 8, 5, stride=2, padding=2, bias=True), nn.Tanh(),
            nn.Conv2d(8, 6, 3, padding=1, groups=2, bias=True), nn.Tanh(),
            nn.Flatten(), nn.Linear(6 * 4 * 4, 3),
        )
    if kind == "seq_mse":
        return _SeqModel()
    if kind == "shared":
        return _SharedMLP()
    if kind == "mlp":
        return nn.Sequential(
            nn.Linear(12, 16), nn.ReLU(), nn.Linear(16, 16, bias=False), nn.ReLU(), nn.Linear(16, 3)
        )
    if kind == "conv":
        return nn.Sequential(
            nn.Conv2d(3, 4, 3, padding=1, bias=False), nn.ReLU(),
            nn.Conv2d(4, 8, 5, stride=2, padding=2, bias=True), nn.ReLU(),
            nn.Conv2d(8, 6, 3, padding=1, groups=2, bias=True), nn.ReLU(),
            nn.Flatten(), nn.Linear(6 * 4 * 4, 3),
        )
    if kind == "conv8":
        # channel counts / patch sizes that are multiples of 8: eligible for the bf16 MFMA engines
        return nn.Sequential(
            nn.Conv2d(8, 16, 3, padding=1, bias=False), nn.ReLU(),
            nn.Conv2d(16, 16, 3, stride=2, padding=1, bias=False), nn.ReLU(),
            nn.Conv2d(16, 8, 3, padding=1, bias=False), nn.ReLU(),
            nn.Flatten(), nn.Linear(8 * 4 * 4, 8, bias=False),
        )
    if kind == "seq":
        return _SeqModel()
    raise KeyError(kind)


def make_data(kind: str, n: int, seed: int) -> Batch:
    gen = torch.Generator().manual_seed(seed)
    if kind == "mlp_mse":
        return (torch.randn(n, 12, generator=gen), torch.randn(n, 3, generator=gen))
    if kind == "conv_mse":
        return (torch.randn(n, 3, 8, 8, generator=gen), torch.randn(n, 3, generator=gen))
    if kind == "seq_mse":
        t = 6
        ids = torch.randint(0, 20, (n, t), generator=gen)
        lengths = torch.randint(2, t + 1, (n,), generator=gen)
        mask = (torch.arange(t)[None, :] < lengths[:, None]).to(torch.int64)
        return (ids, mask, torch.randn(n, t, 20, generator=gen))
    if kind in ("mlp", "shared"):
        return (torch.randn(n, 12, generator=gen), torch.randint(0, 3, (n,), generator=gen))
    if kind == "conv":
        return (torch.randn(n, 3, 8, 8, generator=gen), torch.randint(0, 3, (n,), generator=gen))
    if kind == "conv8":
        return (torch.randn(n, 8, 8, 8, generator=gen), torch.randint(0, 8, (n,), generator=gen))
    if kind == "seq":
        t = 6
        ids = torch.randint(0, 20, (n, t), generator=gen)
        lengths = torch.randint(2, t + 1, (n,), generator=gen)
        mask = (torch.arange(t)[None, :] < lengths[:, None]).to(torch.int64)
        labels = torch.randint(0, 20, (n, t), generator=gen)
        labels = torch.where(mask.bool(), labels, torch.full_like(labels, -100))
        return (ids, mask, labels)
    raise KeyError(kind)


def _inputs_to(model: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if x.is_floating_point():
        return x.to(dtype=next(model.parameters()).dtype)
    return x


def train_loss(kind: str) -> Callable[[nn.Module, Batch], torch.Tensor]:
    """Summed cross-entropy with the true labels (empirical Fisher; deterministic)."""

    def loss(model: nn.Module, batch: Batch) -> torch.Tensor:
        if kind == "seq_mse":  # padded tokens contribute no loss, hence no gradient (the reference relies on that)
            ids, mask, target = batch
            out = model(ids)
            return 0.5 * (((out - target.to(out.dtype)) ** 2).sum(-1) * mask.to(out.dtype)).sum()
        if is_regression(kind):
            x, target = batch
            out = model(_inputs_to(model, x))
            return 0.5 * ((out - target.to(out.dtype)) ** 2).sum()
        if kind == "seq":
            ids, _mask, labels = batch
            logits = model(ids)
            return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1),
                                   reduction="sum", ignore_index=-100)
        x, y = batch
        return F.cross_entropy(model(_inputs_to(model, x)), y, reduction="sum")

    return loss


def measurement(kind: str) -> Callable[[nn.Module, Batch], torch.Tensor]:
    """A measurement that differs from the loss: summed correct-class margin."""

    def measure(model: nn.Module, batch: Batch) -> torch.Tensor:
        if kind == "seq_mse":
            ids, mask, target = batch
            out = model(ids)
            return ((out * target.to(out.dtype)).sum(-1) * mask.to(out.dtype)).sum()
        if is_regression(kind):
            x, target = batch
            out = model(_inputs_to(model, x))
            return (out * target.to(out.dtype)).sum()
        if kind == "seq":
            ids, mask, labels = batch
            logits = model(ids)
            safe = labels.clamp(min=0)
            picked = logits.gather(-1, safe[..., None])[..., 0]
            margins = picked - torch.logsumexp(logits, dim=-1) * 0.5
            return (margins * mask.to(margins.dtype)).sum()
        x, y = batch
        logits = model(_inputs_to(model, x))
        picked = logits.gather(-1, y[:, None])[:, 0]
        return (picked - 0.5 * torch.logsumexp(logits, dim=-1)).sum()

    return measure


def attention_mask(kind: str) -> Optional[Callable[[Batch], Optional[torch.Tensor]]]:
    if kind in ("seq", "seq_mse"):
        return lambda batch: batch[1]
    return None


def batches(data: Batch, batch_size: int) -> List[Batch]:
    n = data[0].shape[0]
    return [tuple(t[i:i + batch_size] for t in data) for i in range(0, n, batch_size)]


@dataclass
class Fixture:
    kind: str
    n_train: int
    n_query: int
    factor_batch: int
    train_batch: int
    query_batch: int


FIXTURES: Dict[str, Fixture] = {
    "mlp": Fixture("mlp", 48, 6, 16, 12, 3),
    "conv": Fixture("conv", 40, 6, 8, 10, 3),
    "seq": Fixture("seq", 48, 6, 16, 12, 3),
}
# well-conditioned regression fixtures for the default damping 1e-8 (see ``is_regression``)
MSE_FIXTURES: Dict[str, Fixture] = {
    "mlp_mse": Fixture("mlp_mse", 256, 6, 64, 64, 3),
    "conv_mse": Fixture("conv_mse", 256, 6, 32, 32, 3),
    "seq_mse": Fixture("seq_mse", 192, 6, 48, 48, 3),
}
# bf16-engine fixture: not part of FIXTURES (the generic parametrised tests run fp32); see test_pipeline_gpu.py
BF16_FIXTURE = Fixture("conv8", 256, 8, 64, 64, 4)
# shared-parameter fixture (FactorArguments.has_shared_parameters=True)
SHARED_FIXTURE = Fixture("shared", 48, 6, 16, 12, 3)


def spec_of(kind: str) -> Fixture:
    if kind in FIXTURES:
        return FIXTURES[kind]
    if kind in MSE_FIXTURES:
        return MSE_FIXTURES[kind]
    return {"conv8": BF16_FIXTURE, "shared": SHARED_FIXTURE}[kind]
