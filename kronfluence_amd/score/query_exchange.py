"""How the preconditioned query gradients reach every rank of a train-sharded pairwise stage.

The train set is sharded, so every rank needs every query's preconditioned gradient.  Two exact ways to get there:

``gather``     the reference's way (``module/tracker/precondition.py:166-214``, ``score/pairwise.py:239-246``): queries are
               sharded strided over the ranks, each rank preconditions ``Q / P`` of them, and every layer's ``[q, O, I']`` block
               (or low-rank factor pair) is all-gathered and interleaved back into dataset order.  Costs ``Q D s (P-1)/P`` bytes
               INBOUND per rank over xGMI, issued asynchronously from the backward hooks.
``replicate``  no query shard and no collective on the query side: every rank runs ALL query batches itself (model forward /
               backward + per-sample gradient + EK-FAC preconditioner), the train shard and the score-block gather stay as they
               are.  Costs ``(P-1)/P`` of the query phase in redundant flops per rank and zero bytes.

Scores are identical (same per-query arithmetic, same held-query windows, same train passes); which is FASTER is a bytes-versus-
flops question the plan below answers from the layer shapes -- like ``pairwise_score._low_rank_plan`` does for the contraction
order.  On xGMI (7 links, ~0.54 TB/s inbound per GPU at line rate) moving an element costs ~6 ps while preconditioning it costs
``4 (O + I')`` flops ~ 9 ps at the rate the query phase achieves, so ``gather`` wins for every BASELINE config; ``replicate`` wins
on slow transports (gloo, PCIe-only boxes) and is there to be forced (``KF_QUERY_EXCHANGE=replicate``) when the interconnect
misbehaves.  DESIGN.md section 6 carries the time model with the predicted 1/2/4/8 curves of both modes.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn

MODES = ("auto", "gather", "replicate")

# Inbound all-gather bandwidth per rank.  xGMI: 7 links x ~76.5 GB/s per direction = 0.54 TB/s at line rate; RCCL's
# all-gather reaches roughly two thirds of that on a fully connected 8-GPU node.  Anything that is not RCCL (gloo: host
# memory + loopback TCP) is priced at 2 GB/s.  ``KF_XGMI_GBPS`` overrides both.
XGMI_ALLGATHER_GBPS = 350.0
HOST_TRANSPORT_GBPS = 2.0
# Rate of the QUERY phase (model forward / backward of the query batch, per-sample gradient, preconditioner), as a fraction of the
# dense MFMA peak of the precondition dtype, measured on one MI355X (round 6, ``bench.py --phase-split``): GPT-2-small 1.52e12
# algorithmic flop per query in 2.23 ms = 0.27 of 2.5 PF, BERT-base 1.14e12 in 1.10 ms = 0.41; the lower one is used.  The
# exact-fp32 engine sustains about half of its 157 TF.
QUERY_PHASE_FRACTION = {torch.bfloat16: 0.27, torch.float16: 0.27, torch.float32: 0.5, torch.float64: 0.5}
PEAK_TFLOPS = {torch.bfloat16: 2500.0, torch.float16: 2500.0, torch.float32: 157.3, torch.float64: 78.6}


def requested_mode() -> str:
    """``KF_QUERY_EXCHANGE`` = auto (default) | gather | replicate.  An environment switch, not a ``ScoreArguments`` field: the
    arguments are written to ``score_arguments.json`` with exactly the reference's keys (arguments.py)."""
    mode = os.environ.get("KF_QUERY_EXCHANGE", "auto").strip().lower()
    if mode not in MODES:
        raise ValueError(f"KF_QUERY_EXCHANGE must be one of {MODES}, not {mode!r}.")
    return mode


@dataclass
class QueryExchangePlan:
    mode: str                       # "gather" | "replicate"
    reason: str
    world: int
    inbound_bytes: float            # gather: bytes every rank receives over the whole stage
    gather_exchange_seconds: float  # inbound_bytes / bandwidth (no overlap credited)
    gather_seconds: float           # Q / P query phases + the exchange
    replicate_seconds: float        # Q query phases, nothing exchanged
    flops_per_query: float
    bandwidth_gbps: float

    def to_dict(self) -> Dict[str, object]:
        return dict(self.__dict__)


def flops_per_query(shapes: Sequence[Tuple[int, int]], rows: Sequence[int]) -> float:
    """Algorithmic flops of ONE query through the query phase (the ``Q``-term of ``F_pair``, plus the model's own
    dgrad-only forward / backward over the tracked layers ``6 R O I'``): per layer ``[R>1] 2 R O I' + 4 O I' (I' + O) + 6 R O I'``."""
    total = 0.0
    for (o, ip), r in zip(shapes, rows):
        total += (2.0 * r * o * ip if r > 1 else 0.0) + 4.0 * o * ip * (ip + o) + 6.0 * r * o * ip
    return total


def held_elements_per_query(shapes: Sequence[Tuple[int, int]], low_rank: Optional[int]) -> float:
    """Elements one query contributes to the exchange: ``O I'`` per layer, or ``k (O + I')`` where rank-``k`` factor pairs are
    kept (``precondition.py:_store``: only layers with ``min(O, I') > k``)."""
    total = 0.0
    for o, ip in shapes:
        total += float(low_rank * (o + ip)) if (low_rank is not None and min(o, ip) > low_rank) else float(o * ip)
    return total


def plan_query_exchange(shapes: Sequence[Tuple[int, int]], rows: Sequence[int], n_query: int, world: int,
                        score_dtype: torch.dtype = torch.float32, precondition_dtype: torch.dtype = torch.float32,
                        low_rank: Optional[int] = None, backend: Optional[str] = None,
                        bandwidth_gbps: Optional[float] = None, mode: Optional[str] = None) -> QueryExchangePlan:
    """Bytes over the interconnect against redundant flops.  ``shapes``: ``(O, I')`` per tracked layer; ``rows``: rows per sample
    (tokens / output positions) per layer.  ``mode`` (default: ``requested_mode()``) other than "auto" is taken as it is; the
    estimates are filled in either way so a run can report them."""
    mode = mode or requested_mode()
    if bandwidth_gbps is None:
        env = os.environ.get("KF_XGMI_GBPS")
        bandwidth_gbps = float(env) if env else (XGMI_ALLGATHER_GBPS if backend in (None, "nccl", "rccl") else HOST_TRANSPORT_GBPS)
    element = 2 if score_dtype in (torch.bfloat16, torch.float16) else 4
    inbound = float(n_query) * held_elements_per_query(shapes, low_rank) * element * (world - 1) / max(world, 1)
    exchange_s = inbound / (bandwidth_gbps * 1e9)
    per_query = flops_per_query(shapes, rows)
    rate = PEAK_TFLOPS.get(precondition_dtype, 157.3) * 1e12 * QUERY_PHASE_FRACTION.get(precondition_dtype, 0.5)
    t_query = per_query / rate
    gather_s = n_query / max(world, 1) * t_query + exchange_s
    replicate_s = n_query * t_query
    if mode != "auto":
        chosen, why = mode, f"KF_QUERY_EXCHANGE={mode}"
    elif world <= 1:
        chosen, why = "gather", "one rank: nothing to exchange"
    elif replicate_s < gather_s:
        chosen, why = "replicate", (f"exchange {exchange_s:.3g} s at {bandwidth_gbps:g} GB/s exceeds the redundant query phases "
                                    f"({replicate_s - n_query / world * t_query:.3g} s)")
    else:
        chosen, why = "gather", (f"exchange {exchange_s:.3g} s at {bandwidth_gbps:g} GB/s is cheaper than {world - 1}/{world} of the "
                                 f"query phase ({replicate_s - n_query / world * t_query:.3g} s)")
    return QueryExchangePlan(chosen, why, world, inbound, exchange_s, gather_s, replicate_s, per_query, bandwidth_gbps)


def layer_shapes(model: nn.Module, names: Optional[Iterable[str]] = None) -> List[Tuple[int, int]]:
    from kronfluence_amd.module.tracked_module import TrackedModule

    wanted = None if names is None else set(names)
    out = []
    for m in model.modules():
        if isinstance(m, TrackedModule) and (wanted is None or m.name in wanted):
            w = m.original_module.weight
            out.append((int(w.shape[0]), int(w[0].numel()) + int(m.original_module.bias is not None)))
    return out


@torch.no_grad()
def probe_rows(model: nn.Module, measure, names: Optional[Iterable[str]] = None) -> List[int]:
    """Rows per sample ``R`` of every tracked layer (tokens of a Linear, output positions of a Conv2d), read off ONE forward
    pass: ``measure()`` runs the model on a small batch (no gradients); hooks on the wrapped modules divide the output's
    elements by ``batch x O``."""
    from kronfluence_amd.module.tracked_module import TrackedModule

    wanted = None if names is None else set(names)
    rows: Dict[str, int] = {}
    handles = []
    order = []
    for m in model.modules():
        if isinstance(m, TrackedModule) and (wanted is None or m.name in wanted):
            order.append(m.name)

            def hook(mod, inputs, output, m=m):
                o = int(m.original_module.weight.shape[0])
                rows[m.name] = max(1, int(output.numel() // max(1, output.shape[0] * o)))
            handles.append(m.original_module.register_forward_hook(hook))
    try:
        measure()
    finally:
        for h in handles:
            h.remove()
    return [rows.get(name, 1) for name in order]


def mark_replicated(loader, replicated: bool = True):
    """The stage loop (``score/pairwise.py``) reads this flag off the QUERY loader: a loader that yields ALL queries in dataset
    order on every rank (no distributed sampler).  Whoever builds the loader (``Analyzer``, ``bench.py``) sets it."""
    loader.kf_replicated_queries = bool(replicated)
    return loader


def is_replicated(loader) -> bool:
    return bool(getattr(loader, "kf_replicated_queries", False))


def backend_name() -> Optional[str]:
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
