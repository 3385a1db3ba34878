"""Inner loop of pairwise scoring: one pass over the (sharded) train set for the query gradients
currently held (reference ``score/dot_product.py:39-153``).

All tracked layers accumulate into one ``[Q, N_shard]`` fp32 buffer in HBM; nothing crosses PCIe
until the shard is done, then the per-rank blocks are gathered on rank 0 (C5).
"""

from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist
from torch import autocast, nn
from torch.utils import data

from kronfluence_amd.arguments import FactorArguments, ScoreArguments
from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
from kronfluence_amd.module.tracker.pairwise_score import ScoreSink
from kronfluence_amd.module.utils import (
    finalize_all_iterations, finalize_iteration, set_mode, set_score_sink, set_side_stream, synchronize_modules,
)
from kronfluence_amd.task import Task
from kronfluence_amd.utils.comm import exchange
from kronfluence_amd.utils.constants import (
    ACCUMULATED_PRECONDITIONED_GRADIENT_NAME, AGGREGATED_GRADIENT_NAME, ALL_MODULE_NAME, SCORE_TYPE,
)
from kronfluence_amd.utils.dataset import find_batch_size, send_to_device
from kronfluence_amd.utils.state import State, no_sync, to_host


def gather_score_blocks(block: torch.Tensor, state: State, dataset_size: int) -> torch.Tensor:
    """C5: rank 0 receives every rank's ``[Q, ceil(N/P)]`` block and concatenates them along the
    train axis, dropping the wrap-around padding (reference ``dot_product.py:141-150``)."""
    if not state.use_distributed:
        return to_host(block[:, :dataset_size])
    gather_list = [torch.empty_like(block) for _ in range(state.num_processes)] if state.is_main_process else None
    with exchange("score_gather", block.numel() * block.element_size()):
        dist.gather(block, gather_list, dst=0)
    if state.is_main_process:
        return to_host(torch.cat(gather_list, dim=1)[:, :dataset_size])
    return to_host(block)


# ---- the train-side step as a hipGraph ---------------------------------------------------------------------------------------
# A pairwise train pass launches the same ~250 kernels per batch (model forward / backward + the hooks' score kernels) from Python:
# on the small configs the GPU waits for the host between them (ResNet-9: 10 % of a step idle, bench.py ``device_busy``).  Shapes
# are static from the second batch on (the first converts the held queries to their k-tile-major layout, warms MIOpen / hipBLASLt
# and the allocator), the library never allocates or synchronises, so ONE batch -- forward, backward, every hook launch -- is
# captured (``torch.cuda.graph``: stream capture into a private pool) writing its scores into a static ``[Q, b]`` block, and replayed
# for each remaining full batch: inputs copied into the static batch, one ``hipGraphLaunch``, block added into the shard's columns.
# ``KF_TRAIN_GRAPH`` = 0 (default: never) | auto (when the pass has at least ``GRAPH_MIN_BATCHES`` equal batches and nothing stateful
# is in the way -- shared parameters, per-token scores, host offload, the side stream, micro-batch pairing) | 1 (whenever possible).
# A capture that fails falls back to the eager loop for the rest of the pass (``GRAPH_LOG``).  OPT-IN because it does not pay on this
# runtime (profiles/r06_train_graph_ab.log: ResNet-9 idle share 0.103 -> 0.058, step 787.6 -> 782.4 ms, +0.7 %: the idle share is
# thousands of sub-20-us gaps between dependent kernels, which a graph launch does not remove) and because kernels launched from a
# graph cannot be event-timed, which ``bench.py``'s per-launch roofline relies on.
GRAPH_MIN_BATCHES = 6
GRAPH_LOG: dict = {"captures": 0, "replays": 0, "fallbacks": 0, "last_error": None}


def _graph_mode() -> str:
    import os

    return os.environ.get("KF_TRAIN_GRAPH", "0").strip().lower()


def _tensors_of(batch):
    if torch.is_tensor(batch):
        return [batch]
    if isinstance(batch, dict):
        return [t for v in batch.values() for t in _tensors_of(v)]
    if isinstance(batch, (list, tuple)):
        return [t for v in batch for t in _tensors_of(v)]
    return []


def _same_layout(a, b) -> bool:
    ta, tb = _tensors_of(a), _tensors_of(b)
    return len(ta) == len(tb) and all(x.shape == y.shape and x.dtype == y.dtype and x.device == y.device for x, y in zip(ta, tb))


def _clone_batch(batch):
    if torch.is_tensor(batch):
        return batch.clone()
    if isinstance(batch, dict):
        return {k: _clone_batch(v) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(_clone_batch(v) for v in batch)
    return batch


def _graph_eligible(state: State, factor_args: FactorArguments, score_args: ScoreArguments, num_batches: int, batch) -> bool:
    import os

    mode = _graph_mode()
    if mode in ("0", "off", "false") or state.device.type != "cuda":
        return False
    if (factor_args.has_shared_parameters or score_args.compute_per_token_scores or score_args.offload_activations_to_cpu
            or os.environ.get("KF_SIDE_STREAM", "0") != "0"):
        return False
    tensors = _tensors_of(batch)
    if not tensors or not all(t.is_cuda for t in tensors):
        return False
    return num_batches >= (3 if mode == "1" else GRAPH_MIN_BATCHES)


def compute_dot_products_with_loader(model: nn.Module, task: Task, state: State, train_loader: data.DataLoader,
                                     factor_args: FactorArguments, score_args: ScoreArguments,
                                     tracked_module_names: List[str], loss_scale: float = 1.0,
                                     disable_tqdm: bool = False) -> SCORE_TYPE:
    del disable_tqdm
    model.zero_grad(set_to_none=True)
    set_mode(model, ModuleMode.PAIRWISE_SCORE, tracked_module_names, release_memory=False)
    modules = [m for m in model.modules() if isinstance(m, TrackedModule) and m.name in tracked_module_names]
    held = modules[0].storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME]
    num_queries = (held[0] if isinstance(held, list) else held).shape[0]
    shard_size = len(train_loader.sampler) if hasattr(train_loader, "sampler") else len(train_loader.dataset)
    dataset_size = len(train_loader.dataset)

    keys = [m.name for m in modules] if score_args.compute_per_module_scores else [ALL_MODULE_NAME]
    sinks: Dict[str, ScoreSink] = {
        key: ScoreSink(num_queries, shard_size, state.device, per_token=score_args.compute_per_token_scores) for key in keys
    }
    enable_amp = score_args.amp_dtype is not None
    offset = 0

    def key_of(m) -> str:
        return m.name if score_args.compute_per_module_scores else ALL_MODULE_NAME

    def run_batch(batch, into, at: int) -> None:
        """Forward + backward of one train batch; every tracked layer adds its scores into ``into[key]`` at column ``at``."""
        for m in modules:
            m.score_sink = (into[key_of(m)], at)
        with no_sync(model, state):
            model.zero_grad(set_to_none=True)
            with autocast(device_type=state.device.type, enabled=enable_amp, dtype=score_args.amp_dtype):
                loss = task.compute_train_loss(batch=batch, model=model, sample=False)
            (loss * loss_scale if loss_scale != 1.0 else loss).backward()
        if factor_args.has_shared_parameters:
            finalize_iteration(model, tracked_module_names)

    graph = static = blocks = None
    graph_state = "warm"   # -> "replay" once captured, "off" when not eligible or the capture failed
    num_batches = len(train_loader)
    set_side_stream(model, tracked_module_names, True)   # the score kernels may run beside the model's backward pass
    try:
        for index, batch in enumerate(train_loader):
            batch = send_to_device(batch, state.device)
            size = find_batch_size(batch)
            if graph_state == "warm" and index == 1:
                holding = any(getattr(m._trackers.get(ModuleMode.PAIRWISE_SCORE), "_pair_held", None) is not None for m in modules)
                if holding or not _graph_eligible(state, factor_args, score_args, num_batches, batch):
                    graph_state = "off"
                else:
                    try:
                        static = _clone_batch(batch)
                        blocks = {key: ScoreSink(num_queries, size, state.device) for key in keys}
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            for block in blocks.values():
                                block.matrix(1).zero_()
                            run_batch(static, blocks, 0)
                        graph_state = "replay"
                        GRAPH_LOG["captures"] += 1
                    except Exception as error:  # noqa: BLE001 -- any capture failure: eager loop for the rest of the pass
                        graph = static = blocks = None
                        graph_state = "off"
                        GRAPH_LOG["fallbacks"] += 1
                        GRAPH_LOG["last_error"] = f"{type(error).__name__}: {error}"[:300]
                        for m in modules:
                            m.finalize_iteration()
                        torch.cuda.synchronize()
            if graph_state == "replay" and _same_layout(batch, static):
                for dst, src in zip(_tensors_of(static), _tensors_of(batch)):
                    dst.copy_(src)
                graph.replay()
                for key, block in blocks.items():
                    sinks[key].matrix(1)[:, offset:offset + size].add_(block.matrix(1))
                GRAPH_LOG["replays"] += 1
            else:
                run_batch(batch, sinks, offset)
            offset += size
        del graph, static, blocks
        model.zero_grad(set_to_none=True)
        set_score_sink(model, None, tracked_module_names)
        finalize_all_iterations(model, tracked_module_names)   # flushes held micro-batches, joins the side stream
    finally:
        set_side_stream(model, tracked_module_names, False)
    set_mode(model, ModuleMode.PRECONDITION_GRADIENT, tracked_module_names, release_memory=False)

    total_scores: SCORE_TYPE = {}
    for key, sink in sinks.items():
        total_scores[key] = gather_score_blocks(sink.result().to(score_args.score_dtype), state, dataset_size)
    state.wait_for_everyone()
    return total_scores


def compute_aggregated_dot_products_with_loader(model: nn.Module, task: Task, state: State, train_loader: data.DataLoader,
                                                factor_args: FactorArguments, score_args: ScoreArguments,
                                                tracked_module_names: List[str], loss_scale: float = 1.0,
                                                disable_tqdm: bool = False) -> SCORE_TYPE:
    """``aggregate_train_gradients`` (reference ``score/dot_product.py:156-290``): the train pass only SUMS the
    gradients (one GEMM per layer and batch, ``GradientTracker``), the ranks all-reduce the sums, and the scores are
    one ``[Q, O I'] x [O I']`` product per layer -> ``[Q, 1]``."""
    del disable_tqdm
    model.zero_grad(set_to_none=True)
    set_mode(model, ModuleMode.GRADIENT_AGGREGATION, tracked_module_names, release_memory=False)
    modules = [m for m in model.modules() if isinstance(m, TrackedModule) and m.name in tracked_module_names]
    held = modules[0].storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME]
    num_queries = (held[0] if isinstance(held, list) else held).shape[0]
    enable_amp = score_args.amp_dtype is not None
    if not all(m.exist() for m in modules):  # the summed train gradient is reused across query chunks
        for batch in train_loader:
            batch = send_to_device(batch, state.device)
            with no_sync(model, state):
                model.zero_grad(set_to_none=True)
                with autocast(device_type=state.device.type, enabled=enable_amp, dtype=score_args.amp_dtype):
                    loss = task.compute_train_loss(batch=batch, model=model, sample=False)
                (loss * loss_scale if loss_scale != 1.0 else loss).backward()
            if factor_args.has_shared_parameters:
                finalize_iteration(model, tracked_module_names)
            del loss
        if state.use_distributed:
            synchronize_modules(model, tracked_module_names, num_processes=state.num_processes)
    set_mode(model, ModuleMode.PAIRWISE_SCORE, tracked_module_names, release_memory=False)
    keys = [m.name for m in modules] if score_args.compute_per_module_scores else [ALL_MODULE_NAME]
    sinks = {key: ScoreSink(num_queries, 1, state.device) for key in keys}
    for m in modules:
        m.score_sink = (sinks[m.name if score_args.compute_per_module_scores else ALL_MODULE_NAME], 0)
    held = {m.name: m.storage[AGGREGATED_GRADIENT_NAME] for m in modules}
    finalize_all_iterations(model, tracked_module_names)  # scores from the summed gradient (PairwiseScoreTracker)
    for m in modules:
        m.storage[AGGREGATED_GRADIENT_NAME] = held[m.name]  # kept for the next query chunk
    model.zero_grad(set_to_none=True)
    set_score_sink(model, None, tracked_module_names)
    set_mode(model, ModuleMode.PRECONDITION_GRADIENT, tracked_module_names, release_memory=False)
    total_scores: SCORE_TYPE = {key: to_host(sink.result().to(score_args.score_dtype)) for key, sink in sinks.items()}
    state.wait_for_everyone()
    return total_scores
