"""Host-side logic that needs no GPU: arguments, samplers, partitions, wrapper installation, mode
switching, storage keys.  Mirrors the behaviour the reference's CPU tests pin (SURVEY.md section 4:
tests/test_dataset_utils.py, tests/modules/test_modules.py, arguments validation)."""

import pytest
import torch
from torch import nn

import fixtures as fx
from kronfluence_amd import FactorArguments, ScoreArguments, Task, prepare_model
from kronfluence_amd.arguments import unsupported_score_options
from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
from kronfluence_amd.module.utils import (
    get_tracked_module_names, make_modules_partition, set_mode, wrap_tracked_modules,
)
from kronfluence_amd.utils import constants as C
from kronfluence_amd.utils.dataset import (
    DistributedEvalSampler, DistributedSamplerWithStack, ResidentLoader, find_batch_size, make_indices_partition,
    send_to_device,
)
from kronfluence_amd.utils.exceptions import IllegalTaskConfigurationError


class _Task(Task):
    def __init__(self, names=None):
        self.names = names

    def compute_train_loss(self, batch, model, sample=False):
        return model(batch[0]).sum()

    def compute_measurement(self, batch, model):
        return model(batch[0]).sum()

    def get_influence_tracked_modules(self):
        return self.names


def test_argument_validation_matches_reference_rules():
    with pytest.raises(ValueError):
        FactorArguments(covariance_max_examples=0)
    with pytest.raises(ValueError):
        FactorArguments(lambda_data_partitions=0)
    with pytest.raises(ValueError):
        ScoreArguments(damping_factor=-1.0)
    with pytest.raises(ValueError):
        ScoreArguments(query_gradient_accumulation_steps=0)
    with pytest.raises(ValueError):
        ScoreArguments(query_gradient_low_rank=0)
    args = FactorArguments()
    assert args.strategy == "ekfac" and args.eigendecomposition_dtype == torch.float64
    assert args.to_dict()["lambda_dtype"] == "torch.float32"
    assert ScoreArguments().damping_factor == 1e-8
    assert unsupported_score_options(ScoreArguments(query_gradient_low_rank=8)) == {}
    assert unsupported_score_options(ScoreArguments(query_gradient_low_rank=128)) == {}   # ranks above 88: round 4
    assert unsupported_score_options(ScoreArguments()) == {}


def test_storage_keys_and_file_names_are_the_references():
    assert C.COVARIANCE_FACTOR_NAMES == ["activation_covariance", "gradient_covariance",
                                         "num_activation_covariance_processed", "num_gradient_covariance_processed"]
    assert C.EIGENDECOMPOSITION_FACTOR_NAMES == ["activation_eigenvectors", "activation_eigenvalues",
                                                 "gradient_eigenvectors", "gradient_eigenvalues"]
    assert C.LAMBDA_FACTOR_NAMES == ["lambda_matrix", "num_lambda_processed"]
    assert C.ALL_MODULE_NAME == "all_modules" and C.FACTOR_SAVE_PREFIX == "factors_" and C.SCORE_SAVE_PREFIX == "scores_"
    assert C.HEURISTIC_DAMPING_SCALE == 0.1 and C.LAMBDA_DTYPE == torch.float64


@pytest.mark.parametrize("n,world", [(10, 3), (7, 2), (5, 8), (64, 4), (1, 2)])
def test_samplers_shard_like_the_reference(n, world):
    dataset = list(range(n))
    strided = [list(DistributedEvalSampler(dataset, world, r)) for r in range(world)]
    assert sorted(sum(strided, [])) == dataset  # disjoint, complete, no padding
    assert all(s == list(range(r, n, world)) for r, s in enumerate(strided))
    stacked = [list(DistributedSamplerWithStack(dataset, world, r)) for r in range(world)]
    chunk = -(-n // world)
    assert all(len(s) == chunk for s in stacked)
    assert sum(stacked, [])[:n] == dataset  # contiguous blocks concatenate in dataset order
    with pytest.raises(ValueError):
        DistributedEvalSampler(dataset, world, world)


def test_partitions_and_batch_size_helpers():
    import collections

    import numpy as np

    # the reference's split (np.array_split / divmod slices): remainder spread over the LEADING partitions
    assert make_indices_partition(10, 3) == [(0, 4), (4, 7), (7, 10)]
    for total, parts in ((199, 100), (7, 4), (10, 10), (67349, 7), (5, 1)):
        sizes = [len(chunk) for chunk in np.array_split(range(total), parts)]
        got = make_indices_partition(total, parts)
        assert [e - s for s, e in got] == sizes and got[0][0] == 0 and got[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        names = [str(i) for i in range(total)] if total < 1000 else None
        if names is not None:
            groups = make_modules_partition(names, parts)
            assert [len(g) for g in groups] == sizes and sum(groups, []) == names
    with pytest.raises(ValueError):
        make_indices_partition(2, 3)
    assert make_modules_partition(["a", "b", "c", "d", "e"], 2) == [["a", "b", "c"], ["d", "e"]]
    assert make_modules_partition(list("abcdefg"), 4) == [["a", "b"], ["c", "d"], ["e", "f"], ["g"]]
    assert find_batch_size({"x": torch.zeros(4, 2)}) == 4 and find_batch_size([torch.zeros(3), 1]) == 3
    # Mapping that is not a dict (HF BatchEncoding is a UserDict), namedtuples, objects with .to
    encoding = collections.UserDict(input_ids=torch.zeros(5, 3, dtype=torch.int64), attention_mask=torch.ones(5, 3))
    assert find_batch_size(encoding) == 5
    moved = send_to_device(encoding, torch.device("cpu"))
    assert set(moved.keys()) == {"input_ids", "attention_mask"} and moved["input_ids"].shape == (5, 3)
    Pair = collections.namedtuple("Pair", ["x", "y"])
    moved = send_to_device(Pair(torch.zeros(2), torch.ones(2)), torch.device("cpu"))
    assert isinstance(moved, Pair) and moved.y.tolist() == [1.0, 1.0] and find_batch_size(moved) == 2
    with pytest.raises(TypeError):
        find_batch_size({"meta": "no tensors here"})
    loader = ResidentLoader((torch.arange(10), torch.arange(10) * 2), 4, indices=[9, 1, 3])
    batches = list(loader)
    assert len(loader) == 1 and len(loader.dataset) == 10 and len(loader.sampler) == 3
    assert batches[0][0].tolist() == [9, 1, 3] and batches[0][1].tolist() == [18, 2, 6]


def test_prepare_model_wraps_supported_leaves_and_freezes():
    model = prepare_model(fx.make_model("conv"), _Task())
    names = get_tracked_module_names(model)
    assert names == ["0", "2", "4", "7"]
    assert all(not p.requires_grad for n, p in model.named_parameters() if "_constant" not in n)
    assert not model.training
    kinds = [type(m).__name__ for m in model.modules() if isinstance(m, TrackedModule)]
    assert kinds == ["TrackedConv2d", "TrackedConv2d", "TrackedConv2d", "TrackedLinear"]
    # wrapper is transparent in DEFAULT mode and keeps autograd alive through frozen weights
    x = torch.randn(2, 3, 8, 8)
    out = model(x)
    assert out.requires_grad
    assert torch.allclose(out, fx.make_model("conv").eval()(x), atol=1e-6)


def test_task_selected_modules_and_errors():
    model = prepare_model(fx.make_model("mlp"), _Task(names=["0", "4"]))
    assert get_tracked_module_names(model) == ["0", "4"]
    with pytest.raises(IllegalTaskConfigurationError):
        prepare_model(fx.make_model("mlp"), _Task(names=["0", "nope"]))
    with pytest.raises(IllegalTaskConfigurationError):
        prepare_model(nn.Sequential(nn.ReLU()), _Task())
    with pytest.raises(ValueError):
        wrap_tracked_modules(nn.DataParallel(fx.make_model("mlp")), _Task())


def test_mode_switch_registers_and_releases_hooks():
    model = prepare_model(fx.make_model("mlp"), _Task())
    mods = [m for m in model.modules() if isinstance(m, TrackedModule)]
    assert all(len(m._forward_hooks) == 0 for m in mods)
    set_mode(model, ModuleMode.COVARIANCE)
    assert all(len(m._forward_hooks) == 1 and m.current_mode == "covariance" for m in mods)
    set_mode(model, ModuleMode.LAMBDA, release_memory=True)
    assert all(len(m._forward_hooks) == 1 and m.current_mode == "lambda" for m in mods)
    set_mode(model, ModuleMode.DEFAULT)
    assert all(len(m._forward_hooks) == 0 for m in mods)
    for mode in (ModuleMode.SELF_SCORE, ModuleMode.SELF_MEASUREMENT_SCORE, ModuleMode.GRADIENT_AGGREGATION,
                 ModuleMode.PRECONDITION_GRADIENT, ModuleMode.PAIRWISE_SCORE):
        set_mode(model, mode)
        assert all(len(m._forward_hooks) == 1 and m.current_mode == mode for m in mods)
    set_mode(model, ModuleMode.DEFAULT)
    for m in mods:  # storage holds exactly the reference's keys
        assert set(m.storage) >= set(C.COVARIANCE_FACTOR_NAMES + C.EIGENDECOMPOSITION_FACTOR_NAMES + C.LAMBDA_FACTOR_NAMES)


def test_hooks_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU box")
    from kronfluence_amd._native import KfError

    model = prepare_model(fx.make_model("mlp"), _Task())
    set_mode(model, ModuleMode.COVARIANCE)
    with pytest.raises(KfError):
        model(torch.randn(4, 12)).sum().backward()


def test_argument_defaults_and_presets_equal_the_references():
    """tests/golden/presets.json holds the field values of the reference's dataclass defaults and of every preset in
    ``kronfluence/utils/common`` (generated by tests/golden/make_golden.py presets)."""
    import dataclasses
    import json
    import os

    from kronfluence_amd.utils.common import factor_arguments as mf, score_arguments as ms

    with open(os.path.join(os.path.dirname(__file__), "golden", "presets.json"), encoding="utf-8") as handle:
        gold = json.load(handle)

    def plain(obj):
        return {k: (str(v) if isinstance(v, torch.dtype) else v) for k, v in dataclasses.asdict(obj).items()}

    assert plain(FactorArguments()) == gold.pop("FactorArguments()")
    assert plain(ScoreArguments()) == gold.pop("ScoreArguments()")
    assert len(gold) >= 19
    for key, want in gold.items():
        tag, call = key.split("/", 1)
        module = mf if tag == "factor" else ms
        got = eval(f"module.{call}", {"module": module})  # noqa: S307 -- keys are our own fixture's function calls
        assert plain(got) == want, key


def test_low_rank_plan_picks_the_cheaper_exact_order():
    """``PairwiseScoreTracker._low_rank_plan``: the reference lets opt_einsum choose the contraction order of
    "qik,qko,b...i,b...o->qb" per call (module/linear.py:83-99); here the choice is a bytes + flops estimate of the same two exact
    orders.  Shapes only -- no arithmetic, so it runs on the CPU."""
    from types import SimpleNamespace as T

    from kronfluence_amd.module.tracker.pairwise_score import PairwiseScoreTracker

    tracker = PairwiseScoreTracker.__new__(PairwiseScoreTracker)

    def plan(q, o, i, k, b, r, ones, cuda=True, dtype=torch.bfloat16, factor_dtype=None):
        ip = i + int(ones)
        fd = factor_dtype or dtype
        return tracker._low_rank_plan(T(shape=(q, o, k), dtype=fd), T(shape=(q, k, ip), dtype=fd), T(shape=(b, r, o), is_cuda=cuda, dtype=dtype),
                                      T(shape=(b, r, i), dtype=dtype), ones)

    assert plan(100, 1024, 1024, 32, 250, 1, True) == "factored"          # one row per sample: always
    assert plan(872, 768, 768, 64, 512, 128, True) == "expand"             # BERT: narrow layer, large batch, expansion cached
    assert plan(1024, 768, 3072, 64, 128, 512, True) == "expand"           # GPT-2: 2.4 M-element blocks against 128 x 512 rows
    assert plan(1000, 14336, 4096, 64, 16, 512, False) == "factored"       # Llama-3-8B up projection, 16 sequences
    assert plan(1000, 4096, 14336, 64, 8, 512, False) == "factored"        # ... down projection
    assert plan(1000, 14336, 4096, 64, 256, 512, False) == "expand"        # the same layer against 256 sequences: flops win
    assert plan(1000, 14340, 4096, 64, 16, 512, False) == "expand"         # O not a multiple of 8: the GEMM path does not apply
    assert plan(1000, 14336, 4096, 64, 16, 512, False, cuda=False) == "expand"
    # the sequence form holds its two row products in bf16: only for operands that already are bf16
    assert plan(1000, 14336, 4096, 64, 16, 512, False, dtype=torch.float32) == "expand"
    assert plan(1000, 14336, 4096, 64, 16, 512, False, factor_dtype=torch.float32) == "expand"
    assert plan(100, 1024, 1024, 32, 250, 1, True, dtype=torch.float32) == "factored"   # one row per sample: the fp32 form



def test_train_micro_batches_pair_up_flush_and_guard(monkeypatch):
    """``PairwiseScoreTracker._score_rows_paired`` (host logic only; the kernels are replaced by a recorder): a small train
    micro-batch of a sequence layer is held until the same layer's next batch and both go through ONE call; an odd batch, a
    batch that does not continue the held one's score columns and a large batch are scored alone; an in-place change of a held
    tensor is an error; the byte account returns to zero."""
    from types import SimpleNamespace as T

    from kronfluence_amd import ops
    from kronfluence_amd.module.tracker import pairwise_score as ps

    calls = []

    def record(scores, offset, tiled, g, a, ones, scale=1.0, second=None):
        calls.append((offset, g.shape[0], None if second is None else second[0].shape[0]))

    monkeypatch.setattr(ops, "pairwise_score_rows", record)
    tracker = ps.PairwiseScoreTracker.__new__(ps.PairwiseScoreTracker)
    tracker.module = T(score_sink=(object(), 0), name="layer", storage={})
    monkeypatch.setattr(tracker, "_pair_budget", lambda device: 1 << 30)
    account = ps.PairwiseScoreTracker._pair_bytes_all_layers
    assert account[0] == 0
    def queries(q):
        held = ps.TiledQueries.__new__(ps.TiledQueries)
        held.num_queries, held.rows, held.width = q, 64, 72
        return held

    tiled = queries(1024)
    scores = torch.zeros(1024, 1000)

    def batch(n, seed):
        gen = torch.Generator().manual_seed(seed)
        return torch.randn(n, 64, 64, generator=gen).bfloat16(), torch.randn(n, 64, 64, generator=gen).bfloat16()

    g0, a0 = batch(128, 0)
    g1, a1 = batch(128, 1)
    g2, a2 = batch(96, 2)
    tracker._score_rows_paired(scores, 0, tiled, g0, a0, True, 1.0)
    assert calls == [] and account[0] == (g0.numel() + a0.numel()) * 2          # held, nothing launched
    tracker._score_rows_paired(scores, 128, tiled, g1, a1, True, 1.0)
    assert calls == [(0, 128, 128)] and account[0] == 0 and tracker._pair_held is None   # one call for both
    tracker._score_rows_paired(scores, 256, tiled, g2, a2, True, 1.0)              # odd batch: held ...
    assert len(calls) == 1
    tracker.module.storage[ps.ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = tiled
    tracker._flush_pair()                                                           # ... until the end of the pass
    assert calls[-1] == (256, 96, None) and account[0] == 0
    tracker._score_rows_paired(scores, 352, tiled, g2, a2, True, 1.0)
    tracker.module.storage[ps.ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = None
    with pytest.raises(RuntimeError, match="no tiled query gradients"):             # never dropped silently
        tracker._flush_pair()
    assert account[0] == 0

    calls.clear()
    tracker._score_rows_paired(scores, 0, tiled, g0, a0, True, 1.0)
    tracker._score_rows_paired(scores, 500, tiled, g1, a1, True, 1.0)              # not the next columns: the held one goes alone
    assert calls == [(0, 128, None)] and tracker._pair_held[1] == 500
    tracker._score_rows_paired(scores, 628, tiled, g2, a2, False, 1.0)             # other bias setting: no pair either
    assert calls == [(0, 128, None), (500, 128, None)] and tracker._pair_held[1] == 628
    tracker._drop_held()
    assert account[0] == 0

    calls.clear()
    big_g, big_a = batch(ps.PairwiseScoreTracker.PAIR_MAX_BATCH + 1, 3)
    tracker._score_rows_paired(scores, 0, tiled, big_g, big_a, True, 1.0)          # fills a 256-row tile on its own
    few = queries(ps.PairwiseScoreTracker.PAIR_MIN_QUERIES - 1)
    tracker._score_rows_paired(scores, 0, few, g0, a0, True, 1.0)                  # few queries: P is small, nothing to save
    assert calls == [(0, big_g.shape[0], None), (0, 128, None)] and tracker._pair_held is None

    tracker._score_rows_paired(scores, 0, tiled, g0, a0, True, 1.0)
    g0.add_(1)                                                                      # someone writes into a held tensor
    with pytest.raises(RuntimeError, match="modified in place"):
        tracker._score_rows_paired(scores, 128, tiled, g1, a1, True, 1.0)
    tracker._drop_held()
    assert account[0] == 0


def test_offloaded_activations_wait_on_the_host_and_come_back():
    """``BaseTracker._cache_activation`` / ``_take_activation`` with ``offload_activations_to_cpu``: the hooked input is copied to
    host memory and returned on its device when the gradient arrives; the Lambda tracker reads FactorArguments' flag, the
    score-stage trackers ScoreArguments'; without the flag the tensor is held by reference and version-checked."""
    from types import SimpleNamespace as T

    from kronfluence_amd.module.tracker.base import BaseTracker
    from kronfluence_amd.module.tracker.factor import LambdaTracker

    moves = []

    class OnDevice:   # what the hooks touch of a device tensor
        _version = 0

        def __init__(self, where):
            self.device = T(type=where)

        def to(self, target):
            where = target if isinstance(target, str) else target.type
            moves.append(where)
            return OnDevice(where)

    def tracker(cls, factor_flag, score_flag):
        t = cls.__new__(cls)
        t.module = T(name="m", factor_args=T(has_shared_parameters=False, offload_activations_to_cpu=factor_flag),
                     score_args=T(offload_activations_to_cpu=score_flag))
        t.cached_activations = None
        return t

    t = tracker(BaseTracker, factor_flag=False, score_flag=True)      # a score-stage tracker
    t._cache_activation(OnDevice("cuda"))
    assert moves == ["cpu"] and t.cached_activations[0].device.type == "cpu"
    assert t._take_activation().device.type == "cuda" and moves == ["cpu", "cuda"]
    moves.clear()
    t = tracker(BaseTracker, factor_flag=True, score_flag=False)      # FactorArguments' flag is not the score stage's
    x = OnDevice("cuda")
    t._cache_activation(x)
    assert moves == [] and t._take_activation() is x
    t = tracker(LambdaTracker, factor_flag=True, score_flag=False)    # ... it is the Lambda stage's
    t._cache_activation(OnDevice("cuda"))
    assert moves == ["cpu"] and t._take_activation().device.type == "cuda"
    t = tracker(LambdaTracker, factor_flag=False, score_flag=True)
    x = OnDevice("cuda")
    t._cache_activation(x)
    x._version = 1                                                     # held by reference: a later in-place write is caught
    with pytest.raises(RuntimeError, match="modified in place"):
        t._take_activation()


def test_verify_models_equivalence_follows_the_reference_rules():
    """utils/save.py:67-101: same keys, values equal within rtol 1.3e-6 / atol 1e-5 as fp32."""
    from kronfluence_amd.utils.save import verify_models_equivalence

    a = {"w": torch.arange(6.0).reshape(2, 3), "b": torch.ones(3, dtype=torch.float64)}
    assert verify_models_equivalence(a, {k: v.clone() for k, v in a.items()})
    assert verify_models_equivalence(a, {"w": a["w"] + 5e-6, "b": a["b"].to(torch.bfloat16)})
    assert not verify_models_equivalence(a, {"w": a["w"] + 1e-3, "b": a["b"]})
    assert not verify_models_equivalence(a, {"w": a["w"]})
    assert not verify_models_equivalence(a, {"w": a["w"], "c": a["b"]})
    assert not verify_models_equivalence(a, {"w": a["w"].reshape(3, 2), "b": a["b"]})


def test_batch_size_search_halves_on_memory_exhaustion_only():
    """``utils.dataset.find_executable_batch_size`` (reference utils/dataset.py:66-101), also what the Analyzer's automatic batch
    size uses: halve on out-of-memory, propagate anything else, fail at zero."""
    import kronfluence_amd
    from kronfluence_amd.utils.dataset import find_executable_batch_size

    assert kronfluence_amd.utils.dataset.find_executable_batch_size is find_executable_batch_size   # `utils` is exported
    tried = []

    def fits_below_600(batch_size):
        tried.append(batch_size)
        if batch_size >= 600:
            raise RuntimeError("HIP out of memory. Tried to allocate 1.00 GiB")

    assert find_executable_batch_size(fits_below_600, 4096) == 512 and tried == [4096, 2048, 1024, 512]

    def broken(batch_size):
        raise ValueError("not a memory problem")

    with pytest.raises(ValueError):
        find_executable_batch_size(broken, 8)

    def never(batch_size):
        raise torch.cuda.OutOfMemoryError("out of memory")

    with pytest.raises(RuntimeError, match="reached zero"):
        find_executable_batch_size(never, 4)


@pytest.mark.parametrize("q,o,ip,pad,conv,prepadded,blocks", [
    (5, 8, 16, 0, 0, 0, 1),        # Linear, nothing to pad
    (130, 16, 9, 7, 0, 0, 1),      # odd I' = 9 padded to 16; more queries than one conversion chunk (64)
    (70, 16, 9, 0, 0, 7, 3),       # the bf16 preconditioner already appended the 7 zero columns; three query batches
    (33, 8, 27, 0, 3, 0, 2),       # first conv layer: 3 channels x 9 taps, channels padded to 8 in (ky, kx, c) order
    (9, 4, 144, 0, 16, 0, 1),      # 16 channels x 9 taps: nothing padded, only re-ordered
])
def test_k_tile_major_query_layout_is_a_faithful_relayout(q, o, ip, pad, conv, prepadded, blocks):
    """``TiledQueries``: the held bf16 query gradients as ``[D' / 64, Q, 64]`` -- pure data movement, checked on the CPU: element
    ``(query, o, column)`` sits at ``d = o * width + position`` with the patch axis of a convolution re-ordered ``(c, ky, kx)`` ->
    ``(ky, kx, c_padded)``, appended columns / channels are zero, and ``dense()`` gives back the reference's ``[Q, O, I']``."""
    from kronfluence_amd.module.tracker.base import QueryBlocks
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    gen = torch.Generator().manual_seed(q + ip)
    want = torch.randn(q, o, ip, generator=gen).bfloat16()
    held = torch.nn.functional.pad(want, (0, prepadded)) if prepadded else want
    source = held if blocks == 1 else QueryBlocks(list(torch.tensor_split(held, blocks)))
    tiled = TiledQueries(source, pad, conv_channels=conv, prepadded=prepadded)
    width = tiled.shape[2]
    assert tiled.shape[:2] == (q, o) and tiled.tiled.shape == (o * width // 64, q, 64) and tiled.tiled.dtype == torch.bfloat16
    flat = tiled.tiled.transpose(0, 1).reshape(q, o, width)
    if conv:
        taps, cp = ip // conv, conv + (-conv) % 8
        assert width == taps * cp
        by_tap = flat.reshape(q, o, taps, cp)
        assert torch.equal(by_tap[..., :conv], want.reshape(q, o, conv, taps).transpose(2, 3))   # (c, tap) -> (tap, c)
        assert not by_tap[..., conv:].any()
    else:
        assert width == ip + pad + prepadded and width % 8 == 0
        assert torch.equal(flat[..., :ip], want) and not flat[..., ip:].any()
    assert torch.equal(tiled.dense(), want)


def test_conv_layout_helpers_are_exact_relayouts():
    """Pure data movement behind the implicit-im2col kernels, on the CPU: ``ops.conv_patch_order_eigenvectors`` (``Q_A^T`` with the
    patch axis re-ordered (c, ky, kx) -> (ky, kx, c_padded)) contracts a re-ordered patch exactly like ``Q_A`` contracts the
    reference's patch; ``ops.conv_geometry`` resolves string paddings like module/conv2d.py:46-53; ``ops.k_tile_major``."""
    from kronfluence_amd import ops
    from kronfluence_amd.utils.exceptions import UnsupportableModuleError

    channels, taps = 3, 9
    ip = channels * taps                                                 # a bias-free convolution: I' = C * taps
    gen = torch.Generator().manual_seed(0)
    q_a = torch.randn(ip, ip, generator=gen, dtype=torch.float64)
    perm = ops.conv_patch_order_eigenvectors(q_a, channels, taps)
    cp = 8
    assert perm.shape == (ip + (-ip) % 8, taps * cp) and perm.dtype == torch.bfloat16
    patch = torch.randn(channels, taps, generator=gen, dtype=torch.float64)            # reference order (c, tap)
    patch_k = torch.zeros(taps, cp, dtype=torch.float64)
    patch_k[:, :channels] = patch.t()                                                    # kernel order (tap, c_padded)
    want = q_a.to(torch.bfloat16).double().t() @ patch.reshape(-1)                      # (Q_A^T patch)[i'] in the reference's order
    got = perm.double() @ patch_k.reshape(-1)
    assert torch.allclose(got[:ip], want, atol=1e-12) and not got[ip:].any()

    for padding, kernel, dilation, expect in (("same", (3, 5), (1, 1), (1, 2)), ("valid", (3, 3), (1, 1), (0, 0)),
                                              ("same", (3, 3), (2, 2), (2, 2)), ((2, 1), (5, 3), (1, 1), (2, 1))):
        conv = nn.Conv2d(4, 6, kernel, padding=padding, dilation=dilation)
        geometry = ops.conv_geometry(conv)
        assert geometry[4:6] == expect and geometry[:2] == kernel and geometry[6:] == dilation
        x = torch.randn(2, 4, 11, 13)
        h = (11 + 2 * geometry[4] - dilation[0] * (kernel[0] - 1) - 1) // 1 + 1
        w = (13 + 2 * geometry[5] - dilation[1] * (kernel[1] - 1) - 1) // 1 + 1
        assert conv(x).shape[2:] == (h, w)
    with pytest.raises(UnsupportableModuleError):
        ops.conv_geometry(nn.Conv2d(4, 6, (2, 3), padding="same"))        # even kernel: unequal padding, as in the reference

    p = torch.arange(3 * 4 * 32, dtype=torch.float32).reshape(3, 4, 32)   # [rows, ...] with 128 elements per row
    tiled = ops.k_tile_major(p)
    assert tiled.shape == (2, 3, 64) and torch.equal(tiled.transpose(0, 1).reshape(3, 128), p.reshape(3, 128))


def test_side_stream_join_finds_every_tensor_a_layer_keeps():
    """what a hook allocates while the side stream is current is read on the caller's stream after the join, so the
    join marks every tensor of ``module.storage`` as in use there.  The walk that finds them is host logic: tensors, lists of
    tensors, dicts and the tensor attributes of holder objects (``TiledQueries``); CPU tensors and scalars are skipped."""
    from types import SimpleNamespace

    from kronfluence_amd.module.tracker import base

    class FakeDeviceTensor(torch.Tensor):
        is_cuda = True

    def on_device(*shape):
        return torch.zeros(*shape).as_subclass(FakeDeviceTensor)

    a, b, c, d = on_device(2), on_device(3), on_device(4), on_device(5)
    holder = SimpleNamespace(tiled=c, num_queries=7, nested=SimpleNamespace(block=d, deeper=SimpleNamespace(never=on_device(1))))
    storage = {"covariance": a, "count": 3, "host": torch.zeros(2), "none": None, "list": [b, torch.zeros(1)], "queries": holder}
    found = list(base._device_tensors(storage))
    assert [t.shape[0] for t in found] == [2, 3, 4, 5]   # two levels of holder objects, not a third


def test_set_factors_shares_only_read_only_accelerator_factors():
    """``set_factors(share=...)`` (round 6): with ``clone=True`` a factor is handed to the module as it is ONLY when it is named
    read-only for the stage AND already lives on the accelerator; host tensors and every other factor (counters, accumulators a
    stage adds to in place) are cloned as the reference does (module/utils.py:158-177).  ``Ekfac.prepare`` with a bf16
    preconditioner leaves bf16-stored eigenvectors bf16 (converted on first use by the fp32 paths), fp32 ones fp32."""
    from kronfluence_amd.factor.config import FactorConfig
    from kronfluence_amd.module.utils import READ_ONLY_FACTORS, READ_ONLY_FACTORS_WHEN_SCORING, set_factors

    assert READ_ONLY_FACTORS == (C.ACTIVATION_EIGENVECTORS_NAME, C.GRADIENT_EIGENVECTORS_NAME)
    assert READ_ONLY_FACTORS_WHEN_SCORING == READ_ONLY_FACTORS + (C.LAMBDA_MATRIX_NAME,)
    model = prepare_model(fx.make_model("mlp"), _Task())
    module = next(m for m in model.modules() if isinstance(m, TrackedModule))
    host = torch.eye(3)
    set_factors(model, C.ACTIVATION_EIGENVECTORS_NAME, {module.name: host}, clone=True, share=READ_ONLY_FACTORS)
    stored = module.storage[C.ACTIVATION_EIGENVECTORS_NAME]
    assert stored is not host and torch.equal(stored, host)            # a host tensor is cloned whatever its name
    set_factors(model, C.ACTIVATION_EIGENVECTORS_NAME, {module.name: host}, clone=False)
    assert module.storage[C.ACTIVATION_EIGENVECTORS_NAME] is host

    class _Resident(torch.Tensor):   # stands in for an accelerator tensor on a box without one
        is_cuda = True

    resident = torch.eye(3).as_subclass(_Resident)
    set_factors(model, C.GRADIENT_EIGENVECTORS_NAME, {module.name: resident}, clone=True, share=READ_ONLY_FACTORS)
    assert module.storage[C.GRADIENT_EIGENVECTORS_NAME] is resident    # read-only + resident: shared
    set_factors(model, C.LAMBDA_MATRIX_NAME, {module.name: resident}, clone=True, share=READ_ONLY_FACTORS)
    assert module.storage[C.LAMBDA_MATRIX_NAME] is not resident        # the Lambda stage accumulates into Lambda: cloned
    set_factors(model, C.LAMBDA_MATRIX_NAME, {module.name: resident}, clone=True, share=READ_ONLY_FACTORS_WHEN_SCORING)
    assert module.storage[C.LAMBDA_MATRIX_NAME] is resident            # scoring replaces it by its damped inverse: shared
    set_factors(model, C.NUM_LAMBDA_PROCESSED, {module.name: resident}, clone=True, share=READ_ONLY_FACTORS_WHEN_SCORING)
    assert module.storage[C.NUM_LAMBDA_PROCESSED] is not resident      # counters are never shared


def test_base_tracker_converts_eigenvectors_to_fp32_on_first_use_only():
    """``BaseTracker._eigenvectors32``: what the fp32 paths call once ``Ekfac.prepare`` leaves bf16-stored eigenvectors alone."""
    model = prepare_model(fx.make_model("mlp"), _Task())
    module = next(m for m in model.modules() if isinstance(m, TrackedModule))
    tracker = next(iter(module._trackers.values()))
    low = torch.eye(4, dtype=torch.bfloat16)
    module.storage[C.ACTIVATION_EIGENVECTORS_NAME] = low
    first = tracker._eigenvectors32(C.ACTIVATION_EIGENVECTORS_NAME)
    assert first.dtype == torch.float32 and module.storage[C.ACTIVATION_EIGENVECTORS_NAME] is first
    assert tracker._eigenvectors32(C.ACTIVATION_EIGENVECTORS_NAME) is first   # converted once, kept
