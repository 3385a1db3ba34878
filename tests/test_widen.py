"""Rows of SURVEY.md section 8(f) ("next"): data/module partitions + aggregation + resume, the other factor
strategies, self-influence, gradient aggregation, per-module / per-token scores.

Every scenario runs twice: ``[gpu]`` through libkronfluence_hip.so on an MI355X (``-m gpu``), and ``[cpu-hostlogic]``
with the HIP leaf operators replaced by the torch stand-ins of tests/cpu_engine.py (host logic only; see there).
Expected values are the reference's golden tensors (tests/golden/make_golden.py) or invariances the reference's own
tests assert (partitioned == un-partitioned, batch-size independence)."""

import os

import pytest
import torch
from safetensors.torch import load_file

import fixtures as fx
from test_pipeline_gpu import GOLDEN, build, nested, rel

ENGINES = [pytest.param("cpu", id="cpu-hostlogic"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


@pytest.fixture(params=ENGINES)
def engine(request):
    if request.param == "cpu":
        request.getfixturevalue("cpu_engine")
    return request.param


def _tol(engine, cpu, gpu):
    return cpu if engine == "cpu" else gpu


# ---- 8(f)-2: partitions, aggregation, skip-if-exists -------------------------------------------------------------
@pytest.mark.parametrize("kind", ["mlp", "conv"])
def test_partitioned_factors_and_scores_equal_unpartitioned(kind, tmp_path, engine):
    from kronfluence_amd import FactorArguments, ScoreArguments

    spec, analyzer, train, query = build(kind, tmp_path)
    analyzer.fit_all_factors("whole", train, per_device_batch_size=spec.factor_batch,
                             factor_args=FactorArguments(use_empirical_fisher=True))
    split = FactorArguments(use_empirical_fisher=True, covariance_data_partitions=2, covariance_module_partitions=2,
                            lambda_data_partitions=3, lambda_module_partitions=2)
    analyzer.fit_all_factors("split", train, per_device_batch_size=spec.factor_batch, factor_args=split)
    out = analyzer.factors_output_dir("split")
    for i in range(2):
        for j in range(2):
            assert (out / f"activation_covariance_data_partition{i}_module_partition{j}.safetensors").exists()
    assert (out / "lambda_matrix_data_partition2_module_partition1.safetensors").exists()
    tol = _tol(engine, 1e-6, 2e-6)
    whole, parts = analyzer.load_covariance_matrices("whole"), analyzer.load_covariance_matrices("split")
    for factor in whole:
        assert set(whole[factor]) == set(parts[factor])
        for module, want in whole[factor].items():
            if want.dtype == torch.int64:
                assert torch.equal(parts[factor][module], want)
            else:
                assert rel(parts[factor][module], want) <= tol, (factor, module)
    # Lambda of the split run on ITS eigenbasis; compare on a common basis by re-fitting with the whole run's eigenvectors
    analyzer.fit_lambda_matrices("split2", train, per_device_batch_size=spec.factor_batch, factor_args=split,
                                 load_from_factors_name="whole")
    whole_l, parts_l = analyzer.load_lambda_matrices("whole"), analyzer.load_lambda_matrices("split2")
    for module, want in whole_l["lambda_matrix"].items():
        assert rel(parts_l["lambda_matrix"][module], want) <= _tol(engine, 1e-6, 2e-5), module
        assert torch.equal(parts_l["num_lambda_processed"][module], whole_l["num_lambda_processed"][module])

    common = dict(per_device_query_batch_size=spec.query_batch, per_device_train_batch_size=spec.train_batch)
    want = analyzer.compute_pairwise_scores("s_whole", "whole", query, train,
                                            score_args=ScoreArguments(damping_factor=None), **common)["all_modules"]
    got = analyzer.compute_pairwise_scores("s_split", "whole", query, train,
                                           score_args=ScoreArguments(damping_factor=None, data_partitions=3, module_partitions=2),
                                           **common)["all_modules"]
    assert got.shape == want.shape and rel(got, want) <= _tol(engine, 1e-6, 2e-5), rel(got, want)
    assert (analyzer.scores_output_dir("s_split") / "pairwise_scores_data_partition2_module_partition1.safetensors").exists()
    assert rel(analyzer.load_pairwise_scores("s_split")["all_modules"], want) <= _tol(engine, 1e-6, 2e-5)


def test_target_partitions_resume_and_late_aggregation(tmp_path, engine):
    from kronfluence_amd import FactorArguments, ScoreArguments

    spec, analyzer, train, query = build("mlp", tmp_path)
    args = FactorArguments(use_empirical_fisher=True, covariance_data_partitions=2, covariance_module_partitions=1)
    with pytest.raises(ValueError):
        analyzer.fit_covariance_matrices("bad", train, per_device_batch_size=8,
                                         factor_args=FactorArguments(), target_data_partitions=[0])
    with pytest.raises(ValueError):
        analyzer.fit_covariance_matrices("f", train, per_device_batch_size=8, factor_args=args, target_data_partitions=[2])
    analyzer.fit_covariance_matrices("f", train, per_device_batch_size=8, factor_args=args, target_data_partitions=[0])
    assert analyzer.load_covariance_matrices("f") is None  # one partition missing: nothing aggregated yet
    analyzer.fit_covariance_matrices("f", train, per_device_batch_size=8, factor_args=args, target_data_partitions=1)
    cov = analyzer.load_covariance_matrices("f")
    assert cov is not None and int(cov["num_activation_covariance_processed"]["0"].item()) == spec.n_train
    # skip-if-exists: a second call leaves the files untouched
    stamp = (analyzer.factors_output_dir("f") / "activation_covariance.safetensors").stat().st_mtime_ns
    analyzer.fit_covariance_matrices("f", train, per_device_batch_size=8, factor_args=args)
    assert (analyzer.factors_output_dir("f") / "activation_covariance.safetensors").stat().st_mtime_ns == stamp
    # explicit aggregation entry points exist and are idempotent
    analyzer.aggregate_covariance_matrices("f")
    analyzer.perform_eigendecomposition("f", factor_args=args)
    analyzer.fit_lambda_matrices("f", train, per_device_batch_size=8, factor_args=args)
    score_args = ScoreArguments(damping_factor=None, data_partitions=2)
    first = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=3,
                                             per_device_train_batch_size=12, score_args=score_args, target_data_partitions=[1])
    assert first is None and analyzer.load_pairwise_scores("s") is None
    full = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=3,
                                            per_device_train_batch_size=12, score_args=score_args)
    assert full["all_modules"].shape == (spec.n_query, spec.n_train)
    assert torch.equal(analyzer.aggregate_pairwise_scores("s")["all_modules"], full["all_modules"])


# ---- 8(f)-4: identity / diagonal / kfac strategies ---------------------------------------------------------------
def _widen(kind):
    return load_file(os.path.join(GOLDEN, f"widen_{kind}_fp64.safetensors"))


@pytest.mark.parametrize("strategy", ["kfac", "diagonal", "identity"])
@pytest.mark.parametrize("kind", list(fx.FIXTURES))
def test_other_strategies_match_reference(kind, strategy, tmp_path, engine):
    from kronfluence_amd import FactorArguments, ScoreArguments

    gold = _widen(kind)
    spec, analyzer, train, query = build(kind, tmp_path)
    analyzer.fit_all_factors("f", train, per_device_batch_size=spec.factor_batch,
                             factor_args=FactorArguments(strategy=strategy, use_empirical_fisher=True))
    out = analyzer.factors_output_dir("f")
    assert (out / "activation_covariance.safetensors").exists() == (strategy == "kfac")
    assert (out / "lambda_matrix.safetensors").exists() == (strategy == "diagonal")
    if strategy == "diagonal":
        lam = analyzer.load_lambda_matrices("f")
        for key, want in gold.items():
            if key.startswith("strategy/diagonal/lam/lambda_matrix/"):
                module = key.rsplit("/", 1)[1]
                assert rel(lam["lambda_matrix"][module], want) <= _tol(engine, 1e-6, 2e-5), module
    scores = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=spec.query_batch,
                                              per_device_train_batch_size=spec.train_batch,
                                              score_args=ScoreArguments(damping_factor=None))["all_modules"]
    want = gold[f"strategy/{strategy}/scores"]
    assert scores.shape == want.shape
    assert rel(scores, want) <= _tol(engine, 1e-5, 5e-4 if strategy == "kfac" else 1e-4), rel(scores, want)


# ---- 8(f)-3: self-influence -------------------------------------------------------------------------------------
@pytest.mark.parametrize("strategy", ["ekfac", "kfac", "diagonal", "identity"])
@pytest.mark.parametrize("kind", list(fx.FIXTURES))
def test_self_scores_match_reference(kind, strategy, tmp_path, engine):
    from kronfluence_amd import FactorArguments, ScoreArguments

    gold = _widen(kind)
    spec, analyzer, train, _query = build(kind, tmp_path)
    analyzer.fit_all_factors("f", train, per_device_batch_size=spec.factor_batch,
                             factor_args=FactorArguments(strategy=strategy, use_empirical_fisher=True))
    scores = analyzer.compute_self_scores("self", "f", train, per_device_train_batch_size=spec.train_batch,
                                          score_args=ScoreArguments(damping_factor=None))["all_modules"]
    want = gold[f"strategy/{strategy}/self"]
    assert scores.shape == want.shape == (spec.n_train,)
    tol = _tol(engine, 1e-5, 5e-4 if strategy in ("ekfac", "kfac") else 1e-4)
    assert rel(scores, want) <= tol, rel(scores, want)
    assert (analyzer.scores_output_dir("self") / "self_scores.safetensors").exists()
    if strategy != "ekfac":
        return
    # batch-size independence, per-module split, data/module partitions (reference tests/scores/test_self_scores.py)
    other = analyzer.compute_self_scores("self_b", "f", train, per_device_train_batch_size=7,
                                         score_args=ScoreArguments(damping_factor=None, data_partitions=2, module_partitions=2))
    assert rel(other["all_modules"], scores) <= _tol(engine, 1e-6, 2e-5)
    per_module = analyzer.compute_self_scores("self_pm", "f", train, per_device_train_batch_size=spec.train_batch,
                                              score_args=ScoreArguments(damping_factor=None, compute_per_module_scores=True))
    for key, value in gold.items():
        if key.startswith("self_permodule/"):
            assert rel(per_module[key.split("/", 1)[1]], value) <= tol, key
    assert rel(sum(per_module.values()), scores) <= _tol(engine, 1e-6, 2e-5)
    measured = analyzer.compute_self_scores("self_m", "f", train, per_device_train_batch_size=spec.train_batch,
                                            score_args=ScoreArguments(damping_factor=None, use_measurement_for_self_influence=True))
    assert rel(measured["all_modules"], gold["self_measurement"]) <= tol, rel(measured["all_modules"], gold["self_measurement"])


# ---- 8(f)-4: per-module / per-token scores, query / train gradient aggregation ------------------------------------
@pytest.mark.parametrize("kind", list(fx.FIXTURES))
def test_score_reductions_match_reference(kind, tmp_path, engine):
    from kronfluence_amd import FactorArguments, ScoreArguments

    gold = _widen(kind)
    spec, analyzer, train, query = build(kind, tmp_path)
    analyzer.fit_all_factors("f", train, per_device_batch_size=spec.factor_batch,
                             factor_args=FactorArguments(use_empirical_fisher=True))
    common = dict(per_device_query_batch_size=spec.query_batch, per_device_train_batch_size=spec.train_batch)
    tol = _tol(engine, 1e-5, 5e-4)

    def run(name, **kw):
        return analyzer.compute_pairwise_scores(name, "f", query, train, score_args=ScoreArguments(damping_factor=None, **kw),
                                                **common)

    def agg_err(got, tag):
        """Aggregated scores are sums with cancellation: error relative to the norm of what was summed."""
        pairwise = gold["strategy/ekfac/scores"].double()
        dims = {"aggq": (0,), "aggt": (1,), "aggqt": (0, 1)}[tag]
        scale = pairwise.abs().sum(dim=dims, keepdim=True)
        return float(((got.double() - gold[tag].double()).abs() / scale).max())

    per_module = run("pm", compute_per_module_scores=True)
    wanted = {k.split("/", 1)[1]: v for k, v in gold.items() if k.startswith("permodule/")}
    assert set(per_module) == set(wanted)
    for module, want in wanted.items():
        assert rel(per_module[module], want) <= tol, (module, rel(per_module[module], want))
    for tag, kw in (("aggq", dict(aggregate_query_gradients=True)), ("aggt", dict(aggregate_train_gradients=True)),
                    ("aggqt", dict(aggregate_query_gradients=True, aggregate_train_gradients=True))):
        got = run(tag, **kw)["all_modules"]
        assert got.shape == gold[tag].shape, (tag, got.shape)
        assert agg_err(got, tag) <= tol, (tag, agg_err(got, tag))
    # train aggregation adds over data partitions (score_computer.py:122-124)
    got = run("aggt_parts", aggregate_train_gradients=True, data_partitions=2)["all_modules"]
    assert agg_err(got, "aggt") <= tol
    if kind == "seq":
        got = run("tok", compute_per_token_scores=True)["all_modules"]
        assert got.shape == gold["pertoken"].shape
        assert rel(got, gold["pertoken"]) <= tol, rel(got, gold["pertoken"])
        assert rel(got.sum(-1), gold["strategy/ekfac/scores"]) <= tol


def test_score_sink_refuses_mixed_token_axes():
    """A block that first receives a token-less layer and then a sequence layer must refuse, as the reference's
    ``add_`` does (score/dot_product.py:33-36, 114-116)."""
    from kronfluence_amd.module.tracker.pairwise_score import ScoreSink

    sink = ScoreSink(2, 4, torch.device("cpu"), per_token=True)
    assert sink.matrix(1).shape == (2, 4)
    with pytest.raises(RuntimeError, match="token-wise"):
        sink.matrix(6)
    sink = ScoreSink(2, 4, torch.device("cpu"), per_token=True)
    assert sink.matrix(6).shape == (2, 24) and sink.result().shape == (2, 4, 6)


# ---- 8(f)-1: low-rank query batching ------------------------------------------------------------------------------
def _spearman(a, b):
    ra, rb = a.flatten().argsort().argsort().double(), b.flatten().argsort().argsort().double()
    ra, rb = ra - ra.mean(), rb - rb.mean()
    return float((ra @ rb) / (ra.norm() * rb.norm()))


@pytest.mark.parametrize("kind", list(fx.FIXTURES))
def test_low_rank_query_gradients_match_reference(kind, tmp_path, engine):
    """``query_gradient_low_rank=4``: on these small layers the range finder spans the whole row space, so the
    factors are the exact truncated SVD and the scores equal the reference's ``use_full_svd=True`` run; the reference's
    own bar for this option is rank correlation with the full-rank scores (tests/scores/test_pairwise_scores.py:977-978)."""
    from kronfluence_amd import FactorArguments, ScoreArguments
    from kronfluence_amd.module.tracked_module import TrackedModule
    from kronfluence_amd.utils.constants import ACCUMULATED_PRECONDITIONED_GRADIENT_NAME

    gold = _widen(kind)
    spec, analyzer, train, query = build(kind, tmp_path)
    analyzer.fit_all_factors("f", train, per_device_batch_size=spec.factor_batch,
                             factor_args=FactorArguments(use_empirical_fisher=True))
    common = dict(per_device_query_batch_size=spec.query_batch, per_device_train_batch_size=spec.train_batch)
    for name, full_svd in (("lr", False), ("lr_full", True)):
        got = analyzer.compute_pairwise_scores(name, "f", query, train, **common,
                                               score_args=ScoreArguments(damping_factor=None, query_gradient_low_rank=4,
                                                                         use_full_svd=full_svd))["all_modules"]
        assert got.shape == gold["lowrank4"].shape
        assert rel(got, gold["lowrank4"]) <= _tol(engine, 1e-4, 2e-3), (name, rel(got, gold["lowrank4"]))
        assert _spearman(got, gold["strategy/ekfac/scores"]) >= _spearman(gold["lowrank4"], gold["strategy/ekfac/scores"]) - 0.01
    # same through query accumulation, module partitions and query aggregation on low-rank factors
    again = analyzer.compute_pairwise_scores("lr_acc", "f", query, train, per_device_query_batch_size=2,
                                             per_device_train_batch_size=7,
                                             score_args=ScoreArguments(damping_factor=None, query_gradient_low_rank=4,
                                                                       query_gradient_accumulation_steps=2,
                                                                       module_partitions=2))["all_modules"]
    assert rel(again, got) <= _tol(engine, 1e-5, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("held_in_eigenbasis", [True, False])
def test_query_batches_of_mixed_layout_accumulate(held_in_eigenbasis):
    """``PreconditionTracker.accumulate_iterations`` with query batches laid out differently within one accumulation window (a
    one-row batch -- held in the eigenbasis, fp32, unpadded -- among sequence batches -- parameter space, zero-padded -- or the other
    way round): the later block is brought to the held layout (``Q_G^T P Q_A`` / ``Q_G M Q_A^T``, padding), as the reference accepts
    such mixes (module/tracker/precondition.py:203-240)."""
    from kronfluence_amd import Task, prepare_model
    from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
    from kronfluence_amd.utils.constants import (ACCUMULATED_PRECONDITIONED_GRADIENT_NAME, ACTIVATION_EIGENVECTORS_NAME,
                                                 GRADIENT_EIGENVECTORS_NAME, PRECONDITIONED_GRADIENT_NAME)

    class T(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return model(batch[0]).sum()

        def compute_measurement(self, batch, model):
            return model(batch[0]).sum()

    dev = "cuda:0"
    model = prepare_model(torch.nn.Sequential(torch.nn.Linear(15, 8)), T()).to(dev)
    m = [x for x in model.modules() if isinstance(x, TrackedModule)][0]
    m.current_mode = ModuleMode.PRECONDITION_GRADIENT
    gen = torch.Generator().manual_seed(0)
    q_a = torch.linalg.qr(torch.randn(16, 16, generator=gen, dtype=torch.float64))[0]
    q_g = torch.linalg.qr(torch.randn(8, 8, generator=gen, dtype=torch.float64))[0]
    m.storage[ACTIVATION_EIGENVECTORS_NAME], m.storage[GRADIENT_EIGENVECTORS_NAME] = q_a.float().to(dev), q_g.float().to(dev)
    first, second = torch.randn(3, 8, 16, generator=gen), torch.randn(2, 8, 16, generator=gen)
    pad = 0 if held_in_eigenbasis else 8
    # first batch defines the held layout
    m.storage[PRECONDITIONED_GRADIENT_NAME] = torch.nn.functional.pad(first, (0, pad)).to(dev)
    m.queries_in_eigenbasis, m.query_padding = held_in_eigenbasis, pad
    m.accumulate_iterations()
    # second batch arrives in the other layout
    other_pad = 8 if held_in_eigenbasis else 0
    m.storage[PRECONDITIONED_GRADIENT_NAME] = torch.nn.functional.pad(second, (0, other_pad)).to(dev)
    m.queries_in_eigenbasis, m.query_padding = not held_in_eigenbasis, other_pad
    m.accumulate_iterations()
    held = m.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME].dense().double().cpu()
    assert m.queries_in_eigenbasis == held_in_eigenbasis and m.query_padding == pad and held.shape == (5, 8, 16 + pad)
    if held_in_eigenbasis:
        want = q_g.t() @ second.double() @ q_a
    else:
        want = q_g @ second.double() @ q_a.t()
    assert torch.equal(held[:3, :, :16], first.double())
    assert float((held[3:, :, :16] - want).norm() / want.norm()) <= 1e-5
    assert float(held[..., 16:].abs().max()) == 0.0 if pad else True



# ---- memory-strategy options of the reference: same results -----------------------------------------------------------
@pytest.mark.parametrize("kind", ["mlp", "seq"])
def test_memory_strategy_options_change_nothing(kind, tmp_path, engine):
    """``offload_activations_to_cpu`` (FactorArguments for the Lambda stage, ScoreArguments for the score stages; reference
    tracker/factor.py:239, pairwise_score.py:59) and ``use_iterative_lambda_aggregation`` (tracker/factor.py:203-213) choose where an
    activation waits and how a batch is walked -- never what is computed."""
    from kronfluence_amd import FactorArguments, ScoreArguments

    spec, analyzer, train, query = build(kind, tmp_path)
    results = []
    for tag, frugal in (("plain", False), ("frugal", True)):
        analyzer.fit_all_factors(f"f_{tag}", train, per_device_batch_size=spec.factor_batch,
                                 factor_args=FactorArguments(use_empirical_fisher=True, has_shared_parameters=True,
                                                             offload_activations_to_cpu=frugal, use_iterative_lambda_aggregation=frugal))
        scores = analyzer.compute_pairwise_scores(f"s_{tag}", f"f_{tag}", query, train, per_device_query_batch_size=spec.query_batch,
                                                  per_device_train_batch_size=spec.train_batch,
                                                  score_args=ScoreArguments(damping_factor=None, offload_activations_to_cpu=frugal))
        selfs = analyzer.compute_self_scores(f"self_{tag}", f"f_{tag}", train, per_device_train_batch_size=spec.train_batch,
                                             score_args=ScoreArguments(damping_factor=None, offload_activations_to_cpu=frugal))
        results.append((analyzer.load_lambda_matrices(f"f_{tag}")["lambda_matrix"], scores["all_modules"], selfs["all_modules"]))
    (lam0, s0, d0), (lam1, s1, d1) = results
    for module in lam0:   # fp32 storage summed sample by sample instead of batch by batch: rounding order only
        assert rel(lam1[module], lam0[module]) <= _tol(engine, 1e-6, 2e-5), module
    assert rel(s1, s0) <= _tol(engine, 1e-6, 1e-4) and rel(d1, d0) <= _tol(engine, 1e-6, 1e-4)   # GPU: two fits differ by atomic order
