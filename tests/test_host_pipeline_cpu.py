"""Host logic of the full Analyzer pipeline on a GPU-less machine: the HIP leaf operators are replaced by the
torch-CPU stand-ins of tests/cpu_engine.py (``cpu_engine`` fixture), everything above them -- hooks, trackers,
stage loops, file layout -- is the product's.  Checked against the reference's golden tensors."""

import os

import pytest
import torch
from safetensors.torch import load_file

import fixtures as fx
from test_pipeline_gpu import GOLDEN, build, nested, rel


@pytest.mark.parametrize("kind", list(fx.FIXTURES))
def test_pipeline_host_logic_reproduces_reference_goldens(kind, tmp_path, cpu_engine):
    from kronfluence_amd import FactorArguments, ScoreArguments

    gold = load_file(os.path.join(GOLDEN, f"{kind}_fp64.safetensors"))
    spec, analyzer, train, query = build(kind, tmp_path)
    analyzer.fit_all_factors("f", train, per_device_batch_size=spec.factor_batch,
                             factor_args=FactorArguments(use_empirical_fisher=True))
    cov = analyzer.load_covariance_matrices("f")
    for factor, per_module in nested(gold, "cov").items():
        for module, want in per_module.items():
            got = cov[factor][module]
            if want.dtype == torch.int64:
                assert torch.equal(got.reshape(-1), want.reshape(-1)), (factor, module)
            else:
                assert rel(got, want) <= 1e-6, (factor, module)
    lam = analyzer.load_lambda_matrices("f")
    for module, want in nested(gold, "lam")["lambda_matrix"].items():
        assert rel(lam["lambda_matrix"][module], want) <= 1e-4, (module, rel(lam["lambda_matrix"][module], want))
    scores = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=spec.query_batch,
                                              per_device_train_batch_size=spec.train_batch,
                                              score_args=ScoreArguments(damping_factor=None))["all_modules"]
    assert rel(scores, gold["scores/dampNone"]) <= 1e-4, rel(scores, gold["scores/dampNone"])


def test_automatic_batch_size_search_halves_on_out_of_memory(tmp_path, cpu_engine, monkeypatch):
    """``per_device_batch_size=None``: the stage is probed with one batch, halving from the initial attempt until it no
    longer runs out of memory (reference utils/dataset.py:66-101); results equal an explicit-batch-size run."""
    from kronfluence_amd import FactorArguments, ScoreArguments, ops

    spec, analyzer, train, query = build("mlp", tmp_path)
    real_syrk, real_score = ops.syrk_accum, ops.pairwise_score
    seen = {"cov": [], "score": []}

    def syrk(cov, x, n_rows, *args, **kwargs):
        seen["cov"].append(n_rows)
        if n_rows > 20:
            raise RuntimeError("HIP out of memory. Tried to allocate 1.00 GiB")
        return real_syrk(cov, x, n_rows, *args, **kwargs)

    def score(scores, col_offset, p, g, a, *args, **kwargs):
        seen["score"].append(g.shape[0])
        if g.shape[0] > 10:
            raise torch.cuda.OutOfMemoryError("HIP out of memory")
        return real_score(scores, col_offset, p, g, a, *args, **kwargs)

    monkeypatch.setattr(ops, "syrk_accum", syrk)
    monkeypatch.setattr(ops, "pairwise_score", score)
    args = FactorArguments(use_empirical_fisher=True)
    analyzer.fit_all_factors("auto", train, factor_args=args, initial_per_device_batch_size_attempt=64)
    assert max(seen["cov"]) == 48 and 24 in seen["cov"] and 12 in seen["cov"]  # 64 -> min(64, 48) = 48 -> 24 -> 12
    got = analyzer.compute_pairwise_scores("auto", "auto", query, train, per_device_query_batch_size=3,
                                           score_args=ScoreArguments(damping_factor=None),
                                           initial_per_device_train_batch_size_attempt=32)["all_modules"]
    assert 32 in seen["score"] and 16 in seen["score"] and 8 in seen["score"]
    monkeypatch.setattr(ops, "syrk_accum", real_syrk)
    monkeypatch.setattr(ops, "pairwise_score", real_score)
    analyzer.fit_all_factors("fixed", train, per_device_batch_size=12, factor_args=args)
    want = analyzer.compute_pairwise_scores("fixed", "fixed", query, train, per_device_query_batch_size=3,
                                            per_device_train_batch_size=8,
                                            score_args=ScoreArguments(damping_factor=None))["all_modules"]
    assert rel(got, want) <= 1e-5
    # errors that are not memory exhaustion propagate unchanged
    monkeypatch.setattr(ops, "syrk_accum", lambda *a, **k: (_ for _ in ()).throw(ValueError("boom")))
    with pytest.raises(ValueError, match="boom"):
        analyzer.fit_covariance_matrices("bad", train, factor_args=args)


def test_analyzer_bookkeeping_helpers(tmp_path, cpu_engine):
    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments

    spec, analyzer, train, query = build("mlp", tmp_path)
    assert analyzer.load_factor_args("f") is None and analyzer.load_score_args("s") is None
    args = FactorArguments(use_empirical_fisher=True, lambda_dtype=torch.bfloat16)
    analyzer.fit_all_factors("f", train, per_device_batch_size=16, factor_args=args)
    assert analyzer.load_factor_args("f") == args
    sargs = ScoreArguments(damping_factor=None, score_dtype=torch.bfloat16)
    analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=3, per_device_train_batch_size=12,
                                     score_args=sargs)
    assert analyzer.load_score_args("s") == sargs
    with pytest.raises(ValueError):  # same name, different arguments, no overwrite
        analyzer.compute_self_scores("s", "f", train, per_device_train_batch_size=12, score_args=ScoreArguments())
    stored = Analyzer.load_file(analyzer.scores_output_dir("s") / "pairwise_scores.safetensors")
    assert stored["all_modules"].shape == (spec.n_query, spec.n_train) and stored["all_modules"].dtype == torch.bfloat16
    with pytest.raises(FileNotFoundError):
        Analyzer.load_file(str(tmp_path / "missing.safetensors"))
    summary = Analyzer.get_module_summary(fx.make_model("conv"))
    assert summary.startswith("==Model Summary==") and "Module Name: `0`" in summary and "ReLU" not in summary


# ---- GPU scenarios re-run for their HOST logic (the arithmetic below the ops boundary is the stand-in's) -----------
@pytest.mark.parametrize("kind", list(fx.FIXTURES))
def test_stage_isolated_scores_and_batching_invariances_host_logic(kind, tmp_path, cpu_engine):
    from test_pipeline_gpu import test_stage_isolated_scores_on_reference_factors as scenario

    scenario(kind, tmp_path)


@pytest.mark.parametrize("kind", list(fx.MSE_FIXTURES))
def test_default_damping_host_logic(kind, tmp_path, cpu_engine):
    from test_pipeline_gpu import test_default_damping_matches_reference_goldens as scenario

    scenario(kind, tmp_path)


@pytest.mark.parametrize("kind", list(fx.FIXTURES))
def test_default_damping_ill_conditioned_host_logic(kind, tmp_path, cpu_engine):
    from test_pipeline_gpu import test_default_damping_on_ill_conditioned_fixtures as scenario

    scenario(kind, tmp_path)


def test_shared_parameters_host_logic(tmp_path, cpu_engine):
    from test_pipeline_gpu import test_shared_parameters_factors_match_reference_and_scores_match_autograd as scenario

    scenario(tmp_path)


def test_conv8_fp32_host_logic(tmp_path, cpu_engine):
    from test_pipeline_gpu import test_conv8_fp32_matches_reference_goldens as scenario

    scenario(tmp_path)


def test_dataset_metadata_guards_the_output_directory(tmp_path, cpu_engine):
    import json

    from kronfluence_amd import FactorArguments, ScoreArguments

    spec, analyzer, train, query = build("mlp", tmp_path)
    args = FactorArguments(use_empirical_fisher=True)
    analyzer.fit_all_factors("f", train, per_device_batch_size=16, factor_args=args)
    meta = json.load(open(analyzer.factors_output_dir("f") / "covariance_dataset_metadata.json"))
    assert meta == {"type": "TensorDataset", "dataset_size": spec.n_train, "indices": None}
    assert (analyzer.factors_output_dir("f") / "lambda_dataset_metadata.json").exists()
    kw = dict(per_device_query_batch_size=3, per_device_train_batch_size=12, score_args=ScoreArguments(damping_factor=None))
    sub = analyzer.compute_pairwise_scores("s", "f", query, train, query_indices=[0, 2, 4], train_indices=list(range(10, 30)), **kw)
    assert sub["all_modules"].shape == (3, 20)
    meta = json.load(open(analyzer.scores_output_dir("s") / "train_dataset_metadata.json"))
    assert meta["indices"] == list(range(10, 30)) and meta["dataset_size"] == spec.n_train
    full = analyzer.compute_pairwise_scores("s_full", "f", query, train, **kw)["all_modules"]
    assert rel(sub["all_modules"], full[[0, 2, 4]][:, 10:30]) <= 1e-6
    # a different dataset into the same (incomplete) factor directory is refused
    import os
    os.remove(analyzer.factors_output_dir("f") / "lambda_matrix.safetensors")
    shorter = torch.utils.data.TensorDataset(*fx.make_data("mlp", 20, seed=1))
    with pytest.raises(ValueError, match="differs from the one already saved"):
        analyzer.fit_lambda_matrices("f", shorter, per_device_batch_size=10, factor_args=args)
    analyzer.fit_lambda_matrices("f", shorter, per_device_batch_size=10, factor_args=args, overwrite_output_dir=True)


def test_gpt2_shaped_workload_host_logic_matches_oracle(tmp_path, cpu_engine):
    """bench.py's GPT-2-shaped decoder (tiny instance): only the block Linears are tracked (Task.get_influence_tracked_modules),
    sequence activations with bias, causal attention in between -- scores against the CPU oracle on the same weights."""
    import bench
    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from oracle import ekfac_ref as ref
    from torch.utils import data

    torch.manual_seed(0)
    raw = bench.GPT2(layers=2, width=16, heads=2, vocab=40, positions=8)
    twin = bench.GPT2(layers=2, width=16, heads=2, vocab=40, positions=8)
    twin.load_state_dict(raw.state_dict())
    names = raw.tracked_names()
    task = bench.make_lm_task(names)
    spec = dict(vocab=40, tokens=8)
    train, query = bench.synth_tokens(spec, 24, 1, "cpu"), bench.synth_tokens(spec, 4, 2, "cpu")
    analyzer = Analyzer("t", prepare_model(raw, task), task, output_dir=str(tmp_path), disable_tqdm=True)
    from kronfluence_amd.module.utils import get_tracked_module_names
    assert get_tracked_module_names(analyzer.model) == names and len(names) == 8
    args = FactorArguments(use_empirical_fisher=True)
    analyzer.fit_all_factors("f", data.TensorDataset(*train), per_device_batch_size=8, factor_args=args)
    got = analyzer.compute_pairwise_scores("s", "f", data.TensorDataset(*query), data.TensorDataset(*train),
                                           per_device_query_batch_size=2, per_device_train_batch_size=6,
                                           score_args=ScoreArguments(damping_factor=None))["all_modules"]
    engine = ref.OracleEngine(twin.double(), module_names=names)
    chunks = lambda d, bs: [tuple(t[i:i + bs] for t in d) for i in range(0, d[0].shape[0], bs)]  # noqa: E731
    cov = engine.fit_covariance(chunks(train, 8), bench.lm_loss)
    eig = engine.eigendecomposition(cov)
    lam = engine.fit_lambda(chunks(train, 8), bench.lm_loss, eig)
    want = engine.pairwise_scores(chunks(query, 2), chunks(train, 6), bench.lm_loss, bench.lm_loss, eig, lam, None)
    mine = analyzer.load_covariance_matrices("f")
    for module in names:
        assert rel(mine["activation_covariance"][module], cov["activation_covariance"][module]) <= 1e-5, module
    assert got.shape == (4, 24) and rel(got, want) <= 2e-3, rel(got, want)


def test_bert_shaped_workload_host_logic_matches_oracle(tmp_path, cpu_engine):
    """bench.py's BERT-shaped classifier (tiny instance): every Linear tracked, ``[b, T, d]`` activations with random-length
    padding masks (``Task.get_attention_mask``) for the encoder layers, one row per sample for pooler / classifier (the
    mask then does not match the row count and is ignored, reference linear.py:33) -- factors and scores against the CPU
    oracle on the same weights."""
    import bench
    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from kronfluence_amd.module.utils import get_tracked_module_names
    from oracle import ekfac_ref as ref
    from torch.utils import data

    torch.manual_seed(0)
    kw = dict(layers=2, width=16, heads=2, inter=24, vocab=40, positions=8)
    raw, twin = bench.Bert(**kw), bench.Bert(**kw)
    twin.load_state_dict(raw.state_dict())
    task = bench.make_glue_task()
    spec = dict(vocab=40, tokens=8)
    train, query = bench.synth_glue(spec, 24, 1, "cpu"), bench.synth_glue(spec, 4, 2, "cpu")
    assert int(train[1].sum()) < train[1].numel()  # there IS padding
    analyzer = Analyzer("t", prepare_model(raw, task), task, output_dir=str(tmp_path), disable_tqdm=True)
    names = get_tracked_module_names(analyzer.model)
    assert len(names) == 2 * 6 + 2 and names[-2:] == ["pooler", "classifier"]
    args = FactorArguments(use_empirical_fisher=True)
    analyzer.fit_all_factors("f", data.TensorDataset(*train), per_device_batch_size=8, factor_args=args)
    got = analyzer.compute_pairwise_scores("s", "f", data.TensorDataset(*query), data.TensorDataset(*train),
                                           per_device_query_batch_size=2, per_device_train_batch_size=6,
                                           score_args=ScoreArguments(damping_factor=None))["all_modules"]
    engine = ref.OracleEngine(twin.double())
    chunks = lambda d, bs: [tuple(t[i:i + bs] for t in d) for i in range(0, d[0].shape[0], bs)]  # noqa: E731
    cov = engine.fit_covariance(chunks(train, 8), bench.glue_loss, lambda batch: batch[1])
    eig = engine.eigendecomposition(cov)
    lam = engine.fit_lambda(chunks(train, 8), bench.glue_loss, eig)
    want = engine.pairwise_scores(chunks(query, 2), chunks(train, 6), bench.glue_margin, bench.glue_loss, eig, lam, None)
    mine = analyzer.load_covariance_matrices("f")
    for module in names:
        assert rel(mine["activation_covariance"][module], cov["activation_covariance"][module]) <= 1e-5, module
        assert torch.equal(mine["num_activation_covariance_processed"][module].reshape(-1),
                           cov["num_activation_covariance_processed"][module].reshape(-1)), module
    tokens = int(train[1].sum())
    assert int(mine["num_activation_covariance_processed"]["layers.0.query"]) == tokens  # masked token count
    assert int(mine["num_activation_covariance_processed"]["pooler"]) == 24            # one row per sample
    assert got.shape == (4, 24) and rel(got, want) <= 2e-3, rel(got, want)


def test_profile_flag_writes_stage_timing_summaries(tmp_path, cpu_engine):
    import fixtures as fx
    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from test_pipeline_gpu import make_task
    from torch.utils import data

    task = make_task("mlp")
    analyzer = Analyzer("t", prepare_model(fx.make_model("mlp"), task), task, output_dir=str(tmp_path), profile=True)
    train = data.TensorDataset(*fx.make_data("mlp", 24, seed=1))
    analyzer.fit_all_factors("f", train, per_device_batch_size=8, factor_args=FactorArguments(use_empirical_fisher=True))
    analyzer.compute_self_scores("s", "f", train, per_device_train_batch_size=8, score_args=ScoreArguments(damping_factor=None))
    out = analyzer.output_dir / "profiler_output"
    text = (out / "factors_f_summary_rank_0.txt").read_text()
    assert all(stage in text for stage in ("fit_covariance", "perform_eigendecomposition", "fit_lambda"))
    assert "compute_self_scores" in (out / "scores_s_self_summary_rank_0.txt").read_text()


def test_in_place_write_to_a_hooked_activation_is_detected(tmp_path, cpu_engine):
    """The hooks hold the layer input by reference (no clone, as the reference); with all parameters frozen autograd
    would not notice a later in-place write to it, the tracker does."""
    import torch
    from torch import nn

    from kronfluence_amd import FactorArguments, prepare_model
    from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
    from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader, perform_eigendecomposition
    from kronfluence_amd.utils.state import State
    from test_pipeline_gpu import make_task

    class Clobber(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Linear(12, 16), nn.Linear(16, 3)

        def forward(self, x):
            h = self.a(x) + 1.0  # an op that does not save its output: autograd will not notice the write below
            out = self.b(h)
            h.mul_(2.0)  # in-place write to the input of `b` after `b` has run
            return out

    task = make_task("mlp")
    model = prepare_model(Clobber(), task)
    state, fargs = State(), FactorArguments(use_empirical_fisher=True)
    batches = fx.batches(fx.make_data("mlp", 16, seed=1), 8)
    _, cov = fit_covariance_matrices_with_loader(model, state, task, batches, fargs)  # covariance consumes inputs at once
    eig = perform_eigendecomposition(cov, model, state, fargs)
    with pytest.raises(RuntimeError, match="modified in place"):
        fit_lambda_matrices_with_loader(model, state, task, batches, fargs, eig)


def test_model_save_guards_the_output_directory(tmp_path, cpu_engine):
    """``Analyzer(disable_model_save=False)`` (reference analyzer.py:107-143): the first Analyzer of a directory stores the model, a
    later one with the same model is accepted, with other weights refused; the default stores nothing."""
    from kronfluence_amd import Analyzer, prepare_model
    from test_pipeline_gpu import make_task

    kind = "mlp"
    task = make_task(kind)

    def analyzer(name, seed, **kw):
        return Analyzer(name, prepare_model(fx.make_model(kind, seed=seed), task), task, output_dir=str(tmp_path), disable_tqdm=True, **kw)

    first = analyzer("guarded", 0, disable_model_save=False)
    saved = first.output_dir / "model.safetensors"
    assert saved.exists()
    stamp = saved.stat().st_mtime_ns
    analyzer("guarded", 0, disable_model_save=False)                   # the same weights: accepted, file untouched
    assert saved.stat().st_mtime_ns == stamp
    with pytest.raises(ValueError, match="different `analysis_name`"):
        analyzer("guarded", 1, disable_model_save=False)               # other weights under the same name
    analyzer("guarded", 1)                                             # the default does not look
    assert not (analyzer("plain", 0).output_dir / "model.safetensors").exists()


@pytest.mark.parametrize("with_measurement", [False, True])
def test_self_scores_of_shared_modules_match_autograd(tmp_path, cpu_engine, with_measurement):
    """A Linear used three times per forward, identity strategy (so that the expected value needs no factors): the self-influence
    score of sample i is ``sum over parameters <grad m_i, grad L_i>`` with every use of the shared weight in both gradients
    (``m = L`` without measurement).  The reference is exact without measurement and 13 % off with it."""
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from test_pipeline_gpu import make_task

    kind, n = "shared", 12
    task = make_task(kind)
    analyzer = Analyzer("t", prepare_model(fx.make_model(kind).double(), task), task, output_dir=str(tmp_path), disable_tqdm=True)
    train_t = fx.make_data(kind, n, seed=1)
    analyzer.fit_all_factors("f", data.TensorDataset(*train_t), per_device_batch_size=4,
                             factor_args=FactorArguments(strategy="identity", has_shared_parameters=True))
    got = analyzer.compute_self_scores("s", "f", data.TensorDataset(*train_t), per_device_train_batch_size=5,
                                       score_args=ScoreArguments(use_measurement_for_self_influence=with_measurement))["all_modules"]
    model = fx.make_model(kind).double()
    loss, measure = fx.train_loss(kind), fx.measurement(kind)
    params = [p for _, p in model.named_parameters()]
    want = []
    for i in range(n):
        batch = tuple(t[i:i + 1] for t in train_t)
        g_loss = torch.autograd.grad(loss(model, batch), params, allow_unused=True)
        g_meas = torch.autograd.grad((measure if with_measurement else loss)(model, batch), params, allow_unused=True)
        want.append(sum((a * b).sum() for a, b in zip(g_loss, g_meas) if a is not None and b is not None))
    assert rel(got.flatten().double(), torch.stack(want)) <= 1e-6


def test_missing_results_raise_what_the_reference_raises(tmp_path, cpu_engine):
    """Error behaviour compared with the reference side by side in the build container (not in this test): an unknown strategy is
    a ``KeyError``; ``load_all_factors`` of an unknown name a ``FileNotFoundError``, of a name whose strategy lacks a factor group
    a ``FactorsNotFoundError``; every ``aggregate_*`` of an unknown name a ``ValueError``; ``load_*`` of an unknown name ``None``."""
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, prepare_model
    from kronfluence_amd.factor.config import FactorConfig
    from kronfluence_amd.utils.exceptions import FactorsNotFoundError
    from test_pipeline_gpu import make_task

    kind = "mlp"
    task = make_task(kind)
    analyzer = Analyzer("t", prepare_model(fx.make_model(kind), task), task, output_dir=str(tmp_path), disable_tqdm=True)
    with pytest.raises(KeyError):
        FactorConfig.CONFIGS["no-such-strategy"]
    with pytest.raises(NotImplementedError, match="no-such-strategy"):
        FactorConfig.CONFIGS["no-such-strategy"]
    with pytest.raises(FileNotFoundError):
        analyzer.load_all_factors("missing")
    for aggregate in (analyzer.aggregate_pairwise_scores, analyzer.aggregate_self_scores, analyzer.aggregate_covariance_matrices,
                      analyzer.aggregate_lambda_matrices):
        with pytest.raises(ValueError):
            aggregate("missing")
    for load in (analyzer.load_pairwise_scores, analyzer.load_self_scores, analyzer.load_covariance_matrices,
                 analyzer.load_eigendecomposition, analyzer.load_lambda_matrices, analyzer.load_factor_args, analyzer.load_score_args):
        assert load("missing") is None
    train = data.TensorDataset(*fx.make_data(kind, 12, seed=1))
    analyzer.fit_covariance_matrices("partial", train, per_device_batch_size=4, factor_args=FactorArguments(use_empirical_fisher=True))
    with pytest.raises(FactorsNotFoundError, match="Eigendecomposition"):
        analyzer.load_all_factors("partial")                      # ekfac needs eigenvectors and Lambda too


def test_borrowed_factors_and_score_directories_have_the_references_files(tmp_path, cpu_engine):
    """File layout compared with the reference side by side in the build container: ``load_from_factors_name`` makes the borrowed
    covariances / eigendecomposition part of the new name (files + ``factor_loaded_*_arguments.json``), and a scores directory
    records the factor arguments it was computed with next to its score arguments."""
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, prepare_model
    from test_pipeline_gpu import make_task

    kind = "mlp"
    task = make_task(kind)
    analyzer = Analyzer("t", prepare_model(fx.make_model(kind), task), task, output_dir=str(tmp_path), disable_tqdm=True)
    train = data.TensorDataset(*fx.make_data(kind, 20, seed=1))
    query = data.TensorDataset(*fx.make_data(kind, 4, seed=2))
    fargs = FactorArguments(use_empirical_fisher=True)
    analyzer.fit_all_factors("f", train, per_device_batch_size=4, factor_args=fargs)
    analyzer.perform_eigendecomposition("g", factor_args=fargs, load_from_factors_name="f")
    analyzer.fit_lambda_matrices("h", train, per_device_batch_size=4, factor_args=fargs, load_from_factors_name="g")
    g, h = set(os.listdir(analyzer.factors_output_dir("g"))), set(os.listdir(analyzer.factors_output_dir("h")))
    assert {"activation_covariance.safetensors", "gradient_covariance.safetensors", "factor_loaded_covariance_arguments.json",
            "activation_eigenvectors.safetensors", "factor_arguments.json"} <= g
    assert {"activation_eigenvectors.safetensors", "gradient_eigenvalues.safetensors", "factor_loaded_eigendecomposition_arguments.json",
            "lambda_matrix.safetensors"} <= h
    want, got = analyzer.load_covariance_matrices("f"), analyzer.load_covariance_matrices("g")
    assert all(torch.equal(got[k][m], want[k][m]) for k in want for m in want[k])
    assert all(torch.equal(analyzer.load_lambda_matrices("h")[k][m], v) for k, d in analyzer.load_lambda_matrices("f").items() for m, v in d.items())
    analyzer.compute_pairwise_scores("s", "h", query, train, per_device_query_batch_size=2, per_device_train_batch_size=4)
    assert {"score_arguments.json", "factor_arguments.json", "pairwise_scores.safetensors"} <= set(os.listdir(analyzer.scores_output_dir("s")))


@pytest.mark.parametrize("deduplicate", [True, False])
def test_layers_that_share_an_input_share_one_eigendecomposition(tmp_path, cpu_engine, monkeypatch, deduplicate):
    """Query / key / value-like projections consume the same tensor, so their activation covariances are the same matrix (up to
    the order of the accumulation's atomics): ``perform_eigendecomposition`` solves it once and hands every layer its own copy --
    what the reference gets from three bit-identical problems (factor/eigen.py:140-224).  Gradient covariances stay apart."""
    from torch import nn
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, Task, ops, prepare_model
    from kronfluence_amd.factor import eigen

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v, self.o = nn.Linear(6, 5), nn.Linear(6, 5), nn.Linear(6, 5), nn.Linear(5, 3)

        def forward(self, x):
            return self.o(torch.tanh(self.q(x)) * torch.tanh(self.k(x)) + self.v(x))

    class T(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return (model(batch[0]) - batch[1]).square().sum()

        def compute_measurement(self, batch, model):
            return self.compute_train_loss(batch, model)

    monkeypatch.setattr(eigen, "DEDUPLICATE_COVARIANCES", deduplicate)
    solved = []
    real = ops.eigh
    monkeypatch.setattr(ops, "eigh", lambda cov, *a, **k: (solved.append(tuple(cov.shape)), real(cov, *a, **k))[1])
    torch.manual_seed(0)
    task = T()
    model = prepare_model(Attn(), task)
    analyzer = Analyzer("t", model, task, output_dir=str(tmp_path), disable_tqdm=True)
    gen = torch.Generator().manual_seed(1)
    train = data.TensorDataset(torch.randn(40, 6, generator=gen), torch.randn(40, 3, generator=gen))
    analyzer.fit_all_factors("f", train, per_device_batch_size=8, factor_args=FactorArguments(use_empirical_fisher=True))
    # 4 layers x 2 sides = 8 problems; q, k, v share their 7 x 7 activation covariance
    assert len(solved) == (6 if deduplicate else 8) and solved.count((7, 7)) == (1 if deduplicate else 3)
    eig = analyzer.load_eigendecomposition("f")
    for name in ("k", "v"):
        assert torch.equal(eig["activation_eigenvectors"][name], eig["activation_eigenvectors"]["q"])
        assert torch.equal(eig["activation_eigenvalues"][name], eig["activation_eigenvalues"]["q"])
        assert not torch.equal(eig["gradient_eigenvalues"][name], eig["gradient_eigenvalues"]["q"])


def test_layers_that_share_an_input_share_its_covariance_increment(tmp_path, cpu_engine, monkeypatch):
    """CovarianceTracker: the query / key / value-like projections of the model above are handed the same tensor object, so from
    the second batch on the first of them forms the increment ``X'^T X'`` once and the other two add it (reference
    tracker/factor.py:98-112 computes three identical ``addmm_``).  Kernel calls are counted; the factors equal the un-shared run's."""
    from torch import nn
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, Task, ops, prepare_model
    from kronfluence_amd.module.tracker.factor import CovarianceTracker

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v, self.o = nn.Linear(6, 5), nn.Linear(6, 5), nn.Linear(6, 5, bias=False), nn.Linear(5, 3)

        def forward(self, x):
            return self.o(torch.tanh(self.q(x)) * torch.tanh(self.k(x)) + self.v(x))

    class T(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return (model(batch[0]) - batch[1]).square().sum()

        def compute_measurement(self, batch, model):
            return self.compute_train_loss(batch, model)

    gen = torch.Generator().manual_seed(1)
    train = data.TensorDataset(torch.randn(40, 6, generator=gen), torch.randn(40, 3, generator=gen))
    results = {}
    for share in (True, False):
        monkeypatch.setattr(CovarianceTracker, "SHARE_INPUT_INCREMENTS", share)
        calls = []
        real = ops.linear_activation_cov
        monkeypatch.setattr(ops, "linear_activation_cov", lambda cov, *a, **k: (calls.append(tuple(cov.shape)), real(cov, *a, **k))[1])
        torch.manual_seed(0)
        task = T()
        analyzer = Analyzer("t", prepare_model(Attn(), task), task, output_dir=str(tmp_path / str(share)), disable_tqdm=True)
        analyzer.fit_covariance_matrices("f", train, per_device_batch_size=8, factor_args=FactorArguments(use_empirical_fisher=True))
        monkeypatch.setattr(ops, "linear_activation_cov", real)
        results[share] = (calls, analyzer.load_covariance_matrices("f"))
    shared_calls, plain_calls = results[True][0], results[False][0]
    # 5 batches x 4 layers un-shared; shared: the first batch in full, then q (7 x 7, leader), v's bias-free 6 x 6 and o per batch --
    # k follows q; v has no bias column, so its increment is a different matrix and is never shared
    assert len(plain_calls) == 20 and len(shared_calls) == 4 + 4 * 3, (len(plain_calls), len(shared_calls))
    for name, want in results[False][1]["activation_covariance"].items():
        got = results[True][1]["activation_covariance"][name]
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()), name
    for key in ("num_activation_covariance_processed", "gradient_covariance", "num_gradient_covariance_processed"):
        for name, want in results[False][1][key].items():
            assert torch.equal(results[True][1][key][name], want), (key, name)
