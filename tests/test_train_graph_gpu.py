"""The train-side step of the pairwise stage as a hipGraph (score/dot_product.py: ``KF_TRAIN_GRAPH``): forward, backward and every
hook launch of one train batch captured once and replayed for the remaining full batches.  Scores must be those of the eager loop
(same kernels, same inputs; only the launch mechanism differs) -- compared here per fixture, plus the bookkeeping: one capture, the
expected number of replays, no fall-back, a ragged last batch handled eagerly."""

import pytest
import torch
from torch.utils import data

import fixtures as fx
from test_pipeline_gpu import make_task, rel

pytestmark = pytest.mark.gpu


def _scores(kind, tmp_path, mode, monkeypatch, n_train, batch, score_kw=None, bf16=False):
    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from kronfluence_amd.score import dot_product

    monkeypatch.setenv("KF_TRAIN_GRAPH", mode)
    for key in ("captures", "replays", "fallbacks"):
        dot_product.GRAPH_LOG[key] = 0
    dot_product.GRAPH_LOG["last_error"] = None
    torch.manual_seed(0)
    task = make_task(kind)
    model = prepare_model(fx.make_model(kind), task)
    analyzer = Analyzer("t", model, task, output_dir=str(tmp_path / mode), disable_tqdm=True)
    spec = fx.spec_of(kind)
    train = data.TensorDataset(*fx.make_data(kind, n_train, seed=1))
    query = data.TensorDataset(*fx.make_data(kind, spec.n_query, seed=2))
    low = dict(amp_dtype=torch.bfloat16) if bf16 else {}
    fargs = FactorArguments(use_empirical_fisher=True, **low,
                            **(dict(per_sample_gradient_dtype=torch.bfloat16, lambda_dtype=torch.bfloat16) if bf16 else {}))
    analyzer.fit_all_factors("f", train, per_device_batch_size=batch, factor_args=fargs)
    sargs = ScoreArguments(damping_factor=None, query_gradient_accumulation_steps=64, **low,   # all queries held: ONE train pass
                           **(dict(score_dtype=torch.bfloat16, precondition_dtype=torch.bfloat16) if bf16 else {}), **(score_kw or {}))
    out = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=4, per_device_train_batch_size=batch,
                                           score_args=sargs, dataloader_kwargs=None)
    return out["all_modules"], dict(dot_product.GRAPH_LOG)


@pytest.mark.parametrize("kind", ["mlp", "conv", "seq"])
@pytest.mark.parametrize("bf16", [False, True])
def test_graph_replay_matches_the_eager_loop(kind, bf16, tmp_path, monkeypatch):
    spec = fx.spec_of(kind)
    batch = max(2, spec.n_train // 9)
    n_train = 8 * batch + 3                      # 8 full batches + a ragged ninth
    want, log0 = _scores(kind, tmp_path, "0", monkeypatch, n_train, batch, bf16=bf16)
    assert log0["captures"] == 0 and log0["replays"] == 0
    got, log1 = _scores(kind, tmp_path, "1", monkeypatch, n_train, batch, bf16=bf16)
    assert log1["fallbacks"] == 0, log1["last_error"]
    assert log1["captures"] == 1 and log1["replays"] == 7          # batch 0 eager, batches 1..7 replayed, the ragged ninth eager
    assert got.shape == want.shape == (spec.n_query, n_train)
    # same kernels on the same inputs: only split-K atomics may reorder additions
    assert rel(got, want) <= (2e-3 if bf16 else 2e-6), rel(got, want)


def test_graph_mode_steps_aside_for_stateful_options(tmp_path, monkeypatch):
    """Per-token scores keep Python-side state per batch: the pass runs eagerly and says so by not capturing."""
    kind = "seq"
    spec = fx.spec_of(kind)
    batch = max(2, spec.n_train // 9)
    got, log = _scores(kind, tmp_path, "1", monkeypatch, 8 * batch, batch, score_kw=dict(compute_per_token_scores=True))
    assert log["captures"] == 0 and log["fallbacks"] == 0
    assert got.dim() == 3


def test_auto_mode_needs_enough_batches(tmp_path, monkeypatch):
    from kronfluence_amd.score import dot_product

    kind = "mlp"
    spec = fx.spec_of(kind)
    batch = max(2, spec.n_train // 3)
    _, log = _scores(kind, tmp_path, "auto", monkeypatch, 3 * batch, batch)
    assert log["captures"] == 0                                     # 3 batches < GRAPH_MIN_BATCHES: not worth a capture
    assert dot_product.GRAPH_MIN_BATCHES > 3
