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
