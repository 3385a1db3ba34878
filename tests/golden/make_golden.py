"""Golden-vector generator: runs the REAL reference (kronfluence v1.0.1) on CPU, here only.

Usage (build container; the reference is mounted read-only at /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

For each fixture in ``tests/fixtures.py`` and for the fp64 ("pytest" preset) and fp32 (reference
defaults) dtypes it drives ``prepare_model`` / ``Analyzer.fit_all_factors`` /
``Analyzer.compute_pairwise_scores`` of the reference and stores, as
``tests/golden/<fixture>_<dtype>.safetensors``:

  * ``cov/<factor>/<module>``   activation/gradient covariance + the two counters  (A6)
  * ``eig/<factor>/<module>``   eigenvalues and eigenvectors                        (E1)
  * ``lam/<factor>/<module>``   Lambda matrix + counter                             (L4)
  * ``scores/damp1e-8`` and ``scores/dampNone``  pairwise ``all_modules`` scores   (S3)

The two missing third-party imports (``opt_einsum``, ``einconv``) are satisfied by the
no-arithmetic stand-ins in ``tests/golden/_shims``.  Nothing from the
reference is copied into the repository: only inputs-free output tensors are committed (the
inputs are regenerated from seeds by ``tests/fixtures.py``).
"""

import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(HERE))  # tests/
sys.dont_write_bytecode = True

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402
from torch.utils import data  # noqa: E402

from kronfluence.analyzer import Analyzer, prepare_model  # noqa: E402
from kronfluence.arguments import FactorArguments, ScoreArguments  # noqa: E402
from kronfluence.task import Task  # noqa: E402

import fixtures as fx  # noqa: E402


def make_task(kind: str) -> Task:
    loss, measure, mask = fx.train_loss(kind), fx.measurement(kind), fx.attention_mask(kind)

    class GoldenTask(Task):
        def compute_train_loss(self, batch, model, sample=False):
            assert not sample
            return loss(model, tuple(batch))

        def compute_measurement(self, batch, model):
            return measure(model, tuple(batch))

        def get_attention_mask(self, batch):
            return None if mask is None else mask(tuple(batch))

    return GoldenTask()


def run(kind: str, dtype: torch.dtype, out_path: str) -> None:
    spec = fx.FIXTURES.get(kind) or fx.MSE_FIXTURES.get(kind) or {"conv8": fx.BF16_FIXTURE, "shared": fx.SHARED_FIXTURE}[kind]
    model = fx.make_model(kind).to(dtype=dtype)
    train = data.TensorDataset(*fx.make_data(kind, spec.n_train, seed=1))
    query = data.TensorDataset(*fx.make_data(kind, spec.n_query, seed=2))
    task = make_task(kind)
    model = prepare_model(model, task)
    with tempfile.TemporaryDirectory() as tmp:
        analyzer = Analyzer("golden", model, task, cpu=True, disable_tqdm=True, output_dir=tmp)
        fargs = FactorArguments(
            use_empirical_fisher=True, has_shared_parameters=(kind == "shared"),
            activation_covariance_dtype=dtype, gradient_covariance_dtype=dtype,
            per_sample_gradient_dtype=dtype, lambda_dtype=dtype,
        )
        analyzer.fit_all_factors("f", train, per_device_batch_size=spec.factor_batch, factor_args=fargs,
                                 overwrite_output_dir=True)
        factors = analyzer.load_all_factors("f")
        # covariance factors are not part of load_all_factors for ekfac
        cov = analyzer.load_covariance_matrices("f")
        out = {}
        for name, per_module in cov.items():
            for module, tensor in per_module.items():
                out[f"cov/{name}/{module}"] = tensor.contiguous()
        for name, per_module in factors.items():
            group = "lam" if "lambda" in name else "eig"
            for module, tensor in per_module.items():
                out[f"{group}/{name}/{module}"] = tensor.contiguous()
        for tag, damping in (("damp1e-8", 1e-8), ("dampNone", None)):
            sargs = ScoreArguments(damping_factor=damping, per_sample_gradient_dtype=dtype,
                                   precondition_dtype=dtype, score_dtype=dtype)
            analyzer.compute_pairwise_scores(
                f"s_{tag}", "f", query, train,
                per_device_query_batch_size=spec.query_batch, per_device_train_batch_size=spec.train_batch,
                score_args=sargs, overwrite_output_dir=True,
            )
            out[f"scores/{tag}"] = analyzer.load_pairwise_scores(f"s_{tag}")["all_modules"].contiguous()
    save_file(out, out_path)
    print(f"{out_path}: {len(out)} tensors, scores {tuple(out['scores/damp1e-8'].shape)}")


def run_widen(kind: str, out_path: str) -> None:
    """Goldens for the SURVEY.md 8(f) rows, fp64: self-influence (both variants), the identity / diagonal / kfac
    strategies, per-module scores, query / train gradient aggregation and (sequence fixture) per-token scores."""
    dtype = torch.float64
    spec = fx.FIXTURES[kind]
    train = data.TensorDataset(*fx.make_data(kind, spec.n_train, seed=1))
    query = data.TensorDataset(*fx.make_data(kind, spec.n_query, seed=2))
    task = make_task(kind)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for strategy in ("ekfac", "kfac", "diagonal", "identity"):
            model = prepare_model(fx.make_model(kind).to(dtype=dtype), task)
            analyzer = Analyzer(f"widen_{strategy}", model, task, cpu=True, disable_tqdm=True, output_dir=tmp)
            fargs = FactorArguments(strategy=strategy, use_empirical_fisher=True, activation_covariance_dtype=dtype,
                                    gradient_covariance_dtype=dtype, per_sample_gradient_dtype=dtype, lambda_dtype=dtype)
            analyzer.fit_all_factors("f", train, per_device_batch_size=spec.factor_batch, factor_args=fargs,
                                     overwrite_output_dir=True)
            if strategy == "diagonal":
                for name, per_module in analyzer.load_lambda_matrices("f").items():
                    for module, tensor in per_module.items():
                        out[f"strategy/diagonal/lam/{name}/{module}"] = tensor.contiguous()

            def sargs(**kw):
                return ScoreArguments(damping_factor=None, per_sample_gradient_dtype=dtype, precondition_dtype=dtype,
                                      score_dtype=dtype, **kw)

            batch = dict(per_device_query_batch_size=spec.query_batch, per_device_train_batch_size=spec.train_batch)
            analyzer.compute_pairwise_scores("s", "f", query, train, score_args=sargs(), overwrite_output_dir=True, **batch)
            out[f"strategy/{strategy}/scores"] = analyzer.load_pairwise_scores("s")["all_modules"].contiguous()
            analyzer.compute_self_scores("self", "f", train, per_device_train_batch_size=spec.train_batch,
                                         score_args=sargs(), overwrite_output_dir=True)
            out[f"strategy/{strategy}/self"] = analyzer.load_self_scores("self")["all_modules"].contiguous()
            if strategy != "ekfac":
                continue
            analyzer.compute_self_scores("selfm", "f", train, per_device_train_batch_size=spec.train_batch,
                                         score_args=sargs(use_measurement_for_self_influence=True), overwrite_output_dir=True)
            out["self_measurement"] = analyzer.load_self_scores("selfm")["all_modules"].contiguous()
            analyzer.compute_pairwise_scores("pm", "f", query, train, score_args=sargs(compute_per_module_scores=True),
                                             overwrite_output_dir=True, **batch)
            for module, tensor in analyzer.load_pairwise_scores("pm").items():
                out[f"permodule/{module}"] = tensor.contiguous()
            analyzer.compute_self_scores("selfpm", "f", train, per_device_train_batch_size=spec.train_batch,
                                         score_args=sargs(compute_per_module_scores=True), overwrite_output_dir=True)
            for module, tensor in analyzer.load_self_scores("selfpm").items():
                out[f"self_permodule/{module}"] = tensor.contiguous()
            for tag, kw in (("aggq", dict(aggregate_query_gradients=True)), ("aggt", dict(aggregate_train_gradients=True)),
                            ("aggqt", dict(aggregate_query_gradients=True, aggregate_train_gradients=True))):
                analyzer.compute_pairwise_scores(tag, "f", query, train, score_args=sargs(**kw), overwrite_output_dir=True, **batch)
                out[tag] = analyzer.load_pairwise_scores(tag)["all_modules"].contiguous()
            rank = 4
            analyzer.compute_pairwise_scores("lowrank", "f", query, train, overwrite_output_dir=True, **batch,
                                             score_args=sargs(query_gradient_low_rank=rank, use_full_svd=True,
                                                              query_gradient_svd_dtype=dtype))
            out[f"lowrank{rank}"] = analyzer.load_pairwise_scores("lowrank")["all_modules"].contiguous()
            if kind == "seq":
                analyzer.compute_pairwise_scores("tok", "f", query, train, score_args=sargs(compute_per_token_scores=True),
                                                 overwrite_output_dir=True, **batch)
                out["pertoken"] = analyzer.load_pairwise_scores("tok")["all_modules"].contiguous()
    save_file(out, out_path)
    print(f"{out_path}: {len(out)} tensors: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in out.items() if "/" not in k or k.endswith(("scores", "self"))))


def run_presets(out_path: str) -> None:
    """Field values of every argument preset of ``kronfluence/utils/common`` (plus the dataclass defaults)."""
    import dataclasses
    import json

    from kronfluence.utils.common import factor_arguments as rf, score_arguments as rs

    def plain(obj):
        return {k: (str(v) if isinstance(v, torch.dtype) else v) for k, v in dataclasses.asdict(obj).items()}

    out = {"FactorArguments()": plain(FactorArguments()), "ScoreArguments()": plain(ScoreArguments())}
    for module, tag in ((rf, "factor"), (rs, "score")):
        for name in sorted(dir(module)):
            if name.endswith("_arguments"):
                out[f"{tag}/{name}()"] = plain(getattr(module, name)())
    for name in sorted(dir(rs)):
        if name.endswith("_arguments"):
            out[f"score/{name}(query_gradient_low_rank=32)"] = plain(getattr(rs, name)(query_gradient_low_rank=32))
    out["factor/extreme_reduce_memory_factor_arguments(module_partitions=3)"] = plain(
        rf.extreme_reduce_memory_factor_arguments(module_partitions=3))
    with open(out_path, "w", encoding="utf-8") as handle:
        json.dump(out, handle, indent=1, sort_keys=True)
    print(f"{out_path}: {len(out)} presets")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "presets":
        run_presets(os.path.join(HERE, "presets.json"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mse":  # well-conditioned fixtures for the default damping 1e-8
        torch.manual_seed(0)
        for kind in fx.MSE_FIXTURES:
            for tag, dtype in (("fp64", torch.float64), ("fp32", torch.float32)):
                run(kind, dtype, os.path.join(HERE, f"{kind}_{tag}.safetensors"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "widen":
        torch.manual_seed(0)
        for kind in fx.FIXTURES:
            run_widen(kind, os.path.join(HERE, f"widen_{kind}_fp64.safetensors"))
        sys.exit(0)
    torch.manual_seed(0)
    for kind in fx.FIXTURES:
        for tag, dtype in (("fp64", torch.float64), ("fp32", torch.float32)):
            run(kind, dtype, os.path.join(HERE, f"{kind}_{tag}.safetensors"))
    run("conv8", torch.float32, os.path.join(HERE, "conv8_fp32.safetensors"))
    run("shared", torch.float64, os.path.join(HERE, "shared_fp64.safetensors"))
