"""The measurement contract of ``bench.py``: the LAST stdout line is ONE compact JSON object the driver can parse (VERDICT r05
item 1: the 30 KB line of round 5 came back ``parsed: null``).  Built here from canned result dicts -- the full object of a kept
round-5 run and a synthetic worst case stuffed with prose -- without a GPU."""

import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _roofline(prose: int = 400) -> dict:
    return {"bound": "mfma", "kernel": "k" * prose, "achieved": 812.0415083611956, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": 0.32481660334447826, "traffic": 2279587336.9411764, "launches": 9000, "avg_launch_ms": 0.7015239419837793,
            "algorithmic_flops_per_launch": 569666560000.0, "algorithmic_bytes_per_launch": 620285333.3333334,
            "algorithmic_GBps": 884.19, "hbm_frac_of_8TBps": 0.1105, "kernel_share_of_region": 0.3966, "model_share_of_region": 0.569,
            "traffic_source": "s" * prose, "mfma_util": 0.7054737358790268, "traffic_over_algorithmic": 3.67,
            "per_kernel_mfma_util": {f"kernel_{i}<{i}>": 0.1 * i for i in range(12)}}


def _result(workload: str, prose: int = 400) -> dict:
    busy = {"n_train": 2048, "wall_s": 2.33, "kf_kernel_s": 0.53, "model_kernel_s": 1.78, "idle_frac": 0.0012, "kf_kernel_frac": 0.229,
            "model_kernel_frac": 0.764, "method": "m" * prose,
            "largest_idle_after": [{"after": "a" * 70, "gaps": 7, "seconds": 3.6e-4} for _ in range(6)]}
    return {"metric": "pairwise_influence_pairs_per_sec", "value": 62820573.71743225, "unit": "pairs/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 795.9175958007108, "step_ms": [795.91] * 20, "hipmalloc_segments_in_timed_region": 0, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload, "n_train": 50000, "n_query": 1000, "tracked_layers": 9, "D": 2272192, "train_batch": 1000,
                       "parallelism": "train-shard-dp1", "scaled_from": {"n_train": 100000, "n_query": 2000}},
            "roofline": _roofline(prose), "roofline_cov": _roofline(prose), "roofline_cov_f32": _roofline(prose),
            "roofline_lambda": _roofline(prose), "roofline_lambda_update": _roofline(prose),
            "factor_fit": {"samples_per_sec": 194.4, "seconds": {"covariance": 2.03, "eigendecomposition": 5.93, "lambda": 2.56}, "n_fit": 2048,
                           "eigen_dims": list(range(40)), "eigen_sum_d3": 8.7e11, "eigh_paths": {"factor_first": 96},
                           "covariance_samples_per_sec": 1008.7, "lambda_samples_per_sec": 798.2},
            "peak_hbm_gib": 249.7, "device_busy": busy,
            "parity": {"scores_rel_F_vs_fp64_low_rank_contraction": 4.87e-3, "bound": 2e-2, "ok": True, "queries": 8, "train_samples": 8, "what": "w" * prose},
            "exchanges": {"backend": "nccl", "ranks": 8, "query_exchange": "replicate",
                          "factor_fit": {"factor_all_reduce": {"calls": 3, "bytes": 1 << 30, "seconds": 0.0123456789}},
                          "pairwise_timed_steps": {"score_gather": {"calls": 1, "bytes": 1 << 20, "seconds": 0.001}}},
            "cpu_baseline": {"value": 829.5113754188508, "unit": "pairs/s", "cores": 128, "kind": "port", "sample": "x" * prose,
                             "host_cpu_count": 256, "factor_fit_samples_per_sec": 15.39}}


def _worst_case(prose: int = 400) -> dict:
    full = _result("resnet9", prose)
    mnist = _result("mnist_mlp", prose)
    full["targets"] = {"mnist_mlp": {"gpu_pairs_per_sec": 4.2e6, "ms_per_step": 23.5, "cpu_pairs_per_sec": 9139.03, "cpu_cores": 128, "ratio": 464.8,
                                     "target_ratio": 10.0, "scores_rel_F_vs_cpu_oracle": 5.86e-5, "damping": 1e-8, "target_rel": 1e-4,
                                     "roofline": mnist["roofline"], "factor_fit": mnist["factor_fit"], "device_busy": mnist["device_busy"]}}
    full["other_configs"] = {name: _result(name, prose) for name in ("bert_base", "gpt2_small", "llama_block")}
    full["other_configs"]["broken"] = {"error": "RuntimeError: " + "e" * 300}
    return full


def _check(full: dict) -> dict:
    text = bench.render_line(full)
    assert "\n" not in text
    assert len(text) < bench.LINE_HARD_CAP_BYTES, len(text)
    line = json.loads(text)
    assert json.dumps(line, separators=(",", ":")) == text                       # round-trips
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    return line


def test_worst_case_line_is_compact_and_complete():
    full = _worst_case()
    assert len(json.dumps(full)) > 30000                                          # the shape that broke round 5
    line = _check(full)
    assert len(json.dumps(line)) <= bench.LINE_TARGET_BYTES
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-5
    assert all(not isinstance(v, str) or len(v) < 100 for v in roof.values())      # numbers, not prose
    cpu = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cpu) and len(cpu["sample"]) <= 160
    assert line["config"]["workload"] == "resnet9"
    assert line["targets"]["mnist_mlp"]["ratio"] == pytest.approx(464.8)
    assert set(line["other_configs"]) == {"bert_base", "gpt2_small", "llama_block", "broken"}
    gpt2 = line["other_configs"]["gpt2_small"]
    assert gpt2["config"] == {"workload": "gpt2_small", "n_train": 50000, "n_query": 1000, "parallelism": "train-shard-dp1"}
    assert gpt2["parity"]["ok"] is True and gpt2["roofline"]["frac"] == pytest.approx(0.324817)
    assert line["exchanges"]["query_exchange"] == "replicate"
    assert line["exchanges"]["factor_fit"]["factor_all_reduce"][:2] == [3, 1 << 30]
    assert line["extras_file"] == bench.EXTRAS_FILE


def test_absurd_prose_still_under_the_cap():
    """Sections are shed (to the extras file) before the cap is ever exceeded; the contract keys survive."""
    full = _worst_case(prose=200)
    full["other_configs"] = {f"cfg_{i}": _result(f"cfg_{i}") for i in range(60)}
    line = _check(full)
    assert line["other_configs"] == {"moved_to": bench.EXTRAS_FILE}
    assert line["roofline"]["frac"] == pytest.approx(0.324817) and line["cpu_baseline"]["cores"] == 128


def test_nan_and_missing_sections():
    full = _result("resnet9")
    full["roofline"]["traffic"] = None
    full["roofline_cov"] = None
    full["device_busy"] = {"error": "x" * 500}
    full["cpu_baseline"] = None
    full["ms_per_step"] = float("nan")
    line = _check(full)
    assert line["ms_per_step"] is None and line["roofline"]["traffic"] is None and line["roofline_cov"] is None


@pytest.mark.parametrize("name", ["r05_bench_default.json", "r05_bench_gpt2_full_100k_x_2000.json", "r05_bench_llama_4blocks.json"])
def test_kept_round5_objects(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip("kept profile not present")
    with open(path, encoding="utf-8") as handle:
        full = json.load(handle)
    line = _check(full)
    assert len(json.dumps(line)) <= bench.LINE_TARGET_BYTES
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)


def test_emit_prints_the_line_last(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        print("some earlier noise")
        bench.emit(_worst_case())
    last = out.getvalue().strip().splitlines()[-1]
    assert json.loads(last)["metric"] == "pairwise_influence_pairs_per_sec" and len(last) < bench.LINE_HARD_CAP_BYTES
    kept = json.loads((tmp_path / bench.EXTRAS_FILE).read_text())
    assert "largest_idle_after" in kept["device_busy"]                              # nothing is lost: the full object is beside the script
    assert err.getvalue().startswith("[bench extras] ")
