"""Host-side measurement tooling (no GPU): the PMC summary's per-call byte accounting follows the kernel names the library
actually launches, and the committed summary is the one bench.py would quote (same source hash function on both sides)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def _kernels():
    mb = 1e6
    entry = lambda launches, read, write: {"launches": launches, "hbm_read_bytes": read * mb, "hbm_write_bytes": write * mb}
    return {
        "score_gemm_v3_kernel": entry(8, 1000, 50),
        "psg_gemm_v3_kernel<0>": entry(8, 300, 500),        # score path
        "psg_gemm_v3_kernel<1>": entry(4, 300, 300),        # dense-form Lambda
        "psg_gemm_v3_kernel<2>": entry(6, 100, 100),        # query-side preconditioner
        "conv_pad_phases_kernel": entry(16, 90, 110),       # 8 score calls + 4 dense Lambda calls + 4 covariance calls
        "rotate_gemm_v3_kernel<1>": entry(4, 700, 10),
        "kf::lambda_bf16_kernel": entry(4, 400, 1),
        "rotate_gemm_v3_kernel<0>": entry(30, 600, 200),    # rotations: neither a score nor a Lambda-product kernel
    }


def test_score_and_lambda_call_bytes_follow_the_kernel_names():
    pmc = _load("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    kernels = _kernels()
    # 8 score calls: score GEMM + its gradient kernel + one pad each (8 of the 16 pad launches)
    assert abs(pmc.score_call_bytes(kernels, 8) / 8 - (1050 + 800 + 200) * 1e6) < 1.0
    per_call, calls, util = pmc.lambda_call_bytes(kernels)
    # 4 dense calls (pad + rows gradient + sum-of-squares GEMM) and 4 factored calls
    assert calls == 8 and util == {}
    assert abs(per_call - (4 * (200 + 600 + 710) + 4 * 401) * 1e6 / 8) < 1.0


def test_committed_pmc_summary_matches_the_hash_function_of_bench():
    pmc = _load("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    bench = _load("bench_module", os.path.join(ROOT, "bench.py"))
    assert pmc.kernel_source_hash() == bench.kernel_source_hash()
    with open(os.path.join(ROOT, "profiles", "pmc_resnet9.json"), encoding="utf-8") as handle:
        summary = json.load(handle)
    for key in ("kernel_source_sha256", "kf_pairwise_score_bytes_per_launch", "kf_lambda_bytes_per_launch", "mfma_util", "kernels"):
        assert key in summary, key
    # names the summary is built from must exist in the profile it was built from
    assert any(name.startswith("score_gemm_v3_kernel") for name in summary["kernels"])
    assert any(name.startswith("psg_gemm_v3_kernel<0") for name in summary["kernels"])
