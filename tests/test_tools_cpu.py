"""Host-side measurement tooling (no GPU): the PMC summary's per-call byte accounting follows the kernel names the library
actually launches, and the committed summary is the one bench.py would quote (same source hash function on both sides)."""
import importlib.util
import json
import os

import pytest

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


def test_request_schedules_of_the_256_loop_are_ordered():
    """kf_pingpong.h, ISSUE 0 / 1 / 2: every fragment read sees its k-tile landed (RAW) and is over before the next occupant of
    its LDS rows is requested (WAR), for every loop length -- the header's ordering argument as an interval model -- and the
    model does reject a wait that is one piece short and a request issued one segment early."""
    model = _load("pp_schedule_check", os.path.join(ROOT, "tools", "pp_schedule_check.py"))

    for issue in (0, 1, 2):
        for nt in range(1, 10):
            errors, slack = model.check(issue, nt)
            assert not errors, (issue, nt, errors[:3])
            assert slack is None or slack >= 2   # never waited for in (or right after) the issuing segment
    real = model.program

    def short_wait(issue, nt):   # end of L(2t+1): vmcnt(2) -> vmcnt(4) leaves B1(t+1) in flight
        segs = real(issue, nt)
        return [(k, r, i, 4 if (k == "L" and w == 2) else w) for k, r, i, w in segs]

    def early_a1(issue, nt):     # A1(t+1) moved from M(2t) into M(2t-1): the other group is still reading A1(t-1) there
        segs = [list(s) for s in real(issue, nt)]
        for idx in range(len(segs) - 1, 1, -1):
            moved = [q for q in segs[idx][2] if q[0] == "A1"]
            if segs[idx][0] == "M" and moved and idx - 2 >= 1 and segs[idx - 2][0] == "M":
                segs[idx][2] = [q for q in segs[idx][2] if q[0] != "A1"]
                segs[idx - 2][2] = moved + segs[idx - 2][2]
        return [tuple(s) for s in segs]

    for mutant, kind in ((short_wait, "RAW"), (early_a1, "WAR")):
        model.program = mutant
        try:
            errors, _ = model.check(1, 6)
        finally:
            model.program = real
        assert any(e.startswith(kind) for e in errors), (kind, errors[:3])


def test_tn_operand_index_arithmetic_replays_on_the_host(tmp_path):
    """The K-major operand path (csrc/kf_pingpong_tn.h): the LDS images, LDS-DMA lane mappings and transposing-read addresses are
    plain functions (csrc/kf_tn_map.h); tools/tn_map_check.cpp replays them with g++ for all three images -- every chunk of a piece
    written once where the image says, every fragment the MFMA operand layout under the lane semantics of ds_read_b64_tr_b16, all
    64 banks per 32-lane half, the (k-slab, quad) address step a compile-time constant."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "kronfluence_amd", "csrc")
    exe = tmp_path / "tn_map_check"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", csrc, os.path.join(ROOT, "tools", "tn_map_check.cpp"), "-o", str(exe)], check=True)
    done = subprocess.run([str(exe)], capture_output=True, text=True)
    assert done.returncode == 0 and done.stdout.strip() == "ok", done.stdout[-500:]
    # ... and it is a check: an image without its swizzle, a DMA mapping with two k rows swapped, a wrong address step -> rejected
    header = open(os.path.join(csrc, "kf_tn_map.h")).read()
    mutants = [("return k * 256 + ((((fl >> 5) ^ k) & 3) << 6) + (fl & 31) * 2;", "return k * 256 + (((fl >> 5) & 3) << 6) + (fl & 31) * 2;"),
               ("static KF_TN_HD int dma_k(int q, int lane) { return 16 * (q & 3) + (lane >> 2); }",
                "static KF_TN_HD int dma_k(int q, int lane) { return 16 * (q & 3) + ((lane >> 2) ^ 1); }"),
               ("(16 * kk + 8 * quad) * 64", "(16 * kk + 4 * quad) * 64")]
    for index, (old, new) in enumerate(mutants):
        assert header.count(old) == 1, old
        work = tmp_path / f"mutant{index}"
        work.mkdir()
        (work / "kf_tn_map.h").write_text(header.replace(old, new))
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", str(work), os.path.join(ROOT, "tools", "tn_map_check.cpp"), "-o", str(work / "check")], check=True)
        assert subprocess.run([str(work / "check")], capture_output=True).returncode == 1, index
