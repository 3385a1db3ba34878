"""Summarises ``rocprofv3 --pmc`` passes of a bench command into ``profiles/pmc_<workload>.json`` (read back by bench.py for
``roofline.traffic`` / ``roofline.mfma_util``; the summary records a hash of the kernel sources it was taken on and bench.py
refuses it when the sources have changed since).

    python tools/pmc_summary.py <workload> <out.json> <pass_dir> [<pass_dir> ...]

Every pass directory holds one ``*_counter_collection.csv`` (one row per dispatch and counter).  Counters used:

  FETCH_SIZE / WRITE_SIZE   KiB moved over the L2's memory-side interface per dispatch (separate passes: FETCH_SIZE takes 3 of
                            the 4 TCC slots).  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports HALF the bytes of a wide
                            coalesced read stream -> the read side is DOUBLED here; WRITE_SIZE is taken as reported.
  SQ_VALU_MFMA_BUSY_CYCLES  cycles the matrix pipes were busy, summed over the 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16:
                            the count equals 32 x the kernel's MFMA instructions, checked against the algorithmic flops)
  GRBM_GUI_ACTIVE           active cycles of the dispatch, SUMMED OVER THE 8 XCDs (each XCD has its own GRBM; the sum is 8x the
                            kernel-trace duration times the clock)  ->  mfma_util = busy / (active / 8 * 1024 SIMDs)
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORE_GEMM = ("score_gemm_v2_kernel", "score_gemm_v3_kernel", "score_gemm_v4_kernel", "score_gemm_v5_kernel")
PSG = ("psg_gemm_v2_kernel", "psg_gemm_v3_kernel", "psg_gemm_pp_kernel", "psg_gemm_tn_kernel")
# instantiations of the persistent gradient kernel that serve OTHER entry points: <1, .> rows for the dense-form Lambda, <2, .> the
# query-side preconditioner
PSG_NOT_SCORE = ("psg_gemm_v3_kernel<1", "psg_gemm_v3_kernel<2")
COV_GEMM = ("cov_gemm_v2_kernel", "cov_gemm_v3_kernel", "cov_gemm_tn_kernel")
SCORE_KERNELS = SCORE_GEMM + PSG + ("conv_pad_phases_kernel", "pad_grid_kernel", "transpose_rows_kernel", "score_r1_kernel")


def kernel_source_hash() -> str:
    """sha256 over the sources of the kernels the summary reports -- every HIP source and header but the eigensolver's (same
    function as bench.py's): identifies the code a profile was taken on."""
    digest = hashlib.sha256()
    paths = sorted(glob.glob(os.path.join(ROOT, "kronfluence_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "kronfluence_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for path in paths:
        if os.path.basename(path) == "kf_eigh.hip":
            continue   # the eigensolver: none of its kernels is in the PMC summary (its evidence is the eigh logs under profiles/)
        with open(path, "rb") as handle:
            digest.update(os.path.basename(path).encode() + b"\0" + handle.read())
    return digest.hexdigest()


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].strip()


def score_call_bytes(kernels: dict, calls: int) -> float:
    """HBM bytes of all kf_pairwise_score* calls.  A score call launches exactly one per-sample-gradient kernel
    (``psg_gemm_v3_kernel<0, .>`` / ``psg_gemm_pp_kernel`` / ``psg_gemm_v2_kernel``; see PSG_NOT_SCORE for the others) with
    one conv_pad_phases_kernel (convolution) or two transpose_rows_kernel (sequence rows) -- kernels that also serve the
    covariance / Lambda entry points when the profiled command ran the factor fit, so only that share of their launches counts."""
    psg = sum(e["launches"] for n, e in kernels.items() if n.startswith(PSG) and not n.startswith(PSG_NOT_SCORE) and not n.startswith(("psg_gemm_pp", "psg_gemm_tn")))
    total = 0.0
    for n, e in kernels.items():
        if not n.startswith(SCORE_KERNELS) or n.startswith(PSG_NOT_SCORE):
            continue
        share = 1.0
        if n.startswith("conv_pad_phases_kernel"):
            share = min(1.0, psg / e["launches"])
        elif n.startswith("transpose_rows_kernel"):
            share = min(1.0, 2.0 * psg / e["launches"])
        total += share * e["launches"] * (e.get("hbm_read_bytes", 0.0) + e.get("hbm_write_bytes", 0.0))
    return total


def lambda_call_bytes(kernels: dict):
    """-> (HBM bytes per kf_lambda_accum / kf_lambda_rows_accum / kf_lambda_conv2d_accum call, calls, MFMA utilisation per Lambda kernel).  A factored call is one lambda_rows_kernel / lambda_bf16_kernel / lambda_kernel launch; a dense call (Conv2d, R > O) is one
    conv_pad_phases_kernel + psg_gemm_v3_kernel<1> (rows ordered (o', n)) + rotate_gemm_v3_kernel<1> (sum-of-squares GEMM)."""
    dense = [e for n, e in kernels.items() if n.startswith("rotate_gemm_v3_kernel<1")]
    parts = [e for n, e in kernels.items() if n.startswith(("lambda_bf16_kernel", "kf::lambda_bf16_kernel", "lambda_kernel", "lambda_rows_kernel",
                                                             "rotate_gemm_v3_kernel<1", "psg_gemm_v3_kernel<1"))]
    calls = sum(e["launches"] for n, e in kernels.items()
                if n.startswith(("lambda_bf16_kernel", "kf::lambda_bf16_kernel", "lambda_kernel", "lambda_rows_kernel", "rotate_gemm_v3_kernel<1")))
    if not calls:
        return None, 0, None
    total = sum(e["launches"] * (e.get("hbm_read_bytes", 0.0) + e.get("hbm_write_bytes", 0.0)) for e in parts)
    pad = next((e for n, e in kernels.items() if n.startswith("conv_pad_phases_kernel")), None)
    if pad is not None and dense:
        total += sum(e["launches"] for e in dense) * (pad.get("hbm_read_bytes", 0.0) + pad.get("hbm_write_bytes", 0.0))
    names = ("lambda_bf16_kernel", "kf::lambda_bf16_kernel", "lambda_kernel", "lambda_rows_kernel", "rotate_gemm_v3_kernel<1", "psg_gemm_v3_kernel<1")
    util = {n: e.get("mfma_util") for n, e in kernels.items() if n.startswith(names) and e.get("mfma_util") is not None}
    return total / calls, calls, util


def main() -> None:
    if sys.argv[1] == "--rehash":  # python tools/pmc_summary.py --rehash <summary.json> <old tree>: the hash FUNCTION changed (a file
        # was excluded); re-record the hash only if the files it still covers are byte-identical in <old tree> (the tree the
        # profile was taken on, e.g. a `git worktree` of that commit) and in this one, and drop the rows of the excluded kernels
        global ROOT
        with open(sys.argv[2], encoding="utf-8") as handle:
            summary = json.load(handle)
        here = kernel_source_hash()
        keep, ROOT = ROOT, os.path.abspath(sys.argv[3])
        there = kernel_source_hash()
        ROOT = keep
        if here != there:
            raise SystemExit("the covered sources differ between the two trees: take a new profile instead")
        summary["kernel_source_sha256"] = here
        summary["kernels"] = {n: e for n, e in summary["kernels"].items() if "eigh" not in n and "jacobi" not in n}
        with open(sys.argv[2], "w", encoding="utf-8") as handle:
            json.dump(summary, handle, indent=1)
        print(here)
        return
    if sys.argv[1] == "--recompute":  # python tools/pmc_summary.py --recompute <summary.json>: totals from the kept per-kernel means
        with open(sys.argv[2], encoding="utf-8") as handle:
            summary = json.load(handle)
        calls = summary["kf_pairwise_score_calls_profiled"]
        summary["kf_pairwise_score_bytes_per_launch"] = score_call_bytes(summary["kernels"], calls) / calls if calls else None
        summary["kf_lambda_bytes_per_launch"], summary["kf_lambda_calls_profiled"], summary["lambda_mfma_util"] = lambda_call_bytes(summary["kernels"])
        with open(sys.argv[2], "w", encoding="utf-8") as handle:
            json.dump(summary, handle, indent=1)
        print(summary["kf_pairwise_score_bytes_per_launch"])
        return
    workload, out_path, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    values = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> per-dispatch values
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as handle:
                for row in csv.DictReader(handle):
                    values[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {}
    for name, counters in sorted(values.items()):
        entry = {"launches": max(len(v) for v in counters.values())}
        mean = {c: sum(v) / len(v) for c, v in counters.items()}
        if "FETCH_SIZE" in mean:
            entry["FETCH_SIZE_KiB"] = mean["FETCH_SIZE"]
            entry["hbm_read_bytes"] = 2.0 * mean["FETCH_SIZE"] * 1024.0  # gfx950 correction
        if "WRITE_SIZE" in mean:
            entry["WRITE_SIZE_KiB"] = mean["WRITE_SIZE"]
            entry["hbm_write_bytes"] = mean["WRITE_SIZE"] * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
            entry["mfma_busy_cycles"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"]
            entry["gui_active_cycles"] = mean["GRBM_GUI_ACTIVE"]
            entry["mfma_util"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        kernels[name] = entry
    # one kf_pairwise_score* call = its pad / transpose / gradient kernels + one score GEMM (or one score_r1 launch)
    calls = sum(e["launches"] for n, e in kernels.items() if n.startswith(SCORE_GEMM) or n.startswith("score_r1_kernel"))
    total = score_call_bytes(kernels, calls)
    dominant = max((e for n, e in kernels.items() if n.startswith(SCORE_GEMM)), key=lambda e: e["launches"], default=None)
    cov = max((e for n, e in kernels.items() if n.startswith(COV_GEMM)), key=lambda e: e["launches"], default=None)
    summary = {
        "workload": workload,
        "kernel_source_sha256": kernel_source_hash(),
        "source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes) on the bench "
                  "command; FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md, HBM); summary by tools/pmc_summary.py",
        "kf_pairwise_score_bytes_per_launch": total / calls if calls else None,
        "kf_pairwise_score_calls_profiled": calls,
        "mfma_util": dominant.get("mfma_util") if dominant else None,
        # covariance stage (present when the profiled command ran the factor fit): the GEMM kernel of the LDS-DMA covariance path
        "cov_gemm_bytes_per_launch": (cov.get("hbm_read_bytes", 0.0) + cov.get("hbm_write_bytes", 0.0)) if cov else None,
        "cov_gemm_mfma_util": (cov or {}).get("mfma_util"),
        "kernels": {n: e for n, e in kernels.items() if n.startswith(SCORE_KERNELS) or "gemm_bf16" in n or "syrk" in n or "lambda" in n
                    or "im2col" in n or n.startswith("cov_") or n.startswith("rotate_gemm")},
    }
    summary["kf_lambda_bytes_per_launch"], summary["kf_lambda_calls_profiled"], summary["lambda_mfma_util"] = lambda_call_bytes(summary["kernels"])
    with open(out_path, "w", encoding="utf-8") as handle:
        json.dump(summary, handle, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
    for n, e in summary["kernels"].items():
        print(f"  {n[:70]:70s} {e}")


if __name__ == "__main__":
    main()
