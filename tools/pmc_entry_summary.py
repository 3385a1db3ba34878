"""Summarises ``rocprofv3 --pmc`` passes of ``tools/r05_ab.py replay <workload> <entry>`` into ``profiles/pmc_<workload>.json``
(the same keys bench.py's ``_pmc_traffic`` reads for the ResNet-9 summary of tools/pmc_summary.py, plus one object per entry point).

    python tools/pmc_entry_summary.py <workload> <out.json> <base_dir>

``<base_dir>`` holds, per entry point (score | cov | lambda), ``<workload>_<entry>_meta.json`` (calls, algorithmic bytes / flops per
call: written by the replay) and the pass directories ``<workload>_<entry>_{fetch,write,mfma}``.  A replay process makes the calls of
ONE entry point only, so every dispatch in its passes belongs to that entry point: HBM bytes per call = sum over kernels of
launches x (2 x FETCH_SIZE + WRITE_SIZE) KiB / calls (FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md, HBM).
torch's own kernels in the process (randn / zeros of the set-up) are listed apart and not counted.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import kernel_source_hash, short  # noqa: E402

OURS = ("score_gemm", "psg_gemm", "psg_bias_cols", "transpose_rows", "cov_gemm", "cov_finalize", "colsum_accum", "rotate_gemm", "lambda_rows",
        "lambda_bf16", "syrk_kernel", "gemm_bf16", "__amd_rocclr_fillBuffer")
SCORE_GEMM = ("score_gemm_v2_kernel", "score_gemm_v3_kernel", "score_gemm_v4_kernel", "score_gemm_v5_kernel")


def entry_summary(base: str, workload: str, entry: str):
    meta_path = os.path.join(base, f"{workload}_{entry}_meta.json")
    if not os.path.exists(meta_path):
        return None
    with open(meta_path, encoding="utf-8") as handle:
        meta = json.load(handle)
    values = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(base, f"{workload}_{entry}_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                values[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels, total_read, total_write = {}, 0.0, 0.0
    for name, counters in sorted(values.items()):
        if not name.startswith(OURS) and "kf::" not in name:
            continue
        e = {"launches": max(len(v) for v in counters.values())}
        mean = {c: sum(v) / len(v) for c, v in counters.items()}
        if "FETCH_SIZE" in mean:
            e["hbm_read_bytes"] = 2.0 * mean["FETCH_SIZE"] * 1024.0
            total_read += e["launches"] * e["hbm_read_bytes"]
        if "WRITE_SIZE" in mean:
            e["hbm_write_bytes"] = mean["WRITE_SIZE"] * 1024.0
            total_write += e["launches"] * e["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean.get("GRBM_GUI_ACTIVE", 0) > 0:
            e["mfma_util"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        kernels[name] = e
    calls = meta["calls"]
    if not kernels or not calls:
        return None
    out = dict(meta)
    out.update({"hbm_read_bytes_per_call": total_read / calls, "hbm_write_bytes_per_call": total_write / calls,
                "hbm_bytes_per_call": (total_read + total_write) / calls,
                "traffic_over_algorithmic": (total_read + total_write) / calls / meta["algorithmic_bytes_per_call"], "kernels": kernels})
    return out


def main() -> None:
    workload, out_path, base = sys.argv[1], sys.argv[2], sys.argv[3]
    entries = {e: entry_summary(base, workload, e) for e in ("score", "cov", "lambda")}
    score, cov, lam = entries["score"], entries["cov"], entries["lambda"]
    dominant = None
    if score:
        gemms = [e for n, e in score["kernels"].items() if n.startswith(SCORE_GEMM)]
        dominant = max(gemms, key=lambda e: e["launches"], default=None)
    summary = {
        "workload": workload,
        "kernel_source_sha256": kernel_source_hash(),
        "source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes) on "
                  "`python tools/r05_ab.py replay <workload> <entry>`: the calls one train / factor batch of the workload makes to the "
                  "entry point, at bench.py's layer shapes, batch sizes and query count, one process per entry point (the counter "
                  "passes of the whole bench command segfault inside rocprofv3 on the transformer workloads); FETCH_SIZE doubled "
                  "(gfx950: MI355X_MICROARCH.md, HBM); summary by tools/pmc_entry_summary.py",
        "kf_pairwise_score_bytes_per_launch": score["hbm_bytes_per_call"] if score else None,
        "kf_pairwise_score_calls_profiled": score["calls"] if score else 0,
        "kf_pairwise_score_algorithmic_bytes_per_launch": score["algorithmic_bytes_per_call"] if score else None,
        "mfma_util": (dominant or {}).get("mfma_util"),
        "cov_call_bytes_per_launch": cov["hbm_bytes_per_call"] if cov else None,
        "cov_call_algorithmic_bytes_per_launch": cov["algorithmic_bytes_per_call"] if cov else None,
        "cov_gemm_mfma_util": max((e.get("mfma_util", 0.0) for n, e in (cov or {"kernels": {}})["kernels"].items() if n.startswith("cov_gemm")), default=None),
        "kf_lambda_bytes_per_launch": lam["hbm_bytes_per_call"] if lam else None,
        "kf_lambda_calls_profiled": lam["calls"] if lam else 0,
        "lambda_mfma_util": {n: e.get("mfma_util") for n, e in (lam or {"kernels": {}})["kernels"].items() if e.get("mfma_util") is not None} or None,
        "entries": entries,
    }
    with open(out_path, "w", encoding="utf-8") as handle:
        json.dump(summary, handle, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "entries"}, indent=1))
    for name, entry in entries.items():
        if entry:
            print(f"== {name}: {entry['calls']} calls, {entry['hbm_bytes_per_call'] / 1e6:.1f} MB per call "
                  f"({entry['traffic_over_algorithmic']:.2f}x the algorithmic {entry['algorithmic_bytes_per_call'] / 1e6:.1f} MB)")
            for n, e in entry["kernels"].items():
                print(f"  {n[:64]:64s} {e}")


if __name__ == "__main__":
    main()
