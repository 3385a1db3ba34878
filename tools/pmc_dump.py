"""Per-kernel means of every counter found in ``rocprofv3 --pmc`` output directories (one line per kernel of this library).

    python tools/pmc_dump.py <pass_dir> [<pass_dir> ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

OURS = ("score_gemm_v", "psg_gemm_v", "cov_gemm_v", "rotate_gemm_v", "cov_finalize", "conv_pad_phases", "transpose_rows", "gemm_bf16_kernel",
        "lambda_bf16", "score_r1", "syrk_kernel", "eigh_")


def main() -> None:
    values = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as handle:
                for row in csv.DictReader(handle):
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
                    values[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name in sorted(values):
        if not any(k in name for k in OURS):
            continue
        counters = values[name]
        n = max(len(v) for v in counters.values())
        print(f"{name[:60]:60s} launches {n:5d}  " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(counters.items())))


if __name__ == "__main__":
    main()
