"""Round-6 A/B of the blocked eigenbasis rotation (kf_rotate_rows_transposed_bf16 -> rotate_gemm_v3_kernel<0>) under the XCD grids of
``rotate_xcd_m`` (KF_ROT_XCD_M = 0 linear / 2 / 4 / 8) at the Lambda-stage shapes of the transformer configs: HIP-event time per call,
TFLOP/s on 2 n R d^2, results compared bit for bit with the linear order.

    gpurun -- 'python tools/r06_rot.py'            (under rocprofv3 --pmc FETCH_SIZE for the traffic)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kronfluence_amd import ops

DEV = "cuda:0"
CASES = [("gpt2 c_fc grad 3072, 128 x 512", 128, 512, 3072), ("gpt2 c_attn grad 2304, 128 x 512", 128, 512, 2304),
         ("gpt2 act 768, 128 x 512", 128, 512, 768), ("bert intermediate grad 3072, 512 x 128", 512, 128, 3072),
         ("llama 4096, 8 x 512", 8, 512, 4096), ("llama 14336, 8 x 512", 8, 512, 14336)]


def timed(fn, reps=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    only = sys.argv[1:]   # optional: grids to run (default all)
    grids = [int(x) for x in only] if only else [0, 2, 4, 8]
    torch.manual_seed(0)
    for name, n, r, d in CASES:
        x = torch.randn(n, r, d, device=DEV).bfloat16()
        q_t = (torch.randn(d, d, device=DEV) / d ** 0.5).bfloat16().contiguous()
        flops = 2.0 * n * r * d * d
        want = None
        line = []
        for g in grids:
            os.environ["KF_ROT_XCD_M"] = str(g)
            out = ops.rotate_rows_transposed(x, q_t)
            if want is None:
                want = out
            same = bool(torch.equal(out, want))
            ms = timed(lambda: ops.rotate_rows_transposed(x, q_t))
            line.append(f"gm={g}: {ms * 1e3:7.0f} us {flops / ms / 1e9:6.0f} TF/s{'' if same else ' MISMATCH'}")
        os.environ.pop("KF_ROT_XCD_M", None)
        ms = timed(lambda: ops.rotate_rows_transposed(x, q_t))
        print(f"{name:40s} " + " | ".join(line) + f" | default: {ms * 1e3:7.0f} us", flush=True)


if __name__ == "__main__":
    main()
