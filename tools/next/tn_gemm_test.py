"""Driver of tools/next/tn_gemm.hip -- PREPARED FOR ROUND 5, never run on a GPU yet (tools/next/README.md).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/next/tn_gemm.hip -o tools/next/libtn_gemm.so
    gpurun --timeout 120 -- 'python tools/next/tr_probe.py && python tools/next/tn_gemm_test.py'

C = A^T B for K-major bf16 operands on the TN variant of the 256 x 256 main loop, against torch in fp32 on the same bf16 values:
1, 2, 3, 5 and many k-tiles, ragged tiles; then timing against the library's NT path on the same product (rotate-free comparison:
``ops.matmul``-style entry points do not exist for K-major operands, so the yardstick is the transposed copy + kf_gemm_out)."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


def main():
    lib = ctypes.CDLL(os.path.join(HERE, "libtn_gemm.so"))
    lib.tn_gemm.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                            ctypes.c_void_p]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    bad = 0
    for m, n, k in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (512, 256, 320), (768, 776, 512), (264, 1000, 1152), (3072, 776, 128)]:
        a = torch.randn(k, m, device=DEV).bfloat16()
        b = torch.randn(k, n, device=DEV).bfloat16()
        c = torch.zeros(m, n, device=DEV)
        rc = lib.tn_gemm(c.data_ptr(), n, a.data_ptr(), b.data_ptr(), m, n, k, stream)
        torch.cuda.synchronize()
        want = a.float().t() @ b.float()
        err = float((c - want).norm() / want.norm())
        flag = "" if (rc == 0 and err < 1e-5) else "   <-- MISMATCH"
        bad += bool(flag)
        print(f"  M={m:5d} N={n:5d} K={k:5d} (k-tiles {k // 64:3d}): rc {rc} rel_F {err:.1e}{flag}", flush=True)
    if bad:
        return 1
    for m, n, k in [(768, 776, 512), (3072, 776, 512), (4096, 4096, 4096)]:
        a = torch.randn(k, m, device=DEV).bfloat16()
        b = torch.randn(k, n, device=DEV).bfloat16()
        c = torch.zeros(m, n, device=DEV)
        for _ in range(2):
            lib.tn_gemm(c.data_ptr(), n, a.data_ptr(), b.data_ptr(), m, n, k, stream)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            lib.tn_gemm(c.data_ptr(), n, a.data_ptr(), b.data_ptr(), m, n, k, stream)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 10
        print(f"  M={m:5d} N={n:5d} K={k:5d}: {t:7.3f} ms {2.0 * m * n * k / t / 1e9:6.0f} TFLOP/s (one workgroup per tile, no split-K)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
