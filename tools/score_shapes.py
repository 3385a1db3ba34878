"""``kf_pairwise_score`` on the ResNet-9 layer shapes with random data: total time per call and, for the bf16
k-tile-major path, the NT score GEMM alone (-> the TN per-sample-gradient GEMM by difference).

    gpurun -- 'python tools/score_shapes.py [l2 l3 l6 lin l1]'"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kronfluence_amd import ops
from kronfluence_amd.module.tracker.pairwise_score import TiledQueries
dev = "cuda:0"
shapes = {"l2": (128, 1600, 256), "l3": (128, 1152, 256), "l6": (256, 2304, 64), "lin": (10, 128, 1), "l1": (64, 27, 1024)}
which = sys.argv[1:] or list(shapes)
Q = b = 1000
for name in which:
    o, i, r = shapes[name]
    p = torch.randn(Q, o, i, device=dev).bfloat16()
    g = torch.randn(b, r, o, device=dev).bfloat16(); a = torch.randn(b, r, i, device=dev).bfloat16()
    scores = torch.zeros(Q, b, device=dev)
    tq = TiledQueries(p, 0) if (o * i) % 64 == 0 and r > 1 else None
    tiled = tq.tiled if tq is not None else None
    for _ in range(2): ops.pairwise_score(scores, 0, tq if tq is not None else p, g, a, False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): ops.pairwise_score(scores, 0, tq if tq is not None else p, g, a, False)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    fl = 2.0 * Q * b * o * i + (2.0 * b * r * o * i if r > 1 else 0)
    print(f"{name}: {ms:.3f} ms  {fl/ms/1e9:.0f} TF/s")
    if tiled is not None:
        d = o * i
        fake = torch.randn(d // 64, b, 64, device=dev).bfloat16()
        vp = ops.view(tiled, 0, 64, 1, Q, d, k_tile_stride=Q * 64)
        vg = ops.view(fake, 0, 64, 1, b, d, k_tile_stride=b * 64)
        for _ in range(2): ops.gemm(scores, b, 0, vp, vg, 1, 1.0, beta=1.0)
        torch.cuda.synchronize(); s.record()
        for _ in range(5): ops.gemm(scores, b, 0, vp, vg, 1, 1.0, beta=1.0)
        e.record(); torch.cuda.synchronize()
        nt = s.elapsed_time(e) / 5
        print(f"   NT alone: {nt:.3f} ms ({2.0*Q*b*d/nt/1e9:.0f} TF/s)  -> TN psg ~ {ms-nt:.3f} ms ({2.0*b*r*d/(ms-nt)/1e9:.0f} TF/s)")
