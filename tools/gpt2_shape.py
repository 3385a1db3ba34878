"""GPT-2-small layer shapes (Linear with bias on 512-token sequences): ``kf_pairwise_score`` on the fp32-engine
fallback (I' = I + 1 odd) against the zero-padded bf16 path the tracker uses (PairwiseScoreTracker.PAD_PATCH_AXIS)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from kronfluence_amd import ops
dev = "cuda:0"
Q, b, r = 256, 16, 512
for name, (o, i) in {"gpt2 c_fc (3072,768+1)": (3072, 768), "gpt2 c_attn (2304,768+1)": (2304, 768), "gpt2 c_proj (768,3072+1)": (768, 3072)}.items():
    p = torch.randn(Q, o, i + 1, device=dev).bfloat16()
    g = torch.randn(b, r, o, device=dev).bfloat16(); a = torch.randn(b, r, i, device=dev).bfloat16()
    fl = 2.0 * Q * b * o * (i + 1) + 2.0 * b * r * o * (i + 1)
    def run(fn):
        scores = torch.zeros(Q, b, device=dev)
        for _ in range(2): fn(scores)
        torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): fn(scores)
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / 3, scores
    t_plain, s_plain = run(lambda sc: ops.pairwise_score(sc, 0, p, g, a, True))
    pad = (-(i + 1)) % 8
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries
    tiled = TiledQueries(p, pad)
    ap = torch.cat([a, a.new_ones(b, r, 1), a.new_zeros(b, r, pad)], dim=-1)
    t_pad, s_pad = run(lambda sc: ops.pairwise_score(sc, 0, tiled, g, ap, False))
    err = float((s_pad - s_plain).norm() / s_plain.norm())
    print(f"{name}: fp32-engine fallback {t_plain:.2f} ms ({fl/t_plain/1e9:.0f} TF/s) -> padded bf16 {t_pad:.2f} ms ({fl/t_pad/1e9:.0f} TF/s), rel diff {err:.1e}")
