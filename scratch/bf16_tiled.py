import sys, torch
sys.path.insert(0, '.')
from kronfluence_amd import ops
dev='cuda:0'
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
Q,b=1000,1000
for D in (589824, 147456):
    P=torch.randn(Q,D,device=dev).to(torch.bfloat16); G=torch.randn(b,D,device=dev).to(torch.bfloat16)
    want=torch.zeros(Q,b,device=dev)
    ops.gemm(want,b,0,ops.view(P,0,D,1,Q,D),ops.view(G,0,D,1,b,D))
    Pt=P.view(Q,D//64,64).transpose(0,1).contiguous(); Gt=G.view(b,D//64,64).transpose(0,1).contiguous()
    C=torch.zeros(Q,b,device=dev)
    f=lambda: ops.gemm(C,b,0,ops.view(Pt,Q*64,64,1,Q,D),ops.view(Gt,b*64,64,1,b,D),beta=0.0)
    t=timeit(f)
    print(f'D={D} tiled: {t:.3f} ms  {2*Q*b*D/t/1e9:.1f} TF  err {float((C-want).norm()/want.norm()):.2e}', flush=True)
