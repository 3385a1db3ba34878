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
for D in (589824, 204800, 147456):
    for pad in (0, 64, 192, 1088):
        ld=D+pad
        P=torch.randn(Q,ld,device=dev).to(torch.bfloat16); psg=torch.randn(b,ld,device=dev).to(torch.bfloat16)
        C=torch.zeros(Q,b,device=dev)
        t=timeit(lambda: ops.gemm(C,b,0,ops.view(P,0,ld,1,Q,D),ops.view(psg,0,ld,1,b,D),beta=1.0))
        print(f'D={D} pad={pad}: {t:.3f} ms  {2*Q*b*D/t/1e9:.1f} TF', flush=True)
