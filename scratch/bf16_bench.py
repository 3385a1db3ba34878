import sys, time, torch
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
for (O,I,R) in [(256,2304,64),(128,1600,256),(128,1152,256),(64,27,1024)]:
    D=O*I
    P=torch.randn(Q,D,device=dev).to(torch.bfloat16); psg=torch.randn(b,D,device=dev).to(torch.bfloat16)
    C=torch.zeros(Q,b,device=dev)
    t=timeit(lambda: ops.gemm(C,b,0,ops.view(P,0,D,1,Q,D),ops.view(psg,0,D,1,b,D),beta=1.0))
    print(f'NT bf16 score gemm D={D}: {t:.3f} ms  {2*Q*b*D/t/1e9:.1f} TF')
    P32=P.float(); psg32=psg.float()
    t=timeit(lambda: ops.gemm(C,b,0,ops.view(P32,0,D,1,Q,D),ops.view(psg32,0,D,1,b,D),beta=1.0),1)
    print(f'   fp32 engine: {t:.3f} ms  {2*Q*b*D/t/1e9:.1f} TF')
    del P32, psg32
    G=torch.randn(b,R,O,device=dev).to(torch.bfloat16); A=torch.randn(b,R,I,device=dev).to(torch.bfloat16)
    out=torch.empty(b,O,I,device=dev,dtype=torch.bfloat16)
    import ctypes
    from kronfluence_amd import _native as nat
    def psg_call():
        ws=out
        # use the library path through pairwise_score's first half is not exposed; emulate with kf_gemm to fp32
        return ops.per_sample_gradient(G,A,False)
    t=timeit(psg_call)
    print(f'   TN psg (fp32 out) R={R}: {t:.3f} ms  {2*b*R*O*I/t/1e9:.1f} TF')
    S=torch.zeros(Q,b,device=dev)
    t=timeit(lambda: ops.pairwise_score(S,0,P.view(Q,O,I),G,A,False))
    print(f'   kf_pairwise_score total: {t:.3f} ms')
