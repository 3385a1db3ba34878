import sys, time, torch
sys.path.insert(0, '.')
from kronfluence_amd import ops
dev='cuda:0'
q,r,o,i=100,1,1024,1024; ip=i+1
g=torch.randn(q,r,o,device=dev); a=torch.randn(q,r,i,device=dev)
qg=torch.linalg.qr(torch.randn(o,o,device=dev))[0].contiguous(); qa=torch.linalg.qr(torch.randn(ip,ip,device=dev))[0].contiguous()
li=torch.rand(o,ip,device=dev)+0.1
def ev(): return torch.cuda.Event(enable_timing=True)
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    out=ops.precondition(g,a,True,qg,qa,li)
    torch.cuda.synchronize(); t1=time.perf_counter()
    print('ops.precondition wall ms', (t1-t0)*1e3)
    del out
# manual steps with events
for rep in range(2):
    es=[ev() for _ in range(8)]
    torch.cuda.synchronize(); t0=time.perf_counter()
    P=torch.empty(q,o,ip,device=dev); T=torch.empty(q*o,ip,device=dev)
    torch.cuda.synchronize(); t1=time.perf_counter()
    es[0].record()
    gt=ops.matmul_nn(g.reshape(q*r,o),qg); es[1].record()
    at=ops.matmul_nn(a.reshape(q*r,i),qa,append_ones=True); es[2].record()
    ops.gemm(P,ip,o*ip,ops.view(gt,r*o,1,o,o,r),ops.view(at,r*ip,1,ip,ip,r),batch=q,mul=li); es[3].record()
    ops.gemm(T,ip,0,ops.view(P,0,ip,1,q*o,ip),ops.view(qa,0,ip,1,ip,ip)); es[4].record()
    ops.gemm(P,ip,o*ip,ops.view(qg,0,o,1,o,o),ops.view(T,o*ip,1,ip,ip,o),batch=q); es[5].record()
    torch.cuda.synchronize(); t2=time.perf_counter()
    print('alloc ms',(t1-t0)*1e3,'total wall',(t2-t1)*1e3,'steps ms',[round(es[k].elapsed_time(es[k+1]),3) for k in range(5)])
