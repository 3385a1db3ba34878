import sys, time, torch
sys.path.insert(0, '.')
from kronfluence_amd import ops
dev='cuda:0'
torch.manual_seed(0)
n, d = 1000, 1025
x = torch.relu(torch.randn(n, d-1) @ torch.randn(d-1, d-1) * 0.05); x = torch.cat([x, torch.ones(n,1)], 1)
cov = (x.t() @ x).float().to(dev)
ev, evec, sw = ops.eigh(cov, float(n))
w, v = torch.linalg.eigh((cov.double()/n + (cov.double()/n).t())*0.5)
tiny = 1.1754944e-38
for name, q in (('kf', evec.float()), ('torch', v.float())):
    a = q.abs(); print(name, 'denormal frac', float(((a>0)&(a<tiny)).float().mean()), 'zero frac', float((a==0).float().mean()))
M, K = 102400, 1025
A = torch.randn(M, K, device=dev)
def bench_gemm(Q, label):
    Q = Q.contiguous()
    T = torch.empty(M, K, device=dev)
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.gemm(T, K, 0, ops.view(A, 0, K, 1, M, K), ops.view(Q, 0, K, 1, K, K)); e.record(); torch.cuda.synchronize()
    print(label, 'gemm ms', s.elapsed_time(e))
bench_gemm(evec.float(), 'kf eigvecs')
bench_gemm(v.float(), 'torch eigvecs')
q2 = evec.float().clone(); q2[q2.abs() < tiny] = 0
bench_gemm(q2, 'kf flushed')
