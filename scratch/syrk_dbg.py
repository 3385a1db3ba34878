import sys, torch
sys.path.insert(0, '.')
from kronfluence_amd import ops
dev='cuda:0'
torch.manual_seed(0)
for (n,d) in [(64,16),(64,128),(128,128),(300,128),(300,256)]:
    x = torch.randn(n,d).to(torch.bfloat16)
    want = x.double().t() @ x.double()
    cov = torch.zeros(d,d,device=dev); cnt = torch.zeros(1,dtype=torch.int64,device=dev)
    ops.linear_activation_cov(cov, cnt, x.to(dev), None, False)
    err = (cov.double().cpu()-want)
    print((n,d), 'rel', float(err.norm()/want.norm()), 'max err at', divmod(int(err.abs().argmax()), d), 'diag rel', float(err.diag().norm()/want.diag().norm()))
    # generic TN gemm with same operands (non symmetric)
    xd = x.to(dev); C = torch.zeros(d,d,device=dev)
    ops.gemm(C, d, 0, ops.view(xd,0,1,d,d,n), ops.view(xd,0,1,d,d,n))
    print('   plain TN gemm rel', float((C.double().cpu()-want).norm()/want.norm()))
