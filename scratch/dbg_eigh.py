import sys, torch, time, os
sys.path.insert(0, '.')
from kronfluence_amd import ops
from oracle import ekfac_ref as ref
torch.manual_seed(0)
n, d = 1000, 1025
x = torch.relu(torch.randn(n, d-1) @ torch.randn(d-1, d-1) * 0.05)
x[:, :20] = 0  # dead features
x = torch.cat([x, torch.ones(n,1)], 1)
cov = (x.t() @ x).float()
which = sys.argv[1]
c = cov.contiguous().cuda()
if which == 'shift':
    c = c.double(); c = c + torch.eye(d, device='cuda', dtype=torch.float64) * 1e-6 * float(c.norm())
t0=time.time()
try:
    ev, evec, sweeps = ops.eigh(c, float(n), max_sweeps=45)
    torch.cuda.synchronize()
    print(which, 'sweeps', sweeps, 'time', time.time()-t0)
    inv = ref.eigh_invariants(c.cpu(), torch.tensor([n]), ev.cpu(), evec.cpu())
    want, _ = ref.eigendecompose(c.double().cpu(), torch.tensor([n]))
    print(inv, 'eval err', float((ev.cpu()-want).abs().max()/want.abs().max()))
except Exception as e:
    print(which, 'ERR', e)
