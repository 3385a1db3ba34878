"""``kf_eigh_f64`` on rank-graded covariance matrices (d = 300 ... 4096): time, sweeps, orthogonality, reconstruction,
eigenvalues against LAPACK.  ``KF_EIGH_BLOCK8=1`` forces the round-1 8-column VALU kernel, ``KF_EIGH_SCALAR=1`` the scalar one, for comparison."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kronfluence_amd import ops
dev = "cuda:0"
sizes = [(300, 150), (769, 4000), (1025, 4000), (1152, 800), (2304, 5000), (3073, 6000), (4096, 8000)]
if len(sys.argv) > 1:
    sizes = [(d, n) for d, n in sizes if str(d) in sys.argv[1:]]
for d, n in sizes:
    g = torch.Generator().manual_seed(d)
    x = torch.randn(n, d, generator=g) * torch.logspace(0, -3, d)
    cov = (x.t() @ x).to(dev)
    ops.eigh(cov, float(n)); torch.cuda.synchronize()
    t0 = time.perf_counter(); evals, evecs, sweeps = ops.eigh(cov, float(n)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = 0.5 * (cov.double() + cov.double().t()) / n
    ortho = float((evecs.t() @ evecs - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
    recon = float(((evecs * evals) @ evecs.t() - s).norm() / s.norm())
    want = torch.linalg.eigvalsh(s.cpu())
    verr = float((evals.cpu() - want).abs().max() / want.abs().max())
    print(f"d={d}: {dt*1e3:.0f} ms, {sweeps} sweeps, ortho {ortho:.2e}, recon {recon:.2e}, evals {verr:.2e}, ascending {bool((evals[1:] >= evals[:-1]).all())}")
