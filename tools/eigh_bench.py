"""``kf_eigh_f64`` on rank-graded covariance matrices (d = 300 ... 4096): time, sweeps, orthogonality, reconstruction,
eigenvalues against LAPACK.  ``KF_EIGH_BLOCK8=1`` forces the round-1 8-column VALU kernel, ``KF_EIGH_SCALAR=1`` the scalar one, for comparison."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kronfluence_amd import ops
dev = "cuda:0"
sizes = [(300, 150), (769, 4000), (1025, 4000), (1152, 800), (2304, 5000), (3073, 6000), (4096, 8000)]
if len(sys.argv) > 1:
    sizes = [(d, n) for d, n in sizes if str(d) in sys.argv[1:]] if sys.argv[1] != "multi" else []
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

# ---- the eigen stage's own shape: MULTI matrices of one size spread over LANES host threads / HIP streams -------------
# python tools/eigh_bench.py multi <d> <count> [lanes]   e.g. multi 3073 16 8  (a BERT / GPT-2 eigen stage has 24 of 3072-3073)
if len(sys.argv) > 3 and sys.argv[1] == "multi":
    import threading
    from concurrent.futures import ThreadPoolExecutor
    d, count = int(sys.argv[2]), int(sys.argv[3])
    lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    covs = []
    for k in range(count):
        g = torch.Generator().manual_seed(1000 + k)
        x = (torch.randn(2 * d, d, generator=g) * torch.logspace(0, -3, d)).to(dev)
        covs.append(x.t() @ x)
    ops.eigh(covs[0], 2.0 * d)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    queue, lock, sweeps = list(range(count)), threading.Lock(), []

    def worker(stream):
        torch.cuda.set_device(0)
        while True:
            with lock:
                if not queue:
                    return
                k = queue.pop(0)
            with torch.cuda.stream(stream):
                sweeps.append(ops.eigh(covs[k], 2.0 * d)[2])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=lanes) as pool:
        for f in [pool.submit(worker, s) for s in streams]:
            f.result()
    torch.cuda.synchronize()
    print(f"multi d={d} x {count} on {lanes} lanes: {time.perf_counter() - t0:.2f} s, sweeps {sorted(sweeps)}")
