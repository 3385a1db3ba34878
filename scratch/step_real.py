import sys, time, torch
sys.path.insert(0, '.')
import bench
from kronfluence_amd import FactorArguments, ScoreArguments, ops, prepare_model
from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader
from kronfluence_amd.score import pairwise as pw
from kronfluence_amd.utils.dataset import ResidentLoader
from kronfluence_amd.utils.state import State
state = State(); dev = state.device
spec = bench.WORKLOADS['mnist_mlp']
task = bench.make_task()
model = prepare_model(spec['model'](), task).to(dev)
train = bench.synth(spec, 1000, 1, dev); query = bench.synth(spec, 100, 2, dev)
fargs, sargs = FactorArguments(use_empirical_fisher=True), ScoreArguments()
_, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs)
eig = {k: {} for k in ('activation_eigenvectors','gradient_eigenvectors','activation_eigenvalues','gradient_eigenvalues')}
for side in ('activation','gradient'):
    for name, c in cov[f'{side}_covariance'].items():
        cc = c.double().to(dev)/1000; cc = 0.5*(cc+cc.t())
        if 'kf' in sys.argv:
            w, v, _ = ops.eigh(c.to(dev), 1000.0)
        else:
            w, v = torch.linalg.eigh(cc)
        eig[f'{side}_eigenvalues'][name] = w.float().cpu(); eig[f'{side}_eigenvectors'][name] = v.float().contiguous().cpu()
_, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs, eig)
factors = {k: {n: v.to(dev) for n, v in d.items()} for k, d in {**eig, **lam}.items()}
if 'side' in sys.argv:
    ops.eigh(torch.randn(1025,1025,device=dev).double() @ torch.randn(1025,1025,device=dev).double().t(), 1.0)
def step(f, sa):
    return pw.compute_pairwise_scores_with_loaders(f, model, state, task, ResidentLoader(query, 100), 100, ResidentLoader(train, 1000), sa, fargs, None)
for label, sa in (('damp1e-8', ScoreArguments()),):
    for it in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter(); s = step(factors, sa); torch.cuda.synchronize(); dt=(time.perf_counter()-t0)*1e3
    print(label, 'step ms', dt, 'scores absmax', float(s['all_modules'].abs().max()))
# how many denormals in lambda / eigenvectors
for name, l in lam['lambda_matrix'].items():
    x = l.float(); print(name, 'lambda min', float(x.min()), 'frac<1e-30', float((x.abs()<1e-30).float().mean()))

# event timing of precondition and score calls (no syncs inside)
import kronfluence_amd.module.tracker.precondition as pt
log = []
orig = ops.precondition
def timed(*a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); out = orig(*a, **k); e.record(); log.append((s, e, tuple(out.shape))); return out
pt.ops.precondition = timed
ops.SCORE_EVENT_LOG = []
step(factors, ScoreArguments()); torch.cuda.synchronize()
print('precondition ms', [(round(s.elapsed_time(e), 2), shp) for s, e, shp in log])
print('score ms', [round(s.elapsed_time(e), 2) for s, e, _ in ops.SCORE_EVENT_LOG])
P = None
