import sys, time, torch
sys.path.insert(0, '.')
import bench
from torch.profiler import profile, ProfilerActivity
from kronfluence_amd import FactorArguments, ScoreArguments, prepare_model
from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader
from kronfluence_amd.score.pairwise import compute_pairwise_scores_with_loaders
from kronfluence_amd.utils.dataset import ResidentLoader
from kronfluence_amd.utils.state import State
state = State(); dev = state.device
spec = bench.WORKLOADS['resnet9']; task = bench.make_task()
model = prepare_model(spec['model'](), task).to(dev)
train = bench.synth(spec, 2000, 1, dev); query = bench.synth(spec, 250, 2, dev)
fargs = FactorArguments(use_empirical_fisher=True, amp_dtype=torch.bfloat16); sargs = ScoreArguments(amp_dtype=torch.bfloat16)
def top(prof, label, n=8):
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:n]
    print('==', label)
    for e in rows: print(f'   {e.device_time_total/1e3:9.2f} ms  x{e.count:5d}  {e.key[:90]}')
_, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs)
with profile(activities=[ProfilerActivity.CUDA]) as p:
    _, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs)
    torch.cuda.synchronize()
top(p, 'covariance 2x1000')
eig = {k: {} for k in ('activation_eigenvectors','gradient_eigenvectors','activation_eigenvalues','gradient_eigenvalues')}
for side in ('activation','gradient'):
    for name, c in cov[f'{side}_covariance'].items():
        d = c.shape[0]
        eig[f'{side}_eigenvectors'][name] = torch.linalg.qr(torch.randn(d,d))[0].contiguous(); eig[f'{side}_eigenvalues'][name] = torch.rand(d)
_, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs, eig)
with profile(activities=[ProfilerActivity.CUDA]) as p:
    _, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs, eig)
    torch.cuda.synchronize()
top(p, 'lambda 2x1000')
factors = {k: {n: v.to(dev) for n, v in d.items()} for k, d in {**eig, **lam}.items()}
compute_pairwise_scores_with_loaders(factors, model, state, task, ResidentLoader(query, 250), 250, ResidentLoader(train, 1000), sargs, fargs, None)
with profile(activities=[ProfilerActivity.CUDA]) as p:
    compute_pairwise_scores_with_loaders(factors, model, state, task, ResidentLoader(query, 250), 250, ResidentLoader(train, 1000), sargs, fargs, None)
    torch.cuda.synchronize()
top(p, 'pairwise 250 x 2000', 12)
