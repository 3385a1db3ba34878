import sys, time, torch
sys.path.insert(0, '.')
import bench
from kronfluence_amd import FactorArguments, ScoreArguments, ops, prepare_model
from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader, perform_eigendecomposition
from kronfluence_amd.score import pairwise as pw
from kronfluence_amd.score import dot_product as dp
from kronfluence_amd.utils.dataset import ResidentLoader
from kronfluence_amd.utils.state import State
import kronfluence_amd.module.utils as mu

state = State(); dev = state.device
spec = bench.WORKLOADS['mnist_mlp']
task = bench.make_task()
model = prepare_model(spec['model'](), task).to(dev)
train = bench.synth(spec, 1000, 1, dev); query = bench.synth(spec, 100, 2, dev)
fargs, sargs = FactorArguments(use_empirical_fisher=True), ScoreArguments()
_, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs)
eig = perform_eigendecomposition(cov, model, state, fargs)
_, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs, eig)
factors = {**eig, **lam}
dfactors = {k: {n: v.to(dev) for n, v in d.items()} for k, d in factors.items()}

T = {}
def timeit(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    T[name] = T.get(name, 0) + time.perf_counter() - t0; return out

orig_prepare, orig_dot, orig_setf = pw.prepare_modules, pw.compute_dot_products_with_loader, pw.set_factors
pw.prepare_modules = lambda *a, **k: timeit('prepare_modules', lambda: orig_prepare(*a, **k))
pw.compute_dot_products_with_loader = lambda *a, **k: timeit('train_pass_total', lambda: orig_dot(*a, **k))
pw.set_factors = lambda *a, **k: timeit('set_factors', lambda: orig_setf(*a, **k))
orig_gather = dp.gather_score_blocks
dp.gather_score_blocks = lambda *a, **k: timeit('gather_cpu', lambda: orig_gather(*a, **k))
for which, f in (('cpu_factors', factors), ('gpu_factors', dfactors)):
    for it in range(3):
        T.clear()
        t = timeit('step', lambda: pw.compute_pairwise_scores_with_loaders(f, model, state, task, ResidentLoader(query, 100), 100, ResidentLoader(train, 1000), sargs, fargs, None))
    print(which, {k: round(v*1e3, 2) for k, v in T.items()})

# finer: time each ops.precondition call and the pieces around
import kronfluence_amd.module.tracker.precondition as pt
orig_prec = ops.precondition
def timed_prec(*a, **k):
    return timeit('ops.precondition', lambda: orig_prec(*a, **k))
pt.ops.precondition = timed_prec
orig_acc = pw.accumulate_iterations
pw.accumulate_iterations = lambda *a, **k: timeit('accumulate_iterations', lambda: orig_acc(*a, **k))
orig_final = pw.finalize_all_iterations
pw.finalize_all_iterations = lambda *a, **k: timeit('finalize_all', lambda: orig_final(*a, **k))
orig_setmode = pw.set_mode
pw.set_mode = lambda *a, **k: timeit('set_mode', lambda: orig_setmode(*a, **k))
for it in range(3):
    T.clear()
    t = timeit('step', lambda: pw.compute_pairwise_scores_with_loaders(dfactors, model, state, task, ResidentLoader(query, 100), 100, ResidentLoader(train, 1000), sargs, fargs, None))
print('fine', {k: round(v*1e3, 2) for k, v in T.items()})
