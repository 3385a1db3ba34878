"""Where does a ResNet-9 pairwise stage go?  Times the tracker-level ops with HIP events, the model's own
forward/backward alone, and every ``kf_pairwise_score`` launch of one train batch.

    gpurun -- 'python tools/breakdown.py'          (N=4000 train x 1000 query by default; env N=... to change)

Source of the "where the step goes" table in profiles/README.md."""
import collections, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from kronfluence_amd import FactorArguments, ScoreArguments, ops, prepare_model
from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader, perform_eigendecomposition
from kronfluence_amd.score.pairwise import compute_pairwise_scores_with_loaders
from kronfluence_amd.utils.dataset import ResidentLoader
from kronfluence_amd.utils.state import State

n_train = int(os.environ.get("N", 4000)); n_query = 1000
state = State(); dev = state.device
spec = bench.WORKLOADS["resnet9"]
torch.manual_seed(0)
task = bench.make_task()
model = prepare_model(spec["model"](), task).to(dev)
if os.environ.get("CL"):
    model = model.to(memory_format=torch.channels_last)
train = bench.synth(spec, n_train, 1, dev); query = bench.synth(spec, n_query, 2, dev)
amp = torch.bfloat16
fargs = FactorArguments(use_empirical_fisher=True, amp_dtype=amp, per_sample_gradient_dtype=torch.bfloat16, lambda_dtype=torch.bfloat16)
sargs = ScoreArguments(amp_dtype=amp, query_gradient_accumulation_steps=4, score_dtype=torch.bfloat16, precondition_dtype=torch.bfloat16)
_, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs)
eig = perform_eigendecomposition(cov, model, state, fargs)
_, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train, 1000), fargs, eig)
factors = {k: {n: v.to(dev) for n, v in d.items()} for k, d in {**eig, **lam}.items()}

def step():
    return compute_pairwise_scores_with_loaders(factors, model, state, task, ResidentLoader(query, 250), 250, ResidentLoader(train, 1000), sargs, fargs, None)
step()
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); base = time.perf_counter() - t0
print(f"untimed-hooks step: {base*1e3:.1f} ms for {n_train} train x {n_query} query")

log = collections.defaultdict(list)
def wrap(name):
    fn = getattr(ops, name)
    def inner(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = fn(*a, **k); e.record(); log[name].append((s, e)); return out
    setattr(ops, name, inner)
for name in ("im2col", "per_sample_gradient", "pairwise_score", "precondition", "k_tile_major", "matmul_nn", "cast"):
    wrap(name)
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); total = time.perf_counter() - t0
print(f"instrumented step: {total*1e3:.1f} ms")
for name, evs in log.items():
    ms = sum(s.elapsed_time(e) for s, e in evs)
    print(f"  {name:22s} {len(evs):5d} calls {ms:9.2f} ms")
# model alone
from kronfluence_amd.module.utils import set_mode
set_mode(model, "default", release_memory=True)
x, y = train[0][:1000], train[1][:1000]
def fb():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=amp):
        loss = task.compute_train_loss((x, y), model)
    loss.backward()
for _ in range(3): fb()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fb()
torch.cuda.synchronize(); print(f"model fwd+bwd alone (1000 imgs): {(time.perf_counter()-t0)*100:.2f} ms/batch")
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        with torch.autocast("cuda", dtype=amp):
            model(x)
    torch.cuda.synchronize(); print(f"model fwd alone: {(time.perf_counter()-t0)*100:.2f} ms/batch")
# per-layer pairwise_score timing on one train batch
ops.EVENT_LOG = {}
set_mode(model, "default", release_memory=True)
def step1():
    return compute_pairwise_scores_with_loaders(factors, model, state, task, ResidentLoader(query, 250), 250, ResidentLoader((train[0][:1000], train[1][:1000]), 1000), sargs, fargs, None)
step1(); ops.EVENT_LOG = {}
step1(); torch.cuda.synchronize()
for i, (s, e, f, _) in enumerate(ops.EVENT_LOG['pairwise_score']):
    ms = s.elapsed_time(e)
    print(f"  score call {i}: {ms:.3f} ms  {f/ms/1e9:.0f} TFLOP/s  ({f/1e9:.1f} GF)")
