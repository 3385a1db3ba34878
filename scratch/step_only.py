import sys, time, torch
sys.path.insert(0, '.')
import bench
from kronfluence_amd import FactorArguments, ScoreArguments, ops, prepare_model
from kronfluence_amd.score import pairwise as pw
from kronfluence_amd.utils.dataset import ResidentLoader
from kronfluence_amd.utils.state import State
state = State(); dev = state.device
spec = bench.WORKLOADS['mnist_mlp']
task = bench.make_task()
model = prepare_model(spec['model'](), task).to(dev)
train = bench.synth(spec, 1000, 1, dev); query = bench.synth(spec, 100, 2, dev)
fargs, sargs = FactorArguments(use_empirical_fisher=True), ScoreArguments()
# fake but well-formed factors (orthogonal eigenvectors, positive lambda): timing only
factors = {k: {} for k in ('activation_eigenvectors','gradient_eigenvectors','activation_eigenvalues','gradient_eigenvalues','lambda_matrix','num_lambda_processed')}
for name, (o, ip) in zip(['1','3','5','7'], [(1024,785),(1024,1025),(1024,1025),(10,1025)]):
    factors['activation_eigenvectors'][name] = torch.linalg.qr(torch.randn(ip,ip,device=dev))[0].contiguous()
    factors['gradient_eigenvectors'][name] = torch.linalg.qr(torch.randn(o,o,device=dev))[0].contiguous()
    factors['activation_eigenvalues'][name] = torch.rand(ip,device=dev); factors['gradient_eigenvalues'][name] = torch.rand(o,device=dev)
    factors['lambda_matrix'][name] = torch.rand(o,ip,device=dev)+0.1
    factors['num_lambda_processed'][name] = torch.tensor([1000])
def step():
    return pw.compute_pairwise_scores_with_loaders(factors, model, state, task, ResidentLoader(query, 100), 100, ResidentLoader(train, 1000), sargs, fargs, None)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.cuda.synchronize(); t0=time.perf_counter(); step(); torch.cuda.synchronize(); print('step ms', (time.perf_counter()-t0)*1e3)
