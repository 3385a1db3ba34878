import sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '.')
import bench
from kronfluence_amd import prepare_model
from kronfluence_amd.module.utils import set_mode
dev='cuda:0'
task = bench.make_task()
model = prepare_model(bench.resnet9(), task).to(dev)
def timeit(label, bs, mode=None, n=3):
    x = torch.randn(bs,3,32,32,device=dev); y = torch.randint(0,10,(bs,),device=dev)
    if mode: set_mode(model, mode, release_memory=True)
    def step():
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = task.compute_train_loss((x,y), model)
        loss.backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); print(label, bs, 'ms', (time.perf_counter()-t0)/n*1e3, flush=True)
timeit('default-mode', 1000)
timeit('default-mode', 250)
timeit('covariance', 1000, 'covariance')
set_mode(model, 'default', release_memory=True)
# which conv layers are slow? time each tracked conv fwd+bwd alone at bs=1000 in bf16 autocast
x = torch.randn(1000,3,32,32,device=dev)
with torch.autocast('cuda', dtype=torch.bfloat16):
    h = x
    from kronfluence_amd.module.tracked_module import TrackedModule
    def walk(mod, h):
        return mod(h)
    feats = []
    hooks = []
    def mk(name):
        def hook(m, inp, out):
            feats.append((name, inp[0].detach(), m))
        return hook
    for m in model.modules():
        if isinstance(m, TrackedModule): hooks.append(m.register_forward_hook(mk(m.name)))
    model(x)
for hk in hooks: hk.remove()
for name, inp, m in feats:
    inp = inp.clone().requires_grad_(True)
    def step():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = m(inp)
        out.float().sum().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); print('layer', name, tuple(inp.shape), inp.dtype, inp.is_contiguous(), 'ms', (time.perf_counter()-t0)/3*1e3, flush=True)
