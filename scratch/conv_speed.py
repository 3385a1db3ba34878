import sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '.')
import bench
dev='cuda:0'
def run(label, bench_flag, cl):
    torch.backends.cudnn.benchmark = bench_flag
    torch.manual_seed(0)
    model = bench.resnet9().to(dev).eval()
    for p in model.parameters(): p.requires_grad_(False)
    x = torch.randn(1000,3,32,32,device=dev); y = torch.randint(0,10,(1000,),device=dev)
    if cl:
        model = model.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    # make the first conv output require grad like the tracked wrapper does
    c = torch.zeros(1, device=dev, requires_grad=True)
    def step():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            h = model[0][0](x) + c
            h = model[0][2](model[0][1](h))
            for m in list(model)[1:]: h = m(h)
            loss = F.cross_entropy(h.float(), y, reduction='sum')
        loss.backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); print(label, 'fwd+bwd ms', (time.perf_counter()-t0)/5*1e3)
run('default', False, False)
run('benchmark', True, False)
run('channels_last', False, True)
run('channels_last+benchmark', True, True)
