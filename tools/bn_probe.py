import torch, time
from torch.profiler import profile, ProfilerActivity
bn = torch.nn.BatchNorm2d(64).cuda().eval()
x = torch.randn(1000, 64, 32, 32, device="cuda", dtype=torch.bfloat16)
def run(tag, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    print(tag, [(e.key[:60], e.count, round(e.self_device_time_total / e.count, 1)) for e in prof.key_averages() if e.self_device_time_total > 0])
with torch.no_grad():
    run("module fp32-params bf16 input (autocast off)", lambda: bn(x.float()))
    run("torch.batch_norm cudnn False fp32", lambda: torch.batch_norm(x.float(), bn.weight, bn.bias, bn.running_mean, bn.running_var, False, 0.0, bn.eps, False))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        run("autocast module", lambda: bn(x))
        run("autocast torch.batch_norm cudnn False", lambda: torch.batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, False, 0.0, bn.eps, False))
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).view(1, -1, 1, 1)
        shift = (bn.bias - bn.running_mean * scale.view(-1)).view(1, -1, 1, 1)
        run("autocast addcmul", lambda: torch.addcmul(shift.to(x.dtype), x, scale.to(x.dtype)))
    torch.backends.cudnn.enabled = False
    with torch.autocast("cuda", dtype=torch.bfloat16):
        run("autocast module, cudnn.enabled False globally", lambda: bn(x))
