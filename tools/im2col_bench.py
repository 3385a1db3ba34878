"""``kf_im2col`` on the ResNet-9 conv shapes (bf16, 1000 images): exactness against ``F.unfold`` and write bandwidth."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from kronfluence_amd import ops
dev = "cuda:0"
layers = [(3,64,3,1,1,32),(64,128,5,2,2,32),(128,128,3,1,1,16),(128,256,3,1,1,16),(256,256,3,1,1,8),(256,128,3,1,0,8)]
for (cin,cout,k,s,p,hw) in layers:
    conv = nn.Conv2d(cin,cout,k,stride=s,padding=p,bias=False)
    x = torch.randn(1000,cin,hw,hw,device=dev).bfloat16()
    out = ops.im2col(x, conv, False, torch.bfloat16)
    want = F.unfold(x.float(), k, padding=p, stride=s).transpose(1,2)
    assert torch.equal(out.float(), want), "mismatch"
    torch.cuda.synchronize()
    s0,e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(10): out = ops.im2col(x, conv, False, torch.bfloat16)
    e0.record(); torch.cuda.synchronize()
    ms = s0.elapsed_time(e0)/10
    print(f"cin={cin} k={k} s={s} hw={hw}: out {out.numel()*2/1e6:.0f} MB  {ms:.3f} ms  {out.numel()*2/ms/1e9:.2f} TB/s(write)")
