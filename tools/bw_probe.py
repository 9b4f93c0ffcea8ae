import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
x = torch.randn(1, 256, 256, 288, device='cuda')
mean = ops.channel_mean(x)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
y16 = torch.empty_like(x, dtype=torch.bfloat16)
y32 = torch.empty_like(x)
print("torch copy fp32->bf16  us", t(lambda: y16.copy_(x)))
print("torch copy fp32->fp32  us", t(lambda: y32.copy_(x)))
print("torch sum over pixels  us", t(lambda: x.sum(dim=(1, 2))))
print("center_apply -> bf16   us", t(lambda: ops.center_apply(x, mean, out=y16)))
print("center_apply -> fp32   us", t(lambda: ops.center_apply(x, mean, out=y32)))
print("channel_mean           us", t(lambda: ops.channel_mean(x)))
print("cast (axpby)           us", t(lambda: ops.axpby(x, 1.0, None, 0.0, y16)))
g = torch.ones(288, device='cuda'); b_ = torch.zeros(288, device='cuda')
print("layernorm -> bf16      us", t(lambda: ops.layernorm(x, g, b_, out=y16)))
