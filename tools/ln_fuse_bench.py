import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosettafold_pytorch_amd import ops
import torch.nn as nn
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for M, N, K in [(262144, 288, 512), (262144, 288, 1152), (131072, 384, 1536), (131072, 384, 768), (131072, 384, 384)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda"); lnm = nn.LayerNorm(N).cuda()
    t_f = timeit(lambda: ops.linear_residual_ln(x, w, b, res, lnm))
    def unf():
        ops.linear(x, w, b, out=res, residual=res)
        ops.layernorm(res, lnm.weight.detach(), lnm.bias.detach(), out_dtype=torch.bfloat16)
    t_u = timeit(unf)
    t_g = timeit(lambda: ops.linear(x, w, b, out=res, residual=res))
    print(f"M={M} N={N} K={K}: fused {t_f:.3f} ms | gemm {t_g:.3f} + ln {t_u - t_g:.3f} = {t_u:.3f} ms", flush=True)
