import sys, torch
sys.path.insert(0, "/root/repo")
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
torch.manual_seed(0)
B, N, Lr, P, Dout = 1, 64, 64, 32, 288
def randn(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed)).cuda()
g_, b_ = 1.0 + 0.2 * randn(P * P, seed=2), 0.1 * randn(P * P, seed=3)
w, bias = randn(Dout, P * P, seed=4) * 0.05, randn(Dout, seed=5)
for dt in (torch.bfloat16, torch.float16):
    R.set_compute_dtype(dt)
    for off in (0.0, 0.25, 0.5, 1.0, 2.0, 4.0):
        x, y = (randn(B, N, Lr, P) + off).to(dt), (randn(B, N, Lr, P, seed=1) * 0.3 + off).to(dt)
        co = torch.einsum("bniu,bnjv->bijuv", x.float(), y.float()).reshape(B, Lr, Lr, P * P)
        mu, sd = co.mean(-1), co.std(-1)
        ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(co, (P * P,), g_, b_, 1e-5), w, bias)
        out = ops.outer_product_ln_linear(x, y, g_, b_, w, bias, 1e-5)
        e = ((out - ref).abs().max() / ref.abs().max()).item()
        e2 = ((out - ref).norm() / ref.norm()).item()
        print(f"{dt} offset {off}: |mu|/sigma median {(mu.abs() / sd).median().item():.2f} max {(mu.abs() / sd).max().item():.2f}  max-rel {e:.3e} rel-L2 {e2:.3e}", flush=True)
R.set_compute_dtype(torch.bfloat16)
