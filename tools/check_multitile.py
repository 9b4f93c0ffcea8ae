"""GPU check at the BENCH sizes (several tiles per persistent workgroup): the fused residual + LayerNorm epilogue of the
persistent GEMM and the fused outer-product kernel (with its LayerNorm(288) tail) against their unfused forms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops

def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()

torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    R.set_compute_dtype(dt)
    for M, N, K in ((262144, 288, 512), (262144, 288, 1152), (131072, 384, 384), (131072, 384, 1536), (65536, 288, 512)):
        x = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(N, K, device="cuda") * 0.1).to(dt)
        b = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda") * 2 + 0.5
        ln = torch.nn.LayerNorm(N).cuda()
        with torch.no_grad():
            ln.weight.normal_(); ln.bias.normal_()
        o1 = res.clone(); xn1 = ops.linear_residual_ln(x, w, b, o1, ln)
        fam = ops.lib.rf_gemm_last_family()
        o2 = res.clone(); ops.linear(x, w, b, out=o2, residual=o2)
        xn2 = ops.layernorm(o2, ln.weight.detach(), ln.bias.detach(), eps=ln.eps, out_dtype=dt)
        bad = (xn1.float() - xn2.float()).abs().amax(1) > 0.05 * xn2.float().abs().max()
        print(f"{dt} fused LN M={M} N={N} K={K}: family {fam}, stream equal {torch.equal(o1, o2)}, LN max-rel {rel(xn1, xn2):.2e}, bad rows {int(bad.sum())}"
              + (f" first bad rows {bad.nonzero()[:6, 0].tolist()}" if bad.any() else ""))
    B, N, L, P, Dout = 4, 128, 256, 32, 288
    m = R.OuterProductMean(P, Dout).cuda()
    xt = torch.randn(B, L, P, N, device="cuda").to(dt)
    yt = (torch.randn(B, L, P, N, device="cuda") * 0.05).to(dt)
    ln2 = R.LayerNorm(Dout).cuda()
    with torch.no_grad():
        ln2.weight.normal_(); ln2.bias.normal_()
    feat = torch.zeros(B, L, L, 720, device="cuda", dtype=dt)
    m.run_into(xt, yt, ln2, feat, 720)
    a = m.run(xt, yt, N)  # fused, fp32 out
    R.RT.fused_outer = False
    bref = m.run(xt, yt, N)
    R.RT.fused_outer = True
    ref = R.model.ln(ln2, bref, out_dtype=dt)
    d = (feat[..., :Dout].float() - ref.float()).abs().amax(-1)
    print(f"{dt} outer fused fp32-out vs unfused: {rel(a, bref):.2e}; LN2 tail vs unfused+LN: {rel(feat[..., :Dout], ref):.2e}; pairs off by > 5%: {int((d > 0.05 * ref.float().abs().max()).sum())}; tail columns untouched: {bool((feat[..., Dout:] == 0).all())}")
R.set_compute_dtype(torch.bfloat16)
