"""Every op registered with the PyTorch dispatcher (`torch.ops.rfmi.*`, rosettafold-pytorch_amd/custom_ops.py) against
the direct C-ABI path (`ops.*`) on the same inputs -- bitwise, both routes end in the same kernel -- and through
`torch.library.opcheck` (schema, fake-tensor shapes / dtypes / strides, no hidden mutation of the inputs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import rosettafold_pytorch_amd as R  # noqa: E402
from rosettafold_pytorch_amd import _lib as L  # noqa: E402
from rosettafold_pytorch_amd import custom_ops, ops  # noqa: E402

DEV = "cuda"
BF16, F32 = torch.bfloat16, torch.float32


def rn(*s, seed=0, dtype=F32, scale=1.0):
    g = torch.Generator().manual_seed(seed + len(s) + sum(s))
    return (torch.randn(*s, generator=g) * scale).to(DEV).to(dtype)


def xyz_trace(b, l, seed=3):
    g = torch.Generator().manual_seed(seed)
    steps = torch.randn(b, l, 3, generator=g)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)
    xyz = ca[:, :, None, :] + 0.5 * torch.randn(b, l, 3, 3, generator=g)
    xyz[:, :, 1] = ca
    return xyz.to(DEV)


def _cases():
    """name -> (args of the dispatcher op, direct-path callable returning the same output(s))"""
    torch.manual_seed(0)
    c = {}
    x, w, b = rn(64, 96, dtype=BF16), rn(48, 96, seed=1, dtype=BF16, scale=0.2), rn(48, seed=2)
    c["linear"] = ((x, w, b, L.ACT_RELU, False), lambda: ops.linear(x, w, b, act=L.ACT_RELU, out_dtype=BF16))
    xl, g, be = rn(40, 288), 1 + 0.1 * rn(288, seed=3), 0.1 * rn(288, seed=4)
    c["layernorm"] = ((xl, g, be, 1e-5, False), lambda: ops.layernorm(xl, g, be, eps=1e-5, out_dtype=BF16))
    q, k, v = (rn(1, 16, 64, 4, 32, seed=s_, dtype=BF16, scale=0.4) for s_ in (5, 6, 7))
    c["tied_row_attention"] = ((q, k, v), lambda: ops.tied_row_attention(q, k, v))
    m = R.PerformerSelfAttention(dim=96, heads=2, generalized_attention=True).to(DEV)
    qkv, pc = rn(6, 64, 3 * 128, seed=8, dtype=BF16), m.proj_scaled()

    def favor_direct():
        o = torch.empty(6, 64, 128, device=DEV, dtype=BF16)
        ops.favor_attention(qkv, pc, o, (0, 64 * 384, 384, 64), (0, 64 * 128, 128), 0, 128, 256, 1, 6, 2, 64, 64, 266, False, 1e-3)
        return o
    c["performer_attention"] = ((qkv, pc, 2, False), favor_direct)
    xo, yo = rn(1, 64, 16, 32, seed=9, dtype=BF16), rn(1, 64, 16, 32, seed=10, dtype=BF16, scale=0.1)
    go, bo, wo, bio = 1 + 0.1 * rn(1024, seed=11), 0.1 * rn(1024, seed=12), rn(288, 1024, seed=13, scale=0.03), rn(288, seed=14)
    c["outer_product_ln_linear"] = ((xo, yo, go, bo, wo, bio, 1e-5), lambda: ops.outer_product_ln_linear(xo, yo, go, bo, wo, bio, 1e-5))
    xc, wc = rn(1, 16, 16, 32, seed=15, dtype=BF16), rn(32, 3, 3, 32, seed=16, dtype=BF16, scale=0.1)

    def conv_direct():
        out = torch.empty(1, 16, 16, 32, device=DEV, dtype=BF16)
        return ops.gemm(xc, wc.reshape(32, 288), out, 256, 32, 288, conv=(1, 16, 16, 32, 2))
    c["conv3x3_nhwc"] = ((xc, wc, 2), conv_direct)
    xi, gi, bi = rn(2, 12, 12, 32, seed=17, dtype=BF16), 1 + 0.1 * rn(32, seed=18), 0.1 * rn(32, seed=19)
    c["instance_norm_elu"] = ((xi, gi, bi, 1e-6, True), lambda: ops.instnorm(xi, gi, bi, eps=1e-6, act=L.ACT_ELU, out_dtype=F32)[0])
    xyz, aa = xyz_trace(2, 48), torch.arange(48, device=DEV).unsqueeze(0).repeat(2, 1)
    c["knn_mask"] = ((xyz, aa, 16, 9), lambda: ops.knn_mask(xyz, aa, 16, 9))
    return c


CASES = None


def cases():
    global CASES
    if CASES is None:
        CASES = _cases()
    return CASES


def test_every_registered_op_has_a_case():
    assert sorted(custom_ops.OPS) == sorted(cases())


@pytest.mark.parametrize("name", custom_ops.OPS)
def test_dispatcher_route_equals_direct_route(name):
    args, direct = cases()[name]
    got = getattr(torch.ops.rfmi, name)(*args)
    ref = direct()
    got = got if isinstance(got, (tuple, list)) else (got,)
    ref = ref if isinstance(ref, (tuple, list)) else (ref,)
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert a.dtype == b.dtype and a.shape == b.shape
        assert torch.equal(a, b), name


@pytest.mark.parametrize("name", custom_ops.OPS)
def test_opcheck(name):
    args, _ = cases()[name]
    # (no autograd registration: inference-only ops; the remaining utilities check the schema, the fake kernel and aliasing)
    torch.library.opcheck(getattr(torch.ops.rfmi, name).default, args,
                          test_utils=("test_schema", "test_faketensor"))


def test_cpu_tensors_are_rejected_by_the_dispatcher():
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.rfmi.layernorm(torch.zeros(4, 32), torch.ones(32), torch.zeros(32), 1e-5, False)
