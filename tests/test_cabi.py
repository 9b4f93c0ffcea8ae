"""CPU-side checks of the drop-in boundary: librfmi.so loads and exports every entry point that include/rfmi.h
declares; the ctypes table binds exactly those; the struct mirrors the C layout.  No kernel is launched."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "rfmi.h")
LIBS = [os.path.join(ROOT, "rosettafold-pytorch_amd", n) for n in ("librfmi.so", "librfmi_f16.so")]


def declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module", params=LIBS, ids=["bf16-build", "f16-build"])
def lib(request):
    """both builds of the kernel sources (16-bit operand type bfloat16 / IEEE fp16) export the whole header"""
    if not os.path.exists(request.param):
        subprocess.run(["make", "-C", os.path.join(ROOT, "rosettafold-pytorch_amd", "csrc"), "-j8"], check=True)
    return ctypes.CDLL(request.param)


def test_every_declared_symbol_is_exported(lib):
    names = declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from rosettafold_pytorch_amd import _lib
    assert sorted(set(_lib.PROTOTYPES) | {"rf_build_info"}) == declared()
    assert _lib.lib.rf_version() >= 1
    # the two builds say which 16-bit dtype code they take, and the selector routes to the matching one
    assert _lib.LIBS[_lib.RF_BF16].rf_h16_dtype() == _lib.RF_BF16 and _lib.LIBS[_lib.RF_F16].rf_h16_dtype() == _lib.RF_F16
    import torch
    import rosettafold_pytorch_amd as R
    try:
        R.set_compute_dtype(torch.float16)
        assert _lib.lib.rf_h16_dtype() == _lib.RF_F16 and b"fp16" in _lib.lib.rf_build_info()
        with pytest.raises(_lib.RfmiError):  # a bfloat16 tensor is never handed to the fp16 build
            R.ops.dcode(torch.bfloat16)
    finally:
        R.set_compute_dtype(torch.bfloat16)
    assert _lib.lib.rf_h16_dtype() == _lib.RF_BF16


def test_gemm_desc_layout_matches_c():
    """compile a tiny C program printing sizeof/offsetof of rf_gemm_desc and compare with the ctypes mirror."""
    from rosettafold_pytorch_amd._lib import GemmDesc
    fields = ["M", "kc", "a_bs", "a_ko", "b_bs", "c_bs", "c_co", "conv_n", "bias_mode", "act_eps", "alpha", "tile_cfg",
              "A", "bias", "residual", "ln_out", "ln_eps", "rs", "rs_bstride", "rs_rpb", "rs_alpha"]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "rfmi.h"\nint main(){printf("%zu", sizeof(rf_gemm_desc));' + \
           "".join(f'printf(" %zu", offsetof(rf_gemm_desc, {f}));' for f in fields) + "return 0;}"
    exe = "/tmp/rf_layout_check"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=prog.encode(), check=True)
    out = subprocess.run([exe], capture_output=True, check=True).stdout.decode().split()
    assert int(out[0]) == ctypes.sizeof(GemmDesc)
    for f, off in zip(fields, out[1:]):
        assert getattr(GemmDesc, f).offset == int(off), f


def test_product_refuses_cpu_tensors():
    """no CPU fallback: the op layer raises instead of silently computing on the host."""
    import torch
    from rosettafold_pytorch_amd import ops, _lib
    with pytest.raises(_lib.RfmiError):
        ops.linear(torch.zeros(4, 8), torch.zeros(4, 8))
    import rosettafold_pytorch_amd as R
    m = R.RoseTTAFold(d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1,
                      n_three_track_blocks=1, n_encoder_layers=1, max_len=32)
    with pytest.raises(_lib.RfmiError):
        m(torch.zeros(1, 4, 8, dtype=torch.long), torch.zeros(1, 8, dtype=torch.long), torch.arange(8)[None])


def test_error_conventions_of_the_reference():
    """tests/test_module.py:156-160 and 134-143 of the reference: same exceptions at the same places."""
    import rosettafold_pytorch_amd as R
    with pytest.raises(AssertionError):
        R.PositionWiseWeightFactor(d_msa=100, n_heads=12)
    with pytest.raises(AssertionError):
        R.SoftTiedAttentionOverResidues(d_msa=100, n_heads=12)
    with pytest.raises(NotImplementedError):
        R.EncoderLayer(tied=False, performer=False)
    with pytest.raises(NotImplementedError):
        R.EncoderLayer(tied=False, performer=True, return_att=True)
