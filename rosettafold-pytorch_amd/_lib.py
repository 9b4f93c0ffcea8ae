"""ctypes binding of librfmi.so / librfmi_f16.so (include/rfmi.h).  The product has NO fallback: if a HIP
library is missing or a symbol is absent this module raises at import."""
import ctypes as C
import os

import torch  # noqa: F401  -- must be loaded first: librfmi.so binds to the HIP runtime torch already brought in

_HERE = os.path.dirname(os.path.abspath(__file__))
# RFMI_LIB: tuning hook -- load another build of the bf16 library (e.g. an ablation build, csrc/Makefile ABLATION=1 LIB=...)
LIB_PATH = os.environ.get("RFMI_LIB") or os.path.join(_HERE, "librfmi.so")

RF_F32, RF_BF16, RF_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_ELU, ACT_RELU_EPS, ACT_LEAKY, ACT_BLOCK_LN32 = 0, 1, 2, 3, 4, 5
BIAS_NONE, BIAS_COL, BIAS_ROW = 0, 1, 2
AMODE_PLAIN, AMODE_CONV3X3 = 0, 1

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
u64 = C.c_uint64


class GemmDesc(C.Structure):
    """struct rf_gemm_desc (include/rfmi.h)."""
    _fields_ = [
        ("M", i32), ("N", i32), ("K", i32),
        ("nb0", i32), ("nb1", i32), ("nb2", i32),
        ("ab_dtype", i32), ("c_dtype", i32), ("kc", i32), ("a_mode", i32),
        ("a_rc", i32), ("b_rc", i32), ("c_rc", i32), ("c_cc", i32),
        ("a_bs", i64 * 3), ("a_ro", i64), ("a_ri", i64), ("a_ko", i64),
        ("b_bs", i64 * 3), ("b_ro", i64), ("b_ri", i64), ("b_ko", i64),
        ("c_bs", i64 * 3), ("c_ro", i64), ("c_ri", i64), ("c_co", i64),
        ("conv_n", i32), ("conv_h", i32), ("conv_w", i32), ("conv_c", i32), ("conv_dil", i32),
        ("bias_mode", i32), ("act", i32), ("act_nvalid", i32),
        ("act_eps", f32), ("alpha", f32), ("tile_cfg", i32),
        ("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("residual", vp),
        ("ln_out", vp), ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32), ("reserved_", i32),
        ("rs", vp), ("rs_bstride", i64), ("rs_rpb", i32), ("rs_cg", i32), ("rs_ncols", i32), ("rs_alpha", f32),
    ]


I64x4 = i64 * 4
I64x3 = i64 * 3

# name -> argtypes (restype is always int unless noted); mirrors include/rfmi.h declaration order
PROTOTYPES = {
    "rf_gemm": [C.POINTER(GemmDesc), vp],
    "rf_layernorm": [vp, i32, i64, vp, i32, i64, i64, i32, vp, vp, f32, i32, i32, vp],
    "rf_sym_layernorm": [vp, vp, i32, i32, i32, i32, f32, vp],
    "rf_softmax": [vp, i64, i64, vp, i32, i64, i64, i32, f32, vp],
    "rf_softmax_batched": [vp, i64, i64, i64, vp, i32, i64, i64, i64, i32, f32, i32, vp],
    "rf_tied_softmax": [vp, vp, i32, vp, i64, i32, i32, i32, vp],
    "rf_tied_logits_softmax": [vp, vp, i64, i64, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp],
    "rf_tied_av": [vp, vp, C.POINTER(I64x4), vp, C.POINTER(I64x4), i32, i32, i32, i32, i32, vp],
    "rf_tied_logits": [vp, vp, C.POINTER(I64x4), vp, C.POINTER(I64x3), f32, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, vp],
    "rf_tied_attention": [vp, vp, vp, C.POINTER(I64x4), C.POINTER(I64x4), vp, C.POINTER(I64x3), f32, vp, vp, i64, vp,
                          C.POINTER(I64x4), i32, i32, i32, i32, i32, vp, i64, vp],
    "rf_ffn_fused": [vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, f32, i64, i32, i32, vp],
    "rf_outer_product_ln_linear": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, f32, vp, i64, vp],
    "rf_outer_product_pairs": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, f32, vp, i64, vp],
    "rf_poswise_collapsed": [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp],
    "rf_poswise": [vp, i32, i64, vp, i64, i32, i32, i32, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp],
    "rf_weighted_msa_sum": [vp, i32, vp, vp, i64, i32, i32, i32, i32, vp],
    "rf_instnorm_ws_bytes": [i32, i64, i32],  # returns int64
    "rf_instnorm_stats": [vp, i32, vp, i32, i64, i32, vp, i64, vp],
    "rf_instnorm_apply": [vp, i32, vp, vp, vp, f32, vp, i32, vp, i32, vp, i32, i32, i64, i32, vp],
    "rf_instnorm_mean": [vp, vp, i32, i64, i32, vp],
    "rf_channel_mean_ws_bytes": [i32, i64, i32],  # returns int64
    "rf_channel_mean": [vp, vp, i32, i64, i32, vp, i64, vp],
    "rf_sample_mean": [vp, i32, vp, i32, i64, i32, i32, vp],
    "rf_center_apply": [vp, vp, vp, i32, i32, i64, i32, vp],
    "rf_center_rows": [vp, vp, i32, i32, i32, vp],
    "rf_fold_mean": [vp, i64, i32, i32, i32, i64, i32, vp, vp, vp, i32, i32, vp],
    "rf_conv3x3_border_fix": [vp, i32, vp, i32, i32, i32, i32, i32, i32, vp],
    "rf_msa_embed": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "rf_pair_embed": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rf_copy4d": [vp, i32, C.POINTER(I64x4), vp, i32, C.POINTER(I64x4), C.POINTER(I64x4), vp],
    "rf_axpby": [vp, i32, f32, vp, i32, f32, vp, i32, i64, vp],
    "rf_favor_softmax_features": [vp, vp, C.POINTER(I64x4), i32, i32, vp, i32, i64, i32, i32, i32, i32, i32, i32, f32, vp],
    "rf_favor_attention": [vp, vp, vp, C.POINTER(I64x4), C.POINTER(I64x3), i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "rf_linattn_normalize": [vp, i64, vp, i32, i64, i64, i32, vp],
    "rf_tile_1d_feats": [vp, vp, i32, i64, i32, i32, i32, i32, vp],
    "rf_graph_attention": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, f32, vp],
    "rf_graph_attention_dropout": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, f32, f32, u64, u64, vp],
    "rf_dropout": [vp, vp, i32, f32, u64, u64, i64, vp],
    "rf_dist_masked_attention": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "rf_knn_mask": [vp, vp, vp, i32, i32, i32, i32, vp],
    "rf_edges_from_mask": [vp, vp, vp, vp, vp, vp, i32, i32, i64, vp],
    "rf_se3_edge_geometry": [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, vp],
    "rf_se3_message": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, vp],
    "rf_se3_radial_message_supported": [i32, i32, i32, i32, i32],
    "rf_se3_radial_message": [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i64, vp],
    "rf_se3_attention": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, vp],
    "rf_se3_norm_bias": [vp, vp, vp, i64, i32, i32, vp],
    "rf_se3_gram": [vp, vp, i64, i32, i32, vp],
    "rf_se3_attn_apply": [vp, vp, vp, i64, i32, i32, i32, vp],
    "rf_coord_apply": [vp, vp, vp, i64, vp],
    "rf_center_ca": [vp, vp, i64, vp],
    "rf_scale_rows": [vp, vp, i32, vp, i64, i32, vp],
    "rf_fill": [vp, i32, f32, i64, vp],
    "rf_check_inputs": [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, vp],
    "rf_onehot": [vp, vp, i32, i64, i32, i32, i64, vp],
    "rf_seqsep_feature": [vp, vp, i32, i64, i32, i32, i32, vp],
    "rf_add_pos_enc": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "rf_debug_gemm_stamps": [vp],
    "rf_debug_gemm_fast_stamps": [vp],
    "rf_gemm_last_family": [],
    "rf_version": [],
    "rf_h16_dtype": [],
}

def _load(path):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C rosettafold-pytorch_amd/csrc`).  There is no CPU / PyTorch fallback.")
    handle = C.CDLL(path)  # RTLD_LOCAL: the two builds export the same symbol names and never see each other
    for _name, _args in PROTOTYPES.items():
        _fn = getattr(handle, _name)  # AttributeError if the library does not export it
        _fn.argtypes = _args
        _fn.restype = C.c_int
    handle.rf_instnorm_ws_bytes.restype = C.c_int64
    handle.rf_channel_mean_ws_bytes.restype = C.c_int64
    handle.rf_build_info.restype = C.c_char_p
    handle.rf_build_info.argtypes = []
    return handle


LIB16_PATH = os.path.join(_HERE, "librfmi_f16.so")
# librfmi.so: 16-bit MFMA operand type bfloat16 (accepts RF_F32 / RF_BF16); librfmi_f16.so: the same sources built with
# IEEE fp16 as the 16-bit type (accepts RF_F32 / RF_F16).  Both are required: a missing library is an import error.
LIBS = {RF_BF16: _load(LIB_PATH), RF_F16: _load(LIB16_PATH)}
for _code, _h in LIBS.items():
    if _h.rf_h16_dtype() != _code:
        raise ImportError(f"{_h.rf_build_info().decode()} reports 16-bit dtype code {_h.rf_h16_dtype()}, expected {_code}")


class _ActiveLibrary:
    """`lib.rf_xxx(...)` goes to the library selected by select_h16() (model.set_compute_dtype): the bfloat16 build by
    default.  Attribute writes (bench.py wraps rf_gemm to time it) land on the selected library too."""

    def __init__(self):
        object.__setattr__(self, "_code", RF_BF16)

    def __getattr__(self, name):
        return getattr(LIBS[object.__getattribute__(self, "_code")], name)

    def __setattr__(self, name, value):
        setattr(LIBS[object.__getattribute__(self, "_code")], name, value)


lib = _ActiveLibrary()


def select_h16(code):
    """Route every following call to the build whose 16-bit type is `code` (RF_BF16 or RF_F16)."""
    if code not in LIBS:
        raise ValueError(f"no library for 16-bit dtype code {code}")
    object.__setattr__(lib, "_code", code)


def active_h16():
    return object.__getattribute__(lib, "_code")


class RfmiError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        kind = {-1: "RF_EINVAL (inconsistent arguments)", -2: "RF_EALIGN (misaligned pointer/stride)"}.get(
            rc, f"hipError_t {rc}")
        raise RfmiError(f"{what} failed: {kind}")
