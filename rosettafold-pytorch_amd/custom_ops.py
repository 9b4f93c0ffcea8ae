"""PyTorch dispatcher registration of the librfmi.so op groups (`torch.library.custom_op`, namespace `rfmi`).

SURVEY 8(b) / north_star: "launched from Python via PyTorch-ROCm custom ops".  Every op below is a thin functional
wrapper over the C ABI (ops.py -> include/rfmi.h): device type "cuda" only (a CPU tensor raises NotImplementedError from
the dispatcher -- there is no CPU kernel), a device guard around the launch (the kernel goes to the tensor's GPU and its
current stream), and a fake ("meta") implementation for shape inference, so the ops compose with FakeTensor tracing and
`torch.library.opcheck`.

"bf16" below = the 16-bit operand type of the active library (bfloat16, or float16 after
set_compute_dtype(torch.float16)).  The model's own forward calls the same C entry points through ops.py directly: one dispatcher hop costs ~15-20 us of
host time per op (tools/custom_op_overhead.py), the forward issues ~4,000 launches per step and some of them run for
10-30 us, so routing the hot path through the dispatcher would make it host-bound.  Both routes end in the same kernels.

    torch.ops.rfmi.linear(x, w, bias, act, out_fp32)                  nn.Linear (+ReLU/ELU)          rf.py:270-281
    torch.ops.rfmi.layernorm(x, gamma, beta, eps, out_fp32)          nn.LayerNorm                   rf.py:323 ...
    torch.ops.rfmi.tied_row_attention(q, k, v)                       logits + softmax + A.V + sym   rf.py:252-265
    torch.ops.rfmi.performer_attention(qkv, proj, heads, softmax_kernel)   FAVOR+ linear attention  rf.py:313-318,505-518
    torch.ops.rfmi.outer_product_ln_linear(x, y, gamma, beta, w, b, eps)   OuterProductMean         rf.py:412-427
    torch.ops.rfmi.conv3x3_nhwc(x, w, dilation)                      3x3 'same' conv, NHWC           rf.py:452,456
    torch.ops.rfmi.instance_norm_elu(x, gamma, beta, eps, elu)       InstanceNorm2d(affine) (+ELU)   rf.py:453,457
    torch.ops.rfmi.knn_mask(xyz, aa_idx, k, kmin)                    kNN + sequence-band adjacency   rf.py:823-852
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import ops

F32 = torch.float32


def _guard(t):
    return torch.cuda.device(t.device)


@torch.library.custom_op("rfmi::linear", mutates_args=(), device_types="cuda")
def linear(x: Tensor, w: Tensor, bias: Optional[Tensor], act: int, out_fp32: bool) -> Tensor:
    with _guard(x):
        return ops.linear(x.contiguous(), w.contiguous(), bias, act=act, out_dtype=F32 if out_fp32 else x.dtype)


@linear.register_fake
def _(x, w, bias, act, out_fp32):
    return x.new_empty(*x.shape[:-1], w.shape[0], dtype=F32 if out_fp32 else x.dtype)


@torch.library.custom_op("rfmi::layernorm", mutates_args=(), device_types="cuda")
def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, out_fp32: bool) -> Tensor:
    with _guard(x):
        return ops.layernorm(x.contiguous(), gamma, beta, eps=eps, out_dtype=F32 if out_fp32 else ops.h16())


@layernorm.register_fake
def _(x, gamma, beta, eps, out_fp32):
    return x.new_empty(x.shape, dtype=F32 if out_fp32 else ops.h16())


@torch.library.custom_op("rfmi::tied_row_attention", mutates_args=(), device_types="cuda")
def tied_row_attention(q: Tensor, k: Tensor, v: Tensor) -> Tuple[Tensor, Tensor]:
    """q (already scaled by the position weights and d_h^-0.5, rf.py:252), k, v: bf16 [B, N, L, H, 32].
    Returns (out bf16 [B, N, L, H*32], symmetrised attention map fp32 [B, L, L, H])."""
    with _guard(q):
        return ops.tied_row_attention(q, k, v)


@tied_row_attention.register_fake
def _(q, k, v):
    B, N, Lr, H, dh = q.shape
    return q.new_empty(B, N, Lr, H * dh), q.new_empty(B, Lr, Lr, H, dtype=F32)


@torch.library.custom_op("rfmi::performer_attention", mutates_args=(), device_types="cuda")
def performer_attention(qkv: Tensor, proj: Tensor, heads: int, softmax_kernel: bool) -> Tensor:
    """qkv bf16 [S, n, 3*heads*64] (q | k | v), proj bf16 [288, 64] (pre-scaled by 64^-1/4, rows >= 266 zero; times log2 e
    for the softmax kernel).  Returns bf16 [S, n, heads*64]."""
    S, n, W3 = qkv.shape
    inner = W3 // 3
    with _guard(qkv):
        out = torch.empty(S, n, inner, device=qkv.device, dtype=qkv.dtype)
        ops.favor_attention(qkv.contiguous(), proj.contiguous(), out, (0, n * W3, W3, 64), (0, n * inner, inner), 0, inner,
                            2 * inner, 1, S, heads, n, 64, 266, softmax_kernel, 1e-4 if softmax_kernel else 1e-3)
    return out


@performer_attention.register_fake
def _(qkv, proj, heads, softmax_kernel):
    return qkv.new_empty(qkv.shape[0], qkv.shape[1], qkv.shape[2] // 3)


@torch.library.custom_op("rfmi::outer_product_ln_linear", mutates_args=(), device_types="cuda")
def outer_product_ln_linear(x: Tensor, y: Tensor, gamma: Tensor, beta: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """x, y bf16 [B, N, L, 32] -> fp32 [B, L, L, d_out]: Linear(LayerNorm(sum_n x[n,i,:] (x) y[n,j,:])), rf.py:415-427."""
    with _guard(x):
        return ops.outer_product_ln_linear(x, y, gamma, beta, w, b, eps)


@outer_product_ln_linear.register_fake
def _(x, y, gamma, beta, w, b, eps):
    B, N, Lr, P = x.shape
    return x.new_empty(B, Lr, Lr, w.shape[0], dtype=F32)


@torch.library.custom_op("rfmi::conv3x3_nhwc", mutates_args=(), device_types="cuda")
def conv3x3_nhwc(x: Tensor, w: Tensor, dilation: int) -> Tensor:
    """x bf16/fp32 NHWC [B,H,W,C], w [Co, 3, 3, C] (same dtype): zero-padded 'same' convolution as implicit GEMM."""
    B, Hh, Ww, Cc = x.shape
    Co = w.shape[0]
    with _guard(x):
        out = torch.empty(B, Hh, Ww, Co, device=x.device, dtype=x.dtype)
        ops.gemm(x.contiguous(), w.reshape(Co, 9 * Cc).contiguous(), out, B * Hh * Ww, Co, 9 * Cc, conv=(B, Hh, Ww, Cc, dilation))
    return out


@conv3x3_nhwc.register_fake
def _(x, w, dilation):
    return x.new_empty(*x.shape[:3], w.shape[0])


@torch.library.custom_op("rfmi::instance_norm_elu", mutates_args=(), device_types="cuda")
def instance_norm_elu(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, elu: bool) -> Tensor:
    with _guard(x):
        return ops.instnorm(x.contiguous(), gamma, beta, eps=eps, act=L.ACT_ELU if elu else L.ACT_NONE, out_dtype=F32)[0]


@instance_norm_elu.register_fake
def _(x, gamma, beta, eps, elu):
    return x.new_empty(x.shape, dtype=F32)


@torch.library.custom_op("rfmi::knn_mask", mutates_args=(), device_types="cuda")
def knn_mask(xyz: Tensor, aa_idx: Tensor, k: int, kmin: int) -> Tensor:
    with _guard(xyz):
        return ops.knn_mask(xyz.float().contiguous(), aa_idx.contiguous(), k, kmin)


@knn_mask.register_fake
def _(xyz, aa_idx, k, kmin):
    return xyz.new_empty(xyz.shape[0], xyz.shape[1], xyz.shape[1], dtype=torch.uint8)


OPS = ("linear", "layernorm", "tied_row_attention", "performer_attention", "outer_product_ln_linear", "conv3x3_nhwc",
       "instance_norm_elu", "knn_mask")
