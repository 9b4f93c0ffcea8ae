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
  round 4 (the rest of SURVEY 8(b)'s list):
    torch.ops.rfmi.ffn(xn, w1, b1, w2, b2, residual)                  FeedForward in its residual     rf.py:270-281
    torch.ops.rfmi.poswise_weight(xn, u, scale)                       PositionWiseWeightFactor        rf.py:205-217
    torch.ops.rfmi.pair_bias_attention(logits, v, msa)                MSA <- pair attention           rf.py:588-595
    torch.ops.rfmi.masked_dist_attention(q, k, xyz, bins)             MSA <- coordinates map          rf.py:891-913
    torch.ops.rfmi.graph_transformer_dense(q, k, v, e, scale)         GraphTransformer, dense graph   rf.py:644-661
    torch.ops.rfmi.knn_graph_csc(mask, capacity)                      edge list + dense edge-id map   rf.py:853-856
    torch.ops.rfmi.se3_edge_kernel(feat, net0, net1, basis, h0, h1, src, count, mo, dout, eps)   ea/modules.py:246-325, 612-641
    torch.ops.rfmi.segment_softmax_sum(k0, k1, q0, q1, v0, v1, eid, heads)                       ea/modules.py:738-774
    torch.ops.rfmi.embed_msa(...) / embed_pair(...)                   embeddings                      rf.py:106-181
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


# ---- round 4: the remaining op groups of SURVEY 8(b) -------------------------------------------------------------------------
@torch.library.custom_op("rfmi::ffn", mutates_args=(), device_types="cuda")
def ffn(xn: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, residual: Tensor) -> Tensor:
    """residual + W2 relu(W1 xn + b1) + b2 (FeedForward inside its residual wrapper, rf.py:270-281, 18-28): xn 16-bit [M, D]
    (already layer-normed), w1 [hidden, D], w2 [D, hidden] (16-bit, nn.Linear layout), biases / residual fp32 -> fp32 [M, D].
    One launch (csrc/ffn.hip: hidden activations on chip) when the shape has an instance, else two rf_gemm launches."""
    with _guard(xn):
        out = residual.contiguous().clone()
        D, hidden = w1.shape[1], w1.shape[0]
        if ops.ffn_fused_applies(xn.contiguous(), out, D, hidden):
            ops.ffn_fused(xn.contiguous(), ops.ffn_pack(w1, w2, xn.dtype), b1, b2, out)
        else:
            h = ops.linear(xn.contiguous(), w1.contiguous(), b1, act=L.ACT_RELU)
            ops.linear(h, w2.contiguous(), b2, out=out, residual=out)
        return out


@ffn.register_fake
def _(xn, w1, b1, w2, b2, residual):
    return residual.new_empty(residual.shape, dtype=F32)


@torch.library.custom_op("rfmi::poswise_weight", mutates_args=(), device_types="cuda")
def poswise_weight(xn: Tensor, u: Tensor, scale: float) -> Tensor:
    """PositionWiseWeightFactor in collapsed form (rf.py:205-217): xn 16-bit [B, N, L, D], u[b, l, h, :] = W_k,h^T to_q(x_0)
    16-bit [B, L, H, D] -> w fp32 [B, H, N, L] = softmax_n(scale * xn . u)."""
    with _guard(xn):
        return ops.poswise_collapsed(xn.contiguous(), u.contiguous(), scale)


@poswise_weight.register_fake
def _(xn, u, scale):
    return xn.new_empty(xn.shape[0], u.shape[2], xn.shape[1], xn.shape[2], dtype=F32)


@torch.library.custom_op("rfmi::pair_bias_attention", mutates_args=(), device_types="cuda")
def pair_bias_attention(logits: Tensor, v: Tensor, msa: Tensor) -> Tensor:
    """MSA <- pair attention of MsaUpdateWithPairLayer (rf.py:588-595): logits fp32 [B, L, L, H] (= Linear(LN(sym(pair)))),
    v 16-bit [B, N, L, D] (= Linear(LN(msa))), msa fp32 [B, N, L, D] -> msa + softmax_j(logits)[b, h, i, :] . v[b, n, :, (h, :)]."""
    B, Lr, _, H = logits.shape
    _, N, _, D = v.shape
    dv = D // H
    with _guard(v):
        att = torch.empty(H, B, Lr, Lr, device=v.device, dtype=v.dtype)
        ops.softmax_batched(logits.contiguous(), 1, Lr * H, H, att, B * Lr * Lr, Lr, B * Lr, Lr, H)
        v_t = torch.empty(B, N, D, Lr, device=v.device, dtype=v.dtype)
        ops.copy4d(v.contiguous(), (N * Lr * D, Lr * D, D, 1), v_t, (N * D * Lr, D * Lr, 1, Lr), (B, N, Lr, D))
        out = msa.contiguous().clone()
        ops.gemm(att, v_t, out, Lr, N * dv, Lr, batch=(H, B, 1), a_bs=(B * Lr * Lr, Lr * Lr, 0), a_row=(0, 0, Lr),
                 b_bs=(dv * Lr, N * D * Lr, 0), b_row=(dv, D * Lr, Lr), c_bs=(dv, N * Lr * D, 0), c_row=(0, 0, D),
                 c_col=(dv, Lr * D), residual=out)
        return out


@pair_bias_attention.register_fake
def _(logits, v, msa):
    return msa.new_empty(msa.shape, dtype=F32)


@torch.library.custom_op("rfmi::masked_dist_attention", mutates_args=(), device_types="cuda")
def masked_dist_attention(q: Tensor, k: Tensor, xyz: Tensor, bins: Tensor) -> Tensor:
    """MsaUpdateWithPairAndCoord's attention map (rf.py:891-913): q (already scaled), k fp32 [B, L, H*dq], xyz fp32 [B, L, 3, 3],
    bins fp32 [H] (head h sees the pairs whose CA distance is below bins[h]) -> softmax_j map, 16-bit [B, H, L, L]."""
    B, Lr, _ = q.shape
    H = bins.numel()
    with _guard(q):
        att = torch.empty(B, H, Lr, Lr, device=q.device, dtype=ops.h16())
        ops.dist_masked_attention(q.contiguous(), k.contiguous(), xyz.contiguous(), bins.contiguous(), att, B, Lr, H, q.shape[-1] // H)
        return att


@masked_dist_attention.register_fake
def _(q, k, xyz, bins):
    return q.new_empty(q.shape[0], bins.numel(), q.shape[1], q.shape[1], dtype=ops.h16())


@torch.library.custom_op("rfmi::graph_transformer_dense", mutates_args=(), device_types="cuda")
def graph_transformer_dense(q: Tensor, k: Tensor, v: Tensor, e: Tensor, scale: float) -> Tensor:
    """GraphTransformer on the dense L x L graph (rf.py:644-661): q, k, v [B, L, H, d] and e [B, L, L, H*d] (the edge projection,
    added to the keys in the logits and to the values), all of one dtype (16-bit or fp32) -> fp32 [B, L, H*d]."""
    B, Lr, H, d = q.shape
    with _guard(q):
        out = torch.empty(B, Lr, H * d, device=q.device, dtype=F32)
        ops.graph_attention(q.contiguous(), k.contiguous(), v.contiguous(), e.contiguous(), out, B, Lr, H, d, scale)
        return out


@graph_transformer_dense.register_fake
def _(q, k, v, e, scale):
    return q.new_empty(q.shape[0], q.shape[1], q.shape[2] * q.shape[3], dtype=F32)


@torch.library.custom_op("rfmi::knn_graph_csc", mutates_args=(), device_types="cuda")
def knn_graph_csc(mask: Tensor, capacity: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Edge list of a dense adjacency mask in the reference's torch.where order (rf.py:853-856) with a static capacity:
    (src int32 [cap], dst int32 [cap], eid int32 [B, L, L] (-1: no edge), count int32 [2] = (edges kept, edges found))."""
    with _guard(mask):
        return ops.edges_from_mask(mask.contiguous(), capacity, zero_tail=True)  # (entries past count[0]: zero)


@knn_graph_csc.register_fake
def _(mask, capacity):
    i32 = torch.int32
    return (mask.new_empty(capacity, dtype=i32), mask.new_empty(capacity, dtype=i32), mask.new_empty(mask.shape, dtype=i32),
            mask.new_empty(2, dtype=i32))


@torch.library.custom_op("rfmi::se3_edge_kernel", mutates_args=(), device_types="cuda")
def se3_edge_kernel(feat: Tensor, net0: Optional[Tensor], net1: Optional[Tensor], basis: Tensor, h0: Optional[Tensor],
                    h1: Optional[Tensor], src: Tensor, count: Tensor, mo: int, dout: int, ln_eps: float) -> Tensor:
    """Fused per-edge kernel of the SE(3) structure module (csrc/se3.hip: rf_se3_radial_message; ea/modules.py:246-325, 612-641):
    radial MLP -> (x) basis -> matvec with the source node's features.  feat fp32 [E, d_edge+1], net_di: packed fp32 parameters
    of the (di, dout) radial net (include/rfmi.h), basis fp32 [E, 34], h0 [V, mi0], h1 [V, mi1, 3] -> msg fp32 [E, mo, 2 dout+1]
    (rows >= count[0] zero)."""
    with _guard(feat):
        cap = feat.shape[0]
        mi0 = h0.shape[1] if h0 is not None else 0
        mi1 = h1.shape[1] if h1 is not None else 0
        msg = ops.se3_radial_message(feat.contiguous(), feat.shape[1], net0, net1, basis.contiguous(),
                                     h0.contiguous() if h0 is not None else None, h1.contiguous() if h1 is not None else None,
                                     src, count, mo, dout, mi0, mi1, ln_eps, cap, zero_tail=True)
        return msg


@se3_edge_kernel.register_fake
def _(feat, net0, net1, basis, h0, h1, src, count, mo, dout, ln_eps):
    return feat.new_empty(feat.shape[0], mo, 2 * dout + 1, dtype=F32)


@torch.library.custom_op("rfmi::segment_softmax_sum", mutates_args=(), device_types="cuda")
def segment_softmax_sum(k0: Tensor, k1: Tensor, q0: Tensor, q1: Tensor, v0: Tensor, v1: Tensor, eid: Tensor,
                        heads: int) -> Tuple[Tensor, Tensor]:
    """GMABSE3 (ea/modules.py:738-774): e = <k_edge, q[dst]> / sqrt(n_features) per head, softmax over the incoming edges of each
    destination node, out[dst] = sum a v (zeros for in-degree 0).  k0 [E, mk0] k1 [E, mk1, 3] v0 [E, mv0] v1 [E, mv1, 3] per edge,
    q0 [V, mk0, 1] q1 [V, mk1, 3] per node, eid int32 [B, L, L] -> (out0 fp32 [V, mv0, 1], out1 fp32 [V, mv1, 3])."""
    B, Lr, _ = eid.shape
    with _guard(k0):
        return ops.se3_attention(k0.contiguous(), k1.contiguous(), q0.contiguous(), q1.contiguous(), v0.contiguous(), v1.contiguous(),
                                 eid.contiguous(), heads, k0.shape[1], k1.shape[1], v0.shape[1], v1.shape[1], B * Lr, Lr)


@segment_softmax_sum.register_fake
def _(k0, k1, q0, q1, v0, v1, eid, heads):
    V = eid.shape[0] * eid.shape[1]
    return k0.new_empty(V, v0.shape[1], 1, dtype=F32), k0.new_empty(V, v1.shape[1], 3, dtype=F32)


@torch.library.custom_op("rfmi::embed_msa", mutates_args=(), device_types="cuda")
def embed_msa(msa: Tensor, aa_idx: Tensor, emb: Tensor, pe: Tensor, qenc: Tensor) -> Tensor:
    """MsaEmbedding (rf.py:106-120): emb[msa] + pe[aa_idx] + qenc[row 0 ? 0 : 1] -> fp32 [B, N, L, D].  Indices must be in range
    (the module wrappers validate them first: rf_check_inputs)."""
    with _guard(msa):
        return ops.msa_embed(msa.contiguous(), aa_idx.contiguous(), emb.contiguous(), pe.contiguous(), qenc.contiguous())


@embed_msa.register_fake
def _(msa, aa_idx, emb, pe, qenc):
    return emb.new_empty(*msa.shape, emb.shape[1], dtype=F32)


@torch.library.custom_op("rfmi::embed_pair", mutates_args=(), device_types="cuda")
def embed_pair(seq: Tensor, aa_idx: Tensor, tl: Tensor, tr: Tensor, wsep: Tensor, bias: Tensor, pe: Tensor) -> Tensor:
    """PairEmbedding + 2-D positional encoding (rf.py:123-181, 79-103) with the 289 -> d Linear folded into two 21-row tables
    tl / tr, the separation column wsep and the bias -> fp32 [B, L, L, d]."""
    with _guard(seq):
        return ops.pair_embed(seq.contiguous(), aa_idx.contiguous(), tl.contiguous(), tr.contiguous(), wsep.contiguous(),
                              bias.contiguous(), pe.contiguous())


@embed_pair.register_fake
def _(seq, aa_idx, tl, tr, wsep, bias, pe):
    return tl.new_empty(seq.shape[0], seq.shape[1], seq.shape[1], tl.shape[1], dtype=F32)


OPS = ("linear", "layernorm", "tied_row_attention", "performer_attention", "outer_product_ln_linear", "conv3x3_nhwc",
       "instance_norm_elu", "knn_mask", "ffn", "poswise_weight", "pair_bias_attention", "masked_dist_attention",
       "graph_transformer_dense", "knn_graph_csc", "se3_edge_kernel", "segment_softmax_sum", "embed_msa", "embed_pair")
