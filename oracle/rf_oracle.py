"""CPU oracle for the RoseTTAFold forward path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the reference algorithm
(dohlee/rosettafold-pytorch).  It is the *checker* for the HIP path: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product (`rosettafold_pytorch_amd`) never imports it and fails
loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * every function that restates code living under /root/reference is pinned by
    golden vectors captured from the reference's own modules
    (`tools/make_goldens.py` -> `tests/golden/*.npz`, `tests/test_oracle_golden.py`);
  * three third-party dependencies of the reference are NOT under
    /root/reference and not installed anywhere in this pipeline:
      - performer-pytorch (unpinned, reference setup.py:24)  -> `performer_self_attention`
      - dgl (undeclared, unpinned)                           -> the edge gather / edge softmax /
                                                                 scatter-sum inside `se3_*`
      - lie_learn (unpinned, reference setup.py:25)          -> the Q_J change-of-basis constants
    Their published algorithms are restated here from the reference's call
    sites; for exactly those pieces **parity is unpinned**.

All functions are functional: parameters come in a flat dict `P` whose keys are
the reference's `state_dict()` names (plus explicit keys for the layers the
reference hides in plain Python lists, rf.py:602-605 and rf.py:699-702), and
`pre` is the key prefix of the sub-module being evaluated.  Inference
semantics: every dropout is the identity.

`rf.py` below = rosettafold_pytorch/rosettafold_pytorch.py, `ea/` =
rosettafold_pytorch/equivariant_attention/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

N_IDX, CA_IDX, C_IDX = 0, 1, 2  # rf.py:15


# --------------------------------------------------------------------------- helpers
def _lin(P, pre, x):
    return F.linear(x, P[pre + ".weight"], P.get(pre + ".bias"))


def _ln(P, pre, x, eps=1e-5):
    w = P[pre + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, P[pre + ".bias"], eps)


def sinusoid_table(dim, max_len):
    """rf.py:63-68 (same float32 op order as the reference)."""
    pe = torch.zeros(max_len, dim)
    denom = torch.exp(math.log(10000.0) * torch.arange(0, dim, 2) / dim)
    pos = torch.arange(0, max_len).view(-1, 1)
    pe[:, 0::2] = torch.sin(pos / denom)
    pe[:, 1::2] = torch.cos(pos / denom)
    return pe


# --------------------------------------------------------------------------- embeddings
def msa_embedding(P, pre, msa, aa_idx, max_len):
    """rf.py:106-120 + rf.py:57-76.  msa [B,N,L] int64 -> [B,N,L,d_msa]."""
    emb = P[pre + ".to_embedding.weight"][msa]
    d = emb.shape[-1]
    pe = sinusoid_table(d, max_len)[aa_idx]  # [B,L,d]
    x = emb + pe[:, None]
    qidx = torch.ones(msa.shape[1], dtype=torch.long)
    qidx[0] = 0
    return x + P[pre + ".query_enc.weight"][qidx][None, :, None, :]


def pair_embedding(P, pre, seq, aa_idx, max_len, template=None):
    """rf.py:123-181 + rf.py:79-103; template [B,L,L,d_template] selects the use_template branch (rf.py:161-169)."""
    L = seq.shape[-1]
    e = P[pre + ".embed_seq.weight"][seq]  # [B,L,d/2]
    left = e[:, None, :, :].expand(-1, L, -1, -1)  # left[b,i,j] = e[b,j]
    right = e[:, :, None, :].expand(-1, -1, L, -1)  # right[b,i,j] = e[b,i]
    dist = aa_idx.unsqueeze(-1) - aa_idx.unsqueeze(-2)
    sep = torch.log(torch.abs(dist) + 1).unsqueeze(-1)
    feats = [left, right, sep]
    if template is not None:
        feats.append(_ln(P, pre + ".ln_template", template))
    x = _lin(P, pre + ".proj", torch.cat(feats, -1))
    dh = e.shape[-1]
    pe = sinusoid_table(dh, max_len)[aa_idx]  # [B,L,dh]
    pe_row = pe[:, :, None, :].expand(-1, -1, L, -1)
    pe_col = pe[:, None, :, :].expand(-1, L, -1, -1)
    return x + torch.cat([pe_row, pe_col], -1)


# --------------------------------------------------------------------------- MSA row attention
def poswise_weight(P, pre, x, n_heads):
    """rf.py:184-217.  x [B,N,L,d] -> w [B,N,h,L,1]; softmax over N."""
    B, N, L, d = x.shape
    dh = d // n_heads
    q = _lin(P, pre + ".to_q.0", x[:, :1]) * dh ** -0.5  # [B,1,L,d]
    k = _lin(P, pre + ".to_k.0", x)
    q = q.view(B, 1, L, n_heads, dh).permute(0, 2, 3, 1, 4)  # b l h 1 d
    k = k.view(B, N, L, n_heads, dh).permute(0, 2, 3, 1, 4)  # b l h N d
    logits = torch.einsum("blhqd,blhnd->blhqn", q, k)
    att = logits.softmax(-1)  # b l h 1 N
    return att.permute(0, 4, 2, 1, 3)  # b N h l 1


def soft_tied_attention(P, pre, x, n_heads):
    """rf.py:220-267.  Returns (out [B,N,L,d], symmetrised att [B,L,L,h])."""
    B, N, L, d = x.shape
    dh = d // n_heads

    def heads(t):
        return t.view(B, N, L, n_heads, dh).permute(0, 1, 3, 2, 4)  # b n h l d

    q, k, v = (heads(_lin(P, pre + "." + nm, x)) for nm in ("to_q", "to_k", "to_v"))
    q = q * poswise_weight(P, pre + ".poswise_weight", x, n_heads) * dh ** -0.5
    logits = torch.einsum("bnhid,bnhjd->bhij", q, k)
    att = logits.softmax(-1)
    out = torch.einsum("bhij,bnhjd->bnhid", att, v)
    out = out.permute(0, 1, 3, 2, 4).reshape(B, N, L, d)
    out = _lin(P, pre + ".to_out", out)
    att_sym = ((att + att.transpose(-1, -2)) * 0.5).permute(0, 2, 3, 1)
    return out, att_sym


def feed_forward(P, pre, x):
    """rf.py:270-281."""
    return _lin(P, pre + ".net.3", F.relu(_lin(P, pre + ".net.0", x)))


# --------------------------------------------------------------------------- Performer (third party; parity unpinned)
def gaussian_orthogonal_random_matrix(nb_rows, nb_cols, generator):
    """performer-pytorch's projection construction (scaling=0): stacked QR
    Q-factors of Gaussian blocks, rows rescaled by norms of an independent
    Gaussian matrix.  Restated from the published library; unpinned."""
    blocks = []
    for _ in range(nb_rows // nb_cols):
        q, _ = torch.linalg.qr(torch.randn(nb_cols, nb_cols, generator=generator), mode="reduced")
        blocks.append(q.t())
    rem = nb_rows - (nb_rows // nb_cols) * nb_cols
    if rem > 0:
        q, _ = torch.linalg.qr(torch.randn(nb_cols, nb_cols, generator=generator), mode="reduced")
        blocks.append(q.t()[:rem])
    mat = torch.cat(blocks)
    mult = torch.randn(nb_rows, nb_cols, generator=generator).norm(dim=1)
    return torch.diag(mult) @ mat


def favor_softmax_features(data, proj, is_query, eps=1e-4):
    """FAVOR+ positive softmax-kernel features (Choromanski et al. 2020), as in
    performer-pytorch `softmax_kernel`.  data [...,n,d], proj [m,d]."""
    d = data.shape[-1]
    norm = d ** -0.25
    ratio = proj.shape[0] ** -0.5
    dash = (norm * data) @ proj.t()
    diag = (data ** 2).sum(-1, keepdim=True) * 0.5 * norm ** 2
    if is_query:
        mx = dash.amax(-1, keepdim=True)
    else:
        mx = dash.amax((-1, -2), keepdim=True)
    return ratio * (torch.exp(dash - diag - mx) + eps)


def favor_relu_features(data, proj, eps=1e-3):
    """performer-pytorch `generalized_kernel` with kernel_fn=ReLU."""
    d = data.shape[-1]
    return F.relu((d ** -0.25 * data) @ proj.t()) + eps


def linear_attention(q, k, v):
    """Non-causal linear attention D^-1 q'(k'^T v) (performer-pytorch `linear_attention`)."""
    ksum = k.sum(-2)
    dinv = 1.0 / torch.einsum("...nd,...d->...n", q, ksum)
    ctx = torch.einsum("...nd,...ne->...de", k, v)
    return torch.einsum("...de,...nd,...n->...ne", ctx, q, dinv)


def performer_self_attention(P, pre, x, heads, generalized, dim_head=64):
    """performer_pytorch.SelfAttention as instantiated at rf.py:313-318 (softmax
    kernel) and rf.py:505-518 (generalized ReLU kernel).  x [S, n, dim]."""
    S, n, _ = x.shape
    q = F.linear(x, P[pre + ".to_q.weight"], P.get(pre + ".to_q.bias"))
    k = F.linear(x, P[pre + ".to_k.weight"], P.get(pre + ".to_k.bias"))
    v = F.linear(x, P[pre + ".to_v.weight"], P.get(pre + ".to_v.bias"))

    def split(t):
        return t.view(S, n, heads, dim_head).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    proj = P[pre + ".fast_attention.projection_matrix"]
    if generalized:
        qp, kp = favor_relu_features(q, proj), favor_relu_features(k, proj)
    else:
        qp, kp = favor_softmax_features(q, proj, True), favor_softmax_features(k, proj, False)
    out = linear_attention(qp, kp, v)
    out = out.permute(0, 2, 1, 3).reshape(S, n, heads * dim_head)
    return _lin(P, pre + ".to_out", out)


# --------------------------------------------------------------------------- encoder layers
def encoder_layer_tied(P, pre, x, n_heads):
    """rf.py:284-354 with tied=True, return_att=True."""
    a, att = soft_tied_attention(P, pre + ".attn", _ln(P, pre + ".ln", x), n_heads)
    x = x + a
    x = x + feed_forward(P, pre + ".ff.fn.1", _ln(P, pre + ".ff.fn.0", x))
    return x, att


def encoder_layer_performer(P, pre, x, n_heads):
    """rf.py:284-354 with performer=True.  x [B,n1,n2,d]: attention along n2."""
    B, n1, n2, d = x.shape
    xf = x.reshape(B * n1, n2, d)
    a = performer_self_attention(P, pre + ".attn", _ln(P, pre + ".ln", xf), n_heads, False)
    xf = xf + a
    x = xf.view(B, n1, n2, d)
    return x + feed_forward(P, pre + ".ff.fn.1", _ln(P, pre + ".ff.fn.0", x))


def msa_update_using_self_attention(P, pre, x, n_layers, n_heads=12):
    """rf.py:357-409."""
    att = None
    for i in range(n_layers):
        x, att = encoder_layer_tied(P, f"{pre}.residue_wise_encoder_layers.{i}", x, n_heads)
    x = x.transpose(1, 2)  # b l n d
    for i in range(n_layers):
        x = encoder_layer_performer(P, f"{pre}.sequence_wise_encoder_layers.{i}", x, n_heads)
    return x.transpose(1, 2).contiguous(), att


# --------------------------------------------------------------------------- pair update with MSA
def outer_product_mean(P, pre, x, y):
    """rf.py:412-427 (sum over N, not a mean)."""
    o = torch.einsum("bniu,bnjv->bijuv", x, y)
    o = o.reshape(*o.shape[:3], -1)
    return _lin(P, pre + ".to_out.1", _ln(P, pre + ".to_out.0", o))


def instance_norm(P, pre, x, eps=1e-6):
    """nn.InstanceNorm2d(affine=True, eps=1e-6) on NCHW (rf.py:453, resnet.py:29)."""
    return F.instance_norm(x, weight=P[pre + ".weight"], bias=P[pre + ".bias"], eps=eps)


def pair_update_with_msa(P, pre, msa, pair, att):
    """rf.py:430-498."""
    L = msa.shape[2]
    mp = _ln(P, pre + ".proj_msa.2", _lin(P, pre + ".proj_msa.1", _ln(P, pre + ".proj_msa.0", msa)))
    w = poswise_weight(P, pre + ".poswise_weight", mp, 1)  # b n 1 l 1
    w = w[:, :, 0]  # b n l 1
    coevol = outer_product_mean(P, pre + ".outer_product_mean", mp, mp * w)
    coevol = _ln(P, pre + ".ln_coevol_feat", coevol)
    msa_1d = torch.cat([mp.sum(1), mp[:, 0]], -1)  # b l 2*d_proj
    row = msa_1d[:, :, None, :].expand(-1, -1, L, -1)
    col = msa_1d[:, None, :, :].expand(-1, L, -1, -1)
    feat = torch.cat([coevol, row, col, _ln(P, pre + ".ln_pair", pair), att], -1)
    x = _lin(P, pre + ".resnet.0", feat)
    y = x.permute(0, 3, 1, 2)
    y = F.conv2d(y, P[pre + ".resnet.1.fn.1.weight"], padding="same")
    y = F.elu(instance_norm(P, pre + ".resnet.1.fn.2", y))
    y = F.conv2d(y, P[pre + ".resnet.1.fn.5.weight"], padding="same")
    y = instance_norm(P, pre + ".resnet.1.fn.6", y).permute(0, 2, 3, 1)
    return F.elu(y + x)


# --------------------------------------------------------------------------- pair axial attention
def pair_axial_layer(P, pre, x, n_heads=8):
    """rf.py:501-528.  RowWise = sequences along dim 1 (i) for fixed j (rf.py:44-54),
    ColWise = along dim 2 (j) for fixed i (rf.py:31-41)."""
    B, L1, L2, d = x.shape
    xn = _ln(P, pre + ".layer.0.fn.0", x)
    xr = xn.permute(0, 2, 1, 3).reshape(B * L2, L1, d)  # (b l) n d
    a = performer_self_attention(P, pre + ".row_attn", xr, n_heads, True)
    x = x + a.view(B, L2, L1, d).permute(0, 2, 1, 3)
    xn = _ln(P, pre + ".layer.1.fn.0", x)
    a = performer_self_attention(P, pre + ".col_attn", xn.reshape(B * L1, L2, d), n_heads, True)
    x = x + a.view(B, L1, L2, d)
    return x + feed_forward(P, pre + ".ff", _ln(P, pre + ".layer.2.fn.0", x))


def pair_update_with_axial_attention(P, pre, x, n_layers):
    """rf.py:531-547."""
    for i in range(n_layers):
        x = pair_axial_layer(P, f"{pre}.layers.{i}", x)
    return x


# --------------------------------------------------------------------------- MSA update with pair
def msa_update_with_pair_layer(P, pre, msa, pair, n_heads):
    """rf.py:559-595."""
    B, N, L, d = msa.shape
    sym = 0.5 * (pair + pair.transpose(1, 2))
    a = _lin(P, pre + ".pair2att.2", _ln(P, pre + ".pair2att.1", sym))  # b i j h
    att = a.permute(0, 3, 1, 2).softmax(-1)  # b h i j
    v = _lin(P, pre + ".msa2value.1", _ln(P, pre + ".msa2value.0", msa))
    v = v.view(B, N, L, n_heads, d // n_heads).permute(0, 1, 3, 2, 4)  # b n h j d
    upd = torch.einsum("bhij,bnhjd->bnhid", att, v)
    upd = upd.permute(0, 1, 3, 2, 4).reshape(B, N, L, d)
    x = msa + upd
    return x + feed_forward(P, pre + ".ff.fn.1", _ln(P, pre + ".ff.fn.0", x))


def msa_update_with_pair(P, pre, msa, pair, n_layers, n_heads=4):
    """rf.py:598-610 (the reference keeps these layers in a plain list)."""
    for i in range(n_layers):
        msa = msa_update_with_pair_layer(P, f"{pre}.encoder_layers.{i}", msa, pair, n_heads)
    return msa


# --------------------------------------------------------------------------- initial coordinates
def graph_transformer(P, pre, node, edge, n_heads):
    """rf.py:613-664 with edge_mask=None."""
    B, L, _ = node.shape
    dout = P[pre + ".node_to_q.weight"].shape[0] // n_heads

    def heads(t):
        return t.view(B, L, n_heads, dout).permute(0, 2, 1, 3)

    q, k, v = (heads(_lin(P, pre + "." + nm, node)) for nm in ("node_to_q", "node_to_k", "node_to_v"))
    e = _lin(P, pre + ".edge_emb", edge).view(B, L, L, n_heads, dout).permute(0, 3, 1, 2, 4)
    logit = torch.einsum("bhid,bhjd->bhij", q, k) + torch.einsum("bhid,bhijd->bhij", q, e)
    att = (logit * dout ** -0.5).softmax(-1)
    upd = torch.einsum("bhij,bhjd->bhid", att, v) + torch.einsum("bhij,bhijd->bhid", att, e)
    upd = upd.permute(0, 2, 1, 3).reshape(B, L, n_heads * dout)
    return _lin(P, pre + ".node_update", node) + upd


def graph_transformer_block(P, pre, node, edge, n_heads):
    """rf.py:667-676."""
    h = _ln(P, pre + ".ln", graph_transformer(P, pre + ".attn", node, edge, n_heads))
    return F.elu(_lin(P, pre + ".to_out.0", h)) + node


def weighted_msa_node_input(P, pre, msa, seq_onehot):
    """Shared front of rf.py:715-724 and rf.py:789-798."""
    m = _ln(P, pre + ".ln_msa", msa)
    w = poswise_weight(P, pre + ".poswise_weight", m, 1)[:, :, 0]  # b n l 1
    return torch.cat([(m * w).sum(1), seq_onehot], -1)


def initial_coord_generation(P, pre, msa, pair, seq_onehot, aa_idx, n_layers=4, n_heads=4):
    """rf.py:679-749."""
    node = F.elu(_lin(P, pre + ".node_embed.0", weighted_msa_node_input(P, pre, msa, seq_onehot)))
    dist = aa_idx.unsqueeze(-1) - aa_idx.unsqueeze(-2)
    sep = (torch.sign(dist) * torch.log(torch.abs(dist) + 1)).clamp(0.0, 5.5).unsqueeze(-1)
    edge = F.elu(_lin(P, pre + ".edge_embed.0", torch.cat([_ln(P, pre + ".ln_pair", pair), sep], -1)))
    for i in range(n_layers):
        node = graph_transformer_block(P, f"{pre}.blocks.{i}", node, edge, n_heads)
    return _lin(P, pre + ".to_out", node).view(*node.shape[:2], 3, 3)


# --------------------------------------------------------------------------- SE(3) pieces
_SQ3, _SQ6 = math.sqrt(3.0), math.sqrt(6.0)


def real_sh(d):
    """Closed form of ea/from_se3cnn/utils_steerable.py:82-135,290-314 +
    representations.py:103-206 for J=0..2.  d [E,3] in input axis order;
    (y,z,x) = (u0,u1,u2); the zero vector maps to u=(0,1,0) (atan2(0,0)=0)."""
    r = d.norm(dim=-1, keepdim=True)
    u = torch.where(r > 0, d / r.clamp_min(1e-30), torch.tensor([0.0, 1.0, 0.0], dtype=d.dtype))
    y, z, x = u[:, 0], u[:, 1], u[:, 2]
    Y0 = torch.full_like(r, 0.28209479177387814)
    Y1 = -0.4886025119029199 * u
    c = 1.0925484305920792
    Y2 = torch.stack([c * x * y, c * y * z, 0.31539156525252005 * (3 * z * z - 1), c * x * z,
                      0.5462742152960396 * (x * x - y * y)], -1)
    return [Y0, Y1, Y2]


def q_j_constants():
    """Change-of-basis constants Q_J^{d_in,d_out} (ea/from_se3cnn/utils_steerable.py:36-78).
    The reference gets them from lie_learn's Wigner-D (absent here); these are the
    unit-Frobenius-norm solutions of the same intertwiner equation for the D_J implied by
    the reference's own real SH.  Sign convention (documented rule): the first non-zero
    entry in row-major order is positive.  dict[(d_in,d_out)] -> list over J of
    [(2do+1)*(2di+1), 2J+1] float64 tensors.  `tools/derive_qj.py` re-derives and checks them."""
    I3 = torch.eye(3, dtype=torch.float64)
    Q = {}
    Q[(0, 0)] = [torch.ones(1, 1, dtype=torch.float64)]
    Q[(0, 1)] = [I3 / _SQ3]  # [(3*1), 3]
    Q[(1, 0)] = [I3 / _SQ3]
    q110 = (I3 / _SQ3).reshape(9, 1)
    eps = torch.zeros(3, 3, 3, dtype=torch.float64)
    for a, b, c in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[a, b, c] = 1.0
        eps[b, a, c] = -1.0
    q111 = eps.reshape(9, 3) / _SQ6
    # J=2: symmetric-traceless projector in the Y2 basis (xy, yz, 3z^2-1, xz, x^2-y^2), with
    # vector components ordered (y,z,x) = indices (0,1,2).
    s10, s30 = math.sqrt(10.0), math.sqrt(30.0)  # unit Frobenius norm over the whole [9,5] matrix
    q112 = torch.zeros(3, 3, 5, dtype=torch.float64)
    Y, Z, X = 0, 1, 2
    q112[X, Y, 0] = q112[Y, X, 0] = -1 / s10
    q112[Y, Z, 1] = q112[Z, Y, 1] = -1 / s10
    q112[Z, Z, 2] = -2 / s30
    q112[X, X, 2] = q112[Y, Y, 2] = 1 / s30
    q112[X, Z, 3] = q112[Z, X, 3] = -1 / s10
    q112[X, X, 4] = -1 / s10
    q112[Y, Y, 4] = 1 / s10
    Q[(1, 1)] = [q110, q111, q112.reshape(9, 5)]
    return Q


_QJ = None


def se3_basis(d):
    """ea/modules.py:26-75: basis[(di,do)] [E, 2do+1, 2di+1, nJ]."""
    global _QJ
    if _QJ is None:
        _QJ = q_j_constants()
    Y = real_sh(d)
    basis = {}
    for di in (0, 1):
        for do in (0, 1):
            ks = []
            for n, J in enumerate(range(abs(di - do), di + do + 1)):
                Qj = _QJ[(di, do)][n].to(d.dtype)  # [(2do+1)(2di+1), 2J+1]
                ks.append(Y[J] @ Qj.t())
            basis[(di, do)] = torch.stack(ks, -1).view(-1, 2 * do + 1, 2 * di + 1, len(ks))
    return basis


def radial_func(P, pre, feat):
    """ea/modules.py:246-284 (BN = LayerNorm, ea/modules.py:545-558)."""
    h = F.relu(_ln(P, pre + ".net.1.bn", _lin(P, pre + ".net.0", feat)))
    h = F.relu(_ln(P, pre + ".net.4.bn", _lin(P, pre + ".net.3", h)))
    return _lin(P, pre + ".net.6", h)


def gconv_partial(P, pre, h, f_in, f_out, feat, basis, src):
    """ea/modules.py:561-680 (+ PairwiseConv :287-325): node -> edge messages.
    h[d] [V, m, 2d+1]; returns dict d_out -> [E, m_out, 2d_out+1]."""
    out = {}
    for do, mo in f_out.items():
        msg = 0
        for di, mi in f_in.items():
            nf = 2 * min(di, do) + 1
            R = radial_func(P, f"{pre}.kernel_unary.({di},{do}).rp", feat).view(-1, mo, 1, mi, 1, nf)
            kern = (R * basis[(di, do)][:, None, :, None, :, :]).sum(-1)  # E mo 2do+1 mi 2di+1
            kern = kern.reshape(kern.shape[0], mo * (2 * do + 1), mi * (2 * di + 1))
            s = h[di][src].reshape(-1, mi * (2 * di + 1), 1)
            msg = msg + torch.matmul(kern, s)
        out[do] = msg.view(msg.shape[0], mo, 2 * do + 1)
    return out


def g1x1(P, pre, h, degrees):
    """ea/modules.py:328-361."""
    return {d: torch.matmul(P[f"{pre}.transform.{d}"], h[d]) for d in degrees}


def _fiber2head(t, heads, degrees):
    """ea/fibers.py:163-170 with squeeze=True."""
    return torch.cat([t[d].reshape(t[d].shape[0], heads, -1) for d in degrees], -1)


def gmab(v, k, q, f_value, f_key, heads, src, dst, n_nodes):
    """ea/modules.py:683-774.  e = <k_edge, q_dst>/sqrt(n_features(f_key)); softmax over the
    incoming edges of each dst (DGL edge_softmax); sum of a*v into dst (update_all/fn.sum)."""
    kd = _fiber2head(k, heads, sorted(f_key))
    qd = _fiber2head(q, heads, sorted(f_key))
    nfeat = sum(m * (2 * d + 1) for d, m in f_key.items())
    e = (kd * qd[dst]).sum(-1) / math.sqrt(nfeat)  # [E, heads]
    mx = torch.full((n_nodes, heads), -float("inf"), dtype=e.dtype)
    mx = mx.scatter_reduce(0, dst[:, None].expand(-1, heads), e, "amax", include_self=True)
    ex = torch.exp(e - mx[dst])
    den = torch.zeros(n_nodes, heads, dtype=e.dtype).index_add_(0, dst, ex)
    a = ex / den[dst]
    out = {}
    for d, m in f_value.items():
        vv = v[d].view(-1, heads, m // heads, 2 * d + 1) * a[:, :, None, None]
        o = torch.zeros(n_nodes, heads, m // heads, 2 * d + 1, dtype=e.dtype).index_add_(0, dst, vv)
        out[d] = o.view(n_nodes, m, 2 * d + 1)
    return out


def gnorm_bias(P, pre, h):
    """ea/modules.py:364-406."""
    out = {}
    for d, v in h.items():
        norm = v.norm(2, -1, keepdim=True).clamp_min(1e-12)
        t = F.relu(norm[..., 0] + P[f"{pre}.bias.{d}"])
        out[d] = t.unsqueeze(-1) * (v / norm)
    return out


def gattentive_selfint(P, pre, h, f_out):
    """ea/modules.py:409-473."""
    out = {}
    for d, v in h.items():
        m_in, m_out = v.shape[-2], f_out[d]
        s = torch.einsum("nac,nbc->nab", v, v).reshape(-1, m_in * m_in)
        sign = s.sign()
        s = s.abs().clamp_min(1e-12) * sign
        t = F.leaky_relu(_ln(P, f"{pre}.transform.{d}.0", s))
        a = _lin(P, f"{pre}.transform.{d}.2", t).view(-1, m_out, m_in).softmax(-1)
        out[d] = torch.einsum("nom,nmd->nod", a, v)
    return out


def gse3res(P, pre, h, f_in, f_out, div, heads, selfint, feat, basis, src, dst, n_nodes):
    """ea/modules.py:777-857 with skip='cat'."""
    f_mid_out = {d: int(m // div) for d, m in f_out.items()}
    f_mid_in = {d: m for d, m in f_mid_out.items() if d in f_in}
    v = gconv_partial(P, pre + ".GMAB.v", h, f_in, f_mid_out, feat, basis, src)
    k = gconv_partial(P, pre + ".GMAB.k", h, f_in, f_mid_in, feat, basis, src)
    q = g1x1(P, pre + ".GMAB.q", h, sorted(f_mid_in))
    z = gmab(v, k, q, f_mid_out, f_mid_in, heads, src, dst, n_nodes)
    z = {d: (torch.cat([z[d], h[d]], 1) if d in h else z[d]) for d in z}  # GCat ea/modules.py:903-928
    if selfint == "att":
        return gattentive_selfint(P, pre + ".project", z, f_out)
    return g1x1(P, pre + ".project", z, sorted(f_out))


def se3_transformer(P, pre, type0, type1, src, dst, d, w, d_state, num_channels=16, n_heads=4):
    """rosettafold_pytorch/se3_modules.py:83-171 as instantiated at rf.py:774-784
    (num_layers=2, num_degrees=2, div=4, si_m='1x1', si_e='att')."""
    V = type0.shape[0]
    basis = se3_basis(d)
    r = d.norm(dim=-1, keepdim=True)
    feat = torch.cat([w, r], -1)
    h = {0: type0, 1: type1}
    f_in = {0: type0.shape[1], 1: type1.shape[1]}
    f_mid = {0: num_channels, 1: num_channels}
    f_out = {0: d_state, 1: 3}
    h = gse3res(P, pre + ".Gblock.0", h, f_in, f_mid, 4, n_heads, "1x1", feat, basis, src, dst, V)
    h = gnorm_bias(P, pre + ".Gblock.1", h)
    h = gse3res(P, pre + ".Gblock.2", h, f_mid, f_mid, 4, n_heads, "1x1", feat, basis, src, dst, V)
    h = gnorm_bias(P, pre + ".Gblock.3", h)
    h = gse3res(P, pre + ".Gblock.4", h, f_mid, f_out, 1, 1, "att", feat, basis, src, dst, V)
    return h


def knn_graph(xyz, idx, n_neighbors, kmin=9):
    """rf.py:823-862: edge i->j iff j in topk_smallest(pdist[i]) or |idx_i-idx_j| < kmin
    (self loops excluded by the +1e3 / +999.9 diagonals unless k >= L).  Returns (b, i, j)
    in row-major order."""
    B, L = xyz.shape[:2]
    ca = xyz[:, :, CA_IDX]
    pdist = torch.cdist(ca, ca) + torch.eye(L).unsqueeze(0) * 1e3
    sep = (idx[:, None, :] - idx[:, :, None]).abs() + torch.eye(L).unsqueeze(0) * 999.9
    k = min(n_neighbors, L)
    _, nb = torch.topk(pdist, k, largest=False)
    adj = torch.zeros(B, L, L).scatter(2, nb, 1.0)
    cond = torch.logical_or(adj > 0.0, sep < kmin)
    return torch.where(cond)


def coord_update(P, pre, xyz, msa, pair, aa_idx, seq_onehot, n_neighbors, d_state):
    """rf.py:752-821."""
    B, L = xyz.shape[:2]
    node = _ln(P, pre + ".node_embed.2",
               F.elu(_lin(P, pre + ".node_embed.0", weighted_msa_node_input(P, pre, msa, seq_onehot))))
    edge = _ln(P, pre + ".edge_embed.2", F.elu(_lin(P, pre + ".edge_embed.0", _ln(P, pre + ".ln_pair", pair))))
    b, i, j = knn_graph(xyz, aa_idx, n_neighbors)
    src, dst = b * L + i, b * L + j
    d = xyz[b, j, CA_IDX] - xyz[b, i, CA_IDX]
    w = edge[b, i, j]
    type0 = node.reshape(B * L, -1, 1)
    type1 = (xyz - xyz[:, :, CA_IDX].unsqueeze(-2)).reshape(B * L, 3, 3)
    out = se3_transformer(P, pre + ".se3_transformer", type0, type1, src, dst, d, w, d_state)
    state = out[0].view(B, L, -1)
    disp = out[1].view(B, L, 3, 3)
    ca = xyz[:, :, CA_IDX] + disp[:, :, CA_IDX]
    return state, torch.stack([ca + disp[:, :, N_IDX], ca, ca + disp[:, :, C_IDX]], 2)


# --------------------------------------------------------------------------- MSA update with coords
def msa_update_with_pair_and_coord(P, pre, xyz, state, msa, bins=(8, 12, 16, 20), d_inner=32):
    """rf.py:865-920."""
    B, N, L, d = msa.shape
    h = len(bins)
    st = _ln(P, pre + ".ln_state", state)
    m = _ln(P, pre + ".ln_msa", msa)
    scale = (state.shape[-1] // h) ** -0.5  # rf.py:874
    q = _lin(P, pre + ".to_q", st).view(B, L, h, d_inner).permute(0, 2, 1, 3) * scale
    k = _lin(P, pre + ".to_k", st).view(B, L, h, d_inner).permute(0, 2, 1, 3)
    v = _lin(P, pre + ".to_v", m).view(B, N, L, h, d // h).permute(0, 3, 1, 2, 4)  # b h n l d
    ca = xyz[:, :, CA_IDX]
    pd = torch.cdist(ca, ca)
    mask = torch.stack([(pd < t).float() for t in bins], 1)
    logits = torch.einsum("bhid,bhjd->bhij", q, k) + (1.0 - mask) * -1e9
    att = logits.softmax(-1)
    out = torch.einsum("bhij,bhnjd->bhnid", att, v).permute(0, 2, 3, 1, 4).reshape(B, N, L, d)
    x = m + _ln(P, pre + ".ln_out", out)
    return x + feed_forward(P, pre + ".to_out.fn.1", _ln(P, pre + ".to_out.fn.0", x))


# --------------------------------------------------------------------------- prediction head
def resblock2d(P, pre, x, dilation):
    """resnet.py:15-44."""
    y = F.conv2d(x, P[pre + ".layer.fn.0.weight"], padding="same", dilation=dilation)
    y = F.elu(instance_norm(P, pre + ".layer.fn.1", y))
    y = F.conv2d(y, P[pre + ".layer.fn.4.weight"], padding="same", dilation=dilation)
    return F.elu(instance_norm(P, pre + ".layer.fn.5", y) + x)


def resnet(P, pre, x, n_blocks, dilations=(1, 2, 4, 8)):
    """resnet.py:47-83."""
    x = F.elu(instance_norm(P, pre + ".layer.1", F.conv2d(x, P[pre + ".layer.0.weight"])))
    for b in range(n_blocks):
        x = resblock2d(P, f"{pre}.layer.{3 + b}", x, dilations[b % len(dilations)])
    k = 3 + n_blocks
    return F.conv2d(x, P[f"{pre}.layer.{k}.weight"], P[f"{pre}.layer.{k}.bias"])


def prediction_head(P, pre, pair, n_blocks=4):
    """rf.py:1130-1172."""
    x = _lin(P, pre + ".proj.1", _ln(P, pre + ".proj.0", pair)).permute(0, 3, 1, 2)
    xs = (x + x.transpose(-1, -2)) * 0.5
    out = {}
    for name, inp in (("theta", x), ("phi", x), ("dist", xs), ("omega", xs)):
        out[name] = resnet(P, f"{pre}.{name}_head.0", inp, n_blocks).permute(0, 2, 3, 1)
    return out


# --------------------------------------------------------------------------- blocks and model
def two_track_block(P, pre, msa, pair, n_enc):
    """rf.py:923-968."""
    msa, att = msa_update_using_self_attention(P, pre + ".msa_update_using_self_att", msa, n_enc)
    pair = pair_update_with_msa(P, pre + ".pair_update_with_msa", msa, pair, att)
    pair = pair_update_with_axial_attention(P, pre + ".pair_update_with_axial_attention", pair, n_enc)
    msa = msa_update_with_pair(P, pre + ".msa_update_with_pair", msa, pair, n_enc)
    return msa, pair


def three_track_block(P, pre, msa, pair, xyz, seq_onehot, aa_idx, n_enc, n_neighbors, d_state, final=False):
    """rf.py:971-1046 (ThreeTrackBlock) and rf.py:1049-1127 (FinalBlock)."""
    msa, pair = two_track_block(P, pre, msa, pair, n_enc)
    state, xyz = coord_update(P, pre + ".coord_update_with_msa_and_pair", xyz, msa, pair, aa_idx, seq_onehot,
                              n_neighbors, d_state)
    if final:
        plddt = _lin(P, pre + ".plddt_head", state)[..., 0]
        return msa, pair, xyz, plddt
    msa = msa_update_with_pair_and_coord(P, pre + ".msa_update_with_pair_and_coord", xyz, state, msa)
    return msa, pair, xyz


def rosettafold_forward(P, msa, seq, aa_idx, cfg):
    """rf.py:1175-1289.  cfg: dict with d_state, n_two_track_blocks, n_three_track_blocks,
    n_encoder_layers, max_len, n_neighbors."""
    n_enc = cfg["n_encoder_layers"]
    m = msa_embedding(P, "msa_emb", msa, aa_idx, cfg["max_len"])
    p = pair_embedding(P, "pair_emb", seq, aa_idx, cfg["max_len"])
    onehot = F.one_hot(seq, 21).float()
    for i in range(cfg["n_two_track_blocks"]):
        m, p = two_track_block(P, f"two_track_blocks.{i}", m, p, n_enc)
    xyz = initial_coord_generation(P, "initial_coord_generation_with_msa_and_pair", m, p, onehot, aa_idx)
    for i in range(cfg["n_three_track_blocks"] - 1):
        m, p, xyz = three_track_block(P, f"three_track_blocks.{i}", m, p, xyz, onehot, aa_idx, n_enc,
                                      cfg["n_neighbors"][i], cfg["d_state"])
    m, p, xyz, plddt = three_track_block(P, "final_block", m, p, xyz, onehot, aa_idx, n_enc, 32,
                                         cfg["d_state"], final=True)
    return prediction_head(P, "prediction_head", p), xyz, plddt
