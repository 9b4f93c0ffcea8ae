"""Structure track (SE(3)-Transformer over the kNN residue graph), the three-track blocks and the
top-level RoseTTAFold module.  Mirrors rf.py:752-1289, se3_modules.py:83-171 and the used classes of
equivariant_attention/modules.py (parameter names included) over the fp32 kernels of csrc/se3.hip."""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import F32
from .model import (RFModule, Residual, FeedForward, PositionWiseWeightFactor, MsaEmbedding, PairEmbedding,
                    MsaUpdateUsingSelfAttention, PairUpdateWithMsa, PairUpdateWithAxialAttention, MsaUpdateWithPair,
                    InitialCoordGenerationWithMsaAndPair, PredictionHead, LayerNorm, Linear, _node_input, _f, ln, T, pad8,
                    CA_IDX, fresh_f32, check_index_range, RT)


# ================================================================================================
# SE(3)-Transformer parameter containers (names as in ea/modules.py) + device forward
# ================================================================================================
class BN(nn.Module):
    """ea/modules.py:545-558 (a LayerNorm)."""

    def __init__(self, m):
        super().__init__()
        self.bn = LayerNorm(m)


class RadialFunc(nn.Module):
    """ea/modules.py:246-284."""

    def __init__(self, num_freq, in_dim, out_dim, edge_dim=0):
        super().__init__()
        self.num_freq, self.in_dim, self.out_dim, self.edge_dim, self.mid_dim = num_freq, in_dim, out_dim, edge_dim, 32
        self.net = nn.Sequential(Linear(edge_dim + 1, 32), BN(32), nn.ReLU(), Linear(32, 32), BN(32), nn.ReLU(),
                                 Linear(32, num_freq * in_dim * out_dim))
        nn.init.kaiming_uniform_(self.net[0].weight)
        nn.init.kaiming_uniform_(self.net[3].weight)
        nn.init.kaiming_uniform_(self.net[6].weight)


class PairwiseConv(nn.Module):
    """ea/modules.py:287-325."""

    def __init__(self, degree_in, nc_in, degree_out, nc_out, edge_dim=0):
        super().__init__()
        self.degree_in, self.degree_out, self.nc_in, self.nc_out = degree_in, degree_out, nc_in, nc_out
        self.num_freq = 2 * min(degree_in, degree_out) + 1
        self.rp = RadialFunc(self.num_freq, nc_in, nc_out, edge_dim)


class GConvSE3Partial(nn.Module):
    """ea/modules.py:561-680 (x_ij=None)."""

    def __init__(self, f_in, f_out, edge_dim=0):
        super().__init__()
        self.f_in, self.f_out = dict(f_in), dict(f_out)
        self.kernel_unary = nn.ModuleDict()
        for di, mi in f_in.items():
            for do, mo in f_out.items():
                self.kernel_unary[f"({di},{do})"] = PairwiseConv(di, mi, do, mo, edge_dim=edge_dim)


class G1x1SE3(nn.Module):
    """ea/modules.py:328-361."""

    def __init__(self, f_in, f_out):
        super().__init__()
        self.f_in, self.f_out = dict(f_in), dict(f_out)
        self.transform = nn.ParameterDict()
        for do, mo in f_out.items():
            mi = f_in[do]
            self.transform[str(do)] = nn.Parameter(torch.randn(mo, mi) / np.sqrt(mi))

    def run(self, h):
        out = {}
        for d in self.f_out:
            x = h[d]
            V, mi, nc = x.shape
            W = self.transform[str(d)].detach()
            mo = W.shape[0]
            y = torch.empty(V, mo, nc, device=x.device, dtype=F32)
            ops.gemm(x, W, y, V * nc, mo, mi, kc=1, a_row=(nc, mi * nc, 1), a_ko=nc, b_row=(0, 0, mi), b_ko=1,
                     c_row=(nc, mo * nc, 1), c_col=(1, nc))
            out[d] = y
        return out


class GNormBias(nn.Module):
    """ea/modules.py:364-406."""

    def __init__(self, fiber):
        super().__init__()
        self.bias = nn.ParameterDict({str(d): nn.Parameter(torch.randn(m).view(1, m)) for d, m in fiber.items()})

    def run(self, h):
        return {d: ops.se3_norm_bias(v, self.bias[str(d)].detach().reshape(-1), d) for d, v in h.items()}


class GAttentiveSelfInt(nn.Module):
    """ea/modules.py:409-473."""

    def __init__(self, f_in, f_out):
        super().__init__()
        self.f_in, self.f_out = dict(f_in), dict(f_out)
        self.transform = nn.ModuleDict()
        for d, mi in f_in.items():
            mo = f_out[d]
            lin = Linear(mi * mi, mi * mo, bias=True)
            nn.init.kaiming_uniform_(lin.weight)
            self.transform[str(d)] = nn.Sequential(LayerNorm(mi * mi), nn.LeakyReLU(), lin)

    def run(self, h):
        out = {}
        for d, v in h.items():
            net = self.transform[str(d)]
            mi, mo = self.f_in[d], self.f_out[d]
            s = ops.se3_gram(v, d)
            t = ops.layernorm(s, _f(net[0].weight), _f(net[0].bias), eps=net[0].eps, out_dtype=F32, act=L.ACT_LEAKY)
            a = ops.linear(t, net[2].weight.detach(), _f(net[2].bias), out_dtype=F32)
            out[d] = ops.se3_attn_apply(a, v, mo, d)
        return out


class GMABSE3(nn.Module):
    """ea/modules.py:683-774 (no parameters)."""

    def __init__(self, f_value, f_key, n_heads):
        super().__init__()
        self.f_value, self.f_key, self.n_heads = dict(f_value), dict(f_key), n_heads


def _pack_radial_net(rp):
    """[W1^T: ki x 32][b1][ln1 gamma][ln1 beta][W2^T: 32 x 32][b2][ln2 gamma][ln2 beta][W3: rows x 32][b3: rows] fp32."""
    n = rp.net
    parts = [n[0].weight.t(), n[0].bias, n[1].bn.weight, n[1].bn.bias, n[3].weight.t(), n[3].bias, n[4].bn.weight, n[4].bn.bias,
             n[6].weight, n[6].bias]
    return torch.cat([t.detach().float().contiguous().reshape(-1) for t in parts]).contiguous()


class GSE3Res(nn.Module):
    """ea/modules.py:777-857 with skip='cat'."""

    def __init__(self, f_in, f_out, edge_dim=0, div=4, n_heads=1, selfint="1x1"):
        super().__init__()
        self.f_in, self.f_out, self.n_heads, self.edge_dim = dict(f_in), dict(f_out), n_heads, edge_dim
        self.f_mid_out = {d: int(m // div) for d, m in f_out.items()}
        self.f_mid_in = {d: m for d, m in self.f_mid_out.items() if d in f_in}
        self.GMAB = nn.ModuleDict()
        self.GMAB["v"] = GConvSE3Partial(f_in, self.f_mid_out, edge_dim=edge_dim)
        self.GMAB["k"] = GConvSE3Partial(f_in, self.f_mid_in, edge_dim=edge_dim)
        self.GMAB["q"] = G1x1SE3(f_in, self.f_mid_in)
        self.GMAB["attn"] = GMABSE3(self.f_mid_out, self.f_mid_in, n_heads=n_heads)
        f_cat = {d: m + f_in.get(d, 0) for d, m in self.f_mid_out.items()}  # GCat ea/modules.py:903-928
        self.f_cat = f_cat
        self.project = GAttentiveSelfInt(f_cat, f_out) if selfint == "att" else G1x1SE3(f_cat, f_out)
        object.__setattr__(self, "_rfc", None)

    def _nets(self):
        """[(which, di, do, PairwiseConv)] in a fixed order; radial first/second layers are batched over them."""
        nets = []
        for which in ("v", "k"):
            conv = self.GMAB[which]
            for di in conv.f_in:
                for do in conv.f_out:
                    nets.append((which, di, do, conv.kernel_unary[f"({di},{do})"]))
        return nets

    def _packed(self):
        if not self._rfc:  # (None, or emptied by invalidate_weight_caches)
            nets = self._nets()
            with torch.no_grad():
                pk = {
                    "w1": torch.cat([n[3].rp.net[0].weight for n in nets], 0).float().contiguous(),
                    "b1": torch.cat([n[3].rp.net[0].bias for n in nets], 0).float().contiguous(),
                    "g1": torch.cat([n[3].rp.net[1].bn.weight for n in nets], 0).float().contiguous(),
                    "e1": torch.cat([n[3].rp.net[1].bn.bias for n in nets], 0).float().contiguous(),
                    "g2": torch.cat([n[3].rp.net[4].bn.weight for n in nets], 0).float().contiguous(),
                    "e2": torch.cat([n[3].rp.net[4].bn.bias for n in nets], 0).float().contiguous(),
                    # one flat fp32 buffer per net in the order the fused kernel reads it (include/rfmi.h: rf_se3_radial_message)
                    "nets": {(which, di, do): _pack_radial_net(pc.rp) for which, di, do, pc in nets},
                }
            object.__setattr__(self, "_rfc", pk)
        return self._rfc

    def _apply(self, fn, *a, **k):
        object.__setattr__(self, "_rfc", None)
        RT.cache_epoch += 1
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        object.__setattr__(self, "_rfc", None)  # packed radial weights are rebuilt from the loaded parameters
        RT.cache_epoch += 1
        return super()._load_from_state_dict(*a, **k)

    def run(self, h, g):
        """h: {0: [V,m0,1], 1: [V,m1,3]} fp32;  g: graph dict (src, dst, eid, count, basis, feat, cap, V, L)."""
        nets = self._nets()
        G = len(nets)
        pk = self._packed()
        cap, feat = g["cap"], g["feat"]
        eps = nets[0][3].rp.net[1].bn.eps
        h0 = h.get(0)
        h1f = h.get(1)
        mi0 = h0.shape[1] if h0 is not None else 0
        mi1 = h1f.shape[1] if h1f is not None else 0
        ki = feat.shape[1]
        slot = {(which, di, do): gi for gi, (which, di, do, _) in enumerate(nets)}
        last = {(which, di, do): pc.rp.net[6] for which, di, do, pc in nets}
        todo = [(which, do, mo) for which, fo in (("v", self.f_mid_out), ("k", self.f_mid_in)) for do, mo in fo.items()]
        fused = {(which, do): RT.se3_fused_radial and ops.se3_radial_message_supported(
            mo, do, mi0 if (which, 0, do) in last else 0, mi1 if (which, 1, do) in last else 0, ki) for which, do, mo in todo}
        h2 = None
        if not all(fused.values()):
            # radial MLPs as launches (shapes without a fused instance, or RF_SE3_UNFUSED=1): layer 1 for all G nets in one GEMM,
            # grouped LayerNorm + ReLU, per-net 32x32 layer, grouped LayerNorm + ReLU; the output layer follows per net below
            h1 = ops.linear(feat, pk["w1"], pk["b1"], out_dtype=F32)  # [cap, G*32]
            h1 = ops.layernorm(h1, pk["g1"], pk["e1"], eps=eps, out_dtype=F32, rows=cap * G, D=32, groups=G, act=L.ACT_RELU)
            h2 = torch.empty_like(h1)
            for gi, n in enumerate(nets):
                lin = n[3].rp.net[3]
                ops.gemm(h1, lin.weight.detach(), h2, cap, 32, 32, a_off=gi * 32, a_row=(0, 0, G * 32), c_off=gi * 32,
                         c_row=(0, 0, G * 32), bias=_f(lin.bias))
            h2 = ops.layernorm(h2, pk["g2"], pk["e2"], eps=eps, out_dtype=F32, rows=cap * G, D=32, groups=G, act=L.ACT_RELU)
        msg = {}
        for which, do, mo in todo:
            has0, has1 = (which, 0, do) in last, (which, 1, do) in last
            if fused[(which, do)]:
                # the whole radial MLP inside the message kernel: no hidden vectors, no [cap, mo*mi*nf] radial tensors in memory
                msg[(which, do)] = ops.se3_radial_message(
                    feat, ki, pk["nets"].get((which, 0, do)), pk["nets"].get((which, 1, do)), g["basis"],
                    h0 if has0 else None, h1f if has1 else None, g["src"], g["count"], mo, do,
                    mi0 if has0 else 0, mi1 if has1 else 0, eps, cap)
                continue
            R = {}
            for di in (0, 1):
                lin = last.get((which, di, do))
                if lin is None:
                    continue
                nout = lin.weight.shape[0]
                r = torch.empty(cap, nout, device=feat.device, dtype=F32)
                ops.gemm(h2, lin.weight.detach(), r, cap, nout, 32, a_off=slot[(which, di, do)] * 32, a_row=(0, 0, G * 32),
                         bias=_f(lin.bias))
                R[di] = r
            msg[(which, do)] = ops.se3_message(R.get(0), R.get(1), g["basis"], h0, h1f, g["src"], g["count"], mo, do, mi0, mi1, cap)
        q = self.GMAB["q"].run(h)
        fk, fv = self.f_mid_in, self.f_mid_out
        # skip connection 'cat' (GCat, ea/modules.py:903-928): the attention writes the leading channels of the
        # concatenated buffers, the node's input features are copied in behind them
        z0, z1 = ops.se3_attention(msg[("k", 0)], msg[("k", 1)], q[0], q[1], msg[("v", 0)], msg[("v", 1)], g["eid"],
                                   self.n_heads, fk[0], fk[1], fv[0], fv[1], g["V"], g["L"], skip0=h.get(0), skip1=h.get(1))
        return self.project.run({0: z0, 1: z1})


class SE3Transformer(nn.Module):
    """se3_modules.py:83-171 as instantiated at rf.py:774-784."""

    def __init__(self, num_layers=2, num_channels=32, num_degrees=3, n_heads=4, div=4, si_m="1x1", si_e="att",
                 l0_in_features=32, l0_out_features=32, l1_in_features=3, l1_out_features=3, num_edge_features=32,
                 x_ij=None):
        super().__init__()
        if num_degrees != 2 or x_ij is not None or l1_out_features <= 0:
            raise NotImplementedError("only the configuration used by the forward path (rf.py:774-784) is built")
        f_in = {0: l0_in_features, 1: l1_in_features}
        f_mid = {d: num_channels for d in range(num_degrees)}
        f_out = {0: l0_out_features, 1: l1_out_features}
        blocks = []
        fin = f_in
        for _ in range(num_layers):
            blocks.append(GSE3Res(fin, f_mid, edge_dim=num_edge_features, div=div, n_heads=n_heads, selfint=si_m))
            blocks.append(GNormBias(f_mid))
            fin = f_mid
        blocks.append(GSE3Res(f_mid, f_out, edge_dim=num_edge_features, div=1, n_heads=min(1, 2), selfint=si_e))
        self.Gblock = nn.ModuleList(blocks)

    def run(self, g, type0, type1):
        h = {0: type0, 1: type1}
        for blk in self.Gblock:
            h = blk.run(h, g) if isinstance(blk, GSE3Res) else blk.run(h)
        return h


_PENDING_EDGE_COUNTS = []   # (count tensor [min(edges, cap), edges], capacity) of the graphs built since the last check


def check_edge_capacity(pending=None):
    """Raises if a kNN graph had more edges than its static capacity (rf_edges_from_mask drops the overflow and reports the
    true count in count[1]): the capacity bound k + 2*(kmin-1) per row holds for strictly increasing residue indices,
    which the public forward()s verify -- this is the backstop behind that argument.  One 8-byte read-back per graph,
    taken at the END of a public forward (the launches are already queued; no bubble in front of them).
    `pending`: explicit list (graph.GraphedForward keeps the captured graph's counts); default: everything recorded since
    the last check, which is cleared."""
    items = pending if pending is not None else list(_PENDING_EDGE_COUNTS)
    if pending is None:
        _PENDING_EDGE_COUNTS.clear()
    for count, cap in items:
        kept, total = count.tolist()
        if total > cap:
            raise L.RfmiError(f"kNN graph overflow: {total} edges for a capacity of {cap} (residue indices not strictly "
                              f"increasing?) -- {total - kept} edges were dropped, the result is not the reference's")


def build_graph(xyz, edge_emb, aa_idx, n_neighbors, kmin=9, monotonic=True):
    """rf.py:823-862 on the device: dense mask -> compacted edge list (+ dense edge-id map) -> per-edge geometry.
    The edge count stays on the device; every per-edge buffer has a static capacity: B*L*min(L, k+2*(kmin-1)) when
    aa_idx is strictly increasing inside each sample (at most 2*(kmin-1) sequence neighbours besides the k nearest),
    B*L*L otherwise.  The compaction kernel never writes past the capacity (rf_edges_from_mask)."""
    B, Lr = xyz.shape[:2]
    k = min(n_neighbors, Lr)
    per_row = min(Lr, k + 2 * (kmin - 1)) if monotonic else Lr
    cap = (B * Lr * per_row + 63) // 64 * 64
    mask = ops.knn_mask(xyz, aa_idx, k, kmin)
    src, dst, eid, count = ops.edges_from_mask(mask, cap)
    _PENDING_EDGE_COUNTS.append((count, cap))
    del _PENDING_EDGE_COUNTS[:-64]   # callers of the internal run() paths never check: keep the list bounded
    basis, feat = ops.se3_edge_geometry(xyz, edge_emb, src, dst, count, cap)
    return {"src": src, "dst": dst, "eid": eid, "count": count, "basis": basis, "feat": feat, "cap": cap,
            "V": B * Lr, "L": Lr, "mask": mask}


class CoordUpdateWithMsaAndPair(RFModule):
    """rf.py:752-862."""

    def __init__(self, d_msa, d_pair, d_node, d_edge, d_state, n_neighbors, p_dropout=0.1):
        super().__init__()
        self.n_neighbors = n_neighbors
        self.ln_msa = LayerNorm(d_msa)
        self.ln_pair = LayerNorm(d_pair)
        self.poswise_weight = PositionWiseWeightFactor(d_msa, 1, p_dropout)
        self.node_embed = nn.Sequential(Linear(d_msa + 21, d_node), nn.ELU(), LayerNorm(d_node))
        self.edge_embed = nn.Sequential(Linear(d_pair, d_edge), nn.ELU(), LayerNorm(d_edge))
        self.se3_transformer = SE3Transformer(num_layers=2, num_channels=16, n_heads=4, num_degrees=2,
                                              l0_in_features=d_node, l1_in_features=3, l0_out_features=d_state,
                                              l1_out_features=3, num_edge_features=d_edge)

    def run(self, xyz, msa, pair, aa_idx, seq_onehot, monotonic=True):
        B, Lr = xyz.shape[:2]
        # the structure track is fp32 end to end (se3_modules.py:164): its input projections use the exact fp32 GEMM
        nin, Kp = _node_input(self, msa, seq_onehot, out_dtype=F32)
        wn = self.cached("wn32", lambda: torch.cat([self.node_embed[0].weight.detach().float(),
                                                    torch.zeros(self.node_embed[0].weight.shape[0],
                                                                Kp - self.node_embed[0].weight.shape[1],
                                                                device=msa.device)], 1).contiguous())
        node = ops.linear(nin, wn, _f(self.node_embed[0].bias), out_dtype=F32, act=L.ACT_ELU)
        node = ln(self.node_embed[2], node, out_dtype=F32)
        e = ops.linear(ln(self.ln_pair, pair, out_dtype=F32), self.edge_embed[0].weight.detach().float(),
                       _f(self.edge_embed[0].bias), out_dtype=F32, act=L.ACT_ELU)
        edge = ln(self.edge_embed[2], e, out_dtype=F32)  # [B,L,L,d_edge] fp32
        xyz = xyz.contiguous()
        g = build_graph(xyz, edge, aa_idx.contiguous(), self.n_neighbors, monotonic=monotonic)
        type0 = node.view(B * Lr, -1, 1)
        type1 = ops.center_ca(xyz).view(B * Lr, 3, 3)
        out = self.se3_transformer.run(g, type0, type1)
        state = out[0].view(B, Lr, -1)
        return state, ops.coord_apply(xyz, out[1].contiguous())

    def forward(self, xyz, msa, pair, aa_idx, seq_onehot):
        mono = check_index_range(None, None, aa_idx.contiguous(), 1, 2 ** 31 - 1)
        out = self.run(xyz.float(), msa.float().contiguous(), pair.float().contiguous(), aa_idx, seq_onehot.float(),
                       monotonic=mono)
        check_edge_capacity()
        return out


# ================================================================================================
# MSA update with coordinates
# ================================================================================================
class MsaUpdateWithPairAndCoord(RFModule):
    """rf.py:865-920."""

    def __init__(self, d_msa, d_state, d_trfm_inner, d_ff, distance_bins=[8, 12, 16, 20], p_dropout=0.1):
        super().__init__()
        self.distance_bins = distance_bins
        self.n_heads = len(distance_bins)
        self.d_inner = d_trfm_inner
        self.scale = (d_state // self.n_heads) ** -0.5  # rf.py:874
        self.ln_msa = LayerNorm(d_msa)
        self.ln_state = LayerNorm(d_state)
        self.to_q = Linear(d_state, d_trfm_inner * self.n_heads)
        self.to_k = Linear(d_state, d_trfm_inner * self.n_heads)
        self.to_v = Linear(d_msa, d_msa)
        self.ln_out = LayerNorm(d_msa)
        self.to_out = Residual(nn.Sequential(LayerNorm(d_msa), FeedForward(d_msa, d_ff, p_dropout)))

    def run(self, xyz, state, msa):
        """returns the new fp32 msa [B,N,L,D] (the residual base is LayerNorm(msa), rf.py:893,918)."""
        B, N, Lr, D = msa.shape
        H, dq = self.n_heads, self.d_inner
        dv = D // H
        dev = msa.device
        st = ln(self.ln_state, state.contiguous(), out_dtype=F32)
        m32 = ln(self.ln_msa, msa, out_dtype=F32)
        m_t = ops.cast(m32, T())
        q = ops.linear(st, self.to_q.weight.detach(), _f(self.to_q.bias), out_dtype=F32)
        ops.axpby(q, self.scale, None, 0.0, q)
        k = ops.linear(st, self.to_k.weight.detach(), _f(self.to_k.bias), out_dtype=F32)
        bins = self.cached("bins", lambda: torch.tensor(self.distance_bins, dtype=F32, device=dev))
        att = torch.empty(B, H, Lr, Lr, device=dev, dtype=T())
        ops.dist_masked_attention(q, k, xyz.contiguous(), bins, att, B, Lr, H, dq)
        v_t = torch.empty(B, N, D, Lr, device=dev, dtype=T())
        ops.gemm(self.wt("v", self.to_v), m_t, v_t, D, Lr, D, batch=(B * N, 1, 1), b_bs=(Lr * D, 0, 0),
                 c_bs=(D * Lr, 0, 0), c_row=(0, 0, Lr), bias=_f(self.to_v.bias), bias_mode=L.BIAS_ROW)
        out = torch.empty(B, N, Lr, D, device=dev, dtype=F32)
        ops.gemm(att, v_t, out, Lr, N * dv, Lr, batch=(B, H, 1),
                 a_bs=(H * Lr * Lr, Lr * Lr, 0), a_row=(0, 0, Lr),
                 b_bs=(N * D * Lr, dv * Lr, 0), b_row=(dv, D * Lr, Lr),
                 c_bs=(N * Lr * D, dv, 0), c_row=(0, 0, D), c_col=(dv, Lr * D))
        o = ln(self.ln_out, out, out_dtype=F32)
        ops.axpby(m32, 1.0, o, 1.0, m32)
        self.to_out.fn[1].apply_residual(ln(self.to_out.fn[0], m32), m32)
        return m32

    def forward(self, xyz, state, msa):
        return self.run(xyz.float(), state.float(), msa.float().contiguous())


# ================================================================================================
# blocks and model
# ================================================================================================
def _whole_pair(pair, row_group):
    """The whole pair tensor on every rank from the row blocks (identity without a row group)."""
    if row_group is None:
        return pair
    from . import shard
    return shard.all_gather_rows(pair, row_group)


class TwoTrackBlock(RFModule):
    """rf.py:923-968."""

    def __init__(self, d_msa, d_pair, n_encoder_layers, p_dropout=0.1):
        super().__init__()
        self.msa_update_using_self_att = MsaUpdateUsingSelfAttention(d_msa=d_msa, d_ff=d_msa * 4, n_heads=12,
                                                                     n_encoder_layers=n_encoder_layers,
                                                                     p_dropout=p_dropout)
        self.pair_update_with_msa = PairUpdateWithMsa(d_pair=d_pair, n_heads=12, d_msa=d_msa, d_proj=32)
        self.pair_update_with_axial_attention = PairUpdateWithAxialAttention(d_pair=d_pair, d_ff=d_pair * 4, n_heads=8,
                                                                            p_dropout=p_dropout,
                                                                            n_encoder_layers=n_encoder_layers,
                                                                            performer_kws={})
        self.msa_update_with_pair = MsaUpdateWithPair(d_msa=d_msa, d_pair=d_pair, n_heads=4,
                                                      n_encoder_layers=n_encoder_layers, p_dropout=p_dropout)

    def run(self, msa, pair, row_group=None):
        """msa updated in place; returns the new pair tensor.
        row_group: `pair` is this rank's block of rows shard_range(L, world, rank) of the pair tensor, msa is replicated
        (pair-track row-block sharding, shard.two_track_block_row_sharded): the MSA self-attention runs on every rank, the three
        pair-track modules run on the row block with their exchanges."""
        att = self.msa_update_using_self_att.run(msa)
        if row_group is None:
            pair = self.pair_update_with_msa.run(msa, pair, att)
            self.pair_update_with_axial_attention.run(pair)
            self.msa_update_with_pair.run(msa, pair)
            return pair
        pair = self.pair_update_with_msa.run_rows(msa, pair, att, row_group)
        self.pair_update_with_axial_attention.run(pair, row_group=row_group)
        self.msa_update_with_pair.run(msa, pair, row_group=row_group)
        return pair

    def forward(self, msa, pair):
        msa = fresh_f32(msa)
        pair = self.run(msa, pair.float().contiguous())
        return msa, pair


def _set_dropout(module, p):
    """Set every dropout probability below `module` (the nn.Dropout containers and the p_dropout fields the training-mode
    forward reads) -- the reference hard-codes 0.1 for the MsaUpdateWithPair of its three-track / final blocks."""
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = p
        if getattr(m, "p_dropout", None) is not None:
            m.p_dropout = p


class ThreeTrackBlock(TwoTrackBlock):
    """rf.py:971-1046."""

    def __init__(self, d_msa, d_pair, d_node, d_edge, d_state, n_encoder_layers, n_neighbors, p_dropout):
        super().__init__(d_msa, d_pair, n_encoder_layers, p_dropout)
        _set_dropout(self.msa_update_with_pair, 0.1)   # rf.py:1015 (whatever the model's p_dropout)
        self.coord_update_with_msa_and_pair = CoordUpdateWithMsaAndPair(d_msa=d_msa, d_pair=d_pair, d_node=d_node,
                                                                        d_edge=d_edge, d_state=d_state,
                                                                        n_neighbors=n_neighbors, p_dropout=p_dropout)
        self.msa_update_with_pair_and_coord = MsaUpdateWithPairAndCoord(d_msa=d_msa, d_state=d_state, d_trfm_inner=32,
                                                                        d_ff=d_msa * 4, distance_bins=[8, 12, 16, 20],
                                                                        p_dropout=p_dropout)

    def run3(self, msa, pair, xyz, seq_onehot, aa_idx, monotonic=True, row_group=None):
        """row_group: `pair` is this rank's block of rows (TwoTrackBlock.run); the structure track runs replicated on the
        all-gathered pair tensor (it reads the pair rows of the kNN edges of every residue)."""
        pair = self.run(msa, pair, row_group)
        state, xyz = self.coord_update_with_msa_and_pair.run(xyz, msa, _whole_pair(pair, row_group), aa_idx, seq_onehot, monotonic)
        msa = self.msa_update_with_pair_and_coord.run(xyz, state, msa)
        return msa, pair, xyz

    def forward(self, msa, pair, xyz, seq_onehot, aa_idx):
        msa = fresh_f32(msa)
        mono = check_index_range(None, None, aa_idx.contiguous(), 1, 2 ** 31 - 1)
        out = self.run3(msa, pair.float().contiguous(), xyz.float(), seq_onehot.float(), aa_idx, mono)
        check_edge_capacity()
        return out


class FinalBlock(TwoTrackBlock):
    """rf.py:1049-1127."""

    def __init__(self, d_msa, d_pair, d_node, d_edge, d_state, n_encoder_layers, p_dropout, n_neighbors=32):
        super().__init__(d_msa, d_pair, n_encoder_layers, p_dropout)
        _set_dropout(self.msa_update_with_pair, 0.1)   # rf.py:1101
        self.coord_update_with_msa_and_pair = CoordUpdateWithMsaAndPair(d_msa=d_msa, d_pair=d_pair, d_node=d_node,
                                                                        d_edge=d_edge, d_state=d_state,
                                                                        n_neighbors=n_neighbors, p_dropout=p_dropout)
        self.plddt_head = Linear(d_state, 1)

    def run3(self, msa, pair, xyz, seq_onehot, aa_idx, monotonic=True, row_group=None):
        pair = self.run(msa, pair, row_group)
        state, xyz = self.coord_update_with_msa_and_pair.run(xyz, msa, _whole_pair(pair, row_group), aa_idx, seq_onehot, monotonic)
        plddt = ops.linear(state.contiguous(), self.plddt_head.weight.detach(), _f(self.plddt_head.bias), out_dtype=F32)
        return msa, pair, xyz, plddt[..., 0]

    def forward(self, msa, pair, xyz, seq_onehot, aa_idx):
        msa = fresh_f32(msa)
        mono = check_index_range(None, None, aa_idx.contiguous(), 1, 2 ** 31 - 1)
        out = self.run3(msa, pair.float().contiguous(), xyz.float(), seq_onehot.float(), aa_idx, mono)
        check_edge_capacity()
        return out


class RoseTTAFold(RFModule):
    """rf.py:1175-1289: RoseTTAFold(msa, seq, aa_idx) -> (logits dict, xyz [B,L,3,3], plddt [B,L])."""

    def __init__(self, d_input=21, d_msa=384, d_pair=288, d_node=64, d_edge=64, d_state=32, n_two_track_blocks=3,
                 n_three_track_blocks=4, n_encoder_layers=4, max_len=5000, n_neighbors=[128, 128, 64, 64, 64],
                 p_dropout=0.1, use_template=False):
        super().__init__()
        self.d_msa, self.d_pair, self.d_node, self.d_edge, self.d_state = d_msa, d_pair, d_node, d_edge, d_state
        self.n_two_track_blocks, self.n_three_track_blocks = n_two_track_blocks, n_three_track_blocks
        self.n_encoder_layers, self.use_template = n_encoder_layers, use_template
        self.msa_emb = MsaEmbedding(d_input=d_input, d_msa=d_msa, max_len=max_len, p_pe_drop=p_dropout)
        self.pair_emb = PairEmbedding(d_input=d_input, d_pair=d_pair, max_len=max_len, use_template=use_template,
                                      p_pe_drop=p_dropout)
        self.two_track_blocks = nn.ModuleList([TwoTrackBlock(d_msa, d_pair, n_encoder_layers=n_encoder_layers,
                                                             p_dropout=p_dropout) for _ in range(n_two_track_blocks)])
        self.initial_coord_generation_with_msa_and_pair = InitialCoordGenerationWithMsaAndPair(
            d_msa=d_msa, d_pair=d_pair, d_node=d_node, d_edge=d_edge, n_heads=4, n_layers=4, p_dropout=p_dropout)
        self.three_track_blocks = nn.ModuleList([
            ThreeTrackBlock(d_msa, d_pair, d_node, d_edge, d_state, n_encoder_layers=n_encoder_layers,
                            n_neighbors=n_neighbors[i], p_dropout=p_dropout) for i in range(n_three_track_blocks - 1)])
        self.final_block = FinalBlock(d_msa, d_pair, d_node, d_edge, d_state, n_encoder_layers=n_encoder_layers,
                                      n_neighbors=32, p_dropout=p_dropout)
        self.prediction_head = PredictionHead(in_channels=d_pair, n_res_blocks=4, p_dropout=p_dropout)

    @torch.no_grad()
    def forward(self, msa, seq, aa_idx):
        if not msa.is_cuda:
            raise L.RfmiError("RoseTTAFold (MI355X build) needs device tensors; there is no CPU fallback")
        with torch.cuda.device(msa.device):  # device guard: kernels go to the inputs' GPU, whatever the caller's current one
            msa, seq, aa_idx = msa.contiguous(), seq.contiguous(), aa_idx.contiguous()
            # the reference raises IndexError for out-of-range tokens / residue indices (nn.Embedding, rf.py:73,98)
            mono = check_index_range(msa, seq, aa_idx, self.msa_emb.to_embedding.num_embeddings,
                                     min(self.msa_emb.pos_enc.max_len, self.pair_emb.pos_enc.max_len))
            _PENDING_EDGE_COUNTS.clear()
            out = self.forward_validated(msa, seq, aa_idx, mono)
            check_edge_capacity()
            return out

    @torch.no_grad()
    def forward_validated(self, msa, seq, aa_idx, mono=True, row_group=None):
        """forward() behind the input validation: launches only, no host read of device data, so it can be recorded into
        a hipGraph (graph.GraphedForward).  `mono` = aa_idx strictly increasing in every sample (what forward() checks).
        row_group: ONE sample spread over the ranks of a torch.distributed group (shard.forward_row_sharded): every rank gets the
        same inputs, the pair track runs on row blocks shard_range(L, world, rank), the MSA and structure tracks are replicated;
        returns this rank's rows of the logit maps and the whole xyz / plddt."""
        with torch.cuda.device(msa.device):
            m = self.msa_emb.run(msa, aa_idx)
            p = self.pair_emb.run(seq, aa_idx)
            if row_group is not None:
                from . import shard
                p = shard.take_rows(p, row_group)
            onehot = ops.onehot(seq, 21)  # rf.py:1276
            for blk in self.two_track_blocks:
                p = blk.run(m, p, row_group)
            xyz = self.initial_coord_generation_with_msa_and_pair.run(m, _whole_pair(p, row_group), onehot, aa_idx)
            for blk in self.three_track_blocks:
                m, p, xyz = blk.run3(m, p, xyz, onehot, aa_idx, mono, row_group)
            m, p, xyz, plddt = self.final_block.run3(m, p, xyz, onehot, aa_idx, mono, row_group)
            logits = self.prediction_head.run(p, row_group)
        return logits, xyz, plddt


def flat_state(model):
    """state_dict as fp32 CPU tensors (the form the CPU oracle consumes)."""
    return {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
