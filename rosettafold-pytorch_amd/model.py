"""RoseTTAFold forward path on MI355X: the reference's nn.Module call surface over librfmi.so.

Class names, constructor arguments, `state_dict()` key names and forward signatures mirror
rosettafold_pytorch/rosettafold_pytorch.py (cited per class as rf.py:LINE) so that a user of the
reference can switch packages; the arithmetic inside every forward is a sequence of HIP kernels
reached through the C ABI (include/rfmi.h).  PyTorch supplies device memory, the stream and the
parameter containers (nn.Linear / nn.LayerNorm / ... are used for their weights and default init
only -- their own forward is never called).

Inference semantics: every dropout is the identity (the reference's unregistered layer lists
ignore .eval(); here they are proper ModuleLists and their weights appear in
state_dict under `...encoder_layers.N.` / `...blocks.N.`).

Precision policy: the two residual streams (msa, pair) are fp32 in HBM; GEMM operands are the
compute dtype (bf16 by default -> v_mfma_f32_16x16x32_bf16 with fp32 accumulation, or fp32 ->
exact fp32 tiles for parity runs); LayerNorm / InstanceNorm / softmax statistics are fp32; the
SE(3) structure module is fp32 end to end, as in the reference (se3_modules.py:164).
"""
import math

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import F32

N_IDX, CA_IDX, C_IDX = 0, 1, 2  # rf.py:15
M_FEAT = 266   # performer nb_features = int(64 * ln 64)
M_PAD = 288    # padded to a multiple of 32 for the MFMA K loop
VT_ROWS = 80   # 64 value rows + the ones row (k' sums) padded to a multiple of 16


class _Runtime:
    dtype = torch.bfloat16
    # training-mode dropout: masks are Philox4x32-10(train_seed, counter); every dropout call of a forward
    # takes the next ceil(n / 4) counters, so manual_seed(s) in front of a forward reproduces it bit for bit (csrc/ops.hip)
    train_seed = 0
    train_offset = 0
    cache_epoch = 0  # bumped whenever kernel-ready weight copies are dropped (graph.GraphedForward re-records on a change)
    # structure-track node input (LayerNorm(msa) -> position-weighted sum, rf.py:789-798) in fp32 also in the 16-bit modes:
    # the SE(3) stack is discontinuous (GNormBias, kNN, distance bins), so its inputs are not the place to round
    # (tools/depth_parity.py --struct-lowp measures the difference)
    struct_inputs_fp32 = True
    # PredictionHead: remove the per-(sample, channel) mean over the picture from the projected pair tensor before it is rounded
    # to the 16-bit operand type (PredictionHead.run; RF_HEAD_CENTER=0 restores the plain cast)
    head_center = bool(int(__import__("os").environ.get("RF_HEAD_CENTER", "1")))
    # Operand conditioning of the 16-bit modes (csrc/condition.hip; exact algebra): PairUpdateWithMsa's tiled 1-D features and its
    # first convolution see operands with the per-sample constant removed.  RF_CONDITION=0: the plain form (ablation / probes).
    condition = bool(int(__import__("os").environ.get("RF_CONDITION", "1")))
    condition_values = bool(int(__import__("os").environ.get("RF_CONDITION_V", "1")))   # the attention layers' value path (value_conditioning)
    # SE(3) radial MLPs: last Linear inside the message kernel (csrc/se3.hip: rf_se3_radial_message); RF_SE3_UNFUSED=1 writes the
    # radial outputs with a K = 32 GEMM and reads them back (round-3 path, kept for A/B timing and as the form for unusual shapes)
    se3_fused_radial = not bool(int(__import__("os").environ.get("RF_SE3_UNFUSED", "0")))
    fused_favor = True  # use the fused FAVOR+ kernel when the shape allows (bf16, dim_head 64, seq 128/256)
    fused_outer_ln = not bool(int(__import__("os").environ.get("RF_NO_FUSED_OUTER_LN", "0")))  # LayerNorm(1024) in the outer-product GEMM epilogue
    fused_tied = not bool(int(__import__("os").environ.get("RF_NO_FUSED_TIED", "0")))  # tied-attention logits + softmax in one launch
    fused_outer = not bool(int(__import__("os").environ.get("RF_NO_FUSED_OUTER", "0")))  # outer product -> LN -> Linear in one kernel
    tied_v2 = not bool(int(__import__("os").environ.get("RF_TIED_V1", "0")))  # head-major q|k|v + collapsed weights + A.V kernel
    tied_fold_w = not bool(int(__import__("os").environ.get("RF_TIED_NO_FOLD", "0")))  # position weights folded into q by the projection's epilogue
    # pair-track row blocks (shard.forward_row_sharded): how the attention direction that crosses the blocks is computed --
    # "transpose" (two transposing exchanges, fused kernel) or "contexts" (all-reduce of the Performer contexts, GEMM chain)
    rowshard_attention = __import__("os").environ.get("RF_ROWSHARD_ATTENTION", "transpose")
    head_major_qkv = int(__import__("os").environ.get("RF_HEAD_MAJOR_QKV", "0"))  # 1: every FAVOR+ layer, 2: only where the sequence is the inner row index
    # Producer -> consumer chains whose intermediate (q|k|v, feed-forward hidden) is larger than this many bytes are run
    # panel by panel, so the intermediate panel is still in the 256 MB Infinity Cache when its consumer reads it
    # (tools/mall_chunk_bench.py: projection + FAVOR alone 958 -> 842 us at 200 MB panels; inside the full forward the
    # step time did not move, 432 vs 437 ms, so it is opt-in: RF_MALL_PANEL_MB=208).  0 disables.
    mall_panel_bytes = int(__import__("os").environ.get("RF_MALL_PANEL_MB", "0")) << 20


def row_panels(rows, bytes_per_row, unit):
    """Split `rows` into equal panels of whole `unit`s (unit % 256 == 0 keeps every panel on the persistent GEMM path) whose
    intermediate stays under RT.mall_panel_bytes.  Returns the panel length (== rows when no split applies)."""
    cap = RT.mall_panel_bytes
    if cap <= 0 or rows * bytes_per_row <= cap or rows % unit:
        return rows
    units = rows // unit
    for n in range(2, units + 1):
        if units % n == 0 and (rows // n) * bytes_per_row <= cap and (rows // n) >= 16384:
            return rows // n
    return rows


RT = _Runtime()


def set_compute_dtype(dtype):
    """Operand type of the dense contractions (accumulation, residual streams, statistics and the structure track are
    fp32 in every mode):
      torch.bfloat16  v_mfma_f32_16x16x32_bf16, librfmi.so (default: the dtype BASELINE.json quotes the metric on);
      torch.float16   v_mfma_f32_16x16x32_f16, librfmi_f16.so: same rate and bytes, 8x smaller operand rounding
                      (11 significand bits), fp16 range -- the mode that carries the whole-model parity claim at speed;
      torch.float32   exact fp32 tiles (v_mfma_f32_16x16x4_f32, 1/16 of the rate): the strict oracle-parity mode.
    Kernel-ready weight copies are cached per dtype, so switching back and forth costs nothing after the first forward."""
    if dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise TypeError("compute dtype must be bfloat16, float16 or float32")
    RT.dtype = dtype
    L.select_h16(L.RF_F16 if dtype == torch.float16 else L.RF_BF16)


def T():
    return RT.dtype


def manual_seed(seed):
    """Seed of the training-mode dropout masks (model.train(); the inference forward draws nothing).  Like torch.manual_seed:
    the same seed in front of the same forward gives the same masks; consecutive forwards continue the counter stream."""
    RT.train_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    RT.train_offset = 0


def dropout_(t, p):
    """In-place nn.Dropout(p) of a training-mode forward on a contiguous fp32 / 16-bit tensor (rf_dropout); identity for p <= 0."""
    if p is None or p <= 0.0 or t.numel() == 0:
        return t
    if p >= 1.0:
        return ops.fill(t, 0.0)
    if not t.is_contiguous():
        raise ValueError("dropout_: contiguous tensors only")
    off = RT.train_offset
    RT.train_offset += (t.numel() + 3) // 4
    return ops.dropout(t, p, RT.train_seed, off)


def _p(drop_module):
    """p of an nn.Dropout container (0 for anything else)."""
    return float(drop_module.p) if isinstance(drop_module, nn.Dropout) else 0.0


def add_dropped(x_res, compute, drops):
    """x_res += dropout_{p_k}(... dropout_{p_1}(f)) for the training-mode forward: `compute(tmp)` ADDS f to the zeroed fp32 tmp
    (every producer of the forward path accumulates into a residual stream: handing it zeros yields f itself)."""
    tmp = ops.zeros(*x_res.shape, device=x_res.device, dtype=F32)
    compute(tmp)
    for pk in drops:
        dropout_(tmp, pk)
    return ops.axpby(x_res, 1.0, tmp, 1.0, x_res)


def pad8(n):
    return (n + 7) // 8 * 8


def weights_fingerprint(module):
    """Changes whenever a parameter / buffer of `module` is rebound, moved or edited in place THROUGH THE TENSOR ITSELF
    (p.copy_, nn.init.*, optimizer steps bump `_version`; `p.data = other`, `.to()` change `data_ptr`).
    NOT seen: in-place edits through the `.data` alias (`p.data.copy_(w)`, `p.data.normal_()`, EMA `p.data.lerp_`): `.data`
    is a detached tensor with a version counter of its own, and a content probe would cost a device read-back per
    forward.  After such an edit call `invalidate_weight_caches(model)` (tests/test_boundary_gpu.py shows both cases);
    `load_state_dict`, `load_reference_weights`, `load_checkpoint`, `.to()` and `set_compute_dtype` need nothing."""
    h = 0
    for t in list(module.parameters()) + list(module.buffers()):
        h = (h * 1000003 + t.data_ptr() + 7919 * t._version) & 0xFFFFFFFFFFFFFFF
    return h


def invalidate_weight_caches(module):
    """Drop every kernel-ready weight copy held below `module` (they are rebuilt on the next call)."""
    RT.cache_epoch += 1
    for m in module.modules():
        c = getattr(m, "_rfc", None)
        if isinstance(c, dict):
            c.clear()
        elif c is not None:
            object.__setattr__(m, "_rfc", None)


class RFModule(nn.Module):
    """nn.Module with a cache of kernel-ready weights (cast / concatenated / padded).  The cache is validated against the
    live parameters at every PUBLIC call (`module(...)`): internal composition goes through `.run()` / `.attend()`, so
    the check costs one pass over the parameter list per user-level call, not per kernel."""

    def __init__(self):
        super().__init__()
        # Constructed in EVAL mode (nn.Module's default is training): this build's forward is the inference path, and the
        # reference hard-codes non-zero dropout probabilities in places (rf.py:1015,1101; PairUpdateWithMsa's default) that a
        # caller passing p_dropout=0 does not reach.  model.train() switches every dropout site on (Philox masks, manual_seed).
        self.training = False
        object.__setattr__(self, "_rfc", {})
        object.__setattr__(self, "_rf_fp", None)

    def _apply(self, fn, *a, **k):
        invalidate_weight_caches(self)
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._rfc.clear()
        RT.cache_epoch += 1
        return super()._load_from_state_dict(*a, **k)

    def __call__(self, *a, **k):
        fp = weights_fingerprint(self)
        if fp != self._rf_fp:
            if self._rf_fp is not None:
                invalidate_weight_caches(self)
            object.__setattr__(self, "_rf_fp", fp)
        return super().__call__(*a, **k)

    def cached(self, key, fn):
        k = (key, RT.dtype)
        v = self._rfc.get(k)
        if v is None:
            with torch.no_grad():
                v = fn()
            self._rfc[k] = v
        return v

    # operand conditioning of an attention layer's value path (16-bit modes; csrc/condition.hip) -----------------------------------
    def value_conditioning(self, xn, to_v, to_out, lead=()):
        """The attention output o = sum_j a_ij v_j (a_i. sums to one: softmax rows, or FAVOR's ratio) is a per-sample constant
        c ~ mean_j v_j plus a part 10-30x smaller at random init, and it is the 16-bit operand of the output projection:
        rounded with the constant it was 4e-2 of the bf16 mode's 5e-2 logits gap (tools/precision_probe.py --gemm-sweep
        msa_update_using_self_att:bf16; the linear attention's context, rounded for its second GEMM, another 2e-2).  With
        mu = an estimate of mean(xn) (rf_sample_mean) and c = W_v mu + b_v:
            v - c = W_v xn - W_v mu            the value block of the projection gets the bias -W_v mu instead of b_v,
            o - c = sum_j a_ij (v_j - c)       the attention kernels run unchanged on the centred values,
            W_o (o - c) + (b_o + W_o c)        the output projection gets the bias b_o + W_o b_v + (W_o W_v) mu.
        ONE constant serves the whole batch (the identity holds for any c; the constant is dominated by biases, LayerNorm
        offsets and the mean embedding, which the samples share), so the projections stay single launches over all samples.
        Returns (fp32 [1, n_lead + d_v] bias of the projection whose output columns are [lead..., v], fp32 [1, d_out] output bias);
        `lead`: the Linear modules in front of v in a fused projection (their biases pass through)."""
        def make():
            wv, wo = to_v.weight.detach().float(), to_out.weight.detach().float()
            dev = wv.device
            bl = [(_f(m_.bias) if m_.bias is not None else torch.zeros(m_.weight.shape[0], device=dev)) for m_ in lead]
            bo = _f(to_out.bias) if to_out.bias is not None else torch.zeros(wo.shape[0], device=dev)
            if to_v.bias is not None:
                bo = bo + wo @ _f(to_v.bias)
            nl = sum(m_.weight.shape[0] for m_ in lead)
            wc = torch.cat([torch.zeros(nl, wv.shape[1], device=dev), -wv, wo @ wv]).contiguous()
            bc = torch.cat(bl + [torch.zeros(wv.shape[0], device=dev), bo]).contiguous()
            return wc, bc, nl + wv.shape[0]
        wc, bc, npre = self.cached(("vcond", len(lead)), make)
        r = ops.fold_mean(wc, ops.sample_mean(xn.view(1, -1, xn.shape[-1])), bc)
        return r[:, :npre], r[:, npre:]

    # kernel-ready views of parameter containers -------------------------------------------------
    def wt(self, key, lin, kpad=None):
        def make():
            w = lin.weight.detach().reshape(lin.weight.shape[0], -1)
            if kpad is not None and kpad != w.shape[1]:
                w = torch.cat([w, w.new_zeros(w.shape[0], kpad - w.shape[1])], 1)
            return w.to(T()).contiguous()
        return self.cached(("wt", key), make)

    def wcat(self, key, lins):
        return self.cached(("wcat", key), lambda: torch.cat([l.weight.detach() for l in lins], 0).to(T()).contiguous())

    def bcat(self, key, lins):
        return self.cached(("bcat", key), lambda: torch.cat([l.bias.detach() for l in lins], 0).float().contiguous())


def _f(p):
    return None if p is None else p.detach()


def fresh_f32(x):
    """A new contiguous fp32 copy of x (the public forwards never mutate their inputs)."""
    y = torch.empty(x.shape, device=x.device, dtype=F32)
    return ops.axpby(x.detach().contiguous(), 1.0, None, 0.0, y)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm as a parameter container (same state_dict keys); a direct call runs rf_layernorm, not ATen."""

    def forward(self, x):
        return ops.layernorm(x.contiguous(), _f(self.weight), _f(self.bias), eps=self.eps, out_dtype=F32)


class Linear(nn.Linear):
    """nn.Linear as a parameter container; a direct call runs rf_gemm in the current compute dtype (fp32 result)."""

    def forward(self, x):
        xt = ops.cast(x.contiguous(), T())
        return ops.linear(xt, ops.cast(self.weight.detach().contiguous(), T()), _f(self.bias), out_dtype=F32)


def project_into_residual(x, w, bias, x_res, next_ln, drops=(), xn_out=None):
    """x_res += x @ w^T + bias, with the next LayerNorm when the fused epilogue applies (ops.linear_residual_ln); training mode:
    `drops` = the dropouts the reference applies to the projected tensor before the residual add (then returns None)."""
    drops = tuple(d for d in drops if d and d > 0)
    if drops:
        add_dropped(x_res, lambda tmp: ops.linear(x, w, bias, out=tmp, residual=tmp), drops)
        return None
    return ops.linear_residual_ln(x, w, bias, x_res, next_ln, xn_out)


def ln(mod, x, out_dtype=None, **kw):
    return ops.layernorm(x, _f(mod.weight), _f(mod.bias), eps=mod.eps, out_dtype=out_dtype or T(), **kw)


# ================================================================================================
# small building blocks
# ================================================================================================
class Residual(nn.Module):
    """rf.py:18-28: fn(x) + x (dropout is the identity at inference).  The model's own compositions fuse the residual
    add into the producing GEMM's epilogue; this forward serves a directly called / user-wrapped module."""

    def __init__(self, fn, p_dropout=None):
        super().__init__()
        self.training = False   # (eval by default, like every module of this build: RFModule.__init__)
        self.fn = fn
        self.p_dropout = p_dropout

    def forward(self, x):
        fx = self.fn(x)
        if isinstance(fx, tuple):
            raise TypeError("Residual wraps modules that return one tensor")
        if self.training and self.p_dropout:   # rf.py:25-26: dropout(fn(x)) + x
            fx = dropout_(fx.float().contiguous().clone(), self.p_dropout)
        out = torch.empty(x.shape, device=x.device, dtype=F32)
        return ops.axpby(fx.contiguous(), 1.0, x.contiguous(), 1.0, out)


class ColWise(nn.Module):
    """rf.py:31-41: fn over the sequences x[b, n, :, :] (attention along dim 2)."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        b, n, l, d = x.shape
        if isinstance(self.fn, PerformerSelfAttention):  # strided addressing inside the kernels: no reshape
            out = ops.zeros(b, n, l, d, device=x.device, dtype=F32)
            self.fn.attend(ops.cast(x.contiguous(), T()), out, axis=2)
            return out
        return self.fn(x.reshape(b * n, l, d)).reshape(b, n, l, -1)


class RowWise(nn.Module):
    """rf.py:44-54: fn over the sequences x[b, :, l, :] (attention along dim 1)."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        b, n, l, d = x.shape
        if isinstance(self.fn, PerformerSelfAttention):
            out = ops.zeros(b, n, l, d, device=x.device, dtype=F32)
            self.fn.attend(ops.cast(x.contiguous(), T()), out, axis=1)
            return out
        xt = torch.empty(b, l, n, d, device=x.device, dtype=x.dtype)
        ops.copy4d(x.contiguous(), (n * l * d, d, l * d, 1), xt, (l * n * d, n * d, d, 1), (b, l, n, d))
        y = self.fn(xt.view(b * l, n, d)).contiguous()
        do = y.shape[-1]
        out = torch.empty(b, n, l, do, device=x.device, dtype=y.dtype)
        ops.copy4d(y, (l * n * do, n * do, do, 1), out, (n * l * do, do, l * do, 1), (b, l, n, do))
        return out


class FeedForward(RFModule):
    """rf.py:270-281."""

    def __init__(self, d_emb, d_ff, p_dropout=0.1):
        super().__init__()
        self.net = nn.Sequential(Linear(d_emb, d_ff), nn.ReLU(), nn.Dropout(p_dropout), Linear(d_ff, d_emb))

    def apply_residual(self, xn, x_res, next_ln=None, drops=()):
        """x_res += W2 relu(W1 xn + b1) + b2   (x_res fp32, in place).  With `next_ln` the second GEMM's epilogue also
        emits next_ln(x_res) (returned, or None when the fused form does not apply).
        Training mode (self.training): dropout on the hidden activations (rf.py:276) and, from the wrapping module, `drops` on
        the block's output before the residual add (rf.py:25-26, 330) -- two GEMMs with the hidden tensor in memory."""
        ph = _p(self.net[2]) if self.training else 0.0
        drops = tuple(d for d in drops if d and d > 0)
        if ph > 0 or drops:
            w1, b1, w2, b2 = self.wt("w1", self.net[0]), _f(self.net[0].bias), self.wt("w2", self.net[3]), _f(self.net[3].bias)
            h = dropout_(ops.linear(xn, w1, b1, act=L.ACT_RELU), ph)
            add_dropped(x_res, lambda tmp: ops.linear(h, w2, b2, out=tmp, residual=tmp), drops)
            return None
        if ops.ffn_fused_applies(xn, x_res, self.net[0].in_features, self.net[0].out_features):
            # one launch, hidden activations on chip (csrc/ffn.hip); the weights are packed once per module
            wp = self.cached("ffn_packed", lambda: ops.ffn_pack(self.net[0].weight, self.net[3].weight, T()))
            return ops.ffn_fused(xn, wp, _f(self.net[0].bias), _f(self.net[3].bias), x_res, next_ln)
        w1, b1, w2, b2 = self.wt("w1", self.net[0]), _f(self.net[0].bias), self.wt("w2", self.net[3]), _f(self.net[3].bias)
        d_ff, R = w1.shape[0], xn.numel() // xn.shape[-1]
        pr = row_panels(R, d_ff * xn.element_size(), 256) if x_res.is_contiguous() and xn.is_contiguous() else R
        if pr < R:  # hidden panel stays in the Infinity Cache between the two GEMMs
            xn2, xr2 = xn.view(R, -1), x_res.view(R, -1)
            h = torch.empty(pr, d_ff, device=xn.device, dtype=xn.dtype)
            nxt, fused = torch.empty(R, xr2.shape[1], device=xn.device, dtype=xn.dtype) if next_ln is not None else None, True
            for r0 in range(0, R, pr):
                ops.linear(xn2[r0:r0 + pr], w1, b1, act=L.ACT_RELU, out=h)
                fused &= ops.linear_residual_ln(h, w2, b2, xr2[r0:r0 + pr], next_ln, nxt[r0:r0 + pr] if nxt is not None else None) is not None
            return nxt.view(x_res.shape) if nxt is not None and fused else None
        h = ops.linear(xn, w1, b1, act=L.ACT_RELU)
        return ops.linear_residual_ln(h, w2, b2, x_res, next_ln)

    def forward(self, x):
        xn = ops.cast(x.contiguous(), T())
        out = ops.zeros(*x.shape, device=x.device, dtype=F32)
        self.apply_residual(xn, out)
        return out


def sinusoid_table(dim, max_len):
    """rf.py:63-68 (host-side constant table)."""
    pe = torch.zeros(max_len, dim)
    denom = torch.exp(math.log(10000.0) * torch.arange(0, dim, 2) / dim)
    pos = torch.arange(0, max_len).view(-1, 1)
    pe[:, 0::2] = torch.sin(pos / denom)
    pe[:, 1::2] = torch.cos(pos / denom)
    return pe


def check_index_range(msa, seq, aa_idx, d_input, max_len):
    """The reference raises IndexError from nn.Embedding / table indexing (rf.py:73,98,115-119,155) for a token outside
    [0, d_input) or a residue index outside [0, max_len); the kernels index unchecked, so the wrappers validate first
    (one small launch + one 12-byte read-back).  Returns whether aa_idx is strictly increasing inside every sample."""
    bad_tok, bad_idx, not_mono = ops.check_inputs(msa, seq, aa_idx, d_input, max_len)
    if bad_tok:
        raise IndexError(f"token index out of range [0, {d_input})")
    if bad_idx:
        raise IndexError(f"aa_idx out of range [0, {max_len}) (max_len of the positional-encoding table)")
    return not not_mono


class SinusoidalPositionalEncoding(RFModule):
    """rf.py:57-76."""

    def __init__(self, dim, max_len, p_dropout=0.1):
        super().__init__()
        self.dim, self.max_len = dim, max_len
        self.p_dropout = p_dropout
        self.register_buffer("pos_enc", sinusoid_table(dim, max_len), persistent=False)

    def forward(self, x, aa_idx):
        """x [B,N,L,dim] + pos_enc[aa_idx] broadcast over N (rf.py:72-76; dropout: training mode only)."""
        aa_idx = aa_idx.to(x.device).contiguous()
        check_index_range(None, None, aa_idx, 1, self.max_len)
        y = ops.add_pos_enc(x.float().contiguous(), aa_idx, self.pos_enc, two_d=False)
        return dropout_(y, self.p_dropout) if self.training else y


class SinusoidalPositionalEncoding2D(RFModule):
    """rf.py:79-103."""

    def __init__(self, dim, max_len, p_dropout=0.1):
        super().__init__()
        self.max_len = max_len
        self.register_buffer("pos_enc", sinusoid_table(dim // 2, max_len), persistent=False)

    def forward(self, x, aa_idx):
        """x [B,L,L,dim] + [pos_enc[idx_i] | pos_enc[idx_j]] (rf.py:95-103)."""
        aa_idx = aa_idx.to(x.device).contiguous()
        check_index_range(None, None, aa_idx, 1, self.max_len)
        return ops.add_pos_enc(x.float().contiguous(), aa_idx, self.pos_enc, two_d=True)


class MsaEmbedding(RFModule):
    """rf.py:106-120."""

    def __init__(self, d_input=21, d_msa=384, max_len=260, p_pe_drop=0.1):
        super().__init__()
        self.to_embedding = nn.Embedding(d_input, d_msa)
        self.pos_enc = SinusoidalPositionalEncoding(d_msa, max_len, p_pe_drop)
        self.query_enc = nn.Embedding(2, d_msa)

    def forward(self, x, aa_idx):
        check_index_range(x.contiguous(), None, aa_idx.contiguous(), self.to_embedding.num_embeddings, self.pos_enc.max_len)
        return self.run(x, aa_idx)

    def run(self, x, aa_idx):
        """(indices already validated)"""
        emb, pe, qe = _f(self.to_embedding.weight), self.pos_enc.pos_enc, _f(self.query_enc.weight)
        pd = self.pos_enc.p_dropout if self.training else 0.0
        if pd and pd > 0:
            # rf.py:114-120: dropout(emb[msa] + pe[aa_idx]) + query_enc -- the dropout sits between the two additions
            y = dropout_(ops.msa_embed(x.contiguous(), aa_idx.contiguous(), emb, pe, ops.zeros(*qe.shape, device=qe.device, dtype=F32)), pd)
            yq = ops.msa_embed(x.contiguous(), aa_idx.contiguous(), ops.zeros(*emb.shape, device=emb.device, dtype=F32),
                               ops.zeros(*pe.shape, device=pe.device, dtype=F32), qe)
            return ops.axpby(y, 1.0, yq, 1.0, y)
        return ops.msa_embed(x.contiguous(), aa_idx.contiguous(), emb, pe, qe)


class PairEmbedding(RFModule):
    """rf.py:123-181.  The (d_pair+1 [+d_template]) -> d_pair Linear is split by input block: the two sequence-embedding
    blocks fold into two 21-row tables, the separation column into one vector (rf_pair_embed), and with use_template
    the LayerNorm'ed template block (rf.py:141-143,161-169) is one rf_gemm accumulated onto that result."""

    def __init__(self, d_input=21, d_pair=288, max_len=260, p_pe_drop=0.1, use_template=False, d_template=64):
        super().__init__()
        self.half_d_pair = d_pair // 2
        self.embed_seq = nn.Embedding(d_input, self.half_d_pair)
        self.pos_enc = SinusoidalPositionalEncoding2D(d_pair, max_len, p_pe_drop)
        self.use_template = use_template
        if use_template:
            self.ln_template = LayerNorm(d_template)
            self.proj = Linear(d_pair + d_template + 1, d_pair)
        else:
            self.proj = Linear(d_pair + 1, d_pair)

    def forward(self, seq, aa_idx, template=None):
        if not self.use_template and template is not None:
            raise ValueError(f"[{self.__class__.__name__}]: template is not None but use_template is False")
        if self.use_template and template is None:
            # the reference fails inside nn.LayerNorm(None) (rf.py:166); same exception type, clearer text
            raise TypeError(f"[{self.__class__.__name__}]: use_template is True but no template was given")
        check_index_range(None, seq.contiguous(), aa_idx.contiguous(), self.embed_seq.num_embeddings, self.pos_enc.max_len)
        return self.run(seq, aa_idx, template)

    def run(self, seq, aa_idx, template=None):
        """(indices already validated)"""
        h = self.half_d_pair

        def tables():
            e = self.embed_seq.weight.detach().float().contiguous()
            w = self.proj.weight.detach().float()
            tl = ops.linear(e, w[:, :h].contiguous(), None)
            tr = ops.linear(e, w[:, h:2 * h].contiguous(), None)
            return tl, tr, w[:, 2 * h].contiguous()

        tl, tr, wsep = self.cached("tables", tables)
        x = ops.pair_embed(seq.contiguous(), aa_idx.contiguous(), tl, tr, wsep, _f(self.proj.bias), self.pos_enc.pos_enc)
        if self.use_template:
            B, Lr = seq.shape
            if tuple(template.shape[:3]) != (B, Lr, Lr) or template.shape[-1] != self.ln_template.weight.shape[0]:
                raise ValueError(f"[{self.__class__.__name__}]: template must be [B, L, L, {self.ln_template.weight.shape[0]}]")
            wt = self.cached("wtempl", lambda: self.proj.weight.detach()[:, 2 * h + 1:].to(T()).contiguous())
            tn = ln(self.ln_template, template.float().contiguous())
            ops.linear(tn, wt, None, out=x, residual=x)  # x += LayerNorm(template) @ W_template^T (fp32, in place)
        return x


# ================================================================================================
# MSA row (tied) attention
# ================================================================================================
class PositionWiseWeightFactor(RFModule):
    """rf.py:184-217."""

    def __init__(self, d_msa=384, n_heads=12, p_dropout=0.1):
        super().__init__()
        assert d_msa % n_heads == 0, \
            f"[{self.__class__.__name__}]: d_msa ({d_msa}) must be divisible by n_heads ({n_heads})."
        self.n_heads = n_heads
        self.d_head = d_msa // n_heads
        self.scale = self.d_head ** (-0.5)
        self.p_dropout = p_dropout
        self.to_q = nn.Sequential(Linear(d_msa, d_msa), nn.Identity())
        self.to_k = nn.Sequential(Linear(d_msa, d_msa), nn.Identity())

    def drop(self, w):
        """rf.py:217: dropout on the softmax weights (training mode; they then no longer sum to one, as in the reference)."""
        return dropout_(w, self.p_dropout) if self.training else w

    def query_proj(self, xn):
        """to_q on MSA row 0 only: [B*L, d] (T)."""
        B, N, Lr, D = xn.shape
        q0 = torch.empty(B * Lr, D, device=xn.device, dtype=T())
        ops.gemm(xn, self.wt("q", self.to_q[0]), q0, B * Lr, D, D, a_row=(Lr, N * Lr * D, D),
                 bias=_f(self.to_q[0].bias))
        return q0

    def weights(self, xn, w_out=None):
        """xn: T [B,N,L,d] -> w fp32 [B,N,H,L]."""
        B, N, Lr, D = xn.shape
        q0 = self.query_proj(xn)
        k = ops.linear(xn, self.wt("k", self.to_k[0]), _f(self.to_k[0].bias))
        w = torch.empty(B, N, self.n_heads, Lr, device=xn.device, dtype=F32) if w_out is None else w_out
        ops.poswise(q0, D, k, D, 0, self.d_head, self.d_head, w, None, 0, 0, 0, B, N, Lr, self.n_heads, self.scale, 1.0)
        return self.drop(w)

    def weights_collapsed(self, msa, lnm, m):
        """1-head weights for the structure track, q side in fp32: w = softmax_n(scale * m[b,n,l,:] . u[b,l,:]) with
        u = to_q(LN(msa[:,0])) W_k  (to_k's bias is constant over n and drops out of the softmax).
        msa fp32 [B,N,L,D]; lnm = the LayerNorm applied to it; m = lnm(msa) in T."""
        assert self.n_heads == 1
        B, N, Lr, D = msa.shape
        row0 = ops.layernorm(msa[:, 0].contiguous(), _f(lnm.weight), _f(lnm.bias), eps=lnm.eps, out_dtype=F32)
        q0 = ops.linear(row0, self.to_q[0].weight.detach().float(), _f(self.to_q[0].bias), out_dtype=F32)
        wkt = self.cached("wkt32", lambda: self.to_k[0].weight.detach().float().t().contiguous())
        u = ops.linear(q0, wkt, None, out_dtype=F32)
        w = torch.empty(B, N, 1, Lr, device=msa.device, dtype=F32)
        ops.poswise(u, D, m, D, 0, 0, D, w, None, 0, 0, 0, B, N, Lr, 1, self.scale, 1.0)
        return self.drop(w)

    def forward(self, msa_emb):
        w = self.weights(ops.cast(msa_emb.contiguous(), T()))
        return w.unsqueeze(-1)  # b N h l 1


class SoftTiedAttentionOverResidues(RFModule):
    """rf.py:220-267."""

    def __init__(self, d_msa=384, n_heads=12, p_dropout=0.1, return_att=False):
        super().__init__()
        assert d_msa % n_heads == 0, \
            f"[{self.__class__.__name__}]: d_msa ({d_msa}) must be divisible by n_heads ({n_heads})."
        self.n_heads, self.d_head = n_heads, d_msa // n_heads
        self.scale = self.d_head ** (-0.5)
        self.return_att = return_att
        self.p_dropout = p_dropout
        self.poswise_weight = PositionWiseWeightFactor(d_msa, n_heads, p_dropout)
        self.to_q = Linear(d_msa, d_msa)
        self.to_k = Linear(d_msa, d_msa)
        self.to_v = Linear(d_msa, d_msa)
        self.to_out = Linear(d_msa, d_msa)

    def attend(self, xn, x_res, want_att, next_ln=None, drops=()):
        """xn: T [B,N,L,D] (already layer-normed); x_res: fp32 [B,N,L,D] += to_out(attention).  Returns
        (symmetrised attention map fp32 [B,L,L,H] when want_att, next_ln(x_res) when fused else None).
        Training mode: the position weights are dropped out inside (rf.py:217); `drops` = the dropouts on the projected output
        before the residual add (this module's own, rf.py:265-267, and the wrapping EncoderLayer's, rf.py:346)."""
        B, N, Lr, D = xn.shape
        H, dh = self.n_heads, self.d_head
        dev = xn.device
        pw = self.poswise_weight
        if (RT.fused_tied and RT.tied_v2 and ops.is_h16(T()) and dh == 32 and Lr in (64, 128, 192, 256) and H <= 16
                and N % 16 == 0 and N // 16 in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16) and (B * N * Lr) % 256 == 0 and B * N * Lr >= 16384
                and (6 if Lr >= 256 else 8) * (4096 + Lr * 64) + 1024 + N * 256 <= 160 * 1024):
            return self.attend_head_major(xn, x_res, want_att, next_ln, drops)
        if (RT.fused_tied and RT.tied_v2 and RT.tied_fold_w and ops.is_h16(T()) and dh == 32 and Lr in (512, 768, 1024) and H <= 16
                and ops.gemm_takes_row_scale(B * N * Lr, 2 * D, D) and N % 16 == 0 and N // 16 in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16)
                and (B * N * Lr) % 256 == 0 and B * N * Lr >= 16384):
            return self.attend_long_rows(xn, x_res, want_att, next_ln, drops)
        # one GEMM for q | k | poswise-k  (N = 3D)
        wcat = self.wcat("qkp", [self.to_q, self.to_k, pw.to_k[0]])
        bcat = self.bcat("qkp", [self.to_q, self.to_k, pw.to_k[0]])
        qkp = ops.linear(xn, wcat, bcat)  # [B,N,L,3D]
        q0 = pw.query_proj(xn)
        # w = softmax_n(q0.k_pw * scale);  q <- q * w * scale   (rf.py:252)
        if pw.training and pw.p_dropout and pw.p_dropout > 0:
            # training mode: the weights exist as a tensor so that the dropout can sit between the softmax and the scaling
            w = torch.empty(B, N, H, Lr, device=dev, dtype=F32)
            ops.poswise(q0, D, qkp, 3 * D, 2 * D, dh, dh, w, None, 0, 0, 0, B, N, Lr, H, pw.scale, 1.0)
            pw.drop(w)
            wl = ops.copy4d(w, (N * H * Lr, H * Lr, 1, Lr), torch.empty(B, N, Lr, H, device=dev, dtype=F32),
                            (N * Lr * H, Lr * H, H, 1), (B, N, Lr, H))
            ops.axpby(wl, self.scale, None, 0.0, wl)
            qc = ops.copy4d(qkp, (N * Lr * 3 * D, Lr * 3 * D, 3 * D, 1), torch.empty(B, N, Lr, D, device=dev, dtype=T()),
                            (N * Lr * D, Lr * D, D, 1), (B, N, Lr, D))
            ops.scale_rows(qc, wl, B * N * Lr * H, dh, out=qc)
            ops.copy4d(qc, (N * Lr * D, Lr * D, D, 1), qkp, (N * Lr * 3 * D, Lr * 3 * D, 3 * D, 1), (B, N, Lr, D))
        else:
            ops.poswise(q0, D, qkp, 3 * D, 2 * D, dh, dh, None, qkp, 3 * D, 0, dh, B, N, Lr, H, pw.scale, self.scale)
        # v transposed: v_t[b,n,(h,d),l]
        bv, bo = _f(self.to_v.bias), _f(self.to_out.bias)
        if RT.condition and RT.condition_values and ops.is_h16(T()):
            bv, bo = (t_[0] for t_ in self.value_conditioning(xn, self.to_v, self.to_out))
        v_t = torch.empty(B, N, D, Lr, device=dev, dtype=T())
        ops.gemm(self.wt("v", self.to_v), xn, v_t, D, Lr, D, batch=(B * N, 1, 1), b_bs=(Lr * D, 0, 0),
                 c_bs=(D * Lr, 0, 0), c_row=(0, 0, Lr), bias=bv, bias_mode=L.BIAS_ROW)
        # logits[b,h,i,j] = sum_{n,d} q k   (contraction over N*dh, rf.py:254), softmax over j (rf.py:255)
        W3 = 3 * D
        att = torch.empty(B, H, Lr, Lr, device=dev, dtype=T())
        att_sym = torch.empty(B, Lr, Lr, H, device=dev, dtype=F32) if want_att else None
        if RT.fused_tied and ops.is_h16(T()) and dh == 32 and Lr in (64, 128, 192, 256):
            # one launch: 6-8-stage DMA ring over the N steps, logits in registers, wave-local softmax (csrc/tied.hip)
            ops.tied_logits_softmax(qkp, qkp[..., D:], N * Lr * W3, Lr * W3, W3, att, att_sym, B, H, N, Lr, dh)
        else:
            logits = torch.empty(B, H, Lr, Lr, device=dev, dtype=F32)
            ops.gemm(qkp, qkp, logits, Lr, Lr, N * dh, batch=(B, H, 1), b_off=D,
                     a_bs=(N * Lr * W3, dh, 0), a_row=(0, 0, W3), a_ko=Lr * W3,
                     b_bs=(N * Lr * W3, dh, 0), b_row=(0, 0, W3), b_ko=Lr * W3, kc=dh,
                     c_bs=(H * Lr * Lr, Lr * Lr, 0), c_row=(0, 0, Lr))
            ops.tied_softmax(logits, att, att_sym, H)
        # out[b,n,i,(h,d)] = sum_j att[b,h,i,j] v[b,n,h,j,d]   (rf.py:257-258)
        out = torch.empty(B, N, Lr, D, device=dev, dtype=T())
        ops.gemm(att, v_t, out, Lr, N * dh, Lr, batch=(B, H, 1),
                 a_bs=(H * Lr * Lr, Lr * Lr, 0), a_row=(0, 0, Lr),
                 b_bs=(N * D * Lr, dh * Lr, 0), b_row=(dh, D * Lr, Lr),
                 c_bs=(N * Lr * D, dh, 0), c_row=(0, 0, D), c_col=(dh, Lr * D))
        xn_next = project_into_residual(out, self.wt("o", self.to_out), bo, x_res, next_ln, drops)
        return att_sym, xn_next

    def attend_head_major(self, xn, x_res, want_att, next_ln=None, drops=()):
        """The bench path (csrc/tied.hip): one projection GEMM writes q|k|v head-major [B,N,3H,L,32] (every contraction
        step of the attention kernels is then one contiguous tile), the position weights come from the collapsed form
        (no to_k projection over the N rows) and are applied inside the logits kernel, attention.V consumes v with
        transposed LDS reads: no v^T GEMM, no pass over q, no fp32 logits."""
        B, N, Lr, D = xn.shape
        H, dh = self.n_heads, self.d_head
        dev = xn.device
        pw = self.poswise_weight
        G = 3 * H
        lins = [self.to_q, self.to_k, self.to_v]
        # u[b,l,h,:] = W_k[h*dh:(h+1)*dh, :]^T to_q(x_0)[b,l,h*dh:(h+1)*dh]   (rf.py:205-217, collapsed)
        q0 = pw.query_proj(xn)
        wkt = pw.cached("wkT", lambda: pw.to_k[0].weight.detach().t().contiguous().to(T()))
        u = torch.empty(B, Lr, H, D, device=dev, dtype=T())
        ops.gemm(q0, wkt, u, B * Lr, D, dh, batch=(H, 1, 1), a_bs=(dh, 0, 0), a_row=(0, 0, D), b_bs=(dh, 0, 0),
                 b_row=(0, 0, D), c_bs=(D, 0, 0), c_row=(0, 0, H * D))
        w = pw.drop(ops.poswise_collapsed(xn, u, pw.scale))  # fp32 [B,H,N,L]
        qkv = torch.empty(B, N, G, Lr, dh, device=dev, dtype=T())
        # q * w * d_head^-0.5 (rf.py:252) in the projection's epilogue, on the fp32 accumulators: q is rounded once, after
        # the scaling, and the logits kernel neither stages the weights nor rescales its fragments (round 2: 15-20 us of VALU)
        # ... when the projection runs on the kernel whose epilogue knows the row-group scale (d_msa = 288: N = 864 does not)
        fold = RT.tied_fold_w and dh % 16 == 0 and ops.gemm_takes_row_scale(B * N * Lr, 3 * D, D)
        bqkv, bo = self.bcat("qkv", lins), _f(self.to_out.bias)
        if RT.condition and RT.condition_values:
            bqkv, bo = (t_[0] for t_ in self.value_conditioning(xn, self.to_v, self.to_out, lead=(self.to_q, self.to_k)))
        ops.gemm(xn, self.wcat("qkv", lins), qkv, B * N * Lr, 3 * D, D, bias=bqkv,
                 c_row=(Lr, G * Lr * dh, dh), c_col=(dh, Lr * dh),
                 rs=(w, H * N * Lr, N * Lr, dh, D, self.scale) if fold else None)
        att = torch.empty(B, H, Lr, Lr, device=dev, dtype=T())
        att_sym = torch.empty(B, Lr, Lr, H, device=dev, dtype=F32) if want_att else None
        out = torch.empty(B, N, Lr, D, device=dev, dtype=T())
        ops.tied_attention(qkv[:, :, 0:H], qkv[:, :, H:2 * H], qkv[:, :, 2 * H:], out.view(B, N, Lr, H, dh).permute(0, 1, 3, 2, 4),
                           att, w=None if fold else w, qscale=1.0 if fold else self.scale, att_sym=att_sym)
        xn_next = project_into_residual(out, self.wt("o", self.to_out), bo, x_res, next_ln, drops)
        return att_sym, xn_next

    def attend_long_rows(self, xn, x_res, want_att, next_ln=None, drops=()):
        """L in {512, 768, 1024} (BASELINE.json configs[3]).  Same front as attend_head_major -- collapsed position weights, one
        projection GEMM writing q|k head-major with w * d_head^-0.5 folded into q on the fp32 accumulators -- then the
        contraction-split logits kernel over 128-query x 256-key tiles (csrc/tied.hip: rf_tied_logits).  attention . V at
        these lengths is a plain large GEMM per (b, head) (M = L queries, K = L keys, N = n_seq * 32): it goes to rf_gemm with
        v written key-contiguous by its projection (keeping whole probability rows in registers, as the L <= 256 kernel
        does, would need 128 VGPRs per 16 query rows)."""
        B, N, Lr, D = xn.shape
        H, dh = self.n_heads, self.d_head
        dev = xn.device
        pw = self.poswise_weight
        q0 = pw.query_proj(xn)
        wkt = pw.cached("wkT", lambda: pw.to_k[0].weight.detach().t().contiguous().to(T()))
        u = torch.empty(B, Lr, H, D, device=dev, dtype=T())
        ops.gemm(q0, wkt, u, B * Lr, D, dh, batch=(H, 1, 1), a_bs=(dh, 0, 0), a_row=(0, 0, D), b_bs=(dh, 0, 0),
                 b_row=(0, 0, D), c_bs=(D, 0, 0), c_row=(0, 0, H * D))
        w = pw.drop(ops.poswise_collapsed(xn, u, pw.scale))  # fp32 [B,H,N,L]
        G = 2 * H
        lins = [self.to_q, self.to_k]
        qk = torch.empty(B, N, G, Lr, dh, device=dev, dtype=T())
        ops.gemm(xn, self.wcat("qk", lins), qk, B * N * Lr, 2 * D, D, bias=self.bcat("qk", lins),
                 c_row=(Lr, G * Lr * dh, dh), c_col=(dh, Lr * dh), rs=(w, H * N * Lr, N * Lr, dh, D, self.scale))
        att = torch.empty(B, H, Lr, Lr, device=dev, dtype=T())
        att_sym = torch.empty(B, Lr, Lr, H, device=dev, dtype=F32) if want_att else None
        ops.tied_logits(qk[:, :, 0:H], qk[:, :, H:], att, att_sym)
        # v transposed: v_t[b,n,(h,d),l];  out[b,n,i,(h,d)] = sum_j att[b,h,i,j] v[b,n,h,j,d]   (rf.py:257-258)
        bv, bo = _f(self.to_v.bias), _f(self.to_out.bias)
        if RT.condition and RT.condition_values:
            bv, bo = (t_[0] for t_ in self.value_conditioning(xn, self.to_v, self.to_out))
        v_t = torch.empty(B, N, D, Lr, device=dev, dtype=T())
        ops.gemm(self.wt("v", self.to_v), xn, v_t, D, Lr, D, batch=(B * N, 1, 1), b_bs=(Lr * D, 0, 0),
                 c_bs=(D * Lr, 0, 0), c_row=(0, 0, Lr), bias=bv, bias_mode=L.BIAS_ROW)
        out = torch.empty(B, N, Lr, D, device=dev, dtype=T())
        ops.gemm(att, v_t, out, Lr, N * dh, Lr, batch=(B, H, 1),
                 a_bs=(H * Lr * Lr, Lr * Lr, 0), a_row=(0, 0, Lr),
                 b_bs=(N * D * Lr, dh * Lr, 0), b_row=(dh, D * Lr, Lr),
                 c_bs=(N * Lr * D, dh, 0), c_row=(0, 0, D), c_col=(dh, Lr * D))
        xn_next = project_into_residual(out, self.wt("o", self.to_out), bo, x_res, next_ln, drops)
        return att_sym, xn_next

    def forward(self, x):
        xn = ops.cast(x.contiguous(), T())
        out = ops.zeros(*x.shape, device=x.device, dtype=F32)
        att, _ = self.attend(xn, out, self.return_att, drops=(self.p_dropout,) if self.training else ())
        return (out, att) if self.return_att else out


# ================================================================================================
# Performer (FAVOR+) self attention -- restated third-party module (parity unpinned)
# ================================================================================================
class _FastAttention(nn.Module):
    def __init__(self, dim_heads, nb_features):
        super().__init__()
        self.register_buffer("projection_matrix", gaussian_orthogonal_random_matrix(nb_features, dim_heads))


def gaussian_orthogonal_random_matrix(nb_rows, nb_cols, generator=None):
    """performer-pytorch's projection construction (scaling=0); see oracle for the citation."""
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


class PerformerSelfAttention(RFModule):
    """performer_pytorch.SelfAttention as the reference instantiates it (rf.py:313-318, 505-518):
    dim_head=64, nb_features=266, no qkv bias, output bias; softmax-kernel or generalized ReLU features."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, generalized_attention=False, **kw):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        self.p_dropout = dropout   # (performer_pytorch.SelfAttention: nn.Dropout on the projected output)
        self.condition_v = False   # value conditioning of the 16-bit modes (RFModule.value_conditioning): the MSA track switches it on
        self.generalized = generalized_attention
        self.fast_attention = _FastAttention(dim_head, int(dim_head * math.log(dim_head)))
        self.to_q = Linear(dim, inner, bias=False)
        self.to_k = Linear(dim, inner, bias=False)
        self.to_v = Linear(dim, inner, bias=False)
        self.to_out = Linear(inner, dim)

    def proj_scaled(self, log2e=False):
        def make():
            p = self.fast_attention.projection_matrix.detach().float() * self.dim_head ** -0.25
            if log2e:  # the fused softmax-kernel variant exponentiates with exp2
                p = p * 1.4426950408889634
            pp = p.new_zeros(M_PAD, self.dim_head)
            pp[: p.shape[0]] = p
            return pp.to(T()).contiguous()
        return self.cached(("proj", log2e), make)

    def attend(self, xn, x_res, axis, next_ln=None, seq_group=None, drops=()):
        """xn: T [B,L1,L2,D] layer-normed input; sequences run along `axis` (1 or 2); x_res (fp32, same shape)
        += to_out(linear attention).  All intermediates are addressed by strides: no transposes.
        Returns next_ln(x_res) when the fused residual+LayerNorm epilogue applies, else None.
        seq_group: a torch.distributed group over which the SEQUENCE axis is sharded (every rank holds a slice of every
        sequence: pair-track row blocks).  Linear attention communicates contexts, not maps: the local
        k'^T [v | 1] sums are all-reduced (fp32) before the queries are applied (rf.py:505-518 semantics unchanged)."""
        B, L1, L2, D = xn.shape
        H, dh, inner = self.heads, self.dim_head, self.inner
        dev = xn.device
        Ls, Lo = (L1, L2) if axis == 1 else (L2, L1)
        ss, so = (L2, 1) if axis == 1 else (1, L2)  # row strides of the sequence / outer index
        RB = L1 * L2
        R = B * RB
        W2 = 2 * inner
        S = B * Lo * H
        m = self.fast_attention.projection_matrix.shape[0]
        pc = self.proj_scaled()
        gen = self.generalized
        if seq_group is not None and not gen:
            raise NotImplementedError("sequence-sharded attention is built for the generalized (ReLU) feature map of the pair "
                                      "track only: the softmax feature map needs the global key maximum first")
        cond = RT.condition and RT.condition_values and self.condition_v and ops.is_h16(T()) and seq_group is None
        bqkv, bv, bo = None, None, _f(self.to_out.bias)
        if cond:
            bqkv, bo = (t_[0] for t_ in self.value_conditioning(xn, self.to_v, self.to_out, lead=(self.to_q, self.to_k)))
            bv = bqkv[2 * inner:]
        if seq_group is None and RT.fused_favor and ops.is_h16(T()) and dh == 64 and m == M_FEAT and (
                Ls in (64, 128, 256) or (gen and Ls > 256 and Ls % 256 == 0)):
            # fused path: one projection GEMM (q|k|v) + one persistent kernel; q', k', ctx never leave the chip
            W3 = 3 * inner
            o = torch.empty(R, inner, device=dev, dtype=T())
            pcf = pc if gen else self.proj_scaled(log2e=True)
            eps = 1e-3 if gen else 1e-4
            wqkv = self.wcat("qkv", [self.to_q, self.to_k, self.to_v])
            if RT.head_major_qkv == 1 or (RT.head_major_qkv == 2 and axis == 2):
                # q|k|v written head-major [B, Lo, 3, H, Ls, 64] straight from the projection GEMM's epilogue: every
                # (b, o, head) tile the FAVOR kernel DMAs is then one contiguous 8 KB x (Ls/64) block
                qkv = torch.empty(B, Lo, 3, H, Ls, dh, device=dev, dtype=T())
                so_c = 3 * H * Ls * dh
                if axis == 2:
                    # row m = (b, p1, p2) with the sequence along p2: offset (m / L2) * so_c + (m % L2) * dh -- one launch, the
                    # register-resident-weight GEMM's split-C epilogue (csrc/gemm_wreg.hip)
                    ops.gemm(xn, wqkv, qkv, R, W3, D, c_row=(L2, so_c, dh), c_col=(dh, Ls * dh), bias=bqkv)
                else:
                    # sequence along p1: offset b * Lo * so_c + p2 * so_c + p1 * dh is a three-level split: one launch per
                    # batch element (row m = p1 * L2 + p2 inside it: (m / L2) * dh + (m % L2) * so_c)
                    for b in range(B):
                        ops.gemm(xn[b], wqkv, qkv[b], RB, W3, D, c_row=(L2, dh, so_c), c_col=(dh, Ls * dh), bias=bqkv)
                ops.favor_attention(qkv, pcf, o, (Lo * so_c, so_c, dh, Ls * dh), (RB * inner, so * inner, ss * inner),
                                    0, H * Ls * dh, 2 * H * Ls * dh, B, Lo, H, Ls, dh, m, not gen, eps)
            else:
                # whole batch elements per panel (RB rows each: any axis stays addressable inside one element)
                pr = row_panels(R, W3 * 2, RB)
                if pr < R and not drops:
                    nb = pr // RB
                    qkv = torch.empty(pr, W3, device=dev, dtype=T())
                    xn2, xr2 = xn.view(R, D), x_res.view(R, -1)
                    wo = self.wt("o", self.to_out)
                    nxt, fused = torch.empty(R, xr2.shape[1], device=dev, dtype=T()) if next_ln is not None else None, True
                    for r0 in range(0, R, pr):
                        ops.linear(xn2[r0:r0 + pr], wqkv, bqkv, out=qkv)
                        op = o[r0:r0 + pr]
                        ops.favor_attention(qkv, pcf, op, (RB * W3, so * W3, ss * W3, dh), (RB * inner, so * inner, ss * inner),
                                            0, inner, 2 * inner, nb, Lo, H, Ls, dh, m, not gen, eps)
                        fused &= ops.linear_residual_ln(op, wo, bo, xr2[r0:r0 + pr], next_ln, nxt[r0:r0 + pr] if nxt is not None else None) is not None
                    return nxt.view(x_res.shape) if nxt is not None and fused else None
                qkv = ops.linear(xn, wqkv, bqkv)
                ops.favor_attention(qkv, pcf, o, (RB * W3, so * W3, ss * W3, dh), (RB * inner, so * inner, ss * inner),
                                    0, inner, 2 * inner, B, Lo, H, Ls, dh, m, not gen, eps)
            return project_into_residual(o, self.wt("o", self.to_out), bo, x_res, next_ln, drops)
        qk = ops.linear(xn, self.wcat("qk", [self.to_q, self.to_k]), None)  # [R, 2*inner]
        # q' [B,Lo,H,Ls,M_PAD]
        dq = torch.empty(B, Lo, H, Ls, M_PAD, device=dev, dtype=T())
        ops.gemm(qk, pc, dq, Ls * H, M_PAD, dh, batch=(B, Lo, 1),
                 a_bs=(RB * W2, so * W2, 0), a_row=(H, ss * W2, dh),
                 c_bs=(Lo * H * Ls * M_PAD, H * Ls * M_PAD, 0), c_row=(H, M_PAD, Ls * M_PAD),
                 act=L.ACT_RELU_EPS if gen else L.ACT_NONE, act_nvalid=m, act_eps=1e-3)
        # k'^T [B,Lo,H,M_PAD,Ls]
        kt = torch.empty(B, Lo, H, M_PAD, Ls, device=dev, dtype=T())
        ops.gemm(pc, qk, kt, M_PAD, Ls, dh, batch=(B, Lo, H), b_off=inner,
                 b_bs=(RB * W2, so * W2, dh), b_row=(0, 0, ss * W2),
                 c_bs=(Lo * H * M_PAD * Ls, H * M_PAD * Ls, M_PAD * Ls), c_row=(0, 0, Ls),
                 act=L.ACT_RELU_EPS if gen else L.ACT_NONE, act_nvalid=-m, act_eps=1e-3)
        if not gen:
            xs = (RB * W2, so * W2, dh, ss * W2)
            ops.favor_softmax_features(dq, qk, 0, xs, Lo, H, S, Ls, m, M_PAD, dh, 1, 0)
            ops.favor_softmax_features(kt, qk, inner, xs, Lo, H, S, Ls, m, M_PAD, dh, 0, 1)
        # v^T [B,Lo,H,80,Ls] with a ones row at index 64 (-> k' column sums ride along in the context GEMM)
        vt = ops.zeros(B, Lo, H, VT_ROWS, Ls, device=dev, dtype=T())
        ones_row = ops.fill(torch.empty(B * Lo * H, Ls, device=dev, dtype=T()), 1.0)
        ops.copy4d(ones_row, (0, 0, Ls, 1), vt, (0, 0, VT_ROWS * Ls, 1), (1, 1, B * Lo * H, Ls), y_off=dh * Ls)
        ops.gemm(self.wt("v", self.to_v), xn, vt, inner, Ls, D, batch=(B, Lo, 1),
                 b_bs=(RB * D, so * D, 0), b_row=(0, 0, ss * D),
                 c_bs=(Lo * H * VT_ROWS * Ls, H * VT_ROWS * Ls, 0), c_row=(dh, VT_ROWS * Ls, Ls),
                 bias=bv, bias_mode=L.BIAS_ROW if bv is not None else None)
        # context^T [S,80,M_PAD] = v^T k'
        # fp16 operands (range 65504): the context is a sum over the whole sequence, so it is stored scaled by 2^-ceil(log2 Ls_total)
        # (on the fp32 accumulators, before the rounding) exactly as csrc/favor.hip does in the fused kernel; numerator and
        # denominator of the final ratio carry the same power of two, the result is unchanged
        ctx_scale = 1.0
        if T() == torch.float16:
            from . import shard as _shard
            ctx_scale = 2.0 ** -math.ceil(math.log2(max(Ls * (_shard.group_size(seq_group) if seq_group is not None else 1), 1)))
        ctx = torch.empty(S, VT_ROWS, M_PAD, device=dev, dtype=T() if seq_group is None else F32)
        ops.gemm(vt, kt, ctx, VT_ROWS, M_PAD, Ls, batch=(S, 1, 1), a_bs=(VT_ROWS * Ls, 0, 0),
                 b_bs=(M_PAD * Ls, 0, 0), c_bs=(VT_ROWS * M_PAD, 0, 0), alpha=ctx_scale if seq_group is None else 1.0)
        if seq_group is not None:
            # the one exchange of a sequence-sharded layer: [B * Lo * H, 80, 288] fp32 partial contexts (+ k' sums in row 64).
            # The partial sums travel unscaled (the ranks' slices may differ in length); every rank scales its own copy of the
            # full context when it rounds it to the 16-bit type (any power of two cancels in that rank's ratio)
            from . import shard
            shard.all_reduce_sum(ctx, seq_group)
            ctx = ops.axpby(ctx, ctx_scale, None, 0.0, torch.empty(ctx.shape, device=dev, dtype=T()))
        # numerator | denominator: num[b,p1,p2,h,0:64 | 64]
        num = torch.empty(R * H, VT_ROWS, device=dev, dtype=F32)
        ops.gemm(dq, ctx, num, Ls, VT_ROWS, M_PAD, batch=(B, Lo, H),
                 a_bs=(Lo * H * Ls * M_PAD, H * Ls * M_PAD, Ls * M_PAD), a_row=(0, 0, M_PAD),
                 b_bs=(Lo * H * VT_ROWS * M_PAD, H * VT_ROWS * M_PAD, VT_ROWS * M_PAD),
                 c_bs=(RB * H * VT_ROWS, so * H * VT_ROWS, VT_ROWS), c_row=(0, 0, ss * H * VT_ROWS))
        o = torch.empty(R, inner, device=dev, dtype=T())
        ops.linattn_normalize(num, VT_ROWS, o, dh, R * H, dh)
        return project_into_residual(o, self.wt("o", self.to_out), bo, x_res, next_ln, drops)

    def forward(self, x):
        """x [S, n, dim] -> [S, n, dim] (library call surface)."""
        S_, n, D = x.shape
        xn = ops.cast(x.contiguous(), T()).view(1, S_, n, D)
        out = ops.zeros(1, S_, n, D, device=x.device, dtype=F32)
        self.attend(xn, out, axis=2, drops=(self.p_dropout,) if self.training else ())
        return out.view(S_, n, D)


# ================================================================================================
# encoder layers / MSA self attention update
# ================================================================================================
class EncoderLayer(RFModule):
    """rf.py:284-354."""

    def __init__(self, d_msa=384, d_ff=384 * 4, n_heads=12, p_dropout=0.1, tied=False, performer=False,
                 performer_kws={}, return_att=False):
        super().__init__()
        self.tied, self.return_att = tied, return_att
        if tied:
            self.attn = SoftTiedAttentionOverResidues(d_msa=d_msa, n_heads=n_heads, p_dropout=p_dropout,
                                                      return_att=return_att)
        elif performer:
            if return_att:
                raise NotImplementedError("PerformerSelfAttention does not support return_att.")
            self.attn = PerformerSelfAttention(dim=d_msa, heads=n_heads, dropout=p_dropout, **performer_kws)
            self.attn.condition_v = True
        else:
            raise NotImplementedError
        self.ln = LayerNorm(d_msa)
        self.p_dropout = p_dropout   # rf.py:324,346: x = orig + dropout(attn(ln(x)))
        self.ff = Residual(nn.Sequential(LayerNorm(d_msa), FeedForward(d_msa, d_ff, p_dropout=p_dropout),
                                         nn.Dropout(p_dropout)))

    def run(self, x, seq_axis=2, want_att=False, xn=None, next_ln=None):
        """x: fp32 residual stream [B,n1,n2,D], updated in place.  tied: rows = n1 (MSA depth), attention over n2.
        performer: attention along `seq_axis`.  xn: self.ln(x) if the previous GEMM already produced it;
        next_ln: the LayerNorm that will consume x next.  Returns (att, next_ln(x) or None)."""
        if xn is None:
            xn = ln(self.ln, x)
        att = None
        # training mode: the attention module's own output dropout, then this layer's (rf.py:265-267 / performer, 346); the
        # feed-forward's output dropout (rf.py:330) rides into apply_residual
        ad = (self.attn.p_dropout, self.p_dropout) if self.training else ()
        fd = (_p(self.ff.fn[2]),) if self.training else ()
        if self.tied:
            att, xf = self.attn.attend(xn, x, want_att, next_ln=self.ff.fn[0], drops=ad)
        else:
            xf = self.attn.attend(xn, x, seq_axis, next_ln=self.ff.fn[0], drops=ad)
        if xf is None:
            xf = ln(self.ff.fn[0], x)
        return att, self.ff.fn[1].apply_residual(xf, x, next_ln, drops=fd)

    def forward(self, x):
        x = fresh_f32(x)
        if self.tied:
            att, _ = self.run(x, want_att=self.return_att)
            return (x, att) if self.return_att else x
        # reference flattens (b n) l d: attention along dim 2 of [b, n, l, d]
        self.run(x, seq_axis=2)
        return x


class MsaUpdateUsingSelfAttention(RFModule):
    """rf.py:357-409."""

    def __init__(self, d_msa=384, d_ff=384 * 4, n_heads=12, p_dropout=0.1, n_encoder_layers=4, performer_kws={}):
        super().__init__()
        self.residue_wise_encoder_layers = nn.ModuleList([
            EncoderLayer(d_msa=d_msa, d_ff=d_ff, n_heads=n_heads, p_dropout=p_dropout, tied=True, performer=False,
                         return_att=True) for _ in range(n_encoder_layers)])
        self.sequence_wise_encoder_layers = nn.ModuleList([
            EncoderLayer(d_msa=d_msa, d_ff=d_ff, n_heads=n_heads, p_dropout=p_dropout, tied=False, performer=True,
                         performer_kws=performer_kws) for _ in range(n_encoder_layers)])

    def run(self, x):
        att = None
        layers = list(self.residue_wise_encoder_layers) + list(self.sequence_wise_encoder_layers)
        n = len(self.residue_wise_encoder_layers)
        xn = None
        for i, layer in enumerate(layers):
            nxt = layers[i + 1].ln if i + 1 < len(layers) else None  # the next layer's pre-norm rides in this FF2's epilogue
            if i < n:
                a, xn = layer.run(x, want_att=(i == n - 1), xn=xn, next_ln=nxt)  # only the last map is consumed (rf.py:400-401)
                att = a if a is not None else att
            else:
                # the reference transposes to b l n d: sequences run over the MSA depth
                _, xn = layer.run(x, seq_axis=1, xn=xn, next_ln=nxt)
        return att

    def forward(self, x):
        x = fresh_f32(x)
        return x, self.run(x)


# ================================================================================================
# pair update with MSA (outer product + 2-conv ResNet)
# ================================================================================================
class OuterProductMean(RFModule):
    """rf.py:412-427."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.to_out = nn.Sequential(LayerNorm(in_features ** 2), Linear(in_features ** 2, out_features))

    def fused_ok(self, P, N, Lr):
        return (RT.fused_outer and ops.is_h16(T()) and P == 32 and N in (64, 128) and Lr % 16 == 0
                and self.to_out[1].weight.shape[0] == 288)

    def run_into(self, x_t, y_t, ln2, feat, feat_ld):
        """Fused form with the consumer's LayerNorm (PairUpdateWithMsa.ln_coevol_feat) in the epilogue: writes bf16
        feat[..., 0:288] directly; neither the 1024-wide tensor nor the fp32 result exists."""
        lnm, lin = self.to_out[0], self.to_out[1]
        wp, s_, c_ = self.cached("outer_fold", lambda: ops.outer_fold(lin.weight.detach(), lnm.weight.detach(),
                                                                        lnm.bias.detach(), lin.bias.detach()))
        return ops.outer_fused(x_t, y_t, wp, s_, c_, None, lnm.eps, ln2=(_f(ln2.weight), _f(ln2.bias), ln2.eps, feat, feat_ld))

    def run(self, x_t, y_t, N):
        """x_t, y_t: T [B, L, P, N] (MSA depth contiguous).  -> fp32 [B,L,L,out]"""
        B, Lr, P, _ = x_t.shape
        PP = P * P
        lnm = self.to_out[0]
        lin = self.to_out[1]
        if self.fused_ok(P, N, Lr):
            # one kernel: outer product over the MSA depth -> LayerNorm(1024) (folded algebraically) -> Linear; the 1024-wide
            # tensor never leaves the chip (csrc/outer.hip)
            wp, s_, c_ = self.cached("outer_fold", lambda: ops.outer_fold(lin.weight.detach(), lnm.weight.detach(),
                                                                            lnm.bias.detach(), lin.bias.detach()))
            out = torch.empty(B, Lr, Lr, lin.weight.shape[0], device=x_t.device, dtype=F32)
            return ops.outer_fused(x_t, y_t, wp, s_, c_, out, lnm.eps)
        co = torch.empty(B, Lr, Lr, PP, device=x_t.device, dtype=T())
        if ops.is_h16(T()) and P == 32 and (Lr * P) % 256 == 0 and N >= 64 and RT.fused_outer_ln:
            # LayerNorm(1024) of every pair's outer-product block inside the GEMM epilogue (fp32 statistics on the
            # accumulators): the separate pass over the 0.5 GB feature tensor disappears
            ops.gemm(x_t, y_t, co, Lr * P, Lr * P, N, batch=(B, 1, 1), a_bs=(Lr * P * N, 0, 0), b_bs=(Lr * P * N, 0, 0),
                     c_bs=(Lr * Lr * PP, 0, 0), c_row=(P, Lr * PP, P), c_col=(P, PP), act=L.ACT_BLOCK_LN32,
                     block_ln=(_f(lnm.weight), _f(lnm.bias), lnm.eps))
            cn = co
        else:
            ops.gemm(x_t, y_t, co, Lr * P, Lr * P, N, batch=(B, 1, 1), a_bs=(Lr * P * N, 0, 0), b_bs=(Lr * P * N, 0, 0),
                     c_bs=(Lr * Lr * PP, 0, 0), c_row=(P, Lr * PP, P), c_col=(P, PP))
            cn = ln(lnm, co)
        return ops.linear(cn, self.wt("w", self.to_out[1]), _f(self.to_out[1].bias), out_dtype=F32)

    def run_rows(self, x_rows_t, y_t, N):
        """x_rows_t: T [B, h, P, N] (a block of rows), y_t: T [B, L, P, N] -> fp32 [B, h, L, out]: the general path of run()
        on a rectangular block (pair-track row-block sharding)."""
        B, h, P, _ = x_rows_t.shape
        Lr = y_t.shape[1]
        PP = P * P
        lnm = self.to_out[0]
        out = torch.empty(B, h, Lr, self.to_out[1].weight.shape[0], device=x_rows_t.device, dtype=F32)
        # in slabs of rows: the P*P-wide intermediate of a slab stays under 2^28 elements (rf_gemm rejects the 512 x 1024 block of a
        # two-rank split at L = 1024 in one piece; the slab also bounds the 0.5 GB-per-256-rows intermediate)
        step = max(1, (1 << 28) // (Lr * PP * B))
        for i0 in range(0, h, step):
            hs = min(step, h - i0)
            co = torch.empty(B, hs, Lr, PP, device=x_rows_t.device, dtype=T())
            for b in range(B):
                ops.gemm(x_rows_t[b, i0:i0 + hs], y_t[b], co[b], hs * P, Lr * P, N, c_row=(P, Lr * PP, P), c_col=(P, PP))
            y = ops.linear(ln(lnm, co), self.wt("w", self.to_out[1]), _f(self.to_out[1].bias), out_dtype=F32)
            ops.copy4d(y, (hs * Lr * y.shape[-1], Lr * y.shape[-1], y.shape[-1], 1), out,
                       (h * Lr * y.shape[-1], Lr * y.shape[-1], y.shape[-1], 1), (B, hs, Lr, y.shape[-1]), y_off=i0 * Lr * y.shape[-1])
        return out

    def forward(self, x, y=None):
        y = x if y is None else y
        B, N, Lr, P = x.shape
        Np = pad8(N)
        mk = ops.zeros if Np != N else torch.empty  # only the K padding needs zeros
        xt = mk(B, Lr, P, Np, device=x.device, dtype=T())
        yt = mk(B, Lr, P, Np, device=x.device, dtype=T())
        for src, dst in ((x, xt), (y, yt)):
            ops.copy4d(src.contiguous(), (N * Lr * P, P, 1, Lr * P), dst, (Lr * P * Np, P * Np, Np, 1), (B, Lr, P, N))
        return self.run(xt, yt, Np)


class PairUpdateWithMsa(RFModule):
    """rf.py:430-498."""

    def __init__(self, d_msa, d_proj, d_pair, n_heads, p_dropout=0.1):
        super().__init__()
        self.d_proj, self.d_pair, self.n_heads = d_proj, d_pair, n_heads
        self.proj_msa = nn.Sequential(LayerNorm(d_msa), Linear(d_msa, d_proj), LayerNorm(d_proj))
        self.poswise_weight = PositionWiseWeightFactor(d_proj, 1, p_dropout)
        self.outer_product_mean = OuterProductMean(d_proj, d_pair)
        self.ln_coevol_feat = LayerNorm(d_pair)
        self.ln_pair = LayerNorm(d_pair)
        d_feat_full = d_pair * 2 + d_proj * 4 + n_heads
        self.d_feat = d_feat_full
        self.resnet = nn.Sequential(
            Linear(d_feat_full, d_pair),
            Residual(nn.Sequential(
                nn.Identity(),
                nn.Conv2d(d_pair, d_pair, kernel_size=3, padding="same", bias=False),
                nn.InstanceNorm2d(d_pair, affine=True, eps=1e-6),
                nn.ELU(),
                nn.Dropout(p_dropout),
                nn.Conv2d(d_pair, d_pair, kernel_size=3, padding="same", bias=False),
                nn.InstanceNorm2d(d_pair, affine=True, eps=1e-6),
                nn.Identity(),
            )),
            nn.ELU(),
        )

    def _msa_operands(self, msa):
        """Per-position features of the MSA: (msa1d fp32 [B,L,2P], x_t, y_t T [B,L,P,Np] = the transposed operands of the outer
        product, Np)."""
        B, N, Lr, D = msa.shape
        P = self.d_proj
        dev = msa.device
        Np = pad8(N)
        mp_pre = ops.linear(ln(self.proj_msa[0], msa), self.wt("p", self.proj_msa[1]), _f(self.proj_msa[1].bias),
                            out_dtype=F32)
        mp = ln(self.proj_msa[2], mp_pre)  # T [B,N,L,P]
        w = self.poswise_weight.weights(mp)  # fp32 [B,N,1,L]
        # 1-D features: sum over N and the query row
        msa1d = torch.empty(B, Lr, 2 * P, device=dev, dtype=F32)
        ones = ops.fill(torch.empty(B, N, 1, Lr, device=dev, dtype=F32), 1.0)
        ops.weighted_msa_sum(mp, ones, msa1d, 2 * P)
        ops.copy4d(mp, (N * Lr * P, 0, P, 1), msa1d, (Lr * 2 * P, 0, 2 * P, 1), (B, 1, Lr, P), y_off=P)
        # transposed operands of the outer product: x_t[b,i,u,n] = mp ; y_t = mp * w   (rf.py:472-473)
        mpw = ops.scale_rows(mp, w, B * N * Lr, P)
        mk = ops.zeros if Np != N else torch.empty  # only the K padding needs zeros
        xt = mk(B, Lr, P, Np, device=dev, dtype=T())
        yt = mk(B, Lr, P, Np, device=dev, dtype=T())
        for src, dst in ((mp, xt), (mpw, yt)):
            ops.copy4d(src, (N * Lr * P, P, 1, Lr * P), dst, (Lr * P * Np, P * Np, Np, 1), (B, Lr, P, N))
        return msa1d, xt, yt, Np

    # ---- operand conditioning of the 16-bit modes (csrc/condition.hip; tools/precision_probe.py --pum-sweep) ------------------
    # At random init the 1-D features and the projected feature tensor are a per-sample constant plus a part ~20x smaller that
    # varies over the picture, and the ResNet's InstanceNorms keep only the latter: rounded to 16 bits WITH the constant, the
    # tiled 1-D features (1.3e-2), the first convolution's input (9.7e-3) and its output (7.3e-3) were 1.8e-2 of the fp16 mode's
    # 2.0e-2 logits gap at the benchmarked depth.  Both constants are removed before the rounding and carried in fp32:
    #   W (f - m) + (b + W m) == W f + b                           (the 1-D features' mean through the projection's bias)
    #   conv3x3(x - c) + [border terms] == conv3x3(x) - sum_taps W_tap c   (a per-channel constant: the InstanceNorm drops it)
    def _center_1d(self, msa1d, bias):
        P, Dp = self.d_proj, self.d_pair
        w32 = self.cached("f_f32", lambda: self.resnet[0].weight.detach().float().contiguous())
        return ops.fold_mean(w32, ops.center_rows(msa1d), bias, k0=Dp, nseg=2, seg_stride=2 * P)

    def _conv_taps(self, conv, cx):
        Co, Ci = conv.weight.shape[:2]
        w32 = self.cached("c1_f32", lambda: conv.weight.detach().permute(0, 2, 3, 1).reshape(Co, 9 * Ci).float().contiguous())
        return ops.fold_mean(w32, cx, None, nseg=9, seg_stride=Ci, sum_seg=False)   # [B, 9, Co]

    def run_rows(self, msa, pair_rows, att, row_group):
        """run() for a block of pair rows (pair-track row-block sharding, shard.pair_update_with_msa_row_sharded): msa
        [B,N,L,D] and att [B,L,L,H] are replicated, pair_rows fp32 [B,h,L,Dp] are this rank's rows shard_range(L, world, rank).
        The outer product, the feature assembly and the projection are local to the rows; the two 3x3 convolutions fetch one
        halo row from each neighbour and the two InstanceNorms all-reduce their sums.  Returns the new rows (fp32)."""
        from . import shard
        B, N, Lr, D = msa.shape
        P, Dp = self.d_proj, self.d_pair
        dev = msa.device
        h = pair_rows.shape[1]
        r0, r1 = shard.shard_range(Lr, shard.group_size(row_group), shard.group_rank(row_group))
        if r1 - r0 != h or h == 0:
            raise ValueError(f"pair rows {h} do not match this rank's (non-empty) share {r1 - r0} of {Lr}")
        msa1d, xt, yt, Np = self._msa_operands(msa)
        Kf = pad8(self.d_feat)
        feat = ops.zeros(B, h, Lr, Kf, device=dev, dtype=T())
        # outer product of this block's rows with every column: the general GEMM form (the fused kernel walks square pictures)
        xr = ops.copy4d(xt, (Lr * P * Np, P * Np, Np, 1), torch.empty(B, h, P, Np, device=dev, dtype=T()),
                        (h * P * Np, P * Np, Np, 1), (B, h, P, Np), x_off=r0 * P * Np)
        coevol = self.outer_product_mean.run_rows(xr, yt, Np)  # fp32 [B,h,L,Dp]
        ln(self.ln_coevol_feat, coevol, out=feat, out_ld=Kf, out_off=0)
        cond = RT.condition and ops.is_h16(T()) and Dp % 4 == 0   # (rf_center_apply moves 16-byte chunks of a pixel)
        bias = _f(self.resnet[0].bias)
        if cond:   # (the mean over ALL positions: msa is replicated, every rank folds the same constant)
            bias_b = self._center_1d(msa1d, bias)
        # feat[b,i,j, Dp + c] = msa1d[b, r0 + i, c] ; feat[b,i,j, Dp + 2P + c] = msa1d[b, j, c]   (rf_tile_1d_feats on a block)
        fs = (h * Lr * Kf, Lr * Kf, Kf, 1)
        ops.copy4d(msa1d, (Lr * 2 * P, 2 * P, 0, 1), feat, fs, (B, h, Lr, 2 * P), x_off=r0 * 2 * P, y_off=Dp)
        ops.copy4d(msa1d, (Lr * 2 * P, 0, 2 * P, 1), feat, fs, (B, h, Lr, 2 * P), y_off=Dp + 2 * P)
        ln(self.ln_pair, pair_rows, out=feat, out_ld=Kf, out_off=Dp + 4 * P)
        H = att.shape[-1]
        ops.copy4d(att.contiguous(), (Lr * Lr * H, Lr * H, H, 1), feat, fs, (B, h, Lr, H), x_off=r0 * Lr * H, y_off=2 * Dp + 4 * P)
        blk = self.resnet[1].fn
        kw = {"row_group": row_group, "rows_global": Lr}
        if cond:
            x = torch.empty(B, h, Lr, Dp, device=dev, dtype=F32)
            for b in range(B):
                ops.linear(feat[b], self.wt("f", self.resnet[0], kpad=Kf), bias_b[b], out=x[b])
            cx = ops.channel_mean(x, **kw)
            taps = self._conv_taps(blk[1], cx)
            n_r, r_ = shard.group_size(row_group), shard.group_rank(row_group)
            edges = 12 | (1 if r_ == 0 else 0) | (2 if r_ == n_r - 1 else 0)   # left | right, top / bottom on the outer ranks
        else:
            x = ops.linear(feat, self.wt("f", self.resnet[0], kpad=Kf), bias, out_dtype=F32)
        if B == 1:   # one picture: pre-haloed buffers, only the halo rows move (see ResBlock2D._run_rows_b1)
            xh = shard.haloed_buffer(h, Lr, Dp, 1, dev, T())
            if cond:
                ops.center_apply(x, cx, out=shard.interior(xh, 1))
            else:
                ops.axpby(x, 1.0, None, 0.0, shard.interior(xh, 1))
            shard.exchange_row_halos_inplace(xh, 1, row_group)
            y = conv3x3(self, "c1", blk[1], xh, 1)
            if cond:
                ops.conv3x3_border_fix(shard.interior(y, 1), taps, 1, edges)
            yh = shard.haloed_buffer(h, Lr, Dp, 1, dev, T())
            ops.instnorm(shard.interior(y, 1), _f(blk[2].weight), _f(blk[2].bias), eps=blk[2].eps, act=L.ACT_ELU,
                         out=shard.interior(yh, 1), **kw)
            shard.exchange_row_halos_inplace(yh, 1, row_group)
            y = conv3x3(self, "c2", blk[5], yh, 1)
            out, _ = ops.instnorm(shard.interior(y, 1), _f(blk[6].weight), _f(blk[6].bias), eps=blk[6].eps, residual=x,
                                  act=L.ACT_ELU, out_dtype=F32, **kw)
            return out
        conv = lambda key, c, t: shard.drop_row_halos(conv3x3(self, key, c, shard.exchange_row_halos(t, 1, row_group), 1), 1)  # noqa: E731
        if cond:
            y = ops.conv3x3_border_fix(conv("c1", blk[1], ops.center_apply(x, cx, out_dtype=T())), taps, 1, edges)
        else:
            y = conv("c1", blk[1], ops.cast(x, T()))
        y, _ = ops.instnorm(y, _f(blk[2].weight), _f(blk[2].bias), eps=blk[2].eps, act=L.ACT_ELU, out_dtype=T(), **kw)
        y = conv("c2", blk[5], y)
        out, _ = ops.instnorm(y, _f(blk[6].weight), _f(blk[6].bias), eps=blk[6].eps, residual=x, act=L.ACT_ELU,
                              out_dtype=F32, **kw)
        return out

    def run(self, msa, pair, att):
        """msa fp32 [B,N,L,D], pair fp32 [B,L,L,Dp], att fp32 [B,L,L,H] -> new pair fp32."""
        B, N, Lr, D = msa.shape
        P, Dp = self.d_proj, self.d_pair
        dev = msa.device
        msa1d, xt, yt, Np = self._msa_operands(msa)
        # feature tensor (K padded to a multiple of 8)
        Kf = pad8(self.d_feat)
        feat = torch.empty(B, Lr, Lr, Kf, device=dev, dtype=T())  # every feature column is written below; only the K padding
        if Kf > self.d_feat:                                       # needs zeros (a full memset of this tensor is 0.38 GB)
            zpad = ops.zeros(B * Lr * Lr, Kf - self.d_feat, device=dev, dtype=T())
            ops.copy4d(zpad, (0, 0, Kf - self.d_feat, 1), feat, (0, 0, Kf, 1), (1, 1, B * Lr * Lr, Kf - self.d_feat),
                       y_off=self.d_feat)
        if self.outer_product_mean.fused_ok(P, Np, Lr) and Dp == 288:
            self.outer_product_mean.run_into(xt, yt, self.ln_coevol_feat, feat, Kf)  # outer -> LN -> Linear -> LN -> feat[..., :Dp]
        else:
            coevol = self.outer_product_mean.run(xt, yt, Np)  # fp32 [B,L,L,Dp]
            ln(self.ln_coevol_feat, coevol, out=feat, out_ld=Kf, out_off=0)
        H = att.shape[-1]
        blk = self.resnet[1].fn
        cond = RT.condition and ops.is_h16(T()) and Dp % 4 == 0   # (rf_center_apply moves 16-byte chunks of a pixel)
        bias = _f(self.resnet[0].bias)
        if cond:
            bias_b = self._center_1d(msa1d, bias)   # msa1d -= its mean over the positions; [B, Dp] bias that carries W * mean
        ops.tile_1d_feats(msa1d, feat, Kf, Dp, B, Lr, 2 * P)
        ln(self.ln_pair, pair, out=feat, out_ld=Kf, out_off=Dp + 4 * P)
        ops.copy4d(att.contiguous(), (0, 0, H, 1), feat, (0, 0, Kf, 1), (1, 1, B * Lr * Lr, H), y_off=2 * Dp + 4 * P)
        if cond:
            x = torch.empty(B, Lr, Lr, Dp, device=dev, dtype=F32)
            for b in range(B):
                ops.linear(feat[b], self.wt("f", self.resnet[0], kpad=Kf), bias_b[b], out=x[b])
            cx = ops.sample_mean(x)   # an estimate serves: both identities hold for ANY constant (a row-sharded picture needs the
            #                           same constant on every rank: run_rows all-reduces the exact mean)
            y = conv3x3(self, "c1", blk[1], ops.center_apply(x, cx, out_dtype=T()), 1)
            ops.conv3x3_border_fix(y, self._conv_taps(blk[1], cx), 1)
        else:
            x = ops.linear(feat, self.wt("f", self.resnet[0], kpad=Kf), bias, out_dtype=F32)
            y = conv3x3(self, "c1", blk[1], ops.cast(x, T()), 1)
        y, _ = ops.instnorm(y, _f(blk[2].weight), _f(blk[2].bias), eps=blk[2].eps, act=L.ACT_ELU, out_dtype=T())
        if self.training:
            dropout_(y, _p(blk[4]))   # rf.py:455
        y = conv3x3(self, "c2", blk[5], y, 1)
        out, _ = ops.instnorm(y, _f(blk[6].weight), _f(blk[6].bias), eps=blk[6].eps, residual=x, act=L.ACT_ELU,
                              out_dtype=F32)
        return out

    def forward(self, msa, pair, att):
        return self.run(msa.float().contiguous(), pair.float().contiguous(), att.float().contiguous())


def conv3x3(mod, key, conv, x, dilation, out_dtype=None):
    """3x3 'same' convolution on NHWC x (T) as implicit GEMM (rf.py:452,456; resnet.py:19-38)."""
    B, Hh, Ww, Cc = x.shape
    Co = conv.weight.shape[0]
    wk = mod.cached(("conv", key), lambda: conv.weight.detach().permute(0, 2, 3, 1).reshape(Co, 9 * Cc).to(T()).contiguous())
    out = torch.empty(B, Hh, Ww, Co, device=x.device, dtype=out_dtype or T())
    ops.gemm(x, wk, out, B * Hh * Ww, Co, 9 * Cc, conv=(B, Hh, Ww, Cc, dilation),
             bias=_f(conv.bias) if conv.bias is not None else None)
    return out


# ================================================================================================
# pair axial attention
# ================================================================================================
class PairUpdateWithAxialAttentionLayer(RFModule):
    """rf.py:501-528.  RowWise: sequences along dim 1 (i) for fixed j; ColWise: along dim 2 (rf.py:31-54)."""

    def __init__(self, d_pair, d_ff, n_heads, p_dropout, performer_kws):
        super().__init__()
        self.row_attn = PerformerSelfAttention(dim=d_pair, heads=n_heads, dropout=p_dropout,
                                               generalized_attention=True, **performer_kws)
        self.col_attn = PerformerSelfAttention(dim=d_pair, heads=n_heads, dropout=p_dropout,
                                               generalized_attention=True, **performer_kws)
        self.ff = FeedForward(d_pair, d_ff, p_dropout)
        # same aliasing as the reference: layer.k.fn.0 = LayerNorm, layer.0.fn.1.fn = row_attn, ...
        self.layer = nn.Sequential(
            Residual(nn.Sequential(LayerNorm(d_pair), RowWise(self.row_attn))),
            Residual(nn.Sequential(LayerNorm(d_pair), ColWise(self.col_attn))),
            Residual(nn.Sequential(LayerNorm(d_pair), self.ff)),
        )

    def run(self, x, xn=None, next_ln=None, row_group=None):
        """x fp32 in place; xn = layer[0] pre-norm of x if already produced; returns next_ln(x) or None.
        row_group: x holds a block of ROWS (dim 1) of the pair tensor, the other row blocks live on the other ranks of this
        torch.distributed group (shard.pair_axial_layer_row_sharded): the RowWise attention all-reduces its contexts, the
        ColWise attention and the feed-forward are local."""
        l0, l1, l2 = self.layer[0].fn[0], self.layer[1].fn[0], self.layer[2].fn[0]
        if row_group is not None and RT.rowshard_attention == "transpose":
            # the direction that crosses the row blocks, on the TRANSPOSED blocks: two transposing exchanges of the fp32 stream
            # (1/world of the tensor per rank each) instead of the all-reduce of the contexts (0.57 GB per rank at L = 1024 whatever
            # the world size), and every sequence is whole on its rank again -- the fused FAVOR+ kernel runs as in one process
            from . import shard
            xt = shard.transpose_row_sharded(x, row_group)
            self.row_attn.attend(ln(l0, xt), xt, axis=2)
            ops.axpby(shard.transpose_row_sharded(xt, row_group), 1.0, None, 0.0, x)
            xn = None
        else:
            if xn is None:
                xn = ln(l0, x)
            xn = self.row_attn.attend(xn, x, axis=1, next_ln=l1, seq_group=row_group,
                                      drops=(self.row_attn.p_dropout,) if self.training else ())
        if xn is None:
            xn = ln(l1, x)
        xn = self.col_attn.attend(xn, x, axis=2, next_ln=l2, drops=(self.col_attn.p_dropout,) if self.training else ())
        if xn is None:
            xn = ln(l2, x)
        return self.ff.apply_residual(xn, x, next_ln)   # (its hidden dropout, rf.py:519, is the FeedForward's own)

    def forward(self, x):
        x = fresh_f32(x)
        self.run(x)
        return x


class PairUpdateWithAxialAttention(RFModule):
    """rf.py:531-547."""

    def __init__(self, d_pair, d_ff, n_heads, p_dropout, n_encoder_layers, performer_kws={}):
        super().__init__()
        self.layers = nn.ModuleList([PairUpdateWithAxialAttentionLayer(d_pair, d_ff, n_heads, p_dropout, performer_kws)
                                     for _ in range(n_encoder_layers)])

    def run(self, x, row_group=None):
        xn = None
        for i, layer in enumerate(self.layers):
            nxt = self.layers[i + 1].layer[0].fn[0] if i + 1 < len(self.layers) else None
            xn = layer.run(x, xn=xn, next_ln=nxt, row_group=row_group)

    def forward(self, x):
        x = fresh_f32(x)
        self.run(x)
        return x


# ================================================================================================
# MSA update with pair
# ================================================================================================
class Symmetrization(nn.Module):
    """rf.py:550-556 (container; the arithmetic is fused into rf_sym_layernorm)."""

    def forward(self, x):
        xt = torch.empty_like(x)
        B, Lr, _, D = x.shape
        ops.copy4d(x.contiguous(), (Lr * Lr * D, D, Lr * D, 1), xt, (Lr * Lr * D, Lr * D, D, 1), (B, Lr, Lr, D))
        return ops.axpby(x.contiguous(), 0.5, xt, 0.5, torch.empty_like(x))


class MsaUpdateWithPairLayer(RFModule):
    """rf.py:559-595."""

    def __init__(self, d_msa, d_pair, n_heads, p_dropout=0.1):
        super().__init__()
        self.n_heads = n_heads
        self.pair2att = nn.Sequential(Symmetrization(), LayerNorm(d_pair), Linear(d_pair, n_heads),
                                      nn.Dropout(p_dropout), nn.Identity(), nn.Softmax(dim=-1))
        self.msa2value = nn.Sequential(LayerNorm(d_msa), Linear(d_msa, d_msa), nn.Identity())
        self.ff = Residual(nn.Sequential(LayerNorm(d_msa), FeedForward(d_msa, d_msa, p_dropout)), p_dropout=p_dropout)
        self.p_dropout = p_dropout   # rf.py:586,592: dropout on the attention output before it is added to the msa

    def folded_att_proj(self):
        """LayerNorm affine folded into the 288->H projection: W' = W*gamma, b' = W beta + b (fp32)."""
        lnm, lin = self.pair2att[1], self.pair2att[2]
        w = lin.weight.detach().float() * lnm.weight.detach().float()[None, :]
        b = lin.weight.detach().float() @ lnm.bias.detach().float() + lin.bias.detach().float()
        return w, b

    def run(self, msa, att, xn=None, next_ln=None, row_group=None):
        """msa fp32 [B,N,L,D] in place; att: T [H,B,L,L] (softmax over the last dim); xn = msa2value pre-norm if the
        previous GEMM produced it; returns next_ln(msa) or None.
        row_group: att holds this rank's rows [H,B,h,L] of the maps (pair-track row blocks, shard.shard_range); the values are
        computed from the replicated msa, positions r0..r1 are updated here (attention + feed-forward) and the updated position
        slices are all-gathered so that msa is whole again on every rank."""
        if row_group is not None:
            return self._run_rows(msa, att, xn, row_group)
        B, N, Lr, D = msa.shape
        H = self.n_heads
        dv = D // H
        v_t = torch.empty(B, N, D, Lr, device=msa.device, dtype=T())
        lin = self.msa2value[1]
        if xn is None:
            xn = ln(self.msa2value[0], msa)
        ops.gemm(self.wt("v", lin), xn, v_t, D, Lr, D, batch=(B * N, 1, 1), b_bs=(Lr * D, 0, 0),
                 c_bs=(D * Lr, 0, 0), c_row=(0, 0, Lr), bias=_f(lin.bias), bias_mode=L.BIAS_ROW)
        # msa += att @ v  (rf.py:592-595), scattered back to [b,n,i,(h,d)]
        def attv(dst):
            ops.gemm(att, v_t, dst, Lr, N * dv, Lr, batch=(H, B, 1),
                     a_bs=(B * Lr * Lr, Lr * Lr, 0), a_row=(0, 0, Lr),
                     b_bs=(dv * Lr, N * D * Lr, 0), b_row=(dv, D * Lr, Lr),
                     c_bs=(dv, N * Lr * D, 0), c_row=(0, 0, D), c_col=(dv, Lr * D), residual=dst)
        if self.training and self.p_dropout and self.p_dropout > 0:
            add_dropped(msa, attv, (self.p_dropout,))
            return self.ff.fn[1].apply_residual(ln(self.ff.fn[0], msa), msa, next_ln, drops=(self.ff.p_dropout,))
        attv(msa)
        return self.ff.fn[1].apply_residual(ln(self.ff.fn[0], msa), msa, next_ln)


def _msa_update_rows(self, msa, att, xn, row_group):
    """MsaUpdateWithPairLayer.run for a block of map rows (see there)."""
    from . import shard
    B, N, Lr, D = msa.shape
    H = self.n_heads
    dv = D // H
    h = att.shape[2]
    r0, r1 = shard.shard_range(Lr, shard.group_size(row_group), shard.group_rank(row_group))
    if r1 - r0 != h:
        raise ValueError(f"attention rows {h} do not match this rank's share {r1 - r0} of {Lr} positions")
    v_t = torch.empty(B, N, D, Lr, device=msa.device, dtype=T())
    lin = self.msa2value[1]
    if xn is None:
        xn = ln(self.msa2value[0], msa)
    ops.gemm(self.wt("v", lin), xn, v_t, D, Lr, D, batch=(B * N, 1, 1), b_bs=(Lr * D, 0, 0),
             c_bs=(D * Lr, 0, 0), c_row=(0, 0, Lr), bias=_f(lin.bias), bias_mode=L.BIAS_ROW)
    rows = torch.empty(B, N, h, D, device=msa.device, dtype=F32)   # msa[:, :, r0:r1]
    if h > 0:
        ops.copy4d(msa, (N * Lr * D, Lr * D, D, 1), rows, (N * h * D, h * D, D, 1), (B, N, h, D), x_off=r0 * D)
        ops.gemm(att, v_t, rows, h, N * dv, Lr, batch=(H, B, 1),
                 a_bs=(B * h * Lr, h * Lr, 0), a_row=(0, 0, Lr),
                 b_bs=(dv * Lr, N * D * Lr, 0), b_row=(dv, D * Lr, Lr),
                 c_bs=(dv, N * h * D, 0), c_row=(0, 0, D), c_col=(dv, h * D), residual=rows)
        self.ff.fn[1].apply_residual(ln(self.ff.fn[0], rows), rows, None)
    shard.all_gather_positions(rows, msa, row_group)
    return None


MsaUpdateWithPairLayer._run_rows = _msa_update_rows


def pair_to_att(layers, pair, row_group=None):
    """Shared front of the MsaUpdateWithPairLayer stack of one block (they all see the same pair):
    one symmetrise+normalise pass, one GEMM for every layer's head logits, per-(layer,head) softmax.
    Returns a list of T [H,B,L,L] ([H,B,h,L] for a block of h pair rows: row_group, the transposed sub-blocks of the
    symmetrisation come from the other ranks)."""
    B, Lr, _, Dp = pair.shape
    H = layers[0].n_heads
    nl = len(layers)
    if row_group is not None:
        return _pair_rows_to_att(layers, pair, row_group)
    xs = ops.sym_layernorm(pair, T(), eps=layers[0].pair2att[1].eps)
    holder = layers[0]

    def fold():
        ws, bs = zip(*[l.folded_att_proj() for l in layers])
        return torch.cat(ws, 0).to(T()).contiguous(), torch.cat(bs, 0).contiguous()

    wc, bc = holder.cached(("att_fold", nl), fold)
    logits = ops.linear(xs, wc, bc, out_dtype=F32)  # [B,L,L,nl*H]
    NH = nl * H
    if holder.training:
        dropout_(logits, _p(holder.pair2att[3]))   # rf.py:567: Dropout sits between the projection and the softmax
    # every (layer, head) softmax in one launch: problem z = li*H + h is column z of the logits
    att_all = torch.empty(nl, H, B, Lr, Lr, device=pair.device, dtype=T())
    ops.softmax_batched(logits, 1, Lr * NH, NH, att_all, B * Lr * Lr, Lr, B * Lr, Lr, NH)
    return [att_all[li] for li in range(nl)]


def _pair_rows_to_att(layers, pair, row_group):
    from . import shard
    B, h, Lr, Dp = pair.shape
    H, nl = layers[0].n_heads, len(layers)
    NH = nl * H
    xt = shard.transpose_row_sharded(pair, row_group)
    sym = ops.axpby(pair, 0.5, xt, 0.5, torch.empty_like(pair))
    one, zero = ops.fill(torch.empty(Dp, device=pair.device, dtype=F32), 1.0), ops.zeros(Dp, device=pair.device, dtype=F32)
    xs = ops.layernorm(sym, one, zero, eps=layers[0].pair2att[1].eps, out_dtype=T())   # no affine: folded into the projection
    holder = layers[0]

    def fold():
        ws, bs = zip(*[l.folded_att_proj() for l in layers])
        return torch.cat(ws, 0).to(T()).contiguous(), torch.cat(bs, 0).contiguous()

    wc, bc = holder.cached(("att_fold", nl), fold)
    logits = ops.linear(xs, wc, bc, out_dtype=F32)  # [B,h,L,nl*H]
    att_all = torch.empty(nl, H, B, h, Lr, device=pair.device, dtype=T())
    if h > 0:
        ops.softmax_batched(logits, 1, Lr * NH, NH, att_all, B * h * Lr, Lr, B * h, Lr, NH)
    return [att_all[li] for li in range(nl)]


class MsaUpdateWithPair(RFModule):
    """rf.py:598-610 (the reference hides these layers in a plain list; here they are registered)."""

    def __init__(self, d_msa, d_pair, n_heads, n_encoder_layers=4, p_dropout=0.1):
        super().__init__()
        self.encoder_layers = nn.ModuleList([MsaUpdateWithPairLayer(d_msa, d_pair, n_heads, p_dropout)
                                             for _ in range(n_encoder_layers)])

    def run(self, msa, pair, row_group=None):
        """msa fp32 [B,N,L,D] in place (replicated on every rank of row_group); pair: the whole tensor, or this rank's rows."""
        layers = list(self.encoder_layers)
        atts = pair_to_att(layers, pair, row_group)
        xn = None
        for i, (layer, att) in enumerate(zip(layers, atts)):
            nxt = layers[i + 1].msa2value[0] if i + 1 < len(layers) else None
            xn = layer.run(msa, att, xn=xn, next_ln=nxt, row_group=row_group)

    def forward(self, msa, pair):
        msa = fresh_f32(msa)
        self.run(msa, pair.float().contiguous())
        return msa


def _msa_update_with_pair_layer_forward(self, msa, pair):
    msa = fresh_f32(msa)
    self.run(msa, pair_to_att([self], pair.float().contiguous())[0])
    return msa


MsaUpdateWithPairLayer.forward = _msa_update_with_pair_layer_forward


# ================================================================================================
# initial coordinates (graph transformer over the dense residue graph)
# ================================================================================================
class GraphTransformer(RFModule):
    """rf.py:613-664."""

    def __init__(self, d_node_in, d_node_out, d_edge, n_heads, p_dropout=0.15):
        super().__init__()
        self.scale = d_node_out ** (-0.5)
        self.node_update = Linear(d_node_in, d_node_out * n_heads, bias=True)
        self.node_to_q = Linear(d_node_in, d_node_out * n_heads, bias=True)
        self.node_to_k = Linear(d_node_in, d_node_out * n_heads, bias=True)
        self.node_to_v = Linear(d_node_in, d_node_out * n_heads, bias=True)
        self.edge_emb = Linear(d_edge, d_node_out * n_heads, bias=False)
        self.n_heads, self.d_out = n_heads, d_node_out
        self.p_dropout = p_dropout   # rf.py:628,658: att_dropout on the attention probabilities

    def run(self, node, edge_t):
        """node fp32 [B,L,dn]; edge_t T [B,L,L,de] -> fp32 [B,L,H*d]"""
        B, Lr, _ = node.shape
        H, d = self.n_heads, self.d_out
        nt = ops.cast(node, T())
        q = ops.linear(nt, self.wt("q", self.node_to_q), _f(self.node_to_q.bias))
        k = ops.linear(nt, self.wt("k", self.node_to_k), _f(self.node_to_k.bias))
        v = ops.linear(nt, self.wt("v", self.node_to_v), _f(self.node_to_v.bias))
        e = ops.linear(edge_t, self.wt("e", self.edge_emb), None)
        upd = torch.empty(B, Lr, H * d, device=node.device, dtype=F32)
        if self.training and self.p_dropout and self.p_dropout > 0:
            off = RT.train_offset
            RT.train_offset += (B * H * Lr * Lr + 3) // 4
            ops.graph_attention(q, k, v, e, upd, B, Lr, H, d, self.scale, dropout=(self.p_dropout, RT.train_seed, off))
        else:
            ops.graph_attention(q, k, v, e, upd, B, Lr, H, d, self.scale)
        return ops.linear(nt, self.wt("u", self.node_update), _f(self.node_update.bias), out_dtype=F32, residual=upd)

    def forward(self, node_feat, edge_feat, edge_mask):
        if edge_mask is not None:
            raise NotImplementedError("edge_mask is unused on the forward path (rf.py:731)")
        return self.run(node_feat.float().contiguous(), ops.cast(edge_feat.float().contiguous(), T()))


class GraphTransformerBlock(RFModule):
    """rf.py:667-676."""

    def __init__(self, d_node_in, d_node_out, d_edge, n_heads, p_dropout=0.15):
        super().__init__()
        self.attn = GraphTransformer(d_node_in, d_node_out, d_edge, n_heads, p_dropout)
        self.ln = LayerNorm(d_node_out * n_heads)
        self.to_out = nn.Sequential(Linear(d_node_out * n_heads, d_node_in), nn.ELU())

    def run(self, node, edge_t):
        h = ln(self.ln, self.attn.run(node, edge_t))
        return ops.linear(h, self.wt("o", self.to_out[0]), _f(self.to_out[0].bias), out_dtype=F32, act=L.ACT_ELU,
                          residual=node)

    def forward(self, node_feat, edge_feat, edge_mask):
        return self.run(node_feat.float().contiguous(), ops.cast(edge_feat.float().contiguous(), T()))


def _node_input(mod, msa, seq_onehot, out_dtype=None):
    """[sum_n w*LN(msa) | onehot] zero-padded to a multiple of 8 columns, T (rf.py:715-724, 789-798)."""
    B, N, Lr, D = msa.shape
    # the fp32 (structure-track) form keeps LayerNorm(msa) in fp32 too: nothing upstream of the SE(3) stack is rounded
    m = ln(mod.ln_msa, msa, out_dtype=F32 if (out_dtype == F32 and RT.struct_inputs_fp32) else None)
    w = mod.poswise_weight.weights_collapsed(msa, mod.ln_msa, m)
    Kp = pad8(D + 21)
    tmp = ops.zeros(B, Lr, Kp, device=msa.device, dtype=F32)
    ops.weighted_msa_sum(m, w, tmp, Kp)
    ops.copy4d(seq_onehot.contiguous(), (0, 0, 21, 1), tmp, (0, 0, Kp, 1), (1, 1, B * Lr, 21), y_off=D)
    return ops.cast(tmp, out_dtype or T()), Kp


class InitialCoordGenerationWithMsaAndPair(RFModule):
    """rf.py:679-749."""

    def __init__(self, d_msa, d_pair, d_node=64, d_edge=64, n_heads=4, n_layers=4, p_dropout=0.1):
        super().__init__()
        self.ln_msa = LayerNorm(d_msa)
        self.ln_pair = LayerNorm(d_pair)
        self.poswise_weight = PositionWiseWeightFactor(d_msa, 1, p_dropout)
        self.node_embed = nn.Sequential(Linear(d_msa + 21, d_node), nn.ELU())
        self.edge_embed = nn.Sequential(Linear(d_pair + 1, d_edge), nn.ELU())
        self.blocks = nn.ModuleList([GraphTransformerBlock(d_node, d_node, d_edge, n_heads, p_dropout)
                                     for _ in range(n_layers)])
        self.to_out = Linear(d_node, 9)

    def run(self, msa, pair, seq_onehot, aa_idx):
        B, Lr, _, Dp = pair.shape
        Ke = pad8(Dp + 1)
        ein = ops.zeros(B, Lr, Lr, Ke, device=pair.device, dtype=T())
        nin, Kp = _node_input(self, msa, seq_onehot)
        node = ops.linear(nin, self.wt("n", self.node_embed[0], kpad=Kp), _f(self.node_embed[0].bias), out_dtype=F32,
                          act=L.ACT_ELU)
        ln(self.ln_pair, pair, out=ein, out_ld=Ke)
        ops.seqsep_feature(aa_idx.contiguous(), ein, Ke, Dp)  # clamp(sign(d) log(|d|+1), 0, 5.5), rf.py:746-749
        edge = ops.linear(ein, self.wt("e", self.edge_embed[0], kpad=Ke), _f(self.edge_embed[0].bias), act=L.ACT_ELU)
        for blk in self.blocks:
            node = blk.run(node, edge)
        xyz = ops.linear(ops.cast(node, T()), self.wt("o", self.to_out), _f(self.to_out.bias), out_dtype=F32)
        return xyz.view(B, Lr, 3, 3)

    def forward(self, msa, pair, seq_onehot, aa_idx):
        return self.run(msa.float().contiguous(), pair.float().contiguous(), seq_onehot.float(), aa_idx)


# ================================================================================================
# prediction head (ResNet over the pair map)
# ================================================================================================
class ResBlock2D(RFModule):
    """resnet.py:15-44."""

    def __init__(self, channel, kernel_size, dilation, p_dropout=0.15):
        super().__init__()
        self.dilation = dilation
        self.layer = Residual(nn.Sequential(
            nn.Conv2d(channel, channel, kernel_size, dilation=dilation, padding="same", bias=False),
            nn.InstanceNorm2d(channel, affine=True, eps=1e-6), nn.ELU(), nn.Dropout(p_dropout),
            nn.Conv2d(channel, channel, kernel_size, dilation=dilation, padding="same", bias=False),
            nn.InstanceNorm2d(channel, affine=True, eps=1e-6)))

    def _run_rows_b1(self, x_t, x_f, row_group, rows_global, next_halo):
        """Row block of ONE picture (a contiguous slab): the convolutions read pre-haloed buffers whose interiors the producers
        wrote directly -- only the halo rows move (shard.exchange_row_halos_inplace), no full-size copy."""
        from . import shard
        f, d = self.layer.fn, self.dilation
        _, h, W, Cc = x_t.shape
        kw = {"row_group": row_group, "rows_global": rows_global}
        xh = shard.haloed_parent(x_t, d)
        if xh is None:   # (the first block of a chain: its input was not produced into a haloed buffer)
            xh = shard.haloed_buffer(h, W, Cc, d, x_t.device, x_t.dtype)
            ops.axpby(x_t.contiguous(), 1.0, None, 0.0, shard.interior(xh, d))
        shard.exchange_row_halos_inplace(xh, d, row_group)
        y = conv3x3(self, "c1", f[0], xh, d)
        yh = shard.haloed_buffer(h, W, y.shape[-1], d, x_t.device, T())
        ops.instnorm(shard.interior(y, d), _f(f[1].weight), _f(f[1].bias), eps=f[1].eps, act=L.ACT_ELU, out=shard.interior(yh, d), **kw)
        shard.exchange_row_halos_inplace(yh, d, row_group)
        y = conv3x3(self, "c2", f[4], yh, d)
        nxt = shard.haloed_buffer(h, W, y.shape[-1], next_halo, x_t.device, T())
        o_f, o_t = ops.instnorm(shard.interior(y, d), _f(f[5].weight), _f(f[5].bias), eps=f[5].eps, residual=x_f, act=L.ACT_ELU,
                                out_dtype=F32, out2=shard.interior(nxt, next_halo), **kw)
        return o_t, o_f

    def run(self, x_t, x_f, row_group=None, rows_global=None, next_halo=0):
        """x_t: T NHWC (conv input), x_f: fp32 copy (residual).  Returns (T, fp32) of elu(block(x)+x).
        row_group / rows_global: x holds a block of the picture's rows (shard.resblock_row_sharded): every convolution first
        fetches `dilation` rows from each neighbouring rank, the InstanceNorm sums are all-reduced."""
        f = self.layer.fn
        kw = {} if row_group is None else {"row_group": row_group, "rows_global": rows_global}
        if row_group is not None and x_t.shape[0] == 1:
            return self._run_rows_b1(x_t, x_f, row_group, rows_global, next_halo)

        def conv(key, c, x):
            if row_group is None:
                return conv3x3(self, key, c, x, self.dilation)
            from . import shard
            xh = shard.exchange_row_halos(x, self.dilation, row_group)
            return shard.drop_row_halos(conv3x3(self, key, c, xh, self.dilation), self.dilation)

        y = conv("c1", f[0], x_t)
        y, _ = ops.instnorm(y, _f(f[1].weight), _f(f[1].bias), eps=f[1].eps, act=L.ACT_ELU, out_dtype=T(), **kw)
        if self.training:
            dropout_(y, _p(f[3]))   # resnet.py:30
        y = conv("c2", f[4], y)
        o_f, o_t = ops.instnorm(y, _f(f[5].weight), _f(f[5].bias), eps=f[5].eps, residual=x_f, act=L.ACT_ELU,
                                out_dtype=F32, out2_dtype=T(), **kw)
        return o_t, o_f

    def forward(self, x):  # NCHW like the reference
        xf = x.float().permute(0, 2, 3, 1).contiguous()
        return self.run(ops.cast(xf, T()), xf)[1].permute(0, 3, 1, 2)


class ResNet(RFModule):
    """resnet.py:47-83."""

    def __init__(self, n_res_blocks, in_channels, intermediate_channels, out_channels, dilations=[1, 2, 4, 8],
                 p_dropout=0.15):
        super().__init__()
        layers = [nn.Conv2d(in_channels, intermediate_channels, 1, bias=False),
                  nn.InstanceNorm2d(intermediate_channels, affine=True, eps=1e-6), nn.ELU()]
        for b in range(n_res_blocks):
            layers.append(ResBlock2D(intermediate_channels, kernel_size=3, dilation=dilations[b % len(dilations)],
                                     p_dropout=p_dropout))
        layers.append(nn.Conv2d(intermediate_channels, out_channels, 1))
        self.layer = nn.Sequential(*layers)
        self.n_res_blocks = n_res_blocks

    def run(self, x_t, row_group=None, rows_global=None):
        """x_t: T NHWC -> fp32 NHWC logits.  row_group / rows_global: x_t is a block of the picture's rows (see ResBlock2D.run)."""
        l0, l1 = self.layer[0], self.layer[1]
        kw = {} if row_group is None else {"row_group": row_group, "rows_global": rows_global}
        h = ops.linear(x_t, self.wt("in", l0), None)
        dil = [self.layer[3 + b].dilation for b in range(self.n_res_blocks)] + [0]
        if row_group is not None and h.shape[0] == 1:
            from . import shard
            first = shard.haloed_buffer(h.shape[1], h.shape[2], h.shape[3], dil[0], h.device, T())
            h_f, h_t = ops.instnorm(h, _f(l1.weight), _f(l1.bias), eps=l1.eps, act=L.ACT_ELU, out_dtype=F32,
                                    out2=shard.interior(first, dil[0]), **kw)
        else:
            h_f, h_t = ops.instnorm(h, _f(l1.weight), _f(l1.bias), eps=l1.eps, act=L.ACT_ELU, out_dtype=F32, out2_dtype=T(), **kw)
        for b in range(self.n_res_blocks):
            h_t, h_f = self.layer[3 + b].run(h_t, h_f, next_halo=dil[b + 1], **kw)
        lo = self.layer[3 + self.n_res_blocks]
        return ops.linear(h_t, self.wt("out", lo), _f(lo.bias), out_dtype=F32)

    def forward(self, x):  # NCHW in / NCHW out like the reference
        xf = x.float().permute(0, 2, 3, 1).contiguous()
        return self.run(ops.cast(xf, T())).permute(0, 3, 1, 2)


class PredictionHead(RFModule):
    """rf.py:1130-1172."""

    def __init__(self, in_channels, n_res_blocks, p_dropout):
        super().__init__()
        c = in_channels
        self.proj = nn.Sequential(LayerNorm(c), Linear(c, c), nn.Dropout(p_dropout), nn.Identity())
        self.dist_head = nn.Sequential(ResNet(n_res_blocks, c, c, 37, p_dropout=p_dropout), nn.Identity())
        self.omega_head = nn.Sequential(ResNet(n_res_blocks, c, c, 37, p_dropout=p_dropout), nn.Identity())
        self.theta_head = nn.Sequential(ResNet(n_res_blocks, c, c, 37, p_dropout=p_dropout), nn.Identity())
        self.phi_head = nn.Sequential(ResNet(n_res_blocks, c, c, 19, p_dropout=p_dropout), nn.Identity())

    def forward(self, pair):
        return self.run(pair.float().contiguous())

    def run(self, pair, row_group=None):
        """pair fp32 [B, L, L, C] -> the four logit maps (fp32 NHWC).  row_group: `pair` is this rank's block of rows
        [B, h, L, C] (contiguous split shard.shard_range(L, world, rank)); the symmetrisation (rf.py:1160) fetches the
        transposed sub-blocks from the other ranks, the ResNets exchange halo rows and InstanceNorm sums; returns the same rows
        of the logit maps."""
        B, h, Lr, Cc = pair.shape
        kwc = {} if row_group is None else {"row_group": row_group, "rows_global": Lr}
        # Operand conditioning for the 16-bit modes (exact in exact arithmetic): every ResNet starts conv1x1 (no bias) ->
        # InstanceNorm (resnet.py:57-60), which is invariant to a per-channel constant of the conv's input, and the mean over
        # the picture of 0.5 (x + x^T) is the mean of x.  At random init that constant is ~19x the part that varies over the
        # picture (tools/precision_probe.py: the first InstanceNorm amplifies a white input error 19x), so x is rounded to 16
        # bits AFTER removing it -- and so is the projection's own operand LayerNorm(pair) (the other 6e-3 of the fp16 mode's
        # logits gap): W (t - mean t) is the projection minus ITS mean over the picture, bias and all.
        cond = RT.head_center and ops.is_h16(T()) and Cc % 4 == 0 and self.proj[1].weight.shape[0] % 4 == 0
        if cond and not self.training:
            t = ln(self.proj[0], pair, out_dtype=F32)
            x = ops.linear(ops.center_apply(t, ops.channel_mean(t, **kwc), out_dtype=T()), self.wt("p", self.proj[1]), None, out_dtype=F32)
        else:
            x = ops.linear(ln(self.proj[0], pair), self.wt("p", self.proj[1]), _f(self.proj[1].bias), out_dtype=F32)
            if self.training:
                dropout_(x, _p(self.proj[2]))   # rf.py:1138
            if cond:
                ops.center_apply(x, ops.channel_mean(x, **kwc))
        if row_group is None:
            xt = torch.empty_like(x)
            ops.copy4d(x, (Lr * Lr * Cc, Cc, Lr * Cc, 1), xt, (Lr * Lr * Cc, Lr * Cc, Cc, 1), (B, Lr, Lr, Cc))
            kw = {}
        else:
            from . import shard
            xt = shard.transpose_row_sharded(x, row_group)
            kw = {"row_group": row_group, "rows_global": Lr}
        xs = ops.axpby(x, 0.5, xt, 0.5, torch.empty(x.shape, device=x.device, dtype=T()))
        x_t = ops.cast(x, T())
        return {"theta": self.theta_head[0].run(x_t, **kw), "phi": self.phi_head[0].run(x_t, **kw),
                "dist": self.dist_head[0].run(xs, **kw), "omega": self.omega_head[0].run(xs, **kw)}
