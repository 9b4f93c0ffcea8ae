"""Thin Python wrappers over the C ABI of librfmi.so (include/rfmi.h).

PyTorch is used here for device memory (tensors), the current HIP stream and dtype tags only;
every arithmetic step is a kernel of librfmi.so reached through ctypes with raw device pointers.

Device rule: every operand of a call lives on ONE device and that device is the calling thread's current device
(kernels are enqueued on its current stream); anything else raises RfmiError instead of launching on the wrong GPU.
`RoseTTAFold.forward` enters `torch.cuda.device(input.device)` itself, so the model can live on any GPU.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import lib, check, GemmDesc, I64x4, I64x3

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
_H16_CODE = {torch.bfloat16: L.RF_BF16, torch.float16: L.RF_F16}
_H16_TORCH = {L.RF_BF16: torch.bfloat16, L.RF_F16: torch.float16}


def is_h16(dtype):
    """A 16-bit MFMA operand type (bfloat16: librfmi.so, float16: librfmi_f16.so)."""
    return dtype in _H16_CODE


def h16():
    """torch dtype of the 16-bit operand type of the library that is active now (model.set_compute_dtype)."""
    return _H16_TORCH[L.active_h16()]


def dcode(dtype):
    if dtype == torch.float32:
        return L.RF_F32
    code = _H16_CODE.get(dtype)
    if code is None:
        raise TypeError(f"unsupported dtype {dtype}")
    if code != L.active_h16():  # (the library would reject the code too: this names the cause)
        raise L.RfmiError(f"{dtype} tensor passed while the active library computes in {h16()} "
                          "(set_compute_dtype selects the library)")
    return code


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, off_elems=0):
    if t is None:
        return None
    if t.dtype in _H16_CODE and _H16_CODE[t.dtype] != L.active_h16():
        # entry points without a dtype argument read 16-bit memory as the active library's type: never hand them the other one
        raise L.RfmiError(f"{t.dtype} tensor passed while the active library computes in {h16()}")
    return C.c_void_p(t.data_ptr() + off_elems * t.element_size())


def _i4(v):
    return C.byref(I64x4(*[int(x) for x in v]))


def _need_cuda(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise L.RfmiError("librfmi ops need device tensors (there is no CPU fallback)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise L.RfmiError(f"operands on different devices: {dev} and {t.device}")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise L.RfmiError(f"operands live on {dev} but the current device is cuda:{torch.cuda.current_device()} "
                          "(wrap the call in torch.cuda.device(...))")


# --------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, Cout, M, N, K, *, batch=(1, 1, 1), a_off=0, b_off=0, c_off=0,
         a_bs=(0, 0, 0), a_row=(0, 0, None), a_ko=0,
         b_bs=(0, 0, 0), b_row=(0, 0, None), b_ko=0, kc=0,
         c_bs=(0, 0, 0), c_row=(0, 0, None), c_col=(0, 0),
         bias=None, bias_mode=None, act=L.ACT_NONE, act_nvalid=0, act_eps=0.0, alpha=1.0,
         residual=None, res_off=None, conv=None, tile_cfg=0, ln=None, block_ln=None, rs=None):
    """C = epilogue(alpha * A @ B^T) with the strided/batched/chunked addressing of rf_gemm_desc.
    *_row = (rc, ro, ri): offset(m) = (m // rc)*ro + (m % rc)*ri, rc=0 -> m*ri.  ri=None -> K (A,B) / N (C).
    c_col = (cc, co).  Offsets are in elements.  conv = (n, h, w, c, dilation) selects implicit 3x3 im2col."""
    _need_cuda(A, B, Cout, bias, residual)
    if A.dtype != B.dtype:
        raise TypeError("A and B dtypes differ")
    d = GemmDesc()
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.nb0, d.nb1, d.nb2 = [int(x) for x in batch]
    d.ab_dtype, d.c_dtype = dcode(A.dtype), dcode(Cout.dtype)
    d.kc = int(kc)
    d.a_rc, d.a_ro, d.a_ri = int(a_row[0]), int(a_row[1]), int(K if a_row[2] is None else a_row[2])
    d.b_rc, d.b_ro, d.b_ri = int(b_row[0]), int(b_row[1]), int(K if b_row[2] is None else b_row[2])
    d.c_rc, d.c_ro, d.c_ri = int(c_row[0]), int(c_row[1]), int(N if c_row[2] is None else c_row[2])
    d.c_cc, d.c_co = int(c_col[0]), int(c_col[1])
    for i in range(3):
        d.a_bs[i], d.b_bs[i], d.c_bs[i] = int(a_bs[i]), int(b_bs[i]), int(c_bs[i])
    d.a_ko, d.b_ko = int(a_ko), int(b_ko)
    if conv is not None:
        d.a_mode = L.AMODE_CONV3X3
        d.conv_n, d.conv_h, d.conv_w, d.conv_c, d.conv_dil = [int(x) for x in conv]
    if bias is not None:
        if bias.dtype != F32:
            raise TypeError("bias must be fp32")
        d.bias_mode = L.BIAS_COL if bias_mode is None else bias_mode
        d.bias = bias.data_ptr()
    d.act, d.act_nvalid, d.act_eps, d.alpha = int(act), int(act_nvalid), float(act_eps), float(alpha)
    d.tile_cfg = int(tile_cfg)
    d.A = A.data_ptr() + a_off * A.element_size()
    d.B = B.data_ptr() + b_off * B.element_size()
    d.C = Cout.data_ptr() + c_off * Cout.element_size()
    if residual is not None:
        if residual.dtype != F32:
            raise TypeError("residual must be fp32")
        d.residual = residual.data_ptr() + (c_off if res_off is None else res_off) * 4
    if block_ln is not None:  # (gamma[1024], beta[1024], eps) with act=ACT_BLOCK_LN32: LayerNorm over 32x32 output blocks
        g, b, eps = block_ln
        _need_cuda(g, b)
        d.ln_gamma, d.ln_beta, d.ln_eps = g.data_ptr(), b.data_ptr(), float(eps)
    if rs is not None:  # (scale fp32, bstride, rpb, cg, ncols, alpha): row-group scale of the leading columns (rf_gemm_desc.rs)
        t_, bstride, rpb, cg, ncols, alpha_ = rs
        _need_cuda(t_)
        if t_.dtype != F32:
            raise TypeError("rs must be fp32")
        d.rs, d.rs_bstride, d.rs_rpb, d.rs_cg, d.rs_ncols, d.rs_alpha = t_.data_ptr(), int(bstride), int(rpb), int(cg), int(ncols), float(alpha_)
    if ln is not None:  # (out bf16 [M,N], gamma, beta, eps): fused LayerNorm of the result rows
        ln_out, g, b, eps = ln
        _need_cuda(ln_out, g, b)
        d.ln_out, d.ln_gamma, d.ln_beta, d.ln_eps = ln_out.data_ptr(), g.data_ptr(), b.data_ptr(), float(eps)
    check(lib.rf_gemm(C.byref(d), stream()), "rf_gemm")
    return Cout


_NO_WREG = bool(int(__import__("os").environ.get("RF_NO_WREG_GEMM", "0") or "0"))


def gemm_takes_row_scale(M, N, K):
    """The row-group scale (`rs=` of gemm) lives in the epilogue of the register-resident-weights kernel only
    (csrc/gemm_wreg.hip: rf_gemm_wreg_try); every other kernel answers RF_EINVAL to it.  True when a plain 16-bit
    projection of this shape goes to that kernel, i.e. when a caller may fold its per-row factor into the GEMM."""
    return (not _NO_WREG) and K in (288, 384) and M % 64 == 0 and M >= 16384 and N % 128 == 0 and N >= 256  # (+ a split-C output)


def linear(x, w, bias=None, *, out=None, out_dtype=None, act=L.ACT_NONE, residual=None, alpha=1.0, tile_cfg=0, ln=None):
    """out[..., N] = act(x[..., K] @ w[N, K]^T + bias) (+ residual).  x contiguous, w [N, Kw>=K] contiguous."""
    K = x.shape[-1]
    Mrows = x.numel() // K
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear: K mismatch {w.shape} vs {x.shape}")
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=out_dtype or x.dtype)
    gemm(x, w, out, Mrows, N, K, bias=bias, act=act, residual=residual, alpha=alpha, tile_cfg=tile_cfg, ln=ln)
    return out


# --------------------------------------------------------------------------------------------- norms
def layernorm(x, gamma, beta, *, eps=1e-5, out=None, out_dtype=None, out_ld=None, out_off=0, rows=None, D=None,
              x_ld=None, groups=1, act=L.ACT_NONE):
    D = x.shape[-1] if D is None else D
    rows = x.numel() // D if rows is None else rows
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    _need_cuda(x, out, gamma, beta)
    check(lib.rf_layernorm(ptr(x), dcode(x.dtype), D if x_ld is None else x_ld, ptr(out, out_off), dcode(out.dtype),
                           D if out_ld is None else out_ld, rows, D, ptr(gamma), ptr(beta), eps, groups, act,
                           stream()), "rf_layernorm")
    return out


def sym_layernorm(pair, out_dtype, eps=1e-5):
    B, L_, _, D = pair.shape
    out = torch.empty(pair.shape, device=pair.device, dtype=out_dtype)
    _need_cuda(pair)
    check(lib.rf_sym_layernorm(ptr(pair), ptr(out), dcode(out_dtype), B, L_, D, eps, stream()), "rf_sym_layernorm")
    return out


def softmax(x, x_off, x_rs, x_cs, y, y_off, y_rs, rows, cols, scale=1.0):
    _need_cuda(x, y)
    check(lib.rf_softmax(ptr(x, x_off), x_rs, x_cs, ptr(y, y_off), dcode(y.dtype), y_rs, rows, cols, scale, stream()),
          "rf_softmax")


def softmax_batched(x, x_bs, x_rs, x_cs, y, y_bs, y_rs, rows, cols, nbatch, scale=1.0):
    _need_cuda(x, y)
    check(lib.rf_softmax_batched(ptr(x), x_bs, x_rs, x_cs, ptr(y), dcode(y.dtype), y_bs, y_rs, rows, cols, scale, nbatch,
                                 stream()), "rf_softmax_batched")


def tied_logits_softmax(q, k, b_stride, n_stride, l_stride, att, att_sym, B, H, N, L_, d_head):
    """att[b,h,i,:] = softmax_j(sum_{n,d} q k) in one launch (csrc/tied.hip); q, k: bf16 views into the projection output."""
    _need_cuda(q, k, att, att_sym)
    check(lib.rf_tied_logits_softmax(ptr(q), ptr(k), b_stride, n_stride, l_stride, ptr(att), ptr(att_sym),
                                     att_sym.shape[-1] if att_sym is not None else 0, B, H, N, L_, d_head, stream()),
          "rf_tied_logits_softmax")
    return att


def _hstrides(t):
    """(b, n, h, l) element strides of a [B, N, H, L, 32]-indexed view (head slice contiguous)."""
    if t.dim() != 5 or t.stride(4) != 1 or t.shape[4] != 32:
        raise ValueError("expected a [B, N, H, L, 32] view with a contiguous head dimension")
    return I64x4(t.stride(0), t.stride(1), t.stride(2), t.stride(3))


def tied_attention(q, k, v, out, att, w=None, qscale=1.0, att_sym=None, partial_ws=None):
    """Tied MSA-row attention core (csrc/tied.hip).  q, k, v, out: bf16 views indexed [B, N, H, L, 32] (any strides with
    a contiguous head slice); att: bf16 [B, H, L, L] (workspace + result); w: fp32 [B, H, N, L] position weights folded
    into the logits kernel (None: q already carries them); att_sym: fp32 [B, L, L, H] or None.  partial_ws: fp32 workspace
    for the contraction-split logits (L == 256); allocated here when None, pass False to force the one-pass kernel."""
    B, N, H, L_, dh = q.shape
    if partial_ws is None and L_ == 256 and N % 2 == 0:
        partial_ws = torch.empty((4 if N > 128 else 2) * B * H * L_ * L_, device=q.device, dtype=F32)
    if partial_ws is False:
        partial_ws = None
    _need_cuda(q, k, v, out, att, w, att_sym, partial_ws)
    if q.stride() != k.stride():
        raise ValueError("q and k must share their strides")
    ws = I64x3(w.stride(0), w.stride(1), w.stride(2)) if w is not None else I64x3(0, 0, 0)
    if w is not None and (w.stride(3) != 1 or tuple(w.shape) != (B, H, N, L_)):
        raise ValueError("w must be [B, H, N, L] with contiguous L")
    check(lib.rf_tied_attention(ptr(q), ptr(k), ptr(v), C.byref(_hstrides(q)), C.byref(_hstrides(v)), ptr(w), C.byref(ws),
                                float(qscale), ptr(att), ptr(att_sym), att_sym.shape[-1] if att_sym is not None else 0,
                                ptr(out), C.byref(_hstrides(out)), B, H, N, L_, dh, ptr(partial_ws),
                                partial_ws.numel() if partial_ws is not None else 0, stream()), "rf_tied_attention")
    return out


COUNTERS = {"tied_logits_long": 0}  # launches of paths a test wants to see taken


def tied_logits(q, k, att, att_sym=None, qscale=1.0):
    """Logits + softmax of the tied attention on head-major q / k views [B, N, H, L, 32] (position weights already folded
    into q): att 16-bit [B, H, L, L], att_sym fp32 [B, L, L, H] or None.  L in {512, 768, 1024} (BASELINE.json configs[3]):
    contraction-split kernel over 128-query x 256-key tiles with an fp32 workspace (csrc/tied.hip: rf_tied_logits)."""
    B, N, H, L_, dh = q.shape
    nsplit = 1
    while N // nsplit > 64 and N % (2 * nsplit) == 0:
        nsplit *= 2
    ws = torch.empty(max(nsplit, 2 if L_ == 256 else 1) * B * H * L_ * L_, device=q.device, dtype=F32)
    _need_cuda(q, k, att, att_sym, ws)
    if q.stride() != k.stride():
        raise ValueError("q and k must share their strides")
    check(lib.rf_tied_logits(ptr(q), ptr(k), C.byref(_hstrides(q)), None, C.byref(I64x3(0, 0, 0)), float(qscale), ptr(att),
                             ptr(att_sym), att_sym.shape[-1] if att_sym is not None else 0, B, H, N, L_, dh, ptr(ws), ws.numel(),
                             stream()), "rf_tied_logits")
    if L_ > 256:
        COUNTERS["tied_logits_long"] += 1
    return att


def tied_row_attention(q, k, v):
    """Functional form for the dispatcher op: q, k, v h16 [B, N, L, H, 32] -> (out [B, N, L, H*32], att_sym [B, L, L, H])."""
    B, N, L_, H, dh = q.shape
    out = torch.empty(B, N, L_, H * dh, device=q.device, dtype=q.dtype)
    att = torch.empty(B, H, L_, L_, device=q.device, dtype=q.dtype)
    sym = torch.empty(B, L_, L_, H, device=q.device, dtype=F32)
    hm = lambda t: t.permute(0, 1, 3, 2, 4)  # noqa: E731  [B, N, H, L, 32] view
    tied_attention(hm(q.contiguous()), hm(k.contiguous()), hm(v.contiguous()), hm(out.view(B, N, L_, H, dh)), att, att_sym=sym)
    return out, sym


# csrc/outer_pairs.hip ("pairs in registers": the 1024-wide block never leaves a wave's registers) is correct and ties with
# csrc/outer.hip (column split, block through LDS) with the consumer's LayerNorm fused (371 vs 355-367 us) and loses without it
# (370 vs 327 us): both are bound by the rate at which a CU issues its global_load_lds pieces, not by the matrix pipe
# (DESIGN.md section 5).  Opt-in: RF_OUTER_PAIRS=1.
OUTER_PAIRS = bool(int(__import__("os").environ.get("RF_OUTER_PAIRS", "0")))


def outer_fused(xt, yt, wprime, s, c, out, eps, ln2=None):
    """Fused OuterProductMean core: xt, yt 16-bit [B, L, 32, N]; wprime from outer_fold -- step-major [32, 18, 64, 8]
    (csrc/outer_pairs.hip: pairs in registers, opt-in) or chunk-major [16, Dout, 64] (csrc/outer.hip, the default); out fp32 [B,L,L,Dout].
    ln2 = (gamma, beta, eps, y, y_ld): also apply LayerNorm over Dout and write 16-bit y[(b,i,j) * y_ld + o] instead of `out`."""
    B, L_, P, N = xt.shape
    _need_cuda(xt, yt, wprime, s, c, out)
    g2 = b2 = y = None
    eps2, y_ld = 0.0, 0
    if ln2 is not None:
        g2, b2, eps2, y, y_ld = ln2
        _need_cuda(g2, b2, y)
    if wprime.dim() == 4 and tuple(wprime.shape) == (32, 18, 64, 8):
        check(lib.rf_outer_product_pairs(ptr(xt), ptr(yt), ptr(wprime), ptr(s), ptr(c), ptr(out), B, L_, N, P, 288,
                                         float(eps), ptr(g2), ptr(b2), float(eps2), ptr(y), int(y_ld), stream()),
              "rf_outer_product_pairs")
        return y if ln2 is not None else out
    if wprime.dim() != 3 or wprime.shape[0] != 16 or wprime.shape[2] != 64:
        raise ValueError("outer_fused: wprime must come from outer_fold (step-major [32, 18, 64, 8] or chunk-major [16, Dout, 64])")
    check(lib.rf_outer_product_ln_linear(ptr(xt), ptr(yt), ptr(wprime), ptr(s), ptr(c), ptr(out), B, L_, N, P, wprime.shape[1],
                                         float(eps), ptr(g2), ptr(b2), float(eps2), ptr(y), int(y_ld), stream()),
          "rf_outer_product_ln_linear")
    return y if ln2 is not None else out


def outer_fold(w, gamma, beta, bias, dtype=None, pairs=None):
    """(W * gamma in the 16-bit type in kernel layout, its fp32 row sums, W beta + bias): the LayerNorm(1024) affine folded into
    Linear(1024 -> Dout).  pairs (default OUTER_PAIRS, Dout = 288 only): the fragment order csrc/outer_pairs.hip streams,
    [32 v][18 o-tiles][64 lanes][8]: element e of lane 16 fq + fr = W'[16 ot + fr][(16 (e >> 2) + 4 fq + (e & 3)) * 32 + v]."""
    wp = (w.float() * gamma.float()[None, :]).to(dtype or h16()).contiguous()
    s = wp.float().sum(1).contiguous()
    Dout = wp.shape[0]
    c = (w.float() @ beta.float() + bias.float()).contiguous()
    if (OUTER_PAIRS if pairs is None else pairs) and Dout == 288:
        # W'[o = (ot, fr)][k = (u = (hsel, fq, jj), v)] -> [v][ot][fq][fr][hsel][jj]
        wq = wp.view(18, 16, 2, 4, 4, 32).permute(5, 0, 3, 1, 2, 4).contiguous().view(32, 18, 64, 8)
        return wq, s, c
    # chunk-major [16 chunks = (ug, vg)][Dout][64 = (uu, vv)] with feature k = (8 ug + uu) * 32 + 8 vg + vv: the 64 features a
    # chunk of csrc/outer.hip contracts over are one 128-byte line per output column
    wpc = wp.view(Dout, 4, 8, 4, 8).permute(1, 3, 0, 2, 4).contiguous().view(16, Dout, 64)
    return wpc, s, c


def outer_product_ln_linear(x, y, gamma, beta, w, b, eps):
    """Functional form (dispatcher op): x, y bf16 [B, N, L, 32] -> fp32 [B, L, L, Dout]."""
    B, N, L_, P = x.shape
    xt = torch.empty(B, L_, P, N, device=x.device, dtype=x.dtype)
    yt = torch.empty(B, L_, P, N, device=x.device, dtype=x.dtype)
    for src, dst in ((x, xt), (y, yt)):
        copy4d(src.contiguous(), (N * L_ * P, P, 1, L_ * P), dst, (L_ * P * N, P * N, N, 1), (B, L_, P, N))
    wp, s, c = outer_fold(w, gamma, beta, b, x.dtype)
    out = torch.empty(B, L_, L_, w.shape[0], device=x.device, dtype=F32)
    return outer_fused(xt, yt, wp, s, c, out, eps)


def poswise_collapsed(xn, u, scale):
    """w[b,h,n,l] = softmax_n(scale * xn[b,n,l,:] . u[b,l,h,:]); xn bf16 [B,N,L,D], u bf16 [B,L,H,D] -> fp32 [B,H,N,L]."""
    B, N, L_, D = xn.shape
    H = u.shape[2]
    _need_cuda(xn, u)
    w = torch.empty(B, H, N, L_, device=xn.device, dtype=F32)
    check(lib.rf_poswise_collapsed(ptr(xn), ptr(u), ptr(w), B, N, L_, D, H, float(scale), stream()), "rf_poswise_collapsed")
    return w


def tied_softmax(logits, att, att_sym=None, sym_ld=0):
    B, H, L_, _ = logits.shape
    _need_cuda(logits, att, att_sym)
    check(lib.rf_tied_softmax(ptr(logits), ptr(att), dcode(att.dtype), ptr(att_sym), sym_ld, B, H, L_, stream()),
          "rf_tied_softmax")


def poswise(q0, q0_ld, k, k_ld, k_col0, k_hs, dlen, w, q_scale, qs_ld, qs_col0, qs_dh, B, N, L_, H, scale, qscale):
    _need_cuda(q0, k, w, q_scale)
    check(lib.rf_poswise(ptr(q0), dcode(q0.dtype), q0_ld, ptr(k), k_ld, k_col0, k_hs, dlen, ptr(w), ptr(q_scale),
                         qs_ld, qs_col0, qs_dh, dcode(k.dtype), B, N, L_, H, scale, qscale, stream()), "rf_poswise")


def weighted_msa_sum(x, w, y, y_ld):
    B, N, L_, D = x.shape
    _need_cuda(x, w, y)
    check(lib.rf_weighted_msa_sum(ptr(x), dcode(x.dtype), ptr(w), ptr(y), y_ld, B, N, L_, D, stream()),
          "rf_weighted_msa_sum")


def instnorm(x, gamma, beta, *, eps=1e-6, residual=None, act=L.ACT_NONE, out_dtype=None, out2_dtype=None, row_group=None,
             rows_global=None, out=None, out2=None):
    """InstanceNorm2d(affine) over NHWC x [B,H,W,C]; returns (y, y2) where y2 is an optional second copy.
    row_group / rows_global: x is a block of H of the picture's rows_global rows, the other blocks live on the other ranks of
    the torch.distributed group: the per-(b, c) sums are all-reduced (2*B*C doubles) before they are applied."""
    B, H, W, Cc = x.shape
    _need_cuda(x, gamma, beta, residual)
    sums = torch.empty(B * Cc * 2, device=x.device, dtype=torch.float64)  # fully written by the ordered finalize step
    ws_bytes = int(lib.rf_instnorm_ws_bytes(B, H * W, Cc))  # atomics-free, bitwise reproducible statistics
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    check(lib.rf_instnorm_stats(ptr(x), dcode(x.dtype), ptr(sums), B, H * W, Cc, ptr(ws), ws_bytes, stream()),
          "rf_instnorm_stats")
    if row_group is not None:
        from . import shard
        shard.all_reduce_sum(sums, row_group)
        # rf_instnorm_apply divides by ITS pixel count: hand it global sums scaled to the local block (exact up to one rounding)
        sums.mul_(float(H) / float(rows_global))
    # out / out2: caller-owned contiguous destinations (e.g. the interior of a pre-haloed picture, shard.haloed_buffer)
    y = out if out is not None else torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    y2 = out2 if out2 is not None else (torch.empty(x.shape, device=x.device, dtype=out2_dtype) if out2_dtype is not None else None)
    for t in (y, y2):
        if t is not None and (tuple(t.shape) != tuple(x.shape) or not t.is_contiguous()):
            raise ValueError("instnorm: out / out2 must be contiguous tensors of the input's shape")
    _need_cuda(y, y2)
    check(lib.rf_instnorm_apply(ptr(x), dcode(x.dtype), ptr(sums), ptr(gamma), ptr(beta), eps, ptr(residual), act,
                                ptr(y), dcode(y.dtype), ptr(y2), dcode(y2.dtype) if y2 is not None else 0, B, H * W,
                                Cc, stream()), "rf_instnorm_apply")
    return y, y2


def center_channels(x, out=None):
    """y[b, i, j, c] = x[b, i, j, c] - mean_{i,j} x[b, :, :, c]  (fp32 NHWC, in place by default): rf_instnorm_stats + the
    centre-only form of rf_instnorm_apply (gamma = beta = NULL).  Exact in front of anything an InstanceNorm follows through
    per-channel-linear maps (1x1 convolution -> InstanceNorm is invariant to a per-channel constant of its input)."""
    B, H, W, Cc = x.shape
    _need_cuda(x, out)
    sums = torch.empty(B * Cc * 2, device=x.device, dtype=torch.float64)
    ws_bytes = int(lib.rf_instnorm_ws_bytes(B, H * W, Cc))
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    check(lib.rf_instnorm_stats(ptr(x), dcode(x.dtype), ptr(sums), B, H * W, Cc, ptr(ws), ws_bytes, stream()), "rf_instnorm_stats")
    y = x if out is None else out
    check(lib.rf_instnorm_apply(ptr(x), dcode(x.dtype), ptr(sums), None, None, 0.0, None, L.ACT_NONE, ptr(y), dcode(y.dtype), None, 0,
                                B, H * W, Cc, stream()), "rf_instnorm_apply")
    return y


def channel_mean(x, row_group=None, rows_global=None):
    """fp32 [B, C] mean over the picture of NHWC x (rf_instnorm_stats + rf_instnorm_mean; row_group / rows_global as in
    instnorm: x is a block of rows of a sharded picture, the sums are all-reduced)."""
    B, H, W, Cc = x.shape
    _need_cuda(x)
    if row_group is None and x.dtype == F32 and Cc % 4 == 0 and Cc <= 1024 and x.is_contiguous():
        ws_bytes = int(lib.rf_channel_mean_ws_bytes(B, H * W, Cc))
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        mean = torch.empty(B, Cc, device=x.device, dtype=F32)
        check(lib.rf_channel_mean(ptr(x), ptr(mean), B, H * W, Cc, ptr(ws), ws_bytes, stream()), "rf_channel_mean")
        return mean
    sums = torch.empty(B * Cc * 2, device=x.device, dtype=torch.float64)
    ws_bytes = int(lib.rf_instnorm_ws_bytes(B, H * W, Cc))
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    check(lib.rf_instnorm_stats(ptr(x), dcode(x.dtype), ptr(sums), B, H * W, Cc, ptr(ws), ws_bytes, stream()), "rf_instnorm_stats")
    count = H * W
    if row_group is not None:
        from . import shard
        shard.all_reduce_sum(sums, row_group)
        count = rows_global * W
    mean = torch.empty(B, Cc, device=x.device, dtype=F32)
    check(lib.rf_instnorm_mean(ptr(sums), ptr(mean), B, count, Cc, stream()), "rf_instnorm_mean")
    return mean


def sample_mean(x, nsample=512):
    """fp32 [B, C] estimate of the per-channel mean of x [B, ..., C] (fp32 / 16-bit, contiguous) from nsample evenly spaced rows."""
    B, Cc = x.shape[0], x.shape[-1]
    _need_cuda(x)
    if not x.is_contiguous():
        raise ValueError("sample_mean: contiguous input")
    mean = torch.empty(B, Cc, device=x.device, dtype=F32)
    check(lib.rf_sample_mean(ptr(x), dcode(x.dtype), ptr(mean), B, x.numel() // (B * Cc), Cc, nsample, stream()), "rf_sample_mean")
    return mean


def center_apply(x, mean, out=None, out_dtype=None):
    """out[b, ..., c] = x[b, ..., c] - mean[b, c]; x fp32 [B, ..., C] contiguous, out fp32 (default: in place) or 16-bit."""
    B, Cc = x.shape[0], x.shape[-1]
    y = out if out is not None else (x if out_dtype in (None, F32) else torch.empty(x.shape, device=x.device, dtype=out_dtype))
    _need_cuda(x, mean, y)
    if x.dtype != F32 or not x.is_contiguous() or not y.is_contiguous() or tuple(y.shape) != tuple(x.shape) or tuple(mean.shape) != (B, Cc):
        raise ValueError("center_apply: x fp32 contiguous, out of x's shape and contiguous, mean [B, C]")
    check(lib.rf_center_apply(ptr(x), ptr(mean), ptr(y), dcode(y.dtype), B, x.numel() // (B * Cc), Cc, stream()), "rf_center_apply")
    return y


def center_rows(x):
    """x fp32 [B, R, C] (small): x -= mean over R, in place; returns the fp32 [B, C] mean."""
    B, R, Cc = x.shape
    _need_cuda(x)
    if x.dtype != F32 or not x.is_contiguous():
        raise ValueError("center_rows: fp32 contiguous [B, R, C]")
    mean = torch.empty(B, Cc, device=x.device, dtype=F32)
    check(lib.rf_center_rows(ptr(x), ptr(mean), B, R, Cc, stream()), "rf_center_rows")
    return mean


def fold_mean(w, mean, bias=None, *, k0=0, nseg=1, seg_stride=0, sum_seg=True):
    """The constant's way through a weight matrix (rf_fold_mean): w fp32 [N, ldw], mean fp32 [B, K] ->
    sum_seg: [B, N] = bias + sum_s w[:, k0 + s*seg_stride : +K] @ mean ; else [B, nseg, N], one product per segment."""
    N, ldw = w.shape
    B, K = mean.shape
    _need_cuda(w, mean, bias)
    if w.dtype != F32 or mean.dtype != F32 or not w.is_contiguous() or not mean.is_contiguous():
        raise ValueError("fold_mean: fp32 contiguous operands")
    out = torch.empty((B, N) if sum_seg else (B, nseg, N), device=w.device, dtype=F32)
    check(lib.rf_fold_mean(ptr(w), ldw, k0, K, nseg, seg_stride, 1 if sum_seg else 0, ptr(mean), ptr(bias), ptr(out), B, N,
                           stream()), "rf_fold_mean")
    return out


def conv3x3_border_fix(y, taps, dilation=1, edges=15):
    """y NHWC (fp32 / 16-bit, in place) -= the taps [B, 9, C] that fall outside the picture (rf_conv3x3_border_fix)."""
    B, H, W, Cc = y.shape
    _need_cuda(y, taps)
    if not y.is_contiguous() or tuple(taps.shape) != (B, 9, Cc) or taps.dtype != F32 or not taps.is_contiguous():
        raise ValueError("conv3x3_border_fix: y contiguous NHWC, taps fp32 [B, 9, C]")
    check(lib.rf_conv3x3_border_fix(ptr(y), dcode(y.dtype), ptr(taps), B, H, W, Cc, dilation, edges, stream()), "rf_conv3x3_border_fix")
    return y


# --------------------------------------------------------------------------------------------- misc
def msa_embed(msa, aa_idx, emb, pe, qenc):
    B, N, L_ = msa.shape
    D = emb.shape[1]
    y = torch.empty(B, N, L_, D, device=msa.device, dtype=F32)
    _need_cuda(msa, aa_idx, emb, pe, qenc)
    check(lib.rf_msa_embed(ptr(msa), ptr(aa_idx), ptr(emb), ptr(pe), ptr(qenc), ptr(y), B, N, L_, D, stream()),
          "rf_msa_embed")
    return y


def pair_embed(seq, aa_idx, tl, tr, wsep, bias, pe):
    B, L_ = seq.shape
    D = tl.shape[1]
    y = torch.empty(B, L_, L_, D, device=seq.device, dtype=F32)
    _need_cuda(seq, aa_idx, tl, tr, wsep, bias, pe)
    check(lib.rf_pair_embed(ptr(seq), ptr(aa_idx), ptr(tl), ptr(tr), ptr(wsep), ptr(bias), ptr(pe), ptr(y), B, L_, D,
                            stream()), "rf_pair_embed")
    return y


def copy4d(x, xs, y, ys, dims, x_off=0, y_off=0):
    _need_cuda(x, y)
    check(lib.rf_copy4d(ptr(x, x_off), dcode(x.dtype), _i4(xs), ptr(y, y_off), dcode(y.dtype), _i4(ys), _i4(dims),
                        stream()), "rf_copy4d")
    return y


def cast(x, dtype):
    """contiguous copy of x in another dtype."""
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    axpby(x, 1.0, None, 0.0, y)
    return y


def axpby(x, a, z, b, y):
    _need_cuda(x, z, y)
    check(lib.rf_axpby(ptr(x), dcode(x.dtype), a, ptr(z), dcode(z.dtype) if z is not None else 0, b, ptr(y),
                       dcode(y.dtype), x.numel(), stream()), "rf_axpby")
    return y


def favor_softmax_features(dash, x, x_off, xs, n1, n2, S, n, m, m_pad, dh, is_query, transposed, eps=1e-4):
    _need_cuda(dash, x)
    check(lib.rf_favor_softmax_features(ptr(dash), ptr(x, x_off), _i4(xs), n1, n2, ptr(dash), dcode(dash.dtype), S, n,
                                        m, m_pad, dh, is_query, transposed, eps, stream()),
          "rf_favor_softmax_features")


def linattn_normalize(num, num_ld, y, y_ld, rows, dh):
    _need_cuda(num, y)
    check(lib.rf_linattn_normalize(ptr(num), num_ld, ptr(y), dcode(y.dtype), y_ld, rows, dh, stream()),
          "rf_linattn_normalize")


def tile_1d_feats(msa1d, feat, feat_ld, c0, B, L_, P2):
    _need_cuda(msa1d, feat)
    check(lib.rf_tile_1d_feats(ptr(msa1d), ptr(feat), dcode(feat.dtype), feat_ld, c0, B, L_, P2, stream()),
          "rf_tile_1d_feats")


def graph_attention(q, k, v, e, out, B, L_, H, d, scale, dropout=None):
    """dropout = (p, seed, offset): the training-mode form (att_dropout on the probabilities, rf.py:658)."""
    _need_cuda(q, k, v, e, out)
    if dropout is not None:
        pd, seed, off = dropout
        check(lib.rf_graph_attention_dropout(ptr(q), ptr(k), ptr(v), ptr(e), dcode(q.dtype), ptr(out), B, L_, H, d, scale,
                                             float(pd), int(seed), int(off), stream()), "rf_graph_attention_dropout")
        return
    check(lib.rf_graph_attention(ptr(q), ptr(k), ptr(v), ptr(e), dcode(q.dtype), ptr(out), B, L_, H, d, scale,
                                 stream()), "rf_graph_attention")


def dropout(x, p, seed, offset, out=None):
    """nn.Dropout of the training-mode forward (rf_dropout): out[e] = keep ? x[e] / (1 - p) : 0 with the Philox mask
    (seed, offset); in place by default.  fp32 / 16-bit contiguous tensors."""
    y = x if out is None else out
    _need_cuda(x, y)
    if not (x.is_contiguous() and y.is_contiguous()) or x.dtype != y.dtype:
        raise ValueError("dropout: contiguous tensors of one dtype")
    check(lib.rf_dropout(ptr(x), ptr(y), dcode(x.dtype), float(p), int(seed), int(offset), x.numel(), stream()), "rf_dropout")
    return y


def dist_masked_attention(q, k, xyz, bins, att, B, L_, H, dq):
    _need_cuda(q, k, xyz, bins, att)
    check(lib.rf_dist_masked_attention(ptr(q), ptr(k), ptr(xyz), ptr(bins), ptr(att), dcode(att.dtype), B, L_, H, dq,
                                       stream()), "rf_dist_masked_attention")


# --------------------------------------------------------------------------------------------- SE(3)
def knn_mask(xyz, aa_idx, k, kmin=9):
    B, L_ = xyz.shape[:2]
    mask = torch.empty(B, L_, L_, device=xyz.device, dtype=torch.uint8)
    _need_cuda(xyz, aa_idx)
    check(lib.rf_knn_mask(ptr(xyz), ptr(aa_idx), ptr(mask), B, L_, k, kmin, stream()), "rf_knn_mask")
    return mask


def edges_from_mask(mask, capacity, zero_tail=False):
    """count = [min(edges, capacity), edges]; src/dst beyond count[0] are never read by the consumers (zero_tail: zeroed
    instead of left uninitialised -- the functional dispatcher op returns fully defined tensors)."""
    B, L_, _ = mask.shape
    dev = mask.device
    _need_cuda(mask)
    mk = (lambda n: zeros(n, device=dev, dtype=torch.int32)) if zero_tail else (lambda n: torch.empty(n, device=dev, dtype=torch.int32))
    src = mk(capacity)
    dst = mk(capacity)
    eid = torch.empty(B, L_, L_, device=dev, dtype=torch.int32)
    count = torch.empty(2, device=dev, dtype=torch.int32)
    ws = torch.empty(2 * B * L_, device=dev, dtype=torch.int32)
    check(lib.rf_edges_from_mask(ptr(mask), ptr(src), ptr(dst), ptr(eid), ptr(count), ptr(ws), B, L_, capacity, stream()),
          "rf_edges_from_mask")
    return src, dst, eid, count


def se3_edge_geometry(xyz, edge_emb, src, dst, count, capacity):
    B, L_ = xyz.shape[:2]
    de = edge_emb.shape[-1]
    basis = torch.empty(capacity, 34, device=xyz.device, dtype=F32)
    feat = torch.empty(capacity, de + 1, device=xyz.device, dtype=F32)
    _need_cuda(xyz, edge_emb)
    check(lib.rf_se3_edge_geometry(ptr(xyz), ptr(edge_emb), ptr(src), ptr(dst), ptr(count), ptr(basis), ptr(feat),
                                   de + 1, L_, de, capacity, stream()), "rf_se3_edge_geometry")
    return basis, feat


def se3_message(R0, R1, basis, h0, h1, src, count, mo, dout, mi0, mi1, capacity):
    msg = torch.empty(capacity, mo, 2 * dout + 1, device=basis.device, dtype=F32)  # rows >= count are never read
    check(lib.rf_se3_message(ptr(R0), ptr(R1), ptr(basis), ptr(h0), ptr(h1), ptr(src), ptr(count), ptr(msg), mo, dout,
                             mi0, mi1, capacity, stream()), "rf_se3_message")
    return msg


def se3_radial_message_supported(mo, dout, mi0, mi1, ki):
    return bool(lib.rf_se3_radial_message_supported(int(mo), int(dout), int(mi0), int(mi1), int(ki)))


def se3_radial_message(feat, ki, net0, net1, basis, h0, h1, src, count, mo, dout, mi0, mi1, eps, capacity, zero_tail=False):
    """Fused radial MLP + message (csrc/se3.hip: rf_se3_radial_message): feat fp32 [capacity, ld] = [edge embedding | r];
    net_di = packed fp32 parameters of net (di, dout) (layout: include/rfmi.h).  Hidden vectors and radial outputs never exist."""
    _need_cuda(feat, net0, net1, basis, h0, h1, src, count)
    for t in (feat, net0, net1, h0, h1):
        if t is not None and t.dtype != F32:
            raise TypeError("se3_radial_message: fp32 operands")
    for t in (net0, net1, h0, h1):
        if t is not None and not t.is_contiguous():
            raise TypeError("se3_radial_message: contiguous operands")
    # rows >= count are never read (zero_tail: defined anyway, for the functional dispatcher op)
    msg = (zeros if zero_tail else torch.empty)(capacity, mo, 2 * dout + 1, device=basis.device, dtype=F32)
    check(lib.rf_se3_radial_message(ptr(feat), feat.stride(0), ki, ptr(net0), ptr(net1), ptr(basis), ptr(h0), ptr(h1), ptr(src),
                                    ptr(count), ptr(msg), mo, dout, mi0, mi1, float(eps), capacity, stream()),
          "rf_se3_radial_message")
    return msg


def se3_attention(k0, k1, q0, q1, v0, v1, eid, heads, mk0, mk1, mv0, mv1, V, L_, skip0=None, skip1=None):
    """out_d [V, mv_d (+ skip channels), 2d+1]: the attention result in the leading channels; with skip_d the node's
    input features are appended behind it (GCat, ea/modules.py:903-928) -- one buffer, no concatenation pass."""
    dev = k0.device
    c0 = mv0 + (skip0.shape[1] if skip0 is not None else 0)
    c1 = mv1 + (skip1.shape[1] if skip1 is not None else 0)
    out0 = torch.empty(V, c0, 1, device=dev, dtype=F32)
    out1 = torch.empty(V, c1, 3, device=dev, dtype=F32)
    _need_cuda(k0, q0, v0, eid)
    check(lib.rf_se3_attention(ptr(k0), ptr(k1), ptr(q0), ptr(q1), ptr(v0), ptr(v1), ptr(eid), ptr(out0), ptr(out1),
                               heads, mk0, mk1, mv0, mv1, V, L_, c0, 3 * c1, stream()), "rf_se3_attention")
    if skip0 is not None:
        m = skip0.shape[1]
        copy4d(skip0, (0, 0, m, 1), out0, (0, 0, c0, 1), (1, 1, V, m), y_off=mv0)
    if skip1 is not None:
        m = skip1.shape[1] * 3
        copy4d(skip1, (0, 0, m, 1), out1, (0, 0, 3 * c1, 1), (1, 1, V, m), y_off=3 * mv1)
    return out0, out1


def se3_norm_bias(v, bias, deg):
    V, m, _ = v.shape
    y = torch.empty_like(v)
    check(lib.rf_se3_norm_bias(ptr(v), ptr(bias), ptr(y), V, m, deg, stream()), "rf_se3_norm_bias")
    return y


def se3_gram(v, deg):
    V, m, _ = v.shape
    s = torch.empty(V, m * m, device=v.device, dtype=F32)
    check(lib.rf_se3_gram(ptr(v), ptr(s), V, m, deg, stream()), "rf_se3_gram")
    return s


def se3_attn_apply(att, x, m_out, deg):
    V, m_in, nc = x.shape
    y = torch.empty(V, m_out, nc, device=x.device, dtype=F32)
    check(lib.rf_se3_attn_apply(ptr(att), ptr(x), ptr(y), V, m_out, m_in, deg, stream()), "rf_se3_attn_apply")
    return y


def coord_apply(xyz, disp):
    out = torch.empty_like(xyz)
    check(lib.rf_coord_apply(ptr(xyz), ptr(disp), ptr(out), xyz.shape[0] * xyz.shape[1], stream()), "rf_coord_apply")
    return out


def center_ca(xyz):
    y = torch.empty_like(xyz)
    check(lib.rf_center_ca(ptr(xyz), ptr(y), xyz.shape[0] * xyz.shape[1], stream()), "rf_center_ca")
    return y


def scale_rows(x, w, rows, D, out=None):
    """out[r, :] = x[r, :] * w[r]  (out defaults to a fresh tensor; pass out=x for in place)."""
    if out is None:
        out = torch.empty_like(x)
    _need_cuda(x, w, out)
    check(lib.rf_scale_rows(ptr(x), ptr(out), dcode(x.dtype), ptr(w), rows, D, stream()), "rf_scale_rows")
    return out


def fill(y, value=0.0):
    """y[...] = value through rf_fill (fp32 / bf16 contiguous tensors; int32 tensors are zero-filled as fp32 words)."""
    _need_cuda(y)
    if not y.is_contiguous():
        raise ValueError("fill: contiguous tensors only")
    if y.dtype == torch.int32:
        if value != 0:
            raise ValueError("fill: int32 tensors can only be zeroed")
        code = L.RF_F32
    else:
        code = dcode(y.dtype)
    check(lib.rf_fill(ptr(y), code, float(value), y.numel(), stream()), "rf_fill")
    return y


def zeros(*shape, device, dtype):
    return fill(torch.empty(*shape, device=device, dtype=dtype), 0.0)


def check_inputs(msa, seq, aa_idx, d_input, max_len):
    """One launch + one 12-byte read-back: (token out of range, aa_idx out of range, aa_idx not strictly increasing)."""
    ref = aa_idx if aa_idx is not None else (msa if msa is not None else seq)
    _need_cuda(msa, seq, aa_idx)
    flags = zeros(4, device=ref.device, dtype=torch.int32)
    L_ = aa_idx.shape[-1] if aa_idx is not None else 1
    check(lib.rf_check_inputs(ptr(msa), msa.numel() if msa is not None else 0, ptr(seq), seq.numel() if seq is not None else 0,
                              ptr(aa_idx), aa_idx.numel() if aa_idx is not None else 0, L_, d_input, max_len, ptr(flags),
                              stream()), "rf_check_inputs")
    f = flags.tolist()
    return bool(f[0]), bool(f[1]), bool(f[2])


def onehot(idx, n_classes, out=None, out_ld=None, col0=0, dtype=F32):
    rows = idx.numel()
    if out is None:
        out = torch.empty(*idx.shape, n_classes, device=idx.device, dtype=dtype)
        out_ld = n_classes
    _need_cuda(idx, out)
    check(lib.rf_onehot(ptr(idx), ptr(out), dcode(out.dtype), out_ld, col0, n_classes, rows, stream()), "rf_onehot")
    return out


def seqsep_feature(aa_idx, out, out_ld, col):
    B, L_ = aa_idx.shape
    _need_cuda(aa_idx, out)
    check(lib.rf_seqsep_feature(ptr(aa_idx), ptr(out), dcode(out.dtype), out_ld, col, B, L_, stream()), "rf_seqsep_feature")
    return out


def add_pos_enc(x, aa_idx, pe, two_d):
    B, N, L_, D = x.shape
    y = torch.empty_like(x)
    _need_cuda(x, aa_idx, pe)
    check(lib.rf_add_pos_enc(ptr(x), ptr(aa_idx), ptr(pe), ptr(y), B, N, L_, D, 1 if two_d else 0, stream()), "rf_add_pos_enc")
    return y


def favor_attention(qkv, pc, out, x_strides, o_strides, q_off, k_off, v_off, n_b, n_o, n_h, seq_len, dim_head,
                    n_features, softmax_kernel, eps):
    _need_cuda(qkv, pc, out)
    if not (is_h16(qkv.dtype) and pc.dtype == qkv.dtype and out.dtype == qkv.dtype):
        raise TypeError("favor_attention is the 16-bit MFMA path (q|k|v, projection and output in the library's 16-bit type)")
    xs = L.I64x4(*[int(v) for v in x_strides])
    os_ = L.I64x3(*[int(v) for v in o_strides])
    check(lib.rf_favor_attention(ptr(qkv), ptr(pc), ptr(out), C.byref(xs), C.byref(os_), q_off, k_off, v_off, n_b, n_o,
                                 n_h, seq_len, dim_head, n_features, 1 if softmax_kernel else 0, eps, stream()),
          "rf_favor_attention")
    return out


# Fused residual + next-LayerNorm epilogue of the persistent GEMM (csrc/gemm_fast.hip, round 3: row-contiguous register image,
# one barrier per tile): the 288-wide pair rows (256 x 288 tiles) and the 384-wide MSA rows (128 x 384 tiles).  RF_NO_FUSED_LN=1
# restores the separate rf_layernorm launch for A/B timing.
FUSE_LN = not bool(int(__import__("os").environ.get("RF_NO_FUSED_LN", "0")))
# the generic tile kernel's fused form (any N <= 384 whose rows fit one tile; slower than GEMM + rf_layernorm on the forward's
# shapes, so opt-in: tests/test_kernels_gpu.py exercises it)
FUSE_LN_ANY = False


def ffn_pack(w1, w2, dtype=None):
    """Both feed-forward weight matrices (w1 [hidden, D], w2 [D, hidden], nn.Linear layout) in the fragment order
    rf_ffn_fused streams them (include/rfmi.h): per chunk of 32 hidden units D/16 one-KB pieces of W1, then D/16 of W2."""
    hid, D = w1.shape
    if w2.shape != (D, hid) or D % 32 or hid % 32:
        raise ValueError(f"ffn_pack: w1 {tuple(w1.shape)} / w2 {tuple(w2.shape)}")
    NC, KS, NT = hid // 32, D // 32, D // 16
    w1 = w1.detach().float().view(NC, 2, 16, KS, 4, 8).permute(0, 3, 1, 4, 2, 5)      # (c, ks, ht, fq, fr, j)
    w2 = w2.detach().float()
    NTH = NT // 2
    if NTH % 2 == 0:
        # paired column order of the kernel (csrc/ffn.hip: PAIRED): within a wave's half of the columns, MFMA tile i, row r holds
        # output column 32 (i >> 1) + 8 (r >> 2) + 4 (i & 1) + (r & 3) -- a lane's values of tiles 2 q, 2 q + 1 are 8 consecutive columns
        i = torch.arange(NTH).view(NTH, 1)
        r = torch.arange(16).view(1, 16)
        col = 32 * (i // 2) + 8 * (r // 4) + 4 * (i % 2) + (r % 4)                    # [NTH, 16]
        rows = torch.cat([half * NTH * 16 + col.reshape(-1) for half in range(2)]).to(w2.device)
        w2 = w2[rows]
    w2 = w2.view(NT, 16, NC, 2, 4, 4).permute(2, 0, 4, 1, 3, 5)      # (c, nt, fq, fr, j >> 2, j & 3)
    packed = torch.cat([w1.reshape(NC, -1), w2.reshape(NC, -1)], 1)
    return packed.to(dtype if dtype is not None else h16()).contiguous()


# One-launch feed-forward (csrc/ffn.hip).  At the forward's shapes it ties with the two-GEMM path on time (402 vs 397-417 us per MSA
# block, 511-525 vs 485-518 us per pair block; 315.8-317.0 vs 316.6-317.4 ms per forward on the same box) -- the weight stream of a
# 128-token tile through L2 -> LDS costs what the hidden round trip through HBM costs -- but it takes 0.15 TB of HBM traffic out of
# every forward (the hidden activations never exist in memory), so it is the default; RF_FUSED_FFN=0 restores the two GEMMs
# (DESIGN.md section 5 "fused feed-forward").
# RF_FUSED_FFN: 1 (default: both widths) | 0 (off) | 384 | 288 (only the feed-forward blocks of that model width)
_ffn_env = __import__("os").environ.get("RF_FUSED_FFN", "1").strip()
FUSE_FFN = _ffn_env not in ("", "0")
FUSE_FFN_WIDTHS = (int(_ffn_env),) if _ffn_env in ("288", "384") else (288, 384)


def ffn_fused_applies(xn, x_res, D, hidden):
    rows = x_res.numel() // D
    return (FUSE_FFN and is_h16(xn.dtype) and D in FUSE_FFN_WIDTHS and hidden % 32 == 0 and rows % 128 == 0 and rows >= 16384
            and x_res.is_contiguous() and xn.is_contiguous() and x_res.dtype == F32
            and 288 <= hidden and hidden * 4 <= 160 * 1024 - 153 * 1024 - 64)


def ffn_fused(xn, w_packed, b1, b2, x_res, next_ln=None):
    """x_res += W2 relu(W1 xn + b1) + b2 in one launch (csrc/ffn.hip); returns next_ln(x_res) in the 16-bit type when
    `next_ln` (an nn.LayerNorm) is given, else None."""
    D = x_res.shape[-1]
    rows = x_res.numel() // D
    hidden = b1.numel()
    _need_cuda(xn, w_packed, x_res)
    if w_packed.numel() != 2 * D * hidden or w_packed.dtype != xn.dtype:
        raise ValueError("ffn_fused: packed weights do not match (ops.ffn_pack)")
    out_ln = torch.empty(x_res.shape, device=x_res.device, dtype=xn.dtype) if next_ln is not None else None
    g = next_ln.weight.detach() if next_ln is not None else None
    b = next_ln.bias.detach() if next_ln is not None else None
    check(lib.rf_ffn_fused(ptr(xn), D, ptr(w_packed), ptr(b1), ptr(b2), ptr(x_res), D, ptr(x_res), D, ptr(out_ln), D, ptr(g), ptr(b),
                           float(next_ln.eps) if next_ln is not None else 0.0, rows, D, hidden, stream()), "rf_ffn_fused")
    return out_ln


def linear_residual_ln(x, w, bias, x_res, next_ln, xn_out=None):
    """x_res += x @ w^T + bias (fp32, in place).  If `next_ln` (an nn.LayerNorm) is given and the fused epilogue applies
    (bf16 operands, full rows per tile), also returns LayerNorm_next(x_res) in bf16 (written into `xn_out` when given: the
    row-panel callers pass a slice of one full-size tensor); otherwise returns None and the caller normalises with
    rf_layernorm."""
    N = w.shape[0]
    rows = x_res.numel() // N
    # whole rows per tile: the persistent GEMM normalises the updated rows in its epilogue (csrc/gemm_fast.hip)
    fused = (FUSE_LN and N in (288, 384) and rows % 256 == 0 and rows >= 16384) or (FUSE_LN_ANY and N <= 384 and N % 4 == 0)
    if fused and next_ln is not None and is_h16(x.dtype) and x_res.is_contiguous() and x.is_contiguous():
        xn = torch.empty(x_res.shape, device=x_res.device, dtype=x.dtype) if xn_out is None else xn_out
        linear(x, w, bias, out=x_res, residual=x_res,
               ln=(xn, next_ln.weight.detach(), next_ln.bias.detach(), next_ln.eps))
        return xn
    linear(x, w, bias, out=x_res, residual=x_res)
    return None
