/* rfmi.h -- C ABI of librfmi.so, the MI355X (gfx950) RoseTTAFold forward-path kernels.
 *
 * The reference (dohlee/rosettafold-pytorch) has no FFI / plugin interface: its only boundary
 * is the Python nn.Module call surface.  This header is the boundary the
 * build adds underneath that surface: one `extern "C"` entry point per op group of the
 * forward path, each citing the reference lines whose arithmetic it replaces.  The Python
 * host (rosettafold-pytorch_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer; the caller allocates every output and workspace;
 *   - kernels are launched on `stream` (a hipStream_t passed as void*), never allocate,
 *     never synchronise, keep no global state: re-entrant;
 *   - return value: 0 = launched; >0 = hipError_t from the launch; <0 = argument rejected
 *     (RF_EINVAL...) and nothing was launched;
 *   - dtype codes: RF_F32 = 0, RF_BF16 = 1, RF_F16 = 2.  "T" below = activation dtype chosen by the caller.
 *   - the kernel sources are built twice: librfmi.so computes its 16-bit MFMA contractions in bfloat16
 *     (v_mfma_f32_16x16x32_bf16) and accepts {RF_F32, RF_BF16}; librfmi_f16.so computes them in IEEE fp16
 *     (v_mfma_f32_16x16x32_f16: same rate and bytes, 11 significand bits instead of 8, range 65504) and accepts
 *     {RF_F32, RF_F16}.  Both export exactly the symbols declared here; rf_h16_dtype() says which 16-bit code a
 *     loaded library takes, every other 16-bit code is rejected with RF_EINVAL.  Wherever a comment below says
 *     "bf16" for an operand it means "the library's 16-bit type".
 */
#ifndef RFMI_H
#define RFMI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_F32 0
#define RF_BF16 1
#define RF_F16 2

#define RF_EINVAL (-1)   /* inconsistent sizes / unsupported combination */
#define RF_EALIGN (-2)   /* pointer or stride not aligned as the kernel needs */

#define RF_ACT_NONE 0
#define RF_ACT_RELU 1
#define RF_ACT_ELU 2
#define RF_ACT_RELU_EPS 3 /* relu(x)+eps for column < act_nvalid (row < -act_nvalid if negative), 0 beyond (FAVOR+ ReLU features) */
#define RF_ACT_LEAKY 4    /* LeakyReLU(0.01), rf_layernorm only */
#define RF_ACT_BLOCK_LN32 5 /* rf_gemm, bf16 path, 256x256 tiles: LayerNorm over every aligned 32x32 block of the output
                            * (block (i,j) = rows 32i.., cols 32j..; feature k = 32*(row%32) + col%32) with affine
                            * ln_gamma/ln_beta[1024], ln_eps; ln_out must be NULL.  The outer-product features of
                            * OuterProductMean normalised in the GEMM epilogue (rf.py:416,424-426). */

#define RF_BIAS_NONE 0
#define RF_BIAS_COL 1 /* bias[n] */
#define RF_BIAS_ROW 2 /* bias[m] */

#define RF_AMODE_PLAIN 0
#define RF_AMODE_CONV3X3 1 /* A rows are NHWC pixels, K = 9*C im2col on the fly */

/* Strided / batched / chunked GEMM:  C[z][m][n] = epilogue( alpha * sum_k A[z][m][k] * B[z][n][k] ).
 * Both operands are K-contiguous ("TN"); every contraction of the forward path is phrased
 * this way (nn.Linear rf.py:195-281, the einsums rf.py:254,257,424,592,916, conv2d rf.py:452-457,
 * resnet.py:19-38).  Element offsets (in elements of the operand dtype):
 *   batch z = (z0*nb1 + z1)*nb2 + z2
 *   A(z,m,k) = z0*a_bs[0]+z1*a_bs[1]+z2*a_bs[2] + (m / a_rc)*a_ro + (m % a_rc)*a_ri + (k / kc)*a_ko + (k % kc)
 *   B(z,n,k) = likewise with b_*            (kc = K for a plain contiguous K)
 *   C(z,m,n) = z.c_bs + (m / c_rc)*c_ro + (m % c_rc)*c_ri + (n / c_cc)*c_co + (n % c_cc)
 * a_rc/b_rc/c_rc/c_cc <= 0 mean "no split" (offset = m*..ri, n).
 * RF_AMODE_CONV3X3: A is NHWC [conv_n, conv_h, conv_w, conv_c], m = pixel index, k = tap*C + c
 * (tap = 3*(di+1)+(dj+1)), zero padding, dilation conv_dil; a_* strides are ignored.
 */
typedef struct rf_gemm_desc {
  int32_t M, N, K;
  int32_t nb0, nb1, nb2;
  int32_t ab_dtype;  /* RF_BF16 (MFMA path) or RF_F32 (exact f32 path) */
  int32_t c_dtype;   /* RF_BF16 or RF_F32 */
  int32_t kc;
  int32_t a_mode;
  int32_t a_rc, b_rc, c_rc, c_cc;
  int64_t a_bs[3], a_ro, a_ri, a_ko;
  int64_t b_bs[3], b_ro, b_ri, b_ko;
  int64_t c_bs[3], c_ro, c_ri, c_co;
  int32_t conv_n, conv_h, conv_w, conv_c, conv_dil;
  int32_t bias_mode;
  int32_t act;
  int32_t act_nvalid;
  float act_eps;
  float alpha;
  int32_t tile_cfg;  /* 0 = auto; else index into the tile table (tuning/tests) */
  const void* A;
  const void* B;
  void* C;
  const float* bias;     /* fp32 */
  const float* residual; /* fp32, same layout as C; C = residual + epilogue(...) ; may alias C */
  /* optional fused LayerNorm of the result rows (the next sub-layer's pre-norm, rf.py:341 etc.): when ln_out != NULL
   * the kernel also writes ln_out[m, :] = LayerNorm(C[m, :]) * ln_gamma + ln_beta as bf16 [M, N] (row-major).
   * Needs the bf16 path, plain fp32 C, one batch, N <= 384 (a workgroup then owns complete rows). */
  void* ln_out;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int32_t reserved_;
  /* optional row-group scale of the leading output columns, applied to the fp32 accumulators (bias included) before the
   * result is rounded:   C[m][n] *= rs_alpha * rs[(m / rs_rpb) * rs_bstride + (n / rs_cg) * rs_rpb + m % rs_rpb]   for n < rs_ncols.
   * The q | k | v projection of the tied MSA-row attention folds the position weights into q this way (rf.py:252:
   * q * w * d_head^-0.5 with w [B, H, N, L]: rs_rpb = N L, rs_bstride = H N L, rs_cg = d_head, rs_ncols = d_msa), so q is
   * rounded once, after the scaling, and the attention kernels take it as it is.  16-bit path, register-resident-weights
   * kernel only (K = 288 / 384, long M; rs_cg % 16 == 0): RF_EINVAL when rs != NULL and that kernel does not apply. */
  const float* rs;
  int64_t rs_bstride;
  int32_t rs_rpb, rs_cg, rs_ncols;
  float rs_alpha;
} rf_gemm_desc;

int rf_gemm(const rf_gemm_desc* d, void* stream);
/* Introspection for measurement tools: the kernel family the calling thread's last rf_gemm launched (0 exact fp32, 1 generic
 * bf16 tile kernel, 2 conv3x3 implicit GEMM, 3 persistent tile kernel, 4 register-resident-weights skinny-K kernel, -1 none). */
int rf_gemm_last_family(void);

/* LayerNorm over the last dim (nn.LayerNorm, rf.py:323,328,416,435,437,442,443,565,573,580,672,
 * 685,686,758,759,765,771,876,877,883,886,1136; ea/modules.py:553).  rows x D, fp32 statistics. */
int rf_layernorm(const void* x, int x_dtype, int64_t x_ld, void* y, int y_dtype, int64_t y_ld, int64_t rows,
                 int D, const float* gamma, const float* beta, float eps, int groups, int act, void* stream);
/* groups > 1: row r uses gamma/beta[(r % groups)*D ..] (the 8 radial MLPs of one SE(3) layer normalised in one
 * launch, ea/modules.py:265-275); act: RF_ACT_NONE | RF_ACT_RELU | RF_ACT_LEAKY applied after the affine. */

/* y[r, :] = 0.5*(x[b,i,j,:] + x[b,j,i,:]) normalised WITHOUT affine (shared front of the four
 * MsaUpdateWithPairLayer.pair2att of a block: Symmetrization + LayerNorm, rf.py:550-566). */
int rf_sym_layernorm(const float* pair, void* y, int y_dtype, int B, int L, int D, float eps, void* stream);

/* Row softmax with strides: for r in [0,rows): y[r*y_rs + c] = softmax_c(scale * x[r*x_rs + c*x_cs]).
 * (rf.py:215,255,569,657,914). */
int rf_softmax(const float* x, int64_t x_rs, int64_t x_cs, void* y, int y_dtype, int64_t y_rs, int64_t rows,
               int cols, float scale, void* stream);

/* nbatch such problems in one launch: problem z reads x + z*x_bs, writes y + z*y_bs (the per-(layer, head) softmaxes of one
 * MsaUpdateWithPair stack, rf.py:569). */
int rf_softmax_batched(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_cs, void* y, int y_dtype, int64_t y_bs,
                       int64_t y_rs, int64_t rows, int cols, float scale, int nbatch, void* stream);

/* Tied-attention softmax (rf.py:255,261-265): logits fp32 [B,H,L,L] -> att T [B,H,L,L] and, when
 * att_sym != NULL, the symmetrised map 0.5*(att+att^T) as fp32 [B,L,L,H] (written with ld sym_ld). */
int rf_tied_softmax(const float* logits, void* att, int att_dtype, float* att_sym, int64_t sym_ld, int B, int H,
                    int L, void* stream);

/* Tied MSA-row attention, logits + softmax in one launch (rf.py:252-255,261-265), bf16 only:
 *   att[b,h,i,:] = softmax_j( sum_{n,d} q[b,n,i,h,d] * k[b,n,j,h,d] )   -> att bf16 [B,H,L,L]
 * q / k element (b,n,l,h,d) at b*b_stride + n*n_stride + l*l_stride + h*d_head + d (q already scaled, rf.py:252);
 * d_head == 32 and L in {64,128,192,256} (RF_EINVAL otherwise: use rf_gemm + rf_tied_softmax).  att_sym as in
 * rf_tied_softmax (may be NULL). */
int rf_tied_logits_softmax(const void* q, const void* k, int64_t b_stride, int64_t n_stride, int64_t l_stride, void* att,
                           float* att_sym, int64_t sym_ld, int B, int H, int N, int L, int d_head, void* stream);

/* Attention . V of the tied attention (rf.py:257-258): out[b,n,i,h,:] = sum_j att[b,h,i,j] v[b,n,h,j,:], bf16.
 * att: [B,H,L,L]; v / out element (b,n,h,l,d) at b*s[0] + n*s[1] + h*s[2] + l*s[3] + d (the 32-wide head slice
 * contiguous; any layout works, a head-major v -- [.., L, 32] tiles contiguous -- is the fast one).  d_head == 32,
 * L in {64,128,192,256}. */
int rf_tied_av(const void* att, const void* v, const int64_t v_strides[4], void* out, const int64_t o_strides[4], int B,
               int H, int N, int L, int d_head, void* stream);

/* Logits + softmax of the tied attention on head-major operands (rf.py:252-255), the first half of rf_tied_attention:
 *   att[b,h,i,:] = softmax_j( sum_{n,d} (w[b,h,n,i] * qscale) q[b,n,h,i,d] k[b,n,h,j,d] )      (+ att_sym as in rf_tied_softmax)
 * Same arguments as rf_tied_attention.  L in {64,128,192,256}, or L in {512,768,1024} (BASELINE.json configs[3]): long rows run
 * contraction-split over 128-query x 256-key tiles, need partial_ws (>= nsplit * B*H*L*L floats, nsplit = 1 for N <= 64,
 * doubling until N / nsplit <= 64) and take w == NULL (position weights folded into q by the projection: rf_gemm_desc.rs). */
int rf_tied_logits(const void* q, const void* k, const int64_t qk_strides[4], const float* w, const int64_t w_strides[3],
                   float qscale, void* att, float* att_sym, int64_t sym_ld, int B, int H, int N, int L, int d_head,
                   float* partial_ws, int64_t partial_ws_elems, void* stream);

/* Tied MSA-row attention core in one call (rf.py:252-265): logits + softmax, optional symmetrised map, attention . V.
 *   att[b,h,i,:] = softmax_j( sum_{n,d} (w[b,h,n,i] * qscale) q[b,n,h,i,d] k[b,n,h,j,d] ),  out = att . v
 * q / k share qk_strides {b,n,h,l}; w (fp32, may be NULL = q already scaled) has strides {b,h,n} with l contiguous: the
 * position weights of rf.py:252 are applied inside the logits kernel instead of a pass over q.  att: bf16 [B,H,L,L]
 * (caller-owned workspace and result); att_sym as in rf_tied_softmax (may be NULL).
 * partial_ws (may be NULL): fp32 workspace of partial_ws_elems elements.  With L == 256 and at least
 * 2 * B*H*L*L elements (4 * B*H*L*L when N > 128) the logits run contraction-split: workgroups of 128 query rows x 256
 * keys over ranges of <= 64 MSA rows write fp32 partial logits there and a second kernel adds them and takes the row
 * softmax (0.6x the L2 -> LDS bytes of the one-pass kernel).  Without it, or for other shapes, the one-pass kernel runs. */
int rf_tied_attention(const void* q, const void* k, const void* v, const int64_t qk_strides[4], const int64_t v_strides[4],
                      const float* w, const int64_t w_strides[3], float qscale, void* att, float* att_sym, int64_t sym_ld,
                      void* out, const int64_t o_strides[4], int B, int H, int N, int L, int d_head, float* partial_ws,
                      int64_t partial_ws_elems, void* stream);

/* Feed-forward block with its residual in ONE launch (FeedForward, rf.py:270-281, inside the residual wrappers of the
 * encoder / axial layers rf.py:284-354, 483-560):
 *   out[m,:] = residual[m,:] + W2 relu(W1 x[m,:] + b1) + b2          (fp32; out may alias residual: in place)
 *   ln_out[m,:] = gamma * (out[m,:] - mean) * rstd + beta             (optional 16-bit copy: the NEXT layer's LayerNorm)
 * The hidden activations never reach HBM.  x: 16-bit [M, D] (ldx elements between rows), D = 288 or 384, hidden % 32 == 0,
 * M % 128 == 0.  w_packed: both weight matrices in the fragment order the kernel streams (16-bit, 2 * D * hidden elements):
 * for chunk c of 32 hidden units, first D/32 * 2 pieces of W1, then D/16 pieces of W2, a piece = 64 lanes x 8 elements,
 * lane = 16 fq + fr:
 *   W1 piece (ks, ht):  element j of the lane = W1[32 c + 16 ht + fr][32 ks + 8 fq + j]          (W1: [hidden, D])
 *   W2 piece (nt):      element j of the lane = W2[row(nt, fr)][32 c + 16 (j >> 2) + 4 fq + (j & 3)]  (W2: [D, hidden]) with
 *                       row(nt, fr) = 16 nt + fr for D = 288 and, for D = 384 (an even number of column tiles per wave: the lane's
 *                       values of two tiles are 8 consecutive columns, 16-byte pieces of the LayerNorm copy),
 *                       row = 192 (nt / 12) + 32 (i >> 1) + 8 (fr >> 2) + 4 (i & 1) + (fr & 3), i = nt % 12  (ops.ffn_pack)
 * (ops.ffn_pack builds it once per module).  b1 [hidden], b2 [D], gamma / beta [D]: fp32. */
int rf_ffn_fused(const void* x, int64_t ldx, const void* w_packed, const float* b1, const float* b2, const float* residual,
                 int64_t ldr, float* out, int64_t ldo, void* ln_out, int64_t ldn, const float* ln_gamma,
                 const float* ln_beta, float ln_eps, int64_t M, int D, int hidden, void* stream);

/* PositionWiseWeightFactor in collapsed form on the matrix pipe (rf.py:205-217):
 *   w[b,h,n,l] = softmax_n( scale * sum_c xn[b,n,l,c] * u[b,l,h,c] ),   u[b,l,h,:] = W_k[h*dh:(h+1)*dh, :]^T to_q(x_0)[b,l,h,:]
 * (to_k's bias is constant in n and cancels in the softmax, so the to_k projection over all N rows is never formed).
 * xn: bf16 [B,N,L,D]; u: bf16 [B,L,H,D]; w: fp32 [B,H,N,L].  H <= 16, D % 32 == 0, N % 16 == 0, N <= 256. */
int rf_poswise_collapsed(const void* xn, const void* u, float* w, int B, int N, int L, int D, int H, float scale,
                         void* stream);

/* Fused OuterProductMean (rf.py:412-427): out[b,i,j,:] = Linear(LayerNorm_1024(sum_n x[b,n,i,:] (x) y[b,n,j,:])) in one
 * kernel; the 1024-wide feature tensor never exists in HBM.  xt / yt: bf16 [B, L, 32, N] (MSA depth contiguous);
 * wprime: 16-bit [16, Dout, 64] = W * gamma (LayerNorm affine folded in) stored CHUNK-MAJOR: wprime[c][o][8 uu + vv] =
 * (W gamma)[o][k], k = (8 (c / 4) + uu) * 32 + 8 (c % 4) + vv -- the 64 features one chunk of the kernel contracts over are one
 * 128-byte line per output column; s[o] = sum_k wprime[o,k] (fp32, of the 16-bit values);
 * c[o] = sum_k W[o,k] beta[k] + bias[o]; out: fp32 [B, L, L, Dout]:
 *     out = rstd * (sum_k co_k wprime[o,k] - mean * s[o]) + c[o],   mean / rstd over the 1024 features in fp32.
 * Supported: P == 32, Dout == 288, N in {64, 128}, L % 16 == 0 (RF_EINVAL otherwise: rf_gemm with RF_ACT_BLOCK_LN32 + rf_gemm). */
int rf_outer_product_ln_linear(const void* xt, const void* yt, const void* wprime, const float* s, const float* c, float* out,
                               int B, int L, int N, int P, int Dout, float eps, const float* ln2_gamma, const float* ln2_beta,
                               float ln2_eps, void* y, int64_t y_ld, void* stream);

/* The same operator, second kernel (csrc/outer_pairs.hip: a wave owns residue pairs, the 1024-wide block never leaves its
 * registers; opt-in in the Python layer, RF_OUTER_PAIRS=1: it ties with the first kernel).  Same arguments except the weight layout: w_packed = W * gamma (16-bit) in the
 * order the kernel streams it, [32 v][18 o-tiles][64 lanes][8]: element e of lane 16 fq + fr = W'[16 ot + fr][(16 (e >> 2) +
 * 4 fq + (e & 3)) * 32 + v] (feature index k = u * 32 + v as in rf.py:416).  Dout = 288, P = 32, N in {64, 128}, L % 16 == 0.
 * y (when given): 8-byte aligned, y_ld % 4 == 0. */
int rf_outer_product_pairs(const void* xt, const void* yt, const void* w_packed, const float* s, const float* c, float* out,
                           int B, int L, int N, int P, int Dout, float eps, const float* ln2_gamma, const float* ln2_beta,
                           float ln2_eps, void* y, int64_t y_ld, void* stream);
/* Optional tail (PairUpdateWithMsa.ln_coevol_feat, rf.py:443,486): with y != NULL the kernel applies a second LayerNorm
 * (ln2_gamma / ln2_beta [Dout], ln2_eps) over the Dout outputs of every pair and writes 16-bit y[(b,i,j) * y_ld + o] INSTEAD of
 * `out` (which may then be NULL): the fp32 result and the separate LayerNorm pass over it disappear as well.  y: 16-byte
 * aligned, y_ld % 8 == 0 (rows leave as 16-byte pieces). */

/* PositionWiseWeightFactor core (rf.py:205-217): w[b,n,h,l] = softmax_n( scale * sum_{c<dlen} q0[b,l,h,c]*k[b,n,l,h,c] ).
 * q0: [B,L,H*dlen] (dtype q0_dtype, ld q0_ld); k: T rows [B,N,L,*] of ld k_ld, head h at column k_col0 + h*k_hstride.
 *   - direct form:    q0 = to_q(row 0), k = to_k(x), dlen = k_hstride = d_head;
 *   - collapsed form: q0 = to_q(row 0) W_k (one [B*L,d]x[d,d] GEMM), k = x itself, dlen = d, k_hstride = 0
 *     (the to_k GEMM over all N rows disappears; its bias is constant in n and drops out of the softmax).
 * w (fp32 [B,N,H,L]) may be NULL.  When q_scale != NULL the kernel also does the fused `q = q * w * qscale` of
 * rf.py:252 in place on q_scale: T [B,N,L,*] (ld qs_ld, head h = columns qs_col0 + h*qs_dh .. +qs_dh). */
int rf_poswise(const void* q0, int q0_dtype, int64_t q0_ld, const void* k, int64_t k_ld, int k_col0, int k_hstride,
               int dlen, float* w, void* q_scale, int64_t qs_ld, int qs_col0, int qs_dh, int dtype, int B, int N, int L,
               int H, float scale, float qscale, void* stream);

/* y[b,l,:] = sum_n w[b,n,0,l] * x[b,n,l,:]   (rf.py:723, rf.py:797); x T [B,N,L,D], y fp32 [B,L,D] (ld y_ld) */
int rf_weighted_msa_sum(const void* x, int dtype, const float* w, float* y, int64_t y_ld, int B, int N, int L, int D,
                        void* stream);

/* InstanceNorm2d(affine, eps) on NHWC (rf.py:453,457; resnet.py:29,39,63) in two steps:
 * stats: sums[b,c,0..1] = (sum, sumsq) over the L*L pixels (fully written: no zeroing needed);
 * apply: y = act( (x-mean)*rstd*gamma + beta [+ residual] ) ; act = RF_ACT_NONE | RF_ACT_ELU. */
/* workspace (MANDATORY, rf_instnorm_ws_bytes bytes): per-block partial sums reduced in a fixed order -- the library has no
 * atomic accumulation anywhere, results are bitwise reproducible run to run; NULL / too small -> RF_EINVAL. */
int64_t rf_instnorm_ws_bytes(int B, int64_t HW, int C);
int rf_instnorm_stats(const void* x, int x_dtype, void* sums /* 2*B*C doubles */, int B, int64_t HW, int C, void* workspace,
                      int64_t ws_bytes, void* stream);
/* gamma == beta == NULL: centre only, y = x - mean over the picture (no scaling; PredictionHead operand conditioning). */
int rf_instnorm_apply(const void* x, int x_dtype, const void* sums, const float* gamma, const float* beta, float eps,
                      const float* residual, int act, void* y, int y_dtype, void* y2, int y2_dtype, int B, int64_t HW,
                      int C, void* stream);

/* ---- operand conditioning of the 16-bit modes (csrc/condition.hip): exact algebra, the function computed is unchanged ----
 * At random init every stream is a large per-sample constant vector plus a position-dependent part 10-30x smaller, and the
 * InstanceNorms (rf.py:453,457; resnet.py:29,39,63) keep only the latter; an operand rounded to 16 bits WITH the constant is
 * 10-30x coarser than the information that survives.  PairUpdateWithMsa (rf.py:476-498), PredictionHead (rf.py:1130-1172)
 * and the structure track's embeddings remove the constant before the rounding and add W * constant back in fp32:
 *   W (x - m) + (b + W m) == W x + b;   conv3x3(x - m) + [taps outside the picture] == conv3x3(x) - const (InstanceNorm drops it).
 * mean[b,c] = sums[b,c,0] / HW from the sums of rf_instnorm_stats (fp32 [B,C]). */
int rf_instnorm_mean(const void* sums, float* mean, int B, int64_t HW, int C, void* stream);
/* the same mean straight from fp32 NHWC x (C % 4 == 0, 16-byte aligned): vectorised, no atomics; workspace of
 * rf_channel_mean_ws_bytes bytes (MANDATORY: per-block partial sums, added in a fixed order). */
int64_t rf_channel_mean_ws_bytes(int B, int64_t HW, int C);
int rf_channel_mean(const float* x, float* mean, int B, int64_t HW, int C, void* workspace, int64_t ws_bytes, void* stream);
/* an ESTIMATE of that mean from nsample evenly spaced rows of x [B,R,C] (fp32 or 16-bit): the identities hold for any constant,
 * the attention layers condition their value operand with it (model.py: _value_conditioning). */
int rf_sample_mean(const void* x, int x_dtype, float* mean, int B, int64_t R, int C, int nsample, void* stream);
/* y[b,p,c] = x[b,p,c] - mean[b,c]; x fp32 [B,HW,C], y fp32 or the 16-bit type (may alias x when fp32); C % 4 == 0, 16-byte
 * aligned pointers. */
int rf_center_apply(const float* x, const float* mean, void* y, int y_dtype, int B, int64_t HW, int C, void* stream);
/* small tensors: mean[b,c] = mean over the R rows of x[b] (fp32 [B,R,C]) and x -= mean, in place (one block per sample). */
int rf_center_rows(float* x, float* mean, int B, int R, int C, void* stream);
/* the constant's way through a weight matrix (w fp32 [N, ldw]; mean fp32 [B,K]; bias fp32 [N] or NULL):
 *   sum_seg != 0: out[b,n]   = bias[n] + sum_{s<nseg} sum_{k<K} w[n, k0 + s*seg_stride + k] * mean[b,k]
 *   sum_seg == 0: out[b,s,n] = bias[n] +                sum_{k<K} w[n, k0 + s*seg_stride + k] * mean[b,k]   (e.g. the 9 taps
 *   of a 3x3 convolution stored [Co, 9*Ci]: nseg 9, seg_stride Ci). */
int rf_fold_mean(const float* w, int64_t ldw, int k0, int K, int nseg, int64_t seg_stride, int sum_seg, const float* mean,
                 const float* bias, float* out, int B, int N, void* stream);
/* 3x3 'same' (zero-padded) convolution of a centred picture: y[b,i,j,:] -= sum of taps[b, kh*3+kw, :] over the taps whose
 * source pixel (i+(kh-1)d, j+(kw-1)d) lies outside the picture, so that y == conv3x3(x) - (sum of all taps) everywhere.
 * y NHWC fp32 or 16-bit, in place; taps fp32 [B,9,C] (rf_fold_mean); edges: which sides of this block are sides of the picture
 * (1 top | 2 bottom | 4 left | 8 right; 15 for a whole picture, a row block of a sharded picture has neighbours). */
int rf_conv3x3_border_fix(void* y, int y_dtype, const float* taps, int B, int H, int W, int C, int dilation, int edges,
                          void* stream);

/* MsaEmbedding (rf.py:106-120): y[b,n,l,:] = emb[msa[b,n,l]] + pe[aa_idx[b,l]] + qenc[n==0 ? 0 : 1]; fp32 out. */
int rf_msa_embed(const int64_t* msa, const int64_t* aa_idx, const float* emb, const float* pe, const float* qenc,
                 float* y, int B, int N, int L, int D, void* stream);

/* PairEmbedding (rf.py:123-181, 79-103) with the 289->d Linear factorised into two 21-row tables:
 * y[b,i,j,:] = tl[seq[b,j]] + tr[seq[b,i]] + wsep*log(|idx_i-idx_j|+1) + bias + [pe[idx_i] | pe[idx_j]]. */
int rf_pair_embed(const int64_t* seq, const int64_t* aa_idx, const float* tl, const float* tr, const float* wsep,
                  const float* bias, const float* pe, float* y, int B, int L, int D, void* stream);

/* Generic strided copy / cast / transpose (einops rearrange of the reference, e.g. rf.py:258,403,408,593):
 * y[i0,i1,i2,i3] = x[...] over a 4-D index space with element strides per side. */
int rf_copy4d(const void* x, int x_dtype, const int64_t xs[4], void* y, int y_dtype, const int64_t ys[4],
              const int64_t dims[4], void* stream);

/* y = a*x + b*z  elementwise fp32/T (n elements; z may be NULL) */
int rf_axpby(const void* x, int x_dtype, float a, const void* z, int z_dtype, float b, void* y, int y_dtype,
             int64_t n, void* stream);

/* y[r,:] = x[r,:] * w[r]  (msa_proj * position weight, rf.py:472); x, y T [rows, D]; y may alias x */
int rf_scale_rows(const void* x, void* y, int dtype, const float* w, int64_t rows, int D, void* stream);

/* y[0..n) = value (T); y 16-byte aligned.  Zero / one initialisation of caller-owned buffers without a framework kernel. */
int rf_fill(void* y, int dtype, float value, int64_t n, void* stream);

/* Input validation of RoseTTAFold.forward (the reference raises IndexError from nn.Embedding / tensor indexing,
 * rf.py:73,98,115-119,155): flags[0] = a token outside [0, d_input), flags[1] = an aa_idx outside [0, max_len),
 * flags[2] = aa_idx not strictly increasing inside a sample (the kNN edge-capacity bound then does not hold, rf.py:841-852).
 * msa / seq / aa_idx may each be NULL (skipped).  flags: 3 x int32, zeroed by the caller. */
int rf_check_inputs(const int64_t* msa, int64_t n_msa, const int64_t* seq, int64_t n_seq, const int64_t* aa_idx,
                    int64_t n_idx, int L, int d_input, int max_len, int32_t* flags, void* stream);

/* y[r, col0 + c] = (idx[r] == c), c < n_classes  (F.one_hot(seq, 21), rf.py:1276); y T with leading dim y_ld */
int rf_onehot(const int64_t* idx, void* y, int dtype, int64_t y_ld, int col0, int n_classes, int64_t rows, void* stream);

/* y[(b,i,j)*y_ld + col] = clamp(sign(d)*log(|d|+1), 0, 5.5), d = aa_idx[b,i]-aa_idx[b,j]  (rf.py:746-749) */
int rf_seqsep_feature(const int64_t* aa_idx, void* y, int dtype, int64_t y_ld, int col, int B, int L, void* stream);

/* Stand-alone positional encodings, fp32: two_d == 0: SinusoidalPositionalEncoding.forward (rf.py:72-76)
 * y[b,n,l,:] = x + pe[aa_idx[b,l]] (pe [max_len, D]); two_d != 0: SinusoidalPositionalEncoding2D.forward (rf.py:95-103)
 * y[b,i,j,:] = x + [pe[aa_idx[b,i]] | pe[aa_idx[b,j]]] (N == L, pe [max_len, D/2]).  aa_idx must be in range. */
int rf_add_pos_enc(const float* x, const int64_t* aa_idx, const float* pe, float* y, int B, int N, int L, int D, int two_d,
                   void* stream);

/* FAVOR+ softmax-kernel features (performer-pytorch softmax_kernel as called at rf.py:313-318; third party,
 * parity unpinned).  One workgroup per (sequence, head) S.  dash = (d^-1/4 x) P^T from rf_gemm, T, laid out
 * [S][n][m_pad] (transposed=0) or [S][m_pad][n] (transposed=1).  Row r of S in x (T) starts at
 * s0*xs[0]+s1*xs[1]+s2*xs[2]+r*xs[3] with S = (s0*n1+s1)*n2+s2.
 * y = m^-1/2 * (exp(dash - |x|^2 d^-1/2 /2 - max) + eps) for features < m, 0 for the padded ones.
 * is_query: max over the feature axis per row; else over (n, m) per S.  y may alias dash. */
int rf_favor_softmax_features(const void* dash, const void* x, const int64_t xs[4], int n1, int n2, void* y, int dtype,
                              int64_t S, int n, int m, int m_pad, int dh, int is_query, int transposed, float eps,
                              void* stream);

/* Fused FAVOR+ attention (performer-pytorch SelfAttention.fast_attention as called at rf.py:313-318 [softmax
 * kernel] and rf.py:505-518 [generalized ReLU kernel]; third party, parity unpinned): for every (batch b, outer index
 * o, head h) item the sequence s in [0,seq_len) is attended with
 *     q' = phi(q Pc^T), k' = phi(k Pc^T), out = (q' (k'^T v)) / (q' . sum_s k')
 * entirely on-chip.  qkv: bf16 rows holding q|k|v at column offsets q_off/k_off/v_off (+ h*dim_head); the row of
 * (b,o,s) of head h starts at b*x_strides[0] + o*x_strides[1] + s*x_strides[2] + h*x_strides[3] (elements); out
 * (bf16) likewise with o_strides, head h at column h*dim_head.  pc: bf16 [288][64] projection matrix pre-scaled by dim_head^-1/4, rows >= 266 zero.
 * Supported: dim_head 64, n_features 266, seq_len 64/128/256, and (ReLU kernel) any multiple of 256 walked in 256-row
 * chunks -- the L=1024 configuration (other shapes: use the unfused chain of rf_gemm +
 * rf_favor_softmax_features + rf_linattn_normalize).  softmax_kernel != 0: exp features with the library's
 * stabilisers (per-row max for q, per-(b,o,h) max for k) and eps; else relu(x)+eps. */
int rf_favor_attention(const void* qkv, const void* pc, void* out, const int64_t x_strides[4],
                       const int64_t o_strides[3], int q_off, int k_off, int v_off, int n_b, int n_o, int n_h,
                       int seq_len, int dim_head, int n_features, int softmax_kernel, float eps, void* stream);

/* Linear-attention normalisation (performer-pytorch linear_attention): y[r, :dh] = num[r, :dh] / num[r, dh]
 * where column dh of `num` carries q'.ksum (the context matrix has a ones-row appended). */
int rf_linattn_normalize(const float* num, int64_t num_ld, void* y, int y_dtype, int64_t y_ld, int64_t rows, int dh,
                         void* stream);

/* Outer-product features (rf.py:476-485): feat[b,i,j, c0 + (0..2p)] = msa1d[b,i,:], feat[.., c0+2p + (0..2p)] = msa1d[b,j,:] */
int rf_tile_1d_feats(const float* msa1d, void* feat, int dtype, int64_t feat_ld, int c0, int B, int L, int P2,
                     void* stream);

/* GraphTransformer attention core (rf.py:647-662) for one layer: q,k,v T [B,L,H*d]; e T [B,L,L,H*d];
 * out fp32 [B,L,H*d] = sum_j softmax_j(scale*(q.k_j + q.e_ij)) * (v_j + e_ij). */
int rf_graph_attention(const void* q, const void* k, const void* v, const void* e, int dtype, float* out, int B, int L,
                       int H, int d, float scale, void* stream);

/* Training-mode form of rf_graph_attention: dropout(p) on the attention probabilities (att_dropout, rf.py:628,658); mask =
 * Philox4x32-10(seed, offset + element / 4) over the [B, H, L, L] map, like rf_dropout. */
int rf_graph_attention_dropout(const void* q, const void* k, const void* v, const void* e, int dtype, float* out, int B, int L,
                               int H, int d, float scale, float p, uint64_t seed, uint64_t offset, void* stream);

/* nn.Dropout for the training-mode forward (rf.py:18-28, 76, 217, 265-281, 346, 455, 567, 592, 1138; resnet.py:30):
 * y[e] = keep[e] ? x[e] / (1 - p) : 0 over n elements of dtype (fp32 or the build's 16-bit type; y may alias x), keep[e] =
 * word (e % 4) of Philox4x32-10(key = seed, counter = offset + e / 4) >= p * 2^32.  Stateless: the caller gives every call of a
 * forward its own offset range (ceil(n / 4) counters), so a seed reproduces the forward bit for bit.  0 <= p < 1. */
int rf_dropout(const void* x, void* y, int dtype, float p, uint64_t seed, uint64_t offset, int64_t n, void* stream);

/* MsaUpdateWithPairAndCoord attention map (rf.py:899-914): att T [B,4,L,L] = softmax_j(q.k*1 + (dist<bin ? 0 : -1e9));
 * q (pre-scaled by the caller) and k: fp32 [B,L,H*dq]; ca: fp32 xyz [B,L,3,3] (CA = atom 1). */
int rf_dist_masked_attention(const float* q, const float* k, const float* xyz, const float* bins, void* att,
                             int att_dtype, int B, int L, int H, int dq, void* stream);

/* ---- SE(3) structure module (rf.py:752-862, se3_modules.py:83-171, ea/modules.py) -------------------------- */

/* kNN graph (rf.py:823-862) in dense form: mask[b,i,j] = 1 iff edge i->j exists
 * (j among the k nearest CA of i, ties broken by lower j, or |idx_i-idx_j| < kmin; self loops only when k >= L). */
int rf_knn_mask(const float* xyz, const int64_t* aa_idx, uint8_t* mask, int B, int L, int k, int kmin, void* stream);

/* Compact the dense mask into an edge list sorted by (b,i,j) (row-major torch.where order, rf.py:853):
 * src/dst node ids (b*L+i / b*L+j), eid[b,i,j] = edge id or -1.  count: 2 x int32 -- count[0] = min(edges, capacity)
 * (what consumers iterate over), count[1] = the true number of edges.  src/dst hold `capacity` entries; an edge whose
 * id would be >= capacity is DROPPED (eid = -1, nothing written) and shows up as count[1] > capacity, so the caller's
 * buffers are never overrun.  For strictly increasing aa_idx and finite coordinates a row has at most
 * min(L, k + 2*(kmin-1)) edges; otherwise use capacity B*L*L.  row_ws: 2*B*L int32 scratch. */
int rf_edges_from_mask(const uint8_t* mask, int32_t* src, int32_t* dst, int32_t* eid, int32_t* count, int32_t* row_ws,
                       int B, int L, int64_t capacity, void* stream);

/* Per-edge geometry (ea/modules.py:26-108): d = CA[dst]-CA[src]; r; real SH Y0..Y2 and the four equivariant
 * bases folded with the Q_J constants: basis layout [E, 1+3+3+27] floats = (0,0)[1] (0,1)[3x1] (1,0)[1x3] (1,1)[3x3x3];
 * also feat[e, :] = [w(edge embedding, d_edge) | r] with leading dim feat_ld. */
int rf_se3_edge_geometry(const float* xyz, const float* edge_emb, const int32_t* src, const int32_t* dst,
                         const int32_t* count, float* basis, float* feat, int64_t feat_ld, int L, int d_edge,
                         int64_t capacity, void* stream);

/* Partial convolution message (ea/modules.py:612-641 + 287-325): for output degree `dout` with `mo` channels,
 * msg[e, o, a] = sum_{di} sum_{i,b,f} R_{di}[e, o, i, f] * basis_{di,dout}[e, a, b, f] * h_{di}[src[e], i, b].
 * R0/R1: radial outputs (fp32, [E, mo*mi0*nf0] and [E, mo*mi1*nf1]); h0 [V, mi0], h1 [V, mi1, 3]. */
int rf_se3_message(const float* R0, const float* R1, const float* basis, const float* h0, const float* h1,
                   const int32_t* src, const int32_t* count, float* msg, int mo, int dout, int mi0, int mi1,
                   int64_t capacity, void* stream);

/* Fused form (round 4): the whole radial MLP (RadialFunc, ea/modules.py:246-284: Linear(d_edge+1 -> 32) -> LayerNorm
 * -> ReLU -> Linear(32 -> 32) -> LayerNorm -> ReLU -> Linear(32 -> mo*mi*nf)) evaluated per edge inside the message kernel from
 * feat [E, feat_ld] (= [edge embedding | r], ki = d_edge + 1 columns): neither the hidden vectors nor the radial outputs R
 * are written.  net_di: the packed fp32 parameters of net (di, dout), 16-byte aligned,
 *   [W1^T: ki x 32][b1 32][ln1 gamma 32][ln1 beta 32][W2^T: 32 x 32][b2 32][ln2 gamma 32][ln2 beta 32][W3: rows x 32][b3: rows],
 * rows = mo*mi_di*nf_di ordered (o, i, f) as the reference's view(-1, mo, 1, mi, 1, nf) (ea/modules.py:283); mi_di = 0 / NULL:
 * input degree absent.  rf_se3_radial_message_supported tells whether a shape has an instance (else: rf_gemm + rf_se3_message). */
int rf_se3_radial_message_supported(int mo, int dout, int mi0, int mi1, int ki);
int rf_se3_radial_message(const float* feat, int64_t feat_ld, int ki, const float* net0, const float* net1, const float* basis,
                          const float* h0, const float* h1, const int32_t* src, const int32_t* count, float* msg, int mo,
                          int dout, int mi0, int mi1, float ln_eps, int64_t capacity, void* stream);

/* Graph attention (ea/modules.py:738-774): e = <k_edge, q[dst]>/sqrt(nfeat) per head, softmax over incoming edges
 * of each dst node, out[dst] = sum a * v.  k/v per-edge: k0 [E,mk0] k1 [E,mk1,3] v0 [E,mv0] v1 [E,mv1,3]; q per node.
 * One wave per (dst node, head) walks column dst of the dense eid map in fixed order (deterministic).
 * out0 / out1 rows are out0_ld / out1_ld floats apart (0 = packed: mv0 / 3*mv1), so the result can land in the leading
 * channels of the skip-concatenation buffer (GCat, ea/modules.py:903-928). */
int rf_se3_attention(const float* k0, const float* k1, const float* q0, const float* q1, const float* v0,
                     const float* v1, const int32_t* eid, float* out0, float* out1, int heads, int mk0, int mk1,
                     int mv0, int mv1, int V, int L, int64_t out0_ld, int64_t out1_ld, void* stream);

/* GNormBias (ea/modules.py:391-406): y = relu(|v| + b) * v/|v| per channel, v [V, m, 2d+1]. */
int rf_se3_norm_bias(const float* v, const float* bias, float* y, int64_t V, int m, int deg, void* stream);

/* GAttentiveSelfInt front (ea/modules.py:446-455): s[v, a*m+b] = sign-preserving clamp(<v_a, v_b>, 1e-12). */
int rf_se3_gram(const float* v, float* s, int64_t V, int m, int deg, void* stream);
/* GAttentiveSelfInt back (ea/modules.py:458-471): y[v,o,:] = sum_m softmax_m(att[v,o,m]) * x[v,m,:]. */
int rf_se3_attn_apply(const float* att, const float* x, float* y, int64_t V, int m_out, int m_in, int deg,
                      void* stream);

/* type-1 SE(3) input (rf.py:807): y[r,a,:] = xyz[r,a,:] - xyz[r,CA,:] */
int rf_center_ca(const float* xyz, float* y, int64_t nres, void* stream);

/* Coordinate update (rf.py:816-819): xyz_out from xyz and displacement [B*L,3,3]. */
int rf_coord_apply(const float* xyz, const float* disp, float* xyz_out, int64_t nres, void* stream);

/* Library self-description */
/* Timing experiments only: when buf is non-null every bf16 rf_gemm workgroup records 8 x uint64 phase stamps into it. */
int rf_debug_gemm_stamps(void* buf);
int rf_debug_gemm_fast_stamps(void* buf);

int rf_version(void);
const char* rf_build_info(void);
int rf_h16_dtype(void); /* RF_BF16 (librfmi.so) or RF_F16 (librfmi_f16.so) */

#ifdef __cplusplus
}
#endif
#endif /* RFMI_H */
