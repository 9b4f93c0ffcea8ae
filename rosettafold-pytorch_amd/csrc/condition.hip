// Operand conditioning of the 16-bit modes (exact algebra, no change of the function computed).
//
// At random init every stream of the model is a large per-sample constant vector (the "common mode": embedding biases and
// LayerNorm offsets, the same at every position) plus a position-dependent part 10-30x smaller, and the network's
// InstanceNorms (rf.py:453,457; resnet.py:29,39,63) keep only the latter.  A 16-bit operand that still carries the constant is
// rounded relative to the constant, i.e. 10-30x coarser than the information the next InstanceNorm keeps.  These kernels
// remove the constant from an operand BEFORE it is rounded and account for it exactly on the fp32 side of the GEMM:
//     W (x - m) + (b + W m)  ==  W x + b                     (rf_center_rows / rf_center_apply + rf_fold_mean)
//     conv3x3(x - m) - sum over the taps that fall outside the picture of W_tap m  ==  conv3x3(x) - sum over all taps of W_tap m
//                                                             (rf_conv3x3_border_fix; the right side differs from conv3x3(x)
//                                                              by a per-channel constant, which the InstanceNorm removes)
// tools/precision_probe.py --pum-sweep measures what each site costs without it (PairUpdateWithMsa, rf.py:430-498: the tiled
// 1-D features 1.3e-2, the first convolution's input 9.7e-3 and output 7.3e-3 of the fp16 mode's 2.0e-2 logits gap).
#include "common.h"

#define RF_CHECK_DT(dt) \
  if ((dt) != RF_F32 && (dt) != RF_H16) return RF_EINVAL

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// ---- mean[b, c] = sums[b, c, 0] / HW  (sums of rf_instnorm_stats) ----------------------------------------------------------
__global__ __launch_bounds__(256) void sums_mean_kernel(const double* sums, float* mean, double inv, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) mean[i] = (float)(sums[2 * (int64_t)i] * inv);
}

extern "C" int rf_instnorm_mean(const void* sums, float* mean, int B, int64_t HW, int C, void* stream) {
  if (!sums || !mean || B <= 0 || C <= 0 || HW <= 0) return RF_EINVAL;
  const int n = B * C;
  hipLaunchKernelGGL(sums_mean_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)sums, mean,
                     1.0 / (double)HW, n);
  return rf_launch_status();
}

// ---- mean[b, c] over the HW pixels of fp32 NHWC x, C % 4 == 0: 16-byte loads, four pixels in flight per thread ---------------
// (rf_instnorm_stats takes fp32 input through its scalar kernel: 4x slower than this at [65536, 288].)  Per-block partial sums
// in fp32, added in block order in fp64 by the second kernel: no atomics, run-to-run identical.
#define CM_PIX_MIN 64  // fewest pixels per block (the workspace is sized for it)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float4* __restrict__ x, float* __restrict__ partials, int64_t HW,
                                                          int C4, int pix) {
  extern __shared__ float sm[];  // [ppi][C]
  const int b = blockIdx.y;
  const int ppi = 256 / C4;
  const int64_t p0 = (int64_t)blockIdx.x * pix;
  const int64_t p1 = p0 + pix < HW ? p0 + pix : HW;
  const int ch = threadIdx.x % C4, po = threadIdx.x / C4;
  if (po < ppi) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* xb = x + (int64_t)b * HW * C4 + ch;
    for (int64_t p = p0 + po; p < p1; p += 8 * ppi) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t pp = p + (int64_t)u * ppi;
        v[u] = pp < p1 ? xb[pp * C4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    ((float4*)sm)[po * C4 + ch] = s;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 4 * C4; c += 256) {
    float t = 0.f;
    for (int g = 0; g < ppi; ++g) t += sm[g * 4 * C4 + c];
    partials[((int64_t)b * gridDim.x + blockIdx.x) * 4 * C4 + c] = t;
  }
}

// 32 columns x 32 partial-sum groups per block: group g adds blocks g, g + 32, ... in order (fp64), the 32 group sums are added in
// order by the group-0 thread: a fixed summation tree
__global__ __launch_bounds__(1024) void channel_mean_finalize_kernel(const float* __restrict__ partials, float* __restrict__ mean,
                                                                     int nblk, int C, double inv) {
  __shared__ double part[32][33];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double t = 0.0;
  if (c < C)
    for (int k = g; k < nblk; k += 32) t += (double)partials[((int64_t)b * nblk + k) * C + c];
  part[g][cl] = t;
  __syncthreads();
  if (g == 0 && c < C) {
    double a = part[0][cl];
#pragma unroll
    for (int k = 1; k < 32; ++k) a += part[k][cl];
    mean[(int64_t)b * C + c] = (float)(a * inv);
  }
}

extern "C" int64_t rf_channel_mean_ws_bytes(int B, int64_t HW, int C) {
  return (int64_t)B * cdiv(HW, CM_PIX_MIN) * C * (int64_t)sizeof(float);
}

extern "C" int rf_channel_mean(const float* x, float* mean, int B, int64_t HW, int C, void* workspace, int64_t ws_bytes,
                               void* stream) {
  if (!x || !mean || B <= 0 || HW <= 0 || C <= 0 || C % 4 != 0 || C / 4 > 256 || ((uintptr_t)x % 16)) return RF_EINVAL;
  if (!workspace || ws_bytes < rf_channel_mean_ws_bytes(B, HW, C)) return RF_EINVAL;
  // pixels per block: ~2048 blocks per launch (8 per CU), between CM_PIX_MIN and 512
  int64_t pix = (HW * B + 2047) / 2048;
  pix = pix < CM_PIX_MIN ? CM_PIX_MIN : (pix > 512 ? 512 : pix);
  const unsigned nblk = cdiv(HW, pix);
  const int C4 = C / 4, ppi = 256 / C4;
  hipLaunchKernelGGL(channel_sum_kernel, dim3(nblk, B), dim3(256), (size_t)ppi * C * sizeof(float), (hipStream_t)stream,
                     (const float4*)x, (float*)workspace, HW, C4, (int)pix);
  hipLaunchKernelGGL(channel_mean_finalize_kernel, dim3(cdiv(C, 32), B), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, mean, (int)nblk, C, 1.0 / (double)HW);
  return rf_launch_status();
}

// ---- y[b, p, c] = x[b, p, c] - mean[b, c]   (x fp32 NHWC, y fp32 or the 16-bit type; C % 4 == 0) ---------------------------
// The grid stride is a multiple of the C / 4 chunks of a pixel, so a thread keeps ONE channel chunk: its four means live in
// registers and the loop is a 16-byte load, four subtractions and one 8- or 16-byte store.
template <bool H16OUT>
__global__ __launch_bounds__(256) void center_apply_kernel(const float4* __restrict__ x, const float* __restrict__ mean,
                                                           void* __restrict__ y, int64_t chunks, int C4) {
  const int b = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= chunks) return;
  const int c4 = (int)(e % C4);
  const float4 m = ((const float4*)(mean + (int64_t)b * C4 * 4))[c4];
  const float4* xb = x + (int64_t)b * chunks;
  auto put = [&](int64_t i, float4 v) {
    v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
    if (H16OUT) {
      uint2 o;
      o.x = rf_pack2_h16(v.x, v.y);
      o.y = rf_pack2_h16(v.z, v.w);
      ((uint2*)y)[(int64_t)b * chunks + i] = o;
    } else {
      ((float4*)y)[(int64_t)b * chunks + i] = v;
    }
  };
  for (; e + 3 * stride < chunks; e += 4 * stride) {   // four 16-byte loads in flight per thread
    const float4 v0 = xb[e], v1 = xb[e + stride], v2 = xb[e + 2 * stride], v3 = xb[e + 3 * stride];
    put(e, v0); put(e + stride, v1); put(e + 2 * stride, v2); put(e + 3 * stride, v3);
  }
  for (; e < chunks; e += stride) put(e, xb[e]);
}

extern "C" int rf_center_apply(const float* x, const float* mean, void* y, int y_dtype, int B, int64_t HW, int C,
                               void* stream) {
  RF_CHECK_DT(y_dtype);
  if (!x || !mean || !y || B <= 0 || HW <= 0 || C <= 0 || C % 4 != 0) return RF_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)mean % 16) || ((uintptr_t)y % 16)) return RF_EINVAL;
  const int C4 = C / 4;
  const int64_t chunks = HW * C4;
  unsigned gx = min(cdiv(chunks, 4 * 256), 2048u);
  {  // grid stride (gx * 256) a multiple of C4
    unsigned g = (unsigned)C4, r = 256u % (unsigned)C4;
    while (r) { const unsigned t = g % r; g = r; r = t; }   // gcd(C4, 256)
    const unsigned m = (unsigned)C4 / g;
    gx = gx >= m ? gx / m * m : m;
  }
  if (y_dtype == RF_F32)
    hipLaunchKernelGGL(center_apply_kernel<false>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, (const float4*)x, mean, y, chunks, C4);
  else
    hipLaunchKernelGGL(center_apply_kernel<true>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, (const float4*)x, mean, y, chunks, C4);
  return rf_launch_status();
}

// ---- small tensors: mean over the R rows of x[b] (fp32 [B, R, C]) and x -= mean in place ----------------------------------
// One block per (sample, slab of 8 channels): 32 row groups x 8 channels, the groups' sums added in order by the first 8
// threads (fixed summation order, no atomics); C % 8 != 0: one block per sample, a thread per channel.
__global__ __launch_bounds__(256) void center_rows_kernel(float* __restrict__ x, float* __restrict__ mean, int R, int C) {
  __shared__ float part[256];
  __shared__ float mu[8];
  float* xb = x + (int64_t)blockIdx.y * R * C;
  const int t = threadIdx.x;
  if (C % 8 == 0) {
    const int c = blockIdx.x * 8 + (t & 7), g = t >> 3;
    float s = 0.f;
    for (int r = g; r < R; r += 32) s += xb[(int64_t)r * C + c];
    part[t] = s;
    __syncthreads();
    if (t < 8) {
      float a = 0.f;
      for (int k = 0; k < 32; ++k) a += part[k * 8 + t];
      a /= (float)R;
      mu[t] = a;
      mean[(int64_t)blockIdx.y * C + c] = a;
    }
    __syncthreads();
    const float m = mu[t & 7];
    for (int r = g; r < R; r += 32) xb[(int64_t)r * C + c] -= m;
  } else {
    for (int c = t; c < C; c += 256) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += xb[(int64_t)r * C + c];
      s /= (float)R;
      mean[(int64_t)blockIdx.y * C + c] = s;
      for (int r = 0; r < R; ++r) xb[(int64_t)r * C + c] -= s;
    }
  }
}

extern "C" int rf_center_rows(float* x, float* mean, int B, int R, int C, void* stream) {
  if (!x || !mean || B <= 0 || R <= 0 || C <= 0) return RF_EINVAL;
  hipLaunchKernelGGL(center_rows_kernel, dim3(C % 8 == 0 ? C / 8 : 1, B), dim3(256), 0, (hipStream_t)stream, x, mean, R, C);
  return rf_launch_status();
}

// ---- an ESTIMATE of the per-channel mean of a large [B, R, C] tensor from nsample evenly spaced rows -----------------------
// (the identities above hold for ANY constant: the closer to the true mean, the better the conditioning -- 512 rows leave
// 1/22 of the spread.)  One block per (sample, slab of 8 channels): 32 row groups x 8 channels, fixed summation order.
__global__ __launch_bounds__(256) void sample_mean_kernel(const void* __restrict__ x, int dt, float* __restrict__ mean, int64_t R,
                                                          int C, int nsample) {
  __shared__ float part[256];
  const int t = threadIdx.x;
  const int c = blockIdx.x * 8 + (t & 7), g = t >> 3;
  const int64_t base = (int64_t)blockIdx.y * R * C;
  float s = 0.f;
  if (c < C) {
    int k = g;
    for (; k + 96 < nsample; k += 128) {   // four independent loads per round
      const float a0 = ld(x, dt, base + ((int64_t)k * R / nsample) * C + c);
      const float a1 = ld(x, dt, base + ((int64_t)(k + 32) * R / nsample) * C + c);
      const float a2 = ld(x, dt, base + ((int64_t)(k + 64) * R / nsample) * C + c);
      const float a3 = ld(x, dt, base + ((int64_t)(k + 96) * R / nsample) * C + c);
      s += (a0 + a1) + (a2 + a3);
    }
    for (; k < nsample; k += 32) s += ld(x, dt, base + ((int64_t)k * R / nsample) * C + c);
  }
  part[t] = s;
  __syncthreads();
  if (t < 8 && c < C) {
    float a = 0.f;
    for (int k = 0; k < 32; ++k) a += part[k * 8 + t];
    mean[(int64_t)blockIdx.y * C + c] = a / (float)nsample;
  }
}

extern "C" int rf_sample_mean(const void* x, int x_dtype, float* mean, int B, int64_t R, int C, int nsample, void* stream) {
  RF_CHECK_DT(x_dtype);
  if (!x || !mean || B <= 0 || R <= 0 || C <= 0 || nsample <= 0) return RF_EINVAL;
  if (nsample > R) nsample = (int)R;
  hipLaunchKernelGGL(sample_mean_kernel, dim3(cdiv(C, 8), B), dim3(256), 0, (hipStream_t)stream, x, x_dtype, mean, R, C, nsample);
  return rf_launch_status();
}

// ---- the constant's way through a weight matrix --------------------------------------------------------------------------
// sum_seg != 0:  out[b, n]    = (bias ? bias[n] : 0) + sum_{s < nseg} sum_{k < K} w[n, k0 + s * seg_stride + k] * mean[b, k]
// sum_seg == 0:  out[b, s, n] = (bias ? bias[n] : 0) +                 sum_{k < K} w[n, k0 + s * seg_stride + k] * mean[b, k]
// One wave per output element, lanes over k (coalesced rows of w), fixed-order butterfly reduction.
__global__ __launch_bounds__(256) void fold_mean_kernel(const float* __restrict__ w, int64_t ldw, int k0, int K, int nseg,
                                                        int64_t seg_stride, int sum_seg, const float* __restrict__ mean,
                                                        const float* __restrict__ bias, float* __restrict__ out, int N,
                                                        int64_t items) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= items) return;   // (whole waves leave together)
  const int n = (int)(item % N);
  const int so = sum_seg ? 1 : nseg;
  const int s_out = (int)((item / N) % so);
  const int b = (int)(item / ((int64_t)N * so));
  const float* mb = mean + (int64_t)b * K;
  const float* wr = w + (int64_t)n * ldw + k0;
  float acc = 0.f;
  const int s0 = sum_seg ? 0 : s_out, s1 = sum_seg ? nseg : s_out + 1;
  for (int s = s0; s < s1; ++s)
    for (int k = lane; k < K; k += 64) acc += wr[s * seg_stride + k] * mb[k];
  acc = wave_sum(acc);
  if (lane == 0) out[item] = acc + (bias ? bias[n] : 0.f);
}

extern "C" int rf_fold_mean(const float* w, int64_t ldw, int k0, int K, int nseg, int64_t seg_stride, int sum_seg,
                            const float* mean, const float* bias, float* out, int B, int N, void* stream) {
  if (!w || !mean || !out || B <= 0 || N <= 0 || K <= 0 || nseg <= 0 || k0 < 0 || seg_stride < 0) return RF_EINVAL;
  if (k0 + (int64_t)(nseg - 1) * seg_stride + K > ldw) return RF_EINVAL;
  const int64_t items = (int64_t)B * N * (sum_seg ? 1 : nseg);
  hipLaunchKernelGGL(fold_mean_kernel, dim3((unsigned)cdiv(items, 4)), dim3(256), 0, (hipStream_t)stream, w, ldw, k0, K, nseg,
                     seg_stride, sum_seg, mean, bias, out, N, items);
  return rf_launch_status();
}

// ---- 3x3 'same' convolution of a centred picture: what the zero padding owes at the border --------------------------------
// y[b, i, j, :] -= sum over the taps (kh, kw) whose source pixel (i + (kh-1) d, j + (kw-1) d) lies outside the picture of
// taps[b, kh*3 + kw, :]   (taps[b, t, o] = sum_c W[o, c, t] m[b, c], rf_fold_mean on the [Co, 9 Ci] weight).  `edges` names the
// sides of THIS block that are sides of the picture (1 top, 2 bottom, 4 left, 8 right; a row block of a sharded picture has
// neighbours above / below).  One block per border pixel, threads over the channels.
__global__ __launch_bounds__(256) void conv3x3_border_fix_kernel(void* y, int y_dt, const float* __restrict__ taps, int H, int W,
                                                                 int C, int d, int edges) {
  const int b = blockIdx.y;
  int p = blockIdx.x, i, j;
  // border pixels in order: the first d rows, the last d rows (each only if that side is an edge), then for the rows between
  // the first d and the last d columns
  const int top = (edges & 1) ? d : 0, bot = (edges & 2) ? d : 0;
  if (p < top * W) {
    i = p / W; j = p % W;
  } else if (p < (top + bot) * W) {
    p -= top * W;
    i = H - bot + p / W; j = p % W;
  } else {
    p -= (top + bot) * W;
    const int lw = (edges & 4) ? d : 0, rw = (edges & 8) ? d : 0;
    const int per = lw + rw;
    i = top + p / per;
    const int jj = p % per;
    j = jj < lw ? jj : W - rw + (jj - lw);
  }
  bool out_r[3], out_c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ii = i + (k - 1) * d, jj = j + (k - 1) * d;
    out_r[k] = (ii < 0 && (edges & 1)) || (ii >= H && (edges & 2));
    out_c[k] = (jj < 0 && (edges & 4)) || (jj >= W && (edges & 8));
  }
  const float* tb = taps + (int64_t)b * 9 * C;
  const int64_t base = (((int64_t)b * H + i) * W + j) * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float corr = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        if (out_r[kh] || out_c[kw]) corr += tb[(kh * 3 + kw) * C + c];
    st(y, y_dt, base + c, ld(y, y_dt, base + c) - corr);
  }
}

extern "C" int rf_conv3x3_border_fix(void* y, int y_dtype, const float* taps, int B, int H, int W, int C, int dilation,
                                     int edges, void* stream) {
  RF_CHECK_DT(y_dtype);
  if (!y || !taps || B <= 0 || H <= 0 || W <= 0 || C <= 0 || dilation <= 0 || (edges & ~15)) return RF_EINVAL;
  const int d = dilation;
  const int top = (edges & 1) ? d : 0, bot = (edges & 2) ? d : 0, lw = (edges & 4) ? d : 0, rw = (edges & 8) ? d : 0;
  if (top + bot > H || lw + rw > W) return RF_EINVAL;   // (pictures narrower than the stencil: not a shape of the model)
  const int64_t nb = (int64_t)(top + bot) * W + (int64_t)(H - top - bot) * (lw + rw);
  if (nb == 0) return 0;
  hipLaunchKernelGGL(conv3x3_border_fix_kernel, dim3((unsigned)nb, B), dim3(256), 0, (hipStream_t)stream, y, y_dtype, taps, H, W, C,
                     d, edges);
  return rf_launch_status();
}
