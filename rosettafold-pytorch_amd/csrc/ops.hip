// Normalisation, softmax, embedding and small-attention kernels of the RoseTTAFold forward path
// (gfx950).  All of these are HBM-bandwidth / latency bound: one wave (64 lanes) per row where a
// row reduction is needed, wave-shuffle reductions, fp32 statistics, dtype-generic (fp32 / bf16)
// loads and stores.  Reference lines are cited per entry point in include/rfmi.h.
#include "common.h"

#define RF_CHECK_DT(dt) \
  if ((dt) != RF_F32 && (dt) != RF_H16) return RF_EINVAL

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// 4-element accessors: 16-byte (fp32) / 8-byte (bf16)
__device__ __forceinline__ float4 ld4(const void* p, int dt, int64_t e) {
  if (dt == RF_F32) return *(const float4*)((const float*)p + e);
  const uint2 u = *(const uint2*)((const h16_t*)p + e);
  return make_float4(rf_h16_lo(u.x), rf_h16_hi(u.x), rf_h16_lo(u.y), rf_h16_hi(u.y));
}
__device__ __forceinline__ void st4(void* p, int dt, int64_t e, float4 v) {
  if (dt == RF_F32) {
    *(float4*)((float*)p + e) = v;
  } else {
    uint2 w;
    w.x = rf_pack2_h16(v.x, v.y);
    w.y = rf_pack2_h16(v.z, v.w);
    *(uint2*)((h16_t*)p + e) = w;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (NV values per lane)
// ------------------------------------------------------------------------------------------------
template <int NV, bool SYM>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* x, int x_dt, int64_t x_ld, void* y, int y_dt,
                                                        int64_t y_ld, int64_t rows, int D, const float* gamma,
                                                        const float* beta, float eps, int L, int groups, int act) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  int64_t base = row * x_ld, base2 = 0;
  if (SYM) {  // row = (b,i,j) of a [B,L,L,D] tensor; partner row (b,j,i)
    const int64_t j = row % L, i = (row / L) % L, b = row / ((int64_t)L * L);
    base2 = ((b * L + j) * L + i) * x_ld;
  }
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int c = lane + 64 * t;
    float a = 0.f;
    if (c < D) {
      a = ld(x, x_dt, base + c);
      if (SYM) a = 0.5f * (a + ld(x, x_dt, base2 + c));
    }
    v[t] = a;
    s += a;
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int c = lane + 64 * t;
    const float dlt = c < D ? v[t] - mean : 0.f;
    q += dlt * dlt;
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int c = lane + 64 * t;
    if (c < D) {
      float o = (v[t] - mean) * rstd;
      if (gamma) {
        const int64_t g = (groups > 1 ? (row % groups) * D : 0) + c;
        o = o * gamma[g] + beta[g];
      }
      if (act == RF_ACT_RELU) o = fmaxf(o, 0.f);
      if (act == RF_ACT_LEAKY) o = o > 0.f ? o : 0.01f * o;
      st(y, y_dt, row * y_ld + c, o);
    }
  }
}

// Vectorised LayerNorm for the two residual streams (fp32 in, D % 4 == 0, D <= 1024): one wave per row,
// 16-byte loads, NCH float4 chunks per lane, bf16 (8-byte) or fp32 (16-byte) stores; 2 rows in flight per wave.
template <int NCH, int RPI>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* x, int64_t x_ld, void* y, int y_dt, int64_t y_ld,
                                                            int64_t rows, int D, const float* gamma, const float* beta,
                                                            float eps, int act) {
  // RPI rows per wave iteration: all their loads are issued before the first reduction (4.4 TB/s of mixed read + write
  // traffic on the 288-wide stream; more rows in flight do not help)
  const int lane = threadIdx.x & 63;
  const int nch = D >> 2;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row0 = wid * RPI; row0 < rows; row0 += nw * RPI) {
    float4 v[RPI][NCH];
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const int64_t row = row0 + r;
#pragma unroll
      for (int t = 0; t < NCH; ++t) {
        const int c = lane + 64 * t;
        v[r][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch && row < rows) v[r][t] = *(const float4*)(x + row * x_ld + 4 * c);
      }
    }
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const int64_t row = row0 + r;
      if (row >= rows) break;
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < NCH; ++t) s += (v[r][t].x + v[r][t].y) + (v[r][t].z + v[r][t].w);
      const float mean = wave_sum(s) / D;
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < NCH; ++t) {
        const int c = lane + 64 * t;
        if (c < nch) {
          const float a = v[r][t].x - mean, b = v[r][t].y - mean, cc = v[r][t].z - mean, dd = v[r][t].w - mean;
          q += (a * a + b * b) + (cc * cc + dd * dd);
        }
      }
      const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
      for (int t = 0; t < NCH; ++t) {
        const int c = lane + 64 * t;
        if (c < nch) {
          float o[4] = {(v[r][t].x - mean) * rstd, (v[r][t].y - mean) * rstd, (v[r][t].z - mean) * rstd, (v[r][t].w - mean) * rstd};
          if (gamma) {
            const float4 g = *(const float4*)(gamma + 4 * c), b = *(const float4*)(beta + 4 * c);
            o[0] = o[0] * g.x + b.x; o[1] = o[1] * g.y + b.y; o[2] = o[2] * g.z + b.z; o[3] = o[3] * g.w + b.w;
          }
          if (act == RF_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
          } else if (act == RF_ACT_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.01f * o[e];
          }
          if (y_dt == RF_F32) {
            *(float4*)((float*)y + row * y_ld + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
          } else {
            uint2 w;
            w.x = rf_pack2_h16(o[0], o[1]);
            w.y = rf_pack2_h16(o[2], o[3]);
            *(uint2*)((h16_t*)y + row * y_ld + 4 * c) = w;
          }
        }
      }
    }
  }
}

// Residual-stream LayerNorm, eight rows per wave: lane (r = lane/8, q = lane%8) owns the float4 chunks q, q+8, ... of row r
// (KPL chunks per lane: 9 for D = 288, 12 for D = 384), so every load instruction covers 8 rows x 128 contiguous bytes, the
// row statistics are 3-step reductions inside 8-lane groups, and gamma / beta stay in registers.  The one-row-per-wave
// kernel above issues ~150 vector instructions per 1.1 KB row (two of 64 lanes hold a second chunk, every lane pays for
// it) and is ALU-bound at ~4.5 TB/s; this form needs ~27 per row and runs at the streaming rate of a fp32 -> bf16 cast.
template <int KPL, bool SYM = false>
__global__ __launch_bounds__(256) void layernorm_rows8_kernel(const float* x, int64_t x_ld, void* y, int y_dt, int64_t y_ld,
                                                              int64_t rows, const float* gamma, const float* beta, float eps,
                                                              int act, int Lsym = 0) {
  constexpr int D = KPL * 32;
  const int lane = threadIdx.x & 63;
  const int r = lane >> 3, q = lane & 7;
  float4 g[KPL], b[KPL];
#pragma unroll
  for (int k = 0; k < KPL; ++k) {
    g[k] = gamma ? *(const float4*)(gamma + 4 * (q + 8 * k)) : make_float4(1.f, 1.f, 1.f, 1.f);
    b[k] = beta ? *(const float4*)(beta + 4 * (q + 8 * k)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row0 = wid * 8; row0 < rows; row0 += nw * 8) {
    const int64_t row = row0 + r;
    const bool ok = row < rows;
    float4 v[KPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v[k] = *(const float4*)(x + row * x_ld + 4 * (q + 8 * k));
    }
    if constexpr (SYM) {  // row = (b,i,j) of a [B,L,L,D] tensor: normalise 0.5 * (x[b,i,j] + x[b,j,i])  (rf.py:550-556)
      if (ok) {
        const int64_t j = row % Lsym, i = (row / Lsym) % Lsym, bb = row / ((int64_t)Lsym * Lsym);
        const float* xp = x + ((bb * Lsym + j) * Lsym + i) * x_ld;
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
          const float4 t = *(const float4*)(xp + 4 * (q + 8 * k));
          v[k] = make_float4(0.5f * (v[k].x + t.x), 0.5f * (v[k].y + t.y), 0.5f * (v[k].z + t.z), 0.5f * (v[k].w + t.w));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    const float mean = s * (1.0f / D);
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const float a0 = v[k].x - mean, a1 = v[k].y - mean, a2 = v[k].z - mean, a3 = v[k].w - mean;
      qq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    qq += __shfl_xor(qq, 1, 64);
    qq += __shfl_xor(qq, 2, 64);
    qq += __shfl_xor(qq, 4, 64);
    const float rstd = rsqrtf(qq * (1.0f / D) + eps);
    if (!ok) continue;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      float o[4] = {(v[k].x - mean) * rstd * g[k].x + b[k].x, (v[k].y - mean) * rstd * g[k].y + b[k].y,
                    (v[k].z - mean) * rstd * g[k].z + b[k].z, (v[k].w - mean) * rstd * g[k].w + b[k].w};
      if (act == RF_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
      } else if (act == RF_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.01f * o[e];
      }
      const int c = q + 8 * k;
      if (y_dt == RF_F32) {
        *(float4*)((float*)y + row * y_ld + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        uint2 w;
        w.x = rf_pack2_h16(o[0], o[1]);
        w.y = rf_pack2_h16(o[2], o[3]);
        *(uint2*)((h16_t*)y + row * y_ld + 4 * c) = w;
      }
    }
  }
}

// bf16-input twin (the 1024-wide outer-product rows, rf.py:416): 16-byte loads of 8 bf16, NCH chunks per lane
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_vec_bf16_kernel(const h16_t* x, int64_t x_ld, void* y, int y_dt, int64_t y_ld,
                                                                 int64_t rows, int D, const float* gamma, const float* beta,
                                                                 float eps) {
  const int lane = threadIdx.x & 63;
  const int nch = D >> 3;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = wid; row < rows; row += nw) {
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
      const int c = lane + 64 * t;
      uint4 u = make_uint4(0u, 0u, 0u, 0u);
      if (c < nch) u = *(const uint4*)(x + row * x_ld + 8 * c);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[t][2 * e] = rf_h16_lo(w[e]);
        v[t][2 * e + 1] = rf_h16_hi(w[e]);
        s += v[t][2 * e] + v[t][2 * e + 1];
      }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NCH; ++t)
      if (lane + 64 * t < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = v[t][e] - mean;
          q += a * a;
        }
      }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
      const int c = lane + 64 * t;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[t][e] - mean) * rstd;
        if (gamma) {
          const float4 g0 = *(const float4*)(gamma + 8 * c), g1 = *(const float4*)(gamma + 8 * c + 4);
          const float4 b0 = *(const float4*)(beta + 8 * c), b1 = *(const float4*)(beta + 8 * c + 4);
          o[0] = o[0] * g0.x + b0.x; o[1] = o[1] * g0.y + b0.y; o[2] = o[2] * g0.z + b0.z; o[3] = o[3] * g0.w + b0.w;
          o[4] = o[4] * g1.x + b1.x; o[5] = o[5] * g1.y + b1.y; o[6] = o[6] * g1.z + b1.z; o[7] = o[7] * g1.w + b1.w;
        }
        if (y_dt == RF_F32) {
          float* yp = (float*)y + row * y_ld + 8 * c;
          *(float4*)yp = make_float4(o[0], o[1], o[2], o[3]);
          *(float4*)(yp + 4) = make_float4(o[4], o[5], o[6], o[7]);
        } else {
          uint4 w;
          w.x = rf_pack2_h16(o[0], o[1]);
          w.y = rf_pack2_h16(o[2], o[3]);
          w.z = rf_pack2_h16(o[4], o[5]);
          w.w = rf_pack2_h16(o[6], o[7]);
          *(uint4*)((h16_t*)y + row * y_ld + 8 * c) = w;
        }
      }
    }
  }
}

// Narrow rows (D = 32 or 64, fp32 in: the d_proj / d_state / radial-MLP norms): LPR = D/4 lanes per row, 64/LPR rows per
// wave instruction, float4 per lane, statistics reduced inside the LPR-lane group.
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const float* x, int64_t x_ld, void* y, int y_dt, int64_t y_ld,
                                                               int64_t rows, const float* gamma, const float* beta, float eps,
                                                               int groups, int act) {
  constexpr int D = 4 * LPR, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, c = lane % LPR;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wid * RPW; r0 < rows; r0 += nw * RPW) {
    const int64_t row = r0 + sub;
    const bool ok = row < rows;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *(const float4*)(x + row * x_ld + 4 * c);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / D;
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    float q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / D + eps);
    if (ok) {
      float o[4] = {a0 * rstd, a1 * rstd, a2 * rstd, a3 * rstd};
      if (gamma) {
        const int64_t gb = (groups > 1 ? (row % groups) * D : 0) + 4 * c;
        const float4 g = *(const float4*)(gamma + gb), b = *(const float4*)(beta + gb);
        o[0] = o[0] * g.x + b.x; o[1] = o[1] * g.y + b.y; o[2] = o[2] * g.z + b.z; o[3] = o[3] * g.w + b.w;
      }
      if (act == RF_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
      } else if (act == RF_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.01f * o[e];
      }
      if (y_dt == RF_F32) {
        *(float4*)((float*)y + row * y_ld + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        uint2 w;
        w.x = rf_pack2_h16(o[0], o[1]);
        w.y = rf_pack2_h16(o[2], o[3]);
        *(uint2*)((h16_t*)y + row * y_ld + 4 * c) = w;
      }
    }
  }
}

#ifndef LN_RPI
#define LN_RPI 2  // rows in flight per wave (measured: 1 -> 106 us, 2 -> 103 us, 4 -> 134 us on the pair stream)
#endif
template <bool SYM>
static int launch_ln(const void* x, int x_dt, int64_t x_ld, void* y, int y_dt, int64_t y_ld, int64_t rows, int D,
                     const float* g, const float* b, float eps, int L, int groups, int act, hipStream_t s) {
  if (rows <= 0 || D <= 0 || D > 2304) return RF_EINVAL;
  if (SYM && x_dt == RF_F32 && (D == 288 || D == 384) && x_ld % 4 == 0 && y_ld % 4 == 0 && ((uintptr_t)x % 16) == 0 &&
      ((uintptr_t)y % 16) == 0 && (!g || (((uintptr_t)g % 16) == 0 && ((uintptr_t)b % 16) == 0))) {
    const unsigned gr = (unsigned)(rows < 65536 ? cdiv(rows, 32) : 2048);
    if (D == 288)
      hipLaunchKernelGGL((layernorm_rows8_kernel<9, true>), dim3(gr), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, g, b, eps, act, L);
    else
      hipLaunchKernelGGL((layernorm_rows8_kernel<12, true>), dim3(gr), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, g, b, eps, act, L);
    return rf_launch_status();
  }
  if (!SYM && x_dt == RF_F32 && groups <= 1 && D % 4 == 0 && D >= 128 && D <= 1024 && x_ld % 4 == 0 && y_ld % 4 == 0 &&
      ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (!g || (((uintptr_t)g % 16) == 0 && ((uintptr_t)b % 16) == 0))) {
    static const bool ln_rows1 = rf_env_flag("RF_LN_ROWS1");
    if ((D == 288 || D == 384) && groups <= 1 && !ln_rows1) {
      const unsigned gr = (unsigned)(rows < 65536 ? cdiv(rows, 32) : 2048);
      if (D == 288)
        hipLaunchKernelGGL((layernorm_rows8_kernel<9>), dim3(gr), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, g, b, eps, act);
      else
        hipLaunchKernelGGL((layernorm_rows8_kernel<12>), dim3(gr), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, g, b, eps, act);
      return rf_launch_status();
    }
    const unsigned gv = (unsigned)(rows < 8192 * LN_RPI ? cdiv(rows, 4 * LN_RPI) : 2048);
    if (D <= 512)
      hipLaunchKernelGGL((layernorm_vec_kernel<2, LN_RPI>), dim3(gv), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, D, g, b, eps, act);
    else
      hipLaunchKernelGGL((layernorm_vec_kernel<4, 2>), dim3(gv), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, D, g, b, eps, act);
    return rf_launch_status();
  }
  const bool al16 = ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (!g || (((uintptr_t)g % 16) == 0 && ((uintptr_t)b % 16) == 0));
  if (!SYM && x_dt == RF_H16 && groups <= 1 && act == RF_ACT_NONE && D % 8 == 0 && D > 512 && D <= 1024 && x_ld % 8 == 0 &&
      y_ld % 8 == 0 && al16) {
    const unsigned gv = (unsigned)(rows < 8192 ? cdiv(rows, 4) : 2048);
    hipLaunchKernelGGL((layernorm_vec_bf16_kernel<2>), dim3(gv), dim3(256), 0, s, (const h16_t*)x, x_ld, y, y_dt, y_ld, rows, D, g, b, eps);
    return rf_launch_status();
  }
  if (!SYM && x_dt == RF_F32 && (D == 32 || D == 64) && x_ld % 4 == 0 && y_ld % 4 == 0 && al16) {
    const int rpb = 4 * (256 / D);  // rows per workgroup pass
    const unsigned gv = (unsigned)(rows < (int64_t)rpb * 4096 ? cdiv(rows, rpb) : 4096);
    if (D == 32)
      hipLaunchKernelGGL((layernorm_narrow_kernel<8>), dim3(gv), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, g, b, eps, groups, act);
    else
      hipLaunchKernelGGL((layernorm_narrow_kernel<16>), dim3(gv), dim3(256), 0, s, (const float*)x, x_ld, y, y_dt, y_ld, rows, g, b, eps, groups, act);
    return rf_launch_status();
  }
  const dim3 grid(cdiv(rows, 4)), blk(256);
#define RF_LN(NV) hipLaunchKernelGGL((layernorm_kernel<NV, SYM>), grid, blk, 0, s, x, x_dt, x_ld, y, y_dt, y_ld, rows, D, g, b, eps, L, groups, act)
  if (D <= 64) RF_LN(1);
  else if (D <= 128) RF_LN(2);
  else if (D <= 320) RF_LN(5);
  else if (D <= 384) RF_LN(6);
  else if (D <= 512) RF_LN(8);
  else if (D <= 1024) RF_LN(16);
  else RF_LN(36);
#undef RF_LN
  return rf_launch_status();
}

extern "C" int rf_layernorm(const void* x, int x_dtype, int64_t x_ld, void* y, int y_dtype, int64_t y_ld, int64_t rows,
                            int D, const float* gamma, const float* beta, float eps, int groups, int act,
                            void* stream) {
  RF_CHECK_DT(x_dtype);
  RF_CHECK_DT(y_dtype);
  if ((gamma == nullptr) != (beta == nullptr)) return RF_EINVAL;
  return launch_ln<false>(x, x_dtype, x_ld, y, y_dtype, y_ld, rows, D, gamma, beta, eps, 0, groups, act,
                          (hipStream_t)stream);
}

extern "C" int rf_sym_layernorm(const float* pair, void* y, int y_dtype, int B, int L, int D, float eps, void* stream) {
  RF_CHECK_DT(y_dtype);
  return launch_ln<true>(pair, RF_F32, D, y, y_dtype, D, (int64_t)B * L * L, D, nullptr, nullptr, eps, L, 1, 0,
                         (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// strided row softmax: one wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_kernel(const float* x, int64_t x_rs, int64_t x_cs, void* y, int y_dt,
                                                      int64_t y_rs, int64_t rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * x_rs;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c * x_cs] * scale);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += __expf(xr[c * x_cs] * scale - mx);
  const float inv = 1.f / wave_sum(s);
  for (int c = lane; c < cols; c += 64) st(y, y_dt, row * y_rs + c, __expf(xr[c * x_cs] * scale - mx) * inv);
}

// nbatch independent softmax problems in one launch: problem z reads x + z*x_bs and writes y + z*y_bs (elements)
__global__ __launch_bounds__(256) void softmax_batched_kernel(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_cs, void* y,
                                                              int y_dt, int64_t y_bs, int64_t y_rs, int64_t rows, int cols,
                                                              float scale, int nbatch) {
  const int lane = threadIdx.x & 63;
  const int64_t gr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gr >= rows * nbatch) return;
  const int64_t z = gr / rows, row = gr % rows;
  const float* xr = x + z * x_bs + row * x_rs;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c * x_cs] * scale);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += __expf(xr[c * x_cs] * scale - mx);
  const float inv = 1.f / wave_sum(s);
  for (int c = lane; c < cols; c += 64) st(y, y_dt, z * y_bs + row * y_rs + c, __expf(xr[c * x_cs] * scale - mx) * inv);
}

extern "C" int rf_softmax_batched(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_cs, void* y, int y_dtype, int64_t y_bs,
                                  int64_t y_rs, int64_t rows, int cols, float scale, int nbatch, void* stream) {
  RF_CHECK_DT(y_dtype);
  if (rows <= 0 || cols <= 0 || nbatch <= 0) return RF_EINVAL;
  hipLaunchKernelGGL(softmax_batched_kernel, dim3(cdiv(rows * nbatch, 4)), dim3(256), 0, (hipStream_t)stream, x, x_bs, x_rs, x_cs,
                     y, y_dtype, y_bs, y_rs, rows, cols, scale, nbatch);
  return rf_launch_status();
}

extern "C" int rf_softmax(const float* x, int64_t x_rs, int64_t x_cs, void* y, int y_dtype, int64_t y_rs, int64_t rows,
                          int cols, float scale, void* stream) {
  RF_CHECK_DT(y_dtype);
  if (rows <= 0 || cols <= 0) return RF_EINVAL;
  hipLaunchKernelGGL(softmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, x_rs, x_cs, y, y_dtype,
                     y_rs, rows, cols, scale);
  return rf_launch_status();
}

__global__ __launch_bounds__(256) void att_sym_kernel(const void* att, int dt, float* sym, int64_t sym_ld, int B, int H,
                                                      int L) {
  // sym[b,i,j,h] = 0.5*(att[b,h,i,j] + att[b,h,j,i])
  const int64_t n = (int64_t)B * L * L * H;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int h = e % H;
    const int64_t t = e / H;
    const int j = t % L, i = (t / L) % L;
    const int64_t b = t / ((int64_t)L * L);
    const int64_t o = (b * H + h) * L;
    sym[t * sym_ld + h] = 0.5f * (ld(att, dt, (o + i) * L + j) + ld(att, dt, (o + j) * L + i));
  }
}

extern "C" int rf_tied_softmax(const float* logits, void* att, int att_dtype, float* att_sym, int64_t sym_ld, int B,
                               int H, int L, void* stream) {
  RF_CHECK_DT(att_dtype);
  const int64_t rows = (int64_t)B * H * L;
  hipLaunchKernelGGL(softmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, (int64_t)L,
                     (int64_t)1, att, att_dtype, (int64_t)L, rows, L, 1.0f);
  if (att_sym) {
    const int64_t n = (int64_t)B * L * L * H;
    hipLaunchKernelGGL(att_sym_kernel, dim3(min(cdiv(n, 256), 8192u)), dim3(256), 0, (hipStream_t)stream, att,
                       att_dtype, att_sym, sym_ld, B, H, L);
  }
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// PositionWiseWeightFactor core: block per (b,l); thread per (n,h) dot product; softmax over n
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void poswise_kernel(const void* q0, int q0_dt, int64_t q0_ld, const void* k,
                                                      int64_t k_ld, int k_col0, int k_hs, int dlen, float* w, void* qs,
                                                      int64_t qs_ld, int qs_col0, int qs_dh, int dt, int B, int N,
                                                      int L, int H, float scale, float qscale) {
  extern __shared__ float sm[];  // logits [H][N]
  const int bl = blockIdx.x, b = bl / L, l = bl % L;
  const int NH = N * H;
  for (int e = threadIdx.x; e < NH; e += 256) {
    const int n = e / H, h = e % H;
    const int64_t kb = (((int64_t)b * N + n) * L + l) * k_ld + k_col0 + h * k_hs;
    const int64_t qb = ((int64_t)b * L + l) * q0_ld + h * dlen;
    float a = 0.f;
    if (dt == RF_H16 && q0_dt == RF_H16 && (dlen & 7) == 0 && ((kb | qb) & 7) == 0) {
      const h16x8* qp = (const h16x8*)((const h16_t*)q0 + qb);
      const h16x8* kp = (const h16x8*)((const h16_t*)k + kb);
      for (int c = 0; c < (dlen >> 3); ++c) {
        const h16x8 qv = qp[c], kv = kp[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(h2f((h16_t)qv[e]), h2f((h16_t)kv[e]), a);
      }
    } else if (dt == RF_H16 && q0_dt == RF_F32 && (dlen & 7) == 0 && (kb & 7) == 0 && (qb & 3) == 0) {
      const float4* qp = (const float4*)((const float*)q0 + qb);
      const h16x8* kp = (const h16x8*)((const h16_t*)k + kb);
      for (int c = 0; c < (dlen >> 3); ++c) {
        const h16x8 kv = kp[c];
        const float4 q1 = qp[2 * c], q2 = qp[2 * c + 1];
        a = fmaf(q1.x, h2f((h16_t)kv[0]), a); a = fmaf(q1.y, h2f((h16_t)kv[1]), a);
        a = fmaf(q1.z, h2f((h16_t)kv[2]), a); a = fmaf(q1.w, h2f((h16_t)kv[3]), a);
        a = fmaf(q2.x, h2f((h16_t)kv[4]), a); a = fmaf(q2.y, h2f((h16_t)kv[5]), a);
        a = fmaf(q2.z, h2f((h16_t)kv[6]), a); a = fmaf(q2.w, h2f((h16_t)kv[7]), a);
      }
    } else if (dt == RF_F32 && q0_dt == RF_F32 && (dlen & 3) == 0 && ((kb | qb) & 3) == 0) {
      // fp32 x fp32 (structure-track node input: LayerNorm(msa) is kept in fp32 there), same c-ordered fmaf chain as below
      const float4* qp = (const float4*)((const float*)q0 + qb);
      const float4* kp = (const float4*)((const float*)k + kb);
      for (int c = 0; c < (dlen >> 2); ++c) {
        const float4 q1 = qp[c], k1 = kp[c];
        a = fmaf(q1.x, k1.x, a); a = fmaf(q1.y, k1.y, a);
        a = fmaf(q1.z, k1.z, a); a = fmaf(q1.w, k1.w, a);
      }
    } else {
      for (int c = 0; c < dlen; ++c) a = fmaf(ld(q0, q0_dt, qb + c), ld(k, dt, kb + c), a);
    }
    sm[h * N + n] = a * scale;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int h = wv; h < H; h += 4) {
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, sm[h * N + n]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += __expf(sm[h * N + n] - mx);
    const float inv = 1.f / wave_sum(s);
    for (int n = lane; n < N; n += 64) sm[h * N + n] = __expf(sm[h * N + n] - mx) * inv;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NH; e += 256) {
    const int n = e / H, h = e % H;
    const float wv_ = sm[h * N + n];
    if (w) w[(((int64_t)b * N + n) * H + h) * L + l] = wv_;
    if (qs) {
      const int64_t o = (((int64_t)b * N + n) * L + l) * qs_ld + qs_col0 + h * qs_dh;
      const float f = wv_ * qscale;
      if (dt == RF_H16 && (qs_dh & 7) == 0 && (o & 7) == 0) {
        h16x8* qp = (h16x8*)((h16_t*)qs + o);
        for (int c = 0; c < (qs_dh >> 3); ++c) {
          h16x8 v = qp[c];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (short)f2h(h2f((h16_t)v[e]) * f);
          qp[c] = v;
        }
      } else {
        for (int c = 0; c < qs_dh; ++c) st(qs, dt, o + c, ld(qs, dt, o + c) * f);
      }
    }
  }
}

extern "C" int rf_poswise(const void* q0, int q0_dtype, int64_t q0_ld, const void* k, int64_t k_ld, int k_col0,
                          int k_hstride, int dlen, float* w, void* q_scale, int64_t qs_ld, int qs_col0, int qs_dh,
                          int dtype, int B, int N, int L, int H, float scale, float qscale, void* stream) {
  RF_CHECK_DT(dtype);
  RF_CHECK_DT(q0_dtype);
  const size_t lds = (size_t)N * H * sizeof(float);
  if (lds > 64 * 1024) return RF_EINVAL;
  hipLaunchKernelGGL(poswise_kernel, dim3(B * L), dim3(256), lds, (hipStream_t)stream, q0, q0_dtype, q0_ld, k, k_ld,
                     k_col0, k_hstride, dlen, w, q_scale, qs_ld, qs_col0, qs_dh, dtype, B, N, L, H, scale, qscale);
  return rf_launch_status();
}

__global__ __launch_bounds__(256) void weighted_msa_sum_kernel(const void* x, int dt, const float* w, float* y,
                                                               int64_t y_ld, int B, int N, int L, int D) {
  const int bl = blockIdx.x, b = bl / L, l = bl % L;
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f;
    for (int n = 0; n < N; ++n)
      a = fmaf(w[((int64_t)b * N + n) * L + l], ld(x, dt, (((int64_t)b * N + n) * L + l) * D + c), a);
    y[((int64_t)b * L + l) * y_ld + c] = a;
  }
}

extern "C" int rf_weighted_msa_sum(const void* x, int dtype, const float* w, float* y, int64_t y_ld, int B, int N, int L,
                                   int D, void* stream) {
  RF_CHECK_DT(dtype);
  hipLaunchKernelGGL(weighted_msa_sum_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, x, dtype, w, y, y_ld, B, N,
                     L, D);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm over NHWC: statistics (per-block fp32 partials reduced in block order in fp64: no atomics) + apply
// ------------------------------------------------------------------------------------------------
#define IN_PIX 128  // pixels per block
__global__ void instnorm_finalize_kernel(const float* partials, double* sums, int nblk, int C);
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const void* x, int dt, double* sums, float* partials, int64_t HW,
                                                             int C) {
  const int b = blockIdx.y;
  const int64_t p0 = (int64_t)blockIdx.x * IN_PIX;
  const int64_t p1 = p0 + IN_PIX < HW ? p0 + IN_PIX : HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, q = 0.f;
    for (int64_t p = p0; p < p1; ++p) {
      const float v = ld(x, dt, ((int64_t)b * HW + p) * C + c);
      s += v;
      q = fmaf(v, v, q);
    }
    float* pp = partials + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C;  // reduced by instnorm_finalize_kernel
    pp[c] = s;
    pp[C + c] = q;
  }
}

// ---- vectorised fast paths (x bf16 NHWC, C % 8 == 0): 16-byte loads, 8 channels per thread -------------------
#define INV_PIX 512  // pixels per block in the vectorised statistics kernel
__global__ __launch_bounds__(256) void instnorm_stats_vec_kernel(const h16_t* x, double* sums, float* partials, int64_t HW,
                                                                 int C) {
  extern __shared__ float sm[];  // [ppi][2][C]: one slot per pixel group, summed in a fixed order (run-to-run identical)
  const int b = blockIdx.y;
  const int nch = C >> 3;
  const int ppi = 256 / nch;  // pixels handled per iteration
  const int64_t p0 = (int64_t)blockIdx.x * INV_PIX;
  const int64_t p1 = p0 + INV_PIX < HW ? p0 + INV_PIX : HW;
  const int ch = threadIdx.x % nch, po = threadIdx.x / nch;
  if (po < ppi) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    // 8 independent 16-byte loads in flight per thread (one load per iteration left the kernel latency-bound at 1.7 TB/s)
    for (int64_t p = p0 + po; p < p1; p += 8 * ppi) {
      h16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t pp = p + (int64_t)u * ppi;
        v[u] = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (pp < p1) v[u] = *(const h16x8*)(x + ((int64_t)b * HW + pp) * C + ch * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = h2f((h16_t)v[u][e]);
          s[e] += f;
          q[e] = fmaf(f, f, q[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sm[(po * 2 + 0) * C + ch * 8 + e] = s[e];
      sm[(po * 2 + 1) * C + ch * 8 + e] = q[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {  // c < C: sum, c >= C: sum of squares
    float t = 0.f;
    for (int g = 0; g < ppi; ++g) t += sm[g * 2 * C + c];
    partials[((int64_t)b * gridDim.x + blockIdx.x) * 2 * C + c] = t;  // reduced in block order by instnorm_finalize_kernel
  }
}

// 32 columns x 8 partial-sum groups per block: group g adds blocks g, g + 8, ... in order, the 8 group sums are added in
// order by the group-0 thread -- a fixed summation tree (run-to-run identical), 16 + 8 dependent adds deep instead of nblk
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const float* partials, double* sums, int nblk, int C) {
  __shared__ double part[8][32];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double t = 0.0;
  if (c < 2 * C)
    for (int k = g; k < nblk; k += 8) t += (double)partials[((int64_t)b * nblk + k) * 2 * C + c];
  part[g][cl] = t;
  __syncthreads();
  if (g == 0 && c < 2 * C) {
    double a = part[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) a += part[k][cl];
    sums[((int64_t)b * C + (c < C ? c : c - C)) * 2 + (c < C ? 0 : 1)] = a;
  }
}

__global__ __launch_bounds__(256) void instnorm_apply_vec_kernel(const h16_t* x, const double* sums, const float* gamma,
                                                                 const float* beta, float eps, const float* residual,
                                                                 int act, void* y, int y_dt, void* y2, int y2_dt,
                                                                 int64_t HW, int C) {
  extern __shared__ float sm[];  // scale[C], shift[C]
  const int b = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double s = sums[((int64_t)b * C + c) * 2], q = sums[((int64_t)b * C + c) * 2 + 1];
    const double mean = s / (double)HW;
    double var = q / (double)HW - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float sc = rsqrtf((float)var + eps) * gamma[c];
    sm[c] = sc;
    sm[C + c] = beta[c] - (float)mean * sc;
  }
  __syncthreads();
  const int nch = C >> 3;
  const int64_t nchunks = HW * nch;
  const int64_t stride = (int64_t)gridDim.x * 256;
  // the host picks a grid whose stride is a multiple of the chunks per pixel whenever it can: every thread then keeps ONE
  // channel chunk for the whole loop -- scale / shift live in registers and the 64-bit modulo per 16 bytes (which made this
  // kernel ALU-bound at 2.9 TB/s) disappears
  const bool fixed = stride % nch == 0;
  int ch = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % nch);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = sm[ch * 8 + e];
    sh[e] = sm[C + ch * 8 + e];
  }
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nchunks; idx += stride) {
    if (!fixed) {
      ch = (int)(idx % nch);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[e] = sm[ch * 8 + e];
        sh[e] = sm[C + ch * 8 + e];
      }
    }
    const int64_t e0 = ((int64_t)b * HW) * C + idx * 8;  // idx = pixel*nch + ch -> element offset pixel*C + ch*8
    const h16x8 v = *(const h16x8*)(x + e0);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = h2f((h16_t)v[e]) * sc[e] + sh[e];
    if (residual) {
      const float4 r0 = *(const float4*)(residual + e0), r1 = *(const float4*)(residual + e0 + 4);
      o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w;
      o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w;
    }
    if (act == RF_ACT_ELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] > 0.f ? o[e] : __expf(o[e]) - 1.f;
    }
    auto put = [&](void* dst, int dt) {
      if (dt == RF_F32) {
        *(float4*)((float*)dst + e0) = make_float4(o[0], o[1], o[2], o[3]);
        *(float4*)((float*)dst + e0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
      } else {
        uint4 w;
        w.x = rf_pack2_h16(o[0], o[1]);
        w.y = rf_pack2_h16(o[2], o[3]);
        w.z = rf_pack2_h16(o[4], o[5]);
        w.w = rf_pack2_h16(o[6], o[7]);
        *(uint4*)((h16_t*)dst + e0) = w;
      }
    };
    put(y, y_dt);
    if (y2) put(y2, y2_dt);
  }
}

/* workspace size for the atomics-free (bitwise reproducible) statistics path */
extern "C" int64_t rf_instnorm_ws_bytes(int B, int64_t HW, int C) {
  return (int64_t)B * cdiv(HW, IN_PIX) * 2 * C * (int64_t)sizeof(float);  // (sized for the finer of the two block shapes)
}

extern "C" int rf_instnorm_stats(const void* x, int x_dtype, void* sums, int B, int64_t HW, int C, void* workspace,
                                 int64_t ws_bytes, void* stream) {
  RF_CHECK_DT(x_dtype);
  // the workspace (rf_instnorm_ws_bytes) is mandatory: the library has no atomic-accumulation path, so results are
  // bitwise reproducible run to run
  if (!workspace || ws_bytes < rf_instnorm_ws_bytes(B, HW, C)) return RF_EINVAL;
  float* partials = (float*)workspace;
  if (x_dtype == RF_H16 && C % 8 == 0 && C / 8 <= 256 && ((uintptr_t)x % 16) == 0) {
    const unsigned nblk = cdiv(HW, INV_PIX);
    const int ppi = 256 / (C / 8);
    hipLaunchKernelGGL(instnorm_stats_vec_kernel, dim3(nblk, B), dim3(256), (size_t)ppi * 2 * C * sizeof(float),
                       (hipStream_t)stream, (const h16_t*)x, (double*)sums, partials, HW, C);
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(cdiv(2 * C, 32), B), dim3(256), 0, (hipStream_t)stream, partials,
                       (double*)sums, (int)nblk, C);
    return rf_launch_status();
  }
  const unsigned nblk = cdiv(HW, IN_PIX);
  hipLaunchKernelGGL(instnorm_stats_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, x, x_dtype, (double*)sums,
                     partials, HW, C);
  hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(cdiv(2 * C, 32), B), dim3(256), 0, (hipStream_t)stream, partials,
                     (double*)sums, (int)nblk, C);
  return rf_launch_status();
}

__global__ __launch_bounds__(256) void instnorm_apply_kernel(const void* x, int x_dt, const double* sums,
                                                             const float* gamma, const float* beta, float eps,
                                                             const float* residual, int act, void* y, int y_dt,
                                                             void* y2, int y2_dt, int64_t HW, int C, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = e % C;
    const int64_t b = e / (HW * C);
    const double s = sums[(b * C + c) * 2], q = sums[(b * C + c) * 2 + 1];
    const double mean = s / (double)HW;
    double var = q / (double)HW - mean * mean;
    var = var > 0.0 ? var : 0.0;
    // gamma == nullptr: centre only (y = x - mean over the picture; PredictionHead's operand conditioning, model.py)
    float v = gamma ? (ld(x, x_dt, e) - (float)mean) * rsqrtf((float)var + eps) * gamma[c] + beta[c] : ld(x, x_dt, e) - (float)mean;
    if (residual) v += residual[e];
    if (act == RF_ACT_ELU) v = elu1(v);
    st(y, y_dt, e, v);
    if (y2) st(y2, y2_dt, e, v);
  }
}

extern "C" int rf_instnorm_apply(const void* x, int x_dtype, const void* sums, const float* gamma, const float* beta,
                                 float eps, const float* residual, int act, void* y, int y_dtype, void* y2,
                                 int y2_dtype, int B, int64_t HW, int C, void* stream) {
  RF_CHECK_DT(x_dtype);
  RF_CHECK_DT(y_dtype);
  const int64_t total = (int64_t)B * HW * C;
  if (!gamma != !beta) return RF_EINVAL;  // both (InstanceNorm with affine) or neither (centre only: generic kernel)
  if (gamma && x_dtype == RF_H16 && C % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
      (!y2 || ((uintptr_t)y2 % 16) == 0) && (!residual || ((uintptr_t)residual % 16) == 0)) {
    unsigned gx = min(cdiv(HW * (C / 8), 256), 4096u);
    {  // grid stride (gx * 256 chunks) a multiple of the C / 8 chunks of a pixel: every thread keeps one channel chunk
      const unsigned nch = (unsigned)(C / 8);
      unsigned g = nch, r = 256u % nch;
      while (r) { const unsigned t = g % r; g = r; r = t; }   // gcd(nch, 256)
      const unsigned m = nch / g;
      if (gx >= m) gx = gx / m * m;
    }
    hipLaunchKernelGGL(instnorm_apply_vec_kernel, dim3(gx, B), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream,
                       (const h16_t*)x, (const double*)sums, gamma, beta, eps, residual, act, y, y_dtype, y2, y2_dtype,
                       HW, C);
    return rf_launch_status();
  }
  hipLaunchKernelGGL(instnorm_apply_kernel, dim3(min(cdiv(total, 256), 16384u)), dim3(256), 0, (hipStream_t)stream, x,
                     x_dtype, (const double*)sums, gamma, beta, eps, residual, act, y, y_dtype, y2, y2_dtype, HW, C,
                     total);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// embeddings
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msa_embed_kernel(const int64_t* msa, const int64_t* aa_idx, const float* emb,
                                                        const float* pe, const float* qenc, float* y, int N, int L,
                                                        int D, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = e % D;
    const int64_t r = e / D;  // (b,n,l)
    const int l = r % L, n = (r / L) % N;
    const int64_t b = r / ((int64_t)L * N);
    y[e] = emb[msa[r] * D + c] + pe[aa_idx[b * L + l] * D + c] + qenc[(n == 0 ? 0 : 1) * D + c];
  }
}

extern "C" int rf_msa_embed(const int64_t* msa, const int64_t* aa_idx, const float* emb, const float* pe,
                            const float* qenc, float* y, int B, int N, int L, int D, void* stream) {
  const int64_t total = (int64_t)B * N * L * D;
  hipLaunchKernelGGL(msa_embed_kernel, dim3(min(cdiv(total, 256), 16384u)), dim3(256), 0, (hipStream_t)stream, msa,
                     aa_idx, emb, pe, qenc, y, N, L, D, total);
  return rf_launch_status();
}

__global__ __launch_bounds__(256) void pair_embed_kernel(const int64_t* seq, const int64_t* aa_idx, const float* tl,
                                                         const float* tr, const float* wsep, const float* bias,
                                                         const float* pe, float* y, int L, int D, int64_t total) {
  const int dh = D / 2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = e % D;
    const int64_t r = e / D;  // (b,i,j)
    const int j = r % L, i = (r / L) % L;
    const int64_t b = r / ((int64_t)L * L);
    const int64_t ii = aa_idx[b * L + i], jj = aa_idx[b * L + j];
    const int64_t dd = ii > jj ? ii - jj : jj - ii;
    const float sep = logf((float)(dd + 1));
    const float pos = c < dh ? pe[ii * dh + c] : pe[jj * dh + (c - dh)];
    y[e] = tl[seq[b * L + j] * D + c] + tr[seq[b * L + i] * D + c] + wsep[c] * sep + bias[c] + pos;
  }
}

extern "C" int rf_pair_embed(const int64_t* seq, const int64_t* aa_idx, const float* tl, const float* tr,
                             const float* wsep, const float* bias, const float* pe, float* y, int B, int L, int D,
                             void* stream) {
  const int64_t total = (int64_t)B * L * L * D;
  hipLaunchKernelGGL(pair_embed_kernel, dim3(min(cdiv(total, 256), 16384u)), dim3(256), 0, (hipStream_t)stream, seq,
                     aa_idx, tl, tr, wsep, bias, pe, y, L, D, total);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// generic strided copy / cast, axpby
// ------------------------------------------------------------------------------------------------
struct Copy4 {
  int64_t xs[4], ys[4], dims[4];
};
__global__ __launch_bounds__(256) void copy4d_kernel(const void* x, int x_dt, void* y, int y_dt, Copy4 c, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i3 = e % c.dims[3];
    int64_t t = e / c.dims[3];
    const int64_t i2 = t % c.dims[2];
    t /= c.dims[2];
    const int64_t i1 = t % c.dims[1], i0 = t / c.dims[1];
    st(y, y_dt, i0 * c.ys[0] + i1 * c.ys[1] + i2 * c.ys[2] + i3 * c.ys[3],
       ld(x, x_dt, i0 * c.xs[0] + i1 * c.xs[1] + i2 * c.xs[2] + i3 * c.xs[3]));
  }
}

extern "C" int rf_copy4d(const void* x, int x_dtype, const int64_t xs[4], void* y, int y_dtype, const int64_t ys[4],
                         const int64_t dims[4], void* stream) {
  RF_CHECK_DT(x_dtype);
  RF_CHECK_DT(y_dtype);
  Copy4 c;
  int64_t total = 1;
  for (int i = 0; i < 4; ++i) {
    c.xs[i] = xs[i];
    c.ys[i] = ys[i];
    c.dims[i] = dims[i];
    total *= dims[i];
  }
  if (total <= 0) return RF_EINVAL;
  hipLaunchKernelGGL(copy4d_kernel, dim3(min(cdiv(total, 256), 32768u)), dim3(256), 0, (hipStream_t)stream, x, x_dtype,
                     y, y_dtype, c, total);
  return rf_launch_status();
}

__global__ __launch_bounds__(256) void axpby_kernel(const void* x, int x_dt, float a, const void* z, int z_dt, float b,
                                                    void* y, int y_dt, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float v = a * ld(x, x_dt, e);
    if (z) v += b * ld(z, z_dt, e);
    st(y, y_dt, e, v);
  }
}

// 4 elements per thread and iteration, two independent groups in flight (ld4 / st4: see the top of the file)
__global__ __launch_bounds__(256) void axpby_vec_kernel(const void* x, int x_dt, float a, const void* z, int z_dt, float b,
                                                        void* y, int y_dt, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 2 * stride) {
    const int64_t j = i + stride;
    const bool two = j < n4;
    float4 v0 = ld4(x, x_dt, 4 * i), v1 = two ? ld4(x, x_dt, 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
    if (z) {
      w0 = ld4(z, z_dt, 4 * i);
      if (two) w1 = ld4(z, z_dt, 4 * j);
    }
    st4(y, y_dt, 4 * i, make_float4(a * v0.x + b * w0.x, a * v0.y + b * w0.y, a * v0.z + b * w0.z, a * v0.w + b * w0.w));
    if (two) st4(y, y_dt, 4 * j, make_float4(a * v1.x + b * w1.x, a * v1.y + b * w1.y, a * v1.z + b * w1.z, a * v1.w + b * w1.w));
  }
}

extern "C" int rf_axpby(const void* x, int x_dtype, float a, const void* z, int z_dtype, float b, void* y, int y_dtype,
                        int64_t n, void* stream) {
  RF_CHECK_DT(x_dtype);
  RF_CHECK_DT(y_dtype);
  if (n % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (!z || ((uintptr_t)z % 16) == 0)) {
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(axpby_vec_kernel, dim3(min(cdiv(n4, 512), 16384u)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, a, z,
                       z_dtype, b, y, y_dtype, n4);
    return rf_launch_status();
  }
  hipLaunchKernelGGL(axpby_kernel, dim3(min(cdiv(n, 256), 32768u)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, a, z,
                     z_dtype, b, y, y_dtype, n);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// FAVOR+ softmax-kernel features.  One block per (sequence, head) S.
//   dash layout: [S][n][m_pad] (transposed == 0)  or  [S][m_pad][n] (transposed == 1)
//   x row r of S: x + s0*xs[0] + s1*xs[1] + s2*xs[2] + r*xs[3], S = (s0*n1 + s1)*n2 + s2
// ------------------------------------------------------------------------------------------------
struct FavorP {
  int64_t xs[4];
  int n1, n2;
};
__global__ __launch_bounds__(256) void favor_softmax_kernel(const void* dash, const void* x, FavorP fp, void* y, int dt,
                                                            int n, int m, int m_pad, int dh, int is_query,
                                                            int transposed, float eps) {
  extern __shared__ float sm[];  // diag[n], rowmax[n]
  __shared__ float red[4];
  float* diag = sm;
  float* rmax = sm + n;
  const int64_t S = blockIdx.x;
  const int s2 = S % fp.n2, s1 = (S / fp.n2) % fp.n1;
  const int64_t s0 = S / ((int64_t)fp.n2 * fp.n1);
  const int64_t xb = s0 * fp.xs[0] + s1 * fp.xs[1] + s2 * fp.xs[2];
  const float nrm2 = rsqrtf((float)dh);  // (d^-1/4)^2
  const float ratio = rsqrtf((float)m);
  for (int r = threadIdx.x; r < n; r += 256) {
    float a = 0.f;
    for (int c = 0; c < dh; ++c) {
      const float v = ld(x, dt, xb + r * fp.xs[3] + c);
      a = fmaf(v, v, a);
    }
    diag[r] = 0.5f * a * nrm2;
    rmax[r] = -INFINITY;
  }
  __syncthreads();
  const int64_t base = S * (int64_t)n * m_pad;
  const int64_t tot = (int64_t)n * m_pad;
  // pass 1: maxima.  e runs over the stored layout so loads are coalesced.
  float gmax = -INFINITY;
  if (is_query) {
    // per-row max over the valid features: one wave per row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = wv; r < n; r += 4) {
      float mx = -INFINITY;
      for (int f = lane; f < m; f += 64)
        mx = fmaxf(mx, ld(dash, dt, base + (transposed ? (int64_t)f * n + r : (int64_t)r * m_pad + f)));
      mx = wave_max(mx);
      if (lane == 0) rmax[r] = mx;
    }
    __syncthreads();
  } else {
    for (int64_t e = threadIdx.x; e < tot; e += 256) {
      const int f = transposed ? e / n : e % m_pad;
      if (f < m) gmax = fmaxf(gmax, ld(dash, dt, base + e));
    }
    gmax = block_max(gmax, red);
  }
  for (int64_t e = threadIdx.x; e < tot; e += 256) {
    const int f = transposed ? e / n : e % m_pad;
    const int r = transposed ? e % n : e / m_pad;
    float o = 0.f;
    if (f < m) {
      const float mx = is_query ? rmax[r] : gmax;
      o = ratio * (__expf(ld(dash, dt, base + e) - diag[r] - mx) + eps);
    }
    st(y, dt, base + e, o);
  }
}

extern "C" int rf_favor_softmax_features(const void* dash, const void* x, const int64_t xs[4], int n1, int n2, void* y,
                                         int dtype, int64_t S, int n, int m, int m_pad, int dh, int is_query,
                                         int transposed, float eps, void* stream) {
  RF_CHECK_DT(dtype);
  FavorP fp;
  for (int i = 0; i < 4; ++i) fp.xs[i] = xs[i];
  fp.n1 = n1;
  fp.n2 = n2;
  hipLaunchKernelGGL(favor_softmax_kernel, dim3((unsigned)S), dim3(256), 2 * n * sizeof(float), (hipStream_t)stream,
                     dash, x, fp, y, dtype, n, m, m_pad, dh, is_query, transposed, eps);
  return rf_launch_status();
}

// y[r, c] = num[r, c] / num[r, dh]  (c < dh)
__global__ __launch_bounds__(256) void linattn_norm_kernel(const float* num, int64_t num_ld, void* y, int y_dt,
                                                           int64_t y_ld, int64_t rows, int dh) {
  const int64_t total = rows * dh;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = e % dh;
    const int64_t r = e / dh;
    st(y, y_dt, r * y_ld + c, num[r * num_ld + c] / num[r * num_ld + dh]);
  }
}

extern "C" int rf_linattn_normalize(const float* num, int64_t num_ld, void* y, int y_dtype, int64_t y_ld, int64_t rows,
                                    int dh, void* stream) {
  RF_CHECK_DT(y_dtype);
  const int64_t total = rows * dh;
  hipLaunchKernelGGL(linattn_norm_kernel, dim3(min(cdiv(total, 256), 32768u)), dim3(256), 0, (hipStream_t)stream, num,
                     num_ld, y, y_dtype, y_ld, rows, dh);
  return rf_launch_status();
}

// feat[b,i,j,c0 + c] = msa1d[b,i,c] (c < P2) ; feat[b,i,j,c0+P2+c] = msa1d[b,j,c]
__global__ __launch_bounds__(256) void tile_1d_kernel(const float* m1, void* feat, int dt, int64_t ld_, int c0, int L,
                                                      int P2, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = e % (2 * P2);
    const int64_t r = e / (2 * P2);
    const int j = r % L, i = (r / L) % L;
    const int64_t b = r / ((int64_t)L * L);
    const float v = c < P2 ? m1[(b * L + i) * P2 + c] : m1[(b * L + j) * P2 + (c - P2)];
    st(feat, dt, r * ld_ + c0 + c, v);
  }
}

// 4 channels per thread (P2 % 4 == 0, 16-byte loads, 8/16-byte stores): the scalar form spends ~100 integer instructions per
// 2-byte store
__global__ __launch_bounds__(256) void tile_1d_vec_kernel(const float* m1, void* feat, int dt, int64_t ld_, int c0, int L,
                                                          int P2, int64_t total4) {
  const int cpr = (2 * P2) >> 2;  // 4-channel chunks per (b,i,j)
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % cpr) * 4;
    const int64_t r = e / cpr;
    const int j = r % L, i = (r / L) % L;
    const int64_t b = r / ((int64_t)L * L);
    const float4 v = c < P2 ? *(const float4*)(m1 + (b * L + i) * P2 + c) : *(const float4*)(m1 + (b * L + j) * P2 + (c - P2));
    st4(feat, dt, r * ld_ + c0 + c, v);
  }
}

extern "C" int rf_tile_1d_feats(const float* msa1d, void* feat, int dtype, int64_t feat_ld, int c0, int B, int L, int P2,
                                void* stream) {
  RF_CHECK_DT(dtype);
  if (P2 % 4 == 0 && c0 % 4 == 0 && feat_ld % 4 == 0 && ((uintptr_t)msa1d % 16) == 0 && ((uintptr_t)feat % 16) == 0) {
    const int64_t total4 = (int64_t)B * L * L * (2 * P2 / 4);
    hipLaunchKernelGGL(tile_1d_vec_kernel, dim3(min(cdiv(total4, 256), 32768u)), dim3(256), 0, (hipStream_t)stream, msa1d, feat,
                       dtype, feat_ld, c0, L, P2, total4);
    return rf_launch_status();
  }
  const int64_t total = (int64_t)B * L * L * 2 * P2;
  hipLaunchKernelGGL(tile_1d_kernel, dim3(min(cdiv(total, 256), 32768u)), dim3(256), 0, (hipStream_t)stream, msa1d, feat,
                     dtype, feat_ld, c0, L, P2, total);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Training-mode dropout (the reference's nn.Dropout sites, rf.py:18-28, 76, 217, 265-281, 346, 455, 567,
// 592, 658, 1138; resnet.py:30).  Counter-based: Philox4x32-10 keyed by the caller's 64-bit seed, counter = offset + element / 4,
// word element % 4 -- a mask depends on (seed, offset, element index) only, so a fixed seed reproduces a forward bit for bit on
// any grid, and the host hands every dropout call of a forward its own offset range (no state on the device).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t key, unsigned (&r)[4]) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
  unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
// keep[e] of the mask (seed, offset): uniform 32-bit word >= p * 2^32
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t offset, int64_t e, unsigned thresh) {
  unsigned r[4];
  philox4x32_10(offset + (uint64_t)(e >> 2), seed, r);
  return r[e & 3] >= thresh;
}
static inline unsigned dropout_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

__global__ __launch_bounds__(256) void dropout_kernel(const void* x, void* y, int dt, unsigned thresh, float inv_keep, uint64_t seed,
                                                      uint64_t offset, int64_t n) {
  const int64_t n4 = (n + 3) >> 2;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n4; c += (int64_t)gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10(offset + (uint64_t)c, seed, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = 4 * c + j;
      if (e < n) st(y, dt, e, r[j] >= thresh ? ld(x, dt, e) * inv_keep : 0.f);
    }
  }
}

// include/rfmi.h: rf_dropout
extern "C" int rf_dropout(const void* x, void* y, int dtype, float p, uint64_t seed, uint64_t offset, int64_t n, void* stream) {
  RF_CHECK_DT(dtype);
  if (!x || !y || n < 0 || !(p >= 0.f) || !(p < 1.f)) return RF_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_kernel, dim3(min(cdiv((n + 3) / 4, 256), 16384u)), dim3(256), 0, (hipStream_t)stream, x, y, dtype,
                     dropout_threshold(p), 1.f / (1.f - p), seed, offset, n);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// GraphTransformer attention core: block per (b,i), thread per (h,d) channel (H*d <= 256)
// (DROP: the reference's att_dropout on the attention probabilities, rf.py:658, for the training-mode forward)
// ------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256) void graph_attention_kernel(const void* q, const void* k, const void* v, const void* e,
                                                              int dt, float* out, int L, int H, int d, float scale,
                                                              unsigned thresh, float inv_keep, uint64_t seed, uint64_t offset) {
  extern __shared__ float sm[];  // logits [H][L]
  const int bi = blockIdx.x, b = bi / L;
  const int HD = H * d;
  const int t = threadIdx.x;
  const bool act = t < HD;
  const int h = act ? t / d : 0;
  const float qv = act ? ld(q, dt, (int64_t)bi * HD + t) : 0.f;
  // phase 1: logits[h][j] = scale * sum_d q (k_j + e_ij)
  for (int j = 0; j < L; ++j) {
    float p = 0.f;
    if (act) p = qv * (ld(k, dt, ((int64_t)b * L + j) * HD + t) + ld(e, dt, ((int64_t)bi * L + j) * HD + t));
    // reduce over the d lanes of this head (d is a power of two <= 64 and heads are lane-aligned)
    for (int o = d >> 1; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
    if (act && (t % d) == 0) sm[h * L + j] = p * scale;
  }
  __syncthreads();
  const int lane = t & 63, wv = t >> 6;
  for (int hh = wv; hh < H; hh += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sm[hh * L + j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < L; j += 64) s += __expf(sm[hh * L + j] - mx);
    const float inv = 1.f / wave_sum(s);
    for (int j = lane; j < L; j += 64) {
      float pr = __expf(sm[hh * L + j] - mx) * inv;
      if (DROP)  // element index of att[b, h, i, j] in the reference's [b, h, i, j] map
        pr = dropout_keep(seed, offset, (((int64_t)b * H + hh) * L + (bi % L)) * L + j, thresh) ? pr * inv_keep : 0.f;
      sm[hh * L + j] = pr;
    }
  }
  __syncthreads();
  if (act) {
    float a = 0.f;
    for (int j = 0; j < L; ++j)
      a = fmaf(sm[h * L + j], ld(v, dt, ((int64_t)b * L + j) * HD + t) + ld(e, dt, ((int64_t)bi * L + j) * HD + t), a);
    out[(int64_t)bi * HD + t] = a;
  }
}

extern "C" int rf_graph_attention(const void* q, const void* k, const void* v, const void* e, int dtype, float* out,
                                  int B, int L, int H, int d, float scale, void* stream) {
  RF_CHECK_DT(dtype);
  if (H * d > 256 || d > 64 || (d & (d - 1)) != 0) return RF_EINVAL;
  const size_t lds = (size_t)H * L * sizeof(float);
  if (lds > 64 * 1024) return RF_EINVAL;
  hipLaunchKernelGGL(graph_attention_kernel<false>, dim3(B * L), dim3(256), lds, (hipStream_t)stream, q, k, v, e, dtype, out, L,
                     H, d, scale, 0u, 1.f, (uint64_t)0, (uint64_t)0);
  return rf_launch_status();
}

// training-mode form: dropout(p) on the attention probabilities (rf.py:658), mask (seed, offset) over the [B, H, L, L] map
extern "C" int rf_graph_attention_dropout(const void* q, const void* k, const void* v, const void* e, int dtype, float* out,
                                          int B, int L, int H, int d, float scale, float p, uint64_t seed, uint64_t offset,
                                          void* stream) {
  RF_CHECK_DT(dtype);
  if (H * d > 256 || d > 64 || (d & (d - 1)) != 0 || !(p >= 0.f) || !(p < 1.f)) return RF_EINVAL;
  const size_t lds = (size_t)H * L * sizeof(float);
  if (lds > 64 * 1024) return RF_EINVAL;
  hipLaunchKernelGGL(graph_attention_kernel<true>, dim3(B * L), dim3(256), lds, (hipStream_t)stream, q, k, v, e, dtype, out, L,
                     H, d, scale, dropout_threshold(p), 1.f / (1.f - p), seed, offset);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// distance-masked attention map: block per (b,i); att[b,h,i,:]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dist_att_kernel(const float* q, const float* k, const float* xyz,
                                                       const float* bins, void* att, int dt, int L, int H, int dq) {
  extern __shared__ float sm[];  // [H][L]
  const int bi = blockIdx.x, b = bi / L, i = bi % L;
  const float cx = xyz[((int64_t)bi * 3 + 1) * 3 + 0], cy = xyz[((int64_t)bi * 3 + 1) * 3 + 1],
              cz = xyz[((int64_t)bi * 3 + 1) * 3 + 2];
  for (int e = threadIdx.x; e < H * L; e += 256) {
    const int h = e / L, j = e % L;
    const float* qr = q + ((int64_t)bi * H + h) * dq;
    const float* kr = k + (((int64_t)b * L + j) * H + h) * dq;
    float a = 0.f;
    for (int c = 0; c < dq; ++c) a = fmaf(qr[c], kr[c], a);
    const float* cj = xyz + (((int64_t)b * L + j) * 3 + 1) * 3;
    const float dx = cx - cj[0], dy = cy - cj[1], dz = cz - cj[2];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    sm[e] = a + (dist < bins[h] ? 0.f : -1e9f);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int h = wv; h < H; h += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sm[h * L + j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < L; j += 64) s += __expf(sm[h * L + j] - mx);
    const float inv = 1.f / wave_sum(s);
    for (int j = lane; j < L; j += 64)
      st(att, dt, (((int64_t)b * H + h) * L + i) * L + j, __expf(sm[h * L + j] - mx) * inv);
  }
}

extern "C" int rf_dist_masked_attention(const float* q, const float* k, const float* xyz, const float* bins, void* att,
                                        int att_dtype, int B, int L, int H, int dq, void* stream) {
  RF_CHECK_DT(att_dtype);
  const size_t lds = (size_t)H * L * sizeof(float);
  if (lds > 64 * 1024) return RF_EINVAL;
  hipLaunchKernelGGL(dist_att_kernel, dim3(B * L), dim3(256), lds, (hipStream_t)stream, q, k, xyz, bins, att, att_dtype,
                     L, H, dq);
  return rf_launch_status();
}

// y[r, :] = x[r, :] * w[r]   (y may alias x)
__global__ __launch_bounds__(256) void scale_rows_kernel(const void* x, void* y, int dt, const float* w, int64_t total, int D) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    st(y, dt, e, ld(x, dt, e) * w[e / D]);
}

extern "C" int rf_scale_rows(const void* x, void* y, int dtype, const float* w, int64_t rows, int D, void* stream) {
  RF_CHECK_DT(dtype);
  const int64_t total = rows * D;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(min(cdiv(total, 256), 32768u)), dim3(256), 0, (hipStream_t)stream, x, y,
                     dtype, w, total, D);
  return rf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// small helpers that keep ATen kernels off the forward path: fill, input validation, one-hot, sequence-separation
// feature, stand-alone positional encodings
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_kernel(void* y, int dt, float v, int64_t n) {
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (dt == RF_F32) {
    float4* y4 = (float4*)y;
    const int64_t n4 = n >> 2;
    for (int64_t e = t0; e < n4; e += stride) y4[e] = make_float4(v, v, v, v);
    for (int64_t e = (n4 << 2) + t0; e < n; e += stride) ((float*)y)[e] = v;
  } else {
    const unsigned h = f2h(v), w = h | (h << 16);
    uint4* y8 = (uint4*)y;
    const int64_t n8 = n >> 3;
    for (int64_t e = t0; e < n8; e += stride) y8[e] = make_uint4(w, w, w, w);
    for (int64_t e = (n8 << 3) + t0; e < n; e += stride) ((h16_t*)y)[e] = (h16_t)h;
  }
}

extern "C" int rf_fill(void* y, int dtype, float value, int64_t n, void* stream) {
  RF_CHECK_DT(dtype);
  if (n <= 0) return 0;
  if ((uintptr_t)y % 16) return RF_EALIGN;
  hipLaunchKernelGGL(fill_kernel, dim3(min(cdiv(n, 256 * 8), 8192u)), dim3(256), 0, (hipStream_t)stream, y, dtype, value, n);
  return rf_launch_status();
}

// flags[0] = 1 if a token is outside [0, d_input); flags[1] = 1 if an aa_idx is outside [0, max_len); flags[2] = 1 if
// aa_idx is not strictly increasing along a sample.  Plain idempotent stores (every writer stores 1): no atomics.
// flags must be zeroed by the caller.
__global__ __launch_bounds__(256) void check_inputs_kernel(const int64_t* msa, int64_t n_msa, const int64_t* seq, int64_t n_seq,
                                                           const int64_t* aa_idx, int64_t n_idx, int L, int d_input,
                                                           int max_len, int32_t* flags) {
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  bool bad_tok = false, bad_idx = false, not_mono = false;
  for (int64_t e = t0; e < n_msa; e += stride) bad_tok |= msa[e] < 0 || msa[e] >= d_input;
  for (int64_t e = t0; e < n_seq; e += stride) bad_tok |= seq[e] < 0 || seq[e] >= d_input;
  for (int64_t e = t0; e < n_idx; e += stride) {
    const int64_t v = aa_idx[e];
    bad_idx |= v < 0 || v >= max_len;
    if (e % L) not_mono |= aa_idx[e - 1] >= v;
  }
  if (bad_tok) flags[0] = 1;
  if (bad_idx) flags[1] = 1;
  if (not_mono) flags[2] = 1;
}

extern "C" int rf_check_inputs(const int64_t* msa, int64_t n_msa, const int64_t* seq, int64_t n_seq, const int64_t* aa_idx,
                               int64_t n_idx, int L, int d_input, int max_len, int32_t* flags, void* stream) {
  if (!flags || L <= 0) return RF_EINVAL;
  if (!msa) n_msa = 0;
  if (!seq) n_seq = 0;
  if (!aa_idx) n_idx = 0;
  const int64_t n = n_msa > n_idx ? (n_msa > n_seq ? n_msa : n_seq) : (n_idx > n_seq ? n_idx : n_seq);
  hipLaunchKernelGGL(check_inputs_kernel, dim3(min(cdiv(n > 0 ? n : 1, 256), 1024u)), dim3(256), 0, (hipStream_t)stream,
                     msa, n_msa, seq, n_seq, aa_idx, n_idx, L, d_input, max_len, flags);
  return rf_launch_status();
}

// y[r, col0 + c] = (idx[r] == c) for c < n_classes   (F.one_hot(seq, 21), rf.py:1276)
__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* idx, void* y, int dt, int64_t ld_, int col0, int nc,
                                                     int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / nc;
    const int c = e % nc;
    st(y, dt, r * ld_ + col0 + c, idx[r] == c ? 1.f : 0.f);
  }
}

extern "C" int rf_onehot(const int64_t* idx, void* y, int dtype, int64_t y_ld, int col0, int n_classes, int64_t rows,
                         void* stream) {
  RF_CHECK_DT(dtype);
  const int64_t total = rows * n_classes;
  if (total <= 0) return RF_EINVAL;
  hipLaunchKernelGGL(onehot_kernel, dim3(min(cdiv(total, 256), 4096u)), dim3(256), 0, (hipStream_t)stream, idx, y, dtype, y_ld,
                     col0, n_classes, total);
  return rf_launch_status();
}

// y[(b,i,j), col] = clamp(sign(d) * log(|d| + 1), 0, 5.5), d = idx[b,i] - idx[b,j]   (rf.py:746-749: signed, then clamped >= 0)
__global__ __launch_bounds__(256) void seqsep_kernel(const int64_t* aa_idx, void* y, int dt, int64_t ld_, int col, int L,
                                                     int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = e % L, i = (e / L) % L;
    const int64_t b = e / ((int64_t)L * L);
    const int64_t d = aa_idx[b * L + i] - aa_idx[b * L + j];
    const float v = d > 0 ? fminf(logf((float)(d + 1)), 5.5f) : 0.f;
    st(y, dt, e * ld_ + col, v);
  }
}

extern "C" int rf_seqsep_feature(const int64_t* aa_idx, void* y, int dtype, int64_t y_ld, int col, int B, int L, void* stream) {
  RF_CHECK_DT(dtype);
  const int64_t total = (int64_t)B * L * L;
  if (total <= 0) return RF_EINVAL;
  hipLaunchKernelGGL(seqsep_kernel, dim3(min(cdiv(total, 256), 4096u)), dim3(256), 0, (hipStream_t)stream, aa_idx, y, dtype,
                     y_ld, col, L, total);
  return rf_launch_status();
}

// Stand-alone positional encodings (rf.py:72-76 and rf.py:95-103):
//   two_d == 0: y[b,n,l,:] = x[b,n,l,:] + pe[aa_idx[b,l], :]                      (x: [B, N, L, D], pe: [max_len, D])
//   two_d != 0: y[b,i,j,:] = x[b,i,j,:] + [pe[aa_idx[b,i]] | pe[aa_idx[b,j]]]      (x: [B, L, L, D], N == L, pe: [max_len, D/2])
__global__ __launch_bounds__(256) void add_pos_enc_kernel(const float* x, const int64_t* aa_idx, const float* pe, float* y,
                                                          int N, int L, int D, int two_d, int64_t total) {
  const int dh = D / 2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = e % D;
    const int64_t r = e / D;
    const int l = r % L, n = (r / L) % N;
    const int64_t b = r / ((int64_t)L * N);
    float p;
    if (two_d)
      p = c < dh ? pe[aa_idx[b * L + n] * dh + c] : pe[aa_idx[b * L + l] * dh + (c - dh)];
    else
      p = pe[aa_idx[b * L + l] * D + c];
    y[e] = x[e] + p;
  }
}

extern "C" int rf_add_pos_enc(const float* x, const int64_t* aa_idx, const float* pe, float* y, int B, int N, int L, int D,
                              int two_d, void* stream) {
  if (two_d && (N != L || D % 2)) return RF_EINVAL;
  const int64_t total = (int64_t)B * N * L * D;
  if (total <= 0) return RF_EINVAL;
  hipLaunchKernelGGL(add_pos_enc_kernel, dim3(min(cdiv(total, 256), 16384u)), dim3(256), 0, (hipStream_t)stream, x, aa_idx, pe,
                     y, N, L, D, two_d, total);
  return rf_launch_status();
}

extern "C" int rf_version(void) { return 4; }
#ifdef RF_H16_IS_F16
extern "C" const char* rf_build_info(void) { return "librfmi_f16 gfx950 (MI355X) round-3: 16-bit operand type = IEEE fp16"; }
#else
extern "C" const char* rf_build_info(void) { return "librfmi gfx950 (MI355X) round-3: 16-bit operand type = bfloat16"; }
#endif
// the dtype code this build accepts for 16-bit tensors (RF_BF16 in librfmi.so, RF_F16 in librfmi_f16.so)
extern "C" int rf_h16_dtype(void) { return RF_H16; }
