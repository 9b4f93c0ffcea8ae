// rf_gemm: strided / batched / chunked-K "TN" GEMM engine for gfx950 (MI355X).
//
// Every dense contraction of the RoseTTAFold forward path goes through here (nn.Linear
// rf.py:195-281; einsums rf.py:254,257,424,592,916; 3x3 dilated convs rf.py:452-457, resnet.py:19-38
// as implicit GEMM).  Two code paths share one descriptor:
//   * bf16 path: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  256-thread workgroups (4 waves as 2x2),
//     BMxBN output tile, BK-deep K steps, both operand tiles staged global->LDS with
//     global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier per K step.
//     LDS image is lane-linear (the DMA writes base + lane*16), so the bank-conflict swizzle is
//     applied to the per-lane SOURCE address and again on the ds_read_b128 fragment reads.
//     The MFMA is issued with the weight tile as the A operand and the activation tile as the
//     B operand, so each lane ends up with 4 consecutive output columns of one output row:
//     the epilogue (bias / activation / fp32 residual add) stores 8 B (bf16) or 16 B (fp32) per lane.
//     Workgroup ids are remapped so the 8 XCDs each walk a contiguous run of tiles
//     (neighbouring N tiles of one activation row panel share an L2).
//   * f32 path: exact fp32 FMA tiles (parity mode and the SE(3) module, which the reference
//     forces to fp32: se3_modules.py:164).
#include <type_traits>

#include "common.h"

__device__ __attribute__((aligned(16))) unsigned int g_rf_zero16[4];  // zero source for masked DMA lanes

struct GemmP {
  rf_gemm_desc d;
  int tilesM, tilesN;
  int vec_store;  // 1: 4-wide stores are legal for this C layout
  int dbg;        // timing experiments: 1 = skip epilogue stores, 2 = skip the K loop
  int stage_epi;  // 1: C tile goes through LDS and is written as whole rows (16-byte coalesced stores)
  int nt_store;   // 1: streaming (nontemporal) stores for large outputs
  int c_pow2, c_rsh, c_csh;  // C split factors are powers of two: shift amounts
};

__device__ __forceinline__ int64_t split_off(int idx, int rc, int64_t ro, int64_t ri) {
  return rc > 0 ? (int64_t)(idx / rc) * ro + (int64_t)(idx % rc) * ri : (int64_t)idx * ri;
}

__device__ __forceinline__ void batch_decode(const rf_gemm_desc& d, int z, int& z0, int& z1, int& z2) {
  z2 = z % d.nb2;
  int t = z / d.nb2;
  z1 = t % d.nb1;
  z0 = t / d.nb1;
}

__device__ __forceinline__ float elu_call(float x) { return x > 0.f ? x : __expf(x) - 1.f; }

__device__ __forceinline__ float apply_act(float v, int act, float eps, bool valid) {
  if (act == RF_ACT_RELU) return fmaxf(v, 0.f);
  if (act == RF_ACT_ELU) return elu_call(v);
  if (act == RF_ACT_RELU_EPS) return valid ? fmaxf(v, 0.f) + eps : 0.f;
  return v;
}

// rare path: per-element stores when 4-wide stores are not legal for the C layout / tail columns
__device__ __forceinline__ void store_scalar4(const rf_gemm_desc& d, int64_t c_row, int n, float v0, float v1,
                                                        float v2, float v3) {
  const float v[4] = {v0, v1, v2, v3};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n + e >= d.N) break;
    const int64_t o = c_row + split_off(n + e, d.c_cc, d.c_co, 1);
    float x = v[e];
    if (d.residual) x += d.residual[o];
    st(d.C, d.c_dtype, o, x);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
// dynamic LDS per workgroup: the double-buffered operand tiles, or one fp32 row group of the staged epilogue
__host__ __device__ constexpr int lds_bytes_for(int bm, int bn, int bk, int wgm, int ns) {
  const int pipe = ns * (bm + bn) * bk * 2 + (ns > 2 ? 1024 : 0), epi = wgm * 16 * (bn * 4 + 16);
  return pipe > epi ? pipe : epi;
}

template <int BK>
__device__ __forceinline__ int swz(int row) {
  if constexpr (BK == 64)
    return row & 7;  // 8 x 16B chunks per 128-B row
  else
    return (0x78 >> (((row >> 2) & 3) * 2)) & 3;  // 4 chunks per 64-B row: g = {0,2,3,1}[(row>>2)&3]
}

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int BK, int WGM, int WGN, int NS, int AMODE>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_kernel(const GemmP p) {
  constexpr int NW = WGM * WGN;  // waves per workgroup, arranged WGM x WGN over the tile
  constexpr int SPR = BK / 8;    // 16-byte slots per tile row
  constexpr int A_INSTR = BM * SPR / 64, B_INSTR = BN * SPR / 64;  // wave-level DMA instructions per tile
  constexpr int A_PW = (A_INSTR + NW - 1) / NW, B_PW = (B_INSTR + NW - 1) / NW;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int DUMP_OFF = NS * STAGE_BYTES;  // 1 KiB sink for the padding DMAs of the NS > 2 ring
  constexpr int GLDS = A_PW + B_PW;           // DMA instructions per wave per stage (uniform when NS > 2)
  constexpr int TM = BM / WGM, TN = BN / WGN;  // wave tile
  constexpr int WM = TM / 16, WN = TN / 16;    // 16x16 MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const rf_gemm_desc& d = p.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) % NW;
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware bijective remap of the 1-D grid (round-robin dispatch puts block b on XCD b%8)
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tn = lid % p.tilesN;
  const int t2 = lid / p.tilesN;
  const int tm = t2 % p.tilesM;
  const int z = t2 / p.tilesM;
  int z0, z1, z2;
  batch_decode(d, z, z0, z1, z2);
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16_t* Ab = (const bf16_t*)d.A + (AMODE == RF_AMODE_CONV3X3 ? 0 : z0 * d.a_bs[0] + z1 * d.a_bs[1] + z2 * d.a_bs[2]);
  const bf16_t* Bb = (const bf16_t*)d.B + z0 * d.b_bs[0] + z1 * d.b_bs[1] + z2 * d.b_bs[2];

  // ---- per-lane staging state -------------------------------------------------------------
  const bf16_t* a_src[A_PW];
  const bf16_t* b_src[B_PW];
  int a_ij[A_PW];  // conv: (i << 16) | j of the row's pixel
#pragma unroll
  for (int t = 0; t < A_PW; ++t) {
    const int slot = (t * NW + wave) * 64 + lane;
    int m = m0 + slot / SPR;
    m = m < d.M ? m : d.M - 1;
    if constexpr (AMODE == RF_AMODE_CONV3X3) {
      const int hw = d.conv_h * d.conv_w;
      const int pix = m % hw;
      a_ij[t] = ((pix / d.conv_w) << 16) | (pix % d.conv_w);
      a_src[t] = Ab + (int64_t)m * d.conv_c;
    } else {
      a_ij[t] = 0;
      a_src[t] = Ab + split_off(m, d.a_rc, d.a_ro, d.a_ri);
    }
  }
#pragma unroll
  for (int t = 0; t < B_PW; ++t) {
    const int slot = (t * NW + wave) * 64 + lane;
    int n = n0 + slot / SPR;
    n = n < d.N ? n : d.N - 1;
    b_src[t] = Bb + split_off(n, d.b_rc, d.b_ro, d.b_ri);
  }
  // logical K chunk this lane fetches (same for all its slots: see header comment)
  const int row_in_instr = lane / SPR;
  const int c_log = (lane % SPR) ^ swz<BK>(row_in_instr);
  int kpos = c_log * 8;                    // logical k of this lane's chunk in the current K step
  int kq = kpos / d.kc, kr = kpos % d.kc;  // chunk index / offset within chunk
  int a_koff = 0, b_koff = kq * (int)d.b_ko + kr, cdi = 0, cdj = 0;
  auto a_koff_update = [&]() {
    if constexpr (AMODE == RF_AMODE_CONV3X3) {
      cdi = (kq / 3 - 1) * d.conv_dil;
      cdj = (kq % 3 - 1) * d.conv_dil;
      a_koff = (cdi * d.conv_w + cdj) * d.conv_c + kr;
    } else {
      a_koff = kq * (int)d.a_ko + kr;
    }
  };
  a_koff_update();
  const bf16_t* const zsrc = (const bf16_t*)g_rf_zero16;

  auto stage = [&](int buf) {
    char* a_lds = smem + buf * STAGE_BYTES;
    char* b_lds = a_lds + A_BYTES;
    const bool kvalid = kpos < d.K;
#pragma unroll
    for (int t = 0; t < A_PW; ++t) {
      const int instr = t * NW + wave;
      if ((A_INSTR % NW == 0) || instr < A_INSTR) {
        bool ok = kvalid;
        if constexpr (AMODE == RF_AMODE_CONV3X3) {
          const int ii = (a_ij[t] >> 16) + cdi, jj = (a_ij[t] & 0xffff) + cdj;
          ok = ok && ii >= 0 && ii < d.conv_h && jj >= 0 && jj < d.conv_w;
        }
        const bf16_t* src = ok ? a_src[t] + a_koff : zsrc;
        glds16(src, a_lds + instr * 1024);
      } else if constexpr (NS > 2) {
        glds16(zsrc, smem + DUMP_OFF);  // keep the per-wave DMA count uniform for the counted vmcnt waits
      }
    }
#pragma unroll
    for (int t = 0; t < B_PW; ++t) {
      const int instr = t * NW + wave;
      if ((B_INSTR % NW == 0) || instr < B_INSTR) {
        const bf16_t* src = kvalid ? b_src[t] + b_koff : zsrc;
        glds16(src, b_lds + instr * 1024);
      } else if constexpr (NS > 2) {
        glds16(zsrc, smem + DUMP_OFF);
      }
    }
    // advance this lane's K cursor by one step
    kpos += BK;
    kr += BK;
    a_koff += BK;
    b_koff += BK;
    if (kr >= d.kc) {
      do {
        kr -= d.kc;
        ++kq;
      } while (kr >= d.kc);
      b_koff = kq * (int)d.b_ko + kr;
      a_koff_update();
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int nk = (d.K + BK - 1) / BK;
  // DMA ring: NS stages, loads run NS-1 K steps ahead of the MFMAs.  A stage is consumed after (a) this wave's
  // counted s_waitcnt vmcnt(GLDS * stages still allowed in flight) and (b) the workgroup barrier behind it; it is
  // refilled one iteration after its last ds_read (every wave has passed the next barrier by then).
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nk) stage(s0);
  const int nk_run = (p.dbg & 2) ? 1 : nk;
  for (int kt = 0; kt < nk_run; ++kt) {
    if constexpr (NS == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      const int younger = nk - 1 - kt < NS - 2 ? nk - 1 - kt : NS - 2;
      if (younger >= 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GLDS) : "memory");
      else if (younger == 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GLDS) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (kt + NS - 1 < nk) stage((kt + NS - 1) % NS);
    const char* a_lds = smem + (kt % NS) * STAGE_BYTES;
    const char* b_lds = a_lds + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 af[WM], bfr[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = wm * TM + i * 16 + fr;
        af[i] = *(const bf16x8*)(a_lds + (row * SPR + ((kk * 4 + fq) ^ swz<BK>(row))) * 16);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int row = wn * TN + j * 16 + fr;
        bfr[j] = *(const bf16x8*)(b_lds + (row * SPR + ((kk * 4 + fq) ^ swz<BK>(row))) * 16);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          // weight tile as MFMA-A, activation tile as MFMA-B: D[n_local][m_local]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = ..+fr, n = ..+4*fq --------------------------------
  if ((p.dbg & 1) && acc[0][0][0] != 12345.678f) return;
  const int64_t c_z = z0 * d.c_bs[0] + z1 * d.c_bs[1] + z2 * d.c_bs[2];
  int64_t c_col[WN];
  float4 bias_c[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + wn * TN + j * 16 + 4 * fq;
    c_col[j] = split_off(n, d.c_cc, d.c_co, 1);
    bias_c[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias_mode == RF_BIAS_COL) {
      if (n + 3 < d.N) {
        if (p.vec_store)
          bias_c[j] = *(const float4*)(d.bias + n);  // N % 4 == 0 and the bias vector is 16-byte aligned (checked on the host)
        else
          bias_c[j] = make_float4(d.bias[n], d.bias[n + 1], d.bias[n + 2], d.bias[n + 3]);
      } else {
        if (n < d.N) bias_c[j].x = d.bias[n];
        if (n + 1 < d.N) bias_c[j].y = d.bias[n + 1];
        if (n + 2 < d.N) bias_c[j].z = d.bias[n + 2];
      }
    }
  }
  if (p.stage_epi) {
    // Staged epilogue: the scattered 8-byte-per-lane stores of the MFMA layout are store-issue bound (they cost
    // ~2/3 of a short-K GEMM); instead the C tile is written to LDS (the operand buffers are free now) and every
    // thread then moves whole 16-byte chunks of consecutive rows -> full-line coalesced global stores.
    auto staged = [&](auto esz_tag) {
      constexpr int ESZ = decltype(esz_tag)::value;
      constexpr int NT = 64 * NW;
      constexpr int LDS_CAP = lds_bytes_for(BM, BN, BK, WGM, NS) - (NS > 2 ? 1024 : 0);
      constexpr int PITCH = BN * ESZ + 16;
      constexpr int IPMAX = LDS_CAP / (WGM * 16 * PITCH);
      static_assert(IPMAX >= 1, "C tile row group does not fit the operand buffers");
      constexpr int IP = IPMAX < WM ? IPMAX : WM;
      constexpr int CPR = BN * ESZ / 16;  // 16-byte chunks per tile row
      constexpr int EPC = 16 / ESZ;       // elements per chunk
#pragma unroll
      for (int i0 = 0; i0 < WM; i0 += IP) {
        __syncthreads();  // operand tiles (first pass) / previous pass fully consumed
        // staged row index lr = (ii*WGM + wm)*16 + fr: decodes with shifts only on the read side.
        // Common epilogues (no activation / ReLU, no row bias) take the branch-free form max(fma(acc, alpha, bias), lo).
        const bool simple = (d.act == RF_ACT_NONE || d.act == RF_ACT_RELU) && d.bias_mode != RF_BIAS_ROW;
        const float lo = d.act == RF_ACT_RELU ? 0.f : -INFINITY;
        const float alpha = d.alpha;
        const bool raw = simple && d.act == RF_ACT_NONE && d.bias_mode == RF_BIAS_NONE && alpha == 1.0f;
#pragma unroll
        for (int ii = 0; ii < IP; ++ii) {
          const int i = i0 + ii;
          if (i < WM) {
            const int m = m0 + wm * TM + i * 16 + fr;
            char* lrow = smem + ((ii * WGM + wm) * 16 + fr) * PITCH;
            if (raw) {  // alpha = 1, no bias, no activation (e.g. the q|k|v projections): convert and store
#pragma unroll
              for (int j = 0; j < WN; ++j) {
                const int nl = wn * TN + j * 16 + 4 * fq;
                if constexpr (ESZ == 4) {
                  *(f32x4*)(lrow + nl * 4) = acc[i][j];
                } else {
                  uint2 o;
                  o.x = (unsigned)f2bf(acc[i][j][0]) | ((unsigned)f2bf(acc[i][j][1]) << 16);
                  o.y = (unsigned)f2bf(acc[i][j][2]) | ((unsigned)f2bf(acc[i][j][3]) << 16);
                  *(uint2*)(lrow + nl * 2) = o;
                }
              }
            } else if (simple) {
#pragma unroll
              for (int j = 0; j < WN; ++j) {
                const int nl = wn * TN + j * 16 + 4 * fq;
                const float v0 = fmaxf(fmaf(acc[i][j][0], alpha, bias_c[j].x), lo);
                const float v1 = fmaxf(fmaf(acc[i][j][1], alpha, bias_c[j].y), lo);
                const float v2 = fmaxf(fmaf(acc[i][j][2], alpha, bias_c[j].z), lo);
                const float v3 = fmaxf(fmaf(acc[i][j][3], alpha, bias_c[j].w), lo);
                if constexpr (ESZ == 4) {
                  *(float4*)(lrow + nl * 4) = make_float4(v0, v1, v2, v3);
                } else {
                  uint2 o;
                  o.x = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
                  o.y = (unsigned)f2bf(v2) | ((unsigned)f2bf(v3) << 16);
                  *(uint2*)(lrow + nl * 2) = o;
                }
              }
            } else {
              const float bias_m = (d.bias_mode == RF_BIAS_ROW && m < d.M) ? d.bias[m] : 0.f;
#pragma unroll
              for (int j = 0; j < WN; ++j) {
                const int nl = wn * TN + j * 16 + 4 * fq;
                const int n = n0 + nl;
                float v[4];
                const float bc[4] = {bias_c[j].x, bias_c[j].y, bias_c[j].z, bias_c[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float x = acc[i][j][e] * alpha + bc[e] + bias_m;
                  v[e] = apply_act(x, d.act, d.act_eps, (d.act_nvalid < 0 ? m < -d.act_nvalid : n + e < d.act_nvalid));
                }
                if constexpr (ESZ == 4) {
                  *(float4*)(lrow + nl * 4) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                  uint2 o;
                  o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                  o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                  *(uint2*)(lrow + nl * 2) = o;
                }
              }
            }
          }
        }
        __syncthreads();
        if constexpr (ESZ == 4) {
          if (d.ln_out) {
            // Fused "residual add + LayerNorm of the NEXT sub-layer": this tile spans complete rows (tilesN == 1), so
            // one wave per staged row adds the residual, streams the fp32 row out, and emits the normalised row
            // (the next GEMM's A operand) in the same pass -- the separate LayerNorm read of the stream disappears.
            constexpr int ROWS_LN = WGM * IP * 16;
            const int nch = d.N >> 2;
            for (int lr = wave; lr < ROWS_LN; lr += NW) {
              const int w = (lr >> 4) % WGM;
              const int i = i0 + lr / (WGM * 16);
              const int m = m0 + w * TM + i * 16 + (lr & 15);
              if (i >= WM || m >= d.M) continue;
              const int64_t c_row = c_z + (int64_t)m * d.c_ri;
              float4 v[2];
              float sum = 0.f;
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const int c = lane + 64 * t;
                v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < nch) {
                  v[t] = *(const float4*)(smem + lr * PITCH + c * 16);
                  if (d.residual) {
                    const float4 r = *(const float4*)(d.residual + c_row + 4 * c);
                    v[t].x += r.x; v[t].y += r.y; v[t].z += r.z; v[t].w += r.w;
                  }
                  if (p.nt_store)
                    __builtin_nontemporal_store(*(f32x4*)&v[t], (f32x4*)((float*)d.C + c_row + 4 * c));
                  else
                    *(float4*)((float*)d.C + c_row + 4 * c) = v[t];
                  sum += (v[t].x + v[t].y) + (v[t].z + v[t].w);
                }
              }
              const float mean = wave_sum(sum) / d.N;
              float q = 0.f;
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const int c = lane + 64 * t;
                if (c < nch) {
                  const float a = v[t].x - mean, b = v[t].y - mean, cc = v[t].z - mean, dd = v[t].w - mean;
                  q += (a * a + b * b) + (cc * cc + dd * dd);
                }
              }
              const float rstd = rsqrtf(wave_sum(q) / d.N + d.ln_eps);
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const int c = lane + 64 * t;
                if (c < nch) {
                  const float4 g = *(const float4*)(d.ln_gamma + 4 * c), be = *(const float4*)(d.ln_beta + 4 * c);
                  const float o0 = (v[t].x - mean) * rstd * g.x + be.x, o1 = (v[t].y - mean) * rstd * g.y + be.y;
                  const float o2 = (v[t].z - mean) * rstd * g.z + be.z, o3 = (v[t].w - mean) * rstd * g.w + be.w;
                  uint2 wv;
                  wv.x = (unsigned)f2bf(o0) | ((unsigned)f2bf(o1) << 16);
                  wv.y = (unsigned)f2bf(o2) | ((unsigned)f2bf(o3) << 16);
                  *(uint2*)((bf16_t*)d.ln_out + (int64_t)m * d.N + 4 * c) = wv;
                }
              }
            }
            continue;
          }
        }
        // each thread walks 16-byte chunks idx = tid, tid+NT, ... of the staged rows; (row, chunk) advance
        // incrementally (no divisions in the loop).  Large outputs are streamed with nontemporal stores.
        constexpr int ROWS = WGM * IP * 16;
        constexpr int DR = NT / CPR, DC = NT % CPR;
        const bool plain_c = d.c_rc <= 0 && d.c_cc <= 0;
        int lr = tid / CPR, c = tid % CPR;
        for (; lr < ROWS; lr += DR) {
          const int w = (lr >> 4) % WGM;
          const int i = i0 + lr / (WGM * 16);
          const int m = m0 + w * TM + i * 16 + (lr & 15);
          const int n = n0 + c * EPC;
          if (i < WM && m < d.M && n < d.N) {
            int64_t c_off;
            if (plain_c) {
              c_off = c_z + (int64_t)m * d.c_ri + n;
            } else if (p.c_pow2) {  // both split factors are powers of two: shifts instead of divisions
              const int64_t ro = d.c_rc > 0 ? (int64_t)(m >> p.c_rsh) * d.c_ro + (int64_t)(m & (d.c_rc - 1)) * d.c_ri
                                            : (int64_t)m * d.c_ri;
              const int64_t co = d.c_cc > 0 ? (int64_t)(n >> p.c_csh) * d.c_co + (n & (d.c_cc - 1)) : n;
              c_off = c_z + ro + co;
            } else {
              c_off = c_z + split_off(m, d.c_rc, d.c_ro, d.c_ri) + split_off(n, d.c_cc, d.c_co, 1);
            }
            const char* src = smem + lr * PITCH + c * 16;
            if constexpr (ESZ == 4) {
              float4 v = *(const float4*)src;
              if (d.residual) {
                const float4 r = *(const float4*)(d.residual + c_off);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
              }
              if (p.nt_store)
                __builtin_nontemporal_store(*(f32x4*)&v, (f32x4*)((float*)d.C + c_off));
              else
                *(float4*)((float*)d.C + c_off) = v;
            } else {
              if (p.nt_store)
                __builtin_nontemporal_store(*(const f32x4*)src, (f32x4*)((bf16_t*)d.C + c_off));
              else
                *(uint4*)((bf16_t*)d.C + c_off) = *(const uint4*)src;
            }
          }
          c += DC;
          if (c >= CPR) {
            c -= CPR;
            ++lr;
          }
        }
      }
    };
    if (d.c_dtype == RF_F32)
      staged(std::integral_constant<int, 4>{});
    else
      staged(std::integral_constant<int, 2>{});
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int m = m0 + wm * TM + i * 16 + fr;
    if (m >= d.M) continue;
    const int64_t c_row = c_z + split_off(m, d.c_rc, d.c_ro, d.c_ri);
    const float bias_m = d.bias_mode == RF_BIAS_ROW ? d.bias[m] : 0.f;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + wn * TN + j * 16 + 4 * fq;
      if (n >= d.N) continue;
      float v[4];
      const float bc[4] = {bias_c[j].x, bias_c[j].y, bias_c[j].z, bias_c[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = acc[i][j][e] * d.alpha + bc[e] + bias_m;
        v[e] = apply_act(x, d.act, d.act_eps, (d.act_nvalid < 0 ? m < -d.act_nvalid : n + e < d.act_nvalid));
      }
      const int64_t c_off = c_row + c_col[j];
      if (p.vec_store && n + 3 < d.N) {
        if (d.residual) {
          const float4 r = *(const float4*)(d.residual + c_off);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (d.c_dtype == RF_F32) {
          *(float4*)((float*)d.C + c_off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
          o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
          *(uint2*)((bf16_t*)d.C + c_off) = o;
        }
      } else {
        store_scalar4(d, c_row, n, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// exact fp32 kernel: 64x64 tile, BK=16, 256 threads x (4x4) outputs, fmaf accumulation
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float a_elem_f32(const rf_gemm_desc& d, const float* Ab, int m, int k) {
  if (m >= d.M || k >= d.K) return 0.f;
  if (d.a_mode == RF_AMODE_CONV3X3) {
    const int tap = k / d.conv_c, c = k % d.conv_c;
    const int hw = d.conv_h * d.conv_w;
    const int pix = m % hw;
    const int ii = pix / d.conv_w + (tap / 3 - 1) * d.conv_dil, jj = pix % d.conv_w + (tap % 3 - 1) * d.conv_dil;
    if (ii < 0 || ii >= d.conv_h || jj < 0 || jj >= d.conv_w) return 0.f;
    return Ab[((int64_t)(m / hw) * hw + (int64_t)ii * d.conv_w + jj) * d.conv_c + c];
  }
  return Ab[split_off(m, d.a_rc, d.a_ro, d.a_ri) + (int64_t)(k / d.kc) * d.a_ko + (k % d.kc)];
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmP p) {
  // Exact fp32: v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (no reduced-precision inputs), at the
  // fp32 matrix rate.  64x64 tile, BK=16, 4 waves x (32x32 = 2x2 MFMA tiles); generic (strided / conv) operand loads.
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  const rf_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int lid = blockIdx.x;
  const int tn = lid % p.tilesN;
  const int t2 = lid / p.tilesN;
  const int tm = t2 % p.tilesM;
  const int z = t2 / p.tilesM;
  int z0, z1, z2;
  batch_decode(d, z, z0, z1, z2);
  const int m0 = tm * BM, n0 = tn * BN;
  const float* Ab = (const float*)d.A + (d.a_mode == RF_AMODE_CONV3X3 ? 0 : z0 * d.a_bs[0] + z1 * d.a_bs[1] + z2 * d.a_bs[2]);
  const float* Bb = (const float*)d.B + z0 * d.b_bs[0] + z1 * d.b_bs[1] + z2 * d.b_bs[2];
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < d.K; k0 += BK) {
    for (int e = tid; e < BM * BK; e += 256) {
      const int r = e / BK, kk = e % BK;
      As[kk][r] = a_elem_f32(d, Ab, m0 + r, k0 + kk);
      const int n = n0 + r, k = k0 + kk;
      Bs[kk][r] = (n < d.N && k < d.K)
                      ? Bb[split_off(n, d.b_rc, d.b_ro, d.b_ri) + (int64_t)(k / d.kc) * d.b_ko + (k % d.kc)]
                      : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < BK / 4; ++g) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = As[4 * g + fq][wr * 32 + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = Bs[4 * g + fq][wc * 32 + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          // weights as MFMA-A, activations as MFMA-B: D[n_local = 4q+r][m_local = lane&15]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const int64_t c_z = z0 * d.c_bs[0] + z1 * d.c_bs[1] + z2 * d.c_bs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wr * 32 + i * 16 + fr;
    if (m >= d.M) continue;
    const int64_t c_row = c_z + split_off(m, d.c_rc, d.c_ro, d.c_ri);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wc * 32 + j * 16 + 4 * fq + r;
        if (n >= d.N) continue;
        float x = acc[i][j][r] * d.alpha;
        if (d.bias_mode == RF_BIAS_COL) x += d.bias[n];
        if (d.bias_mode == RF_BIAS_ROW) x += d.bias[m];
        x = apply_act(x, d.act, d.act_eps, (d.act_nvalid < 0 ? m < -d.act_nvalid : n < d.act_nvalid));
        const int64_t o = c_row + split_off(n, d.c_cc, d.c_co, 1);
        if (d.residual) x += d.residual[o];
        st(d.C, d.c_dtype, o, x);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
struct TileCfg {
  int bm, bn, bk, ns = 2;
};
static const TileCfg kTiles[] = {
    {0, 0, 0},       // 0 = auto
    {128, 128, 64},  // 1   (4 waves, 2x2)
    {128, 128, 32},  // 2
    {128, 96, 64},   // 3
    {128, 96, 32},   // 4
    {128, 64, 64},   // 5
    {128, 64, 32},   // 6
    {64, 128, 64},   // 7
    {64, 128, 32},   // 8
    {64, 96, 64},    // 9
    {64, 96, 32},    // 10
    {64, 64, 64},    // 11
    {64, 64, 32},    // 12
    {256, 256, 64},  // 13  (8 waves, 4x2: wave tile 64x128)
    {256, 288, 64},  // 14  (8 waves, 4x2: wave tile 64x144)
    {256, 192, 64},  // 15  (8 waves, 4x2: wave tile 64x96)
    {256, 128, 64},  // 16  (8 waves, 4x2: wave tile 64x64)
    {256, 256, 32},  // 17
    {256, 288, 32},  // 18
    {256, 256, 32, 4},  // 19  4-stage DMA ring (loads 3 K steps ahead)
    {256, 288, 32, 4},  // 20
    {256, 192, 32, 4},  // 21
    {256, 128, 32, 4},  // 22
    {128, 128, 32, 4},  // 23
    {128, 384, 64},     // 24  (8 waves, 4x2: wave tile 32x192) full 384-wide rows for the fused LayerNorm epilogue
};
static const int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int BM, int BN, int BK, int WGM, int WGN, int NS>
static int launch_bf16(const GemmP& p, int64_t nblk, hipStream_t s) {
  const size_t lds = lds_bytes_for(BM, BN, BK, WGM, NS);
  if (p.d.a_mode == RF_AMODE_CONV3X3) {
    auto k = gemm_bf16_kernel<BM, BN, BK, WGM, WGN, NS, RF_AMODE_CONV3X3>;
    static bool once = ((void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(64 * WGM * WGN), lds, s, p);
  } else {
    auto k = gemm_bf16_kernel<BM, BN, BK, WGM, WGN, NS, RF_AMODE_PLAIN>;
    static bool once = ((void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(64 * WGM * WGN), lds, s, p);
  }
  return rf_launch_status();
}

static int pick_bn(int N) {
  int best = 128, best_pad = ((N + 127) / 128) * 128;
  const int cands[2] = {96, 64};
  for (int c : cands) {
    const int pad = ((N + c - 1) / c) * c;
    if (pad < best_pad) {
      best = c;
      best_pad = pad;
    }
  }
  return best;
}

extern "C" int rf_gemm(const rf_gemm_desc* dd, void* stream) {
  if (!dd || !dd->A || !dd->B || !dd->C) return RF_EINVAL;
  GemmP p;
  p.d = *dd;
  p.dbg = (dd->tile_cfg >> 8) & 3;
  p.d.tile_cfg &= 0xff;
  rf_gemm_desc& d = p.d;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return RF_EINVAL;
  if (d.nb0 <= 0) d.nb0 = 1;
  if (d.nb1 <= 0) d.nb1 = 1;
  if (d.nb2 <= 0) d.nb2 = 1;
  if (d.kc <= 0 || d.a_mode == RF_AMODE_CONV3X3) d.kc = d.a_mode == RF_AMODE_CONV3X3 ? d.conv_c : d.K;
  if (d.a_mode == RF_AMODE_CONV3X3) {
    if (d.conv_c <= 0 || d.K != 9 * d.conv_c || d.M != d.conv_n * d.conv_h * d.conv_w) return RF_EINVAL;
    if (d.conv_h > 32767 || d.conv_w > 32767 || d.conv_dil < 1) return RF_EINVAL;
    d.b_ko = d.conv_c;  // weights are [N][tap][c] with K contiguous
  }
  if (d.bias_mode != RF_BIAS_NONE && !d.bias) return RF_EINVAL;
  const int64_t batch = (int64_t)d.nb0 * d.nb1 * d.nb2;
  // 4-wide stores legal?
  p.vec_store = 1;
  if (d.N % 4 != 0) p.vec_store = 0;
  if (d.c_cc > 0 && (d.c_cc % 4 != 0 || d.c_co % 4 != 0)) p.vec_store = 0;
  if (d.c_ri % 4 != 0 || (d.c_rc > 0 && d.c_ro % 4 != 0)) p.vec_store = 0;
  for (int i = 0; i < 3; ++i)
    if (d.c_bs[i] % 4 != 0) p.vec_store = 0;
  const size_t esz = d.c_dtype == RF_F32 ? 4 : 2;
  if (((uintptr_t)d.C % (4 * esz)) != 0) p.vec_store = 0;
  if (d.residual && ((uintptr_t)d.residual % 16) != 0) p.vec_store = 0;
  if (d.bias && ((uintptr_t)d.bias % 16) != 0) p.vec_store = 0;
  {
    // staged (LDS -> coalesced 16-byte rows) epilogue legal?
    const int epc = d.c_dtype == RF_F32 ? 4 : 8;
    auto ok = [&](int64_t v) { return (v % epc) == 0; };
    p.stage_epi = ok(d.N) && ok(d.c_ri) && (d.c_rc <= 0 || ok(d.c_ro)) && (d.c_cc <= 0 || (ok(d.c_cc) && ok(d.c_co))) &&
                  ok(d.c_bs[0]) && ok(d.c_bs[1]) && ok(d.c_bs[2]) && ((uintptr_t)d.C % 16) == 0 &&
                  (!d.residual || (d.c_dtype == RF_F32 && ((uintptr_t)d.residual % 16) == 0));
    if (getenv("RF_NO_STAGED_EPILOGUE")) p.stage_epi = 0;
  }
  {
    auto p2 = [](int v) { return v <= 0 || (v & (v - 1)) == 0; };
    p.c_pow2 = p2(d.c_rc) && p2(d.c_cc);
    p.c_rsh = d.c_rc > 0 ? __builtin_ctz(d.c_rc) : 0;
    p.c_csh = d.c_cc > 0 ? __builtin_ctz(d.c_cc) : 0;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool want_ln = d.ln_out != nullptr;

  if (d.ab_dtype == RF_F32) {
    if (want_ln) return RF_EINVAL;  // the fused LayerNorm epilogue exists on the bf16 MFMA path only
    p.tilesM = (d.M + 63) / 64;
    p.tilesN = (d.N + 63) / 64;
    const int64_t nblk = (int64_t)p.tilesM * p.tilesN * batch;
    if (nblk > 0x7fffffffLL) return RF_EINVAL;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)nblk), dim3(256), 0, s, p);
    return rf_launch_status();
  }
  if (d.ab_dtype != RF_BF16) return RF_EINVAL;
  // DMA needs 16-byte aligned sources: all element strides multiples of 8, K chunks multiples of 8
  auto al8 = [](int64_t v) { return (v % 8) == 0; };
  if (((uintptr_t)d.A % 16) || ((uintptr_t)d.B % 16)) return RF_EALIGN;
  if (d.kc % 8 != 0 && d.kc != d.K) return RF_EALIGN;
  if (d.K % 8 != 0) return RF_EALIGN;
  if (d.a_mode == RF_AMODE_CONV3X3) {
    if (d.conv_c % 8 != 0) return RF_EALIGN;
  } else if (!al8(d.a_ri) || !al8(d.a_ro) || !al8(d.a_ko) || !al8(d.a_bs[0]) || !al8(d.a_bs[1]) || !al8(d.a_bs[2])) {
    return RF_EALIGN;
  }
  if (!al8(d.b_ri) || !al8(d.b_ro) || !al8(d.b_ko) || !al8(d.b_bs[0]) || !al8(d.b_bs[1]) || !al8(d.b_bs[2]))
    return RF_EALIGN;

  TileCfg t;
  if (d.tile_cfg > 0 && d.tile_cfg < kNumTiles) {
    t = kTiles[d.tile_cfg];
  } else {
    t.ns = 2;
    t.bk = d.K >= 64 ? 64 : 32;  // BK=64 also for K % 64 != 0: the K tail is DMA'd from the zero word
    const int64_t rows = (int64_t)d.M * batch;
    if (d.M >= 1024 && rows >= 16384 && t.bk == 64 && (d.N % 288 == 0 || d.N % 256 == 0 || d.N % 192 == 0 || d.N == 128)) {
      // long activation panels: 8-wave 256-row tiles halve the DMA bytes per FLOP
      t.bm = 256;
      // measured on MI355X (tools/gemm_bench.py): 256-wide tiles win whenever they divide N (and for N = 1152),
      // 288-wide for the d_pair-wide outputs, 192-wide for the d_msa-wide ones
      t.bn = (d.N % 256 == 0 || d.N == 1152) ? 256 : (d.N % 288 == 0 ? 288 : (d.N % 192 == 0 ? 192 : 128));
    } else {
      t.bn = pick_bn(d.N);
      t.bm = d.M > 64 && ((d.M + 127) / 128) * 128 <= ((d.M + 63) / 64) * 64 + 32 ? 128 : 64;
    }
  }
  if (want_ln) {
    // needs complete rows per workgroup, plain fp32 C, <= 128 float4 chunks per row
    if (!p.stage_epi || d.c_dtype != RF_F32 || d.c_rc > 0 || d.c_cc > 0 || batch != 1 || d.N > 512 || d.N % 4 ||
        !d.ln_gamma || !d.ln_beta || ((uintptr_t)d.ln_out % 8) || ((uintptr_t)d.ln_gamma % 16) || ((uintptr_t)d.ln_beta % 16))
      return RF_EINVAL;
    if (d.N <= 288) { t.bm = 256; t.bn = 288; t.bk = 64; t.ns = 2; }
    else if (d.N <= 384) { t.bm = 128; t.bn = 384; t.bk = 64; t.ns = 2; }
    else return RF_EINVAL;
  }
  p.tilesM = (d.M + t.bm - 1) / t.bm;
  p.tilesN = (d.N + t.bn - 1) / t.bn;
  const int64_t nblk = (int64_t)p.tilesM * p.tilesN * batch;
  if (nblk > 0x7fffffffLL) return RF_EINVAL;
  p.nt_store = ((int64_t)d.M * d.N * batch * (d.c_dtype == RF_F32 ? 4 : 2) > (64ll << 20)) && !getenv("RF_NO_NT_STORE");
#define RF_CASE(BM_, BN_, BK_, WGM_, WGN_) \
  if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_ && t.ns == 2) return launch_bf16<BM_, BN_, BK_, WGM_, WGN_, 2>(p, nblk, s);
#define RF_CASE4(BM_, BN_, BK_, WGM_, WGN_) \
  if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_ && t.ns == 4) return launch_bf16<BM_, BN_, BK_, WGM_, WGN_, 4>(p, nblk, s);
  RF_CASE(128, 128, 64, 2, 2)
  RF_CASE(128, 128, 32, 2, 2)
  RF_CASE(128, 96, 64, 2, 2)
  RF_CASE(128, 96, 32, 2, 2)
  RF_CASE(128, 64, 64, 2, 2)
  RF_CASE(128, 64, 32, 2, 2)
  RF_CASE(64, 128, 64, 2, 2)
  RF_CASE(64, 128, 32, 2, 2)
  RF_CASE(64, 96, 64, 2, 2)
  RF_CASE(64, 96, 32, 2, 2)
  RF_CASE(64, 64, 64, 2, 2)
  RF_CASE(64, 64, 32, 2, 2)
  RF_CASE(256, 256, 64, 4, 2)
  RF_CASE(256, 288, 64, 4, 2)
  RF_CASE(256, 192, 64, 4, 2)
  RF_CASE(256, 128, 64, 4, 2)
  RF_CASE(256, 256, 32, 4, 2)
  RF_CASE(256, 288, 32, 4, 2)
  RF_CASE(128, 384, 64, 4, 2)
  RF_CASE4(256, 256, 32, 4, 2)
  RF_CASE4(256, 288, 32, 4, 2)
  RF_CASE4(256, 192, 32, 4, 2)
  RF_CASE4(256, 128, 32, 4, 2)
  RF_CASE4(128, 128, 32, 2, 2)
#undef RF_CASE4
#undef RF_CASE
  return RF_EINVAL;
}
