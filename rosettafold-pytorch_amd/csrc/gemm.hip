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
  int f32_vec;    // fp32 kernel: bit 0 / 1 = A / B tiles can be staged with float4 loads
  unsigned long long* stamps;  // timing experiments: per-workgroup phase stamps (rf_debug_gemm_stamps), else null
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
__host__ __device__ constexpr int lds_bytes_for(int bm, int bn, int bk, int wgm) {
  const int pipe = 2 * (bm + bn) * bk * 2, epi = wgm * 16 * (bn * 4 + 64);  // (wave strips: 16 bytes of padding per wave column, at most 4)
  return pipe > epi ? pipe : epi;
}

template <int BK>
__device__ __forceinline__ int swz(int row) {
  if constexpr (BK == 64)
    return (row >> 1) & 7;  // 8 x 16B chunks per 128-B row, two rows per 256-B bank row: the 16 rows of a fragment read hit 16 distinct slots
  else
    return (0x78 >> (((row >> 2) & 3) * 2)) & 3;  // 4 chunks per 64-B row: g = {0,2,3,1}[(row>>2)&3]
}

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int BK, int WGM, int WGN, int AMODE>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm_bf16_kernel(const GemmP p) {
  constexpr int NW = WGM * WGN;  // waves per workgroup, arranged WGM x WGN over the tile
  constexpr int SPR = BK / 8;    // 16-byte slots per tile row
  constexpr int A_INSTR = BM * SPR / 64, B_INSTR = BN * SPR / 64;  // wave-level DMA instructions per tile
  constexpr int A_PW = (A_INSTR + NW - 1) / NW, B_PW = (B_INSTR + NW - 1) / NW;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
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
  unsigned long long st0 = 0, st1 = 0, st2 = 0, st_w = 0, st_s = 0, st_x = 0;
  if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();

  const h16_t* Ab = (const h16_t*)d.A + (AMODE == RF_AMODE_CONV3X3 ? 0 : z0 * d.a_bs[0] + z1 * d.a_bs[1] + z2 * d.a_bs[2]);
  const h16_t* Bb = (const h16_t*)d.B + z0 * d.b_bs[0] + z1 * d.b_bs[1] + z2 * d.b_bs[2];

  // ---- per-lane staging state -------------------------------------------------------------
  const h16_t* a_src[A_PW];
  const h16_t* b_src[B_PW];
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
  // (BK = 64: an instruction covers 8 rows, the swizzle period is 16 rows -> the instruction's parity, = the wave's, enters)
  const int c_log = (lane % SPR) ^ swz<BK>(row_in_instr + (BK == 64 ? 8 * (wave & 1) : 0));
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
  const h16_t* const zsrc = (const h16_t*)g_rf_zero16;

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
        const h16_t* src = ok ? a_src[t] + a_koff : zsrc;
        glds16(src, a_lds + instr * 1024);
      }
    }
#pragma unroll
    for (int t = 0; t < B_PW; ++t) {
      const int instr = t * NW + wave;
      if ((B_INSTR % NW == 0) || instr < B_INSTR) {
        const h16_t* src = kvalid ? b_src[t] + b_koff : zsrc;
        glds16(src, b_lds + instr * 1024);
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

  // accumulators start at bias / alpha: the epilogue is then just max(acc * alpha, lo) (or the activation) and no
  // bias registers stay live across the K loop
  f32x4 acc[WM][WN];
  {
    const float inv_alpha = d.alpha == 1.0f ? 1.0f : 1.0f / d.alpha;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.bias_mode == RF_BIAS_COL) {
        const int n = n0 + wn * TN + j * 16 + 4 * (lane >> 4);
        if (n + 3 < d.N) {
          if (p.vec_store)
            bc = *(const float4*)(d.bias + n);  // N % 4 == 0 and the bias vector is 16-byte aligned (checked on the host)
          else
            bc = make_float4(d.bias[n], d.bias[n + 1], d.bias[n + 2], d.bias[n + 3]);
        } else {
          if (n < d.N) bc.x = d.bias[n];
          if (n + 1 < d.N) bc.y = d.bias[n + 1];
          if (n + 2 < d.N) bc.z = d.bias[n + 2];
        }
      }
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        float bm = 0.f;
        if (d.bias_mode == RF_BIAS_ROW) {
          const int m = m0 + wm * TM + i * 16 + (lane & 15);
          bm = m < d.M ? d.bias[m] : 0.f;
        }
        acc[i][j] = (f32x4){(bc.x + bm) * inv_alpha, (bc.y + bm) * inv_alpha, (bc.z + bm) * inv_alpha, (bc.w + bm) * inv_alpha};
      }
    }
  }

  const int fr = lane & 15, fq = lane >> 4;
  const int nk = (d.K + BK - 1) / BK;
  // double-buffered DMA: the loads of K step kt+1 are in flight under the MFMAs of step kt
  stage(0);
  const int nk_run = (RF_DBG(p.dbg) & 2) ? 1 : nk;
  for (int kt = 0; kt < nk_run; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.stamps && kt == 0) st1 = __builtin_amdgcn_s_memrealtime();
    if (kt + 1 < nk) stage((kt + 1) & 1);
    const char* a_lds = smem + (kt & 1) * STAGE_BYTES;
    const char* b_lds = a_lds + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      h16x8 af[WM], bfr[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = wm * TM + i * 16 + fr;
        af[i] = *(const h16x8*)(a_lds + (row * SPR + ((kk * 4 + fq) ^ swz<BK>(row))) * 16);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int row = wn * TN + j * 16 + fr;
        bfr[j] = *(const h16x8*)(b_lds + (row * SPR + ((kk * 4 + fq) ^ swz<BK>(row))) * 16);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          // weight tile as MFMA-A, activation tile as MFMA-B: D[n_local][m_local]
          acc[i][j] = rf_mfma16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  if (p.stamps) st2 = __builtin_amdgcn_s_memrealtime();
  // ---- epilogue: lane holds C[m][n..n+3], m = ..+fr, n = ..+4*fq; the bias is already inside the accumulators ----
  if ((RF_DBG(p.dbg) & 1) && acc[0][0][0] != 12345.678f) return;
  const int64_t c_z = z0 * d.c_bs[0] + z1 * d.c_bs[1] + z2 * d.c_bs[2];
  const bool simple = d.act == RF_ACT_NONE || d.act == RF_ACT_RELU || d.act == RF_ACT_BLOCK_LN32;  // branch-free form max(acc * alpha, lo)
  const float lo = d.act == RF_ACT_RELU ? 0.f : -INFINITY;
  const float alpha = d.alpha;
  if constexpr (BM == 256 && BN == 256 && WGM == 4 && WGN == 2 && AMODE == RF_AMODE_PLAIN) {
    if (d.act == RF_ACT_BLOCK_LN32) {
      // LayerNorm over each aligned 32x32 output block (the 1024 outer-product features of one residue pair): a wave's
      // 64 x 128 block holds 2 x 4 of them, each spread over all 64 lanes (2 x 2 MFMA tiles x 4 registers): statistics
      // are two wave reductions per block, two-pass like the stand-alone kernel.  Feature k = 32 * (row % 32) + col % 32.
#pragma unroll
      for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
          float sm = 0.f;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const f32x4 v = acc[2 * pi + a][2 * pj + b];
              sm += (v[0] + v[1]) + (v[2] + v[3]);
            }
          const float mean = wave_sum(sm) * (1.0f / 1024.0f);
          float q = 0.f;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float dl = acc[2 * pi + a][2 * pj + b][e] - mean;
                q += dl * dl;
              }
          const float rstd = rsqrtf(wave_sum(q) * (1.0f / 1024.0f) + d.ln_eps);
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int k = (a * 16 + fr) * 32 + b * 16 + 4 * fq;
              const float4 g4 = *(const float4*)(d.ln_gamma + k), b4 = *(const float4*)(d.ln_beta + k);
              f32x4& v = acc[2 * pi + a][2 * pj + b];
              v[0] = (v[0] - mean) * rstd * g4.x + b4.x;
              v[1] = (v[1] - mean) * rstd * g4.y + b4.y;
              v[2] = (v[2] - mean) * rstd * g4.z + b4.z;
              v[3] = (v[3] - mean) * rstd * g4.w + b4.w;
            }
        }
    }
  }
  {
    // Wave-private staged epilogue.  The 8-byte-per-lane stores of the MFMA layout are store-issue bound, so each
    // wave converts its own TM x TN accumulator block 16 (or 32) rows at a time into a private LDS strip and reads it
    // back as 16-byte chunks of consecutive columns: one store instruction then covers whole 128-byte lines of 2-4
    // rows.  LDS operations of one wave execute in order, so after the single workgroup barrier that retires the
    // operand tiles there is no further synchronisation: the waves stream {ds_write, ds_read, global_store}
    // independently and the two waves of a SIMD interleave.
    auto wave_epi = [&](auto esz_tag, auto scalar_tag) {
      constexpr int ESZ = decltype(esz_tag)::value;
      constexpr bool SCALAR = decltype(scalar_tag)::value;  // C layout rules out 16-byte chunks: element-wise read-out
      constexpr int PITCHW = TN * ESZ + 16;
      constexpr int LDS_CAP = lds_bytes_for(BM, BN, BK, WGM);
      constexpr int IPW = (WM >= 2 && NW * 32 * PITCHW <= LDS_CAP) ? 2 : 1;
      static_assert(NW * 16 * IPW * PITCHW <= LDS_CAP, "wave strips do not fit the operand buffers");
      constexpr int CPRW = TN * ESZ / 16;  // 16-byte chunks per strip row
      constexpr int EPC = 16 / ESZ;        // elements per chunk
      constexpr int RW = 16 * IPW;         // strip rows per pass
      constexpr int NCH = RW * CPRW;
      constexpr int NIT = (NCH + 63) / 64;
      constexpr int G = WM * WN * 4 > 128 ? 2 : 4;  // chunks in flight per lane (fewer when the accumulators fill the file)
      char* const strip = smem + wave * (RW * PITCHW);
      const bool plain_c = d.c_rc <= 0 && d.c_cc <= 0;
      __syncthreads();  // every wave is done with the operand tiles
#pragma unroll
      for (int i0 = 0; i0 < WM; i0 += IPW) {
        // opaque copy of the pass index: keeps hipcc from hoisting the address arithmetic of every unrolled pass above
        // the first one (that pushed the accumulators of the 288-wide tile into scratch)
        int i0v = i0;
        asm volatile("" : "+s"(i0v));
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii) {
          const int i = i0 + ii;
          if (i < WM) {
            char* lrow = strip + (ii * 16 + fr) * PITCHW;
            if (simple) {
#pragma unroll
              for (int j = 0; j < WN; ++j) {
                const int nl = j * 16 + 4 * fq;
                const float v0 = fmaxf(acc[i][j][0] * alpha, lo), v1 = fmaxf(acc[i][j][1] * alpha, lo);
                const float v2 = fmaxf(acc[i][j][2] * alpha, lo), v3 = fmaxf(acc[i][j][3] * alpha, lo);
                if constexpr (ESZ == 4) {
                  *(float4*)(lrow + nl * 4) = make_float4(v0, v1, v2, v3);
                } else {
                  uint2 o;
                  o.x = rf_pack2_h16(v0, v1);
                  o.y = rf_pack2_h16(v2, v3);
                  *(uint2*)(lrow + nl * 2) = o;
                }
              }
            } else {
              const int m = m0 + wm * TM + i * 16 + fr;
#pragma unroll
              for (int j = 0; j < WN; ++j) {
                const int nl = j * 16 + 4 * fq;
                const int n = n0 + wn * TN + nl;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  v[e] = apply_act(acc[i][j][e] * alpha, d.act, d.act_eps,
                                   (d.act_nvalid < 0 ? m < -d.act_nvalid : n + e < d.act_nvalid));
                if constexpr (ESZ == 4) {
                  *(float4*)(lrow + nl * 4) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                  uint2 o;
                  o.x = rf_pack2_h16(v[0], v[1]);
                  o.y = rf_pack2_h16(v[2], v[3]);
                  *(uint2*)(lrow + nl * 2) = o;
                }
              }
            }
          }
        }
        asm volatile("" ::: "memory");  // compiler fence: keep the strip reads below out of the write block above
        if constexpr (SCALAR) {
          for (int e = lane; e < RW * TN; e += 64) {
            const int r = e / TN, c = e % TN;
            const int i = i0v + (r >> 4);
            const int m = m0 + wm * TM + i * 16 + (r & 15);
            const int n = n0 + wn * TN + c;
            if ((WM % IPW == 0 || i < WM) && m < d.M && n < d.N) {
              float x = *(const float*)(strip + r * PITCHW + c * 4);
              const int64_t o = c_z + split_off(m, d.c_rc, d.c_ro, d.c_ri) + split_off(n, d.c_cc, d.c_co, 1);
              if (d.residual) x += d.residual[o];
              st(d.C, d.c_dtype, o, x);
            }
          }
          continue;
        }
        // read the strip back G chunks per lane at a time: strip row r (-> tile row), chunk c (-> columns)
#pragma unroll
        for (int t0 = 0; t0 < NIT; t0 += G) {
          int64_t coff[G];
          bool okc[G];
          float4 res[G];
          f32x4 vv[G];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int idx = lane + 64 * (t0 + g);
            const int r = idx / CPRW, c = idx % CPRW;
            const int i = i0v + (r >> 4);
            const int m = m0 + wm * TM + i * 16 + (r & 15);
            const int n = n0 + wn * TN + c * EPC;
            okc[g] = t0 + g < NIT && (NCH % 64 == 0 || idx < NCH) && (WM % IPW == 0 || i < WM) && m < d.M && n < d.N;
            if (plain_c) {
              coff[g] = c_z + (int64_t)m * d.c_ri + n;
            } else if (p.c_pow2) {  // both split factors are powers of two: shifts instead of divisions
              const int64_t ro = d.c_rc > 0 ? (int64_t)(m >> p.c_rsh) * d.c_ro + (int64_t)(m & (d.c_rc - 1)) * d.c_ri
                                            : (int64_t)m * d.c_ri;
              const int64_t co = d.c_cc > 0 ? (int64_t)(n >> p.c_csh) * d.c_co + (n & (d.c_cc - 1)) : n;
              coff[g] = c_z + ro + co;
            } else {
              coff[g] = c_z + split_off(m, d.c_rc, d.c_ro, d.c_ri) + split_off(n, d.c_cc, d.c_co, 1);
            }
            if constexpr (ESZ == 4) {
              res[g] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (d.residual && okc[g]) res[g] = *(const float4*)(d.residual + coff[g]);
            }
          }
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int idx = lane + 64 * (t0 + g);
            const int r = idx / CPRW, c = idx % CPRW;
            vv[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t0 + g < NIT && (NCH % 64 == 0 || idx < NCH)) vv[g] = *(const f32x4*)(strip + r * PITCHW + c * 16);
          }
#pragma unroll
          for (int g = 0; g < G; ++g) {
            if (!okc[g]) continue;
            if constexpr (ESZ == 4) {
              f32x4 v = vv[g];
              v[0] += res[g].x; v[1] += res[g].y; v[2] += res[g].z; v[3] += res[g].w;
              if (p.nt_store)
                __builtin_nontemporal_store(v, (f32x4*)((float*)d.C + coff[g]));
              else
                *(f32x4*)((float*)d.C + coff[g]) = v;
            } else {
              if (p.nt_store)
                __builtin_nontemporal_store(vv[g], (f32x4*)((h16_t*)d.C + coff[g]));
              else
                *(f32x4*)((h16_t*)d.C + coff[g]) = vv[g];
            }
          }
        }
      }
    };
    if (!p.stage_epi)
      wave_epi(std::integral_constant<int, 4>{}, std::true_type{});
    else if (d.c_dtype == RF_F32)
      wave_epi(std::integral_constant<int, 4>{}, std::false_type{});
    else
      wave_epi(std::integral_constant<int, 2>{}, std::false_type{});
    if (p.stamps && tid == 0) {
      unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
      o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = __builtin_amdgcn_s_getreg(63492);            // HW_ID
      o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      o[6] = __builtin_amdgcn_s_memrealtime();            // this wave's stores acknowledged
      o[7] = 0;
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
  // plain row-major operands with K % 4 == 0 (the SE(3) / structure-track projections): one float4 per thread and tile
  const bool a_vec = p.f32_vec & 1, b_vec = p.f32_vec & 2;
  const int vr = tid >> 2, vk = (tid & 3) * 4;
  for (int k0 = 0; k0 < d.K; k0 += BK) {
    if (a_vec) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + vr < d.M && k0 + vk < d.K) v = *(const float4*)(Ab + (int64_t)(m0 + vr) * d.a_ri + k0 + vk);
      As[vk][vr] = v.x; As[vk + 1][vr] = v.y; As[vk + 2][vr] = v.z; As[vk + 3][vr] = v.w;
    }
    if (b_vec) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + vr < d.N && k0 + vk < d.K) v = *(const float4*)(Bb + (int64_t)(n0 + vr) * d.b_ri + k0 + vk);
      Bs[vk][vr] = v.x; Bs[vk + 1][vr] = v.y; Bs[vk + 2][vr] = v.z; Bs[vk + 3][vr] = v.w;
    }
    if (!a_vec || !b_vec) {
      for (int e = tid; e < BM * BK; e += 256) {
        const int r = e / BK, kk = e % BK;
        if (!a_vec) As[kk][r] = a_elem_f32(d, Ab, m0 + r, k0 + kk);
        const int n = n0 + r, k = k0 + kk;
        if (!b_vec)
          Bs[kk][r] = (n < d.N && k < d.K)
                          ? Bb[split_off(n, d.b_rc, d.b_ro, d.b_ri) + (int64_t)(k / d.kc) * d.b_ko + (k % d.kc)]
                          : 0.f;
      }
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
      const int nb = n0 + wc * 32 + j * 16 + 4 * fq;
      if (p.vec_store && d.c_cc <= 0 && nb + 3 < d.N) {  // 16-byte (fp32) / 8-byte (bf16) store of the lane's 4 columns
        const int64_t o = c_row + nb;
        float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias_mode == RF_BIAS_COL) bc = *(const float4*)(d.bias + nb);
        const float bm = d.bias_mode == RF_BIAS_ROW ? d.bias[m] : 0.f;
        const float bv[4] = {bc.x + bm, bc.y + bm, bc.z + bm, bc.w + bm};
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = apply_act(acc[i][j][r] * d.alpha + bv[r], d.act, d.act_eps, (d.act_nvalid < 0 ? m < -d.act_nvalid : nb + r < d.act_nvalid));
        if (d.residual) {
          const float4 rr = *(const float4*)(d.residual + o);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (d.c_dtype == RF_F32) {
          *(float4*)((float*)d.C + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 w;
          w.x = rf_pack2_h16(v[0], v[1]);
          w.y = rf_pack2_h16(v[2], v[3]);
          *(uint2*)((h16_t*)d.C + o) = w;
        }
        continue;
      }
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
  int bm, bn, bk;
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
    {256, 128, 32},  // 17  (4 waves, 2x2: wave tile 128x64; 48 KB LDS -> two workgroups per CU)
    {128, 256, 32},  // 18  (4 waves, 2x2: wave tile 64x128; two workgroups per CU)
};
static const int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int BM, int BN, int BK, int WGM, int WGN>
static int launch_bf16(const GemmP& p, int64_t nblk, hipStream_t s) {
  const size_t lds = lds_bytes_for(BM, BN, BK, WGM);
  if (p.d.a_mode == RF_AMODE_CONV3X3) {
    auto k = gemm_bf16_kernel<BM, BN, BK, WGM, WGN, RF_AMODE_CONV3X3>;
    if (const int e = rf_enable_big_lds<gemm_bf16_kernel<BM, BN, BK, WGM, WGN, RF_AMODE_CONV3X3>>()) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(64 * WGM * WGN), lds, s, p);
  } else {
    auto k = gemm_bf16_kernel<BM, BN, BK, WGM, WGN, RF_AMODE_PLAIN>;
    if (const int e = rf_enable_big_lds<gemm_bf16_kernel<BM, BN, BK, WGM, WGN, RF_AMODE_PLAIN>>()) return e;
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

static unsigned long long* g_gemm_stamps = nullptr;
/* timing experiments: 8 x u64 per workgroup {entry, first tile landed, K loop done, stores issued, HW_ID, XCC_ID, stores
 * acknowledged, -} in 10 ns ticks; pass null to switch off.  The buffer must hold 8 * (number of workgroups) words. */
extern "C" int rf_debug_gemm_stamps(void* buf) {
  g_gemm_stamps = (unsigned long long*)buf;
  return 0;
}
/* same for the persistent kernel of gemm_fast.hip: {cycles total, in vmcnt waits, in barriers, in K-step bodies, in
 * epilogues, tiles, -, -} per workgroup (shader clock) */
extern "C" int rf_debug_gemm_fast_stamps(void* buf) {
  rf_gemm_fast_set_stamps(buf);
  return 0;
}

// Which kernel family the calling thread's last rf_gemm chose (bench.py attributes launch times with it instead of
// re-deriving the dispatch rules): 0 exact-fp32, 1 generic bf16 tile, 2 conv3x3 implicit GEMM, 3 persistent tile kernel
// (gemm_fast.hip), 4 register-resident-weights kernel (gemm_wreg.hip); -1 = nothing launched.
static thread_local int g_last_family = -1;
extern "C" int rf_gemm_last_family(void) { return g_last_family; }

extern "C" int rf_gemm(const rf_gemm_desc* dd, void* stream) {
  g_last_family = -1;
  if (!dd || !dd->A || !dd->B || !dd->C) return RF_EINVAL;
  GemmP p;
  p.d = *dd;
  p.dbg = (dd->tile_cfg >> 8) & 3;  // (timing experiments of the tuning build; the production build refuses them)
#ifndef RF_ABLATION
  if (p.dbg) return RF_EINVAL;
#endif
  p.d.tile_cfg &= 0xff;
  rf_gemm_desc& d = p.d;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return RF_EINVAL;
  if (d.rs && (d.ab_dtype != RF_H16 || d.tile_cfg != 0 || d.rs_rpb <= 0 || d.rs_cg <= 0 || d.rs_cg % 16 || d.rs_ncols <= 0 ||
               d.rs_ncols % d.rs_cg))
    return RF_EINVAL;
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
    static const bool no_staged = rf_env_flag("RF_NO_STAGED_EPILOGUE");
    if (no_staged) p.stage_epi = 0;
  }
  {
    auto p2 = [](int v) { return v <= 0 || (v & (v - 1)) == 0; };
    p.c_pow2 = p2(d.c_rc) && p2(d.c_cc);
    p.c_rsh = d.c_rc > 0 ? __builtin_ctz(d.c_rc) : 0;
    p.c_csh = d.c_cc > 0 ? __builtin_ctz(d.c_cc) : 0;
  }
  p.stamps = g_gemm_stamps;
  hipStream_t s = (hipStream_t)stream;
  const bool want_ln = d.ln_out != nullptr;

  if (d.ab_dtype == RF_F32) {
    if (d.act == RF_ACT_BLOCK_LN32) return RF_EINVAL;  // bf16 MFMA path only
    p.f32_vec = 0;
    if (d.K % 4 == 0 && d.kc == d.K) {
      if (d.a_mode == RF_AMODE_PLAIN && d.a_rc <= 0 && d.a_ri % 4 == 0 && ((uintptr_t)d.A % 16) == 0 && d.a_bs[0] % 4 == 0 &&
          d.a_bs[1] % 4 == 0 && d.a_bs[2] % 4 == 0)
        p.f32_vec |= 1;
      if (d.b_rc <= 0 && d.b_ri % 4 == 0 && ((uintptr_t)d.B % 16) == 0 && d.b_bs[0] % 4 == 0 && d.b_bs[1] % 4 == 0 && d.b_bs[2] % 4 == 0)
        p.f32_vec |= 2;
    }
    if (want_ln) return RF_EINVAL;  // the fused LayerNorm epilogue exists on the bf16 MFMA path only
    p.tilesM = (d.M + 63) / 64;
    p.tilesN = (d.N + 63) / 64;
    const int64_t nblk = (int64_t)p.tilesM * p.tilesN * batch;
    if (nblk > 0x7fffffffLL) return RF_EINVAL;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)nblk), dim3(256), 0, s, p);
    g_last_family = 0;
    return rf_launch_status();
  }
  if (d.ab_dtype != RF_H16) return RF_EINVAL;
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

  if (want_ln) {
    // "residual add + LayerNorm of the next sub-layer": fused into the persistent kernel's epilogue for the 288-wide pair
    // rows (one tile = whole rows); otherwise the normalisation runs as a second launch over the fp32 rows just written
    if (d.c_dtype != RF_F32 || d.c_rc > 0 || d.c_cc > 0 || batch != 1 || !d.ln_gamma || !d.ln_beta ||
        ((uintptr_t)d.ln_out % 8) || ((uintptr_t)d.ln_gamma % 16) || ((uintptr_t)d.ln_beta % 16))
      return RF_EINVAL;
  }
  if (d.tile_cfg == 0 && !p.dbg && !p.stamps) {  // (stamps of the generic kernel: keep it on the generic kernel)
    int rc = 0;
    if (d.a_mode == RF_AMODE_CONV3X3 && !want_ln && rf_conv288_try(d, batch, &rc, stream)) {
      g_last_family = 2;
      return rc;
    }
    if (rf_gemm_wreg_try(d, batch, &rc, stream)) {
      g_last_family = 4;
      return rc;
    }
    if (d.rs) return RF_EINVAL;  // the row-group scale lives in the register-resident-weights kernel's epilogue only
    if (rf_gemm_fast_try(d, batch, &rc, stream)) {
      g_last_family = 3;
      return rc;
    }
    if (want_ln) {
      rf_gemm_desc d2 = d;
      d2.ln_out = nullptr;
      if (rf_gemm_fast_try(d2, batch, &rc, stream)) {
        g_last_family = 3;
        if (rc != 0) return rc;
        return rf_layernorm(d.C, RF_F32, d.c_ri, d.ln_out, RF_H16, d.N, d.M, d.N, d.ln_gamma, d.ln_beta, d.ln_eps, 1, RF_ACT_NONE, stream);
      }
    }
  }
  TileCfg t;
  if (d.tile_cfg > 0 && d.tile_cfg < kNumTiles) {
    t = kTiles[d.tile_cfg];
  } else {
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
  if (d.act == RF_ACT_BLOCK_LN32) {
    if (d.a_mode != RF_AMODE_PLAIN || d.M % 256 || d.N % 256 || d.K < 64 || !d.ln_gamma || !d.ln_beta || d.ln_out ||
        d.bias_mode != RF_BIAS_NONE || d.alpha != 1.0f || ((uintptr_t)d.ln_gamma % 16) || ((uintptr_t)d.ln_beta % 16))
      return RF_EINVAL;
    t.bm = 256; t.bn = 256; t.bk = 64;
  }
  p.tilesM = (d.M + t.bm - 1) / t.bm;
  p.tilesN = (d.N + t.bn - 1) / t.bn;
  const int64_t nblk = (int64_t)p.tilesM * p.tilesN * batch;
  if (nblk > 0x7fffffffLL) return RF_EINVAL;
  static const bool no_nt_store = rf_env_flag("RF_NO_NT_STORE");
  p.nt_store = ((int64_t)d.M * d.N * batch * (d.c_dtype == RF_F32 ? 4 : 2) > (64ll << 20)) && !no_nt_store;
  int rc = RF_EINVAL;
#define RF_CASE(BM_, BN_, BK_, WGM_, WGN_) \
  if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_) rc = launch_bf16<BM_, BN_, BK_, WGM_, WGN_>(p, nblk, s);
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
  RF_CASE(256, 128, 32, 2, 2)
  RF_CASE(128, 256, 32, 2, 2)
#undef RF_CASE
  g_last_family = d.a_mode == RF_AMODE_CONV3X3 ? 2 : 1;
  if (rc != 0 || !want_ln) return rc;
  return rf_layernorm(d.C, RF_F32, d.c_ri, d.ln_out, RF_H16, d.N, d.M, d.N, d.ln_gamma, d.ln_beta, d.ln_eps, 1, RF_ACT_NONE, stream);
}
