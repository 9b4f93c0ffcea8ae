// Persistent plain-layout bf16 GEMM for the long activation panels of the forward path (gfx950 / MI355X).
//
// rf_gemm routes here when the operands are plain row-major (activations [M, K], nn.Linear weight [N, K],
// output [M, N]; rf.py:195-281), M is a multiple of 256 and N a multiple of the tile width: the q|k|v / feed-forward /
// output projections of the MSA and pair tracks, i.e. >90 % of the GEMM time of the bench configuration.
//
// What differs from the generic kernel in gemm.hip (same MFMA tile, same LDS image and swizzle):
//   * persistent workgroups: one 512-thread workgroup per CU walks tiles lid = round * grid + slot.  The K steps of
//     consecutive tiles form ONE double-buffered DMA pipeline: the last K step of a tile prefetches the first K step
//     of the next tile, so the launch gap, the address set-up and the first-tile DMA latency (2-4 us of a 16 us
//     tile, measured with tools/gemm_stamps.py) are paid once per workgroup instead of once per tile, and a tile's
//     stores drain under the next tile's K loop.
//   * all global addresses are "uniform base + 32-bit lane offset": no per-slot 64-bit pointers live in VGPRs.
//   * the bias is folded into the accumulator init; the epilogue is max(acc, lo) -> wave-private LDS strips ->
//     16-byte row-contiguous stores (see gemm.hip), specialised at compile time on the output type / residual.
#include <type_traits>

#include "common.h"

static __device__ __attribute__((aligned(16))) unsigned int g_fast_zero16[4];  // zero source for K-tail DMA lanes

struct FastP {
  const h16_t* A;
  const h16_t* B;
  void* C;
  const float* bias;      // fp32 [N] or null
  const float* residual;  // fp32 [M, N] (ldc) or null
  int M, N, K;
  int lda, ldb, ldc;  // elements
  int relu;
  int tilesN, ntiles;
  int nt_store;
  int no_lag;  // experiment switch: 1 = waves 4-7 run the same phase order as waves 0-3
  int dbg;     // timing experiments (RF_GEMM_DBG, results are WRONG when set): 1 skip MFMAs, 2 skip the DMA of every K step but
               // a tile's first, 4 skip the epilogue's global stores / residual loads, 8 skip the fragment reads
  // fused "LayerNorm of the next sub-layer" (LN variant: N == BN, fp32 C with residual): bf16 [M, N] normalised rows
  void* ln_out;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  unsigned long long* stamps;  // timing experiments (tools/gemm_stamps.py): 8 x u64 per workgroup, else null
  // split-C variant (CS): element (m, n) of the result goes to (m / c_rc) * c_ro + (m % c_rc) * ldc + (n / c_cc) * c_co + n % c_cc
  // (rf_gemm_desc's C addressing; c_rc / c_cc <= 0: no split).  The q|k|v projection of the tied MSA-row attention writes
  // its heads this way: [B, N, L, G*32] -> [B, N, G, L, 32] (csrc/tied.hip).  c_rsh / c_csh: log2 when a power of two, else -1.
  int c_rc, c_cc, c_rsh, c_csh;
  int64_t c_ro, c_co;
};

__device__ __forceinline__ void fast_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BN, bool OUT_F32, bool HAS_RES, bool STAMP = false, bool LN = false, bool CS = false, int BM = 256>
__global__ __launch_bounds__(512, 2) void gemm_fast_kernel(const FastP p) {
  static_assert(!LN || (OUT_F32 && HAS_RES), "the fused LayerNorm epilogue normalises the updated fp32 residual rows");
  static_assert(!CS || (!OUT_F32 && !HAS_RES && !LN), "split-C layout: bf16 output without residual");
  static_assert(BM == 256 || BM == 128, "tile rows");
  constexpr int BK = 64, NW = 8, TM = BM / 4, TN = BN / 2, WM = TM / 16, WN = TN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int B_INSTR = BN / 8;                  // wave-level DMA instructions per B tile (8 rows x 128 B each)
  constexpr int A_PW = BM / 64, B_PW = (B_INSTR + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  // DMA source offsets of this lane (bytes): row lane/8 of the instruction's 8 rows; the LDS image is lane-linear, so the
  // bank swizzle (chunk ^ ((row >> 1) & 7): the 16 rows of one fragment read then hit 16 distinct 16-byte slots of the
  // 256-byte bank row) is applied to the SOURCE chunk and again on the ds_read
  const int lrow = lane >> 3;
  const int c_log = (lane & 7) ^ ((4 * (wave & 1) + (lrow >> 1)) & 7);  // swizzle ((row >> 1) & 7) of tile row 8 * instr + lrow
  const unsigned a_lane = (unsigned)(lrow * p.lda + c_log * 8) * 2u;
  const unsigned b_lane = (unsigned)(lrow * p.ldb + c_log * 8) * 2u;
  const int nk = (p.K + BK - 1) / BK;
  const bool k_tail = (p.K % BK) != 0;

  auto stage = [&](int buf, int m0, int n0, int kt) {
    char* a_lds = smem + buf * STAGE_BYTES;
    char* b_lds = a_lds + A_BYTES;
    const char* Ab = (const char*)p.A + ((int64_t)(m0 + wave * 8) * p.lda + kt * BK) * 2;
    const char* Bb = (const char*)p.B + ((int64_t)(n0 + wave * 8) * p.ldb + kt * BK) * 2;
    if (k_tail && kt == nk - 1) {
      const bool kvalid = kt * BK + c_log * 8 < p.K;
#pragma unroll
      for (int t = 0; t < A_PW; ++t)
        fast_glds16(kvalid ? Ab + (int64_t)t * 64 * p.lda * 2 + a_lane : (const char*)g_fast_zero16, a_lds + (t * NW + wave) * 1024);
#pragma unroll
      for (int t = 0; t < B_PW; ++t)
        if ((B_INSTR % NW == 0) || t * NW + wave < B_INSTR)
          fast_glds16(kvalid ? Bb + (int64_t)t * 64 * p.ldb * 2 + b_lane : (const char*)g_fast_zero16, b_lds + (t * NW + wave) * 1024);
    } else {
#pragma unroll
      for (int t = 0; t < A_PW; ++t) fast_glds16(Ab + (int64_t)t * 64 * p.lda * 2 + a_lane, a_lds + (t * NW + wave) * 1024);
#pragma unroll
      for (int t = 0; t < B_PW; ++t)
        if ((B_INSTR % NW == 0) || t * NW + wave < B_INSTR)
          fast_glds16(Bb + (int64_t)t * 64 * p.ldb * 2 + b_lane, b_lds + (t * NW + wave) * 1024);
    }
  };

  // ---- epilogue geometry (tile independent) ----------------------------------------------------------------
  constexpr int ESZ = OUT_F32 ? 4 : 2;
  constexpr int PITCHW = TN * ESZ + 16;
  // strips of RP rows per wave overlay ONE stage buffer (the other one holds the next tile's first K step)
  constexpr int RP = (NW * 32 * PITCHW <= STAGE_BYTES) ? 32 : ((NW * 16 * PITCHW <= STAGE_BYTES) ? 16 : 8);
  static_assert(NW * RP * PITCHW <= STAGE_BYTES, "wave strips do not fit a stage buffer");
  constexpr int CPRW = TN * ESZ / 16;  // 16-byte chunks per strip row
  constexpr int EPC = 16 / ESZ;        // elements per chunk
  constexpr int NCH = RP * CPRW;
  constexpr int NIT = (NCH + 63) / 64;
  constexpr int G = NIT < 4 ? NIT : 4;  // chunks in flight per lane
  auto cs_row = [&](int m) -> int64_t {
    if (p.c_rc <= 0) return (int64_t)m * p.ldc;
    const int q = p.c_rsh >= 0 ? m >> p.c_rsh : m / p.c_rc;
    return (int64_t)q * p.c_ro + (int64_t)(m - q * p.c_rc) * p.ldc;
  };
  auto cs_col = [&](int n) -> int64_t {
    if (p.c_cc <= 0) return n;
    const int q = p.c_csh >= 0 ? n >> p.c_csh : n / p.c_cc;
    return (int64_t)q * p.c_co + (n - q * p.c_cc);
  };
  const float lo = p.relu ? 0.f : -INFINITY;
  constexpr int GB_OFF = 2 * STAGE_BYTES;  // LN variant: gamma | beta (BN floats each) behind the stage buffers
  if constexpr (LN) {
    for (int i = tid; i < 2 * BN; i += 512) ((float*)(smem + GB_OFF))[i] = i < BN ? p.ln_gamma[i] : p.ln_beta[i - BN];
  }
  // the bias row (N <= 384 floats) lives in LDS: read from global at every tile's accumulator init, its per-lane pointer was
  // spilled to scratch by hipcc and every reload came with an s_waitcnt vmcnt(0) -- nine serialised round trips per tile
  constexpr int BIAS_OFF = GB_OFF + (LN ? 2 * BN * 4 : 0);
  for (int i = tid; i < p.tilesN * BN; i += 512) ((float*)(smem + BIAS_OFF))[i] = (p.bias && i < p.N) ? p.bias[i] : 0.f;
  __syncthreads();  // (before any DMA is in flight)

  // ---- persistent tile walk ----------------------------------------------------------------------------------
  // slot: XCD-aware position of this workgroup inside a round of gridDim.x tiles (workgroups b and b+8 share an XCD
  // and get neighbouring tiles -> the N tiles of one activation row panel hit the same L2)
  const int G_ = gridDim.x;
  int slot;
  {
    const int bid = blockIdx.x, q = G_ >> 3, r = G_ & 7, x = bid & 7;
    slot = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  int lid = slot;
  if (lid >= p.ntiles) return;
  unsigned long long tk_vm = 0, tk_bar = 0, tk_mma = 0, tk_epi = 0, tk_0 = 0, tk_x = 0, ntile = 0;
  if constexpr (STAMP) tk_0 = clock64();
  int m0 = (lid / p.tilesN) * BM, n0 = (lid % p.tilesN) * BN;
  int buf = 0;
  stage(0, m0, n0, 0);

  auto run = [&](auto lag_tag) {
  constexpr bool LAG = decltype(lag_tag)::value;
  while (true) {
    const int lid_next = lid + G_;
    const bool has_next = lid_next < p.ntiles;
    const int m0n = has_next ? (lid_next / p.tilesN) * BM : 0, n0n = has_next ? (lid_next % p.tilesN) * BN : 0;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const f32x4 bc = *(const f32x4*)(smem + BIAS_OFF + (n0 + wn * TN + j * 16 + 4 * fq) * 4);
#pragma unroll
      for (int i = 0; i < WM; ++i) acc[i][j] = bc;
    }

    // K loop.  The two waves of a SIMD (w and w + 4) would otherwise run the same phases in lockstep: both issue their
    // fragment reads, then both queue for the matrix pipe, and the older one waits at the K-step barrier for its partner
    // (22 % of a tile, tools/gemm_stamps.py).  Waves 4-7 therefore LAG by one phase: they carry the second-half fragments
    // of a K step across the barrier and issue those MFMAs first thing in the next step, while waves 0-3 are reading:
    //   waves 0-3:  R0 M0 R1 M1 | R0 M0 R1 M1 |        waves 4-7:  M1' R0 M0 R1 | M1' R0 M0 R1 | ... M1'
    // (one barrier per K step as before; the carried fragments are in registers, and a lagging wave retires its R1 reads
    // before the barrier, so the buffer can be re-staged right behind it).
    h16x8 af[WM] = {}, bfr[WN] = {};
    auto read_frags = [&](int kk) {
      if (RF_DBG(p.dbg) & 8) return;
      const char* a_lds = smem + buf * STAGE_BYTES;
      const char* b_lds = a_lds + A_BYTES;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = wm * TM + i * 16 + fr;
        af[i] = *(const h16x8*)(a_lds + (row * 8 + ((kk * 4 + fq) ^ ((row >> 1) & 7))) * 16);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int row = wn * TN + j * 16 + fr;
        bfr[j] = *(const h16x8*)(b_lds + (row * 8 + ((kk * 4 + fq) ^ ((row >> 1) & 7))) * 16);
      }
    };
    auto mfmas = [&]() {
      if (RF_DBG(p.dbg) & 1) return;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          // weight tile as MFMA-A, activation tile as MFMA-B: lane holds C[m = ..+fr][n = ..+4*fq .. +3]
          acc[i][j] = rf_mfma16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    };
    const bool half_tail = k_tail && (p.K % BK) <= 32;  // K tail of <= 32: the second half of the last step is all zeros
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (STAMP) tk_x = clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (STAMP) { const unsigned long long t = clock64(); tk_vm += t - tk_x; tk_x = t; }
      __syncthreads();
      if constexpr (STAMP) { const unsigned long long t = clock64(); tk_bar += t - tk_x; tk_x = t; }
      if (kt + 1 < nk) {
        if (!(RF_DBG(p.dbg) & 2)) stage(buf ^ 1, m0, n0, kt + 1);
      } else if (has_next)
        stage(buf ^ 1, m0n, n0n, 0);
      const bool second = !(half_tail && kt == nk - 1);
      if constexpr (LAG) {
        if (kt > 0) mfmas();  // M1 of the previous step, from the carried fragments
        read_frags(0);
        mfmas();
        if (second) read_frags(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
        read_frags(0);
        mfmas();
        if (second) {
          read_frags(1);
          mfmas();
        }
      }
      buf ^= 1;
      if constexpr (STAMP) {
        asm volatile("s_nop 0" ::: "memory");
        tk_mma += clock64() - tk_x;
      }
    }
    if constexpr (LAG) {
      if (!half_tail) mfmas();  // M1 of the last step
    }
    if constexpr (STAMP) tk_x = clock64();

    // ---- epilogue: strips overlay the buffer of the last K step (buf ^ 1 now); buf holds the next tile's step 0 ----
    __syncthreads();  // every wave is done reading that buffer
    if constexpr (!LN) {
      // strip / global offsets of this lane's chunks, rebuilt per tile from an opaque copy of the lane id: computed once
      // before the tile loop they stay live through the K loop, which has no registers to spare at BN = 288 (29 spilled)
      int soff[NIT], goff[NIT];
      {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
          const int idx = ln + 64 * t;
          const int r = idx / CPRW, c = idx % CPRW;
          soff[t] = r * PITCHW + c * 16;
          goff[t] = CS ? ((r << 16) | (c * EPC)) : r * p.ldc + c * EPC;  // CS: (strip row, column) pair, resolved per tile
        }
      }
      constexpr int WSTRIDE = RP * PITCHW;
      char* const strip = smem + (buf ^ 1) * STAGE_BYTES + wave * WSTRIDE;
      char* const Cw = (char*)p.C + ((int64_t)(m0 + wm * TM) * p.ldc + n0 + wn * TN) * ESZ;
      const char* const Rw = HAS_RES ? (const char*)p.residual + ((int64_t)(m0 + wm * TM) * p.ldc + n0 + wn * TN) * 4 : nullptr;
#pragma unroll
      for (int r0 = 0; r0 < TM; r0 += RP) {
        // write RP rows of the wave's block: MFMA row tile i covers rows 16*i .. 16*i+15 (row = fr)
#pragma unroll
        for (int ii = 0; ii < (RP >= 16 ? RP / 16 : 1); ++ii) {
          const int i = r0 / 16 + ii;
          const bool mine = RP >= 16 || (fr >> 3) == ((r0 >> 3) & 1);
          if (mine) {
            char* lrow = strip + (RP >= 16 ? ii * 16 + fr : (fr & 7)) * PITCHW;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int nl = j * 16 + 4 * fq;
              const float v0 = fmaxf(acc[i][j][0], lo), v1 = fmaxf(acc[i][j][1], lo);
              const float v2 = fmaxf(acc[i][j][2], lo), v3 = fmaxf(acc[i][j][3], lo);
              if constexpr (OUT_F32) {
                *(f32x4*)(lrow + nl * 4) = (f32x4){v0, v1, v2, v3};
              } else {
                uint2 o;
                o.x = rf_pack2_h16(v0, v1);
                o.y = rf_pack2_h16(v2, v3);
                *(uint2*)(lrow + nl * 2) = o;
              }
            }
          }
        }
        // compiler fence: hipcc otherwise sinks the strip reads below into the lane-masked write block above (seen with
        // the 8-row passes: only the writing half of the lanes then read the strip back)
        asm volatile("" ::: "memory");
        // read back 16-byte chunks of consecutive columns and store whole lines
        char* const Cp = Cw + (int64_t)r0 * p.ldc * ESZ;
        const char* const Rp = HAS_RES ? Rw + (int64_t)r0 * p.ldc * 4 : nullptr;
        if constexpr (CS) {
          if (p.c_cc == 32) {
            // head-major groups of 32 columns (csrc/tied.hip): one store instruction = the 64-byte pieces of 16 consecutive
            // rows of ONE group = 1 KB of contiguous memory (consecutive rows of a group are adjacent in that layout)
            const int r16 = lane >> 2, c4 = lane & 3;
            const int abs0 = wn * (TN / 8);        // first 16-byte chunk column of this wave inside the tile
            constexpr int NGI = (TN / 8 + 3) / 4 + ((TN / 8) % 4 ? 1 : 0);
#pragma unroll
            for (int gi = 0; gi < NGI; ++gi) {
              const int absc = ((abs0 >> 2) + gi) * 4 + c4, cc = absc - abs0;
              const bool okc = cc >= 0 && cc < TN / 8;
#pragma unroll
              for (int rr = 0; rr < (RP >= 16 ? RP / 16 : 1); ++rr) {
                const int row = rr * 16 + r16;
                if (okc && row < RP) {
                  const f32x4 v = *(const f32x4*)(strip + row * PITCHW + cc * 16);
                  const int m = m0 + wm * TM + r0 + row, n = n0 + absc * 8;
                  f32x4* dst = (f32x4*)((char*)p.C + (cs_row(m) + (int64_t)(n >> 5) * p.c_co + (n & 31)) * ESZ);
                  if (p.nt_store)
                    __builtin_nontemporal_store(v, dst);
                  else
                    *dst = v;
                }
              }
            }
            continue;
          }
        }
#pragma unroll
        for (int t0 = 0; t0 < NIT; t0 += G) {
          f32x4 res[G], vv[G];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            if constexpr (HAS_RES) {
              res[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
              if (t0 + g < NIT && (NCH % 64 == 0 || lane + 64 * (t0 + g) < NCH) && !(RF_DBG(p.dbg) & 4)) res[g] = *(const f32x4*)(Rp + (unsigned)goff[t0 + g] * 4u);
            }
          }
#pragma unroll
          for (int g = 0; g < G; ++g)
            if (t0 + g < NIT) vv[g] = *(const f32x4*)(strip + soff[t0 + g]);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            if (t0 + g >= NIT || !(NCH % 64 == 0 || lane + 64 * (t0 + g) < NCH)) continue;
            f32x4 v = vv[g];
            if constexpr (HAS_RES) v += res[g];
            f32x4* dst;
            if constexpr (CS) {
              const int m = m0 + wm * TM + r0 + (goff[t0 + g] >> 16), n = n0 + wn * TN + (goff[t0 + g] & 0xffff);
              dst = (f32x4*)((char*)p.C + (cs_row(m) + cs_col(n)) * ESZ);
            } else {
              dst = (f32x4*)(Cp + (unsigned)goff[t0 + g] * (unsigned)ESZ);
            }
            if (RF_DBG(p.dbg) & 4) continue;
            if (p.nt_store)
              __builtin_nontemporal_store(v, dst);
            else
              *dst = v;
          }
        }
      }
    }
    if constexpr (LN) {
      // Fused "LayerNorm of the next sub-layer" (N == BN: the tile spans whole rows).  Per pass of 8 rows a wave moves its
      // accumulator rows through its private strip into a ROW-CONTIGUOUS register image -- lane (r = lane / 8, c8 = lane % 8)
      // holds the 16-byte pieces c8, c8 + 8, ... of row r -- adds the fp32 residual there (whole 128-byte lines per row and
      // instruction, as the plain epilogue) and stores the updated stream.  The half-row sums / sums of squares (three
      // shuffles over the row's 8 lanes) meet the other wave column's in LDS behind a raw workgroup barrier (LDS traffic
      // only: the stores just issued and the next tile's operand DMA stay in flight), then the pass's rows are normalised in
      // registers and leave as 16-bit pieces.  Nothing but the pass's 5-6 pieces per lane is kept: no register image of the
      // whole tile (a first version that kept one and synchronised once per tile spilled 164 registers and lost more than the
      // LayerNorm launch costs).  The separate LayerNorm launch and its re-read of the stream disappear.
      constexpr int RPL = 8, NPASS = TM / RPL, CPR = TN / 4, KCH = (CPR + 7) / 8;
      static_assert(NW * RPL * PITCHW <= 56 * 1024 && 56 * 1024 + BM * 16 <= STAGE_BYTES, "strips / row statistics do not fit a stage buffer");
      char* const strip = smem + (buf ^ 1) * STAGE_BYTES + wave * (RPL * PITCHW);
      float2* const hst = (float2*)(smem + (buf ^ 1) * STAGE_BYTES + 56 * 1024);  // [BM rows][2 wave columns] (sum, sum of squares)
      int ln = lane, frl = fr, fql = fq;  // opaque copies: everything derived from them is rebuilt here, per tile, instead of being
      asm volatile("" : "+v"(ln), "+v"(frl), "+v"(fql));  // hoisted out of the tile loop into registers the K loop has none of
      const int r = ln >> 3, c8 = ln & 7;
      char* const Cw = (char*)p.C + ((int64_t)(m0 + wm * TM + r) * p.ldc + n0 + wn * TN) * 4;
      const char* const Rw = (const char*)p.residual + ((int64_t)(m0 + wm * TM + r) * p.ldc + n0 + wn * TN) * 4;
      const float* const gam = (const float*)(smem + GB_OFF) + wn * TN;
      h16_t* const Lw = (h16_t*)p.ln_out + (int64_t)(m0 + wm * TM + r) * BN + wn * TN;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int r0 = ps * RPL, i = r0 / 16;
        if ((frl >> 3) == ((r0 >> 3) & 1)) {
          char* lrow = strip + (frl & 7) * PITCHW;
#pragma unroll
          for (int j = 0; j < WN; ++j) *(f32x4*)(lrow + (j * 16 + 4 * fql) * 4) = acc[i][j];
        }
        asm volatile("" ::: "memory");  // (keeps the strip reads below out of the lane-masked write block)
        f32x4 v[KCH];
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
          v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (CPR % 8 == 0 || k < KCH - 1 || c8 + 8 * k < CPR)
            v[k] = *(const f32x4*)(Rw + ((int64_t)r0 * p.ldc + (c8 + 8 * k) * 4) * 4);
        }
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
          if (CPR % 8 == 0 || k < KCH - 1 || c8 + 8 * k < CPR) {
            const f32x4 x = *(const f32x4*)(strip + r * PITCHW + (c8 + 8 * k) * 16) + v[k];
            f32x4* dst = (f32x4*)(Cw + ((int64_t)r0 * p.ldc + (c8 + 8 * k) * 4) * 4);
            if (p.nt_store)
              __builtin_nontemporal_store(x, dst);
            else
              *dst = x;
            v[k] = x;
            sm += (x[0] + x[1]) + (x[2] + x[3]);
            sq += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
          }
        }
#pragma unroll
        for (int o = 1; o <= 4; o <<= 1) {
          sm += __shfl_xor(sm, o, 64);
          sq += __shfl_xor(sq, o, 64);
        }
        const int row = wm * TM + r0 + r;
        if (c8 == 0) hst[row * 2 + wn] = make_float2(sm, sq);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // both wave columns' half-row sums of this pass are in place
        asm volatile("" ::: "memory");
        const float2 hb = hst[row * 2 + (wn ^ 1)];
        const float mean = (sm + hb.x) * (1.0f / BN);
        const float rstd = rsqrtf(fmaxf((sq + hb.y) * (1.0f / BN) - mean * mean, 0.f) + p.ln_eps);
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
          if (CPR % 8 == 0 || k < KCH - 1 || c8 + 8 * k < CPR) {
            const int c = (c8 + 8 * k) * 4;
            const f32x4 g4 = *(const f32x4*)(gam + c), b4 = *(const f32x4*)(gam + BN + c);
            const f32x4 o = (v[k] - mean) * rstd * g4 + b4;
            uint2 w;
            w.x = rf_pack2_h16(o[0], o[1]);
            w.y = rf_pack2_h16(o[2], o[3]);
            *(uint2*)(Lw + (int64_t)r0 * BN + c) = w;
          }
        }
        asm volatile("" ::: "memory");
      }
    }
    if constexpr (STAMP) {
      tk_epi += clock64() - tk_x;
      ++ntile;
    }
    if (!has_next) break;
    lid = lid_next;
    m0 = m0n;
    n0 = n0n;
  }
  };
  if (wave < 4 || p.no_lag)
    run(std::false_type{});
  else
    run(std::true_type{});
  if constexpr (STAMP) {
    if (tid == 0 && p.stamps) {
      unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
      o[0] = clock64() - tk_0; o[1] = tk_vm; o[2] = tk_bar; o[3] = tk_mma; o[4] = tk_epi; o[5] = ntile; o[6] = 0; o[7] = 0;
    }
  }
}

template <int BN, bool OUT_F32, bool HAS_RES, bool STAMP = false, bool LN = false, bool CS = false, int BM = 256>
static int launch_fast(const FastP& p, hipStream_t s) {
  constexpr int STAGE = (BM + BN) * 64 * 2;
  auto k = gemm_fast_kernel<BN, OUT_F32, HAS_RES, STAMP, LN, CS, BM>;
  if (const int e = rf_enable_big_lds<gemm_fast_kernel<BN, OUT_F32, HAS_RES, STAMP, LN, CS, BM>>()) return e;
  const int n_cu = rf_num_cus();
  if (n_cu <= 0) return RF_EINVAL;
  const int grid = p.ntiles < n_cu ? p.ntiles : n_cu;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), 2 * STAGE + (LN ? 2 * BN * 4 : 0) + (size_t)p.tilesN * BN * 4, s, p);
  return rf_launch_status();
}

static unsigned long long* g_fast_stamps = nullptr;
void rf_gemm_fast_set_stamps(void* buf) { g_fast_stamps = (unsigned long long*)buf; }

// Returns 1 and launches when the descriptor fits the persistent fast path, 0 when it does not (the caller then uses the
// generic kernel), a negative / HIP error code when the launch failed.
int rf_gemm_fast_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream) {
  *rc = 0;
  static const bool no_fast = rf_env_flag("RF_NO_FAST_GEMM"), no_nt = rf_env_flag("RF_NO_NT_STORE"), no_lag = rf_env_flag("RF_GEMM_NO_LAG");
  if (no_fast) return 0;
  if (d.ab_dtype != RF_H16 || d.a_mode != RF_AMODE_PLAIN || batch != 1) return 0;
  if (d.a_rc > 0 || d.b_rc > 0 || d.kc != d.K) return 0;
  const bool cs = d.c_rc > 0 || d.c_cc > 0;  // split-C layout: bf16 output in whole 16-byte chunks, no residual / LayerNorm
  if (cs && (d.c_dtype != RF_H16 || d.residual || d.ln_out || (d.c_cc > 0 && (d.c_cc % 8 || d.c_co % 8)) ||
             (d.c_rc > 0 && d.c_ro % 8)))
    return 0;
  if (d.M % 256 != 0 || d.M < 16384 || d.K < 64 || d.K % 8 != 0 || d.alpha != 1.0f) return 0;
  if (d.bias_mode == RF_BIAS_ROW || (d.act != RF_ACT_NONE && d.act != RF_ACT_RELU)) return 0;
  // fused next-LayerNorm epilogue: one tile spans whole rows -- the 288-wide pair rows (256 x 288 tiles) and the 384-wide MSA
  // rows (128 x 384 tiles); fp32 C with residual, no activation
  const bool ln = d.ln_out != nullptr;
  if (ln && ((d.N != 288 && d.N != 384) || d.c_dtype != RF_F32 || !d.residual || d.act != RF_ACT_NONE || !d.ln_gamma || !d.ln_beta ||
             ((uintptr_t)d.ln_out % 16) || ((uintptr_t)d.ln_gamma % 16) || ((uintptr_t)d.ln_beta % 16)))
    return 0;
  if (d.a_ri % 8 || d.b_ri % 8 || d.c_ri % 8 || ((uintptr_t)d.A % 16) || ((uintptr_t)d.B % 16) || ((uintptr_t)d.C % 16)) return 0;
  if (d.bias_mode == RF_BIAS_COL && ((uintptr_t)d.bias % 16)) return 0;
  if (d.residual && (d.c_dtype != RF_F32 || ((uintptr_t)d.residual % 16))) return 0;
  if ((int64_t)256 * d.a_ri >= (1ll << 30) || (int64_t)64 * d.c_ri >= (1ll << 28)) return 0;  // 32-bit lane offsets
  // (split-C: the 288-wide tile is at the register limit already and its split epilogue spills into the K loop: 192 first)
  const int bn = cs ? (d.N % 256 == 0 ? 256 : (d.N % 192 == 0 ? 192 : (d.N % 128 == 0 ? 128 : (d.N % 288 == 0 ? 288 : 0))))
                    : (d.N % 256 == 0 ? 256 : (d.N % 288 == 0 ? 288 : (d.N % 192 == 0 ? 192 : (d.N % 128 == 0 ? 128 : 0))));
  if (!bn) return 0;
  FastP p;
  p.A = (const h16_t*)d.A;
  p.B = (const h16_t*)d.B;
  p.C = d.C;
  p.bias = d.bias_mode == RF_BIAS_COL ? d.bias : nullptr;
  p.residual = d.residual;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.lda = (int)d.a_ri; p.ldb = (int)d.b_ri; p.ldc = (int)d.c_ri;
  p.relu = d.act == RF_ACT_RELU;
  const int bm = (ln && d.N == 384) ? 128 : 256;
  p.tilesN = (ln && d.N == 384) ? 1 : d.N / bn;
  const int64_t nt = (int64_t)(d.M / bm) * p.tilesN;
  if (nt > 0x7fffffffLL) return 0;
  p.ntiles = (int)nt;
  p.nt_store = ((int64_t)d.M * d.N * (d.c_dtype == RF_F32 ? 4 : 2) > (64ll << 20)) && !no_nt;
  hipStream_t s = (hipStream_t)stream;
  const bool f32 = d.c_dtype == RF_F32, res = d.residual != nullptr;
  p.stamps = g_fast_stamps;
  p.no_lag = no_lag;
  static int dbg = 0;
  static const int dbg_rc = rf_dbg_env("RF_GEMM_DBG", &dbg);
  if (dbg_rc) { *rc = dbg_rc; return 1; }
  p.dbg = dbg;
  p.ln_out = d.ln_out; p.ln_gamma = d.ln_gamma; p.ln_beta = d.ln_beta; p.ln_eps = d.ln_eps;
  p.c_rc = d.c_rc; p.c_cc = d.c_cc; p.c_ro = d.c_ro; p.c_co = d.c_co;
  p.c_rsh = (d.c_rc > 0 && (d.c_rc & (d.c_rc - 1)) == 0) ? __builtin_ctz(d.c_rc) : -1;
  p.c_csh = (d.c_cc > 0 && (d.c_cc & (d.c_cc - 1)) == 0) ? __builtin_ctz(d.c_cc) : -1;
  if (cs) {
    if (bn == 288) *rc = launch_fast<288, false, false, false, false, true>(p, s);
    else if (bn == 256) *rc = launch_fast<256, false, false, false, false, true>(p, s);
    else if (bn == 192) *rc = launch_fast<192, false, false, false, false, true>(p, s);
    else *rc = launch_fast<128, false, false, false, false, true>(p, s);
    return 1;
  }
#ifdef FAST_STAMP_LN  // timing experiment (tools/gemm_stamps.py LN=1): instrumented twins of the LayerNorm-epilogue kernels
  if (ln && p.stamps) {
    *rc = d.N == 288 ? launch_fast<288, true, true, true, true>(p, s) : launch_fast<384, true, true, true, true, false, 128>(p, s);
    return 1;
  }
#endif
  if (ln) {
    *rc = d.N == 288 ? launch_fast<288, true, true, false, true>(p, s) : launch_fast<384, true, true, false, true, false, 128>(p, s);
    return 1;
  }
  if (p.stamps && bn == 256 && !f32) {  // timing experiment: instrumented twin of the bf16-output 256-wide kernel
    *rc = launch_fast<256, false, false, true>(p, s);
    return 1;
  }
#define RF_FAST(BN_)                                                   \
  if (bn == BN_) {                                                     \
    if (!f32) *rc = launch_fast<BN_, false, false>(p, s);              \
    else if (res) *rc = launch_fast<BN_, true, true>(p, s);            \
    else *rc = launch_fast<BN_, true, false>(p, s);                    \
    return 1;                                                          \
  }
  RF_FAST(256)
  RF_FAST(288)
  RF_FAST(192)
  RF_FAST(128)
#undef RF_FAST
  return 0;
}
