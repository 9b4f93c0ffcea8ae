// Fused OuterProductMean (rf.py:412-427), second form ("pairs in registers") for gfx950 (MI355X):
//
//   co[b,i,j,(u,v)] = sum_n x[b,n,i,u] y[b,n,j,v]                       (stage 1, MFMA, K = N)
//   out[b,i,j,o]    = sum_k LN(co[b,i,j,:])[k] W[o,k] + bias[o]         (stage 2, MFMA, K = 1024)
//   [ y[b,i,j,:]    = LN2(out[b,i,j,:])                                  (the consumer's LayerNorm, rf.py:443,486) ]
//
// with the LayerNorm(1024) folded algebraically as in csrc/outer.hip: out = rstd (sum_k co_k W'_k - mu s) + c.
//
// outer.hip splits the OUTPUT COLUMNS over the waves, so every wave needs the whole 64-feature block of all 128 pairs and the
// block makes a round trip through LDS per chunk (stage 1 -> image -> barrier -> stage 2): that chain, not bandwidth, set its
// time (0.24 of the MFMA peak).  Here a wave owns PAIRS and the block never leaves its registers:
//   * tile = 8 i x 16 j pairs, eight waves (g = wave & 3, ch = wave >> 2).  Wave (g, ch) owns residue i = i0 + 2 g + ch in
//     stage 1 and column half ch (9 of the 18 sixteen-wide output tiles) of BOTH residues i0 + 2 g, i0 + 2 g + 1 in stage 2;
//   * the 1024 features are walked as 32 steps of one v (32 features (u, v) for u = 0..31).  Stage 1, step v:
//     D[u][j] = sum_n x[n,i,u] y[n,j,v] with the x fragments of the wave's residue RESIDENT in registers (MFMA-A, 2 x N/32) and
//     the y slice of the step (16 j x N, 4 KB) as MFMA-B from LDS.  The two 16-wide u tiles of that accumulator layout ARE
//     one MFMA-B fragment of stage 2 once the contraction index is permuted (k slot 8 fq + e <-> u = 16 (e >> 2) + 4 fq +
//     (e & 3)); the same permutation is baked into the pre-packed W'.  Statistics (sum, sum of squares per pair) accumulate
//     from the fp32 accumulators; the rounded fragment goes to the partner wave through 1 KB of LDS (double-buffered);
//   * stage 2 of step v - 1 runs in the same barrier interval as stage 1 of step v (software pipeline): Y^T[o][pair] +=
//     W'_v[o, u] co_v[u][pair] for the wave's 9 output tiles and both residues: 18 MFMAs from 9 fragment reads;
//   * ONE ring of 24 KB LDS slots filled by global_load_lds carries, per tile, the x fragments (4 slots: two waves each), then
//     per step the y slice of v and the W' slice of v - 1, then the constants (s | c | gamma2 | beta2); one barrier per slot;
//   * epilogue in registers: folded LayerNorm, the consumer's LayerNorm (a pair's 288 outputs: 4 lanes in each of two waves:
//     two shuffles + an 8-byte exchange per pass), 8-byte stores into the 720-wide feature tensor (or fp32 rows).
// L2 -> LDS stream: 22 KB per step, 0.72 MB per tile of 128 pairs.  Matrix work per wave and barrier: 26 MFMAs.
#include <type_traits>

#include "common.h"

struct OPairsP {
  const h16_t* xt;   // [B, L, 32, N]   x_t[b,i,u,n]  (MSA depth contiguous)
  const h16_t* yt;   // [B, L, 32, N]
  const h16_t* wq;   // [32 v][18 o-tiles][64 lanes][8]: element e of lane 16 fq + fr = (W gamma)[16 ot + fr][(16 (e >> 2) + 4 fq + (e & 3)) * 32 + v]
  const float* s;    // [288] row sums of W gamma (16-bit rounded)
  const float* c;    // [288] W beta + bias
  float* out;        // [B, L, L, 288] fp32 (when y == NULL)
  const float* g2;   // second LayerNorm (when y != NULL)
  const float* b2;
  h16_t* y;
  int64_t y_ld;
  int B, L, ntiles, it_n, jt_n;   // tiles: it_n = L / 8 residue groups x jt_n = L / 16 column groups x B
  float eps, eps2;
};

__device__ __forceinline__ void op_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
typedef __attribute__((ext_vector_type(4))) unsigned op_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned op_u32x2;
union OpFrag {
  h16x8 v;
  unsigned u[4];
  op_u32x4 q;
};
// (LDS accesses that alias nothing the DMAs write go through inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every
// ds_write that follows a global_load_lds and serialises "read, lgkmcnt(0), use" whatever the source order; see csrc/ffn.hip)
__device__ __forceinline__ void op_lds_write16(unsigned addr, op_u32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void op_lds_write8(unsigned addr, float a, float b) {
  const op_u32x2 v = {__float_as_uint(a), __float_as_uint(b)};
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ op_u32x2 op_lds_read8(unsigned addr) {
  op_u32x2 v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
#define OP_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define OP_LANDED(n, reg) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(n))

// phase timing (tuning build -DOP_STAMP: s_memtime after every phase, waves 0 and 7 of workgroup 0 write their totals over the
// first words of the result: garbage there)
#ifdef OP_STAMP
#define OP_T(i)                                                   \
  {                                                               \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            \
    tacc[i] += now_ - tlast;                                      \
    tlast = now_;                                                 \
  }
#else
#define OP_T(i)
#endif

// NKS = N / 32 (stage-1 K steps): 4 (N = 128) or 2 (N = 64)
// NW = waves per workgroup: 8 (one workgroup per CU, tile = 8 residues x 16) or 4 (TWO independent workgroups per CU, tile = 4 x 16:
// the two waves of a SIMD then belong to different barrier domains and drift out of phase, so one computes while the other
// waits, packs or issues DMAs -- with one 8-wave workgroup both sit in the same phase at the same time).
template <int NKS, bool LN2, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void outer_pairs_kernel(const OPairsP p) {
#ifdef OP_STAMP
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  constexpr int NT = 18, NTH = 9;             // 16-wide output tiles: all / per wave
  constexpr int NV = 32;                      // steps (one v each)
  constexpr int HW = NW / 2;                  // residue pairs per tile
  constexpr int PIECES = NW == 8 ? 24 : 22, SLOT = PIECES * 1024, PD = (PIECES + NW - 1) / NW;   // 1 KB pieces per slot: [0, NKS) y slice, [4, 22) W' slice
  constexpr int WOFF = 4;                     // first W' piece of a slot
  constexpr int NSTG = NW == 8 ? 5 : 3;
  constexpr int DUMP = NSTG * SLOT;
  constexpr int XB_OFF = DUMP + 1024;         // [2 parities][NW waves][1 KB]: lane-for-lane exchange between the two waves of a residue pair
  constexpr int XBP = NW * 1024;              // one parity block
  constexpr int XP = 2 * NKS;                 // x fragments of a wave (pieces of its residue): piece f = ut * NKS + ks
  constexpr int CPIECES = 5;                  // constants: s | c | gamma2 | beta2 (4 x 288 fp32 = 4608 bytes)
  constexpr int NXS = NW / 2;                 // x slots per tile (two waves each)
  constexpr int SPT = NXS + NV + 1 + 1;       // ring steps per tile: x slots, steps v = 0 .. 32 (the last: stage 2 only), constants
  constexpr int PF = 3;                       // W' fragment reads in flight ahead of the MFMAs that use them
  constexpr int N = NKS * 32;
  constexpr int S1 = 2 * NTH;                 // posted stores per wave and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave % HW, ch = wave / HW;
  const int fr = lane & 15, fq = lane >> 4;

  // tile t -> (b, jt, it), it fastest
  auto tile_ij = [&](int t, int& b, int& i0, int& j0) __attribute__((always_inline)) {
    const int it = t % p.it_n, r = t / p.it_n;
    const int jt = r % p.jt_n;
    b = r / p.jt_n;
    i0 = it * NW;
    j0 = jt * 16;
  };

  // ---- producer side: the ring walks the stream
  //   tile -> [x fragments of waves 0,1][2,3][4,5][6,7]  { [y slice of v | W' slice of v - 1] for v = 0 .. 32 }  [constants]
  // Per-lane source offsets (tile independent, bytes):
  //   x piece q (q = (w & 1) * XP + f, f = ut * NKS + ks of wave w): xt[b, i(w), 16 ut + fr, 32 ks + 8 fq ..]
  //   y piece ks:  yt[b, j0 + fr, v, 32 ks + 8 fq ..]      W' piece ot: linear
  // The prefetch cursor keeps PER-LANE source pointers that advance by a constant per step (the tile decomposition -- three
  // integer divisions -- and the 64-bit address products are done once per tile, not once per DMA: phase stamps had prep at
  // 15 % and the DMA address arithmetic inside stage 2 at another ~15 % of the kernel).
  int f_it = 0, f_pos = 0, f_slot = 0;
  int i_kind = 4;  // 0 x slot, 1 step slot, 2 constants, 4 beyond the last tile
  int i_pos = 0;
  char* i_dst = nullptr;
  const h16_t* f_x = p.xt;   // this lane's x source of the cursor's tile: xt[b, i0, fr, 8 fq ..]
  const h16_t* f_y = p.yt;   // yt[b, j0 + fr, v, 8 fq ..] of the cursor's step
  const h16_t* f_w = p.wq;   // W' slice of step v - 1, this lane's 16 bytes of piece 0
  auto prep = [&]() __attribute__((always_inline)) {
    const int tile = blockIdx.x + f_it * gridDim.x;
    i_dst = smem + f_slot * SLOT;
    if (++f_slot == NSTG) f_slot = 0;
    i_pos = f_pos;
    if (tile >= p.ntiles) {
      i_kind = 4;
    } else if (f_pos == 0) {
      int b, i0, j0;
      tile_ij(tile, b, i0, j0);
      f_x = p.xt + (((int64_t)b * p.L + i0) * 32 + fr) * N + fq * 8;
      f_y = p.yt + (((int64_t)b * p.L + j0 + fr) * 32) * N + fq * 8;
      f_w = p.wq + lane * 8;
      i_kind = 0;
    } else if (f_pos < NXS) {
      i_kind = 0;
    } else if (f_pos < SPT - 1) {
      i_kind = 1;
    } else {
      i_kind = 2;
    }
    if (++f_pos == SPT) {
      f_pos = 0;
      ++f_it;
    }
  };
  // after the three DMAs of a step slot: the cursor's pointers move to the next v
  auto advance = [&]() __attribute__((always_inline)) {
    if (i_kind == 1) {
      if (i_pos > NXS) f_w += NT * 512;   // (the slot of v = 0 carries no W')
      f_y += N;
    }
  };
  auto dma = [&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    const int q = t * NW + wave;  // piece of the slot this wave issues
    if (i_kind == 1) {
      if (q < NKS && i_pos < NXS + NV) {
        op_glds16(f_y + q * 32, i_dst + q * 1024);
      } else if (q >= WOFF && q < WOFF + NT && i_pos > NXS) {
        op_glds16(f_w + (q - WOFF) * 512, i_dst + q * 1024);
      } else {
        op_glds16(p.wq, smem + DUMP);  // every wave issues PD instructions per step: the counted vmcnt relies on it
      }
      if (t == PD - 1) advance();
    } else if (i_kind == 0) {
      const int w = 2 * i_pos + (q >= XP ? 1 : 0), f = q >= XP ? q - XP : q;
      if (q < 2 * XP) {
        // residue i0 + 2 (w & 3) + (w >> 2), piece f = (ut, ks): + ((i - i0) * 32 + 16 ut) * N + 32 ks elements
        op_glds16(f_x + ((2 * (w % HW) + (w / HW)) * 32 + (f / NKS) * 16) * N + (f % NKS) * 32, i_dst + q * 1024);
      } else {
        op_glds16(p.wq, smem + DUMP);
      }
    } else if (i_kind == 2 && q < CPIECES) {
      // s | c | gamma2 | beta2 as one virtual array of 16-byte cells j = 64 q + lane (72 cells each)
      const int j = q * 64 + lane;
      const float* src = j < 72 ? p.s + 4 * j : (j < 144 ? p.c + 4 * (j - 72) : (j < 216 ? p.g2 + 4 * (j - 144) : p.b2 + 4 * (j - 216)));
      const bool ok = j < 288 && (j < 144 || LN2);
      op_glds16(ok ? (const void*)src : (const void*)p.s, i_dst + q * 1024);
    } else {
      op_glds16(p.wq, smem + DUMP);
    }
  };
  auto dma_all = [&]() __attribute__((always_inline)) {
    dma(std::integral_constant<int, 0>{});
    dma(std::integral_constant<int, 1>{});
    dma(std::integral_constant<int, 2>{});
    if constexpr (PD > 3) {
      dma(std::integral_constant<int, 3>{});
      dma(std::integral_constant<int, 4>{});
      dma(std::integral_constant<int, 5>{});
    }
    static_assert(PD == 3 || PD == 6, "pieces per wave and step");
  };

  // ---- consumer side (counted waits as in csrc/ffn.hip: besides its DMAs a wave only has the posted stores of its epilogue) ----
  constexpr int WAIT0 = PD * (NSTG - 2), WAIT1 = WAIT0 + S1;
  int c_slot = 0, post = 0;
  auto step = [&]() __attribute__((always_inline)) -> unsigned {
    if (post > 0) {
      --post;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT0) : "memory");
    }
    OP_T(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's exchange write
    __builtin_amdgcn_s_barrier();
    OP_T(1)
    prep();  // the DMAs of this step go into the slot of the previous step: every wave has consumed it
    OP_T(2)
    const unsigned st = (unsigned)(c_slot * SLOT + lane * 16);
    if (++c_slot == NSTG) c_slot = 0;
    return st;
  };
  const unsigned xb_own = (unsigned)(XB_OFF + wave * 1024 + lane * 16), xb_oth = (unsigned)(XB_OFF + (wave ^ HW) * 1024 + lane * 16);

#pragma unroll 1
  for (int s_ = 0; s_ < NSTG - 1; ++s_) {
    prep();
    dma_all();
  }

  for (int it = 0;; ++it) {
    const int tile = blockIdx.x + it * gridDim.x;
    if (tile >= p.ntiles) break;
    int b, i0, j0;
    tile_ij(tile, b, i0, j0);
    // ---- x fragments of this wave's residue: MFMA-A, rows u (two tiles of 16), K = n ------------------------------------
    h16x8 xa[2][NKS];
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const unsigned st = step();
      dma_all();
      if (k == (wave >> 1)) {
#pragma unroll
        for (int f = 0; f < XP; ++f) OP_RD(xa[f / NKS][f % NKS], st + (wave & 1) * (XP * 1024), f * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int f = 0; f < XP; ++f) asm volatile("" : "+v"(xa[f / NKS][f % NKS]));
      }
    }
    f32x4 yacc[NTH][2];
#pragma unroll
    for (int k = 0; k < NTH; ++k) yacc[k][0] = yacc[k][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;  // sum / sum of squares of the raw block of pair (i_own, j0 + fr): this lane's 8 u per step
    OpFrag own;                 // the fragment of step v - 1 (this wave's residue)
    own.q = (op_u32x4){0u, 0u, 0u, 0u};

#pragma unroll 1
    for (int v = 0; v <= NV; ++v) {
      const unsigned st = step();
      // One interval = stage 2 of step v - 1 (18 MFMAs: Y^T[o][pair] += W'_{v-1}[o, u] co_{v-1}[u][pair], both residues, this
      // wave's 9 output tiles) INTERLEAVED with stage 1 of step v (8 MFMAs in two dependent chains of N / 32:
      // D[u][j] = sum_n x[n, i_own, u] y[n, j, v]): the chains' latency hides behind the independent stage-2 MFMAs.
      OpFrag oth;
      h16x8 a[PF], yb[NKS];
      const unsigned wst = st + (WOFF + ch * NTH) * 1024;
      OP_RD(oth.v, xb_oth + ((v - 1) & 1) * XBP, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) OP_RD(yb[ks], st, ks * 1024);
#pragma unroll
      for (int k = 0; k < PF; ++k) OP_RD(a[k], wst, k * 1024);
      OP_LANDED(NKS + PF, oth.v);
      OP_T(6)
      const h16x8 hf0 = ch ? oth.v : own.v, hf1 = ch ? own.v : oth.v;  // residue i0 + 2 g (computed by ch = 0), i0 + 2 g + 1
      f32x4 d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      // (at v = 0 the slot holds no W' and `own` / `oth` are stale: the stage-2 products are multiplied away by zeroed
      // fragments instead of branching around 18 MFMAs: hfz = 0)
      const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      const h16x8 h0 = v > 0 ? hf0 : zero8, h1 = v > 0 ? hf1 : zero8;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) OP_LANDED(PF + NKS - 1 - ks, yb[ks]);   // (issued before the W' reads: landed first)
#pragma unroll
      for (int k = 0; k < NTH; ++k) {
        if (NTH - 1 - k >= PF - 1) OP_LANDED(PF - 1, a[k % PF]); else OP_LANDED(NTH - 1 - k, a[k % PF]);
        // W' tile as MFMA-A (k slots permuted like the fragment), block as MFMA-B: lane holds Y^T[o = 16 ot + 4 fq .. +3][j = fr]
        const h16x8 av = v > 0 ? a[k % PF] : zero8;  // (the slot of v = 0 holds no W': whatever bits lie there must not meet the MFMA)
        yacc[k][0] = rf_mfma16(av, h0, yacc[k][0], 0, 0, 0);
        yacc[k][1] = rf_mfma16(av, h1, yacc[k][1], 0, 0, 0);
        if (k < 2 * NKS) {
          // x tile as MFMA-A (rows u), y slice as MFMA-B (columns j): lane holds co[u = 16 ut + 4 fq .. +3][j = fr]
          d[k & 1] = rf_mfma16(xa[k & 1][k >> 1], yb[k >> 1], d[k & 1], 0, 0, 0);
        }
        if (k + PF < NTH) OP_RD(a[k % PF], wst, (k + PF) * 1024);
        if constexpr (PD == 3) {
          if (k == 1) dma(std::integral_constant<int, 0>{});
          if (k == 4) dma(std::integral_constant<int, 1>{});
          if (k == 7) dma(std::integral_constant<int, 2>{});
        } else {
          if (k == 0) dma(std::integral_constant<int, 0>{});
          if (k == 2) dma(std::integral_constant<int, 1>{});
          if (k == 3) dma(std::integral_constant<int, 2>{});
          if (k == 5) dma(std::integral_constant<int, 3>{});
          if (k == 6) dma(std::integral_constant<int, 4>{});
          if (k == 8) dma(std::integral_constant<int, (PD > 3 ? 5 : 0)>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      static_assert(2 * NKS <= NTH, "stage-1 MFMAs ride in the stage-2 loop");
      OP_T(3)
      if (v < NV) {
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
          s1 += (d[ut][0] + d[ut][1]) + (d[ut][2] + d[ut][3]);
          s2 += (d[ut][0] * d[ut][0] + d[ut][1] * d[ut][1]) + (d[ut][2] * d[ut][2] + d[ut][3] * d[ut][3]);
        }
        // the two accumulators ARE the MFMA-B fragment of stage 2 (k slot 8 fq + e <-> u = 16 (e >> 2) + 4 fq + (e & 3))
        own.u[0] = rf_pack2_h16(d[0][0], d[0][1]);
        own.u[1] = rf_pack2_h16(d[0][2], d[0][3]);
        own.u[2] = rf_pack2_h16(d[1][0], d[1][1]);
        own.u[3] = rf_pack2_h16(d[1][2], d[1][3]);
        op_lds_write16(xb_own + (v & 1) * XBP, own.q);
      }
      OP_T(4)
    }

    // ---- statistics of the two residues' pairs: this wave's own residue from its sums, the other from the partner ----------
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float mu_own = s1 * (1.f / 1024.f);
    const float rs_own = rsqrtf(fmaxf(s2 * (1.f / 1024.f) - mu_own * mu_own, 0.f) + p.eps);
    op_lds_write8(xb_own, mu_own, rs_own);
    const unsigned cst = step() - lane * 16;  // constants slot (the barrier inside also publishes the statistics)
    dma_all();
    const op_u32x2 o1 = op_lds_read8(xb_oth);
    const float mu[2] = {ch ? __uint_as_float(o1.x) : mu_own, ch ? mu_own : __uint_as_float(o1.x)};
    const float rs[2] = {ch ? __uint_as_float(o1.y) : rs_own, ch ? rs_own : __uint_as_float(o1.y)};

    // ---- epilogue: folded LayerNorm(1024), then the consumer's LayerNorm(288) or fp32 rows ------------------------------------
    const int col0 = ch * (NTH * 16) + 4 * fq;
    const unsigned crd = cst + col0 * 4;  // s at + 0, c at + 1152, gamma2 at + 2304, beta2 at + 3456
    float sm[2] = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NTH / 3; ++kb) {
      f32x4 sv[3], cv[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        OP_RD(sv[k], crd, (kb * 3 + k) * 64);
        OP_RD(cv[k], crd, (kb * 3 + k) * 64 + 1152);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        asm volatile("" : "+v"(sv[k]));
        asm volatile("" : "+v"(cv[k]));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          f32x4& a = yacc[kb * 3 + k][tt];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = rs[tt] * (a[e] - mu[tt] * sv[k][e]) + cv[k][e];
          sm[tt] += (a[0] + a[1]) + (a[2] + a[3]);
        }
      }
    }
    const int64_t prow0 = ((int64_t)b * p.L + i0 + 2 * g) * p.L + j0 + fr;  // pair row of residue tt = 0; tt = 1: + L
    if constexpr (!LN2) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        float* orow = p.out + (prow0 + (int64_t)tt * p.L) * 288 + col0;
#pragma unroll
        for (int k = 0; k < NTH; ++k) *(f32x4*)(orow + k * 16) = yacc[k][tt];
      }
    } else {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        sm[tt] += __shfl_xor(sm[tt], 16, 64);
        sm[tt] += __shfl_xor(sm[tt], 32, 64);
      }
      op_lds_write8(xb_own + 8, sm[0], sm[1]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const op_u32x2 o2 = op_lds_read8(xb_oth + 8);
      const float mean[2] = {(sm[0] + __uint_as_float(o2.x)) * (1.f / 288.f), (sm[1] + __uint_as_float(o2.y)) * (1.f / 288.f)};
      float sq[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NTH; ++k) {
          yacc[k][tt] -= mean[tt];
          q += (yacc[k][tt][0] * yacc[k][tt][0] + yacc[k][tt][1] * yacc[k][tt][1]) + (yacc[k][tt][2] * yacc[k][tt][2] + yacc[k][tt][3] * yacc[k][tt][3]);
        }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        sq[tt] = q;
      }
      op_lds_write8(xb_own + XBP, sq[0], sq[1]);  // (the other parity block: the partner may still be reading the first exchange)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const op_u32x2 o3 = op_lds_read8(xb_oth + XBP);
      const float r2[2] = {rsqrtf((sq[0] + __uint_as_float(o3.x)) * (1.f / 288.f) + p.eps2),
                           rsqrtf((sq[1] + __uint_as_float(o3.y)) * (1.f / 288.f) + p.eps2)};
#pragma unroll
      for (int kb = 0; kb < NTH / 3; ++kb) {
        f32x4 gm[3], be[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          OP_RD(gm[k], crd, (kb * 3 + k) * 64 + 2304);
          OP_RD(be[k], crd, (kb * 3 + k) * 64 + 3456);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          asm volatile("" : "+v"(gm[k]));
          asm volatile("" : "+v"(be[k]));
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const f32x4 a = yacc[kb * 3 + k][tt];
            op_u32x2 o;
            o.x = rf_pack2_h16(a[0] * r2[tt] * gm[k][0] + be[k][0], a[1] * r2[tt] * gm[k][1] + be[k][1]);
            o.y = rf_pack2_h16(a[2] * r2[tt] * gm[k][2] + be[k][2], a[3] * r2[tt] * gm[k][3] + be[k][3]);
            *(op_u32x2*)(p.y + (prow0 + (int64_t)tt * p.L) * p.y_ld + col0 + (kb * 3 + k) * 16) = o;
          }
        }
      }
    }
    post = NSTG - 1;
    OP_T(5)
  }
#ifdef OP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7))
    for (int i = 0; i < 8; ++i) ((unsigned long long*)(LN2 ? (void*)p.y : (void*)p.out))[(wave ? 8 : 0) + i] = tacc[i];
#endif
  // drain: dummy / prefetched DMAs must not outlive the workgroup's LDS allocation
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NKS, bool LN2, int NW>
static int launch_outer_pairs_nw(OPairsP& p, hipStream_t s) {
  constexpr size_t lds = (size_t)(NW == 8 ? 5 * 24 : 3 * 22) * 1024 + 1024 + 2 * NW * 1024;
  const int ncu = rf_num_cus();
  if (ncu <= 0) return RF_EINVAL;
  if (p.L % NW) return RF_EINVAL;
  p.it_n = p.L / NW;
  p.jt_n = p.L / 16;
  const int64_t nt = (int64_t)p.B * p.it_n * p.jt_n;
  if (nt > 0x7fffffffLL) return RF_EINVAL;
  p.ntiles = (int)nt;
  const int slots = (NW == 8 ? 1 : 2) * ncu;  // workgroups resident at once
  const int grid = p.ntiles < slots ? p.ntiles : slots;
  if (const int e = rf_enable_big_lds<outer_pairs_kernel<NKS, LN2, NW>>()) return e;
  hipLaunchKernelGGL((outer_pairs_kernel<NKS, LN2, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, s, p);
  return rf_launch_status();
}

template <int NKS, bool LN2>
static int launch_outer_pairs(OPairsP& p, hipStream_t s) {
  static const int nw = getenv("RF_OUTER_PAIRS_NW") ? atoi(getenv("RF_OUTER_PAIRS_NW")) : 8;  // (A/B: 4 = two 4-wave workgroups per CU: measured slower, 418-439 vs 346-364 us)
  return nw == 8 ? launch_outer_pairs_nw<NKS, LN2, 8>(p, s) : launch_outer_pairs_nw<NKS, LN2, 4>(p, s);
}

// include/rfmi.h: rf_outer_product_pairs
extern "C" int rf_outer_product_pairs(const void* xt, const void* yt, const void* w_packed, const float* s, const float* c,
                                      float* out, int B, int L, int N, int P, int Dout, float eps, const float* ln2_gamma,
                                      const float* ln2_beta, float ln2_eps, void* y, int64_t y_ld, void* stream) {
  if (!xt || !yt || !w_packed || !s || !c || B <= 0) return RF_EINVAL;
  if (!y && !out) return RF_EINVAL;
  if (y && (!ln2_gamma || !ln2_beta || y_ld < Dout || y_ld % 4 || ((uintptr_t)y % 8) || ((uintptr_t)ln2_gamma % 16) || ((uintptr_t)ln2_beta % 16)))
    return RF_EINVAL;
  if (P != 32 || Dout != 288 || (N != 128 && N != 64) || L % 16 != 0 || L < 16) return RF_EINVAL;  // (other shapes: rf_gemm + rf_layernorm)
  if (((uintptr_t)xt % 16) || ((uintptr_t)yt % 16) || ((uintptr_t)w_packed % 16) || ((uintptr_t)s % 16) || ((uintptr_t)c % 16) ||
      ((uintptr_t)out % 16))
    return RF_EALIGN;
  OPairsP p;
  p.xt = (const h16_t*)xt; p.yt = (const h16_t*)yt; p.wq = (const h16_t*)w_packed; p.s = s; p.c = c; p.out = out;
  p.g2 = ln2_gamma ? ln2_gamma : s; p.b2 = ln2_beta ? ln2_beta : s; p.y = (h16_t*)y; p.y_ld = y_ld;
  p.B = B; p.L = L; p.eps = eps; p.eps2 = ln2_eps;
  hipStream_t st = (hipStream_t)stream;
  if (y) return N == 128 ? launch_outer_pairs<4, true>(p, st) : launch_outer_pairs<2, true>(p, st);
  return N == 128 ? launch_outer_pairs<4, false>(p, st) : launch_outer_pairs<2, false>(p, st);
}
