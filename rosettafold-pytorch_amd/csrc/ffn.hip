// Fused feed-forward block for gfx950 (MI355X):  x_res += W2 relu(W1 xn + b1) + b2,  optionally followed by the NEXT layer's
// LayerNorm of the updated row (16-bit copy beside the fp32 stream).  Reference: FeedForward (rf.py:270-281) inside the
// residual wrappers of the encoder / axial layers (rf.py:284-354, 483-560).
//
// Why: as two GEMMs the hidden activations [M, 4 d] make an HBM round trip (0.8 GB per pair-track feed-forward at the
// benchmark size: the first GEMM is bound by writing them, the second by reading them); the two launches are 22 % of a
// forward.  Here the hidden activations never leave the registers.
//
// Shape of the kernel (d = 384 or 288, hidden = 4 d):
//   * one 8-wave workgroup per CU (two waves per SIMD), persistent over tiles of 128 tokens.  Wave (g, c): token group g of 32
//     tokens; in step A token tile c (16 tokens) of the group, in step B column half c for both token tiles.  The LayerNormed
//     input rows of its token tile sit in registers as MFMA-B fragments (X[tok][k], d/32 of them), its half of the group's
//     fp32 output block Y^T[n][tok] (d/2 x 32) sits in accumulators;
//   * the hidden dimension is walked in chunks of 32.  Step A: H^T[h][tok] = W1[h,:] X^T for the chunk's two 16-wide hidden
//     tiles (weights as MFMA-A: a lane holds 4 consecutive h of one token).  Those two accumulators ARE one MFMA-B fragment
//     of the second GEMM once the contraction index is permuted (k slot 8 fq + j  <->  h = 16 (j >> 2) + 4 fq + (j & 3)) --
//     the same permutation is baked into the packed W2 -- so bias + ReLU + 16-bit rounding happen in registers; the two waves
//     of a token group swap their token tiles' fragments lane for lane through 1 KB of LDS and step B runs
//     Y^T[n][tok] += W2[n, chunk] H[tok, chunk] on the wave's half of the columns: no transposition, no HBM;
//   * both weight matrices are PRE-PACKED (host, once per module) into the exact fragment order the kernel reads: one linear
//     stream of 1 KB pieces.  A ring of 24 KB (18 KB) LDS slots filled by global_load_lds carries, per tile, 4 input slots
//     (one per token group: the 32 x d input rows as fragments) and then 2 slots per chunk (W1 chunk, W2 chunk); one
//     s_barrier per slot, 24 (18) MFMAs per wave between barriers, 0.75 KB of LDS fragment reads (ds_read_b128) per MFMA on
//     average, issued a few MFMAs ahead of their use;
//   * epilogue in registers: + bias + residual (fp32), store; the LayerNorm statistics of a row (4 lanes in each of the two
//     waves of its token group) by two shuffles and one 8-byte exchange per pass.
// HBM traffic per token: d x (2 in + 4 residual + 4 out [+ 2 LayerNorm copy]) bytes; weights stream from L2.
#include <type_traits>

#include "common.h"

// compile-time ablation (tuning only, results WRONG when non-zero; tools/ffn_ablation.sh builds one library per value):
// 1 no MFMA, 2 fragment reads only for the first pieces, 4 DMAs into the dump area, 8 no epilogue, 16 no DMA at all, 32 no exchange,
// 64 residual not added, 128 no output stores, 256 no residual slots, 512 residual slots filled from the (L2-resident) weights,
// 1024 every workgroup walks the chunks in the same order, 2048 no LayerNorm-copy stores, 4096 no pair exchange of the row statistics
#ifndef FFN_ABL
#define FFN_ABL 0
#endif
// Half-step stagger (round 4, default on; -DFFN_STAGGER=0 builds the lock-step form for A/B timing).  The two waves of a SIMD
// used to be the two column halves of ONE token group: same phase, same barrier, their LDS bursts, DMA issue and MFMA clusters
// coincide (the guide's "two waves per SIMD in lock-step" pattern).  Now a token group is a pair of NEIGHBOURING waves (w, w ^ 1:
// different SIMDs), waves 0-3 own groups 0-1, waves 4-7 groups 2-3, and waves 4-7 run ONE ring step behind waves 0-3: while
// a SIMD's early wave is in step A (24 KB of W1 fragments, 24 MFMAs, pack + exchange) its late partner is in step B of the
// previous chunk (12 KB of W2 fragments, 24 MFMAs), and the late wave's last step B overlaps the early wave's epilogue (the
// HBM burst the matrix pipe used to sit out).  One s_barrier per ring step as before; a slot is recycled one step later (ring
// distance NSTG - 2), the LayerNorm row statistics of a token group are exchanged through LDS flags instead of barriers.
#ifndef FFN_STAGGER
#define FFN_STAGGER 1
#endif

struct FfnP {
  const h16_t* X;   // [M, D] LayerNormed input (ldx elements between rows)
  const h16_t* Wp;  // packed weights: per chunk [2 KS pieces of W1][NT pieces of W2], 512 elements each (rf_ffn_fused doc)
  const float* b1;  // [4 D .. any multiple of 32]
  const float* b2;  // [D]
  const float* res; // [M, D] fp32 (may alias out)
  float* out;       // [M, D] fp32
  h16_t* ln;        // optional [M, D]: gamma * (out - mean) * rstd + beta
  const float* gamma;
  const float* beta;
  int64_t ldx, ldr, ldo, ldn;
  float eps;
  int ntiles, nchunks;
};

__device__ __forceinline__ void ffn_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) unsigned ffn_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned ffn_u32x2;
union FfnFrag {
  h16x8 v;
  unsigned u[4];
  ffn_u32x4 q;
};

// LDS accesses of the half-exchange from inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every ds_write that follows a
// global_load_lds (it cannot see that they do not alias), which would drain the weight ring once per chunk
__device__ __forceinline__ void ffn_lds_write16(unsigned addr, ffn_u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ ffn_u32x4 ffn_lds_read16(unsigned addr) {
  ffn_u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void ffn_lds_write8(unsigned addr, float a, float b) {
  const ffn_u32x2 v = {__float_as_uint(a), __float_as_uint(b)};
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ ffn_u32x2 ffn_lds_read8(unsigned addr) {
  ffn_u32x2 v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// Fragment reads from inline asm with hand-counted waits: (1) hipcc serialises "read, wait lgkmcnt(0), use" whatever the
// source order, (2) it puts s_waitcnt vmcnt(0) in front of LDS reads that follow a global_load_lds in the same block.  Each
// wait names the register it releases, so no consumer can move above it; LDS operations of a wave complete in order.
#define FFN_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define FFN_LANDED(n, reg) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(n))
#define FFN_LANDED_N(n, reg) FFN_LANDED(n, reg)

// phase timing (tuning build -DFFN_STAMP: s_memtime after every phase of a step, waves 0 and 7 of workgroup 0 write their
// totals to the first words of `out`: the results are garbage there)
#ifdef FFN_STAMP
#define FFN_T(i)                                                              \
  {                                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter();             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        \
    tacc[i] += now_ - tlast;                                                  \
    tlast = now_;                                                             \
  }
#else
#define FFN_T(i)
#endif

template <int D>
__global__ __launch_bounds__(512) void ffn_fused_kernel(const FfnP p) {
#ifdef FFN_STAMP
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  constexpr int KS = D / 32, NT = D / 16, NTH = NT / 2;
  constexpr int PIECES = NT;                  // 1 KB pieces per slot: 2 KS = NT for W1, NT for W2, 2 KS for a group's input rows
  static_assert(2 * KS == NT, "slot shapes");
  constexpr int SLOT = PIECES * 1024;
  constexpr int NSTG = (144 * 1024) / SLOT;   // 6 (D = 384) or 8 (D = 288)
  constexpr int PD = (PIECES + 7) / 8;        // DMA instructions per wave and slot (padded with dummies: uniform count)
  constexpr int DUMP = NSTG * SLOT;
  constexpr int XB_OFF = DUMP + 1024;         // 8 x 1 KB: lane-for-lane exchange between the two waves of a token group
  constexpr int FLAG_OFF = XB_OFF + 8192;     // 8 words: sequence flags of the epilogue's pair exchange (stagger form)
  constexpr int B1_OFF = FLAG_OFF + 64;
  constexpr bool STG = FFN_STAGGER != 0;
  constexpr int DIST = STG ? NSTG - 2 : NSTG - 1;  // ring steps between the issue of a slot's DMAs and its first consumer
  constexpr int PFA = 6, PFB = 4;             // fragment reads in flight ahead of the MFMAs that use them (steps A / B)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // token group, column half (= token tile in step A).  Stagger form: the pair (w, w ^ 1) is a group -- two SIMDs -- and the
  // SIMD partners w, w + 4 are different groups, the second one ring step behind the first
  const int g = STG ? wave >> 1 : wave & 3, ch = STG ? wave & 1 : wave >> 2;
  const int pw = STG ? wave ^ 1 : wave ^ 4;   // the other wave of this token group
  const bool late = STG && wave >= 4;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int CPIECES = (3 * D * 4 + 1023) / 1024;  // constants slot: b2 | gamma | beta (fp32)
  constexpr int S1 = 2 * NTH;                 // posted stores per wave and tile (fp32 rows); twice that with the LayerNorm copy

  for (int i = tid; i < p.nchunks * 32; i += 512) ((float*)(smem + B1_OFF))[i] = p.b1[i];
  if (tid < 16) ((unsigned*)(smem + FLAG_OFF))[tid] = 0u;
  __syncthreads();  // (before any DMA is in flight)

  // ---- producer side: the ring walks the stream
  //   tile -> [input rows of token group 0..3] [W1 chunk c0][W2 chunk c0][W1 chunk c0+1] ... [constants b2 | gamma | beta]
  // as 1 KB pieces in the order the consumer reads them.  Per-lane source offsets of an input slot (tile independent):
  // piece q = tt * KS + ks  ->  X[32 g + 16 tt + fr][32 ks + 8 fq ..]  (the MFMA-B fragment of token tile tt, k step ks)
  int xsrc[PD], wsrc[PD];
#pragma unroll
  for (int t = 0; t < PD; ++t) {
    const int q = t * 8 + wave;
    xsrc[t] = (int)(((int64_t)((q / KS) * 16 + fr) * p.ldx + (q % KS) * 32 + fq * 8) * 2);
    wsrc[t] = q * 1024 + lane * 16;
  }
  // Every workgroup walks the chunks in its own rotation (the sum over chunks does not care): the CUs of an XCD do not all ask
  // their L2 for the same 24 KB at the same moment.
  const int c0 = (FFN_ABL & 1024) ? 0 : (int)((blockIdx.x >> 3) * 5u % (unsigned)p.nchunks);
  auto rot = [&](int c) { const int r = c + c0; return r >= p.nchunks ? r - p.nchunks : r; };
  const int SPT = 5 + 2 * p.nchunks;    // ring steps per tile
  int f_it = 0, f_pos = 0, f_slot = 0;  // prefetch cursor: tile iteration, position in the tile's step sequence, ring slot
  // The PD DMA instructions of a step are NOT issued in one burst behind the barrier: eight waves doing that queue up on the
  // CU's address path for 500-700 cycles per step (phase stamps: a third of the kernel).  prep() moves the cursor, dma(t)
  // issues piece t; the MFMA loops of steps A and B call dma() between their MFMAs.
  int i_kind = 3;          // 0 input rows, 1 weights, 2 constants, 3 beyond the last tile (dummies)
  const char* i_base = nullptr;
  char* i_dst = nullptr;
  auto prep = [&]() {
    const int tile = blockIdx.x + f_it * gridDim.x;
    i_dst = smem + f_slot * SLOT;
    if (++f_slot == NSTG) f_slot = 0;
    const int pos = f_pos;
    if (++f_pos == SPT) {
      f_pos = 0;
      ++f_it;
    }
    if (tile >= p.ntiles) {
      i_kind = 3;
    } else if (pos < 4) {
      i_kind = 0;
      i_base = (const char*)(p.X + ((int64_t)tile * 128 + pos * 32) * p.ldx);
    } else if (pos < SPT - 1) {
      const int k = pos - 4;  // W1 of chunk k >> 1 (even k) or W2 (odd k), chunks in this workgroup's rotation
      i_kind = 1;
      i_base = (const char*)p.Wp + (int64_t)(2 * rot(k >> 1) + (k & 1)) * SLOT;
    } else {
      i_kind = 2;
    }
  };
  // (one uniform branch per kind of slot, each with its own offsets: a select between the offset arrays makes hipcc move them
  // to scratch memory and index them dynamically -- a scratch load plus s_waitcnt vmcnt(0) per step drains the ring)
  auto dma = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const int q = t * 8 + wave;
    if (FFN_ABL & 16) return;
    if (i_kind == 1 && q < PIECES && !(FFN_ABL & 4)) {
      ffn_glds16(i_base + wsrc[t], i_dst + q * 1024);
    } else if (i_kind == 0 && q < PIECES) {
      ffn_glds16(i_base + xsrc[t], i_dst + q * 1024);
    } else if (i_kind == 2 && q < CPIECES) {
      // b2 | gamma | beta as one virtual array of 16-byte cells j = 64 q + lane
      const int j = q * 64 + lane;
      const float* src = j < D / 4 ? p.b2 + 4 * j : (j < D / 2 ? p.gamma + (4 * j - D) : p.beta + (4 * j - 2 * D));
      const bool ok = j < 3 * D / 4 && (j < D / 4 || p.ln);
      ffn_glds16(ok ? (const void*)src : (const void*)p.b2, i_dst + q * 1024);
    } else {
      ffn_glds16(p.Wp, smem + DUMP);  // every wave issues PD instructions per step: the counted vmcnt relies on it
    }
  };
  auto dma_all = [&]() {
    dma(std::integral_constant<int, 0>{});
    if constexpr (PD > 1) dma(std::integral_constant<int, 1>{});
    if constexpr (PD > 2) dma(std::integral_constant<int, 2>{});
    static_assert(PD <= 3, "pieces per wave and step");
  };

  // ---- consumer side ---------------------------------------------------------------------------------------------------
  // vmcnt retires in order.  Besides its DMA pieces a wave has, per tile, the residual loads into its accumulators (issued at
  // the start of the tile, first needed in step B of the first chunk) and the posted stores of its epilogue.  For the NSTG - 1
  // steps after an epilogue those sit between the DMAs already issued and the DMA being waited for and may stay in flight: the
  // count grows by their number (capped at the 6-bit counter's 63, which only waits for a few of the oldest stores);
  // from then on the standard count has them retired.
  constexpr int WAIT0 = PD * (DIST - 1);
  constexpr int WAITF = WAIT0 + S1;                                 // first tile: + the residual loads
  constexpr int WAIT1 = WAIT0 + 2 * S1 < 63 ? WAIT0 + 2 * S1 : 63;  // + stores (fp32 rows) + residual loads (as many)
  constexpr int LNST = (NTH % 2 == 0) ? S1 / 2 : S1;                // stores of the LayerNorm copy (16-byte pieces when the column tiles pair up)
  constexpr int WAIT2 = WAIT0 + 2 * S1 + LNST < 63 ? WAIT0 + 2 * S1 + LNST : 63;  // + the LayerNorm copy's stores
  int c_slot = late ? NSTG - 1 : 0;  // (a late wave's dummy first step moves it to slot 0)
  int post = DIST;  // steps left whose DMAs were issued before this tile's residual loads (and the last epilogue's stores)
  bool first = true;
  auto step = [&]() -> unsigned {
    if (post > 0) {
      --post;
      if (first)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITF) : "memory");
      else if (p.ln)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT2) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT0) : "memory");  // this wave's pieces of the slot; the barrier covers the rest
    }
    FFN_T(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's exchange write
    __builtin_amdgcn_s_barrier();
    FFN_T(1)
    prep();  // the DMAs of this step go into the slot of the previous step: every wave has consumed it (its fragments fed MFMAs already issued)
    FFN_T(2)
    const unsigned st = (unsigned)(c_slot * SLOT + lane * 16);
    if (++c_slot == NSTG) c_slot = 0;
    return st;
  };
  const unsigned xb_own = (unsigned)(XB_OFF + wave * 1024 + lane * 16), xb_oth = (unsigned)(XB_OFF + pw * 1024 + lane * 16);
  // pair exchange of the epilogue without a workgroup barrier (the halves of the workgroup are in different ring steps there):
  // publish = data write, then the flag write (a wave's LDS operations complete in order); acquire = spin on the partner's flag
  const unsigned fl_own = (unsigned)(FLAG_OFF + wave * 4), fl_oth = (unsigned)(FLAG_OFF + pw * 4);
  unsigned fl_seq = 0;
  auto pair_publish = [&]() {
    ++fl_seq;
    asm volatile("ds_write_b32 %0, %1" ::"v"(fl_own), "v"(fl_seq) : "memory");
  };
  auto pair_acquire = [&]() {
    unsigned v;
    do {
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(fl_oth) : "memory");
    } while ((unsigned)__builtin_amdgcn_readfirstlane((int)v) < fl_seq);  // (the partner may already have published its next one)
  };
  const unsigned b1_rd = (unsigned)(B1_OFF + fq * 16);  // + 128 c: b1[32 c + 4 fq ..], + 64: the second hidden tile
  // Output-column order of a wave's NTH column tiles.  Plain: tile i, MFMA row r -> column 16 i + r, a lane (rows 4 fq .. 4 fq + 3)
  // holds 4 consecutive columns per tile.  PAIRED (even NTH; the packed W2 carries the same row permutation, ops.ffn_pack): tile i,
  // row r -> column 32 (i >> 1) + 8 (r >> 2) + 4 (i & 1) + (r & 3): the lane's values of tiles 2 q and 2 q + 1 are EIGHT consecutive
  // columns, so the LayerNorm copy leaves as 16-byte pieces (8-byte pieces cost ~2x per byte: FFN_ABL 2048: 462 -> 423 us) and the
  // fp32 rows as 32 contiguous bytes per lane.
  constexpr bool PAIRED = NTH % 2 == 0;
  auto ctile = [](int i) constexpr { return PAIRED ? 32 * (i >> 1) + 4 * (i & 1) : 16 * i; };  // column offset of tile i from col0
  const int col0 = ch * (NTH * 16) + (PAIRED ? 8 : 4) * fq;

#pragma unroll 1
  for (int s = 0; s < DIST; ++s) {
    prep();
    dma_all();
  }
  // one ring step without consumption: barrier + this wave's share of the step's DMAs.  Late waves take it first, early waves
  // last, so every wave of the workgroup executes the same number of barriers
  auto idle_step = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT0) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    prep();
    if (++c_slot == NSTG) c_slot = 0;
    dma_all();
  };
  // (measured: priority 1 for the late half: pair shape 574-580 -> 560 us, MSA shape unchanged: profiles/r04_ffn_bench_prio.log)
#ifndef FFN_LATE_PRIO
#define FFN_LATE_PRIO 1
#endif
  if (late && FFN_LATE_PRIO) __builtin_amdgcn_s_setprio(FFN_LATE_PRIO);
  if (late) idle_step();

  for (int it = 0;; ++it) {
    const int tile = blockIdx.x + it * gridDim.x;
    if (tile >= p.ntiles) break;
    // ---- the accumulators start from the residual rows.  Loads from inline asm: the compiler does not know they are in flight
    // (it would wait for them, and with them for the whole ring, in front of the chunk loop); they are older than every DMA
    // issued from here on, so the counted wait of the first step B (issued 5 steps later) retires them.
    f32x4 y[NTH][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const float* rrow = p.res + ((int64_t)tile * 128 + g * 32 + tt * 16 + fr) * p.ldr + col0;
#pragma unroll
      for (int i = 0; i < NTH; ++i) {
        if (FFN_ABL & (8 | 64))
          y[i][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        else
          asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(y[i][tt]) : "v"(rrow), "i"(ctile(i) * 4));
      }
    }
    // ---- input rows of this wave: token tile ch of its group, KS fragments, resident for the tile ----------------------------
    h16x8 xf[KS];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned st = step();
      dma_all();
      if (w == g) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) FFN_RD(xf[ks], st + ch * (KS * 1024), ks * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
      }
    }

#pragma unroll 1
    for (int c = 0; c < p.nchunks; ++c) {
      // ---- step A: H^T[h][tok] for the chunk's 32 hidden units and this wave's 16 tokens (bias in the accumulator init) ----
      f32x4 h[2];
      {
        const unsigned st = step();
        h16x8 a[PFA];
        FFN_RD(h[0], b1_rd + rot(c) * 128, 0);
        FFN_RD(h[1], b1_rd + rot(c) * 128, 64);
#pragma unroll
        for (int k = 0; k < PFA; ++k) FFN_RD(a[k], st, k * 1024);
        FFN_LANDED(PFA, h[0]);
        FFN_LANDED(PFA, h[1]);
#pragma unroll
        for (int k = 0; k < 2 * KS; ++k) {   // piece k = (ks = k >> 1, ht = k & 1)
          // the reads after piece k that are still allowed in flight: k + 1 .. k + PFA - 1 (fewer at the end)
          if (2 * KS - 1 - k >= PFA - 1) FFN_LANDED(PFA - 1, a[k % PFA]); else FFN_LANDED_N(2 * KS - 1 - k, a[k % PFA]);
          // W1 tile as MFMA-A, input tile as MFMA-B: lane holds H^T[h = 16 ht + 4 fq .. +3][tok = 16 ch + fr]
          if (!(FFN_ABL & 1)) h[k & 1] = rf_mfma16(a[k % PFA], xf[k >> 1], h[k & 1], 0, 0, 0);
          if (k + PFA < 2 * KS && !(FFN_ABL & 2)) FFN_RD(a[k % PFA], st, (k + PFA) * 1024);
          if (k == 2) dma(std::integral_constant<int, 0>{});
          if (PD > 1 && k == 2 + (2 * KS) / 3) dma(std::integral_constant<int, (PD > 1 ? 1 : 0)>{});
          if (PD > 2 && k == 2 + 2 * ((2 * KS) / 3)) dma(std::integral_constant<int, (PD > 2 ? 2 : 0)>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FFN_T(3)
      // ReLU + rounding in registers: the accumulators of the two hidden tiles ARE this token tile's MFMA-B fragment of step B
      // (k slot 8 fq + j <-> h = 16 (j >> 2) + 4 fq + (j & 3)); the group's other token tile comes from the partner wave
      FfnFrag own;
      own.u[0] = rf_pack2_h16(fmaxf(h[0][0], 0.f), fmaxf(h[0][1], 0.f));
      own.u[1] = rf_pack2_h16(fmaxf(h[0][2], 0.f), fmaxf(h[0][3], 0.f));
      own.u[2] = rf_pack2_h16(fmaxf(h[1][0], 0.f), fmaxf(h[1][1], 0.f));
      own.u[3] = rf_pack2_h16(fmaxf(h[1][2], 0.f), fmaxf(h[1][3], 0.f));
      if (!(FFN_ABL & 32)) ffn_lds_write16(xb_own, own.q);
      // ---- step B: Y^T[n][tok] += W2[n, chunk] H[tok, chunk] on this wave's half of the columns, both token tiles ---------
      {
        if (c == 0) {  // the residual loads of this tile: 5 steps' DMAs are younger
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * 5) : "memory");
#pragma unroll
          for (int i = 0; i < NTH; ++i) {
            asm volatile("" : "+v"(y[i][0]));
            asm volatile("" : "+v"(y[i][1]));
          }
        }
        const unsigned st = step() + ch * (NTH * 1024);
        FfnFrag oth;
        h16x8 a[PFB];
        if (!(FFN_ABL & 32)) FFN_RD(oth.v, xb_oth, 0); else oth.v = own.v;
#pragma unroll
        for (int k = 0; k < PFB; ++k) FFN_RD(a[k], st, k * 1024);
        FFN_LANDED(PFB, oth.v);
        FFN_T(4)
        const h16x8 hf0 = ch ? oth.v : own.v, hf1 = ch ? own.v : oth.v;
#pragma unroll
        for (int i = 0; i < NTH; ++i) {
          if (NTH - 1 - i >= PFB - 1) FFN_LANDED(PFB - 1, a[i % PFB]); else FFN_LANDED_N(NTH - 1 - i, a[i % PFB]);
          // W2 tile as MFMA-A (its k slots permuted like the hidden fragment), hidden tile as MFMA-B:
          // lane holds Y^T[n = 16 nt + 4 fq .. +3][tok = 16 tt + fr]
          if (!(FFN_ABL & 1)) {
            y[i][0] = rf_mfma16(a[i % PFB], hf0, y[i][0], 0, 0, 0);
            y[i][1] = rf_mfma16(a[i % PFB], hf1, y[i][1], 0, 0, 0);
          }
          if (i + PFB < NTH && !(FFN_ABL & 2)) FFN_RD(a[i % PFB], st, (i + PFB) * 1024);
          if (i == 1) dma(std::integral_constant<int, 0>{});
          if (PD > 1 && i == 1 + NTH / 3) dma(std::integral_constant<int, (PD > 1 ? 1 : 0)>{});
          if (PD > 2 && i == 1 + 2 * (NTH / 3)) dma(std::integral_constant<int, (PD > 2 ? 2 : 0)>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    FFN_T(5)
    // ---- epilogue: bias, store, LayerNorm of the new row (a token's row: 4 lanes in each wave of its group) -----------------
    const unsigned cst = step() - lane * 16 + col0 * 4;  // constants slot: b2 at + 0, gamma at + 4 D, beta at + 8 D
    dma_all();
    if (FFN_ABL & 8) continue;
    float sm[2], sq[2];
    {
      f32x4 b2[NTH];
#pragma unroll
      for (int i = 0; i < NTH; ++i) FFN_RD(b2[i], cst, ctile(i) * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NTH; ++i) asm volatile("" : "+v"(b2[i]));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        float* orow = p.out + ((int64_t)tile * 128 + g * 32 + tt * 16 + fr) * p.ldo + col0;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NTH; ++i) {
          y[i][tt] += b2[i];
          if (!(FFN_ABL & 128)) *(f32x4*)(orow + ctile(i)) = y[i][tt];
          s += (y[i][tt][0] + y[i][tt][1]) + (y[i][tt][2] + y[i][tt][3]);
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        sm[tt] = s;
      }
    }
    if (p.ln) {
      ffn_lds_write8(xb_own, sm[0], sm[1]);
      if constexpr (STG) {
        pair_publish();
        if (!(FFN_ABL & 4096)) pair_acquire();
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      const ffn_u32x2 o1 = ffn_lds_read8(xb_oth);
      const float mean[2] = {(sm[0] + __uint_as_float(o1.x)) * (1.f / D), (sm[1] + __uint_as_float(o1.y)) * (1.f / D)};
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NTH; ++i) {
          y[i][tt] -= mean[tt];
          q += (y[i][tt][0] * y[i][tt][0] + y[i][tt][1] * y[i][tt][1]) + (y[i][tt][2] * y[i][tt][2] + y[i][tt][3] * y[i][tt][3]);
        }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        sq[tt] = q;
      }
      ffn_lds_write8(xb_own + 8, sq[0], sq[1]);   // (the other 8 bytes of the lane's 16-byte cell: the partner may still be reading the first)
      if constexpr (STG) {
        pair_publish();
        if (!(FFN_ABL & 4096)) pair_acquire();
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      const ffn_u32x2 o2 = ffn_lds_read8(xb_oth + 8);
      float rstd[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) rstd[tt] = rsqrtf((sq[tt] + __uint_as_float(tt ? o2.y : o2.x)) * (1.f / D) + p.eps);
      h16_t* lrow = p.ln + ((int64_t)tile * 128 + g * 32 + fr) * p.ldn + col0;
      constexpr int GB = PAIRED ? 4 : 3;  // column tiles per batch of gamma / beta reads
      static_assert(NTH % GB == 0, "gamma / beta read batches");
#pragma unroll
      for (int b = 0; b < NTH / GB; ++b) {
        f32x4 gm[GB], be[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
          FFN_RD(gm[k], cst, ctile(b * GB + k) * 4 + D * 4);
          FFN_RD(be[k], cst, ctile(b * GB + k) * 4 + D * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < GB; ++k) {
          asm volatile("" : "+v"(gm[k]));
          asm volatile("" : "+v"(be[k]));
          const int i = b * GB + k;
          if constexpr (PAIRED) {
            if (k & 1) continue;   // tiles i, i + 1 together: eight consecutive columns of the lane, one 16-byte store
            asm volatile("" : "+v"(gm[k + 1]));
            asm volatile("" : "+v"(be[k + 1]));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
              ffn_u32x4 o;
              o.x = rf_pack2_h16(y[i][tt][0] * rstd[tt] * gm[k][0] + be[k][0], y[i][tt][1] * rstd[tt] * gm[k][1] + be[k][1]);
              o.y = rf_pack2_h16(y[i][tt][2] * rstd[tt] * gm[k][2] + be[k][2], y[i][tt][3] * rstd[tt] * gm[k][3] + be[k][3]);
              o.z = rf_pack2_h16(y[i + 1][tt][0] * rstd[tt] * gm[k + 1][0] + be[k + 1][0], y[i + 1][tt][1] * rstd[tt] * gm[k + 1][1] + be[k + 1][1]);
              o.w = rf_pack2_h16(y[i + 1][tt][2] * rstd[tt] * gm[k + 1][2] + be[k + 1][2], y[i + 1][tt][3] * rstd[tt] * gm[k + 1][3] + be[k + 1][3]);
              if (!(FFN_ABL & 2048)) *(ffn_u32x4*)(lrow + (int64_t)(tt * 16) * p.ldn + ctile(i)) = o;
            }
          } else {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
              ffn_u32x2 o;
              o.x = rf_pack2_h16(y[i][tt][0] * rstd[tt] * gm[k][0] + be[k][0], y[i][tt][1] * rstd[tt] * gm[k][1] + be[k][1]);
              o.y = rf_pack2_h16(y[i][tt][2] * rstd[tt] * gm[k][2] + be[k][2], y[i][tt][3] * rstd[tt] * gm[k][3] + be[k][3]);
              if (!(FFN_ABL & 2048)) *(ffn_u32x2*)(lrow + (int64_t)(tt * 16) * p.ldn + i * 16) = o;
            }
          }
        }
      }
    }
    post = DIST;
    first = false;
    FFN_T(6)
  }
  if (STG && !late) idle_step();  // the late half's last ring step (the constants slot of its epilogue)
#ifdef FFN_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7))
    for (int i = 0; i < 8; ++i) ((unsigned long long*)p.out)[(wave ? 8 : 0) + i] = tacc[i];
#endif
  // drain: dummy / prefetched DMAs must not outlive the workgroup's LDS allocation
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int D>
static int launch_ffn(FfnP& p, int64_t M, hipStream_t s) {
  constexpr int NSTG = (144 * 1024) / (D / 16 * 1024);
  const size_t lds = (size_t)NSTG * (D / 16 * 1024) + 1024 + 8192 + 64 + (size_t)p.nchunks * 128;
  if (lds > 160 * 1024) return RF_EINVAL;
  const int ncu = rf_num_cus();
  if (ncu <= 0) return RF_EINVAL;
  p.ntiles = (int)(M / 128);
  const int grid = p.ntiles < ncu ? p.ntiles : ncu;
  if (const int e = rf_enable_big_lds<ffn_fused_kernel<D>>()) return e;
  hipLaunchKernelGGL(ffn_fused_kernel<D>, dim3((unsigned)grid), dim3(512), lds, s, p);
  return rf_launch_status();
}

// include/rfmi.h: rf_ffn_fused
extern "C" int rf_ffn_fused(const void* x, int64_t ldx, const void* w_packed, const float* b1, const float* b2,
                            const float* residual, int64_t ldr, float* out, int64_t ldo, void* ln_out, int64_t ldn,
                            const float* ln_gamma, const float* ln_beta, float ln_eps, int64_t M, int D, int hidden,
                            void* stream) {
  if (!x || !w_packed || !b1 || !b2 || !residual || !out || M <= 0) return RF_EINVAL;
  if ((D != 288 && D != 384) || hidden < 288 || hidden % 32 || M % 128 || M / 128 > 0x7fffffffLL) return RF_EINVAL;
  if (ln_out && (!ln_gamma || !ln_beta)) return RF_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)w_packed % 16) || ((uintptr_t)residual % 16) || ((uintptr_t)out % 16) ||
      ((uintptr_t)ln_out % 8) || ldx % 8 || ldr % 4 || ldo % 4 || ldn % 4 || ldx < D || ldr < D || ldo < D || (ln_out && ldn < D))
    return RF_EALIGN;
  if ((int64_t)32 * ldx * 2 >= (1ll << 31)) return RF_EINVAL;  // per-lane source offsets inside an input slot are 32-bit
  FfnP p;
  p.X = (const h16_t*)x; p.Wp = (const h16_t*)w_packed; p.b1 = b1; p.b2 = b2; p.res = residual; p.out = out;
  p.ln = (h16_t*)ln_out; p.gamma = ln_gamma; p.beta = ln_beta;
  p.ldx = ldx; p.ldr = ldr; p.ldo = ldo; p.ldn = ldn; p.eps = ln_eps;
  p.nchunks = hidden / 32;
  hipStream_t s = (hipStream_t)stream;
  return D == 384 ? launch_ffn<384>(p, M, s) : launch_ffn<288>(p, M, s);
}
