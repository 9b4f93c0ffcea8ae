// Skinny-K projection GEMM for gfx950 (MI355X): weights in registers, activations streamed once through LDS.
//
// The forward path's largest GEMM family has a SHORT contraction (K = d_pair = 288 or d_msa = 384: the q|k|v projections
// and the first feed-forward layers, rf.py:270-281,313-318,505-518) and a long activation panel (M = 131072 .. 262144 rows).
// With K that short a 256 x 256 output tile holds only 4.5 - 6 K steps: measured on the persistent tile kernel
// (tools/gemm_dbg_sweep.sh) the time is the SUM of DMA (~100 us), epilogue stores (~100 us) and per-tile overhead (~100 us)
// while the matrix pipe is nearly idle -- nothing overlaps, because every tile is a barrier-coupled prologue / K loop /
// epilogue and `s_waitcnt vmcnt` is in-order across DMAs and stores.
//
// Here the roles are turned around.  A workgroup owns a block of BNW output columns for its whole life:
//   * the weight block W[BNW, K] lives in REGISTERS as MFMA fragments (8 waves x WCT column tiles x K/32 steps x 4 VGPRs);
//     it is loaded once per workgroup and never touches LDS;
//   * the activations stream through a ring of [TMR rows x K] LDS tiles (one global_load_lds pass: each activation byte
//     enters the CU once per column block and is read by the waves that need it);
//   * there is NO K loop in the tile sense: one barrier per row tile, then every wave runs its K/32 x WRT x WCT MFMAs from
//     register-resident weights, converts its 16*WRT x 16*WCT block and stores it through a wave-private LDS strip;
//   * DMAs are issued BEFORE the stores of an iteration, so the counted `vmcnt` of the next iteration leaves the stores of
//     the two previous row tiles in flight: stores drain under the following tiles' MFMAs.
// Workgroup (column block nb, slot j) walks row tiles j, j + per, j + 2 per, ...; workgroups of one slot (different nb) sit on
// one XCD and read the same activation tiles at about the same time (one HBM fetch, L2 hits for the others).
#include <type_traits>

#include "common.h"

static __device__ __attribute__((aligned(16))) unsigned int g_wreg_zero16[4];

struct WregP {
  const h16_t* A;  // [M, K] row-major (lda)
  const h16_t* B;  // [N, K] row-major (ldb): nn.Linear weight
  h16_t* C;        // [M, N] bf16 (ldc), or split layout (c_rc / c_cc, see rf_gemm_desc)
  const float* bias;
  int M, N, K;
  int lda, ldb, ldc;
  int relu;
  int nblocks, per;  // column blocks, workgroups per column block
  int ntm;           // row tiles
  int nt_store;
  int c_rc, c_cc, c_rsh, c_csh;
  int64_t c_ro, c_co;
  // row-group scale of the leading rs_ncols columns (rf_gemm_desc.rs): the tied-attention position weights folded into q
  const float* rs;
  int64_t rs_bstride;
  int rs_rpb, rs_cg, rs_ncols;
  float rs_alpha;
};

__device__ __forceinline__ void wreg_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned wreg_pack2(float a, float b) { return rf_pack2_h16(a, b); }

// KS: K / 32;  TMR: rows per activation tile;  WR: wave rows (8 / WR wave columns);  WCT: 16-column tiles per wave;
// NSTG: ring depth;  CS: split-C addressing
template <int KS, int TMR, int WR, int WCT, int NSTG, bool CS>
__global__ __launch_bounds__(512, 2) void gemm_wreg_kernel(const WregP p) {
  constexpr int WC = 8 / WR;
  constexpr int WRT = TMR / WR / 16;          // 16-row tiles per wave
  constexpr int TN = WCT * 16;                // columns per wave
  constexpr int BNW = WC * TN;                // columns per workgroup
  constexpr int S = KS * 4;                   // 16-byte slots per activation row
  constexpr int ROWB = S * 16;                // bytes per activation row
  constexpr int TILE = TMR * ROWB;
  constexpr int NI = (TMR * S + 63) / 64;     // DMA instructions per tile
  constexpr int PD = (NI + 7) / 8;            // per wave (padded with dummies: uniform count)
  constexpr int DUMP = NSTG * TILE;
  constexpr int PITCH = TN * 2 + 16;
  constexpr int STRIP = 16 * WRT * PITCH;
  constexpr int STRIP0 = DUMP + 1024;
  constexpr int CPR = TN * 2 / 16;            // 16-byte chunks per strip row
  constexpr int NCH = 16 * WRT * CPR;
  constexpr int PS = (NCH + 63) / 64;         // store instructions per wave and tile
  static_assert(WRT >= 1 && TMR % (16 * WR) == 0, "wave rows");
  static_assert((TMR * S) % 64 == 0 && NCH % 64 == 0, "whole DMA / store instructions (the counted vmcnt relies on it)");
  static_assert(S % 16 == 0 || S % 8 == 4, "activation row swizzle: K = 384-like (48 slots) or K = 288-like (36 slots)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int fr = lane & 15, fq = lane >> 4;

  // XCD-aware id: consecutive ids (same slot, different column blocks) share an XCD and therefore an L2
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int nb = lid % p.nblocks, slot = lid / p.nblocks;
  const int n0 = nb * BNW + wc * TN;

  // swizzle of the 16-byte slot inside an activation row (on the DMA source and on the fragment reads): rows 64 B-periodic
  // in the banks (S = 36: stride = 4 slots mod 16) get slot ^ h[(row >> 2) & 3], h = {0,3,2,1}; rows that are a multiple of
  // 256 B (S = 48) get slot ^ (row & 15): the 16 lanes of every ds_read_b128 group then hit 16 distinct slots
  auto swz = [](int row) { return S % 16 == 0 ? (row & 15) : ((0x6C >> (((row >> 2) & 3) * 2)) & 3); };

  // ---- weights of this wave: W[n0 + 16 j + fr][32 s + 8 fq .. +7] as MFMA-A fragments, resident for the whole kernel ----
  h16x8 wf[WCT][KS];
  f32x4 bias4[WCT];
  {
#pragma unroll
    for (int j = 0; j < WCT; ++j) {
      const int n = n0 + j * 16 + fr;
      const h16_t* wrow = p.B + (int64_t)(n < p.N ? n : p.N - 1) * p.ldb + fq * 8;
#pragma unroll
      for (int s = 0; s < KS; ++s) wf[j][s] = *(const h16x8*)(wrow + s * 32);
      bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias && n0 + j * 16 + 4 * fq + 3 < p.N) bias4[j] = *(const f32x4*)(p.bias + n0 + j * 16 + 4 * fq);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < WCT; ++j) {
#pragma unroll
      for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[j][s]));
      asm volatile("" : "+v"(bias4[j]));
    }
  }
  const bool cols_live = n0 < p.N;  // (last column block of an N that is not a multiple of BNW: whole waves idle)

  // ---- DMA source offsets of this lane (tile independent): instruction i covers LDS bytes [1024 i, 1024 i + 1024) ----
  int dsrc[PD];
#pragma unroll
  for (int t = 0; t < PD; ++t) {
    const int q = (t * 8 + wave) * 64 + lane;  // slot index inside the tile image
    const int row = q / S, cph = q % S;
    const int clog = S % 16 == 0 ? ((cph & ~15) | ((cph & 15) ^ swz(row))) : ((cph & ~3) | ((cph & 3) ^ swz(row)));
    dsrc[t] = (row * p.lda + clog * 8) * 2;
  }
  auto stage = [&](int it) {  // it: index into this workgroup's row-tile sequence
    const int mt = slot + it * p.per;
    char* st = smem + (it % NSTG) * TILE;
    const char* Ab = (const char*)p.A + (int64_t)mt * TMR * p.lda * 2;
    const bool live = mt < p.ntm;
#pragma unroll
    for (int t = 0; t < PD; ++t) {
      if (live && t * 8 + wave < NI)  // (wave-uniform: scalar branch, exactly one DMA instruction per t)
        wreg_glds16(Ab + dsrc[t], st + (t * 8 + wave) * 1024);
      else
        wreg_glds16(g_wreg_zero16, smem + DUMP);  // keeps every wave's DMA count per tile at PD
    }
  };

  // fragment read offsets (tile independent): row 16 i + fr of this wave's rows, slot 4 s + fq
  // (every wave row block starts at a multiple of 16 rows, so row & 15 == fr for all of a lane's rows)
  int a_rd[WRT];
#pragma unroll
  for (int i = 0; i < WRT; ++i) a_rd[i] = (wr * (16 * WRT) + i * 16 + fr) * ROWB;
  int s_rd[4];  // byte offset of slot (4 k + fq) after the swizzle, k = s & 3 (S = 36: the same for every k up to + 64 k)
#pragma unroll
  for (int k = 0; k < 4; ++k) s_rd[k] = S % 16 == 0 ? (((4 * k + fq) ^ fr) << 4) : ((4 * k + (fq ^ swz(fr))) << 4);
  char* const strip = smem + STRIP0 + wave * STRIP;

  // Row-group scale (rf_gemm_desc.rs: the tied attention's position weights folded into q): a per-lane gather of RS fp32 values
  // per row tile.  As plain loads hipcc waited for them with s_waitcnt vmcnt(0) -- five times per tile -- which drained the DMA
  // ring and every posted store (the q|k|v projection of the tied attention ran at 2.4 TB/s against 3.2-3.8 for its siblings).
  // They are issued from inline asm ONE TILE AHEAD, at the top of the previous iteration IN FRONT of its DMAs: vmcnt retires in
  // order, so a wait for a load also waits for every older operation -- with the loads issued behind an iteration's DMAs (first
  // form) the wait for them forced the DMAs of the ring's newest slot to land one iteration after their issue (162 us per launch);
  // in front of them, the two newest DMA sets and the previous tile's stores stay in flight (counted wait below).
  constexpr int RS = WRT * WCT;
  // (wave-uniform; instances without the split-C epilogue never see a scale: rf_gemm_wreg_try declines rs there, so the two
  // register sets cost them nothing)
  const bool has_rs = CS && p.rs != nullptr && n0 < p.rs_ncols && cols_live;
  float rs_a[WRT][WCT], rs_b[WRT][WCT];  // two register sets: tile it's scales are used while tile it + 1's are in flight
  auto rs_issue = [&](int it_, float (&rsv)[WRT][WCT]) {
    const int mt_ = slot + it_ * p.per;
    const int mtc = mt_ < p.ntm ? mt_ : p.ntm - 1;   // (past the end: a valid address, the values are never used)
#pragma unroll
    for (int i = 0; i < WRT; ++i) {
      const int m = mtc * TMR + wr * (16 * WRT) + i * 16 + fr;
      const int qb = m / p.rs_rpb;
      const float* rsm = p.rs + (int64_t)qb * p.rs_bstride + (m - qb * p.rs_rpb);
#pragma unroll
      for (int j = 0; j < WCT; ++j) {
        const int n = n0 + j * 16;
        const float* src = rsm + (int64_t)((n < p.rs_ncols ? n : 0) / p.rs_cg) * p.rs_rpb;
        asm volatile("global_load_dword %0, %1, off" : "=v"(rsv[i][j]) : "v"(src) : "memory");
      }
    }
  };

  if (has_rs) rs_issue(0, rs_a);  // (the oldest operations of the wave: in front of the prologue's DMAs)
#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) stage(s);
  // one row tile; returns false behind the last one.  rs_cur / rs_nxt: the scale registers of this tile / the next one (the loop
  // below alternates the two sets: register arrays need static names)
  auto row_tile = [&](int it, float (&rs_cur)[WRT][WCT], float (&rs_nxt)[WRT][WCT]) -> bool {
    const int mt = slot + it * p.per;
    if (mt >= p.ntm) return false;
    // tile `it` landed once only the younger operations of this wave are outstanding: per later tile PD DMAs, per earlier
    // tile (its stores were issued after the DMA being waited for) PS stores -- fewer of each at the start
    // (a wave whose columns lie beyond N issues no stores at all: its count holds DMAs only)
    if (!cols_live)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * (NSTG - 2)) : "memory");
    else if (has_rs) {
      // issue order per iteration: R(it + 1), D(it + NSTG - 1), ..., S(it).  D(it) came in iteration it - NSTG + 1 behind that
      // iteration's scale loads: younger are its stores and the (R, D, S) of the NSTG - 2 iterations since
      static_assert((NSTG - 2) * (RS + PD + PS) + PS <= 63, "vmcnt range");
      if (it >= NSTG - 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (RS + PD + PS) + PS) : "memory");
      else if (it == 0)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * (NSTG - 2)) : "memory");
      else  // (exact for it = 1, conservative up to NSTG - 2: prologue DMAs (NSTG - 2 - it) PD + it (RS + PD + PS))
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 3) * PD + RS + PD + PS) : "memory");
    } else if (it >= NSTG - 1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * (NSTG - 2) + PS * (NSTG - 1)) : "memory");
    else if (it == 0)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * (NSTG - 2)) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * (NSTG - 2) + PS) : "memory");
    __builtin_amdgcn_s_barrier();
    if (has_rs) rs_issue(it + 1, rs_nxt);  // (in front of this iteration's DMAs)
    stage(it + NSTG - 1);  // into the buffer of tile it-1: every wave has consumed its fragments (they fed MFMAs already issued)
    const char* st = smem + (it % NSTG) * TILE;

    if (!cols_live) return true;
    f32x4 acc[WRT][WCT];
#pragma unroll
    for (int i = 0; i < WRT; ++i)
#pragma unroll
      for (int j = 0; j < WCT; ++j) acc[i][j] = bias4[j];
    {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        h16x8 af[WRT];
#pragma unroll
        for (int i = 0; i < WRT; ++i) af[i] = *(const h16x8*)(st + a_rd[i] + (s >> 2) * 256 + s_rd[s & 3]);
#pragma unroll
        for (int i = 0; i < WRT; ++i)
#pragma unroll
          for (int j = 0; j < WCT; ++j)
            // weight tile as MFMA-A, activation tile as MFMA-B: lane holds C[m = 16 i + fr][n = 16 j + 4 fq .. +3]
            acc[i][j] = rf_mfma16(wf[j][s], af[i], acc[i][j], 0, 0, 0);
      }
    }
    // MFMA results are read by VALU instructions from here on (row scale, clamp, pack): explicit wait states in the readers' path
    // (volatile asm statements keep their order and every accumulator passes through one: no reader can move above the nops)
    asm volatile("s_nop 7\n\ts_nop 1" : "+v"(acc[0][0]));
#pragma unroll
    for (int i = 0; i < WRT; ++i)
#pragma unroll
      for (int j = 0; j < WCT; ++j)
        if (i + j > 0) asm volatile("" : "+v"(acc[i][j]));
    if (has_rs) {  // (wave-uniform; rs_cg % 16 == 0: a 16-column tile lies inside one column group)
      // this tile's scales (issued at the top of the previous iteration): younger are that iteration's DMAs and stores and this
      // iteration's scale loads and DMAs (the first tile's set is older than the whole prologue: the same count is safe)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PD + PS + RS) : "memory");
#pragma unroll
      for (int i = 0; i < WRT; ++i)
#pragma unroll
        for (int j = 0; j < WCT; ++j) {
          asm volatile("" : "+v"(rs_cur[i][j]));
          if (n0 + j * 16 < p.rs_ncols) acc[i][j] *= rs_cur[i][j] * p.rs_alpha;
        }
    }
    // ---- epilogue: wave-private strip (in-order LDS per wave: no barrier), 16-byte row-contiguous stores -----------
    // (MFMA wait states: see behind the MFMA loop.)  The accumulators are read by VALU instructions from there on.  gfx950 does not interlock a VALU read of an MFMA destination
    // (8 wait states behind a 16x16x32 MFMA are software's job) and hipcc's hazard recognizer pads straight-line code only: on the
    // path that skips the row-scale block it left 6 (tools/isa_hazard_scan.py, the class of csrc/favor.hip's fv_mfma_done).  The
    // nops sit in the readers' block, so every path passes them.
    const float lo = p.relu ? 0.f : -INFINITY;
#pragma unroll
    for (int i = 0; i < WRT; ++i)
#pragma unroll
      for (int j = 0; j < WCT; ++j) {
        uint2 o;
        o.x = wreg_pack2(fmaxf(acc[i][j][0], lo), fmaxf(acc[i][j][1], lo));
        o.y = wreg_pack2(fmaxf(acc[i][j][2], lo), fmaxf(acc[i][j][3], lo));
        *(uint2*)(strip + (i * 16 + fr) * PITCH + (j * 16 + 4 * fq) * 2) = o;
      }
    asm volatile("" ::: "memory");
    const int m_w = mt * TMR + wr * (16 * WRT);
#pragma unroll
    for (int t = 0; t < PS; ++t) {
      const int idx = lane + 64 * t;
      const int r = idx / CPR, c = idx % CPR;
      const f32x4 v = *(const f32x4*)(strip + r * PITCH + c * 16);
      const int m = m_w + r, n = n0 + c * 8;
      int64_t off;
      if constexpr (CS) {
        const int qr = p.c_rc > 0 ? (p.c_rsh >= 0 ? m >> p.c_rsh : m / p.c_rc) : 0;
        const int qc = p.c_cc > 0 ? (p.c_csh >= 0 ? n >> p.c_csh : n / p.c_cc) : 0;
        off = (p.c_rc > 0 ? (int64_t)qr * p.c_ro + (int64_t)(m - qr * p.c_rc) * p.ldc : (int64_t)m * p.ldc) +
              (p.c_cc > 0 ? (int64_t)qc * p.c_co + (n - qc * p.c_cc) : n);
      } else {
        off = (int64_t)m * p.ldc + n;
      }
      f32x4* dst = (f32x4*)(p.C + off);
      // (all 64 lanes store: N % (16 WCT) == 0 is checked on the host, so a live wave's columns are all inside the matrix)
      if (p.nt_store)
        __builtin_nontemporal_store(v, dst);
      else
        *dst = v;
    }
    return true;
  };
  for (int it = 0;; it += 2) {
    if (!row_tile(it, rs_a, rs_b)) break;
    if (!row_tile(it + 1, rs_b, rs_a)) break;
  }
}

template <int KS, int TMR, int WR, int WCT, int NSTG, bool CS>
static int launch_wreg(WregP& p, hipStream_t s) {
  constexpr int WC = 8 / WR, BNW = WC * WCT * 16, WRT = TMR / WR / 16;
  constexpr int TILE = TMR * KS * 64;
  constexpr int LDS = NSTG * TILE + 1024 + 8 * (16 * WRT * (WCT * 32 + 16));
  static_assert(LDS <= 160 * 1024, "LDS budget");
  const int ncu = rf_num_cus();
  if (ncu <= 0) return RF_EINVAL;
  p.nblocks = (p.N + BNW - 1) / BNW;
  p.ntm = p.M / TMR;
  p.per = ncu / p.nblocks;
  if (p.per < 1) p.per = 1;
  if (p.per > p.ntm) p.per = p.ntm;
  const int grid = p.nblocks * p.per;
  if (const int e = rf_enable_big_lds<gemm_wreg_kernel<KS, TMR, WR, WCT, NSTG, CS>>()) return e;
  hipLaunchKernelGGL((gemm_wreg_kernel<KS, TMR, WR, WCT, NSTG, CS>), dim3((unsigned)grid), dim3(512), LDS, s, p);
  return rf_launch_status();
}

// Returns 1 and launches when the descriptor fits (bf16 in / bf16 out, plain operands, K = 288 or 384, long M), 0 otherwise.
int rf_gemm_wreg_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream) {
  *rc = 0;
  static const bool off = rf_env_flag("RF_NO_WREG_GEMM");
  if (off) return 0;
  if (d.ab_dtype != RF_H16 || d.c_dtype != RF_H16 || d.a_mode != RF_AMODE_PLAIN || batch != 1) return 0;
  if (d.a_rc > 0 || d.b_rc > 0 || d.kc != d.K || d.residual || d.ln_out || d.alpha != 1.0f) return 0;
  if (d.K != 288 && d.K != 384) return 0;
  if (d.M % 64 != 0 || d.M < 16384 || d.N % 128 != 0 || d.N < 256) return 0;
  static const int force = getenv("RF_WREG_VARIANT") ? atoi(getenv("RF_WREG_VARIANT")) : 0;  // A/B: 1 = 256-wide, 2 = 128-wide, 3 = 384-wide
  if (d.bias_mode == RF_BIAS_ROW || (d.act != RF_ACT_NONE && d.act != RF_ACT_RELU)) return 0;
  if (d.a_ri % 8 || d.b_ri % 8 || d.c_ri % 8 || ((uintptr_t)d.A % 16) || ((uintptr_t)d.B % 16) || ((uintptr_t)d.C % 16)) return 0;
  if (d.bias_mode == RF_BIAS_COL && ((uintptr_t)d.bias % 16)) return 0;
  if ((int64_t)64 * d.a_ri * 2 >= (1ll << 31)) return 0;
  const bool cs = d.c_rc > 0 || d.c_cc > 0;
  if (d.rs && !cs) return 0;  // (the row-group scale lives in the split-C instances: the tied attention's head-major projection)
  if (cs && ((d.c_cc > 0 && (d.c_cc % 8 || d.c_co % 8)) || (d.c_rc > 0 && d.c_ro % 8))) return 0;
  WregP p;
  p.A = (const h16_t*)d.A; p.B = (const h16_t*)d.B; p.C = (h16_t*)d.C;
  p.bias = d.bias_mode == RF_BIAS_COL ? d.bias : nullptr;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.lda = (int)d.a_ri; p.ldb = (int)d.b_ri; p.ldc = (int)d.c_ri;
  p.relu = d.act == RF_ACT_RELU;
  static const bool no_nt = rf_env_flag("RF_NO_NT_STORE");
  p.nt_store = ((int64_t)d.M * d.N * 2 > (64ll << 20)) && !no_nt;
  p.c_rc = d.c_rc; p.c_cc = d.c_cc; p.c_ro = d.c_ro; p.c_co = d.c_co;
  p.c_rsh = (d.c_rc > 0 && (d.c_rc & (d.c_rc - 1)) == 0) ? __builtin_ctz(d.c_rc) : -1;
  p.c_csh = (d.c_cc > 0 && (d.c_cc & (d.c_cc - 1)) == 0) ? __builtin_ctz(d.c_cc) : -1;
  p.rs = d.rs; p.rs_bstride = d.rs_bstride; p.rs_rpb = d.rs_rpb; p.rs_cg = d.rs_cg; p.rs_ncols = d.rs_ncols; p.rs_alpha = d.rs_alpha;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = d.N % 256 == 0 && force != 2;
  if (d.N % 384 == 0 && (force == 0 || force == 3)) {
    // 8 waves x 48 columns: three MFMAs per activation fragment read, N / 384 column blocks (1536 -> 4 x 64 workgroups)
    if (d.K == 288) *rc = cs ? launch_wreg<9, 32, 1, 3, 4, true>(p, s) : launch_wreg<9, 32, 1, 3, 4, false>(p, s);
    else *rc = cs ? launch_wreg<12, 32, 1, 3, 4, true>(p, s) : launch_wreg<12, 32, 1, 3, 4, false>(p, s);
    return 1;
  }
  if (d.K == 288) {
    if (wide) *rc = cs ? launch_wreg<9, 64, 2, 4, 3, true>(p, s) : launch_wreg<9, 64, 2, 4, 3, false>(p, s);
    else *rc = cs ? launch_wreg<9, 64, 4, 4, 3, true>(p, s) : launch_wreg<9, 64, 4, 4, 3, false>(p, s);
  } else {
    if (wide) *rc = cs ? launch_wreg<12, 32, 1, 2, 4, true>(p, s) : launch_wreg<12, 32, 1, 2, 4, false>(p, s);
    else *rc = cs ? launch_wreg<12, 32, 2, 2, 4, true>(p, s) : launch_wreg<12, 32, 2, 2, 4, false>(p, s);
  }
  return 1;
}
