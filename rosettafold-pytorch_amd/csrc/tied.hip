// Tied MSA-row attention core (SoftTiedAttentionOverResidues, rf.py:241-267) for gfx950 (MI355X), three kernels:
//
//   poswise_mfma_kernel   w[b,h,n,l]   = softmax_n( scale * xn[b,n,l,:] . u[b,l,h,:] )             (rf.py:205-217, collapsed)
//   tied_logits_kernel    att[b,h,i,:] = softmax_j( sum_{n,d} (w[b,h,n,i] qs) q[b,n,h,i,d] k[b,n,h,j,d] )   (rf.py:252-255)
//   tied_av_kernel        out[b,n,i,h,:] = sum_j att[b,h,i,j] v[b,n,h,j,:]                          (rf.py:257-258)
//
// Layout: the q|k|v projection GEMM writes its output HEAD-MAJOR, [B, N, G, L, 32] with G = 3 H groups of 32 columns
// (csrc/gemm_fast.hip, split-C epilogue), so the operand tile of one contraction step -- the 32-wide head slice of all L
// residues of one MSA row -- is ONE contiguous L x 64-byte block: every DMA instruction moves whole 128-byte lines (the
// round-1 kernel gathered 64-byte slices 2.3 KB apart and ran at the gather rate).  The kernels take element strides
// (b, n, h, l), so any layout with a contiguous head dimension is accepted; speed, not correctness, depends on it.
//
// The position weights are applied INSIDE the logits kernel (the wave scales its 16 query rows of the step in registers
// by w[n, row]), which removes the read-modify-write pass over q; the weights themselves come from the collapsed form
// u[b,l,h,:] = W_k,h^T to_q(x_0)[b,l,h,:]: the to_k projection over all N rows (a third of the projection GEMM) and its
// bias (constant in n, so it cancels in the softmax over n) disappear.
#include <type_traits>

#include "common.h"

static __device__ __attribute__((aligned(16))) unsigned int g_tied_zero16[4];

union TFrag {
  h16x8 v;
  unsigned u[4];
  uint2 h[2];
};
typedef __attribute__((ext_vector_type(4))) short ts16x4;

__device__ __forceinline__ unsigned tpack2(float a, float b) { return rf_pack2_h16(a, b); }

__device__ __forceinline__ void tied_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// logits + softmax
// ------------------------------------------------------------------------------------------------------------------
struct TiedP {
  const h16_t* q;
  const h16_t* k;
  int64_t b_stride, n_stride, h_stride, l_stride;  // elements (q and k share them); head slice = 32 contiguous elements
  const float* w;                                   // position weights [.., l] or null (q used as it is)
  int64_t w_b, w_h, w_n;                            // element strides of w (l contiguous)
  float qscale;                                     // multiplies w (d_head^-0.5, rf.py:252)
  h16_t* att;                                      // [B, H, L, L]
  int B, H, N;
  int dbg;  // timing experiments (RF_TIED_DBG; results are WRONG when set): 1 no DMA after the prologue, 2 fragments read once, 4 no MFMA
};

template <int L, bool SCALE>
__global__ __launch_bounds__(256) void tied_logits_kernel(const TiedP p) {
  constexpr int JT = L / 16;                 // key tiles of 16 columns
  constexpr int NSTG = L >= 256 ? 6 : 8;     // ring stages
  constexpr int Q_BYTES = 64 * 64, K_BYTES = L * 64, STAGE = Q_BYTES + K_BYTES;
  constexpr int QI = 4, KI = L / 16;         // DMA instructions (16 rows x 64 B) per stage
  constexpr int PW = (QI + KI + 3) / 4;      // per wave, uniform (padded with dummies into DUMP)
  constexpr int DUMP = NSTG * STAGE;
  constexpr int W_OFF = DUMP + 1024;         // [N][64] fp32 position weights of this workgroup's query rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // XCD-aware order: the L/64 query tiles of one (b, h) (same K slab) run on the same XCD
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  constexpr int IT = L / 64;
  const int it = lid % IT, h = (lid / IT) % p.H, b = lid / (IT * p.H);
  const h16_t* qb = p.q + (int64_t)b * p.b_stride + (int64_t)h * p.h_stride + (int64_t)(it * 64) * p.l_stride;
  const h16_t* kb = p.k + (int64_t)b * p.b_stride + (int64_t)h * p.h_stride;

  // DMA: an instruction covers 16 rows x 4 chunks (64-byte head slices); lane-linear LDS image with the bank swizzle
  // chunk ^ g((row >> 2) & 3), g = {0, 2, 3, 1}, on the source chunk and on the fragment reads
  const int lrow = lane >> 2;
  const int c_log = (lane & 3) ^ ((0x78 >> (((lrow >> 2) & 3) * 2)) & 3);
  const int64_t lane_off = (int64_t)lrow * p.l_stride + c_log * 8;
  auto stage = [&](int n) {
    char* st = smem + (n % NSTG) * STAGE;
    const bool live = n < p.N && !((RF_DBG(p.dbg) & 1) && n >= NSTG - 1);
#pragma unroll
    for (int t = 0; t < PW; ++t) {
      const int instr = t * 4 + wave;
      if (live && instr < QI)
        tied_glds16(qb + (int64_t)n * p.n_stride + (int64_t)(instr * 16) * p.l_stride + lane_off, st + instr * 1024);
      else if (live && instr < QI + KI)
        tied_glds16(kb + (int64_t)n * p.n_stride + (int64_t)((instr - QI) * 16) * p.l_stride + lane_off, st + instr * 1024);
      else
        tied_glds16(g_tied_zero16, smem + DUMP);  // keeps every wave's DMA count per step at PW
    }
  };

  if constexpr (SCALE) {
    // w tile: rows it*64 .. +63 of every MSA row n (256 contiguous bytes each), by DMA: one instruction = 4 MSA rows.
    // These are the oldest operations of every wave, so the counted wait of the first step covers them.
    const float* wb = p.w + (int64_t)b * p.w_b + (int64_t)h * p.w_h + it * 64 + (lane & 15) * 4;
    for (int i4 = wave; i4 * 4 < p.N; i4 += 4)
      tied_glds16(wb + (int64_t)(i4 * 4 + (lane >> 4)) * p.w_n, smem + W_OFF + i4 * 1024);
  }

  f32x4 acc[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int gq = (0x78 >> (((fr >> 2) & 3) * 2)) & 3;
  const int q_rd = (wave * 16 + fr) * 64 + ((fq ^ gq) * 16);
  const int k_rd = Q_BYTES + fr * 64 + ((fq ^ gq) * 16);
  const float* wrow = (const float*)(smem + W_OFF) + wave * 16 + fr;

#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) stage(s);
  for (int n = 0; n < p.N; ++n) {
    // step n landed once at most the NSTG-2 younger steps are outstanding (this wave's share; the barrier covers the rest)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSTG - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(n + NSTG - 1);  // into the buffer of step n-1, whose fragments every wave has consumed
    const char* st = smem + (n % NSTG) * STAGE;
    TFrag qf;
    h16x8 kf[JT];
    if (!(RF_DBG(p.dbg) & 2) || n == 0) {
      qf.v = *(const h16x8*)(st + q_rd);
#pragma unroll
      for (int j = 0; j < JT; ++j) kf[j] = *(const h16x8*)(st + k_rd + j * 1024);
    }
    if constexpr (SCALE) {
      const float ws = wrow[n * 64] * p.qscale;  // w[b,h,n, row of this lane] * d_head^-0.5: the same rounding point as q*w (rf.py:252)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        qf.u[e] = tpack2(rf_h16_lo(qf.u[e]) * ws, rf_h16_hi(qf.u[e]) * ws);
    }
    if (!(RF_DBG(p.dbg) & 4)) {
#pragma unroll
      for (int j = 0; j < JT; ++j)
        // key tile as MFMA-A, query tile as MFMA-B: lane holds logits[i = fr][j = 16*tile + 4*fq .. +3]
        acc[j] = rf_mfma16(kf[j], qf.v, acc[j], 0, 0, 0);
    }
  }

  // ---- row softmax: a row lives in the four lanes {fr, fr+16, fr+32, fr+48} ------------------------------------
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < JT; ++j) mx = fmaxf(mx, fmaxf(fmaxf(acc[j][0], acc[j][1]), fmaxf(acc[j][2], acc[j][3])));
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sm = 0.f;
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[j][e] = __expf(acc[j][e] - mx);
      sm += acc[j][e];
    }
  sm += __shfl_xor(sm, 16, 64);
  sm += __shfl_xor(sm, 32, 64);
  const float inv = 1.f / sm;

  // ---- probabilities -> wave-private strip (16 rows x L bf16) -> whole rows of att -------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // ring drained and every wave done with its last fragments
  constexpr int PITCH = L * 2 + 16;
  char* strip = smem + wave * (16 * PITCH);
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    uint2 w;
    w.x = tpack2(acc[j][0] * inv, acc[j][1] * inv);
    w.y = tpack2(acc[j][2] * inv, acc[j][3] * inv);
    *(uint2*)(strip + fr * PITCH + (j * 16 + 4 * fq) * 2) = w;
  }
  asm volatile("" ::: "memory");
  constexpr int CPR = L * 2 / 16, NCH = 16 * CPR, NIT = NCH / 64;
  h16_t* arow = p.att + (((int64_t)b * p.H + h) * L + it * 64 + wave * 16) * L;
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    const int idx = lane + 64 * t;
    const int r = idx / CPR, c = idx % CPR;
    const f32x4 v = *(const f32x4*)(strip + r * PITCH + c * 16);
    *(f32x4*)(arow + (int64_t)r * L + c * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Contraction-split form of the logits (L = 256): a workgroup owns 128 query rows x all 256 keys of one (b, h) over a
// RANGE of the MSA rows n and writes fp32 partial logits; tied_split_softmax_kernel adds the partials and takes the row
// softmax.  Why: the 64-row kernel above is bound by its L2 -> LDS stream (20 KB per contraction step for 64 x 256
// outputs: 500 MB per launch at ~7 TB/s; RF_TIED_DBG ablation: 70 of its 90 us remain with the fragment reads and the
// MFMAs switched off).  Here a step moves 24 KB for 128 x 256 outputs (0.6x the bytes per FLOP), eight waves of 64 x 64
// read 8 fragments for 16 MFMAs instead of 17, and the split over n keeps >= 192 workgroups in flight.
// ------------------------------------------------------------------------------------------------------------------
#define TIED_SPLIT_NSTG 5  // ring stages of 24 KB: the stream is latency-bound (~2 us per piece under load), bytes in flight = rate
struct TiedSplitP {
  const h16_t* q;
  const h16_t* k;
  int64_t b_stride, n_stride, h_stride, l_stride;
  const float* w;
  int64_t w_b, w_h, w_n;
  float qscale;
  float* part;          // fp32 [nsplit][B][H][L][L]
  int64_t split_stride; // elements between the partial tensors of consecutive splits
  int B, H, N, nsplit, nper;  // nper = MSA rows per split (N = nsplit * nper)
  int L, nrb, nkb;            // L = 128 nrb = 256 nkb: query blocks of 128 rows x key blocks of 256 columns
  int dbg;                    // RF_TIED_DBG (results WRONG when set): 2 fragments read once, 4 no MFMA, 8 no partial stores
};

template <bool SCALE>
__global__ __launch_bounds__(512) void tied_logits_split_kernel(const TiedSplitP p) {
  constexpr int KB = 256, RB = 128, NSTG = TIED_SPLIT_NSTG;  // a workgroup: 128 query rows x 256 keys (any L = 256 nkb)
  constexpr int Q_BYTES = RB * 64, K_BYTES = KB * 64, STAGE = Q_BYTES + K_BYTES;
  constexpr int QI = RB / 16, PW = 3;        // (QI + KB / 16) DMA pieces of 1 KB per stage = 3 per wave, no padding
  static_assert(QI + KB / 16 == 8 * PW, "pieces per stage");
  const int L = p.L;
  constexpr int DUMP = NSTG * STAGE;
  constexpr int W_OFF = DUMP + 1024;         // [nper][128] fp32 position weights of this workgroup's query rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;   // 64-query half, 64-key quarter
  const int fr = lane & 15, fq = lane >> 4;

  // XCD-aware order: the nrb * nkb * nsplit workgroups of one (b, h) run on the same XCD (they share q / k slabs through its L2)
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tiles = p.nrb * p.nkb;
  const int rb = lid % p.nrb, kbk = (lid / p.nrb) % p.nkb, sp = (lid / tiles) % p.nsplit;
  const int bh = lid / (tiles * p.nsplit), h = bh % p.H, b = bh / p.H;
  const int n0 = sp * p.nper;
  const h16_t* qb = p.q + (int64_t)b * p.b_stride + (int64_t)h * p.h_stride + (int64_t)(rb * RB) * p.l_stride;
  const h16_t* kb = p.k + (int64_t)b * p.b_stride + (int64_t)h * p.h_stride + (int64_t)(kbk * KB) * p.l_stride;

  const int lrow = lane >> 2;
  const int c_log = (lane & 3) ^ ((0x78 >> (((lrow >> 2) & 3) * 2)) & 3);
  const int64_t lane_off = (int64_t)lrow * p.l_stride + c_log * 8;
  auto stage = [&](int t_) {  // t_: step index inside this split
    char* st = smem + (t_ % NSTG) * STAGE;
    const bool live = t_ < p.nper;
    const int64_t noff = (int64_t)(n0 + t_) * p.n_stride;
#pragma unroll
    for (int t = 0; t < PW; ++t) {
      const int instr = t * 8 + wave;
      if (live && instr < QI)
        tied_glds16(qb + noff + (int64_t)(instr * 16) * p.l_stride + lane_off, st + instr * 1024);
      else if (live)
        tied_glds16(kb + noff + (int64_t)((instr - QI) * 16) * p.l_stride + lane_off, st + instr * 1024);
      else
        tied_glds16(g_tied_zero16, smem + DUMP);  // keeps every wave's DMA count per step at PW
    }
  };
  if constexpr (SCALE) {
    // w tile: rows rb*128 .. +127 of the MSA rows of this split (512 contiguous bytes each): one instruction = 2 MSA rows.
    // The oldest operations of every wave: the counted wait of the first step covers them.
    const float* wb = p.w + (int64_t)b * p.w_b + (int64_t)h * p.w_h + rb * RB + (lane & 31) * 4;
    for (int i2 = wave; i2 * 2 < p.nper; i2 += 8)
      tied_glds16(wb + (int64_t)(n0 + i2 * 2 + (lane >> 5)) * p.w_n, smem + W_OFF + i2 * 1024);
  }

  f32x4 acc[4][4];  // [key tile][query tile]
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) acc[kt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int gq = (0x78 >> (((fr >> 2) & 3) * 2)) & 3;
  const int q_rd = (wr * 64 + fr) * 64 + ((fq ^ gq) * 16);
  const int k_rd = Q_BYTES + (wc * 64 + fr) * 64 + ((fq ^ gq) * 16);
  const float* wrow = (const float*)(smem + W_OFF) + wr * 64 + fr;

#pragma unroll
  for (int s_ = 0; s_ < NSTG - 1; ++s_) stage(s_);
  for (int t_ = 0; t_ < p.nper; ++t_) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSTG - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(t_ + NSTG - 1);  // into the buffer of step t_ - 1, whose fragments every wave has consumed
    const char* st = smem + (t_ % NSTG) * STAGE;
    TFrag qf[4];
    h16x8 kf[4];
    if (!(RF_DBG(p.dbg) & 2) || t_ == 0) {
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) qf[qt].v = *(const h16x8*)(st + q_rd + qt * 1024);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) kf[kt] = *(const h16x8*)(st + k_rd + kt * 1024);
    }
    if constexpr (SCALE) {
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const float ws = wrow[t_ * RB + qt * 16] * p.qscale;  // w[b,h,n, query row of this lane] * d_head^-0.5 (rf.py:252)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          qf[qt].u[e] = tpack2(rf_h16_lo(qf[qt].u[e]) * ws, rf_h16_hi(qf[qt].u[e]) * ws);
      }
    }
    if (!(RF_DBG(p.dbg) & 4)) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
          // key tile as MFMA-A, query tile as MFMA-B: lane holds logits[i = 16 qt + fr][j = 16 kt + 4 fq .. +3]
          acc[kt][qt] = rf_mfma16(kf[kt], qf[qt].v, acc[kt][qt], 0, 0, 0);
    }
  }
  if (RF_DBG(p.dbg) & 8) return;
  // partial logits: 16-byte pieces, the four lanes of a row group cover 64 contiguous bytes, a wave's four key tiles 256
  float* pr = p.part + (int64_t)sp * p.split_stride + (((int64_t)b * p.H + h) * L + rb * RB + wr * 64 + fr) * L + kbk * KB + wc * 64 + 4 * fq;
#pragma unroll
  for (int qt = 0; qt < 4; ++qt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) *(f32x4*)(pr + (int64_t)(qt * 16) * L + kt * 16) = acc[kt][qt];
}

// att[row, :] = softmax(sum_s part[s][row, :]) for L = 256 NC: one wave per row, 4 NC columns per lane
template <int NC>
__global__ __launch_bounds__(256) void tied_split_softmax_kernel(const float* part, int64_t split_stride, int nsplit,
                                                                 h16_t* att, int64_t rows) {
  constexpr int L = 256 * NC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) v[c] = *(const f32x4*)(part + row * L + c * 256 + lane * 4);
  for (int s_ = 1; s_ < nsplit; ++s_)
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] += *(const f32x4*)(part + (int64_t)s_ * split_stride + row * L + c * 256 + lane * 4);
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < NC; ++c) mx = fmaxf(mx, fmaxf(fmaxf(v[c][0], v[c][1]), fmaxf(v[c][2], v[c][3])));
  mx = wave_max(mx);
  float sm = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[c][e] = __expf(v[c][e] - mx);
      sm += v[c][e];
    }
  sm = wave_sum(sm);
  const float inv = 1.f / sm;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    uint2 o;
    o.x = tpack2(v[c][0] * inv, v[c][1] * inv);
    o.y = tpack2(v[c][2] * inv, v[c][3] * inv);
    *(uint2*)(att + row * L + c * 256 + lane * 4) = o;
  }
}

__global__ __launch_bounds__(256) void tied_att_sym_kernel(const h16_t* att, float* sym, int64_t sym_ld, int B, int H, int L) {
  // sym[b,i,j,h] = 0.5*(att[b,h,i,j] + att[b,h,j,i])
  const int64_t n = (int64_t)B * L * L * H;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int h = e % H;
    const int64_t t = e / H;
    const int j = t % L, i = (t / L) % L;
    const int64_t b = t / ((int64_t)L * L);
    const int64_t o = (b * H + h) * L;
    sym[t * sym_ld + h] = 0.5f * (h2f(att[(o + i) * L + j]) + h2f(att[(o + j) * L + i]));
  }
}

static int tied_dbg() {  // < 0: the switch is set but this build has no ablation code (RF_EINVAL)
  static int v = 0;
  static const int rc = rf_dbg_env("RF_TIED_DBG", &v);
  return rc ? rc : v;
}

template <int L, bool SCALE>
static int launch_tied(const TiedP& p, hipStream_t s) {
  constexpr int NSTG = L >= 256 ? 6 : 8;
  const size_t lds = (size_t)NSTG * (64 * 64 + L * 64) + 1024 + (SCALE ? (size_t)p.N * 256 : 0);
  if (lds > 160 * 1024) return RF_EINVAL;
  auto k = tied_logits_kernel<L, SCALE>;
  if (const int e = rf_enable_big_lds<tied_logits_kernel<L, SCALE>>()) return e;
  hipLaunchKernelGGL(k, dim3((unsigned)(p.B * p.H * (L / 64))), dim3(256), lds, s, p);
  return rf_launch_status();
}

// contraction-split path: L = 256 .. 1024 in steps of 256, N splits into nsplit ranges of <= 64 rows, workspace of
// nsplit * B*H*L*L floats
static int tied_logits_split(const TiedP& p0, int L, float* ws, int64_t ws_elems, hipStream_t s) {
  static const bool off = rf_env_flag("RF_NO_TIED_SPLIT");
  if (off || !ws || L % 256 || L > 1024 || p0.N % 2) return 1;  // 1 = not applicable: the caller falls back to the one-pass kernel
  const int nrb = L / 128, nkb = L / 256;
  int nsplit = L == 256 ? 2 : 1;  // (at L = 256 two splits keep >= 192 workgroups in flight; longer rows have enough tiles)
  while (p0.N / nsplit > 64 && p0.N % (nsplit * 2) == 0) nsplit *= 2;
  const int nper = p0.N / nsplit;
  const int64_t one = (int64_t)p0.B * p0.H * L * L;
  if (nper > 64 || nper < 4 || (nper & 1) || ws_elems < one * nsplit || ((uintptr_t)ws % 16)) return 1;
  TiedSplitP p;
  p.q = p0.q; p.k = p0.k;
  p.b_stride = p0.b_stride; p.n_stride = p0.n_stride; p.h_stride = p0.h_stride; p.l_stride = p0.l_stride;
  p.w = p0.w; p.w_b = p0.w_b; p.w_h = p0.w_h; p.w_n = p0.w_n; p.qscale = p0.qscale;
  p.part = ws; p.split_stride = one;
  p.B = p0.B; p.H = p0.H; p.N = p0.N; p.nsplit = nsplit; p.nper = nper;
  p.L = L; p.nrb = nrb; p.nkb = nkb;
  p.dbg = p0.dbg;
  const size_t lds = (size_t)TIED_SPLIT_NSTG * (128 * 64 + 256 * 64) + 1024 + (p.w ? (size_t)nper * 512 : 0);
  const int64_t grid64 = (int64_t)p.B * p.H * nrb * nkb * nsplit;
  if (grid64 > 0x7fffffffLL) return RF_EINVAL;
  const unsigned grid = (unsigned)grid64;
  if (p.w) {
    if (const int e = rf_enable_big_lds<tied_logits_split_kernel<true>>()) return e;
    hipLaunchKernelGGL(tied_logits_split_kernel<true>, dim3(grid), dim3(512), lds, s, p);
  } else {
    if (const int e = rf_enable_big_lds<tied_logits_split_kernel<false>>()) return e;
    hipLaunchKernelGGL(tied_logits_split_kernel<false>, dim3(grid), dim3(512), lds, s, p);
  }
  const int64_t rows = (int64_t)p.B * p.H * L;
  const dim3 sg((unsigned)((rows + 3) / 4));
  switch (nkb) {
    case 1: hipLaunchKernelGGL(tied_split_softmax_kernel<1>, sg, dim3(256), 0, s, ws, one, nsplit, p0.att, rows); break;
    case 2: hipLaunchKernelGGL(tied_split_softmax_kernel<2>, sg, dim3(256), 0, s, ws, one, nsplit, p0.att, rows); break;
    case 3: hipLaunchKernelGGL(tied_split_softmax_kernel<3>, sg, dim3(256), 0, s, ws, one, nsplit, p0.att, rows); break;
    default: hipLaunchKernelGGL(tied_split_softmax_kernel<4>, sg, dim3(256), 0, s, ws, one, nsplit, p0.att, rows); break;
  }
  return rf_launch_status();
}

static int tied_logits_dispatch(const TiedP& p, int L, hipStream_t s) {
#define RF_TIED(L_) \
  if (L == L_) return p.w ? launch_tied<L_, true>(p, s) : launch_tied<L_, false>(p, s);
  RF_TIED(256)
  RF_TIED(192)
  RF_TIED(128)
  RF_TIED(64)
#undef RF_TIED
  return RF_EINVAL;
}

static int tied_sym(const void* att, float* att_sym, int64_t sym_ld, int B, int H, int L, hipStream_t s) {
  const int64_t n = (int64_t)B * L * L * H;
  unsigned g = (unsigned)((n + 255) / 256);
  if (g > 8192u) g = 8192u;
  hipLaunchKernelGGL(tied_att_sym_kernel, dim3(g), dim3(256), 0, s, (const h16_t*)att, att_sym, sym_ld, B, H, L);
  return rf_launch_status();
}

extern "C" int rf_tied_logits_softmax(const void* q, const void* k, int64_t b_stride, int64_t n_stride, int64_t l_stride,
                                      void* att, float* att_sym, int64_t sym_ld, int B, int H, int N, int L, int d_head,
                                      void* stream) {
  if (!q || !k || !att || B <= 0 || H <= 0 || N <= 0) return RF_EINVAL;
  if (d_head != 32 || (L != 64 && L != 128 && L != 192 && L != 256)) return RF_EINVAL;  // (the caller falls back to rf_gemm + rf_tied_softmax)
  if (((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)att % 16) || b_stride % 8 || n_stride % 8 || l_stride % 8)
    return RF_EALIGN;
  TiedP p;
  p.q = (const h16_t*)q; p.k = (const h16_t*)k;
  p.b_stride = b_stride; p.n_stride = n_stride; p.l_stride = l_stride; p.h_stride = d_head;
  p.w = nullptr; p.w_b = p.w_h = p.w_n = 0; p.qscale = 1.f;
  p.att = (h16_t*)att; p.B = B; p.H = H; p.N = N;
  p.dbg = tied_dbg();
  if (p.dbg < 0) return p.dbg;
  hipStream_t s = (hipStream_t)stream;
  const int rc = tied_logits_dispatch(p, L, s);
  if (rc != 0 || !att_sym) return rc;
  return tied_sym(att, att_sym, sym_ld, B, H, L, s);
}

// ------------------------------------------------------------------------------------------------------------------
// attention . V: persistent workgroups walk (b, h, n) units; the 16*RT x L probability rows of a wave stay in registers
// as MFMA operands, the value tile of a unit ([L x 32] bf16, one contiguous block in the head-major layout) streams
// through a DMA ring and is consumed with hardware-transposed LDS reads (ds_read_b64_tr_b16).
// ------------------------------------------------------------------------------------------------------------------
struct TiedAvP {
  const h16_t* att;  // [B, H, L, L]
  const h16_t* v;
  int64_t v_b, v_n, v_h, v_l;  // element strides of v (head slice contiguous)
  h16_t* out;
  int64_t o_b, o_n, o_h, o_l;  // element strides of out (head slice contiguous)
  int B, H, N;
  int units_per_wg, nunits;
};

template <int L>
__global__ __launch_bounds__(256, 1) void tied_av_kernel(const TiedAvP p) {
  constexpr int RT = L / 64;    // 16-row query tiles per wave
  constexpr int KS = L / 32;    // contraction steps over the keys
  constexpr int NSTG = 4;       // value-tile ring
  constexpr int STAGE = L * 64;
  constexpr int VI = L / 16, PW = (VI + 3) / 4;
  constexpr int DUMP = NSTG * STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  const int u0 = blockIdx.x * p.units_per_wg;
  const int u1 = u0 + p.units_per_wg < p.nunits ? u0 + p.units_per_wg : p.nunits;
  if (u0 >= u1) return;

  // DMA of one value tile: instruction = 16 rows x 64 B; lane -> (row lane>>2, 16-byte chunk lane&3).  The image is
  // lane-linear; rows r and r+8 share LDS banks, so the chunk index is XORed with 2 on odd 8-row groups (source side,
  // and on the reads): the 32 lanes of a transposed read then touch 64 distinct banks.
  const int lrow = lane >> 2;
  const int c_src = (lane & 3) ^ (((lrow >> 3) & 1) << 1);
  const int64_t lane_off = (int64_t)lrow * p.v_l + c_src * 8;
  auto stage = [&](int u) {
    char* st = smem + ((u - u0) % NSTG) * STAGE;
    const bool live = u < u1;
    const int n = u % p.N, bh = u / p.N, h = bh % p.H, b = bh / p.H;
    const h16_t* vb = p.v + (int64_t)b * p.v_b + (int64_t)n * p.v_n + (int64_t)h * p.v_h;
#pragma unroll
    for (int t = 0; t < PW; ++t) {
      const int instr = t * 4 + wave;
      if (live && instr < VI)
        tied_glds16(vb + (int64_t)(instr * 16) * p.v_l + lane_off, st + instr * 1024);
      else
        tied_glds16(g_tied_zero16, smem + DUMP);
    }
  };
  // transposed-read byte offsets inside a stage (tile independent): key rows 32 s + 8 fq + 4 half + (fr >> 2), head dims
  // 16 c + 4 (fr & 3) .. +3
  int tr_off[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = fq * 8 + half * 4 + (fr >> 2);
      const int chunk = (2 * c + ((fr & 3) >> 1)) ^ (((row >> 3) & 1) << 1);
      tr_off[c][half] = row * 64 + chunk * 16 + (fr & 1) * 8;
    }

#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) stage(u0 + s);
  int u = u0;
  while (u < u1) {
    // ---- a run of units that share (b, h): the probability rows are loaded once, as MFMA-B fragments
    const int bh = u / p.N;
    const int h = bh % p.H, b = bh / p.H;
    const int uend = (bh + 1) * p.N < u1 ? (bh + 1) * p.N : u1;
    h16x8 af[RT][KS];  // att[i = 16 t + fr][32 s + 8 fq .. +7]
    {
      const h16_t* ab = p.att + ((int64_t)bh * L + wave * (16 * RT)) * L;
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) af[t][s] = *(const h16x8*)(ab + (int64_t)(t * 16 + fr) * L + s * 32 + fq * 8);
      // drain here (this also retires every DMA issued so far) and make the fragments opaque, so that the compiler's own
      // wait for these loads sits in front of the unit loop and not inside it
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(af[t][s]));
    }
    for (; u < uend; ++u) {
      // unit u's tile landed once only the younger operations are outstanding: per later unit PW DMAs + 2 RT stores
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSTG - 2) + (NSTG - 1) * 2 * RT) : "memory");
      __builtin_amdgcn_s_barrier();
      stage(u + NSTG - 1);
      const char* st = smem + ((u - u0) % NSTG) * STAGE;
      f32x4 acc[RT][2];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t][0] = acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // Transposed reads through inline asm: hipcc treats the ds_read_tr builtin as possibly aliasing the LDS-DMA writes in
      // flight and puts s_waitcnt vmcnt(0) in front of it, which would drain the ring every unit.  The reads of step s+1
      // are issued before the MFMAs of step s; each wait names its destinations, so no consumer can move above it.
      const unsigned sa0 = (unsigned)(size_t)(st - smem) + tr_off[0][0], sa1 = (unsigned)(size_t)(st - smem) + tr_off[0][1];
      const unsigned sa2 = (unsigned)(size_t)(st - smem) + tr_off[1][0], sa3 = (unsigned)(size_t)(st - smem) + tr_off[1][1];
      uint2 ra[4], rb[4];
#define RF_AV_ISSUE(R, S_)                                                                                       \
      asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%8\n\tds_read_b64_tr_b16 %1, %5 offset:%8\n\t"           \
                   "ds_read_b64_tr_b16 %2, %6 offset:%8\n\tds_read_b64_tr_b16 %3, %7 offset:%8"                 \
                   : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3])                                           \
                   : "v"(sa0), "v"(sa1), "v"(sa2), "v"(sa3), "i"((S_) * 2048))
#define RF_AV_LANDED(R) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]))
#define RF_AV_MFMAS(R, S_)                                                                                       \
      {                                                                                                          \
        TFrag vf[2];                                                                                             \
        vf[0].h[0] = R[0]; vf[0].h[1] = R[1];                                                                    \
        vf[1].h[0] = R[2]; vf[1].h[1] = R[3];                                                                    \
        _Pragma("unroll") for (int t = 0; t < RT; ++t) _Pragma("unroll") for (int c = 0; c < 2; ++c)             \
            /* value tile as MFMA-A, probability tile as MFMA-B: lane holds out[i = fr][d = 16 c + 4 fq .. +3] */ \
            acc[t][c] = rf_mfma16(vf[c].v, af[t][S_], acc[t][c], 0, 0, 0);        \
      }
#define RF_AV_STEP(S_, CUR, NXT)                                    \
      if constexpr (S_ < KS) {                                      \
        RF_AV_LANDED(CUR);                                          \
        if constexpr (S_ + 1 < KS) RF_AV_ISSUE(NXT, S_ + 1);        \
        RF_AV_MFMAS(CUR, (S_ < KS ? S_ : 0))                        \
        __builtin_amdgcn_sched_barrier(0);                          \
      }
      RF_AV_ISSUE(ra, 0);
      RF_AV_STEP(0, ra, rb) RF_AV_STEP(1, rb, ra) RF_AV_STEP(2, ra, rb) RF_AV_STEP(3, rb, ra)
      RF_AV_STEP(4, ra, rb) RF_AV_STEP(5, rb, ra) RF_AV_STEP(6, ra, rb) RF_AV_STEP(7, rb, ra)
#undef RF_AV_STEP
#undef RF_AV_MFMAS
#undef RF_AV_LANDED
#undef RF_AV_ISSUE
      static_assert(KS <= 8, "unrolled key steps");
      const int n = u % p.N;
      h16_t* ob = p.out + (int64_t)b * p.o_b + (int64_t)n * p.o_n + (int64_t)h * p.o_h +
                   (int64_t)(wave * 16 * RT + fr) * p.o_l + 4 * fq;
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          uint2 w;
          w.x = tpack2(acc[t][c][0], acc[t][c][1]);
          w.y = tpack2(acc[t][c][2], acc[t][c][3]);
          *(uint2*)(ob + (int64_t)(t * 16) * p.o_l + c * 16) = w;
        }
    }
  }
}

template <int L>
static int launch_tied_av(TiedAvP& p, hipStream_t s) {
  const int ncu = rf_num_cus() > 0 ? rf_num_cus() : 256;
  p.nunits = p.B * p.H * p.N;
  int grid = 2 * ncu;  // two co-resident workgroups per CU (64 KB of LDS each)
  if (grid > p.nunits) grid = p.nunits;
  p.units_per_wg = (p.nunits + grid - 1) / grid;
  grid = (p.nunits + p.units_per_wg - 1) / p.units_per_wg;
  if (const int e = rf_enable_big_lds<tied_av_kernel<L>>()) return e;
  hipLaunchKernelGGL(tied_av_kernel<L>, dim3((unsigned)grid), dim3(256), 4 * L * 64 + 1024, s, p);
  return rf_launch_status();
}

extern "C" int rf_tied_av(const void* att, const void* v, const int64_t v_strides[4], void* out, const int64_t o_strides[4],
                          int B, int H, int N, int L, int d_head, void* stream) {
  if (!att || !v || !out || B <= 0 || H <= 0 || N <= 0) return RF_EINVAL;
  if (d_head != 32 || (L != 64 && L != 128 && L != 192 && L != 256)) return RF_EINVAL;
  if (((uintptr_t)att % 16) || ((uintptr_t)v % 16) || ((uintptr_t)out % 8)) return RF_EALIGN;
  for (int i = 0; i < 4; ++i)
    if (v_strides[i] % 8 || o_strides[i] % 4) return RF_EALIGN;
  if ((int64_t)B * H * N > 0x7fffffffLL) return RF_EINVAL;
  TiedAvP p;
  p.att = (const h16_t*)att; p.v = (const h16_t*)v; p.out = (h16_t*)out;
  p.v_b = v_strides[0]; p.v_n = v_strides[1]; p.v_h = v_strides[2]; p.v_l = v_strides[3];
  p.o_b = o_strides[0]; p.o_n = o_strides[1]; p.o_h = o_strides[2]; p.o_l = o_strides[3];
  p.B = B; p.H = H; p.N = N;
  hipStream_t s = (hipStream_t)stream;
  if (L == 256) return launch_tied_av<256>(p, s);
  if (L == 192) return launch_tied_av<192>(p, s);
  if (L == 128) return launch_tied_av<128>(p, s);
  return launch_tied_av<64>(p, s);
}

// Logits + softmax of the tied attention on head-major operands (with the position weights folded in when w != NULL), optional
// symmetrised map.  L in {64, 128, 192, 256} (one-pass kernel, or the contraction-split form at 256 when a workspace is given)
// or L in {512, 768, 1024} (contraction-split form over 128-query x 256-key tiles: needs the workspace).
extern "C" int rf_tied_logits(const void* q, const void* k, const int64_t qk_strides[4], const float* w,
                              const int64_t w_strides[3], float qscale, void* att, float* att_sym, int64_t sym_ld, int B, int H,
                              int N, int L, int d_head, float* partial_ws, int64_t partial_ws_elems, void* stream) {
  if (!q || !k || !att || B <= 0 || H <= 0 || N <= 0) return RF_EINVAL;
  const bool small = L == 64 || L == 128 || L == 192 || L == 256;
  if (d_head != 32 || !(small || (L % 256 == 0 && L <= 1024))) return RF_EINVAL;
  if (((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)att % 16)) return RF_EALIGN;
  for (int i = 0; i < 4; ++i)
    if (qk_strides[i] % 8) return RF_EALIGN;
  if (w && (((uintptr_t)w % 16) || w_strides[0] % 4 || w_strides[1] % 4 || w_strides[2] % 4)) return RF_EALIGN;
  if (w && N % 4) return RF_EINVAL;  // the weight tile is staged four MSA rows per DMA instruction
  if (w && !small) return RF_EINVAL; // long rows: fold the weights into q (rf_gemm_desc.rs)
  TiedP p;
  p.q = (const h16_t*)q; p.k = (const h16_t*)k;
  p.b_stride = qk_strides[0]; p.n_stride = qk_strides[1]; p.h_stride = qk_strides[2]; p.l_stride = qk_strides[3];
  p.w = w;
  p.w_b = w ? w_strides[0] : 0; p.w_h = w ? w_strides[1] : 0; p.w_n = w ? w_strides[2] : 0;
  p.qscale = qscale;
  p.att = (h16_t*)att; p.B = B; p.H = H; p.N = N;
  p.dbg = tied_dbg();
  if (p.dbg < 0) return p.dbg;
  hipStream_t s = (hipStream_t)stream;
  int rc = tied_logits_split(p, L, partial_ws, partial_ws_elems, s);
  if (rc == 1) rc = small ? tied_logits_dispatch(p, L, s) : RF_EINVAL;  // (long rows have no one-pass kernel: the workspace is required)
  if (rc != 0) return rc;
  if (att_sym && (rc = tied_sym(att, att_sym, sym_ld, B, H, L, s)) != 0) return rc;
  return 0;
}

// Tied attention core in one call: rf_tied_logits, then attention . V.  q / k / v / out strides: {b, n, h, l} in elements, the
// 32-wide head slice contiguous.  L in {64, 128, 192, 256} (attention . V keeps whole probability rows in registers).
extern "C" int rf_tied_attention(const void* q, const void* k, const void* v, const int64_t qk_strides[4],
                                 const int64_t v_strides[4], const float* w, const int64_t w_strides[3], float qscale,
                                 void* att, float* att_sym, int64_t sym_ld, void* out, const int64_t o_strides[4], int B,
                                 int H, int N, int L, int d_head, float* partial_ws, int64_t partial_ws_elems, void* stream) {
  if (!v || !out) return RF_EINVAL;
  if (L != 64 && L != 128 && L != 192 && L != 256) return RF_EINVAL;
  const int rc = rf_tied_logits(q, k, qk_strides, w, w_strides, qscale, att, att_sym, sym_ld, B, H, N, L, d_head, partial_ws,
                                partial_ws_elems, stream);
  if (rc != 0) return rc;
  return rf_tied_av(att, v, v_strides, out, o_strides, B, H, N, L, d_head, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Position weights, collapsed form, on the matrix pipe: one wave per (b, l):
//   D[n, h] = sum_c xn[b,n,l,c] u[b,l,h,c]   (N/16 row tiles of MFMA 16x16x32, heads padded to 16),
//   w[b,h,n,l] = softmax_n(scale * D[n, h]).
// ------------------------------------------------------------------------------------------------------------------
template <int NT>  // NT = N / 16 row tiles
__global__ __launch_bounds__(256) void poswise_mfma_kernel(const h16_t* xn, const h16_t* u, float* w, int B, int N, int L,
                                                           int D, int H, float scale) {
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const int64_t bl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bl >= (int64_t)B * L) return;
  const int b = bl / L, l = bl % L;
  const h16_t* xr = xn + ((int64_t)b * N * L + l) * D + fq * 8;                  // + n * L * D + c
  const h16_t* ur = u + (bl * H + (fr < H ? fr : H - 1)) * (int64_t)D + fq * 8;  // (lanes >= H: a valid row, results discarded)
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < D; c += 32) {
    const h16x8 uf = *(const h16x8*)(ur + c);
    h16x8 xf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xf[t] = *(const h16x8*)(xr + (int64_t)(t * 16 + fr) * L * D + c);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      // MSA rows as MFMA-A, heads as MFMA-B: lane holds D[n = 16 t + 4 fq + r][h = fr]
      acc[t] = rf_mfma16(xf[t], uf, acc[t], 0, 0, 0);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[t][r] *= scale;
      mx = fmaxf(mx, acc[t][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sm = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[t][r] = __expf(acc[t][r] - mx);
      sm += acc[t][r];
    }
  sm += __shfl_xor(sm, 16, 64);
  sm += __shfl_xor(sm, 32, 64);
  const float inv = 1.f / sm;
  if (fr < H) {
    float* wo = w + (((int64_t)b * H + fr) * N) * L + l;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) wo[(int64_t)(t * 16 + 4 * fq + r) * L] = acc[t][r] * inv;
  }
}

extern "C" int rf_poswise_collapsed(const void* xn, const void* u, float* w, int B, int N, int L, int D, int H, float scale,
                                    void* stream) {
  if (!xn || !u || !w || B <= 0 || L <= 0) return RF_EINVAL;
  if (H < 1 || H > 16 || D % 32 || N % 16 || N < 16 || N > 256) return RF_EINVAL;
  if (((uintptr_t)xn % 16) || ((uintptr_t)u % 16)) return RF_EALIGN;
  const unsigned grid = (unsigned)(((int64_t)B * L + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
#define RF_PW(NT_)                                                                                                           \
  if (N == 16 * NT_) {                                                                                                       \
    hipLaunchKernelGGL(poswise_mfma_kernel<NT_>, dim3(grid), dim3(256), 0, s, (const h16_t*)xn, (const h16_t*)u, w, B, N, \
                       L, D, H, scale);                                                                                      \
    return rf_launch_status();                                                                                               \
  }
  RF_PW(1) RF_PW(2) RF_PW(3) RF_PW(4) RF_PW(5) RF_PW(6) RF_PW(7) RF_PW(8) RF_PW(12) RF_PW(16)
#undef RF_PW
  return RF_EINVAL;
}
