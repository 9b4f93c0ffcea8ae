// Fused OuterProductMean (rf.py:412-427) for gfx950 (MI355X): outer product over the MSA depth -> LayerNorm(1024) ->
// Linear(1024 -> d_pair) [-> the consumer's LayerNorm(d_pair)] in ONE kernel.  Only the [B, L, L, d_pair] result leaves
// the chip: the 1024-wide feature tensor (537 MB in bf16 at config 2, written and re-read by the round-1 path) never exists.
//
//   co[b,i,j,(u,v)] = sum_n x[b,n,i,u] y[b,n,j,v]                       (stage 1, MFMA, K = N)
//   out[b,i,j,o]    = sum_k LN(co[b,i,j,:])[k] W[o,k] + bias[o]         (stage 2, MFMA, K = 1024)
//
// The LayerNorm is folded algebraically so that stage 2 consumes the RAW outer-product block while its statistics are
// still being accumulated:   out = rstd * ( sum_k co_k W'[o,k]  -  mu * s_o ) + c_o,   W' = W * gamma (16-bit),
// s_o = sum_k W'[o,k],  c_o = sum_k W[o,k] beta_k + bias_o;  mu / rstd come from fp32 sums of the stage-1 accumulators.
//
// Round 3 layout: an EIGHT-wave workgroup (two waves on every SIMD, 256 registers each) owns a tile of TI x TJ = 16 x 8
// residue pairs and walks the 1024 features in 16 chunks of (8 u) x (8 v) = 64 features.  (Round 2 used nine waves, one
// per 32 output columns: three waves on one SIMD, so that SIMD carried 4/3 of the matrix work of the others, and the
// 168-register cap of a nine-wave workgroup spilled 46 registers per lane to scratch.)
//   stage 1 (all waves): D[(j,v), (i,u)] for the chunk -- 64 x 128 outputs, 16 MFMAs per wave; the lane that holds four
//            consecutive v of one (pair, u) writes them as 8 bytes into the chunk image A2[pair][64] (LDS, swizzled) and
//            adds them to the pair's running sum / sum of squares (registers);
//   stage 2: the 288 output columns = 8 x 32 "main" columns + 32 "extra" columns.  Wave w owns main columns
//            32 w .. 32 w + 31 for ALL 128 pairs (W' fragments straight from L2 into registers, double-buffered over
//            chunks: every main W' element is read once per tile, by one wave) and the extra columns 256 .. 287 for the
//            16 pairs of row tile w (their 4 KB W' slice per chunk arrives by DMA in LDS and is shared by all waves):
//            36 MFMAs per wave and chunk, the same on every wave.
// x / y / extra-W' chunks arrive by DMA (global_load_lds) one chunk ahead; one barrier per chunk.  The epilogue stages the
// normalised rows through LDS and stores whole 16-byte pieces of 576-byte rows (round 2 stored 8 bytes per lane into 16
// different rows per instruction: 81 of its 355 us), and the first operands of the NEXT tile are already in flight.
#include "common.h"

struct OuterP {
  const h16_t* xt;   // [B, L, 32, N]   x_t[b,i,u,n]  (MSA depth contiguous)
  const h16_t* yt;   // [B, L, 32, N]
  const h16_t* wp;   // [16, Dout, 64]  W * gamma, chunk-major: wp[c][o][uu * 8 + vv] = (W gamma)[o][(8 (c / 4) + uu) * 32 + 8 (c % 4) + vv]
  const float* s;    // [Dout]          row sums of wp
  const float* c;    // [Dout]          W beta + bias
  float* out;        // [B, L, L, Dout] fp32
  int B, L, Dout;
  float eps;
  // optional second LayerNorm over the Dout outputs of every pair (PairUpdateWithMsa.ln_coevol_feat, rf.py:443,486): when
  // y != NULL the kernel writes y[pair, 0:Dout] = LN2(out[pair, :]) in the 16-bit type (row stride y_ld elements) INSTEAD of `out`
  const float* g2;
  const float* b2;
  float eps2;
  h16_t* y;
  int64_t y_ld;
  int ntiles;
};

// global -> LDS DMA issued from inline assembly (M0 = wave-uniform LDS byte address of the instruction's 1 KB; written in the
// same statement that reads it): hipcc's wait-count pass then knows nothing of these DMAs.  With the builtin it tracks "LDS
// written by DMA" and puts s_waitcnt vmcnt(0) in front of every later ds_write it cannot disambiguate -- in the epilogue
// that would drain the next tile's prefetch right behind its issue.  Every consumer of DMA data in this kernel sits behind
// an explicit s_waitcnt vmcnt(0) + workgroup barrier (prologue, top of every chunk iteration).
#pragma clang diagnostic ignored "-Winline-asm"  // m0 on the clobber list is intended
__device__ __forceinline__ void outer_glds16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ unsigned outer_pack2(float a, float b) { return rf_pack2_h16(a, b); }
// workgroup barrier that publishes this wave's LDS writes but leaves its DMAs in flight (a __syncthreads() would drain them)
__device__ __forceinline__ void outer_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// NKS = N / 32 (stage-1 K steps): 4 (N = 128) or 2 (N = 64)
template <int NKS, bool LN2>
__global__ __launch_bounds__(512, 2) void outer_fused_kernel(const OuterP p) {
  constexpr int TI = 16, TJ = 8, PT = TI * TJ;   // pairs per tile
  constexpr int NB = NKS * 64;                   // bytes per operand row (N 16-bit values)
  constexpr int SPR = NKS * 4;                   // 16-byte slots per operand row
  constexpr int XCH = TI * 8 * NB;               // x chunk group: (i, u in group) rows
  constexpr int YCH = TJ * 8 * NB;               // y chunk: (j, v in group) rows
  constexpr int A2B = PT * 128;                  // chunk image: [pair][64 features] 16-bit
  constexpr int WXB = 32 * 128;                  // extra-column W' slice of a chunk: [32 columns][64 features]
  constexpr int X_OFF = 0, Y_OFF = 2 * XCH, A2_OFF = Y_OFF + 2 * YCH, WX_OFF = A2_OFF + 2 * A2B, ST_OFF = WX_OFF + 2 * WXB;
  constexpr int RPI = 1024 / NB;                 // operand rows per DMA instruction
  constexpr int PDX = XCH / 1024 / 8, PDY = YCH / 1024 / 8;  // DMA instructions per wave (exact: no padding)
  static_assert(XCH % 8192 == 0 && YCH % 8192 == 0, "whole DMA instructions per wave");
  // epilogue image of the finished tile: [128 pairs][288 values] 16-bit, rows padded to 592 bytes.  With 128 MSA rows
  // the next tile's first x / y chunks are in flight meanwhile in x[0] / y[0], so the image lives in x[1] (rows 0 .. R0-1)
  // and in y[1] + the chunk images (the rest); with 64 rows the buffers are too small for that: no prefetch, image from 0
  constexpr bool PREF = NKS == 4;
  constexpr int IPITCH = 592;
  constexpr int IMG0 = PREF ? X_OFF + XCH : 0, IMG1 = Y_OFF + YCH;
  constexpr int R0 = PREF ? XCH / IPITCH : PT;
  static_assert(!PREF || (PT - R0) * IPITCH <= YCH + 2 * A2B, "epilogue image does not fit beside the prefetched operands");
  static_assert(PREF || PT * IPITCH <= A2_OFF + 2 * A2B, "epilogue image does not fit");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int N = NKS * 32;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // (0: the kernel has no static __shared__)

  // operand-row swizzle (source side of the DMA and fragment reads): N = 128 -> rows are whole 256-byte bank rows,
  // slot ^ (row & 15); N = 64 -> two rows per bank row, slot ^ ((row >> 1) & 7)
  auto swz = [](int row) { return NKS == 4 ? (row & 15) : ((row >> 1) & 7); };

  // stage-2 constants of this wave: main columns o = 32 wave + 16 cc + 4 fq .. +3
  const int o_w = wave * 32;
  // W' is stored CHUNK-MAJOR, [16 chunks][Dout][64 features of the chunk] (ops.outer_fold): the 64 features a chunk contracts
  // over are one 128-byte line per output column, so a fragment load touches 16 lines of 64 useful bytes.  (Round 2 kept the
  // nn.Linear layout [Dout][1024], where those features are eight 16-byte runs 64 bytes apart: every load instruction
  // fetched 64 separate L2 sectors for 1 KB of fragments, 4096 L2 requests per CU and chunk -- the kernel's real bound.)
  const h16_t* const wrow = p.wp + (int64_t)(o_w + fr) * 64 + fq * 8;  // + (c * Dout + cc * 16) * 64 + s2 * 32

  // stage-1 geometry of this wave: y row tile art (rows = (j, v)), x column tiles bct0 .. bct0 + 3 (cols = (i, u))
  const int art = wave & 3, bct0 = (wave >> 2) * 4;
  const int jl = 2 * art + (fq >> 1);            // j of this lane's four outputs
  // column tile t of this wave: i_l = 2 (bct0 + t) + (fr >> 3): pair = pair0 + 16 t, and ((pair >> 1) & 7) does not depend on t
  const int pair0 = (2 * bct0 + (fr >> 3)) * TJ + jl;
  const int a2w0 = pair0 * 128 + (((fr & 7) ^ ((pair0 >> 1) & 7)) << 4) + (fq & 1) * 8;  // slot = u, swizzled like the reads
  const int y_rd = Y_OFF + (16 * art + fr) * NB;      // A operand rows of the y chunk
  const int x_rd0 = X_OFF + (16 * bct0 + fr) * NB;    // + t * 16 rows
  // (operand rows r with r & 15 == fr; for N = 64 the swizzle needs (r >> 1) & 7 = (fr >> 1) & 7 -- tiles start at multiples of 16)
  const int fsw = NKS == 4 ? fr : ((fr >> 1) & 7);
  // stage-2 reads of the chunk image (pair = 16 rt + fr, slot 4 s2 + fq) and of the extra W' slice (column 16 cc + fr):
  // both images have 128-byte rows and the same swizzle ((row >> 1) & 7 == (fr >> 1) & 7 for every row a lane reads)
  int a2r[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) a2r[s2] = fr * 128 + (((4 * s2 + fq) ^ ((fr >> 1) & 7)) << 4);

  // XCD-aware tile walk.  Workgroups b and b + 8 share an XCD (and its 4 MB L2); the tile sequence (b, i tile, j tile; j
  // fastest) is cut into 8 contiguous ranges, one per XCD group, and the workgroups of a group walk their range side by
  // side: the tiles in flight on one XCD share their 128 KB x rows (same i tile) and the 64 KB y rows of a j tile come
  // back from L2 for every later i tile of the range.  (Round 2 walked `tile = blockIdx.x + k * gridDim.x`: the 8
  // workgroups sharing an i tile sat on 8 different L2s and x / y were fetched ~20 times over, 328 MB per launch.)
  int t_begin = blockIdx.x, t_end = p.ntiles, t_step = gridDim.x;
  if (gridDim.x >= 8) {
    const int x = blockIdx.x & 7, q = p.ntiles >> 3, r = p.ntiles & 7;
    const int lo = x * q + (x < r ? x : r);
    t_end = lo + q + (x < r ? 1 : 0);
    t_step = ((int)gridDim.x - x + 7) >> 3;  // workgroups in this group
    t_begin = lo + (int)(blockIdx.x >> 3);
  }

  const h16_t* xb = nullptr;
  const h16_t* yb = nullptr;
  int b = 0, i0 = 0, j0 = 0;
  auto set_tile = [&](int tile) {
    const int jt = tile % (p.L / TJ), t2 = tile / (p.L / TJ);
    const int it = t2 % (p.L / TI);
    b = t2 / (p.L / TI);
    i0 = it * TI;
    j0 = jt * TJ;
    xb = p.xt + ((int64_t)b * p.L + i0) * 32 * N;
    yb = p.yt + ((int64_t)b * p.L + j0) * 32 * N;
  };
  auto dma_x = [&](int ug) {  // rows (i_l, u_l): i_l = row >> 3, u = 8 ug + (row & 7)
    const unsigned dst = lds0 + X_OFF + (ug & 1) * XCH;
    int ln = lane;  // opaque: the per-lane source offsets are rebuilt at every call instead of living in registers for the whole kernel
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int t = 0; t < PDX; ++t) {
      const int q = wave + 8 * t;
      const int row = q * RPI + ln / SPR, sl = ln % SPR;
      outer_glds16(xb + ((int64_t)(row >> 3) * 32 + ug * 8 + (row & 7)) * N + ((sl ^ (swz(row) & (SPR - 1))) << 3),
                   __builtin_amdgcn_readfirstlane(dst + q * 1024));
    }
  };
  auto dma_y = [&](int c) {   // rows (j_l, v_l): j_l = row >> 3, v = 8 vg + (row & 7)
    const unsigned dst = lds0 + Y_OFF + (c & 1) * YCH;
    const int vg = c & 3;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int t = 0; t < PDY; ++t) {
      const int q = wave + 8 * t;
      const int row = q * RPI + ln / SPR, sl = ln % SPR;
      outer_glds16(yb + ((int64_t)(row >> 3) * 32 + vg * 8 + (row & 7)) * N + ((sl ^ (swz(row) & (SPR - 1))) << 3),
                   __builtin_amdgcn_readfirstlane(dst + q * 1024));
    }
  };
  // extra-column W' slice of chunk c: rows o = 256 + r (r < 32), 4 KB of contiguous memory in the chunk-major layout: one
  // instruction = 8 rows x 8 slots; waves 0-3 issue one each (the slice is the same for every tile: L2-resident)
  auto dma_wx = [&](int c) {
    if (wave < 4) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int row = wave * 8 + (ln >> 3), sl = ln & 7;
      const int uu = sl ^ ((row >> 1) & 7);
      outer_glds16(p.wp + ((int64_t)c * 288 + 256 + row) * 64 + uu * 8,
                   __builtin_amdgcn_readfirstlane(lds0 + WX_OFF + (c & 1) * WXB + wave * 1024));
    }
  };
  auto wload = [&](int c, int s2, int cc) {
    return *(const h16x8*)(wrow + ((int64_t)c * 288 + cc * 16) * 64 + s2 * 32);
  };

  // main W' fragments of two consecutive chunks: [s2][cc]; set A serves even chunks, set B odd ones
  h16x8 wfA[2][2], wfB[2][2];
  if (t_begin < t_end) {
    set_tile(t_begin);
    dma_x(0);
    dma_y(0);
    dma_wx(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) wfA[s2][cc] = wload(0, s2, cc);
  }

  for (int tile = t_begin; tile < t_end; tile += t_step) {
    f32x4 acc[8][2], accx[2];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) acc[rt][0] = acc[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accx[0] = accx[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    // stage 1 of chunk c: D[(j,v)][(i,u)] over the MSA depth -> chunk image A2[c & 1], running statistics
    auto stage1 = [&](int c) {
      const char* xs = smem + ((c >> 2) & 1) * XCH;
      const char* ys = smem + (c & 1) * YCH;
      f32x4 d[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const int sl = ((4 * s + fq) ^ fsw) << 4;
        const h16x8 yf = *(const h16x8*)(ys + y_rd + sl);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const h16x8 xf = *(const h16x8*)(xs + x_rd0 + t * 16 * NB + sl);
          d[t] = rf_mfma16(yf, xf, d[t], 0, 0, 0);  // lane: (j,v) = 4 fq + r, (i,u) = fr
        }
      }
      // (LDS store through inline asm: hipcc puts s_waitcnt vmcnt(0) in front of a visible ds_write while LDS-DMAs are in
      // flight -- it cannot tell that the DMA targets and the chunk image are disjoint.  The dynamic LDS segment starts at
      // LDS address 0: the kernel has no static __shared__.)
      const unsigned a2a = (unsigned)(A2_OFF + (c & 1) * A2B + a2w0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ssum[t] += (d[t][0] + d[t][1]) + (d[t][2] + d[t][3]);
        ssq[t] += (d[t][0] * d[t][0] + d[t][1] * d[t][1]) + (d[t][2] * d[t][2] + d[t][3] * d[t][3]);
        uint2 w;
        w.x = outer_pack2(d[t][0], d[t][1]);
        w.y = outer_pack2(d[t][2], d[t][3]);
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a2a), "v"(w), "i"(t * 2048) : "memory");
      }
    };
    // stage 2 of chunk c: out[pair, o] += A2[pair, chunk] . W'[o, chunk].  W' tile as MFMA-A, chunk image as MFMA-B:
    // lane holds out[pair = 16 rt + fr][o = 32 w + 16 cc + 4 fq .. +3] (main) / [pair = 16 w + fr][o = 256 + ..] (extra)
    auto stage2 = [&](int c, const h16x8 (&wf)[2][2]) {
      const char* a2 = smem + A2_OFF + (c & 1) * A2B;
      const char* wx = smem + WX_OFF + (c & 1) * WXB;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const h16x8 ax = *(const h16x8*)(a2 + wave * 2048 + a2r[s2]);
        h16x8 wxf[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) wxf[cc] = *(const h16x8*)(wx + cc * 2048 + a2r[s2]);
#pragma unroll
        for (int r0 = 0; r0 < 8; r0 += 4) {  // four image fragments in flight per batch of eight MFMAs
          h16x8 af[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) af[k] = *(const h16x8*)(a2 + (r0 + k) * 2048 + a2r[s2]);
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) acc[r0 + k][cc] = rf_mfma16(wf[s2][cc], af[k], acc[r0 + k][cc], 0, 0, 0);
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) accx[cc] = rf_mfma16(wxf[cc], ax, accx[cc], 0, 0, 0);
      }
    };
    // iteration c: everything issued during iteration c - 1 (the DMAs of chunk c + 1, the extra W' slice and the main W'
    // fragments of chunk c) is waited for at the top -- it had a whole iteration to land, and nothing younger exists, so
    // the plain vmcnt(0) is exact.  Then ONE barrier; the DMAs of chunk c + 2, the W' loads of chunk c + 1 into the OTHER
    // register set; stage 2 of chunk c beside stage 1 of chunk c + 1 (independent work in one instruction stream).
    auto iter = [&](int c, h16x8 (&cur)[2][2], h16x8 (&nxt)[2][2]) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) asm volatile("" : "+v"(cur[s2][cc]));  // (pins the compiler's own wait for these loads here, ahead of the DMA issue)
      outer_lds_barrier();  // image of chunk c complete; operands of chunk c + 1 visible; buffers of chunk c - 1 free
      if (c + 2 < 16) {
        dma_y(c + 2);                              // into y[c & 1] (read by stage 1 of chunk c, done)
        if (((c + 2) & 3) == 0) dma_x((c + 2) >> 2);  // group of chunks c + 2 .. c + 5, into the buffer of group (c >> 2) - 1
      }
      if (c + 1 < 16) {
        dma_wx(c + 1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) nxt[s2][cc] = wload(c + 1, s2, cc);
      }
      stage2(c, cur);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < 16) stage1(c + 1);
    };

    // prologue: chunk 0's operands landed -> stage 1 of chunk 0 (its image is consumed in iteration 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    outer_lds_barrier();  // (also: every wave has finished reading the previous tile's epilogue image)
    dma_y(1);
    stage1(0);
    for (int c = 0; c < 16; c += 2) {
      iter(c, wfA, wfB);
      iter(c + 1, wfB, wfA);
    }

    // ---------------- next tile's first operands (x[0], y[0], wx[0] and register set A are free) ----------------
    const int b_t = b, i0_t = i0, j0_t = j0;
    const bool more = tile + t_step < t_end;
    auto prefetch_next = [&]() {
      set_tile(tile + t_step);
      dma_x(0);
      dma_y(0);
      dma_wx(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) wfA[s2][cc] = wload(0, s2, cc);
    };
    if (PREF && more) prefetch_next();

    // ---------------- statistics: reduce over the 16 lanes that share a pair, publish, normalise ----------------
    float* stats = (float*)(smem + ST_OFF);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float a = ssum[t], q = ssq[t];
#pragma unroll
      for (int o = 1; o <= 4; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        q += __shfl_xor(q, o, 64);
      }
      a += __shfl_xor(a, 16, 64);
      q += __shfl_xor(q, 16, 64);
      if ((fr & 7) == 0 && (fq & 1) == 0) *(float2*)(stats + 2 * (pair0 + 16 * t)) = make_float2(a, q);
    }
    outer_lds_barrier();  // statistics visible; every wave is through stage 2 of the last chunk (the chunk images are dead)
    f32x4 s4[2], c4[2], sx[2], cx[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      s4[cc] = *(const f32x4*)(p.s + o_w + cc * 16 + 4 * fq);
      c4[cc] = *(const f32x4*)(p.c + o_w + cc * 16 + 4 * fq);
      sx[cc] = *(const f32x4*)(p.s + 256 + cc * 16 + 4 * fq);
      cx[cc] = *(const f32x4*)(p.c + 256 + cc * 16 + 4 * fq);
    }
    float* part = (float*)(smem + A2_OFF);  // LN2 partial sums [9 column groups][128 pairs][2]
    {
      // extra columns of row tile `wave`: Linear(LayerNorm_1024(co)) in the accumulators, like the main ones below
      const int pr = 16 * wave + fr;
      const float2 sq = *(const float2*)(stats + 2 * pr);
      const float mu = sq.x * (1.0f / 1024.0f);
      const float rstd = rsqrtf(fmaxf(sq.y * (1.0f / 1024.0f) - mu * mu, 0.f) + p.eps);
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        accx[cc] = (accx[cc] - mu * sx[cc]) * rstd + cx[cc];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a += accx[cc][e];
          q = fmaf(accx[cc][e], accx[cc][e], q);
        }
      }
      if constexpr (LN2) {
        a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
        a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
        if (fq == 0) *(float2*)(part + (8 * PT + pr) * 2) = make_float2(a, q);
      } else {
        float* orow = p.out + (((int64_t)b_t * p.L + i0_t + (pr >> 3)) * p.L + j0_t + (pr & 7)) * p.Dout + 256 + 4 * fq;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) *(f32x4*)(orow + cc * 16) = accx[cc];
      }
    }
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
      const int pr = 16 * rt + fr;
      const float2 sq = *(const float2*)(stats + 2 * pr);
      const float mu = sq.x * (1.0f / 1024.0f);
      const float var = fmaxf(sq.y * (1.0f / 1024.0f) - mu * mu, 0.f);
      const float rstd = rsqrtf(var + p.eps);
      if constexpr (LN2) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          acc[rt][cc] = (acc[rt][cc] - mu * s4[cc]) * rstd + c4[cc];  // Linear(LayerNorm_1024(co)), kept in the accumulators
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a += acc[rt][cc][e];
            q = fmaf(acc[rt][cc][e], acc[rt][cc][e], q);
          }
        }
        a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
        a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
        if (fq == 0) *(float2*)(part + (wave * PT + pr) * 2) = make_float2(a, q);
      } else {
        float* orow = p.out + (((int64_t)b_t * p.L + i0_t + (pr >> 3)) * p.L + j0_t + (pr & 7)) * p.Dout + o_w + 4 * fq;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const f32x4 v = (acc[rt][cc] - mu * s4[cc]) * rstd + c4[cc];
          *(f32x4*)(orow + cc * 16) = v;
        }
      }
    }
    if constexpr (LN2) {
      // second LayerNorm (over the 288 outputs of a pair = nine 32-column groups): the partial sums meet in LDS
      outer_lds_barrier();
      if (tid < PT) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w9 = 0; w9 < 9; ++w9) {
          const float2 t2 = *(const float2*)(part + (w9 * PT + tid) * 2);
          a += t2.x;
          q += t2.y;
        }
        const float m2 = a / (float)p.Dout;
        const float v2 = fmaxf(q / (float)p.Dout - m2 * m2, 0.f);
        *(float2*)(stats + 2 * tid) = make_float2(m2, rsqrtf(v2 + p.eps2));
      }
      outer_lds_barrier();  // (the partial sums are dead from here: the image below overlays them)
      auto img_row = [&](int pr) { return smem + (pr < R0 ? IMG0 + pr * IPITCH : IMG1 + (pr - R0) * IPITCH); };
      f32x4 g4[2], e4[2];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        g4[cc] = *(const f32x4*)(p.g2 + o_w + cc * 16 + 4 * fq);
        e4[cc] = *(const f32x4*)(p.b2 + o_w + cc * 16 + 4 * fq);
      }
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        const int pr = 16 * rt + fr;
        const float2 ms = *(const float2*)(stats + 2 * pr);
        char* irow = img_row(pr) + (o_w + 4 * fq) * 2;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const f32x4 v = (acc[rt][cc] - ms.x) * ms.y * g4[cc] + e4[cc];
          uint2 w;
          w.x = outer_pack2(v[0], v[1]);
          w.y = outer_pack2(v[2], v[3]);
          *(uint2*)(irow + cc * 32) = w;
        }
      }
      {
        const int pr = 16 * wave + fr;
        const float2 ms = *(const float2*)(stats + 2 * pr);
        char* irow = img_row(pr) + (256 + 4 * fq) * 2;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const f32x4 gx = *(const f32x4*)(p.g2 + 256 + cc * 16 + 4 * fq), ex = *(const f32x4*)(p.b2 + 256 + cc * 16 + 4 * fq);
          const f32x4 v = (accx[cc] - ms.x) * ms.y * gx + ex;
          uint2 w;
          w.x = outer_pack2(v[0], v[1]);
          w.y = outer_pack2(v[2], v[3]);
          *(uint2*)(irow + cc * 32) = w;
        }
      }
      outer_lds_barrier();  // image complete
      // rows leave as 36 pieces of 16 bytes: a wave instruction stores 1 KB of (almost) contiguous memory
      constexpr int NPC = PT * 36;
#pragma unroll
      for (int k = 0; k < (NPC + 511) / 512; ++k) {
        const int idx = tid + 512 * k;
        if (NPC % 512 == 0 || idx < NPC) {
          const int pr = idx / 36, ch = idx % 36;
          const f32x4 v = *(const f32x4*)(img_row(pr) + ch * 16);
          h16_t* yrow = p.y + (((int64_t)b_t * p.L + i0_t + (pr >> 3)) * p.L + j0_t + (pr & 7)) * p.y_ld + ch * 8;
          *(f32x4*)yrow = v;
        }
      }
    }
    if (!(PREF && more) && more) {
      outer_lds_barrier();  // (no prefetch variant: the image / statistics must be drained before the operands land on them)
      prefetch_next();
    }
  }
}

template <int NKS, bool LN2>
static int launch_outer(OuterP& p, hipStream_t s) {
  constexpr int NB = NKS * 64;
  constexpr int LDS = 2 * (16 * 8 * NB) + 2 * (8 * 8 * NB) + 2 * (128 * 128) + 2 * (32 * 128) + 128 * 8;
  const int ncu = rf_num_cus() > 0 ? rf_num_cus() : 256;
  p.ntiles = p.B * (p.L / 16) * (p.L / 8);
  const int grid = p.ntiles < ncu ? p.ntiles : ncu;
  if (const int e = rf_enable_big_lds<outer_fused_kernel<NKS, LN2>>()) return e;
  hipLaunchKernelGGL((outer_fused_kernel<NKS, LN2>), dim3((unsigned)grid), dim3(512), LDS, s, p);
  return rf_launch_status();
}

extern "C" int rf_outer_product_ln_linear(const void* xt, const void* yt, const void* wprime, const float* s, const float* c,
                                          float* out, int B, int L, int N, int P, int Dout, float eps, const float* ln2_gamma,
                                          const float* ln2_beta, float ln2_eps, void* y, int64_t y_ld, void* stream) {
  if (!xt || !yt || !wprime || !s || !c || B <= 0) return RF_EINVAL;
  if (!y && !out) return RF_EINVAL;
  if (y && (!ln2_gamma || !ln2_beta || y_ld < Dout || y_ld % 8 || ((uintptr_t)y % 16) || ((uintptr_t)ln2_gamma % 16) || ((uintptr_t)ln2_beta % 16)))
    return RF_EINVAL;
  if (P != 32 || Dout != 288 || (N != 128 && N != 64) || L % 16 != 0 || L < 16) return RF_EINVAL;  // (other shapes: rf_gemm + rf_layernorm)
  if (((uintptr_t)xt % 16) || ((uintptr_t)yt % 16) || ((uintptr_t)wprime % 16) || ((uintptr_t)s % 16) || ((uintptr_t)c % 16) ||
      ((uintptr_t)out % 16))
    return RF_EALIGN;
  OuterP p;
  p.xt = (const h16_t*)xt; p.yt = (const h16_t*)yt; p.wp = (const h16_t*)wprime; p.s = s; p.c = c; p.out = out;
  p.B = B; p.L = L; p.Dout = Dout; p.eps = eps;
  p.g2 = ln2_gamma; p.b2 = ln2_beta; p.eps2 = ln2_eps; p.y = (h16_t*)y; p.y_ld = y_ld;
  hipStream_t st = (hipStream_t)stream;
  if (y) return N == 128 ? launch_outer<4, true>(p, st) : launch_outer<2, true>(p, st);
  return N == 128 ? launch_outer<4, false>(p, st) : launch_outer<2, false>(p, st);
}
