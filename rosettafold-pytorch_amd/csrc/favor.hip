// Fused FAVOR+ (Performer) linear attention for gfx950: one persistent 4-wave workgroup per CU walks the
// (batch, outer index, head) items; per item the whole chain
//     k' = phi(k Pc^T)   ctx = k'^T v   ksum = sum_s k'      (phase A)
//     q' = phi(q Pc^T)   out = (q' ctx) / (q' . ksum)        (phase B)
// runs on-chip: q', k' and ctx never touch HBM (the unfused form moves ~5 GB per attention at config 2).
// Replaces performer_pytorch.SelfAttention's fast_attention as called at rf.py:313-318 (softmax kernel,
// MSA columns) and rf.py:505-518 (generalized ReLU kernel, pair rows/columns).  Third-party math restated
// from the published algorithm (parity unpinned); checked against the oracle and against
// the unfused kernel chain.
//
// MFMA plan (v_mfma_f32_16x16x32_bf16).  Accumulator tiles are consumed as operands of the next product
// without leaving registers: a 16x16 f32 accumulator holds, per lane, 4 consecutive ROWS of one column, so
// two vertically adjacent tiles give the 8 k-slots of an A/B fragment whose contraction index is the
// accumulator's row index (k order permuted identically on the other operand):
//   A1: D[s,m]   = K[s,:] . Pc[m,:]         A = K rows (LDS), B = Pc rows (registers, this wave's m slice)
//   A2: ctx[m,d] += k'[s,m] v[s,d]          A = k' (from A1's accumulators), B = V via ds_read_b64_tr_b16
//   B1: D[m,s]   = Pc[m,:] . Q[s,:]         A = Pc rows (LDS), B = Q rows (registers, this wave's s slice)
//   B2: num[d,s] += ctx[m,d] q'[m,s]        A = ctx^T (LDS, bf16), B = q' (from B1's accumulators)
// Phase A splits the 17 feature tiles (266 features) over the 4 waves, phase B splits the sequence.
// K and V of the NEXT item are DMA-prefetched (global_load_lds) while phase B runs.
#include <type_traits>

#include "common.h"

#define FV_DH 64
#define FV_M 266
#define FV_MT 17        // feature tiles of 16 that contain valid features
#define FV_MPAD 288
#define FV_CTX_LD 592   // bytes per ctx^T row (288 bf16 + pad: conflict-free ds_read_b64)
#define FV_DT 5         // value tiles: 4 x 16 head dims + the ones column (d = 64) that yields sum_s k' and the denominator
#define FV_DROWS 80

struct FavorAttnP {
  const h16_t* qkv;  // [.., 3*inner] rows; q | k | v
  const h16_t* pc;   // [288][64] projection pre-scaled by d^-1/4, zero rows beyond 266
  h16_t* out;        // [.., inner]
  int64_t x_b, x_o, x_s;  // element strides of qkv for batch / outer index / sequence index
  int64_t x_h;            // element stride of qkv between heads (64 for rows holding all heads, Ls*64 for head-major tiles)
  int64_t o_b, o_o, o_s;  // same for out
  int q_off, k_off, v_off;
  int n_o, n_h, nitems;
  int nchunks;  // sequence = nchunks * LS rows (ReLU kernel only; softmax kernel: 1)
  int dbg;      // timing experiments only: 1 = skip the phase-A MFMA loop, 2 = skip the phase-B loop
  float eps;
  float ctx_scale;  // f16 build: 2^-ceil(log2(sequence length)), see FV_CS; 1 in the bf16 build
};

// fp16 range (librfmi_f16.so): the context sum_s k'[s,m] v[s,d] and the k' sums in its ones column grow with the sequence
// length and would leave fp16's range (65504) for long sequences.  Numerator (q' ctx) and denominator (q' . sum_s k') of the
// attention both carry the context linearly, so a common power-of-two factor cancels exactly: the f16 build publishes the
// context scaled by 2^-ceil(log2(sequence length)) (a mean instead of a sum).  The bf16 build (fp32's exponent range)
// compiles the factor out.
#ifdef RF_H16_IS_F16
#define FV_CS(x) ((x) * p.ctx_scale)
#else
#define FV_CS(x) (x)
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ void fv_glds(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// DMA a [nrows][64] bf16 tile (rows `stride` elements apart) into the swizzled LDS image at lds_off
__device__ __forceinline__ void fv_load_tile(char* smem, int lds_off, const h16_t* g, int64_t stride, int nrows,
                                             int wave, int lane) {
  const int ninstr = nrows * 8 / 64;
  for (int it = wave; it < ninstr; it += 4) {
    const int slot = it * 64 + lane;
    const int row = slot >> 3;
    const int clog = (slot & 7) ^ (row & 7);
    fv_glds(g + (int64_t)row * stride + clog * 8, smem + lds_off + it * 1024);
  }
}

__device__ __forceinline__ unsigned pack2(float a, float b) { return rf_pack2_h16(a, b); }
__device__ __forceinline__ float rbf(float x) { return h2f(f2h(x)); }

// The value of an MFMA accumulator register that an LDS instruction (the denominator broadcast: __shfl = ds_bpermute_b32) is
// about to read.  gfx950 does not interlock an LDS / VALU / VMEM read of a v_mfma destination: the 7 wait states (4-pass
// 16x16x32) are software's job, and hipcc's hazard recognizer does NOT cover the path that reaches the reader through the
// loop-exit branch behind the accumulator's last MFMA.  Round 2's kernel had exactly that (`v_mfma ... v[26:29]; s_cbranch
// .LBB5_70; .LBB5_70: ds_bpermute_b32 v9, v54, v26`, zero wait states): whenever the SIMD's other wave kept the matrix pipe
// busy, the younger wave's shuffle read the denominator before the last feature block had been added, and its 16 output rows
// came out scaled by 1.01 - 1.08 in a few items per launch (tools/favor_rootcause/: experiments, forensics, ISA).  The nops
// live in the SAME block as the reader, so every path passes them; tools/isa_hazard_scan.py audits the whole library.
__device__ __forceinline__ float fv_mfma_done(float acc_reg) {
  asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc_reg));
  return acc_reg;
}

union Frag {
  h16x8 v;
  unsigned u[4];
  uint2 h[2];
};

template <int LS, bool SOFTMAX>
__global__ __launch_bounds__(256, 1) void favor_attention_kernel(const FavorAttnP p) {
  constexpr int ST = LS / 64;    // s-tiles per wave in phase B
  constexpr int NSB = LS / 32;   // s-blocks (pairs of s-tiles) in phase A
  constexpr int PC_OFF = 0;
  constexpr int K_OFF = FV_MPAD * 128;
  constexpr int V_OFF = K_OFF + LS * 128;
  constexpr int CTX_OFF = V_OFF + LS * 128;
  constexpr int DIAG_OFF = CTX_OFF + FV_DROWS * FV_CTX_LD;
  constexpr int RED_OFF = DIAG_OFF + LS * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
  const int fr = lane & 15, fq = lane >> 4;
  // this wave's feature tiles in phase A: 5,4,4,4
  const int nm = wave == 0 ? 5 : 4;
  const int m0t = wave == 0 ? 0 : 5 + 4 * (wave - 1);
  const f32x4 epsv = {p.eps, p.eps, p.eps, p.eps};
  // "ones column": value column d = 64 is identically 1, so ctx^T row 64 = sum_s k' (the normaliser's k' sums) and
  // the numerator tile of d-tile 4 carries the denominator in its row 64 -- both ride on the MFMA pipe.
  Frag ones;
#pragma unroll
  for (int k = 0; k < 4; ++k) ones.u[k] = fr == 0 ? RF_H16_ONE2 : 0u;

  // one-time: projection image, zeroed ctx^T (its padded columns are read as MFMA operands)
  fv_load_tile(smem, PC_OFF, p.pc, FV_DH, FV_MPAD, wave, lane);
  for (int i = tid; i < FV_DROWS * FV_CTX_LD / 4; i += 256) ((unsigned*)(smem + CTX_OFF))[i] = 0u;

  auto item_base = [&](int item, int64_t& xb, int64_t& ob) {
    const int h = item % p.n_h;
    const int t = item / p.n_h;
    const int o = t % p.n_o, b = t / p.n_o;
    xb = (int64_t)b * p.x_b + (int64_t)o * p.x_o + (int64_t)h * p.x_h;
    ob = (int64_t)b * p.o_b + (int64_t)o * p.o_o + h * FV_DH;
  };

  const int nch = p.nchunks;  // > 1: long sequence walked in LS-row chunks (no cross-item prefetch then)
  int item = blockIdx.x;
  if (item < p.nitems) {
    int64_t xb, ob;
    item_base(item, xb, ob);
    fv_load_tile(smem, K_OFF, p.qkv + xb + p.k_off, p.x_s, LS, wave, lane);
    fv_load_tile(smem, V_OFF, p.qkv + xb + p.v_off, p.x_s, LS, wave, lane);
  }
  bool first = true;
  for (; item < p.nitems; item += gridDim.x) {
    int64_t xb, ob;
    item_base(item, xb, ob);
    // wait for the K/V DMAs but NOT for the previous item's output stores (the ST*4 youngest operations of this wave):
    // a counted vmcnt lets them drain behind this item's phase A
    if (first)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST * 4) : "memory");
    first = false;
    __syncthreads();  // K, V (and Pc on the first item) have landed; previous item fully consumed
    // Q fragments of this wave's rows straight from global; consumed in phase B, so phase A hides the latency
    h16x8 qf[ST][2];
    auto load_q = [&](int chunk) {
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        const int s = chunk * LS + (wave * ST + t) * 16 + fr;
        const h16_t* qrow = p.qkv + xb + p.q_off + (int64_t)s * p.x_s;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[t][kk] = *(const h16x8*)(qrow + (kk * 4 + fq) * 8);
      }
    };
    load_q(0);

    float gmax = 0.f;
    if constexpr (SOFTMAX) {
      // diag_k[s] = |k_s|^2 / (2 sqrt(d)) in log2 units (Pc carries log2 e); one row per thread
      for (int s = tid; s < LS; s += 256) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const h16x8 x = *(const h16x8*)(smem + K_OFF + s * 128 + c * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = h2f((h16_t)x[e]);
            a = fmaf(f, f, a);
          }
        }
        ((float*)(smem + DIAG_OFF))[s] = a * (0.0625f * 1.4426950408889634f);
      }
    }

    // ---------------- phase A ----------------
    h16x8 pf[5][2];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        if (j < nm) pf[j][kk] = *(const h16x8*)(smem + PC_OFF + swz_off((m0t + j) * 16 + fr, kk * 4 + fq));

    if constexpr (SOFTMAX) {
      // pass 0: global max of the key logits over (s, m < 266)
      float mx = -INFINITY;
      for (int u = 0; u < NSB; ++u) {
        h16x8 kf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            kf[t][kk] = *(const h16x8*)(smem + K_OFF + swz_off((2 * u + t) * 16 + fr, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (j < nm) {
            const bool valid = (m0t + j) * 16 + fr < FV_M;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              f32x4 a = {0.f, 0.f, 0.f, 0.f};
              a = rf_mfma16(kf[t][0], pf[j][0], a, 0, 0, 0);
              a = rf_mfma16(kf[t][1], pf[j][1], a, 0, 0, 0);
              if (valid) mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
            }
          }
        }
      }
      mx = wave_max(mx);
      if (lane == 0) ((float*)(smem + RED_OFF))[wave] = mx;
      __syncthreads();  // also publishes diag_k
      const float* red = (const float*)(smem + RED_OFF);
      gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }

    f32x4 ctx[5][FV_DT];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int i = 0; i < FV_DT; ++i) ctx[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ck = 0; ck < nch; ++ck) {
    if (ck > 0) {  // next LS-row chunk of K and V (everyone is done with the previous one)
      __syncthreads();
      fv_load_tile(smem, K_OFF, p.qkv + xb + p.k_off + (int64_t)ck * LS * p.x_s, p.x_s, LS, wave, lane);
      fv_load_tile(smem, V_OFF, p.qkv + xb + p.v_off + (int64_t)ck * LS * p.x_s, p.x_s, LS, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    for (int u = 0; u < NSB; ++u) {
      h16x8 kf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          kf[t][kk] = *(const h16x8*)(smem + K_OFF + swz_off((2 * u + t) * 16 + fr, kk * 4 + fq));
      // V fragments (B operand, k = sequence): hardware-transposed reads of the row-major [s][d] image
      Frag vf[4];
      {
        const int p4 = fr & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int row = u * 32 + half * 16 + 4 * fq + (fr >> 2);
            const int off = V_OFF + swz_off(row, i * 2 + (p4 >> 1)) + (p4 & 1) * 8;
            const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
            vf[i].h[half] = __builtin_bit_cast(uint2, r);
          }
        }
      }
      // accumulator init carries the additive terms: +eps (ReLU kernel) or -(diag + max) (softmax kernel, log2 units)
      f32x4 init[2] = {epsv, epsv};
      if constexpr (SOFTMAX) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float4 d4 = *(const float4*)(smem + DIAG_OFF + ((2 * u + t) * 16 + 4 * fq) * 4);
          init[t] = (f32x4){-(d4.x + gmax), -(d4.y + gmax), -(d4.z + gmax), -(d4.w + gmax)};
        }
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < nm) {
          f32x4 a[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            a[t] = rf_mfma16(kf[t][0], pf[j][0], init[t], 0, 0, 0);
            a[t] = rf_mfma16(kf[t][1], pf[j][1], a[t], 0, 0, 0);
          }
          float f[2][4];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (SOFTMAX)
                f[t][r] = __builtin_amdgcn_exp2f(a[t][r]) + p.eps;
              else
                f[t][r] = fmaxf(a[t][r], p.eps);  // relu(x) + eps with eps folded into the accumulator
            }
          // only feature tile 16 (m = 256..271) holds padded features: mask there, nowhere else (wave-uniform branch)
          if ((m0t + j) == FV_MT - 1 && fr >= FV_M - 16 * (FV_MT - 1)) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) f[t][r] = 0.f;
          }
          Frag kfr;
          kfr.u[0] = pack2(f[0][0], f[0][1]);
          kfr.u[1] = pack2(f[0][2], f[0][3]);
          kfr.u[2] = pack2(f[1][0], f[1][1]);
          kfr.u[3] = pack2(f[1][2], f[1][3]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            ctx[j][i] = rf_mfma16(kfr.v, vf[i].v, ctx[j][i], 0, 0, 0);
          ctx[j][4] = rf_mfma16(kfr.v, ones.v, ctx[j][4], 0, 0, 0);
        }
      }
    }
    }  // chunks of K / V
    // publish ctx^T (bf16), rows 0..63 = values, row 64 = k' sums
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j < nm) {
#pragma unroll
        for (int i = 0; i < FV_DT; ++i) {
          uint2 w;
          w.x = pack2(FV_CS(ctx[j][i][0]), FV_CS(ctx[j][i][1]));
          w.y = pack2(FV_CS(ctx[j][i][2]), FV_CS(ctx[j][i][3]));
          *(uint2*)(smem + CTX_OFF + (i * 16 + fr) * FV_CTX_LD + ((m0t + j) * 16 + 4 * fq) * 2) = w;
        }
      }
    }
    __syncthreads();  // ctx^T visible; K and V tiles are free again

    // prefetch the next item's K and V while phase B runs.  The Q loads issued before phase A are drained first so
    // the compiler's wait at their first use cannot turn into a wait for these DMAs.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
      const int nxt = item + gridDim.x;
      if (nxt < p.nitems) {  // (with nch > 1 K/V hold the LAST chunk of this item: also free now)
        int64_t xb2, ob2;
        item_base(nxt, xb2, ob2);
        fv_load_tile(smem, K_OFF, p.qkv + xb2 + p.k_off, p.x_s, LS, wave, lane);
        fv_load_tile(smem, V_OFF, p.qkv + xb2 + p.v_off, p.x_s, LS, wave, lane);
      }
    }

    // ---------------- phase B ----------------
    f32x4 qinit[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t) qinit[t] = epsv;
    if constexpr (SOFTMAX) {
      float dq[ST], rmax[ST];
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        float a = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = h2f((h16_t)qf[t][kk][e]);
            a = fmaf(f, f, a);
          }
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        dq[t] = a * (0.0625f * 1.4426950408889634f);
        rmax[t] = -INFINITY;
      }
      // pass 0: per-row max of the query logits over m < 266
      for (int j = 0; j < FV_MT; ++j) {
        h16x8 pfr[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) pfr[kk] = *(const h16x8*)(smem + PC_OFF + swz_off(j * 16 + fr, kk * 4 + fq));
#pragma unroll
        for (int t = 0; t < ST; ++t) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          a = rf_mfma16(pfr[0], qf[t][0], a, 0, 0, 0);
          a = rf_mfma16(pfr[1], qf[t][1], a, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j * 16 + 4 * fq + r < FV_M) rmax[t] = fmaxf(rmax[t], a[r]);
        }
      }
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        rmax[t] = fmaxf(rmax[t], __shfl_xor(rmax[t], 16, 64));
        rmax[t] = fmaxf(rmax[t], __shfl_xor(rmax[t], 32, 64));
        const float off = -(dq[t] + rmax[t]);
        qinit[t] = (f32x4){off, off, off, off};
      }
    }
    for (int qc = 0; qc < nch; ++qc) {
    if (qc > 0) load_q(qc);
    f32x4 num[FV_DT][ST];
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int i = 0; i < FV_DT; ++i) num[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one m-block (two feature tiles 2u, 2u+1): q' features -> B fragments -> num += ctx^T q'
    auto mblock = [&](int u, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;  // u == 8: tile 16 is partly padded, tile 17 is all padding
      float f[2][ST][4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if (LAST && jj == 1) {
#pragma unroll
          for (int t = 0; t < ST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) f[jj][t][r] = 0.f;
        } else {
          const int j = 2 * u + jj;
          h16x8 pfr[2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) pfr[kk] = *(const h16x8*)(smem + PC_OFF + swz_off(j * 16 + fr, kk * 4 + fq));
#pragma unroll
          for (int t = 0; t < ST; ++t) {
            f32x4 a = rf_mfma16(pfr[0], qf[t][0], qinit[t], 0, 0, 0);
            a = rf_mfma16(pfr[1], qf[t][1], a, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x;
              if constexpr (SOFTMAX)
                x = __builtin_amdgcn_exp2f(a[r]) + p.eps;
              else
                x = fmaxf(a[r], p.eps);
              if (LAST && 4 * fq + r >= FV_M - 16 * (FV_MT - 1)) x = 0.f;
              f[jj][t][r] = x;
            }
          }
        }
      }
      Frag cf[FV_DT];
#pragma unroll
      for (int i = 0; i < FV_DT; ++i) {
        const char* base = smem + CTX_OFF + (i * 16 + fr) * FV_CTX_LD + (32 * u + 4 * fq) * 2;
        cf[i].h[0] = *(const uint2*)base;
        cf[i].h[1] = *(const uint2*)(base + 32);
      }
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        Frag qfr;
        qfr.u[0] = pack2(f[0][t][0], f[0][t][1]);
        qfr.u[1] = pack2(f[0][t][2], f[0][t][3]);
        qfr.u[2] = pack2(f[1][t][0], f[1][t][1]);
        qfr.u[3] = pack2(f[1][t][2], f[1][t][3]);
#pragma unroll
        for (int i = 0; i < FV_DT; ++i)
          num[i][t] = rf_mfma16(cf[i].v, qfr.v, num[i][t], 0, 0, 0);
      }
    };
    for (int u = 0; u < (FV_MT - 1) / 2; ++u) mblock(u, std::false_type{});
    mblock((FV_MT - 1) / 2, std::true_type{});
    // epilogue: out[s][h*64 + d] = num[d][s] / num[64][s]; lane owns 4 consecutive d of row s
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      const float dn = __shfl(fv_mfma_done(num[4][t][0]), fr, 64);  // row d = 64 lives in register 0 of the fq == 0 lanes
      const float inv = 1.f / dn;
      const int s = qc * LS + (wave * ST + t) * 16 + fr;
      h16_t* orow = p.out + ob + (int64_t)s * p.o_s;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 w;
        w.x = pack2(num[i][t][0] * inv, num[i][t][1] * inv);
        w.y = pack2(num[i][t][2] * inv, num[i][t][3] * inv);
        *(uint2*)(orow + i * 16 + 4 * fq) = w;
      }
    }
    }  // chunks of Q
  }
}

// global -> LDS DMA issued from inline assembly: hipcc's wait-count pass then knows nothing of it.  (With the builtin it
// tracks "LDS written by DMA" and puts s_waitcnt vmcnt(0) before every later ds_read_tr / ds_write / merged ds_read2 it
// cannot disambiguate -- those waits also drain the younger Q loads and output stores.)  All waits for these DMAs are the
// counted s_waitcnt vmcnt(N) written out in the kernel.  lds_addr: wave-uniform LDS byte address of the wave's 1 KB slot.
#pragma clang diagnostic ignored "-Winline-asm"  // m0 on the clobber list is intended: no other M0 user in that kernel
__device__ __forceinline__ void fv_glds_asm(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}

// workgroup barrier that publishes this wave's LDS writes and leaves its global loads / stores / DMAs in flight
// (__syncthreads() carries a release fence that hipcc lowers to s_waitcnt vmcnt(0): it would drain the K/V prefetch, the
// Q loads of the running item and the output stores of the previous one at every barrier)
__device__ __forceinline__ void fv_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// phase timing of the 8-wave kernel (RF_FAVOR_DBG bit 8; tools/favor_bench.py reads it through rf_favor_phase_cycles):
// s_memtime cycles of wave 0 of every workgroup, summed: [0] wait for K/V, [1] phase A, [2] publish + barrier, [3] prefetch issue + combine, [4] phase B, [5] stores, [6] items
__device__ unsigned long long g_fv_cycles[8];
#define FV_STAMP(slot)                                                   \
  if (prof) {                                                            \
    const unsigned long long now = __builtin_readcyclecounter();         \
    acc_cyc[slot] += now - t_last;                                       \
    t_last = now;                                                        \
  }

// ------------------------------------------------------------------------------------------------------------------
// 8-wave variant (2 waves per SIMD: a lone wave issues one VALU op per 4 cycles, two co-resident waves one per 2, and
// the feature maps are VALU-bound).  Phase A splits the 17 FEATURE tiles over the eight waves (3,2,2,2,2,2,2,2: every wave
// walks the whole sequence for its own tiles with its Pc fragments resident in registers) -- round 3; rounds 1-2 split
// (feature group 5,4,4,4) x (sequence half) and combined the halves' partial contexts through the ctx^T image, which cost a
// publish, a barrier, a read-add-republish and a second barrier per item (6.2k of an item's 23.7k cycles) for a phase-A
// critical path of 4.5 tile-sequences per SIMD instead of 5.  Phase B: the sequence is split over all eight waves.  Same
// maths, operands and LDS image as above.
// ------------------------------------------------------------------------------------------------------------------
template <int LS, bool SOFTMAX>
__global__ __launch_bounds__(512, 1) void favor_attention_kernel8(const FavorAttnP p) {
  constexpr int NSB = LS / 32;                    // s-blocks (pairs of s-tiles) per chunk
  constexpr int NJ = 3;                           // feature tiles of a wave in phase A (wave 0: 3, the others 2)
  // s-tiles per wave in phase B: one up to LS = 128 (all eight waves run phase B at LS = 128), two at LS = 256.  Round 2
  // shipped two from LS = 128 up -- only the four older waves ran phase B there, ~13 % slower -- because the one-tile form was
  // not run-to-run reproducible.  Round 3 found the cause (an MFMA result read too early by the denominator shuffle, see
  // fv_mfma_done above) and fixed THAT; the split is the fast one again.
  constexpr int STB = LS >= 128 ? LS / 128 : 1;
  constexpr int NWB = LS / (16 * STB);            // waves active in phase B (8 at LS = 256, 4 at LS = 128 and 64)
  constexpr int PC_OFF = 0;
  constexpr int K_OFF = FV_MPAD * 128;
  constexpr int V_OFF = K_OFF + LS * 128;
  constexpr int CTX_OFF = V_OFF + LS * 128;
  constexpr int DIAG_OFF = CTX_OFF + FV_DROWS * FV_CTX_LD;
  constexpr int RED_OFF = DIAG_OFF + LS * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
  const int fr = lane & 15, fq = lane >> 4;
  // this wave's feature tiles in phase A: m0t .. m0t + nm - 1 (waves w and w + 4 share a SIMD: 5 + 4 + 4 + 4 tiles per SIMD)
  const int nm = wave == 0 ? 3 : 2;
  const int m0t = wave == 0 ? 0 : 1 + 2 * wave;
  const f32x4 epsv = {p.eps, p.eps, p.eps, p.eps};
  Frag ones;
#pragma unroll
  for (int k = 0; k < 4; ++k) ones.u[k] = fr == 0 ? RF_H16_ONE2 : 0u;

  // [nrows][64] bf16 tile -> swizzled LDS image; a wave's instruction covers 8 rows (row = 8*it + lane/8, rows & 7 == lane/8)
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto load_tile8 = [&](int lds_off, const h16_t* gp, int64_t stride, auto nrows_tag) {
    constexpr int NI = decltype(nrows_tag)::value * 8 / 64;
    int ln = lane;  // opaque: the per-lane source offsets are rebuilt at every call instead of living in (spilled) registers
    asm volatile("" : "+v"(ln));
    const int dma_row = ln >> 3;
    const int dma_col = ((ln & 7) ^ dma_row) * 8;
    const h16_t* g0 = gp + (int64_t)dma_row * stride + dma_col;
#pragma unroll
    for (int k = 0; k < (NI + 7) / 8; ++k) {
      const int it = wave + 8 * k;
      if (NI % 8 == 0 || it < NI)
        fv_glds_asm(g0 + (int64_t)(8 * it) * stride, __builtin_amdgcn_readfirstlane(lds_base + lds_off + it * 1024));
    }
  };
  constexpr std::integral_constant<int, LS> LS_TAG{};
  load_tile8(PC_OFF, p.pc, FV_DH, std::integral_constant<int, FV_MPAD>{});
  for (int i = tid; i < FV_DROWS * FV_CTX_LD / 4; i += 512) ((unsigned*)(smem + CTX_OFF))[i] = 0u;

  auto item_base = [&](int item, int64_t& xb, int64_t& ob) {
    const int h = item % p.n_h;
    const int t = item / p.n_h;
    const int o = t % p.n_o, b = t / p.n_o;
    xb = (int64_t)b * p.x_b + (int64_t)o * p.x_o + (int64_t)h * p.x_h;
    ob = (int64_t)b * p.o_b + (int64_t)o * p.o_o + h * FV_DH;
  };
  auto ctx_col = [](int tile, int q4) { return ((tile >> 1) * 32 + 8 * q4 + 4 * (tile & 1)) * 2; };
  const int nch = SOFTMAX ? 1 : p.nchunks;  // (the softmax kernel needs the global key max first: whole sequences only, rf_favor_attention checks)
  int item = blockIdx.x;
  if (item < p.nitems) {
    int64_t xb, ob;
    item_base(item, xb, ob);
    load_tile8(K_OFF, p.qkv + xb + p.k_off, p.x_s, LS_TAG);
    load_tile8(V_OFF, p.qkv + xb + p.v_off, p.x_s, LS_TAG);
  }
  bool first = true;
  const bool prof = (RF_DBG(p.dbg) & 8) && wave == ((RF_DBG(p.dbg) >> 4) & 7);  // (false at compile time outside the ablation build)
  unsigned long long acc_cyc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = prof ? __builtin_readcyclecounter() : 0;
  for (; item < p.nitems; item += gridDim.x) {
    int64_t xb, ob;
    item_base(item, xb, ob);
    // The item's base offsets live in VECTOR registers from here on (opaque copies): as wave-uniform scalars they and the
    // addresses derived from them stayed in SGPRs across both phases, and the softmax variants -- the most scalar-hungry
    // ones -- spilled 16-19 SGPRs into VGPR lanes (v_writelane / v_readlane).  The kernel has vector registers to spare.
    asm volatile("" : "+v"(xb), "+v"(ob));
    if (prof) {
      t_last = __builtin_readcyclecounter();
      acc_cyc[6] += 1;
    }
    if (first || wave >= NWB)  // (waves idle in phase B issued no stores: their youngest operations are the DMAs)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STB * 4) : "memory");
    first = false;
    fv_lds_barrier();
    FV_STAMP(0)
    // the Pc fragments of this wave's feature tiles, resident through phase A (swz_off(16 T + fr, c) = 2048 T + swz_off(fr, c))
    h16x8 pf[NJ][2];
    {
      int frP = fr, fqP = fq;
      asm volatile("" : "+v"(frP), "+v"(fqP));
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          pf[j][kk] = *(const h16x8*)(smem + PC_OFF + m0t * 2048 + swz_off(frP, kk * 4 + fqP) + (j < nm ? j : 0) * 2048);
    }
    h16x8 qf[STB][2];
    auto load_q = [&](int chunk) {
      int frq = fr, fqq = fq;  // opaque copies: keeps the row / chunk offsets out of the kernel-long live ranges
      asm volatile("" : "+v"(frq), "+v"(fqq));
      if (wave < NWB) {
#pragma unroll
        for (int t = 0; t < STB; ++t) {
          const int s = chunk * LS + (wave * STB + t) * 16 + frq;
          const h16_t* qrow = p.qkv + xb + p.q_off + (int64_t)s * p.x_s;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) qf[t][kk] = *(const h16x8*)(qrow + (kk * 4 + fqq) * 8);
        }
      }
    };
    load_q(0);

    float gmax = 0.f;
    if constexpr (SOFTMAX) {
      int frS = fr, fqS = fq;
      asm volatile("" : "+v"(frS), "+v"(fqS));
      for (int s = tid; s < LS; s += 512) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const h16x8 x = *(const h16x8*)(smem + K_OFF + s * 128 + c * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = h2f((h16_t)x[e]);
            a = fmaf(f, f, a);
          }
        }
        ((float*)(smem + DIAG_OFF))[s] = a * (0.0625f * 1.4426950408889634f);
      }
      // pass 0: global max of the key logits (this wave: its feature tiles x the whole sequence)
      float mx = -INFINITY;
      for (int u = 0; u < NSB; ++u) {
        h16x8 kf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            kf[t][kk] = *(const h16x8*)(smem + K_OFF + swz_off((2 * u + t) * 16 + frS, kk * 4 + fqS));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (j < nm) {
            const h16x8* pfj = pf[j];
            // only feature tile 16 (m = 256..271) holds padded features (a per-tile scalar threshold would live in an SGPR for
            // the whole kernel: the softmax variants spilled them into VGPR lanes)
            const bool valid = (m0t + j) != FV_MT - 1 || frS < FV_M - 16 * (FV_MT - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              f32x4 a = {0.f, 0.f, 0.f, 0.f};
              a = rf_mfma16(kf[t][0], pfj[0], a, 0, 0, 0);
              a = rf_mfma16(kf[t][1], pfj[1], a, 0, 0, 0);
              if (valid) mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
            }
          }
        }
      }
      mx = wave_max(mx);
      if (lane == 0) ((float*)(smem + RED_OFF))[wave] = mx;
      fv_lds_barrier();
      const float* red = (const float*)(smem + RED_OFF);
      gmax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
    }

    // ---------------- phase A ----------------
    f32x4 ctx[NJ][FV_DT];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < FV_DT; ++i) ctx[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ck = 0; ck < nch; ++ck) {
      if (ck > 0) {
        __syncthreads();
        load_tile8(K_OFF, p.qkv + xb + p.k_off + (int64_t)ck * LS * p.x_s, p.x_s, LS_TAG);
        load_tile8(V_OFF, p.qkv + xb + p.v_off + (int64_t)ck * LS * p.x_s, p.x_s, LS_TAG);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      // the K fragments and transposed V fragments of s-block u + 1 are read while s-block u is computed (a wave owns only 2-3
      // feature tiles: without the read-ahead every s-block starts with an exposed LDS round trip)
      h16x8 kf[2][2];
      Frag vf[4];
      auto load_sblock = [&](int u, h16x8 (&kfo)[2][2], Frag (&vfo)[4]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            kfo[t][kk] = *(const h16x8*)(smem + K_OFF + swz_off((2 * u + t) * 16 + fr, kk * 4 + fq));
        const int p4 = fr & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int row = u * 32 + half * 16 + 4 * fq + (fr >> 2);
            const int off = V_OFF + swz_off(row, i * 2 + (p4 >> 1)) + (p4 & 1) * 8;
            const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
            vfo[i].h[half] = __builtin_bit_cast(uint2, r);
          }
        }
      };
      load_sblock(0, kf, vf);
      for (int u = 0; u < ((RF_DBG(p.dbg) & 1) ? 0 : NSB); ++u) {
        h16x8 kfn[2][2];
        Frag vfn[4];
        load_sblock(u + 1 < NSB ? u + 1 : u, kfn, vfn);
        f32x4 init[2] = {epsv, epsv};
        if constexpr (SOFTMAX) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float4 d4 = *(const float4*)(smem + DIAG_OFF + ((2 * u + t) * 16 + 4 * fq) * 4);
            init[t] = (f32x4){-(d4.x + gmax), -(d4.y + gmax), -(d4.z + gmax), -(d4.w + gmax)};
          }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (j < nm) {
            const h16x8* pfj = pf[j];  // resident (24 registers: the accumulators shrank from 5 to 3 tiles)
            f32x4 a[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              a[t] = rf_mfma16(kf[t][0], pfj[0], init[t], 0, 0, 0);
              a[t] = rf_mfma16(kf[t][1], pfj[1], a[t], 0, 0, 0);
            }
            float f[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if constexpr (SOFTMAX)
                  f[t][r] = __builtin_amdgcn_exp2f(a[t][r]) + p.eps;
                else
                  f[t][r] = fmaxf(a[t][r], p.eps);
              }
            if ((m0t + j) == FV_MT - 1 && fr >= FV_M - 16 * (FV_MT - 1)) {
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) f[t][r] = 0.f;
            }
            Frag kfr;
            kfr.u[0] = pack2(f[0][0], f[0][1]);
            kfr.u[1] = pack2(f[0][2], f[0][3]);
            kfr.u[2] = pack2(f[1][0], f[1][1]);
            kfr.u[3] = pack2(f[1][2], f[1][3]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              ctx[j][i] = rf_mfma16(kfr.v, vf[i].v, ctx[j][i], 0, 0, 0);
            ctx[j][4] = rf_mfma16(kfr.v, ones.v, ctx[j][4], 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) kf[t][kk] = kfn[t][kk];
#pragma unroll
        for (int i = 0; i < 4; ++i) vf[i] = vfn[i];
      }
    }
    FV_STAMP(1)
    // publish ctx^T (16-bit), rows 0..63 = values, row 64 = k' sums: every wave its own feature tiles, complete
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (j < nm) {
#pragma unroll
        for (int i = 0; i < FV_DT; ++i) {
          uint2 w;
          w.x = pack2(FV_CS(ctx[j][i][0]), FV_CS(ctx[j][i][1]));
          w.y = pack2(FV_CS(ctx[j][i][2]), FV_CS(ctx[j][i][3]));
          *(uint2*)(smem + CTX_OFF + (i * 16 + fr) * FV_CTX_LD + ctx_col(m0t + j, fq)) = w;
        }
      }
    fv_lds_barrier();  // every wave is through phase A: ctx^T is complete, the K and V tiles are free again
    FV_STAMP(2)
    // K/V prefetch of the next item, in flight across the combine and phase B.  (Round 2 had moved it behind the combine's
    // barrier because that made the then unexplained wrong rows rarer: it only shifted the timing of the early MFMA-result
    // read, see fv_mfma_done.)  Pin the Q fragments first (loaded a whole phase ago): hipcc's own wait for them lands here,
    // BEFORE the prefetch DMAs are issued; left to itself it waits at their first use in phase B with vmcnt(0), which also
    // drains the K/V prefetch it cannot count (in-order counter) and serialises the item pipeline.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < STB; ++t) {
      asm volatile("" : "+v"(qf[t][0]), "+v"(qf[t][1]));
    }
    {
      const int nxt = item + gridDim.x;
      if (nxt < p.nitems) {
        int64_t xb2, ob2;
        item_base(nxt, xb2, ob2);
        load_tile8(K_OFF, p.qkv + xb2 + p.k_off, p.x_s, LS_TAG);
        load_tile8(V_OFF, p.qkv + xb2 + p.v_off, p.x_s, LS_TAG);
      }
    }
    FV_STAMP(3)
    // ---------------- phase B ----------------
    for (int qc = 0; qc < nch; ++qc) {
      if (qc > 0) {
        load_q(qc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < STB; ++t) {
          asm volatile("" : "+v"(qf[t][0]), "+v"(qf[t][1]));
        }
      }
      if (wave < NWB) {
        f32x4 qinit[STB];
#pragma unroll
        for (int t = 0; t < STB; ++t) qinit[t] = epsv;
        if constexpr (SOFTMAX) {
          float dq[STB], rmax[STB];
#pragma unroll
          for (int t = 0; t < STB; ++t) {
            float a = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float f = h2f((h16_t)qf[t][kk][e]);
                a = fmaf(f, f, a);
              }
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            dq[t] = a * (0.0625f * 1.4426950408889634f);
            rmax[t] = -INFINITY;
          }
          for (int j = 0; j < FV_MT; ++j) {
            h16x8 pfr[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) pfr[kk] = *(const h16x8*)(smem + PC_OFF + swz_off(j * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int t = 0; t < STB; ++t) {
              f32x4 a = {0.f, 0.f, 0.f, 0.f};
              a = rf_mfma16(pfr[0], qf[t][0], a, 0, 0, 0);
              a = rf_mfma16(pfr[1], qf[t][1], a, 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (j * 16 + 4 * fq + r < FV_M) rmax[t] = fmaxf(rmax[t], a[r]);
            }
          }
#pragma unroll
          for (int t = 0; t < STB; ++t) {
            rmax[t] = fmaxf(rmax[t], __shfl_xor(rmax[t], 16, 64));
            rmax[t] = fmaxf(rmax[t], __shfl_xor(rmax[t], 32, 64));
            const float off = -(dq[t] + rmax[t]);
            qinit[t] = (f32x4){off, off, off, off};
          }
        }
        f32x4 num[FV_DT][STB];
#pragma unroll
        for (int t = 0; t < STB; ++t)
#pragma unroll
          for (int i = 0; i < FV_DT; ++i) num[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto mblock = [&](int u, auto last_tag) {
          constexpr bool LAST = decltype(last_tag)::value;
          float f[2][STB][4];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            if (LAST && jj == 1) {
#pragma unroll
              for (int t = 0; t < STB; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) f[jj][t][r] = 0.f;
            } else {
              const int j = 2 * u + jj;
              h16x8 pfr[2];
#pragma unroll
              for (int kk = 0; kk < 2; ++kk) pfr[kk] = *(const h16x8*)(smem + PC_OFF + swz_off(j * 16 + fr, kk * 4 + fq));
#pragma unroll
              for (int t = 0; t < STB; ++t) {
                f32x4 a = rf_mfma16(pfr[0], qf[t][0], qinit[t], 0, 0, 0);
                a = rf_mfma16(pfr[1], qf[t][1], a, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float x;
                  if constexpr (SOFTMAX)
                    x = __builtin_amdgcn_exp2f(a[r]) + p.eps;
                  else
                    x = fmaxf(a[r], p.eps);
                  if (LAST && 4 * fq + r >= FV_M - 16 * (FV_MT - 1)) x = 0.f;
                  f[jj][t][r] = x;
                }
              }
            }
          }
          Frag cf[FV_DT];  // one 16-byte read per fragment: the image keeps a lane's 8 k-slots (2 tiles x 4 rows) adjacent
#pragma unroll
          for (int i = 0; i < FV_DT; ++i)
            cf[i].v = *(const h16x8*)(smem + CTX_OFF + (i * 16 + fr) * FV_CTX_LD + (32 * u + 8 * fq) * 2);
#pragma unroll
          for (int t = 0; t < STB; ++t) {
            Frag qfr;
            qfr.u[0] = pack2(f[0][t][0], f[0][t][1]);
            qfr.u[1] = pack2(f[0][t][2], f[0][t][3]);
            qfr.u[2] = pack2(f[1][t][0], f[1][t][1]);
            qfr.u[3] = pack2(f[1][t][2], f[1][t][3]);
#pragma unroll
            for (int i = 0; i < FV_DT; ++i)
              num[i][t] = rf_mfma16(cf[i].v, qfr.v, num[i][t], 0, 0, 0);
          }
        };
        for (int u = 0; u < ((RF_DBG(p.dbg) & 2) ? 0 : (FV_MT - 1) / 2); ++u) mblock(u, std::false_type{});
        mblock((FV_MT - 1) / 2, std::true_type{});
        FV_STAMP(4)
#pragma unroll
        for (int t = 0; t < STB; ++t) {
          const float dn = __shfl(fv_mfma_done(num[4][t][0]), fr, 64);
          const float inv = 1.f / dn;
          const int s = qc * LS + (wave * STB + t) * 16 + fr;
          h16_t* orow = p.out + ob + (int64_t)s * p.o_s;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            uint2 w;
            w.x = pack2(num[i][t][0] * inv, num[i][t][1] * inv);
            w.y = pack2(num[i][t][2] * inv, num[i][t][3] * inv);
            *(uint2*)(orow + i * 16 + 4 * fq) = w;
          }
        }
      }
    }
    FV_STAMP(5)
  }
  if (prof && lane == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) atomicAdd(&g_fv_cycles[i], acc_cyc[i]);
  }
}

extern "C" int rf_favor_phase_cycles(unsigned long long* out7, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (out7 && hipMemcpyFromSymbol(out7, HIP_SYMBOL(g_fv_cycles), 7 * sizeof(unsigned long long)) != hipSuccess) return 1000 + (int)hipGetLastError();
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fv_cycles), z, sizeof(z)) != hipSuccess) return 1000 + (int)hipGetLastError();
  return 0;
}

template <int LS, bool SM>
static int launch_favor(const FavorAttnP& p, hipStream_t s) {
  const size_t lds = (size_t)FV_MPAD * 128 + 2 * (size_t)LS * 128 + (size_t)FV_DROWS * FV_CTX_LD + LS * 4 + 64;
  const int ncu = rf_num_cus() > 0 ? rf_num_cus() : 256;
  const int grid = p.nitems < ncu ? p.nitems : ncu;
  // measured (tools/favor_bench.py): the 8-wave kernel wins for the ReLU features (no AGPR traffic at <= 256
  // registers) and for the softmax features up to 128-row sequences (674 vs 720 us on the MSA-column shape; 9 spilled
  // registers); at 256 rows the softmax variant does not fit and stays on the 4-wave kernel.  RF_FAVOR4 forces the 4-wave kernel.
  static const bool force4 = getenv("RF_FAVOR4") != nullptr;
  constexpr bool only4 = SM && LS > 128;  // (the 8-wave form of this variant does not fit 256 registers: never instantiated)
  if (only4 || force4) {
    auto k = favor_attention_kernel<LS, SM>;
    if (const int e = rf_enable_big_lds<favor_attention_kernel<LS, SM>>()) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
  } else if constexpr (!only4) {
    auto k = favor_attention_kernel8<LS, SM>;
    if (const int e = rf_enable_big_lds<favor_attention_kernel8<LS, SM>>()) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, p);
  }
  return rf_launch_status();
}

extern "C" int rf_favor_attention(const void* qkv, const void* pc, void* out, const int64_t x_strides[4],
                                  const int64_t o_strides[3], int q_off, int k_off, int v_off, int n_b, int n_o,
                                  int n_h, int seq_len, int dim_head, int n_features, int softmax_kernel, float eps,
                                  void* stream) {
  if (dim_head != FV_DH || n_features != FV_M) return RF_EINVAL;
  int nchunks = 1, ls = seq_len;
  if (seq_len > 256) {  // long sequences: 256-row chunks (ReLU kernel; the softmax kernel needs the global key max first)
    if (softmax_kernel || seq_len % 256) return RF_EINVAL;
    nchunks = seq_len / 256;
    ls = 256;
  }
  if (ls != 64 && ls != 128 && ls != 256) return RF_EINVAL;
  if (((uintptr_t)qkv % 16) || ((uintptr_t)pc % 16) || ((uintptr_t)out % 8)) return RF_EALIGN;
  for (int i = 0; i < 3; ++i)
    if (x_strides[i] % 8 || o_strides[i] % 4) return RF_EALIGN;
  if (x_strides[3] % 8) return RF_EALIGN;
  if (q_off % 8 || k_off % 8 || v_off % 8) return RF_EALIGN;
  FavorAttnP p;
  p.qkv = (const h16_t*)qkv;
  p.pc = (const h16_t*)pc;
  p.out = (h16_t*)out;
  p.x_b = x_strides[0]; p.x_o = x_strides[1]; p.x_s = x_strides[2]; p.x_h = x_strides[3];
  p.o_b = o_strides[0]; p.o_o = o_strides[1]; p.o_s = o_strides[2];
  p.q_off = q_off; p.k_off = k_off; p.v_off = v_off;
  p.n_o = n_o; p.n_h = n_h;
  p.nitems = n_b * n_o * n_h;
  p.eps = eps;
  p.nchunks = nchunks;
  p.ctx_scale = 1.0f;
#ifdef RF_H16_IS_F16
  for (int n = 1; n < seq_len; n <<= 1) p.ctx_scale *= 0.5f;
#endif
  static int dbg = 0;
  static const int dbg_rc = rf_dbg_env("RF_FAVOR_DBG", &dbg);
  if (dbg_rc) return dbg_rc;
  p.dbg = dbg;
  hipStream_t s = (hipStream_t)stream;
  if (ls == 256) return softmax_kernel ? launch_favor<256, true>(p, s) : launch_favor<256, false>(p, s);
  if (ls == 128) return softmax_kernel ? launch_favor<128, true>(p, s) : launch_favor<128, false>(p, s);
  return softmax_kernel ? launch_favor<64, true>(p, s) : launch_favor<64, false>(p, s);
}
