// Tied MSA-row attention, first half (SoftTiedAttentionOverResidues, rf.py:252-255,261-265) for gfx950 (MI355X):
//
//   logits[b,h,i,j] = sum_{n,d} q[b,n,i,h,d] * k[b,n,j,h,d]        (contraction over N * d_head = 4096 at config 2)
//   att[b,h,i,:]    = softmax_j(logits[b,h,i,:])                    (q already carries the position weights and d_h^-0.5)
//
// As a batched GEMM through the generic engine this contraction ran at 215 TFLOP/s whatever the tile shape: its operand
// rows are 64-byte head slices (32 bf16) scattered 2.3 KB apart, so the two-stage DMA pipeline of the engine is bound by
// the gather latency, not by the matrix pipe.  Here one 4-wave workgroup owns 64 query rows x all L key columns of one
// (b, h): the (n)-steps stream through a 6..8-stage LDS ring (global_load_lds, counted vmcnt, five DMA instructions per
// wave and step, ~100 KB in flight per CU), the 64 x L logits stay in registers (one wave = 16 complete rows), the row
// softmax is wave-local (two shuffles across the lane quads), and the probabilities leave through a wave-private LDS strip
// as whole 512-byte rows.  The fp32 logits tensor and the separate softmax launch disappear.
#include "common.h"

static __device__ __attribute__((aligned(16))) unsigned int g_tied_zero16[4];

struct TiedP {
  const bf16_t* q;
  const bf16_t* k;
  int64_t b_stride, n_stride, l_stride;  // elements
  bf16_t* att;                           // [B, H, L, L]
  int B, H, N;
};

__device__ __forceinline__ void tied_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int L>
__global__ __launch_bounds__(256) void tied_logits_kernel(const TiedP p) {
  constexpr int JT = L / 16;                 // key tiles of 16 columns
  constexpr int NSTG = L >= 256 ? 6 : 8;     // ring stages
  constexpr int Q_BYTES = 64 * 64, K_BYTES = L * 64, STAGE = Q_BYTES + K_BYTES;
  constexpr int QI = 4, KI = L / 16;         // DMA instructions (16 rows x 64 B) per stage
  constexpr int PW = (QI + KI + 3) / 4;      // per wave, uniform (padded with dummies into DUMP)
  constexpr int DUMP = NSTG * STAGE;
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
  const bf16_t* qb = p.q + (int64_t)b * p.b_stride + h * 32 + (int64_t)(it * 64) * p.l_stride;
  const bf16_t* kb = p.k + (int64_t)b * p.b_stride + h * 32;

  // DMA: an instruction covers 16 rows x 4 chunks (64-byte head slices); lane-linear LDS image with the bank swizzle
  // chunk ^ g((row >> 2) & 3), g = {0, 2, 3, 1}, on the source chunk and on the fragment reads
  const int lrow = lane >> 2;
  const int c_log = (lane & 3) ^ ((0x78 >> (((lrow >> 2) & 3) * 2)) & 3);
  const int64_t lane_off = (int64_t)lrow * p.l_stride + c_log * 8;
  auto stage = [&](int n) {
    char* st = smem + (n % NSTG) * STAGE;
    const bool live = n < p.N;
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

  f32x4 acc[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int gq = (0x78 >> (((fr >> 2) & 3) * 2)) & 3;
  const int q_rd = (wave * 16 + fr) * 64 + ((fq ^ gq) * 16);
  const int k_rd = Q_BYTES + fr * 64 + ((fq ^ gq) * 16);

#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) stage(s);
  for (int n = 0; n < p.N; ++n) {
    // step n landed once at most the NSTG-2 younger steps are outstanding (this wave's share; the barrier covers the rest).
    // (Double-buffering the fragments in registers instead of the extra stage of DMA cover measured slower: 97 vs 84 us;
    // the loop is bound by the 64-byte-granule gather, ~15 B/clk/CU through the texture path.)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSTG - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(n + NSTG - 1);  // into the buffer of step n-1, whose fragments every wave has consumed
    const char* st = smem + (n % NSTG) * STAGE;
    const bf16x8 qf = *(const bf16x8*)(st + q_rd);
    bf16x8 kf[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) kf[j] = *(const bf16x8*)(st + k_rd + j * 1024);
#pragma unroll
    for (int j = 0; j < JT; ++j)
      // key tile as MFMA-A, query tile as MFMA-B: lane holds logits[i = fr][j = 16*tile + 4*fq .. +3]
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j], qf, acc[j], 0, 0, 0);
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
    w.x = (unsigned)f2bf(acc[j][0] * inv) | ((unsigned)f2bf(acc[j][1] * inv) << 16);
    w.y = (unsigned)f2bf(acc[j][2] * inv) | ((unsigned)f2bf(acc[j][3] * inv) << 16);
    *(uint2*)(strip + fr * PITCH + (j * 16 + 4 * fq) * 2) = w;
  }
  asm volatile("" ::: "memory");
  constexpr int CPR = L * 2 / 16, NCH = 16 * CPR, NIT = NCH / 64;
  bf16_t* arow = p.att + (((int64_t)b * p.H + h) * L + it * 64 + wave * 16) * L;
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    const int idx = lane + 64 * t;
    const int r = idx / CPR, c = idx % CPR;
    const f32x4 v = *(const f32x4*)(strip + r * PITCH + c * 16);
    *(f32x4*)(arow + (int64_t)r * L + c * 8) = v;
  }
}

__global__ __launch_bounds__(256) void tied_att_sym_kernel(const bf16_t* att, float* sym, int64_t sym_ld, int B, int H, int L) {
  // sym[b,i,j,h] = 0.5*(att[b,h,i,j] + att[b,h,j,i])
  const int64_t n = (int64_t)B * L * L * H;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int h = e % H;
    const int64_t t = e / H;
    const int j = t % L, i = (t / L) % L;
    const int64_t b = t / ((int64_t)L * L);
    const int64_t o = (b * H + h) * L;
    sym[t * sym_ld + h] = 0.5f * (bf2f(att[(o + i) * L + j]) + bf2f(att[(o + j) * L + i]));
  }
}

template <int L>
static int launch_tied(const TiedP& p, hipStream_t s) {
  constexpr int NSTG = L >= 256 ? 6 : 8;
  constexpr int LDS = NSTG * (64 * 64 + L * 64) + 1024;
  auto k = tied_logits_kernel<L>;
  static bool once = ((void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  hipLaunchKernelGGL(k, dim3((unsigned)(p.B * p.H * (L / 64))), dim3(256), LDS, s, p);
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
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k;
  p.b_stride = b_stride; p.n_stride = n_stride; p.l_stride = l_stride;
  p.att = (bf16_t*)att; p.B = B; p.H = H; p.N = N;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (L == 256) rc = launch_tied<256>(p, s);
  else if (L == 192) rc = launch_tied<192>(p, s);
  else if (L == 128) rc = launch_tied<128>(p, s);
  else rc = launch_tied<64>(p, s);
  if (rc != 0 || !att_sym) return rc;
  const int64_t n = (int64_t)B * L * L * H;
  unsigned g = (unsigned)((n + 255) / 256);
  if (g > 8192u) g = 8192u;
  hipLaunchKernelGGL(tied_att_sym_kernel, dim3(g), dim3(256), 0, s, (const bf16_t*)att, att_sym, sym_ld, B, H, L);
  return rf_launch_status();
}
