// 3x3 dilated "same" convolution, 288 -> 288 channels, NHWC 16-bit in / out (the ResNet pair refiner: rf.py:452-457,
// resnet.py:19-38) for gfx950 (MI355X): implicit GEMM with the three column taps of a row served from ONE haloed LDS image.
//
// The generic implicit-GEMM path (csrc/gemm.hip, AMODE = conv) fetches the [256 pixels x 64 channels] operand tile once per
// tap: nine reads of every activation byte, and with 151 MB of activations per launch they miss L2 -- 1.28 GB of the 1.43 GB
// a launch moved (PMC), at 0.45 of the matrix pipe.  Here a tile is 256 consecutive pixels of one image row and the K loop runs
//   for row tap di in (-d, 0, +d):  for channel group cg of 32 (nine groups):
//       ONE image [(256 + 2 d) pixels x 32 channels] of input row i + di (zero outside the picture),
//       THREE weight tiles [288 x 32] (column taps -d, 0, +d),
//       3 x 36 MFMAs per wave: the column tap only shifts the image row a fragment is read from (pixel p + dj d).
// 3.2 reads of every activation byte instead of 9, 108 MFMAs per wave and barrier instead of 72, one s_waitcnt / barrier per
// super-step (two 71 KB buffers); the next super-step's 71 DMA pieces are issued right behind the barrier, the 39
// fragment reads of a super-step run six fragments ahead of the MFMAs that consume them (generated schedule).
#include <type_traits>

#include "common.h"

static __device__ __attribute__((aligned(16))) unsigned int g_conv_zero16[4];

struct Conv288P {
  const h16_t* x;      // [B, H, W, 288]
  const h16_t* w;      // [288 out][9 taps][288 in]  (tap = 3 (di + 1) + (dj + 1), K contiguous)
  const float* bias;   // [288] or null
  h16_t* y;            // [B, H, W, 288]
  int B, H, W, dil;
  int tpr;             // tiles per image row (W / 256)
  int nt_store;
};

__device__ __forceinline__ void conv_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(512) void conv3x3_c288_kernel(const Conv288P p) {
  constexpr int C = 288, BM = 256, BN = 288;
  constexpr int WGN = 2, TM = 64, TN = 144, WM = 4, WN = 9;  // 4 x 2 waves, wave tile 64 pixels x 144 channels
  constexpr int A_PIECES = 17, B_PIECES = 18;                  // 1 KB pieces: image rows / 16, weight rows / 16
  constexpr int A_BYTES = A_PIECES * 1024, B_BYTES = B_PIECES * 1024;
  constexpr int SSTEP = A_BYTES + 3 * B_BYTES;                 // 71 KB per super-step
  constexpr int NPIECES = A_PIECES + 3 * B_PIECES;             // 71
  constexpr int PW = (NPIECES + 7) / 8;                        // 9 per wave (one dummy slot)
  constexpr int NSS = 27;                                      // 3 row taps x 9 channel groups
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int fr = lane & 15, fq = lane >> 4;
  const int d = p.dil;

  // XCD-aware remap: consecutive tiles (neighbouring rows of one picture: they share two of their three input rows) on one XCD
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int jt = lid % p.tpr;
  const int row = lid / p.tpr;     // b * H + i
  const int i = row % p.H;
  const int j0 = jt * BM;
  const h16_t* const xrow0 = p.x + (int64_t)(row - i) * p.W * C;  // picture b

  // 64-byte rows (4 slots of 16 bytes): four rows share a 256-byte bank line; slot ^ ((row >> 2) & 3) makes the 16 rows of a
  // fragment read hit 16 distinct 16-byte bank groups whatever the (tap-dependent) first row is
  auto swz = [](int r) { return (r >> 2) & 3; };
  const int dr = lane >> 2, ds = lane & 3;  // row inside a piece / physical slot of this lane's 16 bytes

  // DMA piece t (of PW per wave) of a super-step: pieces 0 .. 16 the haloed image, 17 .. 70 the three weight tiles.  The
  // per-lane byte offsets relative to the super-step's base pointers are fixed for the tile (doff; -1: outside the picture)
  int doff[PW];
#pragma unroll
  for (int t = 0; t < PW; ++t) {
    const int q = t * 8 + wave;
    if (q < A_PIECES) {
      const int r = q * 16 + dr;                   // image row: pixel j0 - d + r
      const int jj = j0 - d + r;
      doff[t] = (jj >= 0 && jj < p.W && r < BM + 2 * d) ? (jj * C + ((ds ^ swz(r)) * 8)) * 2 : -1;
    } else {
      const int qb = q - A_PIECES;
      const int n = (qb % B_PIECES) * 16 + dr;
      doff[t] = ((n * 9 + qb / B_PIECES) * C + ((ds ^ swz(n)) * 8)) * 2;
    }
  }
  auto stage_piece = [&](int ss, int buf, int t, int off) __attribute__((always_inline)) {
    const int di = ss / 9 - 1, cg = ss % 9;
    char* a_lds = smem + buf * SSTEP;
    const int q = t * 8 + wave;
    if (q < A_PIECES) {
      const int ii = i + di * d;
      const bool ok = ii >= 0 && ii < p.H && off >= 0;
      const char* base = (const char*)(xrow0 + (int64_t)ii * p.W * C + cg * 32);
      conv_glds16(ok ? base + off : (const char*)g_conv_zero16, a_lds + q * 1024);
    } else if (q < NPIECES) {
      const char* base = (const char*)(p.w + (di + 1) * 3 * C + cg * 32);
      conv_glds16(base + off, a_lds + q * 1024);   // (piece q of the buffer: the three weight tiles follow the image)
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    f32x4 bc = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bc = *(const f32x4*)(p.bias + wn * TN + j * 16 + 4 * fq);
#pragma unroll
    for (int ii = 0; ii < WM; ++ii) acc[ii][j] = bc;
  }

  // Fragment reads from inline asm, one column tap AHEAD of the MFMAs that use them (two register sets): left to itself
  // hipcc emits "13 reads, lgkmcnt(0), 36 MFMAs" per tap, and with both waves of a SIMD in the same phase behind the barrier
  // the LDS time (312 KB per super-step) simply added to the matrix time.  Waits name the registers they release.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned b_rd[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = wn * TN + j * 16 + fr;
    b_rd[j] = lds0 + A_BYTES + n * 64 + ((fq ^ swz(n)) * 16);
  }
  const int r_base = wm * TM + fr;  // this lane's pixel of fragment ii: r_base + 16 ii; column tap t reads image row + t d
  h16x8 af[2][WM], bq[8];
#define CONV_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define CONV_RDA(dst, ii, t)                                                        \
  {                                                                                 \
    const int r_ = r_base + (ii) * 16 + (t) * d;                                    \
    CONV_RD(dst, lds0 + sbase + r_ * 64 + ((fq ^ swz(r_)) * 16), 0);                \
  }

#define CONV_DMA(t) \
  if (ss + 1 < NSS) stage_piece(ss + 1, (ss + 1) & 1, t, doff[t]);
#pragma unroll
  for (int t = 0; t < PW; ++t) stage_piece(0, 0, t, doff[t]);
  for (int ss = 0; ss < NSS; ++ss) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < PW; ++t) CONV_DMA(t)
    const unsigned sbase = (unsigned)((ss & 1) * SSTEP);
    // 39 fragment reads issued six fragments ahead of the 108 MFMAs that consume them, counted waits
    // (generated: tools/gen_conv288_schedule.py)
#include "conv288_schedule.inc"
  }
#undef CONV_RD
#undef CONV_RDA
#undef CONV_DMA

  // ---- epilogue: wave-private strips (32 pixels x 144 channels, 16-bit) -> 16-byte row-contiguous stores ---------------------
  __syncthreads();  // every wave is done with the last super-step's buffers
  constexpr int PITCH = TN * 2 + 16;                 // 304 bytes
  constexpr int CPR = TN * 2 / 16;                   // 18 sixteen-byte pieces per strip row
  constexpr int NIT = 32 * CPR / 64;                 // 9 store instructions per pass
  char* const strip = smem + wave * (32 * PITCH);
  h16_t* const ytile = p.y + ((int64_t)row * p.W + j0) * C;
#pragma unroll
  for (int ip = 0; ip < 2; ++ip) {
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const f32x4 a = acc[ip * 2 + ih][j];
        uint2 o;
        o.x = rf_pack2_h16(a[0], a[1]);
        o.y = rf_pack2_h16(a[2], a[3]);
        *(uint2*)(strip + (ih * 16 + fr) * PITCH + (j * 16 + 4 * fq) * 2) = o;
      }
    asm volatile("" ::: "memory");  // (in-order LDS per wave: no barrier)
#pragma unroll
    for (int t = 0; t < NIT; ++t) {
      const int idx = lane + 64 * t;
      const int r = idx / CPR, c = idx % CPR;
      const f32x4 v = *(const f32x4*)(strip + r * PITCH + c * 16);
      f32x4* dst = (f32x4*)(ytile + (int64_t)(wm * TM + ip * 32 + r) * C + wn * TN + c * 8);
      if (p.nt_store)
        __builtin_nontemporal_store(v, dst);
      else
        *dst = v;
    }
    asm volatile("" ::: "memory");
  }
}

// Returns 1 and launches when the descriptor is a 288 -> 288 channel 3x3 convolution with rows that are whole 256-pixel tiles
// (16-bit in / out, no activation, no residual), 0 otherwise (the generic implicit-GEMM path takes it).
int rf_conv288_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream) {
  *rc = 0;
  static const bool off = rf_env_flag("RF_NO_CONV288");
  if (off) return 0;
  if (d.a_mode != RF_AMODE_CONV3X3 || batch != 1 || d.ab_dtype != RF_H16 || d.c_dtype != RF_H16) return 0;
  if (d.conv_c != 288 || d.N != 288 || d.conv_w % 256 != 0 || d.conv_dil < 1 || d.conv_dil > 8) return 0;
  if (d.act != RF_ACT_NONE || d.alpha != 1.0f || d.residual || d.ln_out || d.c_rc > 0 || d.c_cc > 0 || d.c_ri != 288 || d.b_ri != 9 * 288)
    return 0;
  if (d.bias_mode != RF_BIAS_NONE && d.bias_mode != RF_BIAS_COL) return 0;
  if (((uintptr_t)d.A % 16) || ((uintptr_t)d.B % 16) || ((uintptr_t)d.C % 16) || (d.bias_mode == RF_BIAS_COL && ((uintptr_t)d.bias % 16))) return 0;
  Conv288P p;
  p.x = (const h16_t*)d.A; p.w = (const h16_t*)d.B; p.y = (h16_t*)d.C;
  p.bias = d.bias_mode == RF_BIAS_COL ? d.bias : nullptr;
  p.B = d.conv_n; p.H = d.conv_h; p.W = d.conv_w; p.dil = d.conv_dil;
  p.tpr = d.conv_w / 256;
  const int64_t nt = (int64_t)d.conv_n * d.conv_h * p.tpr;
  if (nt > 0x7fffffffLL) return 0;
  static const bool no_nt = rf_env_flag("RF_NO_NT_STORE");
  p.nt_store = ((int64_t)d.M * 288 * 2 > (64ll << 20)) && !no_nt;
  if (const int e = rf_enable_big_lds<conv3x3_c288_kernel>()) {
    *rc = e;
    return 1;
  }
  hipLaunchKernelGGL(conv3x3_c288_kernel, dim3((unsigned)nt), dim3(512), 2 * (17 + 54) * 1024, (hipStream_t)stream, p);
  *rc = rf_launch_status();
  return 1;
}
