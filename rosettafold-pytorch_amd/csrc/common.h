// Shared device helpers for the gfx950 kernels of librfmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "rfmi.h"

// ---- the 16-bit MFMA operand type of this build ("h16") ---------------------------------------------------------------
// The library is compiled twice from the same sources (csrc/Makefile):
//   librfmi.so       h16 = bfloat16   (RF_H16 == RF_BF16):  v_mfma_f32_16x16x32_bf16, v_cvt_pk_bf16_f32
//   librfmi_f16.so   h16 = IEEE half  (RF_H16 == RF_F16, -DRF_H16_IS_F16):  v_mfma_f32_16x16x32_f16, v_cvt_pk_f16_f32
// Same MFMA rate, same bytes; fp16 carries 11 significand bits against bf16's 8, i.e. 8x smaller operand rounding (the
// whole-model gap of the 16-bit path is operand rounding amplified by depth), at the price of fp16's
// range (|x| < 65504): kernels whose 16-bit intermediates are sums over the sequence scale them in the f16 build (favor.hip).
// fp32 accumulation, fp32 residual streams and fp32 statistics are the same in both builds.  Kernel code is written once
// against h16_t / h2f / f2h / rf_pack2_h16 / rf_mfma16; a dtype code other than RF_F32 / RF_H16 is rejected by every entry
// point of a build (RF_EINVAL), so a tensor of the other 16-bit type can never be misread.
typedef __attribute__((ext_vector_type(8))) short h16x8;
typedef __attribute__((ext_vector_type(4))) short h16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short h16_t;
typedef __attribute__((ext_vector_type(2))) float rf_f32x2;

#define RF_WAVE 64

#ifdef RF_H16_IS_F16
#define RF_H16 RF_F16
#define RF_H16_ONE2 0x3C003C00u  // two packed 1.0
typedef __attribute__((ext_vector_type(2))) _Float16 rf_h16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 rf_h16x8n;
__device__ __forceinline__ float h2f(h16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ h16_t f2h(float x) { return __builtin_bit_cast(h16_t, (_Float16)x); }  // round-to-nearest-even
__device__ __forceinline__ f32x4 rf_mfma16(h16x8 a, h16x8 b, f32x4 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rf_h16x8n, a), __builtin_bit_cast(rf_h16x8n, b), c, 0, 0, 0);
}
#else
#define RF_H16 RF_BF16
#define RF_H16_ONE2 0x3F803F80u
typedef __attribute__((ext_vector_type(2))) __bf16 rf_h16x2;
__device__ __forceinline__ float h2f(h16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ h16_t f2h(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32 on gfx950: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(h16_t, b);
}
__device__ __forceinline__ f32x4 rf_mfma16(h16x8 a, h16x8 b, f32x4 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif

// two floats -> packed h16 pair (low half = a): ONE v_cvt_pk_{bf16,f16}_f32.  (f2h(a) | f2h(b) << 16 compiles to two
// conversions plus a shift and an or.)
__device__ __forceinline__ unsigned rf_pack2_h16(float a, float b) {
  const rf_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rf_h16x2));
}
// the two halves of a packed pair as floats
__device__ __forceinline__ float rf_h16_lo(unsigned u) {
#ifdef RF_H16_IS_F16
  return (float)__builtin_bit_cast(rf_h16x2, u)[0];
#else
  return __uint_as_float(u << 16);
#endif
}
__device__ __forceinline__ float rf_h16_hi(unsigned u) {
#ifdef RF_H16_IS_F16
  return (float)__builtin_bit_cast(rf_h16x2, u)[1];
#else
  return __uint_as_float(u & 0xffff0000u);
#endif
}

// dtype-generic scalar load/store (T = activation dtype chosen by the host: fp32 or the build's h16)
__device__ __forceinline__ float ld(const void* p, int dtype, int64_t i) {
  return dtype == RF_F32 ? ((const float*)p)[i] : h2f(((const h16_t*)p)[i]);
}
__device__ __forceinline__ void st(void* p, int dtype, int64_t i, float v) {
  if (dtype == RF_F32)
    ((float*)p)[i] = v;
  else
    ((h16_t*)p)[i] = f2h(v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

static inline int rf_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// Kernels that need more than 64 KB of dynamic LDS: the opt-in attribute is per device, so it is set once per
// (kernel, device the calling thread is on) and its return code is reported.  0 = ok, else a hipError_t.
template <auto Kernel>
static inline int rf_enable_big_lds() {
  static unsigned long long done = 0;  // bit d: set on device d (devices >= 64: set every call)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return 0;
  e = hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < 64) done |= 1ull << dev;
  return 0;
}

// compute units of the device the calling thread is on (cached per device)
static inline int rf_num_cus() {
  static int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64 && cus[dev]) return cus[dev];
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64) cus[dev] = prop.multiProcessorCount;
  return prop.multiProcessorCount;
}

// Ablation / phase-timing switches (RF_GEMM_DBG, RF_TIED_DBG, RF_FAVOR_DBG: results are WRONG when set) exist only in a tuning
// build (make ABLATION=1 -> -DRF_ABLATION).  In the production libraries RF_DBG() folds to 0 at compile time -- the kernels
// carry neither the branches nor the registers of the instrumentation -- and an entry point that finds its switch set in the
// environment refuses to launch (RF_EINVAL) instead of silently ignoring it.
#ifdef RF_ABLATION
#define RF_DBG(x) (x)
static inline int rf_dbg_env(const char* name, int* out) {
  static thread_local int dummy;
  (void)dummy;
  const char* v = getenv(name);
  *out = v ? atoi(v) : 0;
  return 0;
}
#else
#define RF_DBG(x) 0
static inline int rf_dbg_env(const char* name, int* out) {
  *out = 0;
  return getenv(name) ? RF_EINVAL : 0;  // the switch does not exist in this build: say so
}
#endif

// environment switches (A/B experiments) are read once per process, not per launch
static inline bool rf_env_flag(const char* name) { return getenv(name) != nullptr; }

// gemm_fast.hip: persistent plain-layout bf16 GEMM; returns 1 (launched, *rc = status) or 0 (descriptor does not fit)
int rf_gemm_fast_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream);
// gemm_wreg.hip: skinny-K (K = 288 / 384) projection GEMM with register-resident weights; same return convention
int rf_gemm_wreg_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream);
int rf_conv288_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream);  // csrc/conv288.hip
void rf_gemm_fast_set_stamps(void* buf);
