// Shared device helpers for the gfx950 kernels of librfmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "rfmi.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

#define RF_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32 on gfx950: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(bf16_t, b);
}

// two floats -> packed bf16 pair (low half = a): ONE v_cvt_pk_bf16_f32.  (f2bf(a) | f2bf(b) << 16 compiles to two
// conversions plus a shift and an or.)
typedef __attribute__((ext_vector_type(2))) float rf_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 rf_bf16x2;
__device__ __forceinline__ unsigned rf_pack2_bf16(float a, float b) {
  const rf_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rf_bf16x2));
}

// dtype-generic scalar load/store (T = activation dtype chosen by the host: fp32 or bf16)
__device__ __forceinline__ float ld(const void* p, int dtype, int64_t i) {
  return dtype == RF_F32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void st(void* p, int dtype, int64_t i, float v) {
  if (dtype == RF_F32)
    ((float*)p)[i] = v;
  else
    ((bf16_t*)p)[i] = f2bf(v);
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

// environment switches (A/B experiments) are read once per process, not per launch
static inline bool rf_env_flag(const char* name) { return getenv(name) != nullptr; }

// gemm_fast.hip: persistent plain-layout bf16 GEMM; returns 1 (launched, *rc = status) or 0 (descriptor does not fit)
int rf_gemm_fast_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream);
// gemm_wreg.hip: skinny-K (K = 288 / 384) projection GEMM with register-resident weights; same return convention
int rf_gemm_wreg_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream);
void rf_gemm_fast_set_stamps(void* buf);
