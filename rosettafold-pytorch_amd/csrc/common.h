// Shared device helpers for the gfx950 kernels of librfmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// gemm_fast.hip: persistent plain-layout bf16 GEMM; returns 1 (launched, *rc = status) or 0 (descriptor does not fit)
int rf_gemm_fast_try(const rf_gemm_desc& d, int64_t batch, int* rc, void* stream);
void rf_gemm_fast_set_stamps(void* buf);
