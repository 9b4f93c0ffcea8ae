// Microbenchmark (not part of the product): how fast can ONE CU move L2-resident bytes into LDS with global_load_lds, as a
// function of how many waves issue, how many 1 KB pieces each keeps in flight, and the shape of a piece?
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC dma_bench.hip -o libdmabench.so ; python tools/dma_bench/run.py
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// mode 0: piece = 1 KB contiguous (lane * 16);  mode 1: piece = 16 rows x 64 bytes, rows `row_stride` bytes apart;
// mode 2: plain global_load_dwordx4 into registers (1 KB contiguous per instruction), no LDS
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, int64_t src_bytes, int nwaves_active, int pieces_per_wave,
                                                   int mode, int64_t row_stride, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves_active) return;
  // every workgroup walks the whole buffer from its own offset: L2-resident after the first pass
  int64_t off = ((int64_t)blockIdx.x * 8 + wave) * 64 * 1024 % src_bytes;
  const int64_t lane_off = mode == 1 ? (int64_t)(lane >> 2) * row_stride + (lane & 3) * 16 : (int64_t)lane * 16;
  const int64_t step = mode == 1 ? 64 : 1024;   // gather pieces advance along the rows
  char* ring = smem + wave * (DEPTH * 1024);
  unsigned acc = 0;
  if (mode == 2) {
    uint4 r[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < pieces_per_wave; i += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        acc += r[d].x;
        r[d] = *(const uint4*)(src + off + lane_off);
        off += step;
        if (off + 1024 > src_bytes) off = 0;
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += r[d].x;
  } else {
    for (int i = 0; i < pieces_per_wave; i += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        // keep DEPTH pieces in flight: before reusing ring slot d the piece issued DEPTH instructions ago must have landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        glds16(src + off + lane_off, ring + d * 1024);
        off += step;
        if (off + (mode == 1 ? 16 * row_stride : 1024) > src_bytes) off = 0;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *(unsigned*)(ring + lane * 4);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int dma_bench(const void* src, int64_t src_bytes, int grid, int nwaves_active, int pieces_per_wave, int depth, int mode,
                         int64_t row_stride, void* sink, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(D)                                                                                                          \
  {                                                                                                                        \
    hipFuncSetAttribute((const void*)dma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
    hipLaunchKernelGGL(dma_kernel<D>, dim3(grid), dim3(512), 8 * D * 1024, s, (const char*)src, src_bytes, nwaves_active,  \
                       pieces_per_wave, mode, row_stride, (unsigned*)sink);                                               \
  }
  if (depth == 2) LAUNCH(2) else if (depth == 4) LAUNCH(4) else if (depth == 8) LAUNCH(8) else if (depth == 16) LAUNCH(16) else return -1;
#undef LAUNCH
  return (int)hipGetLastError();
}
