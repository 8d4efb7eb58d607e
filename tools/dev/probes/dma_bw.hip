// L2 -> CU bandwidth per path: LDS-DMA (global_load_lds), buffer LDS-DMA, plain dwordx4 loads to VGPRs (+ ds_write).
// hipcc --offload-arch=gfx950 -O3 dma_bw.hip -o dma_bw && ./dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int REGION = 64 * 1024;   // bytes per block, re-read ITER times (L2-resident: 256 blocks x 64 KiB = 16 MiB)

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void bw(const char* src, unsigned* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (long)blockIdx.x * REGION;
  u32x4_t accv = {0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, REGION, 0x00020000);
  constexpr int PIECES = REGION / 1024 / NW;      // pieces per wave per sweep
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int piece = wave * PIECES + p;
      const int off = piece * 1024 + lane * 16;
      if (MODE == 0) __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(smem + piece * 1024), 16, 0, 0);
      else if (MODE == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + piece * 1024), 16, off, 0, 0, 0);
      else if (MODE == 2) { const u32x4_t v = *(const u32x4_t*)(base + off); accv ^= v; }
      else { const u32x4_t v = *(const u32x4_t*)(base + off); *(u32x4_t*)(smem + off) = v; }
    }
    if (MODE < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES > 8 ? 8 : PIECES / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE >= 2 || sink == nullptr) sink[blockIdx.x * NW * 64 + threadIdx.x] = accv[0] ^ accv[1] ^ accv[2] ^ accv[3] ^ ((unsigned*)smem)[threadIdx.x];
}

template <int MODE, int NW>
void run(const char* name, const char* d, unsigned* sink, int blocks) {
  hipFuncSetAttribute((const void*)bw<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, REGION);
  const int iters = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((bw<MODE, NW>), dim3(blocks), dim3(NW * 64), REGION, 0, d, sink, 20);
  hipEventRecord(e0);
  hipLaunchKernelGGL((bw<MODE, NW>), dim3(blocks), dim3(NW * 64), REGION, 0, d, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * REGION * iters;
  printf("%-34s %2d waves x %3d blocks: %7.2f TB/s  = %5.1f B/clk/CU at 2.1 GHz (%.3f ms)\n", name, NW, blocks, bytes / ms / 1e9,
         bytes / blocks / (ms * 1e-3 * 2.1e9) * (blocks > 256 ? blocks / 256.0 : 1.0), ms);
}

int main() {
  char* d; hipMalloc(&d, 512L * REGION); hipMemset(d, 1, 512L * REGION);
  unsigned* sink; hipMalloc(&sink, 512 * 1024 * 4);
  run<0, 8>("global_load_lds", d, sink, 256);
  run<0, 16>("global_load_lds", d, sink, 256);
  run<0, 4>("global_load_lds", d, sink, 256);
  run<1, 8>("buffer_load lds", d, sink, 256);
  run<2, 8>("global_load_dwordx4 -> VGPR", d, sink, 256);
  run<2, 16>("global_load_dwordx4 -> VGPR", d, sink, 256);
  run<3, 8>("global_load_dwordx4 -> ds_write", d, sink, 256);
  run<3, 16>("global_load_dwordx4 -> ds_write", d, sink, 256);
  run<0, 8>("global_load_lds (2 blocks/CU)", d, sink, 512);
  return 0;
}
