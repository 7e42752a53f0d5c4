// CU-side ceiling of the operand path on gfx950: every CU streams 1-KiB wave-wide requests either by LDS-DMA
// (global_load_lds_dwordx4, what the GEMMs use) or by global_load_dwordx4 into VGPRs, with a bounded number of
// requests in flight, over footprints that sit in L2, in MALL, or in HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_peak dma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// each wave: per iteration 8 requests of 1 KiB; block = 8 waves -> 64 KiB per iteration per CU
template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(512) void stream_kernel(const char* src, size_t cu_stride, size_t foot_mask, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* base = src + (size_t)blockIdx.x * cu_stride;
  u32x4 accv = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t off = (((size_t)it * 8 + j) * 8192 + wave * 1024) & foot_mask;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gas_ptr)(base + off + lane * 16), (las_ptr)(smem + ((it & 1) * 8 + j) * 8192 + wave * 1024), 16, 0, 0);
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + off + lane * 16);
        accv ^= v;
      }
    }
    if (MODE == 0) {
      if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((accv[0] ^ accv[1] ^ accv[2] ^ accv[3]) == 0x12345u) sink[0] = 1;
}

// GEMM-shaped request stream: CU b (XCD b % 8) owns tile (tm, tn) of its XCD's 4 x 8 patch; per K step it pulls 256 rows
// of its A panel and 256 rows of its B panel, 128 B per row (8 rows per wave-wide request), rows `ld` bytes apart.
__global__ __launch_bounds__(512) void gemm_stream_kernel(const char* A, const char* B, size_t ld, int ksteps, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, tm = idx >> 3, tn = idx & 7;
  const char* pa = A + (size_t)(xcd * 4 + tm) * 256 * ld;
  const char* pb = B + (size_t)tn * 256 * ld;
  for (int it = 0; it < iters; ++it) {
    const size_t col = (size_t)(it % ksteps) * 128 + (lane & 7) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ((j & 3) * 8 + wave) * 8 + (lane >> 3);
      const char* src = ((j < 4) ? pa : pb) + (size_t)r * ld + col;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(smem + ((it & 1) * 8 + j) * 8192 + wave * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static void run_gemm_stream(const char* buf, size_t ld, int ksteps) {
  const int iters = 2000, blocks = 256;
  hipFuncSetAttribute((const void*)gemm_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* A = buf;
  const char* B = buf + ((size_t)2 << 30);
  gemm_stream_kernel<<<blocks, 512, 131072>>>(A, B, ld, ksteps, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  gemm_stream_kernel<<<blocks, 512, 131072>>>(A, B, ld, ksteps, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 65536 * blocks;
  printf("gemm-shaped stream  ld=%6zu B  K steps %3d   %7.3f ms  %7.2f TB/s  %6.1f GB/s/CU  (%.2f us per 64-KiB K step)\n", ld, ksteps, ms,
         bytes / ms / 1e9, bytes / ms / 1e6 / blocks, ms * 1e3 / iters);
}

template <int MODE, int INFLIGHT>
static void run(const char* what, const char* src, size_t cu_stride, size_t foot, unsigned* sink, int threads) {
  const int iters = 2000, blocks = 256;
  hipFuncSetAttribute((const void*)stream_kernel<MODE, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  stream_kernel<MODE, INFLIGHT><<<blocks, threads, 131072>>>(src, cu_stride, foot - 1, 50, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  stream_kernel<MODE, INFLIGHT><<<blocks, threads, 131072>>>(src, cu_stride, foot - 1, iters, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 8 * 1024 * (threads / 64) * blocks;
  printf("%-44s %-8s inflight/wave %2d  %7.3f ms  %7.2f TB/s  %6.1f GB/s/CU  %5.1f B/clk/CU@2.4GHz\n", what,
         MODE == 0 ? "lds-dma" : "vgpr", MODE == 0 ? INFLIGHT + 8 : 0, ms, bytes / ms / 1e9, bytes / ms / 1e6 / blocks,
         bytes / ms / 1e6 / blocks / 2.4);
}

int main() {
  char* buf;
  unsigned* sink;
  const size_t total = (size_t)4 << 30;
  hipMalloc(&buf, total);
  hipMemset(buf, 1, total);
  hipMalloc(&sink, 4);
  for (size_t ld : {1536, 1536 + 128, 6144, 6144 + 128, 12288, 12288 + 128, 12288 + 256, 4096, 4096 + 128})
    run_gemm_stream(buf, ld, (int)(ld / 128 < 96 ? ld / 128 : 96));
  // (a) every CU reads the same 256 KiB: pure L2-hit, CU-side limit
  run<0, 8>("all CUs same 256 KiB (L2 hot)", buf, 0, 256 << 10, sink, 512);
  run<0, 4>("all CUs same 256 KiB (L2 hot)", buf, 0, 256 << 10, sink, 512);
  run<0, 0>("all CUs same 256 KiB (L2 hot)", buf, 0, 256 << 10, sink, 512);
  run<1, 0>("all CUs same 256 KiB (L2 hot)", buf, 0, 256 << 10, sink, 512);
  run<0, 8>("all CUs same 256 KiB, 4 waves", buf, 0, 256 << 10, sink, 256);
  // (b) each CU its own 64 KiB window (fits L2: 16 MiB total over 8 XCDs)
  run<0, 8>("own 64 KiB per CU (L2)", buf, 64 << 10, 64 << 10, sink, 512);
  run<1, 0>("own 64 KiB per CU (L2)", buf, 64 << 10, 64 << 10, sink, 512);
  // (c) each CU its own 512 KiB window: 128 MiB total -> MALL
  run<0, 8>("own 512 KiB per CU (MALL)", buf, 512 << 10, 512 << 10, sink, 512);
  run<1, 0>("own 512 KiB per CU (MALL)", buf, 512 << 10, 512 << 10, sink, 512);
  // (d) each CU its own 16 MiB window: 4 GiB -> HBM
  run<0, 8>("own 16 MiB per CU (HBM)", buf, (size_t)16 << 20, (size_t)16 << 20, sink, 512);
  run<1, 0>("own 16 MiB per CU (HBM)", buf, (size_t)16 << 20, (size_t)16 << 20, sink, 512);
  return 0;
}
