// Probe (round 6): are 8-byte LDS stores to addresses that are only 2-byte aligned executed correctly on gfx950, and at what cost?
// Every lane stores 8 bytes at row * 128 + 2 * skew with skew = lane-dependent 0..3 (the pattern a row-major skew buffer of the dQ
// kernel's diagonal sums would need), then the block reads the buffer back with aligned loads.  Prints mismatches and cycles per
// store for aligned / unaligned addresses.   build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_unaligned.hip -o tools/probe/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void k(uint16_t* out, unsigned long long* cyc, int unaligned) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 128];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 128 / 4; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0;
  __syncthreads();
  const int skew = unaligned ? (lane & 3) : 0;
  const unsigned addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)(lds + lane * 128 + 2 * skew);
  const unsigned lo = 0x00010000u * (unsigned)(2 * lane + 1) + (unsigned)(2 * lane), hi = lo + 0x00020002u;   // four distinct 16-bit values
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    unsigned a = addr + r * 8;
    asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(((unsigned long long)hi << 32) | lo) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  for (int i = lane; i < 64 * 64; i += 64) out[i] = reinterpret_cast<uint16_t*>(lds)[i];
  if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
  uint16_t* d; unsigned long long* c;
  hipMalloc(&d, 64 * 64 * 2); hipMalloc(&c, 8);
  for (int un = 0; un < 2; ++un) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, un);
    std::vector<uint16_t> h(64 * 64); unsigned long long cy = 0;
    hipMemcpy(h.data(), d, 64 * 64 * 2, hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
      const int skew = un ? (lane & 3) : 0;
      for (int r = 0; r < 8; ++r)
        for (int e = 0; e < 4; ++e) {
          const unsigned lo = 0x00010000u * (unsigned)(2 * lane + 1) + (unsigned)(2 * lane), hi = lo + 0x00020002u;
          const uint16_t want = (uint16_t)((e < 2 ? lo : hi) >> (16 * (e & 1)));
          if (h[lane * 64 + skew + r * 4 + e] != want) ++bad;
        }
    }
    printf("%s 8-byte LDS stores: %d mismatching halves of 2048, %llu cycles for 8 stores per lane\n", un ? "2-byte-aligned" : "8-byte-aligned", bad, cy);
  }
  return 0;
}
