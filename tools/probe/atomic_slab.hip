// Probe for a single-kernel attention backward: how fast can the key-block workgroups of one (b, h)
// accumulate their dQ contributions into an fp32 slab [T, 64] with global_atomic_add_f32?
// Pattern of WavLM-Base at 32 x 15 s: 384 (b, h) x 6 key blocks of 128 keys; every key-block workgroup adds a 128 x 64 fp32
// tile to each of the 6 query blocks of its (b, h): 384 * 6 * 6 * 8192 = 113 M float atomics (453 MB) per layer, against
// 37 MB of bf16 dQ.  Variants: (a) relaxed agent-scope atomics, the 6 workgroups of a (b, h) on ONE XCD (block b -> XCD b % 8)
// or spread over all 8; (b) plain stores of the same tiles (the floor: what the write traffic alone costs); (c) fp32 slabs
// per key block + a reduction pass (6 x 74 MB written, read once).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/atomic_slab.hip -o tools/probe/atomic_slab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define T 768
#define HD 64
#define NBH 384
#define NKB 6
#define NQB 6

template <int MODE, bool SAME_XCD>
__global__ __launch_bounds__(256) void probe_kernel(float* __restrict__ dq, float* __restrict__ slabs, int spin) {
  // work item = (bh, kb).  SAME_XCD: the NKB items of one bh sit on one XCD (consecutive ids on one XCD are id, id + 8, ...)
  const int id = blockIdx.x;
  int bh, kb;
  if (SAME_XCD) { const int x = id & 7, k = id >> 3; const int local = k; bh = (local / NKB) * 8 + x; kb = local % NKB; }
  else { bh = id / NKB; kb = id % NKB; }
  if (bh >= NBH) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v = (float)(lane + 1) * 1e-3f;
  for (int qb = 0; qb < NQB; ++qb) {
    // stand-in for the tile's MFMA work: `spin` dependent fmas per element
    for (int s = 0; s < spin; ++s) v = fmaf(v, 1.0000001f, 1e-7f);
    // the wave's 32 rows x 64 floats of the 128 x 64 tile: one row (256 B) per instruction
    float* base = (MODE == 2 ? slabs + ((size_t)kb * NBH + bh) * (T * HD) : dq + (size_t)bh * (T * HD)) + (size_t)(qb * 128 + wave * 32) * HD;
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
      float* p = base + r * HD + lane;
      if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *p = v;
    }
  }
}

__global__ void reduce_kernel(const float* __restrict__ slabs, unsigned short* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < NKB; ++k) s += slabs[(size_t)k * n + i];
  out[i] = (unsigned short)(__float_as_uint(s) >> 16);
}

int main() {
  const size_t n = (size_t)NBH * T * HD;
  float *dq, *slabs; unsigned short* out;
  hipMalloc(&dq, n * 4); hipMalloc(&slabs, n * 4 * NKB); hipMalloc(&out, n * 2);
  hipMemset(dq, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = NBH * NKB;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.1f us per layer-sized pass\n", name, ms * 100.f);
  };
  for (int spin : {0, 2000}) {
    printf("-- stand-in compute per tile: %d dependent fmas\n", spin);
    run("atomic add f32, the key blocks of a (b,h) on ONE XCD", [&] { hipLaunchKernelGGL((probe_kernel<0, true>), dim3(grid), dim3(256), 0, 0, dq, slabs, spin); });
    run("atomic add f32, key blocks spread over the XCDs", [&] { hipLaunchKernelGGL((probe_kernel<0, false>), dim3(grid), dim3(256), 0, 0, dq, slabs, spin); });
    run("plain stores of the same tiles (floor)", [&] { hipLaunchKernelGGL((probe_kernel<1, true>), dim3(grid), dim3(256), 0, 0, dq, slabs, spin); });
    run("fp32 slabs per key block (stores) + reduction to bf16", [&] {
      hipLaunchKernelGGL((probe_kernel<2, true>), dim3(grid), dim3(256), 0, 0, dq, slabs, spin);
      hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, slabs, out, n); });
  }
  printf("reference: the two-kernel backward of this layer runs 411 us (dQ 207 + dK/dV 190 + 14); its dQ kernel alone 207 us\n");
  return 0;
}
