// Gather rate of the convolution's activation rows into LDS, by how a row's 16-channel units lie against cache lines:
//   layout 0: three-term units of 96 bytes packed back to back (row stride 576 bytes for 96 channels): half of the
//             units straddle two 128-byte lines;
//   layout 1: the same units on a 128-byte stride (row stride 768 bytes): every unit inside one line.
// One workgroup = 4 waves gathers 128 rows x one unit per stage (12 requests, 12 KB) with two stages in flight at most,
// as the convolution does (vmcnt(0) + barrier per stage); the row ids change every 6 stages; two workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o gather_align gather_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(256) void k(const char* src, unsigned rows, unsigned row_stride, unsigned unit_stride, int stages,
                                         unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 x 12 KB used; the rest only limits occupancy
  const int tid = threadIdx.x, wave = tid >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, (int)0xFFFFFF00u, 0x00020000);
  unsigned seed = blockIdx.x * 2654435761u + (tid / 6) * 40503u + 17u;
  unsigned vo[3];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < stages; ++s) {
    if (s % 6 == 0) {
#pragma unroll
      for (int v = 0; v < 3; ++v) {   // slot e = v * 256 + tid: row e / 6 of the tile, piece e % 6
        seed = seed * 1664525u + 1013904223u;
        const unsigned e = v * 256 + tid;
        unsigned r = (blockIdx.x * 7919u + (e / 6) * 2654435761u + (s / 6) * 97u) % rows;   // same row for the six lanes of a unit
        vo[v] = r * row_stride + (e % 6) * 16u;
      }
    }
    char* buf = lds + (s & 1) * 12288;
    const unsigned soff = (s % 6) * unit_stride;
#pragma unroll
    for (int v = 0; v < 3; ++v)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(buf + (v * 256 + wave * 64) * 16), 16, vo[v], soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // the previous stage has landed
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
  const unsigned rows = 273080;
  char* d; unsigned long long* o;
  hipMalloc(&d, (size_t)rows * 768 + 4096); hipMemset(d, 1, (size_t)rows * 768 + 4096); hipMalloc(&o, 8192 * 8);
  const int blocks = 2048, stages = 600;
  for (int rep = 0; rep < 2; ++rep)
    for (int layout = 0; layout < 2; ++layout) {
      const unsigned rs = layout ? 768 : 576, us = layout ? 128 : 96;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, d, rows, rs, us, stages, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)blocks * stages * 12288;
      printf("layout %d (row stride %u, unit stride %u): %.3f ms, %.2f TB/s chip-wide, %.1f GB/s per CU\n", layout, rs, us, ms,
             bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    }
  return 0;
}
