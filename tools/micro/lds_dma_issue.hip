// Issue cost of LDS-DMA requests for the issuing wave: does it depend on M0 changing between requests?
// Each wave issues, per iteration, three `buffer_load_dwordx4 ... offen lds` requests into its own LDS region,
//   mode 0: LDS destinations 4 KB apart (the compiler writes M0 before every request),
//   mode 1: destinations 1 KB apart, expressed through the instruction's immediate offset (one M0 write per iteration;
//           the immediate also moves the memory address, which the per-lane offset takes back),
//   mode 2: like 0 but with a wait for each request before the next (latency of one request),
// and prints cycles per request (s_memtime) for 4 and 8 waves per CU on every CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_issue lds_dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ void k(const char* src, unsigned nbytes, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char* mine = lds + wave * 16384;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, (int)nbytes, 0x00020000);
  // six lanes share a 96-byte row unit (about eleven rows per request), rows pseudo-random in a 32 MB table
  unsigned row = (blockIdx.x * 977u + (threadIdx.x / 6) * 131u) % (nbytes / 576 - 64);
  unsigned long long burst = 0;
  for (int it = -8; it < iters; ++it) {
    const unsigned vo = row * 576u + (lane % 6) * 16u;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1) {  // one M0 write, six immediates
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)mine, 16, vo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)mine, 16, vo + 96 - 1024, 0, 1024, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)mine, 16, vo + 192 - 2048, 0, 2048, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)mine, 16, vo + 288 - 3072, 0, 3072, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + 4096), 16, vo + 384, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + 4096), 16, vo + 480 - 1024, 0, 1024, 0);
    } else if (MODE == 0) {  // M0 written before every request
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)mine, 16, vo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + 4096), 16, vo + 96, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + 8192), 16, vo + 192, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + 12288), 16, vo + 288, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + 1024), 16, vo + 384, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + 5120), 16, vo + 480, 0, 0, 0);
    } else {  // plain loads into registers, for comparison
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      u4 r[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + q * 96, 0, 0);
      asm volatile("" :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]));
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (it >= 0) burst += tb - ta;
    row = (row * 1664525u + 1013904223u) % (nbytes / 576 - 256);
    __builtin_amdgcn_s_sleep(12);   // about 770 cycles of "matrix work"
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = burst;
}

template <int MODE>
static void run(const char* d, unsigned nbytes, int waves, unsigned long long* o, const char* what) {
  const int blocks = 256, iters = 400;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves), waves * 16384, 0, d, nbytes, iters, o);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * waves);
  hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-58s %d waves/CU: %6.1f cycles per request (burst of six)\n", what, waves, s / h.size() / iters / 6);
}

int main() {
  const unsigned nbytes = 32u << 20;
  char* d; unsigned long long* o;
  hipMalloc(&d, nbytes); hipMemset(d, 1, nbytes); hipMalloc(&o, 256 * 8 * 8);
  for (int waves : {4, 8}) {
    run<0>(d, nbytes, waves, o, "M0 written before every request (destinations 4 KB apart)");
    run<1>(d, nbytes, waves, o, "two M0 writes per six requests (immediate offsets)");
    run<2>(d, nbytes, waves, o, "plain buffer loads into registers");
  }
  return 0;
}
