// What does `buffer_load_dwordx4 ... offen lds` leave in LDS for a lane whose offset is beyond num_records?
// Build: hipcc --offload-arch=gfx950 -O2 -o buffer_lds_oob buffer_lds_oob.hip ; prints the first words of each lane's 16 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const char* src, int nbytes, unsigned* out, unsigned soff) {
  __shared__ __attribute__((aligned(16))) unsigned lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, nbytes, 0x00020000);
  // even lanes read inside the buffer, odd lanes far outside
  const unsigned voff = (threadIdx.x & 1) ? 0xFFFFFF00u : threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  char* d; unsigned* o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x11110000u + i;
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  for (unsigned soff : {0u, 0x100u, 0x400u}) {  // 0xFFFFFF00 + 0x100 wraps to 0, + 0x400 wraps to 0x300 (inside the buffer)
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2048, o, soff);
    unsigned r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("scalar offset 0x%x\n", soff);
    for (int l = 0; l < 4; ++l) printf("  lane %d: %08x %08x %08x %08x\n", l, r[4 * l], r[4 * l + 1], r[4 * l + 2], r[4 * l + 3]);
  }
  return 0;
}
