// Microbenchmark: issue rate of v_mfma_f32_32x32x16_bf16 with 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void k(float* out, int iters) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, const char* name) {
  float* out; hipMalloc(&out, 1024 * 1024 * 4);
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * NACC * (threads / 64) / 4.0;  // one block per CU assumed
  printf("%s: %d waves/CU, %d acc: %.3f ms, %.1f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", name, threads / 64, NACC, ms,
         ms * 1e6 / mfma_per_simd, (double)iters * NACC * (threads / 64) * blocks * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  run<1>(256, "dependent chain");
  run<2>(256, "2 acc");
  run<4>(256, "4 acc");
  run<6>(256, "6 acc");
  run<3>(512, "3 acc, 2 waves/SIMD");
  run<6>(512, "6 acc, 2 waves/SIMD");
  return 0;
}
