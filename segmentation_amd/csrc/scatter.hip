// torch_scatter's two operators on the hot path's fringe, as the reference calls them:
//   scatter(src, index, dim=0, reduce="mean" | "sum")   mosaic3d/src/models/utils/structure.py:58 (voxel means),
//                                                       mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:410
//   segment_csr(src, indptr, reduce="mean" | "sum")     mosaic3d/src/models/losses/caption_loss.py:223
// Rows are [*, c] f32.  segment_csr walks each segment in row order (a fixed summation order: run-to-run identical);
// scatter adds rows with float atomics (the order of the adds, and so the last bits, may differ between runs, as
// with torch_scatter's own kernels).
#include "common.h"

namespace m3d {

// One workgroup per segment: thread t owns channel t % cw of row lane t / cw (cw = min(c, 256) rounded to a divisor
// of 256), walks rows lane, lane + lanes, ... in order, and the lanes are combined through LDS in lane order.
__global__ __launch_bounds__(256) void segment_csr_kernel(const float* __restrict__ src, int c,
                                                          const int64_t* __restrict__ indptr, int mean,
                                                          float* __restrict__ out) {
  __shared__ float part[256];
  const int64_t s = blockIdx.x;
  const int64_t lo = indptr[s], hi = indptr[s + 1];
  int cw = 1;
  while (cw < c && cw < 256) cw <<= 1;   // power of two <= 256 covering up to 256 channels per sweep
  const int lanes = 256 / cw;
  const int col = threadIdx.x % cw, lane = threadIdx.x / cw;
  const float inv = (mean && hi > lo) ? 1.f / (float)(hi - lo) : 1.f;
  for (int c0 = 0; c0 < c; c0 += cw) {
    const int ch = c0 + col;
    float acc = 0.f;
    if (ch < c)
      for (int64_t r = lo + lane; r < hi; r += lanes) acc += src[r * c + ch];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (lane == 0 && ch < c) {
      float v = 0.f;
      for (int l = 0; l < lanes; ++l) v += part[l * cw + col];
      out[s * c + ch] = v * inv;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void scatter_add_kernel(const float* __restrict__ src, const int64_t* __restrict__ index,
                                                          int64_t n, int c, int64_t n_out, float* __restrict__ out,
                                                          int32_t* __restrict__ count) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * c) return;
  const int64_t r = e / c;
  const int ch = (int)(e % c);
  const int64_t o = index[r];
  if (o < 0 || o >= n_out) return;
  atomicAdd(out + o * c + ch, src[e]);
  if (count && ch == 0) atomicAdd(count + o, 1);
}

__global__ __launch_bounds__(256) void scatter_mean_kernel(float* __restrict__ out, const int32_t* __restrict__ count,
                                                           int64_t n_out, int c) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out * c) return;
  const int k = count[e / c];
  if (k > 1) out[e] /= (float)k;
}

}  // namespace m3d

using namespace m3d;

extern "C" int m3d_segment_csr(const float* d_src, int c, const int64_t* d_indptr, int64_t n_segments, int mean,
                               float* d_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_segments == 0) return M3D_OK;
  M3D_CHECK_ARG(d_src && d_indptr && d_out && n_segments > 0 && n_segments < ((int64_t)1 << 31) && c >= 1,
                "m3d_segment_csr: bad arguments");
  hipLaunchKernelGGL(segment_csr_kernel, dim3((unsigned)n_segments), dim3(256), 0, stream, d_src, c, d_indptr, mean, d_out);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_scatter_rows(const float* d_src, const int64_t* d_index, int64_t n, int c, int64_t n_out, int mean,
                                float* d_out, int32_t* d_count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n >= 0 && n_out >= 0 && c >= 1, "m3d_scatter_rows: bad sizes");
  M3D_CHECK_ARG(!mean || d_count, "m3d_scatter_rows: a mean needs the count buffer");
  if (n_out == 0) return M3D_OK;
  M3D_CHECK_ARG(d_out && (n == 0 || (d_src && d_index)), "m3d_scatter_rows: null pointer");
  if (d_count) M3D_HIP(hipMemsetAsync(d_count, 0, (size_t)n_out * sizeof(int32_t), stream));
  if (n > 0)
    hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)cdiv(n * c, 256)), dim3(256), 0, stream, d_src, d_index, n, c, n_out,
                       d_out, d_count);
  if (mean)
    hipLaunchKernelGGL(scatter_mean_kernel, dim3((unsigned)cdiv(n_out * c, 256)), dim3(256), 0, stream, d_out, d_count, n_out, c);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}
