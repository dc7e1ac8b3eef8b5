// Open-vocabulary head tail: row argmax of the per-voxel logits and the gather of voxel labels
// to points.  The logits GEMM itself runs on the convolution kernel (dense 1x1 mode).
#include "common.h"

namespace m3d {

// 8 lanes per row; ties resolve to the smallest column index (torch.max(1) on CPU).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, int64_t n,
                                                          int ncls, int64_t ld, int32_t* __restrict__ label) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid >> 3;
  const int sub = (int)(gid & 7);
  float best = -__builtin_inff();
  int bi = 0x7FFFFFFF;
  if (row < n) {
    const float* p = logits + row * ld;
    for (int c = sub; c < ncls; c += 8) {
      const float v = p[c];
      if (v > best || bi == 0x7FFFFFFF) { best = v; bi = c; }
    }
  }
#pragma unroll
  for (int s = 1; s < 8; s <<= 1) {
    const float ov = __shfl_xor(best, s);
    const int oi = __shfl_xor(bi, s);
    if (oi != 0x7FFFFFFF && (bi == 0x7FFFFFFF || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
  }
  if (row < n && sub == 0) label[row] = bi;
}

__global__ void gather_labels_kernel(const int32_t* __restrict__ vlabel, const int64_t* __restrict__ v2p,
                                     int64_t n, int64_t* __restrict__ plabel) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) plabel[i] = vlabel[v2p[i]];
}

}  // namespace m3d

using namespace m3d;

extern "C" int m3d_argmax_rows(const float* d_logits, int64_t n_vox, int n_classes, int64_t ldl,
                               int32_t* d_voxel_label, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_vox == 0) return M3D_OK;
  M3D_CHECK_ARG(d_logits && d_voxel_label, "m3d_argmax_rows: null pointer");
  M3D_CHECK_ARG(n_vox > 0 && n_classes >= 1 && ldl >= n_classes, "m3d_argmax_rows: bad sizes");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)cdiv(n_vox * 8, 256)), dim3(256), 0, stream,
                     d_logits, n_vox, n_classes, ldl, d_voxel_label);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_gather_labels(const int32_t* d_voxel_label, const int64_t* d_v2p_map, int64_t n_pts,
                                 int64_t* d_point_label, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_pts == 0) return M3D_OK;
  M3D_CHECK_ARG(d_voxel_label && d_v2p_map && d_point_label && n_pts > 0, "m3d_gather_labels: bad arguments");
  hipLaunchKernelGGL(gather_labels_kernel, dim3((unsigned)cdiv(n_pts, 256)), dim3(256), 0, stream,
                     d_voxel_label, d_v2p_map, n_pts, d_point_label);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------ average pooling
// out[o][c] = mean over the kernel offsets k with pair[k][o] >= 0 of in[pair[k][o]][c]; rows without any pair
// are zero.  One thread per (output row, channel); the pair column is read once per row from L1.
namespace m3d {
__global__ void avgpool_kernel(const float* __restrict__ in, int c, const int32_t* __restrict__ pair, int k, int64_t ld,
                               int64_t n_out, float* __restrict__ out, int32_t* __restrict__ count) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out * c) return;
  const int64_t o = e / c;
  const int ch = (int)(e % c);
  float s = 0.f;
  int cnt = 0;
  for (int kk = 0; kk < k; ++kk) {
    const int i = pair[(int64_t)kk * ld + o];
    if (i >= 0) {
      s += in[(int64_t)i * c + ch];
      ++cnt;
    }
  }
  out[e] = cnt ? s / (float)cnt : 0.f;
  if (count && ch == 0) count[o] = cnt;
}
}  // namespace m3d

extern "C" int m3d_avgpool_fwd(const float* d_in, int c, const int32_t* d_pair, int k, int64_t ld, int64_t n_out,
                               float* d_out, int32_t* d_count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_out == 0) return M3D_OK;
  M3D_CHECK_ARG(d_in && d_pair && d_out && n_out > 0 && c >= 1 && k >= 1 && ld >= n_out, "m3d_avgpool_fwd: bad arguments");
  hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)cdiv(n_out * c, 256)), dim3(256), 0, stream, d_in, c, d_pair, k, ld,
                     n_out, d_out, d_count);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}
