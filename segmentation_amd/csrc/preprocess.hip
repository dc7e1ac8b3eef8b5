// The steps in front of the voxelizer: Stray Scanner cloud -> Mosaic3D arrays (convert_stray.py:24-39) and the
// val-time transforms CenterShift + NormalizeColor + Collect(feat = color) of
// mosaic3d/src/data/utils/transform.py:149-163, :108-113, :23-46.  Elementwise work plus one min/max reduction.
#include "common.h"

#include <float.h>

namespace m3d {

// (x, y, z) -> (x, -z, y); colour in [0, 1] (float64) -> u8 by truncation of c * 255
__global__ void stray_convert_kernel(const double* __restrict__ pts, const double* __restrict__ col, int64_t n,
                                     float* __restrict__ coord, uint8_t* __restrict__ color) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
  coord[i * 3 + 0] = (float)x;
  coord[i * 3 + 1] = (float)(-z);
  coord[i * 3 + 2] = (float)y;
  if (col) {
#pragma unroll
    for (int a = 0; a < 3; ++a) color[i * 3 + a] = (uint8_t)(int)(col[i * 3 + a] * 255.0);
  }
}

// per-block min / max of the three coordinates -> part[block][6]
__global__ __launch_bounds__(256) void coord_minmax_kernel(const float* __restrict__ coord, int64_t n,
                                                           float* __restrict__ part) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = coord[i * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], s));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], s));
    }
  __shared__ float red[4][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[wave][a] = lo[a];
      red[wave][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int j = threadIdx.x;
    float v = red[0][j];
    for (int w = 1; w < 4; ++w) v = j < 3 ? fminf(v, red[w][j]) : fmaxf(v, red[w][j]);
    part[blockIdx.x * 6 + j] = v;
  }
}

// shift = ((xmin + xmax) / 2, (ymin + ymax) / 2, zmin or 0), all in float32 like the numpy original
__global__ void center_shift_kernel(const float* __restrict__ part, int nblocks, int apply_z, float* __restrict__ shift) {
  if (threadIdx.x != 0) return;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int b = 0; b < nblocks; ++b)
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], part[b * 6 + a]);
      hi[a] = fmaxf(hi[a], part[b * 6 + 3 + a]);
    }
  shift[0] = __fdiv_rn(__fadd_rn(lo[0], hi[0]), 2.f);
  shift[1] = __fdiv_rn(__fadd_rn(lo[1], hi[1]), 2.f);
  shift[2] = apply_z ? lo[2] : 0.f;
}

__global__ void shift_normalize_kernel(const float* __restrict__ coord, const uint8_t* __restrict__ color, int64_t n,
                                       const float* __restrict__ shift, float* __restrict__ coord_out,
                                       float* __restrict__ feat_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    coord_out[i * 3 + a] = __fsub_rn(coord[i * 3 + a], shift[a]);
    if (color) feat_out[i * 3 + a] = (float)((double)color[i * 3 + a] / 127.5 - 1.0);  // float64, then .float()
  }
}

}  // namespace m3d

using namespace m3d;

extern "C" int m3d_stray_convert(const double* d_points, const double* d_colors01, int64_t n, float* d_coord,
                                 uint8_t* d_color, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n >= 0, "m3d_stray_convert: n < 0");
  if (n == 0) return M3D_OK;
  M3D_CHECK_ARG(d_points && d_coord && ((d_colors01 == nullptr) == (d_color == nullptr)), "m3d_stray_convert: null pointer");
  hipLaunchKernelGGL(stray_convert_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, d_points, d_colors01, n,
                     d_coord, d_color);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" size_t m3d_center_shift_workspace_bytes(int64_t n) { return align_up(512 * 6 * sizeof(float)); }

extern "C" int m3d_center_shift_normalize(const float* d_coord, const uint8_t* d_color, int64_t n, int apply_z,
                                          float* d_coord_out, float* d_feat_out, float* d_shift, void* d_ws,
                                          size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n > 0, "m3d_center_shift_normalize: needs at least one point");
  M3D_CHECK_ARG(d_coord && d_coord_out && d_shift && ((d_color == nullptr) == (d_feat_out == nullptr)),
                "m3d_center_shift_normalize: null pointer");
  if (d_ws == nullptr || ws_bytes < m3d_center_shift_workspace_bytes(n)) {
    set_error("m3d_center_shift_normalize: workspace too small (%zu given, %zu needed)", ws_bytes,
              m3d_center_shift_workspace_bytes(n));
    return M3D_ENOSPC;
  }
  float* part = static_cast<float*>(d_ws);
  int nb = (int)cdiv(n, 256);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(coord_minmax_kernel, dim3(nb), dim3(256), 0, stream, d_coord, n, part);
  hipLaunchKernelGGL(center_shift_kernel, dim3(1), dim3(64), 0, stream, part, nb, apply_z, d_shift);
  hipLaunchKernelGGL(shift_normalize_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, d_coord, d_color, n,
                     d_shift, d_coord_out, d_feat_out);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}
