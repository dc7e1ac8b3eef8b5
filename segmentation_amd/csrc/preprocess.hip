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

// ------------------------------------------------------------------------------------------ depth back-projection
// Stray Scanner frame -> world points, fused with the converter's Y-up -> Z-up rotation, compacted in row-major
// pixel order (wave ballot + prefix popcount inside a workgroup, one prefix over the workgroups).
namespace m3d {

struct BackprojArgs {
  const uint16_t* depth;   // [h, w] raw depth units
  const uint8_t* conf;     // [h, w] or nullptr
  const uint8_t* rgb;      // [hc, wc, 3] or nullptr
  int h, w, hc, wc, stride;
  double fx, fy, cx, cy;   // intrinsics at depth resolution
  double pose[12];         // camera-to-world, rows of [R | t]
  double depth_scale, depth_max;
  int conf_min;
};

__device__ inline bool backproj_valid(const BackprojArgs& a, int64_t p, int* u, int* v) {
  const int wq = (a.w + a.stride - 1) / a.stride, hq = (a.h + a.stride - 1) / a.stride;
  if (p >= (int64_t)wq * hq) return false;
  *v = (int)(p / wq) * a.stride;
  *u = (int)(p % wq) * a.stride;
  const uint16_t d = a.depth[(int64_t)*v * a.w + *u];
  if (d == 0) return false;
  if (a.conf && a.conf[(int64_t)*v * a.w + *u] < a.conf_min) return false;
  return (double)d * a.depth_scale <= a.depth_max;
}

__global__ __launch_bounds__(256) void backproj_count_kernel(BackprojArgs a, int* __restrict__ bsum) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int u, v;
  const bool ok = backproj_valid(a, p, &u, &v);
  const int c = __popcll(__ballot(ok));
  __shared__ int ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// one workgroup: exclusive scan of the block counts in place; *total receives base + sum
__global__ __launch_bounds__(256) void backproj_scan_kernel(int* __restrict__ bsum, int nb, const int64_t* __restrict__ base,
                                                            int64_t capacity, int64_t* __restrict__ total,
                                                            int* __restrict__ overflow) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int i = b0 + threadIdx.x;
    const int val = i < nb ? bsum[i] : 0;
    int inc = val;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (i < nb) bsum[i] = before + inc - val;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int64_t t = *base + carry_s;
    *overflow = t > capacity ? 1 : 0;
    *total = t > capacity ? *base : t;  // a frame that does not fit is dropped as a whole
  }
}

__global__ __launch_bounds__(256) void backproj_emit_kernel(BackprojArgs a, const int* __restrict__ bsum,
                                                            const int64_t* __restrict__ base, const int* __restrict__ overflow,
                                                            float* __restrict__ coord, uint8_t* __restrict__ color) {
  if (*overflow) return;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int u = 0, v = 0;
  const bool ok = backproj_valid(a, p, &u, &v);
  const unsigned long long bal = __ballot(ok);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int ws[4];
  if (lane == 0) ws[wave] = __popcll(bal);
  __syncthreads();
  int before = bsum[blockIdx.x];
  for (int w = 0; w < wave; ++w) before += ws[w];
  if (!ok) return;
  const int64_t o = *base + before + __popcll(bal & ((1ull << lane) - 1));
  const double d = (double)a.depth[(int64_t)v * a.w + u] * a.depth_scale;
  const double xc = ((double)u - a.cx) / a.fx * d, yc = ((double)v - a.cy) / a.fy * d, zc = d;
  const double xw = a.pose[0] * xc + a.pose[1] * yc + a.pose[2] * zc + a.pose[3];
  const double yw = a.pose[4] * xc + a.pose[5] * yc + a.pose[6] * zc + a.pose[7];
  const double zw = a.pose[8] * xc + a.pose[9] * yc + a.pose[10] * zc + a.pose[11];
  coord[o * 3 + 0] = (float)xw;  // (x, y, z) -> (x, -z, y)
  coord[o * 3 + 1] = (float)(-zw);
  coord[o * 3 + 2] = (float)yw;
  if (a.rgb) {
    const int uc = min((int)((int64_t)u * a.wc / a.w), a.wc - 1), vc = min((int)((int64_t)v * a.hc / a.h), a.hc - 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) color[o * 3 + ch] = a.rgb[((int64_t)vc * a.wc + uc) * 3 + ch];
  }
}

}  // namespace m3d

extern "C" size_t m3d_backproject_workspace_bytes(int h, int w, int stride) {
  if (stride < 1) stride = 1;
  const int64_t px = (int64_t)((w + stride - 1) / stride) * ((h + stride - 1) / stride);
  return align_up((size_t)cdiv(px, 256) * sizeof(int)) + align_up(64);
}

extern "C" int m3d_backproject(const uint16_t* d_depth, const uint8_t* d_conf, const uint8_t* d_rgb, int h, int w,
                               int hc, int wc, const double intrinsics[4], const double pose_c2w[16],
                               double depth_scale, double depth_max, int conf_min, int stride, float* d_coord,
                               uint8_t* d_color, int64_t capacity, int64_t* d_count, void* d_ws, size_t ws_bytes,
                               void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(d_depth && d_coord && d_count && intrinsics && pose_c2w, "m3d_backproject: null pointer");
  M3D_CHECK_ARG(h > 0 && w > 0 && stride >= 1 && capacity > 0, "m3d_backproject: bad sizes");
  M3D_CHECK_ARG((d_rgb == nullptr) == (d_color == nullptr), "m3d_backproject: rgb and color go together");
  M3D_CHECK_ARG(d_rgb == nullptr || (hc > 0 && wc > 0), "m3d_backproject: bad colour image size");
  M3D_CHECK_ARG(intrinsics[0] != 0.0 && intrinsics[1] != 0.0 && depth_scale > 0.0, "m3d_backproject: bad intrinsics");
  if (d_ws == nullptr || ws_bytes < m3d_backproject_workspace_bytes(h, w, stride)) {
    set_error("m3d_backproject: workspace too small (%zu given, %zu needed)", ws_bytes,
              m3d_backproject_workspace_bytes(h, w, stride));
    return M3D_ENOSPC;
  }
  BackprojArgs a;
  a.depth = d_depth; a.conf = d_conf; a.rgb = d_rgb;
  a.h = h; a.w = w; a.hc = hc; a.wc = wc; a.stride = stride;
  a.fx = intrinsics[0]; a.fy = intrinsics[1]; a.cx = intrinsics[2]; a.cy = intrinsics[3];
  for (int i = 0; i < 12; ++i) a.pose[i] = pose_c2w[i];
  a.depth_scale = depth_scale; a.depth_max = depth_max; a.conf_min = conf_min;
  const int64_t px = (int64_t)((w + stride - 1) / stride) * ((h + stride - 1) / stride);
  const int nb = (int)cdiv(px, 256);
  int* bsum = static_cast<int*>(d_ws);
  int* overflow = reinterpret_cast<int*>(static_cast<char*>(d_ws) + align_up((size_t)nb * sizeof(int)));
  hipLaunchKernelGGL(backproj_count_kernel, dim3(nb), dim3(256), 0, stream, a, bsum);
  // the emit kernel needs the count BEFORE this frame as its base: keep it in the workspace
  int64_t* base = reinterpret_cast<int64_t*>(overflow + 2);
  M3D_HIP(hipMemcpyAsync(base, d_count, sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(backproj_scan_kernel, dim3(1), dim3(256), 0, stream, bsum, nb, base, capacity, d_count, overflow);
  hipLaunchKernelGGL(backproj_emit_kernel, dim3(nb), dim3(256), 0, stream, a, bsum, base, overflow, d_coord, d_color);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}
