// Per-channel reductions of batch normalisation in training mode (PDBatchNorm with batch statistics over all
// rows of this GPU, mosaic3d/src/models/networks/spconv_unet_v1m3_pdnorm.py:57-68): the two passes over the [N, C]
// activations that are pure column reductions.
//
//   m3d_bn_stats            mean[c], biased variance[c] of x[:, c]
//   m3d_bn_backward_reduce  sum_dy[c] = sum_r dy[r][c],  sum_dy_xmu[c] = sum_r dy[r][c] * (x[r][c] - mean[c])
//
// Both are bound by HBM: a workgroup streams a contiguous range of rows as float4 (thread = fixed group of four
// channels, consecutive threads = consecutive 16-byte pieces of a row), adds in float64, combines its row groups
// through LDS and writes one partial per channel; a second, one-workgroup-per-channel-group pass adds the partials
// in a fixed order (deterministic, no atomics).  The elementwise passes (normalise, input gradient) are left to the
// caller.
#include "common.h"

#include <mutex>

namespace m3d {

constexpr int kNormThreads = 256;
constexpr int kNormMaxBlocks = 1024;

// Partial sums of m3d_bn_stats_static: a fixed array inside the code object, one copy per GPU, held until the
// process ends.  4 MiB = kNormMaxBlocks workgroups x 256 channels x 2 quantities x 8 bytes.  Host code never uses this
// name as an address: it asks the runtime for the device address once per device (static_scratch below).
constexpr size_t kStaticScratchBytes = (size_t)kNormMaxBlocks * 256 * 2 * sizeof(double);
static __device__ __align__(256) double scratch[kStaticScratchBytes / sizeof(double)];

// thread layout for c channels: c/4 float4 columns x R row groups, R = kNormThreads / (c/4)
template <int NQ>
__global__ __launch_bounds__(kNormThreads) void norm_partial_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ dy,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ ymask, int64_t n, int c,
                                                                    int64_t rows_per_block,
                                                                    double* __restrict__ partial) {
  // NQ = 2 quantities per channel: (sum x, sum x^2) when dy == nullptr, else (sum g, sum g*(x - mean)) with
  // g = dy, or dy where ymask > 0 and 0 elsewhere (the gradient behind a ReLU whose output is ymask)
  extern __shared__ double red[];  // [R][c][NQ]
  const int c4 = c >> 2;
  const int R = kNormThreads / c4;
  const int col = threadIdx.x % c4, rg = threadIdx.x / c4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  double acc[4][NQ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[i][q] = 0.0;
  if (rg < R) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dy) mu = *reinterpret_cast<const float4*>(mean + 4 * col);
#pragma unroll 4
    for (int64_t r = r0 + rg; r < r1; r += R) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * c + 4 * col);
      if (dy) {
        float4 g = *reinterpret_cast<const float4*>(dy + r * c + 4 * col);
        if (ymask) {
          const float4 y = *reinterpret_cast<const float4*>(ymask + r * c + 4 * col);
          g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f;
          g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
        }
        acc[0][0] += g.x; acc[0][1] += (double)g.x * (double)(v.x - mu.x);
        acc[1][0] += g.y; acc[1][1] += (double)g.y * (double)(v.y - mu.y);
        acc[2][0] += g.z; acc[2][1] += (double)g.z * (double)(v.z - mu.z);
        acc[3][0] += g.w; acc[3][1] += (double)g.w * (double)(v.w - mu.w);
      } else {
        acc[0][0] += v.x; acc[0][1] += (double)v.x * (double)v.x;
        acc[1][0] += v.y; acc[1][1] += (double)v.y * (double)v.y;
        acc[2][0] += v.z; acc[2][1] += (double)v.z * (double)v.z;
        acc[3][0] += v.w; acc[3][1] += (double)v.w * (double)v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < NQ; ++q) red[((int64_t)rg * c + 4 * col + i) * NQ + q] = acc[i][q];
  }
  __syncthreads();
  // row groups are added in index order: one thread per (channel, quantity)
  for (int e = threadIdx.x; e < c * NQ; e += kNormThreads) {
    double s = 0.0;
    for (int g = 0; g < R; ++g) s += red[(int64_t)g * c * NQ + e];
    partial[(int64_t)blockIdx.x * c * NQ + e] = s;
  }
}

// out0/out1 from the per-workgroup partials, added in workgroup order.
//   stats:   out0 = mean = S1 / n, out1 = biased variance = S2 / n - mean^2 (clamped at 0)
//   reduce:  out0 = sum_dy, out1 = sum_dy_xmu
__global__ __launch_bounds__(256) void norm_final_kernel(const double* __restrict__ partial, int nblocks, int c,
                                                         int64_t n, int stats, float* __restrict__ out0,
                                                         float* __restrict__ out1) {
  // a workgroup owns 8 channels = 16 consecutive doubles of every partial row; 16 groups of threads each add every
  // 16th workgroup's partial, then the 16 group sums are added in index order
  __shared__ double red[16][16];
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int ch0 = blockIdx.x * 8;
  const bool live = ch0 + (e >> 1) < c;
  double s = 0.0;
  if (live)
    for (int b = grp; b < nblocks; b += 16) s += partial[((int64_t)b * c + ch0) * 2 + e];
  red[grp][e] = s;
  __syncthreads();
  if (threadIdx.x < 8 && ch0 + threadIdx.x < c) {
    const int ch = ch0 + threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int g = 0; g < 16; ++g) {
      s0 += red[g][2 * threadIdx.x];
      s1 += red[g][2 * threadIdx.x + 1];
    }
    if (stats) {
      const double m = s0 / (double)n;
      const double v = s1 / (double)n - m * m;
      out0[ch] = (float)m;
      out1[ch] = (float)(v > 0.0 ? v : 0.0);
    } else {
      out0[ch] = (float)s0;
      out1[ch] = (float)s1;
    }
  }
}

// Elementwise passes of the fused normalise + residual + ReLU (BasicBlock.forward,
// spconv_unet_v1m3_pdnorm.py:121-140: relu(bn(conv(x))) and relu(bn(conv(x)) + residual)); one thread = four
// consecutive channels of one row, 16-byte accesses.
//   forward:  y = act(x * a[c] + b[c] + residual),  a = weight * invstd, b = bias - mean * a
//   backward: g = dy (where y > 0 behind a ReLU, else 0);  dres = g;
//             dx = (g - sum_g / n - (x - mean) * invstd^2 * sum_g_xmu / n) * invstd * weight
constexpr int kElemtIters = 8;  // float4 elements per thread; one block covers 256 * kElemtIters of them

// Per-channel coefficients are formed once per block in LDS (c <= 1024 channels), so the streaming loop holds no
// division, square root or 64-bit remainder: an element's channel group advances by 256 % c4 per iteration.
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, int64_t total4, int c4,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float eps, const float* __restrict__ weight,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ residual, int relu,
                                                         float* __restrict__ y) {
  extern __shared__ float4 coef[];  // [3][c4]: mean, scale, bias ((x - mean) first: no cancellation when |mean| >> std)
  float* cf = reinterpret_cast<float*>(coef);
  const int c = c4 * 4;
  for (int j = threadIdx.x; j < c; j += 256) {
    cf[j] = mean[j];
    cf[c + j] = (weight ? weight[j] : 1.f) * rsqrtf(var[j] + eps);
    cf[2 * c + j] = bias ? bias[j] : 0.f;
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * kElemtIters);
  int col = (int)((base + threadIdx.x) % c4);
  const int step = 256 % c4;
#pragma unroll
  for (int it = 0; it < kElemtIters; ++it) {
    const int64_t e = base + it * 256 + threadIdx.x;
    if (e < total4) {
      const float4 v = reinterpret_cast<const float4*>(x)[e];
      const float4 mu = coef[col], a = coef[c4 + col], b = coef[2 * c4 + col];
      float4 o = make_float4((v.x - mu.x) * a.x + b.x, (v.y - mu.y) * a.y + b.y, (v.z - mu.z) * a.z + b.z,
                             (v.w - mu.w) * a.w + b.w);
      if (residual) {
        const float4 r = reinterpret_cast<const float4*>(residual)[e];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
      reinterpret_cast<float4*>(y)[e] = o;
    }
    col += step;
    if (col >= c4) col -= c4;
  }
}

__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ ymask, int64_t total4, int c4,
                                                         double inv_n, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float eps,
                                                         const float* __restrict__ weight,
                                                         const float* __restrict__ sum_g,
                                                         const float* __restrict__ sum_g_xmu, float* __restrict__ dx,
                                                         float* __restrict__ dres) {
  // dx = (g - m - (x - mean) * k) * s = g * s - (x - mean) * (k * s) - m * s
  extern __shared__ float4 coef[];  // [4][c4]: s, k * s, mean, m * s
  float* cf = reinterpret_cast<float*>(coef);
  const int c = c4 * 4;
  for (int j = threadIdx.x; j < c; j += 256) {
    const float invstd = rsqrtf(var[j] + eps);
    const float k = (float)((double)sum_g_xmu[j] * inv_n) * invstd * invstd;
    const float m = (float)((double)sum_g[j] * inv_n);
    const float sc = invstd * (weight ? weight[j] : 1.f);
    cf[j] = sc;
    cf[c + j] = k * sc;
    cf[2 * c + j] = mean[j];
    cf[3 * c + j] = m * sc;
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * kElemtIters);
  int col = (int)((base + threadIdx.x) % c4);
  const int step = 256 % c4;
#pragma unroll
  for (int it = 0; it < kElemtIters; ++it) {
    const int64_t e = base + it * 256 + threadIdx.x;
    if (e < total4) {
      const float4 xv = reinterpret_cast<const float4*>(x)[e];
      float4 g = reinterpret_cast<const float4*>(dy)[e];
      if (ymask) {
        const float4 yv = reinterpret_cast<const float4*>(ymask)[e];
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
        g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      }
      if (dres) reinterpret_cast<float4*>(dres)[e] = g;
      const float4 s = coef[col], ks = coef[c4 + col], mu = coef[2 * c4 + col], t = coef[3 * c4 + col];
      reinterpret_cast<float4*>(dx)[e] =
          make_float4(g.x * s.x - (xv.x - mu.x) * ks.x - t.x, g.y * s.y - (xv.y - mu.y) * ks.y - t.y,
                      g.z * s.z - (xv.z - mu.z) * ks.z - t.z, g.w * s.w - (xv.w - mu.w) * ks.w - t.w);
    }
    col += step;
    if (col >= c4) col -= c4;
  }
}

static int norm_blocks(int64_t n, int c) {
  // about 64 KB of input per workgroup, at most kNormMaxBlocks partials
  int64_t rows = (int64_t)65536 / ((int64_t)c * 4);
  if (rows < 8) rows = 8;
  int64_t b = cdiv(n, rows);
  if (b > kNormMaxBlocks) b = kNormMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

static int norm_launch(const char* who, const float* x, const float* dy, const float* mean, int64_t n, int c,
                       float* out0, float* out1, void* d_ws, size_t ws_bytes, hipStream_t stream,
                       const float* ymask = nullptr) {
  M3D_CHECK_ARG(n > 0 && x && out0 && out1, "%s: null pointer or no rows", who);
  M3D_CHECK_ARG(c >= 4 && c % 4 == 0 && c <= 4 * kNormThreads, "%s: channel count %d must be a multiple of 4, at most %d",
                who, c, 4 * kNormThreads);
  M3D_CHECK_ARG(((uintptr_t)x & 15) == 0 && (dy == nullptr || ((uintptr_t)dy & 15) == 0) &&
                    (mean == nullptr || ((uintptr_t)mean & 15) == 0) && (ymask == nullptr || ((uintptr_t)ymask & 15) == 0),
                "%s: pointers must be 16-byte aligned", who);
  const int nb = norm_blocks(n, c);
  const size_t need = (size_t)nb * c * 2 * sizeof(double);
  if (d_ws == nullptr || ws_bytes < need) {
    set_error("%s: workspace too small (%zu given, %zu needed)", who, ws_bytes, need);
    return M3D_ENOSPC;
  }
  const int R = kNormThreads / (c / 4);
  const size_t lds = (size_t)R * c * 2 * sizeof(double);  // <= 64 rows x 16 B x ... = 16 KB at most
  const int64_t rows_per_block = cdiv(n, nb);
  double* partial = static_cast<double*>(d_ws);
  hipLaunchKernelGGL((norm_partial_kernel<2>), dim3(nb), dim3(kNormThreads), lds, stream, x, dy, mean, ymask, n, c,
                     rows_per_block, partial);
  hipLaunchKernelGGL(norm_final_kernel, dim3((unsigned)cdiv(c, 8)), dim3(256), 0, stream, partial, nb, c, n,
                     dy == nullptr ? 1 : 0, out0, out1);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

// Device address of `scratch` on the current device, looked up once per device.
static int static_scratch(const char* who, double** out) {
  constexpr int kMaxDevices = 64;
  static std::mutex mu;
  static double* ptr[kMaxDevices] = {};
  int dev = -1;
  M3D_HIP(hipGetDevice(&dev));
  M3D_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "%s: device index %d out of range", who, dev);
  std::lock_guard<std::mutex> lock(mu);
  if (ptr[dev] == nullptr) {
    void* p = nullptr;
    M3D_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    M3D_CHECK_ARG(p != nullptr, "%s: no device address for the static scratch array", who);
    ptr[dev] = static_cast<double*>(p);
  }
  *out = ptr[dev];
  return M3D_OK;
}

}  // namespace m3d

using namespace m3d;

extern "C" size_t m3d_bn_workspace_bytes(int64_t n, int c) {
  if (n <= 0 || c <= 0) return 256;
  return align_up((size_t)norm_blocks(n, c) * c * 2 * sizeof(double));
}

extern "C" int m3d_bn_stats(const float* d_x, int64_t n, int c, float* d_mean, float* d_var, void* d_ws,
                            size_t ws_bytes, void* stream) {
  return norm_launch("m3d_bn_stats", d_x, nullptr, nullptr, n, c, d_mean, d_var, d_ws, ws_bytes,
                     static_cast<hipStream_t>(stream));
}

extern "C" int m3d_bn_backward_reduce(const float* d_x, const float* d_dy, const float* d_mean, int64_t n, int c,
                                      float* d_sum_dy, float* d_sum_dy_xmu, void* d_ws, size_t ws_bytes,
                                      void* stream) {
  M3D_CHECK_ARG(d_dy && d_mean, "m3d_bn_backward_reduce: null pointer");
  return norm_launch("m3d_bn_backward_reduce", d_x, d_dy, d_mean, n, c, d_sum_dy, d_sum_dy_xmu, d_ws, ws_bytes,
                     static_cast<hipStream_t>(stream));
}

extern "C" int m3d_bn_act_backward_reduce(const float* d_x, const float* d_dy, const float* d_y, const float* d_mean,
                                          int64_t n, int c, float* d_sum_g, float* d_sum_g_xmu, void* d_ws,
                                          size_t ws_bytes, void* stream) {
  M3D_CHECK_ARG(d_dy && d_mean, "m3d_bn_act_backward_reduce: null pointer");
  return norm_launch("m3d_bn_act_backward_reduce", d_x, d_dy, d_mean, n, c, d_sum_g, d_sum_g_xmu, d_ws, ws_bytes,
                     static_cast<hipStream_t>(stream), d_y);
}

static int elemt_check(const char* who, int64_t n, int c, const void* a, const void* b, const void* cc, const void* d,
                       const void* e) {
  M3D_CHECK_ARG(n > 0 && c >= 4 && c % 4 == 0 && c <= 1024,
                "%s: needs rows and a channel count that is a multiple of 4, at most 1024", who);
  const void* ps[5] = {a, b, cc, d, e};
  for (const void* p : ps) M3D_CHECK_ARG(((uintptr_t)p & 15) == 0, "%s: pointers must be 16-byte aligned", who);
  return M3D_OK;
}

extern "C" int m3d_bn_act_fwd(const float* d_x, int64_t n, int c, const float* d_mean, const float* d_var, float eps,
                              const float* d_weight, const float* d_bias, const float* d_residual, int relu,
                              float* d_y, void* stream_) {
  const char* who = "m3d_bn_act_fwd";
  M3D_CHECK_ARG(d_x && d_mean && d_var && d_y, "%s: null pointer", who);
  if (int rc = elemt_check(who, n, c, d_x, d_y, d_residual, nullptr, nullptr)) return rc;
  const int64_t total4 = n * (c / 4);
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((unsigned)cdiv(total4, 256 * kElemtIters)), dim3(256),
                     (size_t)3 * c * sizeof(float), static_cast<hipStream_t>(stream_),
                     d_x, total4, c / 4, d_mean, d_var, eps, d_weight, d_bias, d_residual, relu, d_y);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_bn_act_backward(const float* d_x, const float* d_dy, const float* d_y, int64_t n, int c,
                                   const float* d_mean, const float* d_var, float eps, const float* d_weight,
                                   const float* d_sum_g, const float* d_sum_g_xmu, float* d_dx, float* d_dres,
                                   void* stream_) {
  const char* who = "m3d_bn_act_backward";
  M3D_CHECK_ARG(d_x && d_dy && d_mean && d_var && d_sum_g && d_sum_g_xmu && d_dx, "%s: null pointer", who);
  if (int rc = elemt_check(who, n, c, d_x, d_dy, d_y, d_dx, d_dres)) return rc;
  const int64_t total4 = n * (c / 4);
  hipLaunchKernelGGL(bn_act_bwd_kernel, dim3((unsigned)cdiv(total4, 256 * kElemtIters)), dim3(256),
                     (size_t)4 * c * sizeof(float), static_cast<hipStream_t>(stream_),
                     d_x, d_dy, d_y, total4, c / 4, 1.0 / (double)n, d_mean, d_var, eps, d_weight, d_sum_g, d_sum_g_xmu,
                     d_dx, d_dres);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" size_t m3d_bn_stats_static_max_bytes(void) { return kStaticScratchBytes; }

extern "C" int m3d_bn_stats_static(const float* d_x, int64_t n, int c, float* d_mean, float* d_var, void* stream) {
  const char* who = "m3d_bn_stats_static";
  M3D_CHECK_ARG(n > 0 && d_x && d_mean && d_var, "%s: null pointer or no rows", who);
  M3D_CHECK_ARG(c >= 4 && c % 4 == 0 && c <= 4 * kNormThreads, "%s: channel count %d must be a multiple of 4, at most %d",
                who, c, 4 * kNormThreads);
  const size_t need = m3d_bn_workspace_bytes(n, c);
  if (need > kStaticScratchBytes) {
    set_error("%s: %lld rows x %d channels need %zu bytes of partial sums, the static scratch holds %zu "
              "(use m3d_bn_stats with a workspace)", who, (long long)n, c, need, kStaticScratchBytes);
    return M3D_ENOSPC;
  }
  double* ws = nullptr;
  if (int rc = static_scratch(who, &ws)) return rc;
  return norm_launch(who, d_x, nullptr, nullptr, n, c, d_mean, d_var, ws, kStaticScratchBytes,
                     static_cast<hipStream_t>(stream));
}
