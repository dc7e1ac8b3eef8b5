// Voxelizer: points -> grid coordinates -> 64-bit hash keys -> unique voxels in ascending key
// order, per-voxel mean features and the voxel row of every point.
//
// Semantics follow the reference's Point.sparsify / mean_pooling / fnv_hash_vec
// (mosaic3d/src/models/utils/structure.py:197-251, :34-66, :21-31): trunc-toward-zero grid rule,
// minimum taken over the whole batch, FNV64-1A over the (batch, x, y, z) columns, voxel order =
// ascending key.  Data layout in HBM: structure-of-rows exactly as the reference tensors
// ([n,3] f32 coords, [n,c] f32 feats, [n,4] i32 voxel coords), so the outputs are torch views.
#include "common.h"

#include <limits.h>

namespace m3d {

constexpr uint64_t kFnvOffset = 14695981039346656037ull;
constexpr uint64_t kFnvPrime = 1099511628211ull;

constexpr int kGridFloor64 = 3;  // GridSample's rule: floor(float64(x) / float64(grid_size)) (transform.py:753-754)
constexpr int kHashFnv3 = 2;     // FNV64-1A over (x, y, z) only (transform.py:876-887)

__device__ inline int grid_of(float x, float gs, float inv_gs, int mode, double gs64) {
  if (mode == kGridFloor64) return (int)floor((double)x / gs64);
  float q;
  if (mode == M3D_GRID_RECIP) q = x * inv_gs;
  else q = __fdiv_rn(x, gs);
  if (mode == M3D_GRID_FLOOR) q = floorf(q);
  return (int)q;  // truncation toward zero
}

__device__ inline int batch_of(const int64_t* __restrict__ offset, int b, int64_t i) {
  // largest j in [0, b) with offset[j] <= i
  int lo = 0, hi = b - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (offset[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// mm[0..2] = min grid, mm[3..5] = max grid, mm[6] = min batch, mm[7] = max batch
__global__ void vox_init_kernel(int* mm) {
  const int t = threadIdx.x;
  if (t < 3) mm[t] = INT_MAX;
  else if (t < 6) mm[t] = INT_MIN;
  else if (t == 6) mm[t] = INT_MAX;
  else if (t == 7) mm[t] = INT_MIN;
}

__global__ __launch_bounds__(256) void vox_minmax_kernel(const float* __restrict__ coord,
                                                         const int64_t* __restrict__ offset, int64_t n,
                                                         int b, float gs, float inv_gs, int mode,
                                                         double gs64, int* __restrict__ mm) {
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  int blo = INT_MAX, bhi = INT_MIN;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int g = grid_of(coord[i * 3 + a], gs, inv_gs, mode, gs64);
      lo[a] = min(lo[a], g);
      hi[a] = max(hi[a], g);
    }
    int bi = batch_of(offset, b, i);
    blo = min(blo, bi);
    bhi = max(bhi, bi);
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], s));
      hi[a] = max(hi[a], __shfl_xor(hi[a], s));
    }
    blo = min(blo, __shfl_xor(blo, s));
    bhi = max(bhi, __shfl_xor(bhi, s));
  }
  __shared__ int red[4][8];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[wave][a] = lo[a];
      red[wave][3 + a] = hi[a];
    }
    red[wave][6] = blo;
    red[wave][7] = bhi;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int j = threadIdx.x;
    const bool is_min = j < 3 || j == 6;
    int v = red[0][j];
    for (int w = 1; w < 4; ++w) v = is_min ? min(v, red[w][j]) : max(v, red[w][j]);
    if (is_min) atomicMin(&mm[j], v); else atomicMax(&mm[j], v);
  }
}

__global__ __launch_bounds__(256) void vox_key_kernel(const float* __restrict__ coord,
                                                      const int64_t* __restrict__ offset, int64_t n,
                                                      int b, float gs, float inv_gs, int mode, double gs64,
                                                      int hash_method, const int* __restrict__ mm,
                                                      uint64_t* __restrict__ keys,
                                                      int4* __restrict__ bc,
                                                      int32_t* __restrict__ grid_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int g[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] = grid_of(coord[i * 3 + a], gs, inv_gs, mode, gs64) - mm[a];
  const int bi = batch_of(offset, b, i);
  bc[i] = make_int4(bi, g[0], g[1], g[2]);
  if (grid_out) {
    grid_out[i * 3 + 0] = g[0];
    grid_out[i * 3 + 1] = g[1];
    grid_out[i * 3 + 2] = g[2];
  }
  uint64_t h;
  if (hash_method == kHashFnv3) {
    h = kFnvOffset;
    h *= kFnvPrime; h ^= (uint64_t)(int64_t)g[0];
    h *= kFnvPrime; h ^= (uint64_t)(int64_t)g[1];
    h *= kFnvPrime; h ^= (uint64_t)(int64_t)g[2];
  } else if (hash_method == M3D_HASH_FNV) {
    h = kFnvOffset;
    h *= kFnvPrime; h ^= (uint64_t)(int64_t)bi;
    h *= kFnvPrime; h ^= (uint64_t)(int64_t)g[0];
    h *= kFnvPrime; h ^= (uint64_t)(int64_t)g[1];
    h *= kFnvPrime; h ^= (uint64_t)(int64_t)g[2];
  } else {
    // row-major linear index of (b - bmin, x, y, z) over shape (max - min + 1)
    const uint64_t s1 = (uint64_t)(mm[3] - mm[0] + 1), s2 = (uint64_t)(mm[4] - mm[1] + 1),
                   s3 = (uint64_t)(mm[5] - mm[2] + 1);
    h = (((uint64_t)(bi - mm[6]) * s1 + (uint64_t)g[0]) * s2 + (uint64_t)g[1]) * s3 + (uint64_t)g[2];
  }
  keys[i] = h;
}

__global__ void vox_coords_kernel(const int32_t* __restrict__ rep, const int4* __restrict__ bc,
                                  const int* __restrict__ d_count, int4* __restrict__ voxel_coords) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < *d_count) voxel_coords[r] = bc[rep[r]];
}

// Mean pooling without an atomic per feature: every point pushes itself onto its voxel's list (one exchange per
// point; scattered device-scope atomics are what bounds this stage), then one thread per (voxel, channel) walks the
// list.  The walk adds in float64, so the float32 mean does not depend on the order the points arrived in.
__global__ void vox_link_kernel(const int32_t* __restrict__ inverse, int64_t n, int32_t* __restrict__ head,
                                int32_t* __restrict__ next, int64_t* __restrict__ v2p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = inverse[i];
  v2p[i] = r;
  next[i] = atomicExch(&head[r], (int32_t)i);
}

__global__ void vox_mean_kernel(const float* __restrict__ feat, const int32_t* __restrict__ head,
                                const int32_t* __restrict__ next, const int* __restrict__ d_count, int c,
                                float* __restrict__ voxel_feats, int32_t* __restrict__ cnt) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cc = c > 0 ? c : 1;  // c == 0: counts only
  if (t >= (int64_t)(*d_count) * cc) return;
  const int64_t r = t / cc;
  const int j = (int)(t % cc);
  double sum = 0.0;
  int m = 0;
  for (int32_t p = head[r]; p >= 0; p = next[p]) {
    if (c > 0) sum += (double)feat[(int64_t)p * c + j];
    ++m;
  }
  if (c > 0) voxel_feats[t] = (float)(sum / (double)m);
  if (j == 0) cnt[r] = m;
}

__global__ void vox_meta_kernel(const int* __restrict__ mm, const int* __restrict__ d_count, int pad,
                                int64_t* __restrict__ meta) {
  if (threadIdx.x != 0) return;
  meta[0] = *d_count;
  int64_t bad = 0;
  for (int a = 0; a < 3; ++a) {
    const int64_t extent = (int64_t)mm[3 + a] - (int64_t)mm[a];  // in 64 bits: infinite or huge coordinates saturate
    meta[1 + a] = extent + pad;                                   // the 32-bit grid coordinates at both ends
    meta[4 + a] = mm[a];
    meta[7 + a] = extent;
    if (extent + pad >= (int64_t)INT_MAX) bad = 1;  // grid coordinates relative to the minimum no longer fit 32 bits
  }
  meta[10] = mm[6];
  meta[11] = mm[7];
  meta[12] = bad;
}

struct VoxWs {
  uint64_t* keys;
  int4* bc;
  int32_t* rep;
  int32_t* inverse;
  double* sums;
  int32_t* cnt;
  uint64_t* ukeys;
  int* count;
  int* mm;
  char* uniq;
  size_t uniq_bytes;
};

static size_t vox_layout(int64_t n, int c, void* ws, size_t bytes, VoxWs* out) {
  if (n <= 0) n = 1;
  Arena a(ws, bytes);
  VoxWs w;
  w.keys = a.take<uint64_t>(n);
  w.bc = a.take<int4>(n);
  w.rep = a.take<int32_t>(n);
  w.inverse = a.take<int32_t>(n);
  w.sums = a.take<double>((size_t)n * c);
  w.cnt = a.take<int32_t>(n);
  w.ukeys = a.take<uint64_t>(n);
  w.count = a.take<int>(64);
  w.mm = a.take<int>(64);
  w.uniq_bytes = unique_ws_bytes(n);
  w.uniq = a.take<char>(w.uniq_bytes);
  if (out) *out = w;
  return a.off;  // bytes required (valid pointers only if this fits in `bytes`)
}

}  // namespace m3d

using namespace m3d;

extern "C" size_t m3d_voxelize_workspace_bytes(int64_t n, int c) {
  return vox_layout(n, c < 1 ? 1 : c, nullptr, 0, nullptr);
}

__global__ void copy_counts_kernel(const int32_t* __restrict__ cnt, const int* __restrict__ d_count,
                                   int32_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < *d_count) out[r] = cnt[r];
}

// Shared body of m3d_voxelize (c >= 1 features, float32 grid rule) and m3d_grid_sample (c = 0, float64 rule).
static int voxelize_impl(const char* who, const float* d_coord, const float* d_feat, const int64_t* d_offset,
                         int64_t n, int c, int b, float grid_size, double gs64, int pad, int hash_method,
                         int grid_mode, int32_t* d_voxel_coords, float* d_voxel_feats, int64_t* d_v2p_map,
                         uint64_t* d_keys, int32_t* d_grid_coord, int32_t* d_count_out, int64_t* d_meta,
                         void* d_ws, size_t ws_bytes, hipStream_t stream) {
  VoxWs w;
  size_t need = vox_layout(n, c < 1 ? 1 : c, d_ws, ws_bytes, &w);
  if (need > ws_bytes || d_ws == nullptr) {
    set_error("%s: workspace too small (%zu given, %zu needed)", who, ws_bytes, need);
    return M3D_ENOSPC;
  }
  uint64_t* ukeys = d_keys ? d_keys : w.ukeys;
  const float inv_gs = 1.0f / grid_size;
  const int bs = 256;
  const unsigned gb = (unsigned)cdiv(n, bs);
  hipLaunchKernelGGL(vox_init_kernel, dim3(1), dim3(64), 0, stream, w.mm);
  hipLaunchKernelGGL(vox_minmax_kernel, dim3(gb < 512 ? gb : 512), dim3(bs), 0, stream, d_coord,
                     d_offset, n, b, grid_size, inv_gs, grid_mode, gs64, w.mm);
  hipLaunchKernelGGL(vox_key_kernel, dim3(gb), dim3(bs), 0, stream, d_coord, d_offset, n, b,
                     grid_size, inv_gs, grid_mode, gs64, hash_method, w.mm, w.keys, w.bc, d_grid_coord);
  M3D_LAUNCH_CHECK();
  int rc = unique_sorted(w.keys, n, 64, ukeys, w.rep, w.inverse, w.count, w.uniq, w.uniq_bytes, stream);
  if (rc != M3D_OK) return rc;
  // list heads and links live where the per-voxel sums used to: 2 * n int32 <= n * max(c, 1) doubles
  int32_t* head = reinterpret_cast<int32_t*>(w.sums);
  int32_t* next = head + n;
  M3D_HIP(hipMemsetAsync(head, 0xFF, (size_t)n * sizeof(int32_t), stream));
  hipLaunchKernelGGL(vox_coords_kernel, dim3(gb), dim3(bs), 0, stream, w.rep, w.bc, w.count,
                     reinterpret_cast<int4*>(d_voxel_coords));
  hipLaunchKernelGGL(vox_link_kernel, dim3(gb), dim3(bs), 0, stream, w.inverse, n, head, next, d_v2p_map);
  hipLaunchKernelGGL(vox_mean_kernel, dim3((unsigned)cdiv(n * (c > 0 ? c : 1), bs)), dim3(bs), 0, stream, d_feat, head,
                     next, w.count, c, d_voxel_feats, w.cnt);
  if (d_count_out) hipLaunchKernelGGL(copy_counts_kernel, dim3(gb), dim3(bs), 0, stream, w.cnt, w.count, d_count_out);
  hipLaunchKernelGGL(vox_meta_kernel, dim3(1), dim3(64), 0, stream, w.mm, w.count, pad, d_meta);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_voxelize(const float* d_coord, const float* d_feat, const int64_t* d_offset,
                            int64_t n, int c, int b, float grid_size, int pad, int hash_method,
                            int grid_mode, int32_t* d_voxel_coords, float* d_voxel_feats,
                            int64_t* d_v2p_map, uint64_t* d_keys, int32_t* d_grid_coord,
                            int64_t* d_meta, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n > 0 && n < (int64_t)INT_MAX, "m3d_voxelize: n=%lld out of range", (long long)n);
  M3D_CHECK_ARG(c >= 1 && b >= 1, "m3d_voxelize: c=%d b=%d", c, b);
  M3D_CHECK_ARG(d_coord && d_feat && d_offset && d_voxel_coords && d_voxel_feats && d_v2p_map && d_meta,
                "m3d_voxelize: null pointer");
  M3D_CHECK_ARG(grid_size > 0.f, "m3d_voxelize: grid_size must be positive");
  M3D_CHECK_ARG(hash_method == M3D_HASH_FNV || hash_method == M3D_HASH_RAVEL,
                "m3d_voxelize: unknown hash method %d", hash_method);
  M3D_CHECK_ARG(grid_mode >= 0 && grid_mode <= 2, "m3d_voxelize: unknown grid mode %d", grid_mode);
  return voxelize_impl("m3d_voxelize", d_coord, d_feat, d_offset, n, c, b, grid_size, (double)grid_size, pad,
                       hash_method, grid_mode, d_voxel_coords, d_voxel_feats, d_v2p_map, d_keys, d_grid_coord,
                       nullptr, d_meta, d_ws, ws_bytes, stream);
}

extern "C" size_t m3d_grid_sample_workspace_bytes(int64_t n) { return vox_layout(n, 1, nullptr, 0, nullptr); }

extern "C" int m3d_grid_sample(const float* d_coord, int64_t n, double grid_size, int hash_type,
                               int32_t* d_grid_coord, int64_t* d_inverse, int32_t* d_voxel_coords,
                               int32_t* d_count, uint64_t* d_keys, int64_t* d_meta, void* d_ws,
                               size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n > 0 && n < (int64_t)INT_MAX, "m3d_grid_sample: n=%lld out of range", (long long)n);
  M3D_CHECK_ARG(d_coord && d_grid_coord && d_inverse && d_voxel_coords && d_count && d_meta,
                "m3d_grid_sample: null pointer");
  M3D_CHECK_ARG(grid_size > 0.0, "m3d_grid_sample: grid_size must be positive");
  M3D_CHECK_ARG(hash_type == M3D_HASH_FNV || hash_type == M3D_HASH_RAVEL, "m3d_grid_sample: unknown hash type %d",
                hash_type);
  // one cloud: batch index 0 everywhere (the offset array is never read when b == 1)
  return voxelize_impl("m3d_grid_sample", d_coord, nullptr, nullptr, n, 0, 1, (float)grid_size, grid_size, 0,
                       hash_type == M3D_HASH_FNV ? kHashFnv3 : M3D_HASH_RAVEL, kGridFloor64, d_voxel_coords, nullptr,
                       d_inverse, d_keys, d_grid_coord, d_count, d_meta, d_ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------
// pooling_method = "random" (random_pooling, mosaic3d/src/models/utils/structure.py:69-110): the voxel set, its order
// and v2p_map are those of mean pooling; the feature of a voxel is the feature of ONE of its points, drawn uniformly.
// The reference draws with numpy's global RNG in a Python loop; here every point gets a 32-bit hash of (seed, point
// index) and the point with the largest (hash, index) word of a voxel wins (one 64-bit atomic max per point).
namespace m3d {

__global__ __launch_bounds__(256) void pick_bid_kernel(const int64_t* __restrict__ v2p, int64_t n_pts, uint64_t seed,
                                                       unsigned long long* __restrict__ best) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  const uint64_t h = mix64(seed ^ ((uint64_t)p * 0x9E3779B97F4A7C15ull));
  atomicMax(&best[v2p[p]], (unsigned long long)((h & 0xFFFFFFFF00000000ull) | (uint64_t)p));
}

__global__ __launch_bounds__(256) void pick_gather_kernel(const unsigned long long* __restrict__ best,
                                                          const float* __restrict__ feat, int c, int64_t n_vox,
                                                          int64_t* __restrict__ choice, float* __restrict__ voxel_feats) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_vox * c) return;
  const int64_t v = e / c;
  const int ch = (int)(e % c);
  const int64_t p = (int64_t)(best[v] & 0xFFFFFFFFull);
  if (ch == 0 && choice) choice[v] = p;
  if (voxel_feats) voxel_feats[e] = feat[p * c + ch];
}

}  // namespace m3d

extern "C" size_t m3d_voxel_pick_random_workspace_bytes(int64_t n_vox) {
  return align_up((size_t)(n_vox > 0 ? n_vox : 1) * sizeof(unsigned long long));
}

extern "C" int m3d_voxel_pick_random(const int64_t* d_v2p_map, const float* d_feat, int64_t n_pts, int c,
                                     int64_t n_vox, uint64_t seed, int64_t* d_choice, float* d_voxel_feats,
                                     void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const char* who = "m3d_voxel_pick_random";
  M3D_CHECK_ARG(d_v2p_map && (d_choice || d_voxel_feats), "%s: null pointer", who);
  M3D_CHECK_ARG(d_voxel_feats == nullptr || (d_feat != nullptr && c >= 1), "%s: features requested without an input", who);
  M3D_CHECK_ARG(n_pts >= 1 && n_pts < ((int64_t)1 << 32) && n_vox >= 1 && n_vox <= n_pts, "%s: bad sizes", who);
  const size_t need = (size_t)n_vox * sizeof(unsigned long long);
  if (d_ws == nullptr || ws_bytes < need) {
    set_error("%s: workspace too small (%zu given, %zu needed)", who, ws_bytes, need);
    return M3D_ENOSPC;
  }
  unsigned long long* best = static_cast<unsigned long long*>(d_ws);
  M3D_HIP(hipMemsetAsync(best, 0, need, stream));
  hipLaunchKernelGGL(pick_bid_kernel, dim3((unsigned)cdiv(n_pts, 256)), dim3(256), 0, stream, d_v2p_map, n_pts, seed, best);
  const int cc = d_voxel_feats ? c : 1;
  hipLaunchKernelGGL(pick_gather_kernel, dim3((unsigned)cdiv(n_vox * cc, 256)), dim3(256), 0, stream, best, d_feat, cc,
                     n_vox, d_choice, d_voxel_feats);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}
