// Kernel-map ("rulebook") construction for submanifold (k=3) and strided (k=2, s=2) sparse
// convolutions, plus the tile schedule the convolution kernel walks.
//
// Coordinates are linearised to a 64-bit row-major index over (batch, spatial_shape) -- never a
// 32-bit offset -- and looked up in an open-addressing hash table in HBM (load factor <= 0.5,
// linear probing, 64-bit CAS insert).  The table for a 250 k-voxel level is 6 MB and stays in L2.
#include "common.h"

#include <limits.h>

namespace m3d {


struct Shape3 {
  int64_t s[3];
};

__device__ inline uint64_t lin_key(int b, int x, int y, int z, const Shape3& sh) {
  return (((uint64_t)b * (uint64_t)sh.s[0] + (uint64_t)x) * (uint64_t)sh.s[1] + (uint64_t)y) *
             (uint64_t)sh.s[2] + (uint64_t)z;
}

// ------------------------------------------------------------------------------ SubM k3
__global__ void subm_insert_kernel(const int4* __restrict__ idx, int64_t n, Shape3 sh,
                                   uint64_t* __restrict__ tkeys, int32_t* __restrict__ tvals,
                                   uint64_t mask) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int4 c = idx[j];
  const uint64_t key = lin_key(c.x, c.y, c.z, c.w, sh);
  uint64_t s = mix64(key) & mask;
  while (true) {
    const uint64_t old = atomicCAS((unsigned long long*)&tkeys[s], (unsigned long long)kEmptyKey,
                                   (unsigned long long)key);
    if (old == kEmptyKey) {
      tvals[s] = (int32_t)j;
      return;
    }
    if (old == key) return;  // duplicate row: the first inserter keeps the slot
    s = (s + 1) & mask;
  }
}

// Offsets k and 26 - k are opposite: if voxel r sits at offset k of voxel j, then j sits at offset 26 - k of r.
// Only k = 0 .. 12 are probed in the hash table; a hit also fills the mirrored entry (each entry has exactly one
// writer), the misses of the upper half are the -1 the caller pre-filled it with.  k = 13 is the voxel itself.
__global__ __launch_bounds__(256) void subm_probe_kernel(const int4* __restrict__ idx, int64_t n,
                                                         Shape3 sh, HashTable tab,
                                                         int32_t* __restrict__ pair) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;  // 0 .. 13
  if (j >= n) return;
  if (k == 13) {
    pair[(int64_t)13 * n + j] = (int32_t)j;
    return;
  }
  const int4 c = idx[j];
  const int x = c.y + (k / 9) - 1, y = c.z + ((k / 3) % 3) - 1, z = c.w + (k % 3) - 1;
  int32_t r = -1;
  if (x >= 0 && y >= 0 && z >= 0 && x < sh.s[0] && y < sh.s[1] && z < sh.s[2])
    r = table_find(tab, lin_key(c.x, x, y, z, sh));
  pair[(int64_t)k * n + j] = r;
  if (r >= 0) pair[(int64_t)(26 - k) * n + r] = (int32_t)j;
}

// --- occupancy-bitmap variant of the submanifold map (grids of at most kBitmapMaxCells cells): the voxels set
// their bit in a bitmap over the linearised (batch, shape) grid; the rank of a set bit (per-word prefix count +
// popcount below the bit) numbers the voxels in ascending linear order, and `rowof` maps that number back to the
// row (rows of level 0 are in hash-key order, rows below it already are in linear order).  A neighbour probe is
// then one bitmap word, and for a hit one prefix word and one rowof entry - no keys to compare, no probe chains,
// no table to fill.  The three bitmap kernels are shared with the strided maps (defined below).
__global__ __launch_bounds__(256) void subm_mark_kernel(const int4* __restrict__ idx, int64_t n, Shape3 sh,
                                                        int batch_size, unsigned long long* __restrict__ bitmap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  long long word = -1;  // no early return: every lane takes part in the shuffles
  unsigned long long bits = 0;
  if (i < n) {
    const int4 c = idx[i];
    if (c.x >= 0 && c.x < batch_size && c.y >= 0 && c.z >= 0 && c.w >= 0 && c.y < sh.s[0] && c.z < sh.s[1] &&
        c.w < sh.s[2]) {
      const uint64_t key = lin_key(c.x, c.y, c.z, c.w, sh);
      word = (long long)(key >> 6);
      bits = 1ull << (key & 63);
    }
  }
  // rows in linear order put runs of equal words into neighbouring lanes: OR them together before the atomic
  const long long prev = __shfl_up(word, 1, 64);
  const bool head = lane == 0 || prev != word;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long w = __shfl_down(word, d, 64);
    const unsigned long long b = __shfl_down(bits, d, 64);
    if (lane + d < 64 && w == word) bits |= b;
  }
  if (head && word >= 0) atomicOr(&bitmap[word], bits);
}

__device__ inline int bitmap_rank(const unsigned long long* __restrict__ bitmap, const int* __restrict__ prefix,
                                  uint64_t key) {
  const unsigned long long w = bitmap[key >> 6];
  if (!((w >> (key & 63)) & 1ull)) return -1;
  return prefix[key >> 6] + __popcll(w & ((1ull << (key & 63)) - 1));
}

__global__ __launch_bounds__(256) void subm_rowof_kernel(const int4* __restrict__ idx, int64_t n, Shape3 sh,
                                                         int batch_size,
                                                         const unsigned long long* __restrict__ bitmap,
                                                         const int* __restrict__ prefix, int32_t* __restrict__ rowof) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int4 c = idx[j];
  if (c.x < 0 || c.x >= batch_size || c.y < 0 || c.z < 0 || c.w < 0 || c.y >= sh.s[0] || c.z >= sh.s[1] ||
      c.w >= sh.s[2])
    return;
  rowof[bitmap_rank(bitmap, prefix, lin_key(c.x, c.y, c.z, c.w, sh))] = (int32_t)j;
}

// grid.y = kernel offset 0 .. 13, mirrored entries as in subm_probe_kernel
__global__ __launch_bounds__(256) void subm_probe_bitmap_kernel(const int4* __restrict__ idx, int64_t n, Shape3 sh,
                                                                int batch_size,
                                                                const unsigned long long* __restrict__ bitmap,
                                                                const int* __restrict__ prefix,
                                                                const int32_t* __restrict__ rowof,
                                                                int32_t* __restrict__ pair) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= n) return;
  if (k == 13) {
    pair[(int64_t)13 * n + j] = (int32_t)j;
    return;
  }
  const int4 c = idx[j];
  const int x = c.y + (k / 9) - 1, y = c.z + ((k / 3) % 3) - 1, z = c.w + (k % 3) - 1;
  int32_t r = -1;
  if (c.x >= 0 && c.x < batch_size && x >= 0 && y >= 0 && z >= 0 && x < sh.s[0] && y < sh.s[1] && z < sh.s[2]) {
    const int rank = bitmap_rank(bitmap, prefix, lin_key(c.x, x, y, z, sh));
    if (rank >= 0) r = rowof[rank];
  }
  pair[(int64_t)k * n + j] = r;
  if (r >= 0) pair[(int64_t)(26 - k) * n + r] = (int32_t)j;
}

// ------------------------------------------------------------------------------ strided k2 s2
// Rows whose parent falls outside the output shape get `drop_key` = one past the largest valid key.
__global__ void strided_key_kernel(const int4* __restrict__ idx, int64_t n, const int64_t* __restrict__ d_n,
                                   Shape3 out_sh, uint64_t drop_key, uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (d_n && i >= *d_n) {  // capacity rows past the device-side row count
    keys[i] = drop_key;
    return;
  }
  const int4 c = idx[i];
  const int x = c.y >> 1, y = c.z >> 1, z = c.w >> 1;
  const bool ok = x < out_sh.s[0] && y < out_sh.s[1] && z < out_sh.s[2];
  const uint64_t key = ok ? lin_key(c.x, x, y, z, out_sh) : drop_key;
  keys[i] = key < drop_key ? key : drop_key;
}

__global__ void strided_count_kernel(const uint64_t* __restrict__ ukeys, const int* __restrict__ d_count,
                                     uint64_t drop_key, int* __restrict__ d_nout,
                                     int64_t* __restrict__ n_out64) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int m = *d_count;
  if (m > 0 && ukeys[m - 1] == drop_key) --m;
  *d_nout = m;
  *n_out64 = m;
}

__global__ void strided_pair_kernel(const int4* __restrict__ idx, const int32_t* __restrict__ inverse,
                                    int64_t n, const int* __restrict__ d_nout,
                                    int32_t* __restrict__ pair_fwd, int32_t* __restrict__ pair_bwd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = inverse[i];
  if (o >= *d_nout) return;
  const int4 c = idx[i];
  const int k = ((c.y & 1) * 2 + (c.z & 1)) * 2 + (c.w & 1);
  pair_fwd[(int64_t)k * n + o] = (int32_t)i;
  pair_bwd[(int64_t)k * n + i] = o;
}

__global__ void strided_outidx_kernel(const uint64_t* __restrict__ ukeys, const int* __restrict__ d_nout,
                                      Shape3 out_sh, int4* __restrict__ out_idx) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= *d_nout) return;
  uint64_t r = ukeys[o];
  const int z = (int)(r % (uint64_t)out_sh.s[2]); r /= (uint64_t)out_sh.s[2];
  const int y = (int)(r % (uint64_t)out_sh.s[1]); r /= (uint64_t)out_sh.s[1];
  const int x = (int)(r % (uint64_t)out_sh.s[0]); r /= (uint64_t)out_sh.s[0];
  out_idx[o] = make_int4((int)r, x, y, z);
}

// --- occupancy-bitmap variant: when batch * output volume is small enough, the set of output cells is a
// bitmap over the linearised output grid; the rank of a set bit (prefix popcount) IS the output row,
// because output rows are ordered by ascending linear index.  No hashing, no sorting.
constexpr int kBitmapBlockWords = 1024;  // words per 256-thread block (4 per thread)

// Below level 0 the voxels come in ascending linear order, so the lanes of a wave hit runs of the same bitmap word,
// and atomics on one address are served one at a time: equal words in neighbouring lanes are OR-ed together by a
// segmented shuffle reduction first and only the first lane of each run touches memory.
// The same pass fills column i of both pair arrays with -1 (the pair kernel then writes the entries that exist):
// there are at most as many output rows as input rows, so the live columns are covered without a memset of the
// whole capacity.
__global__ __launch_bounds__(256) void strided_mark_kernel(const int4* __restrict__ idx, int64_t n,
                                                           const int64_t* __restrict__ d_n, Shape3 out_sh,
                                                           uint64_t cells, unsigned long long* __restrict__ bitmap,
                                                           int32_t* __restrict__ pair_fwd,
                                                           int32_t* __restrict__ pair_bwd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  // no early return: every lane takes part in the shuffles (an inactive lane carries word -1, bits 0)
  long long word = -1;
  unsigned long long bits = 0;
  if (i < n && !(d_n && i >= *d_n)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      pair_fwd[(int64_t)k * n + i] = -1;
      pair_bwd[(int64_t)k * n + i] = -1;
    }
    const int4 c = idx[i];
    const int x = c.y >> 1, y = c.z >> 1, z = c.w >> 1;
    if (x < out_sh.s[0] && y < out_sh.s[1] && z < out_sh.s[2]) {
      const uint64_t key = lin_key(c.x, x, y, z, out_sh);
      if (key < cells) {  // batch index or coordinate out of range: never touch memory past the bitmap
        word = (long long)(key >> 6);
        bits = 1ull << (key & 63);
      }
    }
  }
  const long long prev = __shfl_up(word, 1, 64);
  const bool head = lane == 0 || prev != word;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long w = __shfl_down(word, d, 64);
    const unsigned long long b = __shfl_down(bits, d, 64);
    if (lane + d < 64 && w == word) bits |= b;
  }
  // up to 8 children share a cell and 64 cells a word: most runs find their bits already set and skip the atomic
  if (head && word >= 0 && (__builtin_nontemporal_load(&bitmap[word]) & bits) != bits) atomicOr(&bitmap[word], bits);
}

// exclusive scan of one int per thread over a 256-thread block; returns the prefix, *total = block sum
__device__ inline int block_exclusive_scan_256(int v, int* total) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  __syncthreads();  // wsum may still be read by a previous call
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + inc - v;
}

__global__ __launch_bounds__(256) void bitmap_block_sum_kernel(const unsigned long long* __restrict__ bitmap,
                                                               int64_t nwords, int* __restrict__ bsum) {
  const int64_t w0 = (int64_t)blockIdx.x * kBitmapBlockWords + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (w0 + j < nwords) c += __popcll(bitmap[w0 + j]);
  int total;
  block_exclusive_scan_256(c, &total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one block: exclusive scan of the block sums in place, total -> *d_nout / *n_out64
__global__ __launch_bounds__(256) void bitmap_scan_sums_kernel(int* __restrict__ bsum, int nb, int* __restrict__ d_nout,
                                                               int64_t* __restrict__ n_out64) {
  int carry = 0;
  for (int base = 0; base < nb; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int total;
    const int ex = block_exclusive_scan_256(v, &total);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) {
    *d_nout = carry;
    *n_out64 = carry;
  }
}

// per-word exclusive prefix of the set-bit counts
__global__ __launch_bounds__(256) void bitmap_prefix_kernel(const unsigned long long* __restrict__ bitmap,
                                                            int64_t nwords, const int* __restrict__ bsum,
                                                            int* __restrict__ prefix) {
  const int64_t w0 = (int64_t)blockIdx.x * kBitmapBlockWords + threadIdx.x * 4;
  int cnt[4], c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    cnt[j] = w0 + j < nwords ? __popcll(bitmap[w0 + j]) : 0;
    c += cnt[j];
  }
  int total;
  int o = bsum[blockIdx.x] + block_exclusive_scan_256(c, &total);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (w0 + j < nwords) prefix[w0 + j] = o;
    o += cnt[j];
  }
}

__global__ void strided_pair_bitmap_kernel(const int4* __restrict__ idx, int64_t n, const int64_t* __restrict__ d_n,
                                           Shape3 out_sh, uint64_t cells,
                                           const unsigned long long* __restrict__ bitmap,
                                           const int* __restrict__ prefix, int32_t* __restrict__ pair_fwd,
                                           int32_t* __restrict__ pair_bwd, int4* __restrict__ out_idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (d_n && i >= *d_n)) return;
  const int4 c = idx[i];
  const int x = c.y >> 1, y = c.z >> 1, z = c.w >> 1;
  if (x >= out_sh.s[0] || y >= out_sh.s[1] || z >= out_sh.s[2]) return;
  const uint64_t key = lin_key(c.x, x, y, z, out_sh);
  if (key >= cells) return;
  const unsigned long long w = bitmap[key >> 6];
  const int o = prefix[key >> 6] + __popcll(w & ((1ull << (key & 63)) - 1));
  const int k = ((c.y & 1) * 2 + (c.z & 1)) * 2 + (c.w & 1);
  pair_fwd[(int64_t)k * n + o] = (int32_t)i;
  pair_bwd[(int64_t)k * n + i] = o;
  out_idx[o] = make_int4(c.x, x, y, z);  // every child of the cell writes the same value
}

// ------------------------------------------------------------------------------ schedule
__device__ inline uint64_t spread3(uint64_t v) {  // 16 bits -> every third bit
  v &= 0xFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// mode 1: key = batch << (3*mb) | Morton(x, y, z) with mb bits per axis
// mode 2: key = the row's first active kernel offset (rows are then grouped by offset; the sort is
//         stable, so rows keep the order of `base` inside a group)
__global__ void sched_key_kernel(const int32_t* __restrict__ pair, int kvol, int64_t ld, int64_t n,
                                 const int4* __restrict__ idx, int mode, int mb,
                                 const int32_t* __restrict__ base, uint64_t* __restrict__ keys,
                                 uint32_t* __restrict__ vals) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int32_t row = base ? base[t] : (int32_t)t;
  uint64_t key;
  if (mode == 1) {
    const int4 c = idx[row];
    const uint64_t m = (spread3((uint64_t)c.y) << 2) | (spread3((uint64_t)c.z) << 1) | spread3((uint64_t)c.w);
    key = ((uint64_t)c.x << (3 * mb)) | (m & ((1ull << (3 * mb)) - 1));
  } else {
    int kk = kvol;  // rows without any pair go last
    for (int k = 0; k < kvol; ++k)
      if (pair[(int64_t)k * ld + row] >= 0) { kk = k; break; }
    key = (uint64_t)kk;
  }
  keys[t] = key;
  vals[t] = (uint32_t)row;
}

// mode 3: key = batch | Morton of the coarse cell (x, y, z) >> cs | neighbour pattern.  Inside a cell of
// 2^cs voxels per side rows are regrouped by their neighbour pattern, so that the rows of one tile tend
// to miss the same kernel offsets and the tile can skip them.  The pattern is compared with the rarest
// offsets first (corners, then edges of the 3x3x3 stencil); faces and centre are nearly always present.
// The sort is stable: rows keep the base order inside a (cell, pattern) group.
__global__ void sched_pattern_key_kernel(const int32_t* __restrict__ pair, int64_t ld, int64_t n,
                                         const int4* __restrict__ idx, int cs, int cb,
                                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  uint32_t corner = 0, edge = 0;  // 8 + 12 bits
  for (int k = 0; k < 27; ++k) {
    const int nz = (k / 9 != 1) + ((k / 3) % 3 != 1) + (k % 3 != 1);
    if (nz < 2) continue;
    const uint32_t bit = pair[(int64_t)k * ld + row] >= 0 ? 1u : 0u;
    if (nz == 3) corner = (corner << 1) | bit;
    else edge = (edge << 1) | bit;
  }
  const int4 c = idx[row];
  const uint64_t m = (spread3((uint64_t)(c.y >> cs)) << 2) | (spread3((uint64_t)(c.z >> cs)) << 1) |
                     spread3((uint64_t)(c.w >> cs));
  const uint64_t cell = ((uint64_t)c.x << (3 * cb)) | (m & ((1ull << (3 * cb)) - 1));
  keys[row] = (cell << 20) | (corner << 12) | edge;
  vals[row] = (uint32_t)row;
}

// Schedule rows -> gathered kernel map, output rows and tile masks in one pass.  One wave per 64 schedule rows
// (= two 32-row tiles); bit k of a tile's mask says that some row of the tile has a neighbour at offset k.
__global__ __launch_bounds__(256) void sched_gather_mask_kernel(const int32_t* __restrict__ pair, int64_t ld,
                                                                int64_t n, int64_t npad, int kvol,
                                                                const uint32_t* __restrict__ order,
                                                                int32_t* __restrict__ nbr,
                                                                int32_t* __restrict__ out_row,
                                                                uint32_t* __restrict__ tile_mask) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64;
  if (base >= npad) return;
  const int64_t t = base + lane;
  const int32_t row = t < n ? (order ? (int32_t)order[t] : (int32_t)t) : -1;
  out_row[t] = row;
  uint32_t lo = 0, hi = 0;
  for (int k0 = 0; k0 < kvol; k0 += 9) {  // nine independent gathers in flight, then stores and votes
    int32_t v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = (row >= 0 && k0 + i < kvol) ? pair[(int64_t)(k0 + i) * ld + row] : -1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (k0 + i < kvol) {
        nbr[(int64_t)(k0 + i) * npad + t] = v[i];
        const uint64_t bal = __ballot(v[i] >= 0);
        if (bal & 0xFFFFFFFFull) lo |= 1u << (k0 + i);
        if (bal >> 32) hi |= 1u << (k0 + i);
      }
    }
  }
  if (lane == 0) {
    tile_mask[base / 32] = lo;
    tile_mask[base / 32 + 1] = hi;
  }
}

// Launch order of the 128-row tiles: three classes by the number of active kernel offsets (the thirds of this
// schedule's tiles, heaviest third first), schedule order inside a class - a stable counting sort, one workgroup.
// Workgroups are dispatched in index order, so the tiles with the longest reductions start first and the light
// ones fill the tail of the launch; a few coarse classes keep long runs of spatially adjacent tiles together and
// measured as fast as a full sort by offset count.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ mask, int nt,
                                                          int32_t* __restrict__ order) {
  __shared__ int hist[28], cls[28], base[3], wcnt[16][3];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  auto work_of = [&](int t) {
    const uint4 m = *reinterpret_cast<const uint4*>(mask + 4 * (int64_t)t);
    return __popc(m.x | m.y | m.z | m.w);
  };
  if (tid < 28) hist[tid] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += 1024) atomicAdd(&hist[work_of(t)], 1);
  __syncthreads();
  if (tid == 0) {
    // offset counts from 27 down: class 0 until a third of the tiles is covered, class 1 until two thirds, then 2
    int seen = 0, n0 = 0, n1 = 0;
    for (int w = 27; w >= 0; --w) {
      const int c = 3 * seen < nt ? 0 : (3 * seen < 2 * nt ? 1 : 2);
      cls[w] = c;
      seen += hist[w];
      if (c == 0) n0 += hist[w];
      if (c == 1) n1 += hist[w];
    }
    base[0] = 0;
    base[1] = n0;
    base[2] = n0 + n1;
  }
  __syncthreads();
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int c0 = 0; c0 < nt; c0 += 1024) {
    const int t = c0 + tid;
    const int c = t < nt ? cls[work_of(t)] : -1;
    int rank = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const uint64_t bal = __ballot(c == b);
      if (c == b) rank = __popcll(bal & lt);
      if (lane == 0) wcnt[wave][b] = __popcll(bal);
    }
    __syncthreads();
    if (c >= 0) {
      int off = 0;
      for (int v = 0; v < wave; ++v) off += wcnt[v][c];
      order[base[c] + off + rank] = t;
    }
    __syncthreads();
    if (tid < 3) {
      int sum = 0;
      for (int v = 0; v < 16; ++v) sum += wcnt[v][tid];
      base[tid] += sum;
    }
    __syncthreads();
  }
}

static inline uint64_t table_cap(int64_t n) { return next_pow2((uint64_t)(n > 0 ? n : 1) * 2); }

}  // namespace m3d

using namespace m3d;

static int check_shape(const int64_t s[3], const char* who) {
  for (int a = 0; a < 3; ++a)
    if (s[a] <= 0 || s[a] > (int64_t)INT_MAX) {
      set_error("%s: spatial_shape[%d]=%lld out of range", who, a, (long long)s[a]);
      return M3D_EINVAL;
    }
  return M3D_OK;
}

// Largest linearised grid (cells) the occupancy-bitmap variants are used for: 32 MiB of bitmap.
static const uint64_t kBitmapMaxCells = 1ull << 28;

// batch_size * volume when a bitmap over the whole grid is affordable (batch_size >= 1 known), else 0
static uint64_t subm_bitmap_cells(const int64_t spatial_shape[3], int batch_size) {
  if (!spatial_shape || batch_size < 1) return 0;
  long double vol = (long double)batch_size;
  for (int a = 0; a < 3; ++a) {
    if (spatial_shape[a] <= 0) return 0;
    vol *= (long double)spatial_shape[a];
  }
  if (!(vol <= (long double)kBitmapMaxCells)) return 0;
  return (uint64_t)batch_size * (uint64_t)spatial_shape[0] * (uint64_t)spatial_shape[1] * (uint64_t)spatial_shape[2];
}

extern "C" size_t m3d_subm_rulebook_workspace_bytes(int64_t n, const int64_t spatial_shape[3], int batch_size) {
  if (n <= 0) n = 1;
  const uint64_t cap = table_cap(n);
  size_t hash = align_up(cap * sizeof(uint64_t)) + align_up(cap * sizeof(int32_t));
  const uint64_t cells = subm_bitmap_cells(spatial_shape, batch_size);
  if (cells == 0) return hash;
  const uint64_t nwords = cdiv(cells, 64);
  size_t bm = align_up(nwords * sizeof(uint64_t)) + align_up(nwords * sizeof(int)) +
              align_up(cdiv(nwords, kBitmapBlockWords) * sizeof(int)) + align_up(64 * sizeof(int)) +
              align_up(64 * sizeof(int64_t)) + align_up((size_t)n * sizeof(int32_t));
  return bm;
}

extern "C" int m3d_subm_rulebook(const int32_t* d_indices, int64_t n, const int64_t spatial_shape[3], int batch_size,
                                 int32_t* d_pair, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n >= 0 && n < (int64_t)INT_MAX, "m3d_subm_rulebook: n out of range");
  if (n == 0) return M3D_OK;
  M3D_CHECK_ARG(d_indices && d_pair && spatial_shape, "m3d_subm_rulebook: null pointer");
  if (int rc = check_shape(spatial_shape, "m3d_subm_rulebook")) return rc;
  const uint64_t cells = subm_bitmap_cells(spatial_shape, batch_size);
  if (cells != 0) {
    const int64_t nwords = (int64_t)cdiv(cells, 64);
    const int nb = (int)cdiv(nwords, kBitmapBlockWords);
    Arena a(d_ws, ws_bytes);
    unsigned long long* bitmap = a.take<unsigned long long>(nwords);
    int* prefix = a.take<int>(nwords);
    int* bsum = a.take<int>(nb);
    int* nout = a.take<int>(64);
    int64_t* nout64 = a.take<int64_t>(64);
    int32_t* rowof = a.take<int32_t>(n);
    if (!a.ok) {
      set_error("m3d_subm_rulebook: workspace too small (%zu given, %zu needed)", ws_bytes, a.off);
      return M3D_ENOSPC;
    }
    Shape3 sh{{spatial_shape[0], spatial_shape[1], spatial_shape[2]}};
    const unsigned gb = (unsigned)cdiv(n, 256);
    const int4* idx = reinterpret_cast<const int4*>(d_indices);
    M3D_HIP(hipMemsetAsync(bitmap, 0, (size_t)nwords * sizeof(unsigned long long), stream));
    M3D_HIP(hipMemsetAsync(d_pair + (int64_t)14 * n, 0xFF, (size_t)13 * n * sizeof(int32_t), stream));
    hipLaunchKernelGGL(subm_mark_kernel, dim3(gb), dim3(256), 0, stream, idx, n, sh, batch_size, bitmap);
    hipLaunchKernelGGL(bitmap_block_sum_kernel, dim3(nb), dim3(256), 0, stream, bitmap, nwords, bsum);
    hipLaunchKernelGGL(bitmap_scan_sums_kernel, dim3(1), dim3(256), 0, stream, bsum, nb, nout, nout64);
    hipLaunchKernelGGL(bitmap_prefix_kernel, dim3(nb), dim3(256), 0, stream, bitmap, nwords, bsum, prefix);
    hipLaunchKernelGGL(subm_rowof_kernel, dim3(gb), dim3(256), 0, stream, idx, n, sh, batch_size, bitmap, prefix, rowof);
    hipLaunchKernelGGL(subm_probe_bitmap_kernel, dim3(gb, 14), dim3(256), 0, stream, idx, n, sh, batch_size, bitmap,
                       prefix, rowof, d_pair);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
  }
  const uint64_t cap = table_cap(n);
  Arena a(d_ws, ws_bytes);
  uint64_t* tkeys = a.take<uint64_t>(cap);
  int32_t* tvals = a.take<int32_t>(cap);
  if (!a.ok) {
    set_error("m3d_subm_rulebook: workspace too small (%zu given, %zu needed)", ws_bytes, a.off);
    return M3D_ENOSPC;
  }
  Shape3 sh{{spatial_shape[0], spatial_shape[1], spatial_shape[2]}};
  M3D_HIP(hipMemsetAsync(tkeys, 0xFF, cap * sizeof(uint64_t), stream));
  const int bs = 256;
  const unsigned gb = (unsigned)cdiv(n, bs);
  const int4* idx = reinterpret_cast<const int4*>(d_indices);
  hipLaunchKernelGGL(subm_insert_kernel, dim3(gb), dim3(bs), 0, stream, idx, n, sh, tkeys, tvals,
                     cap - 1);
  HashTable tab{tkeys, tvals, cap - 1};
  M3D_HIP(hipMemsetAsync(d_pair + (int64_t)14 * n, 0xFF, (size_t)13 * n * sizeof(int32_t), stream));
  hipLaunchKernelGGL(subm_probe_kernel, dim3(gb, 14), dim3(bs), 0, stream, idx, n, sh, tab, d_pair);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

static bool strided_cells(const int64_t spatial_shape[3], int batch_size, uint64_t* cells) {
  if (!spatial_shape || batch_size < 1) return false;
  long double vol = (long double)batch_size;
  for (int a = 0; a < 3; ++a) {
    if (spatial_shape[a] <= 0) return false;
    vol *= (long double)(spatial_shape[a] / 2);
  }
  if (!(vol < 9.0e18L)) return false;
  *cells = (uint64_t)batch_size * (uint64_t)(spatial_shape[0] / 2) * (uint64_t)(spatial_shape[1] / 2) *
           (uint64_t)(spatial_shape[2] / 2);
  return true;
}

extern "C" size_t m3d_strided_rulebook_workspace_bytes(int64_t n, const int64_t spatial_shape[3], int batch_size) {
  if (n <= 0) n = 1;
  uint64_t cells = 0;
  if (strided_cells(spatial_shape, batch_size, &cells) && cells > 0 && cells <= kBitmapMaxCells) {
    const uint64_t nwords = cdiv(cells, 64);
    return align_up(nwords * sizeof(uint64_t)) + align_up(nwords * sizeof(int)) +
           align_up(cdiv(nwords, kBitmapBlockWords) * sizeof(int)) + align_up(64 * sizeof(int));
  }
  return align_up(n * sizeof(uint64_t)) * 2 + align_up(n * sizeof(int32_t)) * 2 +
         align_up(64 * sizeof(int)) * 2 + align_up(unique_ws_bytes(n));
}

extern "C" int m3d_strided_rulebook(const int32_t* d_indices, int64_t n, const int64_t* d_n_rows,
                                    const int64_t spatial_shape[3], int batch_size, int32_t* d_out_indices, int32_t* d_pair_fwd, int32_t* d_pair_bwd,
                                    int64_t* d_n_out, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n > 0 && n < (int64_t)INT_MAX, "m3d_strided_rulebook: n out of range");
  M3D_CHECK_ARG(d_indices && d_out_indices && d_pair_fwd && d_pair_bwd && d_n_out && spatial_shape,
                "m3d_strided_rulebook: null pointer");
  if (int rc = check_shape(spatial_shape, "m3d_strided_rulebook")) return rc;
  Shape3 osh{{spatial_shape[0] / 2, spatial_shape[1] / 2, spatial_shape[2] / 2}};
  M3D_CHECK_ARG(osh.s[0] > 0 && osh.s[1] > 0 && osh.s[2] > 0,
                "m3d_strided_rulebook: spatial shape too small for a stride-2 convolution");
  M3D_CHECK_ARG(batch_size >= 1, "m3d_strided_rulebook: batch_size must be >= 1");
  const long double vol = (long double)batch_size * osh.s[0] * osh.s[1] * osh.s[2];
  M3D_CHECK_ARG(vol < 9.0e18L, "m3d_strided_rulebook: batch_size * output volume overflows 63 bits");
  const uint64_t drop_key = (uint64_t)batch_size * (uint64_t)osh.s[0] * (uint64_t)osh.s[1] * (uint64_t)osh.s[2];
  const int bs = 256;
  const unsigned gb = (unsigned)cdiv(n, bs);
  const int4* idx = reinterpret_cast<const int4*>(d_indices);
  if (drop_key <= kBitmapMaxCells) {
    const int64_t nwords = (int64_t)cdiv(drop_key, 64);
    const int nb = (int)cdiv(nwords, kBitmapBlockWords);
    Arena a(d_ws, ws_bytes);
    unsigned long long* bitmap = a.take<unsigned long long>(nwords);
    int* prefix = a.take<int>(nwords);
    int* bsum = a.take<int>(nb);
    int* nout = a.take<int>(64);
    if (!a.ok) {
      set_error("m3d_strided_rulebook: workspace too small (%zu given, %zu needed)", ws_bytes, a.off);
      return M3D_ENOSPC;
    }
    M3D_HIP(hipMemsetAsync(bitmap, 0, (size_t)nwords * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(strided_mark_kernel, dim3(gb), dim3(bs), 0, stream, idx, n, d_n_rows, osh, drop_key, bitmap,
                       d_pair_fwd, d_pair_bwd);
    hipLaunchKernelGGL(bitmap_block_sum_kernel, dim3(nb), dim3(256), 0, stream, bitmap, nwords, bsum);
    hipLaunchKernelGGL(bitmap_scan_sums_kernel, dim3(1), dim3(256), 0, stream, bsum, nb, nout, d_n_out);
    hipLaunchKernelGGL(bitmap_prefix_kernel, dim3(nb), dim3(256), 0, stream, bitmap, nwords, bsum, prefix);
    hipLaunchKernelGGL(strided_pair_bitmap_kernel, dim3(gb), dim3(bs), 0, stream, idx, n, d_n_rows, osh, drop_key, bitmap, prefix,
                       d_pair_fwd, d_pair_bwd, reinterpret_cast<int4*>(d_out_indices));
    M3D_LAUNCH_CHECK();
    return M3D_OK;
  }
  const int nbits = bits_for(drop_key);
  Arena a(d_ws, ws_bytes);
  uint64_t* keys = a.take<uint64_t>(n);
  uint64_t* ukeys = a.take<uint64_t>(n);
  int32_t* rep = a.take<int32_t>(n);
  int32_t* inverse = a.take<int32_t>(n);
  int* count = a.take<int>(64);
  int* nout = a.take<int>(64);
  const size_t ub = unique_ws_bytes(n);
  char* uws = a.take<char>(ub);
  if (!a.ok) {
    set_error("m3d_strided_rulebook: workspace too small (%zu given, %zu needed)", ws_bytes, a.off);
    return M3D_ENOSPC;
  }
  hipLaunchKernelGGL(strided_key_kernel, dim3(gb), dim3(bs), 0, stream, idx, n, d_n_rows, osh, drop_key, keys);
  M3D_LAUNCH_CHECK();
  int rc = unique_sorted(keys, n, nbits, ukeys, rep, inverse, count, uws, ub, stream);
  if (rc != M3D_OK) return rc;
  hipLaunchKernelGGL(strided_count_kernel, dim3(1), dim3(64), 0, stream, ukeys, count, drop_key, nout, d_n_out);
  M3D_HIP(hipMemsetAsync(d_pair_fwd, 0xFF, (size_t)8 * n * sizeof(int32_t), stream));
  M3D_HIP(hipMemsetAsync(d_pair_bwd, 0xFF, (size_t)8 * n * sizeof(int32_t), stream));
  hipLaunchKernelGGL(strided_pair_kernel, dim3(gb), dim3(bs), 0, stream, idx, inverse, n, nout,
                     d_pair_fwd, d_pair_bwd);
  hipLaunchKernelGGL(strided_outidx_kernel, dim3(gb), dim3(bs), 0, stream, ukeys, nout, osh,
                     reinterpret_cast<int4*>(d_out_indices));
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int64_t m3d_schedule_npad(int64_t n) { return cdiv(n > 0 ? n : 1, 256) * 256; }

extern "C" size_t m3d_schedule_workspace_bytes(int64_t n) {
  if (n <= 0) n = 1;
  return align_up(n * sizeof(uint64_t)) * 2 + align_up(n * sizeof(uint32_t)) * 2 +
         align_up(64 * sizeof(int)) + align_up(sort_pairs_ws_bytes(n));
}

extern "C" int m3d_build_schedule(const int32_t* d_pair, int k, int64_t ld, int64_t n,
                                  const int32_t* d_indices, const int64_t* spatial_shape,
                                  int batch_size, int mode, const int32_t* d_base_order,
                                  int32_t* d_nbr,
                                  int32_t* d_out_row, uint32_t* d_tile_mask, void* d_ws,
                                  size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n > 0 && n < (int64_t)INT_MAX, "m3d_build_schedule: n out of range");
  M3D_CHECK_ARG(k >= 1 && k <= 27 && ld >= n, "m3d_build_schedule: k=%d ld=%lld", k, (long long)ld);
  M3D_CHECK_ARG(d_pair && d_nbr && d_out_row && d_tile_mask, "m3d_build_schedule: null pointer");
  const int cell_shift = (mode >> 8) & 0xFF;  // mode 3 only: log2 of the cell side, 0 = default
  mode &= 0xFF;
  M3D_CHECK_ARG(mode >= 0 && mode <= 3, "m3d_build_schedule: unknown mode %d", mode);
  M3D_CHECK_ARG(mode != 3 || k == 27, "m3d_build_schedule: mode 3 needs a 3x3x3 kernel map (k=%d)", k);
  const int64_t npad = m3d_schedule_npad(n);
  const int bs = 256;
  const uint32_t* order = nullptr;
  if (mode != 0) {
    Arena a(d_ws, ws_bytes);
    uint64_t* keys = a.take<uint64_t>(n);
    uint64_t* kalt = a.take<uint64_t>(n);
    uint32_t* vals = a.take<uint32_t>(n);
    uint32_t* valt = a.take<uint32_t>(n);
    int* count = a.take<int>(64);
    const size_t sb = sort_pairs_ws_bytes(n);
    char* sws = a.take<char>(sb);
    if (!a.ok) {
      set_error("m3d_build_schedule: workspace too small (%zu given, %zu needed)", ws_bytes, a.off);
      return M3D_ENOSPC;
    }
    int mb = 16, nbits = 4;
    if (mode == 1 || mode == 3) {
      M3D_CHECK_ARG(d_indices != nullptr, "m3d_build_schedule: modes 1 and 3 need the output coordinates");
      if (spatial_shape) {
        int64_t mx = spatial_shape[0];
        if (spatial_shape[1] > mx) mx = spatial_shape[1];
        if (spatial_shape[2] > mx) mx = spatial_shape[2];
        mb = bits_for((uint64_t)(mx > 1 ? mx - 1 : 1));
        if (mb > 16) mb = 16;
      }
      nbits = 3 * mb + bits_for((uint64_t)(batch_size > 1 ? batch_size - 1 : 1));
    }
    if (mode == 3) {
      const int cs = cell_shift ? (cell_shift > 15 ? 15 : cell_shift) : 6;
      const int cb = mb > cs ? mb - cs : 0;  // bits per axis of the cell coordinate
      nbits = 20 + 3 * cb + bits_for((uint64_t)(batch_size > 1 ? batch_size - 1 : 1));
      M3D_CHECK_ARG(d_base_order == nullptr, "m3d_build_schedule: mode 3 takes no base order");
      hipLaunchKernelGGL(sched_pattern_key_kernel, dim3((unsigned)cdiv(n, bs)), dim3(bs), 0, stream, d_pair, ld, n,
                         reinterpret_cast<const int4*>(d_indices), cs, cb, keys, vals);
    } else {
      hipLaunchKernelGGL(sched_key_kernel, dim3((unsigned)cdiv(n, bs)), dim3(bs), 0, stream, d_pair, k, ld,
                         n, reinterpret_cast<const int4*>(d_indices), mode, mb, d_base_order, keys, vals);
    }
    M3D_LAUNCH_CHECK();
    bool in_alt = false;
    (void)count;  // every row takes part: the count is the host's n
    int rc = sort_pairs(keys, vals, kalt, valt, nullptr, n, nbits, sws, sb, stream, &in_alt);
    if (rc != M3D_OK) return rc;
    order = in_alt ? valt : vals;
  }
  if (mode == 0 && d_base_order) order = reinterpret_cast<const uint32_t*>(d_base_order);
  hipLaunchKernelGGL(sched_gather_mask_kernel, dim3((unsigned)cdiv(npad, 256)), dim3(256), 0, stream, d_pair, ld, n,
                     npad, k, order, d_nbr, d_out_row, d_tile_mask);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_schedule_tile_order(const uint32_t* d_tile_mask, int64_t npad, int32_t* d_tile_order, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(d_tile_mask && d_tile_order, "m3d_schedule_tile_order: null pointer");
  M3D_CHECK_ARG(npad > 0 && npad % 128 == 0 && npad / 128 < (int64_t)INT_MAX, "m3d_schedule_tile_order: npad must be a positive multiple of 128");
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, d_tile_mask, (int)(npad / 128), d_tile_order);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}
