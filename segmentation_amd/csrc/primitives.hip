// Device primitives shared by the voxelizer and the kernel-map builders:
//   * sort_pairs     stable LSD radix sort of (u64 key, u32 value), element count on the device
//   * unique_sorted  np.unique(keys, return_inverse=True) + a representative index per key,
//                    built on an open-addressing hash table in HBM + the radix sort
// Everything runs on the caller's stream; no host synchronisation, no allocation.
#include "common.h"

#include <string.h>

namespace m3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* get_error() { return g_err; }

// ------------------------------------------------------------------------------ radix sort
constexpr int kWaveItems = 512;    // elements ranked by one wave per pass
constexpr int kSortBlock = 256;    // 4 waves per workgroup
constexpr int kRounds = kWaveItems / 64;

__global__ __launch_bounds__(kSortBlock) void radix_hist_kernel(
    const uint64_t* __restrict__ keys, const int* __restrict__ d_count, int host_count, int shift,
    uint32_t* __restrict__ counts, int nchunks) {
  __shared__ uint32_t h[kSortBlock / 64][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * (kSortBlock / 64) + wave;
  for (int i = lane; i < 256; i += 64) h[wave][i] = 0;
  __syncthreads();
  if (chunk < nchunks) {
    const int64_t m = d_count ? *d_count : host_count;
    const int64_t base = (int64_t)chunk * kWaveItems;
    uint64_t kr[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int64_t i = base + r * 64 + lane;
      kr[r] = i < m ? keys[i] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
      if (base + r * 64 + lane < m) atomicAdd(&h[wave][(kr[r] >> shift) & 255], 1u);
  }
  __syncthreads();
  if (chunk < nchunks)
    for (int i = lane; i < 256; i += 64) counts[(int64_t)i * nchunks + chunk] = h[wave][i];
}

// One workgroup per digit: exclusive prefix over that digit's per-chunk counts (in place) and the
// digit's total.  The 256 digit totals are turned into digit bases inside the scatter kernel.
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ totals, int nchunks) {
  __shared__ uint32_t wsum[4];
  const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t* row = counts + (int64_t)d * nchunks;
  const int per = (nchunks + 255) / 256;
  const int lo = t * per;
  const int hi = lo + per < nchunks ? lo + per : nchunks;
  uint32_t s = 0;
  for (int i = lo; i < hi; ++i) s += row[i];
  // block-wide inclusive scan of the 256 partial sums: shuffles inside a wave, one LDS round trip between waves
  uint32_t inc = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(inc, off, 64);
    if (lane >= off) inc += v;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wave; ++w) before += wsum[w];
  uint32_t run = before + inc - s;
  for (int i = lo; i < hi; ++i) {
    const uint32_t v = row[i];
    row[i] = run;
    run += v;
  }
  if (t == 255) totals[d] = before + inc;
}

__global__ __launch_bounds__(kSortBlock) void radix_scatter_kernel(
    const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
    uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, const int* __restrict__ d_count, int host_count,
    int shift, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ totals,
    int nchunks) {
  __shared__ uint32_t run_s[kSortBlock / 64][256];
  __shared__ uint32_t base_s[256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * (kSortBlock / 64) + wave;
  // this wave's per-digit offsets are requested first; they land while the digit totals are scanned
  uint32_t offs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) offs[q] = chunk < nchunks ? offsets[(int64_t)(q * 64 + lane) * nchunks + chunk] : 0u;
  const int64_t m = chunk < nchunks ? (int64_t)(d_count ? *d_count : host_count) : 0;
  const int64_t base = (int64_t)chunk * kWaveItems;
  // all of the wave's elements are requested up front: the rounds below are serialised by the running offsets
  // in LDS, and a load issued inside a round would add its full latency to every one of them
  uint64_t keys_r[kRounds];
  uint32_t vals_r[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = base + r * 64 + lane;
    keys_r[r] = i < m ? kin[i] : 0ull;
    vals_r[r] = i < m ? vin[i] : 0u;
  }
  {  // exclusive scan of the 256 digit totals (kSortBlock == 256 threads, one digit each): shuffles inside a wave,
     // one LDS round trip between the four waves
    __shared__ uint32_t wsum[4];
    const uint32_t own = totals[threadIdx.x];
    uint32_t inc = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    base_s[threadIdx.x] = before + inc - own;
    __syncthreads();
  }
  if (chunk < nchunks) {
#pragma unroll
    for (int q = 0; q < 4; ++q) run_s[wave][q * 64 + lane] = base_s[q * 64 + lane] + offs[q];
  }
  __syncthreads();
  if (chunk >= nchunks) return;
  volatile uint32_t* run = run_s[wave];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool valid = i < m;
    const uint64_t key = keys_r[r];
    const uint32_t val = vals_r[r];
    const unsigned d = (unsigned)(key >> shift) & 255u;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const int rank = __popcll(peers & lt);
    const int cnt = __popcll(peers);
    const uint32_t off = run[d];
    if (valid) {
      kout[off + rank] = key;
      vout[off + rank] = val;
      if (rank == 0) run[d] = off + cnt;
    }
  }
}

// Whole sort in one workgroup for short arrays: every pass of the three-kernel path costs three dependent
// launches whatever the length, which is all of its time below a few thousand elements (3 500 elements, four
// passes: 46 -> 33 us; at 15 000 one workgroup is already slower than the three kernels).  Sixteen waves each own
// a contiguous run of the array (so digit order = wave order = input order: stable), count their digits into LDS,
// one scan turns the counts into per-wave bases, and each wave ranks and scatters its run as the scatter kernel
// above does.  The passes ping-pong between the two global buffer pairs; the waves of a workgroup share the CU's
// write-through vector cache, so a workgroup barrier orders a pass's stores before the next pass's loads.
constexpr int kSmallSortWaves = 16;
constexpr int kSmallSortRounds = 4;
constexpr int kSmallSortMax = kSmallSortWaves * kSmallSortRounds * 64;  // 4096: longer arrays are faster on many CUs

__global__ __launch_bounds__(kSmallSortWaves * 64) void radix_sort_small_kernel(
    uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, const int* __restrict__ d_count, int host_count, int passes) {
  __shared__ uint32_t cnt_s[kSmallSortWaves][256];
  __shared__ uint32_t base_s[256];
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = d_count ? *d_count : host_count;
  const int rounds = (m + kSmallSortWaves * 64 - 1) / (kSmallSortWaves * 64);  // per wave, <= kSmallSortRounds
  const int first = wave * rounds * 64;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint64_t* ki = k0;
  uint32_t* vi = v0;
  uint64_t* ko = k1;
  uint32_t* vo = v1;
  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    uint64_t keys_r[kSmallSortRounds];
    uint32_t vals_r[kSmallSortRounds];
#pragma unroll
    for (int r = 0; r < kSmallSortRounds; ++r) {
      const int i = first + r * 64 + lane;
      const bool valid = r < rounds && i < m;
      keys_r[r] = valid ? ki[i] : 0ull;
      vals_r[r] = valid ? vi[i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cnt_s[wave][q * 64 + lane] = 0;
    // (a wave's own LDS stores are ordered before its own LDS atomics)
#pragma unroll
    for (int r = 0; r < kSmallSortRounds; ++r)
      if (r < rounds && first + r * 64 + lane < m) atomicAdd(&cnt_s[wave][(keys_r[r] >> shift) & 255], 1u);
    __syncthreads();
    uint32_t excl = 0, total = 0;
    if (tid < 256) {  // digit tid: exclusive prefix of its counts over the waves ...
#pragma unroll
      for (int w = 0; w < kSmallSortWaves; ++w) {
        const uint32_t c = cnt_s[w][tid];
        cnt_s[w][tid] = total;
        total += c;
      }
      uint32_t inc = total;  // ... and of the digit totals over the digits: shuffles, then one LDS round trip
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
      }
      if (lane == 63) wsum[wave] = inc;
      excl = inc - total;
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t before = 0;
      for (int w = 0; w < wave; ++w) before += wsum[w];
      base_s[tid] = before + excl;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) cnt_s[wave][q * 64 + lane] += base_s[q * 64 + lane];
    // each wave reads and updates only its own row from here on
    volatile uint32_t* run = cnt_s[wave];
#pragma unroll
    for (int r = 0; r < kSmallSortRounds; ++r) {
      if (r < rounds) {
        const bool valid = first + r * 64 + lane < m;
        const uint64_t key = keys_r[r];
        const unsigned d = (unsigned)(key >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const bool bit = (d >> b) & 1u;
          const uint64_t bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
        const int rank = __popcll(peers & lt);
        const int cnt = __popcll(peers);
        const uint32_t off = run[d];
        if (valid) {
          ko[off + rank] = key;
          vo[off + rank] = vals_r[r];
          if (rank == 0) run[d] = off + cnt;
        }
      }
    }
    __syncthreads();  // all stores of this pass are issued and complete before the next pass loads them
    uint64_t* tk = ki; ki = ko; ko = tk;
    uint32_t* tv = vi; vi = vo; vo = tv;
  }
}

size_t sort_pairs_ws_bytes(int64_t nmax) {
  int64_t nchunks = cdiv(nmax > 0 ? nmax : 1, kWaveItems);
  return align_up((size_t)nchunks * 256 * sizeof(uint32_t)) + align_up(256 * sizeof(uint32_t));
}

int sort_pairs(uint64_t* d_keys, uint32_t* d_vals, uint64_t* d_keys_alt, uint32_t* d_vals_alt,
               const int* d_count, int64_t nmax, int nbits, void* d_ws, size_t ws_bytes,
               hipStream_t stream, bool* result_in_alt) {
  if (result_in_alt) *result_in_alt = false;
  if (nmax <= 0) return M3D_OK;
  const int nchunks = (int)cdiv(nmax, kWaveItems);
  if (ws_bytes < sort_pairs_ws_bytes(nmax) || d_ws == nullptr) {
    set_error("sort_pairs: workspace too small");
    return M3D_ENOSPC;
  }
  uint32_t* counts = static_cast<uint32_t*>(d_ws);
  uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(d_ws) +
                                                 align_up((size_t)nchunks * 256 * sizeof(uint32_t)));
  int passes = (nbits + 7) / 8;
  if (passes < 1) passes = 1;
  if (passes > 8) passes = 8;
  if (result_in_alt) *result_in_alt = passes & 1;
  else if (passes & 1) ++passes;  // an even number of passes leaves the result in (d_keys, d_vals)
  if (nmax <= kSmallSortMax) {
    hipLaunchKernelGGL(radix_sort_small_kernel, dim3(1), dim3(kSmallSortWaves * 64), 0, stream, d_keys, d_vals,
                       d_keys_alt, d_vals_alt, d_count, (int)nmax, passes);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
  }
  const int nblocks = (int)cdiv(nchunks, kSortBlock / 64);
  uint64_t* ki = d_keys;
  uint32_t* vi = d_vals;
  uint64_t* ko = d_keys_alt;
  uint32_t* vo = d_vals_alt;
  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(kSortBlock), 0, stream, ki, d_count, (int)nmax,
                       shift, counts, nchunks);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(256), 0, stream, counts, totals, nchunks);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblocks), dim3(kSortBlock), 0, stream, ki, vi,
                       ko, vo, d_count, (int)nmax, shift, counts, totals, nchunks);
    uint64_t* tk = ki; ki = ko; ko = tk;
    uint32_t* tv = vi; vi = vo; vo = tv;
  }
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

// ------------------------------------------------------------------------------ unique
__global__ __launch_bounds__(256) void unique_insert_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                     uint64_t* __restrict__ tkeys, int32_t* __restrict__ trep,
                                     uint64_t mask, int32_t* __restrict__ pslot,
                                     uint64_t* __restrict__ ukeys, uint32_t* __restrict__ uslot,
                                     int* __restrict__ d_count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool is_new = false;
  uint64_t key = 0, s = 0;
  if (i < n) {
    key = keys[i];
    s = mix64(key) & mask;
    while (true) {
      // Most points repeat a key that is already in the table: look before the atomic.  A slot never changes once
      // it holds a key, so a stale read can only show "empty", which the compare-and-swap then settles.
      uint64_t old = *reinterpret_cast<const volatile uint64_t*>(&tkeys[s]);
      if (old == kEmptyKey)
        old = atomicCAS((unsigned long long*)&tkeys[s], (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (old == kEmptyKey) {
        trep[s] = (int32_t)i;
        is_new = true;
        break;
      }
      if (old == key) break;
      s = (s + 1) & mask;
    }
    pslot[i] = (int32_t)s;
  }
  // The list of distinct keys is appended to once per wave, not once per key: all new keys of a wave take
  // consecutive positions behind one atomic add on the shared counter.
  const uint64_t bal = __ballot(is_new);
  if (bal == 0) return;
  const int leader = __builtin_ctzll(bal);
  int base = 0;
  if (lane == leader) base = atomicAdd(d_count, __popcll(bal));
  base = __shfl(base, leader, 64);
  if (is_new) {
    const int pos = base + __popcll(bal & ((1ull << lane) - 1));
    ukeys[pos] = key;
    uslot[pos] = (uint32_t)s;
  }
}

__global__ void unique_rank_kernel(const uint32_t* __restrict__ uslot, const int* __restrict__ d_count,
                                   int32_t* __restrict__ trep, int32_t* __restrict__ rep_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *d_count) return;
  const uint32_t s = uslot[r];
  rep_out[r] = trep[s];
  trep[s] = (int32_t)r;  // the slot now holds the rank of its key
}

__global__ void unique_inverse_kernel(const int32_t* __restrict__ pslot, const int32_t* __restrict__ trank,
                                      int64_t n, int32_t* __restrict__ inverse) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inverse[i] = trank[pslot[i]];
}

static inline uint64_t table_capacity(int64_t n) { return next_pow2((uint64_t)(n > 0 ? n : 1) * 2); }

size_t unique_ws_bytes(int64_t n) {
  if (n <= 0) n = 1;
  const uint64_t cap = table_capacity(n);
  size_t b = 0;
  b += align_up(cap * sizeof(uint64_t));   // table keys
  b += align_up(cap * sizeof(int32_t));    // table rep / rank
  b += align_up(n * sizeof(int32_t));      // pslot
  b += align_up(n * sizeof(uint64_t));     // alt keys
  b += 2 * align_up(n * sizeof(uint32_t)); // slot values + alt
  b += sort_pairs_ws_bytes(n);
  return b;
}

int unique_sorted(const uint64_t* d_keys, int64_t n, int nbits, uint64_t* d_ukeys, int32_t* d_rep,
                  int32_t* d_inverse, int* d_count, void* d_ws, size_t ws_bytes,
                  hipStream_t stream) {
  M3D_HIP(hipMemsetAsync(d_count, 0, sizeof(int), stream));
  if (n <= 0) return M3D_OK;
  const uint64_t cap = table_capacity(n);
  Arena a(d_ws, ws_bytes);
  uint64_t* tkeys = a.take<uint64_t>(cap);
  int32_t* trep = a.take<int32_t>(cap);
  int32_t* pslot = a.take<int32_t>(n);
  uint64_t* kalt = a.take<uint64_t>(n);
  uint32_t* uslot = a.take<uint32_t>(n);
  uint32_t* valt = a.take<uint32_t>(n);
  const size_t sort_bytes = sort_pairs_ws_bytes(n);
  char* sort_ws = a.take<char>(sort_bytes);
  if (!a.ok) {
    set_error("unique_sorted: workspace too small (%zu needed, %zu given)", a.off, ws_bytes);
    return M3D_ENOSPC;
  }
  M3D_HIP(hipMemsetAsync(tkeys, 0xFF, cap * sizeof(uint64_t), stream));
  const int bs = 256;
  hipLaunchKernelGGL(unique_insert_kernel, dim3((unsigned)cdiv(n, bs)), dim3(bs), 0, stream, d_keys,
                     n, tkeys, trep, cap - 1, pslot, d_ukeys, uslot, d_count);
  int rc = sort_pairs(d_ukeys, uslot, kalt, valt, d_count, n, nbits, sort_ws, sort_bytes, stream);
  if (rc != M3D_OK) return rc;
  hipLaunchKernelGGL(unique_rank_kernel, dim3((unsigned)cdiv(n, bs)), dim3(bs), 0, stream, uslot,
                     d_count, trep, d_rep);
  hipLaunchKernelGGL(unique_inverse_kernel, dim3((unsigned)cdiv(n, bs)), dim3(bs), 0, stream, pslot,
                     trep, n, d_inverse);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

}  // namespace m3d

extern "C" int m3d_version(void) { return 1; }
extern "C" const char* m3d_last_error(void) { return m3d::get_error(); }
