// Shared host/device helpers for libm3d_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/m3d.h"

namespace m3d {

void set_error(const char* fmt, ...);
int f16_status_ptr(int** out);  // conv.hip: device address of the f16 status word of the current device

#define M3D_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      m3d::set_error(__VA_ARGS__);          \
      return M3D_EINVAL;                    \
    }                                       \
  } while (0)

#define M3D_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess) {                                                       \
      m3d::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,                 \
                     hipGetErrorString(e__));                                      \
      return M3D_EHIP;                                                             \
    }                                                                              \
  } while (0)

#define M3D_LAUNCH_CHECK() M3D_HIP(hipGetLastError())

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: one of these per kernel
// instantiation (a function-local static) remembers, per device, the largest size already granted.  Racing first
// calls from two threads both set the attribute (harmless); the remembered size is an atomic so that the race is
// only that.
struct LdsGrant {
  std::atomic<size_t> have[64] = {};
  int ensure(const void* fn, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      set_error("hipGetDevice failed or device index out of range");
      return M3D_EHIP;
    }
    if (lds > have[dev].load(std::memory_order_relaxed)) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(%zu bytes of LDS) -> %s", lds, hipGetErrorString(e));
        return M3D_EHIP;
      }
      have[dev].store(lds, std::memory_order_relaxed);
    }
    return M3D_OK;
  }
};

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided workspace.
struct Arena {
  char* base;
  size_t cap;
  size_t off = 0;
  bool ok = true;
  Arena(void* p, size_t n) : base(static_cast<char*>(p)), cap(n) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T));
    if (base == nullptr || off + bytes > cap) {
      ok = false;
      off += bytes;
      return nullptr;
    }
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
};

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

static inline int bits_for(uint64_t max_value) {  // number of bits needed to represent max_value
  int b = 0;
  while (max_value) { ++b; max_value >>= 1; }
  return b < 1 ? 1 : b;
}

static inline uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// ---------------------------------------------------------------- unique / sort primitives
// Sizes of the scratch the primitives below need (host).
size_t sort_pairs_ws_bytes(int64_t nmax);
size_t unique_ws_bytes(int64_t nmax);

// Stable LSD radix sort of (key, value) pairs, ascending by key bits [0, nbits).
// The element count is read from *d_count on the device (<= nmax); d_count == nullptr: all nmax elements.  On return the sorted
// pairs are in (d_keys, d_vals); (d_keys_alt, d_vals_alt) are clobbered.
// With result_in_alt != nullptr the pass count is not rounded up to an even number: *result_in_alt says
// whether the sorted pairs ended in the alt buffers (odd pass count) or in (d_keys, d_vals).
int sort_pairs(uint64_t* d_keys, uint32_t* d_vals, uint64_t* d_keys_alt, uint32_t* d_vals_alt,
               const int* d_count, int64_t nmax, int nbits, void* d_ws, size_t ws_bytes,
               hipStream_t stream, bool* result_in_alt = nullptr);

// np.unique(keys, return_inverse=True) plus one representative input index per unique key.
//   d_keys [n] input; outputs: d_ukeys [n] ascending unique keys, d_rep [n] an input index
//   holding each unique key, d_inverse [n] rank of every input key, d_count [1] number of uniques.
//   nbits: number of low key bits that can be non-zero (sorting passes are skipped above it).
int unique_sorted(const uint64_t* d_keys, int64_t n, int nbits, uint64_t* d_ukeys, int32_t* d_rep,
                  int32_t* d_inverse, int* d_count, void* d_ws, size_t ws_bytes,
                  hipStream_t stream);

// Open-addressing table of distinct keys -> value (insert all, then look up).
struct HashTable {
  uint64_t* keys;
  int32_t* vals;
  uint64_t mask;  // capacity - 1
};

__device__ inline int32_t table_find(const HashTable& t, uint64_t key) {
  uint64_t s = mix64(key) & t.mask;
  while (true) {
    uint64_t k = t.keys[s];
    if (k == key) return t.vals[s];
    if (k == kEmptyKey) return -1;
    s = (s + 1) & t.mask;
  }
}

}  // namespace m3d
