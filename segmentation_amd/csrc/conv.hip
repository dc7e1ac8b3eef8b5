// Sparse convolution forward for gfx950: output-stationary implicit GEMM on the exact-f32 matrix
// instruction v_mfma_f32_32x32x2_f32, with the per-channel affine (folded eval-mode PDBatchNorm),
// residual add and ReLU fused into the epilogue.
//
//   out[row(t)] = act(scale * sum_k W[k] . concat(in0, in1)[nbr[k][t]] + shift + residual[row(t)])
//
// Work decomposition
//   * a workgroup owns BM = 32*WM consecutive rows of the tile schedule and BN = 32*NT*WN output
//     channels; wave (wm, wn) owns 32 rows x 32*NT channels = NT accumulator tiles of 32x32;
//   * the reduction runs over (active kernel offset k) x (32-channel slab of the input); each
//     stage gathers BM x 32 input floats (one 128-B line per row) and BN x 32 weights into LDS
//     (row stride 36 floats: conflict-free ds_read_b128), double-buffered with register
//     prefetch so that one barrier per stage suffices;
//   * kernel offsets with no pair in the workgroup's rows are skipped through the 32-row tile
//     masks; a wave additionally skips the MFMAs of offsets absent from its own 32 rows;
//   * grid.z > 1 splits the kernel offsets over workgroups for small levels (partial slabs in
//     the workspace + a fixed-order reduce, so results are run-to-run deterministic).
//
// HBM/L2 layout: activations [rows, C] f32 row-major (a gathered row slab is one full 128-B
// line), weights [k][Cout][Cin] so that a weight slab is also 128 B per output channel.
#include "common.h"

#include <stdlib.h>

namespace m3d {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* in0;
  const float* in1;
  const float* weight;
  const int32_t* nbr;
  const int32_t* out_row;
  const uint32_t* tile_mask;
  const float* scale;
  const float* shift;
  const float* residual;
  float* out;
  float* slab;  // split-K partial sums [z][npad][cout], or nullptr
  int64_t n, npad, ldo;
  int c0, c1, cout, kvol, relu;
  int dbg;  // timing-only ablation bits (M3D_DBG); 0 in normal operation
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Accumulator register r of a 32x32 tile is row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31.
template <int NT>
__device__ inline void conv_epilogue(const ConvArgs& a, const floatx16 (&acc)[NT], const int* orow_s,
                                     int64_t tile_base, int col_base, int wm, int wn, int lane) {
  const int h = lane >> 5, cl = lane & 31;
  if (a.slab) {
    float* dst = a.slab + ((int64_t)blockIdx.z * a.npad + tile_base + wm * 32) * a.cout + col_base;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = (wn * NT + nt) * 32 + cl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
        dst[(int64_t)rl * a.cout + col] = acc[nt][r];
      }
    }
    return;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = col_base + (wn * NT + nt) * 32 + cl;
    const float sc = a.scale ? a.scale[col] : 1.f;
    const float sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int orow = orow_s[wm * 32 + rl];
      if (orow >= 0) {
        float v = acc[nt][r] * sc + sh;
        if (a.residual) v += a.residual[(int64_t)orow * a.cout + col];
        if (a.relu) v = fmaxf(v, 0.f);
        a.out[(int64_t)orow * a.ldo + col] = v;
      }
    }
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Map block ids so
// that each XCD walks a contiguous range of the (spatially sorted) tile schedule: neighbouring tiles
// gather overlapping input rows, which then hit in that XCD's L2.  Bijective for any tile count.
__device__ inline int64_t xcd_tile(unsigned bid, unsigned ntiles) {
  const unsigned xcd = bid & 7u, q = ntiles >> 3, r = ntiles & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return (int64_t)base + (bid >> 3);
}

// Active kernel offsets of a workgroup (bmask) and of one of its 32-row waves (wmask); with
// gridDim.z > 1 the offsets are dealt round-robin over z.
template <int WM>
__device__ inline void conv_masks(const ConvArgs& a, int64_t tile_base, int wm, uint32_t& bmask,
                                  uint32_t& wmask) {
  bmask = 0;
  wmask = 0;
  if (a.tile_mask) {
#pragma unroll
    for (int w = 0; w < WM; ++w) {
      const uint32_t m = a.tile_mask[tile_base / 32 + w];
      bmask |= m;
      if (w == wm) wmask = m;
    }
  } else {
    bmask = wmask = 1u;
  }
  if (gridDim.z > 1) {
    uint32_t keep = 0;
    int ord = 0;
    for (uint32_t m = bmask; m; m &= m - 1, ++ord)
      if (ord % (int)gridDim.z == (int)blockIdx.z) keep |= m & (0u - m);
    bmask = keep;
    wmask &= keep;
  }
  bmask = __builtin_amdgcn_readfirstlane(bmask);
  wmask = __builtin_amdgcn_readfirstlane(wmask);
}

constexpr int kBK = 32;
constexpr int kLdr = 36;  // LDS row stride in floats

template <int WM, int WN, int NT>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_f32_kernel(ConvArgs a) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  constexpr int A_V = (BM * 8 + NTHR - 1) / NTHR;  // float4 per thread, A slab
  constexpr int W_V = (BN * 8 + NTHR - 1) / NTHR;  // float4 per thread, W slab
  extern __shared__ __align__(16) float lds[];
  float* As = lds;                         // [2][BM][kLdr]
  float* Ws = lds + 2 * BM * kLdr;         // [2][BN][kLdr]
  int* orow_s = reinterpret_cast<int*>(lds + 2 * (BM + BN) * kLdr);  // [BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t tile_base = xcd_tile(blockIdx.x, gridDim.x) * BM;
  const int col_base = blockIdx.y * BN;
  const int cin = a.c0 + a.c1;
  const int nchunk = cin / kBK;

  uint32_t bmask, wmask;
  conv_masks<WM>(a, tile_base, wm, bmask, wmask);

  for (int r = tid; r < BM; r += NTHR) {
    const int64_t t = tile_base + r;
    orow_s[r] = a.out_row ? a.out_row[t] : (t < a.n ? (int)t : -1);
  }

  floatx16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const int nk = __popc(bmask);
  const int nstage = nk * nchunk;

  float4 ra[A_V], rw[W_V];
  int ridx[A_V];

  // gather row indices of kernel offset k for this thread's A rows
  auto load_idx = [&](int k) {
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int e = tid + v * NTHR;
      const int r = e >> 3;
      int id = -1;
      if (e < BM * 8) {
        const int64_t t = tile_base + r;
        id = a.nbr ? a.nbr[(int64_t)k * a.npad + t] : (t < a.n ? (int)t : -1);
      }
      ridx[v] = id;
    }
  };
  auto issue = [&](int k, int cc) {
    const float* src;
    int ld, col;
    if (cc * kBK < a.c0) { src = a.in0; ld = a.c0; col = cc * kBK; }
    else { src = a.in1; ld = a.c1; col = cc * kBK - a.c0; }
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int e = tid + v * NTHR;
      const int c4 = e & 7;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ridx[v] >= 0)
        x = *reinterpret_cast<const float4*>(src + (int64_t)ridx[v] * ld + col + c4 * 4);
      ra[v] = x;
    }
#pragma unroll
    for (int v = 0; v < W_V; ++v) {
      const int e = tid + v * NTHR;
      const int j = e >> 3, c4 = e & 7;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < BN * 8)
        x = *reinterpret_cast<const float4*>(a.weight + ((int64_t)k * a.cout + col_base + j) * cin +
                                             cc * kBK + c4 * 4);
      rw[v] = x;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int e = tid + v * NTHR;
      if (e < BM * 8)
        *reinterpret_cast<float4*>(As + (buf * BM + (e >> 3)) * kLdr + (e & 7) * 4) = ra[v];
    }
#pragma unroll
    for (int v = 0; v < W_V; ++v) {
      const int e = tid + v * NTHR;
      if (e < BN * 8)
        *reinterpret_cast<float4*>(Ws + (buf * BN + (e >> 3)) * kLdr + (e & 7) * 4) = rw[v];
    }
  };

  if (nstage > 0) {
    uint32_t rem = bmask;
    int k = __builtin_ctz(rem);
    int cc = 0;
    load_idx(k);
    issue(k, cc);
    commit(0);
    __syncthreads();
    const int arow = (wm * 32 + (lane & 31)) * kLdr + 4 * (lane >> 5);
    for (int s = 0; s < nstage; ++s) {
      const int kcur = k;
      // advance (k, cc) to the next stage and prefetch it into registers
      int kn = k, ccn = cc + 1;
      if (ccn == nchunk) {
        ccn = 0;
        rem &= rem - 1;
        kn = rem ? __builtin_ctz(rem) : k;
      }
      const bool more = s + 1 < nstage;
      if (more) {
        if (kn != k) load_idx(kn);
        issue(kn, ccn);
      }
      // matrix work on the current buffer
      if ((wmask >> kcur) & 1u) {
        const float* Ab = As + (s & 1) * BM * kLdr + arow;
        const float* Wb = Ws + ((s & 1) * BN + (wn * NT) * 32 + (lane & 31)) * kLdr + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < kBK / 8; ++g) {
          const float4 av = *reinterpret_cast<const float4*>(Ab + g * 8);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float4 bv = *reinterpret_cast<const float4*>(Wb + nt * 32 * kLdr + g * 8);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[nt], 0, 0, 0);
          }
        }
      }
      if (more) commit((s + 1) & 1);
      __syncthreads();
      k = kn;
      cc = ccn;
    }
  } else {
    __syncthreads();
  }

  conv_epilogue<NT>(a, acc, orow_s, tile_base, col_base, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------
// Split-bf16 variant: every f32 multiplicand is split into T bf16 terms (x = x0 + x1 (+ x2), each
// term the bf16 rounding of the remainder) and the product is formed from the leading cross terms
// on the bf16 matrix instruction v_mfma_f32_32x32x16_bf16 with f32 accumulation:
//   T = 2: x0*w0 + x0*w1 + x1*w0                       (relative error ~2^-17 per product)
//   T = 3: + x1*w1 + x0*w2 + x2*w0                     (relative error ~2^-24, i.e. f32 level)
// bf16 products are exact in f32, so only the dropped cross terms and the accumulation round.
// The bf16 pipe is 16x the f32 matrix rate per k, so 3 (6) products cost 3/16 (6/16) of the exact
// kernel's matrix time.  Weights arrive pre-split ([k][Cout][Cin/32][T][32] bf16); activations are
// gathered as f32 and split while they are committed to LDS.
constexpr int kRowB = 80;  // LDS bytes per row and term: 32 bf16 + 16 B pad (conflict-free b128 reads)

template <int WM, int WN, int NT, int T>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_split_kernel(ConvArgs a) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  constexpr int A_V = (BM * 8 + NTHR - 1) / NTHR;       // float4 per thread, A slab
  constexpr int W_E = BN * T * 4;                        // 16-B pieces of the W slab
  constexpr int W_V = (W_E + NTHR - 1) / NTHR;
  constexpr int A_PLANE = BM * kRowB, W_PLANE = BN * kRowB;
  constexpr int BUF = T * (A_PLANE + W_PLANE);
  extern __shared__ __align__(16) char ldsb[];
  int* orow_s = reinterpret_cast<int*>(ldsb + 2 * BUF);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t tile_base = xcd_tile(blockIdx.x, gridDim.x) * BM;
  const int col_base = blockIdx.y * BN;
  const int cin = a.c0 + a.c1;
  const int nchunk = cin / kBK;
  const char* wsplit = reinterpret_cast<const char*>(a.weight);

  uint32_t bmask, wmask;
  conv_masks<WM>(a, tile_base, wm, bmask, wmask);

  for (int r = tid; r < BM; r += NTHR) {
    const int64_t t = tile_base + r;
    orow_s[r] = a.out_row ? a.out_row[t] : (t < a.n ? (int)t : -1);
  }

  floatx16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const int nstage = __popc(bmask) * nchunk;
  float4 ra[A_V];
  uint4 rw[W_V];
  int ridx[A_V];

  auto load_idx = [&](int k) {
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int e = tid + v * NTHR;
      int id = -1;
      if (e < BM * 8) {
        const int64_t t = tile_base + (e >> 3);
        id = a.nbr ? a.nbr[(int64_t)k * a.npad + t] : (t < a.n ? (int)t : -1);
      }
      ridx[v] = id;
    }
  };
  auto issue = [&](int k, int cc) {
    const float* src;
    int ld, col;
    if (cc * kBK < a.c0) { src = a.in0; ld = a.c0; col = cc * kBK; }
    else { src = a.in1; ld = a.c1; col = cc * kBK - a.c0; }
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int e = tid + v * NTHR;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ridx[v] >= 0 && !(a.dbg & 2))
        x = *reinterpret_cast<const float4*>(src + (int64_t)ridx[v] * ld + col + (e & 7) * 4);
      ra[v] = x;
    }
#pragma unroll
    for (int v = 0; v < W_V; ++v) {
      const int e = tid + v * NTHR;
      const int j = e / (T * 4), p = e % (T * 4);
      uint4 x = make_uint4(0, 0, 0, 0);
      if (e < W_E && !(a.dbg & 4))
        x = *reinterpret_cast<const uint4*>(
            wsplit + (((int64_t)k * a.cout + col_base + j) * nchunk + cc) * (T * 64) + p * 16);
      rw[v] = x;
    }
  };
  auto commit = [&](int buf) {
    char* Ab = ldsb + buf * BUF;
    char* Wb = Ab + T * A_PLANE;
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int e = tid + v * NTHR;
      if (e < BM * 8) {
        float x[4] = {ra[v].x, ra[v].y, ra[v].z, ra[v].w};
        char* dst = Ab + (e >> 3) * kRowB + (e & 7) * 8;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          bf16x4 q;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            q[i] = (__bf16)x[i];
            x[i] -= (float)q[i];
          }
          *reinterpret_cast<bf16x4*>(dst + t * A_PLANE) = q;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < W_V; ++v) {
      const int e = tid + v * NTHR;
      if (e < W_E) {
        const int j = e / (T * 4), p = e % (T * 4);
        *reinterpret_cast<uint4*>(Wb + (p >> 2) * W_PLANE + j * kRowB + (p & 3) * 16) = rw[v];
      }
    }
  };

  if (nstage > 0) {
    uint32_t rem = bmask;
    int k = __builtin_ctz(rem);
    int cc = 0;
    load_idx(k);
    issue(k, cc);
    commit(0);
    __syncthreads();
    const int aoff = (wm * 32 + (lane & 31)) * kRowB + (lane >> 5) * 16;
    const int woff = ((wn * NT) * 32 + (lane & 31)) * kRowB + (lane >> 5) * 16;
    for (int s = 0; s < nstage; ++s) {
      const int kcur = k;
      int kn = k, ccn = cc + 1;
      if (ccn == nchunk) {
        ccn = 0;
        rem &= rem - 1;
        kn = rem ? __builtin_ctz(rem) : k;
      }
      const bool more = s + 1 < nstage;
      if (more) {
        if (kn != k && !(a.dbg & 16)) load_idx(kn);
        issue(kn, ccn);
      }
      if (((wmask >> kcur) & 1u) && !(a.dbg & 1)) {
        const char* Ab = ldsb + (s & 1) * BUF + aoff;
        const char* Wb = ldsb + (s & 1) * BUF + T * A_PLANE + woff;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 af[T];
#pragma unroll
          for (int t = 0; t < T; ++t) af[t] = *reinterpret_cast<const bf16x8*>(Ab + t * A_PLANE + ks * 32);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            bf16x8 bf[T];
#pragma unroll
            for (int t = 0; t < T; ++t)
              bf[t] = *reinterpret_cast<const bf16x8*>(Wb + t * W_PLANE + nt * 32 * kRowB + ks * 32);
            if (T == 3) {
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[T - 1], acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[T - 1], bf[0], acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[nt], 0, 0, 0);
            }
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[nt], 0, 0, 0);
          }
        }
      }
      if (more && !(a.dbg & 8)) commit((s + 1) & 1);
      __syncthreads();
      k = kn;
      cc = ccn;
    }
  } else {
    __syncthreads();
  }
  conv_epilogue<NT>(a, acc, orow_s, tile_base, col_base, wm, wn, lane);
}

// Reference layout [Cout][k][Cin] f32 -> kernel layouts.
//   terms == 0: [k][Cout][Cin] f32
//   terms == 2/3: [k][Cout][Cin/32][terms][32] bf16 (term t = bf16 rounding of the remainder)
__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int kvol, int cin, int terms,
                                   void* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)cout * kvol * cin;
  if (e >= total) return;
  const int c = (int)(e % cin);
  const int k = (int)((e / cin) % kvol);
  const int o = (int)(e / ((int64_t)cin * kvol));
  float x = w[e];
  if (terms == 0) {
    static_cast<float*>(out)[((int64_t)k * cout + o) * cin + c] = x;
    return;
  }
  __bf16* dst = static_cast<__bf16*>(out) +
                ((((int64_t)k * cout + o) * (cin / 32) + c / 32) * terms) * 32 + (c & 31);
  for (int t = 0; t < terms; ++t) {
    const __bf16 q = (__bf16)x;
    dst[t * 32] = q;
    x -= (float)q;
  }
}

// Fixed-order reduction of the split-K slabs + epilogue.
__global__ __launch_bounds__(256) void conv_reduce_kernel(ConvArgs a, int nz) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = a.npad * a.cout;
  if (e >= total) return;
  const int64_t t = e / a.cout;
  const int col = (int)(e % a.cout);
  const int orow = a.out_row ? a.out_row[t] : (t < a.n ? (int)t : -1);
  if (orow < 0) return;
  float v = 0.f;
  for (int z = 0; z < nz; ++z) v += a.slab[(int64_t)z * total + e];
  v = v * (a.scale ? a.scale[col] : 1.f) + (a.shift ? a.shift[col] : 0.f);
  if (a.residual) v += a.residual[(int64_t)orow * a.cout + col];
  if (a.relu) v = fmaxf(v, 0.f);
  a.out[(int64_t)orow * a.ldo + col] = v;
}

// Any channel counts (used for the 3-channel stem and for shapes that are not multiples of 32):
// one thread per (schedule row, output channel), plain f32 FMA chain over (k, cin).
__global__ __launch_bounds__(256) void conv_direct_f32_kernel(ConvArgs a) {
  extern __shared__ float wsm[];  // weights, when they fit
  const int cin = a.c0 + a.c1;
  const int64_t wtotal = (int64_t)a.kvol * a.cout * cin;
  const bool w_in_lds = wtotal * sizeof(float) <= 60 * 1024;
  if (w_in_lds) {
    for (int64_t i = threadIdx.x; i < wtotal; i += blockDim.x) wsm[i] = a.weight[i];
    __syncthreads();
  }
  const float* w = w_in_lds ? wsm : a.weight;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.npad * a.cout) return;
  const int64_t t = e / a.cout;
  const int col = (int)(e % a.cout);
  const int orow = a.out_row ? a.out_row[t] : (t < a.n ? (int)t : -1);
  if (orow < 0) return;
  float acc = 0.f;
  for (int k = 0; k < a.kvol; ++k) {
    const int id = a.nbr ? a.nbr[(int64_t)k * a.npad + t] : (int)t;
    if (id < 0) continue;
    const float* wr = w + ((int64_t)k * a.cout + col) * cin;
    const float* x0 = a.in0 + (int64_t)id * a.c0;
    for (int c = 0; c < a.c0; ++c) acc = fmaf(x0[c], wr[c], acc);
    if (a.c1 > 0) {
      const float* x1 = a.in1 + (int64_t)id * a.c1;
      for (int c = 0; c < a.c1; ++c) acc = fmaf(x1[c], wr[a.c0 + c], acc);
    }
  }
  float v = acc * (a.scale ? a.scale[col] : 1.f) + (a.shift ? a.shift[col] : 0.f);
  if (a.residual) v += a.residual[(int64_t)orow * a.cout + col];
  if (a.relu) v = fmaxf(v, 0.f);
  a.out[(int64_t)orow * a.ldo + col] = v;
}

template <int WM, int WN, int NT>
static int launch_igemm(const ConvArgs& a, int nz, hipStream_t stream) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  const size_t lds = (size_t)(2 * (BM + BN) * kLdr) * sizeof(float) + BM * sizeof(int);
  static bool attr_done = false;
  if (!attr_done) {
    M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32_kernel<WM, WN, NT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid((unsigned)(a.npad / BM), (unsigned)(a.cout / BN), (unsigned)nz);
  hipLaunchKernelGGL((conv_igemm_f32_kernel<WM, WN, NT>), grid, dim3(NTHR), lds, stream, a);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

template <int WM, int WN, int NT, int T>
static int launch_split(const ConvArgs& a, int nz, hipStream_t stream) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  const size_t lds = (size_t)2 * T * (BM + BN) * kRowB + BM * sizeof(int);
  static bool attr_done = false;
  if (!attr_done) {
    M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_split_kernel<WM, WN, NT, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid((unsigned)(a.npad / BM), (unsigned)(a.cout / BN), (unsigned)nz);
  hipLaunchKernelGGL((conv_igemm_split_kernel<WM, WN, NT, T>), grid, dim3(NTHR), lds, stream, a);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

// Tile configuration: the largest row tile that still yields >= ~2 workgroups per CU.
struct ConvPlan {
  int wm, wn, nt, nz;
};

static ConvPlan plan_conv(int64_t npad, int cout, int kvol) {
  int bn;
  if (cout % 128 == 0) bn = 128;
  else if (cout % 96 == 0) bn = 96;
  else if (cout % 64 == 0) bn = 64;
  else bn = 32;
  const int64_t ncol = cout / bn;
  ConvPlan p{4, 1, bn / 32, 1};
  const int64_t target = 512;
  if ((npad / 128) * ncol >= target) return p;
  if ((npad / 64) * ncol >= target) {
    if (bn == 128) return ConvPlan{2, 2, 2, 1};
    if (bn == 64) return ConvPlan{2, 2, 1, 1};
    if (bn == 96) return ConvPlan{2, 1, 3, 1};
    return ConvPlan{2, 1, 1, 1};
  }
  p = ConvPlan{1, bn / 32, 1, 1};
  const int64_t blocks = (npad / 32) * ncol;
  if (blocks < target && kvol > 1) {
    int nz = (int)((target + blocks - 1) / blocks);
    if (nz > kvol) nz = kvol;
    if (nz > 9) nz = 9;
    p.nz = nz;
  }
  return p;
}

}  // namespace m3d

using namespace m3d;

extern "C" size_t m3d_conv_workspace_bytes(int64_t n, int cout, int k) {
  const int64_t npad = m3d_schedule_npad(n);
  if (cout <= 0 || cout % 32 != 0) return 256;  // served by the direct kernel, which needs no scratch
  ConvPlan p = plan_conv(npad, cout, k);
  if (p.nz <= 1) return 256;
  return align_up((size_t)p.nz * npad * cout * sizeof(float));
}

extern "C" int m3d_conv_fwd_f32(const float* d_in0, int c0, const float* d_in1, int c1,
                                const float* d_weight, int k, int cout, const int32_t* d_nbr,
                                const int32_t* d_out_row, const uint32_t* d_tile_mask, int64_t n,
                                int64_t npad, const float* d_scale, const float* d_shift,
                                const float* d_residual, int relu, float* d_out, int64_t ldo,
                                int precision, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n >= 0, "m3d_conv_fwd_f32: n < 0");
  if (n == 0) return M3D_OK;
  M3D_CHECK_ARG(d_in0 && d_weight && d_out, "m3d_conv_fwd_f32: null pointer");
  M3D_CHECK_ARG(c0 > 0 && c1 >= 0 && cout > 0 && k >= 1 && k <= 27, "m3d_conv_fwd_f32: bad channel/kernel sizes");
  M3D_CHECK_ARG((c1 == 0) == (d_in1 == nullptr), "m3d_conv_fwd_f32: in1/c1 mismatch");
  M3D_CHECK_ARG(ldo >= cout, "m3d_conv_fwd_f32: ldo < cout");
  M3D_CHECK_ARG(precision >= 0 && precision <= 2, "m3d_conv_fwd_f32: unknown precision %d", precision);
  const bool dense = d_nbr == nullptr;
  if (dense) {
    M3D_CHECK_ARG(k == 1 && d_out_row == nullptr && d_tile_mask == nullptr,
                  "m3d_conv_fwd_f32: dense mode needs k == 1 and no schedule");
    npad = m3d_schedule_npad(n);
  } else {
    M3D_CHECK_ARG(d_out_row && d_tile_mask, "m3d_conv_fwd_f32: schedule pointers missing");
    M3D_CHECK_ARG(npad >= n && npad % 128 == 0, "m3d_conv_fwd_f32: npad=%lld must be a multiple of 128 >= n",
                  (long long)npad);
  }
  M3D_CHECK_ARG(npad < (int64_t)1 << 31, "m3d_conv_fwd_f32: too many rows");
  ConvArgs a;
  a.in0 = d_in0; a.in1 = d_in1; a.weight = d_weight; a.nbr = d_nbr; a.out_row = d_out_row;
  a.tile_mask = d_tile_mask; a.scale = d_scale; a.shift = d_shift; a.residual = d_residual;
  a.out = d_out; a.slab = nullptr; a.n = n; a.npad = npad; a.ldo = ldo; a.c0 = c0; a.c1 = c1;
  a.cout = cout; a.kvol = k; a.relu = relu;
  {
    const char* e = getenv("M3D_DBG");
    a.dbg = e ? atoi(e) : 0;
  }

  const bool mfma_ok = (c0 % 32 == 0) && (c1 % 32 == 0) && (cout % 32 == 0);
  if (!mfma_ok) {
    const int64_t total = npad * cout;
    const size_t wbytes = (size_t)k * cout * (c0 + c1) * sizeof(float);
    const size_t lds = wbytes <= 60 * 1024 ? wbytes : 0;
    hipLaunchKernelGGL(conv_direct_f32_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), lds, stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
  }
  const ConvPlan p = plan_conv(npad, cout, k);
  if (p.nz > 1) {
    const size_t need = (size_t)p.nz * npad * cout * sizeof(float);
    if (d_ws == nullptr || ws_bytes < need) {
      set_error("m3d_conv_fwd_f32: workspace too small (%zu given, %zu needed)", ws_bytes, need);
      return M3D_ENOSPC;
    }
    a.slab = static_cast<float*>(d_ws);
  }
  int rc = M3D_EINVAL;
#define M3D_CASE(WM_, WN_, NT_)                                                       \
  if (p.wm == WM_ && p.wn == WN_ && p.nt == NT_) {                                    \
    if (precision == M3D_PREC_F32) rc = launch_igemm<WM_, WN_, NT_>(a, p.nz, stream); \
    else if (precision == M3D_PREC_BF16X3) rc = launch_split<WM_, WN_, NT_, 2>(a, p.nz, stream); \
    else rc = launch_split<WM_, WN_, NT_, 3>(a, p.nz, stream);                        \
  }
  M3D_CASE(4, 1, 1) M3D_CASE(4, 1, 2) M3D_CASE(4, 1, 3) M3D_CASE(4, 1, 4)
  M3D_CASE(2, 2, 1) M3D_CASE(2, 2, 2) M3D_CASE(2, 1, 3) M3D_CASE(2, 1, 1)
  M3D_CASE(1, 4, 1) M3D_CASE(1, 3, 1) M3D_CASE(1, 2, 1) M3D_CASE(1, 1, 1)
#undef M3D_CASE
  if (rc != M3D_OK) {
    if (rc == M3D_EINVAL) set_error("m3d_conv_fwd_f32: no kernel for plan (%d,%d,%d)", p.wm, p.wn, p.nt);
    return rc;
  }
  if (p.nz > 1) {
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)cdiv(npad * cout, 256)), dim3(256), 0, stream, a,
                       p.nz);
    M3D_LAUNCH_CHECK();
  }
  return M3D_OK;
}

extern "C" size_t m3d_conv_packed_weight_bytes(int cout, int k, int cin, int precision) {
  const bool mfma_ok = cin % 32 == 0 && cout % 32 == 0;
  const size_t elems = (size_t)cout * k * cin;
  if (precision == M3D_PREC_F32 || !mfma_ok) return elems * sizeof(float);
  return elems * 2 * (precision == M3D_PREC_BF16X3 ? 2 : 3);
}

extern "C" int m3d_conv_pack_weight(const float* d_weight_ref, int cout, int k, int cin, int precision,
                                    void* d_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(d_weight_ref && d_out && cout > 0 && k > 0 && cin > 0, "m3d_conv_pack_weight: bad arguments");
  M3D_CHECK_ARG(precision >= 0 && precision <= 2, "m3d_conv_pack_weight: unknown precision %d", precision);
  const bool mfma_ok = cin % 32 == 0 && cout % 32 == 0;
  const int terms = (precision == M3D_PREC_F32 || !mfma_ok) ? 0 : (precision == M3D_PREC_BF16X3 ? 2 : 3);
  const int64_t total = (int64_t)cout * k * cin;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, d_weight_ref,
                     cout, k, cin, terms, d_out);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}
