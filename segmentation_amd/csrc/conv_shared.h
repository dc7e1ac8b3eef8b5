// Pieces shared by the convolution kernels: argument block, sf32 helpers, epilogues, tile bookkeeping.
#pragma once
#include "common.h"

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
  // split-format ("sf32") operands of the LDS-DMA kernel: [rows + 1][C/G][T][G] bf16 (G = 16 for T = 3, else 32),
  // last row zero
  const char* s0;
  const char* s1;
  int64_t zero0, zero1;  // index of the all-zero row of s0 / s1
  char* out_split;       // optional sf32 copy of the output (cout/32 groups per row)
  int terms;             // T of out_split
  const int32_t* tile_order;  // optional: tile taken by workgroup blockIdx.x of a 128-row-tile launch
  const char* zbuf;           // >= 64 zero bytes: what a missing neighbour gathers when s0 / s1 are plain f32 rows
  int* f16_status;            // set to 1 when a value written in the f16 pair format is beyond the f16 range
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

// Split format ("sf32") of an [rows, c] tensor with T terms: [rows + 1][c/G][T][G] bf16, G = 16 channels per group
// for T = 3 and 32 otherwise (T = 1 is plain row-major bf16); term t = bf16 rounding of what the earlier terms left
// over; row `rows` is all zero.  split_group_shift(T) = log2 G; split_ptr = address of term 0 of (row, col), the other
// terms follow at a stride of 2 G bytes.
//
// `terms` doubles as the format code: 1, 2, 3 = that many bf16 terms; kFmtF16 (4) = the f16 pair: two f16 terms in
// the geometry of two bf16 terms, term 0 = f16 rounding of the value, term 1 = f16 rounding of (value - term 0) * 2^11
// (the scale keeps the second term out of the f16 subnormals; products with one second term are accumulated
// apart and scaled back by 2^-11).  An operand is then exact to 2^-22 of its value, and a value beyond the f16
// range (65504) raises the library's f16 status word instead of going unnoticed.
constexpr int kFmtF16 = 4;
// Kernel-internal code of the LDS-DMA convolution (never a storage format of a caller): both operands are plain f32
// rows, 32 channels = 128 bytes per unit -- the geometry of two 16-bit terms -- and the products are exact f32
// (v_mfma_f32_32x32x2_f32).
constexpr int kFmtF32 = 5;
constexpr float kF16Scale = 2048.f, kF16Unscale = 1.f / 2048.f, kF16Max = 65504.f;
__host__ __device__ inline int split_nterms(int fmt) { return fmt == kFmtF16 ? 2 : fmt; }
__host__ __device__ inline int split_group_shift(int terms) { return terms == 3 ? 4 : 5; }

__device__ inline char* split_ptr(char* base, int64_t row, int col, int c, int terms) {
  const int gs = split_group_shift(terms);
  return base + (((row * (c >> gs) + (col >> gs)) * split_nterms(terms)) << (gs + 1)) + ((col & ((1 << gs) - 1)) << 1);
}

// All terms of N consecutive channels (N = 1, 4 or 8; same group) of one row: term t goes to base + t * tstride.
template <int N>
__device__ inline void split_store_n(char* base, int tstride, int fmt, float (&v)[N], int* f16_status) {
  if (fmt == kFmtF16) {
    typedef _Float16 hN __attribute__((ext_vector_type(N)));
    _Float16 q0[N], q1[N];
    bool over = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      over |= !(fabsf(v[i]) <= kF16Max);  // also true for NaN
      q0[i] = (_Float16)v[i];
      q1[i] = (_Float16)((v[i] - (float)q0[i]) * kF16Scale);
    }
    if (over && f16_status) *f16_status = 1;
    if constexpr (N == 1) {
      *reinterpret_cast<_Float16*>(base) = q0[0];
      *reinterpret_cast<_Float16*>(base + tstride) = q1[0];
    } else {
      hN w0, w1;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        w0[i] = q0[i];
        w1[i] = q1[i];
      }
      *reinterpret_cast<hN*>(base) = w0;
      *reinterpret_cast<hN*>(base + tstride) = w1;
    }
    return;
  }
  for (int t = 0; t < fmt; ++t) {
    if constexpr (N == 1) {
      const __bf16 q = (__bf16)v[0];
      *reinterpret_cast<__bf16*>(base + t * tstride) = q;
      v[0] -= (float)q;
    } else {
      typedef __bf16 bN __attribute__((ext_vector_type(N)));
      bN q;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        q[i] = (__bf16)v[i];
        v[i] -= (float)q[i];
      }
      *reinterpret_cast<bN*>(base + t * tstride) = q;
    }
  }
}

// One element of an sf32 tensor.
__device__ inline void store_split(char* base, int64_t row, int col, int c, int terms, float v, int* f16_status = nullptr) {
  float one[1] = {v};
  split_store_n<1>(split_ptr(base, row, col, c, terms), 2 << split_group_shift(terms), terms, one, f16_status);
}

// The all-zero row (index n) of an sf32 output; called by one workgroup of the kernel that writes the output.
__device__ inline void clear_zero_row(const ConvArgs& a) {
  if (a.out_split == nullptr) return;
  const int nt = split_nterms(a.terms);
  const int words = a.cout * nt / 2;  // 4-byte words in one row
  uint32_t* p = reinterpret_cast<uint32_t*>(a.out_split + (int64_t)a.n * a.cout * nt * 2);
  for (int i = threadIdx.x; i < words; i += blockDim.x) p[i] = 0u;
}

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
  if (blockIdx.x == 0 && blockIdx.y == 0) clear_zero_row(a);
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
        if (a.out) a.out[(int64_t)orow * a.ldo + col] = v;
        if (a.out_split) store_split(a.out_split, orow, col, a.cout, a.terms, v, a.f16_status);
      }
    }
  }
}

// Epilogue through an LDS image of the output tile, in two steps.  conv_stage_tile: the accumulators of one
// 32-row x 32*NT-column wave tile (per-channel affine applied) go to S[rows][BN] f32.  conv_store_tile (after a
// barrier): every thread finishes 4 consecutive channels of one row (residual, ReLU) and stores them with 16-byte
// (f32) / 8-byte (bf16) accesses: the f32 row segment and/or the sf32 terms.
template <int NT>
__device__ inline void conv_stage_tile(const ConvArgs& a, const floatx16 (&acc)[NT], float* S, int bn, int row_tile,
                                       int col_tile0, int col_base, int lane) {
  const int h = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int lc = (col_tile0 + nt) * 32 + cl;
    const float sc = a.scale ? a.scale[col_base + lc] : 1.f;
    const float sh = a.shift ? a.shift[col_base + lc] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
      S[(row_tile * 32 + rl) * bn + lc] = acc[nt][r] * sc + sh;
    }
  }
}

template <int BM, int BN, int NTHR>
__device__ inline void conv_store_tile(const ConvArgs& a, const float* S, const int* orow_s, int col_base, int tid) {
  if (a.out == nullptr && a.out_split != nullptr) {
    // split output only: 8-channel pieces, one 16-byte store per term and thread
    constexpr int CPR8 = BN / 8;
    for (int e = tid; e < BM * CPR8; e += NTHR) {
      const int r = e / CPR8, c8 = e % CPR8;
      const int orow = orow_s[r];
      if (orow < 0) continue;
      const float4 v0 = *reinterpret_cast<const float4*>(S + r * BN + c8 * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(S + r * BN + c8 * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const int col = col_base + c8 * 8;
      if (a.residual) {
        const float4* q = reinterpret_cast<const float4*>(a.residual + (int64_t)orow * a.cout + col);
        const float4 q0 = q[0], q1 = q[1];
        v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w;
        v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
      }
      if (a.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
      }
      split_store_n<8>(split_ptr(a.out_split, orow, col, a.cout, a.terms), 2 << split_group_shift(a.terms), a.terms, v,
                       a.f16_status);
    }
    return;
  }
  constexpr int CPR = BN / 4;  // 4-channel pieces per row: consecutive lanes store consecutive 16 B (f32) / 8 B (bf16)
  for (int e = tid; e < BM * CPR; e += NTHR) {
    const int r = e / CPR, c4 = e % CPR;
    const int orow = orow_s[r];
    if (orow < 0) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(S + r * BN + c4 * 4);
    float v[4] = {v0.x, v0.y, v0.z, v0.w};
    const int col = col_base + c4 * 4;
    if (a.residual) {
      const float4 q0 = *reinterpret_cast<const float4*>(a.residual + (int64_t)orow * a.cout + col);
      v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w;
    }
    if (a.relu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (a.out) {
      float* o = a.out + (int64_t)orow * a.ldo + col;
      if ((a.ldo & 3) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[i];
      }
    }
    if (a.out_split)
      split_store_n<4>(split_ptr(a.out_split, orow, col, a.cout, a.terms), 2 << split_group_shift(a.terms), a.terms, v,
                       a.f16_status);
  }
}

template <int WM, int WN, int NT>
__device__ inline void conv_epilogue_staged(const ConvArgs& a, const floatx16 (&acc)[NT], const int* orow_s,
                                            float* S, int col_base, int wm, int wn, int lane, int tid) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  if (blockIdx.x == 0 && blockIdx.y == 0) clear_zero_row(a);
  conv_stage_tile<NT>(a, acc, S, BN, wm, wn * NT, col_base, lane);
  __syncthreads();
  conv_store_tile<BM, BN, NTHR>(a, S, orow_s, col_base, tid);
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

// LDS image geometry of the LDS-DMA kernels.  The reduction runs over units = (kernel offset, channel group of G
// channels); a unit of one row is PIECES 16-byte pieces in HBM (all terms of the group, contiguous).  An LDS row is
// T = 2: one unit of 32 channels (8 pieces, 128 bytes); T = 1: two consecutive units of 32 channels (4 pieces each,
// 128 bytes), so that a stage carries as much matrix work per byte as the others; T = 3: one unit of 16 channels
// (6 pieces, 96 bytes).  Piece p of row r sits in slot p ^ swz(r), which keeps ds_read_b128 fragment reads
// conflict-free although the image is written lane-linearly by the DMA: a ds_read_b128 serves 16 lanes at a time,
// whose rows cover every residue mod 16; with 128-byte rows two rows share a 256-byte bank line and the eight
// row pairs take the eight XOR patterns; with 96-byte rows the rows r and r + 8 start on the same banks and take
// the two slots of a piece pair.
// T = kFmtF16 (the f16 pair) and T = kFmtF32 (plain f32 rows, 32 channels per unit) have the geometry of T = 2.
template <int T>
struct DmaGeom {
  static constexpr int G = T == 3 ? 16 : 32;   // channels per unit
  static constexpr int UPS = T == 1 ? 2 : 1;   // units per stage
  static constexpr int PIECES = T == 1 ? 4 : (T == 3 ? 6 : 8);
  static constexpr int SLOTS = T == 3 ? 6 : 8;
  static constexpr int ROWB = SLOTS * 16;
  __device__ static inline int swz(int r) { return T == 3 ? ((r >> 3) & 1) : ((r >> 1) & 7); }
};

// Activations as plain f32 rows [rows][C] for the three-term products: a unit is 16 channels = 64 bytes = 4 pieces of
// one row (two thirds of the bytes of the pre-split form); the terms are formed in registers after the fragment read.
// Rows r, r + 4, r + 8, r + 12 start on the same banks and take the four XOR patterns.
struct DmaGeomF32 {
  static constexpr int G = 16;
  static constexpr int UPS = 1;
  static constexpr int PIECES = 4;
  static constexpr int SLOTS = 4;
  static constexpr int ROWB = 64;
  __device__ static inline int swz(int r) { return (r >> 2) & 3; }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

}  // namespace m3d
