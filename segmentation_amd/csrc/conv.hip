// Sparse convolution forward for gfx950: output-stationary implicit GEMM on the matrix cores, with the per-channel
// affine (folded eval-mode PDBatchNorm), residual add and ReLU fused into the epilogue.
//
//   out[row(t)] = act(scale * sum_k W[k] . concat(in0, in1)[nbr[k][t]] + shift + residual[row(t)])
//
// Work decomposition
//   * a workgroup owns BM = 32*WM consecutive rows of the tile schedule and BN = 32*NT*WN output
//     channels; wave (wm, wn) owns 32 rows x 32*NT channels = NT accumulator tiles of 32x32;
//   * the reduction runs over (active kernel offset k) x (channel group of the input); each stage moves BM gathered
//     input rows and BN weight rows of one group into LDS (LDS-DMA kernel: conv_igemm_dma_body; the older
//     register-staged kernel for in-kernel splits: conv_igemm_split_kernel), double-buffered, one barrier per stage;
//   * products: exact f32 (v_mfma_f32_32x32x2_f32 on plain f32 rows), or one to three 16-bit terms per operand on
//     v_mfma_f32_32x32x16_{bf16,f16};
//   * kernel offsets with no pair in the workgroup's rows are skipped through the 32-row tile
//     masks; a wave additionally skips the MFMAs of offsets absent from its own 32 rows;
//   * grid.z > 1 splits the kernel offsets over workgroups for small levels (partial slabs in
//     the workspace + a fixed-order reduce, so results are run-to-run deterministic).
//
// HBM/L2 layout: activations [rows, C] f32 row-major or the split format of conv_shared.h, weights [k][Cout][Cin] f32
// or [k][Cout][Cin/G][T][G] 16-bit terms.
#include <mutex>
#include <type_traits>
#include "conv_shared.h"

#include <stdlib.h>

namespace m3d {


// ------------------------------------------------------------------------------------------
// Split-bf16 variant: every f32 multiplicand is split into T bf16 terms (x = x0 + x1 (+ x2), each
// term the bf16 rounding of the remainder) and the product is formed from the leading cross terms
// on the bf16 matrix instruction v_mfma_f32_32x32x16_bf16 with f32 accumulation:
//   T = 2: x0*w0 + x0*w1 + x1*w0                       (relative error ~2^-17 per product)
//   T = 3: + x1*w1 + x0*w2 + x2*w0                     (relative error ~2^-24, i.e. f32 level)
// bf16 products are exact in f32, so only the dropped cross terms and the accumulation round.
// The bf16 pipe is 16x the f32 matrix rate per k, so 3 (6) products cost 3/16 (6/16) of the exact
// kernel's matrix time.  T = 1 is plain bf16: one term, one product.  Weights arrive pre-split
// ([k][Cout][Cin/G][T][G] bf16, G = 16 for T = 3 and 32 otherwise); activations are gathered as f32 and
// split while they are committed to LDS.
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
  const int64_t tile_base = (BM == 128 && a.tile_order ? (int64_t)a.tile_order[blockIdx.x] : xcd_tile(blockIdx.x, gridDim.x)) * BM;
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
      if (ridx[v] >= 0)
        x = *reinterpret_cast<const float4*>(src + (int64_t)ridx[v] * ld + col + (e & 7) * 4);
      ra[v] = x;
    }
#pragma unroll
    for (int v = 0; v < W_V; ++v) {
      const int e = tid + v * NTHR;
      const int j = e / (T * 4), p = e % (T * 4);  // p = 4 * term + 8-channel piece of the 32-channel slab
      uint4 x = make_uint4(0, 0, 0, 0);
      if (e < W_E) {
        const int64_t row = (int64_t)k * a.cout + col_base + j;
        const char* src = T == 3 ? wsplit + (row * (2 * nchunk) + 2 * cc + ((p >> 1) & 1)) * 96 + (p >> 2) * 32 + (p & 1) * 16
                                 : wsplit + (row * nchunk + cc) * (T * 64) + p * 16;
        x = *reinterpret_cast<const uint4*>(src);
      }
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
        if (kn != k) load_idx(kn);
        issue(kn, ccn);
      }
      if ((wmask >> kcur) & 1u) {
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
            if constexpr (T == 3) {
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[T - 1], acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[T - 1], bf[0], acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[nt], 0, 0, 0);
            }
            if constexpr (T >= 2) {
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[T - 1 > 0 ? 1 : 0], acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[T - 1 > 0 ? 1 : 0], bf[0], acc[nt], 0, 0, 0);
            }
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[nt], 0, 0, 0);
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
// LDS-DMA variant of the split-bf16 kernel.  Activations arrive already split ("sf32":
// [rows + 1][C/G][T][G] bf16, the extra row all zero), so a stage is a pure byte move: every lane
// issues global_load_lds_dwordx4 with its own source address (row gather) and the hardware writes the
// 64 lanes' 16-B pieces to consecutive LDS slots -- no staging registers, no VALU, no ds_write.
// The reduction walks units = (active kernel offset, group of G channels).  LDS image: one row per gathered
// row and stage: T = 2 one unit of 32 channels (terms 0 and 1, 128 bytes), T = 3 one unit of 16 channels
// (three terms, 96 bytes), T = 1 two consecutive units of 32 channels (128 bytes) -- every stage then
// carries 12-18 matrix instructions per wave for 21-28 KB of LDS, at two workgroups per CU.
// Slot p' of row r holds piece p' ^ swz(r), so the fragment reads (16 lanes, distinct rows, same piece)
// hit distinct banks although the image is lane-linear.  Missing neighbours read the zero row.
// Pipeline: two LDS buffers; the DMA of stage s+1 is issued right after the barrier that publishes stage s
// and lands while stage s is multiplied (s_waitcnt vmcnt, raw s_barrier); the kernel map of the tile is
// staged in LDS once so that no ordinary global load sits between DMAs.
// T = 3 keeps two accumulator sets: the leading products a0*w0 and everything else (2^-8 and below).  The
// small products then round against their own small running sum instead of the large one, and the result is
// more accurate than a single f32 accumulation chain over the same terms.

#ifdef M3D_STAMP
// Diagnostic build only (make stamp): where a stage of the LDS-DMA kernel spends its cycles.  Sums over all waves of
// [0] cycles from loop top to past the barrier, [1] issuing the next stage's DMA, [2] matrix work, [3] stages,
// [4] prologue, [5] epilogue.  Never compiled into the shipped library.
__device__ unsigned long long m3d_stamp_sums[8];
#define M3D_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#endif

// Size the buffer descriptors of the LDS-DMA kernel claim for every operand: per-lane offsets from here up are out of
// range (the request returns zeros), everything an operand holds must lie below.
constexpr uint32_t kDmaBufferBytes = 0xFFFFFF00u;

struct UnitIt {  // position in the unit sequence of a tile: kernel offset k (lowest set bit of rem), channel group cc
  uint32_t rem;
  int k, cc;
};

// AF: the activations are plain f32 rows (s0 / s1 = [rows][C] f32, no zero row: missing neighbours read zbuf) and
// the terms are formed in registers after the fragment read (T == 2 or 3); the weights are split as always.
// GMAP: the kernel map stays in global memory; LDS holds the rows of two kernel offsets only (the one whose requests
// are being issued and the next one, fetched by an LDS-DMA of its own one offset ahead), which leaves room for a third
// workgroup per CU beside the 128-row x 96-column output staging.
template <int WM, int WN, int NT, int T, int NBUF, bool AF, bool GMAP>
__device__ __forceinline__ void conv_igemm_dma_body(const ConvArgs& a) {
  using G = DmaGeom<T>;                                                        // weight image
  // activation image: 16 f32 channels are 64 bytes (T = 3); 32 f32 channels are the 128 bytes of the two-term unit
  using GA = typename std::conditional<AF && T == 3, DmaGeomF32, DmaGeom<T>>::type;
  static_assert(!AF || T == 3 || T == 2 || T == kFmtF16 || T == kFmtF32, "f32 activations: split products of two or three terms, or exact f32");
  static_assert(AF || T != kFmtF32, "exact f32 products take plain f32 rows");
  static_assert(GA::G == G::G && GA::UPS == G::UPS, "activation and weight units cover the same channels");
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  constexpr int A_BYTES = BM * GA::ROWB, W_BYTES = BN * G::ROWB, BUF = A_BYTES + W_BYTES;
  // DMA instructions per thread and stage; where the slot count does not divide by the thread count the last
  // instruction is issued by the leading waves only (and its last wave by its leading lanes)
  constexpr int A_V = (BM * GA::SLOTS + NTHR - 1) / NTHR, W_V = (BN * G::SLOTS + NTHR - 1) / NTHR;
  constexpr int GRPB = G::PIECES * 16;   // bytes of one (output channel, unit) of the weights in HBM
  constexpr int GRPA = GA::PIECES * 16;  // bytes of one (row, unit) of the activations in HBM
  constexpr int UPS = G::UPS;
  static_assert(A_BYTES % 256 == 0 && BUF % 256 == 0, "every image starts on a bank line");
  // slot lists that divide evenly by the thread count and have no unused slot need no lane mask
  constexpr bool A_EXACT = BM * GA::SLOTS % NTHR == 0, W_EXACT = BN * G::SLOTS % NTHR == 0;
  constexpr bool NO_DEAD = G::PIECES * UPS == G::SLOTS && GA::PIECES * UPS == GA::SLOTS;
  static_assert(NBUF >= 2 && NBUF <= 4, "ring depth");
  extern __shared__ __align__(16) char ldsb[];
  constexpr int RING = NBUF * BUF > BM * BN * 4 ? NBUF * BUF : BM * BN * 4;  // the output tile is staged here too
  int* idx_s = reinterpret_cast<int*>(ldsb + RING);  // [kvol][BM]; GMAP: [2][BM]
  int* orow_s = idx_s + (GMAP ? 2 : a.kvol) * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int64_t tile_base = (BM == 128 && a.tile_order ? (int64_t)a.tile_order[blockIdx.x] : xcd_tile(blockIdx.x, gridDim.x)) * BM;
  const int col_base = blockIdx.y * BN;
  const int nchunk = (a.c0 + a.c1) / G::G;
  const int g0 = a.c0 / G::G;
  const char* wsplit = reinterpret_cast<const char*>(a.weight);
  const uint32_t rowb0 = (uint32_t)g0 * GRPA, rowb1 = (uint32_t)(a.c1 / G::G) * GRPA;
  // Buffer addressing of the stage requests (split activations, one unit per stage): a request is
  // `buffer_load_dwordx4 ... offen lds` with a 32-bit per-lane byte offset that only changes with the kernel offset
  // (gathered row * row bytes + piece) and a scalar offset that carries the channel group of the stage, so a stage
  // spends no vector instruction on addresses (with 64-bit per-lane addresses it spent three per request, and the
  // request itself cost its wave more).  A missing neighbour gets an offset beyond the buffer's size: the hardware
  // then writes zeros to LDS without a memory request.  Operands are limited to kDmaBufferBytes (checked by the host).
  constexpr bool BUFA = UPS == 1 && T != kFmtF32;
  static_assert(!GMAP || (BUFA && BM == 128 && NBUF == 2), "kernel map in global memory: 128-row tiles of the buffer-addressed path");
  constexpr int kRsrcWord3 = 0x00020000;  // raw buffer, 32-bit elements
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.s0), (short)0, (int)kDmaBufferBytes, kRsrcWord3);
  const __amdgpu_buffer_rsrc_t rs1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.s1 ? a.s1 : a.s0), (short)0, (int)kDmaBufferBytes, kRsrcWord3);
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wsplit), (short)0, (int)kDmaBufferBytes, kRsrcWord3);
  uint32_t a_vo0[A_V], a_vo1[A_V];

  uint32_t bmask, wmask;
  conv_masks<WM>(a, tile_base, wm, bmask, wmask);

  for (int r = tid; r < BM; r += NTHR) {
    const int64_t t = tile_base + r;
    orow_s[r] = a.out_row ? a.out_row[t] : (t < a.n ? (int)t : -1);
  }
  if constexpr (GMAP) {  // the rows of the first two active offsets
    const uint32_t rem1 = bmask & (bmask - 1);
    if (tid < BM && bmask) idx_s[tid] = a.nbr[(int64_t)__builtin_ctz(bmask) * a.npad + tile_base + tid];
    if (tid >= NTHR - BM && rem1) idx_s[tid - (NTHR - 2 * BM)] = a.nbr[(int64_t)__builtin_ctz(rem1) * a.npad + tile_base + tid - (NTHR - BM)];
  } else {
    for (int e = tid; e < a.kvol * BM; e += NTHR) {
      const int k = e / BM, r = e % BM;
      const int64_t t = tile_base + r;
      idx_s[e] = a.nbr ? a.nbr[(int64_t)k * a.npad + t] : (t < a.n ? (int)t : -1);
    }
  }
  __syncthreads();

  constexpr int NACC = (T == 3 || T == kFmtF16) ? 2 : 1;
  floatx16 acc[NACC][NT];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[c][i][j] = 0.f;

  const int nunit = __popc(bmask) * nchunk;
  const int nstage = (nunit + UPS - 1) / UPS;

  // Loop-invariant per-thread addressing.  Slot e = v * NTHR + tid of a buffer is (row e / SLOTS, physical
  // slot e % SLOTS) and carries logical piece p = slot ^ swz(row); with two units per stage (T = 1) pieces
  // 0-3 belong to the stage's first unit and 4-7 to its second.
  int a_r[A_V];
  uint32_t a_p[A_V];
  bool a_on[A_V], a_u[A_V];
#pragma unroll
  for (int v = 0; v < A_V; ++v) {
    const int e = v * NTHR + tid;
    a_r[v] = e / GA::SLOTS;
    const int p = (e % GA::SLOTS) ^ GA::swz(a_r[v]);
    a_u[v] = UPS == 2 && p >= GA::PIECES;
    a_p[v] = (uint32_t)(UPS == 2 ? p & 3 : p) * 16u;
    a_on[v] = e < BM * GA::SLOTS && p < GA::PIECES * UPS;
  }
  uint32_t w_o[W_V];
  bool w_on[W_V], w_u[W_V];
#pragma unroll
  for (int v = 0; v < W_V; ++v) {
    const int e = v * NTHR + tid;
    const int j = e / G::SLOTS;
    const int p = (e % G::SLOTS) ^ G::swz(j);
    w_u[v] = UPS == 2 && p >= G::PIECES;
    w_o[v] = (uint32_t)((col_base + j) * nchunk) * GRPB + (uint32_t)(UPS == 2 ? p & 3 : p) * 16u;
    w_on[v] = e < BN * G::SLOTS && p < G::PIECES * UPS;
  }
  const uint32_t wk_stride = (uint32_t)(a.cout * nchunk) * GRPB;  // bytes of one kernel offset's weights

  // (Lambdas here are only ever called from the kernel body itself, never from one another.)
  auto step = [&](UnitIt& u) {
    if (++u.cc == nchunk) {
      u.cc = 0;
      u.rem &= u.rem - 1;
      if (u.rem) u.k = __builtin_ctz(u.rem);
    }
  };
  // Issue iterator: `it` is the next unit to request, a_id the gathered rows of the units of the next stage.
  UnitIt it{bmask, bmask ? __builtin_ctz(bmask) : 0, 0};
  int issued = 0;  // units requested so far
  int a_id[A_V];
  int it_par = 0;  // GMAP: which of the two LDS rows holds the kernel map of it.k
  auto load_ids = [&]() {
    UnitIt n1 = it;
    if (UPS == 2) step(n1);
#pragma unroll
    for (int v = 0; v < A_V; ++v) a_id[v] = idx_s[(GMAP ? it_par : a_u[v] ? n1.k : it.k) * BM + a_r[v]];
    if constexpr (BUFA) {
#pragma unroll
      for (int v = 0; v < A_V; ++v) {
        a_vo0[v] = a_id[v] < 0 ? kDmaBufferBytes : (uint32_t)a_id[v] * rowb0 + a_p[v];
        a_vo1[v] = a_id[v] < 0 ? kDmaBufferBytes : (uint32_t)a_id[v] * rowb1 + a_p[v];
      }
    }
  };
  load_ids();
  // DMA instructions this wave issues per stage (the same for every stage): the counted waits below depend on it
  int gw = 0;
#pragma unroll
  for (int v = 0; v < A_V; ++v) gw += (v * NTHR + wave * 64 < BM * GA::SLOTS) ? 1 : 0;
#pragma unroll
  for (int v = 0; v < W_V; ++v) gw += (v * NTHR + wave * 64 < BN * G::SLOTS) ? 1 : 0;
  auto issue_next = [&](int bi) {
    char* buf = ldsb + bi * BUF;
    if constexpr (BUFA) {
      const int k_before = it.k;
      const bool first = it.cc < g0;
      const uint32_t soff_a = (uint32_t)(first ? it.cc : it.cc - g0) * GRPA;
      const uint32_t soff_w = (uint32_t)it.k * wk_stride + (uint32_t)it.cc * GRPB;
      step(it);
      ++issued;
      if (first) {
#pragma unroll
        for (int v = 0; v < A_V; ++v)
          if (A_EXACT || v * NTHR + wave * 64 < BM * GA::SLOTS)
            if ((A_EXACT && NO_DEAD) || a_on[v])
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(buf + (v * NTHR + wave * 64) * 16), 16, a_vo0[v],
                                                       soff_a, 0, 0);
      } else {
#pragma unroll
        for (int v = 0; v < A_V; ++v)
          if (A_EXACT || v * NTHR + wave * 64 < BM * GA::SLOTS)
            if ((A_EXACT && NO_DEAD) || a_on[v])
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(buf + (v * NTHR + wave * 64) * 16), 16, a_vo1[v],
                                                       soff_a, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < W_V; ++v)
        if (W_EXACT || v * NTHR + wave * 64 < BN * G::SLOTS)
          if ((W_EXACT && NO_DEAD) || w_on[v])
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(buf + A_BYTES + (v * NTHR + wave * 64) * 16), 16,
                                                     w_o[v], soff_w, 0, 0);
      if (issued < nunit && it.k != k_before) {
        if constexpr (GMAP) {
          it_par ^= 1;
          // the row the previous offset used is free (every wave read it before this stage's barrier): it receives
          // the offset after this one, which lands and is published long before its first request is formed
          const uint32_t rem2 = it.rem & (it.rem - 1);
          if (rem2 && wave < BM / 64)
            __builtin_amdgcn_global_load_lds(a.nbr + ((int64_t)__builtin_ctz(rem2) * a.npad + tile_base + wave * 64 + lane),
                                             (lds_ptr_t)(idx_s + (it_par ^ 1) * BM + wave * 64), 4, 0, 0);
        }
        load_ids();
      }
      return;
    }
    const char* src[UPS];
    const char* wsrc[UPS];
    uint32_t rowb[UPS];
    int zero[UPS];
    bool on[UPS];
    const int k_before = it.k;
#pragma unroll
    for (int u = 0; u < UPS; ++u) {
      on[u] = issued < nunit;
      const bool first = it.cc < g0;
      src[u] = (first ? a.s0 : a.s1) + (first ? it.cc : it.cc - g0) * GRPA;
      rowb[u] = first ? rowb0 : rowb1;
      zero[u] = (int)(first ? a.zero0 : a.zero1);
      wsrc[u] = wsplit + ((uint64_t)it.k * wk_stride + (uint32_t)it.cc * GRPB);
      if (on[u]) {
        step(it);
        ++issued;
      }
    }
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const bool u1 = UPS == 2 && a_u[v];
      const int z = u1 ? zero[UPS - 1] : zero[0];
      const uint32_t row = (uint32_t)(a_id[v] < 0 ? z : a_id[v]);
      const char* g = (u1 ? src[UPS - 1] : src[0]) + ((uint64_t)row * (u1 ? rowb[UPS - 1] : rowb[0]) + a_p[v]);
      if (AF && a_id[v] < 0) g = a.zbuf + a_p[v];
      if (A_EXACT || v * NTHR + wave * 64 < BM * GA::SLOTS) {  // wave-uniform: this instruction exists for this wave
        if (((A_EXACT && NO_DEAD) || a_on[v]) && (UPS == 1 || !u1 || on[UPS - 1]))
          __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(buf + (v * NTHR + wave * 64) * 16), 16, 0, 0);
      }
    }
#pragma unroll
    for (int v = 0; v < W_V; ++v) {
      const bool u1 = UPS == 2 && w_u[v];
      if (W_EXACT || v * NTHR + wave * 64 < BN * G::SLOTS) {
        if (((W_EXACT && NO_DEAD) || w_on[v]) && (UPS == 1 || !u1 || on[UPS - 1]))
          __builtin_amdgcn_global_load_lds((u1 ? wsrc[UPS - 1] : wsrc[0]) + w_o[v],
                                           (lds_ptr_t)(buf + A_BYTES + (v * NTHR + wave * 64) * 16), 16, 0, 0);
      }
    }
    // gathered rows of the next stage's units: with one unit per stage they only change with the kernel offset
    if (issued < nunit && (UPS == 2 || it.k != k_before)) load_ids();
  };

  // fragment addresses (loop invariant): row of A / W this lane reads, with the swizzle applied per piece.
  // Fragment q (0..3) is the pair of pieces 2q, 2q + 1 (one per half-wave): T = 2: q = 2 * term + k-step;
  // T = 3: q = term; T = 1: q = 2 * unit + k-step.
  constexpr int NQ = T == 3 ? 3 : 4;
  const int r32 = lane & 31, h = lane >> 5;
  const int arow = wm * 32 + r32;
  uint32_t a_off[NQ], w_off[NT][NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    // f32 rows: the lane's eight consecutive channels of k-step ks are pieces 4 ks + 2h and 4 ks + 2h + 1
    // (T = 3: q = 0, 1, one k-step; T = 2: q = 2 ks + 0, 1)
    // exact f32: no terms to form, the same pieces as the weights (2 q + h)
    a_off[q] = AF && T != kFmtF32 ? arow * GA::ROWB + (((4 * (q >> 1) + 2 * h + (q & 1)) ^ GA::swz(arow)) * 16)
                                  : arow * GA::ROWB + (((2 * q + h) ^ GA::swz(arow)) * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = (wn * NT + nt) * 32 + r32;
      w_off[nt][q] = A_BYTES + j * G::ROWB + (((2 * q + h) ^ G::swz(j)) * 16);
    }
  }

  // matrix work of one stage; m0 / m1: this wave has pairs at the kernel offset of the stage's first / second unit
  auto compute = [&](int bi, bool m0, bool m1) {
    const char* buf = ldsb + bi * BUF;
    if constexpr (T == 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 0 ? !m0 : !m1) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(buf + a_off[u * 2 + ks]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(buf + w_off[nt][u * 2 + ks]);
            acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[0][nt], 0, 0, 0);
          }
        }
      }
    } else if constexpr (T == kFmtF16) {
      if (!m0) return;
      // f16 pair: the two products with one (scaled) second term go to their own accumulator
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        halfx8 af[2];
        if constexpr (AF) {
          const float4 lo = *reinterpret_cast<const float4*>(buf + a_off[2 * ks]);
          const float4 hi = *reinterpret_cast<const float4*>(buf + a_off[2 * ks + 1]);
          const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            af[0][i] = (_Float16)v[i];
            af[1][i] = (_Float16)((v[i] - (float)af[0][i]) * kF16Scale);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t) af[t] = *reinterpret_cast<const halfx8*>(buf + a_off[t * 2 + ks]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          halfx8 bf[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) bf[t] = *reinterpret_cast<const halfx8*>(buf + w_off[nt][t * 2 + ks]);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[1], acc[1][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[0], acc[1][nt], 0, 0, 0);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[0], acc[0][nt], 0, 0, 0);
        }
      }
    } else if constexpr (T == kFmtF32) {
      if (!m0) return;
      // exact f32 products: the lane's four 16-byte pieces hold channels 8 q + 4 h + j (q = piece pair, h = half-wave,
      // j = 0..3); v_mfma_f32_32x32x2_f32 number (q, j) multiplies the channel pair (8 q + j, 8 q + 4 + j), the two
      // halves of the wave supplying one channel each, for the activations and the weights alike
      float4 af[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) af[q] = *reinterpret_cast<const float4*>(buf + a_off[q]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bf = *reinterpret_cast<const float4*>(buf + w_off[nt][q]);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf.x, acc[0][nt], 0, 0, 0);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf.y, acc[0][nt], 0, 0, 0);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf.z, acc[0][nt], 0, 0, 0);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf.w, acc[0][nt], 0, 0, 0);
        }
      }
    } else if constexpr (T == 2) {
      if (!m0) return;
      // products (activation term, weight term), smallest first: (0,1) (1,0) (0,0)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[2];
        if constexpr (AF) {
          const float4 lo = *reinterpret_cast<const float4*>(buf + a_off[2 * ks]);
          const float4 hi = *reinterpret_cast<const float4*>(buf + a_off[2 * ks + 1]);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              af[t][i] = (__bf16)v[i];
              v[i] -= (float)af[t][i];
            }
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t) af[t] = *reinterpret_cast<const bf16x8*>(buf + a_off[t * 2 + ks]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf16x8 bf[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(buf + w_off[nt][t * 2 + ks]);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[0][nt], 0, 0, 0);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[0][nt], 0, 0, 0);
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[0][nt], 0, 0, 0);
        }
      }
    } else {
      if (!m0) return;
      bf16x8 af[3];
      if constexpr (AF) {
        const float4 lo = *reinterpret_cast<const float4*>(buf + a_off[0]);
        const float4 hi = *reinterpret_cast<const float4*>(buf + a_off[1]);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int t = 0; t < 3; ++t)  // term t = bf16 rounding of what the earlier terms left over (as store_split)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            af[t][i] = (__bf16)v[i];
            v[i] -= (float)af[t][i];
          }
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) af[t] = *reinterpret_cast<const bf16x8*>(buf + a_off[t]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 bf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(buf + w_off[nt][t]);
        // the five products of relative size 2^-8 and below go to their own accumulator, smallest first
        acc[NACC - 1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[NACC - 1][nt], 0, 0, 0);
        acc[NACC - 1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[NACC - 1][nt], 0, 0, 0);
        acc[NACC - 1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[NACC - 1][nt], 0, 0, 0);
        acc[NACC - 1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[NACC - 1][nt], 0, 0, 0);
        acc[NACC - 1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[NACC - 1][nt], 0, 0, 0);
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[0][nt], 0, 0, 0);
      }
    }
  };

  if (nstage > 0) {
    // Ring of NBUF buffers, stage s lives in buffer s % NBUF.  The requests of up to NBUF - 2 later stages may still
    // be in flight when stage s is consumed; the buffer that stage s - 1 just released receives stage s + NBUF - 1.
    // One barrier per stage: it publishes stage s to every wave and, because every wave has then finished reading
    // stage s - 1, frees that buffer.  Every wave issues gw DMA instructions per stage, in stage order, so "all but
    // the last j * gw have landed" retires exactly the stages up to s.
    int bi_issue = 0;
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u)
      if (u < nstage) {
        issue_next(u);
        bi_issue = u + 1;
      }
    if (bi_issue == NBUF) bi_issue = 0;
    UnitIt ct{bmask, __builtin_ctz(bmask), 0};
    int done = 0;  // units consumed so far
    int bi = 0;
#ifdef M3D_STAMP
    unsigned long long st_wait = 0, st_issue = 0, st_comp = 0;
#endif
    for (int s = 0; s < nstage; ++s) {
#ifdef M3D_STAMP
      M3D_T(t0);
#endif
      const int ahead = nstage - 1 - s;  // later stages already requested: min(ahead, NBUF - 2)
      if (NBUF >= 4 && ahead >= 2) {
        if (gw == 7) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else if (gw == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (gw == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (gw == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (gw == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (gw == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (NBUF >= 3 && ahead >= 1) {
        if (gw == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (gw == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (gw == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (gw == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (gw == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (gw == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef M3D_STAMP
      M3D_T(t1);
#endif
      if (issued < nunit) {
        issue_next(bi_issue);
        if (++bi_issue == NBUF) bi_issue = 0;
      }
#ifdef M3D_STAMP
      M3D_T(t2);
#endif
      const bool m0 = (wmask >> ct.k) & 1u;
      step(ct);
      ++done;
      bool m1 = false;
      if (UPS == 2 && done < nunit) {
        m1 = (wmask >> ct.k) & 1u;
        step(ct);
        ++done;
      }
      compute(bi, m0, m1);
#ifdef M3D_STAMP
      M3D_T(t3);
      st_wait += t1 - t0; st_issue += t2 - t1; st_comp += t3 - t2;
#endif
      if (++bi == NBUF) bi = 0;
    }
#ifdef M3D_STAMP
    if (lane == 0) {
      atomicAdd(&m3d_stamp_sums[0], st_wait);
      atomicAdd(&m3d_stamp_sums[1], st_issue);
      atomicAdd(&m3d_stamp_sums[2], st_comp);
      atomicAdd(&m3d_stamp_sums[3], (unsigned long long)nstage);
    }
#endif
  }
  if constexpr (NACC == 2) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[0][i][j] += T == kFmtF16 ? acc[1][i][j] * kF16Unscale : acc[1][i][j];
  }
  if (a.slab) {
    conv_epilogue<NT>(a, acc[0], orow_s, tile_base, col_base, wm, wn, lane);
  } else {
    __syncthreads();  // every wave is done with the staging buffers; reuse them for the output tile
    conv_epilogue_staged<WM, WN, NT>(a, acc[0], orow_s, reinterpret_cast<float*>(ldsb), col_base, wm, wn, lane, tid);
  }
}

template <int WM, int WN, int NT, int T, int NBUF, bool AF = false, bool GMAP = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_dma_kernel(ConvArgs a) {
  conv_igemm_dma_body<WM, WN, NT, T, NBUF, AF, GMAP>(a);
}

// f32 [n][c] -> sf32 [n + 1][c/32][terms][32] bf16 (row n zero).  One thread per element.
__global__ void split_f32_kernel(const float* __restrict__ in, int64_t n, int c, int terms, char* __restrict__ out,
                                 int* __restrict__ f16_status) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (n + 1) * c) return;
  const int64_t row = e / c;
  const int col = (int)(e % c);
  store_split(out, row, col, c, terms, row < n ? in[e] : 0.f, f16_status);
}

// Reference layout [Cout][k][Cin] f32 -> kernel layouts.
//   terms == 0: [k][Cout][Cin] f32
//   terms == 1/2/3: [k][Cout][Cin/G][terms][G] bf16 (term t = bf16 rounding of the remainder; G = 16 for three
//   terms, 32 otherwise)
// transposed != 0: the packed weight is W'[o][k][c] = w[c][flip ? kvol - 1 - k : k][o] (w has the reference layout
// [cin'][kvol][cout'] of the layer it came from; cout/cin below are those of W'): the weight of the input-gradient
// convolution, formed without a transposed copy of w.
__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int kvol, int cin, int terms,
                                   int transposed, int flip, void* __restrict__ out, int* __restrict__ f16_status) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)cout * kvol * cin;
  if (e >= total) return;
  const int c = (int)(e % cin);
  const int k = (int)((e / cin) % kvol);
  const int o = (int)(e / ((int64_t)cin * kvol));
  float x = transposed ? w[((int64_t)c * kvol + (flip ? kvol - 1 - k : k)) * cout + o] : w[e];
  if (terms == 0) {
    static_cast<float*>(out)[((int64_t)k * cout + o) * cin + c] = x;
    return;
  }
  const int gs = split_group_shift(terms), g = 1 << gs;
  __bf16* dst = static_cast<__bf16*>(out) +
                (((((int64_t)k * cout + o) * (cin >> gs) + (c >> gs)) * split_nterms(terms)) << gs) + (c & (g - 1));
  if (terms == kFmtF16) {
    float one[1] = {x};
    split_store_n<1>(reinterpret_cast<char*>(dst), 2 * g, terms, one, f16_status);
    return;
  }
  for (int t = 0; t < terms; ++t) {
    const __bf16 q = (__bf16)x;
    dst[t * g] = q;
    x -= (float)q;
  }
}

// Fixed-order reduction of the split-K slabs + epilogue.
__global__ __launch_bounds__(256) void conv_reduce_kernel(ConvArgs a, int nz) {
  if (blockIdx.x == 0) clear_zero_row(a);
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
  if (a.out) a.out[(int64_t)orow * a.ldo + col] = v;
  if (a.out_split) store_split(a.out_split, orow, col, a.cout, a.terms, v, a.f16_status);
}

// Few input channels (the 3-channel stem): one thread per output row computes CO output channels; the
// weights sit in LDS as [k][cin][CO] and are read as wave-wide broadcasts, the kernel map is read once per
// (row, offset) instead of once per output channel.
template <int CO>
__global__ __launch_bounds__(256) void conv_rowwise_f32_kernel(ConvArgs a) {
  extern __shared__ __align__(16) float wsm[];  // [kvol][cin][CO]
  const int cin = a.c0;
  const int co_base = blockIdx.y * CO;
  for (int i = threadIdx.x; i < a.kvol * cin * CO; i += blockDim.x) {
    const int o = i % CO, c = (i / CO) % cin, k = i / (CO * cin);
    wsm[i] = a.weight[((int64_t)k * a.cout + co_base + o) * cin + c];
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) clear_zero_row(a);
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.npad) return;
  const int orow = a.out_row ? a.out_row[t] : (t < a.n ? (int)t : -1);
  if (orow < 0) return;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  for (int k = 0; k < a.kvol; ++k) {
    const int id = a.nbr ? a.nbr[(int64_t)k * a.npad + t] : (int)t;
    if (id < 0) continue;
    const float* x = a.in0 + (int64_t)id * cin;
    for (int c = 0; c < cin; ++c) {
      const float xv = x[c];
      const float4* w4 = reinterpret_cast<const float4*>(wsm + (k * cin + c) * CO);
#pragma unroll
      for (int o4 = 0; o4 < CO / 4; ++o4) {
        const float4 w = w4[o4];
        acc[o4 * 4 + 0] = fmaf(xv, w.x, acc[o4 * 4 + 0]);
        acc[o4 * 4 + 1] = fmaf(xv, w.y, acc[o4 * 4 + 1]);
        acc[o4 * 4 + 2] = fmaf(xv, w.z, acc[o4 * 4 + 2]);
        acc[o4 * 4 + 3] = fmaf(xv, w.w, acc[o4 * 4 + 3]);
      }
    }
  }
  float v[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    const int col = co_base + o;
    float y = acc[o] * (a.scale ? a.scale[col] : 1.f) + (a.shift ? a.shift[col] : 0.f);
    if (a.residual) y += a.residual[(int64_t)orow * a.cout + col];
    if (a.relu) y = fmaxf(y, 0.f);
    v[o] = y;
  }
  if (a.out) {
    float* o = a.out + (int64_t)orow * a.ldo + co_base;
    if ((a.ldo & 3) == 0) {
#pragma unroll
      for (int o4 = 0; o4 < CO / 4; ++o4)
        reinterpret_cast<float4*>(o)[o4] = make_float4(v[o4 * 4], v[o4 * 4 + 1], v[o4 * 4 + 2], v[o4 * 4 + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < CO; ++i) o[i] = v[i];
    }
  }
  if (a.out_split) {
    const int tstride = 2 << split_group_shift(a.terms);
#pragma unroll
    for (int o8 = 0; o8 < CO / 8; ++o8) {
      float v8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v8[i] = v[o8 * 8 + i];
      split_store_n<8>(split_ptr(a.out_split, orow, co_base + o8 * 8, a.cout, a.terms), tstride, a.terms, v8,
                       a.f16_status);
    }
  }
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
  if (blockIdx.x == 0) clear_zero_row(a);
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
  if (a.out) a.out[(int64_t)orow * a.ldo + col] = v;
  if (a.out_split) store_split(a.out_split, orow, col, a.cout, a.terms, v, a.f16_status);
}

template <int WM, int WN, int NT, int T>
static int launch_split(const ConvArgs& a, int nz, hipStream_t stream) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  const size_t lds = (size_t)2 * T * (BM + BN) * kRowB + BM * sizeof(int);
  static LdsGrant grant;
  if (int rc = grant.ensure(reinterpret_cast<const void*>(&conv_igemm_split_kernel<WM, WN, NT, T>), lds)) return rc;
  dim3 grid((unsigned)(a.npad / BM), (unsigned)(a.cout / BN), (unsigned)nz);
  hipLaunchKernelGGL((conv_igemm_split_kernel<WM, WN, NT, T>), grid, dim3(NTHR), lds, stream, a);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

template <int WM, int WN, int NT, int T, int NBUF, bool AF = false, bool GMAP = false>
static int launch_dma_n(const ConvArgs& a, int nz, hipStream_t stream) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
  constexpr int A_ROWB = AF && T == 3 ? DmaGeomF32::ROWB : DmaGeom<T>::ROWB;
  constexpr size_t ring = (size_t)NBUF * (BM * A_ROWB + BN * DmaGeom<T>::ROWB), tile = (size_t)BM * BN * sizeof(float);
  const size_t lds = (ring > tile ? ring : tile) + (size_t)((GMAP ? 2 : a.kvol) + 1) * BM * sizeof(int);
  static LdsGrant grant;
  if (int rc = grant.ensure(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<WM, WN, NT, T, NBUF, AF, GMAP>), lds)) return rc;
  dim3 grid((unsigned)(a.npad / BM), (unsigned)(a.cout / BN), (unsigned)nz);
  hipLaunchKernelGGL((conv_igemm_dma_kernel<WM, WN, NT, T, NBUF, AF, GMAP>), grid, dim3(NTHR), lds, stream, a);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

// Per-device constants of the library: zero bytes for the gathers of missing neighbours when the activations are
// plain f32 rows (never written), and the f16 status word (set to 1 by any kernel that writes a value beyond the
// f16 range into the f16 pair format; read and cleared through m3d_f16_status).
static __device__ __align__(256) char conv_zero_bytes[256];
static __device__ int conv_f16_status_word;

static int device_symbols(const char** zero, int** status) {
  static std::mutex mu;
  static const char* zptr[64] = {};
  static int* sptr[64] = {};
  int dev = 0;
  M3D_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) {
    set_error("conv: device index %d out of range", dev);
    return M3D_EINVAL;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (zptr[dev] == nullptr) {
    void* p = nullptr;
    M3D_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(conv_zero_bytes)));
    void* q = nullptr;
    M3D_HIP(hipGetSymbolAddress(&q, HIP_SYMBOL(conv_f16_status_word)));
    zptr[dev] = static_cast<const char*>(p);
    sptr[dev] = static_cast<int*>(q);
  }
  if (zero) *zero = zptr[dev];
  if (status) *status = sptr[dev];
  return M3D_OK;
}

int f16_status_ptr(int** out) { return device_symbols(nullptr, out); }

// Ring depth: two buffers everywhere.  A third (which the 96-byte rows of the three-term format leave room for at two
// workgroups per CU) measured 4 % slower on the largest level: the stage loop is bound by the issue of its DMA
// instructions (about 160 cycles each beside the matrix work of the other waves), not by their latency.
// Experiment hook: M3D_NBUF.
template <int WM, int WN, int NT, int T>
static int launch_dma(const ConvArgs& a, int nz, hipStream_t stream) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN;
  static const int want = [] {
    const char* e = getenv("M3D_NBUF");
    return e ? atoi(e) : 0;
  }();
  int nbuf = want ? want : 2;
  const size_t per_buf = (size_t)(BM + BN) * DmaGeom<T>::ROWB, idx = (size_t)(a.kvol + 1) * BM * sizeof(int);
  while (nbuf > 2 && nbuf * per_buf + idx > 160 * 1024) --nbuf;
  if (nbuf >= 4) return launch_dma_n<WM, WN, NT, T, 4>(a, nz, stream);
  if (nbuf == 3) return launch_dma_n<WM, WN, NT, T, 3>(a, nz, stream);
  if constexpr (T == 3 && BM == 128 && BN == 96) {
    // Largest level only: with the kernel map out of LDS three workgroups fit a CU.  Launches of fewer tiles keep
    // two (a third workgroup per CU leaves part of the chip without work in their last round).
    static const int64_t gmap_tiles = getenv("M3D_GMAP_TILES") ? atoll(getenv("M3D_GMAP_TILES")) : 1024;
    if (a.nbr && nz == 1 && a.npad / BM >= gmap_tiles) return launch_dma_n<WM, WN, NT, T, 2, false, true>(a, nz, stream);
  }
  return launch_dma_n<WM, WN, NT, T, 2>(a, nz, stream);
}

template <int WM, int WN, int NT>
static int launch_af3(const ConvArgs& a, int nz, hipStream_t stream) {
  if constexpr (WM == 4 && WN == 1 && NT == 3) {
    static const int64_t gmap_tiles = getenv("M3D_GMAP_TILES") ? atoll(getenv("M3D_GMAP_TILES")) : 1024;
    if (a.nbr && nz == 1 && a.npad / 128 >= gmap_tiles) return launch_dma_n<WM, WN, NT, 3, 2, true, true>(a, nz, stream);
  }
  return launch_dma_n<WM, WN, NT, 3, 2, true>(a, nz, stream);
}

// Tile configuration.  Column tile: the widest of 128/96/64/32 dividing cout.  Row tile and offset split:
// short reductions (fewer than three 32-channel slabs per offset) start from 64-row tiles, longer ones from
// 128-row tiles (the weight stream is amortised over more rows); the kernel offsets are split nz ways until
// about two workgroups per CU exist, and the row tile only shrinks when even a 9-way split cannot get there.
struct ConvPlan {
  int wm, wn, nt, nz;
};

static ConvPlan plan_for_tile(int bm, int bn, int nz) {
  if (bm == 128) return ConvPlan{4, 1, bn / 32, nz};
  if (bm == 64) {
    if (bn == 128) return ConvPlan{2, 2, 2, nz};
    if (bn == 64) return ConvPlan{2, 2, 1, nz};
    return ConvPlan{2, 1, bn / 32, nz};  // 96 or 32 columns
  }
  return ConvPlan{1, bn / 32, 1, nz};
}

static ConvPlan plan_conv(int64_t npad, int cout, int kvol, int cin) {
  int bn;
  if (cout % 128 == 0) bn = 128;
  else if (cout % 96 == 0) bn = 96;
  else if (cout % 64 == 0) bn = 64;
  else bn = 32;
  const int64_t ncol = cout / bn;
  const int64_t target = 480, enough = 400;
  static const int nz_cap = getenv("M3D_MAX_NZ") ? atoi(getenv("M3D_MAX_NZ")) : 9;
  const int max_nz = kvol == 27 ? nz_cap : 1;  // strided and inverse maps have about one active offset per output tile
  for (int bm = cin >= 96 ? 128 : 64; bm >= 32; bm /= 2) {
    const int64_t blocks = (npad / bm) * ncol;
    if (blocks >= target) return plan_for_tile(bm, bn, 1);
    int64_t nz = (target + blocks / 2) / blocks;  // nearest split count
    if (nz > max_nz) nz = max_nz;
    if (blocks * nz >= enough || bm == 32) return plan_for_tile(bm, bn, (int)nz);
  }
  return plan_for_tile(32, bn, 1);
}

}  // namespace m3d

using namespace m3d;

extern "C" size_t m3d_conv_workspace_bytes(int64_t n, int cout, int k) {
  const int64_t npad = m3d_schedule_npad(n);
  if (cout <= 0 || cout % 32 != 0) return 256;  // served by the direct kernel, which needs no scratch
  const int nz_short = plan_conv(npad, cout, k, 32).nz, nz_long = plan_conv(npad, cout, k, 1 << 20).nz;
  int nz = nz_short > nz_long ? nz_short : nz_long;  // the plan also depends on the input channels
  if (getenv("M3D_PLAN")) nz = k;
  if (nz <= 1) return 256;
  return align_up((size_t)nz * npad * cout * sizeof(float));
}

// mode: 0 exact f32 without the matrix cores (channel counts that are not multiples of 32); 11 exact f32 by LDS-DMA;
// 5 plain bf16 products with the f32 activations rounded on their way to LDS;
// 3/4/6/9 LDS-DMA on sf32 inputs (T = 2/3/1/f16 pair); 7/8/10 LDS-DMA on f32 inputs, terms formed after the fragment
// read (T = 3/2/f16 pair)
static int run_conv(ConvArgs& a, int mode, bool mfma_ok, void* d_ws, size_t ws_bytes, hipStream_t stream,
                    const char* who) {
  a.slab = nullptr;
  if (!mfma_ok) {
    if (mode == 3 || mode == 4 || mode == 6 || mode >= 7) {
      set_error("%s: channel counts must be multiples of 32", who);
      return M3D_EINVAL;
    }
    if (a.c1 == 0 && a.c0 <= 8 && a.cout % 32 == 0) {  // the stem: rows x 32-channel groups
      const size_t lds = (size_t)a.kvol * a.c0 * 32 * sizeof(float);
      hipLaunchKernelGGL(conv_rowwise_f32_kernel<32>, dim3((unsigned)cdiv(a.npad, 256), (unsigned)(a.cout / 32)),
                         dim3(256), lds, stream, a);
      M3D_LAUNCH_CHECK();
      return M3D_OK;
    }
    const int64_t total = a.npad * a.cout;
    const size_t wbytes = (size_t)a.kvol * a.cout * (a.c0 + a.c1) * sizeof(float);
    const size_t lds = wbytes <= 60 * 1024 ? wbytes : 0;
    hipLaunchKernelGGL(conv_direct_f32_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), lds, stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
  }
  ConvPlan p = plan_conv(a.npad, a.cout, a.kvol, a.c0 + a.c1);
  if (const char* e = getenv("M3D_PLAN")) {  // experiment hook: "wm,wn,nt,nz"
    ConvPlan q{};
    if (sscanf(e, "%d,%d,%d,%d", &q.wm, &q.wn, &q.nt, &q.nz) == 4 && q.wm > 0 && q.wn > 0 && q.nt > 0 && 32 * q.nt * q.wn <= a.cout &&
        a.cout % (32 * q.nt * q.wn) == 0 && a.npad % (32 * q.wm) == 0 && q.nz >= 1 && q.nz <= a.kvol)
      p = q;
  }
  const bool dma = mode == 3 || mode == 4 || mode == 6 || mode >= 7;
  if (dma && p.wm == 1 && p.wn == 3) p = ConvPlan{1, 1, 3, p.nz};  // equal DMA count per wave
  if (p.nz > 1) {
    const size_t need = (size_t)p.nz * a.npad * a.cout * sizeof(float);
    if (d_ws == nullptr || ws_bytes < need) {
      set_error("%s: workspace too small (%zu given, %zu needed)", who, ws_bytes, need);
      return M3D_ENOSPC;
    }
    a.slab = static_cast<float*>(d_ws);
  }
  int rc = M3D_EINVAL;
#define M3D_CASE(WM_, WN_, NT_)                                                     \
  if (p.wm == WM_ && p.wn == WN_ && p.nt == NT_) {                                  \
    if (mode == 5) rc = launch_split<WM_, WN_, NT_, 1>(a, p.nz, stream);            \
  }
#define M3D_CASE_DMA(WM_, WN_, NT_)                                                 \
  if (p.wm == WM_ && p.wn == WN_ && p.nt == NT_) {                                  \
    if (mode == 3) rc = launch_dma<WM_, WN_, NT_, 2>(a, p.nz, stream);              \
    else if (mode == 4) rc = launch_dma<WM_, WN_, NT_, 3>(a, p.nz, stream);         \
    else if (mode == 6) rc = launch_dma<WM_, WN_, NT_, 1>(a, p.nz, stream);         \
    else if (mode == 7) rc = launch_af3<WM_, WN_, NT_>(a, p.nz, stream); \
    else if (mode == 8) rc = launch_dma_n<WM_, WN_, NT_, 2, 2, true>(a, p.nz, stream); \
    else if (mode == 9) rc = launch_dma_n<WM_, WN_, NT_, kFmtF16, 2, false>(a, p.nz, stream); \
    else if (mode == 10) rc = launch_dma_n<WM_, WN_, NT_, kFmtF16, 2, true>(a, p.nz, stream); \
    else if (mode == 11) rc = launch_dma_n<WM_, WN_, NT_, kFmtF32, 2, true>(a, p.nz, stream); \
  }
  if (!dma) {
    M3D_CASE(4, 1, 1) M3D_CASE(4, 1, 2) M3D_CASE(4, 1, 3) M3D_CASE(4, 1, 4)
    M3D_CASE(2, 2, 1) M3D_CASE(2, 2, 2) M3D_CASE(2, 1, 3) M3D_CASE(2, 1, 1)
    M3D_CASE(1, 4, 1) M3D_CASE(1, 3, 1) M3D_CASE(1, 2, 1) M3D_CASE(1, 1, 1)
  } else {
    M3D_CASE_DMA(4, 1, 1) M3D_CASE_DMA(4, 1, 2) M3D_CASE_DMA(4, 1, 3) M3D_CASE_DMA(4, 1, 4)
    M3D_CASE_DMA(2, 2, 1) M3D_CASE_DMA(2, 2, 2) M3D_CASE_DMA(2, 1, 3) M3D_CASE_DMA(2, 1, 1)
    M3D_CASE_DMA(1, 4, 1) M3D_CASE_DMA(1, 1, 3) M3D_CASE_DMA(1, 2, 1) M3D_CASE_DMA(1, 1, 1)
    M3D_CASE_DMA(4, 2, 4)
  }
#undef M3D_CASE
#undef M3D_CASE_DMA
  if (rc != M3D_OK) {
    if (rc == M3D_EINVAL) set_error("%s: no kernel for plan (%d,%d,%d)", who, p.wm, p.wn, p.nt);
    return rc;
  }
  if (p.nz > 1) {
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)cdiv(a.npad * a.cout, 256)), dim3(256), 0, stream, a,
                       p.nz);
    M3D_LAUNCH_CHECK();
  }
  return M3D_OK;
}

static int check_schedule(const int32_t* d_nbr, const int32_t* d_out_row, const uint32_t* d_tile_mask, int k,
                          int64_t n, int64_t* npad, const char* who) {
  if (d_nbr == nullptr) {
    if (!(k == 1 && d_out_row == nullptr && d_tile_mask == nullptr)) {
      set_error("%s: dense mode needs k == 1 and no schedule", who);
      return M3D_EINVAL;
    }
    *npad = m3d_schedule_npad(n);
  } else {
    if (!(d_out_row && d_tile_mask)) {
      set_error("%s: schedule pointers missing", who);
      return M3D_EINVAL;
    }
    if (!(*npad >= n && *npad % 128 == 0)) {
      set_error("%s: npad=%lld must be a multiple of 128 >= n", who, (long long)*npad);
      return M3D_EINVAL;
    }
  }
  if (*npad >= (int64_t)1 << 31) {
    set_error("%s: too many rows", who);
    return M3D_EINVAL;
  }
  return M3D_OK;
}

extern "C" int m3d_conv_fwd_f32(const float* d_in0, int c0, int64_t rows0, const float* d_in1, int c1, int64_t rows1,
                                const float* d_weight, int k, int cout, const int32_t* d_nbr,
                                const int32_t* d_out_row, const uint32_t* d_tile_mask, int64_t n,
                                int64_t npad, const int32_t* d_tile_order, const float* d_scale, const float* d_shift,
                                const float* d_residual, int relu, float* d_out, int64_t ldo,
                                void* d_out_split, int out_terms, int precision, void* d_ws,
                                size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const char* who = "m3d_conv_fwd_f32";
  M3D_CHECK_ARG(n >= 0, "%s: n < 0", who);
  if (n == 0) return M3D_OK;
  M3D_CHECK_ARG(d_in0 && d_weight && (d_out || d_out_split), "%s: null pointer", who);
  M3D_CHECK_ARG(c0 > 0 && c1 >= 0 && cout > 0 && k >= 1 && k <= 27, "%s: bad channel/kernel sizes", who);
  M3D_CHECK_ARG((c1 == 0) == (d_in1 == nullptr), "%s: in1/c1 mismatch", who);
  M3D_CHECK_ARG(d_out == nullptr || ldo >= cout, "%s: ldo < cout", who);
  M3D_CHECK_ARG(precision >= 0 && precision <= M3D_PREC_F16X3, "%s: unknown precision %d", who, precision);
  M3D_CHECK_ARG(d_out_split == nullptr || (out_terms >= 1 && out_terms <= kFmtF16 && cout % 32 == 0),
                "%s: a split output needs out_terms in {1,2,3,4} and cout %% 32 == 0", who);
  if (int rc = check_schedule(d_nbr, d_out_row, d_tile_mask, k, n, &npad, who)) return rc;
  ConvArgs a = {};
  a.in0 = d_in0; a.in1 = d_in1; a.weight = d_weight; a.nbr = d_nbr; a.out_row = d_out_row;
  a.tile_mask = d_tile_mask; a.scale = d_scale; a.shift = d_shift; a.residual = d_residual;
  a.out = d_out; a.n = n; a.npad = npad; a.ldo = ldo; a.c0 = c0; a.c1 = c1;
  a.cout = cout; a.kvol = k; a.relu = relu;
  a.out_split = static_cast<char*>(d_out_split); a.terms = out_terms;
  a.tile_order = d_tile_order;
  const bool mfma_ok = (c0 % 32 == 0) && (c1 % 32 == 0) && (cout % 32 == 0);
  // a weight packed for a bf16 precision is bf16 only where the whole layer runs on the matrix cores: two input
  // slabs whose sum is a multiple of 32 but which are not multiples themselves have no kernel
  M3D_CHECK_ARG(precision == M3D_PREC_F32 || mfma_ok || (c0 + c1) % 32 != 0 || cout % 32 != 0,
                "%s: with a bf16 precision both input slabs (%d, %d) must be multiples of 32 channels", who, c0, c1);
  if (int rc = device_symbols(&a.zbuf, &a.f16_status)) return rc;
  int mode = precision == M3D_PREC_BF16 ? 5 : precision;
  // exact f32 on the matrix cores: rows and weights go to LDS by LDS-DMA with 64-bit addresses (the number of input
  // rows is not an argument here, so no buffer descriptor can bound them)
  if (precision == M3D_PREC_F32 && mfma_ok) {
    mode = 11;
    a.s0 = reinterpret_cast<const char*>(d_in0);
    a.s1 = reinterpret_cast<const char*>(d_in1);
  }
  if (precision == M3D_PREC_F16X3) {
    // the f16 pair has no kernel outside the matrix-core path: small layers (the stem) run in exact f32
    mode = mfma_ok ? 10 : 0;
    a.s0 = reinterpret_cast<const char*>(d_in0);
    a.s1 = reinterpret_cast<const char*>(d_in1);
  }
  // split products: the f32 rows go to LDS as they are and the terms are formed after the fragment read
  if ((mode == 1 || mode == 2) && mfma_ok) {
    mode = mode == 2 ? 7 : 8;
    a.s0 = reinterpret_cast<const char*>(d_in0);
    a.s1 = reinterpret_cast<const char*>(d_in1);
  }
  if (mode == 7 || mode == 8 || mode == 10) {
    // the kernel addresses every operand with 32-bit byte offsets inside one buffer descriptor
    M3D_CHECK_ARG(rows0 > 0 && (d_in1 == nullptr || rows1 > 0), "%s: rows0 / rows1 are required at this precision", who);
    M3D_CHECK_ARG(d_nbr != nullptr || rows0 >= n, "%s: dense mode needs rows0 >= n", who);
    const int64_t cmax = c0 > c1 ? c0 : c1, rmax = rows0 > rows1 ? rows0 : rows1;
    M3D_CHECK_ARG(rmax * cmax * 4 <= (int64_t)kDmaBufferBytes && (int64_t)k * cout * (c0 + c1) * 6 <= (int64_t)kDmaBufferBytes,
                  "%s: an operand of more than 4 GB (%lld rows x %lld channels) is beyond the kernel's buffer addressing",
                  who, (long long)rmax, (long long)cmax);
  }
  return run_conv(a, mode, mfma_ok, d_ws, ws_bytes, stream, who);
}

extern "C" int m3d_conv_fwd_sf32(const void* d_in0, int c0, int64_t rows0, const void* d_in1, int c1,
                                 int64_t rows1, const void* d_weight, int k, int cout,
                                 const int32_t* d_nbr, const int32_t* d_out_row,
                                 const uint32_t* d_tile_mask, int64_t n, int64_t npad, const int32_t* d_tile_order,
                                 const float* d_scale, const float* d_shift, const float* d_residual,
                                 int relu, float* d_out, int64_t ldo, void* d_out_split, int terms,
                                 void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const char* who = "m3d_conv_fwd_sf32";
  M3D_CHECK_ARG(n >= 0, "%s: n < 0", who);
  if (n == 0) return M3D_OK;
  M3D_CHECK_ARG(d_in0 && d_weight && (d_out || d_out_split), "%s: null pointer", who);
  M3D_CHECK_ARG(terms >= 1 && terms <= kFmtF16, "%s: terms must be 1, 2, 3 or 4 (the f16 pair)", who);
  M3D_CHECK_ARG(c0 > 0 && c1 >= 0 && cout > 0 && k >= 1 && k <= 27, "%s: bad channel/kernel sizes", who);
  M3D_CHECK_ARG(c0 % 32 == 0 && c1 % 32 == 0 && cout % 32 == 0, "%s: channel counts must be multiples of 32", who);
  M3D_CHECK_ARG((c1 == 0) == (d_in1 == nullptr), "%s: in1/c1 mismatch", who);
  M3D_CHECK_ARG(rows0 >= 0 && rows1 >= 0, "%s: negative row count", who);
  {  // the kernel addresses every operand with 32-bit byte offsets inside one buffer descriptor
    const int64_t bpc = 2 * split_nterms(terms);  // bytes per channel of a split row
    const int64_t cmax = c0 > c1 ? c0 : c1, rmax = rows0 > rows1 ? rows0 : rows1;
    M3D_CHECK_ARG((rmax + 1) * cmax * bpc <= (int64_t)kDmaBufferBytes && (int64_t)k * cout * (c0 + c1) * bpc <= (int64_t)kDmaBufferBytes,
                  "%s: an operand of more than 4 GB (%lld rows x %lld channels x %lld bytes) is beyond the kernel's buffer addressing",
                  who, (long long)rmax, (long long)cmax, (long long)bpc);
  }
  M3D_CHECK_ARG(d_out == nullptr || ldo >= cout, "%s: ldo < cout", who);
  if (int rc = check_schedule(d_nbr, d_out_row, d_tile_mask, k, n, &npad, who)) return rc;
  M3D_CHECK_ARG(d_nbr != nullptr || rows0 >= n, "%s: dense mode needs rows0 >= n", who);
  ConvArgs a = {};
  a.s0 = static_cast<const char*>(d_in0); a.s1 = static_cast<const char*>(d_in1);
  a.zero0 = rows0; a.zero1 = rows1;
  a.weight = static_cast<const float*>(d_weight); a.nbr = d_nbr; a.out_row = d_out_row;
  a.tile_mask = d_tile_mask; a.scale = d_scale; a.shift = d_shift; a.residual = d_residual;
  a.out = d_out; a.n = n; a.npad = npad; a.ldo = ldo; a.c0 = c0; a.c1 = c1;
  a.cout = cout; a.kvol = k; a.relu = relu;
  a.out_split = static_cast<char*>(d_out_split); a.terms = terms;
  a.tile_order = d_tile_order;
  if (int rc = device_symbols(&a.zbuf, &a.f16_status)) return rc;
  return run_conv(a, terms == 2 ? 3 : (terms == 3 ? 4 : (terms == kFmtF16 ? 9 : 6)), true, d_ws, ws_bytes, stream, who);
}

extern "C" int m3d_f16_status(int* h_status, int clear, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int* status = nullptr;
  if (int rc = device_symbols(nullptr, &status)) return rc;
  if (h_status) {
    M3D_HIP(hipMemcpyAsync(h_status, status, sizeof(int), hipMemcpyDeviceToHost, stream));
    M3D_HIP(hipStreamSynchronize(stream));
  }
  if (clear) M3D_HIP(hipMemsetAsync(status, 0, sizeof(int), stream));
  return M3D_OK;
}

extern "C" size_t m3d_sf32_bytes(int64_t n, int c, int terms) {
  return (size_t)(n + 1) * (size_t)c * 2 * (size_t)split_nterms(terms);
}

extern "C" int m3d_split_f32(const float* d_in, int64_t n, int c, int terms, void* d_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(d_in && d_out && n >= 0 && c > 0 && c % 32 == 0 && terms >= 1 && terms <= kFmtF16,
                "m3d_split_f32: bad arguments");
  int* status = nullptr;
  if (int rc = device_symbols(nullptr, &status)) return rc;
  const int64_t total = (n + 1) * c;
  hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, d_in, n, c, terms,
                     static_cast<char*>(d_out), status);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" size_t m3d_conv_packed_weight_bytes(int cout, int k, int cin, int precision) {
  const bool mfma_ok = cin % 32 == 0 && cout % 32 == 0;
  const size_t elems = (size_t)cout * k * cin;
  if (precision == M3D_PREC_F32 || !mfma_ok) return elems * sizeof(float);
  return elems * 2 * (precision == M3D_PREC_BF16X3 || precision == M3D_PREC_F16X3 ? 2 : (precision == M3D_PREC_BF16X6 ? 3 : 1));
}

static int pack_weight_launch(const char* who, const float* d_weight_ref, int cout, int k, int cin, int precision,
                              int transposed, int flip, void* d_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(d_weight_ref && d_out && cout > 0 && k > 0 && cin > 0, "%s: bad arguments", who);
  M3D_CHECK_ARG(precision >= 0 && precision <= M3D_PREC_F16X3, "%s: unknown precision %d", who, precision);
  const bool mfma_ok = cin % 32 == 0 && cout % 32 == 0;
  const int terms = (precision == M3D_PREC_F32 || !mfma_ok) ? 0
                    : (precision == M3D_PREC_BF16X3 ? 2 : (precision == M3D_PREC_BF16X6 ? 3
                    : (precision == M3D_PREC_F16X3 ? kFmtF16 : 1)));
  int* status = nullptr;
  if (int rc = device_symbols(nullptr, &status)) return rc;
  const int64_t total = (int64_t)cout * k * cin;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, d_weight_ref,
                     cout, k, cin, terms, transposed, flip, d_out, status);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_conv_pack_weight(const float* d_weight_ref, int cout, int k, int cin, int precision,
                                    void* d_out, void* stream) {
  return pack_weight_launch("m3d_conv_pack_weight", d_weight_ref, cout, k, cin, precision, 0, 0, d_out, stream);
}

extern "C" int m3d_conv_pack_weight_transposed(const float* d_weight_ref, int cout, int k, int cin, int flip,
                                               int precision, void* d_out, void* stream) {
  // the packed weight has cin output and cout input channels
  return pack_weight_launch("m3d_conv_pack_weight_transposed", d_weight_ref, cin, k, cout, precision, 1, flip != 0,
                            d_out, stream);
}

#ifdef M3D_STAMP
extern "C" int m3d_debug_stamps(unsigned long long* out8, int reset) {
  unsigned long long z[8] = {};
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(m3d::m3d_stamp_sums), sizeof(z)) != hipSuccess) return -3;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(m3d::m3d_stamp_sums), z, sizeof(z)) != hipSuccess) return -3;
  return 0;
}
#endif
