// Halo-tiled sparse convolution for gfx950 (split-bf16 products on sf32 operands, T = 2).
//
// The LDS-DMA kernel of conv.hip gathers a fresh copy of the input rows for every kernel offset.  For a
// spatially compact tile most of those rows are the same voxels seen through different offsets: a tile of
// 256 outputs of a 3x3x3 submanifold map references ~390 distinct inputs through ~2800 pairs.  Here a
// workgroup loads that union ("halo") ONCE per 32-channel slab into LDS and every kernel offset reads its
// A fragments from it by a per-row local index (ds_read addresses are per lane anyway), so that only the
// weights stream per (offset, slab).  The halo of every tile is found once per kernel map by
// m3d_build_halo and reused by all convolutions on that map.
//
// Workgroup: 8 waves x 32 rows = 256 output rows x BN = 32*NT output channels, one workgroup per CU.
// LDS: halo rows [CAP + 1][128 B] (the last row is zero), a ring of NW weight tiles [BN][128 B], the local
// index table [k][256] u16, the output rows of the tile.
#include "conv_shared.h"

#include <stdlib.h>

namespace m3d {

constexpr int kHaloCap = 640;    // halo rows a tile may have; larger halos take the per-offset path
constexpr int kHaloRows = 256;   // output rows per tile
constexpr int kHaloHash = 2048;  // LDS hash slots of the builder (>= 2 * kHaloCap, power of two)
constexpr uint16_t kHaloNone = 0xFFFF;

// ------------------------------------------------------------------------------------------ builder
// One workgroup per tile.  Distinct neighbour ids are collected in an LDS hash set, ordered by `in_pos`
// (rows that are close in the schedule get close local indices, which keeps the fragment reads of 16
// consecutive output rows on distinct LDS banks) and numbered; loc[k][r] is the number of nbr[k][r].
__global__ __launch_bounds__(256) void halo_build_kernel(const int32_t* __restrict__ nbr, int kvol, int64_t npad,
                                                         const int32_t* __restrict__ in_pos,
                                                         int32_t* __restrict__ halo_cnt,
                                                         int32_t* __restrict__ halo_rows,
                                                         uint16_t* __restrict__ halo_loc) {
  __shared__ int hkey[kHaloHash];
  __shared__ int hval[kHaloHash];
  __shared__ unsigned long long sortbuf[1024];
  __shared__ int cnt_s;
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x, base = tile * kHaloRows;
  for (int i = tid; i < kHaloHash; i += 256) hkey[i] = -1;
  if (tid == 0) cnt_s = 0;
  __syncthreads();
  // 1. distinct ids.  New ids stop being admitted once more than kHaloCap exist, so the table (2048 slots,
  //    at most kHaloCap + one in-flight insert per thread occupied) never fills and every probe ends.
  for (int e = tid; e < kvol * kHaloRows; e += 256) {
    const int k = e / kHaloRows, r = e % kHaloRows;
    const int id = nbr[(int64_t)k * npad + base + r];
    if (id < 0) continue;
    uint32_t s = ((uint32_t)id * 2654435761u) >> 21;  // 11 bits
    while (true) {
      const int cur = hkey[s];
      if (cur == id) break;
      if (cur == -1) {
        if (*(volatile int*)&cnt_s > kHaloCap) break;
        const int old = atomicCAS(&hkey[s], -1, id);
        if (old == -1) {
          atomicAdd(&cnt_s, 1);
          break;
        }
        if (old == id) break;
      }
      s = (s + 1) & (kHaloHash - 1);
    }
  }
  __syncthreads();
  const int cnt = cnt_s;
  if (cnt > kHaloCap) {
    if (tid == 0) halo_cnt[tile] = -1;
    return;
  }
  __syncthreads();
  if (tid == 0) cnt_s = 0;
  __syncthreads();
  // 2. list the occupied slots with their ordering key
  for (int i = tid; i < kHaloHash; i += 256)
    if (hkey[i] >= 0) {
      const int j = atomicAdd(&cnt_s, 1);
      sortbuf[j] = ((unsigned long long)(uint32_t)(in_pos ? in_pos[hkey[i]] : hkey[i]) << 11) | (unsigned)i;
    }
  __syncthreads();
  for (int i = cnt + tid; i < 1024; i += 256) sortbuf[i] = ~0ull;
  __syncthreads();
  // 3. bitonic sort of 1024 keys (position, slot)
  for (int k2 = 2; k2 <= 1024; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < 1024; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = sortbuf[i], b = sortbuf[ixj];
          const bool up = (i & k2) == 0;
          if ((a > b) == up) {
            sortbuf[i] = b;
            sortbuf[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  // 4. number the halo rows
  for (int j = tid; j < cnt; j += 256) {
    const int slot = (int)(sortbuf[j] & (kHaloHash - 1));
    hval[slot] = j;
    halo_rows[tile * kHaloCap + j] = hkey[slot];
  }
  if (tid == 0) halo_cnt[tile] = cnt;
  __syncthreads();
  // 5. local index of every pair
  for (int e = tid; e < kvol * kHaloRows; e += 256) {
    const int k = e / kHaloRows, r = e % kHaloRows;
    const int id = nbr[(int64_t)k * npad + base + r];
    uint16_t loc = kHaloNone;
    if (id >= 0) {
      uint32_t s = ((uint32_t)id * 2654435761u) >> 21;
      while (hkey[s] != id) s = (s + 1) & (kHaloHash - 1);
      loc = (uint16_t)hval[s];
    }
    halo_loc[(tile * kvol + k) * kHaloRows + r] = loc;
  }
}

// Tile order of the convolution grid: tiles on the slower per-offset path first, so that they overlap with the
// rest instead of forming the tail.  One workgroup; stable inside both groups.
__global__ __launch_bounds__(1024) void halo_order_kernel(const int32_t* __restrict__ cnt, int ntile,
                                                          int32_t* __restrict__ order) {
  __shared__ int wsum[16];
  __shared__ int base_s[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 2) base_s[tid] = 0;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 places the overflow tiles, pass 1 the others after them
    for (int t0 = 0; t0 < ntile; t0 += 1024) {
      const int t = t0 + tid;
      const int flag = t < ntile && ((cnt[t] < 0) == (pass == 0)) ? 1 : 0;
      int inc = flag;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(inc, d, 64);
        if (lane >= d) inc += v;
      }
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      int before = base_s[0];
      for (int w = 0; w < wave; ++w) before += wsum[w];
      if (flag) order[before + inc - 1] = t;
      __syncthreads();
      if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        base_s[0] += tot;
      }
      __syncthreads();
    }
  }
}

__global__ void invert_rows_kernel(const int32_t* __restrict__ out_row, int64_t npad, int64_t n_in,
                                   int32_t* __restrict__ pos) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npad) return;
  const int r = out_row[t];
  if (r >= 0 && r < n_in) pos[r] = (int32_t)t;
}

// ------------------------------------------------------------------------------------------ kernel
__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

struct HaloArgs {
  const int32_t* cnt;
  const int32_t* rows;
  const uint16_t* loc;
  const int32_t* order;
};

template <int NT, int NW>
__device__ __forceinline__ void conv_halo_body(const ConvArgs& a, const HaloArgs& hl) {
  using G = DmaGeom<2>;
  constexpr int WV = 4, RW = 2, BM = kHaloRows, BN = 32 * NT, NTHR = 64 * WV;  // a wave owns RW x 32 rows
  constexpr int GRPB = G::PIECES * 16;                   // 128 B: one (row, 32-channel group) in HBM
  constexpr int A_BYTES = (kHaloCap + 1) * G::ROWB;      // halo rows + the zero row
  constexpr int W_BYTES = BN * G::ROWB;
  constexpr int A_V = kHaloCap * G::SLOTS / NTHR;        // DMA instructions per wave for a full halo
  constexpr int NWI = BN * G::SLOTS / 64 / WV;           // weight DMA instructions per wave and stage
  static_assert(WV * RW * 32 == BM, "tile rows");
  static_assert(kHaloCap * G::SLOTS % NTHR == 0, "halo capacity must be a whole number of DMA rounds");
  static_assert(BN * G::SLOTS % (64 * WV) == 0 && (NW - 3) * NWI <= 8, "weight DMAs per wave");
  extern __shared__ __align__(16) char ldsb[];
  char* Wb = ldsb + A_BYTES;
  uint16_t* loc_s = reinterpret_cast<uint16_t*>(Wb + NW * W_BYTES);  // [kvol][BM]
  int* orow_s = reinterpret_cast<int*>(ldsb + A_BYTES + NW * W_BYTES + 27 * BM * 2);
  int* hrow_s = orow_s + BM;  // [kHaloCap] input row of every halo row

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = hl.order[blockIdx.x];
  const int64_t tile_base = tile * BM;
  const int col_base = blockIdx.y * BN;
  const int nchunk = (a.c0 + a.c1) / kBK;
  const int g0 = a.c0 / kBK;
  const char* wsplit = reinterpret_cast<const char*>(a.weight);
  const uint32_t rowb0 = (uint32_t)g0 * GRPB, rowb1 = (uint32_t)(a.c1 / kBK) * GRPB;
  const uint32_t wk_stride = (uint32_t)(a.cout * nchunk) * GRPB;

  uint32_t bmask, wmask_unused;
  conv_masks<BM / 32>(a, tile_base, 0, bmask, wmask_unused);  // offsets present anywhere in the tile
  const int cnt = __builtin_amdgcn_readfirstlane(hl.cnt[tile]);
  const bool overflow = cnt < 0;

  for (int r = tid; r < BM; r += NTHR) orow_s[r] = a.out_row[tile_base + r];
  if (!overflow) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(hl.loc + tile * a.kvol * BM);
    uint32_t* dst = reinterpret_cast<uint32_t*>(loc_s);
    for (int e = tid; e < a.kvol * BM / 2; e += NTHR) dst[e] = src[e];
  }
  if (tid < G::ROWB / 4) reinterpret_cast<uint32_t*>(ldsb + kHaloCap * G::ROWB)[tid] = 0u;  // the zero row
  // halo rows: DMA slot e = v * NTHR + tid is (row e / 8 = v * 32 + tid / 8, physical slot tid % 8); the
  // swizzle of that row does not depend on v, so one source offset per thread serves every round
  for (int j = tid; j < kHaloCap; j += NTHR) hrow_s[j] = (!overflow && j < cnt) ? hl.rows[tile * kHaloCap + j] : -1;
  const uint32_t hp = (uint32_t)((tid % G::SLOTS) ^ G::swz(tid / G::SLOTS)) * 16u;
  // weight DMAs of this wave: instruction i = wave + WV * q covers pieces [64 i, 64 i + 64) of the tile
  uint32_t w_o[NWI];
#pragma unroll
  for (int q = 0; q < NWI; ++q) {
    const int e = (wave + WV * q) * 64 + lane, j = e / G::SLOTS;
    w_o[q] = (uint32_t)((col_base + j) * nchunk) * GRPB + (uint32_t)((e % G::SLOTS) ^ G::swz(j)) * 16u;
  }

  floatx16 acc[RW][NT];
#pragma unroll
  for (int rt = 0; rt < RW; ++rt)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[rt][i][j] = 0.f;

  const int nst = __popc(bmask);   // stages (active offsets) per channel slab
  const int gtot = nst * nchunk;   // weight tiles in all
  __syncthreads();

  // weight stream: global stage g = slab * nst + (ordinal of the offset), tile g lives in ring slot g % NW
  uint32_t irem = bmask;
  int ik = bmask ? __builtin_ctz(bmask) : 0, icc = 0, issued = 0;
  auto issue_w = [&]() {
    char* buf = Wb + (issued % NW) * W_BYTES;
    const char* wsrc = wsplit + ((uint64_t)ik * wk_stride + (uint32_t)icc * GRPB);
#pragma unroll
    for (int q = 0; q < NWI; ++q)
      __builtin_amdgcn_global_load_lds(wsrc + w_o[q], (lds_ptr_t)(buf + ((wave + WV * q) * 64) * 16), 16, 0, 0);
    ++issued;
    irem &= irem - 1;
    if (irem) {
      ik = __builtin_ctz(irem);
    } else {
      irem = bmask;
      ik = bmask ? __builtin_ctz(bmask) : 0;
      ++icc;
    }
  };
  auto issue_halo = [&](int cc) {
    const bool first = cc < g0;
    const char* src = (first ? a.s0 : a.s1) + (first ? cc : cc - g0) * GRPB;
    const uint32_t rowb = first ? rowb0 : rowb1;
    const int zero = (int)(first ? a.zero0 : a.zero1);
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      if ((v * NTHR + wave * 64) / G::SLOTS < cnt) {  // wave-uniform: this instruction's first row exists
        const int hr = hrow_s[v * (NTHR / G::SLOTS) + tid / G::SLOTS];
        const uint32_t row = (uint32_t)(hr < 0 ? zero : hr);
        __builtin_amdgcn_global_load_lds(src + ((uint64_t)row * rowb + hp), (lds_ptr_t)(ldsb + (v * NTHR + wave * 64) * 16),
                                         16, 0, 0);
      }
    }
  };
  // per-offset path of a tile whose halo does not fit: the 256 neighbours of one offset are gathered as rows
  // 0..255 of one of two row buffers in the halo region (like the LDS-DMA kernel of conv.hip does)
  constexpr int R_V = BM * G::SLOTS / NTHR;  // DMA instructions per wave for 256 rows
  static_assert(2 * BM * G::ROWB <= A_BYTES, "two row buffers must fit in the halo region");
  auto load_ids = [&](int (&ids)[R_V], int k) {
#pragma unroll
    for (int v = 0; v < R_V; ++v) ids[v] = a.nbr[(int64_t)k * a.npad + tile_base + (v * NTHR + tid) / G::SLOTS];
  };
  auto issue_rows = [&](const int (&ids)[R_V], int cc, int b) {
    const bool first = cc < g0;
    const char* src = (first ? a.s0 : a.s1) + (first ? cc : cc - g0) * GRPB;
    const uint32_t rowb = first ? rowb0 : rowb1;
    const int zero = (int)(first ? a.zero0 : a.zero1);
#pragma unroll
    for (int v = 0; v < R_V; ++v) {
      const uint32_t row = (uint32_t)(ids[v] < 0 ? zero : ids[v]);
      __builtin_amdgcn_global_load_lds(src + ((uint64_t)row * rowb + hp),
                                       (lds_ptr_t)(ldsb + b * (BM * G::ROWB) + (v * NTHR + wave * 64) * 16), 16, 0, 0);
    }
  };

  const int r32 = lane & 31, h = lane >> 5;
  const int arow0 = wave * (RW * 32) + r32;  // this lane's row of row tile rt is arow0 + 32 * rt
  // weight fragment of column tile nt: row nt * 32 + r32 of the tile; its swizzle does not depend on nt
  uint32_t w_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t) w_off[ks][t] = r32 * G::ROWB + (((t * 4 + ks * 2 + h) ^ G::swz(r32)) * 16);
  constexpr int kNtStride = 32 * G::ROWB;

  auto compute_rows = [&](int slot, int b) {  // per-offset path: this lane's inputs are rows arow0 + 32 rt of row buffer b
    const char* wbuf = Wb + slot * W_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[RW][2];
#pragma unroll
      for (int rt = 0; rt < RW; ++rt) {
        const int ar = arow0 + 32 * rt;
#pragma unroll
        for (int t = 0; t < 2; ++t)
          af[rt][t] = *reinterpret_cast<const bf16x8*>(ldsb + b * (BM * G::ROWB) + ar * G::ROWB +
                                                       (((t * 4 + ks * 2 + h) ^ G::swz(ar)) * 16));
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 bf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(wbuf + w_off[ks][t] + nt * kNtStride);
#pragma unroll
        for (int rt = 0; rt < RW; ++rt) {
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[1], acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], bf[0], acc[rt][nt], 0, 0, 0);
          acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[0], acc[rt][nt], 0, 0, 0);
        }
      }
    }
  };

  // Software pipeline of the main path, at half-stage granularity: a stage is two k-steps (ks = 0, 1) of 16
  // channels.  While the matrix pipe works on k-step 0 of stage t the fragments of its k-step 1 are fetched
  // from LDS, and during k-step 1 those of k-step 0 of stage t + 1, so that no LDS latency sits between the
  // barrier and the first MFMA.  Weight tile t + 1 therefore has to have landed at the barrier of stage t.
  // One wave per SIMD: a wave's 2 x NT accumulator tiles keep the matrix pipe busy on their own, and every
  // weight fragment read from LDS feeds two row tiles.
  bf16x8 fa[2][RW][2];  // [ks][row tile][term]
  bf16x8 fw[2][NT][2];  // [ks][nt][term]
  int lcur[RW];         // LDS rows of this lane's inputs for the current stage
#define M3D_HALO_LOAD(KS, WBUF)                                                                         \
  {                                                                                                     \
    _Pragma("unroll") for (int rt = 0; rt < RW; ++rt) {                                                 \
      const char* abase_ = ldsb + lcur[rt] * G::ROWB;                                                   \
      const int sw_ = G::swz(lcur[rt]);                                                                 \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                     \
        fa[KS][rt][t] = *reinterpret_cast<const bf16x8*>(abase_ + (((t * 4 + KS * 2 + h) ^ sw_) * 16)); \
    }                                                                                                   \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                   \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                     \
        fw[KS][nt][t] = *reinterpret_cast<const bf16x8*>((WBUF) + w_off[KS][t] + nt * kNtStride);       \
  }
// products of one k-step: (activation term, weight term) = (0,1), (1,0), (0,0), each over the RW x NT tiles
#define M3D_HALO_MMA_G(KS, TA, TW)                                                                      \
  {                                                                                                     \
    _Pragma("unroll") for (int rt = 0; rt < RW; ++rt)                                                   \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                 \
        acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[KS][rt][TA], fw[KS][nt][TW], acc[rt][nt], 0, 0, 0); \
  }
#define M3D_HALO_LOCALS(K)                                                                              \
  {                                                                                                     \
    _Pragma("unroll") for (int rt = 0; rt < RW; ++rt) {                                                 \
      const int l_ = (int)loc_s[(K) * BM + arow0 + 32 * rt];                                            \
      lcur[rt] = l_ == (int)kHaloNone ? kHaloCap : l_;                                                  \
    }                                                                                                   \
  }
  if (gtot > 0 && !overflow) {
#pragma unroll
    for (int u = 0; u < NW - 1; ++u)
      if (u < gtot) issue_w();
    int g = 0;
    for (int cc = 0; cc < nchunk; ++cc) {
      // every wave is done with the previous slab's halo; fetch this slab's and wait for everything in flight
      __builtin_amdgcn_s_barrier();
      issue_halo(cc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      uint32_t crem = bmask;
      int kc = __builtin_ctz(crem);
      M3D_HALO_LOCALS(kc)
      M3D_HALO_LOAD(0, Wb + (g % NW) * W_BYTES)
      // The stage body is straight-line code: every wave runs every active offset of the workgroup (skipping the
      // offsets absent from a wave's own rows would put the accumulators and fragments behind conditional
      // updates, which costs more in register copies than the skipped MFMAs save).  The last stage of a slab
      // prefetches fragments nobody uses (the next slab loads its own).
      for (int s = 0; s < nst; ++s, ++g) {
        crem &= crem - 1;
        const int kn = crem ? __builtin_ctz(crem) : kc;
        int ahead = gtot - 2 - g;
        if (ahead > NW - 3) ahead = NW - 3;
        wait_vmcnt(ahead > 0 ? ahead * NWI : 0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < gtot) issue_w();
        const char* wc = Wb + (g % NW) * W_BYTES;
        const char* wn = Wb + ((g + 1) % NW) * W_BYTES;
        M3D_HALO_MMA_G(0, 0, 1)
        __builtin_amdgcn_sched_barrier(0);
        M3D_HALO_LOAD(1, wc)
        int lnext[RW];
#pragma unroll
        for (int rt = 0; rt < RW; ++rt) lnext[rt] = (int)loc_s[kn * BM + arow0 + 32 * rt];
        __builtin_amdgcn_sched_barrier(0);
        M3D_HALO_MMA_G(0, 1, 0)
        M3D_HALO_MMA_G(0, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        M3D_HALO_MMA_G(1, 0, 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RW; ++rt) lcur[rt] = lnext[rt] == (int)kHaloNone ? kHaloCap : lnext[rt];
        M3D_HALO_LOAD(0, wn)
        __builtin_amdgcn_sched_barrier(0);
        M3D_HALO_MMA_G(1, 1, 0)
        M3D_HALO_MMA_G(1, 0, 0)
        kc = kn;
      }
    }
#undef M3D_HALO_MMA_G
#undef M3D_HALO_LOAD
#undef M3D_HALO_LOCALS
  } else if (gtot > 0) {
    // stage g = (slab, offset); row buffer g & 1, weight ring slot g % NW; the rows of stage g + 1 are requested
    // while stage g is consumed, their indices were read a stage earlier
#pragma unroll
    for (int u = 0; u < NW - 1; ++u)
      if (u < gtot) issue_w();
    int ids[R_V];
    uint32_t nrem = bmask;  // iterator over the stages whose rows are requested next
    int ncc = 0;
    auto advance = [&]() {
      nrem &= nrem - 1;
      if (!nrem) {
        nrem = bmask;
        ++ncc;
      }
    };
    load_ids(ids, __builtin_ctz(nrem));
    issue_rows(ids, ncc, 0);
    advance();
    if (gtot > 1) load_ids(ids, __builtin_ctz(nrem));
    for (int g = 0; g < gtot; ++g) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (g + 1 < gtot) {
        issue_rows(ids, ncc, (g + 1) & 1);
        advance();
        if (g + 2 < gtot) load_ids(ids, __builtin_ctz(nrem));
      }
      if (issued < gtot) issue_w();
      compute_rows(g % NW, g & 1);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // every wave is done with the halo and the weight ring; reuse them for the output tile
  float* S = reinterpret_cast<float*>(ldsb);
  if (blockIdx.x == 0 && blockIdx.y == 0) clear_zero_row(a);
#pragma unroll
  for (int rt = 0; rt < RW; ++rt) conv_stage_tile<NT>(a, acc[rt], S, BN, wave * RW + rt, 0, col_base, lane);
  __syncthreads();
  conv_store_tile<BM, BN, NTHR>(a, S, orow_s, col_base, tid);
}

template <int NT, int NW>
__global__ __launch_bounds__(256) void conv_halo_kernel(ConvArgs a, HaloArgs hl) {
  conv_halo_body<NT, NW>(a, hl);
}

template <int NT, int NW>
static int launch_halo(const ConvArgs& a, const HaloArgs& hl, hipStream_t stream) {
  using G = DmaGeom<2>;
  constexpr int BN = 32 * NT;
  constexpr size_t lds = (size_t)(kHaloCap + 1) * G::ROWB + (size_t)NW * BN * G::ROWB + 27 * kHaloRows * 2 + kHaloRows * 4 +
                         kHaloCap * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert((size_t)kHaloRows * BN * 4 <= (size_t)(kHaloCap + 1) * G::ROWB + (size_t)NW * BN * G::ROWB,
                "output tile must fit in the halo + weight regions");
  static LdsGrant grant;
  if (int rc = grant.ensure(reinterpret_cast<const void*>(&conv_halo_kernel<NT, NW>), lds)) return rc;
  dim3 grid((unsigned)(a.npad / kHaloRows), (unsigned)(a.cout / BN), 1);
  hipLaunchKernelGGL((conv_halo_kernel<NT, NW>), grid, dim3(256), lds, stream, a, hl);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

// Called by m3d_conv_fwd_sf32 when the caller supplied a halo and the shape suits it.
int run_conv_halo(const ConvArgs& a, const int32_t* d_cnt, const int32_t* d_rows, const uint16_t* d_loc,
                  const int32_t* d_order, hipStream_t stream) {
  HaloArgs hl{d_cnt, d_rows, d_loc, d_order};
  if (a.cout % 128 == 0) return launch_halo<4, 3>(a, hl, stream);
  if (a.cout % 96 == 0) {
    static const int nw = getenv("M3D_HALO_NW") ? atoi(getenv("M3D_HALO_NW")) : 4;
    if (nw == 3) return launch_halo<3, 3>(a, hl, stream);
    return launch_halo<3, 4>(a, hl, stream);
  }
  if (a.cout % 64 == 0) return launch_halo<2, 4>(a, hl, stream);
  return launch_halo<1, 4>(a, hl, stream);
}

}  // namespace m3d

using namespace m3d;

extern "C" int m3d_halo_cap(void) { return kHaloCap; }
extern "C" int m3d_halo_tile_rows(void) { return kHaloRows; }

extern "C" int m3d_build_halo(const int32_t* d_nbr, const int32_t* d_out_row, int k, int64_t npad, int64_t n_in,
                              int order_by_schedule, int32_t* d_halo_cnt, int32_t* d_halo_rows, uint16_t* d_halo_loc,
                              int32_t* d_halo_order, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(d_nbr && d_halo_cnt && d_halo_rows && d_halo_loc && d_halo_order, "m3d_build_halo: null pointer");
  M3D_CHECK_ARG(k >= 1 && k <= 27, "m3d_build_halo: k=%d", k);
  M3D_CHECK_ARG(npad > 0 && npad % kHaloRows == 0 && npad < ((int64_t)1 << 31), "m3d_build_halo: npad=%lld must be a multiple of %d",
                (long long)npad, kHaloRows);
  int32_t* pos = nullptr;
  if (order_by_schedule) {
    // submanifold map: input rows are the output rows, a row's position in the schedule orders the halo
    M3D_CHECK_ARG(d_out_row != nullptr && n_in > 0, "m3d_build_halo: order_by_schedule needs out_row and n_in");
    if (d_ws == nullptr || ws_bytes < (size_t)n_in * sizeof(int32_t)) {
      set_error("m3d_build_halo: workspace too small (%zu given, %zu needed)", ws_bytes, (size_t)n_in * sizeof(int32_t));
      return M3D_ENOSPC;
    }
    pos = static_cast<int32_t*>(d_ws);
    M3D_HIP(hipMemsetAsync(pos, 0, (size_t)n_in * sizeof(int32_t), stream));
    hipLaunchKernelGGL(invert_rows_kernel, dim3((unsigned)cdiv(npad, 256)), dim3(256), 0, stream, d_out_row, npad, n_in, pos);
  }
  hipLaunchKernelGGL(halo_build_kernel, dim3((unsigned)(npad / kHaloRows)), dim3(256), 0, stream, d_nbr, k, npad, pos,
                     d_halo_cnt, d_halo_rows, d_halo_loc);
  hipLaunchKernelGGL(halo_order_kernel, dim3(1), dim3(1024), 0, stream, d_halo_cnt, (int)(npad / kHaloRows), d_halo_order);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

