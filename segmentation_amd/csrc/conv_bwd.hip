// Weight gradient of the sparse convolution (training step), exact f32 on v_mfma_f32_32x32x2_f32.
//
//   dW[co][k][ci] = sum over schedule rows t of dout[row(t)][co] * in[nbr[k][t]][ci]
//
// For one kernel offset this is a GEMM (Cout x rows) x (rows x Cin) whose reduction runs over the rows.  A
// workgroup owns one offset k, one (32*TA x 32*TB) tile of the (Cout, Cin) matrix and one contiguous range of
// schedule rows; its four waves take 32-row tiles of that range in turn (tiles without a pair at offset k are
// skipped through the schedule's tile masks).  The matrix instruction consumes two rows per issue: lane
// (c, h) holds dout[row(t + h)][co0 + c] as A[m = c][k = h] and in[nbr[k][t + h]][ci0 + c] as B[k = h][n = c],
// both read straight from global memory as 128-byte row segments (no LDS staging: every operand is used once).
// Partial sums of the waves are combined in LDS, those of the row ranges in a second kernel, in a fixed order
// (no atomics: the result is run-to-run deterministic).
//
// The input gradient needs no kernel of its own: it is the forward convolution of dout with transposed weights
// over the transposed kernel map (see m3d.h).
#include "conv_shared.h"

namespace m3d {

struct BwdWArgs {
  const float* in;         // [n_in, cin]
  const float* dout;       // [n_out, cout]
  const int32_t* nbr;      // [k, npad] or nullptr (dense 1x1: row t pairs with row t)
  const int32_t* out_row;  // [npad] or nullptr
  const uint32_t* tile_mask;
  float* slab;             // [nsplit][k][cout][cin]
  int64_t n, npad;
  int cin, cout, kvol, nsplit;
};

// Combine the four waves' partial sums through LDS (fixed order), then one store per element of the
// (split, offset) slab.
template <int TA, int TB>
__device__ __forceinline__ void bwd_weight_store(const BwdWArgs& a, floatx16 (&acc)[TA][TB], int k, int split, int co0,
                                                 int ci0, int lane, int wave, int c, int h) {
  __shared__ float red[3][TA * TB * 1024];  // waves 1..3
  if (wave > 0) {
#pragma unroll
    for (int x = 0; x < TA; ++x)
#pragma unroll
      for (int y = 0; y < TB; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][((x * TB + y) * 16 + r) * 64 + lane] = acc[x][y][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* dst = a.slab + (((int64_t)split * a.kvol + k) * a.cout) * a.cin;
#pragma unroll
    for (int x = 0; x < TA; ++x)
#pragma unroll
      for (int y = 0; y < TB; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int e = ((x * TB + y) * 16 + r) * 64 + lane;
          const float v = ((acc[x][y][r] + red[0][e]) + red[1][e]) + red[2][e];
          const int m = (r & 3) + 8 * (r >> 2) + 4 * h;  // accumulator register r of lane (c, h) is row m, column c
          dst[(int64_t)(co0 + x * 32 + m) * a.cin + ci0 + y * 32 + c] = v;
        }
  }
}

template <int TA, int TB>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(BwdWArgs a) {
  const int k = blockIdx.x, split = blockIdx.z;
  const int tiles_ci = a.cin / (32 * TB);
  const int co0 = (blockIdx.y / tiles_ci) * 32 * TA, ci0 = (blockIdx.y % tiles_ci) * 32 * TB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  // row range of this split, in 32-row tiles
  const int64_t ntile = a.npad / 32;
  const int64_t per = (ntile + a.nsplit - 1) / a.nsplit;
  const int64_t tile0 = (int64_t)split * per, tile1 = tile0 + per < ntile ? tile0 + per : ntile;
  floatx16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int64_t tile = tile0 + wave; tile < tile1; tile += 4) {
    if (a.tile_mask && !((a.tile_mask[tile] >> k) & 1u)) continue;
    const int64_t tb = tile * 32;
#pragma unroll 4
    for (int s = 0; s < 32; s += 2) {
      const int64_t t = tb + s + h;
      int j, i;
      if (a.nbr) {
        j = a.out_row[t];
        i = a.nbr[(int64_t)k * a.npad + t];
      } else {
        j = i = t < a.n ? (int)t : -1;
      }
      const bool ok = j >= 0 && i >= 0;
      float av[TA], bv[TB];
#pragma unroll
      for (int x = 0; x < TA; ++x) av[x] = ok ? a.dout[(int64_t)j * a.cout + co0 + x * 32 + c] : 0.f;
#pragma unroll
      for (int x = 0; x < TB; ++x) bv[x] = ok ? a.in[(int64_t)i * a.cin + ci0 + x * 32 + c] : 0.f;
#pragma unroll
      for (int x = 0; x < TA; ++x)
#pragma unroll
        for (int y = 0; y < TB; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
    }
  }
  bwd_weight_store<TA, TB>(a, acc, k, split, co0, ci0, lane, wave, c, h);
}

// The same reduction with split-bf16 products on v_mfma_f32_32x32x16_bf16: sixteen schedule rows per issue, lane
// (c, h) holds channel c of rows t + 8h .. t + 8h + 7 for both operands (eight 128-byte row segments each, straight
// from global memory), splits every value into hi = bf16(v) and lo = bf16(v - hi) in registers and forms
// hi*hi + hi*lo + lo*hi in the f32 accumulator.  One sixteenth of the matrix-pipe time of the f32 instruction per
// product; what remains is the latency of the gather (row indices, then the rows), so the loop is a two-stage
// pipeline over the wave's items (an item = sixteen rows of a tile that has the offset): while item i is split and
// multiplied, the rows of item i + 1 and the row indices of item i + 2 are in flight.  The tile masks are read
// with wave-uniform addresses (scalar loads, which do not queue behind the vector loads).
template <int TA, int TB>
struct BwdItem {
  int64_t tile;
  int half;  // 0 or 1: rows 0..15 / 16..31 of the tile
  bool ok;
};

template <int TA, int TB, bool DENSE>
__global__ __launch_bounds__(256) void conv_bwd_weight_split_kernel(BwdWArgs a) {
  const int k = blockIdx.x, split = blockIdx.z;
  const int tiles_ci = a.cin / (32 * TB);
  const int co0 = (blockIdx.y / tiles_ci) * 32 * TA, ci0 = (blockIdx.y % tiles_ci) * 32 * TB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t ntile = a.npad / 32;
  const int64_t per = (ntile + a.nsplit - 1) / a.nsplit;
  const int64_t tile0 = (int64_t)split * per, tile1 = tile0 + per < ntile ? tile0 + per : ntile;
  floatx16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* dout_c = a.dout + co0 + c;
  const float* in_c = a.in + ci0 + c;
  using Item = BwdItem<TA, TB>;
  struct Idx { int jj[8], ii[8]; };
  struct Raw { float av[TA][8], bv[TB][8]; unsigned ok; };  // ok: bit i = row i of the lane's eight is a pair

  auto active_from = [&](int64_t t) {  // first tile >= t of this wave's sequence that has offset k
    if (!DENSE)
      while (t < tile1 && !((a.tile_mask[t] >> k) & 1u)) t += 4;
    return t;
  };
  auto next_item = [&](const Item& it) {
    Item n;
    if (it.half == 0) {
      n.tile = it.tile; n.half = 1; n.ok = it.ok;
    } else {
      n.tile = it.ok ? active_from(it.tile + 4) : it.tile; n.half = 0; n.ok = it.ok && n.tile < tile1;
    }
    return n;
  };
  // Loads are issued for every item, valid or not (an item past the end reads the indices of the range's last tile
  // and none of the rows): the number of loads between an issue and its use is then the same on every path, and the
  // waits in front of the matrix work count exactly the younger loads.
  auto load_idx = [&](Idx& ix, const Item& it) {
    const int64_t tile = it.tile < tile1 ? it.tile : tile1 - 1;
    const int64_t t0 = tile * 32 + 16 * it.half + 8 * h;
    int4 j0, j1, i0, i1;
    if (DENSE) {
      const int b = (int)t0;
      j0 = make_int4(b, b + 1, b + 2, b + 3);
      j1 = make_int4(b + 4, b + 5, b + 6, b + 7);
      const int lim = (int)a.n;
      j0.x = j0.x < lim ? j0.x : -1; j0.y = j0.y < lim ? j0.y : -1; j0.z = j0.z < lim ? j0.z : -1; j0.w = j0.w < lim ? j0.w : -1;
      j1.x = j1.x < lim ? j1.x : -1; j1.y = j1.y < lim ? j1.y : -1; j1.z = j1.z < lim ? j1.z : -1; j1.w = j1.w < lim ? j1.w : -1;
      i0 = j0;
      i1 = j1;
    } else {
      j0 = *reinterpret_cast<const int4*>(a.out_row + t0);
      j1 = *reinterpret_cast<const int4*>(a.out_row + t0 + 4);
      i0 = *reinterpret_cast<const int4*>(a.nbr + (int64_t)k * a.npad + t0);
      i1 = *reinterpret_cast<const int4*>(a.nbr + (int64_t)k * a.npad + t0 + 4);
    }
    const int dead = it.ok ? 0 : -1;  // an item past the end: every row index becomes negative
    ix.jj[0] = j0.x | dead; ix.jj[1] = j0.y | dead; ix.jj[2] = j0.z | dead; ix.jj[3] = j0.w | dead;
    ix.jj[4] = j1.x | dead; ix.jj[5] = j1.y | dead; ix.jj[6] = j1.z | dead; ix.jj[7] = j1.w | dead;
    ix.ii[0] = i0.x; ix.ii[1] = i0.y; ix.ii[2] = i0.z; ix.ii[3] = i0.w;
    ix.ii[4] = i1.x; ix.ii[5] = i1.y; ix.ii[6] = i1.z; ix.ii[7] = i1.w;
  };
  // Rows without a pair are read from row 0 and dropped when the operands are split: every load is issued
  // unconditionally (no branch per row), and a missing pair costs a cache hit.
  auto load_rows = [&](Raw& r, const Idx& ix) {
    r.ok = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = ix.jj[i] >= 0 && ix.ii[i] >= 0;
      r.ok |= ok ? (1u << i) : 0u;
      const int64_t jr = ok ? ix.jj[i] : 0, ir = ok ? ix.ii[i] : 0;
#pragma unroll
      for (int x = 0; x < TA; ++x) r.av[x][i] = dout_c[jr * a.cout + x * 32];
#pragma unroll
      for (int x = 0; x < TB; ++x) r.bv[x][i] = in_c[ir * a.cin + x * 32];
    }
  };
  auto multiply = [&](const Raw& r) {
    bf16x8 ahi[TA], alo[TA], bhi[TB], blo[TB];
#pragma unroll
    for (int x = 0; x < TA; ++x)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = (r.ok >> i) & 1u ? r.av[x][i] : 0.f;  // one zero operand is enough
        ahi[x][i] = (__bf16)v;
        alo[x][i] = (__bf16)(v - (float)ahi[x][i]);
      }
#pragma unroll
    for (int x = 0; x < TB; ++x)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bhi[x][i] = (__bf16)r.bv[x][i];
        blo[x][i] = (__bf16)(r.bv[x][i] - (float)bhi[x][i]);
      }
#pragma unroll
    for (int x = 0; x < TA; ++x)
#pragma unroll
      for (int y = 0; y < TB; ++y) {
        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[x], bhi[y], acc[x][y], 0, 0, 0);
        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[x], blo[y], acc[x][y], 0, 0, 0);
        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[x], bhi[y], acc[x][y], 0, 0, 0);
      }
  };

  Item cur;
  cur.tile = active_from(tile0 + wave);
  cur.half = 0;
  cur.ok = cur.tile < tile1;
  if (cur.ok) {  // wave-uniform
    Idx ixa, ixb;
    Raw ra, rb;
    load_idx(ixa, cur);
    load_rows(ra, ixa);
    Item nxt = next_item(cur);
    load_idx(ixb, nxt);
    while (true) {
      // rows of the current item -> ra (in flight); indices of the next item -> ixb (in flight)
      const Item n2 = next_item(nxt);
      load_rows(rb, ixb);
      load_idx(ixa, n2);
      multiply(ra);
      if (!nxt.ok) break;
      const Item n3 = next_item(n2);
      load_rows(ra, ixa);
      load_idx(ixb, n3);
      multiply(rb);
      if (!n2.ok) break;
      nxt = n3;
    }
  }
  bwd_weight_store<TA, TB>(a, acc, k, split, co0, ci0, lane, wave, c, h);
}

// dW[co][k][ci] (reference layout) = sum over splits of slab[s][k][co][ci], in split order
__global__ void conv_bwd_weight_reduce_kernel(const float* __restrict__ slab, int nsplit, int kvol, int cout, int cin,
                                              float* __restrict__ dw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)kvol * cout * cin;
  if (e >= per) return;
  const int ci = (int)(e % cin), co = (int)((e / cin) % cout), k = (int)(e / ((int64_t)cin * cout));
  float v = 0.f;
  for (int s = 0; s < nsplit; ++s) v += slab[(int64_t)s * per + e];
  dw[((int64_t)co * kvol + k) * cin + ci] = v;
}

static int pick_tile(int groups) { return groups % 3 == 0 ? 3 : (groups % 2 == 0 ? 2 : 1); }

static int plan_nsplit(int64_t npad, int kvol, int tiles) {
  // enough workgroups to fill the chip (~4 per CU), at least 8 row tiles per workgroup
  int64_t want = 1024 / ((int64_t)kvol * tiles);
  if (want < 1) want = 1;
  const int64_t max_by_rows = npad / 32 / 8 > 0 ? npad / 32 / 8 : 1;
  if (want > max_by_rows) want = max_by_rows;
  if (want > 64) want = 64;
  return (int)want;
}

}  // namespace m3d

using namespace m3d;

extern "C" size_t m3d_conv_bwd_weight_workspace_bytes(int64_t n, int cin, int cout, int k) {
  if (cin <= 0 || cout <= 0 || cin % 32 || cout % 32 || k < 1) return 256;
  const int64_t npad = m3d_schedule_npad(n);
  const int ta = pick_tile(cout / 32), tb = pick_tile(cin / 32);
  const int nsplit = plan_nsplit(npad, k, (cout / (32 * ta)) * (cin / (32 * tb)));
  return align_up((size_t)nsplit * k * cout * cin * sizeof(float));
}

static int conv_bwd_weight_launch(const char* who, bool split_products, const float* d_in, int cin, const float* d_dout,
                                  int cout, int k, const int32_t* d_nbr, const int32_t* d_out_row,
                                  const uint32_t* d_tile_mask, int64_t n, int64_t npad, float* d_dweight, void* d_ws,
                                  size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(n > 0 && d_in && d_dout && d_dweight, "%s: null pointer or empty input", who);
  M3D_CHECK_ARG(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0 && k >= 1 && k <= 27,
                "%s: channel counts must be multiples of 32 (pad the input), 1 <= k <= 27", who);
  if (d_nbr == nullptr) {
    M3D_CHECK_ARG(k == 1 && d_out_row == nullptr && d_tile_mask == nullptr, "%s: dense mode needs k == 1 and no schedule", who);
    npad = m3d_schedule_npad(n);
  } else {
    M3D_CHECK_ARG(d_out_row && d_tile_mask && npad >= n && npad % 32 == 0, "%s: bad schedule", who);
  }
  const int ta = pick_tile(cout / 32), tb = pick_tile(cin / 32);
  const int tiles = (cout / (32 * ta)) * (cin / (32 * tb));
  const int nsplit = plan_nsplit(npad, k, tiles);
  const size_t need = (size_t)nsplit * k * cout * cin * sizeof(float);
  if (d_ws == nullptr || ws_bytes < need) {
    set_error("%s: workspace too small (%zu given, %zu needed)", who, ws_bytes, need);
    return M3D_ENOSPC;
  }
  BwdWArgs a{d_in, d_dout, d_nbr, d_out_row, d_tile_mask, static_cast<float*>(d_ws), n, npad, cin, cout, k, nsplit};
  dim3 grid((unsigned)k, (unsigned)tiles, (unsigned)nsplit);
#define M3D_BW(TA_, TB_)                                                                                      \
  if (ta == TA_ && tb == TB_) {                                                                                \
    if (split_products) {                                                                                      \
      if (d_nbr)                                                                                               \
        hipLaunchKernelGGL((conv_bwd_weight_split_kernel<TA_, TB_, false>), grid, dim3(256), 0, stream, a);    \
      else                                                                                                     \
        hipLaunchKernelGGL((conv_bwd_weight_split_kernel<TA_, TB_, true>), grid, dim3(256), 0, stream, a);     \
    } else                                                                                                     \
      hipLaunchKernelGGL((conv_bwd_weight_kernel<TA_, TB_>), grid, dim3(256), 0, stream, a);                   \
  }
  M3D_BW(1, 1) M3D_BW(1, 2) M3D_BW(1, 3) M3D_BW(2, 1) M3D_BW(2, 2) M3D_BW(2, 3) M3D_BW(3, 1) M3D_BW(3, 2) M3D_BW(3, 3)
#undef M3D_BW
  hipLaunchKernelGGL(conv_bwd_weight_reduce_kernel, dim3((unsigned)cdiv((int64_t)k * cout * cin, 256)), dim3(256), 0, stream,
                     a.slab, nsplit, k, cout, cin, d_dweight);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_conv_bwd_weight_f32(const float* d_in, int cin, const float* d_dout, int cout, int k,
                                       const int32_t* d_nbr, const int32_t* d_out_row, const uint32_t* d_tile_mask,
                                       int64_t n, int64_t npad, float* d_dweight, void* d_ws, size_t ws_bytes,
                                       void* stream) {
  return conv_bwd_weight_launch("m3d_conv_bwd_weight_f32", false, d_in, cin, d_dout, cout, k, d_nbr, d_out_row,
                                d_tile_mask, n, npad, d_dweight, d_ws, ws_bytes, stream);
}

extern "C" int m3d_conv_bwd_weight_bf16x3(const float* d_in, int cin, const float* d_dout, int cout, int k,
                                          const int32_t* d_nbr, const int32_t* d_out_row, const uint32_t* d_tile_mask,
                                          int64_t n, int64_t npad, float* d_dweight, void* d_ws, size_t ws_bytes,
                                          void* stream) {
  return conv_bwd_weight_launch("m3d_conv_bwd_weight_bf16x3", true, d_in, cin, d_dout, cout, k, d_nbr, d_out_row,
                                d_tile_mask, n, npad, d_dweight, d_ws, ws_bytes, stream);
}
