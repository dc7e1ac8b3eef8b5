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
// product; what remains is the gather.
template <int TA, int TB>
__global__ __launch_bounds__(256) void conv_bwd_weight_split_kernel(BwdWArgs a) {
  const int k = blockIdx.x, split = blockIdx.z;
  const int tiles_ci = a.cin / (32 * TB);
  const int co0 = (blockIdx.y / tiles_ci) * 32 * TA, ci0 = (blockIdx.y % tiles_ci) * 32 * TB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  for (int64_t tile = tile0 + wave; tile < tile1; tile += 4) {
    if (a.tile_mask && !((a.tile_mask[tile] >> k) & 1u)) continue;
#pragma unroll
    for (int s = 0; s < 32; s += 16) {
      const int64_t t0 = tile * 32 + s + 8 * h;
      int jj[8], ii[8];
      if (a.nbr) {
        const int4 j0 = *reinterpret_cast<const int4*>(a.out_row + t0);
        const int4 j1 = *reinterpret_cast<const int4*>(a.out_row + t0 + 4);
        const int4 i0 = *reinterpret_cast<const int4*>(a.nbr + (int64_t)k * a.npad + t0);
        const int4 i1 = *reinterpret_cast<const int4*>(a.nbr + (int64_t)k * a.npad + t0 + 4);
        jj[0] = j0.x; jj[1] = j0.y; jj[2] = j0.z; jj[3] = j0.w; jj[4] = j1.x; jj[5] = j1.y; jj[6] = j1.z; jj[7] = j1.w;
        ii[0] = i0.x; ii[1] = i0.y; ii[2] = i0.z; ii[3] = i0.w; ii[4] = i1.x; ii[5] = i1.y; ii[6] = i1.z; ii[7] = i1.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) jj[i] = ii[i] = t0 + i < a.n ? (int)(t0 + i) : -1;
      }
      float av[TA][8], bv[TB][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = jj[i] >= 0 && ii[i] >= 0;
#pragma unroll
        for (int x = 0; x < TA; ++x) av[x][i] = ok ? dout_c[(int64_t)jj[i] * a.cout + x * 32] : 0.f;
#pragma unroll
        for (int x = 0; x < TB; ++x) bv[x][i] = ok ? in_c[(int64_t)ii[i] * a.cin + x * 32] : 0.f;
      }
      bf16x8 ahi[TA], alo[TA], bhi[TB], blo[TB];
#pragma unroll
      for (int x = 0; x < TA; ++x)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ahi[x][i] = (__bf16)av[x][i];
          alo[x][i] = (__bf16)(av[x][i] - (float)ahi[x][i]);
        }
#pragma unroll
      for (int x = 0; x < TB; ++x)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          bhi[x][i] = (__bf16)bv[x][i];
          blo[x][i] = (__bf16)(bv[x][i] - (float)bhi[x][i]);
        }
#pragma unroll
      for (int x = 0; x < TA; ++x)
#pragma unroll
        for (int y = 0; y < TB; ++y) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[x], bhi[y], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[x], blo[y], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[x], bhi[y], acc[x][y], 0, 0, 0);
        }
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
    if (split_products)                                                                                        \
      hipLaunchKernelGGL((conv_bwd_weight_split_kernel<TA_, TB_>), grid, dim3(256), 0, stream, a);             \
    else                                                                                                       \
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
