/*
 * m3d.h -- C ABI of libm3d_hip.so: the MI355X (gfx950) sparse-convolution engine behind
 * Mosaic3D's SpUNet hot path (voxelize -> kernel maps -> sparse conv -> CLIP-text argmax).
 *
 * The reference (SLAM-Dunk-Prometheus/Segmentation, Python) reaches this arithmetic through the
 * third-party wheels `spconv` and `torch_scatter` plus numpy on the host; each entry point below
 * cites the reference call site it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory owned by the caller (PyTorch tensors in practice);
 *     the library never allocates or frees device memory; its only state is a thread-local error string,
 *     the per-device scratch of m3d_bn_stats_static and the per-device f16 status word (m3d_f16_status);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and nothing synchronises, with one
 *     exception: m3d_f16_status waits for `stream` before it reads the status word back;
 *   - data-dependent output sizes (number of voxels) are written to device memory; outputs are
 *     sized by the caller for the worst case (one row per input row);
 *   - return value: 0 on success, negative M3D_E* code on failure; m3d_last_error() gives text;
 *   - scratch: each call that needs temporary memory takes (d_ws, ws_bytes); the matching
 *     *_workspace_bytes() function returns the size required for given problem sizes.
 */
#ifndef M3D_H_
#define M3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3D_OK 0
#define M3D_EINVAL (-1)   /* bad argument (null pointer, unsupported size/shape) */
#define M3D_ENOSPC (-2)   /* workspace too small */
#define M3D_EHIP (-3)     /* a HIP runtime call or kernel launch failed */

#define M3D_PREC_F32 0     /* exact f32 products on v_mfma_f32_32x32x2_f32 */
#define M3D_PREC_BF16X3 1  /* f32 operands split in 2 bf16 terms, 3 bf16 MFMA products (~2^-17 per product) */
#define M3D_PREC_BF16X6 2  /* f32 operands split in 3 bf16 terms, 6 bf16 MFMA products (~2^-24 per product) */
#define M3D_PREC_BF16 3    /* operands rounded to bf16, one bf16 MFMA product, f32 accumulation (~2^-8 per product) */
#define M3D_PREC_F16X3 4   /* f32 operands as an f16 pair (value = t0 + t1 * 2^-11, exact to 2^-22), 3 f16 MFMA products
                              (~2^-21 per product); operands must lie inside the f16 range (|x| <= 65504): see
                              m3d_f16_status.  Split-format code ("terms") of the pair: M3D_FMT_F16_PAIR */
#define M3D_FMT_F16_PAIR 4

#define M3D_HASH_FNV 0
#define M3D_HASH_RAVEL 1
#define M3D_GRID_DIV 0    /* trunc(coord / grid_size): torch CPU kernel */
#define M3D_GRID_RECIP 1  /* trunc(coord * (1/grid_size)): torch CUDA kernel for tensor/scalar */
#define M3D_GRID_FLOOR 2  /* floor(coord / grid_size): GridSample (transform.py:753-754) */

int m3d_version(void);
const char* m3d_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Voxelize.  Replaces Point.sparsify + mean_pooling + fnv_hash_vec
 * (mosaic3d/src/models/utils/structure.py:197-251, :34-66, :21-31) and offset2batch
 * (mosaic3d/src/models/utils/misc.py:16-21).  With M3D_GRID_FLOOR and pad = 0 it also gives
 * GridSample's hash / unique / inverse (mosaic3d/src/data/utils/transform.py:751-762).
 *
 *   d_coord   [n,3] f32     point coordinates
 *   d_feat    [n,c] f32     point features (c >= 1)
 *   d_offset  [b+1] i64     cumulative point counts with leading 0 (collate.py:96-102)
 * outputs (caller allocates n rows each; the first n_voxels rows are valid)
 *   d_voxel_coords [n,4] i32   (batch, x, y, z), ascending hash-key order
 *   d_voxel_feats  [n,c] f32   per-voxel mean
 *   d_v2p_map      [n]   i64   voxel row of every point
 *   d_keys         [n]   u64   sorted unique hash keys (may be NULL)
 *   d_grid_coord   [n,3] i32   per-point grid coordinate after the min shift (may be NULL)
 *   d_meta         [16]  i64   [0]=n_voxels [1..3]=spatial_shape (max grid + pad)
 *                              [4..6]=grid min before the shift [7..9]=grid max after the shift
 *                              [12]=1 if that extent + pad reaches 2^31 along an axis (infinite or huge coordinates:
 *                              the 32-bit grid coordinates have wrapped and the outputs are not to be used), else 0
 */
size_t m3d_voxelize_workspace_bytes(int64_t n, int c);
int m3d_voxelize(const float* d_coord, const float* d_feat, const int64_t* d_offset, int64_t n,
                 int c, int b, float grid_size, int pad, int hash_method, int grid_mode,
                 int32_t* d_voxel_coords, float* d_voxel_feats, int64_t* d_v2p_map,
                 uint64_t* d_keys, int32_t* d_grid_coord, int64_t* d_meta, void* d_ws,
                 size_t ws_bytes, void* stream);

/* pooling_method = "random" of Point.sparsify (random_pooling, mosaic3d/src/models/utils/structure.py:69-110; selected
 * through spconv_unet_v1m1_base.py:177-178, :335-339): voxel set, order and v2p_map are those of m3d_voxelize; the
 * feature of a voxel is the feature of one of its points, drawn uniformly.  The reference draws with numpy's global
 * RNG; here the draw is a function of (seed, point index), so a seed reproduces it.
 *   d_v2p_map [n_pts] i64 (from m3d_voxelize), d_feat [n_pts, c] f32 (may be NULL with d_voxel_feats)
 *   d_choice [n_vox] i64 chosen point of every voxel (may be NULL); d_voxel_feats [n_vox, c] f32 (may be NULL)
 *   workspace: m3d_voxel_pick_random_workspace_bytes(n_vox); n_pts < 2^32 */
size_t m3d_voxel_pick_random_workspace_bytes(int64_t n_vox);
int m3d_voxel_pick_random(const int64_t* d_v2p_map, const float* d_feat, int64_t n_pts, int c, int64_t n_vox,
                          uint64_t seed, int64_t* d_choice, float* d_voxel_feats, void* d_ws, size_t ws_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * GridSample's deterministic part (mosaic3d/src/data/utils/transform.py:751-762, :813-826, :858-887):
 * grid = floor(float64(coord) / grid_size) - min, key = FNV64-1A or ravel over (x, y, z), voxels in ascending
 * key order.  One cloud per call.
 *   d_grid_coord   [n,3] i32  per-point grid coordinate (after the min shift)
 *   d_inverse      [n]   i64  voxel of every point (GridSample's `inverse`)
 *   d_voxel_coords [n,4] i32  (0, x, y, z) of every voxel (GridSample's `grid_coord` of the sampled points)
 *   d_count        [n]   i32  points per voxel
 *   d_keys         [n]   u64  sorted unique keys (may be NULL)
 *   d_meta         [16]  i64  [0]=n_voxels [4..6]=grid min before the shift (min_coord = that * grid_size)
 * Which point of a voxel is kept (train: random, test: count.max() partitions) is the caller's choice; the
 * reference's own choice depends on numpy's unstable argsort and its global RNG.
 */
size_t m3d_grid_sample_workspace_bytes(int64_t n);
int m3d_grid_sample(const float* d_coord, int64_t n, double grid_size, int hash_type, int32_t* d_grid_coord,
                    int64_t* d_inverse, int32_t* d_voxel_coords, int32_t* d_count, uint64_t* d_keys,
                    int64_t* d_meta, void* d_ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Front end.  m3d_stray_convert replaces the arithmetic of convert_stray.py:24-39: points (float64, Y-up)
 * -> coord f32 with (x, y, z) -> (x, -z, y); colours in [0, 1] (float64) -> u8 by truncation of c * 255
 * (d_colors01 / d_color may both be NULL).  m3d_center_shift_normalize replaces CenterShift + NormalizeColor +
 * Collect(feat_keys=[color]) of transform.py:149-163, :108-113, :23-46: coord - ((xmin+xmax)/2, (ymin+ymax)/2,
 * zmin or 0) in float32; feat = float32(float64(color) / 127.5 - 1) (d_color / d_feat_out may both be NULL);
 * d_shift [3] f32 receives the shift.
 */
int m3d_stray_convert(const double* d_points, const double* d_colors01, int64_t n, float* d_coord,
                      uint8_t* d_color, void* stream);
size_t m3d_center_shift_workspace_bytes(int64_t n);
int m3d_center_shift_normalize(const float* d_coord, const uint8_t* d_color, int64_t n, int apply_z,
                               float* d_coord_out, float* d_feat_out, float* d_shift, void* d_ws,
                               size_t ws_bytes, void* stream);

/* Depth back-projection of one Stray Scanner frame, fused with the converter's Y-up -> Z-up rotation
 * (convert_stray.py:24-29 gives the output contract: f32 coordinates (x, -z, y), u8 colours; the reference
 * itself has no back-projection, it starts from a cloud made elsewhere).
 *   d_depth [h,w] u16 raw units (0 = no reading), d_conf [h,w] u8 or NULL, d_rgb [hc,wc,3] u8 or NULL
 *   intrinsics = (fx, fy, cx, cy) at depth resolution; pose_c2w = row-major 4x4 camera-to-world
 *   (camera axes: x right, y down, z forward); a pixel is kept iff depth != 0, conf >= conf_min and
 *   depth * depth_scale <= depth_max; every stride-th pixel of every stride-th row is visited.
 *   p_cam = ((u - cx) / fx * d, (v - cy) / fy * d, d) in float64, p_world = R p_cam + t.
 * Output: the kept pixels in row-major order are appended at d_coord[*d_count ...] (and d_color), *d_count
 * (device, i64) is advanced; a frame that would exceed `capacity` rows is dropped as a whole.
 */
size_t m3d_backproject_workspace_bytes(int h, int w, int stride);
int m3d_backproject(const uint16_t* d_depth, const uint8_t* d_conf, const uint8_t* d_rgb, int h, int w, int hc,
                    int wc, const double intrinsics[4], const double pose_c2w[16], double depth_scale,
                    double depth_max, int conf_min, int stride, float* d_coord, uint8_t* d_color,
                    int64_t capacity, int64_t* d_count, void* d_ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Submanifold kernel map (k = 3, dilation 1).  Replaces the indice-pair generation inside
 * spconv.SubMConv3d.forward (call sites spconv_unet_v1m3_pdnorm.py:98-117, :205-212).
 *   d_indices [n,4] i32 (batch, d0, d1, d2), rows distinct
 *   d_pair    [27,n] i32  pair[k][j] = i with indices[i] == indices[j] + offset(k), else -1;
 *                          k = (k0*3 + k1)*3 + k2, offset = (k0-1, k1-1, k2-1)
 *   batch_size: SparseConvTensor.batch_size (every batch index < batch_size), or 0 if unknown.  With a known batch
 *               size and batch_size * volume <= 2^28 cells the neighbours are found through an occupancy bitmap of
 *               the grid (bit rank = voxel number); otherwise through an open-addressing hash table of 64-bit keys.
 */
size_t m3d_subm_rulebook_workspace_bytes(int64_t n, const int64_t spatial_shape[3], int batch_size);
int m3d_subm_rulebook(const int32_t* d_indices, int64_t n, const int64_t spatial_shape[3], int batch_size,
                      int32_t* d_pair, void* d_ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Strided kernel map (kernel 2, stride 2, padding 0).  Replaces the indice-pair generation of
 * spconv.SparseConv3d (spconv_unet_v1m3_pdnorm.py:154-161); the same maps serve
 * spconv.SparseInverseConv3d (:184-190) through pair_bwd.
 *   d_out_indices [n,4] i32   output voxels, ascending row-major linear index over
 *                             (batch, spatial_shape/2); first n_out rows valid
 *   d_pair_fwd    [8,n]  i32  pair_fwd[k][o] = i or -1 (column stride n)
 *   d_pair_bwd    [8,n]  i32  pair_bwd[k][i] = o or -1
 *                 (columns at or beyond the row count - d_n_rows, else n - are left unwritten)
 *   d_n_out       [1]    i64
 *   batch_size: SparseConvTensor.batch_size (every batch index must be < batch_size)
 *   d_n_rows      [1]    i64  optional (NULL = all n rows): the number of valid rows of d_indices, read on
 *                             the device; n is then the capacity.  Lets a chain of strided maps (each
 *                             level's d_n_out feeding the next level's d_n_rows) be enqueued without a
 *                             host read-back in between.
 */
size_t m3d_strided_rulebook_workspace_bytes(int64_t n, const int64_t spatial_shape[3], int batch_size);
int m3d_strided_rulebook(const int32_t* d_indices, int64_t n, const int64_t* d_n_rows,
                         const int64_t spatial_shape[3], int batch_size, int32_t* d_out_indices,
                         int32_t* d_pair_fwd, int32_t* d_pair_bwd, int64_t* d_n_out, void* d_ws, size_t ws_bytes,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Tile schedule of a kernel map: the order in which the convolution kernel walks output rows.
 * Has no counterpart in the reference (spconv keeps this inside its implicit-GEMM planner).
 *   d_pair [k,ld] i32  canonical kernel map (ld = column stride, n valid columns)
 *   d_indices [n,4]    coordinates of the OUTPUT rows (modes 1 and 3, else may be NULL)
 *   spatial_shape, batch_size: bounds of d_indices (modes 1 and 3; spatial_shape may be NULL)
 *   d_base_order [n]   optional starting permutation of the rows (NULL = identity)
 *   mode 0: the base order as is; 1: Morton order of d_indices; 2: rows grouped by their first
 *           active kernel offset (inverse conv), base order kept inside a group; 3 (k = 27 only):
 *           rows grouped by coarse spatial cell (2^s voxels per side, cells in Morton order; s is
 *           passed as mode = 3 | s << 8, s = 0 means 6, s = 15 one cell) and, inside a cell, by
 *           which corner/edge offsets of the 3x3x3 stencil they have, so that the rows of a tile
 *           miss the same offsets and the tile skips them
 * outputs, npad = m3d_schedule_npad(n):
 *   d_nbr      [k,npad] i32  nbr[k][t] = pair[k][row(t)], -1 in the padding
 *   d_out_row  [npad]   i32  row(t), -1 in the padding
 *   d_tile_mask[npad/32] u32 bit k set iff any of the tile's 32 rows has a pair at offset k
 */
int64_t m3d_schedule_npad(int64_t n);
size_t m3d_schedule_workspace_bytes(int64_t n);
int m3d_build_schedule(const int32_t* d_pair, int k, int64_t ld, int64_t n,
                       const int32_t* d_indices, const int64_t* spatial_shape, int batch_size,
                       int mode, const int32_t* d_base_order, int32_t* d_nbr, int32_t* d_out_row,
                       uint32_t* d_tile_mask, void* d_ws, size_t ws_bytes, void* stream);

/* Optional companion of a schedule: the order in which a launch with 128-row tiles takes them.  The tiles are
 * split into three classes by their number of active kernel offsets (the thirds of this schedule's tiles); the
 * heaviest class goes first, schedule order inside a class.  Workgroups start in index order, so the long
 * reductions go first and the short ones fill the tail of the launch (measured -12 % / -20 % on the 3x3x3
 * convolutions of the two largest levels, for more HBM fetches: spatial neighbours no longer run side by side).
 * Both convolution entry points apply it where their plan uses 128-row tiles and ignore it elsewhere.
 *   d_tile_mask [npad/32] from m3d_build_schedule;  d_tile_order [npad/128] i32 (a permutation of 0 .. npad/128-1)
 * No counterpart in the reference. */
int m3d_schedule_tile_order(const uint32_t* d_tile_mask, int64_t npad, int32_t* d_tile_order, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution forward, f32 in / f32 out, implicit GEMM on the matrix cores.  `precision`
 * selects how the f32 products are formed (M3D_PREC_*); accumulation is always f32.
 * Replaces the feature arithmetic of spconv.SubMConv3d / SparseConv3d / SparseInverseConv3d
 * (call sites above) and Custom1x1Subm3d's torch.mm (structure.py:336-347), with the
 * eval-mode PDBatchNorm affine, residual add and ReLU of BasicBlock.forward
 * (spconv_unet_v1m3_pdnorm.py:121-140, :57-68) fused into the epilogue:
 *
 *   out[row(t)] = act( scale * sum_k W[k] . concat(in0, in1)[nbr[k][t]] + shift + residual[row(t)] )
 *
 *   d_in0 [*,c0], d_in1 [*,c1] (d_in1 may be NULL, c1 = 0): the two K-slabs of a concatenated
 *        input (spconv_unet_v1m3_pdnorm.py:404) read without materialising the concat
 *   d_weight: the reference's [Cout,k0,k1,k2,Cin] f32 weight converted by m3d_conv_pack_weight for
 *        the same `precision` ([k,cout,cin] f32, or pre-split bf16 terms [k,cout,cin/G,terms,G])
 *   d_nbr/d_out_row/d_tile_mask: from m3d_build_schedule; all three NULL = dense 1x1 (k = 1)
 *   d_tile_order: NULL or m3d_schedule_tile_order's output for this schedule (launch order of 128-row tiles)
 *   d_scale/d_shift [cout] may be NULL (1 / 0); d_residual [n_out_rows,cout] may be NULL
 *   d_out [n_out_rows, ldo] (ldo >= cout, row stride in floats)
 *   n = number of real output rows; npad = schedule length
 * With M3D_PREC_BF16X3 / M3D_PREC_BF16X6 / M3D_PREC_F16X3 and channel counts that are multiples of 32 the f32 rows are
 * gathered into LDS as they are and split into 16-bit terms after the fragment read: the same terms, products and
 * summation order as m3d_conv_fwd_sf32 on a tensor split beforehand, so the two entry points agree bit for bit.  The
 * requests are buffer loads with 32-bit offsets, as there: rows0 / rows1 (the rows d_in0 / d_in1 hold; every index of
 * the kernel map lies below them) are required and every operand must stay below 4 GB, else M3D_EINVAL.  The other
 * precisions ignore rows0 / rows1 (0 is allowed).
 */
size_t m3d_conv_workspace_bytes(int64_t n, int cout, int k);
int m3d_conv_fwd_f32(const float* d_in0, int c0, int64_t rows0, const float* d_in1, int c1, int64_t rows1,
                     const float* d_weight, int k, int cout, const int32_t* d_nbr,
                     const int32_t* d_out_row, const uint32_t* d_tile_mask, int64_t n,
                     int64_t npad, const int32_t* d_tile_order, const float* d_scale, const float* d_shift,
                     const float* d_residual, int relu, float* d_out, int64_t ldo,
                     void* d_out_split, int out_terms, int precision, void* d_ws, size_t ws_bytes,
                     void* stream);

/* Same convolution with the activations in the engine-internal split format "sf32":
 *   [rows + 1][C/G][terms][G] bf16 with G = 16 channels per group for terms = 3 and 32 otherwise (terms = 1 is plain
 *   row-major bf16), term t = bf16 rounding of what terms < t left over, row `rows` all zero (missing neighbours
 *   gather it).  Inputs are moved to LDS by LDS-DMA: with two or three terms and the f16 pair the requests are
 *   buffer loads (buffer_load_dwordx4 ... offen lds: 32-bit per-lane byte offsets inside one descriptor per operand,
 *   a missing neighbour is an out-of-range offset, which the hardware answers with zeros), so every operand --
 *   (rows + 1) rows of either input, the packed weights -- must stay below 4 GB, else M3D_EINVAL; with one term they
 *   carry 64-bit per-lane addresses (global_load_lds_dwordx4) and the missing neighbours gather the zero row.  Products as in
 *   M3D_PREC_BF16 (terms = 1) / M3D_PREC_BF16X3 (terms = 2) / M3D_PREC_BF16X6 (terms = 3; the leading products and
 *   the small ones are accumulated separately and added once at the end).
 *   d_tile_order: NULL or m3d_schedule_tile_order's output for this schedule.
 *   d_weight: m3d_conv_pack_weight output for that precision.  Outputs: d_out (f32, may be NULL) and/or
 *   d_out_split (sf32 with cout/32 groups, may be NULL); m3d_conv_fwd_f32 can write d_out_split too
 *   (out_terms = 0: none); the kernel that writes d_out_split also clears its all-zero row.
 *   No counterpart in the reference: a storage format of this engine. */
int m3d_conv_fwd_sf32(const void* d_in0, int c0, int64_t rows0, const void* d_in1, int c1,
                      int64_t rows1, const void* d_weight, int k, int cout, const int32_t* d_nbr,
                      const int32_t* d_out_row, const uint32_t* d_tile_mask, int64_t n, int64_t npad, const int32_t* d_tile_order,
                      const float* d_scale, const float* d_shift, const float* d_residual, int relu,
                      float* d_out, int64_t ldo, void* d_out_split, int terms, void* d_ws,
                      size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (the backward of SubMConv3d / SparseConv3d / SparseInverseConv3d / the 1x1 convolution that
 * spconv computes for train_spunet_multidata_ppt, mosaic3d/src/models/lightning_modules/language_module.py:209-277).
 *
 * Input gradient: no entry point of its own.  With pair[k][j] = i the forward map,
 *   din[i] = sum_k W[k]^T dout[j]  over the (k, j) with pair[k][j] = i,
 * which is the FORWARD convolution (m3d_conv_fwd_f32) of dout with the transposed weights
 *   - submanifold k = 3: over the same schedule, weights W'[k] = W[26 - k]^T (pair[26-k][i] = j <=> pair[k][j] = i);
 *   - strided k2 s2: over the schedule of pair_bwd, weights W[k]^T;  its inverse: over the schedule of pair_fwd;
 *   - 1x1: dense, weights W^T.
 *
 * Weight gradient:  dW[co][k][ci] = sum_t dout[out_row[t]][co] * in[nbr[k][t]][ci], deterministic (partial sums
 * are combined in a fixed order, no atomics).  cin and cout must be multiples of 32 (pad the 3-channel stem input
 * with zero channels).  d_dweight has the reference layout [cout, k, cin].
 *   m3d_conv_bwd_weight_f32     exact f32 products (v_mfma_f32_32x32x2_f32)
 *   m3d_conv_bwd_weight_bf16x3  every f32 operand split into two bf16 terms in registers, products
 *                               hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, f32 accumulation (relative error
 *                               of a gradient ~1e-5); same arguments and workspace
 */
size_t m3d_conv_bwd_weight_workspace_bytes(int64_t n, int cin, int cout, int k);
int m3d_conv_bwd_weight_f32(const float* d_in, int cin, const float* d_dout, int cout, int k,
                            const int32_t* d_nbr, const int32_t* d_out_row, const uint32_t* d_tile_mask,
                            int64_t n, int64_t npad, float* d_dweight, void* d_ws, size_t ws_bytes,
                            void* stream);
int m3d_conv_bwd_weight_bf16x3(const float* d_in, int cin, const float* d_dout, int cout, int k,
                               const int32_t* d_nbr, const int32_t* d_out_row, const uint32_t* d_tile_mask,
                               int64_t n, int64_t npad, float* d_dweight, void* d_ws, size_t ws_bytes,
                               void* stream);

/* Batch normalisation in training mode (PDBatchNorm with batch statistics over all rows of this GPU,
 * mosaic3d/src/models/networks/spconv_unet_v1m3_pdnorm.py:57-68; nn.BatchNorm1d, eps 1e-3, momentum 0.01): the two
 * per-channel reductions over the [n, c] activations.  c a multiple of 4 (at most 1024), pointers 16-byte aligned,
 * float64 accumulation, partial sums combined in a fixed order (deterministic).  The elementwise passes stay with
 * the caller (y = (x - mean) * invstd * weight + bias;  dx from sum_dy / sum_dy_xmu as in synchronised batch norm).
 *   m3d_bn_stats:            d_mean[c], d_var[c] = mean and BIASED variance of every column of d_x
 *   m3d_bn_backward_reduce:  d_sum_dy[c] = sum_r dy[r][c];  d_sum_dy_xmu[c] = sum_r dy[r][c] * (x[r][c] - mean[c])
 *   workspace: m3d_bn_workspace_bytes(n, c)
 */
size_t m3d_bn_workspace_bytes(int64_t n, int c);
int m3d_bn_stats(const float* d_x, int64_t n, int c, float* d_mean, float* d_var, void* d_ws, size_t ws_bytes,
                 void* stream);
int m3d_bn_backward_reduce(const float* d_x, const float* d_dy, const float* d_mean, int64_t n, int c,
                           float* d_sum_dy, float* d_sum_dy_xmu, void* d_ws, size_t ws_bytes, void* stream);

/* Training-mode normalisation fused with what follows it in BasicBlock.forward / SPConvDown / SPConvUp
 * (spconv_unet_v1m3_pdnorm.py:121-140, :143-199): y = act(bn(x) + residual), one pass forward and two backward
 * instead of separate normalise, add and ReLU passes.  Statistics come from m3d_bn_stats (d_var = biased variance).
 *   m3d_bn_act_fwd:             y = relu?((x - mean) * rsqrt(var + eps) * weight + bias + residual)
 *                               (d_weight / d_bias / d_residual may be NULL = 1 / 0 / none)
 *   m3d_bn_act_backward_reduce: g = dy where y > 0 (d_y = the forward output; NULL = no ReLU, g = dy);
 *                               d_sum_g[c] = sum_r g, d_sum_g_xmu[c] = sum_r g * (x - mean); float64 sums, fixed
 *                               order, workspace m3d_bn_workspace_bytes(n, c)
 *   m3d_bn_act_backward:        dx = (g - sum_g/n - (x - mean) * sum_g_xmu/n / (var + eps)) * rsqrt(var + eps) * weight;
 *                               d_dres (may be NULL) = g, the gradient of the residual input
 *   (the parameter gradients are d_sum_g_xmu * rsqrt(var + eps) and d_sum_g).  c a multiple of 4, 16-byte aligned. */
int m3d_bn_act_fwd(const float* d_x, int64_t n, int c, const float* d_mean, const float* d_var, float eps,
                   const float* d_weight, const float* d_bias, const float* d_residual, int relu, float* d_y,
                   void* stream);
int m3d_bn_act_backward_reduce(const float* d_x, const float* d_dy, const float* d_y, const float* d_mean, int64_t n,
                               int c, float* d_sum_g, float* d_sum_g_xmu, void* d_ws, size_t ws_bytes, void* stream);
int m3d_bn_act_backward(const float* d_x, const float* d_dy, const float* d_y, int64_t n, int c, const float* d_mean,
                        const float* d_var, float eps, const float* d_weight, const float* d_sum_g,
                        const float* d_sum_g_xmu, float* d_dx, float* d_dres, void* stream);

/* m3d_bn_stats without any allocation and without a workspace argument: the per-workgroup partial sums live in a
 * fixed 4 MiB array inside the library's code object (every GPU the library runs on holds its own copy until the
 * process ends).  Has no counterpart in the reference; it is for callers that cannot allocate device memory: code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels, same summation
 * order and bit-identical results as m3d_bn_stats.  Limit: m3d_bn_workspace_bytes(n, c) must not exceed
 * m3d_bn_stats_static_max_bytes() (any n for c <= 256); beyond it the call fails with M3D_ENOSPC and launches
 * nothing.  The device that `stream` belongs to must be current.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams, two threads) race; calls on one stream are ordered and safe.
 * Nothing in the array has to start at zero (every partial a call reads is written by that call first). */
size_t m3d_bn_stats_static_max_bytes(void);
int m3d_bn_stats_static(const float* d_x, int64_t n, int c, float* d_mean, float* d_var, void* stream);

/* `terms` of the split-format entry points is 1, 2, 3 (bf16 terms) or M3D_FMT_F16_PAIR: [rows + 1][C/32][2][32] f16,
 * term 0 = f16(x), term 1 = f16((x - term 0) * 2^11).
 * m3d_f16_status: every kernel that writes the pair format sets a per-device status word when a value lies beyond
 * the f16 range (or is NaN); *h_status receives it (0 = all values were representable; the call waits for the
 * stream), `clear` != 0 resets it afterwards.  Results computed after a set status are not to be trusted: redo them
 * with M3D_PREC_BF16X6.  No counterpart in the reference. */
int m3d_f16_status(int* h_status, int clear, void* stream);
size_t m3d_sf32_bytes(int64_t n, int c, int terms);
int m3d_split_f32(const float* d_in, int64_t n, int c, int terms, void* d_out, void* stream);

/* Weight conversion for m3d_conv_fwd_f32: d_weight_ref is the reference layout [cout, k, cin] f32
 * (spconv's [Cout, k0, k1, k2, Cin] flattened; structure.py:338 relies on it for k = 1). */
size_t m3d_conv_packed_weight_bytes(int cout, int k, int cin, int precision);
int m3d_conv_pack_weight(const float* d_weight_ref, int cout, int k, int cin, int precision,
                         void* d_out, void* stream);
/* The weight of the layer's input-gradient convolution (see "Training step" above), packed straight from the
 * layer's own [cout, k, cin] weight: W'[ci][k'][co] = W[co][flip ? k - 1 - k' : k'][ci] (flip != 0 for submanifold
 * maps, whose transposed map is the same map with the offsets reversed).  The result is a packed weight with cin
 * output and cout input channels: m3d_conv_packed_weight_bytes(cin, k, cout, precision) bytes. */
int m3d_conv_pack_weight_transposed(const float* d_weight_ref, int cout, int k, int cin, int flip, int precision,
                                    void* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Open-vocabulary head.  Replaces clip_feat = features[v2p_map]
 * (mosaic3d/src/models/lightning_modules/language_module.py:202-207),
 * CLIPAlignmentEval.predict (mosaic3d/src/models/losses/clip_alignment_loss.py:138-147) and
 * logits.max(1)[1] (language_module.py:328).  The row argmax is taken per voxel and gathered
 * through v2p_map (argmax commutes with the row gather).
 *   d_logits [n_vox, ldl] f32  per-voxel logits (columns >= n_classes are ignored)
 *   d_voxel_label [n_vox] i32  first index of the row maximum
 *   d_point_label [n_pts] i64  voxel_label[v2p_map[p]]
 */
int m3d_argmax_rows(const float* d_logits, int64_t n_vox, int n_classes, int64_t ldl,
                    int32_t* d_voxel_label, void* stream);
int m3d_gather_labels(const int32_t* d_voxel_label, const int64_t* d_v2p_map, int64_t n_pts,
                      int64_t* d_point_label, void* stream);

/* The same head in one pass: the backbone's last 1x1 layer (SpUNetBase.final, spconv_unet_v1m3_pdnorm.py:355-357
 * and :406), the logits against the text embeddings (clip_alignment_loss.py:138-147) and logits.max(1)[1]
 * (language_module.py:328), with the cmid-channel CLIP features kept on chip:
 *
 *   logits[v][j] = sum_m (scale[m] * (W1[m] . x[v]) + shift[m]) * text[j][m],   label[v] = first argmax_j (j < n_classes)
 *
 *   d_in_split   sf32 activations [n + 1][cin/32][2][32] bf16 (the split output of the last decoder block)
 *   d_w1         m3d_conv_pack_weight(M3D_PREC_BF16X3) of the [cmid, 1, cin] weight
 *   d_scale, d_shift [cmid] f32 or NULL (the layer's bias goes into d_shift)
 *   d_text_packed m3d_head_pack_text output ([classes_padded][cmid] in the order the kernel consumes)
 *   d_logits     [n, ldl] f32 or NULL;  d_voxel_label [n] i32 or NULL (at least one of the two)
 * Products as in M3D_PREC_BF16X3.  Supported: cin 32/64/96/128, classes_padded 128/256, cmid % 32 == 0;
 * anything else returns M3D_EINVAL (callers then chain m3d_conv_fwd_sf32 and m3d_argmax_rows). */
size_t m3d_head_text_bytes(int classes_padded, int cmid);
int m3d_head_pack_text(const float* d_text, int n_classes, int cmid, int classes_padded, void* d_out,
                       void* stream);
int m3d_head_fused_sf32(const void* d_in_split, int cin, int64_t n, const void* d_w1, const float* d_scale,
                        const float* d_shift, int cmid, const void* d_text_packed, int classes_padded,
                        int n_classes, float* d_logits, int64_t ldl, int32_t* d_voxel_label, void* stream);

/* Average pooling over a kernel map.  Replaces spconv.pytorch.ops.indice_avgpool_implicit_gemm as OpenSegment3D's
 * decoder uses it on the strided maps of the backbone
 * (mosaic3d/src/models/networks/opensegment3d/model.py:272-276, :299-307):
 *   out[o] = mean of in[pair[k][o]] over the k with pair[k][o] >= 0 (zero for a row without pairs);
 *   d_pair [k, ld] i32 (ld = column stride), d_count [n_out] i32 receives the number of pairs (may be NULL).
 */
int m3d_avgpool_fwd(const float* d_in, int c, const int32_t* d_pair, int k, int64_t ld, int64_t n_out,
                    float* d_out, int32_t* d_count, void* stream);

/* The fused head for `terms` = 2 (the entry points above) or 3 (M3D_PREC_BF16X6 products, leading and small
 * products accumulated separately): d_in_split, d_w1 (m3d_conv_pack_weight with M3D_PREC_BF16X6) and the packed text
 * are in the split format of that term count ([..][C/16][3][16] bf16 for three terms).  Same shapes supported, same
 * outputs, same reference call sites as m3d_head_fused_sf32.
 * `terms` = M3D_FMT_F16_PAIR: the two-term kernel on f16 pairs.  d_in_split and d_w1 are in the pair format of the
 * convolutions (second term scaled by 2^11; d_w1 from m3d_conv_pack_weight with M3D_PREC_F16X3, any power-of-two
 * factor of the weight undone through d_scale); the features are exact to 2^-22 and checked against the f16 range
 * (m3d_f16_status); the text matrix (m3d_head_pack_text_terms with terms = 4) and the feature tile keep an unscaled
 * second term, because the logits only feed the argmax and take a single accumulator (logits good to ~1e-5). */
size_t m3d_head_text_bytes_terms(int classes_padded, int cmid, int terms);
int m3d_head_pack_text_terms(const float* d_text, int n_classes, int cmid, int classes_padded, int terms, void* d_out,
                             void* stream);
int m3d_head_fused_terms(const void* d_in_split, int cin, int64_t n, int terms, const void* d_w1, const float* d_scale,
                         const float* d_shift, int cmid, const void* d_text_packed, int classes_padded, int n_classes,
                         float* d_logits, int64_t ldl, int32_t* d_voxel_label, void* stream);

/* ------------------------------------------------------------------------------------------
 * The two torch_scatter operators the reference's model and loss code call, on [rows, c] f32 tensors.
 * m3d_segment_csr replaces torch_scatter.segment_csr(src, indptr, reduce="mean" | "sum")
 * (mosaic3d/src/models/losses/caption_loss.py:223): out[s] = sum (mean = 0) or mean (mean = 1) of
 * src[indptr[s] : indptr[s + 1]], zero for an empty segment; rows are summed in a fixed order.
 * m3d_scatter_rows replaces torch_scatter.scatter(src, index, dim=0, out=..., reduce="mean" | "sum")
 * (mosaic3d/src/models/utils/structure.py:58, spconv_unet_v1m3_pdnorm.py:410): d_out [n_out, c] is added to
 * (callers start from zeros, or from `out`), out[index[i]] += src[i] with float atomics; with mean = 1 every row that
 * received k > 1 rows is divided by k afterwards (d_count [n_out] i32 is scratch, required for a mean). */
int m3d_segment_csr(const float* d_src, int c, const int64_t* d_indptr, int64_t n_segments, int mean, float* d_out,
                    void* stream);
int m3d_scatter_rows(const float* d_src, const int64_t* d_index, int64_t n, int c, int64_t n_out, int mean, float* d_out,
                     int32_t* d_count, void* stream);

/* ------------------------------------------------------------------------------------------
 * A whole network enqueued by one call.  Replaces the layer-by-layer walk of SpUNetBase.forward
 * (mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:380-414: conv_input, the encoder stages, the decoder
 * stages with their skip concatenations, final) and of the head (clip_alignment_loss.py:138-147,
 * language_module.py:202-207,328) for a caller that has folded the norms and packed the weights once: per scene it
 * hands over the kernel maps of its levels, the level-0 input rows and one arena, and every layer is launched from
 * inside the library through the entry points above (same kernels, same tile plans, same bits as layer-by-layer calls).
 *
 * Description (built once): values = the tensors of the network (value 0 is the input, external), layers = the
 * operations in launch order.  A layer reads in0 (and in1, the second K-slab of a concatenated input, and an f32
 * residual) and writes an f32 output and/or a split-format output.  Values live in the arena; a value's memory is
 * reused by later values of its level once its last reader has been enqueued (everything runs on one stream). */
#define M3D_NET_CONV 0           /* m3d_conv_fwd_sf32 / m3d_conv_fwd_f32, by the format of in0 */
#define M3D_NET_SPLIT 1          /* m3d_split_f32: f32 rows -> the split format of out_split */
#define M3D_NET_HEAD 2           /* m3d_head_fused_terms: last 1x1 layer + text logits + argmax */
#define M3D_NET_ARGMAX 3         /* m3d_argmax_rows over an f32 logits value (in0; its channel count is the row stride) */
#define M3D_NET_GATHER_LABELS 4  /* m3d_gather_labels: per-point labels through the v2p map */
#define M3D_NET_DENSE 0          /* schedule kinds of a convolution: none (1x1), */
#define M3D_NET_SUBM 1           /* the submanifold map of the output level, */
#define M3D_NET_DOWN 2           /* the strided map from the level above (finer) into the output level, */
#define M3D_NET_UP 3             /* its inverse, from the coarser level into the output level */
typedef struct {
  int32_t level;     /* resolution level of the rows (0 = finest) */
  int32_t channels;
  int32_t format;    /* 0 = f32 rows [rows][channels]; 1, 2, 3, 4 = the split format of that code (see m3d_conv_fwd_sf32) */
} m3d_net_value_t;
typedef struct {
  int32_t op;                      /* M3D_NET_* operation */
  int32_t in0, in1, residual;      /* value ids; in1 / residual = -1 for none */
  int32_t out_f32, out_split;      /* value ids or -1 */
  int32_t sched;                   /* M3D_NET_DENSE / SUBM / DOWN / UP (convolutions) */
  int32_t k, cout, c0, c1, relu;   /* convolution shape (c0, c1 = channels of in0, in1); HEAD: cout = feature channels */
  int32_t precision;               /* M3D_PREC_* of the packed weight (used when in0 is f32 rows) */
  int32_t n_classes, classes_padded;  /* HEAD / ARGMAX */
  const void* weight;              /* m3d_conv_pack_weight output (CONV, HEAD) */
  const float* scale;              /* [cout] or NULL */
  const float* shift;              /* [cout] or NULL */
  const void* text;                /* HEAD: m3d_head_pack_text_terms output */
} m3d_net_layer_t;
typedef struct {
  const int32_t* nbr;
  const int32_t* out_row;
  const uint32_t* tile_mask;
  const int32_t* tile_order;  /* or NULL */
  int64_t n, npad;            /* output rows, schedule length */
  int32_t k;                  /* kernel offsets of the map */
} m3d_net_sched_t;
typedef struct {
  m3d_net_sched_t subm;  /* 3x3x3 map of the level */
  m3d_net_sched_t down;  /* 2x2x2 stride-2 map of this level into the next coarser one (unused on the last level) */
  m3d_net_sched_t up;    /* its inverse: the coarser level back into this one */
} m3d_net_maps_t;
/* m3d_net_create checks the description (ids, levels, shapes, formats) and assigns arena slots; *out_net is freed by
 * m3d_net_destroy.  The pointers inside the layers (weights, scale / shift vectors, packed text) must stay valid.
 * m3d_net_arena_bytes(net, level_rows[n_levels]): bytes of arena m3d_net_run needs for a scene with these row counts
 * (it includes the largest convolution workspace).
 * m3d_net_run: maps[n_levels], level_rows[n_levels], d_input = value 0 ([level_rows[0]][channels] f32), d_arena
 * 256-byte aligned.  HEAD / ARGMAX write d_voxel_label [level_rows[0]] and, if d_logits != NULL, the per-voxel logits
 * (row stride ldl floats); GATHER_LABELS writes d_point_label [n_points] through d_v2p_map.  Nothing is allocated and
 * nothing synchronises. */
int m3d_net_create(const m3d_net_layer_t* layers, int n_layers, const m3d_net_value_t* values, int n_values, int n_levels,
                   void** out_net);
void m3d_net_destroy(void* net);
size_t m3d_net_arena_bytes(const void* net, const int64_t* level_rows);
int m3d_net_run(const void* net, const m3d_net_maps_t* maps, const int64_t* level_rows, const float* d_input, void* d_arena,
                size_t arena_bytes, const int64_t* d_v2p_map, int64_t n_points, int32_t* d_voxel_label,
                int64_t* d_point_label, float* d_logits, int64_t ldl, void* stream);

/* ------------------------------------------------------------------------------------------
 * Kernel maps and tile schedules of all resolution levels of a scene in two calls.  Replaces what SpUNetBase.forward
 * has spconv build on the first use of every indice_key (spconv_unet_v1m3_pdnorm.py:98-117 "subm{s}", :154-161
 * "spconv{s}" and its reuse by the inverse convolutions :184-190, :205-212 "stem") for a caller that wants one
 * library call per phase instead of one per map and schedule; every step goes through m3d_strided_rulebook,
 * m3d_subm_rulebook, m3d_build_schedule and m3d_schedule_tile_order in the order a level-by-level caller uses them
 * (same bits).
 *   m3d_levels_begin:  the chain of stride-2 maps (level l -> l + 1; the row count of a level stays on the device,
 *                      d_counts[l] = rows of level l + 1) and the submanifold map + schedule of level 0.  Only
 *                      levels[0].n is read; level l is processed for levels[l].ld rows at most, where ld must be a
 *                      bound of the level's rows: min(levels[0].n, batch_size * cells of the level's grid) or more,
 *                      levels[0].n at most; next_indices of level l holds ld of level l + 1 rows.
 *                      `parts` selects M3D_LEVELS_CHAIN, M3D_LEVELS_FIRST or both: the two are independent of each
 *                      other, a caller may enqueue them on two streams (each with a workspace of its own).
 *   m3d_levels_finish: after the caller has read d_counts back and filled levels[l].n: for every level l in
 *                      `level_mask` (bit l) the submanifold map + schedule of the level (l >= 1) and the schedules of
 *                      its strided map into level l + 1 and of the inverse.  Levels are independent of each other
 *                      here (level 0's part needs M3D_LEVELS_FIRST done): a caller may spread them over streams. */
#define M3D_LEVELS_CHAIN 1
#define M3D_LEVELS_FIRST 2
typedef struct {
  const int32_t* indices;   /* [n][4]; level l + 1 reads next_indices of level l */
  int64_t n;                /* rows of the level */
  int64_t shape[3];         /* spatial shape of the level */
  int32_t* next_indices;    /* [bound of level l + 1][4] coordinates of the next coarser level (unused on the last level) */
  int32_t* pair_fwd;        /* [8][ld] */
  int32_t* pair_bwd;        /* [8][ld] */
  int64_t ld;               /* column stride of pair_fwd / pair_bwd = the bound of the level's rows (see above) */
  int32_t* pair_subm;       /* [27][n] */
  int32_t subm_mode;        /* m3d_build_schedule mode of the submanifold schedule */
  struct { int32_t* nbr; int32_t* out_row; uint32_t* tile_mask; int32_t* tile_order; } subm, down, up;
  /* subm: [27][npad(n)], [npad(n)], [npad(n)/32], [npad(n)/128] or NULL; down: [8][npad(n_next)] ...; up: [8][npad(n)] ... */
} m3d_level_t;
size_t m3d_levels_workspace_bytes(const m3d_level_t* levels, int n_levels, int batch_size);
int m3d_levels_begin(const m3d_level_t* levels, int n_levels, int batch_size, int parts, int64_t* d_counts, void* d_ws,
                     size_t ws_bytes, void* stream);
int m3d_levels_finish(const m3d_level_t* levels, int n_levels, int batch_size, unsigned level_mask, void* d_ws,
                      size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M3D_H_ */
