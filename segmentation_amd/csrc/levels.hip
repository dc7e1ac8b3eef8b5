// Kernel maps and tile schedules of all resolution levels of a scene, enqueued by two calls (m3d_levels_begin /
// m3d_levels_finish) instead of one call per map and schedule: what SpUNetBase.forward makes spconv build on the
// first use of every indice_key (mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:98-117,154-161,
// 184-190,205-212).  Host code only: every step goes through the entry points of rulebook.hip, in the order a
// level-by-level caller would use them, so the maps are the same bits.
#include <algorithm>

#include "common.h"

using namespace m3d;

// Upper bound of the rows of level l before the counts are known: no more than level 0 has, and no more than the
// level's grid has cells.
static int64_t level_bound(const m3d_level_t* L, int l, int batch_size) {
  const int64_t n0 = L[0].n;
  if (l == 0 || batch_size <= 0) return n0;
  const int64_t cells = (int64_t)batch_size * L[l].shape[0] * L[l].shape[1] * L[l].shape[2];
  return std::max<int64_t>(1, std::min(n0, cells));
}

static size_t levels_ws_need(const m3d_level_t* L, int n_levels, int batch_size, bool begin) {
  size_t need = 256;
  for (int l = 0; l < n_levels; ++l) {
    const int64_t n = begin ? level_bound(L, l, batch_size) : L[l].n;
    if (begin && l + 1 < n_levels) need = std::max(need, m3d_strided_rulebook_workspace_bytes(n, L[l].shape, batch_size));
    if (begin ? l == 0 : l >= 1) {
      need = std::max(need, m3d_subm_rulebook_workspace_bytes(L[l].n, L[l].shape, batch_size));
      need = std::max(need, m3d_schedule_workspace_bytes(L[l].n));
    }
    if (!begin && l + 1 < n_levels) need = std::max(need, m3d_schedule_workspace_bytes(std::max(L[l].n, L[l + 1].n)));
  }
  return need;
}

extern "C" size_t m3d_levels_workspace_bytes(const m3d_level_t* levels, int n_levels, int batch_size) {
  if (!levels || n_levels < 1) return 256;
  m3d_level_t first = levels[0];
  // before the counts are known every level is bounded by the first one and by its own grid
  size_t need = 256;
  for (int l = 0; l < n_levels; ++l) {
    const int64_t n = level_bound(levels, l, batch_size);
    need = std::max(need, m3d_strided_rulebook_workspace_bytes(n, levels[l].shape, batch_size));
    need = std::max(need, m3d_subm_rulebook_workspace_bytes(n, levels[l].shape, batch_size));
  }
  need = std::max(need, m3d_schedule_workspace_bytes(first.n));
  return need;
}

static int subm_of_level(const m3d_level_t& v, int batch_size, void* ws, size_t ws_bytes, void* stream) {
  M3D_CHECK_ARG(v.indices && v.pair_subm && v.subm.nbr && v.subm.out_row && v.subm.tile_mask, "m3d_levels: level outputs missing");
  if (int rc = m3d_subm_rulebook(v.indices, v.n, v.shape, batch_size, v.pair_subm, ws, ws_bytes, stream)) return rc;
  if (int rc = m3d_build_schedule(v.pair_subm, 27, v.n, v.n, v.indices, v.shape, batch_size, v.subm_mode, nullptr, v.subm.nbr,
                                  v.subm.out_row, v.subm.tile_mask, ws, ws_bytes, stream))
    return rc;
  if (v.subm.tile_order)
    if (int rc = m3d_schedule_tile_order(v.subm.tile_mask, m3d_schedule_npad(v.n), v.subm.tile_order, stream)) return rc;
  return M3D_OK;
}

extern "C" int m3d_levels_begin(const m3d_level_t* L, int n_levels, int batch_size, int parts, int64_t* d_counts,
                                void* d_ws, size_t ws_bytes, void* stream) {
  const char* who = "m3d_levels_begin";
  M3D_CHECK_ARG(L && n_levels >= 1 && n_levels <= 16 && L[0].n > 0, "%s: bad arguments", who);
  M3D_CHECK_ARG(parts >= 1 && parts <= 3, "%s: parts must be M3D_LEVELS_CHAIN, M3D_LEVELS_FIRST or both", who);
  M3D_CHECK_ARG(n_levels == 1 || d_counts || !(parts & M3D_LEVELS_CHAIN), "%s: d_counts missing", who);
  const size_t need = levels_ws_need(L, n_levels, batch_size, true);
  if (d_ws == nullptr || ws_bytes < need) {
    set_error("%s: workspace too small (%zu given, %zu needed)", who, ws_bytes, need);
    return M3D_ENOSPC;
  }
  // the chain of strided maps: the row count of a level stays on the device and feeds the next one
  for (int l = 0; (parts & M3D_LEVELS_CHAIN) && l + 1 < n_levels; ++l) {
    M3D_CHECK_ARG(L[l].indices && L[l].next_indices && L[l].pair_fwd && L[l].pair_bwd, "%s: level %d: strided-map outputs missing",
                  who, l);
    M3D_CHECK_ARG(L[l].ld >= level_bound(L, l, batch_size) && L[l].ld <= L[0].n,
                  "%s: level %d: column stride %lld is not a bound of the level's rows (%lld)", who, l, (long long)L[l].ld,
                  (long long)level_bound(L, l, batch_size));
    M3D_CHECK_ARG(L[l + 1].indices == L[l].next_indices, "%s: level %d does not read the outputs of level %d", who, l + 1, l);
    if (int rc = m3d_strided_rulebook(L[l].indices, L[l].ld, l ? d_counts + (l - 1) : nullptr, L[l].shape, batch_size,
                                      L[l].next_indices, L[l].pair_fwd, L[l].pair_bwd, d_counts + l, d_ws, ws_bytes, stream))
      return rc;
  }
  // level 0 needs nothing from the chain: its map and schedule keep the GPU busy while the counts travel
  if (parts & M3D_LEVELS_FIRST) return subm_of_level(L[0], batch_size, d_ws, ws_bytes, stream);
  return M3D_OK;
}

extern "C" int m3d_levels_finish(const m3d_level_t* L, int n_levels, int batch_size, unsigned level_mask, void* d_ws,
                                 size_t ws_bytes, void* stream) {
  const char* who = "m3d_levels_finish";
  M3D_CHECK_ARG(L && n_levels >= 1 && n_levels <= 16, "%s: bad arguments", who);
  for (int l = 0; l < n_levels; ++l) M3D_CHECK_ARG(L[l].n > 0, "%s: level %d has no rows", who, l);
  const size_t need = levels_ws_need(L, n_levels, batch_size, false);
  if (d_ws == nullptr || ws_bytes < need) {
    set_error("%s: workspace too small (%zu given, %zu needed)", who, ws_bytes, need);
    return M3D_ENOSPC;
  }
  for (int l = 0; l < n_levels; ++l) {
    if (!((level_mask >> l) & 1u)) continue;
    if (l >= 1)
      if (int rc = subm_of_level(L[l], batch_size, d_ws, ws_bytes, stream)) return rc;
    if (l + 1 < n_levels) {
      const m3d_level_t& v = L[l];
      M3D_CHECK_ARG(v.down.nbr && v.down.out_row && v.down.tile_mask && v.up.nbr && v.up.out_row && v.up.tile_mask,
                    "%s: level %d: schedule outputs missing", who, l);
      // strided convolution into the coarser level: its rows in their own order
      if (int rc = m3d_build_schedule(v.pair_fwd, 8, v.ld, L[l + 1].n, nullptr, nullptr, batch_size, 0, nullptr, v.down.nbr,
                                      v.down.out_row, v.down.tile_mask, d_ws, ws_bytes, stream))
        return rc;
      // its inverse: rows grouped by their (single) kernel offset, in the order of the level's submanifold schedule
      if (int rc = m3d_build_schedule(v.pair_bwd, 8, v.ld, v.n, nullptr, nullptr, batch_size, 2, v.subm.out_row, v.up.nbr,
                                      v.up.out_row, v.up.tile_mask, d_ws, ws_bytes, stream))
        return rc;
    }
  }
  return M3D_OK;
}
