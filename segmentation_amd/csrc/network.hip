// A whole network of sparse convolutions enqueued by one call (m3d_net_*): the layer list of SpUNetBase.forward
// (mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:380-414) and the head
// (mosaic3d/src/models/losses/clip_alignment_loss.py:138-147, mosaic3d/src/models/lightning_modules/language_module.py:328)
// is described once per engine; per scene the caller hands over the kernel maps of its levels and one arena, and every
// layer is launched from here, through the same entry points a layer-by-layer caller would use (same kernels, same
// plans, same bits).  Host code only.
#include <algorithm>
#include <new>
#include <vector>

#include "common.h"

namespace m3d {

struct NetValue {
  int level, channels, fmt;  // fmt 0 = f32 rows, 1-4 = split format of that code
  int def = -1, last = -1;   // layers that write / last read the value
  int slot = -1;             // arena slot (per level); -1 = external (value 0, the network input)
};

struct Net {
  std::vector<m3d_net_layer_t> layers;
  std::vector<NetValue> values;
  int n_levels = 0;
  std::vector<std::vector<size_t>> slot_bpr;  // [level][slot] bytes per row
};

static size_t value_bpr(const NetValue& v) {
  if (v.fmt == 0) return (size_t)v.channels * 4;
  return (size_t)v.channels * 2 * (v.fmt == 4 ? 2 : v.fmt);
}

static int out_level(const m3d_net_layer_t& L, const std::vector<NetValue>& vals) {
  return vals[L.out_f32 >= 0 ? L.out_f32 : L.out_split].level;
}

}  // namespace m3d

using namespace m3d;

extern "C" int m3d_net_create(const m3d_net_layer_t* layers, int n_layers, const m3d_net_value_t* values, int n_values,
                              int n_levels, void** out_net) {
  const char* who = "m3d_net_create";
  M3D_CHECK_ARG(layers && values && out_net && n_layers > 0 && n_values > 0 && n_levels > 0 && n_levels <= 16,
                "%s: bad arguments", who);
  Net* net = new (std::nothrow) Net();
  M3D_CHECK_ARG(net != nullptr, "%s: out of memory", who);
  try {
    net->layers.assign(layers, layers + n_layers);
    net->values.resize(n_values);
    net->slot_bpr.assign(n_levels, {});
  } catch (const std::bad_alloc&) {
    delete net;
    set_error("%s: out of memory", who);
    return M3D_EINVAL;
  }
  net->n_levels = n_levels;
  for (int i = 0; i < n_values; ++i) {
    NetValue& v = net->values[i];
    v.level = values[i].level;
    v.channels = values[i].channels;
    v.fmt = values[i].format;
    if (v.level < 0 || v.level >= n_levels || v.channels <= 0 || v.fmt < 0 || v.fmt > 4) {
      set_error("%s: value %d: level %d, %d channels, format %d", who, i, v.level, v.channels, v.fmt);
      delete net;
      return M3D_EINVAL;
    }
  }
  auto bad = [&](int li, const char* what) {
    set_error("%s: layer %d: %s", who, li, what);
    delete net;
    return M3D_EINVAL;
  };
  auto ok_id = [&](int id, bool optional) { return (optional && id == -1) || (id >= 0 && id < n_values); };
  for (int li = 0; li < n_layers; ++li) {
    const m3d_net_layer_t& L = net->layers[li];
    if (L.op < M3D_NET_CONV || L.op > M3D_NET_GATHER_LABELS) return bad(li, "unknown op");
    if (!ok_id(L.in0, false) || !ok_id(L.in1, true) || !ok_id(L.residual, true) || !ok_id(L.out_f32, true) ||
        !ok_id(L.out_split, true))
      return bad(li, "value id out of range");
    for (int id : {L.in0, L.in1, L.residual}) {
      if (id < 0) continue;
      if (id != 0 && net->values[id].def < 0) return bad(li, "reads a value no earlier layer wrote");
      net->values[id].last = li;
    }
    for (int id : {L.out_f32, L.out_split}) {
      if (id < 0) continue;
      if (id == 0 || net->values[id].def >= 0) return bad(li, "a value is written twice");
      net->values[id].def = li;
      net->values[id].last = li;
    }
    if (L.op == M3D_NET_CONV) {
      if (L.out_f32 < 0 && L.out_split < 0) return bad(li, "a convolution without an output");
      if (L.sched < M3D_NET_DENSE || L.sched > M3D_NET_UP) return bad(li, "unknown schedule kind");
      if (!L.weight) return bad(li, "null weight");
      const int lo = out_level(L, net->values), li0 = net->values[L.in0].level;
      const int want = L.sched == M3D_NET_DOWN ? lo - 1 : (L.sched == M3D_NET_UP ? lo + 1 : lo);
      if (li0 != want || (L.in1 >= 0 && net->values[L.in1].level != want)) return bad(li, "input level does not fit the schedule");
      if (net->values[L.in0].channels != L.c0 || (L.in1 >= 0 ? net->values[L.in1].channels : 0) != L.c1)
        return bad(li, "channel counts of the inputs");
      if (L.in1 >= 0 && net->values[L.in1].fmt != net->values[L.in0].fmt) return bad(li, "the two inputs differ in format");
      if (L.residual >= 0 && (net->values[L.residual].fmt != 0 || net->values[L.residual].level != lo ||
                              net->values[L.residual].channels != L.cout))
        return bad(li, "the residual must be f32 rows of the output's shape");
      for (int id : {L.out_f32, L.out_split})
        if (id >= 0 && (net->values[id].channels != L.cout || net->values[id].level != lo)) return bad(li, "output shape");
      if (L.out_f32 >= 0 && net->values[L.out_f32].fmt != 0) return bad(li, "out_f32 must be f32 rows");
      if (L.out_split >= 0 && net->values[L.out_split].fmt == 0) return bad(li, "out_split must be a split format");
    }
  }
  // arena slots: per level, a value takes the first slot that is free when it is written and returns it after its
  // last reader; a slot is as wide as the widest value it ever holds
  std::vector<std::vector<int>> busy_until(n_levels);
  std::vector<int> order;
  for (int i = 1; i < n_values; ++i)
    if (net->values[i].def >= 0) order.push_back(i);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return net->values[a].def < net->values[b].def; });
  for (int id : order) {
    NetValue& v = net->values[id];
    auto& until = busy_until[v.level];
    auto& bpr = net->slot_bpr[v.level];
    int pick = -1;
    for (size_t s = 0; s < until.size(); ++s)
      if (until[s] < v.def) {  // free strictly before this layer starts: an input of the layer is never its output
        pick = (int)s;
        break;
      }
    if (pick < 0) {
      until.push_back(0);
      bpr.push_back(0);
      pick = (int)until.size() - 1;
    }
    until[pick] = v.last;
    bpr[pick] = std::max(bpr[pick], value_bpr(v));
    v.slot = pick;
  }
  *out_net = net;
  return M3D_OK;
}

extern "C" void m3d_net_destroy(void* net) { delete static_cast<Net*>(net); }

static size_t slot_bytes(size_t bpr, int64_t rows) { return align_up((size_t)(rows + 1) * bpr); }

// offsets of the slots of every level + the convolution workspace behind them; returns the total
static size_t net_layout(const Net& net, const int64_t* level_rows, std::vector<std::vector<size_t>>* offsets, size_t* ws_off,
                         size_t* ws_bytes) {
  size_t off = 0;
  if (offsets) offsets->assign(net.n_levels, {});
  for (int l = 0; l < net.n_levels; ++l)
    for (size_t s = 0; s < net.slot_bpr[l].size(); ++s) {
      if (offsets) (*offsets)[l].push_back(off);
      off += slot_bytes(net.slot_bpr[l][s], level_rows[l]);
    }
  size_t ws = 256;
  for (const m3d_net_layer_t& L : net.layers)
    if (L.op == M3D_NET_CONV) ws = std::max(ws, m3d_conv_workspace_bytes(level_rows[out_level(L, net.values)], L.cout, L.k));
  if (ws_off) *ws_off = off;
  if (ws_bytes) *ws_bytes = ws;
  return off + align_up(ws);
}

extern "C" size_t m3d_net_arena_bytes(const void* net_, const int64_t* level_rows) {
  const Net* net = static_cast<const Net*>(net_);
  if (!net || !level_rows) return 0;
  return net_layout(*net, level_rows, nullptr, nullptr, nullptr);
}

extern "C" int m3d_net_run(const void* net_, const m3d_net_maps_t* maps, const int64_t* level_rows, const float* d_input,
                           void* d_arena, size_t arena_bytes, const int64_t* d_v2p_map, int64_t n_points,
                           int32_t* d_voxel_label, int64_t* d_point_label, float* d_logits, int64_t ldl, void* stream) {
  const char* who = "m3d_net_run";
  const Net* net = static_cast<const Net*>(net_);
  M3D_CHECK_ARG(net && maps && level_rows && d_input && d_arena, "%s: null pointer", who);
  for (int l = 0; l < net->n_levels; ++l) M3D_CHECK_ARG(level_rows[l] > 0, "%s: level %d has no rows", who, l);
  std::vector<std::vector<size_t>> offs;
  size_t ws_off = 0, ws_bytes = 0;
  const size_t need = net_layout(*net, level_rows, &offs, &ws_off, &ws_bytes);
  if (arena_bytes < need) {
    set_error("%s: arena too small (%zu given, %zu needed)", who, arena_bytes, need);
    return M3D_ENOSPC;
  }
  M3D_CHECK_ARG(((uintptr_t)d_arena & 255) == 0, "%s: the arena must be 256-byte aligned", who);
  char* base = static_cast<char*>(d_arena);
  auto ptr = [&](int id) -> char* {
    if (id < 0) return nullptr;
    if (id == 0) return const_cast<char*>(reinterpret_cast<const char*>(d_input));
    const NetValue& v = net->values[id];
    return base + offs[v.level][v.slot];
  };
  auto sched_of = [&](const m3d_net_layer_t& L, int lo) -> const m3d_net_sched_t* {
    switch (L.sched) {
      case M3D_NET_SUBM: return &maps[lo].subm;
      case M3D_NET_DOWN: return &maps[lo - 1].down;
      case M3D_NET_UP: return &maps[lo].up;
      default: return nullptr;
    }
  };
  for (size_t li = 0; li < net->layers.size(); ++li) {
    const m3d_net_layer_t& L = net->layers[li];
    int rc = M3D_OK;
    if (L.op == M3D_NET_CONV) {
      const int lo = out_level(L, net->values);
      const int64_t n = level_rows[lo];
      const m3d_net_sched_t* s = sched_of(L, lo);
      if (s && (s->n != n || s->k != L.k || !s->nbr)) {
        set_error("%s: layer %zu: the schedule covers %lld rows with %d offsets, the layer has %lld rows and %d offsets", who,
                  li, (long long)s->n, s->k, (long long)n, L.k);
        return M3D_EINVAL;
      }
      const NetValue& vin = net->values[L.in0];
      const float* res = reinterpret_cast<const float*>(ptr(L.residual));
      float* of = reinterpret_cast<float*>(ptr(L.out_f32));
      void* os = ptr(L.out_split);
      const int oterms = L.out_split >= 0 ? net->values[L.out_split].fmt : 0;
      if (vin.fmt == 0) {
        const int64_t rows_in = level_rows[vin.level];
        rc = m3d_conv_fwd_f32(reinterpret_cast<const float*>(ptr(L.in0)), L.c0, rows_in, reinterpret_cast<const float*>(ptr(L.in1)),
                              L.c1, L.in1 >= 0 ? rows_in : 0, static_cast<const float*>(L.weight), L.k, L.cout, s ? s->nbr : nullptr,
                              s ? s->out_row : nullptr, s ? s->tile_mask : nullptr, n, s ? s->npad : 0,
                              s ? s->tile_order : nullptr, L.scale, L.shift, res, L.relu, of, L.cout, os, oterms, L.precision,
                              base + ws_off, ws_bytes, stream);
      } else {
        const int64_t rows_in = level_rows[vin.level];
        rc = m3d_conv_fwd_sf32(ptr(L.in0), L.c0, rows_in, ptr(L.in1), L.c1, L.in1 >= 0 ? rows_in : 0, L.weight, L.k, L.cout,
                               s ? s->nbr : nullptr, s ? s->out_row : nullptr, s ? s->tile_mask : nullptr, n,
                               s ? s->npad : 0, s ? s->tile_order : nullptr, L.scale,
                               L.shift, res, L.relu, of, L.cout, os, vin.fmt, base + ws_off, ws_bytes, stream);
      }
    } else if (L.op == M3D_NET_SPLIT) {
      const NetValue& vin = net->values[L.in0];
      M3D_CHECK_ARG(L.out_split >= 0 && vin.fmt == 0, "%s: layer %zu: split takes f32 rows", who, li);
      rc = m3d_split_f32(reinterpret_cast<const float*>(ptr(L.in0)), level_rows[vin.level], vin.channels,
                         net->values[L.out_split].fmt, ptr(L.out_split), stream);
    } else if (L.op == M3D_NET_HEAD) {
      const NetValue& vin = net->values[L.in0];
      M3D_CHECK_ARG(d_voxel_label, "%s: the head needs d_voxel_label", who);
      rc = m3d_head_fused_terms(ptr(L.in0), vin.channels, level_rows[vin.level], vin.fmt, L.weight, L.scale, L.shift, L.cout,
                                L.text, L.classes_padded, L.n_classes, d_logits, ldl, d_voxel_label, stream);
    } else if (L.op == M3D_NET_ARGMAX) {
      const NetValue& vin = net->values[L.in0];
      M3D_CHECK_ARG(d_voxel_label && vin.fmt == 0, "%s: layer %zu: argmax takes f32 logits and needs d_voxel_label", who, li);
      M3D_CHECK_ARG(L.n_classes > 0 && L.n_classes <= vin.channels, "%s: layer %zu: %d classes in rows of %d logits", who, li,
                    L.n_classes, vin.channels);
      M3D_CHECK_ARG(!d_logits || ldl >= L.n_classes, "%s: logits row stride %lld < %d classes", who, (long long)ldl, L.n_classes);
      rc = m3d_argmax_rows(reinterpret_cast<const float*>(ptr(L.in0)), level_rows[vin.level], L.n_classes, vin.channels,
                           d_voxel_label, stream);
      if (rc == M3D_OK && d_logits) {  // as many columns as the caller's rows hold (the padding columns if there is room)
        const size_t width = (size_t)std::min<int64_t>(ldl, vin.channels) * 4;
        M3D_HIP(hipMemcpy2DAsync(d_logits, (size_t)ldl * 4, ptr(L.in0), (size_t)vin.channels * 4, width,
                                 (size_t)level_rows[vin.level], hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
      }
    } else if (L.op == M3D_NET_GATHER_LABELS) {
      M3D_CHECK_ARG(d_voxel_label && d_v2p_map && d_point_label && n_points > 0, "%s: label gather needs v2p map and outputs", who);
      rc = m3d_gather_labels(d_voxel_label, d_v2p_map, n_points, d_point_label, stream);
    }
    if (rc != M3D_OK) return rc;
  }
  return M3D_OK;
}
