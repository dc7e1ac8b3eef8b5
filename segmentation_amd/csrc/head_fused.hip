// Open-vocabulary head in one kernel: the last 1x1 layer (cin -> cmid CLIP channels), the logits against the
// text embeddings (cmid -> classes) and the row argmax.  The cmid-channel features never leave the CU.
//
// One workgroup = NW waves x 32 voxel rows (NW = 8 where the DMA slot count divides by 512 threads, else 4).  The
// reduction over the cmid features runs in tiles of 32 features ("stages"); per stage the workgroup streams by
// LDS-DMA
//     W1 tile [32 features][cin]        (A operand of the first product)
//     W2 tile [classes][32 features]    (B operand of the second product)
// through a ring of three LDS buffers, and every wave
//   A. forms F^T = W1 . X^T for its 32 rows (32x32x16 bf16 MFMA, operands swapped: M = feature, N = row), with the
//      activation fragments X held in registers for the whole kernel; applies the per-feature affine and splits
//      the f32 values into two bf16 terms IN REGISTERS - the accumulator of the swapped product (lane = row,
//      registers = features) is, up to a fixed permutation of the features inside a 16-wide reduction step, the
//      A-operand layout of the next MFMA; the text matrix is packed with that permutation (m3d_head_pack_text), so
//      no transpose through LDS is needed,
//   B. accumulates logits += F . W2 (M = row, N = class).
// LDS allows one workgroup per CU, so the grid comes in rounds of one workgroup per CU.  The rows that would make
// up a last, partly filled round are dealt in smaller pieces (32 * tail_waves rows per workgroup, the other waves
// only help with the DMA), which costs about half a round instead of a whole one.
// Products follow the split-bf16 scheme of the convolution kernels: (hi,lo) + (lo,hi) + (hi,hi), f32 accumulate.
// Epilogue: optional f32 logits, and the first index of the row maximum over the real classes.
#include "conv_shared.h"

#include <type_traits>

namespace m3d {

struct HeadArgs {
  const char* x;        // sf32 [n + 1][cin/32][2][32] bf16, row n zero
  const char* w1;       // packed [cmid][cin/32][2][32] bf16
  const char* w2;       // packed [classes_padded][cmid/32][2][32] bf16, features permuted inside each group of 32
  const float* scale;   // [cmid] or nullptr
  const float* shift;   // [cmid] or nullptr
  float* logits;        // [n][ldl] or nullptr
  int64_t ldl;
  int32_t* label;       // [n] or nullptr
  int64_t n;
  int cmid, n_classes;
  int full_blocks, tail_waves;  // workgroups below full_blocks take 32 * NW rows, the others 32 * tail_waves
  int* f16_status;              // f16 pair kernels: set when a feature leaves the f16 range
};

constexpr int kHeadMaxFeatures = 1024;  // cmid limit of the fused kernel (scale/shift staging)

// packed position p of a 32-feature group <-> feature channel
__host__ __device__ inline int head_perm_channel(int p) {
  const int ks = p >> 4, hh = (p >> 3) & 1, i = p & 7;
  return 16 * ks + 4 * hh + (i < 4 ? i : i + 4);
}

// text [n_classes][cmid] f32 -> [classes_padded][cmid/32][2][32] bf16 (rows >= n_classes zero)
__global__ void head_pack_text_kernel(const float* __restrict__ text, int n_classes, int cmid, int classes_padded,
                                      __bf16* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)classes_padded * cmid) return;
  const int j = (int)(e / cmid), q = (int)(e % cmid);
  const int grp = q >> 5, p = q & 31;
  float v = j < n_classes ? text[(int64_t)j * cmid + grp * 32 + head_perm_channel(p)] : 0.f;
  __bf16* dst = out + ((int64_t)j * (cmid >> 5) + grp) * 64 + p;
  const __bf16 hi = (__bf16)v;
  dst[0] = hi;
  dst[32] = (__bf16)(v - (float)hi);
}

// The same for the f16 pair kernel: [classes_padded][cmid/32][2][32] f16, second term NOT scaled (the logits only
// feed the argmax and take one accumulator; text rows are unit vectors, far inside the f16 range)
__global__ void head_pack_text_f16_kernel(const float* __restrict__ text, int n_classes, int cmid, int classes_padded,
                                          _Float16* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)classes_padded * cmid) return;
  const int j = (int)(e / cmid), q = (int)(e % cmid);
  const int grp = q >> 5, p = q & 31;
  float v = j < n_classes ? text[(int64_t)j * cmid + grp * 32 + head_perm_channel(p)] : 0.f;
  _Float16* dst = out + ((int64_t)j * (cmid >> 5) + grp) * 64 + p;
  const _Float16 hi = (_Float16)v;
  dst[0] = hi;
  dst[32] = (_Float16)(v - (float)hi);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool F16, class V>
__device__ __forceinline__ floatx16 head_mfma(const V& a, const V& b, const floatx16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS read issued outside the compiler's view (it neither schedules it late nor guards it with a wait on the
// LDS-DMA traffic); the value is only valid after an s_waitcnt that lists the destination as an operand.
template <int OFF, class V>
__device__ __forceinline__ void lds_read16(V& dst, uint32_t addr) {
  static_assert(sizeof(V) == 16 && OFF >= 0 && OFF < 65536, "ds_read_b128 with an immediate offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// F16: operands are f16 pairs.  x and W1 carry the scaled second term of the convolution format (x 2^11), so the
// features accumulate leading and second-term products apart; the feature tile and the text matrix are split without
// the scale and the logits take one accumulator.
template <int G1, int NT2, int NW, bool F16>
__device__ __forceinline__ void head_fused_body(const HeadArgs& a) {
  using G = DmaGeom<2>;
  using V8 = typename std::conditional<F16, halfx8, bf16x8>::type;
  constexpr int NBUF = 3, NTHR = 64 * NW, ROWS = 32 * NW;
  constexpr int W1_BYTES = 32 * G1 * G::ROWB, W2_BYTES = NT2 * 32 * G::ROWB, BUF = W1_BYTES + W2_BYTES;
  constexpr int W1_SLOTS = 32 * G1 * G::SLOTS, SLOTS = W1_SLOTS + NT2 * 32 * G::SLOTS;
  constexpr int GI = SLOTS / NTHR;  // DMA instructions per thread and stage
  static_assert(SLOTS % NTHR == 0, "every wave must issue the same DMA count");
  static_assert(GI < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __align__(16) char ldsb[];
  // per-feature scale and shift; an array of its own, so that its reads are known not to touch the DMA buffers
  __shared__ __align__(16) float aff_s[2 * kHeadMaxFeatures];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)ldsb;
  const uint32_t aff_lds = (uint32_t)(uintptr_t)(lds_ptr_t)aff_s + (uint32_t)(4 * h) * 4u;
  const bool tail = (int)blockIdx.x >= a.full_blocks;
  const int64_t tile_base = tail ? (int64_t)a.full_blocks * ROWS + (int64_t)((int)blockIdx.x - a.full_blocks) * (32 * a.tail_waves)
                                 : (int64_t)blockIdx.x * ROWS;
  const bool active = !tail || wave < a.tail_waves;  // inactive waves only take part in the DMA and the barriers
  const int nstage = a.cmid >> 5;

  // activation fragments of this wave's 32 rows (rows past the end read the zero row)
  V8 xf[G1][2][2];
  {
    const int64_t row = tile_base + wave * 32 + r32;
    const char* xp = a.x + (active && row < a.n ? row : a.n) * (int64_t)(G1 * G::ROWB);
#pragma unroll
    for (int g = 0; g < G1; ++g)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          xf[g][ks][t] = *reinterpret_cast<const V8*>(xp + g * G::ROWB + t * 64 + (ks * 2 + h) * 16);
  }
  for (int i = tid; i < a.cmid; i += NTHR) {
    aff_s[i] = a.scale ? a.scale[i] : 1.f;
    aff_s[kHeadMaxFeatures + i] = a.shift ? a.shift[i] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // DMA addressing (loop invariant).  The W1 image (rows: group * 32 + feature) and the W2 image (rows: class) sit
  // back to back in a buffer and are filled as one list of 16-byte slots: slot e = v * NTHR + tid is (image row
  // e / 8, physical slot e % 8) and carries logical piece (e % 8) ^ swz(row).  Per stage the W1 source advances by
  // one 32-feature tile, the W2 source by one 32-feature group.
  const char* src0[GI];
  uint32_t step[GI];
#pragma unroll
  for (int v = 0; v < GI; ++v) {
    const int e = v * NTHR + tid;
    if (e < W1_SLOTS) {
      const int rr = e / G::SLOTS, g = rr >> 5, m = rr & 31;
      const int p = (e % G::SLOTS) ^ G::swz(rr);
      src0[v] = a.w1 + ((m * G1 + g) * G::ROWB + p * 16);
      step[v] = 32 * G1 * G::ROWB;
    } else {
      const int e2 = e - W1_SLOTS, j = e2 / G::SLOTS;
      const int p = (e2 % G::SLOTS) ^ G::swz(j);
      src0[v] = a.w2 + ((int64_t)j * nstage * G::ROWB + p * 16);
      step[v] = G::ROWB;
    }
  }
  auto issue = [&](int s, int bi) {
    char* buf = ldsb + bi * BUF;
#pragma unroll
    for (int v = 0; v < GI; ++v)
      __builtin_amdgcn_global_load_lds(src0[v] + (uint32_t)s * step[v], (lds_ptr_t)(buf + (v * NTHR + wave * 64) * 16),
                                       16, 0, 0);
  };

  // fragment read offsets: piece (t * 4 + ks * 2 + h) of image row (.. + r32)
  uint32_t frag[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t) frag[ks][t] = r32 * G::ROWB + (((t * 4 + ks * 2 + h) ^ G::swz(r32)) * 16);

  floatx16 acc[NT2];
#pragma unroll
  for (int i = 0; i < NT2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  // Fragment pipeline of one product: NSTEPS steps of (2 LDS reads, 3 MFMA); the reads are issued by hand two steps
  // ahead of the products that consume them and waited for with counted lgkmcnt (LDS reads complete in order, so
  // "at most 4 outstanding" after requesting step i + 2 means step i has arrived).  body(step, hi, lo) issues the
  // three products of a step.
  auto fragment_pipeline = [&](auto nsteps_c, auto base_off_c, uint32_t sbuf, auto&& body) {
    constexpr int NSTEPS = decltype(nsteps_c)::value, BASE = decltype(base_off_c)::value;
    const uint32_t p00 = sbuf + frag[0][0], p01 = sbuf + frag[0][1], p10 = sbuf + frag[1][0], p11 = sbuf + frag[1][1];
    V8 bh[3], bl[3];
    auto request = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < NSTEPS) {
        constexpr int off = BASE + (i >> 1) * 32 * G::ROWB;
        if constexpr ((i & 1) == 0) {
          lds_read16<off>(bh[i % 3], p00);
          lds_read16<off>(bl[i % 3], p01);
        } else {
          lds_read16<off>(bh[i % 3], p10);
          lds_read16<off>(bl[i % 3], p11);
        }
      }
    };
    request(std::integral_constant<int, 0>{});
    request(std::integral_constant<int, 1>{});
    static_for<0, NSTEPS>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      request(std::integral_constant<int, i + 2>{});
      if constexpr (i + 2 < NSTEPS) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[i % 3]), "+v"(bl[i % 3]));
      else if constexpr (i + 1 < NSTEPS) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[i % 3]), "+v"(bl[i % 3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[i % 3]), "+v"(bl[i % 3]));
      body(ic, bh[i % 3], bl[i % 3]);
    });
  };

  V8 ah[2], al[2];  // the current feature tile as A operand: [reduction half] hi / lo terms

  // A(s): F^T tile of features 32 s .. 32 s + 31 (features x rows), affine, split into ah / al
  const bool has_scale = a.scale != nullptr;
  auto stage_a = [&](int s) {
    // register r of f holds feature (r & 3) + 8 * (r >> 2) + 4 * h of the tile
    const uint32_t ap = aff_lds + (uint32_t)(s * 32) * 4u;
    floatx16 f;
    if (has_scale) {
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = 0.f;
    } else {  // shift only (a bias): the accumulator starts from it, nothing is left to do after the products
      f32x4 sh[4];
      lds_read16<kHeadMaxFeatures * 4>(sh[0], ap);
      lds_read16<kHeadMaxFeatures * 4 + 32>(sh[1], ap);
      lds_read16<kHeadMaxFeatures * 4 + 64>(sh[2], ap);
      lds_read16<kHeadMaxFeatures * 4 + 96>(sh[3], ap);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sh[0]), "+v"(sh[1]), "+v"(sh[2]), "+v"(sh[3]));
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = sh[j >> 2][j & 3];
    }
    floatx16 fs;  // F16: products with one scaled second term
    if constexpr (F16) {
#pragma unroll
      for (int j = 0; j < 16; ++j) fs[j] = 0.f;
    }
    fragment_pipeline(std::integral_constant<int, 2 * G1>{}, std::integral_constant<int, 0>{},
                      lds_base + (uint32_t)((s % NBUF) * BUF), [&](auto ic, const V8& wh, const V8& wl) {
                        constexpr int i = decltype(ic)::value, g = i >> 1, ks = i & 1;
                        if constexpr (F16) {
                          fs = head_mfma<F16>(wl, xf[g][ks][0], fs);
                          fs = head_mfma<F16>(wh, xf[g][ks][1], fs);
                        } else {
                          f = head_mfma<F16>(wl, xf[g][ks][0], f);
                          f = head_mfma<F16>(wh, xf[g][ks][1], f);
                        }
                        f = head_mfma<F16>(wh, xf[g][ks][0], f);
                      });
    if constexpr (F16) {
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] += fs[j] * kF16Unscale;
    }
    if (has_scale) {
      static_for<0, 4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        f32x4 sc, sh;
        lds_read16<q * 32>(sc, ap);
        lds_read16<kHeadMaxFeatures * 4 + q * 32>(sh, ap);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh));
#pragma unroll
        for (int i = 0; i < 4; ++i) f[4 * q + i] = f[4 * q + i] * sc[i] + sh[i];
      });
    }
    if constexpr (F16) {
      bool over = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) {  // split into the A operand of the second product (second term not scaled)
        const float v = f[j];
        over |= !(fabsf(v) <= kF16Max);
        const _Float16 hi = (_Float16)v;
        ah[j >> 3][j & 7] = hi;
        al[j >> 3][j & 7] = (_Float16)(v - (float)hi);
      }
      if (over && a.f16_status) *a.f16_status = 1;
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {  // split into the A operand of the second product
        const float v = f[j];
        const __bf16 hi = (__bf16)v;
        ah[j >> 3][j & 7] = hi;
        al[j >> 3][j & 7] = (__bf16)(v - (float)hi);
      }
    }
  };
  // B(s): logits += F tile . W2 tile s
  auto stage_b = [&](int s) {
    fragment_pipeline(std::integral_constant<int, 2 * NT2>{}, std::integral_constant<int, W1_BYTES>{},
                      lds_base + (uint32_t)((s % NBUF) * BUF), [&](auto ic, const V8& bh, const V8& bl) {
                        constexpr int i = decltype(ic)::value, nt = i >> 1, ks = i & 1;
                        acc[nt] = head_mfma<F16>(ah[ks], bl, acc[nt]);
                        acc[nt] = head_mfma<F16>(al[ks], bh, acc[nt]);
                        acc[nt] = head_mfma<F16>(ah[ks], bh, acc[nt]);
                      });
  };

  // Stage loop.  sync(k), once per stage and wave: tile k has landed for every wave, and every wave is done with
  // tile k - 2, whose buffer may then receive tile k + 1.  The lower half of the waves syncs between stages
  // (.. A(s) B(s) | A(s+1) ..), the upper half between the two products (.. A(s) | B(s) A(s+1) | ..): on every SIMD
  // one wave is in the MFMA-dense product B while its partner is in A (short dependent MFMA chain, VALU split) or
  // issuing its share of the next tile's DMA - the lower half right after the barrier, the upper half after its B.
  auto sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  const bool rotated = NW == 8 && wave >= 4;
  issue(0, 0);
  sync();
  if (nstage > 1) issue(1, 1);
  for (int s = 0; s < nstage; ++s) {
    if (active) stage_a(s);
    if (rotated) sync();  // tile s + 1 has landed
    if (active) stage_b(s);
    if (!rotated) sync();
    if (s + 2 < nstage) issue(s + 2, (s + 2) % NBUF);
  }
  if (!active) return;

  // epilogue: accumulator register r is row (r & 3) + 8 * (r >> 2) + 4 * h, class nt * 32 + r32
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = tile_base + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const bool live = row < a.n;
    if (a.logits && live) {
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        const int c = nt * 32 + r32;
        if (c < a.ldl) a.logits[row * a.ldl + c] = acc[nt][r];
      }
    }
    if (a.label) {
      // first index of the maximum: the first real class is taken as is, later ones only when strictly greater
      float best = 0.f;
      int bi = 0x7FFFFFFF;
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        const int c = nt * 32 + r32;
        const float v = acc[nt][r];
        if (c < a.n_classes && (bi == 0x7FFFFFFF || v > best)) {
          best = v;
          bi = c;
        }
      }
#pragma unroll
      for (int sft = 1; sft < 32; sft <<= 1) {
        const float ov = __shfl_xor(best, sft);
        const int oi = __shfl_xor(bi, sft);
        if (oi != 0x7FFFFFFF && (bi == 0x7FFFFFFF || ov > best || (ov == best && oi < bi))) {
          best = ov;
          bi = oi;
        }
      }
      if (live && r32 == 0) a.label[row] = bi;
    }
  }
}

template <int G1, int NT2, int NW, bool F16 = false>
__global__ __launch_bounds__(64 * NW) void head_fused_kernel(HeadArgs a) {
  head_fused_body<G1, NT2, NW, F16>(a);
}

template <int G1, int NT2, int NW, bool F16 = false>
static int launch_head(const HeadArgs& a, hipStream_t stream) {
  constexpr int BUF = (32 * G1 + 32 * NT2) * DmaGeom<2>::ROWB;
  const size_t lds = 3 * (size_t)BUF;
  static LdsGrant grant;
  if (int rc = grant.ensure(reinterpret_cast<const void*>(&head_fused_kernel<G1, NT2, NW, F16>), lds)) return rc;
  // rounds of one workgroup per CU; what is left after the full rounds goes out in pieces small enough for one round
  static const int ncu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      v = 256;
    return v > 0 ? v : 256;
  }();
  constexpr int ROWS = 32 * NW;
  HeadArgs b = a;
  const int64_t blocks = cdiv(a.n, ROWS);
  b.full_blocks = (int)(blocks / ncu * ncu);
  b.tail_waves = NW;
  const int64_t rest = a.n - (int64_t)b.full_blocks * ROWS;
  if (rest > 0) {
    const int64_t k = cdiv(rest, (int64_t)32 * ncu);  // waves per workgroup that cover the rest in one round
    if (k < NW) b.tail_waves = (int)k;
  }
  const int64_t tail_blocks = rest > 0 ? cdiv(rest, (int64_t)32 * b.tail_waves) : 0;
  hipLaunchKernelGGL((head_fused_kernel<G1, NT2, NW, F16>), dim3((unsigned)(b.full_blocks + tail_blocks)), dim3(64 * NW),
                     lds, stream, b);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}


// ------------------------------------------------------------------------------------------
// The same head for three-term operands (M3D_PREC_BF16X6: f32-level products).  Activations, W1 and the text matrix
// arrive in the three-term split format with 16-channel groups ([..][C/16][3][16] bf16), so a stage of 32 features
// is, per feature, G1 = cin/16 rows of 96 bytes of W1 and, per class, two rows of 96 bytes of the text matrix; LDS
// rows are 96 bytes with the pair swizzle of DmaGeom<3>.  Eight waves per workgroup in two roles: waves 0-3 own 32
// voxel rows each and do nothing but read fragments and multiply (activation fragments, logits accumulators and
// operand terms need about 250 registers); waves 4-7 do nothing but request the next stage's 61 KB (15 DMA
// instructions each).  A wave issues in order and a DMA instruction holds it for 100-200 cycles, so with both jobs in
// one wave the matrix pipe of its SIMD stood idle for a third of every stage; apart, the requests of stage s + 1
// (about 2 500 cycles of issue per loader wave) hide under the 120 matrix instructions (3 840 cycles) of stage s.
// Two LDS buffers, one barrier per stage: a loader waits for its own requests of stage s (vmcnt) before the barrier
// that publishes them, and requests stage s + 1 right after it, when every matrix wave has left stage s - 1.  The
// features accumulate leading and small products separately, as the convolution kernel does; the logits, which only
// feed the argmax, use one accumulator per class tile.
// The loader role of head_fused3_body: waves 4-7 of the workgroup (`wave` = 0..3 among them, tid = wave * 64 + lane).
template <int G1, int NT2>
__device__ __forceinline__ void head_fused3_loader(const HeadArgs& a, char* ldsb, int wave, int tid, int nstage, int ngrp) {
  using G = DmaGeom<3>;
  constexpr int NTHR = 256;
  constexpr int W1_ROWS = 32 * G1, W2_ROWS = 2 * NT2 * 32;
  constexpr int BUF = (W1_ROWS + W2_ROWS) * G::ROWB;
  constexpr int W1_SLOTS = W1_ROWS * G::SLOTS, SLOTS = W1_SLOTS + W2_ROWS * G::SLOTS;
  constexpr int GI = (SLOTS + NTHR - 1) / NTHR;
  // DMA addressing (loop invariant): slot e = v * NTHR + tid of a buffer is (image row e / 6, physical slot e % 6)
  // and carries logical piece (e % 6) ^ swz(row).  W1 image rows: group * 32 + feature; text image rows:
  // (k-step * NT2 + class tile) * 32 + class.
  const char* src0[GI];
  uint32_t step[GI];
  bool on[GI];
#pragma unroll
  for (int v = 0; v < GI; ++v) {
    const int e = v * NTHR + tid;
    on[v] = e < SLOTS;
    if (e < W1_SLOTS) {
      const int rr = e / G::SLOTS, g = rr >> 5, m = rr & 31;
      const int p = (e % G::SLOTS) ^ G::swz(rr);
      src0[v] = a.w1 + ((m * G1 + g) * 96 + p * 16);
      step[v] = 32 * G1 * 96;
    } else {
      const int e2 = (e < SLOTS ? e : W1_SLOTS) - W1_SLOTS, rr = e2 / G::SLOTS;
      const int ks = rr / (NT2 * 32), j = rr % (NT2 * 32);
      const int p = (e2 % G::SLOTS) ^ G::swz(rr);
      src0[v] = a.w2 + (((int64_t)j * ngrp + ks) * 96 + p * 16);
      step[v] = 2 * 96;
    }
  }
  auto issue = [&](int s, int bi) {
    char* buf = ldsb + bi * BUF;
#pragma unroll
    for (int v = 0; v < GI; ++v)
      if (v * NTHR + wave * 64 < SLOTS) {  // wave-uniform
        if (on[v])
          __builtin_amdgcn_global_load_lds(src0[v] + (uint32_t)s * step[v], (lds_ptr_t)(buf + (v * NTHR + wave * 64) * 16),
                                           16, 0, 0);
      }
  };

  issue(0, 0);
  for (int s = 0; s < nstage; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage s is in LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 1 < nstage) issue(s + 1, (s + 1) & 1);
  }
}

template <int G1, int NT2>
__device__ __forceinline__ void head_fused3_body(const HeadArgs& a) {
  using G = DmaGeom<3>;
  constexpr int NTHR = 256, ROWS = 128;  // NTHR: threads of one role (four waves)
  constexpr int W1_ROWS = 32 * G1, W2_ROWS = 2 * NT2 * 32;
  constexpr int W1_BYTES = W1_ROWS * G::ROWB, BUF = (W1_ROWS + W2_ROWS) * G::ROWB;
  static_assert(W1_BYTES % 256 == 0 && BUF % 256 == 0, "images start on a bank line");
  extern __shared__ __align__(16) char ldsb[];
  float* aff_s = reinterpret_cast<float*>(ldsb + 2 * BUF);  // [2][kHeadMaxFeatures]: scale, shift

  const int lane = threadIdx.x & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool loader = wave_all >= 4;
  const int wave = wave_all & 3;          // index inside the role
  const int tid = wave * 64 + lane;
  const int r32 = lane & 31, h = lane >> 5;
  const int64_t tile_base = (int64_t)blockIdx.x * ROWS;
  const int nstage = a.cmid >> 5;
  const int ngrp = a.cmid >> 4;  // 16-feature groups of the text matrix
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)ldsb;

  for (int i = threadIdx.x; i < a.cmid; i += 2 * NTHR) {  // published by the first stage barrier
    aff_s[i] = a.scale ? a.scale[i] : 1.f;
    aff_s[kHeadMaxFeatures + i] = a.shift ? a.shift[i] : 0.f;
  }

  if (loader) {
    head_fused3_loader<G1, NT2>(a, ldsb, wave, tid, nstage, ngrp);
    return;  // nothing after the stage loop needs a barrier
  }

  // activation fragments of this wave's 32 rows: [channel group][term], 8 channels per lane
  bf16x8 xf[G1][3];
  {
    const int64_t row = tile_base + wave * 32 + r32;
    const char* xp = a.x + (row < a.n ? row : a.n) * (int64_t)(G1 * 96);
#pragma unroll
    for (int g = 0; g < G1; ++g)
#pragma unroll
      for (int t = 0; t < 3; ++t) xf[g][t] = *reinterpret_cast<const bf16x8*>(xp + g * 96 + t * 32 + h * 16);
  }

  // fragment read offsets inside an image: piece 2 t + h of row (.. + r32); the swizzle depends on r32 only because
  // every image row base is a multiple of 32 rows
  uint32_t frag[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) frag[t] = r32 * G::ROWB + (((2 * t + h) ^ G::swz(r32)) * 16);

  floatx16 acc[NT2];
#pragma unroll
  for (int i = 0; i < NT2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  for (int s = 0; s < nstage; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the activation fragments (first stage only)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // stage s has landed (the loaders waited for it); every wave is done with stage s - 1
    asm volatile("" ::: "memory");

    // A: F^T tile (features x rows) = W1 tile . X^T, six products per channel group.  The three fragment reads of a
    // step are issued by hand (lds_read16) one step ahead of the products that consume them and waited for with a
    // counted lgkmcnt: LDS reads return in order, so "at most 3 outstanding" after requesting step g + 1 means that
    // step g has arrived.
    const uint32_t sb = lds_base + (uint32_t)((s & 1) * BUF);
    const uint32_t p0 = sb + frag[0], p1 = sb + frag[1], p2 = sb + frag[2];
    floatx16 fh, fl;
#pragma unroll
    for (int j = 0; j < 16; ++j) fh[j] = fl[j] = 0.f;
    {
      bf16x8 w[2][3];
      lds_read16<0>(w[0][0], p0);
      lds_read16<0>(w[0][1], p1);
      lds_read16<0>(w[0][2], p2);
      static_for<0, G1>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + 1 < G1) {
          lds_read16<(g + 1) * 32 * G::ROWB>(w[(g + 1) & 1][0], p0);
          lds_read16<(g + 1) * 32 * G::ROWB>(w[(g + 1) & 1][1], p1);
          lds_read16<(g + 1) * 32 * G::ROWB>(w[(g + 1) & 1][2], p2);
          asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w[g & 1][0]), "+v"(w[g & 1][1]), "+v"(w[g & 1][2]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[g & 1][0]), "+v"(w[g & 1][1]), "+v"(w[g & 1][2]));
        }
        fl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][0], xf[g][2], fl, 0, 0, 0);
        fl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][2], xf[g][0], fl, 0, 0, 0);
        fl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][1], xf[g][1], fl, 0, 0, 0);
        fl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][0], xf[g][1], fl, 0, 0, 0);
        fl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][1], xf[g][0], fl, 0, 0, 0);
        fh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][0], xf[g][0], fh, 0, 0, 0);
      });
    }
    // register j holds feature (j & 3) + 8 * (j >> 2) + 4 * h of the tile: affine, then three bf16 terms per k-step
    bf16x8 at[2][3];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ft = s * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      float v = (fh[j] + fl[j]) * aff_s[ft] + aff_s[kHeadMaxFeatures + ft];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const __bf16 q = (__bf16)v;
        at[j >> 3][t][j & 7] = q;
        v -= (float)q;
      }
    }
    // B: logits += F tile . text tile (one accumulator per class tile: the logits only feed the argmax); same
    // hand-issued fragment pipeline.  Step i = k-step * NT2 + class tile: consecutive steps read consecutive rows.
    {
      bf16x8 bq[2][3];
      const uint32_t q0 = p0 + W1_BYTES, q1 = p1 + W1_BYTES, q2 = p2 + W1_BYTES;
      lds_read16<0>(bq[0][0], q0);
      lds_read16<0>(bq[0][1], q1);
      lds_read16<0>(bq[0][2], q2);
      static_for<0, 2 * NT2>([&](auto ic) {
        constexpr int i = decltype(ic)::value, ks = i / NT2, nt = i % NT2;
        if constexpr (i + 1 < 2 * NT2) {
          lds_read16<(i + 1) * 32 * G::ROWB>(bq[(i + 1) & 1][0], q0);
          lds_read16<(i + 1) * 32 * G::ROWB>(bq[(i + 1) & 1][1], q1);
          lds_read16<(i + 1) * 32 * G::ROWB>(bq[(i + 1) & 1][2], q2);
          asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bq[i & 1][0]), "+v"(bq[i & 1][1]), "+v"(bq[i & 1][2]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[i & 1][0]), "+v"(bq[i & 1][1]), "+v"(bq[i & 1][2]));
        }
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[ks][0], bq[i & 1][2], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[ks][2], bq[i & 1][0], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[ks][1], bq[i & 1][1], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[ks][0], bq[i & 1][1], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[ks][1], bq[i & 1][0], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[ks][0], bq[i & 1][0], acc[nt], 0, 0, 0);
      });
    }
  }

  // epilogue: accumulator register r is row (r & 3) + 8 * (r >> 2) + 4 * h, class nt * 32 + r32
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = tile_base + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const bool live = row < a.n;
    float best = 0.f;
    int bi = 0x7FFFFFFF;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      const int c = nt * 32 + r32;
      const float v = acc[nt][r];
      if (a.logits && live && c < a.ldl) a.logits[row * a.ldl + c] = v;
      // first index of the maximum: the first real class is taken as is, later ones only when strictly greater
      if (c < a.n_classes && (bi == 0x7FFFFFFF || v > best)) {
        best = v;
        bi = c;
      }
    }
    if (a.label) {
#pragma unroll
      for (int sft = 1; sft < 32; sft <<= 1) {
        const float ov = __shfl_xor(best, sft);
        const int oi = __shfl_xor(bi, sft);
        if (oi != 0x7FFFFFFF && (bi == 0x7FFFFFFF || ov > best || (ov == best && oi < bi))) {
          best = ov;
          bi = oi;
        }
      }
      if (live && r32 == 0) a.label[row] = bi;
    }
  }
}

template <int G1, int NT2>
__global__ __launch_bounds__(512) void head_fused3_kernel(HeadArgs a) {
  head_fused3_body<G1, NT2>(a);
}

template <int G1, int NT2>
static int launch_head3(const HeadArgs& a, hipStream_t stream) {
  constexpr size_t BUF = (size_t)(32 * G1 + 2 * NT2 * 32) * DmaGeom<3>::ROWB;
  constexpr size_t lds = 2 * BUF + 2 * kHeadMaxFeatures * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static LdsGrant grant;
  if (int rc = grant.ensure(reinterpret_cast<const void*>(&head_fused3_kernel<G1, NT2>), lds)) return rc;
  hipLaunchKernelGGL((head_fused3_kernel<G1, NT2>), dim3((unsigned)cdiv(a.n, 128)), dim3(512), lds, stream, a);
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

// text [n_classes][cmid] f32 -> [classes_padded][cmid/16][3][16] bf16, features permuted inside each group of 32
// exactly as in the two-term packing (rows >= n_classes zero)
__global__ void head_pack_text3_kernel(const float* __restrict__ text, int n_classes, int cmid, int classes_padded,
                                       __bf16* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)classes_padded * cmid) return;
  const int j = (int)(e / cmid), q = (int)(e % cmid);
  float v = j < n_classes ? text[(int64_t)j * cmid + (q & ~31) + head_perm_channel(q & 31)] : 0.f;
  __bf16* dst = out + (((int64_t)j * (cmid >> 4) + (q >> 4)) * 3) * 16 + (q & 15);
  for (int t = 0; t < 3; ++t) {
    const __bf16 b = (__bf16)v;
    dst[t * 16] = b;
    v -= (float)b;
  }
}

}  // namespace m3d

using namespace m3d;

extern "C" size_t m3d_head_text_bytes_terms(int classes_padded, int cmid, int terms) {
  if (classes_padded <= 0 || cmid <= 0 || (terms != 2 && terms != 3 && terms != kFmtF16)) return 0;
  return (size_t)classes_padded * cmid * split_nterms(terms) * sizeof(__bf16);
}

extern "C" size_t m3d_head_text_bytes(int classes_padded, int cmid) {
  return m3d_head_text_bytes_terms(classes_padded, cmid, 2);
}

extern "C" int m3d_head_pack_text(const float* d_text, int n_classes, int cmid, int classes_padded, void* d_out,
                                  void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(d_text && d_out, "m3d_head_pack_text: null pointer");
  M3D_CHECK_ARG(n_classes >= 1 && n_classes <= classes_padded && classes_padded % 32 == 0 && cmid >= 32 && cmid % 32 == 0,
                "m3d_head_pack_text: bad sizes");
  const int64_t total = (int64_t)classes_padded * cmid;
  hipLaunchKernelGGL(head_pack_text_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, d_text, n_classes,
                     cmid, classes_padded, static_cast<__bf16*>(d_out));
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

extern "C" int m3d_head_pack_text_terms(const float* d_text, int n_classes, int cmid, int classes_padded, int terms,
                                        void* d_out, void* stream_) {
  if (terms == 2) return m3d_head_pack_text(d_text, n_classes, cmid, classes_padded, d_out, stream_);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  M3D_CHECK_ARG(terms == 3 || terms == kFmtF16, "m3d_head_pack_text_terms: terms must be 2, 3 or 4 (the f16 pair)");
  M3D_CHECK_ARG(d_text && d_out, "m3d_head_pack_text_terms: null pointer");
  M3D_CHECK_ARG(n_classes >= 1 && n_classes <= classes_padded && classes_padded % 32 == 0 && cmid >= 32 && cmid % 32 == 0,
                "m3d_head_pack_text_terms: bad sizes");
  const int64_t total = (int64_t)classes_padded * cmid;
  if (terms == kFmtF16) {
    hipLaunchKernelGGL(head_pack_text_f16_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, d_text,
                       n_classes, cmid, classes_padded, static_cast<_Float16*>(d_out));
    M3D_LAUNCH_CHECK();
    return M3D_OK;
  }
  hipLaunchKernelGGL(head_pack_text3_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, d_text, n_classes,
                     cmid, classes_padded, static_cast<__bf16*>(d_out));
  M3D_LAUNCH_CHECK();
  return M3D_OK;
}

static int head_fused_two_terms(bool f16, const void* d_in_split, int cin, int64_t n, const void* d_w1,
                                const float* d_scale, const float* d_shift, int cmid, const void* d_text_packed,
                                int classes_padded, int n_classes, float* d_logits, int64_t ldl,
                                int32_t* d_voxel_label, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n == 0) return M3D_OK;
  M3D_CHECK_ARG(d_in_split && d_w1 && d_text_packed, "m3d_head_fused_sf32: null pointer");
  M3D_CHECK_ARG(d_logits || d_voxel_label, "m3d_head_fused_sf32: no output requested");
  M3D_CHECK_ARG(n > 0 && cmid >= 32 && cmid % 32 == 0 && n_classes >= 1 && n_classes <= classes_padded,
                "m3d_head_fused_sf32: bad sizes");
  M3D_CHECK_ARG(d_logits == nullptr || ldl >= n_classes, "m3d_head_fused_sf32: logits rows shorter than n_classes");
  HeadArgs a = {};
  a.x = static_cast<const char*>(d_in_split);
  a.w1 = static_cast<const char*>(d_w1);
  a.w2 = static_cast<const char*>(d_text_packed);
  a.scale = d_scale;
  a.shift = d_shift;
  a.logits = d_logits;
  a.ldl = ldl;
  a.label = d_voxel_label;
  a.n = n;
  a.cmid = cmid;
  a.n_classes = n_classes;
  if (f16) {
    if (int rc = f16_status_ptr(&a.f16_status)) return rc;
  }
  if (cmid > kHeadMaxFeatures) {
    set_error("m3d_head_fused_sf32: at most %d feature channels (got %d)", kHeadMaxFeatures, cmid);
    return M3D_EINVAL;
  }
  // 8 waves x 32 rows per workgroup where the DMA slot count divides by 512 threads, else 4 waves
#define M3D_HEAD_CASE(CIN_, CP_, NW_)                                                                 \
  if (cin == CIN_ && classes_padded == CP_)                                                            \
    return f16 ? launch_head<CIN_ / 32, CP_ / 32, NW_, true>(a, stream) : launch_head<CIN_ / 32, CP_ / 32, NW_, false>(a, stream);
  M3D_HEAD_CASE(96, 32, 8) M3D_HEAD_CASE(96, 96, 8) M3D_HEAD_CASE(96, 160, 8) M3D_HEAD_CASE(96, 224, 8)
  M3D_HEAD_CASE(96, 64, 4) M3D_HEAD_CASE(96, 128, 4) M3D_HEAD_CASE(96, 192, 4) M3D_HEAD_CASE(96, 256, 4)
  M3D_HEAD_CASE(32, 128, 4) M3D_HEAD_CASE(32, 256, 4) M3D_HEAD_CASE(64, 128, 8) M3D_HEAD_CASE(64, 256, 8)
  M3D_HEAD_CASE(128, 128, 8) M3D_HEAD_CASE(128, 256, 4)
#undef M3D_HEAD_CASE
  set_error("m3d_head_fused_sf32: no kernel for %d input channels and %d padded classes (supported: 96 channels "
            "with 32..256 classes in steps of 32; 32/64/128 channels with 128 or 256 classes)", cin, classes_padded);
  return M3D_EINVAL;
}

extern "C" int m3d_head_fused_sf32(const void* d_in_split, int cin, int64_t n, const void* d_w1, const float* d_scale,
                                   const float* d_shift, int cmid, const void* d_text_packed, int classes_padded,
                                   int n_classes, float* d_logits, int64_t ldl, int32_t* d_voxel_label,
                                   void* stream_) {
  return head_fused_two_terms(false, d_in_split, cin, n, d_w1, d_scale, d_shift, cmid, d_text_packed, classes_padded,
                              n_classes, d_logits, ldl, d_voxel_label, stream_);
}

extern "C" int m3d_head_fused_terms(const void* d_in_split, int cin, int64_t n, int terms, const void* d_w1,
                                    const float* d_scale, const float* d_shift, int cmid, const void* d_text_packed,
                                    int classes_padded, int n_classes, float* d_logits, int64_t ldl,
                                    int32_t* d_voxel_label, void* stream_) {
  if (terms == 2 || terms == kFmtF16)
    return head_fused_two_terms(terms == kFmtF16, d_in_split, cin, n, d_w1, d_scale, d_shift, cmid, d_text_packed,
                                classes_padded, n_classes, d_logits, ldl, d_voxel_label, stream_);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const char* who = "m3d_head_fused_terms";
  M3D_CHECK_ARG(terms == 3, "%s: terms must be 2, 3 or 4 (the f16 pair)", who);
  if (n == 0) return M3D_OK;
  M3D_CHECK_ARG(d_in_split && d_w1 && d_text_packed, "%s: null pointer", who);
  M3D_CHECK_ARG(d_logits || d_voxel_label, "%s: no output requested", who);
  M3D_CHECK_ARG(n > 0 && cmid >= 32 && cmid % 32 == 0 && cmid <= kHeadMaxFeatures && n_classes >= 1 &&
                    n_classes <= classes_padded,
                "%s: bad sizes", who);
  M3D_CHECK_ARG(d_logits == nullptr || ldl >= n_classes, "%s: logits rows shorter than n_classes", who);
  HeadArgs a = {};
  a.x = static_cast<const char*>(d_in_split);
  a.w1 = static_cast<const char*>(d_w1);
  a.w2 = static_cast<const char*>(d_text_packed);
  a.scale = d_scale;
  a.shift = d_shift;
  a.logits = d_logits;
  a.ldl = ldl;
  a.label = d_voxel_label;
  a.n = n;
  a.cmid = cmid;
  a.n_classes = n_classes;
#define M3D_HEAD3_CASE(CIN_, CP_) \
  if (cin == CIN_ && classes_padded == CP_) return launch_head3<CIN_ / 16, CP_ / 32>(a, stream);
  M3D_HEAD3_CASE(96, 32) M3D_HEAD3_CASE(96, 64) M3D_HEAD3_CASE(96, 96) M3D_HEAD3_CASE(96, 128) M3D_HEAD3_CASE(96, 160)
  M3D_HEAD3_CASE(96, 192) M3D_HEAD3_CASE(96, 224) M3D_HEAD3_CASE(96, 256)
  M3D_HEAD3_CASE(32, 128) M3D_HEAD3_CASE(32, 256) M3D_HEAD3_CASE(64, 128) M3D_HEAD3_CASE(64, 256)
  M3D_HEAD3_CASE(128, 128) M3D_HEAD3_CASE(128, 256)
#undef M3D_HEAD3_CASE
  set_error("%s: no three-term kernel for %d input channels and %d padded classes (supported: 96 channels with 32..256 "
            "classes in steps of 32; 32/64/128 channels with 128 or 256 classes)", who, cin, classes_padded);
  return M3D_EINVAL;
}
