// K1 / K1e: logits -> (new_logprobs, entropy) and its backward, plus the fused
// one-pass variant that also applies the GRPO token gradient.
//
// Reference pipelinerl/finetune/rl/__init__.py:207-233 does: logits/temperature
// (materialised copy), gather, logsumexp, then a 38-chunk entropy loop that re-reads
// the logits; autograd later re-reads everything to build d logits.  Here each logits
// row (V = 152 064 fp32 = 608 KB) is streamed ONCE by one workgroup with an online
// softmax carrying three running values per lane:
//     M = max_v y_v,  S = sum_v 2^(y_v - M),  W = sum_v (y_v - M) 2^(y_v - M),
//     y = logits * (log2(e) / temperature)
// so that  logsumexp = ln2 (M + log2 S),  entropy = ln2 (log2 S - W / S).
// HBM-bound: V*4 bytes in per token forward; V*4 in + V*4 out backward.  16-byte loads,
// UNROLL independent loads in flight per lane, wave64 shuffle combine, one LDS hop.
//
// Output convention: token-aligned (see include/prl.h): out[u] is computed from logits
// row u-1; column 0 of every batch row is written as 0.

#include <cstdlib>

#include "prl_common.h"
#include "prl_osm.h"
#include "prl_token_math.h"

namespace {

using prl::kWave;
using namespace prl::osm;


// ---- dtype adapters: a 16-byte vector of NV logits --------------------------------
struct F32 {
  using scalar = float;
  static constexpr int NV = 4;
  using vec = float __attribute__((ext_vector_type(4)));
  __device__ static __forceinline__ void unpack(const vec& x, float (&o)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = x[i];
  }
  __device__ static __forceinline__ vec pack(const float (&o)[NV]) {
    vec x;
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = o[i];
    return x;
  }
  __device__ static __forceinline__ float to_float(scalar s) { return s; }
  __device__ static __forceinline__ scalar from_float(float f) { return f; }
};

struct BF16 {
  using scalar = uint16_t;
  static constexpr int NV = 8;
  using vec = uint32_t __attribute__((ext_vector_type(4)));
  __device__ static __forceinline__ float to_float(scalar s) {
    return __uint_as_float(((uint32_t)s) << 16);
  }
  __device__ static __forceinline__ scalar from_float(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (scalar)((u >> 16) | 0x40u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                          // RNE
    return (scalar)(u >> 16);
  }
  __device__ static __forceinline__ void unpack(const vec& x, float (&o)[NV]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(x[i] << 16);
      o[2 * i + 1] = __uint_as_float(x[i] & 0xffff0000u);
    }
  }
  // gfx950 converts two fp32 to packed bf16 (round to nearest even) in one instruction
  // (v_cvt_pk_bf16_f32) - the software RNE above costs ~7 VALU operations per element.
  __device__ static __forceinline__ vec pack(const float (&o)[NV]) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    vec x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 pair = {o[2 * i], o[2 * i + 1]};
      x[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pair, bf16x2));
    }
    return x;
  }
};

template <int BLOCK>
__device__ __forceinline__ Osm osm_block_reduce(Osm s, Osm* lds /* [BLOCK/64] */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Osm t;
    t.M = __shfl_xor(s.M, o, 64);
    t.S = __shfl_xor(s.S, o, 64);
    t.W = __shfl_xor(s.W, o, 64);
    s = osm_merge(s, t);
  }
  constexpr int NW = BLOCK / kWave;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) lds[wid] = s;
  __syncthreads();
  Osm r = lds[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r = osm_merge(r, lds[w]);
  return r;  // identical in every thread
}

// ---- pass 1: stream one row, return the block-wide state -----------------------------
template <class T, int BLOCK, int UNROLL>
__device__ __forceinline__ Osm row_softmax_stats(const typename T::scalar* row, int vocab, float k2,
                                                 bool vec_ok, Osm* lds) {
  using vec = typename T::vec;
  constexpr int NV = T::NV;
  Osm st;
  osm_init(st);
  const int tid = threadIdx.x;
  int done = 0;  // elements covered by the vector path
  if (vec_ok) {
    const vec* rv = reinterpret_cast<const vec*>(row);
    const int nvec = vocab / NV;
    constexpr int TILE = BLOCK * UNROLL;
    const int nfull = (nvec / TILE) * TILE;
    for (int base = 0; base < nfull; base += TILE) {
      vec v[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) v[k] = rv[base + k * BLOCK + tid];
      float y[UNROLL * NV];
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        float f[NV];
        T::unpack(v[k], f);
#pragma unroll
        for (int i = 0; i < NV; ++i) y[k * NV + i] = f[i] * k2;
      }
      osm_push<UNROLL * NV>(st, y);
    }
    for (int j = nfull + tid; j < nvec; j += BLOCK) {
      float f[NV], y[NV];
      T::unpack(rv[j], f);
#pragma unroll
      for (int i = 0; i < NV; ++i) y[i] = f[i] * k2;
      osm_push<NV>(st, y);
    }
    done = nvec * NV;
  }
  for (int j = done + tid; j < vocab; j += BLOCK) {
    float y[1] = {T::to_float(row[j]) * k2};
    osm_push<1>(st, y);
  }
  return osm_block_reduce<BLOCK>(st, lds);
}

// ---- pass 2: write d logits for one row --------------------------------------------
//   dz_v = -g p_v - gH p_v (ln p_v + H);  d logit_v = (dz_v + g 1[v == id]) / temperature
template <bool NT, class V>
__device__ __forceinline__ void store_vec(V* p, const V& v) {
  if constexpr (NT) {
    __builtin_nontemporal_store(v, p);
  } else {
    *p = v;
  }
}

// ---- where a row of d logits goes: the logits' own dtype and layout (may alias the logits)
template <class T>
struct DenseOut {
  typename T::scalar* p;
  __device__ __forceinline__ DenseOut row(int64_t q, int64_t stride) const { return DenseOut{p + q * stride}; }
  template <bool NT>
  __device__ __forceinline__ void put(int j, const float (&o)[T::NV]) const {
    store_vec<NT>(&reinterpret_cast<typename T::vec*>(p)[j], T::pack(o));
  }
  __device__ __forceinline__ void put1(int j, float x) const { p[j] = T::from_float(x); }
};

// REVERSE walks the row back to front: in the fused kernel the tail of the row is what pass 1
// touched last, i.e. what is most likely still in L2 / Infinity Cache.  NT marks the gradient
// stores non-temporal so they do not evict the logits lines pass 2 is about to re-read.
template <class T, int BLOCK, int UNROLL, bool REVERSE = false, bool NT = false, class OUT = DenseOut<T>>
__device__ __forceinline__ void row_write_grad(const typename T::scalar* row,
                                               OUT out, int vocab, float k2,
                                               float inv_temp, float lse2, float H, float g,
                                               float gH, int id, bool vec_ok) {
  using vec = typename T::vec;
  constexpr int NV = T::NV;
  const int tid = threadIdx.x;
  const float gi = g * inv_temp;
  const float ngi = -g * inv_temp;
  const float nhi = -gH * inv_temp;
  const bool use_h = (gH != 0.0f);
  auto one = [&](float x, int v) -> float {
    const float d2 = __builtin_fmaf(x, k2, -lse2);  // log2 p
    const float p = fast_exp2(d2);
    float r = ngi * p;
    if (use_h) r = __builtin_fmaf(nhi * p, __builtin_fmaf(d2, kLn2, H), r);
    if (v == id) r += gi;
    return r;
  };
  int done = 0;
  if (vec_ok) {
    const vec* rv = reinterpret_cast<const vec*>(row);
    const int nvec = vocab / NV;
    constexpr int TILE = BLOCK * UNROLL;
    const int nfull = (nvec / TILE) * TILE;
    auto tail = [&]() {
      for (int j = nfull + tid; j < nvec; j += BLOCK) {
        float f[NV], o[NV];
        T::unpack(rv[j], f);
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = one(f[i], j * NV + i);
        out.template put<NT>(j, o);
      }
    };
    if constexpr (REVERSE) tail();
    for (int it = 0; it < nfull; it += TILE) {
      const int base = REVERSE ? (nfull - TILE - it) : it;
      vec v[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) v[k] = rv[base + k * BLOCK + tid];
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        const int j = base + k * BLOCK + tid;
        float f[NV], o[NV];
        T::unpack(v[k], f);
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = one(f[i], j * NV + i);
        out.template put<NT>(j, o);
      }
    }
    if constexpr (!REVERSE) tail();
    done = nvec * NV;
  }
  for (int j = done + tid; j < vocab; j += BLOCK) out.put1(j, one(T::to_float(row[j]), j));
}

template <class T, int BLOCK, class OUT>
__device__ __forceinline__ void row_write_zero(OUT out, int vocab, bool vec_ok) {
  constexpr int NV = T::NV;
  const int tid = threadIdx.x;
  int done = 0;
  if (vec_ok) {
    const int nvec = vocab / NV;
    float z[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) z[i] = 0.0f;
    for (int j = tid; j < nvec; j += BLOCK) out.template put<false>(j, z);
    done = nvec * NV;
  }
  for (int j = done + tid; j < vocab; j += BLOCK) out.put1(j, 0.0f);
}

struct RowGeom {
  int64_t n;     // rows * cols
  int64_t cols;
  int vocab;
  int64_t stride;  // elements
  int vec_ok;
};

// ---------------------------------------------------------------------------------------
// forward: one workgroup per token-aligned output u (logits row u - 1)
// ---------------------------------------------------------------------------------------
template <class T, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void logprob_entropy_fwd_kernel(
    RowGeom geo, const typename T::scalar* __restrict__ logits, const int64_t* __restrict__ ids,
    float k2, float* __restrict__ nlp, float* __restrict__ ent, float* __restrict__ lse2) {
  __shared__ Osm lds[BLOCK / kWave];
  const int64_t u = blockIdx.x;
  const int64_t col = u % geo.cols;
  if (col == 0) {
    if (threadIdx.x == 0) {
      nlp[u] = 0.0f;
      ent[u] = 0.0f;
      lse2[u] = 0.0f;
    }
    return;
  }
  const typename T::scalar* row = logits + (u - 1) * geo.stride;
  const Osm st = row_softmax_stats<T, BLOCK, UNROLL>(row, geo.vocab, k2, geo.vec_ok, lds);
  if (threadIdx.x == 0) {
    const int64_t id = ids[u];
    const float l2s = __log2f(st.S);
    float y_sel = __builtin_nanf("");
    if (id >= 0 && id < geo.vocab) y_sel = T::to_float(row[id]) * k2;
    nlp[u] = (y_sel - st.M - l2s) * kLn2;
    ent[u] = kLn2 * (l2s - st.W / st.S);
    lse2[u] = st.M + l2s;
  }
}

// ---------------------------------------------------------------------------------------
// backward: one workgroup per logits row q (feeds token u = q + 1 unless q is a row's last)
// ---------------------------------------------------------------------------------------
template <class T, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void logprob_entropy_bwd_kernel(
    RowGeom geo, const typename T::scalar* logits, const int64_t* __restrict__ ids, float k2,
    float inv_temp, const float* __restrict__ lse2, const float* __restrict__ ent,
    const float* __restrict__ g_nlp, const float* __restrict__ g_ent,
    const float* __restrict__ upstream, typename T::scalar* grad) {
  const int64_t q = blockIdx.x;
  const int64_t col = q % geo.cols;
  const DenseOut<T> out{grad + q * geo.stride};
  float g = 0.0f, gH = 0.0f;
  const int64_t u = q + 1;
  if (col != geo.cols - 1) {
    const float sc = upstream ? *upstream : 1.0f;
    g = g_nlp[u] * sc;
    gH = g_ent ? g_ent[u] * sc : 0.0f;
  }
  if (g == 0.0f && gH == 0.0f) {
    row_write_zero<T, BLOCK>(out, geo.vocab, geo.vec_ok);
    return;
  }
  const int64_t id64 = ids[u];
  const int id = (id64 >= 0 && id64 < geo.vocab) ? (int)id64 : -1;
  row_write_grad<T, BLOCK, UNROLL>(logits + q * geo.stride, out, geo.vocab, k2, inv_temp, lse2[u],
                                   ent[u], g, gH, id, geo.vec_ok);
}

// ---------------------------------------------------------------------------------------
// fused: forward stats -> token gradient -> d logits, one workgroup per logits row q
// ---------------------------------------------------------------------------------------
struct FusedArgs {
  prl_loss_config cfg;
  const int64_t* ids;
  const int64_t* labels;
  const float* old_lp;
  const float* ref_lp;
  const float* adv;
  const float* reward;
  const float* group_tokens;
  const float* overflow;
  float* nlp;
  float* ent;
  float* lse2;
  float up;  // expected upstream factor d objective / d loss (prl_loss_config.upstream_scale)
};

template <class T, int BLOCK, int UNROLL, bool REVERSE, bool NT, class OUT = DenseOut<T>>
__global__ __launch_bounds__(BLOCK) void fused_logits_loss_kernel(
    RowGeom geo, FusedArgs a, const typename T::scalar* logits, float k2, float inv_temp,
    OUT grad) {
  // The launch may request extra (unused) dynamic LDS to cap the workgroups resident per CU, so
  // that all rows in flight (256 CUs x k x 608 KB) fit the 256 MB Infinity Cache for pass 2.
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  Osm* lds = reinterpret_cast<Osm*>(dyn_lds);
  const int64_t q = blockIdx.x;
  const int64_t col = q % geo.cols;
  const OUT out = grad.row(q, geo.stride);
  if (col == 0 && threadIdx.x == 0) {  // token-aligned column 0 has no prediction
    a.nlp[q] = 0.0f;
    a.ent[q] = 0.0f;
    a.lse2[q] = 0.0f;
  }
  if (col == geo.cols - 1) {  // last logits row of a batch row predicts nothing
    row_write_zero<T, BLOCK>(out, geo.vocab, geo.vec_ok);
    return;
  }
  const int64_t u = q + 1;
  if (a.cfg.skip_unlabelled && a.labels[u] == -100) {  // nothing downstream reads this row's outputs: do not read the row
    if (threadIdx.x == 0) {
      a.nlp[u] = 0.0f;
      a.ent[u] = 0.0f;
      a.lse2[u] = 0.0f;
    }
    row_write_zero<T, BLOCK>(out, geo.vocab, geo.vec_ok);
    return;
  }
  const typename T::scalar* row = logits + q * geo.stride;
  // read the selected logit BEFORE pass 1's barrier: with grad aliasing logits, pass 2 of a
  // faster wave may already overwrite row[id] once the barrier has been passed.
  const int64_t id64 = a.ids[u];
  const int id = (id64 >= 0 && id64 < geo.vocab) ? (int)id64 : -1;
  float y_sel = __builtin_nanf("");
  if (id >= 0) y_sel = T::to_float(row[id]) * k2;
  const Osm st = row_softmax_stats<T, BLOCK, UNROLL>(row, geo.vocab, k2, geo.vec_ok, lds);
  const float l2s = __log2f(st.S);
  const float nlp = (y_sel - st.M - l2s) * kLn2;
  const float H = kLn2 * (l2s - st.W / st.S);
  const float lse2 = st.M + l2s;
  if (threadIdx.x == 0) {
    a.nlp[u] = nlp;
    a.ent[u] = H;
    a.lse2[u] = lse2;
  }
  const bool m = a.labels[u] != -100;
  float g = 0.0f, gH = 0.0f;
  if (m) {
    PrlTokenIn x{nlp,      H,           a.old_lp[u],       a.ref_lp[u], a.adv[u],
                 a.reward[u], a.group_tokens[u], 1.0f /*num_labels unused*/, a.overflow[u]};
    prl_token_grad(a.cfg, x, 1, &g, &gH);
    g *= a.up;
    gH *= a.up;
  }
  if (g == 0.0f && gH == 0.0f) {
    row_write_zero<T, BLOCK>(out, geo.vocab, geo.vec_ok);
    return;
  }
  row_write_grad<T, BLOCK, UNROLL, REVERSE, NT>(row, out, geo.vocab, k2, inv_temp, lse2, H, g, gH, id,
                                                geo.vec_ok);
}

// ---------------------------------------------------------------------------------------
// fused, row-resident variant: the head of the row stays ON CHIP between the two passes.
//
// A 608 KB fp32 row does not fit one CU (512 KB of VGPRs + 160 KB of LDS minus working set), but
// most of it does: every lane keeps KREG 16-byte vectors in registers and KLDS more in LDS
// (1024 lanes x (16 + 9) x 16 B = 400 KB = 67 % of the row).  Pass 2 recomputes the gradient of
// that part from the chip-resident copy and re-reads only the tail - which pass 1 touched last,
// so it is the part most likely still in L2 / Infinity Cache.  HBM-side traffic drops from
// 3 x V x 4 to about (2 + 0.33) x V x 4 bytes per token.
// ---------------------------------------------------------------------------------------
struct GradParams {
  float k2, lse2, H, ngi, nhi, gi;
  int id;
  bool use_h;
};

__device__ __forceinline__ float grad_one(const GradParams& p, float x, int v) {
  const float d2 = __builtin_fmaf(x, p.k2, -p.lse2);  // log2 p_v
  const float pv = fast_exp2(d2);
  float r = p.ngi * pv;
  if (p.use_h) r = __builtin_fmaf(p.nhi * pv, __builtin_fmaf(d2, kLn2, p.H), r);
  if (v == p.id) r += p.gi;
  return r;
}

// gradient of the 16-byte group j of a row, non-temporal store to wherever the row's gradient goes
template <class T, class OUT>
__device__ __forceinline__ void put_grad(const OUT& out, const GradParams& p, const typename T::vec& in, int j) {
  constexpr int NV = T::NV;
  float f[NV], o[NV];
  T::unpack(in, f);
#pragma unroll
  for (int i = 0; i < NV; ++i) o[i] = grad_one(p, f[i], j * NV + i);
  out.template put<true>(j, o);
}

template <class T>
__device__ __forceinline__ void push_vec(Osm& st, const typename T::vec& v, float k2) {
  constexpr int NV = T::NV;
  float f[NV], y[NV];
  T::unpack(v, f);
#pragma unroll
  for (int i = 0; i < NV; ++i) y[i] = f[i] * k2;
  osm_push<NV>(st, y);
}

template <bool NT, class V>
__device__ __forceinline__ V load_vec(const V* p) {
  if constexpr (NT) {
    return __builtin_nontemporal_load(p);
  } else {
    return *p;
  }
}

// NTHEAD: the chip-resident head is read exactly once -> stream it past the caches so that the
// tail (which IS re-read) keeps its lines in L2 / Infinity Cache.
// Measured in round 4 and NOT kept: TWO 512-thread workgroups per CU, each with the head of its own row on chip (16 + 9 or
// 12 + 9 vectors per lane, registers capped at 128 for four waves per SIMD; one workgroup drains / reduces while the other
// streams): 2020-2070 us against 1800-1820 us for the one 1024-thread workgroup below on the same box
// (profiles/r04d_kernel_sweep_fused_variants.txt).  Each row keeps half as much on chip (33 % / 28 % instead of 67 %), the
// re-read share of pass 2 doubles, and that costs more than the overlap of the two rows' phases buys.
template <class T, int BLOCK, int UNROLL, int KREG, int KLDS, bool NTHEAD = false, class OUT = DenseOut<T>>
__global__ __launch_bounds__(BLOCK) void fused_logits_loss_keep_kernel(
    RowGeom geo, FusedArgs a, const typename T::scalar* logits, float k2, float inv_temp,
    OUT grad) {
  using vec = typename T::vec;
  constexpr int NV = T::NV;
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  Osm* red = reinterpret_cast<Osm*>(dyn_lds);                 // [BLOCK / 64] (<= 256 bytes)
  vec* lds_keep = reinterpret_cast<vec*>(dyn_lds + 256);      // [KLDS][BLOCK]
  const int tid = threadIdx.x;
  const int64_t q = blockIdx.x;
  const int64_t col = q % geo.cols;
  const OUT out = grad.row(q, geo.stride);
  if (col == 0 && tid == 0) {
    a.nlp[q] = 0.0f;
    a.ent[q] = 0.0f;
    a.lse2[q] = 0.0f;
  }
  if (col == geo.cols - 1) {
    row_write_zero<T, BLOCK>(out, geo.vocab, true);
    return;
  }
  const int64_t u = q + 1;
  if (a.cfg.skip_unlabelled && a.labels[u] == -100) {  // nothing downstream reads this row's outputs: do not read the row
    if (tid == 0) {
      a.nlp[u] = 0.0f;
      a.ent[u] = 0.0f;
      a.lse2[u] = 0.0f;
    }
    row_write_zero<T, BLOCK>(out, geo.vocab, true);
    return;
  }
  const typename T::scalar* row = logits + q * geo.stride;
  const int64_t id64 = a.ids[u];
  const int id = (id64 >= 0 && id64 < geo.vocab) ? (int)id64 : -1;
  float y_sel = __builtin_nanf("");
  if (id >= 0) y_sel = T::to_float(row[id]) * k2;

  const vec* rv = reinterpret_cast<const vec*>(row);
  const int nvec = geo.vocab / NV;                 // host guarantees nvec >= (KREG + KLDS) * BLOCK
  constexpr int HEAD = (KREG + KLDS) * BLOCK;      // vectors kept on chip
  constexpr int TILE = BLOCK * UNROLL;

  // ---- pass 1 -------------------------------------------------------------------------
  Osm st;
  osm_init(st);
  vec keep[KREG > 0 ? KREG : 1];
#pragma unroll
  for (int k = 0; k < KREG; ++k) keep[k] = load_vec<NTHEAD>(&rv[k * BLOCK + tid]);
#pragma unroll
  for (int k = 0; k < KREG; ++k) push_vec<T>(st, keep[k], k2);
#pragma unroll
  for (int k = 0; k < KLDS; ++k) {
    const vec v = load_vec<NTHEAD>(&rv[(KREG + k) * BLOCK + tid]);
    lds_keep[k * BLOCK + tid] = v;
    push_vec<T>(st, v, k2);
  }
  const int ntail = nvec - HEAD;
  const int nfull = (ntail / TILE) * TILE;
  for (int base = 0; base < nfull; base += TILE) {
    vec v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = rv[HEAD + base + k * BLOCK + tid];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) push_vec<T>(st, v[k], k2);
  }
  for (int j = HEAD + nfull + tid; j < nvec; j += BLOCK) push_vec<T>(st, rv[j], k2);
  for (int j = nvec * NV + tid; j < geo.vocab; j += BLOCK) {
    float y[1] = {T::to_float(row[j]) * k2};
    osm_push<1>(st, y);
  }
  st = osm_block_reduce<BLOCK>(st, red);

  const float l2s = __log2f(st.S);
  const float nlp = (y_sel - st.M - l2s) * kLn2;
  const float H = kLn2 * (l2s - st.W / st.S);
  const float lse2 = st.M + l2s;
  if (tid == 0) {
    a.nlp[u] = nlp;
    a.ent[u] = H;
    a.lse2[u] = lse2;
  }
  const bool m = a.labels[u] != -100;
  float g = 0.0f, gH = 0.0f;
  if (m) {
    PrlTokenIn x{nlp,      H,           a.old_lp[u],       a.ref_lp[u], a.adv[u],
                 a.reward[u], a.group_tokens[u], 1.0f, a.overflow[u]};
    prl_token_grad(a.cfg, x, 1, &g, &gH);
    g *= a.up;
    gH *= a.up;
  }
  if (g == 0.0f && gH == 0.0f) {
    row_write_zero<T, BLOCK>(out, geo.vocab, true);
    return;
  }

  // ---- pass 2: tail (re-read, back to front) -> LDS part -> register part ------------------
  GradParams p{k2, lse2, H, -g * inv_temp, -gH * inv_temp, g * inv_temp, id, gH != 0.0f};
  for (int j = nvec * NV + tid; j < geo.vocab; j += BLOCK) out.put1(j, grad_one(p, T::to_float(row[j]), j));
  for (int j = HEAD + nfull + tid; j < nvec; j += BLOCK) put_grad<T>(out, p, rv[j], j);
  for (int it = 0; it < nfull; it += TILE) {
    const int base = HEAD + (nfull - TILE - it);
    vec v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = rv[base + k * BLOCK + tid];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const int j = base + k * BLOCK + tid;
      put_grad<T>(out, p, v[k], j);
    }
  }
#pragma unroll
  for (int k = KLDS - 1; k >= 0; --k) {
    const int j = (KREG + k) * BLOCK + tid;
    put_grad<T>(out, p, lds_keep[k * BLOCK + tid], j);
  }
#pragma unroll
  for (int k = KREG - 1; k >= 0; --k) {
    const int j = k * BLOCK + tid;
    put_grad<T>(out, p, keep[k], j);
  }
}

// ---- host-side dispatch ---------------------------------------------------------------
constexpr int kBlock = 256;
constexpr int kUnrollFwd = 8;
constexpr int kUnrollBwd = 4;
constexpr int kDefaultFusedVariant = 21;  // measured fastest on MI355X (profiles/r01_kernel_sweep.txt)

thread_local const char* g_last_fused = "";

// data[i] *= *upstream / expected, skipped entirely (one scalar load per workgroup) when the
// device scalar already has the expected value.
template <class T>
__global__ __launch_bounds__(256) void scale_unless_kernel(typename T::scalar* data, int64_t n,
                                                           const float* __restrict__ upstream, float expected) {
  const float up = *upstream;
  if (up == expected) return;
  const float f = up / expected;
  using vec = typename T::vec;
  constexpr int NV = T::NV;
  const int64_t nvec = (reinterpret_cast<uintptr_t>(data) & 15u) == 0 ? n / NV : 0;
  vec* dv = reinterpret_cast<vec*>(data);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nvec; j += stride) {
    float x[NV];
    T::unpack(dv[j], x);
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] *= f;
    dv[j] = T::pack(x);
  }
  for (int64_t j = nvec * NV + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
    data[j] = T::from_float(T::to_float(data[j]) * f);
}

int check_geom(int64_t rows, int64_t cols, int64_t vocab, const void* logits, int32_t dtype,
               int64_t stride, RowGeom* geo) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(vocab >= 1 && vocab < (int64_t)1 << 31, "vocab out of range: %lld", (long long)vocab);
  PRL_CHECK_ARG(dtype == PRL_DTYPE_F32 || dtype == PRL_DTYPE_BF16, "unsupported logits dtype %d", dtype);
  PRL_CHECK_ARG(stride >= vocab, "logits_row_stride %lld < vocab %lld", (long long)stride, (long long)vocab);
  PRL_CHECK_ARG(rows * cols < ((int64_t)1 << 31), "too many logits rows for one launch: %lld",
                (long long)(rows * cols));
  PRL_CHECK_ARG(logits != nullptr, "logits is null");
  const int esz = dtype == PRL_DTYPE_F32 ? 4 : 2;
  geo->n = rows * cols;
  geo->cols = cols;
  geo->vocab = (int)vocab;
  geo->stride = stride;
  geo->vec_ok = prl::aligned16(logits) && ((stride * esz) % 16 == 0);
  return PRL_OK;
}

}  // namespace

extern "C" int prl_logprob_entropy_fwd(int64_t rows, int64_t cols, int64_t vocab,
                                       const void* logits, int32_t logits_dtype,
                                       int64_t logits_row_stride, const int64_t* input_ids,
                                       float temperature, float* new_logprobs, float* entropy,
                                       float* lse2, prl_stream_t stream) {
  RowGeom geo;
  if (int rc = check_geom(rows, cols, vocab, logits, logits_dtype, logits_row_stride, &geo)) return rc;
  PRL_CHECK_ARG(input_ids && new_logprobs && entropy && lse2, "null pointer");
  PRL_CHECK_ARG(temperature > 0.0f, "temperature must be > 0");
  const float k2 = kLog2e / temperature;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)geo.n), block(kBlock);
  if (logits_dtype == PRL_DTYPE_F32) {
    hipLaunchKernelGGL((logprob_entropy_fwd_kernel<F32, kBlock, kUnrollFwd>), grid, block, 0, s, geo,
                       static_cast<const float*>(logits), input_ids, k2, new_logprobs, entropy, lse2);
  } else {
    hipLaunchKernelGGL((logprob_entropy_fwd_kernel<BF16, kBlock, kUnrollFwd>), grid, block, 0, s, geo,
                       static_cast<const uint16_t*>(logits), input_ids, k2, new_logprobs, entropy,
                       lse2);
  }
  PRL_LAUNCH_CHECK("logprob_entropy_fwd_kernel");
  return PRL_OK;
}

extern "C" int prl_logprob_entropy_bwd(int64_t rows, int64_t cols, int64_t vocab,
                                       const void* logits, int32_t logits_dtype,
                                       int64_t logits_row_stride, const int64_t* input_ids,
                                       float temperature, const float* lse2, const float* entropy,
                                       const float* grad_new_logprobs, const float* grad_entropy,
                                       const float* upstream, void* grad_logits,
                                       prl_stream_t stream) {
  RowGeom geo;
  if (int rc = check_geom(rows, cols, vocab, logits, logits_dtype, logits_row_stride, &geo)) return rc;
  PRL_CHECK_ARG(input_ids && lse2 && entropy && grad_new_logprobs && grad_logits, "null pointer");
  PRL_CHECK_ARG(temperature > 0.0f, "temperature must be > 0");
  geo.vec_ok = geo.vec_ok && prl::aligned16(grad_logits);
  const float k2 = kLog2e / temperature;
  const float inv_temp = 1.0f / temperature;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)geo.n), block(kBlock);
  if (logits_dtype == PRL_DTYPE_F32) {
    hipLaunchKernelGGL((logprob_entropy_bwd_kernel<F32, kBlock, kUnrollBwd>), grid, block, 0, s, geo,
                       static_cast<const float*>(logits), input_ids, k2, inv_temp, lse2, entropy,
                       grad_new_logprobs, grad_entropy, upstream, static_cast<float*>(grad_logits));
  } else {
    hipLaunchKernelGGL((logprob_entropy_bwd_kernel<BF16, kBlock, kUnrollBwd>), grid, block, 0, s, geo,
                       static_cast<const uint16_t*>(logits), input_ids, k2, inv_temp, lse2, entropy,
                       grad_new_logprobs, grad_entropy, upstream,
                       static_cast<uint16_t*>(grad_logits));
  }
  PRL_LAUNCH_CHECK("logprob_entropy_bwd_kernel");
  return PRL_OK;
}

extern "C" const char* prl_last_fused_kernel(void) { return g_last_fused; }

extern "C" int prl_scale_unless(void* data, int64_t n, int32_t dtype, const float* upstream, float expected,
                                prl_stream_t stream) {
  PRL_CHECK_ARG(data != nullptr && upstream != nullptr, "null pointer");
  PRL_CHECK_ARG(n >= 0, "negative element count");
  PRL_CHECK_ARG(dtype == PRL_DTYPE_F32 || dtype == PRL_DTYPE_BF16, "unsupported dtype %d", dtype);
  PRL_CHECK_ARG(expected != 0.0f && expected == expected, "expected factor must be a non-zero number");
  if (n == 0) return PRL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(2048), block(256);
  if (dtype == PRL_DTYPE_F32) {
    hipLaunchKernelGGL((scale_unless_kernel<F32>), grid, block, 0, s, static_cast<float*>(data), n, upstream, expected);
  } else {
    hipLaunchKernelGGL((scale_unless_kernel<BF16>), grid, block, 0, s, static_cast<uint16_t*>(data), n, upstream, expected);
  }
  PRL_LAUNCH_CHECK("scale_unless_kernel");
  return PRL_OK;
}

extern "C" int prl_fused_logits_loss(const prl_loss_config* cfg, int64_t rows, int64_t cols,
                                     int64_t vocab, const void* logits, int32_t logits_dtype,
                                     int64_t logits_row_stride, float temperature,
                                     const int64_t* input_ids, const int64_t* labels,
                                     const float* old_logprobs, const float* ref_logprobs,
                                     const float* advantages, const float* rewards,
                                     const float* group_tokens, const float* overflow,
                                     float* new_logprobs, float* entropy, float* lse2,
                                     void* grad_logits, prl_stream_t stream) {
  RowGeom geo;
  if (int rc = check_geom(rows, cols, vocab, logits, logits_dtype, logits_row_stride, &geo)) return rc;
  PRL_CHECK_ARG(cfg != nullptr, "cfg is null");
  PRL_CHECK_ARG(cfg->policy_loss == PRL_POLICY_PPO || cfg->policy_loss == PRL_POLICY_REINFORCE,
                "unknown policy_loss %d", cfg->policy_loss);
  PRL_CHECK_ARG(input_ids && labels && old_logprobs && ref_logprobs && advantages && rewards &&
                    group_tokens && overflow && new_logprobs && entropy && lse2 && grad_logits,
                "null pointer");
  PRL_CHECK_ARG(temperature > 0.0f, "temperature must be > 0");
  geo.vec_ok = geo.vec_ok && prl::aligned16(grad_logits);
  FusedArgs a{*cfg,      input_ids, labels,       old_logprobs, ref_logprobs, advantages,
              rewards,   group_tokens, overflow,  new_logprobs, entropy,      lse2,
              cfg->upstream_scale == 0.0f ? 1.0f : cfg->upstream_scale};
  const float k2 = kLog2e / temperature;
  const float inv_temp = 1.0f / temperature;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)geo.n);
  // Launch shape (PRL_TUNE_FUSED_VARIANT overrides the built-in choice; no getenv on this path):
  //   21  row-resident: 16 sixteen-byte vectors per lane in VGPRs + 9 in LDS stay on chip between the two passes,
  //       non-temporal head loads - the fp32 default (profiles/r01_kernel_sweep.txt)
  //    4  two-sweep, 512 threads, 2 workgroups per CU, reversed second pass, non-temporal stores - the bf16 default:
  //       bf16 rows carry twice the exp / convert work per byte and a whole step's rows (304 KB x 256 CUs) sit in the
  //       Infinity Cache, so the row-resident structure loses (profiles/r01r_kernel_sweep_bf16.txt: 1091 vs 1664 us)
  //    6  two-sweep, 1024 threads - what rows too short or unaligned for the row-resident shape fall back to
  //    0  two-sweep, 256 threads, forward order, plain stores - the simplest geometry, kept as the tests' A/B witness
  // The 19 other geometries of the round-1 / round-2 sweeps are no longer built; their numbers stay in
  // profiles/r01[c-f]_kernel_sweep*.txt.
  const int variant = (int)prl::tuning(PRL_TUNE_FUSED_VARIANT, logits_dtype == PRL_DTYPE_BF16 ? 4 : kDefaultFusedVariant);
#define PRL_FUSED_LAUNCH(TT, ST, BLK, UNR, REV, NTS, LDSB)                                            \
  do {                                                                                              \
    auto kfn = fused_logits_loss_kernel<TT, BLK, UNR, REV, NTS>;                                    \
    g_last_fused = "fused_logits_loss_kernel<" #TT "," #BLK "," #UNR "," #REV "," #NTS ">";        \
    const size_t lds_bytes = (LDSB) > 0 ? (size_t)(LDSB) : sizeof(Osm) * (BLK / kWave);             \
    static bool attr_set = false; /* one process drives one GPU: set the LDS opt-in once */         \
    if (lds_bytes > 48 * 1024 && !attr_set) {                                                       \
      PRL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                         \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
      attr_set = true;                                                                              \
    }                                                                                               \
    hipLaunchKernelGGL(kfn, grid, dim3(BLK), lds_bytes, s, geo, a, static_cast<const ST*>(logits), k2, \
                       inv_temp, DenseOut<TT>{static_cast<ST*>(grad_logits)});                      \
  } while (0)
#define PRL_KEEP_LAUNCH(TT, ST, BLK, UNR, KR, KL) PRL_KEEP_LAUNCH2(TT, ST, BLK, UNR, KR, KL, false)
#define PRL_KEEP_LAUNCH2(TT, ST, BLK, UNR, KR, KL, NTH)                                                     \
  do {                                                                                              \
    if (!geo.vec_ok || geo.vocab / TT::NV < (KR + KL) * BLK) {                                      \
      PRL_FUSED_LAUNCH(TT, ST, 1024, 4, true, true, 96 * 1024); /* row too short / unaligned */     \
      break;                                                                                        \
    }                                                                                               \
    auto kfn = fused_logits_loss_keep_kernel<TT, BLK, UNR, KR, KL, NTH>;                                 \
    g_last_fused = "fused_logits_loss_keep_kernel<" #TT "," #BLK "," #UNR "," #KR "," #KL "," #NTH ">";  \
    size_t lds_bytes = 256 + (size_t)(KL) * BLK * 16;                                               \
    if (lds_bytes < 96 * 1024) lds_bytes = 96 * 1024; /* keep ONE workgroup per CU */               \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      PRL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                         \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
      attr_set = true;                                                                              \
    }                                                                                               \
    hipLaunchKernelGGL(kfn, grid, dim3(BLK), lds_bytes, s, geo, a, static_cast<const ST*>(logits), k2, \
                       inv_temp, DenseOut<TT>{static_cast<ST*>(grad_logits)});                      \
  } while (0)
#define PRL_FUSED_DISPATCH(TT, ST)                                              \
  switch (variant) {                                                            \
    case 21: PRL_KEEP_LAUNCH2(TT, ST, 1024, 2, 16, 9, true); break;             \
    case 4: PRL_FUSED_LAUNCH(TT, ST, 512, 4, true, true, 64 * 1024); break;     \
    case 6: PRL_FUSED_LAUNCH(TT, ST, 1024, 4, true, true, 96 * 1024); break;    \
    case 0: PRL_FUSED_LAUNCH(TT, ST, 256, 4, false, false, 0); break;           \
    default: return prl::set_error(PRL_EINVAL, "fused variant %d is not built (0, 4, 6, 21)", variant); \
  }
  if (logits_dtype == PRL_DTYPE_F32) {
    PRL_FUSED_DISPATCH(F32, float)
  } else {
    PRL_FUSED_DISPATCH(BF16, uint16_t)
  }
#undef PRL_FUSED_DISPATCH
#undef PRL_KEEP_LAUNCH
#undef PRL_KEEP_LAUNCH2
#undef PRL_FUSED_LAUNCH
  PRL_LAUNCH_CHECK("fused_logits_loss_kernel");
  return PRL_OK;
}
