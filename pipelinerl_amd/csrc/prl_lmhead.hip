// Fused output head: hidden states -> log-prob / entropy of the next token WITHOUT ever writing the
// [T, V] logits to HBM, and its backward.
//
// The reference computes  logits = lm_head(hidden)  with the head forced to fp32
// (pipelinerl/finetune/checkpoints.py:87-103), hands the [1, T, V] fp32 tensor (4.98 GB for a
// Qwen2.5-7B micro-batch) to rl_step (pipelinerl/finetune/rl/__init__.py:204-233), which divides it by
// the temperature, gathers, takes a logsumexp and a chunked entropy, and lets autograd walk back
// through all of it.  Here the contraction runs on the bf16 matrix cores and the soft-max statistics
// are folded into the GEMM epilogue:
//
//   * fp32 accuracy on bf16 MFMA: the hidden states are bf16 already; the fp32 weight is split once
//     per optimizer step into two bf16 planes W = W_hi + W_lo (prl_lm_head_prepare).  bf16 x bf16
//     products are exact in fp32 and accumulate in fp32, so  h W_hi^T + h W_lo^T  reproduces the
//     fp32 product to ~2^-17 relative - the planes are simply further K-steps of ONE accumulator.
//   * forward: each workgroup owns 128 token rows and a range of vocabulary tiles; per 128 x 128 tile
//     the accumulators go straight into per-lane online-softmax states (M, S, W of prl_osm.h) - the
//     logits never leave the registers.  Partial states per (row, vocabulary split) are merged by a
//     small second kernel that also writes the token-aligned new_logprobs / entropy / lse2.
//   * backward: per chunk of rows the logits tile is recomputed by the same main loop, turned into
//     d logits with the saved lse2 / entropy and the per-token loss gradients, split into bf16
//     (hi, lo) planes and written in both layouts to a workspace sized for the chunk only; two more
//     passes of the same NT GEMM core produce  d hidden = d logits W  and  d W += d logits^T hidden.
//
// GEMM core: C[M, N] = sum_terms A_t[M, Kc] B_t[N, Kc]^T, both operands contraction-contiguous bf16,
// 128 x 128 x 64 tiles, 256 threads = 4 waves (2 x 2) of 64 x 64, mfma_f32_16x16x32_bf16, operands
// brought HBM -> LDS by global_load_lds (16 B per lane, no VGPR round trip) into a double buffer,
// 16-byte chunks XOR-swizzled on the SOURCE address so that the fragment ds_read_b128 are bank
// conflict free, one barrier per K-step with the next tile's loads in flight during the MFMAs.
// MFMA-bound by design; roofline = dense bf16 MFMA peak (2.5 PFLOP/s).

#include <cstdlib>

#include "prl_common.h"
#include "prl_lmhead_layout.h"
#include "prl_osm.h"

namespace {

using namespace prl::osm;
using namespace prl::lmhead;

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // double buffered: 64 KB -> two workgroups per CU
constexpr int MAX_TERMS = 3;

struct Terms {
  const uint16_t* a[MAX_TERMS];
  const uint16_t* b[MAX_TERMS];
  int n;
};

struct Geom {
  int M, N, Kc;      // Kc: contraction length per term (multiple of BK)
  int64_t lda, ldb;  // row strides in elements
};

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> (hi, lo) bf16 with round-to-nearest-even in hardware (v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split2(float x, uint16_t& hi, uint16_t& lo) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 a = {x, 0.0f};
  const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)) & 0xffffu;
  const float r = x - __uint_as_float(h << 16);
  const f32x2 b = {r, 0.0f};
  hi = (uint16_t)h;
  lo = (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf16x2)) & 0xffffu);
}

__device__ __forceinline__ uint16_t to_bf16(float x) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 a = {x, 0.0f};
  return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)) & 0xffffu);
}

// -----------------------------------------------------------------------------------------------
// main loop: acc[i][j] (16x16 tiles of this wave's 64 x 64) += sum over terms and K
// -----------------------------------------------------------------------------------------------
template <bool GLDS>
__device__ __forceinline__ void gemm_mainloop(f32x4 (&acc)[4][4], const Terms& t, const Geom& g, int m0, int n0,
                                              char* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging (layout: prl_lmhead_layout.h): this thread fetches chunks q * 256 + tid, q = 0..3,
  // i.e. rows stage_row(tid, q), all from source column stage_kcol(tid)
  const int kcol = stage_kcol(tid);
  int64_t offA[4], offB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int ra = m0 + stage_row(tid, q);
    ra = ra < g.M ? ra : g.M - 1;  // rows past the edge re-read the last row; their results are discarded
    int rb = n0 + stage_row(tid, q);
    rb = rb < g.N ? rb : g.N - 1;
    offA[q] = (int64_t)ra * g.lda + kcol;
    offB[q] = (int64_t)rb * g.ldb + kcol;
  }
  // ---- fragment reads: byte offsets of tile row-block i = 0 for the two 32-deep sub-steps; row-block i
  // adds i * 16 rows * 128 bytes (the swizzle term depends on (row >> 1) & 7 only, unchanged by + 16 i)
  const int rdA0 = frag_lds_byte(lane, wm, 0, 0), rdA1 = frag_lds_byte(lane, wm, 0, 1);
  const int rdB0 = TILE_BYTES + frag_lds_byte(lane, wn, 0, 0), rdB1 = TILE_BYTES + frag_lds_byte(lane, wn, 0, 1);

  const int ksteps = g.Kc / BK;
  const int total = t.n * ksteps;
  // the NEXT tile to stage
  int st_term = 0, st_k = 0;
  const uint16_t* sA = t.a[0];
  const uint16_t* sB = t.b[0];
  auto advance = [&]() {
    st_k += BK;
    if (st_k == g.Kc) {
      st_k = 0;
      ++st_term;
      sA = st_term == 1 ? t.a[1] : t.a[2];
      sB = st_term == 1 ? t.b[1] : t.b[2];
    }
  };

  auto compute = [&](int buf) {
    const char* base = lds + buf * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ra = ks ? rdA1 : rdA0, rb = ks ? rdB1 : rdB0;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + ra + i * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(base + rb + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  __syncthreads();  // whoever used the LDS before (previous tile, an epilogue) is done with it
  if constexpr (GLDS) {
    auto stage = [&](int buf) {
      const unsigned dst = buf * STAGE_BYTES + stage_lds_byte(wave * 64, 0);  // + q * 4096 + lane * 16 (hardware)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sA + offA[q] + st_k),
                                         (__attribute__((address_space(3))) void*)(lds + dst + q * 4096), 16, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sB + offB[q] + st_k),
                                         (__attribute__((address_space(3))) void*)(lds + dst + TILE_BYTES + q * 4096), 16, 0, 0);
    };
    stage(0);
    advance();
    for (int s = 0; s + 1 < total; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of tile s has landed
      __syncthreads();                                    // ... everybody's has; buffer (s+1)&1 is free again
      stage((s + 1) & 1);                                 // in flight during the MFMAs below
      advance();
      compute(s & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    compute((total - 1) & 1);
  } else {
    // register staging (fallback / A-B reference for the DMA path): same LDS image
    struct Regs {
      uint4 a0, a1, a2, a3, b0, b1, b2, b3;
    };
    auto gload = [&]() -> Regs {
      Regs r;
      r.a0 = *reinterpret_cast<const uint4*>(sA + offA[0] + st_k);
      r.a1 = *reinterpret_cast<const uint4*>(sA + offA[1] + st_k);
      r.a2 = *reinterpret_cast<const uint4*>(sA + offA[2] + st_k);
      r.a3 = *reinterpret_cast<const uint4*>(sA + offA[3] + st_k);
      r.b0 = *reinterpret_cast<const uint4*>(sB + offB[0] + st_k);
      r.b1 = *reinterpret_cast<const uint4*>(sB + offB[1] + st_k);
      r.b2 = *reinterpret_cast<const uint4*>(sB + offB[2] + st_k);
      r.b3 = *reinterpret_cast<const uint4*>(sB + offB[3] + st_k);
      return r;
    };
    auto lstore = [&](int buf, const Regs& r) {
      char* base = lds + buf * STAGE_BYTES + stage_lds_byte(tid, 0);
      *reinterpret_cast<uint4*>(base) = r.a0;
      *reinterpret_cast<uint4*>(base + 4096) = r.a1;
      *reinterpret_cast<uint4*>(base + 8192) = r.a2;
      *reinterpret_cast<uint4*>(base + 12288) = r.a3;
      *reinterpret_cast<uint4*>(base + TILE_BYTES) = r.b0;
      *reinterpret_cast<uint4*>(base + TILE_BYTES + 4096) = r.b1;
      *reinterpret_cast<uint4*>(base + TILE_BYTES + 8192) = r.b2;
      *reinterpret_cast<uint4*>(base + TILE_BYTES + 12288) = r.b3;
    };
    {
      const Regs r = gload();
      advance();
      lstore(0, r);
    }
    for (int s = 0; s + 1 < total; ++s) {
      __syncthreads();
      const Regs r = gload();
      advance();
      compute(s & 1);
      lstore((s + 1) & 1, r);
    }
    __syncthreads();
    compute((total - 1) & 1);
  }
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// C/D layout of mfma_f32_16x16x32: element `reg` of acc[i][j] is row 4 * (lane >> 4) + reg, column lane & 15
// of the 16 x 16 tile (i, j)  (prl_lmhead_layout.h: acc_row / acc_col).
struct LaneMap {
  int row0;  // acc_row(lane, wm, 0, 0)   (+ i * 16 + reg)
  int col0;  // acc_col(lane, wn, 0)      (+ j * 16)
};
__device__ __forceinline__ LaneMap lane_map() {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  return LaneMap{acc_row(lane, wave >> 1, 0, 0), acc_col(lane, wave & 1, 0)};
}

// -----------------------------------------------------------------------------------------------
// forward
// -----------------------------------------------------------------------------------------------
struct FwdArgs {
  Terms terms;          // a = hidden [n, hidden] (same pointer for every term), b = weight planes [vocab, hidden]
  Geom geo;             // M = n logits rows, N = vocab, Kc = hidden
  int64_t cols;         // batch columns: logits row q predicts token q + 1 unless q % cols == cols - 1
  const int64_t* ids;   // [n]
  float k2;             // log2(e) / temperature
  int mt, nt, nsplit;
  float* part;          // [nsplit][mt * BM][4]  (M, S, W, -)
  float* ysel;          // [mt * BM] selected logit (base-2 units), written by whichever split owns the column
};

template <bool GLDS>
__global__ __launch_bounds__(NTHREADS, 2) void lmhead_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int tm, split;
  tile_coords(blockIdx.x, a.mt, a.nsplit, tm, split);
  const int m0 = tm * BM;
  const int nt0 = (int)((int64_t)a.nt * split / a.nsplit), nt1 = (int)((int64_t)a.nt * (split + 1) / a.nsplit);
  const LaneMap lm = lane_map();

  Osm st[4][4];
  int tgt[4][4];  // target column of each of this lane's 16 rows, -1: none
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      osm_init(st[i][r]);
      const int64_t q = m0 + lm.row0 + i * 16 + r;
      int id = -1;
      if (q < a.geo.M && (q % a.cols) != a.cols - 1) {
        const int64_t v = a.ids[q + 1];
        if (v >= 0 && v < a.geo.N) id = (int)v;
      }
      tgt[i][r] = id;
    }

  f32x4 acc[4][4];
  for (int tn = nt0; tn < nt1; ++tn) {
    const int n0 = tn * BN;
    zero_acc(acc);
    gemm_mainloop<GLDS>(acc, a.terms, a.geo, m0, n0, lds);
    const int cbase = n0 + lm.col0;
    const bool full = n0 + BN <= a.geo.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = acc[i][j][r] * a.k2;
        const int d = tgt[i][r] - cbase;  // this lane holds columns cbase + 16 j
        if (d >= 0 && d < 64 && (d & 15) == 0) {
          const int64_t q = m0 + lm.row0 + i * 16 + r;
          a.ysel[q] = d == 0 ? y[0] : d == 16 ? y[1] : d == 32 ? y[2] : y[3];
        }
        if (full) {
          osm_push<4>(st[i][r], y);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (cbase + j * 16 < a.geo.N) {
              float one[1] = {y[j]};
              osm_push<1>(st[i][r], one);
            }
        }
      }
  }

  // merge the 16 lanes that share a row, then the two waves that share it (wn = 0, 1)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Osm s = st[i][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        Osm o2;
        o2.M = __shfl_xor(s.M, o, 64);
        o2.S = __shfl_xor(s.S, o, 64);
        o2.W = __shfl_xor(s.W, o, 64);
        s = osm_merge(s, o2);
      }
      st[i][r] = s;
    }
  __syncthreads();  // the last tile's LDS reads are done: reuse the buffer for the cross-wave hand-off
  float4* red = reinterpret_cast<float4*>(lds);  // [2][BM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(wave & 1) * BM + lm.row0 + i * 16 + r] = float4{st[i][r].M, st[i][r].S, st[i][r].W, 0.0f};
  }
  __syncthreads();
  if (tid < BM) {
    const float4 x = red[tid], y = red[BM + tid];
    const Osm m = osm_merge(Osm{x.x, x.y, x.z}, Osm{y.x, y.y, y.z});
    reinterpret_cast<float4*>(a.part)[((int64_t)split * a.mt * BM) + m0 + tid] = float4{m.M, m.S, m.W, 0.0f};
  }
}

// token-aligned outputs from the per-split partial states
__global__ __launch_bounds__(256) void lmhead_fwd_finish_kernel(int64_t n, int64_t cols, int vocab, int nsplit, int64_t padded,
                                                                const float* __restrict__ part, const float* __restrict__ ysel,
                                                                const int64_t* __restrict__ ids, float* __restrict__ nlp,
                                                                float* __restrict__ ent, float* __restrict__ lse2) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  if (u % cols == 0) {
    nlp[u] = 0.0f;
    ent[u] = 0.0f;
    lse2[u] = 0.0f;
    return;
  }
  const int64_t q = u - 1;
  const float4* p = reinterpret_cast<const float4*>(part);
  float4 x = p[q];
  Osm s{x.x, x.y, x.z};
  for (int k = 1; k < nsplit; ++k) {
    x = p[(int64_t)k * padded + q];
    s = osm_merge(s, Osm{x.x, x.y, x.z});
  }
  const float l2s = __log2f(s.S);
  const int64_t id = ids[u];
  const float y = (id >= 0 && id < vocab) ? ysel[q] : __builtin_nanf("");
  nlp[u] = (y - s.M - l2s) * kLn2;
  ent[u] = kLn2 * (l2s - s.W / s.S);
  lse2[u] = s.M + l2s;
}

// -----------------------------------------------------------------------------------------------
// backward, step 1: recompute one logits tile, emit d logits as (hi, lo) bf16 planes in both layouts
// -----------------------------------------------------------------------------------------------
struct DlArgs {
  Terms terms;
  Geom geo;             // M = rows of this chunk, N = vocab, Kc = hidden
  int64_t row_base;     // first logits row of the chunk (global index q)
  int64_t n_total;      // total logits rows
  int64_t cols;
  const int64_t* ids;
  const float* lse2;    // token-aligned [n_total]
  const float* ent;
  const float* g_nlp;   // token-aligned d loss / d new_logprobs
  const float* g_ent;   // nullable
  const float* upstream;  // nullable device scalar
  float k2, inv_temp;
  int mt, nt;
  int chunk_pad;        // rows of the chunk buffers (multiple of BM)
  uint16_t* dl_hi;      // [chunk_pad, vocab]
  uint16_t* dl_lo;
  uint16_t* dlT_hi;     // [vocab, chunk_pad]
  uint16_t* dlT_lo;
};

template <bool GLDS>
__global__ __launch_bounds__(NTHREADS, 2) void lmhead_dlogits_kernel(DlArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int tm, tn;
  tile_coords(blockIdx.x, a.mt, a.nt, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const LaneMap lm = lane_map();
  f32x4 acc[4][4];
  zero_acc(acc);
  gemm_mainloop<GLDS>(acc, a.terms, a.geo, m0, n0, lds);

  const float up = a.upstream ? *a.upstream : 1.0f;
  const int64_t V = a.geo.N;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = m0 + lm.row0 + i * 16 + r;  // row inside the chunk buffers
      const int64_t q = a.row_base + lrow;
      float g = 0.0f, gH = 0.0f, l2 = 0.0f, H = 0.0f;
      int id = -1;
      if (lrow < a.geo.M && (q % a.cols) != a.cols - 1) {
        const int64_t u = q + 1;
        g = a.g_nlp[u] * up;
        gH = a.g_ent ? a.g_ent[u] * up : 0.0f;
        l2 = a.lse2[u];
        H = a.ent[u];
        const int64_t v = a.ids[u];
        if (v >= 0 && v < V) id = (int)v;
      }
      const float gi = g * a.inv_temp, ngi = -g * a.inv_temp, nhi = -gH * a.inv_temp;
      const bool live = (g != 0.0f) || (gH != 0.0f);
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + lm.col0 + j * 16;
        float v = 0.0f;
        if (live) {
          const float d2 = __builtin_fmaf(acc[i][j][r], a.k2, -l2);  // log2 p
          const float p = fast_exp2(d2);
          v = ngi * p;
          if (gH != 0.0f) v = __builtin_fmaf(nhi * p, __builtin_fmaf(d2, kLn2, H), v);
          if (col == id) v += gi;
        }
        split2(v, hi[j], lo[j]);
      }
      if (lrow < a.chunk_pad) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = n0 + lm.col0 + j * 16;
          if (col < V) {
            a.dl_hi[(int64_t)lrow * V + col] = hi[j];
            a.dl_lo[(int64_t)lrow * V + col] = lo[j];
            a.dlT_hi[(int64_t)col * a.chunk_pad + lrow] = hi[j];
            a.dlT_lo[(int64_t)col * a.chunk_pad + lrow] = lo[j];
          }
        }
      }
    }
}

// -----------------------------------------------------------------------------------------------
// backward, steps 2 and 3: plain NT GEMM with a store / accumulate epilogue
// -----------------------------------------------------------------------------------------------
struct GemmArgs {
  Terms terms;
  Geom geo;
  int mt, nt;
  void* out;        // [M, N] row-major, ldc elements
  int64_t ldc;
  int out_bf16;     // 1: bf16 store, 0: fp32
  int accumulate;   // fp32 only: out += acc
};

template <bool GLDS>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int tm, tn;
  tile_coords(blockIdx.x, a.mt, a.nt, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const LaneMap lm = lane_map();
  f32x4 acc[4][4];
  zero_acc(acc);
  gemm_mainloop<GLDS>(acc, a.terms, a.geo, m0, n0, lds);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + lm.row0 + i * 16 + r;
      if (row >= a.geo.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + lm.col0 + j * 16;
        if (col >= a.geo.N) continue;
        const int64_t o = (int64_t)row * a.ldc + col;
        if (a.out_bf16) {
          static_cast<uint16_t*>(a.out)[o] = to_bf16(acc[i][j][r]);
        } else {
          float* dst = static_cast<float*>(a.out) + o;
          *dst = a.accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

// -----------------------------------------------------------------------------------------------
// operand preparation
// -----------------------------------------------------------------------------------------------
// W [V, K] fp32 or bf16 -> planes hi, lo [V, K] (row-major) and their transposes [K, ldt] (ldt >= V).
// 64 x 64 tiles through LDS; nullable outputs are skipped.
template <class SRC>
__global__ __launch_bounds__(256) void split_transpose_kernel(int64_t R, int64_t C, const SRC* __restrict__ src,
                                                              uint16_t* hi, uint16_t* lo, uint16_t* t_hi, uint16_t* t_lo,
                                                              int64_t ldt) {
  __shared__ uint16_t th[64][66];
  __shared__ uint16_t tl[64][66];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 rows per pass
  for (int rr = ty; rr < 64; rr += 4) {
    const int64_t r = r0 + rr, c = c0 + tx;
    uint16_t h = 0, l = 0;
    if (r < R && c < C) {
      float x;
      if constexpr (sizeof(SRC) == 4) {
        x = src[r * C + c];
      } else {
        x = bf16_to_f32(src[r * C + c]);
      }
      split2(x, h, l);
      if (hi) hi[r * C + c] = h;
      if (lo) lo[r * C + c] = l;
    }
    th[rr][tx] = h;
    tl[rr][tx] = l;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int64_t c = c0 + cc, r = r0 + tx;
    if (c < C && r < ldt) {  // columns r >= R of the transposed planes are zero padding
      if (t_hi) t_hi[c * ldt + r] = th[tx][cc];
      if (t_lo) t_lo[c * ldt + r] = tl[tx][cc];
    }
  }
}

// PRL_LMHEAD_STAGING=0 selects register staging instead of the LDS DMA (read per call, so one
// process can A/B the two).
int g_use_glds() {
  const char* e = getenv("PRL_LMHEAD_STAGING");
  return (e && atoi(e) == 0) ? 0 : 1;
}

template <class K, class A>
int launch_tiles(K kfn, int blocks, const A& args, hipStream_t s, const char* name) {
  static thread_local const void* configured[8] = {nullptr};
  const void* key = reinterpret_cast<const void*>(kfn);
  bool seen = false;
  for (auto c : configured) seen = seen || c == key;
  if (!seen) {
    PRL_HIP_CHECK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (auto& c : configured)
      if (c == nullptr) {
        c = key;
        break;
      }
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(NTHREADS), LDS_BYTES, s, args);
  PRL_LAUNCH_CHECK(name);
  return PRL_OK;
}

int fwd_nsplit(int mt, int nt) {
  if (const char* e = getenv("PRL_LMHEAD_NSPLIT")) {
    const int v = atoi(e);
    if (v >= 1) return v < nt ? v : nt;
  }
  int ns = (512 + mt - 1) / mt;  // two workgroups per CU resident, all 256 CUs busy
  if (ns < 1) ns = 1;
  return ns < nt ? ns : nt;
}

int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct BwdLayout {
  int chunk_pad;
  size_t hT, dl_hi, dl_lo, dlT_hi, dlT_lo, total;
};

BwdLayout bwd_layout(int64_t hidden, int64_t vocab, int64_t chunk_rows) {
  BwdLayout L;
  L.chunk_pad = ceil_div(chunk_rows, BM) * BM;
  size_t o = 0;
  L.hT = o;
  o += align256((size_t)hidden * L.chunk_pad * 2);
  const size_t plane = align256((size_t)L.chunk_pad * vocab * 2);
  L.dl_hi = o;
  o += plane;
  L.dl_lo = o;
  o += plane;
  L.dlT_hi = o;
  o += plane;
  L.dlT_lo = o;
  o += plane;
  L.total = o;
  return L;
}

}  // namespace

extern "C" int prl_lm_head_prepare(int64_t vocab, int64_t hidden, const void* weight, int32_t weight_dtype,
                                   uint16_t* w_hi, uint16_t* w_lo, uint16_t* wt_hi, uint16_t* wt_lo,
                                   prl_stream_t stream) {
  PRL_CHECK_ARG(vocab >= 1 && hidden >= 1 && weight != nullptr, "bad arguments");
  PRL_CHECK_ARG(weight_dtype == PRL_DTYPE_F32 || weight_dtype == PRL_DTYPE_BF16, "unsupported weight dtype %d", weight_dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)ceil_div(hidden, 64), (unsigned)ceil_div(vocab, 64));
  if (weight_dtype == PRL_DTYPE_F32) {
    hipLaunchKernelGGL((split_transpose_kernel<float>), grid, dim3(256), 0, s, vocab, hidden, static_cast<const float*>(weight),
                       w_hi, w_lo, wt_hi, wt_lo, vocab);
  } else {
    hipLaunchKernelGGL((split_transpose_kernel<uint16_t>), grid, dim3(256), 0, s, vocab, hidden,
                       static_cast<const uint16_t*>(weight), w_hi, w_lo, wt_hi, wt_lo, vocab);
  }
  PRL_LAUNCH_CHECK("split_transpose_kernel");
  return PRL_OK;
}

extern "C" int prl_lm_head_workspace_bytes(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, int64_t chunk_rows,
                                           size_t* fwd_bytes, size_t* bwd_bytes) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1 && hidden >= 1 && vocab >= 1, "bad sizes");
  const int64_t n = rows * cols;
  const int mt = ceil_div(n, BM), nt = ceil_div(vocab, BN);
  if (fwd_bytes) *fwd_bytes = align256((size_t)fwd_nsplit(mt, nt) * mt * BM * 16) + align256((size_t)mt * BM * 4);
  if (bwd_bytes) {
    PRL_CHECK_ARG(chunk_rows >= 1, "chunk_rows must be >= 1");
    *bwd_bytes = bwd_layout(hidden, vocab, chunk_rows < n ? chunk_rows : n).total;
  }
  return PRL_OK;
}

extern "C" int prl_lm_head_logprob_fwd(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                       const uint16_t* w_hi, const uint16_t* w_lo, const int64_t* input_ids,
                                       float temperature, float* new_logprobs, float* entropy, float* lse2,
                                       void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(hidden >= BK && hidden % BK == 0, "hidden size %lld must be a multiple of %d", (long long)hidden, BK);
  PRL_CHECK_ARG(vocab >= 1 && vocab < ((int64_t)1 << 31), "vocab out of range");
  PRL_CHECK_ARG(rows * cols < ((int64_t)1 << 31) - BM, "too many rows");
  PRL_CHECK_ARG(hidden_bf16 && w_hi && input_ids && new_logprobs && entropy && lse2 && workspace, "null pointer");
  PRL_CHECK_ARG(prl::aligned16(hidden_bf16) && prl::aligned16(w_hi) && (!w_lo || prl::aligned16(w_lo)), "operands must be 16-byte aligned");
  PRL_CHECK_ARG(temperature > 0.0f, "temperature must be > 0");
  const int64_t n = rows * cols;
  FwdArgs a;
  a.terms.n = w_lo ? 2 : 1;
  for (int k = 0; k < MAX_TERMS; ++k) {
    a.terms.a[k] = hidden_bf16;
    a.terms.b[k] = (k == 1 && w_lo) ? w_lo : w_hi;
  }
  a.geo = Geom{(int)n, (int)vocab, (int)hidden, hidden, hidden};
  a.cols = cols;
  a.ids = input_ids;
  a.k2 = kLog2e / temperature;
  a.mt = ceil_div(n, BM);
  a.nt = ceil_div(vocab, BN);
  a.nsplit = fwd_nsplit(a.mt, a.nt);
  const size_t part_bytes = align256((size_t)a.nsplit * a.mt * BM * 16);
  const size_t need = part_bytes + align256((size_t)a.mt * BM * 4);
  if (workspace_bytes < need) return prl::set_error(PRL_ENOMEM, "lm_head forward workspace: %zu bytes given, %zu needed", workspace_bytes, need);
  a.part = static_cast<float*>(workspace);
  a.ysel = reinterpret_cast<float*>(static_cast<char*>(workspace) + part_bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = a.mt * a.nsplit;
  int rc = g_use_glds() ? launch_tiles(lmhead_fwd_kernel<true>, blocks, a, s, "lmhead_fwd_kernel")
                        : launch_tiles(lmhead_fwd_kernel<false>, blocks, a, s, "lmhead_fwd_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(lmhead_fwd_finish_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, n, cols, (int)vocab, a.nsplit,
                     (int64_t)a.mt * BM, a.part, a.ysel, input_ids, new_logprobs, entropy, lse2);
  PRL_LAUNCH_CHECK("lmhead_fwd_finish_kernel");
  return PRL_OK;
}

extern "C" int prl_lm_head_logprob_bwd(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                       const uint16_t* w_hi, const uint16_t* w_lo, const uint16_t* wt_hi,
                                       const uint16_t* wt_lo, const int64_t* input_ids, float temperature,
                                       const float* lse2, const float* entropy, const float* grad_new_logprobs,
                                       const float* grad_entropy, const float* upstream, void* grad_hidden,
                                       int32_t grad_hidden_dtype, float* grad_weight, int64_t chunk_rows, void* workspace,
                                       size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(hidden >= BK && hidden % BK == 0, "hidden size %lld must be a multiple of %d", (long long)hidden, BK);
  PRL_CHECK_ARG(vocab >= BK && vocab % BK == 0 && vocab < ((int64_t)1 << 31),
                "the fused backward contracts over the vocabulary: vocab %lld must be a multiple of %d", (long long)vocab, BK);
  PRL_CHECK_ARG(rows * cols < ((int64_t)1 << 31) - BM, "too many rows");
  PRL_CHECK_ARG(hidden_bf16 && w_hi && wt_hi && input_ids && lse2 && entropy && grad_new_logprobs && workspace, "null pointer");
  PRL_CHECK_ARG((w_lo == nullptr) == (wt_lo == nullptr), "w_lo and wt_lo go together");
  PRL_CHECK_ARG(grad_hidden || grad_weight, "nothing to compute");
  PRL_CHECK_ARG(grad_hidden_dtype == PRL_DTYPE_F32 || grad_hidden_dtype == PRL_DTYPE_BF16, "unsupported grad_hidden dtype");
  PRL_CHECK_ARG(temperature > 0.0f, "temperature must be > 0");
  PRL_CHECK_ARG(chunk_rows >= 1, "chunk_rows must be >= 1");
  const int64_t n = rows * cols;
  if (chunk_rows > n) chunk_rows = n;
  const BwdLayout L = bwd_layout(hidden, vocab, chunk_rows);
  if (workspace_bytes < L.total) return prl::set_error(PRL_ENOMEM, "lm_head backward workspace: %zu bytes given, %zu needed", workspace_bytes, L.total);
  char* ws = static_cast<char*>(workspace);
  uint16_t* hT = reinterpret_cast<uint16_t*>(ws + L.hT);
  uint16_t* dl_hi = reinterpret_cast<uint16_t*>(ws + L.dl_hi);
  uint16_t* dl_lo = reinterpret_cast<uint16_t*>(ws + L.dl_lo);
  uint16_t* dlT_hi = reinterpret_cast<uint16_t*>(ws + L.dlT_hi);
  uint16_t* dlT_lo = reinterpret_cast<uint16_t*>(ws + L.dlT_lo);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool glds = g_use_glds() != 0;
  const int nt = ceil_div(vocab, BN);
  const int kt = ceil_div(hidden, BN);

  for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
    const int64_t m = (n - r0) < chunk_rows ? (n - r0) : chunk_rows;
    const int mt = ceil_div(m, BM);
    // ---- 1. d logits planes of this chunk (recompute the logits tile by tile)
    DlArgs d;
    d.terms.n = w_lo ? 2 : 1;
    for (int k = 0; k < MAX_TERMS; ++k) {
      d.terms.a[k] = hidden_bf16 + r0 * hidden;
      d.terms.b[k] = (k == 1 && w_lo) ? w_lo : w_hi;
    }
    d.geo = Geom{(int)m, (int)vocab, (int)hidden, hidden, hidden};
    d.row_base = r0;
    d.n_total = n;
    d.cols = cols;
    d.ids = input_ids;
    d.lse2 = lse2;
    d.ent = entropy;
    d.g_nlp = grad_new_logprobs;
    d.g_ent = grad_entropy;
    d.upstream = upstream;
    d.k2 = kLog2e / temperature;
    d.inv_temp = 1.0f / temperature;
    d.mt = L.chunk_pad / BM;  // pad rows of the chunk buffers are written too (zeros)
    d.nt = nt;
    d.chunk_pad = L.chunk_pad;
    d.dl_hi = dl_hi;
    d.dl_lo = dl_lo;
    d.dlT_hi = dlT_hi;
    d.dlT_lo = dlT_lo;
    int rc = glds ? launch_tiles(lmhead_dlogits_kernel<true>, d.mt * d.nt, d, s, "lmhead_dlogits_kernel")
                  : launch_tiles(lmhead_dlogits_kernel<false>, d.mt * d.nt, d, s, "lmhead_dlogits_kernel");
    if (rc) return rc;
    // ---- 2. d hidden[chunk] = dl W  (contraction over the vocabulary; hi x hi + lo x hi + hi x lo)
    if (grad_hidden) {
      GemmArgs g;
      g.terms.n = w_lo ? 3 : 2;
      g.terms.a[0] = dl_hi;
      g.terms.b[0] = wt_hi;
      g.terms.a[1] = dl_lo;
      g.terms.b[1] = wt_hi;
      g.terms.a[2] = dl_hi;
      g.terms.b[2] = wt_lo ? wt_lo : wt_hi;
      g.geo = Geom{(int)m, (int)hidden, (int)vocab, vocab, vocab};
      g.mt = mt;
      g.nt = kt;
      g.ldc = hidden;
      g.out_bf16 = grad_hidden_dtype == PRL_DTYPE_BF16;
      g.accumulate = 0;
      g.out = static_cast<char*>(grad_hidden) + (size_t)r0 * hidden * (g.out_bf16 ? 2 : 4);
      rc = glds ? launch_tiles(gemm_nt_kernel<true>, g.mt * g.nt, g, s, "gemm_nt_kernel(d hidden)")
                : launch_tiles(gemm_nt_kernel<false>, g.mt * g.nt, g, s, "gemm_nt_kernel(d hidden)");
      if (rc) return rc;
    }
    // ---- 3. d W += dl^T h  (contraction over the chunk's rows; hidden is exact in bf16)
    if (grad_weight) {
      const dim3 tg((unsigned)ceil_div(hidden, 64), (unsigned)ceil_div(L.chunk_pad, 64));
      hipLaunchKernelGGL((split_transpose_kernel<uint16_t>), tg, dim3(256), 0, s, m, hidden, hidden_bf16 + r0 * hidden,
                         (uint16_t*)nullptr, (uint16_t*)nullptr, hT, (uint16_t*)nullptr, (int64_t)L.chunk_pad);
      PRL_LAUNCH_CHECK("split_transpose_kernel(hidden)");
      GemmArgs g;
      g.terms.n = 2;
      g.terms.a[0] = dlT_hi;
      g.terms.a[1] = dlT_lo;
      g.terms.a[2] = dlT_lo;
      g.terms.b[0] = g.terms.b[1] = g.terms.b[2] = hT;
      g.geo = Geom{(int)vocab, (int)hidden, L.chunk_pad, L.chunk_pad, L.chunk_pad};
      g.mt = nt;
      g.nt = kt;
      g.ldc = hidden;
      g.out_bf16 = 0;
      g.accumulate = 1;
      g.out = grad_weight;
      rc = glds ? launch_tiles(gemm_nt_kernel<true>, g.mt * g.nt, g, s, "gemm_nt_kernel(d weight)")
                : launch_tiles(gemm_nt_kernel<false>, g.mt * g.nt, g, s, "gemm_nt_kernel(d weight)");
      if (rc) return rc;
    }
  }
  return PRL_OK;
}
