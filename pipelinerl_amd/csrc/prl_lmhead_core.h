// Fused output head, shared device code: tile configurations, the MFMA main loops (staging through the LDS DMA, fragment reads,
// hand-placed instruction streams) and the launch helpers of csrc/prl_lmhead_fwd.hip / prl_lmhead_bwd.hip / prl_lmhead_prepare.hip.
// Only what SHIPS lives here.  Every structure and schedule that a committed A/B showed losing (the one-wave-per-SIMD tiles
// 256 x 384 / 256 x 320, the d W hand-placed and phase-shifted streams, the round-2 / round-3 schedules kept as A/B references,
// the f16 + MX-fp8 mixed-precision core - 1.4e-4 on a PPO loss, outside the 1e-4 bar) was removed in round 5; the code is kept,
// unbuilt, in scripts/exp/prl_lmhead_round4_all_variants.hip, the measurements in profiles/r02* .. r04* (DESIGN.md §3b).
//
// Generic GEMM core (small shapes, single-plane 256 x 256): C[M, N] = sum_terms A_t[M, Kc] B_t[N, Kc]^T, both operands
// contraction-contiguous bf16; the large two- / three-plane shapes run on the dual- / triple-plane cores (256 x 256 x 32, planes
// share the staged partner tile).
//   tile     BM x 128 x 64 with BM = 256 (512 threads, 8 waves as 4 x 2) or 128 (256 threads, 2 x 2);
//            every wave computes 64 x 64 as 2 x 2 tiles of v_mfma_f32_32x32x16_bf16
//   staging  HBM/L2 -> LDS by global_load_lds (16 B per lane, no VGPR round trip), 16-byte chunks
//            XOR-swizzled on the SOURCE address so the fragment ds_read_b128 are bank-conflict free
//   pipeline BM = 256: ring of 3 LDS stages (144 KB), two tiles of loads in flight; the wait for a tile
//            is a COUNTED s_waitcnt vmcnt(N) and the barrier a raw s_barrier, so the younger tile's
//            loads stay in flight across it (a `__syncthreads()` would drain them)
// MFMA-bound by design; roofline = dense bf16 MFMA peak (2.5 PFLOP/s).
#pragma once

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "prl_common.h"
#include "prl_lmhead_layout.h"
#include "prl_osm.h"

namespace prl {
namespace lmhead {

using namespace prl::osm;

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_TERMS = 3;

// Workgroup shape: BM x BN tile, (BM / 64) x 2 waves of 64 x (BN / 2), STAGES LDS buffers of (BM + BN) rows
// x 128 bytes.  The staging traffic, not the matrix pipe, bounds these kernels: the LDS DMA sustains about
// 24 bytes per clock and CU out of L2 (measured, profiles/r02c), and a K-step moves (BM + BN) * 128 bytes
// for 2 * BM * BN * 64 flop - 87 flop/byte at 256 x 128 (measured 1.0 PFLOP/s), 131 at 256 x 256.
template <int BM_, int BN_, int STAGES_>
struct Cfg {
  static constexpr int BM = BM_;
  static constexpr int BN = BN_;
  static constexpr int NT = BM_ * 2;  // 64 threads per 32 rows: 8 waves for 256 rows, 4 for 128
  static constexpr int STAGES = STAGES_;
  static constexpr int NJ = BN_ / 64;      // 32-column MFMA tiles per wave (2 x NJ tiles of 32 x 32)
  static constexpr int WCOLS = BN_ / 2;    // columns per wave
  static constexpr int QA = BM_ * 8 / NT;  // 16-byte chunks per thread and stage, A tile
  static constexpr int QB = BN_ * 8 / NT;  //                                       B tile
  static constexpr int LOADS = QA + QB;
  static constexpr int A_BYTES = BM_ * ROW_BYTES;
  static constexpr int STAGE_BYTES = (BM_ + BN_) * ROW_BYTES;
  static constexpr int LDS_BYTES = STAGES_ * STAGE_BYTES;
};
using CfgWide = Cfg<256, 256, 2>;   // 128 KB LDS, one workgroup of 8 waves per CU, 64 x 128 per wave
using CfgBig = Cfg<256, 128, 3>;    // 144 KB LDS, one workgroup of 8 waves per CU, two tiles of loads in flight
using CfgSmall = Cfg<128, 128, 2>;  // 64 KB LDS, two workgroups of 4 waves per CU

// Dual-plane shape: C += (A1 + A2) B^T with the two A planes (W_hi / W_lo, or d logits hi / lo) sharing ONE
// staged B tile.  256 x 256 tile, 32-deep stages of three 16 KB tiles (A1, A2, B) in a ring of 3: 48 KB per
// 32 MFMAs per wave instead of 64 KB (the staging path is what bounds the 256 x 256 shape), two stages of loads
// in flight instead of one, 8 instead of 12 fragment reads per 16 MFMAs.
struct CfgDual {
  static constexpr int BM = 256, BN = 256, NT = 512, STAGES = 3;
  static constexpr int NJ = 4, WCOLS = 128;
  static constexpr int Q = 2;                       // 16-byte chunks per thread, tile and stage
  static constexpr int LOADS = 3 * Q;
  static constexpr int TILE_BYTES = 256 * ROW_BYTES32;  // 16 KB
  static constexpr int STAGE_BYTES = 3 * TILE_BYTES;    // 48 KB
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;  // 144 KB
};

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

// Wait until at most N of this wave's LDS-DMA loads are outstanding (they complete in order, so the
// older tile has landed), retire this wave's own LDS reads, then the workgroup barrier.  One asm
// statement with a memory clobber: the compiler can neither drain the younger loads with a vmcnt(0)
// (what `__syncthreads()` does while an LDS DMA is in flight) nor move LDS traffic across the barrier.
template <int N>
__device__ __forceinline__ void wait_tile_then_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// An MFMA as volatile asm that also clobbers "memory": no LDS read and no LDS-DMA issue moves across it - the instruction stream
// around the MFMAs is the one written in the source (the hand-placed streams below).  The compiler's hazard recogniser does not
// see inside asm: `mfma_settle()` supplies the wait states an MFMA result needs before anything but another MFMA touches it (the
// static check of the generated ISA for exactly this class of hazard is scripts/check_mfma_hazards.py, run as a CPU test).
__device__ __forceinline__ void mfma_bf16_pinned(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}
__device__ __forceinline__ void mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
// The other direction: an accumulator register written by a VALU instruction (the caller's zero fill, a copy) needs wait
// states before an MFMA reads it as SrcC - and hipcc, which does not see the asm MFMAs, likes to sink the zero fill of a
// tile right in front of the first instruction that uses it (found the hard way: `v_mov_b64 v[56:57], 0` immediately
// followed by the first MFMA into v[56:71] left ONE accumulator register of one tile with its stale contents -
// scripts/exp/lmhead_ps_debug.py).  `mfma_pin_acc` makes every tile pass through an opaque asm AT LOOP ENTRY - the fill must
// be complete there - and the pipeline fill that follows (LDS-DMA issue, barrier, fragment reads) puts hundreds of clocks
// between it and the first MFMA; from then on the tiles only flow from asm to asm.
template <int NI, int NJ>
__device__ __forceinline__ void mfma_pin_acc(f32x16 (&acc)[NI][NJ]) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[i][j]));
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
}

// Source address of one LDS-DMA piece: `base` is wave-uniform (a tile's first row at the current contraction offset), `lane_bytes` the
// lane's 32-bit byte offset inside the tile's rows.  The empty asm pins the base in SGPRs and keeps the optimiser from folding the sum
// back into a per-lane 64-bit address (which costs a register pair per stream and a 64-bit VALU add per piece - with those the
// dual-plane kernels spilled): instruction selection then picks `global_load_lds ... v_off, s[base]`.
__device__ __forceinline__ const char* dma_src(const uint16_t* base, uint32_t lane_bytes) {
  const char* ub = reinterpret_cast<const char*>(base);
  asm volatile("" : "+s"(ub));
  return ub + (size_t)lane_bytes;
}

// -----------------------------------------------------------------------------------------------
// main loop: acc[i][j] (32 x 32 tiles of this wave's 64 x 64) += sum over terms and K
// -----------------------------------------------------------------------------------------------
// HAND = false: the compiler places the fragment reads; the LDS-DMA pieces of the next tile ride between the MFMA groups of this
// one, spread over the first three 16-deep sub-steps (a piece costs ~60-180 issue cycles and all waves of the workgroup leave the
// barrier together - one burst would idle the matrix pipe of every SIMD at once: 17.7 ms interleaved vs 19.8 ms burst for the
// 7B forward).  Used by the 128- and 256 x 128 shapes.
// HAND = true (round 4; the single-plane 256 x 256 forward): a hand-placed stream - order-pinning asm MFMAs, one LDS-DMA piece
// after every (MFMAS / LOADS)-th MFMA, the fragment reads of sub-step ks + 1 two per gap after the first MFMAs of sub-step ks
// (7B 7.88 -> 7.53 ms, 32B 11.14 -> 10.56 ms, bit-identical, profiles/r04m_*).
// Measured and not kept on this core (profiles/r02f_*, r02h_*): all pieces after the first group (+-1 %), s_setprio around the
// MFMA groups (-5 %), the sc0 cache-policy bit (0), staggered wave roles (64-deep stages: 256 x 128 forward 16.3 -> 16.7 ms).
template <class C, bool HAND = false>
__device__ __forceinline__ void gemm_mainloop(f32x16 (&acc)[2][C::NJ], const Terms& t, const Geom& g, int m0, int n0,
                                              char* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging (layout: prl_lmhead_layout.h): this thread fetches chunks q * NT + tid of each tile,
  // i.e. rows stage_row(tid, q), all from source column stage_kcol(tid)
  const int kcol = stage_kcol(tid);
  // DMA sources = wave-uniform tile base (SGPRs) + a 32-bit byte offset per lane and stream (`dma_src`): the `saddr + voffset` form
  const int64_t tileA = (int64_t)m0 * g.lda, tileB = (int64_t)n0 * g.ldb;
  uint32_t offA[C::QA], offB[C::QB];
#pragma unroll
  for (int q = 0; q < C::QA; ++q) {
    int ra = m0 + stage_row(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;  // rows past the edge re-read the last row; their results are discarded
    offA[q] = (uint32_t)(((int64_t)(ra - m0) * g.lda + kcol) * 2);
  }
#pragma unroll
  for (int q = 0; q < C::QB; ++q) {
    int rb = n0 + stage_row(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offB[q] = (uint32_t)(((int64_t)(rb - n0) * g.ldb + kcol) * 2);
  }
  // ---- fragment reads: byte offsets of MFMA tile i = 0 for the four 16-deep sub-steps; tile i = 1 adds
  // 32 rows * 128 bytes (the swizzle term depends on (row >> 1) & 7 only, unchanged by + 32)
  int rdA[4], rdB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    rdA[ks] = frag_lds_byte(lane, wm * 64, 0, ks);
    rdB[ks] = C::A_BYTES + frag_lds_byte(lane, wn * C::WCOLS, 0, ks);
  }

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
  // One 16-byte-per-lane LDS-DMA piece `idx` (0 .. LOADS-1: first the A tile's, then the B tile's) of the
  // tile the staging cursor points at, into stage buffer `buf`.
  auto stage_piece = [&](int buf, int idx) {
    const unsigned dst = buf * C::STAGE_BYTES + stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    if (idx < C::QA) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dma_src(sA + tileA + st_k, offA[idx]),
                                       (__attribute__((address_space(3))) void*)(lds + dst + idx * C::NT * 16), 16, 0, 0);
    } else {
      const int q = idx - C::QA;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dma_src(sB + tileB + st_k, offB[q]),
                                       (__attribute__((address_space(3))) void*)(lds + dst + C::A_BYTES + q * C::NT * 16), 16, 0, 0);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int idx = 0; idx < C::LOADS; ++idx) stage_piece(buf, idx);
  };
  // MFMAs of one staged tile; with `sbuf >= 0` the LDS-DMA pieces of the NEXT tile to stage ride in between
  auto compute = [&](int buf, int sbuf) {
    const char* base = lds + buf * C::STAGE_BYTES;
    constexpr int NJ = C::NJ;
    bf16x8 af[2][2], bfr[2][NJ];  // [ping-pong][tile]
#pragma unroll
    for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(base + rdA[0] + i * 32 * ROW_BYTES);
#pragma unroll
    for (int j = 0; j < NJ; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES);
    if constexpr (HAND) {
      constexpr int PER_KS = 2 * NJ, MFMAS = 4 * PER_KS, SPACE = MFMAS / C::LOADS;
      static_assert(SPACE >= 1 && (MFMAS - SPACE / 2 - 1) / SPACE + 1 >= C::LOADS, "not every DMA piece has an MFMA slot");
      int m = 0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            mfma_bf16_pinned(acc[i][j], af[cur][i], bfr[cur][j]);
            const int tpos = i * NJ + j;  // position in this sub-step: the 2 + NJ reads of the next one go two per gap from tpos = 1 on
            if (ks < 3 && tpos >= 1 && 2 * (tpos - 1) < 2 + NJ) {
#pragma unroll
              for (int r = 2 * (tpos - 1); r < 2 * tpos; ++r) {
                if (r < 2) {
                  af[nxt][r] = *reinterpret_cast<const bf16x8*>(base + rdA[ks + 1] + r * 32 * ROW_BYTES);
                } else if (r < 2 + NJ) {
                  bfr[nxt][r - 2] = *reinterpret_cast<const bf16x8*>(base + rdB[ks + 1] + (r - 2) * 32 * ROW_BYTES);
                }
              }
            }
            if (sbuf >= 0 && m % SPACE == SPACE / 2 && m / SPACE < C::LOADS) stage_piece(sbuf, m / SPACE);
            ++m;
          }
      }
    } else {
      constexpr int GROUPS = 3;
      constexpr int PER = (C::LOADS + GROUPS - 1) / GROUPS;  // pieces after each of the first GROUPS MFMA groups
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {  // the reads of sub-step ks + 1 are issued before the MFMAs of ks
#pragma unroll
          for (int i = 0; i < 2; ++i) af[nxt][i] = *reinterpret_cast<const bf16x8*>(base + rdA[ks + 1] + i * 32 * ROW_BYTES);
#pragma unroll
          for (int j = 0; j < NJ; ++j) bfr[nxt][j] = *reinterpret_cast<const bf16x8*>(base + rdB[ks + 1] + j * 32 * ROW_BYTES);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
        if (sbuf >= 0 && ks < GROUPS) {
#pragma unroll
          for (int k = 0; k < PER; ++k)
            if (ks * PER + k < C::LOADS) stage_piece(sbuf, ks * PER + k);
        }
      }
    }
  };

  if constexpr (HAND) mfma_pin_acc(acc);
  constexpr int D = C::STAGES - 1;  // tiles of loads in flight ahead of the one being computed
  __syncthreads();                  // whoever used the LDS before (previous tile, an epilogue) is done with it
#pragma unroll
  for (int p = 0; p < D; ++p)
    if (p < total) {
      stage(p);
      advance();
    }
  int cur = 0, nxt = D % C::STAGES;
  int s = 0;
  // steady state: tile s has landed once only the D - 1 younger stages remain outstanding; every wave has
  // finished computing tile s - 1 when it passes the barrier, so that tile's buffer (where tile s + D goes)
  // is free - its loads ride between the MFMAs of tile s
  for (; s + D < total; ++s) {
    wait_tile_then_barrier<(D - 1) * C::LOADS>();
    compute(cur, nxt);
    advance();
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
    nxt = nxt + 1 == C::STAGES ? 0 : nxt + 1;
  }
  // drain: the last D tiles, nothing left to stage
  for (; s < total; ++s) {
    if (s + D - 1 < total) {
      wait_tile_then_barrier<(D - 1) * C::LOADS>();
    } else {
      wait_tile_then_barrier<0>();
    }
    compute(cur, -1);
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
  }
  if constexpr (HAND) mfma_settle();
}

// -----------------------------------------------------------------------------------------------
// dual-plane main loop: acc += (A1 + A2)[m0.., :] B[n0.., :]^T over Kc, 32-deep stages (see CfgDual)
// -----------------------------------------------------------------------------------------------
// The barrier-at-the-step-start form; the large launches take the phase-shifted loop below and fall back to this one for
// contractions shorter than six stages.  A hand-placed stream, every wave alike: order-pinning asm MFMAs, ONE LDS-DMA piece after
// every 5th MFMA (m = 2, 7, .. 27), the second half's fragment reads two per gap after MFMAs 3-6.  A piece costs its wave ~60
// clocks of issue among bare MFMAs and 100-185 inside a burst (MI355X_MICROARCH.md).
// History of this loop (all measured on the 7B forward, bit-identical outputs; code in scripts/exp/prl_lmhead_round4_all_variants.hip):
// pieces interleaved with the MFMA groups, compiler-placed reads (round 2) 14.7 ms; staggered wave roles - waves 4-7 issue the next
// tile's DMA right after the barrier, waves 0-3 after their MFMA cluster (round 3) 13.8 ms; this stream 13.0-13.2 ms
// (profiles/r04i_*, r04j_*).  Not kept: chaining the forward's vocabulary tiles so the pipeline never refills (17.2 vs 14.6 ms: the
// refill keeps the 32 workgroups of an XCD in step, and in step they share every weight and hidden tile in the XCD's L2; chained
// they drift apart, L2 misses 137 M -> 669 M requests per forward, profiles/r02p_*).
__device__ __forceinline__ void gemm_mainloop_dual(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                   const Geom& g, int m0, int n0, char* lds) {
  using C = CfgDual;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  const int64_t tileA = (int64_t)m0 * g.lda, tileB = (int64_t)n0 * g.ldb;
  uint32_t offA[C::Q], offB[C::Q];
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int ra = m0 + stage_row32(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offA[q] = (uint32_t)(((int64_t)(ra - m0) * g.lda + kcol) * 2);
    offB[q] = (uint32_t)(((int64_t)(rb - n0) * g.ldb + kcol) * 2);
  }
  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    rdA[ks] = frag_lds_byte32(lane, wm * 64, 0, ks);
    rdB[ks] = 2 * C::TILE_BYTES + frag_lds_byte32(lane, wn * C::WCOLS, 0, ks);
  }
  const int total = g.Kc / BK32;
  int st_k = 0;  // contraction offset of the NEXT tile to stage
  auto stage_piece = [&](int buf, int idx) {  // idx 0..5: A1 q0 q1, A2 q0 q1, B q0 q1
    const unsigned dst = buf * C::STAGE_BYTES + stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    const int tile = idx / C::Q, q = idx % C::Q;
    const char* src = tile == 0 ? dma_src(A1 + tileA + st_k, offA[q]) : tile == 1 ? dma_src(A2 + tileA + st_k, offA[q]) : dma_src(B + tileB + st_k, offB[q]);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + dst + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
  };
  // NEXT (a type tag): whether a stage is issued during this step
  auto compute = [&](int buf, int sbuf, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    const char* base = lds + buf * C::STAGE_BYTES;
    bf16x8 a1[2][2], a2[2][2], bfr[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a1[0][i] = *reinterpret_cast<const bf16x8*>(base + rdA[0] + i * 32 * ROW_BYTES32);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES32);
#pragma unroll
    for (int i = 0; i < 2; ++i) a2[0][i] = *reinterpret_cast<const bf16x8*>(base + C::TILE_BYTES + rdA[0] + i * 32 * ROW_BYTES32);
    int m = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mfma_bf16_pinned(acc[i][j], pl == 0 ? a1[ks][i] : a2[ks][i], bfr[ks][j]);
            // the eight fragment reads of the second half two per MFMA gap (after MFMAs 3, 4, 5, 6): all eight in one gap
            // cost 1.4 % (13.16 -> 12.97 ms, profiles/r04i_*)
            if (m == 3) {
              a1[1][0] = *reinterpret_cast<const bf16x8*>(base + rdA[1]);
              bfr[1][0] = *reinterpret_cast<const bf16x8*>(base + rdB[1]);
            } else if (m == 4) {
              bfr[1][1] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + 32 * ROW_BYTES32);
              bfr[1][2] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + 2 * 32 * ROW_BYTES32);
            } else if (m == 5) {
              bfr[1][3] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + 3 * 32 * ROW_BYTES32);
              a1[1][1] = *reinterpret_cast<const bf16x8*>(base + rdA[1] + 32 * ROW_BYTES32);
            } else if (m == 6) {
              a2[1][0] = *reinterpret_cast<const bf16x8*>(base + C::TILE_BYTES + rdA[1]);
              a2[1][1] = *reinterpret_cast<const bf16x8*>(base + C::TILE_BYTES + rdA[1] + 32 * ROW_BYTES32);
            }
            constexpr int PH = 2;  // the piece goes after the 3rd of each run of five MFMAs (after the 1st: +4.7 %, after the 5th: +0.3 %)
            if (NEXT && m % 5 == PH && m / 5 < C::LOADS) stage_piece(sbuf, m / 5);
            ++m;
          }
      }
    }
  };

  mfma_pin_acc(acc);
  constexpr int D = C::STAGES - 1;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < D; ++p)
    if (p < total) {
#pragma unroll
      for (int idx = 0; idx < C::LOADS; ++idx) stage_piece(p, idx);
      st_k += BK32;
    }
  int cur = 0, nxt = D % C::STAGES;
  int s = 0;
  for (; s + D < total; ++s) {
    wait_tile_then_barrier<(D - 1) * C::LOADS>();
    compute(cur, nxt, std::integral_constant<bool, true>{});
    st_k += BK32;
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
    nxt = nxt + 1 == C::STAGES ? 0 : nxt + 1;
  }
  for (; s < total; ++s) {
    if (s + D - 1 < total) {
      wait_tile_then_barrier<(D - 1) * C::LOADS>();
    } else {
      wait_tile_then_barrier<0>();
    }
    compute(cur, -1, std::integral_constant<bool, false>{});
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
  }
  mfma_settle();
}

// -----------------------------------------------------------------------------------------------
// dual-plane main loop, PHASE-SHIFTED: the workgroup barrier sits in the MIDDLE of the step
// -----------------------------------------------------------------------------------------------
// In gemm_mainloop_dual every step begins behind a barrier with eight fragment reads in front of an idle matrix pipe (the
// reads cannot be issued earlier: the barrier is what certifies the stage).  Here the ONE barrier of step s - "A(s)" - sits
// between its two 16-MFMA halves and certifies stage s + 1: the first half's fragments of stage s + 1 are then read during
// the SECOND half of step s (into the registers the first half just released) and step s + 1 starts with MFMAs.  Ring of 3:
//   A(s) = s_waitcnt vmcnt(6) lgkmcnt(0); s_barrier   - every wave's pieces of stage s + 1 have landed (the six pieces of
//          stage s + 2, issued since A(s - 1), may still be out), and every wave's reads of stage s are complete (its second
//          half's were issued at MFMAs 3-6 of this step) -> the buffer of stage s is free
//   pieces issued between A(s) and A(s + 1) (MFMAs 17, 22, 27 of step s; 2, 7, 12 of step s + 1) go to stage s + 3, into the
//          buffer stage s just left; they have a whole step to land before A(s + 2) needs them
// The stream is the hand-placed one of gemm_mainloop_dual (order-pinning asm MFMAs, one piece per five MFMAs, reads two per gap).
// Needs at least 6 stages of contraction (shorter ones take gemm_mainloop_dual).  Measured, same box, interleaved, bit-identical
// outputs (profiles/r04p_*): 7B forward 13.87 (round-3 schedule) -> 13.25 (barrier first) -> 12.87 ms; 32B 19.85 -> 19.02 -> 18.06 ms.
// (Round 2 tried the mid-step barrier on the compiler-scheduled stream and lost 3 %; with every read and DMA issue pinned
// between specific MFMAs it is the other way round.  On the d W product - transposed A operand - the same step measured SLOWER,
// 21.1 vs 16.6 ms, profiles/r04t_*: that kernel keeps gemm_mainloop_dual_tr.)
template <bool P1, int BAR, bool RD, bool P2>
struct PsFlags {
  static constexpr bool p1 = P1, rd = RD, p2 = P2;
  static constexpr int bar = BAR;  // -1: no barrier, else the vmcnt of A(s)
};

__device__ __forceinline__ void gemm_mainloop_dual_ps(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                      const Geom& g, int m0, int n0, char* lds) {
  using C = CfgDual;
  const int total = g.Kc / BK32;
  if (total < 6) {
    gemm_mainloop_dual(acc, A1, A2, B, g, m0, n0, lds);
    return;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  // DMA sources as (wave-uniform tile base in SGPRs) + (32-bit per-lane byte offset inside the tile's rows): the loads take the
  // `saddr + voffset` form, a stream costs ONE register instead of a 64-bit pair that has to be advanced per stage - with 64-bit
  // per-lane addresses the kernel sat at the 256-register limit and reloaded 16 spilled address pairs from scratch, three of
  // them inside the contraction loop, where a scratch load also counts against the hand-counted vmcnt waits of the DMA pipeline
  const char* baseA1 = reinterpret_cast<const char*>(A1 + (int64_t)m0 * g.lda);
  const char* baseA2 = reinterpret_cast<const char*>(A2 + (int64_t)m0 * g.lda);
  const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * g.ldb);
  uint32_t offA[C::Q], offB[C::Q];
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int ra = m0 + stage_row32(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;
    offA[q] = (uint32_t)(((int64_t)(ra - m0) * g.lda + kcol) * 2);
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offB[q] = (uint32_t)(((int64_t)(rb - n0) * g.ldb + kcol) * 2);
  }
  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    rdA[ks] = frag_lds_byte32(lane, wm * 64, 0, ks);
    rdB[ks] = 2 * C::TILE_BYTES + frag_lds_byte32(lane, wn * C::WCOLS, 0, ks);
  }
  // piece idx (0..5: A1 q0 q1, A2 q0 q1, B q0 q1) of the stage at contraction offset `stage_k` into ring buffer `buf`
  auto stage_piece = [&](int buf, int stage_k, int idx) {
    const unsigned dst = buf * C::STAGE_BYTES + stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    const int tile = idx / C::Q, q = idx % C::Q;
    const char* src = dma_src(reinterpret_cast<const uint16_t*>(tile == 0 ? baseA1 : tile == 1 ? baseA2 : baseB) + stage_k, tile == 2 ? offB[q] : offA[q]);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + dst + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
  };
  bf16x8 a1[2][2], a2[2][2], bfr[2][4];  // [half][tile]
  auto rd = [&](const char* base, int off) { return *reinterpret_cast<const bf16x8*>(base + off); };
  // fragment of plane `pl` (0: A1, 1: A2), row tile i, sub-step ks of the stage at `base`
  auto ldA = [&](const char* base, int pl, int i, int ks) -> bf16x8 { return rd(base, pl * C::TILE_BYTES + rdA[ks] + i * 32 * ROW_BYTES32); };
  auto ldB = [&](const char* base, int j, int ks) { return rd(base, rdB[ks] + j * 32 * ROW_BYTES32); };
  // the eight fragments of half `ks` of the stage at `base`, two LDS reads per MFMA gap, gap 0 = after the half's 4th MFMA
  auto read_gap = [&](const char* base, int ks, int gap) {
    if (gap == 0) {
      a1[ks][0] = ldA(base, 0, 0, ks);
      bfr[ks][0] = ldB(base, 0, ks);
    } else if (gap == 1) {
      bfr[ks][1] = ldB(base, 1, ks);
      bfr[ks][2] = ldB(base, 2, ks);
    } else if (gap == 2) {
      bfr[ks][3] = ldB(base, 3, ks);
      a1[ks][1] = ldA(base, 0, 1, ks);
    } else if (gap == 3) {
      a2[ks][0] = ldA(base, 1, 0, ks);
      a2[ks][1] = ldA(base, 1, 1, ks);
    }
  };

  mfma_pin_acc(acc);
  __syncthreads();  // whoever used the LDS before is done with it
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int idx = 0; idx < C::LOADS; ++idx) stage_piece(t, t * BK32, idx);
  wait_tile_then_barrier<2 * C::LOADS>();  // stage 0 has landed
#pragma unroll
  for (int i = 0; i < 2; ++i) a1[0][i] = ldA(lds, 0, i, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) bfr[0][j] = ldB(lds, j, 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) a2[0][i] = ldA(lds, 1, i, 0);

  // buffers: b0 = stage s, b1 = stage s + 1, b2 = stage s + 2 (= where the first-half pieces of stage s + 2 go);
  // stage s + 3 goes into b0 after A(s)
  auto step = [&](int s, int b0, int b1, int b2, auto flags) {
    using F = decltype(flags);
    const char* base = lds + b0 * C::STAGE_BYTES;
    const char* next = lds + b1 * C::STAGE_BYTES;
    int m = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1 && F::bar >= 0) {
        if constexpr (F::bar == 0) {
          wait_tile_then_barrier<0>();
        } else {
          wait_tile_then_barrier<C::LOADS>();
        }
      }
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mfma_bf16_pinned(acc[i][j], pl == 0 ? a1[h][i] : a2[h][i], bfr[h][j]);
            if (h == 0) {  // second half's fragments of THIS stage (gaps after MFMAs 3-6; TR: 3-6, 8, 9)
              if (m >= 3 && m <= 9) read_gap(base, 1, m - 3);
              if (F::p1 && m % 5 == 2) stage_piece(b2, (s + 2) * BK32, 3 + m / 5);  // m = 2, 7, 12 -> pieces 3, 4, 5 of stage s + 2
            } else {  // first half's fragments of the NEXT stage (certified by the barrier above)
              if (F::rd && m >= 19 && m <= 25) read_gap(next, 0, m - 19);
              if (F::p2 && m % 5 == 2) stage_piece(b0, (s + 3) * BK32, (m - 17) / 5);  // m = 17, 22, 27 -> pieces 0, 1, 2 of stage s + 3
            }
            ++m;
          }
    }
  };
  int b0 = 0, b1 = 1, b2 = 2;
  auto rotate = [&]() {
    const int t = b0;
    b0 = b1;
    b1 = b2;
    b2 = t;
  };
  // s = 0: the prologue issued stage 2 completely
  step(0, b0, b1, b2, PsFlags<false, C::LOADS, true, true>{});
  rotate();
  int s = 1;
  for (; s + 3 < total; ++s) {  // stages s + 2 and s + 3 exist
    step(s, b0, b1, b2, PsFlags<true, C::LOADS, true, true>{});
    rotate();
  }
  // s = total - 3: stage s + 2 is the last one
  step(s, b0, b1, b2, PsFlags<true, C::LOADS, true, false>{});
  rotate();
  ++s;
  // s = total - 2: nothing left to issue; A(s) waits for the last stage (nothing younger is out)
  step(s, b0, b1, b2, PsFlags<false, 0, true, false>{});
  rotate();
  ++s;
  // s = total - 1
  step(s, b0, b1, b2, PsFlags<false, -1, false, false>{});
  mfma_settle();
}

// -----------------------------------------------------------------------------------------------
// dual-plane main loop with a TRANSPOSED A operand (d W from the row-major d-logits planes)
// -----------------------------------------------------------------------------------------------
// acc[v][n] += sum over t of (A1 + A2)[t][m0 + v] B[n0 + n][t]: A1 / A2 are [K, lda] row-major with the contraction index
// as their ROW (d logits hi / lo, [tokens, vocabulary]), B is contraction-contiguous as everywhere else.  The A tiles are
// staged as they lie in memory (512-byte row segments) and the fragments come out of ds_read_b64_tr_b16 (layout header).
// Same stage geometry, ring and wave roles as the dual-plane core.
// Schedule: staggered wave roles - the two waves that share a SIMD (w and w + 4) do their non-matrix work at opposite ends of the
// step (waves 4-7 issue the next stage's six DMA pieces in one burst right after the barrier, then run their MFMA cluster; waves
// 0-3 the other way round), compiler-placed reads.  What this kernel rewards is the BURST: the 512-byte row segments of a stage
// (rows 304 KB apart in the planes) requested together - its A planes stream from HBM (5 GB per 8192 rows, L2 hit 80 %).
// Measured against it on the 7B shape and removed (code: scripts/exp/prl_lmhead_round4_all_variants.hip): the forward's hand-placed
// stream (19.1 vs 16.7 ms, profiles/r04k_*), the phase-shifted step (21.1 vs 16.6, r04t_*), hand-placed reads WITH the burst
// (16.80-16.92 vs 16.81-16.94), every wave bursting right after the barrier (18.7-19.1) or after its MFMAs (17.1-17.2) (r04v_*), a
// block-tiled plane layout (timing-only: 16.67-16.74 vs 16.73-16.90, r04w_*), staging global -> VGPR -> LDS instead of the LDS DMA
// (18.3-18.5 vs 16.4-16.6, r04vs_*).
__device__ __forceinline__ void gemm_mainloop_dual_tr(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                      const Geom& g, int m0, int n0, char* lds) {
  using C = CfgDual;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  const int64_t tileB = (int64_t)n0 * g.ldb;
  uint32_t offA[C::Q], offB[C::Q];  // lane offsets in bytes from the tile's uniform base (`dma_src`)
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int v = m0 + 8 * tr_stage_chunk(tid, q, C::NT);
    v = v + 8 <= g.M ? v : g.M - 8;  // entries past the edge re-read the last eight; their results are discarded (M % 8 == 0)
    offA[q] = (uint32_t)(((int64_t)tr_stage_row(tid, q, C::NT) * g.lda + (v - m0)) * 2);
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offB[q] = (uint32_t)(((int64_t)(rb - n0) * g.ldb + kcol) * 2);
  }
  // tile i = 1 is 4 chunks further: bit 2 of the chunk index, which the swizzle may flip - an XOR with 64 bytes, not an add
  int rdA[2], rdB[2];
  rdA[0] = tr_frag_lds_byte(lane, wm * 64, 0, 0, 0);
  rdA[1] = tr_frag_lds_byte(lane, wm * 64, 1, 0, 0);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) rdB[ks] = 2 * C::TILE_BYTES + frag_lds_byte32(lane, wn * C::WCOLS, 0, ks);
  const int total = g.Kc / BK32;
  int st_k = 0;  // contraction (token) offset of the NEXT tile to stage
  auto stage_piece = [&](int buf, int idx) {  // idx 0..5: A1 q0 q1, A2 q0 q1, B q0 q1
    const unsigned dst = buf * C::STAGE_BYTES + stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    const int tile = idx / C::Q, q = idx % C::Q;
    const int64_t rowA = (int64_t)st_k * g.lda + m0;
    const char* src = tile == 0 ? dma_src(A1 + rowA, offA[q]) : tile == 1 ? dma_src(A2 + rowA, offA[q]) : dma_src(B + tileB + st_k, offB[q]);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + dst + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
  };
  const bool dma_first = wave >= 4;
  // 8 tokens of one entry: two transposing reads (tokens 0-3: + 0, tokens 4-7: + 4 rows = 2048 bytes); sub-step ks: + 16 rows
  auto a_frag = [&](const char* tile_base, int i, int ks) {
    const char* p0 = tile_base + rdA[i] + ks * (16 * 512);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * 512));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto compute = [&](int buf, int sbuf) {
    const char* base = lds + buf * C::STAGE_BYTES;
    if (sbuf >= 0 && dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a1[2], a2[2], bfr[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a1[i] = a_frag(base, i, ks);
        a2[i] = a_frag(base + C::TILE_BYTES, i, ks);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(base + rdB[ks] + j * 32 * ROW_BYTES32);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], bfr[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (sbuf >= 0 && !dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
  };
  constexpr int D = C::STAGES - 1;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < D; ++p)
    if (p < total) {
#pragma unroll
      for (int idx = 0; idx < C::LOADS; ++idx) stage_piece(p, idx);
      st_k += BK32;
    }
  int cur = 0, nxt = D % C::STAGES;
  int s = 0;
  for (; s + D < total; ++s) {
    wait_tile_then_barrier<(D - 1) * C::LOADS>();
    compute(cur, nxt);
    st_k += BK32;
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
    nxt = nxt + 1 == C::STAGES ? 0 : nxt + 1;
  }
  for (; s < total; ++s) {
    if (s + D - 1 < total) {
      wait_tile_then_barrier<(D - 1) * C::LOADS>();
    } else {
      wait_tile_then_barrier<0>();
    }
    compute(cur, -1);
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
  }
}
// -----------------------------------------------------------------------------------------------
// triple-plane main loop (d hidden): acc += A1 B1^T + A2 B1^T + A1 B2^T over Kc - the three bf16 products of
// (A1 + A2)(B1 + B2) without the 2^-18 lo x lo term - from FOUR staged tiles per 32-deep stage (A1, A2, B1, B2, 16 KB each).
// -----------------------------------------------------------------------------------------------
// The generic core runs the three products one after the other and stages a (A, B) pair per product: 192 KB through the
// LDS DMA per 64 of contraction for 6144 MFMA cycles per SIMD - 32 bytes per clock and CU, more than the DMA path sustains
// (~24, profiles/r02c), so its matrix pipes sat at 55 % (profiles/r02aj).  Here every staged tile feeds two products:
// 128 KB per 64 of contraction = 21 bytes per clock and CU for the same 6144 MFMA cycles.  64 KB per stage leaves room for
// a ring of two (128 KB): the loads of stage s + 1 are issued while stage s is computed (3072 MFMA cycles per SIMD).
struct CfgTriple {
  static constexpr int BM = 256, BN = 256, NT = 512, STAGES = 2;
  static constexpr int WCOLS = 128;
  static constexpr int Q = 2;                       // 16-byte chunks per thread, tile and stage
  static constexpr int LOADS = 4 * Q;
  static constexpr int TILE_BYTES = 256 * ROW_BYTES32;  // 16 KB
  static constexpr int STAGE_BYTES = 4 * TILE_BYTES;    // 64 KB
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;  // 128 KB
};

// A hand-placed stream (round 4): order-pinning asm MFMAs, one LDS-DMA piece after every 6th MFMA (eight pieces, 48 MFMAs), the
// second half's twelve fragment reads two per gap after MFMAs 16-18 and 24-26 (d hidden 24.5 -> 23.8 ms against the round-3
// schedule with staggered wave roles and a burst of eight pieces, bit-identical; profiles/r04l_*).
__device__ __forceinline__ void gemm_mainloop_triple(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B1,
                                                     const uint16_t* B2, const Geom& g, int m0, int n0, char* lds) {
  using C = CfgTriple;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  const int64_t tileA = (int64_t)m0 * g.lda, tileB = (int64_t)n0 * g.ldb;
  uint32_t offA[C::Q], offB[C::Q];  // lane offsets in bytes from the tile's uniform base (`dma_src`)
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int ra = m0 + stage_row32(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offA[q] = (uint32_t)(((int64_t)(ra - m0) * g.lda + kcol) * 2);
    offB[q] = (uint32_t)(((int64_t)(rb - n0) * g.ldb + kcol) * 2);
  }
  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    rdA[ks] = frag_lds_byte32(lane, wm * 64, 0, ks);
    rdB[ks] = 2 * C::TILE_BYTES + frag_lds_byte32(lane, wn * C::WCOLS, 0, ks);
  }
  const int total = g.Kc / BK32;
  int st_k = 0;  // contraction offset of the NEXT tile to stage
  auto stage_piece = [&](int buf, int idx) {  // idx 0..7: A1 q0 q1, A2 q0 q1, B1 q0 q1, B2 q0 q1
    const unsigned dst = buf * C::STAGE_BYTES + stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    const int tile = idx / C::Q, q = idx % C::Q;
    const char* src = tile == 0 ? dma_src(A1 + tileA + st_k, offA[q]) : tile == 1 ? dma_src(A2 + tileA + st_k, offA[q])
                      : tile == 2 ? dma_src(B1 + tileB + st_k, offB[q]) : dma_src(B2 + tileB + st_k, offB[q]);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + dst + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
  };
  auto compute = [&](int buf, int sbuf) {
    const char* base = lds + buf * C::STAGE_BYTES;
    // Products of a stage, MFMA index m: ks0 a1 b1 (0-7), a2 b1 (8-15), a1 b2 (16-23); ks1 the same at 24-47.  The second half's
    // fragments are read INTO THE REGISTERS OF FRAGMENTS THAT ARE DEAD BY THEN (a full second set next to 128 accumulators
    // spills: 56 registers measured): after m = 15 b1 and a2 are dead -> a1', b1' (needed at 24) are read at m = 16-19; after
    // m = 23 a1 and b2 are dead -> a2' (needed at 32) and b2' (needed at 40) are read at m = 24-28.
    bf16x8 a1[2], a2[2], b1[4], b2[4], a1n[2], a2n[2], b1n[4], b2n[4];
    auto rd = [&](int off) { return *reinterpret_cast<const bf16x8*>(base + off); };
#pragma unroll
    for (int i = 0; i < 2; ++i) a1[i] = rd(rdA[0] + i * 32 * ROW_BYTES32);
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = rd(rdB[0] + j * 32 * ROW_BYTES32);
#pragma unroll
    for (int i = 0; i < 2; ++i) a2[i] = rd(C::TILE_BYTES + rdA[0] + i * 32 * ROW_BYTES32);
#pragma unroll
    for (int j = 0; j < 4; ++j) b2[j] = rd(C::TILE_BYTES + rdB[0] + j * 32 * ROW_BYTES32);
    int m = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {  // a1 b1, a2 b1, a1 b2: the MFMAs into one tile are 8 instructions apart
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (ks == 0) {
              mfma_bf16_pinned(acc[i][j], pr == 1 ? a2[i] : a1[i], pr == 2 ? b2[j] : b1[j]);
            } else {
              mfma_bf16_pinned(acc[i][j], pr == 1 ? a2n[i] : a1n[i], pr == 2 ? b2n[j] : b1n[j]);
            }
            if (m == 16) {
              a1n[0] = rd(rdA[1]);
              b1n[0] = rd(rdB[1]);
            } else if (m == 17) {
              b1n[1] = rd(rdB[1] + 32 * ROW_BYTES32);
              b1n[2] = rd(rdB[1] + 2 * 32 * ROW_BYTES32);
            } else if (m == 18) {
              b1n[3] = rd(rdB[1] + 3 * 32 * ROW_BYTES32);
              a1n[1] = rd(rdA[1] + 32 * ROW_BYTES32);
            } else if (m == 24) {
              a2n[0] = rd(C::TILE_BYTES + rdA[1]);
              a2n[1] = rd(C::TILE_BYTES + rdA[1] + 32 * ROW_BYTES32);
            } else if (m == 25) {
              b2n[0] = rd(C::TILE_BYTES + rdB[1]);
              b2n[1] = rd(C::TILE_BYTES + rdB[1] + 32 * ROW_BYTES32);
            } else if (m == 26) {
              b2n[2] = rd(C::TILE_BYTES + rdB[1] + 2 * 32 * ROW_BYTES32);
              b2n[3] = rd(C::TILE_BYTES + rdB[1] + 3 * 32 * ROW_BYTES32);
            }
            if (sbuf >= 0 && m % 6 == 2 && m / 6 < C::LOADS) stage_piece(sbuf, m / 6);
            ++m;
          }
      }
    }
  };
  mfma_pin_acc(acc);
  __syncthreads();  // whoever used the LDS before (previous segment, an epilogue) is done with it
  if (total > 0) {
#pragma unroll
    for (int idx = 0; idx < C::LOADS; ++idx) stage_piece(0, idx);
    st_k += BK32;
  }
  int cur = 0;
  for (int s = 0; s < total; ++s) {
    // ring of two: stage s has landed when NONE of this wave's loads is outstanding (the loads of s + 1 are issued after
    // this barrier); every wave has finished computing stage s - 1 when it passes it, so that buffer is free
    wait_tile_then_barrier<0>();
    compute(cur, s + 1 < total ? (cur ^ 1) : -1);
    st_k += BK32;
    cur ^= 1;
  }
  mfma_settle();
}
// One call site for both cores: DUAL runs the (phase-shifted) dual-plane loop on (terms.a[0], terms.a[1], terms.b[0]); HAND selects
// the hand-placed stream of the generic loop
template <class C, bool DUAL, bool HAND = false>
__device__ __forceinline__ void run_mainloop(f32x16 (&acc)[2][C::NJ], const Terms& t, const Geom& g, int m0, int n0, char* lds) {
  if constexpr (DUAL) {
    gemm_mainloop_dual_ps(acc, t.a[0], t.a[1], t.b[0], g, m0, n0, lds);
  } else {
    gemm_mainloop<C, HAND>(acc, t, g, m0, n0, lds);
  }
}

template <int NJ>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NJ]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
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
// Workgroup shape per launch.  PRL_TUNE_LMHEAD_TILE = 128 | 256 | 512 (= 256 x 256) forces one (a table read, no
// getenv: one process can A/B them); default: the largest tile whose grid still fills the 256 CUs.
enum Shape { kSmall = 0, kBig = 1, kWide = 2 };
inline Shape pick_shape(int64_t m_rows, int64_t n_cols) {
  switch (prl::tuning(PRL_TUNE_LMHEAD_TILE, 0)) {
    case 128: return kSmall;
    case 256: return kBig;
    case 512: return kWide;
    default: break;
  }
  const int64_t m256 = (m_rows + 255) / 256;
  if (m256 * ((n_cols + 255) / 256) >= 200) return kWide;
  if (m256 * ((n_cols + 127) / 128) >= 200) return kBig;
  return kSmall;
}
// The dual-plane core applies when a launch has exactly two terms that share their B operand (W_hi / W_lo
// against the hidden states; d logits hi / lo against the transposed hidden states) and the 256 x 256 shape was chosen.
inline bool use_dual(Shape shape, const Terms& t) { return shape == kWide && t.n == 2 && t.b[0] == t.b[1]; }
inline int shape_bm(Shape s) { return s == kSmall ? 128 : 256; }
inline int shape_bn(Shape s) { return s == kWide ? 256 : 128; }

template <class K, class A>
int launch_tiles(K kfn, int threads, int lds_bytes, int blocks, const A& args, hipStream_t s, const char* name) {
  static thread_local const void* configured[32] = {nullptr};
  const void* key = reinterpret_cast<const void*>(kfn);
  bool seen = false;
  for (auto c : configured) seen = seen || c == key;
  if (!seen) {
    PRL_HIP_CHECK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    for (auto& c : configured)
      if (c == nullptr) {
        c = key;
        break;
      }
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, s, args);
  PRL_LAUNCH_CHECK(name);
  return PRL_OK;
}

#define PRL_LAUNCH_DUAL(KERNEL, blocks, args, s, name) \
  launch_tiles(KERNEL, CfgDual::NT, CfgDual::LDS_BYTES, blocks, args, s, name)

#define PRL_LAUNCH_CFG(shape, KERNEL, blocks, args, s, name)                                                            \
  ((shape) == kWide  ? launch_tiles(KERNEL<CfgWide>, CfgWide::NT, CfgWide::LDS_BYTES, blocks, args, s, name)            \
   : (shape) == kBig ? launch_tiles(KERNEL<CfgBig>, CfgBig::NT, CfgBig::LDS_BYTES, blocks, args, s, name)               \
                     : launch_tiles(KERNEL<CfgSmall>, CfgSmall::NT, CfgSmall::LDS_BYTES, blocks, args, s, name))

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// vocabulary splits of the forward: enough workgroups to fill 256 CUs (x 2 for the small shape)
inline int fwd_nsplit(int token_tiles, int vocab_tiles, bool one_per_cu) {
  {
    const int v = (int)prl::tuning(PRL_TUNE_LMHEAD_NSPLIT, 0);
    if (v >= 1) return v < vocab_tiles ? v : vocab_tiles;
  }
  // workgroups = token_tiles x splits run in rounds of `slots`; pick the split count (up to four rounds) whose last
  // round is fullest - 24 token tiles: 11 splits would be 264 workgroups = a second round for 8 of them, 32 splits
  // are three full rounds.  Ties go to fewer workgroups (longer vocabulary sweeps per workgroup).
  const int slots = one_per_cu ? 256 : 512;
  int best = 1;
  double best_eff = 0.0;
  for (int ns = 1; ns <= vocab_tiles && (int64_t)token_tiles * ns <= 4 * slots; ++ns) {
    const int64_t wg = (int64_t)token_tiles * ns;
    const double eff = (double)wg / (double)(((wg + slots - 1) / slots) * slots);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = ns;
    }
  }
  return best;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace of the backward: hidden^T of the chunk, the two d-logits planes of the chunk, the split-K slices of d hidden
constexpr int kMaxKSplit = 8;
struct BwdLayout {
  int chunk_pad;
  size_t hT, dl_hi, dl_lo, dh_partial, total;
};
inline BwdLayout bwd_layout(int64_t hidden, int64_t vocab, int64_t chunk_rows) {
  BwdLayout L;
  L.chunk_pad = ceil_div(chunk_rows, 128) * 128;
  size_t o = 0;
  L.hT = o;
  o += align256((size_t)hidden * L.chunk_pad * 2);
  const size_t plane = align256((size_t)L.chunk_pad * vocab * 2);
  L.dl_hi = o;
  o += plane;
  L.dl_lo = o;
  o += plane;
  L.dh_partial = o;
  o += align256((size_t)kMaxKSplit * L.chunk_pad * hidden * 4);
  L.total = o;
  return L;
}

}  // namespace lmhead
}  // namespace prl
