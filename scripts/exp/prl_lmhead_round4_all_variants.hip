// Fused output head: hidden states -> log-prob / entropy of the next token with the soft-max inside the GEMM epilogue, and its
// backward.  Two forms: the logits are never written (recomputing backward, 7 plane products per micro-batch),
// or the TRAINING forward also leaves them behind as fp32 for a backward of 5 products (prl_lm_head_logprob_fwd_keep /
// _bwd_kept, the default of fused_head.FusedLmHead); no fp32 d logits and no autograd copies in either.
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
//     products are exact in fp32 and accumulate in fp32, so  W_hi h^T + W_lo h^T  reproduces the
//     fp32 product to ~2^-17 relative - the planes are simply further K-steps of ONE accumulator.
//   * operand roles: the WEIGHT rows (vocabulary) are the M side of the MFMA and the tokens the N
//     side, so in the accumulator layout of v_mfma_f32_32x32x16 (column = lane & 31, rows spread over
//     the 16 registers) a lane owns ONE token per 32 x 32 tile and its registers run along the
//     vocabulary - the soft-max reduction is register-local, and a lane carries two online-softmax
//     states (M, S, W of prl_osm.h) instead of one per accumulator row.
//   * forward: each workgroup owns 128 tokens and a range of vocabulary tiles; the logits never leave
//     the registers.  Partial states per (token, vocabulary split) are merged by a small second kernel
//     that also writes the token-aligned new_logprobs / entropy / lse2.
//   * backward: per chunk of rows the logits are recomputed by the same main loop (or read back from the kept fp32 logits in
//     one elementwise pass), turned into d logits with the saved lse2 / entropy and the per-token loss gradients, split into
//     bf16 (hi, lo) planes and written ROW-MAJOR to a workspace sized for the chunk only;  d hidden = d logits W  runs on a
//     three-product core (gemm_mainloop_triple, one contraction slice per XCD) and  d W += d logits^T hidden  gathers its
//     fragments from the same row-major planes with transposing LDS reads (gemm_mainloop_dual_tr).
//   * opt-in mixed precision (f16 plane + fp8 residual plane on the MX-scaled MFMA): gemm_mainloop_mx.
//
// Generic GEMM core (small shapes, A/B reference): C[M, N] = sum_terms A_t[M, Kc] B_t[N, Kc]^T, both operands
// contraction-contiguous bf16; the large shapes run on the dual- / triple-plane cores below (256 x 256 x 32, planes share
// the staged partner tile).
//   tile     BM x 128 x 64 with BM = 256 (512 threads, 8 waves as 4 x 2) or 128 (256 threads, 2 x 2);
//            every wave computes 64 x 64 as 2 x 2 tiles of v_mfma_f32_32x32x16_bf16
//   staging  HBM/L2 -> LDS by global_load_lds (16 B per lane, no VGPR round trip), 16-byte chunks
//            XOR-swizzled on the SOURCE address so the fragment ds_read_b128 are bank-conflict free
//   pipeline BM = 256: ring of 3 LDS stages (144 KB), two tiles of loads in flight; the wait for a tile
//            is a COUNTED s_waitcnt vmcnt(N) and the barrier a raw s_barrier, so the younger tile's
//            loads stay in flight across it (a `__syncthreads()` would drain them)
// MFMA-bound by design; roofline = dense bf16 MFMA peak (2.5 PFLOP/s).

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "prl_common.h"
#include "prl_lmhead_layout.h"
#include "prl_osm.h"

namespace {

using namespace prl::osm;
using namespace prl::lmhead;

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
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

// MFMAs as volatile asm with the accumulator bound to a register class by hand (used by the one-wave-per-SIMD core below,
// whose 320-384 accumulators the register allocator cannot place, and by the hand-placed streams: MODE 2 of the loops).
// The compiler's hazard recogniser does not see inside asm: `mfma_settle()` supplies the wait states an MFMA result needs
// before anything but another MFMA touches it.
// PIN: the asm also clobbers "memory", i.e. no LDS read and no LDS-DMA issue moves across it - the instruction stream
// around the MFMAs is the one written in the source (the hand-placed interleave of MODE 2 below).
template <bool IN_AGPR, bool PIN = false>
__device__ __forceinline__ void mfma_bf16_asm(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (IN_AGPR && PIN) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b) : "memory");
  } else if constexpr (IN_AGPR) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  } else if constexpr (PIN) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
  } else {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}
__device__ __forceinline__ void mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
// The other direction: an accumulator register written by a VALU instruction (the caller's zero fill, a copy) needs wait
// states before an MFMA reads it as SrcC - and hipcc, which does not see the asm MFMAs, likes to sink the zero fill of a
// tile right in front of the first instruction that uses it (found the hard way: `v_mov_b64 v[56:57], 0` immediately
// followed by the first MFMA into v[56:71] left ONE accumulator register of one tile with its stale contents -
// scripts/exp/lmhead_ps_debug.py).  `mfma_pin_acc` makes every tile pass through an opaque asm AT LOOP ENTRY - the fill must
// be complete there - and the pipeline fill that follows (LDS-DMA issue, barrier, fragment reads) puts hundreds of clocks
// between it and the first MFMA; from then on the tiles only flow from asm to asm.
template <int AGPR_TILES = 0, int NI, int NJ>
__device__ __forceinline__ void mfma_pin_acc(f32x16 (&acc)[NI][NJ]) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (i * NJ + j < AGPR_TILES) {
        asm volatile("" : "+a"(acc[i][j]));
      } else {
        asm volatile("" : "+v"(acc[i][j]));
      }
    }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
}

// -----------------------------------------------------------------------------------------------
// main loop: acc[i][j] (32 x 32 tiles of this wave's 64 x 64) += sum over terms and K
// -----------------------------------------------------------------------------------------------
// EXP (experiment bits of round 2; 0 = the shipped schedule and the only value instantiated for this loop - the ablation
// launches behind profiles/r02h_lmhead_fwd_ablation.txt produced wrong results by design and are no longer built):
//   1  all LDS-DMA pieces right after the first MFMA group instead of spread over three   } measured: +-1 %
//   2  s_setprio(1) around every MFMA group                                                } -5 %
//   4  LDS-DMA loads with the sc0 cache-policy bit                                         } 0
//   8  pieces spread over the first two MFMA groups                                        } (profiles/r02f_*)
//  16  ablation: no LDS-DMA loads at all (wrong results; MFMA + LDS-read + barrier time only)
//  32  ablation: no MFMAs (wrong results; staging + LDS-read + barrier time only)
//  64  ablation: no fragment reads from LDS inside the loop (wrong results)
// 256  A/B reference for the staggered schedule of the 8-wave shapes: interleaved DMA issue, all waves alike
template <class C, int EXP = 0>
__device__ __forceinline__ void gemm_mainloop(f32x16 (&acc)[2][C::NJ], const Terms& t, const Geom& g, int m0, int n0,
                                              char* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging (layout: prl_lmhead_layout.h): this thread fetches chunks q * NT + tid of each tile,
  // i.e. rows stage_row(tid, q), all from source column stage_kcol(tid)
  const int kcol = stage_kcol(tid);
  int64_t offA[C::QA], offB[C::QB];
#pragma unroll
  for (int q = 0; q < C::QA; ++q) {
    int ra = m0 + stage_row(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;  // rows past the edge re-read the last row; their results are discarded
    offA[q] = (int64_t)ra * g.lda + kcol;
  }
#pragma unroll
  for (int q = 0; q < C::QB; ++q) {
    int rb = n0 + stage_row(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offB[q] = (int64_t)rb * g.ldb + kcol;
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
    constexpr int AUX = (EXP & 4) ? 1 : 0;
    if constexpr ((EXP & 16) != 0) return;
    if (idx < C::QA) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sA + offA[idx] + st_k),
                                       (__attribute__((address_space(3))) void*)(lds + dst + idx * C::NT * 16), 16, 0, AUX);
    } else {
      const int q = idx - C::QA;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sB + offB[q] + st_k),
                                       (__attribute__((address_space(3))) void*)(lds + dst + C::A_BYTES + q * C::NT * 16), 16, 0, AUX);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int idx = 0; idx < C::LOADS; ++idx) stage_piece(buf, idx);
  };
  // MFMAs of one staged tile.  With `sbuf >= 0` the LDS-DMA pieces of the NEXT tile to stage are issued in
  // between the MFMA groups instead of in one burst after the barrier: a piece costs ~60-180 issue cycles,
  // and all waves of the workgroup leave the barrier together - a burst would idle the matrix pipe of
  // every SIMD at once (measured on the same build: 17.7 ms interleaved vs 19.8 ms burst for the 7B forward).
  // STAGGER (8-wave shapes): the two waves that share a SIMD (w and w + 4) do their non-matrix work at opposite
  // ends of the step - waves 4-7 issue the next tile's DMA right after the barrier and then run their MFMA cluster,
  // waves 0-3 run the cluster first and issue afterwards - so that between two barriers one wave of every SIMD is in
  // its matrix cluster while its partner issues loads (dual-plane forward: 14.7 -> 13.8 ms).  4-wave shapes (one wave
  // per SIMD and workgroup, two workgroups per CU) keep the interleaved issue.
  // Measured on this (generic) core it does NOT pay: 256 x 128 forward 16.3 -> 16.7 ms, backward 61.0 -> 62.1 ms -
  // with 64-deep stages the DMA burst of 8 pieces is long enough to delay the partner; kept behind EXP bit 512.
  constexpr bool STAGGER = C::NT == 512 && (EXP & 512) != 0;
  const bool dma_first = STAGGER && wave >= 4;
  // HAND (EXP bit 1024, round 4): the hand-placed stream of gemm_mainloop_dual's MODE 2 on this core - order-pinning asm MFMAs,
  // one LDS-DMA piece after every (MFMAS / LOADS)-th MFMA, the fragment reads of sub-step ks + 1 two per gap after the first
  // MFMAs of sub-step ks
  constexpr bool HAND = (EXP & 1024) != 0;
  auto compute = [&](int buf, int sbuf) {
    const char* base = lds + buf * C::STAGE_BYTES;
    if constexpr (HAND) {
      constexpr int NJ = C::NJ, PER_KS = 2 * NJ, MFMAS = 4 * PER_KS, SPACE = MFMAS / C::LOADS;
      static_assert(SPACE >= 1 && (MFMAS - SPACE / 2 - 1) / SPACE + 1 >= C::LOADS, "not every DMA piece has an MFMA slot");
      bf16x8 af[2][2], bfr[2][NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(base + rdA[0] + i * 32 * ROW_BYTES);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES);
      int m = 0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            mfma_bf16_asm<false, true>(acc[i][j], af[cur][i], bfr[cur][j]);
            const int t = i * NJ + j;  // position in this sub-step: the 2 + NJ reads of the next one go two per gap from t = 1 on
            if (ks < 3 && t >= 1 && 2 * (t - 1) < 2 + NJ) {
#pragma unroll
              for (int r = 2 * (t - 1); r < 2 * t; ++r) {
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
      return;
    }
    constexpr int GROUPS = (EXP & 1) ? 1 : (EXP & 8) ? 2 : 3;
    constexpr int PER = (C::LOADS + GROUPS - 1) / GROUPS;  // pieces after each of the first GROUPS MFMA groups
    if (STAGGER && sbuf >= 0 && dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
    bf16x8 af[2][2], bfr[2][C::NJ];  // [ping-pong][tile]: the reads of sub-step ks + 1 are issued before the MFMAs of ks
#pragma unroll
    for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(base + rdA[0] + i * 32 * ROW_BYTES);
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < 3) {
        if constexpr ((EXP & 64) != 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[nxt][i] = af[cur][i];
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) bfr[nxt][j] = bfr[cur][j];
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[nxt][i] = *reinterpret_cast<const bf16x8*>(base + rdA[ks + 1] + i * 32 * ROW_BYTES);
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) bfr[nxt][j] = *reinterpret_cast<const bf16x8*>(base + rdB[ks + 1] + j * 32 * ROW_BYTES);
        }
      }
      if constexpr ((EXP & 2) != 0) __builtin_amdgcn_s_setprio(1);
      if constexpr ((EXP & 32) != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(af[cur][i]));
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) asm volatile("" ::"v"(bfr[cur][j]));
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
      }
      if constexpr ((EXP & 2) != 0) __builtin_amdgcn_s_setprio(0);
      if (!STAGGER && sbuf >= 0 && ks < GROUPS) {
#pragma unroll
        for (int k = 0; k < PER; ++k)
          if (ks * PER + k < C::LOADS) stage_piece(sbuf, ks * PER + k);
      }
    }
    if (STAGGER && sbuf >= 0 && !dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
  };

  if constexpr ((EXP & 1024) != 0) mfma_pin_acc(acc);
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
  if constexpr ((EXP & 1024) != 0) mfma_settle();
}

// -----------------------------------------------------------------------------------------------
// dual-plane main loop: acc += (A1 + A2)[m0.., :] B[n0.., :]^T over Kc, 32-deep stages (see CfgDual)
// -----------------------------------------------------------------------------------------------
// MODE 0: DMA pieces interleaved with the MFMA groups, all waves alike.
// MODE 1: STAGGERED - the two waves that share a SIMD (w and w + 4) do their non-matrix work at opposite ends of
//         the step: waves 4-7 issue the next tile's DMA right after the barrier, then run their MFMA cluster; waves
//         0-3 run their MFMA cluster first and issue the DMA afterwards.  Between two barriers one wave of every
//         SIMD is in its matrix cluster while its partner issues loads - the role split of the 8-phase GEMM schedule.
//
// Measured and NOT kept: chaining the forward's vocabulary tiles (the last two steps of a tile stage the first two
// stages of the next, so the pipeline never refills).  17.2 ms against 14.6 for the same loop with the refill: the
// refill is what keeps the 32 workgroups of an XCD in step, and in step they share every weight tile (8 workgroups)
// and hidden tile (4 workgroups) in the XCD's L2.  Chained, they drift apart and L2 misses go from 137 M to 669 M
// requests per forward (HBM fetch 8.6 -> 41 GB; profiles/r02p_lmhead_fwd_chained_vs_refill_pmc.txt).
//
// Also measured and NOT kept: the barrier in the MIDDLE of the step (between the MFMAs of its two 16-deep halves,
// synchronising for tile s + 1), fragment reads always half a step ahead of their MFMAs - across the step boundary
// too - with inline-asm ds_read_b128 and counted lgkmcnt(8) waits (the compiler's own bookkeeping waits for ALL reads,
// and its scheduler sinks the reads behind the MFMAs to share one register set).  The instruction stream came out as
// intended (reads x 8, lgkmcnt(8), MFMA x 16, barrier, reads x 8, MFMA x 16; 248 VGPRs, no scratch) and was 3 % SLOWER:
// forward 14.5 vs 14.0 ms, backward 60.9 vs 59.5 ms (profiles/r02w_lmhead_midstep_barrier_ab.txt).  The exposed LDS
// latency after the barrier is not what the schedule is waiting for.
template <int MODE = 2>
__device__ __forceinline__ void gemm_mainloop_dual(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                   const Geom& g, int m0, int n0, char* lds) {
  using C = CfgDual;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  int64_t offA[C::Q], offB[C::Q];
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int ra = m0 + stage_row32(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offA[q] = (int64_t)ra * g.lda + kcol;
    offB[q] = (int64_t)rb * g.ldb + kcol;
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
    const uint16_t* src = tile == 0 ? A1 + offA[q] : tile == 1 ? A2 + offA[q] : B + offB[q];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + st_k),
                                     (__attribute__((address_space(3))) void*)(lds + dst + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
  };
  const bool dma_first = MODE == 1 && wave >= 4;
  // MODE 2 (round 4, the default): a hand-placed stream, every wave alike - order-pinning asm MFMAs, ONE LDS-DMA piece after
  // every 5th MFMA (m = 2, 7, .. 27), the second half's fragment reads two per gap after MFMAs 3-6.  A piece costs its wave
  // ~60 clocks of issue among bare MFMAs and 100-185 inside a burst (MI355X_MICROARCH.md): the staggered bursts of MODE 1
  // hide that behind the partner wave's matrix cluster, single pieces between MFMAs mostly do not incur it.  7B forward, same
  // box, interleaved, bit-identical outputs: 13.77 (MODE 1) -> 13.16 -> 12.97 ms with the spread reads (second box 13.82 -> 13.30;
  // profiles/r04i_*, r04j_*).  NEXT (a type tag): whether a stage is issued.
  auto compute = [&](int buf, int sbuf, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    const char* base = lds + buf * C::STAGE_BYTES;
    if constexpr (MODE == 2) {
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
              mfma_bf16_asm<false, true>(acc[i][j], pl == 0 ? a1[ks][i] : a2[ks][i], bfr[ks][j]);
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
      return;
    }
    if (MODE == 1 && sbuf >= 0 && dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
    bf16x8 a1[2][2], a2[2][2], bfr[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a1[0][i] = *reinterpret_cast<const bf16x8*>(base + rdA[0] + i * 32 * ROW_BYTES32);
      a2[0][i] = *reinterpret_cast<const bf16x8*>(base + C::TILE_BYTES + rdA[0] + i * 32 * ROW_BYTES32);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a1[1][i] = *reinterpret_cast<const bf16x8*>(base + rdA[1] + i * 32 * ROW_BYTES32);
          a2[1][i] = *reinterpret_cast<const bf16x8*>(base + C::TILE_BYTES + rdA[1] + i * 32 * ROW_BYTES32);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[1][j] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + j * 32 * ROW_BYTES32);
      }
      // plane by plane: the two MFMAs that accumulate into the same tile are 8 instructions apart, not
      // back to back (a dependent MFMA waits for its predecessor's result)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
      if (MODE == 0 && sbuf >= 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) stage_piece(sbuf, ks * 3 + k);
      }
    }
    if (MODE == 1 && sbuf >= 0 && !dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
  };

  if constexpr (MODE == 2) mfma_pin_acc(acc);
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
  if constexpr (MODE == 2) mfma_settle();
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
// The stream is the hand-placed one of MODE 2 (order-pinning asm MFMAs, one piece per five MFMAs, reads two per gap).
// Needs at least 6 stages of contraction (shorter ones take gemm_mainloop_dual).  Measured, same box, interleaved, bit-identical
// outputs (profiles/r04p_*): 7B forward 13.87 (round-3 schedule) -> 13.25 (MODE 2) -> 12.87 ms; 32B 19.85 -> 19.02 -> 18.06 ms.
// (Round 2 tried the mid-step barrier on the compiler-scheduled stream and lost 3 %; with every read and DMA issue pinned
// between specific MFMAs it is the other way round.)
template <int HAND>
__device__ __forceinline__ void gemm_mainloop_dual_tr(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                      const Geom& g, int m0, int n0, char* lds);

// TR = true: the TRANSPOSED-A form of gemm_mainloop_dual_tr (d W: A1 / A2 are [K, lda] row-major with the contraction index as
// their ROW; tiles staged as they lie in memory, fragments out of ds_read_b64_tr_b16 - two per fragment, so the second half's
// reads take the gaps after MFMAs 3-6, 8, 9 instead of 3-6) on the same phase-shifted step.
template <bool P1, int BAR, bool RD, bool P2>
struct PsFlags {
  static constexpr bool p1 = P1, rd = RD, p2 = P2;
  static constexpr int bar = BAR;  // -1: no barrier, else the vmcnt of A(s)
};

template <bool TR = false>
__device__ __forceinline__ void gemm_mainloop_dual_ps(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                      const Geom& g, int m0, int n0, char* lds) {
  using C = CfgDual;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const int total = g.Kc / BK32;
  if (total < 6) {
    if constexpr (TR) {
      gemm_mainloop_dual_tr<0>(acc, A1, A2, B, g, m0, n0, lds);
    } else {
      gemm_mainloop_dual<2>(acc, A1, A2, B, g, m0, n0, lds);
    }
    return;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  int64_t offA[C::Q], offB[C::Q];
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    if constexpr (TR) {
      int v = m0 + 8 * tr_stage_chunk(tid, q, C::NT);
      v = v + 8 <= g.M ? v : g.M - 8;  // entries past the edge re-read the last eight; their results are discarded (M % 8 == 0)
      offA[q] = (int64_t)tr_stage_row(tid, q, C::NT) * g.lda + v;
    } else {
      int ra = m0 + stage_row32(tid, q, C::NT);
      ra = ra < g.M ? ra : g.M - 1;
      offA[q] = (int64_t)ra * g.lda + kcol;
    }
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offB[q] = (int64_t)rb * g.ldb + kcol;
  }
  int rdA[2], rdB[2];  // TR: rdA[tile i] (sub-step ks: + 16 token rows); else rdA[ks] (tile i: + 32 rows)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    rdA[ks] = TR ? tr_frag_lds_byte(lane, wm * 64, ks, 0, 0) : frag_lds_byte32(lane, wm * 64, 0, ks);
    rdB[ks] = 2 * C::TILE_BYTES + frag_lds_byte32(lane, wn * C::WCOLS, 0, ks);
  }
  // piece idx (0..5: A1 q0 q1, A2 q0 q1, B q0 q1) of stage `stage` into ring buffer stage % 3
  auto stage_piece = [&](int buf, int stage_k, int idx) {
    const unsigned dst = buf * C::STAGE_BYTES + stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    const int tile = idx / C::Q, q = idx % C::Q;
    const int64_t a_step = TR ? (int64_t)stage_k * g.lda : (int64_t)stage_k;
    const uint16_t* src = tile == 0 ? A1 + offA[q] + a_step : tile == 1 ? A2 + offA[q] + a_step : B + offB[q] + stage_k;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + dst + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
  };
  bf16x8 a1[2][2], a2[2][2], bfr[2][4];  // [half][tile]
  auto rd = [&](const char* base, int off) { return *reinterpret_cast<const bf16x8*>(base + off); };
  // fragment of plane `pl` (0: A1, 1: A2), row tile i, sub-step ks of the stage at `base`
  auto ldA = [&](const char* base, int pl, int i, int ks) -> bf16x8 {
    if constexpr (TR) {  // 8 tokens of one entry: two transposing reads (tokens 0-3, tokens 4-7: + 4 rows = 2048 bytes)
      const char* p0 = base + pl * C::TILE_BYTES + rdA[i] + ks * (16 * 512);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * 512));
      return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else {
      return rd(base, pl * C::TILE_BYTES + rdA[ks] + i * 32 * ROW_BYTES32);
    }
  };
  auto ldB = [&](const char* base, int j, int ks) { return rd(base, rdB[ks] + j * 32 * ROW_BYTES32); };
  // the twelve fragments of half `ks` of the stage at `base`, two LDS reads per MFMA gap, gap 0 = after the half's 4th MFMA
  auto read_gap = [&](const char* base, int ks, int gap) {
    if constexpr (TR) {
      if (gap == 0) {
        a1[ks][0] = ldA(base, 0, 0, ks);
      } else if (gap == 1) {
        bfr[ks][0] = ldB(base, 0, ks);
        bfr[ks][1] = ldB(base, 1, ks);
      } else if (gap == 2) {
        bfr[ks][2] = ldB(base, 2, ks);
        bfr[ks][3] = ldB(base, 3, ks);
      } else if (gap == 3) {
        a1[ks][1] = ldA(base, 0, 1, ks);
      } else if (gap == 5) {
        a2[ks][0] = ldA(base, 1, 0, ks);
      } else if (gap == 6) {
        a2[ks][1] = ldA(base, 1, 1, ks);
      }
    } else {
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
            mfma_bf16_asm<false, true>(acc[i][j], pl == 0 ? a1[h][i] : a2[h][i], bfr[h][j]);
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
// ONE WAVE PER SIMD: 256 x BN tile (BN = 384 | 320) on four waves with up to 512 registers each
// -----------------------------------------------------------------------------------------------
// The 8-wave shapes above hold 128 accumulator registers per lane, which caps the workgroup tile at 256 x 256 - and at
// that tile the dual-plane forward STAGES more than it can hide: 48 KB per 32-deep step at the ~21-24 B/clk/CU the LDS DMA
// path sustains is ~2300 clocks against 2048 clocks of MFMA per SIMD (profiles/r02c, r03ai: matrix pipes 66 % busy, waves
// issue-stalled 50-65 %).  Staged bytes per flop only fall with a larger tile, i.e. more accumulators per lane than two
// waves per SIMD can have.  Here a wave is ALONE on its SIMD (4 waves, __launch_bounds__(256): 512 registers, the
// accumulators spill over into the AGPR half of the file) and owns 128 vocabulary rows x BN / 2 tokens:
//   BN = 384: 4 x 6 tiles = 384 accumulator registers, 56 KB per step (2 x 16 KB planes + 24 KB of hidden states) for
//             96 MFMAs per wave = 3072 clocks: 220 flop / staged byte (256 x 256: 175), staging ~0.87 of the MFMA time;
//             2 stages (112 KB)
//   BN = 320: 4 x 5 tiles = 320 registers, 52 KB per step for 80 MFMAs = 2560 clocks (201 flop/B), 3 stages (156 KB)
// Fragment reads per MFMA fall from 0.5 to 0.29 (8 + 6 reads feed 48 MFMAs).  No partner wave hides this wave's LDS
// latency, so the reads of the NEXT plane / sub-step are issued before the MFMA group that precedes their use.
template <int BN_>
struct CfgOne {
  static constexpr int BM = 256, BN = BN_, NT = 256;
  static constexpr int MI = 4, NJ = BN_ / 64, WROWS = 128, WCOLS = BN_ / 2;
  static constexpr int QA = 4, QB = BN_ / 64;           // 16-byte chunks per thread: a 256-row plane, the BN-row hidden tile
  static constexpr int LOADS = 2 * QA + QB;
  static constexpr int A_BYTES = 256 * ROW_BYTES32, B_BYTES = BN_ * ROW_BYTES32;
  static constexpr int STAGE_BYTES = 2 * A_BYTES + B_BYTES;
  static constexpr int STAGES = (3 * STAGE_BYTES <= 160 * 1024) ? 3 : 2;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
};

// Where an accumulator tile lives is decided HERE, not by the register allocator: handed 320-384 accumulator registers
// and `__builtin_amdgcn_mfma_*`, hipcc selects the VGPR form of the instruction for all of them and uses the AGPR half of
// the file as spill space - 18 v_accvgpr_read / _write per MFMA in the main loop and 850 dwords of scratch (measured on
// the first version of this kernel).  Tiles t < 16 are bound to AGPRs ("+a", 256 registers), the rest to VGPRs ("+v").
// The MFMAs are volatile asm, so the compiler's hazard recogniser does not see them: `mfma_settle()` supplies the wait
// states an MFMA result needs before anything but another MFMA touches it.

// acc[i][j] += (A1 + A2)[m0 + 128 wm + 32 i .., :] B[n0 + WCOLS wn + 32 j .., :]^T over Kc.  HAS_LO = false: one plane (A2 unused).
// MODE 0: the next stage's DMA pieces ride between the MFMA GROUPS (bursts of 3-4 per wave), the compiler places the reads.
// MODE 2: a hand-placed stream - every MFMA is an order-pinning asm, ONE DMA piece after every 6th-7th MFMA, the next group's
// fragment reads after the 4th MFMA of the current one.  Measured (7B forward, same box, interleaved;
// profiles/r04d_lmhead_fwd_one_wave_per_simd_ab.jsonl): MODE 0 14.8 ms, all pieces in one burst after the barrier 15.95 ms,
// MODE 2 13.46 ms against 13.85 ms for the shipped 8-wave dual-plane core (13.77 vs 13.96 on a second box).
template <int BN_, bool HAS_LO, int MODE = 0>
__device__ __forceinline__ void gemm_mainloop_one(f32x16 (&acc)[4][BN_ / 64], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                  const Geom& g, int m0, int n0, char* lds) {
  using C = CfgOne<BN_>;
  constexpr int NJ = C::NJ;
  constexpr int PLANES = HAS_LO ? 2 : 1;
  constexpr int LOADS = PLANES * C::QA + C::QB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  int64_t offA[C::QA], offB[C::QB];
#pragma unroll
  for (int q = 0; q < C::QA; ++q) {
    int ra = m0 + stage_row32(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;
    offA[q] = (int64_t)ra * g.lda + kcol;
  }
#pragma unroll
  for (int q = 0; q < C::QB; ++q) {
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offB[q] = (int64_t)rb * g.ldb + kcol;
  }
  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    rdA[ks] = frag_lds_byte32(lane, wm * C::WROWS, 0, ks);
    rdB[ks] = 2 * C::A_BYTES + frag_lds_byte32(lane, wn * C::WCOLS, 0, ks);
  }
  const int total = g.Kc / BK32;
  int st_k = 0;
  // piece idx: A1 q0..3, [A2 q0..3,] B q0..QB-1
  auto stage_piece = [&](int buf, int idx) {
    const unsigned dst = buf * C::STAGE_BYTES + stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    const uint16_t* src;
    unsigned off;
    if (idx < C::QA) {
      src = A1 + offA[idx];
      off = idx * C::NT * 16;
    } else if (HAS_LO && idx < 2 * C::QA) {
      src = A2 + offA[idx - C::QA];
      off = C::A_BYTES + (idx - C::QA) * C::NT * 16;
    } else {
      const int q = idx - PLANES * C::QA;
      src = B + offB[q];
      off = 2 * C::A_BYTES + q * C::NT * 16;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + st_k),
                                     (__attribute__((address_space(3))) void*)(lds + dst + off), 16, 0, 0);
  };
  // the DMA pieces of the next stage ride between the MFMA groups of this one: GROUPS = 2 * PLANES groups per step
  constexpr int GROUPS = 2 * PLANES;
  constexpr int PER = (LOADS + GROUPS - 1) / GROUPS;
  // MODE 2: ONE piece after every SPACE-th MFMA.  A wave that is alone on its SIMD pays for every cycle a VMEM issue waits for
  // room in the LDS-DMA path with an idle matrix pipe (no partner wave issues MFMAs meanwhile): bursts of 3-4 pieces per wave
  // (16 KB per CU at ~21 B/clk = ~780 clocks) back the queue up, single pieces 6-7 MFMAs (~200 clocks) apart do not.
  constexpr int MFMAS = GROUPS * 4 * NJ;
  // (every 5th / 4th MFMA instead - the last piece issued earlier, more of the step left for it to land before the step-end
  // vmcnt(0) of the two-stage ring - measured the same / 0.8 % slower: the landing time is not what the step waits for)
  constexpr int SPACE = MFMAS / LOADS;
  static_assert(MODE < 2 || (MFMAS - SPACE / 2 - 1) / SPACE + 1 >= LOADS, "not every DMA piece has an MFMA slot");
  // NEXT (a type tag): whether a stage is issued during this step - compile-time, so the step is one straight-line block
  auto compute = [&](int buf, int sbuf, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    constexpr bool PIN = MODE >= 2;
    const char* base = lds + buf * C::STAGE_BYTES;
    bf16x8 bfr[2][NJ], af[2][4];  // bfr[ks]; af ping-pongs between consecutive MFMA groups
    if constexpr (MODE != 0) {  // in the order the first MFMAs consume them
      af[0][0] = *reinterpret_cast<const bf16x8*>(base + rdA[0]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES32);
#pragma unroll
      for (int i = 1; i < 4; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(base + rdA[0] + i * 32 * ROW_BYTES32);
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES32);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(base + rdA[0] + i * 32 * ROW_BYTES32);
    }
    int grp = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) {
        const int cur = grp & 1, nxt = cur ^ 1;
        // the reads the NEXT group needs: before this group's MFMAs (MODE 0 / 1), or after its first four (MODE 2: the pinned
        // stream would otherwise start every group with a burst of reads in front of an idle matrix pipe)
        auto prefetch = [&]() {
          if (pl + 1 < PLANES) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[nxt][i] = *reinterpret_cast<const bf16x8*>(base + C::A_BYTES + rdA[ks] + i * 32 * ROW_BYTES32);
          } else if (ks == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[1][j] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + j * 32 * ROW_BYTES32);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[nxt][i] = *reinterpret_cast<const bf16x8*>(base + rdA[1] + i * 32 * ROW_BYTES32);
          }
        };
        if constexpr (MODE < 2) prefetch();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (i * NJ + j < 16) {
              mfma_bf16_asm<true, PIN>(acc[i][j], af[cur][i], bfr[ks][j]);
            } else {
              mfma_bf16_asm<false, PIN>(acc[i][j], af[cur][i], bfr[ks][j]);
            }
            if constexpr (MODE >= 2) {
              const int m = grp * 4 * NJ + i * NJ + j;  // index of this MFMA in the step
              if (i * NJ + j == 3) prefetch();
              if (NEXT && m % SPACE == SPACE / 2 && m / SPACE < LOADS) stage_piece(sbuf, m / SPACE);
            }
          }
        if (MODE == 0 && NEXT) {
#pragma unroll
          for (int k = 0; k < PER; ++k)
            if (grp * PER + k < LOADS) stage_piece(sbuf, grp * PER + k);
        }
        ++grp;
      }
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;

  mfma_pin_acc<16>(acc);
  constexpr int D = C::STAGES - 1;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < D; ++p)
    if (p < total) {
#pragma unroll
      for (int idx = 0; idx < LOADS; ++idx) stage_piece(p, idx);
      st_k += BK32;
    }
  int cur = 0, nxt = D % C::STAGES;
  int s = 0;
  for (; s + D < total; ++s) {
    wait_tile_then_barrier<(D - 1) * LOADS>();
    compute(cur, nxt, Yes{});
    st_k += BK32;
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
    nxt = nxt + 1 == C::STAGES ? 0 : nxt + 1;
  }
  for (; s < total; ++s) {
    if (s + D - 1 < total) {
      wait_tile_then_barrier<(D - 1) * LOADS>();
    } else {
      wait_tile_then_barrier<0>();
    }
    compute(cur, -1, No{});
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
  }
  mfma_settle();
}

// -----------------------------------------------------------------------------------------------
// dual-plane main loop with a TRANSPOSED A operand (d W from the row-major d-logits planes)
// -----------------------------------------------------------------------------------------------
// acc[v][n] += sum over t of (A1 + A2)[t][m0 + v] B[n0 + n][t]: A1 / A2 are [K, lda] row-major with the contraction index
// as their ROW (d logits hi / lo, [tokens, vocabulary]), B is contraction-contiguous as everywhere else.  The A tiles are
// staged as they lie in memory (512-byte row segments) and the fragments come out of ds_read_b64_tr_b16 (layout header).
// Same stage geometry, ring and wave roles as the dual-plane core.
// HAND 0 (the default HERE): the round-3 schedule - staggered wave roles, the six pieces in one burst.  HAND 1: the hand-placed
// stream of gemm_mainloop_dual's MODE 2 (order-pinning asm MFMAs, one LDS-DMA piece after every 5th MFMA, the second half's
// fragment reads spread over the gaps after MFMAs 3-9) - measured SLOWER on this kernel: d W 19.1 ms against 16.7
// (profiles/r04k_*).  Its A planes stream from HBM (5 GB per 8192 rows, L2 hit 80 %): the early burst of the staggered
// schedule puts the whole stage in flight a step and a half before it is needed, the evenly spread pieces do not (a piece
// every 4th MFMA from the 2nd on: 19.3 ms, the same).  Selectable with PRL_TUNE_LMHEAD_BWD bit 3 for A/B.
// The phase-shifted step (gemm_mainloop_dual_ps<true>, bit 4) - the forward's best schedule, prefetch two and a half steps deep -
// is slower still: 21.1 ms against 16.6 and 18.9, bit-identical d W (profiles/r04t_dw_phase_shift_ab.jsonl).  What this kernel
// rewards is the BURST: the 512-byte row segments of a stage (rows 304 KB apart in the planes) requested together.
// Three more forms were measured and removed again (profiles/r04v_*, same box each): the hand-placed reads WITH the burst
// (16.80-16.92 ms against 16.81-16.94: the MFMA / read placement is not what bounds this kernel), every wave issuing its burst
// right after the barrier (18.7-19.1) and every wave after its MFMAs (17.1-17.2) against the staggered roles (16.8).
// Nor is it DRAM page locality: a timing-only run with the planes ADDRESSED as contiguous [32 tokens x 256 entries] blocks (one
// 16 KB block per staged tile instead of 32 row segments 304 KB apart) took 16.67-16.74 ms against 16.73-16.90
// (profiles/r04w_*) - a block-tiled plane layout is not worth building.
// And it is not the LDS DMA as such: the same stage moved global -> VGPR -> LDS (global_load_dwordx4 at the top of the step into
// 24 registers, ds_write_b128 into the DMA's slots at its end, published by the next barrier; 204 VGPRs, no scratch) took
// 18.3-18.5 ms against 16.4-16.6, bit-identical (profiles/r04vs_*); removed again.
template <int HAND>
__device__ __forceinline__ void gemm_mainloop_dual_tr(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B,
                                                      const Geom& g, int m0, int n0, char* lds) {
  using C = CfgDual;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  int64_t offA[C::Q], offB[C::Q];
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int v = m0 + 8 * tr_stage_chunk(tid, q, C::NT);
    v = v + 8 <= g.M ? v : g.M - 8;  // entries past the edge re-read the last eight; their results are discarded (M % 8 == 0)
    offA[q] = (int64_t)tr_stage_row(tid, q, C::NT) * g.lda + v;
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offB[q] = (int64_t)rb * g.ldb + kcol;
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
    const uint16_t* src = tile == 0 ? A1 + offA[q] + (int64_t)st_k * g.lda : tile == 1 ? A2 + offA[q] + (int64_t)st_k * g.lda : B + offB[q] + st_k;
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
    if constexpr (HAND != 0) {
      bf16x8 a1[2][2], a2[2][2], bfr[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a1[0][i] = a_frag(base, i, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(base + rdB[0] + j * 32 * ROW_BYTES32);
#pragma unroll
      for (int i = 0; i < 2; ++i) a2[0][i] = a_frag(base + C::TILE_BYTES, i, 0);
      int m = 0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              mfma_bf16_asm<false, true>(acc[i][j], pl == 0 ? a1[ks][i] : a2[ks][i], bfr[ks][j]);
              if (m == 3) {
                a1[1][0] = a_frag(base, 0, 1);
              } else if (m == 4) {
                bfr[1][0] = *reinterpret_cast<const bf16x8*>(base + rdB[1]);
                bfr[1][1] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + 32 * ROW_BYTES32);
              } else if (m == 5) {
                bfr[1][2] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + 2 * 32 * ROW_BYTES32);
                bfr[1][3] = *reinterpret_cast<const bf16x8*>(base + rdB[1] + 3 * 32 * ROW_BYTES32);
              } else if (m == 6) {
                a1[1][1] = a_frag(base, 1, 1);
              } else if (m == 8) {
                a2[1][0] = a_frag(base + C::TILE_BYTES, 0, 1);
              } else if (m == 9) {
                a2[1][1] = a_frag(base + C::TILE_BYTES, 1, 1);
              }
              if (sbuf >= 0 && m % 5 == 2 && m / 5 < C::LOADS) stage_piece(sbuf, m / 5);
              ++m;
            }
        }
      }
      return;
    }
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
  if constexpr (HAND != 0) mfma_pin_acc(acc);
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
  if constexpr (HAND != 0) mfma_settle();
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
// Waves 4-7 issue the DMA right after the barrier, waves 0-3 after their MFMA cluster (the staggered roles of the
// dual-plane core).
struct CfgTriple {
  static constexpr int BM = 256, BN = 256, NT = 512, STAGES = 2;
  static constexpr int WCOLS = 128;
  static constexpr int Q = 2;                       // 16-byte chunks per thread, tile and stage
  static constexpr int LOADS = 4 * Q;
  static constexpr int TILE_BYTES = 256 * ROW_BYTES32;  // 16 KB
  static constexpr int STAGE_BYTES = 4 * TILE_BYTES;    // 64 KB
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;  // 128 KB
};

// HAND (round 4, the default): hand-placed stream - order-pinning asm MFMAs, one LDS-DMA piece after every 6th MFMA (eight
// pieces, 48 MFMAs), the second half's twelve fragment reads two per gap after MFMAs 3-8; false: the round-3 schedule.
template <bool HAND = true>
__device__ __forceinline__ void gemm_mainloop_triple(f32x16 (&acc)[2][4], const uint16_t* A1, const uint16_t* A2, const uint16_t* B1,
                                                     const uint16_t* B2, const Geom& g, int m0, int n0, char* lds) {
  using C = CfgTriple;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  int64_t offA[C::Q], offB[C::Q];
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int ra = m0 + stage_row32(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offA[q] = (int64_t)ra * g.lda + kcol;
    offB[q] = (int64_t)rb * g.ldb + kcol;
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
    const uint16_t* src = tile == 0 ? A1 + offA[q] : tile == 1 ? A2 + offA[q] : tile == 2 ? B1 + offB[q] : B2 + offB[q];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + st_k),
                                     (__attribute__((address_space(3))) void*)(lds + dst + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
  };
  const bool dma_first = wave >= 4;
  auto compute = [&](int buf, int sbuf) {
    const char* base = lds + buf * C::STAGE_BYTES;
    if constexpr (HAND) {
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
                mfma_bf16_asm<false, true>(acc[i][j], pr == 1 ? a2[i] : a1[i], pr == 2 ? b2[j] : b1[j]);
              } else {
                mfma_bf16_asm<false, true>(acc[i][j], pr == 1 ? a2n[i] : a1n[i], pr == 2 ? b2n[j] : b1n[j]);
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
      return;
    }
    if (sbuf >= 0 && dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a1[2], a2[2], b1[4], b2[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a1[i] = *reinterpret_cast<const bf16x8*>(base + rdA[ks] + i * 32 * ROW_BYTES32);
        a2[i] = *reinterpret_cast<const bf16x8*>(base + C::TILE_BYTES + rdA[ks] + i * 32 * ROW_BYTES32);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        b1[j] = *reinterpret_cast<const bf16x8*>(base + rdB[ks] + j * 32 * ROW_BYTES32);
        b2[j] = *reinterpret_cast<const bf16x8*>(base + C::TILE_BYTES + rdB[ks] + j * 32 * ROW_BYTES32);
      }
      // product by product: the MFMAs that accumulate into the same tile are 8 instructions apart
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b1[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b2[j], acc[i][j], 0, 0, 0);
    }
    if (sbuf >= 0 && !dma_first) {
#pragma unroll
      for (int k = 0; k < C::LOADS; ++k) stage_piece(sbuf, k);
    }
  };
  if constexpr (HAND) mfma_pin_acc(acc);
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
  if constexpr (HAND) mfma_settle();
}

// -----------------------------------------------------------------------------------------------
// mixed-precision main loop: acc += A16 B16^T (f16 MFMA) + 2^s (A8 B8^T) (MX-scaled fp8 MFMA, twice the f16 rate)
// -----------------------------------------------------------------------------------------------
// The fp32 operand X is held as  X S = X16 + X8 2^-4  with X16 = f16(X S) (11 significant bits) and X8 = fp8_e4m3 of the
// rounding residual (|residual| <= 2^-11 |X S|, 4 more bits): 15 bits where the two bf16 planes had 16, at 3/4 of the
// matrix-pipe time.  In  (A16 + a)(B16 + b)  the cross terms are 2^-11 corrections, so they need 4 bits, not 11: the OTHER
// operand enters them rounded to fp8 and the product runs on the MX instruction, whose 32 x 32 accumulator layout is the
// f16 instruction's (both add into ONE accumulator).  Error of the result: the fp8 roundings, 2^-4 relative on a 2^-11
// term - 2e-5 per term at worst, ~1e-5 of the result's scale over a contraction - against the 4e-6 of the two-bf16-plane
// form; the fp32 GEMM both stand for is exact to 1e-7.  (Probe of instruction semantics, rates and the free choice of K
// slots: profiles/r03b_mx_probe.txt.)
//
// This loop is the forward / recompute form: A16 (weight rows) and B16 (hidden states) are staged as f16, A8 (the weight's
// residual plane) and B8 (the hidden states rounded to fp8) as fp8 in slot order (prl_lmhead_layout.h).  32-deep stages;
// the MX instruction spans two of them (64 deep): the 16-bit tiles live in a ring of three (2 x 16 KB per stage), the fp8
// tiles (2 x 8 KB per stage) in a ring of FOUR, because the tiles of the even stage are read at the end of the odd one,
// when the 16-bit buffer of the even stage is already being refilled.
// (Rounding B16 to fp8 in registers instead of staging B8 would save a sixth of the staged bytes, but the rounded
// fragments of a stage pair have to stay in registers - 32 more per lane next to the 128 accumulators: 47 spilled.)
struct CfgMx {
  static constexpr int BM = 256, BN = 256, NT = 512, STAGES = 3, STAGES8 = 4;
  static constexpr int NJ = 4, WCOLS = 128;
  static constexpr int Q = 2;                            // 16-byte chunks per thread of a 16-bit tile
  static constexpr int LOADS = 2 * Q + 2;                // + one chunk of each fp8 tile
  static constexpr int TILE_BYTES = 256 * ROW_BYTES32;   // 16 KB
  static constexpr int TILE8_BYTES = 256 * 32;           // 8 KB
  static constexpr int STAGE_BYTES = 2 * TILE_BYTES;     // 32 KB of 16-bit tiles per stage
  static constexpr int STAGE8_BYTES = 2 * TILE8_BYTES;   // 16 KB of fp8 tiles per stage
  static constexpr int LDS8_BASE = STAGES * STAGE_BYTES;  // 96 KB
  static constexpr int LDS_BYTES = LDS8_BASE + STAGES8 * STAGE8_BYTES;  // 160 KB: all of a CU's LDS
};
constexpr int kE8M0 = 127;                 // E8M0 exponent of scale 1
constexpr float kHi8Div = 128.0f;          // an f16 plane value (<= 2^15) / 128 -> fp8 range (<= 256 < 448)
constexpr int kHi8Exp = kE8M0 + 7;         // ... and the MX scale that undoes it
constexpr float kLo8Mul = 16.0f;           // a residual (<= 16) * 16 -> fp8 range
constexpr int kLo8Exp = kE8M0 - 4;

__device__ __forceinline__ int e8m0x4(int e) { return e * 0x01010101; }

template <bool HAS_LO>
__device__ __forceinline__ void gemm_mainloop_mx(f32x16 (&acc)[2][4], const uint16_t* A16, const uint16_t* B16, const uint8_t* A8,
                                                 const uint8_t* B8, const Geom& g, int m0, int n0, char* lds) {
  using C = CfgMx;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kcol = stage_kcol32(tid);
  int64_t offA[C::Q], offB[C::Q], offA8, offB8;
#pragma unroll
  for (int q = 0; q < C::Q; ++q) {
    int ra = m0 + stage_row32(tid, q, C::NT);
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + stage_row32(tid, q, C::NT);
    rb = rb < g.N ? rb : g.N - 1;
    offA[q] = (int64_t)ra * g.lda + kcol;
    offB[q] = (int64_t)rb * g.ldb + kcol;
  }
  {  // the fp8 planes have one byte per element: the row strides (in elements) are their row strides in bytes
    int ra = m0 + mx8_stage_row(tid);
    ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + mx8_stage_row(tid);
    rb = rb < g.N ? rb : g.N - 1;
    offA8 = (int64_t)ra * g.lda + 16 * mx8_stage_half(tid);
    offB8 = (int64_t)rb * g.ldb + 16 * mx8_stage_half(tid);
  }
  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    rdA[ks] = frag_lds_byte32(lane, wm * 64, 0, ks);
    rdB[ks] = C::TILE_BYTES + frag_lds_byte32(lane, wn * C::WCOLS, 0, ks);
  }
  // tile i / j: + 32 rows * 32 bytes (the swizzle term depends on (row >> 3) & 1 only, which + 32 leaves unchanged)
  const int rdA8 = mx8_frag_lds_byte(lane, wm * 64, 0), rdB8 = C::TILE8_BYTES + mx8_frag_lds_byte(lane, wn * C::WCOLS, 0);
  const int total = g.Kc / BK32;  // even (Kc is a multiple of 64)
  int st_k = 0;
  auto stage_piece = [&](int buf, int buf8, int idx) {  // idx 0..5: A16 q0 q1, B16 q0 q1, A8, B8
    const unsigned lane0 = stage_lds_byte(wave * 64, 0, C::NT);  // + lane * 16 by the hardware
    if (idx < 4) {
      const int tile = idx >> 1, q = idx & 1;
      const uint16_t* src = tile == 0 ? A16 + offA[q] : B16 + offB[q];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + st_k),
                                       (__attribute__((address_space(3))) void*)(lds + buf * C::STAGE_BYTES + lane0 + tile * C::TILE_BYTES + q * C::NT * 16), 16, 0, 0);
    } else if (HAS_LO) {
      const uint8_t* src = idx == 4 ? A8 + offA8 : B8 + offB8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + st_k),
                                       (__attribute__((address_space(3))) void*)(lds + C::LDS8_BASE + buf8 * C::STAGE8_BYTES + lane0 + (idx - 4) * C::TILE8_BYTES), 16, 0, 0);
    }
  };
  constexpr int NLOADS = HAS_LO ? C::LOADS : C::LOADS - 2;
  const bool dma_first = wave >= 4;
  // one 32-deep stage: 16 f16 MFMAs; an ODD stage closes its pair with the 8 MX MFMAs over both stages' fp8 tiles
  auto compute = [&](int buf, int buf8, int sbuf, int sbuf8, bool odd) {
    const char* base = lds + buf * C::STAGE_BYTES;
    if (sbuf >= 0 && dma_first) {
#pragma unroll
      for (int k = 0; k < NLOADS; ++k) stage_piece(sbuf, sbuf8, k);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 af[2], bfr[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(base + rdA[ks] + i * 32 * ROW_BYTES32);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const f16x8*>(base + rdB[ks] + j * 32 * ROW_BYTES32);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (HAS_LO && odd) {
      const char* even8 = lds + C::LDS8_BASE + ((buf8 + C::STAGES8 - 1) & (C::STAGES8 - 1)) * C::STAGE8_BYTES;
      const char* odd8 = lds + C::LDS8_BASE + buf8 * C::STAGE8_BYTES;
      i32x8 qa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(even8 + rdA8 + i * 32 * 32);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(odd8 + rdA8 + i * 32 * 32);
        qa[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(even8 + rdB8 + j * 32 * 32);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(odd8 + rdB8 + j * 32 * 32);
        const i32x8 qb = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[i], qb, acc[i][j], 0, 0, 0, e8m0x4(kLo8Exp), 0, e8m0x4(kHi8Exp));
      }
    }
    if (sbuf >= 0 && !dma_first) {
#pragma unroll
      for (int k = 0; k < NLOADS; ++k) stage_piece(sbuf, sbuf8, k);
    }
  };

  constexpr int D = C::STAGES - 1;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < D; ++p)
    if (p < total) {
#pragma unroll
      for (int idx = 0; idx < NLOADS; ++idx) stage_piece(p, p, idx);
      st_k += BK32;
    }
  int cur = 0, nxt = D % C::STAGES, cur8 = 0, nxt8 = D;
  for (int s = 0; s < total; ++s) {
    const bool more = s + D < total;
    if (more || s + D - 1 < total) {
      wait_tile_then_barrier<(D - 1) * NLOADS>();
    } else {
      wait_tile_then_barrier<0>();
    }
    compute(cur, cur8, more ? nxt : -1, nxt8, (s & 1) != 0);
    st_k += BK32;
    cur = cur + 1 == C::STAGES ? 0 : cur + 1;
    nxt = nxt + 1 == C::STAGES ? 0 : nxt + 1;
    cur8 = (cur8 + 1) & (C::STAGES8 - 1);
    nxt8 = (nxt8 + 1) & (C::STAGES8 - 1);
  }
}

// One call site for both cores: DUAL runs the dual-plane loop on (terms.a[0], terms.a[1], terms.b[0])
template <class C, bool DUAL, int EXP = 0>
__device__ __forceinline__ void run_mainloop(f32x16 (&acc)[2][C::NJ], const Terms& t, const Geom& g, int m0, int n0, char* lds) {
  if constexpr (DUAL) {
    // default: the phase-shifted hand-placed stream; EXP bit 1024: hand-placed with the barrier at the step start (MODE 2),
    // 512: the round-2 / round-3 schedule (MODE 1), 256: MODE 0 - the A/B references
    if constexpr ((EXP & (256 | 512 | 1024)) == 0) {
      gemm_mainloop_dual_ps(acc, t.a[0], t.a[1], t.b[0], g, m0, n0, lds);
    } else {
      gemm_mainloop_dual<(EXP & 256) ? 0 : (EXP & 512) ? 1 : 2>(acc, t.a[0], t.a[1], t.b[0], g, m0, n0, lds);
    }
  } else {
    gemm_mainloop<C, EXP>(acc, t, g, m0, n0, lds);
  }
}

// CORE: 0 generic, 1 dual-plane, 2 mixed precision (f16 + MX fp8: terms.a[0] = A16, terms.a[1] = A8 residual plane,
// terms.b[0] = B16, terms.b[1] = B8 = B16 rounded to fp8), 3 mixed precision without a residual plane (an f16-exact weight)
template <class C, int CORE, int EXP = 0>
__device__ __forceinline__ void run_core(f32x16 (&acc)[2][C::NJ], const Terms& t, const Geom& g, int m0, int n0, char* lds) {
  if constexpr (CORE == 2) {
    gemm_mainloop_mx<true>(acc, t.a[0], t.b[0], reinterpret_cast<const uint8_t*>(t.a[1]), reinterpret_cast<const uint8_t*>(t.b[1]), g, m0, n0, lds);
  } else if constexpr (CORE == 3) {
    gemm_mainloop_mx<false>(acc, t.a[0], t.b[0], nullptr, nullptr, g, m0, n0, lds);
  } else {
    run_mainloop<C, CORE == 1, EXP>(acc, t, g, m0, n0, lds);
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
// forward.  A = weight planes (M = vocabulary), B = hidden (N = logits rows / tokens)
// -----------------------------------------------------------------------------------------------
struct FwdArgs {
  Terms terms;
  Geom geo;             // M = vocab, N = n logits rows, Kc = hidden
  int64_t cols;         // batch columns: logits row q predicts token q + 1 unless q % cols == cols - 1
  const int64_t* ids;   // [n]
  float k2;             // log2(e) / temperature
  int vt, tt, nsplit;   // vocabulary tiles (of BM), token tiles (of BN), vocabulary splits
  int64_t padded;       // tt * BN
  float* part;          // [nsplit][padded][4]  (M, S, W, -)
  float* ysel;          // [padded] selected logit (base-2 units), written by whichever split owns the row
  const float* scales;  // mixed-precision cores: device floats {S_w, S_h} the operands were multiplied by; nullptr otherwise
  float* logits2;       // nullable [n, vocab]: the logits in base-2 units (logit * log2(e) / temperature), kept for the backward
};

template <class C, int EXP = 0, int CORE = 0>
__global__ __launch_bounds__(C::NT, 2) void lmhead_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int tok_tile, split;
  tile_coords(blockIdx.x, a.tt, a.nsplit, tok_tile, split);
  constexpr int BN = C::BN, NJ = C::NJ;
  const int n0 = tok_tile * BN;
  const int vt0 = (int)((int64_t)a.vt * split / a.nsplit), vt1 = (int)((int64_t)a.vt * (split + 1) / a.nsplit);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * 64, wcol0 = (wave & 1) * C::WCOLS;
  const int V = a.geo.M;

  // the mixed-precision operands carry power-of-two scales: exact to undo
  const float k2 = a.scales ? a.k2 / (a.scales[0] * a.scales[1]) : a.k2;
  Osm st[NJ];
  int tgt[NJ];  // target vocabulary row of this lane's tokens, -1: none
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    osm_init(st[j]);
    const int64_t q = n0 + acc_col(lane, wcol0, j);
    int id = -1;
    if (q < a.geo.N && (q % a.cols) != a.cols - 1) {
      const int64_t v = a.ids[q + 1];
      if (v >= 0 && v < V) id = (int)v;
    }
    tgt[j] = id;
  }

  f32x16 acc[2][NJ];
  for (int tv = vt0; tv < vt1; ++tv) {
    const int m0 = tv * C::BM;
    zero_acc<NJ>(acc);
    run_core<C, CORE, EXP>(acc, a.terms, a.geo, m0, n0, lds);
    const int vbase = m0 + acc_row(lane, wrow0, 0, 0);  // vocabulary row of acc[0][j][0]; + 32 i + (reg & 3) + 8 (reg >> 2)
    const bool full = m0 + C::BM <= V;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float y[32];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[i * 16 + r] = acc[i][j][r] * k2;
      if (a.logits2) {
        // kept logits: registers 4 rg .. 4 rg + 3 are four consecutive vocabulary entries of one token row - one 16-byte store;
        // the two half-waves complete a 32-byte aligned piece of the row (V is a multiple of 8).  Plain stores: the pieces of a
        // row meet in L2 before they go out; as non-temporal stores they cost 2.5 ms more per 8192 x 152 064 launch (measured)
        const int64_t q = n0 + acc_col(lane, wcol0, j);
        if (q < a.geo.N) {
          float* dst = a.logits2 + q * (int64_t)V + vbase;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
              if (vbase + i * 32 + 8 * rg + 3 < V)
                *reinterpret_cast<float4*>(dst + i * 32 + 8 * rg) =
                    float4{y[i * 16 + 4 * rg], y[i * 16 + 4 * rg + 1], y[i * 16 + 4 * rg + 2], y[i * 16 + 4 * rg + 3]};
        }
      }
      const int d = tgt[j] - vbase;
      if (d >= 0 && d < 64 && (d & 7) < 4) {  // the target row is one of this lane's 32
        float sel = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (d == i * 32 + (r & 3) + 8 * (r >> 2)) sel = y[i * 16 + r];
        a.ysel[n0 + acc_col(lane, wcol0, j)] = sel;
      }
      if (full) {
        osm_push<32>(st[j], y);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (vbase + i * 32 + (r & 3) + 8 * (r >> 2) < V) {
              float one[1] = {y[i * 16 + r]};
              osm_push<1>(st[j], one);
            }
      }
    }
  }

  // the two half-waves (lane, lane ^ 32) hold the same tokens; then the waves wm = 0 .. WM-1 that share them
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    Osm o;
    o.M = __shfl_xor(st[j].M, 32, 64);
    o.S = __shfl_xor(st[j].S, 32, 64);
    o.W = __shfl_xor(st[j].W, 32, 64);
    st[j] = osm_merge(st[j], o);
  }
  __syncthreads();  // the last tile's LDS reads are done: reuse the buffer for the cross-wave hand-off
  constexpr int WM = C::BM / 64;
  float4* red = reinterpret_cast<float4*>(lds);  // [WM][BN]
  if (lane < 32) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) red[(wave >> 1) * BN + acc_col(lane, wcol0, j)] = float4{st[j].M, st[j].S, st[j].W, 0.0f};
  }
  __syncthreads();
  if (tid < BN) {
    float4 x = red[tid];
    Osm m{x.x, x.y, x.z};
#pragma unroll
    for (int w = 1; w < WM; ++w) {
      x = red[w * BN + tid];
      m = osm_merge(m, Osm{x.x, x.y, x.z});
    }
    reinterpret_cast<float4*>(a.part)[(int64_t)split * a.padded + n0 + tid] = float4{m.M, m.S, m.W, 0.0f};
  }
}

// The forward on the one-wave-per-SIMD core: same arguments, outputs and split / merge scheme as lmhead_fwd_kernel; a wave
// owns 128 vocabulary rows (4 MFMA tiles) x BN / 2 tokens (NJ tiles), a lane one token per tile column and 64 of its rows.
template <int BN_, bool HAS_LO, int MODE = 0>
__global__ __launch_bounds__(256) void lmhead_fwd1_kernel(FwdArgs a) {
  using C = CfgOne<BN_>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int tok_tile, split;
  tile_coords(blockIdx.x, a.tt, a.nsplit, tok_tile, split);
  constexpr int BN = C::BN, NJ = C::NJ;
  const int n0 = tok_tile * BN;
  const int vt0 = (int)((int64_t)a.vt * split / a.nsplit), vt1 = (int)((int64_t)a.vt * (split + 1) / a.nsplit);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * C::WROWS, wcol0 = (wave & 1) * C::WCOLS;
  const int V = a.geo.M;
  const float k2 = a.k2;
  Osm st[NJ];
  int tgt[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    osm_init(st[j]);
    const int64_t q = n0 + acc_col(lane, wcol0, j);
    int id = -1;
    if (q < a.geo.N && (q % a.cols) != a.cols - 1) {
      const int64_t v = a.ids[q + 1];
      if (v >= 0 && v < V) id = (int)v;
    }
    tgt[j] = id;
  }
  f32x16 acc[4][NJ];
  for (int tv = vt0; tv < vt1; ++tv) {
    const int m0 = tv * C::BM;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    gemm_mainloop_one<BN_, HAS_LO, MODE>(acc, a.terms.a[0], a.terms.a[1], a.terms.b[0], a.geo, m0, n0, lds);
    const int vbase = m0 + acc_row(lane, wrow0, 0, 0);  // vocabulary row of acc[0][j][0]; + 32 i + (reg & 3) + 8 (reg >> 2)
    const bool full = m0 + C::BM <= V;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int64_t q = n0 + acc_col(lane, wcol0, j);
      const int d = tgt[j] - vbase;
      const bool mine = d >= 0 && d < 128 && (d & 7) < 4;  // the target row is one of this lane's 64
      float sel = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // one MFMA tile (16 rows of this lane) at a time: 16 live values, not 64
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = acc[i][j][r] * k2;
        if (a.logits2 && q < a.geo.N) {
          float* dst = a.logits2 + q * (int64_t)V + vbase + i * 32;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            if (vbase + i * 32 + 8 * rg + 3 < V)
              *reinterpret_cast<float4*>(dst + 8 * rg) = float4{y[4 * rg], y[4 * rg + 1], y[4 * rg + 2], y[4 * rg + 3]};
        }
        if (mine) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (d == i * 32 + (r & 3) + 8 * (r >> 2)) sel = y[r];
        }
        if (full) {
          osm_push<16>(st[j], y);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (vbase + i * 32 + (r & 3) + 8 * (r >> 2) < V) {
              float one[1] = {y[r]};
              osm_push<1>(st[j], one);
            }
        }
      }
      if (mine) a.ysel[q] = sel;
    }
  }
  // the two half-waves (lane, lane ^ 32) hold the same tokens; then the two waves (wm = 0, 1) that share them
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    Osm o;
    o.M = __shfl_xor(st[j].M, 32, 64);
    o.S = __shfl_xor(st[j].S, 32, 64);
    o.W = __shfl_xor(st[j].W, 32, 64);
    st[j] = osm_merge(st[j], o);
  }
  __syncthreads();
  float4* red = reinterpret_cast<float4*>(lds);  // [2][BN]
  if (lane < 32) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) red[(wave >> 1) * BN + acc_col(lane, wcol0, j)] = float4{st[j].M, st[j].S, st[j].W, 0.0f};
  }
  __syncthreads();
  for (int c = tid; c < BN; c += C::NT) {
    const float4 x = red[c], z = red[BN + c];
    const Osm m = osm_merge(Osm{x.x, x.y, x.z}, Osm{z.x, z.y, z.z});
    reinterpret_cast<float4*>(a.part)[(int64_t)split * a.padded + n0 + c] = float4{m.M, m.S, m.W, 0.0f};
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
  Geom geo;             // M = vocab, N = rows of this chunk, Kc = hidden
  int64_t row_base;     // first logits row of the chunk (global index q)
  int64_t cols;
  const int64_t* ids;
  const float* lse2;    // token-aligned [n_total]
  const float* ent;
  const float* g_nlp;   // token-aligned d loss / d new_logprobs
  const float* g_ent;   // nullable
  const float* upstream;  // nullable device scalar
  float k2, inv_temp;
  int vt, tt, nsplit;   // vocabulary tiles, token tiles of the chunk, vocabulary ranges (workgroups = tt * nsplit)
  int chunk_pad;        // rows of the chunk buffers (multiple of 128)
  uint16_t* dl_hi;      // [chunk_pad, vocab]
  uint16_t* dl_lo;
  const float* scales;  // mixed-precision recompute: device floats {S_w, S_h}; nullptr otherwise
};

template <class C, int CORE = 0>
__global__ __launch_bounds__(C::NT, 2) void lmhead_dlogits_kernel(DlArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // A workgroup owns one tile of token rows and walks a RANGE of vocabulary tiles, like the forward (round 2 gave every
  // (vocabulary tile, token tile) pair its own workgroup: 19 008 dispatches per 8192-row chunk, each with its own pipeline
  // fill and its own loads of the token statistics; as a loop the recompute costs what the forward's main loop costs)
  int tk, split;
  tile_coords(blockIdx.x, a.tt, a.nsplit, tk, split);
  constexpr int NJ = C::NJ;
  const int n0 = tk * C::BN;
  const int vt0 = (int)((int64_t)a.vt * split / a.nsplit), vt1 = (int)((int64_t)a.vt * (split + 1) / a.nsplit);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * 64, wcol0 = (wave & 1) * C::WCOLS;
  const float up = a.upstream ? *a.upstream : 1.0f;
  const float k2 = a.scales ? a.k2 / (a.scales[0] * a.scales[1]) : a.k2;  // power-of-two operand scales: exact to undo
  const int64_t V = a.geo.M;
  f32x16 acc[2][NJ];
  for (int tv = vt0; tv < vt1; ++tv) {
  const int m0 = tv * C::BM;
  zero_acc<NJ>(acc);
  run_core<C, CORE>(acc, a.terms, a.geo, m0, n0, lds);
  // per-token quantities of this lane's NJ token rows - re-read for every vocabulary tile (five cached scalars per row):
  // kept in registers across the main loop they and the loop's own state exceed the register file (98 spilled)
  float t_gi[NJ], t_nhi[NJ], t_l2[NJ], t_H[NJ];
  int t_id[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int lrow = n0 + acc_col(lane, wcol0, j);  // row inside the chunk buffers
    const int64_t q = a.row_base + lrow;
    float g = 0.0f, gH = 0.0f;
    t_l2[j] = 0.0f;
    t_H[j] = 0.0f;
    t_id[j] = -1;
    if (lrow < a.geo.N && (q % a.cols) != a.cols - 1) {  // rows past the chunk's end (padding) come out as zeros
      const int64_t u = q + 1;
      g = a.g_nlp[u] * up;
      gH = a.g_ent ? a.g_ent[u] * up : 0.0f;
      t_l2[j] = a.lse2[u];
      t_H[j] = a.ent[u];
      const int64_t v = a.ids[u];
      if (v >= 0 && v < V) t_id[j] = (int)v;
    }
    t_gi[j] = g * a.inv_temp;
    t_nhi[j] = -gH * a.inv_temp;
  }
  const int vbase = m0 + acc_row(lane, wrow0, 0, 0);
  // ---- d logits -> two bf16 planes (hi + lo = value), ROW-MAJOR [token row][vocabulary], through LDS images of the tile so
  // that every global store is 16 bytes per lane and a wave writes whole 512-byte row segments (storing straight from the
  // accumulator layout - 8-byte pieces at a row stride of V - left partially written sectors behind: 8.8 GB written for
  // 4.98 GB of planes, profiles/r02ai_*).  The tile goes in two HALVES of token rows (this wave's token tiles j < NJ / 2,
  // then the rest): only half of the values are alive as (hi, lo) pairs next to the accumulators - the whole tile at once
  // spilled 22-98 registers - and both planes of a half share the LDS (2 x [BN / 2][BM * 2 + 8] bytes).
  constexpr int BM = C::BM, BN = C::BN, JH = NJ / 2;
  constexpr int RS = BM * 2 + 8;              // image row: + 8 bytes, conflict-free 8-byte writes
  constexpr int IMG = (BN / 2) * RS;          // one plane of one half
  unsigned char* img = reinterpret_cast<unsigned char*>(lds);
  // The image addresses below are invariant across the vocabulary tiles of this workgroup; hoisted out of that loop they
  // stay alive through the main loop (32 + registers: 111 spilled).  An opaque copy of the lane id pins them to the tile.
  int lane_here = lane, tid_here = tid;
  asm volatile("" : "+v"(lane_here), "+v"(tid_here));
  const int lhalf = lane_here >> 5, l31 = lane_here & 31;
  const int wn = (tid_here >> 6) & 1;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();  // every wave is done with the LDS: the main loop's tiles (h = 0) / the previous half's images
    // 16 accumulator values at a time -> (hi, lo) pairs -> straight into the two plane images: no array of pairs is ever alive
    // next to the 128 accumulators (a whole half of pairs first: 111 registers spilled, some of them inside the main loop)
#pragma unroll
    for (int jj = 0; jj < JH; ++jj) {
      const int j = h * JH + jj;
      const float gi = t_gi[j], ngi = -t_gi[j], nhi = t_nhi[j], l2 = t_l2[j], H = t_H[j];
      const int id = t_id[j];
      const bool live = (gi != 0.0f) || (nhi != 0.0f);
      const int local = wn * (JH * 32) + jj * 32 + l31;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          uint32_t p[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = rg * 4 + e;
            float val = 0.0f;
            if (live) {
              const float d2 = __builtin_fmaf(acc[i][j][r], k2, -l2);  // log2 p
              const float pr = fast_exp2(d2);
              val = ngi * pr;
              if (nhi != 0.0f) val = __builtin_fmaf(nhi * pr, __builtin_fmaf(d2, kLn2, H), val);
              if (vbase + i * 32 + (r & 3) + 8 * (r >> 2) == id) val += gi;
            }
            uint16_t hi, lo;
            split2(val, hi, lo);
            p[e] = (uint32_t)hi | ((uint32_t)lo << 16);
          }
          // the lane's 4 consecutive vocabulary entries (registers 4 rg .. 4 rg + 3) of a token row: 8 bytes per plane
          const int vloc = (tid_here >> 7) * 64 + i * 32 + 8 * rg + 4 * lhalf;
          *reinterpret_cast<uint2*>(img + local * RS + vloc * 2) = uint2{(p[0] & 0xffffu) | (p[1] << 16), (p[2] & 0xffffu) | (p[3] << 16)};
          *reinterpret_cast<uint2*>(img + IMG + local * RS + vloc * 2) = uint2{(p[0] >> 16) | (p[1] & 0xffff0000u), (p[2] >> 16) | (p[3] & 0xffff0000u)};
        }
    }
    __syncthreads();
    for (int c = tid_here; c < (BN / 2) * (BM / 8); c += C::NT) {
      const int local = c / (BM / 8), k = c % (BM / 8);
      const int row = (local / (JH * 32)) * C::WCOLS + h * (JH * 32) + local % (JH * 32);  // token row inside the tile
      const int lrow = n0 + row, v = m0 + k * 8;
      if (lrow < a.chunk_pad && v + 7 < V) {  // V and chunk_pad are multiples of 8: a group of eight is inside or outside as a whole
        const unsigned char* src = img + local * RS + k * 16;
        const uint2 x0 = *reinterpret_cast<const uint2*>(src), x1 = *reinterpret_cast<const uint2*>(src + 8);
        const uint2 y0 = *reinterpret_cast<const uint2*>(src + IMG), y1 = *reinterpret_cast<const uint2*>(src + IMG + 8);
        *reinterpret_cast<uint4*>(a.dl_hi + (int64_t)lrow * V + v) = uint4{x0.x, x0.y, x1.x, x1.y};
        *reinterpret_cast<uint4*>(a.dl_lo + (int64_t)lrow * V + v) = uint4{y0.x, y0.y, y1.x, y1.y};
      }
    }
  }
  __syncthreads();  // the images are read: the next vocabulary tile may stage into the LDS
  }  // vocabulary tiles of this workgroup
}

// backward, step 1 when the forward KEPT its logits (FwdArgs.logits2): no recompute - one pass over the chunk's rows of the kept
// fp32 logits (base-2 units) writes the same two d-logits planes.  One workgroup per token row of the chunk buffers (the pad
// rows and the rows without a gradient are written as zeros without reading anything).
struct KeptArgs {
  const float* logits2;  // [n_total, vocab]
  int64_t vocab, row_base, cols;
  int rows;              // rows of this chunk (the buffers have gridDim.x >= rows: the rest is padding)
  const int64_t* ids;
  const float* lse2;
  const float* ent;
  const float* g_nlp;
  const float* g_ent;     // nullable
  const float* upstream;  // nullable device scalar
  float inv_temp;
  uint16_t* dl_hi;        // [gridDim.x, vocab]
  uint16_t* dl_lo;
};

// Every access wave-contiguous: a lane takes FOUR consecutive entries (one 16-byte load, one 8-byte store per plane) and keeps U
// loads in flight.  Measured against the first form (eight entries per lane: two 16-byte loads at a 32-byte lane stride, one
// 16-byte store per plane), same box, rocprofv3 kernel trace, 8192 x 152 064: 2.022 ms -> U = 2: 1.983, 4: 1.958, 8: 1.948 ms
// = 5.1 TB/s of 9.96 GB read + written (profiles/r04u_*); the arithmetic per entry is unchanged, outputs bit-identical.
template <int U = 8>
__global__ __launch_bounds__(256) void dlogits_from_kept_kernel(KeptArgs a) {
  const int lrow = (int)blockIdx.x;
  const int64_t q = a.row_base + lrow;
  const int64_t V = a.vocab;
  float gi = 0.0f, nhi = 0.0f, l2 = 0.0f, H = 0.0f;
  int id = -1;
  if (lrow < a.rows && (q % a.cols) != a.cols - 1) {
    const float up = a.upstream ? *a.upstream : 1.0f;
    const int64_t u = q + 1;
    gi = a.g_nlp[u] * up * a.inv_temp;
    nhi = a.g_ent ? -(a.g_ent[u] * up) * a.inv_temp : 0.0f;
    l2 = a.lse2[u];
    H = a.ent[u];
    const int64_t v = a.ids[u];
    if (v >= 0 && v < V) id = (int)v;
  }
  const bool live = (gi != 0.0f) || (nhi != 0.0f);
  const float ngi = -gi;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.logits2 + q * V);
  uint2* hi = reinterpret_cast<uint2*>(a.dl_hi + (int64_t)lrow * V);
  uint2* lo = reinterpret_cast<uint2*>(a.dl_lo + (int64_t)lrow * V);
  const int quads = (int)(V / 4);
  for (int g0 = threadIdx.x; g0 < quads; g0 += 256 * U) {
    f32x4 x[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int g = g0 + k * 256;
      x[k] = (live && g < quads) ? __builtin_nontemporal_load(src + g) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int g = g0 + k * 256;
      if (g >= quads) break;
      uint32_t oh[2] = {0, 0}, ol[2] = {0, 0};
      if (live) {
        const float xe[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d2 = xe[e] - l2;  // log2 p
          const float pr = fast_exp2(d2);
          float val = ngi * pr;
          if (nhi != 0.0f) val = __builtin_fmaf(nhi * pr, __builtin_fmaf(d2, kLn2, H), val);
          if (g * 4 + e == id) val += gi;
          uint16_t h16, l16;
          split2(val, h16, l16);
          oh[e >> 1] |= (uint32_t)h16 << (16 * (e & 1));
          ol[e >> 1] |= (uint32_t)l16 << (16 * (e & 1));
        }
      }
      hi[g] = uint2{oh[0], oh[1]};
      lo[g] = uint2{ol[0], ol[1]};
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
  // split-K: workgroup (tile, kz) contracts steps [kz * ksteps, ...) of every term and stores its fp32 partial
  // tile to partial[kz][M][N]; splitk_reduce_kernel adds the slices in a fixed order.  ksplit == 1: off.
  int ksplit, ksteps;
  float* partial;
};

template <class C, bool DUAL = false>
__global__ __launch_bounds__(C::NT, 2) void gemm_nt_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int tm, tn;
  const int tiles = a.mt * a.nt;
  const int kz = a.ksplit > 1 ? (int)blockIdx.x / tiles : 0;
  tile_coords(a.ksplit > 1 ? (int)blockIdx.x - kz * tiles : (int)blockIdx.x, a.mt, a.nt, tm, tn);
  constexpr int NJ = C::NJ;
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * 64, wcol0 = (wave & 1) * C::WCOLS;
  f32x16 acc[2][NJ];
  zero_acc<NJ>(acc);
  if (a.ksplit > 1) {
    Terms t = a.terms;
    Geom g = a.geo;
    const int k0 = kz * a.ksteps * BK;
    const int left = g.Kc - k0;
    g.Kc = left < a.ksteps * BK ? left : a.ksteps * BK;
#pragma unroll
    for (int k = 0; k < MAX_TERMS; ++k) {
      t.a[k] += k0;
      t.b[k] += k0;
    }
    run_mainloop<C, DUAL>(acc, t, g, m0, n0, lds);
  } else {
    run_mainloop<C, DUAL>(acc, a.terms, a.geo, m0, n0, lds);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + acc_row(lane, wrow0, i, r);
      if (row >= a.geo.M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n0 + acc_col(lane, wcol0, j);  // 32 consecutive lanes -> 32 consecutive columns
        if (col >= a.geo.N) continue;
        if (a.ksplit > 1) {
          a.partial[((int64_t)kz * a.geo.M + row) * a.geo.N + col] = acc[i][j][r];
          continue;
        }
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

// d hidden on the triple-plane core.  Work item = (output tile, contraction slice kz).  With 8 slices every XCD works on
// ONE slice (block b runs on XCD b % 8): the 32 workgroups an XCD runs at a time are an 8 x 4 patch of output tiles that
// all walk the same 1/8 of the vocabulary, so each staged d-logits tile is wanted by 4 of them and each weight tile by 8,
// out of the XCD's own L2.  The generic kernel spread the slices of a tile over the XCDs and ran the three products one
// after the other: L2 hit 64 %, HBM fetch 7 x the operands (profiles/r02aj); here 89.5 % (profiles/r03g).
// The contraction can run in SEGMENTS with an empty pipeline in between (the forward's per-tile refill, which keeps ITS
// workgroups in step); measured here it buys nothing - segments of 16 / 32 / 64 / 128 / 256 stages and none at all:
// 53.0 / 52.1 / 52.0-52.2 / 51.9 / 51.7 / 51.7 ms for the whole backward (profiles/r03p_dh_segments.txt) - the shared
// slice alone keeps the patch together.  Default: one segment per slice; PRL_TUNE_LMHEAD_SEG sets a length.
constexpr int kSegSteps = 1 << 20;

struct Dh3Args {
  const uint16_t *a1, *a2, *b1, *b2;  // d logits hi / lo [M, K], W^T hi / lo [N, K]
  Geom geo;                            // M = rows, N = hidden, Kc = vocab
  int mt, nt;
  int ksplit, ksteps;                  // stages (of 32) per slice
  float* partial;                      // [ksplit][M][N] (ksplit > 1) or the fp32 output itself (ksplit == 1)
  int seg;                             // stages per segment
};

// TRIPLE false: two products that share W^T_hi - (dl_hi + dl_lo) W^T_hi, the whole d hidden of a bf16 weight (b2 unused) - on the
// dual-plane core of the forward: same work items, same raster, three staged tiles per stage instead of four.
// (Round 4, measured and not kept: the three products as 2 + 1 - (dl_hi + dl_lo) W_hi on the phase-shifted dual-plane core, then
// dl_hi W_lo on the generic core as a hand-placed stream, into the same accumulators: 24.7 ms against 24.4 for the triple-plane
// core below on the same box (profiles/r04q_*).  Unlike the forward, this product streams its A operand - 5 GB of d-logits
// planes, each token tile re-read by 14 column tiles - and is bound by that traffic, not by the schedule.)
template <bool TRIPLE, bool HAND = true>
__global__ __launch_bounds__(CfgTriple::NT, 2) void gemm_dh_kernel(Dh3Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  using C = CfgTriple;  // (the tile geometry of CfgDual is the same)
  static_assert(CfgTriple::BM == CfgDual::BM && CfgTriple::BN == CfgDual::BN && CfgTriple::NT == CfgDual::NT && CfgTriple::WCOLS == CfgDual::WCOLS, "");
  const int tiles = a.mt * a.nt;
  int kz, L;
  if (a.ksplit == 8) {  // slice = XCD; the tile list of a slice is walked in groups of 8 row tiles, row-fastest
    kz = (int)blockIdx.x & 7;
    L = (int)blockIdx.x >> 3;
  } else {
    kz = (int)blockIdx.x / tiles;
    L = (int)blockIdx.x - kz * tiles;
  }
  constexpr int GM = 8;
  const int per_group = GM * a.nt;
  const int grp = L / per_group;
  const int first_m = grp * GM;
  const int gsz = (a.mt - first_m) < GM ? (a.mt - first_m) : GM;
  const int in = L - grp * per_group;
  const int tm = first_m + in % gsz, tn = in / gsz;
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * 64, wcol0 = (wave & 1) * C::WCOLS;
  f32x16 acc[2][4];
  zero_acc<4>(acc);
  const int total_steps = a.geo.Kc / BK32;
  const int s0 = kz * a.ksteps;
  int s1 = s0 + a.ksteps;
  s1 = s1 < total_steps ? s1 : total_steps;
  for (int s = s0; s < s1; s += a.seg) {
    const int n = (s1 - s) < a.seg ? (s1 - s) : a.seg;
    Geom g = a.geo;
    g.Kc = n * BK32;
    const int64_t k0 = (int64_t)s * BK32;
    if constexpr (TRIPLE) {
      gemm_mainloop_triple<HAND>(acc, a.a1 + k0, a.a2 + k0, a.b1 + k0, a.b2 + k0, g, m0, n0, lds);
    } else {
      if constexpr (HAND) {
        gemm_mainloop_dual_ps(acc, a.a1 + k0, a.a2 + k0, a.b1 + k0, g, m0, n0, lds);
      } else {
        gemm_mainloop_dual<1>(acc, a.a1 + k0, a.a2 + k0, a.b1 + k0, g, m0, n0, lds);
      }
    }
  }
  float* out = a.partial + (a.ksplit > 1 ? (int64_t)kz * a.geo.M * a.geo.N : 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + acc_row(lane, wrow0, i, r);
      if (row >= a.geo.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + acc_col(lane, wcol0, j);  // 32 consecutive lanes -> 32 consecutive columns
        if (col < a.geo.N) out[(int64_t)row * a.geo.N + col] = acc[i][j][r];
      }
    }
}

// d W on the transposed-A dual-plane core: out[v, n] (+)= sum over the chunk's tokens t of (dl_hi + dl_lo)[t, v] hT[n, t].
// terms.a[0] / a[1] = the ROW-MAJOR d-logits planes [Kc, lda], terms.b[0] = hidden^T [N, ldb].
template <int HAND = 0>
__global__ __launch_bounds__(CfgDual::NT, 2) void gemm_dw_tr_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  using C = CfgDual;
  int tm, tn;
  // row-tile groups of a.ksteps (reused as the raster's group size here): the workgroups an XCD runs at a time should cover
  // FEW vocabulary tiles and ALL hidden tiles - the d-logits planes (5 GB per micro-batch) then stream from HBM once instead of
  // once per quartet of hidden tiles, while the re-read operand is the 58 MB of hidden^T that the Infinity Cache holds
  if (a.ksteps < 0) {
    // XCD-local patches (measurement, PRL_TUNE_LMHEAD_DW_GROUP = -PA): block b runs on XCD b % 8; an XCD owns the vocabulary tiles
    // x, x + 8, ... and walks them in patches of PA tiles x all hidden tiles, vocabulary-fastest
    const int PA = -a.ksteps;
    const int x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int per_patch = PA * a.nt;
    const int p = slot / per_patch, in = slot - p * per_patch;
    tm = x + 8 * (p * PA + in % PA);
    tn = in / PA;
    if (tm >= a.mt) return;  // the whole workgroup: before any barrier
  } else {
    tile_coords_g((int)blockIdx.x, a.mt, a.nt, a.ksteps > 0 ? a.ksteps : 8, tm, tn);
  }
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * 64, wcol0 = (wave & 1) * C::WCOLS;
  f32x16 acc[2][4];
  zero_acc<4>(acc);
  if constexpr (HAND == 2) {  // the phase-shifted step (gemm_mainloop_dual_ps<true>)
    gemm_mainloop_dual_ps<true>(acc, a.terms.a[0], a.terms.a[1], a.terms.b[0], a.geo, m0, n0, lds);
  } else {
    gemm_mainloop_dual_tr<HAND>(acc, a.terms.a[0], a.terms.a[1], a.terms.b[0], a.geo, m0, n0, lds);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + acc_row(lane, wrow0, i, r);
      if (row >= a.geo.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + acc_col(lane, wcol0, j);
        if (col >= a.geo.N) continue;
        float* dst = static_cast<float*>(a.out) + (int64_t)row * a.ldc + col;
        *dst = a.accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
      }
    }
}

// out[m, n] = sum over kz (ascending) of partial[kz][m][n]; M * N is a multiple of 4 (N = hidden is a multiple of 64)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int64_t quads, int64_t plane, int ksplit, const float* __restrict__ partial,
                                                            void* out, int out_bf16) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= quads) return;
  float4 sum = reinterpret_cast<const float4*>(partial)[u];
  for (int k = 1; k < ksplit; ++k) {
    const float4 x = reinterpret_cast<const float4*>(partial + (int64_t)k * plane)[u];
    sum.x += x.x;
    sum.y += x.y;
    sum.z += x.z;
    sum.w += x.w;
  }
  if (out_bf16) {
    reinterpret_cast<uint2*>(out)[u] = uint2{(uint32_t)to_bf16(sum.x) | ((uint32_t)to_bf16(sum.y) << 16),
                                             (uint32_t)to_bf16(sum.z) | ((uint32_t)to_bf16(sum.w) << 16)};
  } else {
    reinterpret_cast<float4*>(out)[u] = sum;
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

// ---- mixed-precision operand preparation -------------------------------------------------------------------
// max |x| of a tensor as the bit pattern of a non-negative float (orders like an unsigned integer): *out must be 0 before
template <class SRC>
__global__ __launch_bounds__(256) void absmax_kernel(int64_t n, const SRC* __restrict__ src, uint32_t* out) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x;
    if constexpr (sizeof(SRC) == 4) {
      x = src[i];
    } else {
      x = bf16_to_f32(src[i]);
    }
    x = fabsf(x);
    m = (x == x && x < 3.0e38f) ? fmaxf(m, x) : m;  // NaN / inf do not define a scale (they still poison the product)
  }
  m = prl::wave_max(m);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(out, __float_as_uint(m));
  }
}

// power-of-two scale S with max |x| S in [2^14, 2^15): f16 keeps 11 significant bits down to 2^-14, i.e. 2^-29 of the largest
__device__ __forceinline__ float mx_scale_of(uint32_t max_bits) {
  const float mx = __uint_as_float(max_bits);
  if (!(mx > 0.0f)) return 1.0f;
  int e;
  frexpf(mx, &e);  // mx = m 2^e, m in [0.5, 1)
  e = 15 - e;
  e = e > 40 ? 40 : (e < -40 ? -40 : e);
  return ldexpf(1.0f, e);
}

__device__ __forceinline__ uint16_t f16_bits(float x) {
  const _Float16 h = (_Float16)x;  // round to nearest even
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float f16_value(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

// x [R, K] (fp32 or bf16) -> x16 = f16(x S) [R, K] and, optionally, the residual plane x8 = fp8_e4m3((x S - x16) * 16) in the
// slot order of the MX core (prl_lmhead_layout.h: per row and 32-deep block [half 0: 16 bytes][half 1: 16 bytes]); the
// transposed planes [K, ldt] likewise (their contraction index is the ROW of x).  One thread = 8 consecutive elements of a row.
// S = mx_scale_of(*max_bits) is published to *scale_out by thread 0.
// RESIDUAL false: x8 = fp8(x16 / 128), the plane itself rounded to fp8 (the partner of another operand's residual plane).
template <class SRC, bool RESIDUAL>
__global__ __launch_bounds__(256) void mx_convert_kernel(int64_t R, int64_t K, const SRC* __restrict__ src, const uint32_t* max_bits,
                                                         float* scale_out, uint16_t* x16, uint8_t* x8) {
  const float S = mx_scale_of(*max_bits);
  if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) *scale_out = S;
  const int64_t groups = R * (K / 8);
  for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = gidx / (K / 8), k0 = (gidx % (K / 8)) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (sizeof(SRC) == 4) {
        v[e] = src[row * K + k0 + e] * S;
      } else {
        v[e] = bf16_to_f32(src[row * K + k0 + e]) * S;
      }
    }
    uint16_t h[8];
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h[e] = f16_bits(v[e]);
      r[e] = RESIDUAL ? (v[e] - f16_value(h[e])) * kLo8Mul : f16_value(h[e]) * (1.0f / kHi8Div);
    }
    uint4 pk;
    pk.x = h[0] | ((uint32_t)h[1] << 16);
    pk.y = h[2] | ((uint32_t)h[3] << 16);
    pk.z = h[4] | ((uint32_t)h[5] << 16);
    pk.w = h[6] | ((uint32_t)h[7] << 16);
    *reinterpret_cast<uint4*>(x16 + row * K + k0) = pk;
    if (x8) {
      int d0 = 0, d1 = 0;
      d0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], d0, false);
      d0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], d0, true);
      d1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[4], r[5], d1, false);
      d1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[6], r[7], d1, true);
      const int kb = (int)(k0 & 31);  // 0, 8, 16, 24: eight consecutive elements are eight consecutive slots
      *reinterpret_cast<uint2*>(x8 + row * K + (k0 - kb) + mx_byte_in_block(kb)) = uint2{(uint32_t)d0, (uint32_t)d1};
    }
  }
}

// Workgroup shape per launch.  PRL_TUNE_LMHEAD_TILE = 128 | 256 | 512 (= 256 x 256) forces one (a table read, no
// getenv: one process can A/B them); default: the largest tile whose grid still fills the 256 CUs.
enum Shape { kSmall = 0, kBig = 1, kWide = 2 };
Shape pick_shape(int64_t m_rows, int64_t n_cols) {
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
// against the hidden states; d logits hi / lo against the transposed hidden states) and the 256 x 256 shape
// was chosen.  PRL_TUNE_LMHEAD_DUAL = 0 keeps the generic core (A/B reference).
bool use_dual(Shape shape, const Terms& t) {
  if (shape != kWide || t.n != 2 || t.b[0] != t.b[1]) return false;
  return prl::tuning(PRL_TUNE_LMHEAD_DUAL, 1) != 0;
}
// 384 | 320: run the forward on the one-wave-per-SIMD core with that token tile (a table read; 0: not selected)
int one_wave_bn() {
  const int64_t v = prl::tuning(PRL_TUNE_LMHEAD_TILE, 0);
  return (v == 384 || v == 320) ? (int)v : 0;
}
int shape_bm(Shape s) { return s == kSmall ? 128 : 256; }
int shape_bn(Shape s) { return s == kWide ? 256 : 128; }

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

// the d-logits kernel also stages both planes of HALF an output tile in LDS (its epilogue): 2 x [BN / 2][BM * 2 + 8] bytes
template <class C>
constexpr int dl_lds_bytes() {
  constexpr int e = 2 * (C::BN / 2) * (C::BM * 2 + 8);
  return e > C::LDS_BYTES ? e : C::LDS_BYTES;
}
#define PRL_LAUNCH_DL(shape, blocks, args, s, name)                                                                                  \
  ((shape) == kWide  ? launch_tiles(lmhead_dlogits_kernel<CfgWide>, CfgWide::NT, dl_lds_bytes<CfgWide>(), blocks, args, s, name)      \
   : (shape) == kBig ? launch_tiles(lmhead_dlogits_kernel<CfgBig>, CfgBig::NT, dl_lds_bytes<CfgBig>(), blocks, args, s, name)         \
                     : launch_tiles(lmhead_dlogits_kernel<CfgSmall>, CfgSmall::NT, dl_lds_bytes<CfgSmall>(), blocks, args, s, name))

#define PRL_LAUNCH_CFG(shape, KERNEL, blocks, args, s, name)                                                            \
  ((shape) == kWide  ? launch_tiles(KERNEL<CfgWide>, CfgWide::NT, CfgWide::LDS_BYTES, blocks, args, s, name)            \
   : (shape) == kBig ? launch_tiles(KERNEL<CfgBig>, CfgBig::NT, CfgBig::LDS_BYTES, blocks, args, s, name)               \
                     : launch_tiles(KERNEL<CfgSmall>, CfgSmall::NT, CfgSmall::LDS_BYTES, blocks, args, s, name))

int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// vocabulary splits of the forward: enough workgroups to fill 256 CUs (x 2 for the small shape)
int fwd_nsplit(int token_tiles, int vocab_tiles, bool one_per_cu) {
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

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct BwdLayout {
  int chunk_pad;
  size_t hT, dl_hi, dl_lo, dh_partial, h16, h8, maxbits, total;
};

// Split-K factor of the d hidden product: its grid (chunk rows / 256 x hidden / 256 = 224 tiles at the 7B shape)
// leaves CUs idle in the one round it runs, while the contraction is 152 064 x 3 long.  Pick the factor (<= 8)
// whose grid fills whole rounds of 256 workgroups best; every slice keeps at least 64 steps.
constexpr int kMaxKSplit = 8;
int pick_ksplit(int tiles, int ksteps_total) {
  {
    const int v = (int)prl::tuning(PRL_TUNE_LMHEAD_KSPLIT, 0);
    if (v >= 1 && v <= kMaxKSplit && v <= ksteps_total) return v;
  }
  int best = 1;
  double best_eff = 0.0;
  for (int ks = 1; ks <= kMaxKSplit; ++ks) {
    if (ks > 1 && ksteps_total / ks < 64) break;
    const int64_t wg = (int64_t)tiles * ks;
    const double eff = (double)wg / (double)(((wg + 255) / 256) * 256);
    if (eff > best_eff + 0.02) {  // a larger factor has to buy at least 2 %
      best_eff = eff;
      best = ks;
    }
  }
  return best;
}

BwdLayout bwd_layout(int64_t hidden, int64_t vocab, int64_t chunk_rows) {
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
  // the mixed-precision recompute reads the chunk's hidden states as f16 + fp8
  L.h16 = o;
  o += align256((size_t)L.chunk_pad * hidden * 2);
  L.h8 = o;
  o += align256((size_t)L.chunk_pad * hidden);
  L.maxbits = o;
  o += 256;
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
  // tokens padded to 256; the split count depends on the workgroup shape chosen at launch: size for the largest
  const int64_t padded = (int64_t)ceil_div(n, 256) * 256;
  int ns = 1;
  for (int sh = 0; sh < 3; ++sh) {
    const Shape shp = (Shape)sh;
    const int k = fwd_nsplit(ceil_div(n, shape_bn(shp)), ceil_div(vocab, shape_bm(shp)), shp != kSmall);
    ns = k > ns ? k : ns;
  }
  int64_t padded_max = padded;
  for (int bn : {384, 320}) {  // the one-wave-per-SIMD forward pads the tokens to its own tile and picks its own split count
    const int tt1 = ceil_div(n, bn);
    const int k = fwd_nsplit(tt1, ceil_div(vocab, 256), true);
    ns = k > ns ? k : ns;
    padded_max = (int64_t)tt1 * bn > padded_max ? (int64_t)tt1 * bn : padded_max;
  }
  if (fwd_bytes) *fwd_bytes = align256((size_t)ns * padded_max * 16) + align256((size_t)padded_max * 4);
  if (bwd_bytes) {
    PRL_CHECK_ARG(chunk_rows >= 1, "chunk_rows must be >= 1");
    *bwd_bytes = bwd_layout(hidden, vocab, chunk_rows < n ? chunk_rows : n).total;
  }
  return PRL_OK;
}

static int lm_head_fwd_impl(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                           const uint16_t* w_hi, const uint16_t* w_lo, const int64_t* input_ids,
                           float temperature, float* new_logprobs, float* entropy, float* lse2, float* logits2,
                           void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(!logits2 || (vocab % 8 == 0 && prl::aligned16(logits2)), "kept logits need a vocabulary that is a multiple of 8 and a 16-byte aligned buffer");
  PRL_CHECK_ARG(hidden >= BK && hidden % BK == 0, "hidden size %lld must be a multiple of %d", (long long)hidden, BK);
  PRL_CHECK_ARG(vocab >= 1 && vocab < ((int64_t)1 << 31) - 256, "vocab out of range");
  PRL_CHECK_ARG(rows * cols < ((int64_t)1 << 31) - 256, "too many rows");
  PRL_CHECK_ARG(hidden_bf16 && w_hi && input_ids && new_logprobs && entropy && lse2 && workspace, "null pointer");
  PRL_CHECK_ARG(prl::aligned16(hidden_bf16) && prl::aligned16(w_hi) && (!w_lo || prl::aligned16(w_lo)), "operands must be 16-byte aligned");
  PRL_CHECK_ARG(temperature > 0.0f, "temperature must be > 0");
  const int64_t n = rows * cols;
  FwdArgs a;
  a.terms.n = w_lo ? 2 : 1;
  for (int k = 0; k < MAX_TERMS; ++k) {
    a.terms.a[k] = (k == 1 && w_lo) ? w_lo : w_hi;
    a.terms.b[k] = hidden_bf16;
  }
  a.geo = Geom{(int)vocab, (int)n, (int)hidden, hidden, hidden};
  a.cols = cols;
  a.ids = input_ids;
  a.k2 = kLog2e / temperature;
  const Shape shape = pick_shape(vocab, n);
  const int one_bn = one_wave_bn();  // 384 | 320: the one-wave-per-SIMD core (PRL_TUNE_LMHEAD_TILE), 0: the 8-wave shapes
  a.tt = ceil_div(n, one_bn ? one_bn : shape_bn(shape));
  a.vt = ceil_div(vocab, one_bn ? 256 : shape_bm(shape));
  a.nsplit = fwd_nsplit(a.tt, a.vt, one_bn || shape != kSmall);
  a.padded = (int64_t)a.tt * (one_bn ? one_bn : shape_bn(shape));
  const size_t part_bytes = align256((size_t)a.nsplit * a.padded * 16);
  const size_t need = part_bytes + align256((size_t)a.padded * 4);
  if (workspace_bytes < need) return prl::set_error(PRL_ENOMEM, "lm_head forward workspace: %zu bytes given, %zu needed", workspace_bytes, need);
  a.part = static_cast<float*>(workspace);
  a.ysel = reinterpret_cast<float*>(static_cast<char*>(workspace) + part_bytes);
  a.scales = nullptr;
  a.logits2 = logits2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int exp_bits = (int)prl::tuning(PRL_TUNE_LMHEAD_EXP, 0);
  if (one_bn) {
    const int blocks = a.tt * a.nsplit;
    int rc;
    // the hand-placed stream (MODE 2) is the one-wave core; PRL_TUNE_LMHEAD_EXP = 256 selects the compiler-placed MODE 0 (A/B reference)
    if (one_bn == 384 && exp_bits == 256) {
      rc = w_lo ? launch_tiles(lmhead_fwd1_kernel<384, true, 0>, 256, CfgOne<384>::LDS_BYTES, blocks, a, s, "lmhead_fwd1_kernel<384,grouped dma>")
                : launch_tiles(lmhead_fwd1_kernel<384, false, 0>, 256, CfgOne<384>::LDS_BYTES, blocks, a, s, "lmhead_fwd1_kernel<384,one plane,grouped dma>");
    } else if (one_bn == 384) {
      rc = w_lo ? launch_tiles(lmhead_fwd1_kernel<384, true, 2>, 256, CfgOne<384>::LDS_BYTES, blocks, a, s, "lmhead_fwd1_kernel<384>")
                : launch_tiles(lmhead_fwd1_kernel<384, false, 2>, 256, CfgOne<384>::LDS_BYTES, blocks, a, s, "lmhead_fwd1_kernel<384,one plane>");
    } else {
      rc = w_lo ? launch_tiles(lmhead_fwd1_kernel<320, true, 2>, 256, CfgOne<320>::LDS_BYTES, blocks, a, s, "lmhead_fwd1_kernel<320>")
                : launch_tiles(lmhead_fwd1_kernel<320, false, 2>, 256, CfgOne<320>::LDS_BYTES, blocks, a, s, "lmhead_fwd1_kernel<320,one plane>");
    }
    if (rc) return rc;
  } else if (use_dual(shape, a.terms)) {
    if (exp_bits == 1024) {  // A/B reference: the hand-placed stream with the barrier at the step start (gemm_mainloop_dual MODE 2)
      if (int rc = PRL_LAUNCH_DUAL((lmhead_fwd_kernel<CfgDual, 1024, true>), a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(dual, hand-placed, barrier first)")) return rc;
    } else if (exp_bits == 512) {  // A/B reference: the round-2 / round-3 default (staggered wave roles, DMA pieces in bursts of six)
      if (int rc = PRL_LAUNCH_DUAL((lmhead_fwd_kernel<CfgDual, 512, true>), a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(dual, staggered bursts)")) return rc;
    } else if (exp_bits == 256) {  // A/B reference: DMA pieces interleaved with the MFMA groups, all waves alike
      if (int rc = PRL_LAUNCH_DUAL((lmhead_fwd_kernel<CfgDual, 256, true>), a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(dual, interleaved)")) return rc;
    } else if (int rc = PRL_LAUNCH_DUAL((lmhead_fwd_kernel<CfgDual, 0, true>), a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(dual)")) {
      return rc;
    }
  } else if (shape == kWide && exp_bits != 512) {
    // one plane (a bf16 weight) at the 256 x 256 shape: the generic 64-deep core as a hand-placed stream (round 4; 7B 7.88 -> 7.53 ms,
    // 32B 11.14 -> 10.56 ms, bit-identical, profiles/r04m_*); PRL_TUNE_LMHEAD_EXP = 512 selects the compiler-placed form
    if (int rc = launch_tiles(lmhead_fwd_kernel<CfgWide, 1024, 0>, CfgWide::NT, CfgWide::LDS_BYTES, a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(generic, hand-placed)")) return rc;
  } else if (int rc = PRL_LAUNCH_CFG(shape, lmhead_fwd_kernel, a.tt * a.nsplit, a, s, "lmhead_fwd_kernel")) {
    return rc;
  }
  hipLaunchKernelGGL(lmhead_fwd_finish_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, n, cols, (int)vocab, a.nsplit,
                     a.padded, a.part, a.ysel, input_ids, new_logprobs, entropy, lse2);
  PRL_LAUNCH_CHECK("lmhead_fwd_finish_kernel");
  return PRL_OK;
}

extern "C" int prl_lm_head_logprob_fwd(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                       const uint16_t* w_hi, const uint16_t* w_lo, const int64_t* input_ids,
                                       float temperature, float* new_logprobs, float* entropy, float* lse2,
                                       void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  return lm_head_fwd_impl(rows, cols, hidden, vocab, hidden_bf16, w_hi, w_lo, input_ids, temperature, new_logprobs, entropy, lse2, nullptr,
                          workspace, workspace_bytes, stream);
}

extern "C" int prl_lm_head_logprob_fwd_keep(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                            const uint16_t* w_hi, const uint16_t* w_lo, const int64_t* input_ids,
                                            float temperature, float* new_logprobs, float* entropy, float* lse2, float* logits2,
                                            void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(logits2, "null pointer");
  return lm_head_fwd_impl(rows, cols, hidden, vocab, hidden_bf16, w_hi, w_lo, input_ids, temperature, new_logprobs, entropy, lse2, logits2,
                          workspace, workspace_bytes, stream);
}

namespace {

struct MxFwdLayout {
  size_t part, ysel, h16, h8, maxbits, total;
  int64_t padded;
  int nsplit, tt, vt;
};

MxFwdLayout mx_fwd_layout(int64_t n, int64_t hidden, int64_t vocab) {
  MxFwdLayout L;
  L.tt = ceil_div(n, CfgMx::BN);
  L.vt = ceil_div(vocab, CfgMx::BM);
  L.nsplit = fwd_nsplit(L.tt, L.vt, true);
  L.padded = (int64_t)L.tt * CfgMx::BN;
  size_t o = 0;
  L.part = o;
  o += align256((size_t)L.nsplit * L.padded * 16);
  L.ysel = o;
  o += align256((size_t)L.padded * 4);
  L.h16 = o;
  o += align256((size_t)n * hidden * 2);
  L.h8 = o;
  o += align256((size_t)n * hidden);
  L.maxbits = o;
  o += 256;
  L.total = o;
  return L;
}

template <class SRC, bool RESIDUAL>
int mx_convert(int64_t R, int64_t K, const SRC* src, uint32_t* max_bits, float* scale_out, uint16_t* x16, uint8_t* x8, hipStream_t s) {
  PRL_HIP_CHECK(hipMemsetAsync(max_bits, 0, 4, s));
  const int64_t n = R * K;
  const int blocks = (int)((n / 8 + 255) / 256 < 4096 ? (n / 8 + 255) / 256 : 4096);
  hipLaunchKernelGGL((absmax_kernel<SRC>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, n, src, max_bits);
  PRL_LAUNCH_CHECK("absmax_kernel");
  hipLaunchKernelGGL((mx_convert_kernel<SRC, RESIDUAL>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, R, K, src, max_bits, scale_out, x16, x8);
  PRL_LAUNCH_CHECK("mx_convert_kernel");
  return PRL_OK;
}

}  // namespace

namespace {
struct MxRecompute {  // operands of the mixed-precision recompute (prl_lm_head_prepare_mx); w16 == nullptr: off
  const uint16_t* w16 = nullptr;
  const uint8_t* w8lo = nullptr;
  float* scales = nullptr;
};
}  // namespace

static int lm_head_bwd_impl(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                            const uint16_t* w_hi, const uint16_t* w_lo, const uint16_t* wt_hi,
                            const uint16_t* wt_lo, const int64_t* input_ids, float temperature,
                            const float* lse2, const float* entropy, const float* grad_new_logprobs,
                            const float* grad_entropy, const float* upstream, void* grad_hidden,
                            int32_t grad_hidden_dtype, float* grad_weight, int64_t chunk_rows, int32_t flags,
                            void* workspace, size_t workspace_bytes, prl_stream_t stream, const MxRecompute& mx,
                            const float* kept_logits2 = nullptr) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(hidden >= BK && hidden % BK == 0, "hidden size %lld must be a multiple of %d", (long long)hidden, BK);
  PRL_CHECK_ARG(vocab >= BK && vocab % BK == 0 && vocab < ((int64_t)1 << 31) - 256,
                "the fused backward contracts over the vocabulary: vocab %lld must be a multiple of %d", (long long)vocab, BK);
  PRL_CHECK_ARG(rows * cols < ((int64_t)1 << 31) - 256, "too many rows");
  PRL_CHECK_ARG(hidden_bf16 && (w_hi || kept_logits2) && wt_hi && input_ids && lse2 && entropy && grad_new_logprobs && workspace, "null pointer");
  PRL_CHECK_ARG(kept_logits2 || (w_lo == nullptr) == (wt_lo == nullptr), "w_lo and wt_lo go together");
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
  hipStream_t s = static_cast<hipStream_t>(stream);

  for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
    const int64_t m = (n - r0) < chunk_rows ? (n - r0) : chunk_rows;
    const int m_pad = ceil_div(m, 128) * 128;  // <= L.chunk_pad
    // ---- 1. d logits planes of this chunk: from the logits the forward kept, or by recomputing them tile by tile
    if (kept_logits2) {
      KeptArgs k{kept_logits2, vocab, r0, cols, (int)m, input_ids, lse2, entropy, grad_new_logprobs, grad_entropy, upstream,
                 1.0f / temperature, dl_hi, dl_lo};
      hipLaunchKernelGGL(dlogits_from_kept_kernel<8>, dim3((unsigned)m_pad), dim3(256), 0, s, k);
      PRL_LAUNCH_CHECK("dlogits_from_kept_kernel");
    } else {
      DlArgs d;
      d.terms.n = w_lo ? 2 : 1;
      for (int k = 0; k < MAX_TERMS; ++k) {
        d.terms.a[k] = (k == 1 && w_lo) ? w_lo : w_hi;
        d.terms.b[k] = hidden_bf16 + r0 * hidden;
      }
      d.geo = Geom{(int)vocab, (int)m, (int)hidden, hidden, hidden};
      d.row_base = r0;
      d.cols = cols;
      d.ids = input_ids;
      d.lse2 = lse2;
      d.ent = entropy;
      d.g_nlp = grad_new_logprobs;
      d.g_ent = grad_entropy;
      d.upstream = upstream;
      d.k2 = kLog2e / temperature;
      d.inv_temp = 1.0f / temperature;
      d.chunk_pad = L.chunk_pad;
      d.dl_hi = dl_hi;
      d.dl_lo = dl_lo;
      d.scales = nullptr;
      if (mx.w16) {  // recompute on the mixed-precision core: the logits are those of prl_lm_head_logprob_fwd_mx
        uint16_t* h16 = reinterpret_cast<uint16_t*>(ws + L.h16);
        uint8_t* h8 = mx.w8lo ? reinterpret_cast<uint8_t*>(ws + L.h8) : nullptr;
        if (int rc = mx_convert<uint16_t, false>(m, hidden, hidden_bf16 + r0 * hidden, reinterpret_cast<uint32_t*>(ws + L.maxbits), mx.scales + 1, h16, h8, s)) return rc;
        d.terms.n = mx.w8lo ? 2 : 1;
        for (int k = 0; k < MAX_TERMS; ++k) {
          d.terms.a[k] = k == 1 ? reinterpret_cast<const uint16_t*>(mx.w8lo) : mx.w16;
          d.terms.b[k] = k == 1 ? reinterpret_cast<const uint16_t*>(h8) : h16;
        }
        d.scales = mx.scales;
        d.vt = ceil_div(vocab, CfgMx::BM);
        d.tt = ceil_div(m_pad, CfgMx::BN);
        d.nsplit = fwd_nsplit(d.tt, d.vt, true);
        if (mx.w8lo) {
          if (int rc = launch_tiles(lmhead_dlogits_kernel<CfgMx, 2>, CfgMx::NT, dl_lds_bytes<CfgMx>(), d.tt * d.nsplit, d, s, "lmhead_dlogits_kernel(mx)")) return rc;
        } else if (int rc = launch_tiles(lmhead_dlogits_kernel<CfgMx, 3>, CfgMx::NT, dl_lds_bytes<CfgMx>(), d.tt * d.nsplit, d, s, "lmhead_dlogits_kernel(mx, f16-exact weight)")) {
          return rc;
        }
      } else {
      const Shape shape = pick_shape(vocab, m_pad);
      d.vt = ceil_div(vocab, shape_bm(shape));
      // token tiles of THIS chunk: its rows rounded up to 128 (the pad rows are written as zeros and are what the
      // d W contraction below runs over); a short last chunk does not pay for the whole buffer
      d.tt = ceil_div(m_pad, shape_bn(shape));
      d.nsplit = fwd_nsplit(d.tt, d.vt, shape != kSmall);
      if (use_dual(shape, d.terms)) {
        if (int rc = launch_tiles(lmhead_dlogits_kernel<CfgDual, 1>, CfgDual::NT, dl_lds_bytes<CfgDual>(), d.tt * d.nsplit, d, s,
                                  "lmhead_dlogits_kernel(dual)")) return rc;
      } else if (int rc = PRL_LAUNCH_DL(shape, d.tt * d.nsplit, d, s, "lmhead_dlogits_kernel")) {
        return rc;
      }
      }
    }
    // ---- 2. d hidden[chunk] = dl W  (contraction over the vocabulary; hi x hi + lo x hi + hi x lo)
    if (grad_hidden) {
      GemmArgs g;
      // PRL_LM_HEAD_DH_LEADING_TERM: only d logits_hi x W_hi.  The two dropped products are 2^-9 relative
      // corrections - the size of the rounding d hidden receives anyway when it is delivered in bf16.
      // PRL_LM_HEAD_DH_NO_WEIGHT_LO: (d logits_hi + d logits_lo) x W_hi - only the weight's low plane is dropped.
      g.terms.n = (flags & PRL_LM_HEAD_DH_LEADING_TERM) ? 1 : ((wt_lo && !(flags & PRL_LM_HEAD_DH_NO_WEIGHT_LO)) ? 3 : 2);
      g.terms.a[0] = dl_hi;
      g.terms.b[0] = wt_hi;
      g.terms.a[1] = dl_lo;
      g.terms.b[1] = wt_hi;
      g.terms.a[2] = dl_hi;
      g.terms.b[2] = wt_lo ? wt_lo : wt_hi;
      g.geo = Geom{(int)m, (int)hidden, (int)vocab, vocab, vocab};
      const Shape shape = pick_shape(m, hidden);
      g.mt = ceil_div(m, shape_bm(shape));
      g.nt = ceil_div(hidden, shape_bn(shape));
      g.ldc = hidden;
      g.out_bf16 = grad_hidden_dtype == PRL_DTYPE_BF16;
      g.accumulate = 0;
      g.out = static_cast<char*>(grad_hidden) + (size_t)r0 * hidden * (g.out_bf16 ? 2 : 4);
      g.partial = reinterpret_cast<float*>(ws + L.dh_partial);
      // PRL_TUNE_LMHEAD_BWD bit 0: 1 = the round-2 structure (generic core, the three products one after the other)
      const bool new_core = g.terms.n >= 2 && (prl::tuning(PRL_TUNE_LMHEAD_BWD, 0) & 1) == 0;
      const bool triple = g.terms.n == 3;
      int slices = 1;       // fp32 slices in g.partial to be added (and converted) into g.out; 0: the kernel wrote g.out itself
      if (new_core) {
        Dh3Args d3{dl_hi, dl_lo, wt_hi, wt_lo, g.geo, ceil_div(m, CfgTriple::BM), ceil_div(hidden, CfgTriple::BN), 1, 0, nullptr, kSegSteps};
        if (const int64_t seg = prl::tuning(PRL_TUNE_LMHEAD_SEG, 0); seg > 0) d3.seg = (int)seg;
        const int steps32 = (int)(vocab / BK32);
        d3.ksplit = pick_ksplit(d3.mt * d3.nt, steps32 / 2);
        // one slice per XCD whenever the grid then still fills whole rounds and a slice keeps at least 128 stages
        if (prl::tuning(PRL_TUNE_LMHEAD_KSPLIT, 0) == 0 && ((int64_t)d3.mt * d3.nt * 8) % 256 == 0 && steps32 / 8 >= 128) d3.ksplit = 8;
        d3.ksteps = ceil_div(steps32, d3.ksplit);
        d3.ksplit = ceil_div(steps32, d3.ksteps);  // no empty slice
        const bool direct = d3.ksplit == 1 && !g.out_bf16;  // a single fp32 slice IS the output
        d3.partial = direct ? static_cast<float*>(g.out) : g.partial;
        // PRL_TUNE_LMHEAD_BWD bit 2: the round-3 schedules (staggered wave roles, DMA pieces in bursts) instead of the hand-placed streams
        const bool old_sched = (prl::tuning(PRL_TUNE_LMHEAD_BWD, 0) & 4) != 0;
        const int dh_blocks = d3.mt * d3.nt * d3.ksplit;
        int rc;
        if (triple) {
          rc = old_sched ? launch_tiles(gemm_dh_kernel<true, false>, CfgTriple::NT, CfgTriple::LDS_BYTES, dh_blocks, d3, s, "gemm_dh_kernel(d hidden, 3 products, round-3 schedule)")
                         : launch_tiles(gemm_dh_kernel<true, true>, CfgTriple::NT, CfgTriple::LDS_BYTES, dh_blocks, d3, s, "gemm_dh_kernel(d hidden, 3 products)");
        } else {
          rc = old_sched ? launch_tiles(gemm_dh_kernel<false, false>, CfgDual::NT, CfgDual::LDS_BYTES, dh_blocks, d3, s, "gemm_dh_kernel(d hidden, 2 products, round-3 schedule)")
                         : launch_tiles(gemm_dh_kernel<false, true>, CfgDual::NT, CfgDual::LDS_BYTES, dh_blocks, d3, s, "gemm_dh_kernel(d hidden, 2 products)");
        }
        if (rc) return rc;
        slices = direct ? 0 : d3.ksplit;
      } else {
        const int steps = (int)(vocab / BK);
        g.ksplit = pick_ksplit(g.mt * g.nt, steps);
        g.ksteps = ceil_div(steps, g.ksplit);
        g.ksplit = ceil_div(steps, g.ksteps);  // no empty slice
        if (int rc = PRL_LAUNCH_CFG(shape, gemm_nt_kernel, g.mt * g.nt * g.ksplit, g, s, "gemm_nt_kernel(d hidden)")) return rc;
        slices = g.ksplit > 1 ? g.ksplit : 0;
      }
      if (slices > 0) {
        const int64_t quads = m * hidden / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(quads, 256)), dim3(256), 0, s, quads, m * hidden, slices,
                           g.partial, g.out, g.out_bf16);
        PRL_LAUNCH_CHECK("splitk_reduce_kernel");
      }
    }
    // ---- 3. d W += dl^T h  (contraction over the chunk's rows; hidden is exact in bf16)
    if (grad_weight) {
      const dim3 tg((unsigned)ceil_div(hidden, 64), (unsigned)ceil_div(L.chunk_pad, 64));
      hipLaunchKernelGGL((split_transpose_kernel<uint16_t>), tg, dim3(256), 0, s, m, hidden, hidden_bf16 + r0 * hidden,
                         (uint16_t*)nullptr, (uint16_t*)nullptr, hT, (uint16_t*)nullptr, (int64_t)L.chunk_pad);
      PRL_LAUNCH_CHECK("split_transpose_kernel(hidden)");
      // d W gathers its MFMA fragments from the ROW-MAJOR planes with transposing LDS reads (round 2 had the recompute write
      // transposed copies of both planes for this product: 5 GB more per micro-batch, profiles/r03f_*)
      GemmArgs g;
      g.terms.n = 2;
      g.terms.a[0] = dl_hi;
      g.terms.a[1] = g.terms.a[2] = dl_lo;
      g.terms.b[0] = g.terms.b[1] = g.terms.b[2] = hT;
      g.geo = Geom{(int)vocab, (int)hidden, m_pad, vocab, L.chunk_pad};  // A: [tokens, vocab] row-major, its contraction index is the row
      g.mt = ceil_div(vocab, CfgDual::BM);
      g.nt = ceil_div(hidden, CfgDual::BN);
      g.ldc = hidden;
      g.out_bf16 = 0;
      g.accumulate = (r0 == 0 && (flags & PRL_LM_HEAD_DW_OVERWRITE)) ? 0 : 1;  // later chunks add to the first
      g.out = grad_weight;
      g.ksplit = 1;
      g.partial = nullptr;
      {  // raster group: ONE vocabulary tile with all its hidden tiles when there are many of those (measured at the 7B
        // shape, 14 hidden tiles: groups of 1 / 2 / 3 / 4 / 8 vocabulary tiles -> 53.1 / 53.3 / 53.7 / 53.8 / 54.0 ms for the
        // whole backward, profiles/r03h_dw_raster.txt); a narrow head takes as many as fill 16 CUs
        const int64_t forced = prl::tuning(PRL_TUNE_LMHEAD_DW_GROUP, 0);
        const int gm = forced > 0 ? (int)forced : 16 / g.nt;
        g.ksteps = gm < 1 ? 1 : gm;  // (the raster's group size travels in the otherwise unused split-K field)
        if (forced < 0) g.ksteps = (int)forced;
      }
      int dw_blocks = g.mt * g.nt;
      if (g.ksteps < 0) {
        const int pa = -g.ksteps, per_xcd = ceil_div(ceil_div(g.mt, 8), pa) * pa;
        dw_blocks = 8 * per_xcd * g.nt;
      }
      const int64_t bwd_bits = prl::tuning(PRL_TUNE_LMHEAD_BWD, 0);
      if (bwd_bits & 8) {  // A/B: the hand-placed stream (measured slower here, see gemm_mainloop_dual_tr)
        if (int rc = PRL_LAUNCH_DUAL(gemm_dw_tr_kernel<1>, dw_blocks, g, s, "gemm_dw_tr_kernel(d weight, hand-placed)")) return rc;
      } else if (bwd_bits & 16) {  // A/B: the phase-shifted step
        if (int rc = PRL_LAUNCH_DUAL(gemm_dw_tr_kernel<2>, dw_blocks, g, s, "gemm_dw_tr_kernel(d weight, phase-shifted)")) return rc;
      } else if (int rc = PRL_LAUNCH_DUAL(gemm_dw_tr_kernel<0>, dw_blocks, g, s, "gemm_dw_tr_kernel(d weight)")) {
        return rc;
      }
    }
  }
  return PRL_OK;
}

extern "C" int prl_lm_head_logprob_bwd(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                       const uint16_t* w_hi, const uint16_t* w_lo, const uint16_t* wt_hi,
                                       const uint16_t* wt_lo, const int64_t* input_ids, float temperature,
                                       const float* lse2, const float* entropy, const float* grad_new_logprobs,
                                       const float* grad_entropy, const float* upstream, void* grad_hidden,
                                       int32_t grad_hidden_dtype, float* grad_weight, int64_t chunk_rows, int32_t flags,
                                       void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  return lm_head_bwd_impl(rows, cols, hidden, vocab, hidden_bf16, w_hi, w_lo, wt_hi, wt_lo, input_ids, temperature, lse2, entropy,
                          grad_new_logprobs, grad_entropy, upstream, grad_hidden, grad_hidden_dtype, grad_weight, chunk_rows, flags,
                          workspace, workspace_bytes, stream, MxRecompute{});
}

extern "C" int prl_lm_head_logprob_bwd_mx(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                          const uint16_t* w16, const uint8_t* w8lo, float* scales, const uint16_t* wt_hi,
                                          const uint16_t* wt_lo, const int64_t* input_ids, float temperature,
                                          const float* lse2, const float* entropy, const float* grad_new_logprobs,
                                          const float* grad_entropy, const float* upstream, void* grad_hidden,
                                          int32_t grad_hidden_dtype, float* grad_weight, int64_t chunk_rows, int32_t flags,
                                          void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(w16 && scales, "null pointer");
  PRL_CHECK_ARG(prl::aligned16(w16) && (!w8lo || prl::aligned16(w8lo)), "operands must be 16-byte aligned");
  MxRecompute mx;
  mx.w16 = w16;
  mx.w8lo = w8lo;
  mx.scales = scales;
  // the recompute reads (w16, w8lo); w_hi / w_lo of the plain entry point are not needed: the d hidden product runs on wt_hi / wt_lo
  return lm_head_bwd_impl(rows, cols, hidden, vocab, hidden_bf16, w16, w8lo ? w16 : nullptr, wt_hi, wt_lo, input_ids, temperature, lse2, entropy,
                          grad_new_logprobs, grad_entropy, upstream, grad_hidden, grad_hidden_dtype, grad_weight, chunk_rows, flags,
                          workspace, workspace_bytes, stream, mx);
}

extern "C" int prl_lm_head_logprob_bwd_kept(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                            const float* logits2, const uint16_t* wt_hi, const uint16_t* wt_lo,
                                            const int64_t* input_ids, float temperature, const float* lse2, const float* entropy,
                                            const float* grad_new_logprobs, const float* grad_entropy, const float* upstream,
                                            void* grad_hidden, int32_t grad_hidden_dtype, float* grad_weight, int64_t chunk_rows,
                                            int32_t flags, void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(logits2 && prl::aligned16(logits2), "the kept logits must be a 16-byte aligned device buffer");
  return lm_head_bwd_impl(rows, cols, hidden, vocab, hidden_bf16, nullptr, nullptr, wt_hi, wt_lo, input_ids, temperature, lse2, entropy,
                          grad_new_logprobs, grad_entropy, upstream, grad_hidden, grad_hidden_dtype, grad_weight, chunk_rows, flags,
                          workspace, workspace_bytes, stream, MxRecompute{}, logits2);
}

// ---------------------------------------------------------------------------------------------------
// mixed-precision head (f16 + MX fp8 residual plane): operand preparation and forward
// ---------------------------------------------------------------------------------------------------
extern "C" int prl_lm_head_mx_workspace_bytes(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, size_t* fwd_bytes) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1 && hidden >= 1 && vocab >= 1 && fwd_bytes, "bad arguments");
  *fwd_bytes = mx_fwd_layout(rows * cols, hidden, vocab).total;
  return PRL_OK;
}

extern "C" int prl_lm_head_prepare_mx(int64_t vocab, int64_t hidden, const void* weight, int32_t weight_dtype, uint16_t* w16,
                                      uint8_t* w8lo, float* scales, prl_stream_t stream) {
  PRL_CHECK_ARG(vocab >= 1 && hidden >= 64 && hidden % 64 == 0 && weight && w16 && scales, "bad arguments");
  PRL_CHECK_ARG(weight_dtype == PRL_DTYPE_F32 || weight_dtype == PRL_DTYPE_BF16, "unsupported weight dtype %d", weight_dtype);
  PRL_CHECK_ARG(prl::aligned16(w16) && (!w8lo || prl::aligned16(w8lo)) && prl::aligned16(scales), "outputs must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint32_t* max_bits = reinterpret_cast<uint32_t*>(scales + 2);  // scales[0] = S_w, [1] = S_h (per call), [2] scratch
  if (weight_dtype == PRL_DTYPE_F32) return mx_convert<float, true>(vocab, hidden, static_cast<const float*>(weight), max_bits, scales, w16, w8lo, s);
  return mx_convert<uint16_t, true>(vocab, hidden, static_cast<const uint16_t*>(weight), max_bits, scales, w16, w8lo, s);
}

static int lm_head_fwd_mx_impl(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                              const uint16_t* w16, const uint8_t* w8lo, float* scales, const int64_t* input_ids,
                              float temperature, float* new_logprobs, float* entropy, float* lse2, float* logits2, void* workspace,
                              size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(!logits2 || (vocab % 8 == 0 && prl::aligned16(logits2)), "kept logits need a vocabulary that is a multiple of 8 and a 16-byte aligned buffer");
  PRL_CHECK_ARG(hidden >= 64 && hidden % 64 == 0, "hidden size %lld must be a multiple of 64", (long long)hidden);
  PRL_CHECK_ARG(vocab >= 1 && vocab < ((int64_t)1 << 31) - 256 && rows * cols < ((int64_t)1 << 31) - 256, "shape out of range");
  PRL_CHECK_ARG(hidden_bf16 && w16 && scales && input_ids && new_logprobs && entropy && lse2 && workspace, "null pointer");
  PRL_CHECK_ARG(prl::aligned16(hidden_bf16) && prl::aligned16(w16) && (!w8lo || prl::aligned16(w8lo)), "operands must be 16-byte aligned");
  PRL_CHECK_ARG(temperature > 0.0f, "temperature must be > 0");
  const int64_t n = rows * cols;
  const MxFwdLayout L = mx_fwd_layout(n, hidden, vocab);
  if (workspace_bytes < L.total) return prl::set_error(PRL_ENOMEM, "lm_head mx forward workspace: %zu bytes given, %zu needed", workspace_bytes, L.total);
  char* ws = static_cast<char*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint16_t* h16 = reinterpret_cast<uint16_t*>(ws + L.h16);
  uint8_t* h8 = w8lo ? reinterpret_cast<uint8_t*>(ws + L.h8) : nullptr;
  if (int rc = mx_convert<uint16_t, false>(n, hidden, hidden_bf16, reinterpret_cast<uint32_t*>(ws + L.maxbits), scales + 1, h16, h8, s)) return rc;
  FwdArgs a;
  a.terms.n = w8lo ? 2 : 1;
  for (int k = 0; k < MAX_TERMS; ++k) {
    a.terms.a[k] = k == 1 ? reinterpret_cast<const uint16_t*>(w8lo) : w16;
    a.terms.b[k] = k == 1 ? reinterpret_cast<const uint16_t*>(h8) : h16;
  }
  a.geo = Geom{(int)vocab, (int)n, (int)hidden, hidden, hidden};
  a.cols = cols;
  a.ids = input_ids;
  a.k2 = kLog2e / temperature;
  a.tt = L.tt;
  a.vt = L.vt;
  a.nsplit = L.nsplit;
  a.padded = L.padded;
  a.part = reinterpret_cast<float*>(ws + L.part);
  a.ysel = reinterpret_cast<float*>(ws + L.ysel);
  a.scales = scales;
  a.logits2 = logits2;
  if (w8lo) {
    if (int rc = launch_tiles(lmhead_fwd_kernel<CfgMx, 0, 2>, CfgMx::NT, CfgMx::LDS_BYTES, a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(mx)")) return rc;
  } else if (int rc = launch_tiles(lmhead_fwd_kernel<CfgMx, 0, 3>, CfgMx::NT, CfgMx::LDS_BYTES, a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(mx, f16-exact weight)")) {
    return rc;
  }
  hipLaunchKernelGGL(lmhead_fwd_finish_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, n, cols, (int)vocab, a.nsplit,
                     a.padded, a.part, a.ysel, input_ids, new_logprobs, entropy, lse2);
  PRL_LAUNCH_CHECK("lmhead_fwd_finish_kernel");
  return PRL_OK;
}

extern "C" int prl_lm_head_logprob_fwd_mx(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                          const uint16_t* w16, const uint8_t* w8lo, float* scales, const int64_t* input_ids,
                                          float temperature, float* new_logprobs, float* entropy, float* lse2, void* workspace,
                                          size_t workspace_bytes, prl_stream_t stream) {
  return lm_head_fwd_mx_impl(rows, cols, hidden, vocab, hidden_bf16, w16, w8lo, scales, input_ids, temperature, new_logprobs, entropy, lse2,
                             nullptr, workspace, workspace_bytes, stream);
}

extern "C" int prl_lm_head_logprob_fwd_mx_keep(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                                               const uint16_t* w16, const uint8_t* w8lo, float* scales, const int64_t* input_ids,
                                               float temperature, float* new_logprobs, float* entropy, float* lse2, float* logits2,
                                               void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(logits2, "null pointer");
  return lm_head_fwd_mx_impl(rows, cols, hidden, vocab, hidden_bf16, w16, w8lo, scales, input_ids, temperature, new_logprobs, entropy, lse2,
                             logits2, workspace, workspace_bytes, stream);
}
