// Fused output head, BACKWARD (prl_lm_head_logprob_bwd / _bwd_kept): d logits as two row-major bf16 planes (recomputed tile by
// tile, or in one pass over the logits the forward kept), d hidden = d logits W on the triple-plane core (one contraction slice
// per XCD), d W += d logits^T hidden with transposing LDS reads.  See prl_lmhead_fwd.hip for the scheme, prl_lmhead_core.h for the
// main loops.  Reference: autograd through lm_head + rl_step's soft-max (finetune/checkpoints.py:87-103, rl/__init__.py:204-233).

#include "prl_lmhead_core.h"

namespace {

using namespace prl::osm;
using namespace prl::lmhead;

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
};

template <class C, bool DUAL = false>
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
  const float k2 = a.k2;
  const int64_t V = a.geo.M;
  f32x16 acc[2][NJ];
  for (int tv = vt0; tv < vt1; ++tv) {
  const int m0 = tv * C::BM;
  zero_acc<NJ>(acc);
  run_mainloop<C, DUAL>(acc, a.terms, a.geo, m0, n0, lds);
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
// (Measured and not kept: running the contraction in SEGMENTS with an empty pipeline in between, the forward's per-tile refill -
// segments of 16 .. 256 stages and none at all: 53.0 .. 51.7 ms for the whole backward, profiles/r03p_dh_segments.txt; the shared
// slice alone keeps the patch together.)

struct Dh3Args {
  const uint16_t *a1, *a2, *b1, *b2;  // d logits hi / lo [M, K], W^T hi / lo [N, K]
  Geom geo;                            // M = rows, N = hidden, Kc = vocab
  int mt, nt;
  int ksplit, ksteps;                  // stages (of 32) per slice
  float* partial;                      // [ksplit][M][N] (ksplit > 1) or the fp32 output itself (ksplit == 1)
};

// TRIPLE false: two products that share W^T_hi - (dl_hi + dl_lo) W^T_hi, the whole d hidden of a bf16 weight (b2 unused) - on the
// dual-plane core of the forward: same work items, same raster, three staged tiles per stage instead of four.
// (Round 4, measured and not kept: the three products as 2 + 1 - (dl_hi + dl_lo) W_hi on the phase-shifted dual-plane core, then
// dl_hi W_lo on the generic core as a hand-placed stream, into the same accumulators: 24.7 ms against 24.4 for the triple-plane
// core below on the same box (profiles/r04q_*).  Unlike the forward, this product streams its A operand - 5 GB of d-logits
// planes, each token tile re-read by 14 column tiles - and is bound by that traffic, not by the schedule.)
template <bool TRIPLE>
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
  {
    Geom g = a.geo;
    g.Kc = (s1 - s0) * BK32;
    const int64_t k0 = (int64_t)s0 * BK32;
    if constexpr (TRIPLE) {
      gemm_mainloop_triple(acc, a.a1 + k0, a.a2 + k0, a.b1 + k0, a.b2 + k0, g, m0, n0, lds);
    } else {
      gemm_mainloop_dual_ps(acc, a.a1 + k0, a.a2 + k0, a.b1 + k0, g, m0, n0, lds);
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
__global__ __launch_bounds__(CfgDual::NT, 2) void gemm_dw_tr_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  using C = CfgDual;
  int tm, tn;
  // row-tile groups of a.ksteps (reused as the raster's group size here): the workgroups an XCD runs at a time should cover
  // FEW vocabulary tiles and ALL hidden tiles - the d-logits planes (5 GB per micro-batch) then stream from HBM once instead of
  // once per quartet of hidden tiles, while the re-read operand is the 58 MB of hidden^T that the Infinity Cache holds
  tile_coords_g((int)blockIdx.x, a.mt, a.nt, a.ksteps > 0 ? a.ksteps : 8, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * 64, wcol0 = (wave & 1) * C::WCOLS;
  f32x16 acc[2][4];
  zero_acc<4>(acc);
  gemm_mainloop_dual_tr(acc, a.terms.a[0], a.terms.a[1], a.terms.b[0], a.geo, m0, n0, lds);
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
// Split-K factor of the d hidden product: its grid (chunk rows / 256 x hidden / 256 = 224 tiles at the 7B shape)
// leaves CUs idle in the one round it runs, while the contraction is 152 064 x 3 long.  Pick the factor (<= 8)
// whose grid fills whole rounds of 256 workgroups best; every slice keeps at least 64 steps.
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


}  // namespace

static int lm_head_bwd_impl(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                            const uint16_t* w_hi, const uint16_t* w_lo, const uint16_t* wt_hi,
                            const uint16_t* wt_lo, const int64_t* input_ids, float temperature,
                            const float* lse2, const float* entropy, const float* grad_new_logprobs,
                            const float* grad_entropy, const float* upstream, void* grad_hidden,
                            int32_t grad_hidden_dtype, float* grad_weight, int64_t chunk_rows, int32_t flags,
                            void* workspace, size_t workspace_bytes, prl_stream_t stream, const float* kept_logits2 = nullptr) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(hidden >= BK && hidden % BK == 0, "hidden size %lld must be a multiple of %d", (long long)hidden, BK);
  PRL_CHECK_ARG(vocab >= BK && vocab % BK == 0 && vocab < ((int64_t)1 << 31) - 256,
                "the fused backward contracts over the vocabulary: vocab %lld must be a multiple of %d", (long long)vocab, BK);
  PRL_CHECK_ARG(rows * cols < ((int64_t)1 << 31) - 256, "too many rows");
  PRL_CHECK_ARG(vocab <= ((int64_t)1 << 23) && hidden <= ((int64_t)1 << 23), "vocab / hidden beyond 2^23: a tile's rows would not fit a 32-bit byte offset");
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
      const Shape shape = pick_shape(vocab, m_pad);
      d.vt = ceil_div(vocab, shape_bm(shape));
      // token tiles of THIS chunk: its rows rounded up to 128 (the pad rows are written as zeros and are what the
      // d W contraction below runs over); a short last chunk does not pay for the whole buffer
      d.tt = ceil_div(m_pad, shape_bn(shape));
      d.nsplit = fwd_nsplit(d.tt, d.vt, shape != kSmall);
      if (use_dual(shape, d.terms)) {
        if (int rc = launch_tiles(lmhead_dlogits_kernel<CfgDual, true>, CfgDual::NT, dl_lds_bytes<CfgDual>(), d.tt * d.nsplit, d, s,
                                  "lmhead_dlogits_kernel(dual)")) return rc;
      } else if (int rc = PRL_LAUNCH_DL(shape, d.tt * d.nsplit, d, s, "lmhead_dlogits_kernel")) {
        return rc;
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
      int slices = 1;  // fp32 slices in g.partial to be added (and converted) into g.out; 0: the kernel wrote g.out itself
      if (g.terms.n >= 2) {  // the triple-plane core (fp32 weight) or the phase-shifted dual-plane core (bf16 weight / NO_WEIGHT_LO)
        Dh3Args d3{dl_hi, dl_lo, wt_hi, wt_lo, g.geo, ceil_div(m, CfgTriple::BM), ceil_div(hidden, CfgTriple::BN), 1, 0, nullptr};
        const int steps32 = (int)(vocab / BK32);
        d3.ksplit = pick_ksplit(d3.mt * d3.nt, steps32 / 2);
        // one slice per XCD whenever the grid then still fills whole rounds and a slice keeps at least 128 stages
        if (prl::tuning(PRL_TUNE_LMHEAD_KSPLIT, 0) == 0 && ((int64_t)d3.mt * d3.nt * 8) % 256 == 0 && steps32 / 8 >= 128) d3.ksplit = 8;
        d3.ksteps = ceil_div(steps32, d3.ksplit);
        d3.ksplit = ceil_div(steps32, d3.ksteps);  // no empty slice
        const bool direct = d3.ksplit == 1 && !g.out_bf16;  // a single fp32 slice IS the output
        d3.partial = direct ? static_cast<float*>(g.out) : g.partial;
        const int dh_blocks = d3.mt * d3.nt * d3.ksplit;
        const int rc = g.terms.n == 3 ? launch_tiles(gemm_dh_kernel<true>, CfgTriple::NT, CfgTriple::LDS_BYTES, dh_blocks, d3, s, "gemm_dh_kernel(d hidden, 3 products)")
                                      : launch_tiles(gemm_dh_kernel<false>, CfgDual::NT, CfgDual::LDS_BYTES, dh_blocks, d3, s, "gemm_dh_kernel(d hidden, 2 products)");
        if (rc) return rc;
        slices = direct ? 0 : d3.ksplit;
      } else {  // leading term only: the generic core, split-K over whatever fills the CUs
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
      // raster group: ONE vocabulary tile with all its hidden tiles when there are many of those (measured at the 7B shape, 14 hidden
      // tiles: groups of 1 / 2 / 3 / 4 / 8 vocabulary tiles -> 53.1 / 53.3 / 53.7 / 53.8 / 54.0 ms for the whole backward,
      // profiles/r03h_dw_raster.txt; XCD-local patches measured no better, r03aj_*); a narrow head takes as many as fill 16 CUs
      const int gm = 16 / g.nt;
      g.ksteps = gm < 1 ? 1 : gm;  // (the raster's group size travels in the otherwise unused split-K field)
      if (int rc = PRL_LAUNCH_DUAL(gemm_dw_tr_kernel, g.mt * g.nt, g, s, "gemm_dw_tr_kernel(d weight)")) return rc;
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
                          workspace, workspace_bytes, stream);
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
                          workspace, workspace_bytes, stream, logits2);
}