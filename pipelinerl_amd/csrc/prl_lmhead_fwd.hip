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
//
// This file: the FORWARD (prl_lm_head_logprob_fwd / _fwd_keep).  The main loops are in prl_lmhead_core.h, the backward in
// prl_lmhead_bwd.hip, operand preparation and workspace sizing in prl_lmhead_prepare.hip.

#include "prl_lmhead_core.h"

namespace {

using namespace prl::osm;
using namespace prl::lmhead;

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
  float* logits2;       // nullable [n, vocab]: the logits in base-2 units (logit * log2(e) / temperature), kept for the backward
};

// DUAL: the two weight planes on the phase-shifted dual-plane core (C = CfgDual); HAND: the generic core as a hand-placed stream
template <class C, bool DUAL = false, bool HAND = false>
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

  const float k2 = a.k2;
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
    run_mainloop<C, DUAL, HAND>(acc, a.terms, a.geo, m0, n0, lds);
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

}  // namespace

static int lm_head_fwd_impl(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab, const uint16_t* hidden_bf16,
                           const uint16_t* w_hi, const uint16_t* w_lo, const int64_t* input_ids,
                           float temperature, float* new_logprobs, float* entropy, float* lse2, float* logits2,
                           void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1");
  PRL_CHECK_ARG(!logits2 || (vocab % 8 == 0 && prl::aligned16(logits2)), "kept logits need a vocabulary that is a multiple of 8 and a 16-byte aligned buffer");
  PRL_CHECK_ARG(hidden >= BK && hidden % BK == 0, "hidden size %lld must be a multiple of %d", (long long)hidden, BK);
  PRL_CHECK_ARG(vocab >= 1 && vocab < ((int64_t)1 << 31) - 256, "vocab out of range");
  PRL_CHECK_ARG(rows * cols < ((int64_t)1 << 31) - 256, "too many rows");
  // the DMA sources are a tile base + a 32-bit byte offset per lane (prl_lmhead_core.h `dma_src`): 255 rows x leading dimension x 2 bytes
  PRL_CHECK_ARG(vocab <= ((int64_t)1 << 23) && hidden <= ((int64_t)1 << 23), "vocab / hidden beyond 2^23: a tile's rows would not fit a 32-bit byte offset");
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
  a.tt = ceil_div(n, shape_bn(shape));
  a.vt = ceil_div(vocab, shape_bm(shape));
  a.nsplit = fwd_nsplit(a.tt, a.vt, shape != kSmall);
  a.padded = (int64_t)a.tt * shape_bn(shape);
  const size_t part_bytes = align256((size_t)a.nsplit * a.padded * 16);
  const size_t need = part_bytes + align256((size_t)a.padded * 4);
  if (workspace_bytes < need) return prl::set_error(PRL_ENOMEM, "lm_head forward workspace: %zu bytes given, %zu needed", workspace_bytes, need);
  a.part = static_cast<float*>(workspace);
  a.ysel = reinterpret_cast<float*>(static_cast<char*>(workspace) + part_bytes);
  a.logits2 = logits2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (use_dual(shape, a.terms)) {  // an fp32 weight (two planes) at the 256 x 256 shape
    if (int rc = PRL_LAUNCH_DUAL((lmhead_fwd_kernel<CfgDual, true>), a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(dual)")) return rc;
  } else if (shape == kWide) {      // one plane (a bf16 weight) at the 256 x 256 shape: the generic 64-deep core as a hand-placed stream
    if (int rc = launch_tiles(lmhead_fwd_kernel<CfgWide, false, true>, CfgWide::NT, CfgWide::LDS_BYTES, a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(generic, hand-placed)")) return rc;
  } else if (shape == kBig) {
    if (int rc = launch_tiles(lmhead_fwd_kernel<CfgBig>, CfgBig::NT, CfgBig::LDS_BYTES, a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(256 x 128)")) return rc;
  } else if (int rc = launch_tiles(lmhead_fwd_kernel<CfgSmall>, CfgSmall::NT, CfgSmall::LDS_BYTES, a.tt * a.nsplit, a, s, "lmhead_fwd_kernel(128 x 128)")) {
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