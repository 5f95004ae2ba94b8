// K2 + K3: GRPO / PPO / REINFORCE token loss, masked reduction, 32 statistics and
// d loss / d new_logprobs in ONE streaming pass (reference
// pipelinerl/finetune/rl/__init__.py:238-439 + rl/utils.py:26-92, which issue ~25
// elementwise launches, a Python loop over segments and ~31 .item() host syncs).
//
// HBM-bound: 56 algorithmic bytes per token (labels i64 + position_ids i64 + nine
// fp32 columns in, one fp32 gradient out).  Layout: every column is read with 16-byte
// vector loads on the UNSHIFTED axis (4 tokens per lane per iteration, 13 independent
// 16-byte loads in flight per lane); per-lane fp64 partial sums -> wave64 shuffle
// reduction -> per-block partial record -> a second tiny kernel reduces the per-block
// records in a fixed order (bitwise reproducible, no atomics, no host sync).
//
// Segment handling: the reference sums per packed segment and then across segments
// (sum_sum).  The segments tile the shifted axis exactly once, so the result is the
// plain masked sum; position_ids are still read to count sequence starts
// (num_sequences scales the kl_coef / entropy_bonus_coef stats, rl/__init__.py:431-432).

#include <cstdlib>

#include "prl_common.h"
#include "prl_token_math.h"

namespace {

using prl::kWave;

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8 resident blocks

// per-block partial record (doubles)
enum {
  P_LOSS = 0,       // sum of nan_to_num(loss_t * w)
  P_REWARD,         // sum reward / num_labels
  P_ENTROPY,
  P_OLD,
  P_NEW,
  P_REF,
  P_ADV,
  P_KL,
  P_KL_NO,
  P_ABS_LR,
  P_RATIO,
  P_RATIO_SUM,
  P_RATIO_SQ,
  P_RATIO_REF_NEW,
  P_RATIO_REF_OLD,
  P_CLAMP_RN,
  P_CLAMP_NO,
  P_TW,
  P_NUM_SUMS,  // = 18
  P_N_MASKED = P_NUM_SUMS,
  P_N_SEQ,
  P_BAD_NLP,
  P_BAD_LRRN,
  P_BAD_KL,
  P_BAD_GT,   // group_tokens <= 0 under group_normalization (reference assert :247)
  P_NUM_ADD,  // = 24 additive entries
  P_MAX_REWARD = P_NUM_ADD,
  P_MAX_ADV,
  P_MAX_KL,
  P_MAX_TW,
  P_MIN_REWARD,
  P_MIN_ADV,
  P_MIN_KL,
  P_MIN_TW,
  P_NUM = 32
};
static_assert(P_MIN_TW + 1 <= P_NUM, "partial record overflow");

struct LossArgs {
  prl_loss_config cfg;
  int64_t n;     // rows * cols
  int64_t cols;
  int packed;    // rows == 1 (position_ids counted when non-null)
  const int64_t* labels;
  const int64_t* position_ids;
  const float* nlp;
  const float* ent;
  const float* old_lp;
  const float* ref_lp;
  const float* adv;
  const float* reward;
  const float* group_tokens;
  const float* num_labels;
  const float* overflow;
  const float* ext_g;      // GSPO: per-token gradient coefficient of the token's segment
  const float* ext_clamp;  // GSPO: per-token clip indicator of the token's segment
  float* g_nlp;
  float* g_ent;
  double* partials;  // [gridDim.x][P_NUM]
};

// Per-lane partial sums are fp32 (a lane folds at most a few dozen tokens, so the rounding stays
// ~1e-7 relative, below the reference's own fp32 summation error); everything across lanes,
// waves and blocks is fp64.
struct Acc {
  double loss;  // the loss lane alone stays fp64 end to end (it is the number that trains the model)
  float s[P_NUM_ADD];
  float mx[4];
  float mn[4];
};

__device__ __forceinline__ void acc_init(Acc& a) {
  a.loss = 0.0;
#pragma unroll
  for (int i = 0; i < P_NUM_ADD; ++i) a.s[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a.mx[i] = -INFINITY;
    a.mn[i] = INFINITY;
  }
}

// mask_sum element: (x / num_labels * mask).nan_to_num(0) for a masked token; the division is
// one reciprocal per token shared by the 17 per-label statistics (<= 1 ulp from x / nl).
__device__ __forceinline__ float per_label(float x, float inv_nl) {
  return prl_nan_to_num0(x * inv_nl);
}

// One token on the unshifted axis. `valid_pos`: column >= 1 (a shifted position exists).
// Written without early returns: on real batches ~95 % of the tokens are labelled, so evaluating
// every token and selecting the results keeps the wave convergent (the branchy form of this
// function compiled to ~250 exec-mask save/restore pairs and ~170 branches per 4-token group).
template <bool FAST>
__device__ __forceinline__ void token_step(const LossArgs& a, Acc& acc, bool valid_pos,
                                           int64_t label, bool seq_start, float nlp,
                                           float ent, float old_lp, float ref_lp, float adv,
                                           float reward, float gt, float nl, float ovf,
                                           float ext_g, float ext_clamp, float& g_nlp,
                                           float& g_ent) {
  acc.s[P_N_SEQ] += seq_start ? 1.0f : 0.0f;
  const bool m = valid_pos && (label != -100);
  const float vf = valid_pos ? 1.0f : 0.0f;
  PrlTokenIn x{nlp, ent, old_lp, ref_lp, adv, reward, gt, nl, ovf};
  PrlTokenOut o;
  prl_token_eval(a.cfg, x, o);
  // the reference's finiteness asserts look at every shifted position, labelled or not
  acc.s[P_BAD_NLP] += vf * (float)o.bad_nlp;
  acc.s[P_BAD_LRRN] += vf * (float)o.bad_lrrn;
  acc.s[P_BAD_KL] += vf * (float)o.bad_kl;
  acc.s[P_BAD_GT] += (a.cfg.group_normalization && valid_pos && !(gt > 0.0f)) ? 1.0f : 0.0f;
  const bool gspo = a.cfg.policy_loss == PRL_POLICY_GSPO;
  g_nlp = m ? (gspo ? ext_g : o.g_nlp) : 0.0f;
  g_ent = m ? o.g_ent : 0.0f;
  if (gspo) o.clamp_no = ext_clamp;

  // Every per-label statistic is nan_to_num(x / num_labels) on labelled tokens.  `0 * x` is 0 for
  // finite x and NaN otherwise, so one fused chain over the 14 values tells whether nan_to_num
  // can change anything for this token; with num_labels >= 1 the products cannot overflow either.
  // Finite tokens (all of them in a healthy run) then cost a multiply and an add per statistic;
  // unlabelled tokens ride along with a zero weight.
  const float inv_nl = 1.0f / nl;
  const float ratio_sq = o.ratio_stat * o.ratio_stat;
  float probe = __builtin_fmaf(0.0f, reward, 0.0f);
  probe = __builtin_fmaf(0.0f, ent, probe);
  probe = __builtin_fmaf(0.0f, old_lp, probe);
  probe = __builtin_fmaf(0.0f, nlp, probe);
  probe = __builtin_fmaf(0.0f, ref_lp, probe);
  probe = __builtin_fmaf(0.0f, adv, probe);
  probe = __builtin_fmaf(0.0f, o.kl, probe);
  probe = __builtin_fmaf(0.0f, o.kl_new_old, probe);
  probe = __builtin_fmaf(0.0f, o.abs_lrno, probe);
  probe = __builtin_fmaf(0.0f, ratio_sq, probe);  // finite => ratio_stat finite
  probe = __builtin_fmaf(0.0f, o.exp_lrrn, probe);
  probe = __builtin_fmaf(0.0f, o.exp_ref_old, probe);
  probe = __builtin_fmaf(0.0f, o.w, probe);
  probe = __builtin_fmaf(0.0f, o.contrib, probe);
  if (FAST && probe == 0.0f && nl >= 1.0f) {
    const float mf = m ? 1.0f : 0.0f;
    const float k = mf * inv_nl;
    acc.loss += (double)(mf * o.contrib);
    acc.s[P_REWARD] += reward * k;
    acc.s[P_ENTROPY] += ent * k;
    acc.s[P_OLD] += old_lp * k;
    acc.s[P_NEW] += nlp * k;
    acc.s[P_REF] += ref_lp * k;
    acc.s[P_ADV] += adv * k;
    acc.s[P_KL] += o.kl * k;
    acc.s[P_KL_NO] += o.kl_new_old * k;
    acc.s[P_ABS_LR] += o.abs_lrno * k;
    acc.s[P_RATIO] += o.ratio_stat * k;
    acc.s[P_RATIO_SUM] += o.ratio_stat * mf;
    acc.s[P_RATIO_SQ] += ratio_sq * mf;
    acc.s[P_RATIO_REF_NEW] += o.exp_lrrn * k;
    acc.s[P_RATIO_REF_OLD] += o.exp_ref_old * k;
    acc.s[P_CLAMP_RN] += o.clamp_rn * k;
    acc.s[P_CLAMP_NO] += o.clamp_no * k;
    acc.s[P_TW] += o.w * k;
    acc.s[P_N_MASKED] += mf;
  } else if (m) {
    acc.loss += (double)o.contrib;
    acc.s[P_REWARD] += per_label(reward, inv_nl);
    acc.s[P_ENTROPY] += per_label(ent, inv_nl);
    acc.s[P_OLD] += per_label(old_lp, inv_nl);
    acc.s[P_NEW] += per_label(nlp, inv_nl);
    acc.s[P_REF] += per_label(ref_lp, inv_nl);
    acc.s[P_ADV] += per_label(adv, inv_nl);
    acc.s[P_KL] += per_label(o.kl, inv_nl);
    acc.s[P_KL_NO] += per_label(o.kl_new_old, inv_nl);
    acc.s[P_ABS_LR] += per_label(o.abs_lrno, inv_nl);
    acc.s[P_RATIO] += per_label(o.ratio_stat, inv_nl);
    acc.s[P_RATIO_SUM] += prl_nan_to_num0(o.ratio_stat);
    acc.s[P_RATIO_SQ] += prl_nan_to_num0(ratio_sq);
    acc.s[P_RATIO_REF_NEW] += per_label(o.exp_lrrn, inv_nl);
    acc.s[P_RATIO_REF_OLD] += per_label(o.exp_ref_old, inv_nl);
    acc.s[P_CLAMP_RN] += per_label(o.clamp_rn, inv_nl);
    acc.s[P_CLAMP_NO] += per_label(o.clamp_no, inv_nl);
    acc.s[P_TW] += per_label(o.w, inv_nl);
    acc.s[P_N_MASKED] += 1.0f;
  }
  const float ninf = -INFINITY, pinf = INFINITY;
  acc.mx[0] = fmaxf(acc.mx[0], m ? reward : ninf);
  acc.mn[0] = fminf(acc.mn[0], m ? reward : pinf);
  acc.mx[1] = fmaxf(acc.mx[1], m ? adv : ninf);
  acc.mn[1] = fminf(acc.mn[1], m ? adv : pinf);
  acc.mx[2] = fmaxf(acc.mx[2], m ? o.kl : ninf);
  acc.mn[2] = fminf(acc.mn[2], m ? o.kl : pinf);
  acc.mx[3] = fmaxf(acc.mx[3], m ? o.w : ninf);
  acc.mn[3] = fminf(acc.mn[3], m ? o.w : pinf);
}

__device__ __forceinline__ void block_reduce_store(const Acc& acc, double* out) {
  __shared__ double lds[kWavesPerBlock][P_NUM];
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
#pragma unroll
  for (int i = 0; i < P_NUM_ADD; ++i) {
    double v = prl::wave_sum(i == P_LOSS ? acc.loss : (double)acc.s[i]);
    if (lane == 0) lds[wid][i] = v;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float vx = prl::wave_max(acc.mx[i]);
    float vn = prl::wave_min(acc.mn[i]);
    if (lane == 0) {
      lds[wid][P_MAX_REWARD + i] = (double)vx;
      lds[wid][P_MIN_REWARD + i] = (double)vn;
    }
  }
  __syncthreads();
  if (threadIdx.x < P_NUM) {
    const int i = threadIdx.x;
    double v = lds[0][i];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) {
      const double o = lds[w][i];
      if (i < P_NUM_ADD) {
        v += o;
      } else if (i < P_MIN_REWARD) {
        v = fmax(v, o);
      } else {
        v = fmin(v, o);
      }
    }
    out[i] = v;
  }
}

// TPL consecutive tokens of every input column, fetched with the widest aligned loads (TPL = 4:
// 16 bytes per fp32 column, 2 x 16 bytes per int64 column; TPL = 2: 8 / 16 bytes).
template <int TPL>
struct TokVec {
  int64_t lab[TPL], pos[TPL];
  float nlp[TPL], ent[TPL], old[TPL], ref[TPL], adv[TPL], rew[TPL], gt[TPL], nl[TPL], ov[TPL], xg[TPL], xc[TPL];
};

template <int TPL>
__device__ __forceinline__ void load_f(const float* p, float (&o)[TPL]) {
  if constexpr (TPL == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else {
    const float2 t = *reinterpret_cast<const float2*>(p);
    o[0] = t.x; o[1] = t.y;
  }
}

template <int TPL>
__device__ __forceinline__ void load_i(const int64_t* p, int64_t (&o)[TPL]) {
#pragma unroll
  for (int k = 0; k < TPL; k += 2) {
    const longlong2 t = *reinterpret_cast<const longlong2*>(p + k);
    o[k] = t.x;
    o[k + 1] = t.y;
  }
}

template <int TPL, bool GSPO>
__device__ __forceinline__ void load_vec(const LossArgs& a, bool count_pos, int64_t u0, TokVec<TPL>& v) {
  load_i<TPL>(a.labels + u0, v.lab);
  if (count_pos) {
    load_i<TPL>(a.position_ids + u0, v.pos);
  } else {
#pragma unroll
    for (int k = 0; k < TPL; ++k) v.pos[k] = 1;
  }
  load_f<TPL>(a.nlp + u0, v.nlp);
  load_f<TPL>(a.ent + u0, v.ent);
  load_f<TPL>(a.old_lp + u0, v.old);
  load_f<TPL>(a.ref_lp + u0, v.ref);
  load_f<TPL>(a.adv + u0, v.adv);
  load_f<TPL>(a.reward + u0, v.rew);
  load_f<TPL>(a.group_tokens + u0, v.gt);
  load_f<TPL>(a.num_labels + u0, v.nl);
  load_f<TPL>(a.overflow + u0, v.ov);
  if constexpr (GSPO) {
    load_f<TPL>(a.ext_g + u0, v.xg);
    load_f<TPL>(a.ext_clamp + u0, v.xc);
  } else {
#pragma unroll
    for (int k = 0; k < TPL; ++k) v.xg[k] = v.xc[k] = 0.0f;
  }
}

// VEC = 4 / 2: all base pointers 16-byte aligned; the n % VEC tail is handled by the first lanes
// with scalar accesses.  VEC = 1: fully scalar fallback for odd views.
//
// The per-token math is ~250 VALU instructions, and the accumulators plus one group of inputs
// already take ~140 VGPRs (3 waves per SIMD), so the loads of the NEXT group are issued before
// the current group is evaluated: the SQ counters of the unpipelined loop showed the waves
// parked on s_waitcnt for 47 % of their cycles with the VALU 32 % busy.
template <int VEC, bool FAST, bool GSPO>
__global__ __launch_bounds__(kBlock) void grpo_loss_partial_kernel(LossArgs a) {
  Acc acc;
  acc_init(acc);
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  const bool count_pos = a.packed && a.position_ids != nullptr;
  const bool flat = count_pos && a.cfg.flat_micro_batches;
  const bool gspo = GSPO;

  if constexpr (VEC > 1) {
    const int64_t nv = a.n / VEC;
    TokVec<VEC> cur, nxt;
    int64_t i = tid;
    if (i < nv) load_vec<VEC, GSPO>(a, count_pos, i * VEC, cur);
    for (; i < nv; i += nthreads) {
      const int64_t u0 = i * VEC;
      const int64_t inext = i + nthreads;
      if (inext < nv) load_vec<VEC, GSPO>(a, count_pos, inext * VEC, nxt);
      // column of the first element of the group (cols may be any value >= 1)
      int64_t col = a.packed ? u0 : (u0 % a.cols);
      float g[VEC], gh[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const bool valid_pos = (col != 0) && !(flat && cur.pos[k] == 0);
        const bool seq_start = count_pos && (cur.pos[k] == 0 || (u0 + k) == 0);
        token_step<FAST>(a, acc, valid_pos, cur.lab[k], seq_start, cur.nlp[k], cur.ent[k], cur.old[k], cur.ref[k], cur.adv[k],
                         cur.rew[k], cur.gt[k], cur.nl[k], cur.ov[k], cur.xg[k], cur.xc[k], g[k], gh[k]);
        ++col;
        if (!a.packed && col == a.cols) col = 0;
      }
      if constexpr (VEC == 4) {
        if (a.g_nlp) *reinterpret_cast<float4*>(a.g_nlp + u0) = make_float4(g[0], g[1], g[2], g[3]);
        if (a.g_ent) *reinterpret_cast<float4*>(a.g_ent + u0) = make_float4(gh[0], gh[1], gh[2], gh[3]);
      } else {
        if (a.g_nlp) *reinterpret_cast<float2*>(a.g_nlp + u0) = make_float2(g[0], g[1]);
        if (a.g_ent) *reinterpret_cast<float2*>(a.g_ent + u0) = make_float2(gh[0], gh[1]);
      }
      cur = nxt;
    }
  }

  // scalar path: whole range for VEC == 1, the n % VEC tail otherwise
  {
    const int64_t begin = (VEC > 1) ? ((a.n / VEC) * VEC) : 0;
    for (int64_t u = begin + tid; u < a.n; u += nthreads) {
      const int64_t col = a.packed ? u : (u % a.cols);
      const bool seq_start = count_pos && (a.position_ids[u] == 0 || u == 0);
      float g, gh;
      token_step<FAST>(a, acc, (col != 0) && !(flat && a.position_ids[u] == 0), a.labels[u], seq_start, a.nlp[u], a.ent[u], a.old_lp[u],
                 a.ref_lp[u], a.adv[u], a.reward[u], a.group_tokens[u], a.num_labels[u],
                 a.overflow[u], gspo ? a.ext_g[u] : 0.0f, gspo ? a.ext_clamp[u] : 0.0f, g, gh);
      if (a.g_nlp) a.g_nlp[u] = g;
      if (a.g_ent) a.g_ent[u] = gh;
    }
  }

  block_reduce_store(acc, a.partials + (int64_t)blockIdx.x * P_NUM);
}

// Reduce the per-block records in a fixed order and emit the public stats vector.  1024 threads:
// 32 entries x 32 sub-lanes, each sub-lane folds every 32nd record with 4 independent partial
// accumulators so the (L2-resident) loads overlap instead of forming one latency chain.
constexpr int kFinalBlock = 1024;
constexpr int kFinalLanes = kFinalBlock / P_NUM;  // 32

__device__ __forceinline__ double combine(int i, double a, double b) {
  if (i < P_NUM_ADD) return a + b;
  if (i < P_MIN_REWARD) return fmax(a, b);
  return fmin(a, b);
}

__global__ __launch_bounds__(kFinalBlock) void grpo_loss_finalize_kernel(const double* partials,
                                                                         int nblocks, int64_t rows,
                                                                         int packed_counted,
                                                                         double* stats,
                                                                         float* loss_out) {
  __shared__ double lds[kFinalLanes][P_NUM];
  __shared__ double red[P_NUM];
  const int i = threadIdx.x & (P_NUM - 1);  // entry
  const int k = threadIdx.x / P_NUM;        // sub-lane 0..31
  const double ident = (i < P_NUM_ADD) ? 0.0 : ((i < P_MIN_REWARD) ? -INFINITY : INFINITY);
  double v0 = ident, v1 = ident, v2 = ident, v3 = ident;
  int b = k;
  for (; b + 3 * kFinalLanes < nblocks; b += 4 * kFinalLanes) {
    const double o0 = partials[(int64_t)b * P_NUM + i];
    const double o1 = partials[(int64_t)(b + kFinalLanes) * P_NUM + i];
    const double o2 = partials[(int64_t)(b + 2 * kFinalLanes) * P_NUM + i];
    const double o3 = partials[(int64_t)(b + 3 * kFinalLanes) * P_NUM + i];
    v0 = combine(i, v0, o0);
    v1 = combine(i, v1, o1);
    v2 = combine(i, v2, o2);
    v3 = combine(i, v3, o3);
  }
  for (; b < nblocks; b += kFinalLanes) v0 = combine(i, v0, partials[(int64_t)b * P_NUM + i]);
  lds[k][i] = combine(i, combine(i, v0, v1), combine(i, v2, v3));
  __syncthreads();
  if (threadIdx.x < P_NUM) {
    double r = lds[0][i];
    for (int j = 1; j < kFinalLanes; ++j) r = combine(i, r, lds[j][i]);
    red[i] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int j = 0; j < PRL_NUM_STATS; ++j) stats[j] = 0.0;
    // final loss in fp32 like the reference's scalar (rl/__init__.py:365)
    const double loss = -red[P_LOSS];
    stats[PRL_STAT_LOSS] = loss;
    stats[PRL_STAT_NUM_OUTPUT_TOKENS] = red[P_N_MASKED];
    stats[PRL_STAT_NUM_SEQUENCES] = packed_counted ? red[P_N_SEQ] : (double)rows;
    stats[PRL_STAT_REWARD] = red[P_REWARD];
    stats[PRL_STAT_MAX_REWARD] = red[P_MAX_REWARD];
    stats[PRL_STAT_MIN_REWARD] = red[P_MIN_REWARD];
    stats[PRL_STAT_ENTROPY] = red[P_ENTROPY];
    stats[PRL_STAT_OLD_LOGPROBS] = red[P_OLD];
    stats[PRL_STAT_NEW_LOGPROBS] = red[P_NEW];
    stats[PRL_STAT_REF_LOGPROBS] = red[P_REF];
    stats[PRL_STAT_ADVANTAGE] = red[P_ADV];
    stats[PRL_STAT_MAX_ADVANTAGE] = red[P_MAX_ADV];
    stats[PRL_STAT_MIN_ADVANTAGE] = red[P_MIN_ADV];
    stats[PRL_STAT_KL] = red[P_KL];
    stats[PRL_STAT_KL_NEW_OLD] = red[P_KL_NO];
    stats[PRL_STAT_MEAN_ABS_LOG_RATIO_NEW_OLD] = red[P_ABS_LR];
    stats[PRL_STAT_MAX_KL] = red[P_MAX_KL];
    stats[PRL_STAT_MIN_KL] = red[P_MIN_KL];
    stats[PRL_STAT_RATIO_NEW_OLD] = red[P_RATIO];
    stats[PRL_STAT_RATIO_NEW_OLD_SUM] = red[P_RATIO_SUM];
    stats[PRL_STAT_RATIO_NEW_OLD_SQUARED_SUM] = red[P_RATIO_SQ];
    stats[PRL_STAT_RATIO_REF_NEW] = red[P_RATIO_REF_NEW];
    stats[PRL_STAT_RATIO_REF_OLD] = red[P_RATIO_REF_OLD];
    stats[PRL_STAT_CLAMP_REF_NEW_INDICATOR] = red[P_CLAMP_RN];
    stats[PRL_STAT_CLAMP_NEW_OLD_INDICATOR] = red[P_CLAMP_NO];
    stats[PRL_STAT_TOKEN_WEIGHT] = red[P_TW];
    stats[PRL_STAT_MAX_TOKEN_WEIGHT] = red[P_MAX_TW];
    stats[PRL_STAT_MIN_TOKEN_WEIGHT] = red[P_MIN_TW];
    stats[PRL_STAT_NONFINITE_NEW_LOGPROBS] = red[P_BAD_NLP];
    stats[PRL_STAT_NONFINITE_LOG_RATIO_REF_NEW] = red[P_BAD_LRRN];
    stats[PRL_STAT_NONFINITE_KL] = red[P_BAD_KL];
    stats[PRL_STAT_BAD_GROUP_TOKENS] = red[P_BAD_GT];
    if (loss_out) *loss_out = (float)loss;
  }
}

// Tokens per lane and iteration.  Two make every load wave-contiguous (the int64 columns are 16
// bytes per lane at a 16-byte stride instead of two loads at a 32-byte stride) and fit 3 waves per
// SIMD: 7 % faster for the statistics-only launch of a whole step (347 vs 375 us); with the
// gradient written as well four tokens per lane (16-byte stores) stay ahead (390 vs 399 us).
int tokens_per_lane(bool writes_gradient) { return writes_gradient ? 4 : 2; }

int grid_for(int64_t n, int vec) {
  const int64_t items = (n + vec - 1) / vec;
  int64_t blocks = (items + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  return (int)blocks;
}

// min(wanted, CUs x resident blocks per CU) for `kernel`; the occupancy query runs once per kernel.
template <void (*kernel)(LossArgs)>
int resident_grid(int wanted) {
  struct Probe {
    int cus = 0, per_cu = 0;
  };
  static const Probe probe = [] {  // the occupancy query runs once per kernel
    Probe p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return p;
    if (hipDeviceGetAttribute(&p.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || p.cus <= 0) return Probe{};
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p.per_cu, kernel, kBlock, 0) != hipSuccess || p.per_cu <= 0) return Probe{};
    return p;
  }();
  if (probe.cus <= 0) return wanted < kMaxBlocks ? wanted : kMaxBlocks;
  const int64_t c = (int64_t)probe.cus * probe.per_cu;
  const int cap = (int)(c < kMaxBlocks ? c : kMaxBlocks);
  return wanted < cap ? wanted : cap;
}

}  // namespace

extern "C" int prl_grpo_loss_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes) {
  PRL_CHECK_ARG(bytes != nullptr, "bytes is null");
  PRL_CHECK_ARG(rows >= 0 && cols >= 0, "negative shape");
  (void)rows;
  (void)cols;
  *bytes = (size_t)kMaxBlocks * P_NUM * sizeof(double);
  return PRL_OK;
}

extern "C" int prl_grpo_loss_fwd_bwd(const prl_loss_config* cfg, int64_t rows, int64_t cols,
                                     const int64_t* labels, const int64_t* position_ids,
                                     const float* new_logprobs, const float* entropy,
                                     const float* old_logprobs, const float* ref_logprobs,
                                     const float* advantages, const float* rewards,
                                     const float* group_tokens, const float* num_labels,
                                     const float* overflow, const float* ext_token_grad,
                                     const float* ext_clamp_indicator, float* grad_new_logprobs,
                                     float* grad_entropy, float* loss_out, double* stats,
                                     void* workspace, size_t workspace_bytes,
                                     prl_stream_t stream) {
  PRL_CHECK_ARG(cfg != nullptr, "cfg is null");
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1 (got %lld x %lld)",
                (long long)rows, (long long)cols);
  PRL_CHECK_ARG(cfg->policy_loss == PRL_POLICY_PPO || cfg->policy_loss == PRL_POLICY_REINFORCE ||
                    cfg->policy_loss == PRL_POLICY_GSPO,
                "unknown policy_loss %d", cfg->policy_loss);
  PRL_CHECK_ARG(cfg->policy_loss != PRL_POLICY_GSPO || (ext_token_grad && ext_clamp_indicator),
                "PRL_POLICY_GSPO needs ext_token_grad and ext_clamp_indicator");
  PRL_CHECK_ARG(labels && new_logprobs && entropy && old_logprobs && ref_logprobs && advantages &&
                    rewards && group_tokens && num_labels && overflow && stats,
                "null input pointer");
  PRL_CHECK_ARG(workspace != nullptr, "workspace is null");
  PRL_CHECK_ARG(!cfg->flat_micro_batches || (rows == 1 && position_ids != nullptr),
                "flat_micro_batches needs a packed [1, T] batch with position_ids");
  size_t need = 0;
  prl_grpo_loss_workspace_bytes(rows, cols, &need);
  if (workspace_bytes < need)
    return prl::set_error(PRL_ENOMEM, "workspace too small: %zu < %zu", workspace_bytes, need);

  LossArgs a;
  a.cfg = *cfg;
  a.n = rows * cols;
  a.cols = cols;
  a.packed = (rows == 1);
  a.labels = labels;
  a.position_ids = position_ids;
  a.nlp = new_logprobs;
  a.ent = entropy;
  a.old_lp = old_logprobs;
  a.ref_lp = ref_logprobs;
  a.adv = advantages;
  a.reward = rewards;
  a.group_tokens = group_tokens;
  a.num_labels = num_labels;
  a.overflow = overflow;
  a.ext_g = ext_token_grad;
  a.ext_clamp = ext_clamp_indicator;
  a.g_nlp = grad_new_logprobs;
  a.g_ent = grad_entropy;
  a.partials = static_cast<double*>(workspace);

  const bool vec_ok = prl::aligned16(labels) && (!position_ids || prl::aligned16(position_ids)) &&
                      prl::aligned16(new_logprobs) && prl::aligned16(entropy) &&
                      prl::aligned16(old_logprobs) && prl::aligned16(ref_logprobs) &&
                      prl::aligned16(advantages) && prl::aligned16(rewards) &&
                      prl::aligned16(group_tokens) && prl::aligned16(num_labels) &&
                      prl::aligned16(overflow) &&
                      (!ext_token_grad || prl::aligned16(ext_token_grad)) &&
                      (!ext_clamp_indicator || prl::aligned16(ext_clamp_indicator)) &&
                      (!grad_new_logprobs || prl::aligned16(grad_new_logprobs)) &&
                      (!grad_entropy || prl::aligned16(grad_entropy));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int nblocks;
  if (vec_ok) {
    // grid = what is co-resident (the loop is grid-stride and software-pipelined): a grid of more
    // blocks than fit runs in rounds and leaves the last round partly empty
    const bool is_gspo = a.cfg.policy_loss == PRL_POLICY_GSPO;
    if (is_gspo) {
      nblocks = resident_grid<grpo_loss_partial_kernel<4, true, true>>(grid_for(a.n, 4));
      hipLaunchKernelGGL((grpo_loss_partial_kernel<4, true, true>), dim3(nblocks), dim3(kBlock), 0, s, a);
    } else if (tokens_per_lane(a.g_nlp != nullptr || a.g_ent != nullptr) == 2) {
      nblocks = resident_grid<grpo_loss_partial_kernel<2, true, false>>(grid_for(a.n, 2));
      hipLaunchKernelGGL((grpo_loss_partial_kernel<2, true, false>), dim3(nblocks), dim3(kBlock), 0, s, a);
    } else {
      nblocks = resident_grid<grpo_loss_partial_kernel<4, true, false>>(grid_for(a.n, 4));
      hipLaunchKernelGGL((grpo_loss_partial_kernel<4, true, false>), dim3(nblocks), dim3(kBlock), 0, s, a);
    }
  } else {
    nblocks = grid_for(a.n, 1);
    if (a.cfg.policy_loss == PRL_POLICY_GSPO)
      hipLaunchKernelGGL((grpo_loss_partial_kernel<1, false, true>), dim3(nblocks), dim3(kBlock), 0, s, a);
    else
      hipLaunchKernelGGL((grpo_loss_partial_kernel<1, false, false>), dim3(nblocks), dim3(kBlock), 0, s, a);
  }
  PRL_LAUNCH_CHECK("grpo_loss_partial_kernel");
  const int packed_counted = (a.packed && position_ids != nullptr) ? 1 : 0;
  hipLaunchKernelGGL(grpo_loss_finalize_kernel, dim3(1), dim3(kFinalBlock), 0, s, a.partials, nblocks,
                     rows, packed_counted, stats, loss_out);
  PRL_LAUNCH_CHECK("grpo_loss_finalize_kernel");
  return PRL_OK;
}

// --------------------------------------------------------------------------------------
// GSPO helper: per-segment masked sums (reference rl/utils.py:106-208, index_add_ x3).
// Segment ids of a packed batch (and of a sequence-parallel slice of one) are non-decreasing,
// so a segment is one contiguous run: workgroup s finds its run by binary search and reduces it
// in a FIXED order (per-thread strided fp64 partials -> wave shuffle tree -> waves in order).
// Bitwise reproducible like the rest of the loss path - no atomics, no arrival order.  A row
// whose ids are not sorted is detected by a second tiny launch and answered with NaN sums,
// which the loss turns into the reference's non-finite assertion instead of a silent error.
// --------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(kBlock) void segment_sums_kernel(int64_t cols, int32_t n_segments,
                                                              const int64_t* __restrict__ seg,
                                                              const int64_t* __restrict__ labels,
                                                              const float* __restrict__ a, const float* __restrict__ b,
                                                              double* a_sum, double* b_sum, double* count) {
  __shared__ double red[3][kBlock / prl::kWave];
  const int64_t s = blockIdx.x;
  // first u in [1, cols) with seg[u] >= s, and first with seg[u] > s (column 0 carries no prediction)
  auto bound = [&](int64_t key) {
    int64_t lo = 1, hi = cols;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (seg[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int64_t u0 = bound(s), u1 = bound(s + 1);
  double c = 0.0, sa = 0.0, sb = 0.0;
  for (int64_t u = u0 + threadIdx.x; u < u1; u += kBlock) {
    if (labels[u] == -100) continue;
    c += 1.0;
    sa += (double)a[u];
    sb += (double)b[u];
  }
  c = prl::wave_sum(c);
  sa = prl::wave_sum(sa);
  sb = prl::wave_sum(sb);
  const int lane = threadIdx.x & (prl::kWave - 1), wid = threadIdx.x / prl::kWave;
  if (lane == 0) {
    red[0][wid] = c;
    red[1][wid] = sa;
    red[2][wid] = sb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tc = 0.0, ta = 0.0, tb = 0.0;
    for (int w = 0; w < kBlock / prl::kWave; ++w) {
      tc += red[0][w];
      ta += red[1][w];
      tb += red[2][w];
    }
    count[s] = tc;
    a_sum[s] = ta;
    b_sum[s] = tb;
  }
}

// NaN the results if the ids of [1, cols) are not non-decreasing or leave [0, n_segments)
__global__ __launch_bounds__(kBlock) void segment_order_check_kernel(int64_t cols, int32_t n_segments,
                                                                     const int64_t* __restrict__ seg,
                                                                     const int64_t* __restrict__ labels,
                                                                     double* a_sum, double* b_sum, double* count) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  int mine = 0;
  for (int64_t u = 1 + threadIdx.x; u < cols; u += kBlock) {
    if (u + 1 < cols && seg[u] > seg[u + 1]) mine = 1;
    if (labels[u] != -100 && (seg[u] < 0 || seg[u] >= n_segments)) mine = 1;
  }
  if (mine) atomicOr(&bad, 1);
  __syncthreads();
  if (bad) {
    const double nan = __builtin_nan("");
    for (int i = threadIdx.x; i < n_segments; i += kBlock) {
      a_sum[i] = nan;
      b_sum[i] = nan;
      count[i] = nan;
    }
  }
}

// GSPO (rl/__init__.py:310-352): the four per-segment sums of one micro-batch in ONE pass - log(new/old), advantages, token
// count, token weight - with the per-token columns computed in the loop's prologue (new - old; 1 / group_tokens or the batch
// weight, x (1 - overflow)): what `gspo_segment_terms` used to materialise as [1, T] tensors with eager launches before two
// calls of segment_sums_kernel.  Same run search, same fixed-order fp64 reduction.
__global__ __launch_bounds__(kBlock) void gspo_segment_sums_kernel(prl_loss_config cfg, int64_t cols, int32_t n_segments,
                                                                   const int64_t* __restrict__ seg, const int64_t* __restrict__ labels,
                                                                   const float* __restrict__ new_lp, const float* __restrict__ old_lp,
                                                                   const float* __restrict__ adv, const float* __restrict__ group_tokens,
                                                                   const float* __restrict__ overflow, double* sums /* [4, n_segments] */) {
  __shared__ double red[4][kBlock / prl::kWave];
  const int64_t s = blockIdx.x;
  auto bound = [&](int64_t key) {
    int64_t lo = 1, hi = cols;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (seg[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int64_t u0 = bound(s), u1 = bound(s + 1);
  double c = 0.0, sl = 0.0, sa = 0.0, sw = 0.0;
  for (int64_t u = u0 + threadIdx.x; u < u1; u += kBlock) {
    if (labels[u] == -100) continue;
    float w = cfg.group_normalization ? (1.0f / group_tokens[u]) : cfg.token_weight;  // (:245-255), as prl_token_eval
    if (cfg.overlong_filtering) w = w * (1.0f - overflow[u]);
    c += 1.0;
    sl += (double)(new_lp[u] - old_lp[u]);  // fp32 difference, like the reference's log_ratio_new_old (:257)
    sa += (double)adv[u];
    sw += (double)w;
  }
  c = prl::wave_sum(c);
  sl = prl::wave_sum(sl);
  sa = prl::wave_sum(sa);
  sw = prl::wave_sum(sw);
  const int lane = threadIdx.x & (prl::kWave - 1), wid = threadIdx.x / prl::kWave;
  if (lane == 0) {
    red[0][wid] = sl;
    red[1][wid] = sa;
    red[2][wid] = c;
    red[3][wid] = sw;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int w = 0; w < kBlock / prl::kWave; ++w) t += red[threadIdx.x][w];
    sums[(int64_t)threadIdx.x * n_segments + s] = t;
  }
}

__global__ __launch_bounds__(kBlock) void gspo_order_check_kernel(int64_t cols, int32_t n_segments, const int64_t* __restrict__ seg,
                                                                  const int64_t* __restrict__ labels, double* sums) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  int mine = 0;
  for (int64_t u = 1 + threadIdx.x; u < cols; u += kBlock) {
    if (u + 1 < cols && seg[u] > seg[u + 1]) mine = 1;
    if (labels[u] != -100 && (seg[u] < 0 || seg[u] >= n_segments)) mine = 1;
  }
  if (mine) atomicOr(&bad, 1);
  __syncthreads();
  if (bad) {
    const double nan = __builtin_nan("");
    for (int i = threadIdx.x; i < 4 * n_segments; i += kBlock) sums[i] = nan;
  }
}

// Between the two per-token ends: the O(#segments) arithmetic of the sequence-level term (rl/__init__.py:316-343) in one small
// launch - clipped sequence ratio, its loss contribution, the coefficient d loss / d new_logprobs shares over a segment's tokens
// and the clip indicator - instead of ~20 eager launches on [S]-sized tensors.  One workgroup; the loss is reduced in a fixed order.
__global__ __launch_bounds__(kBlock) void gspo_segment_terms_kernel(prl_loss_config cfg, int32_t n_segments, const double* __restrict__ sums,
                                                                    float grad_scale, int32_t zero_out, float* __restrict__ coef,
                                                                    float* __restrict__ indicator, float* __restrict__ loss_out) {
  __shared__ double red[kBlock / prl::kWave];
  double part = 0.0;
  for (int32_t s = threadIdx.x; s < n_segments; s += kBlock) {
    const float cnt = (float)sums[2 * (int64_t)n_segments + s];
    const float den = cnt < 1e-6f ? 1e-6f : cnt;
    const float ratio = expf((float)sums[s] / den);                       // exp(mean log(new / old)) over the segment (:321-323)
    const float adv = (float)sums[(int64_t)n_segments + s] / den;         // mean advantage (:324)
    const float w = (float)sums[3 * (int64_t)n_segments + s];             // sum of the segment's token weights
    const bool valid = cnt > 0.0f && w > 0.0f;
    const float clipped = ratio < cfg.clip_lo ? cfg.clip_lo : (ratio > cfg.clip_hi ? cfg.clip_hi : ratio);
    const float s1 = ratio * adv, s2 = clipped * adv;
    const bool inside = ratio >= cfg.clip_lo && ratio <= cfg.clip_hi;
    // d min(s1, s2) / d ratio, torch.min's tie rule (half to each side), clamp passing the gradient on [lo, hi] inclusive
    const float in = inside ? 1.0f : 0.0f;
    const float dmin = s1 < s2 ? adv : (s2 < s1 ? adv * in : 0.5f * adv + 0.5f * adv * in);
    const float v = valid ? 1.0f : 0.0f;
    float c = -(w * v) * dmin * ratio / den * grad_scale;
    float m = s1 < s2 ? s1 : s2;
    if (!(s1 == s1) || !(s2 == s2)) m = s1 + s2;  // NaN propagates like torch.minimum
    if (zero_out) c = 0.0f;
    coef[s] = c;
    indicator[s] = (clipped != ratio && valid) ? 1.0f : 0.0f;
    part += (double)(m * v * w);
  }
  part = prl::wave_sum(part);
  const int lane = threadIdx.x & (prl::kWave - 1), wid = threadIdx.x / prl::kWave;
  if (lane == 0) red[wid] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kBlock / prl::kWave; ++w) t += red[w];
    *loss_out = zero_out ? 0.0f : (float)(-t);
  }
}

// The way back: every token takes its segment's gradient coefficient, and the clip indicator of the j-th sequence starting or
// continuing in this (slice of a) batch (rl/__init__.py:347-350: zip(local segments, per-segment values)).
__global__ __launch_bounds__(kBlock) void gspo_expand_kernel(int64_t cols, int32_t n_segments, const int64_t* __restrict__ seg,
                                                             const float* __restrict__ coef, const float* __restrict__ indicator,
                                                             float* __restrict__ token_grad, float* __restrict__ token_indicator) {
  const int64_t first = seg[0];
  const int64_t top = n_segments > 0 ? n_segments - 1 : 0;
  for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < cols; u += (int64_t)gridDim.x * kBlock) {
    int64_t g = seg[u];
    g = g < 0 ? 0 : (g > top ? top : g);
    int64_t l = seg[u] - first;
    l = l < 0 ? 0 : (l > top ? top : l);
    token_grad[u] = coef[g];
    token_indicator[u] = indicator[l];
  }
}

}  // namespace

extern "C" int prl_gspo_segment_sums(const prl_loss_config* cfg, int64_t cols, int32_t n_segments, const int64_t* segment_ids,
                                     const int64_t* labels, const float* new_logprobs, const float* old_logprobs,
                                     const float* advantages, const float* group_tokens, const float* overflow,
                                     double* sums, prl_stream_t stream) {
  PRL_CHECK_ARG(cfg != nullptr, "null config");
  PRL_CHECK_ARG(cols >= 0 && n_segments >= 0, "negative shape");
  PRL_CHECK_ARG(segment_ids && labels && new_logprobs && old_logprobs && advantages && group_tokens && overflow && sums, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_segments == 0) return PRL_OK;
  if (cols <= 1) {
    PRL_HIP_CHECK(hipMemsetAsync(sums, 0, sizeof(double) * 4 * n_segments, s));
    return PRL_OK;
  }
  hipLaunchKernelGGL(gspo_segment_sums_kernel, dim3(n_segments), dim3(kBlock), 0, s, *cfg, cols, n_segments, segment_ids, labels,
                     new_logprobs, old_logprobs, advantages, group_tokens, overflow, sums);
  PRL_LAUNCH_CHECK("gspo_segment_sums_kernel");
  hipLaunchKernelGGL(gspo_order_check_kernel, dim3(1), dim3(kBlock), 0, s, cols, n_segments, segment_ids, labels, sums);
  PRL_LAUNCH_CHECK("gspo_order_check_kernel");
  return PRL_OK;
}

extern "C" int prl_gspo_segment_terms(const prl_loss_config* cfg, int32_t n_segments, const double* sums, float grad_scale,
                                      int32_t zero_out, float* coef, float* indicator, float* loss, prl_stream_t stream) {
  PRL_CHECK_ARG(cfg != nullptr, "null config");
  PRL_CHECK_ARG(n_segments >= 0, "negative shape");
  PRL_CHECK_ARG(loss && (n_segments == 0 || (sums && coef && indicator)), "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gspo_segment_terms_kernel, dim3(1), dim3(kBlock), 0, s, *cfg, n_segments, sums, grad_scale, zero_out, coef, indicator, loss);
  PRL_LAUNCH_CHECK("gspo_segment_terms_kernel");
  return PRL_OK;
}

extern "C" int prl_gspo_expand(int64_t cols, int32_t n_segments, const int64_t* segment_ids, const float* coef,
                               const float* indicator, float* token_grad, float* token_indicator, prl_stream_t stream) {
  PRL_CHECK_ARG(cols >= 0 && n_segments >= 1, "bad shape");
  PRL_CHECK_ARG(segment_ids && coef && indicator && token_grad && token_indicator, "null pointer");
  if (cols == 0) return PRL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t blocks = (cols + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(gspo_expand_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(kBlock), 0, s, cols, n_segments, segment_ids,
                     coef, indicator, token_grad, token_indicator);
  PRL_LAUNCH_CHECK("gspo_expand_kernel");
  return PRL_OK;
}

extern "C" int prl_segment_sums(int64_t cols, int32_t n_segments, const int64_t* segment_ids,
                                const int64_t* labels, const float* a, const float* b,
                                double* a_sum, double* b_sum, double* count,
                                prl_stream_t stream) {
  PRL_CHECK_ARG(cols >= 0 && n_segments >= 0, "negative shape");
  PRL_CHECK_ARG(segment_ids && labels && a && b && a_sum && b_sum && count, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_segments == 0) return PRL_OK;
  if (cols <= 1) {
    PRL_HIP_CHECK(hipMemsetAsync(a_sum, 0, sizeof(double) * n_segments, s));
    PRL_HIP_CHECK(hipMemsetAsync(b_sum, 0, sizeof(double) * n_segments, s));
    PRL_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(double) * n_segments, s));
    return PRL_OK;
  }
  hipLaunchKernelGGL(segment_sums_kernel, dim3(n_segments), dim3(kBlock), 0, s, cols, n_segments,
                     segment_ids, labels, a, b, a_sum, b_sum, count);
  PRL_LAUNCH_CHECK("segment_sums_kernel");
  hipLaunchKernelGGL(segment_order_check_kernel, dim3(1), dim3(kBlock), 0, s, cols, n_segments, segment_ids,
                     labels, a_sum, b_sum, count);
  PRL_LAUNCH_CHECK("segment_order_check_kernel");
  return PRL_OK;
}
