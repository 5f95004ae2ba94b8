// Per-token GRPO / PPO / REINFORCE loss math shared by the K2+K3 kernel, the fused
// logits kernel and the host-compiled unit-test harness (tests/harness/).
//
// Follows reference pipelinerl/finetune/rl/__init__.py:238-365 operation by
// operation in fp32 (build with -ffp-contract=off so no FMA contraction changes the
// rounding), with the backward derived in closed form:
//   * torch.min(surr1, surr2) routes the gradient to the smaller argument and splits
//     it 50/50 on ties; torch.clamp passes gradient on [lo, hi] inclusive.
//   * mask_sum's nan_to_num(0) maps NaN -> 0, +/-inf -> +/-FLT_MAX and zeroes the
//     gradient of non-finite elements (rl/utils.py:26-31).
#pragma once

#include <float.h>
#include <math.h>
#include <stdint.h>

#include "../../include/prl.h"

#if defined(__HIPCC__)
#define PRL_HD __host__ __device__ __forceinline__
#else
#define PRL_HD static inline
#endif

// exp() of quantities that only feed statistics (never the loss or its gradient): the hardware
// v_exp_f32 path on the device (~2 ulp), libm on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define PRL_STAT_EXPF(x) __expf(x)
#else
#define PRL_STAT_EXPF(x) expf(x)
#endif

struct PrlTokenIn {
  float nlp;           // new_logprobs[t]          (:212)
  float ent;           // entropy[t]               (:215-233)
  float old_lp;        // old_logprobs[:, 1:]      (:240)
  float ref_lp;        // ref_logprobs[:, 1:]      (:239)
  float adv;           // advantages[:, 1:]        (:272)
  float reward;        // rewards[:, 1:]           (:238)
  float group_tokens;  // group_tokens[:, 1:]      (:241)
  float num_labels;    // num_labels[:, 1:]        (:242)
  float overflow;      // overflow[:, 1:]          (:243)
};

struct PrlTokenOut {
  float contrib;      // nan_to_num(loss_t * w) for a masked token (enters -sum)
  float g_nlp;        // d total_loss / d nlp for a masked token
  float g_ent;        // d total_loss / d entropy for a masked token
  float w;            // token weight                           (:245-255)
  float A;            // log_p_weights                          (:274-276)
  float ratio_stat;   // ratio_new_old as reported in the stats (clamped for reinforce, :308)
  float kl;           // approx_kl                              (:288)
  float kl_new_old;   // approx_kl_new_old                      (:289)
  float abs_lrno;     // |log_ratio_new_old|                    (:258)
  float exp_lrrn;     // exp(log_ratio_ref_new)   (stats ratio_ref_new, :422)
  float exp_ref_old;  // exp(ref - old)           (stats ratio_ref_old, :423)
  float clamp_rn;     // indicator |log_ratio_ref_new| > C      (:278)
  float clamp_no;     // indicator ratio clipped                (:301 / :306)
  int bad_nlp;        // !isfinite(new_logprobs)                (:213)
  int bad_lrrn;       // !isfinite(log_ratio_ref_new)           (:262)
  int bad_kl;         // !isfinite(approx_kl)                   (:291)
};

PRL_HD int prl_isfinite(float v) { return (v - v) == 0.0f; }

PRL_HD float prl_nan_to_num0(float v) {
  if (v != v) return 0.0f;
  if (v > FLT_MAX) return FLT_MAX;
  if (v < -FLT_MAX) return -FLT_MAX;
  return v;
}

PRL_HD float prl_clampf(float x, float lo, float hi) {
  // torch.clamp semantics: min(max(x, lo), hi); NaN propagates.
  if (x != x) return x;
  float y = x < lo ? lo : x;
  return y > hi ? hi : y;
}

// Cheap pre-check used for every shifted position (masked or not): the reference's
// finiteness asserts look at all tokens.
PRL_HD void prl_token_flags(const prl_loss_config& c, float nlp, float ref_lp, int* bad_nlp,
                            int* bad_lrrn, int* bad_kl) {
  const float lrrn = ref_lp - nlp;
  const float C = c.clamp_log_ratio_ref_new;
  const float cl = prl_clampf(lrrn, -C, C);
  const float kl = expf(cl) - cl - 1.0f;
  *bad_nlp = !prl_isfinite(nlp);
  *bad_lrrn = !prl_isfinite(lrrn);
  *bad_kl = !prl_isfinite(kl);
}

// Full evaluation for a masked token (labels[t+1] != -100).
PRL_HD void prl_token_eval(const prl_loss_config& c, const PrlTokenIn& x, PrlTokenOut& o) {
  // token weights (:245-255)
  float w = c.group_normalization ? (1.0f / x.group_tokens) : c.token_weight;
  if (c.overlong_filtering) w = w * (1.0f - x.overflow);

  const float lrno = x.nlp - x.old_lp;   // log_ratio_new_old (:257)
  const float ratio = expf(lrno);        // (:259)
  const float lrrn = x.ref_lp - x.nlp;   // log_ratio_ref_new (:260)

  float A = c.use_advantages ? x.adv : x.reward;  // (:274)
  if (c.relu_log_p_weights) A = (A != A) ? A : (A < 0.0f ? 0.0f : A);

  const float C = c.clamp_log_ratio_ref_new;
  const float cl = prl_clampf(lrrn, -C, C);        // (:280-284)
  const float ecl = expf(cl);
  const float kl = ecl - cl - 1.0f;                // (:288)
  const float kl_no = ratio - lrno - 1.0f;         // (:289)
  const int kl_inside = (lrrn >= -C) && (lrrn <= C);

  float pol, dpol, ratio_stat, clamp_no;
  if (c.policy_loss == PRL_POLICY_PPO) {
    const float s1 = ratio * A;                                  // (:299)
    const float cr = prl_clampf(ratio, c.clip_lo, c.clip_hi);    // (:300)
    const float s2 = cr * A;                                     // (:302)
    clamp_no = (cr != ratio) ? 1.0f : 0.0f;                      // (:301)
    const int inside = (ratio >= c.clip_lo) && (ratio <= c.clip_hi);
    const float d1 = ratio * A;                     // d surr1 / d nlp
    const float d2 = inside ? (ratio * A) : 0.0f;   // d surr2 / d nlp
    if (s1 < s2) {
      pol = s1;
      dpol = d1;
    } else if (s2 < s1) {
      pol = s2;
      dpol = d2;
    } else if (s1 == s2) {
      pol = s1;
      dpol = 0.5f * d1 + 0.5f * d2;
    } else {  // NaN involved: torch.min propagates NaN
      pol = s1 + s2;
      dpol = pol;
    }
    ratio_stat = ratio;
  } else if (c.policy_loss == PRL_POLICY_REINFORCE) {  // (:304-309)
    clamp_no = (ratio > c.clip_hi) ? 1.0f : 0.0f;
    const float crr = prl_clampf(ratio, 0.0f, c.clip_hi);
    pol = x.nlp * A * crr;
    dpol = A * crr;  // ratio is detached
    ratio_stat = crr;
  } else {  // GSPO (:310-352): the loss lives at segment level; tokens only feed the statistics
    clamp_no = 0.0f;
    pol = 0.0f;
    dpol = 0.0f;
    ratio_stat = ratio;
  }

  // combine (:355-359): loss = policy_loss - kl_coef*approx_kl [+ ent_coef*entropy]
  float loss_t = pol - c.kl_coef * kl;
  if (c.use_entropy_loss) loss_t = loss_t + c.entropy_coef * x.ent;
  const float v = loss_t * w;  // (:363), mask == 1 here

  const int gspo = (c.policy_loss == PRL_POLICY_GSPO);
  const int finite = prl_isfinite(v) && !gspo;
  o.contrib = gspo ? 0.0f : prl_nan_to_num0(v);
  // d kl / d nlp = (1 - exp(cl)) inside the clamp range, else 0
  const float dkl = kl_inside ? (1.0f - ecl) : 0.0f;
  const float dl = dpol - c.kl_coef * dkl;
  o.g_nlp = finite ? -(dl * w) : 0.0f;
  o.g_ent = (finite && c.use_entropy_loss) ? -(c.entropy_coef * w) : 0.0f;

  o.w = w;
  o.A = A;
  o.ratio_stat = ratio_stat;
  o.kl = kl;
  o.kl_new_old = kl_no;
  o.abs_lrno = fabsf(lrno);
  o.exp_lrrn = kl_inside ? ecl : PRL_STAT_EXPF(lrrn);  // inside the clamp range cl == lrrn
  o.exp_ref_old = PRL_STAT_EXPF(x.ref_lp - x.old_lp);
  o.clamp_rn = (fabsf(lrrn) > C) ? 1.0f : 0.0f;
  o.clamp_no = clamp_no;
  o.bad_nlp = !prl_isfinite(x.nlp);
  o.bad_lrrn = !prl_isfinite(lrrn);
  o.bad_kl = !prl_isfinite(kl);
}

// Gradient-only evaluation (fused logits kernel): returns d loss / d nlp and
// d loss / d entropy for one token; zero when the token is masked out.
PRL_HD void prl_token_grad(const prl_loss_config& c, const PrlTokenIn& x, int masked_in,
                           float* g_nlp, float* g_ent) {
  if (!masked_in) {
    *g_nlp = 0.0f;
    *g_ent = 0.0f;
    return;
  }
  PrlTokenOut o;
  prl_token_eval(c, x, o);
  *g_nlp = o.g_nlp;
  *g_ent = o.g_ent;
}
