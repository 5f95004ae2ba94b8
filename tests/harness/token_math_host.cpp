/* Host build of the DEVICE token math (pipelinerl_amd/csrc/prl_token_math.h) so that the exact
 * per-token formulas the HIP kernels execute can be checked against the oracle on a machine
 * without a GPU.  Test-only: nothing in the product links or calls this. */
#include "../../pipelinerl_amd/csrc/prl_token_math.h"

/* evaluates n masked tokens; out_* arrays have length n */
extern "C" void prl_host_token_eval(const prl_loss_config* cfg, long n, const float* nlp, const float* ent,
                         const float* old_lp, const float* ref_lp, const float* adv, const float* reward,
                         const float* group_tokens, const float* num_labels, const float* overflow,
                         float* contrib, float* g_nlp, float* g_ent, float* ratio_stat, float* kl,
                         float* clamp_no) {
  for (long i = 0; i < n; ++i) {
    PrlTokenIn x = {nlp[i], ent[i], old_lp[i], ref_lp[i], adv[i], reward[i], group_tokens[i], num_labels[i], overflow[i]};
    PrlTokenOut o;
    prl_token_eval(*cfg, x, o);
    contrib[i] = o.contrib;
    g_nlp[i] = o.g_nlp;
    g_ent[i] = o.g_ent;
    ratio_stat[i] = o.ratio_stat;
    kl[i] = o.kl;
    clamp_no[i] = o.clamp_no;
  }
}
