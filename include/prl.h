/*
 * prl.h — C ABI of libprl.so: the MI355X (gfx950) implementation of PipelineRL's
 * rollout -> preprocess -> finetune hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, and returns an
 * int status (PRL_OK == 0, negative errno-style codes otherwise).  After a non-zero
 * return `prl_last_error()` gives a thread-local human readable message.
 *
 * Conventions
 *   - "device" pointers are HIP device pointers owned by the caller (PyTorch-ROCm
 *     tensors in the Python host).  The library never allocates device memory:
 *     scratch space is passed in (`workspace`, sized by the *_workspace_bytes call).
 *   - `stream` is a hipStream_t (as void*).  All work is enqueued on it; nothing
 *     synchronises unless stated.
 *   - batch tensors use the reference's dtypes/layout (PipelineBatchEncoding,
 *     reference pipelinerl/finetune/types.py:46-75): int64 [rows, cols] and
 *     float32 [rows, cols], row-major contiguous.  Packed batches are rows == 1.
 *   - "token-aligned" per-token outputs (new_logprobs, entropy, their grads) live
 *     on the UNSHIFTED axis: element [r, c] is the quantity for *predicting token
 *     c of row r* (computed from logits[r, c-1]); column 0 is unused and written
 *     as 0.  The reference's shifted tensor x[:, :-1] therefore equals ours[:, 1:].
 *
 * Reference interfaces replaced (file:line in ServiceNow/PipelineRL):
 *   prl_logprob_entropy_fwd/bwd   pipelinerl/finetune/rl/__init__.py:207-233
 *   prl_grpo_loss_fwd_bwd         pipelinerl/finetune/rl/__init__.py:238-439,
 *                                 pipelinerl/finetune/rl/utils.py:26-92
 *   prl_fused_logits_loss         the two above in one pass (no reference analogue)
 *   prl_segment_sums              pipelinerl/finetune/rl/utils.py:106-208
 *   prl_gspo_segment_sums / prl_gspo_segment_terms / prl_gspo_expand
 *                                 pipelinerl/finetune/rl/__init__.py:310-352 (the per-token ends of the sequence-level term)
 *   prl_value_head_fwd_bwd        pipelinerl/finetune/rl/__init__.py:265-272, 367-381, 441-448
 *                                 (models of pipelinerl/finetune/value_model.py)
 *   prl_seq_scan / prl_group_advantages
 *                                 pipelinerl/finetune/rl/__init__.py:453-570
 *   prl_patch_oov                 pipelinerl/preprocess.py:107-141
 *   prl_pack_collate              pipelinerl/finetune/data.py:215-283
 *                                 (+ rl/__init__.py:573-594 field expansion)
 *   prl_pad_collate               pipelinerl/finetune/data.py:163-212
 *   prl_lm_head_*                 pipelinerl/finetune/rl/__init__.py:204-233 (model forward's output
 *                                 head + K1) with the numerics of
 *                                 pipelinerl/finetune/checkpoints.py:87-103 (fp32 head)
 *   prl_ring_*                    pipelinerl/shared_memory_array.py:9-196
 *   prl_log_*                     pipelinerl/streams.py:120-192, 249-346
 *   prl_publisher_*               pipelinerl/preprocess.py:356-367, 629-648
 *   prl_wsync_* / prl_ipc_* /
 *   prl_bucket_*                  pipelinerl/finetune_loop.py:205-292,
 *                                 pipelinerl/vllm1.py:62-134,
 *                                 pipelinerl/torch_utils.py:70-94
 */
#ifndef PRL_H_
#define PRL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRL_ABI_VERSION 13

#define PRL_OK 0
#define PRL_EINVAL (-22)   /* bad argument                                  */
#define PRL_ENOMEM (-12)   /* workspace too small                           */
#define PRL_EFAULT (-14)   /* HIP / RCCL runtime error (see prl_last_error) */
#define PRL_EAGAIN (-11)   /* ring full / empty in non-blocking mode        */
#define PRL_ETIMEDOUT (-110)
#define PRL_ENOSYS (-38)   /* optional dependency (RCCL) not loadable       */
#define PRL_EMSGSIZE (-90) /* ring record larger than a slot                */

typedef void* prl_stream_t; /* hipStream_t */

int prl_abi_version(void);

/* Launch overrides a caller may need (a shape the default choice handles badly, a split count pinned for bit-reproducibility
 * across GPUs with a different CU count, a test that forces a fallback shape).  A process-wide table of integers read by the
 * launch code - there is NO getenv() on any launch path; the Python host maps its PRL_* environment variables onto this call.
 * `value` PRL_TUNE_UNSET restores the built-in choice.  No reference counterpart.  (The measurement-only keys of earlier ABI
 * versions - schedule A/B bits, raster groups, tokens per lane - are gone with the variants they selected.) */
enum prl_tune_key {
  PRL_TUNE_FUSED_VARIANT = 0,  /* prl_fused_logits_loss: 0 two-sweep 256 threads | 4 two-sweep 2 x 512 (bf16 default) |
                                  6 two-sweep 1024 (fallback of short / unaligned rows) | 21 row-resident (fp32 default) */
  PRL_TUNE_LMHEAD_TILE = 1,    /* fused head workgroup tile: 128 (128 x 128) | 256 (256 x 128, ring of 3) | 512 (256 x 256) */
  PRL_TUNE_LMHEAD_NSPLIT = 2,  /* vocabulary splits of the fused head's forward (summation order of the soft-max partials) */
  PRL_TUNE_LMHEAD_KSPLIT = 3,  /* split-K factor of the d hidden product (summation order of its partial sums) */
  PRL_TUNE_COUNT = 4
};
#define PRL_TUNE_UNSET INT64_MIN
int prl_set_tuning(int32_t key, int64_t value);
int prl_get_tuning(int32_t key, int64_t* value);
const char* prl_last_error(void);

/* ------------------------------------------------------------------------- */
/* K1 / K1e : logits -> (new_logprobs, entropy)                              */
/* ------------------------------------------------------------------------- */

#define PRL_DTYPE_F32 0
#define PRL_DTYPE_BF16 1

/*
 * Forward.  logits: [rows*cols, vocab] (row stride `logits_row_stride` elements,
 * dtype f32 or bf16), input_ids: int64 [rows, cols].
 *   z            = logits / temperature                       (rl/__init__.py:207-208)
 *   new_logprobs[r,c] = z[r,c-1, ids[r,c]] - logsumexp(z[r,c-1,:])      (:209-212)
 *   entropy[r,c]      = -sum_v softmax(z)_v log_softmax(z)_v             (:215-233)
 *   lse2[r,c]    = logsumexp in base-2 units of z*log2(e)   (saved for backward)
 * All three outputs are float32 [rows, cols] token-aligned (col 0 := 0).
 * An out-of-range id yields NaN (the host turns that into the reference's assert).
 */
int prl_logprob_entropy_fwd(int64_t rows, int64_t cols, int64_t vocab,
                            const void* logits, int32_t logits_dtype,
                            int64_t logits_row_stride, const int64_t* input_ids,
                            float temperature, float* new_logprobs, float* entropy,
                            float* lse2, prl_stream_t stream);

/*
 * Backward.  grad_logits[r,c-1,v] = scale * ( g*(1[v==id] - p_v)
 *                                           - gH * p_v*(log p_v + H) ) / temperature
 * with g = grad_new_logprobs[r,c], gH = grad_entropy[r,c] (nullable => 0),
 * scale = *upstream (device scalar, nullable => 1).  Row (r, cols-1) gets zeros.
 * grad_logits may alias logits (in-place); same dtype/stride as logits.
 */
int prl_logprob_entropy_bwd(int64_t rows, int64_t cols, int64_t vocab,
                            const void* logits, int32_t logits_dtype,
                            int64_t logits_row_stride, const int64_t* input_ids,
                            float temperature, const float* lse2, const float* entropy,
                            const float* grad_new_logprobs, const float* grad_entropy,
                            const float* upstream, void* grad_logits,
                            prl_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K2 + K3 : GRPO/PPO/REINFORCE token loss + masked reduce + stats + grad    */
/* ------------------------------------------------------------------------- */

#define PRL_POLICY_PPO 0
#define PRL_POLICY_REINFORCE 1
#define PRL_POLICY_GSPO 2 /* sequence-level ratio: the per-token policy gradient and clip indicator
                              come from the caller (ext_token_grad / ext_clamp_indicator); the
                              kernel produces the statistics and routes the gradient           */

typedef struct prl_loss_config {
  int32_t policy_loss;         /* PRL_POLICY_*                      (RLConfig.policy_loss) */
  int32_t use_advantages;      /* rl/__init__.py:274                                       */
  int32_t relu_log_p_weights;  /* :275-276                                                 */
  int32_t group_normalization; /* :245-250                                                 */
  int32_t overlong_filtering;  /* :252-255                                                 */
  int32_t use_entropy_loss;    /* entropy_bonus != 0 or final_entropy_bonus != 0 (:215)    */
  int32_t flat_micro_batches;  /* 1: the [1, T] batch is several packed micro-batches laid
                                  back to back (one launch per optimizer step); a position
                                  with position_ids == 0 then starts a micro-batch or a
                                  masked sequence and carries no prediction                */
  int32_t skip_unlabelled;     /* prl_fused_logits_loss(_planes) only.  1: a row whose next token has
                                  labels == -100 (prompt / observation tokens, sequence starts, padding)
                                  is not read at all: its new_logprobs / entropy / lse2 are written as 0 and
                                  its gradient row as zeros.  Every term of the loss and every statistic
                                  carries the label mask (rl/__init__.py:238-250), so loss, statistics and
                                  gradients do not change.  0: the outputs of K1 at every position        */
  float token_weight;          /* fp32(1)/fp32(batch_size)                       (:250)    */
  float clip_lo;               /* fp32(1 - epsilon_low)                          (:300)    */
  float clip_hi;               /* fp32(1 + epsilon_high)                         (:300,306)*/
  float kl_coef;               /* linear_decay_coef(...) for this step           (:293)    */
  float entropy_coef;          /* linear_decay_coef(...) for this step           (:292)    */
  float clamp_log_ratio_ref_new; /* :280-286                                               */
  float upstream_scale;        /* prl_fused_logits_loss only: the d(objective)/d(loss) factor the
                                  caller will back-propagate (accelerate's 1/accumulation, a loss
                                  scaler), folded into d logits at no cost; 0 means 1.  The kernel
                                  runs before autograd knows the factor, so the host states what
                                  it expects and prl_scale_unless repairs a wrong guess ON DEVICE */
} prl_loss_config;

/* stats vector (double[PRL_NUM_STATS], device).  Sums follow App. A of SURVEY.md:
 * "mean" stats are sum over masked tokens of x / num_labels. */
enum {
  PRL_STAT_LOSS = 0,
  PRL_STAT_NUM_OUTPUT_TOKENS = 1,
  PRL_STAT_NUM_SEQUENCES = 2,
  PRL_STAT_REWARD = 3,
  PRL_STAT_MAX_REWARD = 4,
  PRL_STAT_MIN_REWARD = 5,
  PRL_STAT_ENTROPY = 6,
  PRL_STAT_OLD_LOGPROBS = 7,
  PRL_STAT_NEW_LOGPROBS = 8,
  PRL_STAT_REF_LOGPROBS = 9,
  PRL_STAT_ADVANTAGE = 10,
  PRL_STAT_MAX_ADVANTAGE = 11,
  PRL_STAT_MIN_ADVANTAGE = 12,
  PRL_STAT_KL = 13,
  PRL_STAT_KL_NEW_OLD = 14,
  PRL_STAT_MEAN_ABS_LOG_RATIO_NEW_OLD = 15,
  PRL_STAT_MAX_KL = 16,
  PRL_STAT_MIN_KL = 17,
  PRL_STAT_RATIO_NEW_OLD = 18,
  PRL_STAT_RATIO_NEW_OLD_SUM = 19,
  PRL_STAT_RATIO_NEW_OLD_SQUARED_SUM = 20,
  PRL_STAT_RATIO_REF_NEW = 21,
  PRL_STAT_RATIO_REF_OLD = 22,
  PRL_STAT_CLAMP_REF_NEW_INDICATOR = 23,
  PRL_STAT_CLAMP_NEW_OLD_INDICATOR = 24,
  PRL_STAT_TOKEN_WEIGHT = 25,
  PRL_STAT_MAX_TOKEN_WEIGHT = 26,
  PRL_STAT_MIN_TOKEN_WEIGHT = 27,
  PRL_STAT_NONFINITE_NEW_LOGPROBS = 28,   /* count; reference assert :213 */
  PRL_STAT_NONFINITE_LOG_RATIO_REF_NEW = 29, /* :263 */
  PRL_STAT_NONFINITE_KL = 30,             /* :291 */
  PRL_STAT_BAD_GROUP_TOKENS = 31,         /* count of group_tokens <= 0 (:247) */
  PRL_NUM_STATS = 32
};

int prl_grpo_loss_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes);

/*
 * labels, position_ids: int64 [rows, cols] (position_ids nullable when rows > 1 or
 * the batch is not packed: num_sequences := rows).  new_logprobs, entropy:
 * token-aligned float32 [rows, cols].  The seven RL columns are the batch's
 * float32 [rows, cols] tensors, unshifted.
 * Outputs: grad_new_logprobs / grad_entropy (token-aligned d loss / d x, nullable),
 * stats (double[PRL_NUM_STATS]).  loss = stats[PRL_STAT_LOSS], also written as a
 * float to loss_out (device, nullable).
 * ext_token_grad / ext_clamp_indicator: token-aligned float32 [rows, cols], required for
 * PRL_POLICY_GSPO (rl/__init__.py:310-352: the gradient coefficient and clip indicator of a
 * token's segment), NULL otherwise.
 */
int prl_grpo_loss_fwd_bwd(const prl_loss_config* cfg, int64_t rows, int64_t cols,
                          const int64_t* labels, const int64_t* position_ids,
                          const float* new_logprobs, const float* entropy,
                          const float* old_logprobs, const float* ref_logprobs,
                          const float* advantages, const float* rewards,
                          const float* group_tokens, const float* num_labels,
                          const float* overflow, const float* ext_token_grad,
                          const float* ext_clamp_indicator, float* grad_new_logprobs,
                          float* grad_entropy, float* loss_out, double* stats,
                          void* workspace, size_t workspace_bytes,
                          prl_stream_t stream);

/*
 * Fused K1 + K2 gradient + K1 backward in ONE pass over the logits (no reference
 * analogue: the reference materialises logits/temperature, log-softmax, and lets
 * autograd re-read them).  Per logits row: online-softmax pass -> per-token loss
 * gradient -> second pass writes d loss / d logits (grad_logits may alias logits).
 * Also writes token-aligned new_logprobs / entropy / lse2 so that
 * prl_grpo_loss_fwd_bwd can produce loss and stats from them.
 */
int prl_fused_logits_loss(const prl_loss_config* cfg, int64_t rows, int64_t cols,
                          int64_t vocab, const void* logits, int32_t logits_dtype,
                          int64_t logits_row_stride, float temperature,
                          const int64_t* input_ids, const int64_t* labels,
                          const float* old_logprobs, const float* ref_logprobs,
                          const float* advantages, const float* rewards,
                          const float* group_tokens, const float* overflow,
                          float* new_logprobs, float* entropy, float* lse2,
                          void* grad_logits, prl_stream_t stream);

/* Name of the kernel the last prl_fused_logits_loss call of this thread launched (the dispatch
 * picks a row-resident or a two-sweep kernel from vocab size, dtype and alignment); test and
 * profiling aid. */
const char* prl_last_fused_kernel(void);

/*
 * In-place `data *= *upstream / expected` over n elements of dtype f32/bf16 - unless the device
 * scalar *upstream already equals `expected`, in which case every workgroup returns after one
 * 4-byte load.  Backward of the fused path: the gradient was produced in the forward launch with
 * the expected upstream factor folded in (prl_loss_config.upstream_scale); this call makes it
 * right for any other factor without a host synchronisation (the reference multiplies through
 * autograd, finetune_loop.py:784-790).
 */
int prl_scale_unless(void* data, int64_t n, int32_t dtype, const float* upstream,
                     float expected, prl_stream_t stream);

/*
 * GSPO helper (rl/utils.py:106-208): per-segment masked sums of a and b plus token
 * counts.  segment_ids int64 [1, cols], NON-DECREASING from column 1 on (a packed batch, or a
 * sequence-parallel slice of one); mask = labels != -100; a, b token-aligned float32.
 * Outputs float64 [n_segments] x3, every entry written by the call, reduced in a fixed order
 * (bitwise reproducible).  Unsorted or out-of-range ids yield NaN in all outputs.
 */
int prl_segment_sums(int64_t cols, int32_t n_segments, const int64_t* segment_ids,
                     const int64_t* labels, const float* a, const float* b,
                     double* a_sum, double* b_sum, double* count,
                     prl_stream_t stream);

/*
 * GSPO (sequence-level policy term, rl/__init__.py:310-352 + rl/utils.py:106-208) on one packed micro-batch or one
 * sequence-parallel slice of it, the two per-token ends of it:
 *   prl_gspo_segment_sums  the four per-segment masked sums in ONE pass, sums = float64 [4, n_segments] in the order
 *       sum(new_logprobs - old_logprobs) [fp32 difference], sum(advantages), token count, sum(token weight) - the weight as in
 *       prl_grpo_loss_fwd_bwd (cfg->group_normalization, token_weight, overlong_filtering).  Columns float32 [1, cols],
 *       token-aligned; segment_ids / labels as for prl_segment_sums (non-decreasing ids; unsorted or out-of-range ids -> NaN).
 *       Fixed-order reduction, bitwise reproducible.
 *   prl_gspo_segment_terms the O(#segments) arithmetic between the two (rl/__init__.py:316-343) in one launch: from `sums` (after the
 *       sequence-parallel all-reduce, if any) the clipped sequence ratio, coef[s] = d loss / d new_logprobs of every token of segment s
 *       (times grad_scale: the SP group size, as a differentiable all-reduce would give), the clip indicator, and
 *       loss = -sum_s min(r_s A_s, clip(r_s) A_s) * (segment's token-weight sum) over valid segments; zero_out != 0 (sentinel batch,
 *       a single column): loss 0 and coef 0.
 *   prl_gspo_expand        per-segment float32 [n_segments] -> per-token float32 [1, cols]: token_grad[u] = coef[segment_ids[u]]
 *       and token_indicator[u] = indicator[segment_ids[u] - segment_ids[0]] (the j-th sequence starting or continuing in the
 *       slice takes the value of global segment j, rl/__init__.py:347-350), indices clamped to [0, n_segments).
 */
int prl_gspo_segment_sums(const prl_loss_config* cfg, int64_t cols, int32_t n_segments, const int64_t* segment_ids,
                          const int64_t* labels, const float* new_logprobs, const float* old_logprobs,
                          const float* advantages, const float* group_tokens, const float* overflow,
                          double* sums, prl_stream_t stream);
int prl_gspo_segment_terms(const prl_loss_config* cfg, int32_t n_segments, const double* sums, float grad_scale,
                           int32_t zero_out, float* coef, float* indicator, float* loss, prl_stream_t stream);
int prl_gspo_expand(int64_t cols, int32_t n_segments, const int64_t* segment_ids, const float* coef,
                    const float* indicator, float* token_grad, float* token_indicator, prl_stream_t stream);

/*
 * Value-head (actor-critic) branch of rl_step (rl/__init__.py:162, 265-272, 367-381, 441-448) for models
 * of finetune/value_model.py, whose forward also returns outputs.value [rows, cols].
 * values: float32 or bfloat16 [rows, cols] (values_dtype = PRL_DTYPE_*), values[r, c] is paired with the
 * target at column c + 1; the last column has no target.  The four RL columns are the batch's float32
 * [rows, cols] tensors, unshifted; token weights as in prl_grpo_loss_fwd_bwd (cfg->group_normalization,
 * token_weight, overlong_filtering).
 * Outputs:
 *   advantages_out float32 [rows, cols], unshifted like batch.advantages: [r, c] = rewards[r, c] -
 *     values[r, c - 1] at EVERY column >= 1, 0 at column 0.  Passed to prl_grpo_loss_fwd_bwd /
 *     prl_fused_logits_loss in place of the batch's advantages column.
 *   grad_values float32 [rows, cols] (nullable) = d value_loss / d values (0 where unlabelled, non-finite,
 *     or in the last column); the caller scales it by value_loss_coef x the upstream gradient.
 *   stats double[PRL_NUM_VALUE_STATS] in the reference's key order; value_loss_out (device float,
 *     nullable) = (float) stats[PRL_VSTAT_VALUE_LOSS] = sum over labelled tokens of
 *     nan_to_num(0.5 (V - reward)^2 w).  final loss = policy loss + value_loss_coef * value_loss.
 * Reductions are fixed-order fp64 (bitwise reproducible); no host sync.
 */
enum prl_value_stat_index {
  PRL_VSTAT_VALUE_MEAN = 0, /* sum V / num_labels over labelled tokens */
  PRL_VSTAT_VALUE_MAX = 1,  /* over labelled tokens; 0 if there are none */
  PRL_VSTAT_VALUE_MIN = 2,
  PRL_VSTAT_VALUE_LOSS = 3,
  PRL_VSTAT_VALUE_MSE = 4,  /* sum (V - reward)^2 / num_labels */
  PRL_NUM_VALUE_STATS = 5
};

int prl_value_head_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes);

int prl_value_head_fwd_bwd(const prl_loss_config* cfg, int64_t rows, int64_t cols,
                           const int64_t* labels, const void* values, int values_dtype,
                           const float* rewards, const float* group_tokens,
                           const float* num_labels, const float* overflow,
                           float* advantages_out, float* grad_values, float* value_loss_out,
                           double* stats, void* workspace, size_t workspace_bytes,
                           prl_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K5 : group-baseline advantages (populate_rl_data)                         */
/* ------------------------------------------------------------------------- */

#define PRL_FINISH_NONE 0    /* no usable finish_reason          (rl/__init__.py:543-552) */
#define PRL_FINISH_LENGTH 1  /* "length"  -> overflow 1                                     */
#define PRL_FINISH_STOP 2    /* "stop" / "content_filter" -> overflow 0                     */

/*
 * Per-sequence scan of the ragged token buffers: num_labels[s] = #labels != -100,
 * overflow[s] per the finish_reason / finished / EOS-presence rule.
 * tokens/labels: int32 ragged, seq_off int64 [n_seqs+1].
 */
int prl_seq_scan(int32_t n_seqs, const int32_t* tokens, const int32_t* labels,
                 const int64_t* seq_off, const uint8_t* finish_code,
                 const uint8_t* finished, int32_t eos_token_id, float* num_labels,
                 float* overflow, prl_stream_t stream);

/*
 * Out-of-vocabulary patch (preprocess.py:107-141 `replace_oov_tokens_with_the`): every token id
 * outside [0, table_size) or with valid[id] == 0 becomes `the_token_id`, in place, on the ragged
 * int32 token buffer; labels are untouched like in the reference.  *patched (device u64, nullable,
 * zeroed by the caller) receives the number of replaced tokens.
 */
int prl_patch_oov(int64_t n_tokens, int32_t* tokens, const uint8_t* valid, int32_t table_size,
                  int32_t the_token_id, uint64_t* patched, prl_stream_t stream);

/*
 * Leave-one-out advantages per (group_id, step_index) key and mean rollout tokens
 * per group, float64 arithmetic like the reference's pandas path.
 *   key_off [n_keys+1], key_members [n_seqs]: CSR of sequences per key, members in
 *       dataset order;  group_off/group_members likewise per group_id;
 *   group_n_rollouts [n_groups]: number of distinct rollout_index per group.
 * Outputs per sequence: advantage (f64 + f32 copy), group_tokens (f64 + f32 copy).
 */
int prl_group_advantages(int32_t n_seqs, int32_t n_keys, int32_t n_groups,
                         const int32_t* key_off, const int32_t* key_members,
                         const int32_t* group_off, const int32_t* group_members,
                         const int32_t* group_n_rollouts, const double* reward,
                         const int64_t* seq_off, int32_t divide_by_std,
                         double* advantage64, double* group_tokens64,
                         float* advantage32, float* group_tokens32,
                         prl_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K6 / K7 : pack / pad collate into PipelineBatchEncoding layout            */
/* ------------------------------------------------------------------------- */

/*
 * Ragged sources (device): tokens/labels int32 [seq_off[n]], logprobs/ref_logprobs
 * float32 over completion tokens only [lp_off[n]] (right-aligned inside each
 * sequence, zero on the left, rl/__init__.py:587-588), per-sequence float32
 * reward/advantage/group_tokens/num_labels/overflow.  `per_token_columns` is a bit
 * mask (bit 0 reward, 1 advantage, 2 group_tokens, 3 num_labels, 4 overflow): a set
 * bit means that column is instead a per-token ragged array aligned with `tokens`
 * (the generic layout of the reference's example dicts).
 *
 * Packing plan (device, built on the host, O(#sequences)):
 *   pk_src [m]    source sequence index, or -1 for a sentinel filler sequence
 *                 (tokens = eos, labels = -100, group_tokens = num_labels = 1)
 *   pk_dst [m+1]  destination offsets into the flat output (cumulative)
 *   pk_seg [m]    segment id = index of the sequence inside its micro-batch
 * Consecutive micro-batches are laid out back to back in the outputs; a sequence
 * with pk_seg > 0 gets labels[first token] = -100 (data.py:264-265).
 * Outputs (flat, length pk_dst[m]): input_ids, labels, attention_mask,
 * position_ids, segment_ids (int64) and the seven float32 RL columns.
 */
int prl_pack_collate(int32_t m, int64_t total_tokens, const int32_t* pk_src,
                     const int64_t* pk_dst, const int32_t* pk_seg,
                     const int32_t* tokens, const int32_t* labels,
                     const float* logprobs, const float* ref_logprobs,
                     const int64_t* seq_off, const int64_t* lp_off,
                     const float* reward, const float* advantage,
                     const float* group_tokens, const float* num_labels,
                     const float* overflow, int32_t per_token_columns,
                     int32_t eos_token_id,
                     int64_t* out_input_ids, int64_t* out_labels,
                     int64_t* out_attention_mask, int64_t* out_position_ids,
                     int64_t* out_segment_ids, float* out_rewards,
                     float* out_advantages, float* out_ref_logprobs,
                     float* out_old_logprobs, float* out_group_tokens,
                     float* out_num_labels, float* out_overflow,
                     prl_stream_t stream);

/*
 * Padded collate: outputs [n_rows, padded_len]; row i holds sequence row_src[i]
 * right- or left-padded (pad_left != 0).  Pad values: labels -100, everything else
 * 0 (data.py:195-205).  No position/segment ids in this mode.
 */
int prl_pad_collate(int32_t n_rows, int64_t padded_len, int32_t pad_left,
                    const int32_t* row_src, const int32_t* tokens,
                    const int32_t* labels, const float* logprobs,
                    const float* ref_logprobs, const int64_t* seq_off,
                    const int64_t* lp_off, const float* reward,
                    const float* advantage, const float* group_tokens,
                    const float* num_labels, const float* overflow,
                    int32_t per_token_columns,
                    int64_t* out_input_ids, int64_t* out_labels,
                    int64_t* out_attention_mask, float* out_rewards,
                    float* out_advantages, float* out_ref_logprobs,
                    float* out_old_logprobs, float* out_group_tokens,
                    float* out_num_labels, float* out_overflow,
                    prl_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Transport: shared-memory record ring (host side, no GPU)                  */
/* ------------------------------------------------------------------------- */

typedef struct prl_ring prl_ring;

/* Create (unlinking any stale object of that name) or attach to a POSIX shm ring
 * of `n_slots` slots of `slot_bytes` payload bytes.  Multi-producer /
 * multi-consumer, FIFO by claim order. */
int prl_ring_create(const char* name, uint32_t n_slots, uint64_t slot_bytes,
                    prl_ring** out);
int prl_ring_attach(const char* name, prl_ring** out);
/* timeout_ms < 0: block forever; == 0: non-blocking (PRL_EAGAIN). */
int prl_ring_put(prl_ring* r, const void* data, uint64_t nbytes, int64_t timeout_ms);
/* Copies the next record into buf (cap bytes) and stores its size in *nbytes. */
int prl_ring_get(prl_ring* r, void* buf, uint64_t cap, uint64_t* nbytes,
                 int64_t timeout_ms);
/* Zero-copy variants: reserve/commit a slot for writing, acquire/release for reading. */
int prl_ring_reserve(prl_ring* r, void** slot_ptr, uint64_t* ticket, int64_t timeout_ms);
int prl_ring_commit(prl_ring* r, uint64_t ticket, uint64_t nbytes);
int prl_ring_acquire(prl_ring* r, const void** slot_ptr, uint64_t* nbytes,
                     uint64_t* ticket, int64_t timeout_ms);
int prl_ring_release(prl_ring* r, uint64_t ticket);
int prl_ring_size(prl_ring* r, uint64_t* n_ready);
int prl_ring_capacity(prl_ring* r, uint32_t* n_slots, uint64_t* slot_bytes);
int prl_ring_max_record_bytes(prl_ring* r, uint64_t* nbytes);
int prl_ring_close(prl_ring* r);           /* detach (creator also unlinks) */
int prl_ring_detach(prl_ring* r);          /* detach only: the segment stays for late readers */
int prl_ring_unlink(const char* name);

/* ------------------------------------------------------------------------- */
/* Transport: shared-memory record LOG (host side, no GPU)                    */
/* ------------------------------------------------------------------------- */

/*
 * Append-only log with per-reader cursors - the semantics of the reference's stream backends
 * (Redis XREAD from id 0, pipelinerl/streams.py:120-192; a JSONL file tailed from offset 0,
 * :281-346): every reader sees every record from the first one, any number of readers, the
 * writer never waits for a reader, a writer that is closed and reopened appends to the same
 * stream (finetune_loop.py:244), readers block without polling.  Segmented POSIX shared memory:
 * /<name> (control block) + /<name>.<k> (segments, grown on demand).
 */
typedef struct prl_log prl_log;

#define PRL_LOG_CREATE 1    /* create the log if it does not exist (attach otherwise)            */
#define PRL_LOG_TRUNCATE 2  /* remove an existing log of that name first (mode "w")              */
#define PRL_LOG_READER 4    /* register a read cursor (lets a trimming writer see this reader)   */
#define PRL_LOG_TRIM 8      /* at creation: unlink segments every registered reader has left     */

/* PRL_EAGAIN: the log is being created by another process right now (retry); PRL_EFAULT: absent. */
int prl_log_open(const char* name, uint64_t segment_bytes, int32_t flags, prl_log** out);
int prl_log_append(prl_log* l, const void* data, uint64_t nbytes);
/* One record gathered from several source ranges - the header and the columns of a PipelineBatchEncoding
 * (pipelinerl/streams.py:262-270 serialises them into one JSON line) - copied straight into the segment, no
 * intermediate record buffer.  Pieces in ascending, non-overlapping `offset` order inside a record of `nbytes`;
 * bytes no piece covers (alignment gaps) are zero. */
typedef struct prl_log_iov {
  const void* ptr;
  uint64_t offset;
  uint64_t nbytes;
} prl_log_iov;
int prl_log_appendv(prl_log* l, const prl_log_iov* iov, int32_t n_iov, uint64_t nbytes);
/* Next record of this handle's cursor (starts at the first retained record): *ptr points INTO
 * the mapping, valid until the next read / close.  timeout_ms < 0 blocks, 0 returns PRL_EAGAIN
 * at the tail, > 0 PRL_ETIMEDOUT. */
int prl_log_read(prl_log* l, const void** ptr, uint64_t* nbytes, int64_t timeout_ms);
int prl_log_stats(prl_log* l, uint64_t* n_records, uint64_t* n_bytes,
                  uint64_t* first_segment, uint64_t* n_segments);
int prl_log_close(prl_log* l);
int prl_log_unlink(const char* name); /* remove the control block and every segment */

/*
 * Publisher: the last leg of the preprocessing loop - `write_micro_batch_slices` at pipelinerl/preprocess.py:356-367, called
 * inline from the loop at :629-648 - on a native worker thread.  A JOB is one drain of the scheduler: the packed block the
 * pack kernel wrote (device memory, nullable), the event that certifies it (hipEvent_t, nullable), and the records to append,
 * in order, each to the log of its trainer partition.  A record is gathered from PIECES: ranges of the block once it has been
 * copied to page-locked host memory (PRL_PUB_FROM_BLOCK, `src` = byte offset into the block), ranges of the job's inline
 * bytes (PRL_PUB_INLINE: record headers, sentinel batches; copied at submit, the caller may free them at once) and ranges of
 * host memory the caller owns (PRL_PUB_FROM_HOST, `src` = the address: the compact `training_data` wire gathers a micro-batch's
 * ragged columns straight from the decoded `actor` records - nothing per token ever goes through the device).  submit
 * returns a ticket and blocks only while two jobs are pending; the caller must keep `dev_block`, `ready_event` and every
 * PRL_PUB_FROM_HOST range alive until prl_publisher_completed reports the ticket.  Records reach the logs in submit order.  An error of the worker (HIP, a log
 * append) is sticky: every later call returns it.
 */
typedef struct prl_publisher prl_publisher;
#define PRL_PUB_FROM_BLOCK 0
#define PRL_PUB_INLINE 1
#define PRL_PUB_FROM_HOST 2
typedef struct prl_pub_piece {
  uint64_t src;    /* byte offset into the staged block / the inline bytes; PRL_PUB_FROM_HOST: a host address */
  uint64_t offset; /* byte offset inside the record (ascending, non-overlapping, as for prl_log_appendv) */
  uint64_t nbytes;
  uint32_t kind;   /* PRL_PUB_FROM_BLOCK | PRL_PUB_INLINE | PRL_PUB_FROM_HOST */
  uint32_t _pad;
} prl_pub_piece;
typedef struct prl_pub_record {
  void* log;            /* prl_log* of the partition (a writer handle no other thread appends to meanwhile) */
  uint64_t nbytes;      /* record size */
  uint32_t first_piece; /* index into the job's piece table */
  uint32_t n_pieces;
} prl_pub_record;
int prl_publisher_create(int32_t device, prl_publisher** out);
int prl_publisher_submit(prl_publisher* p, const void* dev_block, uint64_t block_bytes, void* ready_event,
                         const prl_pub_record* recs, int32_t n_recs, const prl_pub_piece* pieces,
                         int32_t n_pieces, const void* inline_bytes, uint64_t inline_nbytes,
                         uint64_t* ticket);
int prl_publisher_completed(prl_publisher* p, uint64_t* ticket); /* highest ticket whose records are in the logs */
int prl_publisher_wait(prl_publisher* p, uint64_t ticket, int64_t timeout_ms); /* < 0: block; PRL_ETIMEDOUT */
int prl_publisher_stats(prl_publisher* p, uint64_t* busy_ns, uint64_t* copy_ns); /* worker time: whole jobs / their device -> host copies */
int prl_publisher_destroy(prl_publisher* p); /* finishes what was submitted, then joins the worker */

/* ------------------------------------------------------------------------- */
/* Weight sync: trainer -> inference workers over RCCL / xGMI                */
/* ------------------------------------------------------------------------- */

typedef struct prl_wsync prl_wsync;

#define PRL_WSYNC_UID_BYTES 128
/* rank 0 creates the RCCL unique id; the host shares it (TCP store) with peers. */
int prl_wsync_unique_id(uint8_t uid[PRL_WSYNC_UID_BYTES]);
/* Collective: every rank of the update group calls this (rank 0 = trainer). */
int prl_wsync_init(const uint8_t uid[PRL_WSYNC_UID_BYTES], int32_t rank,
                   int32_t world_size, int32_t device, prl_wsync** out);
/* Plain 1->N broadcast of a contiguous byte bucket from rank `src`. */
int prl_wsync_bcast_bucket(prl_wsync* w, void* bucket, uint64_t nbytes, int32_t src,
                           prl_stream_t stream);
/*
 * Scatter + all-gather broadcast from rank 0 to ranks 1..world-1: rank 0 sends a
 * distinct 1/(world-1) slice to each receiver over its own xGMI link, receivers
 * exchange slices among themselves.  Every rank passes the full bucket pointer.
 */
int prl_wsync_bcast_bucket_sag(prl_wsync* w, void* bucket, uint64_t nbytes,
                               prl_stream_t stream);
/* What RCCL itself reports for this communicator (ncclCommCount / ncclCommUserRank): lets a caller - bench.py's
 * N > 1 line - state the communicator size from the library instead of from its own launch arguments.  The reference
 * has no counterpart (it trusts `world_size` of stateless_init_process_group, torch_utils.py:70-94). */
int prl_wsync_comm_size(prl_wsync* w, int32_t* world_size, int32_t* rank);
int prl_wsync_destroy(prl_wsync* w);

/* Colocated hand-off (trainer and inference worker are two processes on ONE GPU, BASELINE config
 * "actor+learner colocated"): a device bucket allocated by the library is exported as a HIP IPC
 * handle; the peer maps the same memory and copies it into its own weights.  Replaces the
 * per-parameter broadcast of finetune_loop.py:279-282 / vllm1.py:118-122 for that topology. */
#define PRL_IPC_HANDLE_BYTES 64
int prl_ipc_alloc(uint64_t nbytes, void** dev_ptr);
int prl_ipc_free(void* dev_ptr);
int prl_ipc_export(const void* dev_ptr, uint8_t handle[PRL_IPC_HANDLE_BYTES]);
int prl_ipc_open(const uint8_t handle[PRL_IPC_HANDLE_BYTES], void** dev_ptr);
int prl_ipc_close(void* dev_ptr);

/* ---- bucket <-> parameter copies ---------------------------------------------------------
 * Flatten the trainer's parameters into a bucket (gather) / hand a received bucket out to the
 * engine's weights (scatter) in one launch per 64 segments, replacing the per-parameter
 * staging of finetune_loop.py:262-282 and vllm1.py:110-127.  `segments` is a HOST array;
 * `tensor` and `bucket` are device pointers; regions must not overlap.  Enqueued on `stream`. */
struct prl_segment {
  void* tensor;          /* device address of the parameter's (contiguous) storage */
  int64_t bucket_offset; /* byte offset of its slot inside the bucket */
  int64_t nbytes;
};
int prl_bucket_gather(void* bucket, int64_t bucket_bytes, const struct prl_segment* segments, int64_t n_segments, void* stream);
int prl_bucket_scatter(const void* bucket, int64_t bucket_bytes, const struct prl_segment* segments, int64_t n_segments, void* stream);

/* ------------------------------------------------------------------------- */
/* Fused output head: hidden states -> new_logprobs / entropy, logits never   */
/* written (reference rl/__init__.py:204-233 after the model's */
/* lm_head, whose fp32 numerics checkpoints.py:87-103 enforces)                */
/* ------------------------------------------------------------------------- */

/*
 * Operand preparation, once per optimizer step: weight [vocab, hidden] (f32 or bf16) ->
 *   w_hi, w_lo   [vocab, hidden] bf16 planes, weight = hi + lo to ~2^-17 relative
 *                (a bf16 weight is its own hi plane: pass w_lo = NULL and use the weight as w_hi)
 *   wt_hi, wt_lo [hidden, vocab] the same planes transposed (backward only; nullable)
 * Any output may be NULL.
 */
int prl_lm_head_prepare(int64_t vocab, int64_t hidden, const void* weight,
                        int32_t weight_dtype, uint16_t* w_hi, uint16_t* w_lo,
                        uint16_t* wt_hi, uint16_t* wt_lo, prl_stream_t stream);

/* Scratch sizes: forward (partial soft-max states per vocabulary split) and backward (the
 * d-logits planes of ONE chunk of `chunk_rows` logits rows, both layouts, + the transposed
 * hidden chunk + 8 fp32 [chunk_rows, hidden] slices for the split-K d hidden product, whose
 * slices are added in a fixed order: bitwise reproducible).  Either output pointer may be NULL. */
int prl_lm_head_workspace_bytes(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab,
                                int64_t chunk_rows, size_t* fwd_bytes, size_t* bwd_bytes);

/*
 * Forward.  hidden_bf16: [rows*cols, hidden] bf16 (the model's last hidden states), input_ids
 * int64 [rows, cols].  With logits[q, v] = sum_k hidden[q, k] * (w_hi + w_lo)[v, k] (fp32
 * accumulation on the bf16 matrix cores) the outputs are those of prl_logprob_entropy_fwd:
 * token-aligned float32 [rows, cols] new_logprobs / entropy / lse2, column 0 := 0.
 * hidden must be a multiple of 64.  No [rows*cols, vocab] buffer exists at any point.
 */
int prl_lm_head_logprob_fwd(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab,
                            const uint16_t* hidden_bf16, const uint16_t* w_hi,
                            const uint16_t* w_lo, const int64_t* input_ids,
                            float temperature, float* new_logprobs, float* entropy,
                            float* lse2, void* workspace, size_t workspace_bytes,
                            prl_stream_t stream);

#define PRL_LM_HEAD_DW_OVERWRITE 2    /* flags: grad_weight = d W (uninitialised memory is fine) instead of +=: saves the caller
                                         a zero fill and the first chunk a read of the [vocab, hidden] fp32 buffer */
#define PRL_LM_HEAD_DH_LEADING_TERM 1 /* flags: d hidden from the leading bf16 product only (d logits_hi x W_hi):
                                         2^-9 relative error, the size of a bf16 rounding - meant for bf16 grad_hidden */
#define PRL_LM_HEAD_DH_NO_WEIGHT_LO 4 /* flags: d hidden = (d logits_hi + d logits_lo) x W_hi: only the fp32 weight's low plane is
                                         dropped (2^-9 of each weight, i.e. a bf16 weight) - two products on the dual-plane core
                                         instead of three; no effect on a bf16 weight, whose d hidden is these two products */

/*
 * Backward of the above for token-aligned upstream gradients grad_new_logprobs /
 * grad_entropy (nullable) and a device scalar `upstream` (nullable => 1), exactly the
 * d logits of prl_logprob_entropy_bwd pushed through the head:
 *   grad_hidden [rows*cols, hidden] (bf16 or f32, overwritten; nullable)
 *                 = d logits (w_hi + w_lo)
 *   grad_weight [vocab, hidden] f32 (ACCUMULATED: +=, or overwritten with
 *                 PRL_LM_HEAD_DW_OVERWRITE; nullable) = d logits^T hidden
 * Works chunk by chunk over `chunk_rows` logits rows: the logits of a chunk are recomputed,
 * its d logits live as bf16 (hi, lo) planes in the workspace only.  vocab and hidden must be
 * multiples of 64.
 */
int prl_lm_head_logprob_bwd(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab,
                            const uint16_t* hidden_bf16, const uint16_t* w_hi,
                            const uint16_t* w_lo, const uint16_t* wt_hi,
                            const uint16_t* wt_lo, const int64_t* input_ids,
                            float temperature, const float* lse2, const float* entropy,
                            const float* grad_new_logprobs, const float* grad_entropy,
                            const float* upstream, void* grad_hidden,
                            int32_t grad_hidden_dtype, float* grad_weight,
                            int64_t chunk_rows, int32_t flags, void* workspace,
                            size_t workspace_bytes, prl_stream_t stream);

/*
 * The same head with the logits KEPT between forward and backward (reference lines as above: rl/__init__.py:204-233 reads the
 * [T, V] fp32 logits of checkpoints.py:87-103's lm_head, autograd keeps them for the backward).  The *_keep forwards compute
 * exactly what prl_lm_head_logprob_fwd computes and also write `logits2` [rows * cols, vocab] fp32 = the logits in
 * base-2 units (logit * log2(e) / temperature) straight from the accumulators (vocab must be a multiple of 8).
 * prl_lm_head_logprob_bwd_kept then forms the d-logits planes in one pass over them instead of recomputing the two plane
 * products: 5 products instead of 7 per micro-batch at the price of rows * cols * vocab * 4 bytes that live from the head's
 * forward to its backward (4.98 GB for 8192 x 152 064).  Workspace: prl_lm_head_workspace_bytes as for the recomputing form.
 * The row-major weight planes are not read by the backward (d hidden runs on wt_hi / wt_lo, d W on the hidden states).
 */
int prl_lm_head_logprob_fwd_keep(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab,
                                 const uint16_t* hidden_bf16, const uint16_t* w_hi, const uint16_t* w_lo,
                                 const int64_t* input_ids, float temperature, float* new_logprobs,
                                 float* entropy, float* lse2, float* logits2, void* workspace,
                                 size_t workspace_bytes, prl_stream_t stream);
int prl_lm_head_logprob_bwd_kept(int64_t rows, int64_t cols, int64_t hidden, int64_t vocab,
                                 const uint16_t* hidden_bf16, const float* logits2, const uint16_t* wt_hi,
                                 const uint16_t* wt_lo, const int64_t* input_ids, float temperature,
                                 const float* lse2, const float* entropy, const float* grad_new_logprobs,
                                 const float* grad_entropy, const float* upstream, void* grad_hidden,
                                 int32_t grad_hidden_dtype, float* grad_weight, int64_t chunk_rows,
                                 int32_t flags, void* workspace, size_t workspace_bytes,
                                 prl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PRL_H_ */
