// Value-head (actor-critic) branch of rl_step in ONE streaming pass (reference
// pipelinerl/finetune/rl/__init__.py:265-272 advantages := rewards - V, :367-381 the value loss
// 0.5 (V - reward)^2 w summed over the labelled tokens, :441-448 five statistics; the value
// predictions come from finetune/value_model.py:40-52, a Linear(hidden, 1) beside the lm_head).
// The reference runs ~12 elementwise launches, 3 sum_sum segment loops and 5 .item() syncs here.
//
// HBM-bound and tiny: per token 8 (label) + 4 or 2 (value) + 16 (reward, group_tokens, num_labels,
// overflow) bytes in, 8 out (the advantages column K2 reads next, d value_loss / d value).  One lane
// per (value[r, c], target c + 1) pair, grid-stride; per-lane fp64 sums -> wave64 shuffle tree ->
// per-block record -> a one-wave kernel folds the records in a fixed order (bitwise reproducible, no
// atomics).  Token weights follow prl_token_eval (prl_token_math.h) operation for operation.

#include "prl_common.h"
#include "prl_token_math.h"

namespace {

using prl::kWave;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kMaxBlocks = 1024;

enum { V_MEAN = 0, V_LOSS, V_MSE, V_COUNT, V_NUM_ADD, V_MAX = V_NUM_ADD, V_MIN, V_NUM = 8 };

struct ValueArgs {
  prl_loss_config cfg;
  int64_t n;  // rows * cols
  int64_t cols;
  const int64_t* labels;
  const void* values;
  const float* reward;
  const float* group_tokens;
  const float* num_labels;
  const float* overflow;
  float* adv_out;
  float* g_value;
  double* partials;  // [gridDim.x][V_NUM]
};

template <class VT>
__device__ __forceinline__ float load_value(const void* p, int64_t i);
template <>
__device__ __forceinline__ float load_value<float>(const void* p, int64_t i) {
  return static_cast<const float*>(p)[i];
}
template <>
__device__ __forceinline__ float load_value<uint16_t>(const void* p, int64_t i) {
  return __uint_as_float((uint32_t) static_cast<const uint16_t*>(p)[i] << 16);  // bf16 -> fp32 is exact
}

template <class VT>
__global__ __launch_bounds__(kBlock) void value_head_partial_kernel(ValueArgs a) {
  double s_mean = 0.0, s_loss = 0.0, s_mse = 0.0, s_cnt = 0.0;
  float mx = -INFINITY, mn = INFINITY;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
    const int64_t c = i % a.cols;
    if (c == 0) a.adv_out[i] = 0.0f;  // column 0 of the unshifted axis has no prediction behind it
    if (c + 1 == a.cols) {            // outputs.value[:, -1] has no target (:267)
      if (a.g_value) a.g_value[i] = 0.0f;
      continue;
    }
    const float v = load_value<VT>(a.values, i);
    const int64_t t = i + 1;  // the target this prediction is paired with
    const float r = a.reward[t];
    a.adv_out[t] = r - v;  // (:272) every position, labelled or not, like the reference's tensor
    float g = 0.0f;
    if (a.labels[t] != -100) {
      float w = a.cfg.group_normalization ? (1.0f / a.group_tokens[t]) : a.cfg.token_weight;  // (:245-255)
      if (a.cfg.overlong_filtering) w = w * (1.0f - a.overflow[t]);
      const float diff = v - r;
      const float sq = diff * diff;
      const float el = (0.5f * sq) * w;  // (:377)
      s_loss += (double)prl_nan_to_num0(el);
      if (prl_isfinite(el)) g = diff * w;  // nan_to_num passes gradient only where finite (rl/utils.py:26-31)
      const float nl = a.num_labels[t];
      s_mean += (double)prl_nan_to_num0(v / nl);
      s_mse += (double)prl_nan_to_num0(sq / nl);
      s_cnt += 1.0;
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
    if (a.g_value) a.g_value[i] = g;
  }
  __shared__ double red[kWaves][V_NUM];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  s_mean = prl::wave_sum(s_mean);
  s_loss = prl::wave_sum(s_loss);
  s_mse = prl::wave_sum(s_mse);
  s_cnt = prl::wave_sum(s_cnt);
  mx = prl::wave_max(mx);
  mn = prl::wave_min(mn);
  if (lane == 0) {
    red[wave][V_MEAN] = s_mean;
    red[wave][V_LOSS] = s_loss;
    red[wave][V_MSE] = s_mse;
    red[wave][V_COUNT] = s_cnt;
    red[wave][V_MAX] = (double)mx;
    red[wave][V_MIN] = (double)mn;
  }
  __syncthreads();
  if (threadIdx.x <= V_MIN) {
    const int e = threadIdx.x;
    double r = red[0][e];
    for (int k = 1; k < kWaves; ++k) r = e < V_NUM_ADD ? r + red[k][e] : (e == V_MAX ? fmax(r, red[k][e]) : fmin(r, red[k][e]));
    a.partials[(int64_t)blockIdx.x * V_NUM + e] = r;
  }
}

__global__ __launch_bounds__(kWave) void value_head_finalize_kernel(const double* partials, int nblocks, double* stats,
                                                                  float* value_loss_out) {
  double acc[V_NUM_ADD] = {0.0, 0.0, 0.0, 0.0};
  double mx = -INFINITY, mn = INFINITY;
  for (int b = threadIdx.x; b < nblocks; b += kWave) {
    const double* p = partials + (int64_t)b * V_NUM;
#pragma unroll
    for (int e = 0; e < V_NUM_ADD; ++e) acc[e] += p[e];
    mx = fmax(mx, p[V_MAX]);
    mn = fmin(mn, p[V_MIN]);
  }
#pragma unroll
  for (int e = 0; e < V_NUM_ADD; ++e) acc[e] = prl::wave_sum(acc[e]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmax(mx, __shfl_xor(mx, o, 64));
    mn = fmin(mn, __shfl_xor(mn, o, 64));
  }
  if (threadIdx.x == 0) {
    const bool any = acc[V_COUNT] > 0.0;  // "if masks_shifted.any() else 0.0" (:443-444)
    stats[PRL_VSTAT_VALUE_MEAN] = acc[V_MEAN];
    stats[PRL_VSTAT_VALUE_MAX] = any ? mx : 0.0;
    stats[PRL_VSTAT_VALUE_MIN] = any ? mn : 0.0;
    stats[PRL_VSTAT_VALUE_LOSS] = acc[V_LOSS];
    stats[PRL_VSTAT_VALUE_MSE] = acc[V_MSE];
    if (value_loss_out) *value_loss_out = (float)acc[V_LOSS];
  }
}

}  // namespace

extern "C" int prl_value_head_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes) {
  PRL_CHECK_ARG(bytes != nullptr, "bytes is null");
  PRL_CHECK_ARG(rows >= 0 && cols >= 0, "negative shape");
  *bytes = (size_t)kMaxBlocks * V_NUM * sizeof(double);
  return PRL_OK;
}

extern "C" int prl_value_head_fwd_bwd(const prl_loss_config* cfg, int64_t rows, int64_t cols, const int64_t* labels,
                                      const void* values, int values_dtype, const float* rewards,
                                      const float* group_tokens, const float* num_labels, const float* overflow,
                                      float* advantages_out, float* grad_values, float* value_loss_out, double* stats,
                                      void* workspace, size_t workspace_bytes, prl_stream_t stream) {
  PRL_CHECK_ARG(cfg != nullptr, "cfg is null");
  PRL_CHECK_ARG(rows >= 1 && cols >= 1, "rows and cols must be >= 1 (got %lld x %lld)", (long long)rows, (long long)cols);
  PRL_CHECK_ARG(values_dtype == PRL_DTYPE_F32 || values_dtype == PRL_DTYPE_BF16, "values must be float32 or bfloat16 (dtype code %d)",
                values_dtype);
  PRL_CHECK_ARG(labels && values && rewards && group_tokens && num_labels && overflow && advantages_out && stats, "null pointer");
  PRL_CHECK_ARG(workspace != nullptr, "workspace is null");
  size_t need = 0;
  prl_value_head_workspace_bytes(rows, cols, &need);
  if (workspace_bytes < need) return prl::set_error(PRL_ENOMEM, "workspace too small: %zu < %zu", workspace_bytes, need);
  ValueArgs a;
  a.cfg = *cfg;
  a.n = rows * cols;
  a.cols = cols;
  a.labels = labels;
  a.values = values;
  a.reward = rewards;
  a.group_tokens = group_tokens;
  a.num_labels = num_labels;
  a.overflow = overflow;
  a.adv_out = advantages_out;
  a.g_value = grad_values;
  a.partials = static_cast<double*>(workspace);
  const int64_t want = (a.n + kBlock - 1) / kBlock;
  const int grid = (int)(want < kMaxBlocks ? want : kMaxBlocks);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (values_dtype == PRL_DTYPE_F32) {
    hipLaunchKernelGGL(value_head_partial_kernel<float>, dim3(grid), dim3(kBlock), 0, s, a);
  } else {
    hipLaunchKernelGGL(value_head_partial_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, s, a);
  }
  PRL_LAUNCH_CHECK("value_head_partial_kernel");
  hipLaunchKernelGGL(value_head_finalize_kernel, dim3(1), dim3(kWave), 0, s, a.partials, grid, stats, value_loss_out);
  PRL_LAUNCH_CHECK("value_head_finalize_kernel");
  return PRL_OK;
}
