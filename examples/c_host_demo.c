/* A plain C99 host of libprl.so: no Python, no PyTorch - only include/prl.h and the HIP runtime.
 *
 * It does what the reference's populate_rl_data does for one group of 4 rollouts
 * (pipelinerl/finetune/rl/__init__.py:453-570): leave-one-out advantages and the group's mean
 * token count, then counts labels / overflow per sequence (K5), and prints the numbers as JSON so
 * tests/test_gpu_c_host.py can compare them with the oracle.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host_demo.c \
 *       -L pipelinerl_amd/lib -lprl -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/pipelinerl_amd/lib -o c_host_demo
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "prl.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_PRL(x) do { int r_ = (x); if (r_ != PRL_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, prl_last_error()); return 3; } } while (0)

static void* to_device(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return NULL;
  if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  /* four rollouts of one problem: lengths 5, 3, 6, 4; prompt = first 2 tokens (labels -100) */
  enum { S = 4, N = 18, EOS = 2 };
  const int64_t seq_off[S + 1] = {0, 5, 8, 14, 18};
  const int32_t tokens[N] = {7, 8, 9, 10, 2,   7, 8, 11,   7, 8, 12, 13, 14, 15,   7, 8, 16, 2};
  int32_t labels[N];
  for (int s = 0; s < S; ++s)
    for (int64_t t = seq_off[s]; t < seq_off[s + 1]; ++t) labels[t] = (t - seq_off[s] < 2) ? -100 : tokens[t];
  const double reward[S] = {1.0, 0.0, 0.0, 1.0};
  const uint8_t finish_code[S] = {PRL_FINISH_STOP, PRL_FINISH_LENGTH, PRL_FINISH_NONE, PRL_FINISH_NONE};
  const uint8_t finished[S] = {1, 0, 0, 1};
  /* one (group, step) key holding all four sequences; one group with four distinct rollouts */
  const int32_t key_off[2] = {0, S}, key_members[S] = {0, 1, 2, 3}, group_off[2] = {0, S}, group_members[S] = {0, 1, 2, 3};
  const int32_t group_n_rollouts[1] = {S};

  if (prl_abi_version() != PRL_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  void *d_seq_off = to_device(seq_off, sizeof seq_off), *d_tokens = to_device(tokens, sizeof tokens),
       *d_labels = to_device(labels, sizeof labels), *d_reward = to_device(reward, sizeof reward),
       *d_code = to_device(finish_code, sizeof finish_code), *d_fin = to_device(finished, sizeof finished),
       *d_key_off = to_device(key_off, sizeof key_off), *d_key_members = to_device(key_members, sizeof key_members),
       *d_group_off = to_device(group_off, sizeof group_off), *d_group_members = to_device(group_members, sizeof group_members),
       *d_nroll = to_device(group_n_rollouts, sizeof group_n_rollouts);
  double *d_adv64, *d_gt64;
  float *d_adv32, *d_gt32, *d_nl, *d_ovf;
  CHECK_HIP(hipMalloc((void**)&d_adv64, S * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&d_gt64, S * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&d_adv32, S * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_gt32, S * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_nl, S * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_ovf, S * sizeof(float)));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  CHECK_PRL(prl_seq_scan(S, d_tokens, d_labels, d_seq_off, d_code, d_fin, EOS, d_nl, d_ovf, stream));
  CHECK_PRL(prl_group_advantages(S, 1, 1, d_key_off, d_key_members, d_group_off, d_group_members, d_nroll, d_reward, d_seq_off,
                                 /*divide_by_std=*/1, d_adv64, d_gt64, d_adv32, d_gt32, stream));
  /* a bad call: the error comes back as a code + message, nothing aborts */
  const int bad = prl_seq_scan(-1, d_tokens, d_labels, d_seq_off, d_code, d_fin, EOS, d_nl, d_ovf, stream);
  CHECK_HIP(hipStreamSynchronize(stream));

  double adv[S], gt[S];
  float nl[S], ovf[S];
  CHECK_HIP(hipMemcpy(adv, d_adv64, sizeof adv, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gt, d_gt64, sizeof gt, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(nl, d_nl, sizeof nl, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(ovf, d_ovf, sizeof ovf, hipMemcpyDeviceToHost));
  printf("{\"advantage\": [%.17g, %.17g, %.17g, %.17g], \"group_tokens\": [%.17g, %.17g, %.17g, %.17g], "
         "\"num_labels\": [%g, %g, %g, %g], \"overflow\": [%g, %g, %g, %g], \"bad_call\": %d, \"bad_call_message\": \"%s\"}\n",
         adv[0], adv[1], adv[2], adv[3], gt[0], gt[1], gt[2], gt[3], nl[0], nl[1], nl[2], nl[3], ovf[0], ovf[1], ovf[2], ovf[3],
         bad, prl_last_error());
  return 0;
}
