// Online softmax state shared by the logits kernels (prl_logprob.hip) and the fused lm_head
// kernels (prl_lmhead.hip).  Everything is in base-2 units: y = logit * log2(e) / temperature,
//     M = max y,  S = sum 2^(y - M),  W = sum (y - M) 2^(y - M)
// so that  logsumexp = ln2 (M + log2 S)  and  entropy = ln2 (log2 S - W / S)
// (reference pipelinerl/finetune/rl/__init__.py:207-233).
#pragma once

#include <hip/hip_runtime.h>

namespace prl {
namespace osm {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -3.0e38f;  // finite "minus infinity" (keeps 0 * x well defined)

struct Osm {
  float M, S, W;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ void osm_init(Osm& s) {
  s.M = kNegBig;
  s.S = 0.0f;
  s.W = 0.0f;
}

// fold N values (already scaled to base-2 units) into the state
template <int N>
__device__ __forceinline__ void osm_push(Osm& s, const float (&y)[N]) {
  float mx = y[0];
#pragma unroll
  for (int i = 1; i < N; ++i) mx = fmaxf(mx, y[i]);
  if (mx > s.M) {
    const float dm = s.M - mx;
    const float sc = fast_exp2(dm);
    s.W = sc * __builtin_fmaf(dm, s.S, s.W);
    s.S = sc * s.S;
    s.M = mx;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float d = y[i] - s.M;
    const float e = fast_exp2(d);
    s.S += e;
    s.W = __builtin_fmaf(d, e, s.W);
  }
}

__device__ __forceinline__ Osm osm_merge(const Osm& a, const Osm& b) {
  Osm r;
  r.M = fmaxf(a.M, b.M);
  const float da = a.M - r.M, db = b.M - r.M;
  const float ea = fast_exp2(da), eb = fast_exp2(db);
  r.S = ea * a.S + eb * b.S;
  r.W = ea * __builtin_fmaf(da, a.S, a.W) + eb * __builtin_fmaf(db, b.S, b.W);
  return r;
}

}  // namespace osm
}  // namespace prl
