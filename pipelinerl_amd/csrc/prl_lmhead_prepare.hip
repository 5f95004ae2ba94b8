// Fused output head: operand preparation (fp32 / bf16 weight -> bf16 planes and their transposes) and workspace sizing.

#include "prl_lmhead_core.h"

using namespace prl::lmhead;

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
  if (fwd_bytes) *fwd_bytes = align256((size_t)ns * padded * 16) + align256((size_t)padded * 4);
  if (bwd_bytes) {
    PRL_CHECK_ARG(chunk_rows >= 1, "chunk_rows must be >= 1");
    *bwd_bytes = bwd_layout(hidden, vocab, chunk_rows < n ? chunk_rows : n).total;
  }
  return PRL_OK;
}