// Bucket <-> parameter copies for the weight update (flatten on the trainer, unflatten on the
// inference worker) as ONE launch per bucket instead of one copy per parameter.
//
// Replaces the per-parameter traffic of pipelinerl/finetune_loop.py:262-282 (one broadcast per
// tensor) / pipelinerl/vllm1.py:110-127 (one empty + load per tensor) on the staging side.
//
// HBM-bound byte work: 16 B/lane non-temporal loads and stores, 64 KiB chunks, a workgroup per
// chunk.  The segment table travels in the kernel arguments (scalar loads, no device allocation).
#include "prl_common.h"

namespace prl {

constexpr int kCopySegs = 64;            // segments per launch (table = 1.8 KB of kernarg)
constexpr int kCopyBlock = 256;
constexpr int64_t kCopyChunk = 64 << 10; // bytes per workgroup

struct CopyTable {
  const char* src[kCopySegs];
  char* dst[kCopySegs];
  int64_t nbytes[kCopySegs];
  int32_t first_chunk[kCopySegs + 1];  // prefix sums of ceil(nbytes / chunk)
  int32_t n;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kCopyBlock) void segment_copy_kernel(const CopyTable tab) {
  const int chunk = blockIdx.x;
  // scalar binary search: which segment owns this chunk
  int lo = 0, hi = tab.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab.first_chunk[mid] <= chunk) lo = mid; else hi = mid;
  }
  const int64_t begin = int64_t(chunk - tab.first_chunk[lo]) * kCopyChunk;
  const int64_t total = tab.nbytes[lo];
  const int64_t len = (total - begin < kCopyChunk) ? (total - begin) : kCopyChunk;
  const char* __restrict__ s = tab.src[lo] + begin;
  char* __restrict__ d = tab.dst[lo] + begin;
  const int tid = threadIdx.x;
  const bool vec_ok = (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0);
  if (vec_ok) {
    const int64_t nvec = len >> 4;
    const u32x4* __restrict__ sv = reinterpret_cast<const u32x4*>(s);
    u32x4* __restrict__ dv = reinterpret_cast<u32x4*>(d);
    int64_t i = tid;
    // full chunk: 4096 vectors = 16 per lane, 4 loads in flight per lane
    for (; i + 3 * kCopyBlock < nvec; i += 4 * kCopyBlock) {
      u32x4 a = __builtin_nontemporal_load(sv + i);
      u32x4 b = __builtin_nontemporal_load(sv + i + kCopyBlock);
      u32x4 c = __builtin_nontemporal_load(sv + i + 2 * kCopyBlock);
      u32x4 e = __builtin_nontemporal_load(sv + i + 3 * kCopyBlock);
      __builtin_nontemporal_store(a, dv + i);
      __builtin_nontemporal_store(b, dv + i + kCopyBlock);
      __builtin_nontemporal_store(c, dv + i + 2 * kCopyBlock);
      __builtin_nontemporal_store(e, dv + i + 3 * kCopyBlock);
    }
    for (; i < nvec; i += kCopyBlock) __builtin_nontemporal_store(__builtin_nontemporal_load(sv + i), dv + i);
    for (int64_t b = (nvec << 4) + tid; b < len; b += kCopyBlock) d[b] = s[b];
  } else {
    // unaligned view (a sliced parameter): byte lanes, correct but slow; not on the common path
    for (int64_t b = tid; b < len; b += kCopyBlock) d[b] = s[b];
  }
}

static int segment_copy(void* bucket, int64_t bucket_bytes, const prl_segment* segs, int64_t n, bool gather, hipStream_t stream) {
  PRL_CHECK_ARG(n >= 0, "negative segment count");
  PRL_CHECK_ARG(n == 0 || (bucket != nullptr && segs != nullptr), "null bucket or segment table");
  for (int64_t i = 0; i < n; ++i) {
    PRL_CHECK_ARG(segs[i].nbytes >= 0 && segs[i].bucket_offset >= 0, "segment %lld: negative size or offset", (long long)i);
    PRL_CHECK_ARG(segs[i].bucket_offset + segs[i].nbytes <= bucket_bytes, "segment %lld: [%lld, +%lld) exceeds the %lld-byte bucket",
                  (long long)i, (long long)segs[i].bucket_offset, (long long)segs[i].nbytes, (long long)bucket_bytes);
    PRL_CHECK_ARG(segs[i].nbytes == 0 || segs[i].tensor != nullptr, "segment %lld: null tensor pointer", (long long)i);
  }
  int64_t i = 0;
  while (i < n) {
    CopyTable tab;
    tab.n = 0;
    tab.first_chunk[0] = 0;
    int64_t chunks = 0;
    for (; i < n && tab.n < kCopySegs; ++i) {
      if (segs[i].nbytes == 0) continue;
      const int64_t c = (segs[i].nbytes + kCopyChunk - 1) / kCopyChunk;
      if (chunks + c > (int64_t(1) << 30)) break;  // keep the grid and the prefix sums in int32
      char* in_bucket = static_cast<char*>(bucket) + segs[i].bucket_offset;
      char* tensor = static_cast<char*>(segs[i].tensor);
      tab.src[tab.n] = gather ? tensor : in_bucket;
      tab.dst[tab.n] = gather ? in_bucket : tensor;
      tab.nbytes[tab.n] = segs[i].nbytes;
      chunks += c;
      tab.first_chunk[++tab.n] = int32_t(chunks);
    }
    if (tab.n == 0) {
      PRL_CHECK_ARG(i >= n, "segment %lld is too large for one launch", (long long)i);
      break;
    }
    hipLaunchKernelGGL(segment_copy_kernel, dim3(unsigned(chunks)), dim3(kCopyBlock), 0, stream, tab);
    PRL_LAUNCH_CHECK("segment_copy_kernel");
  }
  return PRL_OK;
}

}  // namespace prl

extern "C" int prl_bucket_gather(void* bucket, int64_t bucket_bytes, const prl_segment* segments, int64_t n_segments, void* stream) {
  return prl::segment_copy(bucket, bucket_bytes, segments, n_segments, true, static_cast<hipStream_t>(stream));
}

extern "C" int prl_bucket_scatter(const void* bucket, int64_t bucket_bytes, const prl_segment* segments, int64_t n_segments, void* stream) {
  return prl::segment_copy(const_cast<void*>(bucket), bucket_bytes, segments, n_segments, false, static_cast<hipStream_t>(stream));
}
