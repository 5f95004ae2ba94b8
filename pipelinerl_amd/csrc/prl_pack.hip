// K5 / K6 / K7: group-baseline advantages and pack / pad collate on device.
//
// The reference builds these with pandas + Python lists (per-token lists of one
// repeated scalar, rl/__init__.py:453-594) and a Python loop of torch.tensor(list)
// per sequence and per field (data.py:163-283).  Here the rollouts live in HBM as
// ragged SoA buffers (int32 tokens/labels, fp32 completion logprobs, per-sequence
// scalars) and ONE launch expands a whole optimizer step's worth of sequences into
// the PipelineBatchEncoding layout (5 x int64 + 7 x fp32 per token = 68 B written,
// 16 B read per token): every lane owns 4 consecutive output tokens, so each of the
// 12 output arrays is written with full 16/32-byte stores, wave-contiguous.

#include <cstdlib>

#include "prl_common.h"

namespace {

using prl::kWave;
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 4096;
constexpr bool kPackNtDefault = true;  // +3 % at 33.5 M tokens (scripts/pack_bench.py)

// ---------------------------------------------------------------------------------------
// K5a: per-sequence scan (num_labels, overflow flag).  One workgroup per sequence.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void seq_scan_kernel(
    int32_t n_seqs, const int32_t* __restrict__ tokens, const int32_t* __restrict__ labels,
    const int64_t* __restrict__ seq_off, const uint8_t* __restrict__ finish_code,
    const uint8_t* __restrict__ finished, int32_t eos, float* __restrict__ num_labels,
    float* __restrict__ overflow) {
  __shared__ int lds_cnt[kBlock / kWave];
  __shared__ int lds_eos[kBlock / kWave];
  for (int s = blockIdx.x; s < n_seqs; s += gridDim.x) {
    const int64_t b = seq_off[s], e = seq_off[s + 1];
    const int code = finish_code ? finish_code[s] : PRL_FINISH_NONE;
    const bool fin = finished ? (finished[s] != 0) : false;
    // the EOS scan only matters when nothing else decides (rl/__init__.py:550-552)
    const bool need_eos = (code == PRL_FINISH_NONE) && !fin;
    int cnt = 0, has_eos = 0;
    // ragged offsets are 4-byte aligned only: up to three elements on either side of the 16-byte aligned body go one by
    // one, the body in 16-byte loads (4 elements per lane: 4 KB per wave-instruction instead of 256 bytes)
    int64_t a0 = (b + 3) & ~int64_t(3), a1 = e & ~int64_t(3);
    // no aligned quad inside [b, e), or unaligned bases: everything is "head" (head [b, a0), body [a0, a1), tail [a1, e))
    if (a0 > a1 || ((reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(tokens)) & 15)) a0 = a1 = e;
    for (int64_t i = b + threadIdx.x; i < a0; i += kBlock) {
      cnt += (labels[i] != -100);
      if (need_eos) has_eos |= (tokens[i] == eos);
    }
    const int4* l4 = reinterpret_cast<const int4*>(labels);
    const int4* t4 = reinterpret_cast<const int4*>(tokens);
    for (int64_t q = (a0 >> 2) + threadIdx.x; q < (a1 >> 2); q += kBlock) {
      const int4 l = l4[q];
      cnt += (l.x != -100) + (l.y != -100) + (l.z != -100) + (l.w != -100);
      if (need_eos) {
        const int4 t = t4[q];
        has_eos |= (t.x == eos) | (t.y == eos) | (t.z == eos) | (t.w == eos);
      }
    }
    for (int64_t i = a1 + threadIdx.x; i < e; i += kBlock) {
      cnt += (labels[i] != -100);
      if (need_eos) has_eos |= (tokens[i] == eos);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cnt += __shfl_xor(cnt, o, 64);
      has_eos |= __shfl_xor(has_eos, o, 64);
    }
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) {
      lds_cnt[wid] = cnt;
      lds_eos[wid] = has_eos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int c = 0, h = 0;
      for (int w = 0; w < kBlock / kWave; ++w) {
        c += lds_cnt[w];
        h |= lds_eos[w];
      }
      float ov;
      if (code == PRL_FINISH_LENGTH) {
        ov = 1.0f;
      } else if (code == PRL_FINISH_STOP) {
        ov = 0.0f;
      } else if (fin) {
        ov = 0.0f;
      } else {
        ov = h ? 0.0f : 1.0f;
      }
      num_labels[s] = (float)c;
      overflow[s] = ov;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// K5b: leave-one-out advantages per (group_id, step_index) key; mean rollout tokens per
// group.  fp64 like the reference's pandas path.  One WAVE per key / per group: the members'
// values are loaded by the 64 lanes at once (one lane per member: two dependent loads deep instead
// of 3 x n), then every lane adds them up in DATASET ORDER out of its neighbours' registers
// (`__shfl`), so the sums are the ones a single lane walking the members would form - the results
// do not depend on the launch geometry.  Keys larger than a wave are walked in chunks of 64.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double ordered_wave_sum(double acc, double v, int count) {
  for (int j = 0; j < count; ++j) acc += __shfl(v, j, kWave);
  return acc;
}

__global__ __launch_bounds__(kBlock) void group_adv_kernel(
    int32_t n_keys, int32_t n_groups, const int32_t* __restrict__ key_off,
    const int32_t* __restrict__ key_members, const int32_t* __restrict__ group_off,
    const int32_t* __restrict__ group_members, const int32_t* __restrict__ group_n_rollouts,
    const double* __restrict__ reward, const int64_t* __restrict__ seq_off, int divide_by_std,
    double* __restrict__ adv64, double* __restrict__ gt64, float* __restrict__ adv32,
    float* __restrict__ gt32) {
  const int lane = threadIdx.x & (kWave - 1);
  const int w = (blockIdx.x * kBlock + threadIdx.x) / kWave;  // one wave per unit: keys first, then groups
  if (w < n_keys) {
    const int b = key_off[w], e = key_off[w + 1];
    const int n = e - b;
    double sum = 0.0;
    for (int c = b; c < e; c += kWave) {
      const int i = c + lane;
      const double v = i < e ? reward[key_members[i]] : 0.0;
      sum = ordered_wave_sum(sum, v, min(kWave, e - c));
    }
    double sd = 0.0;  // nan_to_num(std): NaN for singleton keys -> 0
    if (n > 1) {
      const double mean = sum / (double)n;
      double ss = 0.0;
      for (int c = b; c < e; c += kWave) {
        const int i = c + lane;
        const double d = i < e ? reward[key_members[i]] - mean : 0.0;
        ss = ordered_wave_sum(ss, d * d, min(kWave, e - c));
      }
      sd = sqrt(ss / (double)(n - 1));  // pandas std, ddof = 1
    }
    for (int i = b + lane; i < e; i += kWave) {
      const int s = key_members[i];
      const double r = reward[s];
      const double loo = (n > 1) ? (sum - r) / (double)(n - 1) : r;
      const double a = divide_by_std ? (r - loo) / (sd + 1e-4) : (r - loo);
      adv64[s] = a;
      adv32[s] = (float)a;
    }
  } else if (w - n_keys < n_groups) {
    const int t = w - n_keys;
    const int b = group_off[t], e = group_off[t + 1];
    int64_t tok = 0;  // integers: any order gives the same sum
    for (int i = b + lane; i < e; i += kWave) {
      const int s = group_members[i];
      tok += seq_off[s + 1] - seq_off[s];
    }
    for (int off = kWave / 2; off > 0; off >>= 1) tok += __shfl_xor(tok, off, kWave);
    const double g = (double)tok / (double)group_n_rollouts[t];
    for (int i = b + lane; i < e; i += kWave) {
      const int s = group_members[i];
      gt64[s] = g;
      gt32[s] = (float)g;
    }
  }
}

// ---------------------------------------------------------------------------------------
// K6: pack collate
// ---------------------------------------------------------------------------------------
struct PackArgs {
  int32_t m;
  int64_t total;
  const int32_t* pk_src;
  const int64_t* pk_dst;
  const int32_t* pk_seg;
  const int32_t* tokens;
  const int32_t* labels;
  const float* lp;
  const float* ref_lp;
  const int64_t* seq_off;
  const int64_t* lp_off;
  const float* reward;
  const float* adv;
  const float* gt;
  const float* nl;
  const float* ovf;
  int32_t per_token;  // bit k set: column k (reward, adv, gt, nl, ovf) is a per-token ragged array
  int32_t eos;
  int64_t* o_ids;
  int64_t* o_labels;
  int64_t* o_mask;
  int64_t* o_pos;
  int64_t* o_seg;
  float* o_rewards;
  float* o_adv;
  float* o_ref;
  float* o_old;
  float* o_gt;
  float* o_nl;
  float* o_ovf;
};

// largest j in [0, m) with dst[j] <= t  (dst non-decreasing, dst[0] == 0, t < dst[m])
__device__ __forceinline__ int find_seq(const int64_t* __restrict__ dst, int m, int64_t t) {
  int lo = 0, hi = m;  // invariant: dst[lo] <= t < dst[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (dst[mid] <= t) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  return lo;
}

struct SeqCtx {
  int64_t dst_b, dst_e;  // destination range
  int64_t src_b;         // source token offset (or -1 for a sentinel filler)
  int64_t lp_b;          // source logprob offset
  int64_t lp_skip;       // L - n_logprobs: left zero padding inside the sequence
  int32_t seg;
  float reward, adv, gt, nl, ovf;
};

__device__ __forceinline__ void load_seq(const PackArgs& a, int j, SeqCtx& c) {
  c.dst_b = a.pk_dst[j];
  c.dst_e = a.pk_dst[j + 1];
  c.seg = a.pk_seg[j];
  const int s = a.pk_src[j];
  if (s < 0) {  // sentinel filler (finetune/utils.py:62-78)
    c.src_b = -1;
    c.lp_b = 0;
    c.lp_skip = c.dst_e - c.dst_b;
    c.reward = 0.0f;
    c.adv = 0.0f;
    c.gt = 1.0f;
    c.nl = 1.0f;
    c.ovf = 0.0f;
  } else {
    c.src_b = a.seq_off[s];
    c.lp_b = a.lp_off[s];
    c.lp_skip = (c.dst_e - c.dst_b) - (a.lp_off[s + 1] - c.lp_b);
    c.reward = (a.per_token & 1) ? 0.0f : a.reward[s];
    c.adv = (a.per_token & 2) ? 0.0f : a.adv[s];
    c.gt = (a.per_token & 4) ? 0.0f : a.gt[s];
    c.nl = (a.per_token & 8) ? 0.0f : a.nl[s];
    c.ovf = (a.per_token & 16) ? 0.0f : a.ovf[s];
  }
}

struct Tok {
  int64_t id, label, pos, seg;
  float reward, adv, ref, old, gt, nl, ovf;
};

__device__ __forceinline__ void make_token(const PackArgs& a, const SeqCtx& c, int64_t t, Tok& o) {
  const int64_t i = t - c.dst_b;
  o.pos = i;
  o.seg = c.seg;
  o.reward = c.reward;
  o.adv = c.adv;
  o.gt = c.gt;
  o.nl = c.nl;
  o.ovf = c.ovf;
  if (c.src_b < 0) {
    o.id = a.eos;
    o.label = -100;
    o.ref = 0.0f;
    o.old = 0.0f;
    return;
  }
  o.id = a.tokens[c.src_b + i];
  if (a.per_token) {
    if (a.per_token & 1) o.reward = a.reward[c.src_b + i];
    if (a.per_token & 2) o.adv = a.adv[c.src_b + i];
    if (a.per_token & 4) o.gt = a.gt[c.src_b + i];
    if (a.per_token & 8) o.nl = a.nl[c.src_b + i];
    if (a.per_token & 16) o.ovf = a.ovf[c.src_b + i];
  }
  const int32_t lab = a.labels[c.src_b + i];
  // first token of every packed sequence but the first is never a target (data.py:264-265)
  o.label = (i == 0 && c.seg > 0) ? -100 : (int64_t)lab;
  const int64_t k = i - c.lp_skip;
  if (k >= 0) {
    o.old = a.lp[c.lp_b + k];
    o.ref = a.ref_lp ? a.ref_lp[c.lp_b + k] : o.old;
  } else {
    o.old = 0.0f;
    o.ref = 0.0f;
  }
}

// TPL tokens per lane and iteration.  Store shapes matter more than anything else here (measured with
// scripts/exp/store_patterns.hip on 33.5 M tokens): the int64 columns written as two 16-byte stores
// at a 32-byte lane stride cost 442 us (plain) / 820 us (non-temporal) for the twelve columns, while
// wave-contiguous stores - TPL = 2: one 16-byte store per int64 column, one 8-byte store per fp32
// column - take 404 us with the non-temporal hint.
template <bool VEC, bool NT, int TPL>
__global__ __launch_bounds__(kBlock) void pack_collate_kernel(PackArgs a) {
  // Each workgroup owns one contiguous run of token groups, so the sequence index only moves
  // forward: ONE binary search (12 dependent L2 round trips) per workgroup instead of one per
  // group, then every lane walks on from where its previous group ended - zero steps most of the
  // time, a few when short sequences end inside the span.
  const int64_t ngroups = (a.total + TPL - 1) / TPL;
  const int64_t per_block = (ngroups + gridDim.x - 1) / gridDim.x;
  const int64_t g_begin = (int64_t)blockIdx.x * per_block;
  const int64_t g_end = (g_begin + per_block < ngroups) ? (g_begin + per_block) : ngroups;
  if (g_begin >= g_end) return;
  int j = find_seq(a.pk_dst, a.m, g_begin * TPL);
  SeqCtx c;
  int cj = -1;  // sequence whose scalars `c` holds
  for (int64_t gidx = g_begin + threadIdx.x; gidx < g_end; gidx += kBlock) {
    const int64_t t0 = gidx * TPL;
    while (t0 >= a.pk_dst[j + 1]) ++j;  // t0 < total = pk_dst[m] bounds the walk
    if (cj != j) {
      load_seq(a, j, c);
      cj = j;
    }
    Tok tk[TPL];
    const int cnt = (a.total - t0) < TPL ? (int)(a.total - t0) : TPL;
    // The lane's TPL tokens inside ONE sequence with scalar columns (all but a few groups at sequence ends): their ids, labels and
    // log-probs are TPL consecutive 4-byte words each - one load per column instead of TPL (the ragged offsets are only 4-byte
    // aligned; global loads of 8 / 16 bytes need no more than that on gfx950)
    if (VEC && a.per_token == 0 && c.src_b >= 0 && t0 + TPL <= c.dst_e) {
      typedef int iv __attribute__((ext_vector_type(TPL), aligned(4)));
      typedef float fv __attribute__((ext_vector_type(TPL), aligned(4)));
      const int64_t i0 = t0 - c.dst_b;
      const iv ids = *reinterpret_cast<const iv*>(a.tokens + c.src_b + i0);
      const iv labs = *reinterpret_cast<const iv*>(a.labels + c.src_b + i0);
      const int64_t k0 = i0 - c.lp_skip;
      fv old, ref;
      if (k0 >= 0) {
        old = *reinterpret_cast<const fv*>(a.lp + c.lp_b + k0);
        ref = a.ref_lp ? *reinterpret_cast<const fv*>(a.ref_lp + c.lp_b + k0) : old;
      } else {
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
          old[k] = k0 + k >= 0 ? a.lp[c.lp_b + k0 + k] : 0.0f;
          ref[k] = k0 + k >= 0 ? (a.ref_lp ? a.ref_lp[c.lp_b + k0 + k] : old[k]) : 0.0f;
        }
      }
#pragma unroll
      for (int k = 0; k < TPL; ++k) {
        tk[k].id = ids[k];
        tk[k].label = (i0 + k == 0 && c.seg > 0) ? -100 : (int64_t)labs[k];  // data.py:264-265
        tk[k].pos = i0 + k;
        tk[k].seg = c.seg;
        tk[k].reward = c.reward;
        tk[k].adv = c.adv;
        tk[k].gt = c.gt;
        tk[k].nl = c.nl;
        tk[k].ovf = c.ovf;
        tk[k].old = old[k];
        tk[k].ref = ref[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < TPL; ++k) {
        if (k < cnt) {
          const int64_t t = t0 + k;
          while (t >= c.dst_e) {  // also skips zero-length sequences
            ++j;
            load_seq(a, j, c);
            cj = j;
          }
          make_token(a, c, t, tk[k]);
        } else {
          tk[k] = tk[0];
        }
      }
    }
    if (VEC && cnt == TPL) {
      using l2 = long __attribute__((ext_vector_type(2)));
      using f4 = float __attribute__((ext_vector_type(4)));
      using f2 = float __attribute__((ext_vector_type(2)));
      auto put = [&](auto* p, auto v) {
        if constexpr (NT) {  // the batch is written once and read much later: keep it out of L2
          __builtin_nontemporal_store(v, p);
        } else {
          *p = v;
        }
      };
      if constexpr (TPL == 4) {
        auto st2 = [&](int64_t* p, int64_t x0, int64_t x1, int64_t x2, int64_t x3) {
          put(reinterpret_cast<l2*>(p + t0), l2{x0, x1});
          put(reinterpret_cast<l2*>(p + t0 + 2), l2{x2, x3});
        };
        auto stf = [&](float* p, float x0, float x1, float x2, float x3) { put(reinterpret_cast<f4*>(p + t0), f4{x0, x1, x2, x3}); };
        st2(a.o_ids, tk[0].id, tk[1].id, tk[2].id, tk[3].id);
        st2(a.o_labels, tk[0].label, tk[1].label, tk[2].label, tk[3].label);
        st2(a.o_mask, 1, 1, 1, 1);
        st2(a.o_pos, tk[0].pos, tk[1].pos, tk[2].pos, tk[3].pos);
        st2(a.o_seg, tk[0].seg, tk[1].seg, tk[2].seg, tk[3].seg);
        stf(a.o_rewards, tk[0].reward, tk[1].reward, tk[2].reward, tk[3].reward);
        stf(a.o_adv, tk[0].adv, tk[1].adv, tk[2].adv, tk[3].adv);
        stf(a.o_ref, tk[0].ref, tk[1].ref, tk[2].ref, tk[3].ref);
        stf(a.o_old, tk[0].old, tk[1].old, tk[2].old, tk[3].old);
        stf(a.o_gt, tk[0].gt, tk[1].gt, tk[2].gt, tk[3].gt);
        stf(a.o_nl, tk[0].nl, tk[1].nl, tk[2].nl, tk[3].nl);
        stf(a.o_ovf, tk[0].ovf, tk[1].ovf, tk[2].ovf, tk[3].ovf);
      } else {
        auto st2 = [&](int64_t* p, int64_t x0, int64_t x1) { put(reinterpret_cast<l2*>(p + t0), l2{x0, x1}); };
        auto stf = [&](float* p, float x0, float x1) { put(reinterpret_cast<f2*>(p + t0), f2{x0, x1}); };
        st2(a.o_ids, tk[0].id, tk[1].id);
        st2(a.o_labels, tk[0].label, tk[1].label);
        st2(a.o_mask, 1, 1);
        st2(a.o_pos, tk[0].pos, tk[1].pos);
        st2(a.o_seg, tk[0].seg, tk[1].seg);
        stf(a.o_rewards, tk[0].reward, tk[1].reward);
        stf(a.o_adv, tk[0].adv, tk[1].adv);
        stf(a.o_ref, tk[0].ref, tk[1].ref);
        stf(a.o_old, tk[0].old, tk[1].old);
        stf(a.o_gt, tk[0].gt, tk[1].gt);
        stf(a.o_nl, tk[0].nl, tk[1].nl);
        stf(a.o_ovf, tk[0].ovf, tk[1].ovf);
      }
    } else {
#pragma unroll
      for (int k = 0; k < TPL; ++k) {
        if (k >= cnt) break;
        const int64_t t = t0 + k;
        a.o_ids[t] = tk[k].id;
        a.o_labels[t] = tk[k].label;
        a.o_mask[t] = 1;
        a.o_pos[t] = tk[k].pos;
        a.o_seg[t] = tk[k].seg;
        a.o_rewards[t] = tk[k].reward;
        a.o_adv[t] = tk[k].adv;
        a.o_ref[t] = tk[k].ref;
        a.o_old[t] = tk[k].old;
        a.o_gt[t] = tk[k].gt;
        a.o_nl[t] = tk[k].nl;
        a.o_ovf[t] = tk[k].ovf;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// K7: padded collate  ([n_rows, padded_len], one sequence per row)
// ---------------------------------------------------------------------------------------
struct PadArgs {
  int32_t n_rows;
  int64_t padded_len;
  int32_t pad_left;
  const int32_t* row_src;
  const int32_t* tokens;
  const int32_t* labels;
  const float* lp;
  const float* ref_lp;
  const int64_t* seq_off;
  const int64_t* lp_off;
  const float* reward;
  const float* adv;
  const float* gt;
  const float* nl;
  const float* ovf;
  int32_t per_token;
  int64_t* o_ids;
  int64_t* o_labels;
  int64_t* o_mask;
  float* o_rewards;
  float* o_adv;
  float* o_ref;
  float* o_old;
  float* o_gt;
  float* o_nl;
  float* o_ovf;
};

__global__ __launch_bounds__(kBlock) void pad_collate_kernel(PadArgs a) {
  const int64_t total = (int64_t)a.n_rows * a.padded_len;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += nthreads) {
    const int64_t row = t / a.padded_len;
    const int64_t col = t - row * a.padded_len;
    const int s = a.row_src[row];
    const int64_t sb = a.seq_off[s];
    const int64_t L = a.seq_off[s + 1] - sb;
    const int64_t i = a.pad_left ? (col - (a.padded_len - L)) : col;
    const bool valid = (i >= 0) && (i < L);
    int64_t id = 0, lab = -100, mask = 0;
    float rw = 0.f, ad = 0.f, rf = 0.f, ol = 0.f, g = 0.f, n = 0.f, ov = 0.f;
    if (valid) {
      id = a.tokens[sb + i];
      lab = a.labels[sb + i];
      mask = 1;
      rw = (a.per_token & 1) ? a.reward[sb + i] : a.reward[s];
      ad = (a.per_token & 2) ? a.adv[sb + i] : a.adv[s];
      g = (a.per_token & 4) ? a.gt[sb + i] : a.gt[s];
      n = (a.per_token & 8) ? a.nl[sb + i] : a.nl[s];
      ov = (a.per_token & 16) ? a.ovf[sb + i] : a.ovf[s];
      const int64_t lb = a.lp_off[s];
      const int64_t k = i - (L - (a.lp_off[s + 1] - lb));
      if (k >= 0) {
        ol = a.lp[lb + k];
        rf = a.ref_lp ? a.ref_lp[lb + k] : ol;
      }
    }
    a.o_ids[t] = id;
    a.o_labels[t] = lab;
    a.o_mask[t] = mask;
    a.o_rewards[t] = rw;
    a.o_adv[t] = ad;
    a.o_ref[t] = rf;
    a.o_old[t] = ol;
    a.o_gt[t] = g;
    a.o_nl[t] = n;
    a.o_ovf[t] = ov;
  }
}

int blocks_for(int64_t items) {
  int64_t b = (items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// tokens[i] := the_id wherever tokens[i] is outside [0, table_size) or valid[tokens[i]] == 0
// (reference preprocess.py:107-141, `replace_oov_tokens_with_the`: only input_ids change, labels keep
// what the actor sent).  One 4-byte read + one conditional 4-byte write per token; *patched counts them.
__global__ __launch_bounds__(kBlock) void patch_oov_kernel(int64_t n, int32_t* __restrict__ tokens,
                                                           const uint8_t* __restrict__ valid, int32_t table_size,
                                                           int32_t the_id, unsigned long long* __restrict__ patched) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t t = tokens[i];
    const bool ok = t >= 0 && t < table_size && valid[t] != 0;
    if (!ok) {
      tokens[i] = the_id;
      ++mine;
    }
  }
  if (patched && mine) atomicAdd(patched, mine);
}

}  // namespace

extern "C" int prl_patch_oov(int64_t n_tokens, int32_t* tokens, const uint8_t* valid, int32_t table_size,
                             int32_t the_token_id, uint64_t* patched, prl_stream_t stream) {
  PRL_CHECK_ARG(n_tokens >= 0 && table_size >= 0, "negative size");
  if (n_tokens == 0) return PRL_OK;
  PRL_CHECK_ARG(tokens && (valid || table_size == 0), "null pointer");
  const int64_t want = (n_tokens + kBlock - 1) / kBlock;
  const int nb = (int)(want < kMaxBlocks ? want : kMaxBlocks);
  hipLaunchKernelGGL(patch_oov_kernel, dim3(nb), dim3(kBlock), 0, static_cast<hipStream_t>(stream), n_tokens, tokens, valid,
                     table_size, the_token_id, reinterpret_cast<unsigned long long*>(patched));
  PRL_LAUNCH_CHECK("patch_oov_kernel");
  return PRL_OK;
}

extern "C" int prl_seq_scan(int32_t n_seqs, const int32_t* tokens, const int32_t* labels,
                            const int64_t* seq_off, const uint8_t* finish_code,
                            const uint8_t* finished, int32_t eos_token_id, float* num_labels,
                            float* overflow, prl_stream_t stream) {
  PRL_CHECK_ARG(n_seqs >= 0, "n_seqs < 0");
  if (n_seqs == 0) return PRL_OK;
  PRL_CHECK_ARG(tokens && labels && seq_off && num_labels && overflow, "null pointer");
  const int nb = n_seqs < kMaxBlocks ? n_seqs : kMaxBlocks;
  hipLaunchKernelGGL(seq_scan_kernel, dim3(nb), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     n_seqs, tokens, labels, seq_off, finish_code, finished, eos_token_id, num_labels,
                     overflow);
  PRL_LAUNCH_CHECK("seq_scan_kernel");
  return PRL_OK;
}

extern "C" int prl_group_advantages(int32_t n_seqs, int32_t n_keys, int32_t n_groups,
                                    const int32_t* key_off, const int32_t* key_members,
                                    const int32_t* group_off, const int32_t* group_members,
                                    const int32_t* group_n_rollouts, const double* reward,
                                    const int64_t* seq_off, int32_t divide_by_std,
                                    double* advantage64, double* group_tokens64,
                                    float* advantage32, float* group_tokens32,
                                    prl_stream_t stream) {
  PRL_CHECK_ARG(n_seqs >= 0 && n_keys >= 0 && n_groups >= 0, "negative count");
  if (n_seqs == 0) return PRL_OK;
  PRL_CHECK_ARG(key_off && key_members && group_off && group_members && group_n_rollouts && reward &&
                    seq_off && advantage64 && group_tokens64 && advantage32 && group_tokens32,
                "null pointer");
  const int64_t waves = (int64_t)n_keys + n_groups;  // one wave per key and per group
  const int64_t per_block = kBlock / kWave;
  hipLaunchKernelGGL(group_adv_kernel, dim3((unsigned)((waves + per_block - 1) / per_block)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), n_keys, n_groups, key_off, key_members,
                     group_off, group_members, group_n_rollouts, reward, seq_off, divide_by_std,
                     advantage64, group_tokens64, advantage32, group_tokens32);
  PRL_LAUNCH_CHECK("group_adv_kernel");
  return PRL_OK;
}

extern "C" int prl_pack_collate(int32_t m, int64_t total_tokens, const int32_t* pk_src,
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
                                prl_stream_t stream) {
  PRL_CHECK_ARG(m >= 0 && total_tokens >= 0, "negative size");
  if (m == 0 || total_tokens == 0) return PRL_OK;
  PRL_CHECK_ARG(pk_src && pk_dst && pk_seg && tokens && labels && logprobs && seq_off && lp_off &&
                    reward && advantage && group_tokens && num_labels && overflow,
                "null input pointer");
  PRL_CHECK_ARG(out_input_ids && out_labels && out_attention_mask && out_position_ids &&
                    out_segment_ids && out_rewards && out_advantages && out_ref_logprobs &&
                    out_old_logprobs && out_group_tokens && out_num_labels && out_overflow,
                "null output pointer");
  PackArgs a{m,          total_tokens, pk_src,       pk_dst,         pk_seg,     tokens,
             labels,     logprobs,     ref_logprobs, seq_off,        lp_off,     reward,
             advantage,  group_tokens, num_labels,   overflow,       per_token_columns, eos_token_id,
             out_input_ids, out_labels, out_attention_mask, out_position_ids, out_segment_ids,
             out_rewards, out_advantages, out_ref_logprobs, out_old_logprobs, out_group_tokens,
             out_num_labels, out_overflow};
  const bool vec = prl::aligned16(out_input_ids) && prl::aligned16(out_labels) &&
                   prl::aligned16(out_attention_mask) && prl::aligned16(out_position_ids) &&
                   prl::aligned16(out_segment_ids) && prl::aligned16(out_rewards) &&
                   prl::aligned16(out_advantages) && prl::aligned16(out_ref_logprobs) &&
                   prl::aligned16(out_old_logprobs) && prl::aligned16(out_group_tokens) &&
                   prl::aligned16(out_num_labels) && prl::aligned16(out_overflow);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // two tokens per lane + non-temporal stores (every store wave-contiguous; profiles/r01u_pack_variants.txt: the four-tokens-per-lane
  // and plain-store forms measured slower and are no longer built); unaligned outputs take the scalar-store form
  const int tpl = vec ? 2 : 4;
  const int nb = blocks_for((total_tokens + tpl - 1) / tpl);
  if (vec) {
    hipLaunchKernelGGL((pack_collate_kernel<true, kPackNtDefault, 2>), dim3(nb), dim3(kBlock), 0, s, a);
  } else {
    hipLaunchKernelGGL((pack_collate_kernel<false, false, 4>), dim3(nb), dim3(kBlock), 0, s, a);
  }
  PRL_LAUNCH_CHECK("pack_collate_kernel");
  return PRL_OK;
}

extern "C" int prl_pad_collate(int32_t n_rows, int64_t padded_len, int32_t pad_left,
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
                               prl_stream_t stream) {
  PRL_CHECK_ARG(n_rows >= 0 && padded_len >= 0, "negative size");
  if (n_rows == 0 || padded_len == 0) return PRL_OK;
  PRL_CHECK_ARG(row_src && tokens && labels && logprobs && seq_off && lp_off && reward && advantage &&
                    group_tokens && num_labels && overflow,
                "null input pointer");
  PRL_CHECK_ARG(out_input_ids && out_labels && out_attention_mask && out_rewards && out_advantages &&
                    out_ref_logprobs && out_old_logprobs && out_group_tokens && out_num_labels &&
                    out_overflow,
                "null output pointer");
  PadArgs a{n_rows,  padded_len, pad_left,     row_src,    tokens,       labels,   logprobs,
            ref_logprobs, seq_off, lp_off,     reward,     advantage,    group_tokens, num_labels,
            overflow, per_token_columns, out_input_ids, out_labels, out_attention_mask, out_rewards, out_advantages,
            out_ref_logprobs, out_old_logprobs, out_group_tokens, out_num_labels, out_overflow};
  const int nb = blocks_for((int64_t)n_rows * padded_len);
  hipLaunchKernelGGL(pad_collate_kernel, dim3(nb), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a);
  PRL_LAUNCH_CHECK("pad_collate_kernel");
  return PRL_OK;
}
