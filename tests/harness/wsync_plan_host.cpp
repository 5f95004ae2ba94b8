// Host harness for the scatter + all-gather transfer plan of prl_wsync_bcast_bucket_sag: the SAME header the RCCL
// code executes (pipelinerl_amd/csrc/prl_wsync_plan.h), run for every rank of a simulated group against a table of
// sends and receives.  Test infrastructure (built by tests/test_wsync_plan_host.py with g++); no GPU, no RCCL.
//
//   prl_wsync_plan_check(world, nbytes, execute) -> 0, or a negative code naming the violated property:
//     -1 slices do not tile [0, nbytes) / overlap      -2 a non-empty slice does not start on a 256-byte boundary
//     -3 a send without its matching receive (peer, offset, length, order between the pair), or vice versa
//     -4 an op of zero length or outside the bucket     -5 rank 0 takes part in phase 2 / receives anything
//     -6 (execute) a receiver does not end up with the sender's bytes
#include <cstdint>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "../../pipelinerl_amd/csrc/prl_wsync_plan.h"

using prl::wsync::Op;

namespace {

struct Rec {
  uint64_t off, len;
};

// per ordered pair (from, to): the sequence of sends as the sender issues them / of receives as the receiver issues them
typedef std::map<std::pair<int, int>, std::vector<Rec>> Table;

int match(const Table& sends, const Table& recvs) {
  if (sends.size() != recvs.size()) return -3;
  for (const auto& kv : sends) {
    auto it = recvs.find(kv.first);
    if (it == recvs.end() || it->second.size() != kv.second.size()) return -3;
    for (size_t k = 0; k < kv.second.size(); ++k)
      if (kv.second[k].off != it->second[k].off || kv.second[k].len != it->second[k].len) return -3;
  }
  return 0;
}

}  // namespace

extern "C" int prl_wsync_plan_check(int world, uint64_t nbytes, int execute) {
  const int R = world - 1;
  if (R < 1) return 0;
  // ---- slices tile the bucket
  uint64_t at = 0;
  for (int i = 0; i < R; ++i) {
    const uint64_t lo = prl::wsync::slice_lo(nbytes, R, i), len = prl::wsync::slice_len(nbytes, R, i);
    if (lo != at) return -1;
    if (len && (lo % 256)) return -2;
    at = lo + len;
  }
  if (at != nbytes || prl::wsync::slice_lo(nbytes, R, R) != nbytes) return -1;

  std::vector<std::vector<uint8_t>> buf;
  if (execute) {
    buf.assign(world, std::vector<uint8_t>(nbytes, 0));
    for (uint64_t b = 0; b < nbytes; ++b) buf[0][b] = (uint8_t)(b * 131 + (b >> 8) * 7 + 1);
  }
  for (int phase = 0; phase < 2; ++phase) {
    Table sends, recvs;
    int bad = 0;
    for (int rank = 0; rank < world; ++rank) {
      auto emit = [&](const Op& op) {
        if (op.len == 0 || op.off + op.len > nbytes || op.peer < 0 || op.peer >= world || op.peer == rank) bad = -4;
        if (rank == 0 && (phase == 1 || !op.send)) bad = -5;
        if (op.send) {
          sends[{rank, op.peer}].push_back(Rec{op.off, op.len});
        } else {
          recvs[{op.peer, rank}].push_back(Rec{op.off, op.len});
        }
      };
      if (phase == 0) {
        prl::wsync::scatter_ops(rank, world, nbytes, emit);
      } else {
        prl::wsync::allgather_ops(rank, world, nbytes, emit);
      }
    }
    if (bad) return bad;
    if (int rc = match(sends, recvs)) return rc;
    if (execute) {
      // all transfers of a phase read the state BEFORE the phase (a group is one concurrent exchange)
      const auto before = buf;
      for (const auto& kv : sends)
        for (const Rec& r : kv.second) memcpy(buf[kv.first.second].data() + r.off, before[kv.first.first].data() + r.off, r.len);
    }
  }
  if (execute)
    for (int rank = 1; rank < world; ++rank)
      if (memcmp(buf[rank].data(), buf[0].data(), nbytes) != 0) return -6;
  return 0;
}

extern "C" uint64_t prl_wsync_plan_slice_lo(uint64_t nbytes, int receivers, int i) { return prl::wsync::slice_lo(nbytes, receivers, i); }
