// Transfer plan of prl_wsync_bcast_bucket_sag (scatter + all-gather broadcast of one byte bucket from rank 0 to the
// receivers 1 .. world-1), kept apart from RCCL so that the host-compiled test harness
// (tests/harness/wsync_plan_host.cpp) executes the SAME arithmetic against a table of simulated rank buffers.
//
//   slices   R = world - 1 receivers; slice i = bytes [lo(i), lo(i + 1)) with lo(i) = min(i * per, nbytes) and
//            per = ceil(nbytes / R) rounded up to 256 bytes (every slice starts on a 256-byte boundary; trailing
//            slices of a small bucket are empty and generate no transfer)
//   phase 1  rank 0 sends slice i to rank i + 1                       (R transfers, one per xGMI link of rank 0)
//   phase 2  receiver a sends its slice to every other receiver b and receives b's slice   (R (R - 1) transfers)
//
// Every op is emitted exactly once on each side (a Send on the sender's list has its Recv, same offset and length, on
// the peer's list), in an order both sides agree on - what ncclGroupStart / ncclGroupEnd needs.
#pragma once

#include <stdint.h>

namespace prl {
namespace wsync {

struct Op {
  int peer;       // the other rank
  uint64_t off;   // byte offset inside the bucket (same on both sides)
  uint64_t len;   // > 0
  bool send;
};

inline uint64_t slice_stride(uint64_t nbytes, int receivers) {
  const uint64_t r = (uint64_t)receivers;
  return ((nbytes + r - 1) / r + 255) / 256 * 256;
}
inline uint64_t slice_lo(uint64_t nbytes, int receivers, int i) {
  const uint64_t b = slice_stride(nbytes, receivers) * (uint64_t)i;
  return b < nbytes ? b : nbytes;
}
inline uint64_t slice_len(uint64_t nbytes, int receivers, int i) {
  return slice_lo(nbytes, receivers, i + 1) - slice_lo(nbytes, receivers, i);
}

// phase 1 ops of `rank`
template <class F>
inline void scatter_ops(int rank, int world, uint64_t nbytes, F&& emit) {
  const int R = world - 1;
  if (R < 1 || nbytes == 0) return;
  if (rank == 0) {
    for (int i = 0; i < R; ++i)
      if (slice_len(nbytes, R, i)) emit(Op{i + 1, slice_lo(nbytes, R, i), slice_len(nbytes, R, i), true});
  } else {
    const int i = rank - 1;
    if (slice_len(nbytes, R, i)) emit(Op{0, slice_lo(nbytes, R, i), slice_len(nbytes, R, i), false});
  }
}

// phase 2 ops of `rank` (none for rank 0 and for a single receiver)
template <class F>
inline void allgather_ops(int rank, int world, uint64_t nbytes, F&& emit) {
  const int R = world - 1;
  if (R < 2 || rank == 0 || nbytes == 0) return;
  const int me = rank - 1;
  for (int j = 0; j < R; ++j) {
    if (j == me) continue;
    if (slice_len(nbytes, R, me)) emit(Op{j + 1, slice_lo(nbytes, R, me), slice_len(nbytes, R, me), true});
    if (slice_len(nbytes, R, j)) emit(Op{j + 1, slice_lo(nbytes, R, j), slice_len(nbytes, R, j), false});
  }
}

}  // namespace wsync
}  // namespace prl
