// Index arithmetic of the MFMA GEMM core in prl_lmhead.hip, kept in one header so that the device
// kernels and the host-compiled unit-test harness (tests/harness/lmhead_layout_host.cpp) use the SAME
// functions: tile raster, LDS image of a staged operand tile, fragment reads, accumulator layout.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define PRL_LHD __host__ __device__ __forceinline__
#else
#define PRL_LHD static inline
#endif

namespace prl {
namespace lmhead {

constexpr int BM = 128, BN = 128, BK = 64;  // tile: rows x columns x contraction (bf16 elements)
constexpr int NTHREADS = 256;               // 4 waves as 2 (rows) x 2 (columns), 64 x 64 each
constexpr int TILE_BYTES = BM * BK * 2;     // one operand tile in LDS: 128 rows x 128 bytes

// Block id -> tile coordinates.  The hardware deals consecutive block ids round-robin over the 8
// XCDs (each with its own L2): give every XCD a contiguous range of the tile list, and walk that
// list in groups of 8 row tiles x all column tiles, row-fastest, so that the ~64 workgroups an XCD
// runs at a time cover ~8 x 8 tiles and share each A / B panel 8 ways in its L2.
PRL_LHD void tile_coords(int bid, int mt, int nt, int& tm, int& tn) {
  const int total = mt * nt;
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  constexpr int GM = 8;
  const int per_group = GM * nt;
  const int grp = L / per_group;
  const int first_m = grp * GM;
  const int gsz = (mt - first_m) < GM ? (mt - first_m) : GM;
  const int in = L - grp * per_group;
  tm = first_m + in % gsz;
  tn = in / gsz;
}

// ---- staging.  An operand tile is 1024 chunks of 16 bytes; thread `tid` moves chunks q * 256 + tid,
// q = 0..3.  global_load_lds writes lane-linearly (wave base + lane * 16), so chunk c sits at LDS byte
// c * 16 = row (c >> 3), slot (c & 7); the XOR swizzle is applied to the SOURCE: the slot holds the
// logical k-chunk  slot ^ ((row >> 1) & 7).
PRL_LHD int stage_row(int tid, int q) { return q * 32 + (tid >> 3); }
// ((q * 32 + (tid >> 3)) >> 1) & 7 == (tid >> 4) & 7 for every q: one source column per thread
PRL_LHD int stage_kcol(int tid) { return ((tid & 7) ^ ((tid >> 4) & 7)) * 8; }  // elements
PRL_LHD int stage_lds_byte(int tid, int q) { return (q * 256 + tid) * 16; }

// ---- fragment reads for mfma_f32_16x16x32_bf16: lane l supplies row (l & 15) and the 8 contraction
// elements 8 * (l >> 4) .. + 7 of a 32-deep sub-step `ks` (0, 1) of the 64-deep tile.
// Row r, logical chunk kc is at byte r * 128 + ((kc ^ ((r >> 1) & 7)) << 4); within a 16-lane group
// (16 consecutive rows, one kc) these are 16 distinct 16-byte slots of the 256-byte bank row.
PRL_LHD int frag_row(int lane, int w, int i) { return w * 64 + i * 16 + (lane & 15); }  // w: wave row / column
PRL_LHD int frag_kchunk(int lane, int ks) { return ks * 4 + (lane >> 4); }
PRL_LHD int frag_lds_byte(int lane, int w, int i, int ks) {
  const int r = frag_row(lane, w, i);
  return r * 128 + ((frag_kchunk(lane, ks) ^ ((r >> 1) & 7)) << 4);
}

// ---- accumulators: element `reg` of the 16 x 16 tile (i, j) of wave (wm, wn)
PRL_LHD int acc_row(int lane, int wm, int i, int reg) { return wm * 64 + i * 16 + 4 * (lane >> 4) + reg; }
PRL_LHD int acc_col(int lane, int wn, int j) { return wn * 64 + j * 16 + (lane & 15); }

}  // namespace lmhead
}  // namespace prl
