// Index arithmetic of the MFMA GEMM cores in prl_lmhead_core.h, kept in one header so that the device
// kernels and the host-compiled unit-test harness (tests/harness/lmhead_layout_host.cpp) use the SAME
// functions: tile raster, LDS image of a staged operand tile, fragment reads, accumulator layout.
//
// Tile: BM x BN x 64 (rows x columns x contraction, bf16).  The waves of a workgroup form a
// (waves / 2) x 2 grid; a wave computes 64 rows x (BN / 2) columns as 2 x (BN / 64) MFMA tiles of
// 32 x 32 (v_mfma_f32_32x32x16_bf16, the full-rate bf16 shape: 32 cycles per 32 768 flop).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define PRL_LHD __host__ __device__ __forceinline__
#else
#define PRL_LHD static inline
#endif

namespace prl {
namespace lmhead {

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;  // one tile row in LDS: 128 bytes = 8 chunks of 16

// Block id -> tile coordinates.  The hardware deals consecutive block ids round-robin over the 8
// XCDs (each with its own L2): give every XCD a contiguous range of the tile list, and walk that
// list in groups of 8 row tiles x all column tiles, row-fastest, so that the workgroups an XCD runs
// at a time cover a compact patch of tiles and share each A / B panel in its L2.
PRL_LHD void tile_coords_g(int bid, int mt, int nt, int gm, int& tm, int& tn) {
  const int total = mt * nt;
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int per_group = gm * nt;
  const int grp = L / per_group;
  const int first_m = grp * gm;
  const int gsz = (mt - first_m) < gm ? (mt - first_m) : gm;
  const int in = L - grp * per_group;
  tm = first_m + in % gsz;
  tn = in / gsz;
}
PRL_LHD void tile_coords(int bid, int mt, int nt, int& tm, int& tn) { tile_coords_g(bid, mt, nt, 8, tm, tn); }

// ---- staging.  An operand tile of R rows is R * 8 chunks of 16 bytes; thread `tid` of an NT-thread
// workgroup moves chunks q * NT + tid.  global_load_lds writes lane-linearly (wave base + lane * 16), so
// chunk c sits at LDS byte c * 16 = row (c >> 3), slot (c & 7); the XOR swizzle is applied to the SOURCE:
// the slot holds the logical k-chunk  slot ^ ((row >> 1) & 7).
PRL_LHD int stage_row(int tid, int q, int nt) { return q * (nt >> 3) + (tid >> 3); }
// ((q * NT / 8 + (tid >> 3)) >> 1) & 7 == (tid >> 4) & 7 for NT in {256, 512}: one source column per thread
PRL_LHD int stage_kcol(int tid) { return ((tid & 7) ^ ((tid >> 4) & 7)) * 8; }  // elements
PRL_LHD int stage_lds_byte(int tid, int q, int nt) { return (q * nt + tid) * 16; }

// ---- fragment reads for mfma_f32_32x32x16_bf16: lane l supplies row (l & 31) and the 8 contraction
// elements 8 * (l >> 5) .. + 7 of a 16-deep sub-step `ks` (0..3) of the 64-deep tile.
// Row r, logical chunk kc is at byte r * 128 + ((kc ^ ((r >> 1) & 7)) << 4); within a 16-lane group
// (16 consecutive rows, one kc) these are 16 distinct 16-byte slots of the 256-byte bank row.
PRL_LHD int frag_row(int lane, int wave_row0, int i) { return wave_row0 + i * 32 + (lane & 31); }  // wave_row0: first tile row of the wave
PRL_LHD int frag_kchunk(int lane, int ks) { return ks * 2 + (lane >> 5); }
PRL_LHD int frag_lds_byte(int lane, int wave_row0, int i, int ks) {
  const int r = frag_row(lane, wave_row0, i);
  return r * ROW_BYTES + ((frag_kchunk(lane, ks) ^ ((r >> 1) & 7)) << 4);
}

// ---- the 32-deep variant (dual-plane core: two A planes share one B tile per stage).  A tile row is
// 64 bytes = 4 chunks, four rows fill a 256-byte bank row; chunk c = q * NT + tid is row (c >> 2),
// slot (c & 3) and holds the logical k-chunk  slot ^ ((row >> 2) & 3).
constexpr int BK32 = 32;
constexpr int ROW_BYTES32 = BK32 * 2;
PRL_LHD int stage_row32(int tid, int q, int nt) { return q * (nt >> 2) + (tid >> 2); }
// ((q * NT / 4 + (tid >> 2)) >> 2) & 3 == (tid >> 4) & 3 for NT = 512
PRL_LHD int stage_kcol32(int tid) { return ((tid & 3) ^ ((tid >> 4) & 3)) * 8; }  // elements
// fragment of sub-step ks (0, 1): lane l supplies row (l & 31), k-chunk ks * 2 + (l >> 5)
PRL_LHD int frag_lds_byte32(int lane, int wave_row0, int i, int ks) {
  const int r = frag_row(lane, wave_row0, i);
  return r * ROW_BYTES32 + (((ks * 2 + (lane >> 5)) ^ ((r >> 2) & 3)) << 4);
}

// ---- d W from ROW-MAJOR d-logits planes: the contraction (token) index is the ROW of the staged tile, so the MFMA A
// fragment (8 consecutive tokens of one vocabulary entry) is gathered by ds_read_b64_tr_b16: a 16-lane group reads a
// [4 tokens][16 entries] block - lane q supplies the 8 bytes of token (q >> 2), entries 4 (q & 3) .. + 3 - and receives
// entry q's 4 tokens (profiles/r03b_mx_probe.txt).  A stage is 32 tokens x 256 entries: 32 rows of 32 sixteen-byte
// chunks, chunk c of row t at slot  t * 32 + (c ^ (4 (t & 3)))  - the 32 lanes of one LDS pass (4 rows x 4 chunks) then
// touch 16 distinct bank quads.  global_load_lds writes slots lane-linearly: the swizzle goes on the SOURCE chunk.
PRL_LHD int tr_stage_row(int tid, int q, int nt) { return (q * nt + tid) >> 5; }                        // token row 0..31
PRL_LHD int tr_stage_chunk(int tid, int q, int nt) { return ((q * nt + tid) & 31) ^ (4 * (((q * nt + tid) >> 5) & 3)); }  // logical chunk
// byte address of the tr read `r` (0: tokens 0-3 of the lane's 8, 1: tokens 4-7) of sub-step ks for MFMA tile i
PRL_LHD int tr_frag_lds_byte(int lane, int wave_row0, int i, int ks, int r) {
  const int g = lane >> 4, q = lane & 15;
  const int t = 16 * ks + 8 * (g >> 1) + 4 * r + (q >> 2);
  const int v = wave_row0 + 32 * i + 16 * (g & 1) + 4 * (q & 3);
  return ((t * 32 + ((v >> 3) ^ (4 * (t & 3)))) << 4) + 8 * (q & 1);
}
// which (token, entry) the lane's four results are: entry = wave_row0 + 32 i + (lane & 31), tokens 16 ks + 8 (lane >> 5) + 4 r + 0..3
PRL_LHD int tr_frag_entry(int lane, int wave_row0, int i) { return wave_row0 + 32 * i + (lane & 31); }
PRL_LHD int tr_frag_token0(int lane, int ks, int r) { return 16 * ks + 8 * (lane >> 5) + 4 * r; }

// ---- accumulators: element `reg` (0..15) of the 32 x 32 tile (i, j) of the wave whose sub-tile starts
// at (wave_row0, wave_col0) = (64 * (wave >> 1), (BN / 2) * (wave & 1))
PRL_LHD int acc_row_in_tile(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
PRL_LHD int acc_row(int lane, int wave_row0, int i, int reg) { return wave_row0 + i * 32 + acc_row_in_tile(lane, reg); }
PRL_LHD int acc_col(int lane, int wave_col0, int j) { return wave_col0 + j * 32 + (lane & 31); }

}  // namespace lmhead
}  // namespace prl
