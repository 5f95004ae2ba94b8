// Host build of the index arithmetic of the MFMA GEMM core (pipelinerl_amd/csrc/prl_lmhead_layout.h):
// a CPU emulation of one workgroup's staging, fragment reads and MFMA data flow, so that the swizzle,
// the lane maps and the tile raster are checked on a machine without a GPU.  Test infrastructure only.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../pipelinerl_amd/csrc/prl_lmhead_layout.h"

using namespace prl::lmhead;

extern "C" {

void lmh_tile_coords(int bid, int mt, int nt, int* tm, int* tn) { tile_coords(bid, mt, nt, *tm, *tn); }

// Emulates the staging of ONE operand tile of `rows` rows by an `nthreads`-thread workgroup:
// src is [rows][64] uint16 (row-major), lds_out rows * 128 bytes.
void lmh_stage(const uint16_t* src, uint8_t* lds_out, int rows, int nthreads) {
  const int nq = rows * 8 / nthreads;
  for (int tid = 0; tid < nthreads; ++tid)
    for (int q = 0; q < nq; ++q)
      memcpy(lds_out + stage_lds_byte(tid, q, nthreads), src + stage_row(tid, q, nthreads) * BK + stage_kcol(tid), 16);
}

// The 8 elements lane `lane` of wave row/column `w` feeds into the MFMA for tile i, sub-step ks,
// plus the (row, first k) the instruction's operand layout expects them to be.
void lmh_fragment(const uint8_t* lds, int lane, int w, int i, int ks, uint16_t* out8, int* row, int* k0) {
  memcpy(out8, lds + frag_lds_byte(lane, w, i, ks), 16);
  *row = frag_row(lane, w, i);
  *k0 = ks * 16 + 8 * (lane >> 5);
}

int lmh_frag_byte(int lane, int w, int i, int ks) { return frag_lds_byte(lane, w, i, ks); }

// ---- the 32-deep layout of the dual-plane core
void lmh_stage32(const uint16_t* src, uint8_t* lds_out, int rows, int nthreads) {  // src [rows][32]
  const int nq = rows * 4 / nthreads;
  for (int tid = 0; tid < nthreads; ++tid)
    for (int q = 0; q < nq; ++q)
      memcpy(lds_out + stage_lds_byte(tid, q, nthreads), src + stage_row32(tid, q, nthreads) * BK32 + stage_kcol32(tid), 16);
}
void lmh_fragment32(const uint8_t* lds, int lane, int row0, int i, int ks, uint16_t* out8, int* row, int* k0) {
  memcpy(out8, lds + frag_lds_byte32(lane, row0, i, ks), 16);
  *row = frag_row(lane, row0, i);
  *k0 = ks * 16 + 8 * (lane >> 5);
}
int lmh_frag_byte32(int lane, int row0, int i, int ks) { return frag_lds_byte32(lane, row0, i, ks); }

// Full emulation of C = A B^T for one (64 * wm_count) x bn x 64 tile through the staged images, the
// fragment reads and the documented MFMA semantics D[m][n] += sum_k A[m][k] B[n][k] with
// operand lane l: m (or n) = l & 31, k = 8 (l >> 5) + e;  D lane l, reg r: m = (r & 3) + 8 (r >> 2) + 4 (l >> 5), n = l & 31.
void lmh_emulate_tile(const uint16_t* a_src, const uint16_t* b_src, double* c_out /* [64 wm_count][bn] */, int wm_count, int bn) {
  const int bm = 64 * wm_count, nthreads = 128 * wm_count, BN = bn;
  std::vector<uint8_t> la(bm * ROW_BYTES), lb(BN * ROW_BYTES);
  lmh_stage(a_src, la.data(), bm, nthreads);
  lmh_stage(b_src, lb.data(), BN, nthreads);
  for (int wave = 0; wave < 2 * wm_count; ++wave) {
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * (bn / 2);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < bn / 64; ++j) {
        double d[32][32] = {{0}};
        for (int ks = 0; ks < 4; ++ks) {
          double am[32][16], bm_[32][16];
          for (int lane = 0; lane < 64; ++lane) {
            uint16_t fa[8], fb[8];
            memcpy(fa, la.data() + frag_lds_byte(lane, wm, i, ks), 16);
            memcpy(fb, lb.data() + frag_lds_byte(lane, wn, j, ks), 16);
            for (int e = 0; e < 8; ++e) {
              am[lane & 31][8 * (lane >> 5) + e] = (double)fa[e];
              bm_[lane & 31][8 * (lane >> 5) + e] = (double)fb[e];
            }
          }
          for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n)
              for (int k = 0; k < 16; ++k) d[m][n] += am[m][k] * bm_[n][k];
        }
        for (int lane = 0; lane < 64; ++lane)
          for (int reg = 0; reg < 16; ++reg)
            c_out[acc_row(lane, wm, i, reg) * BN + acc_col(lane, wn, j)] = d[acc_row_in_tile(lane, reg)][lane & 31];
      }
  }
}

// ---- d W from row-major planes (transposing LDS reads).  src is a [32 tokens][256 entries] stage tile, row-major.
void lmh_stage_tr(const uint16_t* src, uint8_t* lds_out, int nthreads) {
  for (int tid = 0; tid < nthreads; ++tid)
    for (int q = 0; q < 2; ++q)
      memcpy(lds_out + stage_lds_byte(tid, q, nthreads), src + tr_stage_row(tid, q, nthreads) * 256 + 8 * tr_stage_chunk(tid, q, nthreads), 16);
}
// ds_read_b64_tr_b16 as measured (profiles/r03b_mx_probe.txt): in a 16-lane group, lane q loads 8 bytes; result[q][e] =
// element (q & 3) of what lane (4 e + (q >> 2)) loaded.  `addr` = the 64 per-lane byte addresses; out[lane][0..3].
void lmh_tr_read(const uint8_t* lds, const int* addr, uint16_t* out /* [64][4] */) {
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, q = lane & 15;
    for (int e = 0; e < 4; ++e) {
      const int src_lane = 16 * g + 4 * e + (q >> 2);
      uint16_t piece[4];
      memcpy(piece, lds + addr[src_lane], 8);
      out[lane * 4 + e] = piece[q & 3];
    }
  }
}
int lmh_tr_frag_byte(int lane, int w, int i, int ks, int r) { return tr_frag_lds_byte(lane, w, i, ks, r); }
int lmh_tr_frag_entry(int lane, int w, int i) { return tr_frag_entry(lane, w, i); }
int lmh_tr_frag_token0(int lane, int ks, int r) { return tr_frag_token0(lane, ks, r); }
}
