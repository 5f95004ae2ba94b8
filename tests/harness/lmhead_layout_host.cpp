// Host build of the index arithmetic of the MFMA GEMM core (pipelinerl_amd/csrc/prl_lmhead_layout.h):
// a CPU emulation of one workgroup's staging, fragment reads and MFMA data flow, so that the swizzle,
// the lane maps and the tile raster are checked on a machine without a GPU.  Test infrastructure only.
#include <stdint.h>
#include <string.h>

#include "../../pipelinerl_amd/csrc/prl_lmhead_layout.h"

using namespace prl::lmhead;

extern "C" {

void lmh_tile_coords(int bid, int mt, int nt, int* tm, int* tn) { tile_coords(bid, mt, nt, *tm, *tn); }

// Emulates the staging of ONE operand tile: src is [128][64] uint16 (row-major), lds_out 16 KB.
void lmh_stage(const uint16_t* src, uint8_t* lds_out) {
  for (int tid = 0; tid < NTHREADS; ++tid)
    for (int q = 0; q < 4; ++q)
      memcpy(lds_out + stage_lds_byte(tid, q), src + stage_row(tid, q) * BK + stage_kcol(tid), 16);
}

// The 8 elements lane `lane` of wave row/column `w` feeds into the MFMA for row-block i, sub-step ks,
// plus the (row, first k) the instruction's operand layout expects them to be.
void lmh_fragment(const uint8_t* lds, int lane, int w, int i, int ks, uint16_t* out8, int* row, int* k0) {
  memcpy(out8, lds + frag_lds_byte(lane, w, i, ks), 16);
  *row = frag_row(lane, w, i);
  *k0 = ks * 32 + 8 * (lane >> 4);
}

int lmh_frag_byte(int lane, int w, int i, int ks) { return frag_lds_byte(lane, w, i, ks); }

// Full emulation of C = A B^T for one 128 x 128 x 64 tile through the staged images, the fragment
// reads and the documented MFMA semantics D[m][n] += sum_k A[m][k] B[n][k] with
// A operand lane l: m = l & 15, k = 8 (l >> 4) + e;  D lane l, reg r: m = 4 (l >> 4) + r, n = l & 15.
void lmh_emulate_tile(const uint16_t* a_src, const uint16_t* b_src, double* c_out /* [128][128] */) {
  static uint8_t la[TILE_BYTES], lb[TILE_BYTES];
  lmh_stage(a_src, la);
  lmh_stage(b_src, lb);
  for (int wave = 0; wave < 4; ++wave) {
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double d[16][16] = {{0}};
        for (int ks = 0; ks < 2; ++ks) {
          double am[16][32], bm[16][32];
          for (int lane = 0; lane < 64; ++lane) {
            uint16_t fa[8], fb[8];
            memcpy(fa, la + frag_lds_byte(lane, wm, i, ks), 16);
            memcpy(fb, lb + frag_lds_byte(lane, wn, j, ks), 16);
            for (int e = 0; e < 8; ++e) {
              am[lane & 15][8 * (lane >> 4) + e] = (double)fa[e];
              bm[lane & 15][8 * (lane >> 4) + e] = (double)fb[e];
            }
          }
          for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n)
              for (int k = 0; k < 32; ++k) d[m][n] += am[m][k] * bm[n][k];
        }
        for (int lane = 0; lane < 64; ++lane)
          for (int reg = 0; reg < 4; ++reg)
            c_out[acc_row(lane, wm, i, reg) * BN + acc_col(lane, wn, j)] = d[4 * (lane >> 4) + reg][lane & 15];
      }
  }
}
}
