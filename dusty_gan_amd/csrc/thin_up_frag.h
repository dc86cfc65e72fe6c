// Weight fragments of thin_up_mfma (conv_thin.hip): for each boundary class of an output row (0 interior, 1 first row,
// 2 last row) the 18 MFMA A fragments [frag f = ((dr + 1) * 3 + (dc + 1)) * 2 + half][64 lanes][8 bf16], i.e. the taps of
// the 3 x 3 coarse neighbourhood folded per (row parity, column parity, output channel).  Built either by
// thin_up_prep_kernel in front of a launch (from the bf16 shadow) or, for weights registered by the caller, by the extra
// blocks of dg_transpose_shadow_multi_frags whenever the shadows are rebuilt (from the fp32 master, rounded to bf16 first:
// the same numbers).
#pragma once
#include "common.h"

constexpr int UP_FRAG_BYTES = 3 * 18 * 1024;
constexpr int UP_FRAG_BLOCKS = 36;               // blocks of 256 elements per class

// element e (0 .. 9215) of class cls; w(tap, n, ci) -> the bf16-rounded weight as float
// LO (the split-bf16 form of the fp32x3 mode: w returns the fp32 weight): the table of lo = bf16(v - bf16(v)) beside the table of
// hi = bf16(v), v the folded fp32 weight
template <bool LO = false, typename LoadW>
__device__ __forceinline__ void up_frag_element(int cls, int e, int N, int Hc, int adj, LoadW w, unsigned char* frag) {
  const int m = cls == 0 ? 1 : (cls == 1 ? 0 : Hc - 1);
  if (cls == 0 && Hc < 3) return;
  const int j = e & 7, l = (e >> 3) & 63, f = e >> 9;
  const int half = f & 1, dc = (f >> 1) % 3 - 1, dr = (f >> 1) / 3 - 1;
  const int mp = l & 15, ci = 32 * half + 8 * (l >> 4) + j;
  const int px = mp & 1, q = mp >> 1, py = q / N, n = q % N;
  float v = 0.f;
  if (py < 2) {
    // column tap of parity px at offset dc (circular axis): px 0: (dc 0, kx 1), (dc -1, kx 3); px 1: (dc +1, kx 0), (dc 0, kx 2)
    int kx;
    if (px == 0) kx = dc == 0 ? 1 : (dc == -1 ? 3 : -1);
    else kx = dc == 1 ? 0 : (dc == 0 ? 2 : -1);
    if (kx >= 0) {
      for (int i = 0; i < 4; ++i) {
        int r, ky;
        if (dg_tap1d(MODE_UP, adj, 0, 2 * m + py, Hc, i, r, ky) && r - m == dr) v += w(ky * 4 + kx, n, ci);
      }
    }
  }
  *(bf16*)(frag + (((cls * 18 + f) * 64 + l) * 8 + j) * 2) = LO ? (bf16)(v - (float)(bf16)v) : (bf16)v;
}
