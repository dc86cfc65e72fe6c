// "Big-tile" persistent implicit-GEMM conv on the matrix cores (gfx950, bf16): ONE wave per SIMD, a 128-pixel x 64-channel
// output tile per wave, fragment reads and LDS-DMA software-pipelined under the wave's own matrix instructions.
// Same passes, tile geometry (256 pixels x 128 channels per workgroup, or both column parities x 64 channels), tile order,
// pair stages and epilogue arithmetic as the ping-pong kernel (conv_mfma_pp.hip; reference: models/gans/dcgan_eqlr.py:19-26,
// 75-82 with Pad / EqualLR / FusedLeakyReLU of models/ops/common.py fused) - a different schedule:
//
//   * ping-pong kernel: 8 waves of 64 x 64, the two waves of a SIMD alternate LOAD | MFMA halves of 32 matrix instructions
//     with a workgroup barrier between any two halves.  Round-4 stamps: a wave spends 27 % of its time in those barriers and
//     each half is lengthened by its partner's issue traffic - the matrix pipe is busy ~54 % of the K loop.
//   * here: 4 waves of 128 x 64 (accumulators 128 registers; the 512-register file of a lone wave holds them beside two
//     fragment buffers).  A k-step is 32 matrix instructions with the 12 fragment reads of the NEXT k-step and 4 LDS-DMA pieces
//     of the next pair stage issued between them; ONE s_waitcnt vmcnt(0) + ONE barrier per PAIR (128 matrix instructions),
//     and the last quarter of a pair's matrix instructions runs behind that barrier, over the latency of the next pair's first
//     fragment reads.  scripts/micro/bigtile_loop.hip measured this loop structure (128 x 128 per wave, no epilogue) at 0.76 of
//     the MFMA-only rate on MI355X; the ping-pong K loop runs at ~0.5.
//
// Accumulators are transposed as in the ping-pong kernel (weights are the MFMA A operand with permuted rows, pixels B): a lane
// ends up with 16 consecutive channels of one pixel per 16-pixel block row; the output leaves through a wave-private 2 KB LDS
// strip per block row as whole 128-byte runs.
#include "conv_mfma_persist_impl.h"

#include <type_traits>

namespace bt {

using persist::Geo;
using persist::Tile;

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int CTRL>
__device__ __forceinline__ float row_add(float v) {  // v + (v of the lane CTRL selects inside the 16-lane row)
  const int s = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, s);
}

// MK: 0 = EPI_LRELU / EPI_LINEAR (bias; optional DgConv.mask_out), 1 = EPI_MASK with the slope taken from the saved
// activation (aux), 2 = EPI_MASK with the slope taken from the saved 1-bit masks (DgConv.mask_in).
// DUAL (MODE_UP, 64 output channels): the tile is 256 coarse columns x both column parities x 64 channels (wave column wn =
// the parity), see conv_mfma_pp.hip.
template <int MODE, int MK, bool DUAL>
__global__ __launch_bounds__(256, 1) void conv_bt_kernel(ConvP p, Geo g) {
  static_assert(!DUAL || MODE == MODE_UP, "DUAL: both column parities of a 64-channel MODE_UP layer");
  constexpr bool MASK = MK != 0;
  constexpr int BN = 128;
  constexpr int NCH = DUAL ? 64 : 128;         // real output channels per tile
  constexpr int BM = 256, NWV = 4;
  constexpr int SB = 128;                      // bytes of K per tile row and stage (64 bf16 channels)
  constexpr int WC = 64;                       // channels per wave
  constexpr int TM = 8, TN = 4;                // 16 x 16 blocks per wave: pixels x channels
  constexpr int CPL = 16;                      // consecutive channels one lane ends up with
  constexpr int IMG_ROWS = 264;
  constexpr int AIMG = IMG_ROWS * SB;
  constexpr int BT = BN * SB;
  constexpr int PSTAGE = AIMG + 2 * BT;        // one pair stage: a pixel image of SW + 1 (+ 2) columns per segment + two weight tiles
  constexpr int NPS = 2;
  constexpr int IA = 8, IB = BN / 8 / NWV;     // pieces per wave: image (+ piece 32: wave 0), one weight tile (4)
  constexpr int NDB = 512;
  constexpr int NPAIR = MODE == MODE_S2 ? 2 : 1;
  constexpr int SCR = 16 * WC * 2;             // the wave's output strip: 16 pixels x 64 channels
  constexpr int NHT = 256;
  constexpr int LDS_HT = NPS * PSTAGE + 3 * NDB * 4 + NWV * SCR;
  static_assert(LDS_HT + NHT * 8 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_HT + NHT * 8];

  // ---- this workgroup's tiles: an XCD owns a contiguous range of the tile order, its workgroups walk it round-robin
  //      (conv_mfma_pp.hip)
  const int G = gridDim.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = blockIdx.x & 7;
  const int gi0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int nwx = q8 + (xcd < r8 ? 1 : 0);
  const int tq = g.ntiles / G, tr = g.ntiles % G;
  const int xs = gi0 * tq + (gi0 < tr ? gi0 : tr);
  const int xe = (gi0 + nwx) * tq + (gi0 + nwx < tr ? gi0 + nwx : tr);
  const int t0 = xs + (int)(blockIdx.x >> 3);
  const int tcount = t0 < xe ? (xe - t0 + nwx - 1) / nwx : 0;
  if (tcount == 0) return;

  const int tid = threadIdx.x;
  const int KC = p.K / 64;
  const int Ws = MODE == MODE_S2 ? 2 * p.Wc : p.Wc, cmul = MODE == MODE_S2 ? 2 : 1;
  const int Wo = MODE == MODE_S2 ? p.Wc : 2 * p.Wc;
  const int rows = MODE == MODE_S2 ? p.Hc : 2 * p.Hc;
  const int tiles_n = g.tiles_n, tiles_x = g.tiles_x;

  auto tile_at = [&](int t) {                  // tile order: N tile fastest, then column tile, parity, row, sample group
    Tile r;
    int mt = t / tiles_n;
    r.nt = t % tiles_n;
    r.px = 0;
    r.xt = mt % tiles_x; mt /= tiles_x;
    if (MODE == MODE_UP && !DUAL) { r.px = mt & 1; mt >>= 1; }
    r.Y = mt % rows; r.bt = mt / rows;
    r.nt = __builtin_amdgcn_readfirstlane(r.nt); r.xt = __builtin_amdgcn_readfirstlane(r.xt);
    r.px = __builtin_amdgcn_readfirstlane(r.px); r.Y = __builtin_amdgcn_readfirstlane(r.Y);
    r.bt = __builtin_amdgcn_readfirstlane(r.bt);
    return r;
  };
  const Tile first = tile_at(t0);
  Tile dstep;                                  // the walk from tile t to tile t + nwx as a carry chain
  {
    int r = nwx;
    dstep.nt = r % tiles_n; r /= tiles_n;
    dstep.xt = r % tiles_x; r /= tiles_x;
    dstep.px = 0;
    if (MODE == MODE_UP && !DUAL) { dstep.px = r & 1; r >>= 1; }
    dstep.Y = r % rows; dstep.bt = r / rows;
    dstep.nt = __builtin_amdgcn_readfirstlane(dstep.nt); dstep.xt = __builtin_amdgcn_readfirstlane(dstep.xt);
    dstep.px = __builtin_amdgcn_readfirstlane(dstep.px); dstep.Y = __builtin_amdgcn_readfirstlane(dstep.Y);
    dstep.bt = __builtin_amdgcn_readfirstlane(dstep.bt);
  }
  auto tile_next = [&](Tile& t) __attribute__((always_inline)) {
    int c;
    t.nt += dstep.nt; c = t.nt >= tiles_n; if (c) t.nt -= tiles_n;
    t.xt += dstep.xt + c; c = t.xt >= tiles_x; if (c) t.xt -= tiles_x;
    if (MODE == MODE_UP && !DUAL) { t.px += dstep.px + c; c = t.px >> 1; t.px &= 1; }
    t.Y += dstep.Y + c; c = t.Y >= rows; if (c) t.Y -= rows;
    t.bt += dstep.bt + c;
  };

  const bf16* in = (const bf16*)p.in;
  const bf16* w = (const bf16*)p.w;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lrow = lane >> 3, pos = lane & 7;
  auto swzA = [](int row) { return ((row >> 1) & 3) << 1; };   // conflict-free fragment reads at any row offset (conv_mfma_pp.hip)
  auto swzB = [](int row) { return ((row >> 1) & 1) | (((row / CPL) & 3) << 1); };
  const int pitch = g.SW + (DUAL ? 2 : 1);

  // ---- issue side: per-lane source offsets of this wave's pieces
  unsigned voffA[NPAIR][IA + 1], voffB[IB];
  int colA[IA + 1];
  unsigned sampA[IA + 1];
#pragma unroll
  for (int u = 0; u <= IA; ++u) {
    const int m = (u < IA ? (wave + NWV * u) : 32) * 8 + lrow;
    int seg = m / pitch, c = m - seg * pitch;
    if (seg >= g.NSB) { seg = 0; c = 0; }      // pad rows of the last piece: any valid address
    colA[u] = cmul * c;
    sampA[u] = (unsigned)(seg * (int)p.in_sb * 2 + (pos ^ swzA(m)) * 16);
  }
#pragma unroll
  for (int u = 0; u < IB; ++u) {
    const int row = (wave + NWV * u) * 8 + lrow;   // (DUAL: LDS rows 64-127 are channels 0-63 of the other parity's tap)
    voffB[u] = (unsigned)((DUAL ? row & 63 : row) * (int)p.w_sn * 2 + (pos ^ swzB(row)) * 16);
  }
  const unsigned dst_wave = (unsigned)wave * 1024u;

  // ---- compute side: this lane's fragment addresses inside a pair stage.  A block row (16 pixels) lies inside one sample
  //      segment; the wave's eight block rows may span two (SW = 64).  k-step 1 = ^ 64.
  const int wm = wave >> 1, wn = wave & 1;
  const int a16 = lane & 15, g4 = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // (per tap and 64-row half: the four block rows of a half are consecutive image rows of one segment - SW >= 64 - and a
  //  multiple of 16 rows does not move the swizzle, so they are immediate offsets from one address)
  unsigned pa[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int trow = wm * 128 + 64 * hf;
    const int rowbase = (trow >> g.lsw) * pitch + (trow & (g.SW - 1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = rowbase + a16 + t + (DUAL ? wn : 0);
      pa[t][hf] = (unsigned)(r * SB + ((g4 ^ swzA(r)) << 4));
    }
  }
  const unsigned wrow = (unsigned)(wn * WC + (a16 >> 2) * CPL + (a16 & 3));
  const unsigned wbase = (unsigned)(AIMG + wrow * SB + ((g4 ^ (((a16 >> 1) & 1) | ((a16 >> 2) << 1))) << 4));

  float* s_bias = (float*)(lds + NPS * PSTAGE);
  float* s_db = s_bias + NDB;
  const unsigned sbias0 = lds0 + NPS * PSTAGE, sdb0 = sbias0 + NDB * 4;
  float* s_rs = s_db + NDB;
  const unsigned srs0 = sdb0 + NDB * 4;
  const bool want_db = MASK && p.dbias != nullptr;
  for (int i = tid; i < NDB; i += 64 * NWV) {
    s_bias[i] = (!MASK && p.bias && i < p.N) ? p.bias[i % p.bias_mod] * (p.epi == EPI_LRELU ? SQRT2 : 1.f) : 0.f;
    s_db[i] = 0.f;
    s_rs[i] = (p.rowscale && i < p.B) ? p.rowscale[i] : 1.f;
  }
  const unsigned sht0 = lds0 + LDS_HT;         // H-tap lists of every output row (count in bits 60-62)
  for (int y = tid; y < rows; y += 64 * NWV) {
    unsigned long long hl;
    const int nh = persist::pack_htaps<MODE>(p.adj, y, p.Hc, hl);
    ((unsigned long long*)(lds + LDS_HT))[y] = hl | ((unsigned long long)nh << 60);
  }
  __syncthreads();

  f32x4_t acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  i32x4 fp[2][TM], fw[2][TN];                  // two fragment buffers of one k-step (32 channels) each
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fp[b][i] = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < TN; ++j) fw[b][j] = i32x4{0, 0, 0, 0};
  }

  // ---- epilogue
  bf16* out = (bf16*)p.out;
  const int ostep = MODE == MODE_S2 ? 1 : 2;   // output pixels between consecutive tile rows of a segment
  const unsigned lane_coff = (unsigned)(((DUAL ? 0 : wn * WC) + g4 * CPL) * 2);   // the lane's first channel inside the N tile, bytes
  const unsigned par_off = DUAL ? (unsigned)(wn * (int)p.out_sp * 2) : 0u;
  const unsigned pix_lane = (unsigned)(ostep * a16 * (int)p.out_sp * 2) + lane_coff + par_off;   // accumulator layout: pixel a16 of a block row
  const unsigned pix_lane_m = pix_lane >> 4;   // ... inside a mask buffer (1 bit per element)
  constexpr int CH = WC / 8;                   // 16-byte chunks per pixel in the strip (8)
  constexpr int RPB = 64 / CH;                 // pixels per store instruction (8)
  constexpr int NRD = 16 / RPB;                // store instructions per block row (2)
  auto swzS = [](int px) { return px & 7; };
  const unsigned scr0 = lds0 + NPS * PSTAGE + 3 * NDB * 4 + (unsigned)wave * SCR;
  const unsigned scr_w = scr0 + (unsigned)(a16 * (CH * 16) + (((g4 * NRD) ^ swzS(a16)) << 4));   // chunk h: ^ (h << 4)
  unsigned scr_r, st_lane;
  {
    const int pr = lane / CH, c = lane % CH;
    scr_r = scr0 + (unsigned)(pr * (CH * 16) + ((c ^ swzS(pr)) << 4));                            // store h: + h * 1024
    st_lane = (unsigned)(ostep * pr * (int)p.out_sp * 2 + c * 16) + (DUAL ? par_off : (unsigned)((wn * WC) * 2));
  }
  auto tile_off = [&](const Tile& t) __attribute__((always_inline)) -> long {  // element offset of (sample group, row Y, first column, first channel)
    const int n0 = t.xt * BM;
    return (long)(t.bt * g.NSB) * p.out_sb + ((long)t.Y * Wo + (MODE == MODE_S2 ? n0 : 2 * n0 + t.px)) * p.out_sp +
           t.nt * NCH;
  };
  // byte offset of block row i of this wave inside the tile (wave-uniform)
  auto row_off = [&](int i) __attribute__((always_inline)) -> long {
    const int trow = wm * 128 + 16 * i;
    return ((long)(trow >> g.lsw) * p.out_sb + (long)(ostep * (trow & (g.SW - 1))) * p.out_sp) * 2;
  };
  i32x4 axr[MK == 1 ? 2 * TM : 1];             // aux form: the lane's 16 channels of its pixel, per block row (2 x 16 bytes)
  unsigned mbits[MK == 2 ? TM : 1];            // bits form: 16 mask bits per block row
  auto load_aux = [&](const Tile& t) __attribute__((always_inline)) {
    if constexpr (MK == 2) {
      const char* mb = (const char*)p.mask_in + (tile_off(t) >> 3);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* src = mb + (row_off(i) >> 4);
        asm volatile("global_load_ushort %0, %1, %2" : "=v"(mbits[i]) : "v"(pix_lane_m), "s"(src) : "memory");
      }
    } else if constexpr (MK == 1) {
      const char* ab = (const char*)((const bf16*)p.aux + tile_off(t));
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* src = ab + row_off(i);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(axr[2 * i]) : "v"(pix_lane), "s"(src) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(axr[2 * i + 1]) : "v"(pix_lane), "s"(src) : "memory");
      }
    }
  };
  auto flush_db = [&](float (&dbacc)[CPL], int nt) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0xB1>(dbacc[c]);    // quad_perm [1,0,3,2]
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0x4E>(dbacc[c]);    // quad_perm [2,3,0,1]
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0x124>(dbacc[c]);   // row_ror 4
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0x128>(dbacc[c]);   // row_ror 8
    if (a16 == 0) {
      const unsigned ad = sdb0 + (unsigned)(nt * NCH * 4) + lane_coff * 2;
#pragma unroll
      for (int c = 0; c < CPL; ++c) asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(ad), "v"(dbacc[c]), "n"(c * 4) : "memory");
    }
  };
  // The whole epilogue of a finished tile (all four waves at the same point of the program: between two pair iterations).
  // Arithmetic as in the ping-pong kernel; per block row the 16 channels of the lane's pixel go through the strip and leave
  // as two stores of whole 128-byte runs.
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
#ifdef BT_NOEPI                                  // (ablation builds, `make variant VSRC=conv_mfma_bt VFLAGS=-DBT_NOEPI`: garbage outputs)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"a"(acc[i][j]));
    zero_acc();
    return;
#endif
    const long toff = tile_off(t);
    char* ob = (char*)(out + toff);
    if (MK == 1) {
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i) asm volatile("" : "+v"(axr[i]));
    }
    if (MK == 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(mbits[i]));
    }
    char* mob = nullptr;                         // mask_out: this tile's bits (EPI_LRELU)
    if (!MASK && p.mask_out) mob = (char*)p.mask_out + (toff >> 3);
    f32x4_t bias[TN];
    if (!MASK) {
      const unsigned ba = sbias0 + (unsigned)(t.nt * NCH * 4) + lane_coff * 2;
      // (read and wait in ONE statement: with accumulators in AGPRs hipcc uses the rest of that file as spill space and may
      //  copy a register there right behind the asm that defines it - in front of a separate s_waitcnt, i.e. before the LDS
      //  data has arrived.  First version of this kernel: NaNs in fixed lane groups of the forward layers.)
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                   "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(bias[0]), "=&v"(bias[1]), "=&v"(bias[2]), "=&v"(bias[3]) : "v"(ba) : "memory");
    }
    float dbacc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = 0.f;
    float c_pos = p.scale * SQRT2, c_neg = p.scale * (LRELU_SLOPE * SQRT2);
    asm volatile("" : "+v"(c_pos), "+v"(c_neg));
    const float c_lin = p.epi == EPI_LRELU ? c_pos : p.scale;
    const float slope = p.epi == EPI_LRELU ? LRELU_SLOPE : 1.f;
    i32x4 rd[NRD];
    auto store_row = [&](int i) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const char* rb = ob + row_off(i);
#pragma unroll
      for (int h = 0; h < NRD; ++h) {
        // (wave-uniform base in scalar registers + the lane's 32-bit offset: a C store makes hipcc keep a 64-bit per-lane
        //  address for each of the tile's 16 stores live across the whole kernel)
        const char* base = rb + (long)(ostep * h * RPB) * p.out_sp * 2;
        asm volatile("" : "+v"(rd[h]));          // (the strip read is complete only behind the wait above: no copy of rd may be made in front of it)
        // (s_nop: a store of more than 8 bytes reads its data registers a cycle or two AFTER it issues - the "VMEM store data"
        //  hazard hipcc pads for stores it can see - and the next VALU instruction may well be the one that reuses them: without
        //  it single dwords of the tile came out as the NEXT block row's temporaries in fixed lane groups)
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 2" ::"v"(st_lane), "v"(rd[h]), "s"(base) : "memory");
      }
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float rs = 0.f;                            // per-sample weight of the bias-gradient sums: a block row lies in one sample
      if (want_db) {
        const unsigned ra = srs0 + (unsigned)(t.bt * g.NSB + ((wm * 128 + 16 * i) >> g.lsw)) * 4;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rs) : "v"(ra) : "memory");
      }
      unsigned mrow = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        i32x4 pk;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          float v2[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int c = 8 * h + 2 * e2 + q, j = c >> 2, r = c & 3;
            float av;                            // (explicit AGPR read at the point of use: left to itself hipcc copies all
            asm("v_accvgpr_read_b32 %0, %1" : "=v"(av) : "a"(acc[i][j][r]));   // 128 accumulators to VGPRs up front)
            float v;
            if (MK == 2) {
              int sel;
              float kf;
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(mbits[MK == 2 ? i : 0]), "n"(c));
              asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(kf) : "v"(sel), "v"(c_pos), "v"(c_neg));
              v = av * kf;
              dbacc[c] = fmaf(v, rs, dbacc[c]);
            } else if (MK == 1) {
              const int w32 = axr[MK == 1 ? 2 * i + h : 0][e2];
              const bool posv = q ? w32 > 0xffff : (short)w32 > 0;
              v = av * (posv ? c_pos : c_neg);
              dbacc[c] = fmaf(v, rs, dbacc[c]);
            } else {
              v = fmaf(av, c_lin, bias[j][r]);
              v = fmaxf(v, slope * v);
            }
            v2[q] = v;
          }
          unsigned pw;
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw) : "v"(v2[0]), "v"(v2[1]));
          pk[e2] = (int)pw;
        }
        if (!MASK && mob) {                      // the saved mask of these 8 channels, from the rounded values (conv_mfma_pp.hip)
          unsigned tb[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(tb[e2]) : "v"(pk[e2]));
          const unsigned mm = tb[0] | (tb[1] << 2) | (tb[2] << 4) | (tb[3] << 6);
          const unsigned gb = (mm & 0x55u) | ((mm >> 15) & 0xAAu);
          if (h == 0) mrow = gb; else mrow |= gb << 8;
          if (h == 1) {
            char* dstm = mob + (row_off(i) >> 4);
            asm volatile("global_store_short %0, %1, %2" ::"v"(pix_lane_m), "v"(mrow), "s"(dstm) : "memory");
          }
        }
        if (h == 0 && i > 0) store_row(i - 1);   // (its strip reads were issued a block row of arithmetic ago)
        asm volatile("ds_write_b128 %0, %1" ::"v"(scr_w ^ (unsigned)(h << 4)), "v"(pk) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int h = 0; h < NRD; ++h)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[h]) : "v"(scr_r), "n"(h * 1024) : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    store_row(TM - 1);
    if (want_db) flush_db(dbacc, t.nt);
    zero_acc();
  };

  // ---- main loop
  const long tap_b = (long)p.w_st * 2;         // bytes per weight tap
  const unsigned spb = (unsigned)p.in_sp * 2u; // bytes per input pixel (< 2^24)
  auto dma_s = [&](unsigned voff, const char* sbase, unsigned ldsbase, auto off_tag) __attribute__((always_inline)) {
    constexpr int OFF = decltype(off_tag)::value;
#ifdef BT_NODMA
    return;
#endif
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 ::"s"(ldsbase), "v"(voff), "s"(sbase), "n"(OFF) : "memory", "scc");
  };
  unsigned so_c = PSTAGE, so_i = 0;            // LDS offsets of the pair stage read / refilled (swapped per pair)

  // piece q of the pair being issued (this wave's 16, + the image's 33rd piece: wave 0, q == 16)
  struct Src { const char *sA, *sB0, *sB1, *sB0b, *sB1b; int pi; };
  unsigned vcur[IA + 1];                       // the image offsets of the pair being issued (voffA[pair])
  auto piece = [&](auto q_tag, const Src& s) __attribute__((always_inline)) {
    constexpr int q = decltype(q_tag)::value;
    const unsigned dst = lds0 + so_i + dst_wave;
    if constexpr (q < IA) dma_s(vcur[q], s.sA, dst, std::integral_constant<int, NWV * q * 1024>{});
    else if constexpr (q < IA + IB) {
      constexpr int u = q - IA;                  // (DUAL: the wave's pieces u = 2, 3 are rows 64-127: parity 1's tap)
      dma_s(voffB[u], (DUAL && u >= IB / 2) ? s.sB0b : s.sB0, dst, std::integral_constant<int, AIMG + NWV * u * 1024>{});
    } else if constexpr (q < IA + 2 * IB) {
      constexpr int u = q - IA - IB;
      dma_s(voffB[u], (DUAL && u >= IB / 2) ? s.sB1b : s.sB1, dst, std::integral_constant<int, AIMG + BT + NWV * u * 1024>{});
    } else {
      if (wave == 0) dma_s(vcur[IA], s.sA, lds0 + so_i, std::integral_constant<int, 32 * 1024>{});
    }
  };
  // the fragment reads of k-step (tap t, ks) of the stage at LDS offset `so` into fragment buffer `buf`: three addresses
  // (pixel rows 0-63 / 64-127 of the wave, weight rows) + immediate offsets; read r = 0 .. TM + TN - 1
  unsigned ra_[3];
  auto rdaddr = [&](auto t_tag, auto ks_tag, unsigned so) __attribute__((always_inline)) {
    constexpr int t = decltype(t_tag)::value, ks = decltype(ks_tag)::value;
    ra_[0] = lds0 + so + (ks ? (pa[t][0] ^ 64u) : pa[t][0]);
    ra_[1] = lds0 + so + (ks ? (pa[t][1] ^ 64u) : pa[t][1]);
    ra_[2] = lds0 + so + (ks ? ((wbase ^ 64u) + t * BT) : (wbase + t * BT));
  };
  auto rdfrag = [&](auto buf_tag, auto r_tag) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_tag)::value, r = decltype(r_tag)::value;
#ifdef BT_NOREADS
    return;
#endif
    if constexpr (r < TM) {
      i32x4& dst = fp[buf][r];                   // (a plain use: an asm operand alone does not make a generic lambda capture)
      const unsigned a = ra_[r / 4];
      LDS_READ128(dst, a, (r & 3) * 16 * SB);
    } else {
      constexpr int j = r - TM;
      i32x4& dst = fw[buf][j];
      const unsigned a = ra_[2];
      LDS_READ128(dst, a, j * 4 * SB);
    }
  };
  auto pin = [&](auto buf_tag) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_tag)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TM; ++i) { i32x4& f = fp[buf][i]; asm volatile("" : "+v"(f)); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { i32x4& f = fw[buf][j]; asm volatile("" : "+v"(f)); }
  };
  constexpr int NMF = TM * TN, NRF = TM + TN;  // 32 matrix instructions and 12 fragment reads per k-step
  // One k-step: the NMF matrix instructions on buffer CUR; between them the fragment reads of the next k-step (tap NT, k-step
  // NK of the same stage) into the other buffer and pieces Q0 .. Q0 + 3 of the pair being issued.
  // HOLD > 0 (the pair's last k-step): the last HOLD matrix instructions run behind `handoff()` - the pair's vmcnt(0) +
  // barrier - and behind the first fragment reads of the NEXT pair (tap 0, k-step 0 of the stage just filled).
  auto kstep = [&](auto cur_tag, auto nt_tag, auto nk_tag, auto q0_tag, auto hold_tag, auto iss_tag,
                   const Src& s) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_tag)::value, NT = decltype(nt_tag)::value, NK = decltype(nk_tag)::value;
    constexpr int Q0 = decltype(q0_tag)::value, HOLD = decltype(hold_tag)::value;
    constexpr bool iss = decltype(iss_tag)::value;   // (compile time: a run-time flag put a branch in front of every instruction)
    constexpr int NB = NMF - HOLD;               // matrix instructions in front of the hand-off
    auto reads_between = [&](auto n_tag) __attribute__((always_inline)) {   // the reads due in front of matrix instruction n
      constexpr int n = decltype(n_tag)::value;
      // all twelve reads go out under the FIRST half of the k-step's matrix instructions: the k-step ends with a wait for
      // them, and a read issued under the last instructions would expose its whole LDS latency (~200 cycles of the 512)
      constexpr int RW = NB / 2;
      constexpr int lo = n < RW ? n * NRF / RW : NRF, hi = n < RW ? (n + 1) * NRF / RW : NRF;
      if constexpr (HOLD == 0 && n == 0) rdaddr(std::integral_constant<int, NT>{}, std::integral_constant<int, NK>{}, so_c);
      if constexpr (HOLD == 0 && hi > lo) {
        rdfrag(std::integral_constant<int, 1 - CUR>{}, std::integral_constant<int, lo>{});
        if constexpr (hi > lo + 1) rdfrag(std::integral_constant<int, 1 - CUR>{}, std::integral_constant<int, lo + 1>{});
      }
    };
    auto one = [&](auto n_tag) __attribute__((always_inline)) {
      constexpr int n = decltype(n_tag)::value;
      constexpr int i = n / TN, j = n % TN;
      if constexpr (HOLD > 0 && n == NB) {
        // the pair's hand-off: this wave's share of the stage just issued has landed, everyone's after the barrier, and
        // everyone is done reading the stage the next pair overwrites; then the next pair's first fragments
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rdaddr(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, so_i);
#define BT_RD(R) rdfrag(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{});
        BT_RD(0) BT_RD(1) BT_RD(2) BT_RD(3) BT_RD(4) BT_RD(5) BT_RD(6) BT_RD(7) BT_RD(8) BT_RD(9) BT_RD(10) BT_RD(11)
#undef BT_RD
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        // (inline asm with the accumulator tied to the AGPR class: the epilogue reads accumulators through "a" operands, and with
        //  the builtin hipcc then moved every accumulator between AGPR ranges and VGPR copies once per pair - 128
        //  v_accvgpr_read + 128 v_accvgpr_write beside 128 matrix instructions, 2.5x the loop's time)
        f32x4_t& c_ = acc[i][j];
        const i32x4& a_ = fw[CUR][j];
        const i32x4& b_ = fp[CUR][i];
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c_) : "v"(a_), "v"(b_));
      }
      if constexpr (n < NB) {
        reads_between(std::integral_constant<int, n>{});
        // the pair's 16 pieces go out during its FIRST two k-steps (8 each, one per four matrix instructions): what bounds this
        // loop is the pieces' latency from L2 / HBM against the vmcnt(0) at the end of the pair, so they are issued as early as
        // the stage is free (spread over all four k-steps the last pieces had a quarter of a pair to land)
        constexpr int NPQ = Q0 < 16 ? 8 : 0;
        constexpr int PSTEP = NB / (NPQ > 0 ? NPQ : 1);
        if constexpr (iss && NPQ > 0 && n % PSTEP == PSTEP / 2) piece(std::integral_constant<int, Q0 + n / PSTEP>{}, s);
        if constexpr (iss && Q0 == 0 && n == 1) piece(std::integral_constant<int, IA + 2 * IB>{}, s);   // (the image's 33rd piece: wave 0)
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#define BT_ONE(N) one(std::integral_constant<int, N>{});
    BT_ONE(0) BT_ONE(1) BT_ONE(2) BT_ONE(3) BT_ONE(4) BT_ONE(5) BT_ONE(6) BT_ONE(7) BT_ONE(8) BT_ONE(9) BT_ONE(10) BT_ONE(11)
    BT_ONE(12) BT_ONE(13) BT_ONE(14) BT_ONE(15) BT_ONE(16) BT_ONE(17) BT_ONE(18) BT_ONE(19) BT_ONE(20) BT_ONE(21) BT_ONE(22)
    BT_ONE(23) BT_ONE(24) BT_ONE(25) BT_ONE(26) BT_ONE(27) BT_ONE(28) BT_ONE(29) BT_ONE(30) BT_ONE(31)
#undef BT_ONE
    pin(std::integral_constant<int, (HOLD > 0 ? 0 : 1 - CUR)>{});
  };
  // One pair: the four k-steps (tap 0 ks 0, tap 0 ks 1, tap 1 ks 0, tap 1 ks 1) of the pair stage at so_c, while the pieces
  // of the next pair go into so_i.  On entry buffer 0 holds (tap 0, ks 0) of so_c; on exit it holds (tap 0, ks 0) of the
  // stage just filled, and the two stages have swapped roles.
  auto pair_iter = [&](auto iss_tag, const Src& s) __attribute__((always_inline)) {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    kstep(I0{}, I0{}, I1{}, std::integral_constant<int, 0>{}, I0{}, iss_tag, s);                  // t0 k0 | reads t0 k1 | pieces 0-7 (+ 16)
    kstep(I1{}, I1{}, I0{}, std::integral_constant<int, 8>{}, I0{}, iss_tag, s);                  // t0 k1 | reads t1 k0 | pieces 8-15
    kstep(I0{}, I1{}, I1{}, std::integral_constant<int, 16>{}, I0{}, iss_tag, s);                 // t1 k0 | reads t1 k1
    kstep(I1{}, I0{}, I0{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 16>{}, iss_tag, s);  // t1 k1 | hand-off
    const unsigned sw_ = so_c; so_c = so_i; so_i = sw_;
  };
  // The very first pair of the workgroup: nothing to compute yet - issue its pieces, hand off, first fragments.
  auto first_pair = [&](const Src& s) __attribute__((always_inline)) {
#define BT_PC(Q) piece(std::integral_constant<int, Q>{}, s);
    BT_PC(0) BT_PC(1) BT_PC(2) BT_PC(3) BT_PC(4) BT_PC(5) BT_PC(6) BT_PC(7) BT_PC(8) BT_PC(9) BT_PC(10) BT_PC(11) BT_PC(12)
    BT_PC(13) BT_PC(14) BT_PC(15) BT_PC(16)
#undef BT_PC
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rdaddr(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, so_i);
#define BT_RD(R) rdfrag(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{});
    BT_RD(0) BT_RD(1) BT_RD(2) BT_RD(3) BT_RD(4) BT_RD(5) BT_RD(6) BT_RD(7) BT_RD(8) BT_RD(9) BT_RD(10) BT_RD(11)
#undef BT_RD
    pin(std::integral_constant<int, 0>{});
    const unsigned sw_ = so_c; so_c = so_i; so_i = sw_;
  };

  auto refetch0 = [&]() __attribute__((always_inline)) {   // buffer 0 <- (tap 0, k-step 0) of the stage about to be computed
    rdaddr(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, so_c);
#define BT_RD(R) rdfrag(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{});
    BT_RD(0) BT_RD(1) BT_RD(2) BT_RD(3) BT_RD(4) BT_RD(5) BT_RD(6) BT_RD(7) BT_RD(8) BT_RD(9) BT_RD(10) BT_RD(11)
#undef BT_RD
    pin(std::integral_constant<int, 0>{});
  };
  Tile ti = first, tprev = first;
  bool warm = true;                            // nothing to compute during the very first pair
  bool pending = false;                        // the tile whose last pair was just computed waits for its epilogue
  for (int c = 0; c < tcount; ++c) {
    unsigned long long hl;
    int nh;
    {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
      u32x2_t e;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(sht0 + (unsigned)ti.Y * 8u) : "memory");
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)e.x), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)e.y);
      nh = (int)(hi >> 28) & 7;
      hl = ((unsigned long long)(hi & 0x0fffffffu) << 32) | lo;
    }
    const char* in_t = (const char*)(in + (long)(ti.bt * g.NSB) * p.in_sb);
    const char* w_t = (const char*)(w + (long)(ti.nt * NCH) * p.w_sn);
    const int x0 = cmul * ti.xt * BM;
#pragma unroll
    for (int pi = 0; pi < NPAIR; ++pi) {
      const int cb = MODE == MODE_S2 ? pi - 1 : ((DUAL || ti.px == 0) ? -1 : 0);
#pragma unroll
      for (int u = 0; u <= IA; ++u)              // circular columns: Ws is a power of two (checked by the launcher)
        voffA[pi][u] = __umul24((unsigned)((x0 + colA[u] + cb) & (Ws - 1)), spb) + sampA[u];
    }
    const int kxa = MODE == MODE_S2 ? 0 : (ti.px == 0 ? 3 : 2), kxb = MODE == MODE_S2 ? 2 : (ti.px == 0 ? 1 : 0);
    bool firstg = true;
    for (int h = 0; h < nh; ++h) {
      const int it_r = (int)(hl & 1023) >> 2, it_ky = (int)hl & 3;
      hl >>= 10;
      const char* sA_row = in_t + (long)it_r * Ws * spb;
      const char* sB_row = w_t + (long)(it_ky * 4) * tap_b;
      for (int kc = 0; kc < KC; ++kc) {
        const char* sA_k = sA_row + kc * SB;
        const char* sB_k = sB_row + kc * SB;
#pragma unroll 1
        for (int pi = 0; pi < NPAIR; ++pi) {     // (ONE copy of the pair body: the two pairs of MODE_S2 differ in offsets only)
          const Src s{sA_k, sB_k + (long)(kxa + pi) * tap_b, sB_k + (long)(kxb + pi) * tap_b, sB_k + 2 * tap_b, sB_k, pi};
#pragma unroll
          for (int u = 0; u <= IA; ++u) vcur[u] = (NPAIR == 2 && pi == 1) ? voffA[NPAIR - 1][u] : voffA[0][u];
          // the first pair of a tile still computes the LAST pair of the previous one: that tile's mask source is requested
          // in front of this pair's pieces (its vmcnt(0) covers it) and its epilogue follows the pair
          if (warm) first_pair(s);
          else {
            if (firstg && MASK) load_aux(tprev);
            pair_iter(std::true_type{}, s);
            if (firstg) {
              // (the fragments the hand-off read are dropped over the epilogue - 48 registers it needs - and read again)
              epilogue(tprev);
              refetch0();
            }
          }
          firstg = false;
          warm = false;
        }
      }
    }
    tprev = ti;
    pending = true;
    tile_next(ti);
  }
  if (pending) {
    const Src none{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (MASK) load_aux(tprev);
    pair_iter(std::false_type{}, none);        // the last pair of the last tile (its hand-off reads nothing anybody uses)
    epilogue(tprev);
  }
  if (want_db) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int n = tid; n < p.N; n += 64 * NWV) atomicAdd(&p.dbias[n % p.bias_mod], s_db[n]);
  }
}

template <int MODE, bool MASK, bool DUAL>
int launch(const ConvP* p, const Geo& g0, hipStream_t stream, int wg_cap, DgConvPlan* plan) {
  Geo g = g0;
  static int resident = 0;
  if (!resident) {
    int dev = 0, cus = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    HIP_CHECK_RET(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    resident = cus;                            // ~150 KB of LDS and 512 registers per lane: one workgroup per CU
  }
  g.dbg = 0;
  const bool bits_ok = p->out_sn == 1 && p->out_sb % 16 == 0 && p->out_sp % 16 == 0 && p->N % 16 == 0;
  if (!MASK && p->mask_out && !bits_ok) return DG_EUNSUPPORTED;
  const bool bits = MASK && p->mask_in && bits_ok;
  if (MASK && !bits) return DG_EUNSUPPORTED;   // (the aux form - 64 more registers per lane in the epilogue - stays on the ping-pong kernel)
  const int cap = (wg_cap > 0 && wg_cap < resident) ? wg_cap : resident;
  const int G = g.ntiles < cap ? g.ntiles : cap;
  if (plan) {
    plan->family = 7; plan->bm = DUAL ? 512 : 256; plan->bn = DUAL ? 64 : 128; plan->tiles = g.ntiles; plan->workgroups = G;
    plan->tiles_per_wg = (g.ntiles + G - 1) / G;
    plan->mask_bits = MASK ? (bits ? 2 : 0) : 1;
    return DG_OK;
  }
  if constexpr (MASK) conv_bt_kernel<MODE, 2, DUAL><<<(unsigned)G, 256, 0, stream>>>(*p, g);
  else conv_bt_kernel<MODE, 0, DUAL><<<(unsigned)G, 256, 0, stream>>>(*p, g);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

}  // namespace bt

// bf16 layers that tile into 256 pixels x 128 channels (or 512 pixels x 64 channels, both parities of a MODE_UP layer);
// DG_EUNSUPPORTED otherwise (the caller falls back to the ping-pong kernel and its 64-channel tile).
int dg_conv_mfma_bt_launch(const ConvP* p, hipStream_t stream, int min_tiles, int wg_cap, DgConvPlan* plan) {
  if (p->mode != MODE_S2 && p->mode != MODE_UP) return DG_EUNSUPPORTED;
  if (p->in_dtype != DG_BF16 || p->out_dtype != DG_BF16 || p->w_dtype != DG_BF16) return DG_EUNSUPPORTED;
  if (p->K % 64 != 0 || !p->ring || p->nscale) return DG_EUNSUPPORTED;
  if (p->in_sk != 1 || p->w_sk != 1 || p->out_sn != 1) return DG_EUNSUPPORTED;
  if (p->out_sp % 8 != 0 || p->in_sp % 8 != 0 || p->w_sn % 8 != 0) return DG_EUNSUPPORTED;   // 16-byte pieces
  if (p->N > 512 || (p->bias && p->bias_mod < p->N && p->N % p->bias_mod != 0)) return DG_EUNSUPPORTED;
  if (p->dbias && p->bias_mod < p->N) return DG_EUNSUPPORTED;
  const int Ws = p->mode == MODE_S2 ? 2 * p->Wc : p->Wc;
  if ((Ws & (Ws - 1)) != 0 || p->in_sp * 2 >= (1 << 24)) return DG_EUNSUPPORTED;  // column wrap by mask, 24-bit multiply
  if ((p->mode == MODE_S2 ? p->Hc : 2 * p->Hc) > 256) return DG_EUNSUPPORTED;       // rows of the kernel's H-tap table
  const bool mask = p->epi == EPI_MASK;
  if (mask ? p->bias != nullptr : p->dbias != nullptr) return DG_EUNSUPPORTED;
  if (p->rowscale && p->B > 512) return DG_EUNSUPPORTED;
  persist::Geo g;
  if (p->mode == MODE_UP && p->N % 128 != 0 && p->N % 64 == 0 && persist::make_geo<256, 64>(p, g) && g.SW >= 64 &&
      g.NSB * (g.SW + 2) <= 264 && g.ntiles / 2 >= min_tiles) {
    g.ntiles /= 2;                             // one tile = both column parities of 256 coarse columns
    return mask ? bt::launch<MODE_UP, true, true>(p, g, stream, wg_cap, plan)
                : bt::launch<MODE_UP, false, true>(p, g, stream, wg_cap, plan);
  }
  if (!(p->N % 128 == 0 && persist::make_geo<256, 128>(p, g) && g.SW >= 64 && g.ntiles >= min_tiles)) return DG_EUNSUPPORTED;
  if (p->mode == MODE_S2)
    return mask ? bt::launch<MODE_S2, true, false>(p, g, stream, wg_cap, plan) : bt::launch<MODE_S2, false, false>(p, g, stream, wg_cap, plan);
  return mask ? bt::launch<MODE_UP, true, false>(p, g, stream, wg_cap, plan) : bt::launch<MODE_UP, false, false>(p, g, stream, wg_cap, plan);
}
