// "Ping-pong" persistent implicit-GEMM conv on the matrix cores (gfx950, bf16): the kernel behind the fat layers of the
// benchmark step.  Same passes as conv_mfma_persist_impl.h (Down forward / R1 tangent, Up forward, both backward-data
// passes; reference: models/gans/dcgan_eqlr.py:19-26,75-82 with Pad / EqualLR / FusedLeakyReLU of models/ops/common.py
// fused), same tile order (walked round-robin by the workgroups of an XCD, see the kernel), same LDS-DMA issue side - a
// different compute side.
//
// Why: ablation builds of the previous kernel (scripts/bench_conv.py, DG_CONV_DBG) showed its time to be the SUM of its
// parts - skeleton (LDS fragment reads + barriers) 41 %, MFMA 22 %, DMA issue 19 %, epilogue 18 % - i.e. nothing
// overlapped: its 8 waves run in lock step (one barrier per K step), so both waves of a SIMD read fragments, issue DMA
// and feed the matrix pipe at the same moments.  Here the two waves of every SIMD are in OPPOSITE phases
// (cdna_hip_programming.md §5 "8-phase" template; MI355X_MICROARCH.md "Two waves per SIMD" item 9):
//
//     wave group A (waves 0-3):  LOAD(t) | MFMA(t) | LOAD(t+1) | MFMA(t+1) | ...
//     wave group B (waves 4-7):          | LOAD(t) | MFMA(t)   | LOAD(t+1) | ...        ( | = workgroup barrier )
//
//   LOAD(t): all 16 fragment reads of K step t (ds_read_b128, 64 VGPRs) alternated with the step's LDS-DMA pieces ;
//            s_waitcnt {everything but this half's pieces landed, fragments in} ; (epilogue of the finished tile, if any)
//   MFMA(t): 32 x v_mfma_f32_16x16x32_bf16 straight from registers - no LDS, no waits.
// Both groups run the same program; group B starts one barrier late.
//
// Ring stages hold a PAIR of K steps: the two W taps (kx, kx + 2) of a pair read the same input pixels one column apart,
// so a stage is one pixel image of SW + 1 columns per sample segment (33 pieces of 8 rows) plus the two taps' weight
// tiles (2 x 65 KB for the 128-channel N tile); tap t reads image rows r + t.  The compute side runs one pair behind the
// issue side; a stage is refilled one pair after its last read (each LOAD half ends with lgkmcnt(0) in front of a
// barrier).  Details at the issue side and at `pair_iter` below.
//
// Accumulators are TRANSPOSED (weights as the A operand, pixels as B): a lane then holds 4 consecutive channels of one
// pixel per 16 x 16 block, and with the weight rows of block j permuted (row 16g+4j+r feeds lane group g, register r)
// 16 CONSECUTIVE channels of its pixel: the epilogue's arithmetic and its mask source need no cross-lane movement.  The
// output leaves through a wave-private 2 KB LDS strip per 16-pixel block row (128-channel N tile: whole 128-byte runs per
// store instruction) or straight from registers (64-channel N tile: the four lanes of a pixel already cover its 64 bytes).
#include "conv_mfma_persist_impl.h"

#include <type_traits>

namespace pp {

using persist::Geo;
using persist::Tile;

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define PP_WAIT(vm) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(vm) : "memory")

// H-tap lists of every output row (up to 6 taps of 10 bits, the count in bits 60-62), built by the LAUNCHER and passed as a
// kernel argument: a tile reads its row's entry with one scalar load.  (Rounds 4-5 built this table in LDS in the kernel's
// prologue - six rounds of boundary cases per row on one wave while the other seven waited at the barrier behind it: a good part
// of the 2 500 cycles every launch spent before its first LDS-DMA piece, profiles/r05a_conv_lifetime_phases_b32.txt.)
constexpr int NHT = 256;                       // rows of the table (the launcher checks rows <= NHT)
struct HTab { unsigned long long e[NHT]; };

// Diagnostics are compiled in only with -DDG_PP_DIAG=<bits> (make diag DIAGBITS=<bits>, default 8): every runtime check in the K-step loop costs issue
// slots the loop does not have (the LOAD half is instruction-issue bound: ~6-7 cycles per instruction beside the partner
// wave's MFMAs).  Bits as in the lock-step kernel's DG_CONV_DBG (1 no DMA, 2 no MFMA, 4 no epilogue, 16 no
// fragment reads, 32 half of the pixel-tile DMA, 64 no mask-source loads, 128 no output stores); 8 adds shader-clock stamps around the segments of a K step, summed per wave and
// written over the first bytes of the OUTPUT by workgroup 0 (scripts/bench_conv.py prints them; the output is garbage)
__device__ __forceinline__ unsigned long long pp_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <int CTRL>
__device__ __forceinline__ float row_add(float v) {  // v + (v of the lane CTRL selects inside the 16-lane row)
  const int s = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, s);
}

// MASK: the backward / tangent flavour (EPI_MASK: multiply by the saved activation's slope, optional bias-gradient sums,
// no bias); otherwise EPI_LRELU / EPI_LINEAR with bias.  Two instantiations keep either epilogue's registers out of the
// other (both in one kernel spilled past 256 VGPRs, and a scratch reload next to LDS-DMA costs a vmcnt(0) drain).
// DUAL (MODE_UP, 64 output channels): the tile is 256 coarse columns x BOTH column parities x 64 channels - the "N"
// dimension of the 128-row weight tile is (parity, channel): rows 0-63 hold parity 0's W tap, rows 64-127 parity 1's, and
// wave column wn IS the parity (its pixel fragments sit one image row further).  One pixel image of SW + 2 columns then
// feeds four W taps instead of two: the 64-channel MODE_UP layers (Down2 backward-data, Up3 forward) were bound by the
// LDS-DMA issue of their LOAD halves - 6 pieces per wave and pair for 32 MFMAs against 8 for 64 in the 128-channel tile.
// BITS (MASK only): the slope comes from the saved 1-bit masks (DgConv.mask_in, 2 bytes per lane and block row) instead of the
// saved activation itself (aux, 32 bytes per lane and block row, and 32 VGPRs to hold a tile's worth of it).
// X2 (round 5, the fp32x3 precision mode on THIS kernel): input, weights, output and mask source are DG_BF16X2 - per 64
// channels 128 bytes of hi = bf16(x) followed by 128 bytes of lo = bf16(x - hi) (include/dusty_gan_hip.h).  A 64-channel K
// chunk then is three K steps on the same stages and fragments, x_hi w_hi + x_lo w_hi + x_hi w_lo - the chunk's source
// addresses move by 128 bytes, nothing else in the loop changes - and the epilogue stores both halves of its fp32 values
// (straight from the accumulator layout: 64-byte runs).  The launcher hands over strides in bf16 units (twice the elements).
template <int BN, int MODE, bool MASK, bool DUAL = false, bool BITS = false, bool X2 = false>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(ConvP p, Geo g, HTab ht) {
  static_assert(MASK || !BITS, "BITS: a flavour of the EPI_MASK epilogue");
  static_assert(!X2 || !BITS, "X2: mask source = the saved activation's hi half");
  static_assert(!DUAL || (MODE == MODE_UP && BN == 128), "DUAL: both column parities of a 64-channel MODE_UP layer");
  constexpr int NCH = DUAL ? BN / 2 : BN;      // real output channels per tile
  constexpr int BM = 256, NWV = 8, WN = 2;
  constexpr int SB = 128;                      // bytes of K per tile row and stage (64 bf16 channels)
  constexpr int WC = BN / WN;                  // channels per wave
  constexpr int TM = 4, TN = WC / 16;          // 16 x 16 blocks per wave: pixels x channels
  constexpr int CPL = 4 * TN;                  // consecutive channels one lane ends up with
  // One ring stage = one PAIR of K steps: the two W taps of a pair read the SAME input pixels one column apart, so
  // the pair shares one pixel image of SW + 1 columns per sample segment (<= 260 rows, 33 pieces of 8 rows) and owns
  // two weight tiles.  (With one 256-row pixel tile per K step the LDS-DMA of the pixels was issued and written twice:
  // a what-if build that dropped half of those pieces ran the layer set 11 % faster.)
  constexpr int IMG_ROWS = 264;
  constexpr int AIMG = IMG_ROWS * SB;
  constexpr int BT = BN * SB;
  constexpr int PSTAGE = AIMG + 2 * BT;
  constexpr int NPS = 2;
  constexpr int IA = 4, IB = BN / 8 / NWV;     // pieces per wave: image (+ piece 32: wave 0), one weight tile
  constexpr int NST = TM * CPL * 2 / 16;       // 16-byte stores per lane per tile
  constexpr int NDB = 512;
  constexpr int NPAIR = MODE == MODE_S2 ? 2 : 1;   // pairs per (H tap, 64-channel chunk)
  constexpr bool STRIP = BN == 128 && !X2;     // output stores through a per-wave LDS transpose strip (below)
  constexpr int SCR = STRIP ? 16 * WC * 2 : 0; // 16 pixels x the wave's channels
  // Bias-gradient sums (EPI_MASK): one ROW of accumulators per wave that can touch a channel (waves that differ only in
  // their pixel block wm - in the both-parities tile all eight), NDBR * NDB floats laid over [s_bias (unused under EPI_MASK) |
  // s_db | NDBX more]: a wave adds to its own row in its own program order and the rows are summed in a fixed order at the
  // end, so a workgroup's partial sums do not depend on which wave's ds_add arrived first (round 5; one shared row before)
  constexpr int NDBR = 4;                      // rows of NDB floats available (row stride = p.N, rows = DBW: DBW * N <= NDBR * NDB)
  constexpr int NDBX = MASK ? (NDBR - 2) * NDB : 0;
  constexpr int DBW = DUAL ? NWV : NWV / WN;   // waves that share a channel
  constexpr int LDS_HT = NPS * PSTAGE + 3 * NDB * 4 + NDBX * 4 + NWV * SCR;
  static_assert(IB >= 1 && LDS_HT <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_HT];
#ifdef DG_PP_DIAG
  constexpr int dbg = DG_PP_DIAG;              // compile-time bit mask (make diag DIAGBITS=..): no runtime checks
#else
  constexpr int dbg = 0;
#endif

  // ---- this workgroup's tiles.  XCD x (blocks x, x + 8, ...) owns a CONTIGUOUS range of the tile order (the split of
  // the lock-step kernel, conv_mfma_persist_impl.h) and its workgroups walk that range ROUND-ROBIN: at any moment the
  // XCD works on ~32 consecutive tiles - the N tiles, the two column parities and the neighbouring rows of the same
  // input window - so one L2 fill serves them all (a contiguous chunk per workgroup kept those re-uses a whole chunk
  // apart in time, by when the 4 MB L2 had been streamed through several times).
  const int G = gridDim.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = blockIdx.x & 7;
  const int gi0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int nwx = q8 + (xcd < r8 ? 1 : 0);     // workgroups on this XCD
  const int tq = g.ntiles / G, tr = g.ntiles % G;
  const int xs = gi0 * tq + (gi0 < tr ? gi0 : tr);
  const int xe = (gi0 + nwx) * tq + (gi0 + nwx < tr ? gi0 + nwx : tr);
  const int t0 = xs + (int)(blockIdx.x >> 3);
  const int tcount = t0 < xe ? (xe - t0 + nwx - 1) / nwx : 0;
  if (tcount == 0) {                           // (never with the launcher's grid; a zero row if it ever happens)
    if (MASK && p.dbias && p.dbias_part)
      for (int n = threadIdx.x; n < p.N; n += 64 * NWV) p.dbias_part[(long)blockIdx.x * p.N + n] = 0.f;
    return;
  }

  const unsigned long long tk0 = (dbg & 8) ? pp_stamp() : 0ull;   // (diagnostic build: phases of the workgroup's lifetime)
  // ... and the constant 100 MHz counter beside the shader-clock stamp: (tk2 - tk0) / (tr2 - tr0) x 100 MHz is the clock the
  // chip HELD while this workgroup ran (MI355X_MICROARCH.md, DVFS give-back, check 6; scripts/conv_clock.py)
  const unsigned long long tr0 = (dbg & 8) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
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
  // The walk from tile t to tile t + nwx as a carry chain over (N tile, column tile, parity, row, sample group): the
  // stride's digits are worked out once; per tile a dozen scalar instructions instead of three integer divisions.  (Round-4
  // stamps: the per-tile prologue - tile_at, the H-tap list, the per-lane offsets - cost 900-2400 cycles per tile and group, and
  // because a group that is not in its LOAD / MFMA pairing holds up its partner at the next barrier, twice that per tile on
  // the kernel's critical path: 8-19 % of the launch.)
  Tile dstep;
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
  const int lrow = lane >> 3, pos = lane & 7;
  // 16-byte chunk c of image row r lives at chunk c ^ swzA(r).  The fragment reads of the two taps start at rows r and
  // r + 1 of segments pitched SW + 1 rows, i.e. at ANY row offset: ((r >> 1) & 3) << 1 keeps the four 16-lane groups of
  // ds_read_b128 conflict-free for every offset (exhaustive check over the linear maps of the row bits; the previous
  // (r >> 1) & 7 is conflict-free for even offsets only).
  auto swzA = [](int row) { return ((row >> 1) & 3) << 1; };
  auto swzB = [](int row) { return ((row >> 1) & 1) | (((row / CPL) & 3) << 1); };

  // ---- issue side.  K steps of a tile run (H tap, 64-channel chunk, pair, tap of the pair).  A pair's image row m of
  //      segment s holds input column cmul (x_seg + m) + cb (circular), cb = -1 / 0 for the odd / even stride-2 pair
  //      and -1 / 0 for column parity 0 / 1 of MODE_UP; tap t of the pair reads image rows m + t:
  //        MODE_S2 pair 0: kx 0, 2 (columns 2x - 1, 2x + 1)   pair 1: kx 1, 3 (columns 2x, 2x + 2)
  //        MODE_UP px 0:   kx 3, 1 (columns x - 1, x)         px 1:   kx 2, 0 (columns x, x + 1)
  //      The per-lane source offsets change with the tile only (voffA[pair][piece]); a K step costs ~20 scalar
  //      instructions.  (A state machine advanced once per K step cost ~80 and 16 branches - SQ_INSTS_SALU was 2.4x
  //      SQ_INSTS_MFMA - and the instruction issue of the LOAD half, not the LDS, the texture path or the matrix
  //      pipe, bounded the kernel.)
  unsigned voffA[NPAIR][IA + 1], voffB[IB];    // per-lane byte offsets (image: per tile and pair, weights: constant)
  int colA[IA + 1];                            // (stride-scaled) image column inside its sample segment
  unsigned sampA[IA + 1];                      // byte offset of the row's sample inside the sample group + swizzled chunk
  {
    const int pitch = g.SW + (DUAL ? 2 : 1);
#pragma unroll
    for (int u = 0; u <= IA; ++u) {
      const int m = (u < IA ? (wave + NWV * u) : 32) * 8 + lrow;
      int seg = m / pitch, c = m - seg * pitch;
      if (seg >= g.NSB) { seg = 0; c = 0; }    // pad rows of the last piece: any valid address
      colA[u] = cmul * c;
      sampA[u] = (unsigned)(seg * (int)p.in_sb * 2 + (pos ^ swzA(m)) * 16);
    }
  }
#pragma unroll
  for (int u = 0; u < IB; ++u) {
    const int row = (wave + NWV * u) * 8 + lrow;   // (DUAL: LDS rows 64-127 are channels 0-63 of the other parity's tap)
    voffB[u] = (unsigned)((DUAL ? row & 63 : row) * (int)p.w_sn * 2 + (pos ^ swzB(row)) * 16);
  }
  const unsigned dst_wave = (unsigned)wave * 1024u;

  // ---- compute side constants
  const int wm = wave >> 1, wn = wave & 1;
  const int a16 = lane & 15, g4 = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // fragment addresses inside a pair stage: image rows of tap 0 / tap 1 (B operand) and permuted weight rows (A
  // operand); k-step 1 = ^ 64.  The wave's 64 tile rows lie in one sample segment (SW >= 64).
  unsigned pbase[2];
  {
    const int rowbase = ((wm * 64) >> g.lsw) * (g.SW + (DUAL ? 2 : 1)) + ((wm * 64) & (g.SW - 1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = rowbase + a16 + t + (DUAL ? wn : 0);
      pbase[t] = (unsigned)(r * SB + ((g4 ^ swzA(r)) << 4));
    }
  }
  const unsigned wrow = (unsigned)(wn * WC + (a16 >> 2) * CPL + (a16 & 3));
  const unsigned wbase = (unsigned)(AIMG + wrow * SB + ((g4 ^ (((a16 >> 1) & 1) | ((a16 >> 2) << 1))) << 4));

  float* s_rs = (float*)(lds + NPS * PSTAGE);  // per-sample weights of the bias-gradient sums (B <= NDB)
  float* s_bias = s_rs + NDB;
  float* s_db = s_bias + NDB;
  const unsigned srs0 = lds0 + NPS * PSTAGE, sbias0 = srs0 + NDB * 4, sdb0 = sbias0 + NDB * 4;
  const bool want_db = MASK && p.dbias != nullptr;
  // per-wave rows when they fit (every layer of the step: N <= 512 with four sharing waves, N = 64 with eight)
  const bool db_rows = MASK && DBW * p.N <= NDBR * NDB;
  const unsigned sdbw0 = db_rows ? sbias0 + (unsigned)((DUAL ? wave : wave / WN) * p.N) * 4u : sdb0;
  for (int i = tid; i < NDB; i += 64 * NWV) {
    s_bias[i] = (!MASK && p.bias && i < p.N) ? p.bias[i % p.bias_mod] * (p.epi == EPI_LRELU ? SQRT2 : 1.f) : 0.f;
    s_db[i] = 0.f;
    s_rs[i] = (p.rowscale && i < p.B) ? p.rowscale[i] : 1.f;
  }
  if (MASK)
    for (int i = tid; i < NDBX; i += 64 * NWV) s_db[NDB + i] = 0.f;
  // (round 5: the same tables by LDS-DMA instead of ordinary loads - no dependent global round trip in front of the first
  //  tile's DMA - measured 400 cycles SLOWER per launch: the loads' latency already sat under the H-tap table's arithmetic,
  //  and a DMA piece costs ~100 cycles to issue; profiles/r05a_conv_lifetime_phases_b32.txt)
  __syncthreads();
  const unsigned long long tk1 = (dbg & 8) ? pp_stamp() : 0ull;

  f32x4_t acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  i32x4 fp[2][TM], fw[2][TN];                  // fragments of one whole K step

  // ---- epilogue pieces
  bf16* out = (bf16*)p.out;
  const unsigned lane_c0 = (unsigned)((DUAL ? 0 : wn * WC) + g4 * CPL);           // this lane's first channel inside the N tile
  const unsigned lane_coff = X2 ? lane_c0 * 2 + ((lane_c0 >> 6) << 7) : lane_c0 * 2;   // ... in bytes of the output row
  const unsigned par_off = DUAL ? (unsigned)(wn * (int)p.out_sp * 2) : 0u;          // DUAL: the wave's column parity, bytes
  const long px_b = (long)(MODE == MODE_S2 ? 1 : 2) * p.out_sp * 2;   // bytes between consecutive tile rows of a segment
  unsigned pix_off;                            // byte offset of this lane's pixel of block row 0 from the tile base
  {
    const int trow = wm * 64 + a16;
    const int sb = trow >> g.lsw, x = trow & (g.SW - 1);
    pix_off = (unsigned)((sb * (int)p.out_sb + (MODE == MODE_S2 ? x : 2 * x) * (int)p.out_sp) * 2) + lane_coff + par_off;
  }
  // Output stores go through a wave-private LDS strip, one 16-pixel block row at a time: a lane owns 16 B pieces of 16
  // DIFFERENT pixels' rows (stride out_sp), so a store straight from the accumulator layout touched 64 cache lines per
  // instruction, 16-32 B each - ablation: those stores were half of the epilogue's cost and the epilogue a quarter of
  // the kernel.  Through the strip a store instruction writes whole 128 B runs of 8 pixels.  (N tile 64: a lane owns ONE
  // 16 B piece per pixel and the four lanes of a pixel already cover a 64 B run - all of the pixel this wave has - so
  // the strip would only add its LDS round trips: direct stores.)
  constexpr int CH = WC / 8;                   // 16-byte chunks per pixel in the strip (8 or 4)
  constexpr int RPB = 64 / CH;                 // pixels per store instruction
  constexpr int NRD = 16 / RPB;                // store instructions per block row (== NST / TM)
  static_assert(NRD == NST / TM, "strip geometry");
  auto swzS = [](int px) { return CH == 8 ? (px & 7) : ((px >> 1) & 3); };
  const unsigned scr0 = lds0 + NPS * PSTAGE + 3 * NDB * 4 + NDBX * 4 + (unsigned)wave * SCR;   // (behind the bias-gradient rows)
  // (block row i / second store h: wave-uniform steps from one per-lane offset - the wave's 64 tile rows lie in one segment)
  const unsigned scr_w = scr0 + (unsigned)(a16 * (CH * 16) + (((g4 * NRD) ^ swzS(a16)) << 4));   // chunk h: ^ (h << 4)
  unsigned scr_r, st_off;
  {
    const int pr = lane / CH, c = lane % CH;
    scr_r = scr0 + (unsigned)(pr * (CH * 16) + ((c ^ swzS(pr)) << 4));                            // store h: + h * 1024
    const int trow = wm * 64 + pr;
    const int sb = trow >> g.lsw, x = trow & (g.SW - 1);
    st_off = (unsigned)((sb * (int)p.out_sb + (MODE == MODE_S2 ? x : 2 * x) * (int)p.out_sp) * 2 + c * 16) +
             (DUAL ? par_off : (unsigned)((wn * WC) * 2));
  }
  static_assert(NRD == 1 || RPB * CH * 16 == 1024, "strip read offset");
  auto tile_off = [&](const Tile& t) __attribute__((always_inline)) -> long {  // element offset of (sample group, row Y, first column, first channel)
    const int n0 = t.xt * BM;
    return (long)(t.bt * g.NSB) * p.out_sb + ((long)t.Y * Wo + (MODE == MODE_S2 ? n0 : 2 * n0 + t.px)) * p.out_sp +
           t.nt * NCH * (X2 ? 2 : 1);
  };
  // The leaky-relu mask source of a tile being finished (EPI_MASK) is fetched during the tile's last MFMA half into
  // registers of its own (one lane's 4 pixels x 16 TN bytes); the epilogue runs at the END of the next LOAD half, behind that
  // half's ordinary wait - which, counting in order, covers these loads too (they were issued before the half's
  // pieces).  (Round-2 finding: with the mask source aliased onto fragment registers the epilogue had to run FIRST in
  // the half behind a vmcnt(0), i.e. drain every LDS-DMA piece in flight once per tile - the EPI_MASK layers stayed at
  // 720-740 TFLOP/s while the others reached 830-1150.)
  i32x4 axr[(MASK && !BITS) ? NST : 1];
  unsigned mbits[BITS ? TM : 1];               // BITS: the lane's CPL mask bits of block row i
  const unsigned pix_off_m = pix_off >> 4;     // the lane's byte offset inside a mask buffer (1 bit per element)
  auto load_aux = [&](const Tile& t) __attribute__((always_inline)) {
    if constexpr (BITS) {
      const char* mb = (const char*)p.mask_in + (tile_off(t) >> 3);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* src = mb + (long)i * px_b;   // 16 tile rows further: 16 px_b bytes of the tensor = px_b bytes of bits
        if (CPL == 16) asm volatile("global_load_ushort %0, %1, %2" : "=v"(mbits[i]) : "v"(pix_off_m), "s"(src) : "memory");
        else asm volatile("global_load_ubyte %0, %1, %2" : "=v"(mbits[i]) : "v"(pix_off_m), "s"(src) : "memory");
      }
    } else {
      const char* ab = (const char*)((const bf16*)p.aux + tile_off(t));
#pragma unroll
      for (int i = 0; i < TM; ++i) {             // wave-uniform base of block row i + the lane's 32-bit offset
        const char* src = ab + (long)(i * 16) * px_b;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(axr[(NST / TM) * i]) : "v"(pix_off), "s"(src) : "memory");
        if (NST / TM == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(axr[2 * i + 1]) : "v"(pix_off), "s"(src) : "memory");
      }
    }
  };
  // Bias-gradient sums (EPI_MASK with dbias): per lane 4 TN channel sums over its pixels, weighted per sample, of the
  // fp32 values BEFORE rounding; reduced over the 16 pixel lanes and added to the LDS accumulators once per tile.
  // (Keeping the sums in registers across the tiles of a workgroup - one N tile - cost 16 persistent VGPRs and measured
  // 5-10 % slower on the MODE_UP layers.)
  auto flush_db = [&](float (&dbacc)[CPL], int nt) __attribute__((always_inline)) {
    // stage by stage over all CPL channels (independent DPP chains fill each other's hazard slots), then ONE predicated
    // block of LDS adds (a branch per channel cost 8-16 exec save/restores per tile)
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0xB1>(dbacc[c]);    // quad_perm [1,0,3,2]
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0x4E>(dbacc[c]);    // quad_perm [2,3,0,1]
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0x124>(dbacc[c]);   // row_ror 4
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = row_add<0x128>(dbacc[c]);   // row_ror 8
    if (a16 == 0) {
      const unsigned ad = sdbw0 + (unsigned)(nt * NCH * 4) + lane_c0 * 4;
#pragma unroll
      for (int c = 0; c < CPL; ++c) asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(ad), "v"(dbacc[c]), "n"(c * 4) : "memory");
    }
  };
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
    char* ob = (char*)(out + tile_off(t));
    if (MASK && !BITS) {                       // (the caller's wait covered the loads: pin the uses behind it)
#pragma unroll
      for (int i = 0; i < NST; ++i) asm volatile("" : "+v"(axr[i]));
    }
    if (BITS) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(mbits[i]));
    }
    char* mob = nullptr;                         // mask_out: this tile's bits (producer side, EPI_LRELU)
    if (!MASK && p.mask_out) mob = (char*)p.mask_out + (tile_off(t) >> 3);
    f32x4_t bias[TN];
    float rs = 0.f;
    if (want_db) {  // the wave's 64 pixels belong to one sample (SW >= 64): one per-sample weight per tile
      const unsigned ra = srs0 + (unsigned)(t.bt * g.NSB + ((wm * 64) >> g.lsw)) * 4;
      asm volatile("ds_read_b32 %0, %1" : "=v"(rs) : "v"(ra) : "memory");
    }
    if (!MASK) {
      const unsigned ba = sbias0 + (unsigned)(t.nt * NCH * 4) + lane_c0 * 4;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bias[j]) : "v"(ba), "n"(j * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bias[j]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(rs));
    }
    // (no pointers into fp / fw / acc anywhere below: an address-taken register array ends up in scratch memory)
    // MASK:  out = acc * (aux > 0 ? scale sqrt2 : 0.2 scale sqrt2)          unpack, compare, select, multiply
    // else:  out = lrelu(acc * scale' + bias') with sqrt2 folded into scale' and the staged bias (lrelu commutes with a
    //        positive factor): fma, multiply, max
    float dbacc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) dbacc[c] = 0.f;
    float c_pos = p.scale * SQRT2, c_neg = p.scale * (LRELU_SLOPE * SQRT2);
    asm volatile("" : "+v"(c_pos), "+v"(c_neg));   // (opaque: else the select is made between constants and every
                                                   //  element pays a second multiply by the scale)
    const float c_lin = p.epi == EPI_LRELU ? c_pos : p.scale;
    const float slope = p.epi == EPI_LRELU ? LRELU_SLOPE : 1.f;          // max(v, 1 v) = v: no select per element
    i32x4 rd[NRD];                               // block row i - 1, read back pixel-major, waiting for its stores
    unsigned mrow = 0;                           // (mask_out) the CPL bits of the block row being made
    auto store_row = [&](int i) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int h = 0; h < NRD; ++h) {
        asm volatile("" : "+v"(rd[h]));
        char* dstp = ob + (long)(i * 16 + h * RPB) * px_b + st_off;
        if (!(dbg & 128)) *(i32x4*)dstp = rd[h]; else asm volatile("" ::"v"(rd[h]));
      }
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // strip: the chunks of block row i are written as they are made, read back pixel-major, and stored after the
      // arithmetic of block row i + 1 (which runs under the read latency)
#pragma unroll
      for (int h = 0; h < NST / TM; ++h) {
        const i32x4 ax = axr[(MASK && !BITS) ? (NST / TM) * i + h : 0];   // mask source of channels 8h .. 8h+7 (MASK, aux form)
        i32x4 pk, pkl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {                 // two channels per 32-bit word
          float v2[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int c = 8 * h + 2 * e2 + q, j = c >> 2, r = c & 3;
            float v;
            if (MASK && BITS) {
              // bit c of the row's mask word -> 0 / -1 (v_bfe_i32) -> one of the two factors (v_bfi_b32) -> multiply
              // (inline asm: from the C form hipcc makes v_and + v_cmp + two wait states + v_cndmask)
              int sel;
              float kf;
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(mbits[i]), "n"(c));
              asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(kf) : "v"(sel), "v"(c_pos), "v"(c_neg));
              v = acc[i][j][r] * kf;
              dbacc[c] = fmaf(v, rs, dbacc[c]);
            } else if (MASK) {
              // bf16 a > 0  <=>  its 16 bits as a signed integer > 0: the halves are compared in place (low half: 16-bit
              // compare of the word's low bits, high half: the word above 0xffff), no unpacking
              const int w32 = ax[e2];
              const bool posv = q ? w32 > 0xffff : (short)w32 > 0;
              v = acc[i][j][r] * (posv ? c_pos : c_neg);
              dbacc[c] = fmaf(v, rs, dbacc[c]);     // (rs = 0 without dbias: no branch per element)
            } else {
              v = fmaf(acc[i][j][r], c_lin, bias[j][r]);
              v = fmaxf(v, slope * v);
            }
            v2[q] = v;
          }
          unsigned pw;                                   // both halves in one conversion (RNE, as (bf16)v)
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw) : "v"(v2[0]), "v"(v2[1]));
          pk[e2] = (int)pw;
          if constexpr (X2) {                            // lo = bf16(v - hi)
            const float l0 = v2[0] - __builtin_bit_cast(float, pw << 16);
            const float l1 = v2[1] - __builtin_bit_cast(float, pw & 0xffff0000u);
            unsigned pl;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pl) : "v"(l0), "v"(l1));
            pkl[e2] = (int)pl;
          }
        }
        if (!MASK && mob) {
          // the saved mask of these 8 channels from the ROUNDED values (what an EPI_MASK pass would test on the stored
          // activation): per packed pair clamp the halves to [0, 1] (negative -> 0, positive -> 1: v_pk_max_i16, v_pk_min_u16),
          // then shift-or the four words together - plain VALU, ~2 instructions per channel.  (v_cmp into VCC + v_addc would
          // be 2 as well, but a VALU write of VCC needs two wait states before a VALU reads it as carry or mask on gfx950.)
          unsigned tb[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(tb[e2]) : "v"(pk[e2]));
          const unsigned mm = tb[0] | (tb[1] << 2) | (tb[2] << 4) | (tb[3] << 6);   // low halves at bits 0,2,4,6, high at 16,18,..
          const unsigned gb = (mm & 0x55u) | ((mm >> 15) & 0xAAu);
          if (h == 0) mrow = gb; else mrow |= gb << 8;
          if (h == NST / TM - 1) {
            char* dstm = mob + (long)i * px_b;
            if (CPL == 16) asm volatile("global_store_short %0, %1, %2" ::"v"(pix_off_m), "v"(mrow), "s"(dstm) : "memory");
            else asm volatile("global_store_byte %0, %1, %2" ::"v"(pix_off_m), "v"(mrow), "s"(dstm) : "memory");
          }
        }
        if (!STRIP) {                                    // N tile 64: the four lanes of a pixel already write one 64 B run
          char* dstp = ob + (long)(i * 16) * px_b + pix_off + h * 16;
          if (!(dbg & 128)) *(i32x4*)dstp = pk; else asm volatile("" ::"v"(pk));
          if constexpr (X2) {
            if (!(dbg & 128)) *(i32x4*)(dstp + 128) = pkl; else asm volatile("" ::"v"(pkl));
          }
          continue;
        }
        if (h == 0 && i > 0) store_row(i - 1);           // (its reads were issued a block row of arithmetic ago)
        asm volatile("ds_write_b128 %0, %1" ::"v"(scr_w ^ (unsigned)(h << 4)), "v"(pk) : "memory");
      }
      if (STRIP) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < NRD; ++h)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[h]) : "v"(scr_r), "n"(h * 1024) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);       // one pixel block at a time: keeps the live ranges (and VGPRs) short
    }
    if (STRIP) store_row(TM - 1);
    if (want_db) flush_db(dbacc, t.nt);
    zero_acc();
  };

  // ---- main loop
  const bool stamps = (dbg & 8) != 0;
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};  // LOAD work, LOAD waits, barrier 1, MFMA half, barrier 2, between steps
  unsigned long long tend = 0, tkf = 0;

  if (wave >= 4) __builtin_amdgcn_s_barrier(); // group B runs one barrier behind group A
  unsigned so_c = PSTAGE, so_i = 0;            // LDS offsets of the pair stage read / refilled (two stages, swapped per pair)
  bool pending = false;                        // a finished tile waits for its epilogue
  bool warm = true;                            // the very first pair: nothing to compute yet
  Tile tprev = first;                          // the tile the compute side is finishing / has finished
  const long tap_b = (long)p.w_st * 2;         // bytes per weight tap
  const unsigned spb = (unsigned)p.in_sp * 2u; // bytes per input pixel (< 2^24)

  // one LDS-DMA piece: wave-uniform 64-bit base + per-lane 32-bit offset -> LDS at M0 = stage base + constant (written in
  // the same statement; nothing else in this kernel uses M0)
  auto dma_s = [&](unsigned voff, const char* sbase, unsigned ldsbase, auto off_tag) __attribute__((always_inline)) {
    constexpr int OFF = decltype(off_tag)::value;
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 ::"s"(ldsbase), "v"(voff), "s"(sbase), "n"(OFF) : "memory", "scc");
  };

  // One pair = two K steps (the two W taps that share a pixel image), each
  //   LOAD half (fragment reads of the compute step alternated with the step's DMA; waits; epilogue of a finished
  //   tile) | barrier | MFMA half | barrier
  // The compute side runs ONE PAIR behind the issue side: while pair q is issued into one stage, pair q - 1 is read
  // from the other.  Tap 0's LOAD half issues the whole image and the weight tile of tap 0, tap 1's the weight tile of
  // tap 1; each LOAD half ends with "everything but this half's pieces has landed", so the image and tile 0 have a full
  // interval to land before the stage is read and tile 1 one and a half.  A stage is refilled one pair after its last
  // read (each LOAD half ends with lgkmcnt(0) in front of a barrier).
  // first: the pair opens a tile (or is the drain pair after the last one): its compute steps close the previous tile.
  // sB0 / sB1: weight tap of K step 0 / 1 of the pair; DUAL: sB0b / sB1b = the taps of parity 1 (rows 64-127 of the tile)
  auto pair_iter = [&](auto iss_tag, const bool first, const int pi, const char* sA, const char* sB0, const char* sB1,
                       const char* sB0b, const char* sB1b) __attribute__((always_inline)) {
    constexpr bool ISS = decltype(iss_tag)::value;
    const bool comp = !warm;
    auto step = [&](auto t_tag) __attribute__((always_inline)) {
      constexpr int t = decltype(t_tag)::value;
      unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
      if (stamps) { t0 = pp_stamp(); if (tend) tsum[5] += t0 - tend; }
      const bool last = t == 1 && first && !warm;             // the compute step closes the previous tile
      {
        const unsigned so = lds0 + so_c;
        const unsigned pa0 = so + pbase[t], pa1 = so + (pbase[t] ^ 64u);
        const unsigned wa0 = so + wbase + t * BT, wa1 = so + ((wbase ^ 64u) + t * BT);
        const unsigned dst = lds0 + so_i + dst_wave;
        constexpr int NP = t == 0 ? IA + IB : IB;   // pieces of this half (wave 0: + the image's 33rd piece, issued first)
        auto piece = [&](auto q_tag) __attribute__((always_inline)) {
          constexpr int q = decltype(q_tag)::value;
          if (!ISS || (dbg & 1)) return;
          if constexpr (t == 0 && q < IA) dma_s(voffA[pi][q], sA, dst, std::integral_constant<int, NWV * q * 1024>{});
          else if constexpr (t == 0) dma_s(voffB[q - IA], (DUAL && q - IA == 1) ? sB0b : sB0, dst, std::integral_constant<int, AIMG + NWV * (q - IA) * 1024>{});
          else dma_s(voffB[q], (DUAL && q == 1) ? sB1b : sB1, dst, std::integral_constant<int, AIMG + BT + NWV * q * 1024>{});
        };
        if (ISS && t == 0 && wave == 0 && !(dbg & 1))
          dma_s(voffA[pi][IA], sA, lds0 + so_i, std::integral_constant<int, 32 * 1024>{});
        constexpr int NR = 2 * TM + 2 * TN;      // fragment reads (unconditional: in the very first pair they fetch bytes
        int q = 0;                               // nobody uses - cheaper than a second register set for a conditionally
                                                 // written fragment)
        auto pieces_upto = [&](int r) __attribute__((always_inline)) {   // a piece after every NR / NP reads
#define PP_PIECE(Q) if constexpr (Q < NP) { if (q == Q && r * NP >= Q * NR) { piece(std::integral_constant<int, Q>{}); ++q; } }
          PP_PIECE(0) PP_PIECE(1) PP_PIECE(2) PP_PIECE(3) PP_PIECE(4) PP_PIECE(5) PP_PIECE(6) PP_PIECE(7)
#undef PP_PIECE
        };
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          pieces_upto(r);
          if (dbg & 16) continue;
          if (r < 2 * TM) {
            const int i = r >> 1;
            if (r & 1) LDS_READ128(fp[1][i], pa1, i * 16 * SB); else LDS_READ128(fp[0][i], pa0, i * 16 * SB);
          } else {
            const int jj = (r - 2 * TM) >> 1;
            if (r & 1) LDS_READ128(fw[1][jj], wa1, jj * 4 * SB); else LDS_READ128(fw[0][jj], wa0, jj * 4 * SB);
          }
        }
        pieces_upto(1 << 20);
        pieces_upto(1 << 20);
      }
      if (stamps) t1 = pp_stamp();
      // everything but this half's pieces has landed (the stores of an epilogue that ran one half earlier are waited
      // for too - once per tile, mostly retired), and the fragments are in
      if (!ISS) PP_WAIT(0);
      else if (t == 0) { if (wave == 0) PP_WAIT(IA + IB + 1); else PP_WAIT(IA + IB); }
      else PP_WAIT(IB);
      // The finished tile's epilogue: behind the wait (mask source landed), in front of the MFMA half that restarts the
      // accumulators.  Both groups run theirs in the SAME barrier interval: the lagging group (waves 4-7) at the end of its
      // LOAD half, the leading group one barrier later at the head of its MFMA half.  (Round 3 had each group's epilogue
      // in its own LOAD half, i.e. in consecutive intervals: the partner's 32 MFMAs covered a fifth of it and the partner
      // then sat at the barrier - MFMA + epilogue + skeleton added up exactly in the ablations.  Two waves of a SIMD doing
      // VALU work together each still issue at their single-wave rate, so one epilogue time per tile is gone.)
      auto finish_tile = [&]() __attribute__((always_inline)) {
        if (!(dbg & 4)) epilogue(tprev);
        else {                                   // (ablation: keep the MFMAs alive without their consumer)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) asm volatile("" ::"v"(acc[i][jj]));
          zero_acc();
        }
        pending = false;
      };
      if (t == 0 && pending && wave >= 4) finish_tile();
      if (stamps) t2 = pp_stamp();
      __builtin_amdgcn_s_barrier();
      if (stamps) t3 = pp_stamp();
      __builtin_amdgcn_sched_barrier(0);
      if (t == 0 && pending) finish_tile();      // (waves 0-3)
      __builtin_amdgcn_sched_barrier(0);
      if (comp) {
        if (stamps && tkf == 0) tkf = t3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (!(dbg & 2)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int jj = 0; jj < TN; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[ks][jj]),
                                                                      __builtin_bit_cast(bf16x8, fp[ks][i]), acc[i][jj], 0, 0, 0);
          }
          if (ks == 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (MASK && last && !(dbg & (4 | 64))) load_aux(tprev);
          }
        }
        if (last) pending = true;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (stamps) t4 = pp_stamp();
      // (the drain pair's very last barrier: group A's pairs with group B's late start; group B has none left to pair with and
      //  goes straight from its last matrix half into its epilogue - see the end of the kernel)
      if (ISS || t == 0 || wave < 4) __builtin_amdgcn_s_barrier();
      if (stamps) {
        t5 = pp_stamp();
        tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += t4 - t3; tsum[4] += t5 - t4;
        tend = t5;
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    const unsigned sw_ = so_c; so_c = so_i; so_i = sw_;
  };

  Tile ti = first;
  for (int c = 0; c < tcount; ++c) {
    unsigned long long hl;
    int nh;
    {
      // (a scalar load from the kernel-argument segment: lgkmcnt only, nothing the LDS-DMA pieces in flight are ordered against)
      const unsigned long long e = ht.e[ti.Y];
      nh = (int)(e >> 60) & 7;
      hl = e & 0x0fffffffffffffffull;
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
    // weight taps of the pair's two K steps (see the table at the issue side)
    // (DUAL: ti.px == 0, i.e. kx 3, 1 for parity 0; parity 1 reads kx 2, 0 one image row further)
    const int kxa = MODE == MODE_S2 ? 0 : (ti.px == 0 ? 3 : 2), kxb = MODE == MODE_S2 ? 2 : (ti.px == 0 ? 1 : 0);
    bool firstg = true;
    for (int h = 0; h < nh; ++h) {
      const int it_r = (int)(hl & 1023) >> 2, it_ky = (int)hl & 3;
      hl >>= 10;
      const char* sA_row = in_t + (long)it_r * Ws * spb;
      const char* sB_row = w_t + (long)(it_ky * 4) * tap_b;
      for (int kc = 0; kc < (X2 ? 3 : 1) * KC; ++kc) {
        // X2: chunk kc / 3 as (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo): the lo halves sit 128 bytes behind their hi halves
        const int kq = X2 ? kc / 3 : kc, sub = X2 ? kc - 3 * kq : 0;
        const char* sA_k = sA_row + kq * (X2 ? 2 * SB : SB) + (sub == 1 ? SB : 0);
        const char* sB_k = sB_row + kq * (X2 ? 2 * SB : SB) + (sub == 2 ? SB : 0);
#pragma unroll
        for (int pi = 0; pi < NPAIR; ++pi) {
          pair_iter(std::true_type{}, firstg, pi, sA_k, sB_k + (long)(kxa + pi) * tap_b, sB_k + (long)(kxb + pi) * tap_b,
                    sB_k + 2 * tap_b, sB_k);
          if (firstg) { firstg = false; warm = false; }
        }
      }
    }
    tprev = ti;
    tile_next(ti);
  }
  pair_iter(std::false_type{}, true, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  const unsigned long long tk2 = (dbg & 8) ? pp_stamp() : 0ull;
  const unsigned long long tr2 = (dbg & 8) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (pending && !(dbg & 4)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    epilogue(tprev);
  }
  if (stamps && blockIdx.x == 0 && lane == 0) {
    float* sink = (float*)p.out + wave * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the final epilogue's stores acknowledged: part of its phase)
    const unsigned long long tk3 = pp_stamp();
    for (int k = 0; k < 6; ++k) sink[k] = (float)tsum[k];
    // phases of the workgroup's lifetime, cycles from kernel entry: prologue end, first matrix half, loop end, stores acknowledged
    float* ph = (float*)p.out + 64 + wave * 8;
    ph[0] = (float)(tk1 - tk0); ph[1] = (float)(tkf - tk0); ph[2] = (float)(tk2 - tk0); ph[3] = (float)(tk3 - tk0);
    ph[4] = (float)tcount;
    ph[5] = (float)(tr2 - tr0);                // 100 MHz ticks over the same span as ph[2] (shader cycles): the held clock
  }
  // Both groups' FINAL epilogues run side by side: group A's last barrier (the one that pairs with group B's late start) is
  // the drain pair's last one, taken BEFORE its epilogue.  (Round 4 had it after group A's epilogue: group B, one barrier
  // behind, sat in its last barrier until group A had stored its whole tile - round-5 stamps: 4 000 of a one-tile launch's
  // 70 000 cycles, in every launch.)
  if (want_db) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int n = tid; n < p.N; n += 64 * NWV) {
      float v;
      if (db_rows) {
        v = s_bias[n];                           // (row 0; rows in a fixed order)
#pragma unroll
        for (int r = 1; r < DBW; ++r) v += s_bias[r * p.N + n];
      } else v = s_db[n];
      // dbias_part: this workgroup's row of the caller's workspace (summed by dg_wgrad_reduce in a fixed order) - else atomics
      if (p.dbias_part) p.dbias_part[(long)blockIdx.x * p.N + n] = v;
      else atomicAdd(&p.dbias[n % p.bias_mod], v);
    }
  }
}

template <int BN, int MODE, bool MASK, bool DUAL = false, bool X2 = false>
int launch(const ConvP* p, const Geo& g0, hipStream_t stream, int wg_cap, DgConvPlan* plan) {
  constexpr int CPL_ = (BN / 2 / 16) * 4;      // bits per lane and block row: whole bytes / 16-bit words of the mask buffers
  const bool bits_ok = !X2 && p->out_sn == 1 && p->out_sb % 8 == 0 && p->out_sp % CPL_ == 0 && p->N % CPL_ == 0;
  if (!MASK && p->mask_out && !bits_ok) return DG_EUNSUPPORTED;
  const bool bits = MASK && p->mask_in && bits_ok;
  Geo g = g0;
  static int resident = 0;
  if (!resident) {
    int dev = 0, cus = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    HIP_CHECK_RET(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    resident = cus;                            // 148 KB of LDS: one workgroup per CU
  }
  g.dbg = 0;
  const int cap = (wg_cap > 0 && wg_cap < resident) ? wg_cap : resident;
  const int G = g.ntiles < cap ? g.ntiles : cap;
  if (plan) {
    plan->family = 5; plan->bm = DUAL ? 512 : 256; plan->bn = DUAL ? BN / 2 : BN; plan->tiles = g.ntiles; plan->workgroups = G;
    plan->tiles_per_wg = (g.ntiles + G - 1) / G;
    plan->mask_bits = MASK ? (bits ? 2 : 0) : 1;
    // (per-wave bias-gradient rows need DBW * N floats of the kernel's LDS rows: 2048)
    plan->dbias_rows = (MASK && (DUAL ? 8 : 4) * p->N <= 2048) ? G : 0;
    return DG_OK;
  }
  HTab ht;
  {
    const int rows = MODE == MODE_S2 ? p->Hc : 2 * p->Hc;      // (<= NHT: checked by dg_conv_mfma_pp_launch)
    for (int y = 0; y < NHT; ++y) {
      unsigned long long hl = 0;
      const int nh = y < rows ? persist::pack_htaps<MODE>(p->adj, y, p->Hc, hl) : 0;
      ht.e[y] = hl | ((unsigned long long)nh << 60);
    }
  }
  if constexpr (X2) {
    conv_pp_kernel<BN, MODE, MASK, DUAL, false, true><<<(unsigned)G, 512, 0, stream>>>(*p, g, ht);
  } else if constexpr (MASK) {
    if (bits) conv_pp_kernel<BN, MODE, true, DUAL, true><<<(unsigned)G, 512, 0, stream>>>(*p, g, ht);
    else conv_pp_kernel<BN, MODE, true, DUAL, false><<<(unsigned)G, 512, 0, stream>>>(*p, g, ht);
  } else conv_pp_kernel<BN, MODE, false, DUAL><<<(unsigned)G, 512, 0, stream>>>(*p, g, ht);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

}  // namespace pp

// bf16 layers that tile into 256-pixel x 128- (or 64-) channel tiles; DG_EUNSUPPORTED otherwise (the caller falls back
// to the lock-step kernels).  min_tiles: the auto rule wants every CU busy.
// dual: 1 = a 64-channel MODE_UP layer may take the both-parities tile (512 pixels x 64 channels), 0 = never (A/B, tests)
int dg_conv_mfma_pp_launch(const ConvP* p0, hipStream_t stream, int min_tiles, int wg_cap, DgConvPlan* plan, int dual) {
  if (p0->mode != MODE_S2 && p0->mode != MODE_UP) return DG_EUNSUPPORTED;
  const bool x2 = p0->in_dtype == DG_BF16X2;
  if (p0->in_dtype != p0->out_dtype || p0->in_dtype != p0->w_dtype || (p0->in_dtype != DG_BF16 && !x2)) return DG_EUNSUPPORTED;
  ConvP q = *p0;
  if (x2) {
    // DG_BF16X2: whole 64-channel groups everywhere, 256-byte aligned tensors; from here on strides in bf16 units
    if (p0->K % 64 || p0->N % 64 || p0->in_sp % 64 || p0->in_sb % 64 || p0->out_sp % 64 || p0->out_sb % 64 || p0->w_sn % 64 ||
        p0->w_st % 64 || (((size_t)p0->in | (size_t)p0->out | (size_t)p0->w | (size_t)p0->aux) & 255))
      return DG_EUNSUPPORTED;
    if (p0->mask_out || p0->mask_in) return DG_EUNSUPPORTED;   // (saved 1-bit masks: bf16 tensors only)
    q.in_sb *= 2; q.in_sp *= 2; q.out_sb *= 2; q.out_sp *= 2; q.w_st *= 2; q.w_sn *= 2;
  }
  const ConvP* p = &q;
  if (p->K % 64 != 0 || !p->ring || p->nscale) return DG_EUNSUPPORTED;
  // a tile must span at least two pair iterations (the compute side finishes the previous tile during the first one and
  // its epilogue runs at the start of the second): the adjoint MODE_UP pass with K == 64 could have one (a single H tap at
  // a border row x one channel chunk x one pair; the forward flavour reflects and always has two H taps)
  if (p->mode == MODE_UP && p->adj && p->K < 128) return DG_EUNSUPPORTED;
  if (p->in_sk != 1 || p->w_sk != 1 || p->out_sn != 1) return DG_EUNSUPPORTED;
  if (p->out_sp % 8 != 0 || p->in_sp % 8 != 0 || p->w_sn % 8 != 0) return DG_EUNSUPPORTED;   // 16-byte pieces
  if (p->N > 512 || (p->bias && p->bias_mod < p->N && p->N % p->bias_mod != 0)) return DG_EUNSUPPORTED;
  if (p->dbias && p->bias_mod < p->N) return DG_EUNSUPPORTED;
  const int Ws = p->mode == MODE_S2 ? 2 * p->Wc : p->Wc;
  if ((Ws & (Ws - 1)) != 0 || p->in_sp * 2 >= (1 << 24)) return DG_EUNSUPPORTED;  // column wrap by mask, 24-bit multiply
  if ((p->mode == MODE_S2 ? p->Hc : 2 * p->Hc) > 256) return DG_EUNSUPPORTED;       // rows of the kernel's H-tap table
  const bool mask = p->epi == EPI_MASK;
  if (mask ? p->bias != nullptr : p->dbias != nullptr) return DG_EUNSUPPORTED;   // combinations no layer uses
  if (p->rowscale && p->B > 512) return DG_EUNSUPPORTED;
  persist::Geo g;
  int bn = 0;
  if (dual && p->mode == MODE_UP && p->N % 128 != 0 && p->N % 64 == 0 && persist::make_geo<256, 64>(p, g) && g.SW >= 64 &&
      g.NSB * (g.SW + 2) <= 264 && g.ntiles / 2 >= min_tiles) {
    g.ntiles /= 2;                             // one tile = both column parities of 256 coarse columns
    if (x2) return mask ? pp::launch<128, MODE_UP, true, true, true>(p, g, stream, wg_cap, plan)
                        : pp::launch<128, MODE_UP, false, true, true>(p, g, stream, wg_cap, plan);
    return mask ? pp::launch<128, MODE_UP, true, true>(p, g, stream, wg_cap, plan)
                : pp::launch<128, MODE_UP, false, true>(p, g, stream, wg_cap, plan);
  }
  if (p->N % 128 == 0 && persist::make_geo<256, 128>(p, g) && g.SW >= 64 && g.ntiles >= min_tiles) bn = 128;
  else if (p->N % 64 == 0 && persist::make_geo<256, 64>(p, g) && g.SW >= 64 && g.ntiles >= min_tiles) bn = 64;
  if (!bn) return DG_EUNSUPPORTED;
  const int sel = (bn == 128 ? 0 : 4) + (p->mode == MODE_UP ? 2 : 0) + (mask ? 1 : 0);
  if (x2) switch (sel) {
    case 0: return pp::launch<128, MODE_S2, false, false, true>(p, g, stream, wg_cap, plan);
    case 1: return pp::launch<128, MODE_S2, true, false, true>(p, g, stream, wg_cap, plan);
    case 2: return pp::launch<128, MODE_UP, false, false, true>(p, g, stream, wg_cap, plan);
    case 3: return pp::launch<128, MODE_UP, true, false, true>(p, g, stream, wg_cap, plan);
    case 4: return pp::launch<64, MODE_S2, false, false, true>(p, g, stream, wg_cap, plan);
    case 5: return pp::launch<64, MODE_S2, true, false, true>(p, g, stream, wg_cap, plan);
    case 6: return pp::launch<64, MODE_UP, false, false, true>(p, g, stream, wg_cap, plan);
    default: return pp::launch<64, MODE_UP, true, false, true>(p, g, stream, wg_cap, plan);
  }
  switch (sel) {
    case 0: return pp::launch<128, MODE_S2, false>(p, g, stream, wg_cap, plan);
    case 1: return pp::launch<128, MODE_S2, true>(p, g, stream, wg_cap, plan);
    case 2: return pp::launch<128, MODE_UP, false>(p, g, stream, wg_cap, plan);
    case 3: return pp::launch<128, MODE_UP, true>(p, g, stream, wg_cap, plan);
    case 4: return pp::launch<64, MODE_S2, false>(p, g, stream, wg_cap, plan);
    case 5: return pp::launch<64, MODE_S2, true>(p, g, stream, wg_cap, plan);
    case 6: return pp::launch<64, MODE_UP, false>(p, g, stream, wg_cap, plan);
    default: return pp::launch<64, MODE_UP, true>(p, g, stream, wg_cap, plan);
  }
}
