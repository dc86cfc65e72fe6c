// Persistent implicit-GEMM conv on the matrix cores (gfx950) with large tiles.
//   Down forward / R1 tangent pass (MODE_S2, adj=0), Up forward (MODE_UP, adj=0), Down backward-data (MODE_UP, adj=1),
//   Up backward-data (MODE_S2, adj=1).
// Reference ops: models/gans/dcgan_eqlr.py:6-26,75-82 with models/ops/common.py Pad/EqualLR/FusedLeakyReLU fused
// (padding = index arithmetic in the tile loader; EqualLR scale, bias, leaky-relu or its derivative mask and the
// bias-gradient column sums in the epilogue).
//
// Why this kernel exists: the one-tile-per-workgroup kernel of conv_mfma.hip (128 x 128 tiles, 2 workgroups per CU)
// issues 8 LDS-DMA pieces per 16 MFMAs per wave and measured DMA-only ~= MFMA-only ~= half of the combined time: it
// is bound by the DMA issue / L1 path, not by latency (a persistent 128 x 128 variant with cross-tile prefetch ran
// at the same speed).  The lever is bytes per flop, i.e. the tile, plus fewer barriers per byte (128-byte stages).
// Shipped tiles: 256 x 128 (8 waves, the default for every layer with >= 256 tiles), 256 x 64 (N == 64 layers with
// the mask epilogue) and 128 x 128 with 8 waves (small layers); 256 x 256 only fits the LDS with 64-byte stages x 4,
// measured equal to 256 x 128 on 128-byte stages x 3, and is not instantiated.  What it takes for conv layers:
//   * an M tile of 256 output pixels of ONE image row (and one column parity in MODE_UP) keeps the tap list
//     workgroup-uniform; layers narrower than 256 build the tile from the SAME row segment of NSB consecutive
//     samples (the tap list only depends on the row), so every fat layer gets 256-row tiles;
//   * the grid is one residency wave; each workgroup owns a contiguous chunk of the tile order (N tile fastest,
//     then x tile, column parity, row, sample group) and walks it with a carry chain - no divisions;
//   * the LDS-DMA ring (NS stages of SB bytes of K per tile row) runs ACROSS tile boundaries with counted vmcnt
//     waits (vmcnt retires in issue order, so the epilogue's stores are counted, not drained), one raw barrier per
//     K step, and the DMA pieces of step s+NS-1 are issued between the MFMA groups of step s;
//   * DMA addressing = wave-uniform 64-bit base (SGPRs) + per-lane 32-bit offset that only changes with the tile or
//     the W tap; the H taps of a row are packed into one 64-bit scalar;
//   * the epilogue is wave-private: each wave transposes its SR-row slabs through its own LDS strip (outside the
//     ring) with inline-asm ds ops and writes whole 128-byte pixel rows - no workgroup barrier, no vmcnt(0) drain.
#pragma once
#include "mfma_common.h"

namespace persist {

template <typename T> struct StageWrite;
template <> struct StageWrite<bf16> {
  static __device__ __forceinline__ void put(unsigned addr, float v) {
    const bf16 h = (bf16)v;
    const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, h);
    asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(bits) : "memory");
  }
};
template <> struct StageWrite<float> {
  static __device__ __forceinline__ void put(unsigned addr, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
  }
};

// position in the tile order; bt = sample group (samples bt*NSB .. bt*NSB+NSB-1)
struct Tile {
  int nt, xt, px, Y, bt;
};

template <int MODE>
__device__ __forceinline__ bool next_tile(Tile& t, int tiles_n, int tiles_x, int rows) {  // true when Y changed
  if (++t.nt < tiles_n) return false;
  t.nt = 0;
  if (++t.xt < tiles_x) return false;
  t.xt = 0;
  if (MODE == MODE_UP) {
    if (++t.px < 2) return false;
    t.px = 0;
  }
  if (++t.Y == rows) { t.Y = 0; ++t.bt; }
  return true;
}

// H taps of output row Y: up to 6 entries (src row << 2 | ky), 10 bits each, first tap in the low bits
template <int MODE>
__host__ __device__ __forceinline__ int pack_htaps(int adj, int Y, int Hc, unsigned long long& list) {
  list = 0;
  int n = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int r, ky;
    if (dg_tap1d(MODE, adj, 0, Y, Hc, i, r, ky)) {
      list |= (unsigned long long)((r << 2) | ky) << (10 * n);
      ++n;
    }
  }
  return n;
}

#define DG_WAITV(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// geometry passed by the launcher
struct Geo {
  int tiles_n, tiles_x, ntiles;
  int SW, NSB, lsw;                            // M tile = NSB sample segments of SW = 1 << lsw columns
  int dbg;
};

template <typename T, int BM, int BN, int WM, int WN, int SB, int NS, int MODE>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_kernel(ConvP p, Geo g) {
  constexpr int NWV = WM * WN;                 // waves
  constexpr int ES = sizeof(T);
  constexpr int BK = SB / ES;
  constexpr int EPC = 16 / ES;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // 32 x 32 MFMA blocks per wave
  constexpr int RPI = 1024 / SB;               // tile rows per DMA piece (1 KiB per wave instruction)
  constexpr int CPR = SB / 16;                 // 16-byte chunks per tile row
  constexpr int KS = SB / 32;                  // MFMA k-steps per stage
  constexpr int STAGE = (BM + BN) * SB;
  constexpr int IA = BM / RPI / NWV, IB = BN / RPI / NWV;
  constexpr int IPT = IA + IB;                 // DMA pieces per wave per K step
  constexpr int E = NS - 1;                    // DMA of step s+E is issued in the middle of step s
  constexpr int WC = BN / WN;                  // epilogue strip of one wave: SR pixel rows x WC channels (+16 B pad)
  constexpr int RB = WC * ES;
  constexpr int RS = RB + 16;
  constexpr int CR = RB / 16;                  // 16-byte chunks per strip row
  constexpr int NDB0 = 512;
  // bias-gradient rows in LDS: fp32 (the parity mode) keeps one per wave ROW, each written by one wave per channel in tile order -
  // the workgroup's sums do not depend on which wave adds first (round 6); bf16 has no room for them beside its ring
  constexpr int DBR = sizeof(T) == 4 ? WM : 1;
  // strip rows: 16 unless the ring + 16-row strips would not fit the 160 KiB of LDS
  constexpr int SR = (NS * STAGE + NWV * 16 * RS + NDB0 * 4 * DBR > 160 * 1024) ? 8 : 16;
  constexpr int CPL = SR * CR / 64;            // chunks per lane per strip
  constexpr int STRIP = SR * RS;
  constexpr int NQ = 32 / SR;                  // strips per 32-row MFMA block
  constexpr int NST = TM * NQ * CPL;           // global stores per wave per tile epilogue
  constexpr int nW = MODE == MODE_S2 ? 4 : 2;
  static_assert(IA >= 1 && IB >= 1, "piece distribution");
  static_assert(NS >= 3 && NS <= 4 && KS >= 2 && KS % 2 == 0 && 2 * IPT + NST <= 63, "ring / vmcnt immediates");
  constexpr int NDB = 512;                     // bias-gradient accumulators (N <= 512 when dbias is wanted)
  static_assert(NS * STAGE + NWV * STRIP + NDB * 4 * DBR <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE + NWV * STRIP + NDB * 4 * DBR];

  // ---- this workgroup's chunk of the tile order
  // XCD-aware, bijective chunk assignment (blocks id and id+8 share an XCD): the chunks of one XCD are CONTIGUOUS in
  // the tile order, so the input rows that neighbouring chunks share (vertical tap overlap) are fetched into one L2
  // instead of being pulled from the fabric by several (PMC: 294 MB/launch against 112 MB algorithmic without it)
  const int G = gridDim.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = blockIdx.x & 7;
  const int gi = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tq = g.ntiles / G, tr = g.ntiles % G;
  const int t0 = gi * tq + (gi < tr ? gi : tr);
  const int tcount = tq + (gi < tr ? 1 : 0);
  if (tcount == 0) return;

  const int tid = threadIdx.x;
  const int KC = p.K / BK;
  const int Ws = MODE == MODE_S2 ? 2 * p.Wc : p.Wc, cmul = MODE == MODE_S2 ? 2 : 1;
  const int Wo = MODE == MODE_S2 ? p.Wc : 2 * p.Wc;
  const int rows = MODE == MODE_S2 ? p.Hc : 2 * p.Hc;
  const int tiles_n = g.tiles_n, tiles_x = g.tiles_x;

  Tile first;
  {
    int mt = t0 / tiles_n;
    first.nt = t0 % tiles_n;
    first.px = 0;
    first.xt = mt % tiles_x; mt /= tiles_x;
    if (MODE == MODE_UP) { first.px = mt & 1; mt >>= 1; }
    first.Y = mt % rows; first.bt = mt / rows;
    first.nt = __builtin_amdgcn_readfirstlane(first.nt); first.xt = __builtin_amdgcn_readfirstlane(first.xt);
    first.px = __builtin_amdgcn_readfirstlane(first.px); first.Y = __builtin_amdgcn_readfirstlane(first.Y);
    first.bt = __builtin_amdgcn_readfirstlane(first.bt);
  }

  const T* in = (const T*)p.in;
  const T* w = (const T*)p.w;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lrow = lane / CPR, pos = lane % CPR;
  auto swz = [](int row) { return SB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // ---- issue side: runs D K-steps ahead of the compute side
  Tile ti = first;
  int i_left = tcount;                         // tiles not yet fully issued (including the current one)
  unsigned long long hl = 0, hl_row;           // remaining H taps of the current tile / tap list of row ti.Y
  int h_left = 0;
  int nh_row = pack_htaps<MODE>(p.adj, ti.Y, p.Hc, hl_row);
  int it_j = 0, it_kc = 0;
  unsigned voffA[IA], voffB[IB];               // per-lane byte offsets (A: per tile and W tap, B: constant)
  int colA[IA];                                // tile-row column inside its sample segment
  unsigned sampA[IA];                          // byte offset of the row's sample inside the sample group + swizzled chunk
  const char* sA = nullptr;                    // wave-uniform bases
  const char* sB = nullptr;
#pragma unroll
  for (int u = 0; u < IA; ++u) {
    const int row = (wave + NWV * u) * RPI + lrow;
    colA[u] = row & (g.SW - 1);
    sampA[u] = (unsigned)((row >> g.lsw) * (int)p.in_sb * ES + (pos ^ swz(row)) * 16);
  }
#pragma unroll
  for (int u = 0; u < IB; ++u) {
    const int row = (wave + NWV * u) * RPI + lrow;
    voffB[u] = (unsigned)(row * (int)p.w_sn * ES + (pos ^ swz(row)) * 16);
  }
  auto set_wtap = [&]() {                      // after the tile, the H tap or the W tap changed
    const int it_r = (int)(hl & 1023) >> 2, it_ky = (int)hl & 3;
    int coff, kx;
    if (MODE == MODE_S2) { coff = it_j - 1; kx = it_j; }
    else if (ti.px == 0) { coff = it_j == 0 ? 0 : -1; kx = it_j == 0 ? 1 : 3; }
    else { coff = it_j == 0 ? 1 : 0; kx = it_j == 0 ? 0 : 2; }
    const int wt = it_ky * 4 + kx;
    sB = (const char*)(w + (long)wt * p.w_st + (long)(ti.nt * BN) * p.w_sn);
    sA = (const char*)(in + (long)(ti.bt * g.NSB) * p.in_sb + (long)it_r * Ws * p.in_sp);
#pragma unroll
    for (int u = 0; u < IA; ++u) {
      int c = cmul * (ti.xt * BM + colA[u]) + coff;
      if (c < 0) c += Ws; else if (c >= Ws) c -= Ws;
      voffA[u] = (unsigned)(c * (int)p.in_sp * ES) + sampA[u];
    }
  };
  auto start_tile = [&]() { hl = hl_row; h_left = nh_row; it_j = 0; it_kc = 0; set_wtap(); };
  start_tile();
  auto advance = [&]() {
    if (++it_kc < KC) return;
    it_kc = 0;
    if (++it_j < nW) { set_wtap(); return; }
    it_j = 0;
    hl >>= 10;
    if (--h_left > 0) { set_wtap(); return; }
    if (--i_left == 0) return;
    if (next_tile<MODE>(ti, tiles_n, tiles_x, rows)) nh_row = pack_htaps<MODE>(p.adj, ti.Y, p.Hc, hl_row);
    start_tile();
  };
  // piece pc (0 .. IPT-1) of the current issue step into stage st
  auto issue_piece = [&](int st, int pc) {
    unsigned char* base = lds + st * STAGE;
    const unsigned k0b = (unsigned)it_kc * SB;
    if (pc < IA) dma16(sA + k0b + voffA[pc], base + (wave + NWV * pc) * 1024);
    else dma16(sB + k0b + voffB[pc - IA], base + BM * SB + (wave + NWV * (pc - IA)) * 1024);
  };

  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int sw = swz(lr);                      // fragment row bases are multiples of 32, so swz(row) == swz(lr)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  unsigned fragA[KS], fragB[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int off = ((2 * ks + lh) ^ sw) * 16;
    fragA[ks] = lds0 + (wm * (BM / WM) + lr) * SB + off;
    fragB[ks] = lds0 + (BM + wn * (BN / WN) + lr) * SB + off;
  }

  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();

  // Fragment reads are inline asm (a compiler-visible LDS load after an LDS-DMA makes hipcc drain vmcnt(0)); the reads
  // of MFMA k-step ks+1 are issued before the MFMAs of ks, counted lgkmcnt + sched_barrier keep the order.  The DMA
  // pieces of the step being prefetched are issued behind each MFMA group.
  i32x4 fa[2][TM], fb[2][TN];
  auto read_frags = [&](int set, unsigned a_addr, unsigned b_addr) {
#pragma unroll
    for (int i = 0; i < TM; ++i) LDS_READ128(fa[set][i], a_addr + i * 32 * SB, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) LDS_READ128(fb[set][j], b_addr + j * 32 * SB, 0);
  };
  auto mfma_group = [&](int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_tile((const T*)nullptr, fa[set][i], fb[set][j], acc[i][j]);
  };
  // wait until all but the DMA of the `yd` youngest issued K steps (and, with `st`, the previous epilogue's NST
  // stores, which are younger than the step waited for) have landed
  auto wait_dma = [&](int yd, bool st) {
    if (st) {
      if (yd <= 0) DG_WAITV(NST);
      else if (yd == 1) DG_WAITV(IPT + NST);
      else DG_WAITV(2 * IPT + NST);
    } else {
      if (yd <= 0) DG_WAITV(0);
      else if (yd == 1) DG_WAITV(IPT);
      else DG_WAITV(2 * IPT);
    }
  };

  // ---- epilogue constants
  const unsigned strip = lds0 + NS * STAGE + wave * STRIP;
  // bias-gradient column sums of ALL tiles of this workgroup accumulate in LDS (ds_add_f32 in inline asm, so that no
  // compiler-visible LDS access sits behind an in-flight DMA) and are flushed once at the end of the kernel
  float* s_db = (float*)(lds + NS * STAGE + NWV * STRIP);
  const unsigned sdb0 = lds0 + NS * STAGE + NWV * STRIP;
  const bool want_db = p.dbias != nullptr;
  if (want_db) {
    for (int i = tid; i < NDB * DBR; i += 64 * NWV) s_db[i] = 0.f;
    __syncthreads();
  }
  T* out = (T*)p.out;
  const int part = lane % CR;                  // this lane's 16-byte chunk of the strip rows (64 % CR == 0)
  const unsigned wr_base = strip + (4 * lh) * RS + lr * ES;
  const unsigned rd_base = strip + (lane / CR) * RS + part * 16;
  const bool counted = !(g.dbg & 4);           // the epilogue issues exactly NST stores per wave

  // ---- K loop.  Stage gs % NS holds step gs.  The barrier of a step sits in the MIDDLE of the previous one:
  //        step gs:  MFMA group(s) of the first half
  //                  wait DMA(gs+1) landed (counted vmcnt) ; s_barrier ; issue DMA(gs+E) into the stage step gs-1 used
  //                  MFMA group(s) of the second half, with the first fragments of step gs+1 read behind them
  //      so the barrier skew, the LDS read latency of the next step and the DMA issue all overlap MFMAs in flight.
  //      The barrier orders (a) every wave's share of DMA(gs+1) against the reads that follow and (b) every wave's
  //      reads of stage (gs-1) % NS (all done: everyone is inside step gs) against the DMA that refills it.
  int issued = 0;                              // K steps whose DMA has been issued
#pragma unroll
  for (int d = 0; d < E; ++d) {
    if (i_left > 0) {
      if (!(g.dbg & 1)) {
#pragma unroll
        for (int pc = 0; pc < IPT; ++pc) issue_piece(d, pc);
      }
      advance();
      ++issued;
    }
  }
  wait_dma(issued - 1, false);
  __builtin_amdgcn_s_barrier();
  int gs = 0;                                  // K-step counter of this workgroup
  int st_c = 0, st_i = E % NS;                 // stage of step gs / stage refilled during it
  Tile tc = first;
  unsigned long long dummy;
  int nh_c = pack_htaps<MODE>(p.adj, tc.Y, p.Hc, dummy);
  for (int c = 0; c < tcount; ++c) {
    const int nsteps = nh_c * nW * KC;
    for (int s = 0; s < nsteps; ++s, ++gs) {
      const unsigned st_off = (unsigned)st_c * STAGE;
      const int st_n = st_c + 1 == NS ? 0 : st_c + 1;
      const unsigned nx_off = (unsigned)st_n * STAGE;
      const bool pref = s + 1 < nsteps;        // the next step belongs to this tile: read its first fragments early
      const bool has_next = pref || c + 1 < tcount;
      if (s == 0) read_frags(0, fragA[0] + st_off, fragB[0] + st_off);  // tile start: nothing was prefetched
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
          read_frags((ks + 1) & 1, fragA[ks + 1] + st_off, fragB[ks + 1] + st_off);
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
        } else if (pref) {
          read_frags(0, fragA[0] + nx_off, fragB[0] + nx_off);
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(g.dbg & 2)) mfma_group(ks & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == KS / 2 - 1 && has_next) {
          // the stores of the previous tile's epilogue are younger than DMA(gs+1) during the first E-1 steps
          wait_dma(issued - 2 - gs, counted && c > 0 && s < E - 1);
          if (!counted) DG_WAITV(0);
          __builtin_amdgcn_s_barrier();
          if (i_left > 0) {
            if (!(g.dbg & 1)) {
#pragma unroll
              for (int pc = 0; pc < IPT; ++pc) issue_piece(st_i, pc);
            }
            advance();
            ++issued;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      st_c = st_n;
      st_i = st_i + 1 == NS ? 0 : st_i + 1;
    }
    if (!(g.dbg & 4)) {
      // ---- wave-private epilogue: TM x NQ strips of SR rows; lane (lr, lh) owns channel column lr of rows
      //      (e & 3) + 8 (e >> 2) + 4 lh of each 32 x 32 MFMA block
      const int colb = tc.nt * BN + wn * WC;   // first channel of this wave's strip
      const int n0 = tc.xt * BM;
      // element offset of (sample group, row Y, column n0 [+ parity], strip channel 0)
      const long toff = (long)(tc.bt * g.NSB) * p.out_sb +
                        ((long)tc.Y * Wo + (MODE == MODE_S2 ? n0 : 2 * n0 + tc.px)) * p.out_sp + colb;
      char* obase = (char*)(out + toff);
      const char* abase = (const char*)((const T*)p.aux + toff);
      float csum[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) csum[e] = 0.f;
      float bias[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = colb + j * 32 + lr;
        bias[j] = p.bias ? p.bias[n % p.bias_mod] : 0.f;
      }
      // byte offset of this lane's chunk u of strip (i, q) from the tile's wave-uniform base
      auto chunk_off = [&](int i, int q, int u) -> unsigned {
        const int trow = wm * (BM / WM) + i * 32 + q * SR + u * (64 / CR) + lane / CR;
        const int sb = trow >> g.lsw, x = trow & (g.SW - 1);
        return (unsigned)((sb * (int)p.out_sb + (MODE == MODE_S2 ? x : 2 * x) * (int)p.out_sp + part * EPC) * ES);
      };
      // leaky-relu mask source, one 32-row block ahead of its use (double buffer)
      uint4 araw[2][NQ][CPL];
      auto load_aux = [&](int i) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int u = 0; u < CPL; ++u) araw[i & 1][q][u] = *(const uint4*)(abase + chunk_off(i, q, u));
      };
      if (p.epi == EPI_MASK) load_aux(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (p.epi == EPI_MASK && i + 1 < TM) load_aux(i + 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int es = 0; es < SR / 2; ++es) {  // this lane's SR/2 rows of the strip
              float v = acc[i][j][(SR / 2) * q + es] * p.scale + bias[j];
              if (p.epi == EPI_LRELU) v = (v > 0.f ? v : LRELU_SLOPE * v) * SQRT2;
              StageWrite<T>::put(wr_base + ((es & 3) + 8 * (es >> 2)) * RS + j * 32 * ES, v);
            }
          }
          i32x4 raw[CPL];
#pragma unroll
          for (int u = 0; u < CPL; ++u) LDS_READ128(raw[u], rd_base + u * (64 / CR) * RS, 0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          float rs = 1.f;                      // per-sample weight of the bias gradient (strip rows share a sample)
          if (want_db && p.rowscale) rs = p.rowscale[tc.bt * g.NSB + ((wm * (BM / WM) + i * 32 + q * SR) >> g.lsw)];
#pragma unroll
          for (int u = 0; u < CPL; ++u) {
            T* v = (T*)&raw[u];
            if (p.epi == EPI_MASK) {
              const T* av = (const T*)&araw[i & 1][q][u];
#pragma unroll
              for (int e = 0; e < EPC; ++e) {
                const float f = (float)v[e] * ((float)av[e] > 0.f ? SQRT2 : LRELU_SLOPE * SQRT2);
                v[e] = (T)f;
              }
            }
            if (want_db) {
#pragma unroll
              for (int e = 0; e < EPC; ++e) csum[e] += (float)v[e] * rs;
            }
            *(i32x4*)(obase + chunk_off(i, q, u)) = raw[u];
          }
        }
      }
      if (want_db) {
        // lanes with equal `part` inside a wave are CR lanes apart
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float v = csum[e];
          for (int d = CR; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
          if (lane < CR) {
            const unsigned a = sdb0 + (unsigned)((DBR > 1 ? wm * NDB : 0) + colb + part * EPC + e) * 4;
            asm volatile("ds_add_f32 %0, %1" ::"v"(a), "v"(v) : "memory");
          }
        }
      }
    }
    zero_acc();
    if (next_tile<MODE>(tc, tiles_n, tiles_x, rows)) nh_c = pack_htaps<MODE>(p.adj, tc.Y, p.Hc, dummy);
  }
  if (want_db) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int n = tid; n < p.N; n += 64 * NWV) {
      float v = s_db[n];
#pragma unroll
      for (int r = 1; r < DBR; ++r) v += s_db[r * NDB + n];
      // dbias_part: this workgroup's row of the caller's workspace (summed by dg_wgrad_reduce in a fixed order) - else atomics
      if (p.dbias_part) p.dbias_part[(long)blockIdx.x * p.N + n] = v;
      else atomicAdd(&p.dbias[n % p.bias_mod], v);
    }
  }
}

// geometry of a launch with BM x BN tiles; false when the layer does not tile that way
template <int BM, int BN>
bool make_geo(const ConvP* p, Geo& g) {
  if (p->N % BN != 0) return false;
  const int Wm = p->Wc;                        // width of an output row (S2) / of one column parity of it (UP)
  if (Wm >= BM) {
    if (Wm % BM != 0) return false;
    g.SW = BM; g.NSB = 1; g.tiles_x = Wm / BM;
  } else {
    if (BM % Wm != 0 || Wm < 64 || (Wm & (Wm - 1)) != 0) return false;
    g.NSB = BM / Wm;
    if (p->B % g.NSB != 0) return false;
    g.SW = Wm; g.tiles_x = 1;
  }
  g.lsw = 0;
  while ((1 << g.lsw) < g.SW) ++g.lsw;
  g.tiles_n = p->N / BN;
  const long rows = p->mode == MODE_S2 ? p->Hc : 4L * p->Hc;  // UP: 2 Hc rows x 2 parities
  const long nt = (long)(p->B / g.NSB) * rows * g.tiles_x * g.tiles_n;
  if (nt <= 0 || nt > 0x7fffffffL) return false;
  g.ntiles = (int)nt;
  return true;
}

template <typename T, int BM, int BN, int WM, int WN, int SB, int NS, int MODE>
int launch(const ConvP* p, const Geo& g0, hipStream_t stream, int wg_cap, DgConvPlan* plan) {
  Geo g = g0;
  static int resident = 0;                     // workgroups the device holds at once
  if (!resident) {
    int occ = 0, dev = 0, cus = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    HIP_CHECK_RET(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_CHECK_RET(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_kernel<T, BM, BN, WM, WN, SB, NS, MODE>,
                                                               64 * WM * WN, 0));
    if (occ < 1) occ = 1;
    resident = occ * cus;
  }
  g.dbg = 0;  // (ablation bits of the kernel-development builds: 1 no DMA, 2 no MFMA, 4 no epilogue)
  // wg_cap (dg_conv_ex): fewer workgroups than the device holds, i.e. longer tile chunks per workgroup (parity tests
  // of the cross-tile ring on small problems)
  const int cap = (wg_cap > 0 && wg_cap < resident) ? wg_cap : resident;
  const int G = g.ntiles < cap ? g.ntiles : cap;
  if (plan) {
    plan->family = 4; plan->bm = BM; plan->bn = BN; plan->tiles = g.ntiles; plan->workgroups = G;
    plan->tiles_per_wg = (g.ntiles + G - 1) / G;
    plan->dbias_rows = (sizeof(T) == 4 && p->bias_mod == p->N) ? G : 0;   // fp32: one partial row per workgroup (DgConv.dbias_part)
    return DG_OK;
  }
  conv_kernel<T, BM, BN, WM, WN, SB, NS, MODE><<<(unsigned)G, 64 * WM * WN, 0, stream>>>(*p, g);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// which tile the persistent kernel uses for this layer: 0 none (caller falls back to the 128-wide one-tile kernel),
// 2 = 256 x 128, 3 = 256 x 64, 4 = 128 x 128.  `auto_rule`: large tiles only when every CU gets one (measured: layers with fewer
// than 256 such tiles lose badly).  With 128-byte stages the 256 x 128 tile wins on every layer that tiles that way
// (scripts/bench_conv.py, MI355X, bf16: -7...-17 % per layer against the 128-wide kernel).
inline int pick(const ConvP* p, Geo& g, bool auto_rule) {
  if (p->mode != MODE_S2 && p->mode != MODE_UP) return 0;
  const int es = p->in_dtype == DG_BF16 ? 2 : 4;
  if (p->K % (64 / es) != 0) return 0;
  if (p->dbias && (p->N > 512 || p->bias_mod < p->N)) return 0;
  const int min_tiles = auto_rule ? 256 : 1;
  // (a 256 x 256 tile on 64-byte stages x 4 - all the LDS allows - measured equal to 256 x 128 on 128-byte stages on the
  // two layers that have enough of them, so it is not instantiated)
  if (make_geo<256, 128>(p, g) && g.ntiles >= min_tiles) return 2;
  // 64-channel layers: 256 x 64 tiles (8 waves of 32 x 64) where the epilogue reads the mask source (Down2 backward-data
  // -14 %); the lrelu layer of that shape (Up3 forward) measured 5 % slower and stays on the 128-wide kernel
  if (p->N == 64 && (!auto_rule || p->epi == EPI_MASK) && make_geo<256, 64>(p, g) && g.ntiles >= min_tiles) return 3;
  // layers with too few 256-row tiles (the 4 x 64 maps at batch 32): 128 x 128 tiles on 8 waves of 32 x 64, built from
  // two samples' row segments where the map is 64 wide (Up1 backward-data -13 %, Down4 forward at B = 32 -10 %)
  if (make_geo<128, 128>(p, g) && g.ntiles >= min_tiles) return 4;
  return 0;
}

template <typename T>
int launch_dtype(const ConvP* p, hipStream_t stream, bool auto_rule, int wg_cap, DgConvPlan* plan) {
  Geo g;
  const int which = pick(p, g, auto_rule);
  if (which == 2) {
    // bf16: 128-byte stages x 3 (one barrier per 64 channels, 8-row epilogue strips to fit the LDS) measured 8-12 %
    // faster than 64 x 4 on every layer of this tile
    if constexpr (sizeof(T) == 2) {
      if (p->K % 64 == 0)
        return p->mode == MODE_S2 ? launch<T, 256, 128, 4, 2, 128, 3, MODE_S2>(p, g, stream, wg_cap, plan)
                                  : launch<T, 256, 128, 4, 2, 128, 3, MODE_UP>(p, g, stream, wg_cap, plan);
    }
    return p->mode == MODE_S2 ? launch<T, 256, 128, 4, 2, 64, 4, MODE_S2>(p, g, stream, wg_cap, plan)
                              : launch<T, 256, 128, 4, 2, 64, 4, MODE_UP>(p, g, stream, wg_cap, plan);
  }
  if (which == 4) {
    if constexpr (sizeof(T) == 2) {
      if (p->K % 64 == 0)
        return p->mode == MODE_S2 ? launch<T, 128, 128, 4, 2, 128, 3, MODE_S2>(p, g, stream, wg_cap, plan)
                                  : launch<T, 128, 128, 4, 2, 128, 3, MODE_UP>(p, g, stream, wg_cap, plan);
    }
  }
  if (which == 3) {
    if constexpr (sizeof(T) == 2) {
      if (p->K % 64 == 0)
        return p->mode == MODE_S2 ? launch<T, 256, 64, 8, 1, 128, 3, MODE_S2>(p, g, stream, wg_cap, plan)
                                  : launch<T, 256, 64, 8, 1, 128, 3, MODE_UP>(p, g, stream, wg_cap, plan);
    }
  }
  return DG_EUNSUPPORTED;
}

}  // namespace persist
