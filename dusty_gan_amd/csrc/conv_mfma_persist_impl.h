// Persistent implicit-GEMM conv on the matrix cores (gfx950), the default kernel for every "fat" conv-like pass:
//   Down forward / R1 tangent pass (MODE_S2, adj=0), Up forward (MODE_UP, adj=0), Down backward-data (MODE_UP, adj=1),
//   Up backward-data (MODE_S2, adj=1) and the Proj GEMM (MODE_GEMM).
// Reference ops: models/gans/dcgan_eqlr.py:6-26,75-82 with models/ops/common.py Pad/EqualLR/FusedLeakyReLU fused
// (padding = index arithmetic in the tile loader; EqualLR scale, bias, leaky-relu or its derivative mask and the
// bias-gradient column sums in the epilogue).
//
// Same tiles, taps, swizzle and arithmetic as the one-tile-per-workgroup kernel in conv_mfma.hip, but
//   * the grid is one residency wave (occupancy x CUs); each workgroup owns a CONTIGUOUS chunk of the tile order
//     (N-tile fastest, then x-tile, column parity, row, sample) and steps through it with a carry chain - no
//     divisions, and the A rows of one M-tile are re-read from L2 by the same workgroup for its N-tiles;
//   * the LDS-DMA ring runs ACROSS tile boundaries: the first K step of the next tile is in flight during the last
//     MFMAs and the epilogue of the current one (the one-tile kernel pays a bare DMA latency at every tile start);
//   * DMA addressing is a wave-uniform 64-bit base (SGPRs) + a per-lane 32-bit offset that only changes with the
//     tile or the W tap, so a K step issues its 8 pieces with no vector address math; the H taps of a row are
//     packed into one 64-bit scalar per tile;
//   * the epilogue is wave-private: each wave transposes its 16-row slabs through its own LDS strip (outside the
//     ring) with inline-asm ds ops - no workgroup barrier and no compiler-inserted vmcnt(0) drain; the wait at
//     the next tile's first step counts the epilogue's stores (vmcnt is in issue order) instead of draining them.
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

// position in the tile order; next() is the carry chain.  GEMM: xt is the M-tile index.
struct Tile {
  int nt, xt, px, Y, b;
};

template <int MODE>
__device__ __forceinline__ bool next_tile(Tile& t, int tiles_n, int tiles_x, int rows) {  // true when Y changed
  if (++t.nt < tiles_n) return false;
  t.nt = 0;
  if (MODE == MODE_GEMM) { ++t.xt; return false; }
  if (++t.xt < tiles_x) return false;
  t.xt = 0;
  if (MODE == MODE_UP) {
    if (++t.px < 2) return false;
    t.px = 0;
  }
  if (++t.Y == rows) { t.Y = 0; ++t.b; }
  return true;
}

// H taps of output row Y: up to 6 entries (src row << 2 | ky), 10 bits each, first tap in the low bits
template <int MODE>
__device__ __forceinline__ int pack_htaps(int adj, int Y, int Hc, unsigned long long& list) {
  list = 0;
  if (MODE == MODE_GEMM) return 1;
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

// waves are WM x WN over the tile; WM = 4 for 64-channel N tiles so that a wave's strip rows are whole 128-byte
// pixel rows (two waves writing 64-byte halves of each line measured 2x slower on the N = 64 layers)
template <int BM, int BN> struct Cfg {
  static constexpr int WM = (BM == 128 && BN == 64) ? 4 : 2;
  static constexpr int WN = 4 / WM;
  static constexpr int OCC = 2;                           // workgroups per CU (3 measured slower on every layer)
};

template <typename T, int BM, int BN, int MODE>
__global__ __launch_bounds__(256, (Cfg<BM, BN>::OCC)) void conv_kernel(ConvP p, int tiles_n, int tiles_x, int ntiles, int dbg) {
  constexpr int SB = 128, NS = 2;              // bytes of K per tile row per stage; LDS stages
  constexpr int ES = sizeof(T);
  constexpr int BK = SB / ES;
  constexpr int EPC = 16 / ES;
  constexpr int WM = Cfg<BM, BN>::WM, WN = Cfg<BM, BN>::WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // 32 x 32 MFMA blocks per wave
  constexpr int RPI = 1024 / SB;               // tile rows per DMA piece (1 KiB per wave instruction)
  constexpr int CPR = SB / 16;
  constexpr int KS = SB / 32;
  constexpr int STAGE = (BM + BN) * SB;
  constexpr int IA = BM / RPI / 4, IB = BN / RPI / 4;
  constexpr int WC = BN / WN;                  // epilogue strip of one wave: SR pixel rows x WC channels (+16 B pad)
  constexpr int RB = WC * ES;
  constexpr int RS = RB + 16;
  constexpr int CR = RB / 16;                  // 16-byte chunks per strip row
  constexpr int SR = (Cfg<BM, BN>::OCC == 3 && 8 * CR >= 64) ? 8 : 16;  // strip rows (8 keeps 3 workgroups per CU)
  constexpr int CPL = SR * CR / 64;            // chunks per lane per strip
  constexpr int STRIP = SR * RS;
  constexpr int NQ = 32 / SR;                  // strips per 32-row MFMA block
  constexpr int NST = TM * NQ * CPL;           // global stores per wave per tile epilogue
  constexpr int nW = MODE == MODE_S2 ? 4 : (MODE == MODE_UP ? 2 : 1);
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE + 4 * STRIP + BN * 4];

  // ---- this workgroup's chunk of the tile order
  const int G = gridDim.x, g = blockIdx.x;
  const int tq = ntiles / G, tr = ntiles % G;
  const int t0 = g * tq + (g < tr ? g : tr);
  const int tcount = tq + (g < tr ? 1 : 0);
  if (tcount == 0) return;

  const int tid = threadIdx.x;
  const int KC = p.K / BK;
  const int Ws = MODE == MODE_S2 ? 2 * p.Wc : p.Wc, cmul = MODE == MODE_S2 ? 2 : 1;
  const int Wo = MODE == MODE_S2 ? p.Wc : 2 * p.Wc;
  const int rows = MODE == MODE_S2 ? p.Hc : 2 * p.Hc;

  Tile first;
  {
    int mt = t0 / tiles_n;
    first.nt = t0 % tiles_n;
    first.px = 0; first.Y = 0; first.b = 0;
    if (MODE == MODE_GEMM) first.xt = mt;
    else {
      first.xt = mt % tiles_x; mt /= tiles_x;
      if (MODE == MODE_UP) { first.px = mt & 1; mt >>= 1; }
      first.Y = mt % rows; first.b = mt / rows;
    }
    first.nt = __builtin_amdgcn_readfirstlane(first.nt); first.xt = __builtin_amdgcn_readfirstlane(first.xt);
    first.px = __builtin_amdgcn_readfirstlane(first.px); first.Y = __builtin_amdgcn_readfirstlane(first.Y);
    first.b = __builtin_amdgcn_readfirstlane(first.b);
  }

  const T* in = (const T*)p.in;
  const T* w = (const T*)p.w;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lrow = lane / CPR, pos = lane % CPR;
  auto swz = [](int row) { return (row >> 1) & 7; };

  // ---- issue side: runs one K step ahead of the compute side
  Tile ti = first;
  int i_left = tcount;                         // tiles not yet fully issued (including the current one)
  unsigned long long hl;                       // remaining H taps of the current tile
  int h_left;
  unsigned long long hl_row;                   // tap list of row ti.Y (reused by every tile of the row)
  int nh_row = pack_htaps<MODE>(p.adj, ti.Y, p.Hc, hl_row);
  int it_j = 0, it_kc = 0;
  unsigned voffA[IA], voffB[IB];
  int rowA[IA], pcA[IA];
  const char* sA = nullptr;
  const char* sB = nullptr;
#pragma unroll
  for (int u = 0; u < IA; ++u) {
    rowA[u] = (wave + 4 * u) * RPI + lrow;
    pcA[u] = (pos ^ swz(rowA[u])) * 16;
  }
#pragma unroll
  for (int u = 0; u < IB; ++u) {
    const int row = (wave + 4 * u) * RPI + lrow;
    voffB[u] = (unsigned)(row * (int)p.w_sn * ES + (pos ^ swz(row)) * 16);
  }
  auto set_wtap = [&]() {                      // after the tile, the H tap or the W tap changed
    const int it_r = (int)(hl & 1023) >> 2, it_ky = (int)hl & 3;
    int coff = 0, kx = 0;
    if (MODE == MODE_S2) { coff = it_j - 1; kx = it_j; }
    else if (MODE == MODE_UP) {
      if (ti.px == 0) { coff = it_j == 0 ? 0 : -1; kx = it_j == 0 ? 1 : 3; }
      else { coff = it_j == 0 ? 1 : 0; kx = it_j == 0 ? 0 : 2; }
    }
    const int wt = MODE == MODE_GEMM ? 0 : it_ky * 4 + kx;
    sB = (const char*)(w + (long)wt * p.w_st + (long)(ti.nt * BN) * p.w_sn);
    if (MODE == MODE_GEMM) {
      sA = (const char*)in;
#pragma unroll
      for (int u = 0; u < IA; ++u) {
        int br = ti.xt * BM + rowA[u];
        if (br >= p.B) br = p.B - 1;           // rows past the batch: duplicate data, dropped in the epilogue
        voffA[u] = (unsigned)(br * (int)p.in_sb * ES + pcA[u]);
      }
    } else {
      sA = (const char*)(in + (long)ti.b * p.in_sb + (long)it_r * Ws * p.in_sp);
#pragma unroll
      for (int u = 0; u < IA; ++u) {
        int c = cmul * (ti.xt * BM + rowA[u]) + coff;
        if (c < 0) c += Ws; else if (c >= Ws) c -= Ws;
        voffA[u] = (unsigned)(c * (int)p.in_sp * ES + pcA[u]);
      }
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
  auto issue_dma = [&](int st) {
    unsigned char* base = lds + st * STAGE;
    const unsigned k0b = (unsigned)it_kc * SB;
#pragma unroll
    for (int u = 0; u < IA; ++u) dma16(sA + k0b + voffA[u], base + (wave + 4 * u) * 1024);
#pragma unroll
    for (int u = 0; u < IB; ++u) dma16(sB + k0b + voffB[u], base + BM * SB + (wave + 4 * u) * 1024);
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
  // of MFMA k-step ks+1 are issued before the MFMAs of ks, counted lgkmcnt + sched_barrier keep the order.
  i32x4 fa[2][TM], fb[2][TN];
  auto read_frags = [&](int set, unsigned a_addr, unsigned b_addr) {
    LDS_READ128(fa[set][0], a_addr, 0);
    if constexpr (TM == 2) LDS_READ128(fa[set][1], a_addr, 32 * SB);
    LDS_READ128(fb[set][0], b_addr, 0);
    if constexpr (TN == 2) LDS_READ128(fb[set][1], b_addr, 32 * SB);
  };
  auto compute = [&](unsigned st_off) {
    read_frags(0, fragA[0] + st_off, fragB[0] + st_off);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
        read_frags((ks + 1) & 1, fragA[ks + 1] + st_off, fragB[ks + 1] + st_off);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          mma_tile((const T*)nullptr, fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue constants
  const unsigned strip = lds0 + NS * STAGE + wave * STRIP;
  float* s_db = (float*)(lds + NS * STAGE + 4 * STRIP);
  const bool want_db = p.dbias != nullptr;
  T* out = (T*)p.out;
  const int part = lane % CR;                  // this lane's 16-byte chunk of the strip rows (64 % CR == 0)
  const unsigned wr_base = strip + (4 * lh) * RS + lr * ES;
  const unsigned rd_base = strip + (lane / CR) * RS + part * 16;
  const bool counted = MODE != MODE_GEMM && !want_db && !(dbg & 4);  // epilogue store count per wave is exactly NST

  issue_dma(0);
  advance();
  unsigned gs = 0;                             // K-step counter of this workgroup; step gs lives in stage gs & 1
  Tile tc = first;
  unsigned long long dummy;
  int nh_c = pack_htaps<MODE>(p.adj, tc.Y, p.Hc, dummy);
  for (int c = 0; c < tcount; ++c) {
    const int nsteps = nh_c * nW * KC;
    if (want_db && tid < BN) s_db[tid] = 0.f;
    for (int s = 0; s < nsteps; ++s, ++gs) {
      // step gs has landed (only the previous epilogue's stores may still be in flight); everyone finished reading
      // the other stage -> refill it with step gs+1, then compute step gs
      if (counted && s == 0 && c > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i_left > 0) {
        if (!(dbg & 1)) issue_dma((gs + 1) & 1);
        advance();
      }
      if (!(dbg & 2)) compute((gs & 1) * STAGE);
    }
    if (!(dbg & 4)) {
      // ---- wave-private epilogue: TM x NQ strips of SR rows; lane (lr, lh) owns channel column lr of rows
      //      (e & 3) + 8 (e >> 2) + 4 lh of each 32 x 32 MFMA block
      const int colb = tc.nt * BN + wn * WC;   // first channel of this wave's strip
      const int n0 = tc.xt * BM;
      long toff;                               // element offset of (tile row 0, strip column 0)
      if (MODE == MODE_GEMM) toff = (long)n0 * p.out_sb + colb;
      else toff = (long)tc.b * p.out_sb + ((long)tc.Y * Wo + (MODE == MODE_S2 ? n0 : 2 * n0 + tc.px)) * p.out_sp + colb;
      char* obase = (char*)(out + toff);
      const char* abase = (const char*)((const T*)p.aux + toff);
      float csum[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) csum[e] = 0.f;
      float bias[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = colb + j * 32 + lr;
        bias[j] = (p.bias && n < p.N) ? p.bias[n % p.bias_mod] : 0.f;
      }
      // byte offsets of this lane's chunks from the tile's wave-uniform base, and the leaky-relu mask source: all
      // aux loads of the tile are issued up front (one exposed latency per tile, not one per strip)
      unsigned off[TM][NQ][CPL];
      bool ok[TM][NQ][CPL];
      uint4 araw[TM][NQ][CPL];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int u = 0; u < CPL; ++u) {
            const int trow = wm * (BM / WM) + i * 32 + q * SR + u * (64 / CR) + lane / CR;
            ok[i][q][u] = true;                // N % BN == 0 (launch_dtype), so only GEMM rows past the batch drop
            if (MODE == MODE_GEMM) {
              ok[i][q][u] = n0 + trow < p.B;
              off[i][q][u] = (unsigned)((trow * (int)p.out_sb + part * EPC) * ES);
            } else {
              off[i][q][u] = (unsigned)(((MODE == MODE_S2 ? trow : 2 * trow) * (int)p.out_sp + part * EPC) * ES);
            }
            if (p.epi == EPI_MASK)
              araw[i][q][u] = ok[i][q][u] ? *(const uint4*)(abase + off[i][q][u]) : make_uint4(0, 0, 0, 0);
          }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
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
#pragma unroll
          for (int u = 0; u < CPL; ++u) {
            if (MODE == MODE_GEMM && !ok[i][q][u]) continue;
            T* v = (T*)&raw[u];
            if (p.epi == EPI_MASK) {
              const T* av = (const T*)&araw[i][q][u];
#pragma unroll
              for (int e = 0; e < EPC; ++e) {
                const float f = (float)v[e] * ((float)av[e] > 0.f ? SQRT2 : LRELU_SLOPE * SQRT2);
                v[e] = (T)f;
              }
            }
            if (want_db) {
#pragma unroll
              for (int e = 0; e < EPC; ++e) csum[e] += (float)v[e];
            }
            *(i32x4*)(obase + off[i][q][u]) = raw[u];
          }
        }
      }
      if (want_db) {
        // lanes with equal `part` inside a wave are CR lanes apart
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float v = csum[e];
          for (int d = CR; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
          if (lane < CR) atomicAdd(&s_db[wn * WC + part * EPC + e], v);
        }
        __syncthreads();
        if (tid < BN && tc.nt * BN + tid < p.N) {
          const float rs = p.rowscale ? p.rowscale[tc.b] : 1.f;
          atomicAdd(&p.dbias[(tc.nt * BN + tid) % p.bias_mod], s_db[tid] * rs);
        }
      }
    }
    zero_acc();
    if (next_tile<MODE>(tc, tiles_n, tiles_x, rows)) nh_c = pack_htaps<MODE>(p.adj, tc.Y, p.Hc, dummy);
  }
}

template <typename T, int BM, int BN, int MODE>
int launch(const ConvP* p, hipStream_t stream) {
  const int tiles_n = (p->N + BN - 1) / BN;
  int tiles_x = 1;
  long tiles_m;
  if (MODE == MODE_S2) { tiles_x = p->Wc / BM; tiles_m = (long)p->B * p->Hc * tiles_x; }
  else if (MODE == MODE_UP) { tiles_x = p->Wc / BM; tiles_m = (long)p->B * 2 * p->Hc * 2 * tiles_x; }
  else tiles_m = (p->B + BM - 1) / BM;
  const long ntiles = tiles_m * tiles_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffL) return DG_EINVAL;
  static int resident = 0;                     // workgroups the device holds at once
  if (!resident) {
    int occ = 0, dev = 0, cus = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    HIP_CHECK_RET(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_CHECK_RET(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_kernel<T, BM, BN, MODE>, 256, 0));
    if (occ < 1) occ = 1;
    resident = occ * cus;
    const char* e = getenv("DG_CONV_WGS");
    if (e && atoi(e) > 0) resident = atoi(e);
  }
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("DG_CONV_DBG"); dbg = e ? atoi(e) : 0; }
  const int G = (int)(ntiles < resident ? ntiles : resident);
  conv_kernel<T, BM, BN, MODE><<<(unsigned)G, 256, 0, stream>>>(*p, tiles_n, tiles_x, (int)ntiles, dbg);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

template <typename T>
int launch_dtype(const ConvP* p, hipStream_t stream) {
  const bool n128 = p->N % 128 == 0;
  if (p->mode == MODE_GEMM) return n128 ? launch<T, 64, 128, MODE_GEMM>(p, stream) : launch<T, 64, 64, MODE_GEMM>(p, stream);
  const bool m128 = p->Wc % 128 == 0;
  if (p->mode == MODE_S2) {
    if (m128 && n128) return launch<T, 128, 128, MODE_S2>(p, stream);
    if (m128) return launch<T, 128, 64, MODE_S2>(p, stream);
    if (n128) return launch<T, 64, 128, MODE_S2>(p, stream);
    return launch<T, 64, 64, MODE_S2>(p, stream);
  }
  if (m128 && n128) return launch<T, 128, 128, MODE_UP>(p, stream);
  if (m128) return launch<T, 128, 64, MODE_UP>(p, stream);
  if (n128) return launch<T, 64, 128, MODE_UP>(p, stream);
  return launch<T, 64, 64, MODE_UP>(p, stream);
}

}  // namespace persist
