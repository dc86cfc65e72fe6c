// Weight-gradient GEMM on the matrix cores with an LDS-DMA ring (bf16, Down = wmode 0, Up = wmode 1):
//   dW[tap][ci][co] += scale * sum_b rs[b] sum_pixels A[src_a][ci] G[src_g][co]
// Autograd counterpart in the reference: the weight gradients of nn.Conv2d / nn.ConvTranspose2d inside EqualLR
// (models/gans/dcgan_eqlr.py:9,24,80; models/ops/common.py:132-133), incl. the R1 double-backward terms.
//
// Same GEMM view and fragment reads as wgrad_mfma.hip (M = ci, N = co, K = pixels, ds_read_b64_tr_b16), but the tiles
// go global -> LDS by LDS-DMA into a 4-stage ring of 32-pixel chunks with counted vmcnt waits and ONE raw barrier per
// chunk: the register-staged kernel is bound by the exposed load latency of its short chunks (its waves are parked
// 33 % and issue address arithmetic 43 % of the time, profiles/r01f_pmc_sq_summary.csv); the ring keeps three chunks in
// flight with no staging registers and no per-chunk 64-bit address math.
//   * LDS rows are unpadded (DMA pieces are lane-linear); the 4-row x 64-byte footprint of a transposing read is
//     spread over the banks by an XOR swizzle of the 64-byte column group with the row (applied on the DMA source
//     address, undone in the read address, which only depends on the lane);
//   * transposing reads are inline asm (a compiler-visible LDS read behind an in-flight DMA drains vmcnt(0));
//   * a workgroup owns a PAIR of W taps (kx, kx + 2): both read the same G pixels and A pixels one column apart, so a
//     chunk is one G tile + one A image of 64 + 1 pixels, the G fragments feed both taps' MFMAs, and the LDS - which
//     bounded the one-tap version (fragment reads ~50 % + DMA writes ~50 % of its bandwidth at 40 % MFMA use) - moves
//     half the DMA bytes and three quarters of the fragment bytes per flop.
#include "common.h"

#include <stdlib.h>

#include <type_traits>

// (experiment knobs, `make variant VSRC=wgrad_mfma_dma VFLAGS=-DDG_WG_BM_MAX=64 ...`: the largest tile of a group launch's layers)
#ifndef DG_WG_BM_MAX
#define DG_WG_BM_MAX 128
#endif
#ifndef DG_WG_BN_MAX
#define DG_WG_BN_MAX 128
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BKP = 64;  // coarse pixels per ring stage
constexpr int NS = 2;    // ring stages
constexpr int PAIR_MIN_WS = 24;  // tap pairs with a split-K workspace: chunks a workgroup must still walk

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

#define DG_WAITV(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#ifdef DG_WG_DIAG
#define TR16(dst, addr, imm) do { if (!(DG_WG_DIAG & 16)) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory"); else asm volatile("" : "=v"(dst)); } while (0)
#else
#define TR16(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#endif

// NT = W taps per workgroup: 1, or 2 = the pair (kx, kx + 2) on one G tile and one A image of 64 + 1 pixels
#ifdef DG_WG_DIAG
constexpr int wg_dbg = DG_WG_DIAG;   // ablation builds (make wgdiag WGDIAG=bits): 1 no DMA, 2 no MFMA, 4 no epilogue, 16 no LDS reads
#else
constexpr int wg_dbg = 0;
#endif
template <int BM, int BN, int NT>
constexpr int stage_bytes() { return BKP * BM * 2 + (NT == 2 ? 1024 : 0) + BKP * BN * 2; }

// One workgroup of the (tile, tap or tap pair, K split) grid gx x gy x gz, linear index `id`; `lds`: NS stages.  The body of
// wgrad_dma_kernel (one layer per launch) and of wgrad_group_kernel (several layers' workgroups in ONE launch, below).
// X2 (round 5, the fp32x3 precision mode on THIS kernel): a and g are DG_BF16X2 - per 64 channels 128 bytes of hi = bf16(x),
// then 128 bytes of lo = bf16(x - hi) (include/dusty_gan_hip.h).  Every 64-pixel chunk is staged and multiplied three times,
// (a_hi, g_hi), (a_lo, g_hi), (a_hi, g_lo): only the pieces' source addresses change (the lo half of a channel group sits 128
// bytes behind its hi half), stages, fragment reads and matrix instructions are the bf16 kernel's.  Strides arrive in bf16
// units (twice the elements: the launcher).
template <int WMODE, int BM, int BN, int NT, bool X2 = false>
__device__ __forceinline__ void wgrad_dma_body(const WgradP& p, const int tiles_n, const int accumulate, const int id,
                                               const int gx, const int gy, const int gz, unsigned char* lds) {
  constexpr int RA = BM * 2, RG = BN * 2;                 // LDS row bytes (one pixel)
  constexpr int STA = BKP * RA + (NT == 2 ? 1024 : 0), STG = BKP * RG, STAGE = STA + STG;   // pair: 64 + 1 pixels (one more piece of rows)
  constexpr int PA = BKP * RA / 1024 / 4, PG = STG / 1024 / 4;  // DMA pieces per wave per stage (+ the A image's last piece: wave 0)
  constexpr int TM = BM / 64, TN = BN / 64;
  static_assert(PA >= 1 && PG >= 1, "tile too small for the piece distribution");
  static_assert(STAGE == stage_bytes<BM, BN, NT>(), "stage size");

  const int tid = threadIdx.x;
  // XCD-aware, bijective remap of the (tile, tap, split) grid (blocks id and id+8 share an XCD): the 16 taps x tiles of
  // one K split read the same pixel rows, so each XCD gets a contiguous range of splits and fetches their rows once
  const int nwg = gx * gy * gz;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int bx = logical % gx, by = (logical / gx) % gy, bz = logical / (gx * gy);
  const int ct = bx % tiles_n, mt = bx / tiles_n;
  const int ci0 = mt * BM, co0 = ct * BN;
  const int ky = NT == 2 ? by >> 1 : by >> 2, kxp = NT == 2 ? by & 1 : by & 3;   // pair: kx = kxp and kxp + 2; single: kx = kxp
  const long units = (long)p.B * p.Hc;
  const long u0 = units * bz / gz, u1 = units * (bz + 1) / gz;
  const int cpr = p.Wc / BKP;                             // chunks per row
  const long nchunks64 = (u1 - u0) * cpr;
  const int Wa = WMODE == 0 ? 2 * p.Wc : p.Wc;
  const int Wg = WMODE == 1 ? 2 * p.Wc : p.Wc;
  const bf16* A = (const bf16*)p.a;
  const bf16* G = (const bf16*)p.g;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int asp = (int)p.a_sp, gsp = (int)p.g_sp;

  // ---- DMA side.  Piece = 1 KiB = 64 lanes x 16 B = (1024 / row bytes) whole rows; lane -> (row in piece, chunk);
  //      the chunk's 64-byte group is XORed with the row (256-byte rows: row & 3; 128-byte rows: (row >> 1) & 1).
  //      All loop state is 32-bit and advanced incrementally (the first version divided 64-bit unit indices per chunk
  //      and rebuilt 64-bit addresses per piece: SQ_INSTS_SALU was 8.8x, SQ_INSTS_VALU 7.8x SQ_INSTS_MFMA, the matrix
  //      pipe 27 % busy); a piece is one saddr-form LDS-DMA: wave-uniform row base + per-lane 32-bit offset.
  constexpr int CA = RA / 16, CG = RG / 16;               // 16-byte chunks per row
  constexpr int RPA = 64 / CA, RPG = 64 / CG;             // rows per piece
  auto swz = [](int row, int rowbytes) { return rowbytes == 256 ? (row & 3) : ((row >> 1) & 1); };
  int rowA[PA], rowG[PG];
  unsigned chA[PA], chG[PG];                              // byte offset of the lane's (swizzled) chunk inside its pixel
  // (X2: chunk c of the tile's hi - or lo - half lives in channel group c / 8, 256 bytes per group)
  auto chunk_off = [](int c) { return (unsigned)(X2 ? (c >> 3) * 256 + (c & 7) * 16 : c * 16); };
#pragma unroll
  for (int v = 0; v < PA; ++v) {
    const int r = (wave + 4 * v) * RPA + lane / CA, c = lane % CA;
    rowA[v] = r;
    chA[v] = chunk_off((((c >> 2) ^ swz(r, RA)) << 2) | (c & 3));
  }
#pragma unroll
  for (int v = 0; v < PG; ++v) {
    const int r = (wave + 4 * v) * RPG + lane / CG, c = lane % CG;
    rowG[v] = r;
    chG[v] = chunk_off((((c >> 2) ^ swz(r, RG)) << 2) | (c & 3));
  }
  const int rowX = BKP + lane / CA;                       // the image's last piece (rows 64 ..): wave 0
  const unsigned chX = chunk_off(((((lane % CA) >> 2) ^ swz(rowX, RA)) << 2) | (lane % CA & 3));
  const unsigned aspb = (unsigned)asp * 2u, gspb = (unsigned)gsp * 2u;   // bytes per pixel (< 2^24)
  // A image row m of a chunk starting at coarse column x0 holds A-grid column amul (x0 + m) + da; the pair's taps read
  // image rows r + 0 and r + 1 for chunk pixel r:
  //   wmode 0 (Down): columns 2x + kx - 1:  kx = kxp -> row r, kx = kxp + 2 -> row r + 1;         da = kxp - 1
  //   wmode 1 (Up):   columns x + d(kx), d = +1, 0, 0, -1:  pair 0: kx 2 -> r, kx 0 -> r + 1 (da = 0);
  //                                                         pair 1: kx 3 -> r, kx 1 -> r + 1 (da = -1)
  //   single tap kx: image row r holds the tap's own column (Down: 2x + kx - 1; Up: x + d(kx))
  const int da = NT == 2 ? (WMODE == 0 ? kxp - 1 : -kxp)
                         : (WMODE == 0 ? kxp - 1 : (kxp == 0 ? 1 : (kxp == 3 ? -1 : 0)));
  const int kx_s0 = NT == 2 ? (WMODE == 0 ? kxp : kxp + 2) : kxp, kx_s1 = WMODE == 0 ? kxp + 2 : kxp;
  const int dg = WMODE == 1 ? (NT == 2 ? 1 - kxp : ((kxp == 0 || kxp == 2) ? 1 : 0)) : 0;   // column parity on the G grid
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma_s = [&](unsigned voff, const char* sbase, unsigned ldsaddr) __attribute__((always_inline)) {
    if (wg_dbg & 1) return;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsaddr), "v"(voff), "s"(sbase) : "memory");
  };
  const int nchunks = (int)nchunks64;
  int ib = (int)(u0 / p.Hc), im = (int)(u0 % p.Hc), ixc = 0;   // issue position: sample, coarse row, chunk in row
  const char* abase = nullptr;
  const char* gbase = nullptr;
  auto set_unit = [&](int b, int m) __attribute__((always_inline)) {
    int rowa, rowg;
    dg_wgrad1d(WMODE, 0, m, p.Hc, ky, rowa, rowg);
    // (readfirstlane: the "s" operands of the DMA statement must be provably wave-uniform)
    auto uni = [](const bf16* q) __attribute__((always_inline)) {
      const unsigned long long v = (unsigned long long)q;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return (const char*)(((unsigned long long)hi << 32) | lo);
    };
    const int bg = p.g_mod > 0 ? b % p.g_mod : b;          // (one launch over real | fake | tangent input samples)
    abase = uni(A + (long)b * p.a_sb + (long)rowa * Wa * asp + ci0 * (X2 ? 2 : 1));
    gbase = uni(G + (long)bg * p.g_sb + (long)rowg * Wg * gsp + co0 * (X2 ? 2 : 1));
  };
  set_unit(ib, im);
  int iseg = 0;                                             // X2: the chunk's pass (a_hi g_hi | a_lo g_hi | a_hi g_lo)
  auto issue = [&](unsigned st_off) __attribute__((always_inline)) {
    const unsigned base = lds0 + st_off + (unsigned)wave * 1024u;
    const int x0 = ixc * BKP;
    const char* ab = abase + ((X2 && iseg == 1) ? 128 : 0);
    const char* gb = gbase + ((X2 && iseg == 2) ? 128 : 0);
#pragma unroll
    for (int v = 0; v < PA; ++v) {                          // circular columns: Wa is a power of two (launcher check)
      const unsigned ca = (unsigned)(((WMODE == 0 ? 2 : 1) * (x0 + rowA[v]) + da) & (Wa - 1));
      dma_s(__umul24(ca, aspb) + chA[v], ab, base + 4 * v * 1024);
    }
    if (NT == 2 && wave == 0) {
      const unsigned ca = (unsigned)(((WMODE == 0 ? 2 : 1) * (x0 + rowX) + da) & (Wa - 1));
      dma_s(__umul24(ca, aspb) + chX, ab, lds0 + st_off + 4 * PA * 1024);
    }
#pragma unroll
    for (int v = 0; v < PG; ++v) {
      const unsigned cg = (unsigned)((WMODE == 1 ? 2 : 1) * (x0 + rowG[v]) + dg);
      dma_s(__umul24(cg, gspb) + chG[v], gb, base + STA + 4 * v * 1024);
    }
    if (X2 && ++iseg < 3) return;
    iseg = 0;
    if (++ixc == cpr) {
      ixc = 0;
      if (++im == p.Hc) { im = 0; ++ib; }
      set_unit(ib, im);                                     // (one unit past the end at the very last chunk: never used)
    }
  };

  // ---- read side: lane -> (row block q, column quad pp) inside its 16-lane group; group -> (k half, column block)
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int g16 = lane >> 4, i16 = lane & 15;
  const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pp = i16 & 3;
  unsigned offA[NT][TM], offG[TN];                         // byte offset of this lane's lo read at kq = 0, stage 0 (A: per tap)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int colb = (wm * (BM / 2) + i * 32 + 16 * cb + 4 * pp) * 2;
      offA[t][i] = lds0 + (8 * kh + q + t) * RA + ((((colb >> 6) ^ swz(q + t, RA)) << 6) | (colb & 63));
    }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int colb = (wn * (BN / 2) + j * 32 + 16 * cb + 4 * pp) * 2;
    offG[j] = lds0 + STA + (8 * kh + q) * RG + ((((colb >> 6) ^ swz(q, RG)) << 6) | (colb & 63));
  }

  f32x16 acc[NT][TM][TN];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

  int cb_s = (int)(u0 / p.Hc), cm = (int)(u0 % p.Hc), cxc = 0;   // compute position (sample for the per-sample weight)
  float cur_rs = 1.f;
  if (p.rowscale && nchunks > 0) {
    cur_rs = p.rowscale[cb_s];
    if (fabsf(cur_rs) < 1e-30f) cur_rs = cur_rs < 0.f ? -1e-30f : 1e-30f;
  }

  // ---- ring of NS = 2 stages: chunk s lives in stage s & 1; at the top of chunk s wait until it landed, barrier
  //      (everyone's share landed, everyone finished reading the other stage), refill the other stage with chunk s + 1,
  //      compute chunk s
  if (nchunks > 0) issue(0);
  unsigned so = 0;                                          // LDS offset of the stage being computed
  const int npass = nchunks * (X2 ? 3 : 1);
  int cseg = 0;
  for (int s = 0; s < npass; ++s) {
    DG_WAITV(0);
    __builtin_amdgcn_s_barrier();
    if (s + 1 < npass) issue(so ^ (unsigned)STAGE);
    // the reads of k-step kq+1 are issued before the MFMAs of kq (two fragment sets, counted lgkmcnt)
    i32x2 alo[2][NT][TM], ahi[2][NT][TM], glo[2][TN], ghi[2][TN];   // [set][tap][..]
    unsigned ra[NT][TM], rg[TN];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i) ra[t][i] = offA[t][i] + so;
#pragma unroll
    for (int j = 0; j < TN; ++j) rg[j] = offG[j] + so;
#define DG_READ_SET(kq)                                            \
  do {                                                             \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {               \
      TR16(alo[(kq) & 1][t][i], ra[t][i], (kq) * 16 * RA);         \
      TR16(ahi[(kq) & 1][t][i], ra[t][i], (kq) * 16 * RA + 4 * RA);\
    }                                                              \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {               \
      TR16(glo[(kq) & 1][j], rg[j], (kq) * 16 * RG);               \
      TR16(ghi[(kq) & 1][j], rg[j], (kq) * 16 * RG + 4 * RG);      \
    }                                                              \
  } while (0)
    auto mfmas = [&](int set) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const i32x4 fg = {glo[set][j][0], glo[set][j][1], ghi[set][j][0], ghi[set][j][1]};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const i32x4 fa = {alo[set][t][i][0], alo[set][t][i][1], ahi[set][t][i][0], ahi[set][t][i][1]};
            if (!(wg_dbg & 2))
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa),
                                                                     __builtin_bit_cast(bf16x8, fg), acc[t][i][j], 0, 0, 0);
            else asm volatile("" ::"v"(fa), "v"(fg));
          }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(BKP / 16 == 4, "four k-steps per chunk");
    DG_READ_SET(0);
    DG_READ_SET(1);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NT * TM + TN)) : "memory");
    mfmas(0);
    DG_READ_SET(2);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NT * TM + TN)) : "memory");
    mfmas(1);
    DG_READ_SET(3);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NT * TM + TN)) : "memory");
    mfmas(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mfmas(1);
    so ^= (unsigned)STAGE;
    // per-sample weights: running sum kept divided by the current sample's weight (wgrad_mfma.hip)
    if (X2 && ++cseg < 3) continue;
    cseg = 0;
    if (++cxc == cpr) {
      cxc = 0;
      if (++cm == p.Hc) {
        cm = 0;
        ++cb_s;
        if (p.rowscale && s + 1 < npass) {
          float rn = p.rowscale[cb_s];
          if (fabsf(rn) < 1e-30f) rn = rn < 0.f ? -1e-30f : 1e-30f;
          const float ratio = cur_rs / rn;
          cur_rs = rn;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[t][i][j] *= ratio;
        }
      }
    }
  }

  // D layout: col = lane & 31 (co), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (ci)
  // split-K partial tiles: plain stores into the caller's workspace (summed by dg_wgrad_reduce), or fp32 atomics onto dw
  float* const obase = p.ws ? p.ws + (long)bz * 16 * p.Ci * p.Co : p.dw;
  const bool plain = p.ws || !accumulate;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float* dw = obase + (long)(ky * 4 + (t == 0 ? kx_s0 : kx_s1)) * p.Ci * p.Co;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = ci0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const int co = co0 + wn * (BN / 2) + j * 32 + lr;
          const float v = acc[t][i][j][e] * (p.scale * cur_rs);
          float* dst = dw + (long)ci * p.Co + co;
          if (wg_dbg & 4) { asm volatile("" ::"v"(v)); continue; }
          if (plain) *dst = v;
          else atomicAdd(dst, v);
        }
  }
}

template <int WMODE, int BM, int BN, int NT, bool X2 = false>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(WgradP p, int tiles_n, int accumulate) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * stage_bytes<BM, BN, NT>()];
  wgrad_dma_body<WMODE, BM, BN, NT, X2>(p, tiles_n, accumulate, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)),
                                        (int)gridDim.x, (int)gridDim.y, (int)gridDim.z, lds);
}

// ---- several layers in ONE launch.  A weight-gradient launch is one residency round of ~512 workgroups that start
// together, fill their rings together (a cold burst of 130 KB per CU) and store their partial tiles together (16-67 MB): at
// 40-100 us per launch a fifth to a quarter of it is that ramp (round-5 stamps of the conv kernel, same structure; the B-sized
// launches of the generator ran at 0.31 of the matrix peak against 0.40 for the 3B-sized ones of the discriminator).  The
// weight gradients of a network's layers are independent of each other - they all read finished activations and gradient
// chains - so their workgroups go into ONE grid: item k owns the blocks [first_k, first_k + pad8(count_k)), and as the
// workgroups of one layer drain, the next layer's take their slots: fill and tail of one layer under the matrix work of
// its neighbours.  Partial tiles, splits and the reduce are exactly those of the single launches (same plan function).
constexpr int GROUP_MAX = 4;
struct GroupItem { WgradP p; int tiles_n, gx, gy, gz, first, variant, accumulate; };
struct GroupP { GroupItem it[GROUP_MAX]; int n; };
constexpr int variant_code(int wmode, int bm, int bn, int nt, bool x2 = false) {
  return (x2 ? 16 : 0) + wmode * 8 + (bm == 128 ? 4 : 0) + (bn == 128 ? 2 : 0) + (nt == 2 ? 1 : 0);
}

__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(GroupP g) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * stage_bytes<128, 128, 2>()];
  // (the item by unrolled scalar selects: a dynamically indexed kernel argument would be copied to scratch, and values that
  //  come back from scratch are no longer provably wave-uniform - the "s" operands of the LDS-DMA statements need that)
  GroupItem it = g.it[0];
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.it[i].first) it = g.it[i];
  const int id = (int)blockIdx.x - it.first;
  if (id >= it.gx * it.gy * it.gz) return;       // (items are padded to multiples of 8 blocks: id & 7 stays the XCD label)
#define DG_GROUP_CASE(W, M, N, T) \
  case variant_code(W, M, N, T): wgrad_dma_body<W, M, N, T>(it.p, it.tiles_n, it.accumulate, id, it.gx, it.gy, it.gz, lds); break; \
  case variant_code(W, M, N, T, true): wgrad_dma_body<W, M, N, T, true>(it.p, it.tiles_n, it.accumulate, id, it.gx, it.gy, it.gz, lds); break;
  switch (it.variant) {
    DG_GROUP_CASE(0, 64, 64, 1) DG_GROUP_CASE(0, 64, 64, 2) DG_GROUP_CASE(0, 64, 128, 1) DG_GROUP_CASE(0, 64, 128, 2)
    DG_GROUP_CASE(0, 128, 64, 1) DG_GROUP_CASE(0, 128, 64, 2) DG_GROUP_CASE(0, 128, 128, 1) DG_GROUP_CASE(0, 128, 128, 2)
    DG_GROUP_CASE(1, 64, 64, 1) DG_GROUP_CASE(1, 64, 64, 2) DG_GROUP_CASE(1, 64, 128, 1) DG_GROUP_CASE(1, 64, 128, 2)
    DG_GROUP_CASE(1, 128, 64, 1) DG_GROUP_CASE(1, 128, 64, 2) DG_GROUP_CASE(1, 128, 128, 1) DG_GROUP_CASE(1, 128, 128, 2)
    default: break;
  }
#undef DG_GROUP_CASE
}

// the launch geometry of one layer: K split, tap pairs or single taps (pairs: 0 = by the K range per workgroup, 1 = always,
// 2 = never; ws: partial tiles go to a workspace)
struct DmaGeo { int tiles_m, tiles_n; long split; bool use_pairs; };
// target: workgroups the launch should have (512 = two per CU for a launch of its own; a layer inside a group launch gets its
// share of the group's workgroups: fewer, longer K ranges and proportionally fewer partial tiles)
template <int BM, int BN>
DmaGeo dma_geo(const WgradP* p, int accumulate, int pairs, bool ws, long target = 512) {
  const int tiles_m = p->Ci / BM, tiles_n = p->Co / BN;
  const long units = (long)p->B * p->Hc;
  const bool can_split = accumulate || ws;
  auto split_for = [&](long tiles) {
    long split = 1;
    if (can_split) {
      split = (target + tiles - 1) / tiles;  // aim at >= 2 workgroups per CU
      if (split > units) split = units;
      if (split < 1) split = 1;
    }
    return split;
  };
  // Tap pairs halve the DMA bytes and the (tile, tap) workgroups, so the same workgroup count needs twice the K split -
  // twice the partial tiles - and each workgroup's pixel range halves.  With atomics (no workspace) measured
  // (scripts/bench_conv.py): Down2 at batch 64 +16 % (64 chunks per workgroup), Up3 equal, the layers left with <= 32
  // chunks per workgroup 5-25 % slower: pairs where a workgroup still walks >= 48 chunks.  With the workspace the partial
  // tiles are plain stores and the threshold is lower (PAIR_MIN_WS).
  const long split2 = split_for((long)tiles_m * tiles_n * 8);
  const long chunks2 = units / split2 * (p->Wc / BKP);
  const bool use_pairs = pairs == 1 ? can_split : (pairs == 2 ? false : (can_split && chunks2 >= (ws ? PAIR_MIN_WS : 48)));
  const long split = use_pairs ? split2 : split_for((long)tiles_m * tiles_n * 16);
  return DmaGeo{tiles_m, tiles_n, split, use_pairs};
}

// DG_BF16X2 operands: the kernel takes their strides in bf16 units
static WgradP x2_strides(const WgradP& p) {
  WgradP q = p;
  if (p.a_dtype == DG_BF16X2) { q.a_sb *= 2; q.a_sp *= 2; q.g_sb *= 2; q.g_sp *= 2; }
  return q;
}

// plan != NULL: describe the launch, launch nothing.
template <int WMODE, int BM, int BN>
int launch_dma(const WgradP* p, int accumulate, int pairs, hipStream_t stream, DgWgradPlan* plan) {
  const bool ws = plan ? true : p->ws != nullptr;          // (a plan describes the launch WITH a workspace)
  const DmaGeo ge = dma_geo<BM, BN>(p, accumulate, pairs, ws);
  const int tiles_m = ge.tiles_m, tiles_n = ge.tiles_n;
  const long split = ge.split;
  if (plan) {
    plan->splits = (int)split;
    plan->ws_floats = split * 16L * p->Ci * p->Co;
    plan->tap_pairs = ge.use_pairs ? 1 : 0;
    return DG_OK;
  }
  const WgradP q = x2_strides(*p);
  const bool x2 = p->a_dtype == DG_BF16X2;
  if (ge.use_pairs) {
    dim3 grid((unsigned)(tiles_m * tiles_n), 8u, (unsigned)split);
    if (x2) wgrad_dma_kernel<WMODE, BM, BN, 2, true><<<grid, 256, 0, stream>>>(q, tiles_n, accumulate);
    else wgrad_dma_kernel<WMODE, BM, BN, 2><<<grid, 256, 0, stream>>>(q, tiles_n, accumulate);
  } else {
    dim3 grid((unsigned)(tiles_m * tiles_n), 16u, (unsigned)split);
    if (x2) wgrad_dma_kernel<WMODE, BM, BN, 1, true><<<grid, 256, 0, stream>>>(q, tiles_n, accumulate);
    else wgrad_dma_kernel<WMODE, BM, BN, 1><<<grid, 256, 0, stream>>>(q, tiles_n, accumulate);
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// dw[i] (+)= sum_s ws[s * numel + i]: one thread per 16 bytes of each layer, the splits' loads in flight eight at a time
// Layers with more than RED_WIDE partial tiles (the thin matrix-core kernels: one tile per block, 512-768 of them over a few
// KB): a block owns 16 elements and its 16 thread groups split the partials, met in LDS in a fixed order - 32-48 loads per
// thread instead of 768 dependent batches for one.
constexpr int RED_WIDE = 64;
constexpr int RED_MAX = 16;   // layers per reduce launch (round 5: bias-gradient rows are items too - eight per network at the benchmark)
struct ReduceItems { DgWgradReduce it[RED_MAX]; int first_block[RED_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceItems r) {
  __shared__ f32x4 s_part[16][17];
  int k = 0;
#pragma unroll
  for (int i = 1; i < RED_MAX; ++i)
    if (i < r.n && (int)blockIdx.x >= r.first_block[i]) k = i;
  const DgWgradReduce it = r.it[k];
  if (it.splits > RED_WIDE) {
    const int el = threadIdx.x & 15, sg = threadIdx.x >> 4;          // element of the block, split group
    const long i4 = (long)(blockIdx.x - r.first_block[k]) * 16 + el;
    const long stride = it.numel / 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (4 * i4 < it.numel) {
      const f32x4* src = (const f32x4*)it.ws + i4;
      int sp = sg;
      for (; sp + 7 * 16 < it.splits; sp += 8 * 16) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(src + (long)(sp + 16 * j) * stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
      }
      for (; sp < it.splits; sp += 16) acc += __builtin_nontemporal_load(src + (long)sp * stride);
    }
    s_part[sg][el] = acc;
    __syncthreads();
    if (sg == 0 && 4 * i4 < it.numel) {
      f32x4 t = s_part[0][el];
#pragma unroll
      for (int j = 1; j < 16; ++j) t += s_part[j][el];
      f32x4* dst = (f32x4*)it.dw + i4;
      if (it.accumulate) t += *dst;
      *dst = t;
    }
    return;
  }
  const long i4 = (long)(blockIdx.x - r.first_block[k]) * 256 + threadIdx.x;
  if (4 * i4 >= it.numel) return;
  const f32x4* src = (const f32x4*)it.ws + i4;
  const long stride = it.numel / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= it.splits; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(src + (long)(s + j) * stride);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; s < it.splits; ++s) acc += __builtin_nontemporal_load(src + (long)s * stride);
  f32x4* dst = (f32x4*)it.dw + i4;
  if (it.accumulate) acc += *dst;
  *dst = acc;
}

}  // namespace

int dg_wgrad_mfma_dma_supported(const WgradP* p) {
  const bool x2 = p->a_dtype == DG_BF16X2;
  if (p->a_dtype != p->g_dtype || (p->a_dtype != DG_BF16 && !x2)) return 0;
  if (x2 && (p->a_sp % 64 || p->g_sp % 64 || p->a_sb % 64 || p->g_sb % 64 || (((size_t)p->a | (size_t)p->g) & 255))) return 0;
  if (p->wmode != 0 && p->wmode != 1) return 0;
  if (!p->ring || p->a_sc != 1 || p->g_sc != 1) return 0;
  if (p->Ci % 64 != 0 || p->Co % 64 != 0 || p->Wc % BKP != 0 || p->Hc < 2) return 0;
  if (p->a_sp % 8 != 0 || p->g_sp % 8 != 0) return 0;      // 16-byte DMA granules
  if ((p->Wc & (p->Wc - 1)) != 0) return 0;                // circular column wrap by mask
  if (p->a_sp * (x2 ? 4 : 2) >= (1 << 24) || p->g_sp * (x2 ? 4 : 2) >= (1 << 24)) return 0;  // 24-bit offset multiply
  if ((long)p->B * p->Hc * (p->Wc / BKP) > 0x7fffffffL) return 0;
  return 1;
}

int dg_wgrad_mfma_dma_launch(const WgradP* p, int accumulate, int pairs, hipStream_t stream, DgWgradPlan* plan) {
  if (!dg_wgrad_mfma_dma_supported(p)) return DG_EUNSUPPORTED;
  if (p->g_mod < 0 || (p->ws && ((size_t)p->ws & 15) != 0)) return DG_EINVAL;
  const bool m128 = p->Ci % 128 == 0, n128 = p->Co % 128 == 0;
  if (p->wmode == 0) {
    if (m128 && n128) return launch_dma<0, 128, 128>(p, accumulate, pairs, stream, plan);
    if (m128) return launch_dma<0, 128, 64>(p, accumulate, pairs, stream, plan);
    if (n128) return launch_dma<0, 64, 128>(p, accumulate, pairs, stream, plan);
    return launch_dma<0, 64, 64>(p, accumulate, pairs, stream, plan);
  }
  if (m128 && n128) return launch_dma<1, 128, 128>(p, accumulate, pairs, stream, plan);
  if (m128) return launch_dma<1, 128, 64>(p, accumulate, pairs, stream, plan);
  if (n128) return launch_dma<1, 64, 128>(p, accumulate, pairs, stream, plan);
  return launch_dma<1, 64, 64>(p, accumulate, pairs, stream, plan);
}

// up to GROUP_MAX layers as one launch (see wgrad_group_kernel); every item must take the LDS-DMA kernel.  rounds > 0: the
// group as a whole aims at rounds * 512 workgroups, shared among the items by their FLOPs (each item's K split shrinks
// accordingly: fewer partial tiles to store and to reduce); rounds <= 0: every item keeps the geometry of a launch of its own.
// plans != NULL: fill them (splits, ws_floats, tap_pairs) and launch nothing; else every item brings its workspace
// (DgWgrad.ws sized by the same call with plans).
int dg_wgrad_mfma_dma_group_launch(const WgradP* items, int n, int pairs, int rounds, hipStream_t stream, DgWgradPlan* plans) {
  if (n < 1 || n > GROUP_MAX) return DG_EINVAL;
  GroupP g{};
  int blocks = 0;
  double fl[GROUP_MAX], fsum = 0.0;
  for (int i = 0; i < n; ++i) {
    fl[i] = (double)items[i].B * items[i].Hc * items[i].Wc * items[i].Ci * items[i].Co;
    fsum += fl[i];
  }
  for (int i = 0; i < n; ++i) {
    const WgradP* p = &items[i];
    if (!dg_wgrad_mfma_dma_supported(p) || (!plans && !p->ws)) return DG_EUNSUPPORTED;
    if (p->g_mod < 0 || (!plans && ((size_t)p->ws & 15) != 0)) return DG_EINVAL;
    long target = 512;
    if (rounds > 0) {
      target = (long)(rounds * 512.0 * fl[i] / fsum + 0.5);
      if (target < 64) target = 64;
    }
    const int bm = (p->Ci % 128 == 0 && DG_WG_BM_MAX >= 128) ? 128 : 64, bn = (p->Co % 128 == 0 && DG_WG_BN_MAX >= 128) ? 128 : 64;
    const DmaGeo ge = bm == 128 ? (bn == 128 ? dma_geo<128, 128>(p, 1, pairs, true, target) : dma_geo<128, 64>(p, 1, pairs, true, target))
                                : (bn == 128 ? dma_geo<64, 128>(p, 1, pairs, true, target) : dma_geo<64, 64>(p, 1, pairs, true, target));
    if (plans) {
      plans[i].variant = 5;
      plans[i].splits = (int)ge.split;
      plans[i].ws_floats = ge.split * 16L * p->Ci * p->Co;
      plans[i].tap_pairs = ge.use_pairs ? 1 : 0;
      continue;
    }
    GroupItem& it = g.it[i];
    it.p = x2_strides(*p);
    it.tiles_n = ge.tiles_n;
    it.gx = ge.tiles_m * ge.tiles_n; it.gy = ge.use_pairs ? 8 : 16; it.gz = (int)ge.split;
    it.first = blocks;
    it.variant = variant_code(p->wmode, bm, bn, ge.use_pairs ? 2 : 1, p->a_dtype == DG_BF16X2);
    it.accumulate = 1;
    blocks += (it.gx * it.gy * it.gz + 7) / 8 * 8;
  }
  if (plans) return DG_OK;
  g.n = n;
  wgrad_group_kernel<<<(unsigned)blocks, 256, 0, stream>>>(g);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

extern "C" int dg_wgrad_reduce(const DgWgradReduce* items, int n, void* stream) {
  if (!items || n < 1 || n > RED_MAX) return DG_EINVAL;
  ReduceItems r{};
  int blocks = 0;
  // longest sums first: a thread of a 32-split layer makes four dependent round trips, one of a 4-split layer one - with
  // the deep layers' blocks at the END of the grid the launch finished on a handful of CUs (25.5 us for the D phase's three
  // layers against 15 us for the sum of its parts, scripts/bench_reduce.py)
  int order[RED_MAX];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && items[order[j]].splits > items[order[j - 1]].splits; --j) {
      const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
    }
  for (int i = 0; i < n; ++i) {
    const DgWgradReduce& it = items[order[i]];
    if (!it.ws || !it.dw || it.numel <= 0 || it.numel % 4 != 0 || it.splits < 1) return DG_EINVAL;
    if (((size_t)it.ws & 15) != 0 || ((size_t)it.dw & 15) != 0) return DG_EINVAL;
    r.it[i] = it;
    r.first_block[i] = blocks;
    blocks += (int)((it.numel / 4 + (it.splits > RED_WIDE ? 15 : 255)) / (it.splits > RED_WIDE ? 16 : 256));
  }
  r.first_block[n] = blocks;
  r.n = n;
  wgrad_reduce_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(r);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
